"""Throughput of the decode step on the other BASELINE workload shapes (bench.py times configs[2] only).
usage: python tools/bench_configs.py [n_frames [paths [configs]]]   -- prints one line per (config, kernel path), after a
bit-exactness check; `paths` / `configs` are comma-separated filters (e.g. waves config4,config5)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import claxon_amd as cx, synth
from parity_cases import workload_descs

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
only_paths = sys.argv[2].split(",") if len(sys.argv) > 2 else None
only_cfgs = sys.argv[3].split(",") if len(sys.argv) > 3 else None
ctx = cx.Context(0, wait_s=120)
makers = [("config2 (mono FIXED-2, k=4, one partition)", lambda: synth.config2(n)),
          ("config3 (stereo LPC-8 M/S, 16 partitions)", lambda: synth.config3(n)),
          ("config4 (24-bit LPC-32, Rice2, wasted bits, mixed)", lambda: synth.config4(n)),
          ("config5 (mixed real-world shapes)", lambda: synth.config5_unique(n))]
for name, make in makers:
    if only_cfgs and name.split()[0] not in only_cfgs:
        continue
    w = make()
    descs = workload_descs(w)
    d_arena = torch.from_numpy(w.arena).cuda()
    d_out = torch.zeros(w.pcm.size, dtype=torch.int32, device="cuda")
    ref = torch.from_numpy(w.pcm).cuda()
    for pname, path in (("waves", cx.PATH_WAVES), ("lanes-split", cx.PATH_LANES | cx.LANES_SPLIT), ("lanes-fused", cx.PATH_LANES | cx.LANES_FUSED)):
        if only_paths and pname not in only_paths:
            continue
        batch = ctx.plan(descs, w.out_offs, path=path)
        for _ in range(3):
            batch.run(d_arena.data_ptr(), w.arena_len, d_out.data_ptr())
        torch.cuda.synchronize()
        ok = bool(np.all(batch.results()["status"] == 0)) and bool(torch.equal(d_out, ref))
        t = time.perf_counter()
        for _ in range(10):
            batch.run(d_arena.data_ptr(), w.arena_len, d_out.data_ptr())
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t) * 100
        batch.set_profiling(True); batch.run(d_arena.data_ptr(), w.arena_len, d_out.data_ptr()); torch.cuda.synchronize()
        km = {k: round(v, 3) for k, v in batch.kernel_times().items()}
        print(f"{name:52s} {pname:11s}: {ms:7.3f} ms  {w.pcm.size / ms / 1e6:7.1f} Gsamples/s  bit_exact={ok}  {km}", flush=True)
        batch.close()
