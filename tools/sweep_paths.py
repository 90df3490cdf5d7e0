"""Step time of the decode on BASELINE configs[2] frames (bench.py's workload) over batch sizes and kernel selections: where
the automatic choices in clx_batch_run should switch.  usage: python tools/sweep_paths.py [frames ...]"""
import sys, time
import numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import claxon_amd as cx, synth, bench
from parity_cases import workload_descs

sizes = [int(a) for a in sys.argv[1:]] or [5000, 10000, 16000, 24000, 32000, 48000, 64000]
ctx = cx.Context(0, wait_s=120)
sel = (("waves", cx.PATH_WAVES), ("lanes-split", cx.PATH_LANES | cx.LANES_SPLIT), ("lanes-fused", cx.PATH_LANES | cx.LANES_FUSED))
for n in sizes:
    w = bench._seeded(synth, synth.config3, n, 0)
    descs = workload_descs(w)
    d_arena = torch.from_numpy(w.arena).cuda()
    d_out = torch.zeros(w.pcm.size, dtype=torch.int32, device="cuda")
    ref = torch.from_numpy(w.pcm).cuda()
    line = "%6d frames:" % n
    for pname, path in sel:
        batch = ctx.plan(descs, w.out_offs, path=path)
        for _ in range(3): batch.run(d_arena.data_ptr(), w.arena_len, d_out.data_ptr())
        torch.cuda.synchronize()
        ok = bool(np.all(batch.results()["status"] == 0)) and bool(torch.equal(d_out, ref))
        t = time.perf_counter()
        for _ in range(20): batch.run(d_arena.data_ptr(), w.arena_len, d_out.data_ptr())
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t) * 50
        line += "  %s %.3f ms%s" % (pname, ms, "" if ok else " MISMATCH")
        batch.close()
    print(line, flush=True)
    del d_arena, d_out, ref
