"""Step time of the decode on the four single-GPU BASELINE workload shapes, over batch sizes and kernel selections -- the data
behind the library's choice of kernels (clx_batch_create) and the check that "auto" stays close to the best column.
usage: python tools/bench_configs.py [sizes [configs]]     e.g.  4000,8000,16000,32000 config4,config5
Prints one line per (config, size): ms per step for every selection (run = one batch at a time; sub = pipelined submissions,
SUBMIT_DEPTH in flight), after a bit-exactness check of each."""
import os, sys, time
import os
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")     # a hardware queue per internal stream of clx_batch_submit (bench.py does the same)
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import claxon_amd as cx, synth
from parity_cases import workload_descs, head

sizes = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [4000, 16000]
only_cfgs = sys.argv[2].split(",") if len(sys.argv) > 2 else None
ctx = cx.Context(0, wait_s=120)
makers = [("config2", synth.config2), ("config3", synth.config3), ("config4", synth.config4), ("config5", synth.config5_unique)]
paths = (("auto", 0), ("waves", cx.PATH_WAVES | cx.K2_LATENCY), ("waves-1w", cx.PATH_WAVES | cx.K2_THROUGHPUT),
         ("lanes-split", cx.PATH_LANES | cx.LANES_SPLIT), ("lanes-fused", cx.PATH_LANES | cx.LANES_FUSED))
stream = torch.cuda.current_stream().cuda_stream
for name, make in makers:
    if only_cfgs and name not in only_cfgs:
        continue
    big = make(max(sizes))
    for n in sizes:
        w = head(big, n)
        descs = workload_descs(w)
        d_arena = torch.from_numpy(w.arena).cuda()
        outs = [torch.zeros(w.pcm.size, dtype=torch.int32, device="cuda") for _ in range(cx.SUBMIT_DEPTH)]
        ref = torch.from_numpy(w.pcm).cuda()
        cols = []
        for pname, path in paths:
            batch = ctx.plan(descs, w.out_offs, path=path)
            res = {}
            for mode in ("run", "sub"):
                f = batch.run if mode == "run" else batch.submit
                for o in outs: o.zero_()
                torch.cuda.synchronize()          # (the library's stream does not wait for torch's)
                for i in range(len(outs)): f(d_arena.data_ptr(), w.arena_len, outs[i].data_ptr(), stream)
                batch.flush(stream); torch.cuda.synchronize()
                ok = bool(np.all(batch.results()["status"] == 0)) and all(bool(torch.equal(o, ref)) for o in outs)
                reps = 40
                t = time.perf_counter()
                for i in range(reps): f(d_arena.data_ptr(), w.arena_len, outs[i % len(outs)].data_ptr(), stream)
                batch.flush(stream); torch.cuda.synchronize()
                res[mode] = ((time.perf_counter() - t) / reps * 1e3, ok)
            batch.set_profiling(True); batch.run(d_arena.data_ptr(), w.arena_len, outs[0].data_ptr(), stream); torch.cuda.synchronize()
            kn = "+".join(k.replace("clx_k_", "") for k in batch.kernel_times().keys())
            batch.close()
            cols.append("%s run %.3f sub %.3f%s%s" % (pname, res["run"][0], res["sub"][0], "" if res["run"][1] and res["sub"][1] else " NOT-BIT-EXACT",
                                                     " [" + kn + "]" if pname == "auto" else ""))
        print("%s n=%-6d bits/sample %.2f | %s" % (name, n, 8.0 * w.compressed_bytes / w.total_samples, " | ".join(cols)), flush=True)
        del d_arena, outs, ref
