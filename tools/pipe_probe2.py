"""Two (or more) batches of the same plan in flight on streams of their own, submissions alternating between them: does the
Rice stage of one fill the tail of the other's?  (One batch: tools/pipe_probe.py.)
usage: python tools/pipe_probe2.py [frames] [steps] [batches]"""
import sys, time
import numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import claxon_amd as cx, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
nb = int(sys.argv[3]) if len(sys.argv) > 3 else 2
w = synth.config3(n)
ctx = cx.Context(0, wait_s=120)
descs, _ = cx.descs_from_offsets(w.arena[:w.arena_len], w.offs, w.lens)
d_arena = torch.from_numpy(w.arena).cuda()
pcm = torch.from_numpy(w.pcm).cuda()
for k in range(1, nb + 1):
    streams = [torch.cuda.Stream() for _ in range(k)]
    outs = [[torch.zeros(w.pcm.size, dtype=torch.int32, device="cuda") for _ in range(2)] for _ in range(k)]
    batches = [ctx.plan(descs, w.out_offs, path=cx.PATH_WAVES) for _ in range(k)]
    for mode in ("run", "submit"):
        torch.cuda.synchronize()
        def go(count):
            for i in range(count):
                j = i % k
                b = batches[j]
                (b.run if mode == "run" else b.submit)(d_arena.data_ptr(), w.arena_len, outs[j][(i // k) & 1].data_ptr(), streams[j].cuda_stream)
            for j in range(k): batches[j].flush(streams[j].cuda_stream)
            torch.cuda.synchronize()
        go(4 * k)
        t = time.perf_counter(); go(steps); dt = (time.perf_counter() - t) / steps
        ok = all(bool(np.all(b.results()["status"] == 0)) for b in batches) and all(bool(torch.equal(o, pcm)) for oo in outs for o in oo)
        print("%d batch(es) in flight, %-6s %.4f ms/step  %.1f Gsamples/s  bit-exact %s" % (k, mode, dt * 1e3, w.total_samples / dt / 1e9, ok), flush=True)
    for b in batches: b.close()
