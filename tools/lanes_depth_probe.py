"""Whole runs of the lane kernels (fused build) in flight on streams of their own: ms per step over the number in flight.
usage: GPU_MAX_HW_QUEUES=8 python tools/lanes_depth_probe.py [frames] [depths...]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import claxon_amd as cx, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
depths = [int(x) for x in sys.argv[2:]] or [2, 4, 6, 8]
w = synth.config3(n)
ctx = cx.Context(0, wait_s=120)
descs, _ = cx.descs_from_offsets(w.arena[:w.arena_len], w.offs, w.lens)
d_arena = torch.from_numpy(w.arena).cuda()
pcm = torch.from_numpy(w.pcm).cuda()
for path, pname in ((cx.PATH_LANES | cx.LANES_FUSED, "lanes-fused"), (cx.PATH_LANES | cx.LANES_SPLIT, "lanes-split")):
    for k in depths:
        outs = [torch.zeros(w.pcm.size, dtype=torch.int32, device="cuda") for _ in range(k)]
        streams = [torch.cuda.Stream() for _ in range(k)]
        batches = [ctx.plan(descs, w.out_offs, path=path) for _ in range(k)]
        torch.cuda.synchronize()
        def go(count):
            for i in range(count):
                j = i % k
                batches[j].run(d_arena.data_ptr(), w.arena_len, outs[j].data_ptr(), streams[j].cuda_stream)
            torch.cuda.synchronize()
        go(2 * k)
        ok = all(bool(torch.equal(o, pcm)) for o in outs)
        steps = 16 * k
        t = time.perf_counter(); go(steps); dt = (time.perf_counter() - t) / steps
        print("queues %s %s, %2d in flight: %.4f ms/step  %.1f Gsamples/s  bit-exact %s" % (os.environ["GPU_MAX_HW_QUEUES"], pname, k, dt * 1e3, w.total_samples / dt / 1e9, ok), flush=True)
        for b in batches: b.close()
        del outs
