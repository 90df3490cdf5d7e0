"""Where a timed region of 20 steps goes (the driver's --steps 20): host time of the 20 submissions, of the flush, and the wait for the
device -- from an idle machine, as every timed region starts.  usage: region_probe.py [steps] [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import claxon_amd as cx, synth
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ctx = cx.Context(0, wait_s=120)
w = synth.config3(10000)
descs, _ = cx.descs_from_offsets(w.arena[:w.arena_len], w.offs, w.lens)
d_arena = torch.from_numpy(w.arena).cuda()
b = ctx.plan(descs, w.out_offs, verify_crc=True)
depth = b.submit_depth
outs = [torch.zeros(w.total_samples, dtype=torch.int32, device="cuda") for _ in range(depth)]
st = torch.cuda.current_stream().cuda_stream
for r in range(reps + 2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    marks = []
    for i in range(steps):
        b.submit(d_arena.data_ptr(), w.arena_len, outs[i % depth].data_ptr(), st)
        marks.append(time.perf_counter())
    t1 = time.perf_counter()
    b.flush(st)
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    if r >= 2:
        first = (marks[b.submit_merge - 1] - t0) * 1e3 if steps >= b.submit_merge else float("nan")
        print("submits %.3f ms (first launch issued at %.3f) | flush %.3f | wait %.3f | region %.3f ms = %.4f per step" % (
            (t1 - t0) * 1e3, first, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t3 - t0) * 1e3, (t3 - t0) * 1e3 / steps))
