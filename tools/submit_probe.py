"""How long does the HOST need to submit a step (clx_batch_submit: events, memsets, kernel launches) compared with what the GPU
needs to run it?  Submits K steps without waiting, takes the time at which the loop returns and the time at which the GPU is done."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np, torch
import claxon_amd as cx, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
ctx = cx.Context(0, wait_s=120)
w = synth.config3(n)
descs, _ = cx.descs_from_offsets(w.arena[:w.arena_len], w.offs, w.lens)
d_arena = torch.from_numpy(w.arena).cuda()
for crc in (False, True):
    b = ctx.plan(descs, w.out_offs, verify_crc=crc)
    depth = b.submit_depth
    outs = [torch.zeros(w.total_samples, dtype=torch.int32, device="cuda") for _ in range(depth)]
    st = torch.cuda.current_stream().cuda_stream
    for i in range(2 * depth):
        b.submit(d_arena.data_ptr(), w.arena_len, outs[i % depth].data_ptr(), st)
    b.flush(st); torch.cuda.synchronize()
    for K in (12, 48, 192):
        t0 = time.perf_counter()
        for i in range(K):
            b.submit(d_arena.data_ptr(), w.arena_len, outs[i % depth].data_ptr(), st)
        t1 = time.perf_counter()
        b.flush(st); torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("frames %d crc %d steps %3d: host submit loop %.3f ms/step, until the GPU is done %.3f ms/step" % (n, crc, K, 1e3 * (t1 - t0) / K, 1e3 * (t2 - t0) / K))
    b.close()
