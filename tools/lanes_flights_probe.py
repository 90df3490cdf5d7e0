"""The lane kernels through clx_batch_submit (eight submissions in flight on the library's streams): ms per step.
usage: GPU_MAX_HW_QUEUES=16 python tools/lanes_flights_probe.py [frames] [steps]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import claxon_amd as cx, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 128
w = synth.config3(n)
ctx = cx.Context(0, wait_s=120)
descs, _ = cx.descs_from_offsets(w.arena[:w.arena_len], w.offs, w.lens)
d_arena = torch.from_numpy(w.arena).cuda()
pcm = torch.from_numpy(w.pcm).cuda()
stream = torch.cuda.current_stream().cuda_stream
for crc in (False, True):
    b = ctx.plan(descs, w.out_offs, verify_crc=crc, path=cx.PATH_LANES | cx.LANES_FUSED)
    k = b.submit_depth
    outs = [torch.zeros(w.pcm.size, dtype=torch.int32, device="cuda") for _ in range(k)]
    torch.cuda.synchronize()
    for i in range(2 * k): b.submit(d_arena.data_ptr(), w.arena_len, outs[i % k].data_ptr(), stream)
    b.flush(stream); torch.cuda.synchronize()
    ok = bool(np.all(b.results()["status"] == 0)) and all(bool(torch.equal(o, pcm)) for o in outs)
    t = time.perf_counter()
    for i in range(steps): b.submit(d_arena.data_ptr(), w.arena_len, outs[i % k].data_ptr(), stream)
    b.flush(stream); torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / steps
    print("queues %s, lanes-fused through the library, %d in flight, crc %s: %.4f ms/step  %.1f Gsamples/s  bit-exact %s" % (os.environ["GPU_MAX_HW_QUEUES"], k, crc, dt * 1e3, w.total_samples / dt / 1e9, ok), flush=True)
    b.close(); del outs
