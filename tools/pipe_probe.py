"""Step time of the wave path when consecutive submissions are pipelined (clx_batch_submit: up to SUBMIT_DEPTH whole runs in
flight on internal streams) against plain runs; checks every output buffer bit for bit.
usage: python tools/pipe_probe.py [frames] [steps] [output buffers]"""
import sys, time
import os
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")     # a hardware queue per internal stream of clx_batch_submit (bench.py does the same)
import numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import claxon_amd as cx, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
nbuf = int(sys.argv[3]) if len(sys.argv) > 3 else cx.SUBMIT_DEPTH          # output buffers in rotation
w = synth.config3(n)
ctx = cx.Context(0, wait_s=120)
descs, _ = cx.descs_from_offsets(w.arena[:w.arena_len], w.offs, w.lens)
d_arena = torch.from_numpy(w.arena).cuda()
outs = [torch.zeros(w.pcm.size, dtype=torch.int32, device="cuda") for _ in range(nbuf)]
pcm = torch.from_numpy(w.pcm).cuda()
stream = torch.cuda.current_stream().cuda_stream
for crc in (False, True):
    b = ctx.plan(descs, w.out_offs, verify_crc=crc, path=cx.PATH_WAVES)
    for mode in ("run", "submit"):
        f = b.run if mode == "run" else b.submit
        for o in outs: o.zero_()
        torch.cuda.synchronize()          # (the library's stream does not wait for torch's)
        for i in range(2 * len(outs)): f(d_arena.data_ptr(), w.arena_len, outs[i % len(outs)].data_ptr(), stream)
        b.flush(stream); torch.cuda.synchronize()
        t = time.perf_counter()
        for i in range(steps): f(d_arena.data_ptr(), w.arena_len, outs[i % len(outs)].data_ptr(), stream)
        b.flush(stream); torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / steps
        res = b.results()
        ok = bool(np.all(res["status"] == 0)) and all(bool(torch.equal(o, pcm)) for o in outs)
        print("crc %-5s %-6s %.4f ms/step  %.1f Gsamples/s  bit-exact %s" % (crc, mode, dt * 1e3, w.total_samples / dt / 1e9, ok))
    b.close()
