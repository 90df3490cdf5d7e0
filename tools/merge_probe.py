"""One merged launch of the fused lane kernels at a time: submit M steps (M <= CLX_SUBMIT_MERGE goes out as one grid), wait, repeat.
Under `rocprofv3 --kernel-trace` this gives the kernels' durations for M runs in one grid without any other launch beside them."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import claxon_amd as cx, synth
M = int(sys.argv[1]) if len(sys.argv) > 1 else 6
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
n = int(sys.argv[3]) if len(sys.argv) > 3 else 10000
ctx = cx.Context(0, wait_s=120)
w = synth.config3(n)
descs, _ = cx.descs_from_offsets(w.arena[:w.arena_len], w.offs, w.lens)
d_arena = torch.from_numpy(w.arena).cuda()
b = ctx.plan(descs, w.out_offs, verify_crc=True)
outs = [torch.zeros(w.total_samples, dtype=torch.int32, device="cuda") for _ in range(max(M, 1))]
st = torch.cuda.current_stream().cuda_stream
for r in range(reps + 2):
    t0 = time.perf_counter()
    for i in range(M):
        b.submit(d_arena.data_ptr(), w.arena_len, outs[i].data_ptr(), st)
    b.flush(st); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if r >= 2:
        print("M %d: %.3f ms per launch group, %.3f ms/step" % (M, dt * 1e3, dt * 1e3 / M))
ref = torch.from_numpy(w.pcm).cuda()
print("exact:", all(bool(torch.equal(o, ref)) for o in outs), "statuses ok:", bool(np.all(b.results()["status"] == 0)))
