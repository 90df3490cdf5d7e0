"""One batch at a time, cut into slices (round 5): the 10 000 config-3 frames as k batches of 10 000 / k frames, each run
(clx_batch_run: the wave kernels K1 -> K2 -> CRC) on a stream of its own, all started together -- so that the predictor stage of one
slice overlaps the Rice stage of another -- against the one batch of 10 000.  Prints ms per whole step (all slices done)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import claxon_amd as cx, synth
n = 10000
ctx = cx.Context(0, wait_s=120)
w = synth.config3(n)
d_arena = torch.from_numpy(w.arena).cuda()
ref = torch.from_numpy(w.pcm).cuda()
for k in (1, 2, 3, 4):
    bounds = [n * i // k for i in range(k + 1)]
    parts = []
    for i in range(k):
        lo, hi = bounds[i], bounds[i + 1]
        descs, _ = cx.descs_from_offsets(w.arena[:w.arena_len], w.offs[lo:hi], w.lens[lo:hi])
        out_offs = (w.out_offs[lo:hi] - w.out_offs[lo]).astype(np.uint64)
        nsamp = int(w.out_offs[hi - 1] - w.out_offs[lo]) + 2 * 4096
        b = ctx.plan(descs, out_offs, verify_crc=True, path=cx.PATH_WAVES)
        parts.append((b, torch.zeros(nsamp, dtype=torch.int32, device="cuda"), torch.cuda.Stream(), int(w.out_offs[lo]), nsamp))
    def step():
        for b, o, st, _, _ in parts:
            b.run(d_arena.data_ptr(), w.arena_len, o.data_ptr(), st.cuda_stream)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    ts = []
    for _ in range(20):
        torch.cuda.synchronize(); t0 = time.perf_counter(); step(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    ok = all(bool(torch.equal(o, ref[a:a + m])) for _, o, _, a, m in parts) and all(bool(np.all(b.results()["status"] == 0)) for b, *_ in parts)
    print("%d slice(s) on %d stream(s): median %.4f ms  min %.4f ms  exact: %s" % (k, k, 1e3 * float(np.median(ts)), 1e3 * min(ts), ok))
    for b, *_ in parts:
        b.close()
