"""Per-wave timeline of clx_k_lean in a merged launch of M runs (needs a library built with -DCLX_TIMELINE -DCLX_TUNING; CLX_TUNE_MERGE=M
CLX_TUNE_STREAMS=1): when the waves started and ended, how long they lived, the shader clock they saw, how many shared a SIMD.
usage: CLAXON_HIP_LIB=... CLX_TUNE_MERGE=M CLX_TUNE_STREAMS=1 python tools/timeline_lean.py M"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import claxon_amd as cx, synth
M = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = 10000
ctx = cx.Context(0, wait_s=120)
w = synth.config3(n)
descs, _ = cx.descs_from_offsets(w.arena[:w.arena_len], w.offs, w.lens)
d_arena = torch.from_numpy(w.arena).cuda()
b = ctx.plan(descs, w.out_offs, verify_crc=True)
outs = [torch.zeros(w.total_samples, dtype=torch.int32, device="cuda") for _ in range(M)]
st = torch.cuda.current_stream().cuda_stream
for r in range(4):
    for i in range(M):
        b.submit(d_arena.data_ptr(), w.arena_len, outs[i].data_ptr(), st)
    b.flush(st); torch.cuda.synchronize()
nw = M * ((2 * n + 63) // 64)
cx.lib().clx_debug_timeline.argtypes = [C.c_int, C.c_void_p, C.c_size_t]
tl = np.zeros((nw, 14), dtype=np.uint64)
assert cx.lib().clx_debug_timeline(3, tl.ctypes.data_as(C.c_void_p), nw) == 0
tl = tl[tl[:, 1] != 0]
r0, r1, c0, c1 = (tl[:, i].astype(np.int64) for i in range(4))
hw = tl[:, 4]
base = r0.min()
start_us, end_us = (r0 - base) / 100.0, (r1 - base) / 100.0
dur = end_us - start_us
mhz = (c1 - c0) / np.maximum(r1 - r0, 1) * 100.0
hwid = (hw & 0xffffffff).astype(np.int64); xcc = (hw >> 32).astype(np.int64) & 0xf
cu = (hwid >> 8) & 0xf; sh = (hwid >> 12) & 1; se = (hwid >> 13) & 0x7; simd = (hwid >> 4) & 3
q = lambda a: "/".join("%.0f" % v for v in np.percentile(a, [0, 10, 50, 90, 100]))
print("clx_k_lean, merged launch of %d runs: %d waves recorded; percentiles 0/10/50/90/100" % (M, tl.shape[0]))
print("  start us %s   end us %s   duration us %s   shader MHz %s" % (q(start_us), q(end_us), q(dur), q(mhz)))
print("  kernel span %.0f us; sum of wave time %.1f ms = %.0f waves resident on average (%.2f per SIMD)" % (end_us.max(), dur.sum() / 1e3, dur.sum() / end_us.max(), dur.sum() / end_us.max() / 1024))
key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
print("  CUs used %d" % np.unique(key).size)
# resident waves over time, and per SIMD: sample at 20 instants
for t in np.linspace(0.05, 0.95, 10) * end_us.max():
    live = (start_us <= t) & (end_us > t)
    ks = key[live] * 4 + simd[live]
    per_simd = np.bincount(np.unique(ks, return_inverse=True)[1]) if live.any() else np.array([0])
    hist = np.bincount(per_simd, minlength=5)
    print("  t = %6.0f us: %4d waves live; SIMDs holding 1/2/3/4 waves: %s; idle SIMDs %d" % (t, live.sum(), hist[1:5].tolist(), 1024 - per_simd.size))
# duration by start order (first-round waves against second-round ones)
order = np.argsort(start_us)
for lo, hi in ((0, 0.25), (0.25, 0.5), (0.5, 0.75), (0.75, 1.0)):
    sel = order[int(lo * order.size): int(hi * order.size)]
    print("  waves %3d-%3d %% by start: start %s  duration %s" % (100 * lo, 100 * hi, q(start_us[sel]), q(dur[sel])))
