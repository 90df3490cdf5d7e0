"""Debug aid: per-wave timeline of the decode kernels.  Needs a library built with -DCLX_TIMELINE:
     CLX_EXTRA_FLAGS=-DCLX_TIMELINE python -c 'import claxon_amd; claxon_amd.build(force=True)'
   usage: python tools/timeline.py [frames] [waves|lanes]
Prints when each kernel's waves started / ended (s_memrealtime, 100 MHz), their shader clock and where they ran --
the kernel's duration is the LAST wave's end, so stragglers and tail waves show up here and nowhere else."""
import ctypes as C, sys, numpy as np, torch
sys.path.insert(0, ".")
import claxon_amd as cx, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
path = sys.argv[2] if len(sys.argv) > 2 else "waves"
w = synth.config3(n)
ctx = cx.Context(0, wait_s=120)
d_arena = torch.from_numpy(w.arena).cuda()
d_out = torch.zeros(w.pcm.size, dtype=torch.int32, device="cuda")
descs, _ = cx.descs_from_offsets(w.arena[:w.arena_len], w.offs, w.lens)
batch = ctx.plan(descs, w.out_offs, path=cx.PATH_WAVES if path == "waves" else cx.PATH_LANES)
for _ in range(5):
    batch.run(d_arena.data_ptr(), w.arena_len, d_out.data_ptr(), 0)
torch.cuda.synchronize()
nslot_waves = (2 * n + 63) // 64
kernels = [(0, "clx_k_residual", n), (1, "clx_k_predict", 8 * ((nslot_waves + 1) // 2))] if path == "waves" else \
          [(2, "clx_k_scan", (n + 63) // 64), (3, "clx_k_lanes", nslot_waves)]
cx.lib().clx_debug_timeline.argtypes = [C.c_int, C.c_void_p, C.c_size_t]
for kid, name, nw in kernels:
    nw = min(nw, 65536)
    tl = np.zeros((nw, 14), dtype=np.uint64)
    assert cx.lib().clx_debug_timeline(kid, tl.ctypes.data_as(C.c_void_p), nw) == 0
    ids = np.nonzero(tl[:, 1] != 0)[0]
    tl = tl[tl[:, 1] != 0]
    r0, r1, c0, c1 = (tl[:, i].astype(np.int64) for i in range(4))
    hw = tl[:, 4]
    base = r0.min()
    start_us, end_us = (r0 - base) / 100.0, (r1 - base) / 100.0
    dur_us = end_us - start_us
    mhz = (c1 - c0) / np.maximum(r1 - r0, 1) * 100.0
    hwid = (hw & 0xffffffff).astype(np.int64); xcc = (hw >> 32).astype(np.int64) & 0xf
    cu = (hwid >> 8) & 0xf; sh = (hwid >> 12) & 1; se = (hwid >> 13) & 0x7
    q = lambda a: "/".join(f"{v:.1f}" for v in np.percentile(a, [0, 50, 90, 99, 100]))
    print(f"{name}: {tl.shape[0]} waves recorded; percentiles 0/50/90/99/100:")
    print(f"   start us {q(start_us)}   end us {q(end_us)}   duration us {q(dur_us)}   shader MHz {np.median(mhz):.0f}")
    key = xcc * 100000 + se * 1000 + sh * 100 + cu
    uniq, cnt = np.unique(key, return_counts=True)
    print(f"   CUs used {uniq.size}; waves per CU min/max {cnt.min()}/{cnt.max()}; sum of wave time {dur_us.sum() / 1e3:.2f} ms"
          f" = {dur_us.sum() / end_us.max() :.0f} waves resident on average")
    if tl[:, 5].any():
        wt = tl[:, 5].astype(np.int64) / np.maximum(c1 - c0, 1)
        print(f"   fraction of wave time inside instrumented waits: all {q(wt * 100)} %")
    if tl[:, 6:14].any():
        ph = tl[:, 6:14].astype(np.float64).sum(axis=0)
        print('   phase shares of wave time %:', ' '.join(f'{v:.1f}' for v in ph / (c1 - c0).sum() * 100))
        print('   raw phase sums per wave:', ' '.join(f'{v:.0f}' for v in ph / tl.shape[0]))
    for i in np.argsort(end_us)[-3:]:
        print(f"     wave {i}: start {start_us[i]:.1f} end {end_us[i]:.1f} us")
    simd = (hwid >> 4) & 3
    # how the waves of shared CUs sit on the SIMDs (a CU with k waves on fewer than min(k, 4) SIMDs makes them share issue slots)
    from collections import Counter
    shape = Counter()
    for u in uniq[cnt > 1]:
        sel = key == u
        shape[tuple(sorted(np.bincount(simd[sel], minlength=4).tolist(), reverse=True))] += 1
    print(f"   SIMD occupancy shapes of CUs with more than one wave: {dict(shape)}")
    if name == "clx_k_predict" and tl[:, 5].any():
        # the eight waves of a workgroup by role: 0,1 predictors, 2,3 finishers (even tiles), 4,5 loaders, 6,7 finishers (odd tiles)
        wt = tl[:, 5].astype(np.int64) / np.maximum(c1 - c0, 1)
        for role, rname in enumerate(("predictor", "finisher even", "loader", "finisher odd")):
            sel = ((ids % 8) >> 1) == role
            if sel.any():
                print(f"   {rname:14s}: {int(sel.sum())} waves, duration us {q(dur_us[sel])}, share of time in barriers % {q(wt[sel] * 100)}")
    if name == "clx_k_predict":
        for u in uniq[cnt > 2][:3]:
            sel = np.nonzero(key == u)[0]
            print("     CU", u, [(int(j), int(simd[j]), round(float(dur_us[j]), 1)) for j in sel])
