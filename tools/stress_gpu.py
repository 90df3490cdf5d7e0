"""GPU stress (not part of the test suite): many more corrupted / truncated frames than the regular tests, on every
kernel selection, against the oracle.  usage: python tools/stress_gpu.py [seed0] [n_seeds]
Frames that decode differently are saved under gpurun_out/ (copy them to tests/golden/regress/ with a fix)."""
import os, sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import oracle, claxon_amd as cx, synth
import parity_cases as pc
from parity_util import GpuBackend, product_frame_decode

seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
n_seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ctx = cx.Context(0, wait_s=120)
paths = (("waves", cx.PATH_WAVES), ("lanes-split", cx.PATH_LANES | cx.LANES_SPLIT), ("lanes-fused", cx.PATH_LANES | cx.LANES_FUSED))
n_bad = n_all = 0
for si in range(n_seeds):
    seed = seed0 + 1000 * si
    for bs in (64, 272, 1152):
        w = synth.small_mixed(24 if bs > 64 else 48, bs=bs, seed_off=seed)
        rng = np.random.default_rng(seed + bs)
        for i in range(w.n):
            fr = w.arena[int(w.offs[i]):int(w.offs[i] + w.lens[i])].copy()
            _, _, h = cx.parse_frame_header(fr)
            for trial in range(24):
                g = fr.copy()
                if trial % 6 == 5:                                   # a truncation instead of bit flips
                    g = g[:int(rng.integers(h.header_bytes + 1, len(g)))].copy()
                else:
                    for _ in range(int(rng.integers(1, 4))):
                        lo = h.header_bytes * 8
                        hi = min(len(g) * 8, lo + 200) if rng.uniform() < 0.5 else len(g) * 8
                        pos = int(rng.integers(lo, hi))
                        g[pos >> 3] ^= 0x80 >> (pos & 7)
                info, ref = oracle.frame_decode(g, False)
                for name, path in paths:
                    st, msg, eb, got, hh = product_frame_decode(GpuBackend(ctx, path), g, False, fill=0x13131313)
                    n_all += 1
                    same = (st, msg) == (info.status, info.msg) and (st != 0 or (eb == info.end_bit and np.array_equal(got, ref)))
                    if not same:
                        n_bad += 1
                        print("MISMATCH", name, "seed", seed, "bs", bs, "frame", i, "trial", trial, "oracle", info.status, info.msg, "product", st, msg, flush=True)
                        os.makedirs("gpurun_out", exist_ok=True)
                        np.save("gpurun_out/bad_%s_%d_%d_%d_%d.npy" % (name, seed, bs, i, trial), g)
                        if n_bad >= 8:
                            print("too many mismatches"); sys.exit(1)
    print("seed", seed, "done:", n_all, "decodes,", n_bad, "mismatches", flush=True)
for n in (257, 1000):
    w = synth.config5_unique(n)
    for name, path in paths:
        pc.check_workload(oracle, GpuBackend(ctx, path), w)
w = synth.config4(300)
for name, path in paths:
    pc.check_workload(oracle, GpuBackend(ctx, path), w)
print("workloads ok; total mismatches:", n_bad)
sys.exit(1 if n_bad else 0)
