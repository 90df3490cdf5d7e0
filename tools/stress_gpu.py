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
paths = (("waves", cx.PATH_WAVES | cx.K2_LATENCY), ("waves-1w", cx.PATH_WAVES | cx.K2_THROUGHPUT), ("lanes-split", cx.PATH_LANES | cx.LANES_SPLIT), ("lanes-fused", cx.PATH_LANES | cx.LANES_FUSED))
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
# whole batches of 4096-sample frames with a quarter of the frames corrupted: the sixteen-sample tiers of the lane kernels,
# the lanes of a wave leaving them at different samples
def batch_flips(w, tag, trials):
    global n_bad, n_all
    descs = pc.workload_descs(w)
    heads = [cx.parse_frame_header(w.arena[int(w.offs[i]):int(w.offs[i] + w.lens[i])])[2].header_bytes for i in range(w.n)]
    for trial in range(trials):
        rng = np.random.default_rng(seed0 + 77 * trial)
        a = w.arena.copy()
        for i in rng.choice(w.n, size=max(1, w.n // 4), replace=False):
            for _ in range(int(rng.integers(1, 4))):
                lo = int(w.offs[i]) * 8 + heads[i] * 8
                pos = int(rng.integers(lo, (int(w.offs[i]) + int(w.lens[i])) * 8))
                a[pos >> 3] ^= 0x80 >> (pos & 7)
        ref = np.zeros(w.pcm.size, dtype=np.int32)
        r = oracle.decode_batch(a[:w.arena_len], w.offs, w.lens, out=ref, out_offs=w.out_offs, check_crc=False)
        for name, path in paths:
            out, res = GpuBackend(ctx, path).decode(a, w.arena_len, descs, w.out_offs, False, fill=0x13131313)
            n_all += w.n
            for i in range(w.n):
                lo = int(w.out_offs[i]); hi = lo + int(descs["n_channels"][i]) * int(descs["block_size"][i])
                same = (int(res["status"][i]), int(res["msg"][i])) == (int(r["statuses"][i]), int(r["msgs"][i])) and \
                       (res["status"][i] != 0 or (res["end_bit"][i] == r["end_bits"][i] and np.array_equal(out[lo:hi], ref[lo:hi])))
                if not same:
                    n_bad += 1
                    print("MISMATCH (batch)", tag, name, "trial", trial, "frame", i, "oracle", r["statuses"][i], r["msgs"][i], "product", res["status"][i], res["msg"][i], flush=True)
                    os.makedirs("gpurun_out", exist_ok=True)
                    np.save("gpurun_out/badbatch_%s_%s_%d_%d.npy" % (tag, name, trial, i), a[int(w.offs[i]):int(w.offs[i] + w.lens[i])])
                    if n_bad >= 8:
                        print("too many mismatches"); sys.exit(1)
    print("batch flips", tag, "done:", n_all, "decodes,", n_bad, "mismatches", flush=True)


batch_flips(synth.config5_unique(160), "config5", 6)
batch_flips(synth.config4(96), "config4", 4)
batch_flips(pc.range_hop_workload(), "hops", 6)
for n in (257, 1000):
    w = synth.config5_unique(n)
    for name, path in paths:
        pc.check_workload(oracle, GpuBackend(ctx, path), w)
w = synth.config4(300)
for name, path in paths:
    pc.check_workload(oracle, GpuBackend(ctx, path), w)
print("workloads ok; total mismatches:", n_bad)
sys.exit(1 if n_bad else 0)
