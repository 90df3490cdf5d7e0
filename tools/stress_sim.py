"""CPU stress of the PRODUCTION kernel sources under the wave simulator (tests/wavesim): whole batches with a quarter of
the frames corrupted by bit flips (CRC checks off on both sides), against the oracle -- the shapes that exercise K1's
table walks and DPP propagation (short codes, one long partition, streams that never resynchronise) next to the mixed
ones.  Not part of the test suite.   usage: python tools/stress_sim.py [trials] [waves|lanes|lanes-fused ...]"""
import sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import oracle, claxon_amd as cx, synth
import parity_cases as pc
from parity_util import SimBackend

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 6
names = sys.argv[2:] or ["waves"]
sel = {"waves": cx.PATH_WAVES, "lanes": cx.PATH_LANES | cx.LANES_SPLIT, "lanes-fused": cx.PATH_LANES | cx.LANES_FUSED,
       "lanes-fused-pool": cx.PATH_LANES | cx.LANES_FUSED | cx.POOL}      # (round 6: the scan and the 16-bit tier as clx_k_pool's tickets)
n_bad = n_all = 0


def batch_flips(w, tag):
    global n_bad, n_all
    descs = pc.workload_descs(w)
    heads = [cx.parse_frame_header(w.arena[int(w.offs[i]):int(w.offs[i] + w.lens[i])])[2].header_bytes for i in range(w.n)]
    for trial in range(trials):
        rng = np.random.default_rng(9000 + 131 * trial)
        a = w.arena.copy()
        for i in rng.choice(w.n, size=max(1, w.n // 4), replace=False):
            for _ in range(int(rng.integers(1, 4))):
                lo = int(w.offs[i]) * 8 + heads[i] * 8
                pos = int(rng.integers(lo, (int(w.offs[i]) + int(w.lens[i])) * 8))
                a[pos >> 3] ^= 0x80 >> (pos & 7)
        ref = np.zeros(w.pcm.size, dtype=np.int32)
        r = oracle.decode_batch(a[:w.arena_len], w.offs, w.lens, out=ref, out_offs=w.out_offs, check_crc=False)
        for name in names:
            out, res = SimBackend(sel[name]).decode(a, w.arena_len, descs, w.out_offs, False, fill=0x13131313)
            n_all += w.n
            for i in range(w.n):
                lo = int(w.out_offs[i]); hi = lo + int(descs["n_channels"][i]) * int(descs["block_size"][i])
                same = (int(res["status"][i]), int(res["msg"][i])) == (int(r["statuses"][i]), int(r["msgs"][i])) and \
                       (res["status"][i] != 0 or (res["end_bit"][i] == r["end_bits"][i] and np.array_equal(out[lo:hi], ref[lo:hi])))
                if not same:
                    n_bad += 1
                    print("MISMATCH", tag, name, "trial", trial, "frame", i, "oracle", r["statuses"][i], r["msgs"][i],
                          "product", res["status"][i], res["msg"][i], flush=True)
                    np.save("/tmp/bad_%s_%s_%d_%d.npy" % (tag, name, trial, i), a[int(w.offs[i]):int(w.offs[i] + w.lens[i])])
    print("batch flips", tag, "done:", n_all, "decodes,", n_bad, "mismatches", flush=True)


batch_flips(synth.config3(24), "config3")
def one_partition_frames(n=24, bs=4096):
    """config 2's shape (FIXED-2, k = 4 forced, one partition of 4096 codes: spans of 32-bit chunks) as framed mono streams."""
    ws = []
    for i in range(n):
        pcm = synth.pcm_sine_noise(i, bs)[0][None, None]
        fp = synth.FrameParams(0, 0, i)
        fp.sf[0] = synth.sf(synth.SF_FIXED, 2, 0, 0, rice_param=4)
        ws.append(synth.encode_frames("c2", pcm, 1, bs, 16, [fp]))
    return synth.concat("one partition", ws)


batch_flips(one_partition_frames(), "one-partition")
batch_flips(pc.resync_workload(), "resync")
batch_flips(synth.config5_unique(64), "config5")
batch_flips(synth.config4(12), "config4")
print("total mismatches:", n_bad)
sys.exit(1 if n_bad else 0)
