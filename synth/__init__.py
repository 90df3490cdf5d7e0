"""Synthetic FLAC workload generator (test / bench infrastructure).

PCM models follow SURVEY.md §8(d); the C encoder (flacsynth.c) turns them into
valid frames.  Every workload returns a `Workload` with the compressed arena,
the frame index, and the source PCM (= the expected decode, planar i32).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libflacsynth.so")

SF_CONSTANT, SF_VERBATIM, SF_FIXED, SF_LPC = range(4)
CH_INDEPENDENT, CH_LEFT_SIDE, CH_RIGHT_SIDE, CH_MID_SIDE = range(4)
BASE_SEED = 20260925


class SubframeParams(C.Structure):
    _fields_ = [("type", C.c_int32), ("order", C.c_int32), ("qlp_precision", C.c_int32),
                ("partition_order", C.c_int32), ("rice_param", C.c_int32), ("force_rice2", C.c_int32),
                ("wasted", C.c_int32), ("reserved", C.c_int32)]


class FrameParams(C.Structure):
    _fields_ = [("channel_assignment", C.c_int32), ("variable_blocking", C.c_int32),
                ("number", C.c_uint64), ("sf", SubframeParams * 8)]


def build(force=False):
    src = os.path.join(_HERE, "flacsynth.c")
    if not force and os.path.exists(_SO) and os.path.getmtime(_SO) >= os.path.getmtime(src):
        return _SO
    subprocess.check_call(["gcc", "-O3", "-march=native", "-fPIC", "-std=c11", "-shared", "-o", _SO, src, "-lm"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.synth_encode_frames.restype = C.c_size_t
        L.synth_encode_frames.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                          C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p]
        L.synth_encode_subframes.restype = C.c_size_t
        L.synth_encode_subframes.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p,
                                             C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p]
        L.synth_restamp_frame.restype = C.c_int
        L.synth_restamp_frame.argtypes = [C.c_void_p, C.c_size_t, C.c_uint64]
        L.synth_tile_frames.restype = C.c_size_t
        L.synth_tile_frames.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint64, C.c_uint64, C.c_uint64,
                                        C.c_void_p, C.c_size_t, C.c_void_p]
        _lib = L
    return _lib


def sf(type=SF_LPC, order=8, precision=12, partition_order=4, rice_param=-1, force_rice2=0, wasted=-1):
    return SubframeParams(type, order, precision, partition_order, rice_param, force_rice2, wasted, 0)


class Workload:
    """A batch of frames (or bare subframes) in one contiguous arena."""

    def __init__(self, name, arena, offs, lens, channels, block_sizes, bps, assignments, pcm, out_offs,
                 bare_subframes=False, header_bytes=None):
        self.name = name
        self.arena = arena                  # uint8, padded with >=32 zero bytes, len multiple of 16
        self.arena_len = int(offs[-1] + lens[-1]) if len(offs) else 0   # meaningful bytes
        self.offs = np.asarray(offs, dtype=np.uint64)
        self.lens = np.asarray(lens, dtype=np.uint32)
        self.channels = np.asarray(channels, dtype=np.uint8)
        self.block_sizes = np.asarray(block_sizes, dtype=np.uint16)
        self.bps = np.asarray(bps, dtype=np.uint8)
        self.assignments = np.asarray(assignments, dtype=np.uint8)
        self.pcm = pcm                      # flat int32: expected decode, frame i at out_offs[i], planar
        self.out_offs = np.asarray(out_offs, dtype=np.uint64)
        self.bare_subframes = bare_subframes
        self.header_bytes = header_bytes

    @property
    def n(self):
        return int(self.offs.size)

    @property
    def total_samples(self):
        return int((self.channels.astype(np.int64) * self.block_sizes.astype(np.int64)).sum())

    @property
    def compressed_bytes(self):
        return int(self.lens.astype(np.int64).sum())

    @property
    def algorithmic_bytes(self):
        """compressed bytes read once + 4 B per decoded i32 written once."""
        return self.compressed_bytes + 4 * self.total_samples


def _pad_arena(buf, used):
    cap = (used + 15) // 16 * 16 + 64
    out = np.zeros(cap, dtype=np.uint8)
    out[:used] = buf[:used]
    return out


def encode_frames(name, pcm, channels, bs, bps, frame_params, sample_rate=44100):
    """pcm: int32 [n][channels][bs]; frame_params: list/array of FrameParams."""
    pcm = np.ascontiguousarray(pcm, dtype=np.int32)
    n = pcm.shape[0]
    fps = (FrameParams * n)(*frame_params)
    cap = int(n * (channels * bs * ((bps + 8) // 8 + 1) + 64) + 64)
    arena = np.zeros(cap, dtype=np.uint8)
    offs = np.zeros(n, dtype=np.uint64)
    lens = np.zeros(n, dtype=np.uint32)
    used = lib().synth_encode_frames(pcm.ctypes.data, n, channels, bs, bps, sample_rate, C.addressof(fps),
                                     arena.ctypes.data, cap, 0, offs.ctypes.data, lens.ctypes.data)
    if used == 0:
        raise RuntimeError("synth arena overflow")
    out_offs = np.arange(n, dtype=np.uint64) * np.uint64(channels * bs)
    assignments = np.array([fp.channel_assignment for fp in frame_params], dtype=np.uint8)
    return Workload(name, _pad_arena(arena, used), offs, lens, np.full(n, channels), np.full(n, bs),
                    np.full(n, bps), assignments, pcm.reshape(-1), out_offs)


def encode_subframes(name, x, bs, bps, params):
    x = np.ascontiguousarray(x, dtype=np.int32)
    n = x.shape[0]
    ps = (SubframeParams * n)(*params)
    cap = int(n * (bs * ((bps + 8) // 8 + 1) + 64) + 64)
    arena = np.zeros(cap, dtype=np.uint8)
    offs = np.zeros(n, dtype=np.uint64)
    lens = np.zeros(n, dtype=np.uint32)
    used = lib().synth_encode_subframes(x.ctypes.data, n, bs, bps, C.addressof(ps), arena.ctypes.data, cap, 0,
                                        offs.ctypes.data, lens.ctypes.data)
    if used == 0:
        raise RuntimeError("synth arena overflow")
    out_offs = np.arange(n, dtype=np.uint64) * np.uint64(bs)
    return Workload(name, _pad_arena(arena, used), offs, lens, np.ones(n), np.full(n, bs), np.full(n, bps),
                    np.zeros(n), x.reshape(-1), out_offs, bare_subframes=True)


def concat(name, workloads):
    """Concatenate frame workloads (possibly of different shapes) into one arena."""
    arenas, offs, lens, ch, bsz, bps, asg, pcm, oo = [], [], [], [], [], [], [], [], []
    apos, opos = 0, 0
    for w in workloads:
        arenas.append(w.arena[:w.arena_len])
        offs.append(w.offs + np.uint64(apos))
        lens.append(w.lens)
        ch.append(w.channels); bsz.append(w.block_sizes); bps.append(w.bps); asg.append(w.assignments)
        pcm.append(w.pcm)
        oo.append(w.out_offs + np.uint64(opos))
        apos += w.arena_len
        opos += w.pcm.size
    arena = np.concatenate(arenas)
    return Workload(name, _pad_arena(arena, arena.size), np.concatenate(offs), np.concatenate(lens),
                    np.concatenate(ch), np.concatenate(bsz), np.concatenate(bps), np.concatenate(asg),
                    np.concatenate(pcm), np.concatenate(oo))


# ------------------------------------------------------------------------- PCM models

def _rng(index):
    return np.random.Generator(np.random.PCG64(BASE_SEED + int(index)))


def pcm_sine_noise(index, n, amp=(1000, 6000), freq=(40, 200), sigma=4.5, bits=16):
    """x[t] = round(A sin(2 pi f t / 44100 + phi)) + round(N(0, sigma)), clipped to `bits`."""
    g = _rng(index)
    A = g.uniform(*amp); f = g.uniform(*freq); phi = g.uniform(0, 2 * np.pi)
    t = np.arange(n)
    x = np.rint(A * np.sin(2 * np.pi * f * t / 44100.0 + phi)) + np.rint(g.normal(0, sigma, n))
    lim = 1 << (bits - 1)
    return np.clip(x, -lim, lim - 1).astype(np.int32), g


def config2(n=10000, bs=4096):
    """10k independent subframes, 16-bit, FIXED order 2, Rice k=4 forced, partition order 0."""
    x = np.empty((n, bs), dtype=np.int32)
    for i in range(n):
        x[i], _ = pcm_sine_noise(i, bs)
    params = [sf(SF_FIXED, 2, 0, 0, rice_param=4, wasted=0)] * n
    return encode_subframes("config2: %d subframes bs=%d 16-bit FIXED-2 k=4" % (n, bs), x, bs, 16, params)


def pcm_stereo(index, n, bits=16):
    """L = sinusoid+noise (as config 2), R = 0.8 L + independent noise sigma=6."""
    L, g = pcm_sine_noise(index, n, bits=bits)
    R = np.rint(0.8 * L + g.normal(0, 6.0, n))
    lim = 1 << (bits - 1)
    return L, np.clip(R, -lim, lim - 1).astype(np.int32)


def config3(n=10000, bs=4096, order=8, precision=None, partition_order=4):
    """10k stereo 16-bit frames, mid/side, both subframes LPC order 8, coefficient precision 12-14 (frame g
    of the job's index takes 12 + g mod 3), per-partition optimal k."""
    pcm = np.empty((n, 2, bs), dtype=np.int32)
    fps = []
    first = BASE_SEED - 20260925               # bench.py shifts BASE_SEED by the first frame of a rank's range
    for i in range(n):
        pcm[i, 0], pcm[i, 1] = pcm_stereo(i, bs)
        fp = FrameParams(CH_MID_SIDE, 0, i)
        prec = precision if precision is not None else 12 + (first + i) % 3
        fp.sf[0] = sf(SF_LPC, order, prec, partition_order)
        fp.sf[1] = sf(SF_LPC, order, prec, partition_order)
        fps.append(fp)
    return encode_frames("config3: %d stereo frames bs=%d 16-bit M/S LPC-%d P%d" % (n, bs, order, partition_order),
                         pcm, 2, bs, 16, fps)


def config4(n=10000, bs=4096, order=32):
    """24-bit stereo, LPC order 32 (precision 15), mixed channel assignments, partition order 0-8
    per subframe, wasted bits in {0,0,4,8}; side channel at 25 bps => i64 accumulator mandatory.
    `order`: the same with another predictor order (12: what 24-bit music usually carries; the split tier's <= 12-tap kernel)."""
    pcm = np.empty((n, 2, bs), dtype=np.int32)
    fps = []
    t = np.arange(bs)
    for i in range(n):
        g = _rng(10_000_000 + i)
        ws = g.choice([0, 0, 4, 8], size=2)
        chans = []
        base = np.zeros(bs)
        for _ in range(4):
            base += g.uniform(2e5, 1.2e6) * np.sin(2 * np.pi * g.uniform(40, 4000) * t / 44100.0 + g.uniform(0, 2 * np.pi))
        for c in range(2):
            x = (base if c == 0 else 0.7 * base) + g.normal(0, 300.0, bs)
            x = np.clip(np.rint(x), -(1 << 23), (1 << 23) - 1).astype(np.int64)
            w = int(ws[c])
            chans.append(((x >> w) << w).astype(np.int32))
        pcm[i, 0], pcm[i, 1] = chans
        fp = FrameParams(int(g.integers(0, 4)), 0, i)
        for c in range(2):
            # drawn 0-8 as SURVEY §8(d) says; order 32 needs >= 32 samples in the first partition
            # (subframe.rs:275-277), so 8 (16 samples/partition) is clamped to 7
            fp.sf[c] = sf(SF_LPC, order, 15, min(int(g.integers(0, 9)), 7))
        fps.append(fp)
    return encode_frames("config4: %d stereo frames bs=%d 24-bit LPC-%d P0-8 wasted" % (n, bs, order), pcm, 2, bs, 24, fps)


def pcm_music_like(index, n, bits=16):
    """Richer 'real-world-shaped' stereo: a few partials + shaped noise (k lands around 5-9)."""
    g = _rng(20_000_000 + index)
    t = np.arange(n)
    base = np.zeros(n)
    for _ in range(int(g.integers(2, 6))):
        base += g.uniform(300, 5000) * np.sin(2 * np.pi * g.uniform(50, 6000) * t / 44100.0 + g.uniform(0, 2 * np.pi))
    noise = g.normal(0, g.uniform(8, 120), n + 2)
    noise = 0.5 * noise[2:] + 0.3 * noise[1:-1] + 0.2 * noise[:-2]
    L = base + noise
    R = g.uniform(0.3, 1.0) * base + g.normal(0, g.uniform(8, 120), n)
    lim = 1 << (bits - 1)
    return (np.clip(np.rint(L), -lim, lim - 1).astype(np.int32),
            np.clip(np.rint(R), -lim, lim - 1).astype(np.int32), g)


def config5_unique(n_unique=1024, bs=4096, number_base=0):
    """Mixed real-world-shaped stereo 16-bit frames: 88% LPC (orders 1-12, triangular around 8),
    10% FIXED 0-4, 1% CONSTANT, 1% VERBATIM; 40% M/S, 25% L/S, 15% R/S, 20% independent;
    partition order 0-6, optimal k."""
    pcm = np.empty((n_unique, 2, bs), dtype=np.int32)
    fps = []
    for i in range(n_unique):
        L, R, g = pcm_music_like(i, bs)
        ca = int(g.choice([CH_MID_SIDE, CH_LEFT_SIDE, CH_RIGHT_SIDE, CH_INDEPENDENT], p=[0.40, 0.25, 0.15, 0.20]))
        fp = FrameParams(ca, 0, number_base + i)
        for c in range(2):
            u = g.uniform()
            if u < 0.88:
                order = int(np.clip(np.rint(g.triangular(1, 8, 12)), 1, 12))
                fp.sf[c] = sf(SF_LPC, order, int(g.integers(12, 15)), int(g.integers(0, 7)))
            elif u < 0.98:
                fp.sf[c] = sf(SF_FIXED, int(g.integers(0, 5)), 0, int(g.integers(0, 7)))
            elif u < 0.99:
                fp.sf[c] = sf(SF_CONSTANT, 0, 0, 0)
                if c == 0:
                    L = np.full(bs, int(g.integers(-5, 6)), dtype=np.int32)
                else:
                    R = np.full(bs, int(g.integers(-5, 6)), dtype=np.int32)
            else:
                fp.sf[c] = sf(SF_VERBATIM, 0, 0, 0)
                if c == 0:
                    L = g.integers(-32768, 32768, bs).astype(np.int32)
                else:
                    R = g.integers(-32768, 32768, bs).astype(np.int32)
        # a CONSTANT/VERBATIM choice applies to the *coded* channel; keep it simple: only honour
        # them for independent coding, otherwise fall back to FIXED-0 / LPC so the data still round-trips
        if ca != CH_INDEPENDENT:
            for c in range(2):
                if fp.sf[c].type == SF_CONSTANT:
                    fp.sf[c] = sf(SF_FIXED, 0, 0, 0)
                elif fp.sf[c].type == SF_VERBATIM:
                    fp.sf[c] = sf(SF_VERBATIM, 0, 0, 0)
        pcm[i, 0], pcm[i, 1] = L, R
        fps.append(fp)
    return encode_frames("config5: %d unique mixed stereo frames bs=%d" % (n_unique, bs), pcm, 2, bs, 16, fps)


TILE_NUMBER_BASE = 65536      # frame numbers 65536 .. 2097151 all take a 4-byte number field: every tile can be re-stamped in place


class TiledStream:
    """Config 5: `total` frames, frame i = unique frame i % U re-stamped with frame number
    TILE_NUMBER_BASE + i (distinct header, CRC-8 and CRC-16 per frame).  Only the index lives here; `slice(lo, hi)` builds the
    bytes of a contiguous range -- what one rank of a sharded job uploads."""

    def __init__(self, unique, total):
        assert TILE_NUMBER_BASE + total <= 2097152, "frame numbers must keep a 4-byte number field"
        self.unique, self.total = unique, int(total)
        U = unique.n
        idx = np.arange(self.total, dtype=np.int64) % U
        self.lens = unique.lens[idx]
        self.channels = unique.channels[idx]
        self.block_sizes = unique.block_sizes[idx]

    def weights(self):
        return self.lens.astype(np.int64) + 4 * self.channels.astype(np.int64) * self.block_sizes.astype(np.int64)

    def slice(self, lo, hi):
        """Workload of frames [lo, hi): arena bytes, offsets, output offsets (frame order); .pcm is None (the expected
        decode of frame i is unique.pcm of frame i % U -- see `expected_index`)."""
        u = self.unique
        n = int(hi - lo)
        cap = int(self.lens[lo:hi].astype(np.int64).sum()) if n else 0
        arena = np.zeros((cap + 15) // 16 * 16 + 64, dtype=np.uint8)
        offs = np.zeros(max(n, 1), dtype=np.uint64)
        ua = np.ascontiguousarray(u.arena)
        used = lib().synth_tile_frames(ua.ctypes.data, u.offs.ctypes.data, u.lens.ctypes.data, u.n, int(lo), int(hi),
                                       TILE_NUMBER_BASE, arena.ctypes.data, cap, offs.ctypes.data) if n else 0
        if n and used != cap:
            raise RuntimeError("synth_tile_frames failed")
        per = self.channels[lo:hi].astype(np.uint64) * self.block_sizes[lo:hi].astype(np.uint64)
        out_offs = np.concatenate([[0], np.cumsum(per)[:-1]]).astype(np.uint64) if n else np.zeros(0, dtype=np.uint64)
        w = Workload("config5 tiled frames [%d, %d) of %d (%d unique)" % (lo, hi, self.total, u.n), arena, offs[:n], self.lens[lo:hi],
                     self.channels[lo:hi], self.block_sizes[lo:hi], np.full(n, 16), u.assignments[(np.arange(lo, hi) % u.n)],
                     None, out_offs)
        w.expected_index = (np.arange(lo, hi, dtype=np.int64) % u.n)
        return w


def config5_tiled(total=1_000_000, n_unique=16384, bs=4096):
    """The config-5 stream: `n_unique` unique mixed frames (config5_unique) tiled to `total` frames."""
    return TiledStream(config5_unique(n_unique, bs, number_base=TILE_NUMBER_BASE), total)


def small_mixed(n=64, bs=256, seed_off=0):
    """Small ragged mix for CPU-side tests: every subframe type, all channel assignments,
    odd block sizes, 8/12/16/20/24 bps, wasted bits, Rice2."""
    ws = []
    for i in range(n):
        g = _rng(30_000_000 + seed_off + i)
        bps = int(g.choice([8, 12, 16, 20, 24]))
        channels = int(g.choice([1, 2, 2, 2, 3, 6, 8]))
        b = int(g.choice([bs, 16, 17, 100, 192, 255, 256, 576, 1000, 4096, 4410]))
        lim = 1 << (bps - 1)
        pcm = np.empty((1, channels, b), dtype=np.int32)
        t = np.arange(b)
        for c in range(channels):
            x = 0.4 * lim * np.sin(2 * np.pi * g.uniform(50, 3000) * t / 44100.0 + g.uniform(0, 6.28)) \
                + g.normal(0, max(1.0, lim * g.choice([1e-4, 1e-3, 1e-2, 0.2])), b)
            w = int(g.choice([0, 0, 0, 1, 3]))
            x = np.clip(np.rint(x), -lim, lim - 1).astype(np.int64)
            pcm[0, c] = ((x >> w) << w).astype(np.int32)
        ca = int(g.integers(0, 4)) if channels == 2 else 0
        fp = FrameParams(ca, int(g.integers(0, 2)), int(g.integers(0, 1 << 20)))
        for c in range(channels):
            kind = int(g.choice([SF_CONSTANT, SF_VERBATIM, SF_FIXED, SF_FIXED, SF_LPC, SF_LPC, SF_LPC]))
            po_max = 0
            while po_max < 8 and b % (1 << (po_max + 1)) == 0:
                po_max += 1
            if kind == SF_CONSTANT and ca == 0:
                pcm[0, c] = int(g.integers(-lim, lim))
                fp.sf[c] = sf(SF_CONSTANT, 0, 0, 0)
            elif kind == SF_VERBATIM or kind == SF_CONSTANT:
                fp.sf[c] = sf(SF_VERBATIM, 0, 0, 0)
            elif kind == SF_FIXED:
                order = int(g.integers(0, 5))
                po = int(g.integers(0, po_max + 1))
                while (b >> po) < order:
                    po -= 1
                fp.sf[c] = sf(SF_FIXED, min(order, b), 0, po, force_rice2=int(g.integers(0, 2)))
            else:
                order = int(g.integers(1, 33))
                order = min(order, b)
                po = int(g.integers(0, po_max + 1))
                while (b >> po) < order:
                    po -= 1
                fp.sf[c] = sf(SF_LPC, order, int(g.integers(5, 16)), po, force_rice2=int(g.integers(0, 2)))
        ws.append(encode_frames("f%d" % i, pcm, channels, b, bps, [fp]))
    return concat("small_mixed(%d)" % n, ws)
