"""Parity cases shared by the wave-simulator tests (CPU, `-m "not gpu"`) and the GPU tests (`-m gpu`):
each runs the product path (host header parser + HIP kernels through a backend) and compares it with
the oracle on the same seeded inputs -- bit-exact samples, identical status / message / end_bit."""
import glob
import os

import numpy as np

import claxon_amd as cx
import synth
from claxon_msgs import MSG, MSG_NAME
from conftest import FIXTURES
from parity_util import assert_same_as_oracle, product_frame_decode


def workload_descs(w):
    if w.bare_subframes:
        return cx.descs_for_subframes(w.offs, w.block_sizes, w.bps)
    descs, _ = cx.descs_from_offsets(w.arena[:w.arena_len], w.offs, w.lens)
    return descs


def head(w, k):
    """The first k frames of a workload (frames are laid out back to back, outputs in frame order)."""
    k = min(int(k), w.n)
    used = int(w.offs[k - 1] + w.lens[k - 1]) if k else 0
    total = int(w.out_offs[k - 1]) + int(w.channels[k - 1]) * int(w.block_sizes[k - 1]) if k else 0
    return synth.Workload("%s[:%d]" % (w.name, k), synth._pad_arena(w.arena, used), w.offs[:k], w.lens[:k], w.channels[:k],
                          w.block_sizes[:k], w.bps[:k], w.assignments[:k], w.pcm[:total], w.out_offs[:k],
                          bare_subframes=w.bare_subframes, header_bytes=w.header_bytes)


def check_workload(oracle, backend, w, verify_crc=True):
    """decode(w) == source PCM == oracle decode; statuses OK; end_bit matches the oracle."""
    descs = workload_descs(w)
    out, res = backend.decode(w.arena, w.arena_len, descs, w.out_offs, verify_crc and not w.bare_subframes,
                              fill=0x5a5a5a5a)
    assert np.all(res["status"] == cx.OK), (np.unique(res["status"]), [MSG_NAME[m] for m in np.unique(res["msg"])])
    assert np.array_equal(out[:w.pcm.size], w.pcm)
    ref = np.zeros(w.pcm.size, dtype=np.int32)
    if w.bare_subframes:
        r = oracle.decode_subframes(w.arena[:w.arena_len], w.offs, w.block_sizes, w.bps, out=ref, out_offs=w.out_offs)
    else:
        r = oracle.decode_batch(w.arena[:w.arena_len], w.offs, w.lens, out=ref, out_offs=w.out_offs)
    assert np.array_equal(out[:w.pcm.size], ref)
    assert np.array_equal(res["end_bit"], r["end_bits"])


def check_truncations(oracle, backend, n_frames=8, cuts_per_frame=24, seed=1000):
    """EOF parity: every prefix of a frame must fail (or succeed) exactly as the reference does."""
    w = synth.small_mixed(n_frames, bs=64, seed_off=seed)
    e = edge_workload()
    pick = np.random.default_rng(seed).choice(e.n, size=min(e.n, max(4, n_frames // 2)), replace=False)
    frames = [w.arena[int(w.offs[i]):int(w.offs[i] + w.lens[i])] for i in range(w.n)] + \
             [e.arena[int(e.offs[i]):int(e.offs[i] + e.lens[i])] for i in pick if e.lens[i] < 400]
    seen = set()
    for i, fr in enumerate(frames):
        L = len(fr)
        rng = np.random.default_rng(seed + i)
        cuts = sorted(set(list(range(0, min(L, 12))) + rng.integers(0, L, cuts_per_frame).tolist() + [L - 3, L - 2, L - 1, L]))
        for c in cuts:
            if c < 0:
                continue
            seen.add(assert_same_as_oracle(oracle, backend, fr[:c].copy(), True, "frame %d cut %d/%d" % (i, c, L)))
    assert (cx.OK, 0) in seen and (cx.IO_ERROR, MSG["CLX_MSG_UNEXPECTED_EOF"]) in seen and (cx.END_OF_STREAM, 0) in seen


def check_bitflips(oracle, backend, n_frames=12, trials=16, seed=2000):
    """Garbage in, the SAME garbage (or the same error) out: wrapping arithmetic, reserved values, first error in
    stream order.  CRC checks off on both sides, as under cfg(fuzzing)."""
    w = synth.small_mixed(n_frames, bs=64, seed_off=seed)
    rng = np.random.default_rng(seed)
    seen = set()
    for i in range(w.n):
        fr = w.arena[int(w.offs[i]):int(w.offs[i] + w.lens[i])].copy()
        _, _, h = cx.parse_frame_header(fr)
        for trial in range(trials):
            g = fr.copy()
            for _ in range(int(rng.integers(1, 4))):
                lo = h.header_bytes * 8
                hi = min(len(g) * 8, lo + 200) if rng.uniform() < 0.6 else len(g) * 8
                pos = int(rng.integers(lo, hi))
                g[pos >> 3] ^= (0x80 >> (pos & 7))
            seen.add(assert_same_as_oracle(oracle, backend, g, False, "frame %d trial %d" % (i, trial)))
    return seen


def stream_decode_both(oracle, backend, data, check_crc):
    """FlacReader::new + blocks() loop on both sides; asserts equality frame by frame."""
    data = np.frombuffer(bytes(data), dtype=np.uint8)
    st, msg, si, off = cx.read_stream_header(data)
    st2, msg2, si2, off2 = oracle.stream_open(data)
    assert (st, msg) == (st2, msg2)
    if st != cx.OK:
        return ("open", st, msg), [], None
    assert off == off2 and bytes(si.md5sum) == bytes(si2.md5sum)
    pos, blocks = off, []
    while True:
        info, ref = oracle.frame_decode(data[pos:], check_crc)
        st, msg, eb, got, h = product_frame_decode(backend, data[pos:].copy(), check_crc, fill=13)
        assert (st, msg) == (info.status, info.msg), (pos, st, msg, info.status, info.msg)
        if st != cx.OK:
            return ("end", st, msg), blocks, si
        assert np.array_equal(got, ref) and eb == info.end_bit
        blocks.append((h, got))
        pos += int(info.bytes_consumed)


def check_fixtures(oracle, backend):
    import hashlib
    md5s = {"pop.flac": "68464288fa5e19835516972dcf47223c", "short.flac": "927598b89c89c1129a152eecfc14075e",
            "wasted_bits.flac": "4fbca4cf30f188453c0676e0cd700c71", "non_subset.flac": None,
            "repeated_vorbis_comment.flac": "68464288fa5e19835516972dcf47223c",
            "empty_vorbis_comment.flac": "68464288fa5e19835516972dcf47223c"}
    for name, md5 in md5s.items():
        data = open(os.path.join(FIXTURES, name), "rb").read()
        end, blocks, si = stream_decode_both(oracle, backend, data, True)
        assert end == ("end", cx.END_OF_STREAM, 0), (name, end)
        if md5:
            nbytes = (si.bits_per_sample + 7) // 8
            h = hashlib.md5()
            for hd, s in blocks:
                inter = s.reshape(hd.n_channels, hd.block_size).T.astype("<i4")
                h.update(np.ascontiguousarray(inter.view(np.uint8).reshape(hd.block_size, hd.n_channels, 4)[:, :, :nbytes]).tobytes())
            assert h.hexdigest() == md5, name


def check_fuzz_corpus(oracle, backend):
    seen = set()
    files = sorted(glob.glob(os.path.join(FIXTURES, "fuzz", "*.flac")))
    assert len(files) == 23
    for p in files:
        data = open(p, "rb").read()
        for crc in (True, False):
            end, blocks, _ = stream_decode_both(oracle, backend, data, crc)
            seen.add(end)
    assert len(seen) >= 8


def edge_workload():
    """Block sizes equal to the predictor order (no residual samples at all, yet the residual header and
    its partition parameter are still in the stream), tiny blocks, 32-tap predictors next to constants --
    mixed into one batch so that lanes of one wave hit these edges at different sample indices."""
    rng = np.random.default_rng(99)
    ws = []
    cases = [(8, 8, synth.SF_LPC, 16), (16, 16, synth.SF_LPC, 24), (4, 4, synth.SF_FIXED, 16), (32, 32, synth.SF_LPC, 16),
             (1, 0, synth.SF_FIXED, 8), (1, 1, synth.SF_LPC, 12), (2, 1, synth.SF_FIXED, 20), (24, 3, synth.SF_FIXED, 16),
             (4096, 32, synth.SF_LPC, 24), (40, 8, synth.SF_LPC, 16), (9, 8, synth.SF_LPC, 16), (12, 12, synth.SF_LPC, 16)]
    for rep in range(3):
        for bs, order, kind, bps in cases:
            for ca, channels in ((0, 1), (3, 2), (1, 2), (2, 2), (0, 3)):
                lim = 1 << (bps - 1)
                pcm = rng.integers(-lim // 4, lim // 4, size=(1, channels, bs)).astype(np.int32)
                fp = synth.FrameParams(ca, 0, int(rng.integers(0, 1000)))
                for c in range(channels):
                    fp.sf[c] = synth.sf(kind, order, int(rng.integers(2, 16)), 0, force_rice2=int(rng.integers(0, 2)))
                ws.append(synth.encode_frames("e", pcm, channels, bs, bps, [fp]))
    return synth.concat("edges", ws)


def range_hop_workload(n=72, bs=1024):
    """Signals that wander in and out of the range in which the kernels may evaluate the predictor with 24-bit factors
    (sum|c| * |s| < 2^31; the limit comes from the coefficients, the check is on the data): a loud burst in a quiet
    signal, staggered from frame to frame so that the lanes of a wave cross the limit at different samples; 24-bit and
    16-bit, high-precision order-12 / order-32 predictors, all channel assignments."""
    rng = np.random.default_rng(4242)
    ws = []
    t = np.arange(bs)
    for i in range(n):
        bps = 24 if i % 3 else 16
        full = 1 << (bps - 1)
        loud, quiet = (0.7 * full, 900.0) if bps == 24 else (0.95 * full, 40.0)
        start = 256 + (5 * i) % 64 + (512 if i % 2 else 0) * (i % 5 == 0)
        env = np.where((t >= start) & (t < start + 96 + i % 40), loud, quiet)       # one loud burst, elsewhere quiet
        if i % 9 == 8:
            env = np.full(bs, quiet)                       # some lanes never leave the range
        base = env * np.sin(2 * np.pi * (200 + 13 * i) * t / 44100.0) + rng.normal(0, 6.0, bs)
        chans = [np.clip(np.rint(base * g), -full, full - 1).astype(np.int32) for g in (1.0, 0.55)]
        ca = (0, 3, 1, 2)[i % 4]
        channels = 1 if ca == 0 and i % 8 == 0 else 2
        pcm = np.stack(chans[:channels])[None]
        fp = synth.FrameParams(ca if channels == 2 else 0, 0, i)
        for c in range(channels):
            fp.sf[c] = synth.sf(synth.SF_LPC, 32 if i % 7 == 3 else 12, 15, int(rng.integers(0, 5)))
        ws.append(synth.encode_frames("h", pcm, channels, bs, bps, [fp]))
    return synth.concat("range hops", ws)


def resync_workload():
    """Rice streams that never resynchronise: with parameter k the residual -2^(k-1) (0 for k = 0) is the code `1` +
    k ones, so a partition of it is a run of ones in which a decoder that starts at the wrong bit stays wrong for ever --
    the exit state of every chunk of the wave-parallel decode depends on its entry state (the worst case for the
    propagation of entry states).  Mixed with stretches of ordinary noise, with codes whose remainder is all zeros
    (resynchronises at once), long unary runs, and with every k the table path / the arithmetic walks handle."""
    rng = np.random.default_rng(777)
    ws = []
    for k in (0, 1, 2, 3, 4, 5, 6, 7, 9, 12):
        for bs, po in ((4096, 0), (1024, 2), (256, 0), (192, 0)):
            stuck = 0 if k == 0 else -(1 << (k - 1))
            noise = rng.integers(-(1 << k), (1 << k) + 1, size=bs)
            zeros_rem = (rng.integers(0, 3, size=bs) << k) >> 1 << 1          # even u, multiples of 2^k: remainder bits all zero
            runs = np.where(rng.random(bs) < 0.01, rng.integers(0, 40 << k, size=bs), stuck)
            variants = [np.full(bs, stuck), np.where(np.arange(bs) % 512 < 300, stuck, noise), zeros_rem, runs,
                        np.where(np.arange(bs) < bs // 2, noise, stuck)]
            for ca, channels in ((0, 1), (0, 2)):
                for v in variants:
                    chans = [np.asarray(v, dtype=np.int32), np.asarray(v[::-1], dtype=np.int32)][:channels]
                    pcm = np.stack(chans)[None]
                    fp = synth.FrameParams(ca, 0, len(ws))
                    for c in range(channels):
                        fp.sf[c] = synth.sf(synth.SF_FIXED, 0, 0, po, rice_param=k)
                    ws.append(synth.encode_frames("r", pcm, channels, bs, 16, [fp]))
    # the same with the five-bit parameter field (Rice2, subframe.rs:358-380): small parameters take the same table path
    for k in (0, 3, 5, 6, 15, 20):
        stuck = 0 if k == 0 else -(1 << (min(k, 14) - 1))
        for bs, po in ((512, 3), (100, 0)):
            noise = rng.integers(-(1 << min(k, 13)), (1 << min(k, 13)) + 1, size=bs)
            for v in (np.full(bs, stuck), noise, np.where(np.arange(bs) % 64 < 40, stuck, noise)):
                fp = synth.FrameParams(0, 0, len(ws))
                fp.sf[0] = synth.sf(synth.SF_FIXED, 0, 0, po, rice_param=k, force_rice2=1)
                ws.append(synth.encode_frames("r2", np.asarray(v, dtype=np.int32)[None, None], 1, bs, 16, [fp]))
    return synth.concat("resync", ws)


def lean_workload(scale=1):
    """What the lean fused lane kernel (clx_k_lean: the 16-bit tier, clx_lean.hip) takes -- waves of 64 subframes of <= 16-bit audio in
    rows of one block size that is a multiple of 16 -- arranged so that every one of its tiers and exits is used: lean turns with 4 /
    8 / 12 taps, partition starts at every multiple of four codes, Rice2 parameters, constants and verbatim subframes riding along,
    every channel assignment; and the ways out of a lean turn: history outside the 16-bit range (loud side channels), codes longer
    than 32 bits (spikes under a small parameter), partitions that are not a multiple of four codes long, streams that outrun the
    ring (16 bits per code), subframes that end right after the prologue.  Families are 64 subframes each, so every family is one
    wave; `scale` repeats the lot with other seeds (GPU runs)."""
    rng = np.random.default_rng(31337)
    S = synth
    ws = []

    def family(bs, n_frames, channels, make):
        pcm = np.empty((n_frames, channels, bs), dtype=np.int32)
        fps = []
        for i in range(n_frames):
            chans, fp = make(i)
            for c in range(channels):
                pcm[i, c] = chans[c]
            fps.append(fp)
        ws.append(S.encode_frames("lean", pcm, channels, bs, 16, fps))

    def music(i, bs, loud=1.0):
        L, R, g = S.pcm_music_like(int(rng.integers(0, 1 << 20)) + i, bs)
        return (np.clip(L * loud, -32768, 32767).astype(np.int32), np.clip(R * loud, -32768, 32767).astype(np.int32)), g

    for rep in range(scale):
        # (a) the plain cases: orders <= 4 / <= 8 / <= 12 (one instantiation of the turn each), every channel assignment
        for omax, bs in ((4, 1024), (8, 1024), (12, 1024), (12, 4096)):
            def mk(i, omax=omax, bs=bs):
                (L, R), g = music(i, bs)
                fp = S.FrameParams(i % 4, 0, i)
                for c in range(2):
                    if g.uniform() < 0.75:
                        fp.sf[c] = S.sf(S.SF_LPC, int(g.integers(1, omax + 1)), int(g.integers(5, 16)), int(g.integers(0, 9 if bs >= 4096 else 7)),
                                        force_rice2=int(g.uniform() < 0.15))
                    else:
                        fp.sf[c] = S.sf(S.SF_FIXED, int(g.integers(0, min(omax, 4) + 1)), 0, int(g.integers(0, 7)))
                return (L, R), fp
            family(bs, 32, 2, mk)
        # (b) constants and verbatim subframes among them (independent coding only: the generator codes what it is given)
        def mk_b(i):
            (L, R), g = music(i, 512)
            fp = S.FrameParams(0, 0, i)
            kinds = [S.SF_LPC, S.SF_LPC]
            if i % 5 == 1:
                kinds[i % 2] = S.SF_CONSTANT
            elif i % 5 == 3:
                kinds[(i // 5) % 2] = S.SF_VERBATIM
            ch = [L, R]
            for c in range(2):
                if kinds[c] == S.SF_CONSTANT:
                    ch[c] = np.full(512, int(g.integers(-9, 10)) * (4 if i % 3 == 0 else 1), dtype=np.int32)     # (some with wasted bits)
                    fp.sf[c] = S.sf(S.SF_CONSTANT, 0, 0, 0)
                elif kinds[c] == S.SF_VERBATIM:
                    ch[c] = g.integers(-32768, 32768, 512).astype(np.int32)
                    fp.sf[c] = S.sf(S.SF_VERBATIM, 0, 0, 0)
                else:
                    fp.sf[c] = S.sf(S.SF_LPC, int(g.integers(1, 13)), 12, int(g.integers(0, 6)))
            return ch, fp
        family(512, 32, 2, mk_b)
        # (c) partitions of 18 and 9 codes (144 >> 3, >> 4): edges inside a four -- the slow turn's, then the give-up
        def mk_c(i):
            (L, R), g = music(i, 144)
            fp = S.FrameParams(3 if i % 2 else 1, 0, i)
            for c in range(2):
                fp.sf[c] = S.sf(S.SF_LPC, 8, 12, (3, 4, 2, 1)[i % 4])
            return (L, R), fp
        family(144, 32, 2, mk_c)
        # (d) loud and out of phase: the side channel leaves the 16-bit range -- for four turns in every frame of the first family
        #     (the wave goes through wide turns -- the 24-bit form -- and returns to lean ones), for good in frames of the second
        for burst in (True, False):
            def mk_d(i, burst=burst):
                bs = 2048
                t = np.arange(bs)
                env = np.where((t >= 610) & (t < 660), 30000.0, 400.0) if (burst or i % 3) else np.full(bs, 30000.0)
                a = env * np.sin(2 * np.pi * (180 + 7 * i) * t / 44100.0)
                L = np.clip(np.rint(a + rng.normal(0, 5, bs)), -32768, 32767).astype(np.int32)
                R = np.clip(np.rint(-a + rng.normal(0, 5, bs)), -32768, 32767).astype(np.int32)
                fp = S.FrameParams((3, 1, 2)[i % 3], 0, i)
                for c in range(2):
                    fp.sf[c] = S.sf(S.SF_LPC, 8, 14, 4)
                return (L, R), fp
            family(2048, 32, 2, mk_d)
        # (e) spikes under a small Rice parameter: codes far longer than 32 bits (mono: 64 frames are one wave)
        def mk_e(i):
            bs = 1024
            x = rng.integers(-6, 7, bs).astype(np.int32)
            if i % 4 != 3:
                x[rng.integers(40, bs, size=1 + i % 3)] = rng.integers(3000, 30000) * (1 if i % 2 else -1)
            fp = S.FrameParams(0, 0, i)
            fp.sf[0] = S.sf(S.SF_FIXED, 0, 0, 0, rice_param=2)
            return (x,), fp
        family(1024, 64, 1, mk_e)
        # (f) full-scale noise, 15-16 bits per code: the ring cannot keep up for long
        def mk_f(i):
            bs = 1024
            fp = S.FrameParams(0, 0, i)
            ch = [rng.integers(-32768, 32768, bs).astype(np.int32), rng.integers(-3000, 3000, bs).astype(np.int32)]
            fp.sf[0] = S.sf(S.SF_FIXED, 0, 0, 2, rice_param=14)
            fp.sf[1] = S.sf(S.SF_LPC, 4, 10, 2)
            return ch, fp
        family(1024, 32, 2, mk_f)
        # (h) wasted bits (subframe.rs:216-225): samples that are multiples of 4 / 8 in one channel or both, every channel assignment
        def mk_h(i):
            (L, R), g = music(i, 512, loud=0.2)
            wl, wr = (2, 0, 3, 1)[i % 4], (0, 2, 3, 0)[i % 4]
            L = (L >> wl) << wl; R = (R >> wr) << wr
            fp = S.FrameParams(i % 4 if (wl == wr) else 0, 0, i)       # (side / mid of unequal shifts would not be multiples any more)
            for c in range(2):
                fp.sf[c] = S.sf(S.SF_LPC if i % 3 else S.SF_FIXED, int(g.integers(1, 5)) if i % 3 == 0 else int(g.integers(1, 13)), 12, int(g.integers(0, 5)))
            return (L, R), fp
        family(512, 32, 2, mk_h)
        # (h2) wasted bits that differ between the two channels of a stereo form (the shift happens in front of the decorrelation,
        #      per channel: subframe.rs:216-225 then frame.rs:319-389): left a multiple of 4 beside a free side, a right of multiples of 8 beside
        #      a free side, a side of multiples of 8 beside a free mid
        def mk_h2(i):
            (L, R), g = music(i, 512, loud=0.2)
            form = (1, 2, 3)[i % 3]                         # left/side, right/side, mid/side
            if form == 1: L = (L >> 2) << 2
            elif form == 2: R = (R >> 3) << 3
            else: L = R + (((L - R) >> 3) << 3)
            L = L.astype(np.int32); R = R.astype(np.int32)
            assert L.min() >= -32768 and L.max() <= 32767
            fp = S.FrameParams(form, 0, i)
            for c in range(2):
                fp.sf[c] = S.sf(S.SF_LPC if i % 4 else S.SF_FIXED, int(g.integers(1, 5)) if i % 4 == 0 else int(g.integers(1, 13)), 12, int(g.integers(0, 5)))
            return (L, R), fp
        family(512, 32, 2, mk_h2)
        # (g) blocks that end with (or right after) the prologue
        for bs in (32, 48):
            def mk_g(i, bs=bs):
                (L, R), g = music(i, bs)
                fp = S.FrameParams(i % 4, 0, i)
                for c in range(2):
                    fp.sf[c] = S.sf(S.SF_LPC, int(g.integers(1, 13)), 10, int(g.integers(0, 2)))
                return (L, R), fp
            family(bs, 32, 2, mk_g)
    # (j) last: a wave with idle lanes (10 subframes) and an odd number of tiles -- the dump slots of the pair stores and of the
    #     lone last tile's store
    def mk_j(i):
        (L, R), g = music(i, 48)
        fp = S.FrameParams(i % 4, 0, i)
        for c in range(2):
            fp.sf[c] = S.sf(S.SF_LPC, int(g.integers(1, 13)), 11, int(g.integers(0, 2)))
        return (L, R), fp
    family(48, 5, 2, mk_j)
    return synth.concat("lean tiers", ws)


def lean24_workload(scale=1):
    """What clx_k_lean24 takes (the split tier of the fused lane build, clx_lean.hip): waves of 64 subframes of <= 24-bit audio (or
    of <= 16-bit audio with more than 12 taps) in rows of one block size that is a multiple of 16, >= 64.  Families of 64 subframes
    (one wave each): 24- and 20-bit music with <= 12 and <= 32 taps in every channel assignment, constants and verbatim subframes
    riding along, partitions that end inside a four, 16-bit audio with 13-32 taps, long codes, streams that outrun the ring, wasted
    bits, blocks that end with the prologue, and signals / coefficient rows that leave the range of the split evaluation (the wave
    gives the group up to the general kernels)."""
    rng = np.random.default_rng(424242)
    S = synth
    ws = []

    def family(bs, n_frames, channels, bps, make):
        pcm = np.empty((n_frames, channels, bs), dtype=np.int32)
        fps = []
        for i in range(n_frames):
            chans, fp = make(i)
            for c in range(channels):
                pcm[i, c] = chans[c]
            fps.append(fp)
        ws.append(S.encode_frames("lean24", pcm, channels, bs, bps, fps))

    def music(i, bs, bits, loud=1.0):
        L, R, g = S.pcm_music_like(int(rng.integers(0, 1 << 20)) + i, bs)
        k = loud * (1 << (bits - 16))
        lim = 1 << (bits - 1)
        return (np.clip(L * k, -lim, lim - 1).astype(np.int32), np.clip(R * k, -lim, lim - 1).astype(np.int32)), g

    for rep in range(scale):
        # (a) the plain cases: <= 12 taps and <= 32 taps (one instantiation of the turn each), 24 and 20 bits, every channel assignment
        for omax, bs, bits in ((12, 1024, 24), (32, 1024, 24), (32, 4096, 24), (32, 512, 20)):
            def mk(i, omax=omax, bs=bs, bits=bits):
                (L, R), g = music(i, bs, bits)
                fp = S.FrameParams(i % 4, 0, i)
                for c in range(2):
                    if g.uniform() < 0.8:
                        fp.sf[c] = S.sf(S.SF_LPC, int(g.integers(1 if omax <= 12 else 9, omax + 1)), int(g.integers(6, 16)), int(g.integers(0, 8 if bs >= 4096 else 5)),
                                        force_rice2=int(g.uniform() < 0.3))
                    else:
                        fp.sf[c] = S.sf(S.SF_FIXED, int(g.integers(0, 5)), 0, int(g.integers(0, 5)))
                return (L, R), fp
            family(bs, 32, 2, bits, mk)
        # (b) constants and verbatim subframes among them
        def mk_b(i):
            (L, R), g = music(i, 512, 24)
            fp = S.FrameParams(0, 0, i)
            kinds = [S.SF_LPC, S.SF_LPC]
            if i % 5 == 1:
                kinds[i % 2] = S.SF_CONSTANT
            elif i % 5 == 3:
                kinds[(i // 5) % 2] = S.SF_VERBATIM
            ch = [L, R]
            for c in range(2):
                if kinds[c] == S.SF_CONSTANT:
                    ch[c] = np.full(512, int(g.integers(-500000, 500000)) * (16 if i % 3 == 0 else 1), dtype=np.int32)
                    fp.sf[c] = S.sf(S.SF_CONSTANT, 0, 0, 0)
                elif kinds[c] == S.SF_VERBATIM:
                    ch[c] = g.integers(-(1 << 23), 1 << 23, 512).astype(np.int32)
                    fp.sf[c] = S.sf(S.SF_VERBATIM, 0, 0, 0)
                else:
                    fp.sf[c] = S.sf(S.SF_LPC, int(g.integers(1, 33)), 13, int(g.integers(0, 4)))
            return ch, fp
        family(512, 32, 2, 24, mk_b)
        # (c) partitions of 18 and 9 codes (144 >> 3, >> 4): edges inside a four -- the slow turn's, then the give-up
        def mk_c(i):
            (L, R), g = music(i, 144, 24)
            fp = S.FrameParams(3 if i % 2 else 1, 0, i)
            for c in range(2):
                fp.sf[c] = S.sf(S.SF_LPC, 8, 12, (3, 4, 2, 1)[i % 4])
            return (L, R), fp
        family(144, 32, 2, 24, mk_c)
        # (d) 16-bit audio with 13-32 taps: clx_k_lean leaves it (more than 12 taps), the split tier takes it
        def mk_d(i):
            (L, R), g = music(i, 1024, 16)
            fp = S.FrameParams(i % 4, 0, i)
            for c in range(2):
                fp.sf[c] = S.sf(S.SF_LPC, int(g.integers(13, 33)), int(g.integers(8, 16)), int(g.integers(0, 5)))
            return (L, R), fp
        family(1024, 32, 2, 16, mk_d)
        # (e) spikes under a small Rice parameter: codes far longer than 32 bits (mono: 64 frames are one wave)
        def mk_e(i):
            bs = 1024
            x = rng.integers(-600, 700, bs).astype(np.int32)
            if i % 4 != 3:
                x[rng.integers(40, bs, size=1 + i % 3)] = rng.integers(30000, 300000) * (1 if i % 2 else -1)
            fp = S.FrameParams(0, 0, i)
            fp.sf[0] = S.sf(S.SF_FIXED, 0, 0, 0, rice_param=8)
            return (x,), fp
        family(1024, 64, 1, 24, mk_e)
        # (f) full-scale noise, 23-24 bits per code: the ring cannot keep up for long
        def mk_f(i):
            bs = 1024
            fp = S.FrameParams(0, 0, i)
            ch = [rng.integers(-(1 << 23), 1 << 23, bs).astype(np.int32), rng.integers(-300000, 300000, bs).astype(np.int32)]
            fp.sf[0] = S.sf(S.SF_FIXED, 0, 0, 2, rice_param=22)
            fp.sf[1] = S.sf(S.SF_LPC, 4, 10, 2)
            return ch, fp
        family(1024, 32, 2, 24, mk_f)
        # (h) wasted bits, every channel assignment
        def mk_h(i):
            (L, R), g = music(i, 512, 24, loud=0.2)
            wl, wr = (4, 0, 8, 1)[i % 4], (0, 4, 8, 0)[i % 4]
            L = (L >> wl) << wl; R = (R >> wr) << wr
            fp = S.FrameParams(i % 4 if (wl == wr) else 0, 0, i)
            for c in range(2):
                fp.sf[c] = S.sf(S.SF_LPC if i % 3 else S.SF_FIXED, int(g.integers(1, 5)) if i % 3 == 0 else int(g.integers(1, 33)), 14, int(g.integers(0, 4)))
            return (L, R), fp
        family(512, 32, 2, 24, mk_h)
        # (g) blocks that end with (or right after) the prologue (48 samples for more than 28 taps)
        for bs in (64, 80):
            def mk_g(i, bs=bs):
                (L, R), g = music(i, bs, 24)
                fp = S.FrameParams(i % 4, 0, i)
                for c in range(2):
                    fp.sf[c] = S.sf(S.SF_LPC, int(g.integers(1, 33)), 10, int(g.integers(0, 2)))
                return (L, R), fp
            family(bs, 32, 2, 24, mk_g)
        # (i) full scale and out of phase under 32 taps of 15 bits: a 25-bit side channel and large coefficient sums -- where the
        #     split evaluation's range ends, the wave hands the group to the general kernels
        def mk_i(i):
            bs = 1024
            t = np.arange(bs)
            a = 8.3e6 * np.sin(2 * np.pi * (90 + 11 * i) * t / 44100.0) * (1.0 if i % 2 else np.minimum(1.0, t / 600.0))
            L = np.clip(np.rint(a + rng.normal(0, 50, bs)), -(1 << 23), (1 << 23) - 1).astype(np.int32)
            R = np.clip(np.rint(-a + rng.normal(0, 50, bs)), -(1 << 23), (1 << 23) - 1).astype(np.int32)
            fp = S.FrameParams((3, 1, 2, 0)[i % 4], 0, i)
            for c in range(2):
                fp.sf[c] = S.sf(S.SF_LPC, 32, 15, 3)
            return (L, R), fp
        family(1024, 32, 2, 24, mk_i)
    # (k) one wave of 24-bit and 16-bit frames side by side (clx_k_lean leaves it for the 24-bit lanes' sake), and (l) one of
    #     eight-channel frames (independent channels: no partner lanes)
    for bits in (24, 16):
        def mk_k(i, bits=bits):
            (L, R), g = music(i, 1024, bits)
            fp = S.FrameParams(i % 4, 0, i)
            for c in range(2):
                fp.sf[c] = S.sf(S.SF_LPC, int(g.integers(1, 13)), 12, int(g.integers(0, 4)))
            return (L, R), fp
        family(1024, 16, 2, bits, mk_k)
    def mk_l(i):
        chans = []
        g = None
        for c in range(4):
            (L, R), g = music(8 * i + c, 512, 24, loud=0.5)
            chans += [L, R]
        fp = S.FrameParams(0, 0, i)
        for c in range(8):
            fp.sf[c] = S.sf(S.SF_LPC if c % 3 else S.SF_FIXED, int(g.integers(1, 5)) if c % 3 == 0 else int(g.integers(1, 33)), 13, int(g.integers(0, 3)))
        return chans, fp
    family(512, 8, 8, 24, mk_l)
    # (j) last: a wave with idle lanes and an odd number of tiles
    def mk_j(i):
        (L, R), g = music(i, 80, 24)
        fp = S.FrameParams(i % 4, 0, i)
        for c in range(2):
            fp.sf[c] = S.sf(S.SF_LPC, int(g.integers(1, 33)), 12, int(g.integers(0, 2)))
        return (L, R), fp
    family(80, 3, 2, 24, mk_j)
    return synth.concat("lean24 tiers", ws)


def check_regressions(oracle, backend):
    """Frames that once decoded differently from the oracle on some kernel selection (found by tools/stress_gpu.py)."""
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "regress", "*.npy")))
    assert files
    for p in files:
        assert_same_as_oracle(oracle, backend, np.load(p), False, os.path.basename(p))


def check_footer_read_in_batch(oracle, backend):
    """A corrupted frame whose subframes end inside its last two bytes: the footer read fails (frame.rs:754) even when the
    CRC is not compared and even though, in a batch, the bytes that follow belong to the next frame and are readable."""
    bad = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "regress", "flip_ends_inside_footer.npy"))
    good = synth.config3(2)
    arena = np.concatenate([bad, good.arena])
    offs = np.concatenate([[0], good.offs + np.uint64(len(bad))]).astype(np.uint64)
    lens = np.concatenate([[len(bad)], good.lens]).astype(np.uint32)
    alen = len(bad) + good.arena_len
    descs, _ = cx.descs_from_offsets(arena[:alen], offs, lens, check_crc=False)
    out_offs = np.concatenate([[0], good.out_offs + np.uint64(2 * 4096)]).astype(np.uint64)
    for crc in (False, True):
        out, res = backend.decode(arena, alen, descs, out_offs, crc, fill=0x21212121)
        r = oracle.decode_batch(arena[:alen], offs, lens, out=np.zeros(out.size, dtype=np.int32), out_offs=out_offs, check_crc=crc)
        assert (int(r["statuses"][0]), int(r["msgs"][0])) == (cx.IO_ERROR, MSG["CLX_MSG_UNEXPECTED_EOF"])
        assert [int(x) for x in res["status"]] == [int(x) for x in r["statuses"]]
        assert [int(x) for x in res["msg"]] == [int(x) for x in r["msgs"]]
        assert np.array_equal(out[2 * 4096:2 * 4096 + good.pcm.size], good.pcm)


def check_crc_in_batch(oracle, backend, w, seed=77, frac=0.4, loose_every=0):
    """CRC-16 parity inside a batch: a share of the frames gets bits flipped -- in the payload, in the footer, in the header's
    neighbourhood -- and the frames that still parse must report "frame CRC mismatch" exactly where the reference does (the lean
    kernels' lanes gather the CRC of the frames they decode: clx_crct.h).  `loose_every`: every such frame's descriptor only
    bounds the frame (max_bytes runs into the next frame), as a reader that does not know the frame lengths hands them over."""
    rng = np.random.default_rng(seed)
    arena = w.arena.copy()
    hit = np.zeros(w.n, dtype=bool)
    descs, _ = cx.descs_from_offsets(w.arena[:w.arena_len], w.offs, w.lens, check_crc=False)
    for i in range(w.n):
        if rng.uniform() >= frac:
            continue
        hit[i] = True
        lo, hi = int(w.offs[i]), int(w.offs[i] + w.lens[i])
        body = lo + int(descs["header_bytes"][i])                         # (the headers stay intact: the host has parsed them)
        kind = int(rng.integers(0, 4))
        for _ in range(int(rng.integers(1, 4))):
            if kind == 0:
                pos = int(rng.integers(8 * (hi - 2), 8 * hi))             # the footer itself
            elif kind == 1:
                pos = int(rng.integers(8 * max(body, hi - 40), 8 * hi))   # near the end: the last subframe's tail
            else:
                pos = int(rng.integers(8 * body, 8 * hi))                 # anywhere behind the header
            arena[pos >> 3] ^= (0x80 >> (pos & 7))
    lens = w.lens.copy()
    if loose_every:
        for i in range(0, w.n - 1, loose_every):
            lens[i] = np.uint32(min(int(w.lens[i]) + 9 + i % 23, int(w.offs[i + 1] + w.lens[i + 1] - w.offs[i])))
        descs = descs.copy()
        descs["max_bytes"] = lens
    out, res = backend.decode(arena, w.arena_len, descs, w.out_offs, True, fill=0x31313131)
    ref = np.full(out.size, 0x31313131, dtype=np.int32)
    r = oracle.decode_batch(arena[:w.arena_len], w.offs, lens, out=ref, out_offs=w.out_offs, check_crc=True)
    st, ms = np.asarray(res["status"]), np.asarray(res["msg"])
    bad = np.nonzero((st != r["statuses"]) | (ms != r["msgs"]))[0]
    assert bad.size == 0, [(int(i), int(st[i]), MSG_NAME[int(ms[i])], int(r["statuses"][i]), MSG_NAME[int(r["msgs"][i])]) for i in bad[:6]]
    n_mismatch = int(np.sum(ms == MSG["CLX_MSG_FRAME_CRC_MISMATCH"]))
    assert n_mismatch >= max(1, int(hit.sum()) // 4), (n_mismatch, int(hit.sum()))
    assert np.all(st[~hit] == cx.OK)
    for i in np.nonzero(st == cx.OK)[0]:
        a, b = int(w.out_offs[i]), int(w.out_offs[i]) + int(w.channels[i]) * int(w.block_sizes[i])
        assert np.array_equal(out[a:b], ref[a:b]) and int(res["end_bit"][i]) == int(r["end_bits"][i]), i
    return n_mismatch


def giveup_workload(n=128, seed=4242):
    """Stereo 16-bit frames of 4608 samples in families of 32 (one wave of 64 subframes each), alternating: partitions of 144 codes
    (the lean kernels' turns all the way) and partitions of 18 codes (a partition edge inside a four at every other edge: slow turn
    after slow turn, so the wave gives its group up -- CLN_SLOW_BUDGET -- and the general kernels decode it from the start)."""
    S = synth
    bs = 4608
    pcm = np.empty((n, 2, bs), dtype=np.int32)
    fps = []
    for i in range(n):
        L, R, _ = S.pcm_music_like(seed + i, bs)
        pcm[i, 0], pcm[i, 1] = L, R
        fp = S.FrameParams(3 if i % 3 else 0, 0, i)
        po = 8 if (i // 32) % 2 else 5
        for c in range(2):
            fp.sf[c] = S.sf(S.SF_LPC, 8, 12, po)
        fps.append(fp)
    return S.encode_frames("give-up families", pcm, 2, bs, 16, fps)


def crc_share_workload(seed=9090):
    """Frames whose CRC-16 the decode lanes gather in shares of every shape (clx_crct.h): 1, 2, 3, 6 and 8 channels (a frame's shares
    chain over up to eight lanes, and its lanes straddle waves), short blocks (32 .. 160 samples: subframes that begin and end inside one
    16-byte granule, frames that start at every offset inside their first granule), constant and verbatim subframes among the predicted
    ones, 16- and 24-bit audio.  Every family is a run of frames of one shape that fills whole waves of 64 subframes, so that the lean
    kernels take them (three-channel frames then sit across wave boundaries)."""
    S = synth
    ws = []
    for k, (channels, bs, bps, count) in enumerate([(1, 64, 16, 64), (2, 32, 16, 64), (3, 96, 16, 64), (6, 64, 16, 32), (8, 160, 16, 16),
                                                    (2, 64, 24, 64), (8, 64, 24, 16), (2, 4096, 16, 32)]):
        pcm = np.empty((count, channels, bs), dtype=np.int32)
        fps = []
        lim = 1 << (bps - 1)
        t = np.arange(bs)
        for i in range(count):
            g = np.random.default_rng(seed + 1000 * k + i)
            for c in range(channels):
                x = 0.3 * lim * np.sin(2 * np.pi * g.uniform(50, 3000) * t / 44100.0 + g.uniform(0, 6.28)) + g.normal(0, lim * g.choice([1e-4, 1e-3, 1e-2]), bs)
                pcm[i, c] = np.clip(np.rint(x), -lim, lim - 1).astype(np.int32)
            ca = int(g.integers(0, 4)) if channels == 2 else 0
            fp = S.FrameParams(ca, 0, 7000 + i)
            for c in range(channels):
                u = g.uniform()
                if ca == 0 and u < 0.08:
                    pcm[i, c] = int(g.integers(-100, 100)); fp.sf[c] = S.sf(S.SF_CONSTANT, 0, 0, 0)
                elif ca == 0 and u < 0.16:
                    fp.sf[c] = S.sf(S.SF_VERBATIM, 0, 0, 0)
                elif u < 0.4:
                    fp.sf[c] = S.sf(S.SF_FIXED, int(g.integers(0, 5)), 0, int(g.integers(0, 2)))
                else:
                    fp.sf[c] = S.sf(S.SF_LPC, int(g.integers(1, 13)), int(g.integers(8, 15)), int(g.integers(0, 2)))
            fps.append(fp)
        ws.append(S.encode_frames("crc shares %d ch bs %d %d bit" % (channels, bs, bps), pcm, channels, bs, bps, fps))
    return synth.concat("crc shares", ws)


def pcm16_workload():
    """What the narrow output (CLX_OUT_PCM16) has to get right: stereo frames the lean kernel writes as whole interleaved lines, of every
    channel assignment, with constant / verbatim subframes riding along, block sizes that are odd multiples of 16 (a lone last tile) and
    16 mod 32; waves that give their group up (the general kernels decode into staging rows and narrow them themselves: clx_narrow_row); waves of
    mono frames (the lean kernel's since round 6: 64-byte rows); three-channel frames and blocks that are no multiple of 16 (never the
    lean kernel's: the general kernels' way)."""
    S = synth
    rng = np.random.default_rng(606)
    parts = [lean_workload(), giveup_workload(128), S.config5_unique(96)]
    for ch, bs, n in ((1, 4096, 70), (3, 1152, 24), (2, 1000, 40), (2, 4096 + 16, 64), (2, 48, 64)):
        t = np.arange(bs)
        pcm = np.empty((n, ch, bs), dtype=np.int32)
        for i in range(n):
            for c in range(ch):
                pcm[i, c] = np.clip(np.round(3000.0 * np.sin(2 * np.pi * (60 + 13 * i + 7 * c) * t / 44100.0 + 0.3 * c) + rng.normal(0, 5.0, bs)), -32768, 32767)
        fp = [S.FrameParams() for _ in range(n)]
        for i, f in enumerate(fp):
            f.channel_assignment = (i % 4) if ch == 2 else 0
            for c in range(ch):
                f.sf[c] = S.sf(S.SF_LPC if (i + c) % 3 else S.SF_FIXED, order=8 if (i + c) % 3 else 2, precision=12, partition_order=min(3, max(0, int(np.log2(bs)) - 5)))
        parts.append(S.encode_frames("pcm16 %dch bs%d" % (ch, bs), pcm, ch, bs, 16, fp))
    return S.concat("pcm16", parts)


def check_pcm16(oracle, backend, w, damage=0.0, seed=1):
    """The narrow output against the oracle: every OK frame's bytes are the interleaved low 16 bits of the oracle's samples; statuses,
    messages and end bits as in planar mode.  `backend.path` must carry cx.OUT_PCM16."""
    rng = np.random.default_rng(seed)
    arena = w.arena.copy()
    descs, _ = cx.descs_from_offsets(w.arena[:w.arena_len], w.offs, w.lens, check_crc=False)
    for i in range(w.n):
        if rng.uniform() < damage:
            lo, hi = int(w.offs[i]) + int(descs["header_bytes"][i]), int(w.offs[i] + w.lens[i])
            pos = int(rng.integers(8 * lo, 8 * hi))
            arena[pos >> 3] ^= (0x80 >> (pos & 7))
    out, res = backend.decode(arena, w.arena_len, descs, w.out_offs, True, fill=0x1111)
    ref = np.zeros(w.pcm.size, dtype=np.int32)
    r = oracle.decode_batch(arena[:w.arena_len], w.offs, w.lens, out=ref, out_offs=w.out_offs, check_crc=True)
    st, ms = np.asarray(res["status"]), np.asarray(res["msg"])
    assert np.array_equal(st, r["statuses"]) and np.array_equal(ms, r["msgs"])
    ok = np.nonzero(st == cx.OK)[0]
    assert np.array_equal(np.asarray(res["end_bit"])[ok], r["end_bits"][ok])
    out = np.asarray(out).view(np.int16) if np.asarray(out).dtype != np.int16 else np.asarray(out)
    for i in ok:
        a, c, bs = int(w.out_offs[i]), int(w.channels[i]), int(w.block_sizes[i])
        want = ref[a:a + c * bs].reshape(c, bs).T.reshape(-1).astype(np.int16)
        assert np.array_equal(out[a:a + c * bs], want), "frame %d (%d ch, bs %d)" % (int(i), c, bs)
    return int(ok.size)


def pcm24_workload():
    """What the packed 24-bit output (CLX_OUT_PCM24, round 6) has to get right -- every frame goes through the general kernels' staging
    rows and clx_narrow_row: 24-bit stereo frames of every channel assignment and 32 taps (config 4's), 16-bit frames in the same
    batch, mono and three-channel frames, blocks that are no multiple of 16, wasted bits."""
    S = synth
    return S.concat("pcm24", [S.config4(64), S.config5_unique(64), lean24_workload(), S.small_mixed(60)])


def check_pcm24(oracle, backend, w, damage=0.0, seed=1):
    """The packed 24-bit output against the oracle: every OK frame's bytes are the interleaved low 24 bits of the oracle's samples, little
    endian, from byte 3 * out_offs[i]; statuses, messages and end bits as in planar mode.  `backend.path` must carry cx.OUT_PCM24."""
    rng = np.random.default_rng(seed)
    arena = w.arena.copy()
    descs, _ = cx.descs_from_offsets(w.arena[:w.arena_len], w.offs, w.lens, check_crc=False)
    for i in range(w.n):
        if rng.uniform() < damage:
            lo, hi = int(w.offs[i]) + int(descs["header_bytes"][i]), int(w.offs[i] + w.lens[i])
            pos = int(rng.integers(8 * lo, 8 * hi))
            arena[pos >> 3] ^= (0x80 >> (pos & 7))
    out, res = backend.decode(arena, w.arena_len, descs, w.out_offs, True, fill=0x11)
    ref = np.zeros(w.pcm.size, dtype=np.int32)
    r = oracle.decode_batch(arena[:w.arena_len], w.offs, w.lens, out=ref, out_offs=w.out_offs, check_crc=True)
    st, ms = np.asarray(res["status"]), np.asarray(res["msg"])
    assert np.array_equal(st, r["statuses"]) and np.array_equal(ms, r["msgs"])
    ok = np.nonzero(st == cx.OK)[0]
    assert np.array_equal(np.asarray(res["end_bit"])[ok], r["end_bits"][ok])
    out = np.asarray(out).view(np.uint8)
    for i in ok:
        a, c, bs = int(w.out_offs[i]), int(w.channels[i]), int(w.block_sizes[i])
        v = ref[a:a + c * bs].reshape(c, bs).T.reshape(-1).astype(np.int32).view(np.uint32)
        want = np.stack([v & 0xff, (v >> 8) & 0xff, (v >> 16) & 0xff], axis=1).astype(np.uint8).reshape(-1)
        assert np.array_equal(out[3 * a:3 * (a + c * bs)], want), "frame %d (%d ch, bs %d)" % (int(i), c, bs)
    return int(ok.size)


def ms_wild_workload(n=96, bs=1024, seed=31337):
    """Mid/side 16-bit frames (order 8) whose quantisation shift has been lowered after encoding: the residual's codes still parse, the
    predictor's loop gain is 2, 4 or 2^shift, and the samples run away -- past the 16-bit turns' range, past the wide turns', past 2^29,
    wrapping.  What the reference computes from such a stream is defined (wrapping arithmetic: subframe.rs:575-582, frame.rs:371-389)
    and the oracle computes it; clx_k_lean's short stereo forms are exact inside the turns' range checks only, so its waves must take
    slow turn after slow turn and leave the group to the general kernels, whose stereo step is exact for every value.  Every third
    family of 32 frames is left intact.  Returns (workload, patched arena)."""
    S = synth
    pcm = np.empty((n, 2, bs), dtype=np.int32)
    fps = []
    for i in range(n):
        L, R, _ = S.pcm_music_like(seed + i, bs)
        pcm[i, 0], pcm[i, 1] = L, R
        fp = S.FrameParams(3, 0, i)
        for c in range(2):
            fp.sf[c] = S.sf(S.SF_LPC, 8, 12, 3)
        fps.append(fp)
    w = S.encode_frames("mid/side, shifts lowered", pcm, 2, bs, 16, fps)
    arena = w.arena.copy()
    descs, _ = cx.descs_from_offsets(w.arena[:w.arena_len], w.offs, w.lens, check_crc=False)
    for i in range(n):
        fam = (i // 32) % 3
        if fam == 2:
            continue
        # subframe 0: 8 header bits, 8 warm-up samples of 16 bits, 4 bits of precision, then the 5-bit shift
        pos = 8 * (int(w.offs[i]) + int(descs["header_bytes"][i])) + 8 + 8 * 16 + 4
        v = 0
        for k in range(5):
            v = (v << 1) | ((int(arena[(pos + k) >> 3]) >> (7 - ((pos + k) & 7))) & 1)
        assert 2 <= v < 16, v
        nv = (0 if i % 4 == 3 else v - 1 - (i % 2)) if fam == 0 else v - 1        # family 0: gains 2, 4, 2^shift; family 1: gain 2 only (runs away late)
        for k in range(5):
            byte, bit = (pos + k) >> 3, 7 - ((pos + k) & 7)
            arena[byte] = (int(arena[byte]) & ~(1 << bit)) | (((nv >> (4 - k)) & 1) << bit)
    return w, arena


def check_ms_wild(oracle, backend, bs=1024):
    """(bs: short blocks -- 256 -- are the ones whose waves never give a group up by the slow turns' budget: 'not when the end is near'.)"""
    w, arena = ms_wild_workload(bs=bs)
    descs, _ = cx.descs_from_offsets(w.arena[:w.arena_len], w.offs, w.lens, check_crc=False)
    out, res = backend.decode(arena, w.arena_len, descs, w.out_offs, False, fill=0x5a5a5a5a)
    ref = np.zeros(w.pcm.size, dtype=np.int32)
    r = oracle.decode_batch(arena[:w.arena_len], w.offs, w.lens, out=ref, out_offs=w.out_offs, check_crc=False)
    assert np.array_equal(np.asarray(res["status"]), r["statuses"]) and np.array_equal(np.asarray(res["msg"]), r["msgs"])
    assert np.all(r["statuses"] == cx.OK)
    assert np.array_equal(np.asarray(res["end_bit"]), r["end_bits"])
    assert int(np.abs(ref.astype(np.int64)).max()) > (1 << 29), "the samples were meant to run away"
    assert np.array_equal(np.asarray(out)[:w.pcm.size], ref)
    return int(np.count_nonzero(ref != w.pcm))


def ms_mover_workload(lone_tail=False):
    """Waves in which EVERY lane belongs to a plain mid/side pair (no wasted bits): clx_k_lean stages mid and side as decoded and its movers
    undo the pair (cln_ms4, round 6).  Whole pairs of tiles and a lone last tile (block sizes 16 mod 32), the shortest blocks, side channels
    loud enough for wide turns and for a 17th bit, every predictor build (<= 4 / 8 / 12 taps), a verbatim channel among them; families of whole
    waves, and a ragged last wave (rows that do not exist: the movers' per-row stores) -- of whole pairs of tiles, or (`lone_tail`) of a
    block size that ends in a lone tile."""
    S = synth
    rng = np.random.default_rng(2025)
    parts = []
    fams = [(4096, 64, 8, 1.0), (4096 + 16, 32, 12, 1.0), (48, 64, 4, 1.0), (1024, 32, 8, 9.0), (2048, 32, 12, 9.0), (32, 64, 2, 1.0)]
    fams.append((1024 + 16, 33, 8, 1.0) if lone_tail else (4096, 40, 8, 1.0))
    for bs, n, omax, loud in fams:
        pcm = np.empty((n, 2, bs), dtype=np.int32)
        fps = []
        for i in range(n):
            L, R, g = S.pcm_music_like(int(rng.integers(0, 1 << 20)) + i, bs)
            L = np.clip(L * loud, -32768, 32767).astype(np.int32)
            R = np.clip(R * (-loud if loud > 1.0 and i % 2 else loud), -32768, 32767).astype(np.int32)      # (out of phase: the side channel is the loud one)
            fp = S.FrameParams(3, 0, i)
            for c in range(2):
                po = min(int(g.integers(0, 5)), max(0, int(np.log2(bs)) - 5))
                while (bs >> po) % 4:                        # (partitions of whole fours: anything else is the slow turn's, and then the general kernels')
                    po -= 1
                fp.sf[c] = S.sf(S.SF_LPC if (i + c) % 4 else S.SF_FIXED, int(g.integers(1, omax + 1)) if (i + c) % 4 else int(g.integers(0, min(omax, 4) + 1)),
                                int(g.integers(8, 15)), po)
            if bs == 48 and i % 9 == 4:
                fp.sf[1] = S.sf(S.SF_VERBATIM, 0, 0, 0)
            pcm[i, 0], pcm[i, 1] = L, R
            fps.append(fp)
        parts.append(S.encode_frames("mid/side bs%d" % bs, pcm, 2, bs, 16, fps))
    return S.concat("mid/side for the movers", parts)


def ms_mover24_workload():
    """The same for the split tier (clx_k_lean24): waves of plain mid/side pairs of 24- and 20-bit audio, <= 12 and <= 32 taps, whole pairs of tiles
    and a lone last tile, a ragged last wave."""
    S = synth
    rng = np.random.default_rng(2424)
    parts = []
    for bs, n, omax, bits in ((1024, 32, 12, 24), (1024 + 16, 32, 32, 24), (4096, 32, 32, 24), (512, 32, 32, 20), (1024, 40, 32, 24)):
        pcm = np.empty((n, 2, bs), dtype=np.int32)
        fps = []
        lim = 1 << (bits - 1)
        for i in range(n):
            L, R, g = S.pcm_music_like(int(rng.integers(0, 1 << 20)) + i, bs)
            k = 1 << (bits - 16)
            pcm[i, 0] = np.clip(L * k, -lim, lim - 1)
            pcm[i, 1] = np.clip(R * (-k if i % 3 == 0 else k), -lim, lim - 1)                 # (every third frame out of phase: a loud side channel)
            fp = S.FrameParams(3, 0, i)
            for c in range(2):
                po = min(int(g.integers(0, 5)), max(0, int(np.log2(bs)) - 5))
                while (bs >> po) % 4:
                    po -= 1
                fp.sf[c] = S.sf(S.SF_LPC if (i + c) % 5 else S.SF_FIXED, int(g.integers(1 if omax <= 12 else 9, omax + 1)) if (i + c) % 5 else int(g.integers(0, 5)),
                                int(g.integers(8, 16)), po, force_rice2=int(g.uniform() < 0.3))
            fps.append(fp)
        parts.append(S.encode_frames("mid/side %d bits bs%d" % (bits, bs), pcm, 2, bs, bits, fps))
    return S.concat("mid/side for the split tier's movers", parts)
