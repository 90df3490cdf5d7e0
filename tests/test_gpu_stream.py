"""clx_decode_frames_stream: the host-to-host pipeline (upload | decode | download, three chunks in flight) must give what the
one-shot entry gives -- the oracle's samples, statuses and end bits -- as planar i32, as interleaved 16 / 24 / 32-bit PCM
(lib.rs:473-520: the order FlacSamples walks a block), with pinned and with ordinary host memory, twice in a row on one
context (the second call reuses the first one's device buffers and plans)."""
import numpy as np
import pytest

import claxon_amd as cx
import parity_cases as pc
import synth

pytestmark = pytest.mark.gpu


def interleave_ref(w, ref, statuses, sample_bytes):
    out = np.zeros(ref.size * sample_bytes, dtype=np.uint8)
    for i in range(w.n):
        if statuses[i] != 0:
            continue
        a = int(w.out_offs[i]); c = int(w.channels[i]); bs = int(w.block_sizes[i])
        x = ref[a:a + c * bs].reshape(c, bs).T.reshape(-1).astype("<i4")          # sample-major, channel-minor
        b = x.view(np.uint8).reshape(-1, 4)[:, :sample_bytes]
        out[a * sample_bytes:(a + c * bs) * sample_bytes] = b.reshape(-1)
    return out


def test_stream_decode_matches_oracle(oracle):
    ctx = cx.Context(0, wait_s=120)
    w = synth.concat("mix", [synth.config3(1300), synth.small_mixed(200, seed_off=21), synth.config5_unique(300)])
    arena = w.arena.copy()
    arena[int(w.offs[700] + w.lens[700]) - 2] ^= 0x01                         # a CRC footer that no longer matches
    descs = pc.workload_descs(w)
    ref = np.zeros(w.pcm.size, dtype=np.int32)
    r = oracle.decode_batch(arena[:w.arena_len], w.offs, w.lens, out=ref, out_offs=w.out_offs, nthreads=8)
    assert int(r["statuses"][700]) == cx.FORMAT_ERROR
    bad = r["statuses"] != 0
    for i in np.nonzero(bad)[0]:                                              # samples of failed frames read as zeros
        a = int(w.out_offs[i]); ref[a:a + int(w.channels[i]) * int(w.block_sizes[i])] = 0
    pin_in = cx.PinnedArray(arena.shape, np.uint8)
    pin_in.array[:] = arena
    for pinned in (False, True):
        src = pin_in.array if pinned else arena
        ctx.set_stream_chunk(0 if pinned else 250)      # the default (three chunks) / eight chunks: every slot is reused inside a call
        for rep in range(2):
            out, res = ctx.decode_frames_stream(src[:w.arena_len], descs, w.out_offs, verify_crc=True)
            assert np.array_equal(res["status"], r["statuses"]) and np.array_equal(res["msg"], r["msgs"])
            assert np.array_equal(res["end_bit"][~bad], r["end_bits"][~bad])
            assert np.array_equal(out, ref)
        for sb in (2, 3, 4):
            want = interleave_ref(w, ref, r["statuses"], sb)
            dst = cx.PinnedArray((ref.size * sb,), np.uint8) if pinned else None
            out, res = ctx.decode_frames_stream(src[:w.arena_len], descs, w.out_offs, out=dst.array if dst else None, sample_bytes=sb, verify_crc=True)
            assert np.array_equal(res["status"], r["statuses"])
            assert np.array_equal(out, want), sb
        _, res = ctx.decode_frames_stream(src[:w.arena_len], descs, w.out_offs, verify_crc=True, copy_back=False)
        assert np.array_equal(res["status"], r["statuses"]) and np.array_equal(res["end_bit"][~bad], r["end_bits"][~bad])
    ctx.close()


@pytest.mark.parametrize("n_frames,chunk", [(900, 0), (6000, 2000)])      # wave kernels (small chunks) / lane kernels (large ones)
def test_stream_slots_never_hand_back_an_earlier_call_s_samples(oracle, n_frames, chunk):
    """The context's slot buffers are reused from call to call and are only cleared where they must be (gaps, the narrow stage).
    A first call fills them with real PCM; a second call over the SAME geometry in which many frames fail -- damaged footers,
    damaged subframe headers, truncated descriptors -- must hand back zeros for every failed frame and the oracle's samples for
    the others: nothing of the first call may survive."""
    ctx = cx.Context(0, wait_s=120)
    ctx.set_stream_chunk(chunk)
    w = synth.config3(n_frames)
    descs = pc.workload_descs(w)
    out0, res0 = ctx.decode_frames_stream(w.arena[:w.arena_len], descs, w.out_offs, verify_crc=True)
    assert np.all(res0["status"] == 0) and np.array_equal(out0, w.pcm)
    rng = np.random.default_rng(5)
    arena = w.arena.copy()
    for i in rng.choice(w.n, size=w.n // 3, replace=False):
        lo, hi = int(w.offs[i]), int(w.offs[i] + w.lens[i])
        kind = int(rng.integers(0, 3))
        if kind == 0:
            arena[hi - 1] ^= 0x10                                             # the footer
        elif kind == 1:
            arena[lo + int(descs["header_bytes"][i])] |= 0x80                 # the first subframe's padding bit: "invalid subframe header"
        else:
            pos = int(rng.integers(8 * (lo + int(descs["header_bytes"][i]) + 8), 8 * (hi - 2)))
            arena[pos >> 3] ^= (0x80 >> (pos & 7))                            # anywhere in the body
    ref = np.zeros(w.pcm.size, dtype=np.int32)
    r = oracle.decode_batch(arena[:w.arena_len], w.offs, w.lens, out=ref, out_offs=w.out_offs, nthreads=8)
    bad = r["statuses"] != 0
    assert int(bad.sum()) >= w.n // 4
    for i in np.nonzero(bad)[0]:
        a = int(w.out_offs[i]); ref[a:a + int(w.channels[i]) * int(w.block_sizes[i])] = 0
    for sb in (0, 2):
        out, res = ctx.decode_frames_stream(arena[:w.arena_len], descs, w.out_offs, sample_bytes=sb, verify_crc=True)
        assert np.array_equal(res["status"], r["statuses"]) and np.array_equal(res["msg"], r["msgs"])
        want = ref if sb == 0 else interleave_ref(w, ref, r["statuses"], sb)
        assert np.array_equal(out, want), sb
    ctx.close()
