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
