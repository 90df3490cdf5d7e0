"""The synthetic generator is a miniature FLAC encoder written independently of the oracle:
oracle(decode(encode(pcm))) == pcm is therefore a two-sided check."""
import numpy as np
import pytest

import synth
from claxon_msgs import STATUS


def _decode(oracle, w, nthreads=1):
    out = np.full(w.pcm.size, 0x5a5a5a5a, dtype=np.int32)
    if w.bare_subframes:
        r = oracle.decode_subframes(w.arena[:w.arena_len], w.offs, w.block_sizes, w.bps, out=out, out_offs=w.out_offs)
    else:
        r = oracle.decode_batch(w.arena[:w.arena_len], w.offs, w.lens, out=out, out_offs=w.out_offs,
                                check_crc=True, nthreads=nthreads)
    return r, out


@pytest.mark.parametrize("make", [
    lambda: synth.config2(64), lambda: synth.config3(64), lambda: synth.config4(32),
    lambda: synth.config5_unique(96), lambda: synth.small_mixed(160),
], ids=["config2", "config3", "config4", "config5", "small_mixed"])
def test_roundtrip(oracle, make):
    w = make()
    r, out = _decode(oracle, w, nthreads=2)
    assert np.all(r["statuses"] == STATUS["CLX_OK"]), np.unique(r["msgs"])
    assert r["samples"] == w.total_samples
    assert np.array_equal(out, w.pcm)
    # every frame is consumed exactly: end_bit rounded up to a byte + 2 CRC bytes == frame length
    if not w.bare_subframes:
        assert np.array_equal((r["end_bits"] + 7) // 8 + 2, w.lens.astype(np.uint64))


def test_config_shapes():
    w = synth.config3(8)
    assert w.n == 8 and w.total_samples == 8 * 2 * 4096
    assert set(w.assignments.tolist()) == {synth.CH_MID_SIDE}
    assert w.arena.size % 16 == 0 and w.arena.size >= w.arena_len + 32
    assert w.algorithmic_bytes == w.compressed_bytes + 4 * w.total_samples
    # frame header: ff f8 | c9 (4096, 44.1k) | a8 (M/S, 16 bit)
    assert w.arena[:4].tolist() == [0xff, 0xf8, 0xc9, 0xa8]


def test_restamp(oracle):
    w = synth.config3(2)
    frame = w.arena[int(w.offs[1]):int(w.offs[1] + w.lens[1])].copy()
    info0, s0 = oracle.frame_decode(frame)
    assert synth.lib().synth_restamp_frame(frame.ctypes.data, frame.size, 77) == 1
    info1, s1 = oracle.frame_decode(frame)          # CRC-8 and CRC-16 still valid
    assert info1.status == STATUS["CLX_OK"] and info1.time == 77 * 4096
    assert np.array_equal(s0, s1)
