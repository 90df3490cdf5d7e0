"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/claxon_hip.h declares,
its host logic (frame header parser, CRC, stream header, frame indexer, message table) agrees with the
oracle / the reference's vectors -- and it refuses to decode without a GPU (no CPU fallback)."""
import os
import re

import numpy as np
import pytest

import claxon_amd as cx
import synth
from claxon_msgs import MSG, MSG_NAME, STATUS
from conftest import FIXTURES, ROOT, fixture_bytes


@pytest.fixture(scope="module")
def L():
    cx.build()
    return cx.lib()


def test_exports_match_header(L):
    hdr = open(os.path.join(ROOT, "include", "claxon_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(clx_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(cx.EXPORTS)
    for name in declared:
        assert hasattr(L, name), name


def test_version_and_messages(L):
    assert L.clx_version() == (0 << 16) | (2 << 8) | 0          # 0.2: metadata blocks, packet descriptors
    # every message id has a string and an error variant; spot-check strings the reference's tests compare
    for name, mid in MSG.items():
        if name in ("CLX_MSG_NONE", "CLX_MSG_COUNT"):
            continue
        assert cx.message(mid) != "" and L.clx_message_status(mid) in (1, 2, 3), name
    assert cx.message(MSG["CLX_MSG_FRAME_CRC_MISMATCH"]) == "frame CRC mismatch"
    assert cx.message(MSG["CLX_MSG_INVALID_RESIDUAL"]) == "invalid residual"
    assert cx.message(MSG["CLX_MSG_UNENCODED_BINARY"]) == "unencoded binary is not yet implemented"
    assert L.clx_message_status(MSG["CLX_MSG_UNENCODED_BINARY"]) == STATUS["CLX_UNSUPPORTED"]
    assert L.clx_message_status(MSG["CLX_MSG_NO_BPS_IN_HEADER"]) == STATUS["CLX_UNSUPPORTED"]


def test_message_strings_match_reference_source():
    """Every message string must occur verbatim in the reference source (only checked where it is mounted)."""
    ref = "/root/reference/src"
    if not os.path.isdir(ref):
        pytest.skip("reference tree not present on this box")
    src = "".join(open(os.path.join(ref, f)).read() for f in os.listdir(ref))
    src = re.sub(r'\\\n\s*', "", src)          # Rust line continuations inside string literals
    for name, mid in MSG.items():
        if name in ("CLX_MSG_NONE", "CLX_MSG_COUNT", "CLX_MSG_UNEXPECTED_EOF"):
            continue
        assert '"%s"' % cx.message(mid) in src, name


def test_crc_vectors(L):
    # crc.rs:197-209
    assert cx.crc8(bytes([0x1f])) == 0x5d and cx.crc8(bytes([4, 1])) == 0x53 and cx.crc8(b"abc") == 0x5f
    assert cx.crc16(bytes([0x1f])) == 0x8041 and cx.crc16(bytes([4, 1])) == 0x1806 and cx.crc16(b"abc") == 0xcadb


def test_frame_header_matches_oracle(oracle, L):
    w = synth.small_mixed(80, seed_off=5)
    for i in range(w.n):
        fr = w.arena[int(w.offs[i]):int(w.offs[i] + w.lens[i])]
        st, msg, h = cx.parse_frame_header(fr)
        info, _ = oracle.frame_decode(fr)
        assert st == 0 and info.status == 0
        assert (h.block_size, h.n_channels, h.bps, h.channel_assignment, h.header_bytes, h.time, h.sample_rate) == \
               (info.block_size, info.channels, info.bps, info.channel_assignment, info.header_bytes, info.time, info.sample_rate)
    # example from the format notes in SURVEY §8: wasted_bits.flac first frame header
    st, msg, h = cx.parse_frame_header(bytes([0xff, 0xf8, 0xc9, 0x08, 0x00, 0x95]))
    assert (st, h.block_size, h.n_channels, h.bps, h.header_bytes) == (0, 4096, 1, 16, 6)


def test_frame_header_errors_match_oracle(oracle, L):
    """Mutate header bytes: the host parser must produce the oracle's (status, message) in every case."""
    w = synth.small_mixed(16, seed_off=9)
    rng = np.random.default_rng(3)
    seen = set()
    for i in range(w.n):
        fr = w.arena[int(w.offs[i]):int(w.offs[i] + w.lens[i])].copy()
        _, _, h = cx.parse_frame_header(fr)
        for trial in range(60):
            g = fr[:h.header_bytes + 4].copy()
            for _ in range(int(rng.integers(1, 3))):
                pos = int(rng.integers(0, h.header_bytes * 8))
                g[pos >> 3] ^= 0x80 >> (pos & 7)
            cut = int(rng.integers(0, len(g) + 1)) if rng.uniform() < 0.3 else len(g)
            g = g[:cut]
            for crc in (True, False):
                st, msg, _ = cx.parse_frame_header(g, crc)
                info, _ = oracle.frame_decode(g, crc)
                if st == 0:
                    # the oracle goes on to decode subframes; only the header verdict is compared
                    assert info.header_bytes > 0
                else:
                    assert (st, msg) == (info.status, info.msg), (g.tobytes().hex(), crc)
                seen.add((st, msg))
    assert len(seen) >= 6, [MSG_NAME[m] for _, m in seen]


def test_stream_header_matches_oracle(oracle, L):
    import glob
    files = sorted(glob.glob(os.path.join(FIXTURES, "*.flac")) + glob.glob(os.path.join(FIXTURES, "fuzz", "*.flac")))
    for p in files:
        data = np.frombuffer(open(p, "rb").read(), dtype=np.uint8)
        st, msg, si, off = cx.read_stream_header(data)
        st2, msg2, si2, off2 = oracle.stream_open(data)
        assert (st, msg) == (st2, msg2), p
        if st == 0:
            assert off == off2
            for f in ("min_block_size", "max_block_size", "min_frame_size", "max_frame_size", "sample_rate",
                      "channels", "bits_per_sample", "samples"):
                assert getattr(si, f) == getattr(si2, f)
            assert bytes(si.md5sum) == bytes(si2.md5sum)
    st, msg, si, off = cx.read_stream_header(fixture_bytes("pop.flac"))
    assert (si.channels, si.bits_per_sample, si.sample_rate, si.samples) == (1, 16, 44100, 100)


def test_index_frames(oracle, L):
    """The host indexer must find exactly the frame starts the oracle's sequential decode visits."""
    w = synth.concat("s", [synth.config5_unique(40), synth.small_mixed(40, seed_off=3)])
    stream = w.arena[:w.arena_len]
    descs, hdrs, stop = cx.index_frames(stream)
    assert stop == w.arena_len and len(descs) == w.n
    assert np.array_equal(descs["byte_off"], w.offs)
    assert np.array_equal(descs["block_size"], w.block_sizes) and np.array_equal(descs["n_channels"], w.channels)
    assert np.array_equal(descs["max_bytes"].astype(np.int64), w.arena_len - w.offs.astype(np.int64))
    # garbage after the last frame stops the chain at the right place
    junk = np.concatenate([stream, np.frombuffer(b"\x00\x01\x02junk", dtype=np.uint8)])
    descs2, _, stop2 = cx.index_frames(junk)
    assert len(descs2) == w.n - 1 and stop2 == int(w.offs[-1])   # the last frame's end can no longer be confirmed
    for name in ("pop.flac", "wasted_bits.flac", "non_subset.flac"):
        data = np.frombuffer(fixture_bytes(name), dtype=np.uint8)
        st, msg, si, off = cx.read_stream_header(data)
        d, h, stop = cx.index_frames(data, off)
        si2, blocks, _, _ = oracle.decode_stream(data)
        assert len(d) == len(blocks) and stop == len(data)


def test_no_cpu_fallback():
    """Without a GPU the product must fail loudly, never decode on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(cx.ClaxonError) as e:
        cx.Context(0)
    assert "no CPU fallback" in str(e.value)
    src = ""
    for root, _, files in os.walk(os.path.join(ROOT, "claxon_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp")):
                src += open(os.path.join(root, f)).read()
    assert "claxon_oracle" not in src and "import oracle" not in src and "libflacsynth" not in src
