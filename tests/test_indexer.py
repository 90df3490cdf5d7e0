"""Frame indexer for raw streams: the device indexer must give exactly what the host indexer gives
(descriptors, headers, stop offset); its kernels are also run under the wave simulator against host references."""
import glob
import os

import numpy as np
import pytest

import claxon_amd as cx
import synth
from conftest import FIXTURES


def aligned_padded(data):
    buf = np.zeros(data.size + 96, dtype=np.uint8)
    base = (-buf.ctypes.data) % 16
    al = buf[base:base + data.size + 48]
    al[:data.size] = data
    return al


def streams():
    """(name, bytes, start offset)"""
    out = []
    for name in ("pop.flac", "short.flac", "wasted_bits.flac", "non_subset.flac"):
        data = np.frombuffer(open(os.path.join(FIXTURES, name), "rb").read(), dtype=np.uint8)
        st, _, si, off = cx.read_stream_header(data)
        assert st == cx.OK
        out.append((name, data, off))
    w = synth.config5_unique(120)
    body = w.arena[:w.arena_len].copy()
    out.append(("synthetic", body, 0))
    rng = np.random.default_rng(11)
    # sync-looking garbage inside and behind the stream: false candidates the chain must step over / stop at
    noisy = body.copy()
    tail = rng.integers(0, 256, size=5000, dtype=np.uint8)
    tail[100:102] = [0xff, 0xf8]; tail[2000:2006] = [0xff, 0xf8, 0xc9, 0xa8, 0x00, 0x12]
    out.append(("garbage_tail", np.concatenate([noisy, tail]), 0))
    broken = body.copy()
    broken[int(w.offs[40]) + 300] ^= 0x10                     # frame 40's CRC-16 no longer matches: the chain stops there
    out.append(("broken_frame", broken, 0))
    out.append(("mid_start", body, int(w.offs[7])))
    out.append(("not_a_frame", body, int(w.offs[7]) + 1))
    out.append(("empty", body[:1], 0))
    return out


def test_sim_find_headers_and_span_crc(oracle):
    import ctypes as C
    import simlib
    L = simlib.lib()
    L.sim_find_headers.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint32, C.c_void_p]
    L.sim_span_crc16.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    for name, data, start in streams():
        if data.size > 200000:
            data = data[:200000]
        al = aligned_padded(data)
        cand = np.zeros(4096, dtype=np.uint64)
        cnt = C.c_uint32(0)
        assert L.sim_find_headers(al.ctypes.data, data.size, start, cand.ctypes.data, cand.size, C.byref(cnt)) == 0
        got = np.sort(cand[:cnt.value])
        # reference: every position >= start where the host parser accepts a header (CRC-8 checked); the device filter
        # may only ADD positions (it is a superset; the host re-parses), never miss one
        want = []
        idx = np.nonzero((data[:-1] == 0xff) & ((data[1:] & 0xfe) == 0xf8))[0]
        for p in idx[idx >= start]:
            st, msg, hdr = cx.parse_frame_header(data[int(p):int(p) + 20], check_crc=True)
            if st == cx.OK:
                want.append(int(p))
        assert set(want) <= set(int(x) for x in got), name
        assert len(got) <= len(want) + 2, name
        if len(got) >= 2:
            pos = np.concatenate([got, [data.size]]).astype(np.uint64)
            crc = np.zeros(len(got), dtype=np.uint16)
            assert L.sim_span_crc16(al.ctypes.data, pos.ctypes.data, len(got), crc.ctypes.data) == 0
            for j in range(len(got)):
                assert int(crc[j]) == oracle.crc16(data[int(pos[j]):int(pos[j + 1])]), (name, j)


@pytest.mark.gpu
def test_gpu_indexer_matches_host():
    ctx = cx.Context(0, wait_s=120)
    cases = streams()
    for p in sorted(glob.glob(os.path.join(FIXTURES, "fuzz", "*.flac"))):
        data = np.frombuffer(open(p, "rb").read(), dtype=np.uint8)
        st, _, si, off = cx.read_stream_header(data)
        if st == cx.OK:
            cases.append((os.path.basename(p), data, off))
    assert len(cases) >= 12
    n_frames = 0
    for name, data, start in cases:
        hd, hh, hstop = cx.index_frames(data, start)
        dd, dh, dstop = ctx.index_frames(data, start)
        assert hstop == dstop, name
        assert hd.tobytes() == dd.tobytes() and hh.tobytes() == dh.tobytes(), name
        n_frames += hd.size
    assert n_frames > 150
