"""Loader for the wave simulator build of the production kernels (tests/wavesim)."""
import ctypes as C
import os
import subprocess

import numpy as np

import claxon_amd as cx

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "wavesim")
_SO = os.path.join(_DIR, "libwavesim.so")
_CSRC = os.path.join(os.path.dirname(_DIR), "..", "claxon_amd", "csrc")

SF_DESC_DTYPE = np.dtype([("out_base", "<u8"), ("n", "<u2"), ("lim_log2", "u1"), ("flags", "u1"), ("order", "u1"), ("shift", "u1"), ("wasted", "u1"),
                          ("decor", "u1"), ("coef", "<i2", (32,))])
assert SF_DESC_DTYPE.itemsize == 80


def build(force=False):
    deps = [os.path.join(_DIR, f) for f in ("sim_lib.cpp", "wavesim.h")] + \
           [os.path.join(_CSRC, f) for f in ("clx_kernels.hip", "clx_lanes.hip", "clx_lean.hip", "clx_device.h", "clx_crct.h", "clx_plan.h")] + \
           [os.path.join(_DIR, "fake", "clx_intrin.h"), os.path.join(_DIR, "fake", "clx_k2_dot2.h")]
    if not force and os.path.exists(_SO) and os.path.getmtime(_SO) >= max(os.path.getmtime(d) for d in deps):
        return _SO
    tmp = "%s.%d.tmp" % (_SO, os.getpid())                   # (pytest -n: several workers may build at once -- each to its own name, then a rename)
    subprocess.check_call(["g++", "-O2", "-g", "-std=c++17", "-fPIC", "-shared", "-x", "c++",
                           "-I", os.path.join(_DIR, "fake"), "-I", _CSRC, "-o", tmp, os.path.join(_DIR, "sim_lib.cpp")])
    os.replace(tmp, _SO)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.sim_decode_frames.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_int]
        _lib.sim_interleave.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
    return _lib


def interleave(planar, descs, out_offs, sample_bytes, results=None, pcm=None):
    """K4 under simulation (same contract as claxon_amd.Context.interleave)."""
    planar = np.ascontiguousarray(planar, dtype=np.int32)
    descs = np.ascontiguousarray(descs, dtype=cx.FRAME_DESC_DTYPE)
    out_offs = np.ascontiguousarray(out_offs, dtype=np.uint64)
    n = descs.size
    total = int((out_offs + descs["n_channels"].astype(np.uint64) * descs["block_size"].astype(np.uint64)).max()) if n else 0
    if pcm is None:
        pcm = np.zeros(total * sample_bytes + 8, dtype=np.uint8)
    if results is not None:
        results = np.ascontiguousarray(results, dtype=cx.FRAME_RESULT_DTYPE)
    st = lib().sim_interleave(planar.ctypes.data, descs.ctypes.data, n, out_offs.ctypes.data,
                              results.ctypes.data if results is not None else None, pcm.ctypes.data, sample_bytes)
    assert st == 0
    return pcm[:total * sample_bytes]


def decode(arena, arena_len, descs, out_offs, out=None, verify_crc=False, k1_only=False, fill=0, path=0):
    """Run K1 (+K2, +K3) under simulation.  `arena` must be 16-byte padded beyond arena_len.  With cx.OUT_PCM16 in `path` the output
    is interleaved 16-bit PCM: `out` is then (or is made) an int16 array indexed by the same sample offsets."""
    arena = np.ascontiguousarray(arena, dtype=np.uint8)
    # the simulator reads the arena exactly like the GPU: 16-byte aligned base, padded allocation
    buf = np.zeros(arena.size + 64, dtype=np.uint8)
    base = (-buf.ctypes.data) % 16
    al = buf[base:base + arena.size]
    al[:] = arena
    descs = np.ascontiguousarray(descs, dtype=cx.FRAME_DESC_DTYPE)
    out_offs = np.ascontiguousarray(out_offs, dtype=np.uint64)
    n = descs.size
    total = int((out_offs + descs["n_channels"].astype(np.uint64) * descs["block_size"].astype(np.uint64)).max()) if n else 0
    if out is None:
        if path & cx.OUT_PCM24:          # (packed 24-bit PCM: bytes, frame i's block from byte 3 * out_offs[i])
            out = np.full(3 * total + 16, fill & 0xff, dtype=np.uint8)[:3 * total]
        else:
            out = np.full(total + 8, fill & 0x7fff, dtype=np.int16)[:total] if (path & cx.OUT_PCM16) else np.full(total, fill, dtype=np.int32)
    res = np.zeros(n, dtype=cx.FRAME_RESULT_DTYPE)
    nslots = C.c_uint64(0)
    sfd = np.zeros(int(descs["n_channels"].sum()) + n + 2, dtype=SF_DESC_DTYPE)
    flags = (cx.VERIFY_CRC16 if verify_crc else 0) | path          # (the ABI's flags and nothing else: k1_only is an argument of its own)
    st = lib().sim_decode_frames(al.ctypes.data, arena_len, descs.ctypes.data, n, out.ctypes.data, out_offs.ctypes.data,
                                 res.ctypes.data, flags, sfd.ctypes.data, C.byref(nslots), 1 if k1_only else 0)
    if st != 0:
        raise cx.ClaxonError(cx.API_ERROR, 0, "the simulator rejects this combination of flags (0x%x)" % flags)
    return out, res, sfd[:nslots.value]


def decode_runs(arenas, arena_len, descs, out_offs, verify_crc=False, fill=0, path=0, first_gen=7):
    """Consecutive runs of ONE planned batch on ONE set of scratch under simulation (sim_decode_frames_runs: the multi-run state of
    clx_batch_submit -- generation-tagged marks and CRC parts, slot maps re-dealt run by run, scratch left cleared by
    clx_k_finalize).  Run r decodes arenas[r]; returns [(out, results), ...]."""
    descs = np.ascontiguousarray(descs, dtype=cx.FRAME_DESC_DTYPE)
    out_offs = np.ascontiguousarray(out_offs, dtype=np.uint64)
    n = descs.size
    total = int((out_offs + descs["n_channels"].astype(np.uint64) * descs["block_size"].astype(np.uint64)).max()) if n else 0
    keep, ptrs = [], []
    for a in arenas:
        a = np.ascontiguousarray(a, dtype=np.uint8)
        buf = np.zeros(a.size + 64, dtype=np.uint8)
        base = (-buf.ctypes.data) % 16
        al = buf[base:base + a.size]
        al[:] = a
        keep.append((buf, al)); ptrs.append(al.ctypes.data)
    outs = [np.full(total, fill, dtype=np.int32) for _ in arenas]
    ress = [np.zeros(n, dtype=cx.FRAME_RESULT_DTYPE) for _ in arenas]
    k = len(arenas)
    VP = C.c_void_p * k
    flags = (cx.VERIFY_CRC16 if verify_crc else 0) | path
    L = lib()
    L.sim_decode_frames_runs.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]
    st = L.sim_decode_frames_runs(VP(*ptrs), arena_len, k, descs.ctypes.data, n, VP(*[o.ctypes.data for o in outs]), out_offs.ctypes.data,
                                  VP(*[r.ctypes.data for r in ress]), flags, first_gen & 0xffffffff)
    assert st == 0
    return list(zip(outs, ress))


def decode_pool(arenas, arena_len, descs, out_offs, verify_crc=False, fill=0, path=0, order=None, workers=3):
    """ONE merged launch of len(arenas) runs of one planned batch through clx_k_pool under simulation (sim_decode_frames_pool): every
    run on its own scratch, the tickets taken in the order `order` gives (a permutation of range(tickets); None: as they come) by
    `workers` workgroups.  Returns ([(out, results), ...], stuck) -- stuck: decode tickets that gave up waiting for their run's scan."""
    descs = np.ascontiguousarray(descs, dtype=cx.FRAME_DESC_DTYPE)
    out_offs = np.ascontiguousarray(out_offs, dtype=np.uint64)
    n = descs.size
    total = int((out_offs + descs["n_channels"].astype(np.uint64) * descs["block_size"].astype(np.uint64)).max()) if n else 0
    keep, ptrs = [], []
    for a in arenas:
        a = np.ascontiguousarray(a, dtype=np.uint8)
        buf = np.zeros(a.size + 64, dtype=np.uint8)
        base = (-buf.ctypes.data) % 16
        al = buf[base:base + a.size]
        al[:] = a
        keep.append((buf, al)); ptrs.append(al.ctypes.data)
    outs = [np.full(total, fill, dtype=np.int32) for _ in arenas]
    ress = [np.zeros(n, dtype=cx.FRAME_RESULT_DTYPE) for _ in arenas]
    k = len(arenas)
    VP = C.c_void_p * k
    flags = (cx.VERIFY_CRC16 if verify_crc else 0) | path
    if order is not None:
        order = np.ascontiguousarray(order, dtype=np.uint32)
    stuck = C.c_uint32(0)
    L = lib()
    L.sim_decode_frames_pool.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32,
                                         C.c_void_p, C.c_uint32, C.c_void_p]
    st = L.sim_decode_frames_pool(VP(*ptrs), arena_len, k, descs.ctypes.data, n, VP(*[o.ctypes.data for o in outs]), out_offs.ctypes.data,
                                  VP(*[r.ctypes.data for r in ress]), flags, order.ctypes.data if order is not None else None, workers, C.byref(stuck))
    assert st == 0
    return list(zip(outs, ress)), int(stuck.value)


def pool_tickets(descs, n_runs):
    """(scan tickets, decode tickets) of a merged launch of n_runs runs of the batch `descs` describe (clx_k_pool's numbering: every
    run's scan waves first -- run-major --, then every run's groups of 64 slots)."""
    descs = np.ascontiguousarray(descs, dtype=cx.FRAME_DESC_DTYPE)
    slot = 0
    for ch, a in zip(descs["n_channels"], descs["channel_assignment"]):
        if a != cx.CH_INDEPENDENT and (slot & 1):
            slot += 1
        slot += int(ch)
    n_multi = int(np.sum(descs["n_channels"] > 1))
    return n_runs * ((n_multi + 63) // 64), n_runs * ((slot + 63) // 64)
