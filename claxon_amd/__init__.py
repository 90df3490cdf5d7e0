"""claxon_amd -- MI355X (gfx950) batched FLAC frame decoder behind Claxon's FrameReader/Block API.

This package is a thin ctypes binding of the C ABI in ``include/claxon_hip.h``
(implemented by ``claxon_amd/csrc`` as ``libclaxon_hip.so``: hand-written HIP
kernels + a C++ host layer).  It exists so that tests and ``bench.py`` can drive
the library; the product is the shared library.

There is no CPU decode path: creating a :class:`Context` raises when the
library or a gfx950 device is missing.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
# CLAXON_HIP_LIB selects another build of the same library (e.g. one compiled with -DCLX_TIMELINE for tools/timeline.py)
LIB_PATH = os.environ.get("CLAXON_HIP_LIB") or os.path.join(_HERE, "libclaxon_hip.so")

OK, IO_ERROR, FORMAT_ERROR, UNSUPPORTED, END_OF_STREAM, API_ERROR = range(6)
CH_INDEPENDENT, CH_LEFT_SIDE, CH_RIGHT_SIDE, CH_MID_SIDE = range(4)
ARENA_ON_DEVICE, OUT_ON_DEVICE, VERIFY_CRC16, PATH_WAVES, PATH_LANES, PCM_ON_DEVICE, LANES_FUSED, LANES_SPLIT = 1, 2, 4, 8, 16, 32, 64, 128
K2_LATENCY, K2_THROUGHPUT = 256, 512
LANES_GENERAL = 1024        # the fused lane build without clx_k_lean (the 16-bit tier): every group through the general kernels
COMPOSE, NO_COMPOSE = 2048, 4096   # waves composed by content (clx_k_compose) forced on / off (default: by the descriptors)
OUT_PCM16 = 8192            # planned batches: the output buffers hold interleaved little-endian 16-bit PCM, written by the decode itself
POOL = 16384                # pipelined submissions: the scan and the 16-bit tier as clx_k_pool's tickets (round 6's other launch form; off by default)
OUT_PCM24 = 32768           # the same with packed 24-bit samples (3 bytes each), written by the general lane kernels
SUBMIT_DEPTH = 24           # CLX_SUBMIT_DEPTH: the most submissions a Batch keeps in flight (Batch.submit_depth: this batch's)


class ClaxonError(RuntimeError):
    """Mirrors claxon::Error (error.rs:18-32): .status is the variant, .message the reference's string."""

    def __init__(self, status, msg=0, text=None):
        self.status, self.msg = status, msg
        self.message = text if text is not None else (message(msg) if _lib is not None else "")
        super().__init__("status %d: %s" % (status, self.message))


class FrameDesc(C.Structure):
    _fields_ = [("byte_off", C.c_uint64), ("max_bytes", C.c_uint32), ("header_bytes", C.c_uint16),
                ("block_size", C.c_uint16), ("n_channels", C.c_uint8), ("channel_assignment", C.c_uint8),
                ("bps", C.c_uint8), ("reserved", C.c_uint8 * 5)]


class FrameResult(C.Structure):
    _fields_ = [("status", C.c_int32), ("msg", C.c_uint32), ("end_bit", C.c_uint64)]


class FrameHeader(C.Structure):
    _fields_ = [("time", C.c_uint64), ("sample_rate", C.c_uint32), ("frame_or_sample_lo", C.c_uint32),
                ("block_size", C.c_uint16), ("header_bytes", C.c_uint16), ("n_channels", C.c_uint8),
                ("channel_assignment", C.c_uint8), ("bps", C.c_uint8), ("variable_blocking", C.c_uint8)]


class StreamInfo(C.Structure):
    _fields_ = [("min_block_size", C.c_uint16), ("max_block_size", C.c_uint16),
                ("min_frame_size", C.c_uint32), ("max_frame_size", C.c_uint32),
                ("sample_rate", C.c_uint32), ("channels", C.c_uint32), ("bits_per_sample", C.c_uint32),
                ("samples", C.c_uint64), ("md5sum", C.c_uint8 * 16)]


class MetadataBlock(C.Structure):
    """clx_metadata_block (metadata::MetadataBlock, metadata.rs:104-131)"""
    _fields_ = [("kind", C.c_uint32), ("length", C.c_uint32), ("streaminfo", StreamInfo), ("application_id", C.c_uint32),
                ("application_data", C.c_void_p), ("application_len", C.c_size_t), ("tags", C.c_void_p)]


BLOCK_STREAMINFO, BLOCK_PADDING, BLOCK_APPLICATION, BLOCK_VORBIS_COMMENT, BLOCK_RESERVED = 0, 1, 2, 4, 126


class BlockInfo(C.Structure):
    _fields_ = [("time", C.c_uint64), ("block_size", C.c_uint32), ("channels", C.c_uint32)]


FRAME_DESC_DTYPE = np.dtype([("byte_off", "<u8"), ("max_bytes", "<u4"), ("header_bytes", "<u2"),
                             ("block_size", "<u2"), ("n_channels", "u1"), ("channel_assignment", "u1"),
                             ("bps", "u1"), ("reserved", "u1", (5,))])
FRAME_RESULT_DTYPE = np.dtype([("status", "<i4"), ("msg", "<u4"), ("end_bit", "<u8")])
FRAME_HEADER_DTYPE = np.dtype([("time", "<u8"), ("sample_rate", "<u4"), ("frame_or_sample_lo", "<u4"),
                               ("block_size", "<u2"), ("header_bytes", "<u2"), ("n_channels", "u1"),
                               ("channel_assignment", "u1"), ("bps", "u1"), ("variable_blocking", "u1")])
assert FRAME_DESC_DTYPE.itemsize == C.sizeof(FrameDesc) == 24
assert FRAME_RESULT_DTYPE.itemsize == C.sizeof(FrameResult) == 16
assert FRAME_HEADER_DTYPE.itemsize == C.sizeof(FrameHeader) == 24

EXPORTS = [
    "clx_message", "clx_message_status", "clx_version", "clx_parse_frame_header", "clx_crc8", "clx_crc16",
    "clx_create", "clx_destroy", "clx_last_error", "clx_decode_frames", "clx_decode_frames_multi", "clx_decode_frames_stream", "clx_set_stream_chunk", "clx_host_alloc", "clx_host_free", "clx_decode_subframes", "clx_interleave",
    "clx_batch_create", "clx_batch_run", "clx_batch_submit", "clx_batch_submit_depth", "clx_batch_submit_lanes", "clx_batch_submit_merge", "clx_batch_flush", "clx_batch_interleave", "clx_batch_results", "clx_batch_slots", "clx_batch_set_profiling",
    "clx_batch_kernel_ms", "clx_batch_kernel_name", "clx_batch_destroy", "clx_read_stream_header", "clx_read_stream_header_ext",
    "clx_tags_vendor", "clx_tags_count", "clx_tags_get", "clx_tags_lookup", "clx_tags_free", "clx_reader_tags", "clx_reader_open", "clx_reader_new",
    "clx_reader_streaminfo", "clx_reader_next_block", "clx_reader_close", "clx_index_frames", "clx_index_frames_device",
    "clx_read_metadata_block", "clx_read_metadata_block_with_header", "clx_describe_packets",
]


def build(force=False, verbose=False):
    """Compile the HIP extension for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    srcs = [os.path.join(_CSRC, f) for f in ("clx_api.hip", "clx_kernels.hip", "clx_lanes.hip", "clx_lean.hip", "clx_device.h", "clx_crct.h", "clx_plan.h",
                                            os.path.join("intrin", "clx_intrin.h"), os.path.join("intrin", "clx_k2_dot2.h"), os.path.join("host", "claxon.hpp"))]
    srcs.append(os.path.join(_HERE, "..", "include", "claxon_hip.h"))
    if (not force and os.path.exists(LIB_PATH)
            and os.path.getmtime(LIB_PATH) >= max(os.path.getmtime(s) for s in srcs)):
        return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-I", os.path.join(_CSRC, "intrin"),
           os.path.join(_CSRC, "clx_api.hip"), "-o", LIB_PATH] + os.environ.get("CLX_EXTRA_FLAGS", "").split()
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=_CSRC)
    return LIB_PATH


_lib = None


def lib():
    """Load libclaxon_hip.so (raises if it has not been built -- there is no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ClaxonError(API_ERROR, 0, "libclaxon_hip.so is not built (run `python -c 'import __graft_entry__ as g; "
                                        "g.build()'`); claxon_amd has no CPU fallback")
    # When PyTorch shares the process (it is the allocator / stream owner for tests and bench.py) its wheel
    # brings its own copy of the HIP runtime under the unversioned name libamdhip64.so.  Importing torch FIRST
    # makes our NEEDED libamdhip64.so.7 resolve to that already-loaded copy (same SONAME); the other order
    # would put two HIP/HSA runtimes in one process and the second one finds no GPU.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, sz, u32p = C.c_void_p, C.c_size_t, C.POINTER(C.c_uint32)
    L.clx_message.restype = C.c_char_p
    L.clx_message.argtypes = [C.c_uint32]
    L.clx_message_status.argtypes = [C.c_uint32]
    L.clx_version.restype = C.c_uint32
    L.clx_parse_frame_header.argtypes = [vp, sz, C.c_int, C.POINTER(FrameHeader), u32p]
    L.clx_crc8.restype = C.c_uint8
    L.clx_crc8.argtypes = [vp, sz]
    L.clx_crc16.restype = C.c_uint16
    L.clx_crc16.argtypes = [vp, sz]
    L.clx_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.clx_destroy.argtypes = [vp]
    L.clx_destroy.restype = None
    L.clx_last_error.restype = C.c_char_p
    L.clx_last_error.argtypes = [vp]
    L.clx_decode_frames.argtypes = [vp, vp, sz, vp, sz, vp, vp, vp, C.c_uint32]
    L.clx_decode_subframes.argtypes = [vp, vp, sz, vp, vp, vp, sz, vp, vp, vp, C.c_uint32]
    L.clx_batch_create.argtypes = [vp, vp, sz, vp, C.c_uint32, C.POINTER(vp)]
    L.clx_decode_frames_multi.argtypes = [vp, sz, vp, sz, vp, sz, vp, vp, vp, C.c_uint32]
    L.clx_decode_frames_stream.argtypes = [vp, vp, sz, vp, sz, vp, C.c_uint32, vp, vp, C.c_uint32]
    L.clx_host_alloc.restype = vp
    L.clx_host_alloc.argtypes = [sz]
    L.clx_host_free.argtypes = [vp]
    L.clx_set_stream_chunk.argtypes = [vp, sz]
    L.clx_set_stream_chunk.restype = None
    L.clx_batch_run.argtypes = [vp, vp, sz, vp, vp]
    L.clx_batch_submit.argtypes = [vp, vp, sz, vp, vp]
    L.clx_batch_flush.argtypes = [vp, vp]
    L.clx_batch_submit_depth.argtypes = [vp]
    L.clx_batch_submit_lanes.argtypes = [vp]
    L.clx_batch_submit_merge.argtypes = [vp]
    L.clx_batch_submit_merge.restype = C.c_int
    L.clx_batch_results.argtypes = [vp, vp]
    L.clx_batch_interleave.argtypes = [vp, vp, vp, C.c_uint32, vp]
    L.clx_index_frames_device.argtypes = [vp, vp, sz, sz, vp, vp, sz, C.POINTER(sz), C.POINTER(sz), C.c_uint32]
    L.clx_interleave.argtypes = [vp, vp, vp, sz, vp, vp, vp, C.c_uint32, C.c_uint32]
    L.clx_batch_slots.restype = C.c_uint64
    L.clx_batch_slots.argtypes = [vp]
    L.clx_batch_set_profiling.argtypes = [vp, C.c_int]
    L.clx_batch_kernel_ms.argtypes = [vp, C.c_int, C.POINTER(C.c_float)]
    L.clx_batch_kernel_name.restype = C.c_char_p
    L.clx_batch_kernel_name.argtypes = [vp, C.c_int]
    L.clx_batch_destroy.argtypes = [vp]
    L.clx_batch_destroy.restype = None
    L.clx_read_stream_header.argtypes = [vp, sz, C.POINTER(StreamInfo), C.POINTER(sz), u32p]
    L.clx_read_stream_header_ext.argtypes = [vp, sz, C.c_uint32, C.POINTER(StreamInfo), C.POINTER(sz), C.POINTER(vp), u32p]
    L.clx_tags_vendor.restype = vp
    L.clx_tags_vendor.argtypes = [vp, C.POINTER(sz)]
    L.clx_tags_count.restype = sz
    L.clx_tags_count.argtypes = [vp]
    L.clx_tags_get.argtypes = [vp, sz, C.POINTER(vp), C.POINTER(sz), C.POINTER(vp), C.POINTER(sz)]
    L.clx_tags_lookup.restype = vp
    L.clx_tags_lookup.argtypes = [vp, C.c_char_p, sz, C.POINTER(sz)]
    L.clx_tags_free.argtypes = [vp]
    L.clx_tags_free.restype = None
    L.clx_reader_tags.restype = vp
    L.clx_reader_tags.argtypes = [vp]
    L.clx_reader_open.argtypes = [vp, C.c_char_p, C.POINTER(vp), u32p]
    L.clx_reader_new.argtypes = [vp, vp, sz, C.POINTER(vp), u32p]
    L.clx_reader_streaminfo.argtypes = [vp, C.POINTER(StreamInfo)]
    L.clx_reader_next_block.argtypes = [vp, vp, sz, C.POINTER(BlockInfo), u32p]
    L.clx_reader_close.argtypes = [vp]
    L.clx_reader_close.restype = None
    L.clx_index_frames.argtypes = [vp, sz, sz, vp, vp, sz, C.POINTER(sz), C.POINTER(sz)]
    L.clx_read_metadata_block.argtypes = [vp, sz, C.c_uint8, C.c_uint32, C.POINTER(MetadataBlock), C.POINTER(sz), u32p]
    L.clx_read_metadata_block_with_header.argtypes = [vp, sz, C.POINTER(MetadataBlock), C.POINTER(C.c_int), C.POINTER(sz), u32p]
    L.clx_describe_packets.argtypes = [vp, sz, vp, vp, sz, C.c_int, vp, vp, vp]
    _lib = L
    return L


def message(msg):
    return lib().clx_message(int(msg)).decode()


def _np_ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _u8(data):
    if isinstance(data, np.ndarray):
        return np.ascontiguousarray(data, dtype=np.uint8)
    return np.frombuffer(bytes(data), dtype=np.uint8)


# ----------------------------------------------------------------------------- host-side helpers

def crc8(data):
    a = _u8(data)
    return int(lib().clx_crc8(_np_ptr(a), a.size))


def crc16(data):
    a = _u8(data)
    return int(lib().clx_crc16(_np_ptr(a), a.size))


def parse_frame_header(data, check_crc=True):
    """read_frame_header_or_eof (frame.rs:131-316).  Returns (status, msg, FrameHeader)."""
    a = _u8(data)
    h = FrameHeader()
    m = C.c_uint32(0)
    st = lib().clx_parse_frame_header(_np_ptr(a), a.size, 1 if check_crc else 0, C.byref(h), C.byref(m))
    return st, int(m.value), h


def read_stream_header(data):
    a = _u8(data)
    si = StreamInfo()
    off = C.c_size_t(0)
    m = C.c_uint32(0)
    st = lib().clx_read_stream_header(_np_ptr(a), a.size, C.byref(si), C.byref(off), C.byref(m))
    return st, int(m.value), si, int(off.value)


def _tags_to_py(t):
    """(vendor bytes | None, [(name bytes, value bytes)]) from a clx_tags handle (not freed here)."""
    if not t:
        return None, []
    n = C.c_size_t(0)
    v = lib().clx_tags_vendor(t, C.byref(n))
    vendor = C.string_at(v, n.value)
    out = []
    for i in range(lib().clx_tags_count(t)):
        pn, pv, ln, lv = C.c_void_p(), C.c_void_p(), C.c_size_t(0), C.c_size_t(0)
        assert lib().clx_tags_get(t, i, C.byref(pn), C.byref(ln), C.byref(pv), C.byref(lv)) == OK
        out.append((C.string_at(pn, ln.value), C.string_at(pv, lv.value)))
    return vendor, out


def _tags_lookup(t, name):
    """metadata::GetTag (metadata.rs:197-211): every value whose name matches ASCII-case-insensitively, in order."""
    out, k = [], 0
    while True:
        n = C.c_size_t(0)
        v = lib().clx_tags_lookup(t, name.encode() if isinstance(name, str) else name, k, C.byref(n))
        if not v:
            return out
        out.append(C.string_at(v, n.value))
        k += 1


def read_stream_header_ext(data, metadata_only=False, read_vorbis_comment=True):
    """FlacReader::new_ext (lib.rs:230-307) on bytes: (status, msg, StreamInfo, audio offset, vendor | None, [(name, value)])."""
    a = _u8(data)
    si = StreamInfo()
    off = C.c_size_t(0)
    m = C.c_uint32(0)
    t = C.c_void_p()
    opts = (1 if metadata_only else 0) | (0 if read_vorbis_comment else 2)
    st = lib().clx_read_stream_header_ext(_np_ptr(a), a.size, opts, C.byref(si), C.byref(off), C.byref(t), C.byref(m))
    vendor, tags = _tags_to_py(t.value)
    if t.value:
        lib().clx_tags_free(t)
    return st, int(m.value), si, int(off.value), vendor, tags


def read_metadata_block(data, block_type=None, length=None):
    """metadata::read_metadata_block (metadata.rs:261) when block_type / length are given, else
    read_metadata_block_with_header (metadata.rs:244).  Same dict as oracle.read_metadata_block."""
    a = _u8(data)
    blk = MetadataBlock()
    used, m = C.c_size_t(0), C.c_uint32(0)
    out = {}
    if block_type is None:
        last = C.c_int(0)
        st = lib().clx_read_metadata_block_with_header(_np_ptr(a), a.size, C.byref(blk), C.byref(last), C.byref(used), C.byref(m))
        out["is_last"] = bool(last.value)
    else:
        st = lib().clx_read_metadata_block(_np_ptr(a), a.size, int(block_type), int(length), C.byref(blk), C.byref(used), C.byref(m))
    out.update(status=st, msg=int(m.value))
    if st != OK:
        return out
    out.update(kind=int(blk.kind), length=int(blk.length), consumed=int(used.value))
    if blk.kind == BLOCK_STREAMINFO:
        out["streaminfo"] = blk.streaminfo
    elif blk.kind == BLOCK_APPLICATION:
        out["app_id"] = int(blk.application_id)
        out["app_data"] = C.string_at(blk.application_data, blk.application_len) if blk.application_len else b""
    elif blk.kind == BLOCK_VORBIS_COMMENT:
        out["vendor"], out["tags"] = _tags_to_py(blk.tags)
        lib().clx_tags_free(blk.tags)
    return out


def describe_packets(arena, offs, lens, check_crc=True):
    """Container packets -> frame descriptors (clx_describe_packets): returns (descs, headers, results)."""
    a = _u8(arena)
    offs = np.ascontiguousarray(offs, dtype=np.uint64)
    lens = np.ascontiguousarray(lens, dtype=np.uint32)
    n = offs.size
    descs = np.zeros(n, dtype=FRAME_DESC_DTYPE)
    hdrs = np.zeros(n, dtype=FRAME_HEADER_DTYPE)
    res = np.zeros(n, dtype=FRAME_RESULT_DTYPE)
    st = lib().clx_describe_packets(_np_ptr(a), a.size, _np_ptr(offs), _np_ptr(lens), n, 1 if check_crc else 0,
                                    _np_ptr(descs), _np_ptr(hdrs), _np_ptr(res))
    if st == API_ERROR:
        raise ClaxonError(API_ERROR, 0, "clx_describe_packets: a packet lies outside the arena")
    return descs, hdrs, res


def index_frames(data, start=0, cap=1 << 20):
    """Host frame indexer.  Returns (descs[np FRAME_DESC_DTYPE], headers[np FRAME_HEADER_DTYPE], stop_offset)."""
    a = _u8(data)
    cap = max(1, min(cap, a.size // 8 + 2))
    descs = np.zeros(cap, dtype=FRAME_DESC_DTYPE)
    hdrs = np.zeros(cap, dtype=FRAME_HEADER_DTYPE)
    n = C.c_size_t(0)
    stop = C.c_size_t(0)
    st = lib().clx_index_frames(_np_ptr(a), a.size, start, _np_ptr(descs), _np_ptr(hdrs), cap, C.byref(n), C.byref(stop))
    if st != OK:
        raise ClaxonError(st)
    return descs[:n.value].copy(), hdrs[:n.value].copy(), int(stop.value)


def descs_from_offsets(arena, offs, max_bytes=None, check_crc=True):
    """Build frame descriptors for frames whose start offsets are known (containers, the synthetic
    generator): parses each frame header on the host.  Raises on a malformed header."""
    a = _u8(arena)
    offs = np.asarray(offs, dtype=np.uint64)
    descs = np.zeros(offs.size, dtype=FRAME_DESC_DTYPE)
    hdrs = np.zeros(offs.size, dtype=FRAME_HEADER_DTYPE)
    L = lib()
    h = FrameHeader()
    m = C.c_uint32(0)
    base = a.ctypes.data
    for i, off in enumerate(offs.tolist()):
        avail = a.size - off if max_bytes is None else min(int(max_bytes[i]), a.size - off)
        st = L.clx_parse_frame_header(C.c_void_p(base + off), avail, 1 if check_crc else 0, C.byref(h), C.byref(m))
        if st != OK:
            raise ClaxonError(st, int(m.value))
        descs[i] = (off, avail, h.header_bytes, h.block_size, h.n_channels, h.channel_assignment, h.bps, (0,) * 5)
        hdrs[i] = (h.time, h.sample_rate, h.frame_or_sample_lo, h.block_size, h.header_bytes, h.n_channels,
                   h.channel_assignment, h.bps, h.variable_blocking)
    return descs, hdrs


def descs_for_subframes(offs, block_sizes, bps):
    n = len(offs)
    d = np.zeros(n, dtype=FRAME_DESC_DTYPE)
    d["byte_off"] = offs
    d["max_bytes"] = 0xffffffff
    d["block_size"] = block_sizes
    d["n_channels"] = 1
    d["bps"] = bps
    d["reserved"][:, 0] = 1
    return d


# ----------------------------------------------------------------------------- device objects

class PinnedArray:
    """A numpy view of pinned host memory from clx_host_alloc (freed with the object): buffers handed to
    Context.decode_frames_stream in this kind of memory are copied asynchronously at link speed."""

    def __init__(self, shape, dtype=np.uint8):
        dt = np.dtype(dtype)
        n = int(np.prod(shape)) * dt.itemsize
        self._p = lib().clx_host_alloc(max(n, 1))
        if not self._p:
            raise MemoryError("clx_host_alloc(%d) failed" % n)
        buf = (C.c_uint8 * max(n, 1)).from_address(self._p)
        self.array = np.frombuffer(buf, dtype=dt, count=int(np.prod(shape))).reshape(shape)

    def close(self):
        if self._p:
            self.array = None
            lib().clx_host_free(self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def decode_frames_multi(ctxs, arena, descs, out_offs, out=None, verify_crc=False, path=0):
    """clx_decode_frames_multi: one batch, several contexts (one per GPU, or several on one), no exchange between them."""
    a = _u8(arena)
    descs = np.ascontiguousarray(descs, dtype=FRAME_DESC_DTYPE)
    out_offs = np.ascontiguousarray(out_offs, dtype=np.uint64)
    n = descs.size
    total = int((out_offs + descs["n_channels"].astype(np.uint64) * descs["block_size"].astype(np.uint64)).max()) if n else 0
    if out is None:
        out = np.zeros(total, dtype=np.int32)
    res = np.zeros(n, dtype=FRAME_RESULT_DTYPE)
    hs = (C.c_void_p * len(ctxs))(*[c._h for c in ctxs])
    st = lib().clx_decode_frames_multi(hs, len(ctxs), _np_ptr(a), a.size, _np_ptr(descs), n, _np_ptr(out), _np_ptr(out_offs),
                                       _np_ptr(res), (VERIFY_CRC16 if verify_crc else 0) | path)
    ctxs[0]._check(st)
    return out, res


def _wait_for_gpu_node(deadline):
    """Poll (in child processes) until the kernel driver exposes a gfx950 agent or the deadline passes."""
    import time
    probe = "/opt/rocm/bin/rocminfo"
    while time.time() < deadline:
        if os.path.exists("/dev/kfd"):
            if not os.path.exists(probe):
                return
            try:
                r = subprocess.run([probe], capture_output=True, timeout=30)
                if r.returncode == 0 and b"gfx950" in r.stdout:
                    return
            except Exception:
                pass
        time.sleep(1.0)


class Context:
    """clx_ctx: one per GPU / stream; not thread safe."""

    def __init__(self, device=0, wait_s=0.0):
        """`wait_s`: keep retrying for this long while the device is not (yet) visible -- a freshly
        booted GPU box can take a few seconds to expose /dev/kfd.  It never falls back to the CPU."""
        import time
        self._h = C.c_void_p(None)
        deadline = time.time() + wait_s
        if wait_s > 0:
            _wait_for_gpu_node(deadline)       # probe from a child process: never poison this one's HIP runtime
        while True:
            st = lib().clx_create(int(device), C.byref(self._h))
            if (st == OK and self._h) or time.time() >= deadline:
                break
            time.sleep(1.0)
        if st != OK or not self._h:
            self._h = None
            raise ClaxonError(API_ERROR, 0, "clx_create(%d) failed: no usable gfx950 HIP device; claxon_amd has no "
                                            "CPU fallback" % device)
        self.device = device

    def close(self):
        if self._h:
            lib().clx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def last_error(self):
        return lib().clx_last_error(self._h).decode()

    def _check(self, st):
        if st != OK:
            raise ClaxonError(st, 0, self.last_error())

    def index_frames(self, data, start=0, cap=1 << 20):
        """Device frame indexer: same contract as claxon_amd.index_frames (host), byte work on the GPU."""
        a = _u8(data)
        cap = max(1, min(cap, a.size // 8 + 2))
        descs = np.zeros(cap, dtype=FRAME_DESC_DTYPE)
        hdrs = np.zeros(cap, dtype=FRAME_HEADER_DTYPE)
        n = C.c_size_t(0)
        stop = C.c_size_t(0)
        st = lib().clx_index_frames_device(self._h, _np_ptr(a), a.size, start, _np_ptr(descs), _np_ptr(hdrs), cap,
                                           C.byref(n), C.byref(stop), 0)
        self._check(st)
        return descs[:n.value].copy(), hdrs[:n.value].copy(), int(stop.value)

    def decode_frames(self, arena, descs, out_offs, out=None, verify_crc=False, path=0):
        """One-shot host->device->host decode.  Returns (out int32, results np FRAME_RESULT_DTYPE)."""
        a = _u8(arena)
        descs = np.ascontiguousarray(descs, dtype=FRAME_DESC_DTYPE)
        out_offs = np.ascontiguousarray(out_offs, dtype=np.uint64)
        n = descs.size
        total = int((out_offs + descs["n_channels"].astype(np.uint64) * descs["block_size"].astype(np.uint64)).max()) if n else 0
        if out is None:
            out = np.zeros(total, dtype=np.int32)
        assert out.dtype == np.int32 and out.size >= total
        res = np.zeros(n, dtype=FRAME_RESULT_DTYPE)
        st = lib().clx_decode_frames(self._h, _np_ptr(a), a.size, _np_ptr(descs), n, _np_ptr(out), _np_ptr(out_offs),
                                     _np_ptr(res), (VERIFY_CRC16 if verify_crc else 0) | path)
        self._check(st)
        return out, res

    def set_stream_chunk(self, frames_per_chunk):
        """Frames per chunk of decode_frames_stream (0: the library's rule: a third of the batch, 256 .. 8192)."""
        lib().clx_set_stream_chunk(self._h, int(frames_per_chunk))

    def decode_frames_stream(self, arena, descs, out_offs, out=None, sample_bytes=0, verify_crc=False, path=0, copy_back=True):
        """clx_decode_frames_stream: host-to-host decode, chunks pipelined (upload | decode | download).  sample_bytes 0: planar
        int32 (returned as int32 array); 1..4: channel-interleaved little-endian PCM (returned as uint8 array);
        copy_back=False: only the results come back.  Returns (out or None, results)."""
        a = _u8(arena)
        descs = np.ascontiguousarray(descs, dtype=FRAME_DESC_DTYPE)
        out_offs = np.ascontiguousarray(out_offs, dtype=np.uint64)
        n = descs.size
        total = int((out_offs + descs["n_channels"].astype(np.uint64) * descs["block_size"].astype(np.uint64)).max()) if n else 0
        if copy_back and out is None:
            out = np.zeros(total, dtype=np.int32) if sample_bytes == 0 else np.zeros(total * sample_bytes, dtype=np.uint8)
        if copy_back:
            assert out.nbytes >= total * (sample_bytes or 4)
        res = np.zeros(n, dtype=FRAME_RESULT_DTYPE)
        st = lib().clx_decode_frames_stream(self._h, _np_ptr(a), a.size, _np_ptr(descs), n, _np_ptr(out) if copy_back else None, sample_bytes,
                                            _np_ptr(out_offs), _np_ptr(res), (VERIFY_CRC16 if verify_crc else 0) | path)
        self._check(st)
        return (out if copy_back else None), res

    def interleave(self, planar, descs, out_offs, sample_bytes, results=None, pcm=None):
        """One-shot interleave / narrow stage on host arrays: planar i32 -> channel-interleaved little-endian PCM of
        `sample_bytes` bytes per sample (uint8 array, frame i at byte out_offs[i] * sample_bytes).  Frames whose
        `results` status is not OK keep whatever `pcm` held."""
        planar = np.ascontiguousarray(planar, dtype=np.int32)
        descs = np.ascontiguousarray(descs, dtype=FRAME_DESC_DTYPE)
        out_offs = np.ascontiguousarray(out_offs, dtype=np.uint64)
        n = descs.size
        total = int((out_offs + descs["n_channels"].astype(np.uint64) * descs["block_size"].astype(np.uint64)).max()) if n else 0
        assert planar.size >= total
        if pcm is None:
            pcm = np.zeros(total * sample_bytes, dtype=np.uint8)
        assert pcm.dtype == np.uint8 and pcm.size >= total * sample_bytes
        if results is not None:
            results = np.ascontiguousarray(results, dtype=FRAME_RESULT_DTYPE)
        st = lib().clx_interleave(self._h, _np_ptr(planar), _np_ptr(descs), n, _np_ptr(out_offs),
                                  _np_ptr(results) if results is not None else None, _np_ptr(pcm), sample_bytes, 0)
        self._check(st)
        return pcm

    def decode_subframes(self, arena, offs, block_sizes, bps, out_offs, out=None):
        a = _u8(arena)
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        bsz = np.ascontiguousarray(block_sizes, dtype=np.uint16)
        bp = np.ascontiguousarray(bps, dtype=np.uint8)
        out_offs = np.ascontiguousarray(out_offs, dtype=np.uint64)
        n = offs.size
        total = int((out_offs + bsz.astype(np.uint64)).max()) if n else 0
        if out is None:
            out = np.zeros(total, dtype=np.int32)
        res = np.zeros(n, dtype=FRAME_RESULT_DTYPE)
        st = lib().clx_decode_subframes(self._h, _np_ptr(a), a.size, _np_ptr(offs), _np_ptr(bsz), _np_ptr(bp), n,
                                        _np_ptr(out), _np_ptr(out_offs), _np_ptr(res), 0)
        self._check(st)
        return out, res

    def plan(self, descs, out_offs, verify_crc=False, path=0):
        """path: 0 = automatic, PATH_WAVES or PATH_LANES to force a kernel path."""
        return Batch(self, descs, out_offs, verify_crc, path)


class Batch:
    """clx_batch: a planned batch, run on device-resident buffers (what bench.py times)."""

    def __init__(self, ctx, descs, out_offs, verify_crc=False, path=0):
        self.ctx = ctx
        descs = np.ascontiguousarray(descs, dtype=FRAME_DESC_DTYPE)
        out_offs = np.ascontiguousarray(out_offs, dtype=np.uint64)
        self.n = descs.size
        self._h = C.c_void_p(None)
        st = lib().clx_batch_create(ctx._h, _np_ptr(descs), self.n, _np_ptr(out_offs),
                                    (VERIFY_CRC16 if verify_crc else 0) | path, C.byref(self._h))
        ctx._check(st)

    @property
    def slots(self):
        return int(lib().clx_batch_slots(self._h))

    def run(self, d_arena_ptr, arena_len, d_out_ptr, stream=0):
        """d_arena_ptr / d_out_ptr: integer device addresses (e.g. torch tensor .data_ptr())."""
        st = lib().clx_batch_run(self._h, C.c_void_p(d_arena_ptr), arena_len, C.c_void_p(d_out_ptr),
                                 C.c_void_p(stream) if stream else None)
        self.ctx._check(st)

    def submit(self, d_arena_ptr, arena_len, d_out_ptr, stream=0):
        """Pipelined run (clx_batch_submit): up to self.submit_depth submissions in flight on internal streams.  Rotate over
        that many output buffers; flush() (or results()) before reading them."""
        st = lib().clx_batch_submit(self._h, C.c_void_p(d_arena_ptr), arena_len, C.c_void_p(d_out_ptr),
                                    C.c_void_p(stream) if stream else None)
        self.ctx._check(st)

    @property
    def submit_depth(self):
        """Submissions this batch keeps in flight = output buffers to rotate over (clx_batch_submit_depth)."""
        return int(lib().clx_batch_submit_depth(self._h))

    @property
    def submit_lanes(self):
        """True when this batch's pipelined submissions run the fused lane kernels (clx_batch_submit_lanes)."""
        return bool(lib().clx_batch_submit_lanes(self._h))

    @property
    def submit_merge(self):
        """How many consecutive submissions go out as one launch (clx_batch_submit_merge)."""
        return int(lib().clx_batch_submit_merge(self._h))

    def flush(self, stream=0):
        self.ctx._check(lib().clx_batch_flush(self._h, C.c_void_p(stream) if stream else None))

    def interleave(self, d_planar_ptr, d_pcm_ptr, sample_bytes, stream=0):
        """Device-resident interleave / narrow stage after run(): integer device addresses, async on `stream`."""
        st = lib().clx_batch_interleave(self._h, C.c_void_p(d_planar_ptr), C.c_void_p(d_pcm_ptr), sample_bytes,
                                        C.c_void_p(stream) if stream else None)
        self.ctx._check(st)

    def results(self):
        res = np.zeros(self.n, dtype=FRAME_RESULT_DTYPE)
        self.ctx._check(lib().clx_batch_results(self._h, _np_ptr(res)))
        return res

    def set_profiling(self, on=True):
        """True / 1: plain runs with per-kernel events; 2: pipelined submissions, events around the kernels of each merged launch."""
        lib().clx_batch_set_profiling(self._h, int(on) if not isinstance(on, bool) else (1 if on else 0))

    def kernel_ms(self, kernel):
        ms = C.c_float(0)
        self.ctx._check(lib().clx_batch_kernel_ms(self._h, kernel, C.byref(ms)))
        return float(ms.value)

    def kernel_times(self):
        """{kernel name: ms} of the last profiled run, in launch order."""
        out, k = {}, 0
        while True:
            name = lib().clx_batch_kernel_name(self._h, k)
            if not name:
                return out
            out[name.decode()] = self.kernel_ms(k)
            k += 1

    def close(self):
        if self._h:
            lib().clx_batch_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Block:
    """frame.rs:402-529"""

    def __init__(self, time, block_size, channels, buffer):
        self._time, self._bs, self._ch, self._buf = time, block_size, channels, buffer

    def time(self):
        return self._time

    def len(self):
        return self._bs * self._ch

    def duration(self):
        return self._bs

    def channels(self):
        return self._ch

    def channel(self, ch):
        if ch >= self._ch:
            raise IndexError(ch)
        return self._buf[ch * self._bs:(ch + 1) * self._bs]

    def sample(self, ch, i):
        return int(self._buf[ch * self._bs + i])

    def into_buffer(self):
        return self._buf

    def stereo_samples(self):
        if self._ch != 2:
            raise ValueError("stereo_samples() must only be called for blocks with two channels.")
        return zip(self._buf[:self._bs].tolist(), self._buf[self._bs:2 * self._bs].tolist())


class FlacReader:
    """lib.rs:93-97, 207-471 over an in-memory stream; frames are decoded on the GPU in batches."""

    def __init__(self, ctx, data=None, path=None):
        self.ctx = ctx
        self._h = C.c_void_p(None)
        m = C.c_uint32(0)
        if path is not None:
            st = lib().clx_reader_open(ctx._h, os.fsencode(path), C.byref(self._h), C.byref(m))
        else:
            a = _u8(data)
            st = lib().clx_reader_new(ctx._h, _np_ptr(a), a.size, C.byref(self._h), C.byref(m))
        if st != OK:
            self._h = None
            raise ClaxonError(st, int(m.value))
        self._si = StreamInfo()
        self._buf = None
        lib().clx_reader_streaminfo(self._h, C.byref(self._si))

    @classmethod
    def open(cls, ctx, path):
        return cls(ctx, path=path)

    def streaminfo(self):
        return self._si

    def vendor(self):
        """lib.rs:321: the encoder's vendor string (str), or None when the stream has no Vorbis comment block."""
        v, _ = _tags_to_py(lib().clx_reader_tags(self._h))
        return None if v is None else v.decode("utf-8")

    def tags(self):
        """lib.rs:335: (name, value) pairs in stream order."""
        _, t = _tags_to_py(lib().clx_reader_tags(self._h))
        return [(n.decode("utf-8"), v.decode("utf-8")) for n, v in t]

    def get_tag(self, name):
        """lib.rs:356: every value of the tag `name` (ASCII-case-insensitive), in stream order."""
        t = lib().clx_reader_tags(self._h)
        return [v.decode("utf-8") for v in _tags_lookup(t, name)] if t else []

    def read_next_or_eof(self):
        """Returns a Block, or None at the end of the stream; raises ClaxonError like the reference returns Err."""
        # one staging buffer per reader, sized from STREAMINFO (max block size x channels); a block that is larger than announced
        # stays pending in the library and is fetched again into a buffer of the size it reports
        if self._buf is None:
            si = self._si
            self._buf = np.empty(max(1, int(si.max_block_size or 65535) * int(si.channels or 8)), dtype=np.int32)
        buf = self._buf
        info = BlockInfo()
        m = C.c_uint32(0)
        st = lib().clx_reader_next_block(self._h, _np_ptr(buf), buf.size, C.byref(info), C.byref(m))
        if st == API_ERROR and info.block_size * info.channels > buf.size:
            self._buf = buf = np.empty(int(info.block_size) * int(info.channels), dtype=np.int32)
            st = lib().clx_reader_next_block(self._h, _np_ptr(buf), buf.size, C.byref(info), C.byref(m))
        if st == END_OF_STREAM:
            return None
        if st != OK:
            raise ClaxonError(st, int(m.value), self.ctx.last_error() if st == API_ERROR else None)
        n = info.block_size * info.channels
        return Block(info.time, info.block_size, info.channels, buf[:n].copy())

    def blocks(self):
        while True:
            b = self.read_next_or_eof()
            if b is None:
                return
            yield b

    def samples(self):
        """Interleaved samples (lib.rs:473-520)."""
        for b in self.blocks():
            inter = b.into_buffer().reshape(b.channels(), b.duration()).T.reshape(-1)
            for s in inter.tolist():
                yield s

    def close(self):
        if self._h:
            lib().clx_reader_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
