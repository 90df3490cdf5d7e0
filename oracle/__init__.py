"""ctypes binding of the CPU oracle (oracle/claxon_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, by __graft_entry__.smoke() and by
bench.py's cpu_baseline leg -- as the checker / the reported CPU baseline,
never by the product package `claxon_amd`.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libclaxon_oracle.so")

STATUS_OK, STATUS_IO, STATUS_FORMAT, STATUS_UNSUPPORTED, STATUS_EOS, STATUS_API = range(6)


def build(force=False):
    """Compile the oracle with gcc (a few hundred ms)."""
    src = os.path.join(_HERE, "claxon_oracle.c")
    hdr = os.path.join(_HERE, "..", "include", "claxon_hip.h")
    if (not force and os.path.exists(_SO)
            and os.path.getmtime(_SO) >= max(os.path.getmtime(src), os.path.getmtime(hdr))):
        return _SO
    subprocess.check_call(["make", "-C", _HERE, "-B", "libclaxon_oracle.so"],
                          stdout=subprocess.DEVNULL)
    return _SO


class StreamInfo(C.Structure):
    _fields_ = [("min_block_size", C.c_uint16), ("max_block_size", C.c_uint16),
                ("min_frame_size", C.c_uint32), ("max_frame_size", C.c_uint32),
                ("sample_rate", C.c_uint32), ("channels", C.c_uint32),
                ("bits_per_sample", C.c_uint32), ("samples", C.c_uint64),
                ("md5sum", C.c_uint8 * 16)]


class FrameInfo(C.Structure):
    _fields_ = [("status", C.c_int32), ("msg", C.c_uint32), ("time", C.c_uint64),
                ("bytes_consumed", C.c_uint64), ("end_bit", C.c_uint64),
                ("block_size", C.c_uint32), ("channels", C.c_uint32), ("bps", C.c_uint32),
                ("channel_assignment", C.c_uint32), ("sample_rate", C.c_uint32),
                ("header_bytes", C.c_uint32)]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_SO):
        build()
    L = C.CDLL(_SO)
    u8p, i32p, u32p, u64p, i16p, u16p = (C.POINTER(t) for t in
                                         (C.c_uint8, C.c_int32, C.c_uint32, C.c_uint64, C.c_int16, C.c_uint16))
    L.clxo_crc8.restype = C.c_uint8
    L.clxo_crc8.argtypes = [C.c_void_p, C.c_size_t]
    L.clxo_crc16.restype = C.c_uint16
    L.clxo_crc16.argtypes = [C.c_void_p, C.c_size_t]
    for n in ("clxo_extend_sign_u16", "clxo_extend_sign_u32"):
        getattr(L, n).restype = C.c_int32
        getattr(L, n).argtypes = [C.c_uint32, C.c_uint32]
    L.clxo_rice_to_signed.restype = C.c_int32
    L.clxo_rice_to_signed.argtypes = [C.c_uint32]
    L.clxo_predict_fixed.argtypes = [C.c_uint32, C.c_void_p, C.c_size_t]
    L.clxo_predict_lpc_low_order.argtypes = [C.c_void_p, C.c_size_t, C.c_int32, C.c_void_p, C.c_size_t]
    L.clxo_predict_lpc_high_order.argtypes = [C.c_void_p, C.c_size_t, C.c_int32, C.c_void_p, C.c_size_t]
    for n in ("clxo_decode_left_side", "clxo_decode_right_side", "clxo_decode_mid_side"):
        getattr(L, n).argtypes = [C.c_void_p, C.c_size_t]
    L.clxo_read_var_length_int.restype = C.c_int
    L.clxo_read_var_length_int.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), u64p, u32p]
    L.clxo_bitstream_script.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t,
                                        C.c_void_p, C.c_void_p]
    L.clxo_frame_decode.restype = C.c_int
    L.clxo_frame_decode.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int,
                                    C.POINTER(FrameInfo)]
    L.clxo_subframe_decode.restype = C.c_int
    L.clxo_subframe_decode.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p, C.c_size_t, u32p, u64p]
    L.clxo_stream_open.restype = C.c_int
    L.clxo_stream_open.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(StreamInfo), u64p, u32p]
    L.clxo_decode_batch.restype = C.c_uint64
    L.clxo_decode_batch.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_int, C.c_int]
    L.clxo_bench_batch.restype = C.c_uint64
    L.clxo_bench_batch.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int,
                                   C.c_void_p, C.POINTER(C.c_double)]
    L.clxo_decode_subframes.restype = C.c_uint64
    L.clxo_decode_subframes.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    _lib = L
    return L


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _bytes_arr(data):
    a = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else data
    return np.ascontiguousarray(a, dtype=np.uint8)


def crc8(data):
    a = _bytes_arr(data)
    return int(lib().clxo_crc8(_ptr(a), a.size))


def crc16(data):
    a = _bytes_arr(data)
    return int(lib().clxo_crc16(_ptr(a), a.size))


def bitstream_script(data, script):
    """script: list of (op, bits) with op in
    {'bit','unary','leq_u8','gt_u8_leq_u16','leq_u16','leq_u32'}.
    Returns list of (value, ok)."""
    kinds = {"bit": 0, "unary": 1, "leq_u8": 2, "gt_u8_leq_u16": 3, "leq_u16": 4, "leq_u32": 5}
    a = _bytes_arr(data)
    ops = np.array([kinds[o] for o, _ in script], dtype=np.int32)
    args = np.array([b for _, b in script], dtype=np.uint32)
    vals = np.zeros(len(script), dtype=np.uint32)
    errs = np.zeros(len(script), dtype=np.int32)
    lib().clxo_bitstream_script(_ptr(a), a.size, _ptr(ops), _ptr(args), len(script), _ptr(vals), _ptr(errs))
    return [(int(v), int(e) == 0) for v, e in zip(vals, errs)]


def predict_fixed(order, buf):
    b = np.array(buf, dtype=np.int32)
    lib().clxo_predict_fixed(order, _ptr(b), b.size)
    return b


def predict_lpc(coefs, shift, buf, high=None):
    """coefs in the reference's *stored* (application) order: coefs[j] multiplies buf[i-order+j]."""
    c = np.array(coefs, dtype=np.int16)
    b = np.array(buf, dtype=np.int32)
    if high is None:
        high = c.size > 12
    f = lib().clxo_predict_lpc_high_order if high else lib().clxo_predict_lpc_low_order
    f(_ptr(c), c.size, int(shift), _ptr(b), b.size)
    return b


def decorrelate(kind, buf):
    b = np.array(buf, dtype=np.int32)
    getattr(lib(), "clxo_decode_%s" % kind)(_ptr(b), b.size)
    return b


def read_var_length_ints(data, count):
    a = _bytes_arr(data)
    pos = C.c_size_t(0)
    out = []
    for _ in range(count):
        v = C.c_uint64(0)
        m = C.c_uint32(0)
        st = lib().clxo_read_var_length_int(_ptr(a), a.size, C.byref(pos), C.byref(v), C.byref(m))
        out.append((st, int(v.value), int(m.value)))
    return out


def frame_decode(data, check_crc=True, out=None):
    """FrameReader::read_next_or_eof on Cursor(data).  Returns (FrameInfo, planar int32 array or None)."""
    a = _bytes_arr(data)
    info = FrameInfo()
    buf = np.empty(8 * 65535, dtype=np.int32) if out is None else out
    lib().clxo_frame_decode(_ptr(a), a.size, _ptr(buf), buf.size, 1 if check_crc else 0, C.byref(info))
    if info.status != STATUS_OK:
        return info, None
    n = info.block_size * info.channels
    return info, (buf[:n].copy() if out is None else buf[:n])


def subframe_decode(data, bps, n):
    a = _bytes_arr(data)
    out = np.zeros(n, dtype=np.int32)
    msg = C.c_uint32(0)
    eb = C.c_uint64(0)
    st = lib().clxo_subframe_decode(_ptr(a), a.size, bps, _ptr(out), n, C.byref(msg), C.byref(eb))
    return st, int(msg.value), int(eb.value), out


def stream_open(data):
    a = _bytes_arr(data)
    si = StreamInfo()
    off = C.c_uint64(0)
    msg = C.c_uint32(0)
    st = lib().clxo_stream_open(_ptr(a), a.size, C.byref(si), C.byref(off), C.byref(msg))
    return st, int(msg.value), si, int(off.value)


def stream_open_ext(data, metadata_only=False, read_vorbis_comment=True):
    """FlacReader::new_ext (lib.rs:230-307).  Returns (status, msg, streaminfo, audio offset, vendor|None, [(name, value)])
    with vendor / names / values as bytes."""
    import struct
    a = _bytes_arr(data)
    si = StreamInfo()
    off = C.c_uint64(0)
    msg = C.c_uint32(0)
    cap = a.size + 64
    buf = np.zeros(cap, dtype=np.uint8)
    tl = C.c_size_t(0)
    L = lib()
    L.clxo_stream_open_ext.restype = C.c_int
    L.clxo_stream_open_ext.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.POINTER(StreamInfo), C.POINTER(C.c_uint64),
                                       C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_uint32)]
    opts = (1 if metadata_only else 0) | (0 if read_vorbis_comment else 2)
    st = L.clxo_stream_open_ext(_ptr(a), a.size, opts, C.byref(si), C.byref(off), buf.ctypes.data, cap, C.byref(tl), C.byref(msg))
    vendor, tags = None, []
    if st == STATUS_OK and tl.value:
        raw = buf[:tl.value].tobytes()
        (vl,) = struct.unpack_from("=I", raw, 0)
        vendor = raw[4:4 + vl]
        (n,) = struct.unpack_from("=I", raw, 4 + vl)
        p = 8 + vl
        for _ in range(n):
            ln, sep = struct.unpack_from("=II", raw, p)
            c = raw[p + 8:p + 8 + ln]
            tags.append((c[:sep], c[sep + 1:]))
            p += 8 + ln
    return st, int(msg.value), si, int(off.value), vendor, tags


def _parse_tags_record(raw):
    import struct
    (vl,) = struct.unpack_from("=I", raw, 0)
    vendor = raw[4:4 + vl]
    (n,) = struct.unpack_from("=I", raw, 4 + vl)
    p, tags = 8 + vl, []
    for _ in range(n):
        ln, sep = struct.unpack_from("=II", raw, p)
        c = raw[p + 8:p + 8 + ln]
        tags.append((c[:sep], c[sep + 1:]))
        p += 8 + ln
    return vendor, tags


def read_metadata_block(data, block_type=None, length=None):
    """metadata::read_metadata_block (metadata.rs:261) when block_type / length are given, else
    read_metadata_block_with_header (metadata.rs:244).  Returns a dict: status, msg, and on success kind, length, consumed,
    is_last (with header), streaminfo | (app_id, app_data) | (vendor, tags)."""
    a = _bytes_arr(data)
    L = lib()
    si = StreamInfo()
    app = (C.c_uint64 * 3)()
    cap = a.size + 64
    buf = np.zeros(cap, dtype=np.uint8)
    tl, used, msg, kind = C.c_size_t(0), C.c_size_t(0), C.c_uint32(0), C.c_uint32(0)
    out = {}
    if block_type is None:
        ln, last = C.c_uint32(0), C.c_int(0)
        L.clxo_read_metadata_block_with_header.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_int),
                                                           C.POINTER(StreamInfo), C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t),
                                                           C.POINTER(C.c_size_t), C.POINTER(C.c_uint32)]
        st = L.clxo_read_metadata_block_with_header(_ptr(a), a.size, C.byref(kind), C.byref(ln), C.byref(last), C.byref(si), app,
                                                    buf.ctypes.data, cap, C.byref(tl), C.byref(used), C.byref(msg))
        out["is_last"] = bool(last.value)
        length = int(ln.value)
    else:
        L.clxo_read_metadata_block.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(StreamInfo),
                                               C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_uint32)]
        st = L.clxo_read_metadata_block(_ptr(a), a.size, int(block_type), int(length), C.byref(kind), C.byref(si), app,
                                        buf.ctypes.data, cap, C.byref(tl), C.byref(used), C.byref(msg))
    out.update(status=st, msg=int(msg.value))
    if st != STATUS_OK:
        return out
    out.update(kind=int(kind.value), length=int(length), consumed=int(used.value))
    if kind.value == 0:
        out["streaminfo"] = si
    elif kind.value == 2:
        out["app_id"] = int(app[0])
        out["app_data"] = a[int(app[1]):int(app[1]) + int(app[2])].tobytes()
    elif kind.value == 4:
        out["vendor"], out["tags"] = _parse_tags_record(buf[:tl.value].tobytes())
    return out


def decode_stream(data, check_crc=True):
    """FlacReader::new + blocks() loop.  Returns (streaminfo, [(FrameInfo, samples)], final_status, final_msg)."""
    a = _bytes_arr(data)
    st, msg, si, off = stream_open(a)
    if st != STATUS_OK:
        return None, [], st, msg
    blocks = []
    pos = off
    while True:
        info, samples = frame_decode(a[pos:], check_crc)
        if info.status == STATUS_EOS:
            return si, blocks, STATUS_OK, 0
        if info.status != STATUS_OK:
            return si, blocks, info.status, info.msg
        blocks.append((info, samples))
        pos += info.bytes_consumed


def decode_batch(arena, offs, max_bytes=None, out=None, out_offs=None, check_crc=True, nthreads=1,
                 want_results=True):
    """Decode frames at arena[offs[i]:]; returns dict(samples, statuses, msgs, end_bits)."""
    arena = _bytes_arr(arena)
    offs = np.ascontiguousarray(offs, dtype=np.uint64)
    n = offs.size
    mb = None if max_bytes is None else np.ascontiguousarray(max_bytes, dtype=np.uint32)
    st = np.zeros(n, dtype=np.int32) if want_results else None
    ms = np.zeros(n, dtype=np.uint32) if want_results else None
    eb = np.zeros(n, dtype=np.uint64) if want_results else None
    oo = None if out_offs is None else np.ascontiguousarray(out_offs, dtype=np.uint64)
    total = lib().clxo_decode_batch(_ptr(arena), arena.size, _ptr(offs), _ptr(mb), n, _ptr(out), _ptr(oo),
                                    _ptr(st), _ptr(ms), _ptr(eb), 1 if check_crc else 0, int(nthreads))
    return dict(samples=int(total), statuses=st, msgs=ms, end_bits=eb)


def bench_batch(arena, offs, max_bytes=None, check_crc=True, nthreads=1, passes=1, cpus=None):
    """Timed decode for bench.py's cpu_baseline: pooled, pre-warmed threads, thread t pinned to cpus[t] when `cpus` is given;
    returns (samples, seconds) of the timed region (`passes` passes over the frames by `nthreads` threads)."""
    arena = _bytes_arr(arena)
    offs = np.ascontiguousarray(offs, dtype=np.uint64)
    mb = None if max_bytes is None else np.ascontiguousarray(max_bytes, dtype=np.uint32)
    sec = C.c_double(0.0)
    nthreads = max(1, int(nthreads))
    cp = None
    if cpus is not None:
        cp = np.ascontiguousarray(np.resize(np.asarray(cpus, dtype=np.int32), nthreads))
    total = lib().clxo_bench_batch(_ptr(arena), arena.size, _ptr(offs), _ptr(mb), offs.size, 1 if check_crc else 0,
                                   nthreads, int(passes), _ptr(cp), C.byref(sec))
    return int(total), float(sec.value)


def decode_subframes(arena, offs, block_sizes, bps, out=None, out_offs=None):
    arena = _bytes_arr(arena)
    offs = np.ascontiguousarray(offs, dtype=np.uint64)
    bsz = np.ascontiguousarray(block_sizes, dtype=np.uint16)
    bp = np.ascontiguousarray(bps, dtype=np.uint8)
    n = offs.size
    st = np.zeros(n, dtype=np.int32)
    ms = np.zeros(n, dtype=np.uint32)
    eb = np.zeros(n, dtype=np.uint64)
    oo = None if out_offs is None else np.ascontiguousarray(out_offs, dtype=np.uint64)
    total = lib().clxo_decode_subframes(_ptr(arena), arena.size, _ptr(offs), _ptr(bsz), _ptr(bp), n,
                                        _ptr(out), _ptr(oo), _ptr(st), _ptr(ms), _ptr(eb))
    return dict(samples=int(total), statuses=st, msgs=ms, end_bits=eb)
