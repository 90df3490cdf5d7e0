"""Shared helpers: run `FrameReader::read_next_or_eof` on one in-memory frame through the product's
host header parser + a device backend (the GPU, or the wave simulator), for comparison with the oracle."""
import numpy as np

import claxon_amd as cx
from claxon_msgs import MSG


def pad16(data):
    a = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else data
    out = np.zeros((a.size + 15) // 16 * 16 + 64, dtype=np.uint8)
    out[:a.size] = a
    return out, a.size


class SimBackend:
    name = "wavesim"

    def __init__(self, path=0):
        self.path = path          # 0 auto (= waves in the simulator), cx.PATH_WAVES, cx.PATH_LANES

    def decode(self, arena, arena_len, descs, out_offs, verify_crc, fill=0):
        import simlib
        out, res, _ = simlib.decode(arena, arena_len, descs, out_offs, verify_crc=verify_crc, fill=fill, path=self.path)
        return out, res


class GpuBackend:
    """Planned-batch path on device-resident buffers (torch is only the allocator here)."""
    name = "gpu"

    def __init__(self, ctx=None, path=0):
        import torch
        self.torch = torch
        self.ctx = ctx or cx.Context(0)
        self.path = path

    def decode(self, arena, arena_len, descs, out_offs, verify_crc, fill=0):
        torch = self.torch
        n = len(descs)
        out_offs = np.asarray(out_offs, dtype=np.uint64)
        total = int((out_offs + descs["n_channels"].astype(np.uint64) *
                     descs["block_size"].astype(np.uint64)).max()) if n else 0
        d_arena = torch.from_numpy(np.ascontiguousarray(arena)).to("cuda:0")
        if self.path & cx.OUT_PCM24:     # (packed 24-bit PCM: bytes, frame i's block from byte 3 * out_offs[i])
            d_out = torch.full((3 * max(total, 1) + 16,), int(fill) & 0xff, dtype=torch.uint8, device="cuda:0")
            total *= 3
        elif self.path & cx.OUT_PCM16:   # (narrow output: the buffer holds interleaved 16-bit PCM at the same sample offsets)
            d_out = torch.full((max(total, 1) + 8,), int(fill) & 0x7fff, dtype=torch.int16, device="cuda:0")
        else:
            d_out = torch.full((max(total, 1),), int(np.int32(fill)), dtype=torch.int32, device="cuda:0")
        batch = self.ctx.plan(descs, out_offs, verify_crc=verify_crc, path=self.path)
        torch.cuda.synchronize()
        batch.run(d_arena.data_ptr(), int(arena_len), d_out.data_ptr())
        res = batch.results()
        out = d_out.cpu().numpy()[:total]
        batch.close()
        return out, res


def product_frame_decode(backend, data, check_crc=True, fill=0):
    """Returns (status, msg, end_bit, samples|None, header|None) for the frame at data[0:]."""
    arena, n = pad16(data)
    st, msg, h = cx.parse_frame_header(arena[:n], check_crc)
    if st != cx.OK:
        return st, msg, 0, None, None
    if h.bps == 0:
        return cx.UNSUPPORTED, MSG["CLX_MSG_NO_BPS_IN_HEADER"], 0, None, h
    descs = np.zeros(1, dtype=cx.FRAME_DESC_DTYPE)
    descs[0] = (0, n, h.header_bytes, h.block_size, h.n_channels, h.channel_assignment, h.bps, (0,) * 5)
    out, res = backend.decode(arena, n, descs, np.zeros(1, dtype=np.uint64), check_crc, fill=fill)
    r = res[0]
    if r["status"] != cx.OK:
        return int(r["status"]), int(r["msg"]), int(r["end_bit"]), None, h
    return cx.OK, 0, int(r["end_bit"]), out[:h.block_size * h.n_channels], h


def assert_same_as_oracle(oracle, backend, data, check_crc=True, ctx=""):
    info, ref = oracle.frame_decode(data, check_crc)
    st, msg, end_bit, got, h = product_frame_decode(backend, data, check_crc, fill=0x13131313)
    assert (st, msg) == (info.status, info.msg), "%s: product (%d,%d) vs oracle (%d,%d)" % (ctx, st, msg, info.status, info.msg)
    if st == cx.OK:
        assert end_bit == info.end_bit, ctx
        assert np.array_equal(got, ref), ctx
        assert (h.block_size, h.n_channels, h.time) == (info.block_size, info.channels, info.time), ctx
    return st, msg
