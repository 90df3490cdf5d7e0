"""Interleave / narrow output stage (FlacSamples order lib.rs:473-520, i16 WAV examples/decode.rs:48-62).
Reference = a numpy restatement below; end-to-end pin = the STREAMINFO MD5 of the fixtures (interleaved
little-endian samples at ceil(bps/8) bytes, metadata.rs:52-53), computed from the DEVICE-interleaved bytes."""
import hashlib
import os

import numpy as np
import pytest

import claxon_amd as cx
import synth
from conftest import FIXTURES
from parity_cases import workload_descs


def ref_interleave(planar, descs, out_offs, sb, results=None, fill=0xee):
    total = int((np.asarray(out_offs, dtype=np.uint64) + descs["n_channels"].astype(np.uint64) * descs["block_size"].astype(np.uint64)).max())
    pcm = np.full(total * sb, fill, dtype=np.uint8)
    for i in range(descs.size):
        if results is not None and results["status"][i] != cx.OK:
            continue
        c, bs, off = int(descs["n_channels"][i]), int(descs["block_size"][i]), int(out_offs[i])
        x = planar[off:off + c * bs].reshape(c, bs).T.astype("<i4")                 # [sample][channel]
        pcm[off * sb:(off + c * bs) * sb] = np.ascontiguousarray(x).view(np.uint8).reshape(bs, c, 4)[:, :, :sb].reshape(-1)
    return pcm


def cases():
    """(name, planar, descs, out_offs, results) covering stereo / mono / 8 channels, odd offsets, failed frames."""
    out = []
    w = synth.small_mixed(24, bs=192, seed_off=77)
    d = workload_descs(w)
    out.append(("small_mixed", w.pcm, d, w.out_offs, None))
    res = np.zeros(d.size, dtype=cx.FRAME_RESULT_DTYPE)
    res["status"][::5] = cx.FORMAT_ERROR
    out.append(("skip_failed", w.pcm, d, w.out_offs, res))
    e = __import__("parity_cases").edge_workload()
    out.append(("edges", e.pcm, workload_descs(e), e.out_offs, None))
    # odd sample offsets (mono frames of odd length in front of stereo frames): the unaligned store paths
    rng = np.random.default_rng(5)
    descs = np.zeros(6, dtype=cx.FRAME_DESC_DTYPE)
    descs["n_channels"] = [1, 2, 2, 8, 1, 2]; descs["block_size"] = [17, 16, 33, 5, 1, 4096]; descs["bps"] = 16
    offs = np.concatenate([[0], np.cumsum(descs["n_channels"].astype(np.uint64) * descs["block_size"])[:-1]]).astype(np.uint64)
    planar = rng.integers(-2 ** 31, 2 ** 31 - 1, size=int(offs[-1]) + 2 * 4096, dtype=np.int64).astype(np.int32)
    out.append(("odd_offsets", planar, descs, offs, None))
    return out


@pytest.mark.parametrize("sb", [1, 2, 3, 4])
def test_sim_interleave(sb):
    import simlib
    for name, planar, descs, offs, res in cases():
        want = ref_interleave(planar, descs, offs, sb, res, fill=0)
        got = simlib.interleave(planar, descs, offs, sb, results=res)
        assert np.array_equal(got, want), (name, sb)


@pytest.mark.gpu
@pytest.mark.parametrize("sb", [1, 2, 3, 4])
def test_gpu_interleave_host_api(sb):
    ctx = cx.Context(0, wait_s=120)
    for name, planar, descs, offs, res in cases():
        want = ref_interleave(planar, descs, offs, sb, res, fill=0xee)
        pcm = np.full(want.size, 0xee, dtype=np.uint8)
        got = ctx.interleave(planar, descs, offs, sb, results=res, pcm=pcm)
        assert np.array_equal(got, want), (name, sb)


@pytest.mark.gpu
def test_gpu_decode_then_interleave_md5(oracle):
    """decode on the device -> interleave on the device -> MD5 on the host == the MD5 the file's STREAMINFO claims."""
    import torch
    ctx = cx.Context(0, wait_s=120)
    md5s = {"pop.flac": "68464288fa5e19835516972dcf47223c", "short.flac": "927598b89c89c1129a152eecfc14075e",
            "wasted_bits.flac": "4fbca4cf30f188453c0676e0cd700c71"}
    for name, md5 in md5s.items():
        data = np.frombuffer(open(os.path.join(FIXTURES, name), "rb").read(), dtype=np.uint8)
        st, _, si, audio_off = cx.read_stream_header(data)
        assert st == cx.OK
        descs, hdrs, stop = cx.index_frames(data, audio_off)
        assert stop == data.size and descs.size > 0
        sizes = descs["n_channels"].astype(np.uint64) * descs["block_size"].astype(np.uint64)
        offs = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.uint64)
        total = int(sizes.sum())
        sb = (si.bits_per_sample + 7) // 8
        arena = np.zeros(data.size + 64, dtype=np.uint8); arena[:data.size] = data
        d_arena = torch.from_numpy(arena).cuda()
        d_out = torch.zeros(total, dtype=torch.int32, device="cuda")
        d_pcm = torch.zeros(total * sb, dtype=torch.uint8, device="cuda")
        batch = ctx.plan(descs, offs, verify_crc=True)
        batch.run(d_arena.data_ptr(), data.size, d_out.data_ptr())
        batch.interleave(d_out.data_ptr(), d_pcm.data_ptr(), sb)
        res = batch.results()
        assert np.all(res["status"] == cx.OK), name
        assert hashlib.md5(d_pcm.cpu().numpy().tobytes()).hexdigest() == md5, name
        batch.close()
    # the bench workload shape: 16-bit stereo -> i16 pairs, against numpy
    w = synth.config3(64)
    descs = workload_descs(w)
    d_arena = torch.from_numpy(w.arena).cuda()
    d_out = torch.zeros(w.pcm.size, dtype=torch.int32, device="cuda")
    d_pcm = torch.zeros(w.pcm.size * 2, dtype=torch.uint8, device="cuda")
    batch = ctx.plan(descs, w.out_offs)
    batch.run(d_arena.data_ptr(), w.arena_len, d_out.data_ptr())
    batch.interleave(d_out.data_ptr(), d_pcm.data_ptr(), 2)
    assert np.all(batch.results()["status"] == cx.OK)
    assert np.array_equal(d_pcm.cpu().numpy(), ref_interleave(w.pcm, descs, w.out_offs, 2))
    batch.close()


@pytest.mark.gpu
def test_gpu_narrow_output_from_the_decode_md5(oracle):
    """CLX_OUT_PCM16 (round 5): the decode itself writes interleaved 16-bit PCM -- no planar i32, no narrowing pass -- and the bytes
    hash to the MD5 the 16-bit fixtures' STREAMINFO claims (pop.flac: stereo; short.flac: a block of four samples the lean kernel
    never takes; wasted_bits.flac: mono -- both through the planar scratch and clx_k_narrow_left)."""
    import torch
    ctx = cx.Context(0, wait_s=120)
    md5s = {"pop.flac": "68464288fa5e19835516972dcf47223c", "short.flac": "927598b89c89c1129a152eecfc14075e",
            "wasted_bits.flac": "4fbca4cf30f188453c0676e0cd700c71"}
    for name, md5 in md5s.items():
        data = np.frombuffer(open(os.path.join(FIXTURES, name), "rb").read(), dtype=np.uint8)
        st, _, si, audio_off = cx.read_stream_header(data)
        assert st == cx.OK and si.bits_per_sample == 16, name
        descs, hdrs, stop = cx.index_frames(data, audio_off)
        sizes = descs["n_channels"].astype(np.uint64) * descs["block_size"].astype(np.uint64)
        offs = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.uint64)
        total = int(sizes.sum())
        arena = np.zeros(data.size + 64, dtype=np.uint8); arena[:data.size] = data
        d_arena = torch.from_numpy(arena).cuda()
        d_pcm = torch.zeros(total + 8, dtype=torch.int16, device="cuda")
        batch = ctx.plan(descs, offs, verify_crc=True, path=cx.OUT_PCM16)
        batch.run(d_arena.data_ptr(), data.size, d_pcm.data_ptr())
        assert np.all(batch.results()["status"] == cx.OK), name
        assert hashlib.md5(d_pcm[:total].cpu().numpy().tobytes()).hexdigest() == md5, name
        batch.close()
    # a batch with a 24-bit frame is refused, and so is the combination with the wave kernels
    w = synth.config4(4)
    with pytest.raises(cx.ClaxonError):
        ctx.plan(workload_descs(w), w.out_offs, path=cx.OUT_PCM16)
    w = synth.config3(8)
    with pytest.raises(cx.ClaxonError):
        ctx.plan(workload_descs(w), w.out_offs, path=cx.OUT_PCM16 | cx.PATH_WAVES)
