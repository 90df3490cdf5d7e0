"""The parity tests proper: the HIP path on a real MI355X, through the C ABI, against the oracle."""
import numpy as np
import pytest

import claxon_amd as cx
import parity_cases as pc
import synth
from parity_util import GpuBackend

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    return cx.Context(0, wait_s=120)


@pytest.fixture(scope="module", params=[cx.PATH_WAVES | cx.K2_LATENCY, cx.PATH_WAVES | cx.K2_THROUGHPUT, cx.PATH_LANES | cx.LANES_SPLIT,
                                        cx.PATH_LANES | cx.LANES_FUSED, cx.PATH_LANES | cx.LANES_FUSED | cx.LANES_GENERAL,
                                        cx.PATH_LANES | cx.LANES_FUSED | cx.COMPOSE],
                ids=["waves", "waves-1w", "lanes", "lanes-fused", "lanes-general", "lanes-composed"])
def gpu(ctx, request):
    """Both kernel paths, every build of each: wave-per-frame (clx_kernels.hip) with the multi-wave and the one-wave
    predictor kernels, lane-per-subframe (clx_lanes.hip) with the split and the fused decode kernels -- the fused build with the
    lean 16-bit tier (clx_k_lean, clx_lean.hip) in front of the general kernels, with the general kernels alone, and with the waves
    composed by content (clx_k_compose: every window of stereo frames dealt to the lanes by class)."""
    return GpuBackend(ctx, request.param)


@pytest.mark.parametrize("make", [
    lambda: synth.config2(300), lambda: synth.config3(300), lambda: synth.config4(200),
    lambda: synth.config5_unique(400), lambda: synth.small_mixed(300),
], ids=["config2", "config3", "config4", "config5", "small_mixed"])
def test_gpu_workloads(oracle, gpu, make):
    pc.check_workload(oracle, gpu, make())


def test_gpu_lean_tiers(oracle, gpu):
    """every tier and exit of clx_k_lean (see parity_cases.lean_workload); the other selections decode the same frames"""
    pc.check_workload(oracle, gpu, pc.lean_workload(scale=2))


@pytest.mark.gpu
def test_gpu_lean24_tiers(oracle, gpu):
    """every tier and exit of clx_k_lean24, the split tier for > 16-bit audio and > 12 taps (see parity_cases.lean24_workload)"""
    pc.check_workload(oracle, gpu, pc.lean24_workload(scale=2))


def test_gpu_mid_side_that_runs_away(oracle, gpu):
    """parity_cases.ms_wild_workload: mid/side streams whose samples run away past every range check (and wrap): clx_k_lean's waves give
    them up, the general kernels decode them -- the oracle's wrapping arithmetic everywhere, with every kernel selection"""
    assert pc.check_ms_wild(oracle, gpu) > 0
    for bs in (256, 64):       # (short blocks, whose groups the slow turns' budget never gives up: the prologue and the slow turn do, by what they stage)
        assert pc.check_ms_wild(oracle, gpu, bs=bs) > 0


def test_gpu_edges(oracle, gpu):
    pc.check_workload(oracle, gpu, pc.edge_workload())


def test_gpu_never_resynchronising_streams(oracle, gpu):
    pc.check_workload(oracle, gpu, pc.resync_workload())


def test_gpu_range_hops(oracle, gpu):
    pc.check_workload(oracle, gpu, pc.range_hop_workload())


def test_gpu_footer_read_in_batch(oracle, gpu):
    pc.check_footer_read_in_batch(oracle, gpu)


def test_gpu_crc16_in_batch(oracle, gpu):
    """damaged frames inside batches, CRC-16 verified: "frame CRC mismatch" exactly where the oracle says -- from the decode lanes'
    own gathering (lean kernels, clx_crct.h) or from the stand-alone kernel, whichever the selection uses"""
    pc.check_crc_in_batch(oracle, gpu, synth.config3(300))
    pc.check_crc_in_batch(oracle, gpu, synth.config5_unique(400), seed=3)
    pc.check_crc_in_batch(oracle, gpu, synth.config4(120), seed=5)
    pc.check_crc_in_batch(oracle, gpu, synth.config3(300), seed=9, frac=0.1, loose_every=3)
    ws = pc.crc_share_workload()          # 1 .. 8 channels, short blocks, frames at every offset inside their first granule
    pc.check_workload(oracle, gpu, ws)
    pc.check_crc_in_batch(oracle, gpu, ws, seed=3, frac=0.3)


def test_gpu_truncations(oracle, gpu):
    pc.check_truncations(oracle, gpu, n_frames=10, cuts_per_frame=24)


def test_gpu_bitflips(oracle, gpu):
    seen = pc.check_bitflips(oracle, gpu, n_frames=24, trials=20)
    assert len(seen) >= 6


def test_gpu_fixtures(oracle, gpu):
    pc.check_fixtures(oracle, gpu)


def test_gpu_fuzz_corpus(oracle, gpu):
    pc.check_fuzz_corpus(oracle, gpu)


def test_gpu_prefill_independence(oracle, gpu):
    """fuzz/fuzzers/diff.rs idea: decode into buffers pre-filled with 13 and with 17 -- outputs must match."""
    w = synth.small_mixed(64, seed_off=77)
    descs = pc.workload_descs(w)
    a, ra = gpu.decode(w.arena, w.arena_len, descs, w.out_offs, True, fill=13)
    b, rb = gpu.decode(w.arena, w.arena_len, descs, w.out_offs, True, fill=17)
    assert np.array_equal(a, b) and np.array_equal(ra, rb)


def test_gpu_one_shot_host_api(oracle, gpu):
    """clx_decode_frames / clx_decode_subframes with HOST buffers (H2D + decode + D2H inside the call)."""
    w = synth.config3(40)
    descs = pc.workload_descs(w)
    out, res = gpu.ctx.decode_frames(w.arena[:w.arena_len], descs, w.out_offs, verify_crc=True)
    assert np.all(res["status"] == cx.OK) and np.array_equal(out, w.pcm)
    w = synth.config2(40)
    out, res = gpu.ctx.decode_subframes(w.arena[:w.arena_len], w.offs, w.block_sizes, w.bps, w.out_offs)
    assert np.all(res["status"] == cx.OK) and np.array_equal(out, w.pcm)


def test_gpu_crc_mismatch_detected(oracle, gpu):
    """A flipped bit in the *padding* leaves the decode intact but must fail the CRC-16 (frame.rs:761)."""
    from claxon_msgs import MSG
    w = synth.config3(4)
    arena = w.arena.copy()
    # corrupt the CRC footer of frame 2
    end = int(w.offs[2] + w.lens[2])
    arena[end - 1] ^= 0x01
    descs = pc.workload_descs(w)
    out, res = gpu.decode(arena, w.arena_len, descs, w.out_offs, True)
    assert res["status"].tolist() == [0, 0, cx.FORMAT_ERROR, 0]
    assert int(res["msg"][2]) == MSG["CLX_MSG_FRAME_CRC_MISMATCH"]


def test_gpu_flac_reader_streams(oracle, gpu):
    """FlacReader::open / blocks() (lib.rs:455, 367) on whole streams: host indexer + device batches."""
    import os
    from conftest import FIXTURES
    for name in ("pop.flac", "short.flac", "wasted_bits.flac", "non_subset.flac"):
        path = os.path.join(FIXTURES, name)
        si, blocks, st, msg = oracle.decode_stream(open(path, "rb").read())
        rd = cx.FlacReader.open(gpu.ctx, path)
        got = list(rd.blocks())
        assert len(got) == len(blocks)
        for b, (info, ref) in zip(got, blocks):
            assert (b.time(), b.duration(), b.channels()) == (info.time, info.block_size, info.channels)
            assert np.array_equal(b.into_buffer(), ref)
        assert rd.read_next_or_eof() is None
    # a longer synthetic stream: fLaC + STREAMINFO + 300 frames back to back
    w = synth.config5_unique(300)
    si = bytearray(34)
    si[0:2] = (4096).to_bytes(2, "big"); si[2:4] = (4096).to_bytes(2, "big")
    si[10:14] = ((44100 << 12) | (1 << 9) | (15 << 4)).to_bytes(4, "big")
    stream = b"fLaC" + bytes([0x80, 0, 0, 34]) + bytes(si) + w.arena[:w.arena_len].tobytes()
    rd = cx.FlacReader(gpu.ctx, data=stream)
    n = 0
    for i, b in enumerate(rd.blocks()):
        ref = w.pcm[int(w.out_offs[i]):int(w.out_offs[i]) + 2 * 4096]
        assert np.array_equal(b.into_buffer(), ref)
        n += 1
    assert n == 300
    # errors surface exactly like the reference: truncate mid-frame
    cut = stream[:len(stream) - 1000]
    rd = cx.FlacReader(gpu.ctx, data=cut)
    got, err = 0, None
    try:
        for b in rd.blocks():
            got += 1
    except cx.ClaxonError as e:
        err = e
    si2, blocks2, st2, msg2 = oracle.decode_stream(cut)
    assert got == len(blocks2) and err is not None and (err.status, err.msg) == (st2, msg2)


def test_gpu_regressions(oracle, gpu):
    pc.check_regressions(oracle, gpu)
