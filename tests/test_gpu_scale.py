"""Parity at scale: the kernel builds the library picks BY ITSELF at larger batches must actually run on the GPU and
agree with the oracle -- status, message, end_bit and every sample (not only the source PCM).

  * above 512 groups of predictor slots the wave path switches to the one-wave K2 builds (clx_k_predict_1w, and
    clx_k_predict_1w_hi for groups with a predictor order above 12)          -- subframe.rs:524-614
  * below, groups of aligned 16-bit rows of at most 8 taps go to clx_k_predict16, the rest to clx_k_predict (or to the one-wave
    kernels when the plan knows that every frame is 16-bit and aligned)
  * from 52 000 subframes of this shape (content dependent: clx_select_path, clx_plan.h) the default path is the lane
    kernels, fused build above 40 000 subframes                                  -- frame.rs:705-742
  * BASELINE configs 2 / 4 / 5 at >= 8 000 frames with flags 0: the selection the measurements ask for (tools/bench_configs.py)

Every case asserts WHICH kernels ran (names recorded by the library around its launches), so a silent change of the
selection thresholds cannot turn these tests into repeats of the small ones."""
import os

import numpy as np
import pytest

import claxon_amd as cx
import parity_cases as pc
import synth

pytestmark = pytest.mark.gpu

NTHREADS = max(1, min(64, os.cpu_count() or 1))


@pytest.fixture(scope="module")
def ctx():
    return cx.Context(0, wait_s=120)


@pytest.fixture(scope="module")
def big3():
    return synth.config3(28672)


def run_and_compare(oracle, ctx, w, flags, expect, forbid=()):
    import torch
    descs = pc.workload_descs(w)
    d_arena = torch.from_numpy(w.arena).to("cuda:0")
    d_out = torch.full((w.pcm.size,), 0x5a5a5a5a, dtype=torch.int32, device="cuda:0")
    crc = not w.bare_subframes
    batch = ctx.plan(descs, w.out_offs, verify_crc=crc, path=flags)
    batch.set_profiling(True)
    torch.cuda.synchronize()
    batch.run(d_arena.data_ptr(), w.arena_len, d_out.data_ptr())
    res = batch.results()
    ran = set(batch.kernel_times().keys())
    batch.close()
    assert set(expect) <= ran, (sorted(ran), expect)
    assert not (set(forbid) & ran), (sorted(ran), forbid)
    got = d_out.cpu().numpy()
    del d_out, d_arena
    ref = np.zeros(w.pcm.size, dtype=np.int32)
    if w.bare_subframes:
        r = oracle.decode_subframes(w.arena[:w.arena_len], w.offs, w.block_sizes, w.bps, out=ref, out_offs=w.out_offs)
    else:
        r = oracle.decode_batch(w.arena[:w.arena_len], w.offs, w.lens, out=ref, out_offs=w.out_offs, nthreads=NTHREADS)
    assert np.array_equal(res["status"], r["statuses"])
    assert np.array_equal(res["msg"], r["msgs"])
    assert np.array_equal(res["end_bit"], r["end_bits"])
    assert np.all(res["status"] == cx.OK)
    assert np.array_equal(got, ref)
    assert np.array_equal(got, w.pcm)
    return ran


def test_auto_wave_path_one_wave_predictor(oracle, ctx, big3):
    """20 480 config-3 frames + 640 order-32 frames: 42 240 slots < 48 000 keeps the wave path, 660 groups > 512 select
    the one-wave predictor builds; the order-32 frames make clx_k_predict_1w_hi do real work."""
    w = synth.concat("config3 x 20480 + config4 x 640", [pc.head(big3, 20480), synth.config4(640)])
    run_and_compare(oracle, ctx, w, 0, ["clx_k_residual", "clx_k_predict_1w", "clx_k_predict_1w_hi", "clx_k_crc16"],
                    forbid=["clx_k_predict", "clx_k_predict16", "clx_k_lanes"])


def test_auto_lane_path_fused(oracle, ctx, big3):
    """28 672 stereo frames = 57 344 subframes of ~5 bits per sample: the default is the lane path, fused build."""
    run_and_compare(oracle, ctx, big3, 0, ["clx_k_scan", "clx_k_lanes", "clx_k_finalize", "clx_k_crc16"],
                    forbid=["clx_k_residual", "clx_k_lanes2"])


def test_bench_configuration_against_the_oracle(oracle, ctx):
    """Exactly what bench.py times (BASELINE configs[2]): 10 000 stereo 4096-sample 16-bit mid/side LPC-8 frames, the library's own
    kernel choice, CRC-16 verified in the step, consecutive steps submitted with submit_depth output buffers in rotation (the fused
    lane kernels, lean 16-bit tier, several submissions merged per launch) -- every buffer, every status, message and end bit
    against the oracle, over more steps than buffers so that every scratch set and both internal streams are used twice."""
    import torch
    w = synth.config3(10000)
    descs = pc.workload_descs(w)
    d_arenas = [torch.from_numpy(w.arena).to("cuda:0") for _ in range(2)]
    batch = ctx.plan(descs, w.out_offs, verify_crc=True)
    depth = batch.submit_depth
    assert depth == cx.SUBMIT_DEPTH and batch.submit_lanes          # the pipelined choice for this workload: fused lane kernels
    outs = [torch.full((w.pcm.size,), 0x5a5a5a5a, dtype=torch.int32, device="cuda:0") for _ in range(depth)]
    st = torch.cuda.current_stream().cuda_stream
    for i in range(2 * depth + 5):
        batch.submit(d_arenas[i % 2].data_ptr(), w.arena_len, outs[i % depth].data_ptr(), st)
    batch.flush(st)
    torch.cuda.synchronize()
    res = batch.results()
    ref = np.zeros(w.pcm.size, dtype=np.int32)
    r = oracle.decode_batch(w.arena[:w.arena_len], w.offs, w.lens, out=ref, out_offs=w.out_offs, nthreads=NTHREADS)
    assert np.array_equal(res["status"], r["statuses"]) and np.all(res["status"] == cx.OK)
    assert np.array_equal(res["msg"], r["msgs"])
    assert np.array_equal(res["end_bit"], r["end_bits"])
    d_ref = torch.from_numpy(ref).to("cuda:0")
    for k, o in enumerate(outs):
        assert bool(torch.equal(o, d_ref)), "output buffer %d differs from the oracle" % k
    assert np.array_equal(ref, w.pcm)
    # one run at a time with the same kernels, profiled: the 16-bit tier decoded every group (the general kernels found nothing)
    bl = ctx.plan(descs, w.out_offs, verify_crc=True, path=cx.PATH_LANES | cx.LANES_FUSED)
    bl.set_profiling(True)
    outs[0].fill_(0x13131313)
    bl.run(d_arenas[0].data_ptr(), w.arena_len, outs[0].data_ptr())
    torch.cuda.synchronize()
    kt = bl.kernel_times()
    assert {"clx_k_scan", "clx_k_lean", "clx_k_lanes", "clx_k_finalize", "clx_k_crc16"} <= set(kt), sorted(kt)
    assert kt["clx_k_lanes"] < 0.05 * kt["clx_k_lean"], kt           # (ms: clx_k_lanes + _hi only looked at the taken flags)
    assert bool(torch.equal(outs[0], d_ref))
    bl.close(); batch.close()


def test_config4_at_scale_against_the_oracle(oracle, ctx):
    """BASELINE configs[3] at its full size: 10 000 stereo 4096-sample 24-bit frames, 32 taps of 15 bits, Rice2, wasted bits, every
    channel assignment -- pipelined submissions (the split tier clx_k_lean24 behind the scan), every buffer, status, message and end
    bit against the oracle; and one profiled run: the split tier decoded every group."""
    import torch
    w = synth.config4(10000)
    descs = pc.workload_descs(w)
    d_arena = torch.from_numpy(w.arena).to("cuda:0")
    batch = ctx.plan(descs, w.out_offs, verify_crc=True)
    depth = min(batch.submit_depth, 6)
    assert batch.submit_lanes
    outs = [torch.full((w.pcm.size,), 0x5a5a5a5a, dtype=torch.int32, device="cuda:0") for _ in range(depth)]
    st = torch.cuda.current_stream().cuda_stream
    for i in range(batch.submit_depth + 3):
        batch.submit(d_arena.data_ptr(), w.arena_len, outs[i % depth].data_ptr(), st)
    batch.flush(st)
    torch.cuda.synchronize()
    res = batch.results()
    ref = np.zeros(w.pcm.size, dtype=np.int32)
    r = oracle.decode_batch(w.arena[:w.arena_len], w.offs, w.lens, out=ref, out_offs=w.out_offs, nthreads=NTHREADS)
    assert np.array_equal(res["status"], r["statuses"]) and np.all(res["status"] == cx.OK)
    assert np.array_equal(res["msg"], r["msgs"])
    assert np.array_equal(res["end_bit"], r["end_bits"])
    d_ref = torch.from_numpy(ref).to("cuda:0")
    for k, o in enumerate(outs):
        assert bool(torch.equal(o, d_ref)), "output buffer %d differs from the oracle" % k
    assert np.array_equal(ref, w.pcm)
    bl = ctx.plan(descs, w.out_offs, verify_crc=True, path=cx.PATH_LANES | cx.LANES_FUSED)
    bl.set_profiling(True)
    outs[0].fill_(0x13131313)
    bl.run(d_arena.data_ptr(), w.arena_len, outs[0].data_ptr())
    torch.cuda.synchronize()
    kt = bl.kernel_times()
    assert {"clx_k_scan", "clx_k_lean24", "clx_k_lanes", "clx_k_finalize", "clx_k_crc16"} <= set(kt) and "clx_k_lean" not in kt, sorted(kt)
    assert kt["clx_k_lanes"] < 0.05 * kt["clx_k_lean24"], kt         # (ms: clx_k_lanes + _hi only looked at the taken flags)
    assert bool(torch.equal(outs[0], d_ref))
    bl.close(); batch.close()


def test_config2_at_scale_pipelined_against_the_oracle(oracle, ctx):
    """BASELINE configs[1] at its full size through pipelined submissions (clx_k_lean without a scan: mono subframes), every buffer
    against the oracle's decode of the bare subframes."""
    import torch
    w = synth.config2(10000)
    descs = pc.workload_descs(w)
    d_arena = torch.from_numpy(w.arena).to("cuda:0")
    batch = ctx.plan(descs, w.out_offs, verify_crc=False)
    depth = batch.submit_depth
    assert batch.submit_lanes and depth > 1
    outs = [torch.full((w.pcm.size,), 0x5a5a5a5a, dtype=torch.int32, device="cuda:0") for _ in range(depth)]
    st = torch.cuda.current_stream().cuda_stream
    for i in range(depth + 7):
        batch.submit(d_arena.data_ptr(), w.arena_len, outs[i % depth].data_ptr(), st)
    batch.flush(st)
    torch.cuda.synchronize()
    res = batch.results()
    ref = np.zeros(w.pcm.size, dtype=np.int32)
    r = oracle.decode_subframes(w.arena[:w.arena_len], w.offs, w.block_sizes, w.bps, out=ref, out_offs=w.out_offs)
    assert np.array_equal(res["status"], r["statuses"]) and np.all(res["status"] == cx.OK)
    assert np.array_equal(res["end_bit"], r["end_bits"])
    d_ref = torch.from_numpy(ref).to("cuda:0")
    for k, o in enumerate(outs):
        assert bool(torch.equal(o, d_ref)), "output buffer %d differs from the oracle" % k
    assert np.array_equal(ref, w.pcm)
    batch.close()


def test_rotation_over_more_buffers_than_depth(oracle, ctx):
    """Thirty output buffers under a depth of twenty-four, two distinct inputs in turn, no flush in between: a merged launch on one
    internal stream then writes buffers that a launch TWO launches earlier on the other stream wrote (not the latest one there) --
    the library has to order them (it keeps the last writer of every output buffer), or stale writes of the earlier launch could
    land on top of the later one's.  Every buffer must hold the decode of the LAST input submitted into it."""
    import torch
    ws = [synth.config3(1500), _seeded_config3(1500, 7000)]
    # both inputs under ONE plan: frame i of either sits at i * S, the descriptors only bound the frames (max_bytes = S)
    S = (int(max(int(w.lens.max()) for w in ws)) + 31) & ~15
    n = ws[0].n
    offs = (np.arange(n, dtype=np.uint64) * np.uint64(S))
    arenas = []
    for w in ws:
        a = np.zeros(n * S + 64, dtype=np.uint8)
        for i in range(n):
            a[i * S:i * S + int(w.lens[i])] = w.arena[int(w.offs[i]):int(w.offs[i] + w.lens[i])]
        arenas.append(a)
    lens = np.full(n, S, dtype=np.uint32)
    descs, _ = cx.descs_from_offsets(arenas[0][:n * S], offs, lens, check_crc=False)
    d2, _ = cx.descs_from_offsets(arenas[1][:n * S], offs, lens, check_crc=False)
    assert descs.tobytes() == d2.tobytes()
    refs = []
    for w, a in zip(ws, arenas):
        ref = np.zeros(w.pcm.size, dtype=np.int32)
        r = oracle.decode_batch(a[:n * S], offs, lens, out=ref, out_offs=w.out_offs, nthreads=NTHREADS)
        assert np.all(r["statuses"] == cx.OK) and np.array_equal(ref, w.pcm)
        refs.append(torch.from_numpy(ref).to("cuda:0"))
    d_arenas = [torch.from_numpy(a).to("cuda:0") for a in arenas]
    batch = ctx.plan(descs, ws[0].out_offs, verify_crc=True)
    assert batch.submit_lanes and batch.submit_depth == cx.SUBMIT_DEPTH
    nbuf = cx.SUBMIT_DEPTH + 6
    outs = [torch.full((ws[0].pcm.size,), 0x5a5a5a5a, dtype=torch.int32, device="cuda:0") for _ in range(nbuf)]
    st = torch.cuda.current_stream().cuda_stream
    last = {}
    for i in range(2 * nbuf + 15):
        which = (i * 7 // 3) & 1
        batch.submit(d_arenas[which].data_ptr(), n * S, outs[i % nbuf].data_ptr(), st)
        last[i % nbuf] = which
    batch.flush(st)
    torch.cuda.synchronize()
    res = batch.results()
    assert np.all(res["status"] == cx.OK)
    for k, o in enumerate(outs):
        assert bool(torch.equal(o, refs[last[k]])), "output buffer %d does not hold the last input submitted into it" % k
    batch.close()


def _seeded_config3(n, first):
    base = synth.BASE_SEED
    synth.BASE_SEED = base + first
    try:
        return synth.config3(n)
    finally:
        synth.BASE_SEED = base


def test_lean_give_up_at_scale(oracle, ctx):
    """10 000 frames of which every other wave gives its group up (parity_cases.giveup_workload: clx_k_lean's slow-turn budget), through
    pipelined submissions: half the groups are decoded twice -- started by clx_k_lean, abandoned, decoded from the start by
    clx_k_lanes -- and every sample, status and end bit still matches the oracle (no timing involved)."""
    import torch
    w = pc.giveup_workload(10000)
    descs = pc.workload_descs(w)
    d_arena = torch.from_numpy(w.arena).to("cuda:0")
    batch = ctx.plan(descs, w.out_offs, verify_crc=True)
    assert batch.submit_lanes
    depth = min(batch.submit_depth, 4)
    outs = [torch.full((w.pcm.size,), 0x5a5a5a5a, dtype=torch.int32, device="cuda:0") for _ in range(depth)]
    st = torch.cuda.current_stream().cuda_stream
    for i in range(batch.submit_depth + 2):
        batch.submit(d_arena.data_ptr(), w.arena_len, outs[i % depth].data_ptr(), st)
    batch.flush(st)
    torch.cuda.synchronize()
    res = batch.results()
    ref = np.zeros(w.pcm.size, dtype=np.int32)
    r = oracle.decode_batch(w.arena[:w.arena_len], w.offs, w.lens, out=ref, out_offs=w.out_offs, nthreads=NTHREADS)
    assert np.array_equal(res["status"], r["statuses"]) and np.all(res["status"] == cx.OK)
    assert np.array_equal(res["end_bit"], r["end_bits"])
    d_ref = torch.from_numpy(ref).to("cuda:0")
    for k, o in enumerate(outs):
        assert bool(torch.equal(o, d_ref)), "output buffer %d differs from the oracle" % k
    assert np.array_equal(ref, w.pcm)
    # both kernels really worked: one profiled run
    bl = ctx.plan(descs, w.out_offs, verify_crc=True, path=cx.PATH_LANES | cx.LANES_FUSED)
    bl.set_profiling(True)
    bl.run(d_arena.data_ptr(), w.arena_len, outs[0].data_ptr())
    torch.cuda.synchronize()
    kt = bl.kernel_times()
    assert kt["clx_k_lanes"] > 0.2 * kt["clx_k_lean"], kt              # (the general kernel decoded the given-up half)
    assert bool(torch.equal(outs[0], d_ref))
    bl.close(); batch.close()


def test_forced_builds_at_scale(oracle, ctx, big3):
    """The builds the thresholds would not pick at this size, forced by flag on the same 12 288 frames."""
    w = pc.head(big3, 12288)
    # (every frame 16-bit and aligned: what clx_k_predict16 leaves -- nothing here -- goes to the one-wave kernels, not to clx_k_predict)
    run_and_compare(oracle, ctx, w, cx.PATH_WAVES | cx.K2_LATENCY, ["clx_k_residual", "clx_k_predict16", "clx_k_predict_1w"], forbid=["clx_k_predict"])
    # 24-bit frames in the batch: the general multi-wave kernel takes their groups
    wm = synth.concat("config3 x 2048 + config4 x 512", [pc.head(big3, 2048), synth.config4(512)])
    run_and_compare(oracle, ctx, wm, cx.PATH_WAVES | cx.K2_LATENCY, ["clx_k_residual", "clx_k_predict16", "clx_k_predict"], forbid=["clx_k_predict_1w"])
    run_and_compare(oracle, ctx, w, cx.PATH_WAVES | cx.K2_THROUGHPUT, ["clx_k_residual", "clx_k_predict_1w"])
    run_and_compare(oracle, ctx, w, cx.PATH_LANES | cx.LANES_SPLIT, ["clx_k_lanes2"])
    run_and_compare(oracle, ctx, w, cx.PATH_LANES | cx.LANES_FUSED, ["clx_k_lanes"])


@pytest.mark.parametrize("make,expect,forbid", [
    (lambda: synth.config2(8192), ["clx_k_residual", "clx_k_predict16"], ["clx_k_lanes", "clx_k_lanes2"]),     # 8 192 mono subframes, 5.7 bits/sample
    (lambda: synth.config4(8192), ["clx_k_lanes2"], ["clx_k_residual"]),                                      # 24-bit: lane kernels, two-wave build
    (lambda: synth.config5_unique(8192), ["clx_k_residual", "clx_k_predict16"], ["clx_k_lanes", "clx_k_lanes2"]),  # 16 384 subframes at 9.5 bits/sample
    (lambda: synth.config5_unique(12288), ["clx_k_lanes2"], ["clx_k_residual"]),                              # 24 576 of them: lane kernels
], ids=["config2", "config4", "config5", "config5-more"])
def test_baseline_configs_default_selection(oracle, ctx, make, expect, forbid):
    run_and_compare(oracle, ctx, make(), 0, expect, forbid=forbid)


def test_pipelined_submissions_of_the_lane_kernels(oracle, ctx):
    """6 144 config-5 frames with flags 0: one run at a time takes the library's choice for a run, pipelined submissions the fused
    lane kernels, 24 in flight (two merged launches of twelve) with a set of scratch buffers each (clx_select_path, `pipelined`;
    clx_batch_submit_depth) -- with a plain run in between, on the same batch.  All of it against the oracle."""
    import torch
    w = synth.config5_unique(6144)
    descs = pc.workload_descs(w)
    d_arena = torch.from_numpy(w.arena).to("cuda:0")
    batch = ctx.plan(descs, w.out_offs, verify_crc=True)
    assert batch.submit_depth == cx.SUBMIT_DEPTH == 24 and batch.submit_lanes
    outs = [torch.full((w.pcm.size,), 0x5a5a5a5a, dtype=torch.int32, device="cuda:0") for _ in range(batch.submit_depth + 1)]
    torch.cuda.synchronize()
    for i in range(2 * batch.submit_depth + 3):
        batch.submit(d_arena.data_ptr(), w.arena_len, outs[i % batch.submit_depth].data_ptr())
    batch.run(d_arena.data_ptr(), w.arena_len, outs[-1].data_ptr())
    batch.submit(d_arena.data_ptr(), w.arena_len, outs[0].data_ptr())
    batch.submit(d_arena.data_ptr(), w.arena_len, outs[0].data_ptr())          # the same buffer again: waits for the writer before
    res = batch.results()
    batch.close()
    ref = np.zeros(w.pcm.size, dtype=np.int32)
    r = oracle.decode_batch(w.arena[:w.arena_len], w.offs, w.lens, out=ref, out_offs=w.out_offs, nthreads=NTHREADS)
    assert np.array_equal(res["status"], r["statuses"]) and np.all(res["status"] == cx.OK)
    assert np.array_equal(res["end_bit"], r["end_bits"])
    for o in outs:
        assert np.array_equal(o.cpu().numpy(), ref)


def test_composed_waves_at_scale(oracle, ctx):
    """12 288 mixed config-5 frames through pipelined submissions with the waves composed by content (the default for a batch whose
    descriptors differ in their channel assignment: two windows of clx_k_compose): every buffer, status, message and end bit against
    the oracle; the same in stream order (CLX_NO_COMPOSE); and a profiled run shows clx_k_compose in the chain."""
    import torch
    w = synth.config5_unique(12288)
    descs = pc.workload_descs(w)
    d_arena = torch.from_numpy(w.arena).to("cuda:0")
    ref = np.zeros(w.pcm.size, dtype=np.int32)
    r = oracle.decode_batch(w.arena[:w.arena_len], w.offs, w.lens, out=ref, out_offs=w.out_offs, nthreads=NTHREADS)
    assert np.array_equal(ref, w.pcm)
    d_ref = torch.from_numpy(ref).to("cuda:0")
    for flags in (0, cx.NO_COMPOSE):
        batch = ctx.plan(descs, w.out_offs, verify_crc=True, path=flags)
        assert batch.submit_lanes
        depth = min(batch.submit_depth, 6)
        outs = [torch.full((w.pcm.size,), 0x5a5a5a5a, dtype=torch.int32, device="cuda:0") for _ in range(depth)]
        st = torch.cuda.current_stream().cuda_stream
        for i in range(batch.submit_depth + 3):
            batch.submit(d_arena.data_ptr(), w.arena_len, outs[i % depth].data_ptr(), st)
        batch.flush(st)
        torch.cuda.synchronize()
        res = batch.results()
        assert np.array_equal(res["status"], r["statuses"]) and np.all(res["status"] == cx.OK)
        assert np.array_equal(res["msg"], r["msgs"]) and np.array_equal(res["end_bit"], r["end_bits"])
        for k, o in enumerate(outs):
            assert bool(torch.equal(o, d_ref)), "output buffer %d differs from the oracle (flags %d)" % (k, flags)
        batch.close()
        bl = ctx.plan(descs, w.out_offs, verify_crc=True, path=cx.PATH_LANES | cx.LANES_FUSED | flags)
        bl.set_profiling(True)
        outs[0].fill_(0x13131313)
        bl.run(d_arena.data_ptr(), w.arena_len, outs[0].data_ptr())
        torch.cuda.synchronize()
        kt = bl.kernel_times()
        assert ("clx_k_compose" in kt) == (flags == 0), sorted(kt)
        assert bool(torch.equal(outs[0], d_ref))
        bl.close()


def test_config5_rank_share_against_the_oracle(oracle, ctx):
    """The node-scale workload's per-GPU share at its full size (BASELINE configs[4]: rank 3 of 8 of the 1 M-frame job, 125 003 frames
    of 16 384 unique ones tiled with their own frame numbers and CRCs; what `bench.py --workload config5 --shard-of 8 --shard-rank 3`
    times): pipelined submissions, waves composed by content, CRC-16 verified -- every status, message, end bit and sample against
    the ORACLE's decode of the very same 1.2 GB of frames (not the generator's PCM)."""
    import torch
    from claxon_amd import shard
    ts = synth.config5_tiled(1_000_000, 16384)
    lo, hi = shard.balanced_ranges(ts.weights(), 8)[3]
    w = ts.slice(lo, hi)
    assert w.n >= 125000
    descs, _ = cx.descs_from_offsets(w.arena[:w.arena_len], w.offs, w.lens)
    ref = np.zeros(w.total_samples, dtype=np.int32)
    r = oracle.decode_batch(w.arena[:w.arena_len], w.offs, w.lens, out=ref, out_offs=w.out_offs, nthreads=NTHREADS)
    assert np.all(r["statuses"] == cx.OK)
    d_arena = torch.from_numpy(w.arena).to("cuda:0")
    d_ref = torch.from_numpy(ref).to("cuda:0")
    del ref
    batch = ctx.plan(descs, w.out_offs, verify_crc=True)
    assert batch.submit_lanes
    outs = [torch.full((w.total_samples,), 0x5a5a5a5a, dtype=torch.int32, device="cuda:0") for _ in range(3)]
    st = torch.cuda.current_stream().cuda_stream
    for i in range(batch.submit_depth + 3):
        batch.submit(d_arena.data_ptr(), w.arena_len, outs[i % 3].data_ptr(), st)
    batch.flush(st)
    torch.cuda.synchronize()
    res = batch.results()
    assert np.array_equal(res["status"], r["statuses"]) and np.array_equal(res["msg"], r["msgs"])
    assert np.array_equal(res["end_bit"], r["end_bits"])
    for k, o in enumerate(outs):
        assert bool(torch.equal(o, d_ref)), "output buffer %d differs from the oracle" % k
    batch.close()


def test_device_indexer_against_oracle_offsets(oracle, ctx):
    """clx_index_frames_device against the frame starts the ORACLE's reader walks through (oracle.decode_stream), not
    against the product's own host indexer: a 3 MB raw stream of mixed frames, with and without a garbage tail."""
    w = synth.config5_unique(600)
    stream = w.arena[:w.arena_len].tobytes()
    hdr = bytearray(34)
    hdr[0:2] = (4096).to_bytes(2, "big"); hdr[2:4] = (4096).to_bytes(2, "big")
    hdr[10:14] = ((44100 << 12) | (1 << 9) | (15 << 4)).to_bytes(4, "big")
    head = b"fLaC" + bytes([0x80, 0, 0, 34]) + bytes(hdr)
    for tail in (b"", bytes(range(256)) * 3):
        data = head + stream + tail
        si, blocks, st, msg = oracle.decode_stream(data)
        starts, pos = [], len(head)
        for info, _ in blocks:
            starts.append(pos)
            pos += int(info.bytes_consumed)
        assert len(starts) == 600
        descs, hdrs, stop = ctx.index_frames(np.frombuffer(data, dtype=np.uint8), start=len(head))
        # the index lists the frames whose END is confirmed (a CRC-valid header, or the end of the stream, right behind their
        # CRC-16): with garbage behind the last frame that one stays for the reader to decode on its own (it does:
        # test_gpu_flac_reader_streams), and `stop` says where it starts
        n_idx = 600 if not tail else 599
        assert descs["byte_off"].tolist() == starts[:n_idx]
        assert stop == (pos if not tail else starts[599])
        assert hdrs["block_size"].tolist() == [int(i.block_size) for i, _ in blocks[:n_idx]]
        assert hdrs["time"].tolist() == [int(i.time) for i, _ in blocks[:n_idx]]


def test_everything_left_to_the_general_kernels_behind_the_tiers(oracle, ctx):
    """16-bit audio with 20-tap predictors and no wider frame in the batch: clx_k_lean takes no group (more than 12 taps), the split tier
    is not launched (nothing beyond 16 bits) -- EVERY group is on the list clx_k_left makes, for a reason only the stream knows.  Through
    pipelined submissions: the first merged launch runs the general kernels with the small grid the descriptors suggest (each workgroup
    loops over several groups), later ones with a grid sized by the longest list seen; every output buffer against the oracle."""
    import torch
    S = synth
    n, bs = 3072, 1024
    rng = np.random.default_rng(77)
    t = np.arange(bs)
    pcm = np.empty((n, 2, bs), dtype=np.int32)
    for i in range(n):
        for c in range(2):
            pcm[i, c] = np.clip(np.round(4000.0 * np.sin(2 * np.pi * (50 + i % 97 + 11 * c) * t / 44100.0) + rng.normal(0, 6.0, bs)), -32768, 32767)
    fp = [S.FrameParams() for _ in range(n)]
    for i, f in enumerate(fp):
        f.channel_assignment = i % 4
        f.sf[0] = S.sf(S.SF_LPC, order=20, precision=12, partition_order=3)
        f.sf[1] = S.sf(S.SF_LPC, order=16 + i % 5, precision=12, partition_order=2)
    w = S.encode_frames("16-bit, 16-20 taps", pcm, 2, bs, 16, fp)
    descs = pc.workload_descs(w)
    d_arena = torch.from_numpy(w.arena).to("cuda:0")
    batch = ctx.plan(descs, w.out_offs, verify_crc=True)
    assert batch.submit_lanes
    depth = min(batch.submit_depth, 5)
    outs = [torch.full((w.pcm.size,), 0x5a5a5a5a, dtype=torch.int32, device="cuda:0") for _ in range(depth)]
    st = torch.cuda.current_stream().cuda_stream
    for i in range(3 * batch.submit_merge + 7):
        batch.submit(d_arena.data_ptr(), w.arena_len, outs[i % depth].data_ptr(), st)
        if i == batch.submit_merge:          # (let the first launch's list length come back before the later launches are sized)
            batch.flush(st); torch.cuda.synchronize()
    batch.flush(st)
    torch.cuda.synchronize()
    res = batch.results()
    ref = np.zeros(w.pcm.size, dtype=np.int32)
    r = oracle.decode_batch(w.arena[:w.arena_len], w.offs, w.lens, out=ref, out_offs=w.out_offs, nthreads=NTHREADS)
    assert np.array_equal(res["status"], r["statuses"]) and np.all(res["status"] == cx.OK)
    assert np.array_equal(res["end_bit"], r["end_bits"])
    assert np.array_equal(ref, w.pcm)
    d_ref = torch.from_numpy(ref).to("cuda:0")
    for k, o in enumerate(outs):
        assert bool(torch.equal(o, d_ref)), "output buffer %d differs from the oracle" % k
    # which kernels worked: clx_k_lean found nothing to take, the order > 12 twin of the general kernels decoded everything
    bl = ctx.plan(descs, w.out_offs, verify_crc=True, path=cx.PATH_LANES | cx.LANES_FUSED)
    bl.set_profiling(True)
    bl.run(d_arena.data_ptr(), w.arena_len, outs[0].data_ptr())
    torch.cuda.synchronize()
    kt = bl.kernel_times()
    assert "clx_k_left" in kt and kt["clx_k_lanes"] > 5.0 * kt["clx_k_lean"], kt
    bl.close(); batch.close()
