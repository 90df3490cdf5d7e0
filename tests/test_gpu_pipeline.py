"""clx_batch_submit / clx_batch_flush: several submissions in flight (whole runs on internal streams, each with its own scratch
buffers and results: four of the wave kernels, twelve of the fused lane kernels) must give exactly what clx_batch_run gives -- the oracle's samples, statuses and end bits -- whatever the
caller does with its output buffers: a rotation over SUBMIT_DEPTH buffers, two alternating buffers or the same buffer every time
(the library then waits for the earlier writer), runs and submissions mixed, with the CRC-16 kernel in the step, and for the
kernel selections that fall back to plain runs."""
import numpy as np
import pytest

import claxon_amd as cx
import parity_cases as pc
import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setup(oracle):
    import torch
    ctx = cx.Context(0, wait_s=120)
    w = synth.concat("mix", [synth.config3(2500), synth.config5_unique(500), synth.small_mixed(150, seed_off=33)])
    arena = w.arena.copy()
    arena[int(w.offs[1234] + w.lens[1234]) - 1] ^= 0x40                        # one frame whose CRC-16 no longer matches
    ref = np.zeros(w.pcm.size, dtype=np.int32)
    r = oracle.decode_batch(arena[:w.arena_len], w.offs, w.lens, out=ref, out_offs=w.out_offs, nthreads=8)
    d_arena = torch.from_numpy(arena).to("cuda:0")
    torch.cuda.synchronize()
    return ctx, w, pc.workload_descs(w), d_arena, ref, r


def check(w, out, res, ref, r, crc):
    got = out.cpu().numpy()
    want_status = r["statuses"].copy()
    if not crc:
        want_status[1234] = 0                                                   # without the comparison the damaged footer goes unnoticed
    assert np.array_equal(res["status"], want_status)
    ok = want_status == 0
    assert np.array_equal(res["end_bit"][ok], r["end_bits"][ok])
    for i in np.nonzero(ok)[0][::7]:
        a = int(w.out_offs[i]); b = a + int(w.channels[i]) * int(w.block_sizes[i])
        assert np.array_equal(got[a:b], ref[a:b]), i
    bad = ~ok
    mask = np.ones(got.size, dtype=bool)
    for i in np.nonzero(bad)[0]:
        a = int(w.out_offs[i]); mask[a:a + int(w.channels[i]) * int(w.block_sizes[i])] = False
    assert np.array_equal(got[mask], ref[mask])


@pytest.mark.parametrize("flags,crc", [(cx.PATH_WAVES | cx.K2_LATENCY, False), (cx.PATH_WAVES | cx.K2_LATENCY, True), (0, True),
                                       (cx.PATH_WAVES | cx.K2_THROUGHPUT, True), (cx.PATH_LANES | cx.LANES_SPLIT, True),
                                       (cx.PATH_LANES | cx.LANES_FUSED, True), (cx.PATH_LANES | cx.LANES_FUSED, False)],
                         ids=["waves", "waves-crc", "auto-crc", "waves-1w-crc", "lanes-crc", "lanes-fused-crc", "lanes-fused"])
def test_submit_matches_run(setup, flags, crc):
    import torch
    ctx, w, descs, d_arena, ref, r = setup
    outs = [torch.full((w.pcm.size,), 0x5a5a5a5a, dtype=torch.int32, device="cuda:0") for _ in range(cx.SUBMIT_DEPTH)]
    torch.cuda.synchronize()
    b = ctx.plan(descs, w.out_offs, verify_crc=crc, path=flags)
    # a buffer per submission in flight
    for i in range(2 * cx.SUBMIT_DEPTH + 1):
        b.submit(d_arena.data_ptr(), w.arena_len, outs[i % cx.SUBMIT_DEPTH].data_ptr())
    res = b.results()                                                          # flushes
    for o in outs:
        check(w, o, res, ref, r, crc)
    # two alternating buffers: every submission waits for the one before the previous one
    for o in outs:
        o.fill_(0x2b2b2b2b)
    torch.cuda.synchronize()
    for i in range(5):
        b.submit(d_arena.data_ptr(), w.arena_len, outs[i & 1].data_ptr())
    res = b.results()
    for o in outs[:2]:
        check(w, o, res, ref, r, crc)
    # the same buffer every time, a plain run in between, then a flush and the narrow stage on the last output
    for o in outs:
        o.fill_(0x13131313)
    torch.cuda.synchronize()
    b.submit(d_arena.data_ptr(), w.arena_len, outs[0].data_ptr())
    b.submit(d_arena.data_ptr(), w.arena_len, outs[0].data_ptr())
    b.run(d_arena.data_ptr(), w.arena_len, outs[1].data_ptr())
    b.submit(d_arena.data_ptr(), w.arena_len, outs[0].data_ptr())
    b.flush()
    res = b.results()
    check(w, outs[0], res, ref, r, crc)
    check(w, outs[1], res, ref, r, crc)
    b.close()


def test_merged_launch_profiling(setup):
    """clx_batch_set_profiling(b, 2): pipelined submissions go out as usual and the per-kernel events bracket the kernels of the
    merged launch (what bench.py's roofline.merged_launch reports); the outputs are still exact and profiling can be switched off."""
    import torch
    ctx, w, descs, d_arena, ref, r = setup
    b = ctx.plan(descs, w.out_offs, verify_crc=True, path=cx.PATH_LANES | cx.LANES_FUSED)
    depth = b.submit_depth
    assert depth > 1 and b.submit_lanes
    outs = [torch.full((w.pcm.size,), 0x5a5a5a5a, dtype=torch.int32, device="cuda:0") for _ in range(depth // 2)]
    torch.cuda.synchronize()
    b.set_profiling(2)
    assert b.submit_depth == depth                                  # (mode 1 would turn submissions into plain runs)
    for o in outs:
        b.submit(d_arena.data_ptr(), w.arena_len, o.data_ptr())
    b.flush()
    torch.cuda.synchronize()
    kt = b.kernel_times()
    assert "clx_k_finalize" in kt and "clx_k_crc16" in kt and ("clx_k_lean" in kt or "clx_k_lean24" in kt or "clx_k_lanes" in kt), sorted(kt)
    assert all(v >= 0.0 for v in kt.values())
    res = b.results()
    for o in outs:
        check(w, o, res, ref, r, True)
    b.set_profiling(False)
    b.submit(d_arena.data_ptr(), w.arena_len, outs[0].data_ptr())
    res = b.results()
    check(w, outs[0], res, ref, r, True)
    b.close()


def test_small_batch_choices(setup):
    """flags 0 on a small batch: one run at a time takes the wave kernels (the lower latency), pipelined submissions the merged lane
    kernels (round 3: ahead at every size) -- and the lane kernels' plan data, made with the first submission, serves both orders."""
    import torch
    ctx, w, descs, d_arena, ref, r = setup
    b = ctx.plan(descs, w.out_offs, verify_crc=True)
    assert b.submit_lanes and b.submit_depth == cx.SUBMIT_DEPTH
    out = torch.full((w.pcm.size,), 0x5a5a5a5a, dtype=torch.int32, device="cuda:0")
    torch.cuda.synchronize()
    b.set_profiling(True)
    b.run(d_arena.data_ptr(), w.arena_len, out.data_ptr())
    torch.cuda.synchronize()
    assert "clx_k_residual" in b.kernel_times()                     # a plain run of this batch: wave kernels
    b.set_profiling(2)
    outs = [torch.full((w.pcm.size,), 0x5a5a5a5a, dtype=torch.int32, device="cuda:0") for _ in range(3)]
    for o in outs:
        b.submit(d_arena.data_ptr(), w.arena_len, o.data_ptr())
    b.flush()
    torch.cuda.synchronize()
    kt = b.kernel_times()
    assert "clx_k_residual" not in kt and "clx_k_finalize" in kt, sorted(kt)      # the merged launch: lane kernels
    res = b.results()
    for o in outs:
        check(w, o, res, ref, r, True)
    b.set_profiling(False)
    b.run(d_arena.data_ptr(), w.arena_len, out.data_ptr())
    res = b.results()
    check(w, out, res, ref, r, True)
    b.close()


@pytest.mark.gpu
def test_bench_line_with_the_collectives_through_rccl_on_one_rank():
    """`bench.py --process-group --backend nccl` with one rank: init_process_group("nccl", device_id=...), the barrier around the
    timed regions and the MAX / SUM / all_gather reductions on device tensors run through RCCL on the GPU that is there -- the
    N-rank line's collectives on hardware (round 5; no second GPU on this box).  The closing barrier is outside the clock."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    env["MASTER_ADDR"] = "127.0.0.1"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--process-group", "--backend", "nccl", "--frames", "1024", "--steps", "8",
                        "--warmup", "2", "--repeats", "2", "--no-cpu-baseline", "--no-extras"], env=env, capture_output=True, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    pg = j["config"]["process_group"]
    assert pg["backend"].startswith("nccl") and pg["world_size"] == 1 and pg["barrier_us"] is not None and pg["barrier_us"] >= 0
    assert j["n_gpus"] == 1 and j["value"] > 0 and j["config"]["bit_exact"] is True


@pytest.mark.gpu
def test_narrow_output_parity_on_the_gpu(oracle):
    """parity_cases.pcm16_workload through CLX_OUT_PCM16 on the GPU (one run at a time: the general kernels keep their full grid), in
    stream order, composed, and with a fifth of the frames damaged: every OK frame's bytes against the oracle."""
    from parity_util import GpuBackend
    ctx = cx.Context(0, wait_s=120)
    w = pc.pcm16_workload()
    for extra in (cx.NO_COMPOSE, cx.COMPOSE):
        assert pc.check_pcm16(oracle, GpuBackend(ctx, cx.OUT_PCM16 | extra), w) == w.n
    assert pc.check_pcm16(oracle, GpuBackend(ctx, cx.OUT_PCM16), w, damage=0.2, seed=3) < w.n
    ctx.close()


@pytest.mark.gpu
def test_narrow_output_pipelined_at_scale(oracle):
    """4 096 config-3 frames + the give-up workload through pipelined submissions with CLX_OUT_PCM16 (merged launches; the groups that
    are given up go through the general kernels' staging rows -- one allocation per internal stream since round 6, not a planar scratch
    per flight -- with the SMALL grid): every output buffer holds the interleaved low 16 bits of the source PCM."""
    import torch
    ctx = cx.Context(0, wait_s=120)
    w = synth.concat("pcm16 at scale", [synth.config3(4096), pc.giveup_workload(1024)])
    descs = pc.workload_descs(w)
    d_arena = torch.from_numpy(w.arena).to("cuda:0")
    batch = ctx.plan(descs, w.out_offs, verify_crc=True, path=cx.OUT_PCM16)
    assert batch.submit_lanes
    depth = min(batch.submit_depth, 6)
    outs = [torch.full((w.pcm.size + 8,), 0x1111, dtype=torch.int16, device="cuda:0") for _ in range(depth)]
    st = torch.cuda.current_stream().cuda_stream
    for i in range(batch.submit_depth + 5):
        batch.submit(d_arena.data_ptr(), w.arena_len, outs[i % depth].data_ptr(), st)
    batch.flush(st)
    torch.cuda.synchronize()
    res = batch.results()
    assert np.all(res["status"] == cx.OK)
    want = np.zeros(w.pcm.size, dtype=np.int16)
    for i in range(w.n):
        a, c, bs = int(w.out_offs[i]), int(w.channels[i]), int(w.block_sizes[i])
        want[a:a + c * bs] = w.pcm[a:a + c * bs].reshape(c, bs).T.reshape(-1).astype(np.int16)
    d_want = torch.from_numpy(want).to("cuda:0")
    for k, o in enumerate(outs):
        assert bool(torch.equal(o[:w.pcm.size], d_want)), "output buffer %d" % k
    batch.close(); ctx.close()


@pytest.mark.gpu
def test_packed_24_bit_output_on_the_gpu(oracle):
    """CLX_OUT_PCM24 (round 6) on the GPU: parity_cases.pcm24_workload one run at a time, intact and damaged, and pipelined (merged
    launches: several runs' staging rows side by side in the stream's allocation) -- every OK frame's bytes against the oracle."""
    import torch
    from parity_util import GpuBackend
    ctx = cx.Context(0, wait_s=120)
    w = pc.pcm24_workload()
    assert pc.check_pcm24(oracle, GpuBackend(ctx, cx.OUT_PCM24), w) == w.n
    assert pc.check_pcm24(oracle, GpuBackend(ctx, cx.OUT_PCM24), w, damage=0.2, seed=5) < w.n
    descs = pc.workload_descs(w)
    d_arena = torch.from_numpy(w.arena).to("cuda:0")
    batch = ctx.plan(descs, w.out_offs, verify_crc=True, path=cx.OUT_PCM24)
    outs = [torch.full((3 * w.pcm.size + 16,), 0x11, dtype=torch.uint8, device="cuda:0") for _ in range(4)]
    st = torch.cuda.current_stream().cuda_stream
    for i in range(batch.submit_depth + 3):
        batch.submit(d_arena.data_ptr(), w.arena_len, outs[i % 4].data_ptr(), st)
    batch.flush(st)
    torch.cuda.synchronize()
    assert np.all(batch.results()["status"] == cx.OK)
    v = np.zeros(w.pcm.size, dtype=np.int32)
    for i in range(w.n):
        a, c, bs = int(w.out_offs[i]), int(w.channels[i]), int(w.block_sizes[i])
        v[a:a + c * bs] = w.pcm[a:a + c * bs].reshape(c, bs).T.reshape(-1)
    u = v.view(np.uint32)
    want = torch.from_numpy(np.stack([u & 0xff, (u >> 8) & 0xff, (u >> 16) & 0xff], axis=1).astype(np.uint8).reshape(-1)).to("cuda:0")
    covered = np.zeros(3 * w.pcm.size, dtype=bool)
    for i in range(w.n):
        a, c, bs = int(w.out_offs[i]), int(w.channels[i]), int(w.block_sizes[i])
        covered[3 * a:3 * (a + c * bs)] = True
    d_cov = torch.from_numpy(covered).to("cuda:0")
    for k, o in enumerate(outs):
        assert bool(torch.equal(o[:3 * w.pcm.size][d_cov], want[d_cov])), "output buffer %d" % k
    batch.close(); ctx.close()


@pytest.mark.gpu
def test_pool_tickets_on_the_gpu(oracle):
    """CLX_POOL (round 6) on the GPU: merged launches whose scan waves and 16-bit-tier decode waves are tickets of one resident grid
    (clx_k_pool) -- more submissions than output buffers (every buffer is re-used while launches are in flight), a workload with groups
    the lean tier leaves and gives up (the general kernels behind the pool), and a damaged arena among the intact ones: every buffer
    against the oracle's decode of what was submitted into it last."""
    import torch
    ctx = cx.Context(0, wait_s=120)
    w = synth.concat("pool on the gpu", [synth.config3(1500), pc.giveup_workload(256), synth.config5_unique(300), synth.small_mixed(80)])
    descs = pc.workload_descs(w)
    rng = np.random.default_rng(2026)
    bad = w.arena.copy()
    for i in rng.choice(w.n, size=w.n // 10, replace=False):
        lo, hi = int(w.offs[i]) + int(descs["header_bytes"][i]), int(w.offs[i] + w.lens[i])
        pos = int(rng.integers(8 * lo, 8 * hi))
        bad[pos >> 3] ^= (0x80 >> (pos & 7))
    d_ok, d_bad = torch.from_numpy(w.arena).to("cuda:0"), torch.from_numpy(bad).to("cuda:0")
    batch = ctx.plan(descs, w.out_offs, verify_crc=True, path=cx.PATH_LANES | cx.LANES_FUSED | cx.NO_COMPOSE | cx.POOL)
    assert batch.submit_lanes
    outs = [torch.full((w.pcm.size,), 0x5a5a5a5a, dtype=torch.int32, device="cuda:0") for _ in range(5)]
    st = torch.cuda.current_stream().cuda_stream
    n_sub = 2 * batch.submit_depth + 7
    last = {}
    for i in range(n_sub):
        damaged = (i % 9 == 4)
        batch.submit((d_bad if damaged else d_ok).data_ptr(), w.arena_len, outs[i % 5].data_ptr(), st)
        last[i % 5] = damaged
    batch.flush(st)
    torch.cuda.synchronize()
    res = batch.results()                                   # (the LAST submission's)
    ref_ok = np.zeros(w.pcm.size, dtype=np.int32)
    r_ok = oracle.decode_batch(w.arena[:w.arena_len], w.offs, w.lens, out=ref_ok, out_offs=w.out_offs, check_crc=True)
    ref_bad = np.zeros(w.pcm.size, dtype=np.int32)
    r_bad = oracle.decode_batch(bad[:w.arena_len], w.offs, w.lens, out=ref_bad, out_offs=w.out_offs, check_crc=True)
    r_last = r_bad if ((n_sub - 1) % 9 == 4) else r_ok
    assert np.array_equal(res["status"], r_last["statuses"]) and np.array_equal(res["msg"], r_last["msgs"])
    assert np.all(r_ok["statuses"] == cx.OK) and int(np.sum(r_bad["statuses"] != cx.OK)) >= 10
    for k, o in enumerate(outs):
        got = o.cpu().numpy()
        ref, r = (ref_bad, r_bad) if last[k] else (ref_ok, r_ok)
        for i in np.nonzero(r["statuses"] == cx.OK)[0]:
            lo, hi = int(w.out_offs[i]), int(w.out_offs[i]) + int(w.channels[i]) * int(w.block_sizes[i])
            assert np.array_equal(got[lo:hi], ref[lo:hi]), (k, int(i))
    batch.close(); ctx.close()


@pytest.mark.gpu
def test_mid_side_undone_by_the_movers_on_the_gpu(oracle):
    """parity_cases.ms_mover_workload on the GPU: waves of plain mid/side pairs, whose turns stage mid and side as decoded and whose movers write
    left and right (cln_ms4, round 6) -- planar and CLX_OUT_PCM16, a ragged last wave of whole pairs of tiles and one that ends in a lone tile,
    damaged frames, the frames' CRC-16."""
    from parity_util import GpuBackend
    ctx = cx.Context(0, wait_s=120)
    for lone_tail in (False, True):
        w = pc.ms_mover_workload(lone_tail)
        pc.check_workload(oracle, GpuBackend(ctx, cx.PATH_LANES | cx.LANES_FUSED), w, verify_crc=True)
        assert pc.check_pcm16(oracle, GpuBackend(ctx, cx.OUT_PCM16), w) == w.n
        assert pc.check_pcm16(oracle, GpuBackend(ctx, cx.OUT_PCM16), w, damage=0.2, seed=5) < w.n
    pc.check_crc_in_batch(oracle, GpuBackend(ctx, cx.PATH_LANES | cx.LANES_FUSED), w, seed=12)
    w = pc.ms_mover24_workload()                                                      # (the split tier's waves of such pairs: the same movers)
    pc.check_workload(oracle, GpuBackend(ctx, cx.PATH_LANES | cx.LANES_FUSED), w, verify_crc=True)
    pc.check_crc_in_batch(oracle, GpuBackend(ctx, cx.PATH_LANES | cx.LANES_FUSED), w, seed=13)
    ctx.close()
