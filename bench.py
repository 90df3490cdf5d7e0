#!/usr/bin/env python
"""bench.py -- decoded Msamples/s of the batched FLAC frame decode hot path on MI355X.

A "step" = one pass of the hot path (Rice/residual decode, predictor synthesis, stereo decorrelation) over one batch of
synthetic frames whose compressed bytes, descriptors and output buffer are already resident in HBM.

Workloads (`--workload`):
  config3 (default, the configuration BASELINE.json's metric is quoted on): 10 000 stereo 16-bit frames per GPU, block size
          4096, mid/side, both subframes LPC order 8.  At N GPUs the job is ONE frame index of N x 10 000
          frames (frame g is seeded by g), cut into contiguous ranges by claxon_amd.shard.balanced_ranges; rank r generates
          and uploads only its range (weak scaling: the per-GPU share is fixed).
  config2 / config4: the other single-GPU BASELINE shapes (parity-test cases; here for profiling them).
  config5: `--total-frames` (default 1 000 000) mixed real-world-shaped frames, `--unique` (default 16 384) unique frames
          tiled with distinct frame numbers / CRCs, sharded over the ranks by algorithmic bytes (strong scaling: the total is
          fixed).  `--shard-of N --shard-rank r` runs rank r's share of an N-way split on one GPU without a process group.
Frames are independent: ranks share nothing and there is no collective on the data path (barrier + MAX/SUM reductions only).

The timed region -- exactly `--steps` steps between barrier + synchronize on both sides, MAX over the ranks -- is repeated `--repeats`
times (default 5): `value` / `ms_per_step` are the MEDIAN region, `ms_per_step_min` / `_max` and `value_min` / `_max` travel beside
them.  `--devices 0,0` maps local ranks to devices (two ranks on one GPU: the rehearsal of the N-rank line on the hardware that
exists); `--compose on|off` forces the waves' composition by content; `cpu_baseline` is in the line at every N (rank 0).

One JSON line on rank 0; DESIGN.md section 5 says how `roofline` and `cpu_baseline` are derived.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# (No HIP environment variable is set here: the timed steps go out as merged launches (up to twelve steps per grid) on two internal streams of the library,
# which HIP's default number of hardware queues covers.  GPU_MAX_HW_QUEUES, if the caller sets it, is carried in the line.)

import numpy as np  # noqa: E402

PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md)


_REAL_STDOUT = None


def _own_stdout():
    """The contract is ONE JSON line on rank 0's stdout.  Libraries under this process write there too -- RCCL prints a version banner
    of five lines when its first communicator comes up (seen with `--process-group --backend nccl`: profiles/r05_rccl_one_rank.log) -- so
    file descriptor 1 is pointed at stderr for the life of the process and the line goes out through a private copy of the real one."""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)


def _emit(line):
    sys.stdout.flush()
    if _REAL_STDOUT is None:
        print(line, flush=True)
    else:
        os.write(_REAL_STDOUT, (line + "\n").encode())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=96)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--workload", choices=["config2", "config3", "config4", "config5"], default="config3")
    ap.add_argument("--frames", type=int, default=10000, help="frames per GPU for config2/3/4 (BASELINE: 10000)")
    ap.add_argument("--total-frames", type=int, default=1000000, help="config5: frames of the whole job")
    ap.add_argument("--unique", type=int, default=16384, help="config5: unique frames that are tiled")
    ap.add_argument("--shard-of", type=int, default=0, help="config5 on one GPU: pretend to be one rank of this many")
    ap.add_argument("--shard-rank", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pipeline", action="store_true", help="time steps that do not overlap (clx_batch_run instead of clx_batch_submit)")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary figures (unverified step, host-buffer rates)")
    ap.add_argument("--no-crc", action="store_true", help="time the step WITHOUT the CRC-16 check (secondary figure; `value` verifies by default)")
    ap.add_argument("--path", choices=["auto", "waves", "lanes", "lanes-fused", "lanes-general"], default="auto",
                    help="kernel path (default: library's choice); lanes-general = the fused lane build without the 16-bit tier clx_k_lean")
    ap.add_argument("--order", type=int, default=0, help="config4 only: another predictor order than BASELINE's 32 (12: the split tier's <= 12-tap kernel)")
    ap.add_argument("--compose", choices=["auto", "on", "off"], default="auto",
                    help="waves composed by content (clx_k_compose): the library's choice by the descriptors, or forced on / off")
    ap.add_argument("--pool", choices=["on", "off"], default="off",
                    help="on: merged launches take the scan and the 16-bit tier as clx_k_pool's tickets (CLX_POOL: round 6's other launch form, measured slower)")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl", help="process group backend for the barrier and the MAX / SUM reductions (nccl = RCCL)")
    ap.add_argument("--repeats", type=int, default=5,
                    help="how often the timed region of --steps steps is repeated: `value` is the median region, min / max are carried beside it")
    ap.add_argument("--devices", default="",
                    help="comma-separated device index per local rank (default: rank r on device r); `--gpus 2 --devices 0,0 --backend gloo` "
                         "rehearses the N-rank path -- spawn, process group, barrier, reductions, one line -- on ONE GPU")
    ap.add_argument("--process-group", action="store_true",
                    help="create the process group even when WORLD_SIZE is 1: the barrier and the MAX / SUM / all_gather reductions then go "
                         "through the backend (RCCL on device tensors with --backend nccl) on the one GPU that is there")
    ap.add_argument("--spaced-figure", action="store_true",
                    help="internal: print the secondary figure of the same steps with the frames' output blocks spaced apart (config.spaced_output_blocks) and nothing else")
    ap.add_argument("--pcm16-figure", action="store_true",
                    help="internal: only the CLX_OUT_PCM16 secondary figure of this workload, as one JSON object (the default line runs this in a "
                         "process of its own: the figure depends on which hardware queues the batch's streams get, i.e. on what ran before it)")
    ap.add_argument("--launcher-selftest", action="store_true",
                    help="no decode: only the launcher, the rank plan and the cross-rank reductions (CPU, gloo); prints a line with value null")
    args = ap.parse_args()

    # `--gpus N` by itself starts the N ranks (one process per GPU); under an external launcher (torchrun: WORLD_SIZE is set)
    # this process is one of its ranks already
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.shard_of:
        sys.exit(_spawn_ranks(args.gpus))
    _own_stdout()
    if args.launcher_selftest:
        return _launcher_selftest(args)

    import torch
    import claxon_amd as cx
    import synth
    from claxon_amd import shard
    dist = None

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dev_index = _device_of(args.devices, local_rank)
    use_pg = world > 1 or args.process_group       # collectives go through the process group (always with several ranks)
    if use_pg:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        torch.cuda.set_device(dev_index)
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
    ctx = cx.Context(dev_index, wait_s=120)   # waits for the device to appear; raises if there is none
    dev = torch.device("cuda", dev_index)
    torch.cuda.set_device(dev)
    # (gloo reduces host tensors; RCCL device tensors)
    red_dev = dev if (not use_pg or args.backend == "nccl") else None

    # ---- this rank's share of the job's frame index
    t_gen = time.time()
    sh_world, sh_rank = (args.shard_of, args.shard_rank) if (args.shard_of and world == 1) else (world, rank)
    w, ts, shard_info, workload_name, scaling = _rank_share(args, synth, shard, sh_world, sh_rank)
    gen_s = time.time() - t_gen
    if w.bare_subframes:
        descs = cx.descs_for_subframes(w.offs, w.block_sizes, w.bps)
    else:
        descs, _ = cx.descs_from_offsets(w.arena[:w.arena_len], w.offs, w.lens)

    d_arena = torch.from_numpy(w.arena).to(dev)
    d_out = torch.zeros(w.total_samples, dtype=torch.int32, device=dev)
    path = {"auto": 0, "waves": cx.PATH_WAVES, "lanes": cx.PATH_LANES, "lanes-fused": cx.PATH_LANES | cx.LANES_FUSED,
            "lanes-general": cx.PATH_LANES | cx.LANES_FUSED | cx.LANES_GENERAL}[args.path]
    path |= {"auto": 0, "on": cx.COMPOSE, "off": cx.NO_COMPOSE}[args.compose]
    path |= cx.POOL if args.pool == "on" else 0
    # `value` is the VERIFIED step: every frame's CRC-16 footer is checked on the device inside it, as the reference does for every
    # frame it decodes (frame.rs:752-763) and as the cpu_baseline leg does; bare subframes (config 2) have no footer
    with_crc = (not w.bare_subframes) and not args.no_crc
    if args.pcm16_figure:
        fig = _pcm16_from_the_decode(torch, ctx, cx, w, descs, d_arena, dev, args.steps, args.repeats, with_crc, path)
        _emit(json.dumps(fig))
        return
    if args.spaced_figure:
        fig = _spaced_output_blocks(torch, ctx, cx, w, descs, d_arena, dev, args.steps, args.repeats, with_crc, path)
        _emit(json.dumps(fig))
        return
    batch = ctx.plan(descs, w.out_offs, verify_crc=with_crc, path=path)
    # consecutive steps are submitted with up to batch.submit_depth of them in flight (clx_batch_submit: each step a whole run on
    # an internal stream of the library), so they rotate over that many output buffers -- when there is room for them
    depth = batch.submit_depth
    pipelined = (not args.no_pipeline) and depth > 1 and depth * 4 * w.total_samples < 128 * (1 << 30)
    outs = [d_out] + ([torch.zeros(w.total_samples, dtype=torch.int32, device=dev) for _ in range(depth - 1)] if pipelined else [])
    # the steps in flight read DISTINCT copies of the compressed input (same bytes, different addresses), so that no step finds the
    # input of its neighbour in a cache: every step's compressed bytes come from HBM, as roofline.achieved assumes
    arenas = [d_arena] + ([d_arena.clone() for _ in range(depth - 1)] if pipelined and depth * w.arena.size < 8 * (1 << 30) else [])
    stream = torch.cuda.current_stream(dev).cuda_stream

    barrier_s = []                                  # what the closing barrier of each timed region took on this rank (outside the clock)

    def barrier():
        if use_pg:
            dist.barrier()

    def timed(b, steps, pipe):
        """`steps` passes of batch b, bracketed by barrier + synchronize on both sides; seconds (this rank).  The clock stops when
        this rank's work is done (after synchronize) and BEFORE the closing barrier: the job's time is the MAX over the ranks of
        these local times, so a collective inside a 3-ms region would only add its own latency to every rank's figure."""
        barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        if pipe:
            for i in range(steps):
                b.submit(arenas[i % len(arenas)].data_ptr(), w.arena_len, outs[i % len(outs)].data_ptr(), stream)
            b.flush(stream)
        else:
            for _ in range(steps):
                b.run(d_arena.data_ptr(), w.arena_len, d_out.data_ptr(), stream)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        barrier()
        barrier_s.append(time.perf_counter() - t1)
        return t1 - t0

    for i in range(max(args.warmup, len(outs) if pipelined else 0)):
        if pipelined:
            batch.submit(arenas[i % len(arenas)].data_ptr(), w.arena_len, outs[i % len(outs)].data_ptr(), stream)
        else:
            batch.run(d_arena.data_ptr(), w.arena_len, d_out.data_ptr(), stream)
    batch.flush(stream)
    torch.cuda.synchronize()

    # ---- parity gate before anything is timed: statuses OK and bit-exact vs the source PCM (every output buffer in use)
    def outputs_exact(bufs):
        if w.pcm is not None:
            ref_pcm = torch.from_numpy(w.pcm).to(dev)
            return all(bool(torch.equal(o, ref_pcm)) for o in bufs)
        return all(_tiled_equal(torch, o, w, ts.unique, dev) for o in bufs)

    res = batch.results()
    if not (bool(np.all(res["status"] == 0)) and outputs_exact(outs)):
        raise SystemExit("bench: decode is not bit-exact; refusing to report a number")

    # ---- per-kernel durations: HIP events recorded by the library on the launch stream, around each of its kernels (steps one at
    #      a time); they also say which kernels the library selected -- only the wave path with the latency build of the predictor keeps several steps in flight
    kernel_ms = _kernel_ms(torch, batch, lambda: batch.run(d_arena.data_ptr(), w.arena_len, d_out.data_ptr(), stream), args.steps)
    kernel_ms_run = dict(kernel_ms)            # what clx_batch_run selects (config.one_step_at_a_time)
    path_tag = ""
    if pipelined and batch.submit_lanes and "clx_k_lanes" not in kernel_ms:
        # the pipelined steps run the fused lane kernels while one run at a time takes the wave kernels: the roofline block is
        # about the kernels of the TIMED steps, so their durations are taken from a batch forced onto them
        bl = ctx.plan(descs, w.out_offs, verify_crc=with_crc, path=cx.PATH_LANES | cx.LANES_FUSED | (path & (cx.COMPOSE | cx.NO_COMPOSE | cx.POOL)))
        kernel_ms = _kernel_ms(torch, bl, lambda: bl.run(d_arena.data_ptr(), w.arena_len, d_out.data_ptr(), stream), args.steps)
        bl.close()
        path_tag = "_lanes"

    # ---- the launch shape of the TIMED steps, one launch at a time: the library merges consecutive submissions into one grid
    #      (batch.submit_depth / 2 runs per launch); its kernels' durations with nothing beside them, HIP events on the internal
    #      stream they are launched on (clx_batch_set_profiling(b, 2)) -- what `rocprofv3 --kernel-trace --stats` of
    #      tools/merge_probe.py shows (profiles/r03_config3_merged12_kernel_stats.csv)
    merged = None
    if pipelined and batch.submit_lanes:
        n_merge = max(1, batch.submit_merge)
        batch.set_profiling(2)
        acc = {}
        reps = 3
        for rep in range(reps + 1):
            for i in range(n_merge):
                batch.submit(arenas[i % len(arenas)].data_ptr(), w.arena_len, outs[i % len(outs)].data_ptr(), stream)
            batch.flush(stream)
            torch.cuda.synchronize()
            if rep:
                for name, ms in batch.kernel_times().items():
                    acc[name] = acc.get(name, 0.0) + ms / reps
        batch.set_profiling(False)
        if acc:
            tot = float(sum(acc.values()))
            merged = {"runs_per_launch": n_merge, "kernel_ms": {k: round(v, 4) for k, v in acc.items()}, "ms_per_run": round(tot / n_merge, 4),
                      "achieved": round(w.algorithmic_bytes * n_merge / (tot * 1e-3) / 1e9, 1),
                      "frac": round(w.algorithmic_bytes * n_merge / (tot * 1e-3) / 1e9 / PEAK_GBS, 4),
                      "note": "ONE merged launch at a time (its kernels one after the other, nothing beside them); the timed steps keep two such launches in flight on two streams"}

    # the timed steps write into buffers that were cleared after the gate: what is compared afterwards is what THEY wrote
    for o in outs:
        o.zero_()
    torch.cuda.synchronize()
    # The timed region -- exactly --steps steps between barrier + synchronize on both sides, MAX over the ranks -- is repeated
    # --repeats times (a median and a minimum, not one sample): `value` is the MEDIAN region.
    regions, regions_local = [], []
    samples_all = w.total_samples
    for _ in range(max(1, args.repeats)):
        el_local = timed(batch, args.steps, pipelined)
        # whole-job figures: MAX elapsed over ranks, SUM of samples per step
        el, samples_all, _ = shard.reduce_job(dist if use_pg else None, el_local, w.total_samples, 0, device=red_dev)
        regions.append(el); regions_local.append(el_local)
    res = batch.results()
    if not outputs_exact(outs[:min(len(outs), args.steps)]):
        raise SystemExit("bench: a timed step did not reproduce the source PCM; refusing to report a number")
    _, _, n_bad = shard.reduce_job(dist if use_pg else None, 0.0, 0, int((res["status"] != 0).sum()), device=red_dev)      # SUM of failed frames (must be 0)
    if n_bad:
        raise SystemExit("bench: %d frames failed to decode in the timed region" % n_bad)
    elapsed = float(np.median(regions))
    elapsed_local = float(np.median(regions_local))
    ms_per_step = 1e3 * elapsed / args.steps
    value = samples_all / (ms_per_step * 1e-3) / 1e6
    # every rank's own step time and share (all_gather of three numbers): how even the ranks were
    per_rank = _gather_floats(dist if use_pg else None, [1e3 * elapsed_local / args.steps, float(w.algorithmic_bytes), float(w.n)], world, device=red_dev)
    if world > 1:      # algorithmic bytes: max over ranks / mean
        algs = [r[1] for r in per_rank]
        shard_info["imbalance"] = round(max(algs) / (sum(algs) / len(algs)) - 1.0, 5)

    path_ms = float(sum(kernel_ms.values()))             # all kernels of the path, one after the other (SURVEY section 8d's t_kernel)
    dom_name = max(kernel_ms, key=kernel_ms.get)
    alg_bytes = w.algorithmic_bytes                      # compressed bytes read once + 4 B per decoded sample written once
    # one step's share of the timed region is what its kernels cost the machine once they overlap the neighbouring steps';
    # never less than max(kernel), never more than their sum when steps do not overlap
    t_path_ms = ms_per_step if pipelined else path_ms
    achieved = alg_bytes / (t_path_ms * 1e-3) / 1e9
    # (which committed profile the counters come from: the fused lane kernels when the timed steps ran them -- pipelined
    #  submissions of these workloads do --, else the kernels clx_batch_run selected)
    timed_lanes = "clx_k_lean" in kernel_ms or "clx_k_lanes" in kernel_ms
    traffic, traffic_src, prof = _pmc_traffic(args.workload + ("_lanes" if timed_lanes else ""), w.n)
    roofline = {"bound": "hbm", "kernel": "+".join(kernel_ms.keys()), "achieved": round(achieved, 1), "peak": PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                "path_ms": round(t_path_ms, 4), "path_ms_basis": ("ms_per_step of the pipelined steps (kernels of consecutive steps overlap)" if pipelined
                                                                  else "sum of the path's kernel durations"),
                "kernel_ms_sum_unpipelined": round(path_ms, 4), "kernel_ms": {k: round(v, 4) for k, v in kernel_ms.items()},
                "kernel_ms_one_step_at_a_time": {k: round(v, 4) for k, v in kernel_ms_run.items()},
                "algorithmic_bytes_per_launch": alg_bytes,
                "dominant_kernel": {"name": dom_name, "ms": round(kernel_ms[dom_name], 4),
                                    "note": "one of the path's kernels; the path's bytes over its time alone would overstate it"},
                "step_achieved": round(alg_bytes / (ms_per_step * 1e-3) / 1e9, 1),
                "step_frac": round(alg_bytes / (ms_per_step * 1e-3) / 1e9 / PEAK_GBS, 4)}
    # every rank's own fraction (its algorithmic bytes over its own step time): the line's `frac` is rank 0's share over the job's time
    fr = [r[1] / (r[0] * 1e-3) / 1e9 / PEAK_GBS for r in per_rank if r[0] > 0]
    if fr:
        roofline["per_rank_frac"] = {"min": round(min(fr), 4), "max": round(max(fr), 4)}
    if merged:
        roofline["merged_launch"] = merged
    if prof and prof.get("insts"):
        # How busy the vector pipes are (round 5's measurement, profiles/r05_ubench_coissue.txt): a SIMD of gfx950
        # takes one vector instruction per ~4.35 cycles in code like this whatever the instruction is (the 2.3-cycle rate of
        # mov / add / shift only shows when nothing else is mixed in), and scalar, LDS and branch instructions of OTHER waves issue
        # beside it (+0.5 .. 2 cycles per scalar instruction, ~0 per LDS instruction) -- they are NOT added on top, as rounds 3-4's
        # flat sum (4.2 / 2.2 / 8.0 cycles) did.  So the bound is the vector instructions alone; what is left of the step is the
        # pipes waiting: for a wave that is ready (two or three waves per SIMD), for LDS / memory round trips, for the fill and
        # drain of the rotation.  Counts: wave-instructions per launch from the same committed --pmc profile as `traffic`.
        ins = prof["insts"].values()
        valu, salu, lds = (sum(k[c] for k in ins) for c in ("valu", "salu", "lds"))
        smp = float(prof.get("samples_per_launch") or w.total_samples)
        VALU_CYCLES, CLOCK_MHZ = 4.35, 2280.0                                # (measured: 16 slow-class VALU at 8 waves per SIMD; shader clock)
        cyc = VALU_CYCLES * valu / 1024.0                                    # per SIMD (256 CUs x 4)
        busy_ms = cyc / (CLOCK_MHZ * 1e3)
        scale = w.total_samples / smp
        roofline["issue"] = {"bound": "vector pipe (VALU issue)", "valu_per_sample": round(64.0 * valu / smp, 1), "salu_per_sample": round(64.0 * salu / smp, 1),
                             "lds_per_sample": round(64.0 * lds / smp, 2), "valu_cycles_each": VALU_CYCLES, "clock_mhz": CLOCK_MHZ,
                             "simd_cycles_per_launch": int(cyc * scale),
                             "valu_busy_ms": round(busy_ms * scale, 4), "valu_busy_frac": round(busy_ms * scale / t_path_ms, 4),
                             "waiting_frac": round(1.0 - busy_ms * scale / t_path_ms, 4),
                             "note": "instruction counts: committed rocprofv3 --pmc profile (traffic_source); valu_busy_frac = the vector pipes' "
                                     "busy time / time of the step; scalar and LDS instructions co-issue (profiles/r05_ubench_coissue.txt, r06_ubench_coissue.txt) and are not added; in a full machine the launches also move ~4.5 TB/s of HBM traffic against a copy ceiling of ~5 (profiles/r06_final_figures.txt): the step is bound by both"}

    extras = rank == 0 and not args.no_extras
    if extras:
        # achievable-copy ceiling of this box: device-to-device copy of 1 GiB, read + write bytes
        src = torch.empty(1 << 28, dtype=torch.int32, device=dev); dst = torch.empty_like(src)
        dst.copy_(src); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            dst.copy_(src)
        e1.record(); torch.cuda.synchronize()
        copy_gbs = 5 * 2 * src.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9
        roofline["copy_ceiling"] = round(copy_gbs, 1)
        roofline["frac_of_copy_ceiling"] = round(achieved / copy_gbs, 4)
        del src, dst

    cfg = {"workload": workload_name, "frames_this_rank": w.n, "samples_per_step": samples_all,
           "compressed_bytes_this_rank": w.compressed_bytes, "bits_per_sample": round(8.0 * w.compressed_bytes / w.total_samples, 3),
           "parallelism": "one frame index sharded over %d GPU(s), no collective on the data path" % world, "shard": shard_info,
           "per_rank": [{"rank": i, "ms_per_step": round(r[0], 4), "frames": int(r[2])} for i, r in enumerate(per_rank)],
           "process_group": {"backend": (args.backend + (" (RCCL)" if args.backend == "nccl" else "")) if use_pg else None, "world_size": world,
                             "barrier_us": round(1e6 * float(np.median(barrier_s)), 1) if (use_pg and barrier_s) else None,
                             "barrier_note": "the closing barrier of a timed region, outside the clock (rank 0's median)"},
           "launcher": os.environ.get("CLX_BENCH_LAUNCHER", "external (WORLD_SIZE in the environment)" if world > 1 else "single process"),
           "bit_exact": True, "bit_exact_checked": "every output buffer vs the source PCM before the timed steps, and again -- on buffers cleared in between -- after them",
           "crc16_in_step": bool(with_crc), "kernel_path": args.path, "compose": args.compose, "pool": args.pool, "gen_seconds": round(gen_s, 1),
           "steps_in_flight": depth if pipelined else 1, "distinct_input_copies_in_flight": len(arenas),
           "merged_launches_per_region": _launch_sizes(args.steps, batch.submit_merge) if (pipelined and batch.submit_lanes) else None,
           "devices": args.devices or None,
           "value_basis": ("throughput of consecutive steps (one 10 000-frame batch each), up to %d in flight on the library's internal streams; "
                           "config.one_step_at_a_time is a single batch's latency" % depth) if pipelined else "one step at a time",
           "hip_env": {"GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES")}}
    out = {
        "metric": "decoded Msamples/s (whole node), 4096-sample stereo 16-bit frames",
        "value": round(value, 1), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
        "repeats": len(regions), "timed_region_s": round(elapsed, 6),
        "ms_per_step_min": round(1e3 * min(regions) / args.steps, 4), "ms_per_step_max": round(1e3 * max(regions) / args.steps, 4),
        "value_min": round(samples_all / (max(regions) / args.steps) / 1e6, 1), "value_max": round(samples_all / (min(regions) / args.steps) / 1e6, 1),
        "value_basis": "median of `repeats` timed regions of `steps` steps each (every region: barrier + synchronize on both sides, MAX over ranks)",
        "dtype": "i32 (i64 LPC accumulate)", "data": "synthetic", "config": cfg, "roofline": roofline,
    }

    if pipelined and not args.no_extras:
        # ---- the same steps one at a time (clx_batch_run: nothing of step i+1 starts before step i has finished)
        el_1 = timed(batch, args.steps, False)
        el_1, samples_1, _ = shard.reduce_job(dist if use_pg else None, el_1, w.total_samples, 0, device=red_dev)
        ms_1 = 1e3 * el_1 / args.steps
        cfg["one_step_at_a_time"] = {"value": round(samples_1 / (ms_1 * 1e-3) / 1e6, 1), "unit": "Msamples/s", "ms_per_step": round(ms_1, 4),
                                     "frac": round(alg_bytes / (ms_1 * 1e-3) / 1e9 / PEAK_GBS, 4),
                                     "note": "clx_batch_run (the kernels it selects: roofline.kernel_ms_one_step_at_a_time): a batch's latency; `value` is the throughput of consecutive batches with up to %d in flight" % depth}
    batch.close()                              # (its internal streams give their hardware queues back)
    if with_crc and not args.no_extras:
        # ---- the same step WITHOUT the CRC-16 check (what earlier rounds reported as `value`): secondary
        bc = ctx.plan(descs, w.out_offs, verify_crc=False, path=path)
        bc.run(d_arena.data_ptr(), w.arena_len, d_out.data_ptr(), stream)
        if pipelined:                          # (the first submissions set the internal streams and buffers up)
            for i in range(len(outs)):
                bc.submit(arenas[i % len(arenas)].data_ptr(), w.arena_len, outs[i].data_ptr(), stream)
        torch.cuda.synchronize()
        rc = bc.results()
        el_c = timed(bc, args.steps, pipelined)
        el_c, samples_c, bad_c = shard.reduce_job(dist if use_pg else None, el_c, w.total_samples, int((rc["status"] != 0).sum()), device=red_dev)
        bc.close()
        if bad_c == 0:
            ms_c = 1e3 * el_c / args.steps
            cfg["without_crc16"] = {"value": round(samples_c / (ms_c * 1e-3) / 1e6, 1), "unit": "Msamples/s", "ms_per_step": round(ms_c, 4),
                                    "note": "the same steps with the CRC-16 footer check left out (BENCH_r01 / r02's `value`); `value` verifies"}
    if extras and world == 1 and args.workload == "config3" and w.pcm is not None and path_tag:
        cfg["wave_kernels_pipelined"] = _wave_kernels_pipelined(torch, ctx, cx, w, descs, d_arena, dev, args.steps)
    if extras and world == 1 and not w.bare_subframes and w.pcm is not None:
        cfg["host_buffers"] = _host_buffer_rates(ctx, cx, w, descs)
    if extras and world == 1 and pipelined and not w.bare_subframes and w.pcm is not None and int(np.max(w.bps)) <= 24:
        # ---- narrow output straight from the decode (CLX_OUT_PCM16; config 4: CLX_OUT_PCM24), in a process of its own: measured in this one the figure depended
        #      on its place among the secondary figures (in front of the host-buffer block: 0.117 ms per step and that block's upload
        #      figure 1.61 instead of 1.40 ms; behind it: 0.132 and 1.40 -- the internal streams a batch creates get their hardware queues
        #      by what was created before them, tools/stream_probe.py).  A fresh process is the state `value` is measured in.
        fig = _pcm16_subprocess(args)
        if fig is not None:
            cfg["pcm16_from_the_decode" if fig.get("sample_bytes", 2) == 2 else "pcm24_from_the_decode"] = fig
        if w.pcm is not None and len(set((w.channels.astype(np.int64) * w.block_sizes.astype(np.int64)).tolist())) == 1:
            fig = _pcm16_subprocess(args, "--spaced-figure")
            if fig is not None:
                cfg["spaced_output_blocks"] = fig
    if rank == 0 and not args.no_cpu_baseline:
        # (at N > 1 too, on rank 0's share, behind the timed regions: north_star wants the CPU path timed in the same run at every N;
        #  the other ranks wait at the end)
        out["cpu_baseline"] = _cpu_baseline(w)
    batch.close()
    if use_pg:
        dist.destroy_process_group()
    if rank == 0:
        _emit(json.dumps(out))


def _narrow_want(w, sample_bytes, dev, torch):
    """(expected bytes of a narrow output buffer as a uint8 tensor, mask of the bytes some frame covers): frame by frame, whatever the
    channel counts and block sizes -- interleaved, little-endian, the low `sample_bytes` bytes of every sample."""
    inter = np.zeros(w.pcm.size, dtype=np.int32)
    covered = np.zeros(w.pcm.size, dtype=bool)
    for i in range(w.n):
        a, c, bs = int(w.out_offs[i]), int(w.channels[i]), int(w.block_sizes[i])
        inter[a:a + c * bs] = w.pcm[a:a + c * bs].reshape(c, bs).T.reshape(-1)
        covered[a:a + c * bs] = True
    u = inter.view(np.uint32)
    by = np.stack([(u >> (8 * k)) & 0xff for k in range(sample_bytes)], axis=1).astype(np.uint8).reshape(-1)
    return torch.from_numpy(by).to(dev), torch.from_numpy(np.repeat(covered, sample_bytes)).to(dev)


def _pcm16_from_the_decode(torch, ctx, cx, w, descs, d_arena, dev, steps, repeats, with_crc, path):
    """Narrow output straight from the decode, consecutive steps like `value`'s: CLX_OUT_PCM16 (interleaved 16-bit PCM written by the
    lean kernel from the tiles it stages anyway -- half the bytes through the write path) for audio of <= 16 bits, CLX_OUT_PCM24 (packed
    24-bit, the split tier's: round 6) beyond.  A secondary figure, never `value`: Claxon's Block is planar i32 (frame.rs:402-411);
    this is what examples/decode.rs:48-62 and lib.rs:473-520 do with it right afterwards.  Not bit-exact, no figure."""
    stream = torch.cuda.current_stream(dev).cuda_stream
    sb = 2 if int(descs["bps"].max()) <= 16 else 3
    flag = cx.OUT_PCM16 if sb == 2 else cx.OUT_PCM24
    bp = ctx.plan(descs, w.out_offs, verify_crc=with_crc, path=(path & (cx.COMPOSE | cx.NO_COMPOSE | cx.POOL)) | flag)
    depth = bp.submit_depth
    arenas = [d_arena] + [d_arena.clone() for _ in range(depth - 1)]          # (distinct copies of the input, like `value`'s steps)
    pouts = [torch.zeros(sb * w.total_samples + 16, dtype=torch.uint8, device=dev) for _ in range(depth)]
    for i in range(len(pouts)):
        bp.submit(arenas[i % len(arenas)].data_ptr(), w.arena_len, pouts[i].data_ptr(), stream)
    bp.flush(stream); torch.cuda.synchronize()
    want, cov = _narrow_want(w, sb, dev, torch)
    n_by = sb * w.pcm.size
    exact = all(bool(torch.equal(o[:n_by][cov], want[cov])) for o in pouts) and bool(np.all(bp.results()["status"] == 0))
    if not exact:
        bp.close()
        raise RuntimeError("bench: the narrow output (%d bytes per sample) is not bit-exact" % sb)
    for o in pouts:
        o.zero_()
    regs = []
    for _ in range(max(1, repeats)):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(steps):
            bp.submit(arenas[i % len(arenas)].data_ptr(), w.arena_len, pouts[i % len(pouts)].data_ptr(), stream)
        bp.flush(stream); torch.cuda.synchronize(); regs.append(time.perf_counter() - t0)
    exact = all(bool(torch.equal(o[:n_by][cov], want[cov])) for o in pouts[:min(steps, len(pouts))])      # (what the TIMED steps wrote)
    bp.close(); del pouts
    if not exact:
        raise RuntimeError("bench: the narrow output of the timed steps is not bit-exact")
    ms_p = 1e3 * float(np.median(regs)) / steps
    alg = w.compressed_bytes + sb * w.total_samples
    return {"value": round(w.total_samples / (ms_p * 1e-3) / 1e6, 1), "unit": "Msamples/s", "ms_per_step": round(ms_p, 4), "steps": steps,
            "sample_bytes": sb, "bit_exact": True, "algorithmic_bytes": alg, "achieved_GBps": round(alg / (ms_p * 1e-3) / 1e9, 1),
            "frac": round(alg / (ms_p * 1e-3) / 1e9 / PEAK_GBS, 4),
            "note": ("CLX_OUT_PCM16: interleaved little-endian 16-bit PCM" if sb == 2 else "CLX_OUT_PCM24: interleaved packed little-endian 24-bit PCM") +
                    " written by the decode kernel itself (%d bytes per sample out instead of 4), checked frame by frame before and after the timed "
                    "steps, measured in a process of its own; secondary -- `value` is planar i32, Claxon's Block" % sb}


def _spaced_output_blocks(torch, ctx, cx, w, descs, d_arena, dev, steps, repeats, with_crc, path, pad=96):
    """The same consecutive steps with every frame's output block `pad` samples further from its neighbour than it has to be (the caller's choice:
    out_sample_offsets).  `value`'s layout puts the frames' blocks one behind the other, so every row of a store instruction of the decode waves starts
    a multiple of 16 KiB from the next -- a stride that does not spread over the memory channels; 384 bytes between the blocks do (tools/layout_probe.py:
    0, 128, 256, 384, 640, 4 224 bytes measured).  A secondary figure, never `value`: what INTEGRATION.md tells a binder about laying a batch out."""
    stream = torch.cuda.current_stream(dev).cuda_stream
    size = int(w.channels[0]) * int(w.block_sizes[0])                       # (the caller made sure every frame has this many samples)
    offs = np.asarray(w.out_offs, dtype=np.uint64) + np.arange(w.n, dtype=np.uint64) * np.uint64(pad)
    assert np.array_equal(np.asarray(w.out_offs, dtype=np.uint64), np.arange(w.n, dtype=np.uint64) * np.uint64(size))
    total = w.n * (size + pad)
    bp = ctx.plan(descs, offs, verify_crc=with_crc, path=path & (cx.COMPOSE | cx.NO_COMPOSE | cx.POOL))
    depth = bp.submit_depth
    arenas = [d_arena] + [d_arena.clone() for _ in range(depth - 1)]          # (distinct copies of the input, like `value`'s steps)
    outs = [torch.zeros(total, dtype=torch.int32, device=dev) for _ in range(depth)]
    ref = torch.from_numpy(w.pcm).to(dev).view(w.n, size)
    exact_all = lambda bufs: all(bool(torch.equal(torch.as_strided(o, (w.n, size), (size + pad, 1)), ref)) for o in bufs)
    for i in range(depth):
        bp.submit(arenas[i % depth].data_ptr(), w.arena_len, outs[i].data_ptr(), stream)
    bp.flush(stream); torch.cuda.synchronize()
    if not (exact_all(outs) and bool(np.all(bp.results()["status"] == 0))):
        bp.close()
        raise RuntimeError("bench: the spaced output layout is not bit-exact")
    for o in outs:
        o.zero_()
    regs = []
    for _ in range(max(1, repeats)):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(steps):
            bp.submit(arenas[i % depth].data_ptr(), w.arena_len, outs[i % depth].data_ptr(), stream)
        bp.flush(stream); torch.cuda.synchronize(); regs.append(time.perf_counter() - t0)
    exact = exact_all(outs[:min(steps, depth)])                              # (what the TIMED steps wrote)
    bp.close(); del outs
    if not exact:
        raise RuntimeError("bench: the spaced output layout of the timed steps is not bit-exact")
    ms_p = 1e3 * float(np.median(regs)) / steps
    alg = w.compressed_bytes + 4 * w.total_samples
    return {"value": round(w.total_samples / (ms_p * 1e-3) / 1e6, 1), "unit": "Msamples/s", "ms_per_step": round(ms_p, 4), "steps": steps,
            "pad_samples": pad, "bit_exact": True, "frac": round(alg / (ms_p * 1e-3) / 1e9 / PEAK_GBS, 4),
            "note": "the same steps, every frame's output block %d bytes further from its neighbour (out_sample_offsets: the caller's choice) -- `value`'s frames lie one "
                    "behind the other, 16 KiB from row to row, which does not spread over the memory channels; checked frame by frame before and after the timed "
                    "steps, measured in a process of its own; secondary, never `value`" % (4 * pad)}


def _pcm16_subprocess(args, which="--pcm16-figure"):
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), which, "--steps", str(args.steps), "--repeats", str(args.repeats),
           "--workload", args.workload, "--frames", str(args.frames), "--total-frames", str(args.total_frames), "--unique", str(args.unique),
           "--shard-of", str(args.shard_of), "--shard-rank", str(args.shard_rank), "--compose", args.compose, "--devices", args.devices]
    if args.no_crc:
        cmd.append("--no-crc")
    cmd += ["--pool", args.pool]
    r = None
    try:
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        return json.loads(r.stdout.decode().strip().splitlines()[-1])
    except Exception as e:                       # (a secondary figure: the line goes out without it)
        print("bench: the narrow-output figure's process failed (no figure): %r %s" % (e, r.stderr.decode()[-300:] if r is not None else ""), file=sys.stderr)
        return None


def _device_of(devices, local_rank):
    """--devices "0,0": local rank r runs on device devices[r]; default: device r."""
    if not devices:
        return local_rank
    ds = [int(x) for x in devices.split(",") if x.strip() != ""]
    return ds[local_rank % len(ds)]


def _launch_sizes(steps, merge):
    """How the steps of one timed region go out: the library merges consecutive submissions, `merge` per grid (two streams in turn)."""
    m = max(1, merge)
    return [m] * (steps // m) + ([steps % m] if steps % m else [])


def _rank_share(args, synth, shard, sh_world, sh_rank, generate=True):
    """This rank's share of the job's ONE frame index: (workload or None, tiled stream or None, shard_info, workload name,
    scaling).  `generate=False` (the launcher self-test) plans without encoding any frame of config2/3/4."""
    shard_info = {"ranks": sh_world, "rank": sh_rank}
    ts = None
    if args.workload == "config5":
        ts = synth.config5_tiled(args.total_frames, args.unique)
        ranges = shard.balanced_ranges(ts.weights(), sh_world)
        lo, hi = ranges[sh_rank]
        w = ts.slice(lo, hi) if generate else None
        wsum = [int(ts.weights()[a:b].sum()) for a, b in ranges]
        shard_info.update({"plan": "shard.balanced_ranges over algorithmic bytes of %d frames (%d unique)" % (ts.total, ts.unique.n),
                           "range": [int(lo), int(hi)], "imbalance": round(max(wsum) / (sum(wsum) / len(wsum)) - 1.0, 5)})
        workload_name = ("BASELINE configs[4]: %d mixed real-world-shaped stereo 16-bit frames (orders 0-12, all channel modes, "
                         "%d unique frames tiled with distinct frame numbers / CRCs), rank share %d frames"
                         % (ts.total, ts.unique.n, hi - lo))
        scaling = "strong"
    else:
        lo, hi = sh_rank * args.frames, (sh_rank + 1) * args.frames
        gen = {"config2": synth.config2, "config3": synth.config3, "config4": synth.config4}[args.workload]
        if args.workload == "config4" and args.order:
            gen = lambda n, _o=args.order: synth.config4(n, order=_o)
        w = _seeded(synth, gen, args.frames, lo) if generate else None
        # the ranges are what balanced_ranges gives for the job's index: every frame of these shapes has the same decoded size
        # and (to within a percent) the same compressed size; the measured imbalance is carried in the line
        shard_info.update({"plan": "contiguous ranges of %d frames of one index of %d (frame g seeded by g)" % (args.frames, sh_world * args.frames),
                           "range": [int(lo), int(hi)]})
        workload_name = {
            "config2": "BASELINE configs[1]: %d mono 16-bit subframes/GPU, bs 4096, FIXED order 2, Rice k=4, one partition",
            "config3": "BASELINE configs[2]: %d stereo 16-bit frames/GPU, bs 4096, mid/side, LPC order 8 (coefficient precision 12-14), Rice partition order 4, optimal k",
            "config4": "BASELINE configs[3]: %d stereo 24-bit frames/GPU, bs 4096, LPC order 32, mixed partition orders 0-7, Rice2, wasted bits",
        }[args.workload] % args.frames
        if args.workload == "config4" and args.order:
            workload_name = workload_name.replace("LPC order 32", "LPC order %d (NOT the BASELINE shape: --order)" % args.order)
        scaling = "weak"
    return w, ts, shard_info, workload_name, scaling


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _spawn_ranks(n):
    """`python bench.py --gpus N` with no launcher around it: start one process per GPU (rank r on device r), each a copy of this
    command line with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in its environment -- exactly what
    `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...` would hand them.  Rank 0 prints the JSON line on
    this process's stdout; the other ranks' stdout goes to stderr.  Returns the first non-zero exit code (the others are stopped)."""
    import subprocess
    port = os.environ.get("MASTER_PORT") or str(_free_port())
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                    "MASTER_ADDR": os.environ.get("MASTER_ADDR", "127.0.0.1"), "MASTER_PORT": port,
                    "CLX_BENCH_LAUNCHER": "bench.py --gpus %d (self-spawned)" % n})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else sys.stderr))
    rc = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                code = p.poll()
                if code is None:
                    continue
                pending.remove(p)
                if code != 0 and rc == 0:
                    rc = code
                    for q in pending:           # a failed rank would leave the others waiting at a barrier
                        q.terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def _launcher_selftest(args):
    """The N-rank bookkeeping of this file without any decode (CPU; `--backend gloo`): the launcher's environment, the rank plan
    (`_rank_share`), the barrier, the MAX / SUM reductions and the imbalance figure, with made-up elapsed times (1 + rank ms per
    step) and the planned sample counts.  Prints a line shaped like the bench line with `value` null and `selftest` true."""
    import torch
    import torch.distributed as dist
    import synth
    from claxon_amd import shard
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    w, ts, shard_info, workload_name, scaling = _rank_share(args, synth, shard, world, rank, generate=(args.workload == "config5"))
    lo, hi = shard_info["range"]
    if w is not None:
        samples, alg = w.total_samples, w.algorithmic_bytes
    else:
        ch = 1 if args.workload == "config2" else 2
        samples, alg = (hi - lo) * ch * 4096, (hi - lo) * ch * 4096 * 4
    if world > 1:
        dist.barrier()
    elapsed, samples_all, n_bad = shard.reduce_job(dist if world > 1 else None, 1e-3 * (1 + rank) * args.steps, samples, 0)
    per_rank = _gather_floats(dist if world > 1 else None, [1.0 + rank, float(alg), float(lo), float(hi)], world)
    algs = [r[1] for r in per_rank]
    shard_info["imbalance"] = round(max(algs) / (sum(algs) / len(algs)) - 1.0, 5)
    if rank == 0:
        _emit(json.dumps({"metric": "decoded Msamples/s (whole node), 4096-sample stereo 16-bit frames", "value": None, "selftest": True,
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4),
                          "scaling": scaling, "config": {"workload": workload_name, "samples_per_step": samples_all, "shard": shard_info,
                                                         "per_rank": [{"rank": i, "ms_per_step": r[0], "range": [int(r[2]), int(r[3])]} for i, r in enumerate(per_rank)],
                                                         "process_group": {"backend": "gloo", "world_size": world},
                                                         "devices": args.devices or None,
                                                         "device_of_rank": [_device_of(args.devices, r) for r in range(world)],
                                                         "launcher": os.environ.get("CLX_BENCH_LAUNCHER", "external (WORLD_SIZE in the environment)" if world > 1 else "single process")}}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def _gather_floats(dist, vals, world, device=None):
    """Every rank's list of floats, on every rank: [[rank 0's], [rank 1's], ...] (all_gather of one small tensor)."""
    import torch
    t = torch.tensor(vals, dtype=torch.float64, device=device)
    if dist is None:
        return [t.tolist()]
    out = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    return [o.tolist() for o in out]


def _seeded(synth, gen, n, first):
    """Frames [first, first + n) of the job's index: frame g's PCM comes from seed BASE_SEED + g."""
    base = synth.BASE_SEED
    synth.BASE_SEED = base + first
    try:
        return gen(n)
    finally:
        synth.BASE_SEED = base


def _tiled_equal(torch, d_out, w, unique, dev):
    """d_out == the unique frames' PCM tiled in this rank's order (all frames of config 5 have the same size)."""
    per = int(unique.channels[0]) * int(unique.block_sizes[0])
    exp = torch.from_numpy(unique.pcm.reshape(unique.n, per)).to(dev)
    idx = torch.from_numpy(w.expected_index).to(dev)
    got = d_out.view(-1, per)
    ok = True
    for a in range(0, got.shape[0], 8192):
        ok = ok and bool(torch.equal(got[a:a + 8192], exp[idx[a:a + 8192]]))
    return ok


def _kernel_ms(torch, batch, step, steps):
    batch.set_profiling(True)
    acc = {}
    for _ in range(max(5, min(steps, 20))):
        step()
        torch.cuda.synchronize()
        for name, ms in batch.kernel_times().items():
            acc.setdefault(name, []).append(ms)
    batch.set_profiling(False)
    return {k: float(np.mean(v)) for k, v in acc.items()}


def _wave_kernels_pipelined(torch, ctx, cx, w, descs, d_arena, dev, steps):
    """The same steps forced onto the OTHER kernel family: the wave-per-frame kernels, pipelined with their own depth (four in
    flight).  What small batches of short codes run by default; here for comparison with `value`."""
    try:
        b = ctx.plan(descs, w.out_offs, path=cx.PATH_WAVES)
        depth = b.submit_depth
        outs = [torch.zeros(w.total_samples, dtype=torch.int32, device=dev) for _ in range(depth)]
        stream = torch.cuda.current_stream(dev).cuda_stream
        torch.cuda.synchronize()

        def go(k):
            for i in range(k):
                b.submit(d_arena.data_ptr(), w.arena_len, outs[i % depth].data_ptr(), stream)
            b.flush(stream); torch.cuda.synchronize()
        go(2 * depth)
        ref = torch.from_numpy(w.pcm).to(dev)
        ok = all(bool(torch.equal(o, ref)) for o in outs) and bool(np.all(b.results()["status"] == 0))
        del ref
        t = time.perf_counter(); go(steps); dt = (time.perf_counter() - t) / steps
        b.close()
        if not ok:
            return {"error": "not bit-exact"}
        return {"value": round(w.total_samples / dt / 1e6, 1), "unit": "Msamples/s", "ms_per_step": round(dt * 1e3, 4), "steps_in_flight": depth,
                "kernels": "clx_k_residual + clx_k_predict16 (+ what it leaves, clx_k_predict_1w / _1w_hi / clx_k_predict)",
                "frac": round(w.algorithmic_bytes / dt / 1e9 / PEAK_GBS, 4)}
    except Exception as e:
        return {"error": "%s: %s" % (type(e).__name__, e)}


def _host_buffer_rates(ctx, cx, w, descs):
    """The same batch handed over in HOST memory (SURVEY section 8d "second figure"; never `value`): clx_decode_frames_stream, the
    chunked pipeline upload | decode | download, on pinned buffers (clx_host_alloc) -- with nothing but the results coming back,
    with the PCM coming back as interleaved 16-bit (what a caller of the reference writes to a WAV: examples/decode.rs:48-62), with
    planar i32 coming back -- and the one-shot clx_decode_frames on ordinary memory.  Best of 3 each; link: PCIe Gen5 x16, 63 GB/s."""
    out = {}
    try:
        arena = w.arena[:w.arena_len]
        pin_in = cx.PinnedArray((w.arena_len,), np.uint8)
        pin_in.array[:] = arena
        bad = lambda res: not np.all(res["status"] == 0)

        def best_of(f, reps=3):
            best = None
            for _ in range(reps):
                t = time.perf_counter()
                r = f()
                dt = time.perf_counter() - t
                best = dt if best is None else min(best, dt)
            return best, r

        link = _link_rates()
        out["link_measured_GBps"] = {k: round(v / 1e9, 1) for k, v in link.items()}

        def entry(dt, h2d, d2h, note):
            # the link's own bound for these bytes: each direction at its rate, and both together at the rate they reach when
            # they run at the same time (measured on this box with pinned torch copies: well below twice one direction)
            link_s = max(h2d / link["h2d"], d2h / link["d2h"], (h2d + d2h) / (2.0 * link["both_each"]))
            return {"value": round(w.total_samples / dt / 1e6, 1), "unit": "Msamples/s", "ms": round(dt * 1e3, 3), "h2d_bytes": int(h2d), "d2h_bytes": int(d2h),
                    "link_bound_ms": round(link_s * 1e3, 3), "frac_of_link_bound": round(link_s / dt, 3), "note": note}

        ctx.decode_frames_stream(pin_in.array, descs, w.out_offs, copy_back=False)          # first call: device buffers, plans
        dt, (_, res) = best_of(lambda: ctx.decode_frames_stream(pin_in.array, descs, w.out_offs, copy_back=False))
        if bad(res):
            return {"error": "stream decode failed"}
        out["upload_included_no_pcm_download"] = entry(dt, w.arena_len, 16 * w.n, "pinned input; the PCM stays on the device, only the per-frame results return")
        pin16 = cx.PinnedArray((w.total_samples * 2,), np.uint8)
        dt, (o, res) = best_of(lambda: ctx.decode_frames_stream(pin_in.array, descs, w.out_offs, out=pin16.array, sample_bytes=2))
        want = w.pcm.reshape(w.n, 2, -1).transpose(0, 2, 1).astype("<i2").reshape(-1).view(np.uint8) if (w.channels == 2).all() else None
        if bad(res) or (want is not None and not np.array_equal(o, want)):
            return {"error": "16-bit stream decode is not bit-exact"}
        out["pcm16_download"] = entry(dt, w.arena_len, 2 * w.total_samples, "pinned buffers; interleaved little-endian 16-bit PCM back (narrow stage on the device)")
        pin32 = cx.PinnedArray((w.total_samples,), np.int32)
        dt, (o, res) = best_of(lambda: ctx.decode_frames_stream(pin_in.array, descs, w.out_offs, out=pin32.array))
        if bad(res) or not np.array_equal(o, w.pcm):
            return {"error": "i32 stream decode is not bit-exact"}
        out["i32_download"] = entry(dt, w.arena_len, 4 * w.total_samples, "pinned buffers; planar i32 (Block layout) back")
        host = np.zeros(w.pcm.size, dtype=np.int32)
        dt, (_, res) = best_of(lambda: ctx.decode_frames(arena, descs, w.out_offs, out=host))
        if bad(res) or not np.array_equal(host, w.pcm):
            return {"error": "host-buffer decode is not bit-exact"}
        out["one_shot_pageable"] = entry(dt, w.arena_len, 4 * w.total_samples, "clx_decode_frames: ordinary host memory, device buffers allocated per call, i32 back")
        for p_ in (pin_in, pin16, pin32):
            p_.close()
    except Exception as e:                      # never let a secondary figure take the bench line down
        out["error"] = "%s: %s" % (type(e).__name__, e)
    return out


def _link_rates():
    """Host <-> device copy rates with pinned memory, bytes per second: one direction at a time and both at once (each)."""
    import torch
    n = 128 << 20
    h = torch.empty(n, dtype=torch.uint8, pin_memory=True); d = torch.empty(n, dtype=torch.uint8, device="cuda")
    h2 = torch.empty(n, dtype=torch.uint8, pin_memory=True); d2 = torch.empty(n, dtype=torch.uint8, device="cuda")
    r = {}
    for name, f in (("h2d", lambda: d.copy_(h, non_blocking=True)), ("d2h", lambda: h.copy_(d, non_blocking=True))):
        f(); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(4):
            f()
        torch.cuda.synchronize()
        r[name] = 4 * n / (time.perf_counter() - t)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    best = 0.0
    for rep in range(3):          # (the first round sets the two streams' queues up -- round 5's boxes read 18 GB/s without it, 32 with -- best of the rest)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(4):
            with torch.cuda.stream(s1):
                d.copy_(h, non_blocking=True)
            with torch.cuda.stream(s2):
                h2.copy_(d2, non_blocking=True)
        torch.cuda.synchronize()
        if rep:
            best = max(best, 4 * n / (time.perf_counter() - t))
    r["both_each"] = best
    return r


def _cpu_topology():
    """(all hardware threads, one hardware thread per physical core), from sysfs; falls back to 0..n-1."""
    n = os.cpu_count() or 1
    allowed = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(n))
    firsts = []
    for c in allowed:
        try:
            with open("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list" % c) as f:
                sib = f.read().strip().replace("-", ",").split(",")
            if int(sib[0]) == c:
                firsts.append(c)
        except Exception:
            return allowed, allowed
    return allowed, (firsts or allowed)


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except Exception:
        return None


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def _cpu_baseline(w):
    """The oracle (a C restatement of Claxon's decode path, `kind: port`; real Claxon is Rust and cannot be built here) on
    this box's host cores, decoding the same arena from memory with per-thread recycled output buffers
    (examples/bench_decode.rs:55-78), verifying every CRC as the reference does.  Threads are created, pinned and warmed by an
    untimed pass before the clock starts; the timed region is a number of passes over the frames that lasts >= 0.5 s
    (tools/benchmark.sh:39-41 methodology).  Three thread counts: 1, one per physical core, every hardware thread."""
    import oracle
    arena = w.arena[:w.arena_len]
    # a bounded sample of the step's workload: the first frames of it, enough for every thread to have a few
    n = min(w.n, 10000)
    offs, lens = w.offs[:n], w.lens[:n]
    all_cpus, core_cpus = _cpu_topology()

    def throttled():
        try:
            with open("/sys/fs/cgroup/cpu.stat") as f:
                return {k: int(v) for k, v in (line.split() for line in f)}.get("throttled_usec", 0)
        except Exception:
            return None

    def run(nt, cpus, budget_s):
        s1, t1 = oracle.bench_batch(arena, offs, lens, check_crc=True, nthreads=nt, passes=1, cpus=cpus)
        passes = int(min(max(1, np.ceil(0.6 / max(t1, 1e-6))), max(1, budget_s / max(t1, 1e-6))))
        th0 = throttled()
        s, t = oracle.bench_batch(arena, offs, lens, check_crc=True, nthreads=nt, passes=passes, cpus=cpus)
        th1 = throttled()
        r = {"threads": nt, "pinned": cpus is not None, "value": round(s / t / 1e6, 1), "passes": passes, "timed_s": round(t, 2),
             "first_pass_value": round(s1 / t1 / 1e6, 1)}
        if th0 is not None and th1 is not None:
            r["cgroup_throttled_ms"] = round((th1 - th0) / 1e3, 1)
        return r

    single = run(1, core_cpus[:1], 4.0)
    runs = [single]
    # a container may be allowed fewer CPUs than the box has (cgroup v2 cpu.max = "quota period"): then that many threads is the
    # fair "all cores" case, and more threads only queue for the same quota
    quota = None
    try:
        q, per = (_read("/sys/fs/cgroup/cpu.max") or "max").split()[:2]
        quota = None if q == "max" else max(1, int(np.ceil(int(q) / int(per))))
    except Exception:
        pass
    if quota and 1 < quota < len(core_cpus):
        runs.append(run(quota, None, 5.0))
    if len(core_cpus) > 1:
        runs.append(run(len(core_cpus), core_cpus, 5.0))          # one thread per physical core, pinned
        runs.append(run(len(core_cpus), None, 5.0))               # the same number, placed by the scheduler
    if len(all_cpus) > len(core_cpus):
        runs.append(run(len(all_cpus), None, 5.0))                # every hardware thread
    best = max(runs, key=lambda r: r["value"])
    for r in runs:
        r["parallel_efficiency"] = round(r["value"] / (single["value"] * r["threads"]), 3)
    sample_msamples = float((w.channels[:n].astype(np.int64) * w.block_sizes[:n].astype(np.int64)).sum()) / 1e6
    return {"value": best["value"], "unit": "Msamples/s", "cores": best["threads"], "kind": "port",
            "single_thread": single["value"], "runs": runs,
            "cpu_model": _cpu_model(), "hardware_threads": len(all_cpus), "physical_cores": len(core_cpus), "verifies_crc16": True,
            "cgroup_cpu_max": _read("/sys/fs/cgroup/cpu.max"), "cgroup_cpus": quota,
            "sample": "the first %d frames of the step's workload (%.1f Msamples per pass), `passes` passes per run (>= 0.5 s timed each); "
                      "pooled, pre-warmed threads that take 4 frames at a time off a shared counter (first_pass_value: the single calibration "
                      "pass before it; cgroup_throttled_ms: CPU time the container's quota withheld during the run); includes Claxon's per-byte CRC-16 and "
                      "the CRC-8 / CRC-16 checks, like `value` (config.crc16_in_step)" % (n, sample_msamples)}


def _pmc_traffic(workload, frames):
    """HBM bytes per step from a committed rocprofv3 --pmc summary of this workload at this size (profiles/pmc_traffic.json:
    (2 x FETCH_SIZE + WRITE_SIZE) x 1024 summed over the path's kernels, collected in separate --pmc passes), or null.  The
    counters cannot be read from inside this process; the source (profile directory + commit) travels with the number."""
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(p) as f:
            e = json.load(f).get("%s_frames_%d" % (workload, frames))
        if e:
            # the counters belong to the kernel sources they were taken with: an entry from other sources is refused, not quoted
            have = kernel_source_sha16()
            if e.get("kernel_src_sha16") != have:
                return None, "STALE: %s was taken with kernel sources %s, these are %s (tools/profile_round.sh + tools/update_traffic.py renew it)" % (
                    e.get("source"), e.get("kernel_src_sha16"), have), None
            return e.get("path_bytes"), e.get("source"), e
    except Exception:
        pass
    return None, None, None


def kernel_source_sha16():
    """sha256 over the KERNEL sources (claxon_amd/csrc: clx_kernels / clx_lanes / clx_lean .hip, clx_device.h, clx_crct.h, intrin/*.h --
    not the host layer clx_api.hip / clx_plan.h / host/) with comments and white space taken out, first 16 hex digits."""
    import glob
    import hashlib
    h = hashlib.sha256()
    base = os.path.join(ROOT, "claxon_amd", "csrc")
    names = ["clx_kernels.hip", "clx_lanes.hip", "clx_lean.hip", "clx_device.h", "clx_crct.h"]
    import re
    for f in [os.path.join(base, n) for n in names] + sorted(glob.glob(os.path.join(base, "intrin", "*.h"))):
        h.update(os.path.basename(f).encode())
        with open(f, "r", errors="replace") as fh:
            text = fh.read()
        # (what the compiler sees: comments out, white space collapsed -- a reworded comment does not make a counter profile stale)
        text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
        text = re.sub(r"//[^\n]*", " ", text)
        h.update(" ".join(text.split()).encode())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    main()
