#!/usr/bin/env python
"""bench.py -- decoded Msamples/s of the batched FLAC frame decode hot path on MI355X.

A "step" = one pass of the hot path (K1 Rice/residual decode, K2 predictor + decorrelation) over one
batch of synthetic frames whose compressed bytes, descriptors and output buffer are already resident
in HBM.  Workload at every N: BASELINE.json configs[2] -- 10 000 stereo 16-bit frames, block size
4096, mid/side, both subframes LPC order 8 (SURVEY.md §8d "config 3") -- per GPU (weak scaling:
frames are independent, ranks share nothing, no collective on the data path).

One JSON line on rank 0; see DESIGN.md §6 for how roofline / cpu_baseline are derived.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=10000, help="frames per GPU (BASELINE config: 10000)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--verify-crc", action="store_true", help="also run the CRC-16 kernel inside the step")
    ap.add_argument("--path", choices=["auto", "waves", "lanes"], default="auto", help="kernel path (default: library's choice)")
    args = ap.parse_args()

    import torch
    import claxon_amd as cx
    import synth
    dist = None

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    ctx = cx.Context(local_rank, wait_s=120)   # waits for the device to appear; raises if there is none
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    # ---- synthetic workload for this rank (distinct seeds per rank: frame indices are offset)
    t_gen = time.time()
    w = synth.config3(args.frames) if rank == 0 else _config3_shard(synth, args.frames, rank)
    gen_s = time.time() - t_gen
    descs, _ = cx.descs_from_offsets(w.arena[:w.arena_len], w.offs, w.lens)

    d_arena = torch.from_numpy(w.arena).to(dev)
    d_out = torch.zeros(w.pcm.size, dtype=torch.int32, device=dev)
    path = {"auto": 0, "waves": cx.PATH_WAVES, "lanes": cx.PATH_LANES}[args.path]
    batch = ctx.plan(descs, w.out_offs, verify_crc=args.verify_crc, path=path)
    stream = torch.cuda.current_stream(dev).cuda_stream

    def step():
        batch.run(d_arena.data_ptr(), w.arena_len, d_out.data_ptr(), stream)

    def barrier():
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()

    # ---- parity gate before anything is timed: statuses OK and bit-exact vs the source PCM
    res = batch.results()
    ok = bool(np.all(res["status"] == 0)) and bool(torch.equal(d_out, torch.from_numpy(w.pcm).to(dev)))
    if not ok:
        raise SystemExit("bench: decode is not bit-exact; refusing to report a number")

    barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize(); barrier()
    elapsed = time.perf_counter() - t0
    # whole-job figures: MAX elapsed over ranks, SUM of samples per step, SUM of failed frames (must be 0)
    from claxon_amd import shard
    res = batch.results()
    elapsed, samples_per_step_all, n_bad = shard.reduce_job(dist if world > 1 else None, elapsed, w.total_samples,
                                                            int((res["status"] != 0).sum()), device=dev)
    if n_bad:
        raise SystemExit("bench: %d frames failed to decode in the timed region" % n_bad)
    ms_per_step = 1e3 * elapsed / args.steps
    value = samples_per_step_all / (ms_per_step * 1e-3) / 1e6

    # ---- per-kernel durations (HIP events recorded by the library on the launch stream)
    batch.set_profiling(True)
    acc = {}
    reps = max(5, min(args.steps, 20))
    for _ in range(reps):
        step()
        torch.cuda.synchronize()
        for name, ms in batch.kernel_times().items():
            acc.setdefault(name, []).append(ms)
    batch.set_profiling(False)
    kernel_ms = {k: float(np.mean(v)) for k, v in acc.items()}
    dom_name = max(kernel_ms, key=kernel_ms.get)
    dom_ms = kernel_ms[dom_name]
    alg_bytes = w.algorithmic_bytes          # compressed bytes read once + 4 B per decoded sample written once
    peak = 8000.0                            # GB/s, MI355X HBM3E spec (MI355X_MICROARCH.md)
    achieved = alg_bytes / (dom_ms * 1e-3) / 1e9
    traffic = _pmc_traffic(dom_name, args.frames)
    roofline = {"bound": "hbm", "kernel": dom_name, "achieved": round(achieved, 1), "peak": peak, "unit": "GB/s",
                "frac": round(achieved / peak, 4), "traffic": traffic,
                "kernel_ms": {k: round(v, 4) for k, v in kernel_ms.items()},
                "algorithmic_bytes_per_launch": alg_bytes,
                "step_achieved": round(alg_bytes / (ms_per_step * 1e-3) / 1e9, 1),
                "step_frac": round(alg_bytes / (ms_per_step * 1e-3) / 1e9 / peak, 4)}

    # achievable-copy ceiling of this box: device-to-device copy of 1 GiB, read + write bytes
    if rank == 0:
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

    out = {
        "metric": "decoded Msamples/s (whole node), 4096-sample stereo 16-bit frames",
        "value": round(value, 1), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "i32 (i64 LPC accumulate)", "data": "synthetic",
        "config": {"workload": "BASELINE configs[2]: %d stereo 16-bit frames/GPU, bs 4096, mid/side, LPC order 8 "
                               "(precision 12), Rice partition order 4, optimal k" % args.frames,
                   "frames_per_gpu": args.frames, "samples_per_step": samples_per_step_all,
                   "compressed_bytes_per_gpu": w.compressed_bytes, "bits_per_sample": round(8.0 * w.compressed_bytes / w.total_samples, 3),
                   "parallelism": "frames sharded across %d GPU(s), no collective" % world,
                   "bit_exact": True, "crc16_in_step": bool(args.verify_crc), "kernel_path": args.path, "gen_seconds": round(gen_s, 1)},
        "roofline": roofline,
    }
    if rank == 0 and world == 1:
        out["config"]["pcie_inclusive"] = _pcie_inclusive(ctx, w, descs)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = _cpu_baseline(w)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def _config3_shard(synth, n, rank):
    """Rank r decodes frames with PCM seeds offset by r*n (distinct data, same distribution)."""
    base = synth.BASE_SEED
    synth.BASE_SEED = base + rank * 1_000_003
    try:
        return synth.config3(n)
    finally:
        synth.BASE_SEED = base


def _pcie_inclusive(ctx, w, descs):
    """The same batch through the one-shot entry that takes and returns HOST buffers (clx_decode_frames: H2D of the
    compressed bytes, decode, D2H of the PCM, device buffers allocated inside the call): the rate a caller without
    device-resident data sees.  Reported next to `value`, never as it (SURVEY section 8d "second figure").  Best of 3."""
    try:
        arena = w.arena[:w.arena_len]
        out = np.zeros(w.pcm.size, dtype=np.int32)
        best = None
        for _ in range(3):
            t = time.perf_counter()
            _, res = ctx.decode_frames(arena, descs, w.out_offs, out=out)
            dt = time.perf_counter() - t
            if not (np.all(res["status"] == 0) and np.array_equal(out, w.pcm)):
                return {"error": "host-buffer decode is not bit-exact"}
            best = dt if best is None else min(best, dt)
        return {"value": round(w.total_samples / best / 1e6, 1), "unit": "Msamples/s", "ms": round(best * 1e3, 3),
                "h2d_bytes": int(w.arena_len), "d2h_bytes": int(4 * w.total_samples),
                "note": "pageable host buffers, device buffers allocated per call; best of 3"}
    except Exception as e:                      # never let the secondary figure take the bench line down
        return {"error": "%s: %s" % (type(e).__name__, e)}


def _cpu_baseline(w):
    """The oracle (a C restatement of Claxon's decode path, `kind: port`; real Claxon is Rust and cannot be
    built here) on this box's host cores, decoding the same arena from memory with per-thread recycled output
    buffers (examples/bench_decode.rs:55-78 methodology)."""
    import oracle
    ncpu = os.cpu_count() or 1
    arena = w.arena[:w.arena_len]

    def run(nthreads, reps):
        best = 0.0
        for _ in range(reps):
            t = time.perf_counter()
            r = oracle.decode_batch(arena, w.offs, w.lens, check_crc=True, nthreads=nthreads, want_results=False)
            dt = time.perf_counter() - t
            assert r["samples"] == w.total_samples
            best = max(best, r["samples"] / dt / 1e6)
        return best

    run(1, 1)                                   # warm
    single = run(1, 3)
    multi = run(ncpu, 5) if ncpu > 1 else single
    return {"value": round(multi, 1), "unit": "Msamples/s", "cores": ncpu, "kind": "port",
            "single_thread": round(single, 1),
            "sample": "the full step workload (%d frames = %.1f Msamples), best of 5 passes on %d threads; "
                      "single_thread = best of 3 passes on 1 thread; includes Claxon's per-byte CRC-16" %
                      (w.n, w.total_samples / 1e6, ncpu)}


def _pmc_traffic(kernel, frames):
    """HBM bytes per launch from a committed rocprofv3 --pmc summary of this same workload size, if there is one."""
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(p) as f:
            return json.load(f).get("frames_%d" % frames, {}).get(kernel)
    except Exception:
        return None


if __name__ == "__main__":
    main()
