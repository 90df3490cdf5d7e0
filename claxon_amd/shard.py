"""Sharding of independent frames across ranks (one process per GPU).

Every frame is an independent unit (frame.rs:667-779 touches only its own bytes and buffer), so the N-GPU
path is embarrassingly parallel: contiguous frame ranges balanced by algorithmic bytes, each rank decodes its
range on its own GPU, outputs stay on the producing GPU.  No collective on the data path -- the only cross-rank
operations are a barrier around the timed region, a MAX over elapsed times and a SUM of sample counts / status
histograms, all through torch.distributed (RCCL on GPUs, gloo in the CPU tests).
"""
import numpy as np


def balanced_ranges(weights, world):
    """Split range(len(weights)) into `world` contiguous ranges of near-equal total weight.
    Returns a list of (lo, hi).  Deterministic; every rank computes the same plan."""
    w = np.asarray(weights, dtype=np.float64)
    n = w.size
    if n == 0:
        return [(0, 0)] * world
    cum = np.concatenate([[0.0], np.cumsum(w)])
    total = cum[-1]
    cuts = [0]
    for r in range(1, world):
        target = total * r / world
        i = int(np.searchsorted(cum, target, side="left"))
        i = min(max(i, cuts[-1]), n)
        cuts.append(i)
    cuts.append(n)
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def frame_weights(compressed_lens, channels, block_sizes):
    """Algorithmic bytes per frame: compressed bytes read once + 4 B per decoded sample."""
    return (np.asarray(compressed_lens, dtype=np.int64)
            + 4 * np.asarray(channels, dtype=np.int64) * np.asarray(block_sizes, dtype=np.int64))


def reduce_job(dist, elapsed_s, samples, n_bad, device=None):
    """Whole-job figures: max elapsed over ranks, total samples, total failed frames."""
    import torch
    t = torch.tensor([float(elapsed_s)], dtype=torch.float64, device=device)
    c = torch.tensor([int(samples), int(n_bad)], dtype=torch.int64, device=device)
    if dist is not None and dist.is_initialized():      # (a group of one rank too: `bench.py --process-group` rehearses the collectives)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
    return float(t.item()), int(c[0].item()), int(c[1].item())
