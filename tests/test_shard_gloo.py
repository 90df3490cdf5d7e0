"""The N>1 path on CPU: world_size-2 gloo processes shard a batch exactly like bench.py / a multi-GPU caller would,
each rank handles only its frames, and the whole-job reduction (MAX time, SUM samples) is checked.  The decode
inside each rank is done by the oracle here (no GPU on this box) -- what is under test is the sharding plan and
the cross-rank reduction, which are identical on GPUs (backend nccl = RCCL)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    import synth
    from claxon_amd import shard
    w = synth.concat("mix", [synth.config3(24), synth.small_mixed(40, seed_off=11)])     # same on every rank (seeded)
    ranges = shard.balanced_ranges(shard.frame_weights(w.lens, w.channels, w.block_sizes), world)
    lo, hi = ranges[rank]
    out = np.zeros(w.pcm.size, dtype=np.int32)
    r = oracle.decode_batch(w.arena[:w.arena_len], w.offs[lo:hi], w.lens[lo:hi], out=out, out_offs=w.out_offs[lo:hi])
    n_bad = int((r["statuses"] != 0).sum())
    elapsed, samples, bad = shard.reduce_job(dist, 0.5 + rank, r["samples"], n_bad)
    # every rank wrote only its own frames
    a = int(w.out_offs[lo]) if hi > lo else 0
    b = int(w.out_offs[hi - 1] + int(w.channels[hi - 1]) * int(w.block_sizes[hi - 1])) if hi > lo else 0
    ok_own = bool(np.array_equal(out[a:b], w.pcm[a:b]))
    ok_other = bool(not out[:a].any() and not out[b:].any())
    np.save(os.path.join(out_dir, "r%d.npy" % rank), np.array([elapsed, samples, bad, lo, hi, ok_own, ok_other, w.total_samples]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding(tmp_path):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    res = [np.load(os.path.join(str(tmp_path), "r%d.npy" % r)) for r in range(world)]
    for r in res:
        assert r[0] == 1.5                      # MAX over ranks of (0.5 + rank)
        assert r[1] == r[7] and r[2] == 0       # SUM of samples = whole batch, no failures
        assert r[5] == 1 and r[6] == 1
    assert res[0][3] == 0 and res[0][4] == res[1][3] and res[1][4] == 64     # contiguous, disjoint, complete


def test_balanced_ranges_properties():
    sys.path.insert(0, ROOT)
    from claxon_amd import shard
    rng = np.random.default_rng(1)
    for n in (0, 1, 7, 1000):
        w = rng.integers(1, 100, n)
        for world in (1, 2, 3, 8):
            rs = shard.balanced_ranges(w, world)
            assert len(rs) == world and rs[0][0] == 0 and rs[-1][1] == n
            assert all(rs[i][1] == rs[i + 1][0] for i in range(world - 1))
            if n >= 8 * world:
                tot = [w[a:b].sum() for a, b in rs]
                assert max(tot) - min(tot) <= 2 * w.max()
