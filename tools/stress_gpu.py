"""One-off GPU stress: many more corrupted / truncated frames than the regular tests, all three kernel selections."""
import sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import oracle, claxon_amd as cx, synth
import parity_cases as pc
from parity_util import GpuBackend
ctx = cx.Context(0, wait_s=120)
for name, path in (("waves", cx.PATH_WAVES), ("lanes-split", cx.PATH_LANES | cx.LANES_SPLIT), ("lanes-fused", cx.PATH_LANES | cx.LANES_FUSED)):
    be = GpuBackend(ctx, path)
    for seed in (3000, 4000, 5000):
        seen = pc.check_bitflips(oracle, be, n_frames=48, trials=40, seed=seed)
        pc.check_truncations(oracle, be, n_frames=16, cuts_per_frame=40, seed=seed)
    print(name, "ok", len(seen), flush=True)
# large mixed workloads through the batch API, both lanes builds and waves, compared with the source PCM
for n in (257, 1000):
    w = synth.config5_unique(n)
    for name, path in (("waves", cx.PATH_WAVES), ("lanes-split", cx.PATH_LANES | cx.LANES_SPLIT), ("lanes-fused", cx.PATH_LANES | cx.LANES_FUSED)):
        pc.check_workload(oracle, GpuBackend(ctx, path), w)
    print("config5", n, "ok", flush=True)
w = synth.config4(300)
for name, path in (("waves", cx.PATH_WAVES), ("lanes-split", cx.PATH_LANES | cx.LANES_SPLIT), ("lanes-fused", cx.PATH_LANES | cx.LANES_FUSED)):
    pc.check_workload(oracle, GpuBackend(ctx, path), w)
print("config4 ok")
