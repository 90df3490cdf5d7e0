"""Which tier of the lane kernels the blocks of a workload take, counted by running the production kernel source under the
wave simulator (tests/wavesim; the CLX_STAT points compile to nothing on the device).  Counts are per lane.
usage: python tools/sim_tiers.py [config] [n_frames]"""
import ctypes as C, sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import claxon_amd as cx, synth, simlib
from parity_cases import workload_descs

NAMES = {0: "scan lean ok", 1: "scan lean failed", 2: " .. bad param", 3: " .. partition edge inside", 4: " .. ring/eof margin", 5: " .. code > 32 bits",
         6: " .. tail lane", 7: "scan general ok", 8: "scan lean16 ok",
         16: "D lean16 ok", 17: "D lean16 failed", 18: " .. not transitioned", 19: " .. bad param", 20: " .. partition edge inside",
         21: " .. ring/eof margin", 22: " .. code > 32 bits", 23: " .. row end", 24: " .. needs i64", 25: " .. out of range", 26: " .. idle",
         32: "D lean4 ok", 33: "D general entered", 34: " .. with no_lean", 35: "D general ok", 36: " .. wide",
         40: "lean4 fail: not transitioned", 41: " .. bad param", 42: " .. partition edge inside", 43: " .. ring/eof margin", 44: " .. code > 32 bits",
         45: " .. row end", 46: " .. needs i64", 47: " .. out of range", 48: " .. idle"}
which = sys.argv[1] if len(sys.argv) > 1 else "config5"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 64
w = {"config2": synth.config2, "config3": synth.config3, "config4": synth.config4, "config5": synth.config5_unique}[which](n)
descs = workload_descs(w)
stats = (C.c_uint64 * 64).in_dll(simlib.lib(), "sim_stats")
for i in range(64): stats[i] = 0
out, res = simlib.decode(w.arena, w.arena_len, descs, w.out_offs, path=cx.PATH_LANES | cx.LANES_FUSED)[:2]
print(which, n, "frames; bit exact:", bool(np.array_equal(np.asarray(out)[:w.pcm.size], w.pcm.ravel())))
for i in range(64):
    if stats[i] or i in (0, 16, 32): print("%3d %-34s %12d" % (i, NAMES.get(i, "?"), stats[i] // 1))
