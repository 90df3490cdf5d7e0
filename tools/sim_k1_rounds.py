"""How K1 resolves the entry states of a workload's spans, counted by running the production kernel source under the wave
simulator (the CLX_STAT points compile to nothing on the device): spans on the table path (k <= CLX_LUT_KMAX) and the DPP
propagation rounds they take, spans on the arithmetic-walk path.   usage: python tools/sim_k1_rounds.py [config] [n_frames]"""
import ctypes as C, sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import claxon_amd as cx, synth, simlib
from parity_cases import workload_descs, resync_workload

which = sys.argv[1] if len(sys.argv) > 1 else "config3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
w = resync_workload() if which == "resync" else \
    {"config3": synth.config3, "config4": synth.config4, "config5": synth.config5_unique}[which](n)
descs = workload_descs(w)
stats = (C.c_uint64 * 64).in_dll(simlib.lib(), "sim_stats")
for i in range(64): stats[i] = 0
out, res = simlib.decode(w.arena, w.arena_len, descs, w.out_offs, path=cx.PATH_WAVES)[:2]
ok = bool(np.array_equal(np.asarray(out)[:w.pcm.size], w.pcm.ravel()))
spans, rounds, walks = int(stats[56]), int(stats[57]), int(stats[58])
print("%s, %d frames (bit exact: %s): %d spans on the table path, %.2f propagation rounds per span; %d spans on the walk path"
      % (which, w.n, ok, spans, rounds / max(spans, 1), walks))
