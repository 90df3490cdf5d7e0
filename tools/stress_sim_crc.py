"""CPU stress of the CRC-16 the decode lanes gather (clx_crct.h) under the wave simulator: whole batches of which a third of the frames
is damaged -- footer, last subframe's tail, anywhere behind the header -- with the CRC verified, in stream order and with the waves
composed by content, with exact and with bounding descriptors, against the oracle (statuses, messages, end bits, samples of every
frame that passes).  Not part of the test suite.   usage: python tools/stress_sim_crc.py [first seed] [last seed (exclusive)]"""
import sys
import time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import oracle, claxon_amd as cx, synth, simlib
import parity_cases as pc
from parity_util import SimBackend

oracle.build(); simlib.build()
lo, hi = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (0, 4)
ws = [("config3", synth.config3(70)), ("config5", synth.config5_unique(128)), ("shares", pc.crc_share_workload()), ("config4", synth.config4(40)),
      ("lean", pc.lean_workload())]
t0, n = time.time(), 0
for seed in range(lo, hi):
    for name, w in ws:
        for flags in (0, cx.COMPOSE):
            for loose in (0, 4):
                try:
                    pc.check_crc_in_batch(oracle, SimBackend(cx.PATH_LANES | cx.LANES_FUSED | flags), w, seed=1000 + seed, frac=0.35, loose_every=loose)
                except AssertionError as e:
                    if str(e).startswith("("):           # (too few CRC mismatches in this draw: not a parity failure)
                        continue
                    print("MISMATCH", name, "seed", seed, "flags", flags, "loose", loose, str(e)[:400]); sys.exit(1)
                n += 1
    print("seed", seed, "ok:", n, "batches, %.0f s" % (time.time() - t0), flush=True)
print("total mismatches: 0 in", n, "batches")
