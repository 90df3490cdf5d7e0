"""The library's kernel selection (clx_select_path, claxon_amd/csrc/clx_plan.h) on the batch shapes it was measured on
(tools/bench_configs.py; profiles/r02_bench_configs_sweep_b.txt): the default must be the measured-fastest kernel family for the
four BASELINE workload shapes at 10 000 and 32 000 frames.  Pure host logic, through the simulator build of the same header."""
import ctypes as C

import simlib

BS = 4096


ARGTYPES = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_int]


def choose(frames, channels, bits_per_sample, wide=False, pipelined=False):
    L = simlib.lib()
    L.sim_select_path.argtypes = ARGTYPES
    samples = frames * channels * BS
    r = L.sim_select_path(frames * channels, samples, int(samples * bits_per_sample / 8), 1 if wide else 0, 1 if channels == 1 else 0,
                          1 if pipelined else 0)
    return ("lanes" if r & 1 else "waves"), ("split" if r & 2 else "fused")


def test_selection_follows_the_measurements():
    assert choose(10000, 1, 5.67)[0] == "waves"                       # config 2: 0.270 (waves) against 0.448 ms
    assert choose(32000, 1, 5.67) == ("lanes", "split")               # 0.473 against 0.553
    assert choose(10000, 2, 5.03)[0] == "waves"                       # config 3: 0.411 against 0.709
    assert choose(16000, 2, 5.03)[0] == "waves"
    assert choose(32000, 2, 5.03) == ("lanes", "fused")               # 0.992 against 1.135
    assert choose(2000, 2, 9.8, wide=True) == ("lanes", "split")      # config 4: lanes from a few thousand subframes
    assert choose(32000, 2, 9.8, wide=True) == ("lanes", "split")     # 2.260 (two-wave build) against 2.580 (fused)
    assert choose(8000, 2, 9.5)[0] == "waves"                         # config 5: 1.046 against 1.101
    assert choose(10000, 2, 9.5) == ("lanes", "split")                # 1.127 against 1.160
    assert choose(32000, 2, 9.5) == ("lanes", "fused")                # 1.567 against 1.951


def test_selection_with_several_batches_in_flight():
    """clx_batch_submit: up to twelve runs of the fused lane kernels go out as one grid, which is ahead of the wave kernels' four in
    flight at every size measured (profiles/r03_path_sweep.txt; round 2's unmerged runs lost to them on small batches of short codes)."""
    p = dict(pipelined=True)
    assert choose(600, 1, 5.67, **p)[0] == "lanes"                    # config 2: 0.023 (lanes, merged) against 0.099 ms
    assert choose(2500, 1, 5.67, **p)[0] == "lanes"                   # 0.025 against 0.110
    assert choose(10000, 1, 5.67, **p)[0] == "lanes"
    assert choose(600, 2, 5.03, **p)[0] == "lanes"                    # config 3: 0.038 against 0.137
    assert choose(2500, 2, 5.03, **p)[0] == "lanes"                   # 0.047 against 0.154
    assert choose(10000, 2, 5.03, **p)[0] == "lanes"
    assert choose(1250, 2, 9.8, wide=True, **p)[0] == "lanes"         # config 4
    assert choose(10000, 2, 9.8, wide=True, **p)[0] == "lanes"
    assert choose(1250, 2, 9.5, **p)[0] == "lanes"                    # config 5
    assert choose(10000, 2, 9.5, **p)[0] == "lanes"


def test_unknown_frame_lengths_take_the_middle():
    L = simlib.lib()
    L.sim_select_path.argtypes = ARGTYPES
    assert L.sim_select_path(20000, 20000 * BS, 0, 0, 0, 0) & 1 == 0  # bytes unknown: 7.5 bits per sample assumed -> 36 000 subframes
    assert L.sim_select_path(40000, 40000 * BS, 0, 0, 0, 0) & 1 == 1
