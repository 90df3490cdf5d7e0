"""CPU-side logic check of the PRODUCTION kernel source: claxon_amd/csrc/clx_kernels.hip is compiled
unmodified by g++ against a wave64 lock-step simulator (tests/wavesim) and compared with the oracle.
This is test infrastructure -- the product never runs on the CPU -- but it lets every kernel change be
checked bit-exactly before GPU time is spent.  The same cases run on the real GPU in test_gpu_parity.py."""
import numpy as np
import pytest

import parity_cases as pc
import synth
from parity_util import SimBackend


import claxon_amd as cx


@pytest.fixture(scope="module", params=[cx.PATH_WAVES | cx.K2_LATENCY, cx.PATH_WAVES | cx.K2_THROUGHPUT, cx.PATH_WAVES, cx.PATH_LANES | cx.LANES_SPLIT,
                                        cx.PATH_LANES | cx.LANES_FUSED, cx.PATH_LANES | cx.LANES_FUSED | cx.LANES_GENERAL,
                                        cx.PATH_LANES | cx.LANES_FUSED | cx.COMPOSE, cx.PATH_LANES | cx.LANES_FUSED | cx.POOL],
                ids=["waves", "waves-1w", "waves-mixed", "lanes", "lanes-fused", "lanes-general", "lanes-composed", "lanes-fused-pool"])
def sim(request):
    """The GPU suite's six kernel selections under the same names and the same ABI flags (tests/test_gpu_parity.py; round 6: the
    simulator takes CLX_K2_LATENCY / CLX_K2_THROUGHPUT as the library does) -- wave-per-frame with the multi-wave and the one-wave
    predictor kernels, lane-per-subframe with the split and the fused decode kernels, the fused build with the lean tiers in front,
    with the general kernels alone and with the waves composed by content -- and two of the simulator's own: `waves-mixed` (no K2 flag:
    every predictor build gets its turn by the parity of the slot count) and `lanes-fused-pool` (the scan and the 16-bit tier as
    clx_k_pool's tickets: merged launches with CLX_POOL)."""
    import simlib
    simlib.build()
    return SimBackend(request.param)


@pytest.mark.parametrize("make", [
    lambda: synth.config2(6), lambda: synth.config3(6), lambda: synth.config4(4),
    lambda: synth.config5_unique(24), lambda: synth.small_mixed(60),
], ids=["config2", "config3", "config4", "config5", "small_mixed"])
def test_sim_workloads(oracle, sim, make):
    pc.check_workload(oracle, sim, make())


def test_sim_edges(oracle, sim):
    pc.check_workload(oracle, sim, pc.edge_workload())


def test_sim_never_resynchronising_streams(oracle, sim):
    pc.check_workload(oracle, sim, pc.resync_workload())


def test_sim_range_hops(oracle, sim):
    pc.check_workload(oracle, sim, pc.range_hop_workload())


def test_sim_footer_read_in_batch(oracle, sim):
    pc.check_footer_read_in_batch(oracle, sim)


def test_sim_truncations(oracle, sim):
    pc.check_truncations(oracle, sim, n_frames=6, cuts_per_frame=16)


def test_sim_bitflips(oracle, sim):
    # (the GPU test's cases are 24 frames x 20 flips; half of the flips here keep the CPU suite, which runs the simulator's fibers one wave at a
    #  time, inside a few minutes -- tools/stress_sim.py and tools/stress_gpu.py run thousands)
    seen = pc.check_bitflips(oracle, sim, n_frames=24, trials=10)
    assert len(seen) >= 5


def test_sim_fixtures(oracle, sim):
    pc.check_fixtures(oracle, sim)


def test_sim_fuzz_corpus(oracle, sim):
    pc.check_fuzz_corpus(oracle, sim)


def test_sim_detects_divergent_collectives():
    """The simulator itself must refuse cross-lane operations under divergent control flow."""
    import os, subprocess, sys, tempfile, textwrap
    here = os.path.dirname(os.path.abspath(__file__))
    src = textwrap.dedent("""
        #include <hip/hip_runtime.h>
        __global__ void bad(int* out) { int lane = threadIdx.x; int v = lane;
          if (lane & 1)
            v = __shfl(v, 0, 64);
          else
            v = __shfl(v, 1, 64);
          out[lane] = v; }
        int main() { static int out[64]; SIM_LAUNCH(bad, 1, 64, out); return 0; }
    """)
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.cpp"), "w").write(src)
        exe = os.path.join(d, "t")
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(here, "wavesim", "fake"), "-o", exe,
                               os.path.join(d, "t.cpp")])
        r = subprocess.run([exe], capture_output=True)
        assert r.returncode != 0 and b"divergent" in r.stderr


def test_sim_collectives_semantics():
    """Lanes that leave the kernel AFTER taking part in a vote must still count in it (a bug here once made
    the simulator disagree with the hardware); lanes that left before are neutral."""
    import os, subprocess, tempfile, textwrap
    here = os.path.dirname(os.path.abspath(__file__))
    src = textwrap.dedent("""
        #include <hip/hip_runtime.h>
        __global__ void k(int* out) { int lane = threadIdx.x;
          if (lane >= 60) return;                       // early leavers are neutral
          int a = __all(lane != 4);                     // lane 4 vetoes
          int b = __any(lane == 61);                    // nobody left says yes
          unsigned long long m = __ballot(lane & 1);
          int c = __all(lane != 4);                     // still vetoed although early finishers exit while others read
          int d = __shfl_xor(lane, 1, 64);
          out[lane] = a * 8 + b * 4 + c * 2 + (m == 0x0aaaaaaaaaaaaaaaull) + 16 * d; }
        int main() { static int out[64]; SIM_LAUNCH(k, 1, 64, out);
          for (int i = 0; i < 60; i++) if (out[i] != 1 + 16 * (i ^ 1)) return 1; return 0; }
    """)
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.cpp"), "w").write(src)
        exe = os.path.join(d, "t")
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(here, "wavesim", "fake"), "-o", exe,
                               os.path.join(d, "t.cpp")])
        assert subprocess.run([exe]).returncode == 0


def test_sim_regressions(oracle, sim):
    pc.check_regressions(oracle, sim)


def test_sim_range_hops_take_both_predictors(oracle):
    """The range-hop workload must really cross the limit in the fused lane kernel: 24-bit lean turns before and after the burst,
    i64 blocks inside it (tier counters compiled into the simulator build only)."""
    import ctypes as C
    import simlib
    w = pc.range_hop_workload()
    stats = (C.c_uint64 * 64).in_dll(simlib.lib(), "sim_stats")
    for i in range(64):
        stats[i] = 0
    pc.check_workload(oracle, SimBackend(cx.PATH_LANES | cx.LANES_FUSED), w, verify_crc=False)
    lean, wide = stats[16] + stats[32], stats[36]
    assert lean > 1000 and wide > 1000, (lean, wide)


def test_sim_lean_tiers(oracle):
    """clx_k_lean (the 16-bit tier of the fused lane build): bit-exact on a workload built to use every tier and every way out
    of a lean turn -- and it really does (tier counters compiled into the simulator build only)."""
    import ctypes as C
    import simlib
    simlib.build()
    w = pc.lean_workload()
    stats = (C.c_uint64 * 64).in_dll(simlib.lib(), "sim_stats")
    for i in range(64):
        stats[i] = 0
    pc.check_workload(oracle, SimBackend(cx.PATH_LANES | cx.LANES_FUSED), w, verify_crc=True)
    lean, slow, wide, taken = stats[50] // 64, stats[51] // 64, stats[58] // 64, stats[52]
    why = {"partition edge inside a four / escape": stats[53], "code longer than 32 bits": stats[54], "ring ran dry / end of frame": stats[55],
           "history outside the 16-bit range": stats[56]}
    assert lean > 300 and slow > 10 and wide > 50 and taken >= 8, (lean, slow, wide, taken)      # wide: the 24-bit form of the turn (loud side channels)
    assert all(v > 0 for v in why.values()), why
    assert stats[57] // 64 >= 1          # groups given up to the general kernels (and still bit-exact: they decoded them)


def test_sim_lean24_tiers(oracle):
    """clx_k_lean24 (the split tier: > 16-bit audio, > 12 taps): bit-exact on a workload built to use both instantiations of its turn
    and every way out of one -- and it really takes those groups (tier counters compiled into the simulator build only)."""
    import ctypes as C
    import simlib
    simlib.build()
    w = pc.lean24_workload()
    stats = (C.c_uint64 * 64).in_dll(simlib.lib(), "sim_stats")
    for i in range(64):
        stats[i] = 0
    pc.check_workload(oracle, SimBackend(cx.PATH_LANES | cx.LANES_FUSED), w, verify_crc=True)
    split, slow, bailed, refilled, taken = stats[9] // 64, stats[10] // 64, stats[11] // 64, stats[12] // 64, stats[13]
    why = {"partition edge inside a four / escape": stats[53], "code longer than 32 bits": stats[54], "ring ran dry / end of frame": stats[55]}
    assert stats[52] == 0                      # nothing here is clx_k_lean's (more than 16 bits, or more than 12 taps)
    assert split > 500 and slow > 10 and refilled > 20 and taken >= 10 and bailed >= 1, (split, slow, bailed, refilled, taken)
    assert all(v > 0 for v in why.values()), why


def test_sim_lean24_takes_config4(oracle):
    """config 4 (24-bit, 32 taps, wasted bits, every channel assignment) goes through clx_k_lean24, all turns split ones."""
    import ctypes as C
    import simlib
    simlib.build()
    stats = (C.c_uint64 * 64).in_dll(simlib.lib(), "sim_stats")
    for i in range(64):
        stats[i] = 0
    pc.check_workload(oracle, SimBackend(cx.PATH_LANES | cx.LANES_FUSED), synth.config4(32), verify_crc=True)
    assert stats[13] == 1 and stats[10] == 0 and stats[9] // 64 == 253, (stats[9], stats[10], stats[13])


def test_sim_lean24_truncations_and_flips(oracle):
    """EOF and garbage inside frames the split tier takes (24-bit stereo, up to 32 taps): every cut / flip must give the reference's
    status, message, end bit and samples -- garbage drives the history out of the split evaluation's range (the wave gives the group
    up) often enough."""
    import ctypes as C
    import simlib
    simlib.build()
    stats = (C.c_uint64 * 64).in_dll(simlib.lib(), "sim_stats")
    for i in range(64):
        stats[i] = 0
    sim = SimBackend(cx.PATH_LANES | cx.LANES_FUSED)
    w = synth.config4(6, bs=256)
    rng = np.random.default_rng(7)
    seen = set()
    for i in range(w.n):
        fr = w.arena[int(w.offs[i]):int(w.offs[i] + w.lens[i])].copy()
        for c in sorted(set(rng.integers(8, len(fr), 10).tolist() + [len(fr) - 2, len(fr) - 1, len(fr)])):
            seen.add(pc.assert_same_as_oracle(oracle, sim, fr[:c].copy(), True, "frame %d cut %d" % (i, c)))
        _, _, h = cx.parse_frame_header(fr)
        for trial in range(12):
            g = fr.copy()
            for _ in range(int(rng.integers(1, 4))):
                pos = int(rng.integers(h.header_bytes * 8, len(g) * 8))
                g[pos >> 3] ^= (0x80 >> (pos & 7))
            seen.add(pc.assert_same_as_oracle(oracle, sim, g, False, "frame %d flip %d" % (i, trial)))
    assert len(seen) >= 3, seen
    assert stats[13] > 50 and stats[11] // 64 >= 1, (stats[13], stats[11])


def test_sim_lean_takes_the_bench_shapes(oracle):
    """configs 2 / 3 (the bench workload's shape) and the mixed shapes of config 5 go through clx_k_lean, all turns lean."""
    import ctypes as C
    import simlib
    simlib.build()
    stats = (C.c_uint64 * 64).in_dll(simlib.lib(), "sim_stats")
    for make, groups in ((lambda: synth.config3(32), 1), (lambda: synth.config2(64), 1)):
        for i in range(64):
            stats[i] = 0
        pc.check_workload(oracle, SimBackend(cx.PATH_LANES | cx.LANES_FUSED), make(), verify_crc=False)
        assert stats[52] == groups and stats[51] == 0 and stats[50] // 64 == 255 * groups, (stats[50], stats[51], stats[52])


def test_sim_lean_truncations_and_flips(oracle):
    """EOF and garbage parity inside frames the lean kernel takes (4096-sample 16-bit stereo): every cut / flip must give the
    reference's status, message, end bit and samples."""
    import simlib
    simlib.build()
    sim = SimBackend(cx.PATH_LANES | cx.LANES_FUSED)
    w = synth.config5_unique(6, bs=256)
    rng = np.random.default_rng(5)
    seen = set()
    for i in range(w.n):
        fr = w.arena[int(w.offs[i]):int(w.offs[i] + w.lens[i])].copy()
        for c in sorted(set(rng.integers(8, len(fr), 10).tolist() + [len(fr) - 2, len(fr) - 1, len(fr)])):
            seen.add(pc.assert_same_as_oracle(oracle, sim, fr[:c].copy(), True, "frame %d cut %d" % (i, c)))
        _, _, h = cx.parse_frame_header(fr)
        for trial in range(10):
            g = fr.copy()
            for _ in range(int(rng.integers(1, 4))):
                pos = int(rng.integers(h.header_bytes * 8, len(g) * 8))
                g[pos >> 3] ^= (0x80 >> (pos & 7))
            seen.add(pc.assert_same_as_oracle(oracle, sim, g, False, "frame %d flip %d" % (i, trial)))
    assert len(seen) >= 3, seen


def test_sim_crc16_gathered_by_the_decode_lanes(oracle):
    """Round 4: the lean kernels' lanes gather their frames' CRC-16 from the words they stage anyway (clx_crct.h: the frame's polynomial
    modulo x^15 + x + 1 and its parity -- P = (x + 1)(x^15 + x + 1)), clx_k_finalize judges it, and the stand-alone kernel only
    checks what is left (`todo`: groups the general kernels decoded, frames whose descriptor only bounds them).  Intact batches:
    every frame of a taken group is settled by the lanes.  Damaged batches: "frame CRC mismatch" exactly where the oracle says."""
    import ctypes as C
    import simlib
    simlib.build()
    stats = (C.c_uint64 * 64).in_dll(simlib.lib(), "sim_stats")
    sim = SimBackend(cx.PATH_LANES | cx.LANES_FUSED)

    def counted(f):
        for i in range(64):
            stats[i] = 0
        r = f()
        return r, int(stats[14]), int(stats[15])            # frames settled by the lanes | left to clx_k_crc16_runs

    for w in (synth.config3(70), synth.config5_unique(96), synth.config4(40)):
        _, done, todo = counted(lambda: pc.check_workload(oracle, sim, w, verify_crc=True))
        assert (done, todo) == (w.n, 0), (w.name, done, todo)
    # a third of the frames damaged: the verdicts match the oracle's; some of them are the lanes' own
    n, done, todo = counted(lambda: pc.check_crc_in_batch(oracle, sim, synth.config5_unique(96)))
    assert n >= 10 and done >= 48, (n, done, todo)
    pc.check_crc_in_batch(oracle, sim, synth.config4(40), seed=5)
    # descriptors that only bound their frames (max_bytes runs into the next frame): those frames are the stand-alone kernel's
    n, done, todo = counted(lambda: pc.check_crc_in_batch(oracle, sim, synth.config3(70), seed=9, frac=0.1, loose_every=3))
    assert todo >= 23, (done, todo)
    pc.check_crc_in_batch(oracle, sim, pc.lean_workload(), seed=11, loose_every=5)
    # shares of every shape: 1 .. 8 channels (a frame's shares chain over up to eight lanes, across wave boundaries), blocks of 32 .. 160
    # samples (subframes inside one granule, frames at every offset inside their first granule), 16- and 24-bit: all but a frame or two settled by the lanes
    ws = pc.crc_share_workload()
    _, done, todo = counted(lambda: pc.check_workload(oracle, sim, ws, verify_crc=True))
    assert done >= ws.n - 4 and done + todo == ws.n, (done, todo, ws.n)
    n, done, todo = counted(lambda: pc.check_crc_in_batch(oracle, sim, ws, seed=3, frac=0.3))
    assert n >= 60 and done >= 150, (n, done, todo)


def test_sim_groups_given_up_by_the_lean_kernel(oracle):
    """Every other wave of parity_cases.giveup_workload keeps needing the slow turn and gives its group up: the general kernels decode
    it from its start (rows rewritten), the frames' CRC-16 goes to the stand-alone kernel -- and everything still matches the oracle."""
    import ctypes as C
    import simlib
    simlib.build()
    stats = (C.c_uint64 * 64).in_dll(simlib.lib(), "sim_stats")
    for i in range(64):
        stats[i] = 0
    w = pc.giveup_workload(128)
    pc.check_workload(oracle, SimBackend(cx.PATH_LANES | cx.LANES_FUSED), w, verify_crc=True)
    given_up, taken = stats[57] // 64, stats[52]
    assert (given_up, taken) == (2, 2), (given_up, taken)                  # four waves: two stay with clx_k_lean, two are given up
    assert (stats[14], stats[15]) == (64, 64), (stats[14], stats[15])      # CRC-16: the lanes' own gathering | the stand-alone kernel's


def test_sim_waves_composed_by_content(oracle):
    """clx_k_compose (round 4): behind the scan the frames of a window are dealt to the lanes by content class -- predictor order
    class, constant / verbatim subframes, channel assignment -- so a wave holds one class.  Same samples, statuses and end bits as
    in stream order (the simulator harness also checks that what was dealt is a permutation of the plan's slot pairs inside the
    windows); far fewer wide turns on the mixed workload (one loud or verbatim lane no longer moves a wave of 63 others); and the
    default composes a batch whose descriptors differ in their channel assignment, not one of a single shape."""
    import ctypes as C
    import simlib
    simlib.build()
    stats = (C.c_uint64 * 64).in_dll(simlib.lib(), "sim_stats")

    def counted(w, flags):
        for i in range(64):
            stats[i] = 0
        pc.check_workload(oracle, SimBackend(cx.PATH_LANES | cx.LANES_FUSED | flags), w, verify_crc=True)
        return int(stats[48]), int(stats[50]) // 64, int(stats[58]) // 64          # windows composed, lean turns, wide turns

    w5 = synth.config5_unique(256)
    win0, lean0, wide0 = counted(w5, cx.NO_COMPOSE)
    win1, lean1, wide1 = counted(w5, cx.COMPOSE)
    assert win0 == 0 and win1 == 1
    assert abs((lean0 + wide0) - (lean1 + wide1)) <= 16 and wide1 < 0.6 * wide0, (lean0, wide0, lean1, wide1)      # (the rest: a few slow turns)
    assert counted(w5, 0)[0] == 1                                  # mixed channel assignments: composed by default
    assert counted(synth.config3(300), 0)[0] == 0                  # one shape: left in stream order
    assert counted(synth.config3(70), cx.COMPOSE)[0] == 1          # (forced: still exact)
    # windows end where the block size, the width class or the channel count changes; mono and multi-channel frames stay where they are
    wm = synth.concat("mixed shapes", [synth.config5_unique(80), synth.config4(70), synth.small_mixed(40), synth.config3(66)])
    assert counted(wm, cx.COMPOSE)[0] >= 3


@pytest.mark.parametrize("compose", [cx.NO_COMPOSE, cx.COMPOSE], ids=["stream-order", "composed"])
@pytest.mark.parametrize("first_gen", [7, 0xffffffff], ids=["gen7", "gen-wrap"])
def test_sim_consecutive_runs_on_one_scratch(oracle, compose, first_gen):
    """The multi-run state of clx_batch_submit (round 4) under the simulator: consecutive runs of ONE planned batch
    on ONE set of scratch -- sf_start / errkey as clx_k_finalize leaves them, group marks and CRC parts tagged with the run's
    generation number (a stale part or mark of an EARLIER run must not be honoured, also across a wrap of the number), the run's
    slot maps re-dealt by clx_k_compose from whatever the run before left.  Run r decodes the same frames with DIFFERENT damage
    (which frames fail, which groups are given up and which CRC parts exist changes from run to run): every run matches the
    oracle's decode of its own arena."""
    import simlib
    simlib.build()
    w = synth.concat("runs", [synth.config5_unique(96), pc.giveup_workload(64), synth.config3(40)])
    descs, _ = cx.descs_from_offsets(w.arena[:w.arena_len], w.offs, w.lens, check_crc=False)
    rng = np.random.default_rng(23)

    def damaged(frac):
        a = w.arena.copy()
        for i in range(w.n):
            if rng.uniform() >= frac:
                continue
            lo, hi = int(w.offs[i]) + int(descs["header_bytes"][i]), int(w.offs[i] + w.lens[i])
            for _ in range(int(rng.integers(1, 3))):
                pos = int(rng.integers(8 * lo, 8 * hi))
                a[pos >> 3] ^= (0x80 >> (pos & 7))
        return a

    arenas = [w.arena.copy(), damaged(0.5), damaged(0.3), w.arena.copy(), damaged(0.7)]
    runs = simlib.decode_runs(arenas, w.arena_len, descs, w.out_offs, verify_crc=True, fill=0x31313131,
                              path=cx.PATH_LANES | cx.LANES_FUSED | compose, first_gen=first_gen)
    n_bad = []
    for a, (out, res) in zip(arenas, runs):
        ref = np.full(out.size, 0x31313131, dtype=np.int32)
        r = oracle.decode_batch(a[:w.arena_len], w.offs, w.lens, out=ref, out_offs=w.out_offs, check_crc=True)
        assert np.array_equal(res["status"], r["statuses"]) and np.array_equal(res["msg"], r["msgs"])
        ok = np.nonzero(res["status"] == cx.OK)[0]
        assert np.array_equal(res["end_bit"][ok], r["end_bits"][ok])
        for i in ok:
            lo, hi = int(w.out_offs[i]), int(w.out_offs[i]) + int(w.channels[i]) * int(w.block_sizes[i])
            assert np.array_equal(out[lo:hi], ref[lo:hi]), int(i)
        n_bad.append(int(np.sum(res["status"] != cx.OK)))
    assert n_bad[0] == 0 and n_bad[3] == 0 and min(n_bad[1], n_bad[2], n_bad[4]) >= 10, n_bad


def test_sim_narrow_output_from_the_decode(oracle):
    """CLX_OUT_PCM16 (round 5): the lean kernel writes a stereo frame's 32 samples as one 128-byte line of interleaved 16-bit PCM from the
    tiles it stages anyway; what it leaves to the general kernels goes through the planar scratch and clx_k_narrow_left.  Every frame's
    bytes against the oracle, in stream order and with the waves composed by content, intact and with a fifth of the frames damaged."""
    import ctypes as C
    import simlib
    simlib.build()
    stats = (C.c_uint64 * 64).in_dll(simlib.lib(), "sim_stats")
    w = pc.pcm16_workload()
    for extra in (cx.NO_COMPOSE, cx.COMPOSE):
        for i in range(64):
            stats[i] = 0
        n_ok = pc.check_pcm16(oracle, SimBackend(cx.PATH_LANES | cx.LANES_FUSED | cx.OUT_PCM16 | extra), w)
        assert n_ok == w.n
        assert stats[52] >= 8 and stats[49] >= 4, (int(stats[52]), int(stats[49]))      # groups the lean kernel wrote itself | groups left to the general kernels
    n_ok = pc.check_pcm16(oracle, SimBackend(cx.PATH_LANES | cx.LANES_FUSED | cx.OUT_PCM16), w, damage=0.2, seed=3)
    assert n_ok < w.n


def test_sim_mid_side_that_runs_away_is_left_to_the_general_kernels(oracle):
    """parity_cases.ms_wild_workload: mid/side streams whose samples run away past every range check of clx_k_lean's turns (and past
    2^29, where the short mid/side form stops being the reference's).  The waves that hold them keep needing the slow turn and are given
    up; the general kernels decode them; the intact family stays with clx_k_lean.  Everything equals the oracle's wrapping arithmetic,
    planar and narrow."""
    import ctypes as C
    import simlib
    simlib.build()
    stats = (C.c_uint64 * 64).in_dll(simlib.lib(), "sim_stats")
    for i in range(64):
        stats[i] = 0
    changed = pc.check_ms_wild(oracle, SimBackend(cx.PATH_LANES | cx.LANES_FUSED))
    assert changed > 0
    given_up, taken = stats[57] // 64, stats[52]
    assert (given_up, taken) == (2, 1), (given_up, taken)                  # three waves: two run away and are given up, one stays
    # short blocks: the slow turns' budget never gives such a group up ("not when the end is near") -- since the movers undo mid/side with
    # the short form (cln_ms4, round 6), the slow turn and the prologue do: they look at what they stage
    for bs in (256, 64):
        assert pc.check_ms_wild(oracle, SimBackend(cx.PATH_LANES | cx.LANES_FUSED), bs=bs) > 0
    # the narrow output of the same streams (the low halves of the same samples)
    w, arena = pc.ms_wild_workload()
    descs, _ = cx.descs_from_offsets(w.arena[:w.arena_len], w.offs, w.lens, check_crc=False)
    out, res = SimBackend(cx.PATH_LANES | cx.LANES_FUSED | cx.OUT_PCM16).decode(arena, w.arena_len, descs, w.out_offs, False, fill=0x1111)
    ref = np.zeros(w.pcm.size, dtype=np.int32)
    oracle.decode_batch(arena[:w.arena_len], w.offs, w.lens, out=ref, out_offs=w.out_offs, check_crc=False)
    out = np.asarray(out).view(np.int16) if np.asarray(out).dtype != np.int16 else np.asarray(out)
    bs = int(w.block_sizes[0])
    want = ref.reshape(-1, 2, bs).transpose(0, 2, 1).reshape(-1).astype(np.int16)
    assert np.all(np.asarray(res["status"]) == cx.OK) and np.array_equal(out[:want.size], want)


def test_sim_pool_tickets_in_any_order(oracle):
    """clx_k_pool (round 6): ONE merged launch of several runs whose scan waves and decode waves are tickets off one counter.  The
    tickets are taken (a) as they come, (b) in a random order that still puts every run's scan tickets in front of its decode
    tickets -- runs interleaved, groups out of order, several workers --, (c) in ANY order: a decode ticket taken before its run's scan
    is over gives up waiting (nobody else runs in the simulator) and leaves its group to the general kernels behind the pool, which
    must decode it all the same.  Every run decodes different damage of the same frames; all of it against the oracle."""
    import simlib
    simlib.build()
    w = synth.concat("pool", [synth.config3(40), synth.config5_unique(34), synth.small_mixed(30), pc.giveup_workload(64)])
    descs, _ = cx.descs_from_offsets(w.arena[:w.arena_len], w.offs, w.lens, check_crc=False)
    rng = np.random.default_rng(611)

    def damaged(frac):
        a = w.arena.copy()
        for i in range(w.n):
            if rng.uniform() >= frac:
                continue
            lo, hi = int(w.offs[i]) + int(descs["header_bytes"][i]), int(w.offs[i] + w.lens[i])
            pos = int(rng.integers(8 * lo, 8 * hi))
            a[pos >> 3] ^= (0x80 >> (pos & 7))
        return a

    arenas = [w.arena.copy(), damaged(0.4), damaged(0.2)]
    n_scan, n_dec = simlib.pool_tickets(descs, len(arenas))
    assert n_scan >= 6 and n_dec >= 12
    scan_per_run = n_scan // len(arenas)
    dec_per_run = n_dec // len(arenas)
    # (b) a random interleaving of the runs' ticket streams, each stream = its scan tickets (shuffled) then its groups (shuffled)
    streams = []
    for r in range(len(arenas)):
        sc = rng.permutation(np.arange(r * scan_per_run, (r + 1) * scan_per_run))
        de = rng.permutation(np.arange(n_scan + r * dec_per_run, n_scan + (r + 1) * dec_per_run))
        streams.append(list(sc) + list(de))
    ordered = []
    while any(streams):
        r = int(rng.integers(0, len(streams)))
        if streams[r]:
            ordered.append(streams[r].pop(0))
    orders = {"as they come": (None, 2), "runs interleaved": (np.array(ordered, dtype=np.uint32), 5),
              "any order": (rng.permutation(n_scan + n_dec).astype(np.uint32), 4)}
    refs = []
    for a in arenas:
        ref = np.full(int(w.pcm.size), 0x17171717, dtype=np.int32)
        refs.append((ref, oracle.decode_batch(a[:w.arena_len], w.offs, w.lens, out=ref, out_offs=w.out_offs, check_crc=True)))
    for name, (order, workers) in orders.items():
        runs, stuck = simlib.decode_pool(arenas, w.arena_len, descs, w.out_offs, verify_crc=True, fill=0x17171717,
                                         path=cx.PATH_LANES | cx.LANES_FUSED | cx.NO_COMPOSE | cx.POOL, order=order, workers=workers)
        assert (stuck > 0) == (name == "any order"), (name, stuck)
        for (out, res), (ref, r) in zip(runs, refs):
            assert np.array_equal(res["status"], r["statuses"]) and np.array_equal(res["msg"], r["msgs"]), name
            ok = np.nonzero(res["status"] == cx.OK)[0]
            assert np.array_equal(res["end_bit"][ok], r["end_bits"][ok]), name
            for i in ok:
                lo, hi = int(w.out_offs[i]), int(w.out_offs[i]) + int(w.channels[i]) * int(w.block_sizes[i])
                assert np.array_equal(out[lo:hi], ref[lo:hi]), (name, int(i))
        assert sum(int(np.sum(res["status"] != cx.OK)) for _, res in runs[1:]) >= 10


def test_sim_takes_the_abi_flags(oracle):
    """Round 5's review: `SimBackend(cx.PATH_WAVES | cx.K2_LATENCY)` returned residuals as PCM -- the harness kept a private "stop after K1"
    switch in the flag word, at CLX_K2_LATENCY's bit.  The simulator now takes the ABI's flags and nothing else (its switch is an
    argument), picks the predictor build by them as the library does, and refuses what the ABI does not define."""
    import simlib
    simlib.build()
    w = synth.small_mixed(40, seed_off=90517)
    for flags in (cx.PATH_WAVES | cx.K2_LATENCY, cx.PATH_WAVES | cx.K2_THROUGHPUT):
        pc.check_workload(oracle, SimBackend(flags), w)
    descs, _ = cx.descs_from_offsets(w.arena[:w.arena_len], w.offs, w.lens)
    for bad in (cx.PATH_WAVES | cx.K2_LATENCY | cx.K2_THROUGHPUT, cx.PATH_LANES | cx.LANES_FUSED | cx.K2_LATENCY):
        with pytest.raises(cx.ClaxonError):
            simlib.decode(w.arena, w.arena_len, descs, w.out_offs, path=bad)
    # the harness's own switch still works, apart from the flags: residuals, not samples
    out, res, sfd = simlib.decode(w.arena, w.arena_len, descs, w.out_offs, path=cx.PATH_WAVES | cx.K2_LATENCY, k1_only=True)
    assert np.all(res["status"] == 0) and not np.array_equal(out, w.pcm) and sfd.size > 0


def test_sim_packed_24_bit_output(oracle):
    """CLX_OUT_PCM24 (round 6): packed little-endian 24-bit PCM straight from the decode -- the split tier writes a stereo frame's 32 sample
    pairs as twelve 16-byte pieces from its stage (cln_store_pcm24); everything else goes through the general kernels, which decode
    into staging rows of their workgroup's own and narrow every row themselves (clx_narrow_row; also what CLX_OUT_PCM16 does with the
    groups the lean kernel leaves).  Every frame's bytes against the oracle, intact and with a fifth of the frames damaged."""
    import ctypes as C
    import simlib
    simlib.build()
    stats = (C.c_uint64 * 64).in_dll(simlib.lib(), "sim_stats")
    w = pc.pcm24_workload()
    for i in range(64):
        stats[i] = 0
    assert pc.check_pcm24(oracle, SimBackend(cx.PATH_LANES | cx.LANES_FUSED | cx.OUT_PCM24), w) == w.n
    # the split tier wrote the stereo frames' twelve pieces itself (16-bit frames too: clx_k_lean is not launched), the rest went
    # through the general kernels' staging rows
    assert stats[13] >= 4 and stats[52] == 0 and stats[49] >= 2, (int(stats[13]), int(stats[52]), int(stats[49]))
    assert pc.check_pcm24(oracle, SimBackend(cx.PATH_LANES | cx.LANES_FUSED | cx.OUT_PCM24), w, damage=0.2, seed=5) < w.n


def test_sim_narrow_output_of_mono_frames(oracle):
    """CLX_OUT_PCM16, waves of MONO frames (round 6): the lean kernel's own -- 64 bytes of 16-bit PCM per row and pair of tiles, four lanes
    to a row (cln_store_pcm16_mono) -- with block sizes that end in a lone tile and frames that fail."""
    import ctypes as C
    import simlib
    simlib.build()
    stats = (C.c_uint64 * 64).in_dll(simlib.lib(), "sim_stats")
    S = synth
    rng = np.random.default_rng(77)
    parts = []
    for bs, n in ((4096, 70), (1024 + 16, 64), (64, 64)):
        t = np.arange(bs)
        pcm = np.empty((n, 1, bs), dtype=np.int32)
        for i in range(n):
            pcm[i, 0] = np.clip(np.round(4000.0 * np.sin(2 * np.pi * (50 + 11 * i) * t / 44100.0) + rng.normal(0, 6.0, bs)), -32768, 32767)
        fp = [S.FrameParams() for _ in range(n)]
        for i, f in enumerate(fp):
            f.sf[0] = S.sf(S.SF_LPC if i % 3 else S.SF_FIXED, order=8 if i % 3 else 2, precision=12, partition_order=min(3, max(0, int(np.log2(bs)) - 5)))
        parts.append(S.encode_frames("mono bs%d" % bs, pcm, 1, bs, 16, fp))
    w = S.concat("mono pcm16", parts)
    for i in range(64):
        stats[i] = 0
    assert pc.check_pcm16(oracle, SimBackend(cx.PATH_LANES | cx.LANES_FUSED | cx.OUT_PCM16), w) == w.n
    assert stats[52] >= 2 and stats[49] <= 2, (int(stats[52]), int(stats[49]))      # groups the lean kernel wrote itself | groups left (where block sizes meet)
    assert pc.check_pcm16(oracle, SimBackend(cx.PATH_LANES | cx.LANES_FUSED | cx.OUT_PCM16), w, damage=0.25, seed=9) < w.n


def test_sim_mid_side_undone_by_the_movers(oracle):
    """parity_cases.ms_mover_workload: waves of plain mid/side pairs -- clx_k_lean's turns stage mid and side as decoded, its movers write left and
    right (cln_ms4: four instructions per pair of samples where the lanes took four per sample each).  Planar and CLX_OUT_PCM16, intact and
    with damaged frames, a ragged last wave of whole pairs of tiles and one that ends in a lone tile; the lean kernel keeps every group."""
    import ctypes as C
    import simlib
    simlib.build()
    stats = (C.c_uint64 * 64).in_dll(simlib.lib(), "sim_stats")
    for lone_tail in (False, True):
        w = pc.ms_mover_workload(lone_tail)
        for i in range(64):
            stats[i] = 0
        pc.check_workload(oracle, SimBackend(cx.PATH_LANES | cx.LANES_FUSED), w, verify_crc=True)
        assert stats[58] >= 64 and stats[49] <= 1, (int(stats[58]), int(stats[49]))   # wide turns (per lane): the loud families | groups left to the general kernels (a wave that gave up)
        for i in range(64):
            stats[i] = 0
        assert pc.check_pcm16(oracle, SimBackend(cx.PATH_LANES | cx.LANES_FUSED | cx.OUT_PCM16), w) == w.n
        assert stats[52] >= 10 and stats[49] <= 1, (int(stats[52]), int(stats[49]))   # groups the lean kernel wrote itself | groups left
        assert pc.check_pcm16(oracle, SimBackend(cx.PATH_LANES | cx.LANES_FUSED | cx.OUT_PCM16), w, damage=0.2, seed=5) < w.n
    pc.check_crc_in_batch(oracle, SimBackend(cx.PATH_LANES | cx.LANES_FUSED), w, seed=12)
    # the split tier's waves of plain mid/side pairs (24- and 20-bit audio): the same movers
    w = pc.ms_mover24_workload()
    for i in range(64):
        stats[i] = 0
    pc.check_workload(oracle, SimBackend(cx.PATH_LANES | cx.LANES_FUSED), w, verify_crc=True)
    assert stats[49] <= 1, int(stats[49])                                             # groups left to the general kernels
    pc.check_crc_in_batch(oracle, SimBackend(cx.PATH_LANES | cx.LANES_FUSED), w, seed=13)
