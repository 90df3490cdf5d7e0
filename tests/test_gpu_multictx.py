"""Several contexts at once (one host thread + HIP stream + clx_ctx per GPU; here several on one GPU, which is
what a one-GPU box can show): the C ABI promises that distinct contexts may be used concurrently (include/claxon_hip.h;
`&mut self` at frame.rs:667 makes one reader one thread, distinct readers independent).

  * clx_decode_frames_multi: one batch cut over 1, 2 and 3 contexts by algorithmic weight, each share decoded by its own host
    thread from its own slice of the arena -- results must be identical to the oracle's and to each other
  * two planned batches of different content run 30 times each from two Python threads on two contexts at the same time (ctypes
    releases the GIL: the calls do overlap), every output compared with the oracle"""
import threading

import numpy as np
import pytest

import claxon_amd as cx
import parity_cases as pc
import synth
from claxon_msgs import MSG

pytestmark = pytest.mark.gpu


def oracle_decode(oracle, w, arena=None):
    ref = np.zeros(w.pcm.size, dtype=np.int32)
    a = w.arena[:w.arena_len] if arena is None else arena[:w.arena_len]
    r = oracle.decode_batch(a, w.offs, w.lens, out=ref, out_offs=w.out_offs, nthreads=8)
    return ref, r


def test_multi_context_sharded_decode(oracle):
    w = synth.concat("mix", [synth.config3(300), synth.small_mixed(160, seed_off=5), synth.config5_unique(200)])
    arena = w.arena.copy()
    # damage two frames: a flipped bit in a CRC footer and one in the middle of a frame (garbage or an error -- the same as the oracle's)
    arena[int(w.offs[37] + w.lens[37]) - 1] ^= 0x10
    arena[int(w.offs[411]) + int(w.lens[411]) // 2] ^= 0x04
    descs = pc.workload_descs(w)
    ref, r = oracle_decode(oracle, w, arena)
    assert int(r["statuses"][37]) == cx.FORMAT_ERROR and int(r["msgs"][37]) == MSG["CLX_MSG_FRAME_CRC_MISMATCH"]
    ctxs = [cx.Context(0, wait_s=120) for _ in range(3)]
    ok = r["statuses"] == 0
    for k in (1, 2, 3):
        out, res = cx.decode_frames_multi(ctxs[:k], arena[:w.arena_len], descs, w.out_offs, verify_crc=True)
        assert np.array_equal(res["status"], r["statuses"]), k
        assert np.array_equal(res["msg"], r["msgs"]), k
        assert np.array_equal(res["end_bit"][ok], r["end_bits"][ok]), k
        for i in np.nonzero(ok)[0]:
            a = int(w.out_offs[i]); b = a + int(w.channels[i]) * int(w.block_sizes[i])
            assert np.array_equal(out[a:b], ref[a:b]), (k, i)
    for c in ctxs:
        c.close()


def test_two_contexts_run_concurrently(oracle):
    import torch
    ws = [synth.config3(1500), synth.concat("m", [synth.config5_unique(700), synth.small_mixed(120, seed_off=9)])]
    refs = [oracle_decode(oracle, w) for w in ws]
    ctxs = [cx.Context(0, wait_s=120) for _ in ws]
    state = []
    for w, c in zip(ws, ctxs):
        descs = pc.workload_descs(w)
        d_arena = torch.from_numpy(w.arena).to("cuda:0")
        d_out = torch.full((w.pcm.size,), 0x5a5a5a5a, dtype=torch.int32, device="cuda:0")
        state.append((c.plan(descs, w.out_offs, verify_crc=True), d_arena, d_out))
    torch.cuda.synchronize()
    errors = []
    start = threading.Barrier(len(ws))

    def worker(k):
        try:
            batch, d_arena, d_out = state[k]
            start.wait()
            for _ in range(30):
                batch.run(d_arena.data_ptr(), ws[k].arena_len, d_out.data_ptr())     # the context's own stream
            res = batch.results()                                                          # waits for that stream
            ref, r = refs[k]
            assert np.array_equal(res["status"], r["statuses"]) and np.array_equal(res["end_bit"], r["end_bits"])
            assert np.array_equal(d_out.cpu().numpy(), ref)
        except Exception as e:                                                              # pragma: no cover
            errors.append((k, repr(e)))

    ts = [threading.Thread(target=worker, args=(k,)) for k in range(len(ws))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors
    for b, _, _ in state:
        b.close()
    for c in ctxs:
        c.close()
