"""Does the spacing of the frames' output blocks matter?  Config 3's frames decode into rows of 16 KiB, one behind the other (Block layout, one frame
behind the other): every row of a wave's store instruction then starts a multiple of 16 KiB from its neighbour's.  The same pipelined steps with every
frame's block `pad` samples further apart (the caller's choice: out_sample_offsets), timed against the contiguous layout.  usage: layout_probe.py [steps]"""
import sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
import claxon_amd as cx, synth
from parity_cases import workload_descs
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 96
ctx = cx.Context(0, wait_s=120)
w = synth.config3(10000)
descs = workload_descs(w)
dev = "cuda:0"
d_arena = torch.from_numpy(w.arena).to(dev)
st = torch.cuda.current_stream().cuda_stream
for pad in (0, 32, 64, 96, 160, 1056, 0):
    offs = np.asarray(w.out_offs, dtype=np.uint64) + np.arange(w.n, dtype=np.uint64) * np.uint64(pad)
    total = int(offs[-1]) + 2 * 4096
    batch = ctx.plan(descs, offs, verify_crc=True, path=0)
    depth = batch.submit_depth
    outs = [torch.zeros(total, dtype=torch.int32, device=dev) for _ in range(depth)]
    arenas = [d_arena] + [d_arena.clone() for _ in range(depth - 1)]
    for i in range(depth):
        batch.submit(arenas[i % depth].data_ptr(), w.arena_len, outs[i % depth].data_ptr(), st)
    batch.flush(st); torch.cuda.synchronize()
    res = batch.results()
    ok = bool(np.all(res["status"] == 0))
    got = outs[0].cpu().numpy()
    ok = ok and all(np.array_equal(got[int(offs[i]):int(offs[i]) + 8192], w.pcm.reshape(w.n, -1)[i]) for i in range(0, w.n, 997))
    ts = []
    for rep in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(steps):
            batch.submit(arenas[i % depth].data_ptr(), w.arena_len, outs[i % depth].data_ptr(), st)
        batch.flush(st); torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / steps * 1e3)
    print("pad %5d samples (%6d B between frames' blocks): %.4f ms per step (min %.4f)  exact %s" % (pad, 4 * pad, sorted(ts)[2], min(ts), ok), flush=True)
    batch.close(); del outs, arenas
