"""Experiment driver: per-kernel times for a given library build / output layout (not part of the test suite)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import claxon_amd as cx
import synth
lib = sys.argv[1] if len(sys.argv) > 1 else "libclaxon_hip.so"
skew = int(sys.argv[2]) if len(sys.argv) > 2 else 0          # extra samples between consecutive frames' outputs
path = {"waves": cx.PATH_WAVES, "lanes": cx.PATH_LANES}[sys.argv[3]] if len(sys.argv) > 3 else cx.PATH_WAVES
cx.LIB_PATH = os.path.join(os.path.dirname(cx.LIB_PATH), lib)
ctx = cx.Context(0, wait_s=60)
w = synth.config3(10000)
descs, _ = cx.descs_from_offsets(w.arena[:w.arena_len], w.offs, w.lens)
out_offs = w.out_offs + np.arange(w.n, dtype=np.uint64) * np.uint64(skew)
total = int(out_offs[-1]) + 2 * 4096
d_arena = torch.from_numpy(w.arena).to("cuda:0"); d_out = torch.zeros(total, dtype=torch.int32, device="cuda:0")
b = ctx.plan(descs, out_offs, path=path)
st = torch.cuda.current_stream().cuda_stream
for _ in range(3): b.run(d_arena.data_ptr(), w.arena_len, d_out.data_ptr(), st)
torch.cuda.synchronize(); b.set_profiling(True)
acc = {}
for _ in range(10):
    b.run(d_arena.data_ptr(), w.arena_len, d_out.data_ptr(), st); torch.cuda.synchronize()
    for k, v in b.kernel_times().items(): acc.setdefault(k, []).append(v)
print(lib, "skew", skew, {k: round(float(np.mean(v)), 4) for k, v in acc.items()})
