import sys, os, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import claxon_amd as cx
cx.LIB_PATH = os.path.join('/root/repo/claxon_amd', sys.argv[1])
import synth
ctx = cx.Context(0, wait_s=60)
w = synth.config3(10000)
descs,_ = cx.descs_from_offsets(w.arena[:w.arena_len], w.offs, w.lens)
d_arena = torch.from_numpy(w.arena).to('cuda:0'); d_out = torch.zeros(w.pcm.size, dtype=torch.int32, device='cuda:0')
b = ctx.plan(descs, w.out_offs, path=cx.PATH_WAVES)
st = torch.cuda.current_stream().cuda_stream
for _ in range(3): b.run(d_arena.data_ptr(), w.arena_len, d_out.data_ptr(), st)
torch.cuda.synchronize(); b.set_profiling(True)
acc={}
for _ in range(10):
    b.run(d_arena.data_ptr(), w.arena_len, d_out.data_ptr(), st); torch.cuda.synchronize()
    for k,v in b.kernel_times().items(): acc.setdefault(k,[]).append(v)
print(sys.argv[1], {k: round(float(np.mean(v)),4) for k,v in acc.items()})
