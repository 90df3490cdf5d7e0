"""Per-wave timeline of a TIMED REGION of bench.py's kind -- `steps` pipelined submissions from an idle machine, flush, synchronize -- for
the scan and the lean decode kernel (a library built with -DCLX_TIMELINE): how many waves of each are resident over time, i.e. where
the machine is full, where it fills and where it drains.  usage: CLAXON_HIP_LIB=... python tools/timeline_region.py [steps] [bin_us]"""
import ctypes as C, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import claxon_amd as cx, synth
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
binus = float(sys.argv[2]) if len(sys.argv) > 2 else 100.0
n = 10000
ctx = cx.Context(0, wait_s=120)
# WORKLOAD=config5share: one rank's share of the 1 M-frame job (125 003 frames of 16 384 unique ones, tiled) instead of config 3
if os.environ.get("WORKLOAD", "config3") == "config5share":
    from claxon_amd import shard
    ts = synth.config5_tiled(1_000_000, 16384)
    lo, hi = shard.balanced_ranges(ts.weights(), 8)[3]
    w = ts.slice(lo, hi)
else:
    w = synth.config3(n)
descs, _ = cx.descs_from_offsets(w.arena[:w.arena_len], w.offs, w.lens)
d_arena = torch.from_numpy(w.arena).cuda()
b = ctx.plan(descs, w.out_offs, verify_crc=True, path=cx.POOL if os.environ.get("POOL", "off") == "on" else 0)      # POOL=on: clx_k_pool's tickets
depth = b.submit_depth
n_out = depth if depth * 4 * w.total_samples < (64 << 30) else 3        # (the share: 4.1 GB per output buffer)
outs = [torch.zeros(w.total_samples, dtype=torch.int32, device="cuda") for _ in range(n_out)]
n_in = int(os.environ.get("COPIES", "0")) or (depth if depth * w.arena_len < (8 << 30) else 4)       # distinct copies of the input (bench.py: one per step in flight)
arenas = [d_arena] + [d_arena.clone() for _ in range(n_in - 1)]
st = torch.cuda.current_stream().cuda_stream
L = cx.lib()
L.clx_debug_timeline.argtypes = [C.c_int, C.c_void_p, C.c_size_t]
L.clx_debug_timeline_count.argtypes = [C.c_int, C.POINTER(C.c_uint32)]
def region():
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(steps):
        b.submit(arenas[i % len(arenas)].data_ptr(), w.arena_len, outs[i % len(outs)].data_ptr(), st)
    b.flush(st); torch.cuda.synchronize()
    return time.perf_counter() - t0
for _ in range(3):
    region()
assert L.clx_debug_timeline_reset() == 0
el = region()
print("region of %d steps: %.3f ms wall = %.4f ms per step" % (steps, el * 1e3, el * 1e3 / steps))
recs = {}
for kid, name in ((2, "scan"), (3, "lean")):
    cnt = C.c_uint32(0)
    assert L.clx_debug_timeline_count(kid, C.byref(cnt)) == 0
    nw = min(int(cnt.value), 65536)
    tl = np.zeros((max(nw, 1), 14), dtype=np.uint64)
    assert L.clx_debug_timeline(kid, tl.ctypes.data_as(C.c_void_p), max(nw, 1)) == 0
    recs[name] = tl[:nw]
base = min(int(r[:, 0].min()) for r in recs.values() if r.shape[0])
end = max(int(r[:, 1].max()) for r in recs.values() if r.shape[0])
print("device span of the region's waves: %.0f us; waves recorded: scan %d, lean %d" % ((end - base) / 100.0, recs["scan"].shape[0], recs["lean"].shape[0]))
for name, r in recs.items():
    s, e = (r[:, 0].astype(np.int64) - base) / 100.0, (r[:, 1].astype(np.int64) - base) / 100.0
    mhz = (r[:, 3].astype(np.int64) - r[:, 2].astype(np.int64)) / np.maximum(r[:, 1].astype(np.int64) - r[:, 0].astype(np.int64), 1) * 100.0
    q = lambda a: "/".join("%.0f" % v for v in np.percentile(a, [0, 10, 50, 90, 100]))
    print("%s: start us %s  end us %s  duration us %s  shader MHz %s  wave-time %.1f ms" % (name, q(s), q(e), q(e - s), q(mhz), (e - s).sum() / 1e3))
print("resident waves over time (bin %d us):   t_us  scan  lean" % binus)
t = 0.0
S, E = [(r[:, 0].astype(np.int64) - base) / 100.0 for r in recs.values()], [(r[:, 1].astype(np.int64) - base) / 100.0 for r in recs.values()]
while t < (end - base) / 100.0:
    mid = t + binus / 2
    print("   %6.0f  %5d %5d" % (t, int(((S[0] <= mid) & (E[0] > mid)).sum()), int(((S[1] <= mid) & (E[1] > mid)).sum())))
    t += binus
