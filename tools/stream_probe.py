"""Host-to-device pipeline (clx_decode_frames_stream) on the bench workload: ms per call for the three output modes, best of 5.
usage: python tools/stream_probe.py [frames [frames per chunk (0: the default rule)]]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import claxon_amd as cx, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
w = synth.config3(n)
ctx = cx.Context(0, wait_s=120)
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ctx.set_stream_chunk(chunk)
descs, _ = cx.descs_from_offsets(w.arena[:w.arena_len], w.offs, w.lens)
pin_in = cx.PinnedArray((w.arena_len,), np.uint8); pin_in.array[:] = w.arena[:w.arena_len]
pin16 = cx.PinnedArray((w.total_samples * 2,), np.uint8)
pin32 = cx.PinnedArray((w.total_samples,), np.int32)
def best(f, reps=5):
    f(); b = 1e9
    for _ in range(reps):
        t = time.perf_counter(); f(); b = min(b, time.perf_counter() - t)
    return b * 1e3
a = best(lambda: ctx.decode_frames_stream(pin_in.array, descs, w.out_offs, copy_back=False))
b = best(lambda: ctx.decode_frames_stream(pin_in.array, descs, w.out_offs, out=pin16.array, sample_bytes=2))
c = best(lambda: ctx.decode_frames_stream(pin_in.array, descs, w.out_offs, out=pin32.array))
e = best(lambda: ctx.decode_frames_stream(pin_in.array[:int(w.offs[8] + w.lens[8])], descs[:8], w.out_offs[:8], copy_back=False))
print("chunk %s: no pcm back %.3f ms | pcm16 back %.3f ms | i32 back %.3f ms | 8 frames only %.3f ms" % (chunk or "default", a, b, c, e))
