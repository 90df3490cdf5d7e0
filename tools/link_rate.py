"""Measured host<->device copy rates of this box with pinned memory (what bounds any host-to-host decode pipeline)."""
import time, torch
n = 256 << 20
h = torch.empty(n, dtype=torch.uint8, pin_memory=True); d = torch.empty(n, dtype=torch.uint8, device="cuda")
for name, f in (("H2D", lambda: d.copy_(h, non_blocking=True)), ("D2H", lambda: h.copy_(d, non_blocking=True))):
    f(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5): f()
    torch.cuda.synchronize()
    print("%s pinned: %.1f GB/s" % (name, 5 * n / (time.perf_counter() - t) / 1e9))
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
d2 = torch.empty(n, dtype=torch.uint8, device="cuda"); h2 = torch.empty(n, dtype=torch.uint8, pin_memory=True)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(5):
    with torch.cuda.stream(s1): d.copy_(h, non_blocking=True)
    with torch.cuda.stream(s2): h2.copy_(d2, non_blocking=True)
torch.cuda.synchronize()
print("both directions at once: %.1f GB/s each" % (5 * n / (time.perf_counter() - t) / 1e9))
