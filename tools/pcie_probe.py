#!/usr/bin/env python3
"""Developer probe: pinned H2D / D2H copy rate vs size (the e2e path's ceiling on this host)."""
import time, torch
torch.cuda.set_device(0)
for mb in (1, 4, 7, 16, 64, 256):
    n = mb << 20
    h = torch.empty(n, dtype=torch.uint8).pin_memory()
    d = torch.empty(n, dtype=torch.uint8, device="cuda")
    for name, fn in (("H2D", lambda: d.copy_(h, non_blocking=True)), ("D2H", lambda: h.copy_(d, non_blocking=True))):
        fn(); torch.cuda.synchronize()
        reps = max(4, 512 // mb)
        t0 = time.perf_counter()
        for _ in range(reps): fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        print(f"{name} {mb:4d} MiB: {dt*1e6:9.1f} us  {n/dt/1e9:6.1f} GB/s", flush=True)
# two streams concurrently
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
n = 7 << 20
h1 = torch.empty(n, dtype=torch.uint8).pin_memory(); h2 = torch.empty(n, dtype=torch.uint8).pin_memory()
d1 = torch.empty(n, dtype=torch.uint8, device="cuda"); d2 = torch.empty(n, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50):
    with torch.cuda.stream(s1): d1.copy_(h1, non_blocking=True)
    with torch.cuda.stream(s2): h2.copy_(d2, non_blocking=True)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
print(f"H2D+D2H concurrently 7 MiB each: {dt*1e6:.1f} us per pair  {2*n/dt/1e9:.1f} GB/s total")
