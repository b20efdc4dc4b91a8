import os, sys
sys.path.insert(0, os.getcwd())
import __graft_entry__ as entry
import torch
pkg = entry.load_package(); L = pkg.lib()
for (h, w, c, r, n) in ((1080, 1920, 3, 1, 64), (256, 256, 3, 1, 5000)):
    ctx = pkg.Context(0, w, h, c, r, max_batch=1, n_slots=1)
    ctx.resident_alloc(n); ctx.resident_fill_synthetic(0)
    for batch in (1, 4, 16, 35, 64, 256, n):
        if batch > n: continue
        for _ in range(60): ctx.resident_run_fused(n, batch)
        ctx.sync(); ctx.reset_timing()
        for _ in range(100): ctx.resident_run_fused(n, batch, timed=True)
        tm = ctx.sync()
        print(f"{w}x{h} n={n} fused batch {batch}: {tm['kernel_ms']*1e3/tm['launches']:.2f} us per pass  kernel {L.mi_blur_last_kernel().decode()}", flush=True)
    for _ in range(60): ctx.resident_run(n, n)
    ctx.sync(); ctx.reset_timing()
    for _ in range(100): ctx.resident_run(n, n, timed=True)
    tm = ctx.sync()
    print(f"{w}x{h} n={n} plain one launch: {tm['kernel_ms']*1e3/tm['launches']:.2f} us  kernel {L.mi_blur_last_kernel().decode()}", flush=True)
    ctx.close()
