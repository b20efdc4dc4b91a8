import os, sys, time
sys.path.insert(0, os.getcwd())
import __graft_entry__ as entry
import torch
pkg = entry.load_package(); L = pkg.lib()
for (h, w, c, r, n) in ((250, 250, 3, 1, 5000), (250, 250, 3, 2, 5000), (768, 1366, 3, 1, 128), (1080, 1918, 3, 2, 64), (256, 256, 3, 1, 5000)):
    ctx = pkg.Context(0, w, h, c, r, max_batch=1, n_slots=1)
    ctx.resident_alloc(n); ctx.resident_fill_synthetic(0)
    isz = h * w * c
    res = []
    for form, batch in (("fused", 35), ("fused", n), ("launch", 35), ("launch", n)):
        run = (lambda: ctx.resident_run_fused(n, batch, timed=True)) if form == "fused" else (lambda: ctx.resident_run(n, batch, timed=1))
        t_end = time.perf_counter() + 0.15
        while time.perf_counter() < t_end:
            for _ in range(4): run()
            ctx.sync()
        ctx.reset_timing()
        for _ in range(20): run()
        tm = ctx.sync()
        us = tm["kernel_ms"] * 1e3 / 20
        res.append(f"{form} b{batch}: {us:8.1f} us {2.0 * isz * n / us / 1e3:6.0f} GB/s")
    print(f"{w}x{h}x{c} r{r} n={n}: " + " | ".join(res), flush=True)
    ctx.close()
