import os, sys, time
sys.path.insert(0, os.getcwd())
import __graft_entry__ as entry
import torch
pkg = entry.load_package(); L = pkg.lib()
for (w, h) in ((8192, 8192), (8176, 8192), (8208, 8192), (4096, 8192), (16384, 4096), (5456, 8192), (2736, 8192)):
    c, r, n = 3, 1, 2
    ctx = pkg.Context(0, w, h, c, r, max_batch=1, n_slots=1)
    ctx.resident_alloc(n); ctx.resident_fill_synthetic(0)
    isz = w * h * c
    res = []
    for form in ("launch", "fused", "fused window 1"):
        pkg.check(L.mi_blur_set_option(b"fused_window", 1 if form.endswith("1") else 8))
        run = (lambda: ctx.resident_run(n, 1, timed=1)) if form == "launch" else (lambda: ctx.resident_run_fused(n, 1, timed=True))
        t_end = time.perf_counter() + 0.15
        while time.perf_counter() < t_end:
            for _ in range(4): run()
            ctx.sync()
        ctx.reset_timing()
        for _ in range(20): run()
        tm = ctx.sync()
        us = tm["kernel_ms"] * 1e3 / 20
        res.append(f"{form}: {2.0 * isz * n / us / 1e3:6.0f} GB/s")
    pkg.check(L.mi_blur_set_option(b"fused_window", 8))
    print(f"{w}x{h}x3 (row {w*3} B): " + " | ".join(res), flush=True)
    ctx.close()
