import os, sys, time, ctypes as C
sys.path.insert(0, os.getcwd())
import __graft_entry__ as entry
import numpy as np
import torch
pkg = entry.load_package(); L = pkg.lib()
W = H = 8192; c = 3; n = 11                      # 11 x 201 MB = 2.2 GB per buffer: past 2^31 bytes
isz = W * H * c
GOLD = "d283787bcc5b6dfd"
p_in, p_out = L.mi_blur_host_alloc(n * isz), L.mi_blur_host_alloc(n * isz)
assert p_in and p_out
L.mi_blur_fill_synthetic(p_in, W, H, c, 0, 1, 8)
for i in range(1, n): C.memmove(p_in + i * isz, p_in, isz)
for label, opts in (("server", {}), ("per-batch launch", {"zero_copy_server": 0}), ("staged DMA", {"zero_copy": 0})):
    for k, v in opts.items(): pkg.check(L.mi_blur_set_option(k.encode(), v))
    C.memset(p_out, 0, n * isz)
    with pkg.Context(0, W, H, c, 1, max_batch=n, n_slots=1) as ctx:
        t0 = time.perf_counter()
        ctx.submit(p_in, p_out, n)
        ctx.sync()
        dt = time.perf_counter() - t0
        bad = [i for i in range(n) if f"{L.mi_blur_fnv1a64(p_out + i * isz, isz):016x}" != GOLD]
        print(f"{label}: {n} x 8192^2 in one submit ({n * isz / 1e9:.2f} GB each way) {dt * 1e3:.0f} ms, kernel {L.mi_blur_last_kernel().decode()}, mismatching images: {bad}", flush=True)
    for k in opts: pkg.check(L.mi_blur_set_option(k.encode(), 1))
# resident: one launch over the whole 2.2 GB pool, and fused
with pkg.Context(0, W, H, c, 1, max_batch=1, n_slots=1) as ctx:
    ctx.resident_alloc(n)
    for i in range(n): ctx.resident_upload(i, p_in, 1)
    for form in ("launch", "fused"):
        if form == "launch": ctx.resident_run(n, n)
        else: ctx.resident_run_fused(n, 3)
        ctx.sync()
        ctx.resident_download(0, p_out, n)
        bad = [i for i in range(n) if f"{L.mi_blur_fnv1a64(p_out + i * isz, isz):016x}" != GOLD]
        print(f"resident {form}: kernel {L.mi_blur_last_kernel().decode()}, mismatching images: {bad}", flush=True)
