#!/usr/bin/env python3
"""Developer probe: device-to-device copy rate on this GPU (the practical ceiling for a read-once/write-once kernel),
next to the blur kernel on the same buffers, interleaved in one process."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
import torch
pkg = entry.load_package(); L = pkg.lib()
torch.cuda.set_device(0)
for name, (n, h, w, c, r) in {"a1one": (5000, 256, 256, 3, 1), "hd5": (64, 1080, 1920, 3, 2), "big": (2, 8192, 8192, 3, 1)}.items():
    nbytes = n * h * w * c
    src = torch.randint(0, 256, (nbytes,), dtype=torch.uint8, device="cuda")
    dst = torch.empty_like(src)
    src32, dst32 = src.view(torch.int32), dst.view(torch.int32)
    def t_copy():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); dst32.copy_(src32); e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) * 1e3
    def t_add():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); torch.add(src32, 1, out=dst32); e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) * 1e3
    def t_blur():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        pkg.check(L.mi_blur_enqueue(src.data_ptr(), dst.data_ptr(), w, h, c, r, n, torch.cuda.current_stream().cuda_stream))
        e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) * 1e3
    res = {"copy_": [], "add(out=)": [], "blur": []}
    for rep in range(8):
        for k, f in (("copy_", t_copy), ("add(out=)", t_add), ("blur", t_blur)):
            t = f()
            if rep: res[k].append(t)
    for k, v in res.items():
        v.sort(); med = v[len(v) // 2]
        print(f"{name:6s} {k:10s} med {med:8.1f} us  {2 * nbytes / med / 1e6:6.2f} TB/s ({2 * nbytes / med / 1e6 / 8 * 100:4.1f}% of 8 TB/s)", flush=True)
