#!/usr/bin/env python3
"""Developer probe: how much does a fresh ALLOCATION of the same buffers move the kernel's rate (physical placement /
TLB fragment size), and does the allocation size or order change the odds?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402
import ctypes as C


def main():
    import torch
    pkg = entry.load_package()
    L = pkg.lib()
    hip = C.CDLL("libamdhip64.so", mode=C.RTLD_GLOBAL) if False else None
    h, w, c, r, n = 256, 256, 3, 1, 5000
    nbytes = n * h * w * c
    stream = torch.cuda.current_stream().cuda_stream
    burst = 40

    def measure(a_ptr, b_ptr, reps=3):
        ts = []
        for rep in range(reps + 1):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(burst):
                pkg.check(L.mi_blur_enqueue(a_ptr, b_ptr, w, h, c, r, n, stream))
            e1.record()
            torch.cuda.synchronize()
            if rep:
                ts.append(e0.elapsed_time(e1) * 1e3 / burst)
        return sorted(ts)[len(ts) // 2]

    for label, size in (("exact", nbytes), ("1GiB", 1 << 30), ("exact+spacer", nbytes)):
        out = []
        for trial in range(10):
            spacer = torch.empty(37 << 20, dtype=torch.uint8, device="cuda") if label == "exact+spacer" else None
            a = torch.empty(size, dtype=torch.uint8, device="cuda"); a[:nbytes].random_(0, 256)
            b = torch.empty(size, dtype=torch.uint8, device="cuda")
            us = measure(a.data_ptr(), b.data_ptr())
            out.append(us)
            print(f"{label:14s} trial {trial}: in {a.data_ptr():#x} out {b.data_ptr():#x}: {us:8.2f} us  {2 * nbytes / us / 1e3:7.1f} GB/s", flush=True)
            del a, b, spacer
            torch.cuda.empty_cache()
        print(f"{label}: min {min(out):.1f} max {max(out):.1f}")
    # same buffers, measured repeatedly over ~3 s: does the rate flip without any re-allocation?
    a = torch.empty(nbytes, dtype=torch.uint8, device="cuda"); a.random_(0, 256)
    b = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    t0 = time.perf_counter()
    k = 0
    while time.perf_counter() - t0 < 3.0:
        us = measure(a.data_ptr(), b.data_ptr(), reps=1)
        print(f"same buffers t={time.perf_counter() - t0:5.2f}s: {us:8.2f} us", flush=True)
        k += 1
        if k % 5 == 0:
            time.sleep(0.2)


if __name__ == "__main__":
    main()
