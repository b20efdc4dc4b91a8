#!/usr/bin/env python3
"""Developer probe: the XCD-contiguous map gives each XCD one eighth of the launch, so the eight streams run N/8 images
apart.  Does the rate depend on that spacing (images per launch) — i.e. on the eight streams colliding in the memory
channel hash — and does that explain the allocation-to-allocation spread?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402


def main():
    import torch
    pkg = entry.load_package()
    L = pkg.lib()
    h, w, c, r = 256, 256, 3, 1
    isz = h * w * c
    nmax = 5200
    stream = torch.cuda.current_stream().cuda_stream

    def timed(fn, burst=30, reps=2):
        ts = []
        for rep in range(reps + 1):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(burst):
                fn()
            e1.record()
            torch.cuda.synchronize()
            if rep:
                ts.append(e0.elapsed_time(e1) * 1e3 / burst)
        return sorted(ts)[len(ts) // 2]

    x = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.5:
        x.add_(1)
    torch.cuda.synchronize()
    for trial in range(6):
        junk = [torch.empty((trial * 53 + 7) << 20, dtype=torch.uint8, device="cuda") for _ in range(trial % 3)]
        a = torch.empty(nmax * isz, dtype=torch.uint8, device="cuda"); a.random_(0, 256)
        b = torch.empty(nmax * isz, dtype=torch.uint8, device="cuda")
        cells = []
        for n in (5000, 4992, 4096, 5001, 5003, 5008, 5040, 5120, 4999, 5200):
            us = timed(lambda: pkg.check(L.mi_blur_enqueue(a.data_ptr(), b.data_ptr(), w, h, c, r, n, stream)))
            cells.append(f"n={n}: {2 * n * isz / us / 1e3:5.0f}")
        print(f"trial {trial} GB/s  " + "  ".join(cells), flush=True)
        del a, b, junk
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
