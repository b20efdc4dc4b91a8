#!/usr/bin/env python3
"""Developer probe: direct (LDS-free) vs tiled kernel by launch size (256x256x3 batches) and on other shapes."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402


def main():
    import torch
    pkg = entry.load_package()
    L = pkg.lib()
    stream = torch.cuda.current_stream().cuda_stream
    cases = [("256x256x3", 256, 256, 3, r, n) for r in (1, 2) for n in (8, 35, 70, 140, 280, 500, 1000, 2000, 5000)]
    cases += [("320x240x3", 240, 320, 3, 1, 35), ("320x240x3", 240, 320, 3, 1, 500), ("1024x1024x3", 1024, 1024, 3, 1, 100), ("1024x1024x3", 1024, 1024, 3, 2, 100),
              ("512x512x4", 512, 512, 4, 1, 400), ("512x512x1", 512, 512, 1, 1, 1600), ("4096x4096x3 band 2048+2", 2050, 4096, 3, 1, 1), ("1920x1080x3", 1080, 1920, 3, 1, 8)]
    for name, h, w, c, r, n in cases:
        a = torch.empty(n * h * w * c, dtype=torch.uint8, device="cuda"); a.random_(0, 256)
        b = torch.empty_like(a)
        res = {}
        burst = max(20, min(400, int(20000 / max(n * h * w * c / 2.8e6, 1))))
        for rep in range(3):
            for v in (pkg.VARIANT_TILED, pkg.VARIANT_DIRECT):
                for _ in range(burst // 2):
                    pkg.check(L.mi_blur_enqueue_ex(a.data_ptr(), b.data_ptr(), w, h, c, r, n, 0, h, v, stream))
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(burst):
                    pkg.check(L.mi_blur_enqueue_ex(a.data_ptr(), b.data_ptr(), w, h, c, r, n, 0, h, v, stream))
                e1.record(); torch.cuda.synchronize()
                res.setdefault(v, []).append(e0.elapsed_time(e1) * 1e3 / burst)
        t, d = sorted(res[pkg.VARIANT_TILED])[1], sorted(res[pkg.VARIANT_DIRECT])[1]
        print(f"{name:26s} r={r} n={n:5d}: tiled {t:8.2f} us  direct {d:8.2f} us  direct/tiled {d / t:5.3f}", flush=True)
        del a, b


if __name__ == "__main__":
    main()
