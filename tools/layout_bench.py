#!/usr/bin/env python3
"""Developer bench: HBM rate of the planar <-> interleaved repack kernels (algorithmic bytes = 2*W*H*C per image)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402


def main():
    import torch
    pkg = entry.load_package()
    L = pkg.lib()
    for (h, w, c, n) in [(256, 256, 3, 5000), (1080, 1920, 3, 64), (1080, 1920, 4, 64), (8192, 8192, 3, 2), (256, 256, 1, 5000)]:
        a = torch.randint(0, 256, (n * h * w * c,), dtype=torch.uint8, device="cuda")
        b = torch.empty_like(a)
        for name, fn in (("planar->interleaved", L.mi_blur_planar_to_interleaved), ("interleaved->planar", L.mi_blur_interleaved_to_planar)):
            st = torch.cuda.current_stream().cuda_stream
            for _ in range(3):
                pkg.check(fn(a.data_ptr(), b.data_ptr(), w, h, c, n, st))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            reps = 20
            for _ in range(reps):
                pkg.check(fn(a.data_ptr(), b.data_ptr(), w, h, c, n, st))
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / reps
            print(f"{n}x{w}x{h}x{c} {name}: {us:9.1f} us  {2 * a.numel() / us / 1e3:7.1f} GB/s", flush=True)


if __name__ == "__main__":
    main()
