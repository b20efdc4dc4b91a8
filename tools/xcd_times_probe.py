#!/usr/bin/env python3
"""Developer probe: per-XCD finish times of single launches of the tiled kernel (mi_blur_debug_xcd_times): which XCD does a
launch wait for, and by how much?  Plain one-launch 3x3 stream, 1080p 5x5, 8192^2; several fresh allocations."""
import os, sys, time
import ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402


def main():
    import torch
    pkg = entry.load_package()
    L = pkg.lib()
    shapes = {"a1one": (256, 256, 3, 1, 5000), "hd5": (1080, 1920, 3, 2, 64), "big1": (8192, 8192, 3, 1, 1)}
    stream = torch.cuda.current_stream().cuda_stream
    end, beg = (C.c_uint64 * 8)(), (C.c_uint64 * 8)()
    for name, (h, w, c, r, n) in shapes.items():
        nbytes = n * h * w * c
        for alloc in range(3):
            a = torch.empty(nbytes, dtype=torch.uint8, device="cuda"); a.random_(0, 256)
            b = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
            for _ in range(150):                                     # clock ramp
                pkg.check(L.mi_blur_enqueue(a.data_ptr(), b.data_ptr(), w, h, c, r, n, stream))
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(40):
                pkg.check(L.mi_blur_enqueue(a.data_ptr(), b.data_ptr(), w, h, c, r, n, stream))
            e1.record(); torch.cuda.synchronize()
            plain = e0.elapsed_time(e1) * 1e3 / 40
            pkg.check(L.mi_blur_set_option(b"debug_xcd_times", 1))
            rows = []
            for rep in range(5):
                pkg.check(L.mi_blur_debug_xcd_times(end, beg, 1))
                for _ in range(3):                                   # keep the GPU busy in front of the examined launch
                    pkg.check(L.mi_blur_enqueue(a.data_ptr(), b.data_ptr(), w, h, c, r, n, stream))
                torch.cuda.synchronize()
                pkg.check(L.mi_blur_debug_xcd_times(end, beg, 1))
                pkg.check(L.mi_blur_enqueue(a.data_ptr(), b.data_ptr(), w, h, c, r, n, stream))
                pkg.check(L.mi_blur_debug_xcd_times(end, beg, 0))
                t0 = min(beg)
                rows.append([(e - t0) / 100.0 for e in end])
            pkg.check(L.mi_blur_set_option(b"debug_xcd_times", 0))
            med = [sorted(r[i] for r in rows)[len(rows) // 2] for i in range(8)]
            print(f"{name} alloc {alloc}: sustained {plain:7.2f} us/launch; one launch, XCD end - first start (us): "
                  + " ".join(f"{v:6.1f}" for v in med) + f"   spread {max(med) - min(med):5.1f} us ({(max(med) - min(med)) / max(med) * 100:4.1f} %)", flush=True)
            del a, b
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
