#!/usr/bin/env python3
"""Developer probe: the per-GPU band kernel of BASELINE configs[4] at N = 1, 2, 4, 8 (8192x8192x3 row-split: bands of
8192 / 4096 / 2048 / 1024 owned rows + halo rows) on one GPU, by rows per thread: what each rank's launch will cost."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402


def main():
    import torch
    pkg = entry.load_package()
    L = pkg.lib()
    W, c = 8192, 3
    pitch = W * c
    stream = torch.cuda.current_stream().cuda_stream
    for radius in (1, 2):
        for N in (1, 2, 4, 8):
            owned = 8192 // N
            ht = radius if N > 1 else 0
            rows = owned + 2 * ht
            band = torch.empty(rows * pitch, dtype=torch.uint8, device="cuda"); band.random_(0, 256)
            out = torch.empty(owned * pitch, dtype=torch.uint8, device="cuda")
            cells = []
            for rpt in (0, 4, 8):
                pkg.check(L.mi_blur_set_option(b"rows_per_thread", rpt))
                for _ in range(300):
                    pkg.check(L.mi_blur_enqueue_band(band.data_ptr(), out.data_ptr(), W, rows, c, radius, ht, ht + owned, stream))
                torch.cuda.synchronize()
                ts = []
                for rep in range(5):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(100):
                        pkg.check(L.mi_blur_enqueue_band(band.data_ptr(), out.data_ptr(), W, rows, c, radius, ht, ht + owned, stream))
                    e1.record(); torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1) * 1e3 / 100)
                us = sorted(ts)[2]
                cells.append(f"rows/thread {rpt}: {us:6.2f} us ({2 * owned * pitch / us / 1e3:5.0f} GB/s)")
            pkg.check(L.mi_blur_set_option(b"rows_per_thread", 0))
            if N > 1:      # the step with the halo rows pulled out of a neighbouring shard first (same device here: the kernel cost only)
                other = torch.empty(rows * pitch, dtype=torch.uint8, device="cuda"); other.random_(0, 256)
                src_top, src_bot = other.data_ptr() + owned * pitch, other.data_ptr() + ht * pitch
                ts = []
                for rep in range(5):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(100):
                        pkg.check(L.mi_blur_halo_pull(band.data_ptr(), src_top, src_bot, W, c, owned, radius, stream))
                        pkg.check(L.mi_blur_enqueue_band(band.data_ptr(), out.data_ptr(), W, rows, c, radius, ht, ht + owned, stream))
                    e1.record(); torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1) * 1e3 / 100)
                cells.append(f"pull + band: {sorted(ts)[2]:6.2f} us")
                ts = []                                # ... and as ONE launch: the band kernel reads the neighbour's rows in place
                for rep in range(5):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(100):
                        pkg.check(L.mi_blur_enqueue_band_peer(band.data_ptr(), out.data_ptr(), W, rows, c, radius, ht, ht + owned, src_top, src_bot, stream))
                    e1.record(); torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1) * 1e3 / 100)
                cells.append(f"band reading peer rows: {sorted(ts)[2]:6.2f} us")
            print(f"radius {radius}  N={N} band of {owned} rows: " + "   ".join(cells), flush=True)
            del band, out


if __name__ == "__main__":
    main()
