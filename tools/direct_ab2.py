#!/usr/bin/env python3
"""Developer probe: direct vs tiled where the first sweep was ambiguous — (1) one 8192x8192x3 image (and its N = 2, 4, 8
bands) on the same buffers over several allocations; (2) the 143-launch batch-35 stream on 1 and 4 streams."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402


def main():
    import torch
    pkg = entry.load_package()
    L = pkg.lib()
    stream = torch.cuda.current_stream().cuda_stream
    W, c = 8192, 3
    for alloc in range(3):
        for rows in (8192, 4098, 2050, 1026):
            a = torch.empty(rows * W * c, dtype=torch.uint8, device="cuda"); a.random_(0, 256)
            b = torch.empty_like(a)
            res = {}
            for rep in range(3):
                for v in (pkg.VARIANT_TILED, pkg.VARIANT_DIRECT):
                    for _ in range(100):
                        pkg.check(L.mi_blur_enqueue_ex(a.data_ptr(), b.data_ptr(), W, rows, c, 1, 1, 0, rows, v, stream))
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(200):
                        pkg.check(L.mi_blur_enqueue_ex(a.data_ptr(), b.data_ptr(), W, rows, c, 1, 1, 0, rows, v, stream))
                    e1.record(); torch.cuda.synchronize()
                    res.setdefault(v, []).append(e0.elapsed_time(e1) * 1e3 / 200)
            t, d = sorted(res[pkg.VARIANT_TILED])[1], sorted(res[pkg.VARIANT_DIRECT])[1]
            print(f"alloc {alloc} 8192-wide x {rows:5d} rows, 3x3: tiled {t:7.2f} us  direct {d:7.2f} us  direct/tiled {d / t:5.3f}", flush=True)
            del a, b
        torch.cuda.empty_cache()
    # batch-35 stream, 143 launches per pass
    for streams in (1, 4):
        for pd in (0, 1, 0, 1):
            pkg.check(L.mi_blur_set_option(b"prefer_direct", pd))
            ctx = pkg.Context(0, 256, 256, 3, 1, max_batch=1, n_slots=streams)
            ctx.resident_alloc(5000); ctx.resident_fill_synthetic(0)
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 0.3:
                ctx.resident_run(5000, 35); ctx.sync()
            t0 = time.perf_counter()
            for _ in range(40):
                ctx.resident_run(5000, 35)
            ctx.sync()
            dt = time.perf_counter() - t0
            print(f"batch-35 stream, {streams} stream(s), prefer_direct {pd}: {40 * 5000 / dt / 1e6:6.2f} M img/s", flush=True)
            ctx.close()
    pkg.check(L.mi_blur_set_option(b"prefer_direct", 1))


if __name__ == "__main__":
    main()
