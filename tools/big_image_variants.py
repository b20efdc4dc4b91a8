import os, sys
sys.path.insert(0, os.getcwd())
import __graft_entry__ as entry
import torch
pkg = entry.load_package(); L = pkg.lib()
W, H, c = 8192, 8192, 3
pitch = W * c
stream = torch.cuda.current_stream().cuda_stream
for trial in range(3):
    a = torch.empty(H * pitch, dtype=torch.uint8, device="cuda"); a.random_(0, 256)
    o = torch.empty(H * pitch, dtype=torch.uint8, device="cuda")
    for radius in (1, 2):
        cells = []
        for name, opts in (("auto", {}), ("direct", {"prefer_direct": 2}), ("direct bh16", {"prefer_direct": 2, "direct_bh": 16}), ("tiled", {"prefer_direct": 0}), ("tiled rpt4", {"prefer_direct": 0, "rows_per_thread": 4}), ("tiled rpt16", {"prefer_direct": 0, "rows_per_thread": 16})):
            for k, v in opts.items(): pkg.check(L.mi_blur_set_option(k.encode(), v))
            for _ in range(400): pkg.check(L.mi_blur_enqueue(a.data_ptr(), o.data_ptr(), W, H, c, radius, 1, stream))
            torch.cuda.synchronize()
            ts = []
            for rep in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(50): pkg.check(L.mi_blur_enqueue(a.data_ptr(), o.data_ptr(), W, H, c, radius, 1, stream))
                e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3 / 50)
            cells.append(f"{name} {sorted(ts)[2]:.2f}")
            pkg.check(L.mi_blur_set_option(b"prefer_direct", 1)); pkg.check(L.mi_blur_set_option(b"direct_bh", 8)); pkg.check(L.mi_blur_set_option(b"rows_per_thread", 0))
        print(f"alloc {trial} radius {radius}: " + "  ".join(cells), flush=True)
    del a, o
