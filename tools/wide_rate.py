import os, sys, time
sys.path.insert(0, os.getcwd())
import __graft_entry__ as entry
import torch
pkg = entry.load_package(); L = pkg.lib()
for (w, h, c, n) in ((512, 512, 6, 400), (256, 256, 5, 2000), (1000, 1000, 8, 60), (1920, 1080, 6, 40)):
    d_in = torch.randint(0, 256, (n * h * w * c,), dtype=torch.uint8, device="cuda")
    d_out = torch.empty_like(d_in)
    res = []
    for variant, name in ((pkg.VARIANT_AUTO, "tiled (auto)"), (pkg.VARIANT_GENERIC, "generic")):
        for _ in range(3):
            pkg.check(L.mi_blur_enqueue_ex(d_in.data_ptr(), d_out.data_ptr(), w, h, c, 1, n, 0, h, variant, None))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20 if variant == pkg.VARIANT_AUTO else 3
        e0.record()
        for _ in range(reps):
            pkg.check(L.mi_blur_enqueue_ex(d_in.data_ptr(), d_out.data_ptr(), w, h, c, 1, n, 0, h, variant, torch.cuda.current_stream().cuda_stream))
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        res.append(f"{name}: {us:9.1f} us = {2 * d_in.numel() / us / 1e6:6.2f} TB/s")
    print(f"{n} x {w}x{h}x{c} ({d_in.numel() / 1e6:.0f} MB), 3x3: " + "   ".join(res), flush=True)
