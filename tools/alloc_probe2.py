#!/usr/bin/env python3
"""Developer probe, one strategy per process (argv[1]): the library's own resident pool (two hipMallocs), re-allocated k
times; sustained a1one bursts after each allocation."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402


def main():
    import torch  # noqa
    strategy = sys.argv[1] if len(sys.argv) > 1 else "realloc"
    pkg = entry.load_package()
    L = pkg.lib()
    h, w, c, r, n = 256, 256, 3, 1, 5000

    def measure(ctx, fused):
        ctx.reset_timing()
        for _ in range(60):
            if fused:
                ctx.resident_run_fused(n, 35, timed=True)
            else:
                ctx.resident_run(n, n, timed=1)
        tm = ctx.sync()
        return tm["kernel_ms"] * 1e3 / tm["launches"]

    ctx = pkg.Context(0, w, h, c, r, max_batch=1, n_slots=1)
    if strategy == "realloc":
        for k in range(6):
            ctx.resident_alloc(n); ctx.resident_fill_synthetic(0)
            print(f"{strategy} alloc {k}: in {L.mi_blur_resident_in(ctx.h):#x} out {L.mi_blur_resident_out(ctx.h):#x}  plain {measure(ctx, False):7.2f} us  fused {measure(ctx, True):7.2f} us", flush=True)
    elif strategy == "spacer_first":
        sp = torch.empty(3 << 30, dtype=torch.uint8, device="cuda"); sp.zero_(); torch.cuda.synchronize(); del sp; torch.cuda.empty_cache()
        for k in range(3):
            ctx.resident_alloc(n); ctx.resident_fill_synthetic(0)
            print(f"{strategy} alloc {k}: plain {measure(ctx, False):7.2f} us  fused {measure(ctx, True):7.2f} us", flush=True)
    elif strategy == "once_longer":
        ctx.resident_alloc(n); ctx.resident_fill_synthetic(0)
        for k in range(6):
            print(f"{strategy} same pool, measurement {k}: plain {measure(ctx, False):7.2f} us  fused {measure(ctx, True):7.2f} us", flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
