#!/usr/bin/env python3
"""Developer probe for profiles/r03_placement_channels.txt: the 1080p 5x5 pool (64 images in + 64 out) allocated `n` times
in one process (earlier pools are kept, so every pool lands somewhere else), `launches` launches on each.  Run under
rocprofv3 --kernel-trace [--pmc ...]: dispatch i of blur_direct_kernel belongs to allocation i // launches.
    python3 tools/placement_probe.py [n_allocations] [launches]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402


def main():
    import torch  # noqa: F401
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    launches = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    pkg = entry.load_package()
    keep = []
    for a in range(n):
        ctx = pkg.Context(0, 1920, 1080, 3, 2, max_batch=1, n_slots=1)
        ctx.resident_alloc(64)
        ctx.resident_fill_synthetic(0)
        ctx.reset_timing()
        t0 = time.perf_counter()
        for _ in range(launches):
            ctx.resident_run(64, 64, timed=1)
        tm = ctx.sync()
        print(f"allocation {a}: in {pkg.lib().mi_blur_resident_in(ctx.h):#x} out {pkg.lib().mi_blur_resident_out(ctx.h):#x}  "
              f"{tm['kernel_ms'] * 1e3 / launches:.1f} us per launch (dispatch timestamps), wall {(time.perf_counter() - t0) / launches * 1e6:.1f} us", flush=True)
        keep.append(ctx)
    for ctx in keep:
        ctx.close()


if __name__ == "__main__":
    main()
