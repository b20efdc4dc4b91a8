#!/usr/bin/env python3
"""Developer A/B of library builds (e.g. other hipcc scheduling strategies): python3 tools/sched_ab.py <libmi_blur variant .so>
Times, on one allocation each, the one-launch 3x3 stream, the fused stream, 1080p 5x5 and the 8192^2 launch."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402


def main():
    import torch  # noqa: F401
    pkg = entry.load_package()
    pkg.LIB_PATH = os.path.abspath(sys.argv[1])
    L = pkg.lib()
    pkg.check(L.mi_blur_set_option(b"resident_place_trials", 0))
    out = [os.path.basename(sys.argv[1])]
    for (w, h, c, r, pool, label) in ((256, 256, 3, 1, 5000, "3x3 one launch"), (1920, 1080, 3, 2, 64, "1080p 5x5"), (8192, 8192, 3, 1, 1, "8192^2")):
        ctx = pkg.Context(0, w, h, c, r, max_batch=1, n_slots=1)
        ctx.resident_alloc(pool)
        ctx.resident_fill_synthetic(0)
        for _ in range(150):
            ctx.resident_run(pool, pool)
        ctx.sync(); ctx.reset_timing()
        for _ in range(100):
            ctx.resident_run(pool, pool, timed=1)
        tm = ctx.sync()
        out.append(f"{label} {tm['kernel_ms'] * 10:.1f} us")
        if pool == 5000:
            for _ in range(50):
                ctx.resident_run_fused(pool, 35)
            ctx.sync(); ctx.reset_timing()
            for _ in range(100):
                ctx.resident_run_fused(pool, 35, timed=True)
            tm = ctx.sync()
            out.append(f"fused {tm['kernel_ms'] * 10:.1f} us")
        ctx.close()
    print("  ".join(out), flush=True)


if __name__ == "__main__":
    main()
