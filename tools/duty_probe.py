#!/usr/bin/env python3
"""Developer probe: per-dispatch duration of big launches back-to-back vs with a host sync (idle gap) between them,
for the full kernel and for its copy-only skeleton (debug_copy).  Separates power/clock effects of sustained load
from kernel structure."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402


def main():
    import torch  # noqa: F401
    pkg = entry.load_package()
    L = pkg.lib()
    shapes = {"hd5": (1080, 1920, 3, 2, 64), "hd3": (1080, 1920, 3, 1, 64), "a1one": (256, 256, 3, 1, 5000)}
    want = sys.argv[1].split(",") if len(sys.argv) > 1 else list(shapes)
    for name, (h, w, c, r, pool) in ((k, shapes[k]) for k in want):
        ctx = pkg.Context(0, w, h, c, r, max_batch=1, n_slots=1)
        ctx.resident_alloc(pool)
        ctx.resident_fill_synthetic(0)
        for dbg in (0, 1):
            pkg.check(L.mi_blur_set_option(b"debug_copy", dbg))
            for mode in ("gapped", "back-to-back", "gapped", "back-to-back"):
                ctx.resident_run(pool, pool, timed=False); ctx.sync(); ctx.reset_timing()
                n = 200
                t0 = time.perf_counter()
                for _ in range(n):
                    ctx.resident_run(pool, pool, timed=1)
                    if mode == "gapped":
                        ctx.sync()
                        time.sleep(0.0002)
                tm = ctx.sync()
                wall = time.perf_counter() - t0
                us = tm["kernel_ms"] * 1e3 / tm["launches"]
                print(f"{name:6s} {'copy-only' if dbg else 'blur     '} {mode:13s}: {us:8.2f} us/dispatch  "
                      f"{tm['bytes_alg'] / tm['launches'] / us / 1e3:7.1f} GB/s   (wall {wall / n * 1e6:7.1f} us per launch)", flush=True)
                ctx.reset_timing()
        pkg.check(L.mi_blur_set_option(b"debug_copy", 0))
        ctx.close()


if __name__ == "__main__":
    main()
