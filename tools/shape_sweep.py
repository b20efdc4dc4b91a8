#!/usr/bin/env python3
"""Developer probe: the resident launch forms over a grid of frame sizes and batch sizes — looking for cliffs off the BASELINE shapes.
Prints algorithmic GB/s (2*W*H*C bytes per image / dispatch time) for one launch per batch and for the fused pass."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402


def main():
    import torch  # noqa: F401
    pkg = entry.load_package(); L = pkg.lib()
    radius_list = (1, 2)
    shapes = [(64, 64, 3), (250, 250, 3), (256, 256, 3), (320, 240, 3), (640, 480, 1), (1280, 720, 4), (1366, 768, 3), (1920, 1080, 3), (3840, 2160, 3), (8192, 8192, 3)]
    target = 400 << 20                                   # bytes of input per pass (pool), so every pass is HBM-sized
    for (w, h, c) in shapes:
        isz = w * h * c
        n = max(1, min(20000, target // isz))
        for r in radius_list:
            ctx = pkg.Context(0, w, h, c, r, max_batch=1, n_slots=1)     # one stream: launches do not overlap, durations add up
            ctx.resident_alloc(n); ctx.resident_fill_synthetic(0)
            cells = []
            for batch in (1, 8, 35, 500, n):
                if batch > n or (batch != n and n // batch > 4000):
                    continue
                for form in ("launch", "fused"):
                    run = (lambda: ctx.resident_run(n, batch, timed=1)) if form == "launch" else (lambda: ctx.resident_run_fused(n, batch, timed=True))
                    try:
                        t_end = time.perf_counter() + 0.12          # past the ~40 ms clock ramp that follows any idle gap
                        while time.perf_counter() < t_end:
                            for _ in range(4): run()
                            ctx.sync()
                        ctx.reset_timing()
                        reps = 12
                        for _ in range(reps): run()
                        tm = ctx.sync()
                    except Exception as e:
                        cells.append(f"b{batch} {form}: {type(e).__name__}")
                        continue
                    us_pass = tm["kernel_ms"] * 1e3 / reps
                    gbs = 2.0 * isz * n / us_pass / 1e3 if us_pass > 0 else 0.0
                    cells.append(f"b{batch} {form} {gbs:5.0f}")
            print(f"{w}x{h}x{c} r{r} n={n}: " + " | ".join(cells), flush=True)
            ctx.close()


if __name__ == "__main__":
    main()
