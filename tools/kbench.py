#!/usr/bin/env python3
"""Developer A/B bench of kernel options on one GPU (not part of the judged bench contract).

    python tools/kbench.py [--shape a1|a1one|hd5|hd3|big] [--reps 5]

Interleaves the option sets in one process (cdna guide §5.4 rule 24) and prints per-launch
kernel time from the dispatch start/stop events plus achieved algorithmic GB/s.
"""
import argparse
import itertools
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

SHAPES = {
    # name: (h, w, c, radius, pool, images per pass, batch)
    "a1": (256, 256, 3, 1, 5000, 5000, 35),
    "a1one": (256, 256, 3, 1, 5000, 5000, 5000),
    "hd5": (1080, 1920, 3, 2, 64, 64, 64),
    "hd3": (1080, 1920, 3, 1, 64, 64, 64),
    "big": (8192, 8192, 3, 1, 2, 2, 1),
    "big1": (8192, 8192, 3, 1, 1, 1, 1),          # the a2 bench shape: one image per launch
    "hd5x": (1080, 1920, 3, 2, 128, 128, 128),    # same frames, twice the pool
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="a1,a1one,hd5")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--streams", type=int, default=1)
    ap.add_argument("--burst", type=int, default=1,
                    help="launches issued back to back per measurement (1 = isolated launches; 100+ = the sustained regime "
                         "bench.py measures, where the clock settles under load)")
    ap.add_argument("--fused", action="store_true", help="the pass as ONE fused dispatch (mi_blur_resident_run_fused) instead of one launch per batch")
    ap.add_argument("--opts", default="stage_dma=0,1;rows_per_thread=8,16;xcd_remap=1")
    args = ap.parse_args()
    import torch  # noqa: F401  (before the library: one HIP runtime)
    pkg = entry.load_package()
    L = pkg.lib()
    keys, vals = [], []
    for kv in args.opts.split(";"):
        k, v = kv.split("=")
        keys.append(k)
        vals.append([int(x) for x in v.split(",")])
    combos = list(itertools.product(*vals))
    for name in args.shape.split(","):
        h, w, c, r, pool, per_pass, batch = SHAPES[name]
        ctx = pkg.Context(0, w, h, c, r, max_batch=1, n_slots=args.streams)
        ctx.resident_alloc(pool)
        ctx.resident_fill_synthetic(0)
        res = {cb: [] for cb in combos}
        for rep in range(args.reps + 1):
            for cb in combos:
                for k, v in zip(keys, cb):
                    pkg.check(L.mi_blur_set_option(k.encode(), v))
                ctx.reset_timing()
                for _ in range(args.burst):
                    if args.fused:
                        ctx.resident_run_fused(per_pass, batch, timed=True)
                    else:
                        ctx.resident_run(per_pass, batch, timed=True)
                tm = ctx.sync()
                if rep:
                    res[cb].append((tm["kernel_ms"] * 1e3 / tm["launches"], tm["bytes_alg"] / tm["launches"]))
        for cb in combos:
            us = sorted(x[0] for x in res[cb])
            med = us[len(us) // 2]
            bpl = res[cb][0][1]
            print(f"{name:6s} {dict(zip(keys, cb))}: launch med {med:9.2f} us  min {us[0]:9.2f} us  "
                  f"{bpl / med / 1e3:8.1f} GB/s alg ({bpl / med / 1e3 / 8000 * 100:5.1f}% of 8 TB/s)", flush=True)
        ctx.close()


if __name__ == "__main__":
    main()
