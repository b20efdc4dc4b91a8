#!/usr/bin/env python3
"""Developer probe: the batch-35 stream as 143 launches per pass over 4 streams (created first in the process, so they get
4 hardware queues), tiled vs direct, 3x3 and 5x5."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402


def main():
    import torch  # noqa
    pkg = entry.load_package()
    L = pkg.lib()
    r = int(sys.argv[1]); pd = int(sys.argv[2]); streams = int(sys.argv[3])
    pkg.check(L.mi_blur_set_option(b"prefer_direct", pd))
    ctx = pkg.Context(0, 256, 256, 3, r, max_batch=1, n_slots=streams)
    ctx.resident_alloc(5000); ctx.resident_fill_synthetic(0)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.4:
        ctx.resident_run(5000, 35); ctx.sync()
    best = 0
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(40):
            ctx.resident_run(5000, 35)
        ctx.sync()
        best = max(best, 40 * 5000 / (time.perf_counter() - t0))
    print(f"radius {r}  {streams} stream(s)  prefer_direct {pd}: {best / 1e6:6.2f} M img/s", flush=True)


if __name__ == "__main__":
    main()
