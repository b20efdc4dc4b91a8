#!/usr/bin/env python3
"""Developer probe: how the CPU device (mi_blur_cpu_run) scales with threads, and with where the threads may run."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402
import numpy as np


def main():
    pkg = entry.load_package(); L = pkg.lib()
    h, w, c = 256, 256, 3
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 560          # images per call (the batch the pool is woken for)
    gap_us = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0   # the caller's own work between two calls (busy), e.g. building the batch
    a = np.random.default_rng(0).integers(0, 256, (n, h, w, c), dtype=np.uint8)
    o = np.empty_like(a)
    print("allowed CPUs:", len(os.sched_getaffinity(0)), sorted(os.sched_getaffinity(0))[:20], flush=True)
    for nt in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1,2,4,8,16,32,64").split(",")]:
        t_end = time.perf_counter() + 1.5                      # let the scheduler spread the threads
        while time.perf_counter() < t_end:
            L.mi_blur_cpu_run(a.ctypes.data, o.ctypes.data, w, h, c, 1, n, nt)
        t0 = time.perf_counter(); reps = 0; busy = 0.0
        while time.perf_counter() - t0 < 1.0:
            L.mi_blur_cpu_run(a.ctypes.data, o.ctypes.data, w, h, c, 1, n, nt); reps += 1
            if gap_us:
                g0 = time.perf_counter()
                while (time.perf_counter() - g0) * 1e6 < gap_us:
                    pass
                busy += time.perf_counter() - g0
        dt = (time.perf_counter() - t0 - busy) / reps
        print(f"threads {nt:3d}, {n} images per call: {n / dt:10.0f} img/s  {2.0 * a.nbytes / dt / 1e9:7.1f} GB/s  ({dt * 1e6:.0f} us per call)", flush=True)


if __name__ == "__main__":
    main()
