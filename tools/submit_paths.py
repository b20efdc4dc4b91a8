#!/usr/bin/env python3
"""Developer bench: mi_blur_submit end to end for the three kinds of caller memory — pageable (the reference's malloc),
pinned with staged copies (zero_copy off), pinned zero-copy (default)."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402
import numpy as np


def main():
    import torch  # noqa: F401
    pkg = entry.load_package(); L = pkg.lib()
    h, w, c, batches = 256, 256, 3, 300
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 35
    slots = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    print(f"batch {n}, {slots} slots")
    nbytes = n * h * w * c
    for kind in ("pageable", "pinned staged", "pinned zero-copy"):
        pkg.check(L.mi_blur_set_option(b"zero_copy", 0 if kind == "pinned staged" else 1))
        if kind == "pageable":
            keep = [(np.zeros(nbytes, np.uint8), np.zeros(nbytes, np.uint8)) for _ in range(slots)]
            bufs = [(a.ctypes.data, b.ctypes.data) for a, b in keep]
        else:
            bufs = [(L.mi_blur_host_alloc(nbytes), L.mi_blur_host_alloc(nbytes)) for _ in range(slots)]
        for (pi, _po) in bufs:
            L.mi_blur_fill_synthetic(pi, w, h, c, 0, n, 4)
        with pkg.Context(0, w, h, c, 1, max_batch=n, n_slots=slots) as ctx:
            for i in range(6):
                ctx.submit(bufs[i % slots][0], bufs[i % slots][1], n)
            ctx.sync(); ctx.reset_timing()
            t0 = time.perf_counter()
            for i in range(batches):
                ctx.submit(bufs[i % slots][0], bufs[i % slots][1], n)
            tm = ctx.sync()
            dt = time.perf_counter() - t0
        print(f"{kind:17s}: {batches * n / dt / 1e3:7.1f} k img/s   h2d {tm['h2d_ms']:.1f} ms  kernel {tm['kernel_ms']:.1f} ms  d2h {tm['d2h_ms']:.1f} ms", flush=True)
        if kind != "pageable":
            for (pi, po) in bufs:
                L.mi_blur_host_free(pi); L.mi_blur_host_free(po)
    pkg.check(L.mi_blur_set_option(b"zero_copy", 1))


if __name__ == "__main__":
    main()
