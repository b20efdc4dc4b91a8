#!/usr/bin/env python3
"""Developer probe: where a worker workgroup of the zero-copy batch server spends a batch (device-clock stamps,
mi_blur_debug_zc_trace):  python3 tools/zc_trace_probe.py [batch] [slots] [submits] [key=value ...]"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402


def main():
    import torch  # noqa: F401
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 35
    ns = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    n_sub = int(sys.argv[3]) if len(sys.argv) > 3 else 200
    pkg = entry.load_package()
    L = pkg.lib()
    pkg.check(L.mi_blur_set_option(b"zero_copy_trace", 1))
    for a in sys.argv[4:]:
        k, v = a.split("=")
        pkg.check(L.mi_blur_set_option(k.encode(), int(v)), k)
    w, h, c, r = 256, 256, 3, 1
    nbytes = nb * h * w * c
    bufs = [(L.mi_blur_host_alloc(nbytes), L.mi_blur_host_alloc(nbytes)) for _ in range(ns)]
    for (pi, _po) in bufs:
        L.mi_blur_fill_synthetic(pi, w, h, c, 0, nb, 4)
    warm = pkg.Context(0, w, h, c, r, max_batch=nb, n_slots=ns)          # clocks and code objects warm on another context:
    t_end = time.perf_counter() + 0.3                                    # the trace covers a context's FIRST 512 batches
    i = 0
    while time.perf_counter() < t_end:
        warm.submit(bufs[i % ns][0], bufs[i % ns][1], nb)
        i += 1
    warm.sync()
    warm.close()
    ctx = pkg.Context(0, w, h, c, r, max_batch=nb, n_slots=ns)
    n_sub = min(n_sub, 500)
    t0 = time.perf_counter()
    for i in range(n_sub):
        ctx.submit(bufs[i % ns][0], bufs[i % ns][1], nb)
    ctx.sync()
    dt = time.perf_counter() - t0
    nw, head = C.c_int(), C.c_uint()
    out = np.zeros(512 * 2048 * 5, np.uint64)
    n = L.mi_blur_debug_zc_trace(ctx.h, out.ctypes.data_as(C.POINTER(C.c_uint64)), n_sub, C.byref(nw), C.byref(head))
    if n < 0:
        raise SystemExit(f"trace: status {n}")
    t = out[:n * nw.value * 5].reshape(n, nw.value, 5).astype(np.int64)[8:]      # drop the pipeline fill
    tiles = t[:, :, 1]                                                           # tiles a worker took of a batch
    took = tiles > 0
    per_tile = t[:, :, 2][took] / tiles[took] / 100.0                            # us inside a tile (100 MHz ticks)
    first_in = np.where(took, t[:, :, 0], np.iinfo(np.int64).max).min(axis=1)
    last_out = t[:, :, 3].max(axis=1)
    span = (last_out - first_in) / 100.0
    period = np.diff(last_out) / 100.0
    print(f"batch {nb}, {ns} slots, {nw.value} workers {' '.join(sys.argv[4:])}: {dt / n_sub * 1e6:.1f} us per submit (host), {n_sub * nb / dt:.0f} img/s, "
          f"{n_sub * nbytes / dt / 1e9:.1f} GB/s each way")
    print(f"  tiles per worker per batch: min {tiles.min()}  median {np.median(tiles):.0f}  max {tiles.max()}  (even share {tiles.sum(axis=1).mean() / nw.value:.2f}); "
          f"workers that took none of a batch: {100.0 * (~took).mean():.1f} %")
    print(f"  time inside one tile, us: p10 {np.percentile(per_tile, 10):.1f}  median {np.median(per_tile):.1f}  mean {per_tile.mean():.1f}  "
          f"p90 {np.percentile(per_tile, 90):.1f}  max {per_tile.max():.1f}")
    print(f"  batch span (first tile taken -> last tile done) median {np.median(span):.1f} us; batches complete every {np.median(period):.1f} us (median), {period.mean():.1f} us (mean)")
    ctx.close()
    for (pi, po) in bufs:
        L.mi_blur_host_free(pi)
        L.mi_blur_host_free(po)


if __name__ == "__main__":
    main()
