#!/usr/bin/env python3
"""Developer soak of the zero-copy batch server: `seconds` of submits with random batch sizes, producer stalls around the idle
time-out, a small budget (frequent roll-over), every batch verified against a reference output and poisoned before its
buffers are reused.   python3 tools/server_soak.py [seconds] [key=value ...]"""
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
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
    pkg = entry.load_package()
    L = pkg.lib()
    O = entry.load_oracle()
    mixed = False
    for a in sys.argv[2:]:
        k, v = a.split("=")
        if k == "mixed":                 # mixed=1 (the probe's own): every submit draws pinned or pageable buffers for each side
            mixed = int(v) != 0
        else:
            pkg.check(L.mi_blur_set_option(k.encode(), int(v)), k)
    h, w, c, n, nbuf, radius = 96, 320, 3, 12, 4, 1
    host = O.lcg_stream(nbuf * n, h, w, c, first_index=4242)
    want = O.blur_batch(host, radius)
    nbytes = n * h * w * c
    bufs = [(L.mi_blur_host_alloc(nbytes), L.mi_blur_host_alloc(nbytes)) for _ in range(nbuf)]
    as_np = lambda ptr: np.ctypeslib.as_array((C.c_uint8 * nbytes).from_address(ptr)).reshape(n, h, w, c)
    for k, (pi, po) in enumerate(bufs):
        C.memmove(pi, host[k * n:(k + 1) * n].ctypes.data, nbytes)
        C.memset(po, 0xEE, nbytes)
    pag = [(np.ascontiguousarray(host[k * n:(k + 1) * n]).copy(), np.full((n, h, w, c), 0xEE, np.uint8)) for k in range(nbuf)]
    out_pageable = [False] * nbuf
    rng = np.random.default_rng(99)
    ctx = pkg.Context(0, w, h, c, radius, max_batch=n, n_slots=nbuf)
    sizes = [n] * nbuf
    t_end = time.perf_counter() + seconds
    i, t_last = 0, time.perf_counter()
    while time.perf_counter() < t_end:
        k = i % nbuf
        if i >= nbuf:
            ctx.wait_oldest()
            got = pag[k][1] if out_pageable[k] else as_np(bufs[k][1])
            if not (np.array_equal(got[:sizes[k]], want[k * n:k * n + sizes[k]]) and bool((got[sizes[k]:] == 0xEE).all())):
                raise SystemExit(f"MISMATCH at batch {i - nbuf}")
            got[:] = 0xEE
        sizes[k] = int(rng.integers(1, n + 1))
        r = rng.random()
        if r < 0.02:
            time.sleep(float(rng.uniform(0.0001, 0.0008)))         # stalls around the idle time-out
        in_pageable = mixed and rng.random() < 0.5
        out_pageable[k] = mixed and rng.random() < 0.5
        ctx.submit(pag[k][0].ctypes.data if in_pageable else bufs[k][0], pag[k][1].ctypes.data if out_pageable[k] else bufs[k][1], sizes[k])
        i += 1
        if time.perf_counter() - t_last > 10:
            print(f"  {i} batches verified so far", flush=True)
            t_last = time.perf_counter()
    ctx.sync()
    ctx.close()
    print(f"soak OK: {i} batches, every one verified, {seconds:.0f} s {' '.join(sys.argv[2:])}", flush=True)


if __name__ == "__main__":
    main()
