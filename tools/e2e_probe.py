#!/usr/bin/env python3
"""Developer probe: end-to-end (pinned host buffers in -> pinned host buffers out, zero-copy submits) rate by batch size,
number of streams the zero-copy launches alternate over, and rows per thread (smaller tiles = more blocks per launch)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402


def main():
    import torch  # noqa
    pkg = entry.load_package()
    L = pkg.lib()
    w, h, c, r = 256, 256, 3, 1
    for nb in (35, 70, 140, 500):
        nbytes = nb * h * w * c
        nslots = 4
        bufs = [(L.mi_blur_host_alloc(nbytes), L.mi_blur_host_alloc(nbytes)) for _ in range(nslots)]
        for (pi, _po) in bufs:
            L.mi_blur_fill_synthetic(pi, w, h, c, 0, nb, 4)
        for (zs, rpt) in ((1, 0), (2, 0), (3, 0), (2, 24), (3, 16), (3, 24), (3, 32), (4, 16), (4, 24)):
            if True:       # (second knob is now the cap on resident workgroups)
                pkg.check(L.mi_blur_set_option(b"zero_copy_streams", zs))
                pkg.check(L.mi_blur_set_option(b"zero_copy_blocks", rpt))
                ctx = pkg.Context(0, w, h, c, r, max_batch=nb, n_slots=nslots)
                for i in range(8):
                    ctx.submit(bufs[i % nslots][0], bufs[i % nslots][1], nb)
                ctx.sync(); ctx.reset_timing()
                nbatches = max(40, 10000 // nb)
                vals = []
                for rep in range(7):
                    t0 = time.perf_counter()
                    for i in range(nbatches):
                        ctx.submit(bufs[i % nslots][0], bufs[i % nslots][1], nb)
                    ctx.sync()
                    dt = time.perf_counter() - t0
                    vals.append(nbatches * nb / dt)
                best = sorted(vals)[len(vals) // 2]
                import ctypes, hashlib
                dig = hashlib.md5(ctypes.string_at(bufs[0][1], nbytes)).hexdigest()[:8]
                print(f"[out {dig}] batch {nb:4d}  zero_copy_streams {zs}  zero_copy_blocks {rpt}: {best:9.0f} img/s (median of 7; min {min(vals):.0f} max {max(vals):.0f})  {best * h * w * c / 1e9:5.1f} GB/s each way", flush=True)
                ctx.close()
        for (pi, po) in bufs:
            L.mi_blur_host_free(pi); L.mi_blur_host_free(po)
    pkg.check(L.mi_blur_set_option(b"zero_copy_streams", 1)); pkg.check(L.mi_blur_set_option(b"zero_copy_blocks", 0))


if __name__ == "__main__":
    main()
