#!/usr/bin/env python3
"""Developer probe: per-dispatch duration of back-to-back launches from a cold (idle) GPU — how long the clock ramp lasts,
for the plain one-launch kernel, the fused stream and the 1080p 5x5 launch.  Idle gaps of 0.2 / 1 / 3 s in between."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402
import ctypes as C


def main():
    import torch  # noqa
    pkg = entry.load_package()
    L = pkg.lib()
    for name, (h, w, c, r, n, fused) in {"a1one_plain": (256, 256, 3, 1, 5000, False), "a1_fused": (256, 256, 3, 1, 5000, True),
                                         "hd5": (1080, 1920, 3, 2, 64, False)}.items():
        ctx = pkg.Context(0, w, h, c, r, max_batch=1, n_slots=1)
        ctx.resident_alloc(n); ctx.resident_fill_synthetic(0)
        for gap in (3.0, 1.0, 0.2, 0.02):
            time.sleep(gap)
            nl = 600
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(nl // 20)]
            stream = torch.cuda.current_stream().cuda_stream
            # torch events around groups of 20 launches on the library's own stream are not visible to torch: use the
            # library's per-launch timestamps instead, harvested in groups
            out = []
            t0 = time.perf_counter()
            for g in range(nl // 20):
                ctx.reset_timing()
                for _ in range(20):
                    if fused:
                        ctx.resident_run_fused(n, 35, timed=True)
                    else:
                        ctx.resident_run(n, n, timed=1)
                tm = ctx.sync()
                out.append((time.perf_counter() - t0, tm["kernel_ms"] * 1e3 / tm["launches"]))
            line = "  ".join(f"{t * 1e3:5.0f}ms:{us:6.1f}" for t, us in out[:12]) + "  ...  " + "  ".join(f"{t * 1e3:5.0f}ms:{us:6.1f}" for t, us in out[-3:])
            print(f"{name:12s} after {gap:4.2f}s idle: {line}", flush=True)
        ctx.close()


if __name__ == "__main__":
    main()
