#!/usr/bin/env python3
"""Developer probe: the fused stream's window (batches per window of its blockIdx -> tile map) over several fresh pools, next
to the plain one-launch kernel on the same pool; 5000x256x256x3, batch 35."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402


def main():
    import torch, numpy as np  # noqa
    pkg = entry.load_package()
    L = pkg.lib()
    n = 5000
    wins = (1, 2, 4, 8, 16, 32)
    table = {w: [] for w in wins}; table["plain"] = []
    junk = []
    for pool in range(6):
        if pool % 2:
            junk.append(torch.empty((pool * 97 + 13) << 20, dtype=torch.uint8, device="cuda"))
        ctx = pkg.Context(0, 256, 256, 3, 1, max_batch=1, n_slots=1)
        ctx.resident_alloc(n); ctx.resident_fill_synthetic(0)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.25:
            for _ in range(10):
                ctx.resident_run_fused(n, 35)
            ctx.sync()
        res = {k: [] for k in table}
        for rep in range(3):
            for mode in list(wins) + ["plain"]:
                if mode != "plain":
                    pkg.check(L.mi_blur_set_option(b"fused_window", mode))
                ctx.reset_timing()
                for _ in range(40):
                    if mode == "plain":
                        ctx.resident_run(n, n, timed=1)
                    else:
                        ctx.resident_run_fused(n, 35, timed=True)
                tm = ctx.sync()
                if mode != "plain":
                    assert ctx.resident_batches_done() == 143, (mode, ctx.resident_batches_done())
                if rep:
                    res[mode].append(tm["kernel_ms"] * 1e3 / tm["launches"])
        for k in table:
            table[k].append(sorted(res[k])[0] if len(res[k]) < 2 else sum(res[k]) / len(res[k]))
        pkg.check(L.mi_blur_set_option(b"fused_window", 1))
        ctx.close()
    for k in table:
        v = table[k]
        print(f"window {str(k):6s}: " + " ".join(f"{x:7.2f}" for x in v) + f"   median {sorted(v)[len(v) // 2]:7.2f} us", flush=True)


if __name__ == "__main__":
    main()
