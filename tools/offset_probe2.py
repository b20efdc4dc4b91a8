#!/usr/bin/env python3
"""Developer probe: input and output streams inside ONE large allocation, output placed D bytes after the input for a
coarse sweep of D — is there a relative placement that the one-launch 5000x256x256x3 blur prefers?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402


def main():
    import torch
    pkg = entry.load_package()
    L = pkg.lib()
    h, w, c, r, n = 256, 256, 3, 1, 5000
    nbytes = n * h * w * c
    stream = torch.cuda.current_stream().cuda_stream
    MB = 1 << 20

    def timed(fn, burst=30, reps=2):
        ts = []
        for rep in range(reps + 1):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(burst):
                fn()
            e1.record()
            torch.cuda.synchronize()
            if rep:
                ts.append(e0.elapsed_time(e1) * 1e3 / burst)
        return sorted(ts)[len(ts) // 2]

    x = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.5:
        x.add_(1)
    torch.cuda.synchronize()
    for arena_try in range(2):
        arena = torch.empty(6 << 30, dtype=torch.uint8, device="cuda")
        arena[:nbytes].random_(0, 256)
        base = arena.data_ptr()
        print(f"arena {arena_try} at {base:#x}")
        for D in [938, 960, 992, 1024, 1056, 1088, 1152, 1280, 1408, 1536, 1792, 2048, 2304, 2560, 3072, 3584, 4096, 4608, 5000]:
            us = timed(lambda: pkg.check(L.mi_blur_enqueue(base, base + D * MB, w, h, c, r, n, stream)))
            print(f"   out = in + {D:5d} MiB: {us:7.2f} us  {2 * nbytes / us / 1e3:6.0f} GB/s", flush=True)
        # input moved too: both shifted by S
        for S in [0, 64, 512, 1024]:
            arena[S * MB:S * MB + nbytes].random_(0, 256)
            us = timed(lambda: pkg.check(L.mi_blur_enqueue(base + S * MB, base + (S + 2048) * MB, w, h, c, r, n, stream)))
            print(f"   in at +{S} MiB, out = in + 2048 MiB: {us:7.2f} us", flush=True)
        del arena
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
