#!/usr/bin/env python3
"""Developer probe: does the RELATIVE placement of the input and output streams in HBM change the kernel's rate?

One big allocation; input at its start, output at input + bytes + pad for a sweep of pads; each point = bursts of
back-to-back launches (torch events around the burst), interleaved over repetitions.  Also re-allocates the buffers a few
times to see how much a fresh allocation alone moves the number."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402


def main():
    import torch
    pkg = entry.load_package()
    L = pkg.lib()
    shapes = {"a1one": (256, 256, 3, 1, 5000), "hd5": (1080, 1920, 3, 2, 64), "big1": (8192, 8192, 3, 1, 1)}
    want = sys.argv[1].split(",") if len(sys.argv) > 1 else list(shapes)
    pads = [0, 256, 4096, 16384, 32768, 65536, 131072, 262144, 524288, 1 << 20, (1 << 20) + 32768, 2 << 20, (2 << 20) + 16384, 3 << 20]
    burst, reps = 40, 4
    stream = torch.cuda.current_stream().cuda_stream
    for name in want:
        h, w, c, r, n = shapes[name]
        nbytes = n * h * w * c
        for alloc in range(3):
            buf = torch.empty(2 * nbytes + (8 << 20), dtype=torch.uint8, device="cuda")
            buf[:nbytes].random_(0, 256)
            base = buf.data_ptr()
            res = {p: [] for p in pads}
            for rep in range(reps + 1):
                for p in pads:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(burst):
                        pkg.check(L.mi_blur_enqueue(base, base + nbytes + p, w, h, c, r, n, stream))
                    e1.record()
                    torch.cuda.synchronize()
                    if rep:
                        res[p].append(e0.elapsed_time(e1) * 1e3 / burst)
            print(f"{name} allocation {alloc} base {base:#x} (base % 2MiB = {base % (2 << 20):#x})")
            for p in pads:
                us = sorted(res[p])[len(res[p]) // 2]
                print(f"   out = in + N + {p:>8d}: {us:8.2f} us  {2 * nbytes / us / 1e3:7.1f} GB/s ({2 * nbytes / us / 1e3 / 80:5.1f} %)", flush=True)
            del buf
            torch.cuda.empty_cache()
        # separate allocations, as the library's resident pool makes them
        for alloc in range(3):
            a = torch.empty(nbytes, dtype=torch.uint8, device="cuda"); a.random_(0, 256)
            b = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
            ts = []
            for rep in range(reps + 1):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(burst):
                    pkg.check(L.mi_blur_enqueue(a.data_ptr(), b.data_ptr(), w, h, c, r, n, stream))
                e1.record()
                torch.cuda.synchronize()
                if rep:
                    ts.append(e0.elapsed_time(e1) * 1e3 / burst)
            us = sorted(ts)[len(ts) // 2]
            print(f"{name} separate allocations {alloc}: in {a.data_ptr():#x} out {b.data_ptr():#x}: {us:8.2f} us  {2 * nbytes / us / 1e3:7.1f} GB/s", flush=True)
            del a, b
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
