#!/usr/bin/env python3
"""Developer probe: per fresh allocation, the blur (plain one-launch 5000x256x256x3) next to a torch elementwise kernel and
a device memcpy on the SAME buffers — is the allocation-to-allocation spread the kernel's or the memory system's?"""
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

    def timed(fn, burst=40, reps=3):
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

    # warm the clocks first so trial 0 is not a ramp artefact
    x = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.5:
        x.add_(1)
    torch.cuda.synchronize()
    for trial in range(12):
        if trial % 4 == 3:
            junk = [torch.empty((trial * 37 + 11) << 20, dtype=torch.uint8, device="cuda") for _ in range(3)]   # perturb the allocator
        else:
            junk = None
        a = torch.empty(nbytes, dtype=torch.uint8, device="cuda"); a.random_(0, 256)
        b = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
        a32, b32 = a.view(torch.int32), b.view(torch.int32)
        blur = timed(lambda: pkg.check(L.mi_blur_enqueue(a.data_ptr(), b.data_ptr(), w, h, c, r, n, stream)))
        add = timed(lambda: torch.add(a32, 1, out=b32))
        cpy = timed(lambda: b.copy_(a))
        blur2 = timed(lambda: pkg.check(L.mi_blur_enqueue(a.data_ptr(), b.data_ptr(), w, h, c, r, n, stream)))
        gb = 2 * nbytes / 1e3
        print(f"trial {trial:2d} in {a.data_ptr():#x} out {b.data_ptr():#x}: blur {blur:7.2f} us ({gb / blur:6.0f} GB/s)  add {add:7.2f} us ({gb / add:6.0f})  "
              f"memcpy {cpy:7.2f} us ({gb / cpy:6.0f})  blur again {blur2:7.2f} us", flush=True)
        del a, b, a32, b32, junk
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
