#!/usr/bin/env python3
"""Developer probe, ONE allocation pattern per fresh process (argv[1]): does the order / neighbourhood in which the input
and output pools are allocated decide the placement-dependent rate of the one-launch 5000x256x256x3 blur?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402


def main():
    import torch
    pat = sys.argv[1]
    pkg = entry.load_package()
    L = pkg.lib()
    h, w, c, r, n = 256, 256, 3, 1, 5000
    nbytes = n * h * w * c
    stream = torch.cuda.current_stream().cuda_stream
    E = lambda nb: torch.empty(nb, dtype=torch.uint8, device="cuda")
    keep = []
    if pat == "in_out":
        a = E(nbytes); b = E(nbytes)
    elif pat == "out_in":
        b = E(nbytes); a = E(nbytes)
    elif pat == "dummy1g_first":
        keep.append(E(1 << 30)); a = E(nbytes); b = E(nbytes)
    elif pat == "dummy_between":
        a = E(nbytes); keep.append(E(256 << 20)); b = E(nbytes)
    elif pat == "free_realloc":
        a = E(nbytes); b = E(nbytes); del a, b; torch.cuda.empty_cache(); a = E(nbytes); b = E(nbytes)
    elif pat == "swap_after_free":
        a = E(nbytes); b = E(nbytes); del a, b; torch.cuda.empty_cache(); b = E(nbytes); a = E(nbytes)
    elif pat == "rounded_1g":
        a = E(1 << 30); b = E(1 << 30)
    else:
        raise SystemExit("pattern?")
    a[:nbytes].random_(0, 256)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.4:          # clock ramp
        for _ in range(10):
            pkg.check(L.mi_blur_enqueue(a.data_ptr(), b.data_ptr(), w, h, c, r, n, stream))
        torch.cuda.synchronize()
    ts = []
    for rep in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(40):
            pkg.check(L.mi_blur_enqueue(a.data_ptr(), b.data_ptr(), w, h, c, r, n, stream))
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / 40)
    add = []
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(40):
            torch.add(a[:nbytes].view(torch.int32), 1, out=b[:nbytes].view(torch.int32))
        e1.record()
        torch.cuda.synchronize()
        add.append(e0.elapsed_time(e1) * 1e3 / 40)
    us = sorted(ts)[len(ts) // 2]
    print(f"{pat:16s} in {a.data_ptr():#x} out {b.data_ptr():#x}: blur {us:7.2f} us ({2 * nbytes / us / 1e3:6.0f} GB/s)   add {sorted(add)[1]:7.2f} us", flush=True)


if __name__ == "__main__":
    main()
