#!/usr/bin/env python3
"""Developer probe: can the blur kernel read its input from / write its output to pinned HOST memory directly
(over PCIe, no staging copy), and is that faster than copy engines on this platform?

    python tools/zerocopy_probe.py [--images 140]
Variants, each checked against the staged result:
  staged      H2D copy -> kernel (HBM->HBM) -> D2H copy          (what mi_blur_submit does)
  zc_in       kernel reads pinned host input, writes HBM -> D2H copy
  zc_out      H2D copy -> kernel writes pinned host output
  zc_both     kernel reads pinned host input and writes pinned host output
"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=140)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--streams", type=int, default=2)
    args = ap.parse_args()
    import numpy as np
    import torch
    pkg = entry.load_package()
    L = pkg.lib()
    h, w, c, r, n = 256, 256, 3, 1, args.images
    nbytes = n * h * w * c
    torch.cuda.set_device(0)
    S = args.streams
    hin = [torch.empty(nbytes, dtype=torch.uint8).pin_memory() for _ in range(S)]
    hout = [torch.empty(nbytes, dtype=torch.uint8).pin_memory() for _ in range(S)]
    din = [torch.empty(nbytes, dtype=torch.uint8, device="cuda") for _ in range(S)]
    dout = [torch.empty(nbytes, dtype=torch.uint8, device="cuda") for _ in range(S)]
    streams = [torch.cuda.Stream() for _ in range(S)]
    for s in range(S):
        L.mi_blur_fill_synthetic(hin[s].data_ptr(), w, h, c, 0, n, 4)

    def kern(src, dst, st):
        pkg.check(L.mi_blur_enqueue(src.data_ptr(), dst.data_ptr(), w, h, c, r, n, st.cuda_stream))

    def staged(s):
        with torch.cuda.stream(streams[s]):
            din[s].copy_(hin[s], non_blocking=True); kern(din[s], dout[s], streams[s]); hout[s].copy_(dout[s], non_blocking=True)

    def zc_in(s):
        with torch.cuda.stream(streams[s]):
            kern(hin[s], dout[s], streams[s]); hout[s].copy_(dout[s], non_blocking=True)

    def zc_out(s):
        with torch.cuda.stream(streams[s]):
            din[s].copy_(hin[s], non_blocking=True); kern(din[s], hout[s], streams[s])

    def zc_both(s):
        kern(hin[s], hout[s], streams[s])

    want = None
    for name, fn in (("staged", staged), ("zc_in", zc_in), ("zc_out", zc_out), ("zc_both", zc_both)):
        for s in range(S):
            hout[s].zero_()
        for s in range(S):
            fn(s)
        torch.cuda.synchronize()
        got = hout[0].numpy().copy()
        if want is None:
            want = got
        ok = bool(np.array_equal(got, want))
        t0 = time.perf_counter()
        for i in range(args.reps):
            fn(i % S)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.reps
        print(f"{name:8s} {n} images/batch, {S} streams: {dt*1e6:9.1f} us/batch  {n/dt/1e3:8.1f} k img/s  "
              f"{2*nbytes/dt/1e9:6.1f} GB/s over the link  same bytes as staged: {ok}", flush=True)


if __name__ == "__main__":
    main()
