#!/usr/bin/env python3
"""Developer A/B that is robust to WHERE the buffers lie: every option set is measured on the same buffers (sustained bursts,
interleaved), over several fresh allocations; reports each allocation and the median over allocations.

    python tools/ab_alloc.py --shape hd5 --opts "prefer_direct=2;xcd_run=0,16,64" --allocs 5
"""
import argparse, itertools, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

SHAPES = {"a1one": (256, 256, 3, 1, 5000), "a1one5": (256, 256, 3, 2, 5000), "hd5": (1080, 1920, 3, 2, 64), "hd3": (1080, 1920, 3, 1, 64),
          "big1": (8192, 8192, 3, 1, 1), "band1024": (1026, 8192, 3, 1, 1), "a1b35": (256, 256, 3, 1, 35), "k1024": (1024, 1024, 3, 1, 100)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="hd5")
    ap.add_argument("--opts", default="prefer_direct=0,2")
    ap.add_argument("--allocs", type=int, default=5)
    ap.add_argument("--burst", type=int, default=40)
    args = ap.parse_args()
    import torch
    pkg = entry.load_package()
    L = pkg.lib()
    keys, vals = [], []
    for kv in args.opts.split(";"):
        k, v = kv.split("=")
        keys.append(k); vals.append([int(x) for x in v.split(",")])
    combos = list(itertools.product(*vals))
    stream = torch.cuda.current_stream().cuda_stream
    for name in args.shape.split(","):
        h, w, c, r, n = SHAPES[name]
        nbytes = n * h * w * c
        table = {cb: [] for cb in combos}
        for alloc in range(args.allocs):
            junk = [torch.empty((alloc * 61 + 5) << 20, dtype=torch.uint8, device="cuda") for _ in range(alloc % 3)]
            a = torch.empty(nbytes, dtype=torch.uint8, device="cuda"); a.random_(0, 256)
            b = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
            res = {cb: [] for cb in combos}
            burst = args.burst if nbytes > (64 << 20) else 400
            for rep in range(4):
                for cb in combos:
                    for k, v in zip(keys, cb):
                        pkg.check(L.mi_blur_set_option(k.encode(), v))
                    for _ in range(burst // 2):
                        pkg.check(L.mi_blur_enqueue(a.data_ptr(), b.data_ptr(), w, h, c, r, n, stream))
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(burst):
                        pkg.check(L.mi_blur_enqueue(a.data_ptr(), b.data_ptr(), w, h, c, r, n, stream))
                    e1.record(); torch.cuda.synchronize()
                    if rep:
                        res[cb].append(e0.elapsed_time(e1) * 1e3 / burst)
            for cb in combos:
                table[cb].append(sorted(res[cb])[1])
            del a, b, junk
            torch.cuda.empty_cache()
        print(f"## {name}: us per launch on {args.allocs} fresh allocations, then the median (GB/s at the median)")
        for cb in combos:
            v = table[cb]
            med = sorted(v)[len(v) // 2]
            print(f"{name:9s} {str(dict(zip(keys, cb))):60s} " + " ".join(f"{x:7.2f}" for x in v) + f"  | median {med:7.2f}  ({2 * nbytes / med / 1e3:5.0f} GB/s, {2 * nbytes / med / 1e3 / 80:4.1f} %)", flush=True)


if __name__ == "__main__":
    main()
