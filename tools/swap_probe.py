#!/usr/bin/env python3
"""Developer probe: the same two buffers, blur(A -> B) against blur(B -> A), over several fresh allocations — is the
allocation-dependent rate a property of WHICH buffer is read and which is written?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402


def main():
    import torch
    pkg = entry.load_package()
    L = pkg.lib()
    stream = torch.cuda.current_stream().cuda_stream
    shapes = {"hd5": (1080, 1920, 3, 2, 64), "a1one": (256, 256, 3, 1, 5000), "big1": (8192, 8192, 3, 1, 1)}
    for name, (h, w, c, r, n) in shapes.items():
        nbytes = n * h * w * c
        for alloc in range(4):
            junk = [torch.empty((alloc * 61 + 5) << 20, dtype=torch.uint8, device="cuda") for _ in range(alloc % 3)]
            A = torch.empty(nbytes, dtype=torch.uint8, device="cuda"); A.random_(0, 256)
            B = torch.empty(nbytes, dtype=torch.uint8, device="cuda"); B.random_(0, 256)
            res = {"A->B": [], "B->A": []}
            for rep in range(4):
                for lab, (x, y) in (("A->B", (A, B)), ("B->A", (B, A))):
                    for _ in range(20):
                        pkg.check(L.mi_blur_enqueue(x.data_ptr(), y.data_ptr(), w, h, c, r, n, stream))
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(40):
                        pkg.check(L.mi_blur_enqueue(x.data_ptr(), y.data_ptr(), w, h, c, r, n, stream))
                    e1.record(); torch.cuda.synchronize()
                    if rep:
                        res[lab].append(e0.elapsed_time(e1) * 1e3 / 40)
            ab, ba = sorted(res["A->B"])[1], sorted(res["B->A"])[1]
            print(f"{name} alloc {alloc}: A {A.data_ptr():#x} B {B.data_ptr():#x}   A->B {ab:7.2f} us   B->A {ba:7.2f} us   ratio {ba / ab:5.3f}", flush=True)
            del A, B, junk
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
