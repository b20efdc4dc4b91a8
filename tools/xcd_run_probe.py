#!/usr/bin/env python3
"""Developer probe: blockIdx -> tile maps (identity / XCD-contiguous / runs of r tiles dealt to the XCDs in turn) across
fresh allocations of the same buffers, next to a torch elementwise kernel on those buffers."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402


def main():
    import torch
    pkg = entry.load_package()
    L = pkg.lib()
    shapes = {"a1one": (256, 256, 3, 1, 5000, [0, 8, 16, 64, 512]), "hd5": (1080, 1920, 3, 2, 64, [0, 6, 24, 204, 816]),
              "big1": (8192, 8192, 3, 1, 1, [0, 25, 100, 400, 1600])}
    want = sys.argv[1].split(",") if len(sys.argv) > 1 else list(shapes)
    stream = torch.cuda.current_stream().cuda_stream

    def timed(fn, burst=30, reps=3):
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
    for name in want:
        h, w, c, r, n, runs = shapes[name]
        nbytes = n * h * w * c
        for trial in range(8):
            junk = [torch.empty((trial * 53 + 7) << 20, dtype=torch.uint8, device="cuda") for _ in range(trial % 3)]
            a = torch.empty(nbytes, dtype=torch.uint8, device="cuda"); a.random_(0, 256)
            b = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
            ref = None
            cells = []
            for run in runs + [-1]:
                pkg.check(L.mi_blur_set_option(b"xcd_remap", 0 if run < 0 else 1))
                pkg.check(L.mi_blur_set_option(b"xcd_run", max(run, 0)))
                us = timed(lambda: pkg.check(L.mi_blur_enqueue(a.data_ptr(), b.data_ptr(), w, h, c, r, n, stream)))
                if ref is None:
                    ref = b.clone()
                else:
                    assert torch.equal(ref, b), f"map {run} changed the output"
                cells.append(f"{'ident' if run < 0 else ('contig' if run == 0 else 'run' + str(run))} {us:7.2f}")
            pkg.check(L.mi_blur_set_option(b"xcd_remap", 1)); pkg.check(L.mi_blur_set_option(b"xcd_run", 0))
            add = timed(lambda: torch.add(a.view(torch.int32), 1, out=b.view(torch.int32)))
            print(f"{name} trial {trial}: " + "  ".join(cells) + f"  | add {add:7.2f} us", flush=True)
            del a, b, ref, junk
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
