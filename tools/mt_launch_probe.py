#!/usr/bin/env python3
"""Developer probe: does launching the batch-35 stream from T host threads (T contexts on one GPU) scale?"""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
import torch
pkg = entry.load_package(); L = pkg.lib()
N = 5000
for T, streams in ((1, 4), (2, 2), (2, 4), (4, 2), (4, 4)):
    per = N // T
    ctxs = []
    for t in range(T):
        c = pkg.Context(0, 256, 256, 3, 1, max_batch=1, n_slots=streams)
        c.resident_alloc(per); c.resident_fill_synthetic(t * per); ctxs.append(c)
    def work(c, reps):
        for _ in range(reps):
            c.resident_run(per, 35, 0)
    for c in ctxs: work(c, 2); 
    for c in ctxs: c.sync()
    torch.cuda.synchronize()
    reps = 20
    th = [threading.Thread(target=work, args=(c, reps)) for c in ctxs]
    t0 = time.perf_counter()
    for x in th: x.start()
    for x in th: x.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"threads {T} x streams {streams}: {N*reps/dt/1e6:6.2f} M img/s  ({dt/reps*1e3:.3f} ms per 5000 images, {2*196608*N*reps/dt/1e12:.2f} TB/s)", flush=True)
    for c in ctxs: c.close()
