#!/usr/bin/env python3
"""Developer probe: wall-clock per launch of the batch=35 stream vs stream count / event timing."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
import torch
pkg = entry.load_package(); L = pkg.lib()
for streams in (1, 2, 4, 8):
    ctx = pkg.Context(0, 256, 256, 3, 1, max_batch=1, n_slots=streams)
    ctx.resident_alloc(5000); ctx.resident_fill_synthetic(0)
    for timed in (False, True):
        for batch in (35, 140, 500):
            ctx.resident_run(5000, batch, timed=timed); ctx.sync(); ctx.reset_timing()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10):
                ctx.resident_run(5000, batch, timed=timed)
            t_enq = time.perf_counter() - t0
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            tm = ctx.sync(); n = tm["launches"]
            print(f"streams {streams} timed {int(timed)} batch {batch:4d}: wall/launch {dt/n*1e6:7.2f} us  enqueue/launch {t_enq/n*1e6:6.2f} us  "
                  f"kernel/launch {tm['kernel_ms']*1e3/n:7.2f} us  -> {50000/dt/1e6:6.2f} M img/s", flush=True)
            ctx.reset_timing()
    ctx.close()
