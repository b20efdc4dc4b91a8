#!/usr/bin/env python3
"""Developer probe: dispatch duration vs images per launch (fixed launch overhead vs marginal bandwidth)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
import torch
pkg = entry.load_package(); L = pkg.lib()
for (h, w, c, r) in [(256, 256, 3, 1), (16, 16, 3, 1)]:
    ctx = pkg.Context(0, w, h, c, r, max_batch=1, n_slots=1)
    ctx.resident_alloc(5000); ctx.resident_fill_synthetic(0)
    for rpg in (4, 8):
        L.mi_blur_set_option(b"rows_per_thread", rpg)
        for batch in (1, 2, 4, 8, 16, 35, 70, 140, 280, 560):
            ctx.resident_run(batch * 8, batch, timed=0); ctx.sync(); ctx.reset_timing()
            ctx.resident_run(batch * 8, batch, timed=1)
            tm = ctx.sync()
            us = tm["kernel_ms"] * 1e3 / tm["launches"]
            print(f"{w}x{h} rpg {rpg} batch {batch:4d}: {us:8.2f} us/launch  {2*h*w*c*batch/us/1e3:8.1f} GB/s", flush=True)
    ctx.close()
