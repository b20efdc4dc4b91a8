import sys, time
sys.path.insert(0, "/root/repo")
import __graft_entry__ as entry
import torch
pkg = entry.load_package(); L = pkg.lib()
for n in (5000, 50000):
    ctx = pkg.Context(0, 256, 256, 3, 1, max_batch=1, n_slots=4)
    ctx.resident_alloc(n); ctx.resident_fill_synthetic(0)
    ctx.resident_run_fused(min(n, 70), 35); ctx.sync()
    for rep in range(4):
        ctx.reset_timing()
        t0 = time.perf_counter(); ctx.resident_run_fused(n, 35, timed=True); t1 = time.perf_counter(); tm = ctx.sync(); t2 = time.perf_counter()
        print(n, "rep", rep, f"enqueue {1e3*(t1-t0):.2f} ms  sync {1e3*(t2-t1):.2f} ms  kernel {tm['kernel_ms']:.2f} ms")
    ctx.close()
ctx = pkg.Context(0, 256, 256, 3, 1, max_batch=1, n_slots=4)
ctx.resident_alloc(50000); ctx.resident_fill_synthetic(0)
ctx.resident_run_fused(50000, 35); ctx.sync()
for rep in range(4):
    t0 = time.perf_counter(); n = ctx.resident_batches_done(); t1 = time.perf_counter()
    print("batches_done", n, f"{1e3*(t1-t0):.3f} ms")
