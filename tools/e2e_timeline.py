#!/usr/bin/env python3
"""Developer probe: a short end-to-end run (pinned host buffers in -> out, zero-copy submits) meant to be traced:

    rocprofv3 --hip-trace --kernel-trace --output-format csv -d gpurun_out/e2e_trace -- python3 tools/e2e_timeline.py [batch] [slots] [submits]

Warm-up submits first (so code-object load and clock ramp stay out of the traced window), then `submits` timed ones with a
host timestamp around every mi_blur_submit call.  Prints the host-side split; tools/e2e_timeline_report.py folds the
rocprofv3 CSVs into profiles/r03_e2e_timeline.md."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402


def main():
    import torch  # noqa: F401  (loads the HIP runtime first, see hoipe_amd.lib)
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 35
    ns = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    n_sub = int(sys.argv[3]) if len(sys.argv) > 3 else 60
    pkg = entry.load_package()
    L = pkg.lib()
    L.mi_blur_bind_thread_to_device(0)                         # as the hosts' feeder threads do: this thread, the helper threads it
                                                                # starts and the pages it touches first stay on the GPU's socket
    opts = [a.split("=") for a in sys.argv[4:]]                  # key=value pairs for mi_blur_set_option; arena=MiB is the probe's own
    arena_mib = 0
    shape = (256, 256)
    for k, v in opts:
        if k == "arena":
            arena_mib = int(v)
        elif k in ("radius", "pageable", "thp"):                   # pageable=1: ordinary (malloc'd) caller buffers, the reference's own kind
            pass
        elif k == "shape":                                 # shape=WxH (probe's own): e.g. a frame whose rows are not a multiple of 16 bytes
            shape = tuple(int(x) for x in v.split("x"))
        else:
            pkg.check(L.mi_blur_set_option(k.encode(), int(v)), k)
    w, h, c, r = shape[0], shape[1], 3, int(dict(opts).get("radius", 1))
    nbytes = nb * h * w * c
    arena = None
    if arena_mib:       # ONE pinned allocation of at least arena MiB, the batch buffers carved out of it at 2 MiB-aligned offsets
        step = (nbytes + (2 << 20) - 1) // (2 << 20) * (2 << 20)
        total = max(arena_mib << 20, 2 * ns * step + (2 << 20))
        arena = L.mi_blur_host_alloc(total)
        base = (arena + (2 << 20) - 1) // (2 << 20) * (2 << 20)
        bufs = [(base + (2 * i) * step, base + (2 * i + 1) * step) for i in range(ns)]
    elif int(dict(opts).get("thp", 0)):                     # thp=1: anonymous memory with transparent huge pages asked for, registered in place
        import mmap
        import ctypes
        keep = []
        bufs = []
        step = (nbytes + (2 << 20) - 1) // (2 << 20) * (2 << 20)
        for _ in range(ns):
            pair = []
            for _side in range(2):
                m = mmap.mmap(-1, step + (2 << 20))
                addr = ctypes.addressof(ctypes.c_char.from_buffer(m))
                base = (addr + (2 << 20) - 1) // (2 << 20) * (2 << 20)
                m.madvise(mmap.MADV_HUGEPAGE)
                ctypes.memset(base, 1, step)                 # touch: pages (huge, if the kernel grants them) exist before they are pinned
                pkg.check(L.mi_blur_host_register(base, nbytes))
                keep.append(m)
                pair.append(base)
            bufs.append(tuple(pair))
        try:
            thp = [l for l in open("/proc/meminfo") if l.startswith("AnonHugePages")][0].split()[1]
            print(f"AnonHugePages now {thp} kB; /sys/kernel/mm/transparent_hugepage/enabled: {open('/sys/kernel/mm/transparent_hugepage/enabled').read().strip()}", flush=True)
        except Exception as e:
            print("thp state:", e)
    elif int(dict(opts).get("pageable", 0)):
        import numpy as np
        keep = [(np.zeros(nbytes, np.uint8), np.zeros(nbytes, np.uint8)) for _ in range(ns)]
        bufs = [(a.ctypes.data, b.ctypes.data) for a, b in keep]
        if int(dict(opts).get("pageable", 0)) == 2:           # pageable=2: the same malloc'd buffers, registered in place once
            for (pi, po) in bufs:
                pkg.check(L.mi_blur_host_register(pi, nbytes)); pkg.check(L.mi_blur_host_register(po, nbytes))
    else:
        bufs = [(L.mi_blur_host_alloc(nbytes), L.mi_blur_host_alloc(nbytes)) for _ in range(ns)]
    for (pi, _po) in bufs:
        L.mi_blur_fill_synthetic(pi, w, h, c, 0, nb, 4)
    ctx = pkg.Context(0, w, h, c, r, max_batch=nb, n_slots=ns)
    t_end = time.perf_counter() + 0.3
    i = 0
    while time.perf_counter() < t_end:                  # warm: code objects, clocks
        ctx.submit(bufs[i % ns][0], bufs[i % ns][1], nb)
        i += 1
    ctx.sync()
    ctx.reset_timing()
    calls = []
    t0 = time.perf_counter()
    for i in range(n_sub):
        a = time.perf_counter()
        ctx.submit(bufs[i % ns][0], bufs[i % ns][1], nb)
        calls.append(time.perf_counter() - a)
    tm = ctx.sync()
    dt = time.perf_counter() - t0
    calls_us = sorted(x * 1e6 for x in calls)
    print(f"batch {nb} slots {ns} {' '.join('='.join(o) for o in opts)}: {n_sub} submits in {dt * 1e3:.2f} ms = {dt / n_sub * 1e6:.1f} us per submit, "
          f"{n_sub * nb / dt:.0f} img/s, {n_sub * nbytes / dt / 1e9:.1f} GB/s each way; "
          f"mi_blur_submit call: median {calls_us[len(calls_us) // 2]:.1f} us, min {calls_us[0]:.1f}, max {calls_us[-1]:.1f}; "
          f"kernel bucket {tm['kernel_ms']:.2f} ms (union of dispatch intervals) = {tm['kernel_ms'] / (dt * 1e3) * 100:.0f} % of the wall clock",
          flush=True)
    ctx.close()
    if arena:
        L.mi_blur_host_free(arena)
    elif int(dict(opts).get("pageable", 0)) or int(dict(opts).get("thp", 0)):
        pass
    else:
        for (pi, po) in bufs:
            L.mi_blur_host_free(pi)
            L.mi_blur_host_free(po)


if __name__ == "__main__":
    main()
