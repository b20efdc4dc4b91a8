#!/usr/bin/env python3
"""bench.py — images/sec + achieved HBM GB/s of the blur hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload a1|hd5|a2] [--batch B]

Workloads (BASELINE.json configs):
  a1  (default, configs[1]; per-GPU share of configs[3] at N>1)
        5000 x 256x256x3 images per GPU, 3x3 blur, Approach-1 image-level dispatch, batch=35:
        a STEP is one pass over the GPU's resident 5000-image stream = 143 launches
        (142 x 35 images + 1 x 30, heterogeneous_blur.c:418-427).  Images are independent, so
        N GPUs shard the stream with NO collective ("weak" scaling: 5000 images per GPU).
  hd5 (configs[2]) 1920x1080x3, 5x5 blur, pool of 64 distinct images (796 MB in+out > MALL),
        one launch per pass — the HBM-bound rocprof point.
  a2  (configs[4]) one 8192x8192x3 image row-split over N GPUs, RCCL send/recv halo rows over
        xGMI, then each GPU blurs its band ("strong" scaling).

Inputs are synthetic (LCG bytes, seed 0x9E3779B9 ^ image index) and RESIDENT IN HBM before the
timed region starts.  Timing: W warm-up steps, then exactly K steps between barrier +
torch.cuda.synchronize() on both sides, MAX over ranks.  Rank 0 prints ONE JSON line.

`roofline`: dominant kernel = blur_tiled_kernel; achieved = algorithmic bytes per launch
(2*W*H*C per image x images per launch) / average launch duration, where the duration of every
launch in the timed region is read from that dispatch's own start/stop timestamps
(hipExtLaunchKernel events on the launch stream — the HIP analogue of the reference's
clGetEventProfilingInfo, heterogeneous_blur.c:567-577); peak = 8 TB/s HBM3E.
`cpu_baseline`: the oracle (kind "port": scalar per-pixel restatement of gaussian_kernel.cl,
what an OpenCL CPU device executes) on all host cores, rank 0 at N=1 only, bounded sample.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "0")   # before any HIP initialisation; see csrc/mi_blur_api.cpp
import __graft_entry__ as entry  # noqa: E402

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
REFERENCE_IMG_S = 8568.10        # data/approach1/35_run_1.txt:79 — 320x240, i7-12700 + UHD 770 together


def shard_range(n_units: int, rank: int, world: int) -> tuple[int, int]:
    """Image-level sharding: rank g owns [n*g/G, n*(g+1)/G)."""
    return n_units * rank // world, n_units * (rank + 1) // world


def aggregate_max(local_seconds: float, dist, device=None) -> float:
    """MAX over ranks of the timed region."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return local_seconds
    import torch
    t = torch.tensor([local_seconds], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def load_traffic(workload: str):
    """HBM bytes per launch from the committed PMC runs (profiles/traffic.json), or None."""
    p = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(p) as f:
            return json.load(f).get(workload, {}).get("hbm_bytes_per_launch")
    except Exception:
        return None


def effective_cpus() -> int:
    """CPUs this process can actually run on: the affinity mask capped by the cgroup CPU quota (a 1-GPU box shows all
    of the host's hardware threads in the mask but grants a 16-CPU share)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]             # cgroup v2
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period) + 0.5)))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())                 # cgroup v1
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and per > 0:
                n = min(n, max(1, int(q / per + 0.5)))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline(n_target: int, h: int, w: int, c: int, radius: int) -> dict:
    """Oracle (scalar per-pixel restatement) on the host cores this process may use; ~10 s of wall clock."""
    O = entry.load_oracle()
    cores = effective_cpus()
    lib = O.lib()
    probe = O.lcg_stream(8, h, w, c)
    out = probe.copy()
    t0 = time.perf_counter()
    lib.oracle_blur_batch(probe.ctypes.data, out.ctypes.data, w, h, c, radius, 8)
    per_img = (time.perf_counter() - t0) / 8
    n = int(min(n_target, max(cores * 4, 12.0 * cores / per_img)))     # ~12 s wall
    import numpy as np
    src = np.empty((n, h, w, c), np.uint8)
    pkg = entry.load_package()
    pkg.lib().mi_blur_fill_synthetic(src.ctypes.data, w, h, c, 0, n, cores)
    dst = np.empty_like(src)
    dst[:] = 0                                                          # touch pages outside the timed part
    bounds = [(n * i // cores, n * (i + 1) // cores) for i in range(cores)]

    isz = h * w * c

    def work(b, e):
        if e > b:
            lib.oracle_blur_batch(src[b:e].ctypes.data, dst[b:e].ctypes.data, w, h, c, radius, e - b)

    t0 = time.perf_counter()
    work(0, min(n, 4))                                                    # steady-state cost (the probe above paid first-touch)
    per_img = max(per_img, (time.perf_counter() - t0) / min(n, 4))
    reps = int(max(1, min(64, round(8.0 * cores / (per_img * n)))))       # ~8-12 s of wall clock whatever the core count

    def work_reps(b, e):
        for _ in range(reps):
            work(b, e)

    th = [threading.Thread(target=work_reps, args=be) for be in bounds]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    res = {"value": round(n * reps / dt, 2), "unit": "img/s", "cores": cores, "kind": "port",
           "sample": f"{reps} pass(es) over {n} of the same synthetic {w}x{h}x{c} images, radius {radius}, "
                     f"oracle_blur_batch (scalar per-pixel C restatement) on {cores} threads "
                     f"(affinity mask {len(os.sched_getaffinity(0))}, cgroup quota applied), {dt:.1f} s wall"}
    # Beside it, when the build container's oracle/_ref travelled here: the UNMODIFIED reference kernel
    # (gaussian_kernel.cl compiled for x86-64 by oracle/Makefile, one call per work-item of the padded NDRange; 3x3 only),
    # on a smaller sample.  It is slower than the port (its min/max/get_global_id are out-of-line calls), so the port
    # stays the quoted baseline: the conservative one.
    if radius == 1 and O.ref_available():
        rlib = O.ref()
        m = int(min(n, max(cores, 3.0 * cores / (per_img * 3.0))))

        def rwork(b, e):
            for i in range(b, e):
                rlib.ref_gaussian_blur(src.ctypes.data + i * isz, dst.ctypes.data + i * isz, w, h, c)

        th = [threading.Thread(target=rwork, args=(m * i // cores, m * (i + 1) // cores)) for i in range(cores)]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        rdt = time.perf_counter() - t0
        res["reference_kernel"] = {"value": round(m / rdt, 2), "unit": "img/s", "cores": cores,
                                   "sample": f"{m} images, unmodified gaussian_kernel.cl -> x86-64 (oracle/_ref), {rdt:.1f} s wall"}
    return res


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=["a1", "hd5", "a2"], default="a1")
    ap.add_argument("--batch", type=int, default=35)
    ap.add_argument("--images", type=int, default=5000, help="images per GPU per step (a1)")
    ap.add_argument("--dispatch", choices=["fused", "batched"], default="fused",
                    help="a1: 'fused' = one dispatch per pass whose blocks walk the batches in order and count every finished "
                         "batch in for the host (mi_blur_resident_run_fused; batch = unit of completion); 'batched' = one launch "
                         "per batch over --streams HIP streams (mi_blur_resident_run; batch = unit of dispatch)")
    ap.add_argument("--streams", type=int, default=int(os.environ.get("MI_BLUR_BENCH_STREAMS", "0")),
                    help="HIP streams the launches alternate over (default: 4 for a1 — small launches whose dispatch floors "
                         "must overlap — and 1 for hd5, whose launches fill the GPU on their own)")
    ap.add_argument("--time-every", type=int, default=32,
                    help="a1: every n-th launch of the timed region carries dispatch timestamp events")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--extra", action="store_true",
                    help="also measure (after the timed region) the one-launch, PCIe-inclusive and 1080p 5x5 points")
    ap.add_argument("--no-extra", action="store_true", help="(default; kept for old command lines)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    # Rehearsal hooks (tests only): run the N>1 control path on a one-GPU box — every rank on the same device,
    # gloo instead of RCCL for the barrier / max-over-ranks (RCCL refuses two ranks on one device).
    backend = os.environ.get("MI_BLUR_BENCH_BACKEND", "nccl")
    if "MI_BLUR_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["MI_BLUR_BENCH_DEVICE"])
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    def dist_barrier():
        if backend == "nccl":
            dist.barrier(device_ids=[local_rank])
        else:
            dist.barrier()

    pkg = entry.load_package()
    L = pkg.lib()
    K, W = args.steps, args.warmup

    def barrier_sync():
        if world > 1:
            dist_barrier()
        torch.cuda.synchronize()

    fused = args.workload == "a1" and args.dispatch == "fused" and args.batch < args.images
    if args.streams <= 0:
        args.streams = 4 if (args.workload == "a1" and not fused) else 1
    extra = {}
    other_line, other_key = None, None
    if args.workload in ("a1", "hd5"):
        if args.workload == "a1":
            h, w, c, radius, per_gpu, batch, pool = 256, 256, 3, 1, args.images, args.batch, args.images
            name = (f"{per_gpu}x256x256x3 per GPU, 3x3 blur, Approach-1 image-level dispatch, batch={batch}, device-resident, "
                    + ("one fused dispatch per pass with per-batch completion counters" if fused else "one launch per batch"))
        else:
            h, w, c, radius, per_gpu, batch, pool = 1080, 1920, 3, 2, 64, 64, 64
            name = "1920x1080x3, 5x5 blur, pool of 64 distinct resident images, one launch per pass"
        # host threads only generate the synthetic stream; keep ranks from oversubscribing the node between them
        host_threads = max(2, min(32, len(os.sched_getaffinity(0)) // max(world, 1)))
        # The context that will issue one launch per batch on 4 streams is created FIRST when it is only the secondary
        # measurement: HIP hands hardware queues to streams in creation order, and 4 streams that do not get 4 distinct
        # queues run like 2-3 streams (6.8 instead of 10 M img/s).
        alt = None
        if world == 1 and fused:
            alt = pkg.Context(local_rank, w, h, c, radius, max_batch=1, n_slots=4, n_threads=host_threads)
        ctx = pkg.Context(local_rank, w, h, c, radius, max_batch=1, n_slots=args.streams, n_threads=host_threads)   # resident runs use no staging
        ctx.resident_alloc(pool)
        ctx.resident_fill_synthetic(rank * per_gpu)
        def one_pass(timed):
            if fused:
                ctx.resident_run_fused(per_gpu, batch, timed=bool(timed))
            else:
                ctx.resident_run(per_gpu, batch, timed=timed)

        for _ in range(W):
            one_pass(False)
        ctx.sync()
        ctx.reset_timing()
        barrier_sync()
        t0 = time.perf_counter()
        time_every = args.time_every if (args.workload == "a1" and not fused) else 1
        for _ in range(K):
            one_pass(time_every)
        torch.cuda.synchronize()
        local = time.perf_counter() - t0          # this rank's K steps, from the common (barrier + sync) start to its own drain;
        if world > 1:                             # the closing barrier + sync follow, then MAX over ranks: the job's time is
            dist_barrier()                        # the slowest rank's, without the barrier's own latency added to every rank
            torch.cuda.synchronize()
        tm = ctx.sync()
        elapsed = aggregate_max(local, dist if world > 1 else None, dev if backend == "nccl" else None)
        units = per_gpu * world * K
        value = units / elapsed
        scaling = "weak"
        launches, (timed_n, timed_bytes) = tm["launches"], ctx.timed_coverage()
        if tm["kernel_ms"] > 0 and timed_n > 0:
            bytes_per_launch = timed_bytes / timed_n
            avg_launch_s = tm["kernel_ms"] / 1e3 / timed_n
            timing_src = f"dispatch start/stop timestamp events on every {time_every}-th launch of the timed region"
        else:                                  # should not happen; keep the line honest if it does
            bytes_per_launch = tm["bytes_alg"] / max(launches, 1)
            avg_launch_s = local / launches
            timed_n = launches
            timing_src = "wall clock (dispatch events unavailable)"
        config = {"workload": name, "images_per_gpu_per_step": per_gpu, "batch": batch,
                  "launches_per_step": launches // max(K, 1), "streams": args.streams,
                  "reference_published_img_s": REFERENCE_IMG_S,
                  "reference_published_on": "320x240x3, i7-12700 + UHD 770 (CPU+iGPU together)"}

        # ---- N=1, a1: the OTHER dispatch form beside the headline, outside the timed region (same pool shape, own context)
        if world == 1 and args.workload == "a1" and batch < per_gpu:
            if fused:
                # one launch per batch: 4 streams so the ~4 us per-dispatch floors overlap (timestamps on every 32nd
                # launch), then the same launches on one stream with every dispatch timestamped (the regime
                # rocprofv3 --stats reproduces: tracing un-overlaps the dispatches)
                alt.resident_alloc(pool); alt.resident_fill_synthetic(0)
                for _ in range(max(W, 3)):
                    alt.resident_run(per_gpu, batch, timed=False)
                alt.sync(); alt.reset_timing()
                ta0 = time.perf_counter()
                for _ in range(K):
                    alt.resident_run(per_gpu, batch, timed=args.time_every)
                ta = alt.sync()
                dta = time.perf_counter() - ta0
                an, ab = alt.timed_coverage()
                other_line = {"img_s": round(K * per_gpu / dta, 0), "launches_per_step": (per_gpu + batch - 1) // batch, "streams": 4}
                if an and ta["kernel_ms"] > 0:
                    a_us = ta["kernel_ms"] * 1e3 / an
                    other_line.update({"overlapped_dispatch_us": round(a_us, 2), "overlapped_frac": round(ab / an / a_us / 1e3 / HBM_PEAK_GBS, 4)})
                alt.close()
                ser = pkg.Context(local_rank, w, h, c, radius, max_batch=1, n_slots=1, n_threads=host_threads)
                ser.resident_alloc(pool); ser.resident_fill_synthetic(0)
                ser.resident_run(per_gpu, batch, timed=False); ser.sync(); ser.reset_timing()
                for _ in range(3):
                    ser.resident_run(per_gpu, batch, timed=1)
                ts = ser.sync()
                sn, sb = ser.timed_coverage()
                if sn and ts["kernel_ms"] > 0:
                    s_us = ts["kernel_ms"] * 1e3 / sn
                    other_line.update({"serial_dispatch_us": round(s_us, 2), "serial_frac": round(sb / sn / s_us / 1e3 / HBM_PEAK_GBS, 4)})
                ser.close()
                other_key = "per_batch_launches"
            else:
                alt = pkg.Context(local_rank, w, h, c, radius, max_batch=1, n_slots=1, n_threads=host_threads)
                alt.resident_alloc(pool); alt.resident_fill_synthetic(0)
                alt.resident_run_fused(per_gpu, batch); alt.sync(); alt.reset_timing()
                tf0 = time.perf_counter()
                for _ in range(20):
                    alt.resident_run_fused(per_gpu, batch, timed=True)
                tf = alt.sync()
                dtf = time.perf_counter() - tf0
                fn, fb = alt.timed_coverage()
                other_line = {"img_s": round(20 * per_gpu / dtf, 0), "launches_per_step": 1, "batch": batch,
                              "batches_counted_in": alt.resident_batches_done()}
                if fn and tf["kernel_ms"] > 0:
                    f_us = tf["kernel_ms"] * 1e3 / fn
                    other_line.update({"dispatch_us": round(f_us, 1), "frac": round(fb / fn / f_us / 1e3 / HBM_PEAK_GBS, 4)})
                alt.close()
                other_key = "fused_stream"
        if fused:
            config["batches_counted_in_last_pass"] = ctx.resident_batches_done()

        # ---- extras at N=1: PCIe-inclusive rate, one-launch (HBM-bound) point, hd5 point
        if world == 1 and args.extra and args.workload == "a1":
            ctx.reset_timing()
            for _ in range(3):
                ctx.resident_run(per_gpu, per_gpu, timed=1)             # whole stream in ONE launch
                t1 = ctx.sync()                                         # one at a time: no overlap between them
            s = t1["kernel_ms"] / 1e3 / max(t1["launches"], 1)
            if s > 0:
                extra["one_launch_5000_images"] = {"launch_us": round(s * 1e6, 1),
                                                   "achieved_gbs": round(t1["bytes_alg"] / t1["launches"] / s / 1e9, 1),
                                                   "frac_of_8TBs": round(t1["bytes_alg"] / t1["launches"] / s / 1e9 / HBM_PEAK_GBS, 4),
                                                   "img_s": round(per_gpu / s, 0)}
            # e2e: pinned host buffers in, pinned host buffers out (zero-copy submits: the kernel reads and writes them
            # in place over PCIe), 3 rotating buffer pairs (PCIe-inclusive; never `value`)
            nb = 35
            e2e = pkg.Context(local_rank, w, h, c, radius, max_batch=nb, n_slots=3)
            nbytes = nb * h * w * c
            bufs = [(L.mi_blur_host_alloc(nbytes), L.mi_blur_host_alloc(nbytes)) for _ in range(3)]
            for (pi, _po) in bufs:
                L.mi_blur_fill_synthetic(pi, w, h, c, 0, nb, 4)
            for i in range(6):
                e2e.submit(bufs[i % 3][0], bufs[i % 3][1], nb)
            e2e.sync(); e2e.reset_timing()
            nbatches = 143 * 2
            t0e = time.perf_counter()
            for i in range(nbatches):
                e2e.submit(bufs[i % 3][0], bufs[i % 3][1], nb)
            te = e2e.sync()
            dte = time.perf_counter() - t0e
            extra["e2e_pcie_inclusive"] = {"img_s": round(nbatches * nb / dte, 0), "batch": nb, "slots": 3,
                                           "zero_copy_submits": int(L.mi_blur_zero_copy_launches(e2e.h)),
                                           "h2d_ms": round(te["h2d_ms"], 2), "kernel_ms": round(te["kernel_ms"], 2),
                                           "d2h_ms": round(te["d2h_ms"], 2)}
            for (pi, po) in bufs:
                L.mi_blur_host_free(pi); L.mi_blur_host_free(po)
            e2e.close()
            # hd5: configs[2]
            hd = pkg.Context(local_rank, 1920, 1080, 3, 2, max_batch=1, n_slots=1)
            hd.resident_alloc(64); hd.resident_fill_synthetic(0)
            hd.resident_run(64, 64); hd.sync(); hd.reset_timing()
            for _ in range(5):
                hd.resident_run(64, 64, timed=1)
                th = hd.sync()
            s = th["kernel_ms"] / 1e3 / max(th["launches"], 1)
            if s > 0:
                extra["hd1080_5x5"] = {"launch_us": round(s * 1e6, 1), "images_per_launch": 64,
                                       "achieved_gbs": round(th["bytes_alg"] / th["launches"] / s / 1e9, 1),
                                       "frac_of_8TBs": round(th["bytes_alg"] / th["launches"] / s / 1e9 / HBM_PEAK_GBS, 4),
                                       "img_s": round(64 / s, 0)}
            hd.close()
        ctx.close()
        base_shape = (h, w, c, radius)
    else:   # a2: one 8192x8192x3 image, row-split, RCCL halo exchange
        H = Wd = 8192
        c, radius = 3, 1
        b = pkg.band_of(H, radius, rank, world)
        owned = b["row_end"] - b["row_begin"]
        rows = owned + b["halo_top"] + b["halo_bottom"]
        pitch = Wd * c
        band = torch.empty(rows * pitch, dtype=torch.uint8, device=dev)
        out = torch.empty(owned * pitch, dtype=torch.uint8, device=dev)
        # synthetic content: rank g's owned rows = LCG image seeded by rank (content is irrelevant to timing)
        hostrows = np.empty((owned, Wd, c), np.uint8)
        L.mi_blur_fill_synthetic(hostrows.ctypes.data, Wd, owned, c, rank, 1, 1)
        band[b["halo_top"] * pitch:(b["halo_top"] + owned) * pitch] = torch.from_numpy(hostrows.reshape(-1)).to(dev)
        comm = C.c_void_p()
        idbuf = torch.zeros(pkg.UNIQUE_ID_BYTES, dtype=torch.uint8)
        if world > 1:
            if rank == 0:
                raw = (C.c_uint8 * pkg.UNIQUE_ID_BYTES)()
                pkg.check(L.mi_blur_comm_unique_id(raw), "comm_unique_id")
                idbuf = torch.tensor(list(raw), dtype=torch.uint8)
            idd = idbuf.to(dev)
            dist.broadcast(idd, src=0)
            idbuf = idd.cpu()
        idarr = (C.c_uint8 * pkg.UNIQUE_ID_BYTES)(*idbuf.tolist())
        pkg.check(L.mi_blur_comm_init_rank(C.byref(comm), world, rank, idarr), "comm_init_rank")
        stream = torch.cuda.current_stream().cuda_stream

        def step():
            pkg.check(L.mi_blur_halo_exchange(comm, band.data_ptr(), Wd, c, owned, radius, stream), "halo_exchange")
            pkg.check(L.mi_blur_enqueue_band(band.data_ptr(), out.data_ptr(), Wd, rows, c, radius,
                                             b["halo_top"], b["halo_top"] + owned, stream), "enqueue_band")
        for _ in range(W):
            step()
        barrier_sync()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record()
        for _ in range(K):
            step()
        ev1.record()
        torch.cuda.synchronize()
        local = time.perf_counter() - t0          # this rank's K steps, from the common (barrier + sync) start to its own drain;
        if world > 1:                             # the closing barrier + sync follow, then MAX over ranks: the job's time is
            dist_barrier()                        # the slowest rank's, without the barrier's own latency added to every rank
            torch.cuda.synchronize()
        elapsed = aggregate_max(local, dist if world > 1 else None, dev if backend == "nccl" else None)
        L.mi_blur_comm_destroy(comm)
        units = K
        value = units / elapsed
        scaling = "strong"
        launches = timed_n = K
        bytes_per_launch = 2.0 * owned * pitch
        avg_launch_s = ev0.elapsed_time(ev1) / 1e3 / K          # exchange + kernel on this rank's stream
        timing_src = "stream events around halo exchange + band kernel"
        config = {"workload": f"one 8192x8192x3 image per step, 3x3, row-split over {world} GPU(s), RCCL halo exchange",
                  "rows_per_gpu": owned, "halo_bytes_per_neighbour": radius * pitch}
        base_shape = (H, Wd, c, radius)

    # which committed PMC run (profiles/traffic.json) matches this command's dominant kernel and launch shape
    traffic_key = args.workload
    if args.workload == "a1" and not fused:
        traffic_key = "a1_one_launch" if args.batch >= args.images else ("a1_serial" if args.streams == 1 else "a1_batched")
    achieved = bytes_per_launch / avg_launch_s / 1e9
    roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": load_traffic(traffic_key),
                "kernel": "blur_fused_kernel" if fused else "blur_tiled_kernel", "algorithmic_bytes_per_launch": round(bytes_per_launch),
                "avg_launch_us": round(avg_launch_s * 1e6, 2), "launches_timed": timed_n, "timing": timing_src}
    if args.workload != "a2":
        # With one launch per batch the launches of independent batches overlap on the GPU (one HIP stream each), so a
        # dispatch's own duration is longer than its share of the step: the whole-step figure is reported beside it.
        step_bytes = 2.0 * h * w * c * per_gpu
        roofline["concurrent_streams"] = args.streams
        roofline["whole_step_gbs_per_gpu"] = round(step_bytes * K / local / 1e9, 1)
        roofline["whole_step_frac"] = round(step_bytes * K / local / 1e9 / HBM_PEAK_GBS, 4)

    line = {"metric": "images_per_sec", "value": round(value, 1), "unit": "img/s", "n_gpus": world,
            "steps": K, "warmup": W, "ms_per_step": round(elapsed / K * 1e3, 4), "higher_is_better": True,
            "scaling": scaling, "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": config, "roofline": roofline}
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            hh, ww, cc, rr = base_shape
            if args.workload == "a2":
                hh, ww = 1024, 8192           # a band-sized slice of the same image keeps the sample bounded
            line["cpu_baseline"] = cpu_baseline(5000 if args.workload == "a1" else 64, hh, ww, cc, rr)
        if other_line:
            line[other_key] = other_line
        if extra:
            line["extra"] = extra
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
