#!/usr/bin/env python3
"""bench.py — images/sec + achieved HBM GB/s of the blur hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload a1|hd5|a2] [--batch B]

Workloads (BASELINE.json configs):
  a1  (default) N=1: configs[1] — 5000 x 256x256x3, 3x3 blur, Approach-1 image-level dispatch, batch=35.
        N>1: configs[3] — 50 000 images sharded image-level over the N GPUs, 50000 // N per GPU (6250 at N=8),
        NO collective on the data path ("weak": per-GPU work is of the same order at every N; the stream is resident).
        A STEP is one pass over the GPU's resident stream = 143 batches at N=1 (142 x 35 images + 1 x 30,
        heterogeneous_blur.c:418-427).
  hd5 (configs[2]) 1920x1080x3, 5x5 blur, pool of 64 distinct images (796 MB in+out > MALL),
        one launch per pass — the HBM-bound rocprof point.
  a2  (configs[4]) one 8192x8192x3 image row-split over N GPUs, RCCL send/recv halo rows over
        xGMI, then each GPU blurs its band ("strong" scaling).

`--gpus N` with no launcher (WORLD_SIZE unset) starts the N ranks itself: N fresh child processes of this script with
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT set, started BEFORE this process makes any HIP or
torch.cuda call (the parent never touches the GPU); rank 0's JSON line is the parent's stdout.  Under
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` the ranks already exist and nothing is
spawned.  Fewer than N visible devices, or WORLD_SIZE != --gpus, is an error (exit != 0) — never a silent N=1 run.

Inputs are synthetic (LCG bytes, seed 0x9E3779B9 ^ image index) and RESIDENT IN HBM before the
timed region starts.  Timing: --ramp-seconds of untimed passes (the GPU reaches its sustained clock; a 20-step region
is 8 ms, shorter than the clock ramp), W warm-up steps, then exactly K steps between barrier +
torch.cuda.synchronize() on both sides, MAX over ranks.  Rank 0 prints ONE JSON line.

`roofline`: dominant kernel; achieved = algorithmic bytes per launch (2*W*H*C per image x images per launch) /
average launch duration, the duration of every launch in the timed region read from that dispatch's own start/stop
timestamps (hipExtLaunchKernel events on the launch stream — the HIP analogue of the reference's
clGetEventProfilingInfo, heterogeneous_blur.c:567-577); peak = 8 TB/s HBM3E.
`cpu_baseline`: the oracle (kind "port": scalar per-pixel restatement of gaussian_kernel.cl,
what an OpenCL CPU device executes) on all host cores, rank 0 at N=1 only, bounded sample.
At N=1 the a1 line also carries, measured after the timed region: `sustained_img_s` (>= 1 s of back-to-back passes),
`per_batch_launches` (the same stream as 143 launches per pass), and `extra` = {one_launch_5000_images, hd1080_5x5
(configs[2]), a2_8192_1gpu (configs[4] at N=1), e2e_pcie_inclusive (host buffers in -> host buffers out, batch 35
and 500; comparable to the reference's wall clock, never `value`)}.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "0")   # before any HIP initialisation; see csrc/mi_blur_api.cpp
import __graft_entry__ as entry  # noqa: E402

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
REFERENCE_IMG_S = 8568.10        # data/approach1/35_run_1.txt:79 — 320x240, i7-12700 + UHD 770 together
CONFIG3_IMAGES = 50000           # BASELINE configs[3]: 50 000 images over the node
SECONDARY_WARM_S = 0.25          # untimed launches before each secondary point: after ANY idle gap the first ~40 ms of
                                 # launches run 5-25 % slow while the clocks ramp (profiles/r02_clock_ramp.txt)


def shard_range(n_units: int, rank: int, world: int) -> tuple[int, int]:
    """Image-level sharding: rank g owns [n*g/G, n*(g+1)/G)."""
    return n_units * rank // world, n_units * (rank + 1) // world


def default_images(world: int) -> int:
    """Images per GPU per step of the a1 workload: configs[1] at N=1, the per-GPU share of configs[3] at N>1."""
    return 5000 if world <= 1 else CONFIG3_IMAGES // world


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


# ----------------------------------------------------------------------------------------------------------------
# rank spawning (parent process: no HIP / torch.cuda call is made here)
# ----------------------------------------------------------------------------------------------------------------
def spawn_ranks(n: int, argv: list[str]) -> int:
    """Start n children of this script, one rank each, and return the job's exit code.  The analogue of the reference
    driving both of its devices from one command line (heterogeneous_blur.c:482-539)."""
    import socket
    rehearsal = "MI_BLUR_BENCH_DEVICE" in os.environ          # tests: every rank on one named device, gloo barrier
    if not rehearsal:
        import torch                                           # device_count() does not initialise the GPU on this image
        have = torch.cuda.device_count()
        if have < n:
            print(f"bench.py: --gpus {n} but only {have} HIP device(s) visible; launch on a node with {n} GPUs "
                  f"(or under `python -m torch.distributed.run --nproc-per-node {n}`)", file=sys.stderr)
            return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))      # rank 0 prints the one JSON line
    rc = 0
    pending = set(range(n))
    while pending:
        for r in list(pending):
            code = procs[r].poll()
            if code is None:
                continue
            pending.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                print(f"bench.py: rank {r} exited with {code}; stopping the other ranks", file=sys.stderr)
                for q in pending:
                    procs[q].terminate()                       # exactly the children started above, by handle
        time.sleep(0.05)
    return rc


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=["a1", "hd5", "a2"], default="a1")
    ap.add_argument("--batch", type=int, default=35)
    ap.add_argument("--images", type=int, default=0,
                    help="a1: images per GPU per step (default: 5000 at N=1 = configs[1]; 50000 // N at N>1 = configs[3])")
    ap.add_argument("--dispatch", choices=["fused", "batched"], default="fused",
                    help="a1: 'fused' = one dispatch per pass whose blocks walk the batches in order and count every finished "
                         "batch in for the host (mi_blur_resident_run_fused; batch = unit of completion); 'batched' = one launch "
                         "per batch over --streams HIP streams (mi_blur_resident_run; batch = unit of dispatch)")
    ap.add_argument("--streams", type=int, default=int(os.environ.get("MI_BLUR_BENCH_STREAMS", "0")),
                    help="HIP streams the launches alternate over (default: 4 for a1 — small launches whose dispatch floors "
                         "must overlap — and 1 for hd5, whose launches fill the GPU on their own)")
    ap.add_argument("--time-every", type=int, default=32,
                    help="a1: every n-th launch of the timed region carries dispatch timestamp events")
    ap.add_argument("--ramp-seconds", type=float, default=0.5,
                    help="untimed passes before the warm-up steps, so the timed region runs at the sustained clock")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--extra", action="store_true", help="(default at N=1; kept for old command lines)")
    ap.add_argument("--no-extra", action="store_true",
                    help="N=1 a1: skip the points measured after the timed region (sustained, per-batch launches, hd5, 8192^2, e2e)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))          # parent: started the ranks, relays rank 0's line

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: start it as `python bench.py --gpus N` "
                         f"or under a launcher that creates exactly N ranks")

    import numpy as np
    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    # Rehearsal hooks (tests only): run the N>1 control path on a one-GPU box — every rank on the same device,
    # gloo instead of RCCL for the barrier / max-over-ranks (RCCL refuses two ranks on one device).
    backend = os.environ.get("MI_BLUR_BENCH_BACKEND", "nccl")
    if "MI_BLUR_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["MI_BLUR_BENCH_DEVICE"])
    elif local_rank >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} wants device {local_rank} but only {torch.cuda.device_count()} visible")
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
    per_gpu_images = args.images if args.images > 0 else default_images(world)
    do_extra = world == 1 and args.workload == "a1" and not args.no_extra

    def barrier_sync():
        if world > 1:
            dist_barrier()
        torch.cuda.synchronize()

    def frac_of(bytes_per_launch: float, launch_us: float) -> float:
        return round(bytes_per_launch / launch_us / 1e3 / HBM_PEAK_GBS, 4) if launch_us > 0 else 0.0

    # ------------------------------------------------------------------------------------------------------------
    # secondary points (N=1 only, outside the timed region).  Each is a small self-contained measurement.
    # ------------------------------------------------------------------------------------------------------------
    def point_resident(w, h, c, radius, pool, per_pass, batch, launches, label, allocations=3) -> dict:
        """`launches` back-to-back passes of a resident pool, every dispatch timestamped — on `allocations` fresh pools: where
        the buffers happen to lie moves this kernel between two levels 6 % apart (profiles/r02_allocation_placement.txt), so
        the point is the MEDIAN allocation and every allocation's figure is listed."""
        runs = []
        keep = []                                                  # held until the end so that each pool is a new allocation
        for _ in range(allocations):
            ctx = pkg.Context(local_rank, w, h, c, radius, max_batch=1, n_slots=1)
            ctx.resident_alloc(pool); ctx.resident_fill_synthetic(0)
            warm_until = time.perf_counter() + SECONDARY_WARM_S    # filling the pool left the GPU idle: ramp the clock again
            while time.perf_counter() < warm_until:
                for _ in range(10):
                    ctx.resident_run(per_pass, batch)
                ctx.sync()
            ctx.reset_timing()
            t0 = time.perf_counter()
            for _ in range(launches):
                ctx.resident_run(per_pass, batch, timed=1)
            tm = ctx.sync()
            wall = time.perf_counter() - t0
            n = max(tm["launches"], 1)
            runs.append((tm["kernel_ms"] * 1e3 / n, tm["bytes_alg"] / n, per_pass * launches / wall, int(n)))
            keep.append(ctx)
        kernel = L.mi_blur_last_kernel().decode()
        for ctx in keep:
            ctx.close()
        us, bpl, img_s, n = sorted(runs)[len(runs) // 2]
        return {"workload": label, "kernel": kernel, "launches_timed": n, "images_per_launch": batch, "launch_us": round(us, 2),
                "launch_us_by_allocation": [round(r[0], 2) for r in runs],
                "achieved_gbs": round(bpl / us / 1e3, 1) if us > 0 else 0.0, "frac": frac_of(bpl, us),
                "img_s": round(img_s, 1), "img_s_from_launch_us": round(batch / us * 1e6, 1) if us > 0 else 0.0}

    def point_a2_1gpu(steps) -> dict:
        """configs[4] at N=1: the whole 8192x8192x3 image as one band launch (no exchange partner: both edges clamp)."""
        H = Wd = 8192
        c, radius, pitch = 3, 1, 8192 * 3
        band = torch.empty(H * pitch, dtype=torch.uint8, device=dev)
        out = torch.empty(H * pitch, dtype=torch.uint8, device=dev)
        hostrows = np.empty((H, Wd, c), np.uint8)
        L.mi_blur_fill_synthetic(hostrows.ctypes.data, Wd, H, c, 0, 1, 8)
        band.copy_(torch.from_numpy(hostrows.reshape(-1)))
        stream = torch.cuda.current_stream().cuda_stream
        warm_until = time.perf_counter() + SECONDARY_WARM_S
        while time.perf_counter() < warm_until:
            for _ in range(20):
                pkg.check(L.mi_blur_enqueue_band(band.data_ptr(), out.data_ptr(), Wd, H, c, radius, 0, H, stream), "enqueue_band")
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(steps):
            pkg.check(L.mi_blur_enqueue_band(band.data_ptr(), out.data_ptr(), Wd, H, c, radius, 0, H, stream), "enqueue_band")
        e1.record()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        us = e0.elapsed_time(e1) * 1e3 / steps
        got = out.cpu().numpy()
        fnv = f"{L.mi_blur_fnv1a64(got.ctypes.data, got.size):016x}"           # tests/golden: d283787bcc5b6dfd (reference kernel)
        del band, out
        return {"workload": "one 8192x8192x3 image per step, 3x3, one GPU (configs[4] at N=1)", "kernel": L.mi_blur_last_kernel().decode(), "steps": steps,
                "step_us": round(us, 2), "achieved_gbs": round(2.0 * H * pitch / us / 1e3, 1), "frac": frac_of(2.0 * H * pitch, us),
                "img_s": round(steps / wall, 1), "out_fnv": fnv}

    def point_e2e(w, h, c, radius, nb, nbatches) -> dict:
        """Host buffers in -> host buffers out (zero-copy submits over PCIe), 4 rotating pinned buffer pairs."""
        NS = 4
        e2e = pkg.Context(local_rank, w, h, c, radius, max_batch=nb, n_slots=NS)
        nbytes = nb * h * w * c
        bufs = [(L.mi_blur_host_alloc(nbytes), L.mi_blur_host_alloc(nbytes)) for _ in range(NS)]
        for (pi, _po) in bufs:
            L.mi_blur_fill_synthetic(pi, w, h, c, 0, nb, 4)
        for i in range(2 * NS):
            e2e.submit(bufs[i % NS][0], bufs[i % NS][1], nb)
        e2e.sync(); e2e.reset_timing()
        t0e = time.perf_counter()
        for i in range(nbatches):
            e2e.submit(bufs[i % NS][0], bufs[i % NS][1], nb)
        te = e2e.sync()
        dte = time.perf_counter() - t0e
        res = {"img_s": round(nbatches * nb / dte, 0), "batch": nb, "batches": nbatches, "slots": NS,
               "zero_copy_submits": int(L.mi_blur_zero_copy_launches(e2e.h)),
               "pcie_gbs_each_way": round(nbatches * nbytes / dte / 1e9, 1),
               "h2d_ms": round(te["h2d_ms"], 2), "kernel_ms": round(te["kernel_ms"], 2), "d2h_ms": round(te["d2h_ms"], 2)}
        for (pi, po) in bufs:
            L.mi_blur_host_free(pi); L.mi_blur_host_free(po)
        e2e.close()
        return res

    fused = args.workload == "a1" and args.dispatch == "fused" and args.batch < per_gpu_images
    if args.streams <= 0:
        args.streams = 4 if (args.workload == "a1" and not fused) else 1
    extra = {}
    other_line, other_key = None, None
    sustained = None
    if args.workload in ("a1", "hd5"):
        if args.workload == "a1":
            h, w, c, radius, per_gpu, batch, pool = 256, 256, 3, 1, per_gpu_images, args.batch, per_gpu_images
            if world == 1 and per_gpu == 5000:
                which = "BASELINE configs[1]"
            elif world > 1 and per_gpu == CONFIG3_IMAGES // world:
                which = f"BASELINE configs[3]: {CONFIG3_IMAGES} images sharded image-level over the node"
            else:
                which = "custom --images"
            name = (f"{per_gpu}x256x256x3 per GPU x {world} GPU(s) = {per_gpu * world} images per step [{which}], 3x3 blur, Approach-1 "
                    f"image-level dispatch, batch={batch}, device-resident, "
                    + ("one fused dispatch per pass with per-batch completion counters" if fused else "one launch per batch"))
        else:
            h, w, c, radius, per_gpu, batch, pool = 1080, 1920, 3, 2, 64, 64, 64
            name = "1920x1080x3, 5x5 blur, pool of 64 distinct resident images, one launch per pass [BASELINE configs[2]]"
        # host threads only generate the synthetic stream; keep ranks from oversubscribing the node between them
        host_threads = max(2, min(32, len(os.sched_getaffinity(0)) // max(world, 1)))
        # The context that will issue one launch per batch on 4 streams is created FIRST when it is only the secondary
        # measurement: HIP hands hardware queues to streams in creation order, and 4 streams that do not get 4 distinct
        # queues run like 2-3 streams (6.8 instead of 10 M img/s).
        alt = None
        if do_extra and fused:
            alt = pkg.Context(local_rank, w, h, c, radius, max_batch=1, n_slots=4, n_threads=host_threads)
        ctx = pkg.Context(local_rank, w, h, c, radius, max_batch=1, n_slots=args.streams, n_threads=host_threads)   # resident runs use no staging
        ctx.resident_alloc(pool)
        first, _ = shard_range(per_gpu * world, rank, world)       # image-level sharding: rank g owns [first, first + per_gpu)
        ctx.resident_fill_synthetic(first)

        def one_pass(timed):
            if fused:
                ctx.resident_run_fused(per_gpu, batch, timed=bool(timed))
            else:
                ctx.resident_run(per_gpu, batch, timed=timed)

        # ramp: untimed passes until --ramp-seconds of wall clock have gone by (synchronising every few passes)
        ramp_passes = 0
        t_r = time.perf_counter()
        while time.perf_counter() - t_r < args.ramp_seconds:
            for _ in range(8):
                one_pass(False)
            ctx.sync()
            ramp_passes += 8
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
        dominant_kernel = L.mi_blur_last_kernel().decode()        # what the timed launches went to (the library's choice)
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
        config = {"workload": name, "images_per_gpu_per_step": per_gpu, "images_per_step": per_gpu * world, "batch": batch,
                  "launches_per_step": launches // max(K, 1), "streams": args.streams, "ramp_passes": ramp_passes,
                  "reference_published_img_s": REFERENCE_IMG_S,
                  "reference_published_on": "320x240x3, i7-12700 + UHD 770 (CPU+iGPU together)"}
        if fused:
            config["batches_counted_in_last_pass"] = ctx.resident_batches_done()

        if do_extra:
            # ---- sustained: >= 1 s of back-to-back passes of the headline form (every 16th dispatch timestamped)
            n_sus = max(K, int(1.2 / max(elapsed / K, 1e-6)))
            ctx.reset_timing()
            t_s = time.perf_counter()
            for i in range(n_sus):
                one_pass(1 if i % 16 == 0 else 0)
                if i % 256 == 255:
                    ctx.sync()                                     # bound the queue depth; harvest the events
            ts = ctx.sync()
            dts = time.perf_counter() - t_s
            sn, sb = ctx.timed_coverage()
            sustained = {"img_s": round(n_sus * per_gpu / dts, 1), "passes": n_sus, "seconds": round(dts, 3)}
            if sn and ts["kernel_ms"] > 0:
                s_us = ts["kernel_ms"] * 1e3 / sn
                sustained.update({"avg_launch_us": round(s_us, 2), "frac": frac_of(sb / sn, s_us), "launches_timed": int(sn)})

            # ---- the OTHER dispatch form beside the headline (same pool shape, own context)
            if batch < per_gpu:
                if fused:
                    # one launch per batch: 4 streams so the ~4 us per-dispatch floors overlap (timestamps on every 32nd
                    # launch), then the same launches on one stream with every dispatch timestamped (the regime
                    # rocprofv3 --stats reproduces: tracing un-overlaps the dispatches)
                    alt.resident_alloc(pool); alt.resident_fill_synthetic(0)
                    warm_until = time.perf_counter() + SECONDARY_WARM_S
                    while time.perf_counter() < warm_until:
                        alt.resident_run(per_gpu, batch, timed=False)
                        alt.sync()
                    alt.reset_timing()
                    ka = min(K, 50)
                    ta0 = time.perf_counter()
                    for _ in range(ka):
                        alt.resident_run(per_gpu, batch, timed=args.time_every)
                    ta = alt.sync()
                    dta = time.perf_counter() - ta0
                    an, ab = alt.timed_coverage()
                    other_line = {"img_s": round(ka * per_gpu / dta, 0), "launches_per_step": (per_gpu + batch - 1) // batch, "streams": 4}
                    if an and ta["kernel_ms"] > 0:
                        a_us = ta["kernel_ms"] * 1e3 / an
                        other_line.update({"overlapped_dispatch_us": round(a_us, 2), "overlapped_frac": frac_of(ab / an, a_us)})
                    alt.close()
                    ser = pkg.Context(local_rank, w, h, c, radius, max_batch=1, n_slots=1, n_threads=host_threads)
                    ser.resident_alloc(pool); ser.resident_fill_synthetic(0)
                    warm_until = time.perf_counter() + SECONDARY_WARM_S
                    while time.perf_counter() < warm_until:
                        ser.resident_run(per_gpu, batch, timed=False)
                        ser.sync()
                    ser.reset_timing()
                    for _ in range(3):
                        ser.resident_run(per_gpu, batch, timed=1)
                    ts2 = ser.sync()
                    sn2, sb2 = ser.timed_coverage()
                    if sn2 and ts2["kernel_ms"] > 0:
                        s_us = ts2["kernel_ms"] * 1e3 / sn2
                        other_line.update({"serial_dispatch_us": round(s_us, 2), "serial_frac": frac_of(sb2 / sn2, s_us)})
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
                        other_line.update({"dispatch_us": round(f_us, 1), "frac": frac_of(fb / fn, f_us)})
                    alt.close()
                    other_key = "fused_stream"

            # ---- the whole stream in ONE plain launch (the 3x3 kernel's HBM-bound point without the batch counters)
            warm_until = time.perf_counter() + SECONDARY_WARM_S
            while time.perf_counter() < warm_until:
                for _ in range(10):
                    ctx.resident_run(per_gpu, per_gpu)
                ctx.sync()
            ctx.reset_timing()
            for _ in range(20):
                ctx.resident_run(per_gpu, per_gpu, timed=1)
            t1 = ctx.sync()
            s = t1["kernel_ms"] / 1e3 / max(t1["launches"], 1)
            if s > 0:
                extra["one_launch_5000_images"] = {"launch_us": round(s * 1e6, 1), "launches_timed": int(t1["launches"]),
                                                   "achieved_gbs": round(t1["bytes_alg"] / t1["launches"] / s / 1e9, 1),
                                                   "frac": round(t1["bytes_alg"] / t1["launches"] / s / 1e9 / HBM_PEAK_GBS, 4),
                                                   "img_s": round(per_gpu / s, 0)}
        ctx.close()
        if do_extra:
            # BASELINE configs[2], configs[4] at N=1, and the PCIe-inclusive rate (host buffers in -> host buffers out)
            extra["hd1080_5x5"] = point_resident(1920, 1080, 3, 2, 64, 64, 64, 300,
                                                 "64 x 1920x1080x3 per launch, 5x5, resident pool of 64 (configs[2])")
            extra["a2_8192_1gpu"] = point_a2_1gpu(300)
            extra["e2e_pcie_inclusive"] = {"batch_35": point_e2e(256, 256, 3, 1, 35, 143 * 2),
                                           "batch_500": point_e2e(256, 256, 3, 1, 500, 40),
                                           "note": "pinned host buffers in and out, kernel works on them in place over PCIe; "
                                                   "comparable to the reference's wall clock (heterogeneous_blur.c:415,603)"}
        base_shape = (h, w, c, radius)
    else:   # a2: one 8192x8192x3 image, row-split, RCCL halo exchange
        H = Wd = 8192
        c, radius = 3, 1
        b = pkg.band_of(H, radius, rank, world)
        owned = b["row_end"] - b["row_begin"]
        ht = b["halo_top"]
        rows = owned + ht + b["halo_bottom"]
        pitch = Wd * c
        band = torch.empty(rows * pitch, dtype=torch.uint8, device=dev)
        out = torch.empty(owned * pitch, dtype=torch.uint8, device=dev)
        # synthetic content: rank g's owned rows = LCG image seeded by rank (content is irrelevant to timing)
        hostrows = np.empty((owned, Wd, c), np.uint8)
        L.mi_blur_fill_synthetic(hostrows.ctypes.data, Wd, owned, c, rank, 1, 1)
        band[ht * pitch:(ht + owned) * pitch] = torch.from_numpy(hostrows.reshape(-1)).to(dev)
        comm = C.c_void_p()
        idbuf = torch.zeros(pkg.UNIQUE_ID_BYTES, dtype=torch.uint8)
        # Rehearsal (tests, one-GPU box: MI_BLUR_BENCH_DEVICE set): RCCL refuses two ranks on one device, so the ranks
        # run the whole control path — streams, events, barriers, reductions, the overlapped step — with the exchange
        # itself left out.  Never taken on a real multi-GPU run.
        fake_exchange = world > 1 and "MI_BLUR_BENCH_DEVICE" in os.environ
        if world > 1 and not fake_exchange:
            if rank == 0:
                raw = (C.c_uint8 * pkg.UNIQUE_ID_BYTES)()
                pkg.check(L.mi_blur_comm_unique_id(raw), "comm_unique_id")
                idbuf = torch.tensor(list(raw), dtype=torch.uint8)
            idd = idbuf.to(dev) if backend == "nccl" else idbuf
            dist.broadcast(idd, src=0)
            idbuf = idd.cpu()
        idarr = (C.c_uint8 * pkg.UNIQUE_ID_BYTES)(*idbuf.tolist())
        if not fake_exchange:
            pkg.check(L.mi_blur_comm_init_rank(C.byref(comm), world, rank, idarr), "comm_init_rank")
        main_stream = torch.cuda.current_stream()
        stream = main_stream.cuda_stream

        def exchange(on_stream):
            if not fake_exchange:
                pkg.check(L.mi_blur_halo_exchange(comm, band.data_ptr(), Wd, c, owned, radius, on_stream), "halo_exchange")

        def blur_rows(y0, y1, dst_off):
            pkg.check(L.mi_blur_enqueue_band(band.data_ptr(), out.data_ptr() + dst_off, Wd, rows, c, radius, y0, y1, stream), "enqueue_band")

        def step():
            exchange(stream)
            blur_rows(ht, ht + owned, 0)

        # untimed ramp past the ~40 ms clock ramp that follows any idle gap.  A FIXED step count, the same on every rank:
        # each step holds a send/recv pair, so ranks must not decide by their own clocks how many to run.
        ramp_steps = int(args.ramp_seconds * 1e6 / 100.0)
        for i in range(ramp_steps):
            step()
            if i % 64 == 63:
                torch.cuda.synchronize()
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
        dominant_kernel = L.mi_blur_last_kernel().decode()
        elapsed = aggregate_max(local, dist if world > 1 else None, dev if backend == "nccl" else None)
        units = K
        value = units / elapsed
        scaling = "strong"
        launches = timed_n = K
        bytes_per_launch = 2.0 * owned * pitch
        avg_launch_s = ev0.elapsed_time(ev1) / 1e3 / K          # exchange + kernel on this rank's stream
        timing_src = "stream events around halo exchange + band kernel"

        # ---- per-step decomposition (outside the timed region): exchange vs band kernel on this rank's stream, and the
        # same step with the exchange on a stream of its own, hidden behind the interior rows (edge rows follow the halos)
        n_i = max(1, min(K, 50))
        ea = [torch.cuda.Event(enable_timing=True) for _ in range(3 * n_i)]
        barrier_sync()
        for i in range(n_i):
            ea[3 * i].record()
            exchange(stream)
            ea[3 * i + 1].record()
            blur_rows(ht, ht + owned, 0)
            ea[3 * i + 2].record()
        torch.cuda.synchronize()
        x_us = sum(ea[3 * i].elapsed_time(ea[3 * i + 1]) for i in range(n_i)) * 1e3 / n_i
        k_us = sum(ea[3 * i + 1].elapsed_time(ea[3 * i + 2]) for i in range(n_i)) * 1e3 / n_i
        decomp = {"instrumented_steps": n_i, "halo_exchange_us": round(x_us, 2), "band_kernel_us": round(k_us, 2),
                  "band_kernel_frac": frac_of(bytes_per_launch, k_us)}
        if world > 1 and owned > 2 * radius:
            xs = torch.cuda.Stream(device=dev)
            ev_done, ev_halo = torch.cuda.Event(), torch.cuda.Event()

            def step_overlapped():
                xs.wait_event(ev_done)                                     # the previous step has read its halo rows
                exchange(xs.cuda_stream)
                ev_halo.record(xs)
                blur_rows(ht + radius, ht + owned - radius, radius * pitch)    # interior: reads no halo row
                main_stream.wait_event(ev_halo)
                blur_rows(ht, ht + radius, 0)
                blur_rows(ht + owned - radius, ht + owned, (owned - radius) * pitch)
                ev_done.record(main_stream)

            pair = []
            for fn in (step, step_overlapped):
                ev_done.record(main_stream)
                fn()
                barrier_sync()
                t_p = time.perf_counter()
                for _ in range(n_i):
                    fn()
                torch.cuda.synchronize()
                pair.append(aggregate_max(time.perf_counter() - t_p, dist, dev if backend == "nccl" else None) * 1e6 / n_i)
                if world > 1:
                    dist_barrier()
            decomp.update({"step_us_plain": round(pair[0], 2), "step_us_overlapped": round(pair[1], 2)})
        if world > 1:      # the slowest rank's figures (rank 0 and the last rank have one neighbour only)
            t = torch.tensor([decomp["halo_exchange_us"], decomp["band_kernel_us"]], dtype=torch.float64,
                             device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            decomp["halo_exchange_us_max_over_ranks"], decomp["band_kernel_us_max_over_ranks"] = round(float(t[0]), 2), round(float(t[1]), 2)
        if not fake_exchange:
            L.mi_blur_comm_destroy(comm)
        config = {"workload": f"one 8192x8192x3 image per step, 3x3, row-split over {world} GPU(s), RCCL halo exchange [BASELINE configs[4]]",
                  "rows_per_gpu": owned, "halo_bytes_per_neighbour": radius * pitch, "ramp_steps": ramp_steps, "step_decomposition": decomp}
        if fake_exchange:
            config["rehearsal"] = "halo exchange left out (two ranks share one device); control path only"
        base_shape = (H, Wd, c, radius)

    # which committed PMC run (profiles/traffic.json) matches this command's dominant kernel and launch shape
    traffic_key = args.workload
    if args.workload == "a1" and not fused:
        traffic_key = "a1_one_launch" if args.batch >= per_gpu_images else ("a1_serial" if args.streams == 1 else "a1_batched")
    achieved = bytes_per_launch / avg_launch_s / 1e9
    roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": load_traffic(traffic_key) if (args.workload != "a1" or per_gpu_images == 5000) else None,
                "kernel": dominant_kernel, "algorithmic_bytes_per_launch": round(bytes_per_launch),
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
        if sustained:
            line["sustained_img_s"] = sustained["img_s"]
            line["sustained"] = sustained
        if other_line:
            line[other_key] = other_line
        if extra:
            line["extra"] = extra
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
