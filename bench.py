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
`per_batch_launches` (the same stream as 143 launches per pass), `batch_completion_us` (when the host SEES batch k of the
fused pass complete: first / p50 / last batch, polled during >= 24 passes — the per-batch clFinish of
heterogeneous_blur.c:538-539), `release_mode_us` (the same pass with the architectural release-ordered completion add), and
`extra` = {one_launch_5000_images, hd1080_5x5 (configs[2]), a2_8192_1gpu (configs[4] at N=1), copy_kernel_same_box (torch's
elementwise copy of one pass's bytes on this box, for scale: boxes differ), e2e_pcie_inclusive (host
buffers in -> host buffers out, batch 35 and 500 on pinned buffers and batch 35 on pageable (malloc'd) ones, the reference's own
kind; comparable to the reference's wall clock, never `value`)}.
At N>1 the a1 line carries BOTH multi-GPU configs: configs[3] is `value`; after its timed region the same ranks run
configs[4] (`extra.a2_8192_rowsplit`: the 8192x8192x3 image row-split over the N GPUs, halo rows by RCCL send/recv, the
plain step and the overlapped step both timed, `rccl_ranks` from the communicator).
`parity`: every output image of every rank's shard (a1) and every rank's band of the 8192^2 output (a2) is hashed
(FNV-1a-64) and compared with tests/golden (hashes of the UNMODIFIED reference kernel's output, tests/golden/make_golden.py);
a mismatch on any rank fails the job (exit 3) instead of printing a line.
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
LINK_ONE_WAY_GBS = 56.4          # profiles/r01_pcie_probe.txt: measured one-way DMA rate of this host link (256 MiB copies; 52.6 at 7 MiB)
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


GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def load_golden() -> dict:
    """tests/golden/blur_golden.json (+ the per-image hashes of the 50 000-image stream): reference-kernel outputs as data."""
    with open(os.path.join(GOLDEN_DIR, "blur_golden.json")) as f:
        g = json.load(f)
    try:
        import numpy as np
        g["_stream50k_image_fnv"] = np.load(os.path.join(GOLDEN_DIR, g["stream50k"]["file"]))
    except (KeyError, OSError):
        g["_stream50k_image_fnv"] = None
    return g


def hash_images(L, host, n: int, image_bytes: int, threads: int):
    """FNV-1a-64 of each of n images laid end to end in the numpy buffer `host` (ctypes releases the GIL)."""
    import numpy as np
    out = np.empty(n, dtype=np.uint64)
    base = host.ctypes.data
    threads = max(1, min(threads, n))

    def work(b, e):
        for i in range(b, e):
            out[i] = L.mi_blur_fnv1a64(base + i * image_bytes, image_bytes)

    th = [threading.Thread(target=work, args=(n * t // threads, n * (t + 1) // threads)) for t in range(threads)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    return out


def verify_resident_stream(pkg, L, ctx, first_index: int, n_images: int, image_bytes: int, golden: dict, threads: int) -> dict:
    """Download EVERY output image of the resident pool and compare its hash with the reference kernel's
    (tests/golden/stream50k_image_fnv.npy, image i of the LCG stream).  Returns {checked, mismatches, first_bad, seconds}."""
    import numpy as np
    gold = golden.get("_stream50k_image_fnv")
    if gold is None or first_index + n_images > len(gold):
        return {"checked": 0, "mismatches": 0, "note": "no golden hashes for this range"}
    t0 = time.perf_counter()
    chunk = max(1, min(n_images, (256 << 20) // image_bytes))
    host = np.empty(chunk * image_bytes, np.uint8)
    bad, first_bad = 0, None
    for i in range(0, n_images, chunk):
        m = min(chunk, n_images - i)
        ctx.resident_download(i, host.ctypes.data, m)
        got = hash_images(L, host, m, image_bytes, threads)
        ne = np.nonzero(got != gold[first_index + i:first_index + i + m])[0]
        if len(ne):
            bad += len(ne)
            if first_bad is None:
                first_bad = first_index + i + int(ne[0])
    return {"checked": n_images, "mismatches": int(bad), "first_bad": first_bad, "seconds": round(time.perf_counter() - t0, 2)}


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


def cpu_baseline(n_target: int, h: int, w: int, c: int, radius: int, batch: int = 35) -> dict:
    """The CPU path timed beside the GPU on this box's host cores, every leg in the reference's batch-loop
    shape: per batch, build the batch buffer from the stream (the replicate memcpy of heterogeneous_blur.c:439-442, which
    the reference times, :415,603), then blur it.
      value               oracle port (scalar per-pixel C restatement of gaussian_kernel.cl — what an OpenCL CPU device
                          executes), one thread per usable CPU, each looping over its own batches
      one_thread          the same port on ONE thread
      product_cpu_device  mi_blur_cpu_run per batch (the hosts' `cpu` device: what `./heterogeneous_blur cpu`,
                          BASELINE configs[0], executes), on min(cores, 16) threads and on one
      reference_kernel    the UNMODIFIED reference kernel (oracle/_ref) when it travelled here, smaller sample
    ~20-25 s of wall clock in total."""
    import numpy as np
    O = entry.load_oracle()
    pkg = entry.load_package()
    L = pkg.lib()
    olib = O.lib()
    cores = effective_cpus()
    isz = h * w * c
    batch = max(1, min(batch, n_target))

    probe = O.lcg_stream(4, h, w, c)
    pout = probe.copy()
    olib.oracle_blur_batch(probe.ctypes.data, pout.ctypes.data, w, h, c, radius, 4)          # first touch / code load
    t0 = time.perf_counter()
    olib.oracle_blur_batch(probe.ctypes.data, pout.ctypes.data, w, h, c, radius, 4)
    per_img = max((time.perf_counter() - t0) / 4, 1e-6)
    n = int(min(n_target, max(cores * batch, 6.0 * cores / per_img)))                        # ~6 s on all cores
    src = np.empty((n, h, w, c), np.uint8)
    L.mi_blur_fill_synthetic(src.ctypes.data, w, h, c, 0, n, cores)

    def batch_loop(blur, b, e, reps):
        """The reference's loop over [b, e): batch buffer built inside the timed region, then blurred."""
        bin_, bout = np.empty(batch * isz, np.uint8), np.zeros(batch * isz, np.uint8)
        for _ in range(reps):
            for s0 in range(b, e, batch):
                m = min(batch, e - s0)
                C.memmove(bin_.ctypes.data, src.ctypes.data + s0 * isz, m * isz)               # heterogeneous_blur.c:439-442
                blur(bin_.ctypes.data, bout.ctypes.data, m)

    def port(pin, pout_, m):
        olib.oracle_blur_batch(pin, pout_, w, h, c, radius, m)

    def timed_threads(blur, n_img, threads, reps):
        th = [threading.Thread(target=batch_loop, args=(blur, n_img * i // threads, n_img * (i + 1) // threads, reps))
              for i in range(threads)]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        return time.perf_counter() - t0

    reps = int(max(1, min(64, round(6.0 * cores / (per_img * n)))))
    dt = timed_threads(port, n, cores, reps)
    res = {"value": round(n * reps / dt, 2), "unit": "img/s", "cores": cores, "kind": "port",
           "sample": f"{reps} pass(es) over {n} of the same synthetic {w}x{h}x{c} images in batches of {batch}, radius {radius}: per batch "
                     f"memcpy into the batch buffer + oracle_blur_batch (scalar per-pixel C restatement), {cores} threads each on its own "
                     f"batches (affinity mask {len(os.sched_getaffinity(0))}, cgroup quota applied), {dt:.1f} s wall"}
    n1 = int(max(batch, min(n, 3.5 / per_img)))
    dt1 = timed_threads(port, n1, 1, 1)
    res["one_thread"] = {"value": round(n1 / dt1, 2), "unit": "img/s", "cores": 1,
                         "sample": f"{n1} images, same loop on one thread, {dt1:.1f} s wall"}

    # the product's own cpu device (separable, vectorised; a named device, never a fallback): one call per batch
    nt = min(cores, 16)

    def product(threads):
        def blur(pin, pout_, m):
            pkg.check(L.mi_blur_cpu_run(pin, pout_, w, h, c, radius, m, threads), "mi_blur_cpu_run")
        blur(probe.ctypes.data, pout.ctypes.data, 4)                                              # worker pool start-up
        batch_loop(blur, 0, min(n, 8 * batch), 1)                                               # page in, spin the pool up
        t0 = time.perf_counter()
        batch_loop(blur, 0, min(n, 8 * batch), 1)
        est = max((time.perf_counter() - t0) / min(n, 8 * batch), 1e-7)
        m_img = int(max(batch, min(n, 2.5 / est)))
        r = int(max(1, min(64, round(2.5 / (est * m_img)))))
        t0 = time.perf_counter()
        batch_loop(blur, 0, m_img, r)
        d = time.perf_counter() - t0
        return {"value": round(m_img * r / d, 2), "unit": "img/s", "cores": threads,
                "sample": f"{r} pass(es) over {m_img} images, per batch of {batch}: memcpy into the batch buffer + mi_blur_cpu_run on {threads} thread(s), {d:.1f} s wall"}

    pr = product(nt)
    pr["what"] = "the hosts' `cpu` device (what `./heterogeneous_blur cpu`, BASELINE configs[0], executes)"
    pr["one_thread"] = product(1)
    res["product_cpu_device"] = pr

    # Beside it, when the build container's oracle/_ref travelled here: the UNMODIFIED reference kernel
    # (gaussian_kernel.cl compiled for x86-64 by oracle/Makefile, one call per work-item of the padded NDRange; 3x3 only),
    # on a smaller sample.  It is slower than the port (its min/max/get_global_id are out-of-line calls), so the port
    # stays the quoted baseline: the conservative one.
    if radius == 1 and O.ref_available():
        rlib = O.ref()
        m = int(min(n, max(cores, 3.0 * cores / (per_img * 3.0))))
        dst = np.empty((m, h, w, c), np.uint8)

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
    ap.add_argument("--no-parity", action="store_true",
                    help="skip the check of every output image / band against tests/golden (reference-kernel hashes)")
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
    # this rank's host work (stream generation, hashing, the a2 image, launch threads) stays on its GPU's socket; threads
    # started from here on inherit the mask.  0 = nothing changed (MI_BLUR_NO_AFFINITY, or sysfs does not expose the topology)
    host_cpus_bound = L.mi_blur_bind_thread_to_device(local_rank)
    host_cpulist, host_node = pkg.device_cpulist(local_rank)
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

    def point_e2e(w, h, c, radius, nb, nbatches, pageable=False) -> dict:
        """Host buffers in -> host buffers out (zero-copy submits over PCIe), 4 rotating buffer pairs: pinned (mi_blur_host_alloc),
        or — pageable — ordinary malloc'd memory as the reference allocates its batch buffers (heterogeneous_blur.c:431-432),
        which the library gathers into pinned staging and scatters back."""
        NS = 4
        e2e = pkg.Context(local_rank, w, h, c, radius, max_batch=nb, n_slots=NS)
        nbytes = nb * h * w * c
        keep = []
        if pageable:
            keep = [(np.zeros(nbytes, np.uint8), np.zeros(nbytes, np.uint8)) for _ in range(NS)]
            bufs = [(a.ctypes.data, b.ctypes.data) for a, b in keep]
        else:
            bufs = [(L.mi_blur_host_alloc(nbytes), L.mi_blur_host_alloc(nbytes)) for _ in range(NS)]
        for (pi, _po) in bufs:
            L.mi_blur_fill_synthetic(pi, w, h, c, 0, nb, 4)
        warm_until = time.perf_counter() + SECONDARY_WARM_S       # filling the buffers left GPU and link idle: ramp both again
        i = 0
        while time.perf_counter() < warm_until or i < 2 * NS:
            e2e.submit(bufs[i % NS][0], bufs[i % NS][1], nb)
            i += 1
        e2e.sync(); e2e.reset_timing()
        t0e = time.perf_counter()
        for i in range(nbatches):
            e2e.submit(bufs[i % NS][0], bufs[i % NS][1], nb)
        te = e2e.sync()
        dte = time.perf_counter() - t0e
        res = {"img_s": round(nbatches * nb / dte, 0), "batch": nb, "batches": nbatches, "slots": NS,
               "zero_copy_submits": int(L.mi_blur_zero_copy_launches(e2e.h)), "kernel": L.mi_blur_last_kernel().decode(),
               "pcie_gbs_each_way": round(nbatches * nbytes / dte / 1e9, 1),
               "frac_of_link": round(nbatches * nbytes / dte / 1e9 / LINK_ONE_WAY_GBS, 3),
               "h2d_ms": round(te["h2d_ms"], 2), "kernel_ms": round(te["kernel_ms"], 2), "d2h_ms": round(te["d2h_ms"], 2)}
        e2e.close()
        if pageable:
            res["buffers"] = "pageable (malloc'd), gathered into / scattered from the library's pinned staging on host threads"
        else:
            for (pi, po) in bufs:
                L.mi_blur_host_free(pi); L.mi_blur_host_free(po)
        del keep
        return res

    def point_copy_kernel(nbytes, launches=60) -> dict:
        """What THIS box gives a kernel that only moves the same bytes: torch's elementwise copy of `nbytes` (one pass's input) into
        another buffer of the same size — reads nbytes, writes nbytes, like a blur pass.  Boxes of the pool differ by up to 25 % on
        every kernel alike; this figure, taken in the same process right after the timed region, says which kind of box it was."""
        a = torch.empty(nbytes, dtype=torch.uint8, device=dev); a.random_(0, 256)
        b = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        warm_until = time.perf_counter() + SECONDARY_WARM_S
        while time.perf_counter() < warm_until:
            for _ in range(10):
                b.copy_(a)
            torch.cuda.synchronize()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(launches)]
        for e0, e1 in evs:
            e0.record(); b.copy_(a); e1.record()
        torch.cuda.synchronize()
        us = sorted(e0.elapsed_time(e1) * 1e3 for e0, e1 in evs)[launches // 2]
        del a, b
        return {"what": "torch elementwise copy, as many bytes read and written as one pass of the headline stream", "launch_us": round(us, 2),
                "achieved_gbs": round(2.0 * nbytes / us / 1e3, 1), "frac": frac_of(2.0 * nbytes, us)}

    def guarded(name, fn):
        """A secondary point must never take the headline line down: a failure becomes {"error": ...} in its place (and a
        message on stderr); the tests assert the points' contents, so a silent failure cannot pass them."""
        try:
            return fn()
        except BaseException as e:          # SystemExit from a deadline inside a point included
            if isinstance(e, KeyboardInterrupt):
                raise
            print(f"bench.py: secondary point {name} failed: {e!r}", file=sys.stderr, flush=True)
            return {"error": f"{type(e).__name__}: {e}"}

    golden = load_golden()
    hash_threads = max(1, min(16, effective_cpus() // max(world, 1)))

    def check_parity(results: dict) -> None:
        """Every rank brings {name: {"ok": bool, ...}}; ANY mismatch on ANY rank fails the whole job (exit 3, no JSON line)."""
        bad = [k for k, v in results.items() if not v.get("ok", False)]
        flag = torch.tensor([len(bad)], dtype=torch.int64, device=dev if (world > 1 and backend == "nccl") else "cpu")
        if world > 1:
            dist.all_reduce(flag, op=dist.ReduceOp.SUM)
        if bad:
            print(f"bench.py: rank {rank}: PARITY FAILURE vs tests/golden (reference kernel): "
                  + "; ".join(f"{k}: {results[k]}" for k in bad), file=sys.stderr, flush=True)
        if int(flag.item()) > 0:
            if world > 1:
                dist.destroy_process_group()
            raise SystemExit(3)

    def run_a2(K2: int, W2: int, ramp_seconds: float) -> dict:
        """BASELINE configs[4]: ONE 8192x8192x3 image, rows [H*g/G, H*(g+1)/G) resident on rank g; per step the ranks
        exchange `radius` boundary rows with their neighbours (mi_blur_halo_exchange: ncclSend/ncclRecv pairs in one RCCL
        group — split_image_blur.c:511-541 re-uploads the overlapping rows from the host instead) and blur their band.
        Content is the LCG image the golden hashes were made from; the halo rows start as POISON, so a band that hashes
        like the reference kernel's rows proves the exchange carried them.  Both step forms are timed over K2 steps each
        (plain: exchange then band on one stream; overlapped: exchange on its own stream behind the interior rows, the 2R
        edge rows after it) and the faster one is quoted."""
        H = Wd = 8192
        c, radius = 3, 1
        b = pkg.band_of(H, radius, rank, world)
        owned = b["row_end"] - b["row_begin"]
        ht, hb = b["halo_top"], b["halo_bottom"]
        rows = owned + ht + hb
        pitch = Wd * c
        band = torch.full((rows * pitch,), 0xA5, dtype=torch.uint8, device=dev)
        out = torch.zeros(owned * pitch, dtype=torch.uint8, device=dev)
        image = np.empty((H, Wd, c), np.uint8)                     # every rank generates the image and keeps its rows
        L.mi_blur_fill_synthetic(image.ctypes.data, Wd, H, c, 0, 1, 1)
        band[ht * pitch:(ht + owned) * pitch] = torch.from_numpy(image[b["row_begin"]:b["row_end"]].reshape(-1)).to(dev)
        comm = C.c_void_p()
        idbuf = torch.zeros(pkg.UNIQUE_ID_BYTES, dtype=torch.uint8)
        # Rehearsal (tests, one-GPU box: MI_BLUR_BENCH_DEVICE set): RCCL refuses two ranks on one device, so the ranks
        # run the whole control path — streams, events, barriers, reductions, the overlapped step, the parity check — with
        # the exchange itself left out (the halo rows are uploaded instead).  Never taken on a real multi-GPU run.
        fake_exchange = world > 1 and "MI_BLUR_BENCH_DEVICE" in os.environ
        if fake_exchange:
            if ht:
                band[:ht * pitch] = torch.from_numpy(image[b["row_begin"] - ht:b["row_begin"]].reshape(-1)).to(dev)
            if hb:
                band[(ht + owned) * pitch:] = torch.from_numpy(image[b["row_end"]:b["row_end"] + hb].reshape(-1)).to(dev)
        del image
        if world > 1 and not fake_exchange:
            if rank == 0:
                raw = (C.c_uint8 * pkg.UNIQUE_ID_BYTES)()
                uid_rc = L.mi_blur_comm_unique_id(raw)              # a failure here (RCCL not loadable) shows up as an init failure below
                idbuf = torch.tensor(list(raw), dtype=torch.uint8) if uid_rc == 0 else idbuf
            idd = idbuf.to(dev) if backend == "nccl" else idbuf
            dist.broadcast(idd, src=0)
            idbuf = idd.cpu()
        idarr = (C.c_uint8 * pkg.UNIQUE_ID_BYTES)(*idbuf.tolist())
        comm_ranks, comm_transport = 1, 0
        setup_error = ""
        if not fake_exchange:
            try:
                pkg.check(L.mi_blur_comm_init_rank(C.byref(comm), world, rank, idarr), "comm_init_rank")
                nr, rk, tr = C.c_int(), C.c_int(), C.c_int()
                pkg.check(L.mi_blur_comm_info(comm, C.byref(nr), C.byref(rk), C.byref(tr)), "comm_info")
                if nr.value != world or rk.value != rank:
                    raise RuntimeError(f"communicator reports rank {rk.value} of {nr.value}, expected {rank} of {world}")
                comm_ranks, comm_transport = nr.value, tr.value
            except Exception as e:                      # every rank must learn of it before anyone enters an exchange
                setup_error = f"rank {rank}: {e}"
        if world > 1:
            bad = torch.tensor([1 if setup_error else 0], dtype=torch.int64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(bad, op=dist.ReduceOp.SUM)
            if int(bad.item()) > 0:
                if comm:
                    L.mi_blur_comm_destroy(comm)
                return {"error": setup_error or f"RCCL communicator set-up failed on {int(bad.item())} other rank(s)"}
        elif setup_error:
            return {"error": setup_error}
        # ---- third step form: every rank PULLS its halo rows out of its neighbours' shards with one small copy kernel that reads
        # the peer's memory directly (mi_blur_halo_pull; IPC handles of the shards travel through an all_gather).  Optional: any
        # failure to set it up is agreed on by all ranks and the form is left out.  (In the one-GPU rehearsal the "peers" are
        # other processes on the same device: the IPC plumbing and the kernel are the real ones.)
        peer_ptrs, pull_note = {}, ""
        top_src = bottom_src = 0
        if world > 1:
            mine = torch.zeros(pkg.PEER_HANDLE_BYTES + 8, dtype=torch.uint8)
            try:
                hraw, off = (C.c_uint8 * pkg.PEER_HANDLE_BYTES)(), C.c_uint64()
                pkg.check(L.mi_blur_peer_export(band.data_ptr(), hraw, C.byref(off)), "peer_export")
                mine = torch.tensor(list(hraw) + list(int(off.value).to_bytes(8, "little")), dtype=torch.uint8)
                ok = 1
            except Exception as e:
                ok, pull_note = 0, f"rank {rank}: {e}"
            send = mine.to(dev) if backend == "nccl" else mine
            gathered = [torch.zeros_like(send) for _ in range(world)]
            dist.all_gather(gathered, send)
            if ok:
                try:
                    for nb_rank in (rank - 1, rank + 1):
                        if 0 <= nb_rank < world:
                            g = gathered[nb_rank].cpu().numpy().tobytes()
                            hb = (C.c_uint8 * pkg.PEER_HANDLE_BYTES)(*g[:pkg.PEER_HANDLE_BYTES])
                            noff = int.from_bytes(g[pkg.PEER_HANDLE_BYTES:], "little")
                            ptr = C.c_void_p()
                            pkg.check(L.mi_blur_peer_open(hb, noff, C.byref(ptr)), "peer_open")
                            peer_ptrs[nb_rank] = (ptr.value, noff)
                    if rank - 1 in peer_ptrs:      # the LAST `radius` owned rows of the shard above
                        ab = pkg.band_of(H, radius, rank - 1, world)
                        top_src = peer_ptrs[rank - 1][0] + (ab["halo_top"] + ab["row_end"] - ab["row_begin"] - radius) * pitch
                    if rank + 1 in peer_ptrs:      # the FIRST `radius` owned rows of the shard below
                        bottom_src = peer_ptrs[rank + 1][0] + pkg.band_of(H, radius, rank + 1, world)["halo_top"] * pitch
                except Exception as e:
                    ok, pull_note = 0, f"rank {rank}: {e}"
            flag = torch.tensor([0 if ok else 1], dtype=torch.int64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.SUM)
            if int(flag.item()) > 0:
                pull_note = pull_note or f"peer set-up failed on {int(flag.item())} other rank(s)"
                for (ptr_v, noff) in peer_ptrs.values():
                    L.mi_blur_peer_close(ptr_v, noff)
                peer_ptrs = {}
        pull_ok = world > 1 and bool(peer_ptrs) and not pull_note

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

        forms = {"plain": step}
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

            forms["overlapped"] = step_overlapped
        if pull_ok:
            def step_pull():
                pkg.check(L.mi_blur_halo_pull(band.data_ptr(), top_src or None, bottom_src or None, Wd, c, owned, radius, stream), "halo_pull")
                blur_rows(ht, ht + owned, 0)

            forms["pull"] = step_pull

            def step_peer():       # the whole step as ONE launch: the band kernel reads the neighbours' rows in place
                pkg.check(L.mi_blur_enqueue_band_peer(band.data_ptr(), out.data_ptr(), Wd, rows, c, radius, ht, ht + owned,
                                                      top_src or None, bottom_src or None, stream), "enqueue_band_peer")

            forms["peer"] = step_peer

        # untimed ramp past the ~40 ms clock ramp that follows any idle gap.  A FIXED step count, the same on every rank:
        # each step holds a send/recv pair, so ranks must not decide by their own clocks how many to run.
        ramp_steps = int(ramp_seconds * 1e6 / 100.0)
        for i in range(ramp_steps):
            step()
            if i % 64 == 63:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        want = golden.get("bands8192", {}).get("bands", {}).get(str(world))
        want = want[rank] if want else None
        timed, parity = {}, {"ok": True, "golden": "tests/golden bands8192 (reference kernel)", "band_fnv": {}}
        halo_rows_host = None
        if fake_exchange:      # rehearsal: what the (left-out) RCCL exchange would have delivered, kept for re-upload
            halo_rows_host = (band[:ht * pitch].clone(), band[(ht + owned) * pitch:].clone())
        for fname, fn in forms.items():
            out.zero_()
            # every form starts from poisoned halo rows, so its OWN exchange is what the band hash proves (rehearsal: the two
            # RCCL forms get the rows uploaded instead, as their exchange is left out; the pull form is real there as well)
            if world > 1:
                if fake_exchange and fname not in ("pull", "peer"):
                    band[:ht * pitch] = halo_rows_host[0]
                    band[(ht + owned) * pitch:] = halo_rows_host[1]
                else:
                    band[:ht * pitch].fill_(0xA5)
                    band[(ht + owned) * pitch:].fill_(0xA5)
                torch.cuda.synchronize()
                barrier_sync()                  # nobody starts pulling / sending before every rank's rows are in place
            if fname == "overlapped":
                ev_done.record(main_stream)
            for _ in range(W2):
                fn()
            barrier_sync()
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            ev0.record()
            for _ in range(K2):
                fn()
            ev1.record()
            torch.cuda.synchronize()
            loc = time.perf_counter() - t0        # this rank's K steps, from the common (barrier + sync) start to its own drain;
            if world > 1:                         # the closing barrier + sync follow, then MAX over ranks: the job's time is
                dist_barrier()                    # the slowest rank's, without the barrier's own latency added to every rank
                torch.cuda.synchronize()
            el = aggregate_max(loc, dist if world > 1 else None, dev if backend == "nccl" else None)
            got = out.cpu().numpy()
            fnv = f"{L.mi_blur_fnv1a64(got.ctypes.data, got.size):016x}"
            parity["band_fnv"][fname] = fnv
            if want is None:
                parity["note"] = f"no golden band hashes for G={world}"
            elif fnv != want:
                parity["ok"] = False
                parity["expected"] = want
            timed[fname] = {"elapsed": el, "local": loc, "step_us": round(el / K2 * 1e6, 2), "stream_us": ev0.elapsed_time(ev1) * 1e3 / K2,
                            "kernel": L.mi_blur_last_kernel().decode()}
        quoted = min(timed, key=lambda k: timed[k]["elapsed"])
        kernel_name = timed[quoted]["kernel"]
        bytes_per_launch = 2.0 * owned * pitch

        # ---- per-step decomposition (outside the timed regions): exchange vs band kernel on this rank's stream
        n_i = max(1, min(K2, 50))
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
                  "band_kernel_frac": frac_of(bytes_per_launch, k_us),
                  "step_us_plain": timed["plain"]["step_us"]}
        if "overlapped" in timed:
            decomp["step_us_overlapped"] = timed["overlapped"]["step_us"]
        if world > 1:      # the slowest rank's figures (rank 0 and the last rank have one neighbour only)
            t = torch.tensor([decomp["halo_exchange_us"], decomp["band_kernel_us"]], dtype=torch.float64,
                             device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            decomp["halo_exchange_us_max_over_ranks"], decomp["band_kernel_us_max_over_ranks"] = round(float(t[0]), 2), round(float(t[1]), 2)
        if world > 1:
            barrier_sync()                      # no rank closes or frees a shard a neighbour may still be pulling from
        for (ptr_v, noff) in peer_ptrs.values():
            L.mi_blur_peer_close(ptr_v, noff)
        if not fake_exchange:
            L.mi_blur_comm_destroy(comm)
        config = {"workload": f"one 8192x8192x3 image per step, 3x3, row-split over {world} GPU(s), RCCL halo exchange [BASELINE configs[4]]",
                  "rows_per_gpu": owned, "halo_bytes_per_neighbour": radius * pitch, "ramp_steps": ramp_steps,
                  "steps_per_form": K2, "step_forms_us": {k: v["step_us"] for k, v in timed.items()}, "quoted_form": quoted,
                  "rccl_step_us": min(v["step_us"] for k, v in timed.items() if k not in ("pull", "peer")),
                  "pull_form": ("halo rows pulled out of the neighbours' shards by one copy kernel reading peer memory (IPC-mapped)"
                                if pull_ok else f"not run: {pull_note or 'one rank'}"),
                  "peer_form": ("one launch per step: the band kernel reads its halo rows in place from the neighbours' shards "
                                "(IPC-mapped peer memory); this shard's own halo rows stay poisoned"
                                if pull_ok else f"not run: {pull_note or 'one rank'}"),
                  "rccl_ranks": comm_ranks if comm_transport == 1 else 0,
                  "halo_transport": {0: "none (one rank: both image edges clamp)", 1: "RCCL ncclSend/ncclRecv", 2: "peer copies"}[comm_transport],
                  "step_decomposition": decomp}
        if fake_exchange:
            config["rehearsal"] = "halo exchange left out (two ranks share one device; halo rows uploaded); control path + parity only"
            config["halo_transport"] = "none (rehearsal)"
        parity["checked"] = f"rank band of {owned} rows, every step form"
        q = timed[quoted]
        del band, out
        return {"value": K2 / q["elapsed"], "elapsed": q["elapsed"], "local": q["local"], "bytes_per_launch": bytes_per_launch,
                "avg_launch_s": q["stream_us"] / 1e6, "kernel": kernel_name, "config": config, "parity": parity,
                "timing_src": f"stream events around the {quoted} step (halo exchange + band kernel) on this rank's stream"}

    fused = args.workload == "a1" and args.dispatch == "fused" and args.batch < per_gpu_images
    if args.streams <= 0:
        # several streams only help small launches whose dispatch floors must overlap; one launch per pass stays on one stream
        args.streams = 4 if (args.workload == "a1" and not fused and args.batch < per_gpu_images) else 1
    extra = {}
    other_line, other_key = None, None
    sustained = None
    parity_line, completion, release_mode = None, None, None
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
        pool_placement = ctx.resident_placement()                  # the library tried a few placements of the pool and kept the fastest
        first, _ = shard_range(per_gpu * world, rank, world)       # image-level sharding: rank g owns [first, first + per_gpu)
        # test hook (tests/test_cli.py): the last rank loads the WRONG shard, to show that the parity check fails the job
        wrong = 1 if (os.environ.get("MI_BLUR_BENCH_FAULT") == "wrong_shard" and rank == world - 1) else 0
        ctx.resident_fill_synthetic(first + wrong)

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
        per_rank = None
        if world > 1:      # every rank's own rate and dispatch time: a straggler (a GPU on a busy socket, a slow link) shows here
            mine = torch.tensor([per_gpu * K / local, tm["kernel_ms"] * 1e3 / max(ctx.timed_coverage()[0], 1)], dtype=torch.float64,
                                device=dev if backend == "nccl" else "cpu")
            allr = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(allr, mine)
            per_rank = {"img_s": [round(float(t[0]), 0) for t in allr], "avg_launch_us": [round(float(t[1]), 2) for t in allr]}
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
        if per_rank:
            config["per_rank"] = per_rank
        if pool_placement["candidates_us"]:
            config["pool_placement"] = dict(pool_placement, what="mi_blur_resident_alloc timed this many candidate placements of the pool "
                                                                 "(one launch over the whole pool each) and kept the fastest: where a pool "
                                                                 "lands in HBM moves the big launches between two levels ~6 % apart")

        # ---- parity (after the clock has stopped): EVERY output image of this rank's shard is downloaded, hashed and compared
        # with the unmodified reference kernel's hash of the same image of the stream (tests/golden/stream50k_image_fnv.npy).
        # A rank that blurred the wrong shard, or wrongly, fails the job here.
        parity_results = {}
        if args.workload == "a1" and not args.no_parity:
            pv = verify_resident_stream(pkg, L, ctx, first, per_gpu, h * w * c, golden, hash_threads)
            pv["ok"] = pv["mismatches"] == 0
            parity_results["a1_stream"] = pv
        check_parity(parity_results)
        if parity_results:
            tot = torch.tensor([parity_results["a1_stream"]["checked"]], dtype=torch.int64,
                               device=dev if (world > 1 and backend == "nccl") else "cpu")
            if world > 1:
                dist.all_reduce(tot, op=dist.ReduceOp.SUM)
            parity_line = {"status": "ok" if int(tot.item()) > 0 else "unchecked",
                           "a1_stream": {"images_checked_all_ranks": int(tot.item()), "mismatches": 0,
                                         "rank0_first_image": first, "rank0_seconds": parity_results["a1_stream"].get("seconds"),
                                         "against": "tests/golden/stream50k_image_fnv.npy: FNV-1a-64 of every output image of the "
                                                    "50 000-image LCG stream through the unmodified reference kernel"}}
            if "note" in parity_results["a1_stream"]:
                parity_line["a1_stream"]["note"] = parity_results["a1_stream"]["note"]

        if do_extra and fused:
            # ---- the batch as the unit of COMPLETION, as the host sees it (per-batch clFinish, heterogeneous_blur.c:538-539):
            # a poll loop on mi_blur_resident_batches_done during `passes` fused passes, each batch stamped with the host
            # time at which a poll first reported it, measured from just before the dispatch call.
            def batch_completion(passes=24, watch=True):
                nb = (per_gpu + batch - 1) // batch
                seen_at = np.full((passes, nb), np.nan)
                polls, poll_s, disp = 0, 0.0, []
                for p in range(passes):
                    ctx.sync()
                    t_0 = time.perf_counter()
                    ctx.resident_run_fused(per_gpu, batch, watch=watch)
                    seen, deadline = 0, t_0 + 2.0
                    while seen < nb:
                        ta = time.perf_counter()
                        nd = ctx.resident_batches_done()
                        tb = time.perf_counter()
                        polls += 1
                        poll_s += tb - ta
                        if nd > seen:
                            seen_at[p, seen:nd] = tb - t_0
                            seen = nd
                        if tb > deadline:
                            raise SystemExit(f"bench.py: fused pass: only {seen} of {nb} batches counted in after 2 s")
                    ctx.sync()
                    disp.append(time.perf_counter() - t_0)
                med = np.median(seen_at, axis=0) * 1e6
                return {"first": round(float(med[0]), 1), "p50": round(float(med[nb // 2]), 1), "last": round(float(med[-1]), 1),
                        "batches": nb, "passes": passes, "polls_per_pass": round(polls / passes, 1), "poll_us": round(poll_s / polls * 1e6, 1),
                        "pass_wall_us": round(float(np.median(disp)) * 1e6, 1),
                        "how": "median over the passes of the host time (from just before the dispatch call) at which a "
                               "mi_blur_resident_batches_done poll first reported batch k complete; "
                               + ("the pass is WATCHED (a one-wave kernel keeps the count in pinned host memory), so a poll is a read "
                                  "of the host's own memory" if watch else
                                  "every poll is a counter read-back on its own stream while the dispatch runs, so resolution = one poll")}

            def completion_point():
                batch_completion(3)                                 # poll stream / code paths warm
                res = batch_completion()
                res["by_counter_readback"] = {k: v for k, v in batch_completion(12, watch=False).items() if k != "how"}
                return res

            completion = guarded("batch_completion_us", completion_point)

            # ---- the same pass with the ARCHITECTURAL completion protocol (release-ordered add at agent scope)
            def release_point():
                pkg.check(L.mi_blur_set_option(b"fused_release", 1), "set_option")
                try:
                    ctx.resident_run_fused(per_gpu, batch); ctx.sync(); ctx.reset_timing()
                    n_rel = 5
                    t_r0 = time.perf_counter()
                    for _ in range(n_rel):
                        ctx.resident_run_fused(per_gpu, batch, timed=True)
                    t_rel = ctx.sync()
                    wall_rel = time.perf_counter() - t_r0
                    rn, _rb = ctx.timed_coverage()
                    return {"dispatch_us": round(t_rel["kernel_ms"] * 1e3 / max(rn, 1), 1), "passes": n_rel,
                            "img_s": round(n_rel * per_gpu / wall_rel, 0), "batches_counted_in": ctx.resident_batches_done(),
                            "what": "mi_blur_set_option(\"fused_release\", 1): every block publishes with a release-ordered agent-scope "
                                    "add (an L2 write-back per block) instead of write-through stores + relaxed add"}
                finally:
                    pkg.check(L.mi_blur_set_option(b"fused_release", 0), "set_option")
                    ctx.resident_run_fused(per_gpu, batch); ctx.sync(); ctx.reset_timing()

            release_mode = guarded("release_mode_us", release_point)

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
            extra["hd1080_5x5"] = guarded("hd1080_5x5", lambda: point_resident(1920, 1080, 3, 2, 64, 64, 64, 300,
                                                                             "64 x 1920x1080x3 per launch, 5x5, resident pool of 64 (configs[2])"))
            extra["a2_8192_1gpu"] = guarded("a2_8192_1gpu", lambda: point_a2_1gpu(300))
            extra["copy_kernel_same_box"] = guarded("copy_kernel", lambda: point_copy_kernel(per_gpu * h * w * c))
            extra["e2e_pcie_inclusive"] = {"batch_35": guarded("e2e batch 35", lambda: point_e2e(256, 256, 3, 1, 35, 143 * 4)),
                                           "batch_500": guarded("e2e batch 500", lambda: point_e2e(256, 256, 3, 1, 500, 40)),
                                           "batch_35_pageable": guarded("e2e batch 35 pageable", lambda: point_e2e(256, 256, 3, 1, 35, 143 * 4, pageable=True)),
                                           "hd1080_5x5_batch_8": guarded("e2e 1080p 5x5", lambda: point_e2e(1920, 1080, 3, 2, 8, 64)),   # configs[2] end to end
                                           "link_one_way_gbs": LINK_ONE_WAY_GBS,
                                           "note": "pinned host buffers in and out, the batch server's workgroups work on them in place over PCIe "
                                                   "(both directions at once; frac_of_link = each-way rate / the measured ONE-way DMA rate); "
                                                   "comparable to the reference's wall clock (heterogeneous_blur.c:415,603); batch_35_pageable = the "
                                                   "reference's own kind of batch buffers (malloc), unchanged: one host copy each way on top"}
        base_shape = (h, w, c, radius)
    else:   # a2: one 8192x8192x3 image, row-split, RCCL halo exchange
        r2 = run_a2(K, W, args.ramp_seconds)
        if "error" in r2:
            raise SystemExit(f"bench.py: --workload a2: {r2['error']}")
        check_parity({"a2_8192_rowsplit": r2["parity"]})
        value, elapsed, scaling, units = r2["value"], r2["elapsed"], "strong", K
        launches = timed_n = K
        bytes_per_launch, avg_launch_s, timing_src = r2["bytes_per_launch"], r2["avg_launch_s"], r2["timing_src"]
        dominant_kernel, config = r2["kernel"], r2["config"]
        local = r2["local"]
        parity_line = {"status": "unchecked" if r2["parity"].get("note") else "ok", "a2_8192_rowsplit": r2["parity"]}
        base_shape = (8192, 8192, 3, 1)

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

    config["host_placement"] = ({"cpus": host_cpulist, "numa_node": host_node, "cpus_allowed_there": host_cpus_bound} if host_cpus_bound
                                else "not pinned (MI_BLUR_NO_AFFINITY or topology not exposed)")
    line = {"metric": "images_per_sec", "value": round(value, 1), "unit": "img/s", "n_gpus": world,
            "steps": K, "warmup": W, "ms_per_step": round(elapsed / K * 1e3, 4), "higher_is_better": True,
            "scaling": scaling, "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": config, "roofline": roofline}
    if world > 1 and args.workload == "a1" and not args.no_extra:
        # ---- the OTHER multi-GPU config on the same ranks: BASELINE configs[4], the 8192^2 row split with RCCL halo exchange.
        # RCCL has never carried these halos on real xGMI before the first multi-GPU lease, so the configs[3] line above must
        # survive whatever happens here: set-up errors are agreed on by all ranks and recorded; a step that never completes
        # (a send without its receive) trips a watchdog on every rank, rank 0 prints the line with the error in place of
        # the a2 figures, and the ranks leave.  A WRONG band, on the other hand, fails the job like any parity failure.
        def a2_error_line(msg):
            ln = dict(line)
            ln["extra"] = dict(extra, a2_8192_rowsplit={"error": msg})
            if parity_line:
                ln["parity"] = dict(parity_line, status="ok (configs[3] only; configs[4] did not run: see extra.a2_8192_rowsplit.error)")
            return ln

        finished = threading.Event()

        def watchdog(limit_s=float(os.environ.get("MI_BLUR_BENCH_A2_LIMIT_S", "240"))):
            if finished.wait(limit_s):
                return
            if rank == 0:
                print(json.dumps(a2_error_line(f"configs[4] leg did not complete within {limit_s:.0f} s (halo exchange or barrier stuck); ranks stopped")), flush=True)
            sys.stderr.write(f"bench.py: rank {rank}: configs[4] leg timed out; leaving\n")
            sys.stderr.flush()
            os._exit(0)

        threading.Thread(target=watchdog, daemon=True).start()
        r2 = run_a2(200, 5, min(args.ramp_seconds, 0.25))
        finished.set()
        if "error" in r2:
            extra["a2_8192_rowsplit"] = {"error": r2["error"]}
            if parity_line:
                parity_line["status"] = "ok (configs[3] only; configs[4] did not run: see extra.a2_8192_rowsplit.error)"
        else:
            check_parity({"a2_8192_rowsplit": r2["parity"]})
            extra["a2_8192_rowsplit"] = {"img_s": round(r2["value"], 1), "step_us": round(r2["elapsed"] / 200 * 1e6, 2), "scaling": "strong",
                                         "kernel": r2["kernel"], "band_frac_of_hbm_peak": r2["config"]["step_decomposition"]["band_kernel_frac"],
                                         **r2["config"]}
            if parity_line is not None:
                parity_line["a2_8192_rowsplit"] = r2["parity"]
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            hh, ww, cc, rr = base_shape
            if args.workload == "a2":
                hh, ww = 1024, 8192           # a band-sized slice of the same image keeps the sample bounded
            line["cpu_baseline"] = guarded("cpu_baseline", lambda: cpu_baseline(5000 if args.workload == "a1" else 64, hh, ww, cc, rr))
        if sustained:
            line["sustained_img_s"] = sustained["img_s"]
            line["sustained"] = sustained
        if other_line:
            line[other_key] = other_line
        if completion:
            line["batch_completion_us"] = completion
        if release_mode:
            line["release_mode_us"] = release_mode
        if parity_line:
            line["parity"] = parity_line
        if extra:
            line["extra"] = extra
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
