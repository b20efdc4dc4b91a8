"""N>1 host logic on CPU with gloo, world_size 2 and 3 (no GPU needed).

* Approach 1 (image-level sharding, no collective): every rank blurs its shard_range of the stream;
  the union is the whole stream and bench.py's MAX-over-ranks timing aggregation works.
* Approach 2 (row shards + halo exchange): ranks hold [halo_top][owned][halo_bottom] bands laid out
  by mi_blur_band_of, exchange `radius` boundary rows with their neighbours using the SAME
  send/recv pattern mi_blur_halo_exchange issues over RCCL (here over gloo), blur their band with
  band semantics and reproduce the whole-image result byte for byte.
The band blur here is the oracle (this is a CPU test of the distribution logic; the GPU band kernel
itself is covered by tests/test_gpu_parity.py::test_band_semantics_and_split_equals_whole).
"""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, radius, shape, out_dir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import __graft_entry__ as entry
    import bench

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg, O = entry.load_package(), entry.load_oracle()
    h, w, c = shape

    # ---- Approach 1: shard the stream, no collective on the data path
    n_stream = 23
    b, e = bench.shard_range(n_stream, rank, world)
    assert (b, e) == pkg.shard_range(n_stream, rank, world)
    mine = O.lcg_stream(e - b, 16, 16, 3, first_index=b)
    got = O.blur_batch(mine, radius) if e > b else mine
    np.save(os.path.join(out_dir, f"a1_{rank}.npy"), got)
    t = bench.aggregate_max(0.25 * (rank + 1), dist)          # MAX over ranks
    assert abs(t - 0.25 * world) < 1e-12

    # ---- Approach 2: row shards + neighbour halo exchange
    img = O.lcg_image(h, w, c)                                 # every rank can regenerate the image; it only USES its rows
    band = pkg.band_of(h, radius, rank, world)
    owned = band["row_end"] - band["row_begin"]
    rows = owned + band["halo_top"] + band["halo_bottom"]
    buf = torch.zeros((rows, w, c), dtype=torch.uint8)
    top = band["halo_top"]
    buf[top:top + owned] = torch.from_numpy(img[band["row_begin"]:band["row_end"]])
    # same pattern as mi_blur_halo_exchange: send first/last `radius` OWNED rows, receive into the halo rows
    ops = []
    if rank > 0:
        ops.append(dist.P2POp(dist.isend, buf[top:top + radius].contiguous(), rank - 1))
        recv_top = torch.empty((radius, w, c), dtype=torch.uint8)
        ops.append(dist.P2POp(dist.irecv, recv_top, rank - 1))
    if rank < world - 1:
        ops.append(dist.P2POp(dist.isend, buf[top + owned - radius:top + owned].contiguous(), rank + 1))
        recv_bot = torch.empty((radius, w, c), dtype=torch.uint8)
        ops.append(dist.P2POp(dist.irecv, recv_bot, rank + 1))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    if rank > 0:
        buf[:top] = recv_top
    if rank < world - 1:
        buf[top + owned:] = recv_bot
    blurred = O.blur(np.ascontiguousarray(buf.numpy()), radius)   # band semantics: clamp at the band's own edges
    np.save(os.path.join(out_dir, f"a2_{rank}.npy"), blurred[top:top + owned])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,radius", [(2, 1), (2, 2), (3, 1)])
def test_sharding_and_halo_exchange_gloo(world, radius, tmp_path, O, pkg):
    import torch.multiprocessing as mp

    shape = (37, 24, 3)
    port = _free_port()
    mp.spawn(_worker, args=(world, port, radius, shape, str(tmp_path)), nprocs=world, join=True)
    # Approach 1: the union of the shards is the blurred stream
    got = np.concatenate([np.load(tmp_path / f"a1_{r}.npy") for r in range(world)])
    assert np.array_equal(got, O.blur_batch(O.lcg_stream(23, 16, 16, 3), radius))
    # Approach 2: shards reassemble to the single-device blur
    whole = np.concatenate([np.load(tmp_path / f"a2_{r}.npy") for r in range(world)])
    assert np.array_equal(whole, O.blur(O.lcg_image(*shape), radius))


def test_bench_json_helpers():
    sys.path.insert(0, ROOT)
    import bench
    assert bench.shard_range(5000, 0, 1) == (0, 5000)
    parts = [bench.shard_range(50000, r, 8) for r in range(8)]
    assert parts[0][0] == 0 and parts[-1][1] == 50000 and all(e - b == 6250 for b, e in parts)
    assert bench.aggregate_max(1.5, None) == 1.5
    assert bench.load_traffic("no-such-workload") is None


def test_bench_default_images_follow_the_baseline_configs():
    """N=1 is configs[1] (5000 images); N>1 shards configs[3]'s 50 000 images, so N=8 is exactly 6250 per GPU."""
    sys.path.insert(0, ROOT)
    import bench
    assert [bench.default_images(n) for n in (1, 2, 4, 8)] == [5000, 25000, 12500, 6250]
    assert bench.default_images(8) * 8 == bench.CONFIG3_IMAGES


def _run_bench(args, env_extra=None, drop=()):
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK") + tuple(drop)}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, cwd=ROOT,
                          capture_output=True, text=True, timeout=600)


def test_bench_gpus_n_never_degrades_to_one_rank():
    """`bench.py --gpus N` must either run N ranks or fail: (a) with fewer than N devices visible and no launcher it
    exits non-zero naming the shortfall; (b) a launcher that made a different number of ranks is refused; (c) when it
    spawns the ranks itself and one of them fails, the job fails (here: the children find no GPU) and says which rank."""
    import torch
    if torch.cuda.device_count() < 2:
        r = _run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0"], drop=("MI_BLUR_BENCH_DEVICE",))
        assert r.returncode != 0 and "--gpus 2 but only" in r.stderr and not r.stdout.strip(), r.stdout + r.stderr
    r = _run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0"], {"WORLD_SIZE": "3", "RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=3" in r.stderr
    r = _run_bench(["--gpus", "1", "--steps", "1", "--warmup", "0"], {"WORLD_SIZE": "2", "RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr
    if not torch.cuda.is_available():
        # rehearsal env skips the device-count gate, so the parent really spawns 2 children; without a GPU each child
        # stops at "needs a GPU" and the parent must report the failure instead of printing a line
        r = _run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0"], {"MI_BLUR_BENCH_DEVICE": "0", "MI_BLUR_BENCH_BACKEND": "gloo"})
        assert r.returncode != 0 and "exited with" in r.stderr and "needs a GPU" in r.stderr, r.stdout + r.stderr
        assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
