"""The C++ hosts keep the reference command lines and report sections
(heterogeneous_blur.c:41-100,609-724; split_image_blur.c:62-102,615-721).
CPU-only cases run here; cases that need a GPU carry the gpu marker."""
import os
import subprocess

import numpy as np
import pytest


@pytest.fixture(scope="module")
def apps(pkg):
    pkg.build_native()
    a = os.path.join(pkg.APPS, "heterogeneous_blur")
    b = os.path.join(pkg.APPS, "split_image_blur")
    assert os.path.exists(a) and os.path.exists(b)
    return a, b


def run(cmd, cwd):
    return subprocess.run(cmd, cwd=cwd, capture_output=True, text=True, timeout=600)


def write_ppm(path, img):
    h, w, c = img.shape
    with open(path, "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % (w, h) if c == 3 else b"P5\n%d %d\n255\n" % (w, h))
        f.write(img.tobytes())


def read_ppm(path):
    with open(path, "rb") as f:
        magic = f.readline().strip()
        w, h = map(int, f.readline().split())
        assert f.readline().strip() == b"255"
        c = 3 if magic == b"P6" else 1
        return np.frombuffer(f.read(), np.uint8).reshape(h, w, c)


A1_SECTIONS = ["========== HETEROGENEOUS CONFIGURATION ==========", "Number of images in stream:", "Batch size:",
               "Number of batches:", "Work-group size: 16x16", "Execution mode :", "Original image loaded:",
               "Size of one image:", "Starting batch processing of", "All batches finished!",
               "========== PERFORMANCE RESULTS ==========", "1. OVERALL EXECUTION TIME", "Total wall-clock time:",
               "Total images processed:", "7. THROUGHPUT", "Overall throughput:", "Images per second:"]


def test_a1_cpu_mode_report_and_pixels(apps, O, tmp_path):
    """`heterogeneous_blur cpu` (BASELINE config 0): plumbing without a GPU; output pixels == oracle."""
    het, _ = apps
    img = O.lcg_image(48, 64, 3)
    write_ppm(tmp_path / "in.ppm", img)
    r = run([het, "cpu", "0.5", "35", "--image", "in.ppm", "--images", "100", "--save", "out.ppm", "--csv", "run.csv"], tmp_path)
    assert r.returncode == 0, r.stdout + r.stderr
    for s in A1_SECTIONS + ["Mode: CPU ONLY", "Batch size: 35 images", "Number of batches: 3", "Execution mode : 1",
                            "2. CPU DEVICE (processed 100 images)", "Original image loaded: 64x48, 3 channels"]:
        assert s in r.stdout, s
    assert "3. GPU DEVICE" not in r.stdout and "4. DEVICE COMPARISON" not in r.stdout     # heterogeneous_blur.c:638,654
    assert np.array_equal(read_ppm(tmp_path / "out.ppm"), O.blur(img, 1))
    rows = open(tmp_path / "run.csv").read().splitlines()
    assert rows[0].startswith("batch_size_file,run,file,mode,") and rows[0].endswith("hbm_gbps,roofline_frac,n_gpus")
    assert len(rows) == 2 and rows[1].split(",")[6] == "100"


def test_a1_argument_handling_matches_reference(apps, tmp_path):
    het, _ = apps
    r = run([het, "cpu", "1.5", "99999", "--images", "40", "--size", "32x16"], tmp_path)
    assert r.returncode == 0
    assert "Warning: gpu_ratio must be between 0.0 and 1.0. Using 0.5" in r.stdout          # :72-75
    assert "Warning: BATCH_SIZE must be between 1 and 40. Using 500" in r.stdout            # :80-83
    r = run([het, "sideways", "--images", "10", "--size", "32x16"], tmp_path)
    assert "Usage:" in r.stdout and "Defaulting to heterogeneous mode." in r.stdout           # :63-64


def test_a1_ksize5_cpu(apps, O, tmp_path):
    het, _ = apps
    img = O.lcg_image(40, 48, 3)
    write_ppm(tmp_path / "in.ppm", img)
    r = run([het, "cpu", "0.5", "7", "--image", "in.ppm", "--images", "20", "--ksize", "5", "--save", "out.ppm"], tmp_path)
    assert r.returncode == 0, r.stdout
    assert np.array_equal(read_ppm(tmp_path / "out.ppm"), O.blur(img, 2))


def _write_frames(O, d, n, h, w, c, pattern="f_%04d"):
    os.makedirs(d, exist_ok=True)
    frames = O.lcg_stream(n, h, w, c, first_index=1000)          # distinct content per frame
    for i in range(n):
        write_ppm(os.path.join(d, (pattern % i) + (".ppm" if c == 3 else ".pgm")), frames[i])
    return frames


def _check_saved_frames(O, d, frames, radius, pattern="f_%04d"):
    want = O.blur_batch(frames, radius)
    ext = ".ppm" if frames.shape[3] == 3 else ".pgm"
    for i in range(len(frames)):
        assert np.array_equal(read_ppm(os.path.join(d, (pattern % i) + ext)), want[i]), f"frame {i}"
    assert len([f for f in os.listdir(d)]) == len(frames)


FRAME_SECTIONS = ["Input frames:", "Frame geometry (from", "Starting frame stream:", "All batches finished!", "1. OVERALL EXECUTION TIME",
                  "7. THROUGHPUT", "10. FRAME INGEST (distinct frames; ingest-inclusive figures)", "Decode:", "Ingest-inclusive throughput:"]


def test_frames_stream_on_the_cpu_device(apps, O, tmp_path):
    """--frames: a stream of DISTINCT frames (the reference copies one decoded image N times,
    heterogeneous_blur.c:106-135,439-442).  cpu device here; every saved frame must equal the oracle's blur of ITS input."""
    het, _ = apps
    frames = _write_frames(O, tmp_path / "in", 11, 37, 52, 3)
    r = run([het, "cpu", "0.5", "4", "--frames", "in", "--save-dir", "out"], tmp_path)
    assert r.returncode == 0, r.stdout + r.stderr
    for sct in FRAME_SECTIONS + ["Number of images in stream: 11", "Number of batches: 3", "2. CPU DEVICE (processed 11 frames)", "Save:"]:
        assert sct in r.stdout, sct
    _check_saved_frames(O, tmp_path / "out", frames, 1)
    # printf-style pattern, 5x5, planar output (the save path de-interleaves), grey frames, --images truncates the list
    grey = _write_frames(O, tmp_path / "g", 9, 40, 48, 1, pattern="g%02d")
    r = run([het, "cpu", "0.5", "5", "--frames", "g/g%02d.pgm", "--save-dir", "gout", "--ksize", "5", "--planar-out", "--images", "7"], tmp_path)
    assert r.returncode == 0 and "Number of images in stream: 7" in r.stdout, r.stdout + r.stderr
    _check_saved_frames(O, tmp_path / "gout", grey[:7], 2, pattern="g%02d")
    r = run([het, "cpu", "0.5", "4", "--frames", "in", "--save-dir", "out3", "--planar-out"], tmp_path)
    assert r.returncode == 0
    _check_saved_frames(O, tmp_path / "out3", frames, 1)
    r = run([het, "cpu", "0.5", "4", "--frames", "in", "--save-dir", "out4", "--native-layout", "--ksize", "5"], tmp_path)
    assert r.returncode == 0 and "interleaved on disk -> pinned interleaved batch" in r.stdout
    _check_saved_frames(O, tmp_path / "out4", frames, 2)
    r = run([het, "cpu", "0.5", "4", "--frames", "in", "--native-layout", "--planar-out"], tmp_path)
    assert r.returncode != 0 and "exclude each other" in r.stdout
    # errors: nothing found; both devices at once; a frame of another size in the stream
    r = run([het, "cpu", "0.5", "4", "--frames", "nowhere"], tmp_path)
    assert r.returncode != 0 and "Error: no frame files found" in r.stdout
    r = run([het, "both", "0.5", "4", "--frames", "in"], tmp_path)
    assert r.returncode != 0 and "use mode cpu or gpu" in r.stdout
    write_ppm(tmp_path / "in" / "f_0011.ppm", O.lcg_image(20, 20, 3))
    r = run([het, "cpu", "0.5", "4", "--frames", "in"], tmp_path)
    assert r.returncode != 0 and "could not be read (or differs from 52x37x3)" in r.stdout


@pytest.mark.gpu
def test_frames_stream_on_the_gpu(apps, O, tmp_path):
    """--frames on the GPU: helper threads decode into pinned PLANAR batch buffers, the interleave is the GPU's repack-in
    kernel inside mi_blur_submit_planar (reading the frames over the host link), the blur runs in HBM, and the way out is a
    copy or the repack-out kernel.  96 distinct frames; every saved frame == oracle blur of its own input."""
    het, _ = apps
    frames = _write_frames(O, tmp_path / "in", 96, 256, 256, 3)
    r = run([het, "gpu", "1.0", "35", "--frames", "in", "--save-dir", "out", "--csv", "f.csv"], tmp_path)
    assert r.returncode == 0, r.stdout + r.stderr
    for sct in FRAME_SECTIONS + ["Number of images in stream: 96", "Number of batches: 3", "3. GPU DEVICE (processed 96 frames)", "GPU repack-in:",
                                 "GPU blur:", "GPU copy-out:", "9. MI355X KERNEL ROOFLINE", "GPU 0 host placement:"]:
        assert sct in r.stdout, sct
    _check_saved_frames(O, tmp_path / "out", frames, 1)
    row, = read_csv(tmp_path / "f.csv")
    assert row["mode"] == "gpu-frames" and row["images"] == "96" and float(row["gpu_in_ms"]) > 0 and float(row["gpu_kernel_ms"]) > 0
    # planar out: the planar frames are blurred as one-channel images, in place over PCIe (batch server) — no repack at all
    r = run([het, "gpu", "1.0", "20", "--frames", "in", "--save-dir", "out5", "--ksize", "5", "--planar-out", "--slots", "2"], tmp_path)
    assert r.returncode == 0 and "planar frames blurred as one-channel images" in r.stdout and "GPU blur in place:" in r.stdout, r.stdout + r.stderr
    _check_saved_frames(O, tmp_path / "out5", frames, 2)
    # native layout: PPM is interleaved on disk, so it is read straight into the pinned interleaved batch buffer
    r = run([het, "gpu", "1.0", "35", "--frames", "in", "--save-dir", "out6", "--native-layout"], tmp_path)
    assert r.returncode == 0 and "interleaved on disk -> pinned interleaved batch" in r.stdout and "GPU blur in place:" in r.stdout, r.stdout + r.stderr
    _check_saved_frames(O, tmp_path / "out6", frames, 1)
    # two logical GPUs (batch k -> GPU k % 2), grey frames whose plane is not a multiple of 16 pixels (byte repack kernel, ragged blur)
    grey = _write_frames(O, tmp_path / "g", 40, 167, 250, 1, pattern="g%02d")
    r = subprocess.run([het, "gpu", "1.0", "7", "--frames", "g", "--save-dir", "gout", "--gpus", "2"], cwd=tmp_path, capture_output=True, text=True,
                       timeout=600, env=dict(os.environ, MI_BLUR_VIRTUAL_GPUS="1"))
    assert r.returncode == 0, r.stdout + r.stderr
    _check_saved_frames(O, tmp_path / "gout", grey, 1, pattern="g%02d")
    odd = _write_frames(O, tmp_path / "o", 24, 33, 50, 3, pattern="o%02d")
    r = run([het, "gpu", "1.0", "10", "--frames", "o", "--save-dir", "oout", "--planar-out"], tmp_path)
    assert r.returncode == 0, r.stdout + r.stderr
    _check_saved_frames(O, tmp_path / "oout", odd, 1, pattern="o%02d")


def test_gpu_modes_fail_loudly_without_gpu(apps, L, tmp_path):
    if L.mi_blur_device_count() > 0:
        pytest.skip("a GPU is visible here")
    het, spl = apps
    r = run([het, "gpu", "1.0", "35", "--images", "10", "--size", "32x16"], tmp_path)
    assert r.returncode != 0 and "Error: Could not find" in r.stdout                          # :181-184
    r = run([spl, "0.837", "35", "--images", "10", "--size", "32x16"], tmp_path)
    assert r.returncode != 0 and "Error: Could not find both CPU and GPU devices" in r.stdout


# Column names of the reference's aggregated log table, data/approach2/approach2/per_run.csv:1 (33 columns); --csv writes
# exactly these, in this order, followed by three MI355X columns.
REF_PER_RUN_COLUMNS = ("batch_size_file,run,file,mode,gpu_ratio_cfg,cpu_ratio_cfg,images,batches,img_w,img_h,wg_w,wg_h,wall_ms,cpu_images,"
                       "cpu_total_ms,cpu_in_ms,cpu_kernel_ms,cpu_out_ms,cpu_ms_per_img,gpu_images,gpu_total_ms,gpu_in_ms,gpu_kernel_ms,"
                       "gpu_out_ms,gpu_ms_per_img,speedup_gpu_vs_cpu,imbalance_pct,bottleneck,bottleneck_delta_ms,mpix_per_sec,img_per_sec,"
                       "recommended_gpu_ratio,batch_size_log").split(",")
MI355X_CSV_COLUMNS = ["hbm_gbps", "roofline_frac", "n_gpus"]


def read_csv(path):
    rows = [l.split(",") for l in open(path).read().splitlines()]
    assert rows[0] == REF_PER_RUN_COLUMNS + MI355X_CSV_COLUMNS
    assert all(len(r) == len(rows[0]) for r in rows[1:])
    return [dict(zip(rows[0], r)) for r in rows[1:]]


def test_csv_columns_are_the_reference_table_plus_three(apps, tmp_path):
    ref = "/root/reference/data/approach2/approach2/per_run.csv"
    if os.path.exists(ref):                       # build container: the constant above IS the reference file's header
        assert open(ref).readline().strip().split(",") == REF_PER_RUN_COLUMNS
    assert len(REF_PER_RUN_COLUMNS) == 33
    het, _ = apps
    r = run([het, "cpu", "0.5", "35", "--size", "64x48", "--images", "70", "--csv", "run.csv"], tmp_path)
    assert r.returncode == 0, r.stdout + r.stderr
    row, = read_csv(tmp_path / "run.csv")
    assert row["mode"] == "cpu" and row["images"] == "70" and row["batches"] == "2" and (row["img_w"], row["img_h"]) == ("64", "48")
    assert float(row["hbm_gbps"]) == 0.0 and float(row["roofline_frac"]) == 0.0      # no GPU kernel ran in cpu mode


@pytest.mark.gpu
def test_csv_rows_in_gpu_modes(apps, tmp_path):
    """--csv with a GPU in the loop: the reference's 33 columns + hbm_gbps / roofline_frac / n_gpus, filled from the
    dispatch timestamps.  gpu mode, both mode, two logical GPUs (MI_BLUR_VIRTUAL_GPUS on a one-GPU box), the resident
    stream, and split_image_blur append to one file."""
    het, spl = apps
    env = dict(os.environ, MI_BLUR_VIRTUAL_GPUS="1")

    def go(cmd, e=None):
        r = subprocess.run(cmd + ["--csv", "runs.csv"], cwd=tmp_path, capture_output=True, text=True, timeout=600, env=e or os.environ)
        assert r.returncode == 0, r.stdout + r.stderr

    go([het, "gpu", "1.0", "35", "--size", "256x256", "--images", "700"])
    go([het, "both", "0.7", "35", "--size", "256x256", "--images", "700"])
    go([het, "gpu", "1.0", "35", "--size", "256x256", "--images", "700", "--gpus", "2"], env)
    go([het, "gpu", "1.0", "35", "--size", "256x256", "--images", "5000", "--resident"])
    go([spl, "0.837", "35", "--size", "320x240", "--images", "140"])
    rows = read_csv(tmp_path / "runs.csv")
    assert [r["mode"] for r in rows] == ["gpu", "both", "gpu", "gpu", "split"]
    assert [r["n_gpus"] for r in rows] == ["1", "1", "2", "1", "1"]
    for r in rows:
        assert float(r["hbm_gbps"]) > 0 and 0 < float(r["roofline_frac"]) <= 1, r
        assert float(r["wall_ms"]) > 0 and float(r["img_per_sec"]) > 0 and float(r["gpu_kernel_ms"]) > 0, r
        assert abs(float(r["gpu_ratio_cfg"]) + float(r["cpu_ratio_cfg"]) - 1) < 1e-6
    assert rows[0]["gpu_images"] == "700" and rows[0]["cpu_images"] == "0"
    assert rows[1]["gpu_images"] == "480" and rows[1]["cpu_images"] == "220"      # 20 x (11 cpu + 24 gpu), heterogeneous_blur.c:449-458
    assert rows[3]["images"] == "5000" and float(rows[3]["roofline_frac"]) > 0.05   # resident stream: kernel-bound figure


@pytest.mark.gpu
def test_a1_gpu_and_both_modes(apps, O, tmp_path):
    het, _ = apps
    img = O.lcg_image(240, 320, 3)
    write_ppm(tmp_path / "in.ppm", img)
    want = O.blur(img, 1)
    r = run([het, "gpu", "1.0", "35", "--image", "in.ppm", "--images", "500", "--save", "gpu.ppm"], tmp_path)
    assert r.returncode == 0, r.stdout + r.stderr
    for s in A1_SECTIONS + ["Mode: GPU ONLY", "3. GPU DEVICE (processed 500 images)", "9. MI355X KERNEL ROOFLINE"]:
        assert s in r.stdout, s
    assert np.array_equal(read_ppm(tmp_path / "gpu.ppm"), want)
    r = run([het, "both", "0.728", "35", "--image", "in.ppm", "--images", "500", "--save", "both.ppm"], tmp_path)
    assert r.returncode == 0, r.stdout + r.stderr
    for s in ["Mode: HETEROGENEOUS (CPU + GPU)", "GPU ratio: 72.8% GPU, 27.2% CPU", "2. CPU DEVICE (processed 143 images)",
              "3. GPU DEVICE (processed 357 images)", "4. DEVICE COMPARISON", "5. WORKLOAD BALANCE",
              "6. BOTTLENECK IDENTIFICATION", "8. OPTIMAL RATIO RECOMMENDATION", "Run with: ./heterogeneous_blur both"]:
        assert s in r.stdout, s      # 500 images / 35: 14 x (10 cpu + 25 gpu) + last batch of 10 -> (3 cpu, 7 gpu)
    assert np.array_equal(read_ppm(tmp_path / "both.ppm"), want)
    r = run([het, "both", "auto", "35", "--image", "in.ppm", "--images", "500", "--save", "auto.ppm"], tmp_path)
    assert r.returncode == 0 and "Auto-calibrated GPU ratio:" in r.stdout, r.stdout + r.stderr
    assert "Auto ratio after" in r.stdout and "per-batch updates" in r.stdout, r.stdout    # keeps rebalancing per batch
    assert np.array_equal(read_ppm(tmp_path / "auto.ppm"), want)
    # --malloc: the batch buffers are ordinary malloc'd memory, as the reference allocates them (heterogeneous_blur.c:431-432):
    # every submit goes through the library's pinned staging (which the batch server blurs in place); same pixels, same report
    for mode_args, fname in ((["gpu", "1.0", "35"], "gpu_m.ppm"), (["both", "0.728", "35"], "both_m.ppm"), (["gpu", "1.0", "500"], "gpu_m500.ppm")):
        r = run([het] + mode_args + ["--image", "in.ppm", "--images", "500", "--malloc", "--save", fname], tmp_path)
        assert r.returncode == 0 and "Batch buffers: malloc (pageable)" in r.stdout and "7. THROUGHPUT" in r.stdout, r.stdout + r.stderr
        assert np.array_equal(read_ppm(tmp_path / fname), want), mode_args
    r = run([het, "gpu", "1.0", "35", "--size", "256x256", "--images", "5000", "--resident"], tmp_path)
    assert r.returncode == 0 and "9. MI355X KERNEL ROOFLINE" in r.stdout, r.stdout + r.stderr
    r = run([het, "gpu", "1.0", "35", "--size", "256x256", "--images", "5000", "--resident", "--fused"], tmp_path)
    assert r.returncode == 0 and "one fused dispatch, 143 batches counted in" in r.stdout, r.stdout + r.stderr


def _parse_cpulist(text):
    cpus = set()
    for part in text.split(","):
        a, _, b = part.partition("-")
        cpus.update(range(int(a), int(b or a) + 1))
    return cpus


@pytest.mark.gpu
def test_host_work_is_placed_on_the_gpus_socket(apps, pkg, L, tmp_path):
    """Feeder threads, batch-building helpers and pinned buffers keep to the CPUs of their GPU's socket (sysfs
    local_cpulist of the GPU's PCI function); the banner names the list; MI_BLUR_NO_AFFINITY=1 turns it off.  On a one-GPU
    box this is a no-op for performance, but the whole mechanism runs."""
    import threading
    import torch
    het, spl = apps
    p = torch.cuda.get_device_properties(0)
    sysfs = f"/sys/bus/pci/devices/{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
    want_list, want_node = open(sysfs + "/local_cpulist").read().strip(), int(open(sysfs + "/numa_node").read())
    assert pkg.device_cpulist(0) == (want_list, want_node)
    local = _parse_cpulist(want_list)
    seen = {}

    def worker():                                   # a fresh thread: binding it must not touch the test runner's own mask
        seen["before"] = os.sched_getaffinity(0)
        seen["n"] = L.mi_blur_bind_thread_to_device(0)
        seen["after"] = os.sched_getaffinity(0)

    t = threading.Thread(target=worker)
    t.start(); t.join()
    assert seen["after"] == (seen["before"] & local) and seen["n"] == len(seen["after"]) > 0
    buf = L.mi_blur_host_alloc_on(0, 1 << 20)
    assert buf
    L.mi_blur_host_free(buf)
    banner = f"GPU 0 host placement: feeder + batch-building threads and pinned buffers on CPUs {want_list} (NUMA node {want_node})"
    for cmd in ([het, "gpu", "1.0", "35", "--size", "256x256", "--images", "700"], [het, "both", "0.7", "35", "--size", "256x256", "--images", "700"],
                [het, "gpu", "1.0", "35", "--size", "256x256", "--images", "5000", "--resident"], [spl, "0.837", "35", "--size", "320x240", "--images", "140"]):
        r = run(cmd, tmp_path)
        assert r.returncode == 0 and banner in r.stdout, r.stdout + r.stderr
    r = subprocess.run([het, "gpu", "1.0", "35", "--size", "256x256", "--images", "700"], cwd=tmp_path, capture_output=True, text=True,
                       timeout=600, env=dict(os.environ, MI_BLUR_NO_AFFINITY="1"))
    assert r.returncode == 0 and "GPU 0 host placement: not pinned (MI_BLUR_NO_AFFINITY set)" in r.stdout, r.stdout


@pytest.mark.gpu
def test_a2_split_host(apps, O, tmp_path):
    _, spl = apps
    img = O.lcg_image(240, 320, 3)
    write_ppm(tmp_path / "in.ppm", img)
    r = run([spl, "0.837", "35", "--image", "in.ppm", "--images", "200", "--save", "split.ppm"], tmp_path)
    assert r.returncode == 0, r.stdout + r.stderr
    for s in ["========== SPLIT-IMAGE CONFIGURATION ==========", "GPU ratio: 83.7% (rows to GPU)", "Halo size: 1 row(s)",
              "Split row: 39 (CPU: rows 0-38, GPU: rows 39-239)", "CPU: 40 input rows (inc. halo), 39 output rows",
              "GPU: 202 input rows (inc. halo), 201 output rows", "2. CPU DEVICE (processed 200 images - top 39 rows each)",
              "3. GPU DEVICE (processed 200 images - bottom 201 rows each)", "4. DEVICE COMPARISON", "5. WORKLOAD BALANCE",
              "6. BOTTLENECK IDENTIFICATION", "7. THROUGHPUT", "8. SPLIT-IMAGE STATISTICS", "9. OPTIMAL RATIO RECOMMENDATION",
              "Run with: ./split_image_blur"]:
        assert s in r.stdout, s      # geometry lines as in data/approach2/35_run_1.txt:16-20
    assert np.array_equal(read_ppm(tmp_path / "split.ppm"), O.blur(img, 1))
    r = run([spl, "0.5", "16", "--image", "in.ppm", "--images", "64", "--ksize", "5", "--save", "split5.ppm"], tmp_path)
    assert r.returncode == 0 and "Halo size: 2 row(s)" in r.stdout
    assert np.array_equal(read_ppm(tmp_path / "split5.ppm"), O.blur(img, 2))
    # resident row-shard mode on one GPU (no exchange partner: both image edges clamp)
    r = run([spl, "--resident", "--gpus", "1", "--size", "2048x1024", "--iters", "5"], tmp_path)
    assert r.returncode == 0 and "EQUALS the single-device blur" in r.stdout, r.stdout + r.stderr
    assert "halo_exchange_us" in r.stdout and "band_kernel_us" in r.stdout, r.stdout
    # BASELINE configs[4] at full size on one GPU: the printed hash is the reference kernel's (tests/golden k3 8192x8192x3)
    r = run([spl, "--resident", "--gpus", "1", "--size", "8192x8192", "--iters", "5"], tmp_path)
    assert r.returncode == 0 and "EQUALS the single-device blur (fnv d283787bcc5b6dfd)" in r.stdout, r.stdout + r.stderr
    # iterated blur (output shard feeds the next iteration): 3 successive 3x3 blurs == oracle applied 3 times
    r = run([spl, "--resident", "--iterate", "--gpus", "1", "--size", "320x240", "--iters", "3", "--save", "it3.ppm"], tmp_path)
    assert r.returncode == 0 and "EQUALS the single-device blur" in r.stdout, r.stdout + r.stderr
    src = O.lcg_image(240, 320, 3)
    for _ in range(3):
        src = O.blur(src, 1)
    assert np.array_equal(read_ppm(tmp_path / "it3.ppm"), src)


def _bench_line(r):
    import json
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line, printed by rank 0"
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_spawns_its_own_ranks(pkg):
    """`python bench.py --gpus 2` with NO launcher: the parent starts two ranks itself (fresh children, before any GPU call)
    and relays rank 0's one JSON line with n_gpus = 2.  Rehearsal on a one-GPU box: both ranks share cuda:0 and use gloo
    for the barrier (RCCL refuses two ranks on one device).  Without the rehearsal env the same command must FAIL on a
    one-GPU box — never run one rank and report it as two."""
    import sys
    import torch
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    cmd = [sys.executable, os.path.join(pkg.ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--ramp-seconds", "0.05"]
    r = subprocess.run(cmd + ["--images", "700"], cwd=pkg.ROOT, env=dict(env, MI_BLUR_BENCH_BACKEND="gloo", MI_BLUR_BENCH_DEVICE="0"),
                       capture_output=True, text=True, timeout=600)
    d = _bench_line(r)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0 and "cpu_baseline" not in d
    assert d["config"]["images_per_gpu_per_step"] == 700 and d["config"]["images_per_step"] == 1400 and d["roofline"]["bound"] == "hbm"
    assert len(d["config"]["per_rank"]["img_s"]) == 2 and min(d["config"]["per_rank"]["img_s"]) > 0 and d["value"] <= sum(d["config"]["per_rank"]["img_s"]) * 1.001
    # ONE invocation carries BOTH multi-GPU configs and has checked the pixels of both against the reference kernel's hashes:
    # every image of both shards (configs[3]) and each rank's band of the 8192^2 output in both step forms (configs[4])
    assert d["parity"]["status"] == "ok" and d["parity"]["a1_stream"] == dict(d["parity"]["a1_stream"], images_checked_all_ranks=1400, mismatches=0)
    a2 = d["extra"]["a2_8192_rowsplit"]
    assert "configs[4]" in a2["workload"] and a2["rows_per_gpu"] == 4096 and a2["scaling"] == "strong" and a2["img_s"] > 0
    assert set(a2["step_forms_us"]) == {"plain", "overlapped", "pull", "peer"} and a2["quoted_form"] in a2["step_forms_us"] and a2["steps_per_form"] == 200
    assert a2["rccl_ranks"] == 0 and "rehearsal" in a2                    # RCCL exchange leg left out on a one-GPU box, and the line says so
    assert a2["rccl_step_us"] == min(a2["step_forms_us"]["plain"], a2["step_forms_us"]["overlapped"]) and "pulled out of the neighbours" in a2["pull_form"]
    # the pull form is REAL even here (the peer is another process on the same device): it starts from poisoned halo rows
    pa = d["parity"]["a2_8192_rowsplit"]
    assert pa["ok"] and set(pa["band_fnv"]) == {"plain", "overlapped", "pull", "peer"} and set(pa["band_fnv"].values()) == {"0d04249de0140100"}
    # the configs[3] line survives a configs[4] leg that never completes (a stuck exchange on the first real multi-GPU run):
    # the watchdog fires on every rank, rank 0 prints the line with the error in place of the a2 figures, exit code 0
    r = subprocess.run(cmd + ["--images", "700"], cwd=pkg.ROOT, capture_output=True, text=True, timeout=600,
                       env=dict(env, MI_BLUR_BENCH_BACKEND="gloo", MI_BLUR_BENCH_DEVICE="0", MI_BLUR_BENCH_A2_LIMIT_S="0.02"))
    d = _bench_line(r)
    assert d["n_gpus"] == 2 and d["value"] > 0 and "did not complete" in d["extra"]["a2_8192_rowsplit"]["error"]
    assert d["parity"]["a1_stream"]["images_checked_all_ranks"] == 1400 and "configs[3] only" in d["parity"]["status"]
    # the check bites: a rank that holds the wrong shard fails the whole job, and no line is printed
    r = subprocess.run(cmd + ["--images", "700", "--no-extra"], cwd=pkg.ROOT, capture_output=True, text=True, timeout=600,
                       env=dict(env, MI_BLUR_BENCH_BACKEND="gloo", MI_BLUR_BENCH_DEVICE="0", MI_BLUR_BENCH_FAULT="wrong_shard"))
    assert r.returncode != 0 and "PARITY FAILURE" in r.stderr and not [l for l in r.stdout.splitlines() if l.startswith("{")], r.stdout + r.stderr
    # default share at N=2 is 50000 // 2 per GPU (configs[3]); two ranks on one device hold 2 x 9.8 GB
    r = subprocess.run(cmd, cwd=pkg.ROOT, env=dict(env, MI_BLUR_BENCH_BACKEND="gloo", MI_BLUR_BENCH_DEVICE="0"),
                       capture_output=True, text=True, timeout=900)
    d = _bench_line(r)
    assert d["n_gpus"] == 2 and d["config"]["images_per_gpu_per_step"] == 25000 and "configs[3]" in d["config"]["workload"]
    assert d["parity"]["status"] == "ok" and d["parity"]["a1_stream"]["images_checked_all_ranks"] == 50000
    assert d["extra"]["a2_8192_rowsplit"]["img_s"] > 0 and d["parity"]["a2_8192_rowsplit"]["ok"]
    if torch.cuda.device_count() < 2:
        r = subprocess.run(cmd, cwd=pkg.ROOT, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode != 0 and "only 1 HIP device" in r.stderr and not r.stdout.strip(), r.stdout + r.stderr


@pytest.mark.gpu
def test_bench_two_rank_rehearsal_under_launcher(pkg):
    """The driver's form: ranks created by torch.distributed.run (rank env, sharding offsets, barrier, max over ranks,
    rank-0 JSON); nothing is spawned by bench.py itself."""
    import socket
    import sys
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MI_BLUR_BENCH_BACKEND="gloo", MI_BLUR_BENCH_DEVICE="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(pkg.ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--images", "700", "--ramp-seconds", "0.05"]
    r = subprocess.run(cmd, cwd=pkg.ROOT, env=env, capture_output=True, text=True, timeout=600)
    d = _bench_line(r)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0 and "cpu_baseline" not in d
    assert d["config"]["images_per_gpu_per_step"] == 700 and d["roofline"]["bound"] == "hbm"


@pytest.mark.gpu
def test_bench_a2_four_rank_rehearsal_has_middle_ranks(pkg):
    """Four ranks on the one GPU: ranks 1 and 2 have a neighbour on BOTH sides — two IPC handles opened, both halo sides pulled
    / read in place — which no two-rank run exercises.  Every rank's band (2048 rows) in every step form must carry the
    reference kernel's hash for G = 4; one wrong band on any rank fails the job (status is agreed over all ranks)."""
    import json
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(MI_BLUR_BENCH_BACKEND="gloo", MI_BLUR_BENCH_DEVICE="0")
    r = subprocess.run([sys.executable, os.path.join(pkg.ROOT, "bench.py"), "--workload", "a2", "--gpus", "4", "--steps", "10", "--warmup", "2"],
                       cwd=pkg.ROOT, env=env, capture_output=True, text=True, timeout=900)
    d = _bench_line(r)
    assert d["n_gpus"] == 4 and d["scaling"] == "strong" and d["config"]["rows_per_gpu"] == 2048
    assert set(d["config"]["step_forms_us"]) == {"plain", "overlapped", "pull", "peer"} and "rehearsal" in d["config"]
    assert "read" in d["config"]["peer_form"] and "not run" not in d["config"]["pull_form"]
    golden = json.load(open(os.path.join(pkg.ROOT, "tests", "golden", "blur_golden.json")))["bands8192"]["bands"]["4"]
    pa = d["parity"]["a2_8192_rowsplit"]
    assert d["parity"]["status"] == "ok" and pa["ok"] and set(pa["band_fnv"].values()) == {golden[0]}


@pytest.mark.gpu
def test_bench_a2_two_rank_rehearsal(pkg):
    """`bench.py --workload a2 --gpus 2` (BASELINE configs[4] shape of work) spawning its own ranks on a one-GPU box: the
    row split (4096 rows per rank), the timed loop, the per-step decomposition and the overlapped three-launch step all
    run; only the RCCL exchange itself is left out (two ranks cannot share a device), and the line says so."""
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(MI_BLUR_BENCH_BACKEND="gloo", MI_BLUR_BENCH_DEVICE="0")
    r = subprocess.run([sys.executable, os.path.join(pkg.ROOT, "bench.py"), "--workload", "a2", "--gpus", "2", "--steps", "10", "--warmup", "2"],
                       cwd=pkg.ROOT, env=env, capture_output=True, text=True, timeout=600)
    d = _bench_line(r)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0
    assert d["config"]["rows_per_gpu"] == 4096 and d["config"]["halo_bytes_per_neighbour"] == 8192 * 3 and "rehearsal" in d["config"]
    assert d["config"]["steps_per_form"] == 10 and d["config"]["quoted_form"] in ("plain", "overlapped", "pull", "peer") and d["config"]["rccl_ranks"] == 0
    assert abs(d["ms_per_step"] * 1e3 - min(d["config"]["step_forms_us"].values())) < 0.06      # value quotes the faster form (ms rounded to 1e-4)
    assert d["parity"]["status"] == "ok" and d["parity"]["a2_8192_rowsplit"]["band_fnv"]["plain"] == "0d04249de0140100"
    dec = d["config"]["step_decomposition"]
    assert dec["band_kernel_us"] > 0 and dec["halo_exchange_us"] >= 0 and dec["step_us_plain"] > 0 and dec["step_us_overlapped"] > 0
    assert 0 < dec["band_kernel_frac"] <= 1 and dec["band_kernel_us_max_over_ranks"] >= dec["band_kernel_us"] - 1e-6


@pytest.mark.gpu
def test_bench_default_line_carries_every_single_gpu_config(pkg):
    """The default N=1 line: configs[1] as `value`, plus configs[2] (hd1080_5x5 with its own frac), configs[4] at N=1
    (a2_8192_1gpu, whose output hashes like the reference kernel's), the PCIe-inclusive rate at batch 35 and 500, the
    sustained repeat, roofline and cpu_baseline."""
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(pkg.ROOT, "bench.py"), "--steps", "20", "--warmup", "5"], cwd=pkg.ROOT, env=env,
                       capture_output=True, text=True, timeout=900)
    d = _bench_line(r)
    assert d["n_gpus"] == 1 and d["dtype"] == "u8" and "configs[1]" in d["config"]["workload"] and d["vs_baseline"] is None
    assert 0 < d["roofline"]["frac"] <= 1 and d["roofline"]["kernel"] == "blur_fused_tail_kernel" and d["roofline"]["launches_timed"] == 20
    cb = d["cpu_baseline"]
    assert cb["value"] > 0 and cb["cores"] >= 1 and cb["kind"] == "port"
    assert cb["one_thread"]["cores"] == 1 and 0 < cb["one_thread"]["value"] <= cb["value"] * 1.05
    assert cb["product_cpu_device"]["value"] > 0 and cb["product_cpu_device"]["one_thread"]["cores"] == 1
    assert d["parity"]["status"] == "ok" and d["parity"]["a1_stream"]["images_checked_all_ranks"] == 5000
    bc = d["batch_completion_us"]
    assert bc["batches"] == 143 and 0 < bc["first"] <= bc["p50"] <= bc["last"] and bc["passes"] >= 20
    assert bc["first"] < 0.6 * bc["last"], bc          # batches are visible to the host long before the dispatch ends
    assert d["release_mode_us"]["dispatch_us"] > d["roofline"]["avg_launch_us"] and d["release_mode_us"]["batches_counted_in"] == 143
    assert d["sustained"]["seconds"] >= 1.0 and d["sustained_img_s"] > 0
    ex = d["extra"]
    assert set(ex) >= {"hd1080_5x5", "a2_8192_1gpu", "e2e_pcie_inclusive", "one_launch_5000_images", "copy_kernel_same_box"}
    assert 0 < ex["copy_kernel_same_box"]["frac"] <= 1
    assert 0 < ex["hd1080_5x5"]["frac"] <= 1 and 0 < ex["a2_8192_1gpu"]["frac"] <= 1
    assert ex["a2_8192_1gpu"]["out_fnv"] == "d283787bcc5b6dfd"             # tests/golden k3 8192x8192x3 (reference kernel)
    assert ex["e2e_pcie_inclusive"]["batch_35"]["img_s"] > 0 and ex["e2e_pcie_inclusive"]["batch_500"]["img_s"] > 0
    assert ex["e2e_pcie_inclusive"]["hd1080_5x5_batch_8"]["img_s"] > 0 and ex["e2e_pcie_inclusive"]["hd1080_5x5_batch_8"]["kernel"] == "blur_server_kernel"
    pg = ex["e2e_pcie_inclusive"]["batch_35_pageable"]          # malloc'd caller buffers: through the pinned staging, served by the batch server
    assert pg["img_s"] > 0 and pg["kernel"] == "blur_server_kernel" and pg["zero_copy_submits"] > 0 and "pageable" in pg["buffers"]
    assert d["per_batch_launches"]["launches_per_step"] == 143


@pytest.mark.gpu
def test_multi_gpu_sharding_on_virtual_gpus(apps, O, tmp_path):
    """The hosts' --gpus G sharding (Approach 1: contiguous image shares per batch; Approach 2: GPU rows shared out
    with their own halos) on G logical GPUs that share the one physical device (MI_BLUR_VIRTUAL_GPUS=1)."""
    het, spl = apps
    img = O.lcg_image(240, 320, 3)
    write_ppm(tmp_path / "in.ppm", img)
    env = dict(os.environ, MI_BLUR_VIRTUAL_GPUS="1")

    def runv(cmd):
        return subprocess.run(cmd, cwd=tmp_path, capture_output=True, text=True, timeout=600, env=env)

    r = runv([het, "gpu", "1.0", "35", "--image", "in.ppm", "--images", "300", "--gpus", "4", "--save", "g4.ppm"])
    assert r.returncode == 0, r.stdout + r.stderr
    assert "3. GPU DEVICE (processed 300 images)" in r.stdout and r.stdout.count("logical GPU") >= 4
    assert np.array_equal(read_ppm(tmp_path / "g4.ppm"), O.blur(img, 1))
    r = runv([het, "both", "0.8", "35", "--image", "in.ppm", "--images", "300", "--gpus", "3", "--ksize", "5", "--save", "b3.ppm"])
    assert r.returncode == 0 and np.array_equal(read_ppm(tmp_path / "b3.ppm"), O.blur(img, 2)), r.stdout + r.stderr
    r = runv([spl, "0.837", "35", "--image", "in.ppm", "--images", "140", "--gpus", "3", "--save", "s3.ppm"])
    assert r.returncode == 0, r.stdout + r.stderr
    assert np.array_equal(read_ppm(tmp_path / "s3.ppm"), O.blur(img, 1))
    r = runv([spl, "0.5", "16", "--image", "in.ppm", "--images", "64", "--gpus", "2", "--ksize", "5", "--save", "s2.ppm"])
    assert r.returncode == 0 and np.array_equal(read_ppm(tmp_path / "s2.ppm"), O.blur(img, 2)), r.stdout + r.stderr


@pytest.mark.gpu
def test_resident_row_shards_on_virtual_gpus(apps, O, tmp_path):
    """The resident row-shard flow of split_image_blur (BASELINE config 5 shape of work) with 4 shards on the one
    physical GPU: halo rows travel between shards through the peer-copy transport (RCCL refuses two ranks on one
    device), results must equal the single-device blur and the oracle, also when iterated (recurring exchange)."""
    _, spl = apps
    env = dict(os.environ, MI_BLUR_VIRTUAL_GPUS="1")
    src = O.lcg_image(240, 320, 3)
    for ksize, radius in (("3", 1), ("5", 2)):
        r = subprocess.run([spl, "--resident", "--gpus", "4", "--size", "320x240", "--ksize", ksize, "--iters", "3", "--save", "one.ppm"],
                           cwd=tmp_path, capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0 and "EQUALS the single-device blur" in r.stdout and "hipMemcpyPeerAsync" in r.stdout, r.stdout + r.stderr
        assert np.array_equal(read_ppm(tmp_path / "one.ppm"), O.blur(src, radius))
        r = subprocess.run([spl, "--resident", "--iterate", "--gpus", "4", "--size", "320x240", "--ksize", ksize, "--iters", "4", "--save", "it.ppm"],
                           cwd=tmp_path, capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0 and "EQUALS the single-device blur" in r.stdout, r.stdout + r.stderr
        want = src
        for _ in range(4):
            want = O.blur(want, radius)
        assert np.array_equal(read_ppm(tmp_path / "it.ppm"), want)
        # the same with the exchange on its own stream, hidden behind the interior rows (edge rows follow the halos)
        for extra, fname, passes in ((["--iterate"], "ov_it.ppm", 5), ([], "ov_one.ppm", 1)):
            r = subprocess.run([spl, "--resident", "--overlap", "--gpus", "4", "--size", "320x240", "--ksize", ksize, "--iters", "5",
                                "--save", fname] + extra, cwd=tmp_path, capture_output=True, text=True, timeout=600, env=env)
            assert r.returncode == 0 and "EQUALS the single-device blur" in r.stdout, r.stdout + r.stderr
            assert "halo exchange hidden behind the interior rows" in r.stdout
            want = src
            for _ in range(passes):
                want = O.blur(want, radius)
            assert np.array_equal(read_ppm(tmp_path / fname), want), (ksize, extra)
        # the third transport: halo rows PULLED by one small kernel per shard that reads its neighbours' rows (ordering by the
        # same events as the pushes) — single blur, iterated, and iterated with the exchange hidden behind the interior rows
        for extra, fname, passes in (([], "pl_one.ppm", 1), (["--iterate"], "pl_it.ppm", 4), (["--iterate", "--overlap"], "pl_ov.ppm", 4)):
            r = subprocess.run([spl, "--resident", "--transport", "pull", "--gpus", "4", "--size", "320x240", "--ksize", ksize, "--iters", "4",
                                "--save", fname] + extra, cwd=tmp_path, capture_output=True, text=True, timeout=600, env=env)
            assert r.returncode == 0 and "EQUALS the single-device blur" in r.stdout and "pulled by one kernel per GPU" in r.stdout, r.stdout + r.stderr
            want = src
            for _ in range(passes):
                want = O.blur(want, radius)
            assert np.array_equal(read_ppm(tmp_path / fname), want), (ksize, extra, "pull")
        # the fourth: no exchange at all — every shard's band kernel reads its neighbours' rows in place (one launch per GPU per
        # step; the shard's own halo rows are poison throughout).  Single blur, and iterated (steps ordered between the GPUs).
        for extra, fname, passes in (([], "pr_one.ppm", 1), (["--iterate"], "pr_it.ppm", 4)):
            r = subprocess.run([spl, "--resident", "--transport", "peer", "--gpus", "4", "--size", "320x240", "--ksize", ksize, "--iters", "4",
                                "--save", fname] + extra, cwd=tmp_path, capture_output=True, text=True, timeout=600, env=env)
            assert r.returncode == 0 and "EQUALS the single-device blur" in r.stdout and "read in place by the band kernel" in r.stdout, r.stdout + r.stderr
            want = src
            for _ in range(passes):
                want = O.blur(want, radius)
            assert np.array_equal(read_ppm(tmp_path / fname), want), (ksize, extra, "peer")


@pytest.mark.gpu
def test_hosts_with_ragged_frame_sizes(apps, O, tmp_path):
    """Frames whose rows are not a multiple of 16 bytes (250x167x3: pitch 750) through both hosts: zero-copy A1 batches take
    the ragged form of the tiled kernel, A2's strided bands take the staged path; pixels must equal the oracle."""
    het, spl = apps
    img = O.lcg_image(167, 250, 3)
    write_ppm(tmp_path / "in.ppm", img)
    for ksize, radius in (("3", 1), ("5", 2)):
        want = O.blur(img, radius)
        r = run([het, "gpu", "1.0", "35", "--image", "in.ppm", "--images", "300", "--ksize", ksize, "--save", "g.ppm"], tmp_path)
        assert r.returncode == 0, r.stdout + r.stderr
        assert np.array_equal(read_ppm(tmp_path / "g.ppm"), want)
        r = run([het, "both", "0.6", "35", "--image", "in.ppm", "--images", "300", "--ksize", ksize, "--save", "b.ppm"], tmp_path)
        assert r.returncode == 0, r.stdout + r.stderr
        assert np.array_equal(read_ppm(tmp_path / "b.ppm"), want)
        r = run([spl, "0.7", "16", "--image", "in.ppm", "--images", "96", "--ksize", ksize, "--save", "s.ppm"], tmp_path)
        assert r.returncode == 0, r.stdout + r.stderr
        assert np.array_equal(read_ppm(tmp_path / "s.ppm"), want)
