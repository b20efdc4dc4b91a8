"""Host code under AddressSanitizer + UBSan (GPU sanitizers are not available on this pool): the product's
vectorised CPU device against the oracle's C restatement on random shapes, exact-size heap buffers."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_cpu_device_clean_under_asan_ubsan(pkg, tmp_path):
    exe = tmp_path / "san_cpu"
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           "-fno-omit-frame-pointer", "-I", pkg.CSRC, os.path.join(ROOT, "tests", "san_cpu_device.cpp"),
           os.path.join(pkg.CSRC, "cpu_device.cpp"), "-x", "c", os.path.join(ROOT, "oracle", "oracle_blur.c"),
           "-lpthread", "-o", str(exe)]
    subprocess.run(cmd, check=True, capture_output=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "random cases clean" in r.stdout, r.stdout + r.stderr
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_host_thread_pools_clean_under_tsan(pkg, tmp_path):
    """The hosts' helper-thread pools (batch-building Replicator, frame decode/save TaskPool) under ThreadSanitizer:
    every item exactly once, no data race between the feeder thread and the helpers."""
    exe = tmp_path / "san_pools"
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-fsanitize=thread", "-fno-omit-frame-pointer", "-I", pkg.APPS,
           "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "san_host_pools.cpp"), "-lpthread", "-o", str(exe)]
    subprocess.run(cmd, check=True, capture_output=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "pools clean" in r.stdout, r.stdout + r.stderr
    assert "WARNING: ThreadSanitizer" not in r.stderr
