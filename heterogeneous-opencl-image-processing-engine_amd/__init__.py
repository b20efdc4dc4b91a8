"""MI355X-native image-stream blur engine — build + ctypes binding of the C ABI.

The product is ``libmi_blur.so`` (HIP kernels for gfx950 behind ``include/mi_blur.h``)
and the two C++ hosts under ``apps/``.  This module is only the harness-side binding
used by tests/ and bench.py: it adds no behaviour of its own and has NO CPU fallback —
if the shared library is missing or a GPU entry point is called without a GPU, it raises.

The directory name contains hyphens, so import it with ``__graft_entry__.load_package()``
(which registers it as module ``hoipe_amd``).
"""
from __future__ import annotations

import ctypes as C
import os
import re
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG_DIR)
CSRC = os.path.join(PKG_DIR, "csrc")
APPS = os.path.join(PKG_DIR, "apps")
LIB_PATH = os.path.join(PKG_DIR, "libmi_blur.so")
HEADER = os.path.join(ROOT, "include", "mi_blur.h")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# Kernel arguments in host memory: the launch-rate-bound batch stream issues ~2x faster (see the constructor in
# csrc/mi_blur_api.cpp).  Must be in the environment before the HIP runtime initialises; the user's value wins.
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "0")
ARCH = "gfx950"

DEVICE_CPU = -1
VARIANT_AUTO, VARIANT_GENERIC, VARIANT_TILED, VARIANT_STREAM, VARIANT_DIRECT = 0, 1, 2, 3, 4
OK, ERR_INVALID, ERR_NO_DEVICE, ERR_NOMEM, ERR_STATE, ERR_UNSUPPORTED = 0, -1, -2, -3, -4, -5
UNIQUE_ID_BYTES = 128
PEER_HANDLE_BYTES = 64


def _newer(target: str, sources: list[str]) -> bool:
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(s) <= t for s in sources)


def build_native(force: bool = False, verbose: bool = False) -> str:
    """Compile libmi_blur.so (hipcc, --offload-arch=gfx950) and the C++ hosts, in-tree."""
    srcs = [os.path.join(CSRC, f) for f in ("blur_kernels.hip", "layout_kernels.hip", "mi_blur_api.cpp", "cpu_device.cpp")]
    deps = srcs + [os.path.join(CSRC, f) for f in ("blur_launch.h", "cpu_device.h")] + [HEADER]
    if force or not _newer(LIB_PATH, deps):
        cmd = [HIPCC, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wextra",
               "-o", LIB_PATH] + srcs + ["-ldl", "-lpthread"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    if os.path.isdir(APPS):
        for app in ("heterogeneous_blur", "split_image_blur"):
            src = os.path.join(APPS, app + ".cpp")
            exe = os.path.join(APPS, app)
            common = [os.path.join(APPS, f) for f in os.listdir(APPS) if f.endswith((".h", ".hpp"))]
            if os.path.exists(src) and (force or not _newer(exe, [src, LIB_PATH, HEADER] + common)):
                cmd = [HIPCC, "-O2", "-std=c++17", "-Wall", "-Wextra", "-I", os.path.join(ROOT, "include"), "-o", exe, src,
                       "-L", PKG_DIR, "-lmi_blur", f"-Wl,-rpath,$ORIGIN/..", "-lpthread"]
                if verbose:
                    print(" ".join(cmd), flush=True)
                subprocess.run(cmd, check=True)
    return LIB_PATH


def declared_symbols() -> list[str]:
    """Every function include/mi_blur.h declares."""
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mi_blur_[a-z0-9_]+)\s*\(", text)))


class Timing(C.Structure):
    _fields_ = [("h2d_ms", C.c_double), ("kernel_ms", C.c_double), ("d2h_ms", C.c_double),
                ("bytes_h2d", C.c_uint64), ("bytes_d2h", C.c_uint64), ("bytes_alg", C.c_uint64),
                ("images", C.c_uint64), ("launches", C.c_uint64)]

    def as_dict(self) -> dict:
        return {n: getattr(self, n) for n, _ in self._fields_}


class A2Geometry(C.Structure):
    _fields_ = [(n, C.c_int) for n in
                ("split_row", "cpu_input_rows", "cpu_output_rows", "gpu_input_rows", "gpu_output_rows")]


class Band(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("row_begin", "row_end", "halo_top", "halo_bottom")]


class MiBlurError(RuntimeError):
    def __init__(self, status: int, what: str):
        super().__init__(f"{what}: status {status} ({lib().mi_blur_strerror(status).decode()})")
        self.status = status


_lib = None


def lib() -> C.CDLL:
    """Load libmi_blur.so (raises if it has not been built — there is no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FileNotFoundError(f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'`")
    # torch wheels bundle their own libamdhip64/libhsa-runtime64 and load them by path.  If this
    # library were loaded first it would pull /opt/rocm's copies in and the process would hold
    # TWO HIP runtimes fighting over the device (measured: device count 0 / torch unavailable).
    # Harness processes use torch for device memory, so load torch first: libmi_blur.so then binds
    # to the already-loaded runtime by soname.  (The C++ hosts never load torch: one runtime.)
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, i, u8p = C.c_void_p, C.c_int, C.c_void_p
    sig = {
        "mi_blur_strerror": (C.c_char_p, [i]),
        "mi_blur_version": (i, []),
        "mi_blur_last_kernel": (C.c_char_p, []),
        "mi_blur_device_count": (i, []),
        "mi_blur_set_option": (i, [C.c_char_p, i]),
        "mi_blur_enqueue": (i, [u8p, u8p, i, i, i, i, i, vp]),
        "mi_blur_enqueue_band": (i, [u8p, u8p, i, i, i, i, i, i, vp]),
        "mi_blur_enqueue_ex": (i, [u8p, u8p, i, i, i, i, i, i, i, i, vp]),
        "mi_blur_planar_to_interleaved": (i, [u8p, u8p, i, i, i, i, vp]),
        "mi_blur_interleaved_to_planar": (i, [u8p, u8p, i, i, i, i, vp]),
        "mi_blur_create": (i, [C.POINTER(vp), i, i, i, i, i, i, i, i]),
        "mi_blur_destroy": (None, [vp]),
        "mi_blur_host_alloc": (vp, [C.c_size_t]),
        "mi_blur_host_free": (None, [vp]),
        "mi_blur_host_register": (i, [vp, C.c_size_t]),
        "mi_blur_host_unregister": (i, [vp]),
        "mi_blur_device_cpulist": (i, [i, C.c_char_p, C.c_size_t, C.POINTER(i)]),
        "mi_blur_bind_thread_to_device": (i, [i]),
        "mi_blur_host_alloc_on": (vp, [i, C.c_size_t]),
        "mi_blur_submit": (i, [vp, u8p, u8p, i]),
        "mi_blur_submit_band": (i, [vp, u8p, u8p, i, i, i]),
        "mi_blur_submit_bands": (i, [vp, u8p, u8p, i, C.c_size_t, i, i, i]),
        "mi_blur_submit_planar": (i, [vp, u8p, u8p, i, i]),
        "mi_blur_wait_oldest": (i, [vp]),
        "mi_blur_sync": (i, [vp, C.POINTER(Timing)]),
        "mi_blur_reset_timing": (None, [vp]),
        "mi_blur_get_timing": (C.c_int, [vp, C.c_void_p]),
        "mi_blur_resident_run_fused": (C.c_int, [vp, C.c_int, C.c_int, C.c_int]),
        "mi_blur_resident_batches_done": (C.c_int, [vp]),
        "mi_blur_resident_peek": (C.c_int, [vp, C.c_int, u8p, C.c_int]),
        "mi_blur_zero_copy_launches": (C.c_uint64, [vp]),
        "mi_blur_resident_alloc": (i, [vp, i]),
        "mi_blur_resident_placement": (i, [vp, C.POINTER(C.c_float), i, C.POINTER(i)]),
        "mi_blur_resident_fill_synthetic": (i, [vp, i]),
        "mi_blur_resident_upload": (i, [vp, i, u8p, i]),
        "mi_blur_resident_download": (i, [vp, i, u8p, i]),
        "mi_blur_resident_in": (vp, [vp]),
        "mi_blur_resident_out": (vp, [vp]),
        "mi_blur_resident_run": (i, [vp, i, i, i]),
        "mi_blur_timed_coverage": (None, [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
        "mi_blur_cpu_run": (i, [u8p, u8p, i, i, i, i, i, i]),
        "mi_blur_fill_synthetic": (None, [u8p, i, i, i, i, i, i]),
        "mi_blur_fnv1a64": (C.c_uint64, [u8p, C.c_size_t]),
        "mi_blur_debug_zc_trace": (i, [vp, C.POINTER(C.c_uint64), i, C.POINTER(i), C.POINTER(C.c_uint)]),
        "mi_blur_debug_xcd_times": (i, [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), i]),
        "mi_blur_a1_partition": (None, [i, i, C.c_float, C.POINTER(i), C.POINTER(i)]),
        "mi_blur_shard_range": (None, [C.c_longlong, i, i, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]),
        "mi_blur_a2_split": (None, [i, C.c_float, i, C.POINTER(A2Geometry)]),
        "mi_blur_band_of": (None, [i, i, i, i, C.POINTER(Band)]),
        "mi_blur_comm_unique_id": (i, [vp]),
        "mi_blur_comm_init_rank": (i, [C.POINTER(vp), i, i, vp]),
        "mi_blur_comm_init_all": (i, [C.POINTER(vp), i, C.POINTER(i)]),
        "mi_blur_comm_init_p2p": (i, [C.POINTER(vp), i, C.POINTER(i)]),
        "mi_blur_comm_init_pull": (i, [C.POINTER(vp), i, C.POINTER(i)]),
        "mi_blur_comm_destroy": (None, [vp]),
        "mi_blur_comm_info": (i, [vp, C.POINTER(i), C.POINTER(i), C.POINTER(i)]),
        "mi_blur_halo_exchange": (i, [vp, u8p, i, i, i, i, vp]),
        "mi_blur_peer_export": (i, [vp, vp, C.POINTER(C.c_uint64)]),
        "mi_blur_peer_open": (i, [vp, C.c_uint64, C.POINTER(vp)]),
        "mi_blur_peer_close": (i, [vp, C.c_uint64]),
        "mi_blur_halo_pull": (i, [u8p, u8p, u8p, i, i, i, i, vp]),
        "mi_blur_enqueue_band_peer": (i, [u8p, u8p, i, i, i, i, i, i, u8p, u8p, vp]),
        "mi_blur_halo_exchange_all": (i, [C.POINTER(vp), i, C.POINTER(vp), i, i, C.POINTER(i), i, C.POINTER(vp)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)      # AttributeError if the library does not export it
        fn.restype, fn.argtypes = res, args
    _lib = L
    return L


def check(status: int, what: str = "mi_blur") -> None:
    if status != OK:
        raise MiBlurError(status, what)


# ---- thin conveniences over the ABI (no logic beyond argument marshalling) -----------------
def a1_partition(mode: int, batch_count: int, gpu_ratio: float) -> tuple[int, int]:
    nc, ng = C.c_int(), C.c_int()
    lib().mi_blur_a1_partition(mode, batch_count, gpu_ratio, C.byref(nc), C.byref(ng))
    return nc.value, ng.value


def shard_range(n_units: int, g: int, G: int) -> tuple[int, int]:
    b, e = C.c_longlong(), C.c_longlong()
    lib().mi_blur_shard_range(n_units, g, G, C.byref(b), C.byref(e))
    return b.value, e.value


def device_cpulist(device: int) -> tuple[str, int]:
    """(local CPU list as sysfs spells it, NUMA node) of a GPU; ("", -1) when the topology is not exposed."""
    buf = C.create_string_buffer(512)
    node = C.c_int(-1)
    rc = lib().mi_blur_device_cpulist(device, buf, len(buf), C.byref(node))
    return (buf.value.decode() if rc == OK else "", node.value)


def a2_split(height: int, gpu_ratio: float, halo: int = 1) -> dict:
    g = A2Geometry()
    lib().mi_blur_a2_split(height, gpu_ratio, halo, C.byref(g))
    return {n: getattr(g, n) for n, _ in A2Geometry._fields_}


def band_of(height: int, radius: int, g: int, G: int) -> dict:
    b = Band()
    lib().mi_blur_band_of(height, radius, g, G, C.byref(b))
    return {n: getattr(b, n) for n, _ in Band._fields_}


class Context:
    """RAII wrapper of mi_blur_ctx for tests/bench."""

    def __init__(self, device: int, width: int, height: int, channels: int, radius: int = 1,
                 max_batch: int = 1, n_slots: int = 2, n_threads: int = 0):
        self.h = C.c_void_p()
        check(lib().mi_blur_create(C.byref(self.h), device, width, height, channels, radius, max_batch,
                                   n_slots, n_threads), "mi_blur_create")
        self.shape = (height, width, channels)
        self.radius = radius

    def close(self) -> None:
        if self.h:
            lib().mi_blur_destroy(self.h)
            self.h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def submit(self, host_in, host_out, n_images: int) -> None:
        check(lib().mi_blur_submit(self.h, host_in, host_out, n_images), "mi_blur_submit")

    def submit_band(self, host_in, host_out, band_rows: int, halo_top: int, halo_bottom: int) -> None:
        check(lib().mi_blur_submit_band(self.h, host_in, host_out, band_rows, halo_top, halo_bottom),
              "mi_blur_submit_band")

    def submit_planar(self, host_planar_in, host_out, n_images: int, planar_out: bool = False) -> None:
        check(lib().mi_blur_submit_planar(self.h, host_planar_in, host_out, n_images, 1 if planar_out else 0), "mi_blur_submit_planar")

    def wait_oldest(self) -> None:
        check(lib().mi_blur_wait_oldest(self.h), "mi_blur_wait_oldest")

    def sync(self) -> dict:
        t = Timing()
        check(lib().mi_blur_sync(self.h, C.byref(t)), "mi_blur_sync")
        return t.as_dict()

    def timing(self) -> dict:
        """Non-blocking snapshot of the buckets harvested so far."""
        t = Timing()
        check(lib().mi_blur_get_timing(self.h, C.byref(t)), "mi_blur_get_timing")
        return t.as_dict()

    def reset_timing(self) -> None:
        lib().mi_blur_reset_timing(self.h)

    def resident_alloc(self, pool_images: int) -> None:
        check(lib().mi_blur_resident_alloc(self.h, pool_images), "mi_blur_resident_alloc")

    def resident_placement(self) -> dict:
        """What resident_alloc measured when it chose the pool: per-launch us of each candidate placement, and the one kept."""
        ms = (C.c_float * 64)()
        kept = C.c_int()
        n = lib().mi_blur_resident_placement(self.h, ms, 64, C.byref(kept))
        return {"candidates_us": [round(ms[k] * 1e3, 2) for k in range(max(n, 0))], "kept": kept.value}

    def resident_fill_synthetic(self, first_index: int = 0) -> None:
        check(lib().mi_blur_resident_fill_synthetic(self.h, first_index), "mi_blur_resident_fill_synthetic")

    def resident_upload(self, pool_index: int, host_in, n_images: int) -> None:
        check(lib().mi_blur_resident_upload(self.h, pool_index, host_in, n_images), "mi_blur_resident_upload")

    def resident_download(self, pool_index: int, host_out, n_images: int) -> None:
        check(lib().mi_blur_resident_download(self.h, pool_index, host_out, n_images), "mi_blur_resident_download")

    def resident_run(self, n_images: int, batch: int, timed: int | bool = 0) -> None:
        """timed: 0/False none, 1/True every launch, n every n-th launch carries timestamp events."""
        check(lib().mi_blur_resident_run(self.h, n_images, batch, int(timed)), "mi_blur_resident_run")

    def resident_run_fused(self, n_images: int, batch: int, timed: bool = False, watch: bool = False) -> None:
        """watch: a one-wave kernel keeps the pass's progress in pinned host memory (resident_batches_done then costs no HIP call)."""
        check(lib().mi_blur_resident_run_fused(self.h, n_images, batch, (1 if timed else 0) | (2 if watch else 0)), "mi_blur_resident_run_fused")

    def resident_peek(self, pool_index: int, host_out, n_images: int) -> None:
        check(lib().mi_blur_resident_peek(self.h, pool_index, host_out, n_images), "mi_blur_resident_peek")

    def resident_batches_done(self) -> int:
        """Leading batches of the latest fused pass that are complete; raises on a negative status."""
        n = int(lib().mi_blur_resident_batches_done(self.h))
        if n < 0:
            raise MiBlurError(n, "mi_blur_resident_batches_done")
        return n

    def wait_batches(self, want: int, timeout_s: float = 30.0) -> int:
        """Poll until at least `want` leading batches are done; TimeoutError (with the last count) after timeout_s."""
        import time
        deadline = time.monotonic() + timeout_s
        n = self.resident_batches_done()
        while n < want:
            if time.monotonic() > deadline:
                raise TimeoutError(f"fused stream: {n} of {want} batches counted in after {timeout_s:.0f} s")
            n = self.resident_batches_done()
        return n

    def timed_coverage(self) -> tuple[int, int]:
        n, b = C.c_uint64(), C.c_uint64()
        lib().mi_blur_timed_coverage(self.h, C.byref(n), C.byref(b))
        return n.value, b.value


def blur(images, ksize: int = 3, device: int = 0, batch: int = 0):
    """Convenience for Python callers: the reference's blur of a stack of interleaved uint8 images, numpy in -> numpy out.

    images: (H, W), (H, W, C) or (N, H, W, C) uint8.  ksize 3 (the reference kernel, gaussian_kernel.cl:36-41) or
    5.  device: HIP ordinal, or DEVICE_CPU for the host-thread device.  batch: images per submit (0 = all at once, at most 4096).
    Goes through mi_blur_create / mi_blur_submit / mi_blur_sync like any host; there is no other code path behind it."""
    import numpy as np
    a = np.ascontiguousarray(images)
    if a.dtype != np.uint8 or a.ndim not in (2, 3, 4):
        raise ValueError("blur: a uint8 array of shape (H, W), (H, W, C) or (N, H, W, C)")
    if ksize not in (3, 5):
        raise ValueError("blur: ksize 3 or 5")
    single = a.ndim in (2, 3)
    if a.ndim == 2:
        a = a[None, :, :, None]
    elif a.ndim == 3:
        a = a[None]
    n, h, w, c = a.shape
    out = np.empty_like(a)
    if n == 0 or a.size == 0:
        return out[0] if single else out
    per = min(n, batch if batch > 0 else 4096)
    isz = h * w * c
    with Context(device, w, h, c, (ksize - 1) // 2, max_batch=per, n_slots=2) as ctx:
        for i in range(0, n, per):
            m = min(per, n - i)
            ctx.submit(a.ctypes.data + i * isz, out.ctypes.data + i * isz, m)
        ctx.sync()
    return out[0] if single else out
