"""ctypes front-end to the CPU oracle and an independent numpy restatement.

TEST INFRASTRUCTURE, NOT PRODUCT.  Only tests/, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import this module; the product library
(libmi_blur.so) never touches anything under oracle/.

Parity status: the 3x3 path is PINNED (oracle == unmodified reference kernel in
oracle/_ref, == tests/golden/blur_golden.json); the 5x5 path is "parity
unpinned" (no 5x5 kernel exists in the reference, gaussian_kernel.cl:36-41).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "liboracle_blur.so")
REF_PATH = os.path.join(HERE, "_ref", "libref_blur.so")
REFERENCE_ROOT = "/root/reference"

LCG_SEED = 0x9E3779B9


def build(ref: bool | None = None) -> None:
    """Compile liboracle_blur.so, and oracle/_ref when the reference tree is present."""
    targets = ["all"]
    if ref is None:
        ref = os.path.exists(os.path.join(REFERENCE_ROOT, "gaussian_kernel.cl"))
    if ref:
        targets.append("ref")
    subprocess.run(["make", "-s", "-C", HERE] + targets, check=True)


class _Geom(C.Structure):
    _fields_ = [(n, C.c_int) for n in
                ("split_row", "cpu_input_rows", "cpu_output_rows", "gpu_input_rows", "gpu_output_rows")]


_lib = None
_ref = None
_u8p = C.POINTER(C.c_uint8)


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build(ref=False)
        L = C.CDLL(LIB_PATH)
        L.oracle_blur3_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.oracle_blur5_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.oracle_blur_int.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.oracle_blur_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.oracle_a1_partition.argtypes = [C.c_int, C.c_int, C.c_float, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.oracle_a2_geometry.argtypes = [C.c_int, C.c_float, C.c_int, C.POINTER(_Geom)]
        L.oracle_a2_split_blur.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.oracle_splitk_blur.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.oracle_planar_to_interleaved.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.oracle_interleaved_to_planar.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.oracle_lcg_fill.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32]
        L.oracle_fnv1a64.argtypes = [C.c_void_p, C.c_size_t]
        L.oracle_fnv1a64.restype = C.c_uint64
        _lib = L
    return _lib


def ref_available() -> bool:
    return os.path.exists(REF_PATH)


def ref() -> C.CDLL:
    """The unmodified reference kernel (oracle/_ref).  Raises if it was never built."""
    global _ref
    if _ref is None:
        if not ref_available():
            raise FileNotFoundError(f"{REF_PATH} missing: run `make -C oracle ref` in the build container")
        R = C.CDLL(REF_PATH)
        R.ref_gaussian_blur.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
        R.ref_split_image_blur.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        _ref = R
    return _ref


def _chk(img: np.ndarray) -> tuple[int, int, int]:
    assert img.dtype == np.uint8 and img.ndim == 3 and img.flags.c_contiguous, "expect HxWxC uint8 C-contiguous"
    return img.shape


def _p(a: np.ndarray) -> int:
    return a.ctypes.data


# --------------------------------------------------------------------------- C oracle
def blur(img: np.ndarray, radius: int = 1) -> np.ndarray:
    """Integer-form oracle, HxWxC uint8 (gaussian_kernel.cl:19-72 for radius 1)."""
    h, w, c = _chk(img)
    out = np.empty_like(img)
    rc = lib().oracle_blur_int(_p(img), _p(out), w, h, c, radius)
    if rc:
        raise ValueError(f"oracle_blur_int rc={rc}")
    return out


def blur_f32(img: np.ndarray, radius: int = 1) -> np.ndarray:
    """Float-form oracle (statement-by-statement restatement of the reference kernel)."""
    h, w, c = _chk(img)
    out = np.empty_like(img)
    (lib().oracle_blur3_f32 if radius == 1 else lib().oracle_blur5_f32)(_p(img), _p(out), w, h, c)
    return out


def blur_batch(stream: np.ndarray, radius: int = 1) -> np.ndarray:
    """NxHxWxC stream of independent images (heterogeneous_blur.c:431-442,485-486)."""
    assert stream.ndim == 4 and stream.dtype == np.uint8 and stream.flags.c_contiguous
    n, h, w, c = stream.shape
    out = np.empty_like(stream)
    rc = lib().oracle_blur_batch(_p(stream), _p(out), w, h, c, radius, n)
    if rc:
        raise ValueError(f"oracle_blur_batch rc={rc}")
    return out


def a1_partition(mode: int, batch_count: int, gpu_ratio: float) -> tuple[int, int]:
    nc, ng = C.c_int(), C.c_int()
    lib().oracle_a1_partition(mode, batch_count, gpu_ratio, C.byref(nc), C.byref(ng))
    return nc.value, ng.value


def a2_geometry(height: int, gpu_ratio: float, halo: int = 1) -> dict:
    g = _Geom()
    lib().oracle_a2_geometry(height, gpu_ratio, halo, C.byref(g))
    return {n: getattr(g, n) for n, _ in _Geom._fields_}


def a2_split_blur(img: np.ndarray, split_row: int, radius: int = 1) -> np.ndarray:
    h, w, c = _chk(img)
    out = np.empty_like(img)
    rc = lib().oracle_a2_split_blur(_p(img), _p(out), w, h, c, split_row, radius)
    if rc:
        raise ValueError(f"oracle_a2_split_blur rc={rc}")
    return out


def splitk_blur(img: np.ndarray, k: int, radius: int = 1) -> np.ndarray:
    h, w, c = _chk(img)
    out = np.empty_like(img)
    rc = lib().oracle_splitk_blur(_p(img), _p(out), w, h, c, k, radius)
    if rc:
        raise ValueError(f"oracle_splitk_blur rc={rc}")
    return out


def planar_to_interleaved(planar: np.ndarray) -> np.ndarray:
    """planar: (C, H, W) uint8 (CImg storage) -> (H, W, C) interleaved, heterogeneous_blur.c:125-134."""
    assert planar.dtype == np.uint8 and planar.ndim == 3 and planar.flags.c_contiguous
    c, h, w = planar.shape
    out = np.empty((h, w, c), np.uint8)
    lib().oracle_planar_to_interleaved(_p(planar), _p(out), w, h, c)
    return out


def interleaved_to_planar(img: np.ndarray) -> np.ndarray:
    """(H, W, C) interleaved -> (C, H, W) planar, split_image_blur.c:40-56."""
    h, w, c = _chk(img)
    out = np.empty((c, h, w), np.uint8)
    lib().oracle_interleaved_to_planar(_p(img), _p(out), w, h, c)
    return out


def lcg_image(h: int, w: int, c: int, seed: int = LCG_SEED) -> np.ndarray:
    img = np.empty((h, w, c), np.uint8)
    lib().oracle_lcg_fill(_p(img), img.size, seed & 0xFFFFFFFF)
    return img


def lcg_stream(n: int, h: int, w: int, c: int, first_index: int = 0) -> np.ndarray:
    """Image i of the synthetic stream is LCG-filled with seed 0x9E3779B9 ^ i."""
    s = np.empty((n, h, w, c), np.uint8)
    for i in range(n):
        lib().oracle_lcg_fill(_p(s[i]), s[i].size, (LCG_SEED ^ (first_index + i)) & 0xFFFFFFFF)
    return s


def fnv1a64(a: np.ndarray) -> int:
    a = np.ascontiguousarray(a)
    return int(lib().oracle_fnv1a64(_p(a), a.size))


# --------------------------------------------------------------------------- reference kernel
def ref_blur(img: np.ndarray) -> np.ndarray:
    """Output of the UNMODIFIED reference kernel under the NDRange harness."""
    h, w, c = _chk(img)
    out = np.empty_like(img)
    ref().ref_gaussian_blur(_p(img), _p(out), w, h, c)
    return out


def ref_split_blur(img: np.ndarray, split_row: int) -> np.ndarray:
    h, w, c = _chk(img)
    out = np.empty_like(img)
    tmp = np.empty((max(split_row + 1, h - split_row + 1), w, c), np.uint8)
    ref().ref_split_image_blur(_p(img), _p(out), _p(tmp), w, h, c, split_row)
    return out


# --------------------------------------------------------------------------- numpy restatement
_TAPS = {1: np.array([1, 2, 1], np.uint32), 2: np.array([1, 4, 6, 4, 1], np.uint32)}


def np_blur(img: np.ndarray, radius: int = 1) -> np.ndarray:
    """Independent vectorised restatement: edge-replicated pad (= clamp-to-edge,
    gaussian_kernel.cl:56-57), full 2-D tap sum, one truncating shift (:70)."""
    h, w, c = _chk(img)
    t = _TAPS[radius]
    pad = np.pad(img.astype(np.uint32), ((radius, radius), (radius, radius), (0, 0)), mode="edge")
    acc = np.zeros((h, w, c), np.uint32)
    for ky in range(2 * radius + 1):
        for kx in range(2 * radius + 1):
            acc += t[ky] * t[kx] * pad[ky:ky + h, kx:kx + w, :]
    return (acc >> (4 if radius == 1 else 8)).astype(np.uint8)
