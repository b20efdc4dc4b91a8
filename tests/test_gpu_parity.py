"""Parity of the HIP path against the oracle, through the C ABI, on a real MI355X (-m gpu).

Bit-exact is the bar (uint8 path).  torch is used for device memory only; every blur goes
through libmi_blur.so, and the tests fail if the library cannot run on the GPU (no fallback).
"""
import ctypes as C
import os
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda(L):
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    assert L.mi_blur_device_count() >= 1, "libmi_blur.so sees no HIP device"
    torch.cuda.set_device(0)
    return torch


def gpu_blur(pkg, L, torch, host, radius, variant=0, y0=None, y1=None, opts=None):
    """host: N x H x W x C uint8 -> output rows [y0,y1) of every band, via mi_blur_enqueue_ex."""
    n, h, w, c = host.shape
    y0 = 0 if y0 is None else y0
    y1 = h if y1 is None else y1
    for k, v in (opts or {}).items():
        pkg.check(L.mi_blur_set_option(k.encode(), v), f"set_option {k}")
    d_in = torch.from_numpy(host).cuda()
    d_out = torch.full((n, y1 - y0, w, c), 0xA5, dtype=torch.uint8, device="cuda")
    guard = torch.full((4096,), 0x5A, dtype=torch.uint8, device="cuda")   # allocated right after: catches overruns loosely
    rc = L.mi_blur_enqueue_ex(d_in.data_ptr(), d_out.data_ptr(), w, h, c, radius, n, y0, y1, variant,
                              torch.cuda.current_stream().cuda_stream)
    pkg.check(rc, "mi_blur_enqueue_ex")
    torch.cuda.synchronize()
    assert bool((guard == 0x5A).all())
    assert bool((d_in.cpu() == torch.from_numpy(host)).all()), "input buffer modified"
    return d_out.cpu().numpy()


def reset_opts(L):
    L.mi_blur_set_option(b"prefer_stream", 0)
    L.mi_blur_set_option(b"row_shuffle", 0)
    L.mi_blur_set_option(b"stream_band_rows", 0)
    L.mi_blur_set_option(b"stage_dma", 1)
    L.mi_blur_set_option(b"zero_copy", 1)
    L.mi_blur_set_option(b"ragged_tiled", 1)
    L.mi_blur_set_option(b"rows_per_thread", 0)
    L.mi_blur_set_option(b"xcd_remap", 1)
    L.mi_blur_set_option(b"experiment", 0)
    L.mi_blur_set_option(b"stream_updown", 1)
    L.mi_blur_set_option(b"xcd_run", 0)
    L.mi_blur_set_option(b"prefer_direct", 1)
    L.mi_blur_set_option(b"direct_bh", 8)
    L.mi_blur_set_option(b"zero_copy_server", 1)
    L.mi_blur_set_option(b"staged_server", 1)
    L.mi_blur_set_option(b"zero_copy_server_min_kb", 1280)
    L.mi_blur_set_option(b"zero_copy_workers", 48)
    L.mi_blur_set_option(b"zero_copy_idle_us", 300)
    L.mi_blur_set_option(b"zero_copy_budget", 256)
    L.mi_blur_set_option(b"zero_copy_debug_base", 0)
    L.mi_blur_set_option(b"fused_tail", 30)
    L.mi_blur_set_option(b"fused_tail_blocks", 25)
    L.mi_blur_set_option(b"resident_place_trials", 4)


@pytest.fixture
def server_for_small_batches(L):
    """These tests drive the batch server with small frames: switch off the rule that sends submits below 1.25 MiB to per-batch launches."""
    L.mi_blur_set_option(b"zero_copy_server_min_kb", 0)
    yield
    L.mi_blur_set_option(b"zero_copy_server_min_kb", 1280)


def want_batch(O, host, radius):
    return np.stack([O.blur(np.ascontiguousarray(host[i]), radius) for i in range(host.shape[0])])


# shapes: (H, W, C).  pitch % 16 == 0 -> tiled kernel eligible; the rest exercise the generic kernel.
TILED_SHAPES = [(16, 16, 3), (1, 16, 1), (2, 16, 3), (3, 32, 2), (5, 64, 4), (9, 48, 3), (33, 80, 3), (64, 80, 3),
                (240, 320, 3), (256, 256, 3), (47, 1360, 3), (40, 4096, 1), (31, 1040, 4), (100, 16, 4),
                (131, 112, 1), (17, 2064, 2), (2, 262144, 1), (1, 65536, 4)]
GENERIC_SHAPES = [(1, 1, 3), (3, 5, 3), (33, 17, 3), (33, 17, 1), (31, 29, 4), (2, 2, 3), (64, 1, 3), (1, 64, 3),
                  (7, 9, 5), (50, 37, 2)]


def adversarial(O, h, w, c, n, seed):
    rng = np.random.default_rng(seed)
    imgs = [O.lcg_stream(n, h, w, c), rng.integers(0, 256, (n, h, w, c), dtype=np.uint8),
            np.full((n, h, w, c), 255, np.uint8)]
    imp = np.zeros((n, h, w, c), np.uint8)           # impulses at corners / edges
    imp[:, 0, 0, :] = 255; imp[:, -1, -1, :] = 255; imp[:, 0, -1, 0] = 255; imp[:, -1, 0, -1] = 255
    imp[:, h // 2, 0, :] = 255; imp[:, h // 2, -1, :] = 255; imp[:, 0, w // 2, :] = 255; imp[:, -1, w // 2, :] = 255
    imgs.append(imp)
    return imgs


@pytest.mark.parametrize("h,w,c", TILED_SHAPES)
@pytest.mark.parametrize("radius", [1, 2])
def test_tiled_kernel_bit_exact(pkg, L, O, torch_cuda, h, w, c, radius):
    n = 3
    try:
        for host in adversarial(O, h, w, c, n, h * 7 + w):
            want = want_batch(O, host, radius)
            for opts in ({"stage_dma": 1, "rows_per_thread": 8, "xcd_remap": 1, "row_shuffle": 0},
                         {"stage_dma": 0, "rows_per_thread": 16, "xcd_remap": 0, "row_shuffle": 0},
                         {"stage_dma": 1, "rows_per_thread": 0, "xcd_remap": 1, "row_shuffle": 1}):
                got = gpu_blur(pkg, L, torch_cuda, host, radius, pkg.VARIANT_TILED, opts=opts)
                assert np.array_equal(got, want), f"{(got != want).sum()} bytes differ, opts={opts}"
    finally:
        reset_opts(L)


@pytest.mark.parametrize("opts", [{"stage_dma": 1, "rows_per_thread": 16}, {"stage_dma": 0, "rows_per_thread": 8},
                                  {"stage_dma": 1, "rows_per_thread": 4}, {"stage_dma": 0, "rows_per_thread": 4, "xcd_remap": 0}])
@pytest.mark.parametrize("radius", [1, 2])
def test_tiled_kernel_option_cross(pkg, L, O, torch_cuda, opts, radius):
    try:
        for (h, w, c) in [(256, 256, 3), (75, 1360, 3), (40, 64, 4)]:
            host = O.lcg_stream(2, h, w, c)
            got = gpu_blur(pkg, L, torch_cuda, host, radius, pkg.VARIANT_TILED, opts=opts)
            assert np.array_equal(got, want_batch(O, host, radius))
    finally:
        reset_opts(L)


@pytest.mark.parametrize("h,w,c", TILED_SHAPES)
@pytest.mark.parametrize("radius", [1, 2])
def test_stream_kernel_bit_exact(pkg, L, O, torch_cuda, h, w, c, radius):
    """Barrier-free streaming variant (register sliding window + DPP row pass)."""
    n = 3
    try:
        for host in adversarial(O, h, w, c, n, h * 11 + w):
            want = want_batch(O, host, radius)
            for opts in ({"stream_band_rows": 0, "xcd_remap": 1}, {"stream_band_rows": 5, "xcd_remap": 0, "stream_updown": 0},
                         {"stream_band_rows": 64, "xcd_remap": 1, "stream_updown": 1}):
                got = gpu_blur(pkg, L, torch_cuda, host, radius, pkg.VARIANT_STREAM, opts=opts)
                assert np.array_equal(got, want), f"{(got != want).sum()} bytes differ, opts={opts}"
    finally:
        reset_opts(L)


@pytest.mark.parametrize("h,w,c", TILED_SHAPES)
@pytest.mark.parametrize("radius", [1, 2])
def test_direct_kernel_bit_exact(pkg, L, O, torch_cuda, h, w, c, radius):
    """Direct variant (no LDS: rows straight into registers, x-neighbours by DPP wave shifts, 62 computing lanes per
    wave): adversarial images, with and without the XCD map, output row ranges inside a band."""
    n = 3
    try:
        for host in adversarial(O, h, w, c, n, h * 17 + w):
            want = want_batch(O, host, radius)
            for opts in ({"xcd_remap": 1}, {"xcd_remap": 0}):
                got = gpu_blur(pkg, L, torch_cuda, host, radius, pkg.VARIANT_DIRECT, opts=opts)
                assert np.array_equal(got, want), f"{(got != want).sum()} bytes differ, opts={opts}"
        if h >= 4:
            host = O.lcg_stream(n, h, w, c)
            y0, y1 = 1, h - 1
            got = gpu_blur(pkg, L, torch_cuda, host, radius, pkg.VARIANT_DIRECT, y0=y0, y1=y1)
            assert np.array_equal(got, want_batch(O, host, radius)[:, y0:y1])
    finally:
        reset_opts(L)


def test_direct_kernel_sizes_and_dispatch(pkg, L, O, torch_cuda):
    """(1) Launch sizes around the wave / workgroup granularity of the flattened (image, band, chunk) space: 1..130 images
    of a shape whose row is shorter than a wave, other band heights (C = 3 instantiations).  (2) AUTO takes the direct
    kernel for 5x5 and for small 3x3 launches and the tiled kernel for big 3x3 ones — all three say the same bytes."""
    try:
        for n in (1, 2, 7, 61, 62, 63, 130):
            host = O.lcg_stream(n, 9, 80, 3)
            for radius in (1, 2):
                want = want_batch(O, host, radius)
                for bh in (8, 4, 12, 16):
                    got = gpu_blur(pkg, L, torch_cuda, host, radius, pkg.VARIANT_DIRECT, opts={"direct_bh": bh})
                    assert np.array_equal(got, want), (n, radius, bh)
        host = O.lcg_stream(6, 256, 256, 3)
        for radius in (1, 2):
            want = want_batch(O, host, radius)
            for pd in (0, 1, 2):
                assert np.array_equal(gpu_blur(pkg, L, torch_cuda, host, radius, pkg.VARIANT_AUTO, opts={"prefer_direct": pd}), want), (radius, pd)
                assert L.mi_blur_last_kernel() == (b"blur_tiled_kernel" if pd == 0 else b"blur_direct_kernel")
        assert np.array_equal(gpu_blur(pkg, L, torch_cuda, O.lcg_stream(1, 33, 17, 3), 1, pkg.VARIANT_AUTO), want_batch(O, O.lcg_stream(1, 33, 17, 3), 1))
        assert L.mi_blur_last_kernel() == b"blur_tiled_kernel"          # ragged rows: the tiled kernel's ragged form
    finally:
        reset_opts(L)


@pytest.mark.parametrize("radius", [1, 2])
def test_tiled_experiment_variants_bit_exact(pkg, L, O, torch_cuda, radius):
    """Both row-pass forms of the tiled kernel (field pairs straight from the raw window by v_perm / split-then-shift) and
    the blockIdx -> tile maps (per-launch choice, contiguous eighths, runs of r tiles dealt to the XCDs): same bytes as the
    oracle on C = 3 shapes incl. row edges, several strips, short tiles, tile counts that are not multiples of 8 x run."""
    try:
        for (h, w, c) in [(16, 16, 3), (9, 48, 3), (33, 80, 3), (240, 320, 3), (256, 256, 3), (47, 1360, 3), (70, 1920, 3), (3, 21856, 3)]:
            for host in adversarial(O, h, w, c, 2, h * 5 + w):
                want = want_batch(O, host, radius)
                for x in (0, 1):
                    for rpt, run in ((4, 0), (8, 1), (8, 3), (4, 16)):
                        got = gpu_blur(pkg, L, torch_cuda, host, radius, pkg.VARIANT_TILED,
                                       opts={"experiment": x, "rows_per_thread": rpt, "xcd_run": run})
                        assert np.array_equal(got, want), f"{(got != want).sum()} bytes differ, experiment={x} rpt={rpt} run={run} {(h, w, c)}"
    finally:
        reset_opts(L)


@pytest.mark.parametrize("radius", [1, 2])
def test_stream_kernel_updown_bit_exact(pkg, L, O, torch_cuda, radius):
    """Streaming variant with odd bands marching upwards (stream_updown): bit-exact for band heights that do and do not
    divide the image, one band, short last bands, bands (Approach-2 clamp at the band's own edges)."""
    try:
        for (h, w, c) in [(16, 16, 3), (33, 80, 3), (240, 320, 3), (256, 256, 3), (47, 1360, 3), (131, 112, 1), (100, 16, 4), (1080, 1920, 3)]:
            host = O.lcg_stream(2, h, w, c) if h < 1000 else O.lcg_stream(1, h, w, c)
            want = want_batch(O, host, radius)
            for bh in (0, 5, 7, 64, 100):
                got = gpu_blur(pkg, L, torch_cuda, host, radius, pkg.VARIANT_STREAM, opts={"stream_updown": 1, "stream_band_rows": bh})
                assert np.array_equal(got, want), f"{(got != want).sum()} bytes differ, band rows {bh} {(h, w, c)}"
        img = O.lcg_image(240, 320, 3)
        whole = O.blur(img, radius)
        parts = []
        for g in range(3):
            b = pkg.band_of(240, radius, g, 3)
            band = np.ascontiguousarray(img[b["row_begin"] - b["halo_top"]: b["row_end"] + b["halo_bottom"]])[None]
            parts.append(gpu_blur(pkg, L, torch_cuda, band, radius, pkg.VARIANT_STREAM, y0=b["halo_top"],
                                  y1=b["halo_top"] + b["row_end"] - b["row_begin"], opts={"stream_updown": 1, "stream_band_rows": 13})[0])
        assert np.array_equal(np.concatenate(parts), whole)
    finally:
        reset_opts(L)


def test_stream_kernel_bands_and_big(pkg, L, O, torch_cuda):
    try:
        img = O.lcg_image(240, 320, 3)
        for radius in (1, 2):
            whole = O.blur(img, radius)
            for G in (2, 3):
                parts = []
                for g in range(G):
                    b = pkg.band_of(240, radius, g, G)
                    band = np.ascontiguousarray(img[b["row_begin"] - b["halo_top"]: b["row_end"] + b["halo_bottom"]])[None]
                    parts.append(gpu_blur(pkg, L, torch_cuda, band, radius, pkg.VARIANT_STREAM, y0=b["halo_top"],
                                          y1=b["halo_top"] + b["row_end"] - b["row_begin"])[0])
                assert np.array_equal(np.concatenate(parts), whole)
        for (h, w, c, n) in [(256, 256, 3, 70), (1080, 1920, 3, 2), (75, 4096, 4, 2)]:
            host = O.lcg_stream(n, h, w, c)
            for radius in (1, 2):
                a = gpu_blur(pkg, L, torch_cuda, host, radius, pkg.VARIANT_STREAM)
                b = gpu_blur(pkg, L, torch_cuda, host, radius, pkg.VARIANT_TILED)
                assert np.array_equal(a, b)
                assert np.array_equal(a[0], O.blur(np.ascontiguousarray(host[0]), radius))
    finally:
        reset_opts(L)


@pytest.mark.parametrize("h,w,c", GENERIC_SHAPES + TILED_SHAPES[:8])
@pytest.mark.parametrize("radius", [1, 2])
def test_generic_kernel_bit_exact(pkg, L, O, torch_cuda, h, w, c, radius):
    for host in adversarial(O, h, w, c, 2, h * 3 + w):
        got = gpu_blur(pkg, L, torch_cuda, host, radius, pkg.VARIANT_GENERIC)
        assert np.array_equal(got, want_batch(O, host, radius))


def test_auto_dispatch_and_ineligible_tiled(pkg, L, O, torch_cuda):
    host = O.lcg_stream(2, 33, 17, 3)                 # pitch 51: not a multiple of 16
    assert np.array_equal(gpu_blur(pkg, L, torch_cuda, host, 1, pkg.VARIANT_AUTO), want_batch(O, host, 1))
    d = torch_cuda.zeros(2 * 33 * 17 * 3, dtype=torch_cuda.uint8, device="cuda")
    o = torch_cuda.zeros_like(d)
    # pitch 51 takes the ragged form of the tiled kernel; rows shorter than one chunk (pitch 15) cannot; 5 to 8 channels can
    # for the 3x3 (the +-C taps stay inside the staged window) but not for the 5x5, and 9 channels never
    assert L.mi_blur_enqueue_ex(d.data_ptr(), o.data_ptr(), 17, 33, 3, 1, 2, 0, 33, pkg.VARIANT_TILED, None) == pkg.OK
    assert L.mi_blur_enqueue_ex(d.data_ptr(), o.data_ptr(), 5, 33, 3, 1, 2, 0, 33, pkg.VARIANT_TILED, None) == pkg.ERR_INVALID
    assert L.mi_blur_enqueue_ex(d.data_ptr(), o.data_ptr(), 17, 19, 5, 1, 2, 0, 19, pkg.VARIANT_TILED, None) == pkg.OK
    assert L.mi_blur_enqueue_ex(d.data_ptr(), o.data_ptr(), 17, 19, 5, 2, 2, 0, 19, pkg.VARIANT_TILED, None) == pkg.ERR_INVALID
    assert L.mi_blur_enqueue_ex(d.data_ptr(), o.data_ptr(), 17, 11, 9, 1, 2, 0, 11, pkg.VARIANT_TILED, None) == pkg.ERR_INVALID
    assert L.mi_blur_enqueue_ex(d.data_ptr(), o.data_ptr(), 16, 19, 5, 1, 2, 0, 19, pkg.VARIANT_DIRECT, None) == pkg.ERR_INVALID
    assert L.mi_blur_enqueue_ex(d.data_ptr(), o.data_ptr(), 17, 33, 3, 1, 2, 0, 33, pkg.VARIANT_STREAM, None) == pkg.ERR_INVALID
    assert L.mi_blur_enqueue_ex(d.data_ptr(), d.data_ptr(), 17, 33, 3, 1, 2, 0, 33, 0, None) == pkg.ERR_INVALID
    assert L.mi_blur_enqueue_ex(d.data_ptr(), o.data_ptr(), 17, 33, 3, 3, 2, 0, 33, 0, None) == pkg.ERR_INVALID
    assert L.mi_blur_enqueue_ex(d.data_ptr(), o.data_ptr(), 17, 33, 3, 1, 2, 5, 5, 0, None) == pkg.ERR_INVALID
    assert L.mi_blur_enqueue_ex(d.data_ptr(), o.data_ptr(), 17, 33, 3, 1, 0, 0, 33, 0, None) == pkg.OK     # empty batch
    # misaligned device pointers take the ragged form under AUTO
    host = O.lcg_stream(1, 16, 16, 3)
    buf = torch_cuda.zeros(host.size + 64, dtype=torch_cuda.uint8, device="cuda")
    out = torch_cuda.zeros(host.size + 64, dtype=torch_cuda.uint8, device="cuda")
    buf[3:3 + host.size] = torch_cuda.from_numpy(host.reshape(-1)).cuda()
    pkg.check(L.mi_blur_enqueue(buf.data_ptr() + 3, out.data_ptr() + 5, 16, 16, 3, 1, 1, None))
    torch_cuda.cuda.synchronize()
    assert np.array_equal(out[5:5 + host.size].cpu().numpy().reshape(host.shape), want_batch(O, host, 1))


WIDE_SHAPES = [(33, 16, 5), (64, 80, 6), (40, 48, 7), (96, 64, 8), (17, 33, 5), (50, 37, 6), (9, 129, 7), (31, 29, 8), (2, 4, 8),
               (240, 320, 6), (3, 1000, 5)]


@pytest.mark.parametrize("h,w,c", WIDE_SHAPES)
def test_five_to_eight_channel_frames_take_the_tiled_kernel(pkg, L, O, torch_cuda, h, w, c):
    """The reference kernel is generic in `channels` (gaussian_kernel.cl:44).  For the 3x3, frames of 5 to 8 channels run
    the LDS-DMA tiled kernel (aligned and ragged rows) — the +-C byte taps still fall inside the 8 + 16 + 8-byte window and
    the x-clamp selectors reach over two dwords; the 5x5 of such frames (+-2C) stays on the generic kernel.  Bit-exact vs
    the oracle on adversarial images, batches, bands, odd pointer offsets, every rows-per-thread setting."""
    n = 3
    torch = torch_cuda
    try:
        for host in adversarial(O, h, w, c, n, h * 7 + w + c):
            want = want_batch(O, host, 1)
            for opts in ({"rows_per_thread": 0}, {"rows_per_thread": 4}, {"rows_per_thread": 8, "xcd_remap": 0}, {"rows_per_thread": 16}):
                for variant in (pkg.VARIANT_AUTO, pkg.VARIANT_TILED):
                    got = gpu_blur(pkg, L, torch, host, 1, variant, opts=opts)
                    assert np.array_equal(got, want), f"{(got != want).sum()} bytes differ, opts={opts}"
                    assert L.mi_blur_last_kernel() == b"blur_tiled_kernel"
        host = O.lcg_stream(n, h, w, c)
        assert np.array_equal(gpu_blur(pkg, L, torch, host, 2, pkg.VARIANT_AUTO), want_batch(O, host, 2))
        assert L.mi_blur_last_kernel() == b"blur_generic_kernel"
        want = want_batch(O, host, 1)
        if h >= 3:
            assert np.array_equal(gpu_blur(pkg, L, torch, host, 1, pkg.VARIANT_AUTO, y0=1, y1=h - 1), want[:, 1:h - 1])
        buf = torch.full((host.size + 64,), 0xA5, dtype=torch.uint8, device="cuda")
        out = torch.full((host.size + 64,), 0x5A, dtype=torch.uint8, device="cuda")
        buf[1:1 + host.size] = torch.from_numpy(host.reshape(-1)).cuda()
        pkg.check(L.mi_blur_enqueue(buf.data_ptr() + 1, out.data_ptr() + 7, w, h, c, 1, n, None))
        torch.cuda.synchronize()
        res = out.cpu().numpy()
        assert np.array_equal(res[7:7 + host.size].reshape(host.shape), want)
        assert (res[:7] == 0x5A).all() and (res[7 + host.size:] == 0x5A).all()
    finally:
        reset_opts(L)


RAGGED_SHAPES = [(33, 17, 3), (5, 16, 1), (1, 17, 1), (2, 6, 3), (31, 29, 4), (40, 250, 3), (64, 1366, 3), (19, 1000, 3),
                 (9, 9, 2), (300, 33, 1), (7, 2731, 3), (50, 37, 2), (3, 21, 1), (12, 4099, 1), (65, 170, 3)]


@pytest.mark.parametrize("h,w,c", RAGGED_SHAPES)
@pytest.mark.parametrize("radius", [1, 2])
def test_ragged_tiled_bit_exact(pkg, L, O, torch_cuda, h, w, c, radius):
    """Rows that are not a multiple of 16 bytes: the ragged form of the tiled kernel == oracle == generic kernel, for
    batches, for every rows-per-thread setting, at odd pointer offsets, as bands; gpu_blur's guard bytes catch a store
    that spills over the end (the partial last chunk must write only the bytes that exist)."""
    n = 3
    try:
        for host in adversarial(O, h, w, c, n, h * 13 + w):
            want = want_batch(O, host, radius)
            for opts in ({"rows_per_thread": 0, "stage_dma": 1}, {"rows_per_thread": 4, "stage_dma": 0},
                         {"rows_per_thread": 8, "xcd_remap": 0, "stage_dma": 1}, {"rows_per_thread": 16, "stage_dma": 0}):
                got = gpu_blur(pkg, L, torch_cuda, host, radius, pkg.VARIANT_TILED, opts=opts)
                assert np.array_equal(got, want), f"{(got != want).sum()} bytes differ, opts={opts}"
        host = O.lcg_stream(n, h, w, c)
        want = want_batch(O, host, radius)
        assert np.array_equal(gpu_blur(pkg, L, torch_cuda, host, radius, pkg.VARIANT_AUTO, opts={"ragged_tiled": 0}), want)   # generic
        pkg.check(L.mi_blur_set_option(b"ragged_tiled", 1))
        if h >= 3:
            y0, y1 = 1, h - 1
            got = gpu_blur(pkg, L, torch_cuda, host, radius, pkg.VARIANT_TILED, y0=y0, y1=y1)
            assert np.array_equal(got, want[:, y0:y1])
        # odd pointer offsets (input +1, output +7)
        torch = torch_cuda
        buf = torch.full((host.size + 64,), 0xA5, dtype=torch.uint8, device="cuda")
        out = torch.full((host.size + 64,), 0x5A, dtype=torch.uint8, device="cuda")
        buf[1:1 + host.size] = torch.from_numpy(host.reshape(-1)).cuda()
        pkg.check(L.mi_blur_enqueue_ex(buf.data_ptr() + 1, out.data_ptr() + 7, w, h, c, radius, n, 0, h, pkg.VARIANT_TILED, None))
        torch.cuda.synchronize()
        o = out.cpu().numpy()
        assert np.array_equal(o[7:7 + host.size].reshape(host.shape), want)
        assert (o[:7] == 0x5A).all() and (o[7 + host.size:] == 0x5A).all(), "wrote outside the output"
    finally:
        reset_opts(L)
        L.mi_blur_set_option(b"ragged_tiled", 1)


def test_golden_hashes_on_gpu(pkg, L, O, torch_cuda, golden):
    """The committed reference-kernel known answers, reproduced by the HIP path."""
    for e in golden["k3"]:                                   # every committed shape, 8192x8192x3 (201 MB) included
        host = O.lcg_image(e["h"], e["w"], e["c"])[None]
        assert f"{O.fnv1a64(host):016x}" == e["in_fnv"], e
        got = gpu_blur(pkg, L, torch_cuda, host, 1)
        assert f"{L.mi_blur_fnv1a64(got.ctypes.data, got.size):016x}" == e["out_fnv"], e
    for lit in golden["literals"]:
        host = np.array(lit["in"], np.uint8).reshape(1, lit["h"], lit["w"], lit["c"])
        assert gpu_blur(pkg, L, torch_cuda, host, 1).reshape(-1).tolist() == lit["out"], lit["name"]
    for e in golden["k5_unpinned"]:
        host = O.lcg_image(e["h"], e["w"], e["c"])[None]
        assert f"{O.fnv1a64(gpu_blur(pkg, L, torch_cuda, host, 2)):016x}" == e["out_fnv"]


def test_band_semantics_and_split_equals_whole(pkg, L, O, torch_cuda):
    """Approach 2 (split_image_blur.c:401,414,511-541): band kernels with the band height as
    `height`, halo rows dropped; every split row and K-way split reproduce the whole image."""
    for (h, w, c) in [(240, 320, 3), (64, 48, 4), (37, 17, 3)]:
        img = O.lcg_image(h, w, c)
        for radius in (1, 2):
            whole = O.blur(img, radius)
            for split in sorted({radius, 39 % h, h // 2, h - radius}):
                if split < radius or split > h - radius:
                    continue
                top_in = np.ascontiguousarray(img[:split + radius])[None]
                bot_in = np.ascontiguousarray(img[split - radius:])[None]
                top = gpu_blur(pkg, L, torch_cuda, top_in, radius, y0=0, y1=split)
                bot = gpu_blur(pkg, L, torch_cuda, bot_in, radius, y0=radius, y1=h - split + radius)
                assert np.array_equal(np.concatenate([top[0], bot[0]]), whole), (h, w, c, radius, split)
            for G in (2, 3, 8):
                parts = []
                for g in range(G):
                    b = pkg.band_of(h, radius, g, G)
                    band = np.ascontiguousarray(img[b["row_begin"] - b["halo_top"]: b["row_end"] + b["halo_bottom"]])[None]
                    parts.append(gpu_blur(pkg, L, torch_cuda, band, radius, y0=b["halo_top"],
                                          y1=b["halo_top"] + b["row_end"] - b["row_begin"])[0])
                assert np.array_equal(np.concatenate(parts), whole)


def test_context_submit_pageable_and_pinned(pkg, L, O, torch_cuda):
    h, w, c, n = 240, 320, 3, 35
    host = O.lcg_stream(n, h, w, c)
    want = O.blur_batch(host, 1)
    with pkg.Context(0, w, h, c, 1, max_batch=n, n_slots=2) as ctx:
        # pageable caller memory (the reference's malloc'd batch buffers) is gathered into the slot's pinned staging; from there
        # the batch server takes it in place (default), or — "staged_server" 0 — a DMA copy each way around a launch
        for staged_server in (1, 0):
            pkg.check(L.mi_blur_set_option(b"staged_server", staged_server))
            try:
                out = np.zeros_like(host)
                before = L.mi_blur_zero_copy_launches(ctx.h)
                ctx.reset_timing()
                ctx.submit(host.ctypes.data, out.ctypes.data, n)
                ctx.submit(host[:10].ctypes.data, out[:10].ctypes.data, 10)          # second slot, overlapping range rewritten
                tm = ctx.sync()
                assert np.array_equal(out, want), staged_server
                assert tm["images"] == n + 10 and tm["launches"] == 2 and tm["bytes_h2d"] == (n + 10) * h * w * c
                if staged_server:
                    assert L.mi_blur_last_kernel() == b"blur_server_kernel" and L.mi_blur_zero_copy_launches(ctx.h) - before == 2
                    assert tm["kernel_ms"] > 0 and tm["h2d_ms"] == 0 and tm["d2h_ms"] == 0      # the transfer time IS the kernel bucket
                else:
                    assert L.mi_blur_zero_copy_launches(ctx.h) - before == 0
                    assert tm["kernel_ms"] > 0 and tm["h2d_ms"] > 0 and tm["d2h_ms"] > 0
            finally:
                pkg.check(L.mi_blur_set_option(b"staged_server", 1))
        # pinned caller memory: the kernel works on it in place (zero-copy, default) or DMA goes straight from/to it
        nbytes = host.nbytes
        p_in, p_out = L.mi_blur_host_alloc(nbytes), L.mi_blur_host_alloc(nbytes)
        assert p_in and p_out
        C.memmove(p_in, host.ctypes.data, nbytes)
        try:
            for zc in (1, 0):
                pkg.check(L.mi_blur_set_option(b"zero_copy", zc))
                C.memset(p_out, 0, nbytes)
                before = L.mi_blur_zero_copy_launches(ctx.h)
                ctx.reset_timing()
                ctx.submit(p_in, p_out, n)
                ctx.submit(p_in, p_out, 10)                                  # chained behind the first
                tm = ctx.sync()
                got = np.ctypeslib.as_array((C.c_uint8 * nbytes).from_address(p_out)).reshape(host.shape).copy()
                assert np.array_equal(got, want), zc
                assert L.mi_blur_zero_copy_launches(ctx.h) - before == (2 if zc else 0)
                assert tm["bytes_h2d"] == (n + 10) * h * w * c and tm["kernel_ms"] > 0
                if zc:
                    assert tm["h2d_ms"] < 0.25 * tm["kernel_ms"] and tm["d2h_ms"] < 0.25 * tm["kernel_ms"]
                # input untouched by the in-place read
                assert np.array_equal(np.ctypeslib.as_array((C.c_uint8 * nbytes).from_address(p_in)).reshape(host.shape), host)
        finally:
            pkg.check(L.mi_blur_set_option(b"zero_copy", 1))
        L.mi_blur_host_free(p_in); L.mi_blur_host_free(p_out)
        # the caller's own (malloc'd) buffers registered in place: the same in-place path as mi_blur_host_alloc memory
        own_in, own_out = host.copy(), np.zeros_like(host)
        pkg.check(L.mi_blur_host_register(own_in.ctypes.data, nbytes)); pkg.check(L.mi_blur_host_register(own_out.ctypes.data, nbytes))
        before = L.mi_blur_zero_copy_launches(ctx.h)
        ctx.submit(own_in.ctypes.data, own_out.ctypes.data, n)
        ctx.sync()
        assert L.mi_blur_zero_copy_launches(ctx.h) - before == 1 and np.array_equal(own_out, want)
        pkg.check(L.mi_blur_host_unregister(own_in.ctypes.data)); pkg.check(L.mi_blur_host_unregister(own_out.ctypes.data))
        own_out[:] = 0
        ctx.submit(own_in.ctypes.data, own_out.ctypes.data, n)                  # unregistered again: through the staging, same bytes
        ctx.sync()
        assert L.mi_blur_zero_copy_launches(ctx.h) - before == 2 and np.array_equal(own_out, want)
        # Approach-2 bands through the context
        one = np.ascontiguousarray(host[0]); split = 39
        top, bot = np.zeros((split, w, c), np.uint8), np.zeros((h - split, w, c), np.uint8)
        ctx.submit_band(one.ctypes.data, top.ctypes.data, split + 1, 0, 1)
        ctx.submit_band(one[split - 1:].ctypes.data, bot.ctypes.data, h - split + 1, 1, 0)
        ctx.sync()
        assert np.array_equal(np.concatenate([top, bot]), want[0])


@pytest.mark.parametrize("radius", [1, 2])
def test_zero_copy_capped_grid_and_streams(pkg, L, O, torch_cuda, radius):
    """Zero-copy submits run the aligned tiled kernel with a CAPPED grid (workgroups loop over the tiles) on up to
    "zero_copy_streams" streams: every cap (1 workgroup, fewer than / more than the tiles, the default) and stream count
    gives the oracle's bytes; several submits in flight on rotating buffers; strided bands take the same path."""
    h, w, c, n = 96, 320, 3, 9
    host = O.lcg_stream(4 * n, h, w, c)
    want = O.blur_batch(host, radius)
    nbytes = n * h * w * c
    bufs = [(L.mi_blur_host_alloc(nbytes), L.mi_blur_host_alloc(nbytes)) for _ in range(4)]
    try:
        pkg.check(L.mi_blur_set_option(b"zero_copy_server", 0))              # this test is about the one-launch-per-batch form
        for k, (pi, _po) in enumerate(bufs):
            C.memmove(pi, host[k * n:(k + 1) * n].ctypes.data, nbytes)
        for streams, cap in ((1, 1), (4, 1), (2, 5), (4, 24), (3, 100000), (4, 0)):
            pkg.check(L.mi_blur_set_option(b"zero_copy_streams", streams))
            pkg.check(L.mi_blur_set_option(b"zero_copy_blocks", cap))
            with pkg.Context(0, w, h, c, radius, max_batch=n, n_slots=4) as ctx:
                for (_pi, po) in bufs:
                    C.memset(po, 0xEE, nbytes)
                t0 = time.perf_counter()
                for rnd in range(3):
                    for (pi, po) in bufs:
                        ctx.submit(pi, po, n)
                tm = ctx.sync()
                wall_ms = (time.perf_counter() - t0) * 1e3
                assert L.mi_blur_zero_copy_launches(ctx.h) == 12
                # overlapping launches: the kernel bucket is the time at least one of them was executing, so it cannot
                # exceed the wall clock of the loop (a sum of overlapping durations would, with 4 streams)
                assert 0 < tm["kernel_ms"] <= wall_ms * 1.02 + 0.05, (streams, cap, tm["kernel_ms"], wall_ms)
                for k, (_pi, po) in enumerate(bufs):
                    got = np.ctypeslib.as_array((C.c_uint8 * nbytes).from_address(po)).reshape(n, h, w, c)
                    assert np.array_equal(got, want[k * n:(k + 1) * n]), (streams, cap, k)
    finally:
        pkg.check(L.mi_blur_set_option(b"zero_copy_server", 1))
        pkg.check(L.mi_blur_set_option(b"zero_copy_streams", 4))
        pkg.check(L.mi_blur_set_option(b"zero_copy_blocks", 24))
        for (pi, po) in bufs:
            L.mi_blur_host_free(pi); L.mi_blur_host_free(po)


@pytest.mark.parametrize("shape", [(1, 100000, 3), (100000, 1, 3), (3, 65536, 4), (70000, 16, 1), (2, 1000003, 1), (40000, 17, 3),
                                   (65537, 48, 2), (5, 262144, 3)])
def test_extreme_aspect_ratios(pkg, L, O, torch_cuda, shape):
    """One-row, one-column, million-pixel-wide and 70000-row frames (the reference kernel's NDRange takes any W x H): through the
    kernel-level entry and through a context with pageable buffers (staging + batch server), 3x3 and 5x5."""
    torch = torch_cuda
    h, w, c = shape
    img = O.lcg_image(h, w, c)
    for r in (1, 2):
        want = O.blur(img, r)
        d_in = torch.from_numpy(img.reshape(-1)).cuda()
        d_out = torch.zeros_like(d_in)
        pkg.check(L.mi_blur_enqueue(d_in.data_ptr(), d_out.data_ptr(), w, h, c, r, 1, None))
        torch.cuda.synchronize()
        assert np.array_equal(d_out.cpu().numpy().reshape(h, w, c), want), (shape, r, L.mi_blur_last_kernel())
        stack = np.stack([img, np.ascontiguousarray(img[::-1]), img])
        out = np.zeros_like(stack)
        with pkg.Context(0, w, h, c, r, max_batch=3, n_slots=2) as ctx:
            ctx.submit(stack.ctypes.data, out.ctypes.data, 3)
            ctx.sync()
        assert np.array_equal(out, O.blur_batch(stack, r)), (shape, r, "context")


def test_contexts_driven_from_different_threads(pkg, L, O, torch_cuda):
    """A context is single-threaded; DIFFERENT contexts may be driven from different host threads at once (the 8-GPU hosts do,
    one feeder thread per GPU).  Four threads, each with its own context on the one device — two with pinned buffers, two with
    pageable ones (shared gather/scatter pool, one batch server each) — stream at the same time; every batch of every thread is
    verified.  ctypes releases the GIL inside the library, so the threads really overlap."""
    import threading
    h, w, c, n, radius = 128, 256, 3, 16, 1                      # 1.5 MiB per batch: the server's side of the small-submit rule
    host = O.lcg_stream(n, h, w, c, first_index=500)
    want = O.blur_batch(host, radius)
    nbytes = host.nbytes
    errors = []

    def worker(tid, pageable):
        try:
            if pageable:
                bufs = [(host.copy(), np.zeros_like(host)) for _ in range(3)]
                ptrs = [(a.ctypes.data, b.ctypes.data) for a, b in bufs]
                view = lambda k: bufs[k][1]
            else:
                ptrs = [(L.mi_blur_host_alloc(nbytes), L.mi_blur_host_alloc(nbytes)) for _ in range(3)]
                for pi, _po in ptrs:
                    C.memmove(pi, host.ctypes.data, nbytes)
                view = lambda k: np.ctypeslib.as_array((C.c_uint8 * nbytes).from_address(ptrs[k][1])).reshape(host.shape)
            rng = np.random.default_rng(tid)
            sizes = [n] * 3
            with pkg.Context(0, w, h, c, radius, max_batch=n, n_slots=3) as ctx:
                for i in range(120):
                    k = i % 3
                    if i >= 3:
                        ctx.wait_oldest()
                        got = view(k)
                        if not np.array_equal(got[:sizes[k]], want[:sizes[k]]):
                            errors.append((tid, i - 3, "mismatch"))
                            return
                        got[:] = 0
                    sizes[k] = int(rng.integers(n // 2, n + 1))
                    ctx.submit(ptrs[k][0], ptrs[k][1], sizes[k])
                ctx.sync()
            if not pageable:
                for pi, po in ptrs:
                    L.mi_blur_host_free(pi); L.mi_blur_host_free(po)
        except Exception as e:                                   # noqa: BLE001 - reported below
            errors.append((tid, repr(e)))

    threads = [threading.Thread(target=worker, args=(t, t % 2 == 1)) for t in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not any(t.is_alive() for t in threads), "a feeder thread is stuck"
    assert not errors, errors


def test_batches_past_2_31_bytes(pkg, L, O, torch_cuda):
    """Eleven 8192x8192x3 frames in ONE submit — 2.2 GB each way, past 2^31 bytes — in place through the batch server and as one
    launch over a resident pool: every output frame carries the reference kernel's hash of that frame (tests/golden; the
    reference indexes with int, gaussian_kernel.cl:60, and would not get here).  tools/big_batch_check.py runs the other forms."""
    import json
    gold = json.load(open(os.path.join(pkg.ROOT, "tests", "golden", "blur_golden.json")))
    W = H = 8192; c = 3; n = 11
    isz = W * H * c
    want = "d283787bcc5b6dfd"
    assert any(want in json.dumps(v) for v in gold.values())            # the 8192^2 k3 entry
    p_in, p_out = L.mi_blur_host_alloc(n * isz), L.mi_blur_host_alloc(n * isz)
    assert p_in and p_out
    try:
        L.mi_blur_fill_synthetic(p_in, W, H, c, 0, 1, 8)
        for i in range(1, n):
            C.memmove(p_in + i * isz, p_in, isz)
        C.memset(p_out, 0, n * isz)
        with pkg.Context(0, W, H, c, 1, max_batch=n, n_slots=1) as ctx:
            ctx.submit(p_in, p_out, n)
            ctx.sync()
            assert L.mi_blur_last_kernel() == b"blur_server_kernel"
            assert [f"{L.mi_blur_fnv1a64(p_out + i * isz, isz):016x}" for i in range(n)] == [want] * n
        C.memset(p_out, 0, n * isz)
        with pkg.Context(0, W, H, c, 1, max_batch=1, n_slots=1) as ctx:
            ctx.resident_alloc(n)
            ctx.resident_upload(0, p_in, n)
            ctx.resident_run(n, n)
            ctx.sync()
            ctx.resident_download(0, p_out, n)
            assert [f"{L.mi_blur_fnv1a64(p_out + i * isz, isz):016x}" for i in range(n)] == [want] * n
    finally:
        L.mi_blur_host_free(p_in); L.mi_blur_host_free(p_out)


def test_small_in_place_submits_take_one_launch_each(pkg, L, O, torch_cuda):
    """In-place (pinned) submits below 1.25 MiB of output are not worth the server's hand-off (~26 us per batch against ~8 us for
    a launch): they take one launch each, bigger ones the server, on the same context, in any order; same bytes either way."""
    h, w, c, radius = 256, 256, 3, 1
    host = O.lcg_stream(10, h, w, c, first_index=77)
    want = O.blur_batch(host, radius)
    isz = h * w * c
    p_in, p_out = L.mi_blur_host_alloc(10 * isz), L.mi_blur_host_alloc(10 * isz)
    C.memmove(p_in, host.ctypes.data, 10 * isz)
    out = np.ctypeslib.as_array((C.c_uint8 * (10 * isz)).from_address(p_out)).reshape(10, h, w, c)
    try:
        with pkg.Context(0, w, h, c, radius, max_batch=10, n_slots=3) as ctx:
            for n, server in ((1, False), (6, False), (7, True), (10, True), (2, False), (10, True), (1, False)):
                out[:] = 0xEE
                ctx.submit(p_in, p_out, n)
                ctx.sync()
                assert (L.mi_blur_last_kernel() == b"blur_server_kernel") == server, (n, L.mi_blur_last_kernel())
                assert np.array_equal(out[:n], want[:n]) and bool((out[n:] == 0xEE).all()), n
            assert L.mi_blur_zero_copy_launches(ctx.h) == 7            # in place every time, whichever form
            # both forms in flight at once on rotating slots
            outs = [L.mi_blur_host_alloc(10 * isz) for _ in range(3)]
            sizes = [1, 10, 3, 8, 2, 9, 10, 1, 7]
            for i, n in enumerate(sizes):
                if i >= 3:
                    ctx.wait_oldest()
                    m = sizes[i - 3]
                    got = np.ctypeslib.as_array((C.c_uint8 * (m * isz)).from_address(outs[i % 3])).reshape(m, h, w, c)
                    assert np.array_equal(got, want[:m]), i - 3
                ctx.submit(p_in, outs[i % 3], n)
            ctx.sync()
            for o in outs:
                L.mi_blur_host_free(o)
    finally:
        L.mi_blur_host_free(p_in); L.mi_blur_host_free(p_out)


def test_batch_server_mixed_pinned_and_pageable_submits(pkg, L, O, torch_cuda, server_for_small_batches):
    """One context, one server: submits whose buffers are pinned (blurred in place), pageable (gathered into the slot's pinned
    staging, blurred there, scattered back when the slot is harvested) and half-and-half (pinned in / pageable out and the
    reverse), interleaved at random with random batch sizes; every batch verified and poisoned before its buffers are reused."""
    h, w, c, n, radius = 96, 320, 3, 9, 1
    nbuf = 4
    host = O.lcg_stream(nbuf * n, h, w, c, first_index=40)
    want = O.blur_batch(host, radius)
    nbytes = n * h * w * c
    pin = [(L.mi_blur_host_alloc(nbytes), L.mi_blur_host_alloc(nbytes)) for _ in range(nbuf)]
    pag = [(np.ascontiguousarray(host[k * n:(k + 1) * n]).copy(), np.full((n, h, w, c), 0xEE, np.uint8)) for k in range(nbuf)]
    as_np = lambda ptr: np.ctypeslib.as_array((C.c_uint8 * nbytes).from_address(ptr)).reshape(n, h, w, c)
    try:
        for k, (pi, po) in enumerate(pin):
            C.memmove(pi, host[k * n:(k + 1) * n].ctypes.data, nbytes)
            C.memset(po, 0xEE, nbytes)
        rng = np.random.default_rng(99)
        with pkg.Context(0, w, h, c, radius, max_batch=n, n_slots=nbuf) as ctx:
            kinds, sizes = [None] * nbuf, [n] * nbuf
            out_of = lambda k: as_np(pin[k][1]) if kinds[k][1] == "pin" else pag[k][1]
            for i in range(160):
                k = i % nbuf
                if i >= nbuf:
                    ctx.wait_oldest()
                    got = out_of(k)
                    assert np.array_equal(got[:sizes[k]], want[k * n:k * n + sizes[k]]), (i, kinds[k])
                    assert bool((got[sizes[k]:] == 0xEE).all()), (i, kinds[k])
                    got[:] = 0xEE
                kinds[k] = (str(rng.choice(["pin", "page"])), str(rng.choice(["pin", "page"])))
                sizes[k] = int(rng.integers(1, n + 1))
                src = pin[k][0] if kinds[k][0] == "pin" else pag[k][0].ctypes.data
                dst = pin[k][1] if kinds[k][1] == "pin" else pag[k][1].ctypes.data
                ctx.submit(src, dst, sizes[k])
            ctx.sync()
            for k in range(nbuf):
                assert np.array_equal(out_of(k)[:sizes[k]], want[k * n:k * n + sizes[k]]), k
            assert L.mi_blur_zero_copy_launches(ctx.h) == 160 and L.mi_blur_last_kernel() == b"blur_server_kernel"
    finally:
        for (pi, po) in pin:
            L.mi_blur_host_free(pi); L.mi_blur_host_free(po)


@pytest.mark.parametrize("base,budget", [(40, 256), (40, 7), (3, 256), (100, 5)])
def test_batch_server_number_wrap(pkg, L, O, torch_cuda, base, budget, server_for_small_batches):
    """The server numbers batches and tiles in 32 bits through the life of a context; a continuous batch-35 stream of 256x256
    frames uses up the tile numbers in ~80 minutes.  "zero_copy_debug_base" starts a new server `base` batches (7*base tiles)
    short of 2^32, so both counters wrap inside this test — at different batches, with the server rolling over (budget) before,
    at and after the wrap, with the descriptor ring coming round across it.  Every batch is verified and its output poisoned
    again before its buffers are reused."""
    h, w, c, n, radius = 96, 320, 3, 9, 1
    nbuf = 4
    host = O.lcg_stream(nbuf * n, h, w, c, first_index=900)
    want = O.blur_batch(host, radius)
    nbytes = n * h * w * c
    bufs = [(L.mi_blur_host_alloc(nbytes), L.mi_blur_host_alloc(nbytes)) for _ in range(nbuf)]
    as_np = lambda ptr, m=n: np.ctypeslib.as_array((C.c_uint8 * (m * h * w * c)).from_address(ptr)).reshape(m, h, w, c)
    pkg.check(L.mi_blur_set_option(b"zero_copy_debug_base", base))
    pkg.check(L.mi_blur_set_option(b"zero_copy_budget", budget))
    pkg.check(L.mi_blur_set_option(b"zero_copy_trace", 1))        # only for the server's own batch count below
    try:
        for k, (pi, po) in enumerate(bufs):
            C.memmove(pi, host[k * n:(k + 1) * n].ctypes.data, nbytes)
            C.memset(po, 0xEE, nbytes)
        rng = np.random.default_rng(base * 31 + budget)
        with pkg.Context(0, w, h, c, radius, max_batch=n, n_slots=nbuf) as ctx:
            sizes = [n] * nbuf
            total = 3 * base + 150
            for i in range(total):
                k = i % nbuf
                if i >= nbuf:
                    ctx.wait_oldest()
                    assert np.array_equal(as_np(bufs[k][1])[:sizes[k]], want[k * n:k * n + sizes[k]]), i
                    assert bool((as_np(bufs[k][1])[sizes[k]:] == 0xEE).all()), i
                    C.memset(bufs[k][1], 0xEE, nbytes)
                sizes[k] = int(rng.integers(1, n + 1))
                if i % 29 == 28:
                    time.sleep(0.001)                                          # idle time-out: a fresh server picks up across the wrap too
                ctx.submit(bufs[k][0], bufs[k][1], sizes[k])
            ctx.sync()
            for k in range(nbuf):
                assert np.array_equal(as_np(bufs[k][1])[:sizes[k]], want[k * n:k * n + sizes[k]]), k
            assert L.mi_blur_zero_copy_launches(ctx.h) == total and L.mi_blur_last_kernel() == b"blur_server_kernel"
            # the server's own batch number has gone through 2^32: it started at 2^32 - base and stands at total - base now
            nw, head = C.c_int(), C.c_uint()
            tr = np.zeros(48 * 5, np.uint64)
            pkg.check(min(0, L.mi_blur_debug_zc_trace(ctx.h, tr.ctypes.data_as(C.POINTER(C.c_uint64)), 1, C.byref(nw), C.byref(head))))
            assert head.value == total - base
    finally:
        pkg.check(L.mi_blur_set_option(b"zero_copy_trace", 0))
        pkg.check(L.mi_blur_set_option(b"zero_copy_debug_base", 0))
        pkg.check(L.mi_blur_set_option(b"zero_copy_budget", 256))
        for (pi, po) in bufs:
            L.mi_blur_host_free(pi); L.mi_blur_host_free(po)


@pytest.mark.parametrize("radius", [1, 2])
def test_zero_copy_batch_server(pkg, L, O, torch_cuda, radius, server_for_small_batches):
    """The default path of pinned host-to-host submits: ONE long-lived dispatch (blur_server_kernel) takes batch after batch
    from a descriptor ring, tiles handed out by a ticket counter that runs through the batches, completion per batch
    signalled into host memory.  Bit-exact vs the oracle for: a back-to-back stream on rotating buffers; a producer slower
    than the server's idle time-out (every batch finds the server gone and the queued one picks it up); a budget of 3
    batches per server (roll-over to the next queued server in mid-stream); 1, 5 and 300 workers (fewer / more workers than
    tiles); several contexts at once; batches of different sizes and a band submit of another geometry in between (that
    one takes the per-batch launch); destroy right after the last submit."""
    h, w, c, n = 96, 320, 3, 9
    rounds, nbuf = 5, 4
    host = O.lcg_stream(nbuf * n, h, w, c, first_index=300)
    want = O.blur_batch(host, radius)
    nbytes = n * h * w * c
    bufs = [(L.mi_blur_host_alloc(nbytes), L.mi_blur_host_alloc(nbytes)) for _ in range(nbuf)]
    as_np = lambda ptr, m=n: np.ctypeslib.as_array((C.c_uint8 * (m * h * w * c)).from_address(ptr)).reshape(m, h, w, c)
    try:
        for k, (pi, _po) in enumerate(bufs):
            C.memmove(pi, host[k * n:(k + 1) * n].ctypes.data, nbytes)

        def run(ctx, delay=0.0, sizes=None):
            for (_pi, po) in bufs:
                C.memset(po, 0xEE, nbytes)
            for rnd in range(rounds):
                for k, (pi, po) in enumerate(bufs):
                    if delay:
                        time.sleep(delay)
                    ctx.submit(pi, po, sizes[k] if sizes else n)
            tm = ctx.sync()
            for k, (_pi, po) in enumerate(bufs):
                m = sizes[k] if sizes else n
                assert np.array_equal(as_np(po)[:m], want[k * n:k * n + m]), k
                assert bool((as_np(po)[m:] == 0xEE).all())
            return tm

        for opts, delay in (({}, 0.0), ({"zero_copy_idle_us": 50}, 0.002), ({"zero_copy_budget": 3}, 0.0), ({"zero_copy_workers": 1}, 0.0),
                            ({"zero_copy_workers": 5}, 0.0), ({"zero_copy_workers": 300}, 0.0)):
            for key, v in opts.items():
                pkg.check(L.mi_blur_set_option(key.encode(), v))
            try:
                with pkg.Context(0, w, h, c, radius, max_batch=n, n_slots=nbuf) as ctx:
                    t0 = time.perf_counter()
                    tm = run(ctx, delay)
                    wall_ms = (time.perf_counter() - t0) * 1e3
                    assert L.mi_blur_zero_copy_launches(ctx.h) == rounds * nbuf and L.mi_blur_last_kernel() == b"blur_server_kernel"
                    assert tm["images"] == rounds * nbuf * n and 0 < tm["kernel_ms"] <= wall_ms * 1.02 + 0.05, (opts, tm, wall_ms)
                    assert tm["h2d_ms"] == 0 and tm["d2h_ms"] == 0               # the transfer time IS the kernel bucket
                    run(ctx, sizes=[n, 1, 4, n - 1])                             # batches of different sizes through the same server
            finally:
                for key, v in (("zero_copy_idle_us", 300), ("zero_copy_budget", 256), ("zero_copy_workers", 48)):
                    pkg.check(L.mi_blur_set_option(key.encode(), v))

        # a long stream (the descriptor ring, 64 entries, comes round several times; a budget of 7 rolls the server over in
        # mid-stream; random batch sizes): every batch is checked and its output poisoned again BEFORE its buffers are reused,
        # so a batch that was skipped, or done twice into the wrong buffer, cannot hide behind an earlier one's bytes
        pkg.check(L.mi_blur_set_option(b"zero_copy_budget", 7))
        try:
            rng = np.random.default_rng(5 + radius)
            with pkg.Context(0, w, h, c, radius, max_batch=n, n_slots=nbuf) as ctx:
                sizes = [n] * nbuf
                for (_pi, po) in bufs:
                    C.memset(po, 0xEE, nbytes)
                for i in range(230):
                    k = i % nbuf
                    if i >= nbuf:
                        ctx.wait_oldest()                                      # batch i - nbuf: the one that used these buffers
                        assert np.array_equal(as_np(bufs[k][1])[:sizes[k]], want[k * n:k * n + sizes[k]]), i
                        assert bool((as_np(bufs[k][1])[sizes[k]:] == 0xEE).all()), i
                        C.memset(bufs[k][1], 0xEE, nbytes)
                    sizes[k] = int(rng.integers(1, n + 1))
                    if i % 37 == 36:
                        time.sleep(0.001)                                      # now and then the producer stalls past the idle time-out
                    ctx.submit(bufs[k][0], bufs[k][1], sizes[k])
                ctx.sync()
                for k in range(nbuf):
                    assert np.array_equal(as_np(bufs[k][1])[:sizes[k]], want[k * n:k * n + sizes[k]]), k
                assert L.mi_blur_zero_copy_launches(ctx.h) == 230
        finally:
            pkg.check(L.mi_blur_set_option(b"zero_copy_budget", 256))

        # three contexts side by side (each has its own server), fed in turn
        ctxs = [pkg.Context(0, w, h, c, radius, max_batch=n, n_slots=2) for _ in range(3)]
        outs = [[L.mi_blur_host_alloc(nbytes) for _ in range(nbuf)] for _ in ctxs]
        for rnd in range(3):
            for k in range(nbuf):
                for ci, ctx in enumerate(ctxs):
                    ctx.submit(bufs[k][0], outs[ci][k], n)
        for ci, ctx in enumerate(ctxs):
            ctx.sync()
            for k in range(nbuf):
                assert np.array_equal(as_np(outs[ci][k]), want[k * n:(k + 1) * n]), (ci, k)
        # a band submit of another geometry in between takes the per-batch launch; the server carries on afterwards
        ctx = ctxs[0]
        one = np.ascontiguousarray(host[0])
        p_one, p_top = L.mi_blur_host_alloc(one.nbytes), L.mi_blur_host_alloc(one.nbytes)
        C.memmove(p_one, one.ctypes.data, one.nbytes)
        ctx.submit(bufs[1][0], outs[0][1], n)
        ctx.submit_band(p_one, p_top, 39 + radius, 0, radius)
        ctx.submit(bufs[2][0], outs[0][2], n)
        ctx.sync()
        assert np.array_equal(np.ctypeslib.as_array((C.c_uint8 * (39 * w * c)).from_address(p_top)).reshape(39, w, c), want[0][:39])
        assert np.array_equal(as_np(outs[0][1]), want[n:2 * n]) and np.array_equal(as_np(outs[0][2]), want[2 * n:3 * n])
        # frames whose rows are not a multiple of 16 bytes (250x167x3, pitch 750; 33x50x1) go through the server too — the ragged
        # form of the same tile code — from buffers at odd addresses as well
        for (rh, rw, rc, rn) in ((167, 250, 3, 7), (50, 33, 1, 5), (64, 1366, 3, 3)):
            rhost = O.lcg_stream(rn, rh, rw, rc, first_index=900)
            rwant = O.blur_batch(rhost, radius)
            rbytes = rhost.nbytes
            rin, rout = L.mi_blur_host_alloc(rbytes + 64), L.mi_blur_host_alloc(rbytes + 64)
            for off in (0, 3):
                C.memmove(rin + off, rhost.ctypes.data, rbytes)
                C.memset(rout, 0xEE, rbytes + 64)
                with pkg.Context(0, rw, rh, rc, radius, max_batch=rn, n_slots=2) as rctx:
                    for _ in range(3):
                        rctx.submit(rin + off, rout + off, rn)
                    rctx.sync()
                    assert L.mi_blur_last_kernel() == b"blur_server_kernel" and L.mi_blur_zero_copy_launches(rctx.h) == 3
                got = np.ctypeslib.as_array((C.c_uint8 * (rbytes + 64)).from_address(rout))
                assert np.array_equal(got[off:off + rbytes].reshape(rhost.shape), rwant), (rh, rw, rc, off)
                assert bool((got[:off] == 0xEE).all()) and bool((got[off + rbytes:] == 0xEE).all())      # nothing written outside
            L.mi_blur_host_free(rin); L.mi_blur_host_free(rout)
        # the diagnostics trace: every tile of every batch is accounted for
        pkg.check(L.mi_blur_set_option(b"zero_copy_trace", 1))
        try:
            with pkg.Context(0, w, h, c, radius, max_batch=n, n_slots=2) as tctx:
                for k in range(6):
                    tctx.submit(bufs[k % nbuf][0], bufs[k % nbuf][1], n)
                tctx.sync()
                nw, head = C.c_int(), C.c_uint()
                tr = np.zeros(6 * 48 * 5, np.uint64)
                got_n = L.mi_blur_debug_zc_trace(tctx.h, tr.ctypes.data_as(C.POINTER(C.c_uint64)), 6, C.byref(nw), C.byref(head))
                assert got_n == 6 and nw.value == 48 and head.value == 6
                tiles = tr.reshape(6, 48, 5)[:, :, 1]
                assert (tiles.sum(axis=1) == tiles.sum(axis=1)[0]).all() and tiles.sum(axis=1)[0] > 0      # same tile count per (equal) batch
        finally:
            pkg.check(L.mi_blur_set_option(b"zero_copy_trace", 0))
        # destroy with work just submitted: the context drains it, tells its servers to leave and waits for them
        ctxs[1].submit(bufs[3][0], outs[1][3], n)
        for ctx in ctxs:
            ctx.close()
        assert np.array_equal(as_np(outs[1][3]), want[3 * n:4 * n])
        for o in outs:
            for ptr in o:
                L.mi_blur_host_free(ptr)
        L.mi_blur_host_free(p_one); L.mi_blur_host_free(p_top)
    finally:
        for (pi, po) in bufs:
            L.mi_blur_host_free(pi); L.mi_blur_host_free(po)


def test_numpy_convenience(pkg, O, torch_cuda):
    """pkg.blur() on the GPU: pageable numpy arrays through the staged submit path, one submit and several."""
    stack = O.lcg_stream(40, 96, 128, 3, first_index=5)
    for ksize in (3, 5):
        want = O.blur_batch(stack, (ksize - 1) // 2)
        assert np.array_equal(pkg.blur(stack, ksize), want)
        assert np.array_equal(pkg.blur(stack, ksize, batch=16), want)
        assert np.array_equal(pkg.blur(stack[9], ksize), want[9])
    odd = O.lcg_stream(3, 31, 50, 3)                       # rows of 150 bytes: not the tiled kernel's shape
    assert np.array_equal(pkg.blur(odd), O.blur_batch(odd, 1))


@pytest.mark.parametrize("radius", [1, 2])
def test_halo_pull_fills_the_halo_rows(pkg, L, O, torch_cuda, radius):
    """mi_blur_halo_pull: a rank's halo rows copied straight out of its neighbours' shards by one small kernel (here the
    "peers" are other tensors on the same device; across processes the pointers come from mi_blur_peer_open — rehearsed by
    tests/test_cli.py::test_bench_spawns_its_own_ranks).  Three row shards of one image, pulled halos, band blur == whole-image
    blur; aligned rows and rows that are not a multiple of 16 bytes; first / middle / last shard (one or two neighbours)."""
    torch = torch_cuda
    for (h, w, c) in ((96, 320, 3), (61, 250, 3), (48, 33, 1)):
        img = O.lcg_image(h, w, c)
        want = O.blur(img, radius)
        pitch = w * c
        G = 3
        bands = [pkg.band_of(h, radius, g, G) for g in range(G)]
        bufs = []
        for b in bands:                                # every shard: [halo_top poison][owned rows][halo_bottom poison]
            owned = b["row_end"] - b["row_begin"]
            t = torch.full(((b["halo_top"] + owned + b["halo_bottom"]) * pitch,), 0xA5, dtype=torch.uint8, device="cuda")
            t[b["halo_top"] * pitch:(b["halo_top"] + owned) * pitch] = torch.from_numpy(img[b["row_begin"]:b["row_end"]].reshape(-1)).cuda()
            bufs.append(t)
        got = np.zeros_like(img)
        for g, b in enumerate(bands):
            owned = b["row_end"] - b["row_begin"]
            top_src = bottom_src = None
            if g > 0:
                a = bands[g - 1]
                top_src = bufs[g - 1].data_ptr() + (a["halo_top"] + a["row_end"] - a["row_begin"] - radius) * pitch
            if g < G - 1:
                bottom_src = bufs[g + 1].data_ptr() + bands[g + 1]["halo_top"] * pitch
            pkg.check(L.mi_blur_halo_pull(bufs[g].data_ptr(), top_src, bottom_src, w, c, owned, radius, None), "halo_pull")
            rows = b["halo_top"] + owned + b["halo_bottom"]
            out = torch.zeros(owned * pitch, dtype=torch.uint8, device="cuda")
            pkg.check(L.mi_blur_enqueue_band(bufs[g].data_ptr(), out.data_ptr(), w, rows, c, radius, b["halo_top"], b["halo_top"] + owned, None))
            torch.cuda.synchronize()
            got[b["row_begin"]:b["row_end"]] = out.cpu().numpy().reshape(owned, w, c)
        assert np.array_equal(got, want), (h, w, c)
    assert L.mi_blur_halo_pull(None, None, None, 16, 3, 8, 1, None) == pkg.ERR_INVALID
    t = torch.zeros(1024, dtype=torch.uint8, device="cuda")
    assert L.mi_blur_halo_pull(t.data_ptr(), None, None, 16, 3, 8, 1, None) == pkg.OK          # no neighbour: nothing to do


@pytest.mark.parametrize("radius", [1, 2])
def test_band_reads_its_halo_rows_from_the_neighbouring_shards(pkg, L, O, torch_cuda, radius):
    """mi_blur_enqueue_band_peer: the Approach-2 step across GPUs as ONE launch — the band kernel reads the rows above and below
    its output rows out of the neighbouring shards (peer pointers; other tensors on the same device here) and the shard's own halo
    rows stay poisoned, untouched.  Shards of one image, every output row == whole-image blur; 1-4 channels, tall and short
    shards (a shard shorter than one 8-row lane band included), G = 2..5; ineligible shapes are refused, not mis-served."""
    torch = torch_cuda
    for (h, w, c, G) in ((96, 320, 3, 3), (64, 64, 1, 2), (40, 128, 4, 5), (33, 48, 2, 4), (1024, 512, 3, 2)):
        img = O.lcg_image(h, w, c)
        want = O.blur(img, radius)
        pitch = w * c
        bands = [pkg.band_of(h, radius, g, G) for g in range(G)]
        bufs = []
        for b in bands:
            owned = b["row_end"] - b["row_begin"]
            t = torch.full(((b["halo_top"] + owned + b["halo_bottom"]) * pitch,), 0xA5, dtype=torch.uint8, device="cuda")
            t[b["halo_top"] * pitch:(b["halo_top"] + owned) * pitch] = torch.from_numpy(img[b["row_begin"]:b["row_end"]].reshape(-1)).cuda()
            bufs.append(t)
        got = np.zeros_like(img)
        for g, b in enumerate(bands):
            owned = b["row_end"] - b["row_begin"]
            top_src = bottom_src = None
            if g > 0:
                a = bands[g - 1]
                top_src = bufs[g - 1].data_ptr() + (a["halo_top"] + a["row_end"] - a["row_begin"] - radius) * pitch
            if g < G - 1:
                bottom_src = bufs[g + 1].data_ptr() + bands[g + 1]["halo_top"] * pitch
            rows = b["halo_top"] + owned + b["halo_bottom"]
            out = torch.zeros(owned * pitch, dtype=torch.uint8, device="cuda")
            pkg.check(L.mi_blur_enqueue_band_peer(bufs[g].data_ptr(), out.data_ptr(), w, rows, c, radius, b["halo_top"],
                                                  b["halo_top"] + owned, top_src, bottom_src, None), "enqueue_band_peer")
            torch.cuda.synchronize()
            assert "peer halo rows" in L.mi_blur_last_kernel().decode()
            got[b["row_begin"]:b["row_end"]] = out.cpu().numpy().reshape(owned, w, c)
            halo = torch.cat([bufs[g][:b["halo_top"] * pitch], bufs[g][(b["halo_top"] + owned) * pitch:]])
            assert bool((halo == 0xA5).all()), "the shard's own halo rows are neither needed nor written"
        assert np.array_equal(got, want), (h, w, c, G)
    # rows that are not whole 16-byte chunks, more than 4 channels: not this kernel's shapes
    t = torch.zeros(64 * 250 * 3, dtype=torch.uint8, device="cuda")
    o = torch.zeros_like(t)
    assert L.mi_blur_enqueue_band_peer(t.data_ptr(), o.data_ptr(), 250, 64, 3, radius, radius, 60, t.data_ptr(), None, None) == pkg.ERR_UNSUPPORTED
    t = torch.zeros(64 * 64 * 6, dtype=torch.uint8, device="cuda")
    o = torch.zeros_like(t)
    assert L.mi_blur_enqueue_band_peer(t.data_ptr(), o.data_ptr(), 64, 64, 6, radius, radius, 60, t.data_ptr(), None, None) == pkg.ERR_UNSUPPORTED
    assert L.mi_blur_enqueue_band_peer(t.data_ptr(), o.data_ptr(), 64, 64, 4, radius, radius, 60, t.data_ptr() + 4, None, None) == pkg.ERR_INVALID
    # no neighbour on either side: the plain band launch
    img = O.lcg_image(32, 64, 3)
    t = torch.from_numpy(img.reshape(-1)).cuda()
    o = torch.zeros_like(t)
    pkg.check(L.mi_blur_enqueue_band_peer(t.data_ptr(), o.data_ptr(), 64, 32, 3, radius, 0, 32, None, None, None))
    torch.cuda.synchronize()
    assert np.array_equal(o.cpu().numpy().reshape(32, 64, 3), O.blur(img, radius))


def test_resident_pool_placement_trials(pkg, L, O, torch_cuda):
    """mi_blur_resident_alloc places big pools: it times a few candidate placements on the context's kernel and keeps the
    fastest (where a pool lands in HBM moves big launches between two levels ~6 % apart).  The choice is reported; small pools
    and "resident_place_trials" 0 take the first allocation; results do not depend on any of it."""
    h, w, c, n = 256, 256, 3, 1500                        # 295 MB per buffer
    want = None
    for trials in (4, 2, 0):
        pkg.check(L.mi_blur_set_option(b"resident_place_trials", trials))
        try:
            with pkg.Context(0, w, h, c, 1, max_batch=1, n_slots=1) as ctx:
                ctx.resident_alloc(n)
                pl = ctx.resident_placement()
                assert len(pl["candidates_us"]) == (trials * trials if trials > 1 else 0), pl      # every (input, output) pair
                if trials > 1:
                    assert 0 <= pl["kept"] < trials * trials and min(pl["candidates_us"]) == pl["candidates_us"][pl["kept"]] > 0
                ctx.resident_fill_synthetic(0)
                ctx.resident_run(n, n); ctx.sync()
                out = np.zeros((3, h, w, c), np.uint8)
                ctx.resident_download(n - 3, out.ctypes.data, 3)
                if want is None:
                    want = O.blur_batch(O.lcg_stream(3, h, w, c, first_index=n - 3), 1)
                assert np.array_equal(out, want)
                ctx.resident_alloc(8)                      # a small pool: no trial
                assert ctx.resident_placement()["candidates_us"] == []
        finally:
            pkg.check(L.mi_blur_set_option(b"resident_place_trials", 4))


def test_submit_bands_strided_batch(pkg, L, O, torch_cuda):
    """Approach 2 for a whole batch (mi_blur_submit_bands): the same rows of every image of a contiguous batch
    stream, gathered by one 2-D DMA (pinned memory) or through the pinned staging slot (pageable memory)."""
    h, w, c, n, split = 96, 80, 3, 7, 31
    host = O.lcg_stream(n, h, w, c)
    want = O.blur_batch(host, 1)
    pitch, isz = w * c, h * w * c
    for pinned in (False, True, "staged"):
        for n_slots in (1, 3):
            pkg.check(L.mi_blur_set_option(b"zero_copy", 0 if pinned == "staged" else 1))
            with pkg.Context(0, w, h, c, 1, max_batch=n, n_slots=n_slots) as ctx:
                if pinned:
                    p_in, p_out = L.mi_blur_host_alloc(host.nbytes), L.mi_blur_host_alloc(host.nbytes)
                    C.memmove(p_in, host.ctypes.data, host.nbytes)
                    C.memset(p_out, 0, host.nbytes)
                else:
                    out = np.zeros_like(host)
                    p_in, p_out = host.ctypes.data, out.ctypes.data
                # top band: rows [0, split) from input rows [0, split+1); bottom band: rows [split, h) from [split-1, h)
                pkg.check(L.mi_blur_submit_bands(ctx.h, p_in, p_out, n, isz, split + 1, 0, 1), "submit_bands top")
                pkg.check(L.mi_blur_submit_bands(ctx.h, p_in + (split - 1) * pitch, p_out + split * pitch, n, isz,
                                                 h - split + 1, 1, 0), "submit_bands bottom")
                tm = ctx.sync()
                assert tm["launches"] == 2 and tm["images"] == 2 * n
                # pinned + zero_copy: the strided bands are blurred in place in the caller's batch buffers
                # pageable: through the pinned staging — the first band by the batch server, the second (another tile geometry
                # than the running server's) by a DMA copy each way, unless it happens to share the geometry
                zc = L.mi_blur_zero_copy_launches(ctx.h)
                assert (zc == 2) if pinned is True else (zc == 0) if pinned == "staged" else zc in (1, 2)
                if pinned:
                    got = np.ctypeslib.as_array((C.c_uint8 * host.nbytes).from_address(p_out)).reshape(host.shape).copy()
                    L.mi_blur_host_free(p_in); L.mi_blur_host_free(p_out)
                else:
                    got = out
                assert np.array_equal(got, want), (pinned, n_slots)
    pkg.check(L.mi_blur_set_option(b"zero_copy", 1))


def test_batch_beyond_4gib(pkg, L, O, torch_cuda):
    """One launch over a batch whose input and output each exceed 4 GiB (22 000 x 256x256x3 = 4.33 GB): image offsets
    must be 64-bit in every variant.  Checked on the images either side of the 2^32-byte line, the first and the
    last, against the oracle; and the three variants must agree on the whole output."""
    torch = torch_cuda
    h, w, c, n = 256, 256, 3, 22000
    isz = h * w * c
    assert n * isz > (1 << 32)
    g = torch.Generator(device="cuda"); g.manual_seed(7)
    d_in = torch.randint(0, 256, (n, h, w, c), dtype=torch.uint8, device="cuda", generator=g)
    edge = (1 << 32) // isz
    sample = [0, 1, edge - 1, edge, edge + 1, n - 2, n - 1]
    want = {i: O.blur(np.ascontiguousarray(d_in[i].cpu().numpy()), 1) for i in sample}
    outs = []
    try:
        for variant in (pkg.VARIANT_TILED, pkg.VARIANT_STREAM, pkg.VARIANT_GENERIC):
            d_out = torch.zeros_like(d_in)
            pkg.check(L.mi_blur_enqueue_ex(d_in.data_ptr(), d_out.data_ptr(), w, h, c, 1, n, 0, h, variant, None))
            torch.cuda.synchronize()
            for i in sample:
                assert np.array_equal(d_out[i].cpu().numpy(), want[i]), (variant, i)
            outs.append(d_out)
        assert bool((outs[0] == outs[1]).all()) and bool((outs[0] == outs[2]).all())
    finally:
        reset_opts(L)


def test_resident_stream(pkg, L, O, torch_cuda):
    """Device-resident image stream: pool fill (synthetic LCG), batched passes, wrap-around."""
    h, w, c, pool = 64, 64, 3, 100
    with pkg.Context(0, w, h, c, 1, max_batch=35, n_slots=3) as ctx:
        ctx.resident_alloc(pool)
        ctx.resident_fill_synthetic(0)
        ctx.resident_run(250, 35, timed=2)         # 8 launches (7x35 + 5), wraps the 100-image pool
        tm = ctx.sync()
        assert tm["images"] == 250 and tm["launches"] == 8 and tm["kernel_ms"] > 0
        assert ctx.timed_coverage() == (4, (35 * 3 + 35) * 2 * h * w * c)   # launches 0,2,4,6 of 7x35+5
        assert tm["bytes_alg"] == 250 * 2 * h * w * c
        out = np.zeros((pool, h, w, c), np.uint8)
        ctx.resident_download(0, out.ctypes.data, pool)
        src = O.lcg_stream(pool, h, w, c)
        # pool images 0..69 are certainly written (35+35 twice over), and every written image must be exact
        want = O.blur_batch(src, 1)
        assert np.array_equal(out[:70], want[:70])
        ctx.reset_timing()
        ctx.resident_run(pool, pool, timed=False)     # one launch over the whole pool
        tm = ctx.sync()
        assert tm["launches"] == 1 and tm["kernel_ms"] == 0
        ctx.resident_download(0, out.ctypes.data, pool)
        assert np.array_equal(out, want)


def test_full_size_properties(pkg, L, O, torch_cuda):
    """BASELINE sizes, checked through size-independent properties (the oracle would take minutes):
    (i) 5000x256x256x3 stream: every output image of the synthetic stream hashes like the oracle's on a
    sample, and blur(all-255)=255 / blur(0)=0; (ii) 1920x1080 5x5 and 8192x8192 3x3: K-way split ==
    whole image (computed on the GPU both ways), plus oracle rows on a sampled band."""
    torch = torch_cuda
    h, w, c, n = 256, 256, 3, 5000
    with pkg.Context(0, w, h, c, 1, max_batch=35, n_slots=2) as ctx:
        ctx.resident_alloc(n)
        ctx.resident_fill_synthetic(0)
        ctx.resident_run(n, 35)
        ctx.sync()
        for idx in (0, 1, 34, 35, 2499, 4969, 4999):
            got = np.zeros((1, h, w, c), np.uint8)
            ctx.resident_download(idx, got.ctypes.data, 1)
            assert np.array_equal(got[0], O.blur(O.lcg_stream(1, h, w, c, first_index=idx)[0], 1)), idx
    # 8192x8192x3: whole vs 8-way split on the GPU; sampled rows vs oracle
    H = W = 8192
    img = O.lcg_image(H, W, 3)
    d_in = torch.from_numpy(img).cuda()
    d_whole = torch.empty_like(d_in)
    pkg.check(L.mi_blur_enqueue(d_in.data_ptr(), d_whole.data_ptr(), W, H, 3, 1, 1, None))
    d_split = torch.empty_like(d_in)
    pitch = W * 3
    for g in range(8):
        b = pkg.band_of(H, 1, g, 8)
        r0 = b["row_begin"] - b["halo_top"]
        rows = b["row_end"] + b["halo_bottom"] - r0
        pkg.check(L.mi_blur_enqueue_band(d_in.data_ptr() + r0 * pitch, d_split.data_ptr() + b["row_begin"] * pitch,
                                         W, rows, 3, 1, b["halo_top"], b["halo_top"] + b["row_end"] - b["row_begin"], None))
    torch.cuda.synchronize()
    assert bool((d_whole == d_split).all())
    whole = d_whole.cpu().numpy()
    for r0 in (0, 1023, 4090, 8192 - 40):
        band = np.ascontiguousarray(img[max(r0 - 1, 0): r0 + 41])
        ref = O.blur(band, 1)
        off = r0 - max(r0 - 1, 0)
        assert np.array_equal(whole[r0:r0 + 39], ref[off:off + 39]), r0
    # 1920x1080 5x5
    img = O.lcg_image(1080, 1920, 3)
    got = gpu_blur(pkg, L, torch, img[None], 2)[0]
    assert np.array_equal(got[:64], O.blur(np.ascontiguousarray(img[:66]), 2)[:64])
    assert np.array_equal(got[-64:], O.blur(np.ascontiguousarray(img[-66:]), 2)[-64:])
    flat = gpu_blur(pkg, L, torch, np.full((1, 1080, 1920, 3), 255, np.uint8), 2)
    assert int(flat.min()) == 255


def test_rccl_loads_and_single_rank_comm(pkg, L, torch_cuda):
    """The RCCL leg cannot be exercised across GPUs on a one-GPU box; check what can be: the library resolves
    RCCL lazily (sharing torch's copy when one is loaded), hands out a unique id, and a 1-rank communicator's
    exchange is a no-op (both image edges clamp).  The exchange pattern itself: tests/test_multi_rank.py."""
    raw = (C.c_uint8 * pkg.UNIQUE_ID_BYTES)()
    pkg.check(L.mi_blur_comm_unique_id(raw), "mi_blur_comm_unique_id")
    assert any(raw), "unique id is all zeros"
    comm = C.c_void_p()
    pkg.check(L.mi_blur_comm_init_rank(C.byref(comm), 1, 0, raw), "mi_blur_comm_init_rank")
    band = torch_cuda.zeros(64 * 48 * 3, dtype=torch_cuda.uint8, device="cuda")
    pkg.check(L.mi_blur_halo_exchange(comm, band.data_ptr(), 48, 3, 64, 1, None), "mi_blur_halo_exchange")
    assert L.mi_blur_halo_exchange(comm, band.data_ptr(), 48, 3, 0, 1, None) == pkg.ERR_INVALID   # owned < radius
    L.mi_blur_comm_destroy(comm)
    assert L.mi_blur_comm_init_rank(C.byref(comm), 2, 5, raw) == pkg.ERR_INVALID


def test_randomised_sweep(pkg, L, O, torch_cuda):
    """Seeded random sweep over shapes, channels, radius, batch, band row ranges, variants and tuning knobs:
    every combination must reproduce the oracle byte for byte."""
    # MI_BLUR_SWEEP_CASES / MI_BLUR_SWEEP_SEED: a longer or differently seeded run of the same sweep (soak runs)
    rng = np.random.default_rng(int(os.environ.get("MI_BLUR_SWEEP_SEED", "20261004")))
    try:
        for case in range(int(os.environ.get("MI_BLUR_SWEEP_CASES", "160"))):
            c = int(rng.choice([1, 2, 3, 4, 5]))
            radius = int(rng.choice([1, 2]))
            if rng.random() < 0.75 and c <= 4:                     # tiled/stream eligible: pitch multiple of 16
                w = int(rng.integers(1, 90)) * 16 // np.gcd(16, c)
                variant = int(rng.choice([pkg.VARIANT_AUTO, pkg.VARIANT_TILED, pkg.VARIANT_STREAM]))
            else:                                                  # any width: ragged tiled (pitch >= 16, C <= 4) or generic
                w = int(rng.integers(1, 200))
                variant = int(rng.choice([pkg.VARIANT_AUTO, pkg.VARIANT_GENERIC] + ([pkg.VARIANT_TILED] if w * c >= 16 and c <= 4 else [])))
            assert variant != pkg.VARIANT_STREAM or (w * c) % 16 == 0
            h = int(rng.integers(1, 150))
            n = int(rng.integers(1, 6))
            y0 = int(rng.integers(0, h))
            y1 = int(rng.integers(y0 + 1, h + 1))
            if rng.random() < 0.5:
                y0, y1 = 0, h
            opts = {"stage_dma": int(rng.integers(0, 2)), "rows_per_thread": int(rng.choice([0, 4, 8, 16])),
                    "xcd_remap": int(rng.integers(0, 2)), "row_shuffle": int(rng.integers(0, 2)),
                    "stream_band_rows": int(rng.choice([0, 3, 8, 40])), "prefer_stream": int(rng.integers(0, 2))}
            host = rng.integers(0, 256, (n, h, w, c), dtype=np.uint8)
            got = gpu_blur(pkg, L, torch_cuda, host, radius, variant, y0=y0, y1=y1, opts=opts)
            want = want_batch(O, host, radius)[:, y0:y1]
            assert np.array_equal(got, want), f"case {case}: shape {(n, h, w, c)} r{radius} rows [{y0},{y1}) variant {variant} {opts}"
    finally:
        reset_opts(L)
        L.mi_blur_set_option(b"prefer_stream", 0)


def test_p2p_halo_exchange_layout(pkg, L, O, torch_cuda):
    """mi_blur_halo_exchange_all with the peer-copy transport on 3 shards of one device: after the exchange every
    halo row holds the neighbour's edge row, owned rows are untouched, and blurring the bands reproduces the whole."""
    torch = torch_cuda
    H, W, Cn, R, G = 50, 32, 3, 2, 3
    img = O.lcg_image(H, W, Cn)
    pitch = W * Cn
    bands, tens, owned = [], [], []
    for g in range(G):
        b = pkg.band_of(H, R, g, G)
        rows = b["row_end"] - b["row_begin"] + b["halo_top"] + b["halo_bottom"]
        t = torch.full((rows, W, Cn), 0xEE, dtype=torch.uint8, device="cuda")
        t[b["halo_top"]:b["halo_top"] + b["row_end"] - b["row_begin"]] = torch.from_numpy(img[b["row_begin"]:b["row_end"]]).cuda()
        bands.append(b); tens.append(t); owned.append(b["row_end"] - b["row_begin"])
    comms = (C.c_void_p * G)()
    devs = (C.c_int * G)(0, 0, 0)
    pkg.check(L.mi_blur_comm_init_p2p(comms, G, devs), "comm_init_p2p")
    ptrs = (C.c_void_p * G)(*[t.data_ptr() for t in tens])
    own = (C.c_int * G)(*owned)
    pkg.check(L.mi_blur_halo_exchange_all(comms, G, ptrs, W, Cn, own, R, None), "halo_exchange_all")
    torch.cuda.synchronize()
    outs = []
    for g in range(G):
        b = bands[g]
        want_band = img[b["row_begin"] - b["halo_top"]: b["row_end"] + b["halo_bottom"]]
        assert np.array_equal(tens[g].cpu().numpy(), want_band), f"shard {g} halo rows wrong"
        out = torch.empty((owned[g], W, Cn), dtype=torch.uint8, device="cuda")
        pkg.check(L.mi_blur_enqueue_band(tens[g].data_ptr(), out.data_ptr(), W, want_band.shape[0], Cn, R,
                                         b["halo_top"], b["halo_top"] + owned[g], None))
        outs.append(out)
    torch.cuda.synchronize()
    assert np.array_equal(np.concatenate([o.cpu().numpy() for o in outs]), O.blur(img, R))
    assert L.mi_blur_halo_exchange(comms[0], ptrs[0], W, Cn, owned[0], R, None) == pkg.ERR_STATE   # needs every rank
    for g in range(G):
        L.mi_blur_comm_destroy(comms[g])


def test_multi_stream_soak(pkg, L, O, torch_cuda):
    """The bench's launch pattern under load: hundreds of batch-35 launches round-robin over 4 streams with sampled
    timestamp events, repeated; afterwards randomly chosen pool images must still equal the oracle (no cross-stream
    interference, no stale tile, no lost launch)."""
    h, w, c, n = 256, 256, 3, 1500
    rng = np.random.default_rng(5)
    with pkg.Context(0, w, h, c, 1, max_batch=1, n_slots=4) as ctx:
        ctx.resident_alloc(n)
        ctx.resident_fill_synthetic(100)
        for rep in range(40):
            ctx.resident_run(n, 35, timed=7)
        tm = ctx.sync()
        assert tm["images"] == 40 * n and tm["launches"] == 40 * 43 and tm["kernel_ms"] > 0
        got = np.zeros((1, h, w, c), np.uint8)
        for idx in [0, 34, 35, n - 1] + [int(x) for x in rng.integers(0, n, 12)]:
            ctx.resident_download(idx, got.ctypes.data, 1)
            assert np.array_equal(got[0], O.blur(O.lcg_stream(1, h, w, c, first_index=100 + idx)[0], 1)), idx


def test_contexts_driven_from_concurrent_host_threads(pkg, L, O, torch_cuda):
    """include/mi_blur.h: a context is single-threaded, different contexts may be driven from different host threads
    at the same time (one feeder thread per GPU in the hosts).  Three threads, three contexts of different shapes and
    radii on the same device, each mixing pageable submits, zero-copy submits and resident passes."""
    import threading
    specs = [(96, 128, 3, 1, 11), (64, 80, 4, 2, 7), (120, 48, 1, 1, 9)]
    errors = []

    def drive(k, h, w, c, r, n):
        try:
            host = O.lcg_stream(n, h, w, c, first_index=1000 * k)
            want = O.blur_batch(host, r)
            nbytes = host.nbytes
            with pkg.Context(0, w, h, c, r, max_batch=n, n_slots=2) as ctx:
                p_in, p_out = L.mi_blur_host_alloc(nbytes), L.mi_blur_host_alloc(nbytes)
                C.memmove(p_in, host.ctypes.data, nbytes)
                ctx.resident_alloc(n)
                ctx.resident_upload(0, host.ctypes.data, n)
                for rep in range(25):
                    out = np.zeros_like(host)
                    ctx.submit(host.ctypes.data, out.ctypes.data, n)            # staged
                    C.memset(p_out, 0, nbytes)
                    ctx.submit(p_in, p_out, n)                                  # zero-copy
                    ctx.resident_run(n, max(1, n // 2), timed=1)
                    ctx.sync()
                    got = np.ctypeslib.as_array((C.c_uint8 * nbytes).from_address(p_out)).reshape(host.shape)
                    res = np.zeros_like(host)
                    ctx.resident_download(0, res.ctypes.data, n)
                    if not (np.array_equal(out, want) and np.array_equal(got, want) and np.array_equal(res, want)):
                        errors.append((k, rep))
                        break
                L.mi_blur_host_free(p_in); L.mi_blur_host_free(p_out)
        except Exception as e:  # noqa: BLE001
            errors.append((k, repr(e)))

    threads = [threading.Thread(target=drive, args=(k,) + s) for k, s in enumerate(specs)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


@pytest.mark.parametrize("c", [1, 2, 3, 4, 5])
def test_layout_repack_matches_reference_loops(pkg, L, O, torch_cuda, c):
    """mi_blur_planar_to_interleaved / _interleaved_to_planar == the reference's host loops
    (heterogeneous_blur.c:125-134, split_image_blur.c:40-56), for the 16-pixel vector path (W*H % 16 == 0, C <= 4), the
    byte path (ragged sizes, C = 5, unaligned pointers), batches, and as a round trip at a BASELINE size."""
    torch = torch_cuda
    rng = np.random.default_rng(40 + c)
    for (h, w, n) in [(4, 4, 1), (16, 16, 3), (240, 320, 5), (7, 9, 2), (1, 1, 4), (33, 48, 2)]:
        planar = rng.integers(0, 256, (n, c, h, w), dtype=np.uint8)
        want = np.stack([O.planar_to_interleaved(np.ascontiguousarray(planar[i])) for i in range(n)])
        d_p = torch.from_numpy(planar).cuda()
        d_i = torch.zeros((n, h, w, c), dtype=torch.uint8, device="cuda")
        pkg.check(L.mi_blur_planar_to_interleaved(d_p.data_ptr(), d_i.data_ptr(), w, h, c, n, None))
        torch.cuda.synchronize()
        assert np.array_equal(d_i.cpu().numpy(), want), (h, w, n)
        d_back = torch.zeros_like(d_p)
        pkg.check(L.mi_blur_interleaved_to_planar(d_i.data_ptr(), d_back.data_ptr(), w, h, c, n, None))
        torch.cuda.synchronize()
        assert np.array_equal(d_back.cpu().numpy(), planar), (h, w, n)
    # unaligned pointers take the byte path
    h, w, n = 16, 16, 2
    planar = rng.integers(0, 256, (n, c, h, w), dtype=np.uint8)
    buf_p = torch.zeros(planar.size + 3, dtype=torch.uint8, device="cuda")
    buf_p[3:] = torch.from_numpy(planar.reshape(-1)).cuda()
    buf_i = torch.zeros(planar.size + 1, dtype=torch.uint8, device="cuda")
    pkg.check(L.mi_blur_planar_to_interleaved(buf_p.data_ptr() + 3, buf_i.data_ptr() + 1, w, h, c, n, None))
    torch.cuda.synchronize()
    want = np.stack([O.planar_to_interleaved(np.ascontiguousarray(planar[i])) for i in range(n)])
    assert np.array_equal(buf_i[1:].cpu().numpy().reshape(n, h, w, c), want)
    # argument checks
    assert L.mi_blur_planar_to_interleaved(buf_p.data_ptr(), buf_p.data_ptr(), w, h, c, n, None) == pkg.ERR_INVALID
    assert L.mi_blur_interleaved_to_planar(buf_p.data_ptr(), buf_i.data_ptr(), 0, h, c, n, None) == pkg.ERR_INVALID


@pytest.mark.parametrize("radius", [1, 2])
def test_submit_planar_frames(pkg, L, O, torch_cuda, radius):
    """mi_blur_submit_planar: frames arrive PLANAR (CImg storage, heterogeneous_blur.c:106-124); the interleave
    (:125-134) and the way back (split_image_blur.c:40-56) are GPU kernels inside the submit.  Pinned caller memory is read
    / written in place by the repack kernels, pageable memory goes through the slot's staging; interleaved and planar
    output; vector and byte repack paths (plane % 16), aligned and ragged blur shapes, several batches in flight."""
    for (n, h, w, c) in [(35, 256, 256, 3), (6, 167, 250, 3), (9, 48, 64, 1), (5, 33, 50, 4), (3, 240, 320, 2)]:
        inter = O.lcg_stream(n, h, w, c, first_index=77)
        planar = np.ascontiguousarray(inter.transpose(0, 3, 1, 2))
        want_i = want_batch(O, inter, radius)
        want_p = np.ascontiguousarray(want_i.transpose(0, 3, 1, 2))
        nbytes = inter.size
        for pinned in (True, False):
            for planar_out in (False, True):
                with pkg.Context(0, w, h, c, radius, max_batch=n, n_slots=2) as ctx:
                    rounds = 3                                          # more submits than slots: slot reuse + wait_oldest
                    if pinned:
                        pin = [L.mi_blur_host_alloc(nbytes) for _ in range(rounds)]
                        pout = [L.mi_blur_host_alloc(nbytes) for _ in range(rounds)]
                        for p_ in pin:
                            C.memmove(p_, planar.ctypes.data, nbytes)
                        for p_ in pout:
                            C.memset(p_, 0xA5, nbytes)
                        for k in range(rounds):
                            ctx.submit_planar(pin[k], pout[k], n, planar_out)
                        tm = ctx.sync()
                        outs = [np.frombuffer(C.string_at(p_, nbytes), np.uint8) for p_ in pout]
                        for p_ in pin + pout:
                            L.mi_blur_host_free(p_)
                    else:
                        outs = [np.full(nbytes, 0xA5, np.uint8) for _ in range(rounds)]
                        for k in range(rounds):
                            ctx.submit_planar(planar.ctypes.data, outs[k].ctypes.data, n, planar_out)
                        tm = ctx.sync()
                    for o in outs:
                        assert np.array_equal(o, (want_p if planar_out else want_i).reshape(-1)), (n, h, w, c, pinned, planar_out)
                    assert tm["images"] == rounds * n and tm["h2d_ms"] > 0 and tm["kernel_ms"] > 0 and tm["d2h_ms"] > 0
    with pkg.Context(0, 64, 64, 3, radius, max_batch=4, n_slots=1) as ctx:
        a = np.zeros(4 * 64 * 64 * 3, np.uint8)
        assert L.mi_blur_submit_planar(ctx.h, a.ctypes.data, a.ctypes.data, 4, 0) == pkg.ERR_INVALID
        assert L.mi_blur_submit_planar(ctx.h, a.ctypes.data, a.ctypes.data + 1, 5, 0) == pkg.ERR_INVALID


def test_planar_frames_blur_as_one_channel_images(pkg, L, O, torch_cuda):
    """A planar (CImg-layout) stream needs no repack to be blurred: it is a stream of n*C one-channel images.
    blur(planar as C=1) repacked == blur(interleaved), on the GPU end to end, 1080p 5x5 and 256x256 3x3."""
    torch = torch_cuda
    for (h, w, c, n, radius) in [(256, 256, 3, 6, 1), (1080, 1920, 3, 2, 2)]:
        inter = O.lcg_stream(n, h, w, c)
        d_i = torch.from_numpy(inter).cuda()
        d_p = torch.empty((n, c, h, w), dtype=torch.uint8, device="cuda")
        pkg.check(L.mi_blur_interleaved_to_planar(d_i.data_ptr(), d_p.data_ptr(), w, h, c, n, None))
        d_pb = torch.empty_like(d_p)
        pkg.check(L.mi_blur_enqueue(d_p.data_ptr(), d_pb.data_ptr(), w, h, 1, radius, n * c, None))      # planes as images
        d_ib = torch.empty_like(d_i)
        pkg.check(L.mi_blur_planar_to_interleaved(d_pb.data_ptr(), d_ib.data_ptr(), w, h, c, n, None))
        d_direct = torch.empty_like(d_i)
        pkg.check(L.mi_blur_enqueue(d_i.data_ptr(), d_direct.data_ptr(), w, h, c, radius, n, None))
        torch.cuda.synchronize()
        assert bool((d_ib == d_direct).all())
        assert np.array_equal(d_direct[0].cpu().numpy(), O.blur(np.ascontiguousarray(inter[0]), radius))


@pytest.mark.parametrize("W,H", [(26752, 26757), (26750, 26759)])
def test_single_image_near_the_2gib_limit(pkg, L, O, torch_cuda, W, H):
    """The largest frame the interface takes is just under 2^31 bytes (32-bit offsets inside an image, like the
    reference's `int idx`): 26752 x 26757 x 3 (aligned rows) and 26750 x 26759 x 3 (ragged rows, pitch 80250), both
    within 75 kB of the limit.  Sampled row bands (top, around the 2^30-byte offset, middle, bottom) of the output equal
    the oracle on the same rows; one more row is refused."""
    torch = torch_cuda
    c = 3
    assert W * H * c < 2**31 <= W * (H + 1) * c
    g = torch.Generator(device="cuda"); g.manual_seed(11)
    d_in = torch.randint(0, 256, (H, W, c), dtype=torch.uint8, device="cuda", generator=g)
    d_out = torch.empty_like(d_in)
    for radius in (1, 2):
        pkg.check(L.mi_blur_enqueue_ex(d_in.data_ptr(), d_out.data_ptr(), W, H, c, radius, 1, 0, H, pkg.VARIANT_TILED, None))
        torch.cuda.synchronize()
        for r0 in (0, 13376, H // 2 + 5, H - 8):         # 13380 rows * ~80 kB is just past 2^30
            lo, hi = max(r0 - radius, 0), min(r0 + 8 + radius, H)
            band = np.ascontiguousarray(d_in[lo:hi].cpu().numpy())
            ref = O.blur(band, radius)                     # exact for rows whose window lies inside the band; where the
            a = r0 - lo                                    # band touches the image edge the clamp is the image's own
            assert np.array_equal(d_out[r0:r0 + 8].cpu().numpy(), ref[a:a + 8]), (radius, r0)
    d_big = torch.empty(8, dtype=torch.uint8, device="cuda")   # never touched: the call must be refused up front
    assert L.mi_blur_enqueue_ex(d_big.data_ptr(), d_out.data_ptr(), W, H + 1, c, 1, 1, 0, H + 1, 0, None) == pkg.ERR_INVALID


def test_fused_stream_parity_and_batch_flags(pkg, L, O, torch_cuda):
    """mi_blur_resident_run_fused: one dispatch for the pass, batch = unit of completion.  Every image equals the oracle
    (3x3 and 5x5, full and short last batch, repeated passes = epochs); the done-count is monotone, never exceeds the
    number of batches, reaches it after sync, and a batch reported done has its outputs in the pool."""
    for (h, w, c, r, n, batch) in [(64, 64, 3, 1, 250, 35), (48, 80, 4, 2, 77, 10), (256, 256, 3, 1, 1500, 35), (32, 16, 1, 1, 9, 20)]:
        with pkg.Context(0, w, h, c, r, max_batch=1, n_slots=2) as ctx:
            ctx.resident_alloc(n)
            ctx.resident_fill_synthetic(7)
            nb = (n + batch - 1) // batch
            src = O.lcg_stream(n, h, w, c, first_index=7)
            want = O.blur_batch(src, r)
            for rep in range(3):
                ctx.resident_run_fused(n, batch, timed=(rep == 1))
                seen = []
                for _ in range(50):
                    seen.append(ctx.resident_batches_done())
                tm = ctx.sync()
                seen.append(ctx.resident_batches_done())
                assert all(0 <= a <= b2 <= nb for a, b2 in zip(seen, seen[1:])), seen
                assert seen[-1] == nb
                out = np.zeros_like(src)
                ctx.resident_download(0, out.ctypes.data, n)
                assert np.array_equal(out, want), (h, w, c, r, rep)
            assert tm["launches"] == 3 and tm["images"] == 3 * n and tm["kernel_ms"] > 0
            assert ctx.timed_coverage() == (1, 2 * n * h * w * c)
    # the headline shape: poll until the first batches are flagged (this returns while the dispatch is typically still
    # running), then read batch 0 back (resident_download waits for the device) and compare
    h, w, c, n, batch = 256, 256, 3, 5000, 35
    with pkg.Context(0, w, h, c, 1, max_batch=1, n_slots=2) as ctx:
        ctx.resident_alloc(n)
        ctx.resident_fill_synthetic(0)
        ctx.resident_run_fused(n, batch)
        assert ctx.wait_batches(3, timeout_s=30.0) >= 3       # bounded poll; a failed poll raises instead of reading as 0
        got = np.zeros((batch, h, w, c), np.uint8)
        ctx.resident_download(0, got.ctypes.data, batch)
        ctx.sync()
        assert np.array_equal(got, O.blur_batch(O.lcg_stream(batch, h, w, c), 1))
    # rows shorter than one 16-byte chunk have no tile form at all, so no fused form either (longer ragged rows do: see
    # test_fused_stream_ragged_rows)
    with pkg.Context(0, 5, 9, 3, 1, max_batch=1, n_slots=1) as ctx:
        ctx.resident_alloc(4)
        assert L.mi_blur_resident_run_fused(ctx.h, 4, 2, 0) == pkg.ERR_UNSUPPORTED


def test_fused_stream_dynamic_tail(pkg, L, O, torch_cuda):
    """Big fused passes hand their last tiles out dynamically (blur_fused_tail_kernel: extra workgroups at the end of the grid
    draw tickets, so an XCD that runs ahead takes more of the tail).  Every image equals the oracle, every batch is counted
    exactly once per pass — over repeated passes (the ticket counter resets itself), tails of 3 % / 20 % / 50 % of the pass, few
    and many spare workgroups, the release-ordered count, a watched pass; passes below 8192 tiles keep the static kernel."""
    for (h, w, c, r, n, batch) in [(64, 64, 3, 1, 9000, 35), (32, 48, 4, 2, 8200, 100)]:
        src = O.lcg_stream(n, h, w, c, first_index=11)
        want = O.blur_batch(src, r)
        nb = (n + batch - 1) // batch
        for tail, spare, release in ((30, 25, 0), (200, 10, 0), (500, 100, 0), (30, 25, 1), (0, 25, 0)):
            pkg.check(L.mi_blur_set_option(b"fused_tail", tail))
            pkg.check(L.mi_blur_set_option(b"fused_tail_blocks", spare))
            pkg.check(L.mi_blur_set_option(b"fused_release", release))
            try:
                with pkg.Context(0, w, h, c, r, max_batch=1, n_slots=1) as ctx:
                    ctx.resident_alloc(n)
                    ctx.resident_fill_synthetic(11)
                    for rep in range(3):
                        ctx.resident_run_fused(n, batch, watch=(rep == 2))
                        assert L.mi_blur_last_kernel() == (b"blur_fused_tail_kernel" if tail else b"blur_fused_kernel")
                        ctx.sync()
                        assert ctx.resident_batches_done() == nb, (h, w, tail, spare, rep)
                        out = np.zeros_like(src)
                        ctx.resident_download(0, out.ctypes.data, n)
                        assert np.array_equal(out, want), (h, w, c, r, tail, spare, release, rep)
                    # a shorter pass on the same context (another geometry, below the threshold: static kernel) and back
                    ctx.resident_run_fused(1000, batch)
                    assert L.mi_blur_last_kernel() == b"blur_fused_kernel"
                    ctx.sync()
                    assert ctx.resident_batches_done() == (1000 + batch - 1) // batch
                    ctx.resident_run_fused(n, batch)
                    ctx.sync()
                    assert ctx.resident_batches_done() == nb
                    # a WATCHED pass with passes queued straight behind it — the same geometry (the counters count on under the
                    # watcher) and another one (the counters are zeroed): the watcher ends with its own pass either way
                    t0 = time.perf_counter()
                    ctx.resident_run_fused(n, batch, watch=True)
                    ctx.resident_run_fused(n, batch)
                    ctx.resident_run_fused(n, batch, watch=True)
                    ctx.resident_run_fused(n // 2, batch + 1)
                    ctx.sync()
                    assert time.perf_counter() - t0 < 5.0, "a watcher outlived its pass (it would sit out its 10 s hard limit)"
                    assert ctx.resident_batches_done() == (n // 2 + batch) // (batch + 1)
            finally:
                pkg.check(L.mi_blur_set_option(b"fused_tail", 30))
                pkg.check(L.mi_blur_set_option(b"fused_tail_blocks", 25))
                pkg.check(L.mi_blur_set_option(b"fused_release", 0))


def test_fused_stream_big_batches(pkg, L, O, torch_cuda):
    """A batch of thousands of tiles spreads its completion count over more words than a small one (8 .. 256 per batch: an add
    on a hot word costs ~200 ns of that word's time — with eight words a 1080p pass in batches of 35 ran 223 us instead of 138).
    Every image equals the oracle and every batch is reported exactly when complete, through the counter read-back and through
    a watcher, over repeated passes and across changes of batch size (= of the counter layout) on one context."""
    h, w, c, r, n = 64, 64, 3, 1, 9000                        # one tile per image
    src = O.lcg_stream(n, h, w, c, first_index=21)
    want = O.blur_batch(src, r)
    with pkg.Context(0, w, h, c, r, max_batch=1, n_slots=1) as ctx:
        ctx.resident_alloc(n)
        ctx.resident_fill_synthetic(21)
        for batch in (35, 4000, 600, 9000, 35, 1100):             # 8, 64, 16, 256, 8, 32 counters per batch
            nb = (n + batch - 1) // batch
            for rep in range(3):
                ctx.resident_run_fused(n, batch, watch=(rep == 1))
                seen = [ctx.resident_batches_done() for _ in range(20)]
                ctx.sync()
                seen.append(ctx.resident_batches_done())
                assert all(0 <= a <= b2 <= nb for a, b2 in zip(seen, seen[1:])) and seen[-1] == nb, (batch, rep, seen)
            out = np.zeros_like(src)
            ctx.resident_download(0, out.ctypes.data, n)
            assert np.array_equal(out, want), batch
        # a short pass after a long one: fewer batches, other layout; the count of the short pass is its own
        ctx.resident_run_fused(700, 300)
        ctx.sync()
        assert ctx.resident_batches_done() == 3


def test_fused_stream_ragged_rows(pkg, L, O, torch_cuda):
    """The fused stream on frames whose rows are not a multiple of 16 bytes (250x250x3, 1366-wide, odd little shapes): the ragged
    form of the tile code with its stores — whole chunks and the 8/4/2/1-byte pieces of a row's last chunk — written through L2
    like the aligned form's.  Every image equals the oracle, batches are counted exactly, short passes (static map) and long
    ones (dynamic tail), 1-4 channels, both kernel sizes, repeated passes, a watched pass, the release-ordered count."""
    rng = np.random.default_rng(404)
    cases = [(250, 250, 3, 1, 300, 35), (61, 37, 3, 2, 120, 7), (30, 1366, 3, 1, 12, 5), (17, 33, 1, 1, 40, 40), (9, 17, 3, 2, 9, 2),
             (64, 50, 3, 1, 9000, 35), (40, 30, 4, 2, 8500, 500), (33, 21, 2, 1, 64, 9)]
    for case, (h, w, c, r, n, batch) in enumerate(cases):
        assert (w * c) % 16 != 0
        src = O.lcg_stream(n, h, w, c, first_index=case)
        want = O.blur_batch(src, r)
        nb = (n + batch - 1) // batch
        with pkg.Context(0, w, h, c, r, max_batch=1, n_slots=1) as ctx:
            ctx.resident_alloc(n)
            ctx.resident_fill_synthetic(case)
            for rep in range(3):
                pkg.check(L.mi_blur_set_option(b"fused_release", 1 if rep == 2 else 0))
                try:
                    ctx.resident_run_fused(n, batch, watch=(rep == 1))
                    assert L.mi_blur_last_kernel() == b"blur_fused_tail_kernel"
                    ctx.sync()
                finally:
                    pkg.check(L.mi_blur_set_option(b"fused_release", 0))
                assert ctx.resident_batches_done() == nb, (case, rep)
                out = np.zeros_like(src)
                ctx.resident_download(0, out.ctypes.data, n)
                assert np.array_equal(out, want), (case, h, w, c, r, n, batch, rep)
            # batches readable as soon as counted, on the long cases: poll, then peek the last image of the reported batches
            if n >= 8000:
                ctx.resident_run_fused(n, batch)
                done = ctx.wait_batches(max(1, nb // 4), timeout_s=30.0)
                last = min(done * batch, n) - 1
                got = np.zeros((1, h, w, c), np.uint8)
                ctx.resident_peek(last, got.ctypes.data, 1)
                ctx.sync()
                assert np.array_equal(got[0], want[last]), (case, done)


def test_fused_stream_random_shapes(pkg, L, O, torch_cuda):
    """Seeded sweep of the fused stream: aligned shapes, C = 1..4, both radii, batch sizes that do and do not divide the
    stream, one or several strips per row, fewer tiles per batch than XCDs and many more."""
    rng = np.random.default_rng(77)
    for case in range(24):
        c = int(rng.choice([1, 2, 3, 4]))
        w = int(rng.integers(1, 120)) * 16 // int(np.gcd(16, c))
        h = int(rng.integers(1, 90))
        r = int(rng.choice([1, 2]))
        n = int(rng.integers(1, 60))
        batch = int(rng.integers(1, n + 3))
        with pkg.Context(0, w, h, c, r, max_batch=1, n_slots=1) as ctx:
            ctx.resident_alloc(n)
            ctx.resident_fill_synthetic(case)
            ctx.resident_run_fused(n, batch)
            ctx.sync()
            assert ctx.resident_batches_done() == (n + batch - 1) // batch, (case, h, w, c, r, n, batch)
            out = np.zeros((n, h, w, c), np.uint8)
            ctx.resident_download(0, out.ctypes.data, n)
            assert np.array_equal(out, O.blur_batch(O.lcg_stream(n, h, w, c, first_index=case), r)), (case, h, w, c, r, n, batch)


@pytest.mark.parametrize("watch", [False, True])
def test_fused_stream_batches_are_readable_as_soon_as_counted(pkg, L, O, torch_cuda, watch):
    """watch = True: the pass is WATCHED — a one-wave kernel keeps the number of leading complete batches in pinned host
    memory, so the poll is a read of the host's own memory (no copy, no HIP call); same property, many more samples.
    The point of the per-batch counters: while ONE dispatch is still working through a 40 000-image stream, every batch
    the poll reports done must already hold its final bytes when read on another stream (mi_blur_resident_peek does not
    wait for the dispatch).  The output pool is poisoned first (0xEE everywhere: the blur of a constant image), so a batch
    counted in too early would read as poison."""
    torch = torch_cuda
    h, w, c, n, batch = 256, 256, 3, 40000, 35
    isz = h * w * c
    nb = (n + batch - 1) // batch
    with pkg.Context(0, w, h, c, 1, max_batch=1, n_slots=1) as ctx:
        ctx.resident_alloc(n)
        ctx.resident_fill_synthetic(0)
        ctx.resident_run_fused(70, batch); ctx.sync()                       # creates the counters and the poll stream
        out_ptr = L.mi_blur_resident_out(ctx.h)
        const_in = torch.full((1000, h, w, c), 0xEE, dtype=torch.uint8, device="cuda")
        for i in range(0, n, 1000):
            pkg.check(L.mi_blur_enqueue(const_in.data_ptr(), out_ptr + i * isz, w, h, c, 1, min(1000, n - i), None))
        torch.cuda.synchronize()
        got = np.zeros((batch, h, w, c), np.uint8)
        ctx.resident_peek(n - batch, got.ctypes.data, batch)
        assert (got == 0xEE).all()
        ctx.resident_run_fused(n, batch, watch=watch)
        # capture first (fast: one image per newly reported batch — its LAST image), verify after the dispatch has ended
        captured, last, in_flight = [], 0, 0
        deadline = time.monotonic() + 60.0
        while last < nb:
            assert time.monotonic() < deadline, f"fused stream stuck: {last} of {nb} batches counted in after 60 s"
            done = ctx.resident_batches_done()                                 # raises on a negative status
            if done > last:
                b = done - 1                                                   # the newest batch reported done
                idx = min((b + 1) * batch, n) - 1
                one = np.zeros((1, h, w, c), np.uint8)
                ctx.resident_peek(idx, one.ctypes.data, 1)
                captured.append((b, idx, one))
                in_flight += done < nb
                last = done
        ctx.sync()
        assert captured and last == nb
        for b, idx, one in captured:
            want = O.blur(O.lcg_stream(1, h, w, c, first_index=idx)[0], 1)
            assert np.array_equal(one[0], want), f"batch {b} reported done but image {idx} not final"
        # The test only means something if samples were taken WHILE the dispatch was running (a 40 000-image pass lasts
        # ~3 ms, a poll + peek ~50 us: tens of samples).  A host so slow that every sample came after the end would make
        # this pass vacuously, so that is a failure, not a pass.
        print(f"watch={watch}: verified {len(captured)} batches, {in_flight} of them read while the dispatch was still running")
        assert in_flight > 0, "no sample was taken mid-dispatch: the early-readability property was not exercised"
        # a second watched pass right behind an unwatched one, and the count after the end
        ctx.resident_run_fused(700, batch)
        ctx.resident_run_fused(700, batch, watch=True)
        assert ctx.wait_batches(20) == 20
        ctx.sync()
        assert ctx.resident_batches_done() == 20


def test_fused_stream_release_mode_and_geometry_change(pkg, L, O, torch_cuda):
    """(0) Windows of 1, 3, 8 and more batches than the stream has: same pixels, every batch counted in.
    (1) "fused_release" 1 — the architectural (release-ordered) completion add — gives the same pixels and counts.
    (2) Same-shape passes keep the counters counting up; a knob that changes the launch geometry between two such passes
    (rows per thread 4 -> 8: half as many blocks per batch) must zero them instead of leaving the count unreachable."""
    h, w, c, r, n, batch = 64, 128, 3, 1, 120, 16
    nb = (n + batch - 1) // batch
    want = O.blur_batch(O.lcg_stream(n, h, w, c, first_index=3), r)
    try:
        with pkg.Context(0, w, h, c, r, max_batch=1, n_slots=1) as ctx:
            ctx.resident_alloc(n)
            ctx.resident_fill_synthetic(3)
            for rpt, rel, win in ((4, 0, 8), (4, 0, 8), (8, 0, 8), (8, 1, 1), (4, 1, 3), (0, 0, 1), (0, 0, 100)):
                pkg.check(L.mi_blur_set_option(b"rows_per_thread", rpt))
                pkg.check(L.mi_blur_set_option(b"fused_release", rel))
                pkg.check(L.mi_blur_set_option(b"fused_window", win))
                ctx.resident_run_fused(n, batch)
                assert ctx.wait_batches(nb, timeout_s=30.0) == nb, (rpt, rel, win)
                ctx.sync()
                out = np.zeros_like(want)
                ctx.resident_download(0, out.ctypes.data, n)
                assert np.array_equal(out, want), (rpt, rel, win)
    finally:
        pkg.check(L.mi_blur_set_option(b"rows_per_thread", 0))
        pkg.check(L.mi_blur_set_option(b"fused_release", 0))
        pkg.check(L.mi_blur_set_option(b"fused_window", 8))


def test_batches_done_reports_errors_as_negative_status(pkg, L):
    """A poll that cannot be answered is a negative status, never 0 ("none done yet")."""
    assert L.mi_blur_resident_batches_done(None) == pkg.ERR_INVALID
    with pkg.Context(0, 64, 64, 3, 1, max_batch=1, n_slots=1) as ctx:
        assert ctx.resident_batches_done() == 0          # no fused pass yet: a valid count


def test_headline_stream_golden_hash(pkg, L, O, torch_cuda, golden):
    """BASELINE configs[1] at full size against the reference kernel itself: tests/golden "stream" holds the FNV-1a-64 of
    all 983 040 000 output bytes of the 5000 x 256x256x3 synthetic stream as gaussian_kernel.cl produced them (and the
    digest of the 5000 per-image hashes).  Both dispatch forms of the resident stream — one launch per batch of 35 and
    the fused one-dispatch pass the bench line quotes — must reproduce it byte for byte."""
    e = golden["stream"]
    n, h, w, c = e["n"], e["h"], e["w"], e["c"]
    isz = h * w * c
    out = np.empty((n, h, w, c), np.uint8)

    def digest():
        whole = f"{L.mi_blur_fnv1a64(out.ctypes.data, out.size):016x}"
        per = np.array([L.mi_blur_fnv1a64(out.ctypes.data + i * isz, isz) for i in range(n)], dtype="<u8")
        return whole, f"{L.mi_blur_fnv1a64(per.ctypes.data, per.nbytes):016x}", per

    with pkg.Context(0, w, h, c, e["radius"], max_batch=1, n_slots=4) as ctx:
        ctx.resident_alloc(n)
        ctx.resident_fill_synthetic(e["first_index"])
        for form in ("per_batch", "fused", "fused_again"):
            out[:] = 0
            if form == "per_batch":
                ctx.resident_run(n, 35)
            else:
                ctx.resident_run_fused(n, 35)
                assert ctx.wait_batches((n + 34) // 35, timeout_s=30.0) == (n + 34) // 35
            ctx.sync()
            ctx.resident_download(0, out.ctypes.data, n)
            whole, per_digest, per = digest()
            for k, v in e["image_fnv"].items():
                assert f"{int(per[int(k)]):016x}" == v, (form, k)
            assert per_digest == e["per_image_fnv_digest"], form
            assert whole == e["out_fnv"], form


def test_debug_xcd_times_diagnostics(pkg, L, O, torch_cuda):
    """Developer diagnostics: with "debug_xcd_times" on, the tiled kernel's workgroups leave per-XCD start/end times; the
    output bytes are unchanged, every XCD that ran a workgroup reports end > start, and re-arming clears the slots."""
    host = O.lcg_stream(40, 256, 256, 3)
    want = O.blur_batch(host, 1)
    end, beg = (C.c_uint64 * 8)(), (C.c_uint64 * 8)()
    try:
        pkg.check(L.mi_blur_set_option(b"debug_xcd_times", 1))
        pkg.check(L.mi_blur_debug_xcd_times(end, beg, 1))
        got = gpu_blur(pkg, L, torch_cuda, host, 1, pkg.VARIANT_TILED)
        assert np.array_equal(got, want)
        pkg.check(L.mi_blur_debug_xcd_times(end, beg, 1))
        ran = [i for i in range(8) if end[i]]
        assert len(ran) == 8, "320 workgroups are dealt over all 8 XCDs"
        assert all(end[i] > beg[i] for i in ran)
        assert (max(end) - min(beg)) / 100.0 < 5000.0               # 100 MHz ticks: the launch took well under 5 ms
        pkg.check(L.mi_blur_debug_xcd_times(end, beg, 0))
        assert not any(end)                                          # re-armed: nothing since
    finally:
        pkg.check(L.mi_blur_set_option(b"debug_xcd_times", 0))


def test_random_shapes_auto_dispatch(pkg, L, O, torch_cuda):
    """Seeded sweep through the dispatcher (aligned tiled / ragged tiled / generic, chosen by shape): any width, height,
    1-5 channels, both radii, batches, output row ranges inside a band (Approach-2 form: clamp at the band's own edges,
    only rows [y0, y1) produced) — every case byte-equal to the oracle."""
    rng = np.random.default_rng(2025)
    try:
        for case in range(60):
            c = int(rng.choice([1, 2, 3, 3, 3, 4, 5]))
            w = int(rng.choice([rng.integers(1, 24), rng.integers(16, 400), 16 * int(rng.integers(1, 40)) // int(np.gcd(16, c)) or 16]))
            h = int(rng.integers(1, 70))
            r = int(rng.choice([1, 2]))
            n = int(rng.integers(1, 5))
            y0 = int(rng.integers(0, h))
            y1 = int(rng.integers(y0 + 1, h + 1))
            host = rng.integers(0, 256, (n, h, w, c), dtype=np.uint8)
            want = want_batch(O, host, r)[:, y0:y1]
            for opts in ({}, {"rows_per_thread": int(rng.choice([4, 8, 16])), "stage_dma": int(rng.integers(0, 2)), "xcd_run": int(rng.choice([0, 3]))}):
                got = gpu_blur(pkg, L, torch_cuda, host, r, pkg.VARIANT_AUTO, y0=y0, y1=y1, opts=opts)
                assert np.array_equal(got, want), (case, h, w, c, r, n, y0, y1, opts)
    finally:
        reset_opts(L)
