#!/usr/bin/env python3
"""Generate tests/golden/* from the UNMODIFIED reference kernel (oracle/_ref).

Run in the build container only (needs /root/reference and `make -C oracle ref`):

    python tests/golden/make_golden.py

Writes
  blur_golden.json   known answers of gaussian_kernel.cl:19-72 under the NDRange
                     harness (oracle/ref_harness.cpp): FNV-1a-64 of input/output and
                     the first output bytes for LCG-filled images, small literal
                     vectors, and Approach-2 split outputs (split_image_blur.c:511-541).
                     The "k5" section holds 5x5 hashes from the oracle itself — the
                     reference has no 5x5 kernel, so that section is "parity unpinned".
  ref_images.npz     PIL-decoded pixels of the reference's sample input
                     image_320x240.jpg and of split_output.jpg (the one saved output the
                     reference ships, split_image_blur.c:548-553) — data, for a loose
                     PSNR sanity check; JPEG-lossy, never bit-exact.
  "stream" section   the WHOLE headline stream (5000 x 256x256x3, image i = LCG seeded 0x9E3779B9 ^ i)
                     through the reference kernel: FNV-1a-64 of the 983 040 000 output bytes laid end to end, and the
                     FNV of the 5000 per-image output hashes (little-endian u64 each) — what the GPU tests assert for
                     the per-batch-launch and the fused forms of the resident stream.
  "bands8192" section  BASELINE configs[4] (one 8192x8192x3 image row-split over G GPUs): FNV-1a-64 of the reference
                     kernel's output rows [H*g/G, H*(g+1)/G) for G in {1, 2, 3, 4, 6, 8} — what rank g of `bench.py --gpus G`
                     must hold after halo exchange + band blur (split_image_blur.c:511-541 semantics, K-way).
  stream50k_image_fnv.npy + "stream50k" section
                     BASELINE configs[3] (50 000 x 256x256x3, image i = LCG seeded 0x9E3779B9 ^ i) through the reference
                     kernel: the FNV-1a-64 of EVERY output image (little-endian u64 x 50 000 = 400 KB), so that any rank of
                     any N can check every image of its shard; the JSON section holds the digest of that file's payload
                     and of each rank's slice for N in {1, 2, 4, 8}.
Fixtures are data (inputs/expected outputs); no reference source is stored.

    python tests/golden/make_golden.py --stream-only     # recompute only the "stream" section (keeps the rest)
    python tests/golden/make_golden.py --multi-gpu-only  # recompute only "bands8192" / "stream50k" (keeps the rest)
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def hx(v: int) -> str:
    return f"{v:016x}"


def stream_section(n: int = 5000, h: int = 256, w: int = 256, c: int = 3, threads: int = 8) -> dict:
    """The headline stream through the UNMODIFIED reference kernel, image by image on `threads` host threads."""
    import threading
    rlib = O.ref()
    src = O.lcg_stream(n, h, w, c)
    out = np.empty_like(src)
    isz = h * w * c

    def work(b, e):
        for i in range(b, e):
            rlib.ref_gaussian_blur(src.ctypes.data + i * isz, out.ctypes.data + i * isz, w, h, c)

    th = [threading.Thread(target=work, args=(n * t // threads, n * (t + 1) // threads)) for t in range(threads)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for i in (0, 1, 2499, 4999):          # the restatement agrees on a sample
        assert np.array_equal(out[i], O.blur(src[i], 1)), i
    per_image = np.array([O.fnv1a64(out[i]) for i in range(n)], dtype="<u8")
    return {"n": n, "h": h, "w": w, "c": c, "radius": 1, "first_index": 0,
            "in_fnv": hx(O.fnv1a64(src)), "out_fnv": hx(O.fnv1a64(out)),
            "per_image_fnv_digest": hx(O.fnv1a64(per_image.view(np.uint8))),
            "image_fnv": {str(i): hx(int(per_image[i])) for i in (0, 34, 35, 2499, 4969, 4999)}}


def bands8192_section(h: int = 8192, w: int = 8192, c: int = 3) -> dict:
    """configs[4]: per-band hashes of the reference kernel's output of the one LCG image (seed index 0)."""
    img = O.lcg_image(h, w, c)
    out = O.ref_blur(img)
    sec = {"h": h, "w": w, "c": c, "radius": 1, "in_fnv": hx(O.fnv1a64(img)), "out_fnv": hx(O.fnv1a64(out)), "bands": {}}
    for G in (1, 2, 3, 4, 6, 8):
        sec["bands"][str(G)] = [hx(O.fnv1a64(out[h * g // G:h * (g + 1) // G])) for g in range(G)]
    return sec


def stream50k_section(n: int = 50000, h: int = 256, w: int = 256, c: int = 3, threads: int = 8, chunk: int = 1000) -> dict:
    """configs[3]: every image of the 50 000-image stream through the UNMODIFIED reference kernel; per-image FNVs."""
    import threading
    rlib = O.ref()
    isz = h * w * c
    per_image = np.empty(n, dtype="<u8")
    for first in range(0, n, chunk):
        m = min(chunk, n - first)
        src = O.lcg_stream(m, h, w, c, first_index=first)
        out = np.empty_like(src)

        def work(b, e):
            for i in range(b, e):
                rlib.ref_gaussian_blur(src.ctypes.data + i * isz, out.ctypes.data + i * isz, w, h, c)
                per_image[first + i] = O.fnv1a64(out[i])

        th = [threading.Thread(target=work, args=(m * t // threads, m * (t + 1) // threads)) for t in range(threads)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert np.array_equal(out[m - 1], O.blur(src[m - 1], 1)), first      # the restatement agrees on a sample
        print(f"stream50k: {first + m} / {n}", flush=True)
    np.save(os.path.join(HERE, "stream50k_image_fnv.npy"), per_image)
    sec = {"n": n, "h": h, "w": w, "c": c, "radius": 1, "file": "stream50k_image_fnv.npy",
           "per_image_fnv_digest": hx(O.fnv1a64(per_image.view(np.uint8))), "rank_digest": {}}
    for G in (1, 2, 4, 8):
        sec["rank_digest"][str(G)] = [hx(O.fnv1a64(per_image[n * g // G:n * (g + 1) // G].view(np.uint8))) for g in range(G)]
    return sec


def main() -> None:
    O.build(ref=True)
    if "--multi-gpu-only" in sys.argv:
        path = os.path.join(HERE, "blur_golden.json")
        gold = json.load(open(path))
        gold["bands8192"] = bands8192_section()
        print("bands8192:", gold["bands8192"]["out_fnv"], flush=True)
        gold["stream50k"] = stream50k_section()
        with open(path, "w") as f:
            json.dump(gold, f, indent=1)
        return
    if "--stream-only" in sys.argv:
        path = os.path.join(HERE, "blur_golden.json")
        gold = json.load(open(path))
        gold["stream"] = stream_section()
        with open(path, "w") as f:
            json.dump(gold, f, indent=1)
        print("stream:", gold["stream"])
        return
    gold = {"generator": "tests/golden/make_golden.py", "source": "oracle/_ref (unmodified gaussian_kernel.cl)",
            "lcg": {"a": 1664525, "c": 1013904223, "seed": O.LCG_SEED, "byte": "s>>24"},
            "hash": "FNV-1a-64", "k3": [], "literals": [], "a2_split": [], "k5_unpinned": []}

    shapes = [(256, 256, 3), (240, 320, 3), (1080, 1920, 3), (16, 16, 3), (33, 17, 3), (3, 5, 3), (1, 1, 3),
              (64, 64, 1), (33, 17, 1), (64, 64, 4), (31, 29, 4), (48, 64, 2), (1, 64, 3), (64, 1, 3), (2, 2, 3),
              (8192, 8192, 3)]
    for (h, w, c) in shapes:
        img = O.lcg_image(h, w, c)
        out = O.ref_blur(img)
        assert np.array_equal(out, O.blur(img, 1)), f"oracle != reference at {h}x{w}x{c}"
        gold["k3"].append({"h": h, "w": w, "c": c, "in_fnv": hx(O.fnv1a64(img)), "out_fnv": hx(O.fnv1a64(out)),
                           "first": out.reshape(-1)[:8].tolist()})
        print(f"k3 {w}x{h}x{c}: {gold['k3'][-1]['out_fnv']}", flush=True)

    lits = [
        ("impulse3x3", (3, 3, 1), [0, 0, 0, 0, 255, 0, 0, 0, 0]),
        ("ramp3x3", (3, 3, 1), list(range(1, 10))),
        ("rgb2x2", (2, 2, 3), [10, 20, 30, 40, 50, 60, 250, 251, 252, 253, 254, 255]),
        ("all255_4x4", (4, 4, 3), [255] * 48),
        ("all0_4x4", (4, 4, 3), [0] * 48),
        ("corner_impulses_5x4", (4, 5, 1), [255, 0, 0, 0, 255, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 255, 0, 0, 0, 255]),
        ("row1x8", (1, 8, 1), [0, 255, 0, 0, 17, 34, 51, 255]),
        ("col8x1", (8, 1, 1), [0, 255, 0, 0, 17, 34, 51, 255]),
    ]
    for name, (h, w, c), vals in lits:
        img = np.array(vals, np.uint8).reshape(h, w, c)
        out = O.ref_blur(img)
        gold["literals"].append({"name": name, "h": h, "w": w, "c": c, "in": vals, "out": out.reshape(-1).tolist()})

    for (h, w, c, split) in [(240, 320, 3, 39), (240, 320, 3, 120), (256, 256, 3, 1), (256, 256, 3, 255), (33, 17, 3, 16)]:
        img = O.lcg_image(h, w, c)
        out = O.ref_split_blur(img, split)
        whole = O.ref_blur(img)
        gold["a2_split"].append({"h": h, "w": w, "c": c, "split_row": split, "out_fnv": hx(O.fnv1a64(out)),
                                 "equals_whole": bool(np.array_equal(out, whole))})

    for (h, w, c) in [(256, 256, 3), (1080, 1920, 3), (33, 17, 3), (3, 5, 3), (64, 64, 4), (64, 64, 1)]:
        img = O.lcg_image(h, w, c)
        out = O.blur(img, 2)
        assert np.array_equal(out, O.blur_f32(img, 2)) and np.array_equal(out, O.np_blur(img, 2))
        gold["k5_unpinned"].append({"h": h, "w": w, "c": c, "out_fnv": hx(O.fnv1a64(out)),
                                    "first": out.reshape(-1)[:8].tolist()})

    gold["stream"] = stream_section()
    gold["bands8192"] = bands8192_section()
    gold["stream50k"] = stream50k_section()

    with open(os.path.join(HERE, "blur_golden.json"), "w") as f:
        json.dump(gold, f, indent=1)

    from PIL import Image
    src = np.asarray(Image.open(os.path.join(O.REFERENCE_ROOT, "image_320x240.jpg")).convert("RGB"))
    saved = np.asarray(Image.open(os.path.join(O.REFERENCE_ROOT, "split_output.jpg")).convert("RGB"))
    np.savez_compressed(os.path.join(HERE, "ref_images.npz"), image_320x240=src, split_output=saved)
    print("wrote", os.listdir(HERE))


if __name__ == "__main__":
    main()
