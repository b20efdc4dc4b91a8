#!/usr/bin/env python3
"""Static instruction counts of the hot kernels from the gfx950 assembly (hipcc -S): VALU / LDS / VMEM per wave and VALU
per output dword (the unrolled row loop is straight-line code; a thread writes rows_per_thread x 4 output dwords).
    python tools/valu_count.py > profiles/<tag>_valu_counts.txt          (build container, no GPU needed)"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "heterogeneous-opencl-image-processing-engine_amd", "csrc", "blur_kernels.hip")


def main():
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", out, SRC],
                       check=True, stderr=subprocess.DEVNULL)
        text = open(out).read().splitlines()
    want = {
        "3x3 tiled, rows/thread 8 (split-then-shift row pass: shipped)": "_ZN7mi_blur17blur_tiled_kernelILi3ELi1ELi8ELb1ELb0ELb0EEEvNS_11TiledParamsE",
        "3x3 tiled, rows/thread 8 (raw-window row pass: A/B form)": "_ZN7mi_blur19blur_tiled_x_kernelILi3ELi1ELi8ELi1EEEvNS_11TiledParamsE",
        "3x3 fused stream, rows/thread 8": "_ZN7mi_blur17blur_fused_kernelILi3ELi1ELi8EEEvNS_11TiledParamsENS_11FusedParamsE",
        "3x3 fused stream with dynamic tail, rows/thread 8": "_ZN7mi_blur22blur_fused_tail_kernelILi3ELi1ELi8ELb0EEEvNS_11TiledParamsENS_11FusedParamsE",
        "5x5 tiled, rows/thread 8 (raw-window row pass: shipped)": "_ZN7mi_blur17blur_tiled_kernelILi3ELi2ELi8ELb1ELb0ELb0EEEvNS_11TiledParamsE",
        "5x5 tiled, rows/thread 8 (split-then-shift row pass: round-1 form)": "_ZN7mi_blur19blur_tiled_x_kernelILi3ELi2ELi8ELi0EEEvNS_11TiledParamsE",
        "5x5 streaming variant": "_ZN7mi_blur18blur_stream_kernelILi3ELi2EEEvNS_12StreamParamsE",
        "3x3 direct (LDS-free), 8 rows per lane": "_ZN7mi_blur18blur_direct_kernelILi3ELi1ELi8EEEvNS_12DirectParamsE",
        "5x5 direct (LDS-free), 8 rows per lane": "_ZN7mi_blur18blur_direct_kernelILi3ELi2ELi8EEEvNS_12DirectParamsE",
    }
    print("static counts per wave, gfx950, hipcc -O3 (C = 3); VALU per output dword = VALU / (8 rows x 4 dwords)")
    for label, sym in want.items():
        try:
            a = next(i for i, l in enumerate(text) if l.startswith(sym + ":"))
        except StopIteration:
            print(f"{label}: symbol not found"); continue
        b = next(i for i in range(a, len(text)) if text[i].startswith(".Lfunc_end"))
        body = text[a:b]
        ops = [l.split()[0] for l in body if re.match(r"^\s+[a-z]", l) and not l.strip().startswith((".", ";"))]
        valu = sum(o.startswith("v_") for o in ops)
        c = lambda p: sum(o.startswith(p) for o in ops)
        per = f"{valu / 32:.1f}" if "streaming" not in label else "n/a (loop)"
        print(f"{label:70s} VALU {valu:5d} ({per} per output dword)  v_perm {c('v_perm'):4d}  v_alignbit {c('v_alignbit'):4d}  v_and {c('v_and_b32'):4d}  "
              f"v_pk_mad {c('v_pk_mad'):4d}  ds_read {c('ds_read'):3d}  vmem {c('global_') + c('buffer_'):3d}  s_waitcnt {c('s_waitcnt'):3d}")


if __name__ == "__main__":
    main()
