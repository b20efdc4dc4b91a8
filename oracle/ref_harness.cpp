// ref_harness.cpp — NDRange dispatcher for the UNMODIFIED reference kernel.
//
// TEST INFRASTRUCTURE, NOT PRODUCT (see oracle/oracle_blur.c header).
//
// oracle/Makefile compiles /root/reference/gaussian_kernel.cl where it lies with
// the image's own OpenCL C front-end (clang -x cl, host x86-64 target) into an
// object that exports `gaussian_blur` and leaves three OpenCL built-ins
// undefined: get_global_id(uint), min(int,int), max(int,int).  This file plays
// the part of the OpenCL runtime for exactly those three and for the launch:
// it walks the same padded 2-D NDRange the reference host enqueues
// (local 16x16, global rounded up: heterogeneous_blur.c:397-400) and calls the
// kernel once per work-item.  Nothing of the reference is copied here; the
// output (.so) goes to oracle/_ref/ only (git-ignored, travels to the GPU box).
#include <cstddef>
#include <cstdint>

// The three built-ins the compiled kernel imports (C++ mangling matches the
// OpenCL C overloads: _Z13get_global_idj, _Z3minii, _Z3maxii).
#define ORACLE_LOCAL __attribute__((visibility("hidden")))
static thread_local size_t g_gid[3];
ORACLE_LOCAL size_t get_global_id(unsigned int dim) { return dim < 3 ? g_gid[dim] : 0; }
ORACLE_LOCAL int min(int a, int b) { return a < b ? a : b; }
ORACLE_LOCAL int max(int a, int b) { return a > b ? a : b; }

// Entry point emitted by clang for `__kernel void gaussian_blur(...)`
// (gaussian_kernel.cl:19-25).
extern "C" void gaussian_blur(const unsigned char *input, unsigned char *output,
                              int width, int height, int channels);

extern "C" {

// One clEnqueueNDRangeKernel of the reference (heterogeneous_blur.c:507,525).
void ref_gaussian_blur(const uint8_t *in, uint8_t *out, int width, int height, int channels)
{
    const size_t local = 16;
    const size_t gx = ((size_t)width + local - 1) / local * local;
    const size_t gy = ((size_t)height + local - 1) / local * local;
    for (size_t y = 0; y < gy; y++)
        for (size_t x = 0; x < gx; x++) {
            g_gid[0] = x; g_gid[1] = y; g_gid[2] = 0;
            gaussian_blur(in, out, width, height, channels);
        }
}

// Approach-2 per-image procedure exactly as the reference host drives it
// (split_image_blur.c:511-541): two launches on overlapping sub-buffers with
// height = sub-buffer rows incl. halo (:401,414), outputs sized like the
// inputs (:375,383), read-back drops the halo rows (:526,537).  HALO = 1 (:70).
// tmp must hold width*channels*max(split_row+1, height-split_row+1) bytes.
void ref_split_image_blur(const uint8_t *in, uint8_t *out, uint8_t *tmp,
                          int width, int height, int channels, int split_row)
{
    const int HALO = 1;
    const size_t pitch = (size_t)width * channels;
    const int cpu_in = split_row + HALO, gpu_in = (height - split_row) + HALO;
    ref_gaussian_blur(in, tmp, width, cpu_in, channels);
    for (size_t i = 0; i < pitch * (size_t)split_row; i++) out[i] = tmp[i];
    ref_gaussian_blur(in + (size_t)(split_row - HALO) * pitch, tmp, width, gpu_in, channels);
    for (size_t i = 0; i < pitch * (size_t)(height - split_row); i++)
        out[(size_t)split_row * pitch + i] = tmp[(size_t)HALO * pitch + i];
}

}  // extern "C"
