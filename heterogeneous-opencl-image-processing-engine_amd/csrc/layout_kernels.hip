// layout_kernels.hip — planar <-> interleaved repacking on the GPU (gfx950).
//
// The reference loads frames through CImg, whose storage is PLANAR (all R, then all G, then all B:
// byte (x, y, c) at c*W*H + y*W + x), and repacks every frame on one host core into the interleaved
// stream the kernel consumes (heterogeneous_blur.c:125-134), and back when it saves one
// (split_image_blur.c:40-56).  SURVEY 8(f).3 names that repack as the first thing that becomes the
// bottleneck once the blur runs at HBM speed.  Here it is one pass over HBM:
//
//   thread = 16 consecutive pixels of one image: one 16-byte load per plane (coalesced per plane),
//   a byte shuffle in registers (v_perm_b32: every output dword is built from at most C source dwords),
//   C 16-byte stores covering 16*C contiguous interleaved bytes.  The inverse is the mirror image.
//   Algorithmic bytes 2*W*H*C per image, HBM-bound, no LDS.
//
// W*H not a multiple of 16 (or C > 4, or unaligned pointers) takes a byte-per-thread kernel.
// (A planar stream can also be blurred with no repack at all: it IS a stream of n*C one-channel images —
// mi_blur_enqueue(..., channels = 1, n_images = n*C); see INTEGRATION.md.)
#include "blur_launch.h"
#include "../../include/mi_blur.h"

#include <limits.h>

namespace mi_blur {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// One 4-pixel group: in[c] / out[k] are dwords.  P2I: in[c] = 4 consecutive bytes of plane c, out = the 4*C
// interleaved bytes (out byte j = plane j%C, pixel j/C).  !P2I: in = 4*C interleaved bytes, out[c] = plane c.
template <int C, bool P2I>
__device__ __forceinline__ void shuffle4(const uint32_t (&in)[C], uint32_t (&out)[C])
{
#pragma unroll
    for (int k = 0; k < C; k++) {
        uint32_t r = 0;
#pragma unroll
        for (int s = 0; s < C; s++) {                 // fold source dword s into r where it contributes
            uint32_t sel = 0;
            bool used = false;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int j = P2I ? 4 * k + i : i * C + k;              // interleaved byte index of this output byte
                const int src_dw = P2I ? j % C : j / 4, src_b = P2I ? j / C : j % 4;
                const bool mine = src_dw == s;
                used |= mine;
                sel |= (uint32_t)(mine ? 4 + src_b : i) << (8 * i);     // 4..7 = bytes of in[s], 0..3 = keep r
            }
            if (used) r = __builtin_amdgcn_perm(in[s], r, sel);
        }
        out[k] = r;
    }
}

struct LayoutParams {
    const uint8_t *src;
    uint8_t *dst;
    long long plane;        // W*H
    long long groups;       // n_images * plane / 16
    long long total_bytes;  // n_images * plane * C   (byte kernel)
    int channels;
};

template <int C, bool P2I>
__global__ __launch_bounds__(256) void repack16_kernel(const LayoutParams p)
{
    const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
    if (g >= p.groups) return;
    const long long gpi = p.plane / 16;               // 16-pixel groups per image
    const long long img = g / gpi, q = g - img * gpi; // pixel 16q .. 16q+15 of image img
    const uint8_t *planar = (P2I ? p.src : p.dst) + img * p.plane * C + q * 16;
    const uint8_t *inter = (P2I ? p.dst : p.src) + img * p.plane * C + q * 16 * C;
    uint32_t pl[C][4], il[4][C];                      // pl[c][g4] = pixels 4*g4.. of plane c; il[g4][k] = interleaved dwords
    if constexpr (P2I) {
#pragma unroll
        for (int c = 0; c < C; c++) {
            const u32x4 v = *reinterpret_cast<const u32x4 *>(planar + (long long)c * p.plane);
            pl[c][0] = v.x; pl[c][1] = v.y; pl[c][2] = v.z; pl[c][3] = v.w;
        }
#pragma unroll
        for (int g4 = 0; g4 < 4; g4++) {
            uint32_t in[C];
#pragma unroll
            for (int c = 0; c < C; c++) in[c] = pl[c][g4];
            shuffle4<C, true>(in, il[g4]);
        }
        uint32_t flat[4 * C];
#pragma unroll
        for (int g4 = 0; g4 < 4; g4++)
#pragma unroll
            for (int k = 0; k < C; k++) flat[g4 * C + k] = il[g4][k];
#pragma unroll
        for (int k = 0; k < C; k++) {
            u32x4 v; v.x = flat[4 * k]; v.y = flat[4 * k + 1]; v.z = flat[4 * k + 2]; v.w = flat[4 * k + 3];
            *reinterpret_cast<u32x4 *>(const_cast<uint8_t *>(inter) + 16 * k) = v;
        }
    } else {
        uint32_t flat[4 * C];
#pragma unroll
        for (int k = 0; k < C; k++) {
            const u32x4 v = *reinterpret_cast<const u32x4 *>(inter + 16 * k);
            flat[4 * k] = v.x; flat[4 * k + 1] = v.y; flat[4 * k + 2] = v.z; flat[4 * k + 3] = v.w;
        }
#pragma unroll
        for (int g4 = 0; g4 < 4; g4++) {
            uint32_t in[C], out[C];
#pragma unroll
            for (int k = 0; k < C; k++) in[k] = flat[g4 * C + k];
            shuffle4<C, false>(in, out);
#pragma unroll
            for (int c = 0; c < C; c++) pl[c][g4] = out[c];
        }
#pragma unroll
        for (int c = 0; c < C; c++) {
            u32x4 v; v.x = pl[c][0]; v.y = pl[c][1]; v.z = pl[c][2]; v.w = pl[c][3];
            *reinterpret_cast<u32x4 *>(const_cast<uint8_t *>(planar) + (long long)c * p.plane) = v;
        }
    }
}

template <bool P2I>
__global__ __launch_bounds__(256) void repack_bytes_kernel(const LayoutParams p)
{
    const long long step = (long long)gridDim.x * 256;
    const long long isz = p.plane * p.channels;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < p.total_bytes; idx += step) {
        const long long img = idx / isz, j = idx - img * isz;          // j = interleaved byte index in the image
        const long long px = j / p.channels;
        const int c = (int)(j - px * p.channels);
        const long long pj = (long long)c * p.plane + px;               // the same byte in planar order
        if (P2I) p.dst[img * isz + j] = p.src[img * isz + pj];
        else p.dst[img * isz + pj] = p.src[img * isz + j];
    }
}

static inline int hip_status(hipError_t e) { return e == hipSuccess ? MI_BLUR_OK : MI_BLUR_ERR_HIP_BASE - (int)e; }

template <bool P2I>
static int repack(const uint8_t *src, uint8_t *dst, int width, int height, int channels, int n_images, hipStream_t stream)
{
    if (!src || !dst || src == dst || width <= 0 || height <= 0 || channels <= 0 || n_images < 0) return MI_BLUR_ERR_INVALID;
    if ((long long)width * height * channels > INT_MAX) return MI_BLUR_ERR_INVALID;
    if (n_images == 0) return MI_BLUR_OK;
    LayoutParams p{};
    p.src = src; p.dst = dst; p.plane = (long long)width * height; p.channels = channels;
    p.total_bytes = p.plane * channels * n_images;
    const bool vec = p.plane % 16 == 0 && channels <= 4 && (uintptr_t)src % 16 == 0 && (uintptr_t)dst % 16 == 0;
    if (vec) {
        p.groups = p.plane / 16 * n_images;
        const long long blocks = (p.groups + 255) / 256;
        if (blocks > 0x7fffffffLL) return MI_BLUR_ERR_INVALID;
        const dim3 grid((unsigned)blocks), block(256);
        switch (channels) {
        case 1: hipLaunchKernelGGL((repack16_kernel<1, P2I>), grid, block, 0, stream, p); break;
        case 2: hipLaunchKernelGGL((repack16_kernel<2, P2I>), grid, block, 0, stream, p); break;
        case 3: hipLaunchKernelGGL((repack16_kernel<3, P2I>), grid, block, 0, stream, p); break;
        case 4: hipLaunchKernelGGL((repack16_kernel<4, P2I>), grid, block, 0, stream, p); break;
        }
    } else {
        long long blocks = (p.total_bytes + 255) / 256;
        if (blocks > 256LL * 64) blocks = 256LL * 64;
        hipLaunchKernelGGL((repack_bytes_kernel<P2I>), dim3((unsigned)blocks), dim3(256), 0, stream, p);
    }
    return hip_status(hipGetLastError());
}

// Halo pull (mi_blur_halo_pull): up to two runs of `bytes` bytes, each from a (peer) source into this rank's halo rows.
// 16 bytes per thread where both ends are 16-byte aligned, bytes otherwise; blockIdx.y picks the run.
struct PullParams { const uint8_t *src[2]; uint8_t *dst[2]; unsigned long long bytes; };

__global__ __launch_bounds__(256) void halo_pull_kernel(const PullParams p)
{
    const uint8_t *s = p.src[blockIdx.y];
    uint8_t *d = p.dst[blockIdx.y];
    if (!s) return;
    const unsigned long long i = ((unsigned long long)blockIdx.x * 256 + threadIdx.x) * 16ull;
    if (i >= p.bytes) return;
    if (((uintptr_t)s | (uintptr_t)d) % 16 == 0 && i + 16 <= p.bytes) {
        *reinterpret_cast<u32x4 *>(d + i) = *reinterpret_cast<const u32x4 *>(s + i);
    } else {
        for (unsigned long long k = i; k < i + 16 && k < p.bytes; k++) d[k] = s[k];
    }
}

int launch_halo_pull(const uint8_t *top_src, uint8_t *top_dst, const uint8_t *bottom_src, uint8_t *bottom_dst, size_t bytes, hipStream_t stream)
{
    if (bytes == 0 || (!top_src && !bottom_src)) return MI_BLUR_OK;
    PullParams p{};
    p.src[0] = top_src; p.dst[0] = top_dst; p.src[1] = bottom_src; p.dst[1] = bottom_dst; p.bytes = bytes;
    const unsigned blocks = (unsigned)((bytes + 16 * 256 - 1) / (16 * 256));
    hipLaunchKernelGGL(halo_pull_kernel, dim3(blocks, 2), dim3(256), 0, stream, p);
    return hip_status(hipGetLastError());
}

int launch_planar_to_interleaved(const uint8_t *src, uint8_t *dst, int w, int h, int c, int n, hipStream_t s) { return repack<true>(src, dst, w, h, c, n, s); }
int launch_interleaved_to_planar(const uint8_t *src, uint8_t *dst, int w, int h, int c, int n, hipStream_t s) { return repack<false>(src, dst, w, h, c, n, s); }

}  // namespace mi_blur
