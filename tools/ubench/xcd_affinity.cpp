// Developer probe: is there an affinity between XCDs and regions of a buffer?  For every (XCD x, eighth r of the buffers)
// only the workgroups that find themselves on XCD x (s_getreg XCC_ID) stream region r (read 16 B/lane from `in`, write to
// `out`), all other workgroups exit at once.  Prints the 8x8 matrix of GB/s, then the plain one-launch blur on the same
// buffers (libmi_blur), for a few fresh allocations.
//   hipcc --offload-arch=gfx950 -O3 -x hip -o tools/ubench/xcd_affinity tools/ubench/xcd_affinity.cpp -Iinclude -L<pkg> -lmi_blur -Wl,-rpath,<pkg>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "mi_blur.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void region_copy(const uint4 *in, uint4 *out, size_t first, size_t count, int xcd, unsigned *census)
{
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    id &= 0xf;
    if (threadIdx.x == 0 && census) atomicAdd(&census[id], 1u);
    if ((int)id != xcd) return;
    // workgroups of this XCD share the region: rank among them is unknown, so stride by the whole grid / 8
    const size_t stride = (size_t)(gridDim.x / 8) * blockDim.x;
    for (size_t i = (size_t)(blockIdx.x / 8) * blockDim.x + threadIdx.x; i < count; i += stride) {
        uint4 v = in[first + i];
        v.x += 1;
        out[first + i] = v;
    }
}

// The blur's access shape without its arithmetic: workgroup -> one 26 112-byte tile (34 rows x 768 B), tiles dealt so that
// each XCD gets one contiguous eighth (as blur_tiled_kernel does); every workgroup leaves its end time (s_memrealtime,
// 100 MHz) in its XCD's slot, so the spread between XCDs shows which of them the launch waits for.
__global__ __launch_bounds__(192) void tile_copy(const uint4 *in, uint4 *out, unsigned ntiles, unsigned long long *tend, unsigned long long *tbeg)
{
    const unsigned L = blockIdx.x, q = ntiles >> 3, r = ntiles & 7u, x = L & 7u, k = L >> 3;
    const unsigned tile = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + k;
    const size_t base = (size_t)tile * (32 * 768 / 16);           // output rows of the tile, in 16-byte units
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    uint4 v[8];
#pragma unroll
    for (int i = 0; i < 8; i++) v[i] = in[base + (size_t)i * 192 + threadIdx.x];
#pragma unroll
    for (int i = 0; i < 8; i++) { v[i].x += 1; out[base + (size_t)i * 192 + threadIdx.x] = v[i]; }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        id &= 7;
        atomicMax(&tend[id], __builtin_amdgcn_s_memrealtime());
        atomicMin(&tbeg[id], t0);
    }
}

int main()
{
    const int W = 256, H = 256, C = 3, N = 5000;
    const size_t bytes = (size_t)N * W * H * C, n16 = bytes / 16, per = n16 / 8;
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    unsigned *census; CK(hipMalloc(&census, 64)); CK(hipMemset(census, 0, 64));
    std::vector<void *> junk;
    for (int alloc = 0; alloc < 4; alloc++) {
        uint8_t *in, *out;
        if (alloc == 2) { void *j; CK(hipMalloc(&j, (size_t)300 << 20)); junk.push_back(j); }
        CK(hipMalloc((void **)&in, bytes)); CK(hipMalloc((void **)&out, bytes));
        CK(hipMemset(in, 0x5a, bytes));
        // warm the clocks
        for (int i = 0; i < 200; i++) mi_blur_enqueue(in, out, W, H, C, 1, N, s);
        CK(hipStreamSynchronize(s));
        float ms;
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < 40; i++) mi_blur_enqueue(in, out, W, H, C, 1, N, s);
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("allocation %d: in %p out %p   one-launch blur %.2f us (%.0f GB/s)\n", alloc, (void *)in, (void *)out, ms * 1e3 / 40, 2.0 * bytes / (ms / 40 * 1e-3) / 1e9);
        const int grid = 256 * 8;
        if (alloc == 0) {
            region_copy<<<grid, 256, 0, s>>>((const uint4 *)in, (uint4 *)out, 0, 0, -1, census);
            CK(hipStreamSynchronize(s));
            unsigned h[16]; CK(hipMemcpy(h, census, 64, hipMemcpyDeviceToHost));
            printf("  census of %d workgroups by XCC_ID:", grid); for (int i = 0; i < 8; i++) printf(" %u", h[i]); printf("\n");
        }
        printf("  GB/s (read+write) by [XCD x][region r]\n");
        for (int x = 0; x < 8; x++) {
            printf("  xcd %d:", x);
            for (int r = 0; r < 8; r++) {
                for (int rep = 0; rep < 2; rep++) region_copy<<<grid, 256, 0, s>>>((const uint4 *)in, (uint4 *)out, r * per, per, x, nullptr);
                CK(hipEventRecord(e0, s));
                for (int rep = 0; rep < 4; rep++) region_copy<<<grid, 256, 0, s>>>((const uint4 *)in, (uint4 *)out, r * per, per, x, nullptr);
                CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
                printf(" %6.0f", 2.0 * per * 16 / (ms / 4 * 1e-3) / 1e9);
            }
            printf("\n");
        }
        {   // the blur's tile shape and XCD-contiguous map, copy only: when does each XCD finish?
            unsigned long long *tt; CK(hipMalloc(&tt, 256));
            const unsigned ntiles = N * 8;
            for (int rep = 0; rep < 3; rep++) {
                unsigned long long init[32];
                for (int i = 0; i < 16; i++) { init[i] = 0; init[16 + i] = ~0ull; }
                CK(hipMemcpy(tt, init, 256, hipMemcpyHostToDevice));
                CK(hipEventRecord(e0, s));
                tile_copy<<<ntiles, 192, 0, s>>>((const uint4 *)in, (uint4 *)out, ntiles, tt, tt + 16);
                CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
                unsigned long long h[32]; CK(hipMemcpy(h, tt, 256, hipMemcpyDeviceToHost));
                unsigned long long t0 = ~0ull; for (int i = 0; i < 8; i++) if (h[16 + i] < t0) t0 = h[16 + i];
                printf("  tile_copy %.2f us; per-XCD end - launch start (us):", ms * 1e3);
                for (int i = 0; i < 8; i++) printf(" %6.1f", (double)(h[i] - t0) / 100.0);
                printf("\n");
            }
            CK(hipFree(tt));
        }
        CK(hipFree(in)); CK(hipFree(out));
    }
    return 0;
}
