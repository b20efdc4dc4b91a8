// Developer microbenchmark: issue cost (cycles per wave-instruction per SIMD) of the VALU ops the
// blur kernel is built from, on gfx950.  One block per CU; 1, 2 or 4 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define REP16(x) x x x x x x x x x x x x x x x x
#define ITERS 512

#define DEF_KERNEL(NAME, ASM)                                                              \
__global__ __launch_bounds__(1024) void k_##NAME(unsigned long long *out, unsigned seed) { \
    unsigned a0 = threadIdx.x * 2654435761u + seed, a1 = a0 ^ 0x1234567u, a2 = a0 * 3u, a3 = a0 + 77u; \
    unsigned b0 = a0 >> 3, b1 = a1 >> 5, b2 = a2 >> 7, b3 = a3 >> 9;                        \
    unsigned c = seed | 0x00010001u;                                                      \
    unsigned long long t0 = __builtin_amdgcn_s_memtime();                                 \
    for (int i = 0; i < ITERS; i++) {                                                     \
        asm volatile(REP16(ASM) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "v"(c) : "s20", "s21"); \
    }                                                                                     \
    unsigned long long t1 = __builtin_amdgcn_s_memtime();                                 \
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;    \
    if (a0 + a1 + a2 + a3 + b0 + b1 + b2 + b3 == 0x12345) out[0] = 1;                     \
}

// each ASM string = 8 independent instructions (dest a0..a3,b0..b3), so 16 reps = 128 instr / iter
DEF_KERNEL(perm,      "v_perm_b32 %0, %0, %8, %8\n v_perm_b32 %1, %1, %8, %8\n v_perm_b32 %2, %2, %8, %8\n v_perm_b32 %3, %3, %8, %8\n v_perm_b32 %4, %4, %8, %8\n v_perm_b32 %5, %5, %8, %8\n v_perm_b32 %6, %6, %8, %8\n v_perm_b32 %7, %7, %8, %8\n")
DEF_KERNEL(alignbyte, "v_alignbyte_b32 %0, %0, %8, 1\n v_alignbyte_b32 %1, %1, %8, 1\n v_alignbyte_b32 %2, %2, %8, 1\n v_alignbyte_b32 %3, %3, %8, 1\n v_alignbyte_b32 %4, %4, %8, 1\n v_alignbyte_b32 %5, %5, %8, 1\n v_alignbyte_b32 %6, %6, %8, 1\n v_alignbyte_b32 %7, %7, %8, 1\n")
DEF_KERNEL(pk_add,    "v_pk_add_u16 %0, %0, %8\n v_pk_add_u16 %1, %1, %8\n v_pk_add_u16 %2, %2, %8\n v_pk_add_u16 %3, %3, %8\n v_pk_add_u16 %4, %4, %8\n v_pk_add_u16 %5, %5, %8\n v_pk_add_u16 %6, %6, %8\n v_pk_add_u16 %7, %7, %8\n")
DEF_KERNEL(pk_mad,    "v_pk_mad_u16 %0, %0, %8, %8\n v_pk_mad_u16 %1, %1, %8, %8\n v_pk_mad_u16 %2, %2, %8, %8\n v_pk_mad_u16 %3, %3, %8, %8\n v_pk_mad_u16 %4, %4, %8, %8\n v_pk_mad_u16 %5, %5, %8, %8\n v_pk_mad_u16 %6, %6, %8, %8\n v_pk_mad_u16 %7, %7, %8, %8\n")
DEF_KERNEL(pk_shl,    "v_pk_lshlrev_b16 %0, 4, %0 op_sel_hi:[0,1]\n v_pk_lshlrev_b16 %1, 4, %1 op_sel_hi:[0,1]\n v_pk_lshlrev_b16 %2, 4, %2 op_sel_hi:[0,1]\n v_pk_lshlrev_b16 %3, 4, %3 op_sel_hi:[0,1]\n v_pk_lshlrev_b16 %4, 4, %4 op_sel_hi:[0,1]\n v_pk_lshlrev_b16 %5, 4, %5 op_sel_hi:[0,1]\n v_pk_lshlrev_b16 %6, 4, %6 op_sel_hi:[0,1]\n v_pk_lshlrev_b16 %7, 4, %7 op_sel_hi:[0,1]\n")
DEF_KERNEL(pk_mul,    "v_pk_mul_lo_u16 %0, %0, %8\n v_pk_mul_lo_u16 %1, %1, %8\n v_pk_mul_lo_u16 %2, %2, %8\n v_pk_mul_lo_u16 %3, %3, %8\n v_pk_mul_lo_u16 %4, %4, %8\n v_pk_mul_lo_u16 %5, %5, %8\n v_pk_mul_lo_u16 %6, %6, %8\n v_pk_mul_lo_u16 %7, %7, %8\n")
DEF_KERNEL(and32,     "v_and_b32 %0, %0, %8\n v_and_b32 %1, %1, %8\n v_and_b32 %2, %2, %8\n v_and_b32 %3, %3, %8\n v_and_b32 %4, %4, %8\n v_and_b32 %5, %5, %8\n v_and_b32 %6, %6, %8\n v_and_b32 %7, %7, %8\n")
DEF_KERNEL(add32,     "v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n")
DEF_KERNEL(lshr32,    "v_lshrrev_b32 %0, 4, %0\n v_lshrrev_b32 %1, 4, %1\n v_lshrrev_b32 %2, 4, %2\n v_lshrrev_b32 %3, 4, %3\n v_lshrrev_b32 %4, 4, %4\n v_lshrrev_b32 %5, 4, %5\n v_lshrrev_b32 %6, 4, %6\n v_lshrrev_b32 %7, 4, %7\n")
DEF_KERNEL(lshl_or,   "v_lshl_or_b32 %0, %0, 8, %8\n v_lshl_or_b32 %1, %1, 8, %8\n v_lshl_or_b32 %2, %2, 8, %8\n v_lshl_or_b32 %3, %3, 8, %8\n v_lshl_or_b32 %4, %4, 8, %8\n v_lshl_or_b32 %5, %5, 8, %8\n v_lshl_or_b32 %6, %6, 8, %8\n v_lshl_or_b32 %7, %7, 8, %8\n")
DEF_KERNEL(add3,      "v_add3_u32 %0, %0, %8, %8\n v_add3_u32 %1, %1, %8, %8\n v_add3_u32 %2, %2, %8, %8\n v_add3_u32 %3, %3, %8, %8\n v_add3_u32 %4, %4, %8, %8\n v_add3_u32 %5, %5, %8, %8\n v_add3_u32 %6, %6, %8, %8\n v_add3_u32 %7, %7, %8, %8\n")
DEF_KERNEL(mad_u24,   "v_mad_u32_u24 %0, %0, %8, %8\n v_mad_u32_u24 %1, %1, %8, %8\n v_mad_u32_u24 %2, %2, %8, %8\n v_mad_u32_u24 %3, %3, %8, %8\n v_mad_u32_u24 %4, %4, %8, %8\n v_mad_u32_u24 %5, %5, %8, %8\n v_mad_u32_u24 %6, %6, %8, %8\n v_mad_u32_u24 %7, %7, %8, %8\n")
DEF_KERNEL(dot4,      "v_dot4_u32_u8 %0, %0, %8, %0\n v_dot4_u32_u8 %1, %1, %8, %1\n v_dot4_u32_u8 %2, %2, %8, %2\n v_dot4_u32_u8 %3, %3, %8, %3\n v_dot4_u32_u8 %4, %4, %8, %4\n v_dot4_u32_u8 %5, %5, %8, %5\n v_dot4_u32_u8 %6, %6, %8, %6\n v_dot4_u32_u8 %7, %7, %8, %7\n")
DEF_KERNEL(dpp_shr,   "v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %2 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %3 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %4, %4 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %5 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %6 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %7 wave_shr:1 row_mask:0xf bank_mask:0xf\n")
DEF_KERNEL(cndmask,   "v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n")
DEF_KERNEL(sad_u8,    "v_sad_u8 %0, %0, %8, %0\n v_sad_u8 %1, %1, %8, %1\n v_sad_u8 %2, %2, %8, %2\n v_sad_u8 %3, %3, %8, %3\n v_sad_u8 %4, %4, %8, %4\n v_sad_u8 %5, %5, %8, %5\n v_sad_u8 %6, %6, %8, %6\n v_sad_u8 %7, %7, %8, %7\n")
DEF_KERNEL(mad_u16,   "v_mad_u16 %0, %0, %8, %8\n v_mad_u16 %1, %1, %8, %8\n v_mad_u16 %2, %2, %8, %8\n v_mad_u16 %3, %3, %8, %8\n v_mad_u16 %4, %4, %8, %8\n v_mad_u16 %5, %5, %8, %8\n v_mad_u16 %6, %6, %8, %8\n v_mad_u16 %7, %7, %8, %8\n")
DEF_KERNEL(fma_f32,   "v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8\n")
DEF_KERNEL(pk_fma_f16,"v_pk_fma_f16 %0, %0, %8, %8\n v_pk_fma_f16 %1, %1, %8, %8\n v_pk_fma_f16 %2, %2, %8, %8\n v_pk_fma_f16 %3, %3, %8, %8\n v_pk_fma_f16 %4, %4, %8, %8\n v_pk_fma_f16 %5, %5, %8, %8\n v_pk_fma_f16 %6, %6, %8, %8\n v_pk_fma_f16 %7, %7, %8, %8\n")
DEF_KERNEL(bfe,       "v_bfe_u32 %0, %0, 8, 8\n v_bfe_u32 %1, %1, 8, 8\n v_bfe_u32 %2, %2, 8, 8\n v_bfe_u32 %3, %3, 8, 8\n v_bfe_u32 %4, %4, 8, 8\n v_bfe_u32 %5, %5, 8, 8\n v_bfe_u32 %6, %6, 8, 8\n v_bfe_u32 %7, %7, 8, 8\n")
DEF_KERNEL(cvt_pk_u8, "v_cvt_pk_u8_f32 %0, %0, 1, %8\n v_cvt_pk_u8_f32 %1, %1, 1, %8\n v_cvt_pk_u8_f32 %2, %2, 1, %8\n v_cvt_pk_u8_f32 %3, %3, 1, %8\n v_cvt_pk_u8_f32 %4, %4, 1, %8\n v_cvt_pk_u8_f32 %5, %5, 1, %8\n v_cvt_pk_u8_f32 %6, %6, 1, %8\n v_cvt_pk_u8_f32 %7, %7, 1, %8\n")

DEF_KERNEL(bfi,       "v_bfi_b32 %0, %8, %0, %8\n v_bfi_b32 %1, %8, %1, %8\n v_bfi_b32 %2, %8, %2, %8\n v_bfi_b32 %3, %8, %3, %8\n v_bfi_b32 %4, %8, %4, %8\n v_bfi_b32 %5, %8, %5, %8\n v_bfi_b32 %6, %8, %6, %8\n v_bfi_b32 %7, %8, %7, %8\n")
DEF_KERNEL(alignbit,  "v_alignbit_b32 %0, %0, %8, 16\n v_alignbit_b32 %1, %1, %8, 16\n v_alignbit_b32 %2, %2, %8, 16\n v_alignbit_b32 %3, %3, %8, 16\n v_alignbit_b32 %4, %4, %8, 16\n v_alignbit_b32 %5, %5, %8, 16\n v_alignbit_b32 %6, %6, %8, 16\n v_alignbit_b32 %7, %7, %8, 16\n")
DEF_KERNEL(add_lshl,  "v_add_lshl_u32 %0, %0, %8, 2\n v_add_lshl_u32 %1, %1, %8, 2\n v_add_lshl_u32 %2, %2, %8, 2\n v_add_lshl_u32 %3, %3, %8, 2\n v_add_lshl_u32 %4, %4, %8, 2\n v_add_lshl_u32 %5, %5, %8, 2\n v_add_lshl_u32 %6, %6, %8, 2\n v_add_lshl_u32 %7, %7, %8, 2\n")
DEF_KERNEL(lshl_add,  "v_lshl_add_u32 %0, %0, 1, %8\n v_lshl_add_u32 %1, %1, 1, %8\n v_lshl_add_u32 %2, %2, 1, %8\n v_lshl_add_u32 %3, %3, 1, %8\n v_lshl_add_u32 %4, %4, 1, %8\n v_lshl_add_u32 %5, %5, 1, %8\n v_lshl_add_u32 %6, %6, 1, %8\n v_lshl_add_u32 %7, %7, 1, %8\n")
DEF_KERNEL(and_or,    "v_and_or_b32 %0, %0, %8, %8\n v_and_or_b32 %1, %1, %8, %8\n v_and_or_b32 %2, %2, %8, %8\n v_and_or_b32 %3, %3, %8, %8\n v_and_or_b32 %4, %4, %8, %8\n v_and_or_b32 %5, %5, %8, %8\n v_and_or_b32 %6, %6, %8, %8\n v_and_or_b32 %7, %7, %8, %8\n")
DEF_KERNEL(cnd_e64,   "v_cndmask_b32_e64 %0, %0, %8, s[20:21]\n v_cndmask_b32_e64 %1, %1, %8, s[20:21]\n v_cndmask_b32_e64 %2, %2, %8, s[20:21]\n v_cndmask_b32_e64 %3, %3, %8, s[20:21]\n v_cndmask_b32_e64 %4, %4, %8, s[20:21]\n v_cndmask_b32_e64 %5, %5, %8, s[20:21]\n v_cndmask_b32_e64 %6, %6, %8, s[20:21]\n v_cndmask_b32_e64 %7, %7, %8, s[20:21]\n")
DEF_KERNEL(and_lit,   "v_and_b32 %0, 0x00ff00ff, %0\n v_and_b32 %1, 0x00ff00ff, %1\n v_and_b32 %2, 0x00ff00ff, %2\n v_and_b32 %3, 0x00ff00ff, %3\n v_and_b32 %4, 0x00ff00ff, %4\n v_and_b32 %5, 0x00ff00ff, %5\n v_and_b32 %6, 0x00ff00ff, %6\n v_and_b32 %7, 0x00ff00ff, %7\n")
DEF_KERNEL(pk_lshr,   "v_pk_lshrrev_b16 %0, 8, %0 op_sel_hi:[0,1]\n v_pk_lshrrev_b16 %1, 8, %1 op_sel_hi:[0,1]\n v_pk_lshrrev_b16 %2, 8, %2 op_sel_hi:[0,1]\n v_pk_lshrrev_b16 %3, 8, %3 op_sel_hi:[0,1]\n v_pk_lshrrev_b16 %4, 8, %4 op_sel_hi:[0,1]\n v_pk_lshrrev_b16 %5, 8, %5 op_sel_hi:[0,1]\n v_pk_lshrrev_b16 %6, 8, %6 op_sel_hi:[0,1]\n v_pk_lshrrev_b16 %7, 8, %7 op_sel_hi:[0,1]\n")

typedef void (*kfn)(unsigned long long *, unsigned);
struct K { const char *name; kfn fn; };

int main()
{
    K ks[] = {{"v_perm_b32", k_perm}, {"v_alignbyte_b32", k_alignbyte}, {"v_pk_add_u16", k_pk_add}, {"v_pk_mad_u16", k_pk_mad},
              {"v_pk_lshlrev_b16", k_pk_shl}, {"v_pk_mul_lo_u16", k_pk_mul}, {"v_and_b32", k_and32}, {"v_add_u32", k_add32},
              {"v_lshrrev_b32", k_lshr32}, {"v_lshl_or_b32", k_lshl_or}, {"v_add3_u32", k_add3}, {"v_mad_u32_u24", k_mad_u24},
              {"v_dot4_u32_u8", k_dot4}, {"v_mov_b32_dpp wave_shr", k_dpp_shr}, {"v_cndmask_b32", k_cndmask}, {"v_sad_u8", k_sad_u8},
              {"v_mad_u16", k_mad_u16}, {"v_fma_f32", k_fma_f32}, {"v_pk_fma_f16", k_pk_fma_f16}, {"v_bfe_u32", k_bfe}, {"v_cvt_pk_u8_f32", k_cvt_pk_u8},
              {"v_bfi_b32", k_bfi}, {"v_alignbit_b32", k_alignbit}, {"v_add_lshl_u32", k_add_lshl}, {"v_lshl_add_u32", k_lshl_add},
              {"v_and_or_b32", k_and_or}, {"v_cndmask_b32_e64 sgpr", k_cnd_e64}, {"v_and_b32 literal", k_and_lit}, {"v_pk_lshrrev_b16", k_pk_lshr}};
    unsigned long long *d;
    hipMalloc(&d, 256 * 16 * sizeof(unsigned long long));
    std::vector<unsigned long long> h(256 * 16);
    printf("%-24s %10s %10s %10s   (cycles per wave-instruction per SIMD, aggregate over resident waves)\n", "op", "1 w/SIMD", "2 w/SIMD", "4 w/SIMD");
    for (auto &k : ks) {
        printf("%-24s", k.name);
        for (int wps : {1, 2, 4}) {
            const int threads = 256 * wps;
            hipLaunchKernelGGL(k.fn, dim3(256), dim3(threads), 0, 0, d, 1u);   // warm
            hipMemset(d, 0, 256 * 16 * 8);
            hipLaunchKernelGGL(k.fn, dim3(256), dim3(threads), 0, 0, d, 1u);
            hipDeviceSynchronize();
            hipMemcpy(h.data(), d, 256 * 16 * 8, hipMemcpyDeviceToHost);
            double sum = 0; int n = 0;
            for (int b = 0; b < 256; b++) for (int w = 0; w < 4 * wps; w++) { sum += (double)h[b * 16 + w]; n++; }
            const double per_wave = sum / n;                     // cycles for ITERS*128 instr of one wave
            printf(" %10.2f", per_wave / (ITERS * 128.0) / wps); // SIMD-cycles per wave-instr with wps waves sharing it
        }
        printf("\n");
    }
    return 0;
}
