// blur_kernels.hip — gfx950 (MI355X / CDNA4) kernels of the image-stream blur engine.
//
// Replaces `__kernel gaussian_blur` (reference gaussian_kernel.cl:19-72): per pixel,
// per channel, 3x3 {1,2,1}x{1,2,1}/16 (radius 1) or build-defined 5x5
// {1,4,6,4,1}x{1,4,6,4,1}/256 (radius 2), clamp-to-edge (:56-57), truncation (:70),
// on interleaved uint8 with pitch = width*channels (:60).
//
// The reference accumulates in float; every weight is dyadic and every partial sum a
// multiple of 2^-4 (2^-8) below 2^24, so the result equals the integer form
// (sum w_i p_i) >> 4 (>> 8) in ANY summation order.  That licenses what the tiled
// kernel does: separable passes in packed 16-bit integers.
//
// Tiled kernel (bandwidth-bound stencil, no MFMA):
//   * a row is treated as a byte stream of pitch bytes; the x-neighbours of byte b are
//     b-C and b+C, so interleaved RGB needs no de-interleave;
//   * one workgroup stages a (TH + 2R) x (ncols + 2) tile of 16-byte chunks in LDS —
//     the output tile plus an R-row halo above/below (rows clamped at the band edge at
//     staging time, so the y-clamp costs nothing later) and one 16-byte halo chunk left
//     and right.  Staging is one tile row (or several short ones) per wave-instruction:
//     16 B/lane coalesced global_load_lds_dwordx4 (LDS-DMA, no VGPR round trip; the LDS
//     image of a row is exactly the lanes in order), a register path is kept for A/B;
//   * each thread owns one 16-byte chunk column and RPG consecutive rows.  Per row it reads
//     8+16+8 bytes from LDS and splits every dword into its EVEN bytes (x & 0x00ff00ff) and
//     ODD bytes (v_perm) — two 16-bit fields per dword.  No field ever exceeds 16 bits
//     (max 65280), so field arithmetic is plain 32-bit SWAR: v_add_u32 / v_lshl_add_u32
//     (VOP2/VOP3 scalar-width ops, measured ~1.5-2.6 cycles per wave-instruction against
//     2.6 for every v_pk_* / v_perm; tools/ubench/valu_rate.hip).  A byte shift of +-C
//     maps even/odd fields onto whole dwords of the other (or same) parity array, or onto a
//     16-bit funnel shift (v_alignbit_b32) — for C=3 half of the taps are free;
//   * a 2R+1-row sliding window of horizontal sums lives in registers; the vertical
//     combine, the final >>4 (>>8) and the even/odd re-interleave are 3 ops per output dword;
//   * the x-clamp is synthesised in registers (v_perm + v_cndmask) only in waves that contain
//     a lane whose chunk touches the row start/end (byte at position -k equals byte
//     (-k mod C); mirror at the end).
//   HBM traffic = each input byte once + each output byte once, plus the tile-edge halo
//   (2R/TH of the rows, 2/ncols of the columns), which neighbouring tiles keep in L2
//   (blockIdx -> tile map gives each XCD a contiguous run of tiles).  Measured with
//   rocprofv3 PMC: FETCH_SIZE*2 + WRITE_SIZE within 0.1-2.5 % of the algorithmic bytes.
//
// Ragged form of the tiled kernel (RAG): pitch NOT a multiple of 16 and/or unaligned pointers
// (1366-, 1000-, 250-pixel-wide frames).  A row is still cut into 16-byte chunks counted from the
// ROW START, so the LDS tile and the whole compute phase are unchanged; only the edges of the pipeline
// differ: staging gives LDS-DMA the unaligned global address as it is (the hardware splits the fetch); the
// lane that owns a row's last, partial chunk loads — through registers — the 16 bytes that END at the row end
// (never reading past the buffer), stores them at their row-relative LDS position and writes the right-edge clamp
// bytes (copies of the last pixel) behind them, so no lane synthesises the right clamp; outputs are unaligned
// 16-byte stores, the partial chunk a masked 8/4/2/1-byte store of the bytes that exist.
//
// Direct kernel (blur_direct_kernel): the same row-stream arithmetic with no LDS at all — every lane loads the 8 + 2R rows
// of its own chunk column straight into registers (all requests in flight at once, consumed row by row under the
// compiler's counted vmcnt), takes its x-neighbours' bytes from the adjacent lanes by DPP wave shifts, 62 computing lanes
// per wave.  AUTO's choice for 5x5 and for 3x3 launches that do not fill the chip for long; big 3x3 launches stay tiled.
//
// Generic kernel: one output byte per thread, any shape (rows shorter than 16 bytes, more than 4
// channels).  Correct everywhere, fast nowhere.
#include "blur_launch.h"
#include "../../include/mi_blur.h"

#include <hip/hip_ext.h>
#include <type_traits>
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include <algorithm>

namespace mi_blur {

// ----------------------------------------------------------------------------------
// device helpers
// ----------------------------------------------------------------------------------
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u16x2 as_pk(uint32_t x) { return __builtin_bit_cast(u16x2, x); }
__device__ __forceinline__ uint32_t as_u32(u16x2 x) { return __builtin_bit_cast(uint32_t, x); }

constexpr uint32_t EVEN_MASK = 0x00ff00ffu;
// even bytes (b0 | b2<<16) and odd bytes (b1 | b3<<16) of a dword as two 16-bit fields
__device__ __forceinline__ uint32_t even_of(uint32_t x) { return x & EVEN_MASK; }
__device__ __forceinline__ uint32_t odd_of(uint32_t x) { return __builtin_amdgcn_perm(0u, x, 0x0c030c01u); }
// 6*c + t on both 16-bit fields (v_pk_mad_u16; a 32-bit multiply would be quarter rate)
__device__ __forceinline__ uint32_t mad6(uint32_t c, uint32_t t) { return as_u32(as_pk(c) * (unsigned short)6 + as_pk(t)); }

// x-clamp selectors.  Left: the dword holding row-stream bytes [-4q, -4q+3] (q = 1, 2)
// when the chunk starts the row: byte at position p < 0 is a copy of byte (p mod C)
// (pixel x<0 clamps to x=0, same channel).  Indices address the chunk's first EIGHT bytes (v_perm_b32 over chunk dwords
// 1 and 0), so any C <= 8 works.
constexpr uint32_t sel_left(int C, int q)
{
    uint32_t s = 0;
    for (int j = 0; j < 4; j++) {
        int p = -4 * q + j;
        int idx = ((p % C) + C) % C;
        s |= (uint32_t)idx << (8 * j);
    }
    return s;
}
// Right: the dword holding bytes [pitch+4q, pitch+4q+3] (q = 0, 1) when the chunk ends
// the row: byte pitch+k is a copy of byte pitch-C+(k mod C).  Indices address the chunk's LAST eight bytes (v_perm_b32
// over chunk dwords 3 and 2: selector 0-3 = row bytes pitch-8 .. pitch-5, 4-7 = pitch-4 .. pitch-1), any C <= 8.
constexpr uint32_t sel_right(int C, int q)
{
    uint32_t s = 0;
    for (int j = 0; j < 4; j++) {
        int k = 4 * q + j;
        int idx = 8 - C + (k % C);
        s |= (uint32_t)idx << (8 * j);
    }
    return s;
}
// The clamp dwords themselves.  For C <= 4 every selected byte lies in ONE dword (the chunk's first / last), and the perm is
// written with that dword alone: with two live operands hipcc schedules the whole tile loop differently (39 instead of 54
// VGPRs, loads no longer hoisted) and the 3x3 stream loses 20 % — so the two-dword form is kept to C > 4, where it is needed.
template <int C>
__device__ __forceinline__ uint32_t clamp_left(uint32_t d0, uint32_t d1, int q)
{
    if constexpr (C <= 4) { (void)d1; return q == 1 ? __builtin_amdgcn_perm(0u, d0, sel_left(C, 1)) : __builtin_amdgcn_perm(0u, d0, sel_left(C, 2)); }
    else return q == 1 ? __builtin_amdgcn_perm(d1, d0, sel_left(C, 1)) : __builtin_amdgcn_perm(d1, d0, sel_left(C, 2));
}
template <int C, int Q>
__device__ __forceinline__ uint32_t clamp_right(uint32_t d2, uint32_t d3)
{
    if constexpr (C <= 4) { (void)d2; return __builtin_amdgcn_perm(0u, d3, sel_right(C, Q) - 0x04040404u); }   // indices 4..7 -> 0..3 of d3 alone
    else return __builtin_amdgcn_perm(d3, d2, sel_right(C, Q));
}

// Ew[j] / Ow[j] = even / odd bytes of window dword j, the window being row-stream bytes
// [-8, 24) around this thread's chunk (chunk = dwords 2..5).  tap<K, PI, I> = the two
// 16-bit fields that sit K bytes away from the PI-parity (0 even, 1 odd) bytes of chunk
// dword I: bytes 4I+PI+K and 4I+PI+K+2.  They are consecutive elements of one parity
// array, i.e. either a whole dword of it or a 16-bit funnel shift of two.
template <int K, int PI, int I>
__device__ __forceinline__ uint32_t tap(const uint32_t (&Ew)[8], const uint32_t (&Ow)[8])
{
    constexpr int q0 = 8 + 4 * I + PI + K;      // window byte index of the first field
    static_assert(q0 >= 0 && q0 + 2 < 32, "tap outside the staged window");
    constexpr int e0 = q0 >> 1, j = e0 >> 1;    // element / dword index in the parity array
    if constexpr ((q0 & 1) == 0) {
        if constexpr ((e0 & 1) == 0) return Ew[j];
        else return __builtin_amdgcn_alignbit(Ew[j + 1], Ew[j], 16);
    } else {
        if constexpr ((e0 & 1) == 0) return Ow[j];
        else return __builtin_amdgcn_alignbit(Ow[j + 1], Ow[j], 16);
    }
}

// The same two fields taken straight from the RAW window dwords w[0..7] (row-stream bytes [-8, 24)): bytes q0 and q0+2
// are at most 8 bytes apart, so one v_perm_b32 over (w[j+1], w[j]) — or a plain AND when they are bytes 0 and 2 of one
// dword — builds the field pair with no separate even/odd split and no funnel shift.  A row pass needs 24 (5x5) / 18
// (3x3) distinct pairs against 16 + 14 / 12 + 8 split-then-shift operations.
template <int K, int PI, int I>
__device__ __forceinline__ uint32_t tap_raw(const uint32_t (&w)[8])
{
    constexpr int q0 = 8 + 4 * I + PI + K;
    static_assert(q0 >= 0 && q0 + 2 < 32, "tap outside the staged window");
    constexpr int j = q0 >> 2, m = q0 & 3;
    if constexpr (m == 0) return w[j] & EVEN_MASK;
    else if constexpr (m == 1) return __builtin_amdgcn_perm(0u, w[j], 0x0c030c01u);
    else if constexpr (m == 2) return __builtin_amdgcn_perm(w[j + 1], w[j], 0x0c040c02u);
    else return __builtin_amdgcn_perm(w[j + 1], w[j], 0x0c050c03u);
}

template <int C, int R, int PI, int I>
__device__ __forceinline__ uint32_t hsum_raw(const uint32_t (&w)[8])
{
    const uint32_t c = tap_raw<0, PI, I>(w);
    if constexpr (R == 1) {
        return (tap_raw<-C, PI, I>(w) + tap_raw<C, PI, I>(w)) + (c << 1);
    } else {
        const uint32_t t = tap_raw<-2 * C, PI, I>(w) + tap_raw<2 * C, PI, I>(w);
        const uint32_t u = tap_raw<-C, PI, I>(w) + tap_raw<C, PI, I>(w);
        return mad6(c, (u << 2) + t);
    }
}

// Horizontal tap sum for the PI-parity bytes of chunk dword I (two 16-bit fields).
template <int C, int R, int PI, int I>
__device__ __forceinline__ uint32_t hsum(const uint32_t (&Ew)[8], const uint32_t (&Ow)[8])
{
    const uint32_t c = PI ? Ow[2 + I] : Ew[2 + I];
    if constexpr (R == 1) {
        return (tap<-C, PI, I>(Ew, Ow) + tap<C, PI, I>(Ew, Ow)) + (c << 1);          // <= 1020
    } else {
        const uint32_t t = tap<-2 * C, PI, I>(Ew, Ow) + tap<2 * C, PI, I>(Ew, Ow);
        const uint32_t u = tap<-C, PI, I>(Ew, Ow) + tap<C, PI, I>(Ew, Ow);
        return mad6(c, (u << 2) + t);                                                 // <= 4080
    }
}

// Horizontal pass of one staged LDS row for this thread's chunk:
// h[0..3] = even-byte sums of chunk dwords 0..3, h[4..7] = odd-byte sums.
template <int C, int R, int X>
__device__ __forceinline__ void hrow_window(uint32_t (&w)[8], bool any_edge, bool at_start, bool at_end, uint32_t (&h)[8]);

// SHFL = wavefront-shuffle row pass: the 8 bytes either side of the chunk come from the neighbouring lanes'
// registers (v_mov_b32_dpp wave_shr:1 / wave_shl:1); only lanes whose neighbour chunk is not held by the
// adjacent lane (first/last lane of the wave, first/last chunk column of the tile) read them from the LDS halo.
// !SHFL: every lane reads its 8+16+8 bytes from LDS.  Measured (profiles/): the LDS form is not slower — the
// two ds_read_b64 it saves cost no VALU slot, the four DPP moves it adds do — so it is the default.
// X = row-pass form: 0 = split every window dword into even/odd fields, then shift (tap); 1 = field pairs straight from the
// raw window (tap_raw).  Measured (profiles/r02_ab_row_pass.txt): 1 is 1.4 % faster for 5x5 (1129 against 1188 VALU per
// wave), 0 is the shorter code for 3x3 (748 against 783) — rowpass_default picks accordingly.
// (Reading the halo bytes as aligned 16-byte LDS reads instead of the two bank-conflicting 8-byte ones was measured too:
// no change — the LDS pipe is ~20 % busy, its conflicts cost nothing that shows.)
template <int R> constexpr int rowpass_default = R == 2 ? 1 : 0;
template <int C, int R, bool SHFL, int X = rowpass_default<R>>
__device__ __forceinline__ void hrow(const uint8_t *lp, bool any_edge, bool at_start, bool at_end, bool lds_left, bool lds_right,
                                     uint32_t (&h)[8])
{
    uint32_t w[8];
    const uint4 c = *reinterpret_cast<const uint4 *>(lp);
    w[2] = c.x; w[3] = c.y; w[4] = c.z; w[5] = c.w;
    if constexpr (SHFL) {
        uint2 a = make_uint2(0u, 0u), b = make_uint2(0u, 0u);
        if (lds_left) a = *reinterpret_cast<const uint2 *>(lp - 8);
        if (lds_right) b = *reinterpret_cast<const uint2 *>(lp + 16);
        // lane i takes lane i-1's last two dwords / lane i+1's first two; the LDS value where it was read
        const uint32_t l0 = __builtin_amdgcn_update_dpp(0u, c.z, 0x138, 0xf, 0xf, false);   // wave_shr:1
        const uint32_t l1 = __builtin_amdgcn_update_dpp(0u, c.w, 0x138, 0xf, 0xf, false);
        const uint32_t r0 = __builtin_amdgcn_update_dpp(0u, c.x, 0x130, 0xf, 0xf, false);   // wave_shl:1
        const uint32_t r1 = __builtin_amdgcn_update_dpp(0u, c.y, 0x130, 0xf, 0xf, false);
        w[0] = lds_left ? a.x : l0; w[1] = lds_left ? a.y : l1;
        w[6] = lds_right ? b.x : r0; w[7] = lds_right ? b.y : r1;
    } else {
        const uint2 a = *reinterpret_cast<const uint2 *>(lp - 8);
        const uint2 b = *reinterpret_cast<const uint2 *>(lp + 16);
        w[0] = a.x; w[1] = a.y; w[6] = b.x; w[7] = b.y;
    }
    hrow_window<C, R, X>(w, any_edge, at_start, at_end, h);
}

// Same, from a register window w[0..7] = row-stream bytes [-8, 24) around the chunk.
template <int C, int R, int X>
__device__ __forceinline__ void hrow_window(uint32_t (&w)[8], bool any_edge, bool at_start, bool at_end, uint32_t (&h)[8])
{
    if (any_edge) {   // wave-uniform: some lane's chunk starts or ends the image row
        w[1] = at_start ? clamp_left<C>(w[2], w[3], 1) : w[1];
        w[0] = at_start ? clamp_left<C>(w[2], w[3], 2) : w[0];
        w[6] = at_end ? clamp_right<C, 0>(w[4], w[5]) : w[6];
        w[7] = at_end ? clamp_right<C, 1>(w[4], w[5]) : w[7];
    }
    if constexpr (X == 1) {
        h[0] = hsum_raw<C, R, 0, 0>(w); h[1] = hsum_raw<C, R, 0, 1>(w);
        h[2] = hsum_raw<C, R, 0, 2>(w); h[3] = hsum_raw<C, R, 0, 3>(w);
        h[4] = hsum_raw<C, R, 1, 0>(w); h[5] = hsum_raw<C, R, 1, 1>(w);
        h[6] = hsum_raw<C, R, 1, 2>(w); h[7] = hsum_raw<C, R, 1, 3>(w);
        return;
    }
    uint32_t Ew[8], Ow[8];
#pragma unroll
    for (int j = 0; j < 8; j++) { Ew[j] = even_of(w[j]); Ow[j] = odd_of(w[j]); }   // unused ones are dead code
    h[0] = hsum<C, R, 0, 0>(Ew, Ow); h[1] = hsum<C, R, 0, 1>(Ew, Ow);
    h[2] = hsum<C, R, 0, 2>(Ew, Ow); h[3] = hsum<C, R, 0, 3>(Ew, Ow);
    h[4] = hsum<C, R, 1, 0>(Ew, Ow); h[5] = hsum<C, R, 1, 1>(Ew, Ow);
    h[6] = hsum<C, R, 1, 2>(Ew, Ow); h[7] = hsum<C, R, 1, 3>(Ew, Ow);
}

struct TiledParams {
    const uint8_t *in;
    uint8_t *out;
    long long in_stride, out_stride;  // bytes per image
    int pitch, cpr;                   // bytes per row, 16-byte chunks per row
    int H, y0, y1;                    // band rows (clamp range), output rows [y0,y1)
    int ncols, nstrips;               // chunk columns per strip (<= 62), strips per row
    int TH, ntiles_y, ngroups;        // output rows per tile, row tiles per image, row groups per tile
    unsigned nblocks;
    int xcd;                          // blockIdx -> tile map: 0 identity, 1 XCD-contiguous, r >= 2 runs of r tiles dealt to the XCDs in turn
    int debug_copy;                   // ablation only: skip the arithmetic, store the staged centre chunk
    unsigned long long *xcd_times;    // diagnostics only (nullptr normally): per workgroup {end << 4 | XCC_ID, start}
    int tail;                         // RAG: bytes of the row's last chunk that exist (1..16); 16 otherwise
};

// 16 bytes at any address (global or LDS): the backend emits the full-width instruction and the
// hardware (unaligned access mode) splits it where it must.
struct __attribute__((packed, aligned(1))) Unaligned16 { u32x4 v; };
struct __attribute__((packed, aligned(1))) Unaligned8 { u32x2 v; };
struct __attribute__((packed, aligned(1))) Unaligned4 { uint32_t v; };
struct __attribute__((packed, aligned(1))) Unaligned2 { uint16_t v; };

// RAG output: a whole chunk as one unaligned 16-byte store, a row's partial last chunk as the 8/4/2/1-byte
// pieces of the n (< 16) bytes that exist — the byte after them belongs to the next row.
__device__ __forceinline__ void store16_write_through(uint8_t *q, u32x4 v);
__device__ __forceinline__ void store_chunk_ragged_wt(uint8_t *q, u32x4 v, int n);
template <bool WT = false>
__device__ __forceinline__ void store_chunk_ragged(uint8_t *q, u32x4 v, int n)
{
    if constexpr (WT) { store_chunk_ragged_wt(q, v, n); return; }
    if (n >= 16) { reinterpret_cast<Unaligned16 *>(q)->v = v; return; }
    if (n & 8) { u32x2 t; t.x = v.x; t.y = v.y; reinterpret_cast<Unaligned8 *>(q)->v = t; q += 8; v.x = v.z; v.y = v.w; }
    if (n & 4) { reinterpret_cast<Unaligned4 *>(q)->v = v.x; q += 4; v.x = v.y; }
    if (n & 2) { reinterpret_cast<Unaligned2 *>(q)->v = (uint16_t)v.x; q += 2; v.x >>= 16; }
    if (n & 1) *q = (uint8_t)v.x;
}

// ----------------------------------------------------------------------------------
// LDS-tiled vector kernel
// ----------------------------------------------------------------------------------
// blockIdx -> tile.  Blocks b and b+8 share an XCD (round-robin dispatch); give each XCD a contiguous run of the
// n tiles so tile-edge halo rows are L2 hits.  A bijection of [0, n).  Speed only.
__device__ __forceinline__ unsigned xcd_contiguous(unsigned L, unsigned n)
{
    const unsigned q = n >> 3, r = n & 7u, x = L & 7u, k = L >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + k;
}
// The same idea at a finer grain: runs of `run` consecutive tiles (tiles that share halo rows: an image, a few tile rows)
// are dealt to the XCDs in turn, so an XCD's tiles still find their neighbours' rows in its own L2 while its stream walks
// the WHOLE buffer instead of one eighth of it.  A bijection of [0, n): the last n mod 8*run tiles map to themselves.
__device__ __forceinline__ unsigned xcd_runs(unsigned L, unsigned n, unsigned run)
{
    const unsigned span = 8u * run, full = n - n % span;
    if (L >= full) return L;
    const unsigned x = L & 7u, k = L >> 3, j = k / run, o = k - j * run;
    return (j * 8u + x) * run + o;
}
__device__ __forceinline__ unsigned xcd_map(unsigned L, unsigned n, int mode)
{
    return mode == 0 ? L : mode == 1 ? xcd_contiguous(L, n) : xcd_runs(L, n, (unsigned)mode);
}

// 16-byte output store that writes through the (per-XCD, mutually non-coherent) L2 to memory: when its vmcnt has
// drained the bytes are visible device-wide without an L2 write-back fence.  Fused stream only.
__device__ __forceinline__ void store16_write_through(uint8_t *q, u32x4 v)
{
    // s_nop: a store of more than 8 bytes needs two wait states before a VALU may overwrite its data registers; the
    // compiler's hazard pass does not look inside inline asm
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(q), "v"(v) : "memory");
}

// The ragged store written through L2 (fused stream on rows that are not a multiple of 16 bytes): every piece carries sc1.
__device__ __forceinline__ void store_chunk_ragged_wt(uint8_t *q, u32x4 v, int n)
{
    if (n >= 16) { store16_write_through(q, v); return; }      // the hardware splits an unaligned 16-byte store where it must
    if (n & 8) {
        u32x2 t; t.x = v.x; t.y = v.y;
        asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(q), "v"(t) : "memory");
        q += 8; v.x = v.z; v.y = v.w;
    }
    if (n & 4) { asm volatile("global_store_dword %0, %1, off sc1" ::"v"(q), "v"(v.x) : "memory"); q += 4; v.x = v.y; }
    if (n & 2) { asm volatile("global_store_short %0, %1, off sc1" ::"v"(q), "v"(v.x) : "memory"); q += 2; v.x >>= 16; }
    if (n & 1) asm volatile("global_store_byte %0, %1, off sc1" ::"v"(q), "v"(v.x) : "memory");
}

// One workgroup's tile (tile number L of the launch).  Threads may return early; the only barrier is after staging.
// WT: outputs are written through L2 (store16_write_through).
struct NoHook { __device__ __forceinline__ void operator()() const {} };

// `after_staging` runs once the tile is in LDS (every global load AND every earlier store of this wave has been
// acknowledged: s_waitcnt vmcnt(0) + barrier) and before this tile's own stores are issued.
template <int C, int R, int RPG, bool DMA, bool SHFL, bool RAG, bool WT = false, int X = rowpass_default<R>, typename Hook = NoHook>
__device__ __forceinline__ void tiled_tile(const TiledParams &p, unsigned L, Hook after_staging = Hook{})
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int t = threadIdx.x, NT = blockDim.x;
    const int strip = (int)(L % (unsigned)p.nstrips);
    const unsigned t2 = L / (unsigned)p.nstrips;
    const int ty = (int)(t2 % (unsigned)p.ntiles_y);
    const int img = (int)(t2 / (unsigned)p.ntiles_y);

    const int ty0 = p.y0 + ty * p.TH;                 // first output row of the tile (band coords)
    const int rows_out = min(p.TH, p.y1 - ty0);
    const int x0c = strip * p.ncols;                  // first chunk column of the strip
    const int nc = min(p.ncols, p.cpr - x0c);
    const int cpr2 = nc + 2;                          // + one halo chunk each side (<= 64)
    const int nrows = rows_out + 2 * R;
    const uint8_t *img_in = p.in + (long long)img * p.in_stride;

    // ---- stage.  One wave-instruction moves `rpi` whole tile rows: lane = lr*cpr2 + cc, and the
    // LDS image of those rows is exactly the lanes in order (what LDS-DMA requires).  Rows clamp
    // to the band; halo columns outside the row are loaded from a clamped (don't-care) address.
    {
        const int lane = t & 63, wv = t >> 6, nwaves = NT >> 6;
        const int rpi = 64 / cpr2;                    // >= 1
        const int lr = lane / cpr2, cc = lane - lr * cpr2;
        // halo chunks that fall outside the image row are never read (the x-clamp is synthesised from the edge
        // chunk), so they are not loaded either: for full-row tiles that is both halo columns of every row
        const bool dead_halo = (cc == 0 && x0c == 0) || (cc == cpr2 - 1 && x0c + nc == p.cpr);
        const bool act = lr < rpi && !dead_halo;
        const unsigned col_off = (unsigned)min(max(x0c + cc - 1, 0), p.cpr - 1) * 16u;
        for (int u = wv; u * rpi < nrows; u += nwaves) {
            const int row = u * rpi + lr;
            if (act && row < nrows) {
                const int sr = min(max(ty0 - R + row, 0), p.H - 1);
                const uint8_t *g = img_in + ((unsigned)sr * (unsigned)p.pitch + col_off);
                if constexpr (RAG) {
                    const uint8_t *rowp = img_in + (size_t)sr * (size_t)p.pitch;
                    uint8_t *dst = lds + (size_t)(u * rpi * cpr2 + lane) * 16u;
                    if (x0c + cc - 1 == p.cpr - 1) {
                        // the row's last chunk: the 16 bytes that END at the row end, at their row-relative place;
                        // behind them the right-edge clamp (byte pitch+k = byte pitch-C+(k mod C): copies of the
                        // last pixel, all taken from the last dword), as far as any window reaches: the rest of
                        // this chunk, plus 8 bytes of the right halo chunk when this lane is a column of the strip
                        const u32x4 v = reinterpret_cast<const Unaligned16 *>(rowp + p.pitch - 16)->v;
                        reinterpret_cast<Unaligned16 *>(dst - (16 - p.tail))->v = v;
                        const int len = (16 - p.tail) + (cc == cpr2 - 1 ? 0 : 8);
                        uint8_t *pd = dst + p.tail;
                        const uint32_t fill[6] = {clamp_right<C, 0>(v.z, v.w), clamp_right<C, 1>(v.z, v.w), clamp_right<C, 2>(v.z, v.w),
                                                  clamp_right<C, 3>(v.z, v.w), clamp_right<C, 4>(v.z, v.w), clamp_right<C, 5>(v.z, v.w)};
#pragma unroll
                        for (int q = 0; q < 6; q++) {
                            if (4 * q + 4 <= len) reinterpret_cast<Unaligned4 *>(pd + 4 * q)->v = fill[q];
                            else if (4 * q < len)      // the last 1-3 bytes one by one: nothing is written past `len`
                                for (int b = 0; b < len - 4 * q; b++) pd[4 * q + b] = (uint8_t)(fill[q] >> (8 * b));
                        }
                    } else if constexpr (DMA) {
                        // LDS-DMA takes the unaligned global address as it is; the LDS side stays lane-ordered
                        uint8_t *base = lds + (size_t)(u * rpi * cpr2) * 16u;
                        __builtin_amdgcn_global_load_lds(
                            (const void __attribute__((address_space(1))) *)(rowp + col_off),
                            (void __attribute__((address_space(3))) *)base, 16, 0, 0);
                    } else {
                        *reinterpret_cast<u32x4 *>(dst) = reinterpret_cast<const Unaligned16 *>(rowp + col_off)->v;
                    }
                } else if constexpr (DMA) {
                    uint8_t *base = lds + (size_t)(u * rpi * cpr2) * 16u;    // wave-uniform; + lane*16 by HW
                    __builtin_amdgcn_global_load_lds(
                        (const void __attribute__((address_space(1))) *)g,
                        (void __attribute__((address_space(3))) *)base, 16, 0, 0);   // default cache policy: nt (aux 2) measured 0-5 % slower
                } else {
                    *reinterpret_cast<uint4 *>(lds + (size_t)(u * rpi * cpr2 + lane) * 16u) =
                        *reinterpret_cast<const uint4 *>(g);
                }
            }
        }
        if constexpr (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    after_staging();

    // ---- compute: thread = (chunk column col, row group grp); RPG rows, sliding window.
    const int grp = t / nc, col = t - grp * nc;
    const int r0 = grp * RPG;
    if (grp >= p.ngroups || r0 >= rows_out) return;   // no barrier below
    const bool at_start = (x0c + col) == 0;
    const bool at_end = !RAG && (x0c + col) == p.cpr - 1;      // RAG: the right clamp bytes are in LDS already
    const bool any_edge = __builtin_amdgcn_ballot_w64(at_start || at_end) != 0ull;
    const int lrow = cpr2 * 16;
    const uint8_t *lp = lds + ((size_t)r0 * cpr2 + (col + 1)) * 16u;
    uint8_t *op = p.out + (long long)img * p.out_stride +
                  (size_t)(ty0 - p.y0 + r0) * (size_t)p.pitch + (size_t)(x0c + col) * 16u;

    if (p.debug_copy) {   // ablation: the load -> LDS -> store skeleton without the stencil arithmetic
#pragma unroll
        for (int r = 0; r < RPG; r++)
            if (r0 + r < rows_out) {
                const u32x4 v = *reinterpret_cast<const u32x4 *>(lp + (r + R) * lrow);
                if constexpr (RAG) store_chunk_ragged<WT>(op + (size_t)r * (size_t)p.pitch, v, (x0c + col) == p.cpr - 1 ? p.tail : 16);
                else *reinterpret_cast<u32x4 *>(op + (size_t)r * (size_t)p.pitch) = v;
            }
        return;
    }

    constexpr int WIN = 2 * R + 1;
    uint32_t hw[WIN][8];
    // shuffle row pass: which lanes cannot get a neighbour's bytes from the adjacent lane
    const bool lds_left = (t & 63) == 0 || col == 0, lds_right = (t & 63) == 63 || col == nc - 1;
#pragma unroll
    for (int k = 0; k < 2 * R; k++) hrow<C, R, SHFL, X>(lp + k * lrow, any_edge, at_start, at_end, lds_left, lds_right, hw[k]);

#pragma unroll
    for (int r = 0; r < RPG; r++) {
        hrow<C, R, SHFL, X>(lp + (r + 2 * R) * lrow, any_edge, at_start, at_end, lds_left, lds_right, hw[(r + 2 * R) % WIN]);
        uint32_t o[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if constexpr (R == 1) {
                // s = h0 + 2 h1 + h2 (<= 4080 per field).  16*s (<= 65280) still fits the field and puts the
                // output byte s>>4 in the field's HIGH byte: v_lshl_add + v_add_lshl per parity, then one
                // v_perm re-interleaves even/odd high bytes into the output dword.
                const uint32_t se = (((hw[(r + 1) % WIN][i] << 1) + hw[r % WIN][i]) + hw[(r + 2) % WIN][i]) << 4;
                const uint32_t so = (((hw[(r + 1) % WIN][4 + i] << 1) + hw[r % WIN][4 + i]) + hw[(r + 2) % WIN][4 + i]) << 4;
                o[i] = __builtin_amdgcn_perm(so, se, 0x07030501u);
            } else {
                // s = h0 + 4 h1 + 6 h2 + 4 h3 + h4 (<= 65280 per field); byte = s >> 8 = high byte
                const uint32_t se = mad6(hw[(r + 2) % WIN][i],
                                         ((hw[(r + 1) % WIN][i] + hw[(r + 3) % WIN][i]) << 2) +
                                             (hw[r % WIN][i] + hw[(r + 4) % WIN][i]));
                const uint32_t so = mad6(hw[(r + 2) % WIN][4 + i],
                                         ((hw[(r + 1) % WIN][4 + i] + hw[(r + 3) % WIN][4 + i]) << 2) +
                                             (hw[r % WIN][4 + i] + hw[(r + 4) % WIN][4 + i]));
                o[i] = __builtin_amdgcn_perm(so, se, 0x07030501u);
            }
        }
        if (r0 + r < rows_out) {
            u32x4 v; v.x = o[0]; v.y = o[1]; v.z = o[2]; v.w = o[3];
            if constexpr (RAG) store_chunk_ragged<WT>(op + (size_t)r * (size_t)p.pitch, v, (x0c + col) == p.cpr - 1 ? p.tail : 16);
            else if constexpr (WT) store16_write_through(op + (size_t)r * (size_t)p.pitch, v);
            else *reinterpret_cast<u32x4 *>(op + (size_t)r * (size_t)p.pitch) = v;
        }
    }
}

// Diagnostics: which XCD does a launch wait for?  Every workgroup stores (end time << 4 | XCC_ID) and its start time
// (s_memrealtime, 100 MHz) in slots of its own — plain stores, no atomics, so the launch being examined is not slowed.
// Off (nullptr) unless mi_blur_set_option("debug_xcd_times", 1); the host folds the slots per XCD.
constexpr unsigned XCD_DEBUG_SLOTS = 1u << 20;
__device__ __forceinline__ void xcd_time_mark(unsigned long long *t, unsigned long long t0)
{
    if (threadIdx.x == 0 && blockIdx.x < XCD_DEBUG_SLOTS) {
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        t[2u * blockIdx.x] = ((unsigned long long)__builtin_amdgcn_s_memrealtime() << 4) | (id & 7u);
        t[2u * blockIdx.x + 1] = t0;
    }
}

template <int C, int R, int RPG, bool DMA, bool SHFL = false, bool RAG = false>
__global__ __launch_bounds__(256) void blur_tiled_kernel(const TiledParams p)
{
    const unsigned long long t0 = p.xcd_times ? __builtin_amdgcn_s_memrealtime() : 0ull;
    tiled_tile<C, R, RPG, DMA, SHFL, RAG>(p, xcd_map(blockIdx.x, p.nblocks, p.xcd));
    if (p.xcd_times) xcd_time_mark(p.xcd_times, t0);
}

// Capped-grid form: at most `gridDim.x` workgroups walk the launch's tiles (block b takes tiles b, b + G, b + 2G ...).
// For launches whose buffers are PINNED HOST memory (zero-copy submits): the host link needs only ~100 KB in flight to
// run at full rate, and a launch that puts every tile on the chip at once does all of its reads and then all of its
// writes — half duplex.  With a few hundred resident workgroups each on its own read -> compute -> write cycle the two
// directions of the link stay busy together.
template <int C, int R, int RPG>
__global__ __launch_bounds__(256) void blur_tiled_loop_kernel(const TiledParams p)
{
    for (unsigned L = blockIdx.x; L < p.nblocks; L += gridDim.x) {
        tiled_tile<C, R, RPG, true, false, false>(p, L);
        __syncthreads();                      // the next tile is staged into the same LDS
    }
}

// the OTHER row-pass form of the aligned LDS-DMA tiled kernel, for A/B runs (mi_blur_set_option("experiment", 1); C = 3 only)
template <int C, int R, int RPG, int X>
__global__ __launch_bounds__(256) void blur_tiled_x_kernel(const TiledParams p)
{
    tiled_tile<C, R, RPG, true, false, false, false, X>(p, xcd_map(blockIdx.x, p.nblocks, p.xcd));
}

// Fused stream: ONE dispatch for a whole pass of the resident stream, with the batch kept as the unit of completion.
// Blocks are ordered window by window — a window is 8 consecutive batches ("fused_window"), XCD-contiguous inside it, so
// each XCD blurs one whole batch (35 whole images) per window: 2 % faster than batch-by-batch order, where every batch is
// cut into eight 35-tile pieces (profiles/r02_fused_window.txt) — and every tile counts into its own batch; every wave, once its own output
// stores have drained, meets the rest of its block at a barrier; one thread then adds 1 to the batch's counter with a
// fire-and-forget atomic (one of 8 per batch).  The host reads the counters: a batch is complete when they sum to its
// number of blocks.  A consumer
// therefore sees batches complete in stream order while the GPU never pays a per-batch dispatch (4 us floor + ~3.5 us
// of dispatch processing each).
struct FusedParams {
    unsigned *count;          // device, `kcount` counters per batch (spread by block number: less contention), zeroed before the launch
    unsigned kcount;          // 8 .. 256, a power of two: an add on a hot word costs ~200 ns of that word's time, so a batch of
                              // thousands of tiles gets more words (8 serve the 140 tiles of a 35-image batch of 256x256 frames)
    unsigned tiles_per_batch; // batch_images * tiles per image
    int release;              // 1 = release-ordered completion add (agent scope): the architectural form, ~6x slower
    unsigned window;          // batches per window of the blockIdx -> tile map (>= 1)
    // dynamic tail (blur_fused_tail_kernel): the pass's last `ntail` tiles are not mapped to blocks; `nextra` extra blocks at the
    // end of the grid draw them from *tail_ctr one by one and leave when none is left
    unsigned *tail_ctr;
    unsigned nstatic, ntail, nextra;
};

template <int C, int R, int RPG>
__global__ __launch_bounds__(256) void blur_fused_kernel(const TiledParams p, const FusedParams f)
{
    // Blocks are taken window by window; a window is `window` consecutive batches and its tiles are dealt so that each XCD
    // gets one contiguous eighth of it (with 8 batches per window: one whole batch per XCD, i.e. 35 whole images whose
    // tile-edge rows stay in that XCD's L2).  Every tile is counted into ITS batch, so the batch stays the unit of
    // completion; the batches of a window finish at about the same time, windows in stream order.
    const unsigned tpw = f.tiles_per_batch * f.window;
    const unsigned wnd = blockIdx.x / tpw, base = wnd * tpw;
    const unsigned nw = min(tpw, p.nblocks - base);                          // the last window may be short
    const unsigned w = blockIdx.x - base;
    const unsigned tile = base + (nw >= 16 ? xcd_map(w, nw, p.xcd) : w);
    const unsigned b = tile / f.tiles_per_batch;
    tiled_tile<C, R, RPG, true, false, false, true>(p, tile);
    // Outputs were written THROUGH L2 (the XCDs' L2s are not coherent with each other), so once this wave's stores
    // have drained they are in memory.  No device-scope release fence (an L2 write-back per call: 15x the whole pass when
    // every block does one), no returning atomic (its round trip would keep the block's LDS allocated).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                       // every wave of the block has drained (one atomic per block: the counters are hot spots)
    if (threadIdx.x == 0) {
        if (f.release) __hip_atomic_fetch_add(&f.count[b * f.kcount + ((w ^ (w >> 3)) & (f.kcount - 1u))], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        else __hip_atomic_fetch_add(&f.count[b * f.kcount + ((w ^ (w >> 3)) & (f.kcount - 1u))], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// The fused stream with a DYNAMIC TAIL.  The hardware deals workgroups to the eight XCDs in strict rotation, so with one tile per
// workgroup every XCD blurs exactly an eighth of the pass, and the pass ends when the slowest XCD does (3-5 % after the fastest,
// profiles/r02_xcd_finish_times.txt).  Here the last `ntail` tiles of the pass belong to no block: the grid ends in `nextra`
// (> ntail) extra blocks, which every XCD reaches when it is through its share; each draws a ticket and blurs that tail tile, or
// leaves if the tail is gone.  An XCD that runs ahead so takes more of the tail, one that lags takes less.  Every extra
// block draws exactly once, so the one that draws ticket nextra-1 knows it is the last and resets the counter for the next pass.
// RAG: rows that are not a multiple of 16 bytes (the ragged form of the tile code, its stores written through as well); those
// passes always take this kernel, with an empty tail when they are short.
template <int C, int R, int RPG, bool RAG = false>
__global__ __launch_bounds__(256) void blur_fused_tail_kernel(const TiledParams p, const FusedParams f)
{
    // (one call site of the tile code, reached by both kinds of block: with the tile code inside a ticket loop the same
    // compiler schedules the whole kernel 2.6x slower)
    unsigned tile, w;
    if (blockIdx.x >= f.nstatic) {
        __shared__ unsigned s_ticket;
        if (threadIdx.x == 0) {
            const unsigned t = __hip_atomic_fetch_add(f.tail_ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t == f.nextra - 1u) __hip_atomic_store(f.tail_ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_ticket = t;
        }
        __syncthreads();
        const unsigned t = __builtin_amdgcn_readfirstlane(s_ticket);
        if (t >= f.ntail) return;
        tile = f.nstatic + t;
        w = t;
    } else {
        const unsigned tpw = f.tiles_per_batch * f.window;
        const unsigned wnd = blockIdx.x / tpw, base = wnd * tpw;
        const unsigned nw = min(tpw, f.nstatic - base);
        w = blockIdx.x - base;
        tile = base + (nw >= 16 ? xcd_map(w, nw, p.xcd) : w);
    }
    const unsigned b = tile / f.tiles_per_batch;
    tiled_tile<C, R, RPG, true, false, RAG, true>(p, tile);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        if (f.release) __hip_atomic_fetch_add(&f.count[b * f.kcount + ((w ^ (w >> 3)) & (f.kcount - 1u))], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        else __hip_atomic_fetch_add(&f.count[b * f.kcount + ((w ^ (w >> 3)) & (f.kcount - 1u))], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// One wave that follows a fused pass for the host: batch n is complete when its eight counters sum to (its blocks) x
// per_block; the wave then publishes (pass_seq, n + 1) in pinned host memory.  Lanes 0-7 read one counter each.  It ends with
// the pass (every batch complete) or after ZC_HARD_TICKS — a pass that faulted never completes its counters.
__global__ __launch_bounds__(64) void fused_watch_kernel(const unsigned *count, unsigned n_batches, unsigned tiles_per_batch, unsigned total_blocks,
                                                         unsigned per_block, unsigned long long *host_word, unsigned pass_seq, unsigned kcount)
{
    const unsigned lane = threadIdx.x;
    const unsigned long long t0 = wall_clock64();
    unsigned n = 0;
    while (n < n_batches) {
        const unsigned first = n * tiles_per_batch;
        const unsigned want = min(tiles_per_batch, total_blocks - first) * per_block;
        unsigned v = 0;
        for (unsigned k = lane; k < kcount; k += 64u) v += __hip_atomic_load(&count[kcount * n + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8); v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
        v = __builtin_amdgcn_readfirstlane(v);
        if ((int)(v - want) >= 0) {      // >=: the next pass (same counters, counting on) may already be adding to this batch
            n++;
            // skip ahead over batches that are complete already before telling the host (one store per advance is plenty)
            continue;
        }
        if (lane == 0) {
            const unsigned long long cur = ((unsigned long long)pass_seq << 32) | n;
            if (__hip_atomic_load(host_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != cur)
                __hip_atomic_store(host_word, cur, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        if (wall_clock64() - t0 > ZC_HARD_TICKS) return;
        __builtin_amdgcn_s_sleep(2);
    }
    if (lane == 0) __hip_atomic_store(host_word, ((unsigned long long)pass_seq << 32) | n, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ----------------------------------------------------------------------------------
// Zero-copy batch server (see blur_launch.h): batch after batch of pinned host frames through ONE dispatch.
//
// Tiles are handed out by ONE ticket counter that runs through the batches (global tile g belongs to the batch whose
// [tile_first, tile_first + n_tiles) holds it): a worker that finishes early simply draws the next ticket, also across
// a batch boundary.  Over the host link a tile takes anywhere between 10 and 100 us depending on what is queued in
// front of its reads, so a fixed share of tiles per workgroup (the per-batch launch: tile L to workgroup L mod n) leaves
// every batch waiting for its unluckiest workgroup — that, not the dispatch itself, is what a batch boundary costs
// (profiles/r03_e2e_timeline.md).
// ----------------------------------------------------------------------------------
static_assert(sizeof(TiledParams) <= ZC_PARAM_WORDS * 4 && sizeof(TiledParams) % 4 == 0, "TiledParams must fit a ZcBatch");
static_assert(sizeof(ZcBatch) / 4 <= 64, "the poller copies a descriptor with one wave-instruction");

template <int C, int R, int RPG, bool RAG>
__global__ __launch_bounds__(256) void blur_server_kernel(ZcHostCtl *ctl, ZcDevCtl *dev, unsigned seq, unsigned budget, unsigned idle_ticks,
                                                          unsigned long long *trace, int fixed_share)
{
    const unsigned NW = gridDim.x - 1;
    // where the previous server stopped (its dispatch is complete: same stream).  This server's poller writes the OTHER words.
    const unsigned first = dev->next[seq & 1u];

    if (blockIdx.x == NW) {
        // ---- poller: one wave.  Host `tail` -> descriptors copied to device memory -> `avail`; tiles_done -> host `done`.
        // Every exit path tells the workers where the server stops, and it leaves only when every batch it took is signalled.
        const unsigned lane = threadIdx.x;
        if (lane >= 64) return;
        __shared__ unsigned s_ntiles[ZC_RING];
        unsigned g = dev->gnext[seq & 1u];                                   // global tile number of batch `first`
        if (lane == 0) {
            __hip_atomic_store(&dev->ticket, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&dev->start_seq, seq + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
        unsigned k = first, c = first;                                       // published to the workers / signalled to the host
        const unsigned limit = first + budget;
        bool stopping = false;
        unsigned long long t_idle = wall_clock64();
        for (;;) {
            if (!stopping) {
                unsigned tail = 0;
                if (lane == 0) tail = __hip_atomic_load(&ctl->tail, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
                tail = __builtin_amdgcn_readfirstlane(tail);
                const unsigned target = (int)(tail - limit) > 0 ? limit : tail;
                const bool progressed = (int)(target - k) > 0;
                while ((int)(target - k) > 0) {
                    const unsigned slot = k % ZC_RING;
                    unsigned word = 0;
                    if (lane < sizeof(ZcBatch) / 4) {
                        word = __hip_atomic_load(reinterpret_cast<unsigned *>(&ctl->batch[slot]) + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        __hip_atomic_store(reinterpret_cast<unsigned *>(&dev->batch[slot]) + lane, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    const unsigned nt = __builtin_amdgcn_readlane(word, ZC_PARAM_WORDS + 1);     // ZcBatch::n_tiles
                    if (lane == 0) {
                        s_ntiles[slot] = nt;
                        __hip_atomic_store(&ctl->t_begin[slot], wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    }
                    g += nt;
                    k++;
                }
                if (progressed) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every lane's descriptor words are written
                    if (lane == 0) __hip_atomic_store(&dev->avail, k, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                    t_idle = wall_clock64();
                }
                unsigned quit = 0;
                if (lane == 0) quit = __hip_atomic_load(&ctl->quit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                quit = __builtin_amdgcn_readfirstlane(quit);
                if (k == limit || quit || (c == k && wall_clock64() - t_idle > (unsigned long long)idle_ticks)) {
                    stopping = true;                                         // budget spent / told to leave / nothing to do for a while
                    if (lane == 0) {
                        dev->next[(seq + 1u) & 1u] = k;                      // read by the next server (after this dispatch has ended)
                        dev->gnext[(seq + 1u) & 1u] = g;
                        __hip_atomic_store(&dev->stop_at, k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(&dev->stop_tile, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(&dev->stop_seq, seq + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
            // batches complete in order: every tile of batch c has counted itself (after its stores were acknowledged)
            while ((int)(k - c) > 0) {
                const unsigned slot = c % ZC_RING;
                unsigned d = 0;
                if (lane == 0) d = __hip_atomic_load(&dev->tiles_done[slot], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                d = __builtin_amdgcn_readfirstlane(d);
                if (d != s_ntiles[slot]) break;
                if (lane == 0) {
                    __hip_atomic_store(&dev->tiles_done[slot], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // the ring slot comes round again
                    __hip_atomic_store(&ctl->t_end[slot], wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    __hip_atomic_store(&ctl->done[slot], c + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                }
                c++;
                t_idle = wall_clock64();
            }
            if (stopping && c == k) break;
            if (wall_clock64() - t_idle > ZC_HARD_TICKS) {                   // tiles that never complete: give up loudly rather than spin
                if (lane == 0) {
                    __hip_atomic_store(&ctl->error, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                    if (!stopping) {
                        dev->next[(seq + 1u) & 1u] = k; dev->gnext[(seq + 1u) & 1u] = g;
                        __hip_atomic_store(&dev->stop_at, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(&dev->stop_seq, seq + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                break;
            }
            __builtin_amdgcn_s_sleep(8);
        }
        if (lane == 0) __hip_atomic_store(&ctl->servers_done, seq + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }

    // ---- workers
    __shared__ unsigned s_words[sizeof(ZcBatch) / 4];
    const unsigned w = blockIdx.x;
    __shared__ unsigned s_ctl[4];                       // [0] go / stop, [1] first ticket, [2], [3] next ticket of even / odd tiles
    if (threadIdx.x == 0) {
        const unsigned long long t_gate = wall_clock64();
        unsigned ok = 1u;
        while (__hip_atomic_load(&dev->start_seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != seq + 1u) {
            if (wall_clock64() - t_gate > ZC_HARD_TICKS) { ok = 0u; __hip_atomic_store(&ctl->error, 2u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); break; }
            __builtin_amdgcn_s_sleep(2);
        }
        s_ctl[0] = ok;
    }
    __syncthreads();
    if (!s_ctl[0]) return;
    __syncthreads();
    union { TiledParams p; unsigned words[sizeof(TiledParams) / 4]; } u;
    unsigned k = first, have = 0u, tile_first = 0u, n_tiles = 0u;
    unsigned pending = ~0u;                             // ring slot of the tile whose stores may still be in flight
    // count a finished tile into its batch: called when its stores have been acknowledged
    auto count_pending = [&]() {
        if (pending != ~0u && threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");                    // system scope: the tile's bytes before the count
            __hip_atomic_fetch_add(&dev->tiles_done[pending], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
        pending = ~0u;
    };
    auto flush_pending = [&]() {                        // no next tile to hide behind: wait for the stores here
        if (pending != ~0u) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            count_pending();
        }
    };
    // fixed_share (A/B only, "zero_copy_tickets" 0): worker w takes global tiles w, w + NW, ... — the share a per-batch launch
    // gives a workgroup — instead of drawing tickets
    if (threadIdx.x == 0)
        s_ctl[1] = fixed_share ? dev->gnext[seq & 1u] + w : __hip_atomic_fetch_add(&dev->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    unsigned t = s_ctl[1];
    unsigned long long *tr = nullptr;
    for (unsigned it = 0;; it++) {
        // ---- the batch ticket t belongs to
        for (;;) {
            if (!have) {
                // is batch k published?  0 = the server stops before it, 1 = yes, 2 = not yet (and not stopping)
                auto look = [&]() -> unsigned {
                    const unsigned a = __hip_atomic_load(&dev->avail, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                    if ((int)(a - k) > 0) return 1u;
                    if (__hip_atomic_load(&dev->stop_seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == seq + 1u &&
                        (int)(__hip_atomic_load(&dev->stop_at, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - k) <= 0) return 0u;
                    return 2u;
                };
                if (threadIdx.x == 0) s_ctl[0] = look();
                __syncthreads();
                unsigned go = s_ctl[0];
                __syncthreads();
                if (go != 1u) {
                    flush_pending();                 // waiting or leaving: do not sit on a finished tile meanwhile
                    if (go == 2u) {
                        if (threadIdx.x == 0) {
                            unsigned g2;
                            const unsigned long long t_w = wall_clock64();
                            while ((g2 = look()) == 2u) {
                                if (wall_clock64() - t_w > ZC_HARD_TICKS) { g2 = 0u; __hip_atomic_store(&ctl->error, 3u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); break; }
                                __builtin_amdgcn_s_sleep(4);
                            }
                            s_ctl[0] = g2;
                        }
                        __syncthreads();
                        go = s_ctl[0];
                        __syncthreads();
                    }
                    if (!go) return;                 // the server stops before batch k
                }
                // descriptor of batch k: the poller copied it into device memory before it released `avail`
                if (threadIdx.x < sizeof(ZcBatch) / 4)
                    s_words[threadIdx.x] = __hip_atomic_load(reinterpret_cast<unsigned *>(&dev->batch[k % ZC_RING]) + threadIdx.x, __ATOMIC_RELAXED,
                                                             __HIP_MEMORY_SCOPE_AGENT);
                __syncthreads();
#pragma unroll
                for (unsigned i = 0; i < sizeof(TiledParams) / 4; i++) u.words[i] = __builtin_amdgcn_readfirstlane(s_words[i]);   // wave-uniform: SGPRs
                tile_first = __builtin_amdgcn_readfirstlane(s_words[ZC_PARAM_WORDS]);
                n_tiles = __builtin_amdgcn_readfirstlane(s_words[ZC_PARAM_WORDS + 1]);
                have = 1u;
                // the host wrote the frames before it published `tail`; the poller acquired that at system scope and released
                // `avail` at agent scope: take the same view before reading host memory
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
                __syncthreads();
            }
            if ((int)(t - (tile_first + n_tiles)) >= 0) { k++; have = 0u; continue; }     // the ticket lies beyond this batch
            break;
        }
        // diagnostics (trace != nullptr): per worker and batch (the context's first ZC_TRACE_BATCHES only) — first tile taken at,
        // tiles taken, device-clock ticks inside tiles, last tile finished at
        tr = (trace && k < ZC_TRACE_BATCHES && threadIdx.x == 0) ? trace + ((size_t)k * NW + w) * 5u : nullptr;
        if (tr) { if (tr[0] == 0) tr[0] = wall_clock64(); tr[1] += 1; }
        // ---- draw the next ticket now (its round trip hides behind the tile), then the tile
        // (two LDS words, taken in turn: a wave that is already drawing for the next tile must not overwrite the word a slower
        // wave of the block has not read yet — the barriers of one whole tile lie between two writes of the same word)
        if (threadIdx.x == 0)
            s_ctl[2u + (it & 1u)] = fixed_share ? t + NW : __hip_atomic_fetch_add(&dev->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long t_a = trace ? wall_clock64() : 0ull;
        tiled_tile<C, R, RPG, true, false, RAG, false, rowpass_default<R>>(u.p, t - tile_first, count_pending);
        pending = k % ZC_RING;
        __syncthreads();                              // the next tile is staged into the same LDS; the next ticket is in LDS
        if (tr) { tr[2] += wall_clock64() - t_a; tr[3] = wall_clock64(); }
        t = s_ctl[2u + (it & 1u)];
    }
}

// ----------------------------------------------------------------------------------
// streaming kernel: wave-private LDS row ring, no barrier — every wave is an independent stream
// ----------------------------------------------------------------------------------
// Lanes are consecutive 16-byte chunks of the flattened (image, band, chunk-column) space, so all 64 lanes
// are busy whatever the row length.  A wave marches down its band of BH output rows.  Input rows arrive by
// LDS-DMA (global_load_lds_dwordx4, 16 B/lane, coalesced) into a ring of STREAM_D row slots in the wave's
// own LDS region, STREAM_P rows ahead of the row being consumed; a second 2-lane DMA brings the two
// 16-byte chunks beyond the wave's span (the 1-pixel halo) for lanes 0 and 63.  Consuming a row is the
// same 8+16+8-byte LDS read and even/odd SWAR horizontal pass as the tiled kernel, with a 2R+1-row
// sliding window of sums in registers.  No barrier: a wave only reads what it loaded itself, ordered by
// its own counted s_waitcnt vmcnt(N) (vmcnt is in issue order; every step issues exactly 2 DMAs + 1
// store — a store that is not wanted gets an out-of-range buffer offset and is dropped by the range check).
// The band's 2R halo rows are re-read through L2 (2R/BH extra L2->CU traffic).
constexpr int STREAM_P = 6;                    // rows in flight ahead of the consumer
constexpr int STREAM_D = 8;                    // ring slots: a slot is refilled two steps after it was read
constexpr int STREAM_ROWB = 32 + 64 * 16;      // [left halo chunk][right halo chunk][64 chunks]

struct StreamParams {
    const uint8_t *in;
    uint8_t *out;
    long long in_stride, out_stride;
    long long total;                  // n_images * nbands * cpr lanes of work
    long long total_images;
    int pitch, cpr;
    int H, y0, y1;
    int BH, nbands;
    unsigned nblocks;
    int xcd;
    int updown;                       // odd bands march upwards
};

struct StreamLane { unsigned col, band; long long img; };
__device__ __forceinline__ StreamLane stream_decode(unsigned f, const StreamParams &p)
{
    StreamLane l;
    l.col = f % (unsigned)p.cpr;
    const unsigned t2 = f / (unsigned)p.cpr;
    l.band = t2 % (unsigned)p.nbands;
    l.img = (long long)(t2 / (unsigned)p.nbands);
    return l;
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int C, int R>
__global__ __launch_bounds__(256) void blur_stream_kernel(const StreamParams p)
{
    constexpr int WIN = 2 * R + 1, P = STREAM_P, D = STREAM_D, ROWB = STREAM_ROWB;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint8_t *ring = lds + wave * (D * ROWB);
    const unsigned ring_off = (unsigned)(unsigned long long)((__attribute__((address_space(3))) uint8_t *)ring);   // LDS byte address

    unsigned B = blockIdx.x;
    if (p.xcd) {
        const unsigned n = p.nblocks, q = n >> 3, r = n & 7u, x = B & 7u, k = B >> 3;
        B = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + k;
    }
    const unsigned last = (unsigned)p.total - 1u;                       // host keeps total < 2^31
    const unsigned f0 = (B * 4u + (unsigned)wave) * 64u;
    const unsigned f_raw = f0 + (unsigned)lane;
    const bool valid = f_raw <= last;
    const unsigned f = valid ? f_raw : last;                            // dead lanes still run on valid addresses; no stores
    const StreamLane me = stream_decode(f, p);
    // lanes 0 / 1 also fetch the chunk left of the wave's span / right of it (flat neighbours; unused at row edges)
    const unsigned fe = lane == 0 ? (f0 > 0 ? f0 - 1u : 0u) : min(f0 + 64u, last);
    const StreamLane he = stream_decode(fe, p);

    const int row0 = p.y0 + (int)me.band * p.BH, row0e = p.y0 + (int)he.band * p.BH;
    const int rows_out = valid ? min(p.BH, p.y1 - row0) : 0;
    // Marching direction.  The vertical taps are symmetric, so a band can be walked bottom-up with the same arithmetic.
    // With `updown`, odd bands go up.  The seam between band 2k (down) and band 2k+1 (up) is then read by both at the END of
    // their march, the seam between band 2k+1 (up) and band 2k+2 (down) by both at the START: the two readers of a seam's
    // rows come at the same time and the second finds them in L2 instead of fetching them from HBM again.
    const int bh_me = min(p.BH, p.y1 - row0), bh_e = min(p.BH, p.y1 - row0e);
    const bool up = p.updown && (me.band & 1u), up_e = p.updown && (he.band & 1u);
    const int rbase = up ? row0 + bh_me - 1 + R : row0 - R, rstep = up ? -1 : 1;
    const int rbase_e = up_e ? row0e + bh_e - 1 + R : row0e - R, rstep_e = up_e ? -1 : 1;
    const bool at_start = me.col == 0, at_end = (int)me.col == p.cpr - 1;
    const bool any_edge = __builtin_amdgcn_ballot_w64(at_start || at_end) != 0ull;
    const uint8_t *gsrc = p.in + me.img * p.in_stride + (size_t)me.col * 16u;
    const uint8_t *gsrc_e = p.in + he.img * p.in_stride + (size_t)he.col * 16u;

    const long long img0 = __builtin_amdgcn_readfirstlane((int)me.img);
    const long long out_left = p.out_stride * (p.total_images - img0);
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(
        p.out + img0 * p.out_stride, 0, (int)(unsigned)(out_left > 0xffffffffLL ? 0xffffffffLL : out_left), 0x00020000);
    const unsigned OOB = 0xffffffffu;
    const unsigned dst_off = (unsigned)((me.img - img0) * p.out_stride) + (unsigned)(row0 - p.y0) * (unsigned)p.pitch + me.col * 16u;

    // per-lane LDS offsets inside a ring slot: centre chunk, the 8 bytes before it, the 8 bytes after it
    const int oc = 32 + lane * 16;
    const int ol = lane == 0 ? 8 : oc - 8;           // lane 0: tail of the left halo chunk
    const int orr = lane == 63 ? 16 : oc + 16;       // lane 63: head of the right halo chunk

    auto issue_row = [&](int j) {                    // j-th input row of the band (band row row0 - R + j), clamped to the band
        uint8_t *slot = ring + (j % D) * ROWB;       // wave-uniform
        const uint8_t *g = gsrc + (size_t)min(max(rbase + rstep * j, 0), p.H - 1) * (size_t)p.pitch;
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)g,
                                         (void __attribute__((address_space(3))) *)(slot + 32), 16, 0, 0);
        if (lane < 2) {
            const uint8_t *ge = gsrc_e + (size_t)min(max(rbase_e + rstep_e * j, 0), p.H - 1) * (size_t)p.pitch;
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)ge,
                                             (void __attribute__((address_space(3))) *)slot, 16, 0, 0);
        }
    };
#pragma unroll
    for (int j = 0; j < P; j++) issue_row(j);

    uint32_t hw[WIN][8] = {};
    const int niter = (p.BH + 2 * R + WIN - 1) / WIN;        // wave-uniform; short last bands only drop stores
    for (int k = 0; k < niter; k++) {
#pragma unroll
        for (int m = 0; m < WIN; m++) {
            const int j = k * WIN + m;
            // DMAs/stores younger than row j's: prologue rows only count 2 each, steady state 3 per step
            if (j < P) wait_vmcnt<2 * (P - 1)>(); else wait_vmcnt<3 * P - 2>();
            // The row reads are inline asm: hipcc orders every ds_read it can see behind ALL in-flight LDS-DMA
            // with s_waitcnt vmcnt(0), which would serialise the ring.  The statement returns with the data
            // landed (its own lgkmcnt(0)); the counted vmcnt above is what orders it behind row j's DMA.
            const unsigned soff = ring_off + (unsigned)(j % D) * (unsigned)ROWB;
            u32x2 a, b;
            u32x4 c;
            asm volatile("ds_read_b64 %0, %3\n\tds_read_b128 %1, %4\n\tds_read_b64 %2, %5\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(a), "=&v"(c), "=&v"(b)
                         : "v"(soff + (unsigned)ol), "v"(soff + (unsigned)oc), "v"(soff + (unsigned)orr)
                         : "memory");
            __builtin_amdgcn_sched_barrier(0);
            uint32_t w[8];
            w[0] = a.x; w[1] = a.y; w[2] = c.x; w[3] = c.y; w[4] = c.z; w[5] = c.w; w[6] = b.x; w[7] = b.y;
            issue_row(j + P);                                // into the slot read two steps ago
            hrow_window<C, R, rowpass_default<R>>(w, any_edge, at_start, at_end, hw[m]);
            const int i = j - 2 * R;                         // output row completed by this input row
            uint32_t o[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                if constexpr (R == 1) {
                    // rows i, i+1, i+2 sit in window slots m+1, m+2, m (mod 3)
                    const uint32_t se = (((hw[(m + 2) % WIN][q] << 1) + hw[(m + 1) % WIN][q]) + hw[m][q]) << 4;
                    const uint32_t so = (((hw[(m + 2) % WIN][4 + q] << 1) + hw[(m + 1) % WIN][4 + q]) + hw[m][4 + q]) << 4;
                    o[q] = __builtin_amdgcn_perm(so, se, 0x07030501u);
                } else {
                    // rows i..i+4 sit in window slots m+1, m+2, m+3, m+4, m (mod 5)
                    const uint32_t se = mad6(hw[(m + 3) % WIN][q],
                                             ((hw[(m + 2) % WIN][q] + hw[(m + 4) % WIN][q]) << 2) +
                                                 (hw[(m + 1) % WIN][q] + hw[m][q]));
                    const uint32_t so = mad6(hw[(m + 3) % WIN][4 + q],
                                             ((hw[(m + 2) % WIN][4 + q] + hw[(m + 4) % WIN][4 + q]) << 2) +
                                                 (hw[(m + 1) % WIN][4 + q] + hw[m][4 + q]));
                    o[q] = __builtin_amdgcn_perm(so, se, 0x07030501u);
                }
            }
            u32x4 v; v.x = o[0]; v.y = o[1]; v.z = o[2]; v.w = o[3];
            const bool st = i >= 0 && i < rows_out;          // exactly one store per step, dropped when not wanted
            const unsigned orow = (unsigned)(up ? rows_out - 1 - i : i);
            __builtin_amdgcn_raw_buffer_store_b128(v, rout, (int)(st ? dst_off + orow * (unsigned)p.pitch : OOB), 0, 0);
        }
    }
    wait_vmcnt<0>();                                         // the last prefetches land before the LDS is released
}

// ----------------------------------------------------------------------------------
// direct kernel (trial): no LDS at all — rows go from global memory straight into registers
// ----------------------------------------------------------------------------------
// What keeps an elementwise kernel at 6.2 TB/s on this part is the amount of data it has in flight: registers, not the
// 160 KB of LDS, hold it.  Here every lane loads the BH + 2R rows of its own 16-byte chunk column with plain 16-byte
// loads, all issued up front (hipcc counts them: a row is consumed as soon as IT has landed), takes the 8 bytes either side
// from the neighbouring lanes' registers (v_mov_b32_dpp wave_shr:1 / wave_shl:1) and keeps the 2R+1-row window of
// horizontal sums in registers.  Lanes are consecutive chunks of the flattened (image, band, chunk column) space, 62 per
// wave: lanes 0 and 63 duplicate the neighbouring waves' edge chunks and only supply halo bytes.  No barrier, no LDS, no
// workgroup-level phase: ~10 KB in flight per wave, 16-20 waves per CU.
struct DirectParams {
    const uint8_t *in;
    uint8_t *out;
    long long in_stride, out_stride;
    long long total;                  // n_images * nbands * cpr chunk columns of work
    int pitch, cpr;
    int H, y0, y1;
    int nbands;
    unsigned nblocks;
    int xcd;
    const uint8_t *top, *bottom;      // PEER: where band rows [0, y0) / [y1, H) are read instead (nullptr = from the band itself)
};

// PEER (one band per launch, Approach 2 across GPUs): the halo rows above/below the output rows are read where they live —
// in the neighbouring ranks' shards, over xGMI — so a step is this one launch: no exchange, no copy, no halo rows kept.
template <int C, int R, int BH, bool PEER = false>
__global__ __launch_bounds__(256) void blur_direct_kernel(const DirectParams p)
{
    constexpr int WIN = 2 * R + 1, NR = BH + 2 * R;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned B = xcd_map(blockIdx.x, p.nblocks, p.xcd);
    const long long fl = (long long)(B * 4u + (unsigned)wave) * 62 - 1 + lane;
    const bool inrange = fl >= 0 && fl < p.total;
    const unsigned f = (unsigned)(fl < 0 ? 0 : (fl >= p.total ? p.total - 1 : fl));
    const unsigned col = f % (unsigned)p.cpr, t2 = f / (unsigned)p.cpr;
    const unsigned band = t2 % (unsigned)p.nbands;
    const long long img = (long long)(t2 / (unsigned)p.nbands);
    const bool compute = inrange && lane >= 1 && lane <= 62;
    const int row0 = p.y0 + (int)band * BH;
    const int rows_out = compute ? min(BH, p.y1 - row0) : 0;
    const bool at_start = col == 0, at_end = (int)col == p.cpr - 1;
    const bool any_edge = __builtin_amdgcn_ballot_w64(at_start || at_end) != 0ull;
    const uint8_t *src = p.in + img * p.in_stride + (size_t)col * 16u;
    uint8_t *dst = p.out + img * p.out_stride + (size_t)(row0 - p.y0) * (size_t)p.pitch + (size_t)col * 16u;

    uint4 rows[NR];
    // PEER: only the first and last lane bands of the shard reach outside [y0, y1); every other wave takes the plain loop
    bool outside = false;
    if constexpr (PEER) outside = __builtin_amdgcn_ballot_w64(row0 - R < p.y0 || row0 + BH + R > p.y1) != 0ull;
    if (outside) {
#pragma unroll
        for (int j = 0; j < NR; j++) {
            const int sr = min(max(row0 - R + j, 0), p.H - 1);
            const uint8_t *rp = src + (size_t)sr * (size_t)p.pitch;
            if (sr < p.y0 && p.top) rp = p.top + (size_t)col * 16u + (size_t)sr * (size_t)p.pitch;
            if (sr >= p.y1 && p.bottom) rp = p.bottom + (size_t)col * 16u + (size_t)(sr - p.y1) * (size_t)p.pitch;
            rows[j] = *reinterpret_cast<const uint4 *>(rp);
        }
    } else {
#pragma unroll
        for (int j = 0; j < NR; j++)
            rows[j] = *reinterpret_cast<const uint4 *>(src + (size_t)min(max(row0 - R + j, 0), p.H - 1) * (size_t)p.pitch);
    }

    uint32_t hw[WIN][8];
#pragma unroll
    for (int j = 0; j < NR; j++) {
        uint32_t w[8];
        const uint4 c = rows[j];
        w[2] = c.x; w[3] = c.y; w[4] = c.z; w[5] = c.w;
        w[0] = __builtin_amdgcn_update_dpp(0u, c.z, 0x138, 0xf, 0xf, false);    // wave_shr:1: lane i takes lane i-1's last two dwords
        w[1] = __builtin_amdgcn_update_dpp(0u, c.w, 0x138, 0xf, 0xf, false);
        w[6] = __builtin_amdgcn_update_dpp(0u, c.x, 0x130, 0xf, 0xf, false);    // wave_shl:1: lane i+1's first two
        w[7] = __builtin_amdgcn_update_dpp(0u, c.y, 0x130, 0xf, 0xf, false);
        hrow_window<C, R, rowpass_default<R>>(w, any_edge, at_start, at_end, hw[j % WIN]);
        if (j >= 2 * R) {
            const int i = j - 2 * R;                                            // output row completed by this input row
            uint32_t o[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                if constexpr (R == 1) {
                    const uint32_t se = (((hw[(j + 2) % WIN][q] << 1) + hw[(j + 1) % WIN][q]) + hw[j % WIN][q]) << 4;
                    const uint32_t so = (((hw[(j + 2) % WIN][4 + q] << 1) + hw[(j + 1) % WIN][4 + q]) + hw[j % WIN][4 + q]) << 4;
                    o[q] = __builtin_amdgcn_perm(so, se, 0x07030501u);
                } else {
                    // rows i..i+4 sit in window slots (j+1), (j+2), (j+3), (j+4), j (mod 5)
                    const uint32_t se = mad6(hw[(j + 3) % WIN][q], ((hw[(j + 2) % WIN][q] + hw[(j + 4) % WIN][q]) << 2) + (hw[(j + 1) % WIN][q] + hw[j % WIN][q]));
                    const uint32_t so = mad6(hw[(j + 3) % WIN][4 + q], ((hw[(j + 2) % WIN][4 + q] + hw[(j + 4) % WIN][4 + q]) << 2) + (hw[(j + 1) % WIN][4 + q] + hw[j % WIN][4 + q]));
                    o[q] = __builtin_amdgcn_perm(so, se, 0x07030501u);
                }
            }
            if (i < rows_out) {
                u32x4 v; v.x = o[0]; v.y = o[1]; v.z = o[2]; v.w = o[3];
                *reinterpret_cast<u32x4 *>(dst + (size_t)i * (size_t)p.pitch) = v;
            }
        }
    }
}

// ----------------------------------------------------------------------------------
// generic kernel: one output byte per thread
// ----------------------------------------------------------------------------------
struct GenericParams {
    const uint8_t *in;
    uint8_t *out;
    long long in_stride, out_stride, total;
    int width, channels, pitch;
    int H, y0;
};

template <int R>
__global__ __launch_bounds__(256) void blur_generic_kernel(const GenericParams p)
{
    constexpr int T1[3] = {1, 2, 1};
    constexpr int T2[5] = {1, 4, 6, 4, 1};
    const long long step = (long long)gridDim.x * blockDim.x;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < p.total; idx += step) {
        const long long img = idx / p.out_stride;
        const long long rem = idx - img * p.out_stride;
        const int y = p.y0 + (int)(rem / p.pitch);
        const int b = (int)(rem % p.pitch);
        const int x = b / p.channels, c = b - x * p.channels;
        const uint8_t *src = p.in + img * p.in_stride;
        unsigned sum = 0;
#pragma unroll
        for (int ky = -R; ky <= R; ky++) {
            const int ny = min(max(y + ky, 0), p.H - 1);
            const uint8_t *rowp = src + (size_t)ny * (size_t)p.pitch + c;
#pragma unroll
            for (int kx = -R; kx <= R; kx++) {
                const int nx = min(max(x + kx, 0), p.width - 1);
                const int wgt = (R == 1) ? T1[ky + R] * T1[kx + R] : T2[ky + R] * T2[kx + R];
                sum += (unsigned)rowp[(size_t)nx * (size_t)p.channels] * (unsigned)wgt;
            }
        }
        p.out[idx] = (uint8_t)(sum >> (R == 1 ? 4 : 8));
    }
}

// ----------------------------------------------------------------------------------
// host side
// ----------------------------------------------------------------------------------
static std::mutex &tunables_mutex() { static std::mutex m; return m; }
static Tunables &tunables_storage()
{
    static Tunables t = [] {
        Tunables v{};                                    // everything 0 / off unless named here (rpg 0 / stream_bh 0 = choose per launch)
        v.stage_dma = 1; v.xcd_remap = 1; v.zero_copy = 1; v.ragged = 1;
        v.zero_copy_streams = 4; v.zero_copy_blocks = 24; v.stream_updown = 1; v.prefer_direct = 1; v.direct_bh = 8; v.fused_window = 8;
        v.zero_copy_server = 1; v.zero_copy_server_min_kb = 1280; v.staged_server = 1; v.zero_copy_workers = 48; v.zero_copy_idle_us = 300; v.zero_copy_budget = 256; v.zero_copy_tickets = 1;
        v.zero_copy_events = 1;
        v.fused_tail = 30; v.fused_tail_blocks = 25; v.fused_adds_per_word = 32;
        v.resident_place_trials = 4;
        if (const char *e = getenv("MI_BLUR_PLACE_TRIALS")) { const int r = atoi(e); if (r >= 0 && r <= 8) v.resident_place_trials = r; }
        if (const char *e = getenv("MI_BLUR_STAGED_SERVER")) v.staged_server = atoi(e) != 0;
        if (const char *e = getenv("MI_BLUR_FUSED_TAIL")) { const int r = atoi(e); if (r >= 0 && r <= 500) v.fused_tail = r; }
        if (const char *e = getenv("MI_BLUR_STAGE")) v.stage_dma = strcmp(e, "reg") != 0;
        if (const char *e = getenv("MI_BLUR_RPG")) { const int r = atoi(e); v.rpg = (r == 4 || r == 8 || r == 16) ? r : 0; }
        if (const char *e = getenv("MI_BLUR_XCD")) v.xcd_remap = atoi(e) != 0;
        if (const char *e = getenv("MI_BLUR_DIRECT")) { const int r = atoi(e); if (r >= 0 && r <= 2) v.prefer_direct = r; }
        return v;
    }();
    return t;
}
Tunables tunables()
{
    std::lock_guard<std::mutex> g(tunables_mutex());
    return tunables_storage();
}
void set_tunables(const Tunables &t)
{
    std::lock_guard<std::mutex> g(tunables_mutex());
    tunables_storage() = t;
}

// Diagnostics buffer (2 x u64 per workgroup, XCD_DEBUG_SLOTS workgroups, on the current device), allocated on first use
// and never freed.
unsigned long long *debug_xcd_buffer()
{
    static unsigned long long *buf = nullptr;
    static int buf_device = -1;
    static std::once_flag once;
    std::call_once(once, [] {
        if (hipGetDevice(&buf_device) != hipSuccess || hipMalloc((void **)&buf, (size_t)XCD_DEBUG_SLOTS * 16) != hipSuccess) { (void)hipGetLastError(); buf = nullptr; }
        else (void)hipMemset(buf, 0, (size_t)XCD_DEBUG_SLOTS * 16);
    });
    int dev = -1;                                    // the slots live on the device that was current at first use: launches on
    if (hipGetDevice(&dev) != hipSuccess || dev != buf_device) { (void)hipGetLastError(); return nullptr; }   // another device get none
    return buf;
}
unsigned debug_xcd_slots() { return XCD_DEBUG_SLOTS; }

// The ragged form of the tiled kernel: any pitch of at least one chunk, any pointer alignment.
static bool ragged_eligible(int width, int channels)
{
    return channels >= 1 && channels <= 4 && (long long)width * channels >= 16;
}

bool tiled_eligible(const void *in, const void *out, int width, int channels)
{
    if (channels < 1 || channels > 4) return false;
    const long long pitch = (long long)width * channels;
    return (pitch % 16 == 0) && ((uintptr_t)in % 16 == 0) && ((uintptr_t)out % 16 == 0);
}

static thread_local const char *g_last_kernel = "";
const char *last_kernel() { return g_last_kernel; }

static inline int hip_status(hipError_t e) { return e == hipSuccess ? MI_BLUR_OK : MI_BLUR_ERR_HIP_BASE - (int)e; }

template <typename K, typename P>
static int do_launch(K kernel, dim3 grid, dim3 block, size_t lds, const LaunchDesc &d, const P &params)
{
    if (d.start || d.stop)
        hipExtLaunchKernelGGL(kernel, grid, block, lds, d.stream, d.start, d.stop, 0, params);
    else
        hipLaunchKernelGGL(kernel, grid, block, lds, d.stream, params);
    return hip_status(hipGetLastError());
}

template <int C, int R>
static int launch_tiled_cr(const LaunchDesc &d, const TiledParams &p, dim3 grid, dim3 block, size_t lds,
                           int rpg, bool dma, bool ragged, bool row_shuffle, int experiment)
{
    if (ragged) {
        if (dma)
            return rpg == 4 ? do_launch(blur_tiled_kernel<C, R, 4, true, false, true>, grid, block, lds, d, p)
                            : do_launch(blur_tiled_kernel<C, R, 8, true, false, true>, grid, block, lds, d, p);
        return rpg == 4 ? do_launch(blur_tiled_kernel<C, R, 4, false, false, true>, grid, block, lds, d, p)
                        : do_launch(blur_tiled_kernel<C, R, 8, false, false, true>, grid, block, lds, d, p);
    }
    if (d.max_blocks > 0 && dma && rpg != 16 && grid.x > (unsigned)d.max_blocks) {
        const dim3 capped((unsigned)d.max_blocks);
        return rpg == 4 ? do_launch(blur_tiled_loop_kernel<C, R, 4>, capped, block, lds, d, p)
                        : do_launch(blur_tiled_loop_kernel<C, R, 8>, capped, block, lds, d, p);
    }
    if (experiment && dma && C == 3 && rpg != 16) {
        constexpr int OTHER = 1 - rowpass_default<R>;
        return rpg == 4 ? do_launch(blur_tiled_x_kernel<C, R, 4, OTHER>, grid, block, lds, d, p)
                        : do_launch(blur_tiled_x_kernel<C, R, 8, OTHER>, grid, block, lds, d, p);
    }
    if (row_shuffle && dma) {
        if (rpg == 16) return do_launch(blur_tiled_kernel<C, R, 16, true, true>, grid, block, lds, d, p);
        if (rpg == 4) return do_launch(blur_tiled_kernel<C, R, 4, true, true>, grid, block, lds, d, p);
        return do_launch(blur_tiled_kernel<C, R, 8, true, true>, grid, block, lds, d, p);
    }
    if (rpg == 16)
        return dma ? do_launch(blur_tiled_kernel<C, R, 16, true>, grid, block, lds, d, p)
                   : do_launch(blur_tiled_kernel<C, R, 16, false>, grid, block, lds, d, p);
    if (rpg == 4)
        return dma ? do_launch(blur_tiled_kernel<C, R, 4, true>, grid, block, lds, d, p)
                   : do_launch(blur_tiled_kernel<C, R, 4, false>, grid, block, lds, d, p);
    return dma ? do_launch(blur_tiled_kernel<C, R, 8, true>, grid, block, lds, d, p)
               : do_launch(blur_tiled_kernel<C, R, 8, false>, grid, block, lds, d, p);
}

// 5 to 8 channels, 3x3 only (the +-C byte taps still fall inside the 8 + 16 + 8-byte window; 5x5 would need +-2C): the
// LDS-DMA tiled kernel in its aligned and ragged forms, nothing else — the reference kernel is generic in `channels`
// (gaussian_kernel.cl:44), so frames with more than four planes should not drop to one byte per thread.
template <int C>
static int launch_tiled_wide(const LaunchDesc &d, const TiledParams &p, dim3 grid, dim3 block, size_t lds, int rpg, bool ragged)
{
    if (ragged)
        return rpg == 4 ? do_launch(blur_tiled_kernel<C, 1, 4, true, false, true>, grid, block, lds, d, p)
                        : do_launch(blur_tiled_kernel<C, 1, 8, true, false, true>, grid, block, lds, d, p);
    return rpg == 4 ? do_launch(blur_tiled_kernel<C, 1, 4, true>, grid, block, lds, d, p)
                    : do_launch(blur_tiled_kernel<C, 1, 8, true>, grid, block, lds, d, p);
}

static bool wide_channels(int channels, int radius) { return channels >= 5 && channels <= 8 && radius == 1; }

template <int R>
static int launch_tiled_r(const LaunchDesc &d, const TiledParams &p, dim3 grid, dim3 block, size_t lds,
                          int rpg, bool dma, bool ragged, bool row_shuffle, int experiment)
{
    if constexpr (R == 1) {
        switch (d.channels) {
        case 5: return launch_tiled_wide<5>(d, p, grid, block, lds, rpg, ragged);
        case 6: return launch_tiled_wide<6>(d, p, grid, block, lds, rpg, ragged);
        case 7: return launch_tiled_wide<7>(d, p, grid, block, lds, rpg, ragged);
        case 8: return launch_tiled_wide<8>(d, p, grid, block, lds, rpg, ragged);
        }
    }
    switch (d.channels) {
    case 1: return launch_tiled_cr<1, R>(d, p, grid, block, lds, rpg, dma, ragged, row_shuffle, experiment);
    case 2: return launch_tiled_cr<2, R>(d, p, grid, block, lds, rpg, dma, ragged, row_shuffle, experiment);
    case 3: return launch_tiled_cr<3, R>(d, p, grid, block, lds, rpg, dma, ragged, row_shuffle, experiment);
    case 4: return launch_tiled_cr<4, R>(d, p, grid, block, lds, rpg, dma, ragged, row_shuffle, experiment);
    }
    return MI_BLUR_ERR_INVALID;
}

template <int R>
static int launch_fused_r(const LaunchDesc &d, const TiledParams &p, const FusedParams &f, dim3 grid, dim3 block, size_t lds, int rpg, bool ragged = false)
{
    auto go = [&](auto kernel) {
        if (d.start || d.stop) hipExtLaunchKernelGGL(kernel, grid, block, lds, d.stream, d.start, d.stop, 0, p, f);
        else hipLaunchKernelGGL(kernel, grid, block, lds, d.stream, p, f);
        return hip_status(hipGetLastError());
    };
    if (ragged) {                                        // rows/thread 8, the tail kernel (its tail may be empty)
        if (!f.tail_ctr || rpg != 8) return MI_BLUR_ERR_INVALID;
        switch (d.channels) {
        case 1: return go(blur_fused_tail_kernel<1, R, 8, true>);
        case 2: return go(blur_fused_tail_kernel<2, R, 8, true>);
        case 3: return go(blur_fused_tail_kernel<3, R, 8, true>);
        case 4: return go(blur_fused_tail_kernel<4, R, 8, true>);
        }
        return MI_BLUR_ERR_INVALID;
    }
    if (f.tail_ctr) {
        switch (d.channels * 10 + rpg) {
        case 14: return go(blur_fused_tail_kernel<1, R, 4>);
        case 18: return go(blur_fused_tail_kernel<1, R, 8>);
        case 24: return go(blur_fused_tail_kernel<2, R, 4>);
        case 28: return go(blur_fused_tail_kernel<2, R, 8>);
        case 34: return go(blur_fused_tail_kernel<3, R, 4>);
        case 38: return go(blur_fused_tail_kernel<3, R, 8>);
        case 44: return go(blur_fused_tail_kernel<4, R, 4>);
        case 48: return go(blur_fused_tail_kernel<4, R, 8>);
        }
        return MI_BLUR_ERR_INVALID;
    }
    switch (d.channels * 10 + rpg) {
    case 14: return go(blur_fused_kernel<1, R, 4>);
    case 18: return go(blur_fused_kernel<1, R, 8>);
    case 24: return go(blur_fused_kernel<2, R, 4>);
    case 28: return go(blur_fused_kernel<2, R, 8>);
    case 34: return go(blur_fused_kernel<3, R, 4>);
    case 38: return go(blur_fused_kernel<3, R, 8>);
    case 44: return go(blur_fused_kernel<4, R, 4>);
    case 48: return go(blur_fused_kernel<4, R, 8>);
    }
    return MI_BLUR_ERR_INVALID;
}

// Tile geometry of one launch of the tiled kernel family: fills p and returns the output rows per thread it chose.
// rpg_forced > 0 pins that choice (the zero-copy batch server runs one geometry for every batch).
static int tiled_geometry(const LaunchDesc &d, const Tunables &tun, bool ragged, bool fused, int rpg_forced, TiledParams &p,
                          unsigned *block_threads, size_t *lds_bytes)
{
    const int R = d.radius;
    const int pitch = d.width * d.channels, cpr = (pitch + 15) / 16, rows = d.y1 - d.y0;   // ragged: last chunk partial
    // Output rows per thread.  8 amortises the 2R priming rows of the sliding window best when the grid is
    // large; small and mid-size grids (a batch of 35 256x256 images is ~840 waves at 8 rows) finish sooner
    // with 4 — more, shorter waves per CU (measured: 6.2 vs 7.4 us at batch 35, equal by ~20k waves).
    int rpg = rpg_forced > 0 ? rpg_forced : tun.rpg;
    if (rpg == 0) {
        const long long waves8 = (long long)d.n_images * rows * cpr / (8 * 64);
        rpg = waves8 < 16384 ? 4 : 8;
    }
    if ((ragged || fused || d.channels > 4) && rpg == 16) rpg = 8;

    p = TiledParams{};
    p.in = d.in; p.out = d.out;
    p.in_stride = d.in_stride ? d.in_stride : (long long)d.band_rows * pitch;
    p.out_stride = d.out_stride ? d.out_stride : (long long)rows * pitch;
    p.pitch = pitch; p.cpr = cpr; p.H = d.band_rows; p.y0 = d.y0; p.y1 = d.y1;
    p.nstrips = (cpr + 61) / 62;               // ncols + 2 halo chunks <= 64 lanes: one row per wave-instruction
    p.ncols = (cpr + p.nstrips - 1) / p.nstrips;

    // Row groups per tile: the most efficient split of the rows (fewest idle row slots
    // and idle lanes), within 256 threads and 64 KiB of LDS.
    const int maxg = 256 / p.ncols, need = (rows + rpg - 1) / rpg;
    int best_g = 1; double best = -1.0;
    for (int g = 1; g <= maxg && g <= need; g++) {
        const size_t lds = (size_t)(g * rpg + 2 * R) * (p.ncols + 2) * 16;
        if (lds > 64 * 1024) break;
        const int nt = (p.ncols * g + 63) / 64 * 64;
        const int tiles = (rows + g * rpg - 1) / (g * rpg);
        const double eff = (double)rows / ((double)tiles * g * rpg) * ((double)p.ncols * g / nt);
        if (eff >= best - 1e-9) { best = eff; best_g = g; }
    }
    p.ngroups = best_g;
    p.TH = best_g * rpg;
    p.ntiles_y = (rows + p.TH - 1) / p.TH;
    const long long nblocks = (long long)d.n_images * p.ntiles_y * p.nstrips;
    if (nblocks > 0x7fffffffLL) return MI_BLUR_ERR_INVALID;
    p.nblocks = (unsigned)nblocks;
    // blockIdx -> tile map: one contiguous eighth of the launch per XCD (tile-edge halo rows are then hits in that XCD's
    // L2 and HBM traffic stays at 1.00x the algorithmic bytes).  Runs of r tiles dealt to the XCDs in turn ("xcd_run" r)
    // were measured as well (profiles/r02_xcd_runs.txt): +-2 % either way depending on where the buffers happen to lie,
    // and +0.4..3 % HBM fetch for the seams that cross XCDs — not the default.
    int xmap = 0;
    if (tun.xcd_remap && nblocks >= 16) {
        xmap = tun.xcd_run >= 2 ? tun.xcd_run : 1;
        if (xmap >= 2 && (long long)xmap * 8 > nblocks) xmap = 1;
    }
    p.xcd = xmap;
    p.debug_copy = tun.debug_copy;
    p.xcd_times = tun.debug_xcd_times ? debug_xcd_buffer() : nullptr;
    p.tail = pitch % 16 ? pitch % 16 : 16;
    *block_threads = (unsigned)((p.ncols * p.ngroups + 63) / 64 * 64);
    *lds_bytes = (size_t)(p.TH + 2 * R) * (p.ncols + 2) * 16;
    return rpg;
}

static int launch_tiled(const LaunchDesc &d, const Tunables &tun, bool ragged = false, const FusedDesc *fused = nullptr)
{
    if (!(fused && fused->geometry_only))
        g_last_kernel = fused ? "blur_fused_kernel" : (d.max_blocks > 0 && !ragged && d.channels <= 4 ? "blur_tiled_loop_kernel" : "blur_tiled_kernel");
    const int R = d.radius;
    TiledParams p{};
    unsigned threads = 0;
    size_t lds = 0;
    const int rpg = tiled_geometry(d, tun, ragged, fused != nullptr, (fused && ragged) ? 8 : 0, p, &threads, &lds);
    if (rpg < 0) return rpg;
    const long long nblocks = p.nblocks;
    const dim3 grid((unsigned)nblocks), block(threads);
    if (fused) {
        FusedParams f{};
        f.count = fused->count;
        f.release = tun.fused_release;
        f.window = (unsigned)std::max(1, tun.fused_window);
        const long long tpb = (long long)fused->batch_images * p.ntiles_y * p.nstrips;
        if (tpb <= 0 || tpb > 0x7fffffffLL) return MI_BLUR_ERR_INVALID;
        f.tiles_per_batch = (unsigned)tpb;
        // counters per batch: at most ~64 adds per word and pass, within what the caller's counter array holds
        unsigned k = 8;
        while (k < 256u && tpb > (long long)std::max(4, tun.fused_adds_per_word) * k) k *= 2;
        const long long nbatches = ((long long)d.n_images + fused->batch_images - 1) / fused->batch_images;
        while (k > 8u && fused->count_words > 0 && (long long)k * nbatches > fused->count_words) k /= 2;
        f.kcount = k;
        if (fused->counters_per_batch) *fused->counters_per_batch = k;
        if (fused->tiles_per_batch) *fused->tiles_per_batch = (unsigned)tpb;
        if (fused->waves_per_block) *fused->waves_per_block = 1;      // one count per block
        if (fused->total_blocks) *fused->total_blocks = (unsigned)nblocks;
        if (fused->geometry_only) return MI_BLUR_OK;
        p.debug_copy = 0;
        // dynamic tail ("fused_tail" t, per mille of the pass's tiles; 0 = off): worth it only when the pass is many rounds of
        // resident workgroups long; the counter comes from a small ring so that passes in flight on different streams do not share one
        if (tun.fused_tail > 0 && fused->tail_ctr && nblocks >= 8192) {
            const unsigned ntail = (unsigned)(nblocks * tun.fused_tail / 1000);
            if (ntail >= 8) {
                f.tail_ctr = fused->tail_ctr; f.ntail = ntail; f.nstatic = (unsigned)nblocks - ntail;
                f.nextra = ntail + (unsigned)((long long)ntail * std::max(10, tun.fused_tail_blocks) / 100);
                g_last_kernel = "blur_fused_tail_kernel";
                const dim3 tgrid(f.nstatic + f.nextra);
                return R == 1 ? launch_fused_r<1>(d, p, f, tgrid, block, lds, rpg, ragged) : launch_fused_r<2>(d, p, f, tgrid, block, lds, rpg, ragged);
            }
        }
        if (ragged) {      // ragged rows, no tail: the same kernel with every tile mapped to a block
            if (!fused->tail_ctr) return MI_BLUR_ERR_UNSUPPORTED;
            f.tail_ctr = fused->tail_ctr; f.ntail = 0; f.nstatic = (unsigned)nblocks; f.nextra = 0;
            g_last_kernel = "blur_fused_tail_kernel";
        }
        return R == 1 ? launch_fused_r<1>(d, p, f, grid, block, lds, rpg, ragged) : launch_fused_r<2>(d, p, f, grid, block, lds, rpg, ragged);
    }
    return R == 1 ? launch_tiled_r<1>(d, p, grid, block, lds, rpg, tun.stage_dma != 0, ragged, tun.row_shuffle != 0, tun.experiment)
                  : launch_tiled_r<2>(d, p, grid, block, lds, rpg, tun.stage_dma != 0, ragged, tun.row_shuffle != 0, tun.experiment);
}

// ---- zero-copy batch server: host side of the launch interface (blur_launch.h)
int zc_fill_batch(const LaunchDesc &d, ZcGeometry *geo, ZcBatch *b, unsigned *n_tiles)
{
    if (!geo || !b || !d.in || !d.out || d.n_images <= 0) return MI_BLUR_ERR_INVALID;
    Tunables tun = tunables();
    const bool aligned = tiled_eligible(d.in, d.out, d.width, d.channels) && !(d.in_stride % 16) && !(d.out_stride % 16);
    // rows that are not a multiple of 16 bytes / unaligned buffers: the ragged form of the same tile code, dense batches only
    const long long pitch_b = (long long)d.width * d.channels;
    const bool dense = (!d.in_stride || d.in_stride == pitch_b * d.band_rows) && (!d.out_stride || d.out_stride == pitch_b * (d.y1 - d.y0));
    const bool ragged = !aligned && tun.ragged && ragged_eligible(d.width, d.channels) && dense;
    if (!aligned && !ragged) return MI_BLUR_ERR_UNSUPPORTED;
    tun.debug_copy = 0; tun.debug_xcd_times = 0; tun.xcd_remap = 0;      // tiles are dealt to the workers by the server itself
    TiledParams p{};
    unsigned threads = 0;
    size_t lds = 0;
    const int rpg = tiled_geometry(d, tun, ragged, false, 4, p, &threads, &lds);
    if (rpg < 0) return rpg;
    if (geo->threads == 0) { geo->threads = threads; geo->lds = lds; geo->rpg = rpg; geo->channels = d.channels; geo->radius = d.radius; geo->ragged = ragged ? 1 : 0; }
    else if (geo->threads != threads || geo->lds != lds || geo->rpg != rpg || geo->channels != d.channels || geo->radius != d.radius ||
             geo->ragged != (ragged ? 1 : 0))
        return MI_BLUR_ERR_UNSUPPORTED;                                      // another tile shape than the running server's
    memset(b->params, 0, sizeof b->params);
    memcpy(b->params, &p, sizeof p);
    if (n_tiles) *n_tiles = p.nblocks;
    return MI_BLUR_OK;
}

int zc_launch_server(const ZcGeometry &geo, ZcHostCtl *ctl, ZcDevCtl *dev, unsigned seq, unsigned n_workers, unsigned budget,
                     unsigned idle_ticks, hipStream_t stream, unsigned long long *trace, int fixed_share)
{
    if (!ctl || !dev || n_workers == 0 || geo.threads == 0 || geo.rpg != 4) return MI_BLUR_ERR_INVALID;
    g_last_kernel = "blur_server_kernel";
    const dim3 grid(n_workers + 1), block(geo.threads);
    auto go = [&](auto kernel) {
        hipLaunchKernelGGL(kernel, grid, block, geo.lds, stream, ctl, dev, seq, budget, idle_ticks, trace, fixed_share);
        return hip_status(hipGetLastError());
    };
    switch (geo.channels * 100 + geo.radius * 10 + geo.ragged) {
    case 110: return go(blur_server_kernel<1, 1, 4, false>);
    case 120: return go(blur_server_kernel<1, 2, 4, false>);
    case 210: return go(blur_server_kernel<2, 1, 4, false>);
    case 220: return go(blur_server_kernel<2, 2, 4, false>);
    case 310: return go(blur_server_kernel<3, 1, 4, false>);
    case 320: return go(blur_server_kernel<3, 2, 4, false>);
    case 410: return go(blur_server_kernel<4, 1, 4, false>);
    case 420: return go(blur_server_kernel<4, 2, 4, false>);
    case 111: return go(blur_server_kernel<1, 1, 4, true>);
    case 121: return go(blur_server_kernel<1, 2, 4, true>);
    case 211: return go(blur_server_kernel<2, 1, 4, true>);
    case 221: return go(blur_server_kernel<2, 2, 4, true>);
    case 311: return go(blur_server_kernel<3, 1, 4, true>);
    case 321: return go(blur_server_kernel<3, 2, 4, true>);
    case 411: return go(blur_server_kernel<4, 1, 4, true>);
    case 421: return go(blur_server_kernel<4, 2, 4, true>);
    }
    return MI_BLUR_ERR_INVALID;
}

int launch_fused_watch(const unsigned *count, unsigned n_batches, unsigned tiles_per_batch, unsigned total_blocks, unsigned per_block,
                       unsigned long long *host_word, unsigned pass_seq, hipStream_t stream, unsigned counters_per_batch)
{
    if (!count || !host_word || n_batches == 0 || tiles_per_batch == 0 || counters_per_batch < 8 || (counters_per_batch & (counters_per_batch - 1))) return MI_BLUR_ERR_INVALID;
    hipLaunchKernelGGL(fused_watch_kernel, dim3(1), dim3(64), 0, stream, count, n_batches, tiles_per_batch, total_blocks, per_block, host_word, pass_seq, counters_per_batch);
    return hip_status(hipGetLastError());
}

int launch_fused(const LaunchDesc &d, const FusedDesc &f)
{
    if (!d.in || !d.out || d.in == d.out || (!f.count && !f.geometry_only) || f.batch_images <= 0) return MI_BLUR_ERR_INVALID;
    if (d.width <= 0 || d.band_rows <= 0 || d.n_images <= 0 || (d.radius != 1 && d.radius != 2)) return MI_BLUR_ERR_INVALID;
    if (d.y0 != 0 || d.y1 != d.band_rows || d.in_stride || d.out_stride) return MI_BLUR_ERR_INVALID;
    if ((long long)d.width * d.channels * d.band_rows > INT_MAX) return MI_BLUR_ERR_INVALID;
    const Tunables &tn = f.tun ? *f.tun : tunables();
    if (tiled_eligible(d.in, d.out, d.width, d.channels)) return launch_tiled(d, tn, false, &f);
    // rows that are not a multiple of 16 bytes, buffers at odd addresses: the ragged form of the same tiles
    if (tn.ragged && ragged_eligible(d.width, d.channels) && f.tail_ctr) return launch_tiled(d, tn, true, &f);
    return MI_BLUR_ERR_UNSUPPORTED;
}

static int launch_stream(const LaunchDesc &d, const Tunables &tun)
{
    g_last_kernel = "blur_stream_kernel";
    const int pitch = d.width * d.channels, cpr = pitch / 16, rows = d.y1 - d.y0;
    StreamParams p{};
    p.in = d.in; p.out = d.out;
    p.in_stride = (long long)d.band_rows * pitch;
    p.out_stride = (long long)rows * pitch;
    p.pitch = pitch; p.cpr = cpr; p.H = d.band_rows; p.y0 = d.y0; p.y1 = d.y1;
    // Band height: tall bands re-read fewer halo rows (2R/BH), short ones give small grids enough waves.  A wave lives as
    // long as its band is tall and ~16 waves fit a CU, so the launch runs in "rounds" of resident waves: pick the band
    // count whose last round is fullest (a launch of 1.5 rounds runs its second half at half occupancy), weighted by the
    // halo re-read overhead BH / (BH + 2R).
    int BH = tun.stream_bh;
    if (BH <= 0) {
        const double slots = 256.0 * 16.0;
        double best = -1.0;
        BH = 64;
        for (int nb = (rows + 127) / 128; nb <= (rows + 23) / 24; nb++) {
            const int bh = (rows + nb - 1) / nb;
            const double waves = (double)d.n_images * nb * cpr / 64.0;
            const double rounds = waves / slots;
            const double full = rounds >= 1.0 ? rounds / ceil(rounds - 1e-9) : rounds;
            const double score = full * bh / (bh + 2.0 * d.radius);
            if (score > best + 1e-9) { best = score; BH = bh; }
        }
        // small grids (well under one round of resident waves): shorter bands, more waves
        while (BH > 8 && (long long)d.n_images * ((rows + BH - 1) / BH) * cpr / 64 < 256 * 8) BH >>= 1;
    }
    if (BH > rows) BH = rows;
    p.BH = BH;
    p.nbands = (rows + BH - 1) / BH;
    p.total = (long long)d.n_images * p.nbands * cpr;
    p.total_images = d.n_images;
    if (p.total >= 0x7fffffffLL) return MI_BLUR_ERR_INVALID;
    const long long nblocks = (p.total + 255) / 256;
    p.nblocks = (unsigned)nblocks;
    p.xcd = tun.xcd_remap && nblocks >= 16;
    p.updown = tun.stream_updown;
    const dim3 grid((unsigned)nblocks), block(256);
    const size_t lds = 4 * STREAM_D * STREAM_ROWB;
    switch (d.channels * 10 + d.radius) {
    case 11: return do_launch(blur_stream_kernel<1, 1>, grid, block, lds, d, p);
    case 12: return do_launch(blur_stream_kernel<1, 2>, grid, block, lds, d, p);
    case 21: return do_launch(blur_stream_kernel<2, 1>, grid, block, lds, d, p);
    case 22: return do_launch(blur_stream_kernel<2, 2>, grid, block, lds, d, p);
    case 31: return do_launch(blur_stream_kernel<3, 1>, grid, block, lds, d, p);
    case 32: return do_launch(blur_stream_kernel<3, 2>, grid, block, lds, d, p);
    case 41: return do_launch(blur_stream_kernel<4, 1>, grid, block, lds, d, p);
    case 42: return do_launch(blur_stream_kernel<4, 2>, grid, block, lds, d, p);
    }
    return MI_BLUR_ERR_INVALID;
}

template <int BH>
static int launch_direct_bh(const LaunchDesc &d, const Tunables &tun);

// The direct kernel numbers its work in 32 bits: images x bands x chunk columns.
static bool direct_fits(const LaunchDesc &d)
{
    const long long cpr = (long long)d.width * d.channels / 16, rows = d.y1 - d.y0;
    return (long long)d.n_images * ((rows + 3) / 4) * cpr < 0x7fffffffLL;
}

static int launch_direct(const LaunchDesc &d, const Tunables &tun)
{
    if (d.channels == 3 && tun.direct_bh == 4) return launch_direct_bh<4>(d, tun);
    if (d.channels == 3 && tun.direct_bh == 16) return launch_direct_bh<16>(d, tun);
    if (d.channels == 3 && tun.direct_bh == 12) return launch_direct_bh<12>(d, tun);
    return launch_direct_bh<8>(d, tun);
}

template <int BH>
static int launch_direct_bh(const LaunchDesc &d, const Tunables &tun)
{
    g_last_kernel = "blur_direct_kernel";
    const int pitch = d.width * d.channels, cpr = pitch / 16, rows = d.y1 - d.y0;
    DirectParams p{};
    p.in = d.in; p.out = d.out;
    p.in_stride = (long long)d.band_rows * pitch;
    p.out_stride = (long long)rows * pitch;
    p.pitch = pitch; p.cpr = cpr; p.H = d.band_rows; p.y0 = d.y0; p.y1 = d.y1;
    p.nbands = (rows + BH - 1) / BH;
    p.total = (long long)d.n_images * p.nbands * cpr;
    if (p.total >= 0x7fffffffLL) return MI_BLUR_ERR_INVALID;
    const long long waves = (p.total + 61) / 62, nblocks = (waves + 3) / 4;
    p.nblocks = (unsigned)nblocks;
    p.xcd = (tun.xcd_remap && nblocks >= 16) ? 1 : 0;
    const dim3 grid((unsigned)nblocks), block(256);
    if (d.halo_top || d.halo_bottom) {
        if constexpr (BH == 8) {
            p.top = d.halo_top; p.bottom = d.halo_bottom;
            g_last_kernel = "blur_direct_kernel (peer halo rows)";
            switch (d.channels * 10 + d.radius) {
            case 11: return do_launch(blur_direct_kernel<1, 1, 8, true>, grid, block, 0, d, p);
            case 12: return do_launch(blur_direct_kernel<1, 2, 8, true>, grid, block, 0, d, p);
            case 21: return do_launch(blur_direct_kernel<2, 1, 8, true>, grid, block, 0, d, p);
            case 22: return do_launch(blur_direct_kernel<2, 2, 8, true>, grid, block, 0, d, p);
            case 31: return do_launch(blur_direct_kernel<3, 1, 8, true>, grid, block, 0, d, p);
            case 32: return do_launch(blur_direct_kernel<3, 2, 8, true>, grid, block, 0, d, p);
            case 41: return do_launch(blur_direct_kernel<4, 1, 8, true>, grid, block, 0, d, p);
            case 42: return do_launch(blur_direct_kernel<4, 2, 8, true>, grid, block, 0, d, p);
            }
        }
        return MI_BLUR_ERR_INVALID;
    }
    if constexpr (BH != 8) {
        return d.radius == 1 ? do_launch(blur_direct_kernel<3, 1, BH>, grid, block, 0, d, p)
                             : do_launch(blur_direct_kernel<3, 2, BH>, grid, block, 0, d, p);
    } else {
        switch (d.channels * 10 + d.radius) {
        case 11: return do_launch(blur_direct_kernel<1, 1, BH>, grid, block, 0, d, p);
        case 12: return do_launch(blur_direct_kernel<1, 2, BH>, grid, block, 0, d, p);
        case 21: return do_launch(blur_direct_kernel<2, 1, BH>, grid, block, 0, d, p);
        case 22: return do_launch(blur_direct_kernel<2, 2, BH>, grid, block, 0, d, p);
        case 31: return do_launch(blur_direct_kernel<3, 1, BH>, grid, block, 0, d, p);
        case 32: return do_launch(blur_direct_kernel<3, 2, BH>, grid, block, 0, d, p);
        case 41: return do_launch(blur_direct_kernel<4, 1, BH>, grid, block, 0, d, p);
        case 42: return do_launch(blur_direct_kernel<4, 2, BH>, grid, block, 0, d, p);
        }
        return MI_BLUR_ERR_INVALID;
    }
}

static int launch_generic(const LaunchDesc &d)
{
    g_last_kernel = "blur_generic_kernel";
    GenericParams p{};
    const int pitch = d.width * d.channels, rows = d.y1 - d.y0;
    p.in = d.in; p.out = d.out;
    p.in_stride = (long long)d.band_rows * pitch;
    p.out_stride = (long long)rows * pitch;
    p.total = p.out_stride * d.n_images;
    p.width = d.width; p.channels = d.channels; p.pitch = pitch; p.H = d.band_rows; p.y0 = d.y0;
    long long blocks = (p.total + 255) / 256;
    if (blocks > 256LL * 64) blocks = 256LL * 64;   // grid-stride the rest
    const dim3 grid((unsigned)blocks), block(256);
    return d.radius == 1 ? do_launch(blur_generic_kernel<1>, grid, block, 0, d, p)
                         : do_launch(blur_generic_kernel<2>, grid, block, 0, d, p);
}

int launch(const LaunchDesc &d)
{
    if (!d.in || !d.out || d.in == d.out) return MI_BLUR_ERR_INVALID;
    if (d.width <= 0 || d.band_rows <= 0 || d.channels <= 0 || d.n_images < 0) return MI_BLUR_ERR_INVALID;
    if (d.radius != 1 && d.radius != 2) return MI_BLUR_ERR_INVALID;
    if (d.y0 < 0 || d.y1 > d.band_rows || d.y0 >= d.y1) return MI_BLUR_ERR_INVALID;
    if ((long long)d.width * d.channels > INT_MAX / 2) return MI_BLUR_ERR_INVALID;
    if ((long long)d.width * d.channels * d.band_rows > INT_MAX) return MI_BLUR_ERR_INVALID;  // per-image 32-bit
    if (d.n_images == 0) return MI_BLUR_OK;
    const Tunables tun = tunables();                  // one coherent set of knobs for this launch
    const bool wide = wide_channels(d.channels, d.radius);
    const long long row_bytes = (long long)d.width * d.channels;
    const bool can_tile = wide ? (row_bytes % 16 == 0 && (uintptr_t)d.in % 16 == 0 && (uintptr_t)d.out % 16 == 0)
                               : tiled_eligible(d.in, d.out, d.width, d.channels);
    if (d.halo_top || d.halo_bottom) {
        // halo rows read in place from other shards: one dense band, the direct kernel's shapes (rows of whole 16-byte chunks)
        if (d.n_images != 1 || d.in_stride || d.out_stride || d.max_blocks > 0) return MI_BLUR_ERR_INVALID;
        if ((uintptr_t)d.halo_top % 16 || (uintptr_t)d.halo_bottom % 16) return MI_BLUR_ERR_INVALID;
        if (d.variant != MI_BLUR_VARIANT_AUTO && d.variant != MI_BLUR_VARIANT_DIRECT) return MI_BLUR_ERR_UNSUPPORTED;
        if (!can_tile || wide || !direct_fits(d)) return MI_BLUR_ERR_UNSUPPORTED;
        Tunables t8 = tun; t8.direct_bh = 8;
        return launch_direct(d, t8);
    }
    const long long dense_in = (long long)d.band_rows * d.width * d.channels, dense_out = (long long)(d.y1 - d.y0) * d.width * d.channels;
    if ((d.in_stride && d.in_stride != dense_in) || (d.out_stride && d.out_stride != dense_out)) {
        // spaced-out bands (a caller's buffer used in place): tiled kernel only
        if (d.in_stride < 0 || d.out_stride < 0 || (d.in_stride && d.in_stride < dense_in) || (d.out_stride && d.out_stride < dense_out))
            return MI_BLUR_ERR_INVALID;
        if (!can_tile || d.in_stride % 16 || d.out_stride % 16) return MI_BLUR_ERR_UNSUPPORTED;
        if (d.variant != MI_BLUR_VARIANT_AUTO && d.variant != MI_BLUR_VARIANT_TILED) return MI_BLUR_ERR_UNSUPPORTED;
        return launch_tiled(d, tun);
    }
    const bool can_rag = (wide ? row_bytes >= 16 : ragged_eligible(d.width, d.channels)) && tun.ragged;
    switch (d.variant) {
    case MI_BLUR_VARIANT_AUTO:
        if (!can_tile) return can_rag ? launch_tiled(d, tun, true) : launch_generic(d);
        if (wide) return launch_tiled(d, tun);            // the streaming and direct variants exist for 1-4 channels only
        if (tun.prefer_stream) return launch_stream(d, tun);
        // Direct (LDS-free) or tiled?  Measured on MI355X (profiles/r02_direct_kernel.txt): the direct kernel wins every
        // 5x5 launch (5-22 %) and the 3x3 launches that do not fill the chip for long (a batch of 35 256x256 images: 5.4
        // against 6.8 us; parity near 150-400 MB of output); big 3x3 launches stay with the LDS tiles (1-4 % better there:
        // 1.06x instead of 1.25x row traffic into the CUs), and so do small 3x3 launches that the caller overlaps on several
        // streams (10.7 against 9.3 M img/s for the batch-35 stream on 4 streams; alone on one stream 4.5 against 5.4).
        // Zero-copy submits keep the capped-grid tiled kernel.
        if (direct_fits(d) && d.max_blocks <= 0 &&
            (tun.prefer_direct == 2 ||
             (tun.prefer_direct == 1 && (d.radius == 2 || (dense_out * d.n_images <= (128LL << 20) && d.concurrent <= 1)))))
            return launch_direct(d, tun);
        return launch_tiled(d, tun);
    case MI_BLUR_VARIANT_GENERIC: return launch_generic(d);
    case MI_BLUR_VARIANT_TILED: return can_tile ? launch_tiled(d, tun) : can_rag ? launch_tiled(d, tun, true) : MI_BLUR_ERR_INVALID;
    case MI_BLUR_VARIANT_STREAM: return (can_tile && !wide) ? launch_stream(d, tun) : MI_BLUR_ERR_INVALID;
    case MI_BLUR_VARIANT_DIRECT: return (can_tile && !wide && direct_fits(d)) ? launch_direct(d, tun) : MI_BLUR_ERR_INVALID;
    }
    return MI_BLUR_ERR_INVALID;
}

}  // namespace mi_blur
