// host-only sanitizer run: the product's CPU device vs the oracle's C restatement on random shapes,
// exact-size heap buffers so ASan sees any over-read/over-write.
#include "cpu_device.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
extern "C" void oracle_blur_int(const uint8_t *in, uint8_t *out, int W, int H, int C, int radius);
int main()
{
    unsigned s = 12345;
    auto rnd = [&](int n) { s = s * 1664525u + 1013904223u; return (int)((s >> 8) % (unsigned)n); };
    int cases = 0;
    for (int it = 0; it < 400; it++) {
        const int W = 1 + rnd(70), H = 1 + rnd(40), C = 1 + rnd(5), R = 1 + rnd(2), n = 1 + rnd(4), nt = 1 + rnd(5);
        const size_t isz = (size_t)W * H * C;
        uint8_t *in = (uint8_t *)malloc(isz * n), *out = (uint8_t *)malloc(isz * n), *want = (uint8_t *)malloc(isz * n);
        for (size_t i = 0; i < isz * n; i++) in[i] = (uint8_t)rnd(256);
        mi_blur::cpu_blur_batch(in, out, W, H, C, R, n, 0, H, nt, 0, 0);
        for (int i = 0; i < n; i++) oracle_blur_int(in + i * isz, want + i * isz, W, H, C, R);
        if (memcmp(out, want, isz * n)) { printf("MISMATCH W%d H%d C%d R%d n%d nt%d\n", W, H, C, R, n, nt); return 1; }
        // band form: rows [y0,y1) of a band
        const int y0 = rnd(H), y1 = y0 + 1 + rnd(H - y0);
        uint8_t *bo = (uint8_t *)malloc((size_t)(y1 - y0) * W * C * n);
        mi_blur::cpu_blur_batch(in, bo, W, H, C, R, n, y0, y1, nt, 0, 0);
        for (int i = 0; i < n; i++)
            if (memcmp(bo + (size_t)i * (y1 - y0) * W * C, want + i * isz + (size_t)y0 * W * C, (size_t)(y1 - y0) * W * C)) { printf("BAND MISMATCH\n"); return 1; }
        free(bo);
        // planar <-> interleaved repack of the frame path (mi_blur_submit_planar on the CPU device): round trip + spot check
        uint8_t *pl = (uint8_t *)malloc(isz * n), *back = (uint8_t *)malloc(isz * n);
        mi_blur::cpu_repack(in, pl, W, H, C, n, false, nt);             // interleaved -> planar
        mi_blur::cpu_repack(pl, back, W, H, C, n, true, nt);             // planar -> interleaved
        if (memcmp(back, in, isz * n)) { printf("REPACK ROUND TRIP MISMATCH\n"); return 1; }
        const int px = rnd(W * H), ch = rnd(C), im = rnd(n);
        if (pl[(size_t)im * isz + (size_t)ch * W * H + px] != in[(size_t)im * isz + (size_t)px * C + ch]) { printf("REPACK LAYOUT MISMATCH\n"); return 1; }
        free(pl); free(back);
        free(in); free(out); free(want); cases++;
    }
    printf("%d random cases clean\n", cases);
    return 0;
}
