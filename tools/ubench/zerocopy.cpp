// Developer probe: the blur kernel working directly on pinned host memory (zero-copy over PCIe) for different
// hipHostMalloc flags and kernel options, against the staged H2D -> kernel -> D2H path.
//   hipcc -O2 -I include -o tools/ubench/zerocopy tools/ubench/zerocopy.cpp -L <pkg> -lmi_blur -Wl,-rpath,<pkg>
#include "mi_blur.h"
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
#define MK(x) do { int r_ = (x); if (r_) { printf("%s: %s\n", #x, mi_blur_strerror(r_)); return 1; } } while (0)
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv)
{
    const int W = argc > 1 ? atoi(argv[1]) : 256, H = argc > 2 ? atoi(argv[2]) : 256, C = 3, R = 1;
    const int n = argc > 3 ? atoi(argv[3]) : 140, reps = 20;
    const size_t bytes = (size_t)n * W * H * C;
    CK(hipSetDevice(0));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    uint8_t *din, *dout; CK(hipMalloc((void **)&din, bytes)); CK(hipMalloc((void **)&dout, bytes));
    std::vector<uint8_t> want(bytes);

    struct Flag { const char *name; unsigned in, out; };
    const Flag all_flags[] = {
        {"default/default", hipHostMallocDefault, hipHostMallocDefault},
        {"noncoherent/noncoherent", hipHostMallocNonCoherent, hipHostMallocNonCoherent},
        {"coherent/coherent", hipHostMallocCoherent, hipHostMallocCoherent},
        {"writecombined-in/default", hipHostMallocWriteCombined, hipHostMallocDefault},
        {"noncoherent+wc-in/noncoherent", hipHostMallocNonCoherent | hipHostMallocWriteCombined, hipHostMallocNonCoherent},
    };
    const int nflags = argc > 4 ? atoi(argv[4]) : 5;      // 4th argument: how many flag sets to try
    std::vector<Flag> flags(all_flags, all_flags + (nflags < 5 ? nflags : 5));
    bool have_want = false;
    for (const Flag &f : flags) {
        uint8_t *hin = nullptr, *hout = nullptr;
        if (hipHostMalloc((void **)&hin, bytes, f.in) != hipSuccess || hipHostMalloc((void **)&hout, bytes, f.out) != hipSuccess) {
            printf("%-32s hipHostMalloc refused these flags\n", f.name); (void)hipGetLastError(); continue;
        }
        mi_blur_fill_synthetic(hin, W, H, C, 0, n, 8);
        memset(hout, 0, bytes);
        // staged
        auto staged = [&]() -> int {
            CK(hipMemcpyAsync(din, hin, bytes, hipMemcpyHostToDevice, st));
            MK(mi_blur_enqueue(din, dout, W, H, C, R, n, st));
            CK(hipMemcpyAsync(hout, dout, bytes, hipMemcpyDeviceToHost, st));
            return 0;
        };
        if (staged()) return 1;
        CK(hipStreamSynchronize(st));
        if (!have_want) { memcpy(want.data(), hout, bytes); have_want = true; }
        double t0 = now_us();
        for (int i = 0; i < reps; i++) if (staged()) return 1;
        CK(hipStreamSynchronize(st));
        const double t_staged = (now_us() - t0) / reps;
        printf("%-32s staged              %8.1f us/batch %7.1f k img/s\n", f.name, t_staged, n / t_staged * 1e3);
        struct Opt { const char *name; int rpt, stream, dma, xcd; };
        const Opt opts[] = {{"tiled auto", 0, 0, 1, 1}, {"tiled rows/thread 4", 4, 0, 1, 1}, {"tiled rows/thread 8", 8, 0, 1, 1},
                            {"tiled rows/thread 16", 16, 0, 1, 1}, {"stream variant", 0, 1, 1, 1},
                            {"tiled, register staging", 0, 0, 0, 1}, {"tiled, identity tile map", 0, 0, 1, 0},
                            {"tiled, reg staging + identity", 0, 0, 0, 0}};
        for (const Opt &o : opts) {
            mi_blur_set_option("rows_per_thread", o.rpt);
            mi_blur_set_option("prefer_stream", o.stream);
            mi_blur_set_option("stage_dma", o.dma);
            mi_blur_set_option("xcd_remap", o.xcd);
            memset(hout, 0, bytes);
            MK(mi_blur_enqueue(hin, hout, W, H, C, R, n, st));
            CK(hipStreamSynchronize(st));
            const bool ok = memcmp(hout, want.data(), bytes) == 0;
            t0 = now_us();
            for (int i = 0; i < reps; i++) MK(mi_blur_enqueue(hin, hout, W, H, C, R, n, st));
            CK(hipStreamSynchronize(st));
            const double t = (now_us() - t0) / reps;
            printf("%-32s zero-copy %-20s %8.1f us/batch %7.1f k img/s  %5.1f GB/s both ways  %s\n", f.name, o.name, t, n / t * 1e3,
                   2.0 * bytes / t / 1e3, ok ? "bit-exact" : "DIFFERS");
        }
        mi_blur_set_option("rows_per_thread", 0);
        mi_blur_set_option("prefer_stream", 0);
        mi_blur_set_option("stage_dma", 1);
        mi_blur_set_option("xcd_remap", 1);
        CK(hipHostFree(hin)); CK(hipHostFree(hout));
    }
    return 0;
}
