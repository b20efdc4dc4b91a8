// Developer microbenchmark: what a dispatch costs on this box whatever the kernel does.
//   hipcc --offload-arch=gfx950 -O3 -o dispatch_floor dispatch_floor.hip && ./dispatch_floor
// For an empty kernel, a one-block load/store kernel and a 13.8 MB elementwise copy (the bytes of one
// batch of 35 256x256x3 images, read + written):
//   * the dispatch's own start->stop timestamps (hipExtLaunchKernelGGL events, the figure rocprofv3 reports),
//   * the serial issue period on one stream (wall clock / launches, untimed launches),
//   * the aggregate issue period with 4 streams fed by one host thread.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_empty() {}
__global__ void k_empty_args(const uint4 *, uint4 *, size_t) {}
struct Fat { long long a[12]; };
__global__ void k_empty_fat(Fat) {}
__global__ void k_tiny(const uint4 *in, uint4 *out) { out[threadIdx.x] = in[threadIdx.x]; }
__global__ __launch_bounds__(256) void k_copy(const uint4 *in, uint4 *out, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = in[i];
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

template <typename F>
static int run(const char *name, F launch)
{
    const int reps = 400;
    hipStream_t s[4];
    for (auto &x : s) CK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    std::vector<float> d;
    for (int i = 0; i < 60; i++) {
        launch(s[0], a, b);
        CK(hipStreamSynchronize(s[0]));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (i >= 10) d.push_back(ms * 1e3f);
    }
    std::sort(d.begin(), d.end());
    double t0 = now_us();
    for (int i = 0; i < reps; i++) launch(s[0], nullptr, nullptr);
    CK(hipStreamSynchronize(s[0]));
    const double serial = (now_us() - t0) / reps;
    t0 = now_us();
    for (int i = 0; i < reps; i++) launch(s[i & 3], nullptr, nullptr);
    for (auto &x : s) CK(hipStreamSynchronize(x));
    const double par = (now_us() - t0) / reps;
    printf("%-28s dispatch start->stop med %6.2f us (min %5.2f)   1 stream %6.2f us/launch   4 streams %6.2f us/launch\n",
           name, d[d.size() / 2], d[0], serial, par);
    for (auto &x : s) CK(hipStreamDestroy(x));
    return 0;
}

int main()
{
    const size_t bytes = 35ull * 256 * 256 * 3, n = bytes / 16;
    uint4 *in, *out;
    CK(hipMalloc((void **)&in, bytes)); CK(hipMalloc((void **)&out, bytes));
    CK(hipMemset(in, 1, bytes)); CK(hipMemset(out, 0, bytes));
    CK(hipDeviceSynchronize());
    run("empty kernel, 1 block", [&](hipStream_t s, hipEvent_t a, hipEvent_t b) {
        hipExtLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s, a, b, 0); });
    run("empty kernel, 2048 blocks", [&](hipStream_t s, hipEvent_t a, hipEvent_t b) {
        hipExtLaunchKernelGGL(k_empty, dim3(2048), dim3(256), 0, s, a, b, 0); });
    run("empty kernel + 3 args", [&](hipStream_t s, hipEvent_t a, hipEvent_t b) {
        if (a) hipExtLaunchKernelGGL(k_empty_args, dim3(1), dim3(64), 0, s, a, b, 0, in, out, n);
        else hipLaunchKernelGGL(k_empty_args, dim3(1), dim3(64), 0, s, in, out, n); });
    run("empty kernel + 96 B struct", [&](hipStream_t s, hipEvent_t a, hipEvent_t b) {
        Fat f{};
        if (a) hipExtLaunchKernelGGL(k_empty_fat, dim3(1), dim3(64), 0, s, a, b, 0, f);
        else hipLaunchKernelGGL(k_empty_fat, dim3(1), dim3(64), 0, s, f); });
    run("16 B/lane copy, 1 block, <<<>>>", [&](hipStream_t s, hipEvent_t a, hipEvent_t b) {
        if (a) hipExtLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, s, a, b, 0, in, out);
        else hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, s, in, out); });
    run("copy 6.9+6.9 MB, <<<>>>", [&](hipStream_t s, hipEvent_t a, hipEvent_t b) {
        if (a) hipExtLaunchKernelGGL(k_copy, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a, b, 0, in, out, n);
        else hipLaunchKernelGGL(k_copy, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, out, n); });
    run("16 B/lane copy, 1 block", [&](hipStream_t s, hipEvent_t a, hipEvent_t b) {
        hipExtLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, s, a, b, 0, in, out); });
    run("elementwise copy, 6.9+6.9 MB", [&](hipStream_t s, hipEvent_t a, hipEvent_t b) {
        hipExtLaunchKernelGGL(k_copy, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a, b, 0, in, out, n); });
    CK(hipFree(in)); CK(hipFree(out));
    return 0;
}
