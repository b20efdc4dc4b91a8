// Developer probe: the batch-35 stream (143 launches over S streams per 5000-image pass) issued directly vs captured
// once into a hipGraph and replayed.   graph_probe [streams=4] [passes=40]
#include "mi_blur.h"
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
#define MK(x) do { int r_ = (x); if (r_) { printf("%s: %s\n", #x, mi_blur_strerror(r_)); return 1; } } while (0)
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv)
{
    const int S = argc > 1 ? atoi(argv[1]) : 4, passes = argc > 2 ? atoi(argv[2]) : 40;
    const int W = 256, H = 256, C = 3, R = 1, N = 5000, B = 35;
    const size_t isz = (size_t)W * H * C;
    CK(hipSetDevice(0));
    uint8_t *din, *dout;
    CK(hipMalloc((void **)&din, isz * N)); CK(hipMalloc((void **)&dout, isz * N));
    CK(hipMemset(din, 7, isz * N));
    std::vector<hipStream_t> st(S);
    for (auto &s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    std::vector<hipEvent_t> ev(S);
    for (auto &e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    hipEvent_t fork; CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));

    auto issue = [&]() -> int {
        int k = 0;
        for (int i = 0; i < N; i += B, k++) {
            const int n = N - i < B ? N - i : B;
            MK(mi_blur_enqueue(din + i * isz, dout + i * isz, W, H, C, R, n, st[k % S]));
        }
        return 0;
    };
    // direct
    if (issue()) return 1;
    for (auto &s : st) CK(hipStreamSynchronize(s));
    double t0 = now_us();
    for (int p = 0; p < passes; p++) if (issue()) return 1;
    for (auto &s : st) CK(hipStreamSynchronize(s));
    const double direct = (now_us() - t0) / passes;
    printf("direct launches, %d streams: %8.1f us per 5000-image pass  %6.2f M img/s\n", S, direct, N / direct);

    // graph: capture on stream 0, fork the others, join back
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st[0], hipStreamCaptureModeRelaxed));
    CK(hipEventRecord(fork, st[0]));
    for (int s = 1; s < S; s++) CK(hipStreamWaitEvent(st[s], fork, 0));
    if (issue()) return 1;
    for (int s = 1; s < S; s++) { CK(hipEventRecord(ev[s], st[s])); CK(hipStreamWaitEvent(st[0], ev[s], 0)); }
    CK(hipStreamEndCapture(st[0], &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    size_t nn = 0; CK(hipGraphGetNodes(g, nullptr, &nn));
    CK(hipGraphLaunch(ge, st[0])); CK(hipStreamSynchronize(st[0]));
    t0 = now_us();
    for (int p = 0; p < passes; p++) CK(hipGraphLaunch(ge, st[0]));
    CK(hipStreamSynchronize(st[0]));
    const double graph = (now_us() - t0) / passes;
    printf("hipGraph replay (%zu nodes), %d streams: %8.1f us per pass  %6.2f M img/s\n", nn, S, graph, N / graph);
    return 0;
}
