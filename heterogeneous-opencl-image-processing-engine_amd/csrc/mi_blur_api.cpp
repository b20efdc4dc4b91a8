// mi_blur_api.cpp — implementation of the C ABI declared in include/mi_blur.h.
//
// Each block names the reference OpenCL plumbing it replaces (paths relative to the
// reference tree).  No torch, no OpenCL; HIP runtime + (lazily dlopen'ed) RCCL only.
#include "../../include/mi_blur.h"
#include "blur_launch.h"
#include "cpu_device.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dlfcn.h>
#include <sched.h>
#include <time.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <deque>
#include <functional>
#include <condition_variable>
#include <thread>
#include <vector>

using namespace mi_blur;

#define HIP_TRY(expr)                                                       \
    do {                                                                    \
        hipError_t e_ = (expr);                                             \
        if (e_ != hipSuccess) { (void)hipGetLastError(); return MI_BLUR_ERR_HIP_BASE - (int)e_; } \
    } while (0)

// Runtime default, applied when the library is loaded and only if the user has not set it: keep kernel
// arguments in host memory (HIP_FORCE_DEV_KERNARG=0).  With the runtime's gfx9 default (a device-memory
// kernarg pool written over PCIe) a stream of argument-carrying launches issues at ~3.3 us per launch on this
// platform, with host kernargs at ~1.6 us (tools/ubench/dispatch_floor.hip); the batch-35 stream is bounded by
// that rate, not by the kernel (+30 % images/s measured).  The flag is read when the HIP
// runtime initialises, so it takes effect if this library (or the Python package, which sets it too) is
// loaded before the process's first HIP call.
__attribute__((constructor)) static void mi_blur_runtime_defaults() { setenv("HIP_FORCE_DEV_KERNARG", "0", 0); }

// ----------------------------------------------------------------------------------
// status / discovery   (replaces cl_error strings + heterogeneous_blur.c:142-184)
// ----------------------------------------------------------------------------------
extern "C" const char *mi_blur_strerror(int status)
{
    static thread_local char buf[160];
    switch (status) {
    case MI_BLUR_OK: return "success";
    case MI_BLUR_ERR_INVALID: return "invalid argument";
    case MI_BLUR_ERR_NO_DEVICE: return "no usable HIP device";
    case MI_BLUR_ERR_NOMEM: return "out of memory";
    case MI_BLUR_ERR_STATE: return "call not valid in this state";
    case MI_BLUR_ERR_UNSUPPORTED: return "not supported in this build/environment";
    }
    if (status <= MI_BLUR_ERR_RCCL_BASE && status > MI_BLUR_ERR_RCCL_BASE - 100) {
        snprintf(buf, sizeof buf, "RCCL error %d", MI_BLUR_ERR_RCCL_BASE - status);
        return buf;
    }
    if (status <= MI_BLUR_ERR_HIP_BASE) {
        const hipError_t e = (hipError_t)(MI_BLUR_ERR_HIP_BASE - status);
        snprintf(buf, sizeof buf, "HIP error %d: %s", (int)e, hipGetErrorString(e));
        return buf;
    }
    snprintf(buf, sizeof buf, "unknown status %d", status);
    return buf;
}

extern "C" int mi_blur_version(void) { return MI_BLUR_VERSION; }

extern "C" const char *mi_blur_last_kernel(void) { return last_kernel(); }

extern "C" int mi_blur_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

extern "C" int mi_blur_set_option(const char *key, int value)
{
    if (!key) return MI_BLUR_ERR_INVALID;
    Tunables t = tunables();      // copy, edit, publish: launches in other threads see the old set or the new one
    if (!strcmp(key, "stage_dma")) t.stage_dma = value != 0;
    else if (!strcmp(key, "rows_per_thread")) { if (value != 0 && value != 4 && value != 8 && value != 16) return MI_BLUR_ERR_INVALID; t.rpg = value; }
    else if (!strcmp(key, "xcd_remap")) t.xcd_remap = value != 0;
    else if (!strcmp(key, "debug_copy")) t.debug_copy = value != 0;   // ablation only (output is NOT a blur)
    else if (!strcmp(key, "row_shuffle")) t.row_shuffle = value != 0;
    else if (!strcmp(key, "prefer_stream")) t.prefer_stream = value != 0;
    else if (!strcmp(key, "zero_copy")) t.zero_copy = value != 0;
    else if (!strcmp(key, "ragged_tiled")) t.ragged = value != 0;
    else if (!strcmp(key, "stream_band_rows")) { if (value < 0 || value > 4096) return MI_BLUR_ERR_INVALID; t.stream_bh = value; }
    else if (!strcmp(key, "fused_release")) t.fused_release = value != 0;
    else if (!strcmp(key, "experiment")) t.experiment = value != 0;
    else if (!strcmp(key, "stream_updown")) t.stream_updown = value != 0;
    else if (!strcmp(key, "zero_copy_streams")) { if (value < 1 || value > 8) return MI_BLUR_ERR_INVALID; t.zero_copy_streams = value; }
    else if (!strcmp(key, "zero_copy_blocks")) { if (value < 0 || value > (1 << 20)) return MI_BLUR_ERR_INVALID; t.zero_copy_blocks = value; }
    else if (!strcmp(key, "prefer_direct")) { if (value < 0 || value > 2) return MI_BLUR_ERR_INVALID; t.prefer_direct = value; }
    else if (!strcmp(key, "direct_bh")) { if (value != 4 && value != 8 && value != 12 && value != 16) return MI_BLUR_ERR_INVALID; t.direct_bh = value; }
    else if (!strcmp(key, "fused_adds_per_word")) { if (value < 4 || value > 4096) return MI_BLUR_ERR_INVALID; t.fused_adds_per_word = value; }
    else if (!strcmp(key, "fused_tail_blocks")) { if (value < 10 || value > 800) return MI_BLUR_ERR_INVALID; t.fused_tail_blocks = value; }
    else if (!strcmp(key, "fused_tail")) { if (value < 0 || value > 500) return MI_BLUR_ERR_INVALID; t.fused_tail = value; }
    else if (!strcmp(key, "fused_window")) { if (value < 1 || value > 4096) return MI_BLUR_ERR_INVALID; t.fused_window = value; }
    else if (!strcmp(key, "debug_xcd_times")) t.debug_xcd_times = value != 0;
    else if (!strcmp(key, "zero_copy_events")) t.zero_copy_events = value != 0;
    else if (!strcmp(key, "zero_copy_server")) t.zero_copy_server = value != 0;
    else if (!strcmp(key, "staged_server")) t.staged_server = value != 0;
    else if (!strcmp(key, "zero_copy_server_min_kb")) { if (value < 0 || value > (1 << 20)) return MI_BLUR_ERR_INVALID; t.zero_copy_server_min_kb = value; }
    else if (!strcmp(key, "zero_copy_trace")) t.zero_copy_trace = value != 0;
    else if (!strcmp(key, "zero_copy_tickets")) t.zero_copy_tickets = value != 0;
    else if (!strcmp(key, "zero_copy_spin")) t.zero_copy_spin = value != 0;
    else if (!strcmp(key, "zero_copy_debug_base")) { if (value < 0 || value > (1 << 20)) return MI_BLUR_ERR_INVALID; t.zero_copy_debug_base = value; }
    else if (!strcmp(key, "resident_place_trials")) { if (value < 0 || value > 8) return MI_BLUR_ERR_INVALID; t.resident_place_trials = value; }
    else if (!strcmp(key, "zero_copy_workers")) { if (value < 1 || value > 2048) return MI_BLUR_ERR_INVALID; t.zero_copy_workers = value; }
    else if (!strcmp(key, "zero_copy_idle_us")) { if (value < 10 || value > 100000) return MI_BLUR_ERR_INVALID; t.zero_copy_idle_us = value; }
    else if (!strcmp(key, "zero_copy_budget")) { if (value < 1 || value > (1 << 20)) return MI_BLUR_ERR_INVALID; t.zero_copy_budget = value; }
    else if (!strcmp(key, "xcd_run")) { if (value < 0 || value > (1 << 20)) return MI_BLUR_ERR_INVALID; t.xcd_run = value; }
    else return MI_BLUR_ERR_INVALID;
    set_tunables(t);
    return MI_BLUR_OK;
}

// ----------------------------------------------------------------------------------
// kernel level   (clSetKernelArg x5 + clEnqueueNDRangeKernel)
// ----------------------------------------------------------------------------------
extern "C" int mi_blur_enqueue_ex(const uint8_t *d_in, uint8_t *d_out, int width, int band_rows, int channels,
                                  int radius, int n_images, int out_row_begin, int out_row_end, int variant,
                                  void *stream)
{
    if (mi_blur_device_count() <= 0) return MI_BLUR_ERR_NO_DEVICE;
    LaunchDesc d{};
    d.in = d_in; d.out = d_out; d.width = width; d.band_rows = band_rows; d.channels = channels;
    d.radius = radius; d.n_images = n_images; d.y0 = out_row_begin; d.y1 = out_row_end;
    d.variant = variant; d.stream = (hipStream_t)stream;
    return launch(d);
}

extern "C" int mi_blur_enqueue(const uint8_t *d_in, uint8_t *d_out, int width, int height, int channels,
                               int radius, int n_images, void *stream)
{
    return mi_blur_enqueue_ex(d_in, d_out, width, height, channels, radius, n_images, 0, height,
                              MI_BLUR_VARIANT_AUTO, stream);
}

extern "C" int mi_blur_enqueue_band(const uint8_t *d_in, uint8_t *d_out, int width, int band_rows, int channels,
                                    int radius, int out_row_begin, int out_row_end, void *stream)
{
    return mi_blur_enqueue_ex(d_in, d_out, width, band_rows, channels, radius, 1, out_row_begin, out_row_end,
                              MI_BLUR_VARIANT_AUTO, stream);
}

// Band whose halo rows are read in place from the neighbouring shards (peer memory): exchange and blur in one launch.
extern "C" int mi_blur_enqueue_band_peer(const uint8_t *d_in, uint8_t *d_out, int width, int band_rows, int channels,
                                         int radius, int out_row_begin, int out_row_end, const uint8_t *top_src,
                                         const uint8_t *bottom_src, void *stream)
{
    if (mi_blur_device_count() <= 0) return MI_BLUR_ERR_NO_DEVICE;
    if (!top_src && !bottom_src)
        return mi_blur_enqueue_band(d_in, d_out, width, band_rows, channels, radius, out_row_begin, out_row_end, stream);
    LaunchDesc d{};
    d.in = d_in; d.out = d_out; d.width = width; d.band_rows = band_rows; d.channels = channels;
    d.radius = radius; d.n_images = 1; d.y0 = out_row_begin; d.y1 = out_row_end;
    d.variant = MI_BLUR_VARIANT_AUTO; d.stream = (hipStream_t)stream;
    d.halo_top = top_src; d.halo_bottom = bottom_src;
    return launch(d);
}

// Frame layout on the device (replaces the host loops heterogeneous_blur.c:125-134 and split_image_blur.c:40-56).
extern "C" int mi_blur_planar_to_interleaved(const uint8_t *d_planar, uint8_t *d_interleaved, int width, int height,
                                             int channels, int n_images, void *stream)
{
    if (mi_blur_device_count() <= 0) return MI_BLUR_ERR_NO_DEVICE;
    return launch_planar_to_interleaved(d_planar, d_interleaved, width, height, channels, n_images, (hipStream_t)stream);
}

extern "C" int mi_blur_interleaved_to_planar(const uint8_t *d_interleaved, uint8_t *d_planar, int width, int height,
                                             int channels, int n_images, void *stream)
{
    if (mi_blur_device_count() <= 0) return MI_BLUR_ERR_NO_DEVICE;
    return launch_interleaved_to_planar(d_interleaved, d_planar, width, height, channels, n_images, (hipStream_t)stream);
}

// ----------------------------------------------------------------------------------
// queue level
// ----------------------------------------------------------------------------------
namespace {

static int staging_threads_env()
{
    const char *e = getenv("MI_BLUR_STAGING_THREADS");
    const int n = e ? atoi(e) : 0;
    return n >= 1 && n <= 64 ? n : 8;
}
// pageable caller memory <-> pinned staging: one thread moves ~10 GB/s, the link wants ~45 GB/s each way (8 threads: 220 k
// img/s at batch 35 and 500; 4: 145-180 k; MI_BLUR_STAGING_THREADS overrides)
static const int STAGING_COPY_THREADS = staging_threads_env();

struct Slot {
    hipStream_t stream = nullptr;
    uint8_t *h_in = nullptr, *h_out = nullptr;   // pinned staging (used when the caller's memory is pageable)
    uint8_t *d_in = nullptr, *d_out = nullptr;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};   // H2D begin/end, D2H begin/end
    hipEvent_t ks = nullptr, ke = nullptr;                     // kernel dispatch start/stop
    bool busy = false;
    uint8_t *user_out = nullptr;
    size_t out_bytes = 0, out_band = 0, out_stride = 0;   // pending read-back: out_n bands of out_band bytes
    int out_n = 0;
    bool out_staged = false;
    bool zero_copy = false;                               // in flight on the context's zero-copy stream
    bool zc_plain = false;                                // ... as a plain launch without events (experiment)
    hipStream_t zc_stream = nullptr;
    bool zc_server = false;                               // ... as batch zc_index of the context's batch server
    unsigned zc_index = 0;
};

struct TimedLaunch { hipEvent_t s, e; };

// Host side of the zero-copy batch server (blur_launch.h): the descriptor ring in pinned memory, the device-side
// hand-off words, the stream the servers queue on, and what has been published / launched so far.
struct ZcServer {
    ZcHostCtl *ctl = nullptr, *ctl_dev = nullptr;          // pinned host memory and its device address
    ZcDevCtl *dev = nullptr;
    hipStream_t stream = nullptr;
    ZcGeometry geo{};
    unsigned head = 0, launched = 0, tile_base = 0;
    unsigned n_workers = 96, budget = 256, idle_ticks = 30000;
    unsigned long long covered = 0;                      // kernel bucket: the union of [t_begin, t_end] so far reaches this tick
    unsigned long long *trace = nullptr;                 // diagnostics ("zero_copy_trace"): device buffer of per-worker phase stamps
    int fixed_share = 0;                                 // A/B: "zero_copy_tickets" 0
};

// The CPU device's queue: submits run one after another on ONE long-lived thread per context (each of them spreads over the
// worker pool inside cpu_blur_batch) — a thread per submit cost ~60 us, most of a small batch's time.
struct CpuJob {
    std::function<void()> work;
    double ms = 0.0;
    bool done = false;
};
struct CpuWorker {
    std::thread th;
    std::mutex m;
    std::condition_variable cv_work, cv_done;
    std::deque<CpuJob *> q;
    bool stop = false;
    void loop()
    {
        for (;;) {
            CpuJob *j = nullptr;
            {
                std::unique_lock<std::mutex> lk(m);
                cv_work.wait(lk, [&] { return stop || !q.empty(); });
                if (q.empty()) return;                   // stop, nothing left
                j = q.front(); q.pop_front();
            }
            const auto t0 = std::chrono::steady_clock::now();
            j->work();
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            { std::lock_guard<std::mutex> lk(m); j->ms = ms; j->done = true; }
            cv_done.notify_all();
        }
    }
    void push(CpuJob *j)
    {
        { std::lock_guard<std::mutex> lk(m); q.push_back(j); }
        if (!th.joinable()) th = std::thread([this] { loop(); });
        cv_work.notify_one();
    }
    void wait(CpuJob *j)
    {
        std::unique_lock<std::mutex> lk(m);
        cv_done.wait(lk, [&] { return j->done; });
    }
    ~CpuWorker()
    {
        { std::lock_guard<std::mutex> lk(m); stop = true; }
        cv_work.notify_all();
        if (th.joinable()) th.join();
    }
};

}  // namespace

struct mi_blur_ctx {
    int device = 0, W = 0, H = 0, C = 0, R = 1, max_batch = 0, n_threads = 0;
    size_t image_bytes = 0;
    std::vector<Slot> slots;
    int next_slot = 0;
    mi_blur_timing tm{};
    // resident pool
    uint8_t *pool_in = nullptr, *pool_out = nullptr;
    int pool_images = 0;
    long long cursor = 0;
    int rr = 0;
    std::vector<TimedLaunch> ev_pool;
    size_t ev_used = 0;
    uint64_t timed_launches = 0, timed_bytes_alg = 0;   // resident launches that carried timestamp events
    uint64_t zero_copy_launches = 0;
    // Zero-copy launches of one context may overlap (several streams).  Their kernel bucket is the time during which at
    // least ONE of them was executing — the union of the dispatch intervals, measured against a reference event recorded
    // before the first of them since the last sync — not the sum of overlapping durations.
    hipEvent_t zc_ref = nullptr;
    bool zc_ref_valid = false;
    double zc_covered_ms = 0.0;                                  // the union so far reaches this far past zc_ref
    // fused stream: per-batch completion counters (device) and flags (pinned host memory)
    unsigned *fused_count = nullptr, *fused_host = nullptr;    // device counters; pinned host copy for polling
    int fused_cap = 0, fused_batches = 0;
    unsigned fused_tpb = 0, fused_wpb = 0, fused_blocks = 0;     // geometry of the latest fused pass
    unsigned fused_k = 8;                                        // ... and how many counters each of its batches uses (8 .. 256)
    int fused_n = 0, fused_batch = 0;                            // its (n_images, batch): same again = counters keep counting up
    unsigned fused_passes = 0;                                   // passes accumulated in the counters since they were zeroed
    hipStream_t fused_poll = nullptr;
    hipStream_t fused_watch = nullptr;                           // the watcher's own stream (peeks on fused_poll must not queue behind it)
    unsigned long long *fused_word = nullptr, *fused_word_dev = nullptr;   // watcher's progress word: pinned host memory + its device address
    unsigned fused_watch_seq = 0;                                // sequence number of the latest watched pass (0 = none)
    bool fused_watched = false;                                  // the latest pass has a watcher
    ZcServer *zc = nullptr;                                      // batch server, made on the first submit that can use it
    std::vector<float> place_ms;                                 // resident pool: per-launch ms of each candidate placement tried
    int place_kept = 0;
    // CPU device
    std::vector<CpuJob *> cpu_jobs;
    CpuWorker *cpu_worker = nullptr;
    bool is_cpu() const { return device == MI_BLUR_DEVICE_CPU; }
};

static bool is_pinned(const void *p)
{
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeHost;
}

// The device-side address of pinned (hipHostMalloc'd or registered) host memory, nullptr for anything else.
static uint8_t *pinned_device_ptr(const void *p)
{
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return a.type == hipMemoryTypeHost ? (uint8_t *)a.devicePointer : nullptr;
}

extern "C" void *mi_blur_host_alloc(size_t bytes)
{
    void *p = nullptr;
    if (mi_blur_device_count() > 0) {
        // portable + mapped: visible to every GPU's contexts (A1 hands one batch buffer to G devices, each of which
        // may work on its share in place)
        if (hipHostMalloc(&p, bytes, hipHostMallocPortable | hipHostMallocMapped) == hipSuccess) return p;
        (void)hipGetLastError();
        return nullptr;
    }
    return malloc(bytes);   // CPU-only operation: ordinary memory
}

extern "C" int mi_blur_host_register(void *p, size_t bytes)
{
    if (!p || bytes == 0) return MI_BLUR_ERR_INVALID;
    if (mi_blur_device_count() <= 0) return MI_BLUR_OK;
    HIP_TRY(hipHostRegister(p, bytes, hipHostRegisterPortable | hipHostRegisterMapped));
    return MI_BLUR_OK;
}

extern "C" int mi_blur_host_unregister(void *p)
{
    if (!p) return MI_BLUR_ERR_INVALID;
    if (mi_blur_device_count() <= 0) return MI_BLUR_OK;
    HIP_TRY(hipHostUnregister(p));
    return MI_BLUR_OK;
}

extern "C" void mi_blur_host_free(void *p)
{
    if (!p) return;
    if (mi_blur_device_count() > 0 && is_pinned(p)) { (void)hipHostFree(p); return; }
    free(p);
}

// ----------------------------------------------------------------------------------
// NUMA placement of per-GPU host work.  The reference drives both of its devices from one
// host thread on a one-socket desktop (heterogeneous_blur.c:482-539); an 8-GPU MI355X node has two sockets with four
// GPUs each, and a feeder thread, a batch-building memcpy or a pinned buffer on the other socket puts every byte of
// the stream on the inter-socket link first.  No libnuma: the GPU's PCI bus id -> sysfs local_cpulist -> sched_setaffinity.
// ----------------------------------------------------------------------------------
static bool affinity_disabled()
{
    const char *e = getenv("MI_BLUR_NO_AFFINITY");
    return e && atoi(e) != 0;
}

// "0-63,128-191" -> cpu_set_t; false when the text holds no CPU
static bool parse_cpulist(const char *text, cpu_set_t *set)
{
    CPU_ZERO(set);
    bool any = false;
    for (const char *p = text; *p;) {
        char *end = nullptr;
        long a = strtol(p, &end, 10);
        if (end == p) break;
        long b = a;
        if (*end == '-') { p = end + 1; b = strtol(p, &end, 10); if (end == p) break; }
        for (long c = a; c <= b && c < CPU_SETSIZE; c++) if (c >= 0) { CPU_SET((int)c, set); any = true; }
        p = end;
        while (*p == ',' || *p == ' ' || *p == '\n') p++;
    }
    return any;
}

static int device_sysfs(int device, const char *leaf, char *buf, size_t n)
{
    if (!buf || n == 0) return MI_BLUR_ERR_INVALID;
    buf[0] = 0;
    const int ndev = mi_blur_device_count();
    if (ndev <= 0 || device < 0 || device >= ndev) return MI_BLUR_ERR_NO_DEVICE;
    char bdf[64] = {0};
    HIP_TRY(hipDeviceGetPCIBusId(bdf, (int)sizeof bdf, device));
    for (char *q = bdf; *q; q++) if (*q >= 'A' && *q <= 'F') *q = (char)(*q - 'A' + 'a');     // sysfs spells it in lower case
    char path[160];
    snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/%s", bdf, leaf);
    FILE *f = fopen(path, "r");
    if (!f) return MI_BLUR_ERR_UNSUPPORTED;
    const bool ok = fgets(buf, (int)n, f) != nullptr;
    fclose(f);
    if (!ok) { buf[0] = 0; return MI_BLUR_ERR_UNSUPPORTED; }
    for (size_t i = strlen(buf); i > 0 && (buf[i - 1] == '\n' || buf[i - 1] == ' '); i--) buf[i - 1] = 0;
    return MI_BLUR_OK;
}

extern "C" int mi_blur_device_cpulist(int device, char *buf, size_t n, int *numa_node)
{
    if (numa_node) {
        char nb[32];
        *numa_node = device_sysfs(device, "numa_node", nb, sizeof nb) == MI_BLUR_OK ? atoi(nb) : -1;
    }
    return device_sysfs(device, "local_cpulist", buf, n);
}

extern "C" int mi_blur_bind_thread_to_device(int device)
{
    if (affinity_disabled()) return 0;
    char list[512];
    if (device_sysfs(device, "local_cpulist", list, sizeof list) != MI_BLUR_OK) return 0;
    cpu_set_t local, allowed, both;
    if (!parse_cpulist(list, &local)) return 0;
    if (sched_getaffinity(0, sizeof allowed, &allowed) != 0) return 0;
    CPU_AND(&both, &local, &allowed);
    const int n = CPU_COUNT(&both);
    if (n == 0) return 0;                                        // a cpuset that excludes the GPU's socket: leave the thread alone
    if (sched_setaffinity(0, sizeof both, &both) != 0) return 0;
    return n;
}

extern "C" void *mi_blur_host_alloc_on(int device, size_t bytes)
{
    if (mi_blur_device_count() <= 0) return malloc(bytes);
    int prev = 0;
    (void)hipGetDevice(&prev);
    cpu_set_t saved;
    const bool have_saved = sched_getaffinity(0, sizeof saved, &saved) == 0;
    // pin + map with the GPU current and from one of its socket's CPUs: the runtime takes the pages from the host pool
    // nearest the current device, and whatever it leaves to first touch is touched from there as well
    if (hipSetDevice(device) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    const int bound = mi_blur_bind_thread_to_device(device);
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocPortable | hipHostMallocMapped) != hipSuccess) { (void)hipGetLastError(); p = nullptr; }
    if (bound > 0 && have_saved) (void)sched_setaffinity(0, sizeof saved, &saved);
    (void)hipSetDevice(prev);
    return p;
}

static void free_slot(Slot &s)
{
    if (s.h_in) (void)hipHostFree(s.h_in);
    if (s.h_out) (void)hipHostFree(s.h_out);
    if (s.d_in) (void)hipFree(s.d_in);
    if (s.d_out) (void)hipFree(s.d_out);
    for (auto &e : s.ev) if (e) (void)hipEventDestroy(e);
    if (s.ks) (void)hipEventDestroy(s.ks);
    if (s.ke) (void)hipEventDestroy(s.ke);
    if (s.stream) (void)hipStreamDestroy(s.stream);
    s = Slot{};
}

// Context + queue + buffers: heterogeneous_blur.c:194-212 (contexts, profiling queues),
// :341-354 (one in + one out buffer per device).  Here: n_slots streams, each with a
// max_batch-image device buffer pair, so one launch covers a whole batch.
extern "C" int mi_blur_create(mi_blur_ctx **out_ctx, int device, int width, int height, int channels, int radius,
                              int max_batch, int n_slots, int n_threads)
{
    if (!out_ctx) return MI_BLUR_ERR_INVALID;
    *out_ctx = nullptr;
    if (width <= 0 || height <= 0 || channels <= 0 || max_batch <= 0 || n_slots <= 0) return MI_BLUR_ERR_INVALID;
    if (radius != 1 && radius != 2) return MI_BLUR_ERR_INVALID;
    if ((long long)width * channels * height > 0x7fffffffLL) return MI_BLUR_ERR_INVALID;
    mi_blur_ctx *c = new (std::nothrow) mi_blur_ctx;
    if (!c) return MI_BLUR_ERR_NOMEM;
    c->device = device; c->W = width; c->H = height; c->C = channels; c->R = radius;
    c->max_batch = max_batch;
    // 0 = automatic: all cores up to 16.  Waking hundreds of workers for a ~1 ms batch costs more than it returns
    // (256-thread host, 256x256x3, batch 35: 8.5 k img/s with 256 workers, 235 k with 16, 190 k with 32); ask explicitly for more.
    c->n_threads = n_threads > 0 ? n_threads : std::min(hardware_threads(), 16);
    c->image_bytes = (size_t)width * height * channels;
    if (device == MI_BLUR_DEVICE_CPU) { *out_ctx = c; return MI_BLUR_OK; }

    const int ndev = mi_blur_device_count();
    if (ndev <= 0 || device < 0 || device >= ndev) { delete c; return MI_BLUR_ERR_NO_DEVICE; }
    int rc = MI_BLUR_OK;
    auto fail = [&](hipError_t e) { (void)hipGetLastError(); rc = MI_BLUR_ERR_HIP_BASE - (int)e; return true; };
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) { fail(e); delete c; return rc; }
    c->slots.resize(n_slots);
    for (auto &s : c->slots) {
        if ((e = hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking)) != hipSuccess && fail(e)) break;
        // the slot's pinned staging pair and device pair (max_batch images each) are made by the first submit that needs them
        // (slot_staging / slot_device): a context whose submits are all in place — pinned caller buffers, the recommended
        // form — never allocates them (a batch of 5000 320x240 frames on 4 slots: 18 GB and ~4 s of set-up otherwise)
        for (auto &ev : s.ev) if ((e = hipEventCreate(&ev)) != hipSuccess && fail(e)) break;
        if (rc) break;
        if ((e = hipEventCreate(&s.ks)) != hipSuccess && fail(e)) break;
        if ((e = hipEventCreate(&s.ke)) != hipSuccess && fail(e)) break;
    }
    if (rc) { mi_blur_destroy(c); return rc; }
    *out_ctx = c;
    return MI_BLUR_OK;
}

// Lazily made slot buffers (see mi_blur_create).  HIP's allocation calls are synchronous with respect to the device only
// where they must be; the slot's streams carry nothing that touches these buffers before they exist.
static int slot_staging(mi_blur_ctx *c, Slot &s, bool in, bool out)
{
    const size_t bytes = c->image_bytes * (size_t)c->max_batch;
    if (in && !s.h_in) HIP_TRY(hipHostMalloc((void **)&s.h_in, bytes, hipHostMallocDefault));
    if (out && !s.h_out) HIP_TRY(hipHostMalloc((void **)&s.h_out, bytes, hipHostMallocDefault));
    return MI_BLUR_OK;
}
static int slot_device(mi_blur_ctx *c, Slot &s)
{
    const size_t bytes = c->image_bytes * (size_t)c->max_batch;
    if (!s.d_in) HIP_TRY(hipMalloc((void **)&s.d_in, bytes));
    if (!s.d_out) HIP_TRY(hipMalloc((void **)&s.d_out, bytes));
    return MI_BLUR_OK;
}

static int finish_slot(mi_blur_ctx *c, Slot &s)
{
    if (!s.busy) return MI_BLUR_OK;
    float ms = 0.f;
    if (s.zero_copy && s.zc_server) {                            // batch zc_index of the batch server: wait for its done word
        ZcServer &z = *c->zc;
        const unsigned slot = s.zc_index % ZC_RING, want = s.zc_index + 1u;
        const auto t0 = std::chrono::steady_clock::now();
        // Waiting is a feeder's normal state (a batch takes ~160 us of link time): spin for ~20 us — a batch that is nearly
        // done should be noticed at once — then sleep in short steps so that the core is free for the threads that build
        // the next batch; every few ms ask the stream whether the device is still alive.
        const bool spin_only = tunables().zero_copy_spin != 0;
        for (unsigned spins = 0; __atomic_load_n(&z.ctl->done[slot], __ATOMIC_ACQUIRE) != want; spins++) {
            if (spins < 2048) { __builtin_ia32_pause(); continue; }
            if (__atomic_load_n(&z.ctl->error, __ATOMIC_ACQUIRE)) return MI_BLUR_ERR_STATE;   // a server gave up on a wait (see blur_server_kernel)
            if (spin_only) std::this_thread::yield();
            else { struct timespec ts = {0, 20000}; nanosleep(&ts, nullptr); }
            if ((spins & 63u) == 0) {                            // a faulted device never writes the word: ask the stream now and then
                const hipError_t q = hipStreamQuery(z.stream);
                if (q != hipSuccess && q != hipErrorNotReady) { (void)hipGetLastError(); return MI_BLUR_ERR_HIP_BASE - (int)q; }
                (void)hipGetLastError();
                if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(30)) return MI_BLUR_ERR_STATE;
            }
        }
        const unsigned long long b = __atomic_load_n(&z.ctl->t_begin[slot], __ATOMIC_RELAXED), e = __atomic_load_n(&z.ctl->t_end[slot], __ATOMIC_RELAXED);
        if (e >= b) {                                            // kernel bucket = union of the batches' [begin, end] (100 MHz device clock)
            const unsigned long long from = std::max(b, z.covered);
            if (e > from) c->tm.kernel_ms += (double)(e - from) / 1e5;
            z.covered = std::max(z.covered, e);
        }
        if (s.out_staged)                                        // the server wrote the slot's staging: scatter to the caller's memory
            copy_blocks(s.user_out, s.out_stride, s.h_out, s.out_band, s.out_band, s.out_n, STAGING_COPY_THREADS);
        s.zero_copy = false; s.zc_server = false; s.busy = false;
        return MI_BLUR_OK;
    }
    if (s.zero_copy) {                                           // launched on one of the context's zero-copy streams
        if (s.zc_plain) {                                        // experiment: no events on the dispatch
            HIP_TRY(hipStreamSynchronize(s.zc_stream));
            s.zero_copy = false; s.busy = false; s.zc_plain = false;
            return MI_BLUR_OK;
        }
        HIP_TRY(hipEventSynchronize(s.ke));
        float a = 0.f, b = 0.f;
        if (c->zc_ref_valid && hipEventElapsedTime(&a, c->zc_ref, s.ks) == hipSuccess && hipEventElapsedTime(&b, c->zc_ref, s.ke) == hipSuccess && b >= a) {
            const double from = std::max((double)a, c->zc_covered_ms);
            if ((double)b > from) c->tm.kernel_ms += (double)b - from;
            c->zc_covered_ms = std::max(c->zc_covered_ms, (double)b);
        } else if (hipEventElapsedTime(&ms, s.ks, s.ke) == hipSuccess) {
            c->tm.kernel_ms += ms;
        }
        (void)hipGetLastError();
        s.zero_copy = false;
        s.busy = false;
        return MI_BLUR_OK;
    }
    HIP_TRY(hipStreamSynchronize(s.stream));
    if (s.out_staged)
        copy_blocks(s.user_out, s.out_stride, s.h_out, s.out_band, s.out_band, s.out_n, STAGING_COPY_THREADS);
    if (hipEventElapsedTime(&ms, s.ev[0], s.ev[1]) == hipSuccess) c->tm.h2d_ms += ms;
    if (hipEventElapsedTime(&ms, s.ks, s.ke) == hipSuccess) c->tm.kernel_ms += ms;
    if (hipEventElapsedTime(&ms, s.ev[2], s.ev[3]) == hipSuccess) c->tm.d2h_ms += ms;
    (void)hipGetLastError();
    s.busy = false;
    return MI_BLUR_OK;
}

static int harvest_resident(mi_blur_ctx *c)
{
    for (size_t i = 0; i < c->ev_used; i++) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, c->ev_pool[i].s, c->ev_pool[i].e) == hipSuccess) c->tm.kernel_ms += ms;
    }
    (void)hipGetLastError();
    c->ev_used = 0;
    return MI_BLUR_OK;
}

// clFinish + clGetEventProfilingInfo harvest (heterogeneous_blur.c:538-579).
extern "C" int mi_blur_sync(mi_blur_ctx *c, mi_blur_timing *timing)
{
    if (!c) return MI_BLUR_ERR_INVALID;
    if (c->is_cpu()) {
        for (CpuJob *j : c->cpu_jobs) {
            c->cpu_worker->wait(j);
            c->tm.kernel_ms += j->ms;
            delete j;
        }
        c->cpu_jobs.clear();
    } else {
        HIP_TRY(hipSetDevice(c->device));
        for (auto &s : c->slots) {
            int rc = finish_slot(c, s);
            if (rc) return rc;
        }
        for (auto &s : c->slots) HIP_TRY(hipStreamSynchronize(s.stream));
        // a watched fused pass: its watcher (own stream) ends with the pass; after a sync the count it published is final
        if (c->fused_watched && c->fused_watch) HIP_TRY(hipStreamSynchronize(c->fused_watch));
        harvest_resident(c);
        c->zc_ref_valid = false;                                 // everything drained: the next zero-copy launch starts a new window
    }
    if (timing) *timing = c->tm;
    return MI_BLUR_OK;
}

extern "C" int mi_blur_get_timing(mi_blur_ctx *c, mi_blur_timing *timing)
{
    if (!c || !timing) return MI_BLUR_ERR_INVALID;
    *timing = c->tm;
    return MI_BLUR_OK;
}

extern "C" uint64_t mi_blur_zero_copy_launches(mi_blur_ctx *c) { return c ? c->zero_copy_launches : 0; }

extern "C" void mi_blur_reset_timing(mi_blur_ctx *c)
{
    if (!c) return;
    c->tm = mi_blur_timing{};
    c->timed_launches = 0; c->timed_bytes_alg = 0;
}

// For resident runs that time only every n-th launch: how many launches (and how many
// algorithmic bytes) the accumulated kernel_ms covers.  Submit paths time every launch.
extern "C" void mi_blur_timed_coverage(mi_blur_ctx *c, uint64_t *launches, uint64_t *bytes_alg)
{
    if (launches) *launches = c ? c->timed_launches : 0;
    if (bytes_alg) *bytes_alg = c ? c->timed_bytes_alg : 0;
}

// Cleanup: heterogeneous_blur.c:727-744.
extern "C" void mi_blur_destroy(mi_blur_ctx *c)
{
    if (!c) return;
    for (CpuJob *j : c->cpu_jobs) { c->cpu_worker->wait(j); delete j; }
    delete c->cpu_worker;
    if (!c->is_cpu()) {
        (void)hipSetDevice(c->device);
        for (auto &s : c->slots) { if (s.stream) (void)hipStreamSynchronize(s.stream); }
        if (c->zc) {                                             // tell the servers to leave, wait for them, release
            ZcServer &z = *c->zc;
            if (z.ctl) __atomic_store_n(&z.ctl->quit, 1u, __ATOMIC_RELEASE);
            if (getenv("MI_BLUR_DEBUG_SERVER"))
                fprintf(stderr, "[mi_blur] destroy: server state head %u launched %u servers_done %u tail %u; stream query %d\n", z.head, z.launched,
                        z.ctl ? z.ctl->servers_done : 0u, z.ctl ? z.ctl->tail : 0u, z.stream ? (int)hipStreamQuery(z.stream) : -1);
            if (z.stream) { (void)hipStreamSynchronize(z.stream); (void)hipStreamDestroy(z.stream); }
            if (getenv("MI_BLUR_DEBUG_SERVER")) fprintf(stderr, "[mi_blur] destroy: servers gone\n");
            if (z.dev) (void)hipFree(z.dev);
            if (z.trace) (void)hipFree(z.trace);
            if (z.ctl) (void)hipHostFree(z.ctl);
            delete c->zc;
            c->zc = nullptr;
        }
        for (auto &s : c->slots) free_slot(s);
        for (auto &t : c->ev_pool) { (void)hipEventDestroy(t.s); (void)hipEventDestroy(t.e); }
        if (c->zc_ref) (void)hipEventDestroy(c->zc_ref);
        if (c->fused_count) (void)hipFree(c->fused_count);
        if (c->fused_host) (void)hipHostFree(c->fused_host);
        if (c->fused_poll) { (void)hipStreamSynchronize(c->fused_poll); (void)hipStreamDestroy(c->fused_poll); }
        if (c->fused_watch) { (void)hipStreamSynchronize(c->fused_watch); (void)hipStreamDestroy(c->fused_watch); }
        if (c->fused_word) (void)hipHostFree(c->fused_word);
        if (c->pool_in) (void)hipFree(c->pool_in);
        if (c->pool_out) (void)hipFree(c->pool_out);
        (void)hipGetLastError();
    }
    delete c;
}

// Hand one zero-copy batch to the context's batch server.  MI_BLUR_ERR_UNSUPPORTED = "not this path" (another tile shape
// than the running server's): the caller launches the batch the classic way.
static int zc_server_submit(mi_blur_ctx *c, Slot &s, const LaunchDesc &d, const Tunables &tun)
{
    if (c->slots.size() > ZC_RING) return MI_BLUR_ERR_UNSUPPORTED;
    if (!c->zc) {
        ZcServer *z = new (std::nothrow) ZcServer;
        if (!z) return MI_BLUR_ERR_NOMEM;
        // The server stream is non-blocking, i.e. NOT ordered behind the NULL stream, and hipMemset of device memory may
        // return before the fill has run: the zero-fills go on the server stream itself, in front of the first server.
        hipError_t e = hipHostMalloc((void **)&z->ctl, sizeof(ZcHostCtl), hipHostMallocDefault);
        if (e == hipSuccess) { memset(z->ctl, 0, sizeof(ZcHostCtl)); e = hipHostGetDevicePointer((void **)&z->ctl_dev, z->ctl, 0); }
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&z->stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipMalloc((void **)&z->dev, sizeof(ZcDevCtl));
        if (e == hipSuccess) e = hipMemsetAsync(z->dev, 0, sizeof(ZcDevCtl), z->stream);
        if (e == hipSuccess && tun.zero_copy_trace) {
            const size_t tb = (size_t)ZC_TRACE_BATCHES * (size_t)tun.zero_copy_workers * 5u * sizeof(unsigned long long);
            e = hipMalloc((void **)&z->trace, tb);
            if (e == hipSuccess) e = hipMemsetAsync(z->trace, 0, tb, z->stream);
        }
        if (e == hipSuccess && tun.zero_copy_debug_base > 0) {
            // test hook: start just short of the 2^32 wrap of the batch and tile numbers.  Every word that holds a number is set
            // to what a server that had really served that many batches would have left behind.
            const unsigned B = 0u - (unsigned)tun.zero_copy_debug_base, T = 0u - 7u * (unsigned)tun.zero_copy_debug_base;
            ZcDevCtl init{};
            init.next[0] = B; init.gnext[0] = T; init.avail = B; init.ticket = T;
            e = hipMemcpyAsync(z->dev, &init, offsetof(ZcDevCtl, tiles_done), hipMemcpyHostToDevice, z->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(z->stream);      // `init` is on this stack
            z->ctl->tail = B;
            for (unsigned i = 0; i < ZC_RING; i++) z->ctl->done[i] = B - ((B - i - 1u) % ZC_RING);   // (last batch before B in slot i) + 1
            z->head = B; z->tile_base = T;
        }
        if (e == hipSuccess) e = hipStreamSynchronize(z->stream);          // once per context: the fills are done before anything is launched
        if (e != hipSuccess) {
            (void)hipGetLastError();
            if (z->stream) (void)hipStreamDestroy(z->stream);
            if (z->dev) (void)hipFree(z->dev);
            if (z->trace) (void)hipFree(z->trace);
            if (z->ctl) (void)hipHostFree(z->ctl);
            delete z;
            return MI_BLUR_ERR_HIP_BASE - (int)e;
        }
        z->fixed_share = tun.zero_copy_tickets ? 0 : 1;
        z->n_workers = (unsigned)tun.zero_copy_workers;
        z->budget = (unsigned)tun.zero_copy_budget;
        z->idle_ticks = (unsigned)tun.zero_copy_idle_us * 100u;      // 100 MHz device clock
        c->zc = z;
    }
    ZcServer &z = *c->zc;
    ZcBatch b{};
    unsigned n_tiles = 0;
    int rc = zc_fill_batch(d, &z.geo, &b, &n_tiles);
    if (rc) return rc;
    // A worker scans the descriptors from the batch of its last ticket to the batch of its next one, and the ticket counter
    // can be n_workers tiles beyond the newest published batch: with batches of very few tiles that span, plus the batches
    // the host may publish meanwhile, must stay inside the ring.  (Such batches gain nothing from the server anyway.)
    if (n_tiles == 0 || z.n_workers / n_tiles + 2u * (unsigned)c->slots.size() + 2u > ZC_RING) return MI_BLUR_ERR_UNSUPPORTED;
    b.tile_first = z.tile_base;                                          // tiles are numbered through the batches
    b.n_tiles = n_tiles;
    z.tile_base += n_tiles;
    const unsigned k = z.head;
    memcpy(&z.ctl->batch[k % ZC_RING], &b, sizeof b);
    __atomic_store_n(&z.ctl->tail, k + 1u, __ATOMIC_RELEASE);           // the frames and the descriptor are written: publish
    z.head = k + 1u;
    // Keep TWO servers that have not said they stopped: the running one may be deciding to leave (idle, budget) just as
    // this batch is published; the one queued behind it on the same stream then takes it.
    const unsigned stopped = __atomic_load_n(&z.ctl->servers_done, __ATOMIC_ACQUIRE);
    while (z.launched - stopped < 2u) {
        rc = zc_launch_server(z.geo, z.ctl_dev, z.dev, z.launched, z.n_workers, z.budget, z.idle_ticks, z.stream, z.trace, z.fixed_share);
        if (rc) return rc;
        z.launched++;
    }
    s.zc_server = true; s.zc_index = k;
    return MI_BLUR_OK;
}

// One batch through a slot: Write -> NDRange -> Read (heterogeneous_blur.c:520-533), but one
// launch (and one DMA each way) for the whole batch instead of one triple per image.
// in_stride/out_stride: bytes between consecutive images' first band row / first output row in
// HOST memory (a strided batch = Approach 2's "same rows of every image": a 2-D DMA gathers
// the bands into one contiguous device batch).
static int submit_common(mi_blur_ctx *c, const uint8_t *host_in, uint8_t *host_out, int band_rows, int n_images,
                         int y0, int y1, size_t in_stride, size_t out_stride)
{
    const size_t pitch = (size_t)c->W * c->C;
    const size_t band_in = pitch * band_rows, band_out = pitch * (size_t)(y1 - y0);
    const size_t in_bytes = band_in * n_images, out_bytes = band_out * n_images;
    if (in_stride == 0) in_stride = band_in;
    if (out_stride == 0) out_stride = band_out;
    if (c->is_cpu()) {
        CpuJob *j = new (std::nothrow) CpuJob;
        if (!j) return MI_BLUR_ERR_NOMEM;
        const int W = c->W, C = c->C, R = c->R, nt = c->n_threads;
        j->work = [=]() { cpu_blur_batch(host_in, host_out, W, band_rows, C, R, n_images, y0, y1, nt, in_stride, out_stride); };
        if (!c->cpu_worker) { c->cpu_worker = new (std::nothrow) CpuWorker; if (!c->cpu_worker) { delete j; return MI_BLUR_ERR_NOMEM; } }
        c->cpu_worker->push(j);
        c->cpu_jobs.push_back(j);
    } else {
        HIP_TRY(hipSetDevice(c->device));
        Slot &s = c->slots[c->next_slot];
        c->next_slot = (c->next_slot + 1) % (int)c->slots.size();
        int rc = finish_slot(c, s);
        if (rc) return rc;
        // Zero-copy: when both caller buffers are pinned the kernel reads and writes them in place over PCIe.  On this
        // platform one kernel moving both directions sustains ~70 GB/s (35 each way) while the copy engines give ~55 GB/s
        // one way at a time and collapse to ~28 GB/s total when H2D and D2H overlap (profiles/r01_zero_copy.txt): +33-40 %
        // images/s end to end.  Each zero-copy launch keeps only "zero_copy_blocks" workgroups resident (they loop over the
        // tiles) and consecutive launches alternate over up to "zero_copy_streams" of the context's streams, so that reads
        // of one tile and writes of another keep both directions of the link busy (profiles/r02_e2e.txt).
        const Tunables tun = tunables();    // ONE copy of the knobs for this submit
        if (tun.zero_copy) {
            const uint8_t *zin = pinned_device_ptr(host_in);
            uint8_t *zout = pinned_device_ptr(host_out);
            const bool dense = in_stride == band_in && out_stride == band_out;
            if (zin && zout && (dense || (tiled_eligible(zin, zout, c->W, c->C) && in_stride % 16 == 0 && out_stride % 16 == 0))) {
                s.out_staged = false; s.user_out = host_out; s.out_bytes = out_bytes; s.out_band = band_out; s.out_stride = out_stride;
                s.out_n = n_images;
                // consecutive zero-copy launches alternate over the first zn streams of the context
                const int zn = std::max(1, std::min(tun.zero_copy_streams, (int)c->slots.size()));
                const hipStream_t zs = c->slots[(c->zero_copy_launches % (uint64_t)zn)].stream;
                // one dispatch packet, nothing else: the kernel's own stop event doubles as the completion event (every
                // extra hipEventRecord is a barrier packet between two kernels)
                LaunchDesc d{};
                d.in = zin; d.out = zout; d.width = c->W; d.band_rows = band_rows; d.channels = c->C;
                d.radius = c->R; d.n_images = n_images; d.y0 = y0; d.y1 = y1; d.variant = MI_BLUR_VARIANT_AUTO;
                d.in_stride = (long long)in_stride; d.out_stride = (long long)out_stride;
                // small batches stay with one launch each: the server's hand-off (descriptor over the link, poller, completion
                // word, the host's wait) costs ~26 us per batch against ~8 us for a launch — below ~1.3 MB each way the launch
                // wins, by up to 3x for a single 256x256 frame (profiles/r03_e2e_shape_sweep.txt)
                if (tun.zero_copy_server && out_bytes >= (size_t)tun.zero_copy_server_min_kb * 1024u) {
                    rc = zc_server_submit(c, s, d, tun);
                    if (rc == MI_BLUR_OK) {
                        s.zero_copy = true; s.busy = true;
                        c->tm.bytes_h2d += in_bytes; c->tm.bytes_d2h += out_bytes;
                        c->tm.bytes_alg += 2ull * out_bytes;
                        c->tm.images += (uint64_t)n_images;
                        c->tm.launches += 1;
                        c->zero_copy_launches += 1;
                        return MI_BLUR_OK;
                    }
                    if (rc != MI_BLUR_ERR_UNSUPPORTED) return rc;
                    s.zc_server = false;
                }
                const bool with_events = tun.zero_copy_events != 0;
                d.stream = zs;
                if (with_events) { d.start = s.ks; d.stop = s.ke; }
                s.zc_plain = !with_events; s.zc_stream = zs;
                d.max_blocks = tun.zero_copy_blocks;
                // once per sync window, in front of its first launch (and again every ~10 s of a window that never syncs,
                // so the float milliseconds since the reference keep their resolution)
                if (with_events && (!c->zc_ref_valid || c->zc_covered_ms > 10e3)) {
                    if (!c->zc_ref) HIP_TRY(hipEventCreate(&c->zc_ref));
                    HIP_TRY(hipEventRecord(c->zc_ref, zs));
                    c->zc_ref_valid = true; c->zc_covered_ms = 0.0;
                }
                rc = launch(d);
                if (rc) return rc;
                s.zero_copy = true;                            // only now: a failed launch leaves the slot idle and staged
                s.busy = true;
                c->tm.bytes_h2d += in_bytes; c->tm.bytes_d2h += out_bytes;
                c->tm.bytes_alg += 2ull * out_bytes;
                c->tm.images += (uint64_t)n_images;
                c->tm.launches += 1;
                c->zero_copy_launches += 1;
                return MI_BLUR_OK;
            }
        }
        const bool in_pinned = is_pinned(host_in);
        s.zero_copy = false;
        s.out_staged = !is_pinned(host_out);
        s.user_out = host_out; s.out_bytes = out_bytes; s.out_band = band_out; s.out_stride = out_stride; s.out_n = n_images;
        const uint8_t *src = host_in;
        size_t src_stride = in_stride;
        rc = slot_staging(c, s, !in_pinned, s.out_staged);
        if (rc) return rc;
        if (!in_pinned) {                       // pageable caller memory: gather into the slot's pinned staging
            copy_blocks(s.h_in, band_in, host_in, in_stride, band_in, n_images, STAGING_COPY_THREADS);
            src = s.h_in; src_stride = band_in;
        }
        // Pageable caller memory (the reference's malloc'd batch buffers, kept as they are): the staging buffers ARE pinned, so
        // the batch server takes the batch from them in place — staging in -> blur -> staging out as one kernel stream over the
        // link — instead of a DMA copy each way around a launch (copies in both directions at once collapse to ~28 GB/s in
        // total on this platform, profiles/r01_pcie_probe.txt).  What stays is the host's own gather / scatter between the
        // caller's memory and the staging.
        // (any batch size: for small batches too the server on the staging beats two DMA copies around a launch — a single
        // 256x256 frame per submit: 78 k against 36 k img/s)
        if (tun.zero_copy && tun.zero_copy_server && tun.staged_server) {
            const uint8_t *zin = pinned_device_ptr(src);
            uint8_t *zout = pinned_device_ptr(s.out_staged ? s.h_out : host_out);
            const size_t zout_stride = s.out_staged ? band_out : out_stride;
            if (zin && zout) {
                LaunchDesc d{};
                d.in = zin; d.out = zout; d.width = c->W; d.band_rows = band_rows; d.channels = c->C;
                d.radius = c->R; d.n_images = n_images; d.y0 = y0; d.y1 = y1; d.variant = MI_BLUR_VARIANT_AUTO;
                d.in_stride = (long long)src_stride; d.out_stride = (long long)zout_stride;
                rc = zc_server_submit(c, s, d, tun);
                if (rc == MI_BLUR_OK) {
                    s.zero_copy = true; s.busy = true;
                    c->tm.bytes_h2d += in_bytes; c->tm.bytes_d2h += out_bytes;
                    c->tm.bytes_alg += 2ull * out_bytes;
                    c->tm.images += (uint64_t)n_images;
                    c->tm.launches += 1;
                    c->zero_copy_launches += 1;
                    return MI_BLUR_OK;
                }
                if (rc != MI_BLUR_ERR_UNSUPPORTED) return rc;
                s.zc_server = false;
            }
        }
        rc = slot_device(c, s);
        if (rc) return rc;
        HIP_TRY(hipEventRecord(s.ev[0], s.stream));
        if (src_stride == band_in)
            HIP_TRY(hipMemcpyAsync(s.d_in, src, in_bytes, hipMemcpyHostToDevice, s.stream));
        else
            HIP_TRY(hipMemcpy2DAsync(s.d_in, band_in, src, src_stride, band_in, (size_t)n_images, hipMemcpyHostToDevice, s.stream));
        HIP_TRY(hipEventRecord(s.ev[1], s.stream));
        LaunchDesc d{};
        d.in = s.d_in; d.out = s.d_out; d.width = c->W; d.band_rows = band_rows; d.channels = c->C;
        d.radius = c->R; d.n_images = n_images; d.y0 = y0; d.y1 = y1; d.variant = MI_BLUR_VARIANT_AUTO;
        d.stream = s.stream; d.start = s.ks; d.stop = s.ke;
        d.concurrent = (int)c->slots.size();
        rc = launch(d);
        if (rc) return rc;
        HIP_TRY(hipEventRecord(s.ev[2], s.stream));
        if (s.out_staged || out_stride == band_out)
            HIP_TRY(hipMemcpyAsync(s.out_staged ? s.h_out : host_out, s.d_out, out_bytes, hipMemcpyDeviceToHost, s.stream));
        else
            HIP_TRY(hipMemcpy2DAsync(host_out, out_stride, s.d_out, band_out, band_out, (size_t)n_images, hipMemcpyDeviceToHost, s.stream));
        HIP_TRY(hipEventRecord(s.ev[3], s.stream));
        s.busy = true;
        c->tm.bytes_h2d += in_bytes; c->tm.bytes_d2h += out_bytes;
    }
    c->tm.bytes_alg += 2ull * out_bytes;
    c->tm.images += (uint64_t)n_images;
    c->tm.launches += 1;
    return MI_BLUR_OK;
}

extern "C" int mi_blur_submit(mi_blur_ctx *c, const uint8_t *host_in, uint8_t *host_out, int n_images)
{
    if (!c || !host_in || !host_out || host_in == host_out) return MI_BLUR_ERR_INVALID;
    if (n_images < 0 || n_images > c->max_batch) return MI_BLUR_ERR_INVALID;
    if (n_images == 0) return MI_BLUR_OK;
    return submit_common(c, host_in, host_out, c->H, n_images, 0, c->H, 0, 0);
}

extern "C" int mi_blur_submit_band(mi_blur_ctx *c, const uint8_t *host_in, uint8_t *host_out, int band_rows,
                                   int halo_top, int halo_bottom)
{
    if (!c || !host_in || !host_out || host_in == host_out) return MI_BLUR_ERR_INVALID;
    if (band_rows <= 0 || band_rows > c->H || halo_top < 0 || halo_bottom < 0) return MI_BLUR_ERR_INVALID;
    if (halo_top + halo_bottom >= band_rows) return MI_BLUR_ERR_INVALID;
    return submit_common(c, host_in, host_out, band_rows, 1, halo_top, band_rows - halo_bottom, 0, 0);
}

extern "C" int mi_blur_submit_bands(mi_blur_ctx *c, const uint8_t *host_in, uint8_t *host_out, int n_images,
                                    size_t host_image_stride, int band_rows, int halo_top, int halo_bottom)
{
    if (!c || !host_in || !host_out || host_in == host_out) return MI_BLUR_ERR_INVALID;
    if (n_images < 0 || n_images > c->max_batch) return MI_BLUR_ERR_INVALID;
    if (band_rows <= 0 || band_rows > c->H || halo_top < 0 || halo_bottom < 0) return MI_BLUR_ERR_INVALID;
    if (halo_top + halo_bottom >= band_rows) return MI_BLUR_ERR_INVALID;
    if (host_image_stride < (size_t)c->W * c->C * band_rows) return MI_BLUR_ERR_INVALID;
    if (n_images == 0) return MI_BLUR_OK;
    return submit_common(c, host_in, host_out, band_rows, n_images, halo_top, band_rows - halo_bottom,
                         host_image_stride, host_image_stride);
}

// Frames as the reference's loader hands them over: PLANAR (CImg storage).  The reference interleaves every frame on one
// host core before the stream is built (heterogeneous_blur.c:125-134) and de-interleaves on the way out to save one
// (split_image_blur.c:40-56); here both repacks are GPU kernels inside the submit.  The repack-in kernel reads the
// pinned planar frames straight over PCIe (it IS the upload: its duration is the h2d bucket) and writes the
// interleaved batch into HBM; the blur runs HBM -> HBM; the way out is either a D2H copy of the interleaved batch or the
// repack-out kernel writing planar frames straight into pinned host memory.  The slot's two device buffers are enough:
// d_in is free again once the blur has read it, so the planar result is built there.
extern "C" int mi_blur_submit_planar(mi_blur_ctx *c, const uint8_t *host_planar_in, uint8_t *host_out, int n_images, int planar_out)
{
    if (!c || !host_planar_in || !host_out || host_planar_in == host_out) return MI_BLUR_ERR_INVALID;
    if (n_images < 0 || n_images > c->max_batch) return MI_BLUR_ERR_INVALID;
    if (n_images == 0) return MI_BLUR_OK;
    const size_t bytes = c->image_bytes * (size_t)n_images;
    if (c->is_cpu()) {
        CpuJob *j = new (std::nothrow) CpuJob;
        if (!j) return MI_BLUR_ERR_NOMEM;
        const int W = c->W, H = c->H, C = c->C, R = c->R, nt = c->n_threads;
        j->work = [=]() {
            std::vector<uint8_t> a(bytes), b(planar_out ? bytes : 0);
            cpu_repack(host_planar_in, a.data(), W, H, C, n_images, true, nt);
            cpu_blur_batch(a.data(), planar_out ? b.data() : host_out, W, H, C, R, n_images, 0, H, nt);
            if (planar_out) cpu_repack(b.data(), host_out, W, H, C, n_images, false, nt);
        };
        if (!c->cpu_worker) { c->cpu_worker = new (std::nothrow) CpuWorker; if (!c->cpu_worker) { delete j; return MI_BLUR_ERR_NOMEM; } }
        c->cpu_worker->push(j);
        c->cpu_jobs.push_back(j);
    } else {
        HIP_TRY(hipSetDevice(c->device));
        Slot &s = c->slots[c->next_slot];
        c->next_slot = (c->next_slot + 1) % (int)c->slots.size();
        int rc = finish_slot(c, s);
        if (rc) return rc;
        s.zero_copy = false;
        // source the repack-in kernel can read: the caller's frames if they are pinned, the slot's pinned staging otherwise
        const uint8_t *src = pinned_device_ptr(host_planar_in);
        rc = slot_device(c, s);
        if (!rc) rc = slot_staging(c, s, !src, !pinned_device_ptr(host_out));
        if (rc) return rc;
        if (!src) {
            copy_blocks(s.h_in, bytes, host_planar_in, bytes, bytes, 1, STAGING_COPY_THREADS);
            src = pinned_device_ptr(s.h_in);
            if (!src) return MI_BLUR_ERR_STATE;
        }
        uint8_t *dst_pinned = pinned_device_ptr(host_out);
        s.out_staged = !dst_pinned;
        s.user_out = host_out; s.out_bytes = bytes; s.out_band = bytes; s.out_stride = bytes; s.out_n = 1;
        HIP_TRY(hipEventRecord(s.ev[0], s.stream));
        rc = launch_planar_to_interleaved(src, s.d_in, c->W, c->H, c->C, n_images, s.stream);
        if (rc) return rc;
        HIP_TRY(hipEventRecord(s.ev[1], s.stream));
        LaunchDesc d{};
        d.in = s.d_in; d.out = s.d_out; d.width = c->W; d.band_rows = c->H; d.channels = c->C;
        d.radius = c->R; d.n_images = n_images; d.y0 = 0; d.y1 = c->H; d.variant = MI_BLUR_VARIANT_AUTO;
        d.stream = s.stream; d.start = s.ks; d.stop = s.ke;
        d.concurrent = (int)c->slots.size();
        rc = launch(d);
        if (rc) return rc;
        HIP_TRY(hipEventRecord(s.ev[2], s.stream));
        if (planar_out) {
            uint8_t *dst = dst_pinned ? dst_pinned : pinned_device_ptr(s.h_out);
            if (!dst) return MI_BLUR_ERR_STATE;
            rc = launch_interleaved_to_planar(s.d_out, dst, c->W, c->H, c->C, n_images, s.stream);
            if (rc) return rc;
        } else {
            HIP_TRY(hipMemcpyAsync(s.out_staged ? s.h_out : host_out, s.d_out, bytes, hipMemcpyDeviceToHost, s.stream));
        }
        HIP_TRY(hipEventRecord(s.ev[3], s.stream));
        s.busy = true;
        c->tm.bytes_h2d += bytes; c->tm.bytes_d2h += bytes;
    }
    c->tm.bytes_alg += 2ull * bytes;
    c->tm.images += (uint64_t)n_images;
    c->tm.launches += 1;
    return MI_BLUR_OK;
}

// Wait for the OLDEST submit still in flight (its output is then in caller memory), so a host
// that rotates n_slots batch buffers can refill the oldest one while the newer ones run.
extern "C" int mi_blur_wait_oldest(mi_blur_ctx *c)
{
    if (!c) return MI_BLUR_ERR_INVALID;
    if (c->is_cpu()) {
        if (c->cpu_jobs.empty()) return MI_BLUR_OK;
        CpuJob *j = c->cpu_jobs.front();
        c->cpu_worker->wait(j);
        c->tm.kernel_ms += j->ms;
        delete j;
        c->cpu_jobs.erase(c->cpu_jobs.begin());
        return MI_BLUR_OK;
    }
    HIP_TRY(hipSetDevice(c->device));
    const int n = (int)c->slots.size();
    for (int i = 0; i < n; i++) {
        Slot &s = c->slots[(c->next_slot + i) % n];
        if (s.busy) return finish_slot(c, s);
    }
    return MI_BLUR_OK;
}

// ----------------------------------------------------------------------------------
// device-resident stream
// ----------------------------------------------------------------------------------
// Where a pool lands in HBM moves the big launches between two levels ~6 % apart (1080p 5x5: 138 <-> 148 us; the 3x3
// stream: 322 <-> 345 us): same request counts on every channel, but 1.5x the read DRAM-credit stalls on the slow
// placements (profiles/r03_placement_channels.txt).  Nothing in the address says which it will be, and hipMalloc offers no
// handle on it — but a placement keeps its level for as long as it lives, so the pool is CHOSEN: "resident_place_trials"
// candidate inputs and as many candidate outputs are allocated side by side, every (input, output) pair is timed on the
// launch the pool is for (the context's kernel over the whole pool, after a clock ramp), the fastest pair is kept and the
// other buffers are freed.
static int place_pool(mi_blur_ctx *c, size_t bytes, int pool_images, int trials)
{
    // `trials` input and `trials` output candidates; EVERY (input, output) pair is timed — the level belongs to the pair, and
    // n + n buffers give n x n pairs to choose from for the price of n.
    std::vector<uint8_t *> in((size_t)trials, nullptr), out((size_t)trials, nullptr);
    int n_ok = 0;
    for (int i = 0; i < trials; i++) {
        if (hipMalloc((void **)&in[i], bytes) != hipSuccess || hipMalloc((void **)&out[i], bytes) != hipSuccess) {
            (void)hipGetLastError();
            if (in[i]) { (void)hipFree(in[i]); in[i] = nullptr; }
            break;                                              // memory is short: choose among what there is
        }
        n_ok++;
    }
    if (n_ok == 0) return MI_BLUR_ERR_HIP_BASE - (int)hipErrorOutOfMemory;
    c->place_ms.clear();
    int keep_in = 0, keep_out = 0;
    if (n_ok > 1) {
        hipStream_t st = c->slots[0].stream;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        std::vector<float> best((size_t)n_ok * n_ok, 1e30f);
        if (hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess) {
            auto pass = [&](int i, int j) {
                LaunchDesc d{};
                d.in = in[i]; d.out = out[j]; d.width = c->W; d.band_rows = c->H; d.channels = c->C; d.radius = c->R;
                d.n_images = pool_images; d.y0 = 0; d.y1 = c->H; d.variant = MI_BLUR_VARIANT_AUTO; d.stream = st;
                return launch(d);
            };
            int rc = MI_BLUR_OK;
            const auto t0 = std::chrono::steady_clock::now();      // ~50 ms of launches first: the clocks ramp (r02_clock_ramp.txt)
            while (rc == MI_BLUR_OK && std::chrono::steady_clock::now() - t0 < std::chrono::milliseconds(50)) {
                for (int r = 0; r < 4 && rc == MI_BLUR_OK; r++) rc = pass(0, 0);
                if (hipStreamSynchronize(st) != hipSuccess) rc = MI_BLUR_ERR_STATE;
            }
            for (int round = 0; round < 2 && rc == MI_BLUR_OK; round++)
                for (int i = 0; i < n_ok && rc == MI_BLUR_OK; i++)
                    for (int j = 0; j < n_ok && rc == MI_BLUR_OK; j++) {
                        rc = pass(i, j);                            // one untimed launch: this pair's lines and TLB entries
                        (void)hipEventRecord(e0, st);
                        for (int r = 0; r < 2 && rc == MI_BLUR_OK; r++) rc = pass(i, j);
                        (void)hipEventRecord(e1, st);
                        float ms = 0.f;
                        if (hipEventSynchronize(e1) == hipSuccess && hipEventElapsedTime(&ms, e0, e1) == hipSuccess && ms > 0.f)
                            best[(size_t)i * n_ok + j] = std::min(best[(size_t)i * n_ok + j], ms / 2.0f);
                    }
            (void)hipGetLastError();
            for (int i = 0; i < n_ok; i++)
                for (int j = 0; j < n_ok; j++) {
                    c->place_ms.push_back(best[(size_t)i * n_ok + j]);
                    if (best[(size_t)i * n_ok + j] < best[(size_t)keep_in * n_ok + keep_out]) { keep_in = i; keep_out = j; }
                }
        }
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
    }
    c->pool_in = in[keep_in]; c->pool_out = out[keep_out];
    c->place_kept = keep_in * n_ok + keep_out;
    for (int i = 0; i < trials; i++) {
        if (i != keep_in && in[i]) (void)hipFree(in[i]);
        if (i != keep_out && out[i]) (void)hipFree(out[i]);
    }
    return MI_BLUR_OK;
}

extern "C" int mi_blur_resident_alloc(mi_blur_ctx *c, int pool_images)
{
    if (!c || pool_images <= 0) return MI_BLUR_ERR_INVALID;
    if (c->is_cpu()) return MI_BLUR_ERR_STATE;
    HIP_TRY(hipSetDevice(c->device));
    if (c->pool_in) { (void)hipFree(c->pool_in); c->pool_in = nullptr; }
    if (c->pool_out) { (void)hipFree(c->pool_out); c->pool_out = nullptr; }
    c->pool_images = 0; c->cursor = 0;
    const size_t bytes = c->image_bytes * (size_t)pool_images;
    // placement only matters to launches that stream hundreds of MB; candidates are only tried while they are cheap to hold
    const int trials = tunables().resident_place_trials;
    if (trials > 1 && bytes >= ((size_t)128 << 20) && bytes <= ((size_t)8 << 30)) {
        int rc = place_pool(c, bytes, pool_images, std::min(trials, 8));
        if (rc) return rc;
    } else {
        c->place_ms.clear(); c->place_kept = 0;
        HIP_TRY(hipMalloc((void **)&c->pool_in, bytes));
        HIP_TRY(hipMalloc((void **)&c->pool_out, bytes));
    }
    c->pool_images = pool_images;
    return MI_BLUR_OK;
}

// What mi_blur_resident_alloc measured when it chose the pool: the per-launch time (ms) of each candidate placement, and
// which one it kept.  n = 0: no trial was run (small pool, "resident_place_trials" <= 1).
extern "C" int mi_blur_resident_placement(mi_blur_ctx *c, float *ms, int max_n, int *kept)
{
    if (!c) return MI_BLUR_ERR_INVALID;
    const int n = (int)std::min<size_t>(c->place_ms.size(), (size_t)std::max(max_n, 0));
    for (int i = 0; i < n && ms; i++) ms[i] = c->place_ms[i];
    if (kept) *kept = c->place_kept;
    return n;
}

extern "C" int mi_blur_resident_upload(mi_blur_ctx *c, int pool_index, const uint8_t *host_in, int n_images)
{
    if (!c || !host_in || !c->pool_in) return MI_BLUR_ERR_STATE;
    if (pool_index < 0 || n_images < 0 || pool_index + n_images > c->pool_images) return MI_BLUR_ERR_INVALID;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemcpy(c->pool_in + (size_t)pool_index * c->image_bytes, host_in, c->image_bytes * (size_t)n_images,
                      hipMemcpyHostToDevice));
    return MI_BLUR_OK;
}

extern "C" int mi_blur_resident_download(mi_blur_ctx *c, int pool_index, uint8_t *host_out, int n_images)
{
    if (!c || !host_out || !c->pool_out) return MI_BLUR_ERR_STATE;
    if (pool_index < 0 || n_images < 0 || pool_index + n_images > c->pool_images) return MI_BLUR_ERR_INVALID;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(host_out, c->pool_out + (size_t)pool_index * c->image_bytes, c->image_bytes * (size_t)n_images,
                      hipMemcpyDeviceToHost));
    return MI_BLUR_OK;
}

extern "C" int mi_blur_resident_fill_synthetic(mi_blur_ctx *c, int first_index)
{
    if (!c || !c->pool_in) return MI_BLUR_ERR_STATE;
    const int chunk = std::max(1, (int)std::min<size_t>((size_t)c->pool_images, ((size_t)256 << 20) / c->image_bytes));
    std::vector<uint8_t> host;
    try { host.resize(c->image_bytes * (size_t)chunk); } catch (...) { return MI_BLUR_ERR_NOMEM; }
    for (int i = 0; i < c->pool_images; i += chunk) {
        const int n = std::min(chunk, c->pool_images - i);
        fill_synthetic(host.data(), c->W, c->H, c->C, first_index + i, n, c->n_threads);
        int rc = mi_blur_resident_upload(c, i, host.data(), n);
        if (rc) return rc;
    }
    return MI_BLUR_OK;
}

extern "C" void *mi_blur_resident_in(mi_blur_ctx *c) { return c ? c->pool_in : nullptr; }
extern "C" void *mi_blur_resident_out(mi_blur_ctx *c) { return c ? c->pool_out : nullptr; }

// One pass of the stream over the resident pool: the batch loop of
// heterogeneous_blur.c:418-427 with the transfers gone (data already in HBM).
extern "C" int mi_blur_resident_run(mi_blur_ctx *c, int n_images, int batch, int timed_every)
{
    if (!c || n_images < 0 || batch <= 0) return MI_BLUR_ERR_INVALID;
    if (c->is_cpu() || !c->pool_in) return MI_BLUR_ERR_STATE;
    if (batch > c->pool_images) return MI_BLUR_ERR_INVALID;
    HIP_TRY(hipSetDevice(c->device));
    int launch_idx = 0;
    for (int done = 0; done < n_images; done += batch, launch_idx++) {
        const int b = std::min(batch, n_images - done);
        const bool timed = timed_every > 0 && launch_idx % timed_every == 0;
        if (c->cursor + b > c->pool_images) c->cursor = 0;
        Slot &s = c->slots[c->rr];
        c->rr = (c->rr + 1) % (int)c->slots.size();
        LaunchDesc d{};
        d.in = c->pool_in + (size_t)c->cursor * c->image_bytes;
        d.out = c->pool_out + (size_t)c->cursor * c->image_bytes;
        d.width = c->W; d.band_rows = c->H; d.channels = c->C; d.radius = c->R; d.n_images = b;
        d.y0 = 0; d.y1 = c->H; d.variant = MI_BLUR_VARIANT_AUTO; d.stream = s.stream;
        d.concurrent = (int)c->slots.size();
        if (timed) {
            if (c->ev_used == c->ev_pool.size()) {
                TimedLaunch t{};
                HIP_TRY(hipEventCreate(&t.s));
                HIP_TRY(hipEventCreate(&t.e));
                c->ev_pool.push_back(t);
            }
            d.start = c->ev_pool[c->ev_used].s; d.stop = c->ev_pool[c->ev_used].e;
            c->ev_used++;
        }
        int rc = launch(d);
        if (rc) return rc;
        c->cursor += b;
        c->tm.launches += 1;
        if (timed) { c->timed_launches += 1; c->timed_bytes_alg += 2ull * c->image_bytes * (uint64_t)b; }
    }
    c->tm.images += (uint64_t)n_images;
    c->tm.bytes_alg += 2ull * c->image_bytes * (uint64_t)n_images;
    return MI_BLUR_OK;
}

// The same pass as ONE dispatch (blur_fused_kernel): the GPU walks the batches in order and raises a host-visible
// flag per finished batch, so the batch stays the unit of completion without being the unit of dispatch.
// Words of the per-batch completion counters for a pool of `cap` images: 8 per batch at the smallest batch (one image), plus room
// for a few batches of thousands of tiles to spread over up to 256 words each (two 8192x8192 frames are two batches of 3200 tiles)
static size_t fused_words(int cap) { return 8 * (size_t)cap + 16384; }

extern "C" int mi_blur_resident_run_fused(mi_blur_ctx *c, int n_images, int batch, int timed)
{
    if (!c || n_images <= 0 || batch <= 0) return MI_BLUR_ERR_INVALID;
    if (c->is_cpu() || !c->pool_in) return MI_BLUR_ERR_STATE;
    if (n_images > c->pool_images) return MI_BLUR_ERR_INVALID;           // one contiguous run of the pool
    HIP_TRY(hipSetDevice(c->device));
    const int nb = (n_images + batch - 1) / batch;
    if (nb > c->fused_cap) {
        // the previous pass may still be writing its flags
        for (auto &s : c->slots) HIP_TRY(hipStreamSynchronize(s.stream));
        if (c->fused_count) { (void)hipFree(c->fused_count); c->fused_count = nullptr; }
        if (c->fused_host) { (void)hipHostFree(c->fused_host); c->fused_host = nullptr; }
        c->fused_cap = 0;
        const int cap = std::max(nb, c->pool_images);      // enough for any batch size on this pool: never reallocated
        // (+16 words: the ticket counter of a pass's dynamic tail lives behind the batch counters; it is zero between passes)
        HIP_TRY(hipMalloc((void **)&c->fused_count, sizeof(unsigned) * (fused_words(cap) + 16)));
        HIP_TRY(hipMemset(c->fused_count + fused_words(cap), 0, sizeof(unsigned) * 16));
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipHostMalloc((void **)&c->fused_host, sizeof(unsigned) * fused_words(cap), hipHostMallocDefault));
        c->fused_cap = cap;
        c->fused_passes = 0;
    }
    if (!c->fused_poll) {
        HIP_TRY(hipStreamCreateWithFlags(&c->fused_poll, hipStreamNonBlocking));
        // the first device-to-host copy of this size class sets up the copy engine's queue (~8 ms): pay it here, not
        // in the first poll
        HIP_TRY(hipMemcpyAsync(c->fused_host, c->fused_count, sizeof(unsigned) * fused_words(c->fused_cap), hipMemcpyDeviceToHost, c->fused_poll));
        HIP_TRY(hipStreamSynchronize(c->fused_poll));
    }
    if (c->cursor + n_images > c->pool_images) c->cursor = 0;
    Slot &s = c->slots[0];
    LaunchDesc d{};
    d.in = c->pool_in + (size_t)c->cursor * c->image_bytes;
    d.out = c->pool_out + (size_t)c->cursor * c->image_bytes;
    d.width = c->W; d.band_rows = c->H; d.channels = c->C; d.radius = c->R; d.n_images = n_images;
    d.y0 = 0; d.y1 = c->H; d.variant = MI_BLUR_VARIANT_TILED; d.stream = s.stream;
    // The launch geometry (blocks per batch) depends on the tuning knobs as well as on the shape: take ONE copy of the
    // knobs, ask for the geometry first, launch with the same copy.
    const Tunables tun = tunables();
    unsigned tpb = 0, wpb = 0, blocks = 0, kcnt = 8;
    FusedDesc f{c->fused_count, batch, &tpb, &wpb, &blocks, &tun, true, c->fused_count + fused_words(c->fused_cap), (long long)fused_words(c->fused_cap), &kcnt};
    int rc = launch_fused(d, f);
    if (rc) return rc;
    // Repeated passes of the same shape AND geometry do not zero the counters (that would be one more dispatch per
    // pass): every pass adds the same amounts, so batch b of pass p is complete when its counters sum to p x (its blocks).
    if (n_images == c->fused_n && batch == c->fused_batch && tpb == c->fused_tpb && wpb == c->fused_wpb && kcnt == c->fused_k &&
        blocks == c->fused_blocks && c->fused_passes > 0 && c->fused_passes < (1u << 20)) {
        c->fused_passes += 1;
    } else {
        // the counters are about to be zeroed: a watcher of the previous pass must have seen that pass's last counts first
        // (it ends with its pass; zeros under it would leave it waiting for its hard limit)
        if (c->fused_watched && c->fused_watch) HIP_TRY(hipStreamSynchronize(c->fused_watch));
        HIP_TRY(hipMemsetAsync(c->fused_count, 0, sizeof(unsigned) * (size_t)kcnt * (size_t)nb, s.stream));
        c->fused_n = n_images; c->fused_batch = batch; c->fused_passes = 1;
        c->fused_tpb = tpb; c->fused_wpb = wpb; c->fused_blocks = blocks; c->fused_k = kcnt;
    }
    c->fused_batches = nb;
    c->fused_watched = false;
    if (timed & 2) {
        // a watcher wave keeps (pass, leading batches complete) in pinned host memory for this pass.  Counters that were just
        // zeroed must BE zero before it looks at them: it runs on its own stream, so wait for the fill.
        if (c->fused_passes == 1) HIP_TRY(hipStreamSynchronize(s.stream));
        if (!c->fused_watch) HIP_TRY(hipStreamCreateWithFlags(&c->fused_watch, hipStreamNonBlocking));
        if (!c->fused_word) {
            HIP_TRY(hipHostMalloc((void **)&c->fused_word, sizeof(unsigned long long), hipHostMallocDefault));
            *c->fused_word = 0;
            HIP_TRY(hipHostGetDevicePointer((void **)&c->fused_word_dev, c->fused_word, 0));
        }
        c->fused_watch_seq += 1;
        rc = launch_fused_watch(c->fused_count, (unsigned)nb, tpb, blocks, wpb * c->fused_passes, c->fused_word_dev, c->fused_watch_seq, c->fused_watch, c->fused_k);
        if (rc) return rc;
        c->fused_watched = true;
    }
    if (timed & 1) {
        if (c->ev_used == c->ev_pool.size()) {
            TimedLaunch t{};
            HIP_TRY(hipEventCreate(&t.s));
            HIP_TRY(hipEventCreate(&t.e));
            c->ev_pool.push_back(t);
        }
        d.start = c->ev_pool[c->ev_used].s; d.stop = c->ev_pool[c->ev_used].e;
        c->ev_used++;
    }
    f.geometry_only = false;
    rc = launch_fused(d, f);
    if (rc) { c->fused_passes = 0; c->fused_batches = 0; return rc; }     // nothing ran: zero the counters next time
    c->cursor += n_images;
    c->tm.launches += 1;
    c->tm.images += (uint64_t)n_images;
    c->tm.bytes_alg += 2ull * c->image_bytes * (uint64_t)n_images;
    if (timed & 1) { c->timed_launches += 1; c->timed_bytes_alg += 2ull * c->image_bytes * (uint64_t)n_images; }
    return MI_BLUR_OK;
}

// Non-blocking: how many LEADING batches of the latest fused pass are complete (their outputs are in the pool); >= 0.
// A NEGATIVE value is a mi_blur_status: the counters could not be read (a failed copy, a faulted device), which a
// polling caller must treat as the end of the poll, not as "none done yet".
extern "C" int mi_blur_resident_batches_done(mi_blur_ctx *c)
{
    if (!c) return MI_BLUR_ERR_INVALID;
    if (c->is_cpu()) return MI_BLUR_ERR_STATE;
    if (!c->fused_count || !c->fused_batches) return 0;       // no fused pass issued (or the last one failed to launch)
    if (c->fused_watched) {                                   // a watcher wave keeps the answer in host memory: no copy, no HIP call
        const unsigned long long w = __atomic_load_n(c->fused_word, __ATOMIC_ACQUIRE);
        return (unsigned)(w >> 32) == c->fused_watch_seq ? (int)(unsigned)(w & 0xffffffffu) : 0;
    }
    // read the counters on a stream of their own (the pass may still be running on the compute stream)
    HIP_TRY(hipSetDevice(c->device));
    const unsigned K = c->fused_k;
    HIP_TRY(hipMemcpyAsync(c->fused_host, c->fused_count, sizeof(unsigned) * (size_t)K * (size_t)c->fused_batches, hipMemcpyDeviceToHost, c->fused_poll));
    HIP_TRY(hipStreamSynchronize(c->fused_poll));
    int n = 0;
    for (; n < c->fused_batches; n++) {
        const unsigned first = (unsigned)n * c->fused_tpb;
        const unsigned blocks = std::min(c->fused_tpb, c->fused_blocks - first);
        unsigned sum = 0;
        for (unsigned k = 0; k < K; k++) sum += c->fused_host[(size_t)K * n + k];
        if (sum != blocks * c->fused_wpb * c->fused_passes) break;
    }
    return n;
}

// Read outputs of batches that mi_blur_resident_batches_done has reported, WITHOUT waiting for the dispatch that is
// still producing the later ones: a copy on the poll stream, which is ordered behind nothing on the compute streams.
extern "C" int mi_blur_resident_peek(mi_blur_ctx *c, int pool_index, uint8_t *host_out, int n_images)
{
    if (!c || !host_out || !c->pool_out || !c->fused_poll) return MI_BLUR_ERR_STATE;
    if (pool_index < 0 || n_images < 0 || pool_index + n_images > c->pool_images) return MI_BLUR_ERR_INVALID;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemcpyAsync(host_out, c->pool_out + (size_t)pool_index * c->image_bytes, c->image_bytes * (size_t)n_images,
                           hipMemcpyDeviceToHost, c->fused_poll));
    HIP_TRY(hipStreamSynchronize(c->fused_poll));
    return MI_BLUR_OK;
}

// ----------------------------------------------------------------------------------
// CPU device kernel + helpers
// ----------------------------------------------------------------------------------
extern "C" int mi_blur_cpu_run(const uint8_t *in, uint8_t *out, int width, int height, int channels, int radius,
                               int n_images, int n_threads)
{
    if (!in || !out || in == out || width <= 0 || height <= 0 || channels <= 0 || n_images < 0)
        return MI_BLUR_ERR_INVALID;
    if (radius != 1 && radius != 2) return MI_BLUR_ERR_INVALID;
    cpu_blur_batch(in, out, width, height, channels, radius, n_images, 0, height, n_threads);
    return MI_BLUR_OK;
}

extern "C" void mi_blur_fill_synthetic(uint8_t *host, int width, int height, int channels, int first_index,
                                       int n_images, int n_threads)
{
    if (host && width > 0 && height > 0 && channels > 0)
        fill_synthetic(host, width, height, channels, first_index, n_images, n_threads);
}

// Diagnostics (see mi_blur.h): fold the per-workgroup times of the tiled kernel per XCD, and optionally re-arm the slots.
extern "C" int mi_blur_debug_xcd_times(uint64_t end_ticks[8], uint64_t begin_ticks[8], int rearm)
{
    unsigned long long *d = debug_xcd_buffer();
    if (!d) return MI_BLUR_ERR_NOMEM;
    const size_t n = debug_xcd_slots();
    std::vector<unsigned long long> h;
    try { h.resize(2 * n); } catch (...) { return MI_BLUR_ERR_NOMEM; }
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(h.data(), d, 16 * n, hipMemcpyDeviceToHost));
    uint64_t e[8] = {0, 0, 0, 0, 0, 0, 0, 0}, b[8];
    for (int i = 0; i < 8; i++) b[i] = ~0ull;
    for (size_t i = 0; i < n; i++) {
        if (!h[2 * i]) continue;                                  // workgroup slot not written since the last re-arm
        const int x = (int)(h[2 * i] & 7u);
        e[x] = std::max<uint64_t>(e[x], h[2 * i] >> 4);
        b[x] = std::min<uint64_t>(b[x], h[2 * i + 1]);
    }
    for (int i = 0; i < 8; i++) { if (end_ticks) end_ticks[i] = e[i]; if (begin_ticks) begin_ticks[i] = b[i]; }
    if (rearm) HIP_TRY(hipMemset(d, 0, 16 * n));
    return MI_BLUR_OK;
}

// Diagnostics (see mi_blur.h): the batch server's per-worker phase stamps of the most recent batches.
extern "C" int mi_blur_debug_zc_trace(mi_blur_ctx *c, uint64_t *out, int max_batches, int *n_workers, unsigned *batches_published)
{
    if (!c || !out || max_batches <= 0) return MI_BLUR_ERR_INVALID;
    if (c->is_cpu() || !c->zc || !c->zc->trace) return MI_BLUR_ERR_STATE;
    ZcServer &z = *c->zc;
    HIP_TRY(hipSetDevice(c->device));
    const int nb = std::min<int>(max_batches, (int)ZC_TRACE_BATCHES);
    std::vector<unsigned long long> all;
    try { all.resize((size_t)ZC_TRACE_BATCHES * z.n_workers * 5u); } catch (...) { return MI_BLUR_ERR_NOMEM; }
    HIP_TRY(hipMemcpy(all.data(), z.trace, all.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));   // (waits for idle servers to leave)
    // the context's first nb batches (the trace covers batches 0 .. ZC_TRACE_BATCHES-1 of a context)
    const int have = (int)std::min<unsigned>(z.head, (unsigned)nb);
    memcpy(out, all.data(), (size_t)have * z.n_workers * 5u * sizeof(unsigned long long));
    const int nb_out = have;
    if (n_workers) *n_workers = (int)z.n_workers;
    if (batches_published) *batches_published = z.head;
    return nb_out;
}

extern "C" uint64_t mi_blur_fnv1a64(const uint8_t *host, size_t n) { return fnv1a64(host, n); }

// heterogeneous_blur.c:449-458 — evaluated in float exactly as there.
extern "C" void mi_blur_a1_partition(int mode, int batch_count, float gpu_ratio, int *n_cpu, int *n_gpu)
{
    int nc, ng;
    if (mode == 0) { ng = (int)(batch_count * gpu_ratio); nc = batch_count - ng; }
    else if (mode == 1) { nc = batch_count; ng = 0; }
    else { nc = 0; ng = batch_count; }
    if (n_cpu) *n_cpu = nc;
    if (n_gpu) *n_gpu = ng;
}

extern "C" void mi_blur_shard_range(long long n_units, int g, int G, long long *begin, long long *end)
{
    if (G <= 0) G = 1;
    if (begin) *begin = n_units * g / G;
    if (end) *end = n_units * (g + 1) / G;
}

// split_image_blur.c:144-166.
extern "C" void mi_blur_a2_split(int height, float gpu_ratio, int halo, mi_blur_a2_geometry *g)
{
    if (!g) return;
    int split_row = (int)(height * (1.0f - gpu_ratio));
    if (split_row < halo) split_row = halo;
    if (split_row > height - halo) split_row = height - halo;
    g->split_row = split_row;
    g->cpu_input_rows = split_row + halo;
    g->cpu_output_rows = split_row;
    g->gpu_input_rows = (height - split_row) + halo;
    g->gpu_output_rows = height - split_row;
}

extern "C" void mi_blur_band_of(int height, int radius, int g, int G, mi_blur_band *b)
{
    if (!b) return;
    if (G <= 0) G = 1;
    b->row_begin = (int)((long long)height * g / G);
    b->row_end = (int)((long long)height * (g + 1) / G);
    b->halo_top = std::min(radius, b->row_begin);
    b->halo_bottom = std::min(radius, height - b->row_end);
}

// ----------------------------------------------------------------------------------
// RCCL halo exchange (Approach 2 on resident row shards)
// ----------------------------------------------------------------------------------
namespace {

struct Rccl {
    void *h = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclCommUserRank) CommUserRank = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    bool ok = false;
};

Rccl &rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // A process that already holds an RCCL (torch's bundled librccl.so has no soname and is registered under
        // that name) must keep using that one: a second copy would sit on the same HIP runtime.
        for (const char *name : {"librccl.so", "librccl.so.1"}) {
            r.h = dlopen(name, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
            if (r.h) break;
        }
        if (!r.h)
            for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
                r.h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
                if (r.h) break;
            }
        if (!r.h) return;
#define MI_SYM(field, sym) r.field = (decltype(r.field))dlsym(r.h, #sym)
        MI_SYM(GetUniqueId, ncclGetUniqueId);
        MI_SYM(CommInitRank, ncclCommInitRank);
        MI_SYM(CommInitAll, ncclCommInitAll);
        MI_SYM(CommDestroy, ncclCommDestroy);
        MI_SYM(CommCount, ncclCommCount);
        MI_SYM(CommUserRank, ncclCommUserRank);
        MI_SYM(Send, ncclSend);
        MI_SYM(Recv, ncclRecv);
        MI_SYM(GroupStart, ncclGroupStart);
        MI_SYM(GroupEnd, ncclGroupEnd);
#undef MI_SYM
        r.ok = r.GetUniqueId && r.CommInitRank && r.CommInitAll && r.CommDestroy && r.Send && r.Recv &&
               r.GroupStart && r.GroupEnd;
    });
    return r;
}

inline int nccl_status(ncclResult_t e) { return e == ncclSuccess ? MI_BLUR_OK : MI_BLUR_ERR_RCCL_BASE - (int)e; }

}  // namespace

struct mi_blur_comm {
    ncclComm_t comm = nullptr;
    int n_ranks = 1, rank = 0, device = -1;
    bool p2p = false;                            // single-process copy transport instead of RCCL
    bool pull = false;                           // ... whose copies are PULLS: one small kernel per rank reads the neighbours' rows
    hipEvent_t ev_prev = nullptr, ev_push = nullptr;
};

static_assert(sizeof(ncclUniqueId) == MI_BLUR_UNIQUE_ID_BYTES, "ncclUniqueId size");

extern "C" int mi_blur_comm_unique_id(uint8_t id[MI_BLUR_UNIQUE_ID_BYTES])
{
    if (!id) return MI_BLUR_ERR_INVALID;
    Rccl &r = rccl();
    if (!r.ok) return MI_BLUR_ERR_UNSUPPORTED;
    ncclUniqueId u;
    int rc = nccl_status(r.GetUniqueId(&u));
    if (rc) return rc;
    memcpy(id, &u, sizeof u);
    return MI_BLUR_OK;
}

extern "C" int mi_blur_comm_init_rank(mi_blur_comm **comm, int n_ranks, int rank,
                                      const uint8_t id[MI_BLUR_UNIQUE_ID_BYTES])
{
    if (!comm || !id || n_ranks <= 0 || rank < 0 || rank >= n_ranks) return MI_BLUR_ERR_INVALID;
    *comm = nullptr;
    mi_blur_comm *c = new (std::nothrow) mi_blur_comm;
    if (!c) return MI_BLUR_ERR_NOMEM;
    c->n_ranks = n_ranks; c->rank = rank;
    if (hipGetDevice(&c->device) != hipSuccess) { (void)hipGetLastError(); delete c; return MI_BLUR_ERR_NO_DEVICE; }
    if (n_ranks > 1) {
        Rccl &r = rccl();
        if (!r.ok) { delete c; return MI_BLUR_ERR_UNSUPPORTED; }
        ncclUniqueId u;
        memcpy(&u, id, sizeof u);
        int rc = nccl_status(r.CommInitRank(&c->comm, n_ranks, u, rank));
        if (rc) { delete c; return rc; }
    }
    *comm = c;
    return MI_BLUR_OK;
}

extern "C" int mi_blur_comm_init_all(mi_blur_comm **comms, int n_devices, const int *devices)
{
    if (!comms || n_devices <= 0) return MI_BLUR_ERR_INVALID;
    std::vector<ncclComm_t> raw(n_devices, nullptr);
    std::vector<int> devs(n_devices);
    for (int i = 0; i < n_devices; i++) devs[i] = devices ? devices[i] : i;
    if (n_devices > 1) {
        Rccl &r = rccl();
        if (!r.ok) return MI_BLUR_ERR_UNSUPPORTED;
        int rc = nccl_status(r.CommInitAll(raw.data(), n_devices, devs.data()));
        if (rc) return rc;
    }
    for (int i = 0; i < n_devices; i++) comms[i] = nullptr;
    for (int i = 0; i < n_devices; i++) {
        mi_blur_comm *c = new (std::nothrow) mi_blur_comm;
        if (!c) {
            // give back everything made so far: the wrappers already built (each destroys its RCCL communicator)
            // and the raw communicators that have no wrapper yet
            for (int j = 0; j < i; j++) { mi_blur_comm_destroy(comms[j]); comms[j] = nullptr; }
            for (int j = i; j < n_devices; j++) if (raw[j]) (void)rccl().CommDestroy(raw[j]);
            return MI_BLUR_ERR_NOMEM;
        }
        c->comm = raw[i]; c->n_ranks = n_devices; c->rank = i; c->device = devs[i];
        comms[i] = c;
    }
    return MI_BLUR_OK;
}

// Single-process communicator set whose halo rows move with hipMemcpyPeerAsync instead of RCCL: the
// fallback when RCCL is unavailable, and what lets the row-shard flow run with several shards per device.
extern "C" int mi_blur_comm_init_p2p(mi_blur_comm **comms, int n_devices, const int *devices)
{
    if (!comms || n_devices <= 0) return MI_BLUR_ERR_INVALID;
    const int ndev = mi_blur_device_count();
    if (ndev <= 0) return MI_BLUR_ERR_NO_DEVICE;
    for (int i = 0; i < n_devices; i++) comms[i] = nullptr;
    // any failure gives back every rank made so far (mi_blur_comm_destroy releases the events a rank already holds)
    auto fail = [&](int rc) {
        for (int j = 0; j < n_devices; j++) { mi_blur_comm_destroy(comms[j]); comms[j] = nullptr; }
        return rc;
    };
    for (int i = 0; i < n_devices; i++) {
        mi_blur_comm *c = new (std::nothrow) mi_blur_comm;
        if (!c) return fail(MI_BLUR_ERR_NOMEM);
        c->n_ranks = n_devices; c->rank = i; c->device = devices ? devices[i] : i; c->p2p = true;
        comms[i] = c;
        if (c->device < 0 || c->device >= ndev) return fail(MI_BLUR_ERR_NO_DEVICE);
        hipError_t e = hipSetDevice(c->device);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_prev, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_push, hipEventDisableTiming);
        if (e != hipSuccess) { (void)hipGetLastError(); return fail(MI_BLUR_ERR_HIP_BASE - (int)e); }
    }
    for (int i = 0; i + 1 < n_devices; i++) {     // neighbours on different devices: enable direct access both ways (best effort)
        const int a = comms[i]->device, b = comms[i + 1]->device;
        if (a == b) continue;
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, a, b) == hipSuccess && can) { (void)hipSetDevice(a); (void)hipDeviceEnablePeerAccess(b, 0); }
        if (hipDeviceCanAccessPeer(&can, b, a) == hipSuccess && can) { (void)hipSetDevice(b); (void)hipDeviceEnablePeerAccess(a, 0); }
        (void)hipGetLastError();
    }
    return MI_BLUR_OK;
}

// The same single-process set with the halo rows PULLED: every rank runs one small kernel (mi_blur_halo_pull's) that reads its
// neighbours' edge rows through peer access, instead of pushing its own with two hipMemcpyPeerAsync.
extern "C" int mi_blur_comm_init_pull(mi_blur_comm **comms, int n_devices, const int *devices)
{
    const int rc = mi_blur_comm_init_p2p(comms, n_devices, devices);
    if (rc) return rc;
    for (int i = 0; i < n_devices; i++) comms[i]->pull = true;
    return MI_BLUR_OK;
}

// What a communicator IS, as the transport itself reports it: a bench line that says "RCCL carried the halos over N
// ranks" quotes ncclCommCount / ncclCommUserRank, not the number it asked for.
extern "C" int mi_blur_comm_info(mi_blur_comm *c, int *n_ranks, int *rank, int *transport)
{
    if (!c) return MI_BLUR_ERR_INVALID;
    int n = c->n_ranks, r = c->rank, t = c->pull ? 3 : c->p2p ? 2 : (c->comm ? 1 : 0);
    if (c->comm) {
        Rccl &rc = rccl();
        if (!rc.ok || !rc.CommCount || !rc.CommUserRank) return MI_BLUR_ERR_UNSUPPORTED;
        int e = nccl_status(rc.CommCount(c->comm, &n));
        if (!e) e = nccl_status(rc.CommUserRank(c->comm, &r));
        if (e) return e;
    }
    if (n_ranks) *n_ranks = n;
    if (rank) *rank = r;
    if (transport) *transport = t;
    return MI_BLUR_OK;
}

extern "C" void mi_blur_comm_destroy(mi_blur_comm *c)
{
    if (!c) return;
    if (c->comm) (void)rccl().CommDestroy(c->comm);
    if (c->ev_prev || c->ev_push) {
        (void)hipSetDevice(c->device);
        if (c->ev_prev) (void)hipEventDestroy(c->ev_prev);
        if (c->ev_push) (void)hipEventDestroy(c->ev_push);
    }
    delete c;
}

static int halo_exchange_calls(Rccl &r, mi_blur_comm *c, uint8_t *d_band, size_t pitch, int owned_rows, int radius,
                               hipStream_t stream)
{
    const int top = c->rank > 0 ? radius : 0;
    const size_t n = pitch * (size_t)radius;
    ncclResult_t e = ncclSuccess;
    if (c->rank > 0) {
        if ((e = r.Send(d_band + (size_t)top * pitch, n, ncclUint8, c->rank - 1, c->comm, stream)) != ncclSuccess) return nccl_status(e);
        if ((e = r.Recv(d_band, n, ncclUint8, c->rank - 1, c->comm, stream)) != ncclSuccess) return nccl_status(e);
    }
    if (c->rank < c->n_ranks - 1) {
        uint8_t *last = d_band + (size_t)(top + owned_rows - radius) * pitch;
        if ((e = r.Send(last, n, ncclUint8, c->rank + 1, c->comm, stream)) != ncclSuccess) return nccl_status(e);
        if ((e = r.Recv(d_band + (size_t)(top + owned_rows) * pitch, n, ncclUint8, c->rank + 1, c->comm, stream)) != ncclSuccess) return nccl_status(e);
    }
    return MI_BLUR_OK;
}

extern "C" int mi_blur_halo_exchange(mi_blur_comm *c, uint8_t *d_band, int width, int channels, int owned_rows,
                                     int radius, void *stream)
{
    if (!c || !d_band || width <= 0 || channels <= 0 || radius < 1 || owned_rows < radius) return MI_BLUR_ERR_INVALID;
    if (c->n_ranks == 1) return MI_BLUR_OK;            // nothing to exchange: both edges clamp
    if (c->p2p) return MI_BLUR_ERR_STATE;              // the copy transport needs every rank: mi_blur_halo_exchange_all
    Rccl &r = rccl();
    if (!r.ok || !c->comm) return MI_BLUR_ERR_UNSUPPORTED;
    int rc = nccl_status(r.GroupStart());
    if (rc) return rc;
    rc = halo_exchange_calls(r, c, d_band, (size_t)width * channels, owned_rows, radius, (hipStream_t)stream);
    int rc2 = nccl_status(r.GroupEnd());
    return rc ? rc : rc2;
}

// ----------------------------------------------------------------------------------
// Halo pull: a rank reads its halo rows straight out of its neighbours' shards (peer memory) with one small kernel.
// ----------------------------------------------------------------------------------
static_assert(sizeof(hipIpcMemHandle_t) == MI_BLUR_PEER_HANDLE_BYTES, "hipIpcMemHandle_t size");

extern "C" int mi_blur_peer_export(const void *d_ptr, uint8_t handle[MI_BLUR_PEER_HANDLE_BYTES], uint64_t *offset)
{
    if (!d_ptr || !handle || !offset) return MI_BLUR_ERR_INVALID;
    if (mi_blur_device_count() <= 0) return MI_BLUR_ERR_NO_DEVICE;
    // the handle names a whole allocation; callers (torch's caching allocator, for one) hand out pieces of bigger ones
    hipDeviceptr_t base = nullptr;
    size_t size = 0;
    HIP_TRY(hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)d_ptr));
    hipIpcMemHandle_t h;
    HIP_TRY(hipIpcGetMemHandle(&h, base));
    memcpy(handle, &h, sizeof h);
    *offset = (uint64_t)((const uint8_t *)d_ptr - (const uint8_t *)base);
    return MI_BLUR_OK;
}

extern "C" int mi_blur_peer_open(const uint8_t handle[MI_BLUR_PEER_HANDLE_BYTES], uint64_t offset, void **d_ptr)
{
    if (!handle || !d_ptr) return MI_BLUR_ERR_INVALID;
    *d_ptr = nullptr;
    if (mi_blur_device_count() <= 0) return MI_BLUR_ERR_NO_DEVICE;
    hipIpcMemHandle_t h;
    memcpy(&h, handle, sizeof h);
    void *base = nullptr;
    HIP_TRY(hipIpcOpenMemHandle(&base, h, hipIpcMemLazyEnablePeerAccess));
    *d_ptr = (uint8_t *)base + offset;
    return MI_BLUR_OK;
}

extern "C" int mi_blur_peer_close(void *d_ptr, uint64_t offset)
{
    if (!d_ptr) return MI_BLUR_OK;
    HIP_TRY(hipIpcCloseMemHandle((uint8_t *)d_ptr - offset));
    return MI_BLUR_OK;
}

extern "C" int mi_blur_halo_pull(uint8_t *d_band, const uint8_t *top_src, const uint8_t *bottom_src, int width, int channels,
                                 int owned_rows, int radius, void *stream)
{
    if (!d_band || width <= 0 || channels <= 0 || radius < 1 || owned_rows < radius) return MI_BLUR_ERR_INVALID;
    if (mi_blur_device_count() <= 0) return MI_BLUR_ERR_NO_DEVICE;
    const size_t pitch = (size_t)width * channels, n = pitch * (size_t)radius;
    const size_t top = top_src ? (size_t)radius : 0;                      // layout: [halo_top rows][owned rows][halo_bottom rows]
    return launch_halo_pull(top_src, d_band, bottom_src, d_band + (top + (size_t)owned_rows) * pitch, n, (hipStream_t)stream);
}

// All ranks of a single-process communicator set in ONE RCCL group (one host thread
// driving G GPUs must not block on rank 0's group before enqueuing rank 1's).
extern "C" int mi_blur_halo_exchange_all(mi_blur_comm **comms, int n, uint8_t **d_bands, int width, int channels,
                                         const int *owned_rows, int radius, void **streams)
{
    if (!comms || !d_bands || !owned_rows || n <= 0) return MI_BLUR_ERR_INVALID;
    if (n == 1) return MI_BLUR_OK;
    for (int i = 0; i < n; i++) if (!comms[i] || owned_rows[i] < radius) return MI_BLUR_ERR_INVALID;
    if (comms[0]->p2p) {
        // Same rows, same offsets as the RCCL form; each rank PUSHES its edge rows into its neighbours' halo rows on
        // its own stream.  Ordering by events: a push waits until the neighbour has finished whatever it queued
        // before this call (its previous blur may still read those halo rows); a rank's later work waits for the
        // pushes into its halos.
        const size_t pitch = (size_t)width * channels, nbytes = pitch * (size_t)radius;
        auto st = [&](int i) { return streams ? (hipStream_t)streams[i] : (hipStream_t) nullptr; };
        auto top = [&](int i) { return i > 0 ? radius : 0; };
        for (int i = 0; i < n; i++) {
            HIP_TRY(hipSetDevice(comms[i]->device));
            HIP_TRY(hipEventRecord(comms[i]->ev_prev, st(i)));
        }
        for (int i = 0; i < n; i++) {
            HIP_TRY(hipSetDevice(comms[i]->device));
            if (comms[0]->pull) {
                // PULL: rank i reads the last owned rows of rank i-1 and the first owned rows of rank i+1 into its own halo rows
                // with one kernel on its own stream, once both neighbours have finished what they queued before this call
                // (their owned rows are final).  ev_push(i) = "rank i has read its neighbours' rows": they wait for it below
                // before anything they queue later may overwrite those rows.
                const uint8_t *top_src = nullptr, *bottom_src = nullptr;
                if (i > 0) {
                    HIP_TRY(hipStreamWaitEvent(st(i), comms[i - 1]->ev_prev, 0));
                    top_src = d_bands[i - 1] + (size_t)(top(i - 1) + owned_rows[i - 1] - radius) * pitch;
                }
                if (i < n - 1) {
                    HIP_TRY(hipStreamWaitEvent(st(i), comms[i + 1]->ev_prev, 0));
                    bottom_src = d_bands[i + 1] + (size_t)top(i + 1) * pitch;
                }
                const int rc = launch_halo_pull(top_src, d_bands[i], bottom_src, d_bands[i] + (size_t)(top(i) + owned_rows[i]) * pitch, nbytes, st(i));
                if (rc) return rc;
                HIP_TRY(hipEventRecord(comms[i]->ev_push, st(i)));
                continue;
            }
            if (i > 0) {          // first owned rows -> bottom halo of rank i-1
                HIP_TRY(hipStreamWaitEvent(st(i), comms[i - 1]->ev_prev, 0));
                uint8_t *dst = d_bands[i - 1] + (size_t)(top(i - 1) + owned_rows[i - 1]) * pitch;
                HIP_TRY(hipMemcpyPeerAsync(dst, comms[i - 1]->device, d_bands[i] + (size_t)top(i) * pitch, comms[i]->device, nbytes, st(i)));
            }
            if (i < n - 1) {      // last owned rows -> top halo of rank i+1
                HIP_TRY(hipStreamWaitEvent(st(i), comms[i + 1]->ev_prev, 0));
                const uint8_t *src = d_bands[i] + (size_t)(top(i) + owned_rows[i] - radius) * pitch;
                HIP_TRY(hipMemcpyPeerAsync(d_bands[i + 1], comms[i + 1]->device, src, comms[i]->device, nbytes, st(i)));
            }
            HIP_TRY(hipEventRecord(comms[i]->ev_push, st(i)));
        }
        for (int i = 0; i < n; i++) {
            HIP_TRY(hipSetDevice(comms[i]->device));
            if (i > 0) HIP_TRY(hipStreamWaitEvent(st(i), comms[i - 1]->ev_push, 0));
            if (i < n - 1) HIP_TRY(hipStreamWaitEvent(st(i), comms[i + 1]->ev_push, 0));
        }
        return MI_BLUR_OK;
    }
    Rccl &r = rccl();
    if (!r.ok) return MI_BLUR_ERR_UNSUPPORTED;
    int rc = nccl_status(r.GroupStart());
    if (rc) return rc;
    for (int i = 0; i < n && !rc; i++) {
        if (hipSetDevice(comms[i]->device) != hipSuccess) { (void)hipGetLastError(); rc = MI_BLUR_ERR_NO_DEVICE; break; }
        rc = halo_exchange_calls(r, comms[i], d_bands[i], (size_t)width * channels, owned_rows[i], radius,
                                 streams ? (hipStream_t)streams[i] : nullptr);
    }
    int rc2 = nccl_status(r.GroupEnd());
    return rc ? rc : rc2;
}
