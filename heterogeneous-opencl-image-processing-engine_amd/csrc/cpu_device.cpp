// cpu_device.cpp — the engine's host-thread device and host-side helpers.
//
// The reference runs the same OpenCL kernel on a CL_DEVICE_TYPE_CPU device
// (heterogeneous_blur.c:170-176,507).  ROCm's OpenCL has no CPU device, so the
// `cpu` / `both` modes run this native implementation instead: same arithmetic
// (integer form of gaussian_kernel.cl:19-72, see blur_kernels.hip header), rows
// processed separably with an edge-replicated scratch row so the inner loops are
// branch-free and auto-vectorise.  This is a DEVICE THE CALLER ASKS FOR BY NAME
// (MI_BLUR_DEVICE_CPU), never a fallback for a missing GPU, and it shares no code
// with oracle/ (which is test infrastructure).
#include "cpu_device.h"

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace mi_blur {

int hardware_threads()
{
    unsigned n = std::thread::hardware_concurrency();
    return n ? (int)n : 1;
}

// Persistent worker pool: a batch of 10-35 small images is ~1 ms of work, less than spawning a thread per core.
// Workers are created on first use (up to the hardware thread count) and reused by every parallel_for; callers
// are serialised (one parallel_for at a time), which is what a single CPU device wants anyway.
namespace {
class Pool {
public:
    static Pool &get() { static Pool p; return p; }
    // run fn(worker_index) on n_workers threads (including the caller); returns when all are done
    void run(int n_workers, const std::function<void(int)> &fn)
    {
        if (n_workers <= 1) { fn(0); return; }
        std::lock_guard<std::mutex> call(call_m_);
        grow(n_workers - 1);
        // Spinning pays only while batches follow each other closely (a stream: one every ~100 us); a caller that does something
        // long between two uses of the pool — builds a batch on one thread, say — should find the helpers asleep, not burning
        // the cores (and the cgroup's CPU quota) it is working on.  The gap since the previous use decides.
        const auto now = std::chrono::steady_clock::now();
        const double gap_us = std::chrono::duration<double, std::micro>(now - last_done_).count();
        // OFF by default (MI_BLUR_POOL_SPIN=<pause count> turns it on): measured, it is worth +30 % to a caller that does nothing
        // but feed the pool on an idle 8-vCPU VM, nothing on the 256-thread GPU hosts (a sleeping worker is woken fast enough
        // there: 71-95 us per 35-image batch on 16 threads either way), and it costs 4x inside the hosts on the small VM, where
        // the spinners take the cores the batch-building threads need
        static const int spin_max = [] { const char *e = getenv("MI_BLUR_POOL_SPIN"); const int v = e ? atoi(e) : -1; return v >= 0 ? v : 0; }();
        spin_budget_.store(gap_us < 250.0 ? spin_max : 0, std::memory_order_relaxed);
        {
            std::lock_guard<std::mutex> lk(m_);
            fn_ = &fn; active_ = n_workers - 1; pending_ = n_workers - 1; gen_++;
            pending_hint_.store(pending_, std::memory_order_release);
            gen_hint_.store(gen_, std::memory_order_release);
        }
        cv_.notify_all();
        fn(0);
        // the helpers finish within microseconds of the caller: look before sleeping
        for (int i = 0; i < 4000 && pending_hint_.load(std::memory_order_acquire) != 0; i++) cpu_relax();
        std::unique_lock<std::mutex> lk(m_);
        done_.wait(lk, [this] { return pending_ == 0; });
        fn_ = nullptr;
        last_done_ = std::chrono::steady_clock::now();
    }
    ~Pool()
    {
        { std::lock_guard<std::mutex> lk(m_); stop_ = true; gen_++; gen_hint_.store(gen_, std::memory_order_release); }
        cv_.notify_all();
        for (auto &t : th_) t.join();
    }

private:
    void grow(int n)
    {
        while ((int)th_.size() < n) {
            const int id = (int)th_.size() + 1;
            th_.emplace_back([this, id] { loop(id); });
        }
    }
    void loop(int id)
    {
        unsigned long long seen = 0;
        for (;;) {
            const std::function<void(int)> *fn = nullptr;
            // a stream of batches wakes the pool every ~100 us: a worker that has just finished spins that long for the next
            // generation before it goes to sleep on the condition variable (a sleeping worker costs the batch ~30-50 us)
            for (int i = 0, n = spin_budget_.load(std::memory_order_relaxed); i < n && gen_hint_.load(std::memory_order_acquire) == seen; i++) cpu_relax();
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return gen_ != seen; });
                seen = gen_;
                if (stop_) return;
                if (id <= active_) fn = fn_;
            }
            if (fn) {
                (*fn)(id);
                std::lock_guard<std::mutex> lk(m_);
                pending_hint_.store(pending_ - 1, std::memory_order_release);
                if (--pending_ == 0) done_.notify_one();
            }
        }
    }
    static void cpu_relax()
    {
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#endif
    }
    std::atomic<int> spin_budget_{0};
    std::chrono::steady_clock::time_point last_done_{};
    std::atomic<unsigned long long> gen_hint_{0};
    std::atomic<int> pending_hint_{0};
    std::mutex call_m_, m_;
    std::condition_variable cv_, done_;
    std::vector<std::thread> th_;
    const std::function<void(int)> *fn_ = nullptr;
    unsigned long long gen_ = 0;
    int active_ = 0, pending_ = 0;
    bool stop_ = false;
};
}  // namespace

// Rows [y_begin, y_end) of one image / band of H rows.
void cpu_blur_rows(const uint8_t *in, uint8_t *out, int W, int H, int C, int R, int y_begin, int y_end,
                   int out_row_shift)
{
    const int pitch = W * C, pad = R * C;
    std::vector<uint16_t> scratch((size_t)pitch + 2 * pad);
    uint16_t *v = scratch.data() + pad;               // v[-pad .. pitch+pad)
    for (int y = y_begin; y < y_end; y++) {
        // vertical pass: v[b] = sum_k taps[k] * in[clamp(y+k)][b]   (<= 4080)
        const uint8_t *rows[5];
        for (int k = -R; k <= R; k++) rows[k + R] = in + (size_t)std::min(std::max(y + k, 0), H - 1) * pitch;
        if (R == 1) {
            const uint8_t *a = rows[0], *b = rows[1], *c = rows[2];
            for (int i = 0; i < pitch; i++) v[i] = (uint16_t)(a[i] + 2 * b[i] + c[i]);
        } else {
            const uint8_t *a = rows[0], *b = rows[1], *c = rows[2], *d = rows[3], *e = rows[4];
            for (int i = 0; i < pitch; i++) v[i] = (uint16_t)(a[i] + e[i] + 4 * (b[i] + d[i]) + 6 * c[i]);
        }
        // clamp-to-edge in x == replicate the first/last pixel's channels outward
        for (int k = 1; k <= pad; k++) {
            v[-k] = v[((-k % C) + C) % C];
            v[pitch + k - 1] = v[pitch - C + ((k - 1) % C)];
        }
        // horizontal pass on the 16-bit sums, one truncating shift (gaussian_kernel.cl:70)
        uint8_t *o = out + (size_t)(y - out_row_shift) * pitch;
        if (R == 1) {
            for (int i = 0; i < pitch; i++) o[i] = (uint8_t)((v[i - C] + 2 * v[i] + v[i + C]) >> 4);
        } else {
            for (int i = 0; i < pitch; i++)
                o[i] = (uint8_t)((v[i - 2 * C] + v[i + 2 * C] + 4 * (v[i - C] + v[i + C]) + 6 * v[i]) >> 8);
        }
    }
}

// n_images bands of band_rows rows; output rows [y0,y1) of each.  Threads take whole
// images when there are enough of them, else row slices of each image.
void cpu_blur_batch(const uint8_t *in, uint8_t *out, int W, int band_rows, int C, int R, int n_images,
                    int y0, int y1, int n_threads, size_t in_stride, size_t out_stride)
{
    if (n_images <= 0) return;
    if (n_threads <= 0) n_threads = hardware_threads();
    if (in_stride == 0) in_stride = (size_t)W * C * band_rows;
    if (out_stride == 0) out_stride = (size_t)W * C * (y1 - y0);
    const int rows = y1 - y0;
    // work items: (image, row slice)
    // enough items for the threads to end together: a batch of 35 images on 16 threads is three rounds of whole images with the
    // last one half empty; cut into row slices (of at least 16 rows) until there are ~4 items per thread
    static const int per_thread = [] { const char *e = getenv("MI_BLUR_POOL_ITEMS"); const int v = e ? atoi(e) : 0; return v >= 1 && v <= 64 ? v : 4; }();
    int slices = 1;
    if (n_images < per_thread * n_threads)
        slices = std::max(1, std::min(std::max(1, rows / 16), (per_thread * n_threads + n_images - 1) / n_images));
    const long long items = (long long)n_images * slices;
    std::atomic<long long> next{0};
    auto worker = [&]() {
        for (;;) {
            const long long it = next.fetch_add(1, std::memory_order_relaxed);
            if (it >= items) break;
            const int img = (int)(it / slices), s = (int)(it % slices);
            const int ys = y0 + (int)((long long)rows * s / slices), ye = y0 + (int)((long long)rows * (s + 1) / slices);
            cpu_blur_rows(in + img * in_stride, out + img * out_stride, W, band_rows, C, R, ys, ye, y0);
        }
    };
    const int nt = (int)std::min<long long>(n_threads, items);
    Pool::get().run(nt, [&](int) { worker(); });
}

// Synthetic stream: image i = LCG bytes, seed 0x9E3779B9 ^ i.
void fill_synthetic(uint8_t *host, int W, int H, int C, int first_index, int n_images, int n_threads)
{
    if (n_images <= 0) return;
    if (n_threads <= 0) n_threads = hardware_threads();
    const size_t isz = (size_t)W * H * C;
    std::atomic<int> next{0};
    auto worker = [&]() {
        for (;;) {
            const int i = next.fetch_add(1, std::memory_order_relaxed);
            if (i >= n_images) break;
            uint32_t s = 0x9E3779B9u ^ (uint32_t)(first_index + i);
            uint8_t *p = host + (size_t)i * isz;
            for (size_t k = 0; k < isz; k++) {
                s = s * 1664525u + 1013904223u;
                p[k] = (uint8_t)(s >> 24);
            }
        }
    };
    Pool::get().run(std::min(n_threads, n_images), [&](int) { worker(); });
}

// n blocks of `bytes` bytes from src (stride src_stride) to dst (stride dst_stride), split over a few pool threads when
// there is enough to move: the staging copies of a GPU context fed from pageable caller memory.
void copy_blocks(uint8_t *dst, size_t dst_stride, const uint8_t *src, size_t src_stride, size_t bytes, int n, int n_threads)
{
    const size_t total = bytes * (size_t)n;
    if (n_threads <= 1 || total < (1u << 20)) {
        for (int i = 0; i < n; i++) memcpy(dst + (size_t)i * dst_stride, src + (size_t)i * src_stride, bytes);
        return;
    }
    // cut every block into n_threads slices so the split is even whatever n is
    Pool::get().run(n_threads, [&](int t) {
        const size_t b = bytes * (size_t)t / (size_t)n_threads, e = bytes * (size_t)(t + 1) / (size_t)n_threads;
        for (int i = 0; i < n; i++) memcpy(dst + (size_t)i * dst_stride + b, src + (size_t)i * src_stride + b, e - b);
    });
}

// The reference's host loops (heterogeneous_blur.c:125-134 in, split_image_blur.c:40-56 out), over images on pool threads.
void cpu_repack(const uint8_t *src, uint8_t *dst, int W, int H, int C, int n_images, bool p2i, int n_threads)
{
    if (n_images <= 0) return;
    if (n_threads <= 0) n_threads = hardware_threads();
    const size_t plane = (size_t)W * H, isz = plane * C;
    std::atomic<int> next{0};
    Pool::get().run(std::min(n_threads, n_images), [&](int) {
        for (;;) {
            const int i = next.fetch_add(1, std::memory_order_relaxed);
            if (i >= n_images) break;
            const uint8_t *s = src + (size_t)i * isz;
            uint8_t *d = dst + (size_t)i * isz;
            for (int c = 0; c < C; c++)
                for (size_t px = 0; px < plane; px++) {
                    if (p2i) d[px * C + c] = s[(size_t)c * plane + px];
                    else d[(size_t)c * plane + px] = s[px * C + c];
                }
        }
    });
}

uint64_t fnv1a64(const uint8_t *p, size_t n)
{
    uint64_t h = 0xcbf29ce484222325ull;
    for (size_t i = 0; i < n; i++) { h ^= p[i]; h *= 0x100000001b3ull; }
    return h;
}

}  // namespace mi_blur
