// host_common.h — shared pieces of the two C++ hosts (heterogeneous_blur, split_image_blur).
//
// The hosts keep the reference's positional command lines, banners and report sections
// (heterogeneous_blur.c:41-100,609-724; split_image_blur.c:62-102,615-721) and call the HIP
// kernels only through the C ABI in include/mi_blur.h.  What the reference did with OpenCL
// plumbing inline in main() lives behind that ABI; what it did with CImg lives here:
//   * image input: binary PPM/PGM always; any CImg-readable file when built with
//     -DMI_BLUR_WITH_CIMG -I<dir holding CImg.h> (CImg is the reference's third-party
//     dependency and is NOT vendored here); otherwise a synthetic LCG image of the
//     reference's default geometry (320x240x3, heterogeneous_blur.c:43);
//   * image output (split_image_blur.c:40-56): PPM, or CImg when enabled.
#pragma once

#include "mi_blur.h"

#include <dirent.h>
#include <sys/stat.h>
#include <sys/time.h>

#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <algorithm>
#include <string>
#include <thread>
#include <vector>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif

#ifdef MI_BLUR_WITH_CIMG
#define cimg_display 0
#include "CImg.h"
#endif

namespace host {

// cl_error() of the reference (heterogeneous_blur.c:25-30): print "<code> - <msg>" and exit(-1).
inline void mi_check(int code, const char *what)
{
    if (code != MI_BLUR_OK) {
        printf("%d - %s (%s)\n", code, what, mi_blur_strerror(code));
        exit(-1);
    }
}

// heterogeneous_blur.c:32-36
inline double get_time_ms()
{
    struct timeval tv;
    gettimeofday(&tv, NULL);
    return (tv.tv_sec * 1000.0) + (tv.tv_usec / 1000.0);
}

struct Image {
    int width = 0, height = 0, channels = 0;
    std::vector<uint8_t> px;             // interleaved, pitch = width*channels
    std::string source;
    size_t bytes() const { return px.size(); }
};

inline bool skip_ws_comments(FILE *f)
{
    int c;
    while ((c = fgetc(f)) != EOF) {
        if (c == '#') { while ((c = fgetc(f)) != EOF && c != '\n') {} }
        else if (c != ' ' && c != '\t' && c != '\n' && c != '\r') { ungetc(c, f); return true; }
    }
    return false;
}

inline bool load_pnm(const char *path, Image &img)
{
    FILE *f = fopen(path, "rb");
    if (!f) return false;
    char magic[3] = {0, 0, 0};
    if (fread(magic, 1, 2, f) != 2 || magic[0] != 'P' || (magic[1] != '6' && magic[1] != '5')) { fclose(f); return false; }
    int w = 0, h = 0, maxv = 0;
    if (!skip_ws_comments(f) || fscanf(f, "%d", &w) != 1 || !skip_ws_comments(f) || fscanf(f, "%d", &h) != 1 ||
        !skip_ws_comments(f) || fscanf(f, "%d", &maxv) != 1 || maxv != 255 || w <= 0 || h <= 0) { fclose(f); return false; }
    fgetc(f);   // single whitespace after maxval
    img.width = w; img.height = h; img.channels = magic[1] == '6' ? 3 : 1;
    img.px.resize((size_t)w * h * img.channels);
    const bool ok = fread(img.px.data(), 1, img.px.size(), f) == img.px.size();
    fclose(f);
    img.source = path;
    return ok;
}

inline bool save_pnm(const char *path, const uint8_t *interleaved, int w, int h, int c)
{
    if (c != 1 && c != 3) return false;
    FILE *f = fopen(path, "wb");
    if (!f) return false;
    fprintf(f, "P%c\n%d %d\n255\n", c == 3 ? '6' : '5', w, h);
    const bool ok = fwrite(interleaved, 1, (size_t)w * h * c, f) == (size_t)w * h * c;
    fclose(f);
    return ok;
}

// Reference input path: CImg<unsigned char> img(file) + planar -> interleaved
// (heterogeneous_blur.c:106-135).  Falls back to PNM, then to a synthetic image.
inline Image load_image(const std::string &path, int syn_w, int syn_h, int syn_c, bool force_synthetic)
{
    Image img;
    if (!force_synthetic && !path.empty()) {
#ifdef MI_BLUR_WITH_CIMG
        try {
            cimg_library::CImg<unsigned char> ci(path.c_str());
            img.width = ci.width(); img.height = ci.height(); img.channels = ci.spectrum();
            img.px.resize((size_t)img.width * img.height * img.channels);
            for (int y = 0; y < img.height; y++)
                for (int x = 0; x < img.width; x++)
                    for (int c = 0; c < img.channels; c++)
                        img.px[((size_t)y * img.width + x) * img.channels + c] = ci(x, y, 0, c);
            img.source = path;
            return img;
        } catch (...) {}
#endif
        if (load_pnm(path.c_str(), img)) return img;
        printf("Note: cannot read %s (built %s CImg; PPM/PGM always supported) - using a synthetic %dx%dx%d image\n",
               path.c_str(),
#ifdef MI_BLUR_WITH_CIMG
               "with",
#else
               "without",
#endif
               syn_w, syn_h, syn_c);
    }
    img.width = syn_w; img.height = syn_h; img.channels = syn_c;
    img.px.resize((size_t)syn_w * syn_h * syn_c);
    mi_blur_fill_synthetic(img.px.data(), syn_w, syn_h, syn_c, 0, 1, 1);
    img.source = "synthetic LCG (seed 0x9E3779B9)";
    return img;
}

// split_image_blur.c:40-56
inline void save_one_image(const char *filename, const uint8_t *interleaved, int w, int h, int c)
{
#ifdef MI_BLUR_WITH_CIMG
    cimg_library::CImg<unsigned char> out(w, h, 1, c);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            for (int k = 0; k < c; k++) out(x, y, 0, k) = interleaved[((size_t)y * w + x) * c + k];
    out.save(filename);
#else
    if (!save_pnm(filename, interleaved, w, h, c)) printf("Warning: could not write %s (PPM/PGM only without CImg)\n", filename);
#endif
}

// Optional flags shared by both hosts.  Positional arguments stay exactly the reference's.
struct Options {
    std::string image;                   // --image PATH   (default ./image_320x240.jpg as in the reference)
    bool synthetic = false;              // --synthetic
    int syn_w = 320, syn_h = 240, syn_c = 3;   // --size WxH, --channels C
    int ksize = 3;                       // --ksize 3|5
    int images = 5000;                   // --images N   (NUM_IMAGES, heterogeneous_blur.c:44)
    bool images_given = false;
    int gpus = 1;                        // --gpus G
    bool slots_given = false;
    int slots = 2;                       // --slots S    batch buffer sets in flight.  Not given: 4 where a GPU takes part in
                                         // heterogeneous_blur, 3 in split_image_blur, 2 for the CPU device alone
    int threads = 0;                     // --threads T  CPU device threads (0 = all cores)
    int host_threads = 4;                // --host-threads T  helper threads that build each batch's stream
    bool malloc_buffers = false;         // --malloc     batch buffers from malloc (pageable), as the reference allocates them
    bool verbose = false;                // --verbose    per-batch progress lines (heterogeneous_blur.c:420,463,599)
    bool resident = false;               // --resident   device-resident stream (kernel-only)
    std::string csv;                     // --csv FILE   append one per_run.csv-style row
    std::string save;                    // --save FILE  write the first output image
    std::string save_input;              // --save-input FILE  write the decoded input image (what the stream is built from)
    int iters = 100;                     // --iters N    (split_image_blur --resident)
    bool iterate = false;                // --iterate    (split_image_blur --resident): blur the previous iteration's output
    bool fused = false;                  // --fused      (heterogeneous_blur --resident): one dispatch per GPU for the whole stream,
                                         //              batches counted in as they finish (mi_blur_resident_run_fused)
    bool overlap = false;                // --overlap    (split_image_blur --resident): halo exchange on its own stream, hidden
                                         //              behind the blur of the interior rows; edge rows follow it
    std::string transport = "rccl";      // --transport rccl|p2p|pull|peer  (split_image_blur --resident): halo rows by RCCL, pushed peer copies,
                                         //              or pulled by one small kernel per GPU that reads its neighbours' rows
    bool auto_ratio = false;             // gpu_ratio given as "auto": calibrate on the first batches (heterogeneous_blur both)
    bool size_given = false;
    std::string frames;                  // --frames DIR|PATTERN|FILE  (heterogeneous_blur cpu|gpu): a stream of DISTINCT frames, decoded
                                         //              by the helper threads into pinned PLANAR batch buffers, repacked on the GPU
    std::string save_dir;                // --save-dir DIR   (with --frames) write every blurred frame as DIR/<name>.ppm|.pgm
    bool planar_out = false;             // --planar-out     (with --frames) outputs come back planar; the planar frames are then blurred as
                                         //              C one-channel images each — no repack at all (batch server, in place over PCIe)
    bool native_layout = false;          // --native-layout  (with --frames) a file format that is interleaved on disk (PPM/PGM) is read straight
                                         //              into the pinned INTERLEAVED batch buffer: no host de-interleave, no GPU repack
};

// Returns the number of leading positional arguments (those before the first "--flag").
inline int parse_flags(int argc, char **argv, Options &o)
{
    int npos = argc;
    for (int i = 1; i < argc; i++) if (!strncmp(argv[i], "--", 2)) { npos = i; break; }
    for (int i = npos; i < argc; i++) {
        std::string a = argv[i];
        auto next = [&](const char *name) -> const char * {
            if (i + 1 >= argc) { printf("Error: %s needs a value\n", name); exit(-1); }
            return argv[++i];
        };
        if (a == "--image") o.image = next("--image");
        else if (a == "--synthetic") o.synthetic = true;
        else if (a == "--size") { if (sscanf(next("--size"), "%dx%d", &o.syn_w, &o.syn_h) != 2 || o.syn_w <= 0 || o.syn_h <= 0) { printf("Error: --size WxH\n"); exit(-1); } o.synthetic = true; o.size_given = true; }
        else if (a == "--channels") o.syn_c = atoi(next("--channels"));
        else if (a == "--ksize") { o.ksize = atoi(next("--ksize")); if (o.ksize != 3 && o.ksize != 5) { printf("Error: --ksize must be 3 or 5\n"); exit(-1); } }
        else if (a == "--images") { o.images = atoi(next("--images")); if (o.images < 1) { printf("Error: --images must be >= 1\n"); exit(-1); } o.images_given = true; }
        else if (a == "--gpus") o.gpus = atoi(next("--gpus"));
        else if (a == "--slots") { o.slots = atoi(next("--slots")); if (o.slots < 1) o.slots = 1; o.slots_given = true; }
        else if (a == "--threads") o.threads = atoi(next("--threads"));
        else if (a == "--host-threads") o.host_threads = atoi(next("--host-threads"));
        else if (a == "--verbose") o.verbose = true;
        else if (a == "--resident") o.resident = true;
        else if (a == "--csv") o.csv = next("--csv");
        else if (a == "--save") o.save = next("--save");
        else if (a == "--save-input") o.save_input = next("--save-input");
        else if (a == "--iters") o.iters = atoi(next("--iters"));
        else if (a == "--iterate") o.iterate = true;
        else if (a == "--overlap") o.overlap = true;
        else if (a == "--fused") o.fused = true;
        else if (a == "--frames") o.frames = next("--frames");
        else if (a == "--save-dir") o.save_dir = next("--save-dir");
        else if (a == "--planar-out") o.planar_out = true;
        else if (a == "--native-layout") o.native_layout = true;
        else if (a == "--malloc") o.malloc_buffers = true;
        else if (a == "--transport") { o.transport = next("--transport"); if (o.transport != "rccl" && o.transport != "p2p" && o.transport != "pull" && o.transport != "peer") { printf("Error: --transport rccl|p2p|pull|peer\n"); exit(-1); } }
        else { printf("Error: unknown option %s\n", a.c_str()); exit(-1); }
    }
    return npos;
}

constexpr double HBM_PEAK_GBS = 8000.0;   // MI355X HBM3E spec; ~6290 GB/s measured copy ceiling

// Logical GPU g -> HIP ordinal.  Normally the identity (and G must not exceed the visible devices).  With
// MI_BLUR_VIRTUAL_GPUS=1 (tests on a one-GPU box) G logical GPUs share the visible devices round-robin, each
// with its own context and streams, so the sharding logic of the hosts runs unchanged.
inline bool virtual_gpus() { const char *e = getenv("MI_BLUR_VIRTUAL_GPUS"); return e && atoi(e) != 0; }
inline int hip_ordinal(int g) { const int n = mi_blur_device_count(); return virtual_gpus() && n > 0 ? g % n : g; }
inline bool gpus_available(int G) { const int n = mi_blur_device_count(); return n >= 1 && G >= 1 && (G <= n || virtual_gpus()); }

// Host placement banner: which CPUs the feeder / batch-building threads of a GPU keep to (its socket's), from sysfs.
// Binds the CALLING thread as a side effect when `bind` is set.  One line per GPU; tests parse it.
inline void report_placement(int g, int ordinal, bool bind)
{
    char list[512] = "";
    int node = -1;
    const int rc = mi_blur_device_cpulist(ordinal, list, sizeof list, &node);
    const char *off = getenv("MI_BLUR_NO_AFFINITY");
    if (off && atoi(off) != 0) { printf("GPU %d host placement: not pinned (MI_BLUR_NO_AFFINITY set)\n", g); return; }
    if (rc != MI_BLUR_OK || !list[0]) { printf("GPU %d host placement: not pinned (topology not exposed by sysfs)\n", g); return; }
    const int n = bind ? mi_blur_bind_thread_to_device(ordinal) : -1;
    if (bind && n == 0) { printf("GPU %d host placement: not pinned (no allowed CPU among %s)\n", g, list); return; }
    printf("GPU %d host placement: feeder + batch-building threads and pinned buffers on CPUs %s (NUMA node %d)\n", g, list, node);
}

// ------------------------------------------------------------------------------------------------
// Batch stream construction (heterogeneous_blur.c:439-442: memcpy of the source image into every slot of the
// batch buffer, inside the timed region).  At MI355X speeds this single-threaded memcpy is the slowest stage of
// the whole pipeline (6.9 MB per 35-image batch at ~14 GB/s = 0.5 ms against ~0.25 ms of PCIe time), so the
// hosts spread it over a few persistent helper threads.
// ------------------------------------------------------------------------------------------------
// One image into one slot of a batch buffer.  The buffer is written once and next read by the GPU over the host link, never
// by this core, so the stores bypass the cache (no read-for-ownership of 6.9 MB per batch that the link's own DRAM traffic
// competes with); glibc's memcpy keeps cached stores for copies this small.  MI_BLUR_HOST_NT_COPY=0 switches back.
inline bool host_nt_copy()
{
    static const bool on = [] { const char *e = getenv("MI_BLUR_HOST_NT_COPY"); return !(e && e[0] == '0'); }();
    return on;
}
inline void copy_to_batch(uint8_t *dst, const uint8_t *src, size_t n)
{
#if defined(__SSE2__)
    if (host_nt_copy() && n >= 4096) {
        const size_t head = (64 - ((uintptr_t)dst & 63)) & 63;
        memcpy(dst, src, head); dst += head; src += head; n -= head;
        const size_t blocks = n / 64;
        for (size_t i = 0; i < blocks; i++, dst += 64, src += 64) {
            const __m128i a = _mm_loadu_si128((const __m128i *)src), b = _mm_loadu_si128((const __m128i *)(src + 16));
            const __m128i c = _mm_loadu_si128((const __m128i *)(src + 32)), d = _mm_loadu_si128((const __m128i *)(src + 48));
            _mm_stream_si128((__m128i *)dst, a); _mm_stream_si128((__m128i *)(dst + 16), b);
            _mm_stream_si128((__m128i *)(dst + 32), c); _mm_stream_si128((__m128i *)(dst + 48), d);
        }
        memcpy(dst, src, n - blocks * 64);
        _mm_sfence();
        return;
    }
#endif
    memcpy(dst, src, n);
}

class Replicator {
public:
    // device >= 0: the helper threads keep to the CPUs of that GPU's socket (mi_blur_bind_thread_to_device) — they write
    // the batch buffers that GPU reads
    explicit Replicator(int n_threads, int device = -1) : n_(n_threads < 1 ? 1 : n_threads)
    {
        for (int i = 1; i < n_; i++) workers_.emplace_back([this, i, device] { if (device >= 0) mi_blur_bind_thread_to_device(device); loop(i); });
    }
    ~Replicator()
    {
        { std::lock_guard<std::mutex> lk(m_); stop_ = true; gen_++; }
        cv_.notify_all();
        for (auto &t : workers_) t.join();
    }
    // dst[i*image_size .. ) = src for i in [0, count).  for_device: the batch is next read by a GPU (not by CPU threads)
    void run(uint8_t *dst, const uint8_t *src, size_t image_size, int count, bool for_device = false)
    {
        if (n_ == 1 || count < 2 * n_) {
            for (int i = 0; i < count; i++) copy_one(dst + (size_t)i * image_size, src, image_size, for_device);
            return;
        }
        { std::lock_guard<std::mutex> lk(m_); dst_ = dst; src_ = src; size_ = image_size; count_ = count; stream_ = for_device; pending_ = n_ - 1; gen_++; }
        cv_.notify_all();
        part(0);
        std::unique_lock<std::mutex> lk(m_);
        done_.wait(lk, [this] { return pending_ == 0; });
    }

private:
    static void copy_one(uint8_t *dst, const uint8_t *src, size_t n, bool for_device)
    {
        if (for_device) copy_to_batch(dst, src, n); else memcpy(dst, src, n);
    }
    void part(int w) const
    {
        const int b = (int)((long long)count_ * w / n_), e = (int)((long long)count_ * (w + 1) / n_);
        for (int i = b; i < e; i++) copy_one(dst_ + (size_t)i * size_, src_, size_, stream_);
    }
    void loop(int w)
    {
        unsigned long long seen = 0;
        for (;;) {
            { std::unique_lock<std::mutex> lk(m_); cv_.wait(lk, [&] { return gen_ != seen; }); seen = gen_; if (stop_) return; }
            part(w);
            { std::lock_guard<std::mutex> lk(m_); if (--pending_ == 0) done_.notify_one(); }
        }
    }
    const int n_;
    std::vector<std::thread> workers_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    unsigned long long gen_ = 0;
    bool stop_ = false;
    uint8_t *dst_ = nullptr; const uint8_t *src_ = nullptr; size_t size_ = 0; int count_ = 0, pending_ = 0;
    bool stream_ = false;
};

// ------------------------------------------------------------------------------------------------
// Frame source: a stream of DISTINCT frames instead of one image copied N times.  The reference's loader,
// CImg, stores a frame PLANAR and the reference interleaves it on one host core (heterogeneous_blur.c:106-135); here the
// decoder threads only produce the planar frame — in a pinned batch buffer — and the interleave is a GPU kernel inside
// mi_blur_submit_planar.  PPM/PGM always; anything CImg reads when built with -DMI_BLUR_WITH_CIMG.
// ------------------------------------------------------------------------------------------------
inline bool is_dir(const std::string &p) { struct stat st; return stat(p.c_str(), &st) == 0 && S_ISDIR(st.st_mode); }
inline bool is_file(const std::string &p) { struct stat st; return stat(p.c_str(), &st) == 0 && S_ISREG(st.st_mode); }

inline bool frame_extension(const std::string &name)
{
    const size_t dot = name.rfind('.');
    if (dot == std::string::npos) return false;
    std::string e = name.substr(dot + 1);
    for (auto &ch : e) ch = (char)tolower(ch);
    if (e == "ppm" || e == "pgm") return true;
#ifdef MI_BLUR_WITH_CIMG
    if (e == "jpg" || e == "jpeg" || e == "bmp" || e == "png") return true;
#endif
    return false;
}

// DIR -> its frame files in name order; "f_%04d.ppm" -> f_0000.ppm, f_0001.ppm, ... while they exist; FILE -> that file
inline std::vector<std::string> list_frames(const std::string &arg)
{
    std::vector<std::string> files;
    if (is_dir(arg)) {
        if (DIR *d = opendir(arg.c_str())) {
            while (struct dirent *e = readdir(d))
                if (frame_extension(e->d_name) && is_file(arg + "/" + e->d_name)) files.push_back(arg + "/" + e->d_name);
            closedir(d);
        }
        std::sort(files.begin(), files.end());
    } else if (arg.find('%') != std::string::npos) {
        for (int i = 0;; i++) {
            char path[4096];
            snprintf(path, sizeof path, arg.c_str(), i);
            if (!is_file(path)) break;
            files.push_back(path);
        }
    } else if (is_file(arg)) {
        files.push_back(arg);
    }
    return files;
}

// Header only (the first frame fixes the stream's geometry).
inline bool probe_frame(const std::string &path, int &w, int &h, int &c)
{
#ifdef MI_BLUR_WITH_CIMG
    try { cimg_library::CImg<unsigned char> ci(path.c_str()); w = ci.width(); h = ci.height(); c = ci.spectrum(); return true; } catch (...) {}
#endif
    Image img;
    if (!load_pnm(path.c_str(), img)) return false;
    w = img.width; h = img.height; c = img.channels;
    return true;
}

// One frame into `planar` (c*w*h bytes, CImg order).  `scratch` is the calling thread's own buffer.
inline bool decode_frame_planar(const std::string &path, uint8_t *planar, int w, int h, int c, std::vector<uint8_t> &scratch)
{
#ifdef MI_BLUR_WITH_CIMG
    try {
        cimg_library::CImg<unsigned char> ci(path.c_str());
        if (ci.width() != w || ci.height() != h || ci.spectrum() != c) return false;
        memcpy(planar, ci.data(), (size_t)w * h * c);           // CImg storage IS planar
        return true;
    } catch (...) {}
#endif
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    char magic[3] = {0, 0, 0};
    int fw = 0, fh = 0, maxv = 0;
    bool ok = fread(magic, 1, 2, f) == 2 && magic[0] == 'P' && (magic[1] == '6' || magic[1] == '5') && skip_ws_comments(f) &&
              fscanf(f, "%d", &fw) == 1 && skip_ws_comments(f) && fscanf(f, "%d", &fh) == 1 && skip_ws_comments(f) &&
              fscanf(f, "%d", &maxv) == 1 && maxv == 255 && fw == w && fh == h && (magic[1] == '6' ? 3 : 1) == c;
    if (ok) {
        fgetc(f);
        const size_t plane = (size_t)w * h;
        if (c == 1) ok = fread(planar, 1, plane, f) == plane;                  // one plane: already "planar"
        else {
            scratch.resize(plane * c);
            ok = fread(scratch.data(), 1, scratch.size(), f) == scratch.size();
            if (ok)
                for (int k = 0; k < c; k++) {                                    // what CImg's PNM reader does: scatter into planes
                    uint8_t *dst = planar + (size_t)k * plane;
                    const uint8_t *src = scratch.data() + k;
                    for (size_t px = 0; px < plane; px++) dst[px] = src[px * c];
                }
        }
    }
    fclose(f);
    return ok;
}

// A PPM/PGM frame as it lies on disk — interleaved — straight into `dst` (w*h*c bytes of the pinned batch buffer).
inline bool read_pnm_interleaved(const std::string &path, uint8_t *dst, int w, int h, int c)
{
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    char magic[3] = {0, 0, 0};
    int fw = 0, fh = 0, maxv = 0;
    bool ok = fread(magic, 1, 2, f) == 2 && magic[0] == 'P' && (magic[1] == '6' || magic[1] == '5') && skip_ws_comments(f) &&
              fscanf(f, "%d", &fw) == 1 && skip_ws_comments(f) && fscanf(f, "%d", &fh) == 1 && skip_ws_comments(f) &&
              fscanf(f, "%d", &maxv) == 1 && maxv == 255 && fw == w && fh == h && (magic[1] == '6' ? 3 : 1) == c;
    if (ok) { fgetc(f); const size_t n = (size_t)w * h * c; ok = fread(dst, 1, n, f) == n; }
    fclose(f);
    return ok;
}

// Save one output frame; `planar` says how `px` is laid out (split_image_blur.c:40-56 builds the planar form to save).
inline bool save_frame(const std::string &path, const uint8_t *px, int w, int h, int c, bool planar, std::vector<uint8_t> &scratch)
{
#ifdef MI_BLUR_WITH_CIMG
    if (planar) { cimg_library::CImg<unsigned char> out(px, w, h, 1, c); out.save(path.c_str()); return true; }
#endif
    if (!planar || c == 1) return save_pnm(path.c_str(), px, w, h, c);
    const size_t plane = (size_t)w * h;
    scratch.resize(plane * c);
    for (int k = 0; k < c; k++)
        for (size_t i = 0; i < plane; i++) scratch[i * c + k] = px[(size_t)k * plane + i];
    return save_pnm(path.c_str(), scratch.data(), w, h, c);
}

// A few persistent helper threads running fn(item, thread) over items [0, n): frame decode / save tasks.
class TaskPool {
public:
    explicit TaskPool(int n_threads, int device = -1) : n_(n_threads < 1 ? 1 : n_threads)
    {
        for (int i = 1; i < n_; i++) workers_.emplace_back([this, i, device] { if (device >= 0) mi_blur_bind_thread_to_device(device); loop(i); });
    }
    ~TaskPool()
    {
        { std::lock_guard<std::mutex> lk(m_); stop_ = true; gen_++; }
        cv_.notify_all();
        for (auto &t : workers_) t.join();
    }
    int threads() const { return n_; }
    void run(int n_items, const std::function<void(int, int)> &fn)
    {
        if (n_items <= 0) return;
        { std::lock_guard<std::mutex> lk(m_); fn_ = &fn; items_ = n_items; next_.store(0); pending_ = n_ - 1; gen_++; }
        cv_.notify_all();
        work(0);
        std::unique_lock<std::mutex> lk(m_);
        done_.wait(lk, [this] { return pending_ == 0; });
    }

private:
    void work(int t) { for (int i; (i = next_.fetch_add(1)) < items_;) (*fn_)(i, t); }
    void loop(int t)
    {
        unsigned long long seen = 0;
        for (;;) {
            { std::unique_lock<std::mutex> lk(m_); cv_.wait(lk, [&] { return gen_ != seen; }); seen = gen_; if (stop_) return; }
            work(t);
            { std::lock_guard<std::mutex> lk(m_); if (--pending_ == 0) done_.notify_one(); }
        }
    }
    const int n_;
    std::vector<std::thread> workers_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    unsigned long long gen_ = 0;
    bool stop_ = false;
    const std::function<void(int, int)> *fn_ = nullptr;
    int items_ = 0, pending_ = 0;
    std::atomic<int> next_{0};
};

// ------------------------------------------------------------------------------------------------
// Report blocks.  Wording and number formats follow the reference report (heterogeneous_blur.c:609-724,
// split_image_blur.c:615-721) because scripts parse these lines; the layout of the code does not.
// ------------------------------------------------------------------------------------------------
struct DeviceTimes {                       // the reference's three buckets per device (:411-412)
    double in_ms = 0, kernel_ms = 0, out_ms = 0;
    double total() const { return in_ms + kernel_ms + out_ms; }
    void add(const mi_blur_timing &t) { in_ms += t.h2d_ms; kernel_ms += t.kernel_ms; out_ms += t.d2h_ms; }
};

inline void report_header(int batch_size, double wall_ms, int images)
{
    printf("========== PERFORMANCE RESULTS ==========\n\n");
    printf("BATCH SIZE : %d\n", batch_size);
    printf("1. OVERALL EXECUTION TIME\n");
    printf("   Total wall-clock time: %.2f ms (%.2f seconds)\n", wall_ms, wall_ms / 1000.0);
    printf("   Total images processed: %d\n\n", images);
}

// "2. CPU DEVICE (...)" / "3. GPU DEVICE (...)": `what` is the text inside the parentheses.
inline void report_device(int section, const char *dev, const char *what, const DeviceTimes &t, long long per_image_over)
{
    static const char *const label[3] = {"Transfer IN:     ", "Kernel execution:", "Transfer OUT:    "};
    const double part[3] = {t.in_ms, t.kernel_ms, t.out_ms};
    printf("%d. %s DEVICE (%s)\n", section, dev, what);
    printf("   Total %s time:        %.2f ms\n", dev, t.total());
    for (int i = 0; i < 3; i++) printf("   - %s    %.2f ms (%.1f%%)\n", label[i], part[i], part[i] / t.total() * 100);
    if (per_image_over > 0) printf("   Average per image:     %.3f ms\n", t.total() / per_image_over);
}

inline void report_bottleneck_line(const char *dev, const DeviceTimes &t)
{
    printf("   %s bottleneck: ", dev);
    if (t.in_ms + t.out_ms > t.kernel_ms) printf("COMMUNICATION (%.1f%% of time)\n", (t.in_ms + t.out_ms) / t.total() * 100);
    else printf("COMPUTATION (%.1f%% of time)\n", t.kernel_ms / t.total() * 100);
}

struct Comparison { double speedup = 0, imbalance = 0; };

// Sections 4-6 (device comparison, workload balance, bottleneck identification).
inline Comparison report_comparison(const DeviceTimes &cpu, const DeviceTimes &gpu)
{
    Comparison c;
    const double tc = cpu.total(), tg = gpu.total();
    printf("4. DEVICE COMPARISON\n");
    c.speedup = tc / tg;
    if (c.speedup > 1.0) printf("   GPU is %.2fx FASTER than CPU\n", c.speedup);
    else printf("   CPU is %.2fx FASTER than GPU\n", 1.0 / c.speedup);
    printf("   CPU/GPU time ratio: %.2f\n\n", c.speedup);
    printf("5. WORKLOAD BALANCE\n");
    c.imbalance = fabs(tc - tg) / fmax(tc, tg) * 100.0;
    printf("   Workload imbalance: %.1f%%\n", c.imbalance);
    printf("   %s is the BOTTLENECK (%.2f ms slower)\n\n", tc > tg ? "CPU" : "GPU", fabs(tc - tg));
    printf("6. BOTTLENECK IDENTIFICATION\n");
    report_bottleneck_line("CPU", cpu);
    report_bottleneck_line("GPU", gpu);
    return c;
}

struct Throughput { double mpix = 0, img_s = 0; };
inline Throughput report_throughput(int images, int width, int height, double wall_ms)
{
    Throughput t;
    t.mpix = ((double)images * width * height) / (wall_ms / 1000.0) / 1000000.0;
    t.img_s = images / (wall_ms / 1000.0);
    printf("7. THROUGHPUT\n");
    printf("   Overall throughput: %.2f Megapixels/sec\n", t.mpix);
    printf("   Images per second: %.2f\n\n", t.img_s);
    printf("=========================================\n\n");
    return t;
}

// MI355X addendum: kernel-only rate against the HBM roofline (kernel_ms is summed over G concurrent GPUs).
struct Roofline { double gbps = 0, frac = 0; };
// `streams` > 1: dispatches of one GPU overlap, so the bucket is a sum of overlapping durations — the figure is per
// dispatch (what rocprofv3 reports per kernel), and the stream's rate is the wall-clock line of section 7.
inline Roofline report_roofline(int section, int G, uint64_t bytes_alg, uint64_t launches, double kernel_ms_sum, long long images,
                                int streams = 1)
{
    Roofline r;
    if (kernel_ms_sum <= 0 || launches == 0) return r;
    r.gbps = (double)bytes_alg / (kernel_ms_sum / 1000.0) / 1e9 * G;
    r.frac = r.gbps / (HBM_PEAK_GBS * G);
    printf("%d. MI355X KERNEL ROOFLINE (%d GPU%s)\n", section, G, G > 1 ? "s" : "");
    printf("   Launches: %llu (one per batch per GPU; one per GPU with --fused), avg %.2f us\n", (unsigned long long)launches, kernel_ms_sum * 1000.0 / launches);
    printf("   Algorithmic bytes (2*W*H*C per image): %.2f MB\n", bytes_alg / 1e6);
    printf("   %s: %.0f images/sec, %.1f GB/s = %.1f%% of %.0f GB/s HBM peak\n",
           streams > 1 ? "Per-dispatch rate" : "Kernel-only rate", images / (kernel_ms_sum / 1000.0) * G, r.gbps, r.frac * 100,
           HBM_PEAK_GBS * G);
    if (streams > 1)
        printf("   (%d dispatches in flight per GPU: durations overlap, the stream's own rate is the wall-clock figure above)\n", streams);
    return r;
}

// One row in the column order of the reference's data/approach2/approach2/per_run.csv (+ 3 MI355X columns).
inline void append_csv(const std::string &path, int batch, const char *mode, float gpu_ratio, int images, int batches, int width,
                       int height, double wall_ms, long long cpu_images, const DeviceTimes &cpu, long long gpu_images,
                       const DeviceTimes &gpu, const Comparison &cmp, const Throughput &thr, double recommended, const Roofline &rf, int G)
{
    FILE *f = fopen(path.c_str(), "a");
    if (!f) { printf("Warning: cannot append to %s\n", path.c_str()); return; }
    if (ftell(f) == 0)
        fprintf(f, "batch_size_file,run,file,mode,gpu_ratio_cfg,cpu_ratio_cfg,images,batches,img_w,img_h,wg_w,wg_h,wall_ms,"
                   "cpu_images,cpu_total_ms,cpu_in_ms,cpu_kernel_ms,cpu_out_ms,cpu_ms_per_img,gpu_images,gpu_total_ms,gpu_in_ms,"
                   "gpu_kernel_ms,gpu_out_ms,gpu_ms_per_img,speedup_gpu_vs_cpu,imbalance_pct,bottleneck,bottleneck_delta_ms,"
                   "mpix_per_sec,img_per_sec,recommended_gpu_ratio,batch_size_log,hbm_gbps,roofline_frac,n_gpus\n");
    const double tc = cpu.total(), tg = gpu.total();
    fprintf(f, "%d,1,,%s,%.3f,%.3f,%d,%d,%d,%d,16,16,%.2f,%lld,%.2f,%.2f,%.2f,%.2f,%.4f,%lld,%.2f,%.2f,%.2f,%.2f,%.4f,%.2f,%.1f,%s,%.2f,%.2f,%.2f,%.3f,%d,%.1f,%.4f,%d\n",
            batch, mode, gpu_ratio, 1 - gpu_ratio, images, batches, width, height, wall_ms,
            cpu_images, tc, cpu.in_ms, cpu.kernel_ms, cpu.out_ms, cpu_images ? tc / cpu_images : 0.0,
            gpu_images, tg, gpu.in_ms, gpu.kernel_ms, gpu.out_ms, gpu_images ? tg / gpu_images : 0.0,
            cmp.speedup, cmp.imbalance, tc > tg ? "CPU" : "GPU", fabs(tc - tg), thr.mpix, thr.img_s, recommended, batch, rf.gbps, rf.frac, G);
    fclose(f);
}

}  // namespace host
