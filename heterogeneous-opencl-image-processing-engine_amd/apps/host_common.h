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

#include <sys/time.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

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
    int gpus = 1;                        // --gpus G
    int slots = 3;                       // --slots S    staging slots / batch buffers in flight
    int threads = 0;                     // --threads T  CPU device threads (0 = all cores)
    bool verbose = false;                // --verbose    per-batch progress lines (heterogeneous_blur.c:420,463,599)
    bool resident = false;               // --resident   device-resident stream (kernel-only)
    std::string csv;                     // --csv FILE   append one per_run.csv-style row
    std::string save;                    // --save FILE  write the first output image
    int iters = 100;                     // --iters N    (split_image_blur --resident)
    bool size_given = false;
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
        else if (a == "--images") { o.images = atoi(next("--images")); if (o.images < 1) { printf("Error: --images must be >= 1\n"); exit(-1); } }
        else if (a == "--gpus") o.gpus = atoi(next("--gpus"));
        else if (a == "--slots") { o.slots = atoi(next("--slots")); if (o.slots < 1) o.slots = 1; }
        else if (a == "--threads") o.threads = atoi(next("--threads"));
        else if (a == "--verbose") o.verbose = true;
        else if (a == "--resident") o.resident = true;
        else if (a == "--csv") o.csv = next("--csv");
        else if (a == "--save") o.save = next("--save");
        else if (a == "--iters") o.iters = atoi(next("--iters"));
        else { printf("Error: unknown option %s\n", a.c_str()); exit(-1); }
    }
    return npos;
}

constexpr double HBM_PEAK_GBS = 8000.0;   // MI355X HBM3E spec; ~6290 GB/s measured copy ceiling

}  // namespace host
