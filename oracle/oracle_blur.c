/*
 * oracle_blur.c — CPU restatement of the reference blur path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this.  The product library
 * (libmi_blur.so) never links, loads or calls anything in oracle/.
 *
 * Parity status: PINNED for the 3x3 kernel — checked byte-for-byte against the
 * unmodified reference kernel (oracle/_ref/libref_blur.so, built by
 * oracle/Makefile from /root/reference/gaussian_kernel.cl where it lies) and
 * against tests/golden/blur_golden.json generated from that build.
 * The 5x5 kernel does not exist in the reference (only the 3x3 table at
 * gaussian_kernel.cl:36-41, HALO = 1 at split_image_blur.c:70): its oracle is
 * "parity unpinned" — it only obeys the 3x3 kernel's conventions.
 *
 * All citations are relative to /root/reference/.
 */
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* ------------------------------------------------------------------------
 * K3 — float restatement, one "work-item" per pixel.
 * Follows gaussian_kernel.cl:19-72 statement by statement:
 *   :36-41 weight table, :44 channel loop, :48-49 ky/kx loops,
 *   :56-57 clamp-to-edge, :60 interleaved index, :63 float accumulate,
 *   :70 truncating cast.
 * ---------------------------------------------------------------------- */
void oracle_blur3_f32(const uint8_t *in, uint8_t *out, int width, int height, int channels)
{
    static const float w[3][3] = {
        {0.0625f, 0.125f, 0.0625f},
        {0.125f,  0.25f,  0.125f},
        {0.0625f, 0.125f, 0.0625f}};
    for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++)
            for (int c = 0; c < channels; c++) {
                float sum = 0.0f;
                for (int ky = -1; ky <= 1; ky++)
                    for (int kx = -1; kx <= 1; kx++) {
                        int nx = clampi(x + kx, 0, width - 1);
                        int ny = clampi(y + ky, 0, height - 1);
                        size_t idx = ((size_t)ny * width + nx) * channels + c;
                        sum += in[idx] * w[ky + 1][kx + 1];
                    }
                out[((size_t)y * width + x) * channels + c] = (unsigned char)sum;
            }
}

/* ------------------------------------------------------------------------
 * K3 / K5 — integer form.  Every weight is dyadic, every partial sum is a
 * multiple of 2^-4 (2^-8) below 2^24, so the float accumulate above is exact
 * and equals (sum of integer weights * pixels) >> 4 (>> 8); truncation
 * (gaussian_kernel.cl:70) becomes the right shift.
 *   radius 1: taps [1 2 1] (x) [1 2 1], shift 4   (gaussian_kernel.cl:36-41 * 16)
 *   radius 2: taps [1 4 6 4 1] (x) [1 4 6 4 1], shift 8   (build-defined, unpinned)
 * ---------------------------------------------------------------------- */
static const int TAPS1[3] = {1, 2, 1};
static const int TAPS2[5] = {1, 4, 6, 4, 1};

int oracle_blur_int(const uint8_t *in, uint8_t *out, int width, int height, int channels, int radius)
{
    if (radius != 1 && radius != 2) return -1;
    const int *t = radius == 1 ? TAPS1 : TAPS2;
    const int shift = radius == 1 ? 4 : 8;
    for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++)
            for (int c = 0; c < channels; c++) {
                unsigned sum = 0;
                for (int ky = -radius; ky <= radius; ky++) {
                    int ny = clampi(y + ky, 0, height - 1);
                    for (int kx = -radius; kx <= radius; kx++) {
                        int nx = clampi(x + kx, 0, width - 1);
                        sum += (unsigned)in[((size_t)ny * width + nx) * channels + c] *
                               (unsigned)(t[ky + radius] * t[kx + radius]);
                    }
                }
                out[((size_t)y * width + x) * channels + c] = (uint8_t)(sum >> shift);
            }
    return 0;
}

/* K5 in float, written the way the reference writes K3 (weights b(x)b/256 as
 * float literals, float accumulate, truncating cast) — used by the tests to
 * show that the integer 5x5 obeys the same conventions. */
void oracle_blur5_f32(const uint8_t *in, uint8_t *out, int width, int height, int channels)
{
    float w[5][5];
    for (int i = 0; i < 5; i++)
        for (int j = 0; j < 5; j++) w[i][j] = (float)(TAPS2[i] * TAPS2[j]) / 256.0f;
    for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++)
            for (int c = 0; c < channels; c++) {
                float sum = 0.0f;
                for (int ky = -2; ky <= 2; ky++)
                    for (int kx = -2; kx <= 2; kx++) {
                        int nx = clampi(x + kx, 0, width - 1);
                        int ny = clampi(y + ky, 0, height - 1);
                        sum += in[((size_t)ny * width + nx) * channels + c] * w[ky + 2][kx + 2];
                    }
                out[((size_t)y * width + x) * channels + c] = (unsigned char)sum;
            }
}

/* Batched form: n independent images laid end to end (heterogeneous_blur.c:431-442
 * builds exactly this contiguous stream; :485-486 slices it per image). */
int oracle_blur_batch(const uint8_t *in, uint8_t *out, int width, int height, int channels,
                      int radius, int n_images)
{
    size_t isz = (size_t)width * height * channels;
    for (int i = 0; i < n_images; i++) {
        int rc = oracle_blur_int(in + i * isz, out + i * isz, width, height, channels, radius);
        if (rc) return rc;
    }
    return 0;
}

/* ------------------------------------------------------------------------
 * D1 — Approach-1 partition of one batch (heterogeneous_blur.c:449-458,496):
 * the first n_cpu images go to the CPU device, the rest to the GPU device.
 * mode: 0 both, 1 cpu, 2 gpu (heterogeneous_blur.c:51).  ratio is a float
 * in the reference (:48) and the product is evaluated in float (:450).
 * ---------------------------------------------------------------------- */
void oracle_a1_partition(int mode, int batch_count, float gpu_ratio, int *n_cpu, int *n_gpu)
{
    if (mode == 0) {
        *n_gpu = (int)(batch_count * gpu_ratio);
        *n_cpu = batch_count - *n_gpu;
    } else if (mode == 1) {
        *n_cpu = batch_count; *n_gpu = 0;
    } else {
        *n_cpu = 0; *n_gpu = batch_count;
    }
}

/* ------------------------------------------------------------------------
 * D2 — Approach-2 geometry (split_image_blur.c:144-166) with HALO generalised
 * from the constant 1 (:70) to `halo` (= blur radius).
 * ---------------------------------------------------------------------- */
typedef struct {
    int split_row;
    int cpu_input_rows, cpu_output_rows;
    int gpu_input_rows, gpu_output_rows;
} oracle_a2_geom;

void oracle_a2_geometry(int height, float gpu_ratio, int halo, oracle_a2_geom *g)
{
    int split_row = (int)(height * (1.0f - gpu_ratio));      /* :144 */
    if (split_row < halo) split_row = halo;                  /* :147-150 */
    if (split_row > height - halo) split_row = height - halo;/* :151-154 */
    g->split_row = split_row;
    g->cpu_input_rows = split_row + halo;                    /* :157 */
    g->cpu_output_rows = split_row;                          /* :158 */
    g->gpu_input_rows = (height - split_row) + halo;         /* :163 */
    g->gpu_output_rows = height - split_row;                 /* :164 */
}

/* Approach-2 data path for one image (split_image_blur.c:511-541): each device
 * runs the SAME kernel on its sub-buffer with height = sub-buffer rows incl.
 * halo (:401,414), so clamping happens on the sub-buffer; the top device's
 * read-back takes the first cpu_output_size bytes (:526), the bottom device's
 * skips HALO rows (:537). */
int oracle_a2_split_blur(const uint8_t *in, uint8_t *out, int width, int height, int channels,
                         int split_row, int radius)
{
    const int halo = radius;
    size_t pitch = (size_t)width * channels;
    int top_in = split_row + halo, bot_in = (height - split_row) + halo;
    if (split_row < halo || split_row > height - halo) return -1;
    uint8_t *tmp = (uint8_t *)malloc(pitch * (size_t)(top_in > bot_in ? top_in : bot_in));
    if (!tmp) return -2;
    oracle_blur_int(in, tmp, width, top_in, channels, radius);
    memcpy(out, tmp, pitch * split_row);
    oracle_blur_int(in + (size_t)(split_row - halo) * pitch, tmp, width, bot_in, channels, radius);
    memcpy(out + (size_t)split_row * pitch, tmp + (size_t)halo * pitch, pitch * (height - split_row));
    free(tmp);
    return 0;
}

/* K-way row split (the 8-GPU generalisation of D2): shard g owns
 * rows [H*g/K, H*(g+1)/K); it is handed `radius` halo rows from each existing
 * neighbour; image top/bottom clamp instead (gaussian_kernel.cl:57). */
int oracle_splitk_blur(const uint8_t *in, uint8_t *out, int width, int height, int channels,
                       int K, int radius)
{
    size_t pitch = (size_t)width * channels;
    for (int g = 0; g < K; g++) {
        int r0 = (int)((long long)height * g / K), r1 = (int)((long long)height * (g + 1) / K);
        if (r1 <= r0) continue;
        int ht = r0 - radius < 0 ? r0 : radius;             /* halo rows actually available */
        int hb = r1 + radius > height ? height - r1 : radius;
        int rows = (r1 - r0) + ht + hb;
        uint8_t *tmp = (uint8_t *)malloc(pitch * (size_t)rows);
        if (!tmp) return -2;
        oracle_blur_int(in + (size_t)(r0 - ht) * pitch, tmp, width, rows, channels, radius);
        memcpy(out + (size_t)r0 * pitch, tmp + (size_t)ht * pitch, pitch * (size_t)(r1 - r0));
        free(tmp);
    }
    return 0;
}

/* ------------------------------------------------------------------------
 * Frame layout.  The reference loads frames through CImg (planar storage:
 * byte (x, y, c) at c*W*H + y*W + x, CImg.h `operator()(x,y,z,c)`) and repacks
 * them to the interleaved stream with the loop at heterogeneous_blur.c:125-134
 * (idx = (y*width + x)*channels; out[idx + c] = img(x, y, 0, c)); the inverse
 * loop is split_image_blur.c:40-56.  Restated for any channel count.
 * ---------------------------------------------------------------------- */
void oracle_planar_to_interleaved(const uint8_t *planar, uint8_t *interleaved, int W, int H, int C)
{
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            const size_t idx = ((size_t)y * W + x) * C;
            for (int c = 0; c < C; c++) interleaved[idx + c] = planar[(size_t)c * W * H + (size_t)y * W + x];
        }
}

void oracle_interleaved_to_planar(const uint8_t *interleaved, uint8_t *planar, int W, int H, int C)
{
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            const size_t idx = ((size_t)y * W + x) * C;
            for (int c = 0; c < C; c++) planar[(size_t)c * W * H + (size_t)y * W + x] = interleaved[idx + c];
        }
}

/* ------------------------------------------------------------------------
 * Synthetic stream + hashing: LCG s = s*1664525+1013904223
 * (mod 2^32), byte = s>>24, filled in memory order; FNV-1a-64.
 * ---------------------------------------------------------------------- */
void oracle_lcg_fill(uint8_t *buf, size_t n, uint32_t seed)
{
    uint32_t s = seed;
    for (size_t i = 0; i < n; i++) {
        s = s * 1664525u + 1013904223u;
        buf[i] = (uint8_t)(s >> 24);
    }
}

uint64_t oracle_fnv1a64(const uint8_t *buf, size_t n)
{
    uint64_t h = 0xcbf29ce484222325ull;
    for (size_t i = 0; i < n; i++) { h ^= buf[i]; h *= 0x100000001b3ull; }
    return h;
}
