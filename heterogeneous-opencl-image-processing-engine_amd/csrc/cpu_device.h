// cpu_device.h — internal: host-thread device and host helpers (see cpu_device.cpp).
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace mi_blur {

int hardware_threads();
void cpu_blur_rows(const uint8_t *in, uint8_t *out, int W, int H, int C, int R, int y_begin, int y_end,
                   int out_row_shift);
void cpu_blur_batch(const uint8_t *in, uint8_t *out, int W, int band_rows, int C, int R, int n_images,
                    int y0, int y1, int n_threads, size_t in_stride = 0, size_t out_stride = 0);
// planar (CImg storage: all of channel 0, then channel 1, ...) <-> interleaved, n_images frames, a few pool threads
void cpu_repack(const uint8_t *src, uint8_t *dst, int W, int H, int C, int n_images, bool planar_to_interleaved, int n_threads);
void copy_blocks(uint8_t *dst, size_t dst_stride, const uint8_t *src, size_t src_stride, size_t bytes, int n, int n_threads);
void fill_synthetic(uint8_t *host, int W, int H, int C, int first_index, int n_images, int n_threads);
uint64_t fnv1a64(const uint8_t *p, size_t n);

}  // namespace mi_blur
