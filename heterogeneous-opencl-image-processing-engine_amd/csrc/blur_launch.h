// blur_launch.h — internal (not part of the C ABI): launch interface between
// mi_blur_api.cpp and the gfx950 kernels in blur_kernels.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mi_blur {

struct LaunchDesc {
    const uint8_t *in;      // device, n_images bands of band_rows rows, laid end to end
    uint8_t *out;           // device, n_images blocks of (y1-y0) rows
    int width, band_rows, channels, radius;
    int n_images;
    int y0, y1;             // output rows [y0,y1) of each band
    long long in_stride, out_stride;  // bytes between consecutive bands / output blocks; 0 = laid end to end.
                            // Non-dense strides are taken by the tiled kernel only (multiples of 16).
    int concurrent;         // hint: how many streams the caller rotates launches like this one over (0/1 = one after another).
                            // Small 3x3 launches that overlap with their like run faster as LDS tiles, alone as the direct kernel.
    int max_blocks;         // > 0: cap the grid of the aligned tiled kernel; its workgroups then loop over the tiles (zero-copy submits)
    int variant;            // mi_blur_variant
    hipStream_t stream;
    hipEvent_t start, stop; // optional: dispatch start/stop timestamps (hipExtLaunchKernel)
};

// Returns MI_BLUR_OK or a negative mi_blur_status.
int launch(const LaunchDesc &d);

// Fused stream (blur_fused_kernel): one dispatch over d.n_images images whose blocks are ordered in batches of
// batch_images; every block of batch b adds *waves_per_block (1) to one of count[8b .. 8b+7] (device, zeroed by the
// caller) after its stores have drained, so batch b is complete when those eight sum to its blocks (geometry outputs:
// a full batch has *tiles_per_batch blocks, the last one what is left of *total_blocks).
// Aligned tiled shapes only (MI_BLUR_ERR_UNSUPPORTED otherwise).
// geometry_only: fill the geometry outputs for these knobs and return without launching (the caller decides from them
// whether its counters can keep counting up).  tun: the knob set to use (nullptr = the current one); a caller that asks
// for the geometry first passes the same copy to both calls.
struct Tunables;
struct FusedDesc {
    unsigned *count; int batch_images; unsigned *tiles_per_batch, *waves_per_block, *total_blocks;
    const Tunables *tun; bool geometry_only;
};
int launch_fused(const LaunchDesc &d, const FusedDesc &f);

// Planar (CImg storage: plane c of image i at (i*C + c)*W*H) <-> interleaved repack, layout_kernels.hip.
int launch_planar_to_interleaved(const uint8_t *src, uint8_t *dst, int width, int height, int channels, int n_images, hipStream_t s);
int launch_interleaved_to_planar(const uint8_t *src, uint8_t *dst, int width, int height, int channels, int n_images, hipStream_t s);

// Name of the kernel the calling thread's most recent launch() / launch_fused() chose ("" before the first).
const char *last_kernel();

// True when the LDS-tiled vector kernel can take this shape.
bool tiled_eligible(const void *in, const void *out, int width, int channels);

// Tunables: defaults from env (MI_BLUR_STAGE=dma|reg, MI_BLUR_RPG=4|8|16, MI_BLUR_XCD=0|1), changeable at run time
// through mi_blur_set_option.  Process-wide, kept behind a mutex: tunables() returns a coherent COPY and every launch
// works from the one copy it took when it started, so flipping a knob while another thread launches is not a data race
// (that launch sees the old set or the new one, never a mix).
struct Tunables {
    int stage_dma; int rpg; int xcd_remap; int debug_copy; int row_shuffle; int prefer_stream; int stream_bh; int zero_copy; int ragged;
    int fused_release;   // fused stream: 1 = the per-block completion add is release-ordered at agent scope (architectural; slow)
    int experiment;      // 1 = the tiled kernel's OTHER row-pass form (A/B runs; C = 3 only)
    int xcd_run;         // tiled kernel's blockIdx -> tile map: 0/1 = one contiguous eighth of the launch per XCD (default),
                         // r >= 2 = runs of r tiles dealt to the XCDs in turn
    int zero_copy_streams;   // streams the zero-copy submits of a context alternate over (1 = one in-order stream)
    int zero_copy_blocks;    // zero-copy submits: cap on resident workgroups (0 = no cap: one workgroup per tile)
    int stream_updown;   // streaming variant: 1 (default) = odd bands march upwards, so both readers of a band seam come at the same time
    int prefer_direct;   // AUTO and the direct (register-staged, LDS-free) variant: 0 never, 1 (default) where it measured faster
                         // (every 5x5 launch, 3x3 launches up to 128 MiB of output), 2 whenever the shape is eligible
    int direct_bh;       // direct variant: output rows per lane (8; 4 | 12 | 16 instantiated for C = 3 only, A/B runs)
    int debug_xcd_times; // diagnostics: the tiled kernel's workgroups leave start/end times per XCD (mi_blur_debug_xcd_times)
    int fused_window;    // fused stream: batches per window of its blockIdx -> tile map (8: one whole batch per XCD per window)
    int zero_copy_events; // zero-copy submits: 1 = the dispatch carries start/stop timestamp events (kernel bucket + completion),
                         // 0 = plain launch, completion by stream synchronise (timing experiment: no kernel bucket)
};
Tunables tunables();
unsigned long long *debug_xcd_buffer();
unsigned debug_xcd_slots();
void set_tunables(const Tunables &t);

}  // namespace mi_blur
