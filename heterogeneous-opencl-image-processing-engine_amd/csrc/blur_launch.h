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
    const uint8_t *halo_top, *halo_bottom;   // one band only: rows [0, y0) / [y1, band_rows) are read from here (another shard,
                            // possibly another GPU's memory) instead of from `in`; nullptr = from `in`.  Direct kernel only.
    hipStream_t stream;
    hipEvent_t start, stop; // optional: dispatch start/stop timestamps (hipExtLaunchKernel)
};

// Returns MI_BLUR_OK or a negative mi_blur_status.
int launch(const LaunchDesc &d);

// Fused stream (blur_fused_kernel): one dispatch over d.n_images images whose blocks are ordered in batches of
// batch_images; every block of batch b adds *waves_per_block (1) to one of count[kb .. kb+k-1], k = *counters_per_batch
// (device, zeroed by the caller) after its stores have drained, so batch b is complete when those k sum to its blocks (geometry outputs:
// a full batch has *tiles_per_batch blocks, the last one what is left of *total_blocks).
// Shapes of the tiled kernel, aligned or ragged, 1-4 channels (MI_BLUR_ERR_UNSUPPORTED otherwise; ragged rows need f.tail_ctr).
// geometry_only: fill the geometry outputs for these knobs and return without launching (the caller decides from them
// whether its counters can keep counting up).  tun: the knob set to use (nullptr = the current one); a caller that asks
// for the geometry first passes the same copy to both calls.
struct Tunables;
struct FusedDesc {
    unsigned *count; int batch_images; unsigned *tiles_per_batch, *waves_per_block, *total_blocks;
    const Tunables *tun; bool geometry_only;
    unsigned *tail_ctr;     // device word, zero between passes: ticket counter of the pass's dynamic tail ("fused_tail"); nullptr = none
    long long count_words;  // how many words `count` holds (0 = at least 8 per batch, the minimum)
    unsigned *counters_per_batch;   // geometry output: counters each batch uses (8 .. 256; count[k * b .. k * b + k - 1] belong to batch b)
};
int launch_fused(const LaunchDesc &d, const FusedDesc &f);
// Watcher of one fused pass (fused_watch_kernel, one wave on a stream of its own): follows the per-batch counters and keeps
// *host_word (pinned host memory) = (pass_seq << 32) | leading batches complete, so the host reads a batch's completion
// from its own memory instead of copying the counters back.  Ends when all n_batches are complete (or after a hard limit).
int launch_fused_watch(const unsigned *count, unsigned n_batches, unsigned tiles_per_batch, unsigned total_blocks, unsigned per_block,
                       unsigned long long *host_word, unsigned pass_seq, hipStream_t stream, unsigned counters_per_batch);

// Planar (CImg storage: plane c of image i at (i*C + c)*W*H) <-> interleaved repack, layout_kernels.hip.
int launch_planar_to_interleaved(const uint8_t *src, uint8_t *dst, int width, int height, int channels, int n_images, hipStream_t s);
int launch_interleaved_to_planar(const uint8_t *src, uint8_t *dst, int width, int height, int channels, int n_images, hipStream_t s);
// Halo pull: copy `bytes` bytes from each non-null (peer) source into the matching halo destination, one launch.
int launch_halo_pull(const uint8_t *top_src, uint8_t *top_dst, const uint8_t *bottom_src, uint8_t *bottom_dst, size_t bytes, hipStream_t stream);

// ----------------------------------------------------------------------------------------------------------------
// Zero-copy batch server (blur_server_kernel): ONE long-lived dispatch takes batch after batch of pinned host frames from a
// ring of descriptors the host fills, so consecutive batches of a host-fed stream flow through the same workgroups with
// no dispatch boundary between them (a per-batch launch costs the host link ~45 us of a ~200 us batch, profiles/r03_e2e_timeline.md).
//   host -> device: ZcHostCtl::batch[k % ZC_RING] (the tile parameters of batch k), then tail = k + 1
//   device -> host: done[k % ZC_RING] = k + 1 once every output byte of batch k is in host memory
// The grid is n_workers + 1 workgroups: the last one is the POLLER (reads `tail` over the host link, republishes it in
// device memory, decides when the server ends), the others walk the tiles.  A server ALWAYS ends: after `budget` batches,
// after idle_ticks (100 MHz) without a new batch, or when the host sets `quit`; its workers leave once the poller has said
// where the server stops.  The host keeps the next server queued behind the running one on the same stream.
// ----------------------------------------------------------------------------------------------------------------
constexpr unsigned ZC_RING = 64;
constexpr unsigned ZC_PARAM_WORDS = 32;
constexpr unsigned ZC_TRACE_BATCHES = 512;                    // diagnostics: the trace keeps the stamps of a context's first this-many batches
constexpr unsigned long long ZC_HARD_TICKS = 1000000000ull;   // 10 s of the 100 MHz device clock: no wait inside a server outlasts this
struct ZcBatch {
    unsigned params[ZC_PARAM_WORDS];   // TiledParams of the batch, as words
    unsigned tile_first;               // global number of the batch's first tile (tiles are numbered through the batches, mod 2^32)
    unsigned n_tiles;
    unsigned pad[6];
};
struct ZcHostCtl {                     // pinned, device-visible host memory
    unsigned tail;                     // host -> device: descriptors published so far
    unsigned quit;                     // host -> device: leave at the next poll
    unsigned servers_done;             // device -> host: servers that have signalled every batch they took and left
    unsigned error;                    // device -> host: a server's wait ran into its hard limit (ZC_HARD_TICKS) — should never happen
    unsigned pad0[12];
    unsigned done[ZC_RING];            // device -> host: done[k % ZC_RING] = k + 1
    unsigned long long t_begin[ZC_RING], t_end[ZC_RING];   // device clock (100 MHz ticks): batch k taken up / complete
    ZcBatch batch[ZC_RING];
};
struct ZcDevCtl {                      // device memory, zeroed when the server state is created
    unsigned next[2];                  // server `seq` starts at batch next[seq & 1] and leaves next[(seq + 1) & 1]
    unsigned gnext[2];                 // ... and at global tile gnext[seq & 1]
    unsigned start_seq;                // poller -> workers: server (start_seq - 1) has set the ticket counter
    unsigned avail;                    // poller -> workers: batches [.., avail) are published (descriptors in `batch`)
    unsigned stop_seq, stop_at, stop_tile;   // poller -> workers: server (stop_seq - 1) ends before batch stop_at / tile stop_tile
    unsigned ticket;                   // next global tile to hand out
    unsigned pad[6];
    unsigned tiles_done[ZC_RING];      // tiles of the batch in this ring slot whose output is in host memory
    ZcBatch batch[ZC_RING];            // device copies of the descriptors
};
struct ZcGeometry { unsigned threads; size_t lds; int rpg; int channels, radius, ragged; };
// Tile parameters of one batch (d.in/d.out = DEVICE addresses of the pinned buffers) in the server's geometry; *geo is
// filled on the first call (geo->threads == 0) and must match on later ones (MI_BLUR_ERR_UNSUPPORTED otherwise).
int zc_fill_batch(const LaunchDesc &d, ZcGeometry *geo, ZcBatch *b, unsigned *n_tiles);
int zc_launch_server(const ZcGeometry &geo, ZcHostCtl *ctl, ZcDevCtl *dev, unsigned seq, unsigned n_workers, unsigned budget,
                     unsigned idle_ticks, hipStream_t stream, unsigned long long *trace = nullptr, int fixed_share = 0);

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
    int fused_tail;      // fused stream: per mille of a pass's tiles that are handed out dynamically at the end (0 = none)
    int fused_adds_per_word; // fused stream: a batch gets as many completion counters (8 .. 256) as keep the adds per counter and pass under this
    int fused_tail_blocks; // ... by (100 + this) % as many extra workgroups as there are tail tiles (one ticket each; default 100)
    int zero_copy_server;  // zero-copy submits of aligned shapes go through the batch server (one long-lived dispatch per stream of
                           // batches, blur_server_kernel) instead of one launch per batch: 1 (default) | 0
    int zero_copy_server_min_kb; // batch server: submits whose output is smaller than this many KiB take one launch each instead (default 1280)
    int staged_server;     // submits of PAGEABLE caller memory: 1 (default) = the batch server works on the slot's pinned staging buffers in
                           // place, 0 = DMA copy in, launch on device buffers, DMA copy out
    int zero_copy_workers; // batch server: worker workgroups (default 48: 40-64 measured best, profiles/r03_e2e_timeline.md)
    int zero_copy_idle_us; // batch server: leaves after this long without a new batch (default 300)
    int zero_copy_budget;  // batch server: leaves after this many batches, the queued next one carries on (default 256)
    int zero_copy_tickets; // batch server: 1 (default) = tiles by ticket counter, 0 = fixed share per worker (tile g to worker g mod n; A/B only)
    int resident_place_trials; // mi_blur_resident_alloc: candidate placements of a big pool that are timed before one is kept (default 4; 0/1 = none)
    int zero_copy_spin;    // waiting for a batch of the server: 0 (default) = spin ~20 us, then sleep in 20 us steps; 1 = spin + yield only
    int zero_copy_trace;   // diagnostics: the batch server's workers stamp their phases per batch (mi_blur_debug_zc_trace)
    int zero_copy_debug_base; // test hook: a new batch server starts this many batches (and 7x as many tiles) short of 2^32, so that the
                         // wrap of its batch and tile numbers — 80 minutes into a continuous batch-35 stream — happens at once (0 = off)
    int zero_copy_events; // zero-copy submits: 1 = the dispatch carries start/stop timestamp events (kernel bucket + completion),
                         // 0 = plain launch, completion by stream synchronise (timing experiment: no kernel bucket)
};
Tunables tunables();
unsigned long long *debug_xcd_buffer();
unsigned debug_xcd_slots();
void set_tunables(const Tunables &t);

}  // namespace mi_blur
