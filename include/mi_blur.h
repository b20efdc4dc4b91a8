/*
 * mi_blur.h — C ABI of the MI355X-native image-stream blur engine (libmi_blur.so).
 *
 * The reference (CC834/Heterogeneous-OpenCL-Image-Processing-Engine) has no
 * plugin/FFI surface: its hot path sits behind the OpenCL C API called from two
 * main() functions.  Each entry point below replaces one group of those OpenCL
 * calls; the citation is the reference call site it stands in for (paths are
 * relative to the reference tree).  INTEGRATION.md shows the edit a maintainer
 * makes in heterogeneous_blur.c / split_image_blur.c to bind them.
 *
 * Conventions
 *   - plain pointers and sizes only; no C++/torch types; every function returns
 *     MI_BLUR_OK (0) or a negative mi_blur_status / positive passthrough code
 *     (see mi_blur_strerror).  The library never calls exit(): the reference's
 *     cl_error() print-and-exit policy (heterogeneous_blur.c:25-30) stays in the host.
 *   - images are interleaved uint8, row-major, pitch = width*channels, no padding
 *     (gaussian_kernel.cl:60; heterogeneous_blur.c:115,128-135); in != out.
 *   - radius 1 = the reference 3x3 {1,2,1}x{1,2,1}/16 (gaussian_kernel.cl:36-41);
 *     radius 2 = build-defined 5x5 {1,4,6,4,1}x{1,4,6,4,1}/256 (no reference kernel).
 *     Clamp-to-edge (:56-57), truncation (:70).
 *   - a context is single-threaded like a cl_command_queue used from one thread;
 *     different contexts may be driven from different host threads.
 *   - GPU entry points fail with MI_BLUR_ERR_NO_DEVICE when no HIP device is
 *     usable.  There is NO silent CPU fallback: the CPU device exists only when
 *     asked for by name (MI_BLUR_DEVICE_CPU), mirroring the reference's separate
 *     OpenCL CPU device (heterogeneous_blur.c:170-176).
 */
#ifndef MI_BLUR_H
#define MI_BLUR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI_BLUR_VERSION 1
#define MI_BLUR_DEVICE_CPU (-1)      /* native host-thread device (reference: CL_DEVICE_TYPE_CPU) */

typedef enum mi_blur_status {
    MI_BLUR_OK = 0,
    MI_BLUR_ERR_INVALID = -1,        /* bad argument (null, non-positive size, radius not 1|2, in==out) */
    MI_BLUR_ERR_NO_DEVICE = -2,      /* no usable HIP device / device index out of range */
    MI_BLUR_ERR_NOMEM = -3,          /* host or device allocation failed */
    MI_BLUR_ERR_STATE = -4,          /* call not valid in the context's current state */
    MI_BLUR_ERR_UNSUPPORTED = -5,    /* e.g. RCCL entry point in a build without RCCL */
    MI_BLUR_ERR_HIP_BASE = -1000,    /* -(1000 + hipError_t) */
    MI_BLUR_ERR_RCCL_BASE = -2000    /* -(2000 + ncclResult_t) */
} mi_blur_status;

/* Kernel variant selector for mi_blur_enqueue_ex (tests force each path). */
typedef enum mi_blur_variant {
    MI_BLUR_VARIANT_AUTO = 0,        /* pitch%16==0 && channels<=4 && aligned pointers: the direct kernel (5x5; small 3x3 launches) or
                                        the LDS-tiled one (big 3x3 launches); other rows of >= 16 B: ragged tiled; 5-8 channels: the tiled
                                        kernel for the 3x3; else generic */
    MI_BLUR_VARIANT_GENERIC = 1,     /* one output byte per thread, any shape */
    MI_BLUR_VARIANT_TILED = 2,       /* LDS halo tile + 16-B vector loads (ragged form for odd pitches; _INVALID if rows < 16 B, C > 8, or
                                        C > 4 with the 5x5) */
    MI_BLUR_VARIANT_STREAM = 3       /* barrier-free: every wave streams its band through a wave-private LDS row ring
                                        (LDS-DMA, counted vmcnt), sliding window of row sums in registers (same eligibility) */,
    MI_BLUR_VARIANT_DIRECT = 4      /* no LDS: rows straight into registers, x-neighbours by DPP wave shifts (same eligibility) */
} mi_blur_variant;

const char *mi_blur_strerror(int status);
int mi_blur_version(void);
/* Which kernel the calling thread's most recent launch went to ("blur_tiled_kernel", "blur_direct_kernel",
 * "blur_fused_kernel", "blur_tiled_loop_kernel", "blur_stream_kernel", "blur_generic_kernel"; "" before the first):
 * reports name the kernel a profiler will show.  Static string, never NULL. */
const char *mi_blur_last_kernel(void);

/* Kernel tuning knobs (A/B benching and tests; defaults are the shipped configuration).  PROCESS-WIDE: a call
 * publishes a new set of knobs that every launch issued afterwards (by any context, any thread) takes a coherent
 * copy of; a launch racing with the call uses the old set or the new one, never a mix.
 *   "stage_dma"        1 = stage LDS tiles with global_load_lds (default), 0 = through VGPRs
 *   "rows_per_thread"  0 (default: chosen per launch from the grid size) | 4 | 8 | 16 output rows per thread
 *   "xcd_remap"        1 = XCD-aware blockIdx->tile map (default), 0 = identity
 *   "xcd_run"          which XCD-aware map: 0 | 1 (default) one contiguous eighth of the launch's tiles per XCD |
 *                      r >= 2 runs of r consecutive tiles dealt to the XCDs in turn
 *   "experiment"       1 = the tiled kernel's other row-pass form (field pairs straight from the raw window for 3x3,
 *                      split-then-shift for 5x5; C = 3 only) — A/B runs
 *   "stream_updown"    streaming variant: 1 (default) = odd bands march upwards (both readers of a band seam come at the
 *                      same time: fewer HBM re-reads), 0 = every band downwards
 *   "row_shuffle"      1 = x-neighbour bytes by DPP wave shifts (LDS only at wave/tile edges), 0 = from LDS (default)
 *   "prefer_stream"    1 = AUTO picks the streaming variant instead of the tiled one (default 0)
 *   "prefer_direct"    AUTO and the direct (LDS-free) variant: 0 never | 1 (default) where it measured faster: every 5x5
 *                      launch and 3x3 launches of up to 128 MiB of output | 2 for every eligible shape
 *   "stream_band_rows" streaming variant: output rows per wave (0 = chosen per launch)
 *   "ragged_tiled"     1 (default) = rows that are not a multiple of 16 bytes / unaligned pointers take the ragged form of the
 *                      tiled kernel; 0 = they take the generic one-byte-per-thread kernel
 *   "zero_copy"        1 (default) = submits whose input AND output are pinned host memory run the kernel on the
 *                      caller's buffers in place (no staging copies); 0 = always H2D -> kernel -> D2H
 *   "zero_copy_server"  1 (default) = zero-copy submits of aligned shapes go through the context's BATCH SERVER: one long-lived
 *                      dispatch (blur_server_kernel) takes batch after batch from a descriptor ring in pinned memory, hands the
 *                      tiles out through one ticket counter that runs through the batches and signals each complete batch
 *                      into host memory — no launch per batch, and no batch waiting for its unluckiest workgroup (+27 % images/s
 *                      end to end at batch 35, profiles/r03_e2e_timeline.md); 0 = one capped launch per batch (the two knobs below)
 *   "zero_copy_workers" 48 (default): the server's worker workgroups (read when a context's server is made)
 *   "zero_copy_idle_us" 300 (default): a server leaves after this long without a new batch (the next submit starts one)
 *   "zero_copy_budget"  256 (default): ... and after this many batches; the next one, already queued behind it, carries on
 *   "zero_copy_tickets" 1 (default) | 0 = a fixed share of tiles per worker (A/B runs: what a per-batch launch does)
 *   "resident_place_trials" 4 (default): see mi_blur_resident_alloc
 *   "zero_copy_spin"    0 (default) = a wait for a batch of the server spins ~20 us, then sleeps in 20 us steps (the core is
 *                      free for the threads that build the next batch); 1 = spin + yield only
 *   "zero_copy_server_min_kb"  1280 (default): in-place submits whose output is smaller than this many KiB take one launch each
 *                       instead of the server — its hand-off costs ~26 us per batch against ~8 us for a launch (a single 256x256
 *                       frame per submit: 126 k against 40 k img/s); 0 = always the server
 *   "staged_server"    1 (default) | 0: submits of PAGEABLE caller memory (the reference's malloc'd batch buffers) are gathered into
 *                       the slot's pinned staging and blurred there in place by the batch server (staging in -> staging out over the
 *                       link, 220 k img/s at batch 35 and 500) instead of a DMA copy each way around a launch (120-158 k).
 *                       MI_BLUR_STAGING_THREADS (default 8) = host threads of the gather / scatter
 *   "zero_copy_trace"   0 (default) | 1 = the server's workers stamp their phases (mi_blur_debug_zc_trace)
 *   "zero_copy_debug_base"  test hook, 0 (default) | n: a context's server starts n batches / 7n tiles short of 2^32, so the
 *                       wrap of its 32-bit batch and tile numbers (80 minutes into a continuous stream) is reached at once
 *   "zero_copy_events"  1 (default) | 0 = per-batch launches carry no timestamp events (timing experiment; no kernel bucket)
 *   "zero_copy_streams" 4 (default): zero-copy submits of a context alternate over this many of its streams (at most n_slots)
 *   "zero_copy_blocks"  24 (default): zero-copy launches of the aligned tiled kernel keep at most this many workgroups
 *                      resident, each looping over tiles (0 = one workgroup per tile).  The host link needs ~100 KB in
 *                      flight; a launch that puts every tile on the chip at once reads everything, then writes
 *                      everything (half duplex).  4 x 24 workgroups, each on its own read/compute/write cycle, keep both
 *                      directions busy: +10-18 % images/s end to end (profiles/r02_e2e.txt)
 *   "fused_window"     8 (default): the fused stream takes its tiles window by window, a window being this many consecutive
 *                      batches dealt over the XCDs (8 = one whole batch per XCD); the batches of a window complete at about
 *                      the same time, windows in stream order.  1 = batch by batch
 *   "fused_tail"       30 (default; MI_BLUR_FUSED_TAIL in the environment): per mille of a fused pass's tiles that are handed out
 *                      by ticket to spare workgroups at the end of the grid instead of being mapped to workgroups, so that an
 *                      XCD that runs ahead takes more of them (passes of >= 8192 tiles; -1.3..1.8 % per pass).  0 = static map
 *   "fused_tail_blocks" 25 (default): spare workgroups beyond the number of tail tiles, per cent
 *   "fused_release"    0 (default) | 1: how a block of the fused stream publishes "my outputs are in memory" — see
 *                      mi_blur_resident_run_fused */
int mi_blur_set_option(const char *key, int value);

/* Number of visible HIP devices (0 is a valid answer: CPU-device contexts still work).
 * Replaces the platform/device scan, heterogeneous_blur.c:142-184. */
int mi_blur_device_count(void);

/* ------------------------------------------------------------------------
 * Kernel level — replaces clSetKernelArg x5 + clEnqueueNDRangeKernel
 * (heterogeneous_blur.c:380-389,525; split_image_blur.c:407-418,533).
 *
 * d_in/d_out are DEVICE pointers on the current HIP device, n_images images laid
 * end to end (image stride = width*height*channels).  One launch blurs the whole
 * batch.  `stream` is a hipStream_t (NULL = default stream).  Asynchronous.
 * ---------------------------------------------------------------------- */
int mi_blur_enqueue(const uint8_t *d_in, uint8_t *d_out, int width, int height, int channels,
                    int radius, int n_images, void *stream);

/* Band form (Approach 2, split_image_blur.c:401,414,511-541): the input is a
 * band of `band_rows` rows (clamping happens at the band's own first/last row,
 * exactly as when the reference passes the sub-buffer height as `height`);
 * only output rows [out_row_begin, out_row_end) of the band are produced and
 * are written to d_out starting at d_out[0] (the halo rows the reference
 * computes and then drops at read-back, :526,537, are never computed). */
int mi_blur_enqueue_band(const uint8_t *d_in, uint8_t *d_out, int width, int band_rows, int channels,
                         int radius, int out_row_begin, int out_row_end, void *stream);

/* Full-control form used by the tests: variant selection; n_images bands. */
int mi_blur_enqueue_ex(const uint8_t *d_in, uint8_t *d_out, int width, int band_rows, int channels,
                       int radius, int n_images, int out_row_begin, int out_row_end,
                       int variant, void *stream);

/* ------------------------------------------------------------------------
 * Queue level — one context = one device + its in-order stream(s), pinned
 * staging slots, device buffers and event timing.  Replaces context/queue/
 * buffer creation (heterogeneous_blur.c:194-212,341-354), the per-image
 * Write/NDRange/Read triple (:502-533), clFinish (:538-539) and the event
 * bookkeeping (:544-579).
 * ---------------------------------------------------------------------- */
typedef struct mi_blur_ctx mi_blur_ctx;

typedef struct mi_blur_timing {      /* cumulative since create / last reset; mirrors :411-412 */
    double h2d_ms;                   /* transfer IN  (time_*_transfer_in)  */
    double kernel_ms;                /* kernel       (time_*_kernel); zero-copy submits that overlap count the time at
                                        least one of them was executing, not the sum of their durations */
    double d2h_ms;                   /* transfer OUT (time_*_transfer_out) */
    uint64_t bytes_h2d, bytes_d2h;
    uint64_t bytes_alg;              /* algorithmic bytes = 2*W*rows*C per image processed */
    uint64_t images;
    uint64_t launches;
} mi_blur_timing;

/* device: HIP ordinal, or MI_BLUR_DEVICE_CPU (n_threads host threads; 0 = all cores, at most 16 — ask for more by number).
 * max_batch: largest n_images of one submit.  n_slots: submits in flight (>=1).  Staged submits: 2-3 lets H2D(n+1),
 * kernel(n) and D2H(n-1) overlap.  In-place (pinned) submits through the batch server: a deeper queue costs the link
 * nothing and keeps the GPU fed while the host builds the next batch — the hosts use 4 (3 in split_image_blur).  Each
 * slot owns one stream and — made by the first submit that needs them, so never for a context whose submits are all in
 * place — a pinned in/out staging pair and a device in/out pair of max_batch images each. */
int mi_blur_create(mi_blur_ctx **out_ctx, int device, int width, int height, int channels,
                   int radius, int max_batch, int n_slots, int n_threads);
void mi_blur_destroy(mi_blur_ctx *ctx);

/* Page-locked, device-visible host memory for stream buffers.  When both buffers of a submit come
 * from here (or are otherwise pinned) the kernel works on them in place over PCIe — the counterpart of
 * CL_MEM_USE_HOST_PTR on the reference's shared-memory iGPU, and the fastest end-to-end form on this
 * platform; the h2d/d2h buckets of such a submit are ~0 and the transfer time is inside kernel_ms.
 * Such submits are batches of the context's batch server (see "zero_copy_server"): the submit publishes a descriptor,
 * mi_blur_wait_oldest / mi_blur_sync spin on the batch's completion word in host memory.
 * With only one side pinned, or "zero_copy" off, submit() DMA-copies straight from/to pinned memory;
 * ordinary (pageable) caller memory is accepted too and goes through the slot's own pinned staging
 * buffers (one extra host memcpy each way). */
void *mi_blur_host_alloc(size_t bytes);
void mi_blur_host_free(void *p);

/* Pin the caller's OWN buffers in place (hipHostRegister): a host that keeps its malloc'd batch buffers
 * (heterogeneous_blur.c:431-432, hoisted out of the batch loop) gets the in-place path of mi_blur_host_alloc memory without
 * changing its allocator.  The range must stay allocated until mi_blur_host_unregister; registering costs ~1 ms per 10 MB,
 * so do it once, not per batch.  Without a GPU both calls succeed and do nothing. */
int mi_blur_host_register(void *p, size_t bytes);
int mi_blur_host_unregister(void *p);

/* NUMA placement of per-GPU host work.  The reference drives both of its devices from one host thread of a one-socket
 * desktop (heterogeneous_blur.c:482-539); an 8-GPU node has two sockets with four GPUs each, and a feeder thread, a
 * batch-building memcpy (:439-442) or a pinned batch buffer on the other socket puts every byte of the stream on the
 * inter-socket link first.  No libnuma involved: hipDeviceGetPCIBusId -> /sys/bus/pci/devices/<bdf>/local_cpulist ->
 * sched_setaffinity.  MI_BLUR_NO_AFFINITY=1 in the environment turns the two binding calls into no-ops.
 *   mi_blur_device_cpulist         the GPU's local CPU list as sysfs spells it ("0-63,128-191") and its NUMA node
 *                                  (-1 if unknown); MI_BLUR_ERR_UNSUPPORTED when sysfs does not say
 *   mi_blur_bind_thread_to_device  restrict the CALLING thread (and threads it starts afterwards) to the GPU's local CPUs
 *                                  that its current mask allows; returns how many CPUs that is, 0 = nothing was changed
 *                                  (disabled, unknown topology, or no allowed CPU on that socket)
 *   mi_blur_host_alloc_on          mi_blur_host_alloc with `device` current and the calling thread on that GPU's socket for
 *                                  the duration of the call: the buffers of GPU g's feeder belong on GPU g's node */
int mi_blur_device_cpulist(int device, char *buf, size_t n, int *numa_node);
int mi_blur_bind_thread_to_device(int device);
void *mi_blur_host_alloc_on(int device, size_t bytes);

/* Asynchronous H2D -> ONE batched launch -> D2H of n_images images from caller
 * memory.  host_in/host_out must stay valid until mi_blur_sync (the reference
 * frees batch_input/batch_output only after clFinish, heterogeneous_blur.c:538,596-597).
 * Blocks only when every slot is still in flight. */
int mi_blur_submit(mi_blur_ctx *ctx, const uint8_t *host_in, uint8_t *host_out, int n_images);

/* Approach 2: host_in points at the first row of a band of band_rows rows that
 * already includes halo_top/halo_bottom halo rows; host_out receives the
 * band_rows-halo_top-halo_bottom interior rows (split_image_blur.c:511-541).
 * band_rows may differ per call but must be <= the context's height. */
int mi_blur_submit_band(mi_blur_ctx *ctx, const uint8_t *host_in, uint8_t *host_out,
                        int band_rows, int halo_top, int halo_bottom);

/* Approach 2 for a whole batch: the SAME band (rows [r, r+band_rows) incl. halos) of n_images
 * images that lie host_image_stride bytes apart in caller memory (the contiguous batch stream
 * of split_image_blur.c:469-480).  host_in points at the band's first row in image 0, host_out
 * at the band's first OUTPUT row in image 0 (same stride).  One 2-D DMA gathers the bands, one
 * launch blurs them, one 2-D DMA scatters the interior rows back — instead of the reference's
 * per-image Write/NDRange/Read on each device (:520-541). */
int mi_blur_submit_bands(mi_blur_ctx *ctx, const uint8_t *host_in, uint8_t *host_out, int n_images,
                         size_t host_image_stride, int band_rows, int halo_top, int halo_bottom);

/* Frames as the reference's loader hands them over — PLANAR, byte (x, y, c) of frame i at (i*C + c)*W*H + y*W + x (CImg
 * storage; the reference interleaves each frame on one host core, heterogeneous_blur.c:125-134, and de-interleaves to
 * save, split_image_blur.c:40-56).  Same contract as mi_blur_submit, with both repacks done by the GPU inside the
 * submit: the repack-in kernel reads the planar frames over the host link (pinned caller memory is read in place) and
 * writes the interleaved batch to HBM, the blur runs HBM -> HBM, and host_out receives interleaved frames
 * (planar_out = 0) or planar ones (planar_out != 0; written straight into pinned caller memory by the repack-out
 * kernel).  h2d_ms / d2h_ms of such a submit are the repack-in / copy-or-repack-out durations. */
int mi_blur_submit_planar(mi_blur_ctx *ctx, const uint8_t *host_planar_in, uint8_t *host_out, int n_images, int planar_out);

/* Block until the OLDEST submit still in flight has finished (its output is in caller memory).
 * Lets a host rotate n_slots batch buffers: refill the oldest while the newer ones run. */
int mi_blur_wait_oldest(mi_blur_ctx *ctx);

/* clFinish + event harvest (heterogeneous_blur.c:538-579).  timing may be NULL. */
int mi_blur_sync(mi_blur_ctx *ctx, mi_blur_timing *timing);
void mi_blur_reset_timing(mi_blur_ctx *ctx);
/* Non-blocking: the buckets of the submits harvested so far (by wait_oldest / sync).  A host that rotates
 * batch buffers reads it after each mi_blur_wait_oldest to get that batch's device times — what the
 * reference prints once per run (heterogeneous_blur.c:541-579) becomes available per batch, which is
 * what continuous CPU/GPU rebalancing needs (`both auto`). */
int mi_blur_get_timing(mi_blur_ctx *ctx, mi_blur_timing *timing);
/* How many of the context's submits so far were blurred in place over the host link: pinned caller buffers (see
 * mi_blur_host_alloc), and pageable ones by way of the slot's pinned staging ("staged_server", default on). */
uint64_t mi_blur_zero_copy_launches(mi_blur_ctx *ctx);

/* ------------------------------------------------------------------------
 * Frame layout on the device.  CImg (the reference's image loader) stores frames
 * PLANAR — byte (x, y, c) of image i at (i*C + c)*W*H + y*W + x — and the
 * reference repacks each frame to the interleaved stream on one host core
 * (heterogeneous_blur.c:125-134; back again to save, split_image_blur.c:40-56).
 * These do the same repack for n_images frames in one HBM pass each way.
 * Buffers are device (or pinned host) memory, distinct, n_images*W*H*C bytes.
 * (No repack at all is needed to BLUR planar frames: a planar stream is a stream
 * of n_images*C one-channel images — mi_blur_enqueue(..., channels = 1, n*C).)
 * ---------------------------------------------------------------------- */
int mi_blur_planar_to_interleaved(const uint8_t *d_planar, uint8_t *d_interleaved, int width, int height,
                                  int channels, int n_images, void *stream);
int mi_blur_interleaved_to_planar(const uint8_t *d_interleaved, uint8_t *d_planar, int width, int height,
                                  int channels, int n_images, void *stream);

/* ------------------------------------------------------------------------
 * Device-resident stream (no reference analogue: the reference re-uploads
 * every image).  The pool holds pool_images images in HBM (in + out).
 * ---------------------------------------------------------------------- */
/* Pools of 128 MiB .. 8 GiB are PLACED: where a pool lands in HBM moves the big launches between two levels ~6 % apart
 * (profiles/r03_placement_channels.txt: same requests per channel, 1.5x the read DRAM-credit stalls on the slow placements),
 * nothing in the address tells which, and a placement keeps its level while it lives — so "resident_place_trials" (default 4;
 * MI_BLUR_PLACE_TRIALS in the environment; 0/1 = off) candidate inputs and as many candidate outputs are allocated side by
 * side, every (input, output) pair is timed on the context's kernel over the whole pool (after a clock ramp), the fastest
 * pair is kept, the other buffers are freed (~0.15 s, once per pool).  mi_blur_resident_placement reports the pairs'
 * per-launch ms (row-major: input i, output j at i*n + j) and which one was kept. */
int mi_blur_resident_alloc(mi_blur_ctx *ctx, int pool_images);
int mi_blur_resident_placement(mi_blur_ctx *ctx, float *ms, int max_n, int *kept);
/* Fill pool image i with the synthetic LCG image (seed 0x9E3779B9 ^ (first_index+i)). */
int mi_blur_resident_fill_synthetic(mi_blur_ctx *ctx, int first_index);
int mi_blur_resident_upload(mi_blur_ctx *ctx, int pool_index, const uint8_t *host_in, int n_images);
int mi_blur_resident_download(mi_blur_ctx *ctx, int pool_index, uint8_t *host_out, int n_images);
void *mi_blur_resident_in(mi_blur_ctx *ctx);       /* device pointers of the pool */
void *mi_blur_resident_out(mi_blur_ctx *ctx);

/* One pass of the image stream over the resident pool: n_images images taken
 * cyclically from the pool, one launch per `batch` images (the last launch takes
 * the remainder, heterogeneous_blur.c:423-427).  Launches go round-robin over the
 * context's streams.  Every timed_every-th launch of the pass (0 = none, 1 = all)
 * carries the dispatch's own start/stop timestamps (the HIP analogue of
 * clGetEventProfilingInfo); their durations accumulate into the context's
 * kernel_ms and mi_blur_timed_coverage says how many launches/bytes that covers.
 * (Timestamped launches cost the host ~3x an ordinary launch, so a small-batch
 * stream is sampled rather than timed in full.)  Asynchronous; follow with mi_blur_sync. */
int mi_blur_resident_run(mi_blur_ctx *ctx, int n_images, int batch, int timed_every);
void mi_blur_timed_coverage(mi_blur_ctx *ctx, uint64_t *launches, uint64_t *bytes_alg);

/* The same pass as ONE dispatch ("fused stream").  A batch-35 stream issued launch by launch is bound by the
 * GPU's per-dispatch processing (~3.5 us each over 4 hardware queues), not by the kernel; here the kernel walks
 * the batches itself — blocks ordered in windows of 8 batches, in stream order — and every workgroup counts itself into its batch's counter once
 * its outputs are in memory.  The batch stays the unit of COMPLETION (mi_blur_resident_batches_done reads the
 * counters and returns how many leading batches have their outputs ready, without waiting for the dispatch, which
 * may still be running) without being the unit of DISPATCH.  n_images <= pool size (one contiguous run of the pool); 1-4 channels, rows of
 * at least 16 bytes — rows that are a multiple of 16 bytes take the aligned tiles, any other width their ragged form
 * (MI_BLUR_ERR_UNSUPPORTED otherwise).  `timed` is a bit set: 1 = the dispatch carries timestamp events like resident_run;
 * 2 = WATCH this pass: a one-wave kernel on a stream of its own follows the counters and keeps "leading batches complete" in
 * pinned host memory, so that mi_blur_resident_batches_done is a read of the caller's own memory (tens of ns, no HIP call)
 * instead of a counter read-back (~18 us) — for hosts that consume batch by batch while the pass runs.
 * Asynchronous; follow with mi_blur_sync.
 *
 * Ordering of "counted" against "readable".  Default ("fused_release" 0): outputs are stored write-through
 * (global_store ... sc1), every wave drains its stores (s_waitcnt vmcnt(0)), the block meets at a barrier and one lane
 * adds to the counter with a relaxed agent-scope atomic.  That a batch's outputs are in memory once its count is
 * visible is MEASURED behaviour of gfx950 / ROCm 7.2 (MI355X_MICROARCH.md, cross-XCD hand-off table), not an
 * architectural guarantee of the memory model; tests/test_gpu_parity.py checks it mid-dispatch.  "fused_release" 1
 * makes the add a release at agent scope — the architectural form, ~6x slower per pass (an L2 write-back per block). */
int mi_blur_resident_run_fused(mi_blur_ctx *ctx, int n_images, int batch, int timed);
/* >= 0: how many LEADING batches of the latest fused pass are complete.  NEGATIVE: a mi_blur_status — the counters
 * could not be read (failed copy, faulted device); a polling loop must stop on it, it will not recover by itself. */
int mi_blur_resident_batches_done(mi_blur_ctx *ctx);
/* Copy out outputs of batches already reported done, without waiting for the dispatch that is still producing the
 * later ones (mi_blur_resident_download waits for the whole device). */
int mi_blur_resident_peek(mi_blur_ctx *ctx, int pool_index, uint8_t *host_out, int n_images);

/* ------------------------------------------------------------------------
 * CPU device kernel, exposed for the hosts' `cpu` mode and for timing the
 * host-core path beside the GPU (BASELINE.json config 0).  Scalar per-pixel
 * integer form of gaussian_kernel.cl:19-72, n_threads host threads over images
 * (over row bands when n_images < n_threads).  Synchronous.
 * ---------------------------------------------------------------------- */
int mi_blur_cpu_run(const uint8_t *in, uint8_t *out, int width, int height, int channels,
                    int radius, int n_images, int n_threads);

/* Developer diagnostics.  With mi_blur_set_option("debug_xcd_times", 1) every workgroup of the tiled kernel leaves its
 * start and end time (100 MHz ticks) in a slot of the XCD it ran on; this call waits for the device, returns per XCD the
 * LATEST end and the EARLIEST start seen since the last re-arm, and re-arms the slots when asked (call it once with
 * rearm = 1 before the launches to be examined).  Shows which XCD a launch waits for (profiles/r02_xcd_finish_times.txt). */
int mi_blur_debug_xcd_times(uint64_t end_ticks[8], uint64_t begin_ticks[8], int rearm);

/* Developer diagnostics of the zero-copy batch server (mi_blur_set_option("zero_copy_trace", 1) before the context's first
 * zero-copy submit): per worker workgroup and batch, for the context's first 512 batches — device clock (100 MHz ticks)
 * when it took its first tile of the batch, how many tiles it took, ticks spent inside tiles, device clock when its last
 * tile of the batch ended, (unused).  Copies out[batch][worker][5] for the first min(max_batches, 512, published) batches
 * and returns how many that is (negative = mi_blur_status).  profiles/r03_e2e_timeline.md is built from it. */
int mi_blur_debug_zc_trace(mi_blur_ctx *ctx, uint64_t *out, int max_batches, int *n_workers, unsigned *batches_published);

/* Synthetic stream generator shared by hosts, bench and tests. */
void mi_blur_fill_synthetic(uint8_t *host, int width, int height, int channels,
                            int first_index, int n_images, int n_threads);
uint64_t mi_blur_fnv1a64(const uint8_t *host, size_t n);

/* ------------------------------------------------------------------------
 * Work distribution helpers — host logic of the two approaches.
 * ---------------------------------------------------------------------- */
/* Approach 1, heterogeneous_blur.c:449-458: mode 0 both / 1 cpu / 2 gpu. */
void mi_blur_a1_partition(int mode, int batch_count, float gpu_ratio, int *n_cpu, int *n_gpu);
/* Image-level sharding over G devices: shard g owns [begin,end). */
void mi_blur_shard_range(long long n_units, int g, int G, long long *begin, long long *end);

typedef struct mi_blur_a2_geometry { /* split_image_blur.c:144-166 with HALO := halo */
    int split_row;
    int cpu_input_rows, cpu_output_rows;
    int gpu_input_rows, gpu_output_rows;
} mi_blur_a2_geometry;
void mi_blur_a2_split(int height, float gpu_ratio, int halo, mi_blur_a2_geometry *g);

typedef struct mi_blur_band {        /* K-way row split of one image over G devices */
    int row_begin, row_end;          /* owned output rows [begin,end) */
    int halo_top, halo_bottom;       /* halo rows needed from neighbour g-1 / g+1 (0 at image edge) */
} mi_blur_band;
void mi_blur_band_of(int height, int radius, int g, int G, mi_blur_band *b);

/* ------------------------------------------------------------------------
 * Approach 2 on resident row shards: halo rows move GPU<->GPU with RCCL
 * send/recv over xGMI (the reference instead re-uploads overlapping slices
 * from the host, split_image_blur.c:516,520,530).
 * One communicator rank per GPU; either one process per GPU
 * (mi_blur_comm_unique_id on rank 0, broadcast the 128 bytes by any means,
 * mi_blur_comm_init_rank everywhere) or all ranks in one process
 * (mi_blur_comm_init_all).
 * ---------------------------------------------------------------------- */
typedef struct mi_blur_comm mi_blur_comm;
#define MI_BLUR_UNIQUE_ID_BYTES 128
int mi_blur_comm_unique_id(uint8_t id[MI_BLUR_UNIQUE_ID_BYTES]);
int mi_blur_comm_init_rank(mi_blur_comm **comm, int n_ranks, int rank, const uint8_t id[MI_BLUR_UNIQUE_ID_BYTES]);
int mi_blur_comm_init_all(mi_blur_comm **comms, int n_devices, const int *devices);
/* Single-process set that moves halo rows with hipMemcpyPeerAsync instead of RCCL (fallback transport; also the
 * only one that accepts several ranks on one device).  Use with mi_blur_halo_exchange_all. */
int mi_blur_comm_init_p2p(mi_blur_comm **comms, int n_devices, const int *devices);
/* The same single-process set with the halo rows PULLED by one small kernel per rank (mi_blur_halo_pull's) instead of pushed
 * with two peer copies; ordering by the same events.  Use with mi_blur_halo_exchange_all. */
int mi_blur_comm_init_pull(mi_blur_comm **comms, int n_devices, const int *devices);
void mi_blur_comm_destroy(mi_blur_comm *comm);
/* What the communicator is, as its transport reports it: *n_ranks / *rank from ncclCommCount / ncclCommUserRank for an
 * RCCL communicator (the numbers a report should quote for "RCCL carried the halos over N ranks"), the construction
 * arguments otherwise; *transport 0 = none (a single rank: both image edges clamp), 1 = RCCL, 2 = peer copies, 3 = pulled by a kernel.
 * Any out pointer may be NULL. */
int mi_blur_comm_info(mi_blur_comm *comm, int *n_ranks, int *rank, int *transport);

/* d_band: this rank's shard laid out as [halo_top rows][owned rows][halo_bottom rows]
 * (halo_* from mi_blur_band_of; absent halos have zero rows).  Sends the first/last
 * `radius` OWNED rows to rank-1 / rank+1 and receives their mirror into the halo
 * rows, in one RCCL group on `stream`.  Asynchronous. */
int mi_blur_halo_exchange(mi_blur_comm *comm, uint8_t *d_band, int width, int channels,
                          int owned_rows, int radius, void *stream);

/* A second way to fill the halo rows, for hosts that can hand device pointers between ranks: each rank PULLS its halo rows
 * out of its neighbours' shards with a small copy kernel that reads the peer's memory directly (over xGMI between GPUs) —
 * no collective library in the step, one ~4 us launch instead of a send/recv pair.  One process per GPU: every rank exports
 * its shard (mi_blur_peer_export: a 64-byte IPC handle of the allocation + the shard's offset inside it), the handles travel
 * by any means (bench.py: an all_gather), each rank opens its neighbours' (mi_blur_peer_open) and from then on calls
 * mi_blur_halo_pull(d_band, top_src, bottom_src, ...) per step with top_src = address of the LAST `radius` owned rows of rank
 * g-1's shard and bottom_src = the FIRST `radius` owned rows of rank g+1's (NULL where there is no neighbour).  A rank may
 * only pull rows its neighbour is not writing at the time: for the repeated blur of one resident image (BASELINE configs[4])
 * the owned rows never change; an iterated blur needs a cross-rank barrier per step (not provided here).
 * mi_blur_peer_close when done, before the owner frees the shard. */
#define MI_BLUR_PEER_HANDLE_BYTES 64
int mi_blur_peer_export(const void *d_ptr, uint8_t handle[MI_BLUR_PEER_HANDLE_BYTES], uint64_t *offset);
int mi_blur_peer_open(const uint8_t handle[MI_BLUR_PEER_HANDLE_BYTES], uint64_t offset, void **d_ptr);
int mi_blur_peer_close(void *d_ptr, uint64_t offset);
int mi_blur_halo_pull(uint8_t *d_band, const uint8_t *top_src, const uint8_t *bottom_src, int width, int channels,
                      int owned_rows, int radius, void *stream);
/* The step as ONE launch: mi_blur_enqueue_band whose kernel reads band rows [0, out_row_begin) from top_src and rows
 * [out_row_end, band_rows) from bottom_src (each laid out as that many rows, e.g. the same peer addresses mi_blur_halo_pull
 * takes when out_row_begin = radius) instead of from d_in — the halo rows of d_in are never read or written, the neighbours'
 * rows cross xGMI inside the blur kernel's own loads.  NULL for a side = that side's rows come from d_in as usual (both NULL =
 * mi_blur_enqueue_band).  Same rule as the pull: the neighbours must not be writing those rows.  Shapes of the direct kernel
 * only (1-4 channels, rows a multiple of 16 bytes, 16-byte aligned pointers; MI_BLUR_ERR_UNSUPPORTED otherwise — use
 * mi_blur_halo_pull + mi_blur_enqueue_band then). */
int mi_blur_enqueue_band_peer(const uint8_t *d_in, uint8_t *d_out, int width, int band_rows, int channels, int radius,
                              int out_row_begin, int out_row_end, const uint8_t *top_src, const uint8_t *bottom_src,
                              void *stream);

/* Single-process form: all n ranks of a mi_blur_comm_init_all set in one RCCL group. */
int mi_blur_halo_exchange_all(mi_blur_comm **comms, int n, uint8_t **d_bands, int width, int channels,
                              const int *owned_rows, int radius, void **streams);

#ifdef __cplusplus
}
#endif
#endif /* MI_BLUR_H */
