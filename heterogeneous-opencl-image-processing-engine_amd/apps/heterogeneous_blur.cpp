// heterogeneous_blur — Approach 1 (image-level distribution) host, MI355X-native.
//
//   heterogeneous_blur {cpu|gpu|both} [gpu_ratio] [batch]  [--image F | --synthetic | --size WxH] [--channels C]
//                      [--ksize 3|5] [--images N] [--gpus G] [--slots S] [--threads T] [--resident [--fused]]
//                      [--verbose] [--csv FILE] [--save FILE]
//                      [--frames DIR|PATTERN|FILE [--save-dir DIR] [--planar-out | --native-layout]]   (cpu | gpu)
//
// Same positional command line, banner and report sections as the reference host
// (heterogeneous_blur.c:41-100 CLI, :406-601 batch loop, :609-724 report).  What changed:
//   * the OpenCL plumbing (:140-403) is the C ABI of libmi_blur.so: one context per device,
//     the `cpu` device is native host threads (ROCm has no OpenCL CPU device), the `gpu` device is
//     1..G MI355X, each with its own HIP streams — images are independent, no collectives;
//   * a batch is ONE launch per device (plus one DMA each way), not one Write/NDRange/Read per
//     image (:502-533); the first n_cpu images of a batch go to the CPU device, the rest are split
//     contiguously over the GPUs (:449-458,496);
//   * batch buffers are pinned and rotate through `slots` sets, so batch n+1 is built and uploaded
//     while batch n computes; the reference's per-batch malloc/free (:431-432,596-597) is hoisted,
//     its per-batch stream construction (memcpy of the source image into every slot, :440-442)
//     stays inside the timed region as in the reference;
//   * `gpu` mode runs ONE FEEDER THREAD PER GPU (the reference's one host thread drives both of its devices,
//     :482-539): feeder g builds GPU g's contiguous share of every batch in its own pinned buffers and submits it;
//     the feeder, its batch-building helpers and its buffers keep to the CPUs / memory of that GPU's socket
//     (mi_blur_bind_thread_to_device, mi_blur_host_alloc_on; MI_BLUR_NO_AFFINITY=1 turns that off).
#include "host_common.h"

using namespace host;

static int run_frames(const Options &opt, int mode, int BATCH_SIZE, const std::vector<std::string> &files);

struct Dev {
    mi_blur_ctx *ctx = nullptr;
    std::string name;
    mi_blur_timing tm{};
    std::vector<char> submitted;   // per batch: did this device get work?
};

int main(int argc, char **argv)
{
    // ---------------- configuration (heterogeneous_blur.c:41-100)
    int mode = 0;                                 // 0 both, 1 cpu, 2 gpu
    const char *input_filename = "./image_320x240.jpg";
    int BATCH_SIZE = 500;
    const int local_work_size = 16;
    float gpu_ratio = 0.5f;

    Options opt;
    const int npos = parse_flags(argc, argv, opt);
    // --frames: the stream is a set of DISTINCT frame files (not one image copied NUM_IMAGES times, :439-442)
    std::vector<std::string> frame_files;
    if (!opt.frames.empty()) {
        frame_files = list_frames(opt.frames);
        if (frame_files.empty()) { printf("Error: no frame files found for --frames %s\n", opt.frames.c_str()); return -1; }
        if (opt.images_given && opt.images < (int)frame_files.size()) frame_files.resize(opt.images);
        opt.images = (int)frame_files.size();
    }
    const int NUM_IMAGES = opt.images;

    if (npos > 1) {
        if (strcmp(argv[1], "cpu") == 0) { mode = 1; printf("Mode: CPU ONLY\n"); }
        else if (strcmp(argv[1], "gpu") == 0) { mode = 2; printf("Mode: GPU ONLY\n"); }
        else if (strcmp(argv[1], "both") == 0) { mode = 0; printf("Mode: HETEROGENEOUS (CPU + GPU)\n"); }
        else { printf("Usage: %s [cpu|gpu|both]\n", argv[0]); printf("Defaulting to heterogeneous mode.\n"); }
    } else {
        printf("Mode: HETEROGENEOUS (CPU + GPU) [default]\n");
    }
    if (npos > 2 && !strcmp(argv[2], "auto")) {
        opt.auto_ratio = true;                    // closes the loop the reference leaves to the user (section 8 of its report)
    } else if (npos > 2) {
        gpu_ratio = atof(argv[2]);
        if (gpu_ratio < 0.0f || gpu_ratio > 1.0f) {
            printf("Warning: gpu_ratio must be between 0.0 and 1.0. Using 0.5\n");
            gpu_ratio = 0.5f;
        }
    }
    if (npos > 3) {
        BATCH_SIZE = atoi(argv[3]);
        if (BATCH_SIZE < 1 || BATCH_SIZE > NUM_IMAGES) {
            printf("Warning: BATCH_SIZE must be between 1 and %d. Using 500\n", NUM_IMAGES);
            BATCH_SIZE = 500;
        }
    }
    if (BATCH_SIZE > NUM_IMAGES) BATCH_SIZE = NUM_IMAGES;
    const int NUM_BATCHES = (NUM_IMAGES + BATCH_SIZE - 1) / BATCH_SIZE;
    if (!opt.image.empty()) input_filename = opt.image.c_str();
    if (opt.resident && mode != 2) { printf("Error: --resident needs mode gpu\n"); return -1; }

    if (mode == 0 && opt.auto_ratio) printf("GPU ratio: auto (first batch split 50/50, then the measured optimum)\n");
    else if (mode == 0) printf("GPU ratio: %.1f%% GPU, %.1f%% CPU\n", gpu_ratio * 100, (1 - gpu_ratio) * 100);
    printf("========== HETEROGENEOUS CONFIGURATION ==========\n");
    if (!frame_files.empty()) printf("Input frames: %s (%zu files, %s ... %s)\n", opt.frames.c_str(), frame_files.size(), frame_files.front().c_str(), frame_files.back().c_str());
    else printf("Input file: %s\n", opt.synthetic ? "(synthetic)" : input_filename);
    printf("Number of images in stream: %d\n", NUM_IMAGES);
    printf("Batch size: %d images\n", BATCH_SIZE);
    printf("Number of batches: %d\n", NUM_BATCHES);
    printf("Work-group size: %dx%d\n", local_work_size, local_work_size);
    printf("Execution mode : %d\n", mode);
    printf("Blur kernel: %dx%d\n", opt.ksize, opt.ksize);
    printf("================================================\n\n");

    if (!frame_files.empty()) {
        if (mode == 0) { printf("Error: --frames runs on one device kind: use mode cpu or gpu\n"); return -1; }
        if (opt.resident) { printf("Error: --frames and --resident exclude each other\n"); return -1; }
        return run_frames(opt, mode, BATCH_SIZE, frame_files);
    }

    // ---------------- load original image (heterogeneous_blur.c:104-137)
    Image img = load_image(input_filename, opt.syn_w, opt.syn_h, opt.syn_c, opt.synthetic);
    if (!opt.save_input.empty()) save_one_image(opt.save_input.c_str(), img.px.data(), img.width, img.height, img.channels);
    const int width = img.width, height = img.height, channels = img.channels;
    const int radius = opt.ksize == 3 ? 1 : 2;
    printf("Original image loaded: %dx%d, %d channels\n", width, height, channels);
    const size_t image_size = (size_t)width * height * channels;
    printf("Size of one image: %zu bytes (%.2f KB)\n", image_size, image_size / 1024.0);
    printf("Original image source: %s\n\n", img.source.c_str());
    const uint8_t *original_image = img.px.data();

    // ---------------- device discovery (heterogeneous_blur.c:140-212)
    const int ngpu_visible = mi_blur_device_count();
    int G = mode == 1 ? 0 : opt.gpus;
    if (mode != 1 && !gpus_available(G)) {
        printf("Error: Could not find %d GPU device(s) (%d visible)\n", G < 1 ? 1 : G, ngpu_visible);
        return -1;
    }
    // --resident: launches alternate over the context's streams so the ~4 us per-dispatch floors overlap; 4 streams
    // (one per hardware queue) measured best, and only every 16th dispatch carries timestamps (a timestamped launch
    // costs the host ~3x an ordinary one) — the kernel bucket is scaled up from that sample.
    // GPU-only mode from host buffers: 4 rotating buffer sets, so up to 4 zero-copy launches (each capped to a few dozen
    // workgroups, library default) overlap and keep both directions of the host link busy: 173-180 k img/s at batch 35
    // against 141-146 k with 2 sets (profiles/r02_e2e.txt).  With CPU threads in the loop (both) 2 sets stay best.
    // buffer sets in flight: 4 wherever a GPU takes part (with the batch server a deeper queue costs the link nothing and keeps
    // the GPU fed while the host builds / the CPU device works: `both` 35 runs 25-40 % faster on 4 sets than on 2,
    // profiles/r03_hosts_e2e.txt), the reference's 2 for the CPU device alone
    const int nslots = opt.slots_given ? opt.slots : (mode != 1 ? 4 : opt.slots);
    const int resident_timed_every = 16;
    Dev cpu;
    std::vector<Dev> gpus(G);
    if (mode != 2) {
        mi_check(mi_blur_create(&cpu.ctx, MI_BLUR_DEVICE_CPU, width, height, channels, radius, BATCH_SIZE, nslots, opt.threads),
                 "Failed to create CPU context");
        const unsigned hc = std::thread::hardware_concurrency();
        cpu.name = "host threads x" + std::to_string(opt.threads > 0 ? opt.threads : std::min((int)(hc ? hc : 1), 16));
        printf("CPU device: %s\n", cpu.name.c_str());
        cpu.submitted.assign(NUM_BATCHES, 0);
    }
    // gpu mode from host buffers: one feeder thread per GPU, each with its own buffers, placed on the GPU's socket.  A big
    // batch (the reference's default is 500) is built and handed over in up to 4 pieces, so that building the rest of a
    // batch overlaps the transfer of its first piece: the batch stays the unit the buffers rotate by (and of the report),
    // a piece is the unit of submission.
    const bool per_gpu_feeders = mode == 2 && !opt.resident;
    const int share_max = G > 0 ? (BATCH_SIZE + G - 1) / G : BATCH_SIZE;      // the largest contiguous share of a batch (mi_blur_shard_range)
    const int feed_pieces = per_gpu_feeders ? std::max(1, std::min(4, (share_max + 63) / 64)) : 1;
    const int feed_piece = (share_max + feed_pieces - 1) / feed_pieces;
    for (int g = 0; g < G; g++) {
        mi_check(per_gpu_feeders ? mi_blur_create(&gpus[g].ctx, hip_ordinal(g), width, height, channels, radius, feed_piece, nslots * feed_pieces, 0)
                                 : mi_blur_create(&gpus[g].ctx, hip_ordinal(g), width, height, channels, radius, BATCH_SIZE, nslots, 0),
                 "Failed to create GPU context");
        gpus[g].name = "HIP device " + std::to_string(hip_ordinal(g)) + (virtual_gpus() ? " (logical GPU " + std::to_string(g) + ")" : "");
        printf("GPU device: %s\n", gpus[g].name.c_str());
        gpus[g].submitted.assign(NUM_BATCHES, 0);
    }
    for (int g = 0; g < G; g++) report_placement(g, hip_ordinal(g), /*bind the calling thread*/ !per_gpu_feeders && g == 0);
    printf("\nKernel objects created (code objects are embedded in libmi_blur.so; nothing is read from the CWD)\n\n");

    // ---------------- buffers (heterogeneous_blur.c:330-357,431-437): pinned, `nslots` rotating sets
    // (--malloc: ordinary malloc'd memory, as the reference allocates them — every submit then goes through the library's pinned staging)
    auto batch_alloc = [&](int ordinal, size_t bytes) -> uint8_t * {
        if (opt.malloc_buffers) return (uint8_t *)malloc(bytes);
        return (uint8_t *)(ordinal >= 0 ? mi_blur_host_alloc_on(ordinal, bytes) : mi_blur_host_alloc(bytes));
    };
    auto batch_free = [&](uint8_t *p) { if (opt.malloc_buffers) free(p); else mi_blur_host_free(p); };
    if (opt.malloc_buffers) printf("Batch buffers: malloc (pageable), as in the reference; submits are staged through pinned memory by the library\n");
    printf("Allocating device buffers...\n");
    std::vector<uint8_t *> batch_input(nslots, nullptr), batch_output(nslots, nullptr);
    std::vector<std::vector<uint8_t *>> gin(G), gout(G);               // per_gpu_feeders: GPU g's own rotating buffer sets
    if (per_gpu_feeders) {
        for (int g = 0; g < G; g++) {
            const size_t share = (size_t)share_max;
            for (int s = 0; s < nslots; s++) {
                gin[g].push_back(batch_alloc(hip_ordinal(g), share * image_size));
                gout[g].push_back(batch_alloc(hip_ordinal(g), share * image_size));
                if (!gin[g].back() || !gout[g].back()) { printf("Error: Failed to allocate batch memory\n"); return -1; }
            }
        }
    } else if (!opt.resident) {
        for (int s = 0; s < nslots; s++) {
            batch_input[s] = batch_alloc(-1, (size_t)BATCH_SIZE * image_size);
            batch_output[s] = batch_alloc(-1, (size_t)BATCH_SIZE * image_size);
            if (!batch_input[s] || !batch_output[s]) { printf("Error: Failed to allocate batch memory\n"); return -1; }
        }
    } else {
        // device-resident stream: each GPU holds its share of the stream in HBM before the clock starts
        for (int g = 0; g < G; g++) {
            long long b, e;
            mi_blur_shard_range(NUM_IMAGES, g, G, &b, &e);
            int pool = (int)(e - b);
            if (pool < BATCH_SIZE) pool = BATCH_SIZE;
            mi_check(mi_blur_resident_alloc(gpus[g].ctx, pool), "Failed to allocate resident pool");
            std::vector<uint8_t> rep((size_t)std::min(pool, 64) * image_size);
            for (int i = 0; i < std::min(pool, 64); i++) memcpy(rep.data() + (size_t)i * image_size, original_image, image_size);
            for (int i = 0; i < pool; i += 64)
                mi_check(mi_blur_resident_upload(gpus[g].ctx, i, rep.data(), std::min(64, pool - i)), "resident upload failed");
        }
    }
    printf("Device buffers allocated\n\n");
    printf("Global work size: %d x %d\n", (width + 15) / 16 * 16, (height + 15) / 16 * 16);
    printf("Local work size: %d x %d\n\n", local_work_size, local_work_size);

    // Warm-up outside the clock: first-use costs (code-object load, worker threads) would otherwise land in the first
    // batch — harmless for the totals, fatal for "auto", which calibrates on that batch.
    if (!opt.resident && !per_gpu_feeders) {
        const int nw = std::min(BATCH_SIZE, 4);
        for (int i = 0; i < nw; i++) memcpy(batch_input[0] + (size_t)i * image_size, original_image, image_size);
        if (cpu.ctx) { mi_check(mi_blur_submit(cpu.ctx, batch_input[0], batch_output[0], nw), "CPU warm-up failed"); mi_check(mi_blur_sync(cpu.ctx, nullptr), "CPU sync failed"); mi_blur_reset_timing(cpu.ctx); }
        for (auto &d : gpus) { mi_check(mi_blur_submit(d.ctx, batch_input[0], batch_output[0], nw), "GPU warm-up failed"); mi_check(mi_blur_sync(d.ctx, nullptr), "GPU sync failed"); mi_blur_reset_timing(d.ctx); }
    }

    if (opt.resident) {
        // The same entry point as the timed run (each kernel's code object is loaded on its first launch), repeated for
        // ~60 ms: after any idle gap the GPU needs ~40 ms of work to ramp its clocks (profiles/r02_clock_ramp.txt), and
        // this mode exists to measure the kernel, not the ramp.
        const double warm_until = get_time_ms() + 60.0;
        do {
            for (auto &d : gpus) {
                mi_check(opt.fused ? mi_blur_resident_run_fused(d.ctx, std::min(2 * BATCH_SIZE, NUM_IMAGES / G > 0 ? NUM_IMAGES / G : 1), BATCH_SIZE, 0)
                                   : mi_blur_resident_run(d.ctx, std::min(BATCH_SIZE, NUM_IMAGES), BATCH_SIZE, 0), "GPU warm-up failed");
            }
            for (auto &d : gpus) mi_check(mi_blur_sync(d.ctx, nullptr), "GPU sync failed");
        } while (get_time_ms() < warm_until);
        for (auto &d : gpus) mi_blur_reset_timing(d.ctx);
    }

    // ---------------- batch processing (heterogeneous_blur.c:406-601)
    printf("Starting batch processing of %d images in %d batches...\n\n", NUM_IMAGES, NUM_BATCHES);
    int total_images_cpu = 0, total_images_gpu = 0;
    std::vector<uint8_t> first_output;
    Replicator replicate(per_gpu_feeders ? 1 : opt.host_threads, (mode != 1 && G > 0) ? hip_ordinal(0) : -1);
    double last_harvest_ms = 0;
    int rebalances = 0;
    double time_start_total = get_time_ms();

    if (per_gpu_feeders) {
        // One feeder per GPU.  Feeder g owns images [b, e) = mi_blur_shard_range(batch_count, g, G) of EVERY batch (:496),
        // builds them in its own pinned buffers (the memcpy of :439-442, inside the timed region) and submits them; the
        // GPUs never wait for one another.  All feeders warm up (code-object load, worker threads, clocks), meet at a
        // gate, and the clock starts when the gate opens.
        std::atomic<int> at_gate{0};
        std::atomic<bool> go{false};
        std::vector<int> feed_rc(G, MI_BLUR_OK);
        std::vector<std::thread> feeders;
        for (int g = 0; g < G; g++) {
            feeders.emplace_back([&, g]() {
                const int ord = hip_ordinal(g);
                mi_blur_bind_thread_to_device(ord);
                Replicator rep(opt.host_threads, ord);
                mi_blur_ctx *ctx = gpus[g].ctx;
                auto ok = [&](int rc) { if (rc != MI_BLUR_OK && feed_rc[g] == MI_BLUR_OK) feed_rc[g] = rc; return rc == MI_BLUR_OK; };
                {
                    long long b, e;
                    mi_blur_shard_range(BATCH_SIZE, g, G, &b, &e);
                    const int nw = (int)std::min<long long>(e - b, 4);
                    if (nw > 0) {
                        rep.run(gin[g][0], original_image, image_size, nw, !opt.malloc_buffers);
                        ok(mi_blur_submit(ctx, gin[g][0], gout[g][0], nw)) && ok(mi_blur_sync(ctx, nullptr));
                        mi_blur_reset_timing(ctx);
                    }
                }
                at_gate.fetch_add(1);
                while (!go.load(std::memory_order_acquire)) std::this_thread::yield();
                long long k = 0;                                      // this feeder's batches so far
                std::vector<int> pieces_of(nslots, 0);                // how many submits the batch in buffer set s was handed over in
                for (int batch = 0; batch < NUM_BATCHES && feed_rc[g] == MI_BLUR_OK; batch++) {
                    const int batch_start = batch * BATCH_SIZE;
                    const int batch_count = std::min(BATCH_SIZE, NUM_IMAGES - batch_start);
                    long long b, e;
                    mi_blur_shard_range(batch_count, g, G, &b, &e);
                    const int n = (int)(e - b);
                    if (g == 0 && opt.verbose) printf("=== Processing Batch %d/%d === (GPU 0 share: %d images)\n", batch + 1, NUM_BATCHES, n);
                    if (n <= 0) continue;
                    const int s = (int)(k % nslots);
                    if (k >= nslots) {                                // the buffer set was last used by batch k - nslots: all its pieces
                        for (int q = 0; q < pieces_of[s] && feed_rc[g] == MI_BLUR_OK; q++) ok(mi_blur_wait_oldest(ctx));
                        if (feed_rc[g] != MI_BLUR_OK) break;
                        if (g == 0 && opt.save.size() && first_output.empty() && k == nslots)
                            first_output.assign(gout[0][0], gout[0][0] + image_size);
                    }
                    int done = 0, pieces = 0;
                    while (done < n && feed_rc[g] == MI_BLUR_OK) {     // create batch image stream (:439-442) and hand it over, piece by piece
                        const int m = std::min(feed_piece, n - done);
                        rep.run(gin[g][s] + (size_t)done * image_size, original_image, image_size, m, !opt.malloc_buffers);
                        ok(mi_blur_submit(ctx, gin[g][s] + (size_t)done * image_size, gout[g][s] + (size_t)done * image_size, m));
                        done += m; pieces++;
                    }
                    pieces_of[s] = pieces;
                    k++;
                }
                ok(mi_blur_sync(ctx, &gpus[g].tm));                     // clFinish (:538-539)
            });
        }
        while (at_gate.load() < G) std::this_thread::yield();
        time_start_total = get_time_ms();
        go.store(true, std::memory_order_release);
        for (auto &t : feeders) t.join();
        for (int g = 0; g < G; g++) mi_check(feed_rc[g], "GPU feeder failed");
        total_images_gpu = NUM_IMAGES;
        if (opt.save.size() && first_output.empty()) first_output.assign(gout[0][0], gout[0][0] + image_size);
    } else if (opt.resident) {
        // one host thread per GPU: a stream of batch-35 launches is issue-rate-bound, so one thread feeding
        // G GPUs in turn would serialise them
        std::vector<std::thread> feeders;
        std::vector<int> feed_rc(G, MI_BLUR_OK);
        for (int g = 0; g < G; g++) {
            long long b, e;
            mi_blur_shard_range(NUM_IMAGES, g, G, &b, &e);
            total_images_gpu += (int)(e - b);
            feeders.emplace_back([&, g, b, e]() {
                mi_blur_bind_thread_to_device(hip_ordinal(g));
                feed_rc[g] = opt.fused ? mi_blur_resident_run_fused(gpus[g].ctx, (int)(e - b), BATCH_SIZE, 1)
                                       : mi_blur_resident_run(gpus[g].ctx, (int)(e - b), BATCH_SIZE, resident_timed_every);
            });
        }
        for (auto &t : feeders) t.join();
        if (opt.verbose) printf("  enqueued after %.2f ms\n", get_time_ms() - time_start_total);
        for (int g = 0; g < G; g++) mi_check(feed_rc[g], "resident run failed");
        for (int g = 0; g < G; g++) {
            mi_check(mi_blur_sync(gpus[g].ctx, &gpus[g].tm), "GPU sync failed");
            uint64_t timed = 0;
            mi_blur_timed_coverage(gpus[g].ctx, &timed, nullptr);
            if (timed && timed < gpus[g].tm.launches) gpus[g].tm.kernel_ms *= (double)gpus[g].tm.launches / (double)timed;
        }
        if (opt.verbose) printf("  synchronised after %.2f ms\n", get_time_ms() - time_start_total);
        if (opt.fused) {
            for (int g = 0; g < G; g++) {
                const int counted = mi_blur_resident_batches_done(gpus[g].ctx);       // negative = status: the poll itself failed
                if (counted < 0) mi_check(counted, "reading the fused stream's batch counters failed");
                printf("%s: one fused dispatch, %d batches counted in\n", gpus[g].name.c_str(), counted);
            }
            if (opt.verbose) printf("  polled after %.2f ms\n", get_time_ms() - time_start_total);
            printf("\n");
        } else
            printf("(kernel time: dispatch timestamps of every %dth launch, scaled to all launches)\n\n", resident_timed_every);
    } else {
        for (int batch = 0; batch < NUM_BATCHES; batch++) {
            if (opt.verbose) printf("=== Processing Batch %d/%d ===\n", batch + 1, NUM_BATCHES);
            const int batch_start = batch * BATCH_SIZE;
            int batch_count = BATCH_SIZE;
            if (batch_start + batch_count > NUM_IMAGES) batch_count = NUM_IMAGES - batch_start;
            const int s = batch % nslots;
            // the buffer set was last used by batch - nslots: wait for exactly those submits
            if (batch >= nslots) {
                // "auto" keeps going after the first batch.  The device-time buckets the reference's recommendation is
                // built on (:713-715) overlap in a pipelined host (H2D of one batch runs beside D2H of another), so they
                // seed the ratio but cannot balance it.  What is balanced is what the host observes: which side it has
                // to WAIT for.  The wait order alternates per batch; time blocked in the second wait is time that side
                // finished after the other, and that fraction of the batch period moves to the other side.
                const bool both_sides = cpu.ctx && !gpus.empty() && cpu.submitted[batch - nslots] && gpus[0].submitted[batch - nslots];
                const bool cpu_first = (batch & 1) == 0;
                double blocked[2] = {0, 0};
                for (int k = 0; k < 2; k++) {
                    const double w0 = get_time_ms();
                    if ((k == 0) == cpu_first) {
                        if (cpu.ctx && cpu.submitted[batch - nslots]) mi_check(mi_blur_wait_oldest(cpu.ctx), "CPU wait failed");
                    } else {
                        for (auto &d : gpus) if (d.submitted[batch - nslots]) mi_check(mi_blur_wait_oldest(d.ctx), "GPU wait failed");
                    }
                    blocked[k] = get_time_ms() - w0;
                }
                if (opt.auto_ratio && mode == 0 && both_sides) {
                    const double now_ms = get_time_ms(), period = last_harvest_ms > 0 ? now_ms - last_harvest_ms : 0;
                    // harvesting a finished GPU batch still costs tens of us (event queries): only a wait well above
                    // that counts as lateness
                    if (period > 0 && blocked[1] > std::max(0.10 * period, 0.05)) {
                        const float shift = 0.25f * (float)std::min(1.0, blocked[1] / period) * (cpu_first ? gpu_ratio : 1.0f - gpu_ratio);
                        gpu_ratio += cpu_first ? -shift : shift;     // second wait was the GPU's: it is the late side
                        gpu_ratio = std::min(0.98f, std::max(0.02f, gpu_ratio));
                        rebalances++;
                        if (opt.verbose) printf("  Rebalanced GPU ratio: %.1f%%\n", gpu_ratio * 100);
                    }
                    last_harvest_ms = now_ms;
                }
                if (opt.save.size() && first_output.empty() && batch - nslots == 0)
                    first_output.assign(batch_output[s], batch_output[s] + image_size);
            }
            // create batch image stream (contiguous) — heterogeneous_blur.c:439-442
            replicate.run(batch_input[s], original_image, image_size, batch_count, mode == 2 && !opt.malloc_buffers);

            int num_images_cpu = 0, num_images_gpu = 0;
            mi_blur_a1_partition(mode, batch_count, gpu_ratio, &num_images_cpu, &num_images_gpu);
            total_images_cpu += num_images_cpu;
            total_images_gpu += num_images_gpu;
            if (opt.verbose) printf("  Batch work distribution: CPU=%d, GPU=%d\n", num_images_cpu, num_images_gpu);

            // images [0, n_cpu) -> CPU device, the rest -> GPUs in contiguous shares (:496)
            if (num_images_cpu > 0) {
                mi_check(mi_blur_submit(cpu.ctx, batch_input[s], batch_output[s], num_images_cpu), "CPU submit failed");
                cpu.submitted[batch] = 1;
            }
            for (int g = 0; g < G && num_images_gpu > 0; g++) {
                long long b, e;
                mi_blur_shard_range(num_images_gpu, g, G, &b, &e);
                if (e <= b) continue;
                const size_t off = (size_t)(num_images_cpu + b) * image_size;
                mi_check(mi_blur_submit(gpus[g].ctx, batch_input[s] + off, batch_output[s] + off, (int)(e - b)), "GPU submit failed");
                gpus[g].submitted[batch] = 1;
            }
            if (opt.verbose) printf("  Batch %d submitted.\n\n", batch + 1);

            // "auto": measure the first batch and apply the reference's own recommendation
            // (optimal = t_cpu_per_image / (t_cpu_per_image + t_gpu_per_image), heterogeneous_blur.c:713-715)
            if (opt.auto_ratio && mode == 0 && batch == 0 && num_images_cpu > 0 && num_images_gpu > 0) {
                mi_check(mi_blur_sync(cpu.ctx, &cpu.tm), "CPU sync failed");
                DeviceTimes tc, tg;
                tc.add(cpu.tm);
                for (auto &d : gpus) { mi_check(mi_blur_sync(d.ctx, &d.tm), "GPU sync failed"); tg.add(d.tm); d.submitted[0] = 0; }
                cpu.submitted[0] = 0;
                const double t_cpu = tc.total() / num_images_cpu, t_gpu = tg.total() / G / num_images_gpu;
                gpu_ratio = (float)(t_cpu / (t_cpu + t_gpu));
                printf("Auto-calibrated GPU ratio: %.1f%% (CPU %.4f ms/image, GPU %.4f ms/image on the first batch)\n\n",
                       gpu_ratio * 100, t_cpu, t_gpu);
            }
        }
        // clFinish on every queue (heterogeneous_blur.c:538-539)
        if (cpu.ctx) mi_check(mi_blur_sync(cpu.ctx, &cpu.tm), "CPU sync failed");
        for (auto &d : gpus) mi_check(mi_blur_sync(d.ctx, &d.tm), "GPU sync failed");
        if (opt.save.size() && first_output.empty()) first_output.assign(batch_output[0], batch_output[0] + image_size);
    }
    const double time_end_total = get_time_ms();
    const double time_total_processing = time_end_total - time_start_total;
    printf("All batches finished!\n\n");
    if (opt.auto_ratio && mode == 0) printf("Auto ratio after %d per-batch updates: %.1f%% GPU\n\n", rebalances, gpu_ratio * 100);
    if (opt.save.size() && !first_output.empty()) {
        save_one_image(opt.save.c_str(), first_output.data(), width, height, channels);
        printf("Saved example output: %s\n\n", opt.save.c_str());
    }

    // ---------------- performance analysis (heterogeneous_blur.c:609-724): sections come from host_common.h
    DeviceTimes tcpu, tgpu;
    tcpu.add(cpu.tm);
    uint64_t gpu_bytes_alg = 0, gpu_launches = 0;
    for (auto &d : gpus) { tgpu.add(d.tm); gpu_bytes_alg += d.tm.bytes_alg; gpu_launches += d.tm.launches; }

    report_header(BATCH_SIZE, time_total_processing, NUM_IMAGES);
    if (total_images_cpu > 0) {
        report_device(2, "CPU", ("processed " + std::to_string(total_images_cpu) + " images").c_str(), tcpu, total_images_cpu);
        printf("\n");
    }
    if (total_images_gpu > 0) {
        report_device(3, "GPU", ("processed " + std::to_string(total_images_gpu) + " images").c_str(), tgpu, total_images_gpu);
        if (G > 1)
            for (int g = 0; g < G; g++)
                printf("   - %s: %llu images, in %.2f / kernel %.2f / out %.2f ms\n", gpus[g].name.c_str(),
                       (unsigned long long)gpus[g].tm.images, gpus[g].tm.h2d_ms, gpus[g].tm.kernel_ms, gpus[g].tm.d2h_ms);
        printf("\n");
    }
    printf("====================\n");
    const bool both_worked = total_images_cpu > 0 && total_images_gpu > 0;
    Comparison cmp;
    if (both_worked) cmp = report_comparison(tcpu, tgpu);
    printf("\n");
    const Throughput thr = report_throughput(NUM_IMAGES, width, height, time_total_processing);

    double optimal_gpu_ratio = 0;
    if (both_worked) {                                   // heterogeneous_blur.c:712-724
        const double t_cpu_per_image = tcpu.total() / total_images_cpu, t_gpu_per_image = tgpu.total() / total_images_gpu;
        optimal_gpu_ratio = t_cpu_per_image / (t_cpu_per_image + t_gpu_per_image);
        printf("8. OPTIMAL RATIO RECOMMENDATION\n");
        printf("   Based on measured performance:\n");
        printf("   CPU: %.3f ms/image\n", t_cpu_per_image);
        printf("   GPU: %.3f ms/image\n", t_gpu_per_image);
        printf("   Recommended GPU ratio: %.1f%%\n", optimal_gpu_ratio * 100);
        printf("   Run with: ./heterogeneous_blur both %.3f\n\n", optimal_gpu_ratio);
    }

    Roofline rf;
    if (total_images_gpu > 0) {
        rf = report_roofline(9, G, gpu_bytes_alg, gpu_launches, tgpu.kernel_ms, total_images_gpu, (opt.resident && !opt.fused) ? nslots : 1);
        if (!opt.resident && tgpu.in_ms > 0 && tgpu.out_ms > 0)
            printf("   Host link: %.1f GB/s in, %.1f GB/s out (sum over GPUs)\n",
                   (double)total_images_gpu * image_size / (tgpu.in_ms / 1000.0) / 1e9 * G,
                   (double)total_images_gpu * image_size / (tgpu.out_ms / 1000.0) / 1e9 * G);
        else if (!opt.resident && tgpu.kernel_ms > 0)
            // pinned batch buffers blurred in place over PCIe (batch server): the kernel bucket IS the transfer, the bound is
            // the host link, not HBM — 56.4 GB/s is the measured one-way DMA rate of this link (profiles/r01_pcie_probe.txt)
            printf("   Host link (buffers blurred in place over PCIe: link-bound, not HBM-bound): %.1f GB/s each way per GPU = %.0f%% of the 56.4 GB/s one-way DMA rate, both directions at once\n",
                   (double)total_images_gpu * image_size / (tgpu.kernel_ms / 1000.0) / 1e9,
                   (double)total_images_gpu * image_size / (tgpu.kernel_ms / 1000.0) / 1e9 / 56.4 * 100.0);
        printf("\n");
    }
    if (!opt.csv.empty())
        append_csv(opt.csv, BATCH_SIZE, mode == 0 ? "both" : mode == 1 ? "cpu" : "gpu", gpu_ratio, NUM_IMAGES, NUM_BATCHES, width, height,
                   time_total_processing, total_images_cpu, tcpu, total_images_gpu, tgpu, cmp, thr, optimal_gpu_ratio, rf, G);

    // ---------------- cleanup (heterogeneous_blur.c:727-747)
    for (int s = 0; s < nslots; s++) { if (batch_input[s]) batch_free(batch_input[s]); if (batch_output[s]) batch_free(batch_output[s]); }
    for (int g = 0; g < G; g++) for (size_t s = 0; s < gin[g].size(); s++) { batch_free(gin[g][s]); batch_free(gout[g][s]); }
    if (cpu.ctx) mi_blur_destroy(cpu.ctx);
    for (auto &d : gpus) mi_blur_destroy(d.ctx);
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// --frames: distinct frames in, blurred frames out.  What the reference does once for its one input image —
// decode through CImg into PLANAR storage, interleave on one host core (heterogeneous_blur.c:106-135), and for the one
// frame it saves the way back (split_image_blur.c:40-56) — becomes a per-frame stage here, so it is built as a pipeline:
//   helper threads (on the GPU's socket)   decode batch k+1 into a pinned PLANAR batch buffer, save batch k-S
//   GPU, inside mi_blur_submit_planar      repack-in kernel reads the planar frames over the host link -> interleaved batch
//                                          in HBM -> blur -> D2H (or repack-out kernel -> planar frames in pinned memory)
// `S` buffer sets rotate per device; batch k goes to GPU k % G.
// ------------------------------------------------------------------------------------------------------------------
static int run_frames(const Options &opt, int mode, int BATCH_SIZE, const std::vector<std::string> &files)
{
    const int N = (int)files.size();
    if (BATCH_SIZE > N) BATCH_SIZE = N;
    const int NB = (N + BATCH_SIZE - 1) / BATCH_SIZE;
    int width = 0, height = 0, channels = 0;
    if (!probe_frame(files[0], width, height, channels)) { printf("Error: cannot read frame %s\n", files[0].c_str()); return -1; }
    const int radius = opt.ksize == 3 ? 1 : 2;
    const size_t image_size = (size_t)width * height * channels;
    printf("Frame geometry (from %s): %dx%d, %d channels, %zu bytes\n", files[0].c_str(), width, height, channels, image_size);
    const int G = mode == 1 ? 1 : opt.gpus;
    if (mode == 2 && !gpus_available(G)) { printf("Error: Could not find %d GPU device(s) (%d visible)\n", G < 1 ? 1 : G, mi_blur_device_count()); return -1; }
    const int per_dev_slots = opt.slots_given ? opt.slots : 3;
    const int S = per_dev_slots * G;
    // Three ways through the device, by what goes in and what is wanted out:
    //   planar in -> interleaved out   mi_blur_submit_planar: GPU repack-in (reads the frames over the host link), blur in HBM, copy out
    //   planar in -> planar out        no repack at all: a planar batch IS a batch of n*C one-channel images (mi_blur_submit on a C = 1
    //                                  context; pinned buffers, so the batch server blurs them in place over PCIe)
    //   --native-layout                the file format is interleaved on disk (PPM/PGM): read straight into the pinned interleaved batch
    //                                  buffer and mi_blur_submit (batch server) — no host de-interleave, no GPU repack
#ifdef MI_BLUR_WITH_CIMG
    const bool native = false;                     // CImg hands every format over planar
    if (opt.native_layout) printf("Note: --native-layout ignored in a CImg build (CImg storage is planar)\n");
#else
    const bool native = opt.native_layout;
#endif
    if (native && opt.planar_out) { printf("Error: --native-layout and --planar-out exclude each other\n"); return -1; }
    const bool planes_as_images = !native && opt.planar_out;
    const int ctx_channels = planes_as_images ? 1 : channels, per_frame = planes_as_images ? channels : 1;
    std::vector<mi_blur_ctx *> ctx(G, nullptr);
    for (int g = 0; g < G; g++) {
        mi_check(mi_blur_create(&ctx[g], mode == 1 ? MI_BLUR_DEVICE_CPU : hip_ordinal(g), width, height, ctx_channels, radius, BATCH_SIZE * per_frame,
                                per_dev_slots, opt.threads), "Failed to create context");
        if (mode == 2) { printf("GPU device: HIP device %d\n", hip_ordinal(g)); report_placement(g, hip_ordinal(g), g == 0); }
        else printf("CPU device: host threads\n");
    }
    printf("Frame path: %s\n", native ? "interleaved on disk -> pinned interleaved batch -> blur in place (no repack anywhere)"
                               : planes_as_images ? "planar frames blurred as one-channel images, in place (no repack anywhere)"
                                                  : "planar frames -> GPU repack-in -> blur -> interleaved out");
    std::vector<uint8_t *> in(S, nullptr), out(S, nullptr);
    for (int s = 0; s < S; s++) {
        const int g = s % G;
        in[s] = (uint8_t *)(mode == 2 ? mi_blur_host_alloc_on(hip_ordinal(g), (size_t)BATCH_SIZE * image_size) : mi_blur_host_alloc((size_t)BATCH_SIZE * image_size));
        out[s] = (uint8_t *)(mode == 2 ? mi_blur_host_alloc_on(hip_ordinal(g), (size_t)BATCH_SIZE * image_size) : mi_blur_host_alloc((size_t)BATCH_SIZE * image_size));
        if (!in[s] || !out[s]) { printf("Error: Failed to allocate batch memory\n"); return -1; }
    }
    if (!opt.save_dir.empty()) mkdir(opt.save_dir.c_str(), 0777);
    TaskPool pool(opt.host_threads, mode == 2 ? hip_ordinal(0) : -1);
    std::vector<std::vector<uint8_t>> scratch(pool.threads());
    std::vector<double> decode_ms(pool.threads(), 0.0), save_ms(pool.threads(), 0.0);
    std::atomic<int> failed{0};
    auto out_name = [&](int i) {
        std::string base = files[i].substr(files[i].find_last_of('/') == std::string::npos ? 0 : files[i].find_last_of('/') + 1);
        const size_t dot = base.rfind('.');
        if (dot != std::string::npos) base.resize(dot);
#ifdef MI_BLUR_WITH_CIMG
        return opt.save_dir + "/" + base + ".bmp";
#else
        return opt.save_dir + "/" + base + (channels == 3 ? ".ppm" : ".pgm");
#endif
    };
    auto batch_range = [&](int k, int &first, int &n) { first = k * BATCH_SIZE; n = std::min(BATCH_SIZE, N - first); };

    // Warm-up outside the clock, as in the other modes: first-use costs (code-object load, the batch server's control
    // blocks and first launch: ~15 ms) would otherwise sit in the first batch of a stream that takes a few tens of ms.
    for (int g = 0; g < G; g++) {
        memset(in[g], 0, image_size);
        if (native || planes_as_images) mi_check(mi_blur_submit(ctx[g], in[g], out[g], per_frame), "warm-up failed");
        else mi_check(mi_blur_submit_planar(ctx[g], in[g], out[g], 1, 0), "warm-up failed");
        mi_check(mi_blur_sync(ctx[g], nullptr), "warm-up failed");
        mi_blur_reset_timing(ctx[g]);
    }
    printf("\nStarting frame stream: %d frames in %d batches of up to %d, %d helper thread(s), %d buffer set(s) per device%s%s\n\n", N, NB, BATCH_SIZE,
           pool.threads(), per_dev_slots, opt.planar_out ? ", planar output" : "", opt.save_dir.empty() ? "" : ", saving every frame");
    const double t0 = get_time_ms();
    double wait_ms = 0, phase_ms = 0, submit_ms = 0;
    for (int k = 0; k < NB + S; k++) {
        const int s = k % S;
        int old_first = 0, old_n = 0, first = 0, n = 0;
        const bool have_old = k >= S && k - S < NB, have_new = k < NB;
        if (have_old) {                      // batch k-S used this buffer set: its output must be in host memory before it is saved / reused
            const double w0 = get_time_ms();
            mi_check(mi_blur_wait_oldest(ctx[(k - S) % G]), "wait failed");
            wait_ms += get_time_ms() - w0;
            batch_range(k - S, old_first, old_n);
        }
        if (have_new) batch_range(k, first, n);
        const int n_save = (have_old && !opt.save_dir.empty()) ? old_n : 0;
        const double p0 = get_time_ms();
        pool.run(n_save + (have_new ? n : 0), [&](int item, int t) {
            const double a = get_time_ms();
            if (item < n_save) {
                if (!save_frame(out_name(old_first + item), out[s] + (size_t)item * image_size, width, height, channels, opt.planar_out, scratch[t])) failed++;
                save_ms[t] += get_time_ms() - a;
            } else {
                const int j = item - n_save;
                if (!(native ? read_pnm_interleaved(files[first + j], in[s] + (size_t)j * image_size, width, height, channels)
                             : decode_frame_planar(files[first + j], in[s] + (size_t)j * image_size, width, height, channels, scratch[t]))) failed++;
                decode_ms[t] += get_time_ms() - a;
            }
        });
        phase_ms += get_time_ms() - p0;
        if (failed.load()) {
            printf("Error: a frame could not be read (or differs from %dx%dx%d) or written\n", width, height, channels);
            for (auto c : ctx) { (void)mi_blur_sync(c, nullptr); mi_blur_destroy(c); }        // work already submitted drains first
            for (int q = 0; q < S; q++) { mi_blur_host_free(in[q]); mi_blur_host_free(out[q]); }
            return -1;
        }
        if (have_new) {
            if (opt.verbose) printf("  Batch %d/%d: frames %d-%d -> device %d\n", k + 1, NB, first, first + n - 1, k % G);
            const double s0 = get_time_ms();
            if (native || planes_as_images) mi_check(mi_blur_submit(ctx[k % G], in[s], out[s], n * per_frame), "submit failed");
            else mi_check(mi_blur_submit_planar(ctx[k % G], in[s], out[s], n, 0), "submit failed");
            submit_ms += get_time_ms() - s0;
            if (opt.verbose) printf("    submit %d took %.3f ms\n", k, get_time_ms() - s0);
        }
    }
    DeviceTimes td;
    uint64_t bytes_alg = 0, launches = 0;
    for (int g = 0; g < G; g++) { mi_blur_timing tm{}; mi_check(mi_blur_sync(ctx[g], &tm), "sync failed"); td.add(tm); bytes_alg += tm.bytes_alg; launches += tm.launches; }
    const double wall = get_time_ms() - t0;
    printf("All batches finished!\n\n");
    double dsum = 0, ssum = 0;
    for (double v : decode_ms) dsum += v;
    for (double v : save_ms) ssum += v;
    report_header(BATCH_SIZE, wall, N);
    report_device(mode == 1 ? 2 : 3, mode == 1 ? "CPU" : "GPU", ("processed " + std::to_string(N) + " frames").c_str(), td, N);
    printf("\n====================\n\n");
    const Throughput thr = report_throughput(N, width, height, wall);
    printf("10. FRAME INGEST (distinct frames; ingest-inclusive figures)\n");
    printf("   Frames: %d files, %s by %d helper thread(s)\n", N, native ? "read into pinned INTERLEAVED batch buffers" : "decoded into pinned PLANAR batch buffers", pool.threads());
    printf("   Decode:            %.2f ms of helper-thread time (%.4f ms per frame)\n", dsum, dsum / N);
    if (mode == 2 && (native || planes_as_images)) {
        printf("   GPU blur in place: %.2f ms (the batch server reads and writes the pinned frames over the host link)\n", td.kernel_ms);
    } else if (mode == 2) {
        printf("   GPU repack-in:     %.2f ms (planar frames read over the host link, interleaved batch written to HBM)\n", td.in_ms);
        printf("   GPU blur:          %.2f ms\n", td.kernel_ms);
        printf("   GPU copy-out:      %.2f ms\n", td.out_ms);
    } else {
        printf("   CPU repack + blur: %.2f ms\n", td.kernel_ms);
    }
    if (!opt.save_dir.empty()) printf("   Save:              %.2f ms of helper-thread time (%.4f ms per frame) -> %s\n", ssum, ssum / N, opt.save_dir.c_str());
    printf("   Host blocked waiting for a device: %.2f ms of %.2f ms wall\n", wait_ms, wall);
    printf("   Feeder thread: %.2f ms in decode/save phases (with the helpers), %.2f ms in submits\n", phase_ms, submit_ms);
    printf("   Ingest-inclusive throughput: %.2f images/sec (%.2f Megapixels/sec)\n\n", thr.img_s, thr.mpix);
    Roofline rf;
    if (mode == 2) { rf = report_roofline(9, G, bytes_alg, launches, td.kernel_ms, N); printf("\n"); }
    if (!opt.csv.empty()) {
        DeviceTimes none;
        Comparison cmp;
        append_csv(opt.csv, BATCH_SIZE, mode == 1 ? "cpu-frames" : "gpu-frames", mode == 1 ? 0.f : 1.f, N, NB, width, height, wall, mode == 1 ? N : 0,
                   mode == 1 ? td : none, mode == 2 ? N : 0, mode == 2 ? td : none, cmp, thr, 0.0, rf, mode == 2 ? G : 0);
    }
    for (int s = 0; s < S; s++) { mi_blur_host_free(in[s]); mi_blur_host_free(out[s]); }
    for (auto c : ctx) mi_blur_destroy(c);
    return 0;
}
