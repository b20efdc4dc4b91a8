// split_image_blur — Approach 2 (split-image + halo) host, MI355X-native.
//
//   split_image_blur [gpu_ratio] [batch]  [--image F | --synthetic | --size WxH] [--channels C] [--ksize 3|5]
//                    [--images N] [--gpus G] [--slots S] [--threads T] [--verbose] [--csv FILE] [--save FILE]
//   split_image_blur --resident [--gpus G] [--size WxH] [--ksize 3|5] [--iters N] [--iterate] [--overlap]
//                    [--transport rccl|p2p|pull|peer] [--save FILE]
//
// Default mode keeps the reference host's semantics (split_image_blur.c:62-102 CLI, :142-173
// geometry, :441-607 batch loop, :615-721 report): every image is split at
// split_row = (int)(H*(1-gpu_ratio)) clamped to [HALO, H-HALO]; the CPU device blurs rows
// [0, split_row) from input rows [0, split_row+HALO), the GPU device(s) blur rows [split_row, H) from
// input rows [split_row-HALO, H); each device runs the kernel on its own band with the band height as
// `height` and the halo output rows are dropped (:401,414,526,537).  HALO = blur radius (1 for the
// reference 3x3, :70).  Differences: the `cpu` device is native host threads; with --gpus G the GPU
// rows are shared out over G GPUs (each with its own halos); a batch is ONE 2-D DMA in, ONE launch and
// ONE 2-D DMA out per device (mi_blur_submit_bands) instead of a Write/NDRange/Read per image (:520-541).
//
// --resident is the multi-GPU form of the same idea for ONE large image (BASELINE config 5): the image
// is row-sharded once into the GPUs' HBM; every iteration exchanges the `radius` boundary rows between
// neighbouring GPUs with RCCL send/recv over xGMI (the reference re-uploads overlapping slices from the
// host instead, :516,520,530) and blurs each shard.
#include "host_common.h"

using namespace host;

struct Part {                      // one device's share of every image
    mi_blur_ctx *ctx = nullptr;
    std::string name;
    int in_row0 = 0, band_rows = 0, halo_top = 0, halo_bottom = 0, out_row0 = 0, out_rows = 0;
    mi_blur_timing tm{};
};

static int run_resident(const Options &opt);

int main(int argc, char **argv)
{
    // ---------------- configuration (split_image_blur.c:62-102)
    const char *input_filename = "./image_320x240.jpg";
    int BATCH_SIZE = 500;
    const int local_work_size = 16;
    float gpu_ratio = 0.5f;

    Options opt;
    const int npos = parse_flags(argc, argv, opt);
    if (opt.resident) return run_resident(opt);
    const int NUM_IMAGES = opt.images;
    const int HALO = opt.ksize == 3 ? 1 : 2;

    if (npos > 1) {
        gpu_ratio = atof(argv[1]);
        if (gpu_ratio < 0.0f || gpu_ratio > 1.0f) {
            printf("Warning: gpu_ratio must be between 0.0 and 1.0. Using 0.5\n");
            gpu_ratio = 0.5f;
        }
    }
    if (npos > 2) {
        BATCH_SIZE = atoi(argv[2]);
        if (BATCH_SIZE < 1 || BATCH_SIZE > NUM_IMAGES) {
            printf("Warning: BATCH_SIZE must be between 1 and %d. Using 500\n", NUM_IMAGES);
            BATCH_SIZE = 500;
        }
    }
    if (BATCH_SIZE > NUM_IMAGES) BATCH_SIZE = NUM_IMAGES;
    const int NUM_BATCHES = (NUM_IMAGES + BATCH_SIZE - 1) / BATCH_SIZE;
    if (!opt.image.empty()) input_filename = opt.image.c_str();

    printf("========== SPLIT-IMAGE CONFIGURATION ==========\n");
    printf("Input file: %s\n", opt.synthetic ? "(synthetic)" : input_filename);
    printf("Number of images in stream: %d\n", NUM_IMAGES);
    printf("Batch size: %d images\n", BATCH_SIZE);
    printf("Number of batches: %d\n", NUM_BATCHES);
    printf("Work-group size: %dx%d\n", local_work_size, local_work_size);
    printf("GPU ratio: %.1f%% (rows to GPU)\n", gpu_ratio * 100);
    printf("Halo size: %d row(s)\n", HALO);
    printf("================================================\n\n");

    // ---------------- load original image (split_image_blur.c:106-139)
    Image img = load_image(input_filename, opt.syn_w, opt.syn_h, opt.syn_c, opt.synthetic);
    if (!opt.save_input.empty()) save_one_image(opt.save_input.c_str(), img.px.data(), img.width, img.height, img.channels);
    const int width = img.width, height = img.height, channels = img.channels;
    printf("Original image loaded: %dx%d, %d channels\n", width, height, channels);
    const size_t pitch = (size_t)width * channels, image_size = pitch * height;
    printf("Size of one image: %zu bytes (%.2f KB)\n", image_size, image_size / 1024.0);
    printf("Original image source: %s\n\n", img.source.c_str());
    const uint8_t *original_image = img.px.data();
    if (height < 2 * HALO + 1) { printf("Error: image too small for a %d-row halo split\n", HALO); return -1; }

    // ---------------- split dimensions (split_image_blur.c:142-173)
    mi_blur_a2_geometry geo;
    {
        int raw = (int)(height * (1.0f - gpu_ratio));
        if (raw < HALO) printf("Warning: split_row too small, adjusting to %d\n", HALO);
        if (raw > height - HALO) printf("Warning: split_row too large, adjusting to %d\n", height - HALO);
    }
    mi_blur_a2_split(height, gpu_ratio, HALO, &geo);
    const int split_row = geo.split_row;
    printf("Split configuration:\n");
    printf("  Split row: %d (CPU: rows 0-%d, GPU: rows %d-%d)\n", split_row, split_row - 1, split_row, height - 1);
    printf("  CPU: %d input rows (inc. halo), %d output rows\n", geo.cpu_input_rows, geo.cpu_output_rows);
    printf("  GPU: %d input rows (inc. halo), %d output rows\n", geo.gpu_input_rows, geo.gpu_output_rows);
    printf("  CPU input size: %.2f KB, output size: %.2f KB\n", pitch * geo.cpu_input_rows / 1024.0, pitch * geo.cpu_output_rows / 1024.0);
    printf("  GPU input size: %.2f KB, output size: %.2f KB\n\n", pitch * geo.gpu_input_rows / 1024.0, pitch * geo.gpu_output_rows / 1024.0);

    // ---------------- devices (split_image_blur.c:175-247): the reference needs BOTH devices
    const int G = opt.gpus;
    if (!gpus_available(G)) {
        printf("Error: Could not find both CPU and GPU devices (%d GPU(s) visible, %d asked)\n", mi_blur_device_count(), G);
        return -1;
    }
    const int nslots = opt.slots_given ? opt.slots : 3;      // 3 sets in flight: +12 % over 2, steadier than 4 (profiles/r03_hosts_e2e.txt)
    Part cpu;
    cpu.in_row0 = 0; cpu.band_rows = geo.cpu_input_rows; cpu.halo_top = 0; cpu.halo_bottom = HALO;
    cpu.out_row0 = 0; cpu.out_rows = geo.cpu_output_rows;
    mi_check(mi_blur_create(&cpu.ctx, MI_BLUR_DEVICE_CPU, width, height, channels, HALO, BATCH_SIZE, nslots, opt.threads),
             "Failed to create CPU context");
    cpu.name = "host threads";
    printf("CPU device: %s\n", cpu.name.c_str());
    std::vector<Part> gpus(G);
    for (int g = 0; g < G; g++) {
        // GPU rows [split_row, H) shared out evenly; every share carries HALO rows from its neighbours
        Part &p = gpus[g];
        long long b, e;
        mi_blur_shard_range(geo.gpu_output_rows, g, G, &b, &e);
        p.out_row0 = split_row + (int)b; p.out_rows = (int)(e - b);
        p.halo_top = HALO;                                             // row split_row-HALO.. exists (split_row >= HALO)
        p.halo_bottom = std::min(HALO, height - (p.out_row0 + p.out_rows));
        p.in_row0 = p.out_row0 - p.halo_top;
        p.band_rows = p.out_rows + p.halo_top + p.halo_bottom;
        if (p.out_rows <= 0) { printf("Error: more GPUs than GPU rows\n"); return -1; }
        mi_check(mi_blur_create(&p.ctx, hip_ordinal(g), width, height, channels, HALO, BATCH_SIZE, nslots, 0), "Failed to create GPU context");
        p.name = "HIP device " + std::to_string(hip_ordinal(g));
        printf("GPU device: %s (rows %d-%d)\n", p.name.c_str(), p.out_row0, p.out_row0 + p.out_rows - 1);
    }
    // every device reads the SAME batch buffer here (one image split by rows, split_image_blur.c:469-480), built by this
    // one thread: it and its helpers keep to GPU 0's socket, where mi_blur_host_alloc pins the buffer
    for (int g = 0; g < G; g++) report_placement(g, hip_ordinal(g), g == 0);
    printf("\nKernel objects created\n\n");

    printf("Allocating device buffers...\n");
    std::vector<uint8_t *> batch_input(nslots), batch_output(nslots);
    for (int s = 0; s < nslots; s++) {
        batch_input[s] = (uint8_t *)mi_blur_host_alloc((size_t)BATCH_SIZE * image_size);
        batch_output[s] = (uint8_t *)mi_blur_host_alloc((size_t)BATCH_SIZE * image_size);
        if (!batch_input[s] || !batch_output[s]) { printf("Error: Failed to allocate batch memory\n"); return -1; }
    }
    printf("Device buffers allocated\n\n");
    printf("CPU global size: %d x %d\n", (width + 15) / 16 * 16, (geo.cpu_input_rows + 15) / 16 * 16);
    printf("GPU global size: %d x %d\n", (width + 15) / 16 * 16, (geo.gpu_input_rows + 15) / 16 * 16);
    printf("Local size: %d x %d\n\n", local_work_size, local_work_size);

    std::vector<uint8_t> first_output;
    Replicator replicate(opt.host_threads, hip_ordinal(0));
    // Warm-up outside the clock, as in heterogeneous_blur: first-use costs — the kernels' code object, the batch server's
    // control blocks, the CPU device's worker threads, ~25 ms together — belong to set-up (the reference's clock starts after
    // clBuildProgram / clCreateKernel too, split_image_blur.c:441), not to the first batch
    {
        const int nw = std::min(BATCH_SIZE, 4);
        replicate.run(batch_input[0], original_image, image_size, nw);
        mi_check(mi_blur_submit_bands(cpu.ctx, batch_input[0] + (size_t)cpu.in_row0 * pitch, batch_output[0] + (size_t)cpu.out_row0 * pitch,
                                      nw, image_size, cpu.band_rows, cpu.halo_top, cpu.halo_bottom), "CPU warm-up failed");
        for (auto &p : gpus)
            mi_check(mi_blur_submit_bands(p.ctx, batch_input[0] + (size_t)p.in_row0 * pitch, batch_output[0] + (size_t)p.out_row0 * pitch,
                                          nw, image_size, p.band_rows, p.halo_top, p.halo_bottom), "GPU warm-up failed");
        mi_check(mi_blur_sync(cpu.ctx, nullptr), "CPU sync failed"); mi_blur_reset_timing(cpu.ctx);
        for (auto &p : gpus) { mi_check(mi_blur_sync(p.ctx, nullptr), "GPU sync failed"); mi_blur_reset_timing(p.ctx); }
    }

    // ---------------- batch processing (split_image_blur.c:441-607)
    printf("Starting batch processing of %d images in %d batches...\n\n", NUM_IMAGES, NUM_BATCHES);
    const double time_start_total = get_time_ms();
    for (int batch = 0; batch < NUM_BATCHES; batch++) {
        if (opt.verbose) printf("=== Processing Batch %d/%d ===\n", batch + 1, NUM_BATCHES);
        const int batch_start = batch * BATCH_SIZE;
        int batch_count = BATCH_SIZE;
        if (batch_start + batch_count > NUM_IMAGES) batch_count = NUM_IMAGES - batch_start;
        const int s = batch % nslots;
        if (batch >= nslots) {                      // every device took part in every batch
            mi_check(mi_blur_wait_oldest(cpu.ctx), "CPU wait failed");
            for (auto &p : gpus) mi_check(mi_blur_wait_oldest(p.ctx), "GPU wait failed");
            if (opt.save.size() && first_output.empty() && batch - nslots == 0)
                first_output.assign(batch_output[s], batch_output[s] + image_size);
        }
        replicate.run(batch_input[s], original_image, image_size, batch_count);
        if (opt.verbose) printf("  Processing %d images (each split between CPU and GPU)\n", batch_count);

        // CPU: top rows of every image; GPU(s): bottom rows (split_image_blur.c:511-541), batched
        mi_check(mi_blur_submit_bands(cpu.ctx, batch_input[s] + (size_t)cpu.in_row0 * pitch, batch_output[s] + (size_t)cpu.out_row0 * pitch,
                                      batch_count, image_size, cpu.band_rows, cpu.halo_top, cpu.halo_bottom), "CPU submit failed");
        for (auto &p : gpus)
            mi_check(mi_blur_submit_bands(p.ctx, batch_input[s] + (size_t)p.in_row0 * pitch, batch_output[s] + (size_t)p.out_row0 * pitch,
                                          batch_count, image_size, p.band_rows, p.halo_top, p.halo_bottom), "GPU submit failed");
        if (opt.verbose) printf("  Batch %d submitted.\n\n", batch + 1);
    }
    mi_check(mi_blur_sync(cpu.ctx, &cpu.tm), "CPU sync failed");
    for (auto &p : gpus) mi_check(mi_blur_sync(p.ctx, &p.tm), "GPU sync failed");
    if (opt.save.size() && first_output.empty()) first_output.assign(batch_output[0], batch_output[0] + image_size);
    const double time_end_total = get_time_ms();
    const double time_total_processing = time_end_total - time_start_total;
    printf("All batches finished!\n\n");
    if (opt.save.size()) {
        save_one_image(opt.save.c_str(), first_output.data(), width, height, channels);
        printf("Saved example output: %s\n\n", opt.save.c_str());
    }

    // ---------------- performance analysis (split_image_blur.c:615-721): sections come from host_common.h
    DeviceTimes tcpu, tgpu;
    tcpu.add(cpu.tm);
    uint64_t gpu_bytes_alg = 0, gpu_launches = 0;
    for (auto &p : gpus) { tgpu.add(p.tm); gpu_bytes_alg += p.tm.bytes_alg; gpu_launches += p.tm.launches; }
    const int cpu_output_rows = geo.cpu_output_rows, gpu_output_rows = geo.gpu_output_rows;

    report_header(BATCH_SIZE, time_total_processing, NUM_IMAGES);
    report_device(2, "CPU", ("processed " + std::to_string(NUM_IMAGES) + " images - top " + std::to_string(cpu_output_rows) + " rows each").c_str(), tcpu, 0);
    printf("\n");
    report_device(3, "GPU", ("processed " + std::to_string(NUM_IMAGES) + " images - bottom " + std::to_string(gpu_output_rows) + " rows each").c_str(), tgpu, 0);
    if (G > 1)
        for (auto &p : gpus)
            printf("   - %s: rows %d-%d, in %.2f / kernel %.2f / out %.2f ms\n", p.name.c_str(), p.out_row0,
                   p.out_row0 + p.out_rows - 1, p.tm.h2d_ms, p.tm.kernel_ms, p.tm.d2h_ms);
    printf("\n============================\n");
    const Comparison cmp = report_comparison(tcpu, tgpu);
    printf("\n");
    const Throughput thr = report_throughput(NUM_IMAGES, width, height, time_total_processing);

    printf("8. SPLIT-IMAGE STATISTICS\n");                   // split_image_blur.c:703-709
    printf("   CPU time per image: %.3f ms (for %d rows)\n", tcpu.total() / NUM_IMAGES, cpu_output_rows);
    printf("   GPU time per image: %.3f ms (for %d rows)\n", tgpu.total() / NUM_IMAGES, gpu_output_rows);
    printf("   Combined time per image: %.3f ms\n", time_total_processing / NUM_IMAGES);
    printf("   Current GPU ratio: %.1f%%\n\n", gpu_ratio * 100);

    const double cpu_time_per_row = tcpu.total() / ((double)NUM_IMAGES * cpu_output_rows);   // :711-720
    const double gpu_time_per_row = tgpu.total() / ((double)NUM_IMAGES * gpu_output_rows);
    const double optimal_gpu_ratio = cpu_time_per_row / (cpu_time_per_row + gpu_time_per_row);
    printf("9. OPTIMAL RATIO RECOMMENDATION\n");
    printf("   CPU: %.5f ms/row\n", cpu_time_per_row);
    printf("   GPU: %.5f ms/row\n", gpu_time_per_row);
    printf("   Recommended GPU ratio: %.1f%%\n", optimal_gpu_ratio * 100);
    printf("   Run with: ./split_image_blur %.3f\n\n", optimal_gpu_ratio);

    const Roofline rf = report_roofline(10, G, gpu_bytes_alg, gpu_launches, tgpu.kernel_ms, NUM_IMAGES);
    printf("\n");
    if (!opt.csv.empty())
        append_csv(opt.csv, BATCH_SIZE, "split", gpu_ratio, NUM_IMAGES, NUM_BATCHES, width, height, time_total_processing,
                   NUM_IMAGES, tcpu, NUM_IMAGES, tgpu, cmp, thr, optimal_gpu_ratio, rf, G);

    for (int s = 0; s < nslots; s++) { mi_blur_host_free(batch_input[s]); mi_blur_host_free(batch_output[s]); }
    mi_blur_destroy(cpu.ctx);
    for (auto &p : gpus) mi_blur_destroy(p.ctx);
    return 0;
}

// ------------------------------------------------------------------------------------------------
// --resident: ONE large image, row shards resident in G GPUs, RCCL halo exchange per iteration.
// Uses the HIP runtime directly only for device memory and streams (hipMalloc/hipMemcpy/hipStream);
// kernels and the exchange go through the C ABI.
// ------------------------------------------------------------------------------------------------
#include <hip/hip_runtime.h>

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%d - %s\n", (int)e_, hipGetErrorString(e_)); exit(-1); } } while (0)

static int run_resident(const Options &opt)
{
    const int G = opt.gpus, radius = opt.ksize == 3 ? 1 : 2;
    const int W = opt.size_given ? opt.syn_w : 8192, H = opt.size_given ? opt.syn_h : 8192, C = opt.syn_c;
    const size_t pitch = (size_t)W * C;
    if (!gpus_available(G)) { printf("Error: %d GPU(s) asked, %d visible\n", G, mi_blur_device_count()); return -1; }
    // peer: no exchange at all — the band kernel reads the neighbours' rows where they lie (mi_blur_enqueue_band_peer); the
    // communicator of the pull transport is still made, for the peer access it enables between the devices
    const bool peer = opt.transport == "peer";
    const bool pull = opt.transport == "pull" || peer;
    const bool p2p = opt.transport == "p2p" || (virtual_gpus() && !pull);      // RCCL refuses two ranks on one device
    if (peer && (pitch % 16 || C > 4)) { printf("Error: --transport peer needs rows of whole 16-byte chunks and 1-4 channels\n"); return -1; }
    if (H / G < radius) { printf("Error: shards of %d rows are thinner than the halo\n", H / G); return -1; }
    printf("========== SPLIT-IMAGE (RESIDENT, MULTI-GPU) ==========\n");
    printf("Image: %dx%d, %d channels (%.2f MB), %dx%d blur, %d GPU(s), %d iterations\n", W, H, C, pitch * H / 1e6, opt.ksize,
           opt.ksize, G, opt.iters);
    printf("Halo: %d row(s) = %zu bytes per neighbour per direction, %s\n\n", radius, radius * pitch,
           peer ? "read in place by the band kernel (peer reads, no exchange step)" : pull ? "pulled by one kernel per GPU (peer reads)" : p2p ? "hipMemcpyPeerAsync pushes" : "RCCL send/recv");

    std::vector<uint8_t> image(pitch * H);
    mi_blur_fill_synthetic(image.data(), W, H, C, 0, 1, 0);
    std::vector<mi_blur_band> band(G);
    std::vector<uint8_t *> d_band(G), d_out(G);
    std::vector<hipStream_t> stream(G);
    std::vector<int> owned(G), devs(G);
    for (int g = 0; g < G; g++) {
        devs[g] = hip_ordinal(g);
        mi_blur_band_of(H, radius, g, G, &band[g]);
        owned[g] = band[g].row_end - band[g].row_begin;
        const size_t rows = owned[g] + band[g].halo_top + band[g].halo_bottom;
        HIP_OK(hipSetDevice(devs[g]));
        HIP_OK(hipStreamCreateWithFlags(&stream[g], hipStreamNonBlocking));
        HIP_OK(hipMalloc((void **)&d_band[g], rows * pitch));
        HIP_OK(hipMalloc((void **)&d_out[g], (size_t)owned[g] * pitch));
        HIP_OK(hipMemset(d_band[g], peer ? 0xA5 : 0, rows * pitch));     // peer: the shard's own halo rows stay this poison, unread
        // upload OWNED rows only: the halo rows arrive from the neighbours over xGMI
        HIP_OK(hipMemcpy(d_band[g] + (size_t)band[g].halo_top * pitch, image.data() + (size_t)band[g].row_begin * pitch,
                         (size_t)owned[g] * pitch, hipMemcpyHostToDevice));
        printf("GPU %d: rows %d-%d (+%d/+%d halo)\n", g, band[g].row_begin, band[g].row_end - 1, band[g].halo_top, band[g].halo_bottom);
    }
    std::vector<mi_blur_comm *> comm(G, nullptr);
    mi_check(pull ? mi_blur_comm_init_pull(comm.data(), G, devs.data())
                  : p2p ? mi_blur_comm_init_p2p(comm.data(), G, devs.data()) : mi_blur_comm_init_all(comm.data(), G, devs.data()),
             "communicator init failed");

    // --iterate: k successive blurs of the resident image (output shard -> next input shard), the case where the
    // halo exchange genuinely recurs.  Without it every iteration blurs the same input again.
    std::vector<uint8_t *> d_band2(G, nullptr);
    if (opt.iterate)
        for (int g = 0; g < G; g++) {
            HIP_OK(hipSetDevice(devs[g]));
            const size_t rows = owned[g] + band[g].halo_top + band[g].halo_bottom;
            HIP_OK(hipMalloc((void **)&d_band2[g], rows * pitch));
            HIP_OK(hipMemset(d_band2[g], peer ? 0xA5 : 0, rows * pitch));
        }
    auto upload = [&]() {
        for (int g = 0; g < G; g++) {
            HIP_OK(hipSetDevice(devs[g]));
            HIP_OK(hipMemcpy(d_band[g] + (size_t)band[g].halo_top * pitch, image.data() + (size_t)band[g].row_begin * pitch,
                             (size_t)owned[g] * pitch, hipMemcpyHostToDevice));
        }
    };
    // --overlap: the exchange runs on its own stream per GPU while the compute stream blurs the interior rows
    // [R, owned-R) of the shard, which need no halo; the 2R edge rows are blurred once the halos have arrived.  Events:
    // ev_done[g] = GPU g finished the previous step (its edge rows are final, its halo rows are free to overwrite),
    // ev_halo[g] = this step's halos of GPU g are in place.
    std::vector<hipStream_t> xstream(G, nullptr);
    std::vector<hipEvent_t> ev_done(G, nullptr), ev_halo(G, nullptr);
    bool can_overlap = G > 1 && !peer;                   // nothing to overlap when there is no exchange
    for (int g = 0; g < G && can_overlap; g++) if (owned[g] <= 2 * radius) can_overlap = false;     // no interior to hide behind
    bool overlap = opt.overlap && can_overlap;
    if (can_overlap)      // set up even when not asked for: the report times both forms side by side (below)
        for (int g = 0; g < G; g++) {
            HIP_OK(hipSetDevice(devs[g]));
            HIP_OK(hipStreamCreateWithFlags(&xstream[g], hipStreamNonBlocking));
            HIP_OK(hipEventCreateWithFlags(&ev_done[g], hipEventDisableTiming));
            HIP_OK(hipEventCreateWithFlags(&ev_halo[g], hipEventDisableTiming));
            HIP_OK(hipEventRecord(ev_done[g], stream[g]));
        }
    if (opt.overlap) printf("Overlap: %s\n", overlap ? "halo exchange hidden behind the interior rows" : "not applicable (one GPU or shards without interior)");
    auto step_overlapped = [&]() {
        std::vector<void *> xs(G);
        for (int g = 0; g < G; g++) {
            xs[g] = xstream[g];
            HIP_OK(hipSetDevice(devs[g]));
            HIP_OK(hipStreamWaitEvent(xstream[g], ev_done[g], 0));
        }
        mi_check(mi_blur_halo_exchange_all(comm.data(), G, d_band.data(), W, C, owned.data(), radius, xs.data()), "halo exchange failed");
        for (int g = 0; g < G; g++) {
            HIP_OK(hipSetDevice(devs[g]));
            HIP_OK(hipEventRecord(ev_halo[g], xstream[g]));
            const int ht = band[g].halo_top, rows = owned[g] + ht + band[g].halo_bottom;
            uint8_t *dst = opt.iterate ? d_band2[g] + (size_t)ht * pitch : d_out[g];
            // interior first (reads no halo row), then — after the halos — the edges.  An edge that is the image's own
            // edge (no neighbour) needs no halo either, but keeping the two-phase form per GPU keeps the code uniform.
            mi_check(mi_blur_enqueue_band(d_band[g], dst + (size_t)radius * pitch, W, rows, C, radius, ht + radius, ht + owned[g] - radius,
                                          stream[g]), "interior launch failed");
            HIP_OK(hipStreamWaitEvent(stream[g], ev_halo[g], 0));
            mi_check(mi_blur_enqueue_band(d_band[g], dst, W, rows, C, radius, ht, ht + radius, stream[g]), "top edge launch failed");
            mi_check(mi_blur_enqueue_band(d_band[g], dst + (size_t)(owned[g] - radius) * pitch, W, rows, C, radius,
                                          ht + owned[g] - radius, ht + owned[g], stream[g]), "bottom edge launch failed");
            HIP_OK(hipEventRecord(ev_done[g], stream[g]));
        }
        if (opt.iterate) std::swap(d_band, d_band2);
    };
    // peer transport: one launch per GPU per step.  Only an iterated blur needs ordering between the GPUs (a shard's rows
    // change from step to step): a GPU starts step i+1 once both neighbours have finished step i — their new rows are final,
    // and they are done reading the rows this GPU is about to overwrite.
    std::vector<hipEvent_t> ev_step(G, nullptr);
    if (peer)
        for (int g = 0; g < G; g++) { HIP_OK(hipSetDevice(devs[g])); HIP_OK(hipEventCreateWithFlags(&ev_step[g], hipEventDisableTiming)); }
    auto step_peer = [&]() {
        if (opt.iterate)
            for (int g = 0; g < G; g++) { HIP_OK(hipSetDevice(devs[g])); HIP_OK(hipEventRecord(ev_step[g], stream[g])); }
        for (int g = 0; g < G; g++) {
            HIP_OK(hipSetDevice(devs[g]));
            const uint8_t *top = nullptr, *bottom = nullptr;
            if (g > 0) {
                if (opt.iterate) HIP_OK(hipStreamWaitEvent(stream[g], ev_step[g - 1], 0));
                top = d_band[g - 1] + (size_t)(band[g - 1].halo_top + owned[g - 1] - radius) * pitch;
            }
            if (g < G - 1) {
                if (opt.iterate) HIP_OK(hipStreamWaitEvent(stream[g], ev_step[g + 1], 0));
                bottom = d_band[g + 1] + (size_t)band[g + 1].halo_top * pitch;
            }
            uint8_t *dst = opt.iterate ? d_band2[g] + (size_t)band[g].halo_top * pitch : d_out[g];
            mi_check(mi_blur_enqueue_band_peer(d_band[g], dst, W, owned[g] + band[g].halo_top + band[g].halo_bottom, C, radius,
                                               band[g].halo_top, band[g].halo_top + owned[g], top, bottom, stream[g]), "band launch failed");
        }
        if (opt.iterate) std::swap(d_band, d_band2);
    };
    auto step = [&]() {
        if (peer) { step_peer(); return; }
        if (overlap) { step_overlapped(); return; }
        std::vector<void *> st(G);
        for (int g = 0; g < G; g++) st[g] = stream[g];
        mi_check(mi_blur_halo_exchange_all(comm.data(), G, d_band.data(), W, C, owned.data(), radius, st.data()), "halo exchange failed");
        for (int g = 0; g < G; g++) {
            HIP_OK(hipSetDevice(devs[g]));
            uint8_t *dst = opt.iterate ? d_band2[g] + (size_t)band[g].halo_top * pitch : d_out[g];
            mi_check(mi_blur_enqueue_band(d_band[g], dst, W, owned[g] + band[g].halo_top + band[g].halo_bottom, C, radius,
                                          band[g].halo_top, band[g].halo_top + owned[g], stream[g]), "band launch failed");
        }
        if (opt.iterate) std::swap(d_band, d_band2);     // the blurred shard is the next iteration's input
    };
    auto sync_all = [&]() {
        for (int g = 0; g < G; g++) {
            HIP_OK(hipSetDevice(devs[g]));
            HIP_OK(hipStreamSynchronize(stream[g]));
            if (xstream[g]) HIP_OK(hipStreamSynchronize(xstream[g]));
        }
    };
    // warm-up: first RCCL connection set-up, then ~60 ms of steps — after any idle gap the GPU needs ~40 ms of work to
    // ramp its clocks (profiles/r02_clock_ramp.txt), and this mode exists to measure the step, not the ramp
    {
        const double warm_until = get_time_ms() + 60.0;
        do { for (int i = 0; i < 8; i++) step(); sync_all(); } while (get_time_ms() < warm_until);
    }
    if (opt.iterate) upload();                           // start the timed chain from the original image again
    const double t0 = get_time_ms();
    for (int i = 0; i < opt.iters; i++) step();
    sync_all();
    const double ms = get_time_ms() - t0;

    // verify against the single-device result on GPU 0 (one blur, or `iters` successive blurs with --iterate)
    std::vector<uint8_t> got(pitch * H), want(pitch * H);
    for (int g = 0; g < G; g++) {
        HIP_OK(hipSetDevice(devs[g]));
        const uint8_t *src = opt.iterate ? d_band[g] + (size_t)band[g].halo_top * pitch : d_out[g];
        HIP_OK(hipMemcpy(got.data() + (size_t)band[g].row_begin * pitch, src, (size_t)owned[g] * pitch, hipMemcpyDeviceToHost));
    }
    {
        HIP_OK(hipSetDevice(0));
        uint8_t *di, *dout;
        HIP_OK(hipMalloc((void **)&di, pitch * H)); HIP_OK(hipMalloc((void **)&dout, pitch * H));
        HIP_OK(hipMemcpy(di, image.data(), pitch * H, hipMemcpyHostToDevice));
        const int passes = opt.iterate ? opt.iters : 1;
        for (int i = 0; i < passes; i++) {
            mi_check(mi_blur_enqueue(di, dout, W, H, C, radius, 1, nullptr), "whole-image launch failed");
            std::swap(di, dout);
        }
        HIP_OK(hipDeviceSynchronize());
        HIP_OK(hipMemcpy(want.data(), di, pitch * H, hipMemcpyDeviceToHost));
        HIP_OK(hipFree(di)); HIP_OK(hipFree(dout));
    }
    if (!opt.save.empty()) save_one_image(opt.save.c_str(), got.data(), W, H, C);
    const bool same = memcmp(got.data(), want.data(), pitch * H) == 0;
    printf("\nSharded result %s the single-device blur (fnv %016llx)\n", same ? "EQUALS" : "DIFFERS FROM",
           (unsigned long long)mi_blur_fnv1a64(got.data(), got.size()));
    printf("\n========== PERFORMANCE RESULTS ==========\n");
    printf("   Total wall-clock time: %.2f ms for %d iterations (%.3f ms per image)\n", ms, opt.iters, ms / opt.iters);
    printf("   Images per second: %.2f\n", opt.iters / (ms / 1000.0));
    printf("   Overall throughput: %.2f Megapixels/sec\n", (double)opt.iters * W * H / (ms / 1000.0) / 1e6);
    const double gbps = 2.0 * pitch * H * opt.iters / (ms / 1000.0) / 1e9;
    printf("   Algorithmic bandwidth (incl. exchange + launch gaps): %.1f GB/s = %.1f%% of %d x %.0f GB/s\n", gbps,
           gbps / (HBM_PEAK_GBS * G) * 100, G, HBM_PEAK_GBS);
    // Per-step decomposition (device time, stream events; outside the timed loop above): how long the halo exchange and
    // the band kernel each occupy a GPU's stream, and — when shards have an interior — the same step issued plain and
    // with the exchange hidden behind the interior rows.  On real xGMI this says whether RCCL latency or the three-launch
    // overlapped form bounds a step.  Skipped with --iterate (extra steps would advance the blur chain).
    if (peer) printf("   Per-step decomposition: none — a step is one launch per GPU (the halo rows cross the link inside the band kernel's loads)\n");
    if (!opt.iterate && !peer) {
        const int n = std::max(1, std::min(opt.iters, 50));
        std::vector<hipEvent_t> ea(G), eb(G), ec(G);
        for (int g = 0; g < G; g++) {
            HIP_OK(hipSetDevice(devs[g]));
            HIP_OK(hipEventCreate(&ea[g])); HIP_OK(hipEventCreate(&eb[g])); HIP_OK(hipEventCreate(&ec[g]));
        }
        std::vector<double> xus(G, 0.0), kus(G, 0.0);
        std::vector<void *> st(G);
        for (int g = 0; g < G; g++) st[g] = stream[g];
        for (int i = 0; i < n; i++) {
            for (int g = 0; g < G; g++) { HIP_OK(hipSetDevice(devs[g])); HIP_OK(hipEventRecord(ea[g], stream[g])); }
            mi_check(mi_blur_halo_exchange_all(comm.data(), G, d_band.data(), W, C, owned.data(), radius, st.data()), "halo exchange failed");
            for (int g = 0; g < G; g++) {
                HIP_OK(hipSetDevice(devs[g]));
                HIP_OK(hipEventRecord(eb[g], stream[g]));
                mi_check(mi_blur_enqueue_band(d_band[g], d_out[g], W, owned[g] + band[g].halo_top + band[g].halo_bottom, C, radius,
                                              band[g].halo_top, band[g].halo_top + owned[g], stream[g]), "band launch failed");
                HIP_OK(hipEventRecord(ec[g], stream[g]));
            }
            sync_all();
            for (int g = 0; g < G; g++) {
                float a = 0.f, b = 0.f;
                HIP_OK(hipEventElapsedTime(&a, ea[g], eb[g])); HIP_OK(hipEventElapsedTime(&b, eb[g], ec[g]));
                xus[g] += a * 1e3; kus[g] += b * 1e3;
            }
        }
        printf("   Per-step decomposition (%d instrumented steps, stream events):\n", n);
        for (int g = 0; g < G; g++)
            printf("     GPU %d: halo_exchange_us %.2f   band_kernel_us %.2f   (%d rows, kernel alone = %.1f GB/s)\n", g, xus[g] / n, kus[g] / n,
                   owned[g], kus[g] > 0 ? 2.0 * pitch * owned[g] / (kus[g] / n * 1e-6) / 1e9 : 0.0);
        if (can_overlap) {
            double t[2];
            for (int ov = 0; ov < 2; ov++) {
                overlap = ov != 0;
                step(); sync_all();
                const double a = get_time_ms();
                for (int i = 0; i < n; i++) step();
                sync_all();
                t[ov] = (get_time_ms() - a) * 1e3 / n;
            }
            overlap = opt.overlap && can_overlap;
            printf("     step_us plain %.2f   overlapped %.2f   (wall clock over %d back-to-back steps each)\n", t[0], t[1], n);
        }
        for (int g = 0; g < G; g++) {
            HIP_OK(hipSetDevice(devs[g]));
            HIP_OK(hipEventDestroy(ea[g])); HIP_OK(hipEventDestroy(eb[g])); HIP_OK(hipEventDestroy(ec[g]));
        }
    }
    for (int g = 0; g < G; g++) {
        HIP_OK(hipSetDevice(devs[g]));
        mi_blur_comm_destroy(comm[g]);
        HIP_OK(hipFree(d_band[g])); HIP_OK(hipFree(d_out[g])); HIP_OK(hipStreamDestroy(stream[g]));
        if (d_band2[g]) HIP_OK(hipFree(d_band2[g]));
        if (ev_step[g]) HIP_OK(hipEventDestroy(ev_step[g]));
        if (xstream[g]) { HIP_OK(hipStreamDestroy(xstream[g])); HIP_OK(hipEventDestroy(ev_done[g])); HIP_OK(hipEventDestroy(ev_halo[g])); }
    }
    return same ? 0 : 1;
}
