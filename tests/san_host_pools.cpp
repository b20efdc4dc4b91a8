// host-only ThreadSanitizer run of the hosts' helper-thread pools (apps/host_common.h): the Replicator that builds a
// batch stream (heterogeneous_blur.c:439-442 spread over threads) and the TaskPool that decodes / saves frames.
#include "host_common.h"
extern "C" int mi_blur_bind_thread_to_device(int) { return 0; }      // the pools only bind when given a device; never called here
int main()
{
    using namespace host;
    const size_t isz = 5003;      // odd, past the 4096-byte threshold of the non-temporal copy: unaligned head and tail on every slot
    std::vector<uint8_t> src(isz), dst(isz * 64);
    for (size_t i = 0; i < isz; i++) src[i] = (uint8_t)(i * 7);
    for (int threads : {1, 2, 5}) {
        Replicator rep(threads);
        for (int round = 0; round < 200; round++) {
            const int count = 1 + round % 64;
            memset(dst.data(), 0, dst.size());
            rep.run(dst.data(), src.data(), isz, count, round % 2 == 1);      // plain memcpy and the streaming-store copy in turn
            for (int i = 0; i < count; i++)
                if (memcmp(dst.data() + (size_t)i * isz, src.data(), isz)) { printf("REPLICATE MISMATCH\n"); return 1; }
        }
        TaskPool pool(threads);
        std::vector<int> hits(97);
        std::vector<long> per_thread(pool.threads(), 0);
        for (int round = 0; round < 300; round++) {
            const int n = round % 97;
            std::fill(hits.begin(), hits.end(), 0);
            pool.run(n, [&](int item, int t) { hits[item] += 1; per_thread[t] += item; });
            for (int i = 0; i < 97; i++)
                if (hits[i] != (i < n ? 1 : 0)) { printf("TASKPOOL: item %d ran %d times\n", i, hits[i]); return 1; }
        }
    }
    printf("pools clean\n");
    return 0;
}
