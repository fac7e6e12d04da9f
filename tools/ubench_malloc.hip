// micro-benchmark: what a large device allocation costs on this runtime.  Finding (MI355X, ROCm 7.2): hipMalloc /
// hipFree of tens of GB are sub-millisecond by themselves, but the FIRST hipMalloc after a lot of memory has been
// freed pays for giving it back (seconds per 100 GB) -- hence the block cache of common.hpp (DevBuf).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static double timed_malloc(void **p, size_t bytes) {
    const double t0 = now();
    hipError_t e = hipMalloc(p, bytes);
    const double t1 = now();
    if (e != hipSuccess) printf("  (hipMalloc failed: %s)\n", hipGetErrorString(e));
    return t1 - t0;
}
int main() {
    hipFree(nullptr);
    for (int touch = 0; touch < 2; ++touch)
        for (size_t total_gb : {26, 78, 150}) {
            std::vector<void *> blocks;
            for (size_t got = 0; got < total_gb; got += 13) {
                void *p = nullptr;
                timed_malloc(&p, (size_t)13 << 30);
                if (touch) hipMemset(p, 1, (size_t)13 << 30);
                blocks.push_back(p);
            }
            hipDeviceSynchronize();
            double t0 = now();
            for (void *p : blocks) hipFree(p);
            double t1 = now();
            void *q = nullptr;
            const double tm = timed_malloc(&q, (size_t)26 << 30);
            void *r = nullptr;
            const double tm2 = timed_malloc(&r, (size_t)1 << 30);
            hipFree(q); hipFree(r);
            printf("%s %3zu GB in 13-GB blocks, freed in %.3f s; next hipMalloc(26 GB) %.3f s, then hipMalloc(1 GB) %.3f s\n",
                   touch ? "touched  " : "untouched", total_gb, t1 - t0, tm, tm2);
        }
    return 0;
}
