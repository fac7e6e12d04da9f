// micro-benchmark: the random-gather regime of MI355X HBM, the bound of the probe-search kernel.
// Every thread performs `iters` rounds of U independent 8-byte (or 4-byte) loads at pseudo-random
// element indices of a table far larger than the Infinity Cache; with DEP the next round's indices
// depend on the loaded values (the dependent chain prefix table -> key bisection -> suffix array).
// Prints gathers per second; run it under `rocprofv3 --pmc FETCH_SIZE --kernel-trace` to read the
// bytes the memory side counts per gather (calibrates FETCH_SIZE for this access width, which
// MI355X_MICROARCH.md leaves uncalibrated).
//   hipcc -O3 --offload-arch=gfx950 -o tools/bin/ubench_gather tools/ubench_gather.hip
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

__device__ __host__ inline uint64_t mix(uint64_t x) {
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdull;
    x ^= x >> 33;
    x *= 0xc4ceb9fe1a85ec53ull;
    x ^= x >> 33;
    return x;
}

__global__ void fill(uint64_t *t, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        t[i] = mix(i);
}

template <class T, int U, bool DEP>
__global__ __launch_bounds__(256) void gather(const T *__restrict__ table, uint64_t mask, int iters,
                                              unsigned long long *out) {
    uint64_t idx = mix(((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull + 1);
    uint64_t acc = 0;
    for (int it = 0; it < iters; ++it) {
        T v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = table[mix(idx + u) & mask];
        uint64_t s = 0;
#pragma unroll
        for (int u = 0; u < U; ++u) s += v[u];
        acc += s;
        idx = DEP ? mix(idx ^ s) : mix(idx + 0x1234567ull);
    }
    if (acc == 0x123456789abcdefull) out[0] = acc;  // keeps the loads alive
}

template <class T, int U, bool DEP>
static void run(const void *table, size_t bytes, const char *name) {
    const uint64_t mask = bytes / sizeof(T) - 1;
    const int iters = 64 / U * 4, blocks = 256 * 8;
    unsigned long long *d_out;
    hipMalloc(&d_out, 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    gather<T, U, DEP><<<blocks, 256>>>((const T *)table, mask, iters, d_out);  // warm-up
    hipEventRecord(e0);
    const int reps = 5;
    for (int r = 0; r < reps; ++r) gather<T, U, DEP><<<blocks, 256>>>((const T *)table, mask, iters, d_out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double gathers = (double)reps * blocks * 256 * iters * U;
    printf("%-34s %7.2f G gathers/s  (x64 B = %6.2f TB/s, x128 B = %6.2f TB/s)  %.3f ms per launch, %.0f gathers per launch\n",
           name, gathers / ms / 1e6, gathers * 64 / ms / 1e9, gathers * 128 / ms / 1e9, ms / reps,
           gathers / reps);
    hipFree(d_out);
}

int main(int argc, char **argv) {
    const size_t gib = argc > 1 ? (size_t)atoi(argv[1]) : 16;  // power of two
    const size_t bytes = gib << 30;
    uint64_t *table;
    if (hipMalloc(&table, bytes) != hipSuccess) {
        fprintf(stderr, "hipMalloc(%zu GiB) failed\n", gib);
        return 1;
    }
    fill<<<4096, 256>>>(table, bytes / 8);
    hipDeviceSynchronize();
    printf("table %zu GiB, 2048 workgroups x 256 threads (8 waves per SIMD)\n", gib);
    run<uint64_t, 1, true>(table, bytes, "u64 x1 dependent");
    run<uint64_t, 1, false>(table, bytes, "u64 x1 independent rounds");
    run<uint64_t, 4, true>(table, bytes, "u64 x4 in flight, dependent");
    run<uint64_t, 4, false>(table, bytes, "u64 x4 in flight, independent");
    run<uint32_t, 4, false>(table, bytes, "u32 x4 in flight, independent");
    run<uint64_t, 8, false>(table, bytes, "u64 x8 in flight, independent");
    hipFree(table);
    return 0;
}
