// micro-benchmark: what a wave pays for skipping a large block of code with a taken (uniform) branch
#include <hip/hip_runtime.h>
#include <cstdio>
template <int NT, int BLK>
__global__ __launch_bounds__(NT) void k(unsigned long long *out, const unsigned int *flags, int iters) {
    const int tid = threadIdx.x;
    unsigned int acc = tid * 2654435761u;
    unsigned int f[8];
    for (int j = 0; j < 8; ++j) f[j] = __builtin_amdgcn_readfirstlane(flags[j]);
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (f[j] & (1u << (it & 3))) {  // never true
#pragma unroll
                for (int q = 0; q < BLK; ++q) acc = (acc ^ (acc >> 3)) * 2654435761u + q + j;
            }
            acc += j;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) out[0] = t1 - t0;
    if (acc == 12345u) out[1] = acc;
}
template <int NT, int BLK> void run(unsigned long long *d, unsigned int *fl) {
    unsigned long long h[2];
    const int iters = 20000;
    k<NT, BLK><<<1, NT>>>(d, fl, iters);
    hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("skip 8 blocks of %4d x 3 instr, NT=%4d: %7.1f cycles per iteration (%.1f per skipped block)\n", BLK, NT, (double)h[0] / iters, (double)h[0] / iters / 8);
}
int main() {
    unsigned long long *d; unsigned int *fl;
    hipMalloc(&d, 64); hipMalloc(&fl, 64); hipMemset(fl, 0, 64);
    run<64, 4>(d, fl); run<64, 64>(d, fl); run<64, 256>(d, fl);
    run<1024, 4>(d, fl); run<1024, 64>(d, fl); run<1024, 256>(d, fl);
    return 0;
}
