// micro-benchmark: floor cost of a per-probe loop made of barrier-separated phases, each with one
// dependent LDS read, one LDS atomic and a little ALU, for NT threads in one workgroup on an idle CU
#include <hip/hip_runtime.h>
#include <cstdio>
template <int NT, int PH>
__global__ __launch_bounds__(NT) void k(unsigned long long *out, int iters) {
    __shared__ unsigned int s[4096];
    __shared__ unsigned int cnt[2];
    const int tid = threadIdx.x;
    if (tid < 2) cnt[tid] = 0;
    for (int i = tid; i < 4096; i += NT) s[i] = i * 7;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned int acc = tid;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int p = 0; p < PH; ++p) {
            unsigned int v = s[(acc * 13 + i + p) & 4095];      // dependent read
            if ((v & 7) == 0) atomicMax(&s[(v >> 3) & 4095], acc);  // some lanes post
            acc += v;
            if ((tid & 63) == 0 && (acc & 3) == 0) atomicAdd(&cnt[p & 1], 1u);
            __syncthreads();
            acc += __builtin_amdgcn_readfirstlane(cnt[p & 1]);  // uniform counter read
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) { out[0] = t1 - t0; out[1] = acc; }
}
template <int NT, int PH> void run(unsigned long long *d, const char *name) {
    unsigned long long h[2];
    const int iters = 20000;
    k<NT, PH><<<1, NT>>>(d, iters);
    hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("%s: %.0f cycles per iteration, %.0f per phase\n", name, (double)h[0] / iters, (double)h[0] / iters / PH);
}
int main() {
    unsigned long long *d;
    hipMalloc(&d, 64);
    run<64, 4>(d, "NT=64   4 phases");
    run<256, 4>(d, "NT=256  4 phases");
    run<512, 4>(d, "NT=512  4 phases");
    run<1024, 4>(d, "NT=1024 4 phases");
    run<1024, 3>(d, "NT=1024 3 phases");
    run<1024, 1>(d, "NT=1024 1 phase");
    return 0;
}
