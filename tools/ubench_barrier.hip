// micro-benchmark: cost of one __syncthreads()-separated phase for NT threads per workgroup
#include <hip/hip_runtime.h>
#include <cstdio>
template <int NT>
__global__ __launch_bounds__(NT) void k(unsigned long long *out, int iters) {
    __shared__ unsigned int s[4096];
    __shared__ unsigned int cnt;
    const int tid = threadIdx.x;
    if (tid == 0) cnt = 0;
    s[tid] = tid;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned int acc = 0;
    for (int i = 0; i < iters; ++i) {
        s[(tid * 7 + i) & 4095] = acc + i;
        __syncthreads();
        acc += s[(tid * 13 + i) & 4095];
        if ((acc & 1023) == 7) atomicAdd(&cnt, 1u);
        __syncthreads();
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = acc + cnt; }
}
int main() {
    unsigned long long *d, h[8];
    hipMalloc(&d, 64);
    const int iters = 10000;
    k<1024><<<1, 1024>>>(d, iters); hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("NT=1024: %.1f cycles per 2-barrier iteration\n", (double)h[0] / iters);
    k<256><<<1, 256>>>(d, iters); hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("NT=256:  %.1f cycles per 2-barrier iteration\n", (double)h[0] / iters);
    k<64><<<1, 64>>>(d, iters); hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("NT=64:   %.1f cycles per 2-barrier iteration\n", (double)h[0] / iters);
    return 0;
}
