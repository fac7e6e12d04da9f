// micro-benchmarks behind the design of extend_fast_dev.hpp: what one wave (alone, or one of NT/64 waves of a
// workgroup) pays per instruction kind on gfx950.  hipcc --offload-arch=gfx950 tools/ubench_isa.hip -o tools/bin/ubench_isa
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ inline void lds_barrier() { __asm__ volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int NT, int MODE>
__global__ __launch_bounds__(NT) void k(unsigned long long *out, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned long long tab[4096];
    __shared__ unsigned int best[2048];
    __shared__ unsigned int sink[64];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 4096; i += NT) tab[i] = (unsigned long long)i * 0x9E3779B97F4A7C15ull;
    for (int i = tid; i < 2048; i += NT) best[i] = i * 7u;
    __syncthreads();
    unsigned int acc = tid * 2654435761u, acc2 = tid;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0) {  // 64 dependent VALU adds/xors
#pragma unroll
            for (int j = 0; j < 64; ++j) acc = (acc ^ (acc >> 3)) + j;
        } else if constexpr (MODE == 1) {  // 16 x (compare -> exec-masked block)
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                if (((acc >> j) & 15u) == 3u) acc2 += acc * 3u;
                acc += 0x61C88647u;
            }
        } else if constexpr (MODE == 2) {  // 16 x (compare -> select), same work branch-free
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                acc2 = ((acc >> j) & 15u) == 3u ? acc2 + acc * 3u : acc2;
                acc += 0x61C88647u;
            }
        } else if constexpr (MODE == 3) {  // 8 dependent LDS reads
#pragma unroll
            for (int j = 0; j < 8; ++j) acc = best[acc & 2047u] + j;
        } else if constexpr (MODE == 4) {  // 4 x 16-byte reads in flight, then use
            const ulonglong2 *p = reinterpret_cast<const ulonglong2 *>(&tab[(acc & 1023u) * 4u]);
            const ulonglong2 a = p[0], b = p[1];
            const ulonglong2 *q = reinterpret_cast<const ulonglong2 *>(&tab[((acc + 1u) & 1023u) * 4u]);
            const ulonglong2 c = q[0], d = q[1];
            acc += (unsigned int)(a.x ^ a.y ^ b.x ^ b.y ^ c.x ^ c.y ^ d.x ^ d.y);
        } else if constexpr (MODE == 5) {  // 8 unconditional LDS atomics (no return) then a wait
#pragma unroll
            for (int j = 0; j < 8; ++j) atomicMin(((acc >> j) & 7u) == 0u ? &best[(acc >> 8) & 2047u] : &sink[lane], acc);
            __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            acc += 0x61C88647u;
        } else if constexpr (MODE == 6) {  // 8 exec-masked LDS atomics then a wait
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (((acc >> j) & 7u) == 0u) atomicMin(&best[(acc >> 8) & 2047u], acc);
            __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            acc += 0x61C88647u;
        } else if constexpr (MODE == 7) {  // barrier only
            lds_barrier();
        } else if constexpr (MODE == 8) {  // s_memtime pair
            acc += (unsigned int)__builtin_amdgcn_s_memtime();
        } else if constexpr (MODE == 9) {  // ballot + popcount + readlane chain x 8
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const unsigned long long m = __ballot(((acc >> j) & 1u) != 0u);
                acc += (unsigned int)__popcll(m) + (unsigned int)__builtin_amdgcn_readlane((int)acc, j);
            }
        } else if constexpr (MODE == 10) {  // one LDS write + barrier + one LDS read (an exchange)
            best[tid & 2047] = acc;
            lds_barrier();
            acc += best[(tid * 7 + 1) & 2047];
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) { out[0] = t1 - t0; }
    if (acc + acc2 == 12345u) out[1] = acc;
}
template <int NT, int MODE> void run(unsigned long long *d, const char *name) {
    unsigned long long h[2];
    const int iters = 20000;
    k<NT, MODE><<<1, NT>>>(d, iters);
    hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("%-58s NT=%4d: %7.1f cycles per iteration\n", name, NT, (double)h[0] / iters);
}
#define RUN3(M, name) run<64, M>(d, name); run<256, M>(d, name); run<1024, M>(d, name)
int main() {
    unsigned long long *d;
    hipMalloc(&d, 64);
    RUN3(0, "64 dependent VALU ops (128 instr)");
    RUN3(1, "16 x compare -> exec-masked block");
    RUN3(2, "16 x compare -> select");
    RUN3(3, "8 dependent LDS reads");
    RUN3(4, "4 x ds_read_b128 in flight + use");
    RUN3(5, "8 unconditional LDS atomics + wait");
    RUN3(6, "8 exec-masked LDS atomics + wait");
    RUN3(7, "barrier");
    RUN3(8, "s_memtime");
    RUN3(9, "8 x ballot + popcount + readlane");
    RUN3(10, "LDS write + barrier + LDS read");
    return 0;
}
