// Unit check of the wave-level helpers of extend_fast_dev.hpp on the GPU: the DPP inclusive scan and the
// n-th-set-bit select, against host loops.  Build + run: hipcc --offload-arch=gfx950 -I asgart_amd/csrc
// -I include tools/test_wave_scan.hip -o /tmp/tws && /tmp/tws
#include "extend_fast_dev.hpp"

#include <cstdio>
#include <random>
#include <vector>

__global__ void scan_kernel(const uint32_t *in, uint32_t *out) {
    out[blockIdx.x * 64 + threadIdx.x] = asgart::wave_incl_scan(in[blockIdx.x * 64 + threadIdx.x]);
}
__global__ void select_kernel(const unsigned long long *m, const uint32_t *n, uint32_t *out) {
    const int t = blockIdx.x * 64 + threadIdx.x;
    out[t] = asgart::select_bit(m[t], n[t]);
}

int main() {
    const int N = 64 * 256;
    std::mt19937_64 rng(3);
    std::vector<uint32_t> in(N), out(N), nn(N);
    std::vector<unsigned long long> mm(N);
    for (int i = 0; i < N; ++i) {
        in[i] = (uint32_t)(rng() % 65);
        mm[i] = rng() & rng();
        if (i % 7 == 0) mm[i] |= rng();
        if (!mm[i]) mm[i] = 1ull << (rng() % 64);
        nn[i] = (uint32_t)(rng() % __builtin_popcountll(mm[i]));
    }
    uint32_t *d_in, *d_out, *d_n;
    unsigned long long *d_m;
    hipMalloc(&d_in, N * 4); hipMalloc(&d_out, N * 4); hipMalloc(&d_n, N * 4); hipMalloc(&d_m, N * 8);
    hipMemcpy(d_in, in.data(), N * 4, hipMemcpyHostToDevice);
    scan_kernel<<<N / 64, 64>>>(d_in, d_out);
    hipMemcpy(out.data(), d_out, N * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int b = 0; b < N / 64; ++b) {
        uint32_t acc = 0;
        for (int l = 0; l < 64; ++l) {
            acc += in[b * 64 + l];
            if (out[b * 64 + l] != acc && bad++ < 5) printf("scan mismatch block %d lane %d: %u != %u\n", b, l, out[b * 64 + l], acc);
        }
    }
    hipMemcpy(d_m, mm.data(), N * 8, hipMemcpyHostToDevice);
    hipMemcpy(d_n, nn.data(), N * 4, hipMemcpyHostToDevice);
    select_kernel<<<N / 64, 64>>>(d_m, d_n, d_out);
    hipMemcpy(out.data(), d_out, N * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < N; ++i) {
        uint32_t want = 0, left = nn[i];
        for (uint32_t p = 0; p < 64; ++p)
            if (mm[i] >> p & 1) {
                if (left == 0) { want = p; break; }
                --left;
            }
        if (out[i] != want && bad++ < 10) printf("select mismatch %d: %u != %u\n", i, out[i], want);
    }
    printf(bad ? "FAILED (%d)\n" : "wave helpers ok\n", bad);
    return bad ? 1 : 0;
}
