// score.hip -- ComputeScore on the GPU: exact Levenshtein identity of the two arms of each
// duplication (reference `--compute-score`: src/bin/asgart.rs:98-112, ProtoSD::levenshtein
// src/structs.rs:439-452, bio::alignment::distance::levenshtein = unit-cost global edit distance).
//
// One workgroup per duplication, anti-diagonal dynamic programming: on diagonal d = i + j the
// cells D[i][d-i] only depend on the two previous diagonals, so three arrays indexed by i rotate
// (LDS for arms up to kLdsArm bases, an HBM scratch slice per workgroup beyond).  Integer work,
// bit-exact by construction; the identity is formed in f64 like the reference and narrowed to f32.
#include "index.hpp"

#include <algorithm>

namespace asgart {
namespace {

constexpr int kScoreThreads = 256;
constexpr uint32_t kLdsArm = 4095;  // 3 x 4096 x 4 B = 48 KB of LDS per workgroup

// utils::complement_nucleotide / structs::TR on the normalised alphabet; anything else is kept
__device__ inline uint8_t complement_base(uint8_t c) {
    switch (c) {
    case 'A': return 'T';
    case 'T': return 'A';
    case 'G': return 'C';
    case 'C': return 'G';
    case 'a': return 't';
    case 't': return 'a';
    case 'g': return 'c';
    case 'c': return 'g';
    default: return c;
    }
}

__global__ __launch_bounds__(kScoreThreads) void levenshtein_kernel(const uint8_t *__restrict__ text,
                                                                    const asgart_proto_sd *__restrict__ sds,
                                                                    uint64_t n_sd, int reversed, int complemented,
                                                                    uint32_t *__restrict__ scratch, uint64_t scratch_stride,
                                                                    unsigned long long *__restrict__ cursor,
                                                                    float *__restrict__ identity) {
    __shared__ uint32_t l_diag[3][kLdsArm + 1];
    __shared__ unsigned long long s_item;
    const int tid = threadIdx.x;
    for (;;) {
        if (tid == 0) s_item = atomicAdd(cursor, 1ull);
        __syncthreads();
        const unsigned long long item = s_item;
        __syncthreads();
        if (item >= n_sd) break;
        const asgart_proto_sd sd = sds[item];
        // inclusive ranges [p ..= p + len] (src/structs.rs:441-442): len + 1 bases each
        const uint64_t la = sd.left_length + 1u, lb = sd.right_length + 1u;
        const uint8_t *A = text + sd.left;
        const uint8_t *B = text + sd.right;
        auto b_at = [&](uint64_t j) -> uint8_t {  // j-th base of the right arm after reverse/complement
            uint8_t c = B[reversed ? lb - 1u - j : j];
            return complemented ? complement_base(c) : c;
        };
        uint32_t *d0, *d1, *d2;  // diagonals d-2, d-1, d (indexed by i = row of A)
        if (la <= (uint64_t)kLdsArm) {
            d0 = l_diag[0]; d1 = l_diag[1]; d2 = l_diag[2];
        } else {
            uint32_t *base = scratch + (size_t)blockIdx.x * scratch_stride;
            d0 = base; d1 = base + (la + 1u); d2 = base + 2u * (la + 1u);
        }
        // D[i][j]: i in [0, la], j in [0, lb].  Diagonal 0 = {D[0][0] = 0}.
        if (tid == 0) d1[0] = 0u;
        __syncthreads();
        for (uint64_t d = 1; d <= la + lb; ++d) {
            const uint64_t i_lo = d > lb ? d - lb : 0u, i_hi = d < la ? d : la;
            for (uint64_t i = i_lo + tid; i <= i_hi; i += kScoreThreads) {
                const uint64_t j = d - i;
                uint32_t v;
                if (i == 0) {
                    v = (uint32_t)j;
                } else if (j == 0) {
                    v = (uint32_t)i;
                } else {
                    const uint32_t up = d1[i - 1] + 1u;    // D[i-1][j]
                    const uint32_t left = d1[i] + 1u;      // D[i][j-1]
                    const uint32_t diag = d0[i - 1] + (A[i - 1] != b_at(j - 1) ? 1u : 0u);
                    v = min(diag, min(up, left));
                }
                d2[i] = v;
            }
            __syncthreads();
            uint32_t *t = d0; d0 = d1; d1 = d2; d2 = t;
        }
        if (tid == 0) {
            const double dist = (double)d1[la];
            const uint64_t longest = sd.left_length > sd.right_length ? sd.left_length : sd.right_length;
            identity[item] = (float)(100.0 * (1.0 - dist / (double)longest));
        }
        __syncthreads();
    }
}

}  // namespace
}  // namespace asgart

extern "C" int32_t asgart_compute_scores(asgart_index *idx, const asgart_proto_sd *sds, int64_t n_sd,
                                         int32_t reversed, int32_t complemented, float *identity) {
    using namespace asgart;
    if (!idx || n_sd < 0 || (n_sd > 0 && (!sds || !identity))) {
        set_error("asgart_compute_scores: bad argument");
        return ASGART_E_ARG;
    }
    if (n_sd == 0) return 0;
    const uint64_t n = (uint64_t)idx->n;
    uint64_t max_la = 0;
    for (int64_t q = 0; q < n_sd; ++q) {
        const asgart_proto_sd &sd = sds[q];
        // the reference slices [p ..= p + len] and panics past the end of the strand
        if (sd.left > n || sd.left_length >= n - sd.left || sd.right > n || sd.right_length >= n - sd.right) {
            set_error("asgart_compute_scores: duplication %lld reaches past the end of the text", (long long)q);
            return ASGART_E_ARG;
        }
        if (sd.left_length == 0 && sd.right_length == 0) {
            set_error("asgart_compute_scores: duplication %lld has two empty arms", (long long)q);
            return ASGART_E_ARG;
        }
        if (sd.left_length + sd.right_length + 2u >= 0xFFFFFFFFull) {
            set_error("asgart_compute_scores: arms of 2^32 bases are not supported");
            return ASGART_E_CAP;
        }
        max_la = std::max<uint64_t>(max_la, sd.left_length + 1u);
    }
    HIP_TRY(hipSetDevice(idx->device));
    int which = 0;
    SearchCtx &cx = idx->acquire_one(&which);
    struct Unlock {
        asgart_index *i;
        int w;
        ~Unlock() { i->release_one(w); }
    } unlock{idx, which};
    Workspace &w = cx.ws;
    hipStream_t s = cx.stream;
    const unsigned grid = (unsigned)std::min<int64_t>(n_sd, 256 * 3);
    const uint64_t stride = max_la > (uint64_t)kLdsArm ? 3u * (max_la + 1u) : 0u;
    RC_TRY(w.out_a.reserve((size_t)n_sd * sizeof(asgart_proto_sd)));
    RC_TRY(w.out_b.reserve((size_t)n_sd * sizeof(float) + 64));
    if (stride) RC_TRY(w.scratch.reserve((size_t)stride * 4u * grid));
    RC_TRY(w.counters.reserve(1024));  // the search pipeline keeps its device counters here too
    unsigned long long *cursor = w.counters.as<unsigned long long>();
    HIP_TRY(hipMemsetAsync(cursor, 0, 8, s));
    HIP_TRY(hipMemcpyAsync(w.out_a.p, sds, (size_t)n_sd * sizeof(asgart_proto_sd), hipMemcpyHostToDevice, s));
    levenshtein_kernel<<<grid, kScoreThreads, 0, s>>>(idx->d_text, w.out_a.as<asgart_proto_sd>(), (uint64_t)n_sd,
                                                      reversed != 0, complemented != 0, w.scratch.as<uint32_t>(),
                                                      stride, cursor, w.out_b.as<float>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(identity, w.out_b.p, (size_t)n_sd * sizeof(float), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return 0;
}
