// score.hip -- ComputeScore on the GPU: exact Levenshtein identity of the two arms of each
// duplication (reference `--compute-score`: src/bin/asgart.rs:98-112, ProtoSD::levenshtein
// src/structs.rs:439-452, bio::alignment::distance::levenshtein = unit-cost global edit distance).
//
// One WAVEFRONT per duplication, systolic dynamic programming: lane l owns R consecutive rows of the
// DP matrix (the left arm) in registers and walks the columns (the right arm) one step behind lane
// l-1, so a step computes 64 x R cells with two cross-lane moves (the value above the lane's first
// row and the column's base travel down the lanes) and no barrier or LDS traffic.  Arms longer than
// 64 x R rows are processed in bands; the bottom row of a band is the top boundary of the next and
// goes through an HBM scratch row, read and written 64 columns at a time.  Integer work, bit-exact by
// construction; the identity is formed in f64 like the reference and narrowed to f32.
#include "index.hpp"

#include <algorithm>

namespace asgart {
namespace {

constexpr int kScoreThreads = 256;  // four duplications per workgroup
constexpr int kScoreRows = 16;      // rows per lane

// utils::complement_nucleotide / structs::TR on the normalised alphabet; anything else is kept
__device__ inline uint32_t complement_base(uint32_t c) {
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
    constexpr int R = kScoreRows;
    const int lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    uint32_t *row_a = scratch + wave * 2u * scratch_stride, *row_b = row_a + scratch_stride;
    for (;;) {
        unsigned long long item = 0;
        if (lane == 0) item = atomicAdd(cursor, 1ull);
        item = __shfl(item, 0);
        if (item >= n_sd) break;
        const asgart_proto_sd sd = sds[item];
        // inclusive ranges [p ..= p + len] (src/structs.rs:441-442): len + 1 bases each
        const uint32_t la = (uint32_t)sd.left_length + 1u, lb = (uint32_t)sd.right_length + 1u;
        const uint8_t *A = text + sd.left;
        const uint8_t *B = text + sd.right;
        auto b_at = [&](uint32_t j) -> uint32_t {  // j-th base of the right arm after reverse/complement
            const uint32_t c = B[reversed ? lb - 1u - j : j];
            return complemented ? complement_base(c) : c;
        };
        uint32_t result = 0;
        uint32_t *top_row = row_a, *bottom_row = row_b;
        for (uint32_t band = 0; band < la; band += 64u * R) {
            const uint32_t i0 = band + (uint32_t)lane * R;  // rows i0+1 .. i0+R (1-based) are this lane's
            const bool more_bands = band + 64u * R < la;
            uint32_t a[R], col[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                a[r] = i0 + r < la ? (uint32_t)A[i0 + r] : 0x100u;  // padding rows match nothing
                col[r] = i0 + r + 1u;                               // D[i][0] = i
            }
            uint32_t diag_in = i0;       // D[i0][j-1] for the lane's next column
            uint32_t last_out = 0;       // D[i0+R][j] of the column just done
            uint32_t b_pipe = 0;         // base of the lane's current column (travels down the lanes)
            uint32_t top_chunk = 0, b_chunk = 0, out_chunk = 0;
            const uint32_t n_steps = lb + 63u;
            for (uint32_t s = 0; s < n_steps; ++s) {
                if ((s & 63u) == 0u) {  // 64 columns of the top boundary and of the right arm at a time
                    const uint32_t j1 = s + 1u + (uint32_t)lane;  // column of lane 0 at step s + lane
                    // (written by this wave one band ago: read past the vector L1)
                    top_chunk = band == 0 ? j1 : (j1 <= lb ? __atomic_load_n(&top_row[j1], __ATOMIC_RELAXED) : 0u);
                    b_chunk = s + (uint32_t)lane < lb ? b_at(s + (uint32_t)lane) : 0u;
                }
                // the value above the lane's first row, D[i0][j]: lane l-1 finished column j one step ago
                uint32_t up_in = __shfl_up(last_out, 1);
                const uint32_t top0 = __shfl(top_chunk, (int)(s & 63u));
                uint32_t b_in = __shfl_up(b_pipe, 1);
                const uint32_t b0 = __shfl(b_chunk, (int)(s & 63u));
                if (lane == 0) {
                    up_in = top0;
                    b_in = b0;
                }
                b_pipe = b_in;
                const int64_t j = (int64_t)s - lane + 1;
                if (j >= 1 && j <= (int64_t)lb) {
                    uint32_t up = up_in, diag = diag_in;
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const uint32_t left = col[r];
                        const uint32_t v = min(diag + (a[r] != b_in ? 1u : 0u), min(up, left) + 1u);
                        diag = left;
                        up = v;
                        col[r] = v;
                    }
                    diag_in = up_in;
                    last_out = col[R - 1];
                    if ((uint32_t)j == lb && la > i0 && la <= i0 + R) {
                        uint32_t v = 0;
#pragma unroll
                        for (int r = 0; r < R; ++r)
                            if (i0 + r + 1u == la) v = col[r];
                        result = v;
                    }
                }
                if (more_bands) {
                    // lane 63 finishes column jo = s - 62; collect 64 of them, store them together
                    const uint32_t v63 = __shfl(last_out, 63);
                    const int64_t jo = (int64_t)s - 62;
                    if (jo >= 1 && jo <= (int64_t)lb) {
                        if ((uint32_t)lane == ((uint32_t)jo & 63u)) out_chunk = v63;
                        if (((uint32_t)jo & 63u) == 63u || (uint32_t)jo == lb) {
                            const uint32_t jbase = (uint32_t)jo & ~63u;
                            const uint32_t jw = jbase + (uint32_t)lane;
                            if (jw >= 1u && jw <= (uint32_t)jo) bottom_row[jw] = out_chunk;
                        }
                    }
                }
            }
            // the next band reads what this one wrote (same wave: program order; make it visible)
            __threadfence_block();
            uint32_t *t = top_row; top_row = bottom_row; bottom_row = t;
        }
        // the lane that holds row la has the distance
        const uint32_t owner = ((la - 1u) % (64u * R)) / R;  // lane of row la in the last band
        const uint32_t dist = __shfl(result, (int)owner);
        if (lane == 0) {
            const uint64_t longest = sd.left_length > sd.right_length ? sd.left_length : sd.right_length;
            identity[item] = (float)(100.0 * (1.0 - (double)dist / (double)longest));
        }
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
    uint64_t max_lb = 0;
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
        if (sd.left_length + 1u > 64u * (uint64_t)kScoreRows) max_lb = std::max<uint64_t>(max_lb, sd.right_length + 1u);
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
    const unsigned waves_per_wg = kScoreThreads / 64;
    const unsigned grid = (unsigned)std::min<int64_t>((n_sd + waves_per_wg - 1) / waves_per_wg, 512);
    // two boundary rows per wave for the arms that need more than one band
    const uint64_t stride = max_lb ? max_lb + 64u : 0u;
    RC_TRY(w.out_a.reserve((size_t)n_sd * sizeof(asgart_proto_sd)));
    RC_TRY(w.out_b.reserve((size_t)n_sd * sizeof(float) + 64));
    if (stride) RC_TRY(w.scratch.reserve((size_t)stride * 2u * 4u * grid * waves_per_wg));
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
