// score.hip -- ComputeScore on the GPU: exact Levenshtein identity of the two arms of each
// duplication (reference `--compute-score`: src/bin/asgart.rs:98-112, ProtoSD::levenshtein
// src/structs.rs:439-452, bio::alignment::distance::levenshtein = unit-cost global edit distance).
//
// One WAVEFRONT per duplication, systolic dynamic programming: lane l owns R consecutive rows of the
// DP matrix (the left arm) in registers and walks the columns (the right arm) one step behind lane
// l-1, so a step computes 64 x R cells with two cross-lane moves (the value above the lane's first
// row and the column's base travel down the lanes) and no barrier or LDS traffic.  Arms longer than
// 64 x R rows are processed in bands; the bottom row of a band is the top boundary of the next and
// goes through an HBM scratch row, read and written 64 columns at a time.  Long duplications get a
// workgroup of 16 waves that walk 16 consecutive bands as a pipeline (see levenshtein_long_kernel).  Integer work, bit-exact by
// construction; the identity is formed in f64 like the reference and narrowed to f32.
#include "index.hpp"

#include <algorithm>
#include <vector>

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

// the right arm as the DP walks it: reversed / complemented first (src/structs.rs:443-448)
struct RightArm {
    const uint8_t *B;
    uint32_t lb;
    int reversed, complemented;
    __device__ uint32_t at(uint32_t j) const {
        const uint32_t c = B[reversed ? lb - 1u - j : j];
        return complemented ? complement_base(c) : c;
    }
};

// one lane's share of a band of 64 x R rows
struct BandState {
    uint32_t a[kScoreRows], col[kScoreRows];
    uint32_t diag_in;    // D[i0][j-1] for the lane's next column
    uint32_t last_out;   // D[i0+R][j] of the column just done
    uint32_t b_pipe;     // base of the lane's current column (travels down the lanes)
    uint32_t top_chunk, b_chunk, out_chunk;
};

__device__ inline void band_init(BandState &st, uint32_t band, int lane, uint32_t la, const uint8_t *A) {
    constexpr int R = kScoreRows;
    const uint32_t i0 = band + (uint32_t)lane * R;  // rows i0+1 .. i0+R (1-based) are this lane's
#pragma unroll
    for (int r = 0; r < R; ++r) {
        st.a[r] = i0 + r < la ? (uint32_t)A[i0 + r] : 0x100u;  // padding rows match nothing
        st.col[r] = i0 + r + 1u;                               // D[i][0] = i
    }
    st.diag_in = i0;
    st.last_out = st.b_pipe = st.top_chunk = st.b_chunk = st.out_chunk = 0;
}

// steps [s_lo, s_hi) of a band (s_lo a multiple of 64); lane l works on column s - l + 1
__device__ inline void band_steps(BandState &st, uint32_t s_lo, uint32_t s_hi, uint32_t band, int lane, uint32_t la,
                                  const RightArm &rb, const uint32_t *top_row, uint32_t *bottom_row,
                                  uint32_t &result) {
    constexpr int R = kScoreRows;
    const uint32_t lb = rb.lb;
    const uint32_t i0 = band + (uint32_t)lane * R;
    const bool more_bands = band + 64u * R < la;
    for (uint32_t s = s_lo; s < s_hi; ++s) {
        if ((s & 63u) == 0u) {  // 64 columns of the top boundary and of the right arm at a time
            const uint32_t j1 = s + 1u + (uint32_t)lane;  // column of lane 0 at step s + lane
            // (the boundary row was written by this or another wave of the workgroup: read past the L1)
            st.top_chunk = band == 0 ? j1 : (j1 <= lb ? __atomic_load_n(&top_row[j1], __ATOMIC_RELAXED) : 0u);
            st.b_chunk = s + (uint32_t)lane < lb ? rb.at(s + (uint32_t)lane) : 0u;
        }
        // the value above the lane's first row, D[i0][j]: lane l-1 finished column j one step ago
        uint32_t up_in = __shfl_up(st.last_out, 1);
        const uint32_t top0 = __shfl(st.top_chunk, (int)(s & 63u));
        uint32_t b_in = __shfl_up(st.b_pipe, 1);
        const uint32_t b0 = __shfl(st.b_chunk, (int)(s & 63u));
        if (lane == 0) {
            up_in = top0;
            b_in = b0;
        }
        st.b_pipe = b_in;
        const int64_t j = (int64_t)s - lane + 1;
        if (j >= 1 && j <= (int64_t)lb) {
            uint32_t up = up_in, diag = st.diag_in;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const uint32_t left = st.col[r];
                const uint32_t v = min(diag + (st.a[r] != b_in ? 1u : 0u), min(up, left) + 1u);
                diag = left;
                up = v;
                st.col[r] = v;
            }
            st.diag_in = up_in;
            st.last_out = st.col[R - 1];
            if ((uint32_t)j == lb && la > i0 && la <= i0 + R) {
                uint32_t v = 0;
#pragma unroll
                for (int r = 0; r < R; ++r)
                    if (i0 + r + 1u == la) v = st.col[r];
                result = v;
            }
        }
        if (more_bands) {
            // lane 63 finishes column jo = s - 62; collect 64 of them, store them together
            const uint32_t v63 = __shfl(st.last_out, 63);
            const int64_t jo = (int64_t)s - 62;
            if (jo >= 1 && jo <= (int64_t)lb) {
                if ((uint32_t)lane == ((uint32_t)jo & 63u)) st.out_chunk = v63;
                if (((uint32_t)jo & 63u) == 63u || (uint32_t)jo == lb) {
                    const uint32_t jw = ((uint32_t)jo & ~63u) + (uint32_t)lane;
                    if (jw >= 1u && jw <= (uint32_t)jo) bottom_row[jw] = st.out_chunk;
                }
            }
        }
    }
}

__device__ inline float identity_of(const asgart_proto_sd &sd, uint32_t dist) {
    const uint64_t longest = sd.left_length > sd.right_length ? sd.left_length : sd.right_length;
    return (float)(100.0 * (1.0 - (double)dist / (double)longest));  // src/structs.rs:451, then `as f32`
}

// ---- one wave per duplication: the bulk (the band loop written out: the shared band_steps form of
// this kernel hung on gfx950 with ROCm 7.2 although the multi-wave kernel below runs the same code) ---
__global__ __launch_bounds__(kScoreThreads) void levenshtein_kernel(const uint8_t *__restrict__ text,
                                                                    const asgart_proto_sd *__restrict__ sds,
                                                                    const uint32_t *__restrict__ list, uint64_t n_list,
                                                                    int reversed, int complemented,
                                                                    uint32_t *__restrict__ scratch, uint64_t scratch_stride,
                                                                    unsigned long long *__restrict__ cursor,
                                                                    float *__restrict__ identity) {
    constexpr int R = kScoreRows;
    const int lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    uint32_t *row_a = scratch + wave * 2u * scratch_stride, *row_b = row_a + scratch_stride;
    for (;;) {
        unsigned long long pos = 0;
        if (lane == 0) pos = atomicAdd(cursor, 1ull);
        pos = __shfl(pos, 0);
        if (pos >= n_list) break;
        const uint32_t item = list[pos];
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


// ---- one workgroup of kLongWaves waves per LONG duplication -----------------------------------------
// Band b is walked by wave b mod kLongWaves in chunks of C steps; a workgroup barrier separates the
// chunks and band b+1 runs two chunks behind band b (its top boundary is band b's bottom row, whose
// columns up to (c+1) C are complete after chunk c+1).  With at most 2 kLongWaves chunks per band a
// wave finishes band b exactly when band b + kLongWaves may start, so every wave stays busy.
constexpr int kLongWaves = 16;
constexpr int kLongRows = kLongWaves + 2;  // boundary rows in flight

__global__ __launch_bounds__(64 * kLongWaves) void levenshtein_long_kernel(
    const uint8_t *__restrict__ text, const asgart_proto_sd *__restrict__ sds, const uint32_t *__restrict__ list,
    uint64_t n_list, int reversed, int complemented, uint32_t *__restrict__ scratch, uint64_t scratch_stride,
    float *__restrict__ identity) {
    constexpr int R = kScoreRows;
    constexpr uint32_t BAND = 64u * R;
    __shared__ uint32_t s_dist;
    const int lane = threadIdx.x & 63;
    const uint32_t w = threadIdx.x >> 6;
    uint32_t *rows = scratch + (size_t)blockIdx.x * kLongRows * scratch_stride;
    for (uint64_t pos = blockIdx.x; pos < n_list; pos += gridDim.x) {
        const uint32_t item = list[pos];
        const asgart_proto_sd sd = sds[item];
        const uint32_t la = (uint32_t)sd.left_length + 1u, lb = (uint32_t)sd.right_length + 1u;
        const RightArm rb{text + sd.right, lb, reversed, complemented};
        const uint32_t n_bands = (la + BAND - 1u) / BAND;
        const uint32_t n_steps = lb + 63u;
        uint32_t C = (n_steps + 2u * kLongWaves - 1u) / (2u * kLongWaves);
        C = max(128u, (C + 63u) & ~63u);  // >= 128: a band's bottom row is stored 64 columns at a time, 62 steps late
        const uint32_t n_chunks = (n_steps + C - 1u) / C;  // <= 2 kLongWaves
        const uint32_t n_super = 2u * (n_bands - 1u) + n_chunks;
        uint32_t result = 0;
        BandState st;
        for (uint32_t t = 0; t < n_super; ++t) {
            if (t >= 2u * w) {
                const uint32_t rel = t - 2u * w;
                const uint32_t m = rel / (2u * kLongWaves), c = rel % (2u * kLongWaves);
                const uint32_t b = w + m * kLongWaves;
                if (b < n_bands && c < n_chunks) {
                    const uint32_t band = b * BAND;
                    if (c == 0) band_init(st, band, lane, la, text + sd.left);
                    const uint32_t *top_row = rows + (size_t)((b + kLongRows - 1u) % kLongRows) * scratch_stride;
                    uint32_t *bottom_row = rows + (size_t)(b % kLongRows) * scratch_stride;
                    band_steps(st, c * C, min((c + 1u) * C, n_steps), band, lane, la, rb, top_row, bottom_row, result);
                }
            }
            __syncthreads();
        }
        // the wave and lane that hold row la have the distance
        const uint32_t last_band = n_bands - 1u;
        const uint32_t owner = ((la - 1u) % BAND) / R;
        if (w == last_band % kLongWaves && (uint32_t)lane == owner) s_dist = result;
        __syncthreads();
        if (threadIdx.x == 0) identity[item] = identity_of(sd, s_dist);
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
    }
    REFUSE_POISONED(idx);
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
    // Long duplications (many bands) get a whole workgroup each, the rest one wave each; both lists
    // are served largest first.
    std::vector<uint32_t> order((size_t)n_sd);
    for (int64_t q = 0; q < n_sd; ++q) order[(size_t)q] = (uint32_t)q;
    auto cells = [&](uint32_t q) { return (double)(sds[q].left_length + 1u) * (double)(sds[q].right_length + 1u); };
    std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return cells(x) > cells(y); });
    const uint64_t long_rows = 64ull * kScoreRows * (uint64_t)kLongWaves / 2u;  // half a workgroup's bands
    std::vector<uint32_t> long_list, wave_list;
    uint64_t max_lb_long = 0, max_lb_wave = 0;
    for (uint32_t q : order) {
        const asgart_proto_sd &sd = sds[q];
        if (sd.left_length + 1u >= long_rows) {
            long_list.push_back(q);
            max_lb_long = std::max<uint64_t>(max_lb_long, sd.right_length + 1u);
        } else {
            wave_list.push_back(q);
            if (sd.left_length + 1u > 64u * (uint64_t)kScoreRows) max_lb_wave = std::max<uint64_t>(max_lb_wave, sd.right_length + 1u);
        }
    }
    const unsigned waves_per_wg = kScoreThreads / 64;
    const unsigned grid_w = (unsigned)std::min<size_t>((wave_list.size() + waves_per_wg - 1) / waves_per_wg, 512);
    const unsigned grid_l = (unsigned)std::min<size_t>(long_list.size(), 256);
    const uint64_t stride_w = max_lb_wave ? max_lb_wave + 64u : 0u, stride_l = max_lb_long + 64u;
    const size_t scratch_w = (size_t)stride_w * 2u * grid_w * waves_per_wg;          // u32 entries
    const size_t scratch_l = (size_t)stride_l * kLongRows * grid_l;
    RC_TRY(w.out_a.reserve((size_t)n_sd * sizeof(asgart_proto_sd)));
    RC_TRY(w.out_b.reserve((size_t)n_sd * sizeof(float) + 64));
    RC_TRY(w.seg_vals.reserve((size_t)n_sd * 4 + 64));
    RC_TRY(w.scratch.reserve((scratch_w + scratch_l) * 4u + 64));
    RC_TRY(w.counters.reserve(1024));  // the search pipeline keeps its device counters here too
    unsigned long long *cursor = w.counters.as<unsigned long long>();
    uint32_t *d_list = w.seg_vals.as<uint32_t>();
    HIP_TRY(hipMemsetAsync(cursor, 0, 8, s));
    HIP_TRY(hipMemcpyAsync(w.out_a.p, sds, (size_t)n_sd * sizeof(asgart_proto_sd), hipMemcpyHostToDevice, s));
    if (!long_list.empty())
        HIP_TRY(hipMemcpyAsync(d_list, long_list.data(), long_list.size() * 4, hipMemcpyHostToDevice, s));
    if (!wave_list.empty())
        HIP_TRY(hipMemcpyAsync(d_list + long_list.size(), wave_list.data(), wave_list.size() * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(stream_sync(s));  // the host vectors go out of use below
    if (grid_l)
        levenshtein_long_kernel<<<grid_l, 64 * kLongWaves, 0, s>>>(
            idx->d_text, w.out_a.as<asgart_proto_sd>(), d_list, (uint64_t)long_list.size(), reversed != 0,
            complemented != 0, w.scratch.as<uint32_t>() + scratch_w, stride_l, w.out_b.as<float>());
    if (grid_w)
        levenshtein_kernel<<<grid_w, kScoreThreads, 0, s>>>(
            idx->d_text, w.out_a.as<asgart_proto_sd>(), d_list + long_list.size(), (uint64_t)wave_list.size(),
            reversed != 0, complemented != 0, w.scratch.as<uint32_t>(), stride_w, cursor, w.out_b.as<float>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(read_back(identity, w.out_b.p, (size_t)n_sd * sizeof(float), s));
    HIP_TRY(stream_sync(s));
    return 0;
}
