// prep.hip -- input preparation on the GPU: asgart_prepare_data.
//
// Replaces the body of prepare_data behind the FASTA reader, reference src/bin/asgart.rs:273-430: per record the
// alphabet normalisation (:289-301: upper-case unless --skip-masked, then everything outside {A,T,G,C,N} -> N),
// find_chunks_to_process (:317-366: the record is cut at runs of more than 5000 N), the concatenation of the records
// with per-record chunk offsets (:375-395) and the final '$' (:430).  The host code of this step was 70 % of the
// FASTA -> JSON chain at GRCh38 size (numpy: 11 s against 0.6 s for both search passes); the index needs the text on
// the device anyway, so the raw bytes are uploaded once, normalised in place by one streaming kernel, the long N-runs
// are found there, and the index is built from the same device buffer.
#include "index.hpp"

#include <algorithm>

namespace asgart {
namespace {

constexpr uint64_t kNRunThreshold = 5000;  // reference src/bin/asgart.rs:326

// (:291-301) c -> upper case unless skip_masked; then anything outside ATGCN -> N
__device__ inline uint32_t norm_byte(uint32_t c, bool skip_masked) {
    if (!skip_masked && c >= 'a' && c <= 'z') c -= 32u;
    const bool ok = c == 'A' || c == 'T' || c == 'G' || c == 'C' || c == 'N';
    return ok ? c : (uint32_t)'N';
}

__global__ __launch_bounds__(256) void normalise_kernel(uint8_t *__restrict__ text, uint64_t n, int skip_masked) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x * 16u;
    for (uint64_t base = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16u; base < n; base += stride) {
        if (base + 16u <= n) {  // (the buffer is 16-byte aligned: hipMalloc)
            uint4 v = *reinterpret_cast<const uint4 *>(text + base);
            uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                uint32_t o = 0;
#pragma unroll
                for (int b = 0; b < 4; ++b) o |= norm_byte((w[a] >> (8 * b)) & 0xFFu, skip_masked != 0) << (8 * b);
                w[a] = o;
            }
            *reinterpret_cast<uint4 *>(text + base) = make_uint4(w[0], w[1], w[2], w[3]);
        } else {
            for (uint64_t j = base; j < n; ++j) text[j] = (uint8_t)norm_byte(text[j], skip_masked != 0);
        }
    }
}

__device__ inline int record_of(const uint64_t *__restrict__ rec_off, int n_rec, uint64_t p) {
    int lo = 0, hi = n_rec;  // last r with rec_off[r] <= p
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (rec_off[mid] <= p) lo = mid; else hi = mid;
    }
    return lo;
}

// Candidates for a long N-run: positions p where a run STARTS inside its record (the text is normalised: N only) and
// text[p + 5000] is still an N of the same record -- necessary for a run of more than 5000.
__global__ __launch_bounds__(256) void nrun_candidates_kernel(const uint8_t *__restrict__ text, uint64_t n,
                                                              const uint64_t *__restrict__ rec_off, int n_rec,
                                                              uint64_t *__restrict__ cand, unsigned long long cap,
                                                              unsigned long long *__restrict__ n_cand) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += stride) {
        if (text[p] != 'N') continue;
        if (p + kNRunThreshold >= n || text[p + kNRunThreshold] != 'N') continue;
        const int r = record_of(rec_off, n_rec, p);
        if (p + kNRunThreshold >= rec_off[r + 1]) continue;
        if (p > rec_off[r] && text[p - 1] == 'N') continue;  // not the start of the run
        const unsigned long long at = atomicAdd(n_cand, 1ull);
        if (at < cap) cand[at] = p;
    }
}

// One wave per candidate: where the run ends (first non-N, or the end of the record); runs of more than 5000 are kept.
__global__ __launch_bounds__(64) void nrun_extent_kernel(const uint8_t *__restrict__ text,
                                                         const uint64_t *__restrict__ rec_off, int n_rec,
                                                         const uint64_t *__restrict__ cand, unsigned long long n_cand,
                                                         uint64_t *__restrict__ runs, unsigned long long *__restrict__ n_runs) {
    const uint32_t lane = threadIdx.x;
    for (unsigned long long c = blockIdx.x; c < n_cand; c += gridDim.x) {
        const uint64_t p = cand[c];
        const uint64_t rec_end = rec_off[record_of(rec_off, n_rec, p) + 1];
        uint64_t e = rec_end;
        for (uint64_t base = p; base < rec_end; base += 64u * 16u) {
            // 16 bytes per lane, byte by byte near the ends (alignment does not matter for correctness here)
            uint32_t first_bad = 16;
            const uint64_t at = base + (uint64_t)lane * 16u;
#pragma unroll 1
            for (uint32_t j = 0; j < 16u; ++j)
                if (at + j < rec_end && text[at + j] != 'N') {
                    first_bad = j;
                    break;
                }
            const unsigned long long m = __ballot(first_bad < 16u);
            if (m) {
                const int l0 = __ffsll((long long)m) - 1;
                e = base + (uint64_t)l0 * 16u + (uint64_t)__shfl((int)first_bad, l0);
                break;
            }
        }
        if (lane == 0 && e - p > kNRunThreshold) {
            const unsigned long long at = atomicAdd(n_runs, 1ull);
            runs[2 * at] = p;
            runs[2 * at + 1] = e;
        }
    }
}

}  // namespace
}  // namespace asgart

using namespace asgart;

extern "C" int32_t asgart_prepare_data(const uint8_t *const *records, const uint64_t *record_lens, int64_t n_records,
                                       int32_t skip_masked, int32_t device, uint8_t *text_out, uint64_t *chunks,
                                       int64_t chunks_cap, int64_t *n_chunks, asgart_index **index_out) {
    if (index_out) *index_out = nullptr;
    if (n_chunks) *n_chunks = 0;
    if (!records || !record_lens || n_records <= 0 || !n_chunks || (chunks_cap && !chunks) || n_records > (1 << 24)) {
        set_error("asgart_prepare_data: bad argument");
        return ASGART_E_ARG;
    }
    std::vector<uint64_t> off((size_t)n_records + 1, 0);
    for (int64_t r = 0; r < n_records; ++r) {
        if (record_lens[r] && !records[r]) {
            set_error("asgart_prepare_data: record %lld is NULL", (long long)r);
            return ASGART_E_ARG;
        }
        off[(size_t)r + 1] = off[(size_t)r] + record_lens[r];
    }
    const uint64_t n_bases = off[(size_t)n_records], n = n_bases + 1;  // + '$' (src/bin/asgart.rs:430)
    HIP_TRY(hipSetDevice(device));
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || device < 0 || device >= n_dev) {
        (void)hipGetLastError();
        set_error("asgart_prepare_data: no usable device %d (there is no CPU fallback)", device);
        return ASGART_E_HIP;
    }
    DevBuf d_text, d_off, d_cand, d_runs, d_cnt;
    hipStream_t s = nullptr;
    std::vector<uint64_t> h_runs;
    constexpr unsigned long long kCandCap = 1ull << 22;
    int32_t rc = [&]() -> int32_t {
        HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        RC_TRY(d_text.reserve((size_t)n + 64));
        RC_TRY(d_off.reserve(((size_t)n_records + 1) * 8));
        RC_TRY(d_cand.reserve((size_t)kCandCap * 8));
        RC_TRY(d_runs.reserve((size_t)kCandCap * 16));
        RC_TRY(d_cnt.reserve(64));
        uint8_t *text = d_text.as<uint8_t>();
        for (int64_t r = 0; r < n_records; ++r)
            if (record_lens[r])
                HIP_TRY(hipMemcpyAsync(text + off[(size_t)r], records[r], (size_t)record_lens[r], hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(d_off.p, off.data(), ((size_t)n_records + 1) * 8, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemsetAsync(d_cnt.p, 0, 64, s));
        const unsigned grid = (unsigned)std::min<uint64_t>((n_bases + 256u * 16u - 1) / (256u * 16u) + 1, 1u << 16);
        normalise_kernel<<<grid, 256, 0, s>>>(text, n_bases, skip_masked);
        HIP_TRY(hipMemsetAsync(text + n_bases, '$', 1, s));
        HIP_TRY(hipMemsetAsync(text + n, 0, 64, s));
        unsigned long long *cnt = d_cnt.as<unsigned long long>();
        nrun_candidates_kernel<<<(unsigned)std::min<uint64_t>((n_bases + 255) / 256 + 1, 1u << 16), 256, 0, s>>>(
            text, n_bases, d_off.as<uint64_t>(), (int)n_records, d_cand.as<uint64_t>(), kCandCap, cnt);
        HIP_TRY(hipGetLastError());
        unsigned long long h_cnt[2] = {0, 0};
        HIP_TRY(read_back(h_cnt, cnt, 8, s));
        if (h_cnt[0] > kCandCap) {
            set_error("asgart_prepare_data: more than %llu candidate N-runs", kCandCap);
            return ASGART_E_CAP;
        }
        if (h_cnt[0]) {
            nrun_extent_kernel<<<(unsigned)std::min<unsigned long long>(h_cnt[0], 4096ull), 64, 0, s>>>(
                text, d_off.as<uint64_t>(), (int)n_records, d_cand.as<uint64_t>(), h_cnt[0], d_runs.as<uint64_t>(), cnt + 1);
            HIP_TRY(hipGetLastError());
            HIP_TRY(read_back(h_cnt + 1, cnt + 1, 8, s));
            h_runs.resize((size_t)h_cnt[1] * 2);
            if (h_cnt[1]) HIP_TRY(read_back(h_runs.data(), d_runs.p, (size_t)h_cnt[1] * 16, s));
        }
        if (text_out) HIP_TRY(read_back(text_out, text, (size_t)n, s));
        HIP_TRY(stream_sync(s));
        return 0;
    }();
    int64_t nc = 0;
    if (rc == 0) {
        // chunks per record: the pieces between its long runs (src/bin/asgart.rs:317-366), in record order (:375-395)
        std::vector<std::pair<uint64_t, uint64_t>> runs(h_runs.size() / 2);
        for (size_t j = 0; j < runs.size(); ++j) runs[j] = {h_runs[2 * j], h_runs[2 * j + 1]};
        std::sort(runs.begin(), runs.end());
        size_t j = 0;
        auto push = [&](uint64_t a, uint64_t len) {
            if (nc < chunks_cap) {
                chunks[2 * nc] = a;
                chunks[2 * nc + 1] = len;
            }
            ++nc;
        };
        for (int64_t r = 0; r < n_records; ++r) {
            const uint64_t r0 = off[(size_t)r], r1 = off[(size_t)r + 1];
            const int64_t before = nc;
            uint64_t a = r0;
            while (j < runs.size() && runs[j].first < r1) {
                if (runs[j].first > a) push(a, runs[j].first - a);
                a = runs[j].second;
                ++j;
            }
            if (r1 > a) push(a, r1 - a);
            if (nc == before) push(r0, r1 - r0);  // (`if chunks.is_empty() { chunks.push((0, len)) }`)
        }
        *n_chunks = nc;
        if (nc > chunks_cap && chunks_cap) {
            set_error("asgart_prepare_data: %lld chunks, room for %lld (call again with a larger array)", (long long)nc,
                      (long long)chunks_cap);
            rc = ASGART_E_CAP;
        }
    }
    if (rc == 0 && index_out) {
        asgart::Options opt;
        options_from_env(opt);
        const bool wide = n >= 0xFFFFFF00ull || opt.force_wide != 0;
        rc = asgart_index_create_device(d_text.p, (int64_t)n, nullptr, (int64_t)n, wide ? 8 : 4, device, index_out);
    }
    d_text.release();
    d_off.release();
    d_cand.release();
    d_runs.release();
    d_cnt.release();
    if (s) (void)hipStreamDestroy(s);
    return rc;
}
