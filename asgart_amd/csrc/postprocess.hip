// postprocess.hip -- the steps that follow the search step, at the scale of a whole-genome run.
//
// Replaces, for raw family arrays, the reference's FilterNs, ReOrder, ReduceOverlap and Sort (src/bin/asgart.rs:33-96,
// 481-562; order fixed at :738-747; ComputeScore, the optional step between ReduceOverlap and Sort, is
// asgart_compute_scores in score.hip).  SURVEY.md section 8f, N1.  The Python mirror of the same steps
// (asgart_amd/postprocess.py) is the readable restatement the tests compare against; a GRCh38-sized pass hands 350 K
// duplications in 36 K families to this chain, one of them of 14 K members through a quadratic reduction -- seconds
// of interpreter time per step there, milliseconds here:
//   FilterNs       ProtoSD::n_content (src/structs.rs:454-467) counts the N of both arms over the INCLUSIVE ranges
//                  [p ..= p + len]: a GPU kernel over the resident text, one wave per arm; the f32 quotient and the
//                  0.2 threshold on the host, bit for bit as the reference computes them;
//   ReOrder        positions swapped when left > right, the lengths are not (:39-50);
//   ReduceOverlap  _reduce iterated to a fixed point per family (:515-562), merge with its mixed lengths (:497-513);
//                  families are independent: host threads take them from a shared cursor, largest first;
//   Sort           stable by `left` (:59-64).
#include "index.hpp"

#include <algorithm>
#include <atomic>
#include <thread>

namespace asgart {

// N bases of text[p ..= p + len] for 2 * n_sd arms (arm 2j: left of duplication j, 2j + 1: right): one wave per arm
__global__ __launch_bounds__(256) void n_count_kernel(const uint8_t *__restrict__ text, uint64_t n,
                                                      const asgart_proto_sd *__restrict__ sds, uint64_t n_arms,
                                                      unsigned long long *__restrict__ out) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t a = wave; a < n_arms; a += n_waves) {
        const asgart_proto_sd sd = sds[a >> 1];
        const uint64_t p = (a & 1u) ? sd.right : sd.left, len = (a & 1u) ? sd.right_length : sd.left_length;
        unsigned long long cnt = 0;
        // (the reference indexes strand[p ..= p + len] and panics past the end; the host has checked the ranges)
        for (uint64_t j = lane; j <= len; j += 64u) {
            const uint8_t c = text[p + j];
            cnt += (c == 'N' || c == 'n') ? 1u : 0u;
        }
        for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off);
        if (lane == 0) out[a] = cnt;
    }
}

namespace {

inline bool subsegment(uint64_t xs, uint64_t xl, uint64_t ys, uint64_t yl) {  // src/bin/asgart.rs:482-487: x inside y
    return xs >= ys && xs + xl <= ys + yl;
}
inline bool overlap(uint64_t xs, uint64_t xl, uint64_t ys, uint64_t yl) {  // :489-495
    const uint64_t xe = xs + xl, ye = ys + yl;
    return (xs >= ys && xs <= ye && xe >= ye) || (ys >= xs && ys <= xe && ye >= xe);
}

// one pass of _reduce (:516-549) from `in` into `out` (cleared first)
void reduce_pass(const std::vector<asgart_proto_sd> &in, std::vector<asgart_proto_sd> &out) {
    out.clear();
    for (const asgart_proto_sd &x : in) {
        bool placed = false;
        for (asgart_proto_sd &y : out) {
            if (subsegment(x.left, x.left_length, y.left, y.left_length) &&
                subsegment(x.right, x.right_length, y.right, y.right_length)) {  // x inside y: dropped
                placed = true;
                break;
            }
            if (subsegment(y.left, y.left_length, x.left, x.left_length) &&
                subsegment(y.right, y.right_length, x.right, x.right_length)) {  // x contains y: takes its place
                y = x;
                placed = true;
                break;
            }
            if (overlap(x.left, x.left_length, y.left, y.left_length) &&
                overlap(x.right, x.right_length, y.right, y.right_length)) {
                // merge (:497-513): x contributes its LEFT length to both arms, y its RIGHT length to both
                asgart_proto_sd z;
                z.left = std::min(x.left, y.left);
                z.left_length = std::max(x.left + x.left_length, y.left + y.right_length) - z.left;
                z.right = std::min(x.right, y.right);
                z.right_length = std::max(x.right + x.left_length, y.right + y.right_length) - z.right;
                y = z;
                placed = true;
                break;
            }
        }
        if (!placed) out.push_back(x);
    }
}

}  // namespace
}  // namespace asgart

using namespace asgart;

extern "C" int32_t asgart_post_process(asgart_index *idx, const uint64_t *fam_offsets, uint64_t n_families,
                                       const asgart_proto_sd *sds, int32_t threads, asgart_families **out) {
    if (!out) {
        set_error("out is NULL");
        return ASGART_E_ARG;
    }
    *out = nullptr;
    if (!idx || !fam_offsets || (n_families && fam_offsets[n_families] && !sds)) {
        set_error("bad argument");
        return ASGART_E_ARG;
    }
    const uint64_t n_sd = fam_offsets[n_families];
    const uint64_t n = (uint64_t)idx->n;
    for (uint64_t f = 0; f < n_families; ++f)
        if (fam_offsets[f] > fam_offsets[f + 1]) {
            set_error("family offsets must not decrease");
            return ASGART_E_ARG;
        }
    for (uint64_t j = 0; j < n_sd; ++j) {
        // strand[p ..= p + len] must exist (the reference panics otherwise); len == 0 would divide by zero into NaN,
        // which the reference's `<= 0.2` then rejects -- kept as it is
        if (sds[j].left > n || sds[j].left_length >= n - sds[j].left || sds[j].right > n || sds[j].right_length >= n - sds[j].right) {
            set_error("duplication %llu reaches past the text (the reference's n_content indexes [p ..= p + length])",
                      (unsigned long long)j);
            return ASGART_E_ARG;
        }
    }
    asgart_families *res = new (std::nothrow) asgart_families();
    if (!res) {
        set_error("out of host memory");
        return ASGART_E_OOM;
    }
    res->fam_offsets.assign(1, 0);
    if (n_sd == 0) {
        *out = res;
        return 0;
    }
    // ---- FilterNs: N counts of every arm on the GPU ----------------------------------------------------------------
    std::vector<unsigned long long> n_cnt((size_t)n_sd * 2);
    {
        if (idx->poisoned.load()) {
            delete res;
            REFUSE_POISONED(idx);
        }
        HIP_TRY(hipSetDevice(idx->device));
        idx->acquire_all();  // (the text must stay where it is)
        struct Unlock {
            asgart_index *i;
            ~Unlock() { i->release_all(); }
        } unlock{idx};
        hipStream_t s = idx->ctx[0].stream;
        DevBuf d_sds, d_out;
        int32_t rc = [&]() -> int32_t {
            RC_TRY(d_sds.reserve((size_t)n_sd * sizeof(asgart_proto_sd)));
            RC_TRY(d_out.reserve((size_t)n_sd * 2 * 8));
            HIP_TRY(hipMemcpyAsync(d_sds.p, sds, (size_t)n_sd * sizeof(asgart_proto_sd), hipMemcpyHostToDevice, s));
            const uint64_t n_arms = n_sd * 2;
            const unsigned grid = (unsigned)std::min<uint64_t>((n_arms + 3) / 4, 256ull * 32ull);
            n_count_kernel<<<grid, 256, 0, s>>>(idx->d_text, n, d_sds.as<asgart_proto_sd>(), n_arms, d_out.as<unsigned long long>());
            HIP_TRY(hipGetLastError());
            HIP_TRY(read_back(n_cnt.data(), d_out.p, (size_t)n_arms * 8, s));
            HIP_TRY(stream_sync(s));
            return 0;
        }();
        d_sds.release();
        d_out.release();
        if (rc != 0) {
            delete res;
            return rc;
        }
    }
    // ---- per family: retain, re-order, reduce, sort ------------------------------------------------------------------
    std::vector<std::vector<asgart_proto_sd>> fams((size_t)n_families);
    std::vector<uint32_t> order;
    for (uint64_t f = 0; f < n_families; ++f) {
        std::vector<asgart_proto_sd> &fam = fams[(size_t)f];
        for (uint64_t j = fam_offsets[f]; j < fam_offsets[f + 1]; ++j) {
            const float a = (float)n_cnt[2 * j] / (float)sds[j].left_length;       // f32, as the reference computes it
            const float b = (float)n_cnt[2 * j + 1] / (float)sds[j].right_length;
            const float m = a > b ? a : (b > a ? b : (a == a ? a : b));             // f32::max (ignores a NaN operand)
            if (m <= 0.2f) {
                asgart_proto_sd sd = sds[j];
                if (sd.left > sd.right) std::swap(sd.left, sd.right);                // ReOrder: the lengths stay
                fam.push_back(sd);
            }
        }
        if (!fam.empty()) order.push_back((uint32_t)f);
    }
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return fams[a].size() > fams[b].size(); });
    std::atomic<size_t> cursor{0};
    auto work = [&]() {
        std::vector<asgart_proto_sd> tmp;
        for (;;) {
            const size_t at = cursor.fetch_add(1);
            if (at >= order.size()) return;
            std::vector<asgart_proto_sd> &fam = fams[order[at]];
            size_t old_size = fam.size();
            reduce_pass(fam, tmp);
            fam.swap(tmp);
            while (fam.size() < old_size) {
                old_size = fam.size();
                reduce_pass(fam, tmp);
                fam.swap(tmp);
            }
            std::stable_sort(fam.begin(), fam.end(), [](const asgart_proto_sd &a, const asgart_proto_sd &b) { return a.left < b.left; });
        }
    };
    int n_thr = threads > 0 ? threads : (int)std::thread::hardware_concurrency();
    n_thr = std::max(1, std::min<int>(n_thr, 64));
    n_thr = (int)std::min<size_t>((size_t)n_thr, std::max<size_t>(order.size(), 1));
    std::vector<std::thread> pool;
    for (int t = 1; t < n_thr; ++t) pool.emplace_back(work);
    work();
    for (auto &t : pool) t.join();
    for (uint64_t f = 0; f < n_families; ++f) {
        const std::vector<asgart_proto_sd> &fam = fams[(size_t)f];
        if (fam.empty()) continue;  // FilterNs drops the families it emptied (:91-95)
        res->sds.insert(res->sds.end(), fam.begin(), fam.end());
        res->fam_offsets.push_back(res->sds.size());
        res->fam_keys.push_back(f);
    }
    *out = res;
    return 0;
}
