// sa_build.hip -- suffix-array construction on the GPU.
//
// Replaces `divsufsort64` (reference src/divsufsort.rs:10, called once per run
// from src/bin/asgart.rs:473-479, single-threaded C in the reference build,
// build.rs:5-11).  The suffix array of a text is unique, so the algorithm is
// free: this is prefix doubling (Manber-Myers / Larsson-Sadakane) with group
// filtering, every round a device-wide LSD radix sort (rocPRIM, a plain library
// sort) of only the suffixes that are still tied:
//
//   round 0   sort all suffixes by their first h0 characters packed into 63 bits
//             (h0 = 21 for the DNA alphabet {$,A,C,G,N,T}, 7 for arbitrary bytes);
//             characters are stored +1 so that "past the end" (0) sorts first.
//   round r   for the suffixes in groups of size > 1 only: key = (rank[i],
//             rank[i+h]+1 or 0), sort, write back into the same SA slots, split
//             groups, h *= 2.
//
// rank[i] is the first SA slot of i's group, so keys of one group are contiguous
// and the sorted active list maps back onto the active slots in order.
#include "index.hpp"

#include <rocprim/rocprim.hpp>

#include <algorithm>

namespace asgart {

namespace {

inline unsigned grid_for(uint64_t n, unsigned block = 256) {
    return (unsigned)((n + block - 1) / block);
}

template <bool DNA>
__device__ inline uint64_t initial_key(const uint8_t *__restrict__ text, uint64_t n, uint64_t i) {
    uint64_t q = 0;
    if (DNA) {
        for (int j = 0; j < 21; ++j) {
            uint64_t c = (i + j < n) ? (uint64_t)base_code(text[i + j]) + 1u : 0u;
            q = (q << 3) | c;
        }
    } else {
        for (int j = 0; j < 7; ++j) {
            uint64_t c = (i + j < n) ? (uint64_t)text[i + j] + 1u : 0u;
            q = (q << 9) | c;
        }
    }
    return q;
}

__global__ __launch_bounds__(256) void byte_histogram_for_sa(const uint8_t *__restrict__ text, uint64_t n,
                                                             unsigned long long *__restrict__ hist) {
    __shared__ unsigned int sh[256];
    sh[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) atomicAdd(&sh[text[i]], 1u);
    __syncthreads();
    if (sh[threadIdx.x]) atomicAdd(&hist[threadIdx.x], (unsigned long long)sh[threadIdx.x]);
}

template <class IdxT, bool DNA>
__global__ __launch_bounds__(256) void init_keys_kernel(const uint8_t *__restrict__ text, uint64_t n,
                                                        uint64_t *__restrict__ keys,
                                                        IdxT *__restrict__ vals) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    keys[i] = initial_key<DNA>(text, n, i);
    vals[i] = (IdxT)i;
}

// after a sort: group heads among the m sorted entries; head[j] = slot of j if j starts a group
template <class IdxT>
__global__ __launch_bounds__(256) void mark_heads_kernel(const uint64_t *__restrict__ k1,
                                                         const uint64_t *__restrict__ k2,
                                                         const IdxT *__restrict__ slots, uint64_t m,
                                                         IdxT *__restrict__ head, uint64_t slot_base) {
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    bool is_head = j == 0 || k1[j] != k1[j - 1] || (k2 && k2[j] != k2[j - 1]);
    IdxT slot = slots ? slots[j] : (IdxT)(slot_base + j);
    head[j] = is_head ? slot : (IdxT)0;
}

// grp[j] = slot of the head of j's group (inclusive max-scan of head[]).
// Writes sa, rank and the "still tied" flag.
template <class IdxT>
__global__ __launch_bounds__(256) void apply_round_kernel(const IdxT *__restrict__ vals,
                                                          const IdxT *__restrict__ grp,
                                                          const IdxT *__restrict__ slots, uint64_t m,
                                                          IdxT *__restrict__ sa,
                                                          IdxT *__restrict__ rank,
                                                          uint8_t *__restrict__ tied, uint64_t slot_base) {
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    const IdxT x = vals[j];
    const IdxT slot = slots ? slots[j] : (IdxT)(slot_base + j);
    const IdxT g = grp[j];
    sa[slot] = x;
    rank[x] = g;
    const bool head = g == slot;
    const bool next_head = (j + 1 == m) || grp[j + 1] != g;
    tied[j] = (head && next_head) ? 0 : 1;
}

__global__ __launch_bounds__(256) void widen_kernel(const uint32_t *__restrict__ in,
                                                    int64_t *__restrict__ out, uint64_t cnt) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < cnt) out[i] = (int64_t)in[i];
}

// keys of a doubling round for the tied slots
template <class IdxT>
__global__ __launch_bounds__(256) void round_keys_kernel(const IdxT *__restrict__ sa,
                                                         const IdxT *__restrict__ rank,
                                                         const IdxT *__restrict__ slots, uint64_t m,
                                                         uint64_t n, uint64_t h,
                                                         uint64_t *__restrict__ k1,
                                                         uint64_t *__restrict__ k2,
                                                         IdxT *__restrict__ vals) {
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    const uint64_t x = sa[slots[j]];
    const uint64_t a = rank[x];
    const uint64_t b = (x + h < n) ? (uint64_t)rank[x + h] + 1u : 0u;
    if (sizeof(IdxT) == 4) {
        k1[j] = (a << 32) | b;  // composite: one sort
    } else {
        k1[j] = a;
        k2[j] = b;
    }
    vals[j] = (IdxT)x;
}

template <class T>
struct Dbuf {
    DevBuf a, b;
    rocprim::double_buffer<T> db{nullptr, nullptr};
    int32_t reserve(size_t count) {
        RC_TRY(a.reserve(count * sizeof(T)));
        RC_TRY(b.reserve(count * sizeof(T)));
        db = rocprim::double_buffer<T>(a.as<T>(), b.as<T>());
        return 0;
    }
    void release() {
        a.release();
        b.release();
    }
};

template <class IdxT>
int32_t sort_pairs(DevBuf &temp, rocprim::double_buffer<uint64_t> &keys,
                   rocprim::double_buffer<IdxT> &vals, uint64_t m, unsigned begin_bit,
                   unsigned end_bit, hipStream_t s) {
    size_t bytes = 0;
    HIP_TRY(rocprim::radix_sort_pairs(nullptr, bytes, keys, vals, (size_t)m, begin_bit, end_bit, s));
    RC_TRY(temp.reserve(bytes));
    HIP_TRY(rocprim::radix_sort_pairs(temp.p, bytes, keys, vals, (size_t)m, begin_bit, end_bit, s));
    return 0;
}

inline unsigned bits_for(uint64_t v) {
    unsigned b = 1;
    while (b < 64 && (v >> b)) ++b;
    return b;
}

template <class IdxT>
int32_t build_t(const uint8_t *d_text, int64_t n_, IdxT *d_sa, bool dna, hipStream_t s) {
    const uint64_t n = (uint64_t)n_;
    Dbuf<uint64_t> K1, K2;
    Dbuf<IdxT> V;
    DevBuf rank_b, head_b, grp_b, tied_b, slots_a, slots_b, temp, cnt_b;
    auto cleanup = [&]() {
        K1.release(); K2.release(); V.release();
        rank_b.release(); head_b.release(); grp_b.release(); tied_b.release();
        slots_a.release(); slots_b.release(); temp.release(); cnt_b.release();
    };
    int32_t rc = [&]() -> int32_t {
        RC_TRY(K1.reserve(n));
        RC_TRY(V.reserve(n));
        RC_TRY(rank_b.reserve(n * sizeof(IdxT)));
        RC_TRY(head_b.reserve(n * sizeof(IdxT)));
        RC_TRY(grp_b.reserve(n * sizeof(IdxT)));
        RC_TRY(tied_b.reserve(n));
        RC_TRY(cnt_b.reserve(16));
        IdxT *rank = rank_b.as<IdxT>(), *head = head_b.as<IdxT>(), *grp = grp_b.as<IdxT>();
        uint8_t *tied = tied_b.as<uint8_t>();
        size_t *d_count = cnt_b.as<size_t>();

        // ---- round 0 -----------------------------------------------------------
        if (dna) init_keys_kernel<IdxT, true><<<grid_for(n), 256, 0, s>>>(d_text, n, K1.db.current(), V.db.current());
        else init_keys_kernel<IdxT, false><<<grid_for(n), 256, 0, s>>>(d_text, n, K1.db.current(), V.db.current());
        HIP_TRY(hipGetLastError());
        RC_TRY(sort_pairs<IdxT>(temp, K1.db, V.db, n, 0, 63, s));
        uint64_t h = dna ? 21 : 7;
        uint64_t m = n;
        const IdxT *slots = nullptr;  // round 0: slot j == j
        bool two_keys = false;
        for (;;) {
            mark_heads_kernel<IdxT><<<grid_for(m), 256, 0, s>>>(
                K1.db.current(), two_keys ? K2.db.current() : nullptr, slots, m, head, 0);
            HIP_TRY(hipGetLastError());
            size_t bytes = 0;
            HIP_TRY(rocprim::inclusive_scan(nullptr, bytes, head, grp, (size_t)m, rocprim::maximum<IdxT>(), s));
            RC_TRY(temp.reserve(bytes));
            HIP_TRY(rocprim::inclusive_scan(temp.p, bytes, head, grp, (size_t)m, rocprim::maximum<IdxT>(), s));
            apply_round_kernel<IdxT><<<grid_for(m), 256, 0, s>>>(V.db.current(), grp, slots, m, d_sa, rank, tied, 0);
            HIP_TRY(hipGetLastError());
            // ---- compact the slots that are still tied ---------------------------
            DevBuf &outb = (slots == slots_a.as<IdxT>() && slots) ? slots_b : slots_a;
            RC_TRY(outb.reserve(m * sizeof(IdxT)));
            bytes = 0;
            if (!slots) {
                rocprim::counting_iterator<IdxT> it((IdxT)0);
                HIP_TRY(rocprim::select(nullptr, bytes, it, tied, outb.as<IdxT>(), d_count, (size_t)m, s));
                RC_TRY(temp.reserve(bytes));
                HIP_TRY(rocprim::select(temp.p, bytes, it, tied, outb.as<IdxT>(), d_count, (size_t)m, s));
            } else {
                HIP_TRY(rocprim::select(nullptr, bytes, slots, tied, outb.as<IdxT>(), d_count, (size_t)m, s));
                RC_TRY(temp.reserve(bytes));
                HIP_TRY(rocprim::select(temp.p, bytes, slots, tied, outb.as<IdxT>(), d_count, (size_t)m, s));
            }
            size_t h_count = 0;
            HIP_TRY(read_back(&h_count, d_count, sizeof(size_t), s));
            HIP_TRY(stream_sync(s));
            if (h_count == 0) break;
            if (h >= n) {
                set_error("internal: suffix sort did not converge");
                return ASGART_E_CAP;
            }
            m = h_count;
            slots = outb.as<IdxT>();
            // ---- doubling round ---------------------------------------------------
            two_keys = sizeof(IdxT) == 8;
            if (two_keys) RC_TRY(K2.reserve(m));
            round_keys_kernel<IdxT><<<grid_for(m), 256, 0, s>>>(
                d_sa, rank, slots, m, n, h, K1.db.current(), two_keys ? K2.db.current() : nullptr,
                V.db.current());
            HIP_TRY(hipGetLastError());
            const unsigned nb = bits_for(n + 1);
            if (!two_keys) {
                // composite (rank, next rank): low word needs nb bits, high word nb bits
                RC_TRY(sort_pairs<IdxT>(temp, K1.db, V.db, m, 0, 32 + nb, s));
            } else {
                // LSD over two 64-bit words: stable sort by k2, then by k1, carrying the
                // other word along by sorting (key, index) and gathering -- done here by
                // sorting k2 with values = positions, then permuting k1/vals.
                set_error("texts of 2^32 bytes or more: GPU suffix sort not implemented; pass SA");
                return ASGART_E_CAP;
            }
            h *= 2;
        }
        return 0;
    }();
    cleanup();
    return rc;
}

// ---- texts of 2^32 bytes and more ------------------------------------------------------
// Same prefix doubling with 64-bit suffix numbers, laid out so that it fits next to a 6-Gbp index
// even when most suffixes stay tied for several rounds (two similar genomes in one text):
//   round 0 is done class by class -- suffixes starting with the same two bytes (small alphabets;
//   one byte otherwise), a few hundred million each: select the class's positions, key them by their
//   first 21 bases, sort, split into groups, write that stretch of the suffix array.  Classes in
//   byte order concatenate to the order by 21-mer;
//   doubling rounds sort the still-tied suffixes by the pair (rank[i], rank[i + h]), which no longer
//   fits one 64-bit key: LSD over the two words (stable sort by the second, then by the first),
//   carried by a permutation -- and they do it batch by batch (whole groups, <= kWideBatch suffixes):
//   groups are independent, and rank[] of a group only changes when its own batch is applied, so a
//   batch may already see the refined ranks of earlier batches;
//   the list of tied slots is compacted in place (the survivors of a batch never outnumber it).
constexpr uint64_t kWideBatch = 1ull << 29;

struct ClassPred {
    const uint8_t *text;
    uint64_t n;
    int c0, c1;  // c1: second byte, -1 = "the suffix is one byte long", -2 = any
    __device__ bool operator()(uint64_t i) const {
        if (text[i] != (uint8_t)c0) return false;
        if (c1 == -2) return true;
        if (i + 1 >= n) return c1 == -1;
        return c1 >= 0 && text[i + 1] == (uint8_t)c1;
    }
};

template <bool DNA>
__global__ __launch_bounds__(256) void class_keys_kernel(const uint8_t *__restrict__ text, uint64_t n,
                                                         const uint64_t *__restrict__ vals, uint64_t m,
                                                         uint64_t *__restrict__ keys) {
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < m) keys[j] = initial_key<DNA>(text, n, vals[j]);
}

__global__ __launch_bounds__(256) void iota_kernel(uint64_t *__restrict__ p, uint64_t m) {
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < m) p[j] = j;
}

__global__ __launch_bounds__(256) void gather64_kernel(const uint64_t *__restrict__ src,
                                                       const uint64_t *__restrict__ perm, uint64_t m,
                                                       uint64_t *__restrict__ dst) {
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < m) dst[j] = src[perm[j]];
}

__global__ __launch_bounds__(256) void offset_slots_kernel(uint64_t *__restrict__ slots, uint64_t m, uint64_t base) {
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < m) slots[j] += base;
}

// pair histogram over a small alphabet: code[byte] in 0..S-1 (255: absent), bins S * (S + 1)
__global__ __launch_bounds__(256) void pair_histogram_kernel(const uint8_t *__restrict__ text, uint64_t n,
                                                             const uint8_t *__restrict__ code, int S,
                                                             unsigned long long *__restrict__ hist) {
    __shared__ unsigned int sh[16 * 17];
    __shared__ uint8_t s_code[256];
    s_code[threadIdx.x] = code[threadIdx.x];
    for (int j = threadIdx.x; j < 16 * 17; j += 256) sh[j] = 0;
    __syncthreads();
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int a = s_code[text[i]];
        const int b = i + 1 < n ? s_code[text[i + 1]] + 1 : 0;
        atomicAdd(&sh[a * (S + 1) + b], 1u);
    }
    __syncthreads();
    for (int j = threadIdx.x; j < S * (S + 1); j += 256)
        if (sh[j]) atomicAdd(&hist[j], (unsigned long long)sh[j]);
}

// first entry of the next group at or after position p of the tied list: groups are runs of equal
// rank[sa[slot]] (a run of N's is one group of millions), and a batch must end on a group boundary
__global__ void group_end_kernel(const uint64_t *__restrict__ slots, const uint64_t *__restrict__ sa,
                                 const uint64_t *__restrict__ rank, uint64_t m, uint64_t p,
                                 uint64_t *__restrict__ out) {
    if (threadIdx.x || blockIdx.x) return;
    uint64_t lo = p, hi = m;  // rank[sa[slots[j]]] is non-decreasing in j: upper bound of the group of p - 1
    if (p > 0 && p < m) {
        const uint64_t g = rank[sa[slots[p - 1]]];
        while (lo < hi) {
            const uint64_t mid = lo + ((hi - lo) >> 1);
            if (rank[sa[slots[mid]]] <= g) lo = mid + 1; else hi = mid;
        }
    }
    *out = lo;
}

int32_t build_wide(const uint8_t *d_text, int64_t n_, uint64_t *d_sa, bool dna, hipStream_t s, uint64_t batch) {
    using IdxT = uint64_t;
    const uint64_t n = (uint64_t)n_;
    Dbuf<uint64_t> K1, K2, V, Pm;
    DevBuf rank_b, grp_b, tied_b, list_b, out_b, temp, cnt_b, hist_b, code_b, k1o, k2o, vo;
    auto cleanup = [&]() {
        K1.release(); K2.release(); V.release(); Pm.release();
        rank_b.release(); grp_b.release(); tied_b.release(); list_b.release(); out_b.release();
        temp.release(); cnt_b.release(); hist_b.release(); code_b.release(); k1o.release(); k2o.release(); vo.release();
    };
    int32_t rc = [&]() -> int32_t {
        // ---- classes ------------------------------------------------------------------------
        RC_TRY(hist_b.reserve(512 * sizeof(unsigned long long)));
        unsigned long long hist[512];
        HIP_TRY(hipMemsetAsync(hist_b.p, 0, sizeof(hist), s));
        unsigned blocks = grid_for(n, 256 * 16);
        if (blocks > 4096) blocks = 4096;
        byte_histogram_for_sa<<<blocks, 256, 0, s>>>(d_text, n, hist_b.as<unsigned long long>());
        HIP_TRY(hipGetLastError());
        HIP_TRY(read_back(hist, hist_b.p, 256 * 8, s));
        HIP_TRY(stream_sync(s));
        std::vector<int> syms;
        for (int c = 0; c < 256; ++c)
            if (hist[c]) syms.push_back(c);
        const int S = (int)syms.size();
        struct Cls { int c0, c1; uint64_t cnt; };
        std::vector<Cls> classes;
        if (S <= 16) {
            uint8_t code[256];
            memset(code, 255, sizeof code);
            for (int j = 0; j < S; ++j) code[syms[j]] = (uint8_t)j;
            RC_TRY(code_b.reserve(256));
            HIP_TRY(hipMemcpyAsync(code_b.p, code, 256, hipMemcpyHostToDevice, s));
            HIP_TRY(hipMemsetAsync(hist_b.p, 0, sizeof(hist), s));
            pair_histogram_kernel<<<blocks, 256, 0, s>>>(d_text, n, code_b.as<uint8_t>(), S, hist_b.as<unsigned long long>());
            HIP_TRY(hipGetLastError());
            HIP_TRY(read_back(hist, hist_b.p, (size_t)S * (S + 1) * 8, s));
            HIP_TRY(stream_sync(s));
            for (int a = 0; a < S; ++a)
                for (int b = 0; b <= S; ++b)   // b == 0: the one-byte suffix, sorts first
                    if (hist[a * (S + 1) + b]) classes.push_back({syms[a], b ? syms[b - 1] : -1, hist[a * (S + 1) + b]});
        } else {
            for (int c : syms) classes.push_back({c, -2, hist[c]});
        }
        uint64_t biggest = 0;
        for (const Cls &c : classes) biggest = std::max(biggest, c.cnt);
        if (biggest >= 0xFFFFFF00ull) {
            set_error("suffix sort: a first-bytes class holds %llu suffixes (>= 2^32); not supported",
                      (unsigned long long)biggest);
            return ASGART_E_CAP;
        }
        RC_TRY(rank_b.reserve(n * 8));
        RC_TRY(cnt_b.reserve(16));
        uint64_t *rank = rank_b.as<uint64_t>();
        size_t *d_count = cnt_b.as<size_t>();
        RC_TRY(K1.reserve(biggest));
        RC_TRY(V.reserve(biggest));
        RC_TRY(grp_b.reserve(biggest * 8));
        RC_TRY(tied_b.reserve(biggest));
        RC_TRY(out_b.reserve(biggest * 8));
        RC_TRY(list_b.reserve(n * 8));  // the tied slots, ascending; at most every slot
        uint64_t *grp = grp_b.as<uint64_t>();
        uint8_t *tied = tied_b.as<uint8_t>();
        uint64_t *list = list_b.as<uint64_t>();
        uint64_t m = 0;
        size_t bytes = 0;
        rocprim::counting_iterator<uint64_t> it(0);
        // ---- round 0, one class at a time --------------------------------------------------------
        uint64_t slot_base = 0;
        for (const Cls &c : classes) {
            const uint64_t cnt = c.cnt;
            ClassPred pred{d_text, n, c.c0, c.c1};
            bytes = 0;
            HIP_TRY(rocprim::select(nullptr, bytes, it, V.db.current(), d_count, (size_t)n, pred, s));
            RC_TRY(temp.reserve(bytes));
            HIP_TRY(rocprim::select(temp.p, bytes, it, V.db.current(), d_count, (size_t)n, pred, s));
            if (dna) class_keys_kernel<true><<<grid_for(cnt), 256, 0, s>>>(d_text, n, V.db.current(), cnt, K1.db.current());
            else class_keys_kernel<false><<<grid_for(cnt), 256, 0, s>>>(d_text, n, V.db.current(), cnt, K1.db.current());
            HIP_TRY(hipGetLastError());
            RC_TRY(sort_pairs<IdxT>(temp, K1.db, V.db, cnt, 0, 63, s));
            mark_heads_kernel<IdxT><<<grid_for(cnt), 256, 0, s>>>(K1.db.current(), nullptr, nullptr, cnt, grp, slot_base);
            HIP_TRY(hipGetLastError());
            bytes = 0;
            HIP_TRY(rocprim::inclusive_scan(nullptr, bytes, grp, grp, (size_t)cnt, rocprim::maximum<IdxT>(), s));
            RC_TRY(temp.reserve(bytes));
            HIP_TRY(rocprim::inclusive_scan(temp.p, bytes, grp, grp, (size_t)cnt, rocprim::maximum<IdxT>(), s));
            apply_round_kernel<IdxT><<<grid_for(cnt), 256, 0, s>>>(V.db.current(), grp, nullptr, cnt, d_sa, rank, tied, slot_base);
            HIP_TRY(hipGetLastError());
            // tied slots of this class (class-local numbers + slot_base) -> appended to the list
            bytes = 0;
            HIP_TRY(rocprim::select(nullptr, bytes, it, tied, out_b.as<uint64_t>(), d_count, (size_t)cnt, s));
            RC_TRY(temp.reserve(bytes));
            HIP_TRY(rocprim::select(temp.p, bytes, it, tied, out_b.as<uint64_t>(), d_count, (size_t)cnt, s));
            size_t h_count = 0;
            HIP_TRY(read_back(&h_count, d_count, sizeof(size_t), s));
            HIP_TRY(stream_sync(s));
            if (h_count) {
                offset_slots_kernel<<<grid_for(h_count), 256, 0, s>>>(out_b.as<uint64_t>(), h_count, slot_base);
                HIP_TRY(hipGetLastError());
                HIP_TRY(hipMemcpyAsync(list + m, out_b.p, h_count * 8, hipMemcpyDeviceToDevice, s));
                m += h_count;
            }
            slot_base += cnt;
        }
        HIP_TRY(stream_sync(s));
        K1.release(); V.release(); grp_b.release(); tied_b.release(); out_b.release();
        // ---- doubling rounds over the tied suffixes, batch by batch --------------------------------
        uint64_t h = dna ? 21 : 7;
        const unsigned nb = bits_for(n + 1);
        while (m) {
            if (h >= n) {
                set_error("internal: suffix sort did not converge");
                return ASGART_E_CAP;
            }
            const uint64_t B = std::min<uint64_t>(m, batch ? batch : kWideBatch);
            // a batch ends on a group boundary, so it may exceed B by the tail of its last group
            uint64_t done = 0, kept = 0;
            while (done < m) {
                uint64_t end = std::min<uint64_t>(m, done + B);
                if (end < m) {
                    group_end_kernel<<<1, 1, 0, s>>>(list, d_sa, rank, m, end, reinterpret_cast<uint64_t *>(d_count));
                    HIP_TRY(hipGetLastError());
                    HIP_TRY(read_back(&end, d_count, 8, s));
                    HIP_TRY(stream_sync(s));
                }
                const uint64_t bm = end - done;
                const uint64_t *slots = list + done;
                RC_TRY(k1o.reserve(bm * 8));
                RC_TRY(k2o.reserve(bm * 8));
                RC_TRY(vo.reserve(bm * 8));
                RC_TRY(K1.reserve(bm));
                RC_TRY(K2.reserve(bm));
                RC_TRY(Pm.reserve(bm));
                RC_TRY(grp_b.reserve(bm * 8));
                RC_TRY(tied_b.reserve(bm));
                RC_TRY(out_b.reserve(bm * 8));
                grp = grp_b.as<uint64_t>();
                tied = tied_b.as<uint8_t>();
                round_keys_kernel<IdxT><<<grid_for(bm), 256, 0, s>>>(d_sa, rank, slots, bm, n, h, k1o.as<uint64_t>(),
                                                                     k2o.as<uint64_t>(), vo.as<uint64_t>());
                HIP_TRY(hipGetLastError());
                // LSD: stable sort of a permutation by the second word, then by the first
                HIP_TRY(hipMemcpyAsync(K2.db.current(), k2o.p, bm * 8, hipMemcpyDeviceToDevice, s));
                iota_kernel<<<grid_for(bm), 256, 0, s>>>(Pm.db.current(), bm);
                RC_TRY(sort_pairs<IdxT>(temp, K2.db, Pm.db, bm, 0, nb + 1, s));
                gather64_kernel<<<grid_for(bm), 256, 0, s>>>(k1o.as<uint64_t>(), Pm.db.current(), bm, K1.db.current());
                RC_TRY(sort_pairs<IdxT>(temp, K1.db, Pm.db, bm, 0, nb, s));
                // sorted (k1, k2, suffix): k1 is K1.current; the other two through the permutation
                gather64_kernel<<<grid_for(bm), 256, 0, s>>>(k2o.as<uint64_t>(), Pm.db.current(), bm, K2.db.current());
                gather64_kernel<<<grid_for(bm), 256, 0, s>>>(vo.as<uint64_t>(), Pm.db.current(), bm, Pm.db.alternate());
                HIP_TRY(hipGetLastError());
                const uint64_t *vs = Pm.db.alternate();
                mark_heads_kernel<IdxT><<<grid_for(bm), 256, 0, s>>>(K1.db.current(), K2.db.current(), slots, bm, grp, 0);
                HIP_TRY(hipGetLastError());
                bytes = 0;
                HIP_TRY(rocprim::inclusive_scan(nullptr, bytes, grp, grp, (size_t)bm, rocprim::maximum<IdxT>(), s));
                RC_TRY(temp.reserve(bytes));
                HIP_TRY(rocprim::inclusive_scan(temp.p, bytes, grp, grp, (size_t)bm, rocprim::maximum<IdxT>(), s));
                apply_round_kernel<IdxT><<<grid_for(bm), 256, 0, s>>>(vs, grp, slots, bm, d_sa, rank, tied, 0);
                HIP_TRY(hipGetLastError());
                bytes = 0;
                HIP_TRY(rocprim::select(nullptr, bytes, slots, tied, out_b.as<uint64_t>(), d_count, (size_t)bm, s));
                RC_TRY(temp.reserve(bytes));
                HIP_TRY(rocprim::select(temp.p, bytes, slots, tied, out_b.as<uint64_t>(), d_count, (size_t)bm, s));
                size_t h_count = 0;
                HIP_TRY(read_back(&h_count, d_count, sizeof(size_t), s));
                HIP_TRY(stream_sync(s));
                // in-place compaction of the list: the survivors trail the read position
                if (h_count) HIP_TRY(hipMemcpyAsync(list + kept, out_b.p, h_count * 8, hipMemcpyDeviceToDevice, s));
                kept += h_count;
                done = end;
            }
            HIP_TRY(stream_sync(s));
            m = kept;
            h *= 2;
        }
        return 0;
    }();
    cleanup();
    return rc;
}

}  // namespace

// Segment placement of the extension step: ascending radix sort of (tier, longest-first) keys.
// keys/vals hold 2*n entries each (second half = alternate buffer).
int32_t sort_segments(Workspace &w, uint32_t *keys, uint32_t *vals, uint64_t n, hipStream_t s,
                      const uint32_t **sorted_vals, const uint32_t **sorted_keys, bool ties_by_value) {
    rocprim::double_buffer<uint32_t> kd(keys, keys + n), vd(vals, vals + n);
    size_t bytes = 0;
    HIP_TRY(rocprim::radix_sort_pairs(nullptr, bytes, kd, vd, (size_t)n, 0, 32, s));
    RC_TRY(w.sort_tmp.reserve(bytes));
    // (the segment list is appended to by many workgroups: its order differs from call to call.  When several
    // shards must agree on the sorted order, equal keys are first put in order of their values -- the sort below
    // is stable)
    if (ties_by_value) HIP_TRY(rocprim::radix_sort_pairs(w.sort_tmp.p, bytes, vd, kd, (size_t)n, 0, 32, s));
    HIP_TRY(rocprim::radix_sort_pairs(w.sort_tmp.p, bytes, kd, vd, (size_t)n, 0, 32, s));
    *sorted_vals = vd.current();
    *sorted_keys = kd.current();
    return 0;
}

namespace {
__global__ void rec_keys32_kernel(const SdRec *__restrict__ recs, uint64_t n, uint32_t *__restrict__ k32,
                                  uint32_t *__restrict__ idx) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        k32[i] = recs[i].create_seq;
        idx[i] = (uint32_t)i;
    }
}
__global__ void rec_keys64_kernel(const SdRec *__restrict__ recs, const uint32_t *__restrict__ idx, uint64_t n,
                                  uint64_t *__restrict__ k64) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) {
        const SdRec &r = recs[idx[j]];
        k64[j] = ((uint64_t)r.g_start << 32) | r.fam_seq;
    }
}
__global__ void rec_gather_kernel(const SdRec *__restrict__ recs, const uint32_t *__restrict__ idx, uint64_t n,
                                  SdRec *__restrict__ out) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) out[j] = recs[idx[j]];
}
}  // namespace

// Output records in the reference's order: LSD over the composite key -- stable radix sort by the
// creation number, then by (segment start probe, family ordinal) -- on an index, then one gather.
int32_t sort_records(Workspace &w, const SdRec *recs, uint64_t n, hipStream_t s) {
    if (n == 0) return 0;
    if (n >= 0xFFFFFFFFull) {
        set_error("more than 2^32 output records in one call");
        return ASGART_E_CAP;
    }
    RC_TRY(w.rec_k32.reserve((size_t)n * 4 * 2));
    RC_TRY(w.rec_idx.reserve((size_t)n * 4 * 2));
    RC_TRY(w.rec_k64.reserve((size_t)n * 8 * 2));
    RC_TRY(w.rec_sorted.reserve((size_t)n * sizeof(SdRec)));
    uint32_t *k32 = w.rec_k32.as<uint32_t>(), *idx = w.rec_idx.as<uint32_t>();
    uint64_t *k64 = w.rec_k64.as<uint64_t>();
    const unsigned nb = (unsigned)((n + 255) / 256);
    rec_keys32_kernel<<<nb, 256, 0, s>>>(recs, n, k32, idx);
    rocprim::double_buffer<uint32_t> kd(k32, k32 + n), vd(idx, idx + n);
    size_t bytes = 0;
    HIP_TRY(rocprim::radix_sort_pairs(nullptr, bytes, kd, vd, (size_t)n, 0, 32, s));
    RC_TRY(w.sort_tmp.reserve(bytes));
    HIP_TRY(rocprim::radix_sort_pairs(w.sort_tmp.p, bytes, kd, vd, (size_t)n, 0, 32, s));
    rec_keys64_kernel<<<nb, 256, 0, s>>>(recs, vd.current(), n, k64);
    rocprim::double_buffer<uint64_t> kd2(k64, k64 + n);
    bytes = 0;
    HIP_TRY(rocprim::radix_sort_pairs(nullptr, bytes, kd2, vd, (size_t)n, 0, 64, s));
    RC_TRY(w.sort_tmp.reserve(bytes));
    HIP_TRY(rocprim::radix_sort_pairs(w.sort_tmp.p, bytes, kd2, vd, (size_t)n, 0, 64, s));
    rec_gather_kernel<<<nb, 256, 0, s>>>(recs, vd.current(), n, w.rec_sorted.as<SdRec>());
    HIP_TRY(hipGetLastError());
    return 0;
}

int32_t sa_build_device(const uint8_t *d_text, int64_t n, void *d_sa, bool wide, hipStream_t stream,
                        uint64_t wide_batch) {
    bool dna = false;
    RC_TRY(text_is_dna(d_text, n, stream, &dna));
    if (wide) return build_wide(d_text, n, (uint64_t *)d_sa, dna, stream, wide_batch);
    return build_t<uint32_t>(d_text, n, (uint32_t *)d_sa, dna, stream);
}


// ---- position-sorted occurrence lists (index.hpp: asgart_index::d_sap) -------------------------------------
// sap[lo..hi) = the suffix-array entries of the k-mer interval [lo,hi), sorted by POSITION.  The probe filter
// of the reference (src/automaton.rs:105-114) keeps the occurrences beyond a position threshold, so the kept count
// of a large interval is a bisection in this list instead of a read of the interval -- what the cardinality test
// of a repeat-rich genome otherwise streams (16 GB per GRCh38-shaped pass).

// Only the LARGE intervals matter (rank_count_kernel reads the list of an interval of more than kRankMin entries only): the
// suffix array is copied and every run of more than `min_run` equal keys is sorted by position in place, as one segment of a segmented radix
// sort -- a seventh of the slots of a GRCh38-shaped text instead of all of them, and no pair of n-word buffers.
// rocPRIM's segmented sort counts elements in 32 bits: the array is worked through in windows of 2^30 slots, a run
// belongs to the window it starts in.  Runs of k-mers that start with N are left alone: probes that start with N are
// never searched (src/automaton.rs:100-102), and the all-N run of an assembly's gaps is tens of millions long.
namespace {
template <class SlotT>
__global__ __launch_bounds__(256) void big_runs_kernel(const uint64_t *__restrict__ keys, uint64_t n, uint64_t w0, uint64_t w1,
                                                       uint32_t min_run, int k, uint32_t *__restrict__ begins,
                                                       uint32_t *__restrict__ ends, unsigned long long *__restrict__ ctr) {
    const uint32_t lane = threadIdx.x & 63u;
    for (uint64_t r0 = w0 + (((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) & ~63ull); r0 < w1;
         r0 += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = r0 + lane;
        bool big = false;
        uint64_t end = 0;
        if (r < w1 && r + min_run < n) {
            const uint64_t u = keys[r];
            const uint32_t first = (uint32_t)(u >> (3 * (k - 1))) & 7u;
            big = (r == 0 || keys[r - 1] != u) && keys[r + min_run] == u && first != 4u;
            if (big) {  // first slot beyond the run: gallop, then bisect
                uint64_t lo = r + min_run, step = min_run;
                while (lo + step < n && keys[lo + step] == u) {
                    lo += step;
                    step <<= 1;
                }
                uint64_t hi = lo + step < n ? lo + step : n;  // keys[lo] == u, keys[hi] != u (or hi == n)
                while (hi - lo > 1) {
                    const uint64_t mid = lo + ((hi - lo) >> 1);
                    if (keys[mid] == u) lo = mid; else hi = mid;
                }
                end = hi;
            }
        }
        const unsigned long long m = __ballot(big);
        if (m) {
            const int leader = __ffsll((long long)m) - 1;
            unsigned long long at = 0;
            if ((int)lane == leader) at = atomicAdd(&ctr[0], (unsigned long long)__popcll(m));
            at = __shfl(at, leader);
            if (big) {
                const unsigned long long j = at + (unsigned long long)__popcll(m & ((1ull << lane) - 1ull));
                begins[j] = (uint32_t)(r - w0);
                ends[j] = (uint32_t)(end - w0);
                atomicMax(&ctr[1], (unsigned long long)(end - w0));
            }
        }
    }
}
}  // namespace

template <class SlotT>
int32_t build_rank_lists_runs(const uint64_t *d_keys, const SlotT *d_sa, uint64_t n, SlotT *d_sap, uint32_t min_run, int k,
                              hipStream_t s) {
    if (n == 0) return 0;
    constexpr uint64_t kWindow = 1ull << 30;
    const uint64_t max_runs = kWindow / (min_run + 1u) + 2u;  // per window
    DevBuf b_beg, b_end, b_ctr, temp;
    struct Free {
        DevBuf &a, &b, &c, &t;
        ~Free() { a.release(); b.release(); c.release(); t.release(); }
    } guard{b_beg, b_end, b_ctr, temp};
    RC_TRY(b_beg.reserve((size_t)max_runs * 4));
    RC_TRY(b_end.reserve((size_t)max_runs * 4));
    RC_TRY(b_ctr.reserve(16));
    HIP_TRY(hipMemcpyAsync(d_sap, d_sa, (size_t)n * sizeof(SlotT), hipMemcpyDeviceToDevice, s));
    int pos_bits = 1;
    while (pos_bits < 64 && (n >> pos_bits)) ++pos_bits;
    for (uint64_t w0 = 0; w0 < n; w0 += kWindow) {
        const uint64_t w1 = std::min(n, w0 + kWindow);
        HIP_TRY(hipMemsetAsync(b_ctr.p, 0, 16, s));
        big_runs_kernel<SlotT><<<(unsigned)std::min<uint64_t>((w1 - w0 + 255) / 256, 1u << 16), 256, 0, s>>>(
            d_keys, n, w0, w1, min_run, k, b_beg.as<uint32_t>(), b_end.as<uint32_t>(), b_ctr.as<unsigned long long>());
        HIP_TRY(hipGetLastError());
        unsigned long long h[2] = {0, 0};
        HIP_TRY(read_back(h, b_ctr.p, 16, s));
        HIP_TRY(stream_sync(s));
        if (!h[0]) continue;
        if (h[1] >= 0xFFFFFFFFull) {
            set_error("internal: a run of equal keys longer than 2^32 - 2^30 slots");
            return ASGART_E_CAP;
        }
        size_t bytes = 0;
        HIP_TRY(rocprim::segmented_radix_sort_keys(nullptr, bytes, d_sa + w0, d_sap + w0, (unsigned)h[1], (unsigned)h[0],
                                                   b_beg.as<uint32_t>(), b_end.as<uint32_t>(), 0, (unsigned)pos_bits, s));
        RC_TRY(temp.reserve(bytes));
        HIP_TRY(rocprim::segmented_radix_sort_keys(temp.p, bytes, d_sa + w0, d_sap + w0, (unsigned)h[1], (unsigned)h[0],
                                                   b_beg.as<uint32_t>(), b_end.as<uint32_t>(), 0, (unsigned)pos_bits, s));
    }
    HIP_TRY(stream_sync(s));
    return 0;
}
template int32_t build_rank_lists_runs<uint32_t>(const uint64_t *, const uint32_t *, uint64_t, uint32_t *, uint32_t, int, hipStream_t);
template int32_t build_rank_lists_runs<uint64_t>(const uint64_t *, const uint64_t *, uint64_t, uint64_t *, uint32_t, int, hipStream_t);

}  // namespace asgart

extern "C" int32_t asgart_sa_build64(const uint8_t *T, int64_t *SA, int64_t n) {
    using namespace asgart;
    if (n < 0 || (n > 0 && (!T || !SA))) {
        set_error("asgart_sa_build64: bad argument");
        return ASGART_E_ARG;
    }
    if (n == 0) return 0;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
        (void)hipGetLastError();
        set_error("no HIP device available; libasgart_hip has no CPU fallback");
        return ASGART_E_HIP;
    }
    const bool wide = (uint64_t)n >= 0xFFFFFF00ull;
    DevBuf text, sa, out;
    hipStream_t s = nullptr;
    int32_t rc = [&]() -> int32_t {
        HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        RC_TRY(text.reserve((size_t)n + 64));
        RC_TRY(sa.reserve(((size_t)n + 16) * (wide ? 8 : 4)));
        HIP_TRY(hipMemcpyAsync(text.p, T, (size_t)n, hipMemcpyHostToDevice, s));
        RC_TRY(sa_build_device(text.as<uint8_t>(), n, sa.p, wide, s, 0));
        if (wide) {
            HIP_TRY(read_back(SA, sa.p, (size_t)n * 8, s));
        } else {
            const uint64_t slice = 1ull << 27;
            RC_TRY(out.reserve((size_t)(slice < (uint64_t)n ? slice : (uint64_t)n) * 8));
            for (uint64_t off = 0; off < (uint64_t)n; off += slice) {
                const uint64_t cnt = (uint64_t)n - off < slice ? (uint64_t)n - off : slice;
                widen_kernel<<<grid_for(cnt), 256, 0, s>>>(sa.as<uint32_t>() + off,
                                                           out.as<int64_t>(), cnt);
                HIP_TRY(hipGetLastError());
                HIP_TRY(read_back(SA + off, out.p, cnt * 8, s));
                HIP_TRY(stream_sync(s));
            }
        }
        HIP_TRY(stream_sync(s));
        return 0;
    }();
    text.release();
    sa.release();
    out.release();
    if (s) (void)hipStreamDestroy(s);
    // (a stand-alone suffix sort leaves nothing behind in the block cache unless an index of this device may reuse it)
    if (BlockCache::live_indexes() == 0) BlockCache::trim();
    return rc;
}
