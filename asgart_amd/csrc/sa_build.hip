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
                                                         IdxT *__restrict__ head) {
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    bool is_head = j == 0 || k1[j] != k1[j - 1] || (k2 && k2[j] != k2[j - 1]);
    IdxT slot = slots ? slots[j] : (IdxT)j;
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
                                                          uint8_t *__restrict__ tied) {
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    const IdxT x = vals[j];
    const IdxT slot = slots ? slots[j] : (IdxT)j;
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
                K1.db.current(), two_keys ? K2.db.current() : nullptr, slots, m, head);
            HIP_TRY(hipGetLastError());
            size_t bytes = 0;
            HIP_TRY(rocprim::inclusive_scan(nullptr, bytes, head, grp, (size_t)m, rocprim::maximum<IdxT>(), s));
            RC_TRY(temp.reserve(bytes));
            HIP_TRY(rocprim::inclusive_scan(temp.p, bytes, head, grp, (size_t)m, rocprim::maximum<IdxT>(), s));
            apply_round_kernel<IdxT><<<grid_for(m), 256, 0, s>>>(V.db.current(), grp, slots, m, d_sa, rank, tied);
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
            HIP_TRY(hipMemcpyAsync(&h_count, d_count, sizeof(size_t), hipMemcpyDeviceToHost, s));
            HIP_TRY(hipStreamSynchronize(s));
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

}  // namespace

// Segment placement of the extension step: ascending radix sort of (tier, longest-first) keys.
// keys/vals hold 2*n entries each (second half = alternate buffer).
int32_t sort_segments(Workspace &w, uint32_t *keys, uint32_t *vals, uint64_t n, hipStream_t s,
                      const uint32_t **sorted_vals, const uint32_t **sorted_keys) {
    rocprim::double_buffer<uint32_t> kd(keys, keys + n), vd(vals, vals + n);
    size_t bytes = 0;
    HIP_TRY(rocprim::radix_sort_pairs(nullptr, bytes, kd, vd, (size_t)n, 0, 32, s));
    RC_TRY(w.sort_tmp.reserve(bytes));
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

int32_t sa_build_device(const uint8_t *d_text, int64_t n, void *d_sa, bool wide, hipStream_t stream) {
    bool dna = false;
    RC_TRY(text_is_dna(d_text, n, stream, &dna));
    if (wide) return build_t<uint64_t>(d_text, n, (uint64_t *)d_sa, dna, stream);
    return build_t<uint32_t>(d_text, n, (uint32_t *)d_sa, dna, stream);
}

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
        RC_TRY(sa_build_device(text.as<uint8_t>(), n, sa.p, wide, s));
        if (wide) {
            HIP_TRY(hipMemcpyAsync(SA, sa.p, (size_t)n * 8, hipMemcpyDeviceToHost, s));
        } else {
            const uint64_t slice = 1ull << 27;
            RC_TRY(out.reserve((size_t)(slice < (uint64_t)n ? slice : (uint64_t)n) * 8));
            for (uint64_t off = 0; off < (uint64_t)n; off += slice) {
                const uint64_t cnt = (uint64_t)n - off < slice ? (uint64_t)n - off : slice;
                widen_kernel<<<grid_for(cnt), 256, 0, s>>>(sa.as<uint32_t>() + off,
                                                           out.as<int64_t>(), cnt);
                HIP_TRY(hipGetLastError());
                HIP_TRY(hipMemcpyAsync(SA + off, out.p, cnt * 8, hipMemcpyDeviceToHost, s));
                HIP_TRY(hipStreamSynchronize(s));
            }
        }
        HIP_TRY(hipStreamSynchronize(s));
        return 0;
    }();
    text.release();
    sa.release();
    out.release();
    if (s) (void)hipStreamDestroy(s);
    return rc;
}
