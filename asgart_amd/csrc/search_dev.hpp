// search_dev.hpp -- device-side k-mer -> SA-interval lookup.
//
// Replaces Searcher::search (reference src/searcher.rs:145-180).  The
// reference bisects sa[lo8..hi8) of the probe's 8-mer bucket, gathering
// dna[sa[mid]..+k] at every step (two dependent random reads per step).  Here
// the keys array holds the first k bases of every suffix in SA order as one
// sorted u64 array, so the interval is [lower_bound(q), upper_bound(q)) over a
// single array, entered through a 4^d prefix table: ~log2(n/4^d) dependent
// 8-byte reads instead of ~2*log2(n/5^8) (sa, text) pairs.
#pragma once

#include "index.hpp"

namespace asgart {

// Byte accounting of the lookup path: NoBytes compiles to nothing (the product kernels);
// CountBytes sums the bytes each load / store of THIS design moves (the counting pass behind
// `kernel_algorithmic_bytes` of bench.py's roofline object).
struct NoBytes {
    __device__ inline void rd(uint32_t) {}
    __device__ inline void wr(uint32_t) {}
    __device__ inline void rd16() {}
};
struct CountBytes {
    unsigned long long n = 0, n16 = 0;
    __device__ inline void rd(uint32_t b) { n += b; }
    __device__ inline void wr(uint32_t b) { n += b; }
    // ... the wide coalesced loads (16 bytes per lane, a whole wave at once): counted apart, because FETCH_SIZE
    // tallies exactly half their bytes on gfx950 (MI355X_MICROARCH.md, HBM)
    __device__ inline void rd16() { n += 16; n16 += 16; }
};

// index into the ACGT-only d-mer table from a k-mer key; false if one of the
// first d bases is not A/C/G/T.
__device__ inline bool prefix_index(uint64_t q, int k, int d, uint32_t &p) {
    uint32_t idx = 0;
    bool ok = true;
    for (int j = 0; j < d; ++j) {
        uint32_t code = (uint32_t)(q >> (3 * (k - 1 - j))) & 7u;
        uint32_t dg = acgt_digit(code);
        ok &= dg < 4u;
        idx = (idx << 2) | (dg & 3u);
    }
    p = idx;
    return ok;
}

template <class Cnt = NoBytes>
__device__ inline uint64_t lower_bound_keys(const uint64_t *__restrict__ keys, uint64_t lo,
                                            uint64_t hi, uint64_t q, Cnt &&cb = Cnt()) {
    while (lo < hi) {
        uint64_t mid = lo + ((hi - lo) >> 1);
        cb.rd(8);
        if (keys[mid] < q) lo = mid + 1; else hi = mid;
    }
    return lo;
}

template <class Cnt = NoBytes>
__device__ inline uint64_t upper_bound_keys(const uint64_t *__restrict__ keys, uint64_t lo,
                                            uint64_t hi, uint64_t q, Cnt &&cb = Cnt()) {
    while (lo < hi) {
        uint64_t mid = lo + ((hi - lo) >> 1);
        cb.rd(8);
        if (keys[mid] <= q) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// Where the bases of a probe can be read again: its first base and the direction it runs in (a reversed needle runs
// down the text), complemented or not.  Only probes longer than two words (k > 42) are read through it.
struct ProbeRef {
    const uint8_t *p = nullptr;
    int dir = 1;
    bool comp = false;
    __device__ inline uint32_t code(int j) const {
        const uint32_t c = base_code(p[(long long)j * dir]);
        return comp ? comp_code(c) : c;
    }
};

// Probe sizes above kMaxKey: a key word holds the first kk = 21 bases of a suffix, the next (up to) 21 are packed on
// demand from the text into a second word (bases past the end of the text count as 0, like the padding of the key
// words), what lies beyond is compared base by base.  Slots with equal key words are in suffix order, so their tails
// are non-decreasing.
template <class SlotT, class Cnt = NoBytes>
__device__ inline uint64_t tail_key(const IndexView<SlotT> &ix, uint64_t x, Cnt &&cb = Cnt()) {
    uint64_t t = 0;
    const int n2 = ix.k2 < kMaxKey ? ix.k2 : kMaxKey;
    for (int j = 0; j < n2; ++j) {
        const uint64_t p = x + (uint64_t)ix.kk + (uint64_t)j;
        t = (t << 3) | (p < ix.n ? base_code(ix.text[p]) : 0u);
    }
    cb.rd((uint32_t)n2);
    return t;
}

// tail of the suffix at x against the probe's tail (second word q2, the rest through pr): -1 / 0 / 1
template <class SlotT, class Cnt = NoBytes>
__device__ inline int tail_cmp(const IndexView<SlotT> &ix, uint64_t x, uint64_t q2, const ProbeRef &pr, Cnt &&cb = Cnt()) {
    const uint64_t t = tail_key(ix, x, cb);
    if (t != q2) return t < q2 ? -1 : 1;
    for (int j = kMaxKey; j < ix.k2; ++j) {  // (k > 42)
        const uint64_t p = x + (uint64_t)ix.kk + (uint64_t)j;
        const uint32_t a = p < ix.n ? base_code(ix.text[p]) : 0u;
        const uint32_t b = pr.code(ix.kk + j);
        cb.rd(2);
        if (a != b) return a < b ? -1 : 1;
    }
    return 0;
}

// [lo,hi): slots whose key word equals the probe's -> the slots whose tail equals the probe's as well
template <class SlotT, class Cnt = NoBytes>
__device__ inline void refine_tail(const IndexView<SlotT> &ix, uint64_t q2, const ProbeRef &pr, uint64_t &lo, uint64_t &hi,
                                   Cnt &&cb = Cnt()) {
    if (!ix.k2 || lo >= hi) return;
    uint64_t a = lo, b = hi;
    while (a < b) {  // first slot with tail >= the probe's
        const uint64_t mid = a + ((b - a) >> 1);
        cb.rd(sizeof(SlotT));
        if (tail_cmp(ix, (uint64_t)ix.sa[mid], q2, pr, cb) < 0) a = mid + 1; else b = mid;
    }
    const uint64_t l = a;
    b = hi;
    while (a < b) {  // first slot with tail > the probe's
        const uint64_t mid = a + ((b - a) >> 1);
        cb.rd(sizeof(SlotT));
        if (tail_cmp(ix, (uint64_t)ix.sa[mid], q2, pr, cb) <= 0) a = mid + 1; else b = mid;
    }
    lo = l;
    hi = a;
}

// dense index of an 8-mer over the 5-letter alphabet in code order (A,C,G,N,T)
__device__ inline bool cache8_index(uint32_t pre24, uint32_t &idx) {
    uint32_t v = 0;
    bool ok = true;
    for (int j = 0; j < kCacheLen; ++j) {
        uint32_t code = (pre24 >> (3 * (kCacheLen - 1 - j))) & 7u;
        ok &= (code >= 1u && code <= 5u);
        v = v * 5u + (code - 1u);
    }
    idx = v;
    return ok;
}

template <class SlotT>
__device__ inline bool in_tail_list(const IndexView<SlotT> &ix, uint32_t pre24) {
    if (!((ix.tail_bloom >> (pre24 & 63u)) & 1ull)) return false;
    for (int j = 0; j < ix.n_tail8; ++j)
        if (ix.tail8[j] == pre24) return true;
    return false;
}

// Exact emulation of the reference's bisection for the text-tail corner
// (reference src/searcher.rs:164-170 + superslice equal_range_by): comparator
// says Less for suffixes shorter than k although they may sort Greater.
template <class SlotT, class Cnt = NoBytes>
__device__ inline void kmer_range_tail(const IndexView<SlotT> &ix, uint64_t q, uint64_t q2, const ProbeRef &pr,
                                       uint64_t &lo, uint64_t &hi, Cnt &&cb = Cnt()) {
    uint32_t pre24 = (uint32_t)(q >> (3 * (ix.kk - kCacheLen)));
    uint32_t c8;
    if (!cache8_index(pre24, c8)) {  // cannot happen for validated text
        lo = hi = 0;
        return;
    }
    const uint64_t L = ix.c8lo[c8], R = ix.c8hi[c8];
    cb.rd(2 * sizeof(SlotT));
    uint64_t size = R - L;
    if (size == 0) {
        lo = hi = L;
        return;
    }
    auto cmp = [&](uint64_t r) -> int {
        uint64_t x = ix.sa[r];
        cb.rd(sizeof(SlotT) + 8);
        if (x + (uint64_t)ix.k > ix.n) return -1;
        uint64_t kv = ix.keys[r];
        if (kv != q) return kv < q ? -1 : 1;
        if (!ix.k2) return 0;
        return tail_cmp(ix, x, q2, pr, cb);
    };
    uint64_t b0 = 0, b1 = 0;
    while (size > 1) {
        uint64_t half = size >> 1;
        uint64_t m0 = b0 + half, m1 = b1 + half;
        int c0 = cmp(L + m0);
        int c1 = (m1 == m0) ? c0 : cmp(L + m1);
        if (c0 < 0) b0 = m0;
        if (c1 <= 0) b1 = m1;
        size -= half;
    }
    int c0 = cmp(L + b0);
    int c1 = (b1 == b0) ? c0 : cmp(L + b1);
    uint64_t rs = b0 + (c0 < 0 ? 1 : 0), re = b1 + (c1 <= 0 ? 1 : 0);
    if (re < rs) re = rs;
    lo = L + rs;
    hi = L + re;
}

// SA slot interval [lo,hi) of the k-mer with key q.  Returns false when the interval comes from the
// emulated bisection of the text-tail corner (it is then what the reference finds, not necessarily
// the set of all occurrences).
template <class SlotT>
__device__ inline bool is_tail_corner(const IndexView<SlotT> &ix, uint64_t q) {
    return ix.n_tail8 && in_tail_list(ix, (uint32_t)(q >> (3 * (ix.kk - kCacheLen))));
}

// --trim index (reference src/bin/asgart.rs:142-148): exactly the reference's two steps -- the 8-mer
// cache entry (Searcher::new's bisection, replayed when the index was prepared), then the equal range
// inside that bucket: plain bounds over the keys when the bucket is clean, the step-by-step replay of
// equal_range_by when it holds one of the out-of-place suffixes.
template <class SlotT, class Cnt = NoBytes>
__device__ inline void kmer_range_trim(const IndexView<SlotT> &ix, uint64_t q, uint64_t q2, const ProbeRef &pr,
                                       uint64_t &lo, uint64_t &hi, Cnt &&cb = Cnt()) {
    uint32_t c8;
    if (!cache8_index((uint32_t)(q >> (3 * (ix.kk - kCacheLen))), c8)) {
        lo = hi = 0;
        return;
    }
    const uint64_t L = ix.c8lo[c8], R = ix.c8hi[c8];
    cb.rd(2 * sizeof(SlotT));
    bool dirty = false;
    for (int j = 0; j < ix.n_bad; ++j) dirty |= ix.bad[j] >= L && ix.bad[j] < R;
    if (dirty) {
        kmer_range_tail(ix, q, q2, pr, lo, hi, cb);
        return;
    }
    lo = lower_bound_keys(ix.keys, L, R, q, cb);
    hi = upper_bound_keys(ix.keys, lo, R, q, cb);
    refine_tail(ix, q2, pr, lo, hi, cb);
}

template <class SlotT, class Cnt = NoBytes>
__device__ inline bool kmer_range(const IndexView<SlotT> &ix, uint64_t q, uint64_t q2, const ProbeRef &pr, uint64_t &lo,
                                  uint64_t &hi, Cnt &&cb = Cnt()) {
    if (ix.trim) {
        kmer_range_trim(ix, q, q2, pr, lo, hi, cb);
        return false;  // the interval need not hold the probe's own position
    }
    if (is_tail_corner(ix, q)) {
        kmer_range_tail(ix, q, q2, pr, lo, hi, cb);
        return false;
    }
    uint64_t lo0 = 0, hi0 = ix.n_sa;
    uint32_t p;
    if (prefix_index(q, ix.kk, ix.d, p)) {
        lo0 = ix.ptab[p];
        hi0 = ix.ptab[p + 1];
        cb.rd(2 * sizeof(SlotT));
    }
    uint64_t l = lower_bound_keys(ix.keys, lo0, hi0, q, cb);
    // most k-mers are unique or absent: walk a few equal keys (same cache
    // line) before falling back to a second bisection
    uint64_t h = l;
    int walk = 0;
    while (h < hi0 && walk < 8) {
        cb.rd(8);
        if (ix.keys[h] != q) break;
        ++h;
        ++walk;
    }
    if (walk == 8 && h < hi0) {
        cb.rd(8);
        if (ix.keys[h] == q) h = upper_bound_keys(ix.keys, h, hi0, q, cb);
    }
    lo = l;
    hi = h;
    refine_tail(ix, q2, pr, lo, hi, cb);
    return true;
}

// ---- k-mer presence filter ---------------------------------------------------------------
// A blocked two-bit Bloom filter over k-mer keys, small enough (2^filt_bits bits, 128 MiB by
// default) to stay resident in the 256 MiB Infinity Cache: both bits of a key live in one 64-bit
// word, so a test is ONE load that normally never reaches HBM.  One filter per orientation of
// the run (index.hip builds it): it holds every k-mer q of the text whose probe could have a hit,
//   direct pass:      q occurs at least twice in the text (one occurrence is the probe itself),
//   -R / -C / -RC:    q occurs in the text AND T(q) occurs in the text, T = the needle
//                     transformation (an involution) -- the probe IS T(some text k-mer).
// No false negatives: a probe the filter rejects provably has no hit (its SA interval holds just
// the probe itself in the direct pass, nothing otherwise), so it skips the prefix table, the
// key bisection and the suffix-array read -- the random HBM gathers of the lookup.
__device__ inline void filter_slot(uint64_t q, int bits, uint64_t &word, uint64_t &mask) {
    const uint64_t h = q * 0x9E3779B97F4A7C15ull;
    word = h >> (64 - (bits - 6));
    const uint64_t h2 = (q ^ (q >> 29)) * 0xD6E8FEB86659FD93ull;
    mask = (1ull << (h2 >> 58)) | (1ull << ((h2 >> 52) & 63u));
}
__device__ inline bool filter_test(const uint64_t *__restrict__ flt, int bits, uint64_t q) {
    uint64_t w, m;
    filter_slot(q, bits, w, m);
    return (flt[w] & m) == m;
}

// the needle transformation on a packed key: complement every code, then reverse the order
__device__ inline uint64_t transform_key(uint64_t q, int k, bool reverse, bool complement) {
    uint64_t out = 0;
    for (int j = 0; j < k; ++j) {
        uint32_t c = (uint32_t)(q >> (3 * (k - 1 - j))) & 7u;  // j-th base
        if (complement) c = comp_code(c);
        const int dst = reverse ? k - 1 - j : j;               // its place in the result
        out |= (uint64_t)c << (3 * (k - 1 - dst));
    }
    return out;
}

// key of the probe at needle-local offset i of chunk (s, L) under the run's
// orientation: needle = chunk | complemented | reversed (reference
// src/bin/asgart.rs:206-218), probe = needle[i..i+k].  *first = first base code.
// Probes longer than kMaxKey: the key word is the first kMaxKey bases, *q2 the next (up to) kMaxKey, packed.
__device__ inline uint64_t probe_key(const uint8_t *__restrict__ text, uint64_t s, uint64_t L,
                                     uint64_t i, int k, bool reverse, bool complement,
                                     uint32_t *first, uint64_t *q2 = nullptr) {
    uint64_t q = 0, t = 0;
    uint32_t f = 0;
    const uint8_t *p = reverse ? text + s + L - 1 - i : text + s + i;
    for (int j = 0; j < k; ++j) {
        uint32_t c = base_code(reverse ? *(p - j) : p[j]);
        if (complement) c = comp_code(c);
        if (j == 0) f = c;
        if (j < kMaxKey) q = (q << 3) | c;
        else if (j < 2 * kMaxKey) t = (t << 3) | c;  // (the bases beyond are read again when two words tie: ProbeRef)
    }
    *first = f;
    if (q2) *q2 = t;
    return q;
}

}  // namespace asgart
