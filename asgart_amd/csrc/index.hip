// index.hip -- index creation (upload + search structures) and the
// Searcher-level entry points of the C ABI.
//
// Reference counterparts: r_divsufsort + Searcher::new, src/bin/asgart.rs:141-155,
// src/searcher.rs:99-143; Searcher::search, src/searcher.rs:145-180.
#include "index.hpp"
#include "search_dev.hpp"

#include <algorithm>
#include <cctype>
#include <cerrno>
#include <chrono>
#include <cstdlib>

#include <dirent.h>
#include <execinfo.h>
#include <signal.h>
#include <sys/syscall.h>
#include <unistd.h>

namespace asgart {

static thread_local char g_err[512] = "";
thread_local bool tl_owns_pass_mu = false;

// Runtime note.  The extension tiers of one call run on six HIP streams (twelve with
// two calls in flight); ROCm maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and kernels of
// streams that share a queue run one after the other.  The HOST exports GPU_MAX_HW_QUEUES=8 before its first HIP
// call (the Python binding and bench.py do); the library never touches the process environment.
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- options ---------------------------------------------------------------
namespace {
struct OptDesc {
    const char *name;
    int64_t Options::*field;
    int64_t lo, hi;
};
const OptDesc kOptions[] = {
    {"shard_lookback", &Options::shard_lookback, 1, 1ll << 31},
    {"shard_lookahead", &Options::shard_lookahead, 0, 1ll << 31},
    {"force_tier", &Options::force_tier, 0, 7},
    {"arms_kernel", &Options::arms_kernel, 0, 1},
    {"long3", &Options::long3, 0, 1ll << 31},
    {"cap1", &Options::cap1, 1, 256},
    {"debug", &Options::debug, 0, 2},
    {"test_cap_limit", &Options::test_cap_limit, -1, 1ll << 31},
    {"test_genbits", &Options::test_genbits, 2, 22},
    {"test_k8_delay", &Options::test_k8_delay, 0, 1 << 22},
    {"tier_order", &Options::tier_order, 1, 7777777},
    {"ptab_depth", &Options::ptab_depth, 0, 15},
    {"force_wide", &Options::force_wide, 0, 1},
    {"test_wide_batch", &Options::test_wide_batch, 0, 1ll << 40},
    {"kfilter_bits", &Options::kfilter_bits, 0, 34},
    {"lazy_aux", &Options::lazy_aux, 0, 1},
    {"fuse_passes", &Options::fuse_passes, 0, 2},
    {"fuse_pole_pct", &Options::fuse_pole_pct, 1, 1000},
    {"barren", &Options::barren, 0, 2},
    {"dense3", &Options::dense3, 0, 1 << 20},
    {"dense6", &Options::dense6, 0, 1 << 20},
    {"prewarm", &Options::prewarm, 0, 1},
    {"cache_calls", &Options::cache_calls, 0, 1000000},
    {"split", &Options::split, 0, 2},
    {"split_len", &Options::split_len, 0, 1 << 20},
    {"split_runs", &Options::split_runs, 1, 3072},
    {"split_warm", &Options::split_warm, 0, 1 << 20},
    {"split_warm_max", &Options::split_warm_max, 0, 1 << 22},
    {"split_min", &Options::split_min, 0, 1ll << 31},
    {"watchdog_s", &Options::watchdog_s, 0, 86400},
    {"test_stall_s", &Options::test_stall_s, 0, 60},
    {"cap6_pct", &Options::cap6_pct, 100, 200},
    {"rank_lists", &Options::rank_lists, 0, 1},
    {"cap3_pct", &Options::cap3_pct, 100, 400},
    {"cap45_pct", &Options::cap45_pct, 100, 800},
    {"test_fail_alloc", &Options::test_fail_alloc, -1, 1000000000},
    {"cap6w_pct", &Options::cap6w_pct, 100, 400},
    {"solo", &Options::solo, 0, 48},
    {"posbits", &Options::posbits, 0, 2},
};
}  // namespace

// The main stream of a call context carries the chip-wide, short kernels of a call (probe search, scans, CSR
// fill, placement, record sort) and its small copies; the extension tiers, whose persistent workgroups hold
// their CU slots for tens of milliseconds, run on the other six.  The main stream gets the highest priority the
// device offers: when two calls are in flight, the short kernels of one are dispatched into the first slots
// the other's tiers give back instead of queueing behind the tiers' own backlog of workgroups.
int32_t create_ctx_streams(SearchCtx &cx) {
    int least = 0, greatest = 0;
    HIP_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
    HIP_TRY(hipStreamCreateWithPriority(&cx.stream, hipStreamNonBlocking, greatest));
    for (hipStream_t *st : {&cx.stream2, &cx.stream3, &cx.stream4, &cx.stream5, &cx.stream6, &cx.stream7})
        HIP_TRY(hipStreamCreateWithPriority(st, hipStreamNonBlocking, least));
    for (auto &e : cx.ev) HIP_TRY(hipEventCreate(&e));
    // The CSR fill runs on stream2 beside the placement walk -- unless the process has been told to make do with few
    // hardware queues (GPU_MAX_HW_QUEUES below 6: several processes sharing one device, bench.py's one-device mode): streams
    // that share a queue run one after the other, and a tier stream used in the front put the tiers of two ranks on one
    // device behind one another (283 instead of 74 ms for a rank's shard).
    const char *q = getenv("GPU_MAX_HW_QUEUES");
    cx.fill_stream = (q && *q && atoi(q) < 6) ? cx.stream : cx.stream2;
    return 0;
}

int32_t option_set(Options &o, const char *name, int64_t value) {
    if (!name) {
        set_error("option name is NULL");
        return ASGART_E_ARG;
    }
    if (!strncmp(name, "grid", 4) && name[4] >= '1' && name[4] <= '7' && !name[5]) {
        // a tier's grid: every workgroup kernel reserves per-workgroup resources for at most
        // its default grid, so larger requests are clamped at the launch site
        if (value < 0 || value > (1ll << 20)) {
            set_error("option %s: value %lld out of range", name, (long long)value);
            return ASGART_E_ARG;
        }
        o.grid[name[4] - '0'] = value;
        return 0;
    }
    for (const OptDesc &d : kOptions)
        if (!strcmp(name, d.name)) {
            if (value < d.lo || value > d.hi) {
                set_error("option %s: value %lld outside [%lld, %lld]", name, (long long)value,
                          (long long)d.lo, (long long)d.hi);
                return ASGART_E_ARG;
            }
            if (d.field == &Options::tier_order)
                for (int64_t v = value; v; v /= 10)
                    if (v % 10 < 1 || v % 10 > 7) {
                        set_error("option tier_order: digits must be tiers 1..7");
                        return ASGART_E_ARG;
                    }
            o.*(d.field) = value;
            return 0;
        }
    set_error("unknown option '%s'", name);
    return ASGART_E_ARG;
}

// ASGART_<NAME> for every option; malformed or out-of-range values are ignored (defaults stay)
void options_from_env(Options &o) {
    auto one = [&](const char *name) {
        char env[64] = "ASGART_";
        size_t j = 7;
        for (const char *c = name; *c && j + 1 < sizeof(env); ++c) env[j++] = (char)toupper((unsigned char)*c);
        env[j] = 0;
        const char *e = getenv(env);
        if (!e || !*e) return;
        char *end = nullptr;
        const long long v = strtoll(e, &end, 10);
        if (end && *end == 0) (void)option_set(o, name, (int64_t)v);
    };
    for (const OptDesc &d : kOptions) one(d.name);
    for (int t = 1; t <= 7; ++t) {
        char nm[8];
        snprintf(nm, sizeof nm, "grid%d", t);
        one(nm);
    }
}

// ---- kernels ---------------------------------------------------------------

__global__ __launch_bounds__(256) void byte_histogram_kernel(const uint8_t *__restrict__ text,
                                                             uint64_t n,
                                                             unsigned long long *__restrict__ hist) {
    __shared__ unsigned int sh[256];
    sh[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x * 16;
    for (uint64_t base = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16; base < n;
         base += stride) {
        if (base + 16 <= n) {
            uint4 v = *reinterpret_cast<const uint4 *>(text + base);
            uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) atomicAdd(&sh[(w[a] >> (8 * b)) & 0xFF], 1u);
        } else {
            for (uint64_t j = base; j < n; ++j) atomicAdd(&sh[text[j]], 1u);
        }
    }
    __syncthreads();
    if (sh[threadIdx.x]) atomicAdd(&hist[threadIdx.x], (unsigned long long)sh[threadIdx.x]);
}

__global__ __launch_bounds__(256) void narrow_sa_kernel(const int64_t *__restrict__ in,
                                                        uint32_t *__restrict__ out, uint64_t cnt) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < cnt) out[i] = (uint32_t)in[i];
}

__global__ __launch_bounds__(256) void widen_sa_kernel(const uint32_t *__restrict__ in,
                                                       int64_t *__restrict__ out, uint64_t cnt) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < cnt) out[i] = (int64_t)in[i];
}

// keys[r] = first k bases of suffix sa[r], 3-bit codes, end-of-text padded with 0
template <class SlotT>
__global__ __launch_bounds__(256) void build_keys_kernel(const uint8_t *__restrict__ text,
                                                         const SlotT *__restrict__ sa,
                                                         uint64_t *__restrict__ keys, uint64_t n,
                                                         uint64_t n_sa, int k) {
    // grid-stride: a launch holds fewer than 2^32 threads, a 6-Gbp index more than 2^32 slots
    // The k <= 21 bytes of a suffix come as four aligned 8-byte words (a 32-byte window holds them at any alignment;
    // the text allocation has 64 spare bytes behind the text) instead of k byte loads -- a fifth of the requests.
    const uint64_t *__restrict__ words = reinterpret_cast<const uint64_t *>(text);
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_sa; r += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t x = sa[r];
        const uint64_t w0 = words[(x >> 3)], w1 = words[(x >> 3) + 1], w2 = words[(x >> 3) + 2], w3 = words[(x >> 3) + 3];
        const uint32_t sh = (uint32_t)(x & 7ull) * 8u;
        uint64_t v[3];
        v[0] = sh ? (w0 >> sh) | (w1 << (64u - sh)) : w0;
        v[1] = sh ? (w1 >> sh) | (w2 << (64u - sh)) : w1;
        v[2] = sh ? (w2 >> sh) | (w3 << (64u - sh)) : w2;
        uint64_t q = 0;
#pragma unroll
        for (int j = 0; j < kMaxKey; ++j) {  // (unrolled: the word and the shift of base j are constants; k <= kMaxKey)
            if (j < k) {
                const uint32_t byte = (uint32_t)(v[j >> 3] >> (8 * (j & 7))) & 0xFFu;
                const uint32_t c = (x + (uint64_t)j < n) ? base_code((uint8_t)byte) : 0u;
                q = (q << 3) | c;
            }
        }
        keys[r] = q;
    }
}

template <class SlotT>
__global__ __launch_bounds__(256) void build_ptab_kernel(const uint64_t *__restrict__ keys,
                                                         SlotT *__restrict__ ptab, uint64_t n,
                                                         int k, int d) {
    uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t entries = 1ull << (2 * d);
    if (p > entries) return;
    if (p == entries) {
        ptab[p] = (SlotT)n;
        return;
    }
    uint64_t target = 0;
    for (int j = 0; j < d; ++j) {
        uint32_t dg = (uint32_t)(p >> (2 * (d - 1 - j))) & 3u;
        uint32_t code = dg == 3u ? 5u : dg + 1u;
        target |= (uint64_t)code << (3 * (k - 1 - j));
    }
    ptab[p] = (SlotT)lower_bound_keys(keys, 0, n, target);
}

// the reference's 8-mer cache: [lo,hi) of suffixes starting with the 8-mer
template <class SlotT>
__global__ __launch_bounds__(256) void build_cache8_kernel(const uint64_t *__restrict__ keys,
                                                           SlotT *__restrict__ c8lo,
                                                           SlotT *__restrict__ c8hi, uint64_t n,
                                                           int k) {
    uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (uint32_t)kCacheEntries) return;
    uint32_t v = idx, pre24 = 0;
    for (int j = kCacheLen - 1; j >= 0; --j) {
        uint32_t code = v % 5u + 1u;
        v /= 5u;
        pre24 |= code << (3 * (kCacheLen - 1 - j));
    }
    const int sh = 3 * (k - kCacheLen);
    uint64_t lo = lower_bound_keys(keys, 0, n, (uint64_t)pre24 << sh);
    uint64_t hi = lower_bound_keys(keys, lo, n, ((uint64_t)pre24 + 1ull) << sh);
    c8lo[idx] = (SlotT)lo;
    c8hi[idx] = (SlotT)hi;
}

// --trim: Searcher::new's cache entries are whatever `sa_searchb64` finds (reference
// src/searcher.rs:118-128) on an array that is NOT sorted under its comparator for the few suffixes
// ending within 8 bases of the window, so the bisection itself is replayed: libdivsufsort's published
// `sa_search` + `_compare` (lib/utils.c; the fork's bounded variant is taken to be the same routine on
// [init_left, init_right) -- its source is absent), one thread per 8-mer.
__device__ inline int published_compare(const uint8_t *T, uint64_t Tsize, const uint8_t *P, int Psize,
                                        uint64_t suf, int *match) {
    uint64_t i = suf + (uint64_t)*match;
    int j = *match, r = 0;
    for (; i < Tsize && j < Psize && (r = (int)T[i] - (int)P[j]) == 0; ++i, ++j) {
    }
    *match = j;
    return r == 0 ? -(j != Psize) : r;
}

template <class SlotT>
__global__ __launch_bounds__(256) void build_cache8_trim_kernel(const uint8_t *__restrict__ T, uint64_t Tsize,
                                                                const SlotT *__restrict__ SA, uint64_t n_sa,
                                                                SlotT *__restrict__ c8lo, SlotT *__restrict__ c8hi) {
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (uint32_t)kCacheEntries) return;
    uint8_t P[kCacheLen];
    {   // dense index -> 8 bytes, code order A < C < G < N < T (cache8_index)
        const uint8_t alpha[5] = {'A', 'C', 'G', 'N', 'T'};
        uint32_t v = idx;
        for (int j = kCacheLen - 1; j >= 0; --j) {
            P[j] = alpha[v % 5u];
            v /= 5u;
        }
    }
    const int Psize = kCacheLen;
    long long i = 0, j = 0, k = 0, size = (long long)n_sa, half = size >> 1, lsize, rsize;
    int match, lmatch = 0, rmatch = 0, llmatch, lrmatch, rlmatch, rrmatch, r;
    for (; 0 < size; size = half, half >>= 1) {
        match = lmatch < rmatch ? lmatch : rmatch;
        r = published_compare(T, Tsize, P, Psize, (uint64_t)SA[i + half], &match);
        if (r < 0) {
            i += half + 1;
            half -= (size & 1) ^ 1;
            lmatch = match;
        } else if (r > 0) {
            rmatch = match;
        } else {
            lsize = half, j = i, rsize = size - half - 1, k = i + half + 1;
            for (llmatch = lmatch, lrmatch = match, half = lsize >> 1; 0 < lsize; lsize = half, half >>= 1) {
                lmatch = llmatch < lrmatch ? llmatch : lrmatch;
                r = published_compare(T, Tsize, P, Psize, (uint64_t)SA[j + half], &lmatch);
                if (r < 0) {
                    j += half + 1;
                    half -= (lsize & 1) ^ 1;
                    llmatch = lmatch;
                } else {
                    lrmatch = lmatch;
                }
            }
            for (rlmatch = match, rrmatch = rmatch, half = rsize >> 1; 0 < rsize; rsize = half, half >>= 1) {
                rmatch = rlmatch < rrmatch ? rlmatch : rrmatch;
                r = published_compare(T, Tsize, P, Psize, (uint64_t)SA[k + half], &rmatch);
                if (r <= 0) {
                    k += half + 1;
                    half -= (rsize & 1) ^ 1;
                    rlmatch = rmatch;
                } else {
                    rrmatch = rmatch;
                }
            }
            break;
        }
    }
    const long long left = (0 < (k - j)) ? j : i, count = k - j;
    c8lo[idx] = (SlotT)left;
    c8hi[idx] = (SlotT)(left + count);
}

// --trim: slots of the sub-strand suffixes shorter than k (start position > end - k)
template <class SlotT>
__global__ __launch_bounds__(256) void bad_slots_kernel(const SlotT *__restrict__ sa, uint64_t n_sa,
                                                        uint64_t first_bad_pos, unsigned long long *__restrict__ out,
                                                        unsigned long long *__restrict__ count, int cap) {
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_sa; r += (uint64_t)gridDim.x * blockDim.x)
        if ((uint64_t)sa[r] >= first_bad_pos) {
            const unsigned long long at = atomicAdd(count, 1ull);
            if (at < (unsigned long long)cap) out[at] = r;
        }
}

template <class SlotT>
__global__ __launch_bounds__(256) void add_offset_kernel(SlotT *__restrict__ sa, uint64_t cnt, uint64_t add) {
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < cnt; r += (uint64_t)gridDim.x * blockDim.x)
        sa[r] = (SlotT)((uint64_t)sa[r] + add);
}

// presence filter of one orientation (search_dev.hpp): one thread per suffix-array slot, the first
// slot of every run of equal keys decides for its k-mer
template <class SlotT>
__global__ __launch_bounds__(256) void build_filter_kernel(IndexView<SlotT> ix, bool reverse, bool complement,
                                                           unsigned long long *__restrict__ flt, int bits) {
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < ix.n; r += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t u = ix.keys[r];
    if (r > 0 && ix.keys[r - 1] == u) continue;
    bool full = true;  // all k bases inside the text (no '$' / end padding): only those can equal a probe
    for (int j = 0; j < ix.k; ++j) full &= ((u >> (3 * j)) & 7u) != 0u;
    if (!full) continue;
    bool keep;
    if (!reverse && !complement) {
        keep = r + 1 < ix.n && ix.keys[r + 1] == u;  // occurs at least twice
    } else {
        const uint64_t t = transform_key(u, ix.k, reverse, complement);
        uint64_t lo0 = 0, hi0 = ix.n;
        uint32_t p;
        if (prefix_index(t, ix.k, ix.d, p)) {
            lo0 = ix.ptab[p];
            hi0 = ix.ptab[p + 1];
        }
        const uint64_t l = lower_bound_keys(ix.keys, lo0, hi0, t);
        keep = l < hi0 && ix.keys[l] == t;
    }
    if (keep) {
        uint64_t w, m;
        filter_slot(u, bits, w, m);
        atomicOr(&flt[w], (unsigned long long)m);
    }
    }
}

// The filter's answers by text position: bit p = "the probe that covers text[p .. p + k) in this orientation
// passes the filter" (or is one of the text-tail corner probes, which never take the filter).  One thread per
// 64 positions: both rolling keys (forward, and reversed for -R), one filter word per position.
// refine (option posbits = 2): a position that passes the k-mer filter is looked up once, here, and keeps its bit only
// if a hit of its probe can be KEPT.  The hit filter of src/automaton.rs:105-114 in text coordinates: a probe that covers
// text[p .. p + k) keeps the occurrences x > p of its k-mer (needle not reversed: x > i + needle_offset, and i +
// needle_offset = p) or x >= p + k (reversed needle: x >= needle_offset + L - i = p + k; the extra `m.start != i`
// only removes hits) -- a property of the text and the position alone, whatever chunk list a call brings.  In the
// direct pass the LEFT one of every pair of occurrences keeps a hit and the right one does not: half the lookups
// the k-mer filter lets through end with nothing kept.  Decided exactly for intervals of up to kSmallInterval
// occurrences (scanned) and, when the index has the position-sorted lists, for those of more than kRankMin (the
// list's last entry is the interval's largest position); the others keep the k-mer filter's answer.
template <class SlotT>
__global__ __launch_bounds__(256) void build_posbits_kernel(IndexView<SlotT> ix, bool reverse, bool complement,
                                                            const uint64_t *__restrict__ flt, int bits, bool refine,
                                                            unsigned long long *__restrict__ out, uint64_t n_words) {
    const int k = ix.k;
    const uint64_t mask = k >= 21 ? ~0ull >> 1 : (1ull << (3 * k)) - 1ull;
    for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < n_words; w += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t p0 = w * 64ull;
        auto code_at = [&](uint64_t p) -> uint64_t {
            uint32_t c = p < ix.n ? base_code(ix.text[p]) : 0u;
            if (complement && c) c = comp_code(c);
            return (uint64_t)c;
        };
        uint64_t fk = 0, rk = 0;  // keys of the window that ENDS just before the next base
        for (int j = 0; j < k - 1; ++j) {
            const uint64_t c = code_at(p0 + (uint64_t)j);
            fk = (fk << 3) | c;
            rk = (rk >> 3) | (c << (3 * (k - 1)));
        }
        unsigned long long word = 0;
        for (int b = 0; b < 64; ++b) {
            const uint64_t p = p0 + (uint64_t)b;
            const uint64_t c = code_at(p + (uint64_t)(k - 1));
            fk = ((fk << 3) | c) & mask;
            rk = (rk >> 3) | (c << (3 * (k - 1)));
            const uint64_t q = reverse ? rk : fk;
            bool pass = true;
            if (p + (uint64_t)k <= ix.n && !is_tail_corner(ix, q)) {
                pass = filter_test(flt, bits, q);
                if (pass && refine) {
                    uint64_t lo, hi;
                    ProbeRef pr;  // (read only by probes of more than 42 bases: the filter is for one-word probes)
                    if (kmer_range(ix, q, 0ull, pr, lo, hi)) {  // (every occurrence of the k-mer)
                        const uint64_t thr = reverse ? p + (uint64_t)k : p + 1u;  // a hit x is kept iff x >= thr
                        if (hi - lo <= (uint64_t)kSmallInterval) {
                            bool any = false;
                            for (uint64_t r = lo; r < hi; ++r) any |= (uint64_t)ix.sa[r] >= thr;
                            pass = any;
                        } else if (ix.sap && hi - lo > (uint64_t)kRankMin) {
                            pass = (uint64_t)ix.sap[hi - 1u] >= thr;
                        }
                    }
                }
            }
            word |= (unsigned long long)(pass ? 1u : 0u) << b;
        }
        out[w] = word;
    }
}

template <class SlotT>
__global__ __launch_bounds__(256) void cache_get_kernel(IndexView<SlotT> ix,
                                                        const uint8_t *__restrict__ pats,
                                                        int64_t n_pat, uint64_t *__restrict__ lo,
                                                        uint64_t *__restrict__ hi) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_pat) return;
    uint32_t pre24 = 0;
    for (int j = 0; j < kCacheLen; ++j) pre24 = (pre24 << 3) | base_code(pats[t * kCacheLen + j]);
    uint32_t c8;
    if (cache8_index(pre24, c8)) {
        lo[t] = ix.c8lo[c8];
        hi[t] = ix.c8hi[c8];
    } else {
        lo[t] = hi[t] = 0;
    }
}

template <class SlotT>
__global__ __launch_bounds__(256) void pattern_search_kernel(IndexView<SlotT> ix,
                                                             const uint8_t *__restrict__ pats,
                                                             int64_t n_pat,
                                                             uint64_t *__restrict__ lo,
                                                             uint64_t *__restrict__ hi) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_pat) return;
    uint64_t q = 0, q2 = 0;
    for (int j = 0; j < ix.k; ++j) {
        const uint64_t c = base_code(pats[t * ix.k + j]);
        if (j < kMaxKey) q = (q << 3) | c;
        else if (j < 2 * kMaxKey) q2 = (q2 << 3) | c;
    }
    uint64_t l, h;
    ProbeRef pr;
    pr.p = pats + t * ix.k;
    kmer_range(ix, q, q2, pr, l, h);
    lo[t] = l;
    hi[t] = h;
}

// O(n) verifier of the resident suffix array (the rank trick of oracle/sais.c::oracle_sa_check):
// isa[sa[r]] = r must invert sa, and adjacent suffixes must be in strictly increasing bytewise order,
// decided by their first bytes and the ranks of the suffixes one position further.
template <class SlotT>
__global__ __launch_bounds__(256) void isa_scatter_kernel(const SlotT *__restrict__ sa, SlotT *__restrict__ isa,
                                                          uint64_t n) {
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (uint64_t)gridDim.x * blockDim.x)
        if ((uint64_t)sa[r] < n) isa[sa[r]] = (SlotT)r;
}
template <class SlotT>
__global__ __launch_bounds__(256) void sa_check_kernel(const uint8_t *__restrict__ text, const SlotT *__restrict__ sa,
                                                       const SlotT *__restrict__ isa, uint64_t n,
                                                       unsigned long long *__restrict__ errs) {
    unsigned long long n_bad = 0;
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t b = sa[r];
        bool bad = b >= n || (uint64_t)isa[b] != r;
        if (!bad && r > 0) {
            const uint64_t a = sa[r - 1];
            if (a >= n) bad = true;
            else {
                const uint64_t ra = a + 1 < n ? (uint64_t)isa[a + 1] + 1u : 0u;
                const uint64_t rb = b + 1 < n ? (uint64_t)isa[b + 1] + 1u : 0u;
                bad = !(text[a] < text[b] || (text[a] == text[b] && ra < rb));
            }
        }
        n_bad += bad ? 1u : 0u;
    }
    for (int off = 32; off > 0; off >>= 1) n_bad += __shfl_down(n_bad, off);
    if (n_bad && (threadIdx.x & 63) == 0) atomicAdd(errs, n_bad);
}

// ---- host ------------------------------------------------------------------

static int32_t check_device(int device) {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        (void)hipGetLastError();
        set_error("no HIP device available (%s); libasgart_hip has no CPU fallback",
                  e == hipSuccess ? "device count 0" : hipGetErrorString(e));
        return ASGART_E_HIP;
    }
    if (device < 0 || device >= count) {
        set_error("device %d out of range (count %d)", device, count);
        return ASGART_E_ARG;
    }
    HIP_TRY(hipSetDevice(device));
    return 0;
}

static inline unsigned grid_for(uint64_t n, unsigned block = 256) {
    return (unsigned)((n + block - 1) / block);
}
// for grid-stride kernels over up to 2^33 elements: a dispatch holds fewer than 2^32 work-items
static inline unsigned grid_capped(uint64_t n, unsigned block = 256) {
    const uint64_t g = (n + block - 1) / block;
    return (unsigned)(g < (1ull << 22) ? g : (1ull << 22));
}

// the presence filters of every orientation, their position bitmaps and the "no memory for it" marks: ONE place
// (option kfilter_bits used to drop the hashed tables only: the bitmaps leaked on the rebuild and went on filtering
// after the filter had been switched off)
static void free_filters(asgart_index *idx) {
    for (auto &f : idx->d_filter) {
        if (f) dev_free(f);
        f = nullptr;
    }
    for (auto &f : idx->d_pbits) {
        if (f) dev_free(f);
        f = nullptr;
    }
    for (auto &f : idx->filter_off) f = false;
    for (auto &f : idx->pbits_learn) f = false;
    for (auto &u : idx->pbits_uses) u = 0;
    idx->filter_bits = 0;
}

// (the prefix table stays: its size depends on the text length alone unless k is tiny, and it was allocated when the
// index was created -- see prealloc_ptab; asgart_index_destroy frees it)
static void free_k_specific(asgart_index *idx) {
    if (idx->d_keys) dev_free(idx->d_keys);
    if (idx->d_c8lo) dev_free(idx->d_c8lo);
    if (idx->d_c8hi) dev_free(idx->d_c8hi);
    if (idx->d_sap) dev_free(idx->d_sap);
    idx->d_sap = nullptr;
    free_filters(idx);
    idx->d_keys = nullptr;
    idx->d_c8lo = idx->d_c8hi = nullptr;
    idx->filter_bits = 0;
    idx->k = 0;
    idx->sap_tried = false;
    idx->calls_total = 0;
    for (auto &c : idx->mode_calls) c = 0;
    idx->split_blocked.clear();
}

static int choose_depth(int64_t n, uint64_t k, int64_t forced) {
    int d = 12;
    if (forced > 0) d = (int)forced;
    else {
        // ~one table entry per suffix, between 4^6 and 4^15 (measured at n = 3.1 G: search
        // 61 / 53 / 47 / 45 ms per launch for d = 12 / 13 / 14 / 15)
        int bits = 0;
        while ((1ll << bits) < n) ++bits;
        d = (bits + 1) / 2;
        if (d < 6) d = 6;
        if (d > 15) d = 15;
    }
    if (d < 1) d = 1;
    if (d > 15) d = 15;  // 4^16 entries would overflow the 32-bit table index
    if ((uint64_t)d > k) d = (int)k;
    return d;
}

// The prefix table (4^d + 1 slots: 4 GiB at GRCh38 size) is allocated when the index is CREATED, before the suffix sorter
// takes and gives back its 127 GB of scratch: the first allocation that finds no block of its size in the block cache after
// that much memory went back to the driver takes more than a second on many boxes of the pool (ASGART_TRACE_ALLOC=1:
// "hipMalloc 4.0 GiB: 1295.6 ms" -- it was 1.3 of the 1.4 s that "keys + tables" took there, against 0.13 s on the others).
static void prealloc_ptab(asgart_index *idx) {
    if (idx->trimmed || idx->d_ptab) return;
    const int d = choose_depth(idx->n_sa, (uint64_t)kMaxKey, idx->opt.ptab_depth);
    const uint64_t entries = (1ull << (2 * d)) + 1;
    if (dev_malloc(&idx->d_ptab, entries * (idx->wide ? 8 : 4)) != hipSuccess) {
        (void)hipGetLastError();
        idx->d_ptab = nullptr;  // (asgart_index_prepare tries again)
        return;
    }
    idx->ptab_entries = entries;
}

// true iff every byte of the device text is one of {$,A,C,G,N,T}
int32_t text_is_dna(const uint8_t *d_text, int64_t n, hipStream_t s, bool *dna) {
    DevBuf hist_b;
    RC_TRY(hist_b.reserve(256 * sizeof(unsigned long long)));
    unsigned long long *d_hist = hist_b.as<unsigned long long>();
    unsigned long long hist[256];
    int32_t rc = [&]() -> int32_t {
        HIP_TRY(hipMemsetAsync(d_hist, 0, sizeof(hist), s));
        unsigned blocks = grid_for((uint64_t)n, 256 * 16);
        if (blocks > 4096) blocks = 4096;
        byte_histogram_kernel<<<blocks, 256, 0, s>>>(d_text, (uint64_t)n, d_hist);
        HIP_TRY(hipGetLastError());
        HIP_TRY(read_back(hist, d_hist, sizeof(hist), s));
        HIP_TRY(stream_sync(s));
        return 0;
    }();
    hist_b.release();
    if (rc) return rc;
    *dna = true;
    for (int c = 0; c < 256; ++c)
        if (hist[c] && !valid_text_byte((uint8_t)c)) *dna = false;
    return 0;
}

// Position-sorted occurrence lists (one key word per probe, whole suffix array, 32-bit positions); the caller holds
// every call context.  An optimisation only: without the memory for it -- the list or the sort's scratch -- the index
// does without.
static int32_t build_sap_locked(asgart_index *idx, uint64_t k) {
    idx->sap_tried = true;
    const uint64_t n_sa = (uint64_t)idx->n_sa;
    if (!(idx->opt.rank_lists && !idx->trimmed && k <= (uint64_t)kMaxKey && n_sa > 0) || idx->d_sap) return 0;
    const size_t slot = idx->wide ? 8 : 4;
    if (dev_malloc(&idx->d_sap, (n_sa + 16) * slot) != hipSuccess) {
        (void)hipGetLastError();
        idx->d_sap = nullptr;
        return 0;
    }
    hipStream_t s = idx->ctx[0].stream;
    // (kRankMin of pipeline_dev.hpp: rank_count_kernel only consults the list of an interval of more than 256 entries)
    const int32_t rc_rank =
        idx->wide ? build_rank_lists_runs<uint64_t>(idx->d_keys, (const uint64_t *)idx->d_sa, n_sa, (uint64_t *)idx->d_sap, 256u, (int)k, s)
                  : build_rank_lists_runs<uint32_t>(idx->d_keys, (const uint32_t *)idx->d_sa, n_sa, (uint32_t *)idx->d_sap, 256u, (int)k, s);
    if (rc_rank != 0) {
        dev_free(idx->d_sap);
        idx->d_sap = nullptr;
        if (rc_rank != ASGART_E_OOM) return rc_rank;
    }
    if (idx->opt.cache_calls == 0 || idx->calls_total >= (uint64_t)idx->opt.cache_calls) BlockCache::trim();  // (the sort's scratch)
    return 0;
}

// ... for the search path: builds them when the index has none yet and has not tried (keys prepared first)
int32_t index_prepare_sap(asgart_index *idx, uint64_t k) {
    RC_TRY(index_prepare(idx, k));
    {
        std::lock_guard<std::mutex> lk(idx->mu);
        if (idx->k == k && (idx->d_sap || idx->sap_tried)) return 0;
    }
    REFUSE_POISONED(idx);
    idx->acquire_all();
    struct Unlock {
        asgart_index *i;
        ~Unlock() { i->release_all(); }
    } unlock{idx};
    if (idx->k != k || idx->d_sap || idx->sap_tried) return 0;
    HIP_TRY(hipSetDevice(idx->device));
    return build_sap_locked(idx, k);
}

int32_t index_prepare(asgart_index *idx, uint64_t k) {
    if (k < (uint64_t)kCacheLen || k > (uint64_t)kMaxK) {
        set_error("probe_size %llu unsupported: need %d <= k <= %d (the reference needs k >= 8, "
                  "src/searcher.rs:95-97; this build compares a 63-bit key word plus at most 21 more bases)",
                  (unsigned long long)k, kCacheLen, kMaxK);
        return ASGART_E_ARG;
    }
    {
        std::lock_guard<std::mutex> lk(idx->mu);
        if (idx->k == k) return 0;
    }
    REFUSE_POISONED(idx);
    idx->acquire_all();  // no search call may be using the old keys
    struct Unlock {
        asgart_index *i;
        ~Unlock() { i->release_all(); }
    } unlock{idx};
    if (idx->k == k) return 0;
    HIP_TRY(hipSetDevice(idx->device));
    free_k_specific(idx);
    auto t0 = std::chrono::steady_clock::now();
    auto t_lap = t0;
    auto lap = [&](const char *what) {  // option debug: where the build's wall time goes
        if (!idx->opt.debug) return;
        (void)hipDeviceSynchronize();
        const auto t = std::chrono::steady_clock::now();
        fprintf(stderr, "[asgart] index_prepare: %-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(t - t_lap).count());
        t_lap = t;
    };
    const uint64_t n = (uint64_t)idx->n, n_sa = (uint64_t)idx->n_sa;
    const size_t slot = idx->wide ? 8 : 4;
    // a --trim index is searched through the 8-mer cache like the reference: no prefix table
    const uint64_t kk = std::min<uint64_t>(k, (uint64_t)kMaxKey);  // bases in a key word
    const int d = idx->trimmed ? 1 : choose_depth(idx->n_sa, kk, idx->opt.ptab_depth);
    const uint64_t entries = (1ull << (2 * d)) + 1;
    HIP_TRY(dev_malloc((void **)&idx->d_keys, (n_sa + 16) * sizeof(uint64_t)));
    if (idx->d_ptab && idx->ptab_entries < entries) {  // (allocated at creation for another depth)
        dev_free(idx->d_ptab);
        idx->d_ptab = nullptr;
    }
    if (!idx->d_ptab) {
        HIP_TRY(dev_malloc(&idx->d_ptab, entries * slot));
        idx->ptab_entries = entries;
    }
    HIP_TRY(dev_malloc(&idx->d_c8lo, (size_t)kCacheEntries * slot));
    HIP_TRY(dev_malloc(&idx->d_c8hi, (size_t)kCacheEntries * slot));
    hipStream_t s = idx->ctx[0].stream;
    idx->n_bad = 0;
    lap("allocations");
    auto build = [&](auto tag) -> int32_t {
        using SlotT = decltype(tag);
        const SlotT *sa = (const SlotT *)idx->d_sa;
        build_keys_kernel<SlotT><<<grid_capped(n_sa), 256, 0, s>>>(idx->d_text, sa, idx->d_keys, n, n_sa, (int)kk);
        if (!idx->trimmed) {
            build_ptab_kernel<SlotT><<<grid_for(entries), 256, 0, s>>>(idx->d_keys, (SlotT *)idx->d_ptab, n_sa, (int)kk, d);
            build_cache8_kernel<SlotT><<<grid_for(kCacheEntries), 256, 0, s>>>(
                idx->d_keys, (SlotT *)idx->d_c8lo, (SlotT *)idx->d_c8hi, n_sa, (int)kk);
            HIP_TRY(hipGetLastError());
            return 0;
        }
        build_cache8_trim_kernel<SlotT><<<grid_for(kCacheEntries), 256, 0, s>>>(
            idx->d_text, n, sa, n_sa, (SlotT *)idx->d_c8lo, (SlotT *)idx->d_c8hi);
        // the out-of-place slots: sub-strand suffixes shorter than k
        DevBuf &cb = idx->ctx[0].ws.counters;
        RC_TRY(cb.reserve(256 * 8));
        unsigned long long *d_cnt = cb.as<unsigned long long>(), *d_out = d_cnt + 8;
        HIP_TRY(hipMemsetAsync(d_cnt, 0, 8, s));
        const uint64_t first_bad = (uint64_t)idx->trim_end >= k ? (uint64_t)idx->trim_end - k + 1 : 0;
        bad_slots_kernel<SlotT><<<grid_capped(n_sa), 256, 0, s>>>(sa, n_sa, first_bad, d_out, d_cnt, kMaxK + 2);
        HIP_TRY(hipGetLastError());
        unsigned long long h[8 + kMaxK + 2];
        HIP_TRY(read_back(h, d_cnt, sizeof(h), s));
        HIP_TRY(stream_sync(s));
        idx->n_bad = (int)std::min<unsigned long long>(h[0], kMaxK + 2);
        for (int j = 0; j < idx->n_bad; ++j) idx->bad[j] = h[8 + j];
        return 0;
    };
    RC_TRY(idx->wide ? build(uint64_t{}) : build(uint32_t{}));
    HIP_TRY(hipGetLastError());
    HIP_TRY(stream_sync(s));
    lap("keys + tables");
    // position-sorted occurrence lists: at once (option lazy_aux = 0), or by the first search call that has a predecessor
    idx->sap_tried = false;
    if (!idx->opt.lazy_aux) {
        RC_TRY(build_sap_locked(idx, k));
        lap("position-sorted lists");
    }
    // text-tail corner list (host, from the last bytes of the text)
    idx->n_tail8 = 0;
    idx->tail_bloom = 0;
    {
        const int64_t tl = (int64_t)idx->h_tail.size();
        const int64_t base = idx->n - tl;  // text offset of h_tail[0]
        for (int64_t x = idx->n - (int64_t)k + 1; x <= idx->n - kCacheLen; ++x) {
            if (x < 0 || x < base) continue;
            uint32_t pre24 = 0;
            bool ok = true;
            for (int j = 0; j < kCacheLen; ++j) {
                uint32_t c = base_code(idx->h_tail[(size_t)(x - base + j)]);
                ok &= c != 0;
                pre24 = (pre24 << 3) | c;
            }
            if (!ok) continue;
            bool dup = false;
            for (int j = 0; j < idx->n_tail8; ++j) dup |= idx->tail8[j] == pre24;
            if (dup) continue;
            idx->tail8[idx->n_tail8++] = pre24;
            idx->tail_bloom |= 1ull << (pre24 & 63u);
        }
    }
    idx->k = k;
    idx->d = d;
    if (idx->opt.prewarm) {
        // What the first search calls would otherwise pay inside their timed part: the per-probe workspace of both call
        // contexts (an unsharded call over the whole text has at most n / step probes) and the worker thread of the
        // passes call with its per-thread runtime state.  Best effort: without the memory the calls allocate as they go.
        const uint64_t step = k / 2 ? k / 2 : 1;
        const uint64_t W = std::min<uint64_t>((uint64_t)idx->n / step + 1, 0xFFFFFF00ull);
        // (the first context for TWO passes: the direct and the -RC run of a passes call are one job over both passes' probes)
        uint64_t Wc[kNumCtx];
        for (int c = 0; c < kNumCtx; ++c) Wc[c] = c == 0 ? std::min<uint64_t>(2 * W, 0xFFFFFF00ull) : W;
        (void)carve_probe_workspace(idx, Wc);  // (out of one block the suffix sorter has just released, when there is one)
        for (int c = 0; c < kNumCtx; ++c)
            if (reserve_probe_workspace(idx, idx->ctx[c], Wc[c]) != 0) {
                (void)hipGetLastError();
                break;
            }
        // (a passes call that prepares lazily holds pass_mu itself and owns the workers: leave them to it)
        if (!tl_owns_pass_mu && idx->pass_mu.try_lock()) {
            if (idx->pass_workers.empty()) {
                idx->pass_workers.emplace_back(new asgart::PassWorker());
                const int dev = idx->device;
                idx->pass_workers[0]->submit([dev]() {
                    (void)hipSetDevice(dev);
                    (void)hipFree(nullptr);
                });
                idx->pass_workers[0]->wait();
            }
            idx->pass_mu.unlock();
        }
        lap("workspace of both contexts");
    }
    // the sorter's released scratch goes back to the device when the index has answered `cache_calls` calls (run_search_passes)
    if (idx->opt.cache_calls == 0) BlockCache::trim();
    idx->ms_prepare =
        std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return 0;
}

// Option lazy_aux: no filter is built for orientation `mode`; a blank position bitmap (all ones) is set up instead, which
// the orientation's searches fill in (RunParams::learn).  Without memory for it the orientation is searched without.
int32_t index_prepare_learned_bits(asgart_index *idx, uint64_t k, int mode) {
    RC_TRY(index_prepare(idx, k));
    if (idx->opt.kfilter_bits == 0 || idx->opt.posbits == 0 || idx->trimmed || mode < 0 || mode > 3 || k > (uint64_t)kMaxKey) return 0;
    REFUSE_POISONED(idx);
    idx->acquire_all();
    struct Unlock {
        asgart_index *i;
        ~Unlock() { i->release_all(); }
    } unlock{idx};
    if (idx->k != k || idx->d_pbits[mode] || idx->filter_off[mode]) return 0;
    HIP_TRY(hipSetDevice(idx->device));
    const uint64_t n_words = ((uint64_t)idx->n + 63u) / 64u;
    uint64_t *pb = nullptr;
    if (dev_malloc((void **)&pb, (size_t)n_words * 8 + 512) != hipSuccess) {
        (void)hipGetLastError();
        idx->filter_off[mode] = true;
        return 0;
    }
    hipStream_t s = idx->ctx[0].stream;
    const int32_t rc = [&]() -> int32_t {
        HIP_TRY(hipMemsetAsync(pb, 0xFF, (size_t)n_words * 8 + 512, s));
        HIP_TRY(stream_sync(s));
        return 0;
    }();
    if (rc != 0) {
        dev_free(pb);
        return rc;
    }
    idx->d_pbits[mode] = pb;
    idx->pbits_learn[mode] = true;
    idx->pbits_uses[mode] = 0;
    return 0;
}

// Builds the presence filter of orientation `mode` for probe size k (keys prepared first).
int32_t index_prepare_filter(asgart_index *idx, uint64_t k, int mode) {
    RC_TRY(index_prepare(idx, k));
    // (a --trim index has no filter: its array does not hold the probes' own positions)
    // (nor have probes longer than one key word: the filter is keyed by the key word)
    if (idx->opt.kfilter_bits == 0 || idx->trimmed || mode < 0 || mode > 3 || k > (uint64_t)kMaxKey) return 0;
    {
        std::lock_guard<std::mutex> lk(idx->mu);
        if (idx->k == k && (idx->d_filter[mode] || idx->filter_off[mode])) return 0;
    }
    REFUSE_POISONED(idx);
    idx->acquire_all();
    struct Unlock {
        asgart_index *i;
        ~Unlock() { i->release_all(); }
    } unlock{idx};
    if (idx->k != k || idx->d_filter[mode] || idx->filter_off[mode]) return 0;
    HIP_TRY(hipSetDevice(idx->device));
    if (!idx->filter_bits) {
        // about 8 bits per text position, at most the configured size (the default, 2^30 bits =
        // 128 MiB, fits the Infinity Cache next to the streams of a search call)
        int bits = 16;
        while (bits < 40 && (1ll << bits) < idx->n) ++bits;
        bits += 3;
        if (bits > (int)idx->opt.kfilter_bits) bits = (int)idx->opt.kfilter_bits;
        if (bits < 16) bits = 16;
        idx->filter_bits = bits;
    }
    const int bits = idx->filter_bits;
    const size_t bytes = (size_t)1 << (bits - 3);
    uint64_t *flt = nullptr;
    if (dev_malloc((void **)&flt, bytes) != hipSuccess) {
        // the filter is an optimisation: without memory for it this orientation is searched without (every probe
        // takes the lookup); the call that needed it goes on
        (void)hipGetLastError();
        idx->filter_off[mode] = true;
        return 0;
    }
    hipStream_t s = idx->ctx[0].stream;
    const bool rev = (mode & 2) != 0, comp = (mode & 1) != 0;
    int32_t rc = [&]() -> int32_t {
        HIP_TRY(hipMemsetAsync(flt, 0, bytes, s));
        const unsigned g = grid_capped((uint64_t)idx->n);
        if (idx->wide)
            build_filter_kernel<uint64_t><<<g, 256, 0, s>>>(idx->view<uint64_t>(), rev, comp,
                                                            (unsigned long long *)flt, bits);
        else
            build_filter_kernel<uint32_t><<<g, 256, 0, s>>>(idx->view<uint32_t>(), rev, comp,
                                                            (unsigned long long *)flt, bits);
        HIP_TRY(hipGetLastError());
        HIP_TRY(stream_sync(s));
        return 0;
    }();
    if (rc != 0) {
        dev_free(flt);
        return rc;
    }
    if (idx->opt.posbits) {
        // the same answers by text position (n bits, padded so that a workgroup's 16-byte loads never leave it)
        const uint64_t n_words = ((uint64_t)idx->n + 63u) / 64u;
        uint64_t *pb = nullptr;
        if (dev_malloc((void **)&pb, (size_t)n_words * 8 + 512) == hipSuccess) {
            rc = [&]() -> int32_t {
                HIP_TRY(hipMemsetAsync(pb, 0xFF, (size_t)n_words * 8 + 512, s));
                const unsigned g = grid_capped(n_words);
                const bool refine = idx->opt.posbits >= 2 && !idx->trimmed;
                if (idx->wide)
                    build_posbits_kernel<uint64_t><<<g, 256, 0, s>>>(idx->view<uint64_t>(), rev, comp, flt, bits, refine,
                                                                     (unsigned long long *)pb, n_words);
                else
                    build_posbits_kernel<uint32_t><<<g, 256, 0, s>>>(idx->view<uint32_t>(), rev, comp, flt, bits, refine,
                                                                     (unsigned long long *)pb, n_words);
                HIP_TRY(hipGetLastError());
                HIP_TRY(stream_sync(s));
                return 0;
            }();
            if (rc != 0) {
                dev_free(pb);
                dev_free(flt);
                return rc;
            }
            if (idx->d_pbits[mode]) dev_free(idx->d_pbits[mode]);
            idx->d_pbits[mode] = pb;
        } else {
            (void)hipGetLastError();  // no memory: the hashed filter serves
        }
    }
    idx->d_filter[mode] = flt;
    return 0;
}

}  // namespace asgart

using namespace asgart;

extern "C" {

const char *asgart_last_error(void) { return asgart::g_err; }

const char *asgart_version(void) { return "asgart-hip 0.2.0 gfx950"; }

// ---- native stacks of every thread (asgart_hip.h: asgart_debug_dump_stacks) ------------------------------------------
// A last resort beside a debugger from a child process (tests/conftest.py tries gdb / rocgdb first): every thread is
// sent a signal and writes its own stack from the handler.  The signal is a realtime signal of the library's own, its
// handler is installed ONCE and stays installed: a thread that has the signal blocked, or sits in an uninterruptible
// driver wait -- the stall this exists for -- takes it whenever it comes back, finds nobody asking for its stack any more
// and returns; the process is never left with a default disposition and a pending signal.  The handler formats with
// nothing but arithmetic and write(2); backtrace() has been called once beforehand so that libgcc is loaded, and
// backtrace_symbols_fd() does not allocate (glibc) -- what remains is that the unwinder may take the loader lock: a thread
// stopped INSIDE dl_iterate_phdr's critical section could block there, which is why the requester waits for every
// thread's own acknowledgement with a time limit and goes on without it.
namespace {
std::atomic<long> g_dump_want{0};   // the thread whose stack is being asked for (0: nobody's)
std::atomic<long> g_dump_ack{0};    // ... and the last thread that wrote one
int g_dump_signal = 0;
void put_line(const char *pre, long v, const char *post) {  // (async-signal-safe: no stdio)
    char buf[96];
    size_t n = 0;
    for (const char *c = pre; *c && n < 48; ++c) buf[n++] = *c;
    char dig[24];
    int nd = 0;
    unsigned long u = v < 0 ? 0ul - (unsigned long)v : (unsigned long)v;
    do {
        dig[nd++] = (char)('0' + u % 10u);
        u /= 10u;
    } while (u && nd < 22);
    if (v < 0) buf[n++] = '-';
    while (nd) buf[n++] = dig[--nd];
    for (const char *c = post; *c && n + 1 < sizeof buf; ++c) buf[n++] = *c;
    (void)!write(2, buf, n);
}
void dump_stack_handler(int) {
    const int saved_errno = errno;
    const long me = (long)syscall(SYS_gettid);
    if (g_dump_want.load(std::memory_order_acquire) == me) {  // (a late delivery: nobody is asking any more)
        void *frames[64];
        const int n = backtrace(frames, 64);
        put_line("---- native stack of thread ", me, " ----\n");
        backtrace_symbols_fd(frames, n, 2);
        g_dump_ack.store(me, std::memory_order_release);
    }
    errno = saved_errno;
}
}  // namespace

int32_t asgart_debug_dump_stacks(void) {
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    if (!g_dump_signal) {
        // (the first backtrace of a process loads libgcc: not from inside a signal handler)
        void *warm[4];
        (void)backtrace(warm, 4);
        const int sig = SIGRTMIN + 6;  // (a realtime signal nothing else in the process is expected to use)
        if (sig > SIGRTMAX) return -1;
        struct sigaction sa;
        memset(&sa, 0, sizeof sa);
        sa.sa_handler = dump_stack_handler;
        sa.sa_flags = SA_RESTART;
        sigemptyset(&sa.sa_mask);
        if (sigaction(sig, &sa, nullptr) != 0) return -1;  // installed once, never removed
        g_dump_signal = sig;
    }
    int asked = 0;
    const pid_t pid = getpid();
    if (DIR *d = opendir("/proc/self/task")) {
        std::vector<long> tids;
        while (struct dirent *e = readdir(d)) {
            char *end = nullptr;
            const long tid = strtol(e->d_name, &end, 10);
            if (end && *end == 0 && tid > 0) tids.push_back(tid);
        }
        closedir(d);
        for (long tid : tids) {
            g_dump_want.store(tid, std::memory_order_release);
            if (syscall(SYS_tgkill, pid, (pid_t)tid, g_dump_signal) != 0) continue;
            ++asked;
            // one at a time (the dumps would interleave); every thread acknowledges for itself, and one that cannot take
            // the signal now is given 0.5 s -- its delivery, whenever it comes, finds g_dump_want moved on and returns
            bool answered = false;
            for (int spin = 0; spin < 500 && !answered; ++spin) {
                answered = g_dump_ack.load(std::memory_order_acquire) == tid;
                if (!answered) std::this_thread::sleep_for(std::chrono::milliseconds(1));
            }
            if (!answered) put_line("---- thread ", tid, " did not answer within 0.5 s (signal blocked, or in an uninterruptible wait) ----\n");
        }
        g_dump_want.store(0, std::memory_order_release);
        g_dump_ack.store(0, std::memory_order_release);
    }
    return asked;
}

// Teardown in stages, none of which can hold the host for ever (three stalls of round 4 ended in here or in a
// build's read-back, on boxes that were slow to begin with; no wait of this function was polled then):
//   1  every stream of both call contexts is drained by a POLLED wait (limit: option watchdog_s, 0 = for ever).  Streams
//      that do not drain -- or an index the watchdog has given up on whose streams still do not -- mean work is in
//      flight on the buffers: nothing is freed, the index is leaked, the reason goes to stderr and asgart_last_error;
//   2  the worker threads of the passes call are joined (idle by now: a passes call in flight would own the contexts);
//   3  pinned host blocks are freed, 4 device buffers given back, 5 events and streams destroyed, 6 the block cache
//      trimmed when this was the device's last index -- every one of them a runtime call that synchronises with the
//      device inside the runtime and cannot be polled: they run on a helper thread that is waited for watchdog_s
//      seconds PER STAGE; a stage that does not come back is reported (stage name, index, seconds) and the rest of the
//      teardown is abandoned (leak) instead of hanging the caller.
// ASGART_DEBUG=1 (option debug) stamps every stage on stderr.
void asgart_index_destroy(asgart_index *idx) {
    if (!idx) return;
    const bool dbg = idx->opt.debug != 0;
    const double limit = (double)idx->opt.watchdog_s;
    // (an index the watchdog has already given up on: its streams are given a few more seconds, not another full limit)
    const double drain_limit = idx->poisoned.load() ? (limit > 0.0 && limit < 5.0 ? limit : 5.0) : limit;
    const auto t0 = std::chrono::steady_clock::now();
    auto since = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
    auto stamp = [&](const char *what) {
        if (dbg) fprintf(stderr, "[asgart] destroy %p: %s (+%.1f ms)\n", (void *)idx, what, since());
    };
    auto give_up = [&](const char *stage) {
        set_error("asgart_index_destroy: %s did not finish within %.0f s (option watchdog_s); the index and its device memory "
                  "are left behind (leaked) instead of hanging the caller -- report and exit, or continue in a fresh process",
                  stage, limit);
        fprintf(stderr, "[asgart] %s\n", asgart_last_error());
        idx->poisoned.store(true);
    };
    const int dev = idx->device;
    (void)hipSetDevice(dev);
    stamp("begin");
    // ---- 1: polled drain -------------------------------------------------------------------------------------------
    for (int c = 0; c < kNumCtx; ++c) {
        SearchCtx &cx = idx->ctx[c];
        int j = 0;
        for (hipStream_t st : {cx.stream, cx.stream2, cx.stream3, cx.stream4, cx.stream5, cx.stream6, cx.stream7}) {
            ++j;
            if (!st) continue;
            const auto t1 = std::chrono::steady_clock::now();
            for (unsigned spins = 1;; ++spins) {
                const hipError_t q = hipStreamQuery(st);
                if (q != hipErrorNotReady) {
                    (void)hipGetLastError();
                    break;  // (idle, or an error the frees below will meet again: nothing to wait for either way)
                }
                (void)hipGetLastError();
                if (spins > 256) std::this_thread::sleep_for(std::chrono::microseconds(spins > 4096 ? 200 : 20));
                if (drain_limit > 0.0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count() > drain_limit) {
                    char what[64];
                    snprintf(what, sizeof what, "the drain of stream %d of call context %d", j, c);
                    give_up(what);
                    return;
                }
            }
        }
    }
    stamp("streams drained");
    // ---- 2: workers ------------------------------------------------------------------------------------------------
    if (!bounded_call(limit, [idx]() { idx->pass_workers.clear(); })) {
        give_up("joining the worker threads of the passes call");
        return;
    }
    stamp("workers joined");
    // ---- 3: events and streams -------------------------------------------------------------------------------------
    // They go AFTER the memory (stages 4 and 5 run first): destroyed before it, the runtime kept 16 MB per 70 indexes of the
    // failure-injection test (test_out_of_memory_paths_release_what_they_hold; the frees-first order of round 4 does
    // not).  ASGART_TEARDOWN_ORDER=0 (diagnostics) destroys them first.
    static const bool streams_last = !(getenv("ASGART_TEARDOWN_ORDER") && atoi(getenv("ASGART_TEARDOWN_ORDER")) == 0);
    auto destroy_streams = [&]() -> bool {
      return bounded_call(limit, [idx, dev]() {
            (void)hipSetDevice(dev);
            for (auto &cx : idx->ctx) {
                for (auto &e : cx.ev)
                    if (e) {
                        (void)hipEventDestroy(e);
                        e = nullptr;
                    }
                for (hipStream_t *st : {&cx.stream, &cx.stream2, &cx.stream3, &cx.stream4, &cx.stream5, &cx.stream6, &cx.stream7})
                    if (*st) {
                        (void)hipStreamDestroy(*st);
                        *st = nullptr;
                    }
            }
        });
    };
    if (!streams_last) {
        if (!destroy_streams()) {
            give_up("destroying the events and streams");
            return;
        }
        stamp("events and streams destroyed");
    }
    // ---- 4: pinned host memory (the device-mapped heartbeat block among it) ---------------------------------------------
    if (!bounded_call(limit, [idx, dev]() {
            (void)hipSetDevice(dev);
            for (auto &cx : idx->ctx) {
                if (cx.h_pinned) (void)hipHostFree(cx.h_pinned);
                cx.h_pinned = nullptr;
                cx.h_pinned_cap = 0;
                if (cx.h_ctl) (void)hipHostFree(cx.h_ctl);
                cx.h_ctl = nullptr;
                cx.h_ctl_cap = 0;
                if (cx.h_hb) (void)hipHostFree(cx.h_hb);
                cx.h_hb = cx.d_hb = nullptr;
            }
        })) {
        give_up("freeing the pinned host blocks");
        return;
    }
    stamp("pinned host memory freed");
    // ---- 5: device buffers (to the block cache, or to the device) ----------------------------------------------------------
    if (!bounded_call(limit, [idx, dev]() {
            (void)hipSetDevice(dev);
            free_k_specific(idx);
            if (idx->d_ptab) dev_free(idx->d_ptab);
            if (idx->d_text) dev_free(idx->d_text);
            if (idx->d_sa) dev_free(idx->d_sa);
            idx->d_ptab = nullptr;
            idx->d_text = nullptr;
            idx->d_sa = nullptr;
            for (auto &cx : idx->ctx) cx.ws.release_all();  // (the record-ordering buffers used to be missing from a list kept here)
            idx->ws_arena.release();
        })) {
        give_up("releasing the device buffers");
        return;
    }
    stamp("device buffers released");
    if (streams_last) {
        if (!destroy_streams()) {
            give_up("destroying the events and streams");
            return;
        }
        stamp("events and streams destroyed");
    }
    delete idx;
    // ---- 6: the last index of the device takes the cached blocks with it -------------------------------------------------------
    if (!bounded_call(limit, [dev]() {
            (void)hipSetDevice(dev);
            BlockCache::index_gone();
        })) {
        set_error("asgart_index_destroy: giving the cached device blocks back did not finish within %.0f s (option watchdog_s)", limit);
        fprintf(stderr, "[asgart] %s\n", asgart_last_error());
        return;
    }
    if (dbg) fprintf(stderr, "[asgart] destroy: done (+%.1f ms)\n", since());
}

static int32_t index_create_impl(const uint8_t *T, int64_t n, const int64_t *SA, int64_t sa_len,
                                 bool trimmed, int64_t trim_start, int64_t trim_end, int32_t device,
                                 asgart_index **out) {
    if (!out) {
        set_error("asgart_index_create: out is NULL");
        return ASGART_E_ARG;
    }
    *out = nullptr;
    if (!T || n <= 0) {
        set_error("asgart_index_create: empty text");
        return ASGART_E_ARG;
    }
    if (trimmed && !(0 <= trim_start && trim_start < trim_end && trim_end <= n - 1)) {
        // what prepare_data lets through (reference src/bin/asgart.rs:432-463): start < end <= len - 1
        set_error("asgart_index_create_trim: need 0 <= start (%lld) < end (%lld) <= n - 1 (%lld)",
                  (long long)trim_start, (long long)trim_end, (long long)(n - 1));
        return ASGART_E_ARG;
    }
    const int64_t n_sa = trimmed ? trim_end - trim_start + 1 : n;  // + the '$' of the sub-strand
    if (SA && sa_len != n_sa) {
        set_error("asgart_index_create: sa_len (%lld) != %lld (the text length, or end - start + 1 for a "
                  "trimmed index)", (long long)sa_len, (long long)n_sa);
        return ASGART_E_ARG;
    }
    RC_TRY(check_device(device));
    asgart_index *idx = new (std::nothrow) asgart_index();
    if (!idx) {
        set_error("out of host memory");
        return ASGART_E_OOM;
    }
    BlockCache::index_born();  // (check_device made `device` current)
    idx->device = device;
    idx->n = n;
    idx->n_sa = n_sa;
    idx->trimmed = trimmed;
    idx->trim_start = trim_start;
    idx->trim_end = trim_end;
    options_from_env(idx->opt);  // the only place the environment is read
    // force_wide (tests): 64-bit slots and positions also for a small text, so that the
    // instantiations a > 4 Gb input selects can be checked against the oracle
    idx->wide = (uint64_t)n >= 0xFFFFFF00ull || idx->opt.force_wide != 0;
    for (auto &cx : idx->ctx) memset(&cx.stats, 0, sizeof(cx.stats));
    int32_t rc = [&]() -> int32_t {
        for (auto &cx : idx->ctx) RC_TRY(create_ctx_streams(cx));
        HIP_TRY(dev_malloc((void **)&idx->d_text, (size_t)n + 64));
        prealloc_ptab(idx);
        HIP_TRY(hipMemsetAsync(idx->d_text + n, 0, 64, idx->ctx[0].stream));
        HIP_TRY(hipMemcpyAsync(idx->d_text, T, (size_t)n, hipMemcpyHostToDevice, idx->ctx[0].stream));
        // validate the alphabet on the device
        RC_TRY(idx->ctx[0].ws.counters.reserve(256 * sizeof(unsigned long long)));
        unsigned long long *d_hist = idx->ctx[0].ws.counters.as<unsigned long long>();
        HIP_TRY(hipMemsetAsync(d_hist, 0, 256 * sizeof(unsigned long long), idx->ctx[0].stream));
        unsigned blocks = grid_for((uint64_t)n, 256 * 16);
        if (blocks > 4096) blocks = 4096;
        byte_histogram_kernel<<<blocks, 256, 0, idx->ctx[0].stream>>>(idx->d_text, (uint64_t)n, d_hist);
        HIP_TRY(hipGetLastError());
        unsigned long long hist[256];
        HIP_TRY(read_back(hist, d_hist, sizeof(hist), idx->ctx[0].stream));
        HIP_TRY(stream_sync(idx->ctx[0].stream));
        for (int c = 0; c < 256; ++c)
            if (hist[c] && !valid_text_byte((uint8_t)c)) {
                set_error("text contains byte 0x%02x; expected normalised bases {A,C,G,T,N} "
                          "plus a final '$' (reference src/bin/asgart.rs:289-301,430)", c);
                return ASGART_E_ARG;
            }
        if (hist['$'] > 1 || (hist['$'] == 1 && T[n - 1] != '$')) {
            set_error("'$' may only appear once, as the last byte of the text");
            return ASGART_E_ARG;
        }
        const int64_t tl = n < (int64_t)kMaxK + 32 ? n : (int64_t)kMaxK + 32;  // (the text-tail corner list reads the last k bytes)
        idx->h_tail.assign(T + n - tl, T + n);
        const size_t slot = idx->wide ? 8 : 4;
        HIP_TRY(dev_malloc(&idx->d_sa, ((size_t)n_sa + 16) * slot));
        if (!SA && trimmed) {
            // suffix array of data[start..end] + '$', every entry shifted by +start
            // (reference src/bin/asgart.rs:142-148)
            hipStream_t s = idx->ctx[0].stream;
            DevBuf sub;
            RC_TRY(sub.reserve((size_t)n_sa + 64));
            int32_t rc2 = [&]() -> int32_t {
                HIP_TRY(hipMemsetAsync(sub.p, 0, (size_t)n_sa + 64, s));
                HIP_TRY(hipMemcpyAsync(sub.p, idx->d_text + trim_start, (size_t)(n_sa - 1), hipMemcpyDeviceToDevice, s));
                HIP_TRY(hipMemsetAsync(sub.as<uint8_t>() + (n_sa - 1), '$', 1, s));
                RC_TRY(sa_build_device(sub.as<uint8_t>(), n_sa, idx->d_sa, idx->wide, s,
                                       (uint64_t)idx->opt.test_wide_batch));
                if (idx->wide)
                    add_offset_kernel<uint64_t><<<grid_capped((uint64_t)n_sa), 256, 0, s>>>((uint64_t *)idx->d_sa, (uint64_t)n_sa, (uint64_t)trim_start);
                else
                    add_offset_kernel<uint32_t><<<grid_capped((uint64_t)n_sa), 256, 0, s>>>((uint32_t *)idx->d_sa, (uint64_t)n_sa, (uint64_t)trim_start);
                HIP_TRY(hipGetLastError());
                HIP_TRY(stream_sync(s));
                return 0;
            }();
            sub.release();
            RC_TRY(rc2);
        } else if (!SA) {
            RC_TRY(sa_build_device(idx->d_text, n, idx->d_sa, idx->wide, idx->ctx[0].stream,
                                   (uint64_t)idx->opt.test_wide_batch));
        } else if (idx->wide) {
            HIP_TRY(hipMemcpyAsync(idx->d_sa, SA, (size_t)n_sa * 8, hipMemcpyHostToDevice,
                                   idx->ctx[0].stream));
            HIP_TRY(stream_sync(idx->ctx[0].stream));
        } else {
            const uint64_t slice = 1ull << 25;
            DevBuf stage;
            RC_TRY(stage.reserve((size_t)(slice < (uint64_t)n_sa ? slice : (uint64_t)n_sa) * 8));
            for (uint64_t off = 0; off < (uint64_t)n_sa; off += slice) {
                uint64_t cnt = (uint64_t)n_sa - off < slice ? (uint64_t)n_sa - off : slice;
                hipError_t e = hipMemcpyAsync(stage.p, SA + off, cnt * 8, hipMemcpyHostToDevice,
                                              idx->ctx[0].stream);
                if (e == hipSuccess) {
                    narrow_sa_kernel<<<grid_for(cnt), 256, 0, idx->ctx[0].stream>>>(
                        stage.as<int64_t>(), (uint32_t *)idx->d_sa + off, cnt);
                    e = hipStreamSynchronize(idx->ctx[0].stream);
                }
                if (e != hipSuccess) {
                    stage.release();
                    HIP_TRY(e);
                }
            }
            stage.release();
        }
        return 0;
    }();
    if (rc != 0) {
        asgart_index_destroy(idx);
        return rc;
    }
    *out = idx;
    return 0;
}

int32_t asgart_index_create(const uint8_t *T, int64_t n, const int64_t *SA, int64_t sa_len,
                            int32_t device, asgart_index **out) {
    return index_create_impl(T, n, SA, sa_len, false, 0, 0, device, out);
}

int32_t asgart_index_create_trim(const uint8_t *T, int64_t n, const int64_t *SA, int64_t sa_len,
                                 int64_t trim_start, int64_t trim_end, int32_t device,
                                 asgart_index **out) {
    return index_create_impl(T, n, SA, sa_len, true, trim_start, trim_end, device, out);
}

int32_t asgart_index_clone(asgart_index *src, int32_t device, asgart_index **out) {
    if (!out) {
        set_error("asgart_index_clone: out is NULL");
        return ASGART_E_ARG;
    }
    *out = nullptr;
    if (!src) {
        set_error("asgart_index_clone: source index is NULL");
        return ASGART_E_ARG;
    }
    RC_TRY(check_device(device));
    asgart_index *idx = new (std::nothrow) asgart_index();
    if (!idx) {
        set_error("out of host memory");
        return ASGART_E_OOM;
    }
    BlockCache::index_born();
    idx->device = device;
    idx->n = src->n;
    idx->n_sa = src->n_sa;
    idx->trimmed = src->trimmed;
    idx->trim_start = src->trim_start;
    idx->trim_end = src->trim_end;
    idx->wide = src->wide;
    idx->opt = src->opt;
    idx->h_tail = src->h_tail;
    for (auto &cx : idx->ctx) memset(&cx.stats, 0, sizeof(cx.stats));
    src->acquire_all();  // the source's buffers must not change under the copy
    int32_t rc = [&]() -> int32_t {
        HIP_TRY(hipSetDevice(device));
        for (auto &cx : idx->ctx) RC_TRY(create_ctx_streams(cx));
        const size_t slot = idx->wide ? 8 : 4;
        const size_t text_bytes = (size_t)idx->n + 64, sa_bytes = ((size_t)idx->n_sa + 16) * slot;
        HIP_TRY(dev_malloc((void **)&idx->d_text, text_bytes));
        HIP_TRY(dev_malloc(&idx->d_sa, sa_bytes));
        // device-to-device over xGMI when the devices differ (peer copy), a plain copy otherwise
        hipStream_t s = idx->ctx[0].stream;
        HIP_TRY(hipMemcpyPeerAsync(idx->d_text, device, src->d_text, src->device, text_bytes, s));
        HIP_TRY(hipMemcpyPeerAsync(idx->d_sa, device, src->d_sa, src->device, sa_bytes, s));
        HIP_TRY(stream_sync(s));
        return 0;
    }();
    src->release_all();
    if (rc != 0) {
        asgart_index_destroy(idx);
        return rc;
    }
    *out = idx;
    return 0;
}

int64_t asgart_trim_cache(int32_t device) {
    RC_TRY(check_device(device));
    (void)bounded_call(60.0, []() {});  // (a trim a search call left to the background worker: let it finish first)
    const size_t held = BlockCache::held();
    BlockCache::trim();
    return (int64_t)held;
}

int32_t asgart_index_export(asgart_index *idx, const void **d_text, const void **d_sa, int32_t *sa_entry_bytes) {
    if (!idx) {
        set_error("index is NULL");
        return ASGART_E_ARG;
    }
    if (d_text) *d_text = idx->d_text;
    if (d_sa) *d_sa = idx->d_sa;
    if (sa_entry_bytes) *sa_entry_bytes = idx->wide ? 8 : 4;
    return 0;
}

int32_t asgart_index_create_device(const void *d_text, int64_t n, const void *d_sa, int64_t sa_len,
                                   int32_t sa_entry_bytes, int32_t device, asgart_index **out) {
    if (!out) {
        set_error("asgart_index_create_device: out is NULL");
        return ASGART_E_ARG;
    }
    *out = nullptr;
    if (!d_text || n <= 0 || sa_len != n) {
        set_error("asgart_index_create_device: need the text (and a suffix array of the same length, or NULL) on the device");
        return ASGART_E_ARG;
    }
    RC_TRY(check_device(device));
    asgart::Options opt;
    options_from_env(opt);
    const bool wide = (uint64_t)n >= 0xFFFFFF00ull || opt.force_wide != 0;
    if (sa_entry_bytes != (wide ? 8 : 4)) {
        set_error("asgart_index_create_device: a text of %lld bytes takes %d-byte suffix-array entries here (got %d)",
                  (long long)n, wide ? 8 : 4, sa_entry_bytes);
        return ASGART_E_ARG;
    }
    asgart_index *idx = new (std::nothrow) asgart_index();
    if (!idx) {
        set_error("out of host memory");
        return ASGART_E_OOM;
    }
    BlockCache::index_born();
    idx->device = device;
    idx->n = n;
    idx->n_sa = n;
    idx->opt = opt;
    idx->wide = wide;
    for (auto &cx : idx->ctx) memset(&cx.stats, 0, sizeof(cx.stats));
    int32_t rc = [&]() -> int32_t {
        for (auto &cx : idx->ctx) RC_TRY(create_ctx_streams(cx));
        hipStream_t s = idx->ctx[0].stream;
        const size_t slot = wide ? 8 : 4;
        HIP_TRY(dev_malloc((void **)&idx->d_text, (size_t)n + 64));
        HIP_TRY(dev_malloc(&idx->d_sa, ((size_t)n + 16) * slot));
        prealloc_ptab(idx);
        HIP_TRY(hipMemsetAsync(idx->d_text + n, 0, 64, s));
        HIP_TRY(hipMemcpyAsync(idx->d_text, d_text, (size_t)n, hipMemcpyDeviceToDevice, s));
        if (d_sa) HIP_TRY(hipMemcpyAsync(idx->d_sa, d_sa, (size_t)n * slot, hipMemcpyDeviceToDevice, s));
        bool dna = false;
        RC_TRY(text_is_dna(idx->d_text, n, s, &dna));
        if (!d_sa && dna)  // no suffix array given: sorted here (asgart_prepare_data hands over the text it normalised)
            RC_TRY(sa_build_device(idx->d_text, n, idx->d_sa, idx->wide, s, (uint64_t)idx->opt.test_wide_batch));
        const int64_t tl = n < (int64_t)kMaxK + 32 ? n : (int64_t)kMaxK + 32;
        idx->h_tail.resize((size_t)tl);
        HIP_TRY(read_back(idx->h_tail.data(), idx->d_text + (n - tl), (size_t)tl, s));
        HIP_TRY(stream_sync(s));
        if (!dna) {
            set_error("text contains bytes other than the normalised bases {A,C,G,T,N} and '$' "
                      "(reference src/bin/asgart.rs:289-301,430)");
            return ASGART_E_ARG;
        }
        return 0;
    }();
    if (rc != 0) {
        asgart_index_destroy(idx);
        return rc;
    }
    *out = idx;
    return 0;
}

int32_t asgart_index_set_option(asgart_index *idx, const char *name, int64_t value) {
    if (!idx) {
        set_error("index is NULL");
        return ASGART_E_ARG;
    }
    if (name && (!strcmp(name, "force_wide") || !strcmp(name, "ptab_depth"))) {
        set_error("option %s is fixed when the index is created (set ASGART_<NAME> in the environment before)", name);
        return ASGART_E_ARG;
    }
    if (name && !strcmp(name, "kfilter_bits")) REFUSE_POISONED(idx);  // (it frees device buffers)
    idx->acquire_all();  // never changes under a running call
    const int32_t rc = option_set(idx->opt, name, value);
    if (rc == 0 && !strcmp(name, "test_fail_alloc")) asgart::fail_alloc_countdown().store(value);  // (process-wide)
    if (rc == 0 && !strcmp(name, "kfilter_bits")) {  // rebuilt at the new size by the next call (0: searched without)
        (void)hipSetDevice(idx->device);
        free_filters(idx);
    }
    idx->release_all();
    return rc;
}

int64_t asgart_index_check_sa(asgart_index *idx) {
    if (!idx) {
        set_error("index is NULL");
        return ASGART_E_ARG;
    }
    if (idx->trimmed) {
        set_error("asgart_index_check_sa: not for a --trim index (its array covers a window of the text)");
        return ASGART_E_ARG;
    }
    REFUSE_POISONED(idx);
    HIP_TRY(hipSetDevice(idx->device));
    idx->acquire_all();
    struct Unlock {
        asgart_index *i;
        ~Unlock() { i->release_all(); }
    } unlock{idx};
    const uint64_t n = (uint64_t)idx->n;
    const size_t slot = idx->wide ? 8 : 4;
    DevBuf isa, errs;
    hipStream_t s = idx->ctx[0].stream;
    unsigned long long h_errs = 0;
    int32_t rc = [&]() -> int32_t {
        RC_TRY(isa.reserve(n * slot));
        RC_TRY(errs.reserve(8));
        HIP_TRY(hipMemsetAsync(isa.p, 0xFF, n * slot, s));
        HIP_TRY(hipMemsetAsync(errs.p, 0, 8, s));
        const unsigned g = grid_capped(n);
        if (idx->wide) {
            isa_scatter_kernel<uint64_t><<<g, 256, 0, s>>>((const uint64_t *)idx->d_sa, isa.as<uint64_t>(), n);
            sa_check_kernel<uint64_t><<<g, 256, 0, s>>>(idx->d_text, (const uint64_t *)idx->d_sa, isa.as<uint64_t>(), n,
                                                        errs.as<unsigned long long>());
        } else {
            isa_scatter_kernel<uint32_t><<<g, 256, 0, s>>>((const uint32_t *)idx->d_sa, isa.as<uint32_t>(), n);
            sa_check_kernel<uint32_t><<<g, 256, 0, s>>>(idx->d_text, (const uint32_t *)idx->d_sa, isa.as<uint32_t>(), n,
                                                        errs.as<unsigned long long>());
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(read_back(&h_errs, errs.p, 8, s));
        HIP_TRY(stream_sync(s));
        return 0;
    }();
    isa.release();
    errs.release();
    if (rc != 0) return rc;
    return (int64_t)h_errs;
}

int32_t asgart_index_prepare(asgart_index *idx, uint64_t probe_size) {
    if (!idx) {
        set_error("index is NULL");
        return ASGART_E_ARG;
    }
    return index_prepare(idx, probe_size);
}

static bool pattern_bytes_ok(const uint8_t *p, int64_t cnt) {
    for (int64_t j = 0; j < cnt; ++j) {
        uint8_t c = p[j];
        if (!(c == 'A' || c == 'C' || c == 'G' || c == 'T' || c == 'N')) return false;
    }
    return true;
}

static int32_t run_pattern_kernel(asgart_index *idx, const uint8_t *pats, int64_t n_pat,
                                  int64_t width, bool cache, uint64_t *lo, uint64_t *hi) {
    if (n_pat == 0) return 0;
    REFUSE_POISONED(idx);
    HIP_TRY(hipSetDevice(idx->device));
    idx->acquire_all();
    struct Unlock {
        asgart_index *i;
        ~Unlock() { i->release_all(); }
    } unlock{idx};
    Workspace &w = idx->ctx[0].ws;
    RC_TRY(w.pat.reserve((size_t)(n_pat * width)));
    RC_TRY(w.out_a.reserve((size_t)n_pat * 8));
    RC_TRY(w.out_b.reserve((size_t)n_pat * 8));
    hipStream_t s = idx->ctx[0].stream;
    HIP_TRY(hipMemcpyAsync(w.pat.p, pats, (size_t)(n_pat * width), hipMemcpyHostToDevice, s));
    unsigned g = grid_for((uint64_t)n_pat);
    if (idx->wide) {
        auto v = idx->view<uint64_t>();
        if (cache) cache_get_kernel<uint64_t><<<g, 256, 0, s>>>(v, w.pat.as<uint8_t>(), n_pat,
                                                                w.out_a.as<uint64_t>(),
                                                                w.out_b.as<uint64_t>());
        else pattern_search_kernel<uint64_t><<<g, 256, 0, s>>>(v, w.pat.as<uint8_t>(), n_pat,
                                                               w.out_a.as<uint64_t>(),
                                                               w.out_b.as<uint64_t>());
    } else {
        auto v = idx->view<uint32_t>();
        if (cache) cache_get_kernel<uint32_t><<<g, 256, 0, s>>>(v, w.pat.as<uint8_t>(), n_pat,
                                                                w.out_a.as<uint64_t>(),
                                                                w.out_b.as<uint64_t>());
        else pattern_search_kernel<uint32_t><<<g, 256, 0, s>>>(v, w.pat.as<uint8_t>(), n_pat,
                                                               w.out_a.as<uint64_t>(),
                                                               w.out_b.as<uint64_t>());
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(read_back(lo, w.out_a.p, (size_t)n_pat * 8, s));
    HIP_TRY(read_back(hi, w.out_b.p, (size_t)n_pat * 8, s));
    HIP_TRY(stream_sync(s));
    return 0;
}

int32_t asgart_searcher_cache_get(asgart_index *idx, const uint8_t *patterns8, int64_t n_pat,
                                  uint64_t *lo, uint64_t *hi) {
    if (!idx || n_pat < 0 || (n_pat && (!patterns8 || !lo || !hi))) {
        set_error("asgart_searcher_cache_get: bad argument");
        return ASGART_E_ARG;
    }
    if (!pattern_bytes_ok(patterns8, n_pat * kCacheLen)) {
        set_error("8-mer outside the alphabet {A,T,G,C,N} (the reference panics here, "
                  "src/searcher.rs:155-161)");
        return ASGART_E_ARG;
    }
    if (idx->k == 0) RC_TRY(index_prepare(idx, 20));
    return run_pattern_kernel(idx, patterns8, n_pat, kCacheLen, true, lo, hi);
}

int32_t asgart_searcher_search(asgart_index *idx, const uint8_t *patterns, int64_t n_pat,
                               uint64_t k, uint64_t *lo, uint64_t *hi) {
    if (!idx || n_pat < 0 || (n_pat && (!patterns || !lo || !hi))) {
        set_error("asgart_searcher_search: bad argument");
        return ASGART_E_ARG;
    }
    RC_TRY(index_prepare(idx, k));
    if (!pattern_bytes_ok(patterns, n_pat * (int64_t)k)) {
        set_error("pattern outside the alphabet {A,T,G,C,N} (the reference panics when the "
                  "first 8 bytes are, src/searcher.rs:155-161)");
        return ASGART_E_ARG;
    }
    return run_pattern_kernel(idx, patterns, n_pat, (int64_t)k, false, lo, hi);
}

int32_t asgart_sa_read(asgart_index *idx, uint64_t lo, uint64_t hi, int64_t *out) {
    if (!idx || hi < lo || hi > (uint64_t)idx->n_sa || (hi > lo && !out)) {
        set_error("asgart_sa_read: bad range");
        return ASGART_E_ARG;
    }
    if (hi == lo) return 0;
    REFUSE_POISONED(idx);
    HIP_TRY(hipSetDevice(idx->device));
    idx->acquire_all();
    struct Unlock {
        asgart_index *i;
        ~Unlock() { i->release_all(); }
    } unlock{idx};
    const uint64_t cnt = hi - lo;
    if (idx->wide) {
        HIP_TRY(read_back(out, (const uint64_t *)idx->d_sa + lo, cnt * 8, idx->ctx[0].stream));
    } else {
        RC_TRY(idx->ctx[0].ws.out_a.reserve(cnt * 8));
        widen_sa_kernel<<<grid_for(cnt), 256, 0, idx->ctx[0].stream>>>(
            (const uint32_t *)idx->d_sa + lo, idx->ctx[0].ws.out_a.as<int64_t>(), cnt);
        HIP_TRY(hipGetLastError());
        HIP_TRY(read_back(out, idx->ctx[0].ws.out_a.p, cnt * 8, idx->ctx[0].stream));
    }
    HIP_TRY(stream_sync(idx->ctx[0].stream));
    return 0;
}

}  // extern "C"
