// pipeline_dev.hpp -- device kernels of the probe -> search -> extend pipeline.
//
//   K1 probe_count_kernel   one thread per probe: key, SA interval, filtered count
//   K1b big_count_kernel    one wave per probe with a large interval (early exit)
//   K2 scan_{reduce,mid,down}  row offsets + quiet-run segmentation
//   K3 fill_{small,big}_kernel  filtered hits in SA order -> CSR
//   K4 extend_kernel        one wavefront per independent automaton segment
//
// Reference semantics: src/automaton.rs:57-216 (see DESIGN.md for the proof
// that a run of ceil(G/step) processed zero-hit probes empties the arm list,
// which is what makes segments independent).
#pragma once

#include "search_dev.hpp"

#include <type_traits>

namespace asgart {

constexpr int kTiers = 7;  // extension tiers (see the placement in pipeline.hip)

// device counters (u64 each)
enum Counter {
    CT_BIG = 0,       // entries in big_list
    CT_SEG,           // entries in seg_list
    CT_SCAN_TICKET,   // scan_segments_kernel: the next tile
    CT_FAM,           // families emitted
    CT_SD,            // ProtoSDs emitted
    CT_OVF,           // segments that overflowed the arm capacity
    CT_TOTAL_HITS,    // CSR size
    CT_N_SKIPPED,
    CT_CARD_SKIPPED,
    CT_WITH_HITS,
    CT_RAW_HITS,
    CT_SEARCHED,
    CT_BISECT,        // yardstick
    CT_OVF_CURSOR,
    CT_AMBIG,         // sharding: start decisions that need a longer look-back
    CT_RANOUT,        // sharding: segments that ran past the look-ahead window
    // 16..33 and 56..67: per-phase cycle sums of the diagnostic build (-DASGART_PROFILE_EXTEND)
    CT_EARLY_N = 34,    // early cascade launches (of tiers 3 and 6): list lengths ...
    CT_EARLY_CUR = 36,  // ... and work cursors
    CT_RANK = 38,       // entries in rank_list (large intervals counted by bisection of the position-sorted lists)
    CT_BIG0 = 39,       // entries of big_list that big_count_kernel counted (later ones were appended for the fill)
    CT_ALG_BYTES = 68,  // accounting pass: bytes the probe-search kernels move by design
    CT_FLT_REJECTED,    // accounting pass: probes answered by the presence filter alone
    CT_LONGSEG,         // placement: segments the lane-per-segment walk handed to the wave-per-segment kernel
    CT_ALG_BYTES16,     // accounting pass: the part of CT_ALG_BYTES that is wide coalesced loads (16 bytes per lane)
    CT_HIST_PEAK = 72,   // diagnostic build: log2 histograms per launch (16 bins each)
    CT_HIST_PROBES = 88,
    CT_N1 = 104,       // list lengths of the extension tiers 1..kTiers (kTiers entries)
    CT_NF = 111,       // ... of a cascade launch
    CT_CUR1 = 112,     // work cursors of the tiers (kTiers entries)
    CT_CURF = 119,
    CT_OVF1 = 120,     // segments tier t gave up on (kTiers entries; the last one has nowhere to go)
    CT_BUSY1 = 128,    // per tier: sum over its workgroups of their lifetime, in 10-ns ticks (how much of the chip a tier holds:
                       // persistent workgroups own their share of a compute unit from launch to exit) ...
    CT_WGS1 = 136,     // ... and the number of workgroups summed
    CT_TPROBES1 = 144, // placement statistics (option debug only): hit-probes per tier ...
    CT_THITS1 = 152,   // ... and hits per tier
    CT_SEGMAX1 = 160,  // per tier: the longest time one workgroup spent on ONE segment, in 10-ns ticks -- the serial floor of
                       // the extension (what neither more compute units nor more GPUs shorten)
    CT_CLUSTER_BARREN = 168,  // segments cluster_barren_kernel proved barren
    CT_CLUSTER_CUR = 169,     // its work cursors (two launches)
    CT_COUNT = 176
};

__device__ inline int chunk_of(const ChunkTable &ch, uint32_t g) {
    // last c with pbase[c] <= g  (pbase non-decreasing; empty chunks repeat values)
    int lo = 0, hi = ch.n_chunks;  // answer in [lo, hi)
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (ch.pbase[mid] <= g) lo = mid; else hi = mid;
    }
    return lo;
}

// same for a wave-uniform probe number: keeps the bisection in scalar registers / scalar loads
__device__ inline int chunk_of_uniform(const ChunkTable &ch, uint32_t g) {
    g = __builtin_amdgcn_readfirstlane(g);
    int lo = 0, hi = ch.n_chunks;
    while (hi - lo > 1) {
        const int mid = __builtin_amdgcn_readfirstlane((lo + hi) >> 1);
        if (ch.pbase[mid] <= g) lo = mid; else hi = mid;
    }
    return __builtin_amdgcn_readfirstlane(lo);
}

// hit filter of src/automaton.rs:105-114
__device__ inline bool keep_hit(uint64_t x, uint64_t i, uint64_t s, uint64_t L, bool reverse) {
    if (!reverse) return x > i + s;  // implies x != i
    return x != i && x >= s + L - i;
}

// Workgroup barrier for data exchanged through LDS only.  __syncthreads() also waits for the
// wave's outstanding GLOBAL stores (vmcnt(0)): one record written to HBM would stall every wave
// of the workgroup for a memory round trip at the next barrier.  The extension kernels never
// read back what they store to global memory, so their per-probe barriers only drain LDS traffic.
__device__ inline void lds_barrier() { __asm__ volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Wave-uniform values the compiler cannot prove uniform (read from LDS, or a lane of a vector):
// forcing them into scalar registers keeps the per-probe bookkeeping and branches on the scalar
// unit instead of exec-masked vector code and LDS permutes.
__device__ inline uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ inline unsigned long long uni(unsigned long long v) {
    return ((unsigned long long)uni((uint32_t)(v >> 32)) << 32) | uni((uint32_t)v);
}
__device__ inline uint32_t lane_of(uint32_t v, uint32_t l) {
    return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l);
}
__device__ inline unsigned long long lane_of(unsigned long long v, uint32_t l) {
    return ((unsigned long long)lane_of((uint32_t)(v >> 32), l) << 32) | lane_of((uint32_t)v, l);
}

// ---------------------------------------------------------------- K1 ---------
// Probe search: one thread per probe, one workgroup per 256 consecutive probes.
//
//   1. The workgroup's probes cover ONE contiguous text window (stride k/2, so 256 probes =
//      2570 bases at k = 20): it is staged into LDS with one coalesced 16-byte load per lane.
//   2. Every base is converted once: thread h packs the 3-bit codes of "half" h (k/2 bases) and
//      the key of probe t is half[t] ++ half[t+1] (plus one more base when k is odd).  -R walks
//      the window downwards, -C complements the codes: needle preparation (reference
//      src/bin/asgart.rs:206-218) without a needle.
//   3. Presence filter (search_dev.hpp): one load from a cache-resident bitmap; a rejected probe
//      has no hit and is done.
//   4. The others: prefix table -> bisection over the sorted keys -> filtered count of the
//      suffix-array interval (intervals > 32 go to the wave kernel through big_list).
//
// COUNT = true is the accounting pass behind bench.py's `kernel_algorithmic_bytes`: the same
// control flow, no stores, every load / store of the real kernel priced in bytes.
constexpr int kProbeBlock = 256;  // probes per workgroup (one tile)
constexpr int kProbeThreads = 64; // ... run by ONE wave: every lane stages and tests four of them, then the survivors (a fifth of
                                  // the tile on the GRCh38-shaped input: about one per lane) are looked up.  With four waves per tile
                                  // three of them held their slots through the staging only to retire; a wave slot now spends most
                                  // of its life in the dependent gathers of the lookup, and no barrier involves a second wave.
constexpr int kProbeSub = kProbeBlock / kProbeThreads;
constexpr int kMaxHalf = (kMaxKey + 1) / 2;                                 // bases per half (one-word keys)
constexpr int kWinBytes = ((kProbeBlock + 2) * kMaxHalf + 15 + 16 + 15) / 16 * 16;  // <= 256 lanes x 16 B

template <class SlotT, bool COUNT>
__global__ __launch_bounds__(kProbeThreads) void probe_count_kernel(IndexView<SlotT> ix, RunParams rp,
                                                                    SlotT *__restrict__ p_lo,
                                                                    uint32_t *__restrict__ p_raw,
                                                                    uint32_t *__restrict__ p_filt,
                                                                    uint32_t *__restrict__ big_list,
                                                                    uint32_t *__restrict__ rank_list,
                                                                    unsigned long long *__restrict__ ctr) {
    __shared__ __attribute__((aligned(16))) uint8_t s_text[kWinBytes];
    __shared__ uint32_t s_half[kProbeBlock + 2];
    // the filter's answers for the workgroup's probes: 256 probes at stride k/2 = one contiguous run of bits
    constexpr int kPbLoads = (kProbeBlock * kMaxHalf + 127) / 128 + 2;  // 16-byte loads that cover it
    __shared__ __attribute__((aligned(16))) uint32_t s_pb[kPbLoads * 4];
    __shared__ uint8_t s_surv[kProbeBlock];
    typename std::conditional<COUNT, CountBytes, NoBytes>::type cb;
    const uint32_t lane = threadIdx.x;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    uint32_t g_end;  // end of the tile's window
    const uint32_t gb = rp.tile_of(blockIdx.x, (uint32_t)kProbeBlock, g_end);  // first probe of the workgroup
    const uint32_t g_last = min(gb + (uint32_t)kProbeBlock, g_end) - 1u;
    const int k = rp.k, H = rp.step;
    // one chunk for the whole workgroup (all but a handful of workgroups): staged window
    const int c0 = chunk_of_uniform(rp.ch, gb);
    // (probes longer than one key word -- rare -- take the per-thread path as well)
    const bool uniform = rp.ch.pbase[c0 + 1] > g_last && k <= kMaxKey;
    uint32_t n_rej = 0;
    // one probe's lookup: SA interval, filtered count of a small interval (large ones are marked for the wave kernels)
    // lrn_: the pass's position bits when they are being learned (RunParams::learn), else null
    auto lookup = [&](uint32_t g_, uint64_t q_, uint64_t q2_, uint64_t i_, uint64_t s_, uint64_t L_, uint32_t md_,
                      const uint64_t *lrn_) {
        const bool reverse = (md_ & 2u) != 0u, complement = (md_ & 1u) != 0u;
        uint64_t lo, hi;
        ProbeRef pr;  // (read only by probes of more than 42 bases)
        pr.p = reverse ? ix.text + s_ + L_ - 1u - i_ : ix.text + s_ + i_;
        pr.dir = reverse ? -1 : 1;
        pr.comp = complement;
        const bool all_occurrences = kmer_range(ix, q_, q2_, pr, lo, hi, cb);
        const uint64_t raw = hi - lo;
        if (!COUNT) {
            p_lo[g_] = (SlotT)lo;
            p_raw[g_] = (uint32_t)raw;
        }
        cb.wr(sizeof(SlotT) + 4);
        if (raw <= (uint64_t)kSmallInterval) {
            uint32_t cnt = 0;
            // Direct pass: the needle is the text itself, so the probe's own position is one of
            // the occurrences; a single occurrence is that one, and the filter (x > i + s)
            // drops it -- no need to fetch the suffix-array entry.
            const bool only_self = all_occurrences && raw == 1 && md_ == 0u;
            bool any = false;  // an occurrence that could be kept, whatever chunk the probe came in (a reversed needle's
                               // `m.start != i` compares a text position with a chunk offset: left out)
            for (uint64_t r = lo; r < hi && !only_self; ++r) {
                cb.rd(sizeof(SlotT));
                const uint64_t x = ix.sa[r];
                cnt += keep_hit(x, i_, s_, L_, reverse) ? 1u : 0u;
                if (reverse) any |= x >= s_ + L_ - i_;
            }
            if (!reverse) any = cnt != 0u;
            if (!COUNT) p_filt[g_] = cnt > rp.C ? kSkipCard : cnt;
            cb.wr(4);
            if constexpr (!COUNT) {
                // learned position bits: no occurrence of this probe's k-mer can ever be kept at this text position
                if (lrn_ && all_occurrences && !any) {
                    const uint64_t p = reverse ? s_ + L_ - i_ - (uint64_t)k : s_ + i_;
                    atomicAnd(const_cast<unsigned long long *>(reinterpret_cast<const unsigned long long *>(lrn_)) + (p >> 6),
                              ~(1ull << (p & 63u)));
                }
            }
        } else {
            // a large interval is only MARKED here; collect_pending_kernel turns the marks into the two work lists (a
            // workgroup-aggregated append from this kernel cost every workgroup a returning atomic on one of two adjacent
            // counters and a barrier that its retired waves never reached).
            // a whole k-mer interval of some size: its kept count is a bisection of the position-sorted list
            if (!COUNT) p_filt[g_] = (ix.sap && all_occurrences && raw > (uint64_t)kRankMin) ? kPendingRank : kPending;
            cb.wr(4 + 4);  // + its work-list entry (written by the collecting pass)
            cb.rd(4);      //   ... which reads the mark back
        }
    };
    // a probe the presence filter (or its first base) answers: what is written for it; -> true: it has to be looked up
    auto screen = [&](uint32_t g_, uint32_t first_, bool pass_, uint32_t md_) {
        if (first_ == 4u) {  // needle[i] == 'N'  (automaton.rs:100-102)
            if (!COUNT) p_filt[g_] = kSkipN;
            cb.wr(4);
            return false;
        }
        if (!pass_) {
            // the position filter says: no hit is kept (the probe's interval is never looked up; asgart_get_stats does that
            // when the raw interval sizes are asked for)
            if (!COUNT) {
                p_raw[g_] = kRawUnknown;
                p_filt[g_] = 0u;
            }
            cb.wr(8);
            n_rej += 1;
            return false;
        }
        return true;
    };
    if (uniform) {
        const uint32_t pass0 = rp.pass_of(c0), md = rp.mode_of_pass(pass0);  // (the orientation of the chunk's pass)
        const bool reverse = (md & 2u) != 0u, complement = (md & 1u) != 0u;
        const uint64_t *const pbits = rp.pbits[pass0], *const flt = rp.flt[pass0];
        const uint64_t s = rp.ch.start[c0], L = rp.ch.len[c0];
        const uint64_t i0 = (uint64_t)(gb - rp.ch.pbase[c0] + 1) * (uint64_t)H;  // needle offset of probe gb
        const int n_half = kProbeBlock + 2;
        // text positions of half h, base j:  direct  b0 + h*H + j ;  reversed  e0 - h*H - j
        const long long b0 = (long long)(s + i0), e0 = (long long)(s + L - 1u - i0);
        const long long w_lo = reverse ? e0 - (long long)n_half * H + 1 : b0;
        const long long a_lo = w_lo & ~15ll;  // floor to 16 (two's complement: also for negatives)
        const long long w_hi = reverse ? e0 : b0 + (long long)n_half * H - 1;  // inclusive
        const uint32_t n_load = (uint32_t)((w_hi - a_lo) / 16 + 1);  // <= kWinBytes / 16
        for (uint32_t t = lane; t < n_load; t += kProbeThreads) {   // (three rounds, all in flight together)
            const long long a = a_lo + 16ll * t;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (a >= 0 && (uint64_t)a + 16u <= ix.n + 64u) {  // the text allocation has 64 spare bytes
                v = *reinterpret_cast<const uint4 *>(ix.text + a);
                cb.rd16();
            }
            *reinterpret_cast<uint4 *>(s_text + 16u * t) = v;
        }
        long long pb_lo = 0;  // first bit held by s_pb
        if (pbits) {
            // text positions the probes cover: direct  s + i0 + t H ;  reversed  s + L - i0 - k - t H
            const long long p_first = reverse ? (long long)(s + L - i0) - k - (long long)(kProbeBlock - 1) * H
                                                 : (long long)(s + i0);
            const long long byte_lo = (p_first >> 7) * 16;  // (floor: also for the negatives of the idle lanes)
            pb_lo = byte_lo * 8;
            const long long p_last = p_first + (long long)(kProbeBlock - 1) * H;
            const uint32_t n_pb = (uint32_t)((p_last >> 7) - (p_first >> 7) + 1);  // <= kPbLoads
            if (lane < n_pb) {
                const long long a = byte_lo + 16ll * lane;
                uint4 v = make_uint4(~0u, ~0u, ~0u, ~0u);
                if (a >= 0 && (uint64_t)a + 16u <= ((ix.n + 63u) / 64u) * 8u + 512u) {  // (the allocation is padded)
                    v = *reinterpret_cast<const uint4 *>(reinterpret_cast<const uint8_t *>(pbits) + a);
                    cb.rd16();
                }
                *reinterpret_cast<uint4 *>(&s_pb[4u * lane]) = v;
            }
        }
        __syncthreads();  // (one wave: the barrier orders its own LDS traffic)
        for (uint32_t h = lane; h < (uint32_t)n_half; h += kProbeThreads) {
            const long long p0 = reverse ? (e0 - a_lo) - (long long)h * H : (b0 - a_lo) + (long long)h * H;
            uint32_t v = 0;
            for (int j = 0; j < H; ++j) {
                uint32_t c = base_code(s_text[reverse ? p0 - j : p0 + j]);
                if (complement) c = comp_code(c);
                v = (v << 3) | c;
            }
            s_half[h] = v;
        }
        __syncthreads();
        auto key_of = [&](uint32_t t) {
            uint64_t qt = ((uint64_t)s_half[t] << (3 * H)) | (uint64_t)s_half[t + 1];
            if (k & 1) qt = (qt << 3) | (uint64_t)(s_half[t + 2] >> (3 * (H - 1)));
            return qt;
        };
        // ---- presence filter (every probe), then the lookup of the survivors ---------------------------------
        // About four probes in five are answered by the filter.  The lookup behind it -- prefix table, key bisection,
        // suffix-array entries -- is a chain of dependent random gathers, and a wave runs it at the speed of its
        // slowest lane however few lanes are left: the survivors of the tile are compacted (ballots) so that the
        // lookups run with the lanes full.
        uint32_t n_surv = 0;
#pragma unroll
        for (int u = 0; u < kProbeSub; ++u) {
            const uint32_t t = lane + (uint32_t)u * kProbeThreads, g = gb + t;
            bool survivor = false;
            if (g < g_end) {
                const uint64_t q = key_of(t);
                const uint32_t first = (uint32_t)(q >> (3 * (k - 1))) & 7u;
                bool pass = true;
                if (first != 4u) {
                    const uint64_t i = i0 + (uint64_t)t * (uint64_t)H;
                    if (pbits) {  // the filter's answer, by the text position the probe covers
                        const long long p = reverse ? (long long)(s + L - i) - k : (long long)(s + i);
                        const uint32_t b = (uint32_t)(p - pb_lo);
                        pass = (s_pb[b >> 5] >> (b & 31u)) & 1u;
                    } else if (flt && !is_tail_corner(ix, q)) {
                        cb.rd(8);
                        pass = filter_test(flt, rp.flt_bits, q);
                    }
                }
                survivor = screen(g, first, pass, md);
            }
            const unsigned long long sm = __ballot(survivor);
            if (survivor) s_surv[n_surv + (uint32_t)__popcll(sm & lt_mask)] = (uint8_t)t;
            n_surv += (uint32_t)__popcll(sm);
        }
        __syncthreads();
        for (uint32_t j = lane; j < n_surv; j += kProbeThreads) {
            const uint32_t t = s_surv[j];
            lookup(gb + t, key_of(t), 0ull, i0 + (uint64_t)t * (uint64_t)H, s, L, md, ((rp.learn >> pass0) & 1u) ? pbits : nullptr);
        }
    } else {
        // the tile straddles a chunk boundary (or the probes are longer than one key word): every probe on its own
#pragma unroll 1
        for (int u = 0; u < kProbeSub; ++u) {
            const uint32_t g = gb + lane + (uint32_t)u * kProbeThreads;
            if (g >= g_end) continue;
            const int c = chunk_of(rp.ch, g);
            const uint32_t pass_c = rp.pass_of(c), md = rp.mode_of_pass(pass_c);
            const bool reverse = (md & 2u) != 0u, complement = (md & 1u) != 0u;
            const uint64_t *const pbits = rp.pbits[pass_c], *const flt = rp.flt[pass_c];
            const uint64_t s = rp.ch.start[c], L = rp.ch.len[c];
            const uint64_t i = (uint64_t)(g - rp.ch.pbase[c] + 1) * (uint64_t)H;
            uint32_t first = 0;
            uint64_t q2 = 0;
            const uint64_t q = probe_key(ix.text, s, L, i, k, reverse, complement, &first, &q2);
            cb.rd((uint32_t)k);
            bool pass = true;
            if (first != 4u) {
                if (pbits) {
                    const long long p = reverse ? (long long)(s + L - i) - k : (long long)(s + i);
                    cb.rd(8);
                    pass = (pbits[(uint64_t)p >> 6] >> ((uint64_t)p & 63u)) & 1ull;
                } else if (flt && !is_tail_corner(ix, q)) {
                    cb.rd(8);
                    pass = filter_test(flt, rp.flt_bits, q);
                }
            }
            if (screen(g, first, pass, md)) lookup(g, q, q2, i, s, L, md, ((rp.learn >> pass_c) & 1u) ? pbits : nullptr);
        }
    }
    if constexpr (COUNT) {
        unsigned long long b = cb.n, r = n_rej, b16 = cb.n16;
        for (int off = 32; off > 0; off >>= 1) {
            b += __shfl_down(b, off);
            r += __shfl_down(r, off);
            b16 += __shfl_down(b16, off);
        }
        if (lane == 0) {
            atomicAdd(&ctr[CT_ALG_BYTES], b);
            if (r) atomicAdd(&ctr[CT_FLT_REJECTED], r);
            if (b16) atomicAdd(&ctr[CT_ALG_BYTES16], b16);
        }
    }
}

// The probes probe_count_kernel marked as large intervals -> the two work lists (big_list: counted by streaming the
// interval; rank_list: by bisection of its position-sorted list).  Persistent workgroups over contiguous spans of the
// probe sequence: the marks of a tile are compacted by wave ballots into two LDS buffers, which are appended to the
// lists with ONE global atomic per 2 K entries instead of one (or two) per 256 probes.  Streams 4 bytes per probe.
constexpr int kCollectBlock = 256, kCollectItems = 8, kCollectTile = kCollectBlock * kCollectItems;
constexpr int kCollectCap = 2 * kCollectTile;
__global__ __launch_bounds__(kCollectBlock) void collect_pending_kernel(RunParams rp, const uint32_t *__restrict__ p_filt,
                                                                        uint32_t *__restrict__ big_list,
                                                                        uint32_t *__restrict__ rank_list,
                                                                        unsigned long long *__restrict__ ctr) {
    __shared__ uint32_t s_buf[2][kCollectCap];
    __shared__ uint32_t s_cnt[2];
    __shared__ unsigned long long s_base[2];
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    if (tid < 2) s_cnt[tid] = 0;
    __syncthreads();
    const uint32_t n_tiles = rp.n_tiles((uint32_t)kCollectTile);
    // a contiguous run of tiles per workgroup (the lists then follow the probe order in long stretches)
    const uint32_t t0 = (uint32_t)((uint64_t)n_tiles * blockIdx.x / gridDim.x);
    const uint32_t t1 = (uint32_t)((uint64_t)n_tiles * (blockIdx.x + 1u) / gridDim.x);
    auto flush = [&](int which, uint32_t *__restrict__ list, int counter) {  // (workgroup-uniform)
        const uint32_t n = s_cnt[which];
        if (!n) return;
        if (tid == 0) s_base[which] = atomicAdd(&ctr[counter], (unsigned long long)n);
        __syncthreads();
        const unsigned long long b = s_base[which];
        for (uint32_t j = tid; j < n; j += kCollectBlock) list[b + j] = s_buf[which][j];
        __syncthreads();
        if (tid == 0) s_cnt[which] = 0;
        __syncthreads();
    };
    for (uint32_t t = t0; t < t1; ++t) {
        uint32_t g_end;
        const uint32_t g0 = rp.tile_of(t, (uint32_t)kCollectTile, g_end);
#pragma unroll
        for (int a = 0; a < kCollectItems; ++a) {
            const uint32_t g = g0 + (uint32_t)a * kCollectBlock + tid;  // coalesced
            const uint32_t f = g < g_end ? p_filt[g] : 0u;
            const bool big = f == kPending, rank = f == kPendingRank;
            const unsigned long long mb = __ballot(big), mr = __ballot(rank);
            if (mb) {
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(&s_cnt[0], (uint32_t)__popcll(mb));
                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                if (big) s_buf[0][base + (uint32_t)__popcll(mb & lt_mask)] = g;
            }
            if (mr) {
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(&s_cnt[1], (uint32_t)__popcll(mr));
                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                if (rank) s_buf[1][base + (uint32_t)__popcll(mr & lt_mask)] = g;
            }
        }
        __syncthreads();
        // (the decision is taken from a snapshot every thread has read BEFORE anyone may add the next tile's entries:
        // a wave that ran ahead into the next tile's atomics could otherwise push a slower wave's reading over the
        // threshold and send only that wave into flush() and its barriers)
        const uint32_t c0 = s_cnt[0], c1 = s_cnt[1];
        __syncthreads();
        if (c0 > (uint32_t)(kCollectCap - kCollectTile)) flush(0, big_list, CT_BIG);
        if (c1 > (uint32_t)(kCollectCap - kCollectTile)) flush(1, rank_list, CT_RANK);
    }
    flush(0, big_list, CT_BIG);
    flush(1, rank_list, CT_RANK);
}

// Large whole-k-mer intervals when the index holds the position-sorted occurrence lists: the hit filter keeps the
// occurrences beyond a position threshold (src/automaton.rs:105-114), so the kept count is one bisection of
// sap[lo..hi) -- a dozen gathers instead of the ~1000 suffix-array entries a probe of a frequent k-mer streams
// before the early exit at max_cardinality + 1.  One thread per interval.  Probes that stay below the cardinality
// limit are appended to big_list, from which fill_big_kernel materialises their hits (in suffix-array order).
template <class SlotT, bool COUNT>
__global__ __launch_bounds__(256) void rank_count_kernel(IndexView<SlotT> ix, RunParams rp,
                                                         const SlotT *__restrict__ p_lo,
                                                         const uint32_t *__restrict__ p_raw,
                                                         uint32_t *__restrict__ p_filt,
                                                         const uint32_t *__restrict__ rank_list,
                                                         uint32_t *__restrict__ big_list,
                                                         unsigned long long *__restrict__ ctr) {
    const uint64_t n_rank = ctr[CT_RANK];
    const uint32_t lane = threadIdx.x & 63u;
    unsigned long long bytes = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t e0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) & ~63ull; e0 < n_rank; e0 += stride) {
        const uint64_t e = e0 + lane;
        bool refill = false;
        uint32_t g = 0;
        if (e < n_rank) {
            g = rank_list[e];
            const int c = chunk_of(rp.ch, g);
            const uint64_t s = rp.ch.start[c], L = rp.ch.len[c];
            const uint64_t i = (uint64_t)(g - rp.ch.pbase[c] + 1) * (uint64_t)rp.step;
            const SlotT *__restrict__ a = ix.sap + p_lo[g];
            const uint64_t R = p_raw[g];
            bytes += 4 + sizeof(SlotT) + 4 + 4;  // list entry, interval, final count
            // first index whose position is >= t
            auto first_ge = [&](uint64_t t) {
                uint64_t lo = 0, hi = R;
                while (lo < hi) {
                    const uint64_t mid = lo + ((hi - lo) >> 1);
                    bytes += sizeof(SlotT);
                    if ((uint64_t)a[mid] < t) lo = mid + 1; else hi = mid;
                }
                return lo;
            };
            uint64_t cnt;
            if (!(rp.mode_of(c) & 2u)) {
                cnt = R - first_ge(i + s + 1u);  // keep_hit: x > i + s
            } else {
                const uint64_t t = s + L - i;  // keep_hit: x != i && x >= s + L - i
                cnt = R - first_ge(t);
                if (i >= t) {
                    const uint64_t j = first_ge(i);
                    if (j < R && (uint64_t)a[j] == i) --cnt;
                }
            }
            const uint32_t f = cnt > (uint64_t)rp.C ? kSkipCard : (uint32_t)cnt;
            if (!COUNT) p_filt[g] = f;
            refill = f != 0u && f < kPending;
        }
        if (!COUNT) {
            const unsigned long long m = __ballot(refill);
            if (m) {
                const int leader = __ffsll((long long)m) - 1;
                unsigned long long at = 0;
                if ((int)lane == leader) at = atomicAdd(&ctr[CT_BIG], (unsigned long long)__popcll(m));
                at = __shfl(at, leader);
                if (refill) big_list[at + __popcll(m & ((1ull << lane) - 1ull))] = g;
            }
        } else {
            bytes += refill ? 4u : 0u;
        }
    }
    if (COUNT) {
        for (int off = 32; off > 0; off >>= 1) bytes += __shfl_down(bytes, off);
        if (lane == 0 && bytes) atomicAdd(&ctr[CT_ALG_BYTES], bytes);
    }
}

// Large intervals, one wave per 64 of them: the per-probe set-up (work-list entry, chunk, interval) is a
// chain of dependent global loads, so the 64 lanes each set up one probe at the same time; the wave then
// counts the intervals one after the other -- 256 suffix-array entries per round trip, early exit at
// max_cardinality + 1 -- with the first round of the NEXT interval already in flight.
template <class SlotT, bool COUNT>
__global__ __launch_bounds__(256) void big_count_kernel(IndexView<SlotT> ix, RunParams rp,
                                                        const SlotT *__restrict__ p_lo,
                                                        const uint32_t *__restrict__ p_raw,
                                                        uint32_t *__restrict__ p_filt,
                                                        const uint32_t *__restrict__ big_list,
                                                        unsigned long long *__restrict__ ctr) {
    const uint32_t lane = threadIdx.x & 63u;
    // (the accounting pass runs after the call: the list has grown by what rank_count_kernel appended for the fill)
    const uint64_t n_big = COUNT ? ctr[CT_BIG0] : ctr[CT_BIG];
    if (!COUNT && blockIdx.x == 0 && threadIdx.x == 0) ctr[CT_BIG0] = n_big;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    unsigned long long bytes = 0;
    for (uint64_t e0 = wave * 64u; e0 < n_big; e0 += n_waves * 64u) {
        const uint32_t np = (uint32_t)min((uint64_t)64, n_big - e0);
        // lane l sets up probe e0 + l
        uint32_t g = 0, md = 0;
        unsigned long long s = 0, L = 0, i = 0, lo = 0, hi = 0;
        if (lane < np) {
            g = big_list[e0 + lane];
            const int c = chunk_of(rp.ch, g);
            md = rp.mode_of(c);
            s = rp.ch.start[c];
            L = rp.ch.len[c];
            i = (unsigned long long)(g - rp.ch.pbase[c] + 1) * (unsigned long long)rp.step;
            lo = p_lo[g];
            hi = lo + p_raw[g];
        }
        bytes += (unsigned long long)np * (4 + sizeof(SlotT) + 4 + 4);  // list entry, interval, final count
        uint32_t my_cnt = 0;
        // first round of probe 0
        SlotT xn[4];
        {
            const unsigned long long l0 = lane_of(lo, 0u), h0 = lane_of(hi, 0u);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const unsigned long long r = l0 + 64u * u + lane;
                xn[u] = r < h0 ? ix.sa[r] : (SlotT)0;
            }
        }
        for (uint32_t p = 0; p < np; ++p) {
            const unsigned long long lo_p = lane_of(lo, p), hi_p = lane_of(hi, p), i_p = lane_of(i, p),
                                     s_p = lane_of(s, p), L_p = lane_of(L, p);
            const bool rev_p = (lane_of(md, p) & 2u) != 0u;
            SlotT x[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) x[u] = xn[u];
            if (p + 1 < np) {  // the next interval's first 256 entries: in flight while this one is counted
                const unsigned long long l1 = lane_of(lo, p + 1u), h1 = lane_of(hi, p + 1u);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const unsigned long long r = l1 + 64u * u + lane;
                    xn[u] = r < h1 ? ix.sa[r] : (SlotT)0;
                }
            }
            unsigned long long cnt = 0;
            for (unsigned long long base = lo_p;;) {
                bytes += sizeof(SlotT) * (hi_p - base < 256u ? hi_p - base : 256u);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const unsigned long long r = base + 64u * u + lane;
                    const bool keep = r < hi_p && keep_hit(x[u], i_p, s_p, L_p, rev_p);
                    cnt += __popcll(__ballot(keep));
                }
                base += 256;
                if (base >= hi_p || cnt > rp.C) break;
#pragma unroll
                for (int u = 0; u < 4; ++u) {  // (the early exit makes the rounds dependent; requesting the next
                                               // round ahead of the count was measured: 7.0 -> 8.2 ms, more bytes)
                    const unsigned long long r = base + 64u * u + lane;
                    x[u] = r < hi_p ? ix.sa[r] : (SlotT)0;
                }
            }
            if (lane == p) my_cnt = cnt > rp.C ? kSkipCard : (uint32_t)cnt;
        }
        if (!COUNT && lane < np) p_filt[g] = my_cnt;
    }
    if (COUNT && lane == 0 && bytes) atomicAdd(&ctr[CT_ALG_BYTES], bytes);
}

// ---------------------------------------------------------------- K2 ---------
// Row offsets and segmentation: ONE pass over the per-probe hit counts (scan_segments_kernel).
//
// What is scanned, per probe, is a small monoid: the running CSR offset (hits) and "quiet processed probes since the last
// hit-probe, reset at chunk starts" (c, flags) -- a segment starts at a hit-probe preceded by t* quiet probes or by its
// chunk's start.  A window that starts mid-chunk (sharded calls) begins with a reset of its own, marked
// "unknown": what the automaton held in front of it is not known, and stays unknown until the first hit-probe or real
// chunk start behind it.
// Single pass with decoupled look-back: persistent workgroups take tiles of kScanTile probes in ticket order; a tile's
// counts are loaded once (16 loads per lane in flight together), its aggregate is published, the exclusive prefix is
// assembled from the predecessors' published aggregates / prefixes, and the tile is walked again -- row offsets written,
// segment starts decided.  A wave owns 1024 consecutive probes as 16 rounds of 64, and everything a round needs about
// its neighbours comes out of three ballots (hit, quiet, reset) by bit arithmetic.  The round-5 version (reduce, mid, down: three kernels, every tile staged through LDS and
// walked item by item per thread) streamed 1.2 TB/s; this one is bound by its 12 bytes per probe.
struct ScanEl {
    unsigned long long hits;
    uint32_t c;      // processed probes after the last hit-probe (or all, if none)
    uint32_t flags;  // bit0 = contains a hit-probe, bit1 = contains a reset (chunk start / window start), bit2 = the LAST
                     // reset is a window start in mid-chunk (the combine below hands it on with the reset it belongs to)
};
// "no hit-probe since a window started in mid-chunk": c is only a lower bound of the quiet run
__device__ inline bool scan_unknown(const ScanEl &e) { return (e.flags & 5u) == 4u; }

__device__ inline ScanEl scan_identity() { return ScanEl{0ull, 0u, 0u}; }

__device__ inline ScanEl scan_combine(const ScanEl &a, const ScanEl &b) {
    ScanEl r;
    r.hits = a.hits + b.hits;
    if (b.flags & 2u) {
        r.c = b.c;
        r.flags = b.flags;
    } else if (b.flags & 1u) {
        r.c = b.c;
        r.flags = a.flags | 1u;
    } else {
        r.c = a.c + b.c;
        r.flags = a.flags;
    }
    return r;
}

// inclusive prefix sum across the 64 lanes of a wave (gfx9 DPP: row shifts + row broadcasts)
__device__ inline uint32_t wave_incl_scan(uint32_t x) {
#define ASGART_DPP_ADD(ctrl, rows) \
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, ctrl, rows, 0xf, false)
    ASGART_DPP_ADD(0x111, 0xf);  // row_shr:1
    ASGART_DPP_ADD(0x112, 0xf);  // row_shr:2
    ASGART_DPP_ADD(0x114, 0xf);  // row_shr:4
    ASGART_DPP_ADD(0x118, 0xf);  // row_shr:8
    ASGART_DPP_ADD(0x142, 0xa);  // row_bcast:15 -> rows 1, 3
    ASGART_DPP_ADD(0x143, 0xc);  // row_bcast:31 -> rows 2, 3
#undef ASGART_DPP_ADD
    return x;
}

// inclusive running maximum across the 64 lanes of a wave (values >= 0: lanes without a source contribute 0)
__device__ inline uint32_t wave_incl_max_scan(uint32_t x) {
#define ASGART_DPP_MAX(ctrl, rows) \
    x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, ctrl, rows, 0xf, false))
    ASGART_DPP_MAX(0x111, 0xf);  // row_shr:1
    ASGART_DPP_MAX(0x112, 0xf);  // row_shr:2
    ASGART_DPP_MAX(0x114, 0xf);  // row_shr:4
    ASGART_DPP_MAX(0x118, 0xf);  // row_shr:8
    ASGART_DPP_MAX(0x142, 0xa);  // row_bcast:15 -> rows 1, 3
    ASGART_DPP_MAX(0x143, 0xc);  // row_bcast:31 -> rows 2, 3
#undef ASGART_DPP_MAX
    return x;
}

constexpr int kScanBlock = 512;
constexpr int kScanRounds = 16;                                  // rounds of 64 probes per wave and tile
constexpr int kScanWaveSpan = 64 * kScanRounds;                  // probes a wave owns in a tile
constexpr int kScanTile = kScanWaveSpan * (kScanBlock / 64);     // 8192 probes
constexpr int kStartCap = kScanTile + kScanTile / 4;             // segment starts buffered per workgroup (a tile adds <= kScanTile)

// Tile descriptors of the look-back: two 64-bit words per tile, each carrying the descriptor's status in its top two bits
// (0 nothing, 1 the tile's own aggregate, 2 its inclusive prefix), written and read with relaxed device-scope atomics -- a
// reader that finds the two words in different states (a prefix overwriting an aggregate) reads again.
//   w0 = status << 62 | flags << 32 | c         w1 = status << 62 | hits (< 2^62)
constexpr unsigned long long kScanAgg = 1ull, kScanPre = 2ull;
__device__ inline void scan_publish(unsigned long long *desc, unsigned long long tile, unsigned long long status, const ScanEl &e) {
    __hip_atomic_store(&desc[2 * tile], status << 62 | (unsigned long long)e.flags << 32 | e.c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&desc[2 * tile + 1], status << 62 | e.hits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// -> status (0: not there yet, or caught between two states)
__device__ inline uint32_t scan_peek(const unsigned long long *desc, unsigned long long tile, ScanEl &e) {
    const unsigned long long w0 = __hip_atomic_load(&desc[2 * tile], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long w1 = __hip_atomic_load(&desc[2 * tile + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((w0 >> 62) != (w1 >> 62)) return 0u;
    e.hits = w1 & ((1ull << 62) - 1ull);
    e.c = (uint32_t)w0;
    e.flags = (uint32_t)(w0 >> 32) & 7u;
    return (uint32_t)(w0 >> 62);
}

__device__ inline ScanEl shfl_el(const ScanEl &e, int src) {
    ScanEl r;
    r.hits = __shfl(e.hits, src);
    r.c = __shfl(e.c, src);
    r.flags = __shfl(e.flags, src);
    return r;
}
__device__ inline ScanEl shfl_up_el(const ScanEl &e, int d) {
    ScanEl r;
    r.hits = __shfl_up(e.hits, d);
    r.c = __shfl_up(e.c, d);
    r.flags = __shfl_up(e.flags, d);
    return r;
}

// masks of the lanes above / from a lane on (l in 0..63; l = -1: every lane)
__device__ inline unsigned long long lanes_above(int l) { return l < 0 ? ~0ull : (l >= 63 ? 0ull : ~((2ull << l) - 1ull)); }
__device__ inline unsigned long long lanes_from(int l) { return l <= 0 ? ~0ull : ~((1ull << l) - 1ull); }

// The start decision of one hit-probe in a round that holds a chunk start (or the reset of a window that starts in
// mid-chunk): rare, and kept out of line -- the rounds of a tile are unrolled (their counts sit in registers).
__device__ __noinline__ void scan_starts_with_resets(bool hit, unsigned long long hm, unsigned long long qm, unsigned long long rm,
                                                     bool syn, uint32_t run_c, uint32_t run_flags, uint32_t tstar, bool &start,
                                                     bool &amb) {
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long lt_mask = (1ull << lane) - 1ull, le_mask = lt_mask | (1ull << lane);
    start = amb = false;
    if (!hit) return;
    const unsigned long long rb = rm & le_mask;
    const int lr = rb ? 63 - __clzll((long long)rb) : -1;         // the last reset at or in front of this probe
    const unsigned long long hb = hm & lt_mask & lanes_from(lr);  // hit-probes in front of it, not before that reset
    const int lh = hb ? 63 - __clzll((long long)hb) : -1;
    bool has_before, unknown;
    uint32_t quiet;
    if (lh >= 0) {
        has_before = true;
        unknown = false;
        quiet = (uint32_t)__popcll(qm & lt_mask & lanes_above(lh));
    } else if (lr >= 0) {
        has_before = false;
        unknown = syn && lr == 0 && lane != 0u;  // (behind the window's own reset, no hit-probe since)
        quiet = (uint32_t)__popcll(qm & lt_mask & lanes_from(lr));
        if (unknown) has_before = true;  // (it may have one: decided by the quiet run, or not at all)
    } else {
        has_before = (run_flags & 1u) != 0u;
        unknown = (run_flags & 5u) == 4u;
        quiet = run_c + (uint32_t)__popcll(qm & lt_mask);
        if (unknown) has_before = true;
    }
    amb = unknown && quiet < tstar;
    start = !has_before || quiet >= tstar;
}

__global__ __launch_bounds__(kScanBlock) void scan_segments_kernel(RunParams rp, const uint32_t *__restrict__ p_filt,
                                                                   unsigned long long *__restrict__ desc, uint32_t n_tiles,
                                                                   unsigned long long *__restrict__ row_off,
                                                                   uint32_t *__restrict__ seg_list,
                                                                   unsigned long long *__restrict__ ctr) {
    __shared__ uint32_t s_f[kScanTile];         // the tile's counts (every wave its own span: both passes read them from here)
    __shared__ ScanEl s_wave[kScanBlock / 64];  // the waves' aggregates of the tile
    __shared__ ScanEl s_excl;                   // the tile's exclusive prefix
    __shared__ uint32_t s_tile;
    __shared__ uint32_t s_start[kStartCap];
    __shared__ uint32_t s_nstart;
    __shared__ unsigned long long s_gbase;
    __shared__ unsigned long long sh_stat[4];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    if (tid < 4) sh_stat[tid] = 0;
    if (tid == 0) s_nstart = 0;
    const bool small_counts = rp.C < (1u << 25);
    unsigned long long st_n = 0, st_card = 0, st_hit = 0, st_valid = 0;  // (wave-uniform: popcounts of ballots)
    // (block-uniform) append the collected segment starts: order is irrelevant, families are sorted by (start probe,
    // ordinal) afterwards
    auto flush_starts = [&]() {
        const uint32_t ns = s_nstart;
        if (ns) {
            if (tid == 0) s_gbase = atomicAdd(&ctr[CT_SEG], (unsigned long long)ns);
            __syncthreads();
            const unsigned long long gb = s_gbase;
            for (uint32_t idx = tid; idx < ns; idx += kScanBlock) seg_list[gb + idx] = s_start[idx];
            __syncthreads();
            if (tid == 0) s_nstart = 0;
            __syncthreads();
        }
    };
    __syncthreads();
    for (;;) {
        if (tid == 0) s_tile = (uint32_t)atomicAdd(&ctr[CT_SCAN_TICKET], 1ull);  // tiles in ticket order: every predecessor
        __syncthreads();                                                          // of a tile belongs to a running workgroup
        const uint32_t tile = s_tile;
        if (tile >= n_tiles) break;
        uint32_t g_end;
        const uint32_t tile_g0 = rp.tile_of(tile, (uint32_t)kScanTile, g_end);
        const uint32_t wg0 = tile_g0 + wave * (uint32_t)kScanWaveSpan;  // this wave's first probe
        const bool syn_first = rp.init_unknown && tile_g0 + rp.win_len == g_end && wave == 0u;  // (a window that starts mid-chunk)
        // ---- the tile's counts: 16 coalesced loads per lane, all in flight together ----------------------------------
        {
            uint32_t f[kScanRounds];
#pragma unroll
            for (int r = 0; r < kScanRounds; ++r) {
                const uint32_t g = wg0 + (uint32_t)r * 64u + lane;
                f[r] = g < g_end ? p_filt[g] : kSkipN;  // (behind the window's end: nothing)
            }
#pragma unroll
            for (int r = 0; r < kScanRounds; ++r) s_f[wave * (uint32_t)kScanWaveSpan + (uint32_t)r * 64u + lane] = f[r];
        }
        // the chunk starts of this wave's span, round by round: a scalar cursor over the (few) chunks
        int c_first = chunk_of_uniform(rp.ch, wg0);
        if (rp.ch.pbase[c_first] < wg0 || rp.ch.pbase[c_first + 1] == rp.ch.pbase[c_first]) {  // next non-empty chunk that starts at or behind wg0
            ++c_first;
            while (c_first < rp.ch.n_chunks && rp.ch.pbase[c_first + 1] == rp.ch.pbase[c_first]) ++c_first;
        }
        const uint32_t pb_first = c_first < rp.ch.n_chunks ? rp.ch.pbase[c_first] : 0xFFFFFFFFu;
        // first probes of chunks in [ga, ga + 64): cn / pb = the next chunk that has not started yet and its first probe
        auto resets_of = [&](int &cn, uint32_t &pb, uint32_t ga) -> unsigned long long {
            unsigned long long m = 0;
            while (pb < ga + 64u && pb < g_end) {  // (pb = ~0u once the table is exhausted)
                const uint32_t pb_next = rp.ch.pbase[cn + 1];             // (the table has n_chunks + 1 entries)
                if (pb_next > pb) m |= 1ull << (pb - ga);                 // (only non-empty chunks have a first probe)
                ++cn;
                pb = cn < rp.ch.n_chunks ? pb_next : 0xFFFFFFFFu;
            }
            return m;
        };
        // one round's element: what it adds to a running (c, flags)
        auto round_el = [&](unsigned long long hm, unsigned long long qm, unsigned long long rm, bool syn) -> ScanEl {
            ScanEl e{0ull, 0u, 0u};
            const int lh = hm ? 63 - __clzll((long long)hm) : -1, lr = rm ? 63 - __clzll((long long)rm) : -1;
            if (lr >= 0) {
                const bool hit_since = lh >= lr;
                e.c = (uint32_t)__popcll(qm & (hit_since ? lanes_above(lh) : lanes_from(lr)));
                e.flags = 2u | (hit_since ? 1u : 0u) | ((syn && lr == 0) ? 4u : 0u);
            } else if (lh >= 0) {
                e.c = (uint32_t)__popcll(qm & lanes_above(lh));
                e.flags = 1u;
            } else {
                e.c = (uint32_t)__popcll(qm);
            }
            return e;
        };
        // ---- pass 1: the wave's aggregate -------------------------------------------------------------------------------
        ScanEl agg = scan_identity();
        unsigned long long vsum = 0;
        {
            int cn = c_first;
            uint32_t pb = pb_first;
#pragma unroll 1
            for (int r = 0; r < kScanRounds; ++r) {
                const uint32_t fr = s_f[wave * (uint32_t)kScanWaveSpan + (uint32_t)r * 64u + lane];
                const bool hit = fr >= 1u && fr < kPending;
                const unsigned long long hm = __ballot(hit), qm = __ballot(fr == 0u);
                unsigned long long rm = resets_of(cn, pb, wg0 + (uint32_t)r * 64u);
                const bool syn = syn_first && r == 0;
                if (syn) rm |= 1ull;
                vsum += hit ? fr : 0u;
                const ScanEl e = round_el(hm, qm, rm, syn);
                agg = scan_combine(agg, e);
            }
            for (int off = 32; off > 0; off >>= 1) vsum += __shfl_down(vsum, off);
            agg.hits = __shfl(vsum, 0);
        }
        if (lane == 0) s_wave[wave] = agg;
        __syncthreads();
        // ---- the tile's aggregate is published, its exclusive prefix assembled from its predecessors' (wave 0) --------
        if (wave == 0) {
            ScanEl tile_agg = scan_identity();
            for (int w = 0; w < kScanBlock / 64; ++w) tile_agg = scan_combine(tile_agg, s_wave[w]);
            ScanEl excl = scan_identity();
            if (tile == 0) {
                if (lane == 0) scan_publish(desc, 0, kScanPre, tile_agg);
            } else {
                if (lane == 0) scan_publish(desc, tile, kScanAgg, tile_agg);
                // windows of 64 predecessors, nearest first: lane l looks at tile `hi - 64 + l` (lane 63 = the nearest)
                ScanEl acc = scan_identity();  // combination of the tiles [hi, tile)
                long long hi = tile;
                for (;;) {
                    const long long t_l = hi - 64 + (long long)lane;
                    ScanEl e = scan_identity();
                    uint32_t stt = 3u;  // (no such tile: identity, never waited for)
                    if (t_l >= 0) stt = scan_peek(desc, (unsigned long long)t_l, e);
                    const unsigned long long pm = __ballot(stt == (uint32_t)kScanPre), zm = __ballot(stt == 0u);
                    const int pl = pm ? 63 - __clzll((long long)pm) : -1;  // the nearest tile whose inclusive prefix is known
                    if (zm & lanes_above(pl)) {  // a tile between it and us has not published yet
                        __builtin_amdgcn_s_sleep(2);
                        continue;
                    }
                    if ((int)lane < pl || stt == 3u) e = scan_identity();
                    // ordered combination of the lanes pl .. 63 (an inclusive scan; lane 63 holds the result)
#pragma unroll
                    for (int d = 1; d < 64; d <<= 1) {
                        const ScanEl o = shfl_up_el(e, d);
                        if ((int)lane >= d) e = scan_combine(o, e);
                    }
                    acc = scan_combine(shfl_el(e, 63), acc);
                    if (pl >= 0 || hi - 64 <= 0) break;
                    hi -= 64;
                }
                excl = acc;
                if (lane == 0) scan_publish(desc, tile, kScanPre, scan_combine(excl, tile_agg));
            }
            if (lane == 0) {
                s_excl = excl;
                if (tile + 1u == n_tiles) ctr[CT_TOTAL_HITS] = excl.hits + tile_agg.hits;
            }
        }
        __syncthreads();
        // ---- pass 2: row offsets and segment starts (the counts come back from LDS: this loop is not unrolled) -----------
        ScanEl run = s_excl;
        for (uint32_t w = 0; w < wave; ++w) run = scan_combine(run, s_wave[w]);
        {
            int cn = c_first;
            uint32_t pb = pb_first;
            // this call's own range of the tile's window, as probe numbers (wave-uniform)
            const uint32_t own_lo_g = g_end - rp.win_len + rp.own_off_lo, own_hi_g = g_end - rp.win_len + rp.own_off_hi;
#pragma unroll 1
            for (int r = 0; r < kScanRounds; ++r) {
                const uint32_t ga = wg0 + (uint32_t)r * 64u, g = ga + lane;
                const uint32_t fr = s_f[wave * (uint32_t)kScanWaveSpan + (uint32_t)r * 64u + lane];
                const uint32_t n_valid = ga < g_end ? min(64u, g_end - ga) : 0u;  // (lanes behind the window's end hold kSkipN)
                const bool hit = fr >= 1u && fr < kPending;
                const unsigned long long hm = __ballot(hit), qm = __ballot(fr == 0u), cardm = __ballot(fr == kSkipCard);
                unsigned long long rm = resets_of(cn, pb, ga);
                const bool syn = syn_first && r == 0;
                if (syn) rm |= 1ull;
                // statistics (of the window's probes): a probe is a hit-probe, quiet, skipped for its cardinality or for its N
                const uint32_t n_hit = (uint32_t)__popcll(hm), n_card = (uint32_t)__popcll(cardm);
                st_valid += n_valid;
                st_hit += n_hit;
                st_card += n_card;
                st_n += n_valid - n_hit - n_card - (uint32_t)__popcll(qm);
                // row offsets: exclusive prefix of the hit counts (a count is at most max_cardinality: one 32-bit scan while
                // 64 of them fit, else two 16-bit halves)
                const uint32_t v = hit ? fr : 0u;
                unsigned long long incl;
                if (small_counts) {
                    incl = wave_incl_scan(v);
                } else {
                    const uint32_t lo16 = wave_incl_scan(v & 0xFFFFu), hi16 = wave_incl_scan(v >> 16);
                    incl = (unsigned long long)lo16 + ((unsigned long long)hi16 << 16);
                }
                if (lane < n_valid) row_off[g] = run.hits + incl - v;
                // (the row offset behind a window's last probe: what the CSR holds up to there)
                if (lane + 1u == n_valid && g + 1u == g_end) row_off[g_end] = run.hits + incl;
                // segment starts: a hit-probe of this call's own range with no hit-probe in front of it since its chunk
                // started, or with t* quiet probes in between.  (A window's first probe is never owned when the window
                // starts in mid-chunk: the look-back halo is at least one probe.)
                const uint32_t l_lo = own_lo_g > ga ? min(own_lo_g - ga, 64u) : 0u, l_hi = own_hi_g > ga ? min(own_hi_g - ga, 64u) : 0u;
                const unsigned long long ownm = (l_hi >= 64u ? ~0ull : (1ull << l_hi) - 1ull) & ~(l_lo >= 64u ? ~0ull : (1ull << l_lo) - 1ull);
                unsigned long long sm = 0, ambm = 0;
                if (hm & ownm) {
                    if (!rm) {
                        // no chunk starts in the round (all but a handful): the quiet probes below a lane, and -- by an inclusive
                        // max-scan over the hit lanes, whose counts ascend -- the count at the last hit-probe in front of it
                        const uint32_t Q = __builtin_amdgcn_mbcnt_hi((uint32_t)(qm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)qm, 0u));
                        uint32_t xp = __shfl_up(wave_incl_max_scan(hit ? Q + 1u : 0u), 1);
                        if (lane == 0u) xp = 0u;
                        const bool has_in = xp != 0u;                                       // a hit-probe in front of it in this round
                        const bool unknown = !has_in && scan_unknown(run);
                        const bool has_before = has_in || (run.flags & 1u) != 0u || unknown;  // (unknown: it may have one)
                        const uint32_t quiet = has_in ? Q - (xp - 1u) : run.c + Q;
                        sm = __ballot(hit && (!has_before || quiet >= rp.tstar)) & ownm;
                        ambm = __ballot(hit && unknown && quiet < rp.tstar) & ownm;
                    } else {
                        bool start = false, amb = false;
                        scan_starts_with_resets(hit, hm, qm, rm, syn, run.c, run.flags, rp.tstar, start, amb);
                        sm = __ballot(start) & ownm;
                        ambm = __ballot(amb) & ownm;
                    }
                }
                if (ambm && lane == 0u) atomicAdd(&ctr[CT_AMBIG], (unsigned long long)__popcll(ambm));
                if (sm) {
                    uint32_t base = 0;
                    if (lane == 0) base = atomicAdd(&s_nstart, (uint32_t)__popcll(sm));
                    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                    if ((sm >> lane) & 1ull) s_start[base + (uint32_t)__popcll(sm & lt_mask)] = g;
                }
                ScanEl e = round_el(hm, qm, rm, syn);
                e.hits = __shfl(incl, 63);
                run = scan_combine(run, e);
            }
        }
        __syncthreads();  // (s_tile, s_wave, s_excl are rewritten by the next tile)
        if (s_nstart > (uint32_t)(kStartCap - kScanTile)) flush_starts();
    }
    flush_starts();
    if (lane == 0) {
        atomicAdd(&sh_stat[0], st_n);
        atomicAdd(&sh_stat[1], st_card);
        atomicAdd(&sh_stat[2], st_hit);
        atomicAdd(&sh_stat[3], st_valid);
    }
    __syncthreads();
    if (tid == 0) {
        if (sh_stat[0]) atomicAdd(&ctr[CT_N_SKIPPED], sh_stat[0]);
        if (sh_stat[1]) atomicAdd(&ctr[CT_CARD_SKIPPED], sh_stat[1]);
        if (sh_stat[2]) atomicAdd(&ctr[CT_WITH_HITS], sh_stat[2]);
        if (sh_stat[3] > sh_stat[0]) atomicAdd(&ctr[CT_SEARCHED], sh_stat[3] - sh_stat[0]);
    }
}

// The sum of the raw interval sizes over the searched probes (asgart_stats.raw_hits: a statistic of the call, wanted by
// the parity tests and the bench's yardstick, not by the path) -- computed when the statistics are asked for, from the
// per-probe arrays the call left in its workspace, instead of costing every call a second read of 4 bytes per probe.
// A probe the position filter answered has no interval there (p_raw = kRawUnknown): its k-mer is looked up here.
template <class SlotT>
__global__ __launch_bounds__(256) void raw_hits_kernel(IndexView<SlotT> ix, RunParams rp, const uint32_t *__restrict__ p_filt,
                                                       const uint32_t *__restrict__ p_raw, unsigned long long *__restrict__ out) {
    unsigned long long sum = 0;
    const uint32_t n_t = rp.n_tiles(1024u);
    for (uint32_t t = blockIdx.x; t < n_t; t += gridDim.x) {
        uint32_t g_end;
        const uint32_t g0 = rp.tile_of(t, 1024u, g_end);
#pragma unroll 1
        for (int a = 0; a < 4; ++a) {
            const uint32_t g = g0 + (uint32_t)a * 256u + threadIdx.x;
            if (g >= g_end || p_filt[g] == kSkipN) continue;
            uint32_t raw = p_raw[g];
            if (raw == kRawUnknown) {
                const int c = chunk_of(rp.ch, g);
                const uint32_t md = rp.mode_of(c);
                const uint64_t cs = rp.ch.start[c], cl = rp.ch.len[c];
                const uint64_t i = (uint64_t)(g - rp.ch.pbase[c] + 1) * (uint64_t)rp.step;
                uint32_t first = 0;
                uint64_t q2 = 0, lo, hi;
                const uint64_t q = probe_key(ix.text, cs, cl, i, rp.k, (md & 2u) != 0u, (md & 1u) != 0u, &first, &q2);
                ProbeRef pr;
                pr.p = (md & 2u) ? ix.text + cs + cl - 1u - i : ix.text + cs + i;
                pr.dir = (md & 2u) ? -1 : 1;
                pr.comp = (md & 1u) != 0u;
                kmer_range(ix, q, q2, pr, lo, hi);
                raw = (uint32_t)(hi - lo);
            }
            sum += raw;
        }
    }
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off);
    if ((threadIdx.x & 63u) == 0 && sum) atomicAdd(out, sum);
}

// ---------------------------------------------------------------- K3 ---------
template <class SlotT>
__global__ __launch_bounds__(256) void fill_small_kernel(IndexView<SlotT> ix, RunParams rp,
                                                         const SlotT *__restrict__ p_lo,
                                                         const uint32_t *__restrict__ p_raw,
                                                         const uint32_t *__restrict__ p_filt,
                                                         const unsigned long long *__restrict__ row_off,
                                                         SlotT *__restrict__ hits) {
    uint32_t g_end;
    const uint32_t gb = rp.tile_of(blockIdx.x, 256u, g_end);
    const uint32_t g = gb + threadIdx.x;
    // (the chunk of the workgroup's first probe, once, in scalar registers; a thread whose probe lies behind a chunk boundary
    // steps on from there -- 256 consecutive probes rarely span two chunks -- instead of bisecting the table per thread)
    int c = chunk_of_uniform(rp.ch, gb);
    if (g >= g_end) return;
    const uint32_t f = p_filt[g];
    if (f == 0 || f >= kPending) return;
    const uint32_t raw = p_raw[g];
    if (raw > (uint32_t)kSmallInterval) return;
    while (c + 1 < rp.ch.n_chunks && g >= rp.ch.pbase[c + 1]) ++c;
    const uint64_t s = rp.ch.start[c], L = rp.ch.len[c];
    const uint64_t i = (uint64_t)(g - rp.ch.pbase[c] + 1) * (uint64_t)rp.step;
    const uint64_t lo = p_lo[g];
    unsigned long long w = row_off[g];
    for (uint32_t r = 0; r < raw; ++r) {
        const SlotT x = ix.sa[lo + r];
        if (keep_hit(x, i, s, L, (rp.mode_of(c) & 2u) != 0u)) hits[w++] = x;
    }
}

// Large intervals, one wave per 64 of them (set-up in parallel across the lanes, like the count
// kernel): ballot / prefix-popcount compaction of the kept hits, SA order kept.
template <class SlotT>
__global__ __launch_bounds__(256) void fill_big_kernel(IndexView<SlotT> ix, RunParams rp,
                                                       const SlotT *__restrict__ p_lo,
                                                       const uint32_t *__restrict__ p_raw,
                                                       const uint32_t *__restrict__ p_filt,
                                                       const unsigned long long *__restrict__ row_off,
                                                       SlotT *__restrict__ hits,
                                                       const uint32_t *__restrict__ big_list,
                                                       const unsigned long long *__restrict__ ctr) {
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const uint64_t n_big = ctr[CT_BIG];
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t e0 = wave * 64u; e0 < n_big; e0 += n_waves * 64u) {
        const uint32_t np = (uint32_t)min((uint64_t)64, n_big - e0);
        unsigned long long s = 0, L = 0, i = 0, lo = 0, hi = 0, w0 = 0;
        uint32_t md = 0;
        bool want = false;
        if (lane < np) {
            const uint32_t g = big_list[e0 + lane];
            const uint32_t f = p_filt[g];
            want = f != 0 && f < kPending;  // skipped / empty rows have nothing to fill
            if (want) {
                const int c = chunk_of(rp.ch, g);
                md = rp.mode_of(c);
                s = rp.ch.start[c];
                L = rp.ch.len[c];
                i = (unsigned long long)(g - rp.ch.pbase[c] + 1) * (unsigned long long)rp.step;
                lo = p_lo[g];
                hi = lo + p_raw[g];
                w0 = row_off[g];
            }
        }
        unsigned long long todo = __ballot(want);
        while (todo) {
            const uint32_t p = (uint32_t)(__ffsll((long long)todo) - 1);
            todo &= todo - 1;
            const unsigned long long lo_p = lane_of(lo, p), hi_p = lane_of(hi, p), i_p = lane_of(i, p),
                                     s_p = lane_of(s, p), L_p = lane_of(L, p);
            const bool rev_p = (lane_of(md, p) & 2u) != 0u;
            unsigned long long w = lane_of(w0, p);
            for (unsigned long long base = lo_p; base < hi_p; base += 256) {  // four slices in flight per round trip
                SlotT x[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const unsigned long long r = base + 64u * u + lane;
                    x[u] = r < hi_p ? ix.sa[r] : (SlotT)0;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const unsigned long long r = base + 64u * u + lane;
                    const bool keep = r < hi_p && keep_hit(x[u], i_p, s_p, L_p, rev_p);
                    const unsigned long long m = __ballot(keep);
                    if (keep) hits[w + __popcll(m & lt_mask)] = x[u];
                    w += __popcll(m);
                }
            }
        }
    }
}

// ---------------------------------------------------------------- K4 ---------
// Seed-extension automaton, one wavefront per independent segment.
//
// Representation (equivalent to, not a transcription of, src/automaton.rs:87-200):
//   * only LIVE (active) arms are kept.  An arm that turns inactive can never
//     be extended again (try_extend_arms tests `a.active`, :68) and is only
//     looked at once more, when its family is flushed (:182-200), so it is
//     retired at once: written to the output list if len(right) >= M, dropped
//     otherwise.  (The reference's `retain` at :173-179 removes a subset of the
//     same arms; both removals are unobservable.)
//   * the family is flushed when the live list becomes empty; its members are
//     the retired arms, ordered by creation number (== position in the
//     reference's `arms` vector).  The host sorts records by
//     (segment start, family ordinal, creation number).
//   * arms still live at the end of the chunk are dropped AND their family's
//     retired members are void (:201-203): a tombstone record says so.
//   * <= 64 live arms: one arm per lane, in registers; otherwise LDS arrays.
//   * the hit rows of up to 64 consecutive probes are contiguous in the CSR and
//     are staged through LDS with one coalesced load.
constexpr uint32_t kTombstone = 0xFFFFFFFFu;

// Diagnostic build only (-DASGART_PROFILE_EXTEND): per-phase cycle sums of the extension kernel
// are added to ctr[16..]; never enabled in the shipped library.
#ifdef ASGART_PROFILE_EXTEND
#define PROF_DECL unsigned long long pf_t0 = 0, pf_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define PROF_START() pf_t0 = __builtin_amdgcn_s_memtime()
#define PROF_STOP(slot) pf_acc[slot] += __builtin_amdgcn_s_memtime() - pf_t0
#define PROF_COUNT(slot, v) pf_acc[slot] += (v)  // (slots 10, 11: sums of live arms and hits over the hit-probes)
#define PROF_MAX(slot, v) pf_acc[slot] = pf_acc[slot] > (unsigned long long)(v) ? pf_acc[slot] : (unsigned long long)(v)
#define PROF_SEG_BEGIN() const unsigned long long pf_seg0 = __builtin_amdgcn_s_memtime()
#define PROF_FLUSH()                                                             \
    do {                                                                         \
        const unsigned long long pf_dt = __builtin_amdgcn_s_memtime() - pf_seg0; \
        if (lane == 0 && atomicMax(&P.ctr[28], pf_dt) < pf_dt) {                 \
            P.ctr[29] = g0;                                                      \
            P.ctr[30] = ((unsigned long long)pf_acc[5] << 32) | pf_acc[3];       \
            P.ctr[31] = ((unsigned long long)pf_acc[10] << 32) | pf_acc[11];     \
            P.ctr[32] = pf_acc[0];                                               \
            P.ctr[33] = pf_acc[6];                                               \
            P.ctr[25] = pf_acc[9];                                               \
            for (int pf_i = 0; pf_i < 12; ++pf_i) P.ctr[56 + pf_i] = pf_acc[pf_i]; \
        }                                                                        \
        if (lane == 0)                                                           \
            for (int pf_i = 0; pf_i < 12; ++pf_i)                                \
                if (pf_acc[pf_i]) atomicAdd(&P.ctr[16 + pf_i], pf_acc[pf_i]);     \
        if (lane == 0) {                                                         \
            int pf_b = 0;                                                        \
            while (pf_b < 15 && (2ull << pf_b) <= pf_acc[9]) ++pf_b;             \
            atomicAdd(&P.ctr[CT_HIST_PEAK + pf_b], 1ull);                        \
            pf_b = 0;                                                            \
            while (pf_b < 15 && (2ull << pf_b) <= pf_acc[5] + pf_acc[3]) ++pf_b; \
            atomicAdd(&P.ctr[CT_HIST_PROBES + pf_b], pf_acc[5] + pf_acc[3]);     \
        }                                                                        \
        for (int pf_i = 0; pf_i < 12; ++pf_i) pf_acc[pf_i] = 0;                  \
    } while (0)
#define DBG_ADD(slot, v) atomicAdd(&P.ctr[40 + (slot)], (unsigned long long)(v))
#else
#define DBG_ADD(slot, v)
#define PROF_DECL
#define PROF_START()
#define PROF_STOP(slot)
#define PROF_COUNT(slot, v)
#define PROF_MAX(slot, v)
#define PROF_SEG_BEGIN()
#define PROF_FLUSH()
#endif
constexpr int kHitBatch = 1024;  // LDS staging for the hit rows of one probe batch
constexpr uint32_t kEscalateCost = 40000;  // sum of (live arms + hits) over LDS-path probes

// A RUN over part of a long segment (extend_k8_kernel<..., RANGE = true>; plan_ranges_kernel makes them): the walk starts at
// probe g_begin with no arm and stops in front of g_stop; records are written from probe emit_from on (the cut: a hit-probe;
// what lies in front of it is the run's warm-up and belongs to the range before).
struct RangeRun {
    uint32_t g_begin, g_stop;  // [g_begin, g_stop) (g_stop = ~0u: to the segment's end)
    uint32_t g_seg0;           // first probe of the segment (record key, chunk)
    uint32_t emit_from;        // the cut this run reports from (g_seg0: from the start)
    uint32_t flags;            // kRunNoEmit | kRunLast
    uint32_t split;            // which split segment (struct SplitSeg)
    uint32_t pad0, pad1;
};
constexpr uint32_t kRunNoEmit = 1u;    // a warm-up on its own: only its final state is wanted (what the run behind the cut starts from)
constexpr uint32_t kRunLast = 2u;      // the run that reaches the segment's end
constexpr uint32_t kRunDumpCap = 5120; // arms a run can leave alive (the long shape's slots)
// per run two states, 8 words each (run_meta): [0] what it holds when it STOPS, [1] what it holds when it reaches its cut:
// 0 arms written to run_dump  1 flushes since the cut  2 family open  3 probes a flush is still held back for  4 ([0] only) gave up
// (more arms than slots, a probe with more hits than the staging area)
// per run two dumps of kRunDumpCap arms (run_dump), kDumpWords per arm, the creation number first (32-bit positions: creation
// number, left start, left end, right start | right end, threshold, gap, 0; 64-bit: creation number, threshold, gap, 0 | left
// start, left end | right start, right end)
template <class PosT> constexpr uint32_t kDumpWords = sizeof(PosT) == 4 ? 8u : 12u;
struct SplitSeg {
    uint32_t g_seg0, run_base, n_ranges, cut_base;  // runs run_base .. + n_ranges - 1: the ranges; cuts cut_base .. + n_ranges - 2
    uint32_t span, hits, tier, warm;                // (what the placement knew of the segment; the warm-up its ranges got)
};

template <class PosT>
struct ExtParams {
    const RangeRun *runs;                 // (RANGE launches) the work list
    uint32_t *run_meta, *run_dump;        // ... and what the runs leave behind (see RangeRun)
    RunParams rp;
    const uint32_t *p_filt;
    const unsigned long long *row_off;
    const PosT *hits;
    const uint32_t *seg_list;
    const unsigned long long *n_seg_ptr;  // device count of seg_list entries
    unsigned long long *cursor;           // work-fetch cursor
    SdRec *recs;
    unsigned long long rec_cap;
    uint32_t *ovf_list;                   // segments this launch gives up on go here (may be null)
    unsigned long long *ovf_count;        // ... appended at *ovf_count (device counter)
    char *scratch;                        // heavy global tier: per-workgroup arm storage
    uint32_t gen_bits;                    // arm-resident kernels: bits of the table generation counter (tests shrink it)
    uint32_t escalate_cost;               // one-wave tiers: give up after this much LDS-path work
    uint32_t cap_limit;                   // effective live-arm capacity (<= CAP; tests lower it)
    uint32_t heavy_cap;                   // K4b MODE 2 (tier 7): arm slots per workgroup in its HBM slice
    uint32_t solo_hits;                   // K6: probes with up to this many hits may run on wave 0 alone (0: never)
    uint32_t k8_delay;                    // K8 (tests): cycles the ranking wave waits before it reads the free counts
    uint32_t tier;                        // the tier this launch runs as (statistics)
    unsigned long long *seg_slots;        // 4096 words of this launch's tier: start time of the segment a workgroup is on (seg_clock)
    unsigned long long *ctr;
    unsigned long long *hb;               // heartbeat slots of this launch's tier (pinned host memory; null: none)
};

// a workgroup's sign of life (see SearchCtx::heartbeat): which segment it is on and how far
template <class PosT>
__device__ inline void heartbeat(const ExtParams<PosT> &P, uint32_t g0, uint32_t at) {
    if (P.hb) {
        unsigned long long *slot = P.hb + 2u * (blockIdx.x % 256u);
        __builtin_nontemporal_store((unsigned long long)g0 | 1ull << 63, slot);
        __builtin_nontemporal_store((unsigned long long)at, slot + 1);
    }
}

// The whole predicate of try_extend_arms (src/automaton.rs:68-70) for an active arm with right
// segment [rs, re], threshold thr, and a hit m = [x, x+k]:
//     d_ss(a.right, m) < thr  &&  m.end > a.right.end
// Because len(right) >= k always (an arm starts as [x0, x0+k] and re only grows), this is
// exactly   re - k < x < re + thr   (thr >= 1), i.e. (x - lo) < w in unsigned arithmetic with
// lo = re - k + 1, w = thr + k - 1:
//   x <= re : x > re-k >= rs so m.start lies in [rs, re]            -> d_ss = 0 < thr
//   x >  re : no containment, d_ss = min(x+k-rs, x-re) = x - re     -> accept iff x - re < thr
// (thr == 0 accepts nothing.)  tests/test_oracle_golden.py checks the equivalence exhaustively
// against the oracle's literal d_ss.
template <class PosT>
__device__ inline bool arm_accepts(PosT x, PosT re, uint32_t thr, uint32_t k) {
    const PosT lo = (PosT)(re - k + 1u);
    const uint64_t w = thr ? (uint64_t)thr + k - 1u : 0u;
    return (uint64_t)(PosT)(x - lo) < w;
}

// max(e, (0.1 * len as f64) as i64)   (src/automaton.rs:69).  The double product truncates to
// len / 10 for every len < 9e15 (0.1 rounds UP to 0.1000000000000000055, so the product can only
// cross an integer boundary once len / 10 * 1.1e-16 reaches 0.1; checked for 3M values up to 2^40 in
// tests/test_oracle_golden.py::test_tenth_threshold_is_integer_division), so the kernels divide.
__device__ inline uint32_t arm_threshold(uint64_t left_len, uint32_t G) {
    const uint64_t tenth = left_len <= 0xFFFFFFFFull ? (uint64_t)((uint32_t)left_len / 10u) : left_len / 10u;
    const uint64_t thr = tenth > (uint64_t)G ? tenth : (uint64_t)G;
    return thr > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)thr;
}

// Output records are appended to one device-wide list.  A global atomic WITH its return value costs
// a full memory round trip (a microsecond on the critical path of a serial segment), so every wave
// reserves kRecChunk slots at a time and hands them out from registers; what is left of a chunk
// when the wave takes the next one (or exits) is marked void (g_start = kVoidStart: sorts last, the
// host stops there).
constexpr uint32_t kRecChunk = 32;
constexpr uint32_t kVoidStart = 0xFFFFFFFFu;
struct RecAlloc {
    unsigned long long next = 0;  // wave-uniform
    uint32_t left = 0;
};
// Statistics of the persistent workgroups, kept in DEVICE memory (not in registers: the 1024-thread shapes sit at their
// 128-VGPR cap and every value that lives across the segment loop is a spill into it):
//   wg_begin / wg_busy   the workgroup's lifetime goes into its tier's tally (- start, + end: two atomics per workgroup)
//   seg_clock            called by ONE thread where the workgroup takes its next segment: the time since the previous
//                        call is one segment's duration; the longest per tier is the serial floor of the extension
template <class PosT>
__device__ inline void wg_begin(const ExtParams<PosT> &P) {
    if (threadIdx.x == 0 && P.tier >= 1u && P.tier <= (uint32_t)kTiers) {
        atomicAdd(&P.ctr[CT_BUSY1 + P.tier - 1u], 0ull - wall_clock64());
    }
}
// (the slots are zero when a launch starts -- the host clears them, and a workgroup leaves its slot zero -- : the thread
// that fetches the segments and the thread that closes the last one need not be the same)
template <class PosT>
__device__ inline void seg_clock(const ExtParams<PosT> &P, bool last = false) {
    if (!P.seg_slots || P.tier < 2u || P.tier > (uint32_t)kTiers) return;  // (tier 1: a million tiny segments)
    const unsigned long long now = wall_clock64();
    const unsigned long long prev = atomicExch(&P.seg_slots[blockIdx.x & 4095u], last ? 0ull : now);
    if (prev && now > prev) atomicMax(&P.ctr[CT_SEGMAX1 + P.tier - 1u], now - prev);
}
template <class PosT>
__device__ inline void wg_busy(const ExtParams<PosT> &P) {
    if (threadIdx.x == 0 && P.tier >= 1u && P.tier <= (uint32_t)kTiers) {
        seg_clock(P, true);  // (closes the last segment)
        atomicAdd(&P.ctr[CT_BUSY1 + P.tier - 1u], wall_clock64());
        atomicAdd(&P.ctr[CT_WGS1 + P.tier - 1u], 1ull);
    }
}
template <class PosT>
__device__ inline void rec_flush(RecAlloc &ra, const ExtParams<PosT> &P, int lane) {
    if ((uint32_t)lane < ra.left && ra.next + (unsigned)lane < P.rec_cap) P.recs[ra.next + (unsigned)lane].g_start = kVoidStart;
    ra.left = 0;
}
// all 64 lanes call; em = ballot of the emitting lanes (non-zero); returns this lane's slot
template <class PosT>
__device__ inline unsigned long long rec_slot(RecAlloc &ra, const ExtParams<PosT> &P, unsigned long long em, int lane) {
    const uint32_t n = (uint32_t)__popcll(em);
    if (n > ra.left) {
        rec_flush(ra, P, lane);
        const uint32_t take = n > kRecChunk ? n : kRecChunk;
        unsigned long long b = 0;
        if (lane == 0) b = atomicAdd(&P.ctr[CT_SD], (unsigned long long)take);
        ra.next = lane_of(b, 0u);
        ra.left = take;
    }
    const unsigned long long at = ra.next + (unsigned)__popcll(em & ((1ull << lane) - 1ull));
    ra.next += n;
    ra.left -= n;
    return at;
}

template <class PosT, int CAP>
__global__ __launch_bounds__(64) void extend_kernel(ExtParams<PosT> P) {
    __shared__ PosT s_ls[CAP], s_le[CAP], s_rs[CAP], s_re[CAP];
    __shared__ uint32_t s_gap[CAP], s_thr[CAP], s_seq[CAP], s_pend[CAP];
    __shared__ PosT s_hits[kHitBatch];
    // candidate index of the LDS path: arms bucketed by right end (see "LDS path")
    constexpr uint32_t HT = CAP <= 256 ? 256u : (CAP <= 1024 ? 1024u : 4096u);
    __shared__ uint32_t s_head[HT];
    __shared__ uint16_t s_next[CAP], s_wide[CAP];
    __shared__ PosT s_wlo[CAP];      // wide arms, packed: accepts x iff (x - s_wlo[w]) < s_ww[w]
    __shared__ uint32_t s_ww[CAP];
    const int lane = threadIdx.x;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const RunParams &rp = P.rp;
    const uint64_t n_seg = *P.n_seg_ptr;
    const uint32_t k = (uint32_t)rp.k, step = (uint32_t)rp.step, G = rp.G;
    const uint32_t thr0 = arm_threshold(k, G);
    RecAlloc rec_alloc;
    wg_begin(P);
    PROF_DECL;

    // Segments are fetched kFetch at a time (one contended global atomic per group).  The list is
    // sorted longest first: in its head the members of a group are strided, so that the longest
    // segments go to different waves instead of eight of them to the same one.
    constexpr unsigned long long kFetch = 8;
    const unsigned long long head_groups = min((unsigned long long)gridDim.x, n_seg / kFetch);
    const unsigned long long head = head_groups * kFetch;
    unsigned long long seg_base = 0;
    uint32_t seg_j = (uint32_t)kFetch;
    for (;;) {
        if (seg_j == (uint32_t)kFetch) {
            unsigned long long sb = 0;
            if (lane == 0) sb = atomicAdd(P.cursor, kFetch);
            seg_base = uni(sb);
            seg_j = 0;
        }
        const uint32_t j = seg_j++;
        const unsigned long long seg = seg_base < head ? seg_base / kFetch + (unsigned long long)j * head_groups
                                                       : seg_base + j;
        if (seg_base >= n_seg) break;
        if (seg >= n_seg) continue;
        const uint32_t g0 = P.seg_list[seg];
        PROF_SEG_BEGIN();
        if (lane == 0 && j == 0) {
            heartbeat(P, g0, 0u);
            seg_clock(P);
        }  // (once per fetched group of segments)
        const int c = chunk_of_uniform(rp.ch, g0);
        const uint64_t cs = rp.ch.start[c], cl = rp.ch.len[c];
        const bool seg_rev = (rp.mode_of(c) & 2u) != 0u;  // (the orientation of the chunk's pass)
        const uint32_t pb = rp.ch.pbase[c];
        const uint32_t chunk_end = rp.ch.pbase[c + 1];
        const uint32_t g_end = min(chunk_end, rp.win_end(g0));  // (sharded calls: the window ends first)

        // live arms: lane j holds arm j while in_regs (A <= 64), else s_*[0..A)
        PosT r_ls = 0, r_le = 0, r_rs = 0, r_re = 0;
        uint32_t r_gap = 0, r_thr = 0, r_seq = 0;
        uint32_t A = 0, quiet = 0, fam_seq = 0, next_seq = 0, lds_cost = 0;
        bool in_regs = true, overflow = false, done = false, fam_open = false;  // fam_open: a family is pending

        // ---- helpers -------------------------------------------------------
        auto emit_records = [&](bool emit, PosT ls, PosT le, PosT rs, PosT re, uint32_t seq) {
            const unsigned long long em = __ballot(emit);
            if (!em) return;
            const unsigned long long at = rec_slot(rec_alloc, P, em, lane);
            if (emit) {
                if (at < P.rec_cap) {
                    const uint64_t ll = (uint64_t)le - (uint64_t)ls;
                    SdRec r;
                    r.g_start = g0;
                    r.fam_seq = fam_seq;
                    r.create_seq = seq;
                    r.pad = 0;
                    // left fix-up, src/bin/asgart.rs:229-237
                    r.sd.left = seg_rev ? cs + cl - (uint64_t)ls - ll : (uint64_t)ls + cs;
                    r.sd.right = rs;
                    r.sd.left_length = ll;
                    r.sd.right_length = (uint64_t)re - (uint64_t)rs;
                    P.recs[at] = r;
                }
            }
        };
        // the flush of src/automaton.rs:182-200: every arm inactive
        auto maybe_close = [&]() {
            if (fam_open && A == 0) {
                ++fam_seq;
                next_seq = 0;
                fam_open = false;
            }
        };
        // retire arms whose gap reached G (src/automaton.rs:166-171 + flush bookkeeping)
        auto retire_regs = [&]() {
            const bool dead = (uint32_t)lane < A && r_gap >= G;
            if (!__ballot(dead)) return;
            emit_records(dead && (uint64_t)(r_re - r_rs) >= rp.M, r_ls, r_le, r_rs, r_re, r_seq);
            const bool alive = (uint32_t)lane < A && !dead;
            const unsigned long long am = __ballot(alive);
            if (alive) {
                const int d = __popcll(am & lt_mask);
                s_ls[d] = r_ls; s_le[d] = r_le; s_rs[d] = r_rs; s_re[d] = r_re;
                s_gap[d] = r_gap; s_thr[d] = r_thr; s_seq[d] = r_seq;
            }
            __syncthreads();
            A = (uint32_t)__popcll(am);
            if ((uint32_t)lane < A) {
                r_ls = s_ls[lane]; r_le = s_le[lane]; r_rs = s_rs[lane]; r_re = s_re[lane];
                r_gap = s_gap[lane]; r_thr = s_thr[lane]; r_seq = s_seq[lane];
            }
            __syncthreads();
        };
        auto retire_lds = [&]() {
            uint32_t w = 0;
            bool any_dead = false;
            for (uint32_t t0 = 0; t0 < A; t0 += 64) {
                const uint32_t j = t0 + lane;
                PosT ls = 0, le = 0, rs = 0, re = 0;
                uint32_t gp = 0, th = 0, sq = 0;
                bool valid = j < A;
                if (valid) {
                    ls = s_ls[j]; le = s_le[j]; rs = s_rs[j]; re = s_re[j];
                    gp = s_gap[j]; th = s_thr[j]; sq = s_seq[j];
                }
                const bool dead = valid && gp >= G;
                any_dead |= __ballot(dead) != 0ull;
                emit_records(dead && (uint64_t)(re - rs) >= rp.M, ls, le, rs, re, sq);
                const bool alive = valid && !dead;
                const unsigned long long am = __ballot(alive);
                __syncthreads();
                if (alive && any_dead) {
                    const uint32_t d = w + __popcll(am & lt_mask);
                    s_ls[d] = ls; s_le[d] = le; s_rs[d] = rs; s_re[d] = re;
                    s_gap[d] = gp; s_thr[d] = th; s_seq[d] = sq; s_pend[d] = 0;
                }
                w += __popcll(am);
                __syncthreads();
            }
            A = w;
        };
        auto to_lds = [&]() {
            if ((uint32_t)lane < A) {
                s_ls[lane] = r_ls; s_le[lane] = r_le; s_rs[lane] = r_rs; s_re[lane] = r_re;
                s_gap[lane] = r_gap; s_thr[lane] = r_thr; s_seq[lane] = r_seq; s_pend[lane] = 0;
            }
            __syncthreads();
            in_regs = false;
        };
        auto to_regs = [&]() {
            if ((uint32_t)lane < A) {
                r_ls = s_ls[lane]; r_le = s_le[lane]; r_rs = s_rs[lane]; r_re = s_re[lane];
                r_gap = s_gap[lane]; r_thr = s_thr[lane]; r_seq = s_seq[lane];
            }
            __syncthreads();
            in_regs = true;
        };
        // q consecutive processed probes without hits
        auto advance_quiet = [&](uint32_t q) {
            quiet += q;
            if (A > 0) {
                const uint32_t add = q * step;
                if (in_regs) {
                    if ((uint32_t)lane < A) r_gap = r_gap + add < r_gap ? 0xFFFFFFFFu : r_gap + add;
                    retire_regs();
                } else {
                    for (uint32_t j = lane; j < A; j += 64) {
                        const uint32_t gp = s_gap[j];
                        s_gap[j] = gp + add < gp ? 0xFFFFFFFFu : gp + add;
                    }
                    __syncthreads();
                    PROF_STOP(7);
                    PROF_START();
                    retire_lds();
                    if (A <= 32) to_regs();
                    PROF_STOP(8);
                }
            }
            maybe_close();
            if (A == 0 && quiet >= rp.tstar) done = true;
        };

        for (uint32_t g = g0; g < g_end && !done;) {
            // ---- stage a batch of up to 64 probes ------------------------------
            PROF_START();
            const uint32_t nb = min(64u, g_end - g);
            const uint32_t f_l = (uint32_t)lane < nb ? P.p_filt[g + lane] : kSkipN;
            const unsigned long long r_l = (uint32_t)lane < nb ? P.row_off[g + lane] : 0ull;
            const unsigned long long r_hi = P.row_off[g + nb];
            const unsigned long long base = __shfl(r_l, 0);
            unsigned long long r_next = __shfl_down(r_l, 1);
            if ((uint32_t)lane + 1 >= nb) r_next = r_hi;
            const bool fits = (uint32_t)lane < nb && r_next - base <= (unsigned long long)kHitBatch;
            const unsigned long long fm = __ballot(fits);
            uint32_t nbb = (~fm == 0ull) ? 64u : (uint32_t)(__ffsll((long long)~fm) - 1);
            if (nbb > nb) nbb = nb;
            bool first_from_global = false;
            if (nbb == 0) {  // a single row larger than the staging buffer
                nbb = 1;
                first_from_global = true;
            }
            const uint32_t rel_l = (uint32_t)(r_l - base);
            if (!first_from_global) {
                const unsigned long long end = nbb == nb ? r_hi : __shfl(r_l, (int)nbb);
                const uint32_t tot = (uint32_t)(end - base);
                // four loads per lane in flight per round trip; most batches need a single round
                for (uint32_t r0 = 0; r0 < tot; r0 += 256u) {
                    PosT tmp[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const uint32_t r = r0 + lane + 64u * u;
                        tmp[u] = r < tot ? P.hits[base + r] : (PosT)0;
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const uint32_t r = r0 + lane + 64u * u;
                        if (r < tot) s_hits[r] = tmp[u];
                    }
                }
            }
            __syncthreads();
            const unsigned long long in_batch = nbb >= 64 ? ~0ull : ((1ull << nbb) - 1ull);
            const unsigned long long hm = __ballot(f_l >= 1u && f_l < kPending) & in_batch;
            const unsigned long long qm = __ballot(f_l == 0u) & in_batch;
            PROF_STOP(0);
            PROF_COUNT(1, 1);
            uint32_t pos = 0;
            while (!done) {
                const unsigned long long hmr = pos >= 64 ? 0ull : (hm >> pos) << pos;
                if (!hmr) break;
                const uint32_t b = (uint32_t)(__ffsll((long long)hmr) - 1);
                {
                    const unsigned long long range = ((1ull << b) - 1ull) & ~((1ull << pos) - 1ull);
                    const uint32_t q = (uint32_t)__popcll(qm & range);
                    if (q) {
                        advance_quiet(q);
                        if (done) break;
                    }
                }
                quiet = 0;
                pos = b + 1;
                const uint32_t cnt = __shfl(f_l, (int)b);
                const uint32_t off = __shfl(rel_l, (int)b);
                const uint64_t i = (uint64_t)(g + b - pb + 1) * step;
                const unsigned long long row = base + off;
                if (in_regs && A + cnt <= 64u && !first_from_global) {
                    // ---------------- register path -------------------------------
                    PROF_START();
                    PROF_COUNT(3, 1);
                    PROF_MAX(9, A + cnt);
                    bool pend = false;
                    PosT pend_x = 0;
                    uint32_t newc = 0;
                    for (uint32_t t = 0; t < cnt; ++t) {
                        const PosT x = s_hits[off + t];
                        const bool ok = (uint32_t)lane < A && arm_accepts<PosT>(x, r_re, r_thr, k);
                        const unsigned long long m = __ballot(ok);
                        if (m) {  // ExtendArm on the first matching arm; last hit wins
                            if (lane == __ffsll((long long)m) - 1) {
                                pend = true;
                                pend_x = x;
                            }
                        } else {  // NewArm
                            if ((uint32_t)lane == A + newc) {
                                r_ls = (PosT)i; r_le = (PosT)(i + k); r_rs = x; r_re = (PosT)(x + k);
                                r_gap = step;  // not dirty: aged by this very probe
                                r_thr = thr0;
                                r_seq = next_seq + newc;
                            }
                            ++newc;
                        }
                    }
                    if ((uint32_t)lane < A) {
                        if (pend) {
                            r_re = (PosT)(pend_x + k);
                            r_le = (PosT)(i + k);
                            r_thr = arm_threshold((uint64_t)(i + k) - (uint64_t)r_ls, G);
                            r_gap = 0;
                        } else {
                            r_gap += step;
                        }
                    }
                    A += newc;
                    next_seq += newc;
                    retire_regs();
                    PROF_STOP(2);
                } else {
                    // ---------------- LDS path ------------------------------------
                    if (in_regs) to_lds();
                    // hand the segment to the block-cooperative heavy tier when it does not fit
                    // this wave's LDS share, or keeps producing many-hit x many-arm probes
                    lds_cost += A + cnt;
                    if (A + cnt > min((uint32_t)CAP, P.cap_limit) || lds_cost > P.escalate_cost) {
#ifdef ASGART_PROFILE_EXTEND
                        if (lane == 0) printf("[light overflow] g0=%u g=%u A=%u cnt=%u cost=%u first_glob=%d\n", g0, g + b, A, cnt, lds_cost, (int)first_from_global);
#endif
                        overflow = true;
                        done = true;
                        break;
                    }
                    const uint32_t A_old = A;
                    const bool from_lds = !first_from_global;
                    PROF_COUNT(5, 1);
                    PROF_COUNT(10, A_old);
                    PROF_COUNT(11, cnt);
                    PROF_MAX(9, A_old + cnt);
                    PROF_START();
                    // An arm accepts hit x iff  re - k < x < re + thr  (d_ss of src/automaton.rs:207-216
                    // with m = [x, x+k) and len(right) >= k).  So instead of testing every arm
                    // (automaton.rs:67-78) the arms whose thr is the floor G ("narrow") are hashed by
                    // bucket(re) with bucket width G + k: a hit can only be accepted by narrow arms
                    // in two buckets.  The few arms with a long left segment (thr > G) are kept in a
                    // list and tested one by one.  The answer is the smallest accepting arm index.
                    const uint32_t Wb = G + k;
                    uint32_t hmask = 63u;  // table sized to the live arms (power of two <= HT)
                    while (hmask + 1u < HT && hmask + 1u < 2u * A_old) hmask = (hmask << 1) | 1u;
                    for (uint32_t h = lane; h <= hmask; h += 64) s_head[h] = 0xFFFFFFFFu;
                    __syncthreads();
                    uint32_t n_wide = 0;
                    for (uint32_t t0 = 0; t0 < A_old; t0 += 64) {
                        const uint32_t j = t0 + lane;
                        const bool valid = j < A_old;
                        const bool narrow = valid && s_thr[j] <= G;
                        if (narrow) {
                            const uint32_t b = (uint32_t)((uint64_t)s_re[j] / Wb);
                            const uint32_t h = ((b * 2654435761u) >> 12) & hmask;
                            s_next[j] = (uint16_t)atomicExch(&s_head[h], j);
                        }
                        const unsigned long long wm = __ballot(valid && !narrow);
                        if (valid && !narrow) {
                            const uint32_t d = n_wide + __popcll(wm & lt_mask);
                            const uint32_t th = s_thr[j];
                            const uint64_t wv = (uint64_t)th + k - 1u;
                            s_wide[d] = (uint16_t)j;
                            s_wlo[d] = (PosT)(s_re[j] - k + 1u);
                            s_ww[d] = wv > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)wv;
                        }
                        n_wide += __popcll(wm);
                    }
                    __syncthreads();
                    PROF_STOP(4);
                    PROF_START();
                    for (uint32_t t0 = 0; t0 < cnt; t0 += 64) {
                        const uint32_t t = t0 + lane;
                        const bool valid = t < cnt;
                        PosT x = 0;
                        if (valid) x = from_lds ? s_hits[off + t] : P.hits[row + t];
                        uint32_t best = 0xFFFFFFFFu;
                        if (valid) {
                            // narrow candidates: re in (x - G, x + k)
                            const uint64_t lo_re = (uint64_t)x + 1u > (uint64_t)G ? (uint64_t)x + 1u - G : 0u;
                            const uint32_t b0 = (uint32_t)(lo_re / Wb);
                            const uint32_t b1 = (uint32_t)(((uint64_t)x + k - 1u) / Wb);
                            for (uint32_t b = b0; b <= b1; ++b) {
                                uint32_t j = s_head[((b * 2654435761u) >> 12) & hmask];
                                while (j != 0xFFFFFFFFu && j != 0xFFFFu) {
                                    if (j < best && arm_accepts<PosT>(x, s_re[j], s_thr[j], k)) best = j;
                                    j = s_next[j];
                                }
                            }
                        }
                        {   // wide arms: branch-free scan of the packed list (increasing arm index)
                            uint32_t wbest = 0xFFFFFFFFu;
                            uint32_t wdx = 0;
                            for (; wdx + 4 <= n_wide; wdx += 4) {
                                const uint32_t a0 = (uint64_t)(PosT)(x - s_wlo[wdx]) < s_ww[wdx] ? wdx : 0xFFFFFFFFu;
                                const uint32_t a1 = (uint64_t)(PosT)(x - s_wlo[wdx + 1]) < s_ww[wdx + 1] ? wdx + 1 : 0xFFFFFFFFu;
                                const uint32_t a2 = (uint64_t)(PosT)(x - s_wlo[wdx + 2]) < s_ww[wdx + 2] ? wdx + 2 : 0xFFFFFFFFu;
                                const uint32_t a3 = (uint64_t)(PosT)(x - s_wlo[wdx + 3]) < s_ww[wdx + 3] ? wdx + 3 : 0xFFFFFFFFu;
                                wbest = min(wbest, min(min(a0, a1), min(a2, a3)));
                            }
                            for (; wdx < n_wide; ++wdx)
                                wbest = min(wbest, (uint64_t)(PosT)(x - s_wlo[wdx]) < s_ww[wdx] ? wdx : 0xFFFFFFFFu);
                            if (valid && wbest != 0xFFFFFFFFu) best = min(best, (uint32_t)s_wide[wbest]);
                        }
                        const int found = best == 0xFFFFFFFFu ? -1 : (int)best;
                        if (valid && found >= 0) atomicMax(&s_pend[found], t + 1u);
                        const bool is_new = valid && found < 0;
                        const unsigned long long m = __ballot(is_new);
                        if (is_new) {
                            const uint32_t d = A + __popcll(m & lt_mask);
                            s_ls[d] = (PosT)i; s_le[d] = (PosT)(i + k); s_rs[d] = x;
                            s_re[d] = (PosT)(x + k);
                            s_gap[d] = step;
                            s_thr[d] = thr0;
                            s_seq[d] = next_seq + (d - A_old);
                            s_pend[d] = 0;
                        }
                        A += __popcll(m);
                    }
                    next_seq += A - A_old;
                    __syncthreads();
                    PROF_STOP(6);
                    PROF_START();
                    for (uint32_t j = lane; j < A_old; j += 64) {
                        const uint32_t pd = s_pend[j];
                        if (pd) {
                            s_pend[j] = 0;
                            const PosT x = from_lds ? s_hits[off + pd - 1u] : P.hits[row + pd - 1u];
                            s_re[j] = (PosT)(x + k);
                            s_le[j] = (PosT)(i + k);
                            s_thr[j] = arm_threshold((uint64_t)(i + k) - (uint64_t)s_ls[j], G);
                            s_gap[j] = 0;
                        } else {
                            s_gap[j] += step;
                        }
                    }
                    __syncthreads();
                    PROF_STOP(7);
                    PROF_START();
                    retire_lds();
                    if (A <= 32) to_regs();
                    PROF_STOP(8);
                }
                // every hit of this probe extended an arm or created one
                fam_open = true;
                maybe_close();
            }
            if (!done) {
                const unsigned long long range = pos >= 64 ? 0ull : ~((1ull << pos) - 1ull);
                const uint32_t q = (uint32_t)__popcll(qm & range);
                if (q) advance_quiet(q);
            }
            __syncthreads();
            g += nbb;
        }
        // arms still alive at the end of the chunk are dropped together with the
        // unflushed family they belong to (src/automaton.rs:201-203)
        if (!done && g_end < chunk_end) {
            // sharded call: the segment is not finished inside the look-ahead window
            if (lane == 0) atomicAdd(&P.ctr[CT_RANOUT], 1ull);
        } else if (!overflow && fam_open)
            emit_records(lane == 0, (PosT)0, (PosT)0, (PosT)0, (PosT)0, kTombstone);
        PROF_FLUSH();
        if (overflow && lane == 0) {
            const unsigned long long at = atomicAdd(P.ovf_count, 1ull);
            if (P.ovf_list) P.ovf_list[at] = g0;
        }
        __syncthreads();
    }
    rec_flush(rec_alloc, P, lane);
    wg_busy(P);
}

// ---------------------------------------------------------------- placement ---
// Every live arm was created or extended by a distinct hit of the last t* processed probes, so
// B = max over probes of (hits of this probe + hits of the previous t* processed probes) bounds
// live arms + new arms from above: the tier whose capacity fits B never overflows (the cascade
// serves the tiers that accept segments above their capacity by an allowance, and the test knobs).
// key = (tier-1) << 29 | (2^29-1 - min(total hits, 2^29-1)): ascending sort = tier, longest first.
struct PlaceParams {
    uint32_t cap[kTiers - 1];   // live-arm capacity of tiers 1..kTiers-1 (0: tier unused); the last takes the rest
    uint32_t sum1;              // segments with more hits than this never go to the one-wave tier
    int force_tier;             // tests: minimum tier for segments with a multi-hit probe
    uint32_t long3;             // > 0: tier 3 is reserved for segments of at least this many probes
    uint32_t long3_big;         // ... or this many, for segments beyond tier 5's capacity
    uint32_t dense6;            // > 0: segments bound for tier 6 by their arms go to tier 3 (when they fit it) with at least this
                                // many hits per processed probe: the dense ones of ANY length on the kernel with a control wave
    uint32_t stats;             // 1 (option debug): hit-probes and hits per tier are tallied (one atomic pair per segment)
    uint32_t barren;            // 1: segments that provably emit nothing are not run at all (segment_is_barren)
    uint2 *seg_info;            // per segment, for cluster_barren_kernel: x = hits, y = probe positions walked | bit 31: the walk saw the
                                // segment's end (not cut short by a sharded call's window); null: not recorded
    uint32_t k, step, G;        // (of the call: for that bound)
    unsigned long long M;
    uint32_t dense3;            // > 0: ... and only the DENSE ones (at least this many hits per processed probe on average:
                                // tandem arrays); the sparse long ones (a chromosome against its homologue: a few hits per
                                // probe, mostly run by one wave alone) go to tier 6's kernel -- set when tier 3 runs the
                                // kernel with a control wave, whose step costs the same whatever the probe holds
};

// A segment that cannot emit anything need not run.  A family holds the arms whose RIGHT segment is at least M long
// (src/automaton.rs:186-196), and a right segment only grows when its arm is extended (:136-143): to m.end = x + k with
// x < right.end + threshold (:68-70), i.e. by at most threshold + k - 1 per extension, at most once per hit-probe.  With
// H hit-probes in the segment an arm born at the first is extended at most H - 1 times, and its threshold
// max(G, len(left) / 10) is at most max(G, (span * step + k) / 10) while the segment spans `span` probe positions
// (len(left) = i + k - left.start <= span * step + k).  So
//     k + (H - 1) * (thr_max + k - 1) < M   =>   no arm of the segment ever reaches M: it emits no family, and -- segments being
// independent automaton instances -- leaving it out changes nothing.  At the defaults (k = 20, G = 120,
// M = 1000) that is every segment of up to 8 hit-probes: most of the million-odd tiny segments of a genome-sized pass.
// (Not when the walk was cut short by the end of a sharded call's window: the segment may go on beyond it.)
constexpr int kTierBarren = kTiers + 1;
__device__ inline bool segment_is_barren(const PlaceParams &pp, uint32_t n_hit, uint32_t span_probes) {
    if (!pp.barren || n_hit == 0u) return false;
    const unsigned long long len_left_max = (unsigned long long)span_probes * pp.step + pp.k;
    const unsigned long long thr_max = max((unsigned long long)pp.G, len_left_max / 10ull);
    return (unsigned long long)pp.k + (unsigned long long)(n_hit - 1u) * (thr_max + pp.k - 1ull) < pp.M;
}

// Sort key of a segment inside its tier (ascending = launch order).  Heavy tiers: longest first,
// so that the serial chains start early.  Tier 1 holds over a million mostly tiny segments whose
// cost is the latency of fetching their probe rows: apart from its few long ones (first, by
// length) they run in genome order, so that the segments in flight at any time share cache lines
// and pages of p_filt / row_off / hits.
__device__ inline uint32_t placement_key(int tier, unsigned long long sum, uint32_t g0) {
    uint32_t low;
    if (tier == 1) {
        low = sum >= 1024ull ? 1023u - (uint32_t)min(sum >> 5, 1023ull) : 1024u + (g0 >> 4);
    } else {
        const uint32_t s29 = sum > 0x1FFFFFFFull ? 0x1FFFFFFFu : (uint32_t)sum;
        low = 0x1FFFFFFFu - s29;
    }
    return (((uint32_t)tier - 1u) << 29) | low;
}

// Tier of a segment from its live-arm bound, hit total and processed-probe count.  With long3 set,
// tier 3 (lowest per-probe latency, one workgroup of 1024 threads per CU) only takes the long
// segments whose serial chain is the critical path of a pass; everything else goes by capacity.
__device__ inline int place_tier(uint32_t bound, unsigned long long sum, uint32_t n_probes, const PlaceParams &pp) {
    if (bound <= pp.cap[0] && sum <= pp.sum1) return 1;
    // tier 6 pays ~3x tier 3's time per probe (many arms per thread): its segments count as long
    // from a quarter of the threshold on
    if (pp.long3 && bound <= pp.cap[2] &&
        (n_probes >= pp.long3 || (bound > pp.cap[4] && n_probes >= pp.long3_big))) {
        if (!pp.dense3 || sum >= (unsigned long long)pp.dense3 * n_probes || bound > pp.cap[5]) return 3;
        return 6;  // (the long segments that are not dense: mostly run by one wave alone, on K6's solo probes)
    }
    for (int t = 2; t < kTiers; ++t) {
        if (t == 3 && pp.long3) continue;
        if (bound <= pp.cap[t - 1]) {
            if (t == 6 && pp.dense6 && bound <= pp.cap[2] && sum >= (unsigned long long)pp.dense6 * n_probes) return 3;
            return t;
        }
    }
    return kTiers;
}

// The placement walk reads the per-probe hit counts only -- no hit accesses at all.
//
// Most segments are a handful of probes long (3.4 M segments per GRCh38-shaped step, a dozen probes each), and a
// wave per segment spends its time on the per-segment chain of dependent loads.  So the first kernel walks ONE
// SEGMENT PER LANE, 64 segments per wave side by side, for at most kLaneWalk probes; the few segments that are
// longer go to a list and the wave-per-segment kernel below (64 probes per round trip) finishes them.
constexpr uint32_t kLaneWalk = 256;

__global__ __launch_bounds__(64) void seg_stats_lanes_kernel(RunParams rp, const uint32_t *__restrict__ p_filt,
                                                             const uint32_t *__restrict__ seg_list,
                                                             const unsigned long long *__restrict__ n_seg_ptr,
                                                             uint32_t *__restrict__ keys,
                                                             uint32_t *__restrict__ vals, PlaceParams pp,
                                                             uint32_t *__restrict__ long_list,
                                                             unsigned long long *__restrict__ ctr) {
    __shared__ uint32_t s_ring[65 * 64];  // [slot][lane]: hit counts of the last TW + 1 processed probes
    const uint32_t lane = threadIdx.x;
    const uint64_t n_seg = *n_seg_ptr;
    const uint32_t RW = min(rp.tstar, 64u) + 1u;
    for (uint64_t base = (uint64_t)blockIdx.x * 64u; base < n_seg; base += (uint64_t)gridDim.x * 64u) {
        const uint64_t sidx = base + lane;
        const bool have = sidx < n_seg;
        uint32_t g0 = 0, g_end = 0;
        bool window_cut = false;  // the walk may end at the end of a sharded call's window instead of the chunk's
        if (have) {
            g0 = seg_list[sidx];
            const uint32_t chunk_end = rp.ch.pbase[chunk_of(rp.ch, g0) + 1];
            g_end = min(chunk_end, rp.win_end(g0));
            window_cut = g_end < chunk_end;
        }
        for (uint32_t r = 0; r < RW; ++r) s_ring[r * 64u + lane] = 0;
        uint32_t quiet = 0, mx = 0, bound = 0, n_probes = 0, wsum = 0, head = 0, steps = 0, g = g0, n_hit = 0;
        unsigned long long sum = 0;
        bool done = !have, by_quiet = false;
        while (!done && g < g_end && steps < kLaneWalk) {
            const uint32_t f = p_filt[g];
            ++g;
            ++steps;
            const bool hit = f >= 1u && f < kPending;
            if (!hit && f != 0u) continue;  // skipped probes (N, cardinality) are not processed
            if (!hit) {
                if (++quiet >= rp.tstar) {  // t* quiet probes: every arm has been retired, the segment is over
                    done = true;
                    by_quiet = true;
                    break;
                }
            } else {
                quiet = 0;
            }
            const uint32_t v = hit ? f : 0u;
            ++n_probes;
            n_hit += hit ? 1u : 0u;
            const uint32_t at = head * 64u + lane;
            wsum += v - s_ring[at];
            s_ring[at] = v;
            head = head + 1u == RW ? 0u : head + 1u;
            bound = max(bound, wsum);
            mx = max(mx, v);
            sum += v;
        }
        if (g >= g_end) done = true;
        if (have && done) {
            if (rp.tstar > 64u) bound = 0xFFFFFFFFu;  // no estimate for huge gap settings: largest tier
            int tier = place_tier(bound, sum, n_probes, pp);
            if (mx > 1 && pp.force_tier > tier) tier = min(pp.force_tier, kTiers);
            if ((by_quiet || !window_cut) && segment_is_barren(pp, n_hit, g - g0)) tier = kTierBarren;
            keys[sidx] = placement_key(tier, sum, g0);
            vals[sidx] = g0;
            if (pp.seg_info)
                pp.seg_info[sidx] = make_uint2(sum > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)sum,
                                               (g - g0) | ((by_quiet || !window_cut) ? 0x80000000u : 0u));
            if (pp.stats) {
                atomicAdd(&ctr[CT_TPROBES1 + tier - 1], (unsigned long long)n_hit);
                atomicAdd(&ctr[CT_THITS1 + tier - 1], sum);
            }
        }
        const unsigned long long lm = __ballot(have && !done);
        if (lm) {
            const int leader = __ffsll((long long)lm) - 1;
            unsigned long long at = 0;
            if ((int)lane == leader) at = atomicAdd(&ctr[CT_LONGSEG], (unsigned long long)__popcll(lm));
            at = __shfl(at, leader);
            if (have && !done) long_list[at + __popcll(lm & ((1ull << lane) - 1ull))] = (uint32_t)sidx;
        }
    }
}

// One wave per segment: the entries idx_list[0 .. *n_ptr) of the segment list (all of it when idx_list is null).
__global__ __launch_bounds__(64) void seg_stats_kernel(RunParams rp, const uint32_t *__restrict__ p_filt,
                                                       const uint32_t *__restrict__ seg_list,
                                                       const unsigned long long *__restrict__ n_ptr,
                                                       const uint32_t *__restrict__ idx_list,
                                                       uint32_t *__restrict__ keys,
                                                       uint32_t *__restrict__ vals, PlaceParams pp,
                                                       unsigned long long *__restrict__ ctr) {
    __shared__ uint32_t s_ext[64 + 64];  // [0,TW): hit counts of the previous processed probes
    const int lane = threadIdx.x;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const uint64_t n_items = *n_ptr;
    const uint32_t TW = min(rp.tstar, 64u);
    for (uint64_t item = blockIdx.x; item < n_items; item += gridDim.x) {
        const uint64_t sidx = idx_list ? (uint64_t)idx_list[item] : item;
        const uint32_t g0 = seg_list[sidx];
        const int c = chunk_of_uniform(rp.ch, g0);
        const uint32_t g_end = min(rp.ch.pbase[c + 1], rp.win_end(g0));
        const bool window_cut = g_end < rp.ch.pbase[c + 1];
        uint32_t quiet = 0, mx = 0, bound = 0, n_probes = 0, n_hit = 0, g_stop = g_end, g_after_hit = g0 + 1u;
        unsigned long long sum = 0;
        bool done = false;
        s_ext[lane] = 0;
        lds_barrier();
        // the next batch's hit counts are in flight while this one is processed: a long segment's walk is a chain
        // of dependent round trips (64 probes each) and sets the duration of the whole launch
        uint32_t f_next = g0 + (uint32_t)lane < g_end ? p_filt[g0 + lane] : kSkipN;
        for (uint32_t g = g0; g < g_end && !done; g += 64) {
            const uint32_t f = f_next;
            f_next = g + 64u + (uint32_t)lane < g_end ? p_filt[g + 64u + lane] : kSkipN;
            const unsigned long long hm = __ballot(f >= 1u && f < kPending);
            const unsigned long long qm = __ballot(f == 0u);
            if (!(hm | qm)) continue;  // 64 skipped probes (cardinality, N): nothing is processed, nothing changes
            unsigned long long live = ~0ull;  // probes of this batch that belong to the segment
            // The segment ends with the t*-th quiet probe in a row (skipped probes neither count nor interrupt,
            // a hit restarts the count).  Every lane prices the quiet run that ends at its own probe -- no walk
            // from hit to hit: in a dense array with interleaved quiet probes that scalar walk, 40-odd dependent
            // steps per batch, was the whole cost of placing the longest segment.
            {
                const unsigned long long hb = hm & lt_mask;  // hits before this lane
                const int lh = hb ? 63 - __clzll((long long)hb) : -1;
                const unsigned long long above_lh = lh < 0 ? ~0ull : (lh == 63 ? 0ull : ~((2ull << lh) - 1ull));
                const uint32_t run = (uint32_t)__popcll(qm & (lt_mask | (1ull << lane)) & above_lh) + (lh < 0 ? quiet : 0u);
                const unsigned long long term = __ballot(((qm >> lane) & 1ull) && run >= rp.tstar);
                if (term) {
                    // the run that reaches t* starts behind the hit at lh (of the terminating lane): only the
                    // probes up to that hit still belong to the segment
                    const uint32_t pos = (uint32_t)(__shfl(lh, __ffsll((long long)term) - 1) + 1);
                    done = true;
                    // (exactly behind the segment's last hit-probe: the barren tests want an upper bound of the span, the
                    // cutting into ranges must never place a cut on a hit-probe of the NEXT segment)
                    g_stop = pos ? g + pos : g_after_hit;
                    live = (1ull << pos) - 1ull;  // pos <= 63
                } else {
                    const int last = hm ? 63 - __clzll((long long)hm) : -1;
                    if (last >= 0) g_after_hit = g + (uint32_t)last + 1u;
                    const unsigned long long above = last < 0 ? ~0ull : (last == 63 ? 0ull : ~((2ull << last) - 1ull));
                    quiet = (uint32_t)__popcll(qm & above) + (last < 0 ? quiet : 0u);
                }
            }
            const unsigned long long procm = (hm | qm) & live;
            const bool proc = (procm >> lane) & 1ull;
            const uint32_t v = ((hm & live) >> lane) & 1ull ? f : 0u;
            const uint32_t r = (uint32_t)__popcll(procm & lt_mask);
            const uint32_t n_proc = (uint32_t)__popcll(procm);
            n_probes += n_proc;
            n_hit += (uint32_t)__popcll(hm & live);
            if (proc) s_ext[TW + r] = v;
            lds_barrier();
            uint32_t wsum = 0;
            if (proc)
                for (uint32_t d = 0; d <= TW; ++d) wsum += s_ext[TW + r - d];
            uint32_t m = v, wm = wsum;
            unsigned long long sv = v;
            for (int off = 32; off > 0; off >>= 1) {
                sv += __shfl_down(sv, off);
                m = max(m, (uint32_t)__shfl_down(m, off));
                wm = max(wm, (uint32_t)__shfl_down(wm, off));
            }
            sum += __shfl(sv, 0);
            mx = max(mx, (uint32_t)__shfl(m, 0));
            bound = max(bound, (uint32_t)__shfl(wm, 0));
            const uint32_t keep = (uint32_t)lane < TW ? s_ext[n_proc + lane] : 0u;
            lds_barrier();
            if ((uint32_t)lane < TW) s_ext[lane] = keep;
            lds_barrier();
        }
        if (rp.tstar > 64u) bound = 0xFFFFFFFFu;
        if (lane == 0) {
            int tier = place_tier(bound, sum, n_probes, pp);
            if (mx > 1 && pp.force_tier > tier) tier = min(pp.force_tier, kTiers);
            if ((done || !window_cut) && segment_is_barren(pp, n_hit, min(g_stop, g_end) - g0)) tier = kTierBarren;
            keys[sidx] = placement_key(tier, sum, g0);
            vals[sidx] = g0;
            if (pp.seg_info)
                pp.seg_info[sidx] = make_uint2(sum > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)sum,
                                               (min(g_stop, g_end) - g0) | ((done || !window_cut) ? 0x80000000u : 0u));
            if (pp.stats) {
                atomicAdd(&ctr[CT_TPROBES1 + tier - 1], (unsigned long long)n_hit);
                atomicAdd(&ctr[CT_THITS1 + tier - 1], sum);
            }
        }
        lds_barrier();
    }
}

// Barren by POSITION.  An arm's right segment is built from a chain of hits x_0 < x_1 < ... (one per hit-probe at most) with
// x_{j+1} < x_j + k + threshold (src/automaton.rs:68-70 with right.end = x_j + k), and it is reported only when
// x_m + k - x_0 >= M.  With buckets of w = k + thr_max positions (thr_max: the largest threshold any arm of the segment can
// have, as in segment_is_barren) every link of a chain stays in its bucket or moves to the next one, so an arm that reaches M
// leaves a RUN of at least floor((M - k) / w) + 1 consecutive occupied buckets among the segment's hits.  No such run =>
// the segment emits nothing.  That is the fate of the bursts of interspersed repeats, which are most of the extension's
// work at genome scale: a probe inside a repeat hits a hundred other copies, every copy collects a few hundred bases of
// hits over the few dozen probes of the burst, the copies are kilobases apart -- hundreds of arms per probe and not one
// duplication.  Tandem arrays and real duplications keep their runs and are run as before.
// One WAVE per segment: the hits' buckets set bits of a hashed LDS bitmap (BITS bits; a collision can only make a bucket look
// occupied that is not, i.e. a run look longer -- the test stays sound, it only proves a little less), then every hit
// asks whether the need - 1 buckets behind its own are all occupied.  Order-free, no barrier between waves, a few KB of
// LDS per wave (the continuation filter of round 2 answered a related question probe by probe, a wave per segment with
// barriers per probe, and cost more than it saved).  Segments of more than max_hits hits are left alone.
template <class PosT, int BITS>
__global__ __launch_bounds__(64) void cluster_barren_kernel(RunParams rp, PlaceParams pp,
                                                            const unsigned long long *__restrict__ row_off,
                                                            const PosT *__restrict__ hits,
                                                            const uint32_t *__restrict__ seg_list,
                                                            const unsigned long long *__restrict__ n_seg_ptr,
                                                            uint32_t *__restrict__ keys, uint32_t min_hits, uint32_t max_hits,
                                                            unsigned long long *__restrict__ cursor,
                                                            unsigned long long *__restrict__ ctr) {
    static_assert((BITS & (BITS - 1)) == 0 && BITS >= 2048, "bitmap size");
    constexpr uint32_t kWords = BITS / 32, kShift = 32 - __builtin_ctz((unsigned)BITS);
    __shared__ uint32_t s_bits[kWords];
    const uint32_t lane = threadIdx.x;
    const uint64_t n_seg = *n_seg_ptr;
    uint32_t n_barren = 0;
    auto bit_of = [&](uint32_t b) { return (b * 2654435761u) >> kShift; };
    for (;;) {
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(cursor, 64ull);
        base = lane_of(base, 0u);
        if (base >= n_seg) break;
        // the 64 segments of the batch side by side: key, hit total, where its rows are (most are not this launch's: already
        // barren, another size class, cut by a sharded window); the candidates are then walked one after the other, their
        // set-up already in registers
        const uint64_t sj = base + lane;
        bool cand = false;
        uint32_t key_l = 0, span_l = 0;
        unsigned long long lo_l = 0, hi_l = 0;
        if (sj < n_seg) {
            key_l = keys[sj];
            const uint2 info = pp.seg_info[sj];
            span_l = info.y & 0x7FFFFFFFu;
            cand = (key_l >> 29) < (uint32_t)kTiers && (info.y >> 31) && info.x >= min_hits && info.x <= max_hits;
            if (cand) {
                const uint32_t g0 = seg_list[sj];
                lo_l = row_off[g0];
                hi_l = row_off[g0 + span_l];
                cand = hi_l > lo_l && hi_l - lo_l <= (unsigned long long)max_hits;
            }
        }
        unsigned long long todo = __ballot(cand);
        while (todo) {
            const uint32_t l = (uint32_t)(__ffsll((long long)todo) - 1);
            todo &= todo - 1ull;
            const uint32_t key = lane_of(key_l, l), span = lane_of(span_l, l);
            const unsigned long long lo = lane_of(lo_l, l);
            const uint32_t n = (uint32_t)(lane_of(hi_l, l) - lo);
            const unsigned long long len_left_max = (unsigned long long)span * pp.step + pp.k;
            const unsigned long long thr_max = max((unsigned long long)pp.G, len_left_max / 10ull);
            const unsigned long long w = thr_max + pp.k;
            const unsigned long long need = (pp.M - pp.k) / w + 1ull;  // consecutive occupied buckets an emitting arm leaves
            if (need < 2ull || need > 64ull) continue;                 // (one bucket proves nothing; M > k: the host has checked)
            // exact floor(x / w) through a double quotient and one correction (x < 2^53; a 64-bit integer division per hit
            // would be a hundred instructions)
            const double inv_w = 1.0 / (double)w;
            auto bucket_of = [&](unsigned long long x) -> uint32_t {
                unsigned long long q = (unsigned long long)((double)x * inv_w);
                const long long r = (long long)(x - q * w);
                q += r >= (long long)w ? 1ull : 0ull;
                q -= r < 0 ? 1ull : 0ull;
                return (uint32_t)q;
            };
            for (uint32_t j = lane; j < kWords; j += 64u) s_bits[j] = 0u;
            __builtin_amdgcn_wave_barrier();
            __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // (eight independent loads per lane in flight, then the bits: a load-set loop would pay one round trip per pass)
            for (uint32_t j0 = 0; j0 < n; j0 += 512u) {
                PosT xv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const uint32_t j = j0 + (uint32_t)u * 64u + lane;
                    xv[u] = j < n ? hits[lo + j] : (PosT)0;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const uint32_t j = j0 + (uint32_t)u * 64u + lane;
                    if (j < n) {
                        const uint32_t bi = bit_of(bucket_of((unsigned long long)xv[u]));
                        atomicOr(&s_bits[bi >> 5], 1u << (bi & 31u));
                    }
                }
            }
            __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            bool found = false;
            for (uint32_t j0 = 0; j0 < n && !found; j0 += 512u) {
                PosT xv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const uint32_t j = j0 + (uint32_t)u * 64u + lane;
                    xv[u] = j < n ? hits[lo + j] : (PosT)0;
                }
                bool mine = false;
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const uint32_t j = j0 + (uint32_t)u * 64u + lane;
                    if (j < n) {
                        const uint32_t b = bucket_of((unsigned long long)xv[u]);
                        bool all = true;
                        for (uint32_t d = 1; d < (uint32_t)need && all; ++d) {
                            const uint32_t bi = bit_of(b + d);
                            all = (s_bits[bi >> 5] >> (bi & 31u)) & 1u;
                        }
                        mine = mine || all;
                    }
                }
                found = __ballot(mine) != 0ull;
            }
            __builtin_amdgcn_wave_barrier();
            if (!found) {
                if (lane == 0) keys[base + l] = (key & 0x1FFFFFFFu) | ((uint32_t)(kTierBarren - 1) << 29);
                ++n_barren;
            }
        }
    }
    if (lane == 0 && n_barren) atomicAdd(&ctr[CT_CLUSTER_BARREN], (unsigned long long)n_barren);
}

// ---- long segments as ranges that run side by side (option split) ---------------------------------------------------------
// The probes of a segment are strictly serial, and the longest segment of a pass is the floor of its extension.  But the arm
// list a tandem array leaves behind at some probe c is, some way into the array, a function of the last few thousand probes
// only: a run that starts with NO arm `warm` probes in front of c holds, at c, exactly the arms (and the family state) of the
// run that started at the segment's first probe -- when it does.  Whether it does is CHECKED, never assumed: the range in
// front of the cut and a warm-up that stops at the cut both write out what they hold there (validate_cuts_kernel compares),
// where a cut fails the ranges in front of it stand, the records of those behind it are dropped and the rest of the segment
// runs as one more run from the last checked state (the whole segment again when its first cut fails).
//   plan_ranges_kernel     one thread per segment: the long ones of the long-shape tiers (3, 6) are taken off their tier's
//                          list and cut at the first hit-probe at or behind every range_len-th probe
//   validate_cuts_kernel   one workgroup per cut: same arms (by creation number, every field), same family state
//   fixup_records_kernel   one thread per record slot: family ordinals of a range + the flushes of the ranges before it;
//                          records of a segment that failed -> void
// Range size by budget (option split_len = 0): every run holds a compute unit, so the number of runs is what the cutting
// may cost.  A segment of `span` probe positions gets round(span / C) ranges of one length.  split_tally_kernel counts, for
// each candidate C, the runs that cutting every eligible segment would make; split_pick_kernel takes the SMALLEST C whose
// count fits the budget (a small job gets small ranges -- its few long segments are its whole extension --, a genome-sized
// one large ranges); warm-up C positions within [2 048, 6 144]; a segment is cut when it is at least 2 C and at least 3
// warm-ups long.  The choice depends on the segments only: the same in every call over the same input and settings.
constexpr int kSplitCand = 12;
__device__ inline uint32_t split_len_of(int c) {
    constexpr uint32_t t[kSplitCand] = {2048, 3072, 4096, 6144, 8192, 12288, 16384, 24576, 32768, 49152, 65536, 98304};
    return t[c];
}
// (half a range was not enough at the middle sizes: with ranges of 6 144 probes -- what a shard of a genome-sized call gets --
// and 3 072 of warm-up 20 of 44 cut segments had a cut that did not hold; with 6 144 every cut of the unsharded call holds)
__device__ inline uint32_t split_warm_of(uint32_t C) { return min(max(C, 2048u), 6144u); }
struct SplitChoice {
    uint32_t range_len, warm, min_span, pad;
    unsigned long long runs[kSplitCand];
};
__device__ inline bool split_eligible(const RunParams &rp, uint32_t key, uint2 info) {
    const uint32_t tier = (key >> 29) + 1u;
    // (tiers 1 and 2 are one-wave kernels: a wave on its own passes a sparse probe several times faster than the long shape's
    // sixteen waves and their barrier -- cutting a yeast-sized input's longest tier-2 segment in two doubled its step; cutting
    // the one-wave segments of 12 288 probe positions and more left a GRCh38-sized step and its shards where they were:
    // DESIGN_HISTORY.md; tier 7's arms do not fit the long shape)
    if (tier < 3u || tier > 6u) return false;
    const uint32_t span = info.y & 0x7FFFFFFFu;
    if (!(info.y >> 31) || span < 128u) return false;  // (cut short by a shard window: left alone)
    // (creation numbers of a run: needle offset relative to the segment's first probe << 10 | hit index)
    return (unsigned long long)span * (unsigned long long)rp.step < (1ull << 22) - 2ull;
}
__global__ __launch_bounds__(256) void split_tally_kernel(RunParams rp, const unsigned long long *__restrict__ n_seg_ptr,
                                                         const uint32_t *__restrict__ keys, const uint2 *__restrict__ seg_info,
                                                         SplitChoice *__restrict__ choice) {
    const unsigned long long sj = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (sj >= *n_seg_ptr) return;
    const uint2 info = seg_info[sj];
    const uint32_t span = info.y & 0x7FFFFFFFu;
    if (span < 3u * 2048u || !split_eligible(rp, keys[sj], info)) return;
    const unsigned long long cost = (unsigned long long)span;
    for (int c = 0; c < kSplitCand; ++c) {
        const unsigned long long C = split_len_of(c);
        if (cost < 2ull * C || span < 3u * split_warm_of((uint32_t)C)) break;
        atomicAdd(&choice->runs[c], max(2ull, (cost + C / 2ull) / C));
    }
}
__global__ void split_pick_kernel(SplitChoice *choice, uint32_t budget) {
    int pick = kSplitCand - 1;
    for (int c = 0; c < kSplitCand; ++c)
        if (choice->runs[c] <= (unsigned long long)budget) {
            pick = c;
            break;
        }
    choice->range_len = split_len_of(pick);
    choice->warm = split_warm_of(split_len_of(pick));  // (E. coli-sized inputs: cuts with 1 024 probes of warm-up did not hold)
    choice->min_span = 3u * choice->warm;               // (two ranges of a shorter segment are each nearly the segment)
}

constexpr uint32_t kSplitBlockedMax = 2048;  // verdicts of earlier calls handed to one call
// what an earlier call of the index found out about a segment a cut of which did not hold
struct BlockedSeg {
    uint32_t g0;         // the segment (first probe, in this call's numbering)
    uint32_t range_len;  // at which range length (the verdict means nothing at another)
    uint32_t allowed;    // how many of its cuts, counted from the segment's start, are planned again (kAllCuts: all of them)
    uint32_t warm;       // the warm-up its ranges get from now on (0: the call's)
};
constexpr uint32_t kAllCuts = 0xFFFFFFFFu;
struct SplitParams {
    uint32_t range_len, warm, min_span;   // size of a range (probe positions; range_len = 0: cost units, as split_pick_kernel chose),
                                          // warm-up probes in front of a cut, shortest segment that is cut
    uint32_t max_runs, max_cuts, max_splits;
    uint32_t n_blocked;
    const BlockedSeg *blocked;            // segments a cut of which did not hold in an earlier call of the index
};
__global__ __launch_bounds__(256) void plan_ranges_kernel(RunParams rp, SplitParams sp, const uint32_t *__restrict__ p_filt,
                                                         const uint32_t *__restrict__ seg_list,
                                                         const unsigned long long *__restrict__ n_seg_ptr,
                                                         uint32_t *__restrict__ keys, const uint2 *__restrict__ seg_info,
                                                         unsigned long long *__restrict__ hdr,  // 0 runs, 1 cuts, 2 split segments
                                                         RangeRun *__restrict__ runs, uint2 *__restrict__ cuts,
                                                         SplitSeg *__restrict__ splits, const SplitChoice *__restrict__ choice) {
    const unsigned long long sj = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (sj >= *n_seg_ptr) return;
    if (!sp.range_len) {  // (hdr[3] is the runs' work cursor: hdr[4] tells the host which length was used)
        sp.range_len = choice->range_len;
        sp.warm = choice->warm;
        sp.min_span = max(sp.min_span, choice->min_span);
        if (sj == 0) hdr[4] = sp.range_len;
    } else if (sj == 0) {
        hdr[4] = sp.range_len;
    }
    const uint32_t key = keys[sj];
    const uint2 info = seg_info[sj];
    const uint32_t span = info.y & 0x7FFFFFFFu;
    if (span < sp.min_span || !split_eligible(rp, key, info)) return;
    const uint32_t g0 = seg_list[sj];
    // a segment a cut of which did not hold in an earlier call: its ranges start as far in front of their cuts as the oldest
    // arm at the failed cut asked for -- or, where that is further than a range is worth, only the cuts that held are planned again
    uint32_t warm = sp.warm;
    uint32_t n_cut_max = 0xFFFFFFFFu;
    for (uint32_t b = 0; b < sp.n_blocked; ++b) {
        const BlockedSeg v = sp.blocked[b];
        if (v.g0 == g0 && v.range_len == sp.range_len) {
            n_cut_max = min(n_cut_max, v.allowed);
            warm = max(warm, v.warm);
        }
    }
    // ranges of about range_len probe positions each, at least two, their cuts at equal shares of the segment; a segment whose
    // cuts held only up to some point in an earlier call keeps those cuts (same places) and runs the rest as its last range
    // (cuts at equal shares of positions + hits / w instead were measured at GRCh38 size and did not move the step: DESIGN_HISTORY.md)
    const unsigned long long total = span, unit = sp.range_len;
    if (total < 2ull * unit) return;
    const uint32_t n_r_all = (uint32_t)min(1024ull, max(2ull, (total + unit / 2ull) / unit));
    const uint32_t n_r = min(n_r_all, n_cut_max == kAllCuts ? n_r_all : n_cut_max + 1u);
    if (n_r < 2u) return;
    auto cut_of = [&](uint32_t j) -> uint32_t {  // first hit-probe at or behind the j-th share of the cost (0: none)
        uint32_t c = g0 + (uint32_t)(total / n_r_all * j);
        while (c < g0 + span) {
            const uint32_t f = p_filt[c];
            if (f >= 1u && f < kPending) return c;
            ++c;
        }
        return 0u;
    };
    {   // (every cut behind its predecessor, the first behind the segment's start)
        uint32_t before = g0;
        for (uint32_t j = 1; j < n_r; ++j) {
            const uint32_t c = cut_of(j);
            if (c <= before) return;
            before = c;
        }
    }
    const uint32_t n_runs = n_r;
    const uint32_t run_base = (uint32_t)atomicAdd(&hdr[0], (unsigned long long)n_runs);
    if (run_base + n_runs > sp.max_runs) {
        atomicAdd(&hdr[0], 0ull - (unsigned long long)n_runs);
        return;
    }
    const uint32_t cut_base = (uint32_t)atomicAdd(&hdr[1], (unsigned long long)(n_r - 1u));
    const uint32_t split = (uint32_t)atomicAdd(&hdr[2], 1ull);
    if (cut_base + n_r - 1u > sp.max_cuts || split >= sp.max_splits) {  // (sized together with max_runs: does not happen)
        atomicAdd(&hdr[0], 0ull - (unsigned long long)n_runs);
        atomicAdd(&hdr[1], 0ull - (unsigned long long)(n_r - 1u));
        atomicAdd(&hdr[2], 0ull - 1ull);
        return;
    }
    keys[sj] = (key & 0x1FFFFFFFu) | ((uint32_t)(kTierBarren - 1) << 29);  // off its tier's list
    splits[split] = SplitSeg{g0, run_base, n_r, cut_base, span, info.x, (key >> 29) + 1u, warm};
    uint32_t c_prev = g0;
    for (uint32_t j = 0; j < n_r; ++j) {
        const uint32_t c_next = j + 1u < n_r ? cut_of(j + 1u) : 0xFFFFFFFFu;
        RangeRun r{};
        r.g_begin = (j == 0u || c_prev - g0 <= warm) ? g0 : c_prev - warm;
        r.g_stop = c_next;
        r.g_seg0 = g0;
        r.emit_from = c_prev;
        r.flags = j + 1u == n_r ? kRunLast : 0u;
        r.split = split;
        runs[run_base + j] = r;
        // (what range j - 1 holds when it stops at this range's cut against what this range holds when it reaches it)
        if (j) cuts[cut_base + j - 1u] = make_uint2(run_base + j - 1u, run_base + j);
        c_prev = c_next;
    }
}

// (option debug = 2: per tier the placed segment with the most probe positions and the one with the most hits that was NOT cut
// -- top[2 t] and top[2 t + 1] = figure << 32 | segment)
__global__ __launch_bounds__(256) void top_uncut_kernel(const unsigned long long *__restrict__ n_seg_ptr, const uint32_t *__restrict__ keys,
                                                       const uint2 *__restrict__ seg_info, unsigned long long *__restrict__ top) {
    const unsigned long long sj = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (sj >= *n_seg_ptr) return;
    const uint32_t tier = keys[sj] >> 29;
    if (tier >= (uint32_t)kTiers) return;
    const uint2 info = seg_info[sj];
    atomicMax(&top[2u * tier], (unsigned long long)(info.y & 0x7FFFFFFFu) << 32 | sj);
    atomicMax(&top[2u * tier + 1u], (unsigned long long)info.x << 32 | sj);
}

template <uint32_t W>  // (words per dumped arm: kDumpWords)
__global__ __launch_bounds__(256) void validate_cuts_kernel(const uint2 *__restrict__ cuts, const uint32_t *__restrict__ run_meta,
                                                           const uint32_t *__restrict__ run_dump, uint32_t *__restrict__ cut_ok) {
    constexpr uint32_t Q = W / 4u;  // 16-byte quarters of an arm
    constexpr uint32_t kSlots = 16384;  // > 3 x kRunDumpCap
    __shared__ uint32_t s_tab[kSlots];
    __shared__ uint32_t s_ok, s_oldest;
    const uint2 cut = cuts[blockIdx.x];
    // (x: the run in front of the cut, its end state; y: the run behind it, its state at the cut)
    const uint32_t *ma = run_meta + (size_t)cut.x * 16, *mb = run_meta + (size_t)cut.y * 16 + 8;
    const uint32_t n = ma[0];
    const bool meta_ok = n == mb[0] && n <= kRunDumpCap && ma[2] == mb[2] && ma[3] == mb[3] && !ma[4] && !run_meta[(size_t)cut.y * 16 + 4];
    for (uint32_t j = threadIdx.x; j < kSlots; j += blockDim.x) s_tab[j] = 0u;
    if (threadIdx.x == 0) {
        s_ok = meta_ok ? 1u : 0u;
        s_oldest = 0xFFFFFFFFu;
    }
    __syncthreads();
    {   // the oldest arm the range in front of the cut holds there (its creation number: where in the segment it was born): a
        // run that is to hold it at the cut must start in front of that probe -- what the host sizes the next warm-up by
        const uint4 *da = reinterpret_cast<const uint4 *>(run_dump + (size_t)cut.x * 2 * kRunDumpCap * W);
        uint32_t oldest = 0xFFFFFFFFu;
        for (uint32_t j = threadIdx.x; j < min(n, kRunDumpCap); j += blockDim.x) oldest = min(oldest, da[Q * j].x);
        if (oldest != 0xFFFFFFFFu) atomicMin(&s_oldest, oldest);
    }
    if (meta_ok) {
        const uint4 *da = reinterpret_cast<const uint4 *>(run_dump + (size_t)cut.x * 2 * kRunDumpCap * W);
        const uint4 *db = reinterpret_cast<const uint4 *>(run_dump + ((size_t)cut.y * 2 + 1) * kRunDumpCap * W);
        for (uint32_t j = threadIdx.x; j < n; j += blockDim.x) {
            uint32_t h = (da[Q * j].x * 2654435761u) >> 18;
            while (atomicCAS(&s_tab[h], 0u, j + 1u) != 0u) h = (h + 1u) & (kSlots - 1u);
        }
        __syncthreads();
        bool ok = true;
        for (uint32_t j = threadIdx.x; j < n; j += blockDim.x) {
            const uint4 b0 = db[Q * j];
            uint32_t h = (b0.x * 2654435761u) >> 18;
            bool found = false;
            for (uint32_t probe = 0; probe < kSlots; ++probe) {
                const uint32_t e = s_tab[h];
                if (!e) break;
                const uint4 a0 = da[Q * (e - 1u)];
                if (a0.x == b0.x) {  // (creation numbers are unique within a dump)
                    found = a0.y == b0.y && a0.z == b0.z && a0.w == b0.w;
                    for (uint32_t q = 1; q < Q; ++q) {
                        const uint4 a = da[Q * (e - 1u) + q], b = db[Q * j + q];
                        found = found && a.x == b.x && a.y == b.y && a.z == b.z && a.w == b.w;
                    }
                    break;
                }
                h = (h + 1u) & (kSlots - 1u);
            }
            ok = ok && found;
        }
        if (!ok) s_ok = 0u;
    }
    __syncthreads();
    // (bit 0: the cut holds; above it: needle offset, counted from the segment's first probe, at which the oldest arm was born --
    // all ones: no arm)
    if (threadIdx.x == 0) cut_ok[blockIdx.x] = s_ok | ((s_oldest >> 10) << 1);
}

// run_fix[run]: what is added to the family ordinals of the run's records; ~0u: the run's records are dropped; ~0u - 1: not
// decided yet (left as they are for a later pass)
__global__ __launch_bounds__(256) void fixup_records_kernel(SdRec *__restrict__ recs, unsigned long long n, const uint32_t *__restrict__ run_fix) {
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    SdRec &r = recs[i];
    if (r.g_start == kVoidStart || r.pad == 0u) return;
    const uint32_t f = run_fix[r.pad - 1u];
    if (f == 0xFFFFFFFEu) return;
    if (f == 0xFFFFFFFFu) r.g_start = kVoidStart;
    else r.fam_seq += f;
    r.pad = 0u;
}

// tier list lengths from the sorted placement keys (tier-1 = key >> 29): n_t = first index whose
// tier exceeds t, by bisection -- instead of one contended global atomic per segment
__global__ void tier_bounds_kernel(const uint32_t *__restrict__ sorted_keys,
                                   const unsigned long long *__restrict__ n_seg_ptr,
                                   unsigned long long *__restrict__ ctr) {
    const int t = threadIdx.x;  // 0..4
    if (t >= kTiers) return;
    const uint64_t n = *n_seg_ptr;
    auto first_ge = [&](uint32_t tier_idx) {  // first position with (key >> 29) >= tier_idx
        uint64_t lo = 0, hi = n;
        while (lo < hi) {
            const uint64_t mid = (lo + hi) >> 1;
            if ((sorted_keys[mid] >> 29) < tier_idx) lo = mid + 1; else hi = mid;
        }
        return lo;
    };
    ctr[CT_N1 + t] = first_ge((uint32_t)t + 1u) - first_ge((uint32_t)t);
}

// ---------------------------------------------------------------- K4b --------
// Block-cooperative extension kernel: ONE workgroup (NT = 256 or 1024 threads) per segment, for
// the segments whose live-arm bound does not fit the one-wave kernel.  At genome scale these are
// dense-repeat clusters and satellite tails: hundreds of hits per probe, hundreds to thousands of
// live arms, most of them single-hit arms that die t* probes after they were born.
//
// Same results as extend_kernel, different bookkeeping:
//   * arms live in SLOTS; a dead arm's slot goes on a free list and is reused, so there is no
//     order-preserving compaction.  "First matching arm in list order" (src/automaton.rs:67-78)
//     is the accepting arm with the smallest CREATION NUMBER (list order == creation order), so
//     slot order is irrelevant.
//   * per probe: (0) clear hash heads, (1) hash narrow arms by bucket(re) / list wide arms,
//     (2) one thread per hit: two buckets + wide list -> best (creation number, slot),
//     (3) ExtendArm = atomicMax of the hit index on the slot, NewArm = slot from the free list in
//     hit order, (4) apply / age / retire in place.
//   * MODE 2 keeps the arm arrays in an HBM scratch slice per workgroup (up to 16384
//     live arms); the per-probe candidate index stays in LDS.
constexpr int kHeavyThreads = 512;   // heavy tiers (1024 threads would cap VGPRs at 128 -> spills)
constexpr int kMidThreads = 256;     // mid tier: 4 waves per segment, several workgroups per CU
constexpr uint32_t kNoSeq = 0xFFFFFFFFu;  // s_seq value of an empty slot

// MODE 0: every arm field in LDS.  MODE 1 ("hybrid"): everything a probe reads or updates (ls, le,
// re, thr, seq, gap, pend) in LDS; rs, written once at creation and read once at retirement, in an
// HBM scratch slice (no global access on the per-probe path: a pending global store would stall
// every workgroup barrier); 16-bit gap/pend and an index-form wide list -> ~1.9x the capacity.
// MODE 2: all fields in HBM scratch (last resort, up to 16384 live arms).
// atomic max on a 32-bit or (LDS, packed pairs) 16-bit element
__device__ inline void pend_max(uint32_t *a, uint32_t idx, uint32_t v) { atomicMax(&a[idx], v); }
__device__ inline void pend_max(uint16_t *a, uint32_t idx, uint32_t v) {
    // two 16-bit elements per word: the other half must be left untouched -> CAS loop
    uint32_t *w = reinterpret_cast<uint32_t *>(a) + (idx >> 1);
    const uint32_t sh = (idx & 1u) * 16u;
    uint32_t old = *w;
    for (;;) {
        const uint32_t cur = (old >> sh) & 0xFFFFu;
        if (cur >= v) break;
        const uint32_t upd = (old & ~(0xFFFFu << sh)) | (v << sh);
        const uint32_t prev = atomicCAS(w, old, upd);
        if (prev == old) break;
        old = prev;
    }
}

template <class PosT, int CAP, int NT, int MODE>
__global__ __launch_bounds__(NT) void extend_heavy_kernel(ExtParams<PosT> P) {
    constexpr int NW = NT / 64;
    constexpr bool PACKED_WIDE = MODE == 0;
    constexpr int HCAP = MODE != 2 ? CAP : 1;  // hot fields in LDS
    constexpr int CCAP = MODE == 0 ? CAP : 1;  // cold field (rs) in LDS
    // MODE 2: the capacity is a launch parameter (P.heavy_cap: max_cardinality * (ceil(G / step) + 1) bounds the
    // live arms of ANY segment -- every live arm was created or extended within the last t* + 1 processed probes,
    // at most max_cardinality of them per probe), slots are 32-bit and the per-arm index lists live in the HBM slice
    // as well; CAP is ignored.
    const uint32_t cap_rt = MODE == 2 ? P.heavy_cap : (uint32_t)CAP;
    constexpr uint32_t kSlotBits = MODE == 2 ? 24u : 20u;  // (creation number << kSlotBits) | slot
    constexpr uint32_t kSlotMask = (1u << kSlotBits) - 1u;
    using IdxT = typename std::conditional<MODE == 2, uint32_t, uint16_t>::type;
    constexpr uint32_t kEndIdx = MODE == 2 ? 0xFFFFFFFFu : 0xFFFFu;
    __shared__ PosT l_ls[HCAP], l_re[HCAP], l_le[HCAP], l_rs[CCAP];
    // gap and pend are 16-bit in the hybrid tier (gap saturates; the host only uses that tier when
    // G and max_cardinality fit): 24 B of LDS per arm instead of 32
    using SmallT = typename std::conditional<MODE == 1, uint16_t, uint32_t>::type;
    constexpr uint32_t kGapMax = MODE == 1 ? 0xFFFFu : 0xFFFFFFFFu;
    __shared__ uint32_t l_thr[HCAP], l_seq[HCAP];
    __shared__ SmallT l_gap[HCAP], l_pend[HCAP];
    PosT *s_ls = l_ls, *s_le = l_le, *s_rs = l_rs, *s_re = l_re;
    uint32_t *s_thr = l_thr, *s_seq = l_seq;
    SmallT *s_gap = l_gap, *s_pend = l_pend;
    IdxT *g_next = nullptr, *g_free = nullptr, *g_widx = nullptr;
    if constexpr (MODE != 0) {
        const size_t bytes = (size_t)cap_rt * (4 * sizeof(PosT) + (MODE == 2 ? 7 : 4) * sizeof(uint32_t));
        char *b = P.scratch + (size_t)blockIdx.x * bytes;
        PosT *g0p = reinterpret_cast<PosT *>(b);
        s_rs = g0p + 2 * (size_t)cap_rt;
        if constexpr (MODE == 2) {
            s_ls = g0p;
            s_le = g0p + cap_rt;
            s_re = g0p + 3 * (size_t)cap_rt;
            s_gap = reinterpret_cast<SmallT *>(g0p + 4 * (size_t)cap_rt);
            s_thr = reinterpret_cast<uint32_t *>(s_gap + cap_rt);
            s_seq = s_thr + cap_rt;
            s_pend = reinterpret_cast<SmallT *>(s_seq + cap_rt);
            g_next = reinterpret_cast<IdxT *>(s_pend + cap_rt);
            g_free = g_next + cap_rt;
            g_widx = g_free + cap_rt;
        }
    }
    constexpr uint32_t HT = MODE == 2 ? 8192u : (CAP <= 1024 ? 1024u : (MODE == 1 ? 2048u : (CAP <= 4608 ? 4096u : 8192u)));
    constexpr uint32_t WCAP = PACKED_WIDE ? (uint32_t)CAP : 1u;
    __shared__ uint32_t s_head[HT];
    __shared__ uint16_t l_next[MODE == 2 ? 1 : CAP];
    __shared__ uint16_t l_free[MODE == 2 ? 1 : CAP];  // stack of empty slots below the high-water mark
    __shared__ PosT s_ivlo[WCAP];     // wide arm w accepts x iff (x - s_ivlo[w]) < s_ivw[w]
    __shared__ uint32_t s_ivw[WCAP];
    __shared__ unsigned long long s_wkey[WCAP];  // (creation number << 20) | slot of wide arm w
    __shared__ uint16_t l_widx[(PACKED_WIDE || MODE == 2) ? 1 : CAP];  // index form of the wide list
    IdxT *s_next, *s_free, *s_widx;
    if constexpr (MODE == 2) {
        s_next = g_next; s_free = g_free; s_widx = g_widx;
    } else {
        s_next = l_next; s_free = l_free; s_widx = l_widx;
    }
    __shared__ PosT s_hits[kHitBatch];
    __shared__ unsigned long long s_best[NT];  // per hit: (creation number << 20) | slot, or ~0
    __shared__ uint32_t s_nwide, s_nfreed;
    __shared__ uint32_t s_wcnt[NT / 64];
    __shared__ unsigned long long s_bcast;
    const int tid = threadIdx.x, lane = tid & 63;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const RunParams &rp = P.rp;
    const uint64_t n_seg = *P.n_seg_ptr;
    const uint32_t k = (uint32_t)rp.k, step = (uint32_t)rp.step, G = rp.G;
    const uint32_t thr0 = arm_threshold(k, G);
    uint32_t bsh = 3;  // bucket(re) = re >> bsh with 2^bsh >= G + k: a hit meets <= 2 buckets
    while ((1ull << bsh) < (unsigned long long)G + k) ++bsh;
    const uint32_t cap_eff = min(cap_rt, P.cap_limit);
    RecAlloc rec_alloc;
    wg_begin(P);
    PROF_DECL;

    for (;;) {
        if (tid == 0) s_bcast = atomicAdd(P.cursor, 1ull);
        __syncthreads();
        const unsigned long long seg = s_bcast;
        __syncthreads();
        if (seg >= n_seg) break;
        const uint32_t g0 = P.seg_list[seg];
        if (tid == 0) {
            heartbeat(P, g0, 0u);
            seg_clock(P);
        }
        PROF_SEG_BEGIN();
        const int c = chunk_of_uniform(rp.ch, g0);
        const uint64_t cs = rp.ch.start[c], cl = rp.ch.len[c];
        const bool seg_rev = (rp.mode_of(c) & 2u) != 0u;  // (the orientation of the chunk's pass)
        const uint32_t pb = rp.ch.pbase[c];
        const uint32_t chunk_end = rp.ch.pbase[c + 1];
        const uint32_t g_end = min(chunk_end, rp.win_end(g0));  // (sharded calls: the window ends first)
        // block-uniform state: A live arms in slots [0,H), n_free of them empty (on s_free)
        uint32_t A = 0, H = 0, n_free = 0, quiet = 0, fam_seq = 0, next_seq = 0;
        if (tid == 0) s_nfreed = 0;
        bool overflow = false, done = false, fam_open = false;

        auto emit_records = [&](bool emit, PosT ls, PosT le, PosT rs, PosT re, uint32_t seq) {
            const unsigned long long em = __ballot(emit);
            if (!em) return;
            const unsigned long long at = rec_slot(rec_alloc, P, em, lane);
            if (emit) {
                if (at < P.rec_cap) {
                    const uint64_t ll = (uint64_t)le - (uint64_t)ls;
                    SdRec r;
                    r.g_start = g0;
                    r.fam_seq = fam_seq;
                    r.create_seq = seq;
                    r.pad = 0;
                    r.sd.left = seg_rev ? cs + cl - (uint64_t)ls - ll : (uint64_t)ls + cs;
                    r.sd.right = rs;
                    r.sd.left_length = ll;
                    r.sd.right_length = (uint64_t)re - (uint64_t)rs;
                    P.recs[at] = r;
                }
            }
        };
        // Age every live arm by `add` (unless `extended` applies first), retire in place the ones
        // whose gap reaches G.  i, off, row, from_lds describe the probe that may have extended
        // arms (with_pend); block-uniform on exit: A, n_free, fam_seq, next_seq, H.
        auto age_and_retire = [&](uint32_t add, bool with_pend, uint64_t i, uint32_t off,
                                  unsigned long long row, bool from_lds) {
            // s_nfreed was cleared at least one barrier ago (probe start / previous call's end)
            for (uint32_t j0 = 0; j0 < H; j0 += NT) {
                const uint32_t j = j0 + tid;
                bool dead = false;
                PosT ls = 0, le = 0, rs = 0, re = 0;
                uint32_t sq = kNoSeq;
                if (j < H && (sq = s_seq[j]) != kNoSeq) {
                    const uint32_t pd = with_pend ? s_pend[j] : 0u;
                    if (pd) {
                        s_pend[j] = 0;
                        const PosT x = from_lds ? s_hits[off + pd - 1u] : P.hits[row + pd - 1u];
                        s_re[j] = (PosT)(x + k);
                        s_le[j] = (PosT)(i + k);
                        s_thr[j] = arm_threshold((uint64_t)(i + k) - (uint64_t)s_ls[j], G);
                        s_gap[j] = 0;
                    } else {
                        const uint32_t gp = s_gap[j];
                        const uint64_t sum_g = (uint64_t)gp + add;
                        const uint32_t ng = sum_g > kGapMax ? kGapMax : (uint32_t)sum_g;
                        s_gap[j] = (SmallT)ng;
                        if (ng >= G) {
                            dead = true;
                            ls = s_ls[j]; le = s_le[j]; rs = s_rs[j]; re = s_re[j];
                            s_seq[j] = kNoSeq;
                            s_free[n_free + atomicAdd(&s_nfreed, 1u)] = (IdxT)j;
                        }
                    }
                }
                emit_records(dead && (uint64_t)(re - rs) >= rp.M, ls, le, rs, re, sq);
            }
            __syncthreads();
            const uint32_t nd = s_nfreed;
            __syncthreads();
            if (tid == 0) s_nfreed = 0;  // visible after the next barrier, before the next use
            A -= nd;
            n_free += nd;
            if (A == 0) {  // every slot is empty again
                H = 0;
                n_free = 0;
            } else if (H > 2u * A + 128u) {
                // Mostly holes (a long segment past its peak): pack the live arms into [0, A) so
                // that the per-probe loops run over A slots again.  Slot order is free (matching
                // goes by creation number).  Iteration by iteration: read, barrier, write below.
                uint32_t w = 0;
                for (uint32_t j0 = 0; j0 < H; j0 += NT) {
                    const uint32_t j = j0 + tid;
                    const bool live = j < H && s_seq[j] != kNoSeq;
                    PosT ls = 0, le = 0, rs = 0, re = 0;
                    uint32_t gp = 0, th = 0, sq = kNoSeq;
                    if (live) {
                        ls = s_ls[j]; le = s_le[j]; rs = s_rs[j]; re = s_re[j];
                        gp = s_gap[j]; th = s_thr[j]; sq = s_seq[j];
                    }
                    // ordered prefix of `live` over the workgroup
                    const unsigned long long lm = __ballot(live);
                    if (lane == 0) s_wcnt[tid >> 6] = (uint32_t)__popcll(lm);
                    __syncthreads();
                    uint32_t before = 0, tot = 0;
                    for (int wv = 0; wv < NW; ++wv) {
                        const uint32_t v = s_wcnt[wv];
                        if (wv < (tid >> 6)) before += v;
                        tot += v;
                    }
                    if (j < H) s_seq[j] = kNoSeq;  // every slot of this stripe has been read
                    __syncthreads();
                    if (live) {
                        const uint32_t d = w + before + (uint32_t)__popcll(lm & lt_mask);
                        s_ls[d] = ls; s_le[d] = le; s_rs[d] = rs; s_re[d] = re;
                        s_gap[d] = (SmallT)gp; s_thr[d] = th; s_seq[d] = sq; s_pend[d] = 0;
                    }
                    w += tot;
                    __syncthreads();
                }
                H = A;
                n_free = 0;
            }
        };
        // the flush of src/automaton.rs:182-200: every arm inactive
        auto maybe_close = [&]() {
            if (fam_open && A == 0) {
                ++fam_seq;
                next_seq = 0;
                fam_open = false;
            }
        };
        auto advance_quiet = [&](uint32_t q) {
            quiet += q;
            if (A > 0) age_and_retire(q * step, false, 0, 0, 0, true);
            maybe_close();
            if (A == 0 && quiet >= rp.tstar) done = true;
        };

        for (uint32_t g = g0; g < g_end && !done;) {
            // ---- stage a batch of up to 64 probes (every wave computes the same masks) ----
            PROF_START();
            const uint32_t nb = min(64u, g_end - g);
            if (tid == 0) heartbeat(P, g0, g);
            const uint32_t f_l = (uint32_t)lane < nb ? P.p_filt[g + lane] : kSkipN;
            const unsigned long long r_l = (uint32_t)lane < nb ? P.row_off[g + lane] : 0ull;
            const unsigned long long r_hi = P.row_off[g + nb];
            const unsigned long long base = __shfl(r_l, 0);
            unsigned long long r_next = __shfl_down(r_l, 1);
            if ((uint32_t)lane + 1 >= nb) r_next = r_hi;
            const bool fits = (uint32_t)lane < nb && r_next - base <= (unsigned long long)kHitBatch;
            const unsigned long long fm = __ballot(fits);
            uint32_t nbb = (~fm == 0ull) ? 64u : (uint32_t)(__ffsll((long long)~fm) - 1);
            if (nbb > nb) nbb = nb;
            bool first_from_global = false;
            if (nbb == 0) {
                nbb = 1;
                first_from_global = true;
            }
            const uint32_t rel_l = (uint32_t)(r_l - base);
            if (!first_from_global) {
                const unsigned long long end = nbb == nb ? r_hi : __shfl(r_l, (int)nbb);
                const uint32_t tot = (uint32_t)(end - base);
                for (uint32_t r = tid; r < tot; r += NT) s_hits[r] = P.hits[base + r];
            }
            __syncthreads();
            const unsigned long long in_batch = nbb >= 64 ? ~0ull : ((1ull << nbb) - 1ull);
            const unsigned long long hm = __ballot(f_l >= 1u && f_l < kPending) & in_batch;
            const unsigned long long qm = __ballot(f_l == 0u) & in_batch;
            PROF_STOP(0);
            PROF_COUNT(1, 1);
            uint32_t pos = 0;
            while (!done) {
                const unsigned long long hmr = pos >= 64 ? 0ull : (hm >> pos) << pos;
                if (!hmr) break;
                const uint32_t b = (uint32_t)(__ffsll((long long)hmr) - 1);
                {
                    const unsigned long long range = ((1ull << b) - 1ull) & ~((1ull << pos) - 1ull);
                    const uint32_t q = (uint32_t)__popcll(qm & range);
                    if (q) {
                        advance_quiet(q);
                        if (done) break;
                    }
                }
                quiet = 0;
                pos = b + 1;
                const uint32_t cnt = __shfl(f_l, (int)b);
                const uint32_t off = __shfl(rel_l, (int)b);
                const uint64_t i = (uint64_t)(g + b - pb + 1) * step;
                const unsigned long long row = base + off;
                if (A + cnt > cap_eff) {
                    overflow = true;
                    done = true;
                    break;
                }
                const bool from_lds = !first_from_global;
                PROF_COUNT(5, 1);
                PROF_COUNT(10, A);
                PROF_COUNT(11, cnt);
                PROF_MAX(9, A + cnt);
                PROF_START();
                // ---- (0)+(1) candidate index over the live arms ------------------------------
                uint32_t hmask = 63u;
                while (hmask + 1u < HT && hmask + 1u < 2u * A) hmask = (hmask << 1) | 1u;
                for (uint32_t h = tid; h <= hmask; h += NT) s_head[h] = 0xFFFFFFFFu;
                if (tid == 0) s_nwide = 0;
                __syncthreads();
                for (uint32_t j0 = 0; j0 < H; j0 += NT) {
                    const uint32_t j = j0 + tid;
                    uint32_t sq = kNoSeq, th = 0;
                    PosT re = 0;
                    if (j < H && (sq = s_seq[j]) != kNoSeq) {
                        th = s_thr[j];
                        re = s_re[j];
                    }
                    const bool live = sq != kNoSeq;
                    if (live && th <= G) {
                        const uint32_t bkt = (uint32_t)((uint64_t)re >> bsh);
                        s_next[j] = (IdxT)atomicExch(&s_head[((bkt * 2654435761u) >> 12) & hmask], j);
                    }
                    // wide arms: one LDS atomic per wave, not per arm
                    const bool wide = live && th > G;
                    const unsigned long long wm = __ballot(wide);
                    if (wm) {
                        const int leader = __ffsll((long long)wm) - 1;
                        uint32_t wbase = 0;
                        if (lane == leader) wbase = atomicAdd(&s_nwide, (uint32_t)__popcll(wm));
                        wbase = __shfl(wbase, leader);
                        const uint32_t d = wbase + (uint32_t)__popcll(wm & lt_mask);
                        if (wide) {
                            if constexpr (PACKED_WIDE) {
                                const uint64_t wv = (uint64_t)th + k - 1u;
                                s_ivlo[d] = (PosT)(re - k + 1u);
                                s_ivw[d] = wv > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)wv;
                                s_wkey[d] = ((unsigned long long)sq << kSlotBits) | j;
                            } else {
                                s_widx[d] = (IdxT)j;
                            }
                        }
                    }
                }
                __syncthreads();
                PROF_STOP(2);
                PROF_START();
                const uint32_t n_wide = s_nwide;
                const uint32_t seq_base = next_seq;
                for (uint32_t t0 = 0; t0 < cnt; t0 += NT) {
                    const uint32_t ct = min((uint32_t)NT, cnt - t0);
                    if (t0) __syncthreads();
                    // ---- (2) narrow arms: one thread per hit, two buckets ----------------------
                    PosT hx = 0;
                    if ((uint32_t)tid < ct) {
                        hx = from_lds ? s_hits[off + t0 + tid] : P.hits[row + t0 + tid];
                        const uint64_t lo_re = (uint64_t)hx + 1u > (uint64_t)G ? (uint64_t)hx + 1u - G : 0u;
                        const uint32_t b0 = (uint32_t)(lo_re >> bsh);
                        const uint32_t b1 = (uint32_t)(((uint64_t)hx + k - 1u) >> bsh);
                        unsigned long long best = ~0ull;
                        for (uint32_t bkt = b0; bkt <= b1; ++bkt) {
                            uint32_t j = s_head[((bkt * 2654435761u) >> 12) & hmask];
                            while (j != 0xFFFFFFFFu && j != kEndIdx) {
                                if (arm_accepts<PosT>(hx, s_re[j], s_thr[j], k))
                                    best = min(best, ((unsigned long long)s_seq[j] << kSlotBits) | j);
                                j = s_next[j];
                            }
                        }
                        s_best[tid] = best;
                    }
                    __syncthreads();
                    PROF_STOP(4);
                    PROF_START();
                    // ---- wide arms: thread = (hit, part of the packed list), branch-free --------
                    if (n_wide) {
                        const uint32_t Hr = (ct + 63u) & ~63u;  // hits rounded up to waves
                        const uint32_t NP = NT / Hr;            // list parts
                        const uint32_t tl = (uint32_t)tid % Hr, part = (uint32_t)tid / Hr;
                        const bool valid = tl < ct && part < NP;
                        PosT x = 0;
                        if (valid) x = from_lds ? s_hits[off + t0 + tl] : P.hits[row + t0 + tl];
                        if (part < NP) {  // wave-uniform
                            const uint32_t j0 = (uint32_t)((uint64_t)n_wide * part / NP);
                            const uint32_t j1 = (uint32_t)((uint64_t)n_wide * (part + 1) / NP);
                            unsigned long long found = ~0ull;
                            // 8 independent LDS load chains in flight per thread (the scan is
                            // latency-bound otherwise), smallest accepting key wins
                            uint32_t j = j0;
                            if constexpr (PACKED_WIDE) {
                                for (; j + 8 <= j1; j += 8) {
                                    PosT lo8[8];
                                    uint32_t w8[8];
                                    unsigned long long k8[8];
#pragma unroll
                                    for (int u = 0; u < 8; ++u) {
                                        lo8[u] = s_ivlo[j + u];
                                        w8[u] = s_ivw[j + u];
                                        k8[u] = s_wkey[j + u];
                                    }
#pragma unroll
                                    for (int u = 0; u < 8; ++u)
                                        found = min(found, (uint64_t)(PosT)(x - lo8[u]) < w8[u] ? k8[u] : ~0ull);
                                }
                                for (; j < j1; ++j)
                                    found = min(found, (uint64_t)(PosT)(x - s_ivlo[j]) < s_ivw[j] ? s_wkey[j] : ~0ull);
                            } else {
                                for (; j + 8 <= j1; j += 8) {
                                    uint32_t sl8[8], th8[8], sq8[8];
                                    PosT re8[8];
#pragma unroll
                                    for (int u = 0; u < 8; ++u) sl8[u] = s_widx[j + u];
#pragma unroll
                                    for (int u = 0; u < 8; ++u) {
                                        re8[u] = s_re[sl8[u]];
                                        th8[u] = s_thr[sl8[u]];
                                        sq8[u] = s_seq[sl8[u]];
                                    }
#pragma unroll
                                    for (int u = 0; u < 8; ++u) {
                                        const unsigned long long key = ((unsigned long long)sq8[u] << kSlotBits) | sl8[u];
                                        found = min(found, arm_accepts<PosT>(x, re8[u], th8[u], k) ? key : ~0ull);
                                    }
                                }
                                for (; j < j1; ++j) {
                                    const uint32_t slot = s_widx[j];
                                    const unsigned long long key = ((unsigned long long)s_seq[slot] << kSlotBits) | slot;
                                    found = min(found, arm_accepts<PosT>(x, s_re[slot], s_thr[slot], k) ? key : ~0ull);
                                }
                            }
                            if (valid && found != ~0ull) atomicMin(&s_best[tl], found);
                        }
                        __syncthreads();
                    }
                    PROF_STOP(8);
                    PROF_START();
                    // ---- (3) ExtendArm / NewArm, one thread per hit -----------------------------
                    const bool mine = (uint32_t)tid < ct;
                    unsigned long long best = ~0ull;
                    if (mine) {
                        best = s_best[tid];
                        if (best != ~0ull) pend_max(s_pend, (uint32_t)(best & kSlotMask), t0 + tid + 1u);
                    }
                    // unmatched hits become arms
                    const bool is_new = mine && best == ~0ull;
                    // rank of this hit among the new arms, in hit order (= creation order): every
                    // wave recomputes the per-group counts from s_best (no barrier)
                    uint32_t before = 0, n_new = 0;
                    for (uint32_t c0 = 0; c0 < ct; c0 += 64) {
                        const uint32_t hidx = c0 + lane;
                        const bool un = hidx < ct && s_best[hidx] == ~0ull;
                        const unsigned long long nm = __ballot(un);
                        const uint32_t pc = (uint32_t)__popcll(nm);
                        if (c0 < ((uint32_t)tid & ~63u)) before += pc;
                        else if (c0 == ((uint32_t)tid & ~63u)) before += (uint32_t)__popcll(nm & lt_mask);
                        n_new += pc;
                    }
                    if (is_new) {
                        // reuse empty slots first (top of the stack), then grow the high-water mark
                        const uint32_t slot = before < n_free ? (uint32_t)s_free[n_free - 1u - before]
                                                              : H + (before - n_free);
                        s_ls[slot] = (PosT)i; s_le[slot] = (PosT)(i + k); s_rs[slot] = hx;
                        s_re[slot] = (PosT)(hx + k);
                        s_gap[slot] = 0;  // aged to `step` by this very probe in (4)
                        s_thr[slot] = thr0;
                        s_seq[slot] = next_seq + before;
                        s_pend[slot] = 0;
                    }
                    if (n_new <= n_free) {
                        n_free -= n_new;
                    } else {
                        H += n_new - n_free;
                        n_free = 0;
                    }
                    A += n_new;
                    next_seq += n_new;
                    __syncthreads();
                }
                (void)seq_base;
                PROF_STOP(6);
                PROF_START();
                // ---- (4) apply ExtendArm (last hit in SA order wins), age, retire -----------------
                age_and_retire(step, true, i, off, row, from_lds);
                fam_open = true;
                maybe_close();
                PROF_STOP(7);
            }
            if (!done) {
                const unsigned long long range = pos >= 64 ? 0ull : ~((1ull << pos) - 1ull);
                const uint32_t q = (uint32_t)__popcll(qm & range);
                if (q) advance_quiet(q);
            }
            __syncthreads();
            g += nbb;
        }
        if (!done && g_end < chunk_end) {
            if (tid == 0) atomicAdd(&P.ctr[CT_RANOUT], 1ull);
        } else if (!overflow && fam_open) {
            emit_records(tid == 0, (PosT)0, (PosT)0, (PosT)0, (PosT)0, kTombstone);
        }
        if (overflow && tid == 0) {
            const unsigned long long at = atomicAdd(P.ovf_count, 1ull);
            if (P.ovf_list) P.ovf_list[at] = g0;
        }
        if (tid < 64) {
            PROF_FLUSH();
        }
        __syncthreads();
    }
    rec_flush(rec_alloc, P, lane);
    wg_busy(P);
}

// yardstick: sum over searched probes of ceil(log2(b_p + 1)), b_p = size of the
// reference's 8-mer bucket.  Untimed.
template <class SlotT>
__global__ __launch_bounds__(256) void yardstick_kernel(IndexView<SlotT> ix, RunParams rp,
                                                        const uint32_t *__restrict__ p_filt,
                                                        unsigned long long *__restrict__ ctr) {
    uint32_t g_end;
    const uint32_t g = rp.tile_of(blockIdx.x, 256u, g_end) + threadIdx.x;
    unsigned long long steps = 0;
    if (g < g_end && p_filt[g] != kSkipN) {
        const int c = chunk_of(rp.ch, g);
        const uint64_t s = rp.ch.start[c], L = rp.ch.len[c];
        const uint64_t i = (uint64_t)(g - rp.ch.pbase[c] + 1) * (uint64_t)rp.step;
        uint32_t first;
        const uint32_t md = rp.mode_of(c);
        const uint64_t q = probe_key(ix.text, s, L, i, rp.k, (md & 2u) != 0u, (md & 1u) != 0u, &first);
        uint32_t c8;
        if (cache8_index((uint32_t)(q >> (3 * (ix.kk - kCacheLen))), c8)) {
            const uint64_t b = (uint64_t)ix.c8hi[c8] - (uint64_t)ix.c8lo[c8];
            uint64_t v = 1;
            while (v < b + 1) {
                v <<= 1;
                ++steps;
            }
        }
    }
    for (int off = 32; off > 0; off >>= 1) steps += __shfl_down(steps, off);
    if ((threadIdx.x & 63) == 0 && steps) atomicAdd(&ctr[CT_BISECT], steps);
}

}  // namespace asgart
