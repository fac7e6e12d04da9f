// index.hpp -- the device-resident index: text + suffix array + k-specific keys.
//
// HBM layout (one copy per GPU, replicated across ranks):
//   text   u8 [n + 64]      raw bytes as given (reference strand.data incl. '$')
//   sa     u32[n] | u64[n]  suffix array; 32-bit entries whenever n < 2^32-64
//   keys   u64[n]           keys[r] = first k bases of suffix sa[r] as 3-bit
//                           order-preserving codes, '$'/end padded with 0:
//                           a sorted array, so a k-mer lookup is a binary
//                           search over ONE array with no text gathers
//   ptab   slot[4^d + 1]    first slot whose key >= the ACGT d-mer prefix
//   c8lo/c8hi slot[5^8]     the reference's 8-mer cache (Searcher::new)
#pragma once

#include "common.hpp"

#include <functional>
#include <memory>
#include <thread>
#include <condition_variable>
#include <mutex>

namespace asgart {

template <class SlotT>
struct IndexView {
    const uint8_t *text;
    const SlotT *sa;
    const uint64_t *keys;
    const SlotT *ptab;
    const SlotT *c8lo;
    const SlotT *c8hi;
    uint64_t n;
    int k;   // probe size
    int kk;  // bases held by a key word: min(k, kMaxKey)
    int k2;  // k - kk: bases of a long probe compared through the text
    int d;
    // text-tail corner (reference src/searcher.rs:165-166): 8-mer prefixes
    // (24-bit codes) of the suffixes shorter than k; probes with one of these
    // prefixes take the exact-bisection slow path.
    uint32_t tail8[kMaxK];
    int n_tail8;
    uint64_t tail_bloom;
    // (the presence filters and their position bitmaps are per orientation: RunParams::flt / pbits)
    // occurrences of every k-mer interval sorted by position (sa_build.hip: build_rank_lists); null: none
    const SlotT *sap;
    // number of suffix-array slots: n, or end - start + 1 for a --trim index (reference
    // src/bin/asgart.rs:142-148), whose array holds the suffixes of data[start..end] + '$' only
    uint64_t n_sa;
    // --trim: slots whose sub-strand suffix is shorter than k.  The array is sorted by the sub-strand's
    // suffixes but compared through the whole text, so these few are out of place: any 8-mer bucket that
    // contains one is searched by replaying the reference's bisection step by step.
    int trim;
    int n_bad;
    uint64_t bad[kMaxK + 2];
};

struct ChunkTable {
    const uint64_t *start;
    const uint64_t *len;
    const uint32_t *pbase;  // n_chunks + 1
    int n_chunks;
};

struct RunParams {
    ChunkTable ch;
    // The probes this call computes: one WINDOW per pass, all of one length and one stride apart --
    //     window w = [g_lo + w * win_stride, g_lo + w * win_stride + win_len),   w < n_passes
    // An unsharded call: the windows are the passes themselves (win_len == win_stride == probes of a pass).  Shard r of
    // n: every window is the r-th slice of its pass's probe sequence plus halos, and the call numbers its probes
    // VIRTUALLY -- only the chunks a window touches keep their probes in ChunkTable::pbase (the others are empty there),
    // so that the per-probe arrays span [g_lo, g_hi) with nothing but the cut chunks' remainders between the windows.
    // Every kernel that walks probes maps tiles to windows (tile_of) or stops at its window's end (win_end).
    uint32_t g_lo, g_hi;           // extent of the per-probe arrays: first probe of window 0, end of the last window
    uint32_t win_len, win_stride;
    uint32_t own_off_lo, own_off_hi;  // segments that start at window offsets [own_off_lo, own_off_hi) belong to this call
    uint32_t init_unknown;         // 1: the automaton state in front of a window is unknown (the windows start mid-chunk)
    int k, step;
    uint32_t G;           // max_gap_size
    uint32_t tstar;       // ceil(G / step)
    uint64_t M;           // min_duplication_length
    uint32_t C;           // max_cardinality (clamped)
    // The probe sequence of a call is the chunk list once per PASS (one pass = one orientation = one invocation of the
    // reference's binary, src/bin/asgart.rs:677-693): pass p owns chunks [p * pass_chunks, (p + 1) * pass_chunks) of the
    // chunk table and its probes are needles prepared the pass's way (asgart.rs:206-218).  A plain call has one pass;
    // asgart_search_duplications_passes runs up to four as ONE job -- one front over all their probes, one launch per
    // extension tier over the merged, cost-sorted segment list.
    uint32_t n_passes, pass_chunks;
    uint32_t modes;       // 8 bits per pass: bit 1 = reversed needle, bit 0 = complemented
    const uint64_t *flt[4];    // per pass: presence filter of its orientation (search_dev.hpp); null: none
    const uint64_t *pbits[4];  // ... its answers laid out by TEXT POSITION (bit p: the probe that covers text[p .. p + k) in
                               // this orientation passes the filter); null: the kernels test the hashed filter
    int flt_bits;
    uint32_t blank;            // bit p: ... and this call finds them blank (the accounting pass prices its probes unfiltered)
    uint32_t learn;            // bit p: pass p's position bits are LEARNED by the search (option lazy_aux: no filter is ever built for
                               // the orientation; its bitmap starts all ones and probe_count_kernel clears the bit of every probe it
                               // finds without an occurrence that could be kept -- the same index-derived fact the refined build
                               // computes, one call late and only where calls probe)
    __host__ __device__ inline uint32_t pass_of(int c) const {
        if (n_passes <= 1u) return 0u;
        const uint32_t uc = (uint32_t)c;
        return (uc >= pass_chunks ? 1u : 0u) + (uc >= 2u * pass_chunks ? 1u : 0u) + (uc >= 3u * pass_chunks ? 1u : 0u);
    }
    __host__ __device__ inline uint32_t mode_of_pass(uint32_t p) const { return (modes >> (8u * p)) & 3u; }
    __host__ __device__ inline uint32_t mode_of(int c) const { return mode_of_pass(pass_of(c)); }
    // window of probe g (g_lo <= g < g_hi; a probe between two windows counts with the window in front of it)
    __host__ __device__ inline uint32_t win_of(uint32_t g) const {
        if (n_passes <= 1u) return 0u;
        const uint32_t d = g - g_lo;
        return (d >= win_stride ? 1u : 0u) + (d >= 2u * win_stride ? 1u : 0u) + (d >= 3u * win_stride ? 1u : 0u);
    }
    __host__ __device__ inline uint32_t win_lo(uint32_t w) const { return g_lo + w * win_stride; }
    __host__ __device__ inline uint32_t win_end(uint32_t g) const { return win_lo(win_of(g)) + win_len; }
    __host__ __device__ inline bool owned(uint32_t g) const {
        const uint32_t off = g - win_lo(win_of(g));
        return off >= own_off_lo && off < own_off_hi;
    }
    // Tiles of `tile` probes laid over the windows, window by window (a window's last tile may be partial):
    // tile t -> its first probe, and through `end` the end of its window
    __host__ __device__ inline uint32_t tiles_per_window(uint32_t tile) const { return (win_len + tile - 1u) / tile; }
    __host__ __device__ inline uint32_t n_tiles(uint32_t tile) const { return n_passes * tiles_per_window(tile); }
    __host__ __device__ inline uint32_t tile_of(uint32_t t, uint32_t tile, uint32_t &end) const {
        const uint32_t tw = tiles_per_window(tile);
        const uint32_t w = n_passes <= 1u ? 0u : (t >= tw ? 1u : 0u) + (t >= 2u * tw ? 1u : 0u) + (t >= 3u * tw ? 1u : 0u);
        const uint32_t lo = win_lo(w);
        end = lo + win_len;
        return lo + (t - w * tw) * tile;
    }
};

// One emitted duplication arm, keyed for the reference's output order (chunk order, discovery order
// inside the chunk, arm order inside the family).
struct SdRec {
    uint32_t g_start;     // first probe of the segment: sort key 1
    uint32_t fam_seq;     // family ordinal inside the segment: sort key 2
    uint32_t create_seq;  // creation number inside the family: sort key 3; ~0u = tombstone
    uint32_t pad;
    asgart_proto_sd sd;
};

struct Workspace {
    DevBuf chunks;     // ch_start[nc], ch_len[nc] (u64) then pbase[nc+1] (u32)
    DevBuf p_lo;       // SlotT[P]
    DevBuf p_raw;      // u32[P]
    DevBuf p_filt;     // u32[P]
    DevBuf row_off;    // u64[P+1]
    DevBuf blk;        // scan block aggregates
    DevBuf hits;       // PosT[total hits]
    DevBuf big_list;   // u32[P] probes with large SA intervals
    DevBuf rank_list;  // u32[P] ... of them, those counted by bisection (position-sorted lists)
    DevBuf seg_list;   // u32[...] segment start probes
    DevBuf counters;   // u64[32] device counters
    DevBuf fam_sds;    // SdRec[cap] output records of the extension kernel
    DevBuf ovf_list;   // u32 segments that overflowed the small arm tier
    DevBuf scratch;    // arm storage of the global heavy tier
    DevBuf seg_keys, seg_vals, sort_tmp;  // segment placement: (tier, work) keys, double-buffered
    DevBuf rec_k32, rec_k64, rec_idx, rec_sorted;  // ordering of the output records
    DevBuf pat;        // pattern upload scratch
    DevBuf out_a, out_b;
    DevBuf seg_info;   // uint2 per segment: hits and probe positions walked (placement -> cluster_barren_kernel)
    DevBuf seg_slots;  // u64[8][4096]: per tier and workgroup, when its current segment started (longest-segment statistics)
    DevBuf split_buf;  // long segments run as ranges (option split): counters, runs, cuts, split segments, run states, verdicts
    DevBuf split_dump; // ... and the arms every run holds at its cut and at its end (8 words per arm, 2 x kRunDumpCap arms per run)
    // every buffer goes back to the device (or to the block cache): ONE list, next to the members
    void release_all() {
        DevBuf *bufs[] = {&chunks, &p_lo, &p_raw, &p_filt, &row_off, &blk, &hits, &big_list, &rank_list, &seg_list,
                          &counters, &fam_sds, &ovf_list, &scratch, &seg_keys, &seg_vals,
                          &sort_tmp, &rec_k32, &rec_k64, &rec_idx, &rec_sorted, &pat, &out_a, &out_b, &seg_info, &seg_slots, &split_buf, &split_dump};
        static_assert(sizeof(Workspace) == sizeof(bufs) / sizeof(bufs[0]) * sizeof(DevBuf),
                      "a buffer of the workspace is missing from release_all");
        for (DevBuf *b : bufs) b->release();
    }
};

}  // namespace asgart

namespace asgart {
// Everything one search call mutates.  The index owns two of them so that two passes (say the
// direct and the -RC run, reference src/bin/asgart.rs runs them as separate invocations) can be
// in flight at once from two host threads; text, suffix array and keys are shared read-only.
struct SearchCtx {
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr, stream3 = nullptr, stream4 = nullptr;  // concurrent extension tiers
    hipStream_t stream5 = nullptr, stream6 = nullptr, stream7 = nullptr;
    hipStream_t fill_stream = nullptr;  // where the CSR fill runs beside the placement walk: stream2, or the main stream when
                                        // hardware queues are scarce (create_ctx_streams)
    hipEvent_t ev[17] = {};
    Workspace ws;
    asgart_stats stats;
    RunParams last_rp;   // inputs of the last call, kept for the yardstick kernel
    bool has_last = false;
    bool raw_done = false;  // asgart_stats.raw_hits of the last call has been summed up (asgart_get_stats does it on demand)
    uint32_t last_P = 0;
    bool busy = false;
    volatile uint64_t *progress = nullptr;  // of the call in flight (asgart_search_duplications), or null
    // pinned host staging for the sorted output records (a pageable target makes the D2H copy several
    // times slower than the kernels that produce it); grow-only, freed with the index
    void *h_pinned = nullptr;
    size_t h_pinned_cap = 0;
    // host-side parts of the family assembly (run_search_t), kept for their capacity
    struct FamPart {
        std::vector<asgart_proto_sd> sds[4];
        std::vector<uint64_t> ends[4], keys[4];  // per family: records up to and including it (within the part), key
    };
    std::vector<FamPart> fam_parts;
    int32_t pinned(size_t bytes, void **out) {
        if (bytes > h_pinned_cap) {
            if (h_pinned) (void)hipHostFree(h_pinned);
            h_pinned = nullptr;
            h_pinned_cap = 0;
            const size_t want = bytes + bytes / 4 + 4096;
            if (hipHostMalloc(&h_pinned, want, hipHostMallocDefault) != hipSuccess) {
                (void)hipGetLastError();
                h_pinned = nullptr;
                set_error("hipHostMalloc(%zu bytes) failed", want);
                return ASGART_E_OOM;
            }
            h_pinned_cap = want;
        }
        *out = h_pinned;
        return 0;
    }
    // Heartbeats of the persistent extension workgroups (watchdog): a pinned, device-mapped block of kHbTiers x kHbSlots
    // entries {segment start probe | tier << 32 , position / step counter}; a workgroup stamps its slot when it takes a
    // segment (and the workgroup kernels again at every batch of probes).  The host reads it while it waits: as long as
    // it changes, the device is making progress; when a wait outlasts option watchdog_s without any change, the call
    // returns ASGART_E_HIP with the slots that were in flight.
    static constexpr int kHbTiers = 8, kHbSlots = 256;
    unsigned long long *h_hb = nullptr;  // host address
    unsigned long long *d_hb = nullptr;  // device address of the same block
    int32_t heartbeat(bool want) {
        if (!want || h_hb) return 0;
        void *hp = nullptr, *dp = nullptr;
        if (hipHostMalloc(&hp, (size_t)kHbTiers * kHbSlots * 16, hipHostMallocMapped) != hipSuccess ||
            hipHostGetDevicePointer(&dp, hp, 0) != hipSuccess) {
            (void)hipGetLastError();
            if (hp) (void)hipHostFree(hp);
            return 0;  // (diagnostics only: the call goes on without)
        }
        memset(hp, 0, (size_t)kHbTiers * kHbSlots * 16);
        h_hb = static_cast<unsigned long long *>(hp);
        d_hb = static_cast<unsigned long long *>(dp);
        return 0;
    }
    // pinned control block: the counters read back several times per call, then the chunk table.  Transfers
    // to and from pageable memory go through a staging copy KERNEL, which needs a free CU -- with another
    // call's persistent extension workgroups on the chip that wait was 20-30 ms per copy.
    void *h_ctl = nullptr;
    size_t h_ctl_cap = 0;
    int32_t ctl(size_t bytes, void **out) {
        if (bytes > h_ctl_cap) {
            if (h_ctl) (void)hipHostFree(h_ctl);
            h_ctl = nullptr;
            h_ctl_cap = 0;
            const size_t want = bytes + 65536;
            if (hipHostMalloc(&h_ctl, want, hipHostMallocDefault) != hipSuccess) {
                (void)hipGetLastError();
                h_ctl = nullptr;
                set_error("hipHostMalloc(%zu bytes) failed", want);
                return ASGART_E_OOM;
            }
            h_ctl_cap = want;
        }
        *out = h_ctl;
        return 0;
    }
};
constexpr int kNumCtx = 2;

// Tuning / test options of one index.  They are read from the environment (ASGART_<NAME>) exactly
// once, by asgart_index_create, range-checked there, and can be changed between calls with
// asgart_index_set_option; the search path itself never calls getenv.
struct Options {
    int64_t shard_lookback = 4096;  // probes of look-back halo of a sharded call (grows on retry)
    int64_t shard_lookahead = 0;    // 0: max(65536, own range / 16)
    int64_t force_tier = 0;         // tests: minimum extension tier of segments with a multi-hit probe
    int64_t arms_kernel = 1;        // 0: LDS-array kernels in tiers 2, 4, 6 (what max_cardinality > 1024 selects)
    int64_t long3 = 16384;          // probes; longer segments go to the long-segment tiers (3: dense, 6: sparse) whatever their arm
                                    // bound.  Round 5 (the passes of a step as one job: the extension is bound by the compute-unit
                                    // time the tiers hold, not by a serial chain): 4096 -> 16384, 250 -> 236 ms per step at GRCh38
                                    // size -- a long SPARSE segment in tier 6 owns a whole compute unit for one wave's work
    int64_t cap1 = 256;             // live-arm bound up to which a segment may use the one-wave tier
    int64_t debug = 0;              // 1: per-call tier statistics on stderr; 2: also, per tier, the longest and the richest segment that runs whole
    int64_t test_cap_limit = -1;    // tests: shrink every tier's capacity (forces the cascade)
    int64_t test_genbits = 22;      // tests: width of the arm-resident kernels' table generation counter
    int64_t test_k8_delay = 0;      // tests: cycles K8's ranking wave waits in every step before it reads the free counts the arm
                                    // waves published (results must not depend on it: the counts are double-buffered by step parity)
    int64_t tier_order = 3654217;   // launch order of the extension tiers, as decimal digits
    int64_t grid[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // grid[t] > 0: workgroups of tier t (clamped to its maximum)
    int64_t ptab_depth = 0;         // 0: chosen from the text length
    int64_t force_wide = 0;         // tests: 64-bit slots and positions for a small text
    int64_t test_wide_batch = 0;    // tests: batch size of the 64-bit suffix sorter's doubling rounds (0: 2^29)
    int64_t kfilter_bits = 30;      // log2(bits) of the k-mer presence filter (search_dev.hpp); 0: no filter
    int64_t rank_lists = 1;         // 1: position-sorted occurrence lists for the cardinality test (k <= 21, no --trim)
    int64_t cap6_pct = 140;         // tier 6 accepts segments whose arm bound is up to this percentage of its capacity
    int64_t test_fail_alloc = -1;   // tests: the (n+1)-th device allocation from now on fails once (common.hpp); -1 = off
    int64_t solo = 1;               // one-barrier kernel: sparse probes run on wave 0 alone, in registers (see solo_probe): 0 never, 1 = probes of up to 16 hits, n = up to n (<= 48) hits; 8 .. 44 measured alike at cfg4 and cfg5
    int64_t cap6w_pct = 160;        // ... with 64-bit positions (the bound is three to four times what a segment really holds there)
    int64_t cap45_pct = 100;        // tiers 4 and 5 accept segments whose arm bound is up to this percentage of their capacity (what overflows is re-run)
    int64_t cap3_pct = 160;         // tier 3 accepts segments whose arm bound is up to this percentage of its capacity
    int64_t posbits = 2;            // the presence filter's answers laid out by text position (built with the filter; the search reads those):
                                    // 1 = the k-mer filter's answer, 2 = refined -- a position keeps its bit only if a hit of its probe can be
                                    // KEPT (an occurrence behind the probe: half the lookups the k-mer filter lets through keep nothing;
                                    // build_posbits_kernel, index.hip), 0 = none: the kernels test the hashed filter
    int64_t barren = 2;             // segments that provably emit nothing are not run at all: 1 = those with too few hit-probes for any arm to
                                    // reach min_duplication_length (pipeline_dev.hpp: segment_is_barren); 2 = also those whose hits leave
                                    // no run of consecutive occupied position buckets long enough (cluster_barren_kernel: the bursts of
                                    // interspersed repeats); 0 = every segment runs
    int64_t fuse_passes = 1;        // asgart_search_duplications_passes, passes that differ in orientation only: 2 = always as ONE job (one
                                    // front over all their probes, one launch per extension tier over the merged segment list); 0 = always
                                    // as pipelined single-pass calls on the two call contexts; 1 = the first call
                                    // with given settings runs as one job; while the extension is bound by compute-unit time (the longest
                                    // single segment below fuse_pole_pct per cent of the extension: GRCh38-shaped 125 vs 165 ms) so do
                                    // the calls after it; when ONE segment is the extension (a megabase higher-order array, a chromosome
                                    // against its homologue: the other pass's front may hide beside it -- 393 vs 466 ms, 2 071 vs
                                    // 2 178 ms) the next calls are TIMED both ways, two each, and the faster way is kept (a GRCh38-shaped
                                    // step sits at 85-87 per cent: a guess at the threshold would cost it 40 ms)
    int64_t fuse_pole_pct = 88;
    int64_t lazy_aux = 1;           // 1: no presence filter is BUILT: the position bits of an orientation start all ones and its first search
                                    // clears, as a by-product of its lookups, the bit of every probe without an occurrence that could be
                                    // kept (from its second search on the orientation is filtered); the position-sorted lists are built
                                    // when a search call has had a predecessor (0.2 s at GRCh38 size for 0.006 s per pass) -- a host that
                                    // runs every orientation once per index (the reference's own use, src/bin/asgart.rs:677-693) pays for
                                    // neither; 0: filter + refined position bits built on first use (0.3 s per orientation), lists with
                                    // the keys
    int64_t dense3 = 16;            // long segments go to tier 3 (K8) only with at least this many hits per processed probe on average
                                    // (0: all of them); the sparse long ones run on tier 6's kernel (K6, solo probes)
    int64_t dense6 = 32;            // segments of ANY length whose arm bound sends them to tier 6 go to tier 3 instead with at least this
                                    // many hits per processed probe on average (0: off)
    int64_t split = 1;              // 1: the long segments of the long-shape tiers run as RANGES side by side (plan_ranges_kernel,
                                    // pipeline_dev.hpp): every range starts from an empty arm list split_warm probes in front of its cut,
                                    // and what it holds AT the cut is compared with what the range before holds there; a segment with a cut
                                    // that differs keeps the ranges up to it and runs the rest as ONE more run (the whole segment again when
                                    // its first cut fails); the index remembers what the segment needs (split_warm_max).  With 32-bit
                                    // positions; 2: with 64-bit positions as well -- not the default: the one workload of that size here
                                    // (two genomes, 6.18 Gb) gains nothing (its step is a chromosome against its homologue, which no range
                                    // started from an empty arm list joins: 1 993 ms either way) and pays for the cuts that fail in its
                                    // first call (8.4 instead of 5.7 s).  0: every segment is one work item
    int64_t split_len = 0;          // probes per range (about: the ranges of a segment are of one length); 0: the smallest of 2048 ... 32768
                                    // for which the call's runs stay within split_runs (split_tally_kernel), warm-up as long (2048 .. 6144),
                                    // shortest segment cut twice it -- a small job gets short ranges, a genome-sized one long ones
    int64_t split_runs = 224;       // ... that budget: every run holds a compute unit while it runs
    int64_t split_warm = 6144;      // (split_len > 0) probes a range starts in front of its cut
    int64_t split_warm_max = 65536; // a segment with a cut that did not hold gets a longer warm-up in the next call -- as far back as the oldest
                                    // arm in front of the failed cut was born, or the longest this limit allows -- while that stays within two ranges
                                    // and this many probes (beyond: only the cuts that held are planned again); 0: never grown.  (A
                                    // repeat-rich GRCh38-sized input with megabase higher-order arrays, ranges of 24 576 probes: 24 of 24 cut
                                    // segments fail at 6 144 probes of warm-up, 13 at 12 288, 3 at 24 576, 1 at 49 152: 349 -> 256 ms per step)
    int64_t split_min = 0;          // segments shorter than this (probe positions) are not cut (split_len = 0: at least this)
    int64_t cache_calls = 2;        // the blocks an index build released stay in the block cache until the index has answered this many
                                    // search PASSES (a direct + -RC passes call counts two) (then, at its destruction, on an allocation failure and by asgart_trim_cache they go
                                    // back to the device): giving ~100 GB back costs the next allocation of the process 20-30 ms per
                                    // GiB on most boxes of the pool -- 1.3-2.4 s of a cold GRCh38-sized run when it happened at the end of
                                    // asgart_index_prepare (0: there, as in round 4); the second call builds the position-sorted lists
                                    // and the presence filters out of those blocks
    int64_t prewarm = 1;            // 1: asgart_index_prepare also reserves the per-probe workspace of both call contexts (sized for an
                                    // unsharded call over the whole text) and starts the worker thread of the passes call, so that the first
                                    // search calls allocate nothing chip-sized; 0: everything on first use (hosts that only issue sharded calls)
    int64_t test_stall_s = 0;       // tests: every search call first stalls its stream for this many seconds (exercises the watchdog)
    int64_t watchdog_s = 120;       // a search call whose device work makes no progress for this many seconds returns ASGART_E_HIP with the
                                    // last heartbeats of its kernels instead of waiting forever; 0: wait forever
};
int32_t create_ctx_streams(SearchCtx &cx);
template <class SlotT>
int32_t build_rank_lists_runs(const uint64_t *d_keys, const SlotT *d_sa, uint64_t n, SlotT *d_sap, uint32_t min_run, int k,
                              hipStream_t s);  // (sa_build.hip: only the runs of more than min_run equal keys; any slot width)  // the streams and events of one call context (current device)
int32_t option_set(Options &o, const char *name, int64_t value);  // ASGART_E_ARG: unknown name / bad value
void options_from_env(Options &o);
}  // namespace asgart

namespace asgart {
// A host thread that lives as long as its index and runs one job at a time: the passes of
// asgart_search_duplications_passes beyond the first (a thread's first HIP call sets up per-thread runtime state,
// which a thread created per call would pay on every step).
struct PassWorker {
    std::mutex m;
    std::condition_variable cv;
    std::function<void()> job;
    bool busy = false, stop = false;
    std::thread th;
    PassWorker() : th([this] { loop(); }) {}
    ~PassWorker() {
        {
            std::lock_guard<std::mutex> lk(m);
            stop = true;
        }
        cv.notify_all();
        th.join();
    }
    void loop() {
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
            cv.wait(lk, [&] { return stop || (busy && job); });
            if (stop) return;
            std::function<void()> j = std::move(job);
            job = nullptr;
            lk.unlock();
            j();
            lk.lock();
            busy = false;
            cv.notify_all();
        }
    }
    void submit(std::function<void()> j) {
        {
            std::lock_guard<std::mutex> lk(m);
            job = std::move(j);
            busy = true;
        }
        cv.notify_all();
    }
    void wait() {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return !busy; });
    }
};
}  // namespace asgart

struct asgart_index {
    int device = 0;
    int64_t n = 0;
    int64_t n_sa = 0;       // suffix-array slots (== n unless trimmed)
    bool trimmed = false;   // --trim index: SA of data[trim_start..trim_end] + '$', entries shifted by +trim_start
    int64_t trim_start = 0, trim_end = 0;
    int n_bad = 0;
    uint64_t bad[asgart::kMaxK + 2] = {};
    bool wide = false;  // 64-bit slots/positions
    uint8_t *d_text = nullptr;
    void *d_sa = nullptr;
    // k-specific
    uint64_t k = 0;
    int d = 0;
    uint64_t *d_keys = nullptr;
    void *d_ptab = nullptr;
    uint64_t ptab_entries = 0;  // slots the allocation of d_ptab holds (it is made when the index is created)
    void *d_c8lo = nullptr;
    void *d_c8hi = nullptr;
    void *d_sap = nullptr;   // position-sorted occurrence lists (IndexView::sap), or null
    uint64_t *d_filter[4] = {nullptr, nullptr, nullptr, nullptr};  // per orientation: reverse * 2 + complement
    uint64_t *d_pbits[4] = {nullptr, nullptr, nullptr, nullptr};   // ... its answers by text position (n bits + padding)
    bool filter_off[4] = {false, false, false, false};             // no memory for it: this orientation is searched without
    bool pbits_learn[4] = {false, false, false, false};            // its position bits are learned by the searches (RunParams::learn)
    uint64_t pbits_uses[4] = {0, 0, 0, 0};                         // ... search calls that have used them so far
    int filter_bits = 0;                                           // log2 of their size in bits
    uint32_t tail8[asgart::kMaxK];
    int n_tail8 = 0;
    uint64_t tail_bloom = 0;
    std::vector<uint8_t> h_tail;  // last kMaxK + 32 bytes of the text (host copy)
    bool sap_tried = false;        // the position-sorted lists were built or given up on for this probe size
    uint64_t calls_total = 0;      // finished search calls with the current keys ...
    uint64_t mode_calls[4] = {0, 0, 0, 0};  // ... and started ones per orientation (option lazy_aux)
    double ms_prepare = 0.0;
    std::atomic<bool> poisoned{false};  // a call gave up waiting for the device (watchdog): work may still be running on the
                                        // index's streams and buffers, every later call is refused
    double tail_ms[4] = {-1.0, -1.0, -1.0, -1.0};  // per orientation (reverse * 2 + complement): shortest extension time of an unsharded call so far
    asgart::Options opt;
    asgart::SearchCtx ctx[asgart::kNumCtx];
    int last_ctx = 0;  // context of the most recent search call (asgart_get_stats)
    std::mutex mu;
    std::condition_variable cv;
    // what the passes calls with one set of settings have measured so far (option fuse_passes = 1): whether ONE segment is
    // the extension of the job (the last call that ran as one job says), and -- only then -- the shortest call each way
    struct FuseVerdict {
        uint64_t k = 0, M = 0, C = 0, modes = 0;
        uint32_t G = 0;
        int32_t n_passes = 0, shard = 0, n_shards = 1;
        bool seen = false, pole = false;
        int32_t n_fused = 0, n_piped = 0;      // calls timed each way
        int32_t unsettled = 0;                 // calls in a row that refused a cut (not samples, unless three in a row)
        double ms_fused = 1e30, ms_piped = 1e30;
    } fuse_verdict;
    // segments a cut of which did not hold (option split): orientation << 32 | first probe counted from the start of its pass,
    // under which settings and chunk list (sig), how many of their cuts, from the start, held (only those are planned again)
    // and at which range length.  Cleared with the keys; the oldest entries age out.
    struct SplitVerdict {
        uint64_t key, sig;
        uint32_t allowed, range_len;  // (the count belongs to the range length it was found at: one entry per segment and length)
        uint32_t warm;                // the warm-up this segment's ranges get from now on (0: the call's own)
    };
    std::vector<SplitVerdict> split_blocked;
    asgart::DevBuf ws_arena;  // the block the call contexts' per-probe buffers were carved from (carve_probe_workspace), or empty
    std::mutex pass_mu;  // one asgart_search_duplications_passes call at a time per index
    std::vector<std::unique_ptr<asgart::PassWorker>> pass_workers;

    // one free context for a search call / all contexts for calls that change shared state
    asgart::SearchCtx &acquire_one(int *which) {
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            for (int i = 0; i < asgart::kNumCtx; ++i)
                if (!ctx[i].busy) {
                    ctx[i].busy = true;
                    *which = i;
                    return ctx[i];
                }
            cv.wait(lk);
        }
    }
    void acquire_all() {
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            bool any = false;
            for (auto &c : ctx) any |= c.busy;
            if (!any) break;
            cv.wait(lk);
        }
        for (auto &c : ctx) c.busy = true;
    }
    void release_one(int which) {
        {
            std::lock_guard<std::mutex> lk(mu);
            ctx[which].busy = false;
            last_ctx = which;
        }
        cv.notify_all();
    }
    void release_all() {
        {
            std::lock_guard<std::mutex> lk(mu);
            for (auto &c : ctx) c.busy = false;
        }
        cv.notify_all();
    }

    template <class SlotT>
    asgart::IndexView<SlotT> view() const {
        asgart::IndexView<SlotT> v;
        v.text = d_text;
        v.sa = reinterpret_cast<const SlotT *>(d_sa);
        v.keys = d_keys;
        v.ptab = reinterpret_cast<const SlotT *>(d_ptab);
        v.c8lo = reinterpret_cast<const SlotT *>(d_c8lo);
        v.c8hi = reinterpret_cast<const SlotT *>(d_c8hi);
        v.n = (uint64_t)n;
        v.k = (int)k;
        v.kk = (int)std::min<uint64_t>(k, (uint64_t)asgart::kMaxKey);
        v.k2 = v.k - v.kk;
        v.d = d;
        for (int j = 0; j < asgart::kMaxK; ++j) v.tail8[j] = tail8[j];
        v.n_tail8 = n_tail8;
        v.tail_bloom = tail_bloom;
        v.sap = reinterpret_cast<const SlotT *>(d_sap);
        v.n_sa = (uint64_t)n_sa;
        v.trim = trimmed ? 1 : 0;
        v.n_bad = n_bad;
        for (int j = 0; j < asgart::kMaxK + 2; ++j) v.bad[j] = bad[j];
        return v;
    }
};

struct asgart_families {
    std::vector<uint64_t> fam_offsets;  // n_fam + 1
    std::vector<uint64_t> fam_keys;     // n_fam: (first probe of the family's segment << 32) | family ordinal inside it --
                                        // ascending == reference order; what a gatherer merges sharded results by
    std::vector<asgart_proto_sd> sds;
};

namespace asgart {
// An index whose watchdog gave up (or whose teardown did) may still have work running on its streams and buffers: every
// entry point that would touch them refuses it.
#define REFUSE_POISONED(idx)                                                                                              \
    do {                                                                                                                  \
        if ((idx)->poisoned.load()) {                                                                                     \
            ::asgart::set_error("this index gave up waiting for the device in an earlier call (watchdog): work may still be " \
                                "running on its streams; destroy it and continue in a fresh process");                    \
            return ASGART_E_HIP;                                                                                          \
        }                                                                                                                 \
    } while (0)
// set by the passes call on the thread that holds asgart_index::pass_mu (index_prepare's prewarm must not try to lock a
// mutex its own thread owns: undefined for std::mutex)
extern thread_local bool tl_owns_pass_mu;
int32_t index_prepare(asgart_index *idx, uint64_t k);
// per-probe workspace of one call context for a window of W probes (pipeline.hip; also what run_search_t reserves)
int32_t reserve_probe_workspace(asgart_index *idx, SearchCtx &cx, uint64_t W);
// asgart_index_prepare: the per-probe workspace of ALL call contexts (context c for Wc[c] probes) carved out of ONE block that
// the suffix sorter has just released, if the block cache holds one of the right size (false: nothing done)
bool carve_probe_workspace(asgart_index *idx, const uint64_t *Wc);
int32_t index_prepare_filter(asgart_index *idx, uint64_t k, int mode);  // mode = reverse * 2 + complement
int32_t index_prepare_learned_bits(asgart_index *idx, uint64_t k, int mode);  // a blank bitmap the searches fill in (lazy_aux)
int32_t index_prepare_sap(asgart_index *idx, uint64_t k);
int32_t run_search_passes(asgart_index *idx, const uint64_t *chunks, int64_t n_chunks,
                          const asgart_settings *sts, int32_t n_passes, int32_t shard, int32_t n_shards, bool want_csr,
                          asgart_families *const *fams, std::vector<uint8_t> *status_out,
                          std::vector<uint64_t> *rowoff_out, std::vector<uint64_t> *hits_out,
                          volatile uint64_t *progress);
int32_t run_search(asgart_index *idx, const uint64_t *chunks, int64_t n_chunks,
                   const asgart_settings *st, int32_t shard, int32_t n_shards, bool want_csr,
                   asgart_families *fam_out, std::vector<uint8_t> *status_out,
                   std::vector<uint64_t> *rowoff_out, std::vector<uint64_t> *hits_out,
                   volatile uint64_t *progress);
// records -> reference order (g_start, fam_seq, create_seq), stable; result in w.rec_sorted
int32_t sort_records(Workspace &w, const SdRec *recs, uint64_t n, hipStream_t s);
int32_t sort_segments(Workspace &w, uint32_t *keys, uint32_t *vals, uint64_t n, hipStream_t s,
                      const uint32_t **sorted_vals, const uint32_t **sorted_keys, bool ties_by_value);
int32_t text_is_dna(const uint8_t *d_text, int64_t n, hipStream_t s, bool *dna);
int32_t sa_build_device(const uint8_t *d_text, int64_t n, void *d_sa, bool wide,
                        hipStream_t stream, uint64_t wide_batch);
}  // namespace asgart
