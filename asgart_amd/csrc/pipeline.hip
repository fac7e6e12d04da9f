// pipeline.hip -- host driver of the probe -> search -> extend pipeline and the
// remaining C ABI entry points.
//
// Replaces the body of SearchDuplications::run, reference
// src/bin/asgart.rs:201-253 (chunk fan-out, needle preparation, automaton,
// left fix-up, fold in chunk order).
#include "pipeline_dev.hpp"
#include "extend_fast_dev.hpp"
#include "extend_k8_dev.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <string>
#include <thread>

namespace asgart {

#ifdef ASGART_PROFILE_EXTEND
#define PROF_DUMP(tag)                                                                          \
    do {                                                                                       \
        fprintf(stderr, "[extend profile %s] stage=%llu batches=%llu | reg_cyc=%llu reg_probes=%llu | build=%llu "\
                "lds_probes=%llu match=%llu apply=%llu retire=%llu segs=%llu sumA=%llu sumCnt=%llu | longest: "   \
                "cyc=%llu g0=%llu lds_probes=%llu reg_probes=%llu sumA=%llu sumCnt=%llu stage=%llu match=%llu\n", \
                tag, h_ctr[16], h_ctr[17], h_ctr[18], h_ctr[19], h_ctr[20], h_ctr[21], h_ctr[22], h_ctr[23],  \
                h_ctr[24], h_ctr[25], h_ctr[26], h_ctr[27], h_ctr[28], h_ctr[29], h_ctr[30] >> 32,            \
                h_ctr[30] & 0xffffffffull, h_ctr[31] >> 32, h_ctr[31] & 0xffffffffull, h_ctr[32], h_ctr[33]); \
        fprintf(stderr, "    K7 steps=%llu; cycles [to barrier 1, wait, to barrier 2, wait]: control %llu %llu %llu %llu | arm wave 0 %llu %llu %llu %llu | last arm wave %llu %llu %llu %llu\n", \
                h_ctr[52], h_ctr[40], h_ctr[41], h_ctr[42], h_ctr[43], h_ctr[44], h_ctr[45], h_ctr[46], h_ctr[47], h_ctr[48], h_ctr[49], h_ctr[50], h_ctr[51]); \
        fprintf(stderr, "    K7 wave 0: extra row rounds %llu, stash rounds %llu, arms offered cooperatively %llu, resolved cooperatively %llu\n", h_ctr[53], h_ctr[56], h_ctr[54], h_ctr[55]); \
        fprintf(stderr, "    K7 wave 0 cycles: A[cmd %llu resolve %llu offers %llu] B[mid %llu create+offers %llu publish %llu next-cmd+index %llu]\n", h_ctr[57], h_ctr[58], h_ctr[59], h_ctr[60], h_ctr[61], h_ctr[62], h_ctr[63]); \
        (void)hipMemsetAsync(d_ctr + 52, 0, 13 * 8, s);                                        \
        fprintf(stderr, "    P1: arm lookups=%llu linear=%llu chain nodes=%llu accepts=%llu by level:", h_ctr[40], h_ctr[41], h_ctr[42], h_ctr[43]); \
        for (int pf_i = 0; pf_i < 8; ++pf_i) fprintf(stderr, " %llu", h_ctr[44 + pf_i]);       \
        fprintf(stderr, "\n");                                                                 \
        (void)hipMemsetAsync(d_ctr + 40, 0, 12 * 8, s);                                        \
        fprintf(stderr, "    longest slots:");                                                  \
        for (int pf_i = 0; pf_i < 12; ++pf_i) fprintf(stderr, " [%d]=%llu", pf_i, h_ctr[56 + pf_i]); \
        fprintf(stderr, "\n");                                                                 \
        fprintf(stderr, "    segments by peak arms (log2 bins):");                             \
        for (int pf_i = 0; pf_i < 16; ++pf_i) fprintf(stderr, " %llu", h_ctr[CT_HIST_PEAK + pf_i]); \
        fprintf(stderr, "\n    probes by segment length (log2 bins):");                        \
        for (int pf_i = 0; pf_i < 16; ++pf_i) fprintf(stderr, " %llu", h_ctr[CT_HIST_PROBES + pf_i]); \
        fprintf(stderr, "\n");                                                                 \
        (void)hipMemsetAsync(d_ctr + 16, 0, 18 * 8, s);                                        \
        (void)hipMemsetAsync(d_ctr + 56, 0, (CT_N1 - 56) * 8, s);                              \
    } while (0)
// one tier at a time, with its own dump (the diagnostic build gives up the overlap)
#define PROF_TIER(tag, stream, n)                                                               \
    do {                                                                                       \
        const auto pf_c0 = std::chrono::steady_clock::now();                                   \
        (void)hipStreamSynchronize(stream);                                                    \
        const double pf_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - pf_c0).count(); \
        (void)hipMemcpy(h_ctr, d_ctr, kCtrBytes, hipMemcpyDeviceToHost);                   \
        fprintf(stderr, "[tier %s] segments=%llu  %.2f ms\n", tag, (unsigned long long)(n), pf_ms); \
        PROF_DUMP(tag);                                                                        \
        (void)hipStreamSynchronize(s);                                                         \
    } while (0)
#else
#define PROF_DUMP(tag)
#define PROF_TIER(tag, stream, n)
#endif

// default launch order / grid sizes of the extension tiers (see the launch site)
constexpr uint64_t kGrid1 = 256ull * 8ull, kGrid2 = 256ull * 3ull;
// Arm slots of the arm-resident shapes (S arms per thread x NT threads; 64-bit positions hold fewer).
// tier 2, the one-wave shape: 8 x 64 = 512 live arms per wave, probes with up to 512 hits
template <class SlotT> constexpr int kWaveArmsLayers = sizeof(SlotT) == 4 ? 8 : 5;
// tiers 4 and 5: 4 arms per thread x 256 / 512 threads, cold fields in LDS
template <class SlotT> constexpr int kMidArmsLayers = sizeof(SlotT) == 4 ? 4 : 2;
constexpr int kWaveArmsHits = 512;
constexpr uint64_t kGrid2Arms = 256ull * 8ull;
// tier 6 on the one-barrier kernel (extend_fast_dev.hpp): 5 x 1024 arm slots (64-bit positions: 4 x 1024, a smaller table)
template <class SlotT> constexpr int kFastHeavyLayers = sizeof(SlotT) == 4 ? 5 : 4;
template <class SlotT> constexpr int kFastLongRows = sizeof(SlotT) == 4 ? 2048 : 1024;
// tier 3 and the runs over ranges on the kernel with specialised waves (extend_k8_dev.hpp): two of the sixteen waves hold
// no arms, so the shape has S x 896 slots: 5 x 896 (64-bit positions: 4 x 896)
template <class SlotT> constexpr int kK8LongLayers = sizeof(SlotT) == 4 ? 5 : 4;
constexpr uint32_t kK8LongSlots = 896;
// (A half shape -- 512 threads, 5 x 384 slots, 75 KB of LDS, two workgroups per compute unit -- was measured in round 6:
// 37 % less compute-unit time per hit-probe, 25 % more wall time, and the 208 segments of a GRCh38-sized step whose arms
// do not fit it fall to tier 6's kernel, one of them for 145 ms: the step doubled.  DESIGN_HISTORY.md.)
constexpr int kArmCapSmall = 256;   // live arms per wave in LDS, common case
constexpr uint32_t kTier1MaxSum = 20000;  // placement: busier segments never run on a single wave
constexpr int kArmCapMid = 768;     // second tier: block-cooperative kernel, 256 threads per segment
constexpr int kArmCapHybrid32 = 4608;   // tier 4: hot fields in LDS, (rs, le) in HBM scratch
constexpr int kArmCapHybrid64 = 3072;
constexpr int kArmCapBig32 = 2432;  // heavy tier, 32-bit positions: 3072*40 B + hits + scratch = 128 KiB
constexpr int kArmCapBig64 = 1664;  // heavy tier, 64-bit positions: 2048*60 B + hits + scratch = 132 KiB

static inline unsigned grid_for(uint64_t n, unsigned block = 256) {
    return (unsigned)((n + block - 1) / block);
}

// Test hook (option test_stall_s): a kernel that does nothing for that many seconds on the call's main stream, so
// that the watchdog's way out of a stuck device can be exercised (wall_clock64 ticks at 100 MHz).
__global__ void stall_kernel(unsigned long long ticks) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(100);
}

static uint32_t probes_in_chunk(uint64_t L, uint64_t k, uint64_t step, uint64_t M) {
    // loop of src/automaton.rs:92-97: `while i < L - k - step { i += step; ... }`
    if (L < M || L < k + step || L - k - step == 0) return 0;
    return (uint32_t)((L - k - step + step - 1) / step);
}

// Waits for everything queued on stream `st`.  With option watchdog_s > 0 the wait polls: while the heartbeat block of
// the call's extension workgroups keeps changing the device is making progress, however long it takes; a wait that
// outlasts watchdog_s seconds without any change gives up: the call returns ASGART_E_HIP naming the phase and the
// workgroups that were in flight, and the index refuses further calls (its streams may still hold the stuck work) --
// the host should report and exit, or go on in a fresh process.  (One GPU test run of round 3 sat in a call for 40
// minutes with nothing to tell where.)
struct Watchdog {
    asgart_index *idx;
    SearchCtx &cx;
    std::chrono::steady_clock::time_point t0, t_change;
    unsigned long long sig = 0;
    Watchdog(asgart_index *i, SearchCtx &c) : idx(i), cx(c), t0(std::chrono::steady_clock::now()), t_change(t0) {}
    unsigned long long signature() const {
        unsigned long long h = 1469598103934665603ull;
        if (cx.h_hb) {
            const volatile unsigned long long *p = cx.h_hb;
            for (int i = 0; i < SearchCtx::kHbTiers * SearchCtx::kHbSlots * 2; ++i) h = (h ^ p[i]) * 1099511628211ull;
        }
        return h;
    }
    // One look at the heartbeats (never blocks).  true: nothing has changed for watchdog_s seconds -- the error is set and
    // the index poisoned; the caller returns ASGART_E_HIP.
    bool expired(const char *phase) {
        const int64_t limit_s = idx->opt.watchdog_s;
        if (limit_s <= 0) return false;
        const auto now = std::chrono::steady_clock::now();
        const unsigned long long s2 = signature();
        if (s2 != sig) {
            sig = s2;
            t_change = now;
        }
        if (std::chrono::duration<double>(now - t_change).count() <= (double)limit_s) return false;
        std::string who;
        if (cx.h_hb) {
            int shown = 0;
            for (int t = 0; t < SearchCtx::kHbTiers && shown < 12; ++t)
                for (int w = 0; w < SearchCtx::kHbSlots && shown < 12; ++w) {
                    const unsigned long long a = cx.h_hb[2 * (t * SearchCtx::kHbSlots + w)], b = cx.h_hb[2 * (t * SearchCtx::kHbSlots + w) + 1];
                    if (!(a >> 63)) continue;
                    char buf[96];
                    snprintf(buf, sizeof buf, "%s tier %d wg %d: segment at probe %llu, position %llu", shown ? ";" : "", t, w,
                             a & 0xFFFFFFFFull, b);
                    who += buf;
                    ++shown;
                }
        }
        idx->poisoned.store(true);
        set_error("watchdog: no progress for %lld s while waiting for %s (%.1f s into the wait); last heartbeats:%s -- the index is "
                  "unusable from here on (option watchdog_s = 0 waits forever)", (long long)limit_s, phase,
                  std::chrono::duration<double>(now - t0).count(), who.empty() ? " none" : who.c_str());
        return true;
    }
};

static int32_t wd_sync(asgart_index *idx, SearchCtx &cx, hipStream_t st, const char *phase) {
    if (idx->opt.watchdog_s <= 0) {
        HIP_TRY(hipStreamSynchronize(st));
        return 0;
    }
    Watchdog wd(idx, cx);
    for (unsigned spins = 0;; ++spins) {
        const hipError_t q = hipStreamQuery(st);
        if (q == hipSuccess) return 0;
        if (q != hipErrorNotReady) HIP_TRY(q);
        (void)hipGetLastError();
        if (spins < 64) continue;                    // (short waits: no sleep at all)
        std::this_thread::sleep_for(std::chrono::microseconds(spins < 2000 ? 20 : 500));
        if ((spins & 1023u) == 0 && wd.expired(phase)) return ASGART_E_HIP;
    }
}
// ... for an event (a stream may have more queued behind it)
static int32_t wd_event_sync(asgart_index *idx, SearchCtx &cx, hipEvent_t ev, const char *phase) {
    if (idx->opt.watchdog_s <= 0) {
        HIP_TRY(hipEventSynchronize(ev));
        return 0;
    }
    Watchdog wd(idx, cx);
    for (unsigned spins = 0;; ++spins) {
        const hipError_t q = hipEventQuery(ev);
        if (q == hipSuccess) return 0;
        if (q != hipErrorNotReady) HIP_TRY(q);
        (void)hipGetLastError();
        if (spins < 64) continue;
        std::this_thread::sleep_for(std::chrono::microseconds(spins < 2000 ? 20 : 500));
        if ((spins & 1023u) == 0 && wd.expired(phase)) return ASGART_E_HIP;
    }
}

int32_t reserve_probe_workspace(asgart_index *idx, SearchCtx &cx, uint64_t W) {
    Workspace &w = cx.ws;
    const uint64_t n_blk = (W + kScanTile - 1) / kScanTile + 4;  // (every window of a call ends with a partial tile)
    RC_TRY(w.p_lo.reserve((size_t)W * (idx->wide ? 8 : 4)));
    RC_TRY(w.p_raw.reserve((size_t)W * 4));
    RC_TRY(w.p_filt.reserve((size_t)W * 4));
    RC_TRY(w.row_off.reserve(((size_t)W + 1) * 8));
    RC_TRY(w.blk.reserve((size_t)n_blk * sizeof(ScanEl)));
    RC_TRY(w.big_list.reserve((size_t)W * 4));
    if (idx->d_sap) RC_TRY(w.rank_list.reserve((size_t)W * 4));
    RC_TRY(w.counters.reserve(CT_COUNT * 8));
    return 0;
}

// On the slow boxes of the pool every first allocation costs ~30 ms per GiB: the nine buffers of 1-5 GB that the two
// contexts' workspace used to allocate at asgart_index_prepare were 0.8 s of a cold run there.  The sorter's key
// buffers (n 64-bit words each) are in the block cache at that moment: one of them holds both contexts' workspace.
bool carve_probe_workspace(asgart_index *idx, const uint64_t *Wc) {
    if (idx->ws_arena.p) return false;
    size_t off = 0;
    struct Piece {
        DevBuf *b;
        size_t off, bytes;
    };
    std::vector<Piece> pieces;
    for (int c = 0; c < kNumCtx; ++c) {
        Workspace &w = idx->ctx[c].ws;
        const uint64_t W = Wc[c];
        const uint64_t n_blk = (W + kScanTile - 1) / kScanTile + 4;
        DevBuf *bufs[7] = {&w.p_lo, &w.p_raw, &w.p_filt, &w.row_off, &w.blk, &w.big_list, &w.rank_list};
        const size_t want[7] = {(size_t)W * (idx->wide ? 8 : 4), (size_t)W * 4, (size_t)W * 4, ((size_t)W + 1) * 8,
                                (size_t)n_blk * sizeof(ScanEl), (size_t)W * 4, (size_t)W * 4};
        for (int j = 0; j < 7; ++j) {
            if (bufs[j]->p) return false;  // (something is reserved already: leave everything as it is)
            const size_t sz = (want[j] + 4095) & ~(size_t)4095;
            pieces.push_back(Piece{bufs[j], off, sz});
            off += sz;
        }
    }
    size_t cap = 0;
    void *p = BlockCache::take(off, &cap, 30);
    if (!p) return false;
    idx->ws_arena.p = p;
    idx->ws_arena.cap = cap;
    for (const Piece &pc : pieces) {
        pc.b->p = static_cast<char *>(p) + pc.off;
        pc.b->cap = pc.bytes;
        pc.b->view = true;
    }
    return true;
}

// One job over the probes of n_passes passes (orientations) of the same chunk list: sts[p] differ in reverse /
// complement only (the caller has checked); fams[p] receives pass p's families (null: none wanted -- the CSR surface).
// The stages of a call, in the order run() takes them:
//   setup          chunk table, probe numbering, this shard's own range
//   set_window     the windows of this attempt (halos grow on retry), RunParams
//   front          probe search, row offsets + segment starts, hit rows            (the HBM-bound, chip-wide part)
//   csr_out        asgart_probe_hits only: the per-probe hit rows to the host
//   place          per segment: arm bound -> tier, barren tests, long segments cut into ranges, lists sorted by cost
//   run_tiers      every extension tier and the runs over ranges launched together; early re-runs of tiers 3 and 6
//   join_ranges    cuts checked, family ordinals of the ranges chained, refused segments run again
//   finish_tiers   statistics, the cascade of what overflowed
//   records        records -> reference order -> families per pass on the host
//   fill_stats
template <class SlotT>
struct SearchCall {

    // ---- the call ---------------------------------------------------------------------------------------------------------
    asgart_index *const idx;
    SearchCtx &cx;
    const uint64_t *const chunks;
    const int64_t n_chunks_pass;
    const asgart_settings *const sts;
    const int32_t n_passes, shard, n_shards;
    const bool want_csr;
    asgart_families *const *const fams;
    std::vector<uint8_t> *const status_out;
    std::vector<uint64_t> *const rowoff_out, *const hits_out;
    Workspace &w;
    const hipStream_t s;
    const asgart_settings *const st;
    const bool fam_out;
    const int64_t n_chunks;      // entries of the chunk table: the chunk list once per pass
    const uint64_t k, step, n;
    const Options opt;           // options cannot change while this call holds a context
    static constexpr size_t kCtrBytes = (size_t)CT_COUNT * 8;
    static constexpr size_t kSplitMirror = 512 << 10;  // (option split: host copy of Workspace::split_buf)
    static constexpr int caph = sizeof(SlotT) == 4 ? kArmCapHybrid32 : kArmCapHybrid64;
    // ---- set up once (setup) --------------------------------------------------------------------------------------------
    bool nothing_to_do = true;
    size_t ch_bytes = 0;
    unsigned long long *h_ctr = nullptr, *h_scalar = nullptr;  // pinned: the counters read back; source of small host-to-device updates
    uint64_t *h_start = nullptr, *h_len = nullptr;
    uint32_t *h_pbase = nullptr;
    char *h_split = nullptr;
    std::vector<uint32_t> lp;    // probes of the chunk list, counted from the start of a pass
    uint32_t Ppass = 0, P = 0, own_lo = 0, own_hi = 0, lo_lim = 0, hi_lim = 0, K = 0;
    uint64_t look_back = 0, look_ahead = 0, call_sig = 0;
    uint64_t *d_start = nullptr, *d_len = nullptr;
    uint32_t *d_pbase = nullptr;
    std::chrono::steady_clock::time_point t_host0, t_launch;
    // ---- per window (set_window, front) ---------------------------------------------------------------------------------
    RunParams rp;
    uint32_t W = 0, W_probes = 0;
    unsigned long long *d_ctr = nullptr;
    IndexView<SlotT> ix;
    SlotT *p_lo = nullptr, *hits = nullptr;
    uint32_t *p_raw = nullptr, *p_filt = nullptr, *seg_list = nullptr;
    unsigned long long *row_off = nullptr;
    uint64_t total_hits = 0, n_seg = 0, n_overflow = 0, n_heavy = 0;
    bool progress_given = false;
    // ---- placement (place) ----------------------------------------------------------------------------------------------
    uint64_t rec_cap = 0;
    const SdRec *h_recs = nullptr;  // sorted records on the host (pinned staging of the context)
    size_t n_hrec = 0;
    bool arms_kernel = false, split_on = false;
    uint32_t tier_cap[kTiers + 1] = {};
    char *d_split = nullptr;
    const uint32_t *order = nullptr;
    uint64_t n_t[kTiers] = {}, seg_off[kTiers + 1] = {};
    uint64_t h_tp[kTiers] = {}, h_th[kTiers] = {};  // (option debug: hit-probes and hits per tier, as placed)
    const unsigned long long *h_split_hdr = nullptr;
    uint32_t n_runs = 0, n_cuts = 0, n_splits = 0;
    std::vector<SplitSeg> split_segs;
    uint32_t *ovf[kTiers] = {};  // ovf[t-1]: segments tier t gave up on
    uint64_t heavy_cap64 = 0;
    unsigned n_wg7 = 1;
    size_t region = 0;
    char *scratch6 = nullptr, *scratch7 = nullptr;
    // ---- the tiers (run_tiers .. finish_tiers) ---------------------------------------------------------------------------
    ExtParams<SlotT> ep;
    char *scratch_override = nullptr;  // HBM slices of an early cascade launch
    hipStream_t tier_stream[kTiers + 1] = {};
    uint64_t early_n[kTiers + 1] = {};
    double ms_tier2 = 0.0, ms_longest_tier = 0.0, ms_longest_segment = 0.0;
    uint64_t n_split_segments = 0, n_split_refused = 0;

    SearchCall(asgart_index *idx_, SearchCtx &cx_, const uint64_t *chunks_, int64_t n_chunks_pass_, const asgart_settings *sts_,
               int32_t n_passes_, int32_t shard_, int32_t n_shards_, bool want_csr_, asgart_families *const *fams_,
               std::vector<uint8_t> *status_out_, std::vector<uint64_t> *rowoff_out_, std::vector<uint64_t> *hits_out_)
        : idx(idx_), cx(cx_), chunks(chunks_), n_chunks_pass(n_chunks_pass_), sts(sts_), n_passes(n_passes_), shard(shard_),
          n_shards(n_shards_), want_csr(want_csr_), fams(fams_), status_out(status_out_), rowoff_out(rowoff_out_),
          hits_out(hits_out_), w(cx_.ws), s(cx_.stream), st(&sts_[0]), fam_out(fams_ != nullptr),
          n_chunks(n_chunks_pass_ * (int64_t)n_passes_), k(sts_[0].probe_size), step(sts_[0].probe_size / 2),
          n((uint64_t)idx_->n), opt(idx_->opt) {}

    uint32_t pass_of_probe(uint32_t g) const { return std::min<uint32_t>(g / K, (uint32_t)n_passes - 1u); }
    uint32_t pass_offset(uint32_t g) const { return lo_lim + (g - pass_of_probe(g) * K); }  // from the start of its pass
    bool tier_enabled(int t) const { return t >= 1 && t <= kTiers && tier_cap[t] != 0; }
    double since_launch() const {
        return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_launch).count();
    }
    void signal_progress() {  // (only single-pass calls have a progress array)
        for (int64_t c = 0; c < n_chunks_pass; ++c)
            cx.progress[c] = (uint64_t)probes_in_chunk(h_len[c], k, step, st->min_duplication_length) * step;
        progress_given = true;
    }

    // Workspace::split_buf: [4 counters (u64): runs, cuts, split segments, work cursor | runs | cuts | split segments |
    // run states | verdicts per cut | per run: what is added to its records' family ordinals | segments to run again | range
    // size chosen | what earlier calls found out about segments with a cut that did not hold]
    static constexpr uint32_t kMaxRuns = 4096, kMaxCuts = 2048, kMaxSplits = 1024;
    static constexpr size_t kOffRuns = 64, kOffCuts = kOffRuns + kMaxRuns * sizeof(RangeRun), kOffSplits = kOffCuts + kMaxCuts * 8,
                     kOffMeta = kOffSplits + kMaxSplits * sizeof(SplitSeg), kOffOk = kOffMeta + kMaxRuns * 64,
                     kOffFix = kOffOk + kMaxCuts * 4, kOffAgain = kOffFix + kMaxRuns * 4, kOffChoice = kOffAgain + kMaxSplits * 4,
                     kOffBlocked = kOffChoice + 128, kSplitBytes = kOffBlocked + kSplitBlockedMax * sizeof(BlockedSeg);
    static_assert(sizeof(SplitChoice) <= 128 && kOffChoice % 8 == 0, "split choice");
    static_assert(kSplitBytes <= kSplitMirror, "split mirror");

    // ---- chunk table, probe numbering, own range -------------------------------------------------------------------------
    int32_t setup() {
        // ---- chunk table -------------------------------------------------------
        // (in the call context's pinned control block: [counters | 32 scalars | start[nc] | len[nc] | pbase[nc + 1]])
        ch_bytes = (size_t)n_chunks * 16 + ((size_t)n_chunks + 1) * 4;
        void *ctl_p = nullptr;
        RC_TRY(cx.ctl(kCtrBytes + ch_bytes + 256 + 64 + kSplitMirror, &ctl_p));
        h_ctr = static_cast<unsigned long long *>(ctl_p);
        h_scalar = h_ctr + CT_COUNT;  // source of small host-to-device updates
        h_start = reinterpret_cast<uint64_t *>(static_cast<char *>(ctl_p) + kCtrBytes + 256);
        h_len = h_start + n_chunks;
        h_pbase = reinterpret_cast<uint32_t *>(h_len + n_chunks);
        h_split = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(h_pbase + n_chunks + 1) + 63u) & ~(uintptr_t)63u);
        const uint64_t text_end = (idx->h_tail.size() && idx->h_tail.back() == '$') ? n - 1 : n;
        // probes of the chunk list, counted from the start of a pass (every pass walks the same list)
        lp.assign((size_t)n_chunks_pass + 1, 0u);
        {
            uint64_t P64 = 0;
            for (int64_t c = 0; c < n_chunks_pass; ++c) {
                const uint64_t c_start = chunks[2 * c], c_len = chunks[2 * c + 1];
                if (c_start > text_end || c_len > text_end - c_start) {
                    set_error("chunk %lld = (%llu, %llu) exceeds the text (%llu bases before '$')",
                              (long long)c, (unsigned long long)c_start, (unsigned long long)c_len, (unsigned long long)text_end);
                    return ASGART_E_ARG;
                }
                lp[(size_t)c] = (uint32_t)P64;
                P64 += probes_in_chunk(c_len, k, step, st->min_duplication_length);
                if (P64 * (uint64_t)n_passes >= 0xFFFFFF00ull) {
                    set_error("more than 2^32 probes in one call");
                    return ASGART_E_CAP;
                }
            }
            lp[(size_t)n_chunks_pass] = (uint32_t)P64;
        }
        Ppass = lp[(size_t)n_chunks_pass];
        P = Ppass * (uint32_t)n_passes;
        cx.last_P = P;
        memset(&cx.stats, 0, sizeof(cx.stats));
        cx.stats.probes_total = P;
        if (fam_out)
            for (int32_t p = 0; p < n_passes; ++p) {
                fams[p]->fam_offsets.assign(1, 0);
                fams[p]->fam_keys.clear();
                fams[p]->sds.clear();
            }
        if (want_csr) {
            status_out->assign(P, 0);
            rowoff_out->assign((size_t)P + 1, 0);
            hits_out->clear();
        }
        if (P == 0 || n_chunks == 0) return 0;  // (nothing_to_do stays set)
        if (want_csr && (n_shards != 1 || n_passes != 1)) {
            set_error("asgart_probe_hits is one unsharded pass");
            return ASGART_E_ARG;
        }
        // Multi-GPU sharding: shard r owns the automaton segments that START in the r-th slice of EVERY pass's probe
        // sequence (chunk order inside each pass as in src/bin/asgart.rs:201-253).  It computes, per pass, probe-search over
        // that slice plus a look-back halo (to decide whether its first probes continue an earlier segment) and a look-ahead
        // halo (to finish segments that run past the slice): one WINDOW per pass, all passes' windows as ONE job -- one front,
        // one launch per extension tier.  No data is exchanged between shards.
        own_lo = (uint32_t)((uint64_t)Ppass * (uint64_t)shard / (uint64_t)n_shards);  // (from the start of a pass)
        own_hi = (uint32_t)((uint64_t)Ppass * (uint64_t)(shard + 1) / (uint64_t)n_shards);
        if (own_lo == own_hi) return 0;  // (nothing_to_do stays set)
        look_back = (uint64_t)opt.shard_lookback;
        look_ahead = opt.shard_lookahead > 0 ? (uint64_t)opt.shard_lookahead
                                                      : std::max<uint64_t>(65536, (own_hi - own_lo) / 16);
        // The chunks a window can touch: c0 .. c1 of the list (a segment ends with its chunk).  The call numbers its probes
        // VIRTUALLY (RunParams): these chunks keep their probes, the others are empty in the table it uploads -- pass p's kept
        // probes are [p * K, (p + 1) * K), so probe g of the call is the (lo_lim + g - p * K)-th of pass p = g / K.
        const int64_t c0 = (int64_t)(std::upper_bound(lp.begin(), lp.end(), own_lo) - lp.begin()) - 1;
        const int64_t c1 = (int64_t)(std::upper_bound(lp.begin(), lp.end(), own_hi - 1u) - lp.begin()) - 1;
        lo_lim = lp[(size_t)c0];
        hi_lim = lp[(size_t)c1 + 1];
        K = hi_lim - lo_lim;
        for (int64_t c = 0; c < n_chunks; ++c) {
            const int64_t cl = c % n_chunks_pass, p = c / n_chunks_pass;
            h_start[c] = chunks[2 * cl];
            h_len[c] = chunks[2 * cl + 1];
            h_pbase[c] = (uint32_t)p * K + (std::min(std::max(lp[(size_t)cl], lo_lim), hi_lim) - lo_lim);
        }
        h_pbase[n_chunks] = (uint32_t)n_passes * K;
        // what a remembered verdict about a cut segment belongs to: the settings and the chunk list (probe numbers mean nothing
        // under another step, minimum length or list)
        call_sig = 1469598103934665603ull;
        {
            auto mix = [&](uint64_t v) { call_sig = (call_sig ^ v) * 1099511628211ull; };
            mix(k);
            mix(st->max_gap_size);
            mix(st->min_duplication_length);
            mix(st->max_cardinality);
            for (int64_t c = 0; c < 2 * n_chunks_pass; ++c) mix(chunks[c]);
        }
        RC_TRY(w.chunks.reserve(ch_bytes));
        d_start = w.chunks.as<uint64_t>();
        d_len = d_start + n_chunks;
        d_pbase = reinterpret_cast<uint32_t *>(d_len + n_chunks);
        HIP_TRY(hipMemcpyAsync(d_start, h_start, ch_bytes, hipMemcpyHostToDevice, s));  // same layout on the device
        t_host0 = std::chrono::steady_clock::now();
        nothing_to_do = false;
        return 0;
    }

    // ---- the windows of this attempt ---------------------------------------------------------------------------------------
    void set_window() {
        // the window of a pass, counted from the start of the pass
        const uint32_t w_lo = (uint32_t)std::max<uint64_t>(lo_lim, own_lo > look_back ? own_lo - look_back : 0);
        const uint32_t w_hi = (uint32_t)std::min<uint64_t>(hi_lim, (uint64_t)own_hi + look_ahead);
        rp.init_unknown = w_lo != lo_lim ? 1u : 0u;
        rp.ch = ChunkTable{d_start, d_len, d_pbase, (int)n_chunks};
        rp.win_len = w_hi - w_lo;
        rp.win_stride = K;
        rp.g_lo = w_lo - lo_lim;
        rp.g_hi = rp.g_lo + ((uint32_t)n_passes - 1u) * K + rp.win_len;
        rp.own_off_lo = own_lo - w_lo;
        rp.own_off_hi = own_hi - w_lo;
        W = rp.g_hi - rp.g_lo;                     // extent of the per-probe arrays
        W_probes = (uint32_t)n_passes * rp.win_len;  // probes the call computes
        rp.k = (int)k;
        rp.step = (int)step;
        rp.G = st->max_gap_size;
        rp.tstar = (uint32_t)((st->max_gap_size + step - 1) / step);
        if (rp.tstar == 0) rp.tstar = 1;
        rp.M = st->min_duplication_length;
        rp.C = st->max_cardinality > 0xFFFFFF00ull ? 0xFFFFFF00u : (uint32_t)st->max_cardinality;
        rp.n_passes = (uint32_t)n_passes;
        rp.pass_chunks = (uint32_t)n_chunks_pass;
        rp.modes = 0;
        rp.learn = rp.blank = 0;
        rp.flt_bits = idx->filter_bits;
        for (int p = 0; p < 4; ++p) {
            rp.flt[p] = rp.pbits[p] = nullptr;
            if (p >= n_passes) continue;
            const int mode = (sts[p].reverse ? 2 : 0) | (sts[p].complement ? 1 : 0);
            rp.modes |= (uint32_t)mode << (8 * p);
            rp.flt[p] = idx->d_filter[mode];  // null: filter off
            rp.pbits[p] = opt.posbits ? idx->d_pbits[mode] : nullptr;
            if (rp.pbits[p] && idx->pbits_learn[mode]) {
                rp.learn |= 1u << p;
                if (idx->pbits_uses[mode] == 0) rp.blank |= 1u << p;  // (this call finds them all ones)
            }
        }
        cx.last_rp = rp;
        cx.has_last = false;

    }

    // ---- probe search, row offsets + segment starts, hit rows (*ambiguous: a start decision needs a longer look-back) ----------
    int32_t front(bool *ambiguous) {
        // ---- workspace (indexed by absolute probe number through shifted pointers) ---
        const uint32_t n_blk = rp.n_tiles((uint32_t)kScanTile);
        const uint64_t seg_cap = (uint64_t)W_probes / (rp.tstar + 1) + (uint64_t)n_chunks + 64;
        RC_TRY(reserve_probe_workspace(idx, cx, W));
        RC_TRY(w.seg_list.reserve((size_t)seg_cap * 4));
        RC_TRY(w.counters.reserve(CT_COUNT * 8));
        d_ctr = w.counters.as<unsigned long long>();
        HIP_TRY(hipMemsetAsync(d_ctr, 0, CT_COUNT * 8, s));
        ix = idx->template view<SlotT>();
        p_lo = w.p_lo.as<SlotT>() - rp.g_lo;
        p_raw = w.p_raw.as<uint32_t>() - rp.g_lo;
        p_filt = w.p_filt.as<uint32_t>() - rp.g_lo;
        row_off = w.row_off.as<unsigned long long>() - rp.g_lo;
        unsigned long long *scan_desc = w.blk.as<unsigned long long>();  // two words per scan tile (scan_segments_kernel)
        uint32_t *big_list = w.big_list.as<uint32_t>();
        uint32_t *rank_list = w.rank_list.as<uint32_t>();
        seg_list = w.seg_list.as<uint32_t>();

        if (opt.test_stall_s > 0) stall_kernel<<<1, 64, 0, s>>>((unsigned long long)opt.test_stall_s * 100000000ull);
        // ---- K1: probe search + filtered counts -----------------------------------
        HIP_TRY(hipEventRecord(cx.ev[0], s));
        probe_count_kernel<SlotT, false><<<rp.n_tiles((uint32_t)kProbeBlock), kProbeThreads, 0, s>>>(
            ix, rp, p_lo, p_raw, p_filt, big_list, rank_list, d_ctr);
        HIP_TRY(hipEventRecord(cx.ev[11], s));
        collect_pending_kernel<<<std::min<uint32_t>(rp.n_tiles((uint32_t)kCollectTile), 256u * 8u), kCollectBlock, 0, s>>>(
            rp, p_filt, big_list, rank_list, d_ctr);
        big_count_kernel<SlotT, false><<<2048, 256, 0, s>>>(ix, rp, p_lo, p_raw, p_filt, big_list, d_ctr);
        // (after big_count_kernel: what it appends to big_list is for the fill only)
        if (ix.sap)
            rank_count_kernel<SlotT, false><<<2048, 256, 0, s>>>(ix, rp, p_lo, p_raw, p_filt, rank_list, big_list, d_ctr);
        HIP_TRY(hipEventRecord(cx.ev[1], s));
        // ---- K2: scans + segmentation ----------------------------------------------
        HIP_TRY(hipMemsetAsync(scan_desc, 0, (size_t)n_blk * 16, s));
        scan_segments_kernel<<<std::min<uint32_t>(n_blk, 256u * 2u), kScanBlock, 0, s>>>(rp, p_filt, scan_desc, n_blk, row_off, seg_list, d_ctr);
        HIP_TRY(hipEventRecord(cx.ev[2], s));
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(h_ctr, d_ctr, kCtrBytes, hipMemcpyDeviceToHost, s));
        RC_TRY(wd_sync(idx, cx, s, "the probe search and the scans"));
        total_hits = h_ctr[CT_TOTAL_HITS];
        n_seg = h_ctr[CT_SEG];
        if (h_ctr[CT_AMBIG]) {  // a start decision needs more history: the caller widens the look-back halo
            *ambiguous = true;
            return 0;
        }
        if (n_seg > seg_cap) {
            set_error("internal: segment list overflow (%llu > %llu)", (unsigned long long)n_seg,
                      (unsigned long long)seg_cap);
            return ASGART_E_CAP;
        }
        // ---- K3: CSR fill -----------------------------------------------------------
        // On a stream of its own (idle until the tiers are launched; the main stream has just been drained): the placement
        // walk, which reads the per-probe counts only, runs beside it -- the fill is bound by the suffix-array intervals it
        // gathers, the walk by its round trips.  Whoever needs the hit rows waits for ev[3].
        RC_TRY(w.hits.reserve((size_t)(total_hits + 64) * sizeof(SlotT)));
        hits = w.hits.as<SlotT>();
        hipStream_t sf = cx.fill_stream;
        HIP_TRY(hipEventRecord(cx.ev[16], sf));
        fill_small_kernel<SlotT><<<rp.n_tiles(256u), 256, 0, sf>>>(ix, rp, p_lo, p_raw, p_filt, row_off, hits);
        if (h_ctr[CT_BIG])
            fill_big_kernel<SlotT><<<2048, 256, 0, sf>>>(ix, rp, p_lo, p_raw, p_filt, row_off, hits,
                                                         big_list, d_ctr);
        HIP_TRY(hipEventRecord(cx.ev[3], sf));
        HIP_TRY(hipGetLastError());

        return 0;
    }

    // ---- asgart_probe_hits: the per-probe hit rows to the host ----------------------------------------------------------------
    int32_t csr_out() {
        std::vector<uint32_t> h_filt(P);
        std::vector<SlotT> h_hits((size_t)total_hits);
        RC_TRY(wd_sync(idx, cx, s, "the hit rows"));  // (the copies below go to pageable memory: they block inside the copy)
        HIP_TRY(hipMemcpyAsync(h_filt.data(), p_filt, (size_t)P * 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(rowoff_out->data(), row_off, ((size_t)P + 1) * 8,
                               hipMemcpyDeviceToHost, s));
        if (total_hits)
            HIP_TRY(hipMemcpyAsync(h_hits.data(), hits, (size_t)total_hits * sizeof(SlotT),
                                   hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        for (uint32_t g = 0; g < P; ++g)
            (*status_out)[g] = h_filt[g] == kSkipN ? 1 : (h_filt[g] == kSkipCard ? 2 : 0);
        hits_out->resize((size_t)total_hits);
        for (uint64_t j = 0; j < total_hits; ++j) (*hits_out)[j] = h_hits[j];
        return 0;
    }

    // ---- placement: per-segment work estimate -> tier, longest first; barren segments; long segments as ranges ----------------
    int32_t place() {
        rec_cap = std::max<uint64_t>(1u << 18, w.fam_sds.cap / sizeof(SdRec));
        RC_TRY(w.ovf_list.reserve((size_t)(n_seg + 1) * 4 * kTiers));
        h_recs = nullptr;
        n_hrec = 0;
        // ---- placement: per-segment work estimate -> tier, longest first --------------------
        // option force_tier = t (tests): start every segment with a multi-hit probe in tier >= t
        const int force_tier = (int)opt.force_tier;
        RC_TRY(w.seg_keys.reserve((size_t)n_seg * 4 * 2));
        RC_TRY(w.seg_vals.reserve((size_t)n_seg * 4 * 2));
        uint32_t *kbuf = w.seg_keys.as<uint32_t>(), *vbuf = w.seg_vals.as<uint32_t>();
        HIP_TRY(hipMemsetAsync(d_ctr + CT_N1, 0, (size_t)(CT_COUNT - CT_N1) * 8, s));
        // ---- the tiers ------------------------------------------------------------------------
        //  1  one wave per segment, arms in registers / LDS arrays (extend_kernel)                 <= 256 arms
        //  2  arm-resident, one wave, 8 per CU (K6 8x64)                                           <= 512
        //  3  arm-resident, specialised waves, 1024 threads (K8 5x896): the LONG DENSE segments    <= 4480 * 1.6 (by the bound)
        //  4  arm-resident, 256 threads, 4 workgroups per CU (K6 4x256)                            <= 1024
        //  5  arm-resident, 512 threads, 2 per CU (K6 4x512)                                       <= 2048
        //  6  arm-resident, 1024 threads, 1 per CU (K6 5x1024): the long sparse segments           <= 5120 * 1.4 (by the bound)
        //  7  arms in HBM scratch (extend_heavy_kernel MODE 2)                                     any
        // (64-bit positions: 5x64 / 4x896 / 2x256 / 2x512 / 4x1024.)
        // Streams: the short chip-wide kernels on the call's high-priority main stream; tiers 1..6 on six
        // low-priority streams of their own, tier 7 behind tier 2.
        // With max_cardinality > 1024 (or ASGART_ARMS_KERNEL=0, tests) the LDS-array kernels (extend_heavy_kernel) take
        // tiers 2, 4 and 6 (768 / 2432 / 4608 * 1.4 arms) and tiers 3 and 5 stay empty; the small
        // shapes 2 and 4 stage 512 hits per probe and are skipped when max_cardinality > 512.
        // (the arm-resident kernels pack a position into 42 bits of a table entry)
        arms_kernel = rp.C <= (uint64_t)kHitBatch && opt.arms_kernel != 0 && (uint64_t)idx->n < (1ull << 42);
        const bool arms_small = arms_kernel && rp.C <= (uint64_t)kWaveArmsHits;
        for (int t = 0; t <= kTiers; ++t) tier_cap[t] = 0;
        tier_cap[1] = kArmCapSmall;
        tier_cap[kTiers] = 0xFFFFFFFFu;  // (tier 7 takes whatever is left)
        if (arms_kernel) {
            if (arms_small) {
                tier_cap[2] = (uint32_t)kWaveArmsLayers<SlotT> * 64u;
                tier_cap[4] = (uint32_t)((uint64_t)(kMidArmsLayers<SlotT> * 256) * (uint64_t)opt.cap45_pct / 100u);
            }
            tier_cap[3] = (uint32_t)kK8LongLayers<SlotT> * kK8LongSlots;
            tier_cap[5] = (uint32_t)((uint64_t)(kMidArmsLayers<SlotT> * 512) * (uint64_t)opt.cap45_pct / 100u);
            // the window bound is pessimistic for tandem arrays (hits extend arms there) and the HBM
            // tier is several times slower per probe: tier 6 also takes segments whose bound exceeds
            // its capacity by up to 40 % (a real overflow falls through the cascade)
            // (64-bit positions on the one-barrier kernel: the HBM tier is an order of magnitude slower per probe and the
            // bound three to four times what a segment really holds -- at cfg5 every segment that went to tier 7 by its
            // bound peaked below 4 096 arms: tier 6 accepts up to cap6w_pct of its capacity)
            const uint64_t pct6 = sizeof(SlotT) == 8 ? (uint64_t)opt.cap6w_pct : (uint64_t)opt.cap6_pct;
            tier_cap[6] = (uint32_t)((uint64_t)(kFastHeavyLayers<SlotT> * 1024) * pct6 / 100u);
            // tier 3 accepts what tier 6 would accept by the bound (a long segment is no less safe there), but
            // never more than the same allowance over its own capacity (with 64-bit positions it holds fewer
            // arms than tier 6, and what it gives up on is re-run from the start)
            // (the bound of a tandem array is three to four times what it really holds: the one-barrier kernel, a fifth
            // faster per probe on such segments, takes them up to cap3_pct of its capacity)
            tier_cap[3] = std::min<uint32_t>(std::max(tier_cap[3], tier_cap[6]),
                                             (uint32_t)((uint64_t)tier_cap[3] * (uint64_t)opt.cap3_pct / 100u));
        } else {
            tier_cap[2] = kArmCapMid;
            tier_cap[4] = sizeof(SlotT) == 4 ? kArmCapBig32 : kArmCapBig64;
            tier_cap[6] = (rp.G >= 0xFFF0u || rp.C >= 0xFFF0u) ? 0u : (uint32_t)caph * 7 / 5;  // 16-bit gap/pend
        }
        PlaceParams pp;
        pp.long3 = pp.long3_big = pp.dense3 = pp.dense6 = 0;
        pp.stats = opt.debug ? 1u : 0u;
        pp.barren = opt.barren ? 1u : 0u;
        pp.seg_info = nullptr;
        const bool cluster_barren = opt.barren >= 2 && rp.M > (uint64_t)k;
        // long segments as ranges side by side (option split; plan_ranges_kernel in pipeline_dev.hpp): the long shape of the
        // one-barrier kernel; also in a sharded call (the segments its window cuts short are left alone: only
        // a segment whose end the placement walk has seen is cut)
        split_on = opt.split >= (sizeof(SlotT) == 4 ? 1 : 2) && arms_kernel && (opt.split_len == 0 || opt.split_len >= 64);
        if (cluster_barren || split_on) {
            RC_TRY(w.seg_info.reserve((size_t)n_seg * sizeof(uint2)));
            pp.seg_info = w.seg_info.as<uint2>();
        }
        d_split = nullptr;
        if (split_on) {
            RC_TRY(w.split_buf.reserve(kSplitBytes));
            d_split = w.split_buf.as<char>();
            HIP_TRY(hipMemsetAsync(d_split, 0, 64, s));
        }
        pp.k = (uint32_t)k;
        pp.step = (uint32_t)step;
        pp.G = rp.G;
        pp.M = rp.M;
        for (int t = 1; t < kTiers; ++t) pp.cap[t - 1] = tier_cap[t];
        if (arms_kernel) {
            pp.long3 = (uint32_t)opt.long3;
            pp.long3_big = pp.long3 / 4u;
            pp.dense3 = (uint32_t)opt.dense3;
            pp.dense6 = (uint32_t)opt.dense6;
            if (force_tier == 3) {
                pp.long3 = pp.long3_big = 1;
                pp.dense3 = pp.dense6 = 0;
            }
            pp.cap[0] = (uint32_t)std::min<int64_t>(opt.cap1, kArmCapSmall);
        }
        int force_eff = force_tier;  // a forced tier that has no kernel in this mode: the next one that has
        while (force_eff > 1 && force_eff < kTiers && !tier_enabled(force_eff)) ++force_eff;
        pp.sum1 = kTier1MaxSum;
        pp.force_tier = force_eff;
        {
            // one segment per lane for the first kLaneWalk probes, the longer ones wave by wave (their list
            // borrows the overflow lists' buffer, which the tiers only use afterwards)
            uint32_t *long_list = w.ovf_list.as<uint32_t>();
            seg_stats_lanes_kernel<<<(unsigned)std::min<uint64_t>((n_seg + 63) / 64, 256ull * 9ull * 2ull), 64, 0, s>>>(
                rp, p_filt, seg_list, d_ctr + CT_SEG, kbuf, vbuf, pp, long_list, d_ctr);
            seg_stats_kernel<<<(unsigned)std::min<uint64_t>(n_seg, 256ull * 16ull), 64, 0, s>>>(
                rp, p_filt, seg_list, d_ctr + CT_LONGSEG, long_list, kbuf, vbuf, pp, d_ctr);
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamWaitEvent(s, cx.ev[3], 0));  // the hit rows (front(): filled beside the walk above)
        if (cluster_barren) {
            HIP_TRY(hipEventRecord(cx.ev[13], s));
            // barren by position (cluster_barren_kernel): a wave per segment, two bitmap sizes (2 KB: 32 waves per compute
            // unit; 16 KB: 9)
            cluster_barren_kernel<SlotT, 16384><<<256 * 32, 64, 0, s>>>(rp, pp, row_off, hits, seg_list, d_ctr + CT_SEG, kbuf, 1u, 1024u,
                                                                       d_ctr + CT_CLUSTER_CUR, d_ctr);
            HIP_TRY(hipEventRecord(cx.ev[14], s));
            cluster_barren_kernel<SlotT, 131072><<<256 * 9, 64, 0, s>>>(rp, pp, row_off, hits, seg_list, d_ctr + CT_SEG, kbuf, 1025u,
                                                                       16384u, d_ctr + CT_CLUSTER_CUR + 1, d_ctr);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipEventRecord(cx.ev[15], s));
        }
        if (split_on) {
            SplitParams sp{};
            sp.range_len = (uint32_t)opt.split_len;
            sp.warm = (uint32_t)opt.split_warm;
            sp.min_span = (uint32_t)std::min<int64_t>(opt.split_min, 0x7FFFFFFF);
            sp.max_runs = kMaxRuns - kMaxSplits;  // (one more run per cut segment may follow)
            sp.max_cuts = kMaxCuts;
            sp.max_splits = kMaxSplits;
            {   // segments a cut of which failed in an earlier call of this index with these settings and chunks, if this
                // call's windows hold them: as probe numbers of THIS call (the newest verdicts first)
                BlockedSeg *const h_blocked = reinterpret_cast<BlockedSeg *>(h_split + kOffBlocked);
                std::lock_guard<std::mutex> lk(idx->mu);
                for (auto b = idx->split_blocked.rbegin(); b != idx->split_blocked.rend(); ++b)
                    for (int32_t p_ = 0; p_ < n_passes && sp.n_blocked < kSplitBlockedMax; ++p_)
                        if (b->sig == call_sig && ((rp.modes >> (8 * p_)) & 0xFFu) == (uint32_t)(b->key >> 32) &&
                            (uint32_t)b->key >= lo_lim && (uint32_t)b->key < hi_lim)
                            h_blocked[sp.n_blocked++] = BlockedSeg{(uint32_t)p_ * K + ((uint32_t)b->key - lo_lim), b->range_len, b->allowed, b->warm};
            }
            sp.blocked = reinterpret_cast<const BlockedSeg *>(d_split + kOffBlocked);
            if (sp.n_blocked)
                HIP_TRY(hipMemcpyAsync(d_split + kOffBlocked, h_split + kOffBlocked, (size_t)sp.n_blocked * sizeof(BlockedSeg), hipMemcpyHostToDevice, s));
            SplitChoice *const d_choice = reinterpret_cast<SplitChoice *>(d_split + kOffChoice);
            if (!sp.range_len) {  // the range length by budget
                HIP_TRY(hipMemsetAsync(d_choice, 0, sizeof(SplitChoice), s));
                split_tally_kernel<<<grid_for(n_seg), 256, 0, s>>>(rp, d_ctr + CT_SEG, kbuf, pp.seg_info, d_choice);
                split_pick_kernel<<<1, 1, 0, s>>>(d_choice, (uint32_t)opt.split_runs);
            }
            plan_ranges_kernel<<<grid_for(n_seg), 256, 0, s>>>(rp, sp, p_filt, seg_list, d_ctr + CT_SEG, kbuf, pp.seg_info,
                                                              reinterpret_cast<unsigned long long *>(d_split),
                                                              reinterpret_cast<RangeRun *>(d_split + kOffRuns),
                                                              reinterpret_cast<uint2 *>(d_split + kOffCuts),
                                                              reinterpret_cast<SplitSeg *>(d_split + kOffSplits), d_choice);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemcpyAsync(h_split, d_split, kOffMeta, hipMemcpyDeviceToHost, s));
        }
        if (opt.debug >= 2 && pp.seg_info) {  // the longest and the richest segment every tier runs WHOLE
            unsigned long long *d_top = nullptr, h_top[2 * kTiers] = {};
            HIP_TRY(hipMalloc(&d_top, sizeof(h_top)));
            (void)hipMemsetAsync(d_top, 0, sizeof(h_top), s);
            top_uncut_kernel<<<grid_for(n_seg), 256, 0, s>>>(d_ctr + CT_SEG, kbuf, pp.seg_info, d_top);
            (void)hipMemcpyAsync(h_top, d_top, sizeof(h_top), hipMemcpyDeviceToHost, s);
            (void)hipStreamSynchronize(s);
            for (int t = 0; t < 2 * kTiers; ++t) {
                if (!h_top[t]) continue;
                uint2 info{};
                uint32_t g0 = 0;
                const size_t sj = (size_t)(h_top[t] & 0xFFFFFFFFull);
                (void)hipMemcpy(&info, pp.seg_info + sj, sizeof(info), hipMemcpyDeviceToHost);
                (void)hipMemcpy(&g0, seg_list + sj, sizeof(g0), hipMemcpyDeviceToHost);
                fprintf(stderr, "[asgart] tier %d, run whole, the segment with the most %s: at probe %u, %u probe positions%s, %u hits\n", t / 2 + 1,
                        t % 2 ? "hits" : "probe positions", g0, info.y & 0x7FFFFFFFu, (info.y >> 31) ? "" : " (cut short by the window)", info.x);
            }
            (void)hipFree(d_top);
        }
        order = nullptr;
        const uint32_t *sorted_keys = nullptr;
        RC_TRY(sort_segments(w, kbuf, vbuf, n_seg, s, &order, &sorted_keys, false));
        tier_bounds_kernel<<<1, 64, 0, s>>>(sorted_keys, d_ctr + CT_SEG, d_ctr);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(h_ctr, d_ctr, kCtrBytes, hipMemcpyDeviceToHost, s));
        RC_TRY(wd_sync(idx, cx, s, "the hit rows and the placement of the segments"));
        seg_off[0] = 0;
        for (int t = 0; t < kTiers; ++t) {
            n_t[t] = h_ctr[CT_N1 + t];
            seg_off[t + 1] = seg_off[t] + n_t[t];
            h_tp[t] = h_ctr[CT_TPROBES1 + t];
            h_th[t] = h_ctr[CT_THITS1 + t];
        }
        h_split_hdr = reinterpret_cast<const unsigned long long *>(h_split);
        n_runs = split_on ? (uint32_t)h_split_hdr[0] : 0u;
        n_cuts = split_on ? (uint32_t)h_split_hdr[1] : 0u;
        n_splits = split_on ? (uint32_t)h_split_hdr[2] : 0u;
        split_segs.assign(n_splits, SplitSeg{});
        if (n_splits) memcpy(split_segs.data(), h_split + kOffSplits, (size_t)n_splits * sizeof(SplitSeg));
        n_split_segments = n_splits;
        if (opt.debug && split_on)
            for (const SplitSeg &sg : split_segs)
                fprintf(stderr, "[asgart] ranges: segment at probe %u: tier %u, %u probe positions, %u hits -> %u ranges\n", sg.g_seg0, sg.tier, sg.span,
                        sg.hits, sg.n_ranges);
        if (opt.debug && split_on)
            fprintf(stderr, "[asgart] %u long segment(s) cut into ranges: %u runs (ranges of %llu probes%s), %u cuts to check\n",
                    n_splits, n_runs, (unsigned long long)h_split_hdr[4], opt.split_len ? "" : ": the shortest that keeps the runs within the budget", n_cuts);
        if (opt.debug) {
            fprintf(stderr, "[asgart] %llu segments, %llu walked wave by wave; per tier:", (unsigned long long)n_seg,
                    (unsigned long long)h_ctr[CT_LONGSEG]);
            unsigned long long placed = 0;
            for (int t = 0; t < kTiers; ++t) {
                fprintf(stderr, " %llu", (unsigned long long)n_t[t]);
                placed += n_t[t];
            }
            fprintf(stderr, "; %llu barren (no arm can reach min_duplication_length: not run), %llu of them by the positions of their hits\n",
                    (unsigned long long)n_seg - placed, (unsigned long long)h_ctr[CT_CLUSTER_BARREN]);
            if (cluster_barren) {
                float ms_a = 0.f, ms_b = 0.f, ms_p = 0.f;
                (void)hipEventElapsedTime(&ms_a, cx.ev[13], cx.ev[14]);
                (void)hipEventElapsedTime(&ms_b, cx.ev[14], cx.ev[15]);
                (void)hipEventElapsedTime(&ms_p, cx.ev[16], cx.ev[13]);
                fprintf(stderr, "[asgart] hit rows filled, placement walk beside it: %.2f ms; barren by position: segments of up to 1 024 hits %.2f ms, up to 16 384 hits %.2f ms\n",
                        ms_p, ms_a, ms_b);
            }
        }
        for (int t = 0; t < kTiers; ++t) ovf[t] = w.ovf_list.as<uint32_t>() + (size_t)t * (n_seg + 1);
        // HBM arm storage of the LDS-array kernels: tier 6 (MODE 1) and tier 7 (MODE 2) may run at the
        // same time on different streams, so each gets its own region of 256 per-workgroup slices
        // Tier 7 never refuses a segment for its size: max_cardinality * (t* + 1) bounds the live arms of ANY segment
        // (every live arm was created or extended within the last t* + 1 processed probes, at most max_cardinality
        // per probe; src/automaton.rs:87 keeps them in an unbounded Vec), and its slices are sized for that.  Large
        // bounds get fewer workgroups (a 16 GB budget for the four regions), never fewer than one.
        // (rounded up to a multiple of 4: a workgroup's slice holds 64-bit position arrays behind arrays of this many
        // 32-bit words, and slices, regions and the early-cascade regions follow one another at multiples of it)
        heavy_cap64 = (std::max<uint64_t>((uint64_t)rp.C * ((uint64_t)rp.tstar + 1u) + 64u, 4096u) + 3u) & ~3ull;
        if (heavy_cap64 >= (1ull << 24)) {
            set_error("max_cardinality * (max_gap_size / step + 1) = %llu live arms per segment: more than 2^24 are not supported",
                      (unsigned long long)heavy_cap64);
            return ASGART_E_CAP;
        }
        const size_t per_wg7 = (size_t)heavy_cap64 * (4 * sizeof(SlotT) + 28);
        const size_t per_wg6 = (size_t)caph * (4 * sizeof(SlotT) + 16);
        const size_t per_wg = std::max(per_wg6, per_wg7);
        n_wg7 = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(256, (16ull << 30) / (4 * per_wg)));
        const unsigned n_wg_region = per_wg6 * 256 > per_wg * n_wg7 ? 256u : n_wg7;  // (a region serves either kind of launch)
        region = std::max(per_wg6 * 256, per_wg * (size_t)n_wg7);
        (void)n_wg_region;
        RC_TRY(w.scratch.reserve(region * 4));  // tier 6, tier 7, and one region per early cascade launch
        scratch6 = w.scratch.as<char>();
        scratch7 = scratch6 + region;
        return 0;
    }

    // one launch of tier `tier`'s kernel over the list described by ep
    void launch_kernel(int tier, uint64_t n_items, hipStream_t st) {
        // option grid<t> may shrink a tier's grid; the workgroup kernels (tiers 3..7) never get
        // more workgroups than their default (HBM scratch is reserved for that many)
        ep.tier = (uint32_t)tier;
        ep.seg_slots = w.seg_slots.as<unsigned long long>() + (size_t)4096 * (size_t)(tier & 7);
        ep.hb = cx.d_hb ? cx.d_hb + (size_t)2 * SearchCtx::kHbSlots * (size_t)(tier & 7) : nullptr;
        auto grid = [&](uint64_t dflt) -> unsigned {
            uint64_t g = dflt;
            if (opt.grid[tier] > 0) g = tier >= 3 ? std::min<uint64_t>(dflt, (uint64_t)opt.grid[tier]) : (uint64_t)opt.grid[tier];
            return (unsigned)std::min<uint64_t>(n_items, g);
        };
        switch (tier) {
        case 1:
            extend_kernel<SlotT, kArmCapSmall><<<grid(kGrid1), 64, 0, st>>>(ep);
            break;
        case 2:
            if (arms_kernel)
                extend_fast_kernel<SlotT, kWaveArmsLayers<SlotT>, 64, kWaveArmsHits, 256, 2><<<grid(kGrid2Arms), 64, 0, st>>>(ep);
            else
                extend_heavy_kernel<SlotT, kArmCapMid, kMidThreads, 0><<<grid(kGrid2), kMidThreads, 0, st>>>(ep);
            break;
        case 3:  // (arm-resident kernels only: the long dense segments)
            extend_k8_kernel<SlotT, kK8LongLayers<SlotT>, 1024, kHitBatch, kFastLongRows<SlotT>, 2><<<grid(256), 1024, 0, st>>>(ep);
            break;
        case 4:
            if (arms_kernel)
                extend_fast_kernel<SlotT, kMidArmsLayers<SlotT>, 256, kWaveArmsHits, 512, 2><<<grid(256 * 4), 256, 0, st>>>(ep);
            else if constexpr (sizeof(SlotT) == 4)
                extend_heavy_kernel<SlotT, kArmCapBig32, kHeavyThreads, 0><<<grid(256), kHeavyThreads, 0, st>>>(ep);
            else
                extend_heavy_kernel<SlotT, kArmCapBig64, kHeavyThreads, 0><<<grid(256), kHeavyThreads, 0, st>>>(ep);
            break;
        case 5:  // (arm-resident kernels only)
            extend_fast_kernel<SlotT, kMidArmsLayers<SlotT>, 512, kHitBatch, 1024, 2><<<grid(256 * 2), 512, 0, st>>>(ep);
            break;
        case 6:
            if (scratch_override) ep.scratch = scratch_override;
            if (arms_kernel)  // 5 x 1024 slots; 64-bit positions: 4 x 1024 with a smaller table
                extend_fast_kernel<SlotT, kFastHeavyLayers<SlotT>, 1024, kHitBatch, kFastLongRows<SlotT>, 2><<<grid(256), 1024, 0, st>>>(ep);
            else
                extend_heavy_kernel<SlotT, caph, kHeavyThreads, 1><<<grid(256), kHeavyThreads, 0, st>>>(ep);
            ep.scratch = scratch6;
            break;
        default:
            ep.scratch = scratch_override ? scratch_override : scratch7;
            extend_heavy_kernel<SlotT, 1, kHeavyThreads, 2><<<grid(n_wg7), kHeavyThreads, 0, st>>>(ep);
            ep.scratch = scratch6;
            break;
        }
    }

    // the runs over ranges of the cut segments: first, on the main stream (idle while the tiers run) -- they are the
    // longest work items of the call, one workgroup each
    void launch_runs(uint32_t n_items) {  // the runs from the work cursor on, one workgroup each
        ep.runs = reinterpret_cast<const RangeRun *>(d_split + kOffRuns);
        ep.run_meta = reinterpret_cast<uint32_t *>(d_split + kOffMeta);
        ep.run_dump = w.split_dump.as<uint32_t>();
        ep.seg_list = nullptr;
        ep.n_seg_ptr = reinterpret_cast<const unsigned long long *>(d_split);
        ep.cursor = reinterpret_cast<unsigned long long *>(d_split + 24);
        ep.ovf_list = nullptr;
        ep.ovf_count = d_ctr + CT_OVF1 + 2;
        ep.tier = 3;  // (statistics: with the long-segment tier)
        ep.seg_slots = w.seg_slots.as<unsigned long long>();  // (slot block 0: no tier's)
        ep.hb = cx.d_hb ? cx.d_hb : nullptr;
        extend_k8_kernel<SlotT, kK8LongLayers<SlotT>, 1024, kHitBatch, kFastLongRows<SlotT>, 2, true><<<n_items, 1024, 0, s>>>(ep);
    }

    void launch_tier(int tier) {
        if (tier < 1 || tier > kTiers || !n_t[tier - 1]) return;
        ep.seg_list = order + seg_off[tier - 1];
        ep.n_seg_ptr = d_ctr + CT_N1 + (tier - 1);
        ep.cursor = d_ctr + CT_CUR1 + (tier - 1);
        ep.ovf_list = tier < kTiers ? ovf[tier - 1] : nullptr;
        ep.ovf_count = d_ctr + CT_OVF1 + (tier - 1);
        launch_kernel(tier, n_t[tier - 1], tier_stream[tier]);
    #ifdef ASGART_PROFILE_EXTEND
        char tag[8];
        snprintf(tag, sizeof tag, "%d", tier);
        PROF_TIER(tag, tier_stream[tier], n_t[tier - 1]);
    #endif
    }

    // ---- every tier and the runs over ranges, launched together; early re-runs of what tiers 3 and 6 give up on ----------------
    int32_t run_tiers(int attempt) {
        RC_TRY(w.fam_sds.reserve((size_t)rec_cap * sizeof(SdRec)));
        HIP_TRY(hipMemsetAsync(d_ctr + CT_SD, 0, 8, s));
        HIP_TRY(hipMemsetAsync(d_ctr + CT_NF, 0, (size_t)(CT_COUNT - CT_NF) * 8, s));  // NF, cursors, overflow counts
        HIP_TRY(hipMemsetAsync(d_ctr + CT_EARLY_N, 0, 4 * 8, s));                       // early cascade counts + cursors
        HIP_TRY(hipMemsetAsync(d_ctr + CT_BUSY1, 0, (size_t)(CT_COUNT - CT_BUSY1) * 8, s));
        RC_TRY(w.seg_slots.reserve((size_t)8 * 4096 * 8));
        HIP_TRY(hipMemsetAsync(w.seg_slots.p, 0, (size_t)8 * 4096 * 8, s));
        HIP_TRY(hipEventRecord(cx.ev[7], s));
        ep.rp = rp;
        ep.p_filt = p_filt;
        ep.row_off = row_off;
        ep.hits = hits;
        ep.recs = w.fam_sds.as<SdRec>();
        ep.rec_cap = rec_cap;
        ep.scratch = scratch6;
        // option test_cap_limit (tests): shrink the tiers' capacity to exercise the cascade
        ep.cap_limit = opt.test_cap_limit >= 0 ? (uint32_t)opt.test_cap_limit : 0xFFFFFFFFu;
        ep.escalate_cost = 0xFFFFFFFFu;
        ep.heavy_cap = (uint32_t)heavy_cap64;
        ep.solo_hits = opt.solo == 1 ? 16u : (uint32_t)opt.solo;  // (1: the default of 16 hits; other values: that many)
        ep.gen_bits = (uint32_t)opt.test_genbits;
        ep.k8_delay = (uint32_t)opt.test_k8_delay;
        ep.ctr = d_ctr;
        ep.hb = nullptr;
        RC_TRY(cx.heartbeat(opt.watchdog_s > 0));
        if (cx.h_hb) memset(cx.h_hb, 0, (size_t)SearchCtx::kHbTiers * SearchCtx::kHbSlots * 16);
        // The tiers are launched together on separate streams, each with a grid that can fill
        // the chip on its own (persistent workgroups, longest segment first): the hardware
        // back-fills CUs as workgroups retire, so the tails of one tier overlap the next.
        // The window bound guarantees that a segment fits its
        // tier, so the overflow lists normally stay empty (they feed the cascade below).
        const hipStream_t st2 = cx.stream2, st3 = cx.stream3, st4 = cx.stream4, st5 = cx.stream5, st6 = cx.stream6, st7 = cx.stream7;
        {
            const hipStream_t streams[kTiers + 1] = {s, st7, st2, st3, st4, st5, st6, st2};
            for (int t = 0; t <= kTiers; ++t) tier_stream[t] = streams[t];
        }
        HIP_TRY(hipStreamWaitEvent(st2, cx.ev[7], 0));
        HIP_TRY(hipStreamWaitEvent(st3, cx.ev[7], 0));
        HIP_TRY(hipStreamWaitEvent(st4, cx.ev[7], 0));
        HIP_TRY(hipStreamWaitEvent(st5, cx.ev[7], 0));
        HIP_TRY(hipStreamWaitEvent(st6, cx.ev[7], 0));
        HIP_TRY(hipStreamWaitEvent(st7, cx.ev[7], 0));
        // Launch order and grid sizes: workgroups are persistent and hold their LDS until the
        // tier's work list is exhausted, so whatever is dispatched first owns the CUs.  The
        // critical path of a pass is the longest tandem-array segment of the heavy tiers
        // (tens of thousands of probes, strictly serial): those tiers go first, the one-wave
        // tier last, and the heavy grids are sized so that every tier's longest segments
        // start at once instead of queueing behind another tier's bulk.
        const std::string tier_order = std::to_string((long long)opt.tier_order);
        scratch_override = nullptr;
        if (n_runs) {
            // (room for one more run per cut segment: the rest behind the last cut that held, see below)
            RC_TRY(w.split_dump.reserve((size_t)(n_runs + n_splits) * 2 * kRunDumpCap * kDumpWords<SlotT> * 4));
            HIP_TRY(hipMemsetAsync(d_split + 24, 0, 8, s));                         // work cursor
            HIP_TRY(hipMemsetAsync(d_split + kOffMeta, 0, (size_t)n_runs * 64, s));  // run states
            launch_runs(n_runs);
            HIP_TRY(hipGetLastError());
        }
        t_launch = std::chrono::steady_clock::now();
        for (char c : tier_order) launch_tier(c - '0');
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(cx.ev[12], st7));
        HIP_TRY(hipEventRecord(cx.ev[5], st2));
        HIP_TRY(hipEventRecord(cx.ev[6], st3));
        HIP_TRY(hipEventRecord(cx.ev[8], st4));
        HIP_TRY(hipEventRecord(cx.ev[9], st5));
        HIP_TRY(hipEventRecord(cx.ev[10], st6));
        if (cx.progress && attempt == 0) {
            // Progress (reference src/automaton.rs:98 stores every probe's offset for a polled progress
            // bar): every probe of the call has been searched and its hits are materialised -- the
            // HBM-bound, chip-wide part of the call is over, the extension automaton is under way.
            // A host that pipelines calls (bench.py) issues the next one when it sees this: its search
            // phases then run beside this call's extension, whose tail is a few serial segments.
            if (!progress_given) {
                RC_TRY(wd_event_sync(idx, cx, cx.ev[3], "the hit rows"));  // probe search, scans and CSR fill are done
                signal_progress();
            }
        }
        // Early cascades of tiers 3 and 6, the two tiers that accept segments above their real capacity (by
        // the allowance): what they give up on need not wait for the other tiers to be re-run -- the longest
        // tier-3 segment of a GRCh38-shaped pass runs 60+ ms longer than tier 6, and in a two-genome run tier 6
        // runs seconds longer than tier 3.  The host waits for whichever of the two finishes first, reads its
        // overflow count and launches the re-run behind it on the same stream (the main stream is idle
        // meanwhile and has the highest priority); each such launch has its own counters and HBM slices.
        for (int t = 0; t <= kTiers; ++t) early_n[t] = 0;
        {
            struct Early {
                int src, dst;
                hipEvent_t ev;
                hipStream_t st;
                bool pending;
            } early[2] = {{3, 0, cx.ev[6], st3, false}, {6, 0, cx.ev[10], st6, false}};
            int n_pending = 0;
            unsigned early_polls = 0;
            Watchdog early_wd(idx, cx);
            for (int e = 0; e < 2; ++e) {
                Early &E = early[e];
                int dst = E.src + 1;
                while (dst < kTiers && (!tier_enabled(dst) || tier_cap[dst] <= tier_cap[E.src])) ++dst;
                E.dst = dst;
                // (a re-run in the HBM tier would share that tier's slices with its own list, if it has one)
                E.pending = n_t[E.src - 1] && tier_stream[E.src] == E.st &&
                            (dst < kTiers || !n_t[kTiers - 1]);
                n_pending += E.pending ? 1 : 0;
            }
            // (If tier 3's re-run went to tier 6, its kernel would append to tier 6's overflow list while the host
            // snapshots that list's length for tier 6's own early re-run -- a count that may be ahead of the entry:
            // such a re-run waits for the regular cascade below.  With the shipped shapes tier 6 never accepts
            // more than tier 3 and the case does not arise.)
            if (early[0].pending && early[1].pending && early[0].dst == early[1].src) {
                early[0].pending = false;
                --n_pending;
            }
            while (n_pending) {
                bool progressed = false;
                for (int e = 0; e < 2; ++e) {
                    Early &E = early[e];
                    if (!E.pending) continue;
                    const hipError_t q = hipEventQuery(E.ev);
                    if (q == hipErrorNotReady) {
                        (void)hipGetLastError();
                        continue;
                    }
                    HIP_TRY(q);
                    E.pending = false;
                    --n_pending;
                    progressed = true;
                    HIP_TRY(hipMemcpyAsync(h_scalar + 1 + e, d_ctr + CT_OVF1 + E.src - 1, 8, hipMemcpyDeviceToHost, s));
                    RC_TRY(wd_sync(idx, cx, s, "an overflow count"));
                    const uint64_t n_e = h_scalar[1 + e];
                    if (opt.debug)
                        fprintf(stderr, "[asgart] tier %d done %.1f ms after the launches, %llu segment(s) to re-run in tier %d\n",
                                E.src, since_launch(), (unsigned long long)n_e, E.dst);
                    if (!n_e) continue;
                    early_n[E.src] = n_e;
                    // the count the launch works on is fixed now (the list itself may still grow)
                    HIP_TRY(hipMemcpyAsync(d_ctr + CT_EARLY_N + e, h_scalar + 1 + e, 8, hipMemcpyHostToDevice, E.st));
                    ep.seg_list = ovf[E.src - 1];
                    ep.n_seg_ptr = d_ctr + CT_EARLY_N + e;
                    ep.cursor = d_ctr + CT_EARLY_CUR + e;
                    ep.ovf_list = E.dst < kTiers ? ovf[E.dst - 1] : nullptr;
                    ep.ovf_count = d_ctr + CT_OVF1 + E.dst - 1;
                    ep.escalate_cost = 0xFFFFFFFFu;
                    ep.cap_limit = 0xFFFFFFFFu;
                    scratch_override = scratch6 + region * (size_t)(2 + e);
                    launch_kernel(E.dst, n_e, E.st);
                    scratch_override = nullptr;
                    HIP_TRY(hipGetLastError());
                    HIP_TRY(hipEventRecord(E.ev, E.st));
                }
                if (n_pending && !progressed) {
                    std::this_thread::sleep_for(std::chrono::microseconds(50));
                    // (the same watchdog as every other wait of the call -- one look at the heartbeats, never a blocking
                    // wait: the other tier's early re-run must not wait for this one's stream to drain)
                    if ((++early_polls & 0x3FFu) == 0 && early_wd.expired(early[0].pending ? "extension tier 3" : "extension tier 6"))
                        return ASGART_E_HIP;
                }
            }
        }
        return 0;
    }

    // ---- the tiers are through: cuts checked, ranges joined up, refused segments run again --------------------------------------
    int32_t join_ranges() {
        HIP_TRY(hipStreamWaitEvent(s, cx.ev[5], 0));
        HIP_TRY(hipStreamWaitEvent(s, cx.ev[6], 0));
        HIP_TRY(hipStreamWaitEvent(s, cx.ev[8], 0));
        HIP_TRY(hipStreamWaitEvent(s, cx.ev[9], 0));
        HIP_TRY(hipStreamWaitEvent(s, cx.ev[10], 0));
        HIP_TRY(hipStreamWaitEvent(s, cx.ev[12], 0));
        if (n_cuts) {
            validate_cuts_kernel<kDumpWords<SlotT>><<<n_cuts, 256, 0, s>>>(reinterpret_cast<const uint2 *>(d_split + kOffCuts),
                                                       reinterpret_cast<const uint32_t *>(d_split + kOffMeta), w.split_dump.as<uint32_t>(),
                                                       reinterpret_cast<uint32_t *>(d_split + kOffOk));
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemcpyAsync(h_split + kOffMeta, d_split + kOffMeta, kOffFix - kOffMeta, hipMemcpyDeviceToHost, s));
        }
        HIP_TRY(hipMemcpyAsync(h_ctr, d_ctr, kCtrBytes, hipMemcpyDeviceToHost, s));
        RC_TRY(wd_sync(idx, cx, s, "the extension tiers"));
        if (n_runs) {
            // Every cut of a segment held: the family ordinals of range j count on from the flushes of ranges 0 .. j - 1.
            // The first cut that did not hold is cut f: ranges 0 .. f stand (range f started from a state that was checked),
            // the records of the ranges behind it are dropped, and ONE more run covers the rest -- from where range f started
            // (it passes cut f in the checked state, so it is exact from there on), reporting from cut f + 1 to the segment's
            // end.  f = 0, or a run that gave up: the segment runs again as a whole on the ordinary kernel.  The index remembers
            // what the segment needs from then on (remember(), below): a longer warm-up, or only the cuts that held.
            const uint32_t *h_meta = reinterpret_cast<const uint32_t *>(h_split + kOffMeta);
            const uint32_t *h_ok = reinterpret_cast<const uint32_t *>(h_split + kOffOk);
            RangeRun *h_runs = reinterpret_cast<RangeRun *>(h_split + kOffRuns);
            uint32_t *h_fix = reinterpret_cast<uint32_t *>(h_split + kOffFix);
            uint32_t *h_again = reinterpret_cast<uint32_t *>(h_split + kOffAgain);
            uint32_t n_again = 0, n_tail = 0;
            n_split_refused = 0;
            struct Tail {
                uint32_t run, base, g_seg0, run_base, f;
            };
            std::vector<uint32_t> held_base;  // family-ordinal bases of the ranges whose records wait for their segment's last run
            std::vector<Tail> tails;
            // What the index remembers about a segment a cut of which did not hold.  allowed: how many of its cuts, from the start,
            // held; warm_used: the warm-up its ranges had (0: the warm-up is not the matter); need: how far in front of the failed
            // cut the oldest arm of the range before it was born (probes; 0: it held no arm) -- a run that is to hold that arm at
            // the cut has to start in front of its birth.  While a warm-up of that size is still worth a range (no longer than
            // two ranges, nor than option split_warm_max) the next call gives this segment's ranges that much and plans all its
            // cuts again: a shard of a genome-sized call gets short ranges with short warm-ups, and with "only the cuts that
            // held" the rest of such a segment became one long last range, the longest work item of the shard (81 ms of a 90-ms
            // shard).  A cut that failed although every arm was born inside the warm-up (thresholds that had not settled) gets
            // the longest warm-up that limit allows.  Beyond the limit -- the arms of a flat tandem array live from its first
            // probe to its last, those of a chromosome run against its homologue for megabases: every run would walk the
            // stretch again from its start -- only the cuts that held are planned again.
            auto remember = [&](uint32_t g_seg0, uint32_t allowed, uint32_t warm_used, uint32_t need) {
                const uint32_t p_ = pass_of_probe(g_seg0);
                const uint64_t key_ = (uint64_t)((rp.modes >> (8 * p_)) & 0xFFu) << 32 | (uint64_t)pass_offset(g_seg0);
                const uint32_t len_ = (uint32_t)h_split_hdr[4];
                const uint64_t limit = std::min<uint64_t>((uint64_t)opt.split_warm_max, 2ull * len_);
                // (born inside the warm-up and still not the same arms: the longest warm-up a range is worth, at once -- a
                // doubling per call took a two-genome call, 2-3 s each, five calls to settle)
                const uint64_t want = need > warm_used ? (((uint64_t)need + 64u + 255u) & ~255ull) : (warm_used < limit ? limit : ~0ull);
                const uint32_t warm_next = (warm_used && want <= limit) ? (uint32_t)want : 0u;
                if (warm_next) allowed = kAllCuts;
                std::lock_guard<std::mutex> lk(idx->mu);
                // (one verdict per segment AND range length: calls of different shapes -- the passes as one job, a pass alone --
                // get different range lengths from the budget, and each shape keeps what it has learned)
                for (auto &b : idx->split_blocked)
                    if (b.key == key_ && b.sig == call_sig && b.range_len == len_) {
                        b.allowed = warm_next ? allowed : std::min(b.allowed, allowed);
                        if (warm_next) b.warm = warm_next;
                        return;
                    }
                // (verdicts age out, oldest first: a forgotten one costs its segment one more refused cut, no more)
                if (idx->split_blocked.size() >= 4096) idx->split_blocked.erase(idx->split_blocked.begin());
                idx->split_blocked.push_back({key_, call_sig, allowed, len_, warm_next});
            };
            for (const SplitSeg &sg : split_segs) {
                const uint32_t n_cuts_sg = sg.n_ranges - 1;
                uint32_t f = n_cuts_sg;  // first cut that did not hold
                for (uint32_t j = 0; j < n_cuts_sg; ++j)
                    if (!(h_ok[sg.cut_base + j] & 1u)) {
                        f = j;
                        break;
                    }
                const bool last_gave_up = h_meta[(size_t)(sg.run_base + sg.n_ranges - 1) * 16 + 4] != 0u;
                const bool ok = f == n_cuts_sg && !last_gave_up;
                if (f == n_cuts_sg && last_gave_up) f = 0;  // (more arms than the long shape holds: the cascade's business)
                uint32_t base = 0;
                for (uint32_t j = 0; j < sg.n_ranges; ++j) {
                    // (ranges in front of a cut that did not hold: decided when the run over the rest has come back)
                    h_fix[sg.run_base + j] = ok ? base : ((f > 0 && j <= f) ? 0xFFFFFFFEu : 0xFFFFFFFFu);
                    if (!ok && f > 0 && j <= f) held_base.push_back(base);
                    base += h_meta[(size_t)(sg.run_base + j) * 16 + 1];
                    if (!ok && f > 0 && j == f) {
                        RangeRun t = h_runs[sg.run_base + f];
                        t.g_stop = 0xFFFFFFFFu;
                        t.emit_from = h_runs[sg.run_base + f + 1].emit_from;
                        t.flags = kRunLast;
                        h_runs[n_runs + n_tail] = t;
                        tails.push_back(Tail{n_runs + n_tail, base, sg.g_seg0, sg.run_base, f});
                        ++n_tail;
                    }
                }
                if (!ok && opt.debug) {
                    fprintf(stderr, "[asgart] ranges: segment at probe %u (%u ranges): %u cut(s) held; cuts (arms in the range in front / in the range behind, family open, held flush):",
                            sg.g_seg0, sg.n_ranges, f);
                    for (uint32_t j = 0; j < n_cuts_sg; ++j) {
                        const uint32_t *ma = h_meta + (size_t)(sg.run_base + j) * 16, *mb = h_meta + (size_t)(sg.run_base + j + 1) * 16 + 8;
                        fprintf(stderr, " %s%u/%u,%u/%u,%u/%u%s", (h_ok[sg.cut_base + j] & 1u) ? "" : "[", ma[0], mb[0], ma[2], mb[2], ma[3], mb[3],
                                (h_ok[sg.cut_base + j] & 1u) ? "" : "]");
                    }
                    fprintf(stderr, "%s\n", last_gave_up ? " (the last range gave up)" : "");
                }
                if (!ok) {
                    ++n_split_refused;
                    uint32_t need = 0;  // probes between the birth of the oldest arm in front of the failed cut and that cut
                    if (f < n_cuts_sg) {
                        const uint32_t born = (h_ok[sg.cut_base + f] >> 1) / (uint32_t)rp.step, at = h_runs[sg.run_base + f + 1].emit_from - sg.g_seg0;
                        if ((h_ok[sg.cut_base + f] >> 1) != (0xFFFFFFFFu >> 10) && born <= at) need = at - born;
                    }
                    if (opt.debug)
                        fprintf(stderr, "[asgart] ranges: segment at probe %u: cut %u did not hold with %u probes of warm-up; its oldest arm was born %u probes in front of it\n",
                                sg.g_seg0, f, sg.warm, need);
                    remember(sg.g_seg0, f, last_gave_up ? 0u : sg.warm, need);
                    if (f == 0) h_again[n_again++] = sg.g_seg0;
                }
            }
            if (opt.debug) {  // the runs' durations, per cut segment (ms)
                for (const SplitSeg &sg : split_segs) {
                    fprintf(stderr, "[asgart] ranges: segment at probe %u (tier %u, %u positions, %u hits): runs", sg.g_seg0, sg.tier, sg.span, sg.hits);
                    for (uint32_t j = 0; j < sg.n_ranges; ++j) fprintf(stderr, " %.1f", (double)h_meta[(size_t)(sg.run_base + j) * 16 + 5] * 1e-5);
                    fprintf(stderr, " ms\n");
                }
            }
            if (opt.debug)
                fprintf(stderr, "[asgart] ranges: %llu of %u cut segment(s) joined up%s\n", (unsigned long long)(n_splits - n_split_refused), n_splits,
                        n_split_refused ? "; the others: the rest behind the last cut that held as one more run, or the whole segment again" : "");
            for (const Tail &t : tails) h_fix[t.run] = 0xFFFFFFFEu;  // (until the run has come back)
            HIP_TRY(hipMemcpyAsync(d_split + kOffFix, h_fix, kOffAgain - kOffFix, hipMemcpyHostToDevice, s));
            uint64_t n_slots = std::min<uint64_t>(h_ctr[CT_SD], rec_cap);
            if (n_slots) fixup_records_kernel<<<grid_for(n_slots), 256, 0, s>>>(w.fam_sds.as<SdRec>(), n_slots, reinterpret_cast<const uint32_t *>(d_split + kOffFix));
            HIP_TRY(hipGetLastError());
            if (n_tail) {
                HIP_TRY(hipMemcpyAsync(d_split + kOffRuns + (size_t)n_runs * sizeof(RangeRun), h_runs + n_runs, (size_t)n_tail * sizeof(RangeRun),
                                       hipMemcpyHostToDevice, s));
                h_scalar[8] = (unsigned long long)n_runs + n_tail;  // list length
                h_scalar[9] = n_runs;                               // work cursor: behind the runs that are done
                HIP_TRY(hipMemcpyAsync(d_split, h_scalar + 8, 8, hipMemcpyHostToDevice, s));
                HIP_TRY(hipMemcpyAsync(d_split + 24, h_scalar + 9, 8, hipMemcpyHostToDevice, s));
                HIP_TRY(hipMemsetAsync(d_split + kOffMeta + (size_t)n_runs * 64, 0, (size_t)n_tail * 64, s));
                launch_runs(n_tail);
                HIP_TRY(hipMemcpyAsync(h_split + kOffMeta + (size_t)n_runs * 64, d_split + kOffMeta + (size_t)n_runs * 64, (size_t)n_tail * 64,
                                       hipMemcpyDeviceToHost, s));
                HIP_TRY(hipMemcpyAsync(h_ctr, d_ctr, kCtrBytes, hipMemcpyDeviceToHost, s));
                RC_TRY(wd_sync(idx, cx, s, "the rest of the cut segments"));
                h_scalar[8] = n_runs;  // (a second attempt of the call -- record buffer too small -- starts from the planned list)
                HIP_TRY(hipMemcpyAsync(d_split, h_scalar + 8, 8, hipMemcpyHostToDevice, s));
                size_t hb = 0;
                for (const Tail &t : tails) {
                    // (a last run that gave up -- more arms than the long shape holds --: everything the segment's runs wrote
                    // is dropped and the whole segment goes the cascade's way)
                    const bool gave_up = h_meta[(size_t)t.run * 16 + 4] != 0u;
                    h_fix[t.run] = gave_up ? 0xFFFFFFFFu : t.base;
                    for (uint32_t j = 0; j <= t.f; ++j, ++hb) h_fix[t.run_base + j] = gave_up ? 0xFFFFFFFFu : held_base[hb];
                    if (gave_up) {
                        h_again[n_again++] = t.g_seg0;
                        remember(t.g_seg0, 0u, 0u, 0u);  // (more arms than the long shape holds: not a matter of the warm-up)
                    }
                }
                HIP_TRY(hipMemcpyAsync(d_split + kOffFix, h_fix, kOffAgain - kOffFix, hipMemcpyHostToDevice, s));
                n_slots = std::min<uint64_t>(h_ctr[CT_SD], rec_cap);
                if (n_slots) fixup_records_kernel<<<grid_for(n_slots), 256, 0, s>>>(w.fam_sds.as<SdRec>(), n_slots, reinterpret_cast<const uint32_t *>(d_split + kOffFix));
                HIP_TRY(hipGetLastError());
            }
            if (n_again) {
                HIP_TRY(hipMemcpyAsync(d_split + kOffAgain, h_again, (size_t)n_again * 4, hipMemcpyHostToDevice, s));
                *h_scalar = n_again;
                HIP_TRY(hipMemcpyAsync(d_ctr + CT_NF, h_scalar, 8, hipMemcpyHostToDevice, s));
                HIP_TRY(hipMemsetAsync(d_ctr + CT_CURF, 0, 8, s));
                ep.seg_list = reinterpret_cast<const uint32_t *>(d_split + kOffAgain);
                ep.n_seg_ptr = d_ctr + CT_NF;
                ep.cursor = d_ctr + CT_CURF;
                ep.ovf_list = ovf[3 - 1];
                ep.ovf_count = d_ctr + CT_OVF1 + 3 - 1;  // (what the whole segment overflows goes the way of tier 3's own)
                ep.escalate_cost = 0xFFFFFFFFu;
                launch_kernel(3, n_again, s);
                HIP_TRY(hipGetLastError());
            }
            HIP_TRY(hipMemcpyAsync(h_ctr, d_ctr, kCtrBytes, hipMemcpyDeviceToHost, s));
            RC_TRY(wd_sync(idx, cx, s, "the ranges of the cut segments"));
        }
        return 0;
    }

    // ---- statistics of the tiers; the cascade: what tier t gave up on is re-run by the next tier that holds more ------------------
    int32_t finish_tiers() {
        PROF_DUMP("concurrent tiers");
        if (opt.debug) {
            fprintf(stderr, "[asgart] all tiers and early re-runs done %.1f ms after the launches\n", since_launch());
            // how much of the chip each tier held: sum of its workgroups' lifetimes x the share of a compute unit one of
            // them occupies (workgroups per compute unit by LDS / registers: tiers 1..7 = 11, 8, 1, 4, 2, 1, 1)
            static const double per_cu[kTiers] = {11, 8, 1, 4, 2, 1, 1};
            double tot = 0.0;
            fprintf(stderr, "[asgart] compute-unit time held per tier (CU-ms; workgroups):");
            for (int t = 0; t < kTiers; ++t) {
                const double cu_ms = (double)h_ctr[CT_BUSY1 + t] * 1e-5 / per_cu[t];
                tot += cu_ms;
                fprintf(stderr, " %d: %.0f (%llu)", t + 1, cu_ms, (unsigned long long)h_ctr[CT_WGS1 + t]);
            }
            fprintf(stderr, "  total %.0f = %.1f ms of the whole chip\n", tot, tot / 256.0);
            fprintf(stderr, "[asgart] per tier: hit-probes / hits per hit-probe / CU-microseconds per hit-probe:");
            for (int t = 0; t < kTiers; ++t) {
                const double hp = (double)h_tp[t];
                fprintf(stderr, " %d: %.0fK / %.1f / %.2f", t + 1, hp / 1e3, hp > 0 ? (double)h_th[t] / hp : 0.0,
                        hp > 0 ? (double)h_ctr[CT_BUSY1 + t] * 1e-2 / per_cu[t] / hp : 0.0);
            }
            fprintf(stderr, "\n");
        }
        {   // the tier that ran longest (its early re-run included): the serial floor of this call's extension
            float longest = 0.f;
            for (int e : {5, 6, 8, 9, 10, 12}) {
                float t = 0.f;
                if (hipEventElapsedTime(&t, cx.ev[7], cx.ev[e]) == hipSuccess) longest = std::max(longest, t);
                else (void)hipGetLastError();
            }
            ms_longest_tier = longest;
        }
        n_overflow = 0;
        for (int t = 1; t < kTiers; ++t) n_overflow += h_ctr[CT_OVF1 + t - 1];
        if (opt.debug) {
            fprintf(stderr, "[asgart] overflow out of tiers 1..%d:", kTiers - 1);
            for (int t = 1; t < kTiers; ++t) fprintf(stderr, " %llu", (unsigned long long)h_ctr[CT_OVF1 + t - 1]);
            fprintf(stderr, "\n");
        }
        n_heavy = 0;
        for (int t = 3; t <= kTiers; ++t) n_heavy += n_t[t - 1];
        // ---- cascade: what tier t gave up on is re-run from its start by the next tier that
        // is in use and holds more arms (its own overflow is appended to that tier's list) ------
        const auto t_casc0 = std::chrono::steady_clock::now();
        for (int src = 1; src < kTiers; ++src) {
            // (the first early_n entries of the list have been re-run already; a lower tier's cascade into
            // this tier may have appended more)
            const uint64_t skip = early_n[src];
            const uint64_t n_ovf = h_ctr[CT_OVF1 + src - 1] - skip;
            if (!n_ovf) continue;
            int dst = src + 1;
            while (dst < kTiers && (!tier_enabled(dst) || tier_cap[dst] <= tier_cap[src])) ++dst;
            *h_scalar = n_ovf;  // (the previous cascade launch has been waited for)
            HIP_TRY(hipMemcpyAsync(d_ctr + CT_NF, h_scalar, 8, hipMemcpyHostToDevice, s));
            HIP_TRY(hipMemsetAsync(d_ctr + CT_CURF, 0, 8, s));
            ep.seg_list = ovf[src - 1] + skip;
            ep.n_seg_ptr = d_ctr + CT_NF;
            ep.cursor = d_ctr + CT_CURF;
            ep.ovf_list = dst < kTiers ? ovf[dst - 1] : nullptr;
            ep.ovf_count = d_ctr + CT_OVF1 + dst - 1;  // appended behind what is already there
            ep.escalate_cost = 0xFFFFFFFFu;
            ep.cap_limit = 0xFFFFFFFFu;
            launch_kernel(dst, n_ovf, s);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemcpyAsync(h_ctr, d_ctr, kCtrBytes, hipMemcpyDeviceToHost, s));
            RC_TRY(wd_sync(idx, cx, s, "a re-run of overflowed segments"));
            PROF_DUMP("cascade");
        }
        ms_tier2 = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() -
                                                             t_casc0).count();
        if (h_ctr[CT_OVF1 + kTiers - 1]) {
            // (cannot happen: tier 7 holds the bound on the live arms of any segment -- unless a test shrank it)
            set_error("internal: %llu segment(s) overflowed the last extension tier (%llu arm slots)",
                      (unsigned long long)h_ctr[CT_OVF1 + kTiers - 1], (unsigned long long)heavy_cap64);
            return ASGART_E_CAP;
        }
        ms_longest_segment = 0.0;
        for (int t = 0; t < kTiers; ++t) ms_longest_segment = std::max(ms_longest_segment, (double)h_ctr[CT_SEGMAX1 + t] * 1e-5);
        if (opt.debug) {
            fprintf(stderr, "[asgart] longest single segment per tier (ms):");
            for (int t = 0; t < kTiers; ++t) fprintf(stderr, " %d: %.2f", t + 1, (double)h_ctr[CT_SEGMAX1 + t] * 1e-5);
            fprintf(stderr, "\n");
        }
        return 0;
    }

    // ---- records -> reference order -> families per pass ----------------------------------------------------------------------
    int32_t records() {
        HIP_TRY(hipEventRecord(cx.ev[4], s));
        const uint64_t n_rec = h_ctr[CT_SD];
        if (n_rec) {
            void *hp = nullptr;
            RC_TRY(cx.pinned((size_t)n_rec * sizeof(SdRec), &hp));
            h_recs = static_cast<const SdRec *>(hp);
            n_hrec = (size_t)n_rec;
        }
        // Reference order: chunk order, discovery order inside the chunk, arm order
        // inside the family == (segment start probe, family ordinal, creation number): sorted on
        // the GPU.  A tombstone voids its family; segments re-run by a later tier emit some
        // records twice (same key): keep one copy.
        if (n_rec) {
            RC_TRY(sort_records(w, w.fam_sds.as<SdRec>(), n_rec, s));
            HIP_TRY(hipMemcpyAsync(const_cast<SdRec *>(h_recs), w.rec_sorted.p, (size_t)n_rec * sizeof(SdRec),
                                   hipMemcpyDeviceToHost, s));
        }
        const auto t_post0 = std::chrono::steady_clock::now();
        RC_TRY(wd_sync(idx, cx, s, "the ordering of the records"));
        const auto t_post1 = std::chrono::steady_clock::now();
        // (sorted by segment start probe: pass 0's families first, then pass 1's ...; a family's key counts probes from
        // the start of its OWN pass, so that it equals the key a single-pass call gives the same family)
        // Assembly: the records are read once (48 bytes each: a memory-bound loop) -- large lists by up to four host threads,
        // each over a slice that starts at a family boundary, their parts joined in order.
        using Part = SearchCtx::FamPart;  // (kept by the call context: fresh vectors of this size cost their page faults every call)
        auto assemble = [&](size_t b0, size_t b1, Part &out) {
            int32_t pass = 0;
            for (size_t f0 = b0; f0 < b1;) {
                if (h_recs[f0].g_start == kVoidStart) break;  // unused slots of the waves' record chunks: sorted last
                size_t f1 = f0;
                while (f1 < n_hrec && h_recs[f1].g_start == h_recs[f0].g_start && h_recs[f1].fam_seq == h_recs[f0].fam_seq) ++f1;
                while (pass + 1 < n_passes && h_recs[f0].g_start >= (uint32_t)(pass + 1) * K) ++pass;
                if (h_recs[f1 - 1].create_seq != kTombstone) {
                    std::vector<asgart_proto_sd> &sds = out.sds[pass];
                    for (size_t j = f0; j < f1; ++j) {
                        if (j > f0 && h_recs[j].create_seq == h_recs[j - 1].create_seq) continue;
                        sds.push_back(h_recs[j].sd);
                    }
                    out.ends[pass].push_back(sds.size());
                    out.keys[pass].push_back(((uint64_t)(lo_lim + (h_recs[f0].g_start - (uint32_t)pass * K)) << 32) |
                                             (uint64_t)h_recs[f0].fam_seq);
                }
                f0 = f1;
            }
        };
        const size_t n_parts = n_hrec >= (1u << 17) ? 4 : 1;
        size_t bound[5] = {0, 0, 0, 0, n_hrec};
        for (size_t t = 1; t < n_parts; ++t) {  // slice boundaries moved forward to the next family start
            size_t b = std::max(bound[t - 1], n_hrec * t / n_parts);
            while (b > 0 && b < n_hrec && h_recs[b].g_start == h_recs[b - 1].g_start && h_recs[b].fam_seq == h_recs[b - 1].fam_seq) ++b;
            bound[t] = b;
        }
        for (size_t t = n_parts; t < 4; ++t) bound[t] = n_hrec;
        std::vector<Part> &parts = cx.fam_parts;
        parts.resize(4);
        for (size_t t = 0; t < 4; ++t)
            for (int32_t p = 0; p < 4; ++p) {
                parts[t].sds[p].clear();
                parts[t].ends[p].clear();
                parts[t].keys[p].clear();
                if (t < n_parts && p < n_passes)
                    parts[t].sds[p].reserve((bound[t + 1] - bound[t]) / (size_t)n_passes + (bound[t + 1] - bound[t]) / 4 + 16);
            }
        {
            std::vector<std::thread> workers;
            for (size_t t = 1; t < n_parts; ++t) workers.emplace_back([&, t]() { assemble(bound[t], bound[t + 1], parts[t]); });
            assemble(bound[0], bound[1], parts[0]);
            for (auto &th : workers) th.join();
        }
        for (int32_t p = 0; p < n_passes; ++p) {
            asgart_families *const fo = fams[p];
            size_t tot = 0, nf = 0;
            for (const Part &pt : parts) {
                tot += pt.sds[p].size();
                nf += pt.ends[p].size();
            }
            if (!tot && !nf) continue;
            fo->sds.reserve(fo->sds.size() + tot);
            fo->fam_offsets.reserve(fo->fam_offsets.size() + nf);
            fo->fam_keys.reserve(fo->fam_keys.size() + nf);
            for (const Part &pt : parts) {
                const size_t at = fo->sds.size();
                fo->sds.insert(fo->sds.end(), pt.sds[p].begin(), pt.sds[p].end());
                for (uint64_t e : pt.ends[p]) fo->fam_offsets.push_back(at + e);
                fo->fam_keys.insert(fo->fam_keys.end(), pt.keys[p].begin(), pt.keys[p].end());
            }
        }
        if (opt.debug)
            fprintf(stderr, "[asgart] records: %llu slots; ordering + copy to the host %.1f ms, families assembled in %.1f ms\n",
                    (unsigned long long)n_rec, std::chrono::duration<double, std::milli>(t_post1 - t_post0).count(),
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_post1).count());
        return 0;
    }

    int32_t fill_stats() {
        // ---- stats ------------------------------------------------------------------
        asgart_stats &stt = cx.stats;
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, cx.ev[0], cx.ev[1]));
        stt.ms_search = ms;
        HIP_TRY(hipEventElapsedTime(&ms, cx.ev[1], cx.ev[2]));
        stt.ms_scan = ms;
        HIP_TRY(hipEventElapsedTime(&ms, cx.ev[16], cx.ev[3]));
        stt.ms_fill = ms;
        HIP_TRY(hipEventElapsedTime(&ms, cx.ev[3], cx.ev[4]));
        stt.ms_extend = ms;
        stt.ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() -
                                                                t_host0).count();
        stt.probes_n_skipped = h_ctr[CT_N_SKIPPED];
        stt.probes_searched = h_ctr[CT_SEARCHED];
        stt.probes_card_skipped = h_ctr[CT_CARD_SKIPPED];
        stt.probes_with_hits = h_ctr[CT_WITH_HITS];
        stt.raw_hits = 0;  // (asgart_get_stats sums them up when asked: raw_hits_kernel)
        stt.filtered_hits = total_hits;
        stt.segments = n_seg;
        stt.families = stt.proto_sds = 0;
        for (int32_t p = 0; fam_out && p < n_passes; ++p) {
            stt.families += fams[p]->fam_offsets.size() - 1;
            stt.proto_sds += fams[p]->sds.size();
        }
        stt.passes = (uint64_t)n_passes;
        stt.search_launches = 1;
        stt.overflow_segments = n_overflow;
        stt.heavy_segments = n_heavy;

        stt.ms_extend_tier2 = ms_tier2;
        stt.ms_longest_tier = ms_longest_tier;
        stt.ms_longest_segment = ms_longest_segment;
        stt.split_segments = n_split_segments;
        stt.split_refused = n_split_refused;
        HIP_TRY(hipEventElapsedTime(&ms, cx.ev[0], cx.ev[11]));
        stt.ms_probe_count = ms;
        cx.has_last = true;
        cx.raw_done = false;
        return 0;
        return 0;
    }

    int32_t run() {
        RC_TRY(setup());
        if (nothing_to_do) return 0;
        for (int win_try = 0;; ++win_try) {  // (a sharded call widens its halos until every decision is safe)
            if (win_try > 40) {
                set_error("internal: shard window did not converge");
                return ASGART_E_CAP;
            }
            set_window();
            bool ambiguous = false;
            RC_TRY(front(&ambiguous));
            if (ambiguous) {
                look_back *= 8;
                continue;
            }
            // (the hit rows are being filled on a stream of their own; place() waits for them where it needs them)
            if (want_csr || !(fam_out && n_seg)) HIP_TRY(hipStreamWaitEvent(s, cx.ev[3], 0));
            if (want_csr) RC_TRY(csr_out());
            if (fam_out && n_seg) {
                RC_TRY(place());
                for (int attempt = 0;; ++attempt) {  // (again with a larger record buffer when it overflowed)
                    RC_TRY(run_tiers(attempt));
                    RC_TRY(join_ranges());
                    RC_TRY(finish_tiers());
                    if (h_ctr[CT_RANOUT]) break;
                    if (h_ctr[CT_SD] <= rec_cap) break;
                    if (attempt >= 3) {
                        set_error("internal: record buffer keeps overflowing");
                        return ASGART_E_CAP;
                    }
                    rec_cap = h_ctr[CT_SD] * 2;
                }
                if (h_ctr[CT_RANOUT]) {  // a segment runs past the look-ahead halo: widen it
                    look_ahead *= 8;
                    continue;
                }
                RC_TRY(records());
            } else {
                HIP_TRY(hipEventRecord(cx.ev[4], s));
                RC_TRY(wd_sync(idx, cx, s, "the hit rows"));
            }
            break;
        }
        return fill_stats();
    }
};

template <class SlotT>
static int32_t run_search_t(asgart_index *idx, SearchCtx &cx, const uint64_t *chunks, int64_t n_chunks_pass,
                            const asgart_settings *sts, int32_t n_passes, int32_t shard, int32_t n_shards,
                            bool want_csr, asgart_families *const *fams,
                            std::vector<uint8_t> *status_out, std::vector<uint64_t> *rowoff_out,
                            std::vector<uint64_t> *hits_out) {
    // (on the heap: the call's state holds its kernels' parameter blocks and a few KB of tables)
    std::unique_ptr<SearchCall<SlotT>> call(new (std::nothrow) SearchCall<SlotT>(idx, cx, chunks, n_chunks_pass, sts, n_passes, shard,
                                                                                n_shards, want_csr, fams, status_out, rowoff_out, hits_out));
    if (!call) {
        set_error("out of host memory");
        return ASGART_E_OOM;
    }
    return call->run();
}


// n_passes > 1: ONE job over the probes of all passes (sts differ in reverse / complement only; checked by the caller);
// fams: n_passes result objects, or null (the CSR surface of a single pass).
int32_t run_search_passes(asgart_index *idx, const uint64_t *chunks, int64_t n_chunks,
                          const asgart_settings *sts, int32_t n_passes, int32_t shard, int32_t n_shards, bool want_csr,
                          asgart_families *const *fams, std::vector<uint8_t> *status_out,
                          std::vector<uint64_t> *rowoff_out, std::vector<uint64_t> *hits_out,
                          volatile uint64_t *progress) {
    if (!idx || !sts || n_passes < 1 || n_passes > 4 || n_chunks < 0 || (n_chunks && !chunks)) {
        set_error("bad argument");
        return ASGART_E_ARG;
    }
    const asgart_settings *const st = &sts[0];
    if (st->max_gap_size == 0) {
        // the CLI always passes gap + probe_size (reference src/bin/asgart.rs:681), never 0
        set_error("max_gap_size must be >= 1 (it includes probe_size, src/bin/asgart.rs:681)");
        return ASGART_E_ARG;
    }
    if (n_shards < 1 || shard < 0 || shard >= n_shards || (n_passes > 1 && (want_csr || progress))) {
        set_error("bad shard %d of %d", shard, n_shards);
        return ASGART_E_ARG;
    }
    REFUSE_POISONED(idx);
    HIP_TRY(hipSetDevice(idx->device));
    // take a free per-call context; the keys can only change while no context is in use
    int which = 0;
    for (;;) {
        SearchCtx &probe = idx->acquire_one(&which);
        (void)probe;
        // The presence filter and the position-sorted lists are optimisations that cost more than they save in ONE pass
        // (option lazy_aux): an orientation gets its filter at its second search, the index its lists at its second call.
        // With lazy_aux no filter is built at all: an orientation's position bits start blank and its searches fill them in.
        bool want_sap, ready;
        int need_filter = -1;  // an orientation of this call whose filter is due and missing
        int need_blank = -1;   // ... or whose blank position bits are (lazy_aux)
        {
            std::lock_guard<std::mutex> lk(idx->mu);
            ready = idx->k == st->probe_size;
            const bool lazy = idx->opt.lazy_aux != 0;
            const bool filterable = !(idx->opt.kfilter_bits == 0 || idx->trimmed || st->probe_size > (uint64_t)kMaxKey);
            for (int32_t p = 0; p < n_passes && need_filter < 0 && need_blank < 0; ++p) {
                const int mode = (sts[p].reverse ? 2 : 0) | (sts[p].complement ? 1 : 0);
                if (!filterable || !ready || idx->filter_off[mode]) continue;
                if (lazy && idx->opt.posbits != 0) {
                    if (!idx->d_pbits[mode]) need_blank = mode;
                } else if (!(lazy && idx->mode_calls[mode] == 0) && !idx->d_filter[mode]) {
                    need_filter = mode;  // (lazy without position bits: the hashed filter, on second use)
                }
            }
            want_sap = !(lazy && (!ready || idx->calls_total == 0)) && !idx->sap_tried;
            if (ready && need_filter < 0 && need_blank < 0 && !(want_sap && !idx->d_sap)) {
                for (int32_t p = 0; p < n_passes; ++p) ++idx->mode_calls[(sts[p].reverse ? 2 : 0) | (sts[p].complement ? 1 : 0)];
                break;
            }
        }
        idx->release_one(which);
        if (!ready) RC_TRY(index_prepare(idx, st->probe_size));
        else if (want_sap && !idx->d_sap) RC_TRY(index_prepare_sap(idx, st->probe_size));
        else if (need_blank >= 0) RC_TRY(index_prepare_learned_bits(idx, st->probe_size, need_blank));
        else RC_TRY(index_prepare_filter(idx, st->probe_size, need_filter));
    }
    SearchCtx &cx = idx->ctx[which];
    cx.progress = progress;
    int32_t rc;
    if (idx->wide)
        rc = run_search_t<uint64_t>(idx, cx, chunks, n_chunks, sts, n_passes, shard, n_shards, want_csr, fams,
                                    status_out, rowoff_out, hits_out);
    else
        rc = run_search_t<uint32_t>(idx, cx, chunks, n_chunks, sts, n_passes, shard, n_shards, want_csr, fams,
                                    status_out, rowoff_out, hits_out);
    cx.progress = nullptr;
    bool trim_now = false;
    {   // (counted in PASSES: a host that runs the direct and the -RC pass as one passes call -- the CLI's shape -- has made its
        // two searches when that call returns, and must not sit on the sorter's ~100 GB of scratch until it destroys the index)
        std::lock_guard<std::mutex> lk(idx->mu);
        const uint64_t before = idx->calls_total;
        idx->calls_total += (uint64_t)n_passes;
        for (int32_t p = 0; p < n_passes; ++p) ++idx->pbits_uses[(sts[p].reverse ? 2 : 0) | (sts[p].complement ? 1 : 0)];
        trim_now = idx->opt.cache_calls > 0 && before < (uint64_t)idx->opt.cache_calls && idx->calls_total >= (uint64_t)idx->opt.cache_calls;
    }
    if (trim_now) {  // (option cache_calls: what the index build released goes back to the device now -- behind the caller's back:
                     // the hipFree of ~100 GB is not this call's business)
        const int dev = idx->device;
        background_call([dev]() {
            if (hipSetDevice(dev) == hipSuccess) BlockCache::trim();
            (void)hipGetLastError();
        });
    }
    if (rc == 0 && fams && n_shards == 1 && n_passes == 1) {
        // what asgart_search_duplications_passes orders by when it pipelines single-pass calls: the shortest extension seen
        // for the orientation (a call that shared the chip with another one measures longer, and the order must not flip
        // because of that)
        std::lock_guard<std::mutex> lk(idx->mu);
        double &t = idx->tail_ms[(st->reverse ? 2 : 0) | (st->complement ? 1 : 0)];
        t = t < 0.0 ? cx.stats.ms_extend : std::min(t, cx.stats.ms_extend);
    }
    idx->release_one(which);
    return rc;
}

int32_t run_search(asgart_index *idx, const uint64_t *chunks, int64_t n_chunks,
                   const asgart_settings *st, int32_t shard, int32_t n_shards, bool want_csr,
                   asgart_families *fam_out, std::vector<uint8_t> *status_out,
                   std::vector<uint64_t> *rowoff_out, std::vector<uint64_t> *hits_out,
                   volatile uint64_t *progress) {
    asgart_families *one[1] = {fam_out};
    return run_search_passes(idx, chunks, n_chunks, st, 1, shard, n_shards, want_csr, fam_out ? one : nullptr, status_out,
                             rowoff_out, hits_out, progress);
}

}  // namespace asgart

using namespace asgart;

extern "C" {

static int32_t search_shard_impl(asgart_index *idx, const uint64_t *chunks, int64_t n_chunks,
                                 const asgart_settings *settings, int32_t shard, int32_t n_shards,
                                 volatile uint64_t *progress, asgart_families **out) {
    if (!out) {
        set_error("out is NULL");
        return ASGART_E_ARG;
    }
    *out = nullptr;
    asgart_families *f = new (std::nothrow) asgart_families();
    if (!f) {
        set_error("out of host memory");
        return ASGART_E_OOM;
    }
    int32_t rc = run_search(idx, chunks, n_chunks, settings, shard, n_shards, false, f, nullptr,
                            nullptr, nullptr, progress);
    if (rc != 0) {
        delete f;
        return rc;
    }
    *out = f;
    return 0;
}

int32_t asgart_search_duplications_shard(asgart_index *idx, const uint64_t *chunks,
                                         int64_t n_chunks, const asgart_settings *settings,
                                         int32_t shard, int32_t n_shards, asgart_families **out) {
    return search_shard_impl(idx, chunks, n_chunks, settings, shard, n_shards, nullptr, out);
}

int32_t asgart_search_duplications_ex(asgart_index *idx, const uint64_t *chunks, int64_t n_chunks,
                                      const asgart_settings *settings, int32_t shard, int32_t n_shards,
                                      volatile uint64_t *progress, asgart_families **out) {
    return search_shard_impl(idx, chunks, n_chunks, settings, shard, n_shards, progress, out);
}

int32_t asgart_search_duplications(asgart_index *idx, const uint64_t *chunks, int64_t n_chunks,
                                   const asgart_settings *settings, volatile uint64_t *progress,
                                   asgart_families **out) {
    int32_t rc = search_shard_impl(idx, chunks, n_chunks, settings, 0, 1, progress, out);
    if (rc == 0 && progress) {
        // (also written earlier, when the chip-wide phases were over: see run_search_t) 
        const uint64_t k = settings->probe_size, step = k / 2;
        for (int64_t c = 0; c < n_chunks; ++c)
            progress[c] = (uint64_t)probes_in_chunk(chunks[2 * c + 1], k, step,
                                                    settings->min_duplication_length) * step;
    }
    return rc;
}

int32_t asgart_search_duplications_multi(asgart_index *const *indices, int32_t n_devices,
                                         const uint64_t *chunks, int64_t n_chunks,
                                         const asgart_settings *settings, volatile uint64_t *progress,
                                         asgart_families **out) {
    if (!out) {
        set_error("out is NULL");
        return ASGART_E_ARG;
    }
    *out = nullptr;
    if (!indices || n_devices < 1 || n_devices > 64) {
        set_error("bad argument: %d devices", n_devices);
        return ASGART_E_ARG;
    }
    for (int32_t r = 0; r < n_devices; ++r)
        if (!indices[r] || indices[r]->n != indices[0]->n || indices[r]->n_sa != indices[0]->n_sa) {
            set_error("index %d is NULL or not a replica of index 0", r);
            return ASGART_E_ARG;
        }
    // one host thread per device; shard r of n_devices (no exchange between shards)
    std::vector<asgart_families> parts((size_t)n_devices);
    std::vector<int32_t> rcs((size_t)n_devices, 0);
    std::vector<std::string> errs((size_t)n_devices);
    std::vector<std::thread> workers;
    for (int32_t r = 0; r < n_devices; ++r)
        workers.emplace_back([&, r]() {
            rcs[r] = run_search(indices[r], chunks, n_chunks, settings, r, n_devices, false, &parts[r], nullptr,
                                nullptr, nullptr, nullptr);
            if (rcs[r] != 0) errs[r] = asgart_last_error();  // the message is thread-local
        });
    for (auto &t : workers) t.join();
    for (int32_t r = 0; r < n_devices; ++r)
        if (rcs[r] != 0) {
            set_error("shard %d of %d: %s", r, n_devices, errs[r].c_str());
            return rcs[r];
        }
    asgart_families *f = new (std::nothrow) asgart_families();
    if (!f) {
        set_error("out of host memory");
        return ASGART_E_OOM;
    }
    // merge the shards' families by key (segment start probe, family ordinal): reference order, whichever way
    // the segments were dealt out
    struct Ref {
        uint64_t key;
        int32_t r;
        uint32_t j;
    };
    std::vector<Ref> refs;
    for (int32_t r = 0; r < n_devices; ++r)
        for (size_t j = 0; j < parts[r].fam_keys.size(); ++j) refs.push_back(Ref{parts[r].fam_keys[j], r, (uint32_t)j});
    std::stable_sort(refs.begin(), refs.end(), [](const Ref &a, const Ref &b) { return a.key < b.key; });
    f->fam_offsets.assign(1, 0);
    for (const Ref &e : refs) {
        const asgart_families &p = parts[e.r];
        f->sds.insert(f->sds.end(), p.sds.begin() + (ptrdiff_t)p.fam_offsets[e.j], p.sds.begin() + (ptrdiff_t)p.fam_offsets[e.j + 1]);
        f->fam_offsets.push_back(f->sds.size());
        f->fam_keys.push_back(e.key);
    }
    if (progress && settings) {
        const uint64_t k = settings->probe_size, step = k / 2;
        for (int64_t c = 0; c < n_chunks; ++c)
            progress[c] = (uint64_t)probes_in_chunk(chunks[2 * c + 1], k, step, settings->min_duplication_length) * step;
    }
    *out = f;
    return 0;
}

int32_t asgart_search_duplications_passes(asgart_index *idx, const uint64_t *chunks, int64_t n_chunks,
                                          const asgart_settings *settings, int32_t n_passes,
                                          asgart_families **out) {
    return asgart_search_duplications_passes_shard(idx, chunks, n_chunks, settings, n_passes, 0, 1, out);
}

int32_t asgart_search_duplications_passes_shard(asgart_index *idx, const uint64_t *chunks, int64_t n_chunks,
                                                const asgart_settings *settings, int32_t n_passes,
                                                int32_t shard, int32_t n_shards, asgart_families **out) {
    if (!out || n_passes < 0 || (n_passes && !settings)) {
        set_error("bad argument");
        return ASGART_E_ARG;
    }
    for (int32_t j = 0; j < n_passes; ++j) out[j] = nullptr;
    if (!idx || n_chunks < 0 || (n_chunks && !chunks)) {
        set_error("bad argument");
        return ASGART_E_ARG;
    }
    if (n_passes == 0) return 0;
    // ---- the passes as ONE job (option fuse_passes, default) ------------------------------------------------------------
    // Passes that differ in orientation only -- the direct and the -RC run of one genome, what the call exists for --
    // share one probe sequence (pass 0's chunks, then pass 1's, ...: chunk order inside each pass as in
    // src/bin/asgart.rs:201-253): ONE probe search, scan, hit-row fill and placement over all their probes at full chip
    // rate, ONE launch per extension tier over the merged cost-sorted segment list -- every pass's longest segments
    // start at t = 0 of the extension on compute units of their own.  A SHARDED call is the same job over the shard's slice
    // of every pass (run_search_t: one window per pass).  (Pipelined as two calls on two contexts -- below, kept for
    // passes with different settings and for inputs whose extension is ONE segment -- the second pass's front crawled
    // behind the first one's persistent extension workgroups: 117 ms instead of 28 at GRCh38 size.)
    bool timed_pipelined = false;
    {
        bool fusable = idx->opt.fuse_passes != 0 && n_passes >= 2 && n_passes <= 4;  // (option fuse_passes)
        for (int32_t j = 1; fusable && j < n_passes; ++j)
            fusable = settings[j].probe_size == settings[0].probe_size && settings[j].max_gap_size == settings[0].max_gap_size &&
                      settings[j].min_duplication_length == settings[0].min_duplication_length &&
                      settings[j].max_cardinality == settings[0].max_cardinality;
        if (fusable) {  // (2^32 probes per call: the passes together)
            const uint64_t k = settings[0].probe_size, step = k / 2;
            uint64_t P1 = 0;
            for (int64_t c = 0; c < n_chunks && step; ++c)
                P1 += probes_in_chunk(chunks[2 * c + 1], k, step, settings[0].min_duplication_length);
            fusable = step && P1 * (uint64_t)n_passes < 0xFFFFFF00ull;
        }
        uint64_t modes_sig = 0;
        for (int32_t j = 0; fusable && j < n_passes; ++j)
            modes_sig = modes_sig * 4u + ((settings[j].reverse ? 2u : 0u) | (settings[j].complement ? 1u : 0u));
        auto same_as_verdict = [&]() {
            const asgart_index::FuseVerdict &v = idx->fuse_verdict;
            return v.n_passes == n_passes && v.k == settings[0].probe_size && v.G == settings[0].max_gap_size &&
                   v.M == settings[0].min_duplication_length && v.C == settings[0].max_cardinality && v.modes == modes_sig &&
                   v.shard == shard && v.n_shards == n_shards;
        };
        // What the calls with these settings have measured decides -- for an unsharded call: what pipelining can win is the
        // other passes' front beside the one long segment, and a shard's front is 1/N of it, while the second pass's front
        // crawling behind the first one's persistent workgroups costs a shard as much as it costs the whole call (GRCh38-
        // shaped, the shards of N = 8 timed alone: 32-64 ms as one job, 47-78 ms pipelined).  Once a call that ran as one
        // job has seen ONE segment be its extension, the calls are timed both ways in turn (one job, pipelined, one job,
        // pipelined) and the faster way is kept.
        const bool may_pipeline = fusable && idx->opt.fuse_passes == 1 && n_shards == 1;
        if (may_pipeline) {
            std::lock_guard<std::mutex> lk(idx->mu);
            const asgart_index::FuseVerdict &v = idx->fuse_verdict;
            if (same_as_verdict()) {
                if (v.n_fused >= 2 && v.n_piped >= 2) fusable = v.ms_fused <= v.ms_piped;  // (measured: stands from here on)
                else if (v.pole && v.n_piped < 2 && v.n_piped < v.n_fused) fusable = false;
            }
        }
        const auto t_call0 = std::chrono::steady_clock::now();
        auto call_ms = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_call0).count(); };
        timed_pipelined = may_pipeline && !fusable;  // (the pipelined path below is timed where it returns)
        if (fusable) {
            std::lock_guard<std::mutex> pass_lock(idx->pass_mu);
            struct Owns {
                Owns() { asgart::tl_owns_pass_mu = true; }
                ~Owns() { asgart::tl_owns_pass_mu = false; }
            } owns;
            std::vector<asgart_families *> fams((size_t)n_passes, nullptr);
            for (auto &f : fams) {
                f = new (std::nothrow) asgart_families();
                if (!f) {
                    for (auto *g : fams) delete g;
                    set_error("out of host memory");
                    return ASGART_E_OOM;
                }
            }
            const int32_t rc = run_search_passes(idx, chunks, n_chunks, settings, n_passes, shard, n_shards, false, fams.data(),
                                                 nullptr, nullptr, nullptr, nullptr);
            if (rc != 0) {
                for (auto *f : fams) delete f;
                return rc;
            }
            for (int32_t j = 0; j < n_passes; ++j) out[j] = fams[(size_t)j];
            {   // is one segment the extension?  (then the passes MAY be better pipelined: the other pass's front beside it)
                const double ms = call_ms();
                std::lock_guard<std::mutex> lk(idx->mu);
                const asgart_stats &stt = idx->ctx[idx->last_ctx].stats;
                asgart_index::FuseVerdict &v = idx->fuse_verdict;
                if (!same_as_verdict()) {
                    v = asgart_index::FuseVerdict{};
                    v.n_passes = n_passes;
                    v.k = settings[0].probe_size;
                    v.G = settings[0].max_gap_size;
                    v.M = settings[0].min_duplication_length;
                    v.C = settings[0].max_cardinality;
                    v.modes = modes_sig;
                    v.shard = shard;
                    v.n_shards = n_shards;
                }
                const bool pole = n_shards == 1 && stt.passes == (uint64_t)n_passes && stt.ms_extend > 0.0 &&
                                  stt.ms_longest_segment * 100.0 > stt.ms_extend * (double)idx->opt.fuse_pole_pct;
                // (the first call with these settings is not timed: it may be the index's first; nor is one that refused a cut --
                // the index is still learning which ranges join up, the next call will be shorter -- unless three in a row did)
                if (v.seen && (stt.split_refused == 0 || ++v.unsettled >= 3)) {
                    ++v.n_fused;
                    v.ms_fused = std::min(v.ms_fused, ms);
                    v.unsettled = 0;
                }
                v.seen = true;
                v.pole = pole;
                if (idx->opt.debug)
                    fprintf(stderr, "[asgart] passes as one job, %.1f ms: longest segment %.1f ms of %.1f ms of extension%s\n", ms,
                            stt.ms_longest_segment, stt.ms_extend, pole ? " -> timed against pipelined calls" : "");
            }
            return 0;
        }
    }
    // (pipelined: timed for the verdict when it was that verdict's choice)
    struct TimedPipelined {
        asgart_index *idx;
        bool on;
        std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
        bool ok = false;
        ~TimedPipelined() {
            if (!on || !ok) return;
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            std::lock_guard<std::mutex> lk(idx->mu);
            uint64_t refused = 0;
            for (int c = 0; c < asgart::kNumCtx; ++c) refused += idx->ctx[c].stats.split_refused;
            if (refused != 0 && ++idx->fuse_verdict.unsettled < 3) return;  // (still learning which ranges join up: not a sample)
            idx->fuse_verdict.unsettled = 0;
            ++idx->fuse_verdict.n_piped;
            idx->fuse_verdict.ms_piped = std::min(idx->fuse_verdict.ms_piped, ms);
            if (idx->opt.debug) fprintf(stderr, "[asgart] passes pipelined, %.1f ms (timed against the passes as one job)\n", ms);
        }
    } timed{idx, timed_pipelined};
    // issue order: longest extension first (what is known from earlier calls; an orientation never run yet
    // counts as longest, reversed ones ahead of the others: their tandem arrays are walked against the
    // whole text instead of the part behind the probe)
    std::vector<int32_t> order((size_t)n_passes);
    for (int32_t j = 0; j < n_passes; ++j) order[j] = j;
    double tail_ms[4];
    {   // (a plain call on the other context may be updating them: run_search writes under the same lock)
        std::lock_guard<std::mutex> lk(idx->mu);
        for (int m = 0; m < 4; ++m) tail_ms[m] = idx->tail_ms[m];
    }
    auto weight = [&](int32_t j) {
        const int mode = (settings[j].reverse ? 2 : 0) | (settings[j].complement ? 1 : 0);
        const double t = tail_ms[mode];
        return t >= 0.0 ? t : 1e30 + (settings[j].reverse ? 1e29 : 0.0);
    };
    std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return weight(a) > weight(b); });
    // one host thread per pass; thread p starts its call when pass p-1 reports its probes as searched
    // (or has returned); a call blocks in the index until one of the two contexts is free
    std::vector<asgart_families *> fams((size_t)n_passes, nullptr);
    std::vector<int32_t> rcs((size_t)n_passes, 0);
    std::vector<std::string> errs((size_t)n_passes);
    std::vector<std::vector<uint64_t>> prog((size_t)n_passes, std::vector<uint64_t>((size_t)std::max<int64_t>(n_chunks, 1), 0));
    std::vector<std::atomic<int>> finished((size_t)n_passes);
    for (auto &f : finished) f.store(0);
    auto searched = [&](int32_t p) {
        if (finished[p].load(std::memory_order_acquire)) return true;
        const volatile uint64_t *pr = prog[p].data();
        for (int64_t c = 0; c < n_chunks; ++c)
            if (pr[c]) return true;
        return false;
    };
    // the first pass runs on the calling thread, the others on the index's own worker threads
    std::lock_guard<std::mutex> pass_lock(idx->pass_mu);
    struct Owns {  // (index_prepare's prewarm, reached from body(0) below, must not try to lock it again)
        Owns() { asgart::tl_owns_pass_mu = true; }
        ~Owns() { asgart::tl_owns_pass_mu = false; }
    } owns;
    while ((int32_t)idx->pass_workers.size() + 1 < n_passes) idx->pass_workers.emplace_back(new asgart::PassWorker());
    auto body = [&](int32_t p) {
            if (p > 0)
                while (!searched(p - 1)) std::this_thread::sleep_for(std::chrono::microseconds(100));
            const int32_t j = order[p];
            asgart_families *f = new (std::nothrow) asgart_families();
            if (!f) {
                rcs[p] = ASGART_E_OOM;
                errs[p] = "out of host memory";
            } else {
                rcs[p] = run_search(idx, chunks, n_chunks, &settings[j], shard, n_shards, false, f, nullptr, nullptr,
                                    nullptr, prog[p].data());
                if (rcs[p] != 0) {
                    errs[p] = asgart_last_error();  // the message is thread-local
                    delete f;
                } else {
                    fams[p] = f;
                }
            }
            finished[p].store(1, std::memory_order_release);
    };
    for (int32_t p = 1; p < n_passes; ++p) idx->pass_workers[(size_t)p - 1]->submit([&body, p]() { body(p); });
    body(0);
    for (int32_t p = 1; p < n_passes; ++p) idx->pass_workers[(size_t)p - 1]->wait();
    for (int32_t p = 0; p < n_passes; ++p)
        if (rcs[p] != 0) {
            for (auto *f : fams) delete f;
            set_error("pass %d of %d: %s", order[p], n_passes, errs[p].c_str());
            return rcs[p];
        }
    for (int32_t p = 0; p < n_passes; ++p) out[order[p]] = fams[p];
    timed.ok = true;
    return 0;
}

void asgart_families_counts(const asgart_families *f, uint64_t *n_families, uint64_t *n_sds) {
    if (n_families) *n_families = f ? f->fam_offsets.size() - 1 : 0;
    if (n_sds) *n_sds = f ? f->sds.size() : 0;
}

void asgart_families_copy(const asgart_families *f, uint64_t *fam_offsets, asgart_proto_sd *sds) {
    if (!f) return;
    if (fam_offsets) memcpy(fam_offsets, f->fam_offsets.data(), f->fam_offsets.size() * 8);
    if (sds && !f->sds.empty()) memcpy(sds, f->sds.data(), f->sds.size() * sizeof(asgart_proto_sd));
}

void asgart_families_keys(const asgart_families *f, uint64_t *keys) {
    if (f && keys && !f->fam_keys.empty()) memcpy(keys, f->fam_keys.data(), f->fam_keys.size() * 8);
}

void asgart_families_free(asgart_families *f) { delete f; }

int64_t asgart_probe_hits(asgart_index *idx, const uint64_t *chunks, int64_t n_chunks,
                          const asgart_settings *settings, uint8_t *status, uint64_t *row_offsets,
                          uint64_t *hits, uint64_t *n_hits) {
    std::vector<uint8_t> st;
    std::vector<uint64_t> ro, hv;
    int32_t rc = run_search(idx, chunks, n_chunks, settings, 0, 1, true, nullptr, &st, &ro, &hv, nullptr);
    if (rc != 0) return rc;
    if (n_hits) *n_hits = hv.size();
    if (status) {
        if (!st.empty()) memcpy(status, st.data(), st.size());
        memcpy(row_offsets, ro.data(), ro.size() * 8);
        if (!hv.empty()) memcpy(hits, hv.data(), hv.size() * 8);
    }
    return (int64_t)st.size();
}

int32_t asgart_get_stats(asgart_index *idx, uint32_t flags, asgart_stats *out) {
    if (!idx || !out) {
        set_error("bad argument");
        return ASGART_E_ARG;
    }
    idx->acquire_all();
    // bits 8.. of flags select a specific context (ASGART_STATS_CTX(i) = (i+1) << 8); default:
    // the context of the most recent search call
    const int sel = (int)((flags >> 8) & 0xFFu);
    SearchCtx &cx = idx->ctx[sel >= 1 && sel <= kNumCtx ? sel - 1 : idx->last_ctx];
    int32_t rc = [&]() -> int32_t {
        if ((flags & (ASGART_STATS_YARDSTICK | ASGART_STATS_RAW_HITS)) && cx.has_last && cx.last_P && !cx.raw_done) {
            // raw_hits of the last call: summed up now, from the per-probe arrays it left in this context's workspace (a probe
            // the position filter answered has no interval there: it is looked up now)
            REFUSE_POISONED(idx);
            HIP_TRY(hipSetDevice(idx->device));
            unsigned long long *d_ctr = cx.ws.counters.as<unsigned long long>();
            hipStream_t s = cx.stream;
            const RunParams &rp = cx.last_rp;
            HIP_TRY(hipMemsetAsync(d_ctr + CT_RAW_HITS, 0, 8, s));
            const unsigned g_raw = std::min<uint32_t>(rp.n_tiles(1024u), 256u * 16u);
            if (idx->wide)
                raw_hits_kernel<uint64_t><<<g_raw, 256, 0, s>>>(idx->view<uint64_t>(), rp, cx.ws.p_filt.as<uint32_t>() - rp.g_lo,
                                                                cx.ws.p_raw.as<uint32_t>() - rp.g_lo, d_ctr + CT_RAW_HITS);
            else
                raw_hits_kernel<uint32_t><<<g_raw, 256, 0, s>>>(idx->view<uint32_t>(), rp, cx.ws.p_filt.as<uint32_t>() - rp.g_lo,
                                                                cx.ws.p_raw.as<uint32_t>() - rp.g_lo, d_ctr + CT_RAW_HITS);
            HIP_TRY(hipGetLastError());
            unsigned long long v = 0;
            HIP_TRY(read_back(&v, d_ctr + CT_RAW_HITS, 8, s));  // (polled drain first: common.hpp)
            cx.stats.raw_hits = v;
            cx.raw_done = true;
        }
        if ((flags & ASGART_STATS_YARDSTICK) && cx.has_last && cx.last_P) {
            REFUSE_POISONED(idx);
            HIP_TRY(hipSetDevice(idx->device));
            unsigned long long *d_ctr = cx.ws.counters.as<unsigned long long>();
            hipStream_t s = cx.stream;
            HIP_TRY(hipMemsetAsync(d_ctr + CT_BISECT, 0, 8, s));
            RunParams rp = cx.last_rp;
            for (uint32_t p = 0; p < rp.n_passes && p < 4u; ++p) {  // (the filters as they are now)
                rp.flt[p] = idx->d_filter[rp.mode_of_pass(p)];
                rp.pbits[p] = idx->opt.posbits ? idx->d_pbits[rp.mode_of_pass(p)] : nullptr;
                // (a pass whose position bits were blank when the call ran looked every probe up)
                if ((rp.blank >> p) & 1u) rp.flt[p] = rp.pbits[p] = nullptr;
            }
            rp.learn = 0;
            rp.flt_bits = idx->filter_bits;
            const unsigned g = rp.n_tiles(256u);
            if (idx->wide)
                yardstick_kernel<uint64_t><<<g, 256, 0, s>>>(idx->view<uint64_t>(), rp,
                                                             cx.ws.p_filt.as<uint32_t>() - rp.g_lo, d_ctr);
            else
                yardstick_kernel<uint32_t><<<g, 256, 0, s>>>(idx->view<uint32_t>(), rp,
                                                             cx.ws.p_filt.as<uint32_t>() - rp.g_lo, d_ctr);
            HIP_TRY(hipGetLastError());
            // accounting pass of the probe-search kernels (same control flow, loads and stores priced
            // in bytes instead of executed); the work list of the large intervals is the one the
            // call left in the workspace
            HIP_TRY(hipMemsetAsync(d_ctr + CT_ALG_BYTES, 0, 16, s));
            HIP_TRY(hipMemsetAsync(d_ctr + CT_ALG_BYTES16, 0, 8, s));
            const unsigned gp = rp.n_tiles((uint32_t)kProbeBlock);
            auto account = [&](auto slot_tag) {
                using SlotT = decltype(slot_tag);
                IndexView<SlotT> ix = idx->view<SlotT>();
                probe_count_kernel<SlotT, true><<<gp, kProbeThreads, 0, s>>>(
                    ix, rp, nullptr, nullptr, nullptr, nullptr, nullptr, d_ctr);
                big_count_kernel<SlotT, true><<<2048, 256, 0, s>>>(
                    ix, rp, cx.ws.p_lo.as<SlotT>() - rp.g_lo, cx.ws.p_raw.as<uint32_t>() - rp.g_lo, nullptr,
                    cx.ws.big_list.as<uint32_t>(), d_ctr);
                if (ix.sap)
                    rank_count_kernel<SlotT, true><<<2048, 256, 0, s>>>(
                        ix, rp, cx.ws.p_lo.as<SlotT>() - rp.g_lo, cx.ws.p_raw.as<uint32_t>() - rp.g_lo, nullptr,
                        cx.ws.rank_list.as<uint32_t>(), nullptr, d_ctr);
            };
            if (idx->wide) account(uint64_t{}); else account(uint32_t{});
            HIP_TRY(hipGetLastError());
            unsigned long long v = 0, ab[2] = {0, 0}, a16 = 0;
            HIP_TRY(read_back(&v, d_ctr + CT_BISECT, 8, s));  // (polled drain first: common.hpp)
            HIP_TRY(read_back(ab, d_ctr + CT_ALG_BYTES, 16, s));
            HIP_TRY(read_back(&a16, d_ctr + CT_ALG_BYTES16, 8, s));
            cx.stats.bisect_steps = v;
            cx.stats.search_bytes = ab[0];
            cx.stats.probes_filter_rejected = ab[1];
            cx.stats.search_bytes_wide_loads = a16;
        }
        *out = cx.stats;
        return 0;
    }();
    idx->release_all();
    return rc;
}

}  // extern "C"
