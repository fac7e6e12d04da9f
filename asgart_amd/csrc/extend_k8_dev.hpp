// extend_k8_dev.hpp -- "K8": the arm-resident extension kernel with SPECIALISED waves and ONE barrier per hit-probe
// (tier 3: the long dense segments, and every run over a range of a cut segment).
//
// Same automaton as the other extension kernels (reference src/automaton.rs:57-204; representation of
// pipeline_dev.hpp: only live arms are kept, winners by creation number, families by records), same hit table and
// per-arm code as K6 (extend_fast_dev.hpp).  What changes is who runs what.
//
// A wave issues at most one instruction every four cycles, so the time of a hit-probe is the length of the longest
// instruction stream any one wave runs for it.  In K6 every wave that holds arms runs everything: the bookkeeping of
// the probe loop, the ranking of the empty slots and of the unmatched hits, then its arms -- ~1 250 instructions per
// probe on the busy waves of a tandem-array segment, of which the arms themselves are a tenth.  Here
//
//   * the ARM waves (all but the last two) do nothing but their arms: per step they read a command block, pull the
//     arms they were assigned a step earlier, resolve the previous probe for their arms (winners -> ExtendArm / age /
//     retire), offer to the current probe's hits, and publish their free-slot counts;
//   * the RANKING wave holds no arms: it decides flushes and overflow, ranks the unmatched hits of the previous probe
//     into a compact list, ranks the empty slots from the counts the arm waves published, and makes the FIRST offers
//     of the new arms itself -- a newborn arm's window is a function of its hit alone (x + 1, threshold(k) + k - 1)
//     and its creation number is the rank just computed -- leaving the candidates it found beside the hit.  (With the
//     new arms' offers made by the arm waves -- the predecessor of this kernel, "K7", removed in round 6 -- a step
//     needed a second barrier: the arm waves had to wait for the ranking before they could create the arms.)
//   * the PLANNING wave walks the probe sequence (batches, quiet runs, generation numbers, segment end) two steps ahead
//     and writes the commands;
//   * the top arm waves (which hold no arms while a segment is small) index the hits of the next probe.
//
//     arm waves   pull the arms born of t-2 (ranked during step s-1) -> resolve t-1 -> offers to t -> free counts
//     ranking     flush decision of t-1, unmatched hits of t-1 -> list, their offers to t, slot ranks
//     planning    the command of step s+2
//     top waves   index the hits of t+1 (from HBM when t+1 opens a batch: its rows reach LDS during this step)
//     -- barrier --
//
// Slots: the ranking wave assigns the r-th new arm to the r-th empty slot in (layer, wave, lane) order from the counts
// the arm waves published at the end of the step before; the pull that is still pending when it does so (assigned a step
// earlier) is subtracted -- the ranking wave knows how many slots of each (wave, layer) it gave away -- so no slot is
// given twice, and a wave pulls at most the number it was given, into whatever lanes are empty by then.
// Hit rows: three buffers (a batch is staged while the two before it may still be read).  Generation wrap: a step that
// only resolves (and clears the tables), a step that only indexes, then the probe.
// Creation numbers, record keys and every transition are K6's: results are identical (the tier tests force every
// segment through this kernel as well).
#pragma once

#include "extend_fast_dev.hpp"

namespace asgart {

// Diagnostic build (-DASGART_PROFILE_EXTEND): where the waves of a step spend their time -- per wave class (control,
// arm wave 0, last arm wave) the cycles from the start of a step to barrier 1, waiting there, from barrier 1 to
// barrier 2, waiting there; summed into ctr[40..55] (printed by the host's profile dump).
#ifdef ASGART_PROFILE_EXTEND
#define K7T_DECL unsigned long long k7t[4] = {0, 0, 0, 0}, k7t0 = 0, k7n = 0
#define K7T_MARK() k7t0 = __builtin_amdgcn_s_memtime()
#define K7T_LAP(j)                                                   \
    do {                                                             \
        const unsigned long long k7now = __builtin_amdgcn_s_memtime(); \
        k7t[j] += k7now - k7t0;                                      \
        k7t0 = k7now;                                                \
    } while (0)
#define K7T_STEP() ++k7n
#define K7C(slot, v) do { if (wave == 0u && lane == 0) atomicAdd(&P.ctr[slot], (unsigned long long)(v)); } while (0)
#define K7U_DECL unsigned long long k7u[8] = {0, 0, 0, 0, 0, 0, 0, 0}, k7u0 = 0
#define K7U_MARK() k7u0 = __builtin_amdgcn_s_memtime()
#define K7U_LAP(j)                                                   \
    do {                                                             \
        const unsigned long long k7now = __builtin_amdgcn_s_memtime(); \
        k7u[j] += k7now - k7u0;                                      \
        k7u0 = k7now;                                                \
    } while (0)
#define K7U_FLUSH()                                                                       \
    do {                                                                                  \
        if (lane == 0)                                                                    \
            for (int k7j = 0; k7j < 8; ++k7j) atomicAdd(&P.ctr[57 + k7j], k7u[k7j]);       \
        for (int k7j = 0; k7j < 8; ++k7j) k7u[k7j] = 0;                                   \
    } while (0)
#define K7T_FLUSH(cls)                                                                      \
    do {                                                                                    \
        if (lane == 0)                                                                      \
            for (int k7j = 0; k7j < 4; ++k7j) atomicAdd(&P.ctr[40 + 4 * (cls) + k7j], k7t[k7j]); \
        if (lane == 0 && (cls) == 0) atomicAdd(&P.ctr[52], k7n);                            \
        k7t[0] = k7t[1] = k7t[2] = k7t[3] = 0;                                              \
        k7n = 0;                                                                            \
    } while (0)
#else
#define K7T_DECL
#define K7T_MARK()
#define K7T_LAP(j)
#define K7T_STEP()
#define K7C(slot, v)
#define K7U_DECL
#define K7U_MARK()
#define K7U_LAP(j)
#define K7U_FLUSH()
#define K7T_FLUSH(cls)
#endif

// A taken branch costs a lone wave ~40 cycles (tools/ubench_branch.hip) and a step of an arm wave holds dozens of
// conditions that almost never hold (a cooperative arm, a record to write, a third table row, a stash, a late probe):
// the rare side of each is marked, so that the compiler lays the common path out as fall-through.
#define K7_RARE(x) __builtin_expect(!!(x), 0)
#define K7_USUAL(x) __builtin_expect(!!(x), 1)

// command flags (control wave -> everyone, one block per step)
constexpr uint32_t K7_PREV = 1u;    // a previous probe is to be resolved (and its new arms created)
constexpr uint32_t K7_CUR = 2u;     // a current probe receives offers
constexpr uint32_t K7_LATE = 4u;    // the current probe's hits are indexed in interval A, all offers made in interval B
constexpr uint32_t K7_LAST = 8u;    // the step loop ends behind this step
constexpr uint32_t K7_GIVEUP = 16u; // not for this kernel (a probe with more hits than the staging area): leave at once
// mid-step flags (decided in interval A, acted upon in interval B)
constexpr uint32_t K7_OVF = 1u;     // more arms than slots: the segment is given up
constexpr uint32_t K7_STAGE = 2u;   // stage a batch of hit rows
constexpr uint32_t K7_CLEAR = 4u;   // clear both hit tables (generation wrap)


constexpr uint32_t K8_CLEARNOW = 32u;  // command flag: every thread clears the hit tables at the top of this step
// pull block flags
constexpr uint32_t K8_OVF = 1u;        // more arms than slots: the segment is given up
constexpr uint32_t K8_STILL = 2u;      // the new arms of this block die of the quiet probes behind their birth: none is created
constexpr uint32_t K8_BIG = 4u;        // more than 64 new arms: the ranking wave offered for the first 64 only, every wave takes a
                                       // share of the others at the top of the next step (one extra barrier in such a step)

// RANGE: a work item is not a segment but a RUN over part of one (struct RangeRun, pipeline_dev.hpp) --
// long segments cut into ranges that run side by side, each from an empty arm list some way in front of its cut.  What
// differs from a whole segment: where the walk starts and stops; creation numbers that do not depend on what the run has
// seen before (needle offset of the creating probe relative to the segment's first, then the hit's index: the same arm
// gets the same number in every run that holds it, and the order of the numbers is the creation order); family ordinals
// that count the flushes from the cut on (the host adds the ranges before); records only from the cut on; and at the end
// the live arms and the family state are written out, for the comparison that decides whether the cut was sound.
template <class PosT, int S, int NT, int HB, int kRows = 1024, int kE = 2, bool RANGE = false>
__global__ __launch_bounds__(NT) void extend_k8_kernel(ExtParams<PosT> P) {
    constexpr int NW = NT / 64, NWA = NW - 2;  // waves; arm waves (then the RANKING wave and the PLANNING wave)
    constexpr int CAP = S * NWA * 64;
    constexpr int NE = S * NWA;                // (layer, wave) entries of the free counts
    constexpr uint32_t kNone = 0xFFFFFFFFu;    // best[]: no arm accepts this hit
    constexpr uint32_t kNever = 0xFFFFFFFEu;   // what a candidate read of an idle lane returns: no creation number
    constexpr uint32_t kCoop = 0xFFFFFFFFu;    // candidate register: more than three / wide window / stash overflow
    constexpr uint32_t kStash = 64;
    constexpr uint32_t kRowsWalk = 62;
    constexpr uint32_t kBitWords = (uint32_t)kRows / 32u;
    constexpr bool kWidePos = sizeof(PosT) == 8;
    constexpr uint32_t kTagShift = kWidePos ? 42u : 32u;
    constexpr uint32_t kGenMax = kWidePos ? 12u : 22u;
    constexpr unsigned long long kPosMask = (1ull << kTagShift) - 1ull;
    constexpr uint32_t kTabBytes = (uint32_t)(kRows * kE * 8);  // one hit table
    constexpr uint32_t kCmdWords = 32;
    constexpr uint32_t kNewMax = kWidePos ? (uint32_t)HB / 2u : (uint32_t)HB;  // new arms of one probe (more: given up)
    using WinT = typename std::conditional<kWidePos, uint64_t, uint32_t>::type;
    static_assert(!RANGE || S * NWA * 64 <= (int)kRunDumpCap, "runs over ranges: dump capacity");
    static_assert(NW >= 4 && HB <= 1024 && S <= 8 && NE <= 128 && (kRows & (kRows - 1)) == 0 && kE == 2 && kRows <= 2048 && CAP < 65536,
                  "shape");
    if (NT >= 1024) __builtin_amdgcn_s_setprio(3);

    __shared__ __attribute__((aligned(16))) unsigned long long s_tab[2][kRows * kE];
    __shared__ PosT s_hits[3 * HB];
    __shared__ __attribute__((aligned(16))) uint32_t s_best[3][HB];
    __shared__ unsigned long long s_stash[3][kStash];
    __shared__ uint32_t s_rowbits[3][kRows / 32];
    // one block per step parity (written by the control wave during step s, read by everyone at the top of step s + 1):
    //   0 flags  1 new arms  2 first creation number  3 family ordinal  4,5 needle offset of their probe
    //   6 age of the quiet probes behind it  7 -   8..15 (this wave's words are in s_slot)   24..27 stash counts
    __shared__ __attribute__((aligned(16))) uint32_t s_pull[2][8];
    __shared__ uint32_t s_slot[2][NWA][8];                             // per (arm wave, layer): slots given << 16 | rank of the first
    __shared__ __attribute__((aligned(16))) uint32_t s_nstash[4];      // (three in use)
    // per step parity, (arm wave, layer): empty slots.  An arm wave writes the block of ITS step's parity in front of the
    // step's barrier, the ranking wave reads the block of the step BEFORE behind it: writer and reader of one block
    // are always a barrier apart (with one block the publication of step s + 1 raced the ranking wave's read of
    // step s's counts -- nothing but time separated them).  K8_SINGLE_FREE (tools/k8_race.sh only) brings that back.
#ifdef K8_SINGLE_FREE
    constexpr uint32_t kFreeBufs = 1;
#else
    constexpr uint32_t kFreeBufs = 2;
#endif
    __shared__ __attribute__((aligned(16))) uint32_t s_free[kFreeBufs][NWA][8];
    __shared__ PosT s_newx[2][kNewMax];                                // the unmatched hits of a probe, by rank
    __shared__ uint32_t s_newch[2][kNewMax];                           // ... and the candidates of the arm born of each
    __shared__ uint16_t s_newh[2][RANGE ? kNewMax : 1u];               // (RANGE) ... and its index among the probe's hits
    __shared__ __attribute__((aligned(16))) uint32_t s_cmd[3][kCmdWords];  // as K7's (word 10: K7_STAGE of the step before)
    __shared__ __attribute__((aligned(16))) uint32_t s_mid[8];         // before the loop: the first batch's staging request
    __shared__ PosT s_cle[CAP], s_crs[CAP];                            // cold fields of an arm, by slot
    __shared__ unsigned long long s_seg[3];                            // g0, chunk start, chunk length (for the records)
    __shared__ uint32_t s_end[4];                                      // planning wave -> ranking wave, at the end of a segment: [1] ran out of the window; [3] start time of a run
    __shared__ unsigned long long s_bcast;
    __shared__ uint32_t s_sink[64];
    __shared__ uint32_t s_never;

    const int tid = threadIdx.x, lane = tid & 63;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane(tid >> 6);
    const bool is_rank = wave == (uint32_t)NWA, is_plan = wave == (uint32_t)(NWA + 1);
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    RecAlloc rec_alloc;
    wg_begin(P);
    K7T_DECL;
    K7U_DECL;
#ifdef ASGART_PROFILE_EXTEND
    unsigned long long k7_sum_a = 0, k7_sum_cnt = 0, k7_sum_n = 0;
#endif

    PosT a_ls[S], a_re[S];
    uint32_t a_thr[S], a_gap[S], a_seq[S], c_h[S];
#pragma unroll
    for (int L = 0; L < S; ++L) {
        a_seq[L] = kNoSeq;
        a_ls[L] = a_re[L] = 0;
        a_thr[L] = a_gap[L] = 0;
        c_h[L] = 0;
    }
    uint32_t livemask = 0;  // wave-uniform: layers in which this wave may hold an arm
    uint32_t gen = 0, par = 0, tri = 0;  // (control wave) they run on from segment to segment
    auto clear_table = [&]() {
        for (uint32_t e = tid; e < (uint32_t)(2 * kRows * kE); e += NT) (&s_tab[0][0])[e] = 0ull;
    };
    clear_table();
    if (tid == 0) s_never = kNever;
    lds_barrier();

    auto tag_of = [&](unsigned long long e) { return (uint32_t)(e >> kTagShift); };
    auto pos_of = [&](unsigned long long e) { return (PosT)(e & kPosMask); };
    char *const tab0 = reinterpret_cast<char *>(&s_tab[0][0]);
    char *const best0 = reinterpret_cast<char *>(&s_best[0][0]);

    for (;;) {
        if (tid == 0) s_bcast = atomicAdd(P.cursor, 1ull);
        if (tid < 4) s_nstash[tid] = 0u;
        if (tid < 16) (&s_pull[0][0])[tid] = 0u;
        for (uint32_t j = tid; j < 3u * kBitWords; j += NT) (&s_rowbits[0][0])[j] = 0u;
        for (uint32_t j = tid; j < kFreeBufs * (uint32_t)(NWA * 8); j += NT) (&s_free[0][0][0])[j] = 64u;
        for (uint32_t j = tid; j < (uint32_t)(2 * NWA * 8); j += NT) (&s_slot[0][0][0])[j] = 0u;
        lds_barrier();
        const unsigned long long seg = uni(s_bcast);
        lds_barrier();
        if (seg >= *P.n_seg_ptr) break;
        // (RANGE) the run; every wave works out the needle offset its records start at for itself
        RangeRun run{};
        uint32_t emit_from_i = 0;
        if constexpr (RANGE) {
            run = P.runs[seg];
            run.g_begin = uni(run.g_begin);
            run.g_stop = uni(run.g_stop);
            run.g_seg0 = uni(run.g_seg0);
            run.emit_from = uni(run.emit_from);
            run.flags = uni(run.flags);
            const int rc = chunk_of_uniform(P.rp.ch, run.g_seg0);
            emit_from_i = (run.flags & kRunNoEmit) ? 0xFFFFFFFFu
                          : (run.emit_from == run.g_seg0 ? 0u : (run.emit_from - P.rp.ch.pbase[rc] + 1u) * (uint32_t)P.rp.step);
        }
        // (RANGE) one arm into dump `which` of this run (0: what the run holds when it stops, 1: what it holds when it reaches
        // its cut), slot handed out by the dump's counter
        auto dump_arm = [&](uint32_t which, bool live, uint32_t seq, PosT ls, PosT le, PosT rs, PosT re, uint32_t thr, uint32_t gap) {
            if constexpr (RANGE) {
                const unsigned long long m = __ballot(live);
                if (!m) return;
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(&P.run_meta[seg * 16ull + which * 8ull], (uint32_t)__popcll(m));
                base = lane_of(base, 0u);
                const uint32_t at = base + (uint32_t)__popcll(m & lt_mask);
                if (live && at < kRunDumpCap) {
                    uint4 *o = reinterpret_cast<uint4 *>(P.run_dump + ((seg * 2ull + which) * (unsigned long long)kRunDumpCap + at) * kDumpWords<PosT>);
                    if constexpr (kWidePos) {
                        o[0] = make_uint4(seq, thr, gap, 0u);
                        o[1] = make_uint4((uint32_t)ls, (uint32_t)((uint64_t)ls >> 32), (uint32_t)le, (uint32_t)((uint64_t)le >> 32));
                        o[2] = make_uint4((uint32_t)rs, (uint32_t)((uint64_t)rs >> 32), (uint32_t)re, (uint32_t)((uint64_t)re >> 32));
                    } else {
                        o[0] = make_uint4(seq, (uint32_t)ls, (uint32_t)le, (uint32_t)rs);
                        o[1] = make_uint4((uint32_t)re, thr, gap, 0u);
                    }
                }
            }
        };
        // the creation number of the r-th new arm of a probe (lists of buffer pbuf)
        auto seq_of = [&](uint32_t seq_base, uint32_t r, uint32_t pbuf) -> uint32_t {
            if constexpr (RANGE) return seq_base | (uint32_t)s_newh[pbuf][r];
            else return seq_base + r;
        };

        auto emit_records = [&](bool emit, PosT ls, PosT le, PosT rs, PosT re, uint32_t seq, uint32_t fam_seq) {
            const unsigned long long em = __ballot(emit);
            if (!em) return;
            const unsigned long long at = rec_slot(rec_alloc, P, em, lane);
            if (emit && at < P.rec_cap) {
                const uint64_t cs = s_seg[1], cl = s_seg[2] & ~(1ull << 63);
                const bool seg_rev = (s_seg[2] >> 63) != 0ull;  // (the orientation of the chunk's pass rides in the top bit)
                const uint64_t ll = (uint64_t)le - (uint64_t)ls;
                SdRec r;
                r.g_start = (uint32_t)s_seg[0];
                r.fam_seq = fam_seq;
                r.create_seq = seq;
                r.pad = RANGE ? (uint32_t)seg + 1u : 0u;  // (RANGE: the run, for the host's renumbering of its families)
                r.sd.left = seg_rev ? cs + cl - (uint64_t)ls - ll : (uint64_t)ls + cs;  // src/bin/asgart.rs:229-237
                r.sd.right = rs;
                r.sd.left_length = ll;
                r.sd.right_length = (uint64_t)re - (uint64_t)rs;
                P.recs[at] = r;
            }
        };
        // index the hits of one probe under generation tag g10 in the table at byte offset tabo, winners at byte offset
        // besto, stash / occupancy bits bb; its rows are at s_hits[off..] or, when gsrc is set (the probe opens a batch
        // whose rows are on their way to LDS in this very step), at P.hits[gbase..]; the TOP threads do it
        auto insert_hits = [&](uint32_t cnt, uint32_t off, uint32_t tabo, uint32_t besto, uint32_t bb, uint32_t g10, uint32_t bsh,
                               bool gsrc, unsigned long long gbase) {
            const uint32_t me = tid < NWA * 64 ? (uint32_t)(NWA * 64 - 1 - tid) : (uint32_t)tid;
            for (uint32_t h = me; h < cnt; h += NT) {
                const PosT x = gsrc ? P.hits[gbase + h] : s_hits[off + h];
                *reinterpret_cast<uint32_t *>(best0 + besto + 4u * h) = kNone;
                unsigned long long e = ((unsigned long long)(g10 | h) << kTagShift) | ((unsigned long long)x & kPosMask);
                const uint32_t ri = ((uint32_t)((uint64_t)x >> bsh)) & (uint32_t)(kRows - 1);
                unsigned long long *row = reinterpret_cast<unsigned long long *>(tab0 + tabo + ri * (uint32_t)(kE * 8));
                atomicOr(&s_rowbits[bb][ri >> 5], 1u << (ri & 31u));
                bool placed = false;
#pragma unroll
                for (int j = 0; j < kE; ++j) {
                    if (!placed) {
                        const unsigned long long old = atomicExch(&row[j], e);
                        if (tag_of(old) - g10 >= 1024u) placed = true;  // displaced a stale entry: done
                        else e = old;                                   // a hit of this probe: it moves on
                    }
                }
                if (!placed) {
                    const uint32_t at = atomicAdd(&s_nstash[bb], 1u);
                    if (at < kStash) s_stash[bb][at] = e;
                }
            }
        };
        auto indexes = [&](uint32_t cnt) {
            const uint32_t first = wave >= (uint32_t)NWA ? wave : (uint32_t)(NWA - 1) - wave;  // in units of 64 hits
            return first < (cnt + 63u) / 64u;
        };
        constexpr int kPf = (HB + NT - 1) / NT;
        PosT pf_x[kPf];
        auto fetch_rows = [&](unsigned long long base, uint32_t tot) {
#pragma unroll
            for (int j = 0; j < kPf; ++j) {
                const uint32_t r = (uint32_t)tid + (uint32_t)(j * NT);
                pf_x[j] = r < tot ? P.hits[base + r] : (PosT)0;
            }
        };
        auto store_rows = [&](uint32_t tot, uint32_t buf) {
#pragma unroll
            for (int j = 0; j < kPf; ++j) {
                const uint32_t r = (uint32_t)tid + (uint32_t)(j * NT);
                if (r < tot) s_hits[buf * (uint32_t)HB + r] = pf_x[j];
            }
        };
        // The offers of ONE window per lane to the hits of a probe (the lanes with `who`): the table walk of K6 / K7 --
        // two rows, then the rows of the window whose occupancy bit is set, then the stash; wider windows and every
        // window when the stash overflowed go through the wave-cooperative scan.  Returns the candidate word.
        struct Cur {
            uint32_t k, g10, bsh, tabo, besto, bb, cnt, off;
        };
        auto offer_window = [&](const Cur &q, PosT lo, WinT w, uint32_t key, bool who, uint32_t ns) -> uint32_t {
            const uint32_t g10 = q.g10, bsh = q.bsh;
            char *const tab = tab0 + q.tabo;
            char *const best = best0 + q.besto;
            const bool povf = ns > kStash;
            const WinT w_loop = (WinT)(kRowsWalk - 1u) << bsh;
            const bool narrow = who && w <= w_loop;
            const WinT w_eff = narrow ? w : (WinT)0;  // (an empty window accepts nothing)
            uint32_t ch = 0, nc = 0;
            uint32_t *const sink = &s_sink[lane];
            auto offer = [&](unsigned long long e, WinT wl) {
                const uint32_t d = tag_of(e) - g10;
                const WinT t = d < 1024u ? (WinT)(PosT)(pos_of(e) - lo) : ~(WinT)0;
                const bool ok = t < wl;
                atomicMin(ok ? reinterpret_cast<uint32_t *>(best + 4u * (d & 1023u)) : sink, key);
                ch = ok ? ((ch << 10) | d) : ch;
                nc += ok ? 1u : 0u;
            };
            const uint32_t b0 = (uint32_t)((uint64_t)lo >> bsh);
            const uint32_t n_rows = narrow ? (uint32_t)((((uint64_t)lo & ((1ull << bsh) - 1ull)) + (uint64_t)w - 1ull) >> bsh) + 1u : 0u;
            {   // the two rows of a narrow window: all reads in flight together; ONE atomic for their four entries
                const ulonglong2 *r0 = reinterpret_cast<const ulonglong2 *>(tab + (b0 & (uint32_t)(kRows - 1)) * (uint32_t)(kE * 8));
                const ulonglong2 *r1 = reinterpret_cast<const ulonglong2 *>(tab + ((b0 + 1u) & (uint32_t)(kRows - 1)) * (uint32_t)(kE * 8));
                const ulonglong2 e0 = r0[0], e2 = r1[0];
                const unsigned long long ee[4] = {e0.x, e0.y, e2.x, e2.y};
                uint32_t first = 0;
#pragma unroll
                for (int j = 3; j >= 0; --j) {
                    const uint32_t d = tag_of(ee[j]) - g10;
                    const WinT t = d < 1024u ? (WinT)(PosT)(pos_of(ee[j]) - lo) : ~(WinT)0;
                    first = t < w_eff ? d : first;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t d = tag_of(ee[j]) - g10;
                    const WinT t = d < 1024u ? (WinT)(PosT)(pos_of(ee[j]) - lo) : ~(WinT)0;
                    const bool ok = t < w_eff;
                    ch = ok ? ((ch << 10) | d) : ch;
                    nc += ok ? 1u : 0u;
                }
                atomicMin(nc ? reinterpret_cast<uint32_t *>(best + 4u * (first & 1023u)) : sink, key);
                if (K7_RARE(__ballot(nc > 1u) != 0ull)) {
#pragma unroll
                    for (int j = 1; j < 4; ++j) {
                        const uint32_t d = tag_of(ee[j]) - g10;
                        const WinT t = d < 1024u ? (WinT)(PosT)(pos_of(ee[j]) - lo) : ~(WinT)0;
                        atomicMin(t < w_eff ? reinterpret_cast<uint32_t *>(best + 4u * (d & 1023u)) : sink, key);
                    }
                }
            }
            if (__ballot(n_rows > 2u) != 0ull) {
                const uint32_t len = n_rows > 2u ? n_rows - 2u : 0u;                 // <= kRowsWalk - 1 < 64
                const uint32_t s0 = (b0 + 2u) & (uint32_t)(kRows - 1);                // first of them (table row)
                const uint32_t *const bits = &s_rowbits[q.bb][0];
                const uint32_t w0 = s0 >> 5, sh = s0 & 31u;
                const uint32_t v0 = bits[w0], v1 = bits[(w0 + 1u) & (kBitWords - 1u)];
                uint32_t v2 = 0u;
                if (K7_RARE(__ballot(len > 33u) != 0ull)) v2 = bits[(w0 + 2u) & (kBitWords - 1u)];
                unsigned long long m = ((((unsigned long long)v1 << 32) | v0) >> sh) | (sh ? (unsigned long long)v2 << (64u - sh) : 0ull);
                m &= (1ull << len) - 1ull;
                while (__ballot(m != 0ull) != 0ull) {
                    const bool act = m != 0ull;
                    const uint32_t r = act ? (uint32_t)(__ffsll((long long)m) - 1) : 0u;
                    m &= m - 1ull;
                    const ulonglong2 *rr = reinterpret_cast<const ulonglong2 *>(tab + ((s0 + r) & (uint32_t)(kRows - 1)) * (uint32_t)(kE * 8));
                    const WinT wl = act ? w_eff : (WinT)0;
                    const ulonglong2 f0 = rr[0];
                    offer(f0.x, wl);
                    offer(f0.y, wl);
                }
            }
            if (K7_RARE(ns != 0u))
                for (uint32_t s = 0; s < min(ns, kStash); ++s) offer(s_stash[q.bb][s], w_eff);
            ch = nc > 3u ? kCoop : (ch & 0x3FFFFFFFu) | (nc << 30);
            // windows too wide for the table walk -- and every window when the stash overflowed
            unsigned long long sm = __ballot(who && (!narrow || povf));
            if (K7_RARE(sm != 0ull)) {
                if (who && (!narrow || povf)) ch = kCoop;
                while (sm) {
                    const uint32_t l = (uint32_t)(__ffsll((long long)sm) - 1);
                    sm &= sm - 1ull;
                    PosT lo_u;
                    WinT w_u;
                    if constexpr (kWidePos) {
                        lo_u = (PosT)lane_of((unsigned long long)lo, l);
                        w_u = (WinT)lane_of((unsigned long long)w, l);
                    } else {
                        lo_u = (PosT)lane_of((uint32_t)lo, l);
                        w_u = (WinT)lane_of((uint32_t)w, l);
                    }
                    const uint32_t key_u = lane_of(key, l);
                    for (uint32_t h0 = 0; h0 < q.cnt; h0 += 64u) {
                        const uint32_t h = h0 + (uint32_t)lane;
                        if (h < q.cnt && (WinT)(PosT)(s_hits[q.off + h] - lo_u) < w_u)
                            atomicMin(reinterpret_cast<uint32_t *>(best + 4u * h), key_u);
                    }
                }
            }
            return ch;
        };

        // The first offers of the new arms beyond the 64 the ranking wave handled (K8_BIG): round j >= 1 (arms 64 j ...) is
        // taken by wave (j - 1) mod NW; q describes the probe they offer to (the previous step's), lists in buffer pbuf.
        auto big_rounds = [&](const Cur &q, uint32_t n_pull, uint32_t seq_base, uint32_t thr0, uint32_t pbuf) {
            const uint32_t ns_q = uni(s_nstash[q.bb]);
            for (uint32_t j = 1u + wave; j * 64u < n_pull; j += (uint32_t)NW) {
                const uint32_t r = j * 64u + (uint32_t)lane;
                const PosT x = s_newx[pbuf][min(r, n_pull - 1u)];
                const uint32_t ch = offer_window(q, (PosT)(x + 1u), (WinT)thr0 + (WinT)(q.k - 1u), seq_of(seq_base, min(r, n_pull - 1u), pbuf), r < n_pull, ns_q);
                if (r < n_pull) s_newch[pbuf][r] = ch;
            }
        };

        bool overflow = false;
        if (!is_rank && !is_plan) {
            // =====================================================================================================
            // ARM WAVES
            // =====================================================================================================
            lds_barrier();  // (1) the control wave has published the first batch's staging request and two commands
            {
                const uint4 m1 = *reinterpret_cast<const uint4 *>(&s_mid[4]);
                if (uni(s_mid[0]) & K7_STAGE) {
                    fetch_rows(((unsigned long long)uni(m1.y) << 32) | uni(m1.x), uni(m1.z));
                    store_rows(uni(m1.z), uni(m1.w));
                }
            }
            lds_barrier();  // (2) the rows are in
            {   // the first probe's hits (every later probe is indexed during the step before its own)
                const uint32_t w0 = s_cmd[0][lane & 31];
                auto C0 = [&](int j) { return (uint32_t)__builtin_amdgcn_readlane((int)w0, j); };
                if ((C0(0) & K7_CUR) && indexes(C0(1))) insert_hits(C0(1), C0(2), C0(3), C0(4), C0(9), C0(5), C0(19), false, 0ull);
            }
            lds_barrier();  // (3)
            bool pub_prev = false;              // this wave published its free counts in the step before
            uint32_t last_i = 0;                // (RANGE) needle offset of the last probe that has been resolved
            uint32_t pv_off = 0, pv_besto = 0;  // the previous probe's rows and winners (as in the last command)
            uint32_t pv_tabo = 0, pv_g10 = 0;   // ... its hit table and generation tag
            // (lanes 0-7: the pull block; 8-15: this wave's slots; 16-19: the stash counts -- per step parity)
            const uint32_t *const pb_ptr0 = lane < 8 ? &s_pull[0][lane] : (lane < 16 ? &s_slot[0][wave][lane - 8] : (lane < 20 ? &s_nstash[lane - 16] : &s_pull[0][0]));
            const uint32_t *const pb_ptr1 = lane < 8 ? &s_pull[1][lane] : (lane < 16 ? &s_slot[1][wave][lane - 8] : (lane < 20 ? &s_nstash[lane - 16] : &s_pull[1][0]));
            for (uint32_t sc = 0, sp = 0;; sc = sc == 2u ? 0u : sc + 1u, sp ^= 1u) {
                K7T_MARK();
                K7T_STEP();
                K7U_MARK();
                const uint32_t sn = sc == 2u ? 0u : sc + 1u;  // the next step's command
                const uint32_t cw = s_cmd[lane < 32 ? sc : sn][lane & 31];
                const uint32_t pw = *(sp ? pb_ptr0 : pb_ptr1);  // (step s reads the block written during step s - 1)
                auto C = [&](int j) { return (uint32_t)__builtin_amdgcn_readlane((int)cw, j); };
                auto N = [&](int j) { return (uint32_t)__builtin_amdgcn_readlane((int)cw, 32 + j); };
                auto PB = [&](uint32_t j) { return (uint32_t)__builtin_amdgcn_readlane((int)pw, (int)j); };
                // (winners AND positions of an arm's candidates, all requested at once: with one barrier per step every
                // dependent round trip of an arm wave is on the step's critical path)
                uint32_t cb0[3] = {0, 0, 0};
                PosT xb0[3] = {0, 0, 0};
                auto read_candidates = [&](int L, uint32_t (&cb)[3], PosT (&xb)[3]) {
                    const uint32_t ch = c_h[L];
                    const uint32_t nc = (a_seq[L] == kNoSeq || ch == kCoop) ? 0u : ch >> 30;
#pragma unroll
                    for (uint32_t j = 0; j < 3; ++j) {
                        const uint32_t hj = (ch >> (10u * j)) & 1023u;
                        cb[j] = *(j < nc ? reinterpret_cast<const uint32_t *>(best0 + pv_besto + 4u * hj) : &s_never);
                        xb[j] = s_hits[pv_off + (j < nc ? hj : 0u)];
                    }
                };
                if (NT >= 1024) {
                    if (livemask) __builtin_amdgcn_s_setprio(3);
                    else __builtin_amdgcn_s_setprio(0);
                }
                if (livemask & 1u) read_candidates(0, cb0, xb0);
                const uint32_t flags = C(0);
                if (K7_RARE((flags & K7_GIVEUP) || (PB(0) & K8_OVF))) {
                    overflow = true;
                    break;
                }
                const bool has_prev = (flags & K7_PREV) != 0u, has_cur = (flags & K7_CUR) != 0u, more = !(flags & K7_LAST);
                // (RANGE) an arm that dies belongs to the range of the last probe resolved before it is found dead -- in the step
                // that resolves the probe or, behind a generation wrap, in the step that brings the quiet probes' age: the same
                // probe in every run, wherever its wraps fall
                if constexpr (RANGE) last_i = has_prev ? C(12) : last_i;
                bool stale0 = false;  // the winners requested above predate the offers below
                if (K7_RARE(PB(0) & K8_BIG)) {  // (before anything touches the previous probe's table or its stash count)
                    big_rounds(Cur{C(16), pv_g10, C(19), pv_tabo, C(15), C(23), C(11), C(14)}, PB(1), PB(2), C(22), sp ^ 1u);
                    lds_barrier();
                    stale0 = true;
                }
                if (K7_RARE(flags & K8_CLEARNOW)) clear_table();
                // the rows that are staged during this step: requested now, written to LDS in front of the barrier
                const uint32_t npre = more ? N(10) : 0u;
                if (K7_RARE(npre & K7_STAGE)) fetch_rows(((unsigned long long)N(25) << 32) | N(24), N(26));
                // the hits of the NEXT step's probe: indexed by the top arm waves
                if (more) {
                    const uint32_t nflags = N(0), ncnt = N(1);
                    if (K7_RARE((nflags & K7_CUR) && indexes(ncnt))) {
                        const bool gsrc = (nflags & K7_LATE) != 0u;
                        const unsigned long long gbase = (((unsigned long long)N(25) << 32) | N(24)) + (N(2) - N(27) * (uint32_t)HB);
                        insert_hits(ncnt, N(2), N(3), N(4), N(9), N(5), N(19), gsrc, gbase);
                    }
                }
                K7U_LAP(0);
                const bool had_live = livemask != 0u;
                const uint32_t k = C(16), step = C(17), G = C(18), pend = C(8);
                const uint64_t M = ((uint64_t)C(21) << 32) | C(20);
                // ---- the arms born of the probe before the previous one: pulled by rank ---------------------------------
                const uint32_t n_pull = PB(1);
                bool received = false;
                if (K7_RARE(n_pull != 0u)) {
                    const uint32_t pbuf = sp ^ 1u;  // (the lists of the step before)
                    const uint32_t seq_base = PB(2), fam_b = PB(3), thr0 = C(22);
                    PosT b_i;
                    if constexpr (kWidePos) b_i = (PosT)(((uint64_t)PB(5) << 32) | PB(4));
                    else b_i = (PosT)PB(4);
                    const uint64_t g_new = (uint64_t)step + PB(6);  // aged by its own probe, then by the quiet ones
                    const uint32_t gap_new = g_new > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)g_new;
                    const bool stillborn = (PB(0) & K8_STILL) != 0u;  // (only when min_duplication_length <= k are they reported)
#pragma unroll
                    for (int L = 0; L < S; ++L) {
                        const uint32_t packed = PB(8u + (uint32_t)L);
                        const uint32_t base_r = packed & 0xFFFFu, given = packed >> 16;
                        if (base_r >= n_pull) break;  // (ranks ascend with the layer: nothing above either)
                        if (!given) continue;         // (this layer of this wave is full; the one above may not be)
                        received = true;
                        const bool is_free = a_seq[L] == kNoSeq;
                        const unsigned long long fmask = __ballot(is_free);
                        const uint32_t idx = (uint32_t)__popcll(fmask & lt_mask);
                        const uint32_t r = base_r + idx;
                        bool take = is_free && idx < given && r < n_pull;
                        const PosT x = s_newx[pbuf][take ? r : 0u];
                        const uint32_t chn = s_newch[pbuf][take ? r : 0u];
                        const uint32_t seq_n = seq_of(seq_base, take ? r : 0u, pbuf);
                        if (K7_RARE(stillborn)) {
                            const bool report = take && (uint64_t)k >= M && (!RANGE || (uint32_t)b_i >= emit_from_i);
                            if (__ballot(report)) emit_records(report, b_i, (PosT)(b_i + k), x, (PosT)(x + k), seq_n, fam_b);
                            take = false;
                        }
                        a_ls[L] = take ? b_i : a_ls[L];
                        if (take) {
                            s_cle[L * (NWA * 64) + tid] = (PosT)(b_i + k);
                            s_crs[L * (NWA * 64) + tid] = x;
                        }
                        a_re[L] = take ? (PosT)(x + k) : a_re[L];
                        a_gap[L] = take ? gap_new : a_gap[L];
                        a_thr[L] = take ? thr0 : a_thr[L];
                        a_seq[L] = take ? seq_n : a_seq[L];
                        c_h[L] = take ? chn : c_h[L];
                        if (__ballot(take)) livemask |= 1u << L;
                    }
                }
                // ---- resolve the previous probe, age -------------------------------------------------------------------
                if (livemask) {
                    const uint32_t p_i = C(12);  // low word of the previous probe's needle offset
#pragma unroll
                    for (int L = 0; L < S; ++L) {
                        if (!(livemask >> L)) break;
                        if (!(livemask & (1u << L))) continue;
                        const bool was_free = a_seq[L] == kNoSeq;
                        bool won = false;
                        PosT xw = 0;
                        uint32_t cb[3];
                        PosT xb[3];
                        if (L == 0 && !received && !stale0) {
#pragma unroll
                            for (int j = 0; j < 3; ++j) {
                                cb[j] = cb0[j];
                                xb[j] = xb0[j];
                            }
                        } else {
                            read_candidates(L, cb, xb);  // (the arms pulled in this step have their candidates only now)
                        }
                        if (K7_USUAL(has_prev)) {
                            // the last hit (SA order) this arm won, if any: src/automaton.rs:133-150 apply in hit order
                            const uint32_t ch = c_h[L];
                            const bool coop = !was_free && ch == kCoop;
                            uint32_t hw = 0;  // 1 + that hit
#pragma unroll
                            for (uint32_t j = 0; j < 3; ++j) {
                                const uint32_t hj = (ch >> (10u * j)) & 1023u;
                                const bool mine = cb[j] == a_seq[L] && hj + 1u > hw;
                                hw = mine ? hj + 1u : hw;
                                xw = mine ? xb[j] : xw;
                            }
                            hw = (was_free || coop) ? 0u : hw;
                            unsigned long long sm = __ballot(coop);
                            if (K7_RARE(sm != 0ull)) {  // more than three candidates / wide window: resolved cooperatively
                                const uint32_t p_cnt = C(11), p_off = pv_off;
                                if ((uint32_t)__popcll(sm) * 8u >= p_cnt) {  // (many: one pass over the hits, extend_fast_dev.hpp)
                                    for (uint32_t h0 = 0; h0 < p_cnt; h0 += 64u) {
                                        const uint32_t h = h0 + (uint32_t)lane;
                                        const PosT xh = h < p_cnt ? s_hits[p_off + h] : (PosT)0;
                                        const uint32_t bh = h < p_cnt ? *reinterpret_cast<const uint32_t *>(best0 + pv_besto + 4u * h) : kNone;
                                        const uint32_t nh = min(64u, p_cnt - h0);
                                        for (uint32_t j = 0; j < nh; ++j) {
                                            const uint32_t b = lane_of(bh, j);
                                            PosT x;
                                            if constexpr (kWidePos) x = (PosT)lane_of((unsigned long long)xh, j);
                                            else x = (PosT)lane_of((uint32_t)xh, j);
                                            const bool mine = coop && a_seq[L] == b;
                                            hw = mine ? h0 + j + 1u : hw;
                                            xw = mine ? x : xw;
                                        }
                                    }
                                    sm = 0ull;
                                }
                                while (sm) {
                                    const uint32_t l = (uint32_t)(__ffsll((long long)sm) - 1);
                                    sm &= sm - 1ull;
                                    const PosT lo = (PosT)(a_re[L] - k + 1u);
                                    const WinT w = (WinT)a_thr[L] + (WinT)(k - 1u);
                                    PosT lo_u;
                                    WinT w_u;
                                    if constexpr (kWidePos) {
                                        lo_u = (PosT)lane_of((unsigned long long)lo, l);
                                        w_u = (WinT)lane_of((unsigned long long)w, l);
                                    } else {
                                        lo_u = (PosT)lane_of((uint32_t)lo, l);
                                        w_u = (WinT)lane_of((uint32_t)w, l);
                                    }
                                    const uint32_t key = lane_of(a_seq[L], l);
                                    uint32_t hmax = kNone;
                                    PosT x_u = 0;
                                    for (uint32_t h0 = 0; h0 < p_cnt; h0 += 64u) {
                                        const uint32_t h = h0 + (uint32_t)lane;
                                        PosT x = 0;
                                        bool ok = false;
                                        if (h < p_cnt) {
                                            x = s_hits[p_off + h];
                                            ok = (WinT)(PosT)(x - lo_u) < w_u &&
                                                 *reinterpret_cast<const uint32_t *>(best0 + pv_besto + 4u * h) == key;
                                        }
                                        const unsigned long long bm = __ballot(ok);
                                        if (bm) {
                                            const uint32_t top = 63u - (uint32_t)__clzll((long long)bm);
                                            hmax = h0 + top;
                                            if constexpr (kWidePos) x_u = (PosT)lane_of((unsigned long long)x, top);
                                            else x_u = (PosT)lane_of((uint32_t)x, top);
                                        }
                                    }
                                    if ((uint32_t)lane == l && hmax != kNone) {
                                        hw = hmax + 1u;
                                        xw = x_u;
                                    }
                                }
                            }
                            won = hw != 0u;
                        }
                        // ExtendArm (src/automaton.rs:133-150) or one more step of age (:166-171), then the quiet probes
                        // between the previous probe and this one
                        uint32_t thr_new;
                        if constexpr (kWidePos) {
                            const uint64_t p_i64 = ((uint64_t)C(13) << 32) | C(12);
                            thr_new = arm_threshold((uint64_t)(p_i64 + k) - (uint64_t)a_ls[L], G);
                        } else {
                            thr_new = max(G, ((p_i + k) - (uint32_t)a_ls[L]) / 10u);
                        }
                        const uint64_t sum_g = (uint64_t)(won ? 0u : a_gap[L]) + (has_prev && !won ? step : 0u) + pend;
                        const uint32_t aged = sum_g > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)sum_g;
                        a_re[L] = won ? (PosT)(xw + k) : a_re[L];
                        PosT le_new;
                        if constexpr (kWidePos) le_new = (PosT)((((uint64_t)C(13) << 32) | C(12)) + k);
                        else le_new = (PosT)(p_i + k);
                        if (won) s_cle[L * (NWA * 64) + tid] = le_new;
                        a_thr[L] = won ? thr_new : a_thr[L];
                        a_gap[L] = aged;
                        const bool dead = !was_free && aged >= G;  // never matches again
                        if (K7_RARE(__ballot(dead) != 0ull)) {
                            const PosT rs_L = s_crs[L * (NWA * 64) + tid];
                            // (RANGE: deaths found while a probe in front of the cut is resolved belong to the range before)
                            const bool report = dead && (uint64_t)(a_re[L] - rs_L) >= M && (!RANGE || last_i >= emit_from_i);
                            if (__ballot(report) != 0ull)
                                emit_records(report, a_ls[L], s_cle[L * (NWA * 64) + tid], rs_L, a_re[L], a_seq[L], PB(3));
                        }
                        a_seq[L] = dead ? kNoSeq : a_seq[L];
                    }
                }
                K7U_LAP(1);
                if constexpr (RANGE) {
                    // The step whose probe is this run's cut: the arms stand as they stand in front of the cut -- everything before
                    // it resolved and aged (behind a generation wrap: by this very step), the arms born of the probe before it not
                    // yet pulled (the ranking wave writes those, below) -- which is what a run that STOPS at the cut holds at its
                    // end.  Written out for the comparison with the range in front of the cut.
                    if (K7_RARE(has_cur && emit_from_i != 0u && emit_from_i != 0xFFFFFFFFu && C(6) == emit_from_i)) {
#pragma unroll
                        for (int L = 0; L < S; ++L)
                            dump_arm(1u, a_seq[L] != kNoSeq, a_seq[L], a_ls[L], s_cle[L * (NWA * 64) + tid], s_crs[L * (NWA * 64) + tid], a_re[L],
                                     a_thr[L], a_gap[L]);
                    }
                }
                // ---- every arm offers to this step's probe --------------------------------------------------------------
                if (has_cur && livemask) {
                    const Cur q{k, C(5), C(19), C(3), C(4), C(9), C(1), C(2)};
                    const uint32_t ns_b = PB(16u + C(9));  // the stash count of the step's probe (final: indexed a step ago)
#pragma unroll
                    for (int L = 0; L < S; ++L) {
                        if (!(livemask >> L)) break;
                        if (livemask & (1u << L))
                            c_h[L] = offer_window(q, (PosT)(a_re[L] - k + 1u), (WinT)a_thr[L] + (WinT)(k - 1u), a_seq[L], a_seq[L] != kNoSeq, ns_b);
                    }
                }
                K7U_LAP(2);
                // free counts, as the ranking wave will read them in the next step.  A wave that goes idle publishes once
                // more, into the other block: both blocks of an idle wave then say "all empty" and stay that way.
                const bool pub = had_live || received;
                if (pub || pub_prev) {
                    uint32_t nfv[8] = {64u, 64u, 64u, 64u, 64u, 64u, 64u, 64u};
#pragma unroll
                    for (int L = 0; L < S; ++L) {
                        if (!(livemask >> L)) break;
                        if (livemask & (1u << L)) {
                            const uint32_t nf = (uint32_t)__popcll(__ballot(a_seq[L] == kNoSeq));
                            nfv[L] = nf;
                            if (nf == 64u) livemask &= ~(1u << L);
                        }
                    }
                    uint32_t *const fr = &s_free[sp & (kFreeBufs - 1u)][wave][0];
                    *reinterpret_cast<uint4 *>(fr) = make_uint4(nfv[0], nfv[1], nfv[2], nfv[3]);
                    if constexpr (S > 4) *reinterpret_cast<uint4 *>(fr + 4) = make_uint4(nfv[4], nfv[5], nfv[6], nfv[7]);
                }
                pub_prev = pub;
                if (K7_RARE(npre & K7_STAGE)) store_rows(N(26), N(27));
                K7U_LAP(3);
                K7T_LAP(0);
                lds_barrier();  // ---- the barrier of the step ---------------------------------------------------------
                K7T_LAP(1);
                pv_off = C(2);
                pv_besto = C(4);
                pv_tabo = C(3);
                pv_g10 = C(5);
                if (K7_RARE(!more)) break;
            }
            if constexpr (RANGE) {
                // the arms this run leaves alive, in any order (the comparison goes by creation number)
                if (!overflow) {
#pragma unroll
                    for (int L = 0; L < S; ++L)
                        dump_arm(0u, a_seq[L] != kNoSeq, a_seq[L], a_ls[L], s_cle[L * (NWA * 64) + tid], s_crs[L * (NWA * 64) + tid], a_re[L], a_thr[L],
                                 a_gap[L]);
                }
            }
            if (wave == 0u) K7T_FLUSH(1);
            if (wave == 0u) K7U_FLUSH();
#ifdef K8_TRACE_WAVE
            if (wave == (uint32_t)(K8_TRACE_WAVE)) K7T_FLUSH(2);  // (diagnostic build: -DK8_TRACE_WAVE=n traces arm wave n instead)
#endif
        } else if (is_rank) {
            // =====================================================================================================
            // RANKING WAVE: flushes, unmatched hits -> new arms (ranks, slots, first offers), overflow
            // =====================================================================================================
            const RunParams &rp = P.rp;
            const uint32_t k = (uint32_t)rp.k, step = (uint32_t)rp.step, G = rp.G;
            const uint32_t thr0 = arm_threshold(k, G);
            const uint32_t cap_eff = min((uint32_t)CAP, P.cap_limit);
            const uint32_t g0 = RANGE ? run.g_seg0 : P.seg_list[seg];
            uint32_t i_seg0 = 0;  // (RANGE) needle offset of the segment's first probe
            if constexpr (RANGE) i_seg0 = (g0 - rp.ch.pbase[chunk_of_uniform(rp.ch, g0)] + 1u) * step;
#ifdef ASGART_PROFILE_EXTEND
            const unsigned long long k7_seg0 = __builtin_amdgcn_s_memtime();
#endif
            uint32_t fam_seq = 0, next_seq = 0;
            bool fam_open = false;
            lds_barrier();  // (1)
            {
                const uint4 m1 = *reinterpret_cast<const uint4 *>(&s_mid[4]);
                if (uni(s_mid[0]) & K7_STAGE) {
                    fetch_rows(((unsigned long long)uni(m1.y) << 32) | uni(m1.x), uni(m1.z));
                    store_rows(uni(m1.z), uni(m1.w));
                }
            }
            lds_barrier();  // (2)
            {
                const uint32_t w0 = s_cmd[0][lane & 31];
                auto C0 = [&](int j) { return (uint32_t)__builtin_amdgcn_readlane((int)w0, j); };
                if ((C0(0) & K7_CUR) && indexes(C0(1))) insert_hits(C0(1), C0(2), C0(3), C0(4), C0(9), C0(5), C0(19), false, 0ull);
            }
            lds_barrier();  // (3)
            // what the pull that is pending took of each (wave, layer) entry (entry j = lane, and lane + 64)
            uint32_t took1 = 0, took2 = 0, pending_alive = 0;
            uint32_t pv_tabo = 0, pv_g10 = 0, big_n = 0, big_seq = 0;  // (a K8_BIG block of the step before: its size and first number)
            uint32_t sp = 0;  // step parity (after the loop: the last step's)
            for (uint32_t sc = 0;; sc = sc == 2u ? 0u : sc + 1u, sp ^= 1u) {
                const uint32_t sn = sc == 2u ? 0u : sc + 1u;
                const uint32_t cw = s_cmd[lane < 32 ? sc : sn][lane & 31];
                auto C = [&](int j) { return (uint32_t)__builtin_amdgcn_readlane((int)cw, j); };
                auto N = [&](int j) { return (uint32_t)__builtin_amdgcn_readlane((int)cw, 32 + j); };
                const uint32_t flags = C(0);
                if ((flags & K7_GIVEUP) || overflow) {  // (its own decision of the step before, or the planning wave's)
                    overflow = true;
                    break;
                }
                const bool have_prev = (flags & K7_PREV) != 0u, has_cur = (flags & K7_CUR) != 0u;
                const bool more = !(flags & K7_LAST);
                const uint32_t nflags = more ? N(0) : 0u;
                const uint32_t prev_cnt = C(11), prev_off = C(14), prev_bb = C(23);
                const uint32_t bsh = C(19);
                K7T_MARK();
                K7T_STEP();
                if (big_n) {
                    big_rounds(Cur{k, pv_g10, bsh, pv_tabo, C(15), C(23), C(11), C(14)}, big_n, big_seq, thr0, sp ^ 1u);
                    lds_barrier();
                    big_n = 0;
                }
                if (flags & K8_CLEARNOW) clear_table();
                const uint32_t npre = more ? N(10) : 0u;
                const unsigned long long st_base = ((unsigned long long)N(25) << 32) | N(24);
                const uint32_t st_tot = N(26), st_buf = N(27);
                if (npre & K7_STAGE) fetch_rows(st_base, st_tot);
                if (more && (nflags & K7_CUR) && indexes(N(1)))
                    insert_hits(N(1), N(2), N(3), N(4), N(9), N(5), bsh, (nflags & K7_LATE) != 0u, st_base + (N(2) - st_buf * (uint32_t)HB));
                {   // ---- slots: what the arm waves published, minus what the pending pull takes ----------------------
                    // (the block the arm waves wrote in front of the barrier this step began with; test hook k8_delay:
                    // the read is held back that many cycles -- results must not depend on when it happens)
                    if (K7_RARE(P.k8_delay != 0u)) {
                        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
                        while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)P.k8_delay) __builtin_amdgcn_s_sleep(2);
                    }
                    const uint32_t (*const fr)[8] = s_free[(sp ^ 1u) & (kFreeBufs - 1u)];
                    const uint32_t fv = lane < NE ? fr[lane % NWA][lane / NWA] : 0u;
                    const uint32_t fv2 = NE > 64 && lane + 64 < NE ? fr[(lane + 64) % NWA][(lane + 64) / NWA] : 0u;
                    uint32_t n_new = 0, seq_base = 0, pflags = 0;
                    uint32_t av1 = fv - min(fv, took1), av2 = fv2 - min(fv2, took2);
                    uint32_t base1 = 0, base2 = 0;
                    if (have_prev) {
                        const uint32_t h_l = min((uint32_t)lane, prev_cnt - 1u);
                        const uint32_t bv0 = s_best[prev_bb][h_l];
                        const PosT x0 = s_hits[prev_off + h_l];
                        uint32_t bv1 = kNone, bv2 = kNone;
                        PosT x1 = 0, x2 = 0;
                        if (prev_cnt > 64u) {
                            const uint32_t h1 = min(64u + (uint32_t)lane, prev_cnt - 1u), h2 = min(128u + (uint32_t)lane, prev_cnt - 1u);
                            bv1 = s_best[prev_bb][h1];
                            bv2 = s_best[prev_bb][h2];
                            x1 = s_hits[prev_off + h1];
                            x2 = s_hits[prev_off + h2];
                        }
                        // (one scan for both halves of the entries: two 16-bit fields, each total < 2^16)
                        const uint32_t packed = wave_incl_scan(av1 | (av2 << 16));
                        const uint32_t fpk = wave_incl_scan(fv | (fv2 << 16));
                        const uint32_t tot_p = lane_of(packed, 63u), tot_f = lane_of(fpk, 63u);
                        uint32_t total_av = tot_p & 0xFFFFu;
                        base1 = (packed & 0xFFFFu) - av1;
                        if constexpr (NE > 64) {
                            base2 = (packed >> 16) + total_av - av2;
                            total_av += tot_p >> 16;
                        }
                        const uint32_t total_free = (tot_f & 0xFFFFu) + (NE > 64 ? tot_f >> 16 : 0u);
                        // live arms when the previous probe is processed: the ones in their slots and the ones still to be pulled
                        const uint32_t A0 = (uint32_t)CAP - total_free + pending_alive;
#ifdef ASGART_PROFILE_EXTEND
                        k7_sum_a += A0;
                        k7_sum_cnt += prev_cnt;
                        ++k7_sum_n;
#endif
                        if (fam_open && A0 == 0) {  // the flush of src/automaton.rs:182-200
                            if (!RANGE || C(12) >= emit_from_i) ++fam_seq;  // (RANGE: the flushes from the cut on)
                            next_seq = 0;
                            fam_open = false;
                        }
                        // unmatched hits, in hit order (= creation order, src/automaton.rs:151-164) -> s_newx[rank]
                        auto rank_group = [&](uint32_t h0, uint32_t bv, PosT x) {
                            const bool in = h0 + (uint32_t)lane < prev_cnt;
                            const bool un = in && bv == kNone;
                            const unsigned long long m = __ballot(un);
                            const uint32_t at = n_new + (uint32_t)__popcll(m & lt_mask);
                            if (un && at < kNewMax) {
                                s_newx[sp][at] = x;
                                if constexpr (RANGE) s_newh[sp][at] = (uint16_t)(h0 + (uint32_t)lane);
                            }
                            n_new += (uint32_t)__popcll(m);
                        };
                        rank_group(0u, bv0, x0);
                        if (prev_cnt > 64u) {
                            rank_group(64u, bv1, x1);
                            if (prev_cnt > 128u) rank_group(128u, bv2, x2);
                        }
                        for (uint32_t h0 = 192u; h0 < prev_cnt; h0 += 64u) {
                            const uint32_t h = min(h0 + (uint32_t)lane, prev_cnt - 1u);
                            rank_group(h0, s_best[prev_bb][h], s_hits[prev_off + h]);
                        }
                        if (n_new > total_av || n_new > kNewMax || A0 + n_new > cap_eff) pflags |= K8_OVF;
                        seq_base = RANGE ? (C(12) - i_seg0) << 10 : next_seq;
                        next_seq += n_new;
                        fam_open = true;
                    }
                    // the quiet probes between the previous probe and this step's age its new arms before they can match
                    const uint64_t g_new = (uint64_t)step + C(8);
                    const bool stillborn = g_new >= (uint64_t)G;
                    if (stillborn) pflags |= K8_STILL;
                    if constexpr (RANGE) {
                        // (the step whose probe is this run's cut, see the arm waves: the arms born of the probe in front of the cut
                        // as they will be pulled, and the family state)
                        if (K7_RARE(has_cur && emit_from_i != 0u && emit_from_i != 0xFFFFFFFFu && C(6) == emit_from_i)) {
                            if (n_new && !stillborn && !(pflags & K8_OVF)) {
                                const uint32_t gap_n = g_new > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)g_new;
                                for (uint32_t r0 = 0; r0 < n_new; r0 += 64u) {
                                    const uint32_t r = min(r0 + (uint32_t)lane, n_new - 1u);
                                    const uint32_t x = (uint32_t)s_newx[sp][r];
                                    dump_arm(1u, r0 + (uint32_t)lane < n_new, seq_of(seq_base, r, sp), C(12), C(12) + k, x, x + k, thr0, gap_n);
                                }
                            }
                            if (lane == 0) {
                                uint32_t *m = P.run_meta + seg * 16ull + 8ull;
                                m[1] = 0u;
                                m[2] = fam_open ? 1u : 0u;
                                m[3] = 0u;  // (no flush is ever held back: every unmatched hit becomes an arm)
                            }
                        }
                    }
                    // ---- the new arms' offers to this step's probe, on their behalf --------------------------------------
                    if (n_new && !(pflags & K8_OVF)) {
                        const Cur q{k, C(5), C(19), C(3), C(4), C(9), C(1), C(2)};
                        const uint32_t ns_b = has_cur ? uni(s_nstash[C(9)]) : 0u;
                        // (the first 64 here; the others are shared among all waves at the top of the next step: one wave
                        // offering for hundreds of new arms, 64 at a time, would be the step)
                        const bool big = has_cur && !stillborn && n_new > 64u;
                        if (big) {
                            pflags |= K8_BIG;
                            big_n = n_new;
                            big_seq = seq_base;
                        }
                        for (uint32_t r0 = 0; r0 < (big ? 64u : n_new); r0 += 64u) {
                            const uint32_t r = r0 + (uint32_t)lane;
                            const bool who = r < n_new && has_cur && !stillborn;
                            uint32_t ch = 0;
                            if (has_cur && !stillborn) {
                                const PosT x = s_newx[sp][min(r, n_new - 1u)];
                                ch = offer_window(q, (PosT)(x + 1u), (WinT)thr0 + (WinT)(k - 1u), seq_of(seq_base, min(r, n_new - 1u), sp), who, ns_b);
                            }
                            if (r < n_new) s_newch[sp][r] = who ? ch : 0u;
                        }
                    }
                    // ---- the pull block of the next step -------------------------------------------------------------------
                    {
                        const uint32_t tk1 = n_new > base1 ? min(av1, n_new - base1) : 0u;
                        const uint32_t tk2 = n_new > base2 ? min(av2, n_new - base2) : 0u;
                        if (lane < NE) s_slot[sp][lane % NWA][lane / NWA] = (av1 << 16) | base1;
                        if constexpr (NE > 64)
                            if (lane + 64 < NE) s_slot[sp][(lane + 64) % NWA][(lane + 64) / NWA] = (av2 << 16) | base2;
                        took1 = (have_prev && !stillborn) ? tk1 : 0u;
                        took2 = (have_prev && !stillborn) ? tk2 : 0u;
                        pending_alive = stillborn ? 0u : n_new;
                        if (lane == 0) {
                            *reinterpret_cast<uint4 *>(&s_pull[sp][0]) = make_uint4(pflags, n_new, seq_base, fam_seq);
                            *reinterpret_cast<uint4 *>(&s_pull[sp][4]) = make_uint4(C(12), C(13), C(8), 0u);
                        }
                        if (pflags & K8_OVF) overflow = true;
                    }
                    // the stash and the occupancy bits of the previous probe: nobody reads them any more
                    if (lane == 0 && have_prev) s_nstash[prev_bb] = 0u;
                    if (have_prev && (uint32_t)lane < kBitWords) s_rowbits[prev_bb][lane] = 0u;
                }
                if (npre & K7_STAGE) store_rows(st_tot, st_buf);
                pv_tabo = C(3);
                pv_g10 = C(5);
                K7T_LAP(0);
                lds_barrier();  // ---- the barrier of the step -----------------------------------------------------
                K7T_LAP(1);
                if (!more) break;
            }
            K7T_FLUSH(0);
#ifdef ASGART_PROFILE_EXTEND
            if (lane == 0) {
                atomicAdd(&P.ctr[26], k7_sum_a);
                atomicAdd(&P.ctr[27], k7_sum_cnt);
                atomicAdd(&P.ctr[21], k7_sum_n);
                const unsigned long long k7_dt = __builtin_amdgcn_s_memtime() - k7_seg0;
                if (atomicMax(&P.ctr[28], k7_dt) < k7_dt) {
                    P.ctr[29] = g0;
                    P.ctr[30] = k7_sum_n << 32;
                    P.ctr[31] = (k7_sum_a << 32) | (k7_sum_cnt & 0xffffffffull);
                }
            }
            k7_sum_a = k7_sum_cnt = k7_sum_n = 0;
#endif
            if (!overflow) {
                // nothing alive is left behind unless the chunk (or the window of a sharded call) ended first
                // (the counts of the last step: behind its barrier)
                const uint32_t (*const fr)[8] = s_free[sp & (kFreeBufs - 1u)];
                const uint32_t fv = lane < NE ? fr[lane % NWA][lane / NWA] : 0u;
                uint32_t total_free = lane_of(wave_incl_scan(fv), 63u);
                if constexpr (NE > 64) {
                    const uint32_t fv2 = lane + 64 < NE ? fr[(lane + 64) % NWA][(lane + 64) / NWA] : 0u;
                    total_free += lane_of(wave_incl_scan(fv2), 63u);
                }
                const bool ran_out = uni(s_end[1]) != 0u;
                if constexpr (RANGE) {
                    // the family state this run ends in: flushes since the cut, open or not, how long a flush is still held back
                    if (lane == 0) {
                        uint32_t *m = P.run_meta + seg * 16ull;
                        m[1] = fam_seq;
                        m[2] = fam_open ? 1u : 0u;
                        m[3] = 0u;
                        m[5] = (uint32_t)wall_clock64() - s_end[3];  // (10-ns ticks)
                    }
                }
                if (fam_open && total_free == (uint32_t)CAP) fam_open = false;
                if (RANGE && !(run.flags & kRunLast)) {
                    // (the segment goes on behind this run: its end is the last range's business)
                } else if (ran_out) {
                    if (lane == 0) atomicAdd(&P.ctr[CT_RANOUT], 1ull);
                } else if (fam_open) {  // arms alive at the end of the chunk void their family (src/automaton.rs:201-203)
                    emit_records(lane == 0, (PosT)0, (PosT)0, (PosT)0, (PosT)0, kTombstone, fam_seq);
                }
            } else if (lane == 0) {
                if constexpr (RANGE) {
                    P.run_meta[seg * 16ull + 4ull] = 1u;  // (the host runs the whole segment instead)
                } else {
                    const unsigned long long at = atomicAdd(P.ovf_count, 1ull);
                    if (P.ovf_list) P.ovf_list[at] = g0;
                }
            }
        } else {
            // =====================================================================================================
            // PLANNING WAVE: walks the probe sequence two steps ahead of the others, writes the commands
            // =====================================================================================================
            const RunParams &rp = P.rp;
            const uint32_t k = (uint32_t)rp.k, step = (uint32_t)rp.step, G = rp.G;
            const uint32_t thr0 = arm_threshold(k, G);
            uint32_t bsh = 3;
            while ((1ull << bsh) < (unsigned long long)G + k) ++bsh;
                    const uint32_t kGenBits = min(kGenMax, max(2u, P.gen_bits));
            const uint32_t g0 = RANGE ? run.g_seg0 : P.seg_list[seg];
            if (lane == 0) {
            heartbeat(P, g0, 0u);
            seg_clock(P);
        }
            const int c = chunk_of_uniform(rp.ch, g0);
            const uint64_t cs = rp.ch.start[c], cl = rp.ch.len[c];
            const uint32_t pb = rp.ch.pbase[c];
            const uint32_t chunk_end = rp.ch.pbase[c + 1];
            const uint32_t w_end = min(chunk_end, rp.win_end(g0));  // (sharded calls: the window ends first)
            const uint32_t g_end = RANGE ? min(w_end, run.g_stop) : w_end;
            if (lane == 0) {
                if constexpr (RANGE) s_end[3] = (uint32_t)wall_clock64();  // (the run's duration goes into its state: option debug)
                s_seg[0] = g0;
                s_seg[1] = cs;
                s_seg[2] = cl | ((unsigned long long)((rp.mode_of(c) >> 1) & 1u) << 63);
            }
            uint32_t quiet = 0, pend = 0;
            bool done = false, giveup = false;
            uint32_t hbuf = 2;  // (the first batch moves it to 0)
            // ---- the batch under the cursor ---------------------------------------------------------------------
            uint32_t g = RANGE ? run.g_begin : g0, nbb = 0, pos = 0, f_l = 0, rel_l = 0, tot = 0;
            unsigned long long hm = 0, qm = 0, base = 0;
            bool staged = false;  // the rows of the batch under the cursor are in s_hits[hbuf] (or on their way)
            auto load_batch = [&]() {  // -> false: a probe with more hits than the staging area
                const uint32_t nb = min(64u, g_end - g);
                if (lane == 0) heartbeat(P, g0, g);
                f_l = (uint32_t)lane < nb ? P.p_filt[g + lane] : kSkipN;
                const unsigned long long r_l = (uint32_t)lane < nb ? P.row_off[g + lane] : 0ull;
                const unsigned long long r_hi = uni(P.row_off[g + nb]);
                base = lane_of(r_l, 0u);
                unsigned long long r_next = __shfl_down(r_l, 1);
                if ((uint32_t)lane + 1 >= nb) r_next = r_hi;
                const bool fits = (uint32_t)lane < nb && r_next - base <= (unsigned long long)HB;
                const unsigned long long fm = __ballot(fits);
                nbb = (~fm == 0ull) ? 64u : (uint32_t)(__ffsll((long long)~fm) - 1);
                if (nbb > nb) nbb = nb;
                if (nbb == 0) return false;
                rel_l = (uint32_t)(r_l - base);
                tot = (uint32_t)((nbb == nb ? r_hi : lane_of(r_l, nbb)) - base);
                const unsigned long long in_batch = nbb >= 64 ? ~0ull : ((1ull << nbb) - 1ull);
                hm = __ballot(f_l >= 1u && f_l < kPending) & in_batch;
                qm = __ballot(f_l == 0u) & in_batch;
                pos = 0;
                staged = false;
                return true;
            };
            struct Probe {
                uint32_t cnt, off, tb, bb, g10, pend;
                uint64_t i;
            };
            struct Plan {
                uint32_t flags, pre;           // command flags; K7_STAGE: its batch is staged during the step before
                unsigned long long st_base;    // staging request
                uint32_t st_tot, st_buf;
                Probe q;
            };
            const Probe no_probe{0, 0, 0, 0, 0, 0, 0};
            bool opened = false;  // the probe just found opened a batch (its rows are to be staged)
            auto next_probe = [&](Probe &nx) -> bool {
                for (;;) {
                    const unsigned long long hmr = pos >= 64 ? 0ull : (hm >> pos) << pos;
                    const uint32_t b = hmr ? (uint32_t)(__ffsll((long long)hmr) - 1) : 64u;
                    const unsigned long long upto = b >= 64 ? ~0ull : ((1ull << b) - 1ull);
                    const unsigned long long from = pos >= 64 ? 0ull : ~((1ull << pos) - 1ull);
                    const uint32_t q = (uint32_t)__popcll(qm & upto & from);
                    if (q) {
                        quiet += q;
                        pend += q * step;
                        if (quiet >= rp.tstar) {  // every arm is dead (gap >= G): the segment is over
                            done = true;
                            return false;
                        }
                    }
                    if (hmr) {
                        quiet = 0;
                        pos = b + 1;
                        opened = false;
                        if (!staged) {
                            hbuf = hbuf == 2u ? 0u : hbuf + 1u;
                            opened = true;
                            staged = true;
                        }
                        nx.cnt = lane_of(f_l, b);
                        nx.off = hbuf * (uint32_t)HB + lane_of(rel_l, b);
                        nx.i = (uint64_t)(g + b - pb + 1) * step;
                        nx.pend = pend;
                        pend = 0;
                        return true;
                    }
                    g += nbb;
                    if (g >= g_end) return false;
                    if (!load_batch()) {
                        giveup = true;
                        return false;
                    }
                }
            };
            auto number_probe = [&](Probe &q) {
                bool wrap = false;
                if ((gen + 1u) >> kGenBits) {
                    wrap = true;
                    gen = 0;
                }
                ++gen;
                q.g10 = gen << 10;
                q.tb = par;
                q.bb = tri;
                par ^= 1u;
                tri = tri == 2u ? 0u : tri + 1u;
                return wrap;
            };
            // the plan of the step behind one with flags `last_flags`.  A generation wrap takes two steps of its own: one
            // that resolves the probe before (if any) and clears the tables, one in which the new probe's hits are indexed.
            uint32_t held_stage = 0;  // 1: the indexing step is next; 2: then the held probe
            Plan held{};
            bool ended = false, tail_done = false;
            auto advance = [&](uint32_t last_flags) -> Plan {
                Plan n{};
                n.q = no_probe;
                if (held_stage == 1u) {
                    held_stage = 2u;
                    return n;  // (no probe: the held probe's hits are indexed during it)
                }
                if (held_stage == 2u) {
                    held_stage = 0u;
                    return held;
                }
                Probe nx = no_probe;
                bool found = false;
                if (!ended) found = next_probe(nx);
                if (giveup) {
                    n.flags = K7_GIVEUP;
                    return n;
                }
                if (!found) {
                    ended = true;
                    if (!tail_done) {  // the last probe is resolved, the trailing quiet probes age the arms ...
                        tail_done = true;
                        n.flags = (last_flags & K7_CUR) ? K7_PREV : 0u;
                        n.q.pend = pend;
                        pend = 0;
                        return n;
                    }
                    n.flags = K7_LAST;  // ... and the arms born of the last probe are pulled
                    return n;
                }
                const bool wrap = number_probe(nx);
                Plan p{};
                p.q = nx;
                p.pre = opened ? K7_STAGE : 0u;
                p.st_base = base;
                p.st_tot = tot;
                p.st_buf = hbuf;
                p.flags = K7_CUR | (opened ? K7_LATE : 0u);
                if (wrap) {
                    held = p;  // (behind two steps without a probe: no predecessor to resolve)
                    held_stage = 1u;
                    n.flags = ((last_flags & K7_CUR) ? K7_PREV : 0u) | K8_CLEARNOW;
                    return n;
                }
                p.flags |= (last_flags & K7_CUR) ? K7_PREV : 0u;
                return p;
            };
            struct Before {
                uint32_t cnt, i_lo, i_hi, off, bb;
            };
            auto write_cmd = [&](uint32_t slot, const Plan &p, const Before &b) {
                if (lane == 0) {
                    uint4 *o = reinterpret_cast<uint4 *>(&s_cmd[slot][0]);
                    o[0] = make_uint4(p.flags, p.q.cnt, p.q.off, p.q.tb * kTabBytes);
                    o[1] = make_uint4(p.q.bb * (uint32_t)(HB * 4), p.q.g10, (uint32_t)p.q.i, (uint32_t)(p.q.i >> 32));
                    o[2] = make_uint4(p.q.pend, p.q.bb, p.pre, b.cnt);
                    o[3] = make_uint4(b.i_lo, b.i_hi, b.off, b.bb * (uint32_t)(HB * 4));
                    s_cmd[slot][23] = b.bb;
                    if (p.pre & K7_STAGE) o[6] = make_uint4((uint32_t)p.st_base, (uint32_t)(p.st_base >> 32), p.st_tot, p.st_buf);
                }
            };
            auto before_of = [&](const Plan &p) {
                return Before{p.q.cnt, (uint32_t)p.q.i, (uint32_t)(p.q.i >> 32), p.q.off, p.q.bb};
            };
            if (lane < 3) {  // the run's constants, once per segment, in every command block
                uint4 *o = reinterpret_cast<uint4 *>(&s_cmd[lane][0]);
                o[4] = make_uint4(k, step, G, bsh);
                s_cmd[lane][20] = (uint32_t)rp.M;
                s_cmd[lane][21] = (uint32_t)(rp.M >> 32);
                s_cmd[lane][22] = thr0;
            }

            // ---- the first two steps are planned before the loop ---------------------------------------------------
            Plan p_cur{}, p_next{};
            p_cur.q = p_next.q = no_probe;
            if (!load_batch()) giveup = true;
            if (giveup) p_cur.flags = K7_GIVEUP;
            else p_cur = advance(0u);  // (the segment starts with a hit-probe: its batch is staged and indexed below)
            if (lane == 0) {
                s_mid[0] = p_cur.pre;
                *reinterpret_cast<uint4 *>(&s_mid[4]) = make_uint4((uint32_t)p_cur.st_base, (uint32_t)(p_cur.st_base >> 32), p_cur.st_tot, p_cur.st_buf);
            }
            {
                Plan none{};
                none.q = no_probe;
                write_cmd(0u, p_cur, before_of(none));
            }
            if (!(p_cur.flags & (K7_GIVEUP | K7_LAST))) {
                p_next = advance(p_cur.flags);
                write_cmd(1u, p_next, before_of(p_cur));
            }
            lds_barrier();  // (1)
            if (p_cur.pre & K7_STAGE) {
                fetch_rows(p_cur.st_base, p_cur.st_tot);
                store_rows(p_cur.st_tot, p_cur.st_buf);
            }
            lds_barrier();  // (2)
            if ((p_cur.flags & K7_CUR) && indexes(p_cur.q.cnt))
                insert_hits(p_cur.q.cnt, p_cur.q.off, p_cur.q.tb * kTabBytes, p_cur.q.bb * (uint32_t)(HB * 4), p_cur.q.bb, p_cur.q.g10, bsh, false, 0ull);
            lds_barrier();  // (3)
            uint32_t pv_tabo = 0, pv_g10 = 0;
            for (uint32_t sc = 0, sp = 0;; sc = sc == 2u ? 0u : sc + 1u, sp ^= 1u) {
                const uint32_t sn = sc == 2u ? 0u : sc + 1u;
                const uint32_t cw = s_cmd[lane < 32 ? sc : sn][lane & 31];
                const uint32_t pflags_prev = s_pull[sp ^ 1u][0];  // (the ranking wave's decision of the step before)
                const uint32_t pn_prev = s_pull[sp ^ 1u][1], pseq_prev = s_pull[sp ^ 1u][2];
                K7T_MARK();
                auto C = [&](int j) { return (uint32_t)__builtin_amdgcn_readlane((int)cw, j); };
                auto N = [&](int j) { return (uint32_t)__builtin_amdgcn_readlane((int)cw, 32 + j); };
                const uint32_t flags = C(0);
                if ((flags & K7_GIVEUP) || (uni(pflags_prev) & K8_OVF)) {
                    overflow = true;
                    break;
                }
                const bool more = !(flags & K7_LAST);
                const uint32_t nflags = more ? N(0) : 0u;
                if (uni(pflags_prev) & K8_BIG) {
                    big_rounds(Cur{k, pv_g10, bsh, pv_tabo, C(15), C(23), C(11), C(14)}, uni(pn_prev), uni(pseq_prev), thr0, sp ^ 1u);
                    lds_barrier();
                }
                if (flags & K8_CLEARNOW) clear_table();
                const uint32_t npre = more ? N(10) : 0u;
                const unsigned long long st_base = ((unsigned long long)N(25) << 32) | N(24);
                const uint32_t st_tot = N(26), st_buf = N(27);
                if (npre & K7_STAGE) fetch_rows(st_base, st_tot);
                if (more && (nflags & K7_CUR) && indexes(N(1)))
                    insert_hits(N(1), N(2), N(3), N(4), N(9), N(5), bsh, (nflags & K7_LATE) != 0u, st_base + (N(2) - st_buf * (uint32_t)HB));
                // ---- the step after next ---------------------------------------------------------------------------------
                if (more && !(nflags & (K7_LAST | K7_GIVEUP))) {
                    const Plan p_after = advance(nflags);
                    write_cmd(sc == 0u ? 2u : sc - 1u, p_after, Before{N(1), N(6), N(7), N(2), N(9)});
                    if ((p_after.flags & K7_LAST) && lane == 0) {  // (what the ranking wave needs when the segment is over)
                        s_end[1] = (!done && g_end < chunk_end && (!RANGE || w_end == g_end)) ? 1u : 0u;
                    }
                }
                if (npre & K7_STAGE) store_rows(st_tot, st_buf);
                pv_tabo = C(3);
                pv_g10 = C(5);
                K7T_LAP(0);
                lds_barrier();  // ---- the barrier of the step -----------------------------------------------------
                K7T_LAP(1);
                if (!more) break;
            }
#ifndef K8_TRACE_WAVE
            K7T_FLUSH(2);  // (diagnostic build: the planning wave is the third class of the step trace)
#endif
            (void)thr0;
            (void)G;
        }
        // leave no arm behind for the next segment
#pragma unroll
        for (int L = 0; L < S; ++L) a_seq[L] = kNoSeq;
        livemask = 0;
        lds_barrier();
    }
    rec_flush(rec_alloc, P, lane);
    wg_busy(P);
}

}  // namespace asgart
