// extend_k7_dev.hpp -- "K7": the arm-resident extension kernel with SPECIALISED waves.
//
// Same automaton as the other extension kernels (reference src/automaton.rs:57-204; representation of
// pipeline_dev.hpp: only live arms are kept, winners by creation number, families by records), same hit table and
// per-arm code as K6 (extend_fast_dev.hpp).  What changes is who runs what.
//
// A wave issues at most one instruction every four cycles, so the time of a hit-probe is the length of the longest
// instruction stream any one wave runs for it.  In K6 every wave that holds arms runs everything: the bookkeeping of
// the probe loop, the ranking of the empty slots and of the unmatched hits, then its arms -- ~1 250 instructions per
// probe on the busy waves of a tandem-array segment, of which the arms themselves (look up two rows and offer; read
// the winners back and update) are a tenth.  Here
//
//   * the LAST wave is the CONTROL wave.  It holds no arms.  It walks the probe sequence (batches, quiet runs, segment
//     end), numbers the table generations, decides flushes and overflow, ranks the unmatched hits of the previous
//     probe into a compact list, ranks the empty slots from the counts the arm waves publish, and tells the other
//     waves what to do through a command block in LDS;
//   * the other waves are ARM waves: per step they read the command, resolve the previous probe for their arms, age
//     them, offer to the current probe's hits, and -- after the first barrier -- pull their new arms from the
//     control wave's list and let those offer too.  No scans, no ranking, no loop bookkeeping.
//
// A step (one hit-probe t with its predecessor t-1 still to resolve) is two barriers:
//
//     interval A   arm waves: resolve t-1 (winners -> ExtendArm / age / retire), the quiet probes' age, offers to t
//                  control:   flush decision of t-1, unmatched hits of t-1 -> list + count, slot ranks, overflow;
//                             the next step's command (next hit-probe, its quiet run, end of the segment)
//     -- barrier 1 --
//     interval B   arm waves: new arms of t-1 from the list (owner pull by rank), their offers to t; free counts
//                  everyone:  the hits of t+1 are indexed (top threads); a batch of hit rows is staged when the
//                             control wave asked for it
//     -- barrier 2 --  (every offer to t is in: its winners are final)
//
// The hit rows of 64 probes are staged at a time into one of two buffers, so that the rows of t-1 survive the staging
// of the next batch; the first probe of a batch cannot be indexed ahead (its rows arrive with barrier 2): its step is
// "late" -- indexed in interval A, all offers in interval B.  The same form starts a segment and follows a clearing
// of the tables (generation wrap).  Creation numbers, record keys and every transition are those of K6: results are
// identical (the tier tests force every segment through this kernel as well).
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

template <class PosT, int S, int NT, int HB, int kRows = 1024, int kE = 2>
__global__ __launch_bounds__(NT) void extend_k7_kernel(ExtParams<PosT> P) {
    constexpr int NW = NT / 64, NWA = NW - 1;  // waves; arm waves
    constexpr int CAP = S * NWA * 64;
    constexpr int NE = S * NWA;                // (layer, wave) entries of the free counts
    constexpr uint32_t kNone = 0xFFFFFFFFu;    // best[]: no arm accepts this hit
    constexpr uint32_t kNever = 0xFFFFFFFEu;   // what a candidate read of an idle lane returns: no creation number
    constexpr uint32_t kCoop = 0xFFFFFFFFu;    // candidate register: more than three / wide window / stash overflow
    constexpr uint32_t kStash = 64;
    constexpr uint32_t kRowsLoop = 6;
    constexpr bool kWidePos = sizeof(PosT) == 8;
    constexpr uint32_t kTagShift = kWidePos ? 42u : 32u;
    constexpr uint32_t kGenMax = kWidePos ? 12u : 22u;
    constexpr unsigned long long kPosMask = (1ull << kTagShift) - 1ull;
    using WinT = typename std::conditional<kWidePos, uint64_t, uint32_t>::type;
    static_assert(NW >= 2 && HB <= 1024 && S <= 8 && NE <= 128 && (kRows & (kRows - 1)) == 0 && (kE == 2 || kE == 4), "shape");
    if (NT >= 1024 && P.hi_prio) __builtin_amdgcn_s_setprio(3);

    __shared__ __attribute__((aligned(16))) unsigned long long s_tab[2][kRows * kE];
    __shared__ PosT s_hits[2 * HB];
    __shared__ uint8_t s_hflag[2 * HB];
    __shared__ uint32_t s_best[3][HB];
    __shared__ unsigned long long s_stash[3][kStash];
    __shared__ uint32_t s_nstash[3];
    __shared__ __attribute__((aligned(16))) uint32_t s_free[NWA][8];   // per (arm wave, layer): empty slots
    __shared__ __attribute__((aligned(16))) uint32_t s_cmd[2][12];     // the command of a step (by step parity)
    __shared__ __attribute__((aligned(16))) uint32_t s_mid[8];         // decided in interval A, read in interval B
    __shared__ uint32_t s_fam[2];                                      // family ordinal as of the step (by step parity)
    __shared__ uint32_t s_base[NE];                                    // rank of the first empty slot of (layer, wave)
    __shared__ PosT s_new[HB];                                         // the unmatched hits of the previous probe, by rank
    __shared__ PosT s_cle[CAP], s_crs[CAP];                            // cold fields of an arm, by slot
    __shared__ unsigned long long s_seg[3];                            // g0, chunk start, chunk length (for the records)
    __shared__ unsigned long long s_bcast;
    __shared__ uint32_t s_sink[64];
    __shared__ uint32_t s_never;

    const int tid = threadIdx.x, lane = tid & 63;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane(tid >> 6);
    const bool is_ctl = wave == (uint32_t)NWA;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const RunParams &rp = P.rp;
    const uint64_t n_seg = *P.n_seg_ptr;
    const uint32_t k = (uint32_t)rp.k, step = (uint32_t)rp.step, G = rp.G;
    const uint32_t thr0 = arm_threshold(k, G);
    uint32_t bsh = 3;
    while ((1ull << bsh) < (unsigned long long)G + k) ++bsh;
    bsh += P.fast_bsh;
    const uint32_t kGenBits = min(kGenMax, max(2u, P.gen_bits));
    const uint32_t cap_eff = min((uint32_t)CAP, P.cap_limit);
    const WinT w_loop = (WinT)(kRowsLoop - 1u) << bsh;
    const bool use_flag = P.hit_flag != nullptr;
    RecAlloc rec_alloc;
    K7T_DECL;
    K7U_DECL;

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
    // (control wave) generation of the hit tables, table parity, best[] / stash buffer: they run on from segment to
    // segment, so that what an abandoned segment indexed ahead is stale for the next one
    uint32_t gen = 0, par = 0, tri = 0;
    auto clear_table = [&]() {
        for (uint32_t e = tid; e < (uint32_t)(2 * kRows * kE); e += NT) (&s_tab[0][0])[e] = 0ull;
    };
    clear_table();
    if (tid == 0) s_never = kNever;
    lds_barrier();

    auto tag_of = [&](unsigned long long e) { return (uint32_t)(e >> kTagShift); };
    auto pos_of = [&](unsigned long long e) { return (PosT)(e & kPosMask); };

    for (uint32_t n_fetch = 0; !P.max_items || n_fetch < P.max_items; ++n_fetch) {
        if (tid == 0) s_bcast = atomicAdd(P.cursor, 1ull);
        if (tid < 3) s_nstash[tid] = 0u;
        if (tid < 2) s_fam[tid] = 0u;
        for (uint32_t j = tid; j < (uint32_t)(NWA * 8); j += NT) (&s_free[0][0])[j] = 64u;
        lds_barrier();
        const unsigned long long seg = uni(s_bcast);
        lds_barrier();
        if (seg >= n_seg) break;
        const uint32_t g0 = P.seg_list[seg];

        // ---- records (cold: what a record needs beyond the arm is read where it is written) -------------------------
        auto emit_records = [&](bool emit, PosT ls, PosT le, PosT rs, PosT re, uint32_t seq, uint32_t fam_seq) {
            const unsigned long long em = __ballot(emit);
            if (!em) return;
            const unsigned long long at = rec_slot(rec_alloc, P, em, lane);
            if (emit && at < P.rec_cap) {
                const uint64_t cs = s_seg[1], cl = s_seg[2];
                const uint64_t ll = (uint64_t)le - (uint64_t)ls;
                SdRec r;
                r.g_start = g0;
                r.fam_seq = fam_seq;
                r.create_seq = seq;
                r.pad = 0;
                r.sd.left = rp.reverse ? cs + cl - (uint64_t)ls - ll : (uint64_t)ls + cs;  // src/bin/asgart.rs:229-237
                r.sd.right = rs;
                r.sd.left_length = ll;
                r.sd.right_length = (uint64_t)re - (uint64_t)rs;
                P.recs[at] = r;
            }
        };
        // index the hits of one probe (cnt hits at s_hits[off..]) under generation tag g10 in table tb, best[] /
        // stash buffer bb; the TOP threads do it
        auto insert_hits = [&](uint32_t cnt, uint32_t off, uint32_t tb, uint32_t bb, uint32_t g10) {
            for (uint32_t h = (uint32_t)(NT - 1 - tid); h < cnt; h += NT) {
                const PosT x = s_hits[off + h];
                s_best[bb][h] = kNone;
                unsigned long long e = ((unsigned long long)(g10 | h) << kTagShift) | ((unsigned long long)x & kPosMask);
                unsigned long long *row = &s_tab[tb][(((uint32_t)((uint64_t)x >> bsh)) & (uint32_t)(kRows - 1)) * (uint32_t)kE];
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
        auto coop_offer = [&](PosT lo, WinT w, uint32_t key, uint32_t cnt, uint32_t off, uint32_t bb) {
            for (uint32_t h0 = 0; h0 < cnt; h0 += 64u) {
                const uint32_t h = h0 + (uint32_t)lane;
                if (h < cnt && (WinT)(PosT)(s_hits[off + h] - lo) < w) atomicMin(&s_best[bb][h], key);
            }
        };
        auto coop_resolve = [&](PosT lo, WinT w, uint32_t key, uint32_t cnt, uint32_t off, uint32_t bb, PosT &x_out) {
            uint32_t hmax = kNone;
            for (uint32_t h0 = 0; h0 < cnt; h0 += 64u) {
                const uint32_t h = h0 + (uint32_t)lane;
                PosT x = 0;
                bool ok = false;
                if (h < cnt) {
                    x = s_hits[off + h];
                    ok = (WinT)(PosT)(x - lo) < w && s_best[bb][h] == key;
                }
                const unsigned long long bm = __ballot(ok);
                if (bm) {
                    const uint32_t top = 63u - (uint32_t)__clzll((long long)bm);
                    hmax = h0 + top;
                    if constexpr (kWidePos) x_out = (PosT)lane_of((unsigned long long)x, top);
                    else x_out = (PosT)lane_of((uint32_t)x, top);
                }
            }
            return hmax;
        };

        // the probe a command names
        struct Probe {
            uint32_t cnt, off, tb, bb, g10, pend;
            uint64_t i;
        };
        // ---- offers of the arms of layer L to the hits of probe q (table q.tb, winners best[q.bb]); only the lanes with
        // `who` set take part; returns the candidate word of the lane -------------------------------------------------
        auto offers = [&](int L, bool who, const Probe &q, uint32_t ns, bool povf) -> uint32_t {
            const uint32_t g10 = q.g10, tb = q.tb, bb = q.bb;
            const PosT lo = (PosT)(a_re[L] - k + 1u);
            const WinT w = (WinT)a_thr[L] + (WinT)(k - 1u);
            const uint32_t key = a_seq[L];
            const bool narrow = who && w <= w_loop;
            const WinT w_eff = narrow ? w : (WinT)0;  // (an empty window accepts nothing)
            uint32_t ch = 0, nc = 0;
            uint32_t *const sink = &s_sink[lane];
            auto offer = [&](unsigned long long e) {
                const uint32_t d = tag_of(e) - g10;
                const WinT t = d < 1024u ? (WinT)(PosT)(pos_of(e) - lo) : ~(WinT)0;
                const bool ok = t < w_eff;
                atomicMin(ok ? &s_best[bb][d & 1023u] : sink, key);
                ch = ok ? ((ch << 10) | d) : ch;
                nc += ok ? 1u : 0u;
            };
            auto offer_row = [&](uint32_t b) {
                const ulonglong2 *rr = reinterpret_cast<const ulonglong2 *>(&s_tab[tb][(b & (uint32_t)(kRows - 1)) * (uint32_t)kE]);
                if constexpr (kE == 4) {
                    const ulonglong2 f0 = rr[0], f1 = rr[1];
                    offer(f0.x); offer(f0.y); offer(f1.x); offer(f1.y);
                } else {
                    const ulonglong2 f0 = rr[0];
                    offer(f0.x); offer(f0.y);
                }
            };
            const uint32_t b0 = (uint32_t)((uint64_t)lo >> bsh);
            const uint32_t n_rows = narrow ? (uint32_t)((((uint64_t)lo & ((1ull << bsh) - 1ull)) + (uint64_t)w - 1ull) >> bsh) + 1u : 0u;
            {   // the two rows of a narrow window: all reads in flight together
                const ulonglong2 *r0 = reinterpret_cast<const ulonglong2 *>(&s_tab[tb][(b0 & (uint32_t)(kRows - 1)) * (uint32_t)kE]);
                const ulonglong2 *r1 = reinterpret_cast<const ulonglong2 *>(&s_tab[tb][((b0 + 1u) & (uint32_t)(kRows - 1)) * (uint32_t)kE]);
                if constexpr (kE == 4) {
                    const ulonglong2 e0 = r0[0], e1 = r0[1], e2 = r1[0], e3 = r1[1];
                    offer(e0.x); offer(e0.y); offer(e1.x); offer(e1.y);
                    offer(e2.x); offer(e2.y); offer(e3.x); offer(e3.y);
                } else {
                    const ulonglong2 e0 = r0[0], e2 = r1[0];
                    offer(e0.x); offer(e0.y); offer(e2.x); offer(e2.y);
                }
            }
            for (uint32_t r = 2; __ballot(r < n_rows); ++r) {
                K7C(53, 1);
                offer_row(b0 + r);
            }
            for (uint32_t s = 0; s < min(ns, kStash); ++s) {
                K7C(56, 1);
                offer(s_stash[bb][s]);
            }
            ch = nc > 3u ? kCoop : (ch & 0x3FFFFFFFu) | (nc << 30);
            // arms too wide for the table walk -- and every arm when the stash overflowed
            unsigned long long sm = __ballot(who && (!narrow || povf));
            if (sm) {
                if (who && (!narrow || povf)) ch = kCoop;
                K7C(54, __popcll(sm));
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
                    coop_offer(lo_u, w_u, lane_of(key, l), q.cnt, q.off, bb);
                }
            }
            return ch;
        };

        auto read_cmd = [&](uint32_t par, uint32_t &flags, Probe &q) {
            const uint4 c0 = *reinterpret_cast<const uint4 *>(&s_cmd[par][0]);
            const uint4 c1 = *reinterpret_cast<const uint4 *>(&s_cmd[par][4]);
            flags = uni(c0.x);
            q.cnt = uni(c0.y);
            q.off = uni(c0.z);
            q.tb = uni(c0.w) & 1u;
            q.bb = uni(c0.w) >> 1;
            q.g10 = uni(c1.x);
            q.i = ((uint64_t)uni(c1.z) << 32) | uni(c1.y);
            q.pend = uni(c1.w);
        };
        // stage `tot` hit rows starting at CSR entry `base` into buffer `buf` (all threads)
        auto stage_rows = [&](unsigned long long base, uint32_t tot, uint32_t buf) {
            for (uint32_t r = tid; r < tot; r += NT) {
                s_hits[buf * (uint32_t)HB + r] = P.hits[base + r];
                if (use_flag) s_hflag[buf * (uint32_t)HB + r] = P.hit_flag[base + r];
            }
        };
        // what interval B does for everyone once the mid-step block is known
        auto mid_actions = [&](uint32_t mflags) {
            if (mflags & K7_STAGE) {
                const uint4 m1 = *reinterpret_cast<const uint4 *>(&s_mid[4]);
                stage_rows(((unsigned long long)uni(m1.y) << 32) | uni(m1.x), uni(m1.z), uni(m1.w));
            }
            if (mflags & K7_CLEAR) clear_table();
        };

        bool overflow = false;
        if (!is_ctl) {
            // =====================================================================================================
            // ARM WAVES
            // =====================================================================================================
            Probe prev{0, 0, 0, 0, 0, 0, 0};
            lds_barrier();  // (1) the control wave has published the first batch's staging request
            mid_actions(uni(s_mid[0]));
            lds_barrier();  // (2) the rows are in; the first command is there
            for (uint32_t sp = 0;; sp ^= 1u) {
                uint32_t flags;
                Probe cur;
                K7T_MARK();
                K7T_STEP();
                K7U_MARK();
                read_cmd(sp, flags, cur);
                if (flags & K7_GIVEUP) {
                    overflow = true;
                    break;
                }
                const bool has_prev = (flags & K7_PREV) != 0u, has_cur = (flags & K7_CUR) != 0u, late = (flags & K7_LATE) != 0u;
                // ---------------------------------------------------------------- interval A ----------------
                if (has_cur && late) insert_hits(cur.cnt, cur.off, cur.tb, cur.bb, cur.g10);
                const uint32_t fam_a = uni(s_fam[sp ^ 1u]);
                const uint32_t ns_v = (has_cur && !late && livemask) ? s_nstash[cur.bb] : 0u;
                uint32_t wasfree = 0;  // per lane, bit L: the slot of layer L was empty before this step
                K7U_LAP(0);
#pragma unroll
                for (int L = 0; L < S; ++L) {
                    if (!(livemask >> L)) {
                        wasfree |= ((1u << S) - 1u) & ~((1u << L) - 1u);  // this layer and the ones above: all empty
                        break;
                    }
                    const bool was_free = a_seq[L] == kNoSeq;
                    wasfree |= was_free ? 1u << L : 0u;
                    if (!(livemask & (1u << L))) continue;
                    bool won = false;
                    PosT xw = 0;
                    if (has_prev) {
                        // the last hit (SA order) this arm won, if any: src/automaton.rs:133-150 apply in hit order
                        const uint32_t ch = c_h[L];
                        const bool coop = !was_free && ch == kCoop;
                        const uint32_t nc = (was_free || coop) ? 0u : ch >> 30;
                        uint32_t cb[3];
#pragma unroll
                        for (uint32_t j = 0; j < 3; ++j) cb[j] = *(j < nc ? &s_best[prev.bb][(ch >> (10u * j)) & 1023u] : &s_never);
                        uint32_t hw = 0;  // 1 + that hit
#pragma unroll
                        for (uint32_t j = 0; j < 3; ++j) {
                            const uint32_t hj = (ch >> (10u * j)) & 1023u;
                            hw = cb[j] == a_seq[L] ? max(hw, hj + 1u) : hw;
                        }
                        hw = (was_free || coop) ? 0u : hw;
                        xw = s_hits[prev.off + (hw ? hw - 1u : 0u)];
                        unsigned long long sm = __ballot(coop);
                        K7C(55, __popcll(sm));
                        while (sm) {  // more than three candidates / wide window: resolved cooperatively
                            const uint32_t l = (uint32_t)(__ffsll((long long)sm) - 1);
                            sm &= sm - 1ull;
                            const PosT lo = (PosT)(a_re[L] - k + 1u);
                            const WinT w = (WinT)a_thr[L] + (WinT)(k - 1u);
                            PosT lo_u, x_u = 0;
                            WinT w_u;
                            if constexpr (kWidePos) {
                                lo_u = (PosT)lane_of((unsigned long long)lo, l);
                                w_u = (WinT)lane_of((unsigned long long)w, l);
                            } else {
                                lo_u = (PosT)lane_of((uint32_t)lo, l);
                                w_u = (WinT)lane_of((uint32_t)w, l);
                            }
                            const uint32_t hm_ = coop_resolve(lo_u, w_u, lane_of(a_seq[L], l), prev.cnt, prev.off, prev.bb, x_u);
                            if ((uint32_t)lane == l && hm_ != kNone) {
                                hw = hm_ + 1u;
                                xw = x_u;
                            }
                        }
                        won = hw != 0u;
                    }
                    // ExtendArm (src/automaton.rs:133-150) or one more step of age (:166-171), then the quiet probes
                    // between the previous probe and this one
                    uint32_t thr_new;
                    if constexpr (kWidePos) thr_new = arm_threshold((uint64_t)(prev.i + k) - (uint64_t)a_ls[L], G);
                    else thr_new = max(G, ((uint32_t)(prev.i + k) - (uint32_t)a_ls[L]) / 10u);
                    const uint64_t sum_g = (uint64_t)(won ? 0u : a_gap[L]) + (has_prev && !won ? step : 0u) + cur.pend;
                    const uint32_t aged = sum_g > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)sum_g;
                    a_re[L] = won ? (PosT)(xw + k) : a_re[L];
                    if (won) s_cle[L * (NWA * 64) + tid] = (PosT)(prev.i + k);
                    a_thr[L] = won ? thr_new : a_thr[L];
                    a_gap[L] = aged;
                    const bool dead = !was_free && aged >= G;  // never matches again
                    if (__ballot(dead)) {
                        const PosT rs = s_crs[L * (NWA * 64) + tid];
                        const bool report = dead && (uint64_t)(a_re[L] - rs) >= rp.M;
                        if (__ballot(report)) emit_records(report, a_ls[L], s_cle[L * (NWA * 64) + tid], rs, a_re[L], a_seq[L], fam_a);
                    }
                    a_seq[L] = dead ? kNoSeq : a_seq[L];
                }
                uint32_t ns = 0;
                bool povf = false;
                K7U_LAP(1);
                if (has_cur && !late && livemask) {
                    ns = uni(ns_v);
                    povf = ns > kStash;
#pragma unroll
                    for (int L = 0; L < S; ++L) {
                        if (!(livemask >> L)) break;
                        if (livemask & (1u << L)) c_h[L] = offers(L, a_seq[L] != kNoSeq, cur, ns, povf);
                    }
                }
                K7U_LAP(2);
                K7T_LAP(0);
                lds_barrier();  // ---- barrier 1 --------------------------------------------------------------
                K7T_LAP(1);
                K7U_MARK();
                // ---------------------------------------------------------------- interval B ----------------
                const uint4 m0v = *reinterpret_cast<const uint4 *>(&s_mid[0]);
                const uint32_t mflags = uni(m0v.x);
                if (mflags & K7_OVF) {
                    overflow = true;
                    break;
                }
                if (has_cur && late) {
                    ns = uni(s_nstash[cur.bb]);
                    povf = ns > kStash;
                }
                const uint32_t n_new = has_prev ? uni(m0v.y) : 0u, seq_base = uni(m0v.z), fam_b = uni(m0v.w);
                uint32_t base_r[S];
#pragma unroll
                for (int L = 0; L < S; ++L) base_r[L] = n_new ? s_base[L * NWA + (int)wave] : 0u;
                K7U_LAP(3);
#pragma unroll
                for (int L = 0; L < S; ++L) {
                    const uint32_t b_r = n_new ? uni(base_r[L]) : 0u;
                    const bool reach = n_new != 0u && b_r < n_new;  // (the empty slots of the lower layers take the rest)
                    if (!reach && !(livemask >> L)) break;
                    bool take = false;
                    if (reach) {
                        // NewArm by owner pull: the r-th empty slot takes the r-th unmatched hit (src/automaton.rs:151-164)
                        const bool was_free = (wasfree >> L) & 1u;
                        const unsigned long long fmask = __ballot(was_free);
                        const uint32_t r = b_r + (uint32_t)__popcll(fmask & lt_mask);
                        take = was_free && r < n_new;
                        const PosT x = s_new[take ? r : 0u];
                        const uint64_t g_new = (uint64_t)step + cur.pend;  // aged by its own probe, then by the quiet ones
                        const uint32_t gap_new = g_new > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)g_new;
                        const bool stillborn = take && gap_new >= G;
                        if (__ballot(stillborn)) {  // (only when min_duplication_length <= k: a k-base arm is reported)
                            const bool report = stillborn && (uint64_t)k >= rp.M;
                            if (__ballot(report)) emit_records(report, (PosT)prev.i, (PosT)(prev.i + k), x, (PosT)(x + k), seq_base + r, fam_b);
                        }
                        take = take && !stillborn;
                        a_ls[L] = take ? (PosT)prev.i : a_ls[L];
                        if (take) {
                            s_cle[L * (NWA * 64) + tid] = (PosT)(prev.i + k);
                            s_crs[L * (NWA * 64) + tid] = x;
                        }
                        a_re[L] = take ? (PosT)(x + k) : a_re[L];
                        a_gap[L] = take ? gap_new : a_gap[L];
                        a_thr[L] = take ? thr0 : a_thr[L];
                        a_seq[L] = take ? seq_base + r : a_seq[L];
                        if (__ballot(take)) livemask |= 1u << L;
                    }
                    if (has_cur && (livemask & (1u << L))) {
                        if (late) {
                            c_h[L] = offers(L, a_seq[L] != kNoSeq, cur, ns, povf);
                        } else if (__ballot(take)) {
                            const uint32_t ns2 = uni(s_nstash[cur.bb]);
                            const uint32_t chn = offers(L, take, cur, ns2, ns2 > kStash);
                            c_h[L] = take ? chn : c_h[L];
                        }
                    }
                }
                K7U_LAP(4);
                mid_actions(mflags);
                {   // free counts, as the control wave will rank them in the next step
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
                    *reinterpret_cast<uint4 *>(&s_free[wave][0]) = make_uint4(nfv[0], nfv[1], nfv[2], nfv[3]);
                    if constexpr (S > 4) *reinterpret_cast<uint4 *>(&s_free[wave][4]) = make_uint4(nfv[4], nfv[5], nfv[6], nfv[7]);
                }
                K7U_LAP(5);
                if (!(flags & K7_LAST)) {  // the hits of the next step's probe (top threads), unless that step is late
                    uint32_t nflags;
                    Probe nx;
                    read_cmd(sp ^ 1u, nflags, nx);
                    if ((nflags & K7_CUR) && !(nflags & K7_LATE)) insert_hits(nx.cnt, nx.off, nx.tb, nx.bb, nx.g10);
                }
                K7U_LAP(6);
                K7T_LAP(2);
                lds_barrier();  // ---- barrier 2 --------------------------------------------------------------
                K7T_LAP(3);
                prev = cur;
                if (flags & K7_LAST) break;
            }
            if (wave == 0u) K7T_FLUSH(1);
            if (wave == 0u) K7U_FLUSH();
            if (wave == (uint32_t)(NWA - 1)) K7T_FLUSH(2);
            if (!overflow) lds_barrier();  // (3) the control wave reads the final free counts
        } else {
            // =====================================================================================================
            // CONTROL WAVE
            // =====================================================================================================
            const int c = chunk_of_uniform(rp.ch, g0);
            const uint64_t cs = rp.ch.start[c], cl = rp.ch.len[c];
            const uint32_t pb = rp.ch.pbase[c];
            const uint32_t chunk_end = rp.ch.pbase[c + 1];
            const uint32_t g_end = min(chunk_end, rp.g_hi);
            if (lane == 0) {
                s_seg[0] = g0;
                s_seg[1] = cs;
                s_seg[2] = cl;
            }
            uint32_t quiet = 0, pend = 0, fam_seq = 0, next_seq = 0, t_proc = 0, spur_until = 0;
            bool done = false, fam_open = false, giveup = false;
            uint32_t hbuf = 0;
            // ---- the batch under the cursor ---------------------------------------------------------------------
            uint32_t g = g0, nbb = 0, pos = 0, f_l = 0, rel_l = 0, tot = 0;
            unsigned long long hm = 0, qm = 0, base = 0;
            bool staged = false;  // the rows of the batch under the cursor are in s_hits[hbuf] (or on their way)
            auto load_batch = [&]() {  // -> false: a probe with more hits than the staging area
                const uint32_t nb = min(64u, g_end - g);
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
            // the next hit-probe of the segment: -> true with `nx` filled in (its quiet run folded into nx.pend); false:
            // the segment is over (done / end of the chunk or window) or not for this kernel (giveup)
            bool need_stage = false;
            auto next_probe = [&](Probe &nx, uint32_t &t_before, uint32_t &t_after) -> bool {
                for (;;) {
                    const unsigned long long hmr = pos >= 64 ? 0ull : (hm >> pos) << pos;
                    const uint32_t b = hmr ? (uint32_t)(__ffsll((long long)hmr) - 1) : 64u;
                    const unsigned long long upto = b >= 64 ? ~0ull : ((1ull << b) - 1ull);
                    const unsigned long long from = pos >= 64 ? 0ull : ~((1ull << pos) - 1ull);
                    const uint32_t q = (uint32_t)__popcll(qm & upto & from);
                    if (q) {
                        quiet += q;
                        t_proc += q;
                        pend += q * step;
                        if (quiet >= rp.tstar) {  // every arm is dead (gap >= G): the segment is over
                            done = true;
                            return false;
                        }
                    }
                    if (hmr) {
                        quiet = 0;
                        pos = b + 1;
                        if (!staged) {
                            hbuf ^= 1u;
                            need_stage = true;
                            staged = true;
                        }
                        nx.cnt = lane_of(f_l, b);
                        nx.off = hbuf * (uint32_t)HB + lane_of(rel_l, b);
                        nx.i = (uint64_t)(g + b - pb + 1) * step;
                        nx.pend = pend;
                        pend = 0;
                        t_before = t_proc;
                        t_after = ++t_proc;
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
            auto write_cmd = [&](uint32_t sp, uint32_t flags, const Probe &q) {
                if (lane == 0) {
                    *reinterpret_cast<uint4 *>(&s_cmd[sp][0]) = make_uint4(flags, q.cnt, q.off, q.tb | (q.bb << 1));
                    *reinterpret_cast<uint4 *>(&s_cmd[sp][4]) = make_uint4(q.g10, (uint32_t)q.i, (uint32_t)(q.i >> 32), q.pend);
                }
            };
            auto write_stage = [&]() {  // (lane 0) the staging request of the batch under the cursor
                *reinterpret_cast<uint4 *>(&s_mid[4]) = make_uint4((uint32_t)base, (uint32_t)(base >> 32), tot, hbuf);
            };
            // a probe gets its generation (a wrap asks for a clearing of the tables before it is indexed)
            bool need_clear = false;
            auto number_probe = [&](Probe &q) {
                if ((gen + 1u) >> kGenBits) {
                    need_clear = true;
                    gen = 0;
                }
                ++gen;
                q.g10 = gen << 10;
                q.tb = par;
                q.bb = tri;
                par ^= 1u;
                tri = (tri + 1u) % 3u;
            };

            Probe cur{0, 0, 0, 0, 0, 0, 0}, prev{0, 0, 0, 0, 0, 0, 0}, held{0, 0, 0, 0, 0, 0, 0};
            uint32_t cur_tb = 0, cur_ta = 0, prev_tb = 0, prev_ta = 0, held_tb = 0, held_ta = 0;  // processed probes before / after
            bool have_cur = false, have_held = false;
            hbuf = 1;  // (the first batch flips it to 0)
            if (!load_batch()) giveup = true;
            if (!giveup) have_cur = next_probe(cur, cur_tb, cur_ta);  // (the segment starts with a hit-probe)
            if (have_cur) number_probe(cur);
            if (lane == 0) {
                s_mid[0] = ((need_stage && have_cur) ? K7_STAGE : 0u) | ((need_clear && have_cur) ? K7_CLEAR : 0u);
                write_stage();
            }
            need_stage = need_clear = false;
            uint32_t flags = giveup ? K7_GIVEUP : (have_cur ? (K7_CUR | K7_LATE) : K7_LAST);
            write_cmd(0u, flags, cur);
            lds_barrier();  // (1)
            mid_actions(uni(s_mid[0]));
            lds_barrier();  // (2)
            if (giveup) {
                overflow = true;
            } else {
                for (uint32_t sp = 0;; sp ^= 1u) {
                    const bool have_prev = (flags & K7_PREV) != 0u;
                    K7T_MARK();
                    K7T_STEP();
                    // ------------------------------------------------------------ interval A ----------------
                    if ((flags & K7_CUR) && (flags & K7_LATE)) insert_hits(cur.cnt, cur.off, cur.tb, cur.bb, cur.g10);
                    uint32_t mflags = 0, n_new = 0, seq_base = 0;
                    if (have_prev) {
                        // empty slots as published at the end of the previous step, ranked (layer, wave, lane)
                        const uint32_t fv = lane < NE ? s_free[lane % NWA][lane / NWA] : 0u;
                        const uint32_t fv2 = NE > 64 && lane + 64 < NE ? s_free[(lane + 64) % NWA][(lane + 64) / NWA] : 0u;
                        const uint32_t h_l = min((uint32_t)lane, prev.cnt - 1u);
                        const uint32_t bv0 = s_best[prev.bb][h_l];
                        const uint8_t hf0 = use_flag ? s_hflag[prev.off + h_l] : (uint8_t)1;
                        const PosT x0 = s_hits[prev.off + h_l];
                        const uint32_t fincl = wave_incl_scan(fv);
                        uint32_t total_free = lane_of(fincl, 63u);
                        uint32_t fincl2 = 0;
                        if constexpr (NE > 64) {
                            fincl2 = wave_incl_scan(fv2) + total_free;
                            total_free = lane_of(fincl2, 63u);
                        }
                        if (lane < NE) s_base[lane] = fincl - fv;
                        if constexpr (NE > 64)
                            if (lane + 64 < NE) s_base[lane + 64] = fincl2 - fv2;
                        const uint32_t A0 = (uint32_t)CAP - total_free;  // live arms after the quiet probes' deaths
                        if (fam_open && A0 == 0 && prev_tb >= spur_until) {  // the flush of src/automaton.rs:182-200
                            ++fam_seq;
                            next_seq = 0;
                            fam_open = false;
                        }
                        // unmatched hits, in hit order (= creation order, src/automaton.rs:151-164) -> s_new[rank]
                        bool spur = false;
                        {
                            const bool in0 = (uint32_t)lane < prev.cnt;
                            const bool un0 = in0 && bv0 == kNone && hf0 != 0;
                            const unsigned long long m0 = __ballot(un0);
                            if (un0) s_new[(uint32_t)__popcll(m0 & lt_mask)] = x0;
                            n_new = (uint32_t)__popcll(m0);
                            spur = use_flag && __ballot(in0 && bv0 == kNone && hf0 == 0) != 0ull;
                        }
                        for (uint32_t h0 = 64u; h0 < prev.cnt; h0 += 64u) {  // (most probes have <= 64 hits)
                            const uint32_t h = min(h0 + (uint32_t)lane, prev.cnt - 1u);
                            const bool in = h0 + (uint32_t)lane < prev.cnt;
                            const bool un = in && s_best[prev.bb][h] == kNone;
                            const bool hf = use_flag ? s_hflag[prev.off + h] != 0 : true;
                            const unsigned long long nm = __ballot(un && hf);
                            if (use_flag) spur = spur || __ballot(un && !hf) != 0ull;
                            if (un && hf) s_new[n_new + (uint32_t)__popcll(nm & lt_mask)] = s_hits[prev.off + h];
                            n_new += (uint32_t)__popcll(nm);
                        }
                        if (n_new > total_free || A0 + n_new > cap_eff) mflags |= K7_OVF;
                        seq_base = next_seq;
                        next_seq += n_new;
                        fam_open = true;
                        if (spur) spur_until = max(spur_until, prev_ta + rp.tstar - 1u);
                    }
                    // ---- the next step --------------------------------------------------------------------------
                    Probe nx{0, 0, 0, 0, 0, 0, 0};
                    uint32_t nx_tb = 0, nx_ta = 0, nflags = 0;
                    if (have_held) {
                        // the probe that had to wait for the tables to be cleared (this step resolves its predecessor
                        // and clears them; staging, if the probe opened a batch, rides along)
                        nx = held;
                        nx_tb = held_tb;
                        nx_ta = held_ta;
                        have_held = false;
                        nflags = K7_CUR | K7_LATE;
                        mflags |= K7_CLEAR | (need_stage ? K7_STAGE : 0u);
                        need_stage = need_clear = false;
                    } else if (flags & K7_CUR) {
                        const bool have_nx = next_probe(nx, nx_tb, nx_ta);
                        if (have_nx) number_probe(nx);
                        if (giveup) mflags |= K7_OVF;  // (a later probe is not for this kernel: the segment is re-run elsewhere)
                        if (have_nx && need_clear) {
                            // generation wrap: the tables can only be cleared once the current probe has all its offers,
                            // i.e. in the NEXT step, which resolves it and has no probe of its own
                            held = nx;
                            held_tb = nx_tb;
                            held_ta = nx_ta;
                            have_held = true;
                            nx = Probe{0, 0, 0, 0, 0, 0, 0};
                            nflags = K7_PREV;
                        } else {
                            // the probe that follows a staging is indexed when its rows are there
                            nflags = K7_PREV | (have_nx ? K7_CUR : K7_LAST) | ((have_nx && need_stage) ? K7_LATE : 0u);
                            if (!have_nx) nx.pend = pend;  // the trailing quiet probes' age
                            if (need_stage && have_nx) mflags |= K7_STAGE;
                            need_stage = false;
                        }
                    } else if (!(flags & K7_LAST)) {
                        // (cannot happen: a step without a probe of its own is the last one or precedes a held probe)
                        nflags = K7_LAST;
                    }
                    if (lane == 0) {
                        *reinterpret_cast<uint4 *>(&s_mid[0]) = make_uint4(mflags, n_new, seq_base, fam_seq);
                        if (mflags & K7_STAGE) write_stage();
                        s_fam[sp] = fam_seq;
                    }
                    if (!(flags & K7_LAST)) write_cmd(sp ^ 1u, nflags, nx);
                    K7T_LAP(0);
                    lds_barrier();  // ---- barrier 1 ----------------------------------------------------------
                    K7T_LAP(1);
                    // ------------------------------------------------------------ interval B ----------------
                    if (mflags & K7_OVF) {
                        overflow = true;
                        break;
                    }
                    mid_actions(mflags);
                    // the stash of the probe after next is the previous probe's: nobody reads it any more
                    if (lane == 0 && have_prev) s_nstash[prev.bb] = 0u;
                    if (!(flags & K7_LAST) && (nflags & K7_CUR) && !(nflags & K7_LATE))
                        insert_hits(nx.cnt, nx.off, nx.tb, nx.bb, nx.g10);
                    K7T_LAP(2);
                    lds_barrier();  // ---- barrier 2 ----------------------------------------------------------
                    K7T_LAP(3);
                    if (flags & K7_LAST) break;
                    prev = cur;
                    prev_tb = cur_tb;
                    prev_ta = cur_ta;
                    cur = nx;
                    cur_tb = nx_tb;
                    cur_ta = nx_ta;
                    flags = nflags;
                }
            }
            K7T_FLUSH(0);
            if (!overflow) {
                lds_barrier();  // (3)
                // nothing alive is left behind unless the chunk (or the window of a sharded call) ended first
                const uint32_t fv = lane < NE ? s_free[lane % NWA][lane / NWA] : 0u;
                uint32_t total_free = lane_of(wave_incl_scan(fv), 63u);
                if constexpr (NE > 64) {
                    const uint32_t fv2 = lane + 64 < NE ? s_free[(lane + 64) % NWA][(lane + 64) / NWA] : 0u;
                    total_free += lane_of(wave_incl_scan(fv2), 63u);
                }
                if (fam_open && total_free == (uint32_t)CAP && t_proc >= spur_until) fam_open = false;
                if (!done && g_end < chunk_end) {
                    if (lane == 0) atomicAdd(&P.ctr[CT_RANOUT], 1ull);
                } else if (fam_open) {  // arms alive at the end of the chunk void their family (src/automaton.rs:201-203)
                    emit_records(lane == 0, (PosT)0, (PosT)0, (PosT)0, (PosT)0, kTombstone, fam_seq);
                }
            } else if (lane == 0) {
                const unsigned long long at = atomicAdd(P.ovf_count, 1ull);
                if (P.ovf_list) P.ovf_list[at] = g0;
            }
        }
        // leave no arm behind for the next segment
#pragma unroll
        for (int L = 0; L < S; ++L) a_seq[L] = kNoSeq;
        livemask = 0;
        lds_barrier();
    }
    rec_flush(rec_alloc, P, lane);
}

}  // namespace asgart
