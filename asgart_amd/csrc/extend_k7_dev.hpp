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
// The control wave plans TWO steps ahead (the command of step s+2 is written during interval B of step s, which it
// would otherwise spend waiting), so that its interval A holds nothing but the ranking; an arm wave reads one command
// block per step -- the probe, its predecessor's leftovers and the run's constants, kept in vector registers: nothing
// uniform has to survive from step to step in scalar registers -- and issues every read that does not depend on
// another one (command, winners of its candidates, their positions, its cold fields) in one go.
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

template <class PosT, int S, int NT, int HB, int kRows = 1024, int kE = 2>
__global__ __launch_bounds__(NT) void extend_k7_kernel(ExtParams<PosT> P) {
    constexpr int NW = NT / 64, NWA = NW - 1;  // waves; arm waves
    constexpr int CAP = S * NWA * 64;
    constexpr int NE = S * NWA;                // (layer, wave) entries of the free counts
    constexpr uint32_t kNone = 0xFFFFFFFFu;    // best[]: no arm accepts this hit
    constexpr uint32_t kNever = 0xFFFFFFFEu;   // what a candidate read of an idle lane returns: no creation number
    constexpr uint32_t kCoop = 0xFFFFFFFFu;    // candidate register: more than three / wide window / stash overflow
    constexpr uint32_t kStash = 64;
    constexpr uint32_t kRowsWalk = 62;  // rows of a window that an arm walks itself (beyond its first two: by the rows' occupancy bits)
    constexpr uint32_t kBitWords = (uint32_t)kRows / 32u;
    constexpr bool kWidePos = sizeof(PosT) == 8;
    constexpr uint32_t kTagShift = kWidePos ? 42u : 32u;
    constexpr uint32_t kGenMax = kWidePos ? 12u : 22u;
    constexpr unsigned long long kPosMask = (1ull << kTagShift) - 1ull;
    constexpr uint32_t kTabBytes = (uint32_t)(kRows * kE * 8);  // one hit table
    constexpr uint32_t kCmdWords = 32;  // (a command is read as one word per lane, two commands per read)
    using WinT = typename std::conditional<kWidePos, uint64_t, uint32_t>::type;
    static_assert(NW >= 2 && HB <= 1024 && S <= 8 && NE <= 128 && (kRows & (kRows - 1)) == 0 && (kE == 2 || kE == 4) && kRows <= 2048, "shape");
    if (NT >= 1024 && P.hi_prio) __builtin_amdgcn_s_setprio(3);

    __shared__ __attribute__((aligned(16))) unsigned long long s_tab[2][kRows * kE];
    __shared__ PosT s_hits[2 * HB];
    __shared__ uint8_t s_hflag[2 * HB];
    __shared__ __attribute__((aligned(16))) uint32_t s_best[3][HB];
    __shared__ unsigned long long s_stash[3][kStash];
    __shared__ uint32_t s_rowbits[3][kRows / 32];                      // per probe in flight (as the stashes): which rows of its hit table hold a hit
    __shared__ __attribute__((aligned(16))) uint32_t s_nstash[4];      // (three in use)
    __shared__ __attribute__((aligned(16))) uint32_t s_free[NWA][8];   // per (arm wave, layer): empty slots
    __shared__ __attribute__((aligned(16))) uint32_t s_base[NWA][8];   // per (arm wave, layer): rank of its first empty slot
    // The command of a step (three in rotation: the one being run, the next one -- whose hits are indexed ahead --
    // and the one the control wave is writing):
    //   0 flags          1 cnt            2 off (s_hits)    3 byte offset of its hit table
    //   4 byte offset of its best[]       5 generation tag  6,7 needle offset i
    //   8 age of the quiet probes before it                 9 stash index    10 K7_STAGE / K7_CLEAR of the step BEFORE it   11 hits of the previous probe
    //  12,13 needle offset of the previous probe           14 its rows (s_hits)     15 byte offset of its best[]
    //  16 k    17 step    18 G    19 log2 bucket width     20,21 min_duplication_length    22 threshold of a new arm
    //  23 stash / best[] buffer of the previous probe
    //  24,25 first CSR entry, 26 count, 27 buffer of the batch to stage in the step BEFORE it (when word 10 says so)
    //  28,29 processed-probe ordinals before / after the previous probe    30,31 the same of this step's probe (the control
    //  wave keeps no plan in registers: it reads its own commands back)
    __shared__ __attribute__((aligned(16))) uint32_t s_cmd[3][kCmdWords];
    __shared__ __attribute__((aligned(16))) uint32_t s_mid[8];         // decided in interval A, read in interval B
    __shared__ uint32_t s_fam[2];                                      // family ordinal as of the step (by step parity)
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
    const bool use_flag = P.hit_flag != nullptr;
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
    char *const tab0 = reinterpret_cast<char *>(&s_tab[0][0]);
    char *const best0 = reinterpret_cast<char *>(&s_best[0][0]);

    for (uint32_t n_fetch = 0; !P.max_items || n_fetch < P.max_items; ++n_fetch) {
        if (tid == 0) s_bcast = atomicAdd(P.cursor, 1ull);
        if (tid < 4) s_nstash[tid] = 0u;
        for (uint32_t j = tid; j < 3u * kBitWords; j += NT) (&s_rowbits[0][0])[j] = 0u;
        if (tid < 2) s_fam[tid] = 0u;
        for (uint32_t j = tid; j < (uint32_t)(NWA * 8); j += NT) (&s_free[0][0])[j] = 64u;
        lds_barrier();
        const unsigned long long seg = uni(s_bcast);
        lds_barrier();
        if (seg >= *P.n_seg_ptr) break;

        // ---- records (cold: what a record needs beyond the arm is read where it is written) -------------------------
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
                r.pad = 0;
                r.sd.left = seg_rev ? cs + cl - (uint64_t)ls - ll : (uint64_t)ls + cs;  // src/bin/asgart.rs:229-237
                r.sd.right = rs;
                r.sd.left_length = ll;
                r.sd.right_length = (uint64_t)re - (uint64_t)rs;
                P.recs[at] = r;
            }
        };
        // index the hits of one probe (cnt hits at s_hits[off..]) under generation tag g10 in the table at byte offset
        // tabo, winners at byte offset besto, stash bb; the TOP threads do it
        auto insert_hits = [&](uint32_t cnt, uint32_t off, uint32_t tabo, uint32_t besto, uint32_t bb, uint32_t g10, uint32_t bsh) {
            // (thread order: the last arm wave first, downwards; the control wave's lanes come last)
            const uint32_t me = tid < NWA * 64 ? (uint32_t)(NWA * 64 - 1 - tid) : (uint32_t)tid;
            for (uint32_t h = me; h < cnt; h += NT) {
                const PosT x = s_hits[off + h];
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
        // does this wave hold one of the threads that index cnt hits?
        auto indexes = [&](uint32_t cnt) {
            const uint32_t first = is_ctl ? (uint32_t)NWA : (uint32_t)(NWA - 1) - wave;  // in units of 64 hits
            return first < (cnt + 63u) / 64u;
        };
        // stage `tot` hit rows starting at CSR entry `base` into buffer `buf` (all threads)
        auto stage_rows = [&](unsigned long long base, uint32_t tot, uint32_t buf) {
            for (uint32_t r = tid; r < tot; r += NT) {
                s_hits[buf * (uint32_t)HB + r] = P.hits[base + r];
                if (use_flag) s_hflag[buf * (uint32_t)HB + r] = P.hit_flag[base + r];
            }
        };
        // ... in two halves: the rows are requested at the top of a step (the request rides in the NEXT step's command, which
        // every wave reads there) and written to LDS in its interval B -- the HBM / L2 round trip of a batch start (one
        // step in six to twelve on a tandem array) is over by then instead of sitting in front of barrier 2
        constexpr int kPf = (HB + NT - 1) / NT;
        PosT pf_x[kPf];
        uint8_t pf_f[kPf];
        auto fetch_rows = [&](unsigned long long base, uint32_t tot) {
#pragma unroll
            for (int j = 0; j < kPf; ++j) {
                const uint32_t r = (uint32_t)tid + (uint32_t)(j * NT);
                pf_x[j] = r < tot ? P.hits[base + r] : (PosT)0;
                pf_f[j] = (use_flag && r < tot) ? P.hit_flag[base + r] : (uint8_t)0;
            }
        };
        auto store_rows = [&](uint32_t tot, uint32_t buf) {
#pragma unroll
            for (int j = 0; j < kPf; ++j) {
                const uint32_t r = (uint32_t)tid + (uint32_t)(j * NT);
                if (r < tot) {
                    s_hits[buf * (uint32_t)HB + r] = pf_x[j];
                    if (use_flag) s_hflag[buf * (uint32_t)HB + r] = pf_f[j];
                }
            }
        };
        // what interval B does for everyone once the mid-step block is known (in the step loop: fetched = the rows of a
        // staging request are in this thread's registers already)
        auto mid_actions_f = [&](uint32_t mflags, uint32_t tot, uint32_t buf) {
            if (K7_RARE(mflags & K7_STAGE)) store_rows(tot, buf);
            if (K7_RARE(mflags & K7_CLEAR)) clear_table();
        };
        auto mid_actions = [&](uint32_t mflags) {
            if (K7_RARE(mflags & K7_STAGE)) {
                const uint4 m1 = *reinterpret_cast<const uint4 *>(&s_mid[4]);
                stage_rows(((unsigned long long)uni(m1.y) << 32) | uni(m1.x), uni(m1.z), uni(m1.w));
            }
            if (K7_RARE(mflags & K7_CLEAR)) clear_table();
        };

        bool overflow = false;
        if (!is_ctl) {
            // =====================================================================================================
            // ARM WAVES
            // =====================================================================================================
            lds_barrier();  // (1) the control wave has published the first batch's staging request
            mid_actions(uni(s_mid[0]));
            lds_barrier();  // (2) the rows are in; the first two commands are there
            uint32_t pv_off = 0, pv_besto = 0;  // the previous probe's rows and winners (as in the last command)
            const uint32_t *const mid_ptr = lane < 8 ? &s_mid[lane] : (lane < 16 ? &s_base[wave][lane - 8] : (lane < 20 ? &s_nstash[lane - 16] : &s_mid[0]));
            for (uint32_t sc = 0, sp = 0;; sc = sc == 2u ? 0u : sc + 1u, sp ^= 1u) {
                K7T_MARK();
                K7T_STEP();
                K7U_MARK();
                // ---- every read that depends on nothing read in this step, in one go ---------------------------
                const uint32_t sn = sc == 2u ? 0u : sc + 1u;  // the next step's command
                // (this step's command in lanes 0-31, the next step's in lanes 32-63: ONE 4-byte read per lane, the
                // fields taken out with v_readlane.  Sixteen waves that each broadcast six 16-byte words of the same
                // block to all their lanes right behind a barrier kept the LDS return path busy for ~500 cycles.)
                const uint32_t cw = s_cmd[lane < 32 ? sc : sn][lane & 31];
                auto C = [&](int j) { return (uint32_t)__builtin_amdgcn_readlane((int)cw, j); };
                auto N = [&](int j) { return (uint32_t)__builtin_amdgcn_readlane((int)cw, 32 + j); };
                // (of the arms, layer 0 -- where they sit unless a burst of a dense repeat is under way -- reads ahead;
                // the layers above read when their turn comes: five layers' worth of loads in flight would not fit
                // the register file)
                // (the winners of an arm's candidates only: the position of the hit it won is read when the winner is
                // known -- one dependent read in an interval whose length the control wave sets -- and the right start of
                // an arm when it is reported.  Seven reads per layer, step and wave right behind the barrier were the
                // larger part of the burst every wave's first round trip queued in.)
                uint32_t cb0[3] = {0, 0, 0};
                auto read_candidates = [&](int L, uint32_t (&cb)[3]) {
                    const uint32_t ch = c_h[L];
                    const uint32_t nc = (a_seq[L] == kNoSeq || ch == kCoop) ? 0u : ch >> 30;
#pragma unroll
                    for (uint32_t j = 0; j < 3; ++j) {
                        const uint32_t hj = (ch >> (10u * j)) & 1023u;
                        cb[j] = *(j < nc ? reinterpret_cast<const uint32_t *>(best0 + pv_besto + 4u * hj) : &s_never);
                    }
                };
                // (a wave that holds arms goes first wherever it shares a SIMD with waves that only index or wait)
                if (NT >= 1024 && P.hi_prio) {
                    if (livemask) __builtin_amdgcn_s_setprio(3);
                    else __builtin_amdgcn_s_setprio(0);
                }
                if (livemask & 1u) read_candidates(0, cb0);
                const uint32_t flags = C(0);
                // (the rows that this step's interval B stages: requested now)
                const uint32_t npre = (flags & (K7_LAST | K7_GIVEUP)) ? 0u : N(10);
                if (K7_RARE(npre & K7_STAGE)) fetch_rows(((unsigned long long)N(25) << 32) | N(24), N(26));
                if (K7_RARE(flags & K7_GIVEUP)) {
                    overflow = true;
                    break;
                }
                const bool has_prev = (flags & K7_PREV) != 0u, has_cur = (flags & K7_CUR) != 0u, late = (flags & K7_LATE) != 0u;
                // ---------------------------------------------------------------- interval A ----------------
                // the hits of this step's probe when it opened a batch (its rows arrived with the last barrier), and those of
                // the NEXT step's probe otherwise: indexed by the top arm waves while the others resolve
                if (K7_RARE(has_cur && late)) {
                    const uint32_t cnt = C(1);
                    if (indexes(cnt)) insert_hits(cnt, C(2), C(3), C(4), C(9), C(5), C(19));
                }
                if (K7_USUAL(!(flags & K7_LAST))) {
                    const uint32_t nflags = N(0), ncnt = N(1);
                    if (K7_RARE((nflags & K7_CUR) && !(nflags & K7_LATE) && indexes(ncnt)))
                        insert_hits(ncnt, N(2), N(3), N(4), N(9), N(5), N(19));
                }
                K7U_LAP(0);
                uint32_t wasfree = 0;  // per lane, bit L: the slot of layer L was empty before this step
                if (K7_RARE(!livemask)) {
                    wasfree = (1u << S) - 1u;
                } else {
                    const uint32_t k = C(16), step = C(17), G = C(18), pend = C(8);
                    const uint32_t p_i = C(12);  // low word of the previous probe's needle offset
                    const uint64_t M = ((uint64_t)C(21) << 32) | C(20);
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
                        uint32_t cb[3];
                        if (L == 0) {
#pragma unroll
                            for (int j = 0; j < 3; ++j) cb[j] = cb0[j];
                        } else {
                            read_candidates(L, cb);
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
                            }
                            hw = (was_free || coop) ? 0u : hw;
                            xw = s_hits[pv_off + (hw ? hw - 1u : 0u)];
                            unsigned long long sm = __ballot(coop);
                            K7C(55, __popcll(sm));
                            if (K7_RARE(sm != 0ull)) {  // more than three candidates / wide window: resolved cooperatively
                                const uint32_t p_cnt = C(11), p_off = pv_off;
                                // (many of them: one pass over the hits -- the arm that won hit h is the one whose creation
                                // number is best[h]; extend_fast_dev.hpp, phase_b)
                                if ((uint32_t)__popcll(sm) * 8u >= p_cnt) {
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
                            const bool report = dead && (uint64_t)(a_re[L] - rs_L) >= M;
                            if (__ballot(report) != 0ull)
                                emit_records(report, a_ls[L], s_cle[L * (NWA * 64) + tid], rs_L, a_re[L], a_seq[L], uni(s_fam[sp ^ 1u]));
                        }
                        a_seq[L] = dead ? kNoSeq : a_seq[L];
                    }
                }
                K7U_LAP(1);
                // ---- offers of the arms of layer L to the hits of the step's probe; only the lanes with `who` set take
                // part; returns the candidate word of the lane ---------------------------------------------------------
                auto offers = [&](int L, bool who, uint32_t ns) -> uint32_t {
                    const uint32_t k = C(16), g10 = C(5), bsh = C(19);
                    char *const tab = tab0 + C(3);
                    char *const best = best0 + C(4);
                    const bool povf = ns > kStash;
                    const WinT w_loop = (WinT)(kRowsWalk - 1u) << bsh;
                    const PosT lo = (PosT)(a_re[L] - k + 1u);
                    const WinT w = (WinT)a_thr[L] + (WinT)(k - 1u);
                    const uint32_t key = a_seq[L];
                    const bool narrow = who && w <= w_loop;
                    const WinT w_eff = narrow ? w : (WinT)0;  // (an empty window accepts nothing)
                    uint32_t ch = 0, nc = 0;
                    uint32_t *const sink = &s_sink[lane];
                    auto offer = [&](unsigned long long e, WinT wl) {
                        const uint32_t d = tag_of(e) - g10;
                        const WinT t = d < 1024u ? (WinT)(PosT)(pos_of(e) - lo) : ~(WinT)0;
                        const bool ok = t < wl;
#ifdef ASGART_K7_MASKED_MIN
                        if (ok) atomicMin(reinterpret_cast<uint32_t *>(best + 4u * (d & 1023u)), key);
#else
                        atomicMin(ok ? reinterpret_cast<uint32_t *>(best + 4u * (d & 1023u)) : sink, key);
#endif
                        ch = ok ? ((ch << 10) | d) : ch;
                        nc += ok ? 1u : 0u;
                    };
                    const uint32_t b0 = (uint32_t)((uint64_t)lo >> bsh);
                    const uint32_t n_rows = narrow ? (uint32_t)((((uint64_t)lo & ((1ull << bsh) - 1ull)) + (uint64_t)w - 1ull) >> bsh) + 1u : 0u;
                    {   // the two rows of a narrow window: all reads in flight together
                        const ulonglong2 *r0 = reinterpret_cast<const ulonglong2 *>(tab + (b0 & (uint32_t)(kRows - 1)) * (uint32_t)(kE * 8));
                        const ulonglong2 *r1 = reinterpret_cast<const ulonglong2 *>(tab + ((b0 + 1u) & (uint32_t)(kRows - 1)) * (uint32_t)(kE * 8));
                        if constexpr (kE == 4) {
                            const ulonglong2 e0 = r0[0], e1 = r0[1], e2 = r1[0], e3 = r1[1];
                            offer(e0.x, w_eff); offer(e0.y, w_eff); offer(e1.x, w_eff); offer(e1.y, w_eff);
                            offer(e2.x, w_eff); offer(e2.y, w_eff); offer(e3.x, w_eff); offer(e3.y, w_eff);
                        } else {
                            // ONE atomic for the four entries: an arm as a rule accepts one hit of a probe (the lanes that
                            // accept none go to the sink); a wave with a lane that accepts several repeats them all (min is
                            // idempotent).  Four unconditional LDS atomics per layer and wave were a fifth of the step's LDS
                            // instructions.
                            const ulonglong2 e0 = r0[0], e2 = r1[0];
                            const unsigned long long ee[4] = {e0.x, e0.y, e2.x, e2.y};
                            uint32_t first = 0;
#pragma unroll
                            for (int j = 3; j >= 0; --j) {
                                const uint32_t d = tag_of(ee[j]) - g10;
                                const WinT t = d < 1024u ? (WinT)(PosT)(pos_of(ee[j]) - lo) : ~(WinT)0;
                                const bool ok = t < w_eff;
                                first = ok ? d : first;
                            }
#pragma unroll
                            for (int j = 0; j < 4; ++j) {  // (the candidate word in entry order, as before)
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
                    }
                    // The rows behind the first two (an arm of more than ~1 kb has a window of three rows, one of 50 kb
                    // of fifty): the occupancy bits of the probe's table say which of them hold a hit at all -- a
                    // probe's hits are kilobases apart, a wide window mostly holds none -- and only those are read.
                    if (__ballot(n_rows > 2u) != 0ull) {
                        K7C(53, 1);
                        const uint32_t len = n_rows > 2u ? n_rows - 2u : 0u;                 // <= kRowsWalk - 1 < 64
                        const uint32_t s0 = (b0 + 2u) & (uint32_t)(kRows - 1);                // first of them (table row)
                        const uint32_t *const bits = &s_rowbits[C(9)][0];
                        const uint32_t w0 = s0 >> 5, sh = s0 & 31u;
                        // (two words hold the bits of up to 33 rows behind any start; the third one only for a wave with a
                        // wider window)
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
                            if constexpr (kE == 4) {
                                const ulonglong2 f0 = rr[0], f1 = rr[1];
                                offer(f0.x, wl); offer(f0.y, wl); offer(f1.x, wl); offer(f1.y, wl);
                            } else {
                                const ulonglong2 f0 = rr[0];
                                offer(f0.x, wl); offer(f0.y, wl);
                            }
                        }
                    }
                    if (K7_RARE(ns != 0u)) {
                        const uint32_t bb = C(9);
                        for (uint32_t s = 0; s < min(ns, kStash); ++s) offer(s_stash[bb][s], w_eff);
                    }
                    ch = nc > 3u ? kCoop : (ch & 0x3FFFFFFFu) | (nc << 30);
                    // arms too wide for the table walk -- and every arm when the stash overflowed
                    unsigned long long sm = __ballot(who && (!narrow || povf));
                    if (K7_RARE(sm != 0ull)) {
                        if (who && (!narrow || povf)) ch = kCoop;
                        K7C(54, __popcll(sm));
                        const uint32_t cnt = C(1), off = C(2);
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
                            for (uint32_t h0 = 0; h0 < cnt; h0 += 64u) {
                                const uint32_t h = h0 + (uint32_t)lane;
                                if (h < cnt && (WinT)(PosT)(s_hits[off + h] - lo_u) < w_u)
                                    atomicMin(reinterpret_cast<uint32_t *>(best + 4u * h), key_u);
                            }
                        }
                    }
                    return ch;
                };
                K7U_LAP(2);
                K7T_LAP(0);
                lds_barrier();  // ---- barrier 1 --------------------------------------------------------------
                K7T_LAP(1);
                K7U_MARK();
                // ---------------------------------------------------------------- interval B ----------------
                const bool had_live = livemask != 0u;
                // (lanes 0-7: the mid-step block; 8-15: this wave's slot ranks; 16-19: the stash counts)
                const uint32_t mw = *mid_ptr;
                auto MW = [&](uint32_t j) { return (uint32_t)__builtin_amdgcn_readlane((int)mw, (int)j); };
                const uint32_t mflags = MW(0);
                if (K7_RARE(mflags & K7_OVF)) {
                    overflow = true;
                    break;
                }
                const uint32_t n_new = has_prev ? MW(1) : 0u;
                const uint32_t seq_base = MW(2), fam_b = MW(3);
                K7U_LAP(3);
                const bool receives = n_new != 0u && MW(8) < n_new;
                if (receives || (has_cur && livemask)) {
                    // the stash count of the step's probe (final when its hits are indexed)
                    const uint32_t ns_b = has_cur ? MW(16u + C(9)) : 0u;
                    const uint32_t k = C(16), step = C(17), G = C(18), pend = C(8), thr0 = C(22);
                    const uint64_t M = ((uint64_t)C(21) << 32) | C(20);
                    PosT p_i;
                    if constexpr (kWidePos) p_i = (PosT)(((uint64_t)C(13) << 32) | C(12));
                    else p_i = (PosT)C(12);
#pragma unroll
                    for (int L = 0; L < S; ++L) {
                        const uint32_t b_r = n_new ? MW(8u + (uint32_t)L) : 0u;
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
                            const uint64_t g_new = (uint64_t)step + pend;  // aged by its own probe, then by the quiet ones
                            const uint32_t gap_new = g_new > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)g_new;
                            const bool stillborn = take && gap_new >= G;
                            if (K7_RARE(__ballot(stillborn) != 0ull)) {  // (only when min_duplication_length <= k: a k-base arm is reported)
                                const bool report = stillborn && (uint64_t)k >= M;
                                if (__ballot(report)) emit_records(report, p_i, (PosT)(p_i + k), x, (PosT)(x + k), seq_base + r, fam_b);
                            }
                            take = take && !stillborn;
                            a_ls[L] = take ? p_i : a_ls[L];
                            if (take) {
                                s_cle[L * (NWA * 64) + tid] = (PosT)(p_i + k);
                                s_crs[L * (NWA * 64) + tid] = x;
                            }
                            a_re[L] = take ? (PosT)(x + k) : a_re[L];
                            a_gap[L] = take ? gap_new : a_gap[L];
                            a_thr[L] = take ? thr0 : a_thr[L];
                            a_seq[L] = take ? seq_base + r : a_seq[L];
                            if (__ballot(take)) livemask |= 1u << L;
                        }
                        // every arm of the layer, old and new, offers to the step's probe
                        if (has_cur && (livemask & (1u << L))) c_h[L] = offers(L, a_seq[L] != kNoSeq, ns_b);
                    }
                }
                K7U_LAP(4);
                mid_actions_f(mflags, N(26), N(27));
                if (had_live || receives) {  // free counts, as the control wave will rank them in the next step (an idle
                                             // wave's stay as they are: all empty)
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
                K7U_LAP(6);
                K7T_LAP(2);
                lds_barrier();  // ---- barrier 2 --------------------------------------------------------------
                K7T_LAP(3);
                pv_off = C(2);
                pv_besto = C(4);
                if (K7_RARE(flags & K7_LAST)) break;
            }
            if (wave == 0u) K7T_FLUSH(1);
            if (wave == 0u) K7U_FLUSH();
            if (wave == (uint32_t)(NWA - 1)) K7T_FLUSH(2);
            if (!overflow) lds_barrier();  // (3) the control wave reads the final free counts
        } else {
            // =====================================================================================================
            // CONTROL WAVE
            // =====================================================================================================
            const RunParams &rp = P.rp;
            const uint32_t k = (uint32_t)rp.k, step = (uint32_t)rp.step, G = rp.G;
            const uint32_t thr0 = arm_threshold(k, G);
            uint32_t bsh = 3;
            while ((1ull << bsh) < (unsigned long long)G + k) ++bsh;
            bsh += P.fast_bsh;
            const uint32_t kGenBits = min(kGenMax, max(2u, P.gen_bits));
            const uint32_t cap_eff = min((uint32_t)CAP, P.cap_limit);
            const uint32_t g0 = P.seg_list[seg];
            if (lane == 0) {
            heartbeat(P, g0, 0u);
            seg_clock(P);
        }
#ifdef ASGART_PROFILE_EXTEND
            const unsigned long long k7_seg0 = __builtin_amdgcn_s_memtime();
#endif
            const int c = chunk_of_uniform(rp.ch, g0);
            const uint64_t cs = rp.ch.start[c], cl = rp.ch.len[c];
            const uint32_t pb = rp.ch.pbase[c];
            const uint32_t chunk_end = rp.ch.pbase[c + 1];
            const uint32_t g_end = min(chunk_end, rp.g_hi);
            if (lane == 0) {
                s_seg[0] = g0;
                s_seg[1] = cs;
                s_seg[2] = cl | ((unsigned long long)((rp.mode_of(c) >> 1) & 1u) << 63);
            }
            uint32_t quiet = 0, pend = 0, fam_seq = 0, next_seq = 0, t_proc = 0, spur_until = 0;
            bool done = false, fam_open = false, giveup = false;
            uint32_t hbuf = 1;  // (the first batch flips it to 0)
            // ---- the batch under the cursor ---------------------------------------------------------------------
            uint32_t g = g0, nbb = 0, pos = 0, f_l = 0, rel_l = 0, tot = 0;
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
            // One planned step: its command and what has to happen in the interval B of the step BEFORE it (staging of
            // the batch its probe opens, clearing of the tables at a generation wrap).
            struct Probe {
                uint32_t cnt, off, tb, bb, g10, pend, t_before, t_after;
                uint64_t i;
            };
            struct Plan {
                uint32_t flags, pre;           // command flags; K7_STAGE / K7_CLEAR of the step before
                unsigned long long st_base;    // staging request
                uint32_t st_tot, st_buf;
                Probe q;
            };
            const Probe no_probe{0, 0, 0, 0, 0, 0, 0, 0, 0};
            // the next hit-probe of the segment: -> true with `nx` filled in (its quiet run folded into nx.pend); false:
            // the segment is over (done / end of the chunk or window) or not for this kernel (giveup)
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
                        opened = false;
                        if (!staged) {
                            hbuf ^= 1u;
                            opened = true;
                            staged = true;
                        }
                        nx.cnt = lane_of(f_l, b);
                        nx.off = hbuf * (uint32_t)HB + lane_of(rel_l, b);
                        nx.i = (uint64_t)(g + b - pb + 1) * step;
                        nx.pend = pend;
                        pend = 0;
                        nx.t_before = t_proc;
                        nx.t_after = ++t_proc;
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
            // a probe gets its generation -> true: the tables must be cleared before it is indexed (wrap)
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
            // the plan of the step behind `last` (whose probe, if it has one, becomes the new step's predecessor)
            bool have_held = false;
            Plan held{};
            auto advance = [&](uint32_t last_flags) -> Plan {
                Plan n{};
                n.q = no_probe;
                if (have_held) {  // the probe that waited for the tables to be cleared
                    n = held;
                    have_held = false;
                    return n;
                }
                Probe nx = no_probe;
                const bool found = next_probe(nx);
                if (giveup) {
                    n.flags = K7_GIVEUP;
                    return n;
                }
                if (!found) {
                    n.flags = K7_PREV | K7_LAST;
                    n.q.pend = pend;  // the trailing quiet probes' age
                    return n;
                }
                const bool wrap = number_probe(nx);
                Plan p{};
                p.q = nx;
                p.pre = (opened ? K7_STAGE : 0u) | (wrap ? K7_CLEAR : 0u);
                p.st_base = base;
                p.st_tot = tot;
                p.st_buf = hbuf;
                p.flags = K7_CUR | ((opened || wrap) ? K7_LATE : 0u);
                if (wrap && (last_flags & K7_CUR)) {
                    // the tables can only be cleared once the previous probe has all its offers: a step without a probe
                    // of its own resolves that one first, the clearing rides in ITS interval B
                    held = p;
                    have_held = true;
                    n.flags = K7_PREV;
                    return n;
                }
                p.flags |= (last_flags & K7_CUR) ? K7_PREV : 0u;
                return p;
            };
            // (the predecessor's fields as the control wave reads them back from ITS command: the plans of the steps in
            // flight live in the command ring only -- three of them kept in registers were 48 scalar registers, most of them
            // spilled, and their rotation 570 cycles at the end of every step)
            struct Before {
                uint32_t cnt, i_lo, i_hi, off, bb, t_before, t_after;
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
                    o[7] = make_uint4(b.t_before, b.t_after, p.q.t_before, p.q.t_after);
                }
            };
            auto before_of = [&](const Plan &p) {
                return Before{p.q.cnt, (uint32_t)p.q.i, (uint32_t)(p.q.i >> 32), p.q.off, p.q.bb, p.q.t_before, p.q.t_after};
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
            if (giveup) {
                p_cur.flags = K7_GIVEUP;
            } else {
                p_cur = advance(0u);  // (the segment starts with a hit-probe: a late step, its batch staged below)
            }
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
            const bool giveup0 = (p_cur.flags & K7_GIVEUP) != 0u;
            lds_barrier();  // (1)
            mid_actions(uni(s_mid[0]));
            lds_barrier();  // (2)
            if (giveup0) {
                overflow = true;
            } else {
                for (uint32_t sc = 0, sp = 0;; sc = sc == 2u ? 0u : sc + 1u, sp ^= 1u) {
                    // this step's plan (lanes 0-31) and the next one's (32-63), as written two steps / one step ago
                    const uint32_t sn = sc == 2u ? 0u : sc + 1u;
                    const uint32_t cw = s_cmd[lane < 32 ? sc : sn][lane & 31];
                    auto C = [&](int j) { return (uint32_t)__builtin_amdgcn_readlane((int)cw, j); };
                    auto N = [&](int j) { return (uint32_t)__builtin_amdgcn_readlane((int)cw, 32 + j); };
                    const uint32_t flags = C(0);
                    if (flags & K7_GIVEUP) {  // (the arm waves leave when they read it)
                        overflow = true;
                        break;
                    }
                    const bool have_prev = (flags & K7_PREV) != 0u;
                    const bool more = !(flags & K7_LAST);
                    const uint32_t nflags = more ? N(0) : 0u;
                    const uint32_t prev_cnt = C(11), prev_off = C(14), prev_bb = C(23), prev_t_before = C(28), prev_t_after = C(29);
                    K7T_MARK();
                    K7T_STEP();
                    // ------------------------------------------------------------ interval A ----------------
                    if ((flags & K7_CUR) && (flags & K7_LATE) && indexes(C(1))) insert_hits(C(1), C(2), C(3), C(4), C(9), C(5), bsh);
                    if (more && (nflags & K7_CUR) && !(nflags & K7_LATE) && indexes(N(1))) insert_hits(N(1), N(2), N(3), N(4), N(9), N(5), bsh);
                    // what rides in this step's interval B was decided when the NEXT step was planned
                    uint32_t mflags = more ? N(10) : 0u;
                    const unsigned long long st_base = ((unsigned long long)N(25) << 32) | N(24);
                    const uint32_t st_tot = N(26), st_buf = N(27);
                    if (mflags & K7_STAGE) fetch_rows(st_base, st_tot);
                    uint32_t n_new = 0, seq_base = 0;
                    if (have_prev) {
                        // empty slots as published at the end of the previous step, ranked (layer, wave, lane)
                        const uint32_t fv = lane < NE ? s_free[lane % NWA][lane / NWA] : 0u;
                        const uint32_t fv2 = NE > 64 && lane + 64 < NE ? s_free[(lane + 64) % NWA][(lane + 64) / NWA] : 0u;
                        const uint32_t h_l = min((uint32_t)lane, prev_cnt - 1u);
                        const uint32_t bv0 = s_best[prev_bb][h_l];
                        const uint8_t hf0 = use_flag ? s_hflag[prev_off + h_l] : (uint8_t)1;
                        const PosT x0 = s_hits[prev_off + h_l];
                        // (the second and third group of 64 hits -- a tandem array has 80-200 hits per probe -- requested
                        // together with the first: the groups behind them go one round trip at a time)
                        uint32_t bv1 = kNone, bv2 = kNone;
                        uint8_t hf1 = 1, hf2 = 1;
                        PosT x1 = 0, x2 = 0;
                        if (prev_cnt > 64u) {
                            const uint32_t h1 = min(64u + (uint32_t)lane, prev_cnt - 1u), h2 = min(128u + (uint32_t)lane, prev_cnt - 1u);
                            bv1 = s_best[prev_bb][h1];
                            bv2 = s_best[prev_bb][h2];
                            x1 = s_hits[prev_off + h1];
                            x2 = s_hits[prev_off + h2];
                            if (use_flag) {
                                hf1 = s_hflag[prev_off + h1];
                                hf2 = s_hflag[prev_off + h2];
                            }
                        }
                        // (one scan for both halves of the entries: two 16-bit fields, each total < 2^16)
                        const uint32_t packed = wave_incl_scan(fv | (fv2 << 16));
                        const uint32_t fincl = packed & 0xFFFFu;
                        const uint32_t tot_p = lane_of(packed, 63u);
                        uint32_t total_free = tot_p & 0xFFFFu;
                        uint32_t fincl2 = 0;
                        if constexpr (NE > 64) {
                            fincl2 = (packed >> 16) + total_free;
                            total_free += tot_p >> 16;
                        }
                        if (lane < NE) s_base[lane % NWA][lane / NWA] = fincl - fv;
                        if constexpr (NE > 64)
                            if (lane + 64 < NE) s_base[(lane + 64) % NWA][(lane + 64) / NWA] = fincl2 - fv2;
                        const uint32_t A0 = (uint32_t)CAP - total_free;  // live arms after the quiet probes' deaths
#ifdef ASGART_PROFILE_EXTEND
                        k7_sum_a += A0;
                        k7_sum_cnt += prev_cnt;
                        ++k7_sum_n;
#endif
                        if (fam_open && A0 == 0 && prev_t_before >= spur_until) {  // the flush of src/automaton.rs:182-200
                            ++fam_seq;
                            next_seq = 0;
                            fam_open = false;
                        }
                        // unmatched hits, in hit order (= creation order, src/automaton.rs:151-164) -> s_new[rank]
                        bool spur = false;
                        auto rank_group = [&](uint32_t h0, uint32_t bv, uint8_t hf, PosT x) {
                            const bool in = h0 + (uint32_t)lane < prev_cnt;
                            const bool un = in && bv == kNone && hf != 0;
                            const unsigned long long m = __ballot(un);
                            if (un) s_new[n_new + (uint32_t)__popcll(m & lt_mask)] = x;
                            n_new += (uint32_t)__popcll(m);
                            if (use_flag) spur = spur || __ballot(in && bv == kNone && hf == 0) != 0ull;
                        };
                        rank_group(0u, bv0, hf0, x0);
                        if (prev_cnt > 64u) {
                            rank_group(64u, bv1, hf1, x1);
                            if (prev_cnt > 128u) rank_group(128u, bv2, hf2, x2);
                        }
                        for (uint32_t h0 = 192u; h0 < prev_cnt; h0 += 64u) {
                            const uint32_t h = min(h0 + (uint32_t)lane, prev_cnt - 1u);
                            rank_group(h0, s_best[prev_bb][h], use_flag ? s_hflag[prev_off + h] : (uint8_t)1, s_hits[prev_off + h]);
                        }
                        if (n_new > total_free || A0 + n_new > cap_eff) mflags |= K7_OVF;
                        seq_base = next_seq;
                        next_seq += n_new;
                        fam_open = true;
                        if (spur) spur_until = max(spur_until, prev_t_after + rp.tstar - 1u);
                    }
                    if (lane == 0) {
                        *reinterpret_cast<uint4 *>(&s_mid[0]) = make_uint4(mflags, n_new, seq_base, fam_seq);
                        if (mflags & K7_STAGE)
                            *reinterpret_cast<uint4 *>(&s_mid[4]) = make_uint4((uint32_t)st_base, (uint32_t)(st_base >> 32), st_tot, st_buf);
                        s_fam[sp] = fam_seq;
                    }
                    K7T_LAP(0);
                    lds_barrier();  // ---- barrier 1 ----------------------------------------------------------
                    K7T_LAP(1);
                    // ------------------------------------------------------------ interval B ----------------
                    if (mflags & K7_OVF) {
                        overflow = true;
                        break;
                    }
                    mid_actions_f(mflags, st_tot, st_buf);
                    // the stash of the probe after next is the previous probe's: nobody reads it any more
                    if (lane == 0 && have_prev) s_nstash[prev_bb] = 0u;
                    if (have_prev && (uint32_t)lane < kBitWords) s_rowbits[prev_bb][lane] = 0u;  // (kRows <= 2048: one word per lane)
                    // the step after next: planned now, while the arm waves create and offer
                    if (more && !(nflags & (K7_LAST | K7_GIVEUP))) {
                        const Plan p_after = advance(nflags);
                        write_cmd(sc == 0u ? 2u : sc - 1u, p_after, Before{N(1), N(6), N(7), N(2), N(9), N(30), N(31)});  // (slot of step + 2 = slot of step - 1)
                    }
                    K7T_LAP(2);
                    lds_barrier();  // ---- barrier 2 ----------------------------------------------------------
                    K7T_LAP(3);
                    if (!more) break;
                }
            }
            K7T_FLUSH(0);
#ifdef ASGART_PROFILE_EXTEND
            if (lane == 0) {  // (the dump's "sumA", "sumCnt", "lds_probes"; "longest": the segment that took longest)
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
    wg_busy(P);
}

}  // namespace asgart
