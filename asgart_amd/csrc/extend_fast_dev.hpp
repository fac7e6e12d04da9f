// extend_fast_dev.hpp -- "K6": the two-barrier arm-resident extension kernel.
//
// Same automaton as K4 / K4b / K4c (reference src/automaton.rs:57-204, representation of
// pipeline_dev.hpp: only live arms are kept, winners by creation number, families by records),
// organised so that one processed hit-probe costs the workgroup TWO barrier intervals with
// loop-free lookups, and a run of quiet probes costs nothing until the next hit-probe:
//
//   table   the hits of one probe sit INLINE in a bucket table: row = (x >> bsh) mod kRows,
//           four 64-bit entries (generation | hit index | position) per row.  An insert is an
//           exchange that pushes what it displaces one entry on; a fifth hit of a row goes to a
//           small stash.  Consecutive buckets are consecutive rows, so a tandem array spreads
//           perfectly, and a narrow arm's window (< 2^bsh wide) is covered by TWO rows = four
//           16-byte LDS reads issued together: no chains, no loops.
//   phase A every live arm (kept in its owner's registers, S per thread) first takes the age of
//           the quiet probes since the last hit-probe (and retires if that kills it), then reads
//           its rows and offers atomicMin(best[h], creation number) to the hits inside its
//           window; it remembers up to three candidates (hit indices packed in one register).
//           Each wave publishes how many empty slots it has per layer.             | barrier
//   phase B the arm re-reads best[] of its candidates: equal to its own number = ExtendArm
//           (src/automaton.rs:133-150), else it ages / retires (:166-171).  Unmatched hits
//           become arms (:151-164) by OWNER PULL: empty slots are ranked (layer, wave, lane)
//           from the published counts (one DPP scan), unmatched hits are ranked by ballots that
//           every wave recomputes from best[], and the r-th empty slot takes the r-th unmatched
//           hit -- no mailbox, no free list, no atomics, and the arms stay packed in the low
//           waves.  Meanwhile the TOP threads index the next hit-probe's hits.      | barrier
//   Arms wider than kRowsLoop rows, probes whose stash overflowed and arms with more than three
//   candidates take a wave-cooperative path (one arm at a time against 64 hits per step).
// Results are identical to the other extension kernels (tests force every segment through it).
#pragma once

#include "pipeline_dev.hpp"

namespace asgart {

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

// position of the n-th (0-based) set bit of m; n < popcount(m)
__device__ inline uint32_t select_bit(unsigned long long m, uint32_t n) {
    uint32_t word = (uint32_t)m, pos = 0;
    uint32_t c = (uint32_t)__popc(word);
    if (n >= c) {
        n -= c;
        word = (uint32_t)(m >> 32);
        pos = 32;
    }
#pragma unroll
    for (int sh = 16; sh >= 1; sh >>= 1) {
        c = (uint32_t)__popc(word & ((1u << sh) - 1u));
        if (n >= c) {
            n -= c;
            word >>= sh;
            pos += (uint32_t)sh;
        }
    }
    return pos;
}

template <class PosT, int S, int NT, int HB, int kRows = 1024>
__global__ __launch_bounds__(NT) void extend_fast_kernel(ExtParams<PosT> P) {
    constexpr int CAP = S * NT;
    constexpr int NW = NT / 64;
    constexpr uint32_t kNone = 0xFFFFFFFFu;    // best[]: no arm accepts this hit
    // candidate register of an arm: up to three hit indices, 10 bits each, count in bits 30..31;
    // kCoop: more than three, a window too wide for the table walk, or a probe whose stash overflowed
    constexpr uint32_t kCoop = 0xFFFFFFFFu;
    constexpr uint32_t kStash = 64;
    constexpr uint32_t kRowsLoop = 6;          // windows of up to this many rows are looked up by the arm itself
    constexpr bool kWidePos = sizeof(PosT) == 8;
    // entry: 32-bit positions  [gen:22 | hit:10 | x:32];  64-bit positions  [gen:12 | hit:10 | x:42]
    constexpr uint32_t kTagShift = kWidePos ? 42u : 32u;
    constexpr uint32_t kGenMax = kWidePos ? 12u : 22u;
    constexpr unsigned long long kPosMask = (1ull << kTagShift) - 1ull;
    using WinT = typename std::conditional<kWidePos, uint64_t, uint32_t>::type;
    static_assert(HB <= 1024 && S * NW <= 64 && (kRows & (kRows - 1)) == 0, "shape");
    if (NT >= 1024 && P.hi_prio) __builtin_amdgcn_s_setprio(3);

    __shared__ __attribute__((aligned(16))) unsigned long long s_tab[kRows * 4];
    __shared__ PosT s_hits[HB];
    __shared__ uint8_t s_hflag[HB];
    __shared__ uint32_t s_best[2][HB];
    __shared__ unsigned long long s_stash[2][kStash];
    __shared__ uint32_t s_nstash[2];
    __shared__ uint32_t s_free[64];                       // per (layer, wave): empty slots
    __shared__ unsigned long long s_newmask[NW][HB / 64]; // per wave: unmatched hits of each group of 64
    __shared__ unsigned long long s_bcast;

    const int tid = threadIdx.x, lane = tid & 63;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const RunParams &rp = P.rp;
    const uint64_t n_seg = *P.n_seg_ptr;
    const uint32_t k = (uint32_t)rp.k, step = (uint32_t)rp.step, G = rp.G;
    const uint32_t thr0 = arm_threshold(k, G);
    uint32_t bsh = 3;  // bucket width 2^bsh >= G + k: a narrow arm's window spans at most two rows
    while ((1ull << bsh) < (unsigned long long)G + k) ++bsh;
    const uint32_t kGenBits = min(kGenMax, max(2u, P.gen_bits));
    const uint32_t cap_eff = min((uint32_t)CAP, P.cap_limit);
    const WinT w_loop = (WinT)(kRowsLoop - 1u) << bsh;  // windows up to this width span <= kRowsLoop rows
    RecAlloc rec_alloc;
    PROF_DECL;

    PosT a_ls[S], a_le[S], a_rs[S], a_re[S];
    uint32_t a_thr[S], a_gap[S], a_seq[S], c_h[S];
#pragma unroll
    for (int L = 0; L < S; ++L) {
        a_seq[L] = kNoSeq;
        a_ls[L] = a_le[L] = a_rs[L] = a_re[L] = 0;
        a_thr[L] = a_gap[L] = 0;
        c_h[L] = 0;
    }
    auto clear_table = [&]() {
        for (uint32_t e = tid; e < (uint32_t)(kRows * 4); e += NT) s_tab[e] = 0ull;
    };
    clear_table();
    if (tid < 2) s_nstash[tid] = 0u;
    uint32_t gen = 0, par = 0;
    lds_barrier();

    for (uint32_t n_fetch = 0; !P.max_items || n_fetch < P.max_items; ++n_fetch) {
        if (tid == 0) s_bcast = atomicAdd(P.cursor, 1ull);
        if (tid < 2) s_nstash[tid] = 0u;  // (a probe indexed ahead but never reached may have left entries)
        lds_barrier();
        const unsigned long long seg = uni(s_bcast);
        lds_barrier();
        if (seg >= n_seg) break;
        const uint32_t g0 = P.seg_list[seg];
        PROF_SEG_BEGIN();
        const int c = chunk_of_uniform(rp.ch, g0);
        const uint64_t cs = rp.ch.start[c], cl = rp.ch.len[c];
        const uint32_t pb = rp.ch.pbase[c];
        const uint32_t chunk_end = rp.ch.pbase[c + 1];
        const uint32_t g_end = min(chunk_end, rp.g_hi);
        // block-uniform bookkeeping
        uint32_t quiet = 0, pend = 0, fam_seq = 0, next_seq = 0, t_proc = 0, spur_until = 0;
        bool overflow = false, done = false, fam_open = false;

        auto emit_records = [&](bool emit, PosT ls, PosT le, PosT rs, PosT re, uint32_t seq) {
            const unsigned long long em = __ballot(emit);
            if (!em) return;
            const unsigned long long at = rec_slot(rec_alloc, P, em, lane);
            if (emit && at < P.rec_cap) {
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
        auto tag_of = [&](unsigned long long e) { return (uint32_t)(e >> kTagShift); };
        auto pos_of = [&](unsigned long long e) { return (PosT)(e & kPosMask); };
        // index the hits of one probe (cnt hits at s_hits[off..]) under generation `gen`; the TOP threads do it
        auto insert_hits = [&](uint32_t cnt, uint32_t off, uint32_t bp) {
            const uint32_t g10 = gen << 10;
            for (uint32_t h = (uint32_t)(NT - 1 - tid); h < cnt; h += NT) {
                const PosT x = s_hits[off + h];
                s_best[bp][h] = kNone;
                unsigned long long e = ((unsigned long long)(g10 | h) << kTagShift) | ((unsigned long long)x & kPosMask);
                unsigned long long *row = &s_tab[(((uint32_t)((uint64_t)x >> bsh)) & (uint32_t)(kRows - 1)) * 4u];
                bool placed = false;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (!placed) {
                        const unsigned long long old = atomicExch(&row[j], e);
                        if (tag_of(old) - g10 >= 1024u) placed = true;  // displaced a stale entry: done
                        else e = old;                                   // a hit of this probe: it moves on
                    }
                }
                if (!placed) {
                    const uint32_t at = atomicAdd(&s_nstash[bp], 1u);
                    if (at < kStash) s_stash[bp][at] = e;
                }
            }
        };
        // wave-cooperative window walk for ONE arm (lo, w, key wave-uniform): offers to every hit inside
        auto coop_offer = [&](PosT lo, WinT w, uint32_t key, uint32_t cnt, uint32_t off) {
            for (uint32_t h0 = 0; h0 < cnt; h0 += 64u) {
                const uint32_t h = h0 + (uint32_t)lane;
                if (h < cnt && (WinT)(PosT)(s_hits[off + h] - lo) < w) atomicMin(&s_best[par][h], key);
            }
        };
        // ... and the last hit (SA order) inside the window whose winner is `key`; returns its index or kNone
        auto coop_resolve = [&](PosT lo, WinT w, uint32_t key, uint32_t cnt, uint32_t off, PosT &x_out) {
            uint32_t hmax = kNone;
            for (uint32_t h0 = 0; h0 < cnt; h0 += 64u) {
                const uint32_t h = h0 + (uint32_t)lane;
                PosT x = 0;
                bool ok = false;
                if (h < cnt) {
                    x = s_hits[off + h];
                    ok = (WinT)(PosT)(x - lo) < w && s_best[par][h] == key;
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

        // ---- phase A: pending age, lookups, published free counts ------------------------------
        bool povf = false;
        auto phase_a = [&](uint32_t cnt, uint32_t off, bool lookup) {
            const uint32_t g10 = gen << 10;
            const uint32_t ns = lookup ? uni(s_nstash[par]) : 0u;
            povf = ns > kStash;
#pragma unroll
            for (int L = 0; L < S; ++L) {
                bool live = a_seq[L] != kNoSeq;
                if (__ballot(live)) {
                    if (pend) {  // the quiet probes since the last hit-probe: src/automaton.rs:166-171
                        bool dead = false;
                        if (live) {
                            const uint64_t sum_g = (uint64_t)a_gap[L] + pend;
                            a_gap[L] = sum_g > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)sum_g;
                            dead = a_gap[L] >= G;
                        }
                        if (__ballot(dead)) {
                            const bool report = dead && (uint64_t)(a_re[L] - a_rs[L]) >= rp.M;
                            emit_records(report, a_ls[L], a_le[L], a_rs[L], a_re[L], a_seq[L]);
                            if (dead) {
                                a_seq[L] = kNoSeq;
                                live = false;
                            }
                        }
                    }
                    if (lookup) {
                        const PosT lo = (PosT)(a_re[L] - k + 1u);
                        const WinT w = (WinT)a_thr[L] + (WinT)(k - 1u);
                        const uint32_t key = a_seq[L];
                        uint32_t ch = 0, nc = 0;
                        auto offer = [&](unsigned long long e) {
                            const uint32_t d = tag_of(e) - g10;
                            if (d < 1024u && (WinT)(PosT)(pos_of(e) - lo) < w) {
                                atomicMin(&s_best[par][d], key);
                                ch = (ch << 10) | d;
                                ++nc;
                            }
                        };
                        const bool narrow = live && w <= w_loop;
                        if (narrow) {
                            const uint32_t b0 = (uint32_t)((uint64_t)lo >> bsh);
                            const uint32_t n_rows = (uint32_t)((((uint64_t)lo & ((1ull << bsh) - 1ull)) + (uint64_t)w - 1ull) >> bsh) + 1u;
                            const ulonglong2 *r0 = reinterpret_cast<const ulonglong2 *>(&s_tab[(b0 & (uint32_t)(kRows - 1)) * 4u]);
                            const ulonglong2 *r1 = reinterpret_cast<const ulonglong2 *>(&s_tab[((b0 + 1u) & (uint32_t)(kRows - 1)) * 4u]);
                            const ulonglong2 e0 = r0[0], e1 = r0[1], e2 = r1[0], e3 = r1[1];
                            offer(e0.x); offer(e0.y); offer(e1.x); offer(e1.y);
                            offer(e2.x); offer(e2.y); offer(e3.x); offer(e3.y);
                            for (uint32_t r = 2; r < n_rows; ++r) {
                                const ulonglong2 *rr = reinterpret_cast<const ulonglong2 *>(&s_tab[((b0 + r) & (uint32_t)(kRows - 1)) * 4u]);
                                const ulonglong2 f0 = rr[0], f1 = rr[1];
                                offer(f0.x); offer(f0.y); offer(f1.x); offer(f1.y);
                            }
                            for (uint32_t s = 0; s < min(ns, kStash); ++s) offer(s_stash[par][s]);
                            ch = nc > 3u ? kCoop : (ch & 0x3FFFFFFFu) | (nc << 30);
                        }
                        // arms too wide for the table walk -- and every arm when the stash overflowed
                        unsigned long long sm = __ballot(live && (!narrow || povf));
                        if (sm) {
                            if (live && (!narrow || povf)) ch = kCoop;
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
                                coop_offer(lo_u, w_u, lane_of(key, l), cnt, off);
                            }
                        }
                        c_h[L] = ch;
                    }
                }
                const uint32_t nf = (uint32_t)__popcll(__ballot(a_seq[L] == kNoSeq));
                if (lane == 0) s_free[L * NW + (int)wave] = nf;
            }
            if (tid == 0) s_nstash[par ^ 1u] = 0u;  // the next probe's stash (filled after the barrier)
        };

        for (uint32_t g = g0; g < g_end && !done;) {
            // ---- stage a batch of up to 64 probes (every wave computes the same masks) ----
            PROF_START();
            const uint32_t nb = min(64u, g_end - g);
            const uint32_t f_l = (uint32_t)lane < nb ? P.p_filt[g + lane] : kSkipN;
            const unsigned long long r_l = (uint32_t)lane < nb ? P.row_off[g + lane] : 0ull;
            const unsigned long long r_hi = uni(P.row_off[g + nb]);
            const unsigned long long base = lane_of(r_l, 0u);
            unsigned long long r_next = __shfl_down(r_l, 1);
            if ((uint32_t)lane + 1 >= nb) r_next = r_hi;
            const bool fits = (uint32_t)lane < nb && r_next - base <= (unsigned long long)HB;
            const unsigned long long fm = __ballot(fits);
            uint32_t nbb = (~fm == 0ull) ? 64u : (uint32_t)(__ffsll((long long)~fm) - 1);
            if (nbb > nb) nbb = nb;
            if (nbb == 0) {  // one probe with more hits than the staging area: not for this kernel
                overflow = true;
                break;
            }
            const uint32_t rel_l = (uint32_t)(r_l - base);
            {
                const unsigned long long end = nbb == nb ? r_hi : lane_of(r_l, nbb);
                const uint32_t tot = (uint32_t)(end - base);
                for (uint32_t r = tid; r < tot; r += NT) {
                    s_hits[r] = P.hits[base + r];
                    if (P.hit_flag) s_hflag[r] = P.hit_flag[base + r];
                }
            }
            lds_barrier();
            const unsigned long long in_batch = nbb >= 64 ? ~0ull : ((1ull << nbb) - 1ull);
            const unsigned long long hm = __ballot(f_l >= 1u && f_l < kPending) & in_batch;
            const unsigned long long qm = __ballot(f_l == 0u) & in_batch;
            PROF_STOP(0);
            PROF_COUNT(1, 1);
            uint32_t pos = 0;
            bool pre_indexed = false;
            while (!done) {
                const unsigned long long hmr = pos >= 64 ? 0ull : (hm >> pos) << pos;
                if (!hmr) break;
                const uint32_t b = (uint32_t)(__ffsll((long long)hmr) - 1);
                {
                    const unsigned long long range = ((1ull << b) - 1ull) & ~((1ull << pos) - 1ull);
                    const uint32_t q = (uint32_t)__popcll(qm & range);
                    if (q) {  // quiet probes only age: folded into the next pass over the arms
                        quiet += q;
                        t_proc += q;
                        pend += q * step;
                        if (quiet >= rp.tstar) {  // every arm is dead (gap >= G): the segment is over
                            done = true;
                            break;
                        }
                    }
                }
                quiet = 0;
                pos = b + 1;
                const uint32_t cnt = lane_of(f_l, b);
                const uint32_t off = lane_of(rel_l, b);
                const uint64_t i = (uint64_t)(g + b - pb + 1) * step;
                const uint32_t t_before = t_proc;
                ++t_proc;
                PROF_COUNT(5, 1);
                PROF_COUNT(11, cnt);
                PROF_START();
                if (!pre_indexed) {
                    if (++gen >> kGenBits) {  // generation wrap: clear the table once
                        lds_barrier();
                        clear_table();
                        gen = 1;
                        lds_barrier();
                    }
                    insert_hits(cnt, off, par);
                    lds_barrier();
                }
                pre_indexed = false;
                PROF_STOP(2);
                PROF_START();
                phase_a(cnt, off, true);
                pend = 0;
                PROF_STOP(4);
                PROF_START();
                lds_barrier();
                PROF_STOP(3);
                PROF_START();
                // ---- phase B --------------------------------------------------------------------
                // empty slots, ranked (layer, wave, lane)
                const uint32_t fv = lane < S * NW ? s_free[lane] : 0u;
                const uint32_t fincl = wave_incl_scan(fv);
                const uint32_t total_free = lane_of(fincl, (uint32_t)(S * NW - 1));
                const uint32_t A0 = (uint32_t)CAP - total_free;  // live arms after the quiet probes' deaths
                PROF_COUNT(10, A0);
                if (fam_open && A0 == 0 && t_before >= spur_until) {  // the flush of src/automaton.rs:182-200
                    ++fam_seq;
                    next_seq = 0;
                    fam_open = false;
                }
                // unmatched hits, in hit order (= creation order, src/automaton.rs:151-164)
                uint32_t n_new = 0;
                bool spur = false;
                unsigned long long m0 = 0;  // group 0 stays in registers (most probes have <= 64 hits)
                for (uint32_t h0 = 0, gi = 0; h0 < cnt; h0 += 64u, ++gi) {
                    const uint32_t h = h0 + (uint32_t)lane;
                    bool un = false, hf = true;
                    if (h < cnt) {
                        un = s_best[par][h] == kNone;
                        if (P.hit_flag) hf = s_hflag[off + h] != 0;
                    }
                    const unsigned long long nm = __ballot(un && hf);
                    if (P.hit_flag) spur = spur || __ballot(un && !hf) != 0ull;
                    if (gi == 0) m0 = nm;
                    else if (lane == 0) s_newmask[wave][gi] = nm;
                    n_new += (uint32_t)__popcll(nm);
                }
                PROF_MAX(9, A0 + n_new);
                if (n_new > total_free || A0 + n_new > cap_eff) {  // (identical in every wave)
                    overflow = true;
                    done = true;
                    break;
                }
                const uint32_t seq_base = next_seq;
                PROF_STOP(6);
                PROF_START();
                // next hit probe of this staged batch, if any: the top threads index it in this interval
                // (nothing below reads the table; best[] and the stash are double-buffered)
                const unsigned long long nxt = pos >= 64 ? 0ull : (hm >> pos) << pos;
                const bool can_pre = nxt != 0ull && ((gen + 1u) >> kGenBits) == 0u;
#pragma unroll
                for (int L = 0; L < S; ++L) {
                    const bool was_free = a_seq[L] == kNoSeq;
                    const unsigned long long fmask = __ballot(was_free);
                    if (~fmask) {  // some lane holds an arm
                        const PosT lo = (PosT)(a_re[L] - k + 1u);
                        const WinT w = (WinT)a_thr[L] + (WinT)(k - 1u);
                        // the last hit (SA order) this arm won, if any: src/automaton.rs:133-150 apply in hit order
                        uint32_t hwon = kNone;
                        if (!was_free && c_h[L] != kCoop) {
                            const uint32_t ch = c_h[L], nc = ch >> 30;
#pragma unroll
                            for (uint32_t j = 0; j < 3; ++j) {
                                const uint32_t hj = (ch >> (10u * j)) & 1023u;
                                if (j < nc && s_best[par][hj] == a_seq[L] && (hwon == kNone || hj > hwon)) hwon = hj;
                            }
                        }
                        bool won = hwon != kNone;
                        PosT xw = 0;
                        if (won) xw = s_hits[off + hwon];
                        unsigned long long sm = __ballot(!was_free && c_h[L] == kCoop);
                        while (sm) {  // several candidates / wide window: the last hit it won, cooperatively
                            const uint32_t l = (uint32_t)(__ffsll((long long)sm) - 1);
                            sm &= sm - 1ull;
                            PosT lo_u, x_u = 0;
                            WinT w_u;
                            if constexpr (kWidePos) {
                                lo_u = (PosT)lane_of((unsigned long long)lo, l);
                                w_u = (WinT)lane_of((unsigned long long)w, l);
                            } else {
                                lo_u = (PosT)lane_of((uint32_t)lo, l);
                                w_u = (WinT)lane_of((uint32_t)w, l);
                            }
                            const uint32_t hm_ = coop_resolve(lo_u, w_u, lane_of(a_seq[L], l), cnt, off, x_u);
                            if ((uint32_t)lane == l && hm_ != kNone) {
                                won = true;
                                xw = x_u;
                            }
                        }
                        bool dead = false;
                        if (!was_free) {
                            if (won) {  // ExtendArm, src/automaton.rs:133-150
                                a_re[L] = (PosT)(xw + k);
                                a_le[L] = (PosT)(i + k);
                                a_thr[L] = arm_threshold((uint64_t)(i + k) - (uint64_t)a_ls[L], G);
                                a_gap[L] = 0;
                            } else {
                                const uint64_t sum_g = (uint64_t)a_gap[L] + step;
                                a_gap[L] = sum_g > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)sum_g;
                                dead = a_gap[L] >= G;  // src/automaton.rs:166-171: never matches again
                            }
                        }
                        if (__ballot(dead)) {
                            const bool report = dead && (uint64_t)(a_re[L] - a_rs[L]) >= rp.M;
                            emit_records(report, a_ls[L], a_le[L], a_rs[L], a_re[L], a_seq[L]);
                            if (dead) a_seq[L] = kNoSeq;
                        }
                    }
                    // NewArm by owner pull: the r-th empty slot takes the r-th unmatched hit
                    if (n_new && fmask) {
                        const uint32_t base_r = lane_of(fincl - fv, (uint32_t)(L * NW) + wave);
                        if (base_r < n_new) {
                            const uint32_t r = base_r + (uint32_t)__popcll(fmask & lt_mask);
                            const bool take = was_free && r < n_new;
                            uint32_t hsel = 0;
                            if (cnt <= 64u) {
                                if (take) hsel = select_bit(m0, r);
                            } else {
                                uint32_t pfx = 0;
                                for (uint32_t h0 = 0, gi = 0; h0 < cnt; h0 += 64u, ++gi) {
                                    const unsigned long long nm = gi == 0 ? m0 : uni(s_newmask[wave][gi]);
                                    const uint32_t cg = (uint32_t)__popcll(nm);
                                    if (take && r >= pfx && r < pfx + cg) hsel = h0 + select_bit(nm, r - pfx);
                                    pfx += cg;
                                }
                            }
                            if (take) {  // src/automaton.rs:151-164 (aged by this very probe)
                                const PosT x = s_hits[off + hsel];
                                a_ls[L] = (PosT)i;
                                a_le[L] = (PosT)(i + k);
                                a_rs[L] = x;
                                a_re[L] = (PosT)(x + k);
                                a_gap[L] = step;
                                a_thr[L] = thr0;
                                a_seq[L] = seq_base + r;
                            }
                        }
                    }
                }
                next_seq += n_new;
                fam_open = true;
                if (spur) spur_until = max(spur_until, t_proc + rp.tstar - 1u);
                PROF_STOP(7);
                PROF_START();
                if (can_pre) {
                    const uint32_t nb2 = (uint32_t)(__ffsll((long long)nxt) - 1);
                    ++gen;
                    insert_hits(lane_of(f_l, nb2), lane_of(rel_l, nb2), par ^ 1u);
                }
                pre_indexed = can_pre;
                lds_barrier();
                par ^= 1u;
                PROF_STOP(8);
            }
            if (overflow) break;
            if (!done) {
                const unsigned long long range = pos >= 64 ? 0ull : ~((1ull << pos) - 1ull);
                const uint32_t q = (uint32_t)__popcll(qm & range);
                if (q) {
                    quiet += q;
                    t_proc += q;
                    pend += q * step;
                    if (quiet >= rp.tstar) done = true;
                }
            }
            g += nbb;
        }
        if (!overflow) {
            // the age of the trailing quiet probes: whatever it kills is reported, and the family closes if
            // nothing is left
            phase_a(0, 0, false);
            pend = 0;
            lds_barrier();
            const uint32_t fv = lane < S * NW ? s_free[lane] : 0u;
            const uint32_t total_free = lane_of(wave_incl_scan(fv), (uint32_t)(S * NW - 1));
            if (fam_open && total_free == (uint32_t)CAP && t_proc >= spur_until) fam_open = false;
            if (!done && g_end < chunk_end) {
                if (tid == 0) atomicAdd(&P.ctr[CT_RANOUT], 1ull);
            } else if (fam_open) {  // arms alive at the end of the chunk void their family (src/automaton.rs:201-203)
                emit_records(tid == 0, (PosT)0, (PosT)0, (PosT)0, (PosT)0, kTombstone);
            }
        } else if (tid == 0) {
            const unsigned long long at = atomicAdd(P.ovf_count, 1ull);
            if (P.ovf_list) P.ovf_list[at] = g0;
        }
        // leave no arm behind for the next segment
#pragma unroll
        for (int L = 0; L < S; ++L) a_seq[L] = kNoSeq;
        if (tid < 64) {
            PROF_FLUSH();
        }
        lds_barrier();
    }
    rec_flush(rec_alloc, P, lane);
}

}  // namespace asgart
