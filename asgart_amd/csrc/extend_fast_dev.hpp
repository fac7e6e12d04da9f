// extend_fast_dev.hpp -- "K6": the one-barrier arm-resident extension kernel.
//
// Same automaton as the other extension kernels (reference src/automaton.rs:57-204, representation of
// pipeline_dev.hpp: only live arms are kept, winners by creation number, families by records),
// organised so that one processed hit-probe costs the workgroup ONE barrier, with loop-free
// lookups and branch-free per-arm code, and a run of quiet probes costs nothing until the next
// hit-probe:
//
//   table   the hits of one probe sit INLINE in a bucket table: row = (x >> bsh) mod kRows, kE
//           64-bit entries (generation | hit index | position) per row.  An insert is an exchange
//           that pushes what it displaces one entry on; a hit that finds its row full goes to a
//           small stash.  Consecutive buckets are consecutive rows, so a tandem array spreads
//           perfectly, and an arm's window (up to 2^bsh wide) is covered by TWO rows = 16-byte
//           LDS reads issued together: no chains, no loops.
//   "A"     every live arm (kept in its owner's registers, S per thread) first takes the age of
//           the quiet probes since the last hit-probe (and retires if that kills it), then reads
//           its rows and offers atomicMin(best[h], creation number) to the hits inside its
//           window; it remembers up to three candidates (hit indices packed in one register).
//           Each wave publishes how many empty slots it has per layer.
//   "B"     the arm re-reads best[] of its candidates: equal to its own number = ExtendArm
//           (src/automaton.rs:133-150), else it ages / retires (:166-171).  Unmatched hits
//           become arms (:151-164) by OWNER PULL: empty slots are ranked (layer, wave, lane)
//           from the published counts (one DPP scan), unmatched hits are ranked by ballots that
//           every wave recomputes from best[], and the r-th empty slot takes the r-th unmatched
//           hit -- no mailbox, no free list, no atomics, and the arms stay packed in the low
//           waves.
//   Software pipeline: between two barriers a wave runs B of probe t-1, then A of probe t, and the
//   TOP threads index the hits of probe t+1 (tables double-buffered, best[] and the stash triple-
//   buffered): the only thing probe t's B waits for is that every wave has finished offering to
//   probe t's hits -- one barrier per hit-probe, and the waves with little to do (no arms, or
//   only the indexing) absorb the skew of the others.
//   The per-arm code has no divergent branches: selects, unconditional LDS operations aimed at a
//   per-lane sink / a never-matching word when a lane has nothing to do (a lone wave issues
//   dependent vector instructions back to back, but every exec-mask update costs it a round trip
//   through the scalar unit; tools/ubench_isa.hip).
//   Arms wider than kRowsLoop rows, probes whose stash overflowed and arms with more than three
//   candidates take a wave-cooperative path (one arm at a time against 64 hits per step).
// Results are identical to the other extension kernels (tests force every segment through it).
#pragma once

#include "pipeline_dev.hpp"

namespace asgart {

// minimum across the 64 lanes of a wave (same DPP steps as the scan; the result is wave-uniform)
__device__ inline uint32_t wave_min_u32(uint32_t x) {
#define ASGART_DPP_MIN(ctrl, rows) \
    x = min(x, (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)x, ctrl, rows, 0xf, false))
    ASGART_DPP_MIN(0x111, 0xf);
    ASGART_DPP_MIN(0x112, 0xf);
    ASGART_DPP_MIN(0x114, 0xf);
    ASGART_DPP_MIN(0x118, 0xf);
    ASGART_DPP_MIN(0x142, 0xa);
    ASGART_DPP_MIN(0x143, 0xc);
#undef ASGART_DPP_MIN
    return (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
}

// position of the n-th (0-based) set bit of m; n < popcount(m)
__device__ inline uint32_t select_bit(unsigned long long m, uint32_t n) {
    uint32_t word = (uint32_t)m, pos = 0;
    uint32_t c = (uint32_t)__popc(word);
    const bool up = n >= c;
    n = up ? n - c : n;
    word = up ? (uint32_t)(m >> 32) : word;
    pos = up ? 32u : 0u;
#pragma unroll
    for (int sh = 16; sh >= 1; sh >>= 1) {
        c = (uint32_t)__popc(word & ((1u << sh) - 1u));
        const bool u2 = n >= c;
        n = u2 ? n - c : n;
        word = u2 ? word >> sh : word;
        pos = u2 ? pos + (uint32_t)sh : pos;
    }
    return pos;
}

template <class PosT, int S, int NT, int HB, int kRows = 1024, int kE = 4, bool kPipe = true>
__global__ __launch_bounds__(NT, (NT == 512 || NT == 256) ? 4 : 1) void extend_fast_kernel(ExtParams<PosT> P) {  // (256 / 512 threads: four waves per SIMD, so that four / two workgroups share a compute unit)
    constexpr int CAP = S * NT;
    constexpr int NW = NT / 64;
    constexpr uint32_t kNone = 0xFFFFFFFFu;    // best[]: no arm accepts this hit
    constexpr uint32_t kNever = 0xFFFFFFFEu;   // what a candidate read of an idle lane returns: no creation number
    // candidate register of an arm: up to three hit indices, 10 bits each, count in bits 30..31;
    // kCoop: more than three, a window too wide for the table walk, or a probe whose stash overflowed
    constexpr uint32_t kCoop = 0xFFFFFFFFu;
    constexpr uint32_t kStash = 64;
    constexpr uint32_t kRowsLoop = 62;         // windows of up to this many rows are looked up by the arm itself (beyond the first two rows:
                                               // those whose occupancy bit is set)
    constexpr uint32_t kBitWords = (uint32_t)kRows / 32u;
    constexpr bool kWidePos = sizeof(PosT) == 8;
    // entry: 32-bit positions  [gen:22 | hit:10 | x:32];  64-bit positions  [gen:12 | hit:10 | x:42]
    constexpr uint32_t kTagShift = kWidePos ? 42u : 32u;
    constexpr uint32_t kGenMax = kWidePos ? 12u : 22u;
    constexpr unsigned long long kPosMask = (1ull << kTagShift) - 1ull;
    using WinT = typename std::conditional<kWidePos, uint64_t, uint32_t>::type;
    static_assert(HB <= 1024 && S <= 8 && S * NW <= 128 && (kRows & (kRows - 1)) == 0 && (kE == 2 || kE == 4) && kRows >= 128 && kRows <= 2048, "shape");
    if (NT >= 1024) __builtin_amdgcn_s_setprio(3);

    __shared__ __attribute__((aligned(16))) unsigned long long s_tab[2][kRows * kE];
    __shared__ PosT s_hits[HB];
    __shared__ uint32_t s_best[3][HB];
    __shared__ unsigned long long s_stash[3][kStash];
    __shared__ uint32_t s_nstash[3];
    __shared__ uint32_t s_rowbits[3][kRows / 32];  // per probe in flight (as the stashes): which rows of its hit table hold a hit
    __shared__ __attribute__((aligned(16))) uint32_t s_free[2][NW][8];  // per (wave, layer): empty slots
    __shared__ unsigned long long s_newmask[NW][HB / 64]; // per wave: unmatched hits of each group of 64
    __shared__ unsigned long long s_bcast;
    // cold fields of an arm, by slot (layer * NT + thread): the left end is written at every extension and read only
    // when the arm is reported; the right start is read when the arm dies.  Out of the registers they buy a fifth layer.
    __shared__ PosT s_cle[CAP], s_crs[CAP];
    // Bookkeeping of the automaton that every wave keeps in scalar registers (creation counter, family ordinal, open
    // family): wave 0 publishes it after each B; a wave that skipped the hit ranking of a probe (it could
    // not be concerned: see phase B) re-reads it before the next B it takes part in.
    __shared__ __attribute__((aligned(16))) uint32_t s_pub[2][4];
    // after a run of probes that wave 0 went through alone (see solo_probe): position in the batch, probes processed,
    // live arms; alternating
    __shared__ __attribute__((aligned(16))) uint32_t s_run[2][4];
    const uint32_t kSoloHits = min(48u, P.solo_hits);  // option solo (0: never; 1: 16 hits)
    // arms that move to wave 0's first layer when few are left (migrate below): their fields in transit
    constexpr uint32_t kMigMax = 64;
    __shared__ PosT s_x_ls[kMigMax], s_x_re[kMigMax], s_x_le[kMigMax], s_x_rs[kMigMax];
    __shared__ uint32_t s_x_thr[kMigMax], s_x_gap[kMigMax], s_x_seq[kMigMax];
    __shared__ uint32_t s_mig[NW][8];  // per (wave, layer): live arms, for the migration
    __shared__ uint32_t s_sink[64];  // per lane: where the atomicMin of a lane with nothing to offer goes
    __shared__ uint32_t s_never;     // == kNever

    const int tid = threadIdx.x, lane = tid & 63;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const RunParams &rp = P.rp;
    const uint64_t n_seg = *P.n_seg_ptr;
    const uint32_t k = (uint32_t)rp.k, step = (uint32_t)rp.step, G = rp.G;
    const uint32_t thr0 = arm_threshold(k, G);
    // bucket width: the smallest power of two >= G + k (a young arm's window spans at most two rows)
    uint32_t bsh = 3;
    while ((1ull << bsh) < (unsigned long long)G + k) ++bsh;
    const uint32_t kGenBits = min(kGenMax, max(2u, P.gen_bits));
    const uint32_t cap_eff = min((uint32_t)CAP, P.cap_limit);
    const WinT w_loop = (WinT)(kRowsLoop - 1u) << bsh;  // windows up to this width span <= kRowsLoop rows
    RecAlloc rec_alloc;
    wg_begin(P);
    PROF_DECL;

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
    auto clear_table = [&]() {
        for (uint32_t e = tid; e < (uint32_t)(2 * kRows * kE); e += NT) (&s_tab[0][0])[e] = 0ull;
    };
    clear_table();
    if (tid == 0) s_never = kNever;
    for (uint32_t j = tid; j < (uint32_t)(2 * NW * 8); j += NT) (&s_free[0][0][0])[j] = 64u;
    uint32_t gen = 0, par = 0, tri = 0;  // generation; parity (table, free counts); best[] / stash buffer
    lds_barrier();

    for (;;) {
        if (tid == 0) s_bcast = atomicAdd(P.cursor, 1ull);
        if (tid < 3) s_nstash[tid] = 0u;  // (a probe indexed ahead but never reached may have left entries)
        for (uint32_t j = tid; j < 3u * kBitWords; j += NT) (&s_rowbits[0][0])[j] = 0u;
        lds_barrier();
        const unsigned long long seg = uni(s_bcast);
        lds_barrier();
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
        // block-uniform bookkeeping
        uint32_t quiet = 0, pend = 0, fam_seq = 0, next_seq = 0;
        bool overflow = false, done = false, fam_open = false;
        bool ran_full = true;  // this wave took part in the previous probe's hit ranking (its bookkeeping is current)
        // "solo" probes: while every live arm sits in wave 0's first layer and a probe's few hits cannot overfill it,
        // wave 0 runs the probe alone, in registers (solo_probe below): solo_w0 / solo_a are what every wave knows
        bool solo_w0 = true;   // every live arm is in (wave 0, layer 0)
        uint32_t solo_a = 0;   // upper bound on the live arms
        uint32_t solo_par = 0;
        bool want_migrate = false;  // few arms are left and some sit outside (wave 0, layer 0): see migrate

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
                r.sd.left = seg_rev ? cs + cl - (uint64_t)ls - ll : (uint64_t)ls + cs;  // src/bin/asgart.rs:229-237
                r.sd.right = rs;
                r.sd.left_length = ll;
                r.sd.right_length = (uint64_t)re - (uint64_t)rs;
                P.recs[at] = r;
            }
        };
        auto tag_of = [&](unsigned long long e) { return (uint32_t)(e >> kTagShift); };
        auto pos_of = [&](unsigned long long e) { return (PosT)(e & kPosMask); };
        // index the hits of one probe (cnt hits at s_hits[off..]) under generation `gen` in table tb, with
        // best[] / stash buffer bb; the TOP threads do it
        auto insert_hits = [&](uint32_t cnt, uint32_t off, uint32_t tb, uint32_t bb) {
            const uint32_t g10 = gen << 10;
            for (uint32_t h = (uint32_t)(NT - 1 - tid); h < cnt; h += NT) {
                const PosT x = s_hits[off + h];
                s_best[bb][h] = kNone;
                unsigned long long e = ((unsigned long long)(g10 | h) << kTagShift) | ((unsigned long long)x & kPosMask);
                const uint32_t ri = ((uint32_t)((uint64_t)x >> bsh)) & (uint32_t)(kRows - 1);
                unsigned long long *row = &s_tab[tb][ri * (uint32_t)kE];
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
        // wave-cooperative window walk for ONE arm (lo, w, key wave-uniform): offers to every hit inside
        auto coop_offer = [&](PosT lo, WinT w, uint32_t key, uint32_t cnt, uint32_t off, uint32_t bb) {
            for (uint32_t h0 = 0; h0 < cnt; h0 += 64u) {
                const uint32_t h = h0 + (uint32_t)lane;
                if (h < cnt && (WinT)(PosT)(s_hits[off + h] - lo) < w) atomicMin(&s_best[bb][h], key);
            }
        };
        // ... and the last hit (SA order) inside the window whose winner is `key`; returns its index or kNone
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

        // ---- A: pending age, lookups in table tb / best[] bb, published free counts (buffer tb) ----
        auto phase_a = [&](uint32_t cnt, uint32_t off, bool lookup, uint32_t tb, uint32_t bb) {
            const uint32_t g10 = gen << 10;
            // (requested first, used after the first rows have been tested: the round trip hides behind them)
            const uint32_t ns_v = (lookup && livemask) ? s_nstash[bb] : 0u;
            uint32_t ns = 0;
            bool ns_known = false, povf = false;
            uint32_t nfv[8] = {64u, 64u, 64u, 64u, 64u, 64u, 64u, 64u};
#pragma unroll
            for (int L = 0; L < S; ++L) {
                if (!(livemask >> L)) break;  // no arm of this wave in this layer or a higher one (nfv stays 64)
                uint32_t nf = 64u;
                if (livemask & (1u << L)) {
                    bool live = a_seq[L] != kNoSeq;
                    if (pend) {  // the quiet probes since the last hit-probe: src/automaton.rs:166-171
                        const uint64_t sum_g = (uint64_t)a_gap[L] + pend;
                        const uint32_t aged = sum_g > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)sum_g;
                        const bool dead = live && aged >= G;
                        a_gap[L] = aged;
                        if (__ballot(dead)) {
                            const PosT rs = s_crs[L * NT + tid];
                            const bool report = dead && (uint64_t)(a_re[L] - rs) >= rp.M;
                            if (__ballot(report)) emit_records(report, a_ls[L], s_cle[L * NT + tid], rs, a_re[L], a_seq[L]);
                        }
                        a_seq[L] = dead ? kNoSeq : a_seq[L];
                        live = live && !dead;
                    }
                    if (lookup) {
                        const PosT lo = (PosT)(a_re[L] - k + 1u);
                        const WinT w = (WinT)a_thr[L] + (WinT)(k - 1u);
                        const uint32_t key = a_seq[L];
                        const bool narrow = live && w <= w_loop;
                        const WinT w_eff = narrow ? w : (WinT)0;  // (an empty window accepts nothing)
                        uint32_t ch = 0, nc = 0;
                        uint32_t *const sink = &s_sink[lane];
                        auto offer_w = [&](unsigned long long e, WinT wl) {
                            const uint32_t d = tag_of(e) - g10;
                            const WinT t = d < 1024u ? (WinT)(PosT)(pos_of(e) - lo) : ~(WinT)0;  // (no scalar AND of two masks)
                            const bool ok = t < wl;
                            atomicMin(ok ? &s_best[bb][d & 1023u] : sink, key);
                            ch = ok ? ((ch << 10) | d) : ch;
                            nc += ok ? 1u : 0u;
                        };
                        auto offer = [&](unsigned long long e) { offer_w(e, w_eff); };
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
                        if (!ns_known) {
                            ns = uni(ns_v);
                            povf = ns > kStash;
                            ns_known = true;
                        }
                        // The rows behind the first two (an arm of more than ~1 kb has a window of three rows, one of
                        // 50 kb of fifty): the occupancy bits of the probe's table say which of them hold a hit at all --
                        // a probe's hits are kilobases apart, a wide window mostly holds none -- and only those are read.
                        if (__ballot(n_rows > 2u) != 0ull) {
                            const uint32_t len = n_rows > 2u ? n_rows - 2u : 0u;   // <= kRowsLoop - 2 < 64
                            const uint32_t s0 = (b0 + 2u) & (uint32_t)(kRows - 1);  // first of them (table row)
                            const uint32_t *const bits = &s_rowbits[bb][0];
                            const uint32_t w0 = s0 >> 5, sh = s0 & 31u;
                            const uint32_t v0 = bits[w0], v1 = bits[(w0 + 1u) & (kBitWords - 1u)], v2 = bits[(w0 + 2u) & (kBitWords - 1u)];
                            unsigned long long m = ((((unsigned long long)v1 << 32) | v0) >> sh) | (sh ? (unsigned long long)v2 << (64u - sh) : 0ull);
                            m &= (1ull << len) - 1ull;
                            while (__ballot(m != 0ull) != 0ull) {
                                const bool act = m != 0ull;
                                const uint32_t r = act ? (uint32_t)(__ffsll((long long)m) - 1) : 0u;
                                m &= m - 1ull;
                                const ulonglong2 *rr = reinterpret_cast<const ulonglong2 *>(&s_tab[tb][((s0 + r) & (uint32_t)(kRows - 1)) * (uint32_t)kE]);
                                const WinT wl = act ? w_eff : (WinT)0;
                                if constexpr (kE == 4) {
                                    const ulonglong2 f0 = rr[0], f1 = rr[1];
                                    offer_w(f0.x, wl); offer_w(f0.y, wl); offer_w(f1.x, wl); offer_w(f1.y, wl);
                                } else {
                                    const ulonglong2 f0 = rr[0];
                                    offer_w(f0.x, wl); offer_w(f0.y, wl);
                                }
                            }
                        }
                        for (uint32_t s = 0; s < min(ns, kStash); ++s) offer(s_stash[bb][s]);
                        ch = nc > 3u ? kCoop : (ch & 0x3FFFFFFFu) | (nc << 30);
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
                                coop_offer(lo_u, w_u, lane_of(key, l), cnt, off, bb);
                            }
                        }
                        c_h[L] = ch;
                    }
                    nf = (uint32_t)__popcll(__ballot(a_seq[L] == kNoSeq));
                    if (nf == 64u) livemask &= ~(1u << L);
                }
                nfv[L] = nf;
            }
            // (every lane, same 16 bytes)
            *reinterpret_cast<uint4 *>(&s_free[tb][wave][0]) = make_uint4(nfv[0], nfv[1], nfv[2], nfv[3]);
            if constexpr (S > 4) *reinterpret_cast<uint4 *>(&s_free[tb][wave][4]) = make_uint4(nfv[4], nfv[5], nfv[6], nfv[7]);
        };

        // one hit-probe in flight: the probe whose offers have been made and whose B is still to run
        struct Probe {
            uint32_t cnt, off, tb, bb;
            uint64_t i;
        };
        // ---- B of probe q: resolve, age / retire, create --------------------------------------------------
        auto phase_b = [&](const Probe &q) {
            PROF_START();
            const uint32_t cnt = q.cnt, off = q.off, bb = q.bb;
            const uint64_t i = q.i;
            // every independent read first: free counts, the first 64 hits' winners, the candidates' winners
            // (entry j = layer j / NW, wave j % NW; a lane holds entries j = lane and lane + 64)
            const uint32_t fv = lane < S * NW ? s_free[q.tb][lane % NW][lane / NW] : 0u;
            const uint32_t fv2 = S * NW > 64 && lane + 64 < S * NW ? s_free[q.tb][(lane + 64) % NW][(lane + 64) / NW] : 0u;
            // empty slots, ranked (layer, wave, lane)
            const uint32_t fincl = wave_incl_scan(fv);
            uint32_t total_free = lane_of(fincl, 63u);
            uint32_t fincl2 = 0;
            if constexpr (S * NW > 64) {
                fincl2 = wave_incl_scan(fv2) + total_free;
                total_free = lane_of(fincl2, 63u);
            }
            const uint32_t A0 = (uint32_t)CAP - total_free;  // live arms after the quiet probes' deaths
            PROF_COUNT(10, A0);
            // Which waves rank the hits.  The new arms of this probe -- at most cnt -- go to the lowest-ranked empty
            // slots; a wave whose first slot ranks behind cnt others cannot receive one, and if cnt arms fit, nothing
            // can overflow: such a wave only resolves its own arms below.  Wave 0 always ranks (and publishes).
            const uint32_t base0 = lane_of(fincl - fv, wave);  // rank of this wave's first layer-0 slot
            const bool full = wave == 0u || !(base0 >= cnt && cnt <= total_free && A0 + cnt <= cap_eff);
            uint32_t n_new = 0;
            unsigned long long m0 = 0;
            if (full) {
                if (!ran_full) {  // bookkeeping as of the end of the previous probe
                    const uint4 pv = *reinterpret_cast<const uint4 *>(&s_pub[q.tb ^ 1u][0]);
                    next_seq = uni(pv.x);
                    fam_seq = uni(pv.y);
                    fam_open = (uni(pv.z) & 1u) != 0u;
                }
                if (fam_open && A0 == 0) {  // the flush of src/automaton.rs:182-200
                    ++fam_seq;
                    next_seq = 0;
                    fam_open = false;
                }
                // unmatched hits, in hit order (= creation order, src/automaton.rs:151-164)
                const uint32_t h_l = min((uint32_t)lane, cnt - 1u);
                const uint32_t bv0 = s_best[bb][h_l];
                const bool in0 = (uint32_t)lane < cnt;
                m0 = __ballot(in0 && bv0 == kNone);  // group 0 stays in registers
                n_new = (uint32_t)__popcll(m0);
                for (uint32_t h0 = 64u, gi = 1; h0 < cnt; h0 += 64u, ++gi) {  // (most probes have <= 64 hits)
                    const uint32_t h = min(h0 + (uint32_t)lane, cnt - 1u);
                    const bool in = h0 + (uint32_t)lane < cnt;
                    const bool un = in && s_best[bb][h] == kNone;
                    const unsigned long long nm = __ballot(un);
                    s_newmask[wave][gi] = nm;
                    n_new += (uint32_t)__popcll(nm);
                }
                PROF_MAX(9, A0 + n_new);
                if (n_new > total_free || A0 + n_new > cap_eff) {  // (identical in every wave that gets here; the
                    overflow = true;                                //  others have checked that cnt arms fit)
                    done = true;
                }
            }
            // (an overflow is the same decision in every wave: the waves that skipped the ranking know that cnt arms fit)
            if (overflow) return;
            const uint32_t seq_base = next_seq;
            PROF_STOP(6);
            PROF_START();
#pragma unroll
            for (int L = 0; L < S; ++L) {
                if (!(livemask >> L)) {  // no arm of this wave here or above: go on only while new arms can reach the layer
                    const uint32_t before = L == 0 ? 0u : (L * NW <= 64 ? lane_of(fincl, (uint32_t)(L * NW - 1))
                                                                         : lane_of(fincl2, (uint32_t)(L * NW - 65)));
                    if (!full || before >= n_new) break;  // (the empty slots of the lower layers take them all)
                }
                const bool was_free = a_seq[L] == kNoSeq;
                const unsigned long long fmask = __ballot(was_free);
                if (livemask & (1u << L)) {  // this wave may hold an arm here
                    // the last hit (SA order) this arm won, if any: src/automaton.rs:133-150 apply in hit order
                    const uint32_t ch = c_h[L];
                    const bool coop = !was_free && ch == kCoop;
                    const uint32_t nc = (was_free || coop) ? 0u : ch >> 30;
                    uint32_t cb[3];  // the three reads in flight together; a lane without a candidate reads kNever
#pragma unroll
                    for (uint32_t j = 0; j < 3; ++j) cb[j] = *(j < nc ? &s_best[bb][(ch >> (10u * j)) & 1023u] : &s_never);
                    uint32_t hw = 0;  // 1 + that hit
#pragma unroll
                    for (uint32_t j = 0; j < 3; ++j) {
                        const uint32_t hj = (ch >> (10u * j)) & 1023u;
                        hw = cb[j] == a_seq[L] ? max(hw, hj + 1u) : hw;
                    }
                    hw = (was_free || coop) ? 0u : hw;
                    PosT xw = s_hits[off + (hw ? hw - 1u : 0u)];
                    unsigned long long sm = __ballot(coop);
                    // Many such arms at once (a short-period tandem repeat: every arm's window holds a dozen hits of the
                    // probe): one pass over the HITS instead of one over the hits per arm -- the arm that won hit h is the
                    // one whose creation number is best[h], no window test needed; hits in ascending order, so the last
                    // one an arm won stays.  (The longest segments of tiers 2, 4, 5 and 6 of a GRCh38-shaped pass spent
                    // 6-54 K cycles per probe in the per-arm loop below.)
                    if (sm != 0ull && (uint32_t)__popcll(sm) * 8u >= cnt) {
                        if (lane == 0) DBG_ADD(10, 1);
                        for (uint32_t h0 = 0; h0 < cnt; h0 += 64u) {
                            const uint32_t h = h0 + (uint32_t)lane;
                            const PosT xh = h < cnt ? s_hits[off + h] : (PosT)0;
                            const uint32_t bh = h < cnt ? s_best[bb][h] : kNone;
                            const uint32_t nh = min(64u, cnt - h0);
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
                    if (lane == 0 && sm) DBG_ADD(5, __popcll(sm));
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
                        const uint32_t hm_ = coop_resolve(lo_u, w_u, lane_of(a_seq[L], l), cnt, off, bb, x_u);
                        if ((uint32_t)lane == l && hm_ != kNone) {
                            hw = hm_ + 1u;
                            xw = x_u;
                        }
                    }
                    const bool won = hw != 0u;
                    // ExtendArm (src/automaton.rs:133-150) or one more step of age (:166-171), by selects
                    uint32_t thr_new;
                    if constexpr (kWidePos) thr_new = arm_threshold((uint64_t)(i + k) - (uint64_t)a_ls[L], G);
                    else thr_new = max(G, ((uint32_t)(i + k) - (uint32_t)a_ls[L]) / 10u);
                    const uint64_t sum_g = (uint64_t)a_gap[L] + step;
                    const uint32_t aged = sum_g > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)sum_g;
                    a_re[L] = won ? (PosT)(xw + k) : a_re[L];
                    if (won) s_cle[L * NT + tid] = (PosT)(i + k);
                    a_thr[L] = won ? thr_new : a_thr[L];
                    a_gap[L] = won ? 0u : aged;
                    const bool dead = !was_free && !won && aged >= G;  // never matches again
                    if (__ballot(dead)) {
                        if (lane == 0) DBG_ADD(6, 1);
                        const PosT rs = s_crs[L * NT + tid];
                        const bool report = dead && (uint64_t)(a_re[L] - rs) >= rp.M;
                        if (__ballot(report)) {
                            if (lane == 0) DBG_ADD(7, 1);
                            emit_records(report, a_ls[L], s_cle[L * NT + tid], rs, a_re[L], a_seq[L]);
                        }
                    }
                    a_seq[L] = dead ? kNoSeq : a_seq[L];
                }
                // NewArm by owner pull: the r-th empty slot takes the r-th unmatched hit
                if (n_new && fmask) {
                    const uint32_t base_r = L * NW < 64 ? lane_of(fincl - fv, (uint32_t)(L * NW) + wave)
                                                        : lane_of(fincl2 - fv2, (uint32_t)(L * NW - 64) + wave);
                    if (base_r < n_new) {
                        if (lane == 0) DBG_ADD(cnt <= 64u ? 8 : 9, 1);
                        const uint32_t r = base_r + (uint32_t)__popcll(fmask & lt_mask);
                        const bool take = was_free && r < n_new;
                        uint32_t hsel = 0;
                        if (cnt <= 64u) {
                            hsel = select_bit(m0, take ? r : 0u);
                        } else {
                            uint32_t pfx = 0;
                            for (uint32_t h0 = 0, gi = 0; h0 < cnt; h0 += 64u, ++gi) {
                                const unsigned long long nm = gi == 0 ? m0 : uni(s_newmask[wave][gi]);
                                const uint32_t cg = (uint32_t)__popcll(nm);
                                if (take && r >= pfx && r < pfx + cg) hsel = h0 + select_bit(nm, r - pfx);
                                pfx += cg;
                            }
                        }
                        // src/automaton.rs:151-164 (aged by this very probe)
                        const PosT x = s_hits[off + (take ? hsel : 0u)];
                        a_ls[L] = take ? (PosT)i : a_ls[L];
                        if (take) {
                            s_cle[L * NT + tid] = (PosT)(i + k);
                            s_crs[L * NT + tid] = x;
                        }
                        a_re[L] = take ? (PosT)(x + k) : a_re[L];
                        a_gap[L] = take ? step : a_gap[L];
                        a_thr[L] = take ? thr0 : a_thr[L];
                        a_seq[L] = take ? seq_base + r : a_seq[L];
                        livemask |= 1u << L;
                    }
                }
            }
            if (full) {
                next_seq += n_new;
                fam_open = true;
                if (wave == 0u)  // (every lane, same 16 bytes)
                    *reinterpret_cast<uint4 *>(&s_pub[q.tb][0]) = make_uint4(next_seq, fam_seq, 1u, 0u);
            }
            ran_full = full;
            {   // what the next probe can count on (every wave: from the free counts and cnt alone)
                const uint32_t free00 = lane_of(fv, 0u);  // empty slots of (layer 0, wave 0)
                solo_w0 = total_free - free00 == (uint32_t)CAP - 64u && cnt <= free00;  // (the new arms fit there too)
                solo_a = 64u - free00 + cnt;
                want_migrate = kSoloHits != 0u && !solo_w0 && A0 + cnt <= 60u;  // (few arms, some astray)
            }
            PROF_STOP(7);
        };

        // ---- migration: when few arms are left but some sit outside (wave 0, layer 0) -- a burst of a dense repeat is
        // over and, say, the one long arm of a pair of homologous chromosomes was created in wave 5 -- they move there,
        // so that the sparse stretch that follows runs as solo probes.  Two barriers, once per burst.
        auto migrate = [&]() {
            uint32_t lv[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int L = 0; L < S; ++L) lv[L] = (uint32_t)__popcll(__ballot(a_seq[L] != kNoSeq));
            *reinterpret_cast<uint4 *>(&s_mig[wave][0]) = make_uint4(lv[0], lv[1], lv[2], lv[3]);
            if constexpr (S > 4) *reinterpret_cast<uint4 *>(&s_mig[wave][4]) = make_uint4(lv[4], lv[5], lv[6], lv[7]);
            lds_barrier();
            // movers ranked (layer, wave, lane); the pair (0, 0) stays
            uint32_t mv = lane < S * NW ? s_mig[lane % NW][lane / NW] : 0u;
            const uint32_t here = lane_of(mv, 0u);  // live arms already in place
            if (lane == 0) mv = 0u;
            const uint32_t mv2 = S * NW > 64 && lane + 64 < S * NW ? s_mig[(lane + 64) % NW][(lane + 64) / NW] : 0u;
            const uint32_t mincl = wave_incl_scan(mv);
            uint32_t n_mov = lane_of(mincl, 63u);
            uint32_t mincl2 = 0;
            if constexpr (S * NW > 64) {
                mincl2 = wave_incl_scan(mv2) + n_mov;
                n_mov = lane_of(mincl2, 63u);
            }
            const bool go = n_mov > 0u && n_mov <= kMigMax && here + n_mov <= 64u;  // (the same in every wave)
            if (tid == 0) DBG_ADD(go ? 5 : 6, 1);
            if (go) {
#pragma unroll
                for (int L = 0; L < S; ++L) {
                    if (L == 0 && wave == 0u) continue;
                    const bool live = a_seq[L] != kNoSeq;
                    const unsigned long long lm = __ballot(live);
                    if (!lm) continue;
                    const uint32_t base = L * NW < 64 ? lane_of(mincl - mv, (uint32_t)(L * NW) + wave)
                                                      : lane_of(mincl2 - mv2, (uint32_t)(L * NW - 64) + wave);
                    if (live) {
                        const uint32_t r = base + (uint32_t)__popcll(lm & lt_mask);
                        s_x_ls[r] = a_ls[L];
                        s_x_re[r] = a_re[L];
                        s_x_le[r] = s_cle[L * NT + tid];
                        s_x_rs[r] = s_crs[L * NT + tid];
                        s_x_thr[r] = a_thr[L];
                        s_x_gap[r] = a_gap[L];
                        s_x_seq[r] = a_seq[L];
                        a_seq[L] = kNoSeq;
                    }
                    livemask &= ~(1u << L);
                }
            }
            lds_barrier();
            if (go) {
                if (wave == 0u) {
                    const bool is_free = a_seq[0] == kNoSeq;
                    const unsigned long long fmask = __ballot(is_free);
                    const uint32_t r = (uint32_t)__popcll(fmask & lt_mask);
                    const bool take = is_free && r < n_mov;
                    const uint32_t rr = take ? r : 0u;
                    a_ls[0] = take ? s_x_ls[rr] : a_ls[0];
                    a_re[0] = take ? s_x_re[rr] : a_re[0];
                    a_thr[0] = take ? s_x_thr[rr] : a_thr[0];
                    a_gap[0] = take ? s_x_gap[rr] : a_gap[0];
                    a_seq[0] = take ? s_x_seq[rr] : a_seq[0];
                    if (take) {
                        s_cle[tid] = s_x_le[rr];
                        s_crs[tid] = s_x_rs[rr];
                    }
                    livemask |= 1u;
                }
                solo_w0 = true;
                solo_a = here + n_mov;
            }
            want_migrate = false;
        };

        // ---- a probe run by wave 0 alone: every live arm is one of its 64 layer-0 slots, the hits are few ----------
        // Same transitions as A + B (src/automaton.rs:96-200), without the table: every hit is tested against the 64
        // arms at once (ballot), its arm is the accepting one with the smallest creation number (DPP minimum when more
        // than one accepts), the last hit an arm wins extends it, the unmatched hits take the empty lanes in hit order.
        auto solo_probe = [&](const Probe &q) -> uint32_t {
            const uint32_t cnt = q.cnt, off = q.off;
            const uint64_t i = q.i;
            bool live = a_seq[0] != kNoSeq;
            if (pend) {  // the quiet probes since the last hit-probe: src/automaton.rs:166-171
                const uint64_t sum_g = (uint64_t)a_gap[0] + pend;
                const uint32_t aged = sum_g > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)sum_g;
                const bool dead = live && aged >= G;
                a_gap[0] = aged;
                if (__ballot(dead)) {
                    const PosT rs = s_crs[tid];
                    const bool report = dead && (uint64_t)(a_re[0] - rs) >= rp.M;
                    if (__ballot(report)) emit_records(report, a_ls[0], s_cle[tid], rs, a_re[0], a_seq[0]);
                }
                a_seq[0] = dead ? kNoSeq : a_seq[0];
                live = live && !dead;
            }
            if (fam_open && !__ballot(live)) {  // the flush of src/automaton.rs:182-200
                ++fam_seq;
                next_seq = 0;
                fam_open = false;
            }
            const PosT lo = (PosT)(a_re[0] - k + 1u);
            const WinT w = live ? (WinT)a_thr[0] + (WinT)(k - 1u) : (WinT)0;
            const PosT x_l = (uint32_t)lane < cnt ? s_hits[off + lane] : (PosT)0;  // lane j holds hit j (cnt <= kSoloHits)
            // the lanes empty before this probe take its new arms (an arm that dies in this probe frees its lane for the
            // next one: solo_a + cnt <= 64 counted the arms alive before)
            unsigned long long fm = __ballot(!live);
            bool won = false, born = false;
            PosT xw = 0, xb = 0;  // the last hit this arm won; the hit this lane's new arm starts from
            uint32_t rb = 0, n_new = 0;
            for (uint32_t j = 0; j < cnt; ++j) {
                PosT x;
                if constexpr (kWidePos) x = (PosT)lane_of((unsigned long long)x_l, j);
                else x = (PosT)lane_of((uint32_t)x_l, j);
                const bool ok = (WinT)(PosT)(x - lo) < w;
                const unsigned long long okm = __ballot(ok);
                if (okm) {
                    // the first accepting arm in list order (:67-78) = the smallest creation number; one arm as a rule
                    bool mine = ok;
                    if (okm & (okm - 1ull)) {  // (every lane takes part in the reduction: no short-circuit around it)
                        const uint32_t first = wave_min_u32(ok ? a_seq[0] : 0xFFFFFFFFu);
                        mine = ok && a_seq[0] == first;
                    }
                    won = won || mine;
                    xw = mine ? x : xw;  // (hit order: the last one stays, src/automaton.rs:133-150)
                } else {  // NewArm (src/automaton.rs:151-164), creation numbers in hit order
                    const uint32_t tl = (uint32_t)(__ffsll((long long)fm) - 1);
                    fm &= fm - 1ull;
                    const bool me = (uint32_t)lane == tl;
                    born = born || me;
                    xb = me ? x : xb;
                    rb = me ? n_new : rb;
                    ++n_new;
                }
            }
            uint32_t thr_new;
            if constexpr (kWidePos) {
                const uint64_t len = (uint64_t)(i + k) - (uint64_t)a_ls[0];
                if (__ballot(won && (len >> 32) != 0ull)) thr_new = arm_threshold(len, G);  // (a left arm of 4 Gbp and more)
                else thr_new = max(G, (uint32_t)len / 10u);
            } else {
                thr_new = max(G, ((uint32_t)(i + k) - (uint32_t)a_ls[0]) / 10u);
            }
            const uint32_t sum_g = a_gap[0] + step;
            const uint32_t aged = sum_g < step ? 0xFFFFFFFFu : sum_g;  // (saturating)
            a_re[0] = won ? (PosT)(xw + k) : a_re[0];
            if (won) s_cle[tid] = (PosT)(i + k);
            a_thr[0] = won ? thr_new : a_thr[0];
            a_gap[0] = won ? 0u : aged;
            const bool dead = live && !won && aged >= G;
            if (__ballot(dead)) {
                const PosT rs = s_crs[tid];
                const bool report = dead && (uint64_t)(a_re[0] - rs) >= rp.M;
                if (__ballot(report)) emit_records(report, a_ls[0], s_cle[tid], rs, a_re[0], a_seq[0]);
            }
            a_seq[0] = dead ? kNoSeq : a_seq[0];
            if (n_new) {
                a_ls[0] = born ? (PosT)i : a_ls[0];
                if (born) {
                    s_cle[tid] = (PosT)(i + k);
                    s_crs[tid] = xb;
                }
                a_re[0] = born ? (PosT)(xb + k) : a_re[0];
                a_gap[0] = born ? step : a_gap[0];
                a_thr[0] = born ? thr0 : a_thr[0];
                a_seq[0] = born ? next_seq + rb : a_seq[0];
            }
            next_seq += n_new;
            fam_open = true;
            const uint32_t a_now = (uint32_t)__popcll(__ballot(a_seq[0] != kNoSeq));
            livemask = (livemask & ~1u) | (a_now ? 1u : 0u);
            return a_now;
        };

        bool staged_before = false;
        // (Loading the next batch's per-probe words and hit rows ahead, in registers, was measured: the 1024-thread
        // shapes sit at their 128-register cap, the extra live values spill inside the probe loop, and the tandem
        // array went from 78 to 86 ms while the two-genome pass did not move.)
        for (uint32_t g = g0; g < g_end && !done;) {
            // ---- stage a batch of up to 64 probes (every wave computes the same masks) ----
            PROF_START();
            const uint32_t nb = min(64u, g_end - g);
            if (tid == 0) heartbeat(P, g0, g);
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
            const uint32_t tot = (uint32_t)((nbb == nb ? r_hi : lane_of(r_l, nbb)) - base);
            if (tot) {  // (a batch of quiet probes stages nothing and needs no barrier)
                if (staged_before) lds_barrier();  // (the last B of the previous batch read the staged hits)
                staged_before = true;
                for (uint32_t r = tid; r < tot; r += NT) s_hits[r] = P.hits[base + r];
                lds_barrier();
            }
            const unsigned long long in_batch = nbb >= 64 ? ~0ull : ((1ull << nbb) - 1ull);
            const unsigned long long hm = __ballot(f_l >= 1u && f_l < kPending) & in_batch;
            const unsigned long long qm = __ballot(f_l == 0u) & in_batch;
            PROF_STOP(0);
            PROF_COUNT(1, 1);
            uint32_t pos = 0;
            bool pre_indexed = false, have_prev = false;
            Probe prev{0, 0, 0, 0, 0};
            for (;;) {
                // ---- the next hit-probe of the batch; the quiet probes before it only age ----------
                bool have_cur = false;
                Probe cur{0, 0, 0, 0, 0};
                if (!done) {
                    const unsigned long long hmr = pos >= 64 ? 0ull : (hm >> pos) << pos;
                    const uint32_t b = hmr ? (uint32_t)(__ffsll((long long)hmr) - 1) : 64u;
                    const unsigned long long upto = b >= 64 ? ~0ull : ((1ull << b) - 1ull);
                    const unsigned long long from = pos >= 64 ? 0ull : ~((1ull << pos) - 1ull);
                    const uint32_t q = (uint32_t)__popcll(qm & upto & from);
                    if (q) {  // folded into the next pass over the arms
                        quiet += q;
                        pend += q * step;
                        if (quiet >= rp.tstar) done = true;  // every arm is dead (gap >= G): the segment is over
                    }
                    if (hmr && !done) {
                        have_cur = true;
                        quiet = 0;
                        pos = b + 1;
                        cur.cnt = lane_of(f_l, b);
                        cur.off = lane_of(rel_l, b);
                        cur.i = (uint64_t)(g + b - pb + 1) * step;
                        cur.tb = par;
                        cur.bb = tri;
                    } else {
                        pos = 64;
                    }
                }
                if (have_prev) {
                    phase_b(prev);
                    have_prev = false;
                    if (overflow) break;
                    if (want_migrate) migrate();
                }
                if (!have_cur) break;
                PROF_COUNT(5, 1);
                PROF_COUNT(11, cur.cnt);
                if (solo_w0 && cur.cnt <= kSoloHits && solo_a + cur.cnt <= 64u) {
                    // wave 0 alone; the others wait for its count at the barrier and will re-read the bookkeeping
                    if (pre_indexed) {  // (this probe was indexed ahead for nothing: forget its stash)
                        if (tid == 0) s_nstash[tri] = 0u;
                        if ((uint32_t)tid < kBitWords) s_rowbits[tri][tid] = 0u;
                        pre_indexed = false;
                    }
                    if (wave == 0u) {
                        uint32_t a_now = solo_probe(cur);
                        if (lane == 0) DBG_ADD(0, 1);  // (diagnostic build: probes run solo)
                        // ... and goes on through the staged batch, without a barrier, while the next hit-probe is as
                        // sparse (the same accounting of the quiet probes in between as at the top of this loop)
                        for (;;) {
                            const unsigned long long hmr = pos >= 64 ? 0ull : (hm >> pos) << pos;
                            if (!hmr) break;
                            const uint32_t b = (uint32_t)(__ffsll((long long)hmr) - 1);
                            const unsigned long long span = ((1ull << b) - 1ull) & ~((1ull << pos) - 1ull);
                            const uint32_t q = (uint32_t)__popcll(qm & span);
                            if (q >= rp.tstar) break;  // (the segment ends there: left to the workgroup)
                            const uint32_t cnt2 = lane_of(f_l, b);
                            if (cnt2 > kSoloHits || a_now + cnt2 > 64u) break;
                            pend = q * step;
                            Probe nx{0, 0, 0, 0, 0};
                            nx.cnt = cnt2;
                            nx.off = lane_of(rel_l, b);
                            nx.i = (uint64_t)(g + b - pb + 1) * step;
                            pos = b + 1;
                            a_now = solo_probe(nx);
                            if (lane == 0) DBG_ADD(0, 1);
                        }
                        // (every lane, same words) what the other waves need: how far this went, the live count, and
                        // the bookkeeping for their next B
                        *reinterpret_cast<uint4 *>(&s_run[solo_par][0]) = make_uint4(pos, 0u, a_now, 0u);
                        const uint4 pv = make_uint4(next_seq, fam_seq, 1u, 0u);
                        *reinterpret_cast<uint4 *>(&s_pub[0][0]) = pv;
                        *reinterpret_cast<uint4 *>(&s_pub[1][0]) = pv;
                    } else {
                        ran_full = false;
                    }
                    pend = 0;
                    lds_barrier();
                    {
                        const uint4 rv = *reinterpret_cast<const uint4 *>(&s_run[solo_par][0]);
                        pos = uni(rv.x);
                        solo_a = uni(rv.z);
                    }
                    solo_par ^= 1u;
                    continue;
                }
                if (tid == 0) {  // (diagnostic build: probes run by the workgroup, and why not solo)
                    DBG_ADD(1, 1);
                    if (!solo_w0) DBG_ADD(2, 1);
                    else if (cur.cnt > kSoloHits) DBG_ADD(3, 1);
                    else DBG_ADD(4, 1);
                }
                PROF_START();
                if (!pre_indexed) {
                    if (++gen >> kGenBits) {  // generation wrap: clear the tables once
                        lds_barrier();
                        clear_table();
                        gen = 1;
                        lds_barrier();
                    }
                    insert_hits(cur.cnt, cur.off, cur.tb, cur.bb);
                    lds_barrier();
                }
                PROF_STOP(2);
                PROF_START();
                phase_a(cur.cnt, cur.off, true, cur.tb, cur.bb);
                pend = 0;
                if (tid == 0) s_nstash[(tri + 2u) % 3u] = 0u;  // the stash of the probe after next (indexed in the next interval)
                if ((uint32_t)tid < kBitWords) s_rowbits[(tri + 2u) % 3u][tid] = 0u;  // ... and its rows' occupancy bits
                PROF_STOP(4);
                PROF_START();
                {   // next hit probe of this staged batch, if any: the top threads index it in this interval
                    const unsigned long long nxt = pos >= 64 ? 0ull : (hm >> pos) << pos;
                    const bool can_pre = nxt != 0ull && ((gen + 1u) >> kGenBits) == 0u;
                    if (can_pre) {
                        const uint32_t nb2 = (uint32_t)(__ffsll((long long)nxt) - 1);
                        ++gen;
                        insert_hits(lane_of(f_l, nb2), lane_of(rel_l, nb2), par ^ 1u, (tri + 1u) % 3u);
                    }
                    pre_indexed = can_pre;
                }
                PROF_STOP(8);
                PROF_START();
                lds_barrier();
                PROF_STOP(3);
                par ^= 1u;
                tri = (tri + 1u) % 3u;
                if constexpr (kPipe) {
                    prev = cur;
                    have_prev = true;
                } else {  // two barriers per hit-probe: B right away, the next A after every wave is through it
                    phase_b(cur);
                    if (overflow) break;
                    lds_barrier();
                    if (want_migrate) migrate();
                }
            }
            if (overflow) break;
            g += nbb;
        }
        if (!overflow) {
            // the age of the trailing quiet probes: whatever it kills is reported, and the family closes if
            // nothing is left
            phase_a(0, 0, false, par, tri);
            pend = 0;
            lds_barrier();
            const uint32_t fv = lane < S * NW ? s_free[par][lane % NW][lane / NW] : 0u;
            uint32_t total_free = lane_of(wave_incl_scan(fv), 63u);
            if constexpr (S * NW > 64) {
                const uint32_t fv2 = lane + 64 < S * NW ? s_free[par][(lane + 64) % NW][(lane + 64) / NW] : 0u;
                total_free += lane_of(wave_incl_scan(fv2), 63u);
            }
            if (fam_open && total_free == (uint32_t)CAP) fam_open = false;
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
        livemask = 0;
#ifdef ASGART_PROF_WAVE0
        if (wave == 0u) {  // (diagnostic build with -DASGART_PROF_WAVE0: the wave that holds the first arms reports)
#else
        if (wave == min(4u, (uint32_t)(NW - 1))) {  // (diagnostic build: the reporting wave)
#endif
            PROF_FLUSH();
        }
        lds_barrier();
    }
    rec_flush(rec_alloc, P, lane);
    wg_busy(P);
}

}  // namespace asgart
