/*
 * oracle/asgart_oracle.c -- TEST INFRASTRUCTURE ONLY (see asgart_oracle.h,
 * oracle/README.md).  PARITY UNPINNED (no reference-owned vectors exist).
 *
 * Literal CPU restatement, in plain C, of the reference's hot path.  Every
 * function cites the reference lines it follows (paths under /root/reference).
 * Quirks are kept on purpose: position 0 is never
 * probed, only the first base of a probe is tested for 'N', both `continue`s
 * skip ageing/pruning/flushing, ExtendArm is last-writer-wins in SA order,
 * live arms are dropped at the end of a needle, `m.start != i` compares a
 * global coordinate with a needle-local one.
 */
#include "asgart_oracle.h"

#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ALPHABET, src/structs.rs:10 -- order matters only for table indexing here */
static const uint8_t ALPHABET[5] = {'A', 'T', 'G', 'C', 'N'};
#define CACHE_LEN 8 /* src/searcher.rs:15 */
#define CACHE_ENTRIES 390625 /* 5^8 */

struct oracle_searcher {
    uint64_t *lo; /* CACHE_ENTRIES */
    uint64_t *hi;
    uint64_t offset;
    int8_t code[256];
};

/* --------------------------------------------------------------------------
 * sa_searchb64 (src/divsufsort.rs:22-32; delehef fork of libdivsufsort, source
 * absent).  Semantics restated from the upstream sa_search contract: the range
 * of SA slots in [init_left, init_right) whose suffix has P as a prefix; a
 * suffix that ends before P does compares smaller.  Returns the count, *left =
 * first slot (for count==0: the insertion point).
 * ------------------------------------------------------------------------ */
/* `sa_searchb64` of the delehef fork of libdivsufsort (declared at src/divsufsort.rs:22-32, called
 * at src/searcher.rs:118-128 with init_left = 0, init_right = sa.len()).  The fork's source is absent
 * from /root/reference (empty submodule), so this restates the PUBLISHED upstream routine it extends,
 * libdivsufsort 2.0.x lib/utils.c `sa_search` (+ its `_compare`), started on [init_left, init_right)
 * instead of [0, SAsize): a halving bisection that, once a matching suffix is found, bisects the left
 * and the right part separately, skipping the characters already known to match (`match`).  For a
 * suffix array that is sorted under the comparator this returns the unique equal range; the probing
 * order only matters for --trim (src/bin/asgart.rs:142-148), where the array holds the suffixes of
 * data[start..end]+'$' but is compared through the FULL text, so the few suffixes that end within
 * 8 bases of `end` are out of place.  That corner is unpinned twice over (fork source absent, no
 * reference vectors). */
static int published_compare(const uint8_t *T, int64_t Tsize, const uint8_t *P, int64_t Psize,
                             int64_t suf, int64_t *match) {
    int64_t i, j;
    int r = 0;
    for (i = suf + *match, j = *match; (i < Tsize) && (j < Psize) && ((r = (int)T[i] - (int)P[j]) == 0); ++i, ++j) {
    }
    *match = j;
    return (r == 0) ? -(j != Psize) : r;
}

#define MIN_(a, b) ((a) < (b) ? (a) : (b))
static int64_t sa_searchb(const uint8_t *T, int64_t Tsize, const uint8_t *P, int64_t Psize,
                          const int64_t *SA, int64_t *left, int64_t init_left,
                          int64_t init_right) {
    int64_t size, lsize, rsize, half;
    int64_t match, lmatch, rmatch, llmatch, lrmatch, rlmatch, rrmatch;
    int64_t i, j, k;
    int r;
    *left = -1;
    if (Tsize == 0 || init_right <= init_left) return 0;
    for (i = j = k = init_left, lmatch = rmatch = 0, size = init_right - init_left, half = size >> 1; 0 < size;
         size = half, half >>= 1) {
        match = MIN_(lmatch, rmatch);
        r = published_compare(T, Tsize, P, Psize, SA[i + half], &match);
        if (r < 0) {
            i += half + 1;
            half -= (size & 1) ^ 1;
            lmatch = match;
        } else if (r > 0) {
            rmatch = match;
        } else {
            lsize = half, j = i, rsize = size - half - 1, k = i + half + 1;
            /* left part */
            for (llmatch = lmatch, lrmatch = match, half = lsize >> 1; 0 < lsize; lsize = half, half >>= 1) {
                lmatch = MIN_(llmatch, lrmatch);
                r = published_compare(T, Tsize, P, Psize, SA[j + half], &lmatch);
                if (r < 0) {
                    j += half + 1;
                    half -= (lsize & 1) ^ 1;
                    llmatch = lmatch;
                } else {
                    lrmatch = lmatch;
                }
            }
            /* right part */
            for (rlmatch = match, rrmatch = rmatch, half = rsize >> 1; 0 < rsize; rsize = half, half >>= 1) {
                rmatch = MIN_(rlmatch, rrmatch);
                r = published_compare(T, Tsize, P, Psize, SA[k + half], &rmatch);
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
    *left = (0 < (k - j)) ? j : i;
    return k - j;
}
#undef MIN_

/* Searcher::indexize packs 8 bytes little-endian into a u64 HashMap key
 * (src/searcher.rs:95-97); a dense base-5 index over ALPHABET is equivalent. */
static int64_t dense_index(const oracle_searcher *s, const uint8_t *p) {
    int64_t idx = 0;
    for (int j = 0; j < CACHE_LEN; ++j) {
        int c = s->code[p[j]];
        if (c < 0) return -1;
        idx = idx * 5 + c;
    }
    return idx;
}

/* Searcher::new, src/searcher.rs:99-143: one sa_searchb64 per 8-mer over
 * ALPHABET, entry = (start, start+count). */
oracle_searcher *oracle_searcher_new(const uint8_t *dna, int64_t n, const int64_t *sa,
                                     int64_t sa_len, uint64_t offset) {
    oracle_searcher *s = (oracle_searcher *)calloc(1, sizeof(*s));
    if (!s) return NULL;
    s->lo = (uint64_t *)malloc(CACHE_ENTRIES * sizeof(uint64_t));
    s->hi = (uint64_t *)malloc(CACHE_ENTRIES * sizeof(uint64_t));
    if (!s->lo || !s->hi) {
        oracle_searcher_free(s);
        return NULL;
    }
    s->offset = offset;
    memset(s->code, -1, sizeof(s->code));
    for (int c = 0; c < 5; ++c) s->code[ALPHABET[c]] = (int8_t)c;
    uint8_t p[CACHE_LEN];
    for (int64_t idx = 0; idx < CACHE_ENTRIES; ++idx) {
        int64_t v = idx;
        for (int j = CACHE_LEN - 1; j >= 0; --j) {
            p[j] = ALPHABET[v % 5];
            v /= 5;
        }
        int64_t out = 0;
        int64_t count = sa_searchb(dna, n, p, CACHE_LEN, sa, &out, 0, sa_len);
        s->lo[idx] = (uint64_t)out;
        s->hi[idx] = (uint64_t)(out + count);
    }
    return s;
}

void oracle_searcher_free(oracle_searcher *s) {
    if (!s) return;
    free(s->lo);
    free(s->hi);
    free(s);
}

int32_t oracle_searcher_cache_get(const oracle_searcher *s, const uint8_t *p8, uint64_t *lo,
                                  uint64_t *hi) {
    int64_t idx = dense_index(s, p8);
    if (idx < 0) return -1;
    *lo = s->lo[idx];
    *hi = s->hi[idx];
    return 0;
}

/* comparator of src/searcher.rs:164-170 */
enum { ORD_LESS = -1, ORD_EQUAL = 0, ORD_GREATER = 1 };
static inline int search_cmp(const uint8_t *dna, int64_t n, int64_t x, const uint8_t *pattern,
                             int64_t k) {
    if (x + k > n) return ORD_LESS;
    int c = memcmp(dna + x, pattern, (size_t)k);
    return c < 0 ? ORD_LESS : (c > 0 ? ORD_GREATER : ORD_EQUAL);
}

/* superslice 1.0 `Ext::equal_range_by` (crate source absent from
 * /root/reference; call site src/searcher.rs:164).  Restated FROM MEMORY of the
 * published crate: a branch-free simultaneous lower/upper bisection that
 * halves `size` and keeps two bases.  For a comparator that is monotone over
 * the slice this equals [lower_bound, upper_bound) whatever the probing order;
 * the probing order only matters for the <= k-9 text-tail suffixes for which
 * the reference comparator says Less although they sort Greater (SURVEY 8a
 * corner) -- that corner is therefore doubly unpinned. */
static void equal_range_by(const uint8_t *dna, int64_t n, const int64_t *slice, int64_t len,
                           const uint8_t *pattern, int64_t k, int64_t *out_start,
                           int64_t *out_end) {
    if (len == 0) {
        *out_start = *out_end = 0;
        return;
    }
    int64_t size = len, b0 = 0, b1 = 0;
    while (size > 1) {
        int64_t half = size / 2;
        int64_t m0 = b0 + half, m1 = b1 + half;
        int c0 = search_cmp(dna, n, slice[m0], pattern, k);
        int c1 = (m1 == m0) ? c0 : search_cmp(dna, n, slice[m1], pattern, k);
        if (c0 == ORD_LESS) b0 = m0;
        if (c1 != ORD_GREATER) b1 = m1;
        size -= half;
    }
    int c0 = search_cmp(dna, n, slice[b0], pattern, k);
    int c1 = (b1 == b0) ? c0 : search_cmp(dna, n, slice[b1], pattern, k);
    *out_start = b0 + (c0 == ORD_LESS);
    *out_end = b1 + (c1 != ORD_GREATER);
}

/* Searcher::search, src/searcher.rs:145-180 */
int64_t oracle_searcher_search(const oracle_searcher *s, const uint8_t *dna, int64_t n,
                               const int64_t *sa, const uint8_t *pattern, int64_t k,
                               uint64_t *out_starts, int64_t cap, uint64_t *range_lo,
                               uint64_t *range_hi, uint64_t *bucket_size) {
    int64_t idx = dense_index(s, pattern);
    if (idx < 0) return -1; /* reference: panic!("Unable to find ...") :155-161 */
    int64_t lstart = (int64_t)s->lo[idx], rstart = (int64_t)s->hi[idx];
    int64_t rs, re;
    equal_range_by(dna, n, sa + lstart, rstart - lstart, pattern, k, &rs, &re);
    if (re < rs) re = rs; /* reference would panic on an inverted slice; cannot occur */
    if (range_lo) *range_lo = (uint64_t)(lstart + rs);
    if (range_hi) *range_hi = (uint64_t)(lstart + re);
    if (bucket_size) *bucket_size = (uint64_t)(rstart - lstart);
    int64_t count = re - rs;
    for (int64_t j = 0; j < count && j < cap; ++j)
        out_starts[j] = s->offset + (uint64_t)sa[lstart + rs + j];
    return count;
}

/* ------------------------------------------------------------------------ */
struct oracle_families {
    uint64_t n_fam, cap_fam; /* fam_offsets has n_fam+1 valid entries */
    uint64_t *fam_offsets;
    uint64_t n_sd, cap_sd;
    oracle_proto_sd *sds;
};

static oracle_families *families_new(void) {
    oracle_families *f = (oracle_families *)calloc(1, sizeof(*f));
    if (!f) return NULL;
    f->cap_fam = 16;
    f->fam_offsets = (uint64_t *)malloc((f->cap_fam + 1) * sizeof(uint64_t));
    f->cap_sd = 16;
    f->sds = (oracle_proto_sd *)malloc(f->cap_sd * sizeof(oracle_proto_sd));
    if (!f->fam_offsets || !f->sds) {
        oracle_families_free(f);
        return NULL;
    }
    f->fam_offsets[0] = 0;
    return f;
}

static int families_push_sd(oracle_families *f, oracle_proto_sd sd) {
    if (f->n_sd == f->cap_sd) {
        uint64_t nc = f->cap_sd * 2;
        oracle_proto_sd *p = (oracle_proto_sd *)realloc(f->sds, nc * sizeof(*p));
        if (!p) return -2;
        f->sds = p;
        f->cap_sd = nc;
    }
    f->sds[f->n_sd++] = sd;
    return 0;
}

static int families_close_family(oracle_families *f) {
    if (f->n_fam == f->cap_fam) {
        uint64_t nc = f->cap_fam * 2;
        uint64_t *p = (uint64_t *)realloc(f->fam_offsets, (nc + 1) * sizeof(uint64_t));
        if (!p) return -2;
        f->fam_offsets = p;
        f->cap_fam = nc;
    }
    f->fam_offsets[++f->n_fam] = f->n_sd;
    return 0;
}

int32_t oracle_families_from_arrays(const uint64_t *fam_offsets, uint64_t n_fam,
                                    const oracle_proto_sd *sds, oracle_families **out) {
    oracle_families *f = families_new();
    if (!f) return -2;
    for (uint64_t fam = 0; fam < n_fam; ++fam) {
        for (uint64_t j = fam_offsets[fam]; j < fam_offsets[fam + 1]; ++j)
            if (families_push_sd(f, sds[j])) return -2;
        if (families_close_family(f)) return -2;
    }
    *out = f;
    return 0;
}

void oracle_families_counts(const oracle_families *f, uint64_t *n_families, uint64_t *n_sds) {
    *n_families = f->n_fam;
    *n_sds = f->n_sd;
}

void oracle_families_copy(const oracle_families *f, uint64_t *fam_offsets, oracle_proto_sd *sds) {
    memcpy(fam_offsets, f->fam_offsets, (f->n_fam + 1) * sizeof(uint64_t));
    if (f->n_sd) memcpy(sds, f->sds, f->n_sd * sizeof(oracle_proto_sd));
}

void oracle_families_free(oracle_families *f) {
    if (!f) return;
    free(f->fam_offsets);
    free(f->sds);
    free(f);
}

/* d_ss, src/automaton.rs:207-216 */
int64_t oracle_d_ss(uint64_t a_start, uint64_t a_end, uint64_t m_start, uint64_t m_end) {
    if ((m_start >= a_start && m_start <= a_end) || (m_end >= a_start && m_end <= a_end)) return 0;
    int64_t d1 = (int64_t)a_start - (int64_t)m_end;
    int64_t d2 = (int64_t)a_end - (int64_t)m_start;
    if (d1 < 0) d1 = -d1;
    if (d2 < 0) d2 = -d2;
    return d1 < d2 ? d1 : d2;
}

/* Segment / Arm / Operation, src/automaton.rs:10-54 (Segment.tag is always 0) */
typedef struct {
    uint64_t l_start, l_end, r_start, r_end;
    uint64_t gap;
    uint8_t active, dirty;
} arm_t;

typedef struct {
    uint8_t is_extend;
    uint64_t a; /* ExtendArm: arm index ; NewArm: i        */
    uint64_t b; /* ExtendArm: l_end     ; NewArm: m_start  */
    uint64_t c; /* ExtendArm: r_end     ; NewArm: m_end    */
} op_t;

typedef struct {
    uint64_t *v;
    uint64_t n, cap;
} u64vec;

static int u64vec_reserve(u64vec *a, uint64_t need) {
    if (need <= a->cap) return 0;
    uint64_t nc = a->cap ? a->cap : 64;
    while (nc < need) nc *= 2;
    uint64_t *p = (uint64_t *)realloc(a->v, nc * sizeof(uint64_t));
    if (!p) return -2;
    a->v = p;
    a->cap = nc;
    return 0;
}

static uint64_t ceil_log2_plus1(uint64_t b) { /* ceil(log2(b+1)) */
    uint64_t steps = 0, v = 1;
    while (v < b + 1) {
        v <<= 1;
        ++steps;
    }
    return steps;
}

/* The probe + hit-filter part shared by oracle_search_duplications and
 * oracle_probe_hits: automaton.rs:100-117.  Returns status 0/1/2 and leaves
 * the filtered hit starts (SA order) in `hits`. */
static int probe_filtered_hits(const uint8_t *needle, uint64_t needle_len, uint64_t needle_offset,
                               uint64_t i, const uint8_t *strand, int64_t n, const int64_t *sa,
                               const oracle_searcher *searcher, const oracle_settings *st,
                               u64vec *raw, u64vec *hits, oracle_stats *stats) {
    uint64_t k = st->probe_size;
    hits->n = 0;
    if (needle[i] == 'N') { /* :100-102 */
        if (stats) stats->probes_n_skipped++;
        return 1;
    }
    uint64_t bucket = 0;
    int64_t cnt = oracle_searcher_search(searcher, strand, n, sa, needle + i, (int64_t)k, raw->v,
                                         (int64_t)raw->cap, NULL, NULL, &bucket);
    if (cnt < 0) return -1;
    if ((uint64_t)cnt > raw->cap) {
        if (u64vec_reserve(raw, (uint64_t)cnt)) return -2;
        cnt = oracle_searcher_search(searcher, strand, n, sa, needle + i, (int64_t)k, raw->v,
                                     (int64_t)raw->cap, NULL, NULL, NULL);
    }
    if (stats) {
        stats->probes_searched++;
        stats->bisect_steps += ceil_log2_plus1(bucket);
        stats->raw_hits += (uint64_t)cnt;
    }
    if (u64vec_reserve(hits, (uint64_t)cnt)) return -2;
    for (int64_t j = 0; j < cnt; ++j) { /* :105-114 */
        uint64_t m_start = raw->v[j];
        if (m_start == i) continue; /* global vs needle-local on purpose */
        int keep = !st->reverse ? (m_start > i + needle_offset)
                                : (m_start >= needle_offset + needle_len - i);
        if (keep) hits->v[hits->n++] = m_start;
    }
    if (hits->n > st->max_cardinality) { /* :115-117 */
        if (stats) stats->probes_card_skipped++;
        return 2;
    }
    return 0;
}

/* automaton::search_duplications, src/automaton.rs:57-204 */
int32_t oracle_search_duplications(const uint8_t *needle, uint64_t needle_len,
                                   uint64_t needle_offset, const uint8_t *strand, int64_t n,
                                   const int64_t *sa, const oracle_searcher *searcher,
                                   volatile uint64_t *progress, const oracle_settings *st,
                                   oracle_stats *stats, oracle_families **out) {
    oracle_families *r = families_new();
    if (!r) return -2;
    *out = r;
    const uint64_t k = st->probe_size;
    const uint64_t step = k / 2;                       /* :90 */
    if (needle_len < st->min_duplication_length) return 0; /* :92-94 */
    /* `needle.len() - probe_size - step_size` underflows in the reference when
     * the needle is shorter than k+step (it then panics on an OOB index);
     * defined here as "no probes". */
    if (needle_len < k + step || step == 0) return 0;

    arm_t *arms = NULL;
    uint64_t n_arms = 0, cap_arms = 0;
    op_t *todo = NULL;
    uint64_t cap_todo = 0;
    u64vec raw = {0}, hits = {0};
    int32_t rc = 0;
    const int64_t e = (int64_t)st->max_gap_size; /* i64::from(settings.max_gap_size), :129 */

    uint64_t i = 0;
    while (i < needle_len - k - step) { /* :96 */
        i += step;                      /* :97 */
        if (progress) *progress = i;    /* :98 */
        if (stats) stats->probes_total++;

        int status = probe_filtered_hits(needle, needle_len, needle_offset, i, strand, n, sa,
                                         searcher, st, &raw, &hits, stats);
        if (status < 0) {
            rc = status;
            break;
        }
        if (status != 0) continue; /* both continues skip everything below */
        if (stats) {
            stats->filtered_hits += hits.n;
            if (hits.n) stats->probes_with_hits++;
        }

        for (uint64_t a = 0; a < n_arms; ++a) arms[a].dirty = 0; /* :120 */

        if (hits.n > cap_todo) {
            cap_todo = hits.n * 2;
            op_t *p = (op_t *)realloc(todo, cap_todo * sizeof(op_t));
            if (!p) {
                rc = -2;
                break;
            }
            todo = p;
        }
        /* try_extend_arms for every hit against the unchanged arms, :66-85,:122-134 */
        for (uint64_t h = 0; h < hits.n; ++h) {
            uint64_t m_start = hits.v[h], m_end = hits.v[h] + k;
            op_t op = {0, i, m_start, m_end};
            for (uint64_t j = 0; j < n_arms; ++j) {
                const arm_t *a = &arms[j];
                if (stats) stats->arm_tests++;
                if (!a->active) continue;
                int64_t tenth = (int64_t)(0.1 * (double)(a->l_end - a->l_start)); /* :69 */
                int64_t thr = e > tenth ? e : tenth;
                if (oracle_d_ss(a->r_start, a->r_end, m_start, m_end) < thr && m_end > a->r_end) {
                    op.is_extend = 1;
                    op.a = j;
                    op.b = i + k;
                    op.c = m_end;
                    break;
                }
            }
            todo[h] = op;
        }
        for (uint64_t h = 0; h < hits.n; ++h) /* :136-143 */
            if (todo[h].is_extend) {
                arm_t *a = &arms[todo[h].a];
                a->l_end = todo[h].b;
                a->r_end = todo[h].c;
                a->dirty = 1;
                a->gap = 0;
            }
        for (uint64_t h = 0; h < hits.n; ++h) /* :145-163 */
            if (!todo[h].is_extend) {
                if (n_arms == cap_arms) {
                    cap_arms = cap_arms ? cap_arms * 2 : 64;
                    arm_t *p = (arm_t *)realloc(arms, cap_arms * sizeof(arm_t));
                    if (!p) {
                        rc = -2;
                        goto done;
                    }
                    arms = p;
                }
                arm_t na = {todo[h].a, todo[h].a + k, todo[h].b, todo[h].c, 0, 1, 0};
                arms[n_arms++] = na;
            }
        for (uint64_t a = 0; a < n_arms; ++a) /* :166-171 */
            if (!arms[a].dirty) {
                arms[a].gap += step;
                if ((uint32_t)arms[a].gap >= st->max_gap_size) arms[a].active = 0;
            }
        if (n_arms > 200) { /* :173-179 */
            uint64_t w = 0;
            for (uint64_t a = 0; a < n_arms; ++a)
                if (arms[a].active ||
                    arms[a].l_end - arms[a].l_start >= st->min_duplication_length ||
                    arms[a].r_end - arms[a].r_start >= st->min_duplication_length)
                    arms[w++] = arms[a];
            n_arms = w;
        }
        if (n_arms) { /* :182-200 */
            int any_active = 0;
            for (uint64_t a = 0; a < n_arms; ++a) any_active |= arms[a].active;
            if (!any_active) {
                uint64_t before = r->n_sd;
                for (uint64_t a = 0; a < n_arms; ++a)
                    if (arms[a].r_end - arms[a].r_start >= st->min_duplication_length) {
                        oracle_proto_sd sd = {arms[a].l_start, arms[a].r_start,
                                              arms[a].l_end - arms[a].l_start,
                                              arms[a].r_end - arms[a].r_start};
                        if (families_push_sd(r, sd)) {
                            rc = -2;
                            goto done;
                        }
                    }
                if (r->n_sd != before) {
                    if (families_close_family(r)) {
                        rc = -2;
                        goto done;
                    }
                    if (stats) {
                        stats->families++;
                        stats->proto_sds += r->n_sd - before;
                    }
                }
                n_arms = 0;
            }
        }
    }
    /* :201-203 -- arms still alive here are dropped, never flushed */
done:
    free(arms);
    free(todo);
    free(raw.v);
    free(hits.v);
    return rc;
}

int64_t oracle_probe_hits(const uint8_t *needle, uint64_t needle_len, uint64_t needle_offset,
                          const uint8_t *strand, int64_t n, const int64_t *sa,
                          const oracle_searcher *searcher, const oracle_settings *st,
                          uint8_t *status, uint64_t *row_offsets, uint64_t *hits_out,
                          uint64_t *n_hits) {
    const uint64_t k = st->probe_size, step = k / 2;
    *n_hits = 0;
    if (needle_len < st->min_duplication_length || needle_len < k + step || step == 0) return 0;
    u64vec raw = {0}, hits = {0};
    int64_t n_probes = 0;
    uint64_t total = 0, i = 0;
    while (i < needle_len - k - step) {
        i += step;
        int s = probe_filtered_hits(needle, needle_len, needle_offset, i, strand, n, sa, searcher,
                                    st, &raw, &hits, NULL);
        if (s < 0) {
            n_probes = s;
            break;
        }
        if (hits_out) {
            status[n_probes] = (uint8_t)s;
            row_offsets[n_probes] = total;
            if (s == 0) memcpy(hits_out + total, hits.v, hits.n * sizeof(uint64_t));
        }
        if (s == 0) total += hits.n;
        ++n_probes;
    }
    if (hits_out && n_probes >= 0) row_offsets[n_probes] = total;
    *n_hits = total;
    free(raw.v);
    free(hits.v);
    return n_probes;
}

/* utils::complement_nucleotide / complemented, src/utils.rs:1-23 */
void oracle_complemented(const uint8_t *in, uint8_t *out, uint64_t len) {
    for (uint64_t j = 0; j < len; ++j) {
        uint8_t c;
        switch (in[j]) {
            case 'A': c = 'T'; break;
            case 'T': c = 'A'; break;
            case 'G': c = 'C'; break;
            case 'C': c = 'G'; break;
            case 'N': c = 'N'; break;
            case 'a': c = 't'; break;
            case 't': c = 'a'; break;
            case 'g': c = 'c'; break;
            case 'c': c = 'g'; break;
            case 'n': c = 'n'; break;
            default: c = 'N'; break;
        }
        out[j] = c;
    }
}

static void stats_add(oracle_stats *dst, const oracle_stats *src) {
    uint64_t *d = (uint64_t *)dst;
    const uint64_t *s = (const uint64_t *)src;
    for (size_t j = 0; j < sizeof(oracle_stats) / sizeof(uint64_t); ++j) d[j] += s[j];
}

/* SearchDuplications::run, src/bin/asgart.rs:201-253 */
int32_t oracle_run(const uint8_t *strand, int64_t n, const int64_t *sa,
                   const oracle_searcher *searcher, const uint64_t *chunks, int64_t n_chunks,
                   const oracle_settings *st, int32_t threads, volatile uint64_t *progress,
                   oracle_stats *stats, oracle_families **out) {
    oracle_families **per_chunk = (oracle_families **)calloc((size_t)(n_chunks ? n_chunks : 1),
                                                             sizeof(*per_chunk));
    oracle_stats *per_stats = (oracle_stats *)calloc((size_t)(n_chunks ? n_chunks : 1),
                                                     sizeof(*per_stats));
    int32_t rc = 0;
    if (!per_chunk || !per_stats) {
        free(per_chunk);
        free(per_stats);
        return -2;
    }
    for (int64_t c = 0; c < n_chunks; ++c)
        if (chunks[2 * c] + chunks[2 * c + 1] > (uint64_t)n) rc = -1;
    if (rc) {
        free(per_chunk);
        free(per_stats);
        return rc;
    }
#ifdef _OPENMP
    if (threads > 1) omp_set_num_threads(threads);
#pragma omp parallel for schedule(dynamic) if (threads > 1)
#endif
    for (int64_t c = 0; c < n_chunks; ++c) { /* chunks.par_iter().enumerate(), :201-205 */
        uint64_t start = chunks[2 * c], len = chunks[2 * c + 1];
        const uint8_t *needle = strand + start;
        uint8_t *owned = NULL;
        if (st->reverse || st->complement) { /* :206-218 */
            owned = (uint8_t *)malloc(len ? len : 1);
            if (!owned) {
#ifdef _OPENMP
#pragma omp atomic write
#endif
                rc = -2;
                continue;
            }
            if (st->complement) oracle_complemented(strand + start, owned, len);
            else memcpy(owned, strand + start, len);
            if (st->reverse)
                for (uint64_t a = 0, b = len; a + 1 < b; ++a) {
                    --b;
                    uint8_t tmp = owned[a];
                    owned[a] = owned[b];
                    owned[b] = tmp;
                }
            needle = owned;
        }
        oracle_families *f = NULL;
        int32_t r1 = oracle_search_duplications(needle, len, start, strand, n, sa, searcher,
                                                progress ? progress + c : NULL, st,
                                                &per_stats[c], &f);
        if (r1 != 0) {
#ifdef _OPENMP
#pragma omp atomic write
#endif
            rc = r1;
        }
        if (f) /* left-coordinate fix-up, :226-237 */
            for (uint64_t j = 0; j < f->n_sd; ++j) {
                if (!st->reverse) f->sds[j].left += start;
                else f->sds[j].left = start + len - f->sds[j].left - f->sds[j].left_length;
            }
        per_chunk[c] = f;
        free(owned);
    }
    /* fold in chunk order, :241-253 (the reversed/complemented stamps are
     * constants of the run and live in the settings) */
    oracle_families *all = families_new();
    if (!all) rc = -2;
    for (int64_t c = 0; c < n_chunks; ++c) {
        oracle_families *f = per_chunk[c];
        if (f && all && rc == 0)
            for (uint64_t fam = 0; fam < f->n_fam; ++fam) {
                for (uint64_t j = f->fam_offsets[fam]; j < f->fam_offsets[fam + 1]; ++j)
                    if (families_push_sd(all, f->sds[j])) rc = -2;
                if (families_close_family(all)) rc = -2;
            }
        if (stats) stats_add(stats, &per_stats[c]);
        oracle_families_free(f);
    }
    free(per_chunk);
    free(per_stats);
    *out = all;
    return rc;
}

/* read_fasta's per-record normalisation, src/bin/asgart.rs:289-301:
 * without -S the record is upper-cased first; then masked letters become 'N'
 * under -S and anything outside ALPHABET becomes 'N'. */
void oracle_normalise(uint8_t *seq, uint64_t len, int32_t skip_masked) {
    for (uint64_t j = 0; j < len; ++j) {
        uint8_t c = seq[j];
        if (!skip_masked && c >= 'a' && c <= 'z') c = (uint8_t)(c - 'a' + 'A');
        int masked = (c == 'a' || c == 't' || c == 'g' || c == 'c' || c == 'n');
        int in_alpha = (c == 'A' || c == 'T' || c == 'G' || c == 'C' || c == 'N');
        if (masked && skip_masked) c = 'N';
        else if (!in_alpha) c = 'N';
        seq[j] = c;
    }
}

/* find_chunks_to_process, src/bin/asgart.rs:317-366 */
int64_t oracle_find_chunks(const uint8_t *strand, uint64_t len, uint64_t *chunks, int64_t cap) {
    const uint64_t threshold = 5000;
    uint64_t start = 0, count = 0, i = 0;
    int64_t n_chunks = 0;
#define PUSH(s, c)                                   \
    do {                                             \
        if (chunks && n_chunks < cap) {              \
            chunks[2 * n_chunks] = (s);              \
            chunks[2 * n_chunks + 1] = (c);          \
        }                                            \
        ++n_chunks;                                  \
    } while (0)
    while (i < len) {
        uint8_t c = strand[i];
        if (c == 'n' || c == 'N') {
            uint64_t n_count = 0;
            while (i + n_count < len && (strand[i + n_count] == 'n' || strand[i + n_count] == 'N'))
                ++n_count;
            if (n_count > threshold) {
                if (count > 0) {
                    PUSH(start, count);
                    count = 0;
                }
                start = i + n_count;
            } else {
                count += n_count;
            }
            i += n_count;
        } else {
            if (count == 0) {
                count = 1;
                start = i;
            } else {
                count += 1;
            }
            i += 1;
        }
    }
    if (count != 0) PUSH(start, count);
    if (n_chunks == 0) PUSH(0, len);
#undef PUSH
    return n_chunks;
}
