/*
 * oracle/postprocess.c -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).
 *
 * Literal C restatement of the reference's post-processing steps that follow the search step
 * (src/bin/asgart.rs:738-747 fixes the order FilterNs, ReOrder, ReduceOverlap, [ComputeScore], Sort):
 *   FilterNs       src/bin/asgart.rs:81-96  + ProtoSD::n_content src/structs.rs:454-467
 *   ReOrder        src/bin/asgart.rs:33-51
 *   ReduceOverlap  src/bin/asgart.rs:67-79,481-562 (subsegment, overlap, merge, reduce_overlap)
 *   Sort           src/bin/asgart.rs:53-65 (stable sort by `left`)
 * Quirks kept: n_content counts over the INCLUSIVE ranges [p ..= p+len] but divides by len, in f32;
 * ReOrder swaps the positions but not the lengths; merge() uses x.left_length for both arms of x
 * and y.right_length for both arms of y.
 */
#include "asgart_oracle.h"

#include <stdlib.h>
#include <string.h>

typedef oracle_proto_sd sd_t;

static float n_content(const sd_t *sd, const uint8_t *strand) { /* structs.rs:454-467 */
    uint64_t cl = 0, cr = 0;
    for (uint64_t p = sd->left; p <= sd->left + sd->left_length; ++p)
        cl += (strand[p] == 'n' || strand[p] == 'N');
    for (uint64_t p = sd->right; p <= sd->right + sd->right_length; ++p)
        cr += (strand[p] == 'n' || strand[p] == 'N');
    float a = (float)cl / (float)sd->left_length;
    float b = (float)cr / (float)sd->right_length;
    return a > b ? a : b; /* f32::max */
}

static int subsegment(uint64_t xs, uint64_t xl, uint64_t ys, uint64_t yl) { /* asgart.rs:482-487 */
    return xs >= ys && xs + xl <= ys + yl;
}

static int overlap(uint64_t xs, uint64_t xl, uint64_t ys, uint64_t yl) { /* asgart.rs:489-495 */
    uint64_t xe = xs + xl, ye = ys + yl;
    return (xs >= ys && xs <= ye && xe >= ye) || (ys >= xs && ys <= xe && ye >= xe);
}

static sd_t merge(const sd_t *x, const sd_t *y) { /* asgart.rs:497-513 */
    sd_t z;
    uint64_t a, b;
    z.left = x->left < y->left ? x->left : y->left;
    a = x->left + x->left_length;
    b = y->left + y->right_length;
    z.left_length = (a > b ? a : b) - z.left;
    z.right = x->right < y->right ? x->right : y->right;
    a = x->right + x->left_length;
    b = y->right + y->right_length;
    z.right_length = (a > b ? a : b) - z.right;
    return z;
}

/* _reduce, asgart.rs:516-549: out has room for n entries; returns the new count */
static uint64_t reduce_once(const sd_t *in, uint64_t n, sd_t *out) {
    uint64_t m = 0;
    for (uint64_t i = 0; i < n; ++i) {
        const sd_t *x = &in[i];
        int inserted = 0;
        for (uint64_t j = 0; j < m && !inserted; ++j) {
            sd_t *y = &out[j];
            if (subsegment(x->left, x->left_length, y->left, y->left_length) &&
                subsegment(x->right, x->right_length, y->right, y->right_length)) {
                inserted = 1; /* x inside y */
            } else if (subsegment(y->left, y->left_length, x->left, x->left_length) &&
                       subsegment(y->right, y->right_length, x->right, x->right_length)) {
                *y = *x; /* x contains y */
                inserted = 1;
            } else if (overlap(x->left, x->left_length, y->left, y->left_length) &&
                       overlap(x->right, x->right_length, y->right, y->right_length)) {
                sd_t z = merge(x, y);
                *y = z;
                inserted = 1;
            }
        }
        if (!inserted) out[m++] = *x;
    }
    return m;
}

/* In: families as (fam_offsets[n_fam+1], sds).  Out: a new oracle_families handle holding the
 * post-processed families.  strand = the '$'-terminated text. */
int32_t oracle_postprocess(const uint8_t *strand, const uint64_t *fam_offsets, uint64_t n_fam,
                           const oracle_proto_sd *sds, oracle_families **out) {
    uint64_t max_fam = 0;
    for (uint64_t f = 0; f < n_fam; ++f)
        if (fam_offsets[f + 1] - fam_offsets[f] > max_fam) max_fam = fam_offsets[f + 1] - fam_offsets[f];
    sd_t *a = (sd_t *)malloc((max_fam ? max_fam : 1) * sizeof(sd_t));
    sd_t *b = (sd_t *)malloc((max_fam ? max_fam : 1) * sizeof(sd_t));
    uint64_t *offs = (uint64_t *)malloc((n_fam + 1) * sizeof(uint64_t));
    uint64_t total = fam_offsets[n_fam];
    sd_t *res = (sd_t *)malloc((total ? total : 1) * sizeof(sd_t));
    if (!a || !b || !offs || !res) {
        free(a); free(b); free(offs); free(res);
        return -2;
    }
    uint64_t n_out_fam = 0, n_out = 0;
    offs[0] = 0;
    for (uint64_t f = 0; f < n_fam; ++f) {
        /* FilterNs: retain n_content <= 0.2, drop empty families (:87-95) */
        uint64_t n = 0;
        for (uint64_t j = fam_offsets[f]; j < fam_offsets[f + 1]; ++j)
            if (n_content(&sds[j], strand) <= 0.2f) a[n++] = sds[j];
        if (n == 0) continue;
        /* ReOrder (:39-50) */
        for (uint64_t j = 0; j < n; ++j)
            if (a[j].left > a[j].right) {
                uint64_t t = a[j].left;
                a[j].left = a[j].right;
                a[j].right = t;
            }
        /* ReduceOverlap (:551-562) */
        uint64_t old_size = n, new_size = reduce_once(a, n, b);
        sd_t *cur = b, *other = a;
        while (new_size < old_size) {
            old_size = new_size;
            new_size = reduce_once(cur, old_size, other);
            sd_t *t = cur; cur = other; other = t;
        }
        /* Sort: stable by left (:59-64) -- insertion sort keeps equal keys in order */
        for (uint64_t i = 1; i < new_size; ++i) {
            sd_t key = cur[i];
            uint64_t j = i;
            while (j > 0 && cur[j - 1].left > key.left) {
                cur[j] = cur[j - 1];
                --j;
            }
            cur[j] = key;
        }
        memcpy(res + n_out, cur, new_size * sizeof(sd_t));
        n_out += new_size;
        offs[++n_out_fam] = n_out;
    }
    int32_t rc = oracle_families_from_arrays(offs, n_out_fam, res, out);
    free(a); free(b); free(offs); free(res);
    return rc;
}


/* ---- ComputeScore: ProtoSD::levenshtein, src/structs.rs:439-452 -------------------------------
 * bio::alignment::distance::levenshtein (crate `bio`, version unpinned in the reference) is the
 * textbook unit-cost global edit distance; restated with two rolling rows. */
static uint8_t tr_complement(uint8_t c) { /* src/structs.rs:11-26 (TR) */
    switch (c) {
    case 'A': return 'T'; case 'T': return 'A'; case 'G': return 'C'; case 'C': return 'G';
    case 'a': return 't'; case 't': return 'a'; case 'g': return 'c'; case 'c': return 'g';
    default: return c; /* 'N'/'n' map to themselves; the reference panics on anything else */
    }
}

double oracle_levenshtein_identity(const uint8_t *strand, uint64_t left, uint64_t right, uint64_t left_length,
                                   uint64_t right_length, int32_t reversed, int32_t complemented) {
    const uint64_t la = left_length + 1, lb = right_length + 1; /* inclusive ranges, :441-442 */
    uint8_t *b = (uint8_t *)malloc(lb);
    uint32_t *prev = (uint32_t *)malloc((lb + 1) * sizeof(uint32_t));
    uint32_t *cur = (uint32_t *)malloc((lb + 1) * sizeof(uint32_t));
    for (uint64_t j = 0; j < lb; ++j) b[j] = strand[right + j];
    if (reversed) /* :443-445 */
        for (uint64_t j = 0; j < lb / 2; ++j) {
            uint8_t t = b[j];
            b[j] = b[lb - 1 - j];
            b[lb - 1 - j] = t;
        }
    if (complemented) /* :446-448 */
        for (uint64_t j = 0; j < lb; ++j) b[j] = tr_complement(b[j]);
    for (uint64_t j = 0; j <= lb; ++j) prev[j] = (uint32_t)j;
    for (uint64_t i = 1; i <= la; ++i) {
        cur[0] = (uint32_t)i;
        const uint8_t a = strand[left + i - 1];
        for (uint64_t j = 1; j <= lb; ++j) {
            uint32_t v = prev[j - 1] + (a != b[j - 1] ? 1u : 0u);
            if (prev[j] + 1 < v) v = prev[j] + 1;
            if (cur[j - 1] + 1 < v) v = cur[j - 1] + 1;
            cur[j] = v;
        }
        uint32_t *t = prev; prev = cur; cur = t;
    }
    const double dist = (double)prev[lb];
    free(b); free(prev); free(cur);
    const uint64_t longest = left_length > right_length ? left_length : right_length;
    return 100.0 * (1.0 - dist / (double)longest); /* :451 */
}
