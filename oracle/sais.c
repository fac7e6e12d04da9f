/*
 * oracle/sais.c -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).
 *
 * Suffix-array construction for the CPU oracle.  Replaces the reference's
 * call into libdivsufsort (`divsufsort64`, /root/reference/src/divsufsort.rs:10,
 * called from /root/reference/src/bin/asgart.rs:473-479).  libdivsufsort itself
 * is an un-vendored submodule (empty directory in /root/reference), so this is
 * NOT a restatement of its code: the suffix array of a text is mathematically
 * unique (bytewise lexicographic order of all suffixes, a shorter suffix that
 * is a prefix of a longer one sorting first), so any correct algorithm yields
 * the same array.  This file implements induced sorting (SA-IS, Nong/Zhang/Chan
 * 2009, the published algorithm) with a *virtual* sentinel, so it is correct
 * for arbitrary byte strings, not just '$'-terminated ones.
 *
 * `oracle_sa_check` is an independent O(n) verifier used by the tests.
 */
#include "asgart_oracle.h"

#include <stdlib.h>
#include <string.h>

typedef int64_t idx_t;

#define CHR(i) (cs == 1 ? (idx_t)((const uint8_t *)T)[(i)] : ((const idx_t *)T)[(i)])
#define TYPE_L 0
#define TYPE_S 1

static void count_chars(const void *T, idx_t *C, idx_t n, idx_t K, int cs) {
    memset(C, 0, (size_t)K * sizeof(idx_t));
    for (idx_t i = 0; i < n; ++i) C[CHR(i)]++;
}

static void bucket_bounds(const idx_t *C, idx_t *B, idx_t K, int ends) {
    idx_t sum = 0;
    for (idx_t c = 0; c < K; ++c) {
        sum += C[c];
        B[c] = ends ? sum : sum - C[c];
    }
}

/* One L-pass followed by one S-pass over SA (empty slots hold -1). */
static void induce(const void *T, idx_t *SA, const uint8_t *t, const idx_t *C, idx_t *B,
                   idx_t n, idx_t K, int cs) {
    bucket_bounds(C, B, K, 0);
    /* the virtual sentinel (position n) is the smallest suffix; its predecessor
     * n-1 is always L-type */
    SA[B[CHR(n - 1)]++] = n - 1;
    for (idx_t i = 0; i < n; ++i) {
        idx_t j = SA[i];
        if (j > 0 && t[j - 1] == TYPE_L) SA[B[CHR(j - 1)]++] = j - 1;
    }
    bucket_bounds(C, B, K, 1);
    for (idx_t i = n - 1; i >= 0; --i) {
        idx_t j = SA[i];
        if (j > 0 && t[j - 1] == TYPE_S) SA[--B[CHR(j - 1)]] = j - 1;
    }
}

static int sais_rec(const void *T, idx_t *SA, idx_t n, idx_t K, int cs) {
    if (n == 0) return 0;
    if (n == 1) {
        SA[0] = 0;
        return 0;
    }
    uint8_t *t = (uint8_t *)malloc((size_t)n);
    idx_t *C = (idx_t *)malloc((size_t)K * sizeof(idx_t));
    idx_t *B = (idx_t *)malloc((size_t)K * sizeof(idx_t));
    if (!t || !C || !B) {
        free(t);
        free(C);
        free(B);
        return -2;
    }
    t[n - 1] = TYPE_L;
    for (idx_t i = n - 2; i >= 0; --i) {
        idx_t a = CHR(i), b = CHR(i + 1);
        t[i] = (a < b || (a == b && t[i + 1] == TYPE_S)) ? TYPE_S : TYPE_L;
    }
#define IS_LMS(i) ((i) > 0 && t[(i)] == TYPE_S && t[(i) - 1] == TYPE_L)

    /* stage 1: sort the LMS substrings */
    count_chars(T, C, n, K, cs);
    for (idx_t i = 0; i < n; ++i) SA[i] = -1;
    bucket_bounds(C, B, K, 1);
    for (idx_t i = 1; i < n; ++i)
        if (IS_LMS(i)) SA[--B[CHR(i)]] = i;
    induce(T, SA, t, C, B, n, K, cs);

    idx_t m = 0;
    for (idx_t i = 0; i < n; ++i) {
        idx_t p = SA[i];
        if (IS_LMS(p)) SA[m++] = p;
    }
    for (idx_t i = m; i < n; ++i) SA[i] = -1;

    /* name the sorted LMS substrings; names parked at SA[m + pos/2] */
    idx_t names = 0, prev = -1;
    for (idx_t i = 0; i < m; ++i) {
        idx_t pos = SA[i];
        int diff = 0;
        if (prev < 0) {
            diff = 1;
        } else {
            for (idx_t d = 0;; ++d) {
                idx_t p1 = pos + d, p2 = prev + d;
                if (p1 >= n || p2 >= n) { /* one of them runs into the sentinel */
                    diff = 1;
                    break;
                }
                if (CHR(p1) != CHR(p2) || t[p1] != t[p2]) {
                    diff = 1;
                    break;
                }
                if (d > 0) {
                    int l1 = IS_LMS(p1), l2 = IS_LMS(p2);
                    if (l1 || l2) {
                        diff = !(l1 && l2);
                        break;
                    }
                }
            }
        }
        if (diff) {
            ++names;
            prev = pos;
        }
        SA[m + pos / 2] = names - 1;
    }
    {
        idx_t j = n - 1;
        for (idx_t i = n - 1; i >= m; --i)
            if (SA[i] >= 0) SA[j--] = SA[i];
    }
    idx_t *SA1 = SA, *s1 = SA + n - m;
    int rc = 0;
    if (names < m) {
        rc = sais_rec(s1, SA1, m, names, 8);
    } else {
        for (idx_t i = 0; i < m; ++i) SA1[s1[i]] = i;
    }
    if (rc == 0) {
        /* stage 3: induce the full order from the sorted LMS suffixes */
        idx_t j = 0;
        for (idx_t i = 1; i < n; ++i)
            if (IS_LMS(i)) s1[j++] = i;
        for (idx_t i = 0; i < m; ++i) SA1[i] = s1[SA1[i]];
        for (idx_t i = m; i < n; ++i) SA[i] = -1;
        bucket_bounds(C, B, K, 1);
        for (idx_t i = m - 1; i >= 0; --i) {
            idx_t p = SA[i];
            SA[i] = -1;
            SA[--B[CHR(p)]] = p;
        }
        induce(T, SA, t, C, B, n, K, cs);
    }
#undef IS_LMS
    free(t);
    free(C);
    free(B);
    return rc;
}

/* Same signature and contract as divsufsort64 (src/divsufsort.rs:10):
 * caller-allocated SA of n entries, returns 0 on success, <0 on error. */
int32_t oracle_divsufsort64(const uint8_t *T, int64_t *SA, int64_t n) {
    if (n < 0 || (n > 0 && (!T || !SA))) return -1;
    return sais_rec(T, SA, n, 256, 1);
}

/* O(n) verifier: SA is a permutation of 0..n-1 and adjacent suffixes are in
 * strictly increasing bytewise order (rank trick).  0 = ok, >0 = first bad slot+1,
 * <0 = alloc failure / bad permutation. */
int64_t oracle_sa_check(const uint8_t *T, const int64_t *SA, int64_t n) {
    if (n == 0) return 0;
    int64_t *rank = (int64_t *)malloc((size_t)(n + 1) * sizeof(int64_t));
    if (!rank) return -2;
    for (int64_t i = 0; i <= n; ++i) rank[i] = -2;
    for (int64_t r = 0; r < n; ++r) {
        int64_t x = SA[r];
        if (x < 0 || x >= n || rank[x] != -2) {
            free(rank);
            return -1;
        }
        rank[x] = r;
    }
    rank[n] = -1; /* empty suffix sorts first */
    for (int64_t r = 1; r < n; ++r) {
        int64_t a = SA[r - 1], b = SA[r];
        int ok = T[a] < T[b] || (T[a] == T[b] && rank[a + 1] < rank[b + 1]);
        if (!ok) {
            free(rank);
            return r + 1;
        }
    }
    free(rank);
    return 0;
}
