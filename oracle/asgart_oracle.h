/*
 * oracle/asgart_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of ASGART's probe -> suffix-array search -> seed-extension
 * hot path (reference: delehef/asgart 2.5.1, /root/reference).  Only tests/,
 * __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may load this
 * library; the product (asgart_amd/) never does.
 *
 * PARITY UNPINNED: the reference is Rust (no cargo/rustc in this image), has
 * no tests, fixtures or golden vectors, and its libdivsufsort submodule is
 * absent, so this restatement cannot be checked against reference outputs.
 * It is pinned only against (a) a second, independent brute-force restatement
 * (tests/bruteforce.py) and (b) the hand-derived known-answer cases in
 * tests/golden/.  See oracle/README.md.
 */
#ifndef ASGART_ORACLE_H
#define ASGART_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* RunSettings fields that reach the hot path (src/structs.rs:36-58).
 * max_gap_size already includes +probe_size (src/bin/asgart.rs:681). */
typedef struct {
    uint64_t probe_size;
    uint32_t max_gap_size;
    uint64_t min_duplication_length;
    uint64_t max_cardinality;
    uint8_t reverse;
    uint8_t complement;
} oracle_settings;

/* ProtoSD (src/structs.rs:418-429) minus identity (always 0 on this path) and
 * the two flags (constant per run: src/bin/asgart.rs:245-247). */
typedef struct {
    uint64_t left, right, left_length, right_length;
} oracle_proto_sd;

/* Work counters for the roofline yardstick. */
typedef struct {
    uint64_t probes_total;      /* loop iterations of automaton.rs:96-97          */
    uint64_t probes_n_skipped;  /* `needle[i]=='N'` continue, automaton.rs:100     */
    uint64_t probes_searched;   /* Searcher::search calls                          */
    uint64_t probes_card_skipped; /* `matches.len() > max_cardinality`, :115        */
    uint64_t probes_with_hits;  /* processed probes with >=1 filtered hit          */
    uint64_t bisect_steps;      /* sum over searched probes of ceil(log2(b_p+1))   */
    uint64_t raw_hits;          /* sum of h_p (interval sizes)                     */
    uint64_t filtered_hits;     /* sum of h'_p over processed probes               */
    uint64_t arm_tests;         /* predicate evaluations in try_extend_arms        */
    uint64_t proto_sds;         /* emitted ProtoSD                                 */
    uint64_t families;          /* emitted families                                */
} oracle_stats;

/* ---- suffix array (oracle/sais.c) -------------------------------------- */
int32_t oracle_divsufsort64(const uint8_t *T, int64_t *SA, int64_t n);
int64_t oracle_sa_check(const uint8_t *T, const int64_t *SA, int64_t n);

/* ---- Searcher (src/searcher.rs:94-180) --------------------------------- */
typedef struct oracle_searcher oracle_searcher;
oracle_searcher *oracle_searcher_new(const uint8_t *dna, int64_t n, const int64_t *sa,
                                     int64_t sa_len, uint64_t offset);
void oracle_searcher_free(oracle_searcher *s);
/* cache entry for an 8-mer over ALPHABET; returns 0 and fills lo/hi, or -1 if
 * the 8 bytes are not all in ALPHABET (the reference panics there). */
int32_t oracle_searcher_cache_get(const oracle_searcher *s, const uint8_t *p8, uint64_t *lo,
                                  uint64_t *hi);
/* Searcher::search: writes up to cap hit *starts* (SA order) and returns the
 * full hit count; -1 if the pattern's first 8 bytes are outside ALPHABET.
 * Optional out-params: SA slot range [*range_lo,*range_hi) and bucket size. */
int64_t oracle_searcher_search(const oracle_searcher *s, const uint8_t *dna, int64_t n,
                               const int64_t *sa, const uint8_t *pattern, int64_t k,
                               uint64_t *out_starts, int64_t cap, uint64_t *range_lo,
                               uint64_t *range_hi, uint64_t *bucket_size);

/* ---- results ------------------------------------------------------------ */
typedef struct oracle_families oracle_families;
void oracle_families_counts(const oracle_families *f, uint64_t *n_families, uint64_t *n_sds);
/* fam_offsets has n_families+1 entries; sds has n_sds entries */
void oracle_families_copy(const oracle_families *f, uint64_t *fam_offsets, oracle_proto_sd *sds);
void oracle_families_free(oracle_families *f);

int32_t oracle_families_from_arrays(const uint64_t *fam_offsets, uint64_t n_fam,
                                    const oracle_proto_sd *sds, oracle_families **out);
/* FilterNs -> ReOrder -> ReduceOverlap -> Sort (src/bin/asgart.rs:33-96,481-562,738-747) */
int32_t oracle_postprocess(const uint8_t *strand, const uint64_t *fam_offsets, uint64_t n_fam,
                           const oracle_proto_sd *sds, oracle_families **out);

/* ProtoSD::levenshtein (src/structs.rs:439-452): identity in percent, as f64 */
double oracle_levenshtein_identity(const uint8_t *strand, uint64_t left, uint64_t right, uint64_t left_length,
                                   uint64_t right_length, int32_t reversed, int32_t complemented);

/* ---- automaton::search_duplications for ONE needle (src/automaton.rs:57-204).
 * `left` in the result is needle-local, `right` global, exactly as the
 * reference returns them.  progress may be NULL.  Returns 0 or <0. */
int32_t oracle_search_duplications(const uint8_t *needle, uint64_t needle_len,
                                   uint64_t needle_offset, const uint8_t *strand, int64_t n,
                                   const int64_t *sa, const oracle_searcher *searcher,
                                   volatile uint64_t *progress, const oracle_settings *settings,
                                   oracle_stats *stats, oracle_families **out);

/* ---- SearchDuplications::run body, src/bin/asgart.rs:201-253: all chunks,
 * needle preparation, left-coordinate fix-up, concatenation in chunk order.
 * chunks = n_chunks (start,len) pairs.  threads<=1: serial; otherwise OpenMP
 * `parallel for schedule(dynamic)` over chunks (== rayon par_iter, :201-205). */
int32_t oracle_run(const uint8_t *strand, int64_t n, const int64_t *sa,
                   const oracle_searcher *searcher, const uint64_t *chunks, int64_t n_chunks,
                   const oracle_settings *settings, int32_t threads, volatile uint64_t *progress,
                   oracle_stats *stats, oracle_families **out);

/* Per-probe filtered hit lists for one needle, as the automaton sees them
 * (automaton.rs:96-117): for probe index j (i = (j+1)*step), status[j] is
 * 0 = processed, 1 = skipped 'N', 2 = skipped cardinality; row j of the CSR
 * holds the filtered hits (SA order) for status 0 and is empty otherwise.
 * Two-call: pass hits=NULL to get sizes (returns number of probes; *n_hits). */
int64_t oracle_probe_hits(const uint8_t *needle, uint64_t needle_len, uint64_t needle_offset,
                          const uint8_t *strand, int64_t n, const int64_t *sa,
                          const oracle_searcher *searcher, const oracle_settings *settings,
                          uint8_t *status, uint64_t *row_offsets, uint64_t *hits,
                          uint64_t *n_hits);

/* ---- input preparation (src/bin/asgart.rs:278-366, src/utils.rs:1-23) -- */
/* read_fasta's per-record normalisation, in place (asgart.rs:289-301) */
void oracle_normalise(uint8_t *seq, uint64_t len, int32_t skip_masked);
/* find_chunks_to_process (asgart.rs:317-366); two-call (chunks=NULL -> count) */
int64_t oracle_find_chunks(const uint8_t *strand, uint64_t len, uint64_t *chunks, int64_t cap);
/* utils::complemented (utils.rs:1-23) */
void oracle_complemented(const uint8_t *in, uint8_t *out, uint64_t len);
/* d_ss (automaton.rs:207-216) exposed for known-answer tests */
int64_t oracle_d_ss(uint64_t a_start, uint64_t a_end, uint64_t m_start, uint64_t m_end);

#ifdef __cplusplus
}
#endif
#endif
