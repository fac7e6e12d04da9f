/*
 * asgart_hip.h -- C ABI of the MI355X-native segmental-duplication search core.
 *
 * Drop-in boundary for ONE path of delehef/asgart 2.5.1: the body of
 * `SearchDuplications::run` (reference src/bin/asgart.rs:137-258), i.e.
 *   suffix array  ->  Searcher (8-mer interval cache + k-mer equal range)
 *   ->  automaton::search_duplications over every chunk  ->  left fix-up.
 * Everything is plain pointers + sizes in the style of the reference's own FFI
 * to libdivsufsort (reference src/divsufsort.rs:8-33): extern "C", caller-owned
 * inputs, caller-allocated outputs (two-call count/copy), int32 status
 * (0 ok, <0 error), no callbacks, no torch types.
 *
 * All compute runs in hand-written HIP kernels for gfx950; there is NO CPU
 * fallback: every entry point that needs the GPU fails with ASGART_E_HIP when
 * no device is usable.
 *
 * The reference-side binding (Rust `extern "C"` block + build.rs line) is shown
 * in INTEGRATION.md.
 */
#ifndef ASGART_HIP_H
#define ASGART_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ASGART_OK 0
#define ASGART_E_ARG (-1)  /* bad argument / unsupported setting            */
#define ASGART_E_OOM (-2)  /* host or device allocation failed              */
#define ASGART_E_HIP (-3)  /* HIP runtime error, or no usable gfx950 device */
#define ASGART_E_CAP (-4)  /* an internal capacity was exceeded (see msg)   */

/* RunSettings as it reaches the path (reference src/structs.rs:36-58; Copy).
 * max_gap_size already includes +probe_size (reference src/bin/asgart.rs:681). */
typedef struct asgart_settings {
    uint64_t probe_size;              /* -k ; 8 <= k <= 128 in this build         */
    uint32_t max_gap_size;            /* -g + -k                                 */
    uint64_t min_duplication_length;  /* --min-length                            */
    uint64_t max_cardinality;         /* --max-cardinality.  The reference keeps the arms of a chunk in an unbounded
                                         Vec (src/automaton.rs:87).  Here the live arms of one automaton segment never
                                         exceed max_cardinality * (ceil(max_gap_size / (probe_size/2)) + 1) -- 6500 at
                                         the defaults -- and the last extension tier sizes its HBM slices for exactly
                                         that bound, whatever the settings (ASGART_E_CAP only if it reaches 2^24).   */
    uint8_t reverse;                  /* -R                                      */
    uint8_t complement;               /* -C                                      */
} asgart_settings;

/* ProtoSD (reference src/structs.rs:418-429).  `left` is already global (the
 * fix-up of src/bin/asgart.rs:229-237 is applied inside); identity is 0.0 and
 * reversed/complemented equal the settings of the call (:245-247). */
typedef struct asgart_proto_sd {
    uint64_t left, right, left_length, right_length;
} asgart_proto_sd;

/* Device-side timings (HIP events on the library's own stream) and work
 * counters of the LAST asgart_search_duplications / asgart_probe_hits call. */
typedef struct asgart_stats {
    double ms_total;        /* first kernel -> results on host                   */
    double ms_search;       /* probe-search + count kernels (dominant, HBM-bound) */
    double ms_scan;         /* prefix scans + segment detection                   */
    double ms_fill;         /* hit materialisation (CSR fill)                     */
    double ms_extend;       /* placement + all extension tiers + cascade            */
    uint64_t probes_total;
    uint64_t probes_n_skipped;
    uint64_t probes_searched;
    uint64_t probes_card_skipped;
    uint64_t probes_with_hits;
    uint64_t raw_hits;      /* sum of SA-interval sizes h_p (filled only when ASGART_STATS_RAW_HITS or
                               ASGART_STATS_YARDSTICK was requested: an untimed pass over the call's probes) */
    uint64_t filtered_hits; /* CSR size                                            */
    uint64_t segments;      /* independent automaton instances                     */
    uint64_t families;
    uint64_t proto_sds;
    uint64_t bisect_steps;  /* sum ceil(log2(b_p+1)), b_p = 8-mer bucket (yardstick;
                               filled only when ASGART_STATS_YARDSTICK was requested) */
    uint64_t search_launches; /* number of launches of the dominant kernel          */
    uint64_t overflow_segments; /* segments a tier gave up on (re-run by a larger tier)  */
    double ms_extend_tier2;   /* part of ms_extend spent re-running those (cascade)    */
    uint64_t heavy_segments;  /* segments placed in tiers 3..7 (workgroup kernels)      */
    double ms_probe_count;    /* probe_count_kernel alone (first kernel of ms_search) */
    /* filled only with ASGART_STATS_YARDSTICK, by an untimed accounting pass over the same probes: */
    uint64_t search_bytes;    /* bytes the probe-search kernels load and store BY DESIGN (window
                                 staging, filter word, prefix-table entries, keys read by the bisection,
                                 suffix-array entries read, outputs): the algorithmic bytes of this kernel */
    uint64_t probes_filter_rejected; /* probes answered by the k-mer presence filter alone          */
    uint64_t search_bytes_wide_loads; /* ... of search_bytes, those loaded as whole-wave 16-byte-per-lane reads (text windows,
                                         filter bitmaps): what FETCH_SIZE counts at half on gfx950 (accounting pass only) */
    double ms_longest_tier;   /* part of ms_extend: the extension tier that ran longest, from the launch of the tiers
                                 (they run side by side) -- in practice the longest serial automaton segment of the
                                 call, i.e. what sharding the probes over more GPUs cannot shorten              */
    uint64_t passes;          /* passes (orientations) the call ran as ONE job: 1 for a plain call; n for the passes call
                                 when it fuses them (the counters above are then sums over the passes, the timings those
                                 of the one job)                                                                 */
    double ms_longest_segment; /* part of ms_extend: the longest time ONE automaton segment took on its workgroup -- the
                                  serial floor of the call: src/automaton.rs:96-201 is serial per chunk, and neither more
                                  compute units nor more GPUs shorten a segment (ms_longest_tier is the tier that FINISHED
                                  last, which is throughput when the tier holds many segments)                    */
    uint64_t split_segments;   /* long segments that ran as ranges side by side (option split) ...                */
    uint64_t split_refused;    /* ... and those of them with a cut that did not hold (the rest behind it ran as one more
                                  run, or the whole segment again)                                                 */
} asgart_stats;

typedef struct asgart_index asgart_index;
typedef struct asgart_families asgart_families;

/* Replaces `divsufsort64` (reference src/divsufsort.rs:10, call site
 * src/bin/asgart.rs:473-479): signature-identical.  Builds the suffix array of
 * T[0..n) on the GPU (prefix doubling) into the caller-allocated SA. */
int32_t asgart_sa_build64(const uint8_t *T, int64_t *SA, int64_t n);

/* Replaces `Searcher::new(dna, sa, 0)` plus the Arc-sharing of text and SA
 * (reference src/bin/asgart.rs:142-155, src/searcher.rs:99-143): uploads text
 * and suffix array to HBM of `device` and builds the search structures.
 * T must consist of bytes in {A,C,G,T,N} with at most one '$', as its last
 * byte (what prepare_data produces, src/bin/asgart.rs:289-301,430).
 * SA may be NULL: the library then builds it on the GPU itself. */
int32_t asgart_index_create(const uint8_t *T, int64_t n, const int64_t *SA, int64_t sa_len,
                            int32_t device, asgart_index **out);
void asgart_index_destroy(asgart_index *idx);

/* Replicates an index on another device (text + suffix array copied device to device -- over xGMI
 * between two GPUs -- instead of sorting the suffixes once per GPU; the probe_size-specific tables are
 * rebuilt there on first use).  Replaces nothing in the reference (it has one address space); it is the
 * "SA + text replicated in each HBM" step of the multi-GPU design. */
int32_t asgart_index_clone(asgart_index *src, int32_t device, asgart_index **out);

/* The same for a host that is one process per GPU (the multi-GPU arrangement of BASELINE.json's north star): the
 * process that built the suffix array hands out the device addresses of text and suffix array
 * (asgart_index_export: valid while the index lives, read-only), broadcasts them with its collective library
 * (ncclBroadcast over xGMI; asgart_amd/multi.py: replicate_index) and every other process creates its replica from
 * the received DEVICE buffers, which are copied (asgart_index_create_device; d_text: n bytes; d_sa: n entries of
 * sa_entry_bytes = 4, or 8 when n >= 2^32 - 256, as asgart_index_export reports).  No suffix sort and no host copy
 * per GPU.  The text is validated on the device like asgart_index_create validates it. */
int32_t asgart_index_export(asgart_index *idx, const void **d_text, const void **d_sa, int32_t *sa_entry_bytes);
int32_t asgart_index_create_device(const void *d_text, int64_t n, const void *d_sa, int64_t sa_len,
                                   int32_t sa_entry_bytes, int32_t device, asgart_index **out);
/* (d_sa == NULL: the suffixes are sorted on the GPU, as asgart_index_create does for SA == NULL.) */

/* Replaces prepare_data behind the FASTA reader (reference src/bin/asgart.rs:273-430; the reader itself, bio's, stays
 * with the host): records[r] / record_lens[r] are the raw sequence bytes of record r as read (all files' records in
 * order, :375-395).  On the GPU: alphabet normalisation (:289-301: upper-cased unless skip_masked -- then lower-case
 * bases become N --, everything outside {A,T,G,C,N} becomes N), find_chunks_to_process per record (:317-366: cut at
 * runs of more than 5000 N; a record without any piece left gives one chunk over all of it), concatenation and the
 * final '$' (:430).
 *   text_out   nullable; sum(record_lens) + 1 bytes: the prepared strand (Strand.data) for hosts that want it
 *   chunks     (start, len) pairs in global coordinates, record order -- what asgart_search_duplications takes;
 *              *n_chunks receives their number; with chunks_cap too small: ASGART_E_CAP and *n_chunks = the room needed
 *              (chunks_cap = 0: count only)
 *   index_out  nullable; an index over the prepared text, its suffixes sorted on the GPU (what asgart_index_create
 *              would build from text_out, without the text travelling to the host and back)
 * The strand map (record names / offsets, src/structs.rs:60-65) is the host's: offsets are the prefix sums of
 * record_lens. */
int32_t asgart_prepare_data(const uint8_t *const *records, const uint64_t *record_lens, int64_t n_records,
                            int32_t skip_masked, int32_t device, uint8_t *text_out, uint64_t *chunks,
                            int64_t chunks_cap, int64_t *n_chunks, asgart_index **index_out);

/* Device memory the library keeps for reuse (released blocks of 256 MiB and more, so that an index build does not
 * pay the runtime's slow first allocation after large frees) goes back to the device: at the end of
 * asgart_index_prepare, when the last index of a device is destroyed, when one of the library's own allocations
 * fails -- and here, for a host about to allocate a lot by other means.  Returns the bytes released, < 0 on error. */
int64_t asgart_trim_cache(int32_t device);

/* Diagnostics for a host that suspects a stalled call (no reference counterpart: the reference has no device to wait
 * for): writes the NATIVE call stack of every thread of the process to stderr -- glibc backtrace from the handler of
 * a realtime signal of the library's own (SIGRTMIN + 6), installed at the first call and never removed, one thread
 * after the other, each acknowledging for itself -- so that a wait inside the HIP runtime, which a Python- or
 * Rust-level stack dump shows as one opaque frame -- names the runtime call it sits in.  Callable from any thread,
 * in particular from a watchdog thread while the main thread is blocked (tests/conftest.py does on a test time-out).
 * Returns the number of threads asked. */
int32_t asgart_debug_dump_stacks(void);

/* `--trim START END` (reference src/bin/asgart.rs:142-148, validation :432-463, README "trimming"):
 * the suffix array covers only data[start..end] + '$' -- its entries shifted by +start -- and the WHOLE
 * input is then searched against it (Searcher::new(&strand.data, &suffix_array, 0), :151-155).  SA: that
 * shifted array with sa_len == end - start + 1 entries (what the reference's r_divsufsort + shift
 * produces), or NULL to build it on the GPU.  Requires 0 <= start < end <= n - 1 (the trim the
 * reference's prepare_data lets through).  The array is sorted by the sub-strand's suffixes but
 * compared through the full text, so the suffixes ending within k bases of `end` are out of place; the
 * library replays the reference's two bisections (libdivsufsort sa_search for the 8-mer cache,
 * superslice equal_range_by per probe) wherever one of them is in range -- results equal the oracle's. */
int32_t asgart_index_create_trim(const uint8_t *T, int64_t n, const int64_t *SA, int64_t sa_len,
                                 int64_t trim_start, int64_t trim_end, int32_t device,
                                 asgart_index **out);

/* Tuning and test options of an index.  Production code never needs this call: the defaults
 * are the tuned ones.  Each option can also be preset through the environment variable
 * ASGART_<NAME> (upper case), which is read ONCE, inside asgart_index_create -- the search path
 * itself never reads the environment.  Values are range-checked; ASGART_E_ARG for an unknown
 * name or a value out of range.  Names: shard_lookback, shard_lookahead (halo sizes of a sharded
 * call, in probes); force_tier, arms_kernel, long3, cap1, cap3_pct, cap45_pct, cap6_pct, cap6w_pct, dense3, dense6,
 * tier_order, grid1..grid7, solo (placement of segments on the extension kernels); barren (segments that provably emit
 * nothing are not run: 0 none, 1 by their number of hit-probes, 2 also by the positions of their hits); split, split_len,
 * split_runs, split_warm, split_warm_max, split_min (long segments run as ranges side by side, each checked against its
 * predecessor at the cut; where a cut does not hold the ranges in front of it stand, the rest of the segment runs as one
 * more run, and the next call over the same input starts that segment's ranges as far in front of their cuts as the
 * failed one asked for; split = 2: with 64-bit positions as well);
 * fuse_passes, fuse_pole_pct (passes of one call as one job or pipelined); kfilter_bits, posbits, rank_lists,
 * lazy_aux (the position filter and the position-sorted lists, and when they come into being); cache_calls, prewarm
 * (memory); watchdog_s; debug; test_cap_limit, test_genbits, test_k8_delay, test_fail_alloc, test_stall_s,
 * test_wide_batch (parity and failure tests).  RESULTS NEVER DEPEND ON ANY OF THEM.  The full table with ranges is
 * kOptions in asgart_amd/csrc/index.hip, every field is described in struct Options
 * (asgart_amd/csrc/index.hpp).  ptab_depth and force_wide are fixed at creation (environment
 * only).  Blocks until no call is in flight. */
int32_t asgart_index_set_option(asgart_index *idx, const char *name, int64_t value);

/* O(n) verifier of the suffix array held by the index, on the GPU: SA must be a permutation of
 * 0..n-1 whose adjacent suffixes are in strictly increasing bytewise order (the same rank trick as
 * the CPU oracle's checker).  Returns the number of violating slots (0 = valid), < 0 on error.
 * The reference never checks its suffix array (it even ignores divsufsort64's status,
 * src/bin/asgart.rs:475-477); this exists for the full-size parity tests, where the text is too
 * large for the CPU checker (n >= 2^32: 64-bit suffix numbers). */
int64_t asgart_index_check_sa(asgart_index *idx);

/* Optional: build the probe_size-specific search keys now (otherwise done
 * lazily by the first call that needs them; kept until another k is used). */
int32_t asgart_index_prepare(asgart_index *idx, uint64_t probe_size);

/* Replaces the body of SearchDuplications::run from the chunk fan-out to the
 * fold (reference src/bin/asgart.rs:201-253): for every chunk (start,len) runs
 * automaton::search_duplications (src/automaton.rs:57-204) on the prepared
 * needle and returns the families in (chunk, discovery) order.
 * chunks: n_chunks pairs (start, len).  progress: nullable, n_chunks entries; the reference stores
 * the needle offset of every probe as it goes (src/automaton.rs:98, Relaxed) for a progress bar that
 * polls every 500 ms (src/bin/asgart.rs:160-197).  Here all chunks advance together through a few
 * device-wide phases, so every entry jumps to its chunk's final offset at once: when every probe of the
 * call has been searched and its hits materialised (the HBM-bound, chip-wide part is over; the extension
 * automaton is under way) and again when the call returns.  A host that polls the array from another
 * thread can issue its next call at that moment: the new call's search phases then run beside this
 * call's extension, whose tail is a few serial segments on one compute unit each (bench.py pipelines the
 * -RC and the direct pass of a step that way: 338 instead of 441 ms per step on the GRCh38-sized input). */
int32_t asgart_search_duplications(asgart_index *idx, const uint64_t *chunks, int64_t n_chunks,
                                   const asgart_settings *settings, volatile uint64_t *progress,
                                   asgart_families **out);

/* Same, restricted to shard `shard` of `n_shards` (multi-GPU: one process per GPU, index replicated, no
 * exchange between shards): shard r owns the automaton segments that START in the r-th of n_shards equal slices of
 * the probe sequence; it searches its slice plus a look-back and a look-ahead halo (retrying with larger halos when a
 * decision is ambiguous).  The union of the shards' families, merged by asgart_families_keys -- or concatenated in shard
 * order -- is exactly the unsharded result (asgart_amd/multi.py merges by key). */
int32_t asgart_search_duplications_shard(asgart_index *idx, const uint64_t *chunks,
                                         int64_t n_chunks, const asgart_settings *settings,
                                         int32_t shard, int32_t n_shards,
                                         asgart_families **out);

/* Multi-GPU in ONE process: indices[r] is a replica of the same index on device r
 * (asgart_index_create on every device, or asgart_index_clone from the device that built it).  One host
 * thread per device runs shard r of n_devices; the per-shard families -- tens of MB -- come back over
 * each device's own host link and are concatenated in shard order: exactly the result of
 * asgart_search_duplications on one device.  No device-to-device exchange is needed inside one process
 * (the host is common); one-process-per-GPU hosts gather the shards' lists with RCCL instead, see
 * INTEGRATION.md section 4 and asgart_amd/multi.py.  The same index may appear more than once (each call
 * takes one of its internal contexts), which is how the sharding logic is tested on a single GPU. */
int32_t asgart_search_duplications_multi(asgart_index *const *indices, int32_t n_devices,
                                         const uint64_t *chunks, int64_t n_chunks,
                                         const asgart_settings *settings, volatile uint64_t *progress,
                                         asgart_families **out);

/* The general form of the two calls above: shard `shard` of `n_shards` (0 of 1: everything) with the
 * progress array of asgart_search_duplications (entries of ALL chunks are written). */
int32_t asgart_search_duplications_ex(asgart_index *idx, const uint64_t *chunks, int64_t n_chunks,
                                      const asgart_settings *settings, int32_t shard, int32_t n_shards,
                                      volatile uint64_t *progress, asgart_families **out);

/* Several passes over the same chunks in ONE call -- what the `asgart` binary does when it is run with
 * and without -R / -C over one strand (reference src/bin/asgart.rs:677-693 builds one RunSettings per
 * invocation; the direct and the -RC run of BASELINE.json's "direct+RC" are two of them): settings[j] are
 * the RunSettings of pass j, out[j] receives its families (exactly what asgart_search_duplications returns
 * for settings[j]).  Passes that differ in orientation only (same probe_size, max_gap_size,
 * min_duplication_length, max_cardinality; up to four) run as ONE job: the probe sequence is pass 0's
 * chunks followed by pass 1's ... (chunk order inside each pass as in src/bin/asgart.rs:201-253), searched, scanned
 * and placed in one sweep at full chip rate, and every extension tier is ONE launch over the merged, cost-sorted
 * segment list, so that every pass's longest serial segments start at once on compute units of their own
 * (asgart_stats.passes tells).  An unsharded call that finds ONE segment to be its whole extension (its longest single
 * segment above option fuse_pole_pct = 88 % of the extension: the other pass's front might hide beside that segment)
 * makes the index TIME the calls that follow both ways in turn -- one job, pipelined single-pass calls, two each -- and
 * keep the faster way for these settings; sharded calls always run as one job; option fuse_passes = 2 / 0 forces
 * either.  Otherwise (different
 * settings) the library pipelines the passes as single calls: pass j+1 is issued the moment the chip-wide,
 * HBM-bound phases of pass j are over (probe search, scans, hit materialisation -- the moment the `progress`
 * array of a single call jumps), so its search runs beside pass j's extension automaton, whose tail is a few
 * serial segments on one compute unit each; the passes are issued longest extension first (from the
 * durations the index remembers per orientation; reversed orientations first while nothing is known).
 * At most two passes are in flight (the index's two call contexts).  Results do not depend on the order.
 * On error every out[j] is NULL. */
int32_t asgart_search_duplications_passes(asgart_index *idx, const uint64_t *chunks, int64_t n_chunks,
                                          const asgart_settings *settings, int32_t n_passes,
                                          asgart_families **out);
/* ... restricted to shard `shard` of `n_shards`: the shard's slice of EVERY pass (see asgart_search_duplications_shard),
 * all of them as the same ONE job -- rank r of an n-GPU run executes the algorithm a single GPU does, on 1/n of the
 * probes of each pass.  Every pass's families carry keys counted from the start of their own pass. */
int32_t asgart_search_duplications_passes_shard(asgart_index *idx, const uint64_t *chunks, int64_t n_chunks,
                                                const asgart_settings *settings, int32_t n_passes,
                                                int32_t shard, int32_t n_shards, asgart_families **out);

void asgart_families_counts(const asgart_families *f, uint64_t *n_families, uint64_t *n_sds);
/* fam_offsets: n_families+1 entries; sds: n_sds entries */
void asgart_families_copy(const asgart_families *f, uint64_t *fam_offsets, asgart_proto_sd *sds);
/* keys: n_families entries, (first probe of the family's automaton segment << 32) | family ordinal inside it.
 * Ascending keys == the reference's order (chunk order, discovery order inside a chunk, src/bin/asgart.rs:241-253).
 * The shards of a sharded call own contiguous slices of every pass: a gatherer merges their families by key
 * (asgart_search_duplications_multi does; one-process-per-GPU hosts do it after the RCCL gather, asgart_amd/multi.py)
 * or concatenates them in shard order. */
void asgart_families_keys(const asgart_families *f, uint64_t *keys);
void asgart_families_free(asgart_families *f);

/* ---- ComputeScore (`--compute-score`) --------------------------------------
 * Replaces the ComputeScore step, reference src/bin/asgart.rs:98-112 with
 * ProtoSD::levenshtein, src/structs.rs:439-452: for each of the n_sd duplications
 * identity = 100 * (1 - levenshtein(left arm, right arm) / max(left_length, right_length)),
 * arms taken over the INCLUSIVE ranges [p ..= p + length] of the text, the right arm
 * reversed / complemented first when the flags say so; computed in f64, stored as f32
 * like `sd.identity`.  Exact unit-cost edit distance (anti-diagonal DP on the GPU).
 * Errors: a range that reaches past the text (the reference panics), two empty arms. */
int32_t asgart_compute_scores(asgart_index *idx, const asgart_proto_sd *sds, int64_t n_sd,
                              int32_t reversed, int32_t complemented, float *identity);

/* ---- the steps behind the search step ------------------------------------------
 * Replaces FilterNs, ReOrder, ReduceOverlap and Sort of the reference's step chain (src/bin/asgart.rs:33-96 with
 * ProtoSD::n_content src/structs.rs:454-467, reduce_overlap :481-562; order :738-747) for the families of one run,
 * given as the arrays asgart_families_copy fills (fam_offsets: n_families + 1 entries).  ComputeScore, the optional
 * step between ReduceOverlap and Sort, is asgart_compute_scores (call it on the result: Sort does not look at it).
 * The N content of every arm is counted on the GPU over the resident text (inclusive ranges [p ..= p + length], the
 * f32 quotient and the 0.2 threshold as the reference computes them); the reduction runs on `threads` host threads
 * (0: all), one family at a time each.  out: the surviving families (asgart_families_counts / _copy / _free;
 * asgart_families_keys gives the input ordinal of every surviving family).
 * Errors: a duplication whose inclusive range reaches past the text (the reference panics there). */
int32_t asgart_post_process(asgart_index *idx, const uint64_t *fam_offsets, uint64_t n_families,
                            const asgart_proto_sd *sds, int32_t threads, asgart_families **out);

/* ---- finer-grained entry points mirroring the reference's inner API;
 *      used by the parity tests ------------------------------------------ */

/* Searcher cache entries (reference src/searcher.rs:99-143): for each of the
 * n_pat 8-byte patterns the SA slot interval [lo,hi) of suffixes starting
 * with it.  Patterns must be over {A,T,G,C,N}. */
int32_t asgart_searcher_cache_get(asgart_index *idx, const uint8_t *patterns8, int64_t n_pat,
                                  uint64_t *lo, uint64_t *hi);
/* Searcher::search (reference src/searcher.rs:145-180) for n_pat patterns of
 * k bytes each: SA slot interval [lo,hi) whose entries are the hits, in SA
 * order.  Read the hit positions with asgart_sa_read. */
int32_t asgart_searcher_search(asgart_index *idx, const uint8_t *patterns, int64_t n_pat,
                               uint64_t k, uint64_t *lo, uint64_t *hi);
int32_t asgart_sa_read(asgart_index *idx, uint64_t lo, uint64_t hi, int64_t *out);

/* Per-probe filtered hit lists exactly as the automaton consumes them
 * (reference src/automaton.rs:96-117), for all chunks concatenated: probe j of
 * a chunk sits at needle offset (j+1)*(k/2).  status: 0 processed, 1 skipped
 * ('N' first base), 2 skipped (cardinality).  Two-call: with status==NULL
 * returns the probe count and *n_hits; then fills status[n_probes],
 * row_offsets[n_probes+1], hits[n_hits] (hit starts, SA order). <0 on error. */
int64_t asgart_probe_hits(asgart_index *idx, const uint64_t *chunks, int64_t n_chunks,
                          const asgart_settings *settings, uint8_t *status,
                          uint64_t *row_offsets, uint64_t *hits, uint64_t *n_hits);

#define ASGART_STATS_YARDSTICK 1u
#define ASGART_STATS_RAW_HITS 2u
/* asgart_search_duplications is re-entrant: up to two calls on one index may be in flight from
 * different host threads (e.g. the direct and the -RC pass), each in its own internal context.
 * ASGART_STATS_CTX(i), i = 0 or 1, selects the stats of context i instead of the last call's. */
#define ASGART_STATS_CTX(i) (((uint32_t)(i) + 1u) << 8)
/* Stats of the last search call on this index.  With ASGART_STATS_YARDSTICK in
 * `flags` extra (untimed) kernels also fill bisect_steps, the accounting fields and raw_hits; with
 * ASGART_STATS_RAW_HITS raw_hits alone (the probes the position filter answered are looked up for it). */
int32_t asgart_get_stats(asgart_index *idx, uint32_t flags, asgart_stats *out);

/* Thread-local message of the last error returned on this thread. */
const char *asgart_last_error(void);
/* "asgart-hip <version> gfx950" */
const char *asgart_version(void);

#ifdef __cplusplus
}
#endif
#endif
