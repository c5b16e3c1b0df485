/*
 * doppel_amd.h -- C ABI of libdoppel_amd.so, the MI355X (gfx950) implementation of doppel-speller's
 * candidate-generation-and-scoring hot path.
 *
 * Every entry point replaces one piece of the reference's numba-jitted path; the reference interface it stands in
 * for is cited as `file:line` relative to the reference repository (mhaseebtariq/doppel-speller).  The reference is
 * Python, so the "FFI" a maintainer would add is a ctypes binding: see INTEGRATION.md.
 *
 * Conventions
 *   - plain pointers and sizes only; all buffers are caller-owned and contiguous;
 *   - functions return 0 on success and a negative DS_E_* code on failure; ds_last_error() returns a
 *     thread-local human-readable message for the last failure; no exception crosses the boundary;
 *   - "host" entry points take host pointers and copy H<->D themselves (synchronous); "_device" entry points take
 *     device pointers plus a hipStream_t (as void*) and only enqueue work on that stream;
 *   - a handle is bound to one device, owns its device memory, and is NOT thread-safe (one caller at a time,
 *     like the reference's single Python thread);
 *   - there is no CPU fallback: without a usable GPU every compute entry point fails with DS_E_HIP.
 */
#ifndef DOPPEL_AMD_H
#define DOPPEL_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DS_FEATURES_COUNT 66 /* feature_engineering.py:67  FEATURES_COUNT = 6 + 4 * NUMBER_OF_WORDS_FEATURES */
#define DS_WORDS 15          /* settings.py:65            NUMBER_OF_WORDS_FEATURES */
#define DS_MAX_CHARS 255     /* settings.py:68            MAX_CHARACTERS_ALLOWED_IN_THE_TITLE */

#define DS_OK 0
#define DS_E_ARG (-1)      /* invalid argument (null pointer, bad size, unsorted posting list ...) */
#define DS_E_HIP (-2)      /* HIP runtime failure (no device, out of memory, launch failure) */
#define DS_E_TOP_N (-3)    /* fewer than k rows qualify: the reference raises 'top_matches.shape[0] != self.top_n'
                              (match_maker.py:188-189) */
#define DS_E_INTERNAL (-4)

typedef struct ds_index ds_index;   /* truth inverted index resident in HBM (MatchMaker.__init__ product) */
typedef struct ds_titles ds_titles; /* table of encoded titles resident in HBM */
typedef struct ds_timer ds_timer;   /* pair of HIP events */
typedef struct ds_problem ds_problem; /* host-side product of the native index build (next row f-3) */
typedef struct ds_forest ds_forest;   /* tree ensemble resident in HBM (next row f-4) */

/* ---- library ---------------------------------------------------------------------------------------------------- */
const char *ds_last_error(void);
int ds_version(void);
/* Identity of the sources this binary was compiled from: the first 16 hex digits of the SHA-256 over the files of
 * csrc/ and include/ (name and contents, names ascending), passed by the build as -DDS_BUILD_ID.  The Python loader
 * refuses a library whose id differs from the sources next to it (a stale .so cannot pass for a current one). */
const char *ds_build_id(void);
int ds_device_count(int *count);
int ds_device_name(int device, char *name, size_t capacity);

/* ---- truth index:  MatchMaker.__init__ product (match_maker.py:99-107) ------------------------------------------ */
/* rowptr[V+1], truth_idx[nnz]: the V x N inverted index of match_maker.py:122-133 in CSR form (row g = n-gram
 * column g, entries = ascending truth row indexes).  idf32[V] = the constant per-posting value of :130.
 * sums32[N] = sums_matrix_truth of :102,174.  The per-posting value array of the reference is not stored.
 * The host part (tiling, signatures, duplicate ranks) runs on DS_HOST_THREADS threads (default: the CPUs of the
 * process, at most 32); DS_BUILD_LOG=1 prints its phase times on stderr. */
int ds_index_create(const int64_t *rowptr, const int32_t *truth_idx, const float *idf32, const float *sums32,
                    int64_t V, int64_t N, int device, ds_index **out);
void ds_index_destroy(ds_index *index);
/* Host-only part of ds_index_create, exported for tests: rank_out[t] = number of truth rows with the same column set
 * and the same sums32 bits as row t but a larger row index (saturating at 65535).  fast_arg_top_k returns the k
 * LARGEST ROW INDEXES at or above its threshold (match_maker.py:71): a row of rank >= k can never be returned. */
int ds_index_duplicate_ranks(const int64_t *rowptr, const int32_t *truth_idx, const float *sums32, int64_t V, int64_t N,
                             uint16_t *rank_out);
/* Host-only, for tests: FNV-1a digests of the arrays ds_index_create uploads, built for tiles of `tile_rows` rows.
 * digest[0..5] = list pointers, postings, per-posting info, row records, tile minima, signature columns; [6] = posting
 * quads; [7] = 1 when the index would be served by the literal kernel alone.  Must not depend on DS_HOST_THREADS. */
int ds_index_image_digest(const int64_t *rowptr, const int32_t *truth_idx, const float *idf32, const float *sums32,
                          int64_t V, int64_t N, int64_t tile_rows, uint64_t digest[8]);
/* Diagnostics switches of an index.  "count_bytes" (0 / 1): the next ds_jaccard_topk* calls run the instantiation of
 * the fast kernel that also counts the bytes it requests from global memory (reported by ds_jaccard_sync, stats[15]);
 * same work, same results, about 3 % slower -- bench.py runs it once outside the timed region for its roofline.
 * "query_order" (1 / 0, default 1): the fast kernel's work queue hands out the queries with most columns first (a device-side
 * counting sort; 0 = the caller's order; answers do not depend on it).
 * The product's launches read NOTHING from the environment.  Environment read by ds_index_create / the host builds only
 * (tests and A/B measurements): DS_HOST_THREADS, DS_BUILD_LOG=1, DS_SORT_ROWS=0, DS_GEOMETRY=narrow|wide.  The kernels'
 * tuning knobs (DS_SPARSE_QUADS, DS_SELECT_*, DS_DEBUG, DS_PHASE_TIMERS / DS_PHASE_DUMP) exist in -DDS_DIAGNOSTICS builds only. */
int ds_index_option(ds_index *index, const char *name, int64_t value);
/* info[0]=N info[1]=V info[2]=nnz info[3]=tile size info[4]=tiles info[5]=device bytes info[6]=padded postings
 * info[7]=bytes of the forward index (row starts uint32 while nnz < 2^32, columns uint16 while V <= 65536), part of info[5] */
int ds_index_info(const ds_index *index, int64_t info[8]);

/* ---- Jaccard top-k:  fast_jaccard + fast_arg_top_k (match_maker.py:16-71) behind get_closest_matches (:192-203) - */
/* For each query q: columns q_cols[q_rowptr[q] .. q_rowptr[q+1]) in ACCUMULATION ORDER (the order of
 * matrix_non_zero_columns[row], :118), q_maxint[q] = max_intersection_possible (:197, float64).
 * out_rows[q*k .. q*k+k) = truth ROW indexes in descending row-index order, exactly
 * `(array >= threshold).nonzero()[0][::-1][:k]` of :71 (the title_id mapping of :190 is the caller's).
 * Inputs the reference cannot produce are still answered as its loop would answer them (by the literal kernel): a
 * column listed twice in a query is added twice, more than 128 columns, a q_maxint below the columns' idf total. */
int ds_jaccard_topk(ds_index *index, const int64_t *q_rowptr, const int32_t *q_cols, const double *q_maxint,
                    int64_t Q, int32_t k, int32_t *out_rows);
/* Same with every array already in HBM; enqueues on `stream` and returns without synchronising.
 * q_nnz = q_rowptr[Q].  Errors detected on the device (DS_E_TOP_N ...) are reported by ds_jaccard_sync(). */
int ds_jaccard_topk_device(ds_index *index, const int64_t *d_q_rowptr, const int32_t *d_q_cols,
                           const double *d_q_maxint, int64_t Q, int32_t k, int32_t *d_out_rows, void *stream);
/* Waits for `stream`, returns the status of the last ds_jaccard_topk_device on this index;
 * stats[0]=queries answered by the exact dense kernel, stats[1]=queries in error, stats[2]=candidates evaluated
 * exactly (total), stats[3]=threshold selections run (total), stats[4..11]=shader-clock sums per kernel phase
 * (setup, list pointers, scatter, scan, select, exact stage, dense hand-over; only with DS_PHASE_TIMERS=1),
 * stats[12]=tiles handled sparsely, stats[13]=tiles scanned densely, stats[14]=skipped (non-essential) columns
 * summed over queries, stats[16..21]=queries handed to the dense kernel by reason (unsupported shape, work items,
 * candidate overflow in a sparse tile, in a dense tile, ties after pruning, fewer than k positive rows),
 * stats[22..25]=refine passes / raw entries / survivors / raw entries from sparse tiles (diagnostics),
 * stats[26]=duration of ds_jaccard_topk_kernel in microseconds (HIP events on the launch stream),
 * stats[27]=duration of ds_jaccard_dense_kernel in microseconds, stats[28..30]=first out-of-range index a
 * -DDS_BOUNDS_CHECK build caught (site, index, limit; all 0 otherwise), stats[31]=epochs of sparse tiles that were
 * processed again because the candidate buffer overflowed (the threshold is tightened first; results unaffected). */
int ds_jaccard_sync(ds_index *index, void *stream, int64_t stats[32]);
/* Per-query status of the last call (after synchronising `stream`): 0 = answered by the fast kernel, 1 = handed to and
 * answered by the literal kernel, 2 = fewer than k rows qualified (DS_E_TOP_N), 3 = bad column index (DS_E_ARG),
 * (4 is transient: a query whose near-ties did not fit the literal kernel's LDS buffer; ds_jaccard_sync answers it with a
 * full-row scan through a float64[N] scratch vector in HBM, allocated on first need, and the status becomes 1.)
 * stats[15] of ds_jaccard_sync = bytes requested by the fast kernel (only with ds_index_option "count_bytes"). */
int ds_jaccard_status(ds_index *index, void *stream, int32_t *status, int64_t Q);

/* ---- Levenshtein / features:  fast_levenshtein_ratio + construct_features (feature_engineering.py:25-169) ------- */
/* The 9-argument gufunc of feature_engineering.py:69-80 without the `dummy` argument: rows of q_enc / t_enc are
 * `stride` bytes apart (255 in predict.py:199-202), out = float32[n*66] written in place.  Host pointers: the pairs
 * travel in chunks of 16384 through pinned staging buffers (only the titles' own bytes, lengths and word counts are
 * shipped, not the padding), up to 8 host threads with a stream each overlap copy-in / kernel / copy-out; the staging
 * buffers (<= 110 MB pinned per device) are allocated by a device's first call and kept.  Calls for one device are
 * serialised by that device's mutex. */
int ds_construct_features(const uint8_t *q_len, const uint8_t *t_len, const uint8_t *q_enc, const uint8_t *t_enc,
                          const uint32_t *t_word_counts, uint8_t space_code, uint32_t n_truth, int64_t n,
                          int64_t stride, int device, float *out);

/* Encoded titles uploaded once (FeatureEngineering.encode_title rows, feature_engineering.py:298-307, and for a
 * truth table get_truth_words_counts rows, :309-319; word_counts may be NULL for a query table). */
int ds_titles_create(const uint8_t *enc, int64_t stride, const uint8_t *len, const uint32_t *word_counts, int64_t n,
                     int device, ds_titles **out);
void ds_titles_destroy(ds_titles *titles);
/* "truth_records" (default 1): the indexed entry points keep, per row of a TRUTH table, what construct_features derives from
 * the truth title alone (word boundaries, idf_s, ranks: feature_engineering.py:110-123,152-158) -- 160 bytes of HBM per row,
 * built on the first call that names (number_of_truth_titles, space_code).  0 frees them: everything per pair again. */
int ds_titles_option(ds_titles *titles, const char *name, int64_t value);
/* Pairs given as (query row, truth row) indexes into two tables: out[i] = construct_features(q[pair_q[i]],
 * t[pair_t[i]]).  Host pointers for pair_q / pair_t / out. */
int ds_construct_features_indexed(ds_titles *queries, ds_titles *truth, const int32_t *pair_q, const int32_t *pair_t,
                                  uint8_t space_code, uint32_t n_truth, int64_t n, float *out);
/* Device-resident variant for the fused pipeline: d_pair_t = the top-k rows written by ds_jaccard_topk_device,
 * pair i belongs to query row q_first + i / k when d_pair_q is NULL.  d_out = float32[n*66] in HBM. */
int ds_construct_features_indexed_device(ds_titles *queries, ds_titles *truth, const int32_t *d_pair_q,
                                         const int32_t *d_pair_t, int64_t q_first, int32_t k, uint8_t space_code,
                                         uint32_t n_truth, int64_t n, float *d_out, void *stream);
/* fast_levenshtein_ratio (feature_engineering.py:25-63) for n independent pairs of code strings:
 * a = a_chars[a_off[i] .. a_off[i+1]), b likewise; out[i] = ratio (uint8).  Host pointers.
 * method 0 = bit-parallel LCS kernel (exact uint8-wrap DP where lengths require it), 1 = anti-diagonal DP kernel. */
int ds_levenshtein_ratio_batch(const uint8_t *a_chars, const int64_t *a_off, const uint8_t *b_chars,
                               const int64_t *b_off, int64_t n, int method, int device, uint8_t *out);
/* fast_levenshtein_ratio(a, b) (feature_engineering.py:25-63) for ONE pair of code strings on device 0, as SURVEY.md 8b
 * lists it ("for tests"): returns the reference's uint8 result (0..255) or a negative DS_E_* code.  A one-pair wrapper
 * over ds_levenshtein_ratio_batch (method 0): a kernel launch and two copies per call -- a test entry, not a hot path. */
int ds_levenshtein_ratio(const uint8_t *a, int la, const uint8_t *b, int lb);

/* ---- next row: Prediction._find_close_matches (predict.py:140-183) -------------------------------- */
/* For query row q (q_first + q in the query table) and its k candidate truth rows pair_t[q*k .. q*k+k):
 * ratios[q*k+j] = Prediction._get_levenshtein_ratio(query, candidate) (predict.py:147-156: length pre-filter,
 * common.levenshtein_ratio = int(round(python-Levenshtein ratio * 100)), token-sort fallback common.py:165-167);
 * best_row[q] = the candidate with the highest ratio > threshold if exactly one reaches it (predict.py:172-176), else
 * -1.  sort_key[256] = order of the character codes under Python's sorted() (the code points).  python-Levenshtein is
 * not part of the reference tree: its published definition is restated (parity pinned against the tests' CPU
 * restatement only). */
int ds_close_matches(ds_titles *queries, ds_titles *truth, const int32_t *pair_t, int32_t k, int64_t n_queries,
                     uint8_t space_code, const uint8_t *sort_key, int32_t threshold, uint8_t *ratios,
                     int32_t *best_row);
int ds_close_matches_device(ds_titles *queries, ds_titles *truth, const int32_t *d_pair_t, int64_t q_first, int32_t k,
                            int64_t n_queries, uint8_t space_code, const uint8_t *d_sort_key, int32_t threshold,
                            uint8_t *d_ratios, int32_t *d_best_row, void *stream);

/* ---- next row f-2: the pair list between the fuzzy step and the model, on the device (predict.py:172-183,195-204) ---
 * d_best_row[q] >= 0 marks a query the fuzzy step matched (ds_close_matches_device).  The remaining queries keep
 * their order; each contributes its k candidate rows d_rows[q*k .. q*k+k) in order: d_pair_q / d_pair_t (room for
 * n_queries * k entries) receive the (query row, truth row) index pairs for ds_construct_features_indexed_device,
 * query rows offset by q_first.  d_counts = int64[ds_remaining_pairs_counts_size(n_queries)]: [0] = remaining
 * queries, [1] = pairs, the rest is scratch.  Asynchronous on `stream`. */
int64_t ds_remaining_pairs_counts_size(int64_t n_queries);
int ds_remaining_pairs_device(const int32_t *d_best_row, const int32_t *d_rows, int64_t n_queries, int32_t k,
                              int64_t q_first, int32_t *d_pair_q, int32_t *d_pair_t, int64_t *d_counts, void *stream);
/* predict.py:246-252 for n_remaining queries of k consecutive pairs each: d_match_query[r] = the query row of group
 * r, d_match_row[r] = the truth row of its maximum prediction when that maximum is above `threshold`
 * (PREDICTION_PROBABILITY_THRESHOLD, settings.py:76) and a single pair holds it, else -1. */
int ds_select_matches_device(const int32_t *d_pair_q, const int32_t *d_pair_t, const float *d_predictions,
                             int64_t n_remaining, int32_t k, float threshold, int32_t *d_match_query,
                             int32_t *d_match_row, void *stream);

/* ---- next row f-3: native index build ----------------------------------------------------------------------------
 * Replaces the Python / lil_matrix loops of MatchMaker.__init__ (doppelspeller/match_maker.py:84-181) and
 * get_n_grams / get_n_grams_counter (doppelspeller/common.py:145-151): from the transformed titles (byte strings,
 * concatenated, offsets[n + 1]) to the arrays ds_index_create and ds_jaccard_topk take.  Host code.  Column ids
 * ascend with the n-gram's byte string and a title's float32 idf sum runs in first-occurrence order of its n-grams
 * (the reference leaves both to Python's set iteration order).  The arrays belong to the handle.  Threaded over title
 * ranges (DS_HOST_THREADS, default: the CPUs of the process, at most 32); the result does not depend on the count. */
int ds_problem_create(const uint8_t *truth_chars, const int64_t *truth_offsets, int64_t n_truth,
                      const uint8_t *query_chars, const int64_t *query_offsets, int64_t n_queries, int32_t n_gram,
                      ds_problem **out);
void ds_problem_destroy(ds_problem *problem);
/* info: n_truth, n_queries, n_columns, nnz (truth), nnz (queries), n_gram */
int ds_problem_info(const ds_problem *problem, int64_t info[8]);
/* vocabulary[V]: the n-gram of every column as big-endian bytes in a uint32; any out pointer may be NULL */
int ds_problem_arrays(const ds_problem *problem, const uint32_t **vocabulary, const float **idf32, const double **idf64,
                      const int64_t **rowptr, const int32_t **truth_idx, const float **sums32, const int64_t **q_rowptr,
                      const int32_t **q_cols, const double **q_maxint);

/* The encoders of the features' inputs for whole collections, host code, threaded:
 * ds_encode_titles  = FeatureEngineering.encode_title (doppelspeller/feature_engineering.py:298-307) per title:
 *   out_enc[t*stride ..) = the title's characters mapped through code_of[256] (NULL: bytes as they are), 0-padded to
 *   `stride` (255 in the reference, settings.py:68); out_len[t] = its number of characters (predict.py:195-197);
 * ds_truth_word_counts = FeatureEngineering.get_truth_words_counts (:309-319) per title over the counter of
 *   common.py:140-142: out_counts[t*DS_WORDS + j] = the number of truth titles holding the j-th word of title t (a word
 *   repeated inside one title counts once), 0-padded.  separator[256] != 0 marks the bytes str.split() splits on. */
int ds_encode_titles(const uint8_t *chars, const int64_t *offsets, int64_t n, const uint8_t *code_of, int64_t stride,
                     uint8_t *out_enc, uint8_t *out_len);
int ds_truth_word_counts(const uint8_t *chars, const int64_t *offsets, int64_t n, const uint8_t *separator,
                         uint32_t *out_counts);

/* transform_title (doppelspeller/common.py:20-47) for n titles whose Unicode step (NFD + ASCII encoding, Python's
 * unicodedata) is already done: lower case, '-' -> ' ', keep [a-zA-Z0-9\s], ' +' -> ' ', strip, cut to max_characters
 * (settings.py:68 = 255), strip, '0'-pad titles shorter than n_gram.  out_chars needs offsets[n] + n * n_gram bytes. */
int ds_transform_titles(const uint8_t *chars, const int64_t *offsets, int64_t n, int32_t max_characters, int32_t n_gram,
                        uint8_t *out_chars, int64_t *out_offsets);

/* ---- next row f-4: tree-ensemble scoring of the feature matrix -----------------------------------------------------
 * Replaces xgb.DMatrix(features) + model.predict(features_d, ntree_limit=...) (doppelspeller/predict.py:229-234) for a
 * `binary:logistic` booster, on the float32[n, n_features] matrix that ds_construct_features_* left in HBM.  Nodes of
 * all trees are concatenated; tree t owns nodes [tree_offsets[t], tree_offsets[t + 1]); child ids are tree-relative
 * (xgboost's nodeid); feature[i] < 0 marks a leaf whose value is threshold[i]; a NaN feature follows `missing`,
 * value < threshold follows `yes`, else `no`.  base_margin = logit(base_score).  Pass only the first
 * best_ntree_limit trees to reproduce `ntree_limit`.  Either output pointer may be NULL.
 * xgboost is outside the reference tree: the published rule is restated, parity unpinned. */
int ds_forest_create(const int32_t *feature, const float *threshold, const int32_t *yes, const int32_t *no,
                     const int32_t *missing, const int64_t *tree_offsets, int32_t n_trees, int32_t n_features,
                     float base_margin, int device, ds_forest **out);
void ds_forest_destroy(ds_forest *forest);
int ds_forest_predict(ds_forest *forest, const float *rows, int64_t n, float *margins, float *probabilities);
int ds_forest_predict_device(ds_forest *forest, const float *d_rows, int64_t n, float *d_margins,
                             float *d_probabilities, void *stream);

/* ---- device memory / stream / timing plumbing (so tests and bench.py can keep inputs resident in HBM) ----------- */
int ds_malloc(void **ptr, size_t bytes, int device);
int ds_free(void *ptr, int device);
int ds_memcpy_h2d(void *dst, const void *src, size_t bytes, int device);
int ds_memcpy_d2h(void *dst, const void *src, size_t bytes, int device);
int ds_memset(void *dst, int value, size_t bytes, int device);
int ds_memcpy_d2d_async(void *dst, const void *src, size_t bytes, int device, void *stream);
int ds_stream_create(int device, void **stream);   /* a HIP stream for the _device entry points and RCCL */
int ds_stream_destroy(void *stream, int device);
int ds_stream_sync(void *stream, int device);
int ds_timer_create(int device, ds_timer **out);
void ds_timer_destroy(ds_timer *timer);
int ds_timer_start(ds_timer *timer, void *stream);
int ds_timer_stop(ds_timer *timer, void *stream);
int ds_timer_elapsed_ms(ds_timer *timer, float *ms); /* synchronises on the stop event */

#ifdef __cplusplus
}
#endif
#endif /* DOPPEL_AMD_H */
