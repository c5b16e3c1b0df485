/*
 * doppel_oracle.c -- CPU restatement (plain C) of the doppel-speller hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and bench.py's
 * `cpu_baseline` leg may load it; the shipped path is the HIP library in doppel-speller_amd/csrc and it never
 * falls back to this file.
 *
 * Every function restates one numba-jitted function of the reference, keeping the reference's algorithmic
 * structure (dense N-vector scatter-add per query, dense float64 finalise, sequential top-k scan; a freshly
 * zeroed full DP matrix per Levenshtein call, ~77 calls per pair) so that it doubles as the timed CPU baseline:
 *
 *   ds_oracle_fast_jaccard          <- doppelspeller/match_maker.py:16-50
 *   ds_oracle_fast_arg_top_k        <- doppelspeller/match_maker.py:53-71
 *   ds_oracle_jaccard_topk          <- doppelspeller/match_maker.py:183-203 (get_closest_matches, batched, row ids)
 *   ds_oracle_levenshtein_ratio     <- doppelspeller/feature_engineering.py:25-63
 *   ds_oracle_construct_features    <- doppelspeller/feature_engineering.py:75-169
 *
 * Pinning: checked bit-for-bit against the .npz/.json fixtures under tests/golden, which were captured by running the reference's
 * own function bodies (tests/golden/make_golden.py).  The reference is jitted by numba 0.45, which is not
 * installable here; where numba's typing differs from executing the same source under NumPy, `typing` selects:
 *   typing = 0  "numba"  -- the specification the product follows:
 *                           match_maker.py:70   threshold subtraction in float64
 *                           feature_engineering.py:51-61  min() in int64, uint8 truncation on store
 *                           feature_engineering.py:158    ranks computed in float64, rounded to float32 on store
 *   typing = 1  "numpy"  -- what the captured vectors were produced with (float32 subtraction, uint8 wrap before
 *                           min(), float32 ranks).  Used only to pin this file against the goldens on ALL vectors;
 *                           the two modes agree on every vector flagged margin_ok.
 * fastmath=True (H4) is followed in source order (strict IEEE), as the captured vectors are.
 *
 * Build: see oracle/Makefile (gcc -O2 -fopenmp -ffp-contract=off).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define DS_WORDS 15          /* settings.py:65  NUMBER_OF_WORDS_FEATURES */
#define DS_FEATURES 66       /* feature_engineering.py:67 */
#define DS_MAX_CHARS 255     /* settings.py:68 */

void ds_oracle_set_num_threads(int threads)
{
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#else
    (void)threads;
#endif
}

int ds_oracle_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* match_maker.py:45-50.  `scores` is the float32 scratch vector of line 45, `out` the float64 result of line 50
 * (numba types max_intersection_possible as float64, hence a float64 array). */
void ds_oracle_fast_jaccard(int64_t number_of_truth_titles, double max_intersection_possible,
                            const int32_t *non_zero_columns_for_the_row, int64_t n_columns,
                            const int64_t *rowptr, const int32_t *truth_idx, const float *idf32,
                            const float *sums_matrix_truth, float *scores, double *out)
{
    memset(scores, 0, sizeof(float) * (size_t)number_of_truth_titles);
    for (int64_t c = 0; c < n_columns; ++c) {                         /* :46 sequential over the query's columns */
        const int32_t column = non_zero_columns_for_the_row[c];
        const float value = idf32[column];                            /* :130 constant per posting list */
        for (int64_t p = rowptr[column]; p < rowptr[column + 1]; ++p) /* :48 scores[columns] += values */
            scores[truth_idx[p]] += value;
    }
    for (int64_t t = 0; t < number_of_truth_titles; ++t) {            /* :50 */
        const double s = (double)scores[t];
        out[t] = s / ((double)sums_matrix_truth[t] + (max_intersection_possible - s));
    }
}

/* match_maker.py:53-71.  Returns the number of indexes written (== k unless fewer than k values qualify). */
int64_t ds_oracle_fast_arg_top_k(const double *array, int64_t n, int32_t k, int32_t typing, int64_t *out)
{
    float *sorted_indexes = (float *)calloc((size_t)k, sizeof(float)); /* :60 */
    int32_t minimum_index = 0;
    double minimum_index_value = 0.0;                                  /* :62 int64 0 unified with float32 -> f64 */
    for (int64_t i = 0; i < n; ++i) {                                  /* :63 */
        const double value = array[i];
        if (value > minimum_index_value) {                             /* :64 */
            sorted_indexes[minimum_index] = (float)value;              /* :65 float32 store */
            minimum_index = 0;                                         /* :66 argmin, first occurrence */
            for (int32_t j = 1; j < k; ++j)
                if (sorted_indexes[j] < sorted_indexes[minimum_index]) minimum_index = j;
            minimum_index_value = (double)sorted_indexes[minimum_index]; /* :67 */
        }
    }
    free(sorted_indexes);
    const float buffer = 1e-6f;                                        /* settings.py:72 finfo(float32).resolution */
    if (typing == 0)
        minimum_index_value = minimum_index_value - (double)buffer;    /* :70 numba: float64 - float32 */
    else
        minimum_index_value = (double)(float)((float)minimum_index_value - buffer); /* NumPy: float32 arithmetic */
    int64_t found = 0;
    for (int64_t i = n - 1; i >= 0 && found < k; --i)                  /* :71 nonzero()[0][::-1][:k] */
        if (array[i] >= minimum_index_value) out[found++] = i;
    return found;
}

/* match_maker.py:192-203 for a batch of queries, returning truth ROW indexes (the .loc -> title_id map of :190 is
 * host-side).  q_cols holds each query's columns in accumulation order.  Returns 0, or -(q+1) if query q produced
 * fewer than k rows (the reference raises 'top_matches.shape[0] != self.top_n', :188-189). */
int64_t ds_oracle_jaccard_topk(const int64_t *rowptr, const int32_t *truth_idx, const float *idf32,
                               const float *sums32, int64_t n_truth, const int64_t *q_rowptr, const int32_t *q_cols,
                               const double *q_maxint, int64_t n_queries, int32_t k, int32_t typing,
                               int32_t *out_rows)
{
    int64_t status = 0;
#pragma omp parallel
    {
        float *scores = (float *)malloc(sizeof(float) * (size_t)n_truth);
        double *jaccard = (double *)malloc(sizeof(double) * (size_t)n_truth);
        int64_t *top = (int64_t *)malloc(sizeof(int64_t) * (size_t)k);
#pragma omp for schedule(dynamic, 4)
        for (int64_t q = 0; q < n_queries; ++q) {
            ds_oracle_fast_jaccard(n_truth, q_maxint[q], q_cols + q_rowptr[q], q_rowptr[q + 1] - q_rowptr[q],
                                   rowptr, truth_idx, idf32, sums32, scores, jaccard);
            const int64_t found = ds_oracle_fast_arg_top_k(jaccard, n_truth, k, typing, top);
            for (int32_t j = 0; j < k; ++j) out_rows[q * k + j] = j < found ? (int32_t)top[j] : -1;
            if (found != k) {
#pragma omp critical
                if (status == 0 || -(q + 1) > status) status = -(q + 1);
            }
        }
        free(scores);
        free(jaccard);
        free(top);
    }
    return status;
}

/* feature_engineering.py:25-63.  `matrix` is scratch of at least (la+1)*(lb+1) bytes (the np.zeros of :42). */
static uint8_t levenshtein_ratio(const uint8_t *sequence, int64_t length_x, const uint8_t *compare,
                                 int64_t length_y, int32_t typing, uint8_t *matrix)
{
    const int64_t total_length = length_x + length_y;                  /* :33 */
    if (length_x > length_y) {                                         /* :35-37 */
        const int64_t tl = length_x; length_x = length_y; length_y = tl;
        const uint8_t *ts = sequence; sequence = compare; compare = ts;
    }
    const int64_t size_x = length_x + 1, size_y = length_y + 1;
    memset(matrix, 0, (size_t)(size_x * size_y));                      /* :42 */
    for (int64_t x = 0; x < size_x; ++x) matrix[x * size_y] = (uint8_t)x; /* :43-44 (uint8 store wraps) */
    for (int64_t y = 0; y < size_y; ++y) matrix[y] = (uint8_t)y;          /* :45-46 */
    for (int64_t x = 1; x < size_x; ++x) {
        for (int64_t y = 1; y < size_y; ++y) {
            const int64_t up = matrix[(x - 1) * size_y + y];
            const int64_t diagonal = matrix[(x - 1) * size_y + y - 1];
            const int64_t left = matrix[x * size_y + y - 1];
            const int64_t substitution = sequence[x - 1] == compare[y - 1] ? 0 : 2; /* :50 / :59 */
            int64_t a = up + 1, b = diagonal + substitution, c = left + 1;
            if (typing != 0) { a &= 0xff; b &= 0xff; c &= 0xff; }      /* NumPy-2: uint8 + int stays uint8 */
            int64_t m = a < b ? a : b;
            if (c < m) m = c;
            matrix[x * size_y + y] = (uint8_t)m;                       /* uint8 store */
        }
    }
    if (total_length == 0) return 0;  /* 0/0 -> NaN -> uint8 is undefined in the reference; never reached by it */
    const int64_t distance = matrix[length_x * size_y + length_y];
    const double ratio = ((double)(total_length - distance) / (double)total_length) * 100.0; /* :63 source order */
    return (uint8_t)ratio;                                             /* return type numba.uint8 (:25) */
}

uint8_t ds_oracle_levenshtein_ratio(const uint8_t *a, int32_t la, const uint8_t *b, int32_t lb, int32_t typing)
{
    uint8_t *matrix = (uint8_t *)malloc((size_t)(la + 1) * (size_t)(lb + 1));
    const uint8_t r = levenshtein_ratio(a, la, b, lb, typing, matrix);
    free(matrix);
    return r;
}

/* feature_engineering.py:77-169 for ONE pair.  `matrix` scratch >= 272*257 bytes, `response` = float32[66]. */
static void construct_features_one(uint8_t title_number_of_characters, uint8_t truth_number_of_characters,
                                   const uint8_t *title, const uint8_t *title_truth,
                                   const uint32_t *truth_words_counts, uint8_t space_code,
                                   uint32_t number_of_truth_titles, int32_t typing, uint8_t *matrix, float *response)
{
    const int64_t lq = title_number_of_characters, lt = truth_number_of_characters; /* :101-102 */
    int64_t title_number_of_words = 1, truth_number_of_words = 1;
    for (int64_t i = 0; i < lq; ++i) title_number_of_words += title[i] == space_code;       /* :104 */
    for (int64_t i = 0; i < lt; ++i) truth_number_of_words += title_truth[i] == space_code; /* :105 */
    const uint8_t lev_ratio = levenshtein_ratio(title, lq, title_truth, lt, typing, matrix); /* :106 */

    uint8_t title_wo_spaces[DS_MAX_CHARS];                                                   /* :108 */
    int64_t lw = 0;
    for (int64_t i = 0; i < lq; ++i)
        if (title[i] != space_code) title_wo_spaces[lw++] = title[i];

    int64_t space_indexes[DS_WORDS];                                    /* :110-114 truth + [space], first 15 */
    int64_t n_space_indexes = 0;
    for (int64_t i = 0; i <= lt && n_space_indexes < DS_WORDS; ++i)
        if (i == lt || title_truth[i] == space_code) space_indexes[n_space_indexes++] = i;

    uint8_t reconstructed[1 + DS_WORDS * (DS_MAX_CHARS + 1)];           /* :115 */
    int64_t lr = 0;
    reconstructed[lr++] = space_code;
    float best_ratios[DS_WORDS], word_lengths[DS_WORDS], idf_s[DS_WORDS];
    for (int i = 0; i < DS_WORDS; ++i) best_ratios[i] = word_lengths[i] = idf_s[i] = NAN; /* :121-123 */

    int64_t last_index = 0;
    for (int64_t word_index = 0; word_index < n_space_indexes; ++word_index) {              /* :128 */
        const int64_t space_index = space_indexes[word_index];
        const uint8_t *truth_word = title_truth + last_index;                                /* :130-133 */
        const int64_t length_truth_word = space_index - last_index;
        last_index = space_index + 1;                                                        /* :135 */

        int64_t best_ratio = 0;                                                              /* :139 */
        const uint8_t *best_match = &space_code;                                             /* :140 */
        int64_t best_length = 1;
        for (int64_t possible_index = 0; possible_index < lw; ++possible_index) {            /* :141 */
            int64_t possible_length = lw - possible_index;                                   /* :142 */
            if (possible_length > length_truth_word) possible_length = length_truth_word;
            if (possible_length == 0) break;                                                 /* :143-144 */
            const uint8_t r = levenshtein_ratio(title_wo_spaces + possible_index, possible_length, truth_word,
                                                length_truth_word, typing, matrix);          /* :146 */
            if (r > best_ratio) {                                                            /* :147 */
                best_ratio = r;
                best_match = title_wo_spaces + possible_index;
                best_length = possible_length;
            }
        }
        best_ratios[word_index] = (float)best_ratio;                                         /* :151 */
        word_lengths[word_index] = (float)length_truth_word;                                 /* :152 */
        idf_s[word_index] =
            (float)log((double)number_of_truth_titles / (double)truth_words_counts[word_index]); /* :153 */
        memcpy(reconstructed + lr, best_match, (size_t)best_length);                         /* :154-155 */
        lr += best_length;
        reconstructed[lr++] = space_code;
    }

    float maximum = NAN;                                                                     /* :158 np.nanmax */
    for (int i = 0; i < DS_WORDS; ++i)
        if (!isnan(idf_s[i]) && (isnan(maximum) || idf_s[i] > maximum)) maximum = idf_s[i];
    float ranks_idf_s[DS_WORDS];
    for (int i = 0; i < DS_WORDS; ++i) {
        const float difference = maximum - idf_s[i];                   /* float32 - float32 */
        if (typing == 0)                                               /* numba: float32 / int64 -> float64 */
            ranks_idf_s[i] = (float)(1.0 + (double)difference / (double)truth_number_of_words);
        else                                                           /* NumPy-2 weak scalars: all float32 */
            ranks_idf_s[i] = 1.0f + difference / (float)truth_number_of_words;
    }

    const uint8_t reconstructed_lev_ratio =
        levenshtein_ratio(reconstructed + 1, lr - 2, title_truth, lt, typing, matrix);       /* :161-162 */

    response[0] = (float)title_number_of_characters;                                         /* :164-169 */
    response[1] = (float)truth_number_of_characters;
    response[2] = (float)title_number_of_words;
    response[3] = (float)truth_number_of_words;
    response[4] = (float)lev_ratio;
    response[5] = (float)reconstructed_lev_ratio;
    memcpy(response + 6, best_ratios, sizeof(best_ratios));
    memcpy(response + 6 + DS_WORDS, word_lengths, sizeof(word_lengths));
    memcpy(response + 6 + 2 * DS_WORDS, idf_s, sizeof(idf_s));
    memcpy(response + 6 + 3 * DS_WORDS, ranks_idf_s, sizeof(ranks_idf_s));
}

/* The gufunc of feature_engineering.py:69-76 over n pairs (layout '(),(),(l),(l),(m),(),(),(n)->(n)'), rows of
 * title / title_truth are `stride` bytes apart (255 in the reference's callers). */
void ds_oracle_construct_features(const uint8_t *title_number_of_characters,
                                  const uint8_t *truth_number_of_characters, const uint8_t *title,
                                  const uint8_t *title_truth, const uint32_t *truth_words_counts, uint8_t space_code,
                                  uint32_t number_of_truth_titles, int64_t n, int64_t stride, int32_t typing,
                                  float *response)
{
#pragma omp parallel
    {
        uint8_t *matrix = (uint8_t *)malloc((size_t)(DS_MAX_CHARS + DS_WORDS + 3) * (DS_MAX_CHARS + 2));
#pragma omp for schedule(dynamic, 64)
        for (int64_t i = 0; i < n; ++i)
            construct_features_one(title_number_of_characters[i], truth_number_of_characters[i], title + i * stride,
                                   title_truth + i * stride, truth_words_counts + i * DS_WORDS, space_code,
                                   number_of_truth_titles, typing, matrix, response + i * DS_FEATURES);
        free(matrix);
    }
}

/* Work counters for the roofline accounting of SURVEY.md section 8d: DP cells visited by construct_features for
 * one pair (|q||t| + window loop + |r||t|), computed by the same control flow. */
int64_t ds_oracle_feature_cells(uint8_t lq8, uint8_t lt8, const uint8_t *title, const uint8_t *title_truth,
                                uint8_t space_code)
{
    const int64_t lq = lq8, lt = lt8;
    int64_t lw = 0;
    for (int64_t i = 0; i < lq; ++i) lw += title[i] != space_code;
    int64_t cells = lq * lt, last = 0, words = 0, reconstructed = 0;
    for (int64_t i = 0; i <= lt && words < DS_WORDS; ++i) {
        if (i != lt && title_truth[i] != space_code) continue;
        const int64_t lword = i - last;
        last = i + 1;
        ++words;
        for (int64_t p = 0; p < lw; ++p) {
            const int64_t window = lw - p < lword ? lw - p : lword;
            if (window == 0) break;
            cells += window * lword;
        }
        reconstructed += (lword < lw ? lword : (lw ? lw : 1)) + 1;  /* upper estimate of |best window| + space */
    }
    return cells + (reconstructed > 1 ? reconstructed - 1 : 0) * lt;
}

/* ------------------------------------------------------------------------------------------------------------------
 * Next row f-1: the fuzzy "very close match" step of Prediction._find_close_matches
 * (doppelspeller/predict.py:140-183) on top of common.levenshtein_ratio / levenshtein_token_sort_ratio
 * (doppelspeller/common.py:161-167).  The latter call python-Levenshtein 0.12.0 `ratio`, which is NOT under
 * /root/reference (requirements.txt:9) and not installed here: PARITY UNPINNED.  Restated from its published
 * algorithm: ratio(a, b) = (lensum - ldist) / lensum with ldist the edit distance with substitution cost 2
 * (= lensum - 2*LCS), 1.0 for two empty strings; common.py:162 then takes int(round(ratio * 100)) (round half to even).
 * Strings are code arrays; `sort_key[code]` gives the character order Python's sorted() uses (the code point).
 * ------------------------------------------------------------------------------------------------------------------ */
static int64_t lcs_length(const uint8_t *a, int64_t la, const uint8_t *b, int64_t lb)
{
    int32_t *row = (int32_t *)calloc((size_t)lb + 1, sizeof(int32_t));
    for (int64_t i = 1; i <= la; ++i) {
        int32_t diagonal = 0;
        for (int64_t j = 1; j <= lb; ++j) {
            const int32_t up = row[j];
            row[j] = a[i - 1] == b[j - 1] ? diagonal + 1 : (row[j - 1] > up ? row[j - 1] : up);
            diagonal = up;
        }
    }
    const int64_t result = row[lb];
    free(row);
    return result;
}

/* common.py:161-162 */
int32_t ds_oracle_levenshtein_ratio_rounded(const uint8_t *a, int32_t la, const uint8_t *b, int32_t lb)
{
    const int64_t lensum = (int64_t)la + lb;
    if (lensum == 0) return 100;
    const int64_t ldist = lensum - 2 * lcs_length(a, la, b, lb);
    const double ratio = (double)(lensum - ldist) / (double)lensum;
    return (int32_t)nearbyint(ratio * 100);
}

/* ' '.join(sorted(text.split())) of common.py:166 on a code string; returns the new length (<= n) */
static int64_t token_sort(const uint8_t *text, int64_t n, uint8_t space, const uint8_t *sort_key, uint8_t *out)
{
    int64_t starts[256], lengths[256], words = 0;
    for (int64_t i = 0; i < n;) {
        while (i < n && text[i] == space) ++i;
        if (i >= n) break;
        const int64_t start = i;
        while (i < n && text[i] != space) ++i;
        starts[words] = start;
        lengths[words++] = i - start;
    }
    for (int64_t i = 1; i < words; ++i) { /* stable insertion sort, lexicographic on sort_key */
        const int64_t s = starts[i], l = lengths[i];
        int64_t j = i - 1;
        for (; j >= 0; --j) {
            const int64_t m = lengths[j] < l ? lengths[j] : l;
            int cmp = 0;
            for (int64_t c = 0; c < m && cmp == 0; ++c)
                cmp = (int)sort_key[text[starts[j] + c]] - (int)sort_key[text[s + c]];
            if (cmp == 0) cmp = lengths[j] > l ? 1 : 0;
            if (cmp <= 0) break;
            starts[j + 1] = starts[j];
            lengths[j + 1] = lengths[j];
        }
        starts[j + 1] = s;
        lengths[j + 1] = l;
    }
    int64_t at = 0;
    for (int64_t w = 0; w < words; ++w) {
        if (w) out[at++] = space;
        memcpy(out + at, text + starts[w], (size_t)lengths[w]);
        at += lengths[w];
    }
    return at;
}

/* Prediction._get_levenshtein_ratio (predict.py:140-156) */
int32_t ds_oracle_close_ratio(const uint8_t *x, int32_t lx, const uint8_t *y, int32_t ly, uint8_t space,
                              const uint8_t *sort_key, int32_t threshold)
{
    const int64_t total = (int64_t)lx + ly;
    const int64_t delta = lx > ly ? lx - ly : ly - lx;
    if (((double)(total - delta) / (double)total) * 100 < (double)threshold) return 0;      /* :141-151 */
    const int32_t ratio = ds_oracle_levenshtein_ratio_rounded(x, lx, y, ly);                /* :153 */
    if (ratio > threshold) return ratio;
    uint8_t sorted_x[256], sorted_y[256];                                                   /* :154-155 */
    const int64_t sx = token_sort(x, lx, space, sort_key, sorted_x);
    const int64_t sy = token_sort(y, ly, space, sort_key, sorted_y);
    return ds_oracle_levenshtein_ratio_rounded(sorted_x, (int32_t)sx, sorted_y, (int32_t)sy);
}

/* ratios for n pairs given as padded rows (stride bytes apart) */
void ds_oracle_close_ratios(const uint8_t *x_len, const uint8_t *y_len, const uint8_t *x, const uint8_t *y,
                            int64_t n, int64_t stride, uint8_t space, const uint8_t *sort_key, int32_t threshold,
                            uint8_t *out)
{
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t i = 0; i < n; ++i)
        out[i] = (uint8_t)ds_oracle_close_ratio(x + i * stride, x_len[i], y + i * stride, y_len[i], space, sort_key,
                                                threshold);
}

/* ---- next row f-4: gradient-boosted tree ensemble on the 66 features (predict.py:229-234) -------------------------
 * xgboost (requirements.txt) is not part of the reference tree and not installable here: what follows restates the
 * PUBLISHED prediction rule of an xgboost `binary:logistic` booster on a dense float32 matrix with missing = NaN
 * (xgb.DMatrix(features), predict.py:229): per tree, from the root: a missing value follows the node's `missing`
 * child, otherwise value < split_condition follows `yes`, else `no`; the leaf values of the first n_trees trees
 * are summed in tree order in float32 starting from zero and the base margin is added to that sum (xgboost's
 * PredValue / PredLoopSpecalize); the prediction is 1 / (1 + exp(-margin)).  Parity unpinned
 * against the real library.  Node layout: feature < 0 marks a leaf whose value is in `threshold`. */
float ds_oracle_forest_margin(const int32_t *feature, const float *threshold, const int32_t *yes, const int32_t *no,
                              const int32_t *missing, const int64_t *tree_offsets, int32_t n_trees, float base_margin,
                              const float *row)
{
    float margin = 0.f;  /* cpu_predictor PredValue: psum = 0, += leaf per tree; the caller's buffer holds base_margin */
    for (int32_t t = 0; t < n_trees; ++t) {
        int64_t node = tree_offsets[t];
        while (feature[node] >= 0) {
            const float value = row[feature[node]];
            int32_t next;
            if (value != value) next = missing[node];
            else next = value < threshold[node] ? yes[node] : no[node];
            node = tree_offsets[t] + next;
        }
        margin = margin + threshold[node];
    }
    return base_margin + margin;
}

void ds_oracle_forest_predict(const int32_t *feature, const float *threshold, const int32_t *yes, const int32_t *no,
                              const int32_t *missing, const int64_t *tree_offsets, int32_t n_trees, float base_margin,
                              const float *rows, int64_t n, int64_t n_features, float *margins, float *probabilities)
{
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        const float margin = ds_oracle_forest_margin(feature, threshold, yes, no, missing, tree_offsets, n_trees,
                                                     base_margin, rows + i * n_features);
        if (margins) margins[i] = margin;
        if (probabilities) probabilities[i] = 1.0f / (1.0f + expf(-margin));
    }
}
