/*
 * doppel_cpu_abi.c -- "libdoppel_cpu": the CPU oracle behind the SAME C ABI as the product (include/doppel_amd.h).
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE (same rule as doppel_oracle.c: only tests/ may load it).  SURVEY.md 8b asks for
 * "identical signatures in libdoppel_cpu (OpenMP) for parity": this file includes the product's header, so the entry points
 * below are compiled against the very declarations the HIP library implements -- a signature that drifts is a compile error --
 * and the ctypes stub a maintainer of the reference would write (examples/reference_binding.py) runs
 * unchanged against either library (DOPPEL_AMD_LIBRARY selects it): tests/test_cpu_abi_parity.py.
 *
 * Entry points (the core path; everything else of the header belongs to the product only):
 *   ds_index_create / ds_index_destroy / ds_index_info   the CSR of match_maker.py:122-133 kept as host copies
 *   ds_jaccard_topk                                       ds_oracle_jaccard_topk (match_maker.py:16-71, 183-203), numba typing
 *   ds_construct_features                                 ds_oracle_construct_features (feature_engineering.py:75-169)
 *   ds_levenshtein_ratio                                  ds_oracle_levenshtein_ratio (feature_engineering.py:25-63)
 *   ds_last_error, ds_version, ds_build_id, ds_device_count
 */
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/doppel_amd.h"

int64_t ds_oracle_jaccard_topk(const int64_t *rowptr, const int32_t *truth_idx, const float *idf32, const float *sums32,
                               int64_t n_truth, const int64_t *q_rowptr, const int32_t *q_cols, const double *q_maxint,
                               int64_t n_queries, int32_t k, int32_t typing, int32_t *out_rows);
void ds_oracle_construct_features(const uint8_t *title_number_of_characters, const uint8_t *truth_number_of_characters,
                                  const uint8_t *title, const uint8_t *title_truth, const uint32_t *truth_words_counts,
                                  uint8_t space_code, uint32_t number_of_truth_titles, int64_t n, int64_t stride,
                                  int32_t typing, float *response);
uint8_t ds_oracle_levenshtein_ratio(const uint8_t *a, int32_t la, const uint8_t *b, int32_t lb, int32_t typing);

struct ds_index {
    int64_t V, N, nnz;
    int64_t *rowptr;
    int32_t *truth_idx;
    float *idf32, *sums32;
};

static _Thread_local char last_error[512] = "";

static int fail(int code, const char *format, ...)
{
    va_list arguments;
    va_start(arguments, format);
    vsnprintf(last_error, sizeof(last_error), format, arguments);
    va_end(arguments);
    return code;
}

const char *ds_last_error(void) { return last_error; }
int ds_version(void) { return 200; }
const char *ds_build_id(void) { return "cpu-oracle"; }
int ds_device_count(int *count)
{
    if (count) *count = 0;   /* no GPU behind this library */
    return DS_OK;
}

static void *copy_of(const void *source, size_t bytes)
{
    void *copy = malloc(bytes ? bytes : 1);
    if (copy && bytes) memcpy(copy, source, bytes);
    return copy;
}

int ds_index_create(const int64_t *rowptr, const int32_t *truth_idx, const float *idf32, const float *sums32, int64_t V,
                    int64_t N, int device, ds_index **out)
{
    (void)device;
    if (!out) return fail(DS_E_ARG, "ds_index_create: out is null");
    *out = NULL;
    if (!rowptr || !idf32 || !sums32) return fail(DS_E_ARG, "ds_index_create: null input");
    if (V <= 0 || N <= 0) return fail(DS_E_ARG, "ds_index_create: V and N must be positive (V=%lld N=%lld)", (long long)V, (long long)N);
    if (rowptr[0] != 0) return fail(DS_E_ARG, "ds_index_create: rowptr[0] must be 0");
    for (int64_t g = 0; g < V; ++g) {
        if (rowptr[g + 1] < rowptr[g]) return fail(DS_E_ARG, "ds_index_create: rowptr not monotone at column %lld", (long long)g);
        int64_t previous = -1;
        for (int64_t p = rowptr[g]; p < rowptr[g + 1]; ++p) {
            if (!(truth_idx[p] > previous && truth_idx[p] < N))
                return fail(DS_E_ARG, "ds_index_create: posting list of column %lld is not strictly ascending within [0, N)", (long long)g);
            previous = truth_idx[p];
        }
    }
    ds_index *index = (ds_index *)calloc(1, sizeof(ds_index));
    if (!index) return fail(DS_E_INTERNAL, "ds_index_create: out of memory");
    index->V = V;
    index->N = N;
    index->nnz = rowptr[V];
    index->rowptr = (int64_t *)copy_of(rowptr, sizeof(int64_t) * (size_t)(V + 1));
    index->truth_idx = (int32_t *)copy_of(truth_idx, sizeof(int32_t) * (size_t)index->nnz);
    index->idf32 = (float *)copy_of(idf32, sizeof(float) * (size_t)V);
    index->sums32 = (float *)copy_of(sums32, sizeof(float) * (size_t)N);
    *out = index;
    return DS_OK;
}

void ds_index_destroy(ds_index *index)
{
    if (!index) return;
    free(index->rowptr);
    free(index->truth_idx);
    free(index->idf32);
    free(index->sums32);
    free(index);
}

int ds_index_info(const ds_index *index, int64_t info[8])
{
    if (!index || !info) return fail(DS_E_ARG, "ds_index_info: null argument");
    memset(info, 0, sizeof(int64_t) * 8);
    info[0] = index->N;
    info[1] = index->V;
    info[2] = index->nnz;
    return DS_OK;
}

int ds_jaccard_topk(ds_index *index, const int64_t *q_rowptr, const int32_t *q_cols, const double *q_maxint, int64_t Q,
                    int32_t k, int32_t *out_rows)
{
    if (!index) return fail(DS_E_ARG, "ds_jaccard_topk: null index");
    if (Q < 0) return fail(DS_E_ARG, "ds_jaccard_topk: negative query count");
    if (Q == 0) return DS_OK;
    if (!q_rowptr || !q_maxint || !out_rows) return fail(DS_E_ARG, "ds_jaccard_topk: null pointer");
    if (k < 1) return fail(DS_E_ARG, "ds_jaccard_topk: k must be >= 1");
    if (k > index->N)   /* match_maker.py:188-189 */
        return fail(DS_E_TOP_N, "top_matches.shape[0] != self.top_n (k=%d > number of truth titles=%lld)", k, (long long)index->N);
    for (int64_t i = 0; i < q_rowptr[Q]; ++i)
        if (q_cols[i] < 0 || q_cols[i] >= index->V) return fail(DS_E_ARG, "ds_jaccard_topk: a query has a column index outside [0, V)");
    const int64_t status = ds_oracle_jaccard_topk(index->rowptr, index->truth_idx, index->idf32, index->sums32, index->N, q_rowptr,
                                                  q_cols, q_maxint, Q, k, 0, out_rows);
    if (status != 0) return fail(DS_E_TOP_N, "top_matches.shape[0] != self.top_n (query %lld)", (long long)(-status - 1));
    return DS_OK;
}

int ds_construct_features(const uint8_t *q_len, const uint8_t *t_len, const uint8_t *q_enc, const uint8_t *t_enc,
                          const uint32_t *t_word_counts, uint8_t space_code, uint32_t n_truth, int64_t n, int64_t stride,
                          int device, float *out)
{
    (void)device;
    if (n < 0) return fail(DS_E_ARG, "ds_construct_features: negative pair count");
    if (n == 0) return DS_OK;
    if (!q_len || !t_len || !q_enc || !t_enc || !t_word_counts || !out) return fail(DS_E_ARG, "ds_construct_features: null pointer");
    if (stride < 1) return fail(DS_E_ARG, "ds_construct_features: stride must be positive");
    for (int64_t i = 0; i < n; ++i)
        if (q_len[i] > stride || t_len[i] > stride)
            return fail(DS_E_ARG, "ds_construct_features: length of pair %lld exceeds the row stride %lld", (long long)i, (long long)stride);
    ds_oracle_construct_features(q_len, t_len, q_enc, t_enc, t_word_counts, space_code, n_truth, n, stride, 0, out);
    return DS_OK;
}

int ds_levenshtein_ratio(const uint8_t *a, int la, const uint8_t *b, int lb)
{
    if (la < 0 || lb < 0 || (la > 0 && !a) || (lb > 0 && !b)) return fail(DS_E_ARG, "ds_levenshtein_ratio: bad arguments");
    static const uint8_t none = 0;
    return (int)ds_oracle_levenshtein_ratio(la ? a : &none, la, lb ? b : &none, lb, 0);
}
