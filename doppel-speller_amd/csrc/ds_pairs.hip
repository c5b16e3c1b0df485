// Next row f-2: the pair list between the fuzzy step and the model, kept on the device.
//
// Reference (doppelspeller/predict.py):
//   :172-183  _find_close_matches saves the queries that the fuzzy step matched and returns
//             remaining.loc[~remaining[test_index].isin(matched_so_far)] -- every (query, candidate) row of the queries
//             it did NOT match, in the frame's order (query-major, the candidates in match_maker order);
//   :195-204  _find_matches_using_model gathers the encoded titles of exactly those rows for construct_features;
//   :246-252  per query the rows with the maximum prediction, those above PREDICTION_PROBABILITY_THRESHOLD, and the
//             query is matched only when a single row is left (_remove_duplicated_matches, :158-161).
//
// ds_remaining_pairs_device: an order-preserving stream compaction of the queries with best_row < 0 (three small
// kernels: per-block counts, scan of the block counts, per-block scan + write) that emits the (query row, truth row)
// index pairs ds_construct_features_indexed_device consumes -- the top-k rows never leave HBM.
// ds_select_matches_device: one thread per remaining query over its k predictions.
#include "ds_common.h"

namespace ds {

constexpr int kPairBlock = 1024;  // queries per workgroup of the compaction

__global__ __launch_bounds__(kPairBlock) void ds_pairs_count_kernel(const int32_t *best_row, int64_t n_queries,
                                                                    int64_t *block_counts)
{
    __shared__ int wave_counts[kPairBlock / 64];
    const int64_t q = static_cast<int64_t>(blockIdx.x) * kPairBlock + threadIdx.x;
    const bool remaining = q < n_queries && best_row[q] < 0;
    const int count = __popcll(__ballot(remaining));
    if ((threadIdx.x & 63) == 0) wave_counts[threadIdx.x >> 6] = count;
    __syncthreads();
    if (threadIdx.x == 0) {
        int total = 0;
        for (int w = 0; w < kPairBlock / 64; ++w) total += wave_counts[w];
        block_counts[blockIdx.x] = total;
    }
}

// exclusive scan of the block counts in place (one workgroup; the number of blocks is Q / 1024), totals in counts[0..1]
__global__ __launch_bounds__(1024) void ds_pairs_scan_kernel(int64_t *block_counts, int64_t n_blocks, int32_t k,
                                                             int64_t *counts)
{
    __shared__ int64_t partial[1024];
    __shared__ int64_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int64_t base = 0; base < n_blocks; base += 1024) {
        const int64_t i = base + threadIdx.x;
        const int64_t mine = i < n_blocks ? block_counts[i] : 0;
        partial[threadIdx.x] = mine;
        __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) {  // Hillis-Steele inclusive scan
            const int64_t add = static_cast<int>(threadIdx.x) >= d ? partial[threadIdx.x - d] : 0;
            __syncthreads();
            partial[threadIdx.x] += add;
            __syncthreads();
        }
        if (i < n_blocks) block_counts[i] = carry + partial[threadIdx.x] - mine;
        __syncthreads();
        if (threadIdx.x == 0) carry += partial[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        counts[0] = carry;
        counts[1] = carry * k;
    }
}

__global__ __launch_bounds__(kPairBlock) void ds_pairs_write_kernel(const int32_t *best_row, const int32_t *rows,
                                                                    int64_t n_queries, int32_t k, int64_t q_first,
                                                                    const int64_t *block_offsets, int32_t *pair_q,
                                                                    int32_t *pair_t)
{
    __shared__ int wave_counts[kPairBlock / 64];
    __shared__ int32_t kept_query[kPairBlock];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t q = static_cast<int64_t>(blockIdx.x) * kPairBlock + threadIdx.x;
    const bool remaining = q < n_queries && best_row[q] < 0;
    const unsigned long long votes = __ballot(remaining);
    if (lane == 0) wave_counts[wave] = __popcll(votes);
    __syncthreads();
    int before = 0, total = 0;
    for (int w = 0; w < kPairBlock / 64; ++w) {
        before += w < wave ? wave_counts[w] : 0;
        total += wave_counts[w];
    }
    if (remaining) kept_query[before + __popcll(votes & ((1ull << lane) - 1ull))] = static_cast<int32_t>(q);
    __syncthreads();
    // the block's remaining queries, k candidates each: consecutive threads write consecutive pairs
    const int64_t first_pair = block_offsets[blockIdx.x] * k;
    for (int64_t i = threadIdx.x; i < static_cast<int64_t>(total) * k; i += kPairBlock) {
        const int32_t query = kept_query[i / k];
        pair_q[first_pair + i] = static_cast<int32_t>(q_first) + query;
        pair_t[first_pair + i] = rows[static_cast<int64_t>(query) * k + i % k];
    }
}

__global__ void ds_select_matches_kernel(const int32_t *pair_q, const int32_t *pair_t, const float *predictions,
                                         int64_t n_remaining, int32_t k, float threshold, int32_t *match_query,
                                         int32_t *match_row)
{
    const int64_t r = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (r >= n_remaining) return;
    float best = predictions[r * k];
    int where = 0, count = 1;
    for (int j = 1; j < k; ++j) {  // predict.py:246-248: the rows that hold the query's maximum
        const float p = predictions[r * k + j];
        if (p > best) { best = p; where = j; count = 1; }
        else if (p == best) ++count;
    }
    match_query[r] = pair_q[r * k];
    match_row[r] = (best > threshold && count == 1) ? pair_t[r * k + where] : -1;  // :249-250 and :158-161
}

}  // namespace ds

extern "C" {

int64_t ds_remaining_pairs_counts_size(int64_t n_queries)
{
    return 2 + (n_queries + ds::kPairBlock - 1) / ds::kPairBlock;
}

int ds_remaining_pairs_device(const int32_t *d_best_row, const int32_t *d_rows, int64_t n_queries, int32_t k,
                              int64_t q_first, int32_t *d_pair_q, int32_t *d_pair_t, int64_t *d_counts, void *stream)
{
    DS_REQUIRE(n_queries >= 0 && k >= 1, "ds_remaining_pairs_device: bad query count / k");
    DS_REQUIRE(d_counts != nullptr, "ds_remaining_pairs_device: null counts");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (n_queries == 0) {
        DS_HIP(hipMemsetAsync(d_counts, 0, 2 * sizeof(int64_t), s));
        return DS_OK;
    }
    DS_REQUIRE(d_best_row && d_rows && d_pair_q && d_pair_t, "ds_remaining_pairs_device: null pointer");
    DS_REQUIRE(n_queries * k < (int64_t(1) << 40), "ds_remaining_pairs_device: too many pairs");
    const int64_t n_blocks = (n_queries + ds::kPairBlock - 1) / ds::kPairBlock;
    int64_t *block_counts = d_counts + 2;
    hipLaunchKernelGGL(ds::ds_pairs_count_kernel, dim3(static_cast<unsigned>(n_blocks)), dim3(ds::kPairBlock), 0, s,
                       d_best_row, n_queries, block_counts);
    DS_HIP(hipGetLastError());
    hipLaunchKernelGGL(ds::ds_pairs_scan_kernel, dim3(1), dim3(1024), 0, s, block_counts, n_blocks, k, d_counts);
    DS_HIP(hipGetLastError());
    hipLaunchKernelGGL(ds::ds_pairs_write_kernel, dim3(static_cast<unsigned>(n_blocks)), dim3(ds::kPairBlock), 0, s,
                       d_best_row, d_rows, n_queries, k, q_first, block_counts, d_pair_q, d_pair_t);
    DS_HIP(hipGetLastError());
    return DS_OK;
}

int ds_select_matches_device(const int32_t *d_pair_q, const int32_t *d_pair_t, const float *d_predictions,
                             int64_t n_remaining, int32_t k, float threshold, int32_t *d_match_query,
                             int32_t *d_match_row, void *stream)
{
    DS_REQUIRE(n_remaining >= 0 && k >= 1, "ds_select_matches_device: bad count / k");
    if (n_remaining == 0) return DS_OK;
    DS_REQUIRE(d_pair_q && d_pair_t && d_predictions && d_match_query && d_match_row,
               "ds_select_matches_device: null pointer");
    hipLaunchKernelGGL(ds::ds_select_matches_kernel, dim3(static_cast<unsigned>((n_remaining + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), d_pair_q, d_pair_t, d_predictions, n_remaining, k, threshold,
                       d_match_query, d_match_row);
    DS_HIP(hipGetLastError());
    return DS_OK;
}

}  // extern "C"
