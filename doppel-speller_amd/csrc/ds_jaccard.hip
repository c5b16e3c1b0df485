// IDF-weighted Jaccard of a batch of query titles against every truth title + the reference's threshold top-k.
//
// Reference semantics (doppelspeller/match_maker.py):
//   fast_jaccard   :16-50   scores[t] = float32 sum, in the query's column order, of idf32[g] over the query's
//                           n-gram columns g whose posting list contains t;
//                           jaccard[t] = float64(scores[t]) / (float64(sums[t]) + (maxint - float64(scores[t])))
//   fast_arg_top_k :53-71   m = float32(k-th largest positive jaccard, 0 if fewer than k);
//                           threshold = float64(m) - float64(float32(1e-6));
//                           result = the k LARGEST ROW INDEXES among {t : jaccard[t] >= threshold}, descending.
//
// Two kernels (DESIGN.md section "Jaccard kernels"):
//   ds_jaccard_topk_kernel   one 1024-thread workgroup per query pulled from a work queue.  Per tile of 32768 truth
//       rows: (1) scatter -- every posting of the query's columns adds idf32[g] into a float32 score tile in LDS with
//       order-free LDS atomics (approximate: the float32 rounding depends on the order); (2) scan -- rows whose
//       approximate jaccard can still reach the running k-th largest value (minus a rigorous error margin) are
//       appended to a candidate buffer; a radix select over the buffer tightens the running value.  After the last
//       tile only the surviving candidates (k + a few) are evaluated EXACTLY: membership of the row in each query
//       column's posting list by binary search, float32 sum in the reference's column order, float64 finalise,
//       then the reference's threshold/arg-select.  Results are bit-exact; the approximation only decides where the
//       exact arithmetic is spent.
//   ds_jaccard_dense_kernel  the literal algorithm (ordered scatter with a barrier per column, dense float64
//       jaccard row in HBM, radix select of the k-th float32 value, descending collect) for the queries the fast
//       kernel cannot bound: more than 256 columns, maxint <= 0, fewer than k positive rows, massive ties.
#include <cfloat>
#include <cstdlib>

#include "ds_common.h"

namespace ds {

struct JaccardArgs {
    const uint32_t *tile_ptr;
    const uint16_t *postings;
    const float *idf32;
    const float *sums32;
    const int64_t *q_rowptr;
    const int32_t *q_cols;
    const double *q_maxint;
    int32_t *out_rows;
    int32_t *status;
    int32_t *control;
    int32_t *slow_list;
    double *slow_scratch;
    unsigned long long *phase;  // nullable: per-phase shader-clock sums (diagnostics, DS_PHASE_TIMERS=1)
    int64_t n_truth;
    int64_t n_columns;
    int64_t n_queries;
    int32_t n_tiles;
    int32_t k;
    float sums_min;
};

// control words in HBM
enum { kCtlQueue = 0, kCtlSlowCount = 1, kCtlErrors = 2, kCtlExact = 3, kCtlSelects = 4, kCtlSlowQueue = 5 };

// LDS carve-up of the fast kernel (bytes)
constexpr int kScoreFloats = kTile + 64;  // + trash slot for the padding entries of a quad
constexpr int kOffKey = kScoreFloats * 4;
constexpr int kOffRow = kOffKey + kCandidates * 4;
constexpr int kOffCols = kOffRow + kCandidates * 4;
constexpr int kOffIdf = kOffCols + kMaxQueryColumns * 4;
constexpr int kOffBegin = kOffIdf + kMaxQueryColumns * 4;
constexpr int kOffEnd = kOffBegin + kMaxQueryColumns * 4;
constexpr int kOffHist = kOffEnd + kMaxQueryColumns * 4;
constexpr int kOffCtrl = kOffHist + 256 * 4;
constexpr int kFastLdsBytes = kOffCtrl + 64;
static_assert(kFastLdsBytes <= 160 * 1024, "LDS budget of one CU exceeded");
constexpr int kSelectTrigger = kCandidates - kLooseStep;

// LDS control words
enum { kLQuery = 0, kLCount, kLOverflow, kLDigit, kLRemain, kLCoef, kLPre, kLCut, kLKth0, kLKth1, kLBad };

__device__ __forceinline__ float round_down_positive(double x)
{
    float f = static_cast<float>(x);
    if (static_cast<double>(f) > x) f = __uint_as_float(__float_as_uint(f) - 1u);
    return f;
}

// k-th largest key of cand_key[0..m) (m >= k) by a 4 x 8-bit radix select; result in ctrl[kLDigit] history -> return.
__device__ uint32_t radix_select_kth(const uint32_t *cand_key, int m, int k, uint32_t *hist, volatile int32_t *ctrl)
{
    const int tid = threadIdx.x, lane = tid & 63;
    uint32_t prefix = 0, mask = 0;
    int remaining = k;
    for (int shift = 24; shift >= 0; shift -= 8) {
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        for (int i = tid; i < m; i += kThreads) {
            const uint32_t key = cand_key[i];
            if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (tid < 64) {  // lane owns bins 255-4*lane .. 252-4*lane (descending)
            const int b0 = 255 - 4 * lane;
            const uint32_t c0 = hist[b0], c1 = hist[b0 - 1], c2 = hist[b0 - 2], c3 = hist[b0 - 3];
            const uint32_t local = c0 + c1 + c2 + c3;
            uint32_t inclusive = local;
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t other = __shfl_up(inclusive, d);
                if (lane >= d) inclusive += other;
            }
            const uint32_t exclusive = inclusive - local;
            const uint32_t want = static_cast<uint32_t>(remaining);
            if (want > exclusive && want <= inclusive) {
                uint32_t r = want - exclusive;
                int digit = b0;
                if (r > c0) { r -= c0; digit = b0 - 1;
                    if (r > c1) { r -= c1; digit = b0 - 2;
                        if (r > c2) { r -= c2; digit = b0 - 3; } } }
                ctrl[kLDigit] = digit;
                ctrl[kLRemain] = static_cast<int32_t>(r);
            }
        }
        __syncthreads();
        prefix |= static_cast<uint32_t>(ctrl[kLDigit]) << shift;
        mask |= 255u << shift;
        remaining = ctrl[kLRemain];
    }
    return prefix;
}

// Diagnostic phase timers: thread 0 adds the shader-clock delta since its previous stamp to phase[slot].
#define DS_STAMP(slot)                                                     \
    do {                                                                   \
        if (a.phase != nullptr && tid == 0) {                              \
            const unsigned long long now_ = __builtin_amdgcn_s_memtime();  \
            atomicAdd(&a.phase[slot], now_ - stamp_);                      \
            stamp_ = now_;                                                 \
        }                                                                  \
    } while (0)

__global__ __launch_bounds__(kThreads) void ds_jaccard_topk_kernel(JaccardArgs a)
{
    unsigned long long stamp_ = a.phase != nullptr ? __builtin_amdgcn_s_memtime() : 0ull;
    extern __shared__ __align__(16) unsigned char lds[];
    float *scores = reinterpret_cast<float *>(lds);
    uint32_t *cand_key = reinterpret_cast<uint32_t *>(lds + kOffKey);
    int32_t *cand_row = reinterpret_cast<int32_t *>(lds + kOffRow);
    int32_t *cols = reinterpret_cast<int32_t *>(lds + kOffCols);
    float *idf = reinterpret_cast<float *>(lds + kOffIdf);
    uint32_t *list_begin = reinterpret_cast<uint32_t *>(lds + kOffBegin);
    uint32_t *list_end = reinterpret_cast<uint32_t *>(lds + kOffEnd);
    uint32_t *hist = reinterpret_cast<uint32_t *>(lds + kOffHist);
    volatile int32_t *ctrl = reinterpret_cast<volatile int32_t *>(lds + kOffCtrl);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int k = a.k;
    const int64_t ptr_stride = a.n_columns + 1;
    const uint2 *quads = reinterpret_cast<const uint2 *>(a.postings);

    for (int i = tid * 4; i < kScoreFloats; i += kThreads * 4)
        *reinterpret_cast<float4 *>(&scores[i]) = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();

    for (;;) {
        if (tid == 0) {
            ctrl[kLQuery] = atomicAdd(&a.control[kCtlQueue], 1);
            ctrl[kLCount] = 0;
            ctrl[kLOverflow] = 0;
            ctrl[kLBad] = 0;
        }
        __syncthreads();
        const int64_t q = ctrl[kLQuery];
        if (q >= a.n_queries) break;  // exit condition reached by every wave: the queue only grows

        const int64_t qbase = a.q_rowptr[q];
        const int64_t n64 = a.q_rowptr[q + 1] - qbase;
        const double maxint = a.q_maxint[q];
        bool slow = n64 > kMaxQueryColumns || n64 < 0 || !(maxint > 0.0) || !(maxint < 1e30);
        const int n = slow ? 0 : static_cast<int>(n64);
        if (tid < n) {
            const int32_t column = a.q_cols[qbase + tid];
            const bool bad = column < 0 || column >= a.n_columns;
            if (bad) ctrl[kLBad] = 1;
            cols[tid] = bad ? 0 : column;
            idf[tid] = bad ? 0.f : a.idf32[column];
        }
        __syncthreads();
        if (ctrl[kLBad]) {
            if (tid == 0) {
                a.status[q] = kQueryErrorArg;
                atomicAdd(&a.control[kCtlErrors], 1);
            }
            for (int j = tid; j < k; j += kThreads) a.out_rows[q * k + j] = -1;
            __syncthreads();
            continue;
        }

        DS_STAMP(0);
        const float maxint32 = static_cast<float>(maxint);
        // |float32 approximate jaccard - exact jaccard| <= margin (see DESIGN.md "error margin of the prefilter")
        const double margin = (6.0 * n + 64.0) * 5.9604644775390625e-08;
        float coef = 0.f, pre = FLT_MIN, cut = 0.f;  // scan test: s >= pre && s >= coef * (sums[t] + maxint32)
        bool tight = false;
        int next_select = max(4 * k, 64);
        if (next_select > kSelectTrigger) next_select = kSelectTrigger;
        int selects = 0;

        for (int b = 0; b < a.n_tiles && !slow; ++b) {
            const uint32_t *ptr_row = a.tile_ptr + static_cast<int64_t>(b) * ptr_stride;
            if (tid < n) {
                list_begin[tid] = ptr_row[cols[tid]];
                list_end[tid] = ptr_row[cols[tid] + 1];
            }
            __syncthreads();
            DS_STAMP(1);
            // (1) scatter: order-free float32 LDS atomics; padding entries hit the trash slot scores[kTile]
            for (int j = 0; j < n; ++j) {
                const uint32_t begin = list_begin[j], end = list_end[j];
                const float value = idf[j];
                for (uint32_t i = begin + tid; i < end; i += kThreads) {
                    const uint2 quad = quads[i];
                    atomicAdd(&scores[quad.x & 0xffffu], value);
                    atomicAdd(&scores[quad.x >> 16], value);
                    atomicAdd(&scores[quad.y & 0xffffu], value);
                    atomicAdd(&scores[quad.y >> 16], value);
                }
            }
            __syncthreads();
            DS_STAMP(2);
            // (2) scan (and re-zero) the tile
            const int64_t tile_base = static_cast<int64_t>(b) << kTileLog2;
            const int64_t rows_left = a.n_truth - tile_base;
            const int limit = rows_left >= kTile ? kTile : static_cast<int>((rows_left + 3) & ~int64_t(3));
            int r0 = 0;
            while (r0 < limit) {
                const int r1 = tight ? limit : min(r0 + kLooseStep, limit);
                for (int idx = r0 + tid * 4; idx < r1; idx += kThreads * 4) {
                    const float4 s4 = *reinterpret_cast<float4 *>(&scores[idx]);
                    *reinterpret_cast<float4 *>(&scores[idx]) = make_float4(0.f, 0.f, 0.f, 0.f);
                    const bool any = s4.x >= pre || s4.y >= pre || s4.z >= pre || s4.w >= pre;
                    if (__ballot(any) == 0) continue;
                    const float sv[4] = {s4.x, s4.y, s4.z, s4.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float s = sv[e];
                        const int64_t t = tile_base + idx + e;
                        bool pass = s >= pre;
                        uint32_t key = 0;
                        if (pass) {
                            const float sums = a.sums32[t];
                            pass = s >= coef * (sums + maxint32);
                            if (pass) {
                                const float denominator = sums + (maxint32 - s);
                                const float approx = s / denominator;
                                key = (denominator > 0.f && approx == approx) ? __float_as_uint(approx) : 0x7f800000u;
                            }
                        }
                        const unsigned long long votes = __ballot(pass);
                        if (votes != 0) {
                            const int leader = __ffsll(votes) - 1;
                            int base = 0;
                            if (lane == leader) base = atomicAdd(const_cast<int32_t *>(&ctrl[kLCount]), __popcll(votes));
                            base = __shfl(base, leader);
                            if (pass) {
                                const int slot = base + __popcll(votes & ((1ull << lane) - 1ull));
                                if (slot < kCandidates) {
                                    cand_key[slot] = key;
                                    cand_row[slot] = static_cast<int32_t>(t);
                                } else {
                                    ctrl[kLOverflow] = 1;
                                }
                            }
                        }
                    }
                }
                __syncthreads();
                DS_STAMP(3);
                r0 = r1;
                if (ctrl[kLOverflow]) { slow = true; break; }
                const int m = ctrl[kLCount];
                if (m >= next_select) {
                    // tighten: tau = k-th largest approximate jaccard seen so far; keep what can still qualify
                    const uint32_t tau_key = radix_select_kth(cand_key, m, k, hist, ctrl);
                    ++selects;
                    if (tid == 0) {
                        const double tau = static_cast<double>(__uint_as_float(tau_key));
                        const double cut_value = tau - 2.0 * margin - 2e-6;
                        float new_coef = 0.f, new_pre = FLT_MIN, new_cut = 0.f;
                        if (tau_key < 0x7f800000u && cut_value > 0.0) {
                            const double c = cut_value / (1.0 + cut_value) * (1.0 - 9.5367431640625e-07);
                            new_coef = round_down_positive(c);
                            new_cut = round_down_positive(cut_value);
                            const double p = static_cast<double>(new_coef) *
                                             (static_cast<double>(a.sums_min) + static_cast<double>(maxint32)) *
                                             (1.0 - 9.5367431640625e-07);
                            new_pre = p > static_cast<double>(FLT_MIN) ? round_down_positive(p) : FLT_MIN;
                        }
                        ctrl[kLCoef] = __float_as_int(new_coef);
                        ctrl[kLPre] = __float_as_int(new_pre);
                        ctrl[kLCut] = __float_as_int(new_cut);
                    }
                    __syncthreads();
                    coef = __int_as_float(ctrl[kLCoef]);
                    pre = __int_as_float(ctrl[kLPre]);
                    cut = __int_as_float(ctrl[kLCut]);
                    tight = true;
                    // in-place compaction: read everything, barrier, rewrite the survivors
                    uint32_t keep_key[kCandidates / kThreads];
                    int32_t keep_row[kCandidates / kThreads];
#pragma unroll
                    for (int r = 0; r < kCandidates / kThreads; ++r) {
                        const int i = tid + r * kThreads;
                        keep_key[r] = i < m ? cand_key[i] : 0u;
                        keep_row[r] = i < m ? cand_row[i] : -1;
                    }
                    __syncthreads();
                    if (tid == 0) ctrl[kLCount] = 0;
                    __syncthreads();
#pragma unroll
                    for (int r = 0; r < kCandidates / kThreads; ++r) {
                        if (keep_row[r] >= 0 && __uint_as_float(keep_key[r]) >= cut) {
                            const int slot = atomicAdd(const_cast<int32_t *>(&ctrl[kLCount]), 1);
                            cand_key[slot] = keep_key[r];
                            cand_row[slot] = keep_row[r];
                        }
                    }
                    __syncthreads();
                    const int kept = ctrl[kLCount];
                    if (kept > kSelectTrigger) { slow = true; break; }  // massive ties: use the dense kernel
                    next_select = min(kSelectTrigger, max(2 * kept, max(4 * k, 64)));
                    DS_STAMP(4);
                }
            }
        }

        int m = slow ? 0 : ctrl[kLCount];
        if (!slow && m < k) slow = true;  // fewer than k positive rows (never tightened): literal path decides
        if (!slow) {
            // final tightening so that only k + near-ties + margin survivors are evaluated exactly
            if (m > k) {
                const uint32_t tau_key = radix_select_kth(cand_key, m, k, hist, ctrl);
                ++selects;
                const double tau = static_cast<double>(__uint_as_float(tau_key));
                const double cut_value = tau - 2.0 * margin - 2e-6;
                const float final_cut = (tau_key < 0x7f800000u && cut_value > 0.0) ? round_down_positive(cut_value) : 0.f;
                uint32_t keep_key[kCandidates / kThreads];
                int32_t keep_row[kCandidates / kThreads];
#pragma unroll
                for (int r = 0; r < kCandidates / kThreads; ++r) {
                    const int i = tid + r * kThreads;
                    keep_key[r] = i < m ? cand_key[i] : 0u;
                    keep_row[r] = i < m ? cand_row[i] : -1;
                }
                __syncthreads();
                if (tid == 0) ctrl[kLCount] = 0;
                __syncthreads();
#pragma unroll
                for (int r = 0; r < kCandidates / kThreads; ++r) {
                    if (keep_row[r] >= 0 && __uint_as_float(keep_key[r]) >= final_cut) {
                        const int slot = atomicAdd(const_cast<int32_t *>(&ctrl[kLCount]), 1);
                        cand_row[slot] = keep_row[r];
                    }
                }
                __syncthreads();
                m = ctrl[kLCount];
            }
            // exact evaluation, one wave per candidate; results live in the (all-zero) score tile
            double *exact_jaccard = reinterpret_cast<double *>(scores);
            int32_t *exact_row = reinterpret_cast<int32_t *>(scores + 2 * kCandidates);
            for (int i = wave; i < m; i += kThreads / 64) {
                const int32_t t = cand_row[i];
                const int32_t tile = t >> kTileLog2;
                const uint32_t local = static_cast<uint32_t>(t & (kTile - 1));
                const uint32_t *ptr_row = a.tile_ptr + static_cast<int64_t>(tile) * ptr_stride;
                float score = 0.f;
                for (int c0 = 0; c0 < n; c0 += 64) {
                    const int j = c0 + lane;
                    bool hit = false;
                    if (j < n) {
                        uint32_t lo = ptr_row[cols[j]] * 4u;
                        const uint32_t end = ptr_row[cols[j] + 1] * 4u;
                        uint32_t hi = end;
                        while (lo < hi) {
                            const uint32_t mid = (lo + hi) >> 1;
                            if (a.postings[mid] < local) lo = mid + 1; else hi = mid;
                        }
                        hit = lo < end && a.postings[lo] == local;
                    }
                    unsigned long long hits = __ballot(hit);
                    while (hits) {  // float32 accumulation in the query's column order (match_maker.py:46-48)
                        const int jj = __ffsll(hits) - 1;
                        score = score + idf[c0 + jj];
                        hits &= hits - 1ull;
                    }
                }
                if (lane == 0) {
                    const double s = static_cast<double>(score);
                    exact_jaccard[i] = s / (static_cast<double>(a.sums32[t]) + (maxint - s));  // match_maker.py:50
                    exact_row[i] = t;
                }
            }
            __syncthreads();
            // k-th largest exact value (rank by counting; m is small)
            for (int i = tid; i < m; i += kThreads) {
                const double v = exact_jaccard[i];
                int rank = 0;
                for (int j = 0; j < m; ++j) {
                    const double w = exact_jaccard[j];
                    rank += (w > v) || (w == v && j < i);
                }
                if (rank == k - 1) {
                    ctrl[kLKth0] = __double2loint(v);
                    ctrl[kLKth1] = __double2hiint(v);
                }
            }
            __syncthreads();
            const double kth = __hiloint2double(ctrl[kLKth1], ctrl[kLKth0]);
            const float kth32 = static_cast<float>(kth);                                   // match_maker.py:65
            const double threshold = static_cast<double>(kth32) - static_cast<double>(1e-6f);  // match_maker.py:70
            if (threshold <= 0.0) {
                // every row qualifies (zeros included): the k largest row indexes
                for (int j = tid; j < k; j += kThreads) a.out_rows[q * k + j] = static_cast<int32_t>(a.n_truth - 1 - j);
            } else {
                for (int i = tid; i < m; i += kThreads) {
                    if (!(exact_jaccard[i] >= threshold)) continue;                          // match_maker.py:71
                    const int32_t t = exact_row[i];
                    int above = 0;
                    for (int j = 0; j < m; ++j) above += (exact_jaccard[j] >= threshold) && exact_row[j] > t;
                    if (above < k) a.out_rows[q * k + above] = t;
                }
            }
            __syncthreads();
            for (int i = tid; i < 3 * kCandidates; i += kThreads) scores[i] = 0.f;  // give the tile back zeroed
            if (tid == 0) {
                a.status[q] = kQueryDone;
                atomicAdd(&a.control[kCtlExact], m);
                atomicAdd(&a.control[kCtlSelects], selects);
            }
            __syncthreads();
            DS_STAMP(5);
        } else {
            if (tid == 0) {
                a.status[q] = kQuerySlow;
                a.slow_list[atomicAdd(&a.control[kCtlSlowCount], 1)] = static_cast<int32_t>(q);
            }
            for (int i = tid * 4; i < kScoreFloats; i += kThreads * 4)
                *reinterpret_cast<float4 *>(&scores[i]) = make_float4(0.f, 0.f, 0.f, 0.f);
            __syncthreads();
            DS_STAMP(6);
        }
    }
}

// ---- the literal algorithm for the queries the fast kernel hands over ------------------------------------------------
constexpr int kDenseLdsBytes = kScoreFloats * 4 + 256 * 4 + 64 + (kThreads / 64) * 4;

__global__ __launch_bounds__(kThreads) void ds_jaccard_dense_kernel(JaccardArgs a)
{
    extern __shared__ __align__(16) unsigned char lds[];
    float *scores = reinterpret_cast<float *>(lds);
    uint32_t *hist = reinterpret_cast<uint32_t *>(lds + kScoreFloats * 4);
    volatile int32_t *ctrl = reinterpret_cast<volatile int32_t *>(lds + kScoreFloats * 4 + 1024);
    int32_t *wave_counts = const_cast<int32_t *>(ctrl) + 16;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int k = a.k;
    const int64_t n_truth = a.n_truth;
    const int64_t ptr_stride = a.n_columns + 1;
    double *jaccard = a.slow_scratch + static_cast<int64_t>(blockIdx.x) * n_truth;
    const int n_slow = a.control[kCtlSlowCount];

    for (;;) {
        if (tid == 0) ctrl[kLQuery] = atomicAdd(&a.control[kCtlSlowQueue], 1);
        __syncthreads();
        const int item = ctrl[kLQuery];
        __syncthreads();
        if (item >= n_slow) break;
        const int64_t q = a.slow_list[item];
        const int64_t qbase = a.q_rowptr[q];
        const int64_t n = a.q_rowptr[q + 1] - qbase;
        const double maxint = a.q_maxint[q];

        bool bad = false;
        for (int64_t j = tid; j < n; j += kThreads) {
            const int32_t column = a.q_cols[qbase + j];
            bad |= column < 0 || column >= a.n_columns;
        }
        if (__syncthreads_or(bad || n < 0)) {
            if (tid == 0) {
                a.status[q] = kQueryErrorArg;
                atomicAdd(&a.control[kCtlErrors], 1);
            }
            for (int j = tid; j < k; j += kThreads) a.out_rows[q * k + j] = -1;
            continue;
        }

        // fast_jaccard, tile by tile: ordered float32 accumulation (a barrier between two columns), float64 finalise
        for (int b = 0; b < a.n_tiles; ++b) {
            for (int i = tid; i < kScoreFloats; i += kThreads) scores[i] = 0.f;
            __syncthreads();
            const uint32_t *ptr_row = a.tile_ptr + static_cast<int64_t>(b) * ptr_stride;
            for (int64_t j = 0; j < n; ++j) {
                const int32_t column = a.q_cols[qbase + j];
                const float value = a.idf32[column];
                const uint32_t begin = ptr_row[column] * 4u, end = ptr_row[column + 1] * 4u;
                for (uint32_t i = begin + tid; i < end; i += kThreads) {
                    const uint32_t local = a.postings[i];
                    if (local < kTile) scores[local] = scores[local] + value;  // each row at most once per list
                }
                __syncthreads();
            }
            const int64_t tile_base = static_cast<int64_t>(b) << kTileLog2;
            for (int i = tid; i < kTile && tile_base + i < n_truth; i += kThreads) {
                const double s = static_cast<double>(scores[i]);
                jaccard[tile_base + i] = s / (static_cast<double>(a.sums32[tile_base + i]) + (maxint - s));
            }
            __syncthreads();
        }
        __threadfence_block();

        // fast_arg_top_k: the float32 heap ends as the k largest float32(value > 0) padded with zeros
        uint32_t prefix = 0, mask = 0;
        int remaining = k;
        bool fewer = false;
        for (int shift = 24; shift >= 0; shift -= 8) {
            if (tid < 256) hist[tid] = 0;
            __syncthreads();
            for (int64_t t = tid; t < n_truth; t += kThreads) {
                const double v = jaccard[t];
                const uint32_t key = v > 0.0 ? __float_as_uint(static_cast<float>(v)) : 0u;
                if (key != 0u && (key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
            }
            __syncthreads();
            if (tid == 0) {
                int cumulative = 0, digit = -1;
                for (int d = 255; d >= 0; --d) {
                    const int c = static_cast<int>(hist[d]);
                    if (cumulative + c >= remaining) { digit = d; break; }
                    cumulative += c;
                }
                ctrl[kLDigit] = digit;
                ctrl[kLRemain] = remaining - cumulative;
            }
            __syncthreads();
            if (ctrl[kLDigit] < 0) { fewer = true; break; }  // fewer than k positive float32 values: k-th = 0
            prefix |= static_cast<uint32_t>(ctrl[kLDigit]) << shift;
            mask |= 255u << shift;
            remaining = ctrl[kLRemain];
            __syncthreads();
        }
        const float kth32 = fewer ? 0.f : __uint_as_float(prefix);
        const double threshold = static_cast<double>(kth32) - static_cast<double>(1e-6f);

        // (array >= threshold).nonzero()[0][::-1][:k]
        int found = 0;
        for (int64_t top = n_truth - 1; top >= 0 && found < k; top -= kThreads) {
            const int64_t t = top - tid;
            const bool pass = t >= 0 && jaccard[t] >= threshold;
            const unsigned long long votes = __ballot(pass);
            if (lane == 0) wave_counts[wave] = __popcll(votes);
            __syncthreads();
            int before = 0, total = 0;
            for (int w = 0; w < kThreads / 64; ++w) {
                const int c = wave_counts[w];
                before += w < wave ? c : 0;
                total += c;
            }
            if (pass) {
                const int slot = found + before + __popcll(votes & ((1ull << lane) - 1ull));
                if (slot < k) a.out_rows[q * k + slot] = static_cast<int32_t>(t);
            }
            found += total;
            __syncthreads();
        }
        if (found < k) {
            for (int j = found + tid; j < k; j += kThreads) a.out_rows[q * k + j] = -1;
            if (tid == 0) {
                a.status[q] = kQueryErrorTopN;
                atomicAdd(&a.control[kCtlErrors], 1);
            }
        }
        __syncthreads();
    }
}

static int launch(ds_index *index, const int64_t *d_q_rowptr, const int32_t *d_q_cols, const double *d_q_maxint,
                  int64_t Q, int32_t k, int32_t *d_out_rows, hipStream_t stream)
{
    DS_REQUIRE(index != nullptr, "ds_jaccard_topk: null index");
    DS_REQUIRE(Q >= 0 && Q < (int64_t(1) << 31) - 4096, "ds_jaccard_topk: bad query count %lld", (long long)Q);
    DS_REQUIRE(k >= 1, "ds_jaccard_topk: k must be >= 1");
    if (k > index->n_truth) {  // match_maker.py:188-189
        set_error("top_matches.shape[0] != self.top_n (k=%d > number of truth titles=%lld)", k,
                  (long long)index->n_truth);
        return DS_E_TOP_N;
    }
    DS_REQUIRE(k <= 512, "ds_jaccard_topk: k=%d above the supported maximum of 512", k);
    DS_HIP(hipSetDevice(index->device));
    index->last_queries = Q;
    if (Q == 0) return DS_OK;
    DS_REQUIRE(d_q_rowptr && d_q_cols && d_q_maxint && d_out_rows, "ds_jaccard_topk: null pointer");
    if (index->status.count < static_cast<size_t>(Q)) {
        int status = index->status.allocate(static_cast<size_t>(Q));
        if (status == DS_OK) status = index->slow_list.allocate(static_cast<size_t>(Q));
        if (status != DS_OK) return status;
    }
    static bool attributes_set = false;
    if (!attributes_set) {
        DS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(ds_jaccard_topk_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, kFastLdsBytes));
        DS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(ds_jaccard_dense_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, kDenseLdsBytes));
        attributes_set = true;
    }
    JaccardArgs args;
    args.tile_ptr = index->tile_ptr.ptr;
    args.postings = index->postings.ptr;
    args.idf32 = index->idf32.ptr;
    args.sums32 = index->sums32.ptr;
    args.q_rowptr = d_q_rowptr;
    args.q_cols = d_q_cols;
    args.q_maxint = d_q_maxint;
    args.out_rows = d_out_rows;
    args.status = index->status.ptr;
    args.control = index->control.ptr;
    args.slow_list = index->slow_list.ptr;
    args.slow_scratch = index->slow_scratch.ptr;
    args.phase = nullptr;
    if (const char *timers = getenv("DS_PHASE_TIMERS"); timers != nullptr && timers[0] == '1') {
        if (index->phase.count == 0 && index->phase.allocate(16) != DS_OK) return DS_E_HIP;
        DS_HIP(hipMemsetAsync(index->phase.ptr, 0, 16 * sizeof(unsigned long long), stream));
        args.phase = index->phase.ptr;
    }
    args.n_truth = index->n_truth;
    args.n_columns = index->n_columns;
    args.n_queries = Q;
    args.n_tiles = static_cast<int32_t>(index->n_tiles);
    args.k = k;
    args.sums_min = index->sums_min;

    hipDeviceProp_t properties;
    DS_HIP(hipGetDeviceProperties(&properties, index->device));
    const int64_t cus = properties.multiProcessorCount > 0 ? properties.multiProcessorCount : 256;
    const int grid = static_cast<int>(std::min<int64_t>(Q, cus));
    DS_HIP(hipMemsetAsync(index->control.ptr, 0, 16 * sizeof(int32_t), stream));
    hipLaunchKernelGGL(ds_jaccard_topk_kernel, dim3(grid), dim3(kThreads), kFastLdsBytes, stream, args);
    DS_HIP(hipGetLastError());
    hipLaunchKernelGGL(ds_jaccard_dense_kernel, dim3(kSlowSlots), dim3(kThreads), kDenseLdsBytes, stream, args);
    DS_HIP(hipGetLastError());
    return DS_OK;
}

static int collect(ds_index *index, hipStream_t stream, int64_t stats[16])
{
    DS_REQUIRE(index != nullptr, "ds_jaccard_sync: null index");
    DS_HIP(hipSetDevice(index->device));
    int32_t control[16] = {0};
    DS_HIP(hipMemcpyAsync(control, index->control.ptr, sizeof(control), hipMemcpyDeviceToHost, stream));
    DS_HIP(hipStreamSynchronize(stream));
    if (stats) {
        stats[0] = control[kCtlSlowCount];
        stats[1] = control[kCtlErrors];
        stats[2] = control[kCtlExact];
        stats[3] = control[kCtlSelects];
        for (int i = 4; i < 16; ++i) stats[i] = 0;
        if (index->phase.count) {
            unsigned long long phase[16];
            DS_HIP(hipMemcpy(phase, index->phase.ptr, sizeof(phase), hipMemcpyDeviceToHost));
            for (int i = 0; i < 8; ++i) stats[4 + i] = static_cast<int64_t>(phase[i]);
        }
    }
    if (control[kCtlErrors] != 0 && index->last_queries > 0) {
        std::vector<int32_t> status(static_cast<size_t>(index->last_queries));
        DS_HIP(hipMemcpy(status.data(), index->status.ptr, status.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
        for (size_t q = 0; q < status.size(); ++q) {
            if (status[q] == kQueryErrorTopN) {
                set_error("top_matches.shape[0] != self.top_n (query %zu)", q);
                return DS_E_TOP_N;
            }
            if (status[q] == kQueryErrorArg) {
                set_error("ds_jaccard_topk: query %zu has a column index outside [0, V)", q);
                return DS_E_ARG;
            }
        }
    }
    return DS_OK;
}

}  // namespace ds

extern "C" {

int ds_jaccard_topk_device(ds_index *index, const int64_t *d_q_rowptr, const int32_t *d_q_cols,
                           const double *d_q_maxint, int64_t Q, int32_t k, int32_t *d_out_rows, void *stream)
{
    return ds::launch(index, d_q_rowptr, d_q_cols, d_q_maxint, Q, k, d_out_rows, static_cast<hipStream_t>(stream));
}

int ds_jaccard_sync(ds_index *index, void *stream, int64_t stats[16])
{
    return ds::collect(index, static_cast<hipStream_t>(stream), stats);
}

int ds_jaccard_topk(ds_index *index, const int64_t *q_rowptr, const int32_t *q_cols, const double *q_maxint,
                    int64_t Q, int32_t k, int32_t *out_rows)
{
    DS_REQUIRE(index != nullptr, "ds_jaccard_topk: null index");
    DS_REQUIRE(Q >= 0, "ds_jaccard_topk: negative query count");
    if (Q == 0) return DS_OK;
    DS_REQUIRE(q_rowptr && q_maxint && out_rows, "ds_jaccard_topk: null pointer");
    DS_REQUIRE(q_rowptr[0] == 0, "ds_jaccard_topk: q_rowptr[0] must be 0");
    const int64_t q_nnz = q_rowptr[Q];
    DS_REQUIRE(q_nnz >= 0 && (q_nnz == 0 || q_cols), "ds_jaccard_topk: bad q_rowptr / q_cols");
    DS_HIP(hipSetDevice(index->device));
    ds::DeviceBuffer<int64_t> d_rowptr;
    ds::DeviceBuffer<int32_t> d_cols, d_out;
    ds::DeviceBuffer<double> d_maxint;
    int status = d_rowptr.upload(q_rowptr, static_cast<size_t>(Q + 1));
    if (status == DS_OK) status = d_cols.upload(q_cols, static_cast<size_t>(q_nnz > 0 ? q_nnz : 0));
    if (status == DS_OK && q_nnz == 0) status = d_cols.allocate(1);
    if (status == DS_OK) status = d_maxint.upload(q_maxint, static_cast<size_t>(Q));
    if (status == DS_OK) status = d_out.allocate(static_cast<size_t>(Q) * static_cast<size_t>(k > 0 ? k : 1));
    if (status != DS_OK) return status;
    status = ds::launch(index, d_rowptr.ptr, d_cols.ptr, d_maxint.ptr, Q, k, d_out.ptr, index->stream);
    if (status != DS_OK) return status;
    status = ds::collect(index, index->stream, nullptr);
    DS_HIP(hipMemcpy(out_rows, d_out.ptr, static_cast<size_t>(Q) * k * sizeof(int32_t), hipMemcpyDeviceToHost));
    return status;
}

}  // extern "C"
