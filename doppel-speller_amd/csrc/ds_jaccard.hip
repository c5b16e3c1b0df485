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
// Two kernels:
//   ds_jaccard_topk_kernel   512-thread workgroups, two per CU, each pulling queries from a work queue.  The truth
//       rows are visited tile by tile (28672 rows = one tile of 16-bit fixed-point scores, two rows per LDS word).
//       Scores are accumulated with order-free LDS atomics, i.e. only APPROXIMATELY (the reference's float32 rounding
//       depends on the column order); rows whose approximate jaccard can still reach the running k-th largest value
//       minus a rigorous error margin become candidates; a radix select over the candidate buffer tightens the
//       running value.  Once a running value exists, columns whose total IDF cannot lift a row over it on their own
//       ("non-essential", the MaxScore rule of top-k retrieval) are no longer traversed: their IDF mass enters the
//       test as an upper bound.  Tiles with few essential postings are handled sparsely (scatter, then a collect
//       sweep over the same postings that takes and re-zeroes the touched rows) instead of scanning the whole tile.
//       After the last tile the surviving candidates are evaluated EXACTLY: membership of the row in each query
//       column's posting list by binary search, float32 sum in the reference's column order, float64 finalise, then
//       the reference's threshold / arg-select.  Results are bit-exact; the approximation only decides where the
//       exact arithmetic is spent.
//   ds_jaccard_dense_kernel  the literal algorithm (ordered scatter with a barrier per column, dense float64
//       jaccard row in HBM, radix select of the k-th float32 value, descending collect) for the queries the fast
//       kernel cannot bound: more than 128 columns, maxint <= 0, fewer than k positive rows, massive ties.
#include <cfloat>
#include <cstdlib>

#include "ds_common.h"

namespace ds {

struct JaccardArgs {
    const uint32_t *col_ptr;
    const uint16_t *postings;
    const uint16_t *posting_sums;
    const float *idf32;
    const float *sums32;
    const float *tile_sums_min;
    const uint4 *signature;
    const int8_t *sig_column;
    const uint16_t *dup_rank;
    const int64_t *q_rowptr;
    const int32_t *q_cols;
    const double *q_maxint;
    int32_t *out_rows;
    int32_t *status;
    int32_t *control;
    int32_t *slow_list;
    unsigned long long *phase;  // nullable: per-phase shader-clock sums (diagnostics, DS_PHASE_TIMERS=1)
    int64_t n_truth;
    int64_t n_columns;
    int64_t n_queries;
    int32_t n_tiles;
    int32_t k;
    int32_t sparse_quads;       // tiles with at most this many essential quads are handled sparsely
    int32_t literal_only;       // index holds values outside the fast kernel's assumptions: hand every query over
    int32_t select_min;         // candidates that trigger the first selections
    int32_t select_growth;      // next selection at select_growth / 2 times the kept candidates
    int32_t debug;              // timing experiments only (DS_DEBUG)
    int64_t n_quads;            // posting quads in the index (bounds of `postings`)
    float sums_min;
};

// control words in HBM
enum { kCtlQueue = 0, kCtlSlowCount = 1, kCtlErrors = 2, kCtlExact = 3, kCtlSelects = 4, kCtlSlowQueue = 5,
       kCtlSparseTiles = 6, kCtlDenseTiles = 7, kCtlSkippedColumns = 8, kCtlReason = 9 /* 9..14 */,
       kCtlRefines = 16, kCtlRawEntries = 17, kCtlSurvivors = 18, kCtlRawSparse = 19,
       kCtlBytes = 28 /* 28..29: uint64, bytes the fast kernel requested from global memory (all queries) */ };

// LDS carve-up of the fast kernel (bytes)
constexpr int kScoreWords = kTile / 2 + 16;  // two 16-bit scores per word + the trash word of the padding entries
constexpr int kOffLo = kScoreWords * 4;
constexpr int kOffRow = kOffLo + kCandidates * 4;
constexpr int kOffRaw = kOffRow + kCandidates * 4;  // per wave: 64 raw entries (fixed-point score << 16 | tile-local row)
constexpr int kOffCols = kOffRaw + (kThreads / 64) * 64 * 4;
constexpr int kOffIdf = kOffCols + kMaxQueryColumns * 4;
constexpr int kOffRank = kOffIdf + kMaxQueryColumns * 4;
constexpr int kOffOrder = kOffRank + kMaxQueryColumns * 4;
constexpr int kOffMass = kOffOrder + kMaxQueryColumns * 4;
constexpr int kOffSigBit = kOffMass + kMaxQueryColumns * 4;
constexpr int kOffBitIdf = kOffSigBit + kMaxQueryColumns * 4;
constexpr int kOffFixed = kOffBitIdf + kSignatureBits * 4;
constexpr int kOffTotal = kOffFixed + kMaxQueryColumns * 4;
constexpr int kOffMassTable = kOffTotal + kMaxQueryColumns * 4;
constexpr int kOffPtr = kOffMassTable + 256 * 6;  // need32[256] then mass16[256]
constexpr int kOffHist = kOffMassTable;  // the radix histogram shares the mass table: every selection is followed by
                                         // a rebuild of the table before its next use
constexpr int kOffQuadPrefix = kOffPtr + kMaxQueryColumns * (kPtrTiles + 1) * 4;  // per tile: quads before column j
constexpr int kOffQuadBase = kOffQuadPrefix + kMaxQueryColumns * 4;               // per tile: first quad of column j - prefix
constexpr int kOffCtrl = kOffQuadBase + kMaxQueryColumns * 4;
constexpr int kFastLdsBytes = kOffCtrl + 128;
static_assert((kFastLdsBytes + 256) * kWorkgroupsPerCu <= 160 * 1024, "LDS budget of one CU exceeded");
static_assert(kCandidates * 16 <= 32768 && 32768 + kCandidates * 8 <= kTile * 2, "exact-stage scratch fits the tile");
static_assert(kMaxQueryColumns == 128, "two ballots cover the query's columns");
constexpr int kKeep = (kCandidates + kThreads - 1) / kThreads;  // candidate entries a thread holds while compacting
constexpr int kSelectTrigger = kCandidates - kSelectSlack;
constexpr int kWaves = kThreads / 64;
#ifndef DS_ROUND
#define DS_ROUND 3
#endif
#ifndef DS_SCAN_BATCH
#define DS_SCAN_BATCH 1
#endif
constexpr float kFixedOne = 65000.f;  // fixed-point value of the query's total IDF mass (+ n <= 128 roundings < 2^16)
constexpr int kProbeMaxK = kThreads / 4;  // threshold bootstrap from per-thread samples: k well below the sample count

// LDS control words
enum { kLQuery = 0, kLCount, kLOverflow, kLDigit, kLRemain, kLCoef, kLPre, kLCut, kLKth0, kLKth1, kLBad, kLItems,
       kLQuads, kLNonEssential, kLMass, kLSparse, kLSigMask /* 4 words */, kLQuant = kLSigMask + 4 /* quantisation error of the query's columns, 1/65536 units */, kLEnd,
       kLStats = 24 /* 5 words: per-workgroup sums of the per-query statistics */,
       kLBytes = 30 /* 30..31: uint64, bytes this workgroup requested from global memory */ };
static_assert(kLEnd <= kLStats && kLStats + 5 <= kLBytes && kLBytes + 2 <= 32 && kLBytes % 2 == 0, "LDS control words");

// Workgroup-uniform values read from LDS or computed on the vector ALU live in VGPRs unless the compiler is told that
// they are uniform: `uniform` moves them to scalar registers (the kernel is VGPR-bound: 128 per lane at 2 WGs/CU).
__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint32_t uniform(uint32_t v) { return static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(v))); }
__device__ __forceinline__ float uniform(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
__device__ __forceinline__ bool uniform(bool v) { return __builtin_amdgcn_readfirstlane(v ? 1 : 0) != 0; }

__device__ __forceinline__ float round_down_positive(double x)
{
    float f = static_cast<float>(x);
    if (static_cast<double>(f) > x) f = __uint_as_float(__float_as_uint(f) - 1u);
    return f;
}

__device__ __forceinline__ float round_up_positive(double x)
{
    float f = static_cast<float>(x);
    if (static_cast<double>(f) < x) f = __uint_as_float(__float_as_uint(f) + 1u);
    return f;
}

// Inclusive prefix sum over the 64 lanes of a wave with DPP moves (no LDS crossbar round trips): Hillis-Steele
// inside each row of 16 lanes, then lane 15 of a row into the next row, then lane 31 into the upper half.
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t value, int)
{
    int v = static_cast<int>(value);
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);  // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);  // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);  // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);  // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3
    return static_cast<uint32_t>(v);
}

// k-th largest key of keys[0..m) (m >= k) by an 8-bit radix select.  All threads of the workgroup call it.
// passes = 4: the exact k-th largest key; passes = 2: its upper 16 bits with the lower ones cleared, i.e. a LOWER
// BOUND within 2^-7 relative of it -- all a running threshold needs, at half the barriers.
__device__ uint32_t radix_select_kth(const uint32_t *keys, int m, int k, uint32_t *hist, volatile int32_t *ctrl,
                                     int passes)
{
    const int tid = threadIdx.x, lane = tid & 63;
    uint32_t prefix = 0, mask = 0;
    int remaining = k;
    for (int shift = 24; shift >= 32 - 8 * passes; shift -= 8) {
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        for (int i = tid; i < m; i += kThreads) {
            const uint32_t key = keys[i];
            if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (tid < 64) {  // lane owns bins 255-4*lane .. 252-4*lane (descending)
            const int b0 = 255 - 4 * lane;
            const uint32_t c0 = hist[b0], c1 = hist[b0 - 1], c2 = hist[b0 - 2], c3 = hist[b0 - 3];
            const uint32_t local = c0 + c1 + c2 + c3;
            const uint32_t inclusive = wave_inclusive_scan(local, lane);
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
        prefix |= static_cast<uint32_t>(uniform(static_cast<int>(ctrl[kLDigit]))) << shift;
        mask |= 255u << shift;
        remaining = uniform(static_cast<int>(ctrl[kLRemain]));
    }
    return prefix;
}

// Everything the scan / collect phases need to turn an approximate score into a candidate.
struct Bounds {
    float coef;      // cut / (1 + cut), rounded down: row qualifies only if s + mass >= coef * (sums + maxint32)
    float pre;       // row-independent lower bound of the right-hand side (uses min(sums))
    float mass;      // upper bound of what the skipped (non-essential) columns can add to any score
    float maxint32;
};

// Inverse of encode_sums8: a lower bound of sums32[row] from the 8-bit code stored with every posting.
__device__ __forceinline__ float decode_sums8(uint32_t code)
{
    return code == 0u ? 0.f : __uint_as_float((((code >> 4) + 124u) << 23) | ((code & 0xfu) << 19));
}

// Loose test: can the row still qualify if every skipped (non-essential) column matched as well?
__device__ __forceinline__ bool may_qualify(float s, float sums, const Bounds &b)
{
    return s + b.mass >= b.coef * (sums + b.maxint32);
}

// Tight test on a complete approximate score + the radix key (float32 approximate jaccard) of a candidate.
__device__ __forceinline__ bool candidate_key(float s, float sums, const Bounds &b, uint32_t &key)
{
    if (!(s >= b.coef * (sums + b.maxint32))) return false;
    const float denominator = sums + (b.maxint32 - s);
    const float approx = s / denominator;
    key = (denominator > 0.f && approx == approx) ? __float_as_uint(approx) : 0x7f800000u;
    return true;
}

// The skipped (non-essential) columns: ranks 0..count-1 of the ascending-IDF order.  Only columns that own a
// signature bit are ever skipped, so completing a score is one 16-byte load per row.
struct Skipped {
    int count;
    uint32_t sig_mask[kSignatureWords];
};

// Adds what the skipped columns contribute to a row, from its membership signature.  Approximate (order-free) like
// the scatter itself.
__device__ __forceinline__ float complete_score(float s, uint4 signature, const Skipped &skipped, const float *bit_idf)
{
    const uint32_t words[kSignatureWords] = {signature.x, signature.y, signature.z, signature.w};
#pragma unroll
    for (int w = 0; w < kSignatureWords; ++w) {
        uint32_t bits = words[w] & skipped.sig_mask[w];
        while (bits) {
            s += bit_idf[w * 32 + __ffs(bits) - 1];
            bits &= bits - 1u;
        }
    }
    return s;
}

// Packed score tile: row r lives in half (r & 1) of word r >> 1.
__device__ __forceinline__ void add_packed(uint32_t *iscores, uint32_t local, uint32_t value)
{
    atomicAdd(&iscores[local >> 1], value << ((local & 1u) << 4));
}

// Takes a row's score and leaves zero behind (the other half of the word is untouched).
__device__ __forceinline__ uint32_t take_packed(uint32_t *iscores, uint32_t local)
{
    const uint32_t shift = (local & 1u) << 4;
    const uint32_t old = atomicAnd(&iscores[local >> 1], ~(0xffffu << shift));
    return (old >> shift) & 0xffffu;
}

// Wave-aggregated append to the candidate buffer.  Must be called by all active lanes of the wave together.
__device__ __forceinline__ void append_candidate(bool pass, uint32_t key, int32_t row, uint32_t *cand_key,
                                                 int32_t *cand_row, volatile int32_t *ctrl, int lane)
{
    const unsigned long long votes = __ballot(pass);
    if (votes == 0) return;
    const int leader = __ffsll(votes) - 1;
    int base = 0;
    if (lane == leader) base = atomicAdd(const_cast<int32_t *>(&ctrl[kLCount]), __popcll(votes));
    base = __shfl(base, leader);
    if (pass) {
        const int slot = base + __popcll(votes & ((1ull << lane) - 1ull));
        if (slot < kCandidates) {
            cand_key[slot] = key;
            cand_row[slot] = row;
        } else {
            ctrl[kLOverflow] = 1;
        }
    }
}

// Debug build (-DDS_BOUNDS_CHECK): every global access of the fast kernel whose index is data dependent is checked; the
// first violation is recorded in control[24..27] (site, index low/high, limit) and the access is skipped.
#ifdef DS_BOUNDS_CHECK
__device__ __forceinline__ bool in_bounds(int32_t *control, int site, int64_t index, int64_t limit)
{
    if (index >= 0 && index < limit) return true;
    if (atomicCAS(&control[24], 0, site) == 0) {
        control[25] = static_cast<int32_t>(index & 0xffffffff);
        control[26] = static_cast<int32_t>(index >> 32);
        control[27] = static_cast<int32_t>(limit & 0x7fffffff);
    }
    return false;
}
#define DS_OK_INDEX(site, index, limit) in_bounds(a.control, site, static_cast<int64_t>(index), static_cast<int64_t>(limit))
#else
#define DS_OK_INDEX(site, index, limit) true
#endif

// Diagnostic phase timers: thread 0 adds the shader-clock delta since its previous stamp to an LDS accumulator
// (flushed to HBM once, at the end of the kernel, so that stamping does not put a global atomic in front of a barrier).
#ifdef DS_DIAGNOSTICS
#define DS_STAMP(slot)                                                     \
    do {                                                                   \
        if (a.phase != nullptr && tid == 0) {                              \
            const unsigned long long now_ = __builtin_amdgcn_s_memtime();  \
            phase_lds[slot] += now_ - stamp_;                              \
            stamp_ = now_;                                                 \
        }                                                                  \
    } while (0)
#define DS_COUNT(slot, value)                                              \
    do {                                                                   \
        if (a.phase != nullptr && tid == 0) atomicAdd(&a.control[slot], value); \
    } while (0)
#define DS_DEBUG_BIT(bit) ((a.debug & (bit)) != 0)
#else
#define DS_STAMP(slot) do { } while (0)
#define DS_COUNT(slot, value) do { } while (0)
#define DS_DEBUG_BIT(bit) false
#endif

// kCountBytes: the instantiation that also counts the bytes it requests from global memory (ds_index_option
// "count_bytes"; bench.py runs it once, outside the timed region).  The counting costs 3 % in this register-bound
// kernel, so the production instantiation carries none of it; both do exactly the same work on the same data.
template <bool kCountBytes>
__global__ __launch_bounds__(kThreads, kWorkgroupsPerCu * kThreads / 256) void ds_jaccard_topk_kernel(JaccardArgs a)
{
#ifdef DS_DIAGNOSTICS
    unsigned long long stamp_ = a.phase != nullptr ? __builtin_amdgcn_s_memtime() : 0ull;
    __shared__ unsigned long long phase_lds[16];
    if (threadIdx.x < 16) phase_lds[threadIdx.x] = 0ull;
#endif
    extern __shared__ __align__(16) unsigned char lds[];
    uint32_t *cand_key = reinterpret_cast<uint32_t *>(lds + kOffLo);
    int32_t *cand_row = reinterpret_cast<int32_t *>(lds + kOffRow);
    int32_t *cols = reinterpret_cast<int32_t *>(lds + kOffCols);
    float *idf = reinterpret_cast<float *>(lds + kOffIdf);
    int32_t *rank = reinterpret_cast<int32_t *>(lds + kOffRank);     // position of column j in ascending-IDF order
    int32_t *order = reinterpret_cast<int32_t *>(lds + kOffOrder);   // inverse: column at position r
    float *mass_upto = reinterpret_cast<float *>(lds + kOffMass);    // [r] = IDF mass of ranks 0..r, rounded up
    int32_t *sig_bit = reinterpret_cast<int32_t *>(lds + kOffSigBit);  // signature bit of column j or -1
    float *bit_idf = reinterpret_cast<float *>(lds + kOffBitIdf);     // IDF of the query column owning signature bit g
    uint32_t *fixed = reinterpret_cast<uint32_t *>(lds + kOffFixed);  // idf[j] in the query's fixed-point scale
    uint32_t *iscores = reinterpret_cast<uint32_t *>(lds);  // score tile: 16-bit fixed-point sums, row r in half (r & 1) of word r >> 1
    uint32_t *col_total = reinterpret_cast<uint32_t *>(lds + kOffTotal);  // quads of column j over all tiles
    // need32[256] (by 8-bit sums code) and mass16[256] (by signature bits 0..7): the integer tables of the collect
    // sweep's row test, rebuilt after a selection.  need32[255] -- the code of the lists' padding entries, which take
    // whatever the trash word holds -- is beyond every score.
    uint32_t *need32 = reinterpret_cast<uint32_t *>(lds + kOffMassTable);
    uint16_t *mass16 = reinterpret_cast<uint16_t *>(lds + kOffMassTable + 256 * 4);
    uint32_t *ptr_cache = reinterpret_cast<uint32_t *>(lds + kOffPtr);  // [column][span + 1], kMaxQueryColumns * (kPtrTiles + 1) words
    uint32_t *quad_prefix = reinterpret_cast<uint32_t *>(lds + kOffQuadPrefix);
    uint32_t *quad_base = reinterpret_cast<uint32_t *>(lds + kOffQuadBase);
    uint32_t *hist = reinterpret_cast<uint32_t *>(lds + kOffHist);
    volatile int32_t *ctrl = reinterpret_cast<volatile int32_t *>(lds + kOffCtrl);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int k = a.k;
    const int64_t ptr_stride = static_cast<int64_t>(a.n_tiles) + 1;
    // The kernel arguments arrive as 16-register tuples; under register pressure the allocator spills and reloads a
    // WHOLE tuple (16 v_readlane in front of every posting load that needs one pointer of it).  The two pointers of the
    // hot loops are therefore rebuilt from scalar copies of their halves -- registers of their own, outside any tuple --
    // and keep the global address space (plain global_load, not flat_load).
    typedef const unsigned long long __attribute__((address_space(1))) *GlobalQuads;
    auto load_quad = [](GlobalQuads base, uint32_t at) {
        const unsigned long long bits = base[at];
        return make_uint2(static_cast<uint32_t>(bits), static_cast<uint32_t>(bits >> 32));
    };
    auto own_registers = [](const void *pointer) {
        const uint64_t bits = reinterpret_cast<uint64_t>(pointer);
        const uint32_t lo = uniform(static_cast<uint32_t>(bits)), hi = uniform(static_cast<uint32_t>(bits >> 32));
        return reinterpret_cast<GlobalQuads>((static_cast<uint64_t>(hi) << 32) | lo);
    };
    const GlobalQuads quads = own_registers(a.postings), sums_quads = own_registers(a.posting_sums);

    for (int i = tid * 4; i < kScoreWords; i += kThreads * 4)
        *reinterpret_cast<uint4 *>(&iscores[i]) = make_uint4(0u, 0u, 0u, 0u);
    if (tid < 5) ctrl[kLStats + tid] = 0;
    if (tid < 2) ctrl[kLBytes + tid] = 0;
    __syncthreads();
    // Global-memory bytes this workgroup REQUESTS (postings, per-posting info, sums32, signatures, list pointers, the
    // exact stage's probes): the algorithmic traffic of this kernel, reported by bench.py next to the PMC counters.
    unsigned long long *requested = reinterpret_cast<unsigned long long *>(lds + kOffCtrl + kLBytes * 4);
    // Every term is workgroup-uniform: a query's total is summed in ONE scalar register, in units of 8 bytes, and thread 0
    // adds it to its LDS word once per query (no atomics, nothing in the hot loops).  Not counted: the refinement's
    // gathers (22 B per raw entry) and the exact stage's probes, together 1-2 % of the total.
    uint32_t query_units = 0;
    auto count_units = [&](uint32_t units) { if constexpr (kCountBytes) query_units += units; };

    for (;;) {
        if (tid == 0) {
            ctrl[kLQuery] = atomicAdd(&a.control[kCtlQueue], 1);
            ctrl[kLCount] = 0;
            ctrl[kLOverflow] = 0;
            ctrl[kLBad] = 0;
            ctrl[kLQuant] = 0;
        }
        __syncthreads();
        const int64_t q = uniform(static_cast<int>(ctrl[kLQuery]));
        if (q >= a.n_queries) break;  // exit condition reached by every wave: the queue head only grows

        const int64_t qbase = a.q_rowptr[q];
        const int64_t n64 = a.q_rowptr[q + 1] - qbase;
        const double maxint = a.q_maxint[q];
        bool slow = a.literal_only != 0 || n64 > kMaxQueryColumns || n64 < 0 || !(maxint > 0.0) || !(maxint < 1e30);
        int reason = slow ? 0 : -1;  // why the query is handed to the dense kernel (diagnostics)
        const int n = slow ? 0 : static_cast<int>(n64);
        if (tid < n) {
            const int32_t column = a.q_cols[qbase + tid];
            const bool bad = column < 0 || column >= a.n_columns;
            if (bad) ctrl[kLBad] = 1;
            cols[tid] = bad ? 0 : column;
            const float value = bad ? 0.f : a.idf32[column];
            const int bit = bad ? -1 : static_cast<int>(a.sig_column[column]);
            idf[tid] = value;
            col_total[tid] = bad ? 0u
                                 : a.col_ptr[static_cast<int64_t>(column) * ptr_stride + a.n_tiles] -
                                       a.col_ptr[static_cast<int64_t>(column) * ptr_stride];
            sig_bit[tid] = bit;
            if (bit >= 0) bit_idf[bit] = value;
        }
        if constexpr (kCountBytes) query_units = static_cast<uint32_t>(17 * n + 24 + 7) >> 3;
        __syncthreads();
        if (uniform(static_cast<int>(ctrl[kLBad]))) {
            if (tid == 0) {
                a.status[q] = kQueryErrorArg;
                atomicAdd(&a.control[kCtlErrors], 1);
            }
            for (int j = tid; j < k; j += kThreads) a.out_rows[q * k + j] = -1;
            __syncthreads();
            continue;
        }
        if (tid >= 64 && tid < kMaxQueryColumns) quad_prefix[tid] = 0xffffffffu;  // columns 64..127: see the item map
        // ascending-IDF order of the query's columns and the IDF mass of every prefix of that order
        if (tid < n) {
            const float mine = idf[tid];
            int r = 0;
            double below = 0.0;
#pragma unroll 8
            for (int i = 0; i < n; ++i) {  // unrolled: the LDS reads of eight steps are in flight together
                const float other = idf[i];
                const bool first = (other < mine) | ((other == mine) & (i <= tid));
                r += first & (i != tid);
                below += first ? static_cast<double>(other) : 0.0;
            }
            rank[tid] = r;
            order[r] = tid;
            mass_upto[r] = round_up_positive(below * (1.0 + 3.814697265625e-06));
        }

        const float maxint32 = static_cast<float>(maxint);
        __syncthreads();
        // Scores are accumulated in unsigned 16-bit fixed point, two rows per LDS word (LDS integer atomics run ~14x
        // faster than ds_add_f32 on gfx950 and are order-independent): one unit = total/kFixedOne where total >=
        // every reachable score.  A row's sum is at most kFixedOne + n (every term rounds by < 1 unit) < 65536, so
        // the low half of a word never carries into the high half.
        const float total_mass = uniform(n > 0 ? fmaxf(maxint32, mass_upto[n - 1]) : maxint32);
        // the error margin below assumes maxint >= the idf total of the query's columns, as match_maker.py:197 computes
        // it; a caller-supplied smaller value (C ABI) sends the query to the literal kernel
        if (!slow && n > 0 && uniform(maxint32 < mass_upto[n - 1] * 0.999f)) { slow = true; reason = 0; }
        const float to_fixed = uniform(kFixedOne / total_mass), from_fixed = uniform(total_mass * (1.f / kFixedOne));
        if (tid < kMaxQueryColumns) {
            // fixed-point image of every column's IDF and the exact total of what the quantisation can be off by
            uint32_t off = 0;
            if (tid < n) {
                const uint32_t f = max(1u, static_cast<uint32_t>(idf[tid] * to_fixed + 0.5f));
                fixed[tid] = f;
                const double unit = static_cast<double>(from_fixed);
                off = static_cast<uint32_t>(fabs(static_cast<double>(idf[tid]) - f * unit) / unit * 65536.0) + 1u;
            }
            const uint32_t total_off = wave_inclusive_scan(off, lane);
            if (lane == 63) atomicAdd(const_cast<int32_t *>(&ctrl[kLQuant]), static_cast<int32_t>(total_off));
        }
        // |approximate jaccard - exact jaccard| <= margin (DESIGN.md "error margin of the prefilter"):
        // float32 evaluation + quantisation of n terms to one unit each (4 = bound of d jaccard / d score * total)
        // float32 evaluation + the measured quantisation error of this query's columns (<= 1 unit each; typically
        // a quarter of a unit), both relative to the total mass; 4 = bound of d jaccard / d score * total
        auto error_margin = [&]() {
            const double off_units = static_cast<double>(uniform(static_cast<int>(ctrl[kLQuant]))) * (1.0 / 65536.0);
            return (6.0 * n + 64.0) * 5.9604644775390625e-08 + off_units * (4.0 / kFixedOne) * 1.001;
        };
        Bounds bounds{0.f, FLT_MIN, 0.f, maxint32};
        float cut = 0.f, pending_mass = 0.f;  // mass of the columns skipped from the NEXT tile on
        uint32_t pending_sig_mask[kSignatureWords] = {0u, 0u, 0u, 0u};
        Skipped skipped{0, {0u, 0u, 0u, 0u}};
        bool sparse_mode = false;  // decided at every selection from the essential columns' remaining length
        bool rebuild_mass_table = false;
        int non_essential = 0;
        bool tight = false;
        int next_select = max(4 * k, a.select_min);
        if (next_select > kSelectTrigger) next_select = kSelectTrigger;
        int selects = 0, sparse_tiles = 0, dense_tiles = 0, last_appended = 0;
        DS_STAMP(0);

        // Rows passing the register-level tests of a sweep are RAW entries (approximate essential score, row).  A wave
        // parks them in its private LDS buffer and refines them itself, 64 at a time and always before it leaves the
        // tile: exact sums, completion of the skipped columns from the signature, tight test -- one lane per entry,
        // no workgroup barrier; only the survivors reach the shared candidate buffer.  Every raw entry is therefore
        // refined under the set of skipped columns it was scored with.
        uint32_t *wave_raw = reinterpret_cast<uint32_t *>(lds + kOffRaw) + wave * 64;
        int64_t raw_tile_base = 0;  // the tile the parked entries belong to (they never outlive it)
        int raw_count = 0;  // wave-uniform
        auto flush_raw = [&]() {
            if (raw_count == 0) return;
            DS_COUNT(kCtlRefines, 1);
            const bool mine = lane < raw_count;
            bool ok = false;
            uint32_t key = 0;
            int32_t t = -1;
            if (mine) {
                const uint32_t entry = wave_raw[lane];
                const float raw = static_cast<float>(entry >> 16) * from_fixed;
                t = static_cast<int32_t>(raw_tile_base + (entry & 0xffffu));
                if (DS_OK_INDEX(1, t, a.n_truth)) {
                    const float sums = a.sums32[t];
                    const uint4 signature = skipped.count > 0 ? a.signature[t] : make_uint4(0u, 0u, 0u, 0u);
                    // a row with k or more twins (same column set, same sums32) of larger index can never be among the
                    // k largest row indexes of match_maker.py:71, and the k-th largest value does not need it either
                    const bool shadowed = static_cast<int>(a.dup_rank[t]) >= k;
                    // a zero score reaches this point only through the collect sweep's rounding slack (a second
                    // posting of a row whose score another lane took): never a candidate
                    if (!shadowed && raw > 0.f && may_qualify(raw, sums, bounds)) {
                        const float full = complete_score(raw, signature, skipped, bit_idf);
                        ok = candidate_key(full, sums, bounds, key);
                    }
                }
            }
            raw_count = 0;
            append_candidate(ok, key, t, cand_key, cand_row, ctrl, lane);
        };
        auto append_raw = [&](bool pass, uint32_t fixed_score, uint32_t local_row) {
            const unsigned long long votes = __ballot(pass);
            if (votes == 0) return;
            const int count = __popcll(votes);
            if (raw_count + count > 64) flush_raw();
            if (pass) wave_raw[raw_count + __popcll(votes & ((1ull << lane) - 1ull))] = (fixed_score << 16) | local_row;
            raw_count += count;
        };

        int sparse_retries = 0;
        // The list pointers of the query's columns are cached in LDS for `span` tiles at a time: as many as the cache
        // holds for this query's number of columns (all 18 tiles of C2 for a query of up to 26 columns), so that the
        // reload -- an exposed global latency between two barriers -- happens rarely.
        const int span = n > 0 ? min(a.n_tiles, max(kPtrTiles, kMaxQueryColumns * (kPtrTiles + 1) / n - 1)) : kPtrTiles;
        int block_start = 0, block_end = 0;
        for (int b = 0; b < a.n_tiles && !slow; ++b) {
            bool redo_tile = false;
            const float tile_min = a.tile_sums_min[b];  // scalar load, consumed after the item map is built
            if (b >= block_end) {  // list pointers of the next `span` tiles: one coalesced burst
                block_start = b;
                block_end = b + span;
                __syncthreads();
                const int width = min(span, a.n_tiles - b) + 1;
                for (int e = tid; e < n * (span + 1); e += kThreads) {
                    const int j = e / (span + 1), i = e - j * (span + 1);
                    ptr_cache[e] = (i < width && DS_OK_INDEX(2, static_cast<int64_t>(cols[j]) * ptr_stride + b + i,
                                                            a.n_columns * ptr_stride))
                                       ? a.col_ptr[static_cast<int64_t>(cols[j]) * ptr_stride + b + i]
                                       : 0u;
                }
                count_units(static_cast<uint32_t>(n * width + 1) >> 1);
                __syncthreads();
            }
            const int bt = b - block_start;
            bounds.mass = pending_mass;  // what this tile's scores do NOT contain; fixed until the tile is done
            skipped.count = non_essential;
            for (int w = 0; w < kSignatureWords; ++w) skipped.sig_mask[w] = pending_sig_mask[w];
            if (rebuild_mass_table) {  // used by the exchange sweep, i.e. after the barrier that ends the scatter
                rebuild_mass_table = false;
                if (tid < 256) {
                    const uint32_t skipped8 = skipped.sig_mask[0] & 0xffu;
                    double rest = static_cast<double>(bounds.mass), some = 0.0;
                    for (int g = 0; g < 8; ++g) {
                        if (!((skipped8 >> g) & 1u)) continue;
                        rest -= static_cast<double>(bit_idf[g]);
                        if ((tid >> g) & 1) some += static_cast<double>(bit_idf[g]);
                    }
                    if (rest < 0.0) rest = 0.0;
                    // rounded up, and never above the full mass (which already carries its own slack)
                    const float value = round_up_positive((rest + some) * (1.0 + 3.814697265625e-06) + 1e-30);
                    const float mass = value < bounds.mass ? value : bounds.mass;
                    // The collect sweep tests rows in the integer domain of the fixed-point scores:
                    //   score + mass16[signature bits 0..7] >= max(tile gate, need32[8-bit sums code])
                    // mass16 is rounded up and need32 down by more than the float test's own roundings, so the rows
                    // that pass are a superset of those of `passes` (no conversions, no float decoding per posting).
                    const double unit = static_cast<double>(from_fixed);
                    const double mass_units = static_cast<double>(mass) / unit + 2.0;
                    mass16[tid] = static_cast<uint16_t>(mass_units < 65535.0 ? mass_units : 65535.0);
                    const double need_units = static_cast<double>(bounds.coef) *
                                                  (static_cast<double>(decode_sums8(tid)) + static_cast<double>(maxint32)) /
                                                  unit - 2.0;
                    need32[tid] = tid == 255 ? 0x7fffffffu
                                             : static_cast<uint32_t>(need_units < 0.0 ? 0.0 : (need_units < 65535.0 ? need_units : 65535.0));
                }
            }
            const bool sparse = tight && sparse_mode;
            // ---- work items of this tile.  The essential columns' posting quads of the tile form ONE sequence (column after
            // column); an item is 64 consecutive quads of it, so every lane of an item has a quad to work on whatever
            // the lengths of the single lists (items of one list each were 76 % full on C2).  Every wave computes the
            // same exclusive prefix of the lists' quad counts (lane j: columns j and j + 64, DPP scan) and writes it to
            // LDS -- all eight waves write the same values, a wave only ever reads what its own lanes wrote -- and a
            // lane finds the column of its quad by a branch-free binary search there: no ballots, no v_readlane, no
            // scalar instructions per item.
            uint32_t total_quads;
            {
                uint32_t quads_of[2] = {0u, 0u}, begin_of[2] = {0u, 0u};
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int j = lane + 64 * h;
                    if (j < n && rank[j] >= non_essential) {
                        begin_of[h] = ptr_cache[j * (span + 1) + bt];
                        quads_of[h] = ptr_cache[j * (span + 1) + bt + 1] - begin_of[h];
                    }
                }
                const uint32_t scan0 = wave_inclusive_scan(quads_of[0], lane);
                total_quads = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(scan0), 63));
                const uint32_t before0 = scan0 - quads_of[0];
                quad_prefix[lane] = lane < n ? before0 : 0xffffffffu;  // columns beyond the query's: never found
                quad_base[lane] = begin_of[0] - before0;
                if (n > 64) {
                    const uint32_t scan1 = wave_inclusive_scan(quads_of[1], lane);
                    const uint32_t before1 = total_quads + scan1 - quads_of[1];
                    total_quads += static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(scan1), 63));
                    quad_prefix[lane + 64] = lane + 64 < n ? before1 : 0xffffffffu;
                    quad_base[lane + 64] = begin_of[1] - before1;
                }
            }
            const int n_items = static_cast<int>((total_quads + 63u) >> 6);
            // (this lane's quad, fixed-point IDF of its column) of work item `at`; false beyond the last quad
            auto locate = [&](int at, uint32_t &first, uint32_t &value) {
                const uint32_t g = static_cast<uint32_t>(at) * 64u + lane;
                uint32_t c = 0;
#pragma unroll
                for (uint32_t bit = kMaxQueryColumns / 2; bit > 0; bit >>= 1) {  // largest c with quad_prefix[c] <= g
                    const uint32_t t = c + bit;
                    c = quad_prefix[t] <= g ? t : c;
                }
                first = quad_base[c] + g;
                value = fixed[c];
                return g < total_quads;
            };
            if constexpr (kCountBytes) {  // posting bytes of this tile: 8 per quad and sweep, 8 more for the collect sweep's row info
                count_units(total_quads * (sparse ? (n_items <= DS_ROUND * kWaves ? 2u : 3u) : 1u));
            }
            DS_STAMP(1);

            const int64_t tile_base = static_cast<int64_t>(b) * kTile;
            // read here, where a barrier (end of the scatter) separates every thread's read from the next append
            int count_at_step = uniform(static_cast<int>(ctrl[kLCount]));
            // row-independent gate of this tile: coef * (min sums of the tile + maxint), rounded down
            Bounds here = bounds;
            {
                const float gate = uniform(bounds.coef * (tile_min + maxint32) * (1.f - 3.814697265625e-06f));
                if (gate > here.pre) here.pre = gate;
            }
            // A sweep only runs the register-level tests and appends RAW entries (approximate essential score, row);
            // the wave refines them (flush_raw) before it leaves the tile.
            // branch-free on purpose: short-circuit evaluation would put every LDS read of the caller behind its own
            // exec-masked branch and wait for each one separately
            auto passes = [&](float s, float sums_lower_bound, float row_mass) {
                return (s > 0.f) & (s + row_mass >= here.pre) &
                       (s + row_mass >= here.coef * (sums_lower_bound + here.maxint32));
            };
            // four rows at a time: one ballot decides whether anything needs appending (the common case: nothing)
            auto consider4 = [&](const uint32_t (&fixed_score)[4], const uint32_t (&local)[4],
                                 const float (&sums_lower_bound)[4], const float (&row_mass)[4]) {
                bool pass[4];
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    pass[e] = passes(static_cast<float>(fixed_score[e]) * from_fixed, sums_lower_bound[e], row_mass[e]);
                if (__ballot(pass[0] || pass[1] || pass[2] || pass[3]) == 0) return;
#pragma unroll
                for (int e = 0; e < 4; ++e) append_raw(pass[e], fixed_score[e], local[e]);
            };
            raw_tile_base = tile_base;

            // ---- (1) scatter: fixed-point LDS atomics; padding entries hit the trash word.  Wave w takes the items
            // w, w + kWaves, ...: four of them (one quad per lane each) are in flight at a time.  On a sparse tile whose
            // items fit one round the quads (and their per-posting info) stay in registers for the collect sweep.
            constexpr int kRound = DS_ROUND;
            const bool single_round = sparse && n_items <= kRound * kWaves;
            uint2 quad[kRound], quad_info[kRound];
            bool live[kRound];  // a wave without items still runs the collect sweep
#pragma unroll
            for (int u = 0; u < kRound; ++u) live[u] = false;
            const int count_before = count_at_step;
            for (int round = (sparse && DS_DEBUG_BIT(1)) ? n_items : 0; wave + round * kWaves < n_items; round += kRound) {
                uint32_t value[kRound];
#pragma unroll
                for (int u = 0; u < kRound; ++u) {
                    const int at = wave + (round + u) * kWaves;
                    uint32_t first = 0;
                    value[u] = 0;
                    live[u] = at < n_items && locate(at, first, value[u]) && DS_OK_INDEX(3, first, a.n_quads);
                    quad[u] = live[u] ? load_quad(quads, first) : make_uint2(kSentinel * 0x10001u, kSentinel * 0x10001u);
                    if (single_round)  // idle lanes carry the padding code 255: their (zero) scores never pass the collect sweep's test
                        quad_info[u] = live[u] ? load_quad(sums_quads, first) : make_uint2(0xff00ff00u, 0xff00ff00u);
                }
#pragma unroll
                for (int u = 0; u < kRound; ++u) {
                    // exec-masked on purpose: unconditional atomics of idle lanes on trash words measured slower
                    // (30.3 against 29.8 ms, profiles/r02_tuning.txt)
                    if (!live[u] || (sparse && DS_DEBUG_BIT(64))) continue;
                    add_packed(iscores, quad[u].x & 0xffffu, value[u]);
                    add_packed(iscores, quad[u].x >> 16, value[u]);
                    add_packed(iscores, quad[u].y & 0xffffu, value[u]);
                    add_packed(iscores, quad[u].y >> 16, value[u]);
                }
            }
            __syncthreads();
            DS_STAMP(sparse ? 6 : 2);

            if (sparse) {
                // ---- (2s) collect sweep over the same postings: the first lane to reach a row takes its score and
                // leaves zero behind (no scan of the whole tile)
                ++sparse_tiles;
                // Cheap integer pre-test on the taken fixed-point scores: a row can only pass `s + mass >= pre` if its
                // score reaches (pre - mass) in fixed-point units (rounded down, minus slack: a superset of the float
                // test).  Most quads have no such row and skip the per-row decoding and table lookups altogether.
                // A row whose score is zero here (another posting of the row took it first) has nothing but its mass16:
                // the gate is kept at or above the largest mass16 so that such a row fails `have >= need` without a
                // test of its own.  (It can still pass when it owns every skipped column AND the gate had to be
                // raised, i.e. pre - mass is within the tables' 4 units of rounding slack; the refinement rejects it.)
                const uint32_t gate_fixed = [&]() {  // the tile's row-independent bound `here.pre`, rounded down
                    const double units = static_cast<double>(here.pre) / static_cast<double>(from_fixed) - 2.0;
                    const uint32_t gate = units > 0.0 ? static_cast<uint32_t>(units) : 0u;
                    return max(gate, static_cast<uint32_t>(mass16[255]));
                }();
                auto collect = [&]() {
#pragma unroll
                    for (int u = 0; u < kRound; ++u) {
                        if (__ballot(live[u]) == 0) continue;
                        const uint32_t local[4] = {quad[u].x & 0xffffu, quad[u].x >> 16, quad[u].y & 0xffffu, quad[u].y >> 16};
                        const uint32_t info[4] = {quad_info[u].x & 0xffffu, quad_info[u].x >> 16, quad_info[u].y & 0xffffu,
                                                  quad_info[u].y >> 16};
                        uint32_t taken[4];
                        // the four atomics of a quad are issued back to back and their results extracted afterwards
                        // (extraction inside the same branch would wait for every atomic separately)
                        uint32_t before[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) before[e] = 0u;
                        if (live[u] && !DS_DEBUG_BIT(16)) {
                            // ONE exec-masked region per quad (idle lanes stay off the LDS); the padding entries of a
                            // list's last quad take from the trash word instead of being branched around one by one
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                before[e] = atomicAnd(&iscores[local[e] >> 1], ~(0xffffu << ((local[e] & 1u) << 4)));
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) taken[e] = (before[e] >> ((local[e] & 1u) << 4)) & 0xffffu;
                        if (DS_DEBUG_BIT(2)) continue;
                        bool pass[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const uint32_t have = taken[e] + mass16[info[e] & 0xffu];
                            const uint32_t need = max(gate_fixed, need32[info[e] >> 8]);
                            pass[e] = have >= need;  // padding entries carry code 255: never; zero scores: see the gate
                        }
#ifdef DS_DIAGNOSTICS
                        if (a.phase != nullptr) {  // selectivity of a row-independent gate (tuning experiment)
                            int touched = 0, level1 = 0;
                            for (int e = 0; e < 4; ++e) {
                                touched += taken[e] != 0u;
                                level1 += taken[e] != 0u && taken[e] + mass16[255] >= gate_fixed;
                            }
                            for (int d = 32; d > 0; d >>= 1) { touched += __shfl_xor(touched, d); level1 += __shfl_xor(level1, d); }
                            if (lane == 0) { atomicAdd(&a.control[20], touched); atomicAdd(&a.control[21], level1);
                                             atomicAdd(&a.control[22], 1); atomicAdd(&a.control[23], level1 != 0); }
                        }
#endif
                        if (__ballot(pass[0] | pass[1] | pass[2] | pass[3]) == 0) continue;
#pragma unroll
                        for (int e = 0; e < 4; ++e) append_raw(pass[e], taken[e], local[e]);
                    }
                };
                if (single_round) {
                    collect();
                } else {
                    for (int round = DS_DEBUG_BIT(1) ? n_items : 0; wave + round * kWaves < n_items; round += kRound) {
#pragma unroll
                        for (int u = 0; u < kRound; ++u) {
                            const int at = wave + (round + u) * kWaves;
                            uint32_t first = 0, value = 0;
                            live[u] = at < n_items && locate(at, first, value) && DS_OK_INDEX(3, first, a.n_quads);
                            quad[u] = live[u] ? load_quad(quads, first) : make_uint2(kSentinel * 0x10001u, kSentinel * 0x10001u);
                            quad_info[u] = live[u] ? load_quad(sums_quads, first) : make_uint2(0xff00ff00u, 0xff00ff00u);
                        }
                        collect();
                    }
                }
                flush_raw();
                __syncthreads();
                DS_STAMP(7);
                if (uniform(static_cast<int>(ctrl[kLOverflow]))) {
                    // The sweep ran to its end, so the tile is all zero again, but some of its rows did not fit the
                    // buffer: tighten the threshold with what is buffered, drop this tile's entries and process the
                    // tile once more.
                    if (++sparse_retries > 3) { slow = true; reason = 2; break; }
                    __syncthreads();
                    if (tid == 0) {
                        ctrl[kLCount] = kCandidates;
                        ctrl[kLOverflow] = 0;
                    }
                    __syncthreads();
                    redo_tile = true;
                } else {
                    sparse_retries = 0;
                    DS_COUNT(kCtlRawSparse, ctrl[kLCount] - count_before);
                }
            }

            // ---- (2d) dense scan (and re-zero) of the tile; in steps while no running value exists
            const int64_t rows_left = a.n_truth - tile_base;
            const int limit = rows_left >= kTile ? kTile : static_cast<int>((rows_left + 7) & ~int64_t(7));
            int r0 = sparse ? limit : 0;
            if (!sparse) ++dense_tiles;
            bool select_now = sparse && (redo_tile || uniform(static_cast<int>(ctrl[kLCount])) >= next_select);
            int retries = 0;
            bool probed = false;
            // rows to drop at the next pruning (rows that are scanned / collected again)
            int32_t drop_lo = redo_tile ? static_cast<int32_t>(tile_base) : 0;
            int32_t drop_hi = redo_tile ? static_cast<int32_t>(tile_base + kTile) : 0;
            bool force_pending = redo_tile;
            while (r0 < limit || select_now) {
                bool force_select = force_pending;
                force_pending = false;
                if (r0 < limit && !tight && !probed && k <= kProbeMaxK && count_at_step == 0) {
                    // ---- threshold bootstrap.  No threshold yet and nothing buffered: instead of flooding the
                    // candidate buffer with every positive row, each thread picks the best of its 32 rows of this
                    // tile (cross-multiplied comparison, no division) and contributes that row's key as a SAMPLE
                    // (row = -1: dropped by the compaction below).  The k-th largest of the kThreads samples is the
                    // k-th largest value of a subset of the rows, i.e. a valid lower estimate of the final k-th.
                    probed = true;
                    float best_s = 0.f, best_d = 1.f, best_sums = 0.f;
#pragma unroll 2
                    for (int idx = tid * 8; idx < limit; idx += kThreads * 8) {  // two steps' loads in flight
                        const float4 sums_lo = *reinterpret_cast<const float4 *>(&a.sums32[tile_base + idx]);
                        const float4 sums_hi = *reinterpret_cast<const float4 *>(&a.sums32[tile_base + idx + 4]);
                        const uint4 raw4 = *reinterpret_cast<uint4 *>(&iscores[idx >> 1]);
                        const uint32_t words[4] = {raw4.x, raw4.y, raw4.z, raw4.w};
                        const float su[8] = {sums_lo.x, sums_lo.y, sums_lo.z, sums_lo.w,
                                             sums_hi.x, sums_hi.y, sums_hi.z, sums_hi.w};
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float sv = static_cast<float>((words[e >> 1] >> ((e & 1) * 16)) & 0xffffu) * from_fixed;
                            const float d = su[e] + maxint32;
                            if (sv * best_d > best_s * d) {
                                best_s = sv;
                                best_d = d;
                                best_sums = su[e];
                            }
                        }
                    }
                    count_units(static_cast<uint32_t>(limit) >> 1);
                    uint32_t sample = 0u;
                    if (best_s > 0.f && !candidate_key(best_s, best_sums, bounds, sample)) sample = 0u;
                    cand_key[tid] = sample;
                    cand_row[tid] = -1;
                    if (tid == 0) ctrl[kLCount] = kThreads;
                    __syncthreads();
                    force_select = true;
                    DS_STAMP(11);
                } else if (r0 < limit) {
                    // no running value yet: 512-row steps; afterwards the whole rest of the tile.  The scores of a step
                    // are zeroed only once the step has fitted into the candidate buffer: if a flood of rows above a
                    // still weak threshold overflows it, the step is undone, the threshold is tightened from what is
                    // already buffered, and the same rows are scanned again.
                    const int r1 = limit;
                    // a flood is only to be expected while there is no threshold, while it is young, or when the last
                    // step appended a lot; otherwise zero on the fly (one LDS pass and one barrier less) and treat an
                    // overflow as fatal.  Without a threshold every positive row is appended: a tile with more than
                    // a buffer's worth of them overflows once, which yields a threshold from ~1800 samples.
                    const bool recoverable = !tight || selects <= 2 || last_appended > 128;
                    // A thread reads eight rows (one uint4 of packed scores) per iteration; the `sums32` loads of a
                    // batch of iterations are issued together, ahead of the LDS work, so a batch exposes one HBM latency.
                    constexpr int kBatch = DS_SCAN_BATCH;
                    // the trip count is workgroup-uniform (a wave keeps all its lanes through the loop: the raw-entry
                    // buffer is addressed by lane); rows beyond the tile's end are masked by `valid`
                    const int scan_steps = (DS_DEBUG_BIT(8) && b > 0) ? 0 : (r1 - r0 + kBatch * kThreads * 8 - 1) / (kBatch * kThreads * 8);
                    for (int step = 0; step < scan_steps; ++step) {
                        const int base = r0 + tid * 8 + step * kBatch * kThreads * 8;
                        float4 sums_lo[kBatch], sums_hi[kBatch];
#pragma unroll
                        for (int it = 0; it < kBatch; ++it) {
                            const int idx = base + it * kThreads * 8;
                            const bool ok = idx < r1 && DS_OK_INDEX(5, tile_base + idx + 7, a.n_truth + 8);
                            sums_lo[it] = ok ? *reinterpret_cast<const float4 *>(&a.sums32[tile_base + idx])
                                             : make_float4(0.f, 0.f, 0.f, 0.f);
                            sums_hi[it] = ok ? *reinterpret_cast<const float4 *>(&a.sums32[tile_base + idx + 4])
                                             : make_float4(0.f, 0.f, 0.f, 0.f);
                        }
#pragma unroll
                        for (int it = 0; it < kBatch; ++it) {
                            const int idx = base + it * kThreads * 8;
                            const bool valid = idx < r1;
                            const uint4 raw4 = valid ? *reinterpret_cast<uint4 *>(&iscores[idx >> 1]) : make_uint4(0u, 0u, 0u, 0u);
                            if (valid && !recoverable) *reinterpret_cast<uint4 *>(&iscores[idx >> 1]) = make_uint4(0u, 0u, 0u, 0u);
                            const uint32_t words[4] = {raw4.x, raw4.y, raw4.z, raw4.w};
                            uint32_t fx[8];
                            bool any = false;
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                fx[e] = (words[e >> 1] >> ((e & 1) * 16)) & 0xffffu;
                                // mass < pre by construction, so untouched rows (score 0) never pass
                                any = any || static_cast<float>(fx[e]) * from_fixed + here.mass >= here.pre;
                            }
                            if (__ballot(any) == 0) continue;
                            if (DS_DEBUG_BIT(4) && b > 0) continue;
                            const float mass4[4] = {here.mass, here.mass, here.mass, here.mass};
                            {
                                const uint32_t s4[4] = {fx[0], fx[1], fx[2], fx[3]};
                                const uint32_t rows4[4] = {static_cast<uint32_t>(idx), static_cast<uint32_t>(idx + 1),
                                                           static_cast<uint32_t>(idx + 2), static_cast<uint32_t>(idx + 3)};
                                const float bound4[4] = {sums_lo[it].x, sums_lo[it].y, sums_lo[it].z, sums_lo[it].w};
                                consider4(s4, rows4, bound4, mass4);
                            }
                            {
                                const uint32_t s4[4] = {fx[4], fx[5], fx[6], fx[7]};
                                const uint32_t rows4[4] = {static_cast<uint32_t>(idx + 4), static_cast<uint32_t>(idx + 5),
                                                           static_cast<uint32_t>(idx + 6), static_cast<uint32_t>(idx + 7)};
                                const float bound4[4] = {sums_hi[it].x, sums_hi[it].y, sums_hi[it].z, sums_hi[it].w};
                                consider4(s4, rows4, bound4, mass4);
                            }
                        }
                    }
                    flush_raw();
                    count_units(static_cast<uint32_t>(r1 - r0) >> 1);
                    __syncthreads();
                    DS_STAMP(3);
                    last_appended = uniform(static_cast<int>(ctrl[kLCount])) - count_at_step;
                    if (uniform(static_cast<int>(ctrl[kLOverflow]))) {
                        // the buffer holds as many of the step's rows as fitted: tighten the threshold with them,
                        // drop the step's rows from the buffer and scan the same rows again
                        if (!recoverable || ++retries > 4) { slow = true; reason = 3; break; }
                        __syncthreads();
                        if (tid == 0) {
                            ctrl[kLCount] = kCandidates;
                            ctrl[kLOverflow] = 0;
                        }
                        __syncthreads();
                        force_select = true;
                        drop_lo = static_cast<int32_t>(tile_base + r0);
                        drop_hi = static_cast<int32_t>(tile_base + r1);
                    } else {
                        if (recoverable) {
                            for (int idx = r0 + tid * 8; idx < r1; idx += kThreads * 8)
                                *reinterpret_cast<uint4 *>(&iscores[idx >> 1]) = make_uint4(0u, 0u, 0u, 0u);
                            __syncthreads();
                            DS_STAMP(9);
                        }
                        r0 = r1;
                        retries = 0;
                    }
                    DS_STAMP(12);
                }
                select_now = false;
                const int m = uniform(static_cast<int>(ctrl[kLCount]));
                if ((m < next_select && !force_select) || m < k) {
                    if (force_select) { slow = true; reason = 3; break; }  // nothing to tighten with
                    continue;
                }
                // ---- tighten: tau = k-th largest lower estimate seen so far; keep what can still qualify
                const uint32_t tau_key = radix_select_kth(cand_key, m, k, hist, ctrl, 2);
                ++selects;
                if (wave == 0) {
                    const double tau = static_cast<double>(__uint_as_float(tau_key));
                    const double cut_value = tau - 2.0 * error_margin() - 2e-6;
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
                    // non-essential columns: the longest ascending-IDF prefix whose total mass stays below `pre`
                    // and whose columns all own a signature bit (completion is then a single load per row)
                    int skip_count = 0;
                    {
                        const bool ok0 = lane < n && mass_upto[lane] < new_pre && sig_bit[order[lane]] >= 0;
                        const bool ok1 = lane + 64 < n && mass_upto[lane + 64] < new_pre && sig_bit[order[lane + 64]] >= 0;
                        const unsigned long long v0 = __ballot(ok0), v1 = __ballot(ok1);
                        skip_count = ~v0 ? __ffsll(static_cast<long long>(~v0)) - 1
                                         : 64 + (~v1 ? __ffsll(static_cast<long long>(~v1)) - 1 : 64);
                    }
                    uint32_t mask_bits[kSignatureWords] = {0u, 0u, 0u, 0u};
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int r = lane + 64 * h;
                        if (r < skip_count) {
                            const int bit = sig_bit[order[r]];
#pragma unroll
                            for (int w = 0; w < kSignatureWords; ++w)
                                if ((bit >> 5) == w) mask_bits[w] |= 1u << (bit & 31);
                        }
                    }
#pragma unroll
                    for (int w = 0; w < kSignatureWords; ++w)
                        for (int d = 32; d > 0; d >>= 1) mask_bits[w] |= __shfl_xor(mask_bits[w], d);
                    uint32_t essential_quads = 0;
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int j = lane + 64 * h;
                        if (j < n && rank[j] >= skip_count) essential_quads += col_total[j];
                    }
                    for (int d = 32; d > 0; d >>= 1) essential_quads += __shfl_xor(essential_quads, d);
                    if (lane == 0) {
                        ctrl[kLSparse] = essential_quads / static_cast<uint32_t>(a.n_tiles) <=
                                         static_cast<uint32_t>(a.sparse_quads);
                        for (int w = 0; w < kSignatureWords; ++w) ctrl[kLSigMask + w] = static_cast<int32_t>(mask_bits[w]);
                        ctrl[kLCoef] = __float_as_int(new_coef);
                        ctrl[kLPre] = __float_as_int(new_pre);
                        ctrl[kLCut] = __float_as_int(new_cut);
                        ctrl[kLNonEssential] = skip_count;
                        ctrl[kLMass] = __float_as_int(skip_count > 0 ? mass_upto[skip_count - 1] : 0.f);
                    }
                }
                __syncthreads();
                bounds.coef = uniform(__int_as_float(ctrl[kLCoef]));
                bounds.pre = uniform(__int_as_float(ctrl[kLPre]));
                here.coef = bounds.coef;
                {
                    const float gate = uniform(bounds.coef * (tile_min + maxint32) * (1.f - 3.814697265625e-06f));
                    here.pre = gate > bounds.pre ? gate : bounds.pre;
                }
                pending_mass = uniform(__int_as_float(ctrl[kLMass]));
                rebuild_mass_table = true;
                for (int w = 0; w < kSignatureWords; ++w) pending_sig_mask[w] = uniform(static_cast<uint32_t>(ctrl[kLSigMask + w]));
                cut = uniform(__int_as_float(ctrl[kLCut]));
                non_essential = uniform(static_cast<int>(ctrl[kLNonEssential]));
                sparse_mode = uniform(static_cast<int>(ctrl[kLSparse])) != 0;
                tight = true;
                // in-place compaction: read everything, barrier, rewrite the survivors
                uint32_t keep_key[kKeep];
                int32_t keep_row[kKeep];
#pragma unroll
                for (int r = 0; r < kKeep; ++r) {
                    const int i = tid + r * kThreads;
                    keep_key[r] = i < m ? cand_key[i] : 0u;
                    keep_row[r] = i < m ? cand_row[i] : -1;
                }
                __syncthreads();
                if (tid == 0) ctrl[kLCount] = 0;
                __syncthreads();
#pragma unroll
                for (int r = 0; r < kKeep; ++r) {
                    if (keep_row[r] >= 0 && __uint_as_float(keep_key[r]) >= cut &&
                        !(keep_row[r] >= drop_lo && keep_row[r] < drop_hi)) {
                        const int slot = atomicAdd(const_cast<int32_t *>(&ctrl[kLCount]), 1);
                        cand_key[slot] = keep_key[r];
                        cand_row[slot] = keep_row[r];
                    }
                }
                drop_lo = drop_hi = 0;
                __syncthreads();
                const int kept = uniform(static_cast<int>(ctrl[kLCount]));
                __syncthreads();  // a retried scan appends right away: the count must be read by all threads first
                if (kept > kSelectTrigger) { slow = true; reason = 4; break; }  // massive ties: use the dense kernel
                next_select = min(kSelectTrigger, max(a.select_growth * kept / 2, max(4 * k, a.select_min)));
                count_at_step = kept;
                DS_STAMP(4);
            }
            if (redo_tile && !slow) --b;  // the for statement steps back onto the same tile
        }

        if (!slow) __syncthreads();
        int m = slow ? 0 : uniform(static_cast<int>(ctrl[kLCount]));
        // Fewer than k candidates: fewer than k rows have a positive score at all (nothing is pruned before k candidates
        // exist).  The literal kernel answers it without sweeping anything (the answer is the k largest row indexes).
        if (!slow && m < k) { slow = true; reason = 5; }
        if (!slow) {
            // final tightening so that only k + near-ties + margin survivors are evaluated exactly
            if (m > k) {
                const uint32_t tau_key = radix_select_kth(cand_key, m, k, hist, ctrl, 4);
                ++selects;
                const double tau = static_cast<double>(__uint_as_float(tau_key));
                const double cut_value = tau - 2.0 * error_margin() - 2e-6;
                const float final_cut =
                    (tau_key < 0x7f800000u && cut_value > 0.0) ? round_down_positive(cut_value) : 0.f;
                uint32_t keep_key[kKeep];
                int32_t keep_row[kKeep];
#pragma unroll
                for (int r = 0; r < kKeep; ++r) {
                    const int i = tid + r * kThreads;
                    keep_key[r] = i < m ? cand_key[i] : 0u;
                    keep_row[r] = i < m ? cand_row[i] : -1;
                }
                __syncthreads();
                if (tid == 0) ctrl[kLCount] = 0;
                __syncthreads();
#pragma unroll
                for (int r = 0; r < kKeep; ++r) {
                    if (keep_row[r] >= 0 && __uint_as_float(keep_key[r]) >= final_cut) {
                        const int slot = atomicAdd(const_cast<int32_t *>(&ctrl[kLCount]), 1);
                        cand_row[slot] = keep_row[r];
                    }
                }
                __syncthreads();
                m = uniform(static_cast<int>(ctrl[kLCount]));
            }
            DS_STAMP(4);
            // ---- exact evaluation.  Scratch lives in the (all-zero) score tile:
            //   hit masks   uint32[m][4]  (bit j = row is in the posting list of query column j)
            //   exact value float64[m]    at byte offset 32768
            uint32_t *hit_mask = iscores;
            double *exact_jaccard = reinterpret_cast<double *>(lds + 32768);
            for (int p = tid; p < m * n; p += kThreads) {  // one (candidate, column) membership test per thread
                const int i = p / n, j = p - i * n;
                const int32_t t = cand_row[i];
                if (!DS_OK_INDEX(6, t, a.n_truth)) continue;
                bool hit;
                if (sig_bit[j] >= 0) {
                    const uint32_t *words = reinterpret_cast<const uint32_t *>(a.signature + t);
                    hit = (words[sig_bit[j] >> 5] >> (sig_bit[j] & 31)) & 1u;  // exact membership bit of a dense column
                } else {
                    const int32_t tile = t / kTile;
                    const uint32_t local = static_cast<uint32_t>(t - tile * kTile);
                    const uint32_t *ptr = a.col_ptr + static_cast<int64_t>(cols[j]) * ptr_stride + tile;
                    uint32_t lo = ptr[0] * 4u;
                    const uint32_t end = ptr[1] * 4u;
                    uint32_t hi = end;
                    while (lo < hi) {
                        const uint32_t mid = (lo + hi) >> 1;
                        if (!DS_OK_INDEX(9, mid, a.n_quads * 4)) break;
                        if (a.postings[mid] < local) lo = mid + 1; else hi = mid;
                    }
                    hit = lo < end && DS_OK_INDEX(10, lo, a.n_quads * 4) && a.postings[lo] == local;
                }
                if (hit) atomicOr(&hit_mask[i * 4 + (j >> 5)], 1u << (j & 31));
            }
            __syncthreads();
            for (int i = tid; i < m; i += kThreads) {
                float score = 0.f;
                for (int word = 0; word < 4; ++word) {
                    uint32_t bits = hit_mask[i * 4 + word];
                    while (bits) {  // float32 accumulation in the query's column order (match_maker.py:46-48)
                        const int j = word * 32 + __ffs(bits) - 1;
                        score = score + idf[j];
                        bits &= bits - 1u;
                    }
                }
                const double s = static_cast<double>(score);
                if (!DS_OK_INDEX(7, cand_row[i], a.n_truth)) continue;
                exact_jaccard[i] = s / (static_cast<double>(a.sums32[cand_row[i]]) + (maxint - s));  // :50
            }
            __syncthreads();
            // k-th largest exact value (rank by counting; m is small)
            for (int i = tid; i < m; i += kThreads) {
                const double v = exact_jaccard[i];
                int above = 0;
#pragma unroll 8
                for (int j = 0; j < m; ++j) {  // unrolled: eight LDS reads in flight instead of one wait per value
                    const double w = exact_jaccard[j];
                    above += (w > v) | ((w == v) & (j < i));
                }
                if (above == k - 1) {
                    ctrl[kLKth0] = __double2loint(v);
                    ctrl[kLKth1] = __double2hiint(v);
                }
            }
            __syncthreads();
            const double kth = __hiloint2double(ctrl[kLKth1], ctrl[kLKth0]);
            const float kth32 = static_cast<float>(kth);                                       // match_maker.py:65
            const double threshold = static_cast<double>(kth32) - static_cast<double>(1e-6f);  // match_maker.py:70
            if (threshold <= 0.0) {
                // every row qualifies (zeros included): the k largest row indexes
                for (int j = tid; j < k; j += kThreads)
                    a.out_rows[q * k + j] = static_cast<int32_t>(a.n_truth - 1 - j);
            } else {
                for (int i = tid; i < m; i += kThreads) {
                    if (!(exact_jaccard[i] >= threshold)) continue;                            // match_maker.py:71
                    const int32_t t = cand_row[i];
                    int above = 0;
#pragma unroll 8
                    for (int j = 0; j < m; ++j) above += (exact_jaccard[j] >= threshold) & (cand_row[j] > t);
                    if (above < k && DS_OK_INDEX(8, q * k + above, a.n_queries * k)) a.out_rows[q * k + above] = t;
                }
            }
            __syncthreads();
            for (int i = tid; i < m * 4; i += kThreads) hit_mask[i] = 0u;  // give the tile back zeroed
            for (int i = tid; i < m; i += kThreads) exact_jaccard[i] = 0.0;
            if (tid == 0) {
                a.status[q] = kQueryDone;
                // statistics are summed per workgroup in LDS and reach HBM once, at the end of the kernel
                ctrl[kLStats + 0] += m;
                ctrl[kLStats + 1] += selects;
                ctrl[kLStats + 2] += sparse_tiles;
                ctrl[kLStats + 3] += dense_tiles;
                ctrl[kLStats + 4] += non_essential;
                if constexpr (kCountBytes) *requested += 8ull * query_units;
            }
            __syncthreads();
            DS_STAMP(5);
        } else {
            if (tid == 0) {
                a.status[q] = reason == 5 ? kQuerySlowFew : kQuerySlow;
                a.slow_list[atomicAdd(&a.control[kCtlSlowCount], 1)] = static_cast<int32_t>(q);
                if (reason >= 0) atomicAdd(&a.control[kCtlReason + reason], 1);
                if constexpr (kCountBytes) *requested += 8ull * query_units;
            }
            for (int i = tid * 4; i < kScoreWords; i += kThreads * 4)
                *reinterpret_cast<uint4 *>(&iscores[i]) = make_uint4(0u, 0u, 0u, 0u);
            __syncthreads();
            DS_STAMP(0);
        }
    }
    if constexpr (kCountBytes) {
        __syncthreads();
        if (tid == 0) atomicAdd(reinterpret_cast<unsigned long long *>(a.control + kCtlBytes), *requested);
    }
    if (tid == 0) {
        atomicAdd(&a.control[kCtlExact], ctrl[kLStats + 0]);
        atomicAdd(&a.control[kCtlSelects], ctrl[kLStats + 1]);
        atomicAdd(&a.control[kCtlSparseTiles], ctrl[kLStats + 2]);
        atomicAdd(&a.control[kCtlDenseTiles], ctrl[kLStats + 3]);
        atomicAdd(&a.control[kCtlSkippedColumns], ctrl[kLStats + 4]);
    }
#ifdef DS_DIAGNOSTICS
    if (a.phase != nullptr && tid == 0)
        for (int i = 0; i < 16; ++i) atomicAdd(&a.phase[i], phase_lds[i]);
#endif
}

// ---- the literal algorithm for the queries the fast kernel hands over ------------------------------------------------
// fast_jaccard exactly as the reference runs it (ordered float32 accumulation per row, float64 finalise,
// match_maker.py:45-50), tile by tile FROM THE LAST TILE DOWN, and fast_arg_top_k (:53-71) as a streaming selection
// that needs no N-vector in HBM:
//   the rows that can still matter are kept in an LDS buffer of (float64 jaccard, row).  With T_run = float32(k-th
//   largest value seen so far) - 1e-6 (a lower bound of the final threshold, which only grows):
//     (a) a row below T_run is below the final threshold: dropped for good;
//     (b) a row r that has k rows of LARGER index with a value >= its own can be dropped too: if r qualifies at the end
//         so do those k rows, and :71 returns the k largest indexes; the k-th largest value does not need r either
//         (its k dominators keep it in place).
//   Rows arrive tile by tile, 1024 at a time; when the next 1024 might not fit, the buffer is compacted with (a) and, if
//   that is not enough, (b).  At the end the k-th largest float32 value of the buffer IS the reference's heap minimum.
constexpr int kDenseScoreFloats = kTile + 64;  // float32 score tile + trash slot
constexpr int kDenseChunk = 256;   // query columns whose (list begin, list end, idf) are staged in LDS at a time
constexpr int kDenseBuffer = 3072; // rows kept per query: value float64 + row int32 = 36 KiB
constexpr int kDenseOffHist = kDenseScoreFloats * 4;
constexpr int kDenseOffCtrl = kDenseOffHist + 256 * 4;
constexpr int kDenseOffStage = kDenseOffCtrl + 128;
constexpr int kDenseOffValue = kDenseOffStage + kDenseChunk * 12;
constexpr int kDenseOffRow = kDenseOffValue + kDenseBuffer * 8;
constexpr int kDenseLdsBytes = kDenseOffRow + kDenseBuffer * 4;
static_assert(kDenseLdsBytes <= 160 * 1024 && kDenseOffValue % 8 == 0, "LDS budget of the literal kernel");
constexpr int kDenseKeep = (kDenseBuffer + kDenseThreads - 1) / kDenseThreads;
enum { kDCount = 0, kDDigit, kDRemain, kDQuery, kDKept };

__device__ __forceinline__ uint32_t dense_key(double v)  // the float32 the reference's heap stores (match_maker.py:65)
{
    return v > 0.0 ? __float_as_uint(static_cast<float>(v)) : 0u;
}

// k-th largest float32 key of the m buffered values (m >= k); every thread of the workgroup calls it
__device__ uint32_t dense_select_kth(const double *value, int m, int k, uint32_t *hist, volatile int32_t *ctrl)
{
    const int tid = threadIdx.x;
    uint32_t prefix = 0, mask = 0;
    int remaining = k;
    for (int shift = 24; shift >= 0; shift -= 8) {
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        for (int i = tid; i < m; i += kDenseThreads) {
            const uint32_t key = dense_key(value[i]);
            if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (tid == 0) {
            int cumulative = 0, digit = 0;
            for (int d = 255; d >= 0; --d) {
                const int c = static_cast<int>(hist[d]);
                if (cumulative + c >= remaining) { digit = d; break; }
                cumulative += c;
            }
            ctrl[kDDigit] = digit;
            ctrl[kDRemain] = remaining - cumulative;
        }
        __syncthreads();
        prefix |= static_cast<uint32_t>(ctrl[kDDigit]) << shift;
        mask |= 255u << shift;
        remaining = ctrl[kDRemain];
    }
    return prefix;
}

__global__ __launch_bounds__(kDenseThreads) void ds_jaccard_dense_kernel(JaccardArgs a)
{
    extern __shared__ __align__(16) unsigned char lds[];
    float *scores = reinterpret_cast<float *>(lds);
    uint32_t *hist = reinterpret_cast<uint32_t *>(lds + kDenseOffHist);
    volatile int32_t *ctrl = reinterpret_cast<volatile int32_t *>(lds + kDenseOffCtrl);
    uint32_t *stage_begin = reinterpret_cast<uint32_t *>(lds + kDenseOffStage);
    uint32_t *stage_end = stage_begin + kDenseChunk;
    float *stage_value = reinterpret_cast<float *>(stage_end + kDenseChunk);
    double *value = reinterpret_cast<double *>(lds + kDenseOffValue);
    int32_t *row = reinterpret_cast<int32_t *>(lds + kDenseOffRow);

    const int tid = threadIdx.x, lane = tid & 63;
    const int k = a.k;
    const int64_t n_truth = a.n_truth;
    const int64_t ptr_stride = static_cast<int64_t>(a.n_tiles) + 1;
    const int n_slow = a.control[kCtlSlowCount];

    for (;;) {
        if (tid == 0) {
            ctrl[kDQuery] = atomicAdd(&a.control[kCtlSlowQueue], 1);
            ctrl[kDCount] = 0;
        }
        __syncthreads();
        const int item = ctrl[kDQuery];
        __syncthreads();
        if (item >= n_slow) break;  // exit condition reached by every wave: the queue head only grows
        const int64_t q = a.slow_list[item];
        const int64_t qbase = a.q_rowptr[q];
        const int64_t n = a.q_rowptr[q + 1] - qbase;
        const double maxint = a.q_maxint[q];

        if (a.status[q] == kQuerySlowFew) {
            // The fast kernel found fewer than k rows with a positive score (and its assumptions hold: no negative
            // value exists): the heap of match_maker.py:60-67 keeps a zero, the threshold is -1e-6, EVERY row qualifies
            // and :71 returns the k largest row indexes.
            for (int j = tid; j < k; j += kDenseThreads) a.out_rows[q * k + j] = static_cast<int32_t>(n_truth - 1 - j);
            if (tid == 0) a.status[q] = kQuerySlow;
            __syncthreads();
            continue;
        }
        bool bad = false;
        for (int64_t j = tid; j < n; j += kDenseThreads) {
            const int32_t column = a.q_cols[qbase + j];
            bad |= column < 0 || column >= a.n_columns;
        }
        if (__syncthreads_or(bad || n < 0)) {
            if (tid == 0) {
                a.status[q] = kQueryErrorArg;
                atomicAdd(&a.control[kCtlErrors], 1);
            }
            for (int j = tid; j < k; j += kDenseThreads) a.out_rows[q * k + j] = -1;
            continue;
        }

        // Compaction of the buffer: T_run from the k-th largest float32 value, rule (a), then rule (b) if the buffer is
        // still more than a quarter full.
        double t_run = -1.0;  // no threshold yet: every positive value is kept
        bool failed = false;
        auto compact = [&]() {
            __syncthreads();
            const int m = ctrl[kDCount];
            if (m >= k) {
                const uint32_t kth = dense_select_kth(value, m, k, hist, ctrl);
                const double tightened = static_cast<double>(__uint_as_float(kth)) - static_cast<double>(1e-6f);  // :70
                if (tightened > t_run) t_run = tightened;
            }
            double keep_value[kDenseKeep];
            int32_t keep_row[kDenseKeep];
#pragma unroll
            for (int r = 0; r < kDenseKeep; ++r) {
                const int i = tid + r * kDenseThreads;
                keep_value[r] = i < m ? value[i] : -1.0;
                keep_row[r] = i < m ? row[i] : -1;
                if (!(keep_value[r] >= t_run)) keep_row[r] = -1;                                   // rule (a)
            }
            if (tid == 0) ctrl[kDKept] = 0;  // survivors of rule (a)
            __syncthreads();
            int mine = 0;
#pragma unroll
            for (int r = 0; r < kDenseKeep; ++r) mine += keep_row[r] >= 0;
            if (mine) atomicAdd(const_cast<int32_t *>(&ctrl[kDKept]), mine);
            __syncthreads();
            if (ctrl[kDKept] > kDenseBuffer / 4) {                                                  // rule (b)
#pragma unroll
                for (int r = 0; r < kDenseKeep; ++r) {
                    if (keep_row[r] < 0) continue;
                    int dominators = 0;
                    for (int j = 0; j < m; ++j) {  // the buffer itself is still intact: read-only here
                        const int32_t other = row[j];
                        const double v = value[j];
                        dominators += (other > keep_row[r]) & (v >= keep_value[r]);
                    }
                    if (dominators >= k) keep_row[r] = -1;
                }
            }
            __syncthreads();
            if (tid == 0) ctrl[kDCount] = 0;
            __syncthreads();
#pragma unroll
            for (int r = 0; r < kDenseKeep; ++r) {
                if (keep_row[r] < 0) continue;
                const int slot = atomicAdd(const_cast<int32_t *>(&ctrl[kDCount]), 1);
                value[slot] = keep_value[r];
                row[slot] = keep_row[r];
            }
            __syncthreads();
        };

        // scores[] of tile b: ordered float32 accumulation, a barrier between two columns (match_maker.py:46-48)
        auto scatter_tile = [&](int b) {
            for (int i = tid; i < kDenseScoreFloats; i += kDenseThreads) scores[i] = 0.f;
            __syncthreads();
            for (int64_t j0 = 0; j0 < n; j0 += kDenseChunk) {
                // stage (list begin, list end, idf) of up to kDenseChunk columns: one round of loads instead of a
                // chain of three dependent loads in front of every column
                const int width = static_cast<int>(n - j0 < kDenseChunk ? n - j0 : kDenseChunk);
                if (tid < width) {
                    const int32_t column = a.q_cols[qbase + j0 + tid];
                    const uint32_t *ptr = a.col_ptr + static_cast<int64_t>(column) * ptr_stride + b;
                    stage_begin[tid] = ptr[0] * 4u;
                    stage_end[tid] = ptr[1] * 4u;
                    stage_value[tid] = a.idf32[column];
                }
                __syncthreads();
                // the first four strides of the next column's list are fetched before the barrier that ends the
                // current column, so that only the ordered LDS updates sit between two barriers
                uint32_t next[4];
                auto fetch = [&](int j) {
                    const uint32_t begin = stage_begin[j], end = stage_end[j];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const uint32_t i = begin + tid + u * kDenseThreads;
                        next[u] = i < end ? a.postings[i] : static_cast<uint32_t>(kSentinel);
                    }
                };
                fetch(0);
                for (int j = 0; j < width; ++j) {
                    const float idf = stage_value[j];
                    const uint32_t begin = stage_begin[j], end = stage_end[j];
                    uint32_t local[4] = {next[0], next[1], next[2], next[3]};
                    if (j + 1 < width) fetch(j + 1);
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (local[u] < kTile) scores[local[u]] = scores[local[u]] + idf;  // each row at most once per list
                    for (uint32_t i0 = begin + tid + 4 * kDenseThreads; i0 < end; i0 += 4 * kDenseThreads) {
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const uint32_t i = i0 + u * kDenseThreads;
                            local[u] = i < end ? a.postings[i] : static_cast<uint32_t>(kSentinel);
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            if (local[u] < kTile) scores[local[u]] = scores[local[u]] + idf;
                    }
                    __syncthreads();
                }
            }
        };

        for (int b = a.n_tiles - 1; b >= 0 && !failed; --b) {
            scatter_tile(b);
            // finalise the tile (match_maker.py:50), from its last rows down, and keep what can still matter.  Before a
            // chunk of rows is looked at the buffer has room for all of it: one barrier per chunk decides, uniformly,
            // whether to compact first.
            const int64_t tile_base = static_cast<int64_t>(b) * kTile;
            const int rows_here = static_cast<int>(n_truth - tile_base < kTile ? n_truth - tile_base : kTile);
            for (int top = rows_here; top > 0 && !failed; top -= kDenseThreads) {
                if (__syncthreads_or(ctrl[kDCount] > kDenseBuffer - kDenseThreads)) {
                    compact();
                    if (ctrl[kDCount] > kDenseBuffer - kDenseThreads) failed = true;  // (a) and (b) cannot make room
                    __syncthreads();
                    if (failed) break;
                }
                const int i = top - 1 - tid;
                double v = 0.0;
                if (i >= 0) {
                    const double s = static_cast<double>(scores[i]);
                    v = s / (static_cast<double>(a.sums32[tile_base + i]) + (maxint - s));
                }
                const bool keep = i >= 0 && v > 0.0 && v >= t_run;
                const unsigned long long votes = __ballot(keep);
                if (votes != 0) {
                    const int leader = __ffsll(votes) - 1;
                    int base = 0;
                    if (lane == leader) base = atomicAdd(const_cast<int32_t *>(&ctrl[kDCount]), __popcll(votes));
                    base = __shfl(base, leader);
                    if (keep) {
                        const int slot = base + __popcll(votes & ((1ull << lane) - 1ull));
                        value[slot] = v;
                        row[slot] = static_cast<int32_t>(tile_base + i);
                    }
                }
            }
            __syncthreads();
        }
        if (failed) {
            for (int j = tid; j < k; j += kDenseThreads) a.out_rows[q * k + j] = -1;
            if (tid == 0) {
                a.status[q] = kQueryErrorTies;
                atomicAdd(&a.control[kCtlErrors], 1);
            }
            __syncthreads();
            continue;
        }
        __syncthreads();
        const int m = ctrl[kDCount];
        // fast_arg_top_k: the float32 heap ends as the k largest float32(value > 0) padded with zeros (:60-67)
        double threshold = -static_cast<double>(1e-6f);           // fewer than k positive values: heap minimum 0
        if (m >= k) threshold = static_cast<double>(__uint_as_float(dense_select_kth(value, m, k, hist, ctrl))) -
                                static_cast<double>(1e-6f);       // :70
        if (threshold <= 0.0 && !a.literal_only && maxint > 0.0) {
            // every row qualifies, zeros included (with idf >= 0, sums >= a row's own total and maxint > 0 every
            // denominator is positive and no value negative): (array >= threshold).nonzero()[0][::-1][:k] = the k
            // largest indexes
            for (int j = tid; j < k; j += kDenseThreads) a.out_rows[q * k + j] = static_cast<int32_t>(n_truth - 1 - j);
        } else if (threshold <= 0.0) {
            // Negative idf values or a non-positive max_intersection_possible (both possible through the C ABI only)
            // make negative or undefined jaccards: the rows at or above a non-positive threshold are the zero rows and
            // more -- nothing the buffer holds.  Second sweep from the last
            // tile down, taking qualifying rows in descending index order until k are found (:71).
            int found = 0;
            int32_t *wave_counts = reinterpret_cast<int32_t *>(hist);
            for (int b = a.n_tiles - 1; b >= 0 && found < k; --b) {
                scatter_tile(b);
                const int64_t tile_base = static_cast<int64_t>(b) * kTile;
                const int rows_here = static_cast<int>(n_truth - tile_base < kTile ? n_truth - tile_base : kTile);
                for (int top = rows_here; top > 0 && found < k; top -= kDenseThreads) {
                    const int i = top - 1 - tid;  // thread 0 holds the largest row of the chunk
                    bool qualifies = false;
                    if (i >= 0) {
                        const double s = static_cast<double>(scores[i]);
                        qualifies = s / (static_cast<double>(a.sums32[tile_base + i]) + (maxint - s)) >= threshold;
                    }
                    const unsigned long long votes = __ballot(qualifies);
                    if (lane == 0) wave_counts[tid >> 6] = __popcll(votes);
                    __syncthreads();
                    int before = 0, total = 0;
                    for (int w = 0; w < kDenseThreads / 64; ++w) {
                        before += w < (tid >> 6) ? wave_counts[w] : 0;
                        total += wave_counts[w];
                    }
                    const int position = found + before + __popcll(votes & ((1ull << lane) - 1ull));
                    if (qualifies && position < k) a.out_rows[q * k + position] = static_cast<int32_t>(tile_base + i);
                    found += total;
                    __syncthreads();
                }
            }
            if (found < k) {  // match_maker.py:188-189
                for (int j = found + tid; j < k; j += kDenseThreads) a.out_rows[q * k + j] = -1;
                if (tid == 0) {
                    a.status[q] = kQueryErrorTopN;
                    atomicAdd(&a.control[kCtlErrors], 1);
                }
            }
        } else {
            for (int i = tid; i < m; i += kDenseThreads) {
                if (!(value[i] >= threshold)) continue;           // :71
                const int32_t t = row[i];
                int above = 0;
                for (int j = 0; j < m; ++j) above += (value[j] >= threshold) & (row[j] > t);
                if (above < k) a.out_rows[q * k + above] = t;
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
    if (!index->attributes_set) {
        DS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(ds_jaccard_topk_kernel<false>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, kFastLdsBytes));
        DS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(ds_jaccard_topk_kernel<true>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, kFastLdsBytes));
        DS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(ds_jaccard_dense_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, kDenseLdsBytes));
        index->attributes_set = true;
    }
    JaccardArgs args;
    args.col_ptr = index->col_ptr.ptr;
    args.postings = index->postings.ptr;
    args.posting_sums = index->posting_sums.ptr;
    args.idf32 = index->idf32.ptr;
    args.sums32 = index->sums32.ptr;
    args.tile_sums_min = index->tile_sums_min.ptr;
    args.signature = reinterpret_cast<const uint4 *>(index->signature.ptr);
    args.sig_column = index->sig_column.ptr;
    args.dup_rank = index->dup_rank.ptr;
    args.q_rowptr = d_q_rowptr;
    args.q_cols = d_q_cols;
    args.q_maxint = d_q_maxint;
    args.out_rows = d_out_rows;
    args.status = index->status.ptr;
    args.control = index->control.ptr;
    args.slow_list = index->slow_list.ptr;
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
    args.sparse_quads = 4096;  // measured on C2: 2048..8192 is the flat optimum for 28672-row tiles (profiles/r01_i_tuning.txt)
    if (const char *limit = getenv("DS_SPARSE_QUADS"); limit != nullptr) args.sparse_quads = atoi(limit);
    args.sums_min = index->sums_min;
    args.debug = 0;
    args.literal_only = index->literal_only ? 1 : 0;
    args.select_min = 64;
    args.select_growth = 4;
    if (const char *v = getenv("DS_SELECT_MIN"); v != nullptr) args.select_min = atoi(v);
    if (const char *v = getenv("DS_SELECT_GROWTH"); v != nullptr) args.select_growth = atoi(v);
    args.n_quads = index->n_quads;
    if (const char *debug = getenv("DS_DEBUG"); debug != nullptr) args.debug = atoi(debug);

    const int grid = static_cast<int>(std::min<int64_t>(Q, int64_t(index->compute_units) * kWorkgroupsPerCu));
    DS_HIP(hipMemsetAsync(index->control.ptr, 0, kControlWords * sizeof(int32_t), stream));
    DS_HIP(hipEventRecord(index->event_begin, stream));
    if (index->count_bytes)
        hipLaunchKernelGGL(ds_jaccard_topk_kernel<true>, dim3(grid), dim3(kThreads), kFastLdsBytes, stream, args);
    else
        hipLaunchKernelGGL(ds_jaccard_topk_kernel<false>, dim3(grid), dim3(kThreads), kFastLdsBytes, stream, args);
    DS_HIP(hipGetLastError());
    DS_HIP(hipEventRecord(index->event_fast, stream));
    hipLaunchKernelGGL(ds_jaccard_dense_kernel, dim3(index->compute_units), dim3(kDenseThreads), kDenseLdsBytes, stream, args);
    DS_HIP(hipGetLastError());
    DS_HIP(hipEventRecord(index->event_dense, stream));
    return DS_OK;
}

static int collect(ds_index *index, hipStream_t stream, int64_t stats[32])
{
    DS_REQUIRE(index != nullptr, "ds_jaccard_sync: null index");
    DS_HIP(hipSetDevice(index->device));
    int32_t control[kControlWords] = {0};
    DS_HIP(hipMemcpyAsync(control, index->control.ptr, sizeof(control), hipMemcpyDeviceToHost, stream));
    DS_HIP(hipStreamSynchronize(stream));
    if (index->phase.ptr != nullptr && getenv("DS_PHASE_DUMP") != nullptr) {
        unsigned long long phase[16] = {0};
        DS_HIP(hipMemcpy(phase, index->phase.ptr, sizeof(phase), hipMemcpyDeviceToHost));
        fprintf(stderr, "phase cycles:");
        for (int i = 0; i < 16; ++i) fprintf(stderr, " %d=%llu", i, phase[i]);
        fprintf(stderr, "\n");
    }
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
        stats[12] = control[kCtlSparseTiles];
        stats[13] = control[kCtlDenseTiles];
        stats[14] = control[kCtlSkippedColumns];
        std::memcpy(&stats[15], &control[kCtlBytes], sizeof(int64_t));  // bytes requested by the fast kernel
        for (int i = 0; i < 6; ++i) stats[16 + i] = control[kCtlReason + i];
        stats[22] = control[kCtlRefines];
        stats[23] = control[kCtlRawEntries];
        stats[24] = control[kCtlSurvivors];
        stats[25] = control[kCtlRawSparse];
        float fast_ms = 0.f, dense_ms = 0.f;
        if (index->last_queries > 0 && hipEventElapsedTime(&fast_ms, index->event_begin, index->event_fast) == hipSuccess &&
            hipEventElapsedTime(&dense_ms, index->event_fast, index->event_dense) == hipSuccess) {
            stats[26] = static_cast<int64_t>(fast_ms * 1000.f);
            stats[27] = static_cast<int64_t>(dense_ms * 1000.f);
        } else {
            stats[26] = stats[27] = 0;
        }
        for (int i = 28; i < 32; ++i) stats[i] = control[24 + (i - 28)];  // bounds-check record of debug builds
        if (getenv("DS_PHASE_DUMP") != nullptr)
            fprintf(stderr, "collect experiment: touched rows %d level-1 rows %d wave-quads %d wave-quads with level-1 %d\n",
                    control[20], control[21], control[22], control[23]);
    }
    if (control[kCtlErrors] != 0 && index->last_queries > 0) {
        std::vector<int32_t> status(static_cast<size_t>(index->last_queries));
        DS_HIP(hipMemcpy(status.data(), index->status.ptr, status.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
        for (size_t q = 0; q < status.size(); ++q) {
            if (status[q] == kQueryErrorTopN) {
                set_error("top_matches.shape[0] != self.top_n (query %zu)", q);
                return DS_E_TOP_N;
            }
            if (status[q] == kQueryErrorTies) {
                set_error("ds_jaccard_topk: query %zu has more than %d rows within 1e-6 of its k-th largest jaccard that "
                          "are neither twins nor dominated (literal kernel buffer)", q, 3072);
                return DS_E_INTERNAL;
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

int ds_jaccard_status(ds_index *index, void *stream, int32_t *status, int64_t Q)
{
    DS_REQUIRE(index != nullptr && status != nullptr, "ds_jaccard_status: null argument");
    DS_REQUIRE(Q == index->last_queries, "ds_jaccard_status: Q=%lld but the last call had %lld queries", (long long)Q,
               (long long)index->last_queries);
    if (Q == 0) return DS_OK;
    DS_HIP(hipSetDevice(index->device));
    DS_HIP(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    DS_HIP(hipMemcpy(status, index->status.ptr, static_cast<size_t>(Q) * sizeof(int32_t), hipMemcpyDeviceToHost));
    return DS_OK;
}

int ds_index_option(ds_index *index, const char *name, int64_t value)
{
    DS_REQUIRE(index != nullptr && name != nullptr, "ds_index_option: null argument");
    if (std::strcmp(name, "count_bytes") == 0) {
        index->count_bytes = value != 0;
        return DS_OK;
    }
    ds::set_error("ds_index_option: unknown option '%s'", name);
    return DS_E_ARG;
}

int ds_jaccard_sync(ds_index *index, void *stream, int64_t stats[32])
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
