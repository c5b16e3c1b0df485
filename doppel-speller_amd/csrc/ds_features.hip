// fast_levenshtein_ratio + construct_features (doppelspeller/feature_engineering.py:25-169) on gfx950.
//
// Reference semantics: the DP of :39-61 has insert/delete cost 1 and substitute cost 2, i.e. it computes the indel
// distance d = L - 2*LCS(a, b) with L = len(a)+len(b); the returned value is uint8(((L - d) / L) * 100) evaluated in
// float64 in source order (:63).  The DP matrix is uint8 (:42): as long as L <= 255 no cell can wrap and the
// identity is exact; beyond that the stores wrap modulo 256 and only a literal emulation reproduces the result.
//
// Half a wavefront (32 lanes) per (title, truth title) pair -- two pairs share every vector instruction, the kernel being
// bound by instruction issue and most of a pair's work using ~20 lanes (one window start per lane):
//   * LCS by the bit-parallel recurrence V' = (V + (V & M[c])) | (V & ~M[c]) over a 64-bit column vector (pattern =
//     the shorter string, <= 64 chars; M[c] = match mask of the pattern for character code c, built in LDS with one
//     ds_or per pattern character);
//   * the word loop of :128-155 maps one window start per lane: every lane runs the recurrence for its window
//     against the current truth word, then a half-wave max-reduction keeps the first best window (:147-149);
//   * pairs that do not fit (L > 255, pattern > 64 chars, character code >= 64) take the literal path: an
//     anti-diagonal DP with uint8 wrap-around, three diagonals staged in LDS, 64 cells per step.
// All strings, masks and the reconstructed title live in LDS; HBM traffic is the title bytes in and 264 B out.
#include <cmath>
#include <cstddef>
#include <map>
#include <mutex>

#include "ds_common.h"
#include "ds_host.h"

namespace ds {

constexpr int kFeatWaves = 4;
constexpr int kReconCap = 288;  // 1 + sum(word lengths) + 15 separators <= 1 + 255 + 16

struct FeatureArgs {
    const uint8_t *q_enc;
    const uint8_t *q_len;
    const uint8_t *t_enc;
    const uint8_t *t_len;
    const uint32_t *t_counts;
    const int32_t *pair_q;  // nullable
    const int32_t *pair_t;  // nullable
    const uint32_t *q_off;  // nullable: row i of the titles starts at q_enc + q_off[i] instead of q_enc + i * q_stride
    const uint32_t *t_off;  //           (the packed staging of the host-pointer entry point ships only the titles' own bytes)
    const unsigned char *t_records;  // nullable: one TruthRecord per truth row (indexed entry points)
    int32_t *unit_queue;             // nullable: head of the work queue of units (zeroed in front of the launch); without it the
                                     // units are dealt round-robin
    float *out;
    int64_t q_stride, t_stride;
    int64_t n_q, n_t;       // table sizes (bounds for indexes)
    int64_t n;              // pairs
    int32_t unit_pairs;     // consecutive pairs one wave works through, two at a time (a query's title is staged once for its run of them)
    int64_t q_first;
    int32_t k;              // > 0: pair i belongs to query q_first + i / k (when pair_q is null)
    uint32_t n_truth;
    uint8_t space_code;
};

struct PairScratch {           // what the features kernel needs per pair (2,752 B)
    unsigned long long masks[64];
    uint8_t q[256];
    uint8_t t[256];
    uint8_t qw[256];
    uint8_t recon[kReconCap];
    uint8_t diag[3][kReconCap];
    float features[72];
    uint8_t word_begin[16];
    uint8_t word_len[16];
};
struct WaveScratch : PairScratch {
    uint8_t token_begin[128];  // token-sort scratch of the close-match kernel
    uint8_t token_len[128];
};

__device__ __forceinline__ uint8_t ratio_from_lcs(int lcs, int total_length)
{
    if (total_length == 0) return 0;  // 0/0 in the reference; unreachable from its callers
    const double ratio = (static_cast<double>(2 * lcs) / static_cast<double>(total_length)) * 100.0;  // :63
    return static_cast<uint8_t>(ratio);
}

__device__ __forceinline__ void wave_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }

// Build M[c] for pattern[0..m) (m <= 64, all codes < 64) in scratch.masks.  All 64 lanes participate.
__device__ __forceinline__ void build_masks(WaveScratch &w, const uint8_t *pattern, int m, int lane)
{
    w.masks[lane] = 0ull;
    wave_sync();
    if (lane < m) atomicOr(&w.masks[pattern[lane]], 1ull << lane);
    wave_sync();
}

// LCS length of text[0..n) against the pattern whose masks are in scratch.masks (pattern length m <= 64).
__device__ __forceinline__ int lcs_bitparallel(const PairScratch &w, const uint8_t *text, int n, int m)
{
    unsigned long long v = ~0ull;
#pragma unroll 4
    for (int i = 0; i < n; ++i) {  // unrolled: the two dependent LDS reads of four steps are issued ahead of the recurrence
        const unsigned long long match = w.masks[text[i]];
        const unsigned long long u = v & match;
        v = (v + u) | (v & ~match);
    }
    const unsigned long long valid = m >= 64 ? ~0ull : ((1ull << m) - 1ull);
    return __popcll(~v & valid);
}

// (Round 4 tried a 32-bit recurrence for patterns of at most 32 characters -- 95 % of the steps; 5-8 instead of 9-10 vector
// instructions per step in the ISA: the kernel got 17 % SLOWER, 1.70 -> 2.00 ms per 1M pairs, profiles/r04_tuning.txt.  The
// steps are not what binds it: a second code path per call site and 4-byte reads at an 8-byte stride cost more.)
// Literal fast_levenshtein_ratio (:25-63) with uint8 wrap-around, cooperative over the wave; a/b in LDS.
__device__ uint8_t levenshtein_literal(WaveScratch &w, const uint8_t *a, int la, const uint8_t *b, int lb, int lane)
{
    const int total_length = la + lb;
    if (la > lb) {  // :35-37
        const uint8_t *ts = a; a = b; b = ts;
        const int tl = la; la = lb; lb = tl;
    }
    // diagonal d holds matrix[x][d - x] at index x, x in [max(0, d - lb), min(la, d)]
    for (int d = 0; d <= la + lb; ++d) {
        uint8_t *current = w.diag[d % 3];
        const uint8_t *previous = w.diag[(d + 2) % 3];
        const uint8_t *before = w.diag[(d + 1) % 3];
        const int x_low = d > lb ? d - lb : 0;
        const int x_high = d < la ? d : la;
        for (int x = x_low + lane; x <= x_high; x += 64) {
            const int y = d - x;
            int value;
            if (x == 0) value = y;            // :45-46
            else if (y == 0) value = x;       // :43-44
            else {
                const int up = previous[x - 1] + 1;                                   // matrix[x-1, y] + 1
                const int diagonal = before[x - 1] + (a[x - 1] == b[y - 1] ? 0 : 2);  // matrix[x-1, y-1] (+2)
                const int left = previous[x] + 1;                                     // matrix[x, y-1] + 1
                value = min(up, min(diagonal, left));                                 // int64 min, then uint8 store
            }
            current[x] = static_cast<uint8_t>(value);
        }
        wave_sync();
    }
    if (total_length == 0) return 0;
    const int distance = w.diag[(la + lb) % 3][la];
    const double ratio = (static_cast<double>(total_length - distance) / static_cast<double>(total_length)) * 100.0;
    return static_cast<uint8_t>(ratio);
}

__device__ __forceinline__ bool codes_below_64(const uint8_t *s, int n, int lane)
{
    bool ok = true;
    for (int i = lane; i < n; i += 64) ok &= s[i] < 64;
    return __all(ok);
}

// fast_levenshtein_ratio of two LDS strings, wave-uniform result.  `small_alphabet` = every code of both < 64.
__device__ uint8_t levenshtein_wave(WaveScratch &w, const uint8_t *a, int la, const uint8_t *b, int lb, int lane,
                                    bool small_alphabet)
{
    const int shorter = la < lb ? la : lb;
    if (small_alphabet && la + lb <= 255 && shorter <= 64) {
        const uint8_t *pattern = la <= lb ? a : b;
        const uint8_t *text = la <= lb ? b : a;
        build_masks(w, pattern, shorter, lane);
        const int lcs = lcs_bitparallel(w, text, la + lb - shorter, shorter);
        return ratio_from_lcs(lcs, la + lb);
    }
    return levenshtein_literal(w, a, la, b, lb, lane);
}

// ---- construct_features: half a wavefront (32 lanes) per pair --------------------------------------------------------
// The kernel is bound by vector-instruction issue, and most of a pair's work is either uniform over its lanes (the two
// whole-title comparisons) or uses ~20 of them (one window start per lane).  Two pairs share a wavefront: every
// instruction now serves two pairs.  Lanes 0-31 work on one pair, lanes 32-63 on another, each with its own scratch;
// control flow may diverge between the halves (different word counts and lengths), never inside one.
#ifndef DS_FEAT_GROUP
#define DS_FEAT_GROUP 32   // lanes per pair: 32 (two pairs per wave) or 16 (four)
#endif
#ifndef DS_FEAT_WAVES
#define DS_FEAT_WAVES 4    // waves per workgroup of ds_construct_features_kernel
#endif
constexpr int kGroup = DS_FEAT_GROUP, kPairsPerWave = 64 / kGroup, kFeatKernelWaves = DS_FEAT_WAVES;
static_assert((kGroup == 16 || kGroup == 32) && kGroup > DS_WORDS, "a pair's lanes: one per truth word at least");

// the ballot bits of this lane's group
__device__ __forceinline__ uint32_t group_ballot(bool predicate, int group)
{
    if constexpr (kGroup == 32) return static_cast<uint32_t>(__ballot(predicate) >> (group * kGroup));
    return static_cast<uint32_t>(__ballot(predicate) >> (group * kGroup)) & ((1u << (kGroup & 31)) - 1u);
}

__device__ __forceinline__ void build_masks_g(PairScratch &w, const uint8_t *pattern, int m, int gl)
{
    w.masks[gl] = 0ull;
    w.masks[gl + kGroup] = 0ull;
    if constexpr (kGroup == 16) {
        w.masks[gl + 32] = 0ull;
        w.masks[gl + 48] = 0ull;
    }
    wave_sync();
    for (int j = gl; j < m; j += kGroup) atomicOr(&w.masks[pattern[j]], 1ull << j);
    wave_sync();
}

// levenshtein_literal with 32 cooperating lanes
__device__ uint8_t levenshtein_literal_g(PairScratch &w, const uint8_t *a, int la, const uint8_t *b, int lb, int gl)
{
    const int total_length = la + lb;
    if (la > lb) {  // :35-37
        const uint8_t *ts = a; a = b; b = ts;
        const int tl = la; la = lb; lb = tl;
    }
    for (int d = 0; d <= la + lb; ++d) {
        uint8_t *current = w.diag[d % 3];
        const uint8_t *previous = w.diag[(d + 2) % 3];
        const uint8_t *before = w.diag[(d + 1) % 3];
        const int x_low = d > lb ? d - lb : 0;
        const int x_high = d < la ? d : la;
        for (int x = x_low + gl; x <= x_high; x += kGroup) {
            const int y = d - x;
            int value;
            if (x == 0) value = y;            // :45-46
            else if (y == 0) value = x;       // :43-44
            else {
                const int up = previous[x - 1] + 1;
                const int diagonal = before[x - 1] + (a[x - 1] == b[y - 1] ? 0 : 2);
                const int left = previous[x] + 1;
                value = min(up, min(diagonal, left));
            }
            current[x] = static_cast<uint8_t>(value);
        }
        wave_sync();
    }
    if (total_length == 0) return 0;
    const int distance = w.diag[(la + lb) % 3][la];
    const double ratio = (static_cast<double>(total_length - distance) / static_cast<double>(total_length)) * 100.0;
    return static_cast<uint8_t>(ratio);
}

__device__ __forceinline__ bool codes_below_64_g(const uint8_t *s, int n, int gl, int group)
{
    bool ok = true;
    for (int i = gl; i < n; i += kGroup) ok &= s[i] < 64;
    return group_ballot(!ok, group) == 0u;
}

__device__ uint8_t levenshtein_g(PairScratch &w, const uint8_t *a, int la, const uint8_t *b, int lb, int gl,
                                 bool small_alphabet)
{
    const int shorter = la < lb ? la : lb;
    if (small_alphabet && la + lb <= 255 && shorter <= 64) {
        const uint8_t *pattern = la <= lb ? a : b;
        const uint8_t *text = la <= lb ? b : a;
        build_masks_g(w, pattern, shorter, gl);
        const int lcs = lcs_bitparallel(w, text, la + lb - shorter, shorter);
        return ratio_from_lcs(lcs, la + lb);
    }
    return levenshtein_literal_g(w, a, la, b, lb, gl);
}

// ratio_from_lcs for every (total length <= 128, LCS <= 64): what the word loop needs (window <= word <= 64 chars).  The
// float64 evaluation of :63 costs ~35 instructions per window: tabulated ONCE per device (round 4: once per workgroup, 8 % of
// the kernel's instructions), a workgroup copies the 8 KiB into LDS.
constexpr int kRatioLengths = 129, kRatioLcs = 65, kRatioEntries = (kRatioLengths * kRatioLcs + 3) & ~3;
__device__ uint8_t g_ratio_table[kRatioEntries];
__global__ __launch_bounds__(256) void ds_ratio_table_kernel()
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < kRatioEntries) g_ratio_table[i] = i < kRatioLengths * kRatioLcs ? ratio_from_lcs(i % kRatioLcs, i / kRatioLcs) : 0;
}

// What construct_features derives from the TRUTH TITLE alone -- 45 of its 66 outputs and the loop bounds of the rest -- once
// per truth row instead of once per pair (a truth row is a candidate of 2 queries of C2 on average, of 20 at top-100): the
// word boundaries of :110-114, truth_number_of_words (:105), idf_s (:153; the float64 log is ~150 instructions) and
// ranks_idf_s (:158), for one (number_of_truth_titles, space code).  160 bytes per row.
struct TruthRecord {          // 40 words: a half-wave fetches words 0..31 with ONE load (lane = word), words 32..39 with a second
    float idf_s[DS_WORDS];    // NaN beyond the title's words (:121-123)
    float ranks[DS_WORDS];
    uint8_t n_words;          // words of the title, at most 15 (:114)          } word 30
    uint8_t small_alphabet;   // every character code of the title is below 64  }
    uint16_t truth_words;     // spaces + 1 (:105)                              }
    uint32_t unused;          // word 31
    uint8_t word_begin[16], word_len[16];   // words 32..35, 36..39 (entry 15 unused)
};
static_assert(sizeof(TruthRecord) == 160 && offsetof(TruthRecord, n_words) == 120 && offsetof(TruthRecord, word_begin) == 128 &&
                  DS_WORDS == 15 && kGroup == 32, "truth records: two loads of a half-wave");

// idf_s of one word (:153) and the ranks of a title's words (:158): the SAME expressions for the records and for the kernel
// that has none (float64 log of the device library; NaN bit patterns follow x86-64 SSE, the reference's platform: a NaN
// operand propagates unchanged (+qNaN from the np.nan fill of :121-123), an invalid operation (inf - inf when a word count is
// 0) produces the default NaN, which has the sign bit set).
__device__ __forceinline__ float idf_of_word(uint32_t n_truth, uint32_t count)
{
    return static_cast<float>(log(static_cast<double>(n_truth) / static_cast<double>(count)));
}
__device__ __forceinline__ float nan_max(float maximum, float other)  // np.nanmax, two values at a time
{
    return (other != other) ? maximum : ((maximum != maximum || other > maximum) ? other : maximum);
}
__device__ __forceinline__ float rank_of_word(float idf, float maximum, int truth_words)
{
    if (idf != idf) return __uint_as_float(0x7fc00000u);
    const float difference = maximum - idf;  // float32 difference, float64 quotient
    return (difference != difference) ? __uint_as_float(0xffc00000u)
                                      : static_cast<float>(1.0 + static_cast<double>(difference) / static_cast<double>(truth_words));
}

// one thread per truth row (a one-time pass over the title table: 13 GB of reads for C5's 50M rows)
__global__ __launch_bounds__(256) void ds_truth_records_kernel(const uint8_t *enc, int64_t stride, const uint8_t *len,
                                                               const uint32_t *counts, int64_t n, uint32_t n_truth,
                                                               uint8_t space, TruthRecord *records)
{
    const int64_t row = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (row >= n) return;
    const uint8_t *title = enc + row * stride;
    const int lt = len[row];
    TruthRecord r;
    int n_words = 0, previous_end = 0, spaces = 0;
    bool small = true;
    for (int i = 0; i <= lt; ++i) {  // positions of the spaces of truth + [space], first 15 (:110-114)
        const uint8_t c = i < lt ? title[i] : space;
        if (i < lt) {
            small = small && c < 64;
            spaces += c == space;
        }
        if (c == space && n_words < DS_WORDS) {
            r.word_begin[n_words] = static_cast<uint8_t>(previous_end);
            r.word_len[n_words] = static_cast<uint8_t>(i - previous_end);
            previous_end = i + 1;
            ++n_words;
        }
    }
    for (int j = n_words; j < 16; ++j) r.word_begin[j] = r.word_len[j] = 0;
    const int truth_words = spaces + 1;
    float maximum = __uint_as_float(0x7fc00000u);
    for (int j = 0; j < DS_WORDS; ++j) {
        r.idf_s[j] = j < n_words ? idf_of_word(n_truth, counts[row * DS_WORDS + j]) : __uint_as_float(0x7fc00000u);
        maximum = nan_max(maximum, r.idf_s[j]);
    }
    for (int j = 0; j < DS_WORDS; ++j) r.ranks[j] = rank_of_word(r.idf_s[j], maximum, truth_words);
    r.n_words = static_cast<uint8_t>(n_words);
    r.small_alphabet = small ? 1 : 0;
    r.truth_words = static_cast<uint16_t>(truth_words);
    r.unused = 0u;
    records[row] = r;
}

// kRecords: the truth rows come with their TruthRecord (indexed entry points); without them (the 9-argument entry point: the
// caller's own arrays, every pair its own copy of both titles) the kernel derives everything itself.
#ifndef DS_FEAT_POP
#define DS_FEAT_POP 2           // units a wave takes off the work queue at a time
#endif
#ifndef DS_FEAT_MIN_WAVES
#define DS_FEAT_MIN_WAVES 5   // waves per SIMD the register allocation leaves room for (measured: 4 / 5 / 6 -> C2 1.78 / 1.70 / 1.73 ms, top-100 13.0 / 12.2 / 12.7)
#endif
template <bool kRecords>
__global__ __launch_bounds__(kFeatKernelWaves * 64, DS_FEAT_MIN_WAVES) void ds_construct_features_kernel(FeatureArgs a)
{
    __shared__ PairScratch scratch[kFeatKernelWaves * kPairsPerWave];
    __shared__ __align__(4) uint8_t ratio_table[kRatioEntries];
    for (int i = threadIdx.x; i < kRatioEntries / 4; i += kFeatKernelWaves * 64)
        reinterpret_cast<uint32_t *>(ratio_table)[i] = reinterpret_cast<const uint32_t *>(g_ratio_table)[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, group = kGroup == 32 ? lane >> 5 : lane >> 4, gl = lane & (kGroup - 1);
    PairScratch &w = scratch[(threadIdx.x >> 6) * kPairsPerWave + group];
    const int64_t wave_global = static_cast<int64_t>(blockIdx.x) * kFeatKernelWaves + (threadIdx.x >> 6);
    const int64_t wave_count = static_cast<int64_t>(gridDim.x) * kFeatKernelWaves;
    const uint8_t space = a.space_code;
    const float nan = __uint_as_float(0x7fc00000u);
    const int64_t unit_pairs = a.unit_pairs;

    // A wave works through UNITS of `unit_pairs` consecutive pairs, its halves taking alternate pairs: the k candidates of a
    // query are consecutive, so a half stages the query's title (copy, count, squeeze) once for its share of the run -- and
    // both halves work on the SAME query, whose length sets the trip counts of the loops they walk together.
    int64_t taken_next = 0, taken_end = 0;  // the units this wave has taken off the queue and not worked through yet
    for (int64_t unit = wave_global;; unit += wave_count) {
        if (a.unit_queue != nullptr) {  // the waves pull their units from a queue: nobody idles while units are left --
            if (taken_next >= taken_end) {  // DS_FEAT_POP at a time: every pop is an atomic on ONE address (100,000 units: the
                int next = 0;               // queue itself was a third of C2's launch at one unit per pop, profiles/r05_tuning.txt)
                if (lane == 0) next = atomicAdd(a.unit_queue, DS_FEAT_POP);
                taken_next = __builtin_amdgcn_readfirstlane(next);
                taken_end = taken_next + DS_FEAT_POP;
            }
            unit = taken_next++;
        }
        if (unit * unit_pairs >= a.n) break;  // (reached by every wave: the queue head only grows)
        int64_t staged_query = -1;
        int lq = 0, lw = 0, title_words = 0;
        bool small_q = false;
        const int64_t unit_end = min(a.n, (unit + 1) * unit_pairs);
        for (int64_t pair = unit * unit_pairs + group; pair < unit_end; pair += kPairsPerWave) {
        float *out = a.out + pair * DS_FEATURES_COUNT;
        const int64_t qi = a.pair_q ? a.pair_q[pair] : (a.k > 0 ? a.q_first + pair / a.k : pair);
        const int64_t ti = a.pair_t ? a.pair_t[pair] : pair;
        if (qi < 0 || qi >= a.n_q || ti < 0 || ti >= a.n_t) {  // e.g. a -1 row of a failed top-k
            if constexpr (kGroup == 32) {
                out[gl] = nan;
                out[kGroup + gl] = nan;
                if (gl < 2) out[64 + gl] = nan;
            } else {
                for (int i = gl; i < DS_FEATURES_COUNT; i += kGroup) out[i] = nan;
            }
            continue;
        }
        const int lt = a.t_len[ti];                                                        // :102
        const uint8_t *gt = a.t_enc + (a.t_off ? static_cast<int64_t>(a.t_off[ti]) : ti * a.t_stride);
        // the record: words 0..31 (idf_s, ranks, the counts word) and 32..39 (word boundaries), requested here, used behind the
        // staging.  (Round 5 also kept three pairs in flight -- indexes two pairs ahead, row data one pair ahead: no gain, the
        // waves do not wait for these loads; profiles/r05_tuning.txt.)
        uint32_t record_low = 0, record_high = 0;
        if constexpr (kRecords) {
            const uint32_t *record = reinterpret_cast<const uint32_t *>(reinterpret_cast<const TruthRecord *>(a.t_records) + ti);
            record_low = record[gl];
            if (gl < 8) record_high = record[32 + gl];
        }
        wave_sync();
        if (qi != staged_query) {
            // stage the title; count its spaces; squeeze them out (:101, :104, :108)
            staged_query = qi;
            lq = a.q_len[qi];
            const uint8_t *gq = a.q_enc + (a.q_off ? static_cast<int64_t>(a.q_off[qi]) : qi * a.q_stride);
            int spaces_q = 0;
            bool small = true;
            lw = 0;
            for (int base = 0; base < lq; base += kGroup) {
                const int i = base + gl;
                const uint8_t cq = i < lq ? gq[i] : 0;
                if (i < 256) w.q[i] = cq;
                small &= cq < 64;
                const bool is_char = i < lq && cq != space;
                const uint32_t keep = group_ballot(is_char, group);
                if (is_char) w.qw[lw + __popc(keep & ((1u << gl) - 1u))] = cq;
                lw += __popc(keep);
                spaces_q += __popc(group_ballot(i < lq && cq == space, group));
            }
            title_words = spaces_q + 1;
            small_q = group_ballot(!small, group) == 0u;
        }
        // stage the truth title (:102, :105)
        int spaces_t = 0;
        bool small_t = true;
        for (int base = 0; base < lt; base += kGroup) {
            const int i = base + gl;
            const uint8_t ct = i < lt ? gt[i] : 0;
            if (i < 256) w.t[i] = ct;
            if constexpr (!kRecords) {
                small_t &= ct < 64;
                spaces_t += __popc(group_ballot(i < lt && ct == space, group));
            }
        }
        wave_sync();
        int truth_words, n_words = 0;
        bool small_alphabet;
        float record_idf = nan, record_rank = nan;
        if constexpr (kRecords) {
            const uint32_t record_tail = static_cast<uint32_t>(__shfl(static_cast<int>(record_low), 30, kGroup));
            n_words = static_cast<int>(record_tail & 0xffu);
            small_alphabet = small_q && ((record_tail >> 8) & 0xffu) != 0u;
            truth_words = static_cast<int>(record_tail >> 16);
            record_idf = __uint_as_float(record_low);                                                    // lanes 0..14
            record_rank = __uint_as_float(static_cast<uint32_t>(__shfl(static_cast<int>(record_low), (gl + DS_WORDS) & 31, kGroup)));
            if (gl < 8) reinterpret_cast<uint32_t *>(w.word_begin)[gl] = record_high;  // word_begin[16] and word_len[16] are adjacent
        } else {
            truth_words = spaces_t + 1;
            small_alphabet = small_q && group_ballot(!small_t, group) == 0u;
            // truth word boundaries: positions of the spaces of truth + [space], first 15 (:110-114)
            int previous_end = 0;  // start of the current word
            for (int base = 0; base <= lt && n_words < DS_WORDS; base += kGroup) {
                const int i = base + gl;
                const bool is_space = i <= lt && (i == lt || w.t[i] == space);
                uint32_t votes = group_ballot(is_space, group);
                while (votes && n_words < DS_WORDS) {
                    const int position = base + __ffs(votes) - 1;
                    if (gl == 0) {
                        w.word_begin[n_words] = static_cast<uint8_t>(previous_end);
                        w.word_len[n_words] = static_cast<uint8_t>(position - previous_end);
                    }
                    previous_end = position + 1;
                    ++n_words;
                    votes &= votes - 1u;
                }
            }
        }
        wave_sync();

        const uint8_t lev_ratio = levenshtein_g(w, w.q, lq, w.t, lt, gl, small_alphabet);  // :106

        // ---- truth words loop (:128-155)
        if (gl < DS_WORDS) {
            w.features[6 + gl] = nan;
            w.features[6 + DS_WORDS + gl] = nan;
            if constexpr (kRecords) {
                w.features[6 + 2 * DS_WORDS + gl] = record_idf;    // :153
                w.features[6 + 3 * DS_WORDS + gl] = record_rank;   // :158
            } else {
                w.features[6 + 2 * DS_WORDS + gl] = nan;
            }
        }
        int lr = 0;
        if (gl == 0) w.recon[0] = space;  // :115
        lr = 1;
        for (int word = 0; word < n_words; ++word) {
            const int begin = w.word_begin[word], length = w.word_len[word];
            const uint8_t *truth_word = w.t + begin;
            int best_key = 0;  // (ratio << 8) | (255 - start): larger ratio first, then the earliest window
            if (length > 0 && lw > 0) {
                const bool fast = small_alphabet && length <= 64;
                if (fast) build_masks_g(w, truth_word, length, gl);
                for (int base = 0; base < lw; base += kGroup) {
                    const int start = base + gl;
                    int ratio = 0;
                    if (fast) {
                        if (start < lw) {
                            const int window = min(length, lw - start);                      // :142
                            const int lcs = lcs_bitparallel(w, w.qw + start, window, length);
                            ratio = ratio_table[(window + length) * kRatioLcs + lcs];         // :146
                        }
                    } else {
                        for (int s0 = base; s0 < min(base + kGroup, lw); ++s0) {              // literal, one window at a time
                            const int window = min(length, lw - s0);
                            const uint8_t r = levenshtein_literal_g(w, w.qw + s0, window, truth_word, length, gl);
                            if (s0 == start) ratio = r;
                        }
                    }
                    int key = start < lw ? ((ratio << 8) | (255 - start)) : 0;
                    for (int offset = kGroup / 2; offset > 0; offset >>= 1) key = max(key, __shfl_xor(key, offset, kGroup));
                    best_key = max(best_key, key);
                }
            }
            const int best_ratio = best_key >> 8;                                             // :147-149
            int match_start = 0, match_length = 1;
            const bool matched = best_ratio > 0;
            if (matched) {
                match_start = 255 - (best_key & 255);
                match_length = min(length, lw - match_start);
            }
            // reconstructed += best_match + [space]  (:154-155); best_match = [space] when nothing matched (:140)
            for (int i = gl; i < match_length; i += kGroup) w.recon[lr + i] = matched ? w.qw[match_start + i] : space;
            if (gl == 0) {
                w.recon[lr + match_length] = space;
                w.features[6 + word] = static_cast<float>(best_ratio);                        // :151
                w.features[6 + DS_WORDS + word] = static_cast<float>(length);                 // :152
            }
            lr += match_length + 1;
            wave_sync();
        }

        if constexpr (!kRecords) {
            // :153  idf_s of every word at once, one lane per word (the float64 log is ~150 instructions: evaluated once
            // per pair for all lanes instead of once per word on a single lane)
            if (gl < n_words) w.features[6 + 2 * DS_WORDS + gl] = idf_of_word(a.n_truth, a.t_counts[ti * DS_WORDS + gl]);
            wave_sync();
        }

        // :161-162  strip the first and the last space
        const uint8_t recon_ratio = levenshtein_g(w, w.recon + 1, lr - 2, w.t, lt, gl, small_alphabet);

        if constexpr (!kRecords) {
            // :158  ranks = 1 + (nanmax(idf_s) - idf_s) / truth_number_of_words
            const float idf = gl < DS_WORDS ? w.features[6 + 2 * DS_WORDS + gl] : nan;
            float maximum = idf;
            for (int offset = 8; offset > 0; offset >>= 1) maximum = nan_max(maximum, __shfl_xor(maximum, offset, 16));
            if (gl < DS_WORDS) w.features[6 + 3 * DS_WORDS + gl] = rank_of_word(idf, maximum, truth_words);
        }
        if (gl == 0) {  // :164-167
            w.features[0] = static_cast<float>(lq);
            w.features[1] = static_cast<float>(lt);
            w.features[2] = static_cast<float>(title_words);
            w.features[3] = static_cast<float>(truth_words);
            w.features[4] = static_cast<float>(lev_ratio);
            w.features[5] = static_cast<float>(recon_ratio);
        }
        wave_sync();
        if constexpr (kGroup == 32) {
            out[gl] = w.features[gl];
            out[kGroup + gl] = w.features[kGroup + gl];
            if (gl < 2) out[64 + gl] = w.features[64 + gl];
        } else {
#pragma unroll
            for (int i = 0; i < DS_FEATURES_COUNT; i += kGroup)
                if (i + gl < DS_FEATURES_COUNT) out[i + gl] = w.features[i + gl];
        }
        }
    }
}

// fast_levenshtein_ratio for independent pairs (test / cross-check entry point); one wave per pair.
struct LevArgs {
    const uint8_t *a_chars;
    const int64_t *a_off;
    const uint8_t *b_chars;
    const int64_t *b_off;
    uint8_t *out;
    int64_t n;
    int32_t method;
};

__global__ __launch_bounds__(kFeatWaves * 64) void ds_levenshtein_kernel(LevArgs a)
{
    __shared__ WaveScratch scratch[kFeatWaves];
    const int lane = threadIdx.x & 63;
    WaveScratch &w = scratch[threadIdx.x >> 6];
    const int64_t wave_global = static_cast<int64_t>(blockIdx.x) * kFeatWaves + (threadIdx.x >> 6);
    const int64_t wave_count = static_cast<int64_t>(gridDim.x) * kFeatWaves;
    for (int64_t pair = wave_global; pair < a.n; pair += wave_count) {
        const int64_t a0 = a.a_off[pair], b0 = a.b_off[pair];
        const int la = static_cast<int>(a.a_off[pair + 1] - a0), lb = static_cast<int>(a.b_off[pair + 1] - b0);
        wave_sync();
        // strings up to kReconCap-1 / 255 chars: a -> recon buffer, b -> t buffer
        for (int i = lane; i < la; i += 64) w.recon[i] = a.a_chars[a0 + i];
        for (int i = lane; i < lb; i += 64) w.t[i] = a.b_chars[b0 + i];
        wave_sync();
        uint8_t ratio;
        if (a.method == 0) {
            const bool small_alphabet = codes_below_64(w.recon, la, lane) && codes_below_64(w.t, lb, lane);
            ratio = levenshtein_wave(w, w.recon, la, w.t, lb, lane, small_alphabet);
        } else {
            ratio = levenshtein_literal(w, w.recon, la, w.t, lb, lane);
        }
        if (lane == 0) a.out[pair] = ratio;
    }
}

// ---- next row f-1: Prediction._find_close_matches (doppelspeller/predict.py:140-183) ---------------------------------
// ratio(x, y) = int(round(python-Levenshtein ratio * 100)) (common.py:161-162) behind a length pre-filter and a
// token-sort fallback (predict.py:147-156, common.py:165-167).  python-Levenshtein is a third-party C extension that is
// not part of the reference tree: its published definition (ratio = (lensum - ldist)/lensum, substitution cost 2,
// i.e. 2*LCS/lensum) is restated; parity is pinned against the CPU restatement used by the tests only.

// True LCS length of two LDS strings (no uint8 wrap-around semantics here): bit-parallel when the shorter string
// fits 64 bits and the alphabet is small, else an anti-diagonal DP with three uint8 diagonals (LCS <= 255).
__device__ int lcs_wave(WaveScratch &w, const uint8_t *a, int la, const uint8_t *b, int lb, int lane)
{
    if (la > lb) {
        const uint8_t *ts = a; a = b; b = ts;
        const int tl = la; la = lb; lb = tl;
    }
    if (la == 0) return 0;
    if (la <= 64 && codes_below_64(a, la, lane) && codes_below_64(b, lb, lane)) {
        build_masks(w, a, la, lane);
        return lcs_bitparallel(w, b, lb, la);
    }
    for (int d = 0; d <= la + lb; ++d) {
        uint8_t *current = w.diag[d % 3];
        const uint8_t *previous = w.diag[(d + 2) % 3];
        const uint8_t *before = w.diag[(d + 1) % 3];
        const int x_low = d > lb ? d - lb : 0;
        const int x_high = d < la ? d : la;
        for (int x = x_low + lane; x <= x_high; x += 64) {
            const int y = d - x;
            int value = 0;
            if (x > 0 && y > 0)
                value = a[x - 1] == b[y - 1] ? before[x - 1] + 1 : max(previous[x - 1], previous[x]);
            current[x] = static_cast<uint8_t>(value);
        }
        wave_sync();
    }
    return w.diag[(la + lb) % 3][la];
}

__device__ __forceinline__ int rounded_ratio(int lcs, int total)  // common.py:162  int(round(ratio * 100))
{
    if (total == 0) return 100;
    return __double2int_rn((static_cast<double>(2 * lcs) / static_cast<double>(total)) * 100.0);
}

// ' '.join(sorted(text.split())) (common.py:166) of an LDS string into `out`; returns the new length.
__device__ int token_sort_wave(WaveScratch &w, const uint8_t *text, int n, uint8_t space, const uint8_t *sort_key,
                               uint8_t *out, int lane)
{
    // word starts and lengths (up to 128 words in 255 characters)
    int words = 0;
    for (int base = 0; base < n; base += 64) {
        const int i = base + lane;
        const bool starts = i < n && text[i] != space && (i == 0 || text[i - 1] == space);
        const unsigned long long votes = __ballot(starts);
        if (starts) {
            int length = 1;
            while (i + length < n && text[i + length] != space) ++length;
            const int slot = words + __popcll(votes & ((1ull << lane) - 1ull));
            w.token_begin[slot] = static_cast<uint8_t>(i);
            w.token_len[slot] = static_cast<uint8_t>(length);
        }
        words += __popcll(votes);
    }
    wave_sync();
    int total = 0;
    for (int base = 0; base < words; base += 64) {  // lane = word; rank by counting the words that sort before it
        const int i = base + lane;
        int offset = 0, rank = 0, my_begin = 0, my_length = 0;
        if (i < words) {
            my_begin = w.token_begin[i];
            my_length = w.token_len[i];
            for (int j = 0; j < words; ++j) {
                if (j == i) continue;
                const int begin = w.token_begin[j], length = w.token_len[j];
                const int common = min(length, my_length);
                int compare = 0;
                for (int c = 0; c < common && compare == 0; ++c)
                    compare = static_cast<int>(sort_key[text[begin + c]]) - static_cast<int>(sort_key[text[my_begin + c]]);
                if (compare == 0) compare = length - my_length;
                if (compare < 0 || (compare == 0 && j < i)) {  // word j comes first (stable)
                    offset += length + 1;
                    ++rank;
                }
            }
            if (rank > 0) out[offset - 1] = space;
            for (int c = 0; c < my_length; ++c) out[offset + c] = text[my_begin + c];
        }
        (void)rank;
    }
    for (int j = 0; j < words; ++j) total += w.token_len[j] + 1;
    wave_sync();
    return total > 0 ? total - 1 : 0;
}

struct CloseArgs {
    const uint8_t *q_enc;
    const uint8_t *q_len;
    const uint8_t *t_enc;
    const uint8_t *t_len;
    const int32_t *pair_q;  // nullable
    const int32_t *pair_t;
    const uint8_t *sort_key;  // [256] code -> character order of Python's sorted()
    uint8_t *ratios;          // [n]
    int32_t *best_row;        // [n / k] or null
    int64_t q_stride, t_stride, n_q, n_t, n, q_first;
    int32_t k;
    int32_t threshold;
    uint8_t space_code;
};

// Prediction._get_levenshtein_ratio (predict.py:147-156) for every pair; one wavefront per pair.
__global__ __launch_bounds__(kFeatWaves * 64) void ds_close_ratio_kernel(CloseArgs a)
{
    __shared__ WaveScratch scratch[kFeatWaves];
    __shared__ uint8_t sort_key[256];
    sort_key[threadIdx.x] = a.sort_key[threadIdx.x];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    WaveScratch &w = scratch[threadIdx.x >> 6];
    const int64_t wave_global = static_cast<int64_t>(blockIdx.x) * kFeatWaves + (threadIdx.x >> 6);
    const int64_t wave_count = static_cast<int64_t>(gridDim.x) * kFeatWaves;
    for (int64_t pair = wave_global; pair < a.n; pair += wave_count) {
        const int64_t qi = a.pair_q ? a.pair_q[pair] : (a.k > 0 ? a.q_first + pair / a.k : pair);
        const int64_t ti = a.pair_t[pair];
        int ratio = 0;
        if (qi >= 0 && qi < a.n_q && ti >= 0 && ti < a.n_t) {
            const int lx = a.q_len[qi], ly = a.t_len[ti];
            const int total = lx + ly, delta = lx > ly ? lx - ly : ly - lx;
            // predict.py:141-151 length pre-filter, float64 in source order
            const bool may_match =
                total > 0 && !((static_cast<double>(total - delta) / static_cast<double>(total)) * 100.0 <
                               static_cast<double>(a.threshold));
            if (may_match) {
                wave_sync();
                for (int i = lane; i < lx; i += 64) w.q[i] = a.q_enc[qi * a.q_stride + i];
                for (int i = lane; i < ly; i += 64) w.t[i] = a.t_enc[ti * a.t_stride + i];
                wave_sync();
                ratio = rounded_ratio(lcs_wave(w, w.q, lx, w.t, ly, lane), total);               // predict.py:153
                if (ratio <= a.threshold) {                                                          // :154-155
                    const int sx = token_sort_wave(w, w.q, lx, a.space_code, sort_key, w.recon, lane);
                    const int sy = token_sort_wave(w, w.t, ly, a.space_code, sort_key, w.qw, lane);
                    ratio = rounded_ratio(lcs_wave(w, w.recon, sx, w.qw, sy, lane), sx + sy);
                }
            }
        }
        if (lane == 0) a.ratios[pair] = static_cast<uint8_t>(ratio);
    }
}

// predict.py:172-176: per query the candidates with ratio > threshold, the maximum among them, and the query is
// matched only when exactly one candidate reaches it (_remove_duplicated_matches drops the others).
__global__ void ds_close_best_kernel(CloseArgs a)
{
    const int64_t q = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (q >= a.n / a.k) return;
    int best = a.threshold, count = 0, where = -1;
    for (int j = 0; j < a.k; ++j) {
        const int ratio = a.ratios[q * a.k + j];
        if (ratio > best) { best = ratio; count = 1; where = j; }
        else if (ratio == best && where >= 0) ++count;
    }
    a.best_row[q] = (where >= 0 && count == 1) ? a.pair_t[q * a.k + where] : -1;
}


// ---- host-pointer entry point: packed, pinned, chunked (see ds_construct_features below) -------------------------------
constexpr int64_t kStageChunk = 16384;                                  // pairs per chunk
constexpr int64_t kStageInBytes = kStageChunk * (2 + 8 + 4 * DS_WORDS + 2 * DS_MAX_CHARS);
constexpr int kStageMaxSlots = 8;

struct FeatureInputs {
    const uint8_t *q_len, *t_len, *q_enc, *t_enc;
    const uint32_t *t_counts;
    uint8_t space_code;
    uint32_t n_truth;
    int64_t n, stride;
    float *out;
};

struct FeatureSlot {
    hipStream_t stream = nullptr;
    unsigned char *h_in = nullptr, *d_in = nullptr;   // h_*: pinned host memory
    float *h_out = nullptr, *d_out = nullptr;
};

struct FeatureStaging {
    std::mutex mutex;   // one call at a time per device (the handles are documented as not thread-safe; this makes it safe)
    int device = -1;
    std::vector<FeatureSlot> slots;
    int ensure(int wanted_device, int wanted_slots)
    {
        if (device != wanted_device) release();
        device = wanted_device;
        while (static_cast<int>(slots.size()) < wanted_slots) {
            const int added = add_slot();
            if (added != DS_OK) {   // a half-built slot must never be handed out by a later call: start over next time
                release();
                return added;
            }
        }
        return DS_OK;
    }
    int add_slot()
    {
        slots.emplace_back();   // registered first: release() frees whatever of it got allocated
        FeatureSlot &mine = slots.back();
        DS_HIP(hipStreamCreateWithFlags(&mine.stream, hipStreamNonBlocking));
        DS_HIP(hipHostMalloc(reinterpret_cast<void **>(&mine.h_in), kStageInBytes, hipHostMallocDefault));
        DS_HIP(hipHostMalloc(reinterpret_cast<void **>(&mine.h_out), kStageChunk * DS_FEATURES_COUNT * sizeof(float), hipHostMallocDefault));
        DS_HIP(hipMalloc(reinterpret_cast<void **>(&mine.d_in), kStageInBytes));
        DS_HIP(hipMalloc(reinterpret_cast<void **>(&mine.d_out), kStageChunk * DS_FEATURES_COUNT * sizeof(float)));
        return DS_OK;
    }
    void release()
    {
        for (FeatureSlot &slot : slots) {
            if (slot.h_in) (void)hipHostFree(slot.h_in);
            if (slot.h_out) (void)hipHostFree(slot.h_out);
            if (slot.d_in) (void)hipFree(slot.d_in);
            if (slot.d_out) (void)hipFree(slot.d_out);
            if (slot.stream) (void)hipStreamDestroy(slot.stream);
        }
        slots.clear();
        device = -1;
    }
};

// One staging set per DEVICE: callers that alternate devices within a process neither serialise on one mutex nor free and
// re-allocate ~110 MB of pinned buffers at every switch.
static FeatureStaging &feature_staging(int device)
{
    static std::mutex mutex;
    static std::map<int, FeatureStaging *> *per_device = new std::map<int, FeatureStaging *>();   // never destroyed: no HIP calls from static destructors
    std::lock_guard<std::mutex> guard(mutex);
    FeatureStaging *&staging = (*per_device)[device];
    if (staging == nullptr) staging = new FeatureStaging();
    return *staging;
}

static int launch_features(const FeatureArgs &args, int device, hipStream_t stream);

// one chunk [first, first + m) through one slot; returns a status
static int stage_chunk(const FeatureInputs &in, FeatureSlot &slot, int device, int64_t first, int64_t m)
{
    const int64_t m4 = (m + 3) & ~int64_t(3);
    unsigned char *h = slot.h_in;
    uint8_t *h_qlen = h, *h_tlen = h + m4;
    uint32_t *h_qoff = reinterpret_cast<uint32_t *>(h + 2 * m4), *h_toff = reinterpret_cast<uint32_t *>(h + 6 * m4);
    uint32_t *h_counts = reinterpret_cast<uint32_t *>(h + 10 * m4);
    const int64_t chars_at = 10 * m4 + 4 * DS_WORDS * m4;
    unsigned char *h_chars = h + chars_at;
    std::memcpy(h_qlen, in.q_len + first, static_cast<size_t>(m));
    std::memcpy(h_tlen, in.t_len + first, static_cast<size_t>(m));
    std::memcpy(h_counts, in.t_counts + first * DS_WORDS, static_cast<size_t>(m) * DS_WORDS * sizeof(uint32_t));
    uint32_t used = 0;
    for (int64_t i = 0; i < m; ++i) {
        const uint32_t lq = h_qlen[i], lt = h_tlen[i];
        h_qoff[i] = used;
        std::memcpy(h_chars + used, in.q_enc + (first + i) * in.stride, lq);
        used += lq;
        h_toff[i] = used;
        std::memcpy(h_chars + used, in.t_enc + (first + i) * in.stride, lt);
        used += lt;
    }
    DS_HIP(hipMemcpyAsync(slot.d_in, h, static_cast<size_t>(chars_at) + used, hipMemcpyHostToDevice, slot.stream));
    FeatureArgs args{};
    args.q_enc = slot.d_in + chars_at; args.t_enc = slot.d_in + chars_at;
    args.q_len = slot.d_in; args.t_len = slot.d_in + m4;
    args.q_off = reinterpret_cast<const uint32_t *>(slot.d_in + 2 * m4);
    args.t_off = reinterpret_cast<const uint32_t *>(slot.d_in + 6 * m4);
    args.t_counts = reinterpret_cast<const uint32_t *>(slot.d_in + 10 * m4);
    args.pair_q = nullptr; args.pair_t = nullptr; args.out = slot.d_out; args.t_records = nullptr; args.unit_queue = nullptr; args.unit_pairs = 2;
    args.q_stride = 0; args.t_stride = 0; args.n_q = m; args.n_t = m; args.n = m; args.q_first = 0; args.k = 0;
    args.n_truth = in.n_truth; args.space_code = in.space_code;
    const int status = launch_features(args, device, slot.stream);
    if (status != DS_OK) return status;
    const size_t out_bytes = static_cast<size_t>(m) * DS_FEATURES_COUNT * sizeof(float);
    DS_HIP(hipMemcpyAsync(slot.h_out, slot.d_out, out_bytes, hipMemcpyDeviceToHost, slot.stream));
    DS_HIP(hipStreamSynchronize(slot.stream));
    std::memcpy(in.out + first * DS_FEATURES_COUNT, slot.h_out, out_bytes);
    return DS_OK;
}

static int staged_features(const FeatureInputs &in, int device)
{
    FeatureStaging &staging = feature_staging(device);
    std::lock_guard<std::mutex> guard(staging.mutex);
    const int64_t chunks = (in.n + kStageChunk - 1) / kStageChunk;
    const int workers = static_cast<int>(std::min<int64_t>(chunks, std::min(host_threads(), kStageMaxSlots)));
    const int ensured = staging.ensure(device, workers);
    if (ensured != DS_OK) return ensured;
    std::atomic<int64_t> next{0};
    std::atomic<int> failed{DS_OK};
    FirstError error;   // ds_last_error() is thread-local: a worker's message is carried over to the calling thread
    auto work = [&](int worker) {
        if (hipSetDevice(device) != hipSuccess) {
            failed.store(DS_E_HIP);
            error.raise("ds_construct_features: hipSetDevice(%d) failed in a staging thread", device);
            return;
        }
        for (;;) {
            const int64_t chunk = next.fetch_add(1, std::memory_order_relaxed);
            if (chunk >= chunks || failed.load(std::memory_order_relaxed) != DS_OK) break;
            const int64_t first = chunk * kStageChunk;
            const int status = stage_chunk(in, staging.slots[static_cast<size_t>(worker)], device, first,
                                           std::min(kStageChunk, in.n - first));
            if (status != DS_OK) {
                // nothing of this slot may still be in flight when a later call packs its pinned buffers again
                (void)hipStreamSynchronize(staging.slots[static_cast<size_t>(worker)].stream);
                failed.store(status);
                error.raise("%s", ds_last_error());
                break;
            }
        }
    };
    if (workers <= 1) {
        work(0);
    } else {
        std::vector<std::thread> pool;
        for (int t = 0; t < workers; ++t) pool.emplace_back(work, t);
        for (std::thread &thread : pool) thread.join();
    }
    if (failed.load() != DS_OK && error.failed()) set_error("%s", error.message());
    return failed.load();
}

static int launch_features(const FeatureArgs &args, int device, hipStream_t stream)
{
    if (args.n == 0) return DS_OK;
    {   // the ratio table of this device: once per process, complete before any stream's first kernel reads it
        static std::mutex mutex;
        static bool ready[64] = {false};
        std::lock_guard<std::mutex> guard(mutex);
        if (device >= 0 && device < 64 && !ready[device]) {
            hipLaunchKernelGGL(ds_ratio_table_kernel, dim3((kRatioEntries + 255) / 256), dim3(256), 0, stream);
            DS_HIP(hipGetLastError());
            DS_HIP(hipStreamSynchronize(stream));
            ready[device] = true;
        }
    }
    const int64_t units = (args.n + args.unit_pairs - 1) / args.unit_pairs;
    int grid;
    if (args.unit_queue != nullptr) {
        // as many workgroups as the device holds at once (5 per CU: registers and 31 KiB of LDS each); the queue does the rest
        DS_HIP(hipMemsetAsync(args.unit_queue, 0, sizeof(int32_t), stream));
        grid = static_cast<int>(std::min<int64_t>((units + kFeatKernelWaves - 1) / kFeatKernelWaves, 256 * DS_FEAT_MIN_WAVES));
    } else {
        // Persistent waves, every one with the SAME number of units (+- 1): 32,768 waves at most, and as many fewer as keep the
        // last round full (100,000 units on 65,536 waves would take two rounds with a third of the chip idle in the second).
        const int64_t most_waves = 256 * 32 * kFeatKernelWaves;
        const int64_t rounds = (units + most_waves - 1) / most_waves;
        const int64_t waves = (units + rounds - 1) / rounds;
        grid = static_cast<int>((waves + kFeatKernelWaves - 1) / kFeatKernelWaves);
    }
    if (args.t_records != nullptr)
        hipLaunchKernelGGL(ds_construct_features_kernel<true>, dim3(grid), dim3(kFeatKernelWaves * 64), 0, stream, args);
    else
        hipLaunchKernelGGL(ds_construct_features_kernel<false>, dim3(grid), dim3(kFeatKernelWaves * 64), 0, stream, args);
    DS_HIP(hipGetLastError());
    return DS_OK;
}

// The truth table's records for (n_truth, space code): built on `stream` in front of the first launch that names them.
static int ensure_truth_records(ds_titles *truth, uint32_t n_truth, uint8_t space_code, hipStream_t stream)
{
    if (!truth->records_enabled) return DS_OK;
    const size_t bytes = static_cast<size_t>(truth->n) * sizeof(TruthRecord);
    if (truth->records.count == bytes && truth->records_n_truth == n_truth && truth->records_space == space_code) return DS_OK;
    if (truth->records.count != bytes) {
        const int allocated = truth->records.allocate(bytes);
        if (allocated != DS_OK) {   // not enough HBM for them: the kernel derives everything per pair, as before
            truth->records.release();
            truth->records_enabled = false;
            (void)hipGetLastError();
            return DS_OK;
        }
    }
    hipLaunchKernelGGL(ds_truth_records_kernel, dim3(static_cast<unsigned>((truth->n + 255) / 256)), dim3(256), 0, stream,
                       truth->enc.ptr, truth->stride, truth->len.ptr, truth->counts.ptr, truth->n, n_truth, space_code,
                       reinterpret_cast<TruthRecord *>(truth->records.ptr));
    DS_HIP(hipGetLastError());
    DS_HIP(hipStreamSynchronize(stream));   // once per (n_truth, space code): complete before a launch on ANY stream reads them
    truth->records_n_truth = n_truth;
    truth->records_space = space_code;
    return DS_OK;
}

// consecutive pairs one wave works through (two at a time): the k candidates of a query (k <= 16), a divisor of k between 8 and 16, or 10
#ifndef DS_FEAT_UNIT
#define DS_FEAT_UNIT 16
#endif
static int32_t pairs_per_unit(int32_t k)
{
    if (k <= 0) return 8;
    if (k <= DS_FEAT_UNIT) return k;
    for (int32_t d = DS_FEAT_UNIT; d >= (DS_FEAT_UNIT + 1) / 2; --d)
        if (k % d == 0) return d;
    return DS_FEAT_UNIT < 10 ? DS_FEAT_UNIT : 10;
}

}  // namespace ds

extern "C" {

// The reference's own call (predict.py:216-219, feature_engineering.py:358-361) hands over HOST arrays in the padded layout
// of encode_title: 255 + 255 + 60 + 2 bytes per pair, of which ~110 carry information.  The pairs are cut into chunks of
// kStageChunk; a few host threads take chunks from a counter, each with a slot of its own (pinned staging buffers, device
// buffers, a stream): pack the chunk's lengths, word counts and the titles' OWN bytes (row offsets instead of a stride)
// into the pinned buffer, one asynchronous copy in, the kernel, one asynchronous copy out, a threaded memcpy into the
// caller's `out`.  Chunks of different threads overlap on the GPU and on PCIe (H2D / kernel / D2H on separate streams).
// The slots live as long as the process (hipHostMalloc is slow: paid by the first call).
int ds_construct_features(const uint8_t *q_len, const uint8_t *t_len, const uint8_t *q_enc, const uint8_t *t_enc,
                          const uint32_t *t_word_counts, uint8_t space_code, uint32_t n_truth, int64_t n,
                          int64_t stride, int device, float *out)
{
    DS_REQUIRE(n >= 0, "ds_construct_features: negative pair count");
    if (n == 0) return DS_OK;
    DS_REQUIRE(q_len && t_len && q_enc && t_enc && t_word_counts && out, "ds_construct_features: null pointer");
    DS_REQUIRE(stride >= 1, "ds_construct_features: stride must be positive");
    for (int64_t i = 0; i < n; ++i)
        DS_REQUIRE(q_len[i] <= stride && t_len[i] <= stride,
                   "ds_construct_features: length of pair %lld exceeds the row stride %lld", (long long)i,
                   (long long)stride);
    DS_HIP(hipSetDevice(device));
    ds::FeatureInputs inputs{q_len, t_len, q_enc, t_enc, t_word_counts, space_code, n_truth, n, stride, out};
    return ds::staged_features(inputs, device);
}

int ds_titles_create(const uint8_t *enc, int64_t stride, const uint8_t *len, const uint32_t *word_counts, int64_t n,
                     int device, ds_titles **out)
{
    DS_REQUIRE(out != nullptr, "ds_titles_create: out is null");
    *out = nullptr;
    DS_REQUIRE(enc && len && n > 0 && stride >= 1, "ds_titles_create: bad arguments");
    for (int64_t i = 0; i < n; ++i)
        DS_REQUIRE(len[i] <= stride, "ds_titles_create: length of row %lld exceeds the stride", (long long)i);
    DS_HIP(hipSetDevice(device));
    ds_titles *titles = new ds_titles();
    titles->device = device;
    titles->n = n;
    titles->stride = stride;
    titles->has_counts = word_counts != nullptr;
    int status = titles->enc.upload(enc, static_cast<size_t>(n * stride));
    if (status == DS_OK) status = titles->len.upload(len, static_cast<size_t>(n));
    if (status == DS_OK && word_counts) status = titles->counts.upload(word_counts, static_cast<size_t>(n) * DS_WORDS);
    if (status != DS_OK) {
        delete titles;
        return status;
    }
    *out = titles;
    return DS_OK;
}

int ds_titles_option(ds_titles *titles, const char *name, int64_t value)
{
    DS_REQUIRE(titles && name, "ds_titles_option: null argument");
    if (std::strcmp(name, "truth_records") == 0) {   // 160 bytes of HBM per truth row for what depends on the truth title alone
        titles->records_enabled = value != 0;
        if (!titles->records_enabled) {
            DS_HIP(hipSetDevice(titles->device));
            DS_HIP(hipDeviceSynchronize());
            titles->records.release();
        }
        return DS_OK;
    }
    ds::set_error("ds_titles_option: unknown option '%s'", name);
    return DS_E_ARG;
}

void ds_titles_destroy(ds_titles *titles)
{
    if (!titles) return;
    (void)hipSetDevice(titles->device);
    delete titles;
}

int ds_construct_features_indexed_device(ds_titles *queries, ds_titles *truth, const int32_t *d_pair_q,
                                         const int32_t *d_pair_t, int64_t q_first, int32_t k, uint8_t space_code,
                                         uint32_t n_truth, int64_t n, float *d_out, void *stream)
{
    DS_REQUIRE(queries && truth, "ds_construct_features_indexed: null table");
    DS_REQUIRE(truth->has_counts, "ds_construct_features_indexed: the truth table has no word counts");
    DS_REQUIRE(queries->device == truth->device, "ds_construct_features_indexed: tables on different devices");
    DS_REQUIRE(n >= 0, "ds_construct_features_indexed: negative pair count");
    if (n == 0) return DS_OK;
    DS_REQUIRE(d_pair_t && d_out, "ds_construct_features_indexed: null pointer");
    DS_REQUIRE(d_pair_q || k > 0, "ds_construct_features_indexed: need pair_q or k > 0");
    DS_HIP(hipSetDevice(truth->device));
    ds::FeatureArgs args{};
    args.q_enc = queries->enc.ptr; args.q_len = queries->len.ptr; args.t_enc = truth->enc.ptr;
    args.t_len = truth->len.ptr; args.t_counts = truth->counts.ptr; args.pair_q = d_pair_q; args.pair_t = d_pair_t;
    args.out = d_out; args.q_stride = queries->stride; args.t_stride = truth->stride; args.n_q = queries->n;
    args.n_t = truth->n; args.n = n; args.q_first = q_first; args.k = d_pair_q ? 0 : k; args.n_truth = n_truth;
    args.space_code = space_code;
    // a caller that switches between streams orders them itself (the records are written once per (n_truth, space code))
    const int ensured = ds::ensure_truth_records(truth, n_truth, space_code, static_cast<hipStream_t>(stream));
    if (ensured != DS_OK) return ensured;
    args.t_records = truth->records_enabled ? truth->records.ptr : nullptr;
    // the launch's work-queue head: one of 64 words of the truth table, taken in turn -- launches that overlap on the device (other
    // streams) never share a head (the CALLS are still one at a time per table: the handles are not thread-safe)
    constexpr size_t kQueueHeads = 64;
    if (truth->unit_queue.count == 0 && truth->unit_queue.allocate(kQueueHeads) != DS_OK) return DS_E_HIP;
    args.unit_queue = truth->unit_queue.ptr + (truth->unit_queue_next++ % kQueueHeads);
    args.unit_pairs = ds::pairs_per_unit(d_pair_q ? 0 : k);
    return ds::launch_features(args, truth->device, static_cast<hipStream_t>(stream));
}

int ds_construct_features_indexed(ds_titles *queries, ds_titles *truth, const int32_t *pair_q, const int32_t *pair_t,
                                  uint8_t space_code, uint32_t n_truth, int64_t n, float *out)
{
    DS_REQUIRE(queries && truth, "ds_construct_features_indexed: null table");
    DS_REQUIRE(n >= 0, "ds_construct_features_indexed: negative pair count");
    if (n == 0) return DS_OK;
    DS_REQUIRE(pair_q && pair_t && out, "ds_construct_features_indexed: null pointer");
    for (int64_t i = 0; i < n; ++i)
        DS_REQUIRE(pair_q[i] >= 0 && pair_q[i] < queries->n && pair_t[i] >= 0 && pair_t[i] < truth->n,
                   "ds_construct_features_indexed: pair %lld indexes outside the tables", (long long)i);
    DS_HIP(hipSetDevice(truth->device));
    ds::DeviceBuffer<int32_t> d_q, d_t;
    ds::DeviceBuffer<float> d_out;
    int status = d_q.upload(pair_q, static_cast<size_t>(n));
    if (status == DS_OK) status = d_t.upload(pair_t, static_cast<size_t>(n));
    if (status == DS_OK) status = d_out.allocate(static_cast<size_t>(n) * DS_FEATURES_COUNT);
    if (status != DS_OK) return status;
    status = ds_construct_features_indexed_device(queries, truth, d_q.ptr, d_t.ptr, 0, 0, space_code, n_truth, n,
                                                  d_out.ptr, nullptr);
    if (status != DS_OK) return status;
    DS_HIP(hipMemcpy(out, d_out.ptr, d_out.bytes(), hipMemcpyDeviceToHost));
    return DS_OK;
}

int ds_levenshtein_ratio_batch(const uint8_t *a_chars, const int64_t *a_off, const uint8_t *b_chars,
                               const int64_t *b_off, int64_t n, int method, int device, uint8_t *out)
{
    DS_REQUIRE(n >= 0, "ds_levenshtein_ratio_batch: negative pair count");
    if (n == 0) return DS_OK;
    DS_REQUIRE(a_off && b_off && out, "ds_levenshtein_ratio_batch: null pointer");
    DS_REQUIRE(method == 0 || method == 1, "ds_levenshtein_ratio_batch: method must be 0 or 1");
    for (int64_t i = 0; i < n; ++i) {
        const int64_t la = a_off[i + 1] - a_off[i], lb = b_off[i + 1] - b_off[i];
        DS_REQUIRE(la >= 0 && la < ds::kReconCap && lb >= 0 && lb <= DS_MAX_CHARS,
                   "ds_levenshtein_ratio_batch: pair %lld: lengths (%lld, %lld) outside (<=287, <=255)",
                   (long long)i, (long long)la, (long long)lb);
    }
    DS_HIP(hipSetDevice(device));
    ds::DeviceBuffer<uint8_t> d_a, d_b, d_out;
    ds::DeviceBuffer<int64_t> d_aoff, d_boff;
    int status = d_a.upload(a_chars, static_cast<size_t>(a_off[n] > 0 ? a_off[n] : 0));
    if (status == DS_OK && a_off[n] == 0) status = d_a.allocate(1);
    if (status == DS_OK) status = d_b.upload(b_chars, static_cast<size_t>(b_off[n] > 0 ? b_off[n] : 0));
    if (status == DS_OK && b_off[n] == 0) status = d_b.allocate(1);
    if (status == DS_OK) status = d_aoff.upload(a_off, static_cast<size_t>(n + 1));
    if (status == DS_OK) status = d_boff.upload(b_off, static_cast<size_t>(n + 1));
    if (status == DS_OK) status = d_out.allocate(static_cast<size_t>(n));
    if (status != DS_OK) return status;
    ds::LevArgs args{d_a.ptr, d_aoff.ptr, d_b.ptr, d_boff.ptr, d_out.ptr, n, method};
    const int64_t blocks_needed = (n + ds::kFeatWaves - 1) / ds::kFeatWaves;
    const int grid = static_cast<int>(std::min<int64_t>(blocks_needed, 256 * 32));
    hipLaunchKernelGGL(ds::ds_levenshtein_kernel, dim3(grid), dim3(ds::kFeatWaves * 64), 0, nullptr, args);
    DS_HIP(hipGetLastError());
    DS_HIP(hipMemcpy(out, d_out.ptr, static_cast<size_t>(n), hipMemcpyDeviceToHost));
    return DS_OK;
}

int ds_levenshtein_ratio(const uint8_t *a, int la, const uint8_t *b, int lb)
{
    DS_REQUIRE(la >= 0 && lb >= 0 && (la == 0 || a) && (lb == 0 || b), "ds_levenshtein_ratio: bad arguments");
    const int64_t a_off[2] = {0, la}, b_off[2] = {0, lb};
    const uint8_t none = 0;
    uint8_t ratio = 0;
    const int status = ds_levenshtein_ratio_batch(la ? a : &none, a_off, lb ? b : &none, b_off, 1, 0, 0, &ratio);
    return status != DS_OK ? status : static_cast<int>(ratio);
}

int ds_close_matches_device(ds_titles *queries, ds_titles *truth, const int32_t *d_pair_t, int64_t q_first, int32_t k,
                            int64_t n_queries, uint8_t space_code, const uint8_t *d_sort_key, int32_t threshold,
                            uint8_t *d_ratios, int32_t *d_best_row, void *stream)
{
    DS_REQUIRE(queries && truth, "ds_close_matches: null table");
    DS_REQUIRE(queries->device == truth->device, "ds_close_matches: tables on different devices");
    DS_REQUIRE(k >= 1 && n_queries >= 0, "ds_close_matches: bad k / query count");
    if (n_queries == 0) return DS_OK;
    DS_REQUIRE(d_pair_t && d_sort_key && d_ratios, "ds_close_matches: null pointer");
    DS_HIP(hipSetDevice(truth->device));
    ds::CloseArgs args{};
    args.q_enc = queries->enc.ptr; args.q_len = queries->len.ptr; args.t_enc = truth->enc.ptr;
    args.t_len = truth->len.ptr; args.pair_q = nullptr; args.pair_t = d_pair_t; args.sort_key = d_sort_key;
    args.ratios = d_ratios; args.best_row = d_best_row; args.q_stride = queries->stride;
    args.t_stride = truth->stride; args.n_q = queries->n; args.n_t = truth->n; args.n = n_queries * k;
    args.q_first = q_first; args.k = k; args.threshold = threshold; args.space_code = space_code;
    const int64_t blocks_needed = (args.n + ds::kFeatWaves - 1) / ds::kFeatWaves;
    const int grid = static_cast<int>(std::min<int64_t>(blocks_needed, 256 * 32));
    hipLaunchKernelGGL(ds::ds_close_ratio_kernel, dim3(grid), dim3(ds::kFeatWaves * 64), 0,
                       static_cast<hipStream_t>(stream), args);
    DS_HIP(hipGetLastError());
    if (d_best_row) {
        hipLaunchKernelGGL(ds::ds_close_best_kernel, dim3(static_cast<unsigned>((n_queries + 255) / 256)), dim3(256), 0,
                           static_cast<hipStream_t>(stream), args);
        DS_HIP(hipGetLastError());
    }
    return DS_OK;
}

int ds_close_matches(ds_titles *queries, ds_titles *truth, const int32_t *pair_t, int32_t k, int64_t n_queries,
                     uint8_t space_code, const uint8_t *sort_key, int32_t threshold, uint8_t *ratios,
                     int32_t *best_row)
{
    DS_REQUIRE(queries && truth, "ds_close_matches: null table");
    DS_REQUIRE(k >= 1 && n_queries >= 0, "ds_close_matches: bad k / query count");
    if (n_queries == 0) return DS_OK;
    DS_REQUIRE(pair_t && sort_key && ratios && best_row, "ds_close_matches: null pointer");
    DS_REQUIRE(n_queries <= queries->n, "ds_close_matches: more queries than rows in the query table");
    DS_HIP(hipSetDevice(truth->device));
    const size_t n = static_cast<size_t>(n_queries) * static_cast<size_t>(k);
    ds::DeviceBuffer<int32_t> d_t, d_best;
    ds::DeviceBuffer<uint8_t> d_key, d_ratios;
    int status = d_t.upload(pair_t, n);
    if (status == DS_OK) status = d_key.upload(sort_key, 256);
    if (status == DS_OK) status = d_ratios.allocate(n);
    if (status == DS_OK) status = d_best.allocate(static_cast<size_t>(n_queries));
    if (status != DS_OK) return status;
    status = ds_close_matches_device(queries, truth, d_t.ptr, 0, k, n_queries, space_code, d_key.ptr, threshold,
                                     d_ratios.ptr, d_best.ptr, nullptr);
    if (status != DS_OK) return status;
    DS_HIP(hipMemcpy(ratios, d_ratios.ptr, n, hipMemcpyDeviceToHost));
    DS_HIP(hipMemcpy(best_row, d_best.ptr, static_cast<size_t>(n_queries) * sizeof(int32_t), hipMemcpyDeviceToHost));
    return DS_OK;
}

}  // extern "C"
