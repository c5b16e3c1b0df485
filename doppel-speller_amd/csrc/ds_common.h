// Internal helpers shared by the translation units of libdoppel_amd.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "doppel_amd.h"

namespace ds {

// ---- two geometries of the Jaccard kernels (chosen per index by ds_index_create) ---------------
//   wide    2 workgroups of 512 threads per CU, tiles of 28672 truth rows  (large truth sets: fewer tiles per query)
//   narrow  4 workgroups of 256 threads per CU, tiles of 12288 truth rows  (up to kNarrowMaxTruth rows: four independent
//           queries per CU hide each other's barriers and load latencies, and the dense first tile is smaller)
// ds_jaccard_impl.inc is compiled once per geometry (ds_jaccard_wide.hip / ds_jaccard_narrow.hip).
#ifndef DS_WIDE_TILE_ROWS
#define DS_WIDE_TILE_ROWS 28672
#endif
#ifndef DS_NARROW_TILE_ROWS
#define DS_NARROW_TILE_ROWS 12288
#endif
#ifndef DS_NARROW_MAX_TRUTH
#define DS_NARROW_MAX_TRUTH 5000000  // round 5 (profiles/r05_tuning.txt section 11), narrow / wide: 3M rows 39.0 / 41.4 ms, 5M 51.1 / 50.8, 7.5M 57.6 / 55.4, 10M 57.2 / 54.1, 20M 80.1 / 72.9
#endif
constexpr int kWideTileRows = DS_WIDE_TILE_ROWS, kNarrowTileRows = DS_NARROW_TILE_ROWS;
constexpr int64_t kNarrowMaxTruth = DS_NARROW_MAX_TRUTH;
constexpr int kDenseThreads = 1024;          // literal kernel: one 16-wave workgroup per CU, float32 score tile
constexpr int kMaxQueryColumns = 128;        // fast-path limit (example data: p99 50, max 96 tri-grams per title)
constexpr int kSignatureBits = 128;          // densest columns whose membership is kept as a per-row bit (uint4)
constexpr int kSignatureWords = kSignatureBits / 32;
constexpr int kRowRecordWords = 8;          // 32-byte row record: signature, sums32, duplicate rank
constexpr int kControlWords = 32;            // int32 control block in HBM (queue heads, counters)

// 8-bit lower bound of a positive float: 4 exponent bits (2^-3 .. 2^12) and 4 mantissa bits, truncated.
// decode(encode(x)) <= x for every x >= 0; code 0 decodes to 0; a row never gets the padding code 0xff (values from 7936 up share 0xfe).
inline uint32_t encode_sums8(float x)
{
    if (!(x >= 0.125f)) return 0u;
    uint32_t bits;
    memcpy(&bits, &x, sizeof(bits));
    const int exponent = static_cast<int>(bits >> 23) - 124;  // 2^-3 -> 0
    if (exponent > 15) return 0xfeu;  // 0xff is the padding entries' code (never reachable in the collect sweep)
    return std::min<uint32_t>(0xfeu, (static_cast<uint32_t>(exponent) << 4) | ((bits >> 19) & 0xfu));
}

// splitmix64 finaliser: a bijective 64-bit mixer (row-set hashes of the index build)
inline uint64_t mix64(uint64_t x)
{
    x ^= x >> 30;
    x *= 0xbf58476d1ce4e5b9ull;
    x ^= x >> 27;
    x *= 0x94d049bb133111ebull;
    x ^= x >> 31;
    return x;
}

enum QueryStatus : int32_t { kQueryDone = 0, kQuerySlow = 1, kQueryErrorTopN = 2, kQueryErrorArg = 3, kQueryErrorTies = 4, kQuerySlowFew = 5 };

void set_error(const char *format, ...);
void duplicate_ranks(const int64_t *rowptr, const int32_t *truth_idx, const float *sums32, int64_t V, int64_t N,
                     uint16_t *rank_out);
int hip_failed(hipError_t error, const char *what, const char *file, int line);

#define DS_HIP(call)                                                          \
    do {                                                                      \
        hipError_t ds_hip_error_ = (call);                                    \
        if (ds_hip_error_ != hipSuccess) return ::ds::hip_failed(ds_hip_error_, #call, __FILE__, __LINE__); \
    } while (0)

#define DS_REQUIRE(condition, ...)            \
    do {                                      \
        if (!(condition)) {                   \
            ::ds::set_error(__VA_ARGS__);     \
            return DS_E_ARG;                  \
        }                                     \
    } while (0)

template <typename T>
struct DeviceBuffer {
    T *ptr = nullptr;
    size_t count = 0;
    int allocate(size_t n)
    {
        release();
        if (n == 0) return DS_OK;
        DS_HIP(hipMalloc(reinterpret_cast<void **>(&ptr), n * sizeof(T)));
        count = n;  // only a buffer that exists has a size (a failed allocation leaves {nullptr, 0})
        return DS_OK;
    }
    int upload(const T *host, size_t n)
    {
        int status = allocate(n);
        if (status != DS_OK) return status;
        if (n) DS_HIP(hipMemcpy(ptr, host, n * sizeof(T), hipMemcpyHostToDevice));
        return DS_OK;
    }
    void release()
    {
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr;
        count = 0;
    }
    size_t bytes() const { return count * sizeof(T); }
    ~DeviceBuffer() { release(); }
    DeviceBuffer() = default;
    DeviceBuffer(const DeviceBuffer &) = delete;
    DeviceBuffer &operator=(const DeviceBuffer &) = delete;
};

}  // namespace ds

// The truth inverted index as it lives in HBM.
struct ds_index {
    int device = 0;
    int64_t n_truth = 0, n_columns = 0, nnz = 0, n_tiles = 0, n_quads = 0;
    int tile_rows = 0;                     // kWideTileRows or kNarrowTileRows: selects the kernels' geometry
    float sums_min = 0.f;
    ds::DeviceBuffer<uint32_t> col_ptr;    // [n_columns][n_tiles + 1], unit = quads of 4 postings (column-major)
    ds::DeviceBuffer<uint16_t> postings;   // [n_quads * 4] (parity << 15) | (tile-local row >> 1); per (column, tile): even rows, padding, odd rows, padding
    ds::DeviceBuffer<uint16_t> posting_sums; // [n_quads * 4] per posting: 8-bit lower bound of sums32[row] << 8 | signature bits 0..7
    ds::DeviceBuffer<float> idf32;         // [n_columns]
    ds::DeviceBuffer<float> sums32;        // [n_truth] in INTERNAL row order (rows sorted by sums32; word 6 of a row record = the caller's row)
    ds::DeviceBuffer<float> tile_sums_min; // [n_tiles] min(sums32) over the rows of each tile
    ds::DeviceBuffer<float> tile_sums_max; // [n_tiles] max(sums32) over the rows of each tile
    ds::DeviceBuffer<uint32_t> signature;  // [n_truth][8] one 32-byte record per row, ONE cache line per refined row: words 0..3 =
                                           // signature (bit g = row is in the posting list of the g-th densest column), word 4 =
                                           // sums32 bits, word 5 = duplicate rank (below), words 6..7 unused
    ds::DeviceBuffer<int8_t> sig_column;   // [n_columns] signature bit of a column, -1 for all but the 128 densest
    ds::DeviceBuffer<unsigned char> row_start;  // [n_truth + 1] forward index: row t's columns are row_cols[row_start[t] .. row_start[t + 1]);
                                                // uint32 while nnz < 2^32 (forward_wide_start = false), else int64
    ds::DeviceBuffer<unsigned char> row_cols;   // [nnz] ascending column ids per row, rows in internal order (the exact stage's input);
                                                // uint16 while n_columns <= 65536 (forward_wide_cols = false), else int32
    bool forward_wide_start = false, forward_wide_cols = false;
    // duplicate rank of a row (word 5 of its record): rows with the same column set and sums32 bits but a larger index (saturating)
    bool rows_sorted = false;              // internal row order ascends with sums32 (ds_index_create, DS_SORT_ROWS != 0)
    bool literal_only = false;             // idf32 / sums32 hold negative or non-finite values: the bounds of the fast kernel
                                           // do not apply, every query takes the literal kernel
    ds::DeviceBuffer<unsigned char> kernel_args;  // the fast kernel's argument block (read through the constant address space)
    ds::DeviceBuffer<int32_t> control;     // [16] work-queue head, slow-list length, error count, counters
    ds::DeviceBuffer<int32_t> status;      // per-query status of the last call (grown on demand)
    ds::DeviceBuffer<int32_t> slow_list;   // query ids routed to the exact dense kernel
    ds::DeviceBuffer<int32_t> take_order;  // the order in which the fast kernel's work queue hands out the queries of the last
                                           // call: most columns first (longest-processing-time-first keeps the launch's tail short)
    ds::DeviceBuffer<int32_t> order_bins;  // [2][136] counting sort of the queries by their number of columns: counts, cursors
    ds::DeviceBuffer<unsigned long long> phase;  // diagnostic phase timers (DS_PHASE_TIMERS=1)
    hipStream_t stream = nullptr;          // used by the host-pointer entry points
    int compute_units = 256;
    hipEvent_t event_begin = nullptr, event_fast = nullptr, event_dense = nullptr;  // per-kernel timing of the last call
    bool attributes_set = false;
    bool count_bytes = false;              // launch the instantiation of the fast kernel that counts its requested bytes
    bool query_order = true;               // the fast kernel's work queue hands out the queries with most columns first
    int64_t last_queries = 0;
    // last resort of the literal kernel (a query whose near-ties do not fit its LDS buffer): the argument block of the last
    // launch, the geometry's resolver, and a float64[n_truth] scratch vector allocated on first need
    std::vector<unsigned char> last_args;
    int (*resolve_ties)(ds_index *, hipStream_t, const std::vector<int32_t> &) = nullptr;
    ds::DeviceBuffer<double> row_scratch;
};

struct ds_titles {
    int device = 0;
    int64_t n = 0, stride = 0;
    bool has_counts = false;
    ds::DeviceBuffer<uint8_t> enc;      // [n][stride]
    ds::DeviceBuffer<uint8_t> len;      // [n]
    ds::DeviceBuffer<uint32_t> counts;  // [n][15] or empty
    // truth tables: what construct_features derives from the truth title ALONE (word boundaries, idf_s, ranks ...), one 160-byte
    // record per row for one (number_of_truth_titles, space code); built on the first indexed call that names them
    ds::DeviceBuffer<unsigned char> records;
    uint32_t records_n_truth = 0;
    uint8_t records_space = 0;
    ds::DeviceBuffer<int32_t> unit_queue;  // [64] heads of the features kernel's work queue, one per launch in turn (indexed entry points)
    size_t unit_queue_next = 0;
    bool records_enabled = true;        // ds_titles_option("truth_records", 0) switches them off (160 B per row of HBM)
};
