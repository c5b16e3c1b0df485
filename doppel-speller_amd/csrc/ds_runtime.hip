// Library plumbing: error reporting, device memory, timers, and the construction of the tiled truth index in HBM.
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>

#include "ds_common.h"
#include "ds_host.h"

namespace ds {

static thread_local std::string g_last_error;

void set_error(const char *format, ...)
{
    char buffer[1024];
    va_list args;
    va_start(args, format);
    vsnprintf(buffer, sizeof(buffer), format, args);
    va_end(args);
    g_last_error = buffer;
}

int hip_failed(hipError_t error, const char *what, const char *file, int line)
{
    set_error("HIP error %d (%s) at %s:%d in %s", static_cast<int>(error), hipGetErrorString(error), file, line, what);
    return DS_E_HIP;
}

// fn(thread, column, first posting, last posting) for every column's postings whose rows fall into the thread's row range:
// the rows are cut into one contiguous range per thread and a thread finds its part of each (ascending) posting list by
// binary search, so per-row accumulators are written by one thread only and see a row's columns in ascending order.
template <typename F>
static void for_postings_by_row_range(const int64_t *rowptr, const int32_t *truth_idx, int64_t V, int64_t N, int threads, F fn)
{
    parallel_ranges(N, threads, [&](int thread, int64_t row_begin, int64_t row_end) {
        if (row_begin >= row_end) return;
        for (int64_t g = 0; g < V; ++g) {
            const int32_t *first = truth_idx + rowptr[g], *last = truth_idx + rowptr[g + 1];
            const int32_t *from = std::lower_bound(first, last, static_cast<int32_t>(row_begin));
            const int32_t *to = row_end > 0x7fffffffll ? last : std::lower_bound(from, last, static_cast<int32_t>(row_end));
            if (from < to) fn(thread, g, from - truth_idx, to - truth_idx);
        }
    });
}

// Duplicate ranks ("ties"): truth rows with the same column set and the same sums32 bits have the
// same jaccard for EVERY query; rank[t] = how many such rows have a larger row index than t (saturating at 65535).
// fast_arg_top_k returns the k largest row indexes at or above its threshold (match_maker.py:71), so a row with
// rank >= k can never be returned, and leaving it out does not move the k-th largest value either (k of its twins
// stay in).  Rows are grouped by a 128-bit hash of their column set from the last row down; a group member counts only
// after its column list has been compared with the group's first row: a hash collision leaves the row's rank at 0,
// which is always safe (a rank may under-count, never over-count).
// Threaded: hashes and column lists are accumulated per row range; the grouping and the ranking run per hash bucket
// (every thread owns the rows whose hash falls into its bucket and walks them from the last row down, so a group's
// first row and the order of its members are what the sequential walk finds).
void duplicate_ranks(const int64_t *rowptr, const int32_t *truth_idx, const float *sums32, int64_t V, int64_t N,
                     uint16_t *rank_out)
{
    const int threads = host_threads();
    std::vector<uint64_t> hash_a(static_cast<size_t>(N), 0u), hash_b(static_cast<size_t>(N), 0u);
    std::vector<uint32_t> row_columns(static_cast<size_t>(N), 0u);
    std::vector<uint64_t> image_a(static_cast<size_t>(V)), image_b(static_cast<size_t>(V));
    for (int64_t g = 0; g < V; ++g) {
        image_a[static_cast<size_t>(g)] = mix64(static_cast<uint64_t>(g) + 0x9e3779b97f4a7c15ull);
        image_b[static_cast<size_t>(g)] = mix64(static_cast<uint64_t>(g) * 0xd6e8feb86659fd93ull + 0x2545f4914f6cdd1dull);
    }
    for_postings_by_row_range(rowptr, truth_idx, V, N, threads, [&](int, int64_t g, int64_t from, int64_t to) {
        const uint64_t a = image_a[static_cast<size_t>(g)], b = image_b[static_cast<size_t>(g)];
        for (int64_t p = from; p < to; ++p) {
            const size_t t = static_cast<size_t>(truth_idx[p]);
            hash_a[t] += a;
            hash_b[t] += b;
            ++row_columns[t];
        }
    });
    auto sums_bits = [&](int64_t t) {
        uint32_t bits;
        std::memcpy(&bits, &sums32[t], sizeof(bits));
        return bits;
    };
    const int buckets = std::min(threads, 128);  // a bucket id is stored in a byte
    auto bucket_of = [&](int64_t t) {
        return static_cast<int>(mix64(hash_a[static_cast<size_t>(t)] ^ (hash_b[static_cast<size_t>(t)] << 1) ^ sums_bits(t)) >> 40) % buckets;
    };
    std::vector<uint8_t> owner(static_cast<size_t>(N));
    parallel_ranges(N, threads, [&](int, int64_t begin, int64_t end) {
        for (int64_t t = begin; t < end; ++t) owner[static_cast<size_t>(t)] = static_cast<uint8_t>(bucket_of(t));
    });
    std::vector<int32_t> group_first(static_cast<size_t>(N), -1);  // row -> first (largest) row of its hash group
    parallel_ranges(buckets, buckets, [&](int, int64_t bucket_begin, int64_t bucket_end) {
        for (int64_t bucket = bucket_begin; bucket < bucket_end; ++bucket) {
            size_t mine = 0;
            for (int64_t t = 0; t < N; ++t) mine += owner[static_cast<size_t>(t)] == bucket;
            struct Slot { uint64_t a, b; uint32_t bits; int32_t first; };
            size_t capacity = 16;
            while (capacity < mine * 2) capacity <<= 1;
            std::vector<Slot> table(capacity, Slot{0u, 0u, 0u, -1});
            for (int64_t t = N - 1; t >= 0; --t) {
                if (owner[static_cast<size_t>(t)] != bucket) continue;
                const uint32_t bits = sums_bits(t);
                const uint64_t a = hash_a[static_cast<size_t>(t)], b = hash_b[static_cast<size_t>(t)];
                size_t at = static_cast<size_t>(mix64(a ^ (b << 1) ^ bits)) & (capacity - 1);
                while (table[at].first >= 0 && !(table[at].a == a && table[at].b == b && table[at].bits == bits))
                    at = (at + 1) & (capacity - 1);
                if (table[at].first < 0) {
                    table[at] = Slot{a, b, bits, static_cast<int32_t>(t)};
                } else {
                    group_first[static_cast<size_t>(table[at].first)] = table[at].first;  // the first row joins with its first twin
                    group_first[static_cast<size_t>(t)] = table[at].first;
                }
            }
        }
    });
    // column lists of the grouped rows only (row-major; a row's columns arrive in ascending order)
    std::vector<int64_t> list_begin(static_cast<size_t>(N) + 1, 0);
    for (int64_t t = 0; t < N; ++t)
        list_begin[static_cast<size_t>(t) + 1] =
            list_begin[static_cast<size_t>(t)] + (group_first[static_cast<size_t>(t)] >= 0 ? row_columns[static_cast<size_t>(t)] : 0u);
    std::vector<int32_t> lists(static_cast<size_t>(list_begin[static_cast<size_t>(N)]));
    {
        std::vector<int64_t> fill(list_begin.begin(), list_begin.end() - 1);
        for_postings_by_row_range(rowptr, truth_idx, V, N, threads, [&](int, int64_t g, int64_t from, int64_t to) {
            for (int64_t p = from; p < to; ++p) {
                const size_t t = static_cast<size_t>(truth_idx[p]);
                if (group_first[t] >= 0) lists[static_cast<size_t>(fill[t]++)] = static_cast<int32_t>(g);
            }
        });
    }
    std::vector<uint32_t> twins(static_cast<size_t>(N), 0u);  // per group (at its first row): verified members so far
    parallel_ranges(buckets, buckets, [&](int, int64_t bucket_begin, int64_t bucket_end) {
        for (int64_t bucket = bucket_begin; bucket < bucket_end; ++bucket)
            for (int64_t t = N - 1; t >= 0; --t) {
                if (owner[static_cast<size_t>(t)] != bucket) continue;
                rank_out[t] = 0;
                const int32_t first = group_first[static_cast<size_t>(t)];
                if (first < 0 || first == t) continue;
                const int64_t mine = list_begin[static_cast<size_t>(t)], theirs = list_begin[static_cast<size_t>(first)];
                const uint32_t length = row_columns[static_cast<size_t>(t)];
                if (length != row_columns[static_cast<size_t>(first)] ||
                    !std::equal(lists.begin() + mine, lists.begin() + mine + length, lists.begin() + theirs))
                    continue;  // a hash collision: the row keeps rank 0
                const uint32_t rank = ++twins[static_cast<size_t>(first)];  // the group's first row + the twins before this one
                rank_out[t] = static_cast<uint16_t>(std::min<uint32_t>(rank, 0xffffu));
            }
    });
}

}  // namespace ds

extern "C" {

const char *ds_last_error(void) { return ds::g_last_error.c_str(); }

int ds_version(void) { return 200; }

#ifndef DS_BUILD_ID
#define DS_BUILD_ID "unidentified"
#endif
// the marker prefix lets the build script read the id of an existing binary without loading it
static const char kBuildIdMarker[] = "DS_BUILD_ID=" DS_BUILD_ID;
const char *ds_build_id(void) { return kBuildIdMarker + 12; }

int ds_device_count(int *count)
{
    DS_REQUIRE(count != nullptr, "ds_device_count: null pointer");
    *count = 0;
    DS_HIP(hipGetDeviceCount(count));
    return DS_OK;
}

int ds_device_name(int device, char *name, size_t capacity)
{
    DS_REQUIRE(name != nullptr && capacity > 0, "ds_device_name: bad buffer");
    hipDeviceProp_t properties;
    DS_HIP(hipGetDeviceProperties(&properties, device));
    snprintf(name, capacity, "%s (%s, %d CUs)", properties.name, properties.gcnArchName,
             properties.multiProcessorCount);
    return DS_OK;
}

}  // extern "C"

// ---- tiled index ---------------------------------------------------------------------------------------------------
// Input: the V x N inverted index of match_maker.py:122-133 in CSR form.  Output (HBM): the truth rows are cut
// into tiles of tile_rows rows (28672 or 12288: the geometry); every column's posting list is stored tile after tile as uint16 tile-local rows, each
// (column, tile) sub-list holds its even rows, then its odd rows, each part padded to whole quads (one 8-byte load per lane; a posting = (parity << 15) | (local row >> 1), padding = the word after the tile);
// col_ptr[column][tile] is the first quad of that sub-list, col_ptr[column][n_tiles] the end of the column.
// The constant per-posting value of match_maker.py:130 is never stored.
namespace {

#ifndef DS_POSTING_ORDER_DEFAULT
#define DS_POSTING_ORDER_DEFAULT 0
#endif

// Everything ds_index_create derives on the host, ready for upload (also digested by ds_index_image_digest for tests).
struct IndexImage {
    int64_t n_tiles = 0, tile_rows = 0, nnz = 0;
    uint64_t quads = 0;
    float sums_min = 0.f;
    bool literal_only = false, rows_sorted = false;
    std::vector<uint32_t> col_ptr, records;
    std::vector<uint16_t> postings, posting_sums;
    std::vector<float> tile_sums_min, tile_sums_max;
    std::vector<float> sums;      // sums32 in internal row order (what the kernels index)
    std::vector<int8_t> sig_column;
    // forward index (round 4): the columns of every row, ascending, rows in internal order -- row t's are
    // row_cols[row_start[t] .. row_start[t + 1]).  The exact stage of the fast kernel reads a candidate's ~21 columns in one
    // go instead of searching the row in every query column's posting list.
    // Stored as narrow as the index allows (round 5): the row starts as uint32 while nnz < 2^32, the columns as uint16 while
    // V <= 65536 (tri-grams over [a-z0-9 ]: V <= 50,653) -- C5: 2.4 GB instead of 4.7 GB.
    std::vector<int64_t> row_start64;
    std::vector<uint32_t> row_start32;
    std::vector<int32_t> row_cols32;
    std::vector<uint16_t> row_cols16;
    bool wide_start = false, wide_cols = false;
};

int64_t choose_tile_rows(int64_t N)
{
    // geometry: narrow tiles for truth sets of up to kNarrowMaxTruth rows (DS_GEOMETRY=wide|narrow overrides, for tests)
    int64_t tile_rows = N <= ds::kNarrowMaxTruth ? ds::kNarrowTileRows : ds::kWideTileRows;
    if (const char *geometry = getenv("DS_GEOMETRY"); geometry != nullptr) {
        if (std::strcmp(geometry, "wide") == 0) tile_rows = ds::kWideTileRows;
        if (std::strcmp(geometry, "narrow") == 0) tile_rows = ds::kNarrowTileRows;
    }
    return tile_rows;
}

int build_index_image(const int64_t *rowptr, const int32_t *truth_idx, const float *idf32, const float *sums32,
                      int64_t V, int64_t N, int64_t tile_rows, IndexImage &image)
{
    DS_REQUIRE(rowptr && idf32 && sums32, "ds_index_create: null input");
    DS_REQUIRE(V > 0 && N > 0, "ds_index_create: V and N must be positive (V=%lld N=%lld)", (long long)V,
               (long long)N);
    DS_REQUIRE(N < (int64_t(1) << 31) - ds::kWideTileRows, "ds_index_create: N too large for int32 row indexes");
    DS_REQUIRE(tile_rows > 0 && tile_rows % 2 == 0 && tile_rows / 2 < 0x8000, "ds_index_create: bad tile size");
    DS_REQUIRE(rowptr[0] == 0, "ds_index_create: rowptr[0] must be 0");
    const int64_t nnz = rowptr[V];
    DS_REQUIRE(nnz >= 0 && (nnz == 0 || truth_idx), "ds_index_create: bad nnz / truth_idx");
    const int64_t n_tiles = (N + tile_rows - 1) / tile_rows;
    const int64_t stride = n_tiles + 1;
    DS_REQUIRE(V * stride < (int64_t(1) << 40), "ds_index_create: list pointer table too large");

    const int threads = ds::host_threads();
    const bool trace = getenv("DS_BUILD_LOG") != nullptr;  // phase times of the build on stderr
    auto clock_now = [] { return std::chrono::steady_clock::now(); };
    auto phase_started = clock_now();
    auto phase = [&](const char *name) {
        if (trace) fprintf(stderr, "ds_index_create: %-28s %7.3f s\n", name, std::chrono::duration<double>(clock_now() - phase_started).count());
        phase_started = clock_now();
    };
    ds::FirstError error;

    for (int64_t g = 0; g < V; ++g)
        DS_REQUIRE(rowptr[g + 1] >= rowptr[g], "ds_index_create: rowptr not monotone at column %lld", (long long)g);
    // ---- internal row order: the truth rows are visited in ascending order of sums32 (ties: ascending
    // row index).  A tile then spans a narrow range of sums32, which makes the row-independent bound of a tile nearly as
    // sharp as a per-row one, lets a query start with the tiles whose rows can score highest (sums32 ~ its
    // max_intersection_possible) and skip whole tiles whose rows cannot reach the running threshold at all
    // (jaccard <= min(sums, maxint) / max(sums, maxint)).  Results are ORIGINAL row indexes: every row record carries
    // its own (word 6) and fast_arg_top_k's "k largest row indexes" is decided on those.  DS_SORT_ROWS=0 keeps the
    // caller's order (A/B measurements).
    std::vector<int32_t> original(static_cast<size_t>(N));
    std::vector<int32_t> permuted_idx;
    std::vector<float> sums_internal(static_cast<size_t>(N));
    const char *sort_switch = getenv("DS_SORT_ROWS");
    const bool sort_rows = sort_switch == nullptr || atoi(sort_switch) != 0;
    if (sort_rows) {
        // the caller's lists are validated before they are permuted (pass 1 below then sees lists that are valid by construction)
        ds::parallel_dynamic(V, 32, threads, [&](int, int64_t column_begin, int64_t column_end) {
            for (int64_t g = column_begin; g < column_end; ++g) {
                int64_t previous = -1;
                for (int64_t p = rowptr[g]; p < rowptr[g + 1]; ++p) {
                    if (!(truth_idx[p] > previous && truth_idx[p] < N)) {
                        error.raise("ds_index_create: posting list of column %lld is not strictly ascending within [0, N)",
                                    (long long)g);
                        return;
                    }
                    previous = truth_idx[p];
                }
            }
        });
        DS_REQUIRE(!error.failed(), "%s", error.message());
        // stable LSD radix sort of (order-preserving key of sums32, row): 11 + 11 + 10 bits
        auto key_of = [&](int64_t t) {
            uint32_t bits;
            std::memcpy(&bits, &sums32[t], sizeof(bits));
            return bits ^ ((bits >> 31) ? 0xffffffffu : 0x80000000u);  // float order as unsigned order
        };
        std::vector<int32_t> other(static_cast<size_t>(N));
        for (int64_t t = 0; t < N; ++t) original[static_cast<size_t>(t)] = static_cast<int32_t>(t);
        const int shifts[3] = {0, 11, 22}, widths[3] = {11, 11, 10};
        for (int pass = 0; pass < 3; ++pass) {
            // threaded and stable: every thread counts the digits of its contiguous range, the counts are turned into write
            // positions bucket by bucket and, within a bucket, thread by thread, and every thread scatters its range in order
            const uint32_t buckets = 1u << widths[pass], mask = buckets - 1u;
            std::vector<int64_t> start(static_cast<size_t>(threads) * buckets, 0);
            ds::parallel_ranges(N, threads, [&](int thread, int64_t begin, int64_t end) {
                int64_t *mine = start.data() + static_cast<size_t>(thread) * buckets;
                for (int64_t i = begin; i < end; ++i) ++mine[(key_of(original[static_cast<size_t>(i)]) >> shifts[pass]) & mask];
            });
            int64_t position = 0;
            for (uint32_t d = 0; d < buckets; ++d)
                for (int thread = 0; thread < threads; ++thread) {
                    int64_t &slot = start[static_cast<size_t>(thread) * buckets + d];
                    const int64_t here = slot;
                    slot = position;
                    position += here;
                }
            ds::parallel_ranges(N, threads, [&](int thread, int64_t begin, int64_t end) {
                int64_t *mine = start.data() + static_cast<size_t>(thread) * buckets;
                for (int64_t i = begin; i < end; ++i) {
                    const int32_t t = original[static_cast<size_t>(i)];
                    other[static_cast<size_t>(mine[(key_of(t) >> shifts[pass]) & mask]++)] = t;
                }
            });
            original.swap(other);
        }
        std::vector<int32_t> &position = other;  // original row -> internal row
        ds::parallel_ranges(N, threads, [&](int, int64_t begin, int64_t end) {
            for (int64_t i = begin; i < end; ++i) {
                position[static_cast<size_t>(original[static_cast<size_t>(i)])] = static_cast<int32_t>(i);
                sums_internal[static_cast<size_t>(i)] = sums32[original[static_cast<size_t>(i)]];
            }
        });
        permuted_idx.resize(static_cast<size_t>(nnz));
        ds::parallel_dynamic(V, 8, threads, [&](int, int64_t column_begin, int64_t column_end) {
            for (int64_t g = column_begin; g < column_end; ++g) {
                int32_t *list = permuted_idx.data() + rowptr[g];
                const int64_t length = rowptr[g + 1] - rowptr[g];
                for (int64_t p = 0; p < length; ++p) list[p] = position[static_cast<size_t>(truth_idx[rowptr[g] + p])];
                std::sort(list, list + length);
            }
        });
        truth_idx = permuted_idx.data();
        sums32 = sums_internal.data();
        phase("rows sorted by sums32");
    } else {
        for (int64_t t = 0; t < N; ++t) {
            original[static_cast<size_t>(t)] = static_cast<int32_t>(t);
            sums_internal[static_cast<size_t>(t)] = sums32[t];
        }
    }

    // pass 1 (threaded over columns): postings per (column, tile) -> quads, validating the lists; then the offsets
    std::vector<uint32_t> col_ptr(static_cast<size_t>(V * stride), 0u);
    ds::parallel_dynamic(V, 32, threads, [&](int, int64_t column_begin, int64_t column_end) {
        for (int64_t g = column_begin; g < column_end; ++g) {
            uint32_t *row = col_ptr.data() + g * stride;
            int64_t previous = -1, p = rowptr[g];
            const int64_t last = rowptr[g + 1];
            while (p < last) {
                if (!(truth_idx[p] > previous && truth_idx[p] < N)) {
                    error.raise("ds_index_create: posting list of column %lld is not strictly ascending within [0, N)",
                                (long long)g);
                    return;
                }
                const int64_t b = truth_idx[p] / tile_rows, tile_end = (b + 1) * tile_rows;
                uint32_t even = 0, odd = 0;
                for (; p < last && truth_idx[p] > previous && truth_idx[p] < tile_end; ++p) {
                    previous = truth_idx[p];
                    const uint32_t parity = static_cast<uint32_t>(previous - b * tile_rows) & 1u;
                    odd += parity;
                    even += 1u - parity;
                }
                row[b] = (even + 3u) / 4u + (odd + 3u) / 4u;
            }
        }
    });
    DS_REQUIRE(!error.failed(), "%s", error.message());
    uint64_t quads = 0;
    for (int64_t g = 0; g < V; ++g) {
        uint32_t *row = col_ptr.data() + g * stride;
        for (int64_t b = 0; b < n_tiles; ++b) {
            const uint32_t here = row[b];
            DS_REQUIRE(quads < 0xfffffff0ull, "ds_index_create: more than 2^32 posting quads");
            row[b] = static_cast<uint32_t>(quads);
            quads += here;
        }
        row[n_tiles] = static_cast<uint32_t>(quads);
    }
    DS_REQUIRE(quads < 0xfffffff0ull, "ds_index_create: more than 2^32 posting quads");
    phase("list pointers");
    std::vector<float> tile_sums_min(static_cast<size_t>(n_tiles), 0.f), tile_sums_max(static_cast<size_t>(n_tiles), 0.f);
    ds::parallel_dynamic(n_tiles, 16, threads, [&](int, int64_t tile_begin, int64_t tile_end) {
        for (int64_t b = tile_begin; b < tile_end; ++b) {
            const int64_t first = b * tile_rows, last = std::min<int64_t>(N, first + tile_rows);
            float lowest = sums32[first], highest = sums32[first];
            for (int64_t t = first + 1; t < last; ++t) {
                lowest = std::min(lowest, sums32[t]);
                highest = std::max(highest, sums32[t]);
            }
            tile_sums_min[static_cast<size_t>(b)] = lowest;
            tile_sums_max[static_cast<size_t>(b)] = highest;
        }
    });
    float sums_min = sums32[0];
    for (int64_t b = 0; b < n_tiles; ++b) sums_min = std::min(sums_min, tile_sums_min[static_cast<size_t>(b)]);
    // membership signature of the (up to) 128 densest columns: the columns a running threshold lets the kernel skip
    // first are the lowest-IDF = longest lists; one 4-byte load then replaces a binary search per skipped column
    std::vector<int8_t> sig_column(static_cast<size_t>(V), static_cast<int8_t>(-1));
    {
        std::vector<int64_t> by_length(static_cast<size_t>(V));
        for (int64_t g = 0; g < V; ++g) by_length[static_cast<size_t>(g)] = g;
        const size_t top = static_cast<size_t>(std::min<int64_t>(V, ds::kSignatureBits));
        std::partial_sort(by_length.begin(), by_length.begin() + top, by_length.end(), [&](int64_t x, int64_t y) {
            const int64_t lx = rowptr[x + 1] - rowptr[x], ly = rowptr[y + 1] - rowptr[y];
            return lx != ly ? lx > ly : x < y;
        });
        for (size_t bit = 0; bit < top; ++bit) {
            const int64_t g = by_length[bit];
            if ((rowptr[g + 1] - rowptr[g]) * 256 < N) break;  // not dense enough to be worth a bit
            sig_column[static_cast<size_t>(g)] = static_cast<int8_t>(bit);
        }
    }
    // The pruning bounds of the fast kernel assume what match_maker.py:135-142,174 produce: 0 <= idf < inf and
    // sums32[t] = the sum of the idf values of row t's columns (so a row's intersection score never exceeds its
    // sums).  An index that violates this (possible through the C ABI) is served by the literal kernel only.
    bool literal_only = false;
    for (int64_t g = 0; g < V && !literal_only; ++g) literal_only = !(idf32[g] >= 0.f && idf32[g] < 1e30f);
    // one walk over the postings per row range (threaded): signature bits and the idf total of every row
    std::vector<uint32_t> records(static_cast<size_t>(N) * ds::kRowRecordWords, 0u);  // row records, signature in words 0..3
    {
        std::vector<double> row_total(static_cast<size_t>(N), 0.0);
        ds::for_postings_by_row_range(rowptr, truth_idx, V, N, threads, [&](int, int64_t g, int64_t from, int64_t to) {
            const double value = idf32[g];
            for (int64_t p = from; p < to; ++p) row_total[static_cast<size_t>(truth_idx[p])] += value;
            const int bit = sig_column[static_cast<size_t>(g)];
            if (bit >= 0)
                for (int64_t p = from; p < to; ++p)
                    records[static_cast<size_t>(truth_idx[p]) * ds::kRowRecordWords + static_cast<size_t>(bit >> 5)] |= 1u << (bit & 31);
        });
        std::vector<uint8_t> violated(static_cast<size_t>(threads), 0);
        ds::parallel_ranges(N, threads, [&](int thread, int64_t begin, int64_t end) {
            bool bad = false;
            for (int64_t t = begin; t < end && !bad; ++t)
                bad = !(sums32[t] >= 0.f && sums32[t] < 1e30f) ||
                      static_cast<double>(sums32[t]) < row_total[static_cast<size_t>(t)] * (1.0 - 1e-4);
            violated[static_cast<size_t>(thread)] = bad;
        });
        for (uint8_t bad : violated) literal_only = literal_only || bad;
    }
    phase("signatures + row totals");

    // pass 2 (threaded over columns): fill.  A posting is (parity << 15) | (tile-local row >> 1): the LDS word of the row's packed score and the
    // half of it.  Within a (column, tile) sub-list the even rows come first, then the odd rows, each part padded to whole
    // quads with the word after the tile -- all four postings of a quad share their half, so the kernel derives shift and
    // mask once per quad.  (No kernel depends on the order of the postings inside a part: the exact stage reads the forward
    // index, the literal kernels add a column's value to every row of its list, each row once.)
    const uint16_t pad_word = static_cast<uint16_t>(tile_rows / 2);
    std::vector<uint16_t> postings(static_cast<size_t>(quads) * 4), posting_sums(static_cast<size_t>(quads) * 4);
    const char *order_switch = getenv("DS_POSTING_ORDER");  // 1: bank deal (below); 0: ascending rows
    const bool bank_deal = order_switch != nullptr ? atoi(order_switch) != 0 : DS_POSTING_ORDER_DEFAULT != 0;
    ds::parallel_dynamic(V, 32, threads, [&](int, int64_t column_begin, int64_t column_end) {
        std::vector<int32_t> by_bank;
        for (int64_t g = column_begin; g < column_end; ++g) {
            const uint32_t *row = col_ptr.data() + g * stride;
            int64_t p = rowptr[g];
            while (p < rowptr[g + 1]) {
                const int64_t b = truth_idx[p] / tile_rows, tile_first = b * tile_rows, tile_end = tile_first + tile_rows;
                int64_t last = p, even = 0;
                for (; last < rowptr[g + 1] && truth_idx[last] < tile_end; ++last) even += ((truth_idx[last] - tile_first) & 1) == 0;
                uint64_t write[2] = {static_cast<uint64_t>(row[b]) * 4u,
                                     (static_cast<uint64_t>(row[b]) + static_cast<uint64_t>(even + 3) / 4u) * 4u};
                const uint64_t part_end[2] = {write[1], static_cast<uint64_t>(row[b + 1]) * 4u};
                if (bank_deal) {
                    // Order of the postings inside a part (round 5): the 32 lanes of a half-wave hold 32 consecutive quads and
                    // issue their e-th LDS atomic together -- the e-th postings of consecutive quads should lie on different LDS
                    // banks (word mod 32).  The part's postings are dealt bank after bank, round after round (one posting per
                    // non-empty bank and round), and laid out slot-major: the i-th dealt posting is slot i / Q of quad i mod Q
                    // (Q = the part's quads), so that one slot of consecutive quads walks through the banks.
                    for (int part = 0; part < 2; ++part) {
                        uint32_t count[33] = {0};
                        int64_t members = 0;
                        for (int64_t i = p; i < last; ++i) {
                            const uint32_t local = static_cast<uint32_t>(truth_idx[i] - tile_first);
                            if (static_cast<int>(local & 1u) != part) continue;
                            ++count[((local >> 1) & 31u) + 1];
                            ++members;
                        }
                        if (members == 0) continue;
                        for (int bank = 0; bank < 32; ++bank) count[bank + 1] += count[bank];  // first place of every bank
                        by_bank.resize(static_cast<size_t>(members));
                        uint32_t cursor[32];
                        for (int bank = 0; bank < 32; ++bank) cursor[bank] = count[bank];
                        for (int64_t i = p; i < last; ++i) {
                            const uint32_t local = static_cast<uint32_t>(truth_idx[i] - tile_first);
                            if (static_cast<int>(local & 1u) == part) by_bank[cursor[(local >> 1) & 31u]++] = truth_idx[i];
                        }
                        const uint64_t base = write[part], part_quads = (part_end[part] - base) / 4u;
                        for (uint64_t i = 0; i < part_quads * 4u; ++i) {  // every place of the part: padding by default
                            postings[base + i] = static_cast<uint16_t>((part << 15) | pad_word);
                            posting_sums[base + i] = static_cast<uint16_t>(0xff00);
                        }
                        uint64_t dealt = 0;
                        for (uint32_t round = 0; dealt < static_cast<uint64_t>(members); ++round)
                            for (int bank = 0; bank < 32; ++bank) {
                                if (count[bank] + round >= count[bank + 1]) continue;
                                const int64_t t = by_bank[count[bank] + round];
                                const uint32_t local = static_cast<uint32_t>(t - tile_first);
                                const uint64_t at = base + 4u * (dealt % part_quads) + dealt / part_quads;
                                posting_sums[at] = static_cast<uint16_t>((ds::encode_sums8(sums32[t]) << 8) |
                                                                          (records[static_cast<size_t>(t) * ds::kRowRecordWords] & 0xffu));
                                postings[at] = static_cast<uint16_t>(((local & 1u) << 15) | (local >> 1));
                                ++dealt;
                            }
                    }
                    p = last;
                    continue;
                }
                for (; p < last; ++p) {
                    const int64_t t = truth_idx[p];
                    const uint32_t local = static_cast<uint32_t>(t - tile_first);
                    uint64_t &at = write[local & 1u];
                    posting_sums[at] = static_cast<uint16_t>((ds::encode_sums8(sums32[t]) << 8) |
                                                              (records[static_cast<size_t>(t) * ds::kRowRecordWords] & 0xffu));
                    postings[at++] = static_cast<uint16_t>(((local & 1u) << 15) | (local >> 1));
                }
                for (int part = 0; part < 2; ++part)  // padding: the word behind the tile, code 255
                    for (uint64_t i = write[part]; i < part_end[part]; ++i) {
                        postings[i] = static_cast<uint16_t>((part << 15) | pad_word);
                        posting_sums[i] = static_cast<uint16_t>(0xff00);
                    }
            }
        }
    });
    phase("postings");
    std::vector<uint16_t> dup_rank(static_cast<size_t>(N), 0u);
    ds::duplicate_ranks(rowptr, truth_idx, sums32, V, N, dup_rank.data());
    phase("duplicate ranks");
    ds::parallel_ranges(N, threads, [&](int, int64_t begin, int64_t end) {  // row records: what the refinement of a raw
        for (int64_t t = begin; t < end; ++t) {                             // entry gathers, in one cache line
            uint32_t *record = records.data() + static_cast<size_t>(t) * ds::kRowRecordWords;
            std::memcpy(&record[4], &sums32[t], sizeof(float));
            record[5] = dup_rank[static_cast<size_t>(t)];
            record[6] = static_cast<uint32_t>(original[static_cast<size_t>(t)]);  // the caller's row index: what the kernels return
        }
    });
    // forward index: per row range (threaded), columns ascending -- the walk visits the columns in ascending order
    const bool wide_start = nnz >= (int64_t(1) << 32) - 1, wide_cols = V > 65536;
    std::vector<int64_t> row_start(static_cast<size_t>(N) + 1, 0);
    ds::for_postings_by_row_range(rowptr, truth_idx, V, N, threads, [&](int, int64_t, int64_t from, int64_t to) {
        for (int64_t p = from; p < to; ++p) ++row_start[static_cast<size_t>(truth_idx[p]) + 1];
    });
    for (int64_t t = 0; t < N; ++t) row_start[static_cast<size_t>(t) + 1] += row_start[static_cast<size_t>(t)];
    {
        if (wide_cols) image.row_cols32.resize(static_cast<size_t>(nnz)); else image.row_cols16.resize(static_cast<size_t>(nnz));
        int32_t *const cols32 = image.row_cols32.data();
        uint16_t *const cols16 = image.row_cols16.data();
        std::vector<int64_t> fill(row_start.begin(), row_start.end() - 1);
        ds::for_postings_by_row_range(rowptr, truth_idx, V, N, threads, [&](int, int64_t g, int64_t from, int64_t to) {
            if (wide_cols)
                for (int64_t p = from; p < to; ++p) cols32[fill[static_cast<size_t>(truth_idx[p])]++] = static_cast<int32_t>(g);
            else
                for (int64_t p = from; p < to; ++p) cols16[fill[static_cast<size_t>(truth_idx[p])]++] = static_cast<uint16_t>(g);
        });
    }
    if (wide_start) {
        image.row_start64.swap(row_start);
    } else {
        image.row_start32.resize(row_start.size());
        for (size_t t = 0; t < row_start.size(); ++t) image.row_start32[t] = static_cast<uint32_t>(row_start[t]);
    }
    image.wide_start = wide_start;
    image.wide_cols = wide_cols;
    phase("forward index");
    image.n_tiles = n_tiles;
    image.tile_rows = tile_rows;
    image.nnz = nnz;
    image.quads = quads;
    image.sums_min = sums_min;
    image.literal_only = literal_only;
    image.rows_sorted = sort_rows;
    image.col_ptr.swap(col_ptr);
    image.records.swap(records);
    image.postings.swap(postings);
    image.posting_sums.swap(posting_sums);
    image.tile_sums_min.swap(tile_sums_min);
    image.tile_sums_max.swap(tile_sums_max);
    image.sums.swap(sums_internal);
    image.sig_column.swap(sig_column);
    return DS_OK;
}

}  // namespace

extern "C" {

int ds_index_create(const int64_t *rowptr, const int32_t *truth_idx, const float *idf32, const float *sums32,
                    int64_t V, int64_t N, int device, ds_index **out)
{
    DS_REQUIRE(out != nullptr, "ds_index_create: out is null");
    *out = nullptr;
    IndexImage image;
    const auto started = std::chrono::steady_clock::now();
    {
        const int built = build_index_image(rowptr, truth_idx, idf32, sums32, V, N, N > 0 ? choose_tile_rows(N) : 0, image);
        if (built != DS_OK) return built;
    }
    const int64_t n_tiles = image.n_tiles, tile_rows = image.tile_rows, nnz = image.nnz;
    const uint64_t quads = image.quads;
    const float sums_min = image.sums_min;
    const bool literal_only = image.literal_only;
    std::vector<uint32_t> &col_ptr = image.col_ptr, &records = image.records;
    std::vector<uint16_t> &postings = image.postings, &posting_sums = image.posting_sums;
    std::vector<float> &tile_sums_min = image.tile_sums_min;
    std::vector<int8_t> &sig_column = image.sig_column;
    DS_HIP(hipSetDevice(device));
    hipDeviceProp_t properties;
    DS_HIP(hipGetDeviceProperties(&properties, device));
    ds_index *index = new ds_index();
    index->device = device;
    index->compute_units = properties.multiProcessorCount > 0 ? properties.multiProcessorCount : 256;
    index->n_truth = N;
    index->n_columns = V;
    index->nnz = nnz;
    index->n_tiles = n_tiles;
    index->tile_rows = static_cast<int>(tile_rows);
    index->n_quads = static_cast<int64_t>(quads);
    index->sums_min = sums_min;
    index->literal_only = literal_only;
    index->rows_sorted = image.rows_sorted;
    int status = index->col_ptr.upload(col_ptr.data(), col_ptr.size());
    if (status == DS_OK) status = index->postings.upload(postings.data(), postings.size());
    if (status == DS_OK && postings.empty()) status = index->postings.allocate(4);
    if (status == DS_OK) status = index->posting_sums.upload(posting_sums.data(), posting_sums.size());
    if (status == DS_OK && posting_sums.empty()) status = index->posting_sums.allocate(4);
    if (status == DS_OK) status = index->idf32.upload(idf32, static_cast<size_t>(V));
    if (status == DS_OK) {  // padded so that the dense scan may read eight rows at once near the end
        std::vector<float> padded(static_cast<size_t>(N) + 8, 0.f);
        std::memcpy(padded.data(), image.sums.data(), sizeof(float) * static_cast<size_t>(N));
        status = index->sums32.upload(padded.data(), padded.size());
    }
    if (status == DS_OK) status = index->tile_sums_min.upload(tile_sums_min.data(), tile_sums_min.size());
    if (status == DS_OK) status = index->tile_sums_max.upload(image.tile_sums_max.data(), image.tile_sums_max.size());
    if (status == DS_OK) status = index->signature.upload(records.data(), records.size());
    if (status == DS_OK) status = index->sig_column.upload(sig_column.data(), sig_column.size());
    index->forward_wide_start = image.wide_start;
    index->forward_wide_cols = image.wide_cols;
    if (status == DS_OK)
        status = image.wide_start ? index->row_start.upload(reinterpret_cast<const unsigned char *>(image.row_start64.data()), image.row_start64.size() * 8)
                                  : index->row_start.upload(reinterpret_cast<const unsigned char *>(image.row_start32.data()), image.row_start32.size() * 4);
    if (status == DS_OK)
        status = image.wide_cols ? index->row_cols.upload(reinterpret_cast<const unsigned char *>(image.row_cols32.data()), image.row_cols32.size() * 4)
                                 : index->row_cols.upload(reinterpret_cast<const unsigned char *>(image.row_cols16.data()), image.row_cols16.size() * 2);
    if (status == DS_OK && index->row_cols.count == 0) status = index->row_cols.allocate(4);
    if (status == DS_OK) status = index->control.allocate(ds::kControlWords);
    if (status == DS_OK && (hipStreamCreate(&index->stream) != hipSuccess ||
                            hipEventCreate(&index->event_begin) != hipSuccess ||
                            hipEventCreate(&index->event_fast) != hipSuccess ||
                            hipEventCreate(&index->event_dense) != hipSuccess)) {
        ds::set_error("ds_index_create: hipStreamCreate / hipEventCreate failed");
        status = DS_E_HIP;
    }
    if (status != DS_OK) {
        delete index;
        return status;
    }
    if (getenv("DS_BUILD_LOG") != nullptr)
        fprintf(stderr, "ds_index_create: %-28s %7.3f s (host build + upload, %d threads)\n", "total",
                std::chrono::duration<double>(std::chrono::steady_clock::now() - started).count(), ds::host_threads());
    *out = index;
    return DS_OK;
}

// Host-only: FNV-1a digests of the arrays ds_index_create would upload, for a given tile size (tests: the image must
// not depend on the number of build threads).  digest[0..5] = col_ptr, postings, posting_sums, row records,
// tile_sums_min, sig_column; digest[6] = posting quads; digest[7] = literal_only.
int ds_index_image_digest(const int64_t *rowptr, const int32_t *truth_idx, const float *idf32, const float *sums32,
                          int64_t V, int64_t N, int64_t tile_rows, uint64_t digest[8])
{
    DS_REQUIRE(digest != nullptr, "ds_index_image_digest: null digest");
    IndexImage image;
    const int built = build_index_image(rowptr, truth_idx, idf32, sums32, V, N, tile_rows, image);
    if (built != DS_OK) return built;
    auto fnv = [](const void *data, size_t bytes) {
        const uint8_t *at = static_cast<const uint8_t *>(data);
        uint64_t h = 0xcbf29ce484222325ull;
        for (size_t i = 0; i < bytes; ++i) h = (h ^ at[i]) * 0x100000001b3ull;
        return h;
    };
    digest[0] = fnv(image.col_ptr.data(), image.col_ptr.size() * 4);
    digest[1] = fnv(image.postings.data(), image.postings.size() * 2);
    digest[2] = fnv(image.posting_sums.data(), image.posting_sums.size() * 2);
    digest[3] = fnv(image.records.data(), image.records.size() * 4);
    digest[4] = fnv(image.tile_sums_min.data(), image.tile_sums_min.size() * 4) ^ fnv(image.tile_sums_max.data(), image.tile_sums_max.size() * 4) ^
                fnv(image.sums.data(), image.sums.size() * 4);
    digest[5] = fnv(image.sig_column.data(), image.sig_column.size()) ^
                fnv(image.row_start64.data(), image.row_start64.size() * 8) ^ fnv(image.row_start32.data(), image.row_start32.size() * 4) ^
                fnv(image.row_cols32.data(), image.row_cols32.size() * 4) ^ fnv(image.row_cols16.data(), image.row_cols16.size() * 2);
    digest[6] = image.quads;
    digest[7] = image.literal_only ? 1u : 0u;
    return DS_OK;
}

int ds_index_duplicate_ranks(const int64_t *rowptr, const int32_t *truth_idx, const float *sums32, int64_t V, int64_t N,
                             uint16_t *rank_out)
{
    DS_REQUIRE(rowptr && sums32 && rank_out && V > 0 && N > 0, "ds_index_duplicate_ranks: bad argument");
    DS_REQUIRE(rowptr[0] == 0 && (rowptr[V] == 0 || truth_idx), "ds_index_duplicate_ranks: bad rowptr / truth_idx");
    for (int64_t g = 0; g < V; ++g) {
        DS_REQUIRE(rowptr[g + 1] >= rowptr[g], "ds_index_duplicate_ranks: rowptr not monotone");
        for (int64_t p = rowptr[g]; p < rowptr[g + 1]; ++p)
            DS_REQUIRE(truth_idx[p] >= 0 && truth_idx[p] < N, "ds_index_duplicate_ranks: row index outside [0, N)");
    }
    ds::duplicate_ranks(rowptr, truth_idx, sums32, V, N, rank_out);
    return DS_OK;
}

void ds_index_destroy(ds_index *index)
{
    if (!index) return;
    (void)hipSetDevice(index->device);
    if (index->stream) (void)hipStreamDestroy(index->stream);
    if (index->event_begin) (void)hipEventDestroy(index->event_begin);
    if (index->event_fast) (void)hipEventDestroy(index->event_fast);
    if (index->event_dense) (void)hipEventDestroy(index->event_dense);
    delete index;
}

int ds_index_info(const ds_index *index, int64_t info[8])
{
    DS_REQUIRE(index && info, "ds_index_info: null argument");
    info[0] = index->n_truth;
    info[1] = index->n_columns;
    info[2] = index->nnz;
    info[3] = index->tile_rows;
    info[4] = index->n_tiles;
    info[5] = static_cast<int64_t>(index->col_ptr.bytes() + index->postings.bytes() + index->posting_sums.bytes() + index->idf32.bytes() +
                                   index->sums32.bytes() + index->signature.bytes() + index->sig_column.bytes() +
                                   index->row_start.bytes() + index->row_cols.bytes());
    info[6] = index->n_quads * 4;
    info[7] = static_cast<int64_t>(index->row_start.bytes() + index->row_cols.bytes());  // of which: the forward index
    return DS_OK;
}

// ---- device memory / timers ----------------------------------------------------------------------------------------
int ds_malloc(void **ptr, size_t bytes, int device)
{
    DS_REQUIRE(ptr != nullptr, "ds_malloc: null pointer");
    DS_HIP(hipSetDevice(device));
    DS_HIP(hipMalloc(ptr, bytes ? bytes : 1));
    return DS_OK;
}

int ds_free(void *ptr, int device)
{
    DS_HIP(hipSetDevice(device));
    DS_HIP(hipFree(ptr));
    return DS_OK;
}

int ds_memcpy_h2d(void *dst, const void *src, size_t bytes, int device)
{
    DS_HIP(hipSetDevice(device));
    if (bytes) DS_HIP(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
    return DS_OK;
}

int ds_memcpy_d2h(void *dst, const void *src, size_t bytes, int device)
{
    DS_HIP(hipSetDevice(device));
    if (bytes) DS_HIP(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return DS_OK;
}

int ds_memset(void *dst, int value, size_t bytes, int device)
{
    DS_HIP(hipSetDevice(device));
    if (bytes) DS_HIP(hipMemset(dst, value, bytes));
    return DS_OK;
}

int ds_memcpy_d2d_async(void *dst, const void *src, size_t bytes, int device, void *stream)
{
    DS_HIP(hipSetDevice(device));
    if (bytes) DS_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, static_cast<hipStream_t>(stream)));
    return DS_OK;
}

int ds_stream_create(int device, void **stream)
{
    DS_REQUIRE(stream != nullptr, "ds_stream_create: null pointer");
    DS_HIP(hipSetDevice(device));
    hipStream_t created = nullptr;
    DS_HIP(hipStreamCreate(&created));
    *stream = created;
    return DS_OK;
}

int ds_stream_destroy(void *stream, int device)
{
    DS_HIP(hipSetDevice(device));
    if (stream) DS_HIP(hipStreamDestroy(static_cast<hipStream_t>(stream)));
    return DS_OK;
}

int ds_stream_sync(void *stream, int device)
{
    DS_HIP(hipSetDevice(device));
    DS_HIP(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    return DS_OK;
}

}  // extern "C"

struct ds_timer {
    int device = 0;
    hipEvent_t start = nullptr, stop = nullptr;
};

extern "C" {

int ds_timer_create(int device, ds_timer **out)
{
    DS_REQUIRE(out != nullptr, "ds_timer_create: null pointer");
    DS_HIP(hipSetDevice(device));
    ds_timer *timer = new ds_timer();
    timer->device = device;
    if (hipEventCreate(&timer->start) != hipSuccess || hipEventCreate(&timer->stop) != hipSuccess) {
        delete timer;
        ds::set_error("ds_timer_create: hipEventCreate failed");
        return DS_E_HIP;
    }
    *out = timer;
    return DS_OK;
}

void ds_timer_destroy(ds_timer *timer)
{
    if (!timer) return;
    if (timer->start) (void)hipEventDestroy(timer->start);
    if (timer->stop) (void)hipEventDestroy(timer->stop);
    delete timer;
}

int ds_timer_start(ds_timer *timer, void *stream)
{
    DS_REQUIRE(timer != nullptr, "ds_timer_start: null timer");
    DS_HIP(hipEventRecord(timer->start, static_cast<hipStream_t>(stream)));
    return DS_OK;
}

int ds_timer_stop(ds_timer *timer, void *stream)
{
    DS_REQUIRE(timer != nullptr, "ds_timer_stop: null timer");
    DS_HIP(hipEventRecord(timer->stop, static_cast<hipStream_t>(stream)));
    return DS_OK;
}

int ds_timer_elapsed_ms(ds_timer *timer, float *ms)
{
    DS_REQUIRE(timer != nullptr && ms != nullptr, "ds_timer_elapsed_ms: null argument");
    DS_HIP(hipEventSynchronize(timer->stop));
    DS_HIP(hipEventElapsedTime(ms, timer->start, timer->stop));
    return DS_OK;
}

}  // extern "C"
