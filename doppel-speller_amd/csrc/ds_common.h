// Internal helpers shared by the translation units of libdoppel_amd.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "doppel_amd.h"

namespace ds {

// ---- geometry of the Jaccard kernels (see DESIGN.md "HBM layout") ------------------------------------------------
constexpr int kTileLog2 = 15;
constexpr int kTile = 1 << kTileLog2;        // truth rows per tile: one float32 score per row fills 128 KiB of LDS
constexpr int kSentinel = kTile;             // padding entry of a posting quad: lands in the trash slot scores[kTile]
constexpr int kThreads = 1024;               // one 16-wave workgroup per CU
constexpr int kMaxQueryColumns = 256;        // titles are <= 255 chars => <= 253 tri-grams (settings.py:68)
constexpr int kCandidates = 3072;            // capacity of the per-query candidate buffer in LDS
constexpr int kLooseStep = 1024;             // rows scanned between two capacity checks while no threshold exists
constexpr int kSlowSlots = 64;               // concurrent queries of the exact dense kernel (scratch = slots*N*8 B)

enum QueryStatus : int32_t { kQueryDone = 0, kQuerySlow = 1, kQueryErrorTopN = 2, kQueryErrorArg = 3 };

void set_error(const char *format, ...);
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
        count = n;
        if (n == 0) return DS_OK;
        DS_HIP(hipMalloc(reinterpret_cast<void **>(&ptr), n * sizeof(T)));
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
    float sums_min = 0.f;
    ds::DeviceBuffer<uint32_t> tile_ptr;   // [n_tiles][n_columns + 1], unit = quads of 4 postings
    ds::DeviceBuffer<uint16_t> postings;   // [n_quads * 4] tile-local truth rows, kSentinel-padded per (tile, column)
    ds::DeviceBuffer<float> idf32;         // [n_columns]
    ds::DeviceBuffer<float> sums32;        // [n_truth]
    ds::DeviceBuffer<double> slow_scratch; // [kSlowSlots][n_truth] float64 jaccard rows of the exact dense kernel
    ds::DeviceBuffer<int32_t> control;     // [16] work-queue head, slow-list length, error count, counters
    ds::DeviceBuffer<int32_t> status;      // per-query status of the last call (grown on demand)
    ds::DeviceBuffer<int32_t> slow_list;   // query ids routed to the exact dense kernel
    ds::DeviceBuffer<unsigned long long> phase;  // diagnostic phase timers (DS_PHASE_TIMERS=1)
    hipStream_t stream = nullptr;          // used by the host-pointer entry points
    int64_t last_queries = 0;
};

struct ds_titles {
    int device = 0;
    int64_t n = 0, stride = 0;
    bool has_counts = false;
    ds::DeviceBuffer<uint8_t> enc;      // [n][stride]
    ds::DeviceBuffer<uint8_t> len;      // [n]
    ds::DeviceBuffer<uint32_t> counts;  // [n][15] or empty
};
