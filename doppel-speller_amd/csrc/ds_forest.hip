// Next row f-4: the gradient-boosted tree ensemble applied to the 66 features
// (doppelspeller/predict.py:229-234: xgb.DMatrix(features); model.predict(..., ntree_limit=best_ntree_limit)), so that
// the feature matrix written by ds_construct_features_* never leaves HBM (264 B per pair saved).
//
// xgboost is a third-party dependency that is not part of the reference tree; the kernel follows the published
// prediction rule of a `binary:logistic` booster on a dense float32 matrix with missing = NaN: per tree, from the root,
// a NaN follows the node's `missing` child, otherwise value < split_condition follows `yes`, else `no`; leaf values of
// the first n_trees trees are added in tree order in float32 to the base margin; prediction = 1 / (1 + exp(-margin)).
// The GPU tests compare margins bit-for-bit with a CPU restatement; the float32 exp of the sigmoid is within 2 ulp of libm.
// Parity unpinned against the real library.
//
// One thread per row: the rows of a workgroup are staged in LDS (row-major 66-float reads would be uncoalesced), the
// node arrays (a few hundred KiB for a typical model) stay L2-resident and are walked by all rows of a wavefront in
// the same tree at the same time.
#include "ds_common.h"

struct ds_forest {
    int device = 0;
    int32_t n_trees = 0, n_features = 0;
    int64_t n_nodes = 0;
    float base_margin = 0.f;
    ds::DeviceBuffer<int4> nodes;          // (feature or -1, yes, no, missing) per node, tree-relative child ids
    ds::DeviceBuffer<float> threshold;     // split condition, or the leaf value
    ds::DeviceBuffer<int64_t> tree_offsets;
};

namespace ds {

constexpr int kForestThreads = 256;
constexpr int kForestFeaturesMax = 96;  // LDS staging: kForestThreads x n_features floats (66 for this reference)

struct ForestArgs {
    const int4 *nodes;
    const float *threshold;
    const int64_t *tree_offsets;
    const float *rows;
    float *margins;        // nullable
    float *probabilities;  // nullable
    int64_t n;
    int32_t n_trees, n_features;
    float base_margin;
};

__global__ __launch_bounds__(kForestThreads) void ds_forest_kernel(ForestArgs a)
{
    extern __shared__ float staged[];  // [kForestThreads][n_features + 1] (+1: bank spread)
    const int stride = a.n_features + 1;
    for (int64_t first = static_cast<int64_t>(blockIdx.x) * kForestThreads; first < a.n;
         first += static_cast<int64_t>(gridDim.x) * kForestThreads) {
        const int rows_here = static_cast<int>(a.n - first < kForestThreads ? a.n - first : kForestThreads);
        __syncthreads();
        for (int e = threadIdx.x; e < rows_here * a.n_features; e += kForestThreads) {  // coalesced copy
            const int r = e / a.n_features, f = e - r * a.n_features;
            staged[r * stride + f] = a.rows[first * a.n_features + e];
        }
        __syncthreads();
        if (static_cast<int>(threadIdx.x) < rows_here) {
            const float *row = staged + threadIdx.x * stride;
            float margin = 0.f;  // xgboost: PredValue sums the leaves from zero, the base margin is added to that sum
            for (int32_t t = 0; t < a.n_trees; ++t) {
                const int64_t root = a.tree_offsets[t];
                int64_t node = root;
                int4 info = a.nodes[node];
                while (info.x >= 0) {
                    const float value = row[info.x];
                    const int next = (value != value) ? info.w : (value < a.threshold[node] ? info.y : info.z);
                    node = root + next;
                    info = a.nodes[node];
                }
                margin = margin + a.threshold[node];
            }
            margin = a.base_margin + margin;
            if (a.margins) a.margins[first + threadIdx.x] = margin;
            if (a.probabilities) a.probabilities[first + threadIdx.x] = 1.0f / (1.0f + expf(-margin));
        }
    }
}

}  // namespace ds

extern "C" {

int ds_forest_create(const int32_t *feature, const float *threshold, const int32_t *yes, const int32_t *no,
                     const int32_t *missing, const int64_t *tree_offsets, int32_t n_trees, int32_t n_features,
                     float base_margin, int device, ds_forest **out)
{
    DS_REQUIRE(out != nullptr, "ds_forest_create: out is null");
    *out = nullptr;
    DS_REQUIRE(feature && threshold && yes && no && missing && tree_offsets, "ds_forest_create: null input");
    DS_REQUIRE(n_trees >= 0 && n_features >= 1 && n_features <= ds::kForestFeaturesMax,
               "ds_forest_create: n_trees=%d n_features=%d out of range", n_trees, n_features);
    DS_REQUIRE(tree_offsets[0] == 0, "ds_forest_create: tree_offsets[0] must be 0");
    const int64_t n_nodes = tree_offsets[n_trees];
    std::vector<int4> nodes(static_cast<size_t>(n_nodes));
    for (int32_t t = 0; t < n_trees; ++t) {
        const int64_t begin = tree_offsets[t], end = tree_offsets[t + 1];
        DS_REQUIRE(end > begin, "ds_forest_create: tree %d is empty", t);
        for (int64_t i = begin; i < end; ++i) {
            const int32_t f = feature[i];
            DS_REQUIRE(f < n_features, "ds_forest_create: node %lld splits on feature %d >= %d", (long long)i, f,
                       n_features);
            if (f >= 0) {
                const int64_t size = end - begin;
                // children must lie inside the tree and after their parent (no cycles: the walk always terminates)
                DS_REQUIRE(yes[i] > i - begin && yes[i] < size && no[i] > i - begin && no[i] < size &&
                               missing[i] > i - begin && missing[i] < size,
                           "ds_forest_create: node %lld has a child outside its tree or before itself", (long long)i);
            }
            nodes[static_cast<size_t>(i)] = make_int4(f >= 0 ? f : -1, yes[i], no[i], missing[i]);
        }
    }
    DS_HIP(hipSetDevice(device));
    ds_forest *forest = new ds_forest();
    forest->device = device;
    forest->n_trees = n_trees;
    forest->n_features = n_features;
    forest->n_nodes = n_nodes;
    forest->base_margin = base_margin;
    int status = forest->nodes.upload(nodes.data(), nodes.size());
    if (status == DS_OK && nodes.empty()) status = forest->nodes.allocate(1);
    if (status == DS_OK) status = forest->threshold.upload(threshold, static_cast<size_t>(n_nodes));
    if (status == DS_OK && n_nodes == 0) status = forest->threshold.allocate(1);
    if (status == DS_OK) status = forest->tree_offsets.upload(tree_offsets, static_cast<size_t>(n_trees) + 1);
    if (status != DS_OK) {
        delete forest;
        return status;
    }
    *out = forest;
    return DS_OK;
}

void ds_forest_destroy(ds_forest *forest)
{
    if (!forest) return;
    (void)hipSetDevice(forest->device);
    delete forest;
}

int ds_forest_predict_device(ds_forest *forest, const float *d_rows, int64_t n, float *d_margins,
                             float *d_probabilities, void *stream)
{
    DS_REQUIRE(forest != nullptr && n >= 0, "ds_forest_predict: bad arguments");
    if (n == 0) return DS_OK;
    DS_REQUIRE(d_rows && (d_margins || d_probabilities), "ds_forest_predict: null pointer");
    DS_HIP(hipSetDevice(forest->device));
    ds::ForestArgs args;
    args.nodes = forest->nodes.ptr;
    args.threshold = forest->threshold.ptr;
    args.tree_offsets = forest->tree_offsets.ptr;
    args.rows = d_rows;
    args.margins = d_margins;
    args.probabilities = d_probabilities;
    args.n = n;
    args.n_trees = forest->n_trees;
    args.n_features = forest->n_features;
    args.base_margin = forest->base_margin;
    const int64_t blocks = (n + ds::kForestThreads - 1) / ds::kForestThreads;
    const int grid = static_cast<int>(std::min<int64_t>(blocks, 256 * 16));
    const size_t lds = static_cast<size_t>(ds::kForestThreads) * (forest->n_features + 1) * sizeof(float);
    if (lds > 48 * 1024)
        DS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(ds::ds_forest_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    hipLaunchKernelGGL(ds::ds_forest_kernel, dim3(grid), dim3(ds::kForestThreads), lds, static_cast<hipStream_t>(stream),
                       args);
    DS_HIP(hipGetLastError());
    return DS_OK;
}

int ds_forest_predict(ds_forest *forest, const float *rows, int64_t n, float *margins, float *probabilities)
{
    DS_REQUIRE(forest != nullptr && n >= 0, "ds_forest_predict: bad arguments");
    if (n == 0) return DS_OK;
    DS_REQUIRE(rows && (margins || probabilities), "ds_forest_predict: null pointer");
    DS_HIP(hipSetDevice(forest->device));
    ds::DeviceBuffer<float> d_rows, d_margins, d_probabilities;
    int status = d_rows.upload(rows, static_cast<size_t>(n) * forest->n_features);
    if (status == DS_OK && margins) status = d_margins.allocate(static_cast<size_t>(n));
    if (status == DS_OK && probabilities) status = d_probabilities.allocate(static_cast<size_t>(n));
    if (status != DS_OK) return status;
    status = ds_forest_predict_device(forest, d_rows.ptr, n, margins ? d_margins.ptr : nullptr,
                                      probabilities ? d_probabilities.ptr : nullptr, nullptr);
    if (status != DS_OK) return status;
    DS_HIP(hipDeviceSynchronize());
    if (margins) DS_HIP(hipMemcpy(margins, d_margins.ptr, sizeof(float) * n, hipMemcpyDeviceToHost));
    if (probabilities) DS_HIP(hipMemcpy(probabilities, d_probabilities.ptr, sizeof(float) * n, hipMemcpyDeviceToHost));
    return DS_OK;
}

}  // extern "C"
