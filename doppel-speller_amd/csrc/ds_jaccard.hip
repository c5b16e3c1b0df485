// C entry points of the Jaccard top-k path (include/doppel_amd.h) and the choice between the two geometries the kernels
// are compiled for (ds_jaccard_wide.hip / ds_jaccard_narrow.hip, both from ds_jaccard_impl.inc): ds_index_create picks
// the tile size of an index, the launch follows it.
#include "ds_common.h"

namespace ds {
namespace wide {
int launch(ds_index *index, const int64_t *d_q_rowptr, const int32_t *d_q_cols, const double *d_q_maxint, int64_t Q,
           int32_t k, int32_t *d_out_rows, hipStream_t stream);
int collect(ds_index *index, hipStream_t stream, int64_t stats[32]);  // geometry-independent: one copy serves both
}  // namespace wide
namespace narrow {
int launch(ds_index *index, const int64_t *d_q_rowptr, const int32_t *d_q_cols, const double *d_q_maxint, int64_t Q,
           int32_t k, int32_t *d_out_rows, hipStream_t stream);
}  // namespace narrow
}  // namespace ds

typedef int (*ds_launch_function)(ds_index *, const int64_t *, const int32_t *, const double *, int64_t, int32_t, int32_t *,
                                  hipStream_t);

static ds_launch_function launch_for(const ds_index *index)
{
    return index != nullptr && index->tile_rows == ds::kNarrowTileRows ? ds::narrow::launch : ds::wide::launch;
}

extern "C" {

int ds_jaccard_topk_device(ds_index *index, const int64_t *d_q_rowptr, const int32_t *d_q_cols,
                           const double *d_q_maxint, int64_t Q, int32_t k, int32_t *d_out_rows, void *stream)
{
    return launch_for(index)(index, d_q_rowptr, d_q_cols, d_q_maxint, Q, k, d_out_rows, static_cast<hipStream_t>(stream));
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
    if (std::strcmp(name, "query_order") == 0) {  // 1 (default): the work queue hands out the queries with most columns first
        index->query_order = value != 0;
        return DS_OK;
    }
    ds::set_error("ds_index_option: unknown option '%s'", name);
    return DS_E_ARG;
}

int ds_jaccard_sync(ds_index *index, void *stream, int64_t stats[32])
{
    return ds::wide::collect(index, static_cast<hipStream_t>(stream), stats);
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
    status = launch_for(index)(index, d_rowptr.ptr, d_cols.ptr, d_maxint.ptr, Q, k, d_out.ptr, index->stream);
    if (status != DS_OK) return status;
    status = ds::wide::collect(index, index->stream, nullptr);
    DS_HIP(hipMemcpy(out_rows, d_out.ptr, static_cast<size_t>(Q) * k * sizeof(int32_t), hipMemcpyDeviceToHost));
    return status;
}

}  // extern "C"
