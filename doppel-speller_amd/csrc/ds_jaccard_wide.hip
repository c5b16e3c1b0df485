// The Jaccard kernels in their WIDE geometry: 2 workgroups of 512 threads per CU, tiles of 28672 truth rows (56 KiB of
// packed scores per workgroup) -- truth sets above ds::kNarrowMaxTruth rows, where the number of tiles per query counts.
#define DS_GEOMETRY_NAME wide
#define DS_TILE_ROWS 28672
#define DS_THREADS 512
#define DS_WGS_PER_CU 2
#ifndef DS_CANDIDATES
#define DS_CANDIDATES 1600
#endif
#ifndef DS_PTR_TILES
#define DS_PTR_TILES 3
#endif
#include "ds_jaccard_impl.inc"
