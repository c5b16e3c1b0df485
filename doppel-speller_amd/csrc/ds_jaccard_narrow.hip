// The Jaccard kernels in their NARROW geometry: 4 workgroups of 256 threads per CU, tiles of 12288 truth rows (24 KiB of
// packed scores per workgroup).  Four independent queries per CU hide each other's barriers and load latencies and the
// dense first tile is smaller: 23.0 against 25.8 ms on C2 (18 -> 41 tiles per query); at 5M truth rows (175 -> 407
// tiles) the wide geometry is 2.6 % ahead (profiles/r02_tuning.txt).
#define DS_GEOMETRY_NAME narrow
#define DS_TILE_ROWS 12288
#define DS_THREADS 256
#define DS_WGS_PER_CU 4
#ifndef DS_NARROW_CANDIDATES
#define DS_NARROW_CANDIDATES 832
#endif
#define DS_CANDIDATES DS_NARROW_CANDIDATES
#define DS_PTR_TILES 1
#include "ds_jaccard_impl.inc"
