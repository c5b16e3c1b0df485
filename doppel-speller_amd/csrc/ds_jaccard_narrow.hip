// The Jaccard kernels in their NARROW geometry: 4 workgroups of 256 threads per CU, tiles of 12288 truth rows (24 KiB of
// packed scores per workgroup).  Four independent queries per CU hide each other's barriers and load latencies and the
// dense first tile is smaller: 23.0 against 25.8 ms on C2 (18 -> 41 tiles per query); at 5M truth rows (175 -> 407
// tiles) the wide geometry is 2.6 % ahead (profiles/r02_tuning.txt).
#include "doppel_amd.h"
#define DS_GEOMETRY_NAME narrow
#ifndef DS_NARROW_THREADS
#define DS_NARROW_THREADS 256
#endif
#ifndef DS_NARROW_WGS_PER_CU
#define DS_NARROW_WGS_PER_CU 4
#endif
#ifndef DS_NARROW_CANDIDATES
#define DS_NARROW_CANDIDATES 768  // 832 before the item directory took 1.6 KiB of LDS
#endif
#ifndef DS_NARROW_PTR_TILES
#define DS_NARROW_PTR_TILES 1
#endif
#define DS_TILE_ROWS ds::kNarrowTileRows
#define DS_THREADS DS_NARROW_THREADS
#define DS_WGS_PER_CU DS_NARROW_WGS_PER_CU
#define DS_CANDIDATES DS_NARROW_CANDIDATES
#define DS_PTR_TILES DS_NARROW_PTR_TILES
#ifndef DS_NARROW_EPOCH
#define DS_NARROW_EPOCH 4  // measured on C2: 20.78 (1) / 20.47 (2) / 19.87 (4) / 19.70 (8, 800 candidates) / 19.98 ms (16); top-100: 28.96 (4) / 31.7 (8) / 34.3 ms (16)
#endif
#define DS_EPOCH_TILES DS_NARROW_EPOCH
#include "ds_jaccard_impl.inc"
