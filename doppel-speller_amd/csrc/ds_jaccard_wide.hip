// The Jaccard kernels in their WIDE geometry: 2 workgroups of 512 threads per CU, tiles of 28672 truth rows (56 KiB of
// packed scores per workgroup) -- truth sets above ds::kNarrowMaxTruth rows, where the number of tiles per query counts.
#include "doppel_amd.h"
#define DS_GEOMETRY_NAME wide
#ifndef DS_WIDE_THREADS
#define DS_WIDE_THREADS 512
#endif
#ifndef DS_WIDE_WGS_PER_CU
#define DS_WIDE_WGS_PER_CU 2
#endif
#ifndef DS_WIDE_CANDIDATES
#define DS_WIDE_CANDIDATES 1472  // 512 bytes go to the raw entries' tile bytes (epochs of 4 tiles), 1.6 KiB to the item directory
#endif
#ifndef DS_WIDE_PTR_TILES
#define DS_WIDE_PTR_TILES 3
#endif
#define DS_TILE_ROWS ds::kWideTileRows
#define DS_THREADS DS_WIDE_THREADS
#define DS_WGS_PER_CU DS_WIDE_WGS_PER_CU
#define DS_CANDIDATES DS_WIDE_CANDIDATES
#define DS_PTR_TILES DS_WIDE_PTR_TILES
#ifndef DS_WIDE_EPOCH
#define DS_WIDE_EPOCH 16  // round 5 at the C5 shape (20k queries x 50M rows, top-100): 152.6 (4) / 145.8 (8) / 144.1 ms (16); round 3 (40k queries): 361.2 (4) / 349.2 (8) / 347.4 ms (16); round 2 at the C3 shape: 91.1 (1) / 88.9 (2) / 86.5 (4) / 85.9 ms (8)
#endif
#define DS_EPOCH_TILES DS_WIDE_EPOCH
#include "ds_jaccard_impl.inc"
