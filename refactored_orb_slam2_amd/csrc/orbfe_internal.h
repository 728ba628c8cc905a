// orbfe_internal.h -- structs shared by the host side of liborbfe and its HIP kernels (gfx950).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/orbfe.h"

#define ORBFE_EDGE 16            // minBorder = EDGE_THRESHOLD - 3 (L/src/ORBextractor.cc:740)
#define ORBFE_CELL_MAX 66        // largest FAST cell ROI side the cell kernel stages in LDS
#ifndef ORBFE_BLUR_TILE_H
#define ORBFE_BLUR_TILE_H 56      // rows of a blur tile (64 wide); a multiple of 8: its row blocks are the 16 x 8 storage tiles
#endif
#define ORBFE_MAX_INI 256        // largest nIni (root nodes of DistributeOctTree) supported
#ifndef ORBFE_OCT_THREADS
#define ORBFE_OCT_THREADS 256
#endif

// Per-level geometry of one image pyramid, passed to kernels by value.
struct PyrView {
  const uint8_t* base[ORBFE_MAX_LEVELS];  // level plane of image 0
  unsigned long long img_stride[ORBFE_MAX_LEVELS];  // bytes between consecutive images at this level
  int pitch[ORBFE_MAX_LEVELS];
  int w[ORBFE_MAX_LEVELS], h[ORBFE_MAX_LEVELS];
  int n_levels;
  unsigned tiled;   // bit l: level l is stored in 16 x 8-pixel tiles of 128 bytes (orbfe_level_offset) instead of row-major
};

// A raw level written by the fused level kernel (blur_level_kernel<true>) is stored TILED -- 16 pixels x 8 rows = one 128-byte
// line, tiles in raster order, pitch / 16 tiles per tile row, rows padded to 8 (the planes are sized for that) -- because its
// readers fetch windows (the next level's tiles, FAST cells, 31 x 31 orientation patches, 11-row SAD windows) and the memory path
// charges per 128-byte line touched, not per byte (DESIGN lesson 44).  Level 0 (the caller's image) and levels written by the
// stand-alone resize kernels stay row-major; PyrView::tiled says which is which.
#ifndef ORBFE_TILED_LEVELS
#define ORBFE_TILED_LEVELS 1
#endif
// byte offset of pixel (x, y), x, y >= 0; a plane is < 2^25 bytes (4095 x 4095)
__host__ __device__ inline uint32_t orbfe_tiled_offset(int x, int y, int pitch) {
  return (((uint32_t)(y >> 3) * (uint32_t)(pitch >> 4) + (uint32_t)(x >> 4)) << 7) + (uint32_t)((y & 7) * 16 + (x & 15));
}
__host__ __device__ inline uint32_t orbfe_level_offset(int x, int y, int pitch, bool tiled) {
  return tiled ? orbfe_tiled_offset(x, y, pitch) : (uint32_t)y * (uint32_t)pitch + (uint32_t)x;
}

// Horizontal / vertical coefficient tables of cv::resize INTER_LINEAR (8 bytes per destination index)
struct ResizeTap {
  int16_t s0, s1;  // source indices (clamped)
  int16_t c0, c1;  // fixed-point weights, sum 2048
};

// One FAST cell of ComputeKeyPointsOctTree (L/src/ORBextractor.cc:756-771)
struct CellDesc {
  int16_t level;
  int16_t x0, y0;      // ROI origin (iniX, iniY) in level pixels
  int16_t cols, rows;  // ROI size handed to cv::FAST
  int16_t shift_x, shift_y;  // j*wCell, i*hCell added to FAST's local coordinates
  int16_t slot_cap;    // capacity of this cell's output slot
  uint32_t slot_off;   // offset (entries) of this cell's slot inside one image's slot block
  // the quick-test loop's lane layout, computed once at plan time (the kernel spent two scalar divisions and a reciprocal per cell on
  // them): pixel pairs per tested row, rows per iteration of 64 lanes, iterations; 1 / ppr
  int16_t ppr, ri, n_it, pad;
  float inv_ppr;
};

// A run of horizontally adjacent FAST cells of one cell row: one wave of fast_cells_kernel walks them one after the other
#ifndef ORBFE_FG_MAX
#define ORBFE_FG_MAX 4           // cells per run
#define ORBFE_FG_MAX_WIDTH 176   // pixels a run may span
#endif
struct FastGroup {
  int32_t first_cell;            // index into the cell table; the group's cells are consecutive there
  int16_t n_cells, level;
  int16_t x0, y0, width, rows;   // group ROI = union of the cells' ROIs
  int16_t wcell, pad;            // x step between the cells' ROIs
};

// Per-level parameters of DistributeOctTree
struct OctLevel {
  int cell_begin, n_cells;  // range in the cell table
  int N;                    // mnFeaturesPerLevel[level]
  int n_ini;
  float hX;
  int width, height;        // maxX-minX, maxY-minY
  int kp_off, kp_cap;       // slice of the per-image level-keypoint array
  unsigned long long key_off;  // offset (entries) of this level's global key fallback inside one image's block
  int key_cap;
  int ff_depth;             // subdivisions a level with its keys in HBM takes from count tables (octree_select_kernel); 0: none
};

struct OctParams {
  OctLevel lv[ORBFE_MAX_LEVELS];
  const CellDesc* cells;
  const int32_t* cell_cnt;   // [image][total_cells]
  const uint32_t* slots;     // [image][slots_per_image] packed x | y<<12 | score<<24
  int32_t* cell_off;         // [image][total_cells] scratch
  unsigned long long* gkeys; // [image][gkeys_per_image] fallback key storage
  uint32_t* lvl_kp;          // [image][kp_per_image] selected keypoints packed like slots (level coords - 16)
  int32_t* lvl_n;            // [image][n_levels]
  int32_t* err;              // device error word
  int total_cells;
  unsigned long long slots_per_image, gkeys_per_image;
  int kp_per_image;
  int n_levels;
  int max_nodes;             // M: node table capacity (dynamic LDS sized from it)
  int lds_keys;              // CAP: keys that fit the LDS key array
};

struct DescribeParams {
  PyrView pyr;      // un-blurred pyramid (orientation)
  PyrView blur;     // blurred pyramid (descriptor)
  const uint32_t* lvl_kp;
  const int32_t* lvl_n;
  int kp_off[ORBFE_MAX_LEVELS], kp_cap[ORBFE_MAX_LEVELS];
  float scale[ORBFE_MAX_LEVELS];
  float kp_size[ORBFE_MAX_LEVELS];
  int kp_per_image;
  int n_levels;
  orbfe_keypoint* out_kps;
  uint8_t* out_desc;
  int32_t* out_n;
  int cap;
  int xcd_images;   // 1: XCD k works on images k, k+8, ...
};

struct BlurTile {
  int16_t level, tx, ty;  // tile origin = (tx * 64, ty * ORBFE_BLUR_TILE_H)
  // the part of level + 1 this tile produces when the resize step is fused into the blur (blur_level_kernel<true>): the
  // destination dwords (four pixels) [j0, j1) whose first source column lies in this tile column, and the destination rows
  // [r0, r1) whose upper source row lies in this tile row -- every destination dword-row has exactly one owner
  int16_t j0, j1, r0, r1;
  int16_t pad;
};
// the level l -> l + 1 step of cv::resize as the blur tiles of level l see it
struct LevelResize {
  uint8_t* dst;               // level l + 1 of image 0
  unsigned long long dimg;    // bytes between images
  int dpitch, dw, dh;
  int dst_tiled;              // level l + 1 is written in 16 x 8 tiles (PyrView::tiled bit l + 1)
  const ResizeTap* xt;
  const ResizeTap* yt;
};
#define ORBFE_FUSE_DWORDS 16     // destination dwords a tile may own per row (one lane each)
#define ORBFE_FUSE_ROWS 3        // destination rows per thread: a tile may own 16 x 3 rows

// gauss_blur7_mfma_kernel: one wave blurs a strip of 48 output columns over the whole height of a level.  The 7-tap passes are
// band-matrix products on the matrix cores; the band matrices (REFLECT_101 folded into the edge ones) are precomputed per level
// as MFMA operands, 1 KB each (64 lanes x 16 bytes).
#define ORBFE_BLUR_CHUNK 48      // output columns per strip: three 16-column MFMA tiles from one 64-byte row segment
#define ORBFE_BLUR_WINDOW 32     // output rows per step: two 16-row MFMA tiles from a 64-row window
struct BlurStrip {
  int16_t level, chunk;
};
struct BlurMfmaParams {
  const BlurStrip* strips;
  int n_strips;
  const uint8_t* tab;
  uint32_t b_off[ORBFE_MAX_LEVELS];   // horizontal matrices of a level: entry (chunk * 3 + column group)
  uint32_t t_off;                     // the two vertical matrices (row groups 0, 1 of a window)
  int use;                            // the handle asked for this kernel (orbfe_debug_blur_kernel)
};

// launchers (extract_kernels.hip)
void orbfe_launch_resize(const uint8_t* src, int spitch, size_t simg, uint8_t* dst, int dpitch, size_t dimg, int dw,
                         int dh, const ResizeTap* xt, const ResizeTap* yt, int n_images, int mode, hipStream_t s);   // mode: 0 direct gathers, 1 LDS-staged, 2 LDS-staged + 8-byte windows
void orbfe_launch_fast_cells(const PyrView& pyr, const CellDesc* cells, const FastGroup* groups, int n_groups, int total_cells,
                             int cell_rows, int cell_span, int sc_max, int bits_max, int32_t* cell_cnt, uint32_t* slots,
                             unsigned long long slots_per_image, int ini_th, int min_th, int n_images, hipStream_t s);
void orbfe_launch_octree(const OctParams& p, int n_images, size_t lds_bytes, hipStream_t s);
void orbfe_launch_blur(const PyrView& src, const PyrView& dst, const BlurTile* tiles, int n_tiles, const BlurMfmaParams& mf,
                       int n_images, hipStream_t s);
// blur of the levels the tiles name, plus -- rz != nullptr -- the resize step into the next level from the same staged windows
void orbfe_launch_blur_level(const PyrView& src, const PyrView& dst, const BlurTile* tiles, int n_tiles, const LevelResize* rz,
                             int n_images, hipStream_t s);
void orbfe_launch_describe(const DescribeParams& p, int n_images, hipStream_t s);
size_t orbfe_octree_lds_bytes(int max_nodes, int lds_keys);
