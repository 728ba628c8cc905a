// extract_kernels.hip -- HIP kernels of the ORB extractor for gfx950 (MI355X, wave64).
//
// Kernel            replaces (reference file:line, L/ = Source/Libraries/ORB_SLAM2/)
// copy_level0       copyMakeBorder of the input into mvImagePyramid[0]  L/src/ORBextractor.cc:1061
// pyr_resize        cv::resize(INTER_LINEAR) chained level to level      L/src/ORBextractor.cc:1054
// fast_cells        per-cell cv::FAST(th=ini, else th=min) + NMS         L/src/ORBextractor.cc:756-791
// octree_select     DistributeOctTree + DivideNode                       L/src/ORBextractor.cc:475-731
// gauss_blur7       GaussianBlur(7x7, sigma 2, REFLECT_101)              L/src/ORBextractor.cc:1017-1019
// orient_describe   IC_Angle + computeOrbDescriptor + pt*=scale          L/src/ORBextractor.cc:76-146,1028-1035
//
// All arithmetic that decides an output bit is integer, or float/double evaluated exactly as the x86-64
// reference build does (no FMA contraction: this file is compiled with -ffp-contract=off; IEEE divide).
#include <stdlib.h>
#include <string.h>

#include "orbfe_internal.h"
#include "../../include/orb_pattern_data.h"

#define WAVE 64

// ------------------------------------------------------------------------------------------------ helpers
__device__ __forceinline__ int wave_incl_scan(int v) {
  // DPP row shifts + row broadcasts (gfx9): six VALU adds, no LDS crossbar round trips (ds_bpermute) as with __shfl_up
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);  // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);  // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);  // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);  // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1, 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2, 3
  return v;
}

// Inclusive scan over the T threads of a block (thread order = threadIdx.x; T a multiple of 64, <= 1024).  tmp: >= 16 ints of LDS.
// Returns the inclusive prefix; *total receives the block sum.  Contains two __syncthreads().
template <int T>
__device__ __forceinline__ int block_incl_scan_t(int v, int* tmp, int* total) {
  const int tid = threadIdx.x;
  const int lane = tid & (WAVE - 1), wid = tid >> 6;
  int s = wave_incl_scan(v);
  if (T == WAVE) {   // one wave: no LDS, no barrier
    *total = __builtin_amdgcn_readlane(s, WAVE - 1);
    return s;
  }
  __syncthreads();  // protect tmp from a previous use
  if (lane == WAVE - 1) tmp[wid] = s;
  __syncthreads();
  int off = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < T / WAVE; i++) {
    int t = tmp[i];
    if (i < wid) off += t;
    tot += t;
  }
  *total = tot;
  return s + off;
}

// The same with ONE barrier: consecutive calls alternate between two slots of `tmp` (2 * T / WAVE ints, used by nothing else), so a
// call's writes cannot meet the reads of the call before it -- those lie in front of this call's predecessor's barrier.
template <int T>
__device__ __forceinline__ int block_incl_scan_alt(int v, int* tmp, int& slot, int* total) {
  const int tid = threadIdx.x;
  const int lane = tid & (WAVE - 1), wid = tid >> 6;
  const int s = wave_incl_scan(v);
  int* t = tmp + slot * (T / WAVE);
  slot ^= 1;
  if (lane == WAVE - 1) t[wid] = s;
  __syncthreads();
  int off = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < T / WAVE; i++) {
    const int x = t[i];
    if (i < wid) off += x;
    tot += x;
  }
  *total = tot;
  return s + off;
}

__device__ __forceinline__ int reflect101(int p, int len) {
  if (len == 1) return 0;
  while (p < 0 || p >= len) p = (p < 0) ? -p : 2 * (len - 1) - p;
  return p;
}

// ------------------------------------------------------------------------------------------------ level 0
// Copies the caller's images (arbitrary stride/alignment) into the pitched, 256-byte aligned level-0 planes.
// 16 destination bytes per thread: five aligned source dwords + v_alignbyte instead of 16 byte loads (the source rows
// start at arbitrary alignment); the last chunk of a row takes the byte path and replicates the last pixel into the
// pitch padding.
__global__ __launch_bounds__(256) void copy_level0_kernel(const uint8_t* __restrict__ src, int sstride,
                                                           unsigned long long simg, uint8_t* __restrict__ dst,
                                                           int dpitch, unsigned long long dimg, int w, int h, int tiled) {
  const int nchunk = (w + 15) >> 4;  // 16-byte chunks per row; (row, chunk) pairs are dealt to threads in raster order
  const int item = blockIdx.x * 256 + threadIdx.x;
  const int y = (int)((item + 0.5f) * (1.0f / (float)nchunk));  // exact for item < 2^22
  const int x16 = (item - y * nchunk) * 16;
  if (y >= h) return;
  const uint8_t* S = src + (size_t)blockIdx.z * simg + (size_t)y * sstride;
  uint32_t out[4];
  if (x16 + 20 <= w) {
    const uintptr_t A = reinterpret_cast<uintptr_t>(S + x16);
    const uint32_t sh = (uint32_t)(A & 3);
    const uint32_t* B = reinterpret_cast<const uint32_t*>(A - sh);
    uint32_t d[5];
#pragma unroll
    for (int i = 0; i < 5; i++) d[i] = B[i];
#pragma unroll
    for (int i = 0; i < 4; i++) out[i] = __builtin_amdgcn_alignbyte(d[i + 1], d[i], sh);
  } else {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      out[i] = 0;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int x = x16 + 4 * i + j < w ? x16 + 4 * i + j : w - 1;
        out[i] |= (uint32_t)S[x] << (8 * j);
      }
    }
  }
  if (tiled) {   // the copy is written in 16 x 8 tiles like the levels behind it (orbfe_internal.h): a chunk is one tile row
    uint8_t* D = dst + (size_t)blockIdx.z * dimg + orbfe_tiled_offset(x16, y, dpitch);   // dpitch is a multiple of 64: the chunk is whole
    *reinterpret_cast<uint4*>(D) = make_uint4(out[0], out[1], out[2], out[3]);
    return;
  }
  uint8_t* D = dst + (size_t)blockIdx.z * dimg + (size_t)y * dpitch + x16;
  if (x16 + 16 <= dpitch) *reinterpret_cast<uint4*>(D) = make_uint4(out[0], out[1], out[2], out[3]);
  else
    for (int i = 0; i < 4 && x16 + 4 * i < dpitch; i++) reinterpret_cast<uint32_t*>(D)[i] = out[i];
}

// ------------------------------------------------------------------------------------------------ resize
// cv::resize INTER_LINEAR 8UC1, fixed point: horizontal taps sum 2048 (int32 row), vertical
// ((b*(T>>4))>>16 summed, +2, >>2).  4 destination pixels per thread, one dword store.
__global__ __launch_bounds__(256) void pyr_resize_kernel(const uint8_t* __restrict__ src, int spitch,
                                                          unsigned long long simg, uint8_t* __restrict__ dst,
                                                          int dpitch, unsigned long long dimg, int dw, int dh,
                                                          const ResizeTap* __restrict__ xt,
                                                          const ResizeTap* __restrict__ yt) {
  const int x4 = (blockIdx.x * 64 + threadIdx.x) * 4;
  const int y = blockIdx.y * 4 + threadIdx.y;
  if (x4 >= dw || y >= dh) return;
  const uint8_t* S = src + (size_t)blockIdx.z * simg;
  const ResizeTap ty = yt[y];
  const uint8_t* S0 = S + (size_t)ty.s0 * spitch;
  const uint8_t* S1 = S + (size_t)ty.s1 * spitch;
  const int b0 = ty.c0, b1 = ty.c1;
  uint32_t out = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int dx = x4 + i < dw ? x4 + i : dw - 1;
    const ResizeTap tx = xt[dx];
    const int t0 = S0[tx.s0] * tx.c0 + S0[tx.s1] * tx.c1;
    const int t1 = S1[tx.s0] * tx.c0 + S1[tx.s1] * tx.c1;
    const int v = (((b0 * (t0 >> 4)) >> 16) + ((b1 * (t1 >> 4)) >> 16) + 2) >> 2;
    out |= (uint32_t)(v & 0xff) << (8 * i);
  }
  *reinterpret_cast<uint32_t*>(dst + (size_t)blockIdx.z * dimg + (size_t)y * dpitch + x4) = out;
}

#define RS_COLS 560   // bytes of a staged source row: 256 destination pixels x scale <= 2, plus alignment slack
// LDS-staged resize: a workgroup produces a 256 x 16 destination tile; the source window it needs (at most RS2_ROWS rows x
// RS_COLS bytes for scale factors up to 2) is fetched with aligned 16-byte loads (one division per thread, all loads in
// flight before the first LDS store), the per-pixel taps then gather single bytes from LDS.  4 x 4 destination pixels per
// thread: the horizontal taps and all address arithmetic are loaded once and reused for four rows.  Same arithmetic and
// tables as pyr_resize_kernel (the direct-gather fallback for larger scale factors).
#define RS2_ROWS 34
__global__ __launch_bounds__(256) void pyr_resize_lds16_kernel(const uint8_t* __restrict__ src, int spitch,
                                                                unsigned long long simg, uint8_t* __restrict__ dst,
                                                                int dpitch, unsigned long long dimg, int dw, int dh,
                                                                const ResizeTap* __restrict__ xt,
                                                                const ResizeTap* __restrict__ yt) {
  __shared__ __attribute__((aligned(16))) uint8_t tile[RS2_ROWS * RS_COLS];
  const int tid = threadIdx.y * 64 + threadIdx.x;
  const int x0 = blockIdx.x * 256, y0 = blockIdx.y * 16;
  const int xl = min(x0 + 255, dw - 1), yl = min(y0 + 15, dh - 1);
  const int sxa = xt[x0].s0 & ~15;
  const int ncols16 = ((xt[xl].s1 - sxa) >> 4) + 1;  // <= 35
  const int sy_first = yt[y0].s0;
  const int nrows = yt[yl].s1 - sy_first + 1;        // <= 34
  const uint8_t* S = src + (size_t)blockIdx.z * simg;
  {
    const int rpp = 256 / ncols16;  // rows per pass, >= 7
    const int r0 = (int)((tid + 0.5f) * (1.0f / (float)ncols16));
    const int c = tid - r0 * ncols16;
    if (r0 < rpp) {
      const uint8_t* g = S + (size_t)sy_first * spitch + sxa + 16 * c;
      uint4 v[5];
#pragma unroll
      for (int k = 0; k < 5; k++) {
        const int r = r0 + k * rpp;
        v[k] = r < nrows ? *reinterpret_cast<const uint4*>(g + (size_t)r * spitch) : make_uint4(0, 0, 0, 0);
      }
#pragma unroll
      for (int k = 0; k < 5; k++) {
        const int r = r0 + k * rpp;
        if (r < nrows) *reinterpret_cast<uint4*>(tile + r * RS_COLS + 16 * c) = v[k];
      }
    }
  }
  __syncthreads();
  const int x4 = x0 + threadIdx.x * 4;
  if (x4 >= dw) return;
  ResizeTap tx[4];
#pragma unroll
  for (int i = 0; i < 4; i++) tx[i] = xt[min(x4 + i, dw - 1)];
  uint8_t* D = dst + (size_t)blockIdx.z * dimg + x4;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int y = y0 + threadIdx.y * 4 + k;
    if (y >= dh) break;
    const ResizeTap ty = yt[y];
    const uint8_t* S0 = tile + (ty.s0 - sy_first) * RS_COLS - sxa;
    const uint8_t* S1 = tile + (ty.s1 - sy_first) * RS_COLS - sxa;
    const int b0 = ty.c0, b1 = ty.c1;
    uint32_t out = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int t0 = S0[tx[i].s0] * tx[i].c0 + S0[tx[i].s1] * tx[i].c1;
      const int t1 = S1[tx[i].s0] * tx[i].c0 + S1[tx[i].s1] * tx[i].c1;
      const int v = (((b0 * (t0 >> 4)) >> 16) + ((b1 * (t1 >> 4)) >> 16) + 2) >> 2;
      out |= (uint32_t)(v & 0xff) << (8 * i);
    }
    *reinterpret_cast<uint32_t*>(D + (size_t)y * dpitch) = out;
  }
}

// bits 32..47 of the product of two 24-bit operands
__device__ __forceinline__ uint32_t mulhi_u24(uint32_t a, uint32_t b) {
  uint32_t r;
  asm("v_mul_hi_u32_u24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ uint32_t mulhi_u24_s(uint32_t a_uniform, uint32_t b) {   // first operand from a scalar register
  uint32_t r;
  asm("v_mul_hi_u32_u24 %0, %1, %2" : "=v"(r) : "s"(a_uniform), "v"(b));
  return r;
}

// Same tile, staging and arithmetic as pyr_resize_lds16_kernel, restated twice over:
// (1) per-pixel work for the vector ALU:
//   * a thread's four destination pixels read source columns s0 .. s0+7 of a row (scale factors <= 2): three aligned LDS
//     dwords, two v_alignbyte to start the window at s0, and per pixel one v_perm (bytes s0_i, s0_i+1 into 16-bit fields)
//     + one v_dot2_u32_u16 against (c0, c1) -- instead of two single-byte LDS gathers and two multiplies per row;
//   * the horizontal result of a source row is computed once and reused by the next destination row when that row's upper
//     tap is this row's lower one (a wave = one destination row, so the test is wave-uniform);
//   * (b * (t >> 4)) >> 16 == mul_hi_u24(b << 12, t & ~15): one full-rate instruction per tap.
//   At the right border the table gives s1 = s0 with c1 = 0 (build_taps): byte s0+1 is then whatever follows in LDS, times 0.
// (2) persistent workgroups: with one 4 KB tile per workgroup the seven launches of a pyramid ran at the rate workgroups can
//   be dispatched (~630 per microsecond, measured with the body removed), not at the memory rate.  Here a workgroup walks
//   tiles L, L + gridDim.x, ... and the source window of the NEXT tile is already in flight (held in registers) while the
//   current one is interpolated out of LDS.
struct ResizeGeom {
  int x0, y0, bz, sxa, ncols16, sy_first, nrows;
};
template <int TROWS, int SROWS, int COLS>   // COLS: bytes of a staged source row (RS_COLS for scale factors up to 2; 336 when every window fits 21 chunks)
__global__ __launch_bounds__(256) void pyr_resize_dot_kernel(const uint8_t* __restrict__ src, int spitch,
                                                              unsigned long long simg, uint8_t* __restrict__ dst,
                                                              int dpitch, unsigned long long dimg, int dw, int dh,
                                                              const ResizeTap* __restrict__ xt,
                                                              const ResizeTap* __restrict__ yt, int tiles_x, int tiles_y,
                                                              int n_tiles) {
  __shared__ __attribute__((aligned(16))) uint8_t tile[SROWS * COLS + 16];
  constexpr int NLD = (SROWS + 256 / (COLS / 16) - 1) / (256 / (COLS / 16));   // staging passes: at least 256 / (COLS / 16) rows per pass
  const int lane = threadIdx.x, wv = __builtin_amdgcn_readfirstlane((int)threadIdx.y);
  const int tid = wv * 64 + lane;
  auto geom = [&](int t) {
    ResizeGeom g;
    const int bx = t % tiles_x, r = t / tiles_x;
    const int by = r % tiles_y;
    g.bz = r / tiles_y;
    g.x0 = bx * 256;
    g.y0 = by * TROWS;
    const int xl = min(g.x0 + 255, dw - 1), yl = min(g.y0 + TROWS - 1, dh - 1);
    g.sxa = xt[g.x0].s0 & ~15;
    g.ncols16 = ((xt[xl].s1 - g.sxa) >> 4) + 1;   // <= 35
    g.sy_first = yt[g.y0].s0;
    g.nrows = yt[yl].s1 - g.sy_first + 1;         // <= SROWS
    return g;
  };
  // thread -> (row r0 + k * rpp, 16-byte column c) of the window: one division per tile, all loads issued back to back
  auto fetch = [&](const ResizeGeom& g, uint4 (&v)[NLD], int& r0, int& c, int& rpp) {
    rpp = 256 / g.ncols16;
    r0 = (int)((tid + 0.5f) * (1.0f / (float)g.ncols16));
    c = tid - r0 * g.ncols16;
    if (r0 < rpp) {
      const uint8_t* p = src + (size_t)g.bz * simg + (size_t)g.sy_first * spitch + g.sxa + 16 * c;
#pragma unroll
      for (int k = 0; k < NLD; k++) {
        const int r = r0 + k * rpp;
        v[k] = r < g.nrows ? *reinterpret_cast<const uint4*>(p + (size_t)r * spitch) : make_uint4(0, 0, 0, 0);
      }
    }
  };
  typedef __attribute__((ext_vector_type(2))) unsigned short us2;

  int t = (int)blockIdx.x;
  if (t >= n_tiles) return;
  ResizeGeom g = geom(t);
  uint4 v[NLD];
  int r0, c, rpp;
  fetch(g, v, r0, c, rpp);
  for (;;) {
    // horizontal taps of this tile (L2 hits; overlap the wait for the window)
    const int x4 = g.x0 + lane * 4;
    uint2 txr[4];
#pragma unroll
    for (int i = 0; i < 4; i++) txr[i] = reinterpret_cast<const uint2*>(xt)[min(x4 + i, dw - 1)];
    // and its vertical taps: loads return in order, so everything the interpolation needs is requested BEFORE the next
    // tile's window -- a load issued after the prefetch would have to wait for it
    uint2 tyr[TROWS / 4];
#pragma unroll
    for (int k = 0; k < TROWS / 4; k++) tyr[k] = reinterpret_cast<const uint2*>(yt)[min(g.y0 + wv * (TROWS / 4) + k, dh - 1)];
    if (r0 < rpp) {
#pragma unroll
      for (int k = 0; k < NLD; k++) {
        const int r = r0 + k * rpp;
        if (r < g.nrows) *reinterpret_cast<uint4*>(tile + r * COLS + 16 * c) = v[k];
      }
    }
    __syncthreads();
    const ResizeGeom cur = g;
    const int tn = t + (int)gridDim.x;
    if (tn < n_tiles) {
      g = geom(tn);
      fetch(g, v, r0, c, rpp);
    }
    if (x4 < dw) {
      const int s00 = (int)(int16_t)(txr[0].x & 0xffff);
      const int base = s00 - cur.sxa;
      const uint32_t sh = (uint32_t)base & 3u;
      uint32_t sel[4], cp[4];
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const uint32_t o = (uint32_t)((int)(int16_t)(txr[i].x & 0xffff) - s00);   // 0 .. 6
        sel[i] = o | 0x0c000c00u | ((o + 1) << 16);
        cp[i] = txr[i].y;                                                         // c0 | c1 << 16
      }
      const uint32_t* trow = reinterpret_cast<const uint32_t*>(tile + (base & ~3));
      auto hrow = [&](int r, uint32_t (&hh)[4]) {
        const uint32_t* p = trow + r * (COLS / 4);
        const uint32_t* p2 = p + 2;
        asm("" : "+v"(p2));   // keep the third dword a separate ds_read_b32 (a 4-byte aligned ds_read_b96 is slow)
        const uint32_t w0 = p[0], w1 = p[1], w2 = *p2;
        const uint32_t W0 = __builtin_amdgcn_alignbyte(w1, w0, sh), W1 = __builtin_amdgcn_alignbyte(w2, w1, sh);
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const uint32_t u = __builtin_amdgcn_perm(W1, W0, sel[i]);
          hh[i] = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, u), __builtin_bit_cast(us2, cp[i]), 0u, false) & ~15u;
        }
      };
      uint8_t* D = dst + (size_t)cur.bz * dimg + x4;
      uint32_t hp[4] = {0, 0, 0, 0};
      int prev = -1;
#pragma unroll
      for (int k = 0; k < TROWS / 4; k++) {
        const int y = cur.y0 + wv * (TROWS / 4) + k;
        if (y >= dh) break;
        const uint32_t tys = (uint32_t)__builtin_amdgcn_readfirstlane((int)tyr[k].x);
        const uint32_t tyc = (uint32_t)__builtin_amdgcn_readfirstlane((int)tyr[k].y);
        const int ra = (int)(int16_t)(tys & 0xffff) - cur.sy_first, rb = (int)(int16_t)(tys >> 16) - cur.sy_first;
        uint32_t h0[4], h1[4];
        if (ra == prev) {
#pragma unroll
          for (int i = 0; i < 4; i++) h0[i] = hp[i];
        } else {
          hrow(ra, h0);
        }
        if (rb == ra) {
#pragma unroll
          for (int i = 0; i < 4; i++) h1[i] = h0[i];
        } else {
          hrow(rb, h1);
        }
        const uint32_t B0 = (tyc & 0xffffu) << 12, B1 = (tyc >> 16) << 12;
        uint32_t sm[4];
#pragma unroll
        for (int i = 0; i < 4; i++) sm[i] = mulhi_u24(B0, h0[i]) + mulhi_u24(B1, h1[i]) + 2u;
        const uint32_t P01 = (sm[0] | (sm[1] << 16)) >> 2, P23 = (sm[2] | (sm[3] << 16)) >> 2;   // values <= 255 in bytes 0 and 2
        *reinterpret_cast<uint32_t*>(D + (size_t)y * dpitch) = __builtin_amdgcn_perm(P23, P01, 0x06040200u);
#pragma unroll
        for (int i = 0; i < 4; i++) hp[i] = h1[i];
        prev = rb;
      }
    }
    if (tn >= n_tiles) break;
    t = tn;
    __syncthreads();   // every wave is done reading this tile before the next window is written over it
  }
}

// ------------------------------------------------------------------------------------------------ FAST
// ring offsets of FAST-9/16 (OpenCV makeOffsets, patternSize 16)
__device__ constexpr int RDX[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
__device__ constexpr int RDY[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};

// packed (2 x u16) min / max
__device__ __forceinline__ uint32_t pkmin(uint32_t a, uint32_t b) {
  uint32_t r;
  asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ uint32_t pkmax(uint32_t a, uint32_t b) {
  uint32_t r;
  asm("v_pk_max_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// packed (2 x i16) helpers of the pair-wise corner score below
__device__ __forceinline__ uint32_t pkmin_i(uint32_t a, uint32_t b) {
  uint32_t r;
  asm("v_pk_min_i16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ uint32_t pkmax_i(uint32_t a, uint32_t b) {
  uint32_t r;
  asm("v_pk_max_i16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ uint32_t pksub_i(uint32_t a, uint32_t b) {
  uint32_t r;
  asm("v_pk_sub_i16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// max over the sixteen 9-arcs of the arc minimum of t[k] = (v - q_k) ^ m, with the ring held as eight (q_2j, q_2j+1) pairs.
// m = 0: the dark score + 1 of cornerScore<16>.  m = ~0: t = q - v - 1 (bitwise NOT reverses the order), so the result is the
// bright score + 1, minus 1.  One network serves either polarity without a divergent branch around it.
__device__ __forceinline__ int arc9_maxmin_pk(const uint32_t (&P)[8], uint32_t VV, uint32_t m) {
  uint32_t D[8], L[8];
#pragma unroll
  for (int j = 0; j < 8; j++) D[j] = pksub_i(VV, P[j]) ^ m;
#pragma unroll
  for (int j = 0; j < 8; j++) L[j] = pkmin_i(D[j], __builtin_amdgcn_alignbit(D[(j + 1) & 7], D[j], 16));   // min d[i .. i+1]
  uint32_t L4[8];
#pragma unroll
  for (int j = 0; j < 8; j++) L4[j] = pkmin_i(L[j], L[(j + 1) & 7]);                                       // min d[i .. i+3]
#pragma unroll
  for (int j = 0; j < 8; j++) L[j] = pkmin_i(pkmin_i(L4[j], L4[(j + 2) & 7]), D[(j + 4) & 7]);             // min d[i .. i+8]
  const uint32_t A = pkmax_i(pkmax_i(pkmax_i(L[0], L[1]), pkmax_i(L[2], L[3])), pkmax_i(pkmax_i(L[4], L[5]), pkmax_i(L[6], L[7])));
  return max((int)(short)(A & 0xffffu), (int)(short)(A >> 16));
}

// ---- FAST, second formulation (default): one WAVE per cell, no workgroup barriers.
//
// The phases of the per-cell pipeline (stage, quick test, exact score, NMS, ordered emission) have very different widths; in a
// workgroup-wide kernel every phase boundary is a barrier at which most waves idle.  Here every wave owns a run of <= 4
// horizontally adjacent cells and walks them one at a time entirely inside its own LDS slice (~5.4 KB: 28 waves per CU): all
// synchronisation is the in-order execution of one wave's LDS operations and a compute unit holds independent waves in
// different phases.  63 VGPRs; 7 waves per SIMD is what the LDS slices allow; holding the NEXT cell's tile in registers while
// the current one is processed costs 15 registers = one wave per SIMD and measured slower (DESIGN lessons 13, 18).
//   stage   cell ROI (<= 66 x 66) as bytes, shifted one column when that makes the tested region start on an even column (the
//           shift happens in registers: v_alignbyte over one extra aligned dword; LDS stores stay 16-byte aligned)
//   A1      SWAR quick test, two pixels per lane in 16-bit fields: the 16-bit pairs are cut out of aligned dwords with
//           v_perm_b32 (selectors per lane: the pair starts at byte 0 or 2), ten dwords per pair in five ds_read2_b32; packed
//           min / max over the four antipodal pairs, one biased subtract / add per polarity (see below).  Lanes = (row in a
//           band of RI rows, pixel pair): rows are 64 bytes apart, so the lanes of a 32-lane LDS group hit distinct banks.
//           Two tiers: the antipodal pairs (0, 8), (4, 12) first, the other two only if some lane of the wave passes.
//           Survivors -> 384-entry list with their polarity, scored (A2) and emptied whenever more than 256 are waiting.
//           The whole pass runs at iniThFAST first and is repeated at minThFAST (from the tile that is still staged) only for a
//           cell that kept nothing: the reference's own order (L/src/ORBextractor.cc:773-780)
//   A2      exact cornerScore of the surviving polarity: ring held as eight packed pairs, one 9-arc max-min network of packed
//           16-bit min / max for either polarity (arc9_maxmin_pk) -> score plane (tested region + 1-pixel zero frame)
//   B       strict 3x3 NMS inside the cell with lane = survivor (the list still holds every scored pixel; nine LDS reads in flight
//           per lane); kept pixels set bits, one 32-bit word per tested row.  Cells whose list was recycled (> 256 survivors)
//           walk the bitmap instead (lane = word)
//   C       row-major emission from the bitmap, one packed wave scan for the output offsets
#ifndef FC_LIST_CAP
#define FC_LIST_CAP 384
#endif
#ifndef FC_LDS_PAD
#define FC_LDS_PAD 0     // experiment: unused bytes per wave slice (what the kernel's LDS footprint does to its neighbours on the CU)
#endif
#ifndef FC_T2_QUEUE
#define FC_T2_QUEUE 1    // tier 2 of the quick test over QUEUED tier-1 survivors, 64 pairs at a time (0: in place, wave-uniform branch: rounds 3-5)
#endif
#define FC_Q1_CAP 128    // the ring holds < 64 entries between rounds and an iteration adds <= 64
#ifndef FC_T1_ASM
#define FC_T1_ASM 1      // the tier-1 loop as one block of assembly (0: the compiler's loop)
#endif
#ifndef FC_TIMING
#define FC_TIMING 0
#endif
#if FC_TIMING
// -DFC_TIMING=1 instrumentation (tools/fc_phase_profile.py): wave-cycles per phase (wait for pixels, stage + clear, A1, A2, NMS,
// scan + emit) by s_memtime, accumulated per wave in scalars and added to one of 4096 slots at the end (one slot would
// serialise 660 k atomics on one cache line: the kernel took 6 ms)
__device__ unsigned long long g_fc_prof[4096 * 8];   // 4096 slots (spread the atomics), summed on the host
#define FC_T(i) do { const uint32_t _t = (uint32_t)__builtin_readcyclecounter(); tacc##i += _t - tprev; tprev = _t; } while (0)
extern "C" int orbfe_debug_fc_profile(unsigned long long* out, int reset) {
  static unsigned long long h[4096 * 8];
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_fc_prof), sizeof(h)) != hipSuccess) return 1;
  for (int i = 0; i < 8; i++) out[i] = 0;
  for (int sl = 0; sl < 4096; sl++)
    for (int i = 0; i < 8; i++) out[i] += h[sl * 8 + i];
  if (reset) { memset(h, 0, sizeof(h)); if (hipMemcpyToSymbol(HIP_SYMBOL(g_fc_prof), h, sizeof(h)) != hipSuccess) return 1; }
  return 0;
}
#else
#define FC_T(i)
#endif
#ifndef FC_SCALAR_CELL
#define FC_SCALAR_CELL 1   // a cell's descriptor through the scalar cache (eight dwords) instead of the compiler's vector loads of its 16-bit fields
#endif
#ifndef FC_PE_COND
#define FC_PE_COND 1   // the dword in front of a chunk is loaded only for cells whose tile is shifted by a column (-1 %)
#endif
#ifndef FC_WAVES_PER_EU
#define FC_WAVES_PER_EU 7
#endif
#ifndef FC_WG_WAVES
#define FC_WG_WAVES 4   // waves (= runs of cells) per workgroup; the waves share nothing, but a workgroup's LDS and wave slots are held until its last wave ends
#endif
// LDS bytes of one wave's slice: byte tile, score plane, bitmap of scored pixels (nbw words), survivor list
__host__ __device__ inline int fc_wave_lds(int rows_max, int pb, int sc_bytes, int nbw) {
  return rows_max * pb + sc_bytes + nbw * 4 + FC_LIST_CAP * 2 + (FC_T2_QUEUE ? FC_Q1_CAP * 2 : 0) + FC_LDS_PAD;
}
template <int PB, int NLD>   // PB: bytes per staged row (64 or 96); NLD: load rounds of 64 lanes x 16 bytes per cell
__global__ __launch_bounds__(64 * FC_WG_WAVES, FC_WAVES_PER_EU) void fast_cells_kernel(PyrView pyr, const CellDesc* __restrict__ cells,
                                                          const FastGroup* __restrict__ runs, int n_runs, int total_cells,
                                                          int rows_max, int sc_bytes, int nbw, int32_t* __restrict__ cell_cnt,
                                                          uint32_t* __restrict__ slots, unsigned long long slots_per_image,
                                                          int ini_th, int min_th, int xcd_run_shift) {
  extern __shared__ __attribute__((aligned(16))) uint8_t fc_smem[];
  const int lane = threadIdx.x & (WAVE - 1), wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  uint8_t* tile = fc_smem + wid * fc_wave_lds(rows_max, PB, sc_bytes, nbw);
  uint8_t* sc = tile + rows_max * PB;
  uint32_t* scb = reinterpret_cast<uint32_t*>(sc + sc_bytes);   // bitmap of scored pixels, row-major over the tested region
  uint16_t* list = reinterpret_cast<uint16_t*>(scb + nbw);      // quick-test survivors (tile index | polarity)
#if FC_T2_QUEUE
  uint16_t* q1 = list + FC_LIST_CAP;                            // ring of tier-1 survivors: tile index of a lane's pixel pair
#endif
  int img = blockIdx.y;
  const int nblk = (n_runs + FC_WG_WAVES - 1) / FC_WG_WAVES;
  const int q = blockIdx.x >> 3;
  const int unit = 8 << (xcd_run_shift < 0 ? 0 : xcd_run_shift);
  int bid = (xcd_run_shift < 0 || (int)blockIdx.x >= (nblk / unit) * unit)
                ? (int)blockIdx.x
                : ((q >> xcd_run_shift) << (xcd_run_shift + 3)) + ((blockIdx.x & 7) << xcd_run_shift) +
                      (q & ((1 << xcd_run_shift) - 1));
  if (xcd_run_shift == -2) {
    // whole images per XCD (workgroups go to the XCDs round-robin in linear order): XCD j walks images j, j + 8, ... one after
    // the other, so the halo rows / 128-byte lines that neighbouring cells share are fetched into ONE L2 once
    const unsigned gx = gridDim.x;
    const unsigned lin = blockIdx.y * gx + blockIdx.x;
    const unsigned grp = lin / (8u * gx);
    if (8u * grp + 8u <= gridDim.y) {
      const unsigned within = lin - grp * 8u * gx;
      img = (int)(8u * grp + (within & 7u));
      bid = (int)(within >> 3);
    }
  }
  const int rid = bid * FC_WG_WAVES + wid;
  if (rid >= n_runs) return;   // no barriers below
  const FastGroup g = runs[rid];
  const int level = g.level;
  const int pitch = pyr.pitch[level];
  const uint8_t* plane = pyr.base[level] + (size_t)img * pyr.img_stride[level];
  const uint32_t tile_a = (uint32_t)(uintptr_t)tile;

#if FC_TIMING
  uint32_t tacc0 = 0, tacc1 = 0, tacc2 = 0, tacc3 = 0, tacc4 = 0, tacc5 = 0, tprev = (uint32_t)__builtin_readcyclecounter();
#endif
  // a cell's descriptor through the scalar cache (eight dwords at a wave-uniform address): left to itself the compiler reads the 16-bit
  // fields with vector loads, one memory round trip after the other
  static_assert(sizeof(CellDesc) == 32, "eight dwords");
  auto load_cell = [&](int idx) {
    const __attribute__((address_space(4))) uint32_t* q =
        (const __attribute__((address_space(4))) uint32_t*)(uintptr_t)(cells + __builtin_amdgcn_readfirstlane(idx));
    CellDesc c;
    uint32_t w[8];
#pragma unroll
    for (int j = 0; j < 8; j++) w[j] = q[j];
    __builtin_memcpy(&c, w, sizeof(c));
    return c;
  };
  for (int k = 0; k < g.n_cells; k++) {
    const CellDesc cd = FC_SCALAR_CELL ? load_cell(g.first_cell + k) : cells[g.first_cell + k];
    // ---- the cell's pixels: every lane loads in every round (row and chunk clamped into the cell: a duplicate load and,
    //      below, a duplicate LDS store of the same bytes cost nothing, a divergent branch around them does)
    uint4 pv[NLD];
    uint32_t pe[NLD];
    {
      const int ndq = ((cd.x0 & 15) + (((cd.x0 & 15) + 3) & 1) + cd.cols + 15) >> 4;   // chunks of the SHIFTED row
      const uint8_t* src = plane + (size_t)cd.y0 * pitch + (cd.x0 & ~15);
      // a 16-byte chunk of a TILED level is one tile row; the dword in front of it the last dword of the tile row to its left
      const bool ltiled = (pyr.tiled >> level) & 1u;
      const uint32_t tstep = (uint32_t)(pitch >> 4) << 7;
      auto chunk_ptr = [&](int r, int c) {
        const int y = cd.y0 + r;
        return ltiled ? plane + (uint32_t)(y >> 3) * tstep + ((uint32_t)(((cd.x0 & ~15) >> 4) + c) << 7) + (uint32_t)((y & 7) * 16)
                      : src + (size_t)r * pitch + 16 * c;
      };
      auto front_ptr = [&](int r, int c, const uint8_t* p) {   // x0 >= 16: never before the row
        return ltiled ? p - 128 + 12 : p - 4;
      };
      if constexpr (PB == 64) {
        // four 16-byte chunks per staged row: lane -> (row, chunk) by shift and mask, sixteen rows per round
        // (three-chunk cells -- seven of ten at KITTI -- in TWO rounds, lane -> (i / 3, i % 3): measured 0.4285-0.4318 ms against 0.4229-0.426;
        //  unlike the describe kernel's, this kernel's load instructions are not what its vector-memory path is busy with: round 6)
        const int c = min(lane & 3, ndq - 1);
#pragma unroll
        for (int kk = 0; kk < NLD; kk++) {
          const int r = min((lane >> 2) + 16 * kk, cd.rows - 1);
          const uint8_t* p = chunk_ptr(r, c);          pv[kk] = *reinterpret_cast<const uint4*>(p);
#if FC_PE_COND
          pe[kk] = ((cd.x0 & 15) + 3) & 1 ? *reinterpret_cast<const uint32_t*>(front_ptr(r, c, p)) : 0u;   // only a shifted tile needs it (wave-uniform)
#else
          pe[kk] = *reinterpret_cast<const uint32_t*>(front_ptr(r, c, p));   // the dword in front
#endif
        }
      } else {
        const int items = cd.rows * ndq;
        const float inv = 1.0f / (float)ndq;
#pragma unroll
        for (int kk = 0; kk < NLD; kk++) {
          const int i = min(lane + WAVE * kk, items - 1);
          const int r = (int)((i + 0.5f) * inv), c = i - r * ndq;
          const uint8_t* p = chunk_ptr(r, c);
          pv[kk] = *reinterpret_cast<const uint4*>(p);
          pe[kk] = *reinterpret_cast<const uint32_t*>(front_ptr(r, c, p));
        }
      }
    }
#if FC_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    FC_T(0);   // waiting for the cell's pixels
#endif
    const int cols = cd.cols, rows = cd.rows;
    const int xo = cd.x0 & 15, sh = (xo + 3) & 1, xs = xo + sh;
    const int ndq = (xs + cols + 15) >> 4;
    const int th = rows - 6, tw = cols - 6;
    const int c_lo = xs + 3, c_hi = xs + cols - 3;
    const int SCP = tw + 2;
    const int bsh = tw > 32 ? 6 : 5;   // bitmaps: one word (two for wide cells) per tested row, so a bit index splits by shift and mask
    // ---- stage (shifted by `sh` columns), clear the score plane and the bitmaps
    {
      const int items = rows * ndq;
      const float inv = 1.0f / (float)ndq;
#pragma unroll
      for (int kk = 0; kk < NLD; kk++) {
        int r, c;
        if constexpr (PB == 64) { r = min((lane >> 2) + 16 * kk, rows - 1); c = min(lane & 3, ndq - 1); }
        else { const int i = min(lane + WAVE * kk, items - 1); r = (int)((i + 0.5f) * inv); c = i - r * ndq; }
        uint4 d = pv[kk];
        if (sh) {   // pixel j of the chunk moves to column j + 1: the first byte comes from the dword in front
          d.w = __builtin_amdgcn_alignbyte(d.w, d.z, 3);
          d.z = __builtin_amdgcn_alignbyte(d.z, d.y, 3);
          d.y = __builtin_amdgcn_alignbyte(d.y, d.x, 3);
          d.x = __builtin_amdgcn_alignbyte(d.x, pe[kk], 3);
        }
        *reinterpret_cast<uint4*>(tile + r * PB + 16 * c) = d;
      }
      const int nsc = ((th + 2) * SCP + 15) >> 4;
      for (int i = lane; i < nsc; i += WAVE) reinterpret_cast<uint4*>(sc)[i] = make_uint4(0, 0, 0, 0);
      // (the bitmap is cleared where a cell first needs it: when its list is recycled)
    }
    FC_T(1);   // LDS staging + clears

    // ---- A2 (called when the list is nearly full -- `mark`: the list is about to be recycled, so scored pixels are also
    //      recorded in the bitmap -- and at the end): exact score of list[0 .. n), the polarity (or, about once in 10^4, the
    //      two polarities) the quick test left possible
    auto score_list = [&](int n, bool mark, int th_cur) {
      // A pixel whose two polarities both survived the quick test (a blurred edge through the centre: 3-8 % of the survivors
      // on the upper pyramid levels) needs both networks.  Running the second one under a wave-uniform branch cost a whole
      // network whenever ANY of the 64 lanes had such a pixel -- nearly every round.  Instead the pixel is scored as dark here and
      // appended to the list once more as bright-only (at most one of the two scores can reach the threshold: a ring has no room
      // for two 9-arcs); only when the list is full does the second network run in place.
      int n_end = n;
      for (int done = 0; done < n_end;) {
        const int start = done;
        const int lim = min(start + WAVE, n_end);   // this round's entries
        const int i = start + lane;
        done = lim;
        uint32_t en = i < lim ? list[i] : 0u;
        int e = (int)(en & 0x3fffu);
        int y = PB == 64 ? (e >> 6) : (int)((e + 0.5f) * (1.0f / PB));
        int x = e - y * PB;
        bool valid = i < lim && x < c_hi;   // not the pair straddling the right edge of the tested region
        const bool both = valid && (en & 0xC000u) == 0xC000u;
        const unsigned long long mb = __ballot(both);
        bool second_here = false;
        if (mb) {   // wave-uniform
          const int nb = __popcll(mb);
          if (n_end + nb <= FC_LIST_CAP) {
            const int r = __builtin_amdgcn_mbcnt_hi((uint32_t)(mb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mb, (uint32_t)n_end));
            if (both) list[r] = (uint16_t)((uint32_t)e | 0x4000u);
            if (lim == n_end && lim - start + nb <= WAVE) {
              // last round with idle lanes: they take the bright halves right away instead of forming a round of their own
              if (i >= lim && i < lim + nb) {
                en = list[i];
                e = (int)(en & 0x3fffu);
                y = PB == 64 ? (e >> 6) : (int)((e + 0.5f) * (1.0f / PB));
                x = e - y * PB;
                valid = true;
              }
              done = lim + nb;
            }
            n_end += nb;
          } else {
            second_here = true;
          }
        }
        if (!valid) continue;
        // the centre and the sixteen ring bytes, packed in ring order as eight (q_2j, q_2j+1) pairs (d16 byte loads would
        // build the pairs for free, but with SRAM ECC they zero the other half of the register)
        // (addressed from the ring's top-left corner: every offset is then a non-negative immediate of ds_read_u8, where
        // offsets relative to the centre cost eight address additions)
        int eb = e - (3 * PB + 3);
        asm("" : "+v"(eb));   // keep THIS base: the compiler would fold the constant back and re-create the negative offsets
        const uint8_t* c = tile + eb;
        const uint32_t v = c[3 * PB + 3];
        uint32_t P[8];
#pragma unroll
        for (int j = 0; j < 8; j++)
          P[j] = (uint32_t)c[(RDY[2 * j] + 3) * PB + RDX[2 * j] + 3] | ((uint32_t)c[(RDY[2 * j + 1] + 3) * PB + RDX[2 * j + 1] + 3] << 16);
        const uint32_t VV = v | (v << 16);
        const bool bright = (en & 0xC000u) == 0x4000u;   // bright only; dark first where both are possible
        int s = arc9_maxmin_pk(P, VV, bright ? ~0u : 0u) + (bright ? 1 : 0);
        if (second_here) {   // list full (wave-uniform): the bright network of the two-polarity pixels in place
          if (both) s = max(s, arc9_maxmin_pk(P, VV, ~0u) + 1);
        }
        if (s - 1 >= th_cur) {
          const int ty = y - 3, tx = x - c_lo;
          sc[(ty + 1) * SCP + tx + 1] = (uint8_t)(s - 1);
          if (mark) {
            const int p = (ty << bsh) + tx;
            atomicOr(&scb[p >> 5], 1u << (p & 31));
          }
        }
      }
    };

    // The reference's own order (L/src/ORBextractor.cc:773-780): cv::FAST(cell, iniThFAST) first, cv::FAST(cell, minThFAST)
    // only when that returned nothing.  Pass 0 runs quick test, score, NMS at ini_th; a cell whose pass 0 keeps no pixel
    // (wave-uniform) repeats them at min_th from the tile that is still staged.
    uint32_t keep[2] = {0u, 0u};
    int total = 0;
    bool was_flushed = false;
    for (int pass = 0; pass < 2; pass++) {
      const int th_cur = pass ? min_th : ini_th;
      const uint32_t C = (uint32_t)(0x8000 - th_cur - 1) * 0x00010001u;
      bool flushed = false;
      int wcnt = 0;
      // ---- A1
      if (th > 0 && tw > 0) {
        const int ppr = cd.ppr;                        // (tw + 1) >> 1, <= 30
        const int RI = cd.ri;                          // WAVE / ppr: rows per iteration, >= 2
        const int rl = (int)((lane + 0.5f) * cd.inv_ppr), pl = lane - rl * ppr;
        const bool lane_ok = rl < RI;
        const int n_it = cd.n_it;                      // (th + RI - 1) / RI
        const int x = c_lo + 2 * pl;                   // even: the pair (x, x + 1) starts at byte 0 or 2 of its dword
        const bool odd2 = (x & 2) != 0;
        const uint32_t selV = odd2 ? 0x0c030c02u : 0x0c010c00u;    // (x, x+1) and (x, x+1) of rows +-3
#if !FC_T2_QUEUE
        const uint32_t selX = odd2 ? 0x0c010c00u : 0x0c030c02u;    // (x+-2, x+-2+1)
#endif
        const uint32_t selM = odd2 ? 0x0c040c03u : 0x0c020c01u;    // (x-3, x-2) out of dwords {D(x)-1, D(x)}
        const uint32_t selP = odd2 ? 0x0c020c01u : 0x0c040c03u;    // (x+3, x+4) out of dwords {D(x+2), D(x+2)+1}
        // addresses of row (y - 3): a0 = the dword of x minus one dword (the centre row reads {D-1, D}); a1 = the dword of
        // x - 2, which is also the dword of x + 2 minus one dword (x = 4k: D(x) - 1; x = 4k + 2: D(x))
        uint32_t a0 = tile_a + rl * PB + ((x >> 2) << 2) - 4;
        uint32_t a1 = a0 + (odd2 ? 4u : 0u);
#if !(FC_T2_QUEUE && FC_T1_ASM)
        const uint32_t ec = (uint32_t)(3 * PB + 4 + (x & 3)) - tile_a;   // byte index of the pair's first pixel = a0 + ec
        int rows_left = lane_ok ? th - rl : 0;         // lane active while rows_left > 0
#endif
#if FC_T2_QUEUE && FC_T1_ASM
        // Tier 1 runs over all the cell's pixels; the lanes that pass it (a fifth of the 64 on a triggered iteration) used to take the
        // whole wave through tier 2 and the survivor push (35 vector instructions at ~20 % lane use).  Here they only queue their
        // pair (`q1`, first in first out = row-major order: the LDS address of the pair's window | its column parity); tier 2 + push
        // run over 64 QUEUED pairs at a time: whenever the queue holds a wave's worth, and once at the end of the cell (the tile is
        // restaged for the next one).
        // The tier-1 loop itself is one block of assembly: measured on this kernel a scalar instruction costs what a vector
        // instruction costs (eight s_add per iteration: + 3.1 % alone, + 1.1 % on the step; eight v_add: + 3.5 % / + 1.3 %), and the
        // compiler's loop carried ~20 of them per iteration (loop-carried conditions materialised as 64-bit masks, exec save /
        // restore around the queue store, a branch per condition).  Per iteration here: 3 LDS reads, 19 vector, 8 scalar
        // instructions, + 4 / 4 / 1 on the 57 % of the iterations in which a lane passes.
        const unsigned long long m_full = __ballot(lane_ok), m_last = __ballot(lane_ok && rl < th - (n_it - 1) * RI);
        const uint32_t oddbit = odd2 ? 1u : 0u;
        const uint32_t q_base = (uint32_t)(uintptr_t)q1, q_limit = q_base + 2u * WAVE;
        uint32_t q_addr = q_base;
        int it = 0;
        for (int phase = 0; phase < 2; phase++) {
          const unsigned long long actm = phase ? m_last : m_full;
          const int nit = __builtin_amdgcn_readfirstlane(phase ? n_it : n_it - 1);   // (uniform values the compiler keeps in vector registers)
          const int step = __builtin_amdgcn_readfirstlane(RI * PB);
          for (;;) {
            if (it < nit) {
              uint32_t tV, tA, tB, tC, tD, tE;
              uint32_t sc_;
              asm volatile(
                  "1:\n\t"
                  "ds_read2_b32 v[64:65], %[a0] offset0:1 offset1:%[K61]\n\t"
                  "ds_read2_b32 v[66:67], %[a0] offset0:%[K30] offset1:%[K31]\n\t"
                  "ds_read2_b32 v[68:69], %[a1] offset0:%[K31] offset1:%[K32]\n\t"
                  "s_add_u32 %[it], %[it], 1\n\t"
                  "s_waitcnt lgkmcnt(0)\n\t"
                  "v_perm_b32 %[tV], 0, v67, %[selV]\n\t"          // V: the centre pair
                  "v_perm_b32 %[tA], 0, v64, %[selV]\n\t"          // Q8 (row y - 3)
                  "v_perm_b32 %[tB], 0, v65, %[selV]\n\t"          // Q0 (row y + 3)
                  "v_perm_b32 %[tC], v67, v66, %[selM]\n\t"        // Q12 (x - 3, x - 2)
                  "v_perm_b32 %[tD], v69, v68, %[selP]\n\t"        // Q4 (x + 3, x + 4)
                  "v_pk_min_u16 %[tE], %[tB], %[tA]\n\t"           // min(Q0, Q8)
                  "v_pk_max_u16 %[tA], %[tB], %[tA]\n\t"           // max(Q0, Q8)
                  "v_pk_min_u16 %[tB], %[tD], %[tC]\n\t"           // min(Q4, Q12)
                  "v_pk_max_u16 %[tC], %[tD], %[tC]\n\t"           // max(Q4, Q12)
                  "v_pk_max_u16 %[tE], %[tE], %[tB]\n\t"           // lo
                  "v_pk_min_u16 %[tA], %[tA], %[tC]\n\t"           // hi
                  "v_add_u32 %[tB], %[C], %[tV]\n\t"               // AD = V + C
                  "v_sub_u32 %[tC], %[C], %[tV]\n\t"               // AB = C - V
                  "v_sub_u32 %[tB], %[tB], %[tE]\n\t"              // AD - lo
                  "v_add_u32 %[tC], %[tA], %[tC]\n\t"              // hi + AB
                  "v_bitop3_b32 %[tB], %[tB], %[M], %[tC] bitop3:0xc8\n\t"   // (dark | bright) & 0x80008000
                  "v_cmp_ne_u32 vcc, 0, %[tB]\n\t"
                  "s_and_b64 vcc, vcc, %[actm]\n\t"
                  "s_cbranch_scc0 2f\n\t"
                  "v_mbcnt_lo_u32_b32 %[tA], vcc_lo, 0\n\t"
                  "v_mbcnt_hi_u32_b32 %[tA], vcc_hi, %[tA]\n\t"
                  "v_or_b32 %[tC], %[a0], %[odd]\n\t"
                  "v_lshl_add_u32 %[tA], %[tA], 1, %[qa]\n\t"
                  "s_mov_b64 exec, vcc\n\t"
                  "ds_write_b16 %[tA], %[tC]\n\t"
                  "s_mov_b64 exec, -1\n\t"
                  "s_bcnt1_i32_b64 %[sc], vcc\n\t"
                  "s_lshl1_add_u32 %[qa], %[sc], %[qa]\n\t"
                  "2:\n\t"
                  "v_add_u32 %[a0], %[step], %[a0]\n\t"
                  "v_add_u32 %[a1], %[step], %[a1]\n\t"
                  "s_cmp_ge_u32 %[qa], %[ql]\n\t"
                  "s_cbranch_scc1 3f\n\t"
                  "s_cmp_lt_i32 %[it], %[nit]\n\t"
                  "s_cbranch_scc1 1b\n\t"
                  "3:"
                  : [a0] "+v"(a0), [a1] "+v"(a1), [it] "+s"(it), [qa] "+s"(q_addr), [tV] "=&v"(tV), [tA] "=&v"(tA), [tB] "=&v"(tB),
                    [tC] "=&v"(tC), [tD] "=&v"(tD), [tE] "=&v"(tE), [sc] "=&s"(sc_)
                  : [selV] "v"(selV), [selM] "v"(selM), [selP] "v"(selP), [odd] "v"(oddbit), [C] "s"(C), [M] "s"(0x80008000u),
                    [actm] "s"(actm), [step] "s"(step), [ql] "s"(q_limit), [nit] "s"(nit),
                    [K61] "n"(6 * PB / 4 + 1), [K30] "n"(3 * PB / 4), [K31] "n"(3 * PB / 4 + 1), [K32] "n"(3 * PB / 4 + 2)
                  : "memory", "vcc", "scc", "v64", "v65", "v66", "v67", "v68", "v69");
            }
            const int qn = (int)(q_addr - q_base) >> 1;
            if (qn < WAVE && !(phase == 1 && it >= nit && qn > 0)) break;   // wave-uniform: no round due
            // ---- a round of tier 2 over the first min(qn, 64) queued pairs: lane = pair
            const bool on = lane < qn;
            uint32_t ev = q1[lane];
            ev = on ? ev : tile_a;            // idle lanes test the tile's first window and push nothing
            if (qn > WAVE) {                    // what is left moves to the front of the queue
              const uint32_t rest = q1[WAVE + min(lane, qn - WAVE - 1)];
              if (lane < qn - WAVE) q1[lane] = (uint16_t)rest;
            }
            q_addr = q_base + 2u * (uint32_t)(qn > WAVE ? qn - WAVE : 0);
            const uint32_t o1 = ev & 1u, b0 = ev - o1, b1 = b0 + 4u * o1, k2 = o1 * 0x00020002u;
            const uint32_t e2 = b0 - tile_a + (uint32_t)(3 * PB + 4) + 2u * o1;   // tile index of the pair's first pixel
            const uint32_t sV = 0x0c010c00u + k2, sX = 0x0c030c02u - k2, sM = 0x0c020c01u + k2, sP = 0x0c040c03u - k2;
            unsigned long long p08, pc, pp, p62, p1014;
            asm volatile(
                "ds_read2_b32 %0, %5 offset0:1 offset1:%7\n\t"
                "ds_read2_b32 %1, %5 offset0:%8 offset1:%9\n\t"
                "ds_read2_b32 %2, %6 offset0:%9 offset1:%10\n\t"
                "ds_read2_b32 %3, %6 offset0:%11 offset1:%12\n\t"
                "ds_read2_b32 %4, %6 offset0:%13 offset1:%14\n\t"
                "s_waitcnt lgkmcnt(0)"
                : "=&v"(p08), "=&v"(pc), "=&v"(pp), "=&v"(p62), "=&v"(p1014)
                : "v"(b0), "v"(b1), "n"(6 * PB / 4 + 1), "n"(3 * PB / 4), "n"(3 * PB / 4 + 1), "n"(3 * PB / 4 + 2),
                  "n"(PB / 4 + 1), "n"(5 * PB / 4 + 1), "n"(PB / 4), "n"(5 * PB / 4)
                : "memory");
            {
              const uint32_t D0 = (uint32_t)(pc >> 32), Dm = (uint32_t)pc;
              const uint32_t V = __builtin_amdgcn_perm(0u, D0, sV);
              const uint32_t Q8 = __builtin_amdgcn_perm(0u, (uint32_t)p08, sV), Q0 = __builtin_amdgcn_perm(0u, (uint32_t)(p08 >> 32), sV);
              const uint32_t Q12 = __builtin_amdgcn_perm(D0, Dm, sM);
              const uint32_t Q4 = __builtin_amdgcn_perm((uint32_t)(pp >> 32), (uint32_t)pp, sP);
              const uint32_t Q6 = __builtin_amdgcn_perm(0u, (uint32_t)p62, sX), Q2 = __builtin_amdgcn_perm(0u, (uint32_t)(p62 >> 32), sX);
              const uint32_t Q10 = __builtin_amdgcn_perm(0u, (uint32_t)p1014, sX), Q14 = __builtin_amdgcn_perm(0u, (uint32_t)(p1014 >> 32), sX);
              const uint32_t AD = V + C, AB = C - V;
              const uint32_t lo = pkmax(pkmax(pkmin(Q0, Q8), pkmin(Q4, Q12)), pkmax(pkmin(Q2, Q10), pkmin(Q6, Q14)));
              const uint32_t hi = pkmin(pkmin(pkmax(Q0, Q8), pkmax(Q4, Q12)), pkmin(pkmax(Q2, Q10), pkmax(Q6, Q14)));
              const uint32_t dark = AD - lo, brt = hi + AB;
              const uint32_t G = on ? ((dark & 0x80008000u) | ((brt >> 1) & 0x40004000u)) : 0u;
              const bool has0 = (G & 0xC000u) != 0, has1 = (G >> 30) != 0;
              // the list stays in row-major order (the queue is first in, first out; a lane's two pixels are neighbours): the NMS
              // below emits straight from it
              const unsigned long long m0 = __ballot(has0), m1 = __ballot(has1);
              const int c0 = __builtin_amdgcn_mbcnt_hi((uint32_t)(m0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m0, (uint32_t)wcnt));
              const int i0 = __builtin_amdgcn_mbcnt_hi((uint32_t)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m1, (uint32_t)c0));
              const int i1 = i0 + (has0 ? 1 : 0);
              if (has0) list[i0] = (uint16_t)((G & 0xC000u) | e2);
              if (has1) list[i1] = (uint16_t)(((G >> 16) & 0xC000u) | (e2 + 1));
              wcnt += __popcll(m0) + __popcll(m1);
            }
            if (wcnt > FC_LIST_CAP - 2 * WAVE) {   // wave-uniform
              FC_T(2);
              if (!flushed)
                for (int i = lane; i < nbw; i += WAVE) scb[i] = 0;
              flushed = true;
              score_list(wcnt, true, th_cur);
              FC_T(3);
              wcnt = 0;
            }
          }
        }
#elif FC_T2_QUEUE
        // Tier 1 runs over all the cell's pixels; the lanes that pass it (a fifth of the 64 on a triggered iteration) used to take the
        // whole wave through tier 2 and the survivor push (35 vector instructions at ~20 % lane use).  Here they only queue the
        // position of their pair (ring `q1`, FIFO = row-major order); tier 2 + push run over 64 QUEUED pairs at a time: whenever the
        // ring holds a wave's worth, and once at the end of the cell (the tile is restaged for the next one).  Every pass of the loop
        // below is one tier-1 iteration followed by at most one such round; the extra last pass is the drain.
        int q1h = 0, q1t = 0;
        for (int it = 0; it <= n_it; it++) {
          if (it < n_it) {
            // inactive lanes read in-range garbage: rows < rows_max + RI
            const unsigned long long actm = __builtin_amdgcn_sicmp(rows_left, 0, 38 /* ICMP_SGT */);
            unsigned long long p08, pc, pp;
            asm volatile(
                "ds_read2_b32 %0, %3 offset0:1 offset1:%5\n\t"
                "ds_read2_b32 %1, %3 offset0:%6 offset1:%7\n\t"
                "ds_read2_b32 %2, %4 offset0:%7 offset1:%8\n\t"
                "s_waitcnt lgkmcnt(0)"
                : "=&v"(p08), "=&v"(pc), "=&v"(pp)
                : "v"(a0), "v"(a1), "n"(6 * PB / 4 + 1), "n"(3 * PB / 4), "n"(3 * PB / 4 + 1), "n"(3 * PB / 4 + 2)
                : "memory");
            const uint32_t D0 = (uint32_t)(pc >> 32), Dm = (uint32_t)pc;
            const uint32_t V = __builtin_amdgcn_perm(0u, D0, selV);
            const uint32_t Q8 = __builtin_amdgcn_perm(0u, (uint32_t)p08, selV), Q0 = __builtin_amdgcn_perm(0u, (uint32_t)(p08 >> 32), selV);
            const uint32_t Q12 = __builtin_amdgcn_perm(D0, Dm, selM);
            const uint32_t Q4 = __builtin_amdgcn_perm((uint32_t)(pp >> 32), (uint32_t)pp, selP);
            const uint32_t AD = V + C, AB = C - V;
            // one of every antipodal pair is darker than v - t  <=>  max over the pairs of the pair minimum is; same for brighter.
            // First tier: the pairs (0, 8) and (4, 12) alone.
            const uint32_t lo = pkmax(pkmin(Q0, Q8), pkmin(Q4, Q12));
            const uint32_t hi = pkmin(pkmax(Q0, Q8), pkmax(Q4, Q12));
            const uint32_t t1 = ((AD - lo) | (hi + AB)) & 0x80008000u;
            const unsigned long long m1 = __builtin_amdgcn_uicmp(t1, 0u, 33 /* ICMP_NE */) & actm;   // v_cmp + s_and: no lane mask round trip
            if (m1 != 0ull) {   // wave-uniform
              const int r = __builtin_amdgcn_mbcnt_hi((uint32_t)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m1, (uint32_t)q1t));
              if ((t1 != 0u) & (rows_left > 0)) q1[r & (FC_Q1_CAP - 1)] = (uint16_t)(a0 + ec);
              q1t += __popcll(m1);
            }
            a0 += RI * PB; a1 += RI * PB; rows_left -= RI;
          }
          const int qn = q1t - q1h;
          if (it < n_it ? qn < WAVE : qn == 0) continue;   // wave-uniform
          // ---- a round of tier 2 over the first min(qn, 64) queued pairs: lane = pair
          const bool on = lane < qn;
          uint32_t e2 = q1[(q1h + lane) & (FC_Q1_CAP - 1)];
          e2 = on ? e2 : (uint32_t)(3 * PB + 4);   // idle lanes test a valid position and push nothing
          q1h += qn < WAVE ? qn : WAVE;
          const uint32_t xx = PB == 64 ? (e2 & 63u) : e2 - (uint32_t)((int)((e2 + 0.5f) * (1.0f / PB))) * (uint32_t)PB;
          const uint32_t o2 = xx & 2u, k2 = o2 * 0x00010001u;
          const uint32_t sV = 0x0c010c00u + k2, sX = 0x0c030c02u - k2, sM = 0x0c020c01u + k2, sP = 0x0c040c03u - k2;
          const uint32_t b0 = tile_a + e2 - (uint32_t)(3 * PB + 4) - (xx & 3u);
          const uint32_t b1 = b0 + 2u * o2;
          unsigned long long p08, pc, pp, p62, p1014;
          asm volatile(
              "ds_read2_b32 %0, %5 offset0:1 offset1:%7\n\t"
              "ds_read2_b32 %1, %5 offset0:%8 offset1:%9\n\t"
              "ds_read2_b32 %2, %6 offset0:%9 offset1:%10\n\t"
              "ds_read2_b32 %3, %6 offset0:%11 offset1:%12\n\t"
              "ds_read2_b32 %4, %6 offset0:%13 offset1:%14\n\t"
              "s_waitcnt lgkmcnt(0)"
              : "=&v"(p08), "=&v"(pc), "=&v"(pp), "=&v"(p62), "=&v"(p1014)
              : "v"(b0), "v"(b1), "n"(6 * PB / 4 + 1), "n"(3 * PB / 4), "n"(3 * PB / 4 + 1), "n"(3 * PB / 4 + 2),
                "n"(PB / 4 + 1), "n"(5 * PB / 4 + 1), "n"(PB / 4), "n"(5 * PB / 4)
              : "memory");
          {
            const uint32_t D0 = (uint32_t)(pc >> 32), Dm = (uint32_t)pc;
            const uint32_t V = __builtin_amdgcn_perm(0u, D0, sV);
            const uint32_t Q8 = __builtin_amdgcn_perm(0u, (uint32_t)p08, sV), Q0 = __builtin_amdgcn_perm(0u, (uint32_t)(p08 >> 32), sV);
            const uint32_t Q12 = __builtin_amdgcn_perm(D0, Dm, sM);
            const uint32_t Q4 = __builtin_amdgcn_perm((uint32_t)(pp >> 32), (uint32_t)pp, sP);
            const uint32_t Q6 = __builtin_amdgcn_perm(0u, (uint32_t)p62, sX), Q2 = __builtin_amdgcn_perm(0u, (uint32_t)(p62 >> 32), sX);
            const uint32_t Q10 = __builtin_amdgcn_perm(0u, (uint32_t)p1014, sX), Q14 = __builtin_amdgcn_perm(0u, (uint32_t)(p1014 >> 32), sX);
            const uint32_t AD = V + C, AB = C - V;
            const uint32_t lo = pkmax(pkmax(pkmin(Q0, Q8), pkmin(Q4, Q12)), pkmax(pkmin(Q2, Q10), pkmin(Q6, Q14)));
            const uint32_t hi = pkmin(pkmin(pkmax(Q0, Q8), pkmax(Q4, Q12)), pkmin(pkmax(Q2, Q10), pkmax(Q6, Q14)));
            const uint32_t dark = AD - lo, brt = hi + AB;
            const uint32_t G = on ? ((dark & 0x80008000u) | ((brt >> 1) & 0x40004000u)) : 0u;
            const bool has0 = (G & 0xC000u) != 0, has1 = (G >> 30) != 0;
            // the list stays in row-major order (the ring is first in, first out; a lane's two pixels are neighbours): the NMS below
            // emits straight from it
            const unsigned long long m0 = __ballot(has0), m1 = __ballot(has1);
            const int c0 = __builtin_amdgcn_mbcnt_hi((uint32_t)(m0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m0, (uint32_t)wcnt));
            const int i0 = __builtin_amdgcn_mbcnt_hi((uint32_t)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m1, (uint32_t)c0));
            const int i1 = i0 + (has0 ? 1 : 0);
            if (has0) list[i0] = (uint16_t)((G & 0xC000u) | e2);
            if (has1) list[i1] = (uint16_t)(((G >> 16) & 0xC000u) | (e2 + 1));
            wcnt += __popcll(m0) + __popcll(m1);
          }
          if (wcnt > FC_LIST_CAP - 2 * WAVE) {   // wave-uniform
            FC_T(2);
            if (!flushed)
              for (int i = lane; i < nbw; i += WAVE) scb[i] = 0;
            flushed = true;
            score_list(wcnt, true, th_cur);
            FC_T(3);
            wcnt = 0;
          }
        }
#else
        for (int it = 0; it < n_it; it++, a0 += RI * PB, a1 += RI * PB, rows_left -= RI) {
          // inactive lanes read in-range garbage: rows < rows_max + RI
          const unsigned long long actm = __builtin_amdgcn_sicmp(rows_left, 0, 38 /* ICMP_SGT */);
          const bool act = rows_left > 0;
          unsigned long long p08, pc, pp, p62, p1014;
          // all ten dwords requested at once; the first tier needs the first three pairs only (LDS returns in order)
          asm volatile(
              "ds_read2_b32 %0, %5 offset0:1 offset1:%7\n\t"
              "ds_read2_b32 %1, %5 offset0:%8 offset1:%9\n\t"
              "ds_read2_b32 %2, %6 offset0:%9 offset1:%10\n\t"
              "ds_read2_b32 %3, %6 offset0:%11 offset1:%12\n\t"
              "ds_read2_b32 %4, %6 offset0:%13 offset1:%14\n\t"
              "s_waitcnt lgkmcnt(2)"
              : "=&v"(p08), "=&v"(pc), "=&v"(pp), "=&v"(p62), "=&v"(p1014)
              : "v"(a0), "v"(a1), "n"(6 * PB / 4 + 1), "n"(3 * PB / 4), "n"(3 * PB / 4 + 1), "n"(3 * PB / 4 + 2),
                "n"(PB / 4 + 1), "n"(5 * PB / 4 + 1), "n"(PB / 4), "n"(5 * PB / 4)
              : "memory");
          const uint32_t D0 = (uint32_t)(pc >> 32), Dm = (uint32_t)pc;
          const uint32_t V = __builtin_amdgcn_perm(0u, D0, selV);
          const uint32_t Q8 = __builtin_amdgcn_perm(0u, (uint32_t)p08, selV), Q0 = __builtin_amdgcn_perm(0u, (uint32_t)(p08 >> 32), selV);
          const uint32_t Q12 = __builtin_amdgcn_perm(D0, Dm, selM);
          const uint32_t Q4 = __builtin_amdgcn_perm((uint32_t)(pp >> 32), (uint32_t)pp, selP);
          const uint32_t AD = V + C, AB = C - V;
          // one of every antipodal pair is darker than v - t  <=>  max over the pairs of the pair minimum is; same for brighter.
          // First tier: the pairs (0, 8) and (4, 12) alone.  No lane of the wave passes it in 43 % of the iterations at
          // t = 20 on the synthetic KITTI frames (two thirds on the DBoW2 demo images): those skip the other half.
          uint32_t lo = pkmax(pkmin(Q0, Q8), pkmin(Q4, Q12));
          uint32_t hi = pkmin(pkmax(Q0, Q8), pkmax(Q4, Q12));
          const uint32_t t1 = ((AD - lo) | (hi + AB)) & 0x80008000u;
          const bool any1 = (__builtin_amdgcn_uicmp(t1, 0u, 33 /* ICMP_NE */) & actm) != 0ull;   // v_cmp + s_and: no lane mask round trip
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(p62), "+v"(p1014) : : "memory");
          if (!any1) continue;
          const uint32_t Q6 = __builtin_amdgcn_perm(0u, (uint32_t)p62, selX), Q2 = __builtin_amdgcn_perm(0u, (uint32_t)(p62 >> 32), selX);
          const uint32_t Q10 = __builtin_amdgcn_perm(0u, (uint32_t)p1014, selX), Q14 = __builtin_amdgcn_perm(0u, (uint32_t)(p1014 >> 32), selX);
          lo = pkmax(lo, pkmax(pkmin(Q2, Q10), pkmin(Q6, Q14)));
          hi = pkmin(hi, pkmin(pkmax(Q2, Q10), pkmax(Q6, Q14)));
          const uint32_t dark = AD - lo, brt = hi + AB;
          const uint32_t G = act ? ((dark & 0x80008000u) | ((brt >> 1) & 0x40004000u)) : 0u;
          const bool has0 = (G & 0xC000u) != 0, has1 = (G >> 30) != 0;
          // the list stays in row-major order (lanes ascend along a row, then down the rows; a lane's two pixels are neighbours):
          // the NMS below emits straight from it
          const unsigned long long m0 = __ballot(has0), m1 = __ballot(has1);
          const int c0 = __builtin_amdgcn_mbcnt_hi((uint32_t)(m0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m0, (uint32_t)wcnt));
          const int i0 = __builtin_amdgcn_mbcnt_hi((uint32_t)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m1, (uint32_t)c0));
          const int i1 = i0 + (has0 ? 1 : 0);
          const uint32_t e = a0 + ec;
          if (has0) list[i0] = (uint16_t)((G & 0xC000u) | e);
          if (has1) list[i1] = (uint16_t)(((G >> 16) & 0xC000u) | (e + 1));
          wcnt += __popcll(m0) + __popcll(m1);
          if (wcnt > FC_LIST_CAP - 2 * WAVE) {   // wave-uniform
            FC_T(2);
            if (!flushed)
              for (int i = lane; i < nbw; i += WAVE) scb[i] = 0;
            flushed = true;
            score_list(wcnt, true, th_cur);
            FC_T(3);
            wcnt = 0;
          }
        }
#endif
        FC_T(2);
        score_list(wcnt, flushed, th_cur);
        FC_T(3);
      }

      // ---- B: strict 3x3 NMS inside the cell.
      if (!flushed) {
        // lane = survivor: the list still holds every scored pixel of the cell, in row-major order -- the order the reference's
        // keypoints of a cell come in -- so a kept pixel goes straight to its output slot: rank = kept pixels in front of it.
        // Nine reads in flight together.  (Pass 0 of a cell that keeps nothing writes nothing.)
        uint32_t* slot = slots + (size_t)img * slots_per_image + cd.slot_off;
        int emitted = 0;
        for (int base = 0; base < wcnt; base += WAVE) {
          const int i = min(base + lane, wcnt - 1);
          const uint32_t en = list[i];
          const int e = (int)(en & 0x3fffu);
          const int y = PB == 64 ? (e >> 6) : (int)((e + 0.5f) * (1.0f / PB));
          const int x = e - y * PB;
          const int ty = y - 3, tx = x - c_lo;
          const uint8_t* s = sc + (ty + 1) * SCP + min(tx, tw - 1) + 1;
          const int v = s[0];
          const int n0 = s[-SCP - 1], n1 = s[-SCP], n2 = s[-SCP + 1], n3 = s[-1], n4 = s[1], n5 = s[SCP - 1], n6 = s[SCP], n7 = s[SCP + 1];
          const bool kp = (base + lane < wcnt) & (x < c_hi) & (v > n0) & (v > n1) & (v > n2) & (v > n3) & (v > n4) & (v > n5) & (v > n6) & (v > n7);   // v = 0: not scored
          const unsigned long long mk = __ballot(kp);
          const int off = __builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, (uint32_t)emitted));
          const uint32_t rx = (uint32_t)(tx + 3 + cd.x0 - ORBFE_EDGE), ry = (uint32_t)(ty + 3 + cd.y0 - ORBFE_EDGE);
          if (kp && off < cd.slot_cap) slot[off] = rx | (ry << 12) | ((uint32_t)v << 24);
          emitted += __popcll(mk);
        }
        total = emitted;
      } else {
        // the list was recycled (more than 256 quick-test survivors in one cell): lane l walks the bits of words l and l + 64
        // of the bitmap of scored pixels
#pragma unroll
        for (int h = 0; h < 2; h++) {
          const int w = lane + WAVE * h;
          uint32_t m = w < nbw ? scb[w] : 0u;
          keep[h] = 0;
          while (m) {
            const int b = __ffs((int)m) - 1;
            m &= m - 1;
            const int p = w * 32 + b;
            const int ty = p >> bsh, tx = p & ((1 << bsh) - 1);
            const uint8_t* s = sc + (ty + 1) * SCP + tx + 1;
            // all nine reads in flight together: with `&&` every comparison waited for its own LDS round trip
            const int v = s[0];
            const int n0 = s[-SCP - 1], n1 = s[-SCP], n2 = s[-SCP + 1], n3 = s[-1], n4 = s[1], n5 = s[SCP - 1], n6 = s[SCP], n7 = s[SCP + 1];
            const bool kp = (v > n0) & (v > n1) & (v > n2) & (v > n3) & (v > n4) & (v > n5) & (v > n6) & (v > n7);
            if (kp) keep[h] |= 1u << b;
          }
        }
        total = __popcll(__ballot(keep[0] != 0u || keep[1] != 0u));   // > 0 <=> the cell keeps a pixel
      }
      was_flushed = flushed;
      FC_T(4);   // NMS (and, for a cell whose list was never recycled, emission)
      if (total != 0 || pass == 1) break;
      // nothing at ini_th: start over at min_th (the score plane may hold pass 0's plateau pixels; a recycled list's bitmap is
      // cleared again when pass 1 recycles)
      {
        const int nsc = ((th + 2) * SCP + 15) >> 4;
        for (int i = lane; i < nsc; i += WAVE) reinterpret_cast<uint4*>(sc)[i] = make_uint4(0, 0, 0, 0);
      }
    }

    // ---- C: (recycled lists only) row-major emission from registers (one wave scan per bitmap half gives the output offsets)
    if (!was_flushed) {
      if (lane == 0) cell_cnt[(size_t)img * total_cells + g.first_cell + k] = total < cd.slot_cap ? total : cd.slot_cap;
      FC_T(5);
    } else {
      const int pk = __popc(keep[0]) | (__popc(keep[1]) << 16);   // both halves in one packed wave scan
      const int in = wave_incl_scan(pk);
      const int tot = __builtin_amdgcn_readlane(in, WAVE - 1);
      const int tot0 = tot & 0xffff, n_out = tot0 + (tot >> 16);
      const int ex = in - pk;
      uint32_t* slot = slots + (size_t)img * slots_per_image + cd.slot_off;
#pragma unroll
      for (int h = 0; h < 2; h++) {
        int off = h ? tot0 + (ex >> 16) : (ex & 0xffff);
        uint32_t mask = keep[h];
        const int w = lane + WAVE * h;
        while (mask) {
          const int b = __ffs((int)mask) - 1;
          mask &= mask - 1;
          const int p = w * 32 + b;
          const int ty = p >> bsh, tx = p & ((1 << bsh) - 1);
          const uint32_t sv = sc[(ty + 1) * SCP + tx + 1];
          const uint32_t rx = (uint32_t)(tx + 3 + cd.x0 - ORBFE_EDGE), ry = (uint32_t)(ty + 3 + cd.y0 - ORBFE_EDGE);
          if (off < cd.slot_cap) slot[off] = rx | (ry << 12) | (sv << 24);
          off++;
        }
      }
      if (lane == 0) cell_cnt[(size_t)img * total_cells + g.first_cell + k] = n_out < cd.slot_cap ? n_out : cd.slot_cap;
      FC_T(5);   // scans + emission
    }
  }
#if FC_TIMING
  if (lane == 0) {
    unsigned long long* pr = g_fc_prof + (size_t)((rid * 2654435761u) >> 20) * 8;   // 4096 slots
    atomicAdd(&pr[0], (unsigned long long)tacc0); atomicAdd(&pr[1], (unsigned long long)tacc1);
    atomicAdd(&pr[2], (unsigned long long)tacc2); atomicAdd(&pr[3], (unsigned long long)tacc3);
    atomicAdd(&pr[4], (unsigned long long)tacc4); atomicAdd(&pr[5], (unsigned long long)tacc5);
    atomicAdd(&pr[6], 1ull);
    atomicAdd(&pr[7], (unsigned long long)g.n_cells);
  }
#endif
}

// ------------------------------------------------------------------------------------------------ octree
// DistributeOctTree as an array algorithm.  The leaves of the quadtree are kept in an array in std::list
// order (index 0 = list head); because every insertion in the reference is a push_front and the initial
// nodes are push_back'ed, list order == descending creation order, so "sort by (size, node address)" with
// a bump allocator == sort by (size, -position).  One iteration = count keys per (leaf, quadrant), prefix
// sums over leaves, rebuild the array, relabel keys.  Keys never move.
//
// key record (8 bytes): x:int16 | y:int16 | node:uint16 | score:uint8 | quadrant:uint8
__device__ __forceinline__ unsigned long long key_pack(int x, int y, int node, int score, int q) {
  return (unsigned long long)(uint16_t)x | ((unsigned long long)(uint16_t)y << 16) |
         ((unsigned long long)(uint16_t)node << 32) | ((unsigned long long)(uint8_t)score << 48) |
         ((unsigned long long)(uint8_t)q << 56);
}
#define KEY_X(k) ((int)(int16_t)((k) & 0xffff))
#define KEY_Y(k) ((int)(int16_t)(((k) >> 16) & 0xffff))
#define KEY_NODE(k) ((int)(((k) >> 32) & 0xffff))
#define KEY_SCORE(k) ((int)(((k) >> 48) & 0xff))
#define KEY_Q(k) ((int)(((k) >> 56) & 0xff))

struct OctNode {
  int16_t x0, x1, y0, y1;
  int32_t cnt;
};

size_t orbfe_octree_lds_bytes(int M, int lds_keys) {
  // nodeA, nodeB (12 B), cnt4 (16 B), childpos (8 B: 4 x uint16), aux (4 B), aux2 (4 B); the phase-2 sort buffers
  // alias childpos (sort keys) and nodeB (sorted keys), which are dead while the sort runs
  return (size_t)M * (12 + 12 + 16 + 8 + 4 + 4) + (size_t)lds_keys * 8 + 64 * 4 + ORBFE_MAX_INI * 8 + 256;
}

#if FC_TIMING
__device__ unsigned long long g_oct_prof[256];   // [level] cycles, [32 + level] iterations of the subdivision loop, [64 + 8 level + phase] cycles
extern "C" int orbfe_debug_oct_profile(unsigned long long* out, int reset) {   // out: 256 entries
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_oct_prof), sizeof(unsigned long long) * 256) != hipSuccess) return 1;
  if (reset) { unsigned long long z[256]; memset(z, 0, sizeof(z)); if (hipMemcpyToSymbol(HIP_SYMBOL(g_oct_prof), z, sizeof(z)) != hipSuccess) return 1; }
  return 0;
}
// phases (tools/oct_phase_profile.py): 0 cell offsets, 1 key gather, 2 first relabel / histogram, 3 node work of the subdivision
// rounds, 4 their key sweeps, 5 best key + output
#define OCT_TP(i) do { const unsigned long long _t = __builtin_readcyclecounter(); tph[i] += _t - t_prev; t_prev = _t; } while (0)
#else
#define OCT_TP(i)
#endif
#define OCT_T ORBFE_OCT_THREADS   // 256 measured best (64: 0.120, 128: 0.091, 256: 0.069, 512: 0.091 ms per 256 images)
static_assert(ORBFE_MAX_INI <= ORBFE_OCT_THREADS, "one thread per root node in step 3");
__global__ __launch_bounds__(ORBFE_OCT_THREADS) void octree_select_kernel(OctParams P) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int tid = threadIdx.x;
  // grid = (images, levels): workgroups start level-major, the long ones (level 0: ~120 k cycles, level 7: ~45 k) first, so the
  // tail of the launch is made of short workgroups (image-major order ran as two rounds of the longest one: 110 -> 80 us)
  const int level = blockIdx.y, img = blockIdx.x;
#if FC_TIMING
  const unsigned long long t_begin = __builtin_readcyclecounter();
  unsigned long long t_prev = t_begin, tph[6] = {0, 0, 0, 0, 0, 0};
  int n_iter = 0;
#endif
  const OctLevel L = P.lv[level];
  const int M = P.max_nodes;

  // carve dynamic LDS
  uint8_t* sp = smem;
  unsigned long long* lkeys = reinterpret_cast<unsigned long long*>(sp);   sp += (size_t)P.lds_keys * 8;
  uint16_t* childpos = reinterpret_cast<uint16_t*>(sp); sp += (size_t)M * 8;
  OctNode* nodeA = reinterpret_cast<OctNode*>(sp); sp += (size_t)M * 12;   // M is a multiple of 64: 8-byte aligned
  OctNode* nodeB = reinterpret_cast<OctNode*>(sp); sp += (size_t)M * 12;
  int* cnt4 = reinterpret_cast<int*>(sp);          sp += (size_t)M * 16;
  int* aux = reinterpret_cast<int*>(sp);           sp += (size_t)M * 4;   // childbase / flags
  int* aux2 = reinterpret_cast<int*>(sp);          sp += (size_t)M * 4;   // staypos
  int* ini_cnt = reinterpret_cast<int*>(sp);       sp += ORBFE_MAX_INI * 4;
  int* ini_map = reinterpret_cast<int*>(sp);       sp += ORBFE_MAX_INI * 4;
  int* scan_tmp = reinterpret_cast<int*>(sp);      sp += 16 * 4;
  int* sh = reinterpret_cast<int*>(sp);            // small shared scalars

  const int32_t* cnt = P.cell_cnt + (size_t)img * P.total_cells + L.cell_begin;
  int32_t* coff = P.cell_off + (size_t)img * P.total_cells + L.cell_begin;
  const CellDesc* cells = P.cells + L.cell_begin;
  const uint32_t* slots = P.slots + (size_t)img * P.slots_per_image;
  uint32_t* out_kp = P.lvl_kp + (size_t)img * P.kp_per_image + L.kp_off;

  // 1. exclusive offsets of the cells' candidate lists (cell raster order == vToDistributeKeys order).  Offset, count and slot
  //    address of every cell go to LDS (the quadrant-count array, idle until step 3) when the level's cells fit: the gather below
  //    then has the candidates themselves as its only memory requests
  const int tab_cap = (M * 4) / 3;
  const bool use_tab = L.n_cells <= tab_cap;
  int* tb_o = cnt4;
  int* tb_n = cnt4 + tab_cap;
  uint32_t* tb_so = reinterpret_cast<uint32_t*>(cnt4 + 2 * tab_cap);
  int running = 0;
  for (int base = 0; base < L.n_cells; base += OCT_T) {
    const int c = base + tid;
    const int v = c < L.n_cells ? cnt[c] : 0;
    const uint32_t so = c < L.n_cells ? cells[c].slot_off : 0u;
    int tot;
    const int incl = block_incl_scan_t<OCT_T>(v, scan_tmp, &tot);
    if (c < L.n_cells) {
      if (use_tab) { tb_o[c] = running + incl - v; tb_n[c] = v; tb_so[c] = so; }
      else coff[c] = running + incl - v;
    }
    running += tot;
  }
  const int C = running;
  OCT_TP(0);
  if (C > L.key_cap || C > 0xFFFFFF) {  // cannot happen: key_cap = sum of slot caps
    if (tid == 0) { atomicOr(P.err, 1); P.lvl_n[(size_t)img * ORBFE_MAX_LEVELS + level] = 0; }
    return;
  }
  unsigned long long* keys = (C <= P.lds_keys) ? lkeys : (P.gkeys + (size_t)img * P.gkeys_per_image + L.key_off);

  for (int i = tid; i < L.n_ini; i += OCT_T) ini_cnt[i] = 0;
  __syncthreads();

  // 2. gather keys, assign to the root nodes: vpIniNodes[kp.pt.x / hX] (L/src/ORBextractor.cc:559)
  auto put_key = [&](uint32_t e, int at) {
    const int x = e & 0xfff, y = (e >> 12) & 0xfff, s = e >> 24;
    int node = (int)((float)x / L.hX);
    if (node >= L.n_ini) node = L.n_ini - 1;
    keys[at] = key_pack(x, y, node, s, 0);
    atomicAdd(&ini_cnt[node], 1);
  };
  if (use_tab) {
    // G lanes per cell, G = the power of two at or above the mean list length (1 .. 64): few candidates per cell -> many cells per
    // pass; dense texture -> a wave per cell, coalesced reads and writes.  Two cells x four candidates per lane are requested
    // together (thread per cell with one request at a time was a chain of round trips: a quarter of the level-0 workgroup's time,
    // a third with dense texture)
    int gs = 0;
    while (gs < 6 && ((L.n_cells << gs) < C)) gs++;
    const int G = 1 << gs, NG = OCT_T >> gs;
    const int g = tid >> gs, jl = tid & (G - 1);
    for (int c0 = g; c0 < L.n_cells; c0 += 2 * NG) {
      const int c1 = c0 + NG;
      const bool h1 = c1 < L.n_cells;
      const int nA = tb_n[c0], oA = tb_o[c0], nB = h1 ? tb_n[c1] : 0, oB = h1 ? tb_o[c1] : 0;
      const uint32_t* sA = slots + tb_so[c0];
      const uint32_t* sB = slots + (h1 ? tb_so[c1] : 0u);
      for (int j0 = jl; j0 < max(nA, nB); j0 += 4 * G) {
        uint32_t eA[4], eB[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          eA[u] = j0 + u * G < nA ? sA[j0 + u * G] : 0u;
          eB[u] = j0 + u * G < nB ? sB[j0 + u * G] : 0u;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
          if (j0 + u * G < nA) put_key(eA[u], oA + j0 + u * G);
          if (j0 + u * G < nB) put_key(eB[u], oB + j0 + u * G);
        }
      }
    }
  } else {
    for (int c = tid; c < L.n_cells; c += OCT_T) {   // more cells than the table holds (very large images): thread per cell
      const int n = cnt[c];
      const int o = coff[c];
      const uint32_t* src = slots + cells[c].slot_off;
      for (int j = 0; j < n; j++) put_key(src[j], o + j);
    }
  }
  __syncthreads();
  OCT_TP(1);

  // 3. root nodes that hold keys, in order (empty ones are erased, :564-572)
  int size;
  {
    const int has = (tid < L.n_ini && ini_cnt[tid] > 0) ? 1 : 0;
    int tot;
    const int incl = block_incl_scan_t<OCT_T>(has, scan_tmp, &tot);
    if (tid < L.n_ini) {
      ini_map[tid] = incl - has;
      if (has) {
        OctNode nd;
        nd.x0 = (int16_t)(int)(L.hX * (float)tid);
        nd.x1 = (int16_t)(int)(L.hX * (float)(tid + 1));
        nd.y0 = 0;
        nd.y1 = (int16_t)L.height;
        nd.cnt = ini_cnt[tid];
        nodeA[incl - 1] = nd;
      }
    }
    size = tot;
  }
  for (int i = tid; i < size * 4; i += OCT_T) cnt4[i] = 0;
  __syncthreads();
  // A key's new leaf and, in the same visit, the quadrant of that leaf it falls into with the leaf's quadrant count (DivideNode
  // :505-517) for the NEXT subdivision round: one read and one write of every key per round instead of two passes (a count
  // pass and a relabel pass).  Four keys per thread are requested together (the spill path reads them from HBM: one round trip
  // per key otherwise).
  auto relabel_and_count = [&](const OctNode* nodes, auto&& new_leaf) {
    for (int k0 = tid; k0 < C; k0 += 4 * OCT_T) {
      unsigned long long kk[4];
#pragma unroll
      for (int j = 0; j < 4; j++) kk[j] = keys[min(k0 + j * OCT_T, C - 1)];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int k = k0 + j * OCT_T;
        if (k >= C) break;
        const int np = new_leaf(kk[j]);
        const OctNode nd = nodes[np];
        int q = 0;
        if (nd.cnt > 1) {
          const int mx = nd.x0 + ((nd.x1 - nd.x0 + 1) >> 1);  // UL.x + ceil(w/2)
          const int my = nd.y0 + ((nd.y1 - nd.y0 + 1) >> 1);
          q = (KEY_X(kk[j]) >= mx ? 1 : 0) + (KEY_Y(kk[j]) >= my ? 2 : 0);  // 0:n1 1:n2 2:n3 3:n4
          atomicAdd(&cnt4[np * 4 + q], 1);
        }
        keys[k] = (kk[j] & 0x00ff0000ffffffffULL) | ((unsigned long long)(uint16_t)np << 32) | ((unsigned long long)q << 56);
      }
    }
  };
  // ---- fast-forward for levels whose keys live in HBM (dense texture).  Where a node is cut depends on the node alone, so a key's
  // path through the first D = L.ff_depth subdivisions is a function of its position: ONE sweep counts the keys per depth-D cell,
  // sums of four give the counts of every shallower node, and the subdivision rounds below take a new leaf's quadrant counts from
  // those tables instead of sweeping the keys (a sweep = 16 bytes per key through the fabric for all workgroups at once: the four
  // rounds of uniform noise were 630 k of the level-0 workgroup's 1.4 M cycles).  Keys keep their ROOT index meanwhile; the first
  // round that creates a multi-key leaf at depth D writes real labels (`materialise`) and the explicit rounds take over.
  // Tables T_1 .. T_D (T_d: n_ini * 4^d counts, index = 4 * parent index + quadrant), the cell -> leaf map and the leaves'
  // (depth, index) words live in the LDS key array, which is idle exactly when the keys are in HBM.
  const int FD = (keys != lkeys) ? L.ff_depth : 0;
  bool ff = FD > 0;
  int* ffT = reinterpret_cast<int*>(lkeys);                         // T_d starts at n_ini * (4^d - 4) / 3
  const int ff_nT = L.n_ini * (((1 << (2 * FD + 2)) - 4) / 3);
  uint16_t* f2l = reinterpret_cast<uint16_t*>(ffT + ff_nT);         // depth-D cell -> leaf
  const int ff_nF = L.n_ini << (2 * FD);
  uint16_t* infoA = f2l + ((ff_nF + 1) & ~1);                       // leaf -> depth << 12 | index at that depth
  uint16_t* infoB = infoA + M;
  auto ff_tab = [&](int d) { return ffT + L.n_ini * (((1 << (2 * d)) - 4) / 3); };
  auto ff_cell = [&](unsigned long long kk) {   // index of the key's depth-D cell (KEY_NODE = root index)
    const int r = KEY_NODE(kk), x = KEY_X(kk), y = KEY_Y(kk);
    int x0 = (int16_t)(int)(L.hX * (float)r), x1 = (int16_t)(int)(L.hX * (float)(r + 1)), y0 = 0, y1 = (int16_t)L.height;
    int idx = r;
    for (int d = 0; d < FD; d++) {
      const int mx = x0 + ((x1 - x0 + 1) >> 1), my = y0 + ((y1 - y0 + 1) >> 1);
      const int qx = x >= mx ? 1 : 0, qy = y >= my ? 1 : 0;
      if (qx) x0 = mx; else x1 = mx;
      if (qy) y0 = my; else y1 = my;
      idx = idx * 4 + qx + 2 * qy;
    }
    return idx;
  };
  auto ff_fill_f2l = [&](const uint16_t* info, int n) {   // every leaf marks the depth-D cells it covers
    for (int p = tid; p < n; p += OCT_T) {
      const int d = info[p] >> 12, sh = 2 * (FD - d);
      const int first = (int)(info[p] & 0xfffu) << sh;
      for (int i = 0; i < (1 << sh); i++) f2l[first + i] = (uint16_t)p;
    }
  };
  auto ff_counts = [&](const OctNode* nodes, const uint16_t* info, int n) {   // quadrant counts of the multi-key leaves from the tables
    for (int i = tid; i < n * 4; i += OCT_T) {
      const int p = i >> 2;
      if (nodes[p].cnt > 1) cnt4[i] = ff_tab((info[p] >> 12) + 1)[4 * (int)(info[p] & 0xfffu) + (i & 3)];
    }
  };
  if (ff) {
    for (int i = tid; i < ff_nT; i += OCT_T) ffT[i] = 0;
    __syncthreads();
    int* TD = ff_tab(FD);
    for (int k0 = tid; k0 < C; k0 += 8 * OCT_T) {
      unsigned long long kk[8];
#pragma unroll
      for (int j = 0; j < 8; j++) kk[j] = keys[min(k0 + j * OCT_T, C - 1)];
#pragma unroll
      for (int j = 0; j < 8; j++)
        if (k0 + j * OCT_T < C) atomicAdd(&TD[ff_cell(kk[j])], 1);
    }
    __syncthreads();
    for (int d = FD - 1; d >= 1; d--) {
      int* Td = ff_tab(d);
      const int* Tc = ff_tab(d + 1);
      for (int i = tid; i < (L.n_ini << (2 * d)); i += OCT_T) Td[i] = Tc[4 * i] + Tc[4 * i + 1] + Tc[4 * i + 2] + Tc[4 * i + 3];
      __syncthreads();
    }
    if (tid < L.n_ini && ini_cnt[tid] > 0) infoA[ini_map[tid]] = (uint16_t)tid;   // depth 0
    __syncthreads();
    ff_counts(nodeA, infoA, size);
  } else {
    relabel_and_count(nodeA, [&](unsigned long long kk) { return ini_map[KEY_NODE(kk)]; });
  }
  __syncthreads();
  OCT_TP(2);

  // 4. subdivision loop (:581-709)
  int phase = 1;
  int scan_slot = 0;
  bool finish = (size == 0);
  const int N = L.N;
  while (!finish) {
#if FC_TIMING
    n_iter++;
#endif
    const int prev = size;   // cnt4 holds the quadrant counts of every multi-key leaf of nodeA (relabel_and_count)

    // which leaves split, and in which order their children are created.  (The rounds are a chain of barriers -- ~10 k of a
    // workgroup's ~20 k cycles per round whatever the level holds -- so phase 1 takes one packed scan for the children's and the
    // staying leaves' positions, counts the next round's expandable leaves while it builds them, and every thread reads only
    // entries it wrote itself until the barrier behind the build.)
    int K = 0;        // number of children created
    int S = 0;        // leaves that stay
    int nsplit = 0;
    if (tid == 0) { sh[2] = 0; sh[3] = 0; }   // multi-key children created this round (nToExpand) / a leaf below the tables; set behind a barrier
    if (phase == 1) {
      // every multi-key leaf splits, in list order (:592-643)
      int run_c = 0, run_s = 0;
      for (int base = 0; base < size; base += OCT_T) {
        const int p = base + tid;
        int c = 0;
        if (p < size && nodeA[p].cnt > 1)
          c = (cnt4[p * 4] > 0) + (cnt4[p * 4 + 1] > 0) + (cnt4[p * 4 + 2] > 0) + (cnt4[p * 4 + 3] > 0);
        const int st = (p < size && c == 0) ? 1 : 0;   // a multi-key leaf has a non-empty quadrant
        int tot;
        const int incl = block_incl_scan_alt<OCT_T>(c | (st << 16), scan_tmp + 8, scan_slot, &tot);
        if (p < size) {
          aux[p] = c > 0 ? run_c + (incl & 0xffff) - c : -1;  // childbase, -1 = stays
          if (st) aux2[p] = run_s + (incl >> 16) - 1;          // rank among the leaves that stay
        }
        run_c += tot & 0xffff;
        run_s += tot >> 16;
      }
      K = run_c;
      S = run_s;
      for (int p = tid; p < size; p += OCT_T)
        if (aux[p] < 0) aux2[p] += K;   // after the K new children, old order preserved
    } else {
      // phase 2 (:651-707): expandable leaves sorted by (size, address) ascending, walked from the back:
      // larger first, among equal sizes the later-created (= smaller list position) first; stop as soon as
      // the list holds >= N nodes.
      unsigned long long* sortkey = reinterpret_cast<unsigned long long*>(childpos);  // dead until the rebuild below
      unsigned long long* sorted = reinterpret_cast<unsigned long long*>(nodeB);      // previous generation: dead
      int nE = 0;
      for (int base = 0; base < size; base += OCT_T) {
        const int p = base + tid;
        const int e = (p < size && nodeA[p].cnt > 1) ? 1 : 0;
        int tot;
        const int incl = block_incl_scan_t<OCT_T>(e, scan_tmp, &tot);
        if (e) sortkey[nE + incl - 1] = ((unsigned long long)(0xFFFFFFFFu - (uint32_t)nodeA[p].cnt) << 32) | (uint32_t)p;
        if (p < size) aux[p] = -1;
        nE += tot;
      }
      __syncthreads();
      // rank sort (keys are unique)
      for (int i = tid; i < nE; i += OCT_T) {
        const unsigned long long mine = sortkey[i];
        int r = 0;
        for (int j = 0; j < nE; j++) r += (sortkey[j] < mine) ? 1 : 0;
        sorted[r] = mine;
      }
      __syncthreads();
      // walk in sorted order: running list size after each split; first rank reaching N ends the walk
      if (tid == 0) sh[0] = nE;  // index of the last split rank + 1
      int run_inc = 0, run_c = 0;
      for (int base = 0; base < nE; base += OCT_T) {
        const int r = base + tid;
        int c = 0, p = 0;
        if (r < nE) {
          p = (int)(sorted[r] & 0xffffffffu);
          c = (cnt4[p * 4] > 0) + (cnt4[p * 4 + 1] > 0) + (cnt4[p * 4 + 2] > 0) + (cnt4[p * 4 + 3] > 0);
        }
        int tot;
        const int packed = ((c > 0 ? c - 1 : 0) << 16) | c;
        const int incl = block_incl_scan_t<OCT_T>(packed, scan_tmp, &tot);
        if (r < nE) {
          const int size_after = prev + run_inc + (incl >> 16);
          aux2[r] = run_c + (incl & 0xffff) - c;  // childbase by rank (temporarily in aux2)
          if (size_after >= N) atomicMin(&sh[0], r + 1);
        }
        run_inc += tot >> 16;
        run_c += tot & 0xffff;
        __syncthreads();
      }
      __syncthreads();
      nsplit = sh[0];
      __syncthreads();
      if (tid == 0) sh[1] = 0;
      __syncthreads();
      for (int r = tid; r < nsplit; r += OCT_T) {
        const int p = (int)(sorted[r] & 0xffffffffu);
        aux[p] = aux2[r];
        if (r == nsplit - 1) {
          const int c = (cnt4[p * 4] > 0) + (cnt4[p * 4 + 1] > 0) + (cnt4[p * 4 + 2] > 0) + (cnt4[p * 4 + 3] > 0);
          sh[1] = aux2[r] + c;
        }
      }
      __syncthreads();
      K = sh[1];
      __syncthreads();
      // positions of the leaves that stay: after the K new children, old order preserved
      for (int base = 0; base < size; base += OCT_T) {
        const int p = base + tid;
        const int st = (p < size && aux[p] < 0) ? 1 : 0;
        int tot;
        const int incl = block_incl_scan_t<OCT_T>(st, scan_tmp, &tot);
        if (st) aux2[p] = K + S + incl - 1;
        S += tot;
      }
    }
    const int newsize = K + S;
    if (newsize > M) {  // cannot happen: M >= max(N + 3, 4 * nIni)
      if (tid == 0) { atomicOr(P.err, 2); P.lvl_n[(size_t)img * ORBFE_MAX_LEVELS + level] = 0; }
      return;
    }
    // build the new leaf array: children pushed to the front in creation order => reversed
    int n_multi = 0;
    for (int p = tid; p < size; p += OCT_T) {
      const OctNode nd = nodeA[p];
      if (aux[p] >= 0) {
        const int mx = nd.x0 + ((nd.x1 - nd.x0 + 1) >> 1);
        const int my = nd.y0 + ((nd.y1 - nd.y0 + 1) >> 1);
        int j = aux[p];
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const int cq = cnt4[p * 4 + q];
          if (cq > 0) {
            OctNode ch;
            ch.x0 = (q & 1) ? (int16_t)mx : nd.x0;
            ch.x1 = (q & 1) ? nd.x1 : (int16_t)mx;
            ch.y0 = (q & 2) ? (int16_t)my : nd.y0;
            ch.y1 = (q & 2) ? nd.y1 : (int16_t)my;
            ch.cnt = cq;
            const int pos = K - 1 - j;
            nodeB[pos] = ch;
            childpos[p * 4 + q] = (uint16_t)pos;
            if (ff) infoB[pos] = (uint16_t)((((infoA[p] >> 12) + 1) << 12) | ((infoA[p] & 0xfffu) * 4 + q));
            n_multi += cq > 1 ? 1 : 0;
            j++;
          }
        }
      } else {
        nodeB[aux2[p]] = nd;
        if (ff) infoB[aux2[p]] = infoA[p];
      }
    }
    {   // one add per wave: 256 adds to one LDS word are 256 passes through an LDS pipeline that four workgroups share
      const int wsum = wave_incl_scan(n_multi);
      if ((tid & (WAVE - 1)) == WAVE - 1 && wsum) atomicAdd(&sh[2], wsum);
    }
    __syncthreads();
    const int n_expand = sh[2];   // read here: thread 0 clears it again at the top of the next round, two barriers on
    for (int i = tid; i < newsize * 4; i += OCT_T) cnt4[i] = 0;   // the build above was the last reader of this round's counts
    if (ff) {   // a multi-key leaf at depth D: its quadrant counts are not in the tables
      const bool last = newsize >= N || newsize == prev;   // the loop ends below: nobody reads the next round's counts
      // (no __syncthreads_or: it brings 256 bytes of static LDS, and the fourth workgroup no longer fits the compute unit)
      for (int p = tid; p < newsize; p += OCT_T)
        if (nodeB[p].cnt > 1 && (infoB[p] >> 12) >= FD) sh[3] = 1;
      __syncthreads();
      const int deep = sh[3];
      OCT_TP(3);
      if (last) {
      } else if (!deep) {
        ff_counts(nodeB, infoB, newsize);
      } else {
        ff_fill_f2l(infoB, newsize);
        __syncthreads();
        relabel_and_count(nodeB, [&](unsigned long long kk) { return (int)f2l[ff_cell(kk)]; });   // labels from here on
        ff = false;
      }
      { uint16_t* t = infoA; infoA = infoB; infoB = t; }
    } else {
      __syncthreads();
      OCT_TP(3);
      relabel_and_count(nodeB, [&](unsigned long long kk) {
        const int p = KEY_NODE(kk);
        return aux[p] >= 0 ? (int)childpos[p * 4 + KEY_Q(kk)] : aux2[p];
      });
    }
    { OctNode* t = nodeA; nodeA = nodeB; nodeB = t; }
    size = newsize;
    __syncthreads();
    OCT_TP(4);

    // nToExpand = leaves with more than one key: in phase 1 all of them are this round's children (:645-649)
    if (size >= N || size == prev) finish = true;
    else if (phase == 1 && size + 3 * n_expand > N) phase = 2;
  }

  OCT_TP(3);
  // 5. best key of every leaf: max response, first candidate wins ties (:712-728)
  uint32_t* best = reinterpret_cast<uint32_t*>(cnt4);
  for (int p = tid; p < size; p += OCT_T) best[p] = 0;
  __syncthreads();
  if (ff) {   // the tree was finished inside the tables: the keys never got labels, a key's leaf is its cell's
    ff_fill_f2l(infoA, size);
    __syncthreads();
    for (int k0 = tid; k0 < C; k0 += 8 * OCT_T) {
      unsigned long long kk[8];
#pragma unroll
      for (int j = 0; j < 8; j++) kk[j] = keys[min(k0 + j * OCT_T, C - 1)];
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const int k = k0 + j * OCT_T;
        if (k < C) atomicMax(&best[f2l[ff_cell(kk[j])]], ((uint32_t)KEY_SCORE(kk[j]) << 24) | (0xFFFFFFu - (uint32_t)k));
      }
    }
  } else {
    for (int k = tid; k < C; k += OCT_T) {
      const unsigned long long kk = keys[k];
      atomicMax(&best[KEY_NODE(kk)], ((uint32_t)KEY_SCORE(kk) << 24) | (0xFFFFFFu - (uint32_t)k));
    }
  }
  __syncthreads();
  for (int p = tid; p < size; p += OCT_T) {
    const int k = (int)(0xFFFFFFu - (best[p] & 0xFFFFFFu));
    const unsigned long long kk = keys[k];
    if (p < L.kp_cap)
      out_kp[p] = (uint32_t)KEY_X(kk) | ((uint32_t)KEY_Y(kk) << 12) | ((uint32_t)KEY_SCORE(kk) << 24);
  }
  if (tid == 0) {
    if (size > L.kp_cap) atomicOr(P.err, 4);
    P.lvl_n[(size_t)img * ORBFE_MAX_LEVELS + level] = size < L.kp_cap ? size : L.kp_cap;
#if FC_TIMING
    atomicAdd(&g_oct_prof[level], __builtin_readcyclecounter() - t_begin);
    atomicAdd(&g_oct_prof[32 + level], (unsigned long long)n_iter);
    OCT_TP(5);
    for (int i = 0; i < 6; i++) atomicAdd(&g_oct_prof[64 + 8 * level + i], tph[i]);
#endif
  }
}

// ------------------------------------------------------------------------------------------------ blur
// The blurred planes are read by nobody but the descriptor gather, which fetches a 37-row x 64-byte window per keypoint: one
// memory request per window row in a row-major plane (~55 per keypoint).  They are therefore stored TILED: 16 pixels x 8 rows
// = one 128-byte line, tiles in raster order (pitch / 16 tiles per tile row).  A window then touches ~4 x 6 tiles.
// Byte offset of pixel (x, y); x, y >= 0:
__device__ __forceinline__ uint32_t blur_tiled_offset(int x, int y, int pitch) { return orbfe_tiled_offset(x, y, pitch); }   // orbfe_internal.h
// 7x7 sigma-2 Gaussian, OpenCV's 8.8 fixed-point taps [18,34,48,56,48,34,18], exact 16.16 accumulation,
// round half up.  One 64 x BT_H (56) output tile per workgroup, separable through LDS.  Interior tiles stage the
// (64+8) x (56+6) input window with aligned dword loads; tiles touching a level edge take the byte path with
// REFLECT_101 indexing.  Both passes produce 4 adjacent pixels per work item (dword LDS/global accesses).
// ---- the tile kernel's blur on the matrix cores (BT_MFMA, the default): the arithmetic of gauss_blur7_mfma_kernel below -- a 7-tap
// pass is Out = In x Band, pixels as i8 (x ^ 0x80), exact in v_mfma_i32_16x16x64_i8 -- fed from the tile's staged window instead of
// a strip walk.  Wave w of the four owns output columns 16w .. 16w + 15 of the 64 x 56 tile and all of its rows:
//   horizontal: per 16-row block of the window ONE product: A = lane (q, r): window row 16 blk + r, bytes 16w + 16q .. + 15 (one
//     aligned ds_read_b128: the operand layout IS the staged rows), B = the band matrix (output column 4 + 16w + n of the window
//     <- input columns 16w + k: the same matrix for every column group), C = 128.  D: lane (q, c) holds H - 32768 + 128 of rows
//     4q .. 4q + 3 at column c, packed (blur_pack) into the i8 digits (a, b) of H - 32768 = 256 a + b;
//   vertical: those dwords of the four blocks ARE the second product's A operand of the same lane (K order: byte 4 ww + i <->
//     window row 16 ww + 4q + i; the band matrices T_b are written in that order), B = T_b for output rows 16b .. 16b + 15:
//     V = 256 (T a) + (T b) + 256 * 32768, pixel = byte 2 of V + 32768; lane (q, n) holds columns 4q .. 4q + 3 of output row
//     16b + n: one dword of the 16 x 8-tiled plane.
// REFLECT_101 needs no matrix of its own here: the staging has put the reflected pixels into the window.  No second LDS buffer, no
// second barrier, 154 instead of 243 vector instructions per wave; rows 62, 63 of the last block meet zero coefficients only.
typedef int v4i __attribute__((ext_vector_type(4)));
__device__ __attribute__((aligned(16))) int8_t g_blur_tile_tab[5 * 1024];   // [0]: the horizontal band matrix; [1..4]: T_b (1 KB each: 64 lanes x 16 bytes)
__device__ __forceinline__ void blur_pack(const v4i d, int& hi, int& lo) {
  const uint32_t t01 = __builtin_amdgcn_perm((uint32_t)d.y, (uint32_t)d.x, 0x05010400u);   // [d0.b0, d1.b0, d0.b1, d1.b1]
  const uint32_t t23 = __builtin_amdgcn_perm((uint32_t)d.w, (uint32_t)d.z, 0x05010400u);
  lo = (int)(__builtin_amdgcn_perm(t23, t01, 0x05040100u) ^ 0x80808080u);
  hi = (int)__builtin_amdgcn_perm(t23, t01, 0x07060302u);
}
#ifndef BT_MFMA
#define BT_MFMA 1        // 0: the two passes on the vector ALUs (v_dot4 / v_dot2 through a second LDS buffer: rounds 1-4)
#endif
#define BT_W 64
#define BT_H ORBFE_BLUR_TILE_H   // 56: (56 + 6) / 2 = 31 row pairs x 16 groups: two 256-thread passes; 7 blocks of 8 output rows = 7 storage tiles
                                // (58 filled both passes exactly; 56 measured 2-3 % faster: its rows end on storage-tile boundaries)
#if BT_MFMA
#define BT_INP 96   // LDS pitch (bytes) of the input window: six 16-byte pieces, column j <-> level x = ox - 16 + j (a tiled level arrives
#define BT_COL0 12  // as whole tile rows); BT_COL0: the column of x = ox - 4, where the window the two passes need begins
#else
#define BT_INP 80   // LDS pitch (bytes) of the input window: column j <-> level x = ox - 4 + j
#define BT_COL0 0
#endif
#define BT_HP 68    // LDS pitch (dwords) of one row PAIR of the horizontal-pass result (two u16 rows interleaved)
// RESIZE: the tile also produces its part of level + 1 (cv::resize INTER_LINEAR, the arithmetic of pyr_resize_dot_kernel) from
// the window it has staged for the blur: the pyramid chain and the blur then read every level ONCE, and the seven resize
// launches of a pyramid disappear (launch l = blur of level l + resize l -> l + 1; the last level's launch only blurs).  A
// destination dword (four pixels) belongs to the tile that holds the first source column of its first pixel, a destination row
// to the tile that holds its upper source row (BlurTile::j0 .. r1, from the plan): every tap of an owned dword-row then lies
// inside the window -- staged one dword wider (19 instead of 18) than the blur alone needs --, nothing is computed twice and no
// halo is added.  Thread = (dword column, every (BT_THREADS / 16)-th destination row); its horizontal taps are requested before the
// window's pixels, so they have arrived when the barrier behind the staging opens; the rows' vertical taps go through LDS.
#ifndef BT_NT_STORE
#define BT_NT_STORE 1   // the blurred planes are next read by the descriptor gather, 0.5 GB of other traffic later: stored non-temporal
#endif                  // they leave more of the raw levels in the Infinity Cache for FAST (+0.5 % on the step)
#ifndef BT_NT_LOAD
#define BT_NT_LOAD 0     // non-temporal window loads: 0.46 -> 0.53 ms per level chain (neighbouring tiles share lines through L2)
#endif
#ifndef BT_XCD_IMAGES
#define BT_XCD_IMAGES 1  // whole images per XCD instead of runs of eight tiles: 88.1 k -> 90.2 k frames/s (level chain 0.466 -> 0.441 ms)
#endif
#ifndef BT_SCALAR_TILE
#define BT_SCALAR_TILE 1
#endif
#ifndef BT_MIN_WAVES
#define BT_MIN_WAVES 2   // with one argument hipcc puts the MFMA results into AGPRs and copies every one back (lesson 31)
#endif
#ifndef BT_THREADS
#define BT_THREADS 128   // threads per tile: with the matrix-core blur a tile needs 5 KB of LDS and 56 registers, so sixteen two-wave tiles
#endif                   // are resident per CU (256 threads: eight; 64: the fused resize needs 69 registers): 0.506 / 0.474 / 0.492 ms per level chain
template <bool RESIZE, int THREADS>   // THREADS per tile: THREADS for batches (tiles in flight), 256 for a handful of images (latency of a tile)
__global__ __launch_bounds__(THREADS, BT_MIN_WAVES) void blur_level_kernel(PyrView src, PyrView dst, const BlurTile* __restrict__ tiles, int n_tiles,
                                                         LevelResize rz) {
#if BT_MFMA
  __shared__ __attribute__((aligned(16))) uint8_t in[65 * BT_INP];   // 62 staged rows; the last 16-row block reads two more, the last column
                                                                     // group 32 bytes past its row (zero coefficients)
#else
  __shared__ __attribute__((aligned(16))) uint8_t in[(BT_H + 6) * BT_INP];
  __shared__ __attribute__((aligned(16))) uint32_t hbp[((BT_H + 6) / 2) * BT_HP];
#endif
  const int tid = threadIdx.x;
#if BT_MFMA
  // this wave's band matrix and the four vertical ones: 5 x 16 bytes per lane, requested before anything else
  const int mlane = tid & 63, mwave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const v4i* mtab = reinterpret_cast<const v4i*>(g_blur_tile_tab);
  const v4i HB = mtab[mlane];
#endif
#if BT_XCD_IMAGES
  // whole images per XCD (workgroups go to the XCDs round-robin in linear order): XCD k walks images k, k + 8, ... tile by tile in
  // raster order, so the halo rows and the 128-byte lines neighbouring tiles share are fetched into ONE L2 once
  int tile_id = (int)blockIdx.x, img = (int)blockIdx.y;
  {
    const unsigned gx = gridDim.x;
    const unsigned lin = blockIdx.y * gx + blockIdx.x;
    const unsigned grp = lin / (8u * gx);
    if (8u * grp + 8u <= gridDim.y) {
      const unsigned within = lin - grp * 8u * gx;
      img = (int)(8u * grp + (within & 7u));
      tile_id = (int)(within >> 3);
    }
  }
#else
  // XCD-aware mapping: runs of 8 raster-consecutive tiles per XCD, interleaved over the 8 XCDs
  const int q = blockIdx.x >> 3, img = (int)blockIdx.y;
  const int tile_id = (int)blockIdx.x >= (n_tiles / 64) * 64 ? (int)blockIdx.x : (q >> 3) * 64 + (blockIdx.x & 7) * 8 + (q & 7);
#endif
#if BT_SCALAR_TILE
  // the tile's record through the scalar cache (four dwords at a wave-uniform address): the compiler's own choice is a vector
  // dwordx3 + ushort load -- an L2 round trip in front of the window's first load address, on the critical path of every tile
  static_assert(sizeof(BlurTile) == 16, "four dwords");
  BlurTile t;
  {
    const __attribute__((address_space(4))) uint32_t* q =
        (const __attribute__((address_space(4))) uint32_t*)(uintptr_t)(tiles + __builtin_amdgcn_readfirstlane(tile_id));
    uint32_t w4[4];
#pragma unroll
    for (int j = 0; j < 4; j++) w4[j] = q[j];
    __builtin_memcpy(&t, w4, sizeof(t));
  }
#else
  const BlurTile t = tiles[tile_id];
#endif
  const int lvl = t.level;
  const int w = src.w[lvl], h = src.h[lvl], pitch = src.pitch[lvl];
  const uint8_t* S = src.base[lvl] + (size_t)img * src.img_stride[lvl];
  uint8_t* D = const_cast<uint8_t*>(dst.base[lvl]) + (size_t)img * dst.img_stride[lvl];
  const int ox = t.tx * BT_W, oy = t.ty * BT_H;
  const bool stiled = (src.tiled >> lvl) & 1u;   // the level this launch reads lies in 16 x 8 tiles (written by the launch before it)
  // the resize taps: this thread's dword column J (pixels 4J .. 4J + 3) in registers; the vertical taps of the tile's (at most 48)
  // destination rows go through LDS -- a thread's rows are THREADS / 16 apart, held in registers they cost a wave of occupancy
  const int J = t.j0 + (tid & (ORBFE_FUSE_DWORDS - 1));
  __shared__ uint4 ytap[16 * ORBFE_FUSE_ROWS];   // per destination row: byte offsets of its two source rows in the window, the two weights << 12
  __shared__ uint32_t ydst[16 * ORBFE_FUSE_ROWS];   // ... and the byte offset of the row (of its tile row and row inside it) in level + 1
  uint2 txr[4], tyl = make_uint2(0u, 0u);
  if constexpr (RESIZE) {
#pragma unroll
    for (int i = 0; i < 4; i++) txr[i] = reinterpret_cast<const uint2*>(rz.xt)[min(4 * J + i, rz.dw - 1)];
    tyl = reinterpret_cast<const uint2*>(rz.yt)[min(t.r0 + min(tid, 16 * ORBFE_FUSE_ROWS - 1), rz.dh - 1)];
  }
  {
    // (BT_H+6) rows x NC dwords (x = ox-4 .. ox+67, or .. ox+71 with the resize).  Thread -> dword column c = tid % NC and rows
    // r0 + RPP k (one division; 252 / 247 of 256 threads), all loads issued before the first LDS store.  Interior tiles (the
    // common case, wave-uniform) load plain aligned dwords; on edge tiles a dword that straddles the image border (left edge, the
    // partial dword at the right edge, columns beyond it) is assembled from bytes with REFLECT_101 indexing, and rows reflect as a
    // whole.
    uint32_t* in32 = reinterpret_cast<uint32_t*>(in);
    constexpr int NC = RESIZE ? 19 : 18;
    constexpr int RPP = THREADS / NC, NLD = (BT_H + 6 + RPP - 1) / RPP;
    const int r0 = tid / NC, c = tid - r0 * NC;
    const int x = ox - 4 + 4 * c;
    uint32_t v[NLD];
    const bool interior = oy >= 3 && oy + BT_H + 3 <= h && ox >= 4 && ox - 4 + 4 * NC <= w;
    const bool tilepath = stiled && BT_COL0 != 0;   // a tiled level arrives as whole tile rows (below), also at the level's edges
    // (round 6: interior windows staged by LDS-DMA -- piece i = 6 * row + column IS byte 16 i of the window for row-major and, with tile
    //  addresses, for tiled sources -- bit-exact, level chain 0.416-0.421 against 0.422-0.425 on packed input, +-0 in the bench: removed,
    //  profiles/r06_level_chain.md)
    if (r0 < RPP && !tilepath) {
      if (interior && stiled) {
        // (vector-ALU build) an aligned dword of a tiled level lies inside one tile row: the same loads, tile addresses
        const uint32_t xoff = ((uint32_t)(x >> 4) << 7) + (uint32_t)(x & 15), tstep = (uint32_t)(pitch >> 4) << 7;
#pragma unroll
        for (int k = 0; k < NLD; k++) {
          const int y = oy - 3 + r0 + RPP * k;
          v[k] = r0 + RPP * k < BT_H + 6 ? *reinterpret_cast<const uint32_t*>(S + (uint32_t)(y >> 3) * tstep + (uint32_t)((y & 7) * 16) + xoff) : 0u;
        }
      } else if (interior) {
        const uint8_t* p0 = S + (size_t)(oy - 3 + r0) * pitch + x;
#pragma unroll
        for (int k = 0; k < NLD; k++)
#if BT_NT_LOAD
          v[k] = r0 + RPP * k < BT_H + 6 ? __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(p0 + (size_t)(RPP * k) * pitch)) : 0u;
#else
          v[k] = r0 + RPP * k < BT_H + 6 ? *reinterpret_cast<const uint32_t*>(p0 + (size_t)(RPP * k) * pitch) : 0u;
#endif
      } else {
        const bool whole = x >= 0 && x + 3 < w;
        int gx[4];
#pragma unroll
        for (int j = 0; j < 4; j++) gx[j] = reflect101(x + j, w);
#pragma unroll
        for (int k = 0; k < NLD; k++) {
          const int r = r0 + RPP * k;
          v[k] = 0u;
          if (r < BT_H + 6) {
            const int yy = reflect101(oy + r - 3, h);
            auto px = [&](int xx) { return (uint32_t)S[orbfe_level_offset(xx, yy, pitch, stiled)]; };
            if (whole) v[k] = *reinterpret_cast<const uint32_t*>(S + orbfe_level_offset(x, yy, pitch, stiled));
            else v[k] = px(gx[0]) | (px(gx[1]) << 8) | (px(gx[2]) << 16) | (px(gx[3]) << 24);
          }
        }
      }
#pragma unroll
      for (int k = 0; k < NLD; k++) {
        const int r = r0 + RPP * k;
        if (r < BT_H + 6) in32[r * (BT_INP / 4) + BT_COL0 / 4 + c] = v[k];
      }
    }
#if BT_MFMA
    if (tilepath) {
      // A tiled level arrives as whole tile rows: the window's 6 x 9 tiles, eight lanes per tile (one 16-byte tile row each), so a
      // load instruction covers eight complete 128-byte lines -- 54 lines per window where the row-major form touches 124.
      // Rows of the first and last tile row outside the window are dropped at the store; tiles outside the plane are not read.
      // Only the window's 62 rows are requested: rows 5 .. 7 of the first tile row, the seven tile rows in between whole, rows 0 .. 2
      // of the last one (oy is a multiple of 56 = 7 x 8, so the window always starts at row 5 of a tile row) -- 372 pieces of 16 bytes.
      constexpr int N_TOP = 3 * 6, N_MID = 7 * 8 * 6, N_ALL = N_TOP + N_MID + 3 * 6;
      constexpr int NP = (N_ALL + THREADS - 1) / THREADS;
      const int ty0 = (oy - 3) >> 3, tx0 = (ox - 16) >> 4;   // (arithmetic shifts: -1 in the first tile row / column)
      const int tiles_y = (h + 7) >> 3, tiles_x = pitch >> 4;
      const uint32_t tstep = (uint32_t)tiles_x << 7;
      // piece -> (tile row, tile column, row inside the tile): eight (three) consecutive lanes share a tile
      // (integer forms of / 3 and / 6 for the small arguments here: (x * 43) >> 7 and >> 8; the mapping is computed once and kept --
      //  recomputing it for the stores was a tenth of the kernel's vector instructions)
      auto piece = [&](int pid, int& tyi, int& txi, int& rr) {
        if (pid < N_TOP) { txi = (pid * 43) >> 7; rr = 5 + pid - 3 * txi; tyi = 0; }
        else if (pid < N_TOP + N_MID) { const int q = pid - N_TOP, tl = q >> 3; rr = q & 7; tyi = (tl * 43) >> 8; txi = tl - 6 * tyi; tyi += 1; }
        else { const int q = pid - N_TOP - N_MID; txi = (q * 43) >> 7; rr = q - 3 * txi; tyi = 8; }
      };
      uint4 pv[NP];
      uint32_t lds_at[NP];
#pragma unroll
      for (int k = 0; k < NP; k++) {
        int tyi, txi, rr;
        piece(min(tid + k * THREADS, N_ALL - 1), tyi, txi, rr);
        const int tr = min(max(ty0 + tyi, 0), tiles_y - 1), tc = min(max(tx0 + txi, 0), tiles_x - 1);   // clamped: a valid address in any case
        pv[k] = *reinterpret_cast<const uint4*>(S + (uint32_t)tr * tstep + ((uint32_t)tc << 7) + (uint32_t)(rr * 16));
        lds_at[k] = (uint32_t)((tyi * 8 + rr - 5) * BT_INP + 16 * txi);   // window row (ty0 + tyi) * 8 + rr - (oy - 3)
      }
#pragma unroll
      for (int k = 0; k < NP; k++)
        if (tid + k * THREADS < N_ALL) *reinterpret_cast<uint4*>(in + lds_at[k]) = pv[k];
      if (!interior) {
        // REFLECT_101 at the level's edges, inside LDS: what the passes read outside the level is within three pixels of it, and
        // the pixels those reflect to lie inside the window.  Columns first (rows inside the level), then whole rows.
        __syncthreads();
        {
          const int nl = ox == 0 ? 3 : 0, nr = w < ox - 4 + 4 * NC ? 3 : 0;   // x = -3 .. -1; x = w .. w + 2
          for (int i = tid; i < (BT_H + 6) * (nl + nr); i += THREADS) {
            const int wr = (int)((i + 0.5f) * (1.0f / (float)(nl + nr))), j = i - wr * (nl + nr);
            const int xx = j < nl ? j - 3 : w + (j - nl);
            const int y = oy - 3 + wr;
            if (y >= 0 && y < h && xx - (ox - 16) < BT_INP)
              in[wr * BT_INP + (xx - (ox - 16))] = in[wr * BT_INP + (reflect101(xx, w) - (ox - 16))];
          }
        }
        __syncthreads();
        {
          const int nt = oy == 0 ? 3 : 0, nb = h < oy + BT_H + 3 ? 3 : 0;     // y = -3 .. -1; y = h .. h + 2
          for (int i = tid; i < (nt + nb) * (BT_INP / 4); i += THREADS) {
            const int j = (int)((i + 0.5f) * (1.0f / (float)(BT_INP / 4))), cdw = i - j * (BT_INP / 4);
            const int y = j < nt ? j - 3 : h + (j - nt);
            const int wr = y - (oy - 3), ws = reflect101(y, h) - (oy - 3);
            if (wr >= 0 && wr < BT_H + 6) in32[wr * (BT_INP / 4) + cdw] = in32[ws * (BT_INP / 4) + cdw];
          }
        }
      }
    }
#endif
  }
  if constexpr (RESIZE)
    if (tid < 16 * ORBFE_FUSE_ROWS) {
      const int ra = (int)(int16_t)(tyl.x & 0xffff) - (oy - 3), rb = (int)(int16_t)(tyl.x >> 16) - (oy - 3);
      ytap[tid] = make_uint4((uint32_t)(ra * BT_INP), (uint32_t)(rb * BT_INP), (tyl.y & 0xffffu) << 12, (tyl.y >> 16) << 12);
      const int yd = t.r0 + tid;
      ydst[tid] = rz.dst_tiled ? (uint32_t)(yd >> 3) * ((uint32_t)(rz.dpitch >> 4) << 7) + (uint32_t)((yd & 7) * 16) : (uint32_t)yd * (uint32_t)rz.dpitch;
    }
  __syncthreads();
  if constexpr (RESIZE) {
    // level + 1: the destination dword J of rows Y0 .. Y0 + 2.  Per source row three aligned LDS dwords, two v_alignbyte to start
    // the 8-byte window at the first pixel's source column, per pixel one v_perm (bytes s0, s0 + 1 into 16-bit fields) + one
    // v_dot2_u32_u16 against (c0, c1); (b * (t >> 4)) >> 16 == mul_hi_u24(b << 12, t & ~15).  Rows differ between the lanes of a
    // wave here, so the horizontal result of a source row is not carried over to the next destination row (a divergent test).
    typedef __attribute__((ext_vector_type(2))) unsigned short us2;
    if (J < t.j1) {
      const int s00 = (int)(int16_t)(txr[0].x & 0xffff);
      const int base = s00 - (ox - 4) + BT_COL0;   // >= 4 + BT_COL0: the dword's first source column lies in this tile column
      const uint32_t sh = (uint32_t)base & 3u;
      uint32_t sel[4], cp[4];
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const uint32_t o = (uint32_t)((int)(int16_t)(txr[i].x & 0xffff) - s00);   // 0 .. 6
        sel[i] = o | 0x0c000c00u | ((o + 1) << 16);
        cp[i] = txr[i].y;                                                         // c0 | c1 << 16
      }
      const uint32_t trow = (uint32_t)(uintptr_t)(in + (base & ~3));   // LDS byte address of the window column the taps start in
      // the three dwords of BOTH source rows in one block of LDS reads (ds_read2_b32 + ds_read_b32 off one address register per row:
      // a 4-byte aligned ds_read_b96 is slow, and every way of keeping the compiler from forming one -- an asm barrier on the pointer, a
      // volatile access -- turned the third dword into a flat_load_dword with a wait of its own)
      auto hrow2 = [&](uint32_t off_a, uint32_t off_b, uint32_t (&ha)[4], uint32_t (&hb)[4]) {
        unsigned long long a01, b01;
        uint32_t a2, b2;
        asm volatile("ds_read2_b32 %0, %4 offset1:1\n\t"
                     "ds_read_b32 %1, %4 offset:8\n\t"
                     "ds_read2_b32 %2, %5 offset1:1\n\t"
                     "ds_read_b32 %3, %5 offset:8\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&v"(a01), "=&v"(a2), "=&v"(b01), "=&v"(b2)
                     : "v"(trow + off_a), "v"(trow + off_b)
                     : "memory");
        const uint32_t A0 = __builtin_amdgcn_alignbyte((uint32_t)(a01 >> 32), (uint32_t)a01, sh), A1 = __builtin_amdgcn_alignbyte(a2, (uint32_t)(a01 >> 32), sh);
        const uint32_t B0w = __builtin_amdgcn_alignbyte((uint32_t)(b01 >> 32), (uint32_t)b01, sh), B1w = __builtin_amdgcn_alignbyte(b2, (uint32_t)(b01 >> 32), sh);
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const uint32_t ua = __builtin_amdgcn_perm(A1, A0, sel[i]), ub = __builtin_amdgcn_perm(B1w, B0w, sel[i]);
          ha[i] = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, ua), __builtin_bit_cast(us2, cp[i]), 0u, false) & ~15u;
          hb[i] = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, ub), __builtin_bit_cast(us2, cp[i]), 0u, false) & ~15u;
        }
      };
      // rows t.r0 + (tid >> 4), + THREADS / 16, ...: the row groups of a wave take consecutive rows
      // level + 1 is written in 16 x 8 tiles (orbfe_internal.h; row-major where a stand-alone kernel reads it next): the dword's
      // place inside its row / tile row is the thread's, the row's is the loop's
      uint8_t* Nimg = rz.dst + (size_t)img * rz.dimg;   // wave-uniform base; the offsets below are 32-bit (a plane is < 2^25 bytes)
      const uint32_t ncol = rz.dst_tiled ? (((uint32_t)((4 * J) >> 4) << 7) + (uint32_t)((4 * J) & 15)) : (uint32_t)(4 * J);
      for (int yi = tid >> 4; yi < t.r1 - t.r0; yi += THREADS / 16) {
        uint8_t* N = Nimg + (ncol + ydst[yi]);
        {
          const uint4 ty = ytap[yi];
          uint32_t h0[4], h1[4];
          hrow2(ty.x, ty.y, h0, h1);
          const uint32_t B0 = ty.z, B1 = ty.w;
          uint32_t sm[4];
#pragma unroll
          for (int i = 0; i < 4; i++) sm[i] = mulhi_u24(B0, h0[i]) + mulhi_u24(B1, h1[i]) + 2u;
          const uint32_t P01 = (sm[0] | (sm[1] << 16)) >> 2, P23 = (sm[2] | (sm[3] << 16)) >> 2;   // values <= 255 in bytes 0 and 2
          *reinterpret_cast<uint32_t*>(N) = __builtin_amdgcn_perm(P23, P01, 0x06040200u);
        }
      }
    }
  }
#if BT_MFMA
  {
    const int q = mlane >> 4, r = mlane & 15;
    const v4i c128 = {128, 128, 128, 128}, zero = {0, 0, 0, 0};
    const v4i cfin = {8388608 + 32768, 8388608 + 32768, 8388608 + 32768, 8388608 + 32768};
    v4i TB[4];   // requested here, used behind the first product: not live through the resize part above
#pragma unroll
    for (int b = 0; b < 4; b++) TB[b] = mtab[(1 + b) * 64 + mlane];
    const uint32_t bstep = (uint32_t)(dst.pitch[lvl] >> 4) << 8;      // 16 rows further: 2 tile rows x (pitch / 16) tiles x 128 bytes
#pragma unroll
    for (int g = mwave; g < BT_W / 16; g += THREADS / 64) {   // column group g: output columns 16g .. 16g + 15 of the tile
      v4i Xh, Xl;
      {
        const v4i* arow = reinterpret_cast<const v4i*>(in + r * BT_INP + 16 * g + 16 * q);   // window columns 16g + 16q ..: every tap of the group inside
        v4i A[4];
#pragma unroll
        for (int blk = 0; blk < 4; blk++) A[blk] = arow[blk * BT_INP];   // window rows 16 blk + r
        int ph[4], pl[4];
#pragma unroll
        for (int blk = 0; blk < 4; blk++) {
          const v4i d = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[blk] ^ (int)0x80808080, HB, c128, 0, 0, 0);
          blur_pack(d, ph[blk], pl[blk]);
        }
        Xh = v4i{ph[0], ph[1], ph[2], ph[3]};
        Xl = v4i{pl[0], pl[1], pl[2], pl[3]};
      }
      const int gx = ox + 16 * g + 4 * q;   // a multiple of 4: the four pixels lie in one storage-tile row
      uint8_t* o = D + blur_tiled_offset(gx, oy + r, dst.pitch[lvl]);   // output row oy + 16 b + r: two storage-tile rows further per block
#pragma unroll
      for (int b = 0; b < 4; b++) {
        const v4i vh = __builtin_amdgcn_mfma_i32_16x16x64_i8(Xh, TB[b], zero, 0, 0, 0);
        const v4i vl = __builtin_amdgcn_mfma_i32_16x16x64_i8(Xl, TB[b], cfin, 0, 0, 0);
        const uint32_t r0 = ((uint32_t)vh.x << 8) + (uint32_t)vl.x, r1 = ((uint32_t)vh.y << 8) + (uint32_t)vl.y;
        const uint32_t r2 = ((uint32_t)vh.z << 8) + (uint32_t)vl.z, r3 = ((uint32_t)vh.w << 8) + (uint32_t)vl.w;
        const uint32_t out = __builtin_amdgcn_perm(r1, r0, 0x0c0c0602u) | __builtin_amdgcn_perm(r3, r2, 0x06020c0cu);
        const int row = 16 * b + r;
        if (gx < w && row < BT_H && oy + row < h) {
#if BT_NT_STORE
          __builtin_nontemporal_store(out, reinterpret_cast<uint32_t*>(o + b * bstep));
#else
          *reinterpret_cast<uint32_t*>(o + b * bstep) = out;
#endif
        }
      }
    }
  }
}
#else
  // horizontal pass: item = (row pair k, 4-pixel group g).  Output x = 4g+i needs window columns 4g+i+1 .. 4g+i+7:
  // two byte windows cut with v_alignbyte and two v_dot4_u32_u8 against the packed taps.  The two rows of a pair
  // are stored interleaved (even row in the low half) so the vertical pass can use v_dot2_u32_u16.
  for (int it = tid; it < ((BT_H + 6) / 2) * 16; it += THREADS) {
    const int k = it >> 4, g = it & 15;
    uint32_t o[2][4];
#pragma unroll
    for (int rr = 0; rr < 2; rr++) {
      const uint32_t* p = reinterpret_cast<const uint32_t*>(in + (2 * k + rr) * BT_INP + 4 * g);
      const uint32_t w0 = p[0], w1 = p[1], w2 = p[2];
      // output i = sum_j t_j * byte(i + 1 + j) of the 12-byte window (w0, w1, w2): instead of shifting the window with
      // v_alignbyte (a 4-clock instruction like v_dot4 itself) the TAPS are shifted -- compile-time constants: 10 v_dot4
      // and no shifts per four pixels where it was 8 + 6
#define TAPS(a, b, c, d) ((uint32_t)(a) | ((uint32_t)(b) << 8) | ((uint32_t)(c) << 16) | ((uint32_t)(d) << 24))
#define DOT4(x, t, acc) __builtin_amdgcn_udot4((x), (t), (acc), false)
      o[rr][0] = DOT4(w1, TAPS(56, 48, 34, 18), DOT4(w0, TAPS(0, 18, 34, 48), 0u));
      o[rr][1] = DOT4(w2, TAPS(18, 0, 0, 0), DOT4(w1, TAPS(48, 56, 48, 34), DOT4(w0, TAPS(0, 0, 18, 34), 0u)));
      o[rr][2] = DOT4(w2, TAPS(34, 18, 0, 0), DOT4(w1, TAPS(34, 48, 56, 48), DOT4(w0, TAPS(0, 0, 0, 18), 0u)));
      o[rr][3] = DOT4(w2, TAPS(48, 34, 18, 0), DOT4(w1, TAPS(18, 34, 48, 56), 0u));
#undef DOT4
#undef TAPS
    }
    uint4 st;
    st.x = o[0][0] | (o[1][0] << 16);
    st.y = o[0][1] | (o[1][1] << 16);
    st.z = o[0][2] | (o[1][2] << 16);
    st.w = o[0][3] | (o[1][3] << 16);
    *reinterpret_cast<uint4*>(hbp + k * BT_HP + 4 * g) = st;
  }
  __syncthreads();
  // vertical pass: thread = (4-pixel group g, row pair rp): output rows 2rp, 2rp+1 from the row pairs rp .. rp+3
  {
    typedef __attribute__((ext_vector_type(2))) unsigned short us2;
    for (int it = tid; it < (BT_H / 2) * 16; it += THREADS) {
    const int g = it & 15, rp = it >> 4;
    uint4 P[4];
#pragma unroll
    for (int k = 0; k < 4; k++) P[k] = *reinterpret_cast<const uint4*>(hbp + (rp + k) * BT_HP + 4 * g);
    const uint32_t E0 = 18u | (34u << 16), E1 = 48u | (56u << 16), E2 = 48u | (34u << 16), E3 = 18u;        // even row
    const uint32_t O0 = 18u << 16, O1 = 34u | (48u << 16), O2 = 56u | (48u << 16), O3 = 34u | (18u << 16);  // odd row
    uint32_t aE[4], aO[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const uint32_t p0 = i == 0 ? P[0].x : i == 1 ? P[0].y : i == 2 ? P[0].z : P[0].w;
      const uint32_t p1 = i == 0 ? P[1].x : i == 1 ? P[1].y : i == 2 ? P[1].z : P[1].w;
      const uint32_t p2 = i == 0 ? P[2].x : i == 1 ? P[2].y : i == 2 ? P[2].z : P[2].w;
      const uint32_t p3 = i == 0 ? P[3].x : i == 1 ? P[3].y : i == 2 ? P[3].z : P[3].w;
#define UD2(a, b, c) __builtin_amdgcn_udot2(__builtin_bit_cast(us2, (uint32_t)(a)), __builtin_bit_cast(us2, (uint32_t)(b)), (c), false)
      const uint32_t accE = UD2(p3, E3, UD2(p2, E2, UD2(p1, E1, UD2(p0, E0, 32768u))));
      const uint32_t accO = UD2(p3, O3, UD2(p2, O2, UD2(p1, O1, UD2(p0, O0, 32768u))));
#undef UD2
      aE[i] = accE;
      aO[i] = accO;
    }
    // every sum is < 2^24 (256 * 65280 + 32768), so the rounded pixel IS byte 2 of its accumulator: two v_perm + one or per
    // four pixels instead of a shift and a shift-or each (all of them 4-clock instructions)
    const uint32_t outE = __builtin_amdgcn_perm(aE[1], aE[0], 0x0c0c0602u) | __builtin_amdgcn_perm(aE[3], aE[2], 0x06020c0cu);
    const uint32_t outO = __builtin_amdgcn_perm(aO[1], aO[0], 0x0c0c0602u) | __builtin_amdgcn_perm(aO[3], aO[2], 0x06020c0cu);
    const int gy = oy + 2 * rp, gx = ox + 4 * g;
    if (gx < w) {   // gx is a multiple of 4: the four pixels lie in one tile row; gy is even: row gy + 1 is the next row of the same tile
      uint8_t* o = D + blur_tiled_offset(gx, gy, dst.pitch[lvl]);
#if BT_NT_STORE
      if (gy < h) __builtin_nontemporal_store(outE, reinterpret_cast<uint32_t*>(o));
      if (gy + 1 < h) __builtin_nontemporal_store(outO, reinterpret_cast<uint32_t*>(o + 16));
#else
      if (gy < h) *reinterpret_cast<uint32_t*>(o) = outE;
      if (gy + 1 < h) *reinterpret_cast<uint32_t*>(o + 16) = outO;
#endif
    }
    }
  }
}
#endif

// ---- the same blur on the matrix cores.  A 7-tap pass over a row is Out = In x Band, a band (Toeplitz) matrix of the taps; with
// pixels as i8 (x ^ 0x80 = x - 128) and taps <= 56 (folded REFLECT_101 taps <= 96) v_mfma_i32_16x16x64_i8 computes it exactly.
// One wave owns a strip of 48 output columns and walks down the level:
//   horizontal: A = 16 rows x 64 pixels, ONE aligned 16-byte load per lane straight into the operand layout (lane (q, r): row r,
//     pixels 16q .. 16q + 15 of the segment [48c - 16, 48c + 48)); the columns that have all seven taps inside the segment, on
//     4-pixel boundaries, are the strip: [48c - 12, 48c + 36).  B = the strip's three band matrices (64 x 16, from
//     the plan's table, in registers for the whole walk); C = 128.  D: lane (q, c) holds H - 32768 + 128 of rows 4q .. 4q + 3 of
//     column c -- a signed 16-bit number whose bytes (high, low ^ 0x80) are the i8 digits of H - 32768 = 256 a + b;
//   vertical: those bytes, packed four rows to a dword, ARE the operand of the second product: a lane's 16 bytes are the rows
//     16 ww + 4q + i (ww = 0..3 <-> dword, i <-> byte) of a 64-row window -- the K order of a matrix product is free as long as
//     both operands agree, and the other operand (the vertical band matrix, one pair for every window: rows outside the level are
//     loaded from their REFLECT_101 source) is written in that order.  With the
//     data as A and the band matrix as B the result comes out transposed: lane (q, r) holds columns 4q .. 4q + 3 of output row r,
//     one dword of the 16 x 8-tiled blurred plane.  V = 256 (T a) + (T b) + 256 * 32768, pixel = (V + 32768) >> 16 = byte 2.
// A window of four 16-row blocks yields the two middle ones; the walk advances by two blocks.  No LDS, no barrier.
#ifndef BLUR_RING
#define BLUR_RING 2         // passes of source rows in flight per wave (LDS-DMA ring: 2 KB per pass and wave)
#endif
#ifndef BLUR_MIN_BLOCKS
#define BLUR_MIN_BLOCKS 4   // workgroups per CU the register allocation must leave room for
#endif
// One 1 KB piece (16 rows x 64 bytes) global -> LDS without passing registers: lane l's 16 bytes land at lds_dst + 16 l.  M0
// (the destination) is the compiler's register: saved and restored inside the statement.  The compiler does not count this
// load: every wait for it is the explicit BLUR_WAIT_ROWS below.
#define BLUR_DMA(gsrc, lds_dst) do { \
    unsigned keep_m0; \
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" \
                 : "=&s"(keep_m0) : "v"(gsrc), "s"(lds_dst) : "memory"); \
  } while (0)
// vector-memory operations a wave issues AFTER the two pieces of pass p and before it reads them (at the end of pass p - 1):
// per pass in between two pieces and six stores (every lane stores in every pass): (BLUR_RING - 1) * 6 + (BLUR_RING - 2) * 2
#if BLUR_RING == 2
#define BLUR_WAIT_ROWS() asm volatile("s_waitcnt vmcnt(6)" ::: "memory")
#elif BLUR_RING == 3
#define BLUR_WAIT_ROWS() asm volatile("s_waitcnt vmcnt(14)" ::: "memory")
#elif BLUR_RING == 4
#define BLUR_WAIT_ROWS() asm volatile("s_waitcnt vmcnt(22)" ::: "memory")
#else
#error "BLUR_RING must be 2, 3 or 4"
#endif
__global__ __launch_bounds__(256, BLUR_MIN_BLOCKS) void gauss_blur7_mfma_kernel(PyrView src, PyrView dst, BlurMfmaParams P) {
  const int lane = threadIdx.x & 63;
  const int sid = blockIdx.x * 4 + (threadIdx.x >> 6);
  __shared__ v4i ring[4][BLUR_RING][2][64];
  if (sid >= P.n_strips) return;
  const BlurStrip st = P.strips[sid];
  const int lvl = __builtin_amdgcn_readfirstlane((int)st.level), ch = __builtin_amdgcn_readfirstlane((int)st.chunk);
  const int w = src.w[lvl], h = src.h[lvl], sp = src.pitch[lvl], dp = dst.pitch[lvl];
  const uint8_t* S = src.base[lvl] + (size_t)blockIdx.y * src.img_stride[lvl];
  uint8_t* D = const_cast<uint8_t*>(dst.base[lvl]) + (size_t)blockIdx.y * dst.img_stride[lvl];
  const int q = lane >> 4, r = lane & 15;
  const v4i* bt = reinterpret_cast<const v4i*>(P.tab + P.b_off[lvl]) + (size_t)(ch * 3) * 64 + lane;
  const v4i B0 = bt[0], B1 = bt[64], B2 = bt[128];
  const v4i* tt = reinterpret_cast<const v4i*>(P.tab + P.t_off) + lane;
  const v4i T1 = tt[0], T2 = tt[64];
  // Load role: lane l fetches row l >> 2, 16-byte piece lc of the 64-byte segment -- four lanes read 64 contiguous bytes; a lane
  // loading its own operand bytes (16 rows per 16 consecutive lanes) costs four times the cache-line requests -- into LDS slot l.
  // The operand of lane (q, r), row r piece q, is then slot 4r + (q ^ (r >> 2 & 3)): the xor on the SOURCE piece spreads the
  // sixteen rows one b128 read serves over all banks.  A segment that would start left of the image or end right of the pitch
  // is moved inside: the band matrices hold zeros for every pixel outside [0, w).
  const int lrow = lane >> 2, lc = (lane & 3) ^ ((lane >> 4) & 3);
  const int xo = min(max(ORBFE_BLUR_CHUNK * ch - 16 + 16 * lc, 0), sp - 16);
  const int rslot = 4 * r + (q ^ ((r >> 2) & 3));
  const uint8_t* scol = S + xo;
  v4i* const wring = &ring[threadIdx.x >> 6][0][0][0];
  const uint32_t lds0 = __builtin_amdgcn_readfirstlane((int)(uint32_t)(size_t)(__attribute__((address_space(3))) v4i*)wring);
  const v4i c128 = {128, 128, 128, 128}, zero = {0, 0, 0, 0};
  const v4i cfin = {8388608 + 32768, 8388608 + 32768, 8388608 + 32768, 8388608 + 32768};
  // The two pieces of pass p: rows [32 p + 16, 32 p + 32) and [32 p + 32, 32 p + 48), into ring stage `stage`.  Rows outside
  // the level are fetched from their REFLECT_101 source row (one reflection, then clamped: only rows within three of the level
  // matter, and h >= 8), so the vertical band matrix is the plain Toeplitz one in every window.
#define BLUR_ROW(y) min(max((y) < 0 ? -(y) : ((y) >= h ? 2 * (h - 1) - (y) : (y)), 0), h - 1)
#define BLUR_REQUEST(p, stage) do { \
    const uint8_t* g0 = scol + (size_t)BLUR_ROW(32 * (p) + 16 + lrow) * sp; \
    const uint8_t* g1 = scol + (size_t)BLUR_ROW(32 * (p) + 32 + lrow) * sp; \
    BLUR_DMA(g0, lds0 + (uint32_t)(stage) * 2048u); \
    BLUR_DMA(g1, lds0 + (uint32_t)(stage) * 2048u + 1024u); \
  } while (0)
#define BLUR_HBLOCK(a, slot) do { \
    const v4i ax = (a) ^ (int)0x80808080; \
    const v4i d0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(ax, B0, c128, 0, 0, 0); \
    const v4i d1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(ax, B1, c128, 0, 0, 0); \
    const v4i d2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(ax, B2, c128, 0, 0, 0); \
    int ph, pl; \
    blur_pack(d0, ph, pl); Xh[0].slot = ph; Xl[0].slot = pl; \
    blur_pack(d1, ph, pl); Xh[1].slot = ph; Xl[1].slot = pl; \
    blur_pack(d2, ph, pl); Xh[2].slot = ph; Xl[2].slot = pl; \
  } while (0)
  v4i Xh[3] = {zero, zero, zero}, Xl[3] = {zero, zero, zero};   // window blocks 0..3 <-> .x .y .z .w, per column group
  const int n_win = (h + ORBFE_BLUR_WINDOW - 1) / ORBFE_BLUR_WINDOW;
  // Pass s of the walk: first product for rows [32 s + 16, 32 s + 48) -- blocks 2 and 3 of window s --, second product for the
  // two middle blocks of the window, then the window moves down by two blocks.  Pass -1 only fills blocks 0 and 1 (rows -16 .. 15).  The rows of pass p are requested BLUR_RING passes ahead into stage
  // (p + 1) mod BLUR_RING.  Memory operations retire in issue order; inside a pass the order is: products; the next pass's rows
  // LDS -> operands (behind a counted wait: its pieces are older than exactly BLUR_YOUNGER operations); the request for the
  // pass whose stage this pass just vacated; this pass's stores.
  int st_cur = 0;   // stage of pass s
#pragma unroll
  for (int p = -1; p < BLUR_RING - 1; p++) BLUR_REQUEST(p, p + 1);
  const int dtile = dp >> 4;
  uint32_t xoff[3], rowoff[2];
  bool okx[3];
  const uint32_t trash = (uint32_t)dst.img_stride[lvl] - 256u + 4u * (uint32_t)lane;   // extractor.cpp: every plane ends in 256 spare bytes
#pragma unroll
  for (int g = 0; g < 3; g++) {
    const int x = ORBFE_BLUR_CHUNK * ch - 12 + 16 * g + 4 * q;   // a multiple of 4: the dword lies in one tile row
    okx[g] = x >= 0 && x < w;
    xoff[g] = ((uint32_t)(x >> 4) << 7) + (uint32_t)(x & 15);
  }
#pragma unroll
  for (int b = 0; b < 2; b++) {   // output row of pass -1: 16 b + r - 32 (never stored); the offset is taken modulo 2^32
    const int o = 16 * b + r;
    rowoff[b] = (((uint32_t)(o >> 3) * (uint32_t)dtile) << 7) + (uint32_t)((o & 7) * 16) - ((uint32_t)dtile << 9);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the first pieces (and this wave's tables) have landed
  v4i A2 = wring[rslot], A3 = wring[64 + rslot];
  for (int s = -1; s < n_win; s++) {
    BLUR_HBLOCK(A2, z);
    BLUR_HBLOCK(A3, w);
    uint32_t out[3][2] = {{0u, 0u}, {0u, 0u}, {0u, 0u}};
    if (s >= 0) {
#pragma unroll
      for (int g = 0; g < 3; g++) {
#pragma unroll
        for (int b = 0; b < 2; b++) {
          const v4i T = b ? T2 : T1;
          const v4i vh = __builtin_amdgcn_mfma_i32_16x16x64_i8(Xh[g], T, zero, 0, 0, 0);
          const v4i vl = __builtin_amdgcn_mfma_i32_16x16x64_i8(Xl[g], T, cfin, 0, 0, 0);
          const uint32_t r0 = ((uint32_t)vh.x << 8) + (uint32_t)vl.x, r1 = ((uint32_t)vh.y << 8) + (uint32_t)vl.y;
          const uint32_t r2 = ((uint32_t)vh.z << 8) + (uint32_t)vl.z, r3 = ((uint32_t)vh.w << 8) + (uint32_t)vl.w;
          out[g][b] = __builtin_amdgcn_perm(r1, r0, 0x0c0c0602u) | __builtin_amdgcn_perm(r3, r2, 0x06020c0cu);
        }
      }
    }
#pragma unroll
    for (int g = 0; g < 3; g++) {
      Xh[g].x = Xh[g].z; Xh[g].y = Xh[g].w;
      Xl[g].x = Xl[g].z; Xl[g].y = Xl[g].w;
    }
    // rows of pass s + 1: LDS -> operands
    const int st_next = st_cur + 1 == BLUR_RING ? 0 : st_cur + 1;
    BLUR_WAIT_ROWS();
    A2 = wring[st_next * 128 + rslot];
    A3 = wring[st_next * 128 + 64 + rslot];
    BLUR_REQUEST(s + BLUR_RING, st_cur);   // the stage of pass s was read a pass ago
    st_cur = st_next;
    // this pass's stores: the tiled offset of (x, o) splits into a column part (fixed for the strip) and a row part that
    // advances by four tile rows per pass.  Every lane stores in every pass -- a store under a condition is a branch to the
    // compiler, and BLUR_WAIT_ROWS counts on six stores per pass: a dword outside the level goes to the lane's own slot
    // in the 256 spare bytes behind the plane
    {
#pragma unroll
      for (int b = 0; b < 2; b++) {
        const bool oky = s >= 0 && ORBFE_BLUR_WINDOW * s + 16 * b + r < h;
#pragma unroll
        for (int g = 0; g < 3; g++)
          *reinterpret_cast<uint32_t*>(D + ((oky && okx[g]) ? rowoff[b] + xoff[g] : trash)) = out[g][b];
        rowoff[b] += (uint32_t)dtile << 9;
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // pieces requested past the last pass must not land in LDS after the wave has gone
#undef BLUR_REQUEST
#undef BLUR_ROW
#undef BLUR_HBLOCK
}

// ------------------------------------------------------------------------------------------------ describe
// the same pattern as floats, one (x0, y0, x1, y1) record per test pair (filled at start-up by the host)
__device__ __attribute__((aligned(16))) float g_pattern_f[1024];
// umax of ORBextractor's constructor for HALF_PATCH_SIZE 15 (L/src/ORBextractor.cc:449-463)

// cv::fastAtan2 (degrees), scalar OpenCV path, float, un-fused
__device__ __forceinline__ float fast_atan2_deg(float y, float x) {
  const float scale = (float)(180.0 / 3.1415926535897932384626433832795);
  const float p1 = 0.9997878412794807f * scale;
  const float p3 = -0.3258083974640975f * scale;
  const float p5 = 0.1555786518463281f * scale;
  const float p7 = -0.04432655554792128f * scale;
  const float eps = (float)2.2204460492503131e-16;
  const float ax = fabsf(x), ay = fabsf(y);
  float a, c, c2;
  if (ax >= ay) {
    c = ay / (ax + eps);
    c2 = c * c;
    a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  } else {
    c = ax / (ay + eps);
    c2 = c * c;
    a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  }
  if (x < 0) a = 180.f - a;
  if (y < 0) a = 360.f - a;
  return a;
}

// glibc 2.35 sinf/cosf (flt-32/s_sincosf.h) restated: double reduction by pi/2, two double polynomials,
// one rounding to float.  Valid for |x| < 120 (angles here are in [0, 2*pi]).
__device__ __forceinline__ float sc_poly(double x, double x2, int n, bool flip) {
  const double C0 = 0x1p0, C1 = -0x1.ffffffd0c621cp-2, C2 = 0x1.55553e1068f19p-5, C3 = -0x1.6c087e89a359dp-10,
               C4 = 0x1.99343027bf8c3p-16;
  const double S1 = -0x1.555545995a603p-3, S2 = 0x1.1107605230bc4p-7, S3 = -0x1.994eb3774cf24p-13;
  if ((n & 1) == 0) {
    const double x3 = x * x2;
    const double s1 = S2 + x2 * S3;
    const double x7 = x3 * x2;
    const double s = x + x3 * S1;
    return (float)(s + x7 * s1);
  } else {
    const double sg = flip ? -1.0 : 1.0;
    const double x4 = x2 * x2;
    const double c2 = sg * C3 + x2 * (sg * C4);
    const double c1 = sg * C1 + x2 * (sg * C2);
    const double x6 = x4 * x2;
    const double c = sg * C0 + x2 * c1;
    return (float)(c + x6 * c2);
  }
}
__device__ __forceinline__ void glibc_sincosf(float y, float* sn, float* cs) {
  double x = (double)y;
  const uint32_t top = (__float_as_uint(y) >> 20) & 0x7ff;
  if (top < 0x3f4) {  // |y| < pi/4  (abstop12(0x1.921FB6p-1f) = 0x3f4)
    const double x2 = x * x;
    if (top < 0x398) {  // |y| < 2^-12
      *sn = y;
      *cs = 1.0f;
      return;
    }
    *sn = sc_poly(x, x2, 0, false);
    *cs = sc_poly(x, x2, 1, false);
    return;
  }
  const double r = x * 0x1.45F306DC9C883p+23;
  const int n = ((int32_t)r + 0x800000) >> 24;
  x = x - (double)n * 0x1.921FB54442D18p0;
  const double s = ((n + 1) & 2) ? -1.0 : 1.0;  // sign table {1,-1,-1,1}[n&3]
  const bool flip = (n & 2) != 0;
  const double xs = x * s, x2 = x * x;
  *sn = sc_poly(xs, x2, n, flip);
  *cs = sc_poly(xs, x2, n ^ 1, flip);
}

// LDS staging of a keypoint's two patches: the raw 31 x 31 window of IC_Angle and the blurred 37 x 37 window of the pattern
#ifndef ORI_PITCH
#define ORI_PITCH 48   // 3 x 16 B: (cx - 15) & 15 <= 15, 15 + 31 <= 48
#endif
#ifndef DSC_PITCH
#define DSC_PITCH 64   // 4 x 16 B: 15 + 37 <= 64
#endif
// a patch row's 16-byte piece into LDS: one b128 store where the pitch keeps the pieces 16-byte aligned, four dwords otherwise
template <int PITCH>
__device__ __forceinline__ void patch_store16(uint8_t* base, int r, int c, const uint4& v) {
  if constexpr (PITCH % 16 == 0) {
    *reinterpret_cast<uint4*>(base + r * PITCH + 16 * c) = v;
  } else {
    uint32_t* q = reinterpret_cast<uint32_t*>(base + r * PITCH + 16 * c);
    q[0] = v.x; q[1] = v.y; q[2] = v.z; q[3] = v.w;
  }
}
#define ORI_BYTES ((31 * ORI_PITCH + 15) & ~15)            // 1488
#ifndef OD_DMA_STAGE
#define OD_DMA_STAGE 1   // the windows go global -> LDS directly (global_load_lds_dwordx4), two buffers per wave; 0: through registers (rounds 2-5)
#endif
#define OD_BUF (37 * DSC_PITCH)                            // 2368: one LDS buffer holds either window
#if OD_DMA_STAGE
#define PATCH_BYTES (2 * OD_BUF)                           // 4736
// lane l's 16 bytes land at lds_dst + 16 l: the windows' LDS images are exactly that order -- raw patch 31 rows x 3 pieces at pitch 48,
// blurred patch 37 rows x 4 pieces at pitch 64 or x 3 pieces at pitch 48 (piece i = row * pieces + column at byte 16 i).  The compiler
// does not count these loads: every wait for them is an explicit s_waitcnt vmcnt below.  M0 (the destination) is saved and restored.
#define OD_DMA(gsrc, lds_dst) do { \
    unsigned keep_m0; \
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" \
                 : "=&s"(keep_m0) : "v"(gsrc), "s"(lds_dst) : "memory"); \
  } while (0)
#else
#define PATCH_BYTES ((ORI_BYTES + 37 * DSC_PITCH + 15) & ~15)  // 3856
#endif

// ---- orientation + description, eight keypoints per wave ------------------------------------------------------------------
// The per-keypoint work has two kinds of instructions: cooperative ones (patch staging, the 749-pixel moment sums, the 256
// rotated comparisons) and wave-uniform ones (level bookkeeping, fastAtan2, the double-precision sin / cos: ~130 of the ~380
// instructions of a wave-per-keypoint kernel, each of them computing ONE value on 64 lanes).  Here a wave takes eight consecutive
// keypoint slots: phase 1 stages each raw 31 x 31 patch and leaves the keypoint's moments in lane k; phase 2 evaluates
// fastAtan2 / sinf / cosf once, lane k for keypoint k; phase 3 stages each blurred 37 x 37 patch and samples the pattern with
// (a, b) read from lane k.  The moments use v_dot4_i32_i8: a lane owns four (row, 4-column) items of the disc, the per-item
// weights (u and v as signed bytes, 0 outside the disc; a host-filled table) stay in registers for all eight keypoints, and the
// pixels enter as I - 128 (one xor per dword).  The disc is symmetric, sum u = sum v = 0, so sum u (I - 128) = sum u I = m10
// exactly -- integer, hence the same moments as the reference's loops -- and no sum of I is needed: two wave reductions per
// keypoint instead of three.
#ifndef OD_K
#define OD_K 8
#ifndef OD_DSC_NARROW
#define OD_DSC_NARROW 1   // three 16-byte pieces per row of the blurred window where its 37 columns fit them (0: always four)
#endif
#endif
#ifndef OD_FAKE_TILED_RAW
#define OD_FAKE_TILED_RAW 0
#endif
#ifndef OD_FAKE_TABLES
#define OD_FAKE_TABLES 0
#endif
#if FC_TIMING
__device__ unsigned long long g_od_prof[4096 * 8];
extern "C" int orbfe_debug_od_profile(unsigned long long* out, int reset) {
  static unsigned long long h[4096 * 8];
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_od_prof), sizeof(h)) != hipSuccess) return 1;
  for (int i = 0; i < 8; i++) out[i] = 0;
  for (int sl = 0; sl < 4096; sl++)
    for (int i = 0; i < 8; i++) out[i] += h[sl * 8 + i];
  if (reset) { memset(h, 0, sizeof(h)); if (hipMemcpyToSymbol(HIP_SYMBOL(g_od_prof), h, sizeof(h)) != hipSuccess) return 1; }
  return 0;
}
#endif
__device__ __attribute__((aligned(16))) uint32_t g_ic_w[256 * 2];   // [item][u-weights | v-weights] (int8 x 4), item = row * 8 + 4-column group
#ifndef OD_WGS
#define OD_WGS 7   // workgroups (= waves per SIMD) resident per CU: the register budget the compiler is given and the LDS padding below
#endif
__global__ __launch_bounds__(256, OD_WGS) void orient_describe8_kernel(DescribeParams P) {
  __shared__ __attribute__((aligned(16))) uint8_t patch[4][PATCH_BYTES];
  const int lane = threadIdx.x & (WAVE - 1);
  const int wv_id = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  int bx = blockIdx.x, img = blockIdx.y;
  if (P.xcd_images) {   // whole images per XCD (XCD k takes images k, k + 8, ...): the planes one image's patches are gathered from stay in one L2
    const unsigned gx = gridDim.x;
    const unsigned lin = blockIdx.y * gx + blockIdx.x;
    const unsigned grp = lin / (8u * gx);
    if (8u * grp + 8u <= gridDim.y) {
      const unsigned within = lin - grp * 8u * gx;
      img = (int)(8u * grp + (within & 7u));
      bx = (int)(within >> 3);
    }
  }
#if FC_TIMING
  uint32_t tacc0 = 0, tacc1 = 0, tacc2 = 0, tacc3 = 0, tacc4 = 0, tacc5 = 0, tprev = (uint32_t)__builtin_readcyclecounter();
#endif
  const int s0 = (bx * 4 + wv_id) * OD_K;
  const int32_t* ln = P.lvl_n + (size_t)img * ORBFE_MAX_LEVELS;
  if (s0 >= P.kp_per_image) return;
  // Everything the prologue needs is requested before anything is waited for: the level counts (lane l holds ln[l]), the
  // slot's packed record, the pattern and the moment weights.  (A store in front of these loads, or a load per loop
  // iteration, serialised four or five memory round trips here: a fifth of the kernel's time, tools/fc_phase_profile.py.)
  const int ln_lane = lane < P.n_levels ? ln[lane] : 0;
  const int slot_c = min(s0 + (lane & (OD_K - 1)), P.kp_per_image - 1);
  const uint32_t e_lane = P.lvl_kp[(size_t)img * P.kp_per_image + slot_c];
  float4 pk[4];
#pragma unroll
#if OD_FAKE_TABLES   // timing experiment only (wrong results): no table loads
  for (int r = 0; r < 4; r++) pk[r] = make_float4((float)(lane & 7) - 3.f, (float)(r * 2 + (lane >> 4)) - 5.f, (float)(lane & 15) - 8.f, (float)(r) - 2.f);
#else
  for (int r = 0; r < 4; r++) pk[r] = *reinterpret_cast<const float4*>(&g_pattern_f[(r * 64 + lane) * 4]);
#endif
  uint32_t wu[4], wv[4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int it = lane + WAVE * j;   // items >= 248 carry zero weights
#if OD_FAKE_TABLES
    wu[j] = 0x01020304u * (uint32_t)(it & 7); wv[j] = 0x01010101u * (uint32_t)(it >> 3);
#else
    const uint2 w2 = reinterpret_cast<const uint2*>(g_ic_w)[it];
    wu[j] = w2.x; wv[j] = w2.y;
#endif
  }
#if !OD_DMA_STAGE
  uint8_t* ori = &patch[wv_id][0];
  uint8_t* dsc = ori + ORI_BYTES;
#endif

  // slot bookkeeping, lane k for slot s0 + k: level, position, score and output index (-1 = no keypoint in this slot)
  int i_out = -1, i_level = 0, i_cx = 0, i_cy = 0, i_score = 0;
  {
    const int slot = s0 + lane;
    int level = 0, off_level = 0;
    for (int l = 1; l < P.n_levels; l++) {   // kp_off ascends; it lives in the kernel arguments (scalar registers)
      const bool ge = slot >= P.kp_off[l];
      level += ge ? 1 : 0;
      off_level = ge ? P.kp_off[l] : off_level;
    }
    const int idx = slot - off_level;
    int out = idx, ln_level = 0;
    for (int l = 0; l < P.n_levels; l++) {
      const int lnl = __builtin_amdgcn_readlane(ln_lane, l);
      out += l < level ? lnl : 0;
      ln_level = l == level ? lnl : ln_level;
    }
    if (lane < OD_K && slot < P.kp_per_image && idx < ln_level && out < P.cap) {
      const uint32_t e = e_lane;
      i_out = out; i_level = level;
      i_cx = (int)(e & 0xfff) + ORBFE_EDGE; i_cy = (int)((e >> 12) & 0xfff) + ORBFE_EDGE; i_score = (int)(e >> 24);
    }
    if (s0 == 0 && lane == 0) {   // the image's keypoint count = sum of the level counts
      int tot = 0;
      for (int l = 0; l < P.n_levels; l++) tot += __builtin_amdgcn_readlane(ln_lane, l);
      P.out_n[img] = tot;
    }
  }
  const unsigned valid_mask = (unsigned)(__ballot(i_out >= 0) & ((1ull << OD_K) - 1ull));
  if (valid_mask == 0) return;
  FC_T(0);   // tables + slot bookkeeping

#if OD_DSC_NARROW && !OD_DMA_STAGE
  int r3[3], c3[3];   // lane -> (row, piece) of the three-piece blurred window
#pragma unroll
  for (int j = 0; j < 3; j++) { const int i = lane + WAVE * j; r3[j] = i / 3; c3[j] = i - 3 * r3[j]; }
#endif
#if !OD_DMA_STAGE
  // the loads of keypoint k's raw patch (two 16-byte pieces per lane) / blurred patch (three)
  auto issue_ori = [&](int k, uint4 vo[2]) {
    const int level = __builtin_amdgcn_readlane(i_level, k), cx = __builtin_amdgcn_readlane(i_cx, k), cy = __builtin_amdgcn_readlane(i_cy, k);
    const int pitch = P.pyr.pitch[level];
    const uint8_t* plane = P.pyr.base[level] + (size_t)img * P.pyr.img_stride[level];
    const int ax_o = (cx - 15) & ~15;
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int i = lane + WAVE * j;
      const int r = i / 3, c = i - r * 3;
#if OD_FAKE_TILED_RAW   // timing experiment only (wrong angles): the raw patch fetched from the TILED blurred plane -- what the gather would cost with the
      (void)pitch; (void)plane;   // raw levels in 16 x 8 tiles: 0.346 -> 0.277 ms (DESIGN lesson 44)
      vo[j] = i < 31 * 3 ? *reinterpret_cast<const uint4*>(P.blur.base[level] + (size_t)img * P.blur.img_stride[level] + blur_tiled_offset(ax_o + 16 * c, cy - 15 + r, P.blur.pitch[level])) : make_uint4(0, 0, 0, 0);
#else
      vo[j] = i < 31 * 3 ? *reinterpret_cast<const uint4*>(plane + orbfe_level_offset(ax_o + 16 * c, cy - 15 + r, pitch, (P.pyr.tiled >> level) & 1u)) : make_uint4(0, 0, 0, 0);
#endif
    }
  };
  auto issue_dsc = [&](int k, uint4 vd[3]) {
    const int level = __builtin_amdgcn_readlane(i_level, k), cx = __builtin_amdgcn_readlane(i_cx, k), cy = __builtin_amdgcn_readlane(i_cy, k);
    const int bpitch = P.blur.pitch[level];
    const uint8_t* bplane = P.blur.base[level] + (size_t)img * P.blur.img_stride[level];
    const int ax_d = (cx - 18) & ~15;
#if OD_DSC_NARROW
    // Where the window's 37 columns end inside the third 16-byte piece (three keypoints of four) only 37 x 3 pieces are requested, in two
    // rounds -- the kernel is bound by the texture addresser (TA / TD / TCP 0.95-0.99 busy), i.e. by pieces requested.  One code path:
    // the lane -> (row, piece) map and the count are selected by the wave-uniform flag (a branch around the loads would make the
    // compiler wait for them at its end: the next keypoint's patch could no longer be in flight during this one's tests)
    const bool wide = ((cx - 18) & 15) > 11;
    const int n_pieces = wide ? 37 * 4 : 37 * 3;
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const int i = lane + WAVE * j;
      const int r = wide ? (i >> 2) : r3[j], c = wide ? (i & 3) : c3[j];
      vd[j] = i < n_pieces ? *reinterpret_cast<const uint4*>(bplane + blur_tiled_offset(max(ax_d + 16 * c, 0), max(cy - 18 + r, 0), bpitch)) : make_uint4(0, 0, 0, 0);
    }
#else
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const int i = lane + WAVE * j;
      const int r = i >> 2, c = i & 3;
      vd[j] = i < 37 * 4 ? *reinterpret_cast<const uint4*>(bplane + blur_tiled_offset(max(ax_d + 16 * c, 0), max(cy - 18 + r, 0), bpitch)) : make_uint4(0, 0, 0, 0);
    }
#endif
  };
#endif
  // next valid slot after k (OD_K if none)
  auto next_valid = [&](int k) { const unsigned m = valid_mask >> (k + 1); return m ? k + 1 + (__ffs((int)m) - 1) : OD_K; };
  const int k_first = __ffs((int)valid_mask) - 1;

#if OD_DMA_STAGE
  // ---- phase 1: moments of the keypoints, keypoint k's in lane k.  The window of keypoint k + 1 travels global -> LDS (the other
  // buffer) while k's is summed: no staging registers, no LDS store instructions, no wait between a load's arrival and its store.
  const uint32_t lds_buf = (uint32_t)(uintptr_t)&patch[wv_id][0];
  auto dma_ori = [&](int k, uint32_t lds) {
    const int level = __builtin_amdgcn_readlane(i_level, k), cx = __builtin_amdgcn_readlane(i_cx, k), cy = __builtin_amdgcn_readlane(i_cy, k);
    const int pitch = P.pyr.pitch[level];
    const uint8_t* plane = P.pyr.base[level] + (size_t)img * P.pyr.img_stride[level];
    const bool tiled = (P.pyr.tiled >> level) & 1u;
    const int ax_o = (cx - 15) & ~15;
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int i = lane + WAVE * j;
      const int r = (i * 43) >> 7, c = i - 3 * r;   // i / 3, i % 3 for i < 128
      if (i < 31 * 3) OD_DMA(plane + orbfe_level_offset(ax_o + 16 * c, cy - 15 + r, pitch, tiled), lds + 1024u * j);
    }
  };
  int m10v = 0, m01v = 0;
  {
    // every load the compiler knows of (pattern, weights, slot records) has arrived before the first window is requested: it does not
    // see the LDS-DMA loads, and would otherwise wait for "its" loads inside the loops below -- with a count that also drains the
    // window that was just requested (s_waitcnt vmcnt(0), expcnt / lgkmcnt untouched)
    // (the first window is requested in front of that wait: its latency passes together with the tables')
    dma_ori(k_first, lds_buf);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    uint32_t par = 0;
    for (int k = k_first; k < OD_K; k = next_valid(k), par ^= 1u) {
      const int kn = next_valid(k);
      // (the buffer the next window lands in was last read two keypoints ago; those reads were consumed before that iteration ended.
      //  Three buffers with the windows of k + 1 AND k + 2 in flight: 0.269 ms against 0.262 -- more requests in flight are not faster)
      if (kn < OD_K) {
        dma_ori(kn, lds_buf + (par ^ 1u) * OD_BUF);
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");   // everything older than the two loads just issued: this keypoint's window
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      FC_T(1);   // wait for the raw patch
      const uint8_t* ob = &patch[wv_id][0] + par * OD_BUF;
      const int cx = __builtin_amdgcn_readlane(i_cx, k);
      const int m = (cx - 15) & 15;
      int A = 0, B = 0;
      const uint32_t sh = (uint32_t)(m & 3);
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int it = lane + WAVE * j;
        const int r = it >> 3, c = it & 7;
        const uint32_t* q = reinterpret_cast<const uint32_t*>(ob + (r < 31 ? r : 30) * ORI_PITCH + ((m + 4 * c) & ~3));
        // the four pixels u = -15 + 4c .. -12 + 4c of row v = r - 15, as I - 128
        const int px = (int)(__builtin_amdgcn_alignbyte(q[1], q[0], sh) ^ 0x80808080u);
        A = __builtin_amdgcn_sdot4(px, (int)wu[j], A, false);
        B = __builtin_amdgcn_sdot4(px, (int)wv[j], B, false);
      }
      const int At = __builtin_amdgcn_readlane(wave_incl_scan(A), 63), Bt = __builtin_amdgcn_readlane(wave_incl_scan(B), 63);
      if (lane == k) { m10v = At; m01v = Bt; }
      FC_T(2);   // moments
    }
  }
  // ---- phase 2: lane k computes keypoint k's angle and rotation; the first blurred patch is already on its way
  auto dsc_wide = [&](int k) { return ((__builtin_amdgcn_readlane(i_cx, k) - 18) & 15) > 11; };
  auto dma_dsc = [&](int k, uint32_t lds) {
    const int level = __builtin_amdgcn_readlane(i_level, k), cx = __builtin_amdgcn_readlane(i_cx, k), cy = __builtin_amdgcn_readlane(i_cy, k);
    const int bpitch = P.blur.pitch[level];
    const uint8_t* bplane = P.blur.base[level] + (size_t)img * P.blur.img_stride[level];
    const int ax_d = (cx - 18) & ~15;
    // 37 x 3 pieces (two rounds) where the window's 37 columns end inside the third 16-byte piece, 37 x 4 (three) otherwise
    const bool wide = ((cx - 18) & 15) > 11;
    const int n_pieces = wide ? 37 * 4 : 37 * 3;
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const int i = lane + WAVE * j;
      const int q3 = (i * 43) >> 7;
      const int r = wide ? (i >> 2) : q3, c = wide ? (i & 3) : i - 3 * q3;
      if (j < 2 || wide)   // (wave-uniform: the third round exists only for the wide window)
        if (i < n_pieces) OD_DMA(bplane + blur_tiled_offset(max(ax_d + 16 * c, 0), max(cy - 18 + r, 0), bpitch), lds + 1024u * j);
    }
  };
  dma_dsc(k_first, lds_buf);
  const float angle_v = fast_atan2_deg((float)m01v, (float)m10v);
  const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
  float a_v, b_v;
  glibc_sincosf(angle_v * factorPI, &b_v, &a_v);
  FC_T(3);   // angle, sin / cos
  // ---- phase 3: steered BRIEF on the blurred level
  {
    uint32_t par = 0;
    for (int k = k_first; k < OD_K; k = next_valid(k), par ^= 1u) {
      const int level = __builtin_amdgcn_readlane(i_level, k), cx = __builtin_amdgcn_readlane(i_cx, k), cy = __builtin_amdgcn_readlane(i_cy, k);
      const int out = __builtin_amdgcn_readlane(i_out, k), score = __builtin_amdgcn_readlane(i_score, k);
      const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a_v), k)), b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(b_v), k)),
                  angle = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(angle_v), k));
      const int ax_d = (cx - 18) & ~15;
      const bool wide = ((cx - 18) & 15) > 11;
      const int kn = next_valid(k);
      // the queue holds, oldest first: this keypoint's window, the previous keypoint's two stores, then the loads issued here
      if (kn < OD_K) {
        dma_dsc(kn, lds_buf + (par ^ 1u) * OD_BUF);
        if (dsc_wide(kn)) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      FC_T(4);   // wait for the blurred patch
      const uint8_t* dsc = &patch[wv_id][0] + par * OD_BUF;
      // cvRound of the rotated coordinates (L/src/ORBextractor.cc:119-121) by the magic-number addition: v + 1.5 * 2^23 rounds |v| < 2^22
      // to nearest-even in the mantissa's low bits -- one v_add_f32 where v_rndne_f32 + v_cvt_i32_f32 were two.  The integers are never
      // separated from the magic word: in 32-bit wrap-around arithmetic (M + ry) * PITCH + (M + rx) is the patch index plus a constant
      // that goes into the wave-uniform base.  PITCH is 64 for the wide window and 48 for the narrow one (3 * 16: a shift-add more).
      const float MAGIC = 12582912.0f;
      const uint32_t MB = 0x4B400000u;   // its bit pattern
      const uint32_t pitch = wide ? 64u : 48u;
      const uint32_t bc0 = 18u * pitch + (uint32_t)(cx - ax_d) - (MB * pitch + MB);
      uint8_t* dout = P.out_desc + ((size_t)img * P.cap + out) * 32;
      int t0[4], t1[4];
      if (wide) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const float x0 = pk[r].x, y0 = pk[r].y, x1 = pk[r].z, y1 = pk[r].w;
          const uint32_t ry0 = __float_as_uint((x0 * b + y0 * a) + MAGIC), rx0 = __float_as_uint((x0 * a - y0 * b) + MAGIC);
          const uint32_t ry1 = __float_as_uint((x1 * b + y1 * a) + MAGIC), rx1 = __float_as_uint((x1 * a - y1 * b) + MAGIC);
          t0[r] = dsc[bc0 + ry0 * 64u + rx0];
          t1[r] = dsc[bc0 + ry1 * 64u + rx1];
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const float x0 = pk[r].x, y0 = pk[r].y, x1 = pk[r].z, y1 = pk[r].w;
          const uint32_t ry0 = __float_as_uint((x0 * b + y0 * a) + MAGIC), rx0 = __float_as_uint((x0 * a - y0 * b) + MAGIC);
          const uint32_t ry1 = __float_as_uint((x1 * b + y1 * a) + MAGIC), rx1 = __float_as_uint((x1 * a - y1 * b) + MAGIC);
          t0[r] = dsc[bc0 + ry0 * 48u + rx0];
          t1[r] = dsc[bc0 + ry1 * 48u + rx1];
        }
      }
      {
        const unsigned long long b0 = __ballot(t0[0] < t1[0]), b1 = __ballot(t0[1] < t1[1]), b2 = __ballot(t0[2] < t1[2]), b3 = __ballot(t0[3] < t1[3]);
        const unsigned long long mine = lane == 0 ? b0 : lane == 1 ? b1 : lane == 2 ? b2 : b3;
        if (lane < 4) reinterpret_cast<unsigned long long*>(dout)[lane] = mine;
      }
      {
        float fx = (float)cx, fy = (float)cy;
        if (level != 0) {
          fx *= P.scale[level];
          fy *= P.scale[level];
        }
        uint32_t wvv;
        switch (lane) {
          case 0: wvv = __float_as_uint(fx); break;
          case 1: wvv = __float_as_uint(fy); break;
          case 2: wvv = __float_as_uint(P.kp_size[level]); break;
          case 3: wvv = __float_as_uint(angle); break;
          case 4: wvv = __float_as_uint((float)score); break;
          case 5: wvv = (uint32_t)level; break;
          default: wvv = 0xFFFFFFFFu; break;
        }
        // (every keypoint issues exactly these two stores: the wait above counts on it)
        if (lane < 7) reinterpret_cast<uint32_t*>(P.out_kps + (size_t)img * P.cap + out)[lane] = wvv;
      }
      FC_T(5);   // BRIEF + stores
    }
  }
#else
  // ---- phase 1: moments of the keypoints, keypoint k's in lane k.
  // The loads of keypoint k + 1 are in flight while k is summed.  (Requesting the raw patches of TWO keypoints with three load instructions
  // instead of four, lane -> (patch, row, piece): 0.2769 ms against 0.2753-0.2763 -- no gain, removed: profiles/r06_describe.md.)
  int m10v = 0, m01v = 0;
  {
    uint4 vn[2];
    issue_ori(k_first, vn);
    for (int k = k_first; k < OD_K; k = next_valid(k)) {
      const int cx = __builtin_amdgcn_readlane(i_cx, k);
      const int m = (cx - 15) & 15;
#pragma unroll
      for (int j = 0; j < 2; j++) {
        const int i = lane + WAVE * j;
        const int r = i / 3, c = i - r * 3;
        if (i < 31 * 3) patch_store16<ORI_PITCH>(ori, r, c, vn[j]);
      }
      FC_T(1);   // wait for the raw patch + LDS store
      const int kn = next_valid(k);
      if (kn < OD_K) issue_ori(kn, vn);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      int A = 0, B = 0;
      const uint32_t sh = (uint32_t)(m & 3);
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int it = lane + WAVE * j;
        const int r = it >> 3, c = it & 7;
        const uint32_t* q = reinterpret_cast<const uint32_t*>(ori + (r < 31 ? r : 30) * ORI_PITCH + ((m + 4 * c) & ~3));
        // the four pixels u = -15 + 4c .. -12 + 4c of row v = r - 15, as I - 128
        const int px = (int)(__builtin_amdgcn_alignbyte(q[1], q[0], sh) ^ 0x80808080u);
        A = __builtin_amdgcn_sdot4(px, (int)wu[j], A, false);
        B = __builtin_amdgcn_sdot4(px, (int)wv[j], B, false);
      }
      const int At = __builtin_amdgcn_readlane(wave_incl_scan(A), 63), Bt = __builtin_amdgcn_readlane(wave_incl_scan(B), 63);
      if (lane == k) { m10v = At; m01v = Bt; }
      __builtin_amdgcn_wave_barrier();   // every lane has read the patch before the next keypoint overwrites it
      FC_T(2);   // moments
    }
  }
  // ---- phase 2: lane k computes keypoint k's angle and rotation; the first blurred patch is already on its way
  uint4 wn[3];
  issue_dsc(k_first, wn);
  const float angle_v = fast_atan2_deg((float)m01v, (float)m10v);
  const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
  float a_v, b_v;
  glibc_sincosf(angle_v * factorPI, &b_v, &a_v);
  FC_T(3);   // angle, sin / cos
  // ---- phase 3: steered BRIEF on the blurred level
  for (int k = k_first; k < OD_K; k = next_valid(k)) {
    const int level = __builtin_amdgcn_readlane(i_level, k), cx = __builtin_amdgcn_readlane(i_cx, k), cy = __builtin_amdgcn_readlane(i_cy, k);
    const int out = __builtin_amdgcn_readlane(i_out, k), score = __builtin_amdgcn_readlane(i_score, k);
    const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a_v), k)), b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(b_v), k)),
                angle = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(angle_v), k));
    const int ax_d = (cx - 18) & ~15;
    {
#if OD_DSC_NARROW
      const bool wide = ((cx - 18) & 15) > 11;
      const int n_pieces = wide ? 37 * 4 : 37 * 3;
#else
      const bool wide = true;
      const int n_pieces = 37 * 4;
      const int r3[3] = {0, 0, 0}, c3[3] = {0, 0, 0};
#endif
#pragma unroll
      for (int j = 0; j < 3; j++) {
        const int i = lane + WAVE * j;
        if (i < n_pieces) patch_store16<DSC_PITCH>(dsc, wide ? (i >> 2) : r3[j], wide ? (i & 3) : c3[j], wn[j]);
      }
    }
    FC_T(4);   // wait for the blurred patch + LDS store
    const int kn = next_valid(k);
    if (kn < OD_K) issue_dsc(kn, wn);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // cvRound of the rotated coordinates (L/src/ORBextractor.cc:119-121) by the magic-number addition: v + 1.5 * 2^23 rounds |v| < 2^22
    // to nearest-even in the mantissa's low bits -- one v_add_f32 where v_rndne_f32 + v_cvt_i32_f32 were two.  The integers are never
    // separated from the magic word: in 32-bit wrap-around arithmetic (M + ry) * PITCH + (M + rx) is the patch index plus a constant
    // that goes into the wave-uniform base
    const float MAGIC = 12582912.0f;
    const uint32_t MB = 0x4B400000u;   // its bit pattern
    const uint32_t bc0 = (uint32_t)(18 * DSC_PITCH + (cx - ax_d)) - (MB * (uint32_t)DSC_PITCH + MB);
    uint8_t* dout = P.out_desc + ((size_t)img * P.cap + out) * 32;
    // all eight samples requested before the first comparison, and the 32 descriptor bytes stored by lanes 0 .. 3 in one instruction:
    // with a ballot and lane 0's store behind every pair of samples the compiler kept the four rounds apart -- four dependent LDS
    // round trips (with their bank conflicts: the rotated positions are pseudo-random bytes of the window) and four stores per keypoint
    int t0[4], t1[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const float x0 = pk[r].x, y0 = pk[r].y, x1 = pk[r].z, y1 = pk[r].w;
      const uint32_t ry0 = __float_as_uint((x0 * b + y0 * a) + MAGIC), rx0 = __float_as_uint((x0 * a - y0 * b) + MAGIC);
      const uint32_t ry1 = __float_as_uint((x1 * b + y1 * a) + MAGIC), rx1 = __float_as_uint((x1 * a - y1 * b) + MAGIC);
      t0[r] = dsc[bc0 + ry0 * (uint32_t)DSC_PITCH + rx0];
      t1[r] = dsc[bc0 + ry1 * (uint32_t)DSC_PITCH + rx1];
    }
    {
      const unsigned long long b0 = __ballot(t0[0] < t1[0]), b1 = __ballot(t0[1] < t1[1]), b2 = __ballot(t0[2] < t1[2]), b3 = __ballot(t0[3] < t1[3]);
      const unsigned long long mine = lane == 0 ? b0 : lane == 1 ? b1 : lane == 2 ? b2 : b3;
      if (lane < 4) reinterpret_cast<unsigned long long*>(dout)[lane] = mine;
    }
    if (lane < 7) {
      float fx = (float)cx, fy = (float)cy;
      if (level != 0) {
        fx *= P.scale[level];
        fy *= P.scale[level];
      }
      uint32_t wvv;
      switch (lane) {
        case 0: wvv = __float_as_uint(fx); break;
        case 1: wvv = __float_as_uint(fy); break;
        case 2: wvv = __float_as_uint(P.kp_size[level]); break;
        case 3: wvv = __float_as_uint(angle); break;
        case 4: wvv = __float_as_uint((float)score); break;
        case 5: wvv = (uint32_t)level; break;
        default: wvv = 0xFFFFFFFFu; break;
      }
      reinterpret_cast<uint32_t*>(P.out_kps + (size_t)img * P.cap + out)[lane] = wvv;
    }
    __builtin_amdgcn_wave_barrier();
    FC_T(5);   // BRIEF + stores
  }
#endif
#if FC_TIMING
  if (lane == 0) {
    unsigned long long* pr = g_od_prof + (size_t)((((unsigned)(bx * 4 + wv_id) + 977u * (unsigned)img) * 2654435761u) >> 20) * 8;
    atomicAdd(&pr[0], (unsigned long long)tacc0); atomicAdd(&pr[1], (unsigned long long)tacc1);
    atomicAdd(&pr[2], (unsigned long long)tacc2); atomicAdd(&pr[3], (unsigned long long)tacc3);
    atomicAdd(&pr[4], (unsigned long long)tacc4); atomicAdd(&pr[5], (unsigned long long)tacc5);
    atomicAdd(&pr[6], 1ull);
    atomicAdd(&pr[7], (unsigned long long)__popc(valid_mask));
  }
#endif
}

// ------------------------------------------------------------------------------------------------ launchers
void orbfe_launch_copy0(const uint8_t* src, int sstride, size_t simg, uint8_t* dst, int dpitch, size_t dimg, int w,
                        int h, int n_images, int tiled, hipStream_t s) {
  dim3 block(256), grid((((w + 15) / 16) * h + 255) / 256, 1, n_images);
  hipLaunchKernelGGL(copy_level0_kernel, grid, block, 0, s, src, sstride, (unsigned long long)simg, dst, dpitch,
                     (unsigned long long)dimg, w, h, tiled);
}

void orbfe_launch_resize(const uint8_t* src, int spitch, size_t simg, uint8_t* dst, int dpitch, size_t dimg, int dw,
                         int dh, const ResizeTap* xt, const ResizeTap* yt, int n_images, int mode, hipStream_t s) {
  dim3 block(64, 4), grid((dw + 255) / 256, (dh + 3) / 4, n_images);
  if (mode >= 2) {   // scale <= 2 and taps within 8 source bytes: persistent workgroups, perm + dot2 interpolation
#ifndef RS_WGS_NARROW
#define RS_WGS_NARROW 7   // 6: 0.262, 7: 0.254, 8: 0.268 ms per pyramid of 256 images
#endif
    const int wgs = mode == 3 ? RS_WGS_NARROW : 6;   // workgroups per CU (mode 3: every window fits 21 chunks, 14 KB of LDS)
    const int tiles_x = (dw + 255) / 256, tiles_y = (dh + 31) / 32;
    const int n_tiles = tiles_x * tiles_y * n_images;
    const int nb = n_tiles < 256 * wgs ? n_tiles : 256 * wgs;
    if (mode == 3)
      hipLaunchKernelGGL((pyr_resize_dot_kernel<32, 42, 336>), dim3(nb), block, 0, s, src, spitch, (unsigned long long)simg, dst,
                         dpitch, (unsigned long long)dimg, dw, dh, xt, yt, tiles_x, tiles_y, n_tiles);
    else
      hipLaunchKernelGGL((pyr_resize_dot_kernel<32, 42, RS_COLS>), dim3(nb), block, 0, s, src, spitch, (unsigned long long)simg, dst,
                         dpitch, (unsigned long long)dimg, dw, dh, xt, yt, tiles_x, tiles_y, n_tiles);
  }
  else if (mode >= 1)
    hipLaunchKernelGGL(pyr_resize_lds16_kernel, dim3((dw + 255) / 256, (dh + 15) / 16, n_images), block, 0, s, src, spitch,
                       (unsigned long long)simg, dst, dpitch, (unsigned long long)dimg, dw, dh, xt, yt);
  else  // source window of a tile exceeds the staged size (scale factor > 2): direct byte gathers
    hipLaunchKernelGGL(pyr_resize_kernel, grid, block, 0, s, src, spitch, (unsigned long long)simg, dst, dpitch,
                       (unsigned long long)dimg, dw, dh, xt, yt);
}

void orbfe_launch_fast_cells(const PyrView& pyr, const CellDesc* cells, const FastGroup* groups, int n_groups, int total_cells,
                             int cell_rows, int cell_span, int sc_max, int bits_max, int32_t* cell_cnt, uint32_t* slots,
                             unsigned long long slots_per_image, int ini_th, int min_th, int n_images, hipStream_t s) {
  if (n_groups == 0) return;
  const int run_shift = -2;   // whole images per XCD (-1: plain blockIdx order, k >= 0: runs of 2^k workgroups per XCD)
  // wave per run of cells: per-wave LDS slice = byte tile + score plane + bitmap + list
  const int pb = cell_span <= 64 ? 64 : 96;
  const int sc_bytes = (sc_max + 15) & ~15;
  const int nbw = bits_max <= 64 * 32 ? 64 : 128;   // bitmap words (lane l owns words l and l + 64)
  const size_t lds = (size_t)FC_WG_WAVES * fc_wave_lds(cell_rows, pb, sc_bytes, nbw);
  const bool small = cell_rows <= 48;   // a cell loads in three rounds of sixteen rows
  dim3 grid4((n_groups + FC_WG_WAVES - 1) / FC_WG_WAVES, n_images);
#define FC_LAUNCH(PBV, NLDV)                                                                                                   \
  hipLaunchKernelGGL((fast_cells_kernel<PBV, NLDV>), grid4, dim3(64 * FC_WG_WAVES), lds, s, pyr, cells, groups, n_groups, total_cells,       \
                     cell_rows, sc_bytes, nbw, cell_cnt, slots, slots_per_image, ini_th, min_th, run_shift)
  if (lds > 48 * 1024) {   // never for the configured datasets (22 KB at KITTI); the largest cell (66 x 66) needs 46 KB
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fast_cells_kernel<96, 7>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fast_cells_kernel<64, 7>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  if (pb == 64 && small) FC_LAUNCH(64, 3);
  else if (pb == 64) FC_LAUNCH(64, 7);
  else FC_LAUNCH(96, 7);
#undef FC_LAUNCH
}

void orbfe_launch_octree(const OctParams& p, int n_images, size_t lds_bytes, hipStream_t s) {
  dim3 block(ORBFE_OCT_THREADS), grid(n_images, p.n_levels);
  hipLaunchKernelGGL(octree_select_kernel, grid, block, lds_bytes, s, p);
}

// Which blur runs: measured on 256 KITTI images (profiles/r04_blur_mfma.md, DESIGN lesson 31) the matrix-core kernel does the
// arithmetic of the LDS kernel in a third of the vector instructions and is bit-exact, but the stage moves 0.74 GB per launch
// and both kernels take 0.29-0.32 ms for it; the LDS kernel is a few per cent ahead and is what a handle uses unless
// orbfe_debug_blur_kernel() selects the other one (the parity test of the matrix-core kernel does; -DORBFE_BLUR_MFMA=1 makes it
// the default of an A/B build).
#ifndef ORBFE_BLUR_MFMA
#define ORBFE_BLUR_MFMA 0
#endif
void orbfe_launch_blur(const PyrView& src, const PyrView& dst, const BlurTile* tiles, int n_tiles, const BlurMfmaParams& mf,
                       int n_images, hipStream_t s) {
  if ((ORBFE_BLUR_MFMA || mf.use) && mf.n_strips > 0) {
    hipLaunchKernelGGL(gauss_blur7_mfma_kernel, dim3((mf.n_strips + 3) / 4, n_images), dim3(256), 0, s, src, dst, mf);
    return;
  }
  orbfe_launch_blur_level(src, dst, tiles, n_tiles, nullptr, n_images, s);
}

void orbfe_launch_blur_level(const PyrView& src, const PyrView& dst, const BlurTile* tiles, int n_tiles, const LevelResize* rz,
                             int n_images, hipStream_t s) {
  if (n_tiles <= 0) return;
  dim3 grid(n_tiles, n_images);
#ifndef BT_LDS_PAD
#define BT_LDS_PAD 0   // unused dynamic LDS: an occupancy limiter for experiments
#endif
  // a batch wants many tiles in flight per CU (two waves per tile); a one- or two-image call has fewer tiles than the chip has CUs
  // and wants each tile done quickly (four waves per tile: 0.292 -> 0.27 ms for a stereo pair's two extractions)
  const bool few = n_images <= 8;
  if (rz) {
    if (few) hipLaunchKernelGGL((blur_level_kernel<true, 256>), grid, dim3(256), BT_LDS_PAD, s, src, dst, tiles, n_tiles, *rz);
    else hipLaunchKernelGGL((blur_level_kernel<true, BT_THREADS>), grid, dim3(BT_THREADS), BT_LDS_PAD, s, src, dst, tiles, n_tiles, *rz);
  } else {
    if (few) hipLaunchKernelGGL((blur_level_kernel<false, 256>), grid, dim3(256), BT_LDS_PAD, s, src, dst, tiles, n_tiles, LevelResize{});
    else hipLaunchKernelGGL((blur_level_kernel<false, BT_THREADS>), grid, dim3(BT_THREADS), BT_LDS_PAD, s, src, dst, tiles, n_tiles, LevelResize{});
  }
}

void orbfe_launch_describe(const DescribeParams& p, int n_images, hipStream_t s) {
  DescribeParams pp = p;
  pp.xcd_images = 1;
  const int per_block = 4 * OD_K;
  dim3 block(256), grid((p.kp_per_image + per_block - 1) / per_block > 0 ? (p.kp_per_image + per_block - 1) / per_block : 1, n_images);
  // 63 VGPRs would let eight waves per SIMD run; the gather of 2 000 patches per image is L2-miss bound and an eighth wave
  // measured slower (0.352 against 0.341 ms per 256 images): unused dynamic LDS keeps a CU at seven workgroups
#ifndef OD_LDS_PAD
// dynamic LDS that brings a workgroup to (160 KB / (OD_WGS + 1) rounded up to 512 B) + 512: OD_WGS workgroups fit a CU, OD_WGS + 1 do not
#define OD_LDS_WG ((((163840 / (OD_WGS + 1)) + 511) & ~511) + 512)
#define OD_LDS_PAD (4 * PATCH_BYTES < OD_LDS_WG ? OD_LDS_WG - 4 * PATCH_BYTES : 0)
#endif
  hipLaunchKernelGGL(orient_describe8_kernel, grid, block, OD_LDS_PAD, s, pp);
}

int orbfe_upload_pattern_floats() {
  static const int8_t pat[1024] = {ORB_PATTERN_INT8_1024};
  float f[1024];
  for (int i = 0; i < 1024; i++) f[i] = (float)pat[i];
  hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(g_pattern_f), f, sizeof(f));
  if (e != hipSuccess) return (int)e;
  // weights of the moment sums: item = row * 8 + group, row v = r - 15, columns u = -15 + 4 * group + t
  static const int umax[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};
  uint32_t wtab[256 * 2];
  memset(wtab, 0, sizeof(wtab));
  for (int it = 0; it < 248; it++) {
    const int v = (it >> 3) - 15;
    for (int t = 0; t < 4; t++) {
      const int u = -15 + 4 * (it & 7) + t;
      const bool inside = u <= 15 && abs(u) <= umax[abs(v)];
      if (!inside) continue;
      wtab[it * 2] |= (uint32_t)(uint8_t)(int8_t)u << (8 * t);
      wtab[it * 2 + 1] |= (uint32_t)(uint8_t)(int8_t)v << (8 * t);
    }
  }
  e = hipMemcpyToSymbol(HIP_SYMBOL(g_ic_w), wtab, sizeof(wtab));
  if (e != hipSuccess) return (int)e;
  // band matrices of the tile kernel's matrix-core blur (blur_level_kernel, BT_MFMA): lane (q, n) of a matrix holds 16 bytes.
  //   horizontal (one for every column group g): byte j <-> LDS column 16g + 16q + j, output = LDS column BT_COL0 + 4 + 16g + n
  //   vertical, block b:  byte j = 4 ww + i <-> window row 16 ww + 4q + i, output = window row 3 + 16b + n
  static const int taps[7] = {18, 34, 48, 56, 48, 34, 18};
  static int8_t bt[5 * 1024];
  for (int m = 0; m < 5; m++)
    for (int lane = 0; lane < 64; lane++)
      for (int j = 0; j < 16; j++) {
        const int q = lane >> 4, n = lane & 15;
        const int d = m == 0 ? (16 * q + j) - (4 + BT_COL0 + n) : (16 * (j >> 2) + 4 * q + (j & 3)) - (3 + 16 * (m - 1) + n);
        bt[(m * 64 + lane) * 16 + j] = (int8_t)(d >= -3 && d <= 3 ? taps[d + 3] : 0);
      }
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_blur_tile_tab), bt, sizeof(bt));
}

int orbfe_set_octree_lds(size_t lds_bytes) {
  return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(octree_select_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
}
