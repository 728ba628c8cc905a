// match_kernels.hip -- HIP kernels of the ORB matcher path for gfx950 (MI355X, wave64).
//
// Kernel              replaces (reference file:line, L/ = Source/Libraries/ORB_SLAM2/)
// hamming_matrix      ORBmatcher::DescriptorDistance for all pairs            L/src/ORBmatcher.cc:1542-1556
// hamming_bf          best / second-best loops of SearchByBoW                 L/src/ORBmatcher.cc:201-222
// grid_build          Frame::AssignFeaturesToGrid + PosInGrid                 L/src/Frame.cc:250-263,399-410
// proj_candidates     Frame::GetFeaturesInArea + DescriptorDistance           L/src/Frame.cc:341-397
// proj_resolve        order-dependent assignment of SearchByProjection        L/src/ORBmatcher.cc:45-128,1247-1383
// stereo_match        Frame::ComputeStereoMatches (Hamming + 11x11 SAD)        L/src/Frame.cc:477-632
// stereo_median       median-based outlier cut                                 L/src/Frame.cc:634-645
//
// Compiled with -ffp-contract=off: every float expression is evaluated un-fused, as the reference does.
#include <string.h>
#include <atomic>
#include <mutex>

#include "match_internal.h"

#define WAVE 64

__device__ __forceinline__ int hamming256(const uint4 a0, const uint4 a1, const uint4 b0, const uint4 b1) {
  return __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) +
         __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
}
__device__ __forceinline__ void load_desc(const uint8_t* p, uint4& d0, uint4& d1) {
  const uint4* q = reinterpret_cast<const uint4*>(p);
  d0 = q[0];
  d1 = q[1];
}
// descriptors that are only 4-byte aligned (queries embedded in 68-byte records)
__device__ __forceinline__ void load_desc4(const uint8_t* p, uint4& d0, uint4& d1) {
  const uint32_t* q = reinterpret_cast<const uint32_t*>(p);
  d0 = make_uint4(q[0], q[1], q[2], q[3]);
  d1 = make_uint4(q[4], q[5], q[6], q[7]);
}

__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
  // DPP reduction (row shifts, then row broadcasts): lane 63 ends up with the minimum of all 64 lanes
#define ORBFE_MIN_STEP(ctrl, rmask) v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)0xffffffff, (int)v, ctrl, rmask, 0xf, false))
  ORBFE_MIN_STEP(0x111, 0xf); ORBFE_MIN_STEP(0x112, 0xf); ORBFE_MIN_STEP(0x114, 0xf); ORBFE_MIN_STEP(0x118, 0xf);
  ORBFE_MIN_STEP(0x142, 0xa); ORBFE_MIN_STEP(0x143, 0xc);
#undef ORBFE_MIN_STEP
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ int wave_incl_scan_i(int v) {
  // DPP row shifts + row broadcasts (gfx9): six VALU adds, no LDS crossbar round trips (ds_bpermute) as with __shfl_up
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);  // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);  // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);  // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);  // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1, 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2, 3
  return v;
}
__device__ __forceinline__ int wave_sum_i32(int v) { return __builtin_amdgcn_readlane(wave_incl_scan_i(v), 63); }

// ------------------------------------------------------------------------------------------------ all pairs
// Thread = one B column (descriptor in registers); a block stages 64 A rows in LDS; stores are coalesced
// along j.  grid = (ceil(nB/256), ceil(nA/64)).
__global__ __launch_bounds__(256) void hamming_matrix_kernel(const uint8_t* __restrict__ A, int nA,
                                                              const uint8_t* __restrict__ B, int nB,
                                                              uint16_t* __restrict__ out) {
  __shared__ uint4 sa[64 * 2];
  const int j = blockIdx.x * 256 + threadIdx.x;
  const int i0 = blockIdx.y * 64;
  if (threadIdx.x < 128) {
    const int r = i0 + (threadIdx.x >> 1);
    sa[threadIdx.x] = r < nA ? reinterpret_cast<const uint4*>(A)[(size_t)r * 2 + (threadIdx.x & 1)] : make_uint4(0, 0, 0, 0);
  }
  __syncthreads();
  if (j >= nB) return;
  uint4 b0, b1;
  load_desc(B + (size_t)j * 32, b0, b1);
  const int rows = min(64, nA - i0);
  for (int r = 0; r < rows; r++) out[(size_t)(i0 + r) * nB + j] = (uint16_t)hamming256(sa[2 * r], sa[2 * r + 1], b0, b1);
}

// ------------------------------------------------------------------------------------------------ brute force
// Thread = one A row, block = 256 rows of one set; every thread keeps (best (dist, j) key, second dist) of its own row, so there
// is no cross-lane reduction and no merge: the lexicographic (dist, j) minimum IS the reference's strict-< first-wins scan over
// vIndicesF (L/src/ORBmatcher.cc:205-222), the second-smallest distance its bestDist2, whatever order the j arrive in.
//   plain     B streams through LDS in tiles of 256 descriptors; a wave reads one descriptor per step (same address on all lanes:
//             an LDS broadcast) -- 8 v_xor + 8 v_bcnt (popcount with accumulate) per 64 pairs
//   grouped   (vocabulary node ids, the SearchByBoW use): comparing a row with every j and discarding 99 % of them by the group
//             test costs a full scan per row.  Instead the block buckets the B indices by a 10-bit hash of their group id in LDS
//             (counting sort: histogram, scan, scatter -- 4 KB of ids per 1000 descriptors) and every row walks only the bucket
//             of its own group (a dozen candidates at 1000 descriptors in 100 groups), testing the exact id.  B sets beyond
//             HBF_MAX_SORT fall back to the plain scan with the group test.
#define HBF_HASH_BITS 10
#define HBF_BUCKETS (1 << HBF_HASH_BITS)
#define HBF_MAX_SORT 8192
__device__ __forceinline__ unsigned hbf_hash(int g) { return ((unsigned)g * 2654435761u) >> (32 - HBF_HASH_BITS); }
__device__ __forceinline__ void hbf_update(unsigned& best, int& second, int d, int j) {
  const unsigned key = ((unsigned)d << 16) | (unsigned)j;
  if (key < best) { second = min(second, (int)(best >> 16)); best = key; }
  else second = min(second, d);
}
__global__ __launch_bounds__(256) void hamming_bf_kernel(HammingBfParams P) {
  __shared__ __attribute__((aligned(16))) uint4 sb[256 * 2 + 1];   // plain: a tile of B; grouped: start[HBF_BUCKETS + 1] + cur[HBF_BUCKETS] (8196 bytes)
  __shared__ int sg[256];
  __shared__ uint8_t sm[256];
  __shared__ uint16_t order[HBF_MAX_SORT];
  const int set = blockIdx.y;
  const int nA = P.nA[set], nB = P.nB[set];
  if ((int)blockIdx.x * 256 >= nA) return;
  const int a = blockIdx.x * 256 + threadIdx.x;
  const uint8_t* A = P.A + (size_t)set * P.strideA * 32;
  const uint8_t* B = P.B + (size_t)set * P.strideB * 32;
  const int32_t* gB = P.groupB ? P.groupB + (size_t)set * P.strideB : nullptr;
  const uint8_t* mB = P.maskB ? P.maskB + (size_t)set * P.strideB : nullptr;
  uint4 a0 = make_uint4(0, 0, 0, 0), a1 = a0;
  int ga = 0;
  if (a < nA) {
    load_desc(A + (size_t)a * 32, a0, a1);
    if (P.groupA) ga = P.groupA[(size_t)set * P.strideA + a];
  }
  unsigned best = 0xFFFFFFFFu;   // (256 << 16) would do; all ones also marks "none"
  int second = 256;
  if (gB && nB <= HBF_MAX_SORT) {
    int* start = reinterpret_cast<int*>(sb);          // [HBF_BUCKETS + 1]
    int* cur = start + HBF_BUCKETS + 1;               // [HBF_BUCKETS]
    for (int i = threadIdx.x; i < HBF_BUCKETS; i += 256) cur[i] = 0;
    __syncthreads();
    for (int j = threadIdx.x; j < nB; j += 256)
      if (!(mB && mB[j])) atomicAdd(&cur[hbf_hash(gB[j])], 1);
    __syncthreads();
    {   // exclusive scan of 1024 counts: four per thread, wave scan, wave totals through LDS
      const int t = threadIdx.x;
      const int c0 = cur[4 * t], c1 = cur[4 * t + 1], c2 = cur[4 * t + 2], c3 = cur[4 * t + 3];
      const int incl = wave_incl_scan_i(c0 + c1 + c2 + c3);
      if ((t & 63) == 63) sg[t >> 6] = incl;
      __syncthreads();
      int base = incl - (c0 + c1 + c2 + c3);
      for (int w = 0; w < (t >> 6); w++) base += sg[w];
      start[4 * t] = base; start[4 * t + 1] = base + c0; start[4 * t + 2] = base + c0 + c1; start[4 * t + 3] = base + c0 + c1 + c2;
      if (t == 255) start[HBF_BUCKETS] = base + c0 + c1 + c2 + c3;
      __syncthreads();
      cur[4 * t] = 0; cur[4 * t + 1] = 0; cur[4 * t + 2] = 0; cur[4 * t + 3] = 0;
    }
    __syncthreads();
    for (int j = threadIdx.x; j < nB; j += 256)
      if (!(mB && mB[j])) {
        const unsigned h = hbf_hash(gB[j]);
        order[start[h] + atomicAdd(&cur[h], 1)] = (uint16_t)j;
      }
    __syncthreads();
    if (a < nA) {
      const unsigned h = hbf_hash(ga);
      for (int e = start[h], e1 = start[h + 1]; e < e1; e++) {
        const int j = order[e];
        if (gB[j] != ga) continue;   // another group with the same hash
        uint4 b0, b1;
        load_desc(B + (size_t)j * 32, b0, b1);
        hbf_update(best, second, hamming256(a0, a1, b0, b1), j);
      }
    }
  } else {
    for (int t0 = 0; t0 < nB; t0 += 256) {
      __syncthreads();
      {
        const int j = t0 + threadIdx.x;
        if (j < nB) {
          const uint4* q = reinterpret_cast<const uint4*>(B + (size_t)j * 32);
          sb[2 * threadIdx.x] = q[0];
          sb[2 * threadIdx.x + 1] = q[1];
          sg[threadIdx.x] = gB ? gB[j] : 0;
          sm[threadIdx.x] = mB ? mB[j] : 0;
        }
      }
      __syncthreads();
      const int cnt = min(256, nB - t0);
      for (int jj = 0; jj < cnt; jj++) {   // jj is wave-uniform: sb / sg / sm reads are broadcasts
        if (sm[jj]) continue;
        if (gB && sg[jj] != ga) continue;
        hbf_update(best, second, hamming256(a0, a1, sb[2 * jj], sb[2 * jj + 1]), t0 + jj);
      }
    }
  }
  if (a < nA) {
    orbfe_bf_match m;
    if (best == 0xFFFFFFFFu) { m.best_idx = -1; m.best_dist = 256; m.second_dist = 256; }
    else { m.best_idx = (int)(best & 0xffff); m.best_dist = (int)(best >> 16); m.second_dist = second; }
    P.out[(size_t)set * P.strideA + a] = m;
  }
}

// ------------------------------------------------------------------------------------------------ grid
// One block per frame: 64x48 grid in CSR form (cell = ix*48 + iy), lists in ascending keypoint index.
#define GB_THREADS 1024   // one block per frame and one frame per CU: the block's own parallelism is all there is to hide latency with
#define GB_LDS_CAP 20000  // frames up to this capacity keep the per-keypoint state in LDS (6 bytes per keypoint)
// LDS-resident form: two global round trips (keypoints in, records out) instead of nine.  A keypoint's place inside its cell is
// its RANK among the cell's indices (cells hold a handful of keypoints), computed by the thread that owns the keypoint -- no sort,
// and the 16-byte record is written from the owner's registers instead of through the index -> keypoint -> mvuRight chain.
__global__ __launch_bounds__(GB_THREADS) void grid_build_kernel(FrameBatch F) {
  __shared__ int cnt[GRID_CELLS];
  __shared__ int start[GRID_CELLS + 1];
  __shared__ int tmp[GB_THREADS / 64];
  extern __shared__ __attribute__((aligned(16))) uint8_t gdyn[];
  static_assert(GRID_CELLS % GB_THREADS == 0 && GRID_CELLS <= 4096, "cells per thread; the cell id travels in 12 bits");
  constexpr int CPT = GRID_CELLS / GB_THREADS;   // consecutive cells per thread in the scan
  uint32_t* cellslot = reinterpret_cast<uint32_t*>(gdyn);            // per keypoint: cell | arrival slot << 12 (~0u: outside the grid)
  uint16_t* list = reinterpret_cast<uint16_t*>(cellslot + F.cap);    // per CSR entry: keypoint index, arrival order inside a cell
  const int f = blockIdx.x, tid = threadIdx.x;
  const int n = F.n[f];
  const orbfe_keypoint* keys = F.keys + (size_t)f * F.cap;
  const float* ur = F.u_right ? F.u_right + (size_t)f * F.cap : nullptr;
  int32_t* cs = F.cell_start + (size_t)f * (GRID_CELLS + 1);
  int32_t* ci = F.cell_idx + (size_t)f * F.cap;
  uint4* cr = reinterpret_cast<uint4*>(F.cell_rec) + (size_t)f * F.cap;
  for (int c = tid; c < GRID_CELLS; c += GB_THREADS) cnt[c] = 0;
  __syncthreads();
  for (int i0 = tid; i0 < n; i0 += 2 * GB_THREADS) {   // two keypoints per thread and round trip
    const int i1 = i0 + GB_THREADS;
    const float x0 = keys[i0].x, y0 = keys[i0].y;
    float x1 = 0.f, y1 = 0.f;
    if (i1 < n) { x1 = keys[i1].x; y1 = keys[i1].y; }
#pragma unroll
    for (int k = 0; k < 2; k++) {
      const int i = k ? i1 : i0;
      if (i >= n) break;
      const int px = (int)roundf(((k ? x1 : x0) - F.min_x) * F.gw_inv);
      const int py = (int)roundf(((k ? y1 : y0) - F.min_y) * F.gh_inv);
      uint32_t v = ~0u;
      if (px >= 0 && px < ORBFE_GRID_COLS && py >= 0 && py < ORBFE_GRID_ROWS) {
        const int c = px * ORBFE_GRID_ROWS + py;
        v = (uint32_t)c | ((uint32_t)atomicAdd(&cnt[c], 1) << 12);
      }
      cellslot[i] = v;
    }
  }
  __syncthreads();
  // exclusive scan: CPT consecutive cells per thread, DPP wave scan, wave totals through LDS
  {
    int local = 0;
#pragma unroll
    for (int k = 0; k < CPT; k++) local += cnt[tid * CPT + k];
    const int v = wave_incl_scan_i(local);
    if ((tid & 63) == 63) tmp[tid >> 6] = v;
    __syncthreads();
    int off = 0;
    for (int w = 0; w < (tid >> 6); w++) off += tmp[w];
    int run = off + v - local;
#pragma unroll
    for (int k = 0; k < CPT; k++) {
      start[tid * CPT + k] = run;
      run += cnt[tid * CPT + k];
    }
    if (tid == GB_THREADS - 1) start[GRID_CELLS] = run;
  }
  __syncthreads();
  for (int c = tid; c <= GRID_CELLS; c += GB_THREADS) cs[c] = start[c];
  for (int i = tid; i < n; i += GB_THREADS) {
    const uint32_t v = cellslot[i];
    if (v != ~0u) list[start[v & 0xfffu] + (int)(v >> 12)] = (uint16_t)i;
  }
  __syncthreads();
  for (int i0 = tid; i0 < n; i0 += 2 * GB_THREADS) {
    const int i1 = i0 + GB_THREADS;
    const orbfe_keypoint k0 = keys[i0];
    const float u0 = ur ? ur[i0] : -1.0f;
    orbfe_keypoint k1 = k0;
    float u1 = -1.0f;
    if (i1 < n) { k1 = keys[i1]; u1 = ur ? ur[i1] : -1.0f; }
#pragma unroll
    for (int k = 0; k < 2; k++) {
      const int i = k ? i1 : i0;
      if (i >= n) break;
      const uint32_t v = cellslot[i];
      if (v == ~0u) continue;
      const int c = (int)(v & 0xfffu);
      const int s0 = start[c], s1 = start[c + 1];
      int pos = s0;
      for (int j = s0; j < s1; j++) pos += (int)list[j] < i;   // insertion order of the reference = ascending index
      const orbfe_keypoint& kp = k ? k1 : k0;
      ci[pos] = i;
      cr[pos] = make_uint4((uint32_t)i | ((uint32_t)kp.octave << 24), (uint32_t)__float_as_int(kp.x), (uint32_t)__float_as_int(kp.y),
                           (uint32_t)__float_as_int(k ? u1 : u0));
    }
  }
}

// the same for frames beyond GB_LDS_CAP keypoints: index lists sorted in global memory
__global__ __launch_bounds__(GB_THREADS) void grid_build_global_kernel(FrameBatch F) {
  __shared__ int cnt[GRID_CELLS];
  __shared__ int start[GRID_CELLS + 1];
  __shared__ int tmp[GB_THREADS / 64];
  static_assert(GRID_CELLS % GB_THREADS == 0, "cells per thread");
  constexpr int CPT = GRID_CELLS / GB_THREADS;   // consecutive cells per thread in the scan
  const int f = blockIdx.x, tid = threadIdx.x;
  const int n = F.n[f];
  const orbfe_keypoint* keys = F.keys + (size_t)f * F.cap;
  int32_t* cs = F.cell_start + (size_t)f * (GRID_CELLS + 1);
  int32_t* ci = F.cell_idx + (size_t)f * F.cap;
  for (int c = tid; c < GRID_CELLS; c += GB_THREADS) cnt[c] = 0;
  __syncthreads();
  for (int i = tid; i < n; i += GB_THREADS) {
    const int px = (int)roundf((keys[i].x - F.min_x) * F.gw_inv);
    const int py = (int)roundf((keys[i].y - F.min_y) * F.gh_inv);
    if (px >= 0 && px < ORBFE_GRID_COLS && py >= 0 && py < ORBFE_GRID_ROWS) atomicAdd(&cnt[px * ORBFE_GRID_ROWS + py], 1);
  }
  __syncthreads();
  // exclusive scan: CPT consecutive cells per thread, DPP wave scan, wave totals through LDS
  {
    int local = 0;
#pragma unroll
    for (int k = 0; k < CPT; k++) local += cnt[tid * CPT + k];
    const int v = wave_incl_scan_i(local);
    if ((tid & 63) == 63) tmp[tid >> 6] = v;
    __syncthreads();
    int off = 0;
    for (int w = 0; w < (tid >> 6); w++) off += tmp[w];
    int run = off + v - local;
#pragma unroll
    for (int k = 0; k < CPT; k++) {
      start[tid * CPT + k] = run;
      run += cnt[tid * CPT + k];
    }
    if (tid == GB_THREADS - 1) start[GRID_CELLS] = run;
  }
  __syncthreads();
  for (int c = tid; c <= GRID_CELLS; c += GB_THREADS) cs[c] = start[c];
  for (int c = tid; c < GRID_CELLS; c += GB_THREADS) cnt[c] = 0;
  __syncthreads();
  for (int i = tid; i < n; i += GB_THREADS) {
    const int px = (int)roundf((keys[i].x - F.min_x) * F.gw_inv);
    const int py = (int)roundf((keys[i].y - F.min_y) * F.gh_inv);
    if (px >= 0 && px < ORBFE_GRID_COLS && py >= 0 && py < ORBFE_GRID_ROWS) {
      const int c = px * ORBFE_GRID_ROWS + py;
      ci[start[c] + atomicAdd(&cnt[c], 1)] = i;
    }
  }
  __syncthreads();
  // insertion order of the reference = ascending index: sort each (tiny) cell list
  for (int c = tid; c < GRID_CELLS; c += GB_THREADS) {
    const int s = start[c], e = start[c + 1];
    for (int i = s + 1; i < e; i++) {
      const int v = ci[i];
      int j = i - 1;
      while (j >= s && ci[j] > v) { ci[j + 1] = ci[j]; j--; }
      ci[j + 1] = v;
    }
  }
  __syncthreads();
  // everything a window walk tests per entry, in CSR order: one 16-byte record instead of the index -> keypoint -> mvuRight chain
  const float* ur = F.u_right ? F.u_right + (size_t)f * F.cap : nullptr;
  uint4* cr = reinterpret_cast<uint4*>(F.cell_rec) + (size_t)f * F.cap;
  const int total = start[GRID_CELLS];
  for (int e = tid; e < total; e += GB_THREADS) {
    const int i = ci[e];
    const orbfe_keypoint kp = keys[i];
    cr[e] = make_uint4((uint32_t)i | ((uint32_t)kp.octave << 24), (uint32_t)__float_as_int(kp.x), (uint32_t)__float_as_int(kp.y),
                       (uint32_t)__float_as_int(ur ? ur[i] : -1.0f));
  }
}

// Enumerates, in the reference's order, the keypoints GetFeaturesInArea returns for query q that also pass the
// matcher's stereo gate, and calls fn(rank, idx, dist) for each, wave-cooperatively: a chunk of up to 64
// consecutive grid entries is tested per step, survivors get consecutive ranks.  Returns the survivor count.
// GATE 0: the matcher's stereo gate |u_r - mvuRight| <= radius (SearchByProjection); 1: none (Fuse(Sim3), SearchBySim3);
// 2: Fuse's chi-square reprojection gate (L/src/ORBmatcher.cc:833-854) with inv_sigma2 = mvInvLevelSigma2
template <int GATE = 0, typename Fn>
__device__ __forceinline__ int enumerate_window(const FrameBatch& F, int f, const orbfe_query& q, const uint4 q0,
                                                const uint4 q1, Fn fn, const float* inv_sigma2 = nullptr) {
  const int lane = threadIdx.x & (WAVE - 1);
  const float x = q.u, y = q.v, r = q.radius;
  const int nMinCellX = max(0, (int)floorf((x - F.min_x - r) * F.gw_inv));
  if (nMinCellX >= ORBFE_GRID_COLS) return 0;
  const int nMaxCellX = min(ORBFE_GRID_COLS - 1, (int)ceilf((x - F.min_x + r) * F.gw_inv));
  if (nMaxCellX < 0) return 0;
  const int nMinCellY = max(0, (int)floorf((y - F.min_y - r) * F.gh_inv));
  if (nMinCellY >= ORBFE_GRID_ROWS) return 0;
  const int nMaxCellY = min(ORBFE_GRID_ROWS - 1, (int)ceilf((y - F.min_y + r) * F.gh_inv));
  if (nMaxCellY < 0) return 0;
  const bool bCheckLevels = (q.min_level > 0) || (q.max_level >= 0);
  const uint8_t* desc = F.desc + (size_t)f * F.cap * 32;
  const bool has_ur = F.u_right != nullptr;
  const int32_t* cs = F.cell_start + (size_t)f * (GRID_CELLS + 1);
  const uint4* cr = reinterpret_cast<const uint4*>(F.cell_rec) + (size_t)f * F.cap;
  // The window's cells of one grid column (ix, nMinCellY..nMaxCellY) are adjacent in CSR order.  Lane c fetches
  // column c's range; the ranges are concatenated (prefix sum) into one flat candidate sequence -- column-major, then
  // cell row, then ascending index: the reference's enumeration order -- that the wave consumes 64 entries at a time.
  const int ncol = __builtin_amdgcn_readfirstlane(nMaxCellX - nMinCellX + 1);  // <= 64; the query is wave-uniform
  int e0c = 0, cntc = 0;
  if (lane < ncol) {
    const int ix = nMinCellX + lane;
    e0c = cs[ix * ORBFE_GRID_ROWS + nMinCellY];
    cntc = cs[ix * ORBFE_GRID_ROWS + nMaxCellY + 1] - e0c;
  }
  const int inclc = wave_incl_scan_i(cntc);
  const int exclc = inclc - cntc;
  const int n_entries = __builtin_amdgcn_readlane(inclc, WAVE - 1);
  int rank0 = 0;
  for (int base = 0; base < n_entries; base += WAVE) {
    const int t = base + lane;
    int ent = -1;
    for (int c = 0; c < ncol; c++) {
      const int oc = __builtin_amdgcn_readlane(exclc, c), nc = __builtin_amdgcn_readlane(cntc, c), ec = __builtin_amdgcn_readlane(e0c, c);
      if (t >= oc && t < oc + nc) ent = ec + (t - oc);
    }
    bool ok = false;
    int idx = 0, oct = 0;
    if (ent >= 0) {
      const uint4 rec = cr[ent];
      idx = (int)(rec.x & 0xFFFFFFu);
      oct = (int)(rec.x >> 24);
      const float kx = __int_as_float((int)rec.y), ky = __int_as_float((int)rec.z), kur = __int_as_float((int)rec.w);
      ok = true;
      if (bCheckLevels) {
        if (oct < q.min_level) ok = false;
        if (q.max_level >= 0 && oct > q.max_level) ok = false;
      }
      const float distx = kx - x, disty = ky - y;
      if (!(fabsf(distx) < r && fabsf(disty) < r)) ok = false;
      if (GATE == 0) {
        if (ok && has_ur) {
          const float u = kur;
          if (u > 0) {
            const float er = fabsf(q.u_r - u);
            if (er > r) ok = false;
          }
        }
      } else if (GATE == 2) {
        if (ok) {
          const float kpr = kur;   // -1 without mvuRight
          const float ex = x - kx, ey = y - ky;
          const float invs = inv_sigma2[oct & (ORBFE_MAX_LEVELS - 1)];   // the mask only guards memory
          if (kpr >= 0) {
            const float er = q.u_r - kpr;
            const float e2 = ex * ex + ey * ey + er * er;
            if ((double)(e2 * invs) > 7.8) ok = false;
          } else {
            const float e2 = ex * ex + ey * ey;
            if ((double)(e2 * invs) > 5.99) ok = false;
          }
        }
      }
    }
    const unsigned long long m = __ballot(ok);
    if (ok) {
      uint4 d0, d1;
      load_desc(desc + (size_t)idx * 32, d0, d1);
      const int rank = rank0 + __popcll(m & ((1ull << lane) - 1ull));
      fn(rank, idx, hamming256(q0, q1, d0, d1), oct);
    }
    rank0 += __popcll(m);
  }
  return rank0;
}

// One wave per query, independent arg-min (no query blocks another): Fuse, Fuse(Sim3), SearchBySim3.  best = first
// minimum in GetFeaturesInArea order (strict `dist < bestDist`).
template <int GATE>
__global__ __launch_bounds__(256) void proj_best_kernel(FrameBatch F, QueryBatch Q, const float* __restrict__ inv_sigma2,
                                                        int32_t* __restrict__ best_idx, int32_t* __restrict__ best_dist) {
  __shared__ float s_inv[ORBFE_MAX_LEVELS];
  if (threadIdx.x < ORBFE_MAX_LEVELS) s_inv[threadIdx.x] = inv_sigma2 ? inv_sigma2[threadIdx.x] : 0.f;
  __syncthreads();
  const int f = blockIdx.y;
  const int qi = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (qi >= Q.n[f]) return;
  const orbfe_query* qp = Q.q + (size_t)f * Q.cap + qi;
  const orbfe_query q = *qp;
  unsigned key = 0xFFFFFFFFu;
  int mine = -1;
  if (q.valid) {
    uint4 q0, q1;
    load_desc4(qp->desc, q0, q1);
    enumerate_window<GATE>(F, f, q, q0, q1, [&](int rank, int idx, int dist, int) {
      const unsigned k = ((unsigned)dist << 16) | (unsigned)min(rank, 0xffff);
      if (k < key) { key = k; mine = idx; }
    }, s_inv);
  }
  const unsigned b = wave_min_u32(key);
  int bi = -1;
  if (b != 0xFFFFFFFFu) {
    const unsigned long long wm = __ballot(key == b);
    bi = __builtin_amdgcn_readlane(mine, __ffsll((long long)wm) - 1);
  }
  if ((threadIdx.x & 63) == 0) {
    best_idx[(size_t)f * Q.cap + qi] = bi;
    best_dist[(size_t)f * Q.cap + qi] = bi >= 0 ? (int)(b >> 16) : 256;
  }
}

// Candidate lists in enumeration order, SIXTEEN LANES PER QUERY (four queries per wave).  A search window holds a handful of
// grid entries (2.8 survivors per query at th = 7), so a whole wave per query left three quarters of the lanes idle and the
// kernel latency-bound on its four dependent round trips (query, cell ranges, cell records, descriptors) at full occupancy;
// with four queries per wave a quarter of the waves make those trips.  Same order as enumerate_window: the window's grid
// columns left to right (lane c of the group fetches column c's CSR range; 16 columns per batch), entries of the concatenated
// ranges 16 at a time, survivors ranked by a ballot inside the group.
#ifndef ORBFE_CAND_LANES
#define ORBFE_CAND_LANES 16
#endif
template <int G>   // lanes per query: 16 or 8
__global__ __launch_bounds__(256) void proj_candidates_kernel(FrameBatch F, QueryBatch Q, orbfe_cand* __restrict__ cand,
                                                               int32_t* __restrict__ n_cand, int max_cand) {
  const int f = blockIdx.y;
  constexpr int QPW = WAVE / G;   // queries per wave
  const int lane = threadIdx.x & (WAVE - 1), sl = lane & (G - 1), gbase = lane & ~(G - 1);
  const int qi = (blockIdx.x * 4 + (threadIdx.x >> 6)) * QPW + lane / G;
  const int nq = Q.n[f];
  // the group's query: every lane of the group reads the same record (slot min(qi, cap - 1) is always readable)
  const orbfe_query* qp = Q.q + (size_t)f * Q.cap + min(qi, Q.cap - 1);
  const float x = qp->u, y = qp->v, r = qp->radius, qur = qp->u_r;
  const int min_level = qp->min_level, max_level = qp->max_level;
  const bool live = qi < nq && qi < Q.cap;
  bool act = live && qp->valid;
  uint4 q0, q1;
  load_desc4(qp->desc, q0, q1);
  const int nMinCellX = max(0, (int)floorf((x - F.min_x - r) * F.gw_inv));
  const int nMaxCellX = min(ORBFE_GRID_COLS - 1, (int)ceilf((x - F.min_x + r) * F.gw_inv));
  const int nMinCellY = max(0, (int)floorf((y - F.min_y - r) * F.gh_inv));
  const int nMaxCellY = min(ORBFE_GRID_ROWS - 1, (int)ceilf((y - F.min_y + r) * F.gh_inv));
  if (nMinCellX >= ORBFE_GRID_COLS || nMaxCellX < 0 || nMinCellY >= ORBFE_GRID_ROWS || nMaxCellY < 0) act = false;   // Frame.cc:347-366
  const bool bCheckLevels = (min_level > 0) || (max_level >= 0);
  const uint8_t* desc = F.desc + (size_t)f * F.cap * 32;
  const bool has_ur = F.u_right != nullptr;
  const int32_t* cs = F.cell_start + (size_t)f * (GRID_CELLS + 1);
  const uint4* cr = reinterpret_cast<const uint4*>(F.cell_rec) + (size_t)f * F.cap;
  orbfe_cand* out = cand + ((size_t)f * Q.cap + min(qi, Q.cap - 1)) * max_cand;
  const int ncol = act ? nMaxCellX - nMinCellX + 1 : 0;
  int total = 0;
  for (int cb = 0; __any(cb < ncol); cb += G) {
    int e0c = 0, cntc = 0;
    if (cb + sl < ncol) {
      const int ix = nMinCellX + cb + sl;
      e0c = cs[ix * ORBFE_GRID_ROWS + nMinCellY];
      cntc = cs[ix * ORBFE_GRID_ROWS + nMaxCellY + 1] - e0c;
    }
    int inclc = cntc;   // inclusive scan inside the group (DPP row shifts; a shift that would cross into the group below adds 0)
    { const int a = __builtin_amdgcn_update_dpp(0, inclc, 0x111, 0xf, 0xf, false); inclc += sl >= 1 ? a : 0; }
    { const int a = __builtin_amdgcn_update_dpp(0, inclc, 0x112, 0xf, 0xf, false); inclc += sl >= 2 ? a : 0; }
    { const int a = __builtin_amdgcn_update_dpp(0, inclc, 0x114, 0xf, 0xf, false); inclc += sl >= 4 ? a : 0; }
    if (G > 8) { const int a = __builtin_amdgcn_update_dpp(0, inclc, 0x118, 0xf, 0xf, false); inclc += sl >= 8 ? a : 0; }
    const int exclc = inclc - cntc;
    const int n_entries = __shfl(inclc, gbase | (G - 1), WAVE);
    const int ncb = max(0, min(G, ncol - cb));             // columns of this batch (group-uniform)
    int ncb_max = ncb;                                     // wave-uniform loop bound
    if (G < 16) ncb_max = max(ncb_max, __shfl_xor(ncb_max, 8, WAVE));
    ncb_max = max(ncb_max, __shfl_xor(ncb_max, 16, WAVE));
    ncb_max = max(ncb_max, __shfl_xor(ncb_max, 32, WAVE));
    ncb_max = __builtin_amdgcn_readfirstlane(ncb_max);
    for (int base = 0; __any(base < n_entries); base += G) {
      const int t = base + sl;
      int ent = -1;
      for (int c = 0; c < ncb_max; c++) {
        const int oc = __shfl(exclc, gbase | c, WAVE), nc = __shfl(cntc, gbase | c, WAVE), ec = __shfl(e0c, gbase | c, WAVE);
        if (c < ncb && t >= oc && t < oc + nc) ent = ec + (t - oc);
      }
      bool ok = false;
      int idx = 0, oct = 0;
      if (ent >= 0) {
        const uint4 rec = cr[ent];
        idx = (int)(rec.x & 0xFFFFFFu);
        oct = (int)(rec.x >> 24);
        const float kx = __int_as_float((int)rec.y), ky = __int_as_float((int)rec.z), kur = __int_as_float((int)rec.w);
        ok = true;
        if (bCheckLevels) {
          if (oct < min_level) ok = false;
          if (max_level >= 0 && oct > max_level) ok = false;
        }
        const float distx = kx - x, disty = ky - y;
        if (!(fabsf(distx) < r && fabsf(disty) < r)) ok = false;
        if (ok && has_ur && kur > 0) {           // the matcher's stereo gate
          const float er = fabsf(qur - kur);
          if (er > r) ok = false;
        }
      }
      const unsigned bits = (unsigned)((__ballot(ok) >> gbase) & ((1ull << G) - 1ull));
      if (ok) {
        uint4 d0, d1;
        load_desc(desc + (size_t)idx * 32, d0, d1);
        const int rank = total + __popc(bits & ((1u << sl) - 1u));
        if (rank < max_cand) { out[rank].idx = idx; out[rank].dist = hamming256(q0, q1, d0, d1) | (oct << 16); }
      }
      total += __popc(bits);
    }
  }
  if (sl == 0 && live) n_cand[(size_t)f * Q.cap + qi] = total;
}

// ComputeThreeMaxima (L/src/ORBmatcher.cc:1506-1538)
__device__ void three_maxima(const int* hs, int L, int& ind1, int& ind2, int& ind3) {
  int max1 = 0, max2 = 0, max3 = 0;
  ind1 = ind2 = ind3 = -1;
  for (int i = 0; i < L; i++) {
    const int s = hs[i];
    if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
    else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
    else if (s > max3) { max3 = s; ind3 = i; }
  }
  if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
  else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
}

// Order-dependent assignment of SearchByProjection.  One 1024-thread workgroup per frame.  Queries are taken
// up to RC_THREADS (1024) at a time, one per thread, and resolved by a fixed-point iteration against the blocked[] state the
// chunks before left: a thread's choice is its best candidate (mode 0: best and second-best) that no EARLIER thread of the chunk
// claims; claims are exchanged once per round, and the rounds converge to exactly the result of walking the queries one by one
// (L/src/ORBmatcher.cc:52-125, 1270-1361).  A query whose candidate list was truncated (> max_cand) ends the chunk and is
// handled alone by wave 0, which re-enumerates its window.
// mode 0: SearchByProjection(Frame&, vector<MapPoint*>&)   -- best/second with the same-level ratio test
// mode 1: SearchByProjection(Frame& cur, const Frame& last) -- best only, th_high = TH_HIGH, rotation histogram;
//         SearchByProjection(Frame&, KeyFrame*, set, th, ORBdist) is the same walk with th_high = ORBdist (:1385-1504)
#ifndef FC_TIMING
#define FC_TIMING 0
#endif
#if FC_TIMING
// -DFC_TIMING=1 (tools/search_latency.py): cycles of proj_resolve_kernel's workgroup 0 per phase (thread 0's clock after each barrier)
__device__ unsigned long long g_rs_prof[16];
extern "C" int orbfe_debug_rs_profile(unsigned long long* out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_rs_prof), sizeof(unsigned long long) * 16) != hipSuccess) return 1;
  if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_rs_prof), z, sizeof(z)) != hipSuccess) return 1; }
  return 0;
}
#define RS_T(i) do { const unsigned long long _t = __builtin_readcyclecounter(); rs_acc##i += (unsigned)(_t - rs_prev); rs_prev = _t; } while (0)
#define RS_COUNT(i) do { rs_acc##i += 1; } while (0)
#define RS_END() do { if (blockIdx.x == 0 && threadIdx.x == 0) { unsigned a[11] = {rs_acc0, rs_acc1, rs_acc2, rs_acc3, rs_acc4, rs_acc5, rs_acc6, \
    rs_acc7, rs_acc8, rs_acc9, rs_acc10}; for (int _i = 0; _i < 11; _i++) g_rs_prof[_i] += a[_i]; } } while (0)
#else
#define RS_T(i)
#define RS_COUNT(i)
#define RS_END()
#endif
// best / second-best entry by distance, first one wins ties (the reference's strict "<"); RC_INVALID loses against everything.
// The pair is read into locals and written back with selects, no branches: with `if (...) e1 = e; else e2 = e;` on references (or a
// by-reference lambda) the compiler kept the pair in scratch memory and selected the ADDRESS to store to -- private-memory round
// trips inside the fixed-point loop (7 k cycles per round).
__device__ __forceinline__ void rc_take(uint32_t e, uint32_t& e1, uint32_t& e2) {
  const uint32_t a = e1, b = e2;
  const bool lt1 = (e >> 20) < (a >> 20), lt2 = (e >> 20) < (b >> 20);
  e2 = lt1 ? a : (lt2 ? e : b);
  e1 = lt1 ? e : a;
}
#define RC_THREADS 1024
#define RC_LIST_CAP 16384  // staged candidate entries per chunk (64 KiB): what does not fit the registers
#define RC_REG 4           // candidates per query kept in registers
#define RC_INVALID 0xFFFFFFFFu
#define RC_TAG_MAX 0x1fffff // claim words: (RC_TAG_MAX - round) << 10 | thread, 0x7fffffff = never claimed
__global__ __launch_bounds__(RC_THREADS) void proj_resolve_kernel(FrameBatch F, QueryBatch Q, const orbfe_cand* __restrict__ cand,
                                                           const int32_t* __restrict__ n_cand, int max_cand, int mode, int th_high,
                                                           float nnratio, int check_ori, uint8_t* __restrict__ blocked_all,
                                                           int32_t* __restrict__ assigned_all, int32_t* __restrict__ n_matches,
                                                           int32_t* __restrict__ push_idx_all, uint8_t* __restrict__ push_bin_all) {
  __shared__ int hist[ORBFE_HISTO_LENGTH];
  __shared__ uint32_t lc[RC_LIST_CAP];  // staged candidate lists: dist<<20 | octave<<16 | idx
  __shared__ int loff[RC_THREADS + 1];
  __shared__ int scan_tmp[RC_THREADS / WAVE];
  __shared__ int sh_len, sh_nm, sh_npush;
  __shared__ int sh_changed[2];
  extern __shared__ __attribute__((aligned(16))) uint8_t dyn[];
  const int capA = (F.cap + 15) & ~15;
  uint8_t* blocked = dyn;                                   // F.mvpMapPoints[idx]->Observations() > 0
  int* claim = reinterpret_cast<int*>(dyn + capA);          // smallest chunk thread that blocks idx this round
  int* claimB = claim + F.cap;                              // second claim buffer (the two alternate per iteration)
  const int f = blockIdx.x, tid = threadIdx.x, lane = tid & (WAVE - 1), wid = tid >> 6;
  const int nq = Q.n[f];
  uint8_t* blocked_g = blocked_all + (size_t)f * F.cap;
  int32_t* assigned = assigned_all + (size_t)f * F.cap;
  int32_t* push_idx = push_idx_all + (size_t)f * Q.cap;
  uint8_t* push_bin = push_bin_all + (size_t)f * Q.cap;
  const orbfe_keypoint* keys = F.keys + (size_t)f * F.cap;
  const orbfe_query* qbase = Q.q + (size_t)f * Q.cap;
  const int32_t* ncb = n_cand + (size_t)f * Q.cap;
  for (int i = tid; i < F.cap; i += RC_THREADS) { blocked[i] = blocked_g[i]; claim[i] = 0x7fffffff; claimB[i] = 0x7fffffff; }
  if (tid < ORBFE_HISTO_LENGTH) hist[tid] = 0;
  if (tid == 0) { sh_nm = 0; sh_npush = 0; }
#if FC_TIMING
  unsigned rs_acc0 = 0, rs_acc1 = 0, rs_acc2 = 0, rs_acc3 = 0, rs_acc4 = 0, rs_acc5 = 0, rs_acc6 = 0, rs_acc7 = 0, rs_acc8 = 0, rs_acc9 = 0,
           rs_acc10 = 0;
  unsigned long long rs_prev = __builtin_readcyclecounter();
#endif
  __syncthreads();
  RS_T(0);   // set-up
  const bool use_hist = (mode == 1) && check_ori;
  int q0 = 0;
  int round = 1;   // counts on across chunks: stamps the claims
  while (q0 < nq) {
    RS_COUNT(8);
    const int qi = q0 + tid;
    // one round trip: the query's flags, its candidate count and its first RC_REG candidates (slot rows exist for every qi < nq;
    // entries beyond the count are masked below)
    int tot = 0, qblocks = 0;
    float qangle = 0.f;
    uint2 craw[RC_REG];
#pragma unroll
    for (int u = 0; u < RC_REG; u++) craw[u] = make_uint2(0u, 0u);
    if (qi < nq) {
      const orbfe_query* qp = qbase + qi;
      const int qvalid = qp->valid;
      const int n = ncb[qi];
      qblocks = qp->blocks != 0;
      qangle = qp->angle;
      const uint2* row = reinterpret_cast<const uint2*>(cand + ((size_t)f * Q.cap + qi) * max_cand);
#pragma unroll
      for (int u = 0; u < RC_REG; u++) craw[u] = row[min(u, max_cand - 1)];
      tot = qvalid ? n : 0;
    }
    // chunk = queries q0 .. q0+len-1: stops before the first truncated list and before the staging area is full
    if (tid == 0) { sh_len = min(RC_THREADS, nq - q0); sh_changed[0] = 0; sh_changed[1] = 0; }
    __syncthreads();
    {
      const int v = tot > max_cand ? 0 : max(tot - RC_REG, 0);   // entries beyond the registers go through LDS
      const int w = wave_incl_scan_i(v);
      if (lane == WAVE - 1) scan_tmp[wid] = w;
      __syncthreads();
      int off = 0;
      for (int k = 0; k < wid; k++) off += scan_tmp[k];
      const int incl = off + w;
      loff[tid + 1] = incl;
      if (tid == 0) loff[0] = 0;
      if (tot > max_cand || incl > RC_LIST_CAP) atomicMin(&sh_len, tid);
    }
    __syncthreads();
    RS_T(1);   // query records + chunk scan
    const int len = sh_len;
    if (len == 0) {
      // ---- truncated list at q0: wave 0 re-enumerates this one window, the other waves wait
      if (wid == 0) {
        const orbfe_query* qp = qbase + q0;
        const orbfe_query q = *qp;
        uint4 d0, d1;
        load_desc4(qp->desc, d0, d1);
        unsigned k1 = 0xFFFFFFFFu, k2 = 0xFFFFFFFFu;
        int i1 = -1, i2 = -1;
        enumerate_window(F, f, q, d0, d1, [&](int rank, int idx, int dist, int) {
          if (blocked[idx]) return;
          const unsigned key = ((unsigned)dist << 16) | (unsigned)min(rank, 0xffff);
          if (key < k1) { k2 = k1; i2 = i1; k1 = key; i1 = idx; }
          else if (key < k2) { k2 = key; i2 = idx; }
        });
        const unsigned b = wave_min_u32(k1);
        if (b != 0xFFFFFFFFu) {
          const unsigned long long wm = __ballot(k1 == b);
          const int wl = __ffsll((long long)wm) - 1;
          const int bestIdx = __builtin_amdgcn_readlane(i1, wl);
          const int bestDist = (int)(b >> 16);
          bool accept = false;
          if (mode == 0) {
            const unsigned mine = (lane == wl) ? k2 : k1;
            const int mine_i = (lane == wl) ? i2 : i1;
            const unsigned s2 = wave_min_u32(mine);
            int bestDist2 = 256, bestLevel2 = -1;
            if (s2 != 0xFFFFFFFFu) {
              const unsigned long long sm = __ballot(mine == s2);
              const int sl = __ffsll((long long)sm) - 1;
              bestDist2 = (int)(s2 >> 16);
              bestLevel2 = keys[__shfl(mine_i, sl, WAVE)].octave;
            }
            const int bestLevel = keys[bestIdx].octave;
            if (bestDist <= th_high && !(bestLevel == bestLevel2 && (float)bestDist > nnratio * (float)bestDist2))
              accept = true;
          } else {
            accept = bestDist <= th_high;
          }
          if (accept && lane == 0) {
            assigned[bestIdx] = q0;
            blocked[bestIdx] = (uint8_t)(q.blocks != 0);
            sh_nm += 1;
            if (use_hist) {
              float rot = q.angle - keys[bestIdx].angle;
              if (rot < 0.0f) rot += 360.0f;
              int bin = (int)roundf(rot * (1.0f / ORBFE_HISTO_LENGTH));  // the reference's factor (sic), :1255
              if (bin == ORBFE_HISTO_LENGTH) bin = 0;
              push_idx[sh_npush] = bestIdx;
              push_bin[sh_npush] = (uint8_t)bin;
              sh_npush += 1;
              hist[bin]++;
            }
          }
        }
      }
      __syncthreads();
      q0 += 1;
      continue;
    }
    // ---- the chunk's candidates, packed dist<<20 | octave<<16 | idx, COMPACTED in list order: an entry is dropped when its
    // keypoint is blocked (blocked[] only changes when a chunk commits, so that holds for every round below) or when its distance
    // cannot matter -- beyond th_high it is never accepted, and as a second-best (mode 0) it rejects the best one only while
    // nnratio * d < th_high.  What is left is the true match and a near miss or two: the first RC_REG entries stay in registers,
    // a longer list puts the rest in LDS (and its owner re-selects in every round).
    uint32_t creg[RC_REG];
#pragma unroll
    for (int u = 0; u < RC_REG; u++) creg[u] = RC_INVALID;
    int kept = 0, nover = 0;
    auto matters = [&](uint32_t d) {
      return d <= (uint32_t)th_high || (mode == 0 && d < 256u && nnratio * (float)d < (float)th_high);
    };
#pragma unroll
    for (int u = 0; u < RC_REG; u++) {
      const uint32_t idx = craw[u].x & 0xffffu, dd = craw[u].y;
      const uint32_t e = ((dd & 0xffffu) << 20) | (((dd >> 16) & 0xfu) << 16) | idx;
      const bool ok = tid < len && u < tot && matters(dd & 0xffffu) && !blocked[min(idx, (uint32_t)F.cap - 1u)];
#pragma unroll
      for (int j = 0; j <= u; j++) creg[j] = (ok && kept == j) ? e : creg[j];
      kept += ok;
    }
    {
      const int nraw = (tid < len) ? max(tot - RC_REG, 0) : 0;
      const uint2* row = reinterpret_cast<const uint2*>(cand + ((size_t)f * Q.cap + qi) * max_cand) + RC_REG;
      uint32_t* dstl = lc + loff[tid < len ? tid : 0];
      for (int c0 = 0; c0 < nraw; c0 += 4) {
        uint2 r[4];
#pragma unroll
        for (int u = 0; u < 4; u++) r[u] = row[min(c0 + u, nraw - 1)];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const uint32_t idx = r[u].x & 0xffffu, dd = r[u].y;
          if (c0 + u >= nraw || !matters(dd & 0xffffu) || blocked[idx]) continue;
          const uint32_t e = ((dd & 0xffffu) << 20) | (((dd >> 16) & 0xfu) << 16) | idx;
          if (kept < RC_REG) {
#pragma unroll
            for (int j = 0; j < RC_REG; j++) creg[j] = (kept == j) ? e : creg[j];
          } else {
            dstl[nover++] = e;
          }
          kept++;
        }
      }
    }
    __syncthreads();
    RS_T(2);   // staging
    const bool active = kept > 0;
    const uint32_t* mylist = lc + loff[tid < len ? tid : 0];
    // Fixed-point iteration over the chunk.  Thread t's choice = best candidate that is neither blocked nor
    // claimed (blocked-to-be) by an EARLIER thread of the chunk; claims come from the previous round.
    // Thread t's choice is final after at most t+1 rounds (it only depends on earlier threads), so the
    // iteration converges to exactly the sequential result; in practice the dependency chains are 2-5 deep.
    // One barrier per round: a claim is the word (RC_TAG_MAX - round) << 10 | thread, written by atomicMin into the buffer of
    // the round's parity and read back from the other buffer in the next round -- a newer round always wins the min, an older
    // word fails the tag test, so nothing is ever cleared (rounds count on across chunks; 2^21 of them fit).  The "somebody
    // changed" flag carries the round number the same way.  A barrier costs ~0.5 us with sixteen waves, a round hardly more.
    // state = the two best entries; RC_INVALID (distance field 0xfff) = none.  Plain 32-bit values only: bool flags updated
    // through a by-reference lambda ended up in scratch memory, several private-memory round trips per iteration (7 k cycles).
    uint32_t e1 = RC_INVALID, e2 = RC_INVALID;
    uint32_t pmask = ~0u;   // which of the register candidates an earlier query claimed, previous round
    int accept = 0;
    for (int iter = 0;; iter++, round++) {
      const int* cprev = (round & 1) ? claim : claimB;   // written in round - 1
      int* cnew = (round & 1) ? claimB : claim;
      const int want = (RC_TAG_MAX - (round - 1));       // tag of a claim made in the previous round
      bool changed = false;
      if (active) {
        int cl[RC_REG];
        uint32_t m = 0;
        if (iter > 0) {   // (the previous round of the first iteration belongs to the chunk before: no claims yet)
#pragma unroll
          for (int u = 0; u < RC_REG; u++) cl[u] = cprev[creg[u] == RC_INVALID ? 0 : (int)(creg[u] & 0xffffu)];
#pragma unroll
          for (int u = 0; u < RC_REG; u++)
            m |= (creg[u] != RC_INVALID && (cl[u] >> 10) == want && (cl[u] & (RC_THREADS - 1)) < tid) ? (1u << u) : 0u;
        }
        // The choice is a function of the claim pattern alone: same pattern, same choice.
        if (m != pmask || nover > 0) {
          pmask = m;
          const uint32_t pe1 = e1, pe2 = e2;
          e1 = e2 = RC_INVALID;
          accept = 0;
#pragma unroll
          for (int u = 0; u < RC_REG; u++) rc_take(((m >> u) & 1u) ? RC_INVALID : creg[u], e1, e2);
          for (int c0 = 0; c0 < nover; c0 += 4) {  // 4 candidates in flight: independent LDS loads first, then the scan
            uint32_t ev[4];
            int co[4];
#pragma unroll
            for (int u = 0; u < 4; u++) ev[u] = (c0 + u < nover) ? mylist[c0 + u] : RC_INVALID;
#pragma unroll
            for (int u = 0; u < 4; u++) co[u] = cprev[ev[u] == RC_INVALID ? 0 : (int)(ev[u] & 0xffffu)];
#pragma unroll
            for (int u = 0; u < 4; u++) {
              const bool taken = iter > 0 && (co[u] >> 10) == want && (co[u] & (RC_THREADS - 1)) < tid;
              rc_take(taken ? RC_INVALID : ev[u], e1, e2);
            }
          }
          if (e1 != RC_INVALID) {
            const int bestDist = (int)(e1 >> 20);
            if (mode == 0) {
              const int bestDist2 = e2 != RC_INVALID ? (int)(e2 >> 20) : 256;
              const int bestLevel = (int)((e1 >> 16) & 0xf), bestLevel2 = e2 != RC_INVALID ? (int)((e2 >> 16) & 0xf) : -1;
              accept = bestDist <= th_high && !(bestLevel == bestLevel2 && (float)bestDist > nnratio * (float)bestDist2);
            } else {
              accept = bestDist <= th_high;
            }
          }
          changed = e1 != pe1 || e2 != pe2;
        }
      }
      const int bestIdx = (int)(e1 & 0xffff);
      if (accept && qblocks) atomicMin(&cnew[bestIdx], ((RC_TAG_MAX - round) << 10) | tid);
      if (iter == 0 || changed) sh_changed[round & 1] = round;
      __syncthreads();
      RS_T(3);   // fixed-point rounds
      RS_COUNT(9);
      if (sh_changed[round & 1] == round) continue;
      // converged: commit every accepted query of the chunk.  Several queries may take the same keypoint
      // (only when the earlier ones do not block it): the last one in query order wins, as in the reference.
      round++;
      int* cw = (round & 1) ? claimB : claim;
      const int mine = ((RC_TAG_MAX - round) << 10) | (RC_THREADS - 1 - tid);
      if (active && accept) atomicMin(&cw[bestIdx], mine);
      __syncthreads();
      if (active && accept) {
        if (cw[bestIdx] == mine) {
          assigned[bestIdx] = qi;
          blocked[bestIdx] = (uint8_t)qblocks;
        }
        atomicAdd(&sh_nm, 1);
        if (use_hist) {
          float rot = qangle - keys[bestIdx].angle;
          if (rot < 0.0f) rot += 360.0f;
          int bin = (int)roundf(rot * (1.0f / ORBFE_HISTO_LENGTH));  // the reference's factor (sic), :1255
          if (bin == ORBFE_HISTO_LENGTH) bin = 0;
          const int pos = atomicAdd(&sh_npush, 1);
          push_idx[pos] = bestIdx;
          push_bin[pos] = (uint8_t)bin;
          atomicAdd(&hist[bin], 1);
        }
      }
      round++;   // (no barrier here: the next chunk's first round writes the other buffer, and its set-up has barriers of its own)
      RS_T(4);   // commit
      break;
    }
    q0 += len;
  }
  __syncthreads();
  if (use_hist) {
    // rotation consistency (:1363-1379): every recorded match whose bin is not one of the three maxima is removed
    if (tid == 0) {
      int i1, i2, i3;
      three_maxima(hist, ORBFE_HISTO_LENGTH, i1, i2, i3);
      hist[0] = i1; hist[1] = i2; hist[2] = i3;  // the histogram itself is no longer needed
    }
    __syncthreads();
    const int i1 = hist[0], i2 = hist[1], i3 = hist[2];
    const int np = sh_npush;
    int removed = 0;
    for (int k = tid; k < np; k += RC_THREADS) {
      const int bin = push_bin[k];
      if (bin != i1 && bin != i2 && bin != i3) { assigned[push_idx[k]] = -1; blocked[push_idx[k]] = 0; removed++; }   // the slot is NULL again
    }
    if (removed) atomicSub(&sh_nm, removed);
  }
  __syncthreads();
  for (int i = tid; i < F.cap; i += RC_THREADS) blocked_g[i] = blocked[i];
  if (tid == 0) n_matches[f] = sh_nm;
  RS_T(5);   // rotation check + write-back
  RS_COUNT(10);
  RS_END();
}

// ------------------------------------------------------------------------------------------------ SearchByBoW
// SearchByBoW(KeyFrame*, Frame&, ...) (L/src/ORBmatcher.cc:161-273).  The host merge-joins the two FeatureVectors
// on NodeId; every common node is an independent problem (a feature belongs to exactly one node), so one wave
// takes one node: keyframe features in vIndicesKF order, the lanes share the node's frame features (best /
// second-best over those not matched yet, strict "<" in vIndicesF order), matches written immediately so that
// later keyframe features skip them (:208-209).  Rotation histogram by global atomics, finished by bow_finish.
__global__ __launch_bounds__(64) void bow_match_kernel(BowParams P) {
  extern __shared__ __attribute__((aligned(16))) uint8_t taken[];  // per frame feature of this node
  const int lane = threadIdx.x;
  // sequential mode (a frame feature listed under several nodes -- never produced by DBoW2, but legal input): one
  // wave walks all node pairs in NodeId order and reads the match table from memory, as the reference does
  const int first = P.sequential ? 0 : blockIdx.x, last = P.sequential ? P.n_pairs : blockIdx.x + 1;
  for (int pi = first; pi < last; pi++) {
  const BowPair pr = P.pairs[pi];
  __syncthreads();
  for (int i = lane; i < pr.countB; i += WAVE)
    taken[i] = (uint8_t)((P.sequential && __hip_atomic_load(&P.matchB[P.idxB[pr.startB + i]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= 0) ||
                         (P.validB && !P.validB[P.idxB[pr.startB + i]]));  // KF-KF: pMP2 missing or bad (:543-549)
  __syncthreads();
  for (int iKF = 0; iKF < pr.countA; iKF++) {
    const int realIdxKF = P.idxA[pr.startA + iKF];
    if (!P.validA[realIdxKF]) continue;  // no map point, or a bad one (:191-197)
    uint4 a0, a1;
    load_desc(P.descA + (size_t)realIdxKF * 32, a0, a1);
    int bestd = 256, bestj = -1, second = 256;
    for (int iF = lane; iF < pr.countB; iF += WAVE) {
      if (taken[iF]) continue;
      const int realIdxF = P.idxB[pr.startB + iF];
      uint4 b0, b1;
      load_desc(P.descB + (size_t)realIdxF * 32, b0, b1);
      const int d = hamming256(a0, a1, b0, b1);
      if (d < bestd) { second = bestd; bestd = d; bestj = iF; }
      else if (d < second) second = d;
    }
    // merge: first minimum in vIndicesF order, second = smallest of everything else
    const unsigned key = bestj >= 0 ? (((unsigned)bestd << 16) | (unsigned)bestj) : 0xFFFFFFFFu;
    const unsigned b = wave_min_u32(key);
    if (b != 0xFFFFFFFFu) {
      const unsigned other = (key == b) ? 0xFFFFFFFFu : key;
      int s2 = min(second, other != 0xFFFFFFFFu ? (int)(other >> 16) : 256);
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) s2 = min(s2, __shfl_xor(s2, d, WAVE));
      const int bestDist1 = (int)(b >> 16), bestF = (int)(b & 0xffff);
      // KF-Frame accepts bestDist <= TH_LOW (:224), KF-KF only bestDist < TH_LOW (:564)
      if ((P.kf_mode ? bestDist1 < ORBFE_TH_LOW : bestDist1 <= ORBFE_TH_LOW) && (float)bestDist1 < P.nnratio * (float)s2) {
        if (lane == 0) {
          const int realIdxF = P.idxB[pr.startB + bestF];
          __hip_atomic_store(&P.matchB[realIdxF], realIdxKF, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (P.kf_mode) P.matchA[realIdxKF] = realIdxF;
          taken[bestF] = 1;
          if (P.sequential)  // the same feature may be listed again inside this node
            for (int t = 0; t < pr.countB; t++)
              if (P.idxB[pr.startB + t] == realIdxF) taken[t] = 1;
          if (P.check_ori) {
            float rot = P.angleA[realIdxKF] - P.angleB[realIdxF];
            if (rot < 0.0f) rot += 360.0f;
            int bin = (int)roundf(rot * (1.0f / ORBFE_HISTO_LENGTH));
            if (bin == ORBFE_HISTO_LENGTH) bin = 0;
            const int pos = atomicAdd(&P.counters[0], 1);
            P.push_idx[pos] = P.kf_mode ? realIdxKF : realIdxF;  // rotHist holds idx1 in the KF-KF variant (:579)
            P.push_bin[pos] = (uint8_t)bin;
            atomicAdd(&P.counters[2 + bin], 1);
          }
          atomicAdd(&P.counters[1], 1);
        }
      }
    }
    __syncthreads();
  }
  }
}

// SearchForTriangulation (L/src/ORBmatcher.cc:614-764): one wave per common vocabulary node.  Nothing couples two features
// of pKF1 (vbMatched2 is never set), so each is an independent arg-min over the node's pKF2 features that pass the
// gates: dist <= TH_LOW, the epipole distance for two monocular keypoints (:697-703), CheckDistEpipolarLine (:137-159);
// `dist > bestDist` rejects, so of equal distances the LAST one in list order wins.
__global__ __launch_bounds__(64) void triangulation_match_kernel(TriParams T) {
  const BowParams& P = T.b;
  const int lane = threadIdx.x;
  const int first = P.sequential ? 0 : blockIdx.x, last = P.sequential ? P.n_pairs : blockIdx.x + 1;
  for (int pi = first; pi < last; pi++) {
    const BowPair pr = P.pairs[pi];
    for (int i1 = 0; i1 < pr.countA; i1++) {
      const int idx1 = P.idxA[pr.startA + i1];
      if (!P.validA[idx1]) continue;   // has a map point, or monocular under bOnlyStereo (:655-664)
      const bool bStereo1 = T.stereoA[idx1] != 0;
      const orbfe_keypoint kp1 = T.keysA[idx1];
      uint4 a0, a1;
      load_desc(P.descA + (size_t)idx1 * 32, a0, a1);
      // epipolar line in the second image l = x1' F12 = [a b c]
      const float* F = T.ep.F12;
      const float la = kp1.x * F[0] + kp1.y * F[3] + F[6];
      const float lb = kp1.x * F[1] + kp1.y * F[4] + F[7];
      const float lc = kp1.x * F[2] + kp1.y * F[5] + F[8];
      const float den = la * la + lb * lb;
      unsigned key = 0xFFFFFFFFu;   // dist << 16 | (0xffff - position): minimum = smallest distance, last position
      for (int i2 = lane; i2 < pr.countB; i2 += WAVE) {
        const int idx2 = P.idxB[pr.startB + i2];
        if (!P.validB[idx2]) continue;   // matched map point, or monocular under bOnlyStereo (:677-686)
        uint4 b0, b1;
        load_desc(P.descB + (size_t)idx2 * 32, b0, b1);
        const int dist = hamming256(a0, a1, b0, b1);
        if (dist > ORBFE_TH_LOW) continue;
        const orbfe_keypoint kp2 = T.keysB[idx2];
        if (!bStereo1 && !T.stereoB[idx2]) {
          const float distex = T.ep.ex - kp2.x, distey = T.ep.ey - kp2.y;
          if (distex * distex + distey * distey < 100 * T.ep.scale_factors[kp2.octave & (ORBFE_MAX_LEVELS - 1)]) continue;
        }
        const float num = la * kp2.x + lb * kp2.y + lc;
        if (den == 0) continue;
        const float dsqr = num * num / den;
        if (!((double)dsqr < 3.84 * (double)T.ep.level_sigma2[kp2.octave & (ORBFE_MAX_LEVELS - 1)])) continue;
        const unsigned k = ((unsigned)dist << 16) | (unsigned)(0xffff - i2);
        key = min(key, k);
      }
      const unsigned b = wave_min_u32(key);
      if (b != 0xFFFFFFFFu && lane == 0) {
        const int idx2 = P.idxB[pr.startB + (0xffff - (int)(b & 0xffff))];
        P.matchA[idx1] = idx2;
        atomicAdd(&P.counters[1], 1);
        if (P.check_ori) {
          float rot = kp1.angle - T.keysB[idx2].angle;
          if (rot < 0.0f) rot += 360.0f;
          int bin = (int)roundf(rot * (1.0f / ORBFE_HISTO_LENGTH));
          if (bin == ORBFE_HISTO_LENGTH) bin = 0;
          const int pos = atomicAdd(&P.counters[0], 1);
          P.push_idx[pos] = idx1;
          P.push_bin[pos] = (uint8_t)bin;
          atomicAdd(&P.counters[2 + bin], 1);
        }
      }
    }
  }
}

__global__ __launch_bounds__(256) void bow_finish_kernel(BowParams P) {
  __shared__ int top[3];
  __shared__ int removed;
  if (threadIdx.x == 0) {
    removed = 0;
    int i1, i2, i3;
    three_maxima(P.counters + 2, ORBFE_HISTO_LENGTH, i1, i2, i3);
    top[0] = i1; top[1] = i2; top[2] = i3;
  }
  __syncthreads();
  if (P.check_ori) {
    const int np = P.counters[0];
    int r = 0;
    for (int k = threadIdx.x; k < np; k += 256) {
      const int bin = P.push_bin[k];
      if (bin != top[0] && bin != top[1] && bin != top[2]) { (P.kf_mode ? P.matchA : P.matchB)[P.push_idx[k]] = -1; r++; }
    }
    if (r) atomicAdd(&removed, r);
  }
  __syncthreads();
  if (threadIdx.x == 0) P.counters[1] -= removed;
}

// ------------------------------------------------------------------------------------------------ mono init
// ORBmatcher::SearchForInitialization (L/src/ORBmatcher.cc:388-492).  Every F1 keypoint may steal an F2 keypoint
// from an earlier one when it is strictly closer (vMatchedDistance / vnMatches21), so the queries are walked
// strictly in order by ONE wave; the lanes share each query's candidates (stored list, or the re-enumerated
// window when the list was truncated).  Runs a handful of times per session (map initialisation).
__global__ __launch_bounds__(64) void init_resolve_kernel(FrameBatch F, QueryBatch Q, const orbfe_cand* __restrict__ cand,
                                                           const int32_t* __restrict__ n_cand, int max_cand, float nnratio,
                                                           int check_ori, int32_t* __restrict__ matches12,
                                                           float* __restrict__ prev_xy, int32_t* __restrict__ n_matches,
                                                           int32_t* __restrict__ push_idx, uint8_t* __restrict__ push_bin) {
  __shared__ int hist[ORBFE_HISTO_LENGTH];
  extern __shared__ __attribute__((aligned(16))) uint8_t dyn[];
  int* matchedDist = reinterpret_cast<int*>(dyn);  // vMatchedDistance
  int* matches21 = matchedDist + F.cap;             // vnMatches21
  const int lane = threadIdx.x;
  const int n1 = Q.n[0], n2 = F.n[0];
  const orbfe_keypoint* keys2 = F.keys;
  for (int i = lane; i < n2; i += WAVE) { matchedDist[i] = 2147483647; matches21[i] = -1; }
  for (int i = lane; i < n1; i += WAVE) matches12[i] = -1;
  if (lane < ORBFE_HISTO_LENGTH) hist[lane] = 0;
  __syncthreads();
  int npush = 0;
  for (int i1 = 0; i1 < n1; i1++) {
    const orbfe_query* qp = Q.q + i1;
    if (!qp->valid) continue;  // level1 > 0
    const int total = n_cand[i1];
    if (total == 0) continue;
    unsigned k1 = 0xFFFFFFFFu, k2 = 0xFFFFFFFFu;
    int i1b = -1;
    auto consider = [&](int rank, int idx, int dist, int = 0) {
      if (matchedDist[idx] <= dist) return;
      const unsigned key = ((unsigned)dist << 16) | (unsigned)min(rank, 0xffff);
      if (key < k1) { k2 = k1; k1 = key; i1b = idx; }
      else if (key < k2) { k2 = key; }
    };
    if (total <= max_cand) {
      const orbfe_cand* cl = cand + (size_t)i1 * max_cand;
      for (int c = lane; c < total; c += WAVE) consider(c, cl[c].idx, cl[c].dist & 0xffff, 0);
    } else {
      const orbfe_query q = *qp;
      uint4 d0, d1;
      load_desc4(qp->desc, d0, d1);
      enumerate_window(F, 0, q, d0, d1, consider);
    }
    const unsigned b = wave_min_u32(k1);
    if (b != 0xFFFFFFFFu) {
      const unsigned long long wm = __ballot(k1 == b);
      const int wl = __ffsll((long long)wm) - 1;
      const int bestIdx2 = __builtin_amdgcn_readlane(i1b, wl);
      const int bestDist = (int)(b >> 16);
      const unsigned mine = (lane == wl) ? k2 : k1;
      const unsigned s2 = wave_min_u32(mine);
      const int bestDist2 = s2 != 0xFFFFFFFFu ? (int)(s2 >> 16) : 2147483647;
      if (bestDist <= ORBFE_TH_LOW && (float)bestDist < (float)bestDist2 * nnratio) {
        if (lane == 0) {
          const int prev = matches21[bestIdx2];
          if (prev >= 0) matches12[prev] = -1;
          matches12[i1] = bestIdx2;
          matches21[bestIdx2] = i1;
          matchedDist[bestIdx2] = bestDist;
          if (check_ori) {
            float rot = qp->angle - keys2[bestIdx2].angle;
            if (rot < 0.0f) rot += 360.0f;
            int bin = (int)roundf(rot * (1.0f / ORBFE_HISTO_LENGTH));
            if (bin == ORBFE_HISTO_LENGTH) bin = 0;
            push_idx[npush] = i1;
            push_bin[npush] = (uint8_t)bin;
            hist[bin]++;
          }
        }
        npush += check_ori ? 1 : 0;
      }
    }
    __syncthreads();
  }
  __syncthreads();
  if (lane == 0) {
    if (check_ori) {
      int a, b2, c;
      three_maxima(hist, ORBFE_HISTO_LENGTH, a, b2, c);
      for (int k = 0; k < npush; k++) {
        const int bin = push_bin[k];
        if (bin != a && bin != b2 && bin != c) {
          const int idx1 = push_idx[k];
          if (matches12[idx1] >= 0) matches12[idx1] = -1;
        }
      }
    }
  }
  __syncthreads();
  int cnt = 0;
  for (int i = lane; i < n1; i += WAVE) {
    const int m = matches12[i];
    if (m >= 0) {
      cnt++;
      prev_xy[2 * i] = keys2[m].x;      // vbPrevMatched[i1] = F2.mvKeysUn[vnMatches12[i1]].pt (:487-489)
      prev_xy[2 * i + 1] = keys2[m].y;
    }
  }
  // the reference's nmatches (++ on assignment, -- on steal / rotation reject) == F1 keypoints still holding a match
  cnt = wave_sum_i32(cnt);
  if (lane == 0) n_matches[0] = cnt;
}

// ------------------------------------------------------------------------------------------------ stereo
// Right keypoints are first binned by the 8-row buckets their band [floor(y-r), ceil(y+r)], r = 2*scale[octave],
// overlaps (one block per pair).  A left keypoint then only visits the bucket of its row (int)vL and re-tests
// the exact band.  The reference walks its per-row lists in ascending right index with a strict "<"
// (L/src/Frame.cc:493-502,536-553): the (distance, index) lexicographic minimum reproduces that first-wins
// rule whatever order the candidates are visited in.
__global__ __launch_bounds__(256) void stereo_bucket_kernel(StereoParams P) {
  // keys = (row bucket, octave), row-bucket major: the entries a left keypoint of level l may match -- octaves l - 1 .. l + 1
  // of its row bucket -- are ONE contiguous run of the CSR array, a third as long as the whole row bucket (39 entries on
  // average at KITTI geometry instead of 104: one 64-lane chunk for 97 % of the left keypoints instead of two to four)
  extern __shared__ int sb_lds[];
  int* cnt = sb_lds;                 // [n_keys]
  int* start = sb_lds + P.n_keys;    // [n_keys + 1]
  __shared__ int part[256];
  const int pair = blockIdx.x, tid = threadIdx.x;
  const int nR = P.nR[pair];
  const int nb = P.n_buckets, nl = P.n_levels, nk = P.n_keys;
  const orbfe_keypoint* kr = P.kpsR + (size_t)pair * P.cap;
  int32_t* bs = P.bucket_start + (size_t)pair * (nk + 1);
  int4* bi = reinterpret_cast<int4*>(P.bucket_idx) + (size_t)pair * P.cap * STEREO_BUCKET_SPAN;
  for (int b = tid; b < nk; b += 256) cnt[b] = 0;
  __syncthreads();
  for (int pass = 0; pass < 2; pass++) {
    for (int iR = tid; iR < nR; iR += 256) {
      const orbfe_keypoint kp = kr[iR];
      const int oct = min(max(kp.octave, 0), nl - 1);
      const float r = 2.0f * P.scale[oct];
      const int maxr = (int)ceilf(kp.y + r), minr = (int)floorf(kp.y - r);
      int b0 = max(minr, 0) >> 3, b1 = min(maxr >> 3, nb - 1);
      if (b1 - b0 >= STEREO_BUCKET_SPAN) b1 = b0 + STEREO_BUCKET_SPAN - 1;  // cannot happen for scale <= 12
      for (int b = b0; b <= b1; b++) {
        const int key = b * nl + oct;
        const int pos = atomicAdd(&cnt[key], 1);
        if (pass == 1) bi[start[key] + pos] = make_int4(iR, __float_as_int(kp.x), __float_as_int(kp.y), kp.octave);
      }
    }
    __syncthreads();
    if (pass == 0) {
      // exclusive prefix sum over the keys: every thread sums a contiguous slice, the 256 slice totals are scanned serially
      const int per = (nk + 255) / 256;
      const int k0 = min(tid * per, nk), k1 = min(k0 + per, nk);
      int sum = 0;
      for (int k = k0; k < k1; k++) sum += cnt[k];
      part[tid] = sum;
      __syncthreads();
      if (tid == 0) {
        int run = 0;
        for (int i = 0; i < 256; i++) { const int v = part[i]; part[i] = run; run += v; }
        start[nk] = run;
      }
      __syncthreads();
      int run = part[tid];
      for (int k = k0; k < k1; k++) { start[k] = run; run += cnt[k]; }
      __syncthreads();
      for (int b = tid; b <= nk; b += 256) bs[b] = start[b];
      for (int b = tid; b < nk; b += 256) cnt[b] = 0;
      __syncthreads();
    }
  }
}

__device__ __forceinline__ unsigned sad_u32(unsigned a, unsigned b, unsigned c) {  // |a - b| + c
  unsigned r;
  asm("v_sad_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
// DPP rotations inside a row of 16 lanes: every lane of the row ends up with the row's minimum / sum / or
#define ORBFE_ROW_ALL(op, v) do { \
    v = op(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, false)); /* row_ror:8 */ \
    v = op(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x124, 0xf, 0xf, false)); /* row_ror:4 */ \
    v = op(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x122, 0xf, 0xf, false)); /* row_ror:2 */ \
    v = op(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x121, 0xf, 0xf, false)); /* row_ror:1 */ \
  } while (0)
__device__ __forceinline__ unsigned op_min_u(unsigned a, unsigned b) { return min(a, b); }
__device__ __forceinline__ unsigned op_add_u(unsigned a, unsigned b) { return a + b; }
__device__ __forceinline__ unsigned op_or_u(unsigned a, unsigned b) { return a | b; }
__device__ __forceinline__ unsigned row_min_u32(unsigned v) { ORBFE_ROW_ALL(op_min_u, v); return v; }
__device__ __forceinline__ unsigned row_sum_u32(unsigned v) { ORBFE_ROW_ALL(op_add_u, v); return v; }
__device__ __forceinline__ unsigned row_or_u32(unsigned v) { ORBFE_ROW_ALL(op_or_u, v); return v; }

// Per-level constants a lane needs once its keypoint's level is a per-lane value (kernel arguments cannot be indexed by a VGPR)
struct SmLevel {
  const uint8_t* baseL; const uint8_t* baseR;
  unsigned long long strideL, strideR;
  int pitchL, pitchR, wL, hL, wR, hR;
  float scale, inv_scale;
  int tiledL, tiledR;   // the level lies in 16 x 8 tiles (orbfe_internal.h: written by the fused level kernel), not row-major
};
#define SM_G 16                      // lanes per left keypoint
#define SM_KPB (256 / SM_G)          // left keypoints per workgroup
#ifndef SM_ROUNDS
#define SM_ROUNDS 2
#endif
// SM_ROUNDS: bucket entries requested up front per keypoint: SM_ROUNDS * SM_G
#ifndef SM_SAD_T
#define SM_SAD_T 8   // rounds of 16 window pixels (timing experiments only: fewer = wrong sums)
#endif
#define SAD_LP 32   // LDS pitch of the staged left window: 11 pixels at byte offset <= 15 of two aligned 16-byte pieces
#define SAD_RP 48   // LDS pitch of the staged right window: 21 pixels at byte offset <= 15 of three aligned 16-byte pieces
#define SAD_WIN (11 * SAD_LP + 11 * SAD_RP + 16)   // a multiple of 16: every keypoint's slice starts 16-byte aligned
// Sixteen lanes per left keypoint, four keypoints per wave (L/src/Frame.cc:504-632).  What the whole-wave kernel paid once per
// keypoint -- the wave-uniform set-up, six 64-lane reductions, the scalar epilogue -- is paid once per FOUR here: the keypoint's
// own values live in the lanes of its DPP row, reductions are four row rotations, the first-minimum / parabola epilogue runs with
// lane k = shift k.  A keypoint that drops out (no candidate, window outside the level) only idles its row; the wave leaves when
// all four have.
#ifndef SM_XCD_PAIRS
#define SM_XCD_PAIRS 1
#endif
__global__ __launch_bounds__(256) void stereo_match_kernel(StereoParams P) {
  int pair = blockIdx.y, bx = blockIdx.x;
#if SM_XCD_PAIRS
  // whole pairs per XCD (workgroups go to the XCDs round-robin in linear order): XCD k works on pairs k, k + 8, ...: the two
  // pyramids a pair's SAD windows are cut from (3 MB) stay in one L2 instead of every pair passing through all eight
  {
    const unsigned gx = gridDim.x;
    const unsigned lin = blockIdx.y * gx + blockIdx.x;
    const unsigned g8 = lin / (8u * gx);
    if (8u * g8 + 8u <= gridDim.y) {
      const unsigned within = lin - g8 * 8u * gx;
      pair = (int)(8u * g8 + (within & 7u));
      bx = (int)(within >> 3);
    }
  }
#endif
  const int lane = threadIdx.x & (WAVE - 1);
  const int sub = lane & (SM_G - 1);
  const int grp = threadIdx.x / SM_G;
  const int iL = bx * SM_KPB + grp;
  __shared__ SmLevel s_lv[ORBFE_MAX_LEVELS];
  __shared__ float s_r[ORBFE_MAX_LEVELS];  // r = 2 * scale[octave] of a right keypoint (L/src/Frame.cc:496)
  __shared__ __attribute__((aligned(16))) uint8_t sad_win[SM_KPB][SAD_WIN];
  __shared__ uint16_t s_offL[128], s_offR[128];   // window pixel p = 11 * yy + xx -> byte offset inside the staged windows
  if (threadIdx.x >= 128) {
    const unsigned p = threadIdx.x - 128, yy = p / 11u, xx = p - yy * 11u;
    s_offL[p] = (uint16_t)(p < 121 ? yy * SAD_LP + xx : 0);
    s_offR[p] = (uint16_t)(p < 121 ? yy * SAD_RP + xx : 0);
  }
  if (threadIdx.x < ORBFE_MAX_LEVELS) {
    const int l = threadIdx.x;
    const bool in = l < P.n_levels;
    SmLevel v;
    v.baseL = P.pyrL.base[l] + (size_t)pair * P.pyrL.img_stride[l];
    v.baseR = P.pyrR.base[l] + (size_t)pair * P.pyrR.img_stride[l];
    v.strideL = 0; v.strideR = 0;
    v.pitchL = P.pyrL.pitch[l]; v.pitchR = P.pyrR.pitch[l];
    v.wL = in ? P.pyrL.w[l] : 0; v.hL = in ? P.pyrL.h[l] : 0;
    v.wR = in ? P.pyrR.w[l] : 0; v.hR = in ? P.pyrR.h[l] : 0;
    v.scale = P.scale[l]; v.inv_scale = P.inv_scale[l];
    v.tiledL = (int)((P.pyrL.tiled >> l) & 1u); v.tiledR = (int)((P.pyrR.tiled >> l) & 1u);
    s_lv[l] = v;
    s_r[l] = 2.0f * P.scale[l];
  }
  __syncthreads();
  const int cap = P.cap, nl = P.n_levels;
  float* out_ur = P.u_right + (size_t)pair * cap;
  float* out_depth = P.depth + (size_t)pair * cap;
  int32_t* out_sad = P.sad + (size_t)pair * cap;
  const orbfe_keypoint* kl = P.kpsL + (size_t)pair * cap;
  const uint8_t* dl = P.descL + (size_t)pair * cap * 32;
  const uint8_t* dr = P.descR + (size_t)pair * cap * 32;
  // first round trip: count, left keypoint record and left descriptor together (the slot is clamped into the array)
  const int nL = P.nL[pair];
  const int iLc = min(iL, cap - 1);
  const float uL = kl[iLc].x, vL = kl[iLc].y;
  const int octL = kl[iLc].octave;
  uint4 a0, a1;
  load_desc(dl + (size_t)iLc * 32, a0, a1);
  if (sub == 0 && iL < cap) { out_ur[iL] = -1.0f; out_depth[iL] = -1.0f; out_sad[iL] = -1; }
  const int levelL = min(max(octL, 0), nl - 1);
  const int nRows = P.pyrL.h[0];
  const int row = (int)vL;
  const float minD = 0, maxD = P.maxD;
  const float minU = uL - maxD, maxU = uL - minD;
  bool act = iL < nL && row >= 0 && row < nRows && !(maxU < 0);
  const int32_t* bs = P.bucket_start + (size_t)pair * (P.n_keys + 1);
  const int4* bent = reinterpret_cast<const int4*>(P.bucket_idx) + (size_t)pair * cap * STEREO_BUCKET_SPAN;
  // the run of this row bucket's entries with octave levelL - 1 .. levelL + 1 (L/src/Frame.cc:538-539 keeps nothing else)
  const int kb = act ? (row >> 3) * nl : 0;
  const int e0 = bs[kb + max(levelL - 1, 0)];
  const int e1 = act ? bs[kb + min(levelL + 1, nl - 1) + 1] : e0;
  unsigned best = ((unsigned)ORBFE_TH_HIGH << 16) | 0xffffu;  // bestDist starts at TH_HIGH; strict <
  float best_x = 0.f;                                         // x of this lane's best candidate
  const float rowm1 = (float)(row - 1), rowp1 = (float)(row + 1);   // exact: |row| < 2^24 where act
  // SM_ROUNDS x 16 entries of each keypoint per trip: all bucket records (index, x, y, octave) are requested together, then the
  // descriptors of the survivors of the band / disparity tests (the others read descriptor 0: one cached line), then the
  // distances -- two memory round trips for the four keypoints of the wave unless one of them has more than 64 entries
  for (int eb = e0; __ballot(eb < e1) != 0ull; eb += SM_ROUNDS * SM_G) {
    int4 rec[SM_ROUNDS];
    bool ok[SM_ROUNDS];
    float rx[SM_ROUNDS];
#pragma unroll
    for (int r = 0; r < SM_ROUNDS; r++) {
      const int e = eb + r * SM_G + sub;
      ok[r] = e < e1;
      rec[r] = bent[max(min(e, e1 - 1), 0)];
    }
    uint4 b0[SM_ROUNDS], b1[SM_ROUNDS];
#pragma unroll
    for (int r = 0; r < SM_ROUNDS; r++) {
      rx[r] = __int_as_float(rec[r].y);
      const float ry = __int_as_float(rec[r].z);
      const float rad = s_r[rec[r].w & (ORBFE_MAX_LEVELS - 1)];
      // minr = floor(ry - rad) <= row <= ceil(ry + rad) = maxr (L/src/Frame.cc:497-501) for the integer row:
      // ceil(t) >= row <=> t > row - 1, floor(t) <= row <=> t < row + 1, on the same rounded float sums
      ok[r] = ok[r] && (ry + rad > rowm1) && (ry - rad < rowp1) && (rx[r] >= minU && rx[r] <= maxU);
      load_desc(dr + (size_t)(ok[r] ? rec[r].x : 0) * 32, b0[r], b1[r]);
    }
#pragma unroll
    for (int r = 0; r < SM_ROUNDS; r++) {
      if (__ballot(ok[r]) != 0ull) {
        const unsigned key = ((unsigned)hamming256(a0, a1, b0[r], b1[r]) << 16) | (unsigned)rec[r].x;
        if (ok[r] && key < best) {   // (distance, index) minimum among distance < TH_HIGH (best starts at TH_HIGH << 16 | 0xffff)
          best = key;
          best_x = rx[r];
        }
      }
    }
  }
  // first minimum in index order; the winning lane also holds the right keypoint's x
  {
    const unsigned k = ((best >> 16) < (unsigned)ORBFE_TH_HIGH) ? best : 0xFFFFFFFFu;
    best = row_min_u32(k);
    best_x = __int_as_float((int)row_or_u32((k == best && k != 0xFFFFFFFFu) ? (unsigned)__float_as_int(best_x) : 0u));
  }
  const float uR0 = best_x;
  const int thOrbDist = (ORBFE_TH_HIGH + ORBFE_TH_LOW) / 2;
  act = act && best != 0xFFFFFFFFu && (int)(best >> 16) < thOrbDist;
  if (__ballot(act) == 0ull) return;

  // sub-pixel refinement by 11x11 SAD over 11 shifts (L/src/Frame.cc:557-631)
  const SmLevel lv = s_lv[levelL];
  const float sfac = lv.inv_scale;
  const float scaleduL = roundf(uL * sfac);
  const float scaledvL = roundf(vL * sfac);
  const float scaleduR0 = roundf(uR0 * sfac);
  const int w = 5, L = 5;
  const float iniu = scaleduR0 + L - w;
  const float endu = scaleduR0 + L + w + 1;
  act = act && !(iniu < 0 || endu >= (float)lv.wR);
  const int yW = (int)(scaledvL - w), xW = (int)(scaleduL - w), xRc = (int)scaleduR0;
  const int xR0 = xRc - L - w;
  // the reference reads these windows unchecked (cv::Mat::rowRange/colRange would throw); treat as no match
  act = act && !(yW < 0 || yW + 2 * w >= lv.hL || xW < 0 || xW + 2 * w >= lv.wL || xR0 < 0 || xRc + L + w >= lv.wR ||
                 yW + 2 * w >= lv.hR);
  if (__ballot(act) == 0ull) return;
  // stage the 11 x 11 left window and the 11 x 21 right window (all 11 shifts) of the row's keypoint in its LDS slice: lane r of
  // the row loads window row r as aligned 16-byte pieces -- two of the left level, three of the right one, from the piece that
  // holds the window's first column on -- five load instructions for the wave's four keypoints.  A piece is 16 bytes of a row
  // in a row-major level and one tile row in a tiled one (orbfe_level_offset): the same loads, another address.
  uint8_t* winL = &sad_win[grp][0];
  uint8_t* winR = winL + 11 * SAD_LP;
  const int axL = xW & ~15, axR = xR0 & ~15;
  if (act && sub < 11) {
    // (a pointer that comes out of LDS is a generic one to the compiler: say that it is global memory, or the loads are flat_load)
    typedef const __attribute__((address_space(1))) uint8_t* gptr_t;
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const int y = yW + sub;
    u32x4 vl[2], vr[3];
#pragma unroll
    for (int k = 0; k < 2; k++) {   // a piece beyond the row's last one is not part of the window: the last one again
      const int x = min(axL + 16 * k, lv.pitchL - 16);
      vl[k] = *reinterpret_cast<const __attribute__((address_space(1))) u32x4*>((gptr_t)(uintptr_t)lv.baseL + orbfe_level_offset(x, y, lv.pitchL, lv.tiledL != 0));
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const int x = min(axR + 16 * k, lv.pitchR - 16);
      vr[k] = *reinterpret_cast<const __attribute__((address_space(1))) u32x4*>((gptr_t)(uintptr_t)lv.baseR + orbfe_level_offset(x, y, lv.pitchR, lv.tiledR != 0));
    }
    u32x4* wl = reinterpret_cast<u32x4*>(winL + sub * SAD_LP);
    wl[0] = vl[0]; wl[1] = vl[1];
    u32x4* wr = reinterpret_cast<u32x4*>(winR + sub * SAD_RP);
    wr[0] = vr[0]; wr[1] = vr[1]; wr[2] = vr[2];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const uint8_t* WL = winL + (xW - axL);
  const uint8_t* WR = winR + (xR0 - axR);
  // |a - b| with a = IL - cL, b = IR - cR[k]: both biased by +255 so that v_sad_u32 (|x - y| + acc) applies
  const int cLm = (int)WL[w * SAD_LP + w] - 255;
  int cRm[11];
#pragma unroll
  for (int k = 0; k < 11; k++) cRm[k] = (int)WR[w * SAD_RP + w + k] - 255;
  unsigned acc[11];
#pragma unroll
  for (int k = 0; k < 11; k++) acc[k] = 0;
#pragma unroll
  for (int t = 0; t < SM_SAD_T; t++) {
    const int p = sub + SM_G * t;
    if (t < 7 || p < 121) {
      const unsigned a = (unsigned)((int)WL[s_offL[p]] - cLm);
      const uint8_t* rrow = WR + s_offR[p];
#pragma unroll
      for (int k = 0; k < 11; k++) acc[k] = sad_u32(a, (unsigned)((int)rrow[k] - cRm[k]), acc[k]);
    }
  }
  // every sum is <= 121 * 510 < 65536: reduce two per register inside the row; every lane of the row gets the totals
  unsigned pk[6];
#pragma unroll
  for (int j = 0; j < 5; j++) pk[j] = acc[2 * j] | (acc[2 * j + 1] << 16);
  pk[5] = acc[10];
#pragma unroll
  for (int j = 0; j < 6; j++) pk[j] = row_sum_u32(pk[j]);
  // lane k of the row takes the sum of shift k (k < 11).  The reference's `dist < bestDist` walk over float distances
  // (L/src/Frame.cc:593-600; the sums are integers < 2^16, exact as floats) is an integer first-minimum: the minimum of
  // (sum, k) keys over the row.
  unsigned mine = pk[0];
  {
    const int t = sub >> 1;
    mine = t == 1 ? pk[1] : mine; mine = t == 2 ? pk[2] : mine; mine = t == 3 ? pk[3] : mine;
    mine = t == 4 ? pk[4] : mine; mine = t == 5 ? pk[5] : mine;
    mine = (sub & 1) ? (mine >> 16) : (mine & 0xffffu);
  }
  const unsigned kmin = row_min_u32(sub < 11 ? ((mine << 4) | (unsigned)sub) : 0xFFFFFFFFu);
  const int bk = (int)(kmin & 15u);
  const int sadBest = (int)(kmin >> 4);
  const int bestincR = bk - L;
  // the neighbours of the minimum, from the lanes that hold them (for bk = 0 / 10 the lane read is outside 0..10: no match anyway)
  const int rowbase = lane & ~(SM_G - 1);
  const int i1 = __builtin_amdgcn_ds_bpermute((rowbase + ((bk + 15) & 15)) << 2, (int)mine);
  const int i3 = __builtin_amdgcn_ds_bpermute((rowbase + ((bk + 1) & 15)) << 2, (int)mine);
  const int i2 = sadBest;
  if (!act || sub != 0 || bestincR == -L || bestincR == L) return;
  const float dist1 = (float)i1, dist2 = (float)i2, dist3 = (float)i3;
  const float deltaR = (dist1 - dist3) / (2.0f * (dist1 + dist3 - 2.0f * dist2));
  if (deltaR < -1 || deltaR > 1) return;
  float bestuR = lv.scale * ((float)scaleduR0 + (float)bestincR + deltaR);
  float disparity = (uL - bestuR);
  if (disparity >= minD && disparity < maxD) {
    if (disparity <= 0) {
      disparity = (float)0.01;
      bestuR = (float)((double)uL - 0.01);
    }
    out_depth[iL] = P.mbf / disparity;
    out_ur[iL] = bestuR;
    out_sad[iL] = sadBest;
  }
}

// One block per stereo pair: the reference sorts (SAD, iL) pairs, takes the element at size/2 as the
// median and drops every match with SAD >= 1.5*1.4*median (L/src/Frame.cc:634-645).  The order statistic
// is found by a two-level 256-bin radix select (SAD <= 121*510 < 65536).
__global__ __launch_bounds__(256) void stereo_median_kernel(StereoParams P) {
  __shared__ int hist[256];
  __shared__ int sh[4];
  const int pair = blockIdx.x, tid = threadIdx.x;
  const int nL = P.nL[pair];
  float* out_ur = P.u_right + (size_t)pair * P.cap;
  float* out_depth = P.depth + (size_t)pair * P.cap;
  const int32_t* sad = P.sad + (size_t)pair * P.cap;
  hist[tid] = 0;
  if (tid == 0) sh[0] = 0;
  __syncthreads();
  int cntv = 0;
  for (int i = tid; i < nL; i += 256) {
    const int s = sad[i];
    if (s >= 0) { atomicAdd(&hist[(s >> 8) & 0xff], 1); cntv++; }
  }
  atomicAdd(&sh[0], cntv);
  __syncthreads();
  const int total = sh[0];
  if (total == 0) { if (tid == 0) P.n_matched[pair] = 0; return; }
  const int k = total / 2;  // 0-based rank
  if (tid == 0) {
    int run = 0, hb = 0;
    for (; hb < 256; hb++) { if (run + hist[hb] > k) break; run += hist[hb]; }
    sh[1] = hb;
    sh[2] = k - run;
  }
  __syncthreads();
  const int hb = sh[1], k2 = sh[2];
  __syncthreads();
  hist[tid] = 0;
  __syncthreads();
  for (int i = tid; i < nL; i += 256) {
    const int s = sad[i];
    if (s >= 0 && ((s >> 8) & 0xff) == hb) atomicAdd(&hist[s & 0xff], 1);
  }
  __syncthreads();
  if (tid == 0) {
    int run = 0, lb = 0;
    for (; lb < 256; lb++) { if (run + hist[lb] > k2) break; run += hist[lb]; }
    sh[3] = (hb << 8) | lb;
    sh[0] = 0;
  }
  __syncthreads();
  const float median = (float)sh[3];
  const float thDist = 1.5f * 1.4f * median;
  int kept = 0;
  for (int i = tid; i < nL; i += 256) {
    const int s = sad[i];
    if (s >= 0) {
      if ((float)s < thDist) kept++;
      else { out_ur[i] = -1; out_depth[i] = -1; }
    }
  }
  atomicAdd(&sh[0], kept);
  __syncthreads();
  if (tid == 0) P.n_matched[pair] = sh[0];
}

// ------------------------------------------------------------------------------------------------ launchers
// hipFuncSetAttribute applies to the device that is current when it is called: a process with matchers on several devices
// (orbfe_matcher_create takes a device) must raise a kernel's dynamic-LDS limit once on EACH of them.  One bit per device
// ordinal; two threads racing on the same device both set the same value.
static void raise_dynamic_lds(const void* kernel, int bytes, std::atomic<unsigned long long>& done) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  const unsigned long long bit = 1ull << (dev & 63);
  if (done.load(std::memory_order_acquire) & bit) return;
  (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  done.fetch_or(bit, std::memory_order_release);
}
void orbfe_launch_hamming_matrix(const uint8_t* A, int nA, const uint8_t* B, int nB, uint16_t* out, hipStream_t s) {
  if (nA < 1 || nB < 1) return;
  dim3 grid((nB + 255) / 256, (nA + 63) / 64);
  hipLaunchKernelGGL(hamming_matrix_kernel, grid, dim3(256), 0, s, A, nA, B, nB, out);
}
void orbfe_launch_hamming_bf(const HammingBfParams& p, int max_nA, int n_sets, hipStream_t s) {
  if (max_nA < 1 || n_sets < 1) return;
  dim3 grid((max_nA + 255) / 256, n_sets);
  hipLaunchKernelGGL(hamming_bf_kernel, grid, dim3(256), 0, s, p);
}
void orbfe_launch_grid_build(const FrameBatch& f, int n_frames, hipStream_t s) {
  if (f.cap > GB_LDS_CAP) {
    hipLaunchKernelGGL(grid_build_global_kernel, dim3(n_frames), dim3(GB_THREADS), 0, s, f);
    return;
  }
  static std::atomic<unsigned long long> raised{0};
  raise_dynamic_lds(reinterpret_cast<const void*>(grid_build_kernel), 6 * GB_LDS_CAP + 16, raised);
  hipLaunchKernelGGL(grid_build_kernel, dim3(n_frames), dim3(GB_THREADS), (size_t)6 * (size_t)f.cap + 16, s, f);
}
void orbfe_launch_proj_candidates(const FrameBatch& f, const QueryBatch& q, orbfe_cand* cand, int32_t* n_cand,
                                  int max_cand, int n_frames, hipStream_t s) {
  if (q.cap < 1) return;
  constexpr int G = ORBFE_CAND_LANES;       // lanes per query
  dim3 grid((q.cap + 4 * (64 / G) - 1) / (4 * (64 / G)), n_frames);
  hipLaunchKernelGGL(proj_candidates_kernel<G>, grid, dim3(256), 0, s, f, q, cand, n_cand, max_cand);
}
void orbfe_launch_proj_resolve(const FrameBatch& f, const QueryBatch& q, const orbfe_cand* cand, const int32_t* n_cand,
                               int max_cand, int mode, int th_high, float nnratio, int check_ori, uint8_t* blocked,
                               int32_t* assigned, int32_t* n_matches, int32_t* push_idx, uint8_t* push_bin, int n_frames,
                               hipStream_t s) {
  const size_t dyn = (size_t)(((f.cap + 15) & ~15) + 8 * (size_t)f.cap);
  // this kernel's static LDS alone is ~69 KiB: the limit is raised once per device, whichever thread launches first
  static std::atomic<unsigned long long> raised{0};
  raise_dynamic_lds(reinterpret_cast<const void*>(proj_resolve_kernel), 88 * 1024, raised);
  hipLaunchKernelGGL(proj_resolve_kernel, dim3(n_frames), dim3(RC_THREADS), dyn, s, f, q, cand, n_cand, max_cand, mode, th_high,
                     nnratio, check_ori, blocked, assigned, n_matches, push_idx, push_bin);
}
void orbfe_launch_bow(const BowParams& p, int n_pairs, int max_countB, hipStream_t s) {
  if (n_pairs > 0)
    hipLaunchKernelGGL(bow_match_kernel, dim3(p.sequential ? 1 : n_pairs), dim3(64), (size_t)((max_countB + 15) & ~15), s, p);
  hipLaunchKernelGGL(bow_finish_kernel, dim3(1), dim3(256), 0, s, p);
}
void orbfe_launch_proj_best(const FrameBatch& f, const QueryBatch& q, int gate, const float* inv_sigma2, int32_t* best_idx,
                            int32_t* best_dist, int n_frames, hipStream_t s) {
  if (q.cap < 1) return;
  dim3 grid((q.cap + 3) / 4, n_frames);
  if (gate == 2) hipLaunchKernelGGL(proj_best_kernel<2>, grid, dim3(256), 0, s, f, q, inv_sigma2, best_idx, best_dist);
  else if (gate == 1) hipLaunchKernelGGL(proj_best_kernel<1>, grid, dim3(256), 0, s, f, q, inv_sigma2, best_idx, best_dist);
  else hipLaunchKernelGGL(proj_best_kernel<0>, grid, dim3(256), 0, s, f, q, inv_sigma2, best_idx, best_dist);
}
void orbfe_launch_triangulation(const TriParams& p, int n_pairs, hipStream_t s) {
  if (n_pairs > 0) hipLaunchKernelGGL(triangulation_match_kernel, dim3(p.b.sequential ? 1 : n_pairs), dim3(64), 0, s, p);
  hipLaunchKernelGGL(bow_finish_kernel, dim3(1), dim3(256), 0, s, p.b);
}
void orbfe_launch_init_resolve(const FrameBatch& f, const QueryBatch& q, const orbfe_cand* cand, const int32_t* n_cand,
                               int max_cand, float nnratio, int check_ori, int32_t* matches12, float* prev_xy,
                               int32_t* n_matches, int32_t* push_idx, uint8_t* push_bin, hipStream_t s) {
  hipLaunchKernelGGL(init_resolve_kernel, dim3(1), dim3(64), (size_t)f.cap * 8, s, f, q, cand, n_cand, max_cand, nnratio,
                     check_ori, matches12, prev_xy, n_matches, push_idx, push_bin);
}
void orbfe_launch_stereo(const StereoParams& p, int n_pairs, hipStream_t s) {
  const size_t bucket_lds = sizeof(int) * (size_t)(2 * p.n_keys + 1);
  if (bucket_lds > 48 * 1024) {   // tall images with many levels only (KITTI: 3 KB)
    static std::atomic<unsigned long long> raised{0};
    raise_dynamic_lds(reinterpret_cast<const void*>(stereo_bucket_kernel), (int)(sizeof(int) * (2 * STEREO_MAX_KEYS + 1)), raised);
  }
  hipLaunchKernelGGL(stereo_bucket_kernel, dim3(n_pairs), dim3(256), bucket_lds, s, p);
  dim3 grid((p.cap + SM_KPB - 1) / SM_KPB, n_pairs);
  hipLaunchKernelGGL(stereo_match_kernel, grid, dim3(256), 0, s, p);
  hipLaunchKernelGGL(stereo_median_kernel, dim3(n_pairs), dim3(256), 0, s, p);
}
