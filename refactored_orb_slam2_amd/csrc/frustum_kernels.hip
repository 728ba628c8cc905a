// frustum_kernels.hip -- Frame::isInFrustum + MapPoint::PredictScale + the query SearchByProjection(F, vpMapPoints, th)
// forms from each tracked local map point (L/src/Frame.cc:284-339, L/src/MapPoint.cc:409-423, L/src/ORBmatcher.cc:52-71,
// driver loop L/src/Tracking.cc:1053-1067).  SURVEY 8(f) row 3: the queries are produced in HBM and consumed by
// proj_candidates / proj_resolve without a host pass.
#include <hip/hip_runtime.h>

#include "match_internal.h"

#define FQ_THREADS 256
#define WAVE 64
#define MP_DW (int)(sizeof(orbfe_map_point) / 4)   // 18
#define Q_DW (int)(sizeof(orbfe_query) / 4)        // 17
#define TR_DW (int)(sizeof(orbfe_track) / 4)       // 6
static_assert(sizeof(orbfe_last_point) == 60 && sizeof(orbfe_track_pose) == 160 && sizeof(orbfe_unproject_cam) == 64, "record layout");
static_assert(sizeof(orbfe_map_point) == 72 && sizeof(orbfe_query) == 68 && sizeof(orbfe_track) == 24, "record layout");

// glibc 2.35 logf (__logf_data, LOGF_TABLE_BITS = 4): {invc, logc} pairs.  Same constants as the oracle's oo_logf; the
// arithmetic below is the same double sequence (compiled with -ffp-contract=off).
__device__ const double g_logf_tab[32] = {
    0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2, 0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2,
    0x1.49539f0f010b0p+0, -0x1.01eae7f513a67p-2, 0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3,
    0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3, 0x1.25e227b0b8ea0p+0, -0x1.1aa2bc79c8100p-3,
    0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4, 0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4,
    0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5, 0x1.0000000000000p+0, 0x0.0p+0,
    0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5,  0x1.ca4b31f026aa0p-1, 0x1.c5e53aa362eb4p-4,
    0x1.b2036576afce6p-1, 0x1.526e57720db08p-3,  0x1.9c2d163a1aa2dp-1, 0x1.bc2860d224770p-3,
    0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2,  0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2};

// MapPoint::PredictScale.  Inputs logf does not map to a finite value (ratio 0, subnormal, inf, nan, negative) make the
// reference's float->int conversion produce INT_MIN on x86-64 (cvttss2si), hence level 0 after the clamp.
__device__ __forceinline__ int predict_scale(float max_distance, float dist, float log_scale_factor, int n_levels) {
  const float ratio = max_distance / dist;
  const uint32_t ix = __float_as_uint(ratio);
  float l;
  if (ix == 0x3f800000u) {
    l = 0.0f;
  } else if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {
    return 0;
  } else {
    const uint32_t tmp = ix - 0x3f330000u;
    const int i = (int)((tmp >> 19) & 15u);
    const int k = (int32_t)tmp >> 23;
    const uint32_t iz = ix - (tmp & (0x1ffu << 23));
    const double z = (double)__uint_as_float(iz);
    const double r = z * g_logf_tab[2 * i] - 1.0;
    const double y0 = g_logf_tab[2 * i + 1] + (double)k * 0x1.62e42fefa39efp-1;
    const double r2 = r * r;
    double y = 0x1.5575b0be00b6ap-2 * r + -0x1.ffffef20a4123p-2;
    y = -0x1.00ea348b88334p-2 * r2 + y;
    y = y * r2 + (y0 + r);
    l = (float)y;
  }
  const float c = ceilf(l / log_scale_factor);
  if (!(c >= -2147483648.0f && c < 2147483648.0f)) return 0;  // INT_MIN -> clamped to 0
  int nScale = (int)c;
  if (nScale < 0) nScale = 0;
  else if (nScale >= n_levels) nScale = n_levels - 1;
  return nScale;
}

// One thread per local map point; a block stages its 256 records through LDS so that the 72-byte inputs and the 68-byte
// queries move as coalesced dwords.
__global__ __launch_bounds__(FQ_THREADS) void frustum_queries_kernel(const orbfe_frustum* __restrict__ frustums,
                                                                     const orbfe_map_point* __restrict__ points,
                                                                     const int32_t* __restrict__ n_points, int p_cap, float th,
                                                                     float viewing_cos_limit, orbfe_track* __restrict__ track,
                                                                     orbfe_query* __restrict__ queries,
                                                                     int32_t* __restrict__ n_to_match) {
  __shared__ uint32_t rec[FQ_THREADS * MP_DW];  // map points in, queries out (in place: 17 of each thread's 18 dwords)
  __shared__ uint32_t trk[FQ_THREADS * TR_DW];
  __shared__ orbfe_frustum fr;
  const int f = blockIdx.y, tid = threadIdx.x;
  const int np = n_points[f];
  const int p0 = blockIdx.x * FQ_THREADS;
  if (p0 >= np) return;
  const int cnt = min(FQ_THREADS, np - p0);
  {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(points + (size_t)f * p_cap + p0);
    for (int i = tid; i < cnt * MP_DW; i += FQ_THREADS) rec[i] = src[i];
    const uint32_t* fs = reinterpret_cast<const uint32_t*>(frustums + f);
    if (tid < (int)(sizeof(orbfe_frustum) / 4)) reinterpret_cast<uint32_t*>(&fr)[tid] = fs[tid];
  }
  __syncthreads();
  bool in_view = false;
  if (tid < cnt) {
    orbfe_map_point mp;
    uint32_t* mpw = reinterpret_cast<uint32_t*>(&mp);
#pragma unroll
    for (int j = 0; j < MP_DW; j++) mpw[j] = rec[tid * MP_DW + j];
    orbfe_track tr;
    tr.in_view = 0; tr.proj_x = 0.f; tr.proj_y = 0.f; tr.proj_xr = 0.f; tr.level = 0; tr.view_cos = 0.f;
    if (!mp.skip) {   // Tracking.cc:1057-1060
      // ---- Frame::isInFrustum (Frame.cc:284-339); every early `return false` of the reference is a fall-through here
      float Pc[3];
#pragma unroll
      for (int r = 0; r < 3; r++) {
        const float t = fr.Rcw[3 * r] * mp.pos[0] + fr.Rcw[3 * r + 1] * mp.pos[1] + fr.Rcw[3 * r + 2] * mp.pos[2];
        Pc[r] = (float)((double)t * 1.0 + (double)fr.tcw[r] * 1.0);  // cv::gemm small-matrix path: float dot, double epilogue
      }
      if (!(Pc[2] < 0.0f)) {
        const float invz = 1.0f / Pc[2];
        const float u = fr.fx * Pc[0] * invz + fr.cx;
        const float v = fr.fy * Pc[1] * invz + fr.cy;
        if (!(u < fr.min_x || u > fr.max_x) && !(v < fr.min_y || v > fr.max_y)) {
          const float maxDistance = 1.2f * mp.max_distance, minDistance = 0.8f * mp.min_distance;
          float PO[3];
          double s = 0.0, dot = 0.0;
#pragma unroll
          for (int k = 0; k < 3; k++) {
            PO[k] = mp.pos[k] - fr.Ow[k];
            s += (double)PO[k] * (double)PO[k];          // cv::norm: double accumulation in element order
            dot += (double)PO[k] * (double)mp.normal[k]; // Mat::dot: the same
          }
          const float dist = (float)sqrt(s);
          if (!(dist < minDistance || dist > maxDistance)) {
            const float viewCos = (float)(dot / (double)dist);
            if (!(viewCos < viewing_cos_limit)) {
              tr.in_view = 1;
              tr.proj_x = u;
              tr.proj_xr = u - fr.mbf * invz;
              tr.proj_y = v;
              tr.level = predict_scale(mp.max_distance, dist, fr.log_scale_factor, fr.n_levels);
              tr.view_cos = viewCos;
              in_view = true;
            }
          }
        }
      }
    }
    // ---- the query of SearchByProjection(F, vpMapPoints, th) (ORBmatcher.cc:52-71)
    orbfe_query q;
    uint32_t* qw = reinterpret_cast<uint32_t*>(&q);
#pragma unroll
    for (int j = 0; j < Q_DW; j++) qw[j] = 0u;
    if (in_view) {
      float r = ((double)tr.view_cos > 0.998) ? 2.5f : 4.0f;  // RadiusByViewingCos (:130-135)
      if ((double)th != 1.0) r *= th;
      q.u = tr.proj_x; q.v = tr.proj_y; q.u_r = tr.proj_xr;
      q.radius = r * fr.scale_factors[tr.level];   // predict_scale clamps to n_levels - 1 < ORBFE_MAX_LEVELS
      q.min_level = tr.level - 1;
      q.max_level = tr.level;
      q.valid = 1;
      q.blocks = mp.observed != 0;
#pragma unroll
      for (int j = 0; j < 8; j++) reinterpret_cast<uint32_t*>(q.desc)[j] = reinterpret_cast<const uint32_t*>(mp.desc)[j];
    }
#pragma unroll
    for (int j = 0; j < Q_DW; j++) rec[tid * MP_DW + j] = qw[j];
    const uint32_t* tw = reinterpret_cast<const uint32_t*>(&tr);
#pragma unroll
    for (int j = 0; j < TR_DW; j++) trk[tid * TR_DW + j] = tw[j];
  }
  const unsigned long long bal = __ballot(in_view);
  if ((tid & (WAVE - 1)) == 0 && bal) atomicAdd(&n_to_match[f], __popcll(bal));
  __syncthreads();
  uint32_t* qdst = reinterpret_cast<uint32_t*>(queries + (size_t)f * p_cap + p0);
  for (int o = tid; o < cnt * Q_DW; o += FQ_THREADS) qdst[o] = rec[(o / Q_DW) * MP_DW + (o % Q_DW)];
  uint32_t* tdst = reinterpret_cast<uint32_t*>(track + (size_t)f * p_cap + p0);
  for (int o = tid; o < cnt * TR_DW; o += FQ_THREADS) tdst[o] = trk[o];
}

// ---- Frame::UnprojectStereo (L/src/Frame.cc:668-679) for every keypoint: thread per keypoint
__global__ __launch_bounds__(256) void unproject_stereo_kernel(const orbfe_keypoint* __restrict__ kps,
                                                               const uint8_t* __restrict__ desc,
                                                               const int32_t* __restrict__ n,
                                                               const float* __restrict__ depth, int cap,
                                                               const orbfe_unproject_cam* __restrict__ cams, int observed,
                                                               orbfe_last_point* __restrict__ points) {
  const int f = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  if (i >= cap) return;
  const size_t g = (size_t)f * cap + i;
  orbfe_last_point p;
  uint32_t* pw = reinterpret_cast<uint32_t*>(&p);
#pragma unroll
  for (int j = 0; j < (int)(sizeof(p) / 4); j++) pw[j] = 0u;
  if (i < n[f]) {
    const orbfe_unproject_cam c = cams[f];
    const orbfe_keypoint kp = kps[g];
    const float z = depth[g];
    if (z > 0) {
      const float x = (kp.x - c.cx) * z * c.invfx;
      const float y = (kp.y - c.cy) * z * c.invfy;
#pragma unroll
      for (int r = 0; r < 3; r++) {   // mRwc * x3Dc + mOw: cv::gemm small-matrix path (float dot, double epilogue)
        const float t = c.Rwc[3 * r] * x + c.Rwc[3 * r + 1] * y + c.Rwc[3 * r + 2] * z;
        p.pos[r] = (float)((double)t * 1.0 + (double)c.Ow[r] * 1.0);
      }
      p.valid = 1;
    }
    p.observed = observed != 0;
    p.octave = kp.octave;
    p.angle = kp.angle;
    const uint4* d = reinterpret_cast<const uint4*>(desc + g * 32);
    const uint4 d0 = d[0], d1 = d[1];
    uint32_t* dw = reinterpret_cast<uint32_t*>(p.desc);
    dw[0] = d0.x; dw[1] = d0.y; dw[2] = d0.z; dw[3] = d0.w; dw[4] = d1.x; dw[5] = d1.y; dw[6] = d1.z; dw[7] = d1.w;
  }
  uint32_t* out = reinterpret_cast<uint32_t*>(points + g);
#pragma unroll
  for (int j = 0; j < (int)(sizeof(p) / 4); j++) out[j] = pw[j];
}

// ---- the projection part of SearchByProjection(cur, last) (L/src/ORBmatcher.cc:1270-1308): thread per last-frame point
__global__ __launch_bounds__(256) void track_queries_kernel(const orbfe_track_pose* __restrict__ poses,
                                                            const orbfe_last_point* __restrict__ points,
                                                            const int32_t* __restrict__ n_points, int p_cap, int n_frames,
                                                            int frame_shift, orbfe_query* __restrict__ queries,
                                                            int32_t* __restrict__ nq) {
  constexpr int LP_DW = (int)(sizeof(orbfe_last_point) / 4);  // 15
  __shared__ uint32_t rec[256 * Q_DW];  // the block's 256 point records in (15 dwords each), its queries out (17 each):
                                        // both move as coalesced dwords instead of 60- / 68-byte strided accesses
  const int f = blockIdx.y, tid = threadIdx.x, p0 = blockIdx.x * 256;
  int fs = (f - frame_shift) % n_frames;
  if (fs < 0) fs += n_frames;
  const int np = n_points[fs];
  if (p0 == 0 && tid == 0) nq[f] = np;
  const int cnt = min(256, p_cap - p0);        // query slots this block owns
  const int have = max(0, min(cnt, np - p0));  // of which backed by a point
  {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(points + (size_t)fs * p_cap + p0);
    for (int i = tid; i < have * LP_DW; i += 256) rec[i] = src[i];
  }
  __syncthreads();
  orbfe_query q;
  uint32_t* qw = reinterpret_cast<uint32_t*>(&q);
#pragma unroll
  for (int j = 0; j < Q_DW; j++) qw[j] = 0u;
  if (tid < have) {
    const orbfe_track_pose& P = poses[f];   // (by reference: see track_queries_stereo_kernel)
    orbfe_last_point lp;
    uint32_t* lw = reinterpret_cast<uint32_t*>(&lp);
#pragma unroll
    for (int j = 0; j < LP_DW; j++) lw[j] = rec[tid * LP_DW + j];
    if (lp.valid) {
      float xc3[3];
#pragma unroll
      for (int r = 0; r < 3; r++) {   // Rcw * x3Dw + tcw
        const float t = P.Rcw[3 * r] * lp.pos[0] + P.Rcw[3 * r + 1] * lp.pos[1] + P.Rcw[3 * r + 2] * lp.pos[2];
        xc3[r] = (float)((double)t * 1.0 + (double)P.tcw[r] * 1.0);
      }
      const float invzc = (float)(1.0 / (double)xc3[2]);   // `1.0 / x3Dc.at<float>(2)`: double division (:1283)
      if (!(invzc < 0)) {
        const float u = P.fx * xc3[0] * invzc + P.cx;
        const float v = P.fy * xc3[1] * invzc + P.cy;
        if (!(u < P.min_x || u > P.max_x) && !(v < P.min_y || v > P.max_y)) {
          const int oct = lp.octave;
          q.u = u; q.v = v;
          q.u_r = u - P.mbf * invzc;                                   // :1327
          q.radius = P.th * P.scale_factors[oct & (ORBFE_MAX_LEVELS - 1)];   // :1297 (the mask only guards memory: octave < nLevels <= 16)
          if (P.forward) { q.min_level = oct; q.max_level = -1; }      // GetFeaturesInArea(u, v, radius, nLastOctave)
          else if (P.backward) { q.min_level = 0; q.max_level = oct; }
          else { q.min_level = oct - 1; q.max_level = oct + 1; }
          q.valid = 1;
          q.blocks = lp.observed != 0;
          q.angle = lp.angle;
#pragma unroll
          for (int j = 0; j < 8; j++) reinterpret_cast<uint32_t*>(q.desc)[j] = reinterpret_cast<const uint32_t*>(lp.desc)[j];
        }
      }
    }
  }
  __syncthreads();   // every thread has read its point record
#pragma unroll
  for (int j = 0; j < Q_DW; j++) rec[tid * Q_DW + j] = qw[j];
  __syncthreads();
  uint32_t* out = reinterpret_cast<uint32_t*>(queries + (size_t)f * p_cap + p0);
  for (int i = tid; i < cnt * Q_DW; i += 256) out[i] = rec[i];
}

// ---- UnprojectStereo + the projection of SearchByProjection(cur, last) in one pass (L/src/Frame.cc:668-679, L/src/ORBmatcher.cc:1270-1308):
// thread per keypoint of the SOURCE frame.  The two kernels above hand a 60-byte point record per keypoint through HBM (written, read
// once, never used again when the points only feed the next frame's search): here the world point stays in registers -- the same float
// expressions in the same order, so the queries are byte-equal to unproject_stereo_kernel -> track_queries_kernel.  Source of frame f:
// frame f - frame_shift of the batch; in front of the batch the carry frame (`carry` != 0: the last frame of the batch before) or, without
// one, the batch's own tail (mod n_frames: what track_queries_kernel does).
#ifndef TQS_T
#define TQS_T 256   // keypoints (threads) per workgroup
#endif
struct TqsFrame {   // one source frame's arrays
  const orbfe_keypoint* kps; const uint8_t* desc; const int32_t* n; const float* depth; const orbfe_unproject_cam* cam;
};
__global__ __launch_bounds__(TQS_T) void track_queries_stereo_kernel(TqsFrame B, TqsFrame Cy, int carry, int cap, int n_frames, int frame_shift,
                                                                   int observed, const orbfe_track_pose* __restrict__ poses,
                                                                   orbfe_query* __restrict__ queries, int32_t* __restrict__ nq) {
  __shared__ uint32_t rec[TQS_T * Q_DW];   // the block's queries leave as coalesced dwords instead of 68-byte strided stores
  const int f = blockIdx.y, tid = threadIdx.x, p0 = blockIdx.x * TQS_T;
  int fs = f - frame_shift;
  const bool from_carry = fs < 0 && carry;
  if (!from_carry) { fs %= n_frames; if (fs < 0) fs += n_frames; }
  const TqsFrame S = from_carry ? Cy : B;
  if (from_carry) fs = 0;
  const int np = S.n[fs];
  if (p0 == 0 && tid == 0) nq[f] = np;
  const int cnt = min(TQS_T, cap - p0);          // query slots this block owns
  const int have = max(0, min(cnt, np - p0));  // of which backed by a keypoint
  orbfe_query q;
  uint32_t* qw = reinterpret_cast<uint32_t*>(&q);
#pragma unroll
  for (int j = 0; j < Q_DW; j++) qw[j] = 0u;
  if (tid < have) {
    const size_t g = (size_t)fs * cap + p0 + tid;
    const float z = S.depth[g];
    const orbfe_keypoint kp = S.kps[g];
    const uint4* d = reinterpret_cast<const uint4*>(S.desc + g * 32);
    const uint4 d0 = d[0], d1 = d[1];
    if (z > 0) {
      const orbfe_unproject_cam c = S.cam[fs];
      // (read through the pointer: a by-value copy indexed with the per-lane octave below lives in scratch memory -- 164 bytes written
      //  and read back per thread, the kernel's whole time in rounds 3-5)
      const orbfe_track_pose& P = poses[f];
      const float x = (kp.x - c.cx) * z * c.invfx;
      const float y = (kp.y - c.cy) * z * c.invfy;
      float pos[3], xc3[3];
#pragma unroll
      for (int r = 0; r < 3; r++) {   // mRwc * x3Dc + mOw: cv::gemm small-matrix path (float dot, double epilogue)
        const float t = c.Rwc[3 * r] * x + c.Rwc[3 * r + 1] * y + c.Rwc[3 * r + 2] * z;
        pos[r] = (float)((double)t * 1.0 + (double)c.Ow[r] * 1.0);
      }
#pragma unroll
      for (int r = 0; r < 3; r++) {   // Rcw * x3Dw + tcw
        const float t = P.Rcw[3 * r] * pos[0] + P.Rcw[3 * r + 1] * pos[1] + P.Rcw[3 * r + 2] * pos[2];
        xc3[r] = (float)((double)t * 1.0 + (double)P.tcw[r] * 1.0);
      }
      const float invzc = (float)(1.0 / (double)xc3[2]);   // `1.0 / x3Dc.at<float>(2)`: double division (:1283)
      if (!(invzc < 0)) {
        const float u = P.fx * xc3[0] * invzc + P.cx;
        const float v = P.fy * xc3[1] * invzc + P.cy;
        if (!(u < P.min_x || u > P.max_x) && !(v < P.min_y || v > P.max_y)) {
          const int oct = kp.octave;
          q.u = u; q.v = v;
          q.u_r = u - P.mbf * invzc;                                   // :1327
          q.radius = P.th * P.scale_factors[oct & (ORBFE_MAX_LEVELS - 1)];   // :1297
          if (P.forward) { q.min_level = oct; q.max_level = -1; }
          else if (P.backward) { q.min_level = 0; q.max_level = oct; }
          else { q.min_level = oct - 1; q.max_level = oct + 1; }
          q.valid = 1;
          q.blocks = observed != 0;
          q.angle = kp.angle;
          uint32_t* dw = reinterpret_cast<uint32_t*>(q.desc);
          dw[0] = d0.x; dw[1] = d0.y; dw[2] = d0.z; dw[3] = d0.w; dw[4] = d1.x; dw[5] = d1.y; dw[6] = d1.z; dw[7] = d1.w;
        }
      }
    }
  }
  // (68-byte strided stores straight from the registers: 45 us against 37.5 through LDS, before the scratch fix of lesson 58)
#pragma unroll
  for (int j = 0; j < Q_DW; j++) rec[tid * Q_DW + j] = qw[j];
  __syncthreads();
  uint32_t* out = reinterpret_cast<uint32_t*>(queries + (size_t)f * cap + p0);
  for (int i = tid; i < cnt * Q_DW; i += TQS_T) out[i] = rec[i];
}

// ---- projection prologue of Fuse / Fuse(Sim3) / SearchBySim3 / SearchByProjection(KF,Scw) / SearchByProjection(Frame,KF,...)
// (L/src/ORBmatcher.cc:785-816, 925-977, 1075-1113 & 1155-1192, 296-345, 1406-1437): thread per candidate map point.  Every
// float expression is written in the operation order of the mode's reference lines (`1 / z` is a float division, `1.0 / z` a
// double one; Fuse multiplies X * invz first, the relocalisation search fx * xc first); cv::Mat products are the small-matrix
// gemm (float dot, double alpha/beta epilogue), cv::norm / Mat::dot accumulate in double -- as in frustum_queries_kernel.
static_assert(sizeof(orbfe_kf_camera) == 220 && sizeof(orbfe_kf_point) == 72 && sizeof(orbfe_kf_result) == 24, "record layout");
__global__ __launch_bounds__(256) void kf_queries_kernel(const orbfe_kf_camera* __restrict__ camp, const orbfe_kf_point* __restrict__ points,
                                                         int n, int mode, orbfe_query* __restrict__ queries,
                                                         orbfe_kf_result* __restrict__ results) {
  __shared__ orbfe_kf_camera cam;
  if (threadIdx.x < (int)(sizeof(orbfe_kf_camera) / 4))
    reinterpret_cast<uint32_t*>(&cam)[threadIdx.x] = reinterpret_cast<const uint32_t*>(camp)[threadIdx.x];
  __syncthreads();
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const orbfe_kf_point mp = points[i];
  orbfe_query q;
  uint32_t* qw = reinterpret_cast<uint32_t*>(&q);
#pragma unroll
  for (int j = 0; j < Q_DW; j++) qw[j] = 0u;
  orbfe_kf_result res;
  res.best_idx = -1; res.best_dist = 256; res.level = -1; res.u = 0.f; res.v = 0.f; res.u_r = 0.f;
  if (!mp.skip) {
    float Pc[3];
#pragma unroll
    for (int r = 0; r < 3; r++) {
      const float t = cam.R[3 * r] * mp.pos[0] + cam.R[3 * r + 1] * mp.pos[1] + cam.R[3 * r + 2] * mp.pos[2];
      Pc[r] = (float)((double)t * 1.0 + (double)cam.t[r] * 1.0);
    }
    if (mode == ORBFE_KF_SIM3) {   // p3Dc2 = sR21 * p3Dc1 + t21 (:1075) / p3Dc1 = sR12 * p3Dc2 + t12 (:1155)
      float P2[3];
#pragma unroll
      for (int r = 0; r < 3; r++) {
        const float t = cam.R2[3 * r] * Pc[0] + cam.R2[3 * r + 1] * Pc[1] + cam.R2[3 * r + 2] * Pc[2];
        P2[r] = (float)((double)t * 1.0 + (double)cam.t2[r] * 1.0);
      }
      Pc[0] = P2[0]; Pc[1] = P2[1]; Pc[2] = P2[2];
    }
    bool ok = mode == ORBFE_KF_RELOC || !(Pc[2] < 0.0f);   // "Depth must be positive"; the relocalisation search has no such test
    float invz = 0.f, u = 0.f, v = 0.f;
    if (ok) {
      if (mode == ORBFE_KF_FUSE || mode == ORBFE_KF_LOOP) invz = 1.0f / Pc[2];        // `1 / z`   (:797, :317)
      else invz = (float)(1.0 / (double)Pc[2]);                                        // `1.0 / z` (:945, :1083, :1163, :1413)
      if (mode == ORBFE_KF_RELOC) {                                                    // fx * xc * invzc + cx (:1415-1416)
        u = cam.fx * Pc[0] * invz + cam.cx;
        v = cam.fy * Pc[1] * invz + cam.cy;
        ok = !(u < cam.min_x || u > cam.max_x) && !(v < cam.min_y || v > cam.max_y);   // :1418-1421
      } else {                                                                         // x = X * invz; u = fx * x + cx
        const float x = Pc[0] * invz, y = Pc[1] * invz;
        u = cam.fx * x + cam.cx;
        v = cam.fy * y + cam.cy;
        ok = u >= cam.min_x && u < cam.max_x && v >= cam.min_y && v < cam.max_y;       // KeyFrame::IsInImage (KeyFrame.cc:569-571)
      }
    }
    if (ok) {
      const float maxDistance = 1.2f * mp.max_distance, minDistance = 0.8f * mp.min_distance;   // Get{Max,Min}DistanceInvariance
      double s = 0.0, dot = 0.0;
      if (mode == ORBFE_KF_SIM3) {
#pragma unroll
        for (int k = 0; k < 3; k++) s += (double)Pc[k] * (double)Pc[k];   // cv::norm(p3Dc2) (:1097)
      } else {
#pragma unroll
        for (int k = 0; k < 3; k++) {
          const float PO = mp.pos[k] - cam.Ow[k];
          s += (double)PO * (double)PO;
          dot += (double)PO * (double)mp.normal[k];
        }
      }
      const float dist3D = (float)sqrt(s);
      ok = !(dist3D < minDistance || dist3D > maxDistance);
      if (ok && (mode == ORBFE_KF_FUSE || mode == ORBFE_KF_FUSE_SIM3 || mode == ORBFE_KF_LOOP))
        ok = !(dot < 0.5 * (double)dist3D);   // "Viewing angle must be less than 60 deg": PO.dot(Pn) < 0.5 * dist
      if (ok) {
        const int level = predict_scale(mp.max_distance, dist3D, cam.log_scale_factor, cam.n_levels);
        q.u = u; q.v = v;
        q.u_r = u - cam.mbf * invz;   // Fuse :808
        q.radius = cam.th * cam.scale_factors[level];
        q.min_level = level - 1;
        q.max_level = mode == ORBFE_KF_RELOC ? level + 1 : level;
        q.valid = 1;
        q.blocks = 1;
        q.angle = mp.angle;
#pragma unroll
        for (int j = 0; j < 8; j++) reinterpret_cast<uint32_t*>(q.desc)[j] = reinterpret_cast<const uint32_t*>(mp.desc)[j];
        res.level = level; res.u = u; res.v = v; res.u_r = q.u_r;
      }
    }
  }
  queries[i] = q;
  results[i] = res;
}

void orbfe_launch_kf_queries(const orbfe_kf_camera* cam, const orbfe_kf_point* points, int n, int mode, orbfe_query* queries,
                             orbfe_kf_result* results, hipStream_t s) {
  if (n < 1) return;
  hipLaunchKernelGGL(kf_queries_kernel, dim3((n + 255) / 256), dim3(256), 0, s, cam, points, n, mode, queries, results);
}

void orbfe_launch_unproject_stereo(const orbfe_keypoint* kps, const uint8_t* desc, const int32_t* n, const float* depth, int cap,
                                   const orbfe_unproject_cam* cams, int observed, orbfe_last_point* points, int n_frames,
                                   hipStream_t s) {
  hipLaunchKernelGGL(unproject_stereo_kernel, dim3((cap + 255) / 256, n_frames), dim3(256), 0, s, kps, desc, n, depth, cap, cams,
                     observed, points);
}
void orbfe_launch_track_queries(const orbfe_track_pose* poses, const orbfe_last_point* points, const int32_t* n_points, int p_cap,
                                int frame_shift, orbfe_query* queries, int32_t* nq, int n_frames, hipStream_t s) {
  hipLaunchKernelGGL(track_queries_kernel, dim3((p_cap + 255) / 256, n_frames), dim3(256), 0, s, poses, points, n_points, p_cap,
                     n_frames, frame_shift, queries, nq);
}

void orbfe_launch_track_queries_stereo(const orbfe_keypoint* kps, const uint8_t* desc, const int32_t* n, const float* depth, int cap,
                                       const orbfe_unproject_cam* cams, int observed, const orbfe_keypoint* c_kps, const uint8_t* c_desc,
                                       const int32_t* c_n, const float* c_depth, const orbfe_unproject_cam* c_cam,
                                       const orbfe_track_pose* poses, int frame_shift, orbfe_query* queries, int32_t* nq, int n_frames,
                                       hipStream_t s) {
  const TqsFrame B{kps, desc, n, depth, cams}, Cy{c_kps, c_desc, c_n, c_depth, c_cam};
  hipLaunchKernelGGL(track_queries_stereo_kernel, dim3((cap + TQS_T - 1) / TQS_T, n_frames), dim3(TQS_T), 0, s, B, Cy, c_kps ? 1 : 0, cap, n_frames,
                     frame_shift, observed, poses, queries, nq);
}

void orbfe_launch_frustum_queries(const orbfe_frustum* frustums, const orbfe_map_point* points, const int32_t* n_points,
                                  int p_cap, float th, float viewing_cos_limit, orbfe_track* track, orbfe_query* queries,
                                  int32_t* n_to_match, int n_frames, hipStream_t s) {
  if (p_cap < 1 || n_frames < 1) return;
  dim3 grid((p_cap + FQ_THREADS - 1) / FQ_THREADS, n_frames);
  hipLaunchKernelGGL(frustum_queries_kernel, grid, dim3(FQ_THREADS), 0, s, frustums, points, n_points, p_cap, th,
                     viewing_cos_limit, track, queries, n_to_match);
}
