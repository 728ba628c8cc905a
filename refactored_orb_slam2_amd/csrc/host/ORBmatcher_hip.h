// ORBmatcher_hip.h -- host-side adapters that put ORB_SLAM2::ORBmatcher's per-frame searches on liborbfe.
//
// The reference's ORBmatcher methods take SLAM objects (Frame, MapPoint, KeyFrame:
// Source/Libraries/ORB_SLAM2/include/ORBmatcher.h:34-114).  These templates marshal the members those
// methods read into the POD views of include/orbfe.h, run the search on the GPU and write the result back
// exactly where the reference writes it (F.mvpMapPoints).  They are templates over the Frame / MapPoint
// types so that they compile (and are unit-tested) without the reference tree; INTEGRATION.md shows the
// three method bodies a maintainer replaces in ORBmatcher.cc with calls to them.
//
// Members used (same names as the reference):
//   Frame:    N, mvKeysUn, mDescriptors, mvuRight, mvpMapPoints, mvScaleFactors, mnMinX/mnMaxX/mnMinY/mnMaxY
//   MapPoint: mbTrackInView, mTrackProjX, mTrackProjY, mTrackProjXR, mnTrackScaleLevel, mTrackViewCos,
//             isBad(), Observations(), GetDescriptor()
#pragma once
#include <stdio.h>
#include <string.h>

#include <vector>

#include "../../../include/orbfe.h"

namespace ORB_SLAM2 {
namespace orbfe_host {

// ORBmatcher::DescriptorDistance for ONE pair (L/src/ORBmatcher.cc:1542-1556): kept on the host for the
// scalar callers outside the hot path (Frame.cc:547, MapPoint.cc:269,295).  Every batched use goes through
// orbfe_hamming_matrix_device / orbfe_hamming_bf_device / the projection searches.
inline int DescriptorDistance(const uint8_t* a, const uint8_t* b) {
  int dist = 0;
  for (int i = 0; i < 8; i++) {
    uint32_t x, y;
    memcpy(&x, a + 4 * i, 4);
    memcpy(&y, b + 4 * i, 4);
    dist += __builtin_popcount(x ^ y);
  }
  return dist;
}

template <class FrameT>
inline orbfe_frame_view MakeFrameView(const FrameT& F) {
  static_assert(sizeof(F.mvKeysUn[0]) == sizeof(orbfe_keypoint), "cv::KeyPoint layout");
  orbfe_frame_view v;
  v.n = F.N;
  v.keys_un = reinterpret_cast<const orbfe_keypoint*>(F.mvKeysUn.data());
  v.desc = F.mDescriptors.ptr(0);  // N x 32, continuous (created by ORBextractor::operator())
  v.u_right = F.mvuRight.empty() ? nullptr : F.mvuRight.data();
  v.min_x = F.mnMinX; v.max_x = F.mnMaxX; v.min_y = F.mnMinY; v.max_y = F.mnMaxY;
  return v;
}

// blocked[idx] != 0  <=>  F.mvpMapPoints[idx] && F.mvpMapPoints[idx]->Observations() > 0
template <class FrameT>
inline std::vector<uint8_t> BlockedFromFrame(const FrameT& F) {
  std::vector<uint8_t> b((size_t)F.N, 0);
  for (int i = 0; i < F.N; i++)
    if (F.mvpMapPoints[i] && F.mvpMapPoints[i]->Observations() > 0) b[i] = 1;
  return b;
}

inline float RadiusByViewingCos(float viewCos) { return viewCos > 0.998 ? 2.5f : 4.0f; }  // :130-135

// SearchByProjection(Frame&, const vector<MapPoint*>&, th)   L/src/ORBmatcher.cc:45-128
template <class FrameT, class MapPointT>
int SearchByProjectionPoints(FrameT& F, const std::vector<MapPointT*>& vpMapPoints, float th, float nnratio) {
  const bool bFactor = th != 1.0;
  std::vector<orbfe_query> q(vpMapPoints.size());
  for (size_t i = 0; i < vpMapPoints.size(); i++) {
    MapPointT* pMP = vpMapPoints[i];
    orbfe_query& e = q[i];
    memset(&e, 0, sizeof(e));
    if (!pMP->mbTrackInView || pMP->isBad()) continue;  // valid = 0
    const int level = pMP->mnTrackScaleLevel;
    float r = RadiusByViewingCos(pMP->mTrackViewCos);
    if (bFactor) r *= th;
    e.u = pMP->mTrackProjX;
    e.v = pMP->mTrackProjY;
    e.u_r = pMP->mTrackProjXR;
    e.radius = r * F.mvScaleFactors[level];
    e.min_level = level - 1;
    e.max_level = level;
    e.valid = 1;
    e.blocks = pMP->Observations() > 0;
    const auto d = pMP->GetDescriptor();
    memcpy(e.desc, d.ptr(0), 32);
  }
  std::vector<uint8_t> blocked = BlockedFromFrame(F);
  std::vector<int32_t> assigned((size_t)F.N, -1);
  const orbfe_frame_view v = MakeFrameView(F);
  int nmatches = 0;
  const int rc = orbfe_search_by_projection_points(&v, q.data(), (int)q.size(), nnratio, blocked.data(),
                                                   assigned.data(), &nmatches);
  if (rc != ORBFE_OK) {
    fprintf(stderr, "ORBmatcher::SearchByProjection: liborbfe error %d: %s\n", rc, orbfe_last_error());
    return 0;
  }
  for (int i = 0; i < F.N; i++)
    if (assigned[i] >= 0) F.mvpMapPoints[i] = vpMapPoints[(size_t)assigned[i]];  // :122
  return nmatches;
}

// SearchByProjection(Frame& cur, const Frame& last, th, bMono)   L/src/ORBmatcher.cc:1247-1383
// `queries[i]` is filled by the caller from LastFrame point i exactly as the reference projects it
// (:1276-1308): u, v, u_r = u - mbf*invzc, radius = th*scale[octave], the forward/backward level range,
// angle = LastFrame.mvKeysUn[i].angle, desc = pMP->GetDescriptor(), blocks = Observations() > 0, valid = 0
// when the reference `continue`s.  Writes CurrentFrame.mvpMapPoints and returns nmatches.
template <class FrameT>
int SearchByProjectionFrame(FrameT& CurrentFrame, const FrameT& LastFrame, const std::vector<orbfe_query>& queries,
                            bool checkOrientation) {
  std::vector<uint8_t> blocked = BlockedFromFrame(CurrentFrame);
  std::vector<int32_t> assigned((size_t)CurrentFrame.N, -2);  // -2 = untouched, -1 = removed by the rotation check
  const orbfe_frame_view v = MakeFrameView(CurrentFrame);
  int nmatches = 0;
  const int rc = orbfe_search_by_projection_frame(&v, queries.data(), (int)queries.size(), checkOrientation ? 1 : 0,
                                                  blocked.data(), assigned.data(), &nmatches);
  if (rc != ORBFE_OK) {
    fprintf(stderr, "ORBmatcher::SearchByProjection: liborbfe error %d: %s\n", rc, orbfe_last_error());
    return 0;
  }
  for (int i = 0; i < CurrentFrame.N; i++) {
    if (assigned[i] >= 0) CurrentFrame.mvpMapPoints[i] = LastFrame.mvpMapPoints[(size_t)assigned[i]];  // :1344
    else if (assigned[i] == -1) CurrentFrame.mvpMapPoints[i] = nullptr;                                  // :1374
  }
  return nmatches;
}

}  // namespace orbfe_host
}  // namespace ORB_SLAM2
