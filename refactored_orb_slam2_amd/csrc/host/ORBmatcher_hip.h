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
//             ComputeStereoMatches additionally -- mvKeys, mvKeysRight, mDescriptorsRight, mbf, mb, mvDepth
//   MapPoint: mbTrackInView, mTrackProjX, mTrackProjY, mTrackProjXR, mnTrackScaleLevel, mTrackViewCos,
//             isBad(), Observations(), GetDescriptor()
//   SearchLocalPoints additionally -- Frame: mnId, mTcw, GetCameraCenter(), fx, fy, cx, cy, mbf, mfLogScaleFactor, mnScaleLevels;
//             MapPoint: mnLastFrameSeen, GetWorldPos(), GetNormal(), IncreaseVisible(), and the protected mfMinDistance /
//             mfMaxDistance, reached through a pointer to member formed in a derived class (no change to MapPoint.h)
#pragma once
#include <stdio.h>
#include <string.h>

#include <utility>
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

// Row r of a CV_32F cv::Mat with one column (3x1) or three (3x3)
template <class MatT>
inline const float* FloatRow(const MatT& M, int r) { return reinterpret_cast<const float*>(M.ptr(r)); }

// MapPoint::mfMinDistance / mfMaxDistance are protected and only reachable scaled (GetM??DistanceInvariance), while
// MapPoint::PredictScale divides the UN-scaled value: a pointer to member formed inside a derived class reads them without
// touching MapPoint.h.  The struct is never instantiated.
template <class MapPointT>
struct ProtectedDistances : public MapPointT {
  static float MapPointT::*Min() { return &ProtectedDistances::mfMinDistance; }
  static float MapPointT::*Max() { return &ProtectedDistances::mfMaxDistance; }
};

// Second half of Tracking::SearchLocalPoints (L/src/Tracking.cc:1050-1078): replaces the isInFrustum loop and the
// SearchByProjection(mCurrentFrame, mvpLocalMapPoints, th) call.  Projection, frustum tests, scale prediction, window
// search and assignment all run on the GPU; the MapPoint fields isInFrustum writes, IncreaseVisible() and
// F.mvpMapPoints are updated here from the results.  Returns nmatches; *nToMatch as the reference counts it.
template <class FrameT, class MapPointT>
int SearchLocalPoints(FrameT& F, const std::vector<MapPointT*>& vpLocalMapPoints, float th, float nnratio,
                      int* nToMatch = nullptr) {
  const size_t n = vpLocalMapPoints.size();
  orbfe_frustum fr;
  memset(&fr, 0, sizeof(fr));
  const auto Ow = F.GetCameraCenter();   // mOw (mRcw / mtcw are private: read them out of mTcw, Frame.cc:274-279)
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) fr.Rcw[3 * r + c] = FloatRow(F.mTcw, r)[c];
    fr.tcw[r] = FloatRow(F.mTcw, r)[3];
    fr.Ow[r] = FloatRow(Ow, r)[0];
  }
  fr.fx = F.fx; fr.fy = F.fy; fr.cx = F.cx; fr.cy = F.cy; fr.mbf = F.mbf;
  fr.min_x = F.mnMinX; fr.max_x = F.mnMaxX; fr.min_y = F.mnMinY; fr.max_y = F.mnMaxY;
  fr.log_scale_factor = F.mfLogScaleFactor;
  fr.n_levels = F.mnScaleLevels;
  for (int l = 0; l < F.mnScaleLevels && l < ORBFE_MAX_LEVELS; l++) fr.scale_factors[l] = F.mvScaleFactors[l];
  std::vector<orbfe_map_point> mp(n);
  for (size_t i = 0; i < n; i++) {
    MapPointT* pMP = vpLocalMapPoints[i];
    orbfe_map_point& e = mp[i];
    memset(&e, 0, sizeof(e));
    e.skip = (pMP->mnLastFrameSeen == F.mnId) || pMP->isBad();  // :1057-1060
    if (e.skip) continue;
    const auto P = pMP->GetWorldPos();
    const auto Pn = pMP->GetNormal();
    for (int r = 0; r < 3; r++) { e.pos[r] = FloatRow(P, r)[0]; e.normal[r] = FloatRow(Pn, r)[0]; }
    e.min_distance = pMP->*ProtectedDistances<MapPointT>::Min();
    e.max_distance = pMP->*ProtectedDistances<MapPointT>::Max();
    e.observed = pMP->Observations() > 0;
    const auto d = pMP->GetDescriptor();
    memcpy(e.desc, d.ptr(0), 32);
  }
  std::vector<orbfe_track> track(n);
  std::vector<uint8_t> blocked = BlockedFromFrame(F);
  std::vector<int32_t> assigned((size_t)F.N, -1);
  const orbfe_frame_view v = MakeFrameView(F);
  int ntm = 0, nmatches = 0;
  const int rc = orbfe_search_local_points(&v, &fr, mp.data(), (int)n, th, nnratio, track.data(), blocked.data(),
                                           assigned.data(), &ntm, &nmatches);
  if (rc != ORBFE_OK) {
    fprintf(stderr, "Tracking::SearchLocalPoints: liborbfe error %d: %s\n", rc, orbfe_last_error());
    return 0;
  }
  for (size_t i = 0; i < n; i++) {
    if (mp[i].skip) continue;
    MapPointT* pMP = vpLocalMapPoints[i];
    pMP->mbTrackInView = track[i].in_view != 0;       // Frame.cc:285,329
    if (!track[i].in_view) continue;
    pMP->mTrackProjX = track[i].proj_x;
    pMP->mTrackProjXR = track[i].proj_xr;
    pMP->mTrackProjY = track[i].proj_y;
    pMP->mnTrackScaleLevel = track[i].level;
    pMP->mTrackViewCos = track[i].view_cos;
    pMP->IncreaseVisible();                           // Tracking.cc:1063
  }
  for (int i = 0; i < F.N; i++)
    if (assigned[i] >= 0) F.mvpMapPoints[i] = vpLocalMapPoints[(size_t)assigned[i]];  // ORBmatcher.cc:122
  if (nToMatch) *nToMatch = ntm;
  return nmatches;
}

// DBoW2::FeatureVector (std::map<NodeId, std::vector<unsigned>>, ascending node ids) -> the flat form of the C ABI
template <class FeatVecT>
inline void FlattenFeatureVector(const FeatVecT& fv, std::vector<orbfe_featvec_node>& nodes, std::vector<int32_t>& idx) {
  nodes.clear();
  idx.clear();
  for (const auto& kv : fv) {
    orbfe_featvec_node nd;
    nd.node_id = (int32_t)kv.first;
    nd.start = (int32_t)idx.size();
    nd.count = (int32_t)kv.second.size();
    nodes.push_back(nd);
    for (const auto i : kv.second) idx.push_back((int32_t)i);
  }
}

// SearchByBoW(KeyFrame *pKF, Frame &F, vector<MapPoint*> &vpMapPointMatches)   L/src/ORBmatcher.cc:161-273 -- whole body.
// Members used: pKF->GetMapPointMatches(), mFeatVec, mDescriptors, mvKeysUn; F.N, mFeatVec, mDescriptors, mvKeys.
template <class KeyFrameT, class FrameT, class MapPointT>
int SearchByBoW(KeyFrameT* pKF, FrameT& F, std::vector<MapPointT*>& vpMapPointMatches, float nnratio, bool checkOrientation) {
  const std::vector<MapPointT*> vpMapPointsKF = pKF->GetMapPointMatches();
  vpMapPointMatches.assign((size_t)F.N, static_cast<MapPointT*>(nullptr));
  const int nKF = (int)vpMapPointsKF.size();
  std::vector<orbfe_featvec_node> nodesKF, nodesF;
  std::vector<int32_t> idxKF, idxF;
  FlattenFeatureVector(pKF->mFeatVec, nodesKF, idxKF);
  FlattenFeatureVector(F.mFeatVec, nodesF, idxF);
  std::vector<uint8_t> validKF((size_t)nKF, 0);
  std::vector<float> anglesKF((size_t)nKF), anglesF((size_t)F.N);
  for (int i = 0; i < nKF; i++) {
    validKF[i] = vpMapPointsKF[i] && !vpMapPointsKF[i]->isBad();   // :191-197
    anglesKF[i] = pKF->mvKeysUn[i].angle;                          // :229
  }
  for (int j = 0; j < F.N; j++) anglesF[j] = F.mvKeys[j].angle;   // :229 (mvKeys, not mvKeysUn: the angle is the same)
  std::vector<int32_t> matchB((size_t)(F.N > 0 ? F.N : 1), -1);
  int nmatches = 0;
  const int rc = orbfe_search_by_bow(pKF->mDescriptors.ptr(0), anglesKF.data(), validKF.data(), nKF, nodesKF.data(),
                                     (int)nodesKF.size(), idxKF.data(), F.mDescriptors.ptr(0), anglesF.data(), F.N, nodesF.data(),
                                     (int)nodesF.size(), idxF.data(), nnratio, checkOrientation ? 1 : 0, matchB.data(), &nmatches);
  if (rc != ORBFE_OK) {
    fprintf(stderr, "ORBmatcher::SearchByBoW: liborbfe error %d: %s\n", rc, orbfe_last_error());
    return 0;
  }
  for (int j = 0; j < F.N; j++)
    if (matchB[j] >= 0) vpMapPointMatches[(size_t)j] = vpMapPointsKF[(size_t)matchB[j]];   // :226
  return nmatches;
}

// SearchByBoW(KeyFrame *pKF1, KeyFrame *pKF2, vector<MapPoint*> &vpMatches12)   L/src/ORBmatcher.cc:494-612 -- whole body.
template <class KeyFrameT, class MapPointT>
int SearchByBoWKeyFrames(KeyFrameT* pKF1, KeyFrameT* pKF2, std::vector<MapPointT*>& vpMatches12, float nnratio,
                         bool checkOrientation) {
  const std::vector<MapPointT*> vpMapPoints1 = pKF1->GetMapPointMatches();
  const std::vector<MapPointT*> vpMapPoints2 = pKF2->GetMapPointMatches();
  const int n1 = (int)vpMapPoints1.size(), n2 = (int)vpMapPoints2.size();
  vpMatches12.assign((size_t)n1, static_cast<MapPointT*>(nullptr));
  std::vector<orbfe_featvec_node> nodes1, nodes2;
  std::vector<int32_t> idx1, idx2;
  FlattenFeatureVector(pKF1->mFeatVec, nodes1, idx1);
  FlattenFeatureVector(pKF2->mFeatVec, nodes2, idx2);
  std::vector<uint8_t> valid1((size_t)n1), valid2((size_t)n2);
  std::vector<float> ang1((size_t)n1), ang2((size_t)n2);
  for (int i = 0; i < n1; i++) { valid1[i] = vpMapPoints1[i] && !vpMapPoints1[i]->isBad(); ang1[i] = pKF1->mvKeysUn[i].angle; }
  for (int i = 0; i < n2; i++) { valid2[i] = vpMapPoints2[i] && !vpMapPoints2[i]->isBad(); ang2[i] = pKF2->mvKeysUn[i].angle; }
  std::vector<int32_t> matchA((size_t)(n1 > 0 ? n1 : 1), -1);
  int nmatches = 0;
  const int rc = orbfe_search_by_bow_kf(pKF1->mDescriptors.ptr(0), ang1.data(), valid1.data(), n1, nodes1.data(), (int)nodes1.size(),
                                        idx1.data(), pKF2->mDescriptors.ptr(0), ang2.data(), valid2.data(), n2, nodes2.data(),
                                        (int)nodes2.size(), idx2.data(), nnratio, checkOrientation ? 1 : 0, matchA.data(), &nmatches);
  if (rc != ORBFE_OK) {
    fprintf(stderr, "ORBmatcher::SearchByBoW(KF,KF): liborbfe error %d: %s\n", rc, orbfe_last_error());
    return 0;
  }
  for (int i = 0; i < n1; i++)
    if (matchA[i] >= 0) vpMatches12[(size_t)i] = vpMapPoints2[(size_t)matchA[i]];   // :566
  return nmatches;
}

// SearchForTriangulation(pKF1, pKF2, F12, vMatchedPairs, bOnlyStereo)   L/src/ORBmatcher.cc:614-764 -- everything after the
// epipole (:622-630), which the caller passes as (ex, ey) together with F12 as nine row-major floats.
// Members used: N, GetMapPoint(i), mvuRight, mvKeysUn, mDescriptors, mFeatVec; pKF2->mvScaleFactors, mvLevelSigma2.
template <class KeyFrameT>
int SearchForTriangulation(KeyFrameT* pKF1, KeyFrameT* pKF2, const float F12[9], float ex, float ey,
                           std::vector<std::pair<size_t, size_t>>& vMatchedPairs, bool bOnlyStereo, bool checkOrientation) {
  static_assert(sizeof(pKF1->mvKeysUn[0]) == sizeof(orbfe_keypoint), "cv::KeyPoint layout");
  const int n1 = pKF1->N, n2 = pKF2->N;
  std::vector<orbfe_featvec_node> nodes1, nodes2;
  std::vector<int32_t> idx1, idx2;
  FlattenFeatureVector(pKF1->mFeatVec, nodes1, idx1);
  FlattenFeatureVector(pKF2->mFeatVec, nodes2, idx2);
  std::vector<uint8_t> has1((size_t)(n1 > 0 ? n1 : 1)), has2((size_t)(n2 > 0 ? n2 : 1));
  for (int i = 0; i < n1; i++) has1[i] = pKF1->GetMapPoint((size_t)i) != nullptr;
  for (int i = 0; i < n2; i++) has2[i] = pKF2->GetMapPoint((size_t)i) != nullptr;
  orbfe_epipolar ep;
  memset(&ep, 0, sizeof(ep));
  memcpy(ep.F12, F12, sizeof(ep.F12));
  ep.ex = ex; ep.ey = ey;
  for (size_t l = 0; l < pKF2->mvScaleFactors.size() && l < ORBFE_MAX_LEVELS; l++) {
    ep.scale_factors[l] = pKF2->mvScaleFactors[l];
    ep.level_sigma2[l] = pKF2->mvLevelSigma2[l];
  }
  std::vector<int32_t> matchA((size_t)(n1 > 0 ? n1 : 1), -1);
  int nmatches = 0;
  const int rc = orbfe_search_for_triangulation(
      reinterpret_cast<const orbfe_keypoint*>(pKF1->mvKeysUn.data()), pKF1->mDescriptors.ptr(0),
      pKF1->mvuRight.empty() ? nullptr : pKF1->mvuRight.data(), has1.data(), n1, nodes1.data(), (int)nodes1.size(), idx1.data(),
      reinterpret_cast<const orbfe_keypoint*>(pKF2->mvKeysUn.data()), pKF2->mDescriptors.ptr(0),
      pKF2->mvuRight.empty() ? nullptr : pKF2->mvuRight.data(), has2.data(), n2, nodes2.data(), (int)nodes2.size(), idx2.data(), &ep,
      bOnlyStereo ? 1 : 0, checkOrientation ? 1 : 0, matchA.data(), &nmatches);
  vMatchedPairs.clear();
  if (rc != ORBFE_OK) {
    fprintf(stderr, "ORBmatcher::SearchForTriangulation: liborbfe error %d: %s\n", rc, orbfe_last_error());
    return 0;
  }
  vMatchedPairs.reserve((size_t)nmatches);
  for (int i = 0; i < n1; i++)
    if (matchA[i] >= 0) vMatchedPairs.push_back(std::make_pair((size_t)i, (size_t)matchA[i]));   // :754-761
  return nmatches;
}

// Frame::ComputeBoW (L/src/Frame.cc:412-417): mBowVec / mFeatVec from the device-side vocabulary transform.
// BowVecT = DBoW2::BowVector (std::map<WordId, WordValue>), FeatVecT = DBoW2::FeatureVector (std::map<NodeId, vector<unsigned>>).
template <class BowVecT, class FeatVecT>
int ComputeBoW(orbfe_vocabulary* voc, const uint8_t* descriptors, int N, BowVecT& bowVec, FeatVecT& featVec, int levelsup = 4) {
  bowVec.clear();
  featVec.clear();
  if (N <= 0) return ORBFE_OK;
  std::vector<int32_t> bowIds((size_t)N), fvIdx((size_t)N);
  std::vector<double> bowVals((size_t)N);
  std::vector<orbfe_featvec_node> fvNodes((size_t)N);
  int nBow = 0, nFv = 0;
  const int rc = orbfe_compute_bow(voc, descriptors, N, levelsup, nullptr, nullptr, nullptr, bowIds.data(), bowVals.data(), &nBow,
                                   fvNodes.data(), fvIdx.data(), &nFv);
  if (rc != ORBFE_OK) {
    fprintf(stderr, "Frame::ComputeBoW: liborbfe error %d: %s\n", rc, orbfe_last_error());
    return rc;
  }
  for (int i = 0; i < nBow; i++) bowVec.insert(bowVec.end(), std::make_pair(bowIds[i], bowVals[i]));   // ascending ids
  for (int i = 0; i < nFv; i++) {
    auto& v = featVec[fvNodes[i].node_id];
    v.assign(fvIdx.begin() + fvNodes[i].start, fvIdx.begin() + fvNodes[i].start + fvNodes[i].count);
  }
  return ORBFE_OK;
}

// Frame::ComputeStereoMatches (L/src/Frame.cc:477-646) -- whole body:
//     void Frame::ComputeStereoMatches() { orbfe_host::ComputeStereoMatches(*this, mpORBextractorLeft, mpORBextractorRight); }
// Called where the reference calls it (Frame.cc:99), i.e. after the two ExtractORB threads (:91-94) have joined: the pyramids
// both extractors built for this pair are still in HBM, so mvImagePyramid is not needed on the host
// (ORBextractor::SetPyramidDownload(false)).  Members used: N, mvKeys, mvKeysRight, mDescriptors, mDescriptorsRight, mbf, mb,
// mvuRight, mvDepth (and the static fx when mb has not been set yet).
// mb: the reference reads Frame::mb as minZ (:505) BEFORE the constructor assigns it (:124) -- an indeterminate value in the
// first frame, the previous frame's mbf / fx afterwards in practice.  A non-positive or non-finite mb is replaced by mbf / fx.
template <class FrameT, class ExtractorT>
int ComputeStereoMatches(FrameT& F, ExtractorT* pLeft, ExtractorT* pRight) {
  const int N = F.N;
  F.mvuRight.assign((size_t)N, -1.0f);   // :478-479
  F.mvDepth.assign((size_t)N, -1.0f);
  const int Nr = (int)F.mvKeysRight.size();
  if (N <= 0 || Nr <= 0) return 0;
  static_assert(sizeof(F.mvKeys[0]) == sizeof(orbfe_keypoint), "cv::KeyPoint layout");
  float mb = F.mb;
  if (!(mb > 0.0f) || !(mb < 3.0e38f)) mb = F.mbf / F.fx;
  int nMatched = 0;
  const int rc = orbfe_stereo_match(pLeft->Handle(), pRight->Handle(), reinterpret_cast<const orbfe_keypoint*>(F.mvKeys.data()),
                                    F.mDescriptors.ptr(0), N, reinterpret_cast<const orbfe_keypoint*>(F.mvKeysRight.data()),
                                    F.mDescriptorsRight.ptr(0), Nr, F.mbf, mb, F.mvuRight.data(), F.mvDepth.data(), &nMatched);
  if (rc != ORBFE_OK) {
    fprintf(stderr, "Frame::ComputeStereoMatches: liborbfe error %d: %s\n", rc, orbfe_last_error());
    F.mvuRight.assign((size_t)N, -1.0f);
    F.mvDepth.assign((size_t)N, -1.0f);
    return 0;
  }
  return nMatched;
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
