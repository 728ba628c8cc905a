// ORBmatcher.cc -- drop-in ORB_SLAM2::ORBmatcher over liborbfe's C ABI (MI355X): every symbol the reference's
// Source/Libraries/ORB_SLAM2/src/ORBmatcher.cc defines (L/include/ORBmatcher.h:34-114, statics :38-40).
//
// Each method marshals the members the reference method reads into the POD records of include/orbfe.h, runs the search on
// the GPU and writes the result back where the reference writes it (F.mvpMapPoints, vpMatched, vnMatches12, ...).  What
// stays on the host is what acts on SLAM objects: the small pose products in front of a search (twc, tlc, the Scw
// decomposition, sR21 / t21) and the map bookkeeping behind it (Replace / AddObservation / AddMapPoint, vpReplacePoint, the
// mutual check of SearchBySim3), replayed in point order.  The 3x3 products are written out with the arithmetic OpenCV uses
// for them (small-matrix gemm: float dot, double alpha/beta epilogue; A / s = A * (float)(1.0 / s)) so that the same source
// compiles against OpenCV and against cvlite.h; cv::Mat is touched only through at<float>() / ptr().
#include "ORBmatcher.h"

#include <limits.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "ORBmatcher_hip.h"

namespace ORB_SLAM2 {


const int ORBmatcher::TH_HIGH = 100;
const int ORBmatcher::TH_LOW = 50;
const int ORBmatcher::HISTO_LENGTH = 30;

ORBmatcher::ORBmatcher(float nnratio, bool checkOri) : mfNNratio(nnratio), mbCheckOrientation(checkOri) {}

namespace {

// mfMinDistance / mfMaxDistance are protected in MapPoint and have no plain getter (GetM??DistanceInvariance scale them, and
// MapPoint::PredictScale divides the UN-scaled value); a pointer to member formed inside a derived class reaches them without
// touching MapPoint.h.  The struct is never instantiated.
struct MapPointAccess : public MapPoint {
  static float MapPoint::*MinDistance() { return &MapPointAccess::mfMinDistance; }
  static float MapPoint::*MaxDistance() { return &MapPointAccess::mfMaxDistance; }
};
inline float RawMinDistance(MapPoint* p) { return p->*MapPointAccess::MinDistance(); }
inline float RawMaxDistance(MapPoint* p) { return p->*MapPointAccess::MaxDistance(); }

// rows 0..2 of a 3x4 / 4x4 pose: rotation (row-major) and translation
inline void ReadPose(const cv::Mat& T, float R[9], float t[3]) {
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) R[3 * r + c] = T.at<float>(r, c);
    t[r] = T.at<float>(r, 3);
  }
}
inline void ReadMat33(const cv::Mat& M, float R[9]) {
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) R[3 * r + c] = M.at<float>(r, c);
}
inline void ReadVec3(const cv::Mat& v, float o[3]) {
  for (int r = 0; r < 3; r++) o[r] = v.at<float>(r);
}
// A * x + b as cv::gemm evaluates it for 3x3 * 3x1 CV_32F operands
inline void Gemm3(const float A[9], const float x[3], const float b[3], float o[3]) {
  for (int r = 0; r < 3; r++) {
    const float d = A[3 * r] * x[0] + A[3 * r + 1] * x[1] + A[3 * r + 2] * x[2];
    o[r] = (float)((double)d * 1.0 + (double)b[r] * 1.0);
  }
}
// -A.t() * x  (gemm with GEMM_1_T, alpha = -1)
inline void NegTransposedTimes(const float A[9], const float x[3], float o[3]) {
  for (int r = 0; r < 3; r++) {
    const float d = A[r] * x[0] + A[3 + r] * x[1] + A[6 + r] * x[2];
    o[r] = (float)((double)d * -1.0);
  }
}
// -A * x
inline void NegTimes(const float A[9], const float x[3], float o[3]) {
  for (int r = 0; r < 3; r++) {
    const float d = A[3 * r] * x[0] + A[3 * r + 1] * x[1] + A[3 * r + 2] * x[2];
    o[r] = (float)((double)d * -1.0);
  }
}

// Decompose Scw (L/src/ORBmatcher.cc:285-289, 920-924): Rcw = sRcw / scw, tcw = Scw(0:3, 3) / scw, Ow = -Rcw.t() * tcw
inline void DecomposeSim3(const cv::Mat& Scw, float Rcw[9], float tcw[3], float Ow[3]) {
  float sR[9], st[3];
  ReadPose(Scw, sR, st);
  double d = 0;
  for (int c = 0; c < 3; c++) d += (double)sR[c] * (double)sR[c];   // sRcw.row(0).dot(sRcw.row(0))
  const float scw = (float)sqrt(d);
  const float a = (float)(1.0 / (double)scw);                        // Mat / s  ==  convertTo(alpha = 1 / s)
  for (int i = 0; i < 9; i++) Rcw[i] = sR[i] * a;
  for (int i = 0; i < 3; i++) tcw[i] = st[i] * a;
  NegTransposedTimes(Rcw, tcw, Ow);
}

template <class CamT>
inline void FillKeyFrameCamera(orbfe_kf_camera& cam, const CamT* pKF, float th) {
  cam.fx = pKF->fx; cam.fy = pKF->fy; cam.cx = pKF->cx; cam.cy = pKF->cy; cam.mbf = pKF->mbf;
  cam.min_x = (float)pKF->mnMinX; cam.max_x = (float)pKF->mnMaxX; cam.min_y = (float)pKF->mnMinY; cam.max_y = (float)pKF->mnMaxY;
  cam.log_scale_factor = pKF->mfLogScaleFactor;
  cam.n_levels = pKF->mnScaleLevels;
  cam.th = th;
  for (int l = 0; l < pKF->mnScaleLevels && l < ORBFE_MAX_LEVELS; l++) cam.scale_factors[l] = pKF->mvScaleFactors[l];
}

inline void FillPoint(orbfe_kf_point& e, MapPoint* pMP, float angle = 0.f) {
  memset(&e, 0, sizeof(e));
  const cv::Mat P = pMP->GetWorldPos(), Pn = pMP->GetNormal(), d = pMP->GetDescriptor();
  for (int r = 0; r < 3; r++) { e.pos[r] = P.at<float>(r); e.normal[r] = Pn.at<float>(r); }
  e.min_distance = RawMinDistance(pMP);
  e.max_distance = RawMaxDistance(pMP);
  e.angle = angle;
  memcpy(e.desc, d.ptr(0), 32);
}

template <class KFT>
inline orbfe_frame_view KeyFrameView(const KFT* pKF) {
  static_assert(sizeof(cv::KeyPoint) == sizeof(orbfe_keypoint), "cv::KeyPoint layout");
  orbfe_frame_view v;
  v.n = pKF->N;
  v.keys_un = reinterpret_cast<const orbfe_keypoint*>(pKF->mvKeysUn.data());
  v.desc = pKF->mDescriptors.ptr(0);
  v.u_right = pKF->mvuRight.empty() ? nullptr : pKF->mvuRight.data();
  v.min_x = (float)pKF->mnMinX; v.max_x = (float)pKF->mnMaxX; v.min_y = (float)pKF->mnMinY; v.max_y = (float)pKF->mnMaxY;
  return v;
}

inline bool KfSearch(const char* who, const orbfe_frame_view& v, const float* invSigma2, const orbfe_kf_camera& cam,
                     const std::vector<orbfe_kf_point>& pts, int mode, bool checkOri, int maxDist, std::vector<uint8_t>* blocked,
                     std::vector<orbfe_kf_result>& res, int* nMatches) {
  res.resize(pts.size());
  int nm = 0;
  const int rc = orbfe_kf_search(&v, invSigma2, &cam, pts.data(), (int)pts.size(), mode, checkOri ? 1 : 0, maxDist,
                                 blocked ? blocked->data() : nullptr, res.data(), &nm);
  if (rc != ORBFE_OK) {
    fprintf(stderr, "ORBmatcher::%s: liborbfe error %d: %s\n", who, rc, orbfe_last_error());
    return false;
  }
  if (nMatches) *nMatches = nm;
  return true;
}

}  // namespace

// ------------------------------------------------------------------------------------------------ small helpers (L/src/ORBmatcher.cc)
int ORBmatcher::DescriptorDistance(const cv::Mat& a, const cv::Mat& b) {   // :1542-1556
  return orbfe_host::DescriptorDistance(a.ptr(0), b.ptr(0));
}

float ORBmatcher::RadiusByViewingCos(const float& viewCos) { return orbfe_host::RadiusByViewingCos(viewCos); }   // :130-135

// distance of kp2 to the epipolar line of kp1, against the chi-square bound of kp2's level (:137-159)
bool ORBmatcher::CheckDistEpipolarLine(const cv::KeyPoint& kp1, const cv::KeyPoint& kp2, const cv::Mat& F12, const KeyFrame* pKF2) {
  const float a = kp1.pt.x * F12.at<float>(0, 0) + kp1.pt.y * F12.at<float>(1, 0) + F12.at<float>(2, 0);
  const float b = kp1.pt.x * F12.at<float>(0, 1) + kp1.pt.y * F12.at<float>(1, 1) + F12.at<float>(2, 1);
  const float c = kp1.pt.x * F12.at<float>(0, 2) + kp1.pt.y * F12.at<float>(1, 2) + F12.at<float>(2, 2);
  const float num = a * kp2.pt.x + b * kp2.pt.y + c;
  const float den = a * a + b * b;
  if (den == 0) return false;
  const float dsqr = num * num / den;
  return dsqr < 3.84 * pKF2->mvLevelSigma2[kp2.octave];
}

// the three fullest bins; the second / third are dropped when clearly smaller than the first (:1506-1538)
void ORBmatcher::ComputeThreeMaxima(std::vector<int>* histo, const int L, int& ind1, int& ind2, int& ind3) {
  int best[3] = {0, 0, 0};
  int idx[3] = {-1, -1, -1};
  for (int i = 0; i < L; i++) {
    const int s = (int)histo[i].size();
    int k = 3;
    while (k > 0 && s > best[k - 1]) k--;   // strict: an earlier bin of equal size keeps its rank
    if (k == 3) continue;
    for (int j = 2; j > k; j--) { best[j] = best[j - 1]; idx[j] = idx[j - 1]; }
    best[k] = s; idx[k] = i;
  }
  ind1 = idx[0]; ind2 = idx[1]; ind3 = idx[2];
  if (best[1] < 0.1f * (float)best[0]) { ind2 = -1; ind3 = -1; }
  else if (best[2] < 0.1f * (float)best[0]) ind3 = -1;
}

// ------------------------------------------------------------------------------------------------ Tracking
int ORBmatcher::SearchByProjection(Frame& F, const std::vector<MapPoint*>& vpMapPoints, const float th) {   // :45-128
  return orbfe_host::SearchByProjectionPoints(F, vpMapPoints, th, mfNNratio);
}

int ORBmatcher::SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, const float th, const bool bMono) {   // :1247-1383
  float Rcw[9], tcw[3], twc[3], Rlw[9], tlw[3], tlc[3];
  ReadPose(CurrentFrame.mTcw, Rcw, tcw);
  NegTransposedTimes(Rcw, tcw, twc);                       // twc = -Rcw.t() * tcw
  ReadPose(LastFrame.mTcw, Rlw, tlw);
  Gemm3(Rlw, twc, tlw, tlc);                               // tlc = Rlw * twc + tlw
  const bool bForward = tlc[2] > CurrentFrame.mb && !bMono;
  const bool bBackward = -tlc[2] > CurrentFrame.mb && !bMono;
  std::vector<orbfe_query> q((size_t)LastFrame.N);
  for (int i = 0; i < LastFrame.N; i++) {
    orbfe_query& e = q[(size_t)i];
    memset(&e, 0, sizeof(e));
    MapPoint* pMP = LastFrame.mvpMapPoints[i];
    if (!pMP || LastFrame.mvbOutlier[i]) continue;
    const cv::Mat x3Dw = pMP->GetWorldPos();
    float xw[3], xc3[3];
    ReadVec3(x3Dw, xw);
    Gemm3(Rcw, xw, tcw, xc3);
    const float xc = xc3[0], yc = xc3[1];
    const float invzc = (float)(1.0 / xc3[2]);
    if (invzc < 0) continue;
    const float u = CurrentFrame.fx * xc * invzc + CurrentFrame.cx;
    const float v = CurrentFrame.fy * yc * invzc + CurrentFrame.cy;
    if (u < CurrentFrame.mnMinX || u > CurrentFrame.mnMaxX) continue;
    if (v < CurrentFrame.mnMinY || v > CurrentFrame.mnMaxY) continue;
    const int nLastOctave = LastFrame.mvKeys[i].octave;
    e.u = u; e.v = v;
    e.u_r = u - CurrentFrame.mbf * invzc;
    e.radius = th * CurrentFrame.mvScaleFactors[nLastOctave];
    if (bForward) { e.min_level = nLastOctave; e.max_level = -1; }
    else if (bBackward) { e.min_level = 0; e.max_level = nLastOctave; }
    else { e.min_level = nLastOctave - 1; e.max_level = nLastOctave + 1; }
    e.valid = 1;
    e.blocks = pMP->Observations() > 0;
    e.angle = LastFrame.mvKeysUn[i].angle;
    const cv::Mat d = pMP->GetDescriptor();
    memcpy(e.desc, d.ptr(0), 32);
  }
  return orbfe_host::SearchByProjectionFrame(CurrentFrame, LastFrame, q, mbCheckOrientation);
}

int ORBmatcher::SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, const std::set<MapPoint*>& sAlreadyFound, const float th,
                                   const int ORBdist) {   // :1385-1504
  orbfe_kf_camera cam;
  memset(&cam, 0, sizeof(cam));
  ReadPose(CurrentFrame.mTcw, cam.R, cam.t);
  NegTransposedTimes(cam.R, cam.t, cam.Ow);                // Ow = -Rcw.t() * tcw
  FillKeyFrameCamera(cam, &CurrentFrame, th);
  const std::vector<MapPoint*> vpMPs = pKF->GetMapPointMatches();
  std::vector<orbfe_kf_point> pts(vpMPs.size());
  for (size_t i = 0; i < vpMPs.size(); i++) {
    MapPoint* pMP = vpMPs[i];
    if (pMP && !pMP->isBad() && !sAlreadyFound.count(pMP)) FillPoint(pts[i], pMP, pKF->mvKeysUn[i].angle);
    else { memset(&pts[i], 0, sizeof(pts[i])); pts[i].skip = 1; }
  }
  std::vector<uint8_t> blocked((size_t)(CurrentFrame.N > 0 ? CurrentFrame.N : 1), 0);
  for (int i = 0; i < CurrentFrame.N; i++) blocked[(size_t)i] = CurrentFrame.mvpMapPoints[i] != nullptr;   // :1453
  orbfe_frame_view v = orbfe_host::MakeFrameView(CurrentFrame);
  v.u_right = nullptr;
  std::vector<orbfe_kf_result> res;
  int nmatches = 0;
  if (!KfSearch("SearchByProjection", v, nullptr, cam, pts, ORBFE_KF_RELOC, mbCheckOrientation, ORBdist, &blocked, res, &nmatches)) return 0;
  for (size_t i = 0; i < vpMPs.size(); i++)
    if (res[i].best_idx >= 0) CurrentFrame.mvpMapPoints[(size_t)res[i].best_idx] = vpMPs[i];   // :1467; those the rotation check drops stay NULL
  return nmatches;
}

// ------------------------------------------------------------------------------------------------ LoopClosing
int ORBmatcher::SearchByProjection(KeyFrame* pKF, cv::Mat Scw, const std::vector<MapPoint*>& vpPoints, std::vector<MapPoint*>& vpMatched,
                                   int th) {   // :275-386
  orbfe_kf_camera cam;
  memset(&cam, 0, sizeof(cam));
  DecomposeSim3(Scw, cam.R, cam.t, cam.Ow);
  FillKeyFrameCamera(cam, pKF, (float)th);
  std::set<MapPoint*> spAlreadyFound(vpMatched.begin(), vpMatched.end());
  spAlreadyFound.erase(static_cast<MapPoint*>(NULL));
  std::vector<orbfe_kf_point> pts(vpPoints.size());
  for (size_t i = 0; i < vpPoints.size(); i++) {
    MapPoint* pMP = vpPoints[i];
    if (pMP->isBad() || spAlreadyFound.count(pMP)) { memset(&pts[i], 0, sizeof(pts[i])); pts[i].skip = 1; }
    else FillPoint(pts[i], pMP);
  }
  std::vector<uint8_t> blocked(vpMatched.size() ? vpMatched.size() : 1, 0);
  for (size_t i = 0; i < vpMatched.size(); i++) blocked[i] = vpMatched[i] != nullptr;   // :358
  const orbfe_frame_view v = KeyFrameView(pKF);
  std::vector<orbfe_kf_result> res;
  int nmatches = 0;
  if (!KfSearch("SearchByProjection", v, nullptr, cam, pts, ORBFE_KF_LOOP, false, TH_LOW, &blocked, res, &nmatches)) return 0;
  for (size_t i = 0; i < vpPoints.size(); i++)
    if (res[i].best_idx >= 0) vpMatched[(size_t)res[i].best_idx] = vpPoints[i];   // :378
  return nmatches;
}

int ORBmatcher::SearchByBoW(KeyFrame* pKF, Frame& F, std::vector<MapPoint*>& vpMapPointMatches) {   // :161-273
  return orbfe_host::SearchByBoW(pKF, F, vpMapPointMatches, mfNNratio, mbCheckOrientation);
}

int ORBmatcher::SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12) {   // :494-612
  return orbfe_host::SearchByBoWKeyFrames(pKF1, pKF2, vpMatches12, mfNNratio, mbCheckOrientation);
}

int ORBmatcher::SearchForInitialization(Frame& F1, Frame& F2, std::vector<cv::Point2f>& vbPrevMatched, std::vector<int>& vnMatches12,
                                        int windowSize) {   // :388-492
  static_assert(sizeof(cv::Point2f) == 2 * sizeof(float), "cv::Point2f layout");
  vnMatches12.assign(F1.mvKeysUn.size(), -1);
  const orbfe_frame_view v1 = orbfe_host::MakeFrameView(F1), v2 = orbfe_host::MakeFrameView(F2);
  if (vbPrevMatched.size() < F1.mvKeysUn.size()) vbPrevMatched.resize(F1.mvKeysUn.size());
  std::vector<int32_t> m12(F1.mvKeysUn.size() ? F1.mvKeysUn.size() : 1, -1);
  int nmatches = 0;
  const int rc = orbfe_search_for_initialization(&v1, &v2, reinterpret_cast<float*>(vbPrevMatched.data()), windowSize, mfNNratio,
                                                 mbCheckOrientation ? 1 : 0, m12.data(), &nmatches);
  if (rc != ORBFE_OK) {
    fprintf(stderr, "ORBmatcher::SearchForInitialization: liborbfe error %d: %s\n", rc, orbfe_last_error());
    return 0;
  }
  for (size_t i = 0; i < vnMatches12.size(); i++) vnMatches12[i] = m12[i];
  return nmatches;
}

// ------------------------------------------------------------------------------------------------ LocalMapping
int ORBmatcher::SearchForTriangulation(KeyFrame* pKF1, KeyFrame* pKF2, cv::Mat F12,
                                       std::vector<std::pair<std::size_t, std::size_t>>& vMatchedPairs, const bool bOnlyStereo) {   // :614-764
  // epipole of pKF1's camera centre in pKF2's image (:622-630)
  float Cw[3], R2w[9], t2w[3], C2[3];
  ReadVec3(pKF1->GetCameraCenter(), Cw);
  ReadMat33(pKF2->GetRotation(), R2w);
  ReadVec3(pKF2->GetTranslation(), t2w);
  Gemm3(R2w, Cw, t2w, C2);
  const float invz = 1.0f / C2[2];
  const float ex = pKF2->fx * C2[0] * invz + pKF2->cx;
  const float ey = pKF2->fy * C2[1] * invz + pKF2->cy;
  float F[9];
  ReadMat33(F12, F);
  return orbfe_host::SearchForTriangulation(pKF1, pKF2, F, ex, ey, vMatchedPairs, bOnlyStereo, mbCheckOrientation);
}

int ORBmatcher::Fuse(KeyFrame* pKF, const std::vector<MapPoint*>& vpMapPoints, const float th) {   // :766-907
  orbfe_kf_camera cam;
  memset(&cam, 0, sizeof(cam));
  ReadMat33(pKF->GetRotation(), cam.R);
  ReadVec3(pKF->GetTranslation(), cam.t);
  ReadVec3(pKF->GetCameraCenter(), cam.Ow);
  FillKeyFrameCamera(cam, pKF, th);
  const int nMPs = (int)vpMapPoints.size();
  std::vector<orbfe_kf_point> pts((size_t)nMPs);
  for (int i = 0; i < nMPs; i++) {
    MapPoint* pMP = vpMapPoints[(size_t)i];
    if (!pMP || pMP->isBad() || pMP->IsInKeyFrame(pKF)) { memset(&pts[(size_t)i], 0, sizeof(orbfe_kf_point)); pts[(size_t)i].skip = 1; }
    else FillPoint(pts[(size_t)i], pMP);
  }
  const orbfe_frame_view v = KeyFrameView(pKF);
  std::vector<orbfe_kf_result> res;
  if (!KfSearch("Fuse", v, pKF->mvInvLevelSigma2.data(), cam, pts, ORBFE_KF_FUSE, false, TH_LOW, nullptr, res, nullptr)) return 0;
  // the map update of :870-884, in point order.  An earlier fusion can make a later point bad or put it into the keyframe
  // (Replace moves observations), so the two tests of :787 are taken again on the live objects; the search result of a
  // point does not depend on the earlier ones (no candidate is ever skipped for being taken).
  int nFused = 0;
  for (int i = 0; i < nMPs; i++) {
    MapPoint* pMP = vpMapPoints[(size_t)i];
    if (!pMP) continue;
    if (pMP->isBad() || pMP->IsInKeyFrame(pKF)) continue;
    const int bestIdx = res[(size_t)i].best_idx;
    if (bestIdx < 0 || res[(size_t)i].best_dist > TH_LOW) continue;
    MapPoint* pMPinKF = pKF->GetMapPoint((size_t)bestIdx);
    if (pMPinKF) {
      if (!pMPinKF->isBad()) {
        if (pMPinKF->Observations() > pMP->Observations()) pMP->Replace(pMPinKF);
        else pMPinKF->Replace(pMP);
      }
    } else {
      pMP->AddObservation(pKF, (size_t)bestIdx);
      pKF->AddMapPoint(pMP, (size_t)bestIdx);
    }
    nFused++;
  }
  return nFused;
}

// ------------------------------------------------------------------------------------------------ LoopClosing (Sim3)
int ORBmatcher::Fuse(KeyFrame* pKF, cv::Mat Scw, const std::vector<MapPoint*>& vpPoints, float th, std::vector<MapPoint*>& vpReplacePoint) {   // :909-1027
  orbfe_kf_camera cam;
  memset(&cam, 0, sizeof(cam));
  DecomposeSim3(Scw, cam.R, cam.t, cam.Ow);
  FillKeyFrameCamera(cam, pKF, th);
  const std::set<MapPoint*> spAlreadyFound = pKF->GetMapPoints();
  const int nPoints = (int)vpPoints.size();
  std::vector<orbfe_kf_point> pts((size_t)nPoints);
  for (int i = 0; i < nPoints; i++) {
    MapPoint* pMP = vpPoints[(size_t)i];
    if (pMP->isBad() || spAlreadyFound.count(pMP)) { memset(&pts[(size_t)i], 0, sizeof(orbfe_kf_point)); pts[(size_t)i].skip = 1; }
    else FillPoint(pts[(size_t)i], pMP);
  }
  const orbfe_frame_view v = KeyFrameView(pKF);
  std::vector<orbfe_kf_result> res;
  if (!KfSearch("Fuse", v, nullptr, cam, pts, ORBFE_KF_FUSE_SIM3, false, TH_LOW, nullptr, res, nullptr)) return 0;
  int nFused = 0;
  for (int iMP = 0; iMP < nPoints; iMP++) {   // :1010-1022
    const int bestIdx = res[(size_t)iMP].best_idx;
    if (bestIdx < 0 || res[(size_t)iMP].best_dist > TH_LOW) continue;
    MapPoint* pMP = vpPoints[(size_t)iMP];
    MapPoint* pMPinKF = pKF->GetMapPoint((size_t)bestIdx);
    if (pMPinKF) {
      if (!pMPinKF->isBad()) vpReplacePoint[(size_t)iMP] = pMPinKF;
    } else {
      pMP->AddObservation(pKF, (size_t)bestIdx);
      pKF->AddMapPoint(pMP, (size_t)bestIdx);
    }
    nFused++;
  }
  return nFused;
}

int ORBmatcher::SearchBySim3(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12, const float& s12, const cv::Mat& R12,
                             const cv::Mat& t12, const float th) {   // :1029-1245
  float R1w[9], t1w[3], R2w[9], t2w[3], r12[9], t12v[3], sR12[9], sR21[9], t21[3];
  ReadMat33(pKF1->GetRotation(), R1w); ReadVec3(pKF1->GetTranslation(), t1w);
  ReadMat33(pKF2->GetRotation(), R2w); ReadVec3(pKF2->GetTranslation(), t2w);
  ReadMat33(R12, r12); ReadVec3(t12, t12v);
  const float inv_s = (float)(1.0 / (double)s12);
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) {
      sR12[3 * r + c] = r12[3 * r + c] * s12;        // sR12 = s12 * R12
      sR21[3 * r + c] = r12[3 * c + r] * inv_s;      // sR21 = (1.0 / s12) * R12.t()
    }
  NegTimes(sR21, t12v, t21);                         // t21 = -sR21 * t12

  const std::vector<MapPoint*> vpMapPoints1 = pKF1->GetMapPointMatches();
  const int N1 = (int)vpMapPoints1.size();
  const std::vector<MapPoint*> vpMapPoints2 = pKF2->GetMapPointMatches();
  const int N2 = (int)vpMapPoints2.size();
  std::vector<bool> vbAlreadyMatched1((size_t)N1, false), vbAlreadyMatched2((size_t)N2, false);
  for (int i = 0; i < N1; i++) {
    MapPoint* pMP = vpMatches12[(size_t)i];
    if (pMP) {
      vbAlreadyMatched1[(size_t)i] = true;
      const int idx2 = pMP->GetIndexInKeyFrame(pKF2);
      if (idx2 >= 0 && idx2 < N2) vbAlreadyMatched2[(size_t)idx2] = true;
    }
  }
  // one direction: the map points of `from` through (Ra, ta) then (Rb, tb) into `into`; intrinsics are pKF1's in both
  // directions, bounds / levels the target keyframe's, as the reference reads them
  auto direction = [&](const std::vector<MapPoint*>& vpMPs, const std::vector<bool>& done, const float* Ra, const float* ta,
                       const float* Rb, const float* tb, KeyFrame* into, std::vector<int>& vnMatch) -> bool {
    orbfe_kf_camera cam;
    memset(&cam, 0, sizeof(cam));
    memcpy(cam.R, Ra, sizeof(cam.R)); memcpy(cam.t, ta, sizeof(cam.t));
    memcpy(cam.R2, Rb, sizeof(cam.R2)); memcpy(cam.t2, tb, sizeof(cam.t2));
    FillKeyFrameCamera(cam, into, th);
    cam.fx = pKF1->fx; cam.fy = pKF1->fy; cam.cx = pKF1->cx; cam.cy = pKF1->cy;
    std::vector<orbfe_kf_point> pts(vpMPs.size());
    for (size_t i = 0; i < vpMPs.size(); i++) {
      MapPoint* pMP = vpMPs[i];
      if (!pMP || done[i] || pMP->isBad()) { memset(&pts[i], 0, sizeof(pts[i])); pts[i].skip = 1; }
      else FillPoint(pts[i], pMP);
    }
    const orbfe_frame_view v = KeyFrameView(into);
    std::vector<orbfe_kf_result> res;
    if (!KfSearch("SearchBySim3", v, nullptr, cam, pts, ORBFE_KF_SIM3, false, TH_HIGH, nullptr, res, nullptr)) return false;
    for (size_t i = 0; i < vpMPs.size(); i++)
      if (res[i].best_idx >= 0 && res[i].best_dist <= TH_HIGH) vnMatch[i] = res[i].best_idx;
    return true;
  };
  std::vector<int> vnMatch1((size_t)N1, -1), vnMatch2((size_t)N2, -1);
  if (!direction(vpMapPoints1, vbAlreadyMatched1, R1w, t1w, sR21, t21, pKF2, vnMatch1)) return 0;    // :1063-1147
  if (!direction(vpMapPoints2, vbAlreadyMatched2, R2w, t2w, sR12, t12v, pKF1, vnMatch2)) return 0;   // :1149-1223
  int nFound = 0;
  for (int i1 = 0; i1 < N1; i1++) {   // agreement (:1228-1243)
    const int idx2 = vnMatch1[(size_t)i1];
    if (idx2 >= 0 && vnMatch2[(size_t)idx2] == i1) {
      vpMatches12[(size_t)i1] = vpMapPoints2[(size_t)idx2];
      nFound++;
    }
  }
  return nFound;
}

}  // namespace ORB_SLAM2
