// MapPoint.h -- MOCK (test infrastructure) of the reference's MapPoint with the member names, access levels and map
// bookkeeping semantics ORBmatcher relies on (Source/Libraries/ORB_SLAM2/include/MapPoint.h:35-150, src/MapPoint.cc:
// AddObservation :91-103, Replace :184-227, PredictScale :393-423).  No mutexes, no Map.  Written for tests/cpp: lets
// csrc/host/ORBmatcher.cc compile and run without the reference tree (which needs OpenCV).
#ifndef MAPPOINT_H
#define MAPPOINT_H
#include <math.h>

#include <map>

#include "Frame.h"
#include "KeyFrame.h"

namespace ORB_SLAM2 {

class MapPoint {
 public:
  MapPoint(const cv::Mat& Pos, const cv::Mat& Normal, const cv::Mat& Desc, float minD, float maxD)
      : nObs(0), mTrackProjX(0), mTrackProjY(0), mTrackProjXR(0), mbTrackInView(false), mnTrackScaleLevel(0), mTrackViewCos(0),
        mnLastFrameSeen(0), mWorldPos(Pos.clone()), mNormalVector(Normal.clone()), mDescriptor(Desc.clone()), mnVisible(1),
        mnFound(1), mbBad(false), mpReplaced(nullptr), mfMinDistance(minD), mfMaxDistance(maxD) {}
  cv::Mat GetWorldPos() { return mWorldPos.clone(); }
  cv::Mat GetNormal() { return mNormalVector.clone(); }
  std::map<KeyFrame*, std::size_t> GetObservations() { return mObservations; }
  int Observations() { return nObs; }
  void AddObservation(KeyFrame* pKF, std::size_t idx) {
    if (mObservations.count(pKF)) return;
    mObservations[pKF] = idx;
    if (pKF->mvuRight[idx] >= 0) nObs += 2;
    else nObs++;
  }
  int GetIndexInKeyFrame(KeyFrame* pKF) { return mObservations.count(pKF) ? (int)mObservations[pKF] : -1; }
  bool IsInKeyFrame(KeyFrame* pKF) { return mObservations.count(pKF) != 0; }
  void SetBadFlag() { mbBad = true; }
  bool isBad() { return mbBad; }
  void Replace(MapPoint* pMP) {
    if (pMP == this) return;
    std::map<KeyFrame*, std::size_t> obs = mObservations;
    mObservations.clear();
    mbBad = true;
    mpReplaced = pMP;
    for (auto& kv : obs) {
      KeyFrame* pKF = kv.first;
      if (!pMP->IsInKeyFrame(pKF)) {
        pKF->ReplaceMapPointMatch(kv.second, pMP);
        pMP->AddObservation(pKF, kv.second);
      } else {
        pKF->EraseMapPointMatch(kv.second);
      }
    }
    pMP->IncreaseFound(mnFound);
    pMP->IncreaseVisible(mnVisible);
    pMP->ComputeDistinctiveDescriptors();
  }
  MapPoint* GetReplaced() { return mpReplaced; }
  void IncreaseVisible(int n = 1) { mnVisible += n; }
  void IncreaseFound(int n = 1) { mnFound += n; }
  void ComputeDistinctiveDescriptors() { nDescriptorUpdates++; }
  cv::Mat GetDescriptor() { return mDescriptor.clone(); }
  float GetMinDistanceInvariance() { return 0.8f * mfMinDistance; }
  float GetMaxDistanceInvariance() { return 1.2f * mfMaxDistance; }
  int PredictScale(const float& currentDist, KeyFrame* pKF) { return Predict(currentDist, pKF->mfLogScaleFactor, pKF->mnScaleLevels); }
  int PredictScale(const float& currentDist, Frame* pF) { return Predict(currentDist, pF->mfLogScaleFactor, pF->mnScaleLevels); }

 public:
  int nObs;
  float mTrackProjX, mTrackProjY, mTrackProjXR;
  bool mbTrackInView;
  int mnTrackScaleLevel;
  float mTrackViewCos;
  long unsigned int mnLastFrameSeen;
  int nDescriptorUpdates = 0;   // mock only

 protected:
  int Predict(float d, float logsf, int nl) {
    const float ratio = mfMaxDistance / d;
    int nScale = (int)ceil(log(ratio) / logsf);
    if (nScale < 0) nScale = 0;
    else if (nScale >= nl) nScale = nl - 1;
    return nScale;
  }
  cv::Mat mWorldPos;
  std::map<KeyFrame*, std::size_t> mObservations;
  cv::Mat mNormalVector;
  cv::Mat mDescriptor;
  int mnVisible, mnFound;
  bool mbBad;
  MapPoint* mpReplaced;
  float mfMinDistance;
  float mfMaxDistance;
};

}  // namespace ORB_SLAM2

namespace ORB_SLAM2 {
inline std::set<MapPoint*> KeyFrame::GetMapPoints() {   // KeyFrame.cc:237-250: every match that exists and is not bad
  std::set<MapPoint*> s;
  for (MapPoint* pMP : mvpMapPoints)
    if (pMP && !pMP->isBad()) s.insert(pMP);
  return s;
}
}  // namespace ORB_SLAM2
#endif
