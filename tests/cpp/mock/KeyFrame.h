// KeyFrame.h -- MOCK (test infrastructure) of the reference's KeyFrame: the public data members and methods ORBmatcher reads
// and writes (Source/Libraries/ORB_SLAM2/include/KeyFrame.h:40-240), same names, const-ness dropped where the tests fill them.
#ifndef KEYFRAME_H
#define KEYFRAME_H
#include <set>
#include <vector>

#include "Frame.h"

namespace ORB_SLAM2 {
class MapPoint;

class KeyFrame {
 public:
  KeyFrame() {}
  cv::Mat GetPose() { return Tcw.clone(); }
  cv::Mat GetCameraCenter() { return Ow.clone(); }
  cv::Mat GetRotation() { return Tcw(cv::Rect(0, 0, 3, 3)).clone(); }
  cv::Mat GetTranslation() { return Tcw(cv::Rect(3, 0, 1, 3)).clone(); }
  void AddMapPoint(MapPoint* pMP, const std::size_t& idx) { mvpMapPoints[idx] = pMP; }
  void EraseMapPointMatch(const std::size_t& idx) { mvpMapPoints[idx] = nullptr; }
  void ReplaceMapPointMatch(const std::size_t& idx, MapPoint* pMP) { mvpMapPoints[idx] = pMP; }
  std::set<MapPoint*> GetMapPoints();   // non-NULL, not bad (defined where MapPoint is complete)
  std::vector<MapPoint*> GetMapPointMatches() { return mvpMapPoints; }
  MapPoint* GetMapPoint(const std::size_t& idx) { return mvpMapPoints[idx]; }
  bool IsInImage(const float& x, const float& y) const { return (x >= mnMinX && x < mnMaxX && y >= mnMinY && y < mnMaxY); }

 public:
  float fx = 0, fy = 0, cx = 0, cy = 0, invfx = 0, invfy = 0, mbf = 0, mb = 0, mThDepth = 0;
  int N = 0;
  std::vector<cv::KeyPoint> mvKeys;
  std::vector<cv::KeyPoint> mvKeysUn;
  std::vector<float> mvuRight;
  std::vector<float> mvDepth;
  cv::Mat mDescriptors;
  DBoW2::BowVector mBowVec;
  DBoW2::FeatureVector mFeatVec;
  int mnScaleLevels = 0;
  float mfScaleFactor = 0;
  float mfLogScaleFactor = 0;
  std::vector<float> mvScaleFactors;
  std::vector<float> mvLevelSigma2;
  std::vector<float> mvInvLevelSigma2;
  int mnMinX = 0, mnMinY = 0, mnMaxX = 0, mnMaxY = 0;

  // mock set-up helpers
  cv::Mat Tcw, Ow;
  std::vector<MapPoint*> mvpMapPoints;
};

}  // namespace ORB_SLAM2
#endif
