// ORBmatcher.h -- drop-in replacement for the reference header Source/Libraries/ORB_SLAM2/include/ORBmatcher.h:20-118.
// Same namespace, class name, constructor, the eleven search / fuse methods, the static DescriptorDistance, the three
// static constants and the three protected helpers, so Tracking.cc, LocalMapping.cc, LoopClosing.cc, Frame.cc and
// MapPoint.cc compile and link unchanged.  The searches run on an MI355X through liborbfe's C ABI (include/orbfe.h); the
// definitions are in ORBmatcher.cc next to this file.
#ifndef ORBMATCHER_H
#define ORBMATCHER_H

#include <set>
#include <vector>

#ifdef ORBFE_HAVE_OPENCV
#include <opencv2/core/core.hpp>
#include <opencv2/features2d/features2d.hpp>
#endif

#include "Frame.h"
#include "KeyFrame.h"
#include "MapPoint.h"

namespace ORB_SLAM2 {

class ORBmatcher {
 public:
  ORBmatcher(float nnratio = 0.6, bool checkOri = true);

  // Hamming distance between two ORB descriptors (1 x 32 CV_8U rows)
  static int DescriptorDistance(const cv::Mat& a, const cv::Mat& b);

  // Tracking: local map points projected by Frame::isInFrustum -> F.mvpMapPoints; returns the number of matches
  int SearchByProjection(Frame& F, const std::vector<MapPoint*>& vpMapPoints, const float th = 3);

  // Tracking: map points of the last frame projected with the motion model
  int SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, const float th, const bool bMono);

  // Relocalisation: map points of a keyframe projected into the frame
  int SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, const std::set<MapPoint*>& sAlreadyFound, const float th,
                         const int ORBdist);

  // Loop detection: map points projected with a similarity transformation
  int SearchByProjection(KeyFrame* pKF, cv::Mat Scw, const std::vector<MapPoint*>& vpPoints, std::vector<MapPoint*>& vpMatched,
                         int th);

  // Brute force constrained to features under the same vocabulary node (relocalisation / loop detection)
  int SearchByBoW(KeyFrame* pKF, Frame& F, std::vector<MapPoint*>& vpMapPointMatches);
  int SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12);

  // Monocular map initialisation
  int SearchForInitialization(Frame& F1, Frame& F2, std::vector<cv::Point2f>& vbPrevMatched, std::vector<int>& vnMatches12,
                              int windowSize = 10);

  // Triangulation of new map points under the epipolar constraint
  int SearchForTriangulation(KeyFrame* pKF1, KeyFrame* pKF2, cv::Mat F12,
                             std::vector<std::pair<std::size_t, std::size_t>>& vMatchedPairs, const bool bOnlyStereo);

  // Matches between map points seen in KF1 and KF2 under a Sim3 [s12*R12|t12]
  int SearchBySim3(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12, const float& s12, const cv::Mat& R12,
                   const cv::Mat& t12, const float th);

  // Project map points into a keyframe and fuse duplicates
  int Fuse(KeyFrame* pKF, const std::vector<MapPoint*>& vpMapPoints, const float th = 3.0);
  int Fuse(KeyFrame* pKF, cv::Mat Scw, const std::vector<MapPoint*>& vpPoints, float th, std::vector<MapPoint*>& vpReplacePoint);

 public:
  static const int TH_LOW;
  static const int TH_HIGH;
  static const int HISTO_LENGTH;

 protected:
  bool CheckDistEpipolarLine(const cv::KeyPoint& kp1, const cv::KeyPoint& kp2, const cv::Mat& F12, const KeyFrame* pKF);
  float RadiusByViewingCos(const float& viewCos);
  void ComputeThreeMaxima(std::vector<int>* histo, const int L, int& ind1, int& ind2, int& ind3);

  float mfNNratio;
  bool mbCheckOrientation;
};

}  // namespace ORB_SLAM2

#endif  // ORBMATCHER_H
