// ORBextractor.h -- drop-in replacement for the reference header
// Source/Libraries/ORB_SLAM2/include/ORBextractor.h:20-108.  Same namespace, class name, public methods,
// enum, public data member (mvImagePyramid) and protected scale tables, so Tracking.cc (news the extractors,
// L/src/Tracking.cc:112-127) and Frame.cc (calls operator() and the getters, L/src/Frame.cc:78-84,265-270)
// compile and link unchanged.  The work happens on an MI355X through liborbfe's C ABI (include/orbfe.h).
#ifndef ORBEXTRACTOR_H
#define ORBEXTRACTOR_H

#include <list>
#include <vector>

#ifdef ORBFE_HAVE_OPENCV
#include <opencv2/opencv.hpp>
#else
#include "cvlite.h"
#endif

struct orbfe_extractor;

namespace ORB_SLAM2 {

// Kept for source compatibility (declared in the reference header, used only inside its ORBextractor.cc).
class ExtractorNode {
 public:
  ExtractorNode() : bNoMore(false) {}
  std::vector<cv::KeyPoint> vKeys;
  cv::Point2i UL, UR, BL, BR;
  std::list<ExtractorNode>::iterator lit;
  bool bNoMore;
};

class ORBextractor {
 public:
  enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };

  ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST);
  ~ORBextractor();
  ORBextractor(const ORBextractor&) = delete;
  ORBextractor& operator=(const ORBextractor&) = delete;

  // Compute the ORB features and descriptors on an image.  Mask is ignored, as in the reference.
  // Errors never throw: a failure logs to stderr and yields zero keypoints.
  void operator()(cv::InputArray image, cv::InputArray mask, std::vector<cv::KeyPoint>& keypoints,
                  cv::OutputArray descriptors);

  int inline GetLevels() { return nlevels; }
  float inline GetScaleFactor() { return scaleFactor; }
  std::vector<float> inline GetScaleFactors() { return mvScaleFactor; }
  std::vector<float> inline GetInverseScaleFactors() { return mvInvScaleFactor; }
  std::vector<float> inline GetScaleSigmaSquares() { return mvLevelSigma2; }
  std::vector<float> inline GetInverseScaleSigmaSquares() { return mvInvLevelSigma2; }

  // Host copies of the 8-bit pyramid of the last call (read by Frame::ComputeStereoMatches,
  // L/src/Frame.cc:483,567-589).  Each level is a view into a buffer with a 19-pixel REFLECT_101 border,
  // like the reference's.  Disable the per-call download with SetPyramidDownload(false) when the stereo
  // association runs on the device (orbfe_stereo_match_device).
  std::vector<cv::Mat> mvImagePyramid;

  // ---- additions (not in the reference)
  void SetPyramidDownload(bool on) { mbDownloadPyramid = on; }
  // Warm-up for images of this size (plan, work space, code objects, launch graph: orbfe_extractor_prepare) -- e.g. from
  // Tracking::Tracking once Camera.width / Camera.height are read.  Optional: the first operator() does the same work otherwise.
  bool Prepare(int width, int height);
  orbfe_extractor* Handle() const { return mpImpl; }

 protected:
  int nfeatures;
  double scaleFactor;
  int nlevels;
  int iniThFAST;
  int minThFAST;

  std::vector<int> mnFeaturesPerLevel;
  std::vector<float> mvScaleFactor;
  std::vector<float> mvInvScaleFactor;
  std::vector<float> mvLevelSigma2;
  std::vector<float> mvInvLevelSigma2;

  orbfe_extractor* mpImpl;
  bool mbDownloadPyramid;
  std::vector<cv::Mat> mvPadded;  // owners of the bordered level buffers
  std::vector<unsigned char> mvStageKeys, mvStageDesc;  // results of the last call before they are cut to size
};

}  // namespace ORB_SLAM2

#endif
