// ORBextractor.cc -- drop-in ORB_SLAM2::ORBextractor over liborbfe's C ABI (MI355X).
// Behaviour follows Source/Libraries/ORB_SLAM2/src/ORBextractor.cc: constructor tables :407-464 (computed by
// the library, read back through the getters), operator() :978-1039 (empty image -> silent return; zero
// keypoints -> descriptors released; keypoints cleared and refilled), ComputePyramid's bordered level
// buffers :1041-1065.
#include "ORBextractor.h"

#include <stdio.h>
#include <string.h>

#include "../../../include/orbfe.h"

namespace ORB_SLAM2 {

static const int EDGE_THRESHOLD = 19;

ORBextractor::ORBextractor(int _nfeatures, float _scaleFactor, int _nlevels, int _iniThFAST, int _minThFAST)
    : nfeatures(_nfeatures), scaleFactor(_scaleFactor), nlevels(_nlevels), iniThFAST(_iniThFAST),
      minThFAST(_minThFAST), mpImpl(nullptr), mbDownloadPyramid(true) {
  orbfe_params p;
  p.n_features = _nfeatures;
  p.scale_factor = _scaleFactor;
  p.n_levels = _nlevels;
  p.ini_th_fast = _iniThFAST;
  p.min_th_fast = _minThFAST;
  const int rc = orbfe_extractor_create(&p, -1, &mpImpl);
  if (rc != ORBFE_OK) {
    fprintf(stderr, "ORBextractor: orbfe_extractor_create failed (%d): %s\n", rc, orbfe_last_error());
    mpImpl = nullptr;
    nlevels = _nlevels > 0 ? _nlevels : 0;
  }
  mvScaleFactor.assign(nlevels, 1.f);
  mvInvScaleFactor.assign(nlevels, 1.f);
  mvLevelSigma2.assign(nlevels, 1.f);
  mvInvLevelSigma2.assign(nlevels, 1.f);
  mnFeaturesPerLevel.assign(nlevels, 0);
  mvImagePyramid.resize(nlevels);
  mvPadded.resize(nlevels);
  if (mpImpl) {
    orbfe_extractor_scale_factors(mpImpl, mvScaleFactor.data());
    orbfe_extractor_inv_scale_factors(mpImpl, mvInvScaleFactor.data());
    orbfe_extractor_sigma2(mpImpl, mvLevelSigma2.data());
    orbfe_extractor_inv_sigma2(mpImpl, mvInvLevelSigma2.data());
    orbfe_extractor_features_per_level(mpImpl, mnFeaturesPerLevel.data());
  }
}

ORBextractor::~ORBextractor() {
  if (mpImpl) orbfe_extractor_destroy(mpImpl);
}

static inline int reflect101(int p, int len) {
  if (len == 1) return 0;
  while (p < 0 || p >= len) p = p < 0 ? -p : 2 * (len - 1) - p;
  return p;
}

void ORBextractor::operator()(cv::InputArray _image, cv::InputArray /*_mask*/, std::vector<cv::KeyPoint>& _keypoints,
                              cv::OutputArray _descriptors) {
  if (_image.empty()) return;  // :981-982
  cv::Mat image = _image.getMat();
  if (image.type() != CV_8UC1 || !mpImpl) {
    fprintf(stderr, "ORBextractor: %s\n", mpImpl ? "image must be CV_8UC1" : "no device handle");
    _keypoints.clear();
    _descriptors.release();
    return;
  }
  int cap = 0;
  orbfe_extractor_max_keypoints(mpImpl, image.cols, image.rows, &cap);
  // staging owned by the object and reused across calls (one caller thread per extractor, as in the reference)
  if (mvStageKeys.size() < (size_t)cap * sizeof(orbfe_keypoint)) mvStageKeys.resize((size_t)cap * sizeof(orbfe_keypoint));
  if (mvStageDesc.size() < (size_t)cap * 32) mvStageDesc.resize((size_t)cap * 32);
  orbfe_keypoint* kps = reinterpret_cast<orbfe_keypoint*>(mvStageKeys.data());
  uint8_t* desc = mvStageDesc.data();
  int n = 0;
  const int rc = orbfe_extract(mpImpl, image.ptr(0), image.cols, image.rows, (int)image.step, kps, desc, cap, &n);
  if (rc != ORBFE_OK) {
    fprintf(stderr, "ORBextractor: orbfe_extract failed (%d): %s\n", rc, orbfe_last_error());
    n = 0;
  }
  _keypoints.clear();
  if (n == 0) {
    _descriptors.release();  // :999-1000
  } else {
    _descriptors.create(n, 32, CV_8U);
    cv::Mat d = _descriptors.getMat();
    if (d.isContinuous()) memcpy(d.ptr(0), desc, (size_t)n * 32);
    else for (int i = 0; i < n; i++) memcpy(d.ptr(i), desc + (size_t)i * 32, 32);
    _keypoints.resize((size_t)n);
    static_assert(sizeof(cv::KeyPoint) == sizeof(orbfe_keypoint), "cv::KeyPoint layout");
    memcpy((void*)_keypoints.data(), kps, sizeof(orbfe_keypoint) * (size_t)n);
  }
  if (mbDownloadPyramid && rc == ORBFE_OK) {
    std::vector<uint8_t*> dst((size_t)nlevels, nullptr);
    std::vector<int> dstStride((size_t)nlevels, 0);
    for (int level = 0; level < nlevels; ++level) {
      int w = 0, h = 0;
      if (orbfe_pyramid_level_size(mpImpl, image.cols, image.rows, level, &w, &h) != ORBFE_OK) return;
      cv::Mat& temp = mvPadded[level];
      temp.create(h + 2 * EDGE_THRESHOLD, w + 2 * EDGE_THRESHOLD, CV_8U);
      mvImagePyramid[level] = temp(cv::Rect(EDGE_THRESHOLD, EDGE_THRESHOLD, w, h));
      dst[level] = mvImagePyramid[level].ptr(0);
      dstStride[level] = (int)mvImagePyramid[level].step;
    }
    if (orbfe_pyramid_levels(mpImpl, dst.data(), dstStride.data()) != ORBFE_OK) {
      fprintf(stderr, "ORBextractor: orbfe_pyramid_levels failed: %s\n", orbfe_last_error());
      return;
    }
    for (int level = 0; level < nlevels; ++level) {
      cv::Mat& roi = mvImagePyramid[level];
      cv::Mat& temp = mvPadded[level];
      const int w = roi.cols, h = roi.rows;
      // BORDER_REFLECT_101 frame around the level (:1057-1062)
      for (int y = -EDGE_THRESHOLD; y < h + EDGE_THRESHOLD; y++) {
        const uint8_t* src = roi.ptr(0) + (ptrdiff_t)reflect101(y, h) * (ptrdiff_t)roi.step;
        uint8_t* dstp = temp.ptr(y + EDGE_THRESHOLD);
        if (y < 0 || y >= h) memcpy(dstp + EDGE_THRESHOLD, src, (size_t)w);
        for (int x = 0; x < EDGE_THRESHOLD; x++) {
          dstp[x] = src[reflect101(x - EDGE_THRESHOLD, w)];
          dstp[EDGE_THRESHOLD + w + x] = src[reflect101(w + x, w)];
        }
      }
    }
  }
}

bool ORBextractor::Prepare(int width, int height) {
  if (!mpImpl) return false;
  const int rc = orbfe_extractor_prepare(mpImpl, width, height, 1);
  if (rc != ORBFE_OK) fprintf(stderr, "ORBextractor: orbfe_extractor_prepare failed (%d): %s\n", rc, orbfe_last_error());
  return rc == ORBFE_OK;
}

}  // namespace ORB_SLAM2
