// stereo_kitti.cc -- the front-end half of Source/Examples/Stereo/stereo_kitti.cc:36-150 on the C++ drop-in classes.
//
// The reference's example loads a KITTI sequence (LoadImages, :152-210), pushes every stereo pair through
// System::TrackStereo and prints the median / mean tracking time (:137-144).  This driver keeps that loop and runs, per
// pair, what Tracking does with it BEFORE pose optimisation -- on the classes Tracking.cc / Frame.cc link against
// (csrc/host/ORBextractor.{h,cc}, csrc/host/ORBmatcher.{h,cc}):
//
//   Frame::Frame (L/src/Frame.cc:66-127)      scale tables from the left extractor, ORBextractor::operator() for the left and
//                                             the right image on two std::threads (:91-94), UndistortKeyPoints (no
//                                             distortion in the KITTI settings: mvKeysUn = mvKeys), ComputeStereoMatches
//                                             (orbfe_host::ComputeStereoMatches: the pyramids stay in HBM)
//   Tracking::TrackWithMotionModel (L/src/Tracking.cc:857-884)   UpdateLastFrame's stereo points (UnprojectStereo of every
//                                             keypoint with a depth, Frame.cc:668-679), SetPose(velocity * last pose) with
//                                             zero velocity, ORBmatcher::SearchByProjection(cur, last, th = 7, false)
//
// Images are read with the library's zlib PNG reader (orbfe_png_read_gray); Frame / MapPoint are the light headers under
// tests/cpp/mock (same member names as the reference's; the real ones need OpenCV).  Pose optimisation, local mapping and
// loop closing are out of scope.  --dump writes per-frame records that tests/test_dropin_cpp.py compares with the oracle.
//
// Input: the reference loads each pair with cv::imread right before it tracks it (stereo_kitti.cc:88-106), so its frame rate is
// decode + track.  Here a pool of --decode-threads host threads decodes --prefetch pairs ahead into a ring of frame slots while
// the main thread tracks: the report gives the end-to-end rate of the sequence next to the tracking times, and what the decode
// costs.  --decode-threads 0 is the reference's load-then-track loop.  Before the first frame the front end is warmed up for
// the sequence's image size (orbfe_frontend_prepare): no first-frame spike is hidden in, or excluded from, the statistics.
//
//   usage: stereo_kitti <sequence_dir> [--features 2000] [--max-frames N] [--dump file.bin] [--bf 386.1448] [--fx 718.856]
//                       [--fy 718.856] [--cx 607.1928] [--cy 185.2157] [--th 7] [--decode-threads 8] [--prefetch 16]
//                       [--prepare 1]
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../include/orbfe.h"
#include "../refactored_orb_slam2_amd/csrc/host/ORBextractor.h"
#include "../refactored_orb_slam2_amd/csrc/host/ORBmatcher.h"
#include "../refactored_orb_slam2_amd/csrc/host/ORBmatcher_hip.h"
#include "Frame.h"
#include "MapPoint.h"

using namespace ORB_SLAM2;

// LoadImages of stereo_kitti.cc:152-210: times.txt, image_0/%06d.png, image_1/%06d.png
static bool LoadImages(const std::string& dir, std::vector<std::string>& left, std::vector<std::string>& right,
                       std::vector<double>& times) {
  FILE* f = fopen((dir + "/times.txt").c_str(), "r");
  if (!f) return false;
  double t;
  while (fscanf(f, "%lf", &t) == 1) times.push_back(t);
  fclose(f);
  char name[32];
  for (size_t i = 0; i < times.size(); i++) {
    snprintf(name, sizeof(name), "%06zu.png", i);
    left.push_back(dir + "/image_0/" + name);
    right.push_back(dir + "/image_1/" + name);
  }
  return !times.empty();
}

static bool ReadGray(const std::string& path, cv::Mat& im) {
  int w = 0, h = 0;
  if (orbfe_png_info(path.c_str(), &w, &h) != ORBFE_OK) return false;
  im.create(h, w, CV_8U);
  return orbfe_png_read_gray(path.c_str(), im.ptr(0), (int)im.step, h, &w, &h) == ORBFE_OK;
}

// Decoded pairs, produced out of order by the pool, consumed in order by the tracking loop.  Frame i lives in slot i % depth; a
// decoder may fill it once frame i - depth has been consumed.
struct FrameRing {
  struct Slot {
    cv::Mat left, right;
    long frame = -1;      // frame whose images the slot holds
    bool ok = false;
    double decode_s = 0;
  };
  std::vector<Slot> slots;
  std::mutex mu;
  std::condition_variable cv_ready, cv_free;
  long consumed = -1;     // last frame the tracker has finished reading
  std::atomic<long> next{0};
  std::atomic<bool> stop{false};
  std::vector<std::thread> pool;
  double decode_total_s = 0;

  void start(int threads, int depth, int n_frames, const std::vector<std::string>& left, const std::vector<std::string>& right) {
    slots.resize((size_t)depth);
    for (int t = 0; t < threads; t++)
      pool.emplace_back([this, depth, n_frames, &left, &right] {
        for (;;) {
          const long i = next.fetch_add(1);
          if (i >= n_frames || stop.load()) return;
          Slot& s = slots[(size_t)(i % depth)];
          {
            std::unique_lock<std::mutex> lk(mu);
            cv_free.wait(lk, [&] { return stop.load() || i - depth <= consumed; });
            if (stop.load()) return;
          }
          const auto t0 = std::chrono::steady_clock::now();
          const bool ok = ReadGray(left[(size_t)i], s.left) && ReadGray(right[(size_t)i], s.right);
          const double dt = std::chrono::duration_cast<std::chrono::duration<double>>(std::chrono::steady_clock::now() - t0).count();
          {
            std::lock_guard<std::mutex> lk(mu);
            s.ok = ok; s.decode_s = dt; s.frame = i;
            decode_total_s += dt;
          }
          cv_ready.notify_all();
        }
      });
  }
  Slot& wait(long i) {
    Slot& s = slots[(size_t)(i % (long)slots.size())];
    std::unique_lock<std::mutex> lk(mu);
    cv_ready.wait(lk, [&] { return s.frame == i; });
    return s;
  }
  void release(long i) {
    { std::lock_guard<std::mutex> lk(mu); consumed = i; }
    cv_free.notify_all();
  }
  void finish() {
    stop.store(true);
    cv_free.notify_all();
    for (auto& t : pool) t.join();
    pool.clear();
  }
};

static double percentile(const std::vector<float>& sorted, double p) {
  if (sorted.empty()) return 0;
  const size_t k = (size_t)std::min<double>((double)sorted.size() - 1, ceil(p * (double)sorted.size()) - 1 < 0 ? 0 : ceil(p * (double)sorted.size()) - 1);
  return sorted[k];
}

int main(int argc, char** argv) {
  if (argc < 2) {
    fprintf(stderr, "Usage: %s path_to_sequence [--features N] [--max-frames N] [--dump file]\n", argv[0]);
    return 64;
  }
  int nFeatures = 2000, maxFrames = 0, decodeThreads = 8, prefetch = 16, prepare = 1;
  float bf = 386.1448f, fx = 718.856f, fy = 718.856f, cx = 607.1928f, cy = 185.2157f, th = 7.0f;   // KITTI00-02.yaml
  std::string dumpPath;
  for (int i = 2; i + 1 < argc; i += 2) {
    const std::string k = argv[i];
    const char* v = argv[i + 1];
    if (k == "--features") nFeatures = atoi(v);
    else if (k == "--max-frames") maxFrames = atoi(v);
    else if (k == "--dump") dumpPath = v;
    else if (k == "--bf") bf = (float)atof(v);
    else if (k == "--fx") fx = (float)atof(v);
    else if (k == "--fy") fy = (float)atof(v);
    else if (k == "--cx") cx = (float)atof(v);
    else if (k == "--cy") cy = (float)atof(v);
    else if (k == "--th") th = (float)atof(v);
    else if (k == "--decode-threads") decodeThreads = atoi(v);
    else if (k == "--prefetch") prefetch = std::max(1, atoi(v));
    else if (k == "--prepare") prepare = atoi(v);
    else { fprintf(stderr, "unknown option %s\n", k.c_str()); return 64; }
  }
  std::vector<std::string> vstrImageLeft, vstrImageRight;
  std::vector<double> vTimestamps;
  if (!LoadImages(argv[1], vstrImageLeft, vstrImageRight, vTimestamps)) {
    fprintf(stderr, "no times.txt under %s\n", argv[1]);
    return 66;
  }
  int nImages = (int)vstrImageLeft.size();
  if (maxFrames > 0) nImages = std::min(nImages, maxFrames);

  // Tracking::Tracking (L/src/Tracking.cc:112-127): one extractor per eye, the matcher of TrackWithMotionModel (:859)
  ORBextractor* mpORBextractorLeft = new ORBextractor(nFeatures, 1.2f, 8, 20, 7);
  ORBextractor* mpORBextractorRight = new ORBextractor(nFeatures, 1.2f, 8, 20, 7);
  if (!mpORBextractorLeft->Handle() || !mpORBextractorRight->Handle()) return 3;   // no HIP device: no CPU fallback
  mpORBextractorLeft->SetPyramidDownload(false);    // ComputeStereoMatches reads the pyramids in HBM
  mpORBextractorRight->SetPyramidDownload(false);
  ORBmatcher matcher(0.9f, true);

  Frame::fx = fx; Frame::fy = fy; Frame::cx = cx; Frame::cy = cy;
  Frame::invfx = 1.0f / fx; Frame::invfy = 1.0f / fy;

  // warm-up for the sequence's image size: plan, work space, code objects, launch graphs, the tracking thread's matcher handle
  if (prepare && nImages > 0) {
    int w0 = 0, h0 = 0;
    if (orbfe_png_info(vstrImageLeft[0].c_str(), &w0, &h0) != ORBFE_OK) { fprintf(stderr, "cannot read %s\n", vstrImageLeft[0].c_str()); return 65; }
    const auto tp = std::chrono::steady_clock::now();
    if (orbfe_frontend_prepare(mpORBextractorLeft->Handle(), mpORBextractorRight->Handle(), w0, h0, 0) != ORBFE_OK) {
      fprintf(stderr, "orbfe_frontend_prepare: %s\n", orbfe_last_error());
      return 3;
    }
    printf("front end prepared for %dx%d in %.1f ms\n", w0, h0,
           1e3 * std::chrono::duration_cast<std::chrono::duration<double>>(std::chrono::steady_clock::now() - tp).count());
  }
  FrameRing ring;
  if (decodeThreads > 0) ring.start(decodeThreads, std::max(prefetch, 2), nImages, vstrImageLeft, vstrImageRight);

  printf("\n-------\nStart processing sequence ...\nImages in the sequence: %d\n\n", nImages);
  const auto tSequence = std::chrono::steady_clock::now();
  double inlineDecode = 0, waitDecode = 0;
  std::vector<float> vTimesTrack((size_t)nImages, 0.f);
  FILE* dump = dumpPath.empty() ? nullptr : fopen(dumpPath.c_str(), "wb");
  if (!dumpPath.empty() && !dump) { fprintf(stderr, "cannot write %s\n", dumpPath.c_str()); return 73; }

  Frame mLastFrame;
  std::vector<std::unique_ptr<MapPoint>> lastPoints;          // UpdateLastFrame's temporal points
  std::unordered_map<MapPoint*, int> lastIndex;               // point -> keypoint index in the last frame
  long nKeys = 0, nStereo = 0, nTracked = 0;
  std::vector<double> tPhase[3];   // per frame: extraction (two threads), ComputeStereoMatches, SearchByProjection incl. frame set-up
  cv::Mat imLeftOwn, imRightOwn;
  for (int ni = 0; ni < nImages; ni++) {
    const auto t0 = std::chrono::steady_clock::now();
    bool loaded;
    FrameRing::Slot* slot = nullptr;
    if (decodeThreads > 0) {
      slot = &ring.wait(ni);
      loaded = slot->ok;
    } else {
      loaded = ReadGray(vstrImageLeft[(size_t)ni], imLeftOwn) && ReadGray(vstrImageRight[(size_t)ni], imRightOwn);
    }
    if (!loaded) {
      fprintf(stderr, "\nFailed to load image at: %s\n", vstrImageLeft[(size_t)ni].c_str());
      ring.finish();
      return 65;
    }
    cv::Mat& imLeft = slot ? slot->left : imLeftOwn;
    cv::Mat& imRight = slot ? slot->right : imRightOwn;
    const auto t1 = std::chrono::steady_clock::now();
    (decodeThreads > 0 ? waitDecode : inlineDecode) += std::chrono::duration_cast<std::chrono::duration<double>>(t1 - t0).count();
    auto since = [](std::chrono::steady_clock::time_point a) {
      return std::chrono::duration_cast<std::chrono::duration<double>>(std::chrono::steady_clock::now() - a).count();
    };

    // ---- Frame::Frame(imLeft, imRight, ...)   L/src/Frame.cc:66-127
    Frame mCurrentFrame;
    mCurrentFrame.mnId = (unsigned long)ni;
    mCurrentFrame.mnScaleLevels = mpORBextractorLeft->GetLevels();
    mCurrentFrame.mfScaleFactor = mpORBextractorLeft->GetScaleFactor();
    mCurrentFrame.mfLogScaleFactor = logf(mCurrentFrame.mfScaleFactor);
    mCurrentFrame.mvScaleFactors = mpORBextractorLeft->GetScaleFactors();
    mCurrentFrame.mvInvScaleFactors = mpORBextractorLeft->GetInverseScaleFactors();
    mCurrentFrame.mvLevelSigma2 = mpORBextractorLeft->GetScaleSigmaSquares();
    mCurrentFrame.mvInvLevelSigma2 = mpORBextractorLeft->GetInverseScaleSigmaSquares();
    mCurrentFrame.mbf = bf;
    mCurrentFrame.mb = bf / fx;
    {
      std::thread threadLeft([&] { (*mpORBextractorLeft)(imLeft, cv::Mat(), mCurrentFrame.mvKeys, mCurrentFrame.mDescriptors); });
      std::thread threadRight([&] { (*mpORBextractorRight)(imRight, cv::Mat(), mCurrentFrame.mvKeysRight, mCurrentFrame.mDescriptorsRight); });
      threadLeft.join();
      threadRight.join();
    }
    tPhase[0].push_back(since(t1));
    const float imCols = (float)imLeft.cols, imRows = (float)imLeft.rows;
    if (slot) ring.release(ni);   // the extractors have read the images: the slot may be refilled
    const auto tS = std::chrono::steady_clock::now();
    mCurrentFrame.N = (int)mCurrentFrame.mvKeys.size();
    mCurrentFrame.mvKeysUn = mCurrentFrame.mvKeys;                       // UndistortKeyPoints with k1 == 0 (:~700)
    orbfe_host::ComputeStereoMatches(mCurrentFrame, mpORBextractorLeft, mpORBextractorRight);
    tPhase[1].push_back(since(tS));
    const auto tM = std::chrono::steady_clock::now();
    mCurrentFrame.mvpMapPoints.assign((size_t)mCurrentFrame.N, static_cast<MapPoint*>(NULL));
    mCurrentFrame.mvbOutlier.assign((size_t)mCurrentFrame.N, false);
    Frame::mnMinX = 0.0f; Frame::mnMaxX = imCols;                         // ComputeImageBounds without distortion
    Frame::mnMinY = 0.0f; Frame::mnMaxY = imRows;
    mCurrentFrame.mTcw = cv::Mat::eye(4, 4, CV_32F);                      // zero velocity: mVelocity * mLastFrame.mTcw
    mCurrentFrame.mOw = cv::Mat::zeros(3, 1, CV_32F);

    // ---- Tracking::TrackWithMotionModel's search   L/src/Tracking.cc:857-884
    int nmatches = 0;
    if (ni > 0) nmatches = matcher.SearchByProjection(mCurrentFrame, mLastFrame, th, false);

    tPhase[2].push_back(since(tM));
    const auto t2 = std::chrono::steady_clock::now();
    vTimesTrack[(size_t)ni] = (float)std::chrono::duration_cast<std::chrono::duration<double>>(t2 - t1).count();

    const int N = mCurrentFrame.N;
    std::vector<int32_t> assigned((size_t)N, -1);
    for (int i = 0; i < N; i++) {
      MapPoint* pMP = mCurrentFrame.mvpMapPoints[(size_t)i];
      if (pMP) assigned[(size_t)i] = lastIndex.at(pMP);
    }
    nKeys += N;
    nTracked += nmatches;
    for (int i = 0; i < N; i++) nStereo += mCurrentFrame.mvDepth[(size_t)i] > 0;
    if (dump) {
      const int32_t hdr[2] = {N, nmatches};
      fwrite(hdr, sizeof(hdr), 1, dump);
      fwrite(mCurrentFrame.mvKeys.data(), sizeof(cv::KeyPoint), (size_t)N, dump);
      for (int i = 0; i < N; i++) fwrite(mCurrentFrame.mDescriptors.ptr(i), 32, 1, dump);
      fwrite(mCurrentFrame.mvuRight.data(), sizeof(float), (size_t)N, dump);
      fwrite(mCurrentFrame.mvDepth.data(), sizeof(float), (size_t)N, dump);
      fwrite(assigned.data(), sizeof(int32_t), (size_t)N, dump);
    }

    // ---- the frame becomes mLastFrame; Tracking::UpdateLastFrame (L/src/Tracking.cc:~810-855) gives its stereo points a
    //      MapPoint at Frame::UnprojectStereo(i) (L/src/Frame.cc:668-679; identity pose: x3Dw = x3Dc)
    mLastFrame = mCurrentFrame;
    lastPoints.clear();
    lastIndex.clear();
    const cv::Mat zero3 = cv::Mat::zeros(3, 1, CV_32F);
    for (int i = 0; i < N; i++) {
      mLastFrame.mvpMapPoints[(size_t)i] = nullptr;
      const float z = mLastFrame.mvDepth[(size_t)i];
      if (!(z > 0)) continue;
      const float u = mLastFrame.mvKeysUn[(size_t)i].pt.x, v = mLastFrame.mvKeysUn[(size_t)i].pt.y;
      cv::Mat x3D(3, 1, CV_32F);
      x3D.at<float>(0) = (u - Frame::cx) * z * Frame::invfx;
      x3D.at<float>(1) = (v - Frame::cy) * z * Frame::invfy;
      x3D.at<float>(2) = z;
      lastPoints.emplace_back(new MapPoint(x3D, zero3, mLastFrame.mDescriptors.row(i), 0.f, 0.f));
      MapPoint* pMP = lastPoints.back().get();
      pMP->nObs = 1;   // a point that blocks its match (Observations() > 0), as a map point of the local map does
      mLastFrame.mvpMapPoints[(size_t)i] = pMP;
      lastIndex[pMP] = i;
    }
  }
  const double sequence_s = std::chrono::duration_cast<std::chrono::duration<double>>(std::chrono::steady_clock::now() - tSequence).count();
  ring.finish();
  if (dump) fclose(dump);

  // Tracking time statistics (stereo_kitti.cc:137-144): every frame counts, the first one included
  std::sort(vTimesTrack.begin(), vTimesTrack.end());
  float totaltime = 0;
  for (int ni = 0; ni < nImages; ni++) totaltime += vTimesTrack[(size_t)ni];
  printf("-------\n\n");
  printf("median tracking time: %g\n", vTimesTrack[(size_t)nImages / 2]);
  printf("mean tracking time: %g\n", totaltime / nImages);
  printf("p95 tracking time: %g\n", percentile(vTimesTrack, 0.95));
  printf("p99 tracking time: %g\n", percentile(vTimesTrack, 0.99));
  printf("max tracking time: %g\n", vTimesTrack.back());
  printf("sequence: %d frames in %.4f s = %.1f frames/s end to end (decode threads %d, prefetch %d; decode %.4f s of CPU time = %.3f ms per pair; "
         "tracking thread waited %.4f s for images)\n", nImages, sequence_s, nImages / sequence_s, decodeThreads, prefetch,
         decodeThreads > 0 ? ring.decode_total_s : inlineDecode,
         1e3 * (decodeThreads > 0 ? ring.decode_total_s : inlineDecode) / std::max(nImages, 1), decodeThreads > 0 ? waitDecode : 0.0);
  printf("frames: %d, keypoints/left image: %.1f, stereo matches/frame: %.1f, tracked/frame: %.1f\n", nImages,
         (double)nKeys / nImages, (double)nStereo / nImages, (double)nTracked / std::max(nImages - 1, 1));
  for (auto& v : tPhase) std::sort(v.begin(), v.end());
  printf("median per phase [ms]: ORBextractor x2 (two threads) %.4f, ComputeStereoMatches %.4f, SearchByProjection(cur,last) %.4f\n",
         1e3 * tPhase[0][tPhase[0].size() / 2], 1e3 * tPhase[1][tPhase[1].size() / 2], 1e3 * tPhase[2][tPhase[2].size() / 2]);
  delete mpORBextractorLeft;
  delete mpORBextractorRight;
  return 0;
}
