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
//                       [--prepare 1] [--gpus 1] [--gather none|root|all] [--gather-dump file.bin]
//
// --gpus N is the batched-sequence mode: the frame range is cut into N contiguous chunks
// (orbfe_shard_range), one host thread per GPU runs the loop above on its chunk with its own extractors, matcher handle, decode
// pool and stream (the first frame of a chunk has no last frame to track against -- map-dependent matching does not shard), and
// the left images' padded records {n; keypoints[cap]; descriptors[cap][32]} are gathered over RCCL through the C ABI
// (orbfe_gather_create_all = ncclCommInitAll, orbfe_gather_records): to GPU 0 (--gather root) or to every GPU (--gather all).
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

// Decoded pairs, produced out of order by the pool, consumed in order by the tracking loop.  Frame i lives in slot
// (i - first) % depth; a decoder may fill it once frame i - depth has been consumed.
struct FrameRing {
  struct Slot {
    cv::Mat left, right;
    long frame = -1;      // frame whose images the slot holds
    bool ok = false;
  };
  std::vector<Slot> slots;
  std::mutex mu;
  std::condition_variable cv_ready, cv_free;
  long first = 0, consumed = -1;     // consumed: last frame the tracker has finished reading
  std::atomic<long> next{0};
  std::atomic<bool> stop{false};
  std::vector<std::thread> pool;
  double decode_total_s = 0;

  void start(int threads, int depth, long begin, long end, const std::vector<std::string>& left, const std::vector<std::string>& right) {
    slots.resize((size_t)depth);
    first = begin; consumed = begin - 1; next.store(begin);
    for (int t = 0; t < threads; t++)
      pool.emplace_back([this, depth, end, &left, &right] {
        for (;;) {
          const long i = next.fetch_add(1);
          if (i >= end || stop.load()) return;
          Slot& s = slots[(size_t)((i - first) % depth)];
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
            s.ok = ok; s.frame = i;
            decode_total_s += dt;
          }
          cv_ready.notify_all();
        }
      });
  }
  Slot& wait(long i) {
    Slot& s = slots[(size_t)((i - first) % (long)slots.size())];
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
  double k = ceil(p * (double)sorted.size()) - 1;
  k = std::max(0.0, std::min((double)sorted.size() - 1, k));
  return sorted[(size_t)k];
}
static double seconds_since(std::chrono::steady_clock::time_point a) {
  return std::chrono::duration_cast<std::chrono::duration<double>>(std::chrono::steady_clock::now() - a).count();
}

struct Options {
  int nFeatures = 2000, maxFrames = 0, decodeThreads = 8, prefetch = 16, prepare = 1, gpus = 1;
  float bf = 386.1448f, fx = 718.856f, fy = 718.856f, cx = 607.1928f, cy = 185.2157f, th = 7.0f;   // KITTI00-02.yaml
  std::string dumpPath, gather = "default", gatherDump;
};

// What one GPU's host thread produces for its chunk of the sequence
struct Shard {
  int device = 0, begin = 0, end = 0, rc = 0;
  std::vector<float> times;                 // per frame: tracking time (extraction -> search), seconds
  std::vector<double> phase[3];             // per frame: extraction (two threads), ComputeStereoMatches, SearchByProjection
  long nKeys = 0, nStereo = 0, nTracked = 0;
  double wall_s = 0, decode_s = 0, wait_s = 0, prepare_ms = 0, gather_ms = -1;
  std::string dump;                         // per-frame records of --dump, frame order
  // the left images' padded records for the gather: n[frames], keypoints[frames][cap], descriptors[frames][cap][32]
  int cap = 0;
  std::vector<int32_t> rec_n;
  std::vector<orbfe_keypoint> rec_k;
  std::vector<uint8_t> rec_d;
};

// The loop of stereo_kitti.cc:88-106 over frames [sh.begin, sh.end) on the calling thread's device.
static void RunShard(const Options& o, const std::vector<std::string>& vstrImageLeft, const std::vector<std::string>& vstrImageRight,
                     bool keepRecords, int chunkFrames, Shard& sh) {
  const int nImages = sh.end - sh.begin;
  // Tracking::Tracking (L/src/Tracking.cc:112-127): one extractor per eye, the matcher of TrackWithMotionModel (:859)
  std::unique_ptr<ORBextractor> mpORBextractorLeft(new ORBextractor(o.nFeatures, 1.2f, 8, 20, 7));
  std::unique_ptr<ORBextractor> mpORBextractorRight(new ORBextractor(o.nFeatures, 1.2f, 8, 20, 7));
  if (!mpORBextractorLeft->Handle() || !mpORBextractorRight->Handle()) { sh.rc = 3; return; }   // no HIP device: no CPU fallback
  mpORBextractorLeft->SetPyramidDownload(false);    // ComputeStereoMatches reads the pyramids in HBM
  mpORBextractorRight->SetPyramidDownload(false);
  ORBmatcher matcher(0.9f, true);
  int w0 = 0, h0 = 0;
  if (nImages > 0 && orbfe_png_info(vstrImageLeft[(size_t)sh.begin].c_str(), &w0, &h0) != ORBFE_OK) {
    fprintf(stderr, "cannot read %s\n", vstrImageLeft[(size_t)sh.begin].c_str());
    sh.rc = 65;
    return;
  }
  // warm-up for the sequence's image size: plan, work space, code objects, launch graphs, this thread's matcher handle
  if (o.prepare && nImages > 0) {
    const auto tp = std::chrono::steady_clock::now();
    if (orbfe_frontend_prepare(mpORBextractorLeft->Handle(), mpORBextractorRight->Handle(), w0, h0, 0) != ORBFE_OK) {
      fprintf(stderr, "orbfe_frontend_prepare: %s\n", orbfe_last_error());
      sh.rc = 3;
      return;
    }
    sh.prepare_ms = 1e3 * seconds_since(tp);
  }
  if (keepRecords && nImages > 0) {
    if (orbfe_extractor_max_keypoints(mpORBextractorLeft->Handle(), w0, h0, &sh.cap) != ORBFE_OK) { sh.rc = 3; return; }
    sh.rec_n.assign((size_t)chunkFrames, 0);
    sh.rec_k.assign((size_t)chunkFrames * sh.cap, orbfe_keypoint());
    sh.rec_d.assign((size_t)chunkFrames * sh.cap * 32, 0);
  }
  FrameRing ring;
  if (o.decodeThreads > 0) ring.start(o.decodeThreads, std::max(o.prefetch, 2), sh.begin, sh.end, vstrImageLeft, vstrImageRight);
  const auto tSequence = std::chrono::steady_clock::now();
  sh.times.assign((size_t)nImages, 0.f);

  Frame mLastFrame;
  std::vector<std::unique_ptr<MapPoint>> lastPoints;          // UpdateLastFrame's temporal points
  std::unordered_map<MapPoint*, int> lastIndex;               // point -> keypoint index in the last frame
  cv::Mat imLeftOwn, imRightOwn;
  for (int ni = sh.begin; ni < sh.end; ni++) {
    const auto t0 = std::chrono::steady_clock::now();
    bool loaded;
    FrameRing::Slot* slot = nullptr;
    if (o.decodeThreads > 0) {
      slot = &ring.wait(ni);
      loaded = slot->ok;
    } else {
      loaded = ReadGray(vstrImageLeft[(size_t)ni], imLeftOwn) && ReadGray(vstrImageRight[(size_t)ni], imRightOwn);
    }
    if (!loaded) {
      fprintf(stderr, "\nFailed to load image at: %s\n", vstrImageLeft[(size_t)ni].c_str());
      ring.finish();
      sh.rc = 65;
      return;
    }
    cv::Mat& imLeft = slot ? slot->left : imLeftOwn;
    cv::Mat& imRight = slot ? slot->right : imRightOwn;
    const auto t1 = std::chrono::steady_clock::now();
    (o.decodeThreads > 0 ? sh.wait_s : sh.decode_s) += std::chrono::duration_cast<std::chrono::duration<double>>(t1 - t0).count();

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
    mCurrentFrame.mbf = o.bf;
    mCurrentFrame.mb = o.bf / o.fx;
    {
      ORBextractor* exL = mpORBextractorLeft.get();
      ORBextractor* exR = mpORBextractorRight.get();
      const int dev = sh.device;
      std::thread threadLeft([&, exL, dev] { orbfe_set_device(dev); (*exL)(imLeft, cv::Mat(), mCurrentFrame.mvKeys, mCurrentFrame.mDescriptors); });
      std::thread threadRight([&, exR, dev] { orbfe_set_device(dev); (*exR)(imRight, cv::Mat(), mCurrentFrame.mvKeysRight, mCurrentFrame.mDescriptorsRight); });
      threadLeft.join();
      threadRight.join();
    }
    sh.phase[0].push_back(seconds_since(t1));
    const float imCols = (float)imLeft.cols, imRows = (float)imLeft.rows;
    if (slot) ring.release(ni);   // the extractors have read the images: the slot may be refilled
    const auto tS = std::chrono::steady_clock::now();
    mCurrentFrame.N = (int)mCurrentFrame.mvKeys.size();
    mCurrentFrame.mvKeysUn = mCurrentFrame.mvKeys;                       // UndistortKeyPoints with k1 == 0 (:~700)
    orbfe_host::ComputeStereoMatches(mCurrentFrame, mpORBextractorLeft.get(), mpORBextractorRight.get());
    sh.phase[1].push_back(seconds_since(tS));
    const auto tM = std::chrono::steady_clock::now();
    mCurrentFrame.mvpMapPoints.assign((size_t)mCurrentFrame.N, static_cast<MapPoint*>(NULL));
    mCurrentFrame.mvbOutlier.assign((size_t)mCurrentFrame.N, false);
    Frame::mnMinX = 0.0f; Frame::mnMaxX = imCols;                         // ComputeImageBounds without distortion
    Frame::mnMinY = 0.0f; Frame::mnMaxY = imRows;
    mCurrentFrame.mTcw = cv::Mat::eye(4, 4, CV_32F);                      // zero velocity: mVelocity * mLastFrame.mTcw
    mCurrentFrame.mOw = cv::Mat::zeros(3, 1, CV_32F);

    // ---- Tracking::TrackWithMotionModel's search   L/src/Tracking.cc:857-884
    int nmatches = 0;
    if (ni > sh.begin) nmatches = matcher.SearchByProjection(mCurrentFrame, mLastFrame, o.th, false);

    sh.phase[2].push_back(seconds_since(tM));
    sh.times[(size_t)(ni - sh.begin)] = (float)seconds_since(t1);

    const int N = mCurrentFrame.N;
    std::vector<int32_t> assigned((size_t)N, -1);
    for (int i = 0; i < N; i++) {
      MapPoint* pMP = mCurrentFrame.mvpMapPoints[(size_t)i];
      if (pMP) assigned[(size_t)i] = lastIndex.at(pMP);
    }
    sh.nKeys += N;
    sh.nTracked += nmatches;
    for (int i = 0; i < N; i++) sh.nStereo += mCurrentFrame.mvDepth[(size_t)i] > 0;
    if (!o.dumpPath.empty()) {
      const int32_t hdr[2] = {N, nmatches};
      sh.dump.append((const char*)hdr, sizeof(hdr));
      sh.dump.append((const char*)mCurrentFrame.mvKeys.data(), sizeof(cv::KeyPoint) * (size_t)N);
      for (int i = 0; i < N; i++) sh.dump.append((const char*)mCurrentFrame.mDescriptors.ptr(i), 32);
      sh.dump.append((const char*)mCurrentFrame.mvuRight.data(), sizeof(float) * (size_t)N);
      sh.dump.append((const char*)mCurrentFrame.mvDepth.data(), sizeof(float) * (size_t)N);
      sh.dump.append((const char*)assigned.data(), sizeof(int32_t) * (size_t)N);
    }
    if (keepRecords && N <= sh.cap) {
      const size_t f = (size_t)(ni - sh.begin);
      sh.rec_n[f] = N;
      static_assert(sizeof(cv::KeyPoint) == sizeof(orbfe_keypoint), "cv::KeyPoint layout");
      memcpy(&sh.rec_k[f * sh.cap], mCurrentFrame.mvKeys.data(), sizeof(orbfe_keypoint) * (size_t)N);
      for (int i = 0; i < N; i++) memcpy(&sh.rec_d[(f * sh.cap + (size_t)i) * 32], mCurrentFrame.mDescriptors.ptr(i), 32);
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
  sh.wall_s = seconds_since(tSequence);
  ring.finish();
  if (o.decodeThreads > 0) sh.decode_s = ring.decode_total_s;
  orbfe_thread_release();   // this thread's implicit matcher handle (the thread ends with its shard in the batched mode)
}

int main(int argc, char** argv) {
  if (argc < 2) {
    fprintf(stderr, "Usage: %s path_to_sequence [--features N] [--max-frames N] [--dump file] [--gpus N] [--gather none|root|all]\n", argv[0]);
    return 64;
  }
  Options o;
  for (int i = 2; i + 1 < argc; i += 2) {
    const std::string k = argv[i];
    const char* v = argv[i + 1];
    if (k == "--features") o.nFeatures = atoi(v);
    else if (k == "--max-frames") o.maxFrames = atoi(v);
    else if (k == "--dump") o.dumpPath = v;
    else if (k == "--bf") o.bf = (float)atof(v);
    else if (k == "--fx") o.fx = (float)atof(v);
    else if (k == "--fy") o.fy = (float)atof(v);
    else if (k == "--cx") o.cx = (float)atof(v);
    else if (k == "--cy") o.cy = (float)atof(v);
    else if (k == "--th") o.th = (float)atof(v);
    else if (k == "--decode-threads") o.decodeThreads = atoi(v);
    else if (k == "--prefetch") o.prefetch = std::max(1, atoi(v));
    else if (k == "--prepare") o.prepare = atoi(v);
    else if (k == "--gpus") o.gpus = std::max(1, atoi(v));
    else if (k == "--gather") o.gather = v;
    else if (k == "--gather-dump") o.gatherDump = v;
    else { fprintf(stderr, "unknown option %s\n", k.c_str()); return 64; }
  }
  if (o.gather == "default") o.gather = o.gpus > 1 ? "root" : "none";
  if (o.gather != "none" && o.gather != "root" && o.gather != "all") { fprintf(stderr, "--gather none|root|all\n"); return 64; }
  std::vector<std::string> vstrImageLeft, vstrImageRight;
  std::vector<double> vTimestamps;
  if (!LoadImages(argv[1], vstrImageLeft, vstrImageRight, vTimestamps)) {
    fprintf(stderr, "no times.txt under %s\n", argv[1]);
    return 66;
  }
  int nImages = (int)vstrImageLeft.size();
  if (o.maxFrames > 0) nImages = std::min(nImages, o.maxFrames);
  int nDev = 0;
  if (orbfe_device_count(&nDev) != ORBFE_OK || nDev < 1) { fprintf(stderr, "no HIP device: %s\n", orbfe_last_error()); return 3; }
  if (o.gpus > nDev) { fprintf(stderr, "--gpus %d but %d device(s) visible\n", o.gpus, nDev); return 64; }

  Frame::fx = o.fx; Frame::fy = o.fy; Frame::cx = o.cx; Frame::cy = o.cy;
  Frame::invfx = 1.0f / o.fx; Frame::invfy = 1.0f / o.fy;

  const int G = o.gpus;
  const bool gathering = o.gather != "none";
  const int chunkFrames = (nImages + G - 1) / G;      // every rank contributes equally sized (padded) records
  std::vector<Shard> shards((size_t)G);
  std::vector<orbfe_gather*> comms((size_t)G, nullptr);
  if (gathering && orbfe_gather_create_all(G, nullptr, comms.data()) != ORBFE_OK) {
    fprintf(stderr, "orbfe_gather_create_all: %s\n", orbfe_last_error());
    return 3;
  }
  Options oShard = o;
  oShard.decodeThreads = o.decodeThreads > 0 ? std::max(1, o.decodeThreads / G) : 0;
  printf("\n-------\nStart processing sequence ...\nImages in the sequence: %d\n\n", nImages);
  const auto tAll = std::chrono::steady_clock::now();
  // gathered records as rank 0 (and, with --gather all, every rank) receives them
  std::vector<int32_t> all_n;
  std::vector<orbfe_keypoint> all_k;
  std::vector<uint8_t> all_d;
  int gatherOk = 1;
  auto worker = [&](int g) {
    Shard& sh = shards[(size_t)g];
    sh.device = g;
    orbfe_shard_range(nImages, g, G, &sh.begin, &sh.end);
    if (orbfe_set_device(g) != ORBFE_OK) { sh.rc = 3; return; }
    RunShard(oShard, vstrImageLeft, vstrImageRight, gathering, chunkFrames, sh);
    if (!gathering) return;
    // one exchange at the end: this rank's records -> HBM -> RCCL -> (rank 0 | everyone) -> host.  A rank that failed still
    // takes part with empty records: the collective needs every rank.
    if (sh.cap == 0) { sh.cap = 1; sh.rec_n.assign((size_t)chunkFrames, 0); sh.rec_k.assign((size_t)chunkFrames, orbfe_keypoint()); sh.rec_d.assign((size_t)chunkFrames * 32, 0); }
    const size_t bn = sizeof(int32_t) * (size_t)chunkFrames, bk = sizeof(orbfe_keypoint) * (size_t)chunkFrames * sh.cap,
                 bd = (size_t)32 * chunkFrames * sh.cap;
    const bool receives = o.gather == "all" || g == 0;
    void *dn = nullptr, *dk = nullptr, *dd = nullptr, *an = nullptr, *ak = nullptr, *ad = nullptr;
    int rc = orbfe_device_malloc(bn, &dn) | orbfe_device_malloc(bk, &dk) | orbfe_device_malloc(bd, &dd);
    if (receives) rc |= orbfe_device_malloc(bn * G, &an) | orbfe_device_malloc(bk * G, &ak) | orbfe_device_malloc(bd * G, &ad);
    rc |= orbfe_device_upload(dn, sh.rec_n.data(), bn) | orbfe_device_upload(dk, sh.rec_k.data(), bk) | orbfe_device_upload(dd, sh.rec_d.data(), bd);
    const auto tg = std::chrono::steady_clock::now();
    if (rc == ORBFE_OK)
      rc = orbfe_gather_records(comms[(size_t)g], (const int32_t*)dn, (const orbfe_keypoint*)dk, (const uint8_t*)dd, chunkFrames, sh.cap,
                                o.gather == "all" ? ORBFE_GATHER_ALL : ORBFE_GATHER_ROOT, (int32_t*)an, (orbfe_keypoint*)ak, (uint8_t*)ad, nullptr);
    if (rc == ORBFE_OK) rc = orbfe_gather_sync(comms[(size_t)g]);
    sh.gather_ms = 1e3 * seconds_since(tg);
    if (rc == ORBFE_OK && g == 0) {
      all_n.resize((size_t)chunkFrames * G); all_k.resize((size_t)chunkFrames * G * sh.cap); all_d.resize((size_t)chunkFrames * G * sh.cap * 32);
      rc = orbfe_device_download(all_n.data(), an, bn * G) | orbfe_device_download(all_k.data(), ak, bk * G) | orbfe_device_download(all_d.data(), ad, bd * G);
    }
    if (rc != ORBFE_OK) { fprintf(stderr, "gather on GPU %d: %s\n", g, orbfe_last_error()); gatherOk = 0; }
    for (void* p : {dn, dk, dd, an, ak, ad}) orbfe_device_free(p);
  };
  if (G == 1) {
    worker(0);
  } else {
    std::vector<std::thread> ths;
    for (int g = 0; g < G; g++) ths.emplace_back(worker, g);
    for (auto& t : ths) t.join();
  }
  const double all_s = seconds_since(tAll);
  for (auto c : comms) orbfe_gather_destroy(c);
  for (const Shard& sh : shards)
    if (sh.rc) return sh.rc;
  if (!gatherOk) return 4;

  if (!o.dumpPath.empty()) {
    FILE* dump = fopen(o.dumpPath.c_str(), "wb");
    if (!dump) { fprintf(stderr, "cannot write %s\n", o.dumpPath.c_str()); return 73; }
    for (const Shard& sh : shards) fwrite(sh.dump.data(), 1, sh.dump.size(), dump);
    fclose(dump);
  }
  if (gathering) {
    // the gathered records, frame by frame, must be the per-frame results (rank order = frame order for contiguous chunks)
    const int cap = shards[0].cap;
    long bad = 0, frames = 0;
    for (int g = 0; g < G; g++)
      for (int f = 0; f < shards[(size_t)g].end - shards[(size_t)g].begin; f++, frames++) {
        const size_t src = (size_t)f, dst = (size_t)g * chunkFrames + (size_t)f;
        const int n = shards[(size_t)g].rec_n[src];
        bad += all_n[dst] != n || memcmp(&all_k[dst * cap], &shards[(size_t)g].rec_k[src * cap], sizeof(orbfe_keypoint) * (size_t)n) != 0 ||
               memcmp(&all_d[dst * cap * 32], &shards[(size_t)g].rec_d[src * cap * 32], (size_t)32 * n) != 0;
      }
    double gms = 0;
    for (const Shard& sh : shards) gms = std::max(gms, sh.gather_ms);
    printf("gather (%s, %d GPU%s, RCCL through the C ABI): %ld frames' records, %.1f MB per GPU, %.3f ms, %ld frame(s) differ\n", o.gather.c_str(), G,
           G > 1 ? "s" : "", frames, (sizeof(int32_t) + (sizeof(orbfe_keypoint) + 32.0) * cap) * chunkFrames / 1e6, gms, bad);
    if (bad) return 5;
    if (!o.gatherDump.empty()) {
      FILE* gd = fopen(o.gatherDump.c_str(), "wb");
      if (!gd) { fprintf(stderr, "cannot write %s\n", o.gatherDump.c_str()); return 73; }
      for (int g = 0; g < G; g++)
        for (int f = 0; f < shards[(size_t)g].end - shards[(size_t)g].begin; f++) {
          const size_t dst = (size_t)g * chunkFrames + (size_t)f;
          const int32_t n = all_n[dst];
          fwrite(&n, sizeof(n), 1, gd);
          fwrite(&all_k[dst * cap], sizeof(orbfe_keypoint), (size_t)n, gd);
          fwrite(&all_d[dst * cap * 32], 32, (size_t)n, gd);
        }
      fclose(gd);
    }
  }

  // Tracking time statistics (stereo_kitti.cc:137-144): every frame of every shard counts, the first ones included
  std::vector<float> vTimesTrack;
  std::vector<double> tPhase[3];
  long nKeys = 0, nStereo = 0, nTracked = 0;
  double decode_s = 0, wait_s = 0, wall_s = 0, prepare_ms = 0;
  for (const Shard& sh : shards) {
    vTimesTrack.insert(vTimesTrack.end(), sh.times.begin(), sh.times.end());
    for (int p = 0; p < 3; p++) tPhase[p].insert(tPhase[p].end(), sh.phase[p].begin(), sh.phase[p].end());
    nKeys += sh.nKeys; nStereo += sh.nStereo; nTracked += sh.nTracked;
    decode_s += sh.decode_s; wait_s += sh.wait_s;
    wall_s = std::max(wall_s, sh.wall_s);
    prepare_ms = std::max(prepare_ms, sh.prepare_ms);
  }
  if (o.prepare) printf("front end prepared in %.1f ms\n", prepare_ms);
  std::sort(vTimesTrack.begin(), vTimesTrack.end());
  float totaltime = 0;
  for (float t : vTimesTrack) totaltime += t;
  printf("-------\n\n");
  printf("median tracking time: %g\n", vTimesTrack[(size_t)nImages / 2]);
  printf("mean tracking time: %g\n", totaltime / nImages);
  printf("p95 tracking time: %g\n", percentile(vTimesTrack, 0.95));
  printf("p99 tracking time: %g\n", percentile(vTimesTrack, 0.99));
  printf("max tracking time: %g\n", vTimesTrack.back());
  printf("sequence: %d frames in %.4f s = %.1f frames/s end to end (decode threads %d, prefetch %d; decode %.4f s of CPU time = %.3f ms per pair; "
         "tracking thread waited %.4f s for images)\n", nImages, wall_s, nImages / wall_s, oShard.decodeThreads * (oShard.decodeThreads > 0 ? G : 1),
         o.prefetch, decode_s, 1e3 * decode_s / std::max(nImages, 1), wait_s);
  if (G > 1) {
    printf("batched mode: %d GPUs, %d frames per GPU, whole run incl. warm-up and gather %.4f s; per GPU:", G, chunkFrames, all_s);
    for (const Shard& sh : shards) printf(" [%d: frames %d..%d, %.1f frames/s]", sh.device, sh.begin, sh.end, (sh.end - sh.begin) / std::max(sh.wall_s, 1e-9));
    printf("\n");
  }
  printf("frames: %d, keypoints/left image: %.1f, stereo matches/frame: %.1f, tracked/frame: %.1f\n", nImages,
         (double)nKeys / nImages, (double)nStereo / nImages, (double)nTracked / std::max(nImages - G, 1));
  for (auto& v : tPhase) std::sort(v.begin(), v.end());
  printf("median per phase [ms]: ORBextractor x2 (two threads) %.4f, ComputeStereoMatches %.4f, SearchByProjection(cur,last) %.4f\n",
         1e3 * tPhase[0][tPhase[0].size() / 2], 1e3 * tPhase[1][tPhase[1].size() / 2], 1e3 * tPhase[2][tPhase[2].size() / 2]);
  return 0;
}
