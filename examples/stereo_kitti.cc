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
//                       [--batch F] [--slots 3] [--preload 0|1|2|3] [--repeat R] [--outputs all|matches|counts]
//
// --batch F is the batched pipeline from a C++ host (include/orbfe.h: orbfe_pipeline_*): chunks of F pairs go through the device
// batch API -- decode pool -> the pipeline's pinned pitched input -> H2D -> 2 x ORBextractor -> ComputeStereoMatches ->
// UnprojectStereo -> SearchByProjection(cur, last) -> D2H -- with --slots buffer sets, so that decoding, the copies and the kernels
// of neighbouring chunks overlap; per frame the same records as the loop above (--dump writes the same format).  --preload 1
// decodes the whole sequence into host memory before the clock starts (what is timed then: one host copy per frame into the pinned
// slots, PCIe, the kernels); --preload 3 leaves them in the slots' DEVICE input blocks after the first chunks (orbfe_pipeline_submit_resident:
// the handle at its kernels' rate, what a device-side producer gets); --preload 2 leaves the frames resident in the pinned slots after the first chunks (no host work per
// frame: the rate of the pipeline itself, PCIe included; results are not those of the sequence order: no --dump with it);
// --repeat R walks the sequence R times (steady-state rates from a short directory).  With --gpus N every GPU thread runs one
// pipeline on its chunk of the frame range and the left records of every chunk are gathered straight from HBM
// (orbfe_pipeline_gather), overlapped with the next chunk.
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
    { std::lock_guard<std::mutex> lk(mu); stop.store(true); }   // under the lock: a decoder between its predicate and its wait must not miss it
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
  int batch = 0, slots = 3, preload = 0, repeat = 1, outputs = 0;   // outputs: orbfe_pipeline_config.output_mask
  float bf = 386.1448f, fx = 718.856f, fy = 718.856f, cx = 607.1928f, cy = 185.2157f, th = 7.0f;   // KITTI00-02.yaml
  std::string dumpPath, gather = "default", gatherDump;
};

// The GPU threads of one process agree on something before a collective (a rank that cannot take part must not leave the others
// waiting inside RCCL): arrive with a verdict, leave with everybody's
struct HostBarrier {
  std::mutex mu;
  std::condition_variable cv;
  int n = 1, arrived = 0, generation = 0;
  bool bad = false, result = false;
  bool arrive(bool failed) {   // returns true when ANY participant failed
    std::unique_lock<std::mutex> lk(mu);
    bad = bad || failed;
    const int gen = generation;
    if (++arrived == n) { result = bad; bad = false; arrived = 0; generation++; cv.notify_all(); return result; }
    cv.wait(lk, [&] { return generation != gen; });
    return result;
  }
};

// What one GPU's host thread produces for its chunk of the sequence
struct Shard {
  int device = 0, begin = 0, end = 0, rc = 0;
  std::vector<float> times;                 // per frame: tracking time (extraction -> search), seconds
  std::vector<double> phase[3];             // per frame: extraction (two threads), ComputeStereoMatches, SearchByProjection
  long nKeys = 0, nStereo = 0, nTracked = 0;
  double wall_s = 0, decode_s = 0, wait_s = 0, prepare_ms = 0, gather_ms = -1;
  long gatherBad = 0;                       // batched pipeline: gathered rows that differ from the rank's own results
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
  struct ThreadRelease { ~ThreadRelease() { orbfe_thread_release(); } } releaseOnEveryPath;   // this thread's implicit matcher handle
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
  if (keepRecords) {   // sh.cap: the record capacity every rank agreed on before the threads started
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
}

// ---- the batched pipeline (--batch F) ---------------------------------------------------------------------------------------
// Frames of the (possibly repeated) sequence, decoded by a pool straight into the pipeline's pinned, pitched input slots.  Chunk k
// (frames begin + k F ...) lives in slot k % S; a decoder may fill it once chunk k - S has been given back.
struct ChunkFeeder {
  int F = 0, S = 0, w = 0, h = 0;
  long begin = 0, end = 0, nReal = 0;
  std::vector<orbfe_pipeline_input_view> in;     // per slot
  const std::vector<std::string>*left = nullptr, *right = nullptr;
  const std::vector<std::vector<uint8_t>>*preL = nullptr, *preR = nullptr;   // --preload: decoded frames, w bytes per row
  std::mutex mu;
  std::condition_variable cv_done, cv_free;
  long released = 0;                 // chunks given back so far
  std::vector<int> done;             // frames decoded per chunk
  bool failed = false, stop = false, pinnedResident = false;
  std::atomic<long> next{0};
  std::vector<std::thread> pool;
  double decode_total_s = 0;

  void start(int threads) {
    done.assign((size_t)((end - begin + F - 1) / F + 1), 0);
    next.store(begin);
    for (int t = 0; t < threads; t++)
      pool.emplace_back([this] {
        for (;;) {
          const long i = next.fetch_add(1);
          if (i >= end) return;
          const long k = (i - begin) / F, j = (i - begin) % F;
          {
            std::unique_lock<std::mutex> lk(mu);
            cv_free.wait(lk, [&] { return stop || k - S < released; });
            if (stop) return;
          }
          const orbfe_pipeline_input_view& v = in[(size_t)(k % S)];
          uint8_t* dl = v.left + (size_t)j * v.image_bytes;
          uint8_t* dr = v.right + (size_t)j * v.image_bytes;
          const size_t src = (size_t)(i % nReal);
          const auto t0 = std::chrono::steady_clock::now();
          bool ok = true;
          if (preL && pinnedResident && k >= S) {
            // --preload 2: the slots were filled by the first S chunks and are submitted again as they are (the pipeline's rate with
            // its input already in pinned memory: no per-frame host work left; the results are those of the resident frames)
          } else if (preL) {
            for (int y = 0; y < h; y++) {
              memcpy(dl + (size_t)y * v.pitch, (*preL)[src].data() + (size_t)y * w, (size_t)w);
              memcpy(dr + (size_t)y * v.pitch, (*preR)[src].data() + (size_t)y * w, (size_t)w);
            }
          } else {
            int ww = 0, hh = 0;
            ok = orbfe_png_read_gray((*left)[src].c_str(), dl, v.pitch, h, &ww, &hh) == ORBFE_OK && ww == w && hh == h &&
                 orbfe_png_read_gray((*right)[src].c_str(), dr, v.pitch, h, &ww, &hh) == ORBFE_OK && ww == w && hh == h;
          }
          const double dt = seconds_since(t0);
          {
            std::lock_guard<std::mutex> lk(mu);
            done[(size_t)k]++;
            failed = failed || !ok;
            decode_total_s += dt;
          }
          cv_done.notify_all();
        }
      });
  }
  bool wait_chunk(long k, int n) {
    std::unique_lock<std::mutex> lk(mu);
    cv_done.wait(lk, [&] { return done[(size_t)k] >= n || failed; });
    return !failed;
  }
  void release_chunk(long k) {
    { std::lock_guard<std::mutex> lk(mu); released = k + 1; }
    cv_free.notify_all();
  }
  void finish() {
    { std::lock_guard<std::mutex> lk(mu); stop = true; }
    cv_free.notify_all();
    for (auto& t : pool) t.join();
    pool.clear();
  }
};

// What the batched mode's gather needs of a rank, whatever happened to it: every rank joins every chunk's collective
struct GatherPlan {
  orbfe_gather* comm = nullptr;
  int mode = ORBFE_GATHER_ROOT, G = 1, chunks = 0, cap = 0;
  bool receives = false, keep = false;     // keep: --gather-dump wants every rank's records on the host (rank 0)
};

// A rank without a pipeline (no frames to read, creation failed) still takes part in every chunk's collective, with empty records
static int JoinGatherEmpty(const GatherPlan& gp, int F) {
  const size_t bn = sizeof(int32_t) * (size_t)F, bk = sizeof(orbfe_keypoint) * (size_t)F * gp.cap, bd = (size_t)32 * F * gp.cap;
  void *dn = nullptr, *dk = nullptr, *dd = nullptr, *an = nullptr, *ak = nullptr, *ad = nullptr;
  int rc = orbfe_device_malloc(bn, &dn) | orbfe_device_malloc(bk, &dk) | orbfe_device_malloc(bd, &dd);
  if (gp.receives) rc |= orbfe_device_malloc(bn * gp.G, &an) | orbfe_device_malloc(bk * gp.G, &ak) | orbfe_device_malloc(bd * gp.G, &ad);
  std::vector<uint8_t> zeros(std::max(bn, std::max(bk, bd)), 0);
  if (rc == ORBFE_OK) rc = orbfe_device_upload(dn, zeros.data(), bn) | orbfe_device_upload(dk, zeros.data(), bk) | orbfe_device_upload(dd, zeros.data(), bd);
  for (int k = 0; k < gp.chunks; k++) {
    // even after a failure above the collective is entered (with whatever buffers exist): the peers are waiting in it
    (void)orbfe_gather_records(gp.comm, (const int32_t*)dn, (const orbfe_keypoint*)dk, (const uint8_t*)dd, F, gp.cap, gp.mode, (int32_t*)an,
                               (orbfe_keypoint*)ak, (uint8_t*)ad, nullptr);
    (void)orbfe_gather_sync(gp.comm);
  }
  for (void* q : {dn, dk, dd, an, ak, ad}) orbfe_device_free(q);
  return rc;
}

// Gathered records as rank 0 keeps them for --gather-dump and the comparison with the ranks' own results
struct GatheredFrames {
  std::mutex mu;
  std::vector<int32_t> n;                       // per global frame (-1: not received)
  std::vector<std::vector<orbfe_keypoint>> k;
  std::vector<std::vector<uint8_t>> d;
};

// Frames [sh.begin, sh.end) in chunks of o.batch pairs through one orbfe_pipeline on the calling thread's device.
static void RunShardBatched(const Options& o, const std::vector<std::string>& vstrImageLeft, const std::vector<std::string>& vstrImageRight,
                            const std::vector<std::vector<uint8_t>>* preL, const std::vector<std::vector<uint8_t>>* preR, int w0, int h0,
                            long nReal, const GatherPlan& gp, const std::vector<int>& shardBegin, GatheredFrames* gathered, Shard& sh) {
  const int F = o.batch, S = o.slots;
  const long nLocal = sh.end - sh.begin;
  const int K = gp.comm ? gp.chunks : (int)((nLocal + F - 1) / F);
  orbfe_pipeline_config cfg;
  memset(&cfg, 0, sizeof(cfg));
  cfg.extractor.n_features = o.nFeatures; cfg.extractor.scale_factor = 1.2f; cfg.extractor.n_levels = 8;
  cfg.extractor.ini_th_fast = 20; cfg.extractor.min_th_fast = 7;
  cfg.width = w0; cfg.height = h0; cfg.batch = F; cfg.slots = S;
  cfg.fx = o.fx; cfg.fy = o.fy; cfg.cx = o.cx; cfg.cy = o.cy; cfg.bf = o.bf; cfg.th = o.th; cfg.check_orientation = 1;
  cfg.output_mask = o.outputs;
  orbfe_pipeline* pl = nullptr;
  const auto tp = std::chrono::steady_clock::now();
  if (orbfe_pipeline_create(&cfg, sh.device, &pl) != ORBFE_OK) {
    fprintf(stderr, "orbfe_pipeline_create on GPU %d: %s\n", sh.device, orbfe_last_error());
    sh.rc = 3;
    if (gp.comm) JoinGatherEmpty(gp, F);
    return;
  }
  sh.prepare_ms = 1e3 * seconds_since(tp);
  orbfe_pipeline_output_view ov;
  orbfe_pipeline_output(pl, 0, &ov);
  const int cap = ov.cap;
  sh.cap = cap;
  // receive buffers of the gather, one set per slot
  const size_t bn = sizeof(int32_t) * (size_t)F, bk = sizeof(orbfe_keypoint) * (size_t)F * cap, bd = (size_t)32 * F * cap;
  std::vector<void*> an((size_t)S, nullptr), ak((size_t)S, nullptr), ad((size_t)S, nullptr);
  bool gatherOk = true;
  if (gp.comm && gp.receives)
    for (int s = 0; s < S; s++)
      gatherOk = gatherOk && orbfe_device_malloc(bn * gp.G, &an[(size_t)s]) == ORBFE_OK && orbfe_device_malloc(bk * gp.G, &ak[(size_t)s]) == ORBFE_OK &&
                 orbfe_device_malloc(bd * gp.G, &ad[(size_t)s]) == ORBFE_OK;
  ChunkFeeder feeder;
  feeder.F = F; feeder.S = S; feeder.w = w0; feeder.h = h0; feeder.begin = sh.begin; feeder.end = sh.end; feeder.nReal = nReal;
  feeder.left = &vstrImageLeft; feeder.right = &vstrImageRight; feeder.preL = preL; feeder.preR = preR;
  feeder.pinnedResident = o.preload >= 2;
  feeder.in.resize((size_t)S);
  for (int s = 0; s < S; s++) orbfe_pipeline_input(pl, s, &feeder.in[(size_t)s]);
  const auto tSequence = std::chrono::steady_clock::now();
  feeder.start(std::max(1, o.decodeThreads));
  sh.times.assign((size_t)nLocal, 0.f);
  std::vector<std::chrono::steady_clock::time_point> tSubmit((size_t)std::max(K, 1));
  std::vector<int32_t> hn;
  std::vector<orbfe_keypoint> hk;
  std::vector<uint8_t> hd;

  auto finish_chunk = [&](int k) {
    const int slot = k % S;
    const int n = (int)std::max(0L, std::min((long)F, nLocal - (long)k * F));
    if (orbfe_pipeline_wait(pl, slot) != ORBFE_OK) { fprintf(stderr, "orbfe_pipeline_wait: %s\n", orbfe_last_error()); sh.rc = 3; }
    const double dt = seconds_since(tSubmit[(size_t)k]);
    orbfe_pipeline_output(pl, slot, &ov);
    for (int j = 0; j < n; j++) {
      const int N = ov.n_left[j];
      const size_t row = (size_t)j * cap;
      sh.times[(size_t)k * F + (size_t)j] = (float)(dt / n);
      sh.nKeys += N;
      sh.nTracked += ov.n_tracked[j];
      sh.nStereo += ov.n_stereo[j];
      if (!o.dumpPath.empty()) {
        const int32_t hdr[2] = {N, ov.n_tracked[j]};
        sh.dump.append((const char*)hdr, sizeof(hdr));
        sh.dump.append((const char*)(ov.kps_left + row), sizeof(orbfe_keypoint) * (size_t)N);
        sh.dump.append((const char*)(ov.desc_left + row * 32), (size_t)32 * N);
        sh.dump.append((const char*)(ov.u_right + row), sizeof(float) * (size_t)N);
        sh.dump.append((const char*)(ov.depth + row), sizeof(float) * (size_t)N);
        sh.dump.append((const char*)(ov.assigned + row), sizeof(int32_t) * (size_t)N);
      }
    }
    if (gp.comm) {
      if (orbfe_pipeline_gather_wait(pl, slot) != ORBFE_OK) gatherOk = false;
      if (gp.receives && gatherOk) {
        // what arrived in this rank's own rows must be what the pipeline handed to the host; rank 0 keeps everything for --gather-dump
        const bool all = gp.keep && sh.device == 0;
        const int g0 = all ? 0 : sh.device, g1 = all ? gp.G : sh.device + 1;
        hn.resize((size_t)F * gp.G);
        if (orbfe_device_download(hn.data(), an[(size_t)slot], bn * gp.G) != ORBFE_OK) gatherOk = false;
        for (int g = g0; g < g1 && gatherOk; g++) {
          hk.resize((size_t)F * cap); hd.resize((size_t)F * cap * 32);
          if (orbfe_device_download(hk.data(), (const uint8_t*)ak[(size_t)slot] + bk * g, bk) != ORBFE_OK ||
              orbfe_device_download(hd.data(), (const uint8_t*)ad[(size_t)slot] + bd * g, bd) != ORBFE_OK) { gatherOk = false; break; }
          for (int j = 0; j < F; j++) {
            const int32_t nn = hn[(size_t)g * F + (size_t)j];
            if (g == sh.device) {
              const int32_t mine = j < n ? ov.n_left[j] : 0;
              if (nn != mine || (mine > 0 && (memcmp(&hk[(size_t)j * cap], ov.kps_left + (size_t)j * cap, sizeof(orbfe_keypoint) * (size_t)mine) != 0 ||
                                              memcmp(&hd[(size_t)j * cap * 32], ov.desc_left + (size_t)j * cap * 32, (size_t)32 * mine) != 0)))
                sh.gatherBad++;
            }
            if (all && gathered) {
              const long fr = (long)shardBegin[(size_t)g] + (long)k * F + j;
              if (fr < shardBegin[(size_t)g + 1]) {
                std::lock_guard<std::mutex> lk(gathered->mu);
                gathered->n[(size_t)fr] = nn;
                gathered->k[(size_t)fr].assign(&hk[(size_t)j * cap], &hk[(size_t)j * cap] + std::max(nn, 0));
                gathered->d[(size_t)fr].assign(&hd[(size_t)j * cap * 32], &hd[(size_t)j * cap * 32] + (size_t)32 * std::max(nn, 0));
              }
            }
          }
        }
      }
    }
    if (n > 0) feeder.release_chunk(k);
  };

  // Chunk k is decoded into slot k % S, which chunk k - S must have given back first: finish that one before waiting for the
  // decoders (with one slot the loop is sequential); otherwise a chunk's results are consumed while the next one runs.
  int nextToFinish = 0;
  auto finish_upto = [&](int k) { while (nextToFinish <= k) finish_chunk(nextToFinish++); };
  for (int k = 0; k < K; k++) {
    const int n = (int)std::max(0L, std::min((long)F, nLocal - (long)k * F));
    finish_upto(k - S);
    if (n > 0) {
      const auto tw = std::chrono::steady_clock::now();
      if (!feeder.wait_chunk(k, n)) { fprintf(stderr, "\nFailed to load an image of chunk %d\n", k); sh.rc = 65; }
      sh.wait_s += seconds_since(tw);
    }
    tSubmit[(size_t)k] = std::chrono::steady_clock::now();
    // a chunk whose images failed to load is still submitted (with no frames): the collective below needs every rank
    // --preload 3: after the first S chunks the frames are where their uploads put them, in the slots' DEVICE input blocks
    const bool resident = o.preload == 3 && k >= S;
    if ((resident ? orbfe_pipeline_submit_resident(pl, k % S, sh.rc ? 0 : n, k > 0) : orbfe_pipeline_submit(pl, k % S, sh.rc ? 0 : n, k > 0)) != ORBFE_OK) { fprintf(stderr, "orbfe_pipeline_submit: %s\n", orbfe_last_error()); sh.rc = 3; }
    if (gp.comm) {
      const auto tg = std::chrono::steady_clock::now();
      if (orbfe_pipeline_gather(pl, k % S, gp.comm, gp.mode, (int32_t*)an[(size_t)(k % S)], (orbfe_keypoint*)ak[(size_t)(k % S)],
                                (uint8_t*)ad[(size_t)(k % S)]) != ORBFE_OK) {
        fprintf(stderr, "orbfe_pipeline_gather: %s\n", orbfe_last_error());
        gatherOk = false;
      }
      sh.gather_ms = std::max(sh.gather_ms, 0.0) + 1e3 * seconds_since(tg);
    }
    // S chunks stay in flight (round 5 waited for chunk k - 1 here: two in flight, the third slot idle -- 82 k frames/s where the same handle
    // driven with three in flight does 91 k with every block copied out, 105 k with the counts alone: profiles/r06_pipeline.md)
    finish_upto(k - (S - 1));
  }
  finish_upto(K - 1);
  sh.wall_s = seconds_since(tSequence);
  feeder.finish();
  sh.decode_s = feeder.decode_total_s;
  if (!gatherOk) sh.gatherBad++;
  for (int s = 0; s < S; s++)
    for (void* q : {an[(size_t)s], ak[(size_t)s], ad[(size_t)s]}) orbfe_device_free(q);
  orbfe_pipeline_destroy(pl);
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
    else if (k == "--batch") o.batch = std::max(0, atoi(v));
    else if (k == "--slots") o.slots = std::min(4, std::max(1, atoi(v)));
    else if (k == "--preload") o.preload = atoi(v);
    else if (k == "--repeat") o.repeat = std::max(1, atoi(v));
    else if (k == "--outputs") {   // what a chunk copies back to the host besides the per-frame counts
      const std::string w = v;
      if (w == "all") o.outputs = 0;
      else if (w == "matches") o.outputs = ORBFE_PIPE_OUT_ASSIGNED;
      else if (w == "counts") o.outputs = ORBFE_PIPE_OUT_COUNTS;
      else { fprintf(stderr, "--outputs all|matches|counts\n"); return 64; }
    }
    else { fprintf(stderr, "unknown option %s\n", k.c_str()); return 64; }
  }
  if (o.gather == "default") o.gather = o.gpus > 1 ? "root" : "none";
  if (o.outputs != 0 && !o.dumpPath.empty()) { fprintf(stderr, "--dump needs --outputs all\n"); return 64; }
  if (o.preload >= 2 && (!o.dumpPath.empty() || !o.gatherDump.empty())) { fprintf(stderr, "--preload 2 / 3 re-submit resident frames: no --dump / --gather-dump\n"); return 64; }
  if (o.gather != "none" && o.gather != "root" && o.gather != "all") { fprintf(stderr, "--gather none|root|all\n"); return 64; }
  if (o.gather != "none" && o.outputs != 0) { fprintf(stderr, "the gather's self-check compares with the host records: --outputs all\n"); return 64; }
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
  const int nReal = nImages;
  if (o.batch > 0) nImages *= o.repeat;               // --repeat walks the directory again: frame i reads image i % nReal
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
  // ---- the batched pipeline: one orbfe_pipeline per GPU thread
  std::vector<std::vector<uint8_t>> preL, preR;
  GatheredFrames gatheredFrames;
  std::vector<int> shardBegin((size_t)G + 1, 0);
  int w0 = 0, h0 = 0, capAll = 0;
  if (o.batch > 0 || gathering) {
    if (orbfe_png_info(vstrImageLeft[0].c_str(), &w0, &h0) != ORBFE_OK) { fprintf(stderr, "cannot read %s\n", vstrImageLeft[0].c_str()); return 65; }
    // the record capacity of a frame, once, for every rank: byte counts of a collective must agree, also on a rank that has no
    // frames or fails early
    orbfe_params prm = {o.nFeatures, 1.2f, 8, 20, 7};
    orbfe_extractor* probe = nullptr;
    if (orbfe_extractor_create(&prm, 0, &probe) != ORBFE_OK || orbfe_extractor_max_keypoints(probe, w0, h0, &capAll) != ORBFE_OK) {
      fprintf(stderr, "orbfe_extractor_create: %s\n", orbfe_last_error());
      return 3;
    }
    orbfe_extractor_destroy(probe);
  }
  HostBarrier beforeGather;
  beforeGather.n = G;
  if (o.batch > 0) {
    if (o.preload) {   // decode outside the clock: what is timed then is the pipeline itself (host copies and PCIe included)
      preL.resize((size_t)nReal); preR.resize((size_t)nReal);
      std::atomic<int> nx{0};
      std::atomic<bool> bad{false};
      std::vector<std::thread> ths;
      for (int t = 0; t < std::max(1, o.decodeThreads); t++)
        ths.emplace_back([&] {
          for (int i; (i = nx.fetch_add(1)) < nReal;) {
            int ww = 0, hh = 0;
            preL[(size_t)i].resize((size_t)w0 * h0); preR[(size_t)i].resize((size_t)w0 * h0);
            if (orbfe_png_read_gray(vstrImageLeft[(size_t)i].c_str(), preL[(size_t)i].data(), w0, h0, &ww, &hh) != ORBFE_OK || ww != w0 || hh != h0 ||
                orbfe_png_read_gray(vstrImageRight[(size_t)i].c_str(), preR[(size_t)i].data(), w0, h0, &ww, &hh) != ORBFE_OK || ww != w0 || hh != h0)
              bad.store(true);
          }
        });
      for (auto& t : ths) t.join();
      if (bad.load()) { fprintf(stderr, "\nFailed to load the sequence\n"); return 65; }
    }
    for (int g = 0; g <= G; g++) {
      int b = 0, e = 0;
      if (g < G) orbfe_shard_range(nImages, g, G, &b, &e);
      shardBegin[(size_t)g] = g < G ? b : nImages;
    }
    if (gathering && !o.gatherDump.empty()) {
      gatheredFrames.n.assign((size_t)nImages, -1);
      gatheredFrames.k.resize((size_t)nImages);
      gatheredFrames.d.resize((size_t)nImages);
    }
  }
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
    if (o.batch > 0) {
      GatherPlan gp;
      gp.comm = comms[(size_t)g]; gp.mode = o.gather == "all" ? ORBFE_GATHER_ALL : ORBFE_GATHER_ROOT; gp.G = G;
      gp.chunks = (chunkFrames + o.batch - 1) / o.batch; gp.cap = capAll;
      gp.receives = gathering && (o.gather == "all" || g == 0); gp.keep = !o.gatherDump.empty();
      RunShardBatched(oShard, vstrImageLeft, vstrImageRight, o.preload ? &preL : nullptr, o.preload ? &preR : nullptr, w0, h0, nReal, gp,
                      shardBegin, gp.keep ? &gatheredFrames : nullptr, sh);
      return;
    }
    sh.cap = capAll;
    RunShard(oShard, vstrImageLeft, vstrImageRight, gathering, chunkFrames, sh);
    if (!gathering) return;
    // one exchange at the end: this rank's records -> HBM -> RCCL -> (rank 0 | everyone) -> host.  A rank whose loop failed still
    // takes part, with the records it has (zeros behind them): the collective needs every rank, with equal byte counts.
    if (sh.rec_n.size() != (size_t)chunkFrames) { sh.rec_n.assign((size_t)chunkFrames, 0); sh.rec_k.assign((size_t)chunkFrames * sh.cap, orbfe_keypoint()); sh.rec_d.assign((size_t)chunkFrames * sh.cap * 32, 0); }
    const size_t bn = sizeof(int32_t) * (size_t)chunkFrames, bk = sizeof(orbfe_keypoint) * (size_t)chunkFrames * sh.cap,
                 bd = (size_t)32 * chunkFrames * sh.cap;
    const bool receives = o.gather == "all" || g == 0;
    void *dn = nullptr, *dk = nullptr, *dd = nullptr, *an = nullptr, *ak = nullptr, *ad = nullptr;
    int rc = orbfe_device_malloc(bn, &dn) | orbfe_device_malloc(bk, &dk) | orbfe_device_malloc(bd, &dd);
    if (receives) rc |= orbfe_device_malloc(bn * G, &an) | orbfe_device_malloc(bk * G, &ak) | orbfe_device_malloc(bd * G, &ad);
    rc |= orbfe_device_upload(dn, sh.rec_n.data(), bn) | orbfe_device_upload(dk, sh.rec_k.data(), bk) | orbfe_device_upload(dd, sh.rec_d.data(), bd);
    // a rank that could not get its buffers must not leave the others waiting inside the collective: all of them skip it
    bool skipped = false;
    if (beforeGather.arrive(rc != ORBFE_OK)) {
      skipped = rc == ORBFE_OK;
      rc = rc != ORBFE_OK ? rc : ORBFE_ERR_HIP;
    }
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
    if (rc != ORBFE_OK) { fprintf(stderr, "gather on GPU %d: %s\n", g, skipped ? "skipped, another GPU thread could not stage its records" : orbfe_last_error()); gatherOk = 0; }
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
  if (gathering && o.batch > 0) {
    // the pipelines compared every chunk's gathered rows with what they handed to the host (RunShardBatched)
    long bad = 0;
    double gms = 0;
    for (const Shard& sh : shards) { bad += sh.gatherBad; gms = std::max(gms, sh.gather_ms); }
    printf("gather (%s, %d GPU%s, RCCL through the C ABI): %d frames' records in chunks of %d, %.1f MB per GPU and chunk, straight from HBM, overlapped "
           "with the next chunk (%.3f ms of host time to enqueue), %ld frame(s) differ\n", o.gather.c_str(), G, G > 1 ? "s" : "", nImages, o.batch,
           (sizeof(int32_t) + (sizeof(orbfe_keypoint) + 32.0) * capAll) * o.batch / 1e6, gms, bad);
    if (bad) return 5;
    if (!o.gatherDump.empty()) {
      FILE* gd = fopen(o.gatherDump.c_str(), "wb");
      if (!gd) { fprintf(stderr, "cannot write %s\n", o.gatherDump.c_str()); return 73; }
      for (int f = 0; f < nImages; f++) {
        const int32_t n = gatheredFrames.n[(size_t)f];
        if (n < 0) { fprintf(stderr, "frame %d was not gathered\n", f); fclose(gd); return 5; }
        fwrite(&n, sizeof(n), 1, gd);
        fwrite(gatheredFrames.k[(size_t)f].data(), sizeof(orbfe_keypoint), (size_t)n, gd);
        fwrite(gatheredFrames.d[(size_t)f].data(), 32, (size_t)n, gd);
      }
      fclose(gd);
    }
  } else if (gathering) {
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
  if (o.batch > 0)
    printf("batched pipeline: chunks of %d pairs, %d buffer sets, %s; tracking times above are a chunk's submit-to-results time divided by its frames\n",
           o.batch, o.slots, o.preload == 3 ? "frames decoded before the clock started and resident in HBM (the slots' device input blocks) after the first chunks" :
           o.preload == 2 ? "frames decoded before the clock started and resident in the pinned slots after the first chunks" :
           o.preload ? "frames decoded before the clock started" : "PNGs decoded inside the clock");
  else
    printf("median per phase [ms]: ORBextractor x2 (two threads) %.4f, ComputeStereoMatches %.4f, SearchByProjection(cur,last) %.4f\n",
           1e3 * tPhase[0][tPhase[0].size() / 2], 1e3 * tPhase[1][tPhase[1].size() / 2], 1e3 * tPhase[2][tPhase[2].size() / 2]);
  return 0;
}
