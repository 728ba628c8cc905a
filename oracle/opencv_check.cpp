// opencv_check.cpp -- OPTIONAL harness that pins the oracle's OpenCV restatements against a real OpenCV.
//
// TEST INFRASTRUCTURE.  Built only where <opencv2/core.hpp> exists (`make -C oracle opencv_check`; the target prints a
// notice and does nothing otherwise -- this image and the GPU boxes have no OpenCV).  It issues the same
// cv:: calls, in the same sequence and with the same arguments, as the reference's extractor
// (Source/Libraries/ORB_SLAM2/src/ORBextractor.cc) and diffs every result against the oracle's primitive:
//   P1 cvRound                                   :79,108,112-113,437,455,1044-1045   vs oo_cvround / oo_cvroundf
//   P2 cv::resize(..., INTER_LINEAR), chained    :1054                               vs oo_resize_linear_u8
//   P3 cv::GaussianBlur(7x7, 2, 2, REFLECT_101)  :1019                               vs oo_gaussian_blur7_u8
//   P4 cv::FAST(cell, kps, th, true)             :774,778                            vs oo_fast9_16
//   P5 cv::fastAtan2                             :99                                 vs oo_fast_atan2
//   P6 cv::copyMakeBorder(REFLECT_101)           :1057,1061                          vs oo_copy_make_border_reflect101
// and, end to end, ComputePyramid + the per-cell FAST pass of ComputeKeyPointsOctTree (:733-791, 1041-1065) against
// oo_extract's level planes and candidate lists.  Exit code 0 = every comparison bit-equal; any mismatch is a parity bug
// in the ORACLE (fix it there first, then the HIP path follows through the GPU tests).
//
// Usage: opencv_check [image.pgm|png ...]   (no arguments: seeded synthetic images of the three dataset geometries)
#include <opencv2/core.hpp>
#include <opencv2/features2d.hpp>
#include <opencv2/imgcodecs.hpp>
#include <opencv2/imgproc.hpp>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "orb_oracle.h"

static int g_fail = 0;
#define CHECK(cond, ...)                      \
  do {                                        \
    if (!(cond)) {                            \
      if (g_fail < 50) { printf("MISMATCH: "); printf(__VA_ARGS__); printf("\n"); } \
      g_fail++;                               \
    }                                         \
  } while (0)

static uint32_t rng_state = 0xB5EEDu;
static uint32_t rnd() { rng_state = rng_state * 1664525u + 1013904223u; return rng_state >> 8; }

// noise octaves + rectangles + white noise: corners at every pyramid level (same recipe as synth.py, own RNG)
static cv::Mat synth_image(int w, int h, uint32_t seed) {
  rng_state = seed;
  cv::Mat acc(h, w, CV_32F, cv::Scalar(96));
  const int cellsz[3] = {64, 16, 4};
  const float amp[3] = {60, 40, 25};
  for (int o = 0; o < 3; o++) {
    cv::Mat small((h + cellsz[o] - 1) / cellsz[o] + 1, (w + cellsz[o] - 1) / cellsz[o] + 1, CV_32F);
    for (int y = 0; y < small.rows; y++)
      for (int x = 0; x < small.cols; x++) small.at<float>(y, x) = ((rnd() & 0xffff) / 65535.f - 0.5f) * amp[o];
    cv::Mat up;
    cv::resize(small, up, cv::Size(small.cols * cellsz[o], small.rows * cellsz[o]), 0, 0, cv::INTER_LINEAR);
    acc += up(cv::Rect(0, 0, w, h));
  }
  for (int r = 0; r < 300; r++) {
    const int x0 = rnd() % w, y0 = rnd() % h, rw = 4 + rnd() % 60, rh = 4 + rnd() % 60;
    cv::rectangle(acc, cv::Rect(x0, y0, rw, rh) & cv::Rect(0, 0, w, h), cv::Scalar((float)(rnd() % 256)), cv::FILLED);
  }
  cv::Mat out(h, w, CV_8UC1);
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      const float v = acc.at<float>(y, x) + (float)((int)(rnd() % 7) - 3);
      out.at<uchar>(y, x) = (uchar)std::min(255.f, std::max(0.f, v));
    }
  return out;
}

static void check_scalars() {
  // P1: halves and their neighbours, both signs; float and double entry points
  for (int i = -4000; i <= 4000; i++) {
    const double d = i * 0.25;
    CHECK(cvRound(d) == oo_cvround(d), "cvRound(%g) = %d, oracle %d", d, cvRound(d), oo_cvround(d));
    const float f = (float)d + 1e-3f * (i % 3 - 1);
    CHECK(cvRound(f) == oo_cvroundf(f), "cvRound(%gf) = %d, oracle %d", f, cvRound(f), oo_cvroundf(f));
  }
  // P5: a grid of moments as IC_Angle produces them (integers up to ~1e6), all quadrants and the axes
  for (int yi = -60; yi <= 60; yi++)
    for (int xi = -60; xi <= 60; xi++) {
      const float y = (float)(yi * 4099 + (yi & 1)), x = (float)(xi * 3571 - (xi & 3));
      const float a = cv::fastAtan2(y, x), b = oo_fast_atan2(y, x);
      CHECK(a == b, "fastAtan2(%g, %g) = %.9g, oracle %.9g", y, x, a, b);
    }
}

// Cases the whole-image runs cannot reach (each primitive called directly, small crafted inputs)
static void check_edge_cases() {
  // P2: one resize step at each of the seven KITTI level sizes, incl. the right-border columns where sx + 1 >= sw clamps
  // (xmax) and the bottom rows where the source row clamps
  const int lw[8] = {1241, 1034, 862, 718, 598, 499, 416, 346}, lh[8] = {376, 313, 261, 218, 181, 151, 126, 105};
  for (int l = 1; l < 8; l++) {
    cv::Mat src = synth_image(lw[l - 1], lh[l - 1], 0xC0FFEE + (uint32_t)l), dst;
    for (int y = 0; y < src.rows; y++) { src.at<uchar>(y, src.cols - 1) = (uchar)(y * 7); src.at<uchar>(y, src.cols - 2) = (uchar)(255 - y); }
    cv::resize(src, dst, cv::Size(lw[l], lh[l]), 0, 0, cv::INTER_LINEAR);
    std::vector<uint8_t> od((size_t)lw[l] * lh[l]);
    oo_resize_linear_u8(src.data, src.cols, src.rows, (int)src.step, od.data(), lw[l], lh[l], lw[l]);
    int bad = 0, bad_edge = 0;
    for (int y = 0; y < lh[l]; y++)
      for (int x = 0; x < lw[l]; x++) {
        const bool d = dst.at<uchar>(y, x) != od[(size_t)y * lw[l] + x];
        bad += d;
        bad_edge += d && (x >= lw[l] - 2 || y >= lh[l] - 2);
      }
    CHECK(bad == 0, "P2 resize %dx%d -> %dx%d: %d pixels differ (%d of them in the last two columns / rows)", lw[l - 1], lh[l - 1], lw[l],
          lh[l], bad, bad_edge);
  }
  // P3: GaussianBlur on degenerate widths (1 and 7 pixels wide, 1 and 7 pixels high): REFLECT_101 folds several taps onto one
  // sample; and on a saturated image (every sum at its maximum: the rounding constant shows)
  const int dims[][2] = {{1, 40}, {7, 40}, {40, 1}, {40, 7}, {2, 2}, {3, 5}, {64, 64}};
  for (auto& d : dims) {
    cv::Mat m(d[1], d[0], CV_8UC1), b;
    for (int y = 0; y < m.rows; y++)
      for (int x = 0; x < m.cols; x++) m.at<uchar>(y, x) = (d[0] == 64) ? 255 : (uchar)(rnd() & 0xff);
    cv::GaussianBlur(m, b, cv::Size(7, 7), 2, 2, cv::BORDER_REFLECT_101);
    std::vector<uint8_t> og((size_t)m.cols * m.rows);
    oo_gaussian_blur7_u8(m.data, m.cols, m.rows, (int)m.step, og.data(), m.cols);
    int bad = 0;
    for (int y = 0; y < m.rows; y++)
      for (int x = 0; x < m.cols; x++) bad += b.at<uchar>(y, x) != og[(size_t)y * m.cols + x];
    CHECK(bad == 0, "P3 GaussianBlur %dx%d: %d pixels differ", d[0], d[1], bad);
  }
  // P4: FAST on plateau corners -- equal scores on adjacent pixels (strict > in the non-maximum suppression: both go), a corner
  // whose arc is exactly nine long, thresholds at which the arc's weakest pixel is exactly v +- t (strict comparison), a 7 x 7
  // ROI (one tested pixel) and a ROI too small to test anything
  {
    cv::Mat m(40, 40, CV_8UC1, cv::Scalar(100));
    cv::rectangle(m, cv::Rect(10, 10, 12, 12), cv::Scalar(160), cv::FILLED);          // four corners with 2 x 2 plateaus nearby
    cv::rectangle(m, cv::Rect(28, 5, 2, 2), cv::Scalar(121), cv::FILLED);             // differences of exactly 21 and 20
    m.at<uchar>(30, 20) = 120; m.at<uchar>(30, 21) = 120;                             // twin maxima
    for (int th : {7, 19, 20, 21, 59, 60}) {
      for (auto roi : {cv::Rect(0, 0, 40, 40), cv::Rect(7, 7, 20, 20), cv::Rect(17, 17, 7, 7), cv::Rect(0, 0, 6, 6)}) {
        std::vector<cv::KeyPoint> k;
        cv::FAST(m(roi), k, th, true);
        std::vector<int> fx(roi.area() + 1), fy(roi.area() + 1), fs(roi.area() + 1);
        const int n = oo_fast9_16(m.data + (size_t)roi.y * m.step + roi.x, (int)m.step, roi.width, roi.height, th, 1, roi.area(), fx.data(), fy.data(), fs.data());
        CHECK(n == (int)k.size(), "P4 FAST plateau th %d roi %dx%d: %d keypoints, oracle %d", th, roi.width, roi.height, (int)k.size(), n);
        for (int t = 0; t < n && t < (int)k.size(); t++)
          CHECK((int)k[t].pt.x == fx[t] && (int)k[t].pt.y == fy[t] && (int)k[t].response == fs[t], "P4 FAST plateau th %d kp %d differs", th, t);
      }
    }
  }
  // P5: fastAtan2 on the axes, the diagonals, zero and tiny / huge magnitudes
  const float vals[] = {0.f, -0.f, 1.f, -1.f, 1e-30f, -1e-30f, 3e6f, -3e6f, 749.f * 255.f * 15.f, 5e-7f};
  for (float y : vals)
    for (float x : vals) {
      const float a = cv::fastAtan2(y, x), b = oo_fast_atan2(y, x);
      CHECK(a == b, "fastAtan2(%g, %g) = %.9g, oracle %.9g", y, x, a, b);
    }
}

static void check_image(const cv::Mat& image, int nfeatures, const char* name) {
  const int nlevels = 8;
  const float scaleFactor = 1.2f;
  const int iniThFAST = 20, minThFAST = 7, EDGE_THRESHOLD = 19;
  oo_extractor* e = oo_extractor_create(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST);
  const float* inv = oo_extractor_inv_scale_factors(e);
  std::vector<oo_keypoint> kps(nfeatures + 64);
  std::vector<uint8_t> desc((size_t)(nfeatures + 64) * 32);
  int n = 0;
  oo_extract(e, image.data, image.cols, image.rows, (int)image.step, kps.data(), desc.data(), nfeatures + 64, &n);

  // ---- ComputePyramid exactly as the reference writes it (:1041-1065)
  std::vector<cv::Mat> pyr(nlevels);
  for (int level = 0; level < nlevels; ++level) {
    const float scale = inv[level];
    cv::Size sz(cvRound((float)image.cols * scale), cvRound((float)image.rows * scale));
    cv::Size wholeSize(sz.width + EDGE_THRESHOLD * 2, sz.height + EDGE_THRESHOLD * 2);
    cv::Mat temp(wholeSize, image.type());
    pyr[level] = temp(cv::Rect(EDGE_THRESHOLD, EDGE_THRESHOLD, sz.width, sz.height));
    if (level != 0) {
      cv::resize(pyr[level - 1], pyr[level], sz, 0, 0, cv::INTER_LINEAR);
      cv::copyMakeBorder(pyr[level], temp, EDGE_THRESHOLD, EDGE_THRESHOLD, EDGE_THRESHOLD, EDGE_THRESHOLD,
                         cv::BORDER_REFLECT_101 + cv::BORDER_ISOLATED);
    } else {
      cv::copyMakeBorder(image, temp, EDGE_THRESHOLD, EDGE_THRESHOLD, EDGE_THRESHOLD, EDGE_THRESHOLD, cv::BORDER_REFLECT_101);
    }
    int ow = 0, oh = 0, ostride = 0;
    oo_level_size(e, level, &ow, &oh);
    CHECK(ow == sz.width && oh == sz.height, "%s level %d size %dx%d, oracle %dx%d", name, level, sz.width, sz.height, ow, oh);
    const uint8_t* op = oo_level_pixels(e, level, &ostride);
    int bad = 0;
    for (int y = 0; y < sz.height && ow == sz.width && oh == sz.height; y++)
      for (int x = 0; x < sz.width; x++) bad += pyr[level].at<uchar>(y, x) != op[(size_t)y * ostride + x];
    CHECK(bad == 0, "%s P2 resize level %d: %d pixels differ", name, level, bad);   // P2 (chained)
    // P6: the border frame
    std::vector<uint8_t> ob((size_t)wholeSize.width * wholeSize.height);
    oo_copy_make_border_reflect101(op, ow, oh, ostride, ob.data(), wholeSize.width, EDGE_THRESHOLD);
    bad = 0;
    for (int y = 0; y < wholeSize.height; y++)
      for (int x = 0; x < wholeSize.width; x++) bad += temp.at<uchar>(y, x) != ob[(size_t)y * wholeSize.width + x];
    CHECK(bad == 0, "%s P6 copyMakeBorder level %d: %d pixels differ", name, level, bad);
    // P3: the blur of :1017-1019 on a clone of the level
    cv::Mat workingMat = pyr[level].clone();
    cv::GaussianBlur(workingMat, workingMat, cv::Size(7, 7), 2, 2, cv::BORDER_REFLECT_101);
    std::vector<uint8_t> og((size_t)ow * oh);
    oo_gaussian_blur7_u8(op, ow, oh, ostride, og.data(), ow);
    bad = 0;
    for (int y = 0; y < oh; y++)
      for (int x = 0; x < ow; x++) bad += workingMat.at<uchar>(y, x) != og[(size_t)y * ow + x];
    CHECK(bad == 0, "%s P3 GaussianBlur level %d: %d pixels differ", name, level, bad);

    // ---- the per-cell FAST pass of ComputeKeyPointsOctTree (:740-791) vs the oracle's candidate list of the level
    const int minBorderX = EDGE_THRESHOLD - 3, minBorderY = minBorderX;
    const int maxBorderX = pyr[level].cols - EDGE_THRESHOLD + 3, maxBorderY = pyr[level].rows - EDGE_THRESHOLD + 3;
    const float width = (float)(maxBorderX - minBorderX), height = (float)(maxBorderY - minBorderY);
    const int nCols = (int)(width / 30), nRows = (int)(height / 30);
    std::vector<int> cx, cy, cs;
    if (nCols >= 1 && nRows >= 1) {
      const int wCell = (int)std::ceil(width / nCols), hCell = (int)std::ceil(height / nRows);
      for (int i = 0; i < nRows; i++) {
        const float iniY = (float)(minBorderY + i * hCell);
        float maxY = iniY + hCell + 6;
        if (iniY >= maxBorderY - 3) continue;
        if (maxY > maxBorderY) maxY = (float)maxBorderY;
        for (int j = 0; j < nCols; j++) {
          const float iniX = (float)(minBorderX + j * wCell);
          float maxX = iniX + wCell + 6;
          if (iniX >= maxBorderX - 6) continue;
          if (maxX > maxBorderX) maxX = (float)maxBorderX;
          std::vector<cv::KeyPoint> vKeysCell;
          cv::FAST(pyr[level].rowRange((int)iniY, (int)maxY).colRange((int)iniX, (int)maxX), vKeysCell, iniThFAST, true);
          if (vKeysCell.empty())
            cv::FAST(pyr[level].rowRange((int)iniY, (int)maxY).colRange((int)iniX, (int)maxX), vKeysCell, minThFAST, true);
          // P4 on the same ROI, both thresholds
          {
            const int cols = (int)maxX - (int)iniX, rows = (int)maxY - (int)iniY;
            std::vector<int> fx(cols * rows + 1), fy(cols * rows + 1), fs(cols * rows + 1);
            int m = oo_fast9_16(op + (size_t)(int)iniY * ostride + (int)iniX, ostride, cols, rows, iniThFAST, 1, cols * rows, fx.data(),
                                fy.data(), fs.data());
            if (m == 0)
              m = oo_fast9_16(op + (size_t)(int)iniY * ostride + (int)iniX, ostride, cols, rows, minThFAST, 1, cols * rows, fx.data(), fy.data(),
                              fs.data());
            CHECK(m == (int)vKeysCell.size(), "%s P4 FAST level %d cell (%d,%d): %d keypoints, oracle %d", name, level, i, j,
                  (int)vKeysCell.size(), m);
            for (int t = 0; t < m && t < (int)vKeysCell.size(); t++)
              CHECK((int)vKeysCell[t].pt.x == fx[t] && (int)vKeysCell[t].pt.y == fy[t] && (int)vKeysCell[t].response == fs[t],
                    "%s P4 FAST level %d cell (%d,%d) kp %d: (%g,%g,%g) vs oracle (%d,%d,%d)", name, level, i, j, t, vKeysCell[t].pt.x,
                    vKeysCell[t].pt.y, vKeysCell[t].response, fx[t], fy[t], fs[t]);
          }
          for (auto& kp : vKeysCell) {
            cx.push_back((int)kp.pt.x + j * wCell);
            cy.push_back((int)kp.pt.y + i * hCell);
            cs.push_back((int)kp.response);
          }
        }
      }
    }
    const int *ox, *oy, *os_;
    const int on = oo_level_candidates(e, level, &ox, &oy, &os_);
    CHECK(on == (int)cx.size(), "%s level %d: %d candidates, oracle %d", name, level, (int)cx.size(), on);
    for (int t = 0; t < on && t < (int)cx.size(); t++)
      CHECK(cx[t] == ox[t] && cy[t] == oy[t] && cs[t] == os_[t], "%s level %d candidate %d differs", name, level, t);
  }
  printf("%-24s %4d keypoints (oracle), levels / blur / border / FAST cells compared%s\n", name, n, g_fail ? " -- MISMATCHES" : "");
  oo_extractor_destroy(e);
}

int main(int argc, char** argv) {
  check_scalars();
  check_edge_cases();
  if (argc > 1) {
    for (int i = 1; i < argc; i++) {
      cv::Mat im = cv::imread(argv[i], cv::IMREAD_GRAYSCALE);
      if (im.empty()) { printf("cannot read %s\n", argv[i]); return 2; }
      check_image(im, 1000, argv[i]);
    }
  } else {
    check_image(synth_image(1241, 376, 0xB5EED + 1), 2000, "kitti 1241x376/2000");
    check_image(synth_image(640, 480, 0xB5EED + 2), 1000, "tum 640x480/1000");
    check_image(synth_image(752, 480, 0xB5EED + 3), 1200, "euroc 752x480/1200");
  }
  if (g_fail) printf("opencv_check: %d MISMATCHES vs OpenCV %s\n", g_fail, CV_VERSION);
  else printf("opencv_check: oracle primitives and level pipeline bit-equal to OpenCV %s\n", CV_VERSION);
  return g_fail ? 1 : 0;
}
