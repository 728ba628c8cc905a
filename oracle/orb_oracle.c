/*
 * orb_oracle.c -- CPU restatement ("oracle") of the ORB-SLAM2 per-frame front end.
 *
 * TEST INFRASTRUCTURE ONLY (see orb_oracle.h).  PARITY UNPINNED: no reference golden vectors exist
 * and the reference cannot be built here (OpenCV absent); OpenCV primitives are restated from the
 * OpenCV 4.5.x generic algorithms.  Scalar, single-threaded, written for clarity, not speed.
 *
 * L/ = Source/Libraries/ORB_SLAM2/ in the reference checkout.
 * Build: see oracle/Makefile (-O3 -ffp-contract=off: the reference x86-64 build has no FMA contraction).
 */
#include "orb_oracle.h"

#include <float.h>
#include <limits.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/orb_pattern_data.h"

#define PATCH_SIZE 31      /* L/src/ORBextractor.cc:72 */
#define HALF_PATCH_SIZE 15 /* :73 */
#define EDGE_THRESHOLD 19  /* :74 */

static const int8_t g_pattern[1024] = {ORB_PATTERN_INT8_1024};
const int8_t* oo_pattern(void) { return g_pattern; }

static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

/* ------------------------------------------------------------------------------------------------ P1 */
/* cvRound: SSE2 cvtsd2si / cvtss2si under the default rounding mode = round half to even. */
int oo_cvround(double v) { return (int)lrint(v); }
int oo_cvroundf(float v) { return (int)lrintf(v); }
static inline int cv_floor_f(float v) { return (int)floorf(v); }
static inline int16_t sat_short_from_float(float v) {
  int i = oo_cvroundf(v);
  return (int16_t)(i < -32768 ? -32768 : (i > 32767 ? 32767 : i));
}

/* ------------------------------------------------------------------------------------------------ P5 */
/* cv::fastAtan2 scalar path (OpenCV core/src/mathfuncs_core: atan_f32), degrees, un-fused float ops.
 * Called by IC_Angle, L/src/ORBextractor.cc:99. */
float oo_fast_atan2(float y, float x) {
  const float scale = (float)(180.0 / 3.1415926535897932384626433832795);
  const float p1 = 0.9997878412794807f * scale;
  const float p3 = -0.3258083974640975f * scale;
  const float p5 = 0.1555786518463281f * scale;
  const float p7 = -0.04432655554792128f * scale;
  float ax = fabsf(x), ay = fabsf(y);
  float a, c, c2;
  if (ax >= ay) {
    c = ay / (ax + (float)DBL_EPSILON);
    c2 = c * c;
    a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  } else {
    c = ax / (ay + (float)DBL_EPSILON);
    c2 = c * c;
    a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  }
  if (x < 0) a = 180.f - a;
  if (y < 0) a = 360.f - a;
  return a;
}

/* ---------------------------------------------------------------------------------- glibc sinf / cosf */
/* The reference calls cosf/sinf through <cmath> overloads (L/src/ORBextractor.cc:106).  On the boxes this
 * runs on that is glibc 2.35's flt-32 sinf/cosf (ARM optimized-routines design: reduce in double, 2
 * polynomials, round once).  Restated here so the device code can follow the same double arithmetic;
 * tests compare it with libm over the whole [0, 2*pi] float range. */
static const double SC_HPI_INV = 0x1.45F306DC9C883p+23; /* 2/pi * 2^24 */
static const double SC_HPI = 0x1.921FB54442D18p0;       /* pi/2 */
static const double SC_C0 = 0x1p0, SC_C1 = -0x1.ffffffd0c621cp-2, SC_C2 = 0x1.55553e1068f19p-5,
                    SC_C3 = -0x1.6c087e89a359dp-10, SC_C4 = 0x1.99343027bf8c3p-16;
static const double SC_S1 = -0x1.555545995a603p-3, SC_S2 = 0x1.1107605230bc4p-7,
                    SC_S3 = -0x1.994eb3774cf24p-13;

/* evaluates sin (n even) or cos (n odd) polynomial; flip negates the cos polynomial (table entry 1) */
static inline float sc_poly(double x, double x2, int n, int flip) {
  if ((n & 1) == 0) {
    double x3 = x * x2;
    double s1 = SC_S2 + x2 * SC_S3;
    double x7 = x3 * x2;
    double s = x + x3 * SC_S1;
    return (float)(s + x7 * s1);
  } else {
    double sg = flip ? -1.0 : 1.0;
    double x4 = x2 * x2;
    double c2 = sg * SC_C3 + x2 * (sg * SC_C4);
    double c1 = sg * SC_C1 + x2 * (sg * SC_C2);
    double x6 = x4 * x2;
    double c = sg * SC_C0 + x2 * c1;
    return (float)(c + x6 * c2);
  }
}
static inline uint32_t abstop12(float x) {
  uint32_t u;
  memcpy(&u, &x, 4);
  return (u >> 20) & 0x7ff;
}
static const double sc_sign[4] = {1.0, -1.0, -1.0, 1.0};

float oo_sinf(float y) {
  double x = y;
  if (abstop12(y) < abstop12(0x1.921FB6p-1f)) { /* |y| < pi/4 */
    double s = x * x;
    if (abstop12(y) < abstop12(0x1p-12f)) return y;
    return sc_poly(x, s, 0, 0);
  }
  if (abstop12(y) < abstop12(120.0f)) {
    double r = x * SC_HPI_INV;
    int n = ((int32_t)r + 0x800000) >> 24;
    x = x - n * SC_HPI;
    double s = sc_sign[n & 3];
    return sc_poly(x * s, x * x, n, (n & 2) != 0);
  }
  return sinf(y); /* outside the range the front end can produce */
}
float oo_cosf(float y) {
  double x = y;
  if (abstop12(y) < abstop12(0x1.921FB6p-1f)) {
    double x2 = x * x;
    if (abstop12(y) < abstop12(0x1p-12f)) return 1.0f;
    return sc_poly(x, x2, 1, 0);
  }
  if (abstop12(y) < abstop12(120.0f)) {
    double r = x * SC_HPI_INV;
    int n = ((int32_t)r + 0x800000) >> 24;
    x = x - n * SC_HPI;
    double s = sc_sign[n & 3];
    return sc_poly(x * s, x * x, n ^ 1, (n & 2) != 0);
  }
  return cosf(y);
}

/* ------------------------------------------------------------------------------------------------ P2 */
/* cv::resize(INTER_LINEAR) for CV_8UC1, generic fixed-point path (imgproc/src/resize.cpp: coefficient
 * set-up in cv::resize, HResizeLinear + VResizeLinear<uchar,int,short,FixedPtCast<..,22>>).
 * Called by ComputePyramid, L/src/ORBextractor.cc:1054, chained level to level. */
void oo_resize_tables(int s, int d, int32_t* ofs, int16_t* coef) {
  double inv_scale = (double)d / s;
  double scale = 1.0 / inv_scale;
  for (int i = 0; i < d; i++) {
    float f = (float)((i + 0.5) * scale - 0.5);
    int si = cv_floor_f(f);
    f -= si;
    if (si < 0) { f = 0; si = 0; }
    if (si >= s - 1) { f = 0; si = s - 1; }
    ofs[i] = si;
    coef[2 * i] = sat_short_from_float((1.f - f) * 2048);
    coef[2 * i + 1] = sat_short_from_float(f * 2048);
  }
}

void oo_resize_linear_u8(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw, int dh,
                         int dstride) {
  int32_t* xofs = (int32_t*)malloc(sizeof(int32_t) * dw);
  int16_t* ialpha = (int16_t*)malloc(sizeof(int16_t) * 2 * dw);
  int32_t* yofs = (int32_t*)malloc(sizeof(int32_t) * dh);
  int16_t* ibeta = (int16_t*)malloc(sizeof(int16_t) * 2 * dh);
  int32_t* row0 = (int32_t*)malloc(sizeof(int32_t) * dw);
  int32_t* row1 = (int32_t*)malloc(sizeof(int32_t) * dw);
  oo_resize_tables(sw, dw, xofs, ialpha);
  /* rows: the y set-up of cv::resize does not force fy=0 at the borders; rows are clamped when read */
  {
    double scale_y = 1.0 / ((double)dh / sh);
    for (int dy = 0; dy < dh; dy++) {
      float fy = (float)((dy + 0.5) * scale_y - 0.5);
      int sy = cv_floor_f(fy);
      fy -= sy;
      yofs[dy] = sy;
      ibeta[2 * dy] = sat_short_from_float((1.f - fy) * 2048);
      ibeta[2 * dy + 1] = sat_short_from_float(fy * 2048);
    }
  }
  for (int dy = 0; dy < dh; dy++) {
    int sy0 = yofs[dy], sy1 = yofs[dy] + 1;
    sy0 = sy0 < 0 ? 0 : (sy0 >= sh ? sh - 1 : sy0);
    sy1 = sy1 < 0 ? 0 : (sy1 >= sh ? sh - 1 : sy1);
    const uint8_t* S0 = src + (size_t)sy0 * sstride;
    const uint8_t* S1 = src + (size_t)sy1 * sstride;
    for (int dx = 0; dx < dw; dx++) {
      int sx = xofs[dx];
      int sx1 = sx + 1 < sw ? sx + 1 : sx; /* coefficient is 0 there */
      row0[dx] = S0[sx] * ialpha[2 * dx] + S0[sx1] * ialpha[2 * dx + 1];
      row1[dx] = S1[sx] * ialpha[2 * dx] + S1[sx1] * ialpha[2 * dx + 1];
    }
    int b0 = ibeta[2 * dy], b1 = ibeta[2 * dy + 1];
    uint8_t* D = dst + (size_t)dy * dstride;
    for (int dx = 0; dx < dw; dx++)
      D[dx] = (uint8_t)((((b0 * (row0[dx] >> 4)) >> 16) + ((b1 * (row1[dx] >> 4)) >> 16) + 2) >> 2);
  }
  free(xofs); free(ialpha); free(yofs); free(ibeta); free(row0); free(row1);
}

/* ------------------------------------------------------------------------------------------------ P6 */
static inline int reflect101(int p, int len) {
  if (len == 1) return 0;
  while (p < 0 || p >= len) {
    if (p < 0) p = -p;
    else p = 2 * (len - 1) - p;
  }
  return p;
}
void oo_copy_make_border_reflect101(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride,
                                    int border) {
  for (int y = -border; y < h + border; y++) {
    const uint8_t* S = src + (size_t)reflect101(y, h) * sstride;
    uint8_t* D = dst + (size_t)(y + border) * dstride;
    for (int x = -border; x < w + border; x++) D[x + border] = S[reflect101(x, w)];
  }
}

/* ------------------------------------------------------------------------------------------------ P3 */
/* cv::GaussianBlur(7x7, sigma 2) on a non-submatrix CV_8U image: fixed-point (8.8) separable kernel
 * with outside-in error diffusion, exact 16.16 accumulation, round half up.  L/src/ORBextractor.cc:1019. */
void oo_gauss_taps7(int taps[7]) {
  double k[7], sum = 0;
  for (int i = 0; i < 7; i++) {
    double x = i - 3;
    k[i] = exp(-(x * x) / (2.0 * 2.0 * 2.0));
    sum += k[i];
  }
  double err = 0;
  int outer = 0;
  for (int i = 0; i < 3; i++) {
    double v = k[i] / sum * 256.0 + err;
    int q = (int)floor(v + 0.5);
    err = v - q;
    taps[i] = taps[6 - i] = q;
    outer += 2 * q;
  }
  taps[3] = 256 - outer;
}

void oo_gaussian_blur7_u8(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride) {
  int taps[7];
  oo_gauss_taps7(taps);
  uint16_t* tmp = (uint16_t*)malloc(sizeof(uint16_t) * (size_t)w * h);
  for (int y = 0; y < h; y++) {
    const uint8_t* S = src + (size_t)y * sstride;
    for (int x = 0; x < w; x++) {
      unsigned acc = 0;
      for (int k = -3; k <= 3; k++) acc += (unsigned)taps[k + 3] * S[reflect101(x + k, w)];
      tmp[(size_t)y * w + x] = (uint16_t)acc; /* <= 255*256, no saturation */
    }
  }
  for (int y = 0; y < h; y++) {
    uint8_t* D = dst + (size_t)y * dstride;
    for (int x = 0; x < w; x++) {
      uint32_t acc = 0;
      for (int k = -3; k <= 3; k++) acc += (uint32_t)taps[k + 3] * tmp[(size_t)reflect101(y + k, h) * w + x];
      D[x] = (uint8_t)((acc + 32768u) >> 16);
    }
  }
  free(tmp);
}

/* ------------------------------------------------------------------------------------------------ P4 */
/* cv::FAST(img, kps, threshold, nonmax) = FAST_t<16> (features2d/src/fast.cpp) + cornerScore<16>
 * (fast_score.cpp).  Called per cell by ComputeKeyPointsOctTree, L/src/ORBextractor.cc:774-779. */
static const int RING_DX[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
static const int RING_DY[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};

int oo_fast_corner_score(const uint8_t* p, int stride, int threshold) {
  int d[25];
  int v = p[0];
  for (int k = 0; k < 25; k++) d[k] = v - p[RING_DX[k & 15] + RING_DY[k & 15] * stride];
  int a0 = threshold;
  for (int k = 0; k < 16; k += 2) {
    int a = imin(d[k + 1], d[k + 2]);
    a = imin(a, d[k + 3]);
    if (a <= a0) continue;
    a = imin(a, d[k + 4]); a = imin(a, d[k + 5]); a = imin(a, d[k + 6]);
    a = imin(a, d[k + 7]); a = imin(a, d[k + 8]);
    a0 = imax(a0, imin(a, d[k]));
    a0 = imax(a0, imin(a, d[k + 9]));
  }
  int b0 = -a0;
  for (int k = 0; k < 16; k += 2) {
    int b = imax(d[k + 1], d[k + 2]);
    b = imax(b, d[k + 3]); b = imax(b, d[k + 4]); b = imax(b, d[k + 5]);
    if (b >= b0) continue;
    b = imax(b, d[k + 6]); b = imax(b, d[k + 7]); b = imax(b, d[k + 8]);
    b0 = imin(b0, imax(b, d[k]));
    b0 = imin(b0, imax(b, d[k + 9]));
  }
  return -b0 - 1;
}

static int fast_is_corner(const uint8_t* p, int stride, int threshold) {
  int v = p[0];
  int lo = v - threshold, hi = v + threshold;
  int cd = 0, cb = 0; /* running contiguous counts over the 25-long unrolled ring */
  for (int k = 0; k < 25; k++) {
    int x = p[RING_DX[k & 15] + RING_DY[k & 15] * stride];
    if (x < lo) { if (++cd > 8) return 1; } else cd = 0;
    if (x > hi) { if (++cb > 8) return 1; } else cb = 0;
  }
  return 0;
}

int oo_fast9_16(const uint8_t* img, int stride, int cols, int rows, int threshold, int nonmax, int cap,
                int* out_x, int* out_y, int* out_score) {
  threshold = imin(imax(threshold, 0), 255);
  if (cols < 7 || rows < 7) return 0;
  uint8_t* score = (uint8_t*)calloc((size_t)cols * rows, 1);
  uint8_t* corner = (uint8_t*)calloc((size_t)cols * rows, 1);
  for (int i = 3; i < rows - 3; i++)
    for (int j = 3; j < cols - 3; j++) {
      const uint8_t* p = img + (size_t)i * stride + j;
      if (fast_is_corner(p, stride, threshold)) {
        corner[(size_t)i * cols + j] = 1;
        if (nonmax) score[(size_t)i * cols + j] = (uint8_t)oo_fast_corner_score(p, stride, threshold);
      }
    }
  int n = 0;
  for (int i = 3; i < rows - 3; i++)
    for (int j = 3; j < cols - 3; j++) {
      if (!corner[(size_t)i * cols + j]) continue;
      int s = score[(size_t)i * cols + j];
      int keep = 1;
      if (nonmax) {
        const uint8_t* c = score + (size_t)i * cols + j;
        keep = s > c[-1] && s > c[1] && s > c[-cols - 1] && s > c[-cols] && s > c[-cols + 1] &&
               s > c[cols - 1] && s > c[cols] && s > c[cols + 1];
      }
      if (keep) {
        if (n < cap) { out_x[n] = j; out_y[n] = i; out_score[n] = s; }
        n++;
      }
    }
  free(score); free(corner);
  return n;
}

/* --------------------------------------------------------------------------------------- extractor */
typedef struct {
  int w, h, stride;
  uint8_t* pix;     /* mvImagePyramid[level] (ROI pixels, no border) */
  uint8_t* blurred; /* workingMat after GaussianBlur */
  int ncand, cand_cap;
  int *cx, *cy, *cs;
  int nkp;
  oo_keypoint* kps;
} oo_level;

struct oo_extractor {
  int nfeatures;
  double scaleFactor; /* L/include/ORBextractor.h:91: stored as double */
  int nlevels, iniThFAST, minThFAST;
  float mvScaleFactor[OO_MAX_LEVELS], mvInvScaleFactor[OO_MAX_LEVELS];
  float mvLevelSigma2[OO_MAX_LEVELS], mvInvLevelSigma2[OO_MAX_LEVELS];
  int mnFeaturesPerLevel[OO_MAX_LEVELS];
  int umax[HALF_PATCH_SIZE + 1];
  oo_level lv[OO_MAX_LEVELS];
};

/* L/src/ORBextractor.cc:407-464 */
oo_extractor* oo_extractor_create(int nfeatures, float scale_factor, int nlevels, int ini_th, int min_th) {
  if (nlevels < 1 || nlevels > OO_MAX_LEVELS) return NULL;
  oo_extractor* e = (oo_extractor*)calloc(1, sizeof(*e));
  e->nfeatures = nfeatures;
  e->scaleFactor = scale_factor;
  e->nlevels = nlevels;
  e->iniThFAST = ini_th;
  e->minThFAST = min_th;
  e->mvScaleFactor[0] = 1.0f;
  e->mvLevelSigma2[0] = 1.0f;
  for (int i = 1; i < nlevels; i++) {
    e->mvScaleFactor[i] = (float)(e->mvScaleFactor[i - 1] * e->scaleFactor);
    e->mvLevelSigma2[i] = e->mvScaleFactor[i] * e->mvScaleFactor[i];
  }
  for (int i = 0; i < nlevels; i++) {
    e->mvInvScaleFactor[i] = 1.0f / e->mvScaleFactor[i];
    e->mvInvLevelSigma2[i] = 1.0f / e->mvLevelSigma2[i];
  }
  float factor = (float)(1.0f / e->scaleFactor);
  float nDesired = nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nlevels));
  int sum = 0;
  for (int level = 0; level < nlevels - 1; level++) {
    e->mnFeaturesPerLevel[level] = oo_cvroundf(nDesired);
    sum += e->mnFeaturesPerLevel[level];
    nDesired *= factor;
  }
  e->mnFeaturesPerLevel[nlevels - 1] = imax(nfeatures - sum, 0);

  int v, v0;
  int vmax = (int)floorf(HALF_PATCH_SIZE * sqrtf(2.f) / 2 + 1);
  int vmin = (int)ceilf(HALF_PATCH_SIZE * sqrtf(2.f) / 2);
  const double hp2 = HALF_PATCH_SIZE * HALF_PATCH_SIZE;
  for (v = 0; v <= vmax; ++v) e->umax[v] = oo_cvround(sqrt(hp2 - v * v));
  for (v = HALF_PATCH_SIZE, v0 = 0; v >= vmin; --v) {
    while (e->umax[v0] == e->umax[v0 + 1]) ++v0;
    e->umax[v] = v0;
    ++v0;
  }
  return e;
}

static void level_free(oo_level* l) {
  free(l->pix); free(l->blurred); free(l->cx); free(l->cy); free(l->cs); free(l->kps);
  memset(l, 0, sizeof(*l));
}
void oo_extractor_destroy(oo_extractor* e) {
  if (!e) return;
  for (int i = 0; i < OO_MAX_LEVELS; i++) level_free(&e->lv[i]);
  free(e);
}
int oo_extractor_levels(const oo_extractor* e) { return e->nlevels; }
const float* oo_extractor_scale_factors(const oo_extractor* e) { return e->mvScaleFactor; }
const float* oo_extractor_inv_scale_factors(const oo_extractor* e) { return e->mvInvScaleFactor; }
const float* oo_extractor_sigma2(const oo_extractor* e) { return e->mvLevelSigma2; }
const float* oo_extractor_inv_sigma2(const oo_extractor* e) { return e->mvInvLevelSigma2; }
const int* oo_extractor_features_per_level(const oo_extractor* e) { return e->mnFeaturesPerLevel; }
const int* oo_extractor_umax(const oo_extractor* e) { return e->umax; }

/* IC_Angle: L/src/ORBextractor.cc:76-100 */
float oo_ic_angle(const uint8_t* img, int stride, int x, int y, const int* umax) {
  int m_01 = 0, m_10 = 0;
  const uint8_t* center = img + (size_t)y * stride + x;
  for (int u = -HALF_PATCH_SIZE; u <= HALF_PATCH_SIZE; ++u) m_10 += u * center[u];
  for (int v = 1; v <= HALF_PATCH_SIZE; ++v) {
    int v_sum = 0;
    int d = umax[v];
    for (int u = -d; u <= d; ++u) {
      int val_plus = center[u + v * stride], val_minus = center[u - v * stride];
      v_sum += (val_plus - val_minus);
      m_10 += u * (val_plus + val_minus);
    }
    m_01 += v * v_sum;
  }
  return oo_fast_atan2((float)m_01, (float)m_10);
}

/* computeOrbDescriptor: L/src/ORBextractor.cc:102-146 */
void oo_orb_descriptor(const uint8_t* img, int stride, int x, int y, float angle_deg, uint8_t desc[32]) {
  const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
  float angle = angle_deg * factorPI;
  float a = cosf(angle), b = sinf(angle); /* float overloads, :106 */
  const uint8_t* center = img + (size_t)y * stride + x;
  const int8_t* pat = g_pattern;
  for (int i = 0; i < 32; ++i, pat += 32) {
    int val = 0;
    for (int j = 0; j < 8; j++) {
      int x0 = pat[4 * j], y0 = pat[4 * j + 1], x1 = pat[4 * j + 2], y1 = pat[4 * j + 3];
      int t0 = center[oo_cvroundf(x0 * b + y0 * a) * stride + oo_cvroundf(x0 * a - y0 * b)];
      int t1 = center[oo_cvroundf(x1 * b + y1 * a) * stride + oo_cvroundf(x1 * a - y1 * b)];
      val |= (t0 < t1) << j;
    }
    desc[i] = (uint8_t)val;
  }
}

/* ---- DistributeOctTree + DivideNode: L/src/ORBextractor.cc:475-731 -------------------------------- */
typedef struct {
  int ULx, ULy, URx, URy, BLx, BLy, BRx, BRy;
  int* keys; /* indices into the candidate arrays, in vKeys order */
  int nkeys;
  int bNoMore;
  int prev, next; /* std::list links (pool indices, -1 = none) */
  int alive;
} oct_node;

typedef struct {
  oct_node* pool; /* pool index == creation order == "address" under a bump allocator */
  int npool, cap;
  int head, tail, size;
} oct_list;

static int oct_alloc(oct_list* L) {
  if (L->npool == L->cap) {
    L->cap = L->cap ? 2 * L->cap : 256;
    L->pool = (oct_node*)realloc(L->pool, sizeof(oct_node) * L->cap);
  }
  memset(&L->pool[L->npool], 0, sizeof(oct_node));
  L->pool[L->npool].prev = L->pool[L->npool].next = -1;
  return L->npool++;
}
static void oct_push_back(oct_list* L, int id) {
  oct_node* n = &L->pool[id];
  n->alive = 1; n->prev = L->tail; n->next = -1;
  if (L->tail >= 0) L->pool[L->tail].next = id; else L->head = id;
  L->tail = id; L->size++;
}
static void oct_push_front(oct_list* L, int id) {
  oct_node* n = &L->pool[id];
  n->alive = 1; n->next = L->head; n->prev = -1;
  if (L->head >= 0) L->pool[L->head].prev = id; else L->tail = id;
  L->head = id; L->size++;
}
static int oct_erase(oct_list* L, int id) { /* returns next */
  oct_node* n = &L->pool[id];
  int nx = n->next;
  if (n->prev >= 0) L->pool[n->prev].next = n->next; else L->head = n->next;
  if (n->next >= 0) L->pool[n->next].prev = n->prev; else L->tail = n->prev;
  n->alive = 0; L->size--;
  free(n->keys); n->keys = NULL;
  return nx;
}

/* DivideNode (:475-527): fills 4 freshly allocated pool nodes c[0..3] (not yet linked) */
static void oct_divide(oct_list* L, int id, const int* kx, const int* ky, int c[4]) {
  for (int k = 0; k < 4; k++) c[k] = oct_alloc(L);
  oct_node* p = &L->pool[id];
  const int halfX = (int)ceilf((float)(p->URx - p->ULx) / 2);
  const int halfY = (int)ceilf((float)(p->BRy - p->ULy) / 2);
  oct_node *n1 = &L->pool[c[0]], *n2 = &L->pool[c[1]], *n3 = &L->pool[c[2]], *n4 = &L->pool[c[3]];
  n1->ULx = p->ULx; n1->ULy = p->ULy;
  n1->URx = p->ULx + halfX; n1->URy = p->ULy;
  n1->BLx = p->ULx; n1->BLy = p->ULy + halfY;
  n1->BRx = p->ULx + halfX; n1->BRy = p->ULy + halfY;
  n2->ULx = n1->URx; n2->ULy = n1->URy;
  n2->URx = p->URx; n2->URy = p->URy;
  n2->BLx = n1->BRx; n2->BLy = n1->BRy;
  n2->BRx = p->URx; n2->BRy = p->ULy + halfY;
  n3->ULx = n1->BLx; n3->ULy = n1->BLy;
  n3->URx = n1->BRx; n3->URy = n1->BRy;
  n3->BLx = p->BLx; n3->BLy = p->BLy;
  n3->BRx = n1->BRx; n3->BRy = p->BLy;
  n4->ULx = n3->URx; n4->ULy = n3->URy;
  n4->URx = n2->BRx; n4->URy = n2->BRy;
  n4->BLx = n3->BRx; n4->BLy = n3->BRy;
  n4->BRx = p->BRx; n4->BRy = p->BRy;
  for (int k = 0; k < 4; k++) L->pool[c[k]].keys = (int*)malloc(sizeof(int) * (p->nkeys ? p->nkeys : 1));
  for (int i = 0; i < p->nkeys; i++) {
    int key = p->keys[i];
    float px = (float)kx[key], py = (float)ky[key];
    oct_node* t;
    if (px < n1->URx) t = (py < n1->BRy) ? n1 : n3;
    else t = (py < n1->BRy) ? n2 : n4;
    t->keys[t->nkeys++] = key;
  }
  for (int k = 0; k < 4; k++)
    if (L->pool[c[k]].nkeys == 1) L->pool[c[k]].bNoMore = 1;
}

typedef struct { int size, id; } size_ptr;
static int size_ptr_cmp(const void* a, const void* b) { /* std::pair<int, ExtractorNode*> operator< */
  const size_ptr *x = (const size_ptr*)a, *y = (const size_ptr*)b;
  if (x->size != y->size) return x->size < y->size ? -1 : 1;
  return x->id < y->id ? -1 : (x->id > y->id ? 1 : 0);
}

int oo_distribute_octree(const int* x, const int* y, const int* score, int n, int minX, int maxX, int minY,
                         int maxY, int N, int* out_idx) {
  const int nIni = (int)roundf((float)(maxX - minX) / (maxY - minY)); /* :535 */
  if (nIni < 1) return 0; /* reference would index an empty vector (:559); guarded */
  const float hX = (float)(maxX - minX) / nIni;
  oct_list L; memset(&L, 0, sizeof(L)); L.head = L.tail = -1;
  int* ini = (int*)malloc(sizeof(int) * nIni);
  for (int i = 0; i < nIni; i++) {
    int id = oct_alloc(&L);
    oct_node* ni = &L.pool[id];
    ni->ULx = (int)(hX * (float)i); ni->ULy = 0;
    ni->URx = (int)(hX * (float)(i + 1)); ni->URy = 0;
    ni->BLx = ni->ULx; ni->BLy = maxY - minY;
    ni->BRx = ni->URx; ni->BRy = maxY - minY;
    ni->keys = (int*)malloc(sizeof(int) * (n ? n : 1));
    oct_push_back(&L, id);
    ini[i] = id;
  }
  for (int i = 0; i < n; i++) {
    int cell = (int)((float)x[i] / hX);
    if (cell >= nIni) cell = nIni - 1; /* cannot happen for x < maxX-minX; guards the OOB of :559 */
    oct_node* t = &L.pool[ini[cell]];
    t->keys[t->nkeys++] = i;
  }
  free(ini);
  for (int lit = L.head; lit >= 0;) {
    oct_node* nd = &L.pool[lit];
    if (nd->nkeys == 1) { nd->bNoMore = 1; lit = nd->next; }
    else if (nd->nkeys == 0) lit = oct_erase(&L, lit);
    else lit = nd->next;
  }

  int bFinish = 0;
  size_ptr* vSize = NULL; int nvSize = 0, capvSize = 0;
#define VS_PUSH(sz, idv) do { if (nvSize == capvSize) { capvSize = capvSize ? 2 * capvSize : 256; \
    vSize = (size_ptr*)realloc(vSize, sizeof(size_ptr) * capvSize); } \
    vSize[nvSize].size = (sz); vSize[nvSize].id = (idv); nvSize++; } while (0)

  while (!bFinish) {
    int prevSize = L.size;
    int nToExpand = 0;
    nvSize = 0;
    for (int lit = L.head; lit >= 0;) {
      if (L.pool[lit].bNoMore) { lit = L.pool[lit].next; continue; }
      int c[4];
      oct_divide(&L, lit, x, y, c);
      for (int k = 0; k < 4; k++) {
        if (L.pool[c[k]].nkeys > 0) {
          oct_push_front(&L, c[k]);
          if (L.pool[c[k]].nkeys > 1) { nToExpand++; VS_PUSH(L.pool[c[k]].nkeys, c[k]); }
        } else { free(L.pool[c[k]].keys); L.pool[c[k]].keys = NULL; }
      }
      lit = oct_erase(&L, lit);
    }
    if (L.size >= N || L.size == prevSize) {
      bFinish = 1;
    } else if (L.size + nToExpand * 3 > N) {
      while (!bFinish) {
        prevSize = L.size;
        int nprev = nvSize;
        size_ptr* prev = (size_ptr*)malloc(sizeof(size_ptr) * (nprev ? nprev : 1));
        memcpy(prev, vSize, sizeof(size_ptr) * nprev);
        nvSize = 0;
        qsort(prev, nprev, sizeof(size_ptr), size_ptr_cmp);
        for (int j = nprev - 1; j >= 0; j--) {
          int c[4];
          oct_divide(&L, prev[j].id, x, y, c);
          for (int k = 0; k < 4; k++) {
            if (L.pool[c[k]].nkeys > 0) {
              oct_push_front(&L, c[k]);
              if (L.pool[c[k]].nkeys > 1) VS_PUSH(L.pool[c[k]].nkeys, c[k]);
            } else { free(L.pool[c[k]].keys); L.pool[c[k]].keys = NULL; }
          }
          oct_erase(&L, prev[j].id);
          if (L.size >= N) break;
        }
        free(prev);
        if (L.size >= N || L.size == prevSize) bFinish = 1;
      }
    }
  }
  int nout = 0;
  for (int lit = L.head; lit >= 0; lit = L.pool[lit].next) {
    oct_node* nd = &L.pool[lit];
    int best = nd->keys[0];
    float maxResponse = (float)score[best];
    for (int k = 1; k < nd->nkeys; k++)
      if ((float)score[nd->keys[k]] > maxResponse) { best = nd->keys[k]; maxResponse = (float)score[best]; }
    out_idx[nout++] = best;
  }
  for (int i = 0; i < L.npool; i++) free(L.pool[i].keys);
  free(L.pool); free(vSize);
#undef VS_PUSH
  return nout;
}

/* ---- ComputePyramid (:1041-1065) ------------------------------------------------------------------- */
static void compute_pyramid(oo_extractor* e, const uint8_t* img, int w, int h, int stride) {
  for (int level = 0; level < e->nlevels; ++level) {
    float scale = e->mvInvScaleFactor[level];
    int lw = oo_cvroundf((float)w * scale), lh = oo_cvroundf((float)h * scale);
    oo_level* l = &e->lv[level];
    free(l->pix);
    l->w = lw; l->h = lh; l->stride = lw;
    l->pix = (uint8_t*)malloc((size_t)(lw > 0 ? lw : 1) * (lh > 0 ? lh : 1));
    if (level != 0) {
      oo_level* p = &e->lv[level - 1];
      oo_resize_linear_u8(p->pix, p->w, p->h, p->stride, l->pix, lw, lh, l->stride);
    } else {
      for (int r = 0; r < h; r++) memcpy(l->pix + (size_t)r * lw, img + (size_t)r * stride, w);
    }
  }
}

static void cand_push(oo_level* l, int x, int y, int s) {
  if (l->ncand == l->cand_cap) {
    l->cand_cap = l->cand_cap ? 2 * l->cand_cap : 4096;
    l->cx = (int*)realloc(l->cx, sizeof(int) * l->cand_cap);
    l->cy = (int*)realloc(l->cy, sizeof(int) * l->cand_cap);
    l->cs = (int*)realloc(l->cs, sizeof(int) * l->cand_cap);
  }
  l->cx[l->ncand] = x; l->cy[l->ncand] = y; l->cs[l->ncand] = s; l->ncand++;
}

/* ---- ComputeKeyPointsOctTree (:733-815) ----------------------------------------------------------- */
static void compute_keypoints_octree(oo_extractor* e) {
  const float W = 30;
  for (int level = 0; level < e->nlevels; ++level) {
    oo_level* l = &e->lv[level];
    l->ncand = 0; l->nkp = 0;
    const int minBorderX = EDGE_THRESHOLD - 3, minBorderY = minBorderX;
    const int maxBorderX = l->w - EDGE_THRESHOLD + 3, maxBorderY = l->h - EDGE_THRESHOLD + 3;
    const float width = (float)(maxBorderX - minBorderX), height = (float)(maxBorderY - minBorderY);
    const int nCols = (int)(width / W), nRows = (int)(height / W);
    if (nCols < 1 || nRows < 1) continue; /* reference divides by zero (:753-754); guarded: no keypoints */
    const int wCell = (int)ceilf(width / nCols), hCell = (int)ceilf(height / nRows);
    int cell_cap = (wCell + 6) * (hCell + 6);
    int* kx = (int*)malloc(sizeof(int) * cell_cap);
    int* ky = (int*)malloc(sizeof(int) * cell_cap);
    int* ks = (int*)malloc(sizeof(int) * cell_cap);
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
        const uint8_t* roi = l->pix + (size_t)(int)iniY * l->stride + (int)iniX;
        int rc = (int)maxX - (int)iniX, rr = (int)maxY - (int)iniY;
        int nk = oo_fast9_16(roi, l->stride, rc, rr, e->iniThFAST, 1, cell_cap, kx, ky, ks);
        if (nk == 0) nk = oo_fast9_16(roi, l->stride, rc, rr, e->minThFAST, 1, cell_cap, kx, ky, ks);
        for (int k = 0; k < nk; k++) cand_push(l, kx[k] + j * wCell, ky[k] + i * hCell, ks[k]);
      }
    }
    free(kx); free(ky); free(ks);

    int* sel = (int*)malloc(sizeof(int) * (l->ncand ? l->ncand : 1));
    int nsel = oo_distribute_octree(l->cx, l->cy, l->cs, l->ncand, minBorderX, maxBorderX, minBorderY,
                                    maxBorderY, e->mnFeaturesPerLevel[level], sel);
    const int scaledPatchSize = (int)(PATCH_SIZE * e->mvScaleFactor[level]);
    free(l->kps);
    l->kps = (oo_keypoint*)malloc(sizeof(oo_keypoint) * (nsel ? nsel : 1));
    l->nkp = nsel;
    for (int i = 0; i < nsel; i++) {
      oo_keypoint* kp = &l->kps[i];
      kp->x = (float)l->cx[sel[i]] + minBorderX;
      kp->y = (float)l->cy[sel[i]] + minBorderY;
      kp->size = (float)scaledPatchSize;
      kp->angle = -1;
      kp->response = (float)l->cs[sel[i]];
      kp->octave = level;
      kp->class_id = -1;
    }
    free(sel);
  }
  for (int level = 0; level < e->nlevels; ++level) { /* computeOrientation :466-473, :813-814 */
    oo_level* l = &e->lv[level];
    for (int i = 0; i < l->nkp; i++)
      l->kps[i].angle = oo_ic_angle(l->pix, l->stride, oo_cvroundf(l->kps[i].x), oo_cvroundf(l->kps[i].y), e->umax);
  }
}

/* operator(): L/src/ORBextractor.cc:978-1039 */
int oo_extract(oo_extractor* e, const uint8_t* img, int w, int h, int stride, oo_keypoint* kps, uint8_t* desc,
               int cap, int* n_out) {
  if (!img || w <= 0 || h <= 0) { *n_out = -1; return 0; }
  compute_pyramid(e, img, w, h, stride);
  compute_keypoints_octree(e);
  int nkeypoints = 0;
  for (int level = 0; level < e->nlevels; ++level) nkeypoints += e->lv[level].nkp;
  if (nkeypoints > cap) { *n_out = nkeypoints; return -2; }
  int offset = 0;
  for (int level = 0; level < e->nlevels; ++level) {
    oo_level* l = &e->lv[level];
    free(l->blurred); l->blurred = NULL;
    if (l->nkp == 0) continue;
    l->blurred = (uint8_t*)malloc((size_t)l->w * l->h);
    oo_gaussian_blur7_u8(l->pix, l->w, l->h, l->stride, l->blurred, l->w);
    for (int i = 0; i < l->nkp; i++)
      oo_orb_descriptor(l->blurred, l->w, oo_cvroundf(l->kps[i].x), oo_cvroundf(l->kps[i].y), l->kps[i].angle,
                        desc + (size_t)(offset + i) * 32);
    for (int i = 0; i < l->nkp; i++) {
      kps[offset + i] = l->kps[i];
      if (level != 0) {
        float scale = e->mvScaleFactor[level];
        kps[offset + i].x *= scale;
        kps[offset + i].y *= scale;
      }
    }
    offset += l->nkp;
  }
  *n_out = nkeypoints;
  return 0;
}

int oo_level_size(const oo_extractor* e, int level, int* w, int* h) {
  if (level < 0 || level >= e->nlevels) return -1;
  *w = e->lv[level].w; *h = e->lv[level].h; return 0;
}
const uint8_t* oo_level_pixels(const oo_extractor* e, int level, int* stride) {
  *stride = e->lv[level].stride; return e->lv[level].pix;
}
const uint8_t* oo_level_blurred(const oo_extractor* e, int level, int* stride) {
  *stride = e->lv[level].w; return e->lv[level].blurred;
}
int oo_level_candidates(const oo_extractor* e, int level, const int** x, const int** y, const int** score) {
  *x = e->lv[level].cx; *y = e->lv[level].cy; *score = e->lv[level].cs; return e->lv[level].ncand;
}
int oo_level_keypoints(const oo_extractor* e, int level, const oo_keypoint** kps) {
  *kps = e->lv[level].kps; return e->lv[level].nkp;
}

/* ------------------------------------------------------------------------------------------ matcher */
/* DescriptorDistance: L/src/ORBmatcher.cc:1542-1556 (SWAR popcount on 8 int32 words) */
/* glibc 2.35 logf (sysdeps/ieee754/flt-32/e_logf.c, table __logf_data, LOGF_TABLE_BITS = 4, poly order 4), positive
 * normal inputs only -- MapPoint::PredictScale and Frame::mfLogScaleFactor call std::log(float) = logf on this image
 * (L/src/MapPoint.cc:416, L/src/Frame.cc:80).  Constants read from this image's libm.so.6; checked against libm over
 * every positive normal float by tests/c/logf_exhaustive.c. */
static const struct { double invc, logc; } oo_logf_tab[16] = {
  {0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2}, {0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2},
  {0x1.49539f0f010b0p+0, -0x1.01eae7f513a67p-2}, {0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3},
  {0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3}, {0x1.25e227b0b8ea0p+0, -0x1.1aa2bc79c8100p-3},
  {0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4}, {0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4},
  {0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5}, {0x1.0000000000000p+0, 0x0.0p+0},
  {0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5},  {0x1.ca4b31f026aa0p-1, 0x1.c5e53aa362eb4p-4},
  {0x1.b2036576afce6p-1, 0x1.526e57720db08p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.bc2860d224770p-3},
  {0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2},  {0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2}};
float oo_logf(float x) {
  uint32_t ix;
  memcpy(&ix, &x, 4);
  if (ix == 0x3f800000u) return 0.0f;
  if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) return logf(x); /* zero, subnormal, negative, inf, nan: not restated */
  const uint32_t tmp = ix - 0x3f330000u;
  const int i = (int)((tmp >> 19) % 16u);
  const int k = (int32_t)tmp >> 23;
  const uint32_t iz = ix - (tmp & (0x1ffu << 23));
  float zf;
  memcpy(&zf, &iz, 4);
  const double z = (double)zf;
  const double r = z * oo_logf_tab[i].invc - 1.0;
  const double y0 = oo_logf_tab[i].logc + (double)k * 0x1.62e42fefa39efp-1;
  const double r2 = r * r;
  double y = 0x1.5575b0be00b6ap-2 * r + -0x1.ffffef20a4123p-2;
  y = -0x1.00ea348b88334p-2 * r2 + y;
  y = y * r2 + (y0 + r);
  return (float)y;
}

int oo_descriptor_distance(const uint8_t* a, const uint8_t* b) {
  int dist = 0;
  for (int i = 0; i < 8; i++) {
    uint32_t pa, pb;
    memcpy(&pa, a + 4 * i, 4);
    memcpy(&pb, b + 4 * i, 4);
    uint32_t v = pa ^ pb;
    v = v - ((v >> 1) & 0x55555555);
    v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
    dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
  }
  return dist;
}

/* Frame::PosInGrid + AssignFeaturesToGrid: L/src/Frame.cc:250-263, 399-410 */
void oo_frame_build_grid(oo_frame* f) {
  const int NC = OO_GRID_COLS * OO_GRID_ROWS;
  int* cell_of = (int*)malloc(sizeof(int) * (f->n ? f->n : 1));
  memset(f->cell_start, 0, sizeof(f->cell_start));
  for (int i = 0; i < f->n; i++) {
    int posX = (int)roundf((f->keys_un[i].x - f->min_x) * f->grid_w_inv);
    int posY = (int)roundf((f->keys_un[i].y - f->min_y) * f->grid_h_inv);
    if (posX < 0 || posX >= OO_GRID_COLS || posY < 0 || posY >= OO_GRID_ROWS) { cell_of[i] = -1; continue; }
    cell_of[i] = posX * OO_GRID_ROWS + posY;
    f->cell_start[cell_of[i] + 1]++;
  }
  for (int c = 0; c < NC; c++) f->cell_start[c + 1] += f->cell_start[c];
  int* fill = (int*)calloc(NC, sizeof(int));
  for (int i = 0; i < f->n; i++)
    if (cell_of[i] >= 0) f->cell_idx[f->cell_start[cell_of[i]] + fill[cell_of[i]]++] = i;
  free(fill); free(cell_of);
}

/* Frame::GetFeaturesInArea: L/src/Frame.cc:341-397 */
int oo_features_in_area(const oo_frame* f, float x, float y, float r, int minLevel, int maxLevel,
                        int32_t* out) {
  int n = 0;
  const int nMinCellX = imax(0, (int)floorf((x - f->min_x - r) * f->grid_w_inv));
  if (nMinCellX >= OO_GRID_COLS) return 0;
  const int nMaxCellX = imin(OO_GRID_COLS - 1, (int)ceilf((x - f->min_x + r) * f->grid_w_inv));
  if (nMaxCellX < 0) return 0;
  const int nMinCellY = imax(0, (int)floorf((y - f->min_y - r) * f->grid_h_inv));
  if (nMinCellY >= OO_GRID_ROWS) return 0;
  const int nMaxCellY = imin(OO_GRID_ROWS - 1, (int)ceilf((y - f->min_y + r) * f->grid_h_inv));
  if (nMaxCellY < 0) return 0;
  const int bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
  for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
    for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
      int c = ix * OO_GRID_ROWS + iy;
      for (int j = f->cell_start[c]; j < f->cell_start[c + 1]; j++) {
        int idx = f->cell_idx[j];
        const oo_keypoint* kp = &f->keys_un[idx];
        if (bCheckLevels) {
          if (kp->octave < minLevel) continue;
          if (maxLevel >= 0 && kp->octave > maxLevel) continue;
        }
        const float distx = kp->x - x, disty = kp->y - y;
        if (fabsf(distx) < r && fabsf(disty) < r) out[n++] = idx;
      }
    }
  return n;
}

/* ComputeThreeMaxima: L/src/ORBmatcher.cc:1506-1538 */
void oo_three_maxima(const int* hs, int L, int* ind1, int* ind2, int* ind3) {
  int max1 = 0, max2 = 0, max3 = 0;
  *ind1 = *ind2 = *ind3 = -1;
  for (int i = 0; i < L; i++) {
    const int s = hs[i];
    if (s > max1) { max3 = max2; max2 = max1; max1 = s; *ind3 = *ind2; *ind2 = *ind1; *ind1 = i; }
    else if (s > max2) { max3 = max2; max2 = s; *ind3 = *ind2; *ind2 = i; }
    else if (s > max3) { max3 = s; *ind3 = i; }
  }
  if (max2 < 0.1f * (float)max1) { *ind2 = -1; *ind3 = -1; }
  else if (max3 < 0.1f * (float)max1) { *ind3 = -1; }
}

typedef struct { int* v[OO_HISTO_LENGTH]; int n[OO_HISTO_LENGTH], cap[OO_HISTO_LENGTH]; } rot_hist;
static void rh_push(rot_hist* h, int bin, int val) {
  if (h->n[bin] == h->cap[bin]) {
    h->cap[bin] = h->cap[bin] ? 2 * h->cap[bin] : 64;
    h->v[bin] = (int*)realloc(h->v[bin], sizeof(int) * h->cap[bin]);
  }
  h->v[bin][h->n[bin]++] = val;
}
static void rh_free(rot_hist* h) { for (int i = 0; i < OO_HISTO_LENGTH; i++) free(h->v[i]); }
/* the reference's bin rule: factor = 1/HISTO_LENGTH (sic), :174,:1255 */
static int rot_bin(float a1, float a2) {
  const float factor = 1.0f / OO_HISTO_LENGTH;
  float rot = a1 - a2;
  if (rot < 0.0) rot += 360.0f;
  int bin = (int)roundf(rot * factor);
  if (bin == OO_HISTO_LENGTH) bin = 0;
  return bin;
}

/* SearchByProjection(Frame&, const vector<MapPoint*>&, th): L/src/ORBmatcher.cc:45-128 */
int oo_search_by_projection_points(const oo_frame* f, const oo_query* q, int nq, float nnratio, uint8_t* blocked,
                                   int32_t* assigned) {
  int nmatches = 0;
  int32_t* vIndices = (int32_t*)malloc(sizeof(int32_t) * (f->n ? f->n : 1));
  for (int i = 0; i < nq; i++) {
    if (!q[i].valid) continue; /* !mbTrackInView || isBad() */
    int nI = oo_features_in_area(f, q[i].u, q[i].v, q[i].radius, q[i].min_level, q[i].max_level, vIndices);
    if (nI == 0) continue;
    int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
    for (int k = 0; k < nI; k++) {
      const int idx = vIndices[k];
      if (blocked[idx]) continue;
      if (f->u_right && f->u_right[idx] > 0) {
        const float er = fabsf(q[i].u_r - f->u_right[idx]);
        if (er > q[i].radius) continue;
      }
      const int dist = oo_descriptor_distance(q[i].desc, f->desc + (size_t)idx * 32);
      if (dist < bestDist) {
        bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel;
        bestLevel = f->keys_un[idx].octave; bestIdx = idx;
      } else if (dist < bestDist2) {
        bestLevel2 = f->keys_un[idx].octave; bestDist2 = dist;
      }
    }
    if (bestDist <= OO_TH_HIGH) {
      if (bestLevel == bestLevel2 && bestDist > nnratio * bestDist2) continue;
      assigned[bestIdx] = i;
      blocked[bestIdx] = (uint8_t)(q[i].blocks != 0);
      nmatches++;
    }
  }
  free(vIndices);
  return nmatches;
}

/* SearchByProjection(Frame& cur, const Frame& last, th, bMono): L/src/ORBmatcher.cc:1247-1383 */
int oo_search_by_projection_frame(const oo_frame* cur, const oo_query* q, int nq, int check_orientation,
                                  uint8_t* blocked, int32_t* assigned) {
  int nmatches = 0;
  rot_hist rh; memset(&rh, 0, sizeof(rh));
  int32_t* vIndices = (int32_t*)malloc(sizeof(int32_t) * (cur->n ? cur->n : 1));
  for (int i = 0; i < nq; i++) {
    if (!q[i].valid) continue;
    int nI = oo_features_in_area(cur, q[i].u, q[i].v, q[i].radius, q[i].min_level, q[i].max_level, vIndices);
    if (nI == 0) continue;
    int bestDist = 256, bestIdx2 = -1;
    for (int k = 0; k < nI; k++) {
      const int i2 = vIndices[k];
      if (blocked[i2]) continue;
      if (cur->u_right && cur->u_right[i2] > 0) {
        const float er = fabsf(q[i].u_r - cur->u_right[i2]);
        if (er > q[i].radius) continue;
      }
      const int dist = oo_descriptor_distance(q[i].desc, cur->desc + (size_t)i2 * 32);
      if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
    }
    if (bestDist <= OO_TH_HIGH) {
      assigned[bestIdx2] = i;
      blocked[bestIdx2] = (uint8_t)(q[i].blocks != 0);
      nmatches++;
      if (check_orientation) rh_push(&rh, rot_bin(q[i].angle, cur->keys_un[bestIdx2].angle), bestIdx2);
    }
  }
  if (check_orientation) {
    int i1, i2, i3;
    oo_three_maxima(rh.n, OO_HISTO_LENGTH, &i1, &i2, &i3);
    for (int i = 0; i < OO_HISTO_LENGTH; i++)
      if (i != i1 && i != i2 && i != i3)
        for (int j = 0; j < rh.n[i]; j++) { assigned[rh.v[i][j]] = -1; blocked[rh.v[i][j]] = 0; nmatches--; }   /* the slot is NULL again */
  }
  rh_free(&rh);
  free(vIndices);
  return nmatches;
}

/* Frame::UnprojectStereo: L/src/Frame.cc:668-679 (mRwc * x3Dc + mOw through the cv::gemm small-matrix path, see
 * oo_is_in_frustum); the map point created from the keypoint carries its descriptor (L/src/MapPoint.cc:57-86). */
void oo_unproject_stereo(const oo_unproject_cam* cam, const oo_keypoint* kp, float z, const uint8_t* desc, int observed,
                         oo_last_point* out) {
  memset(out, 0, sizeof(*out));
  if (z > 0) {
    const float u = kp->x, v = kp->y;
    const float x = (u - cam->cx) * z * cam->invfx;
    const float y = (v - cam->cy) * z * cam->invfy;
    for (int r = 0; r < 3; r++) {
      const float* a = cam->Rwc + 3 * r;
      const float t = a[0] * x + a[1] * y + a[2] * z;
      out->pos[r] = (float)((double)t * 1.0 + (double)cam->Ow[r] * 1.0);
    }
    out->valid = 1;
  }
  out->observed = observed != 0;
  out->octave = kp->octave;
  out->angle = kp->angle;
  memcpy(out->desc, desc, 32);
}

/* L/src/ORBmatcher.cc:1270-1308, 1326-1327 */
void oo_track_query(const oo_track_pose* P, const oo_last_point* lp, oo_query* q) {
  memset(q, 0, sizeof(*q));
  if (!lp->valid) return;                                     /* pMP == NULL || mvbOutlier[i] */
  float x3Dc[3];
  for (int r = 0; r < 3; r++) {
    const float* a = P->Rcw + 3 * r;
    const float t = a[0] * lp->pos[0] + a[1] * lp->pos[1] + a[2] * lp->pos[2];
    x3Dc[r] = (float)((double)t * 1.0 + (double)P->tcw[r] * 1.0);
  }
  const float xc = x3Dc[0], yc = x3Dc[1];
  const float invzc = 1.0 / x3Dc[2];
  if (invzc < 0) return;
  float u = P->fx * xc * invzc + P->cx;
  float v = P->fy * yc * invzc + P->cy;
  if (u < P->min_x || u > P->max_x) return;
  if (v < P->min_y || v > P->max_y) return;
  const int nLastOctave = lp->octave;
  const float radius = P->th * P->scale_factors[nLastOctave];   /* :1297, unmasked as in the reference */
  q->u = u; q->v = v; q->radius = radius;
  q->u_r = u - P->mbf * invzc;
  if (P->forward) { q->min_level = nLastOctave; q->max_level = -1; }
  else if (P->backward) { q->min_level = 0; q->max_level = nLastOctave; }
  else { q->min_level = nLastOctave - 1; q->max_level = nLastOctave + 1; }
  q->valid = 1;
  q->blocks = lp->observed != 0;
  q->angle = lp->angle;
  memcpy(q->desc, lp->desc, 32);
}

void oo_unproject_stereo_n(const oo_unproject_cam* cam, const oo_keypoint* kps, const float* depth, const uint8_t* desc, int n,
                           int observed, oo_last_point* out) {
  for (int i = 0; i < n; i++) oo_unproject_stereo(cam, &kps[i], depth[i], desc + (size_t)i * 32, observed, &out[i]);
}
void oo_track_queries_n(const oo_track_pose* pose, const oo_last_point* lp, int n, oo_query* q) {
  for (int i = 0; i < n; i++) oo_track_query(pose, &lp[i], &q[i]);
}

/* MapPoint::PredictScale(const float&, Frame*): L/src/MapPoint.cc:409-423.  `using namespace ::std` (:25) makes
 * log(float) -> logf and ceil(float) -> ceilf. */
int oo_predict_scale(float max_distance, float current_dist, float log_scale_factor, int n_levels) {
  const float ratio = max_distance / current_dist;
  int nScale = (int)ceilf(oo_logf(ratio) / log_scale_factor);
  if (nScale < 0) nScale = 0;
  else if (nScale >= n_levels) nScale = n_levels - 1;
  return nScale;
}

/* Frame::isInFrustum: L/src/Frame.cc:284-339.  The cv::Mat expressions, as OpenCV 4.5 evaluates them for CV_32F 3x3 /
 * 3x1 operands (generic build, no FMA contraction):
 *   mRcw*P + mtcw -> one gemm(A,B,1,C,1): float t = a0*b0 + a1*b1 + a2*b2, d = (float)(t*1.0 + c*1.0)   (small-matrix path)
 *   P - mOw       -> float subtraction per element
 *   cv::norm(PO)  -> sqrt of a double sum of squares, accumulated in element order
 *   PO.dot(Pn)    -> double sum of products in element order */
int oo_is_in_frustum(const oo_frustum* fr, const oo_map_point* mp, float viewingCosLimit, oo_track* out) {
  memset(out, 0, sizeof(*out));                    /* pMP->mbTrackInView = false */
  const float* P = mp->pos;
  float Pc[3];
  for (int r = 0; r < 3; r++) {
    const float* a = fr->Rcw + 3 * r;
    const float t = a[0] * P[0] + a[1] * P[1] + a[2] * P[2];
    Pc[r] = (float)((double)t * 1.0 + (double)fr->tcw[r] * 1.0);
  }
  const float PcX = Pc[0], PcY = Pc[1], PcZ = Pc[2];
  if (PcZ < 0.0f) return 0;
  const float invz = 1.0f / PcZ;
  const float u = fr->fx * PcX * invz + fr->cx;
  const float v = fr->fy * PcY * invz + fr->cy;
  if (u < fr->min_x || u > fr->max_x) return 0;
  if (v < fr->min_y || v > fr->max_y) return 0;
  const float maxDistance = 1.2f * mp->max_distance;   /* GetMaxDistanceInvariance, MapPoint.cc:383-386 */
  const float minDistance = 0.8f * mp->min_distance;   /* GetMinDistanceInvariance, MapPoint.cc:378-381 */
  float PO[3];
  for (int k = 0; k < 3; k++) PO[k] = P[k] - fr->Ow[k];
  double s = 0;
  for (int k = 0; k < 3; k++) { const double e = (double)PO[k]; s += e * e; }
  const float dist = (float)sqrt(s);
  if (dist < minDistance || dist > maxDistance) return 0;
  double dot = 0;
  for (int k = 0; k < 3; k++) dot += (double)PO[k] * (double)mp->normal[k];
  const float viewCos = (float)(dot / (double)dist);
  if (viewCos < viewingCosLimit) return 0;
  const int nPredictedLevel = oo_predict_scale(mp->max_distance, dist, fr->log_scale_factor, fr->n_levels);
  out->in_view = 1;
  out->proj_x = u;
  out->proj_xr = u - fr->mbf * invz;
  out->proj_y = v;
  out->level = nPredictedLevel;
  out->view_cos = viewCos;
  return 1;
}

/* L/src/ORBmatcher.cc:52-71 (+ RadiusByViewingCos :130-135, which compares the float with the double 0.998) */
void oo_local_point_query(const oo_frustum* fr, const oo_map_point* mp, const oo_track* tr, float th, oo_query* q) {
  memset(q, 0, sizeof(*q));
  if (!tr->in_view || mp->skip) return;   /* skip covers isBad(); a point seen in this frame keeps mbTrackInView = false (Tracking.cc:1044) */
  const int bFactor = (double)th != 1.0;
  float r = ((double)tr->view_cos > 0.998) ? 2.5f : 4.0f;
  if (bFactor) r *= th;
  q->u = tr->proj_x; q->v = tr->proj_y; q->u_r = tr->proj_xr;
  q->radius = r * fr->scale_factors[tr->level];
  q->min_level = tr->level - 1;
  q->max_level = tr->level;
  q->valid = 1;
  q->blocks = mp->observed != 0;
  memcpy(q->desc, mp->desc, 32);
}

/* Tracking::SearchLocalPoints: L/src/Tracking.cc:1050-1078 */
int oo_search_local_points(const oo_frame* f, const oo_frustum* fr, const oo_map_point* mp, int n, float th, float nnratio,
                           oo_track* track, uint8_t* blocked, int32_t* assigned, int* n_to_match) {
  int nToMatch = 0;
  for (int i = 0; i < n; i++) {
    memset(&track[i], 0, sizeof(oo_track));
    if (mp[i].skip) continue;
    if (oo_is_in_frustum(fr, &mp[i], 0.5f, &track[i])) nToMatch++;
  }
  *n_to_match = nToMatch;
  if (nToMatch == 0) return 0;
  oo_query* q = (oo_query*)malloc(sizeof(oo_query) * (size_t)(n ? n : 1));
  for (int i = 0; i < n; i++) oo_local_point_query(fr, &mp[i], &track[i], th, &q[i]);
  const int nm = oo_search_by_projection_points(f, q, n, nnratio, blocked, assigned);
  free(q);
  return nm;
}

/* SearchByProjection(Frame&, KeyFrame*, set, th, ORBdist): L/src/ORBmatcher.cc:1439-1501 */
int oo_search_by_projection_keyframe(const oo_frame* cur, const oo_query* q, int nq, int check_orientation, int orb_dist,
                                     uint8_t* mappoint_set, int32_t* assigned) {
  int nmatches = 0;
  rot_hist rh; memset(&rh, 0, sizeof(rh));
  int32_t* vIndices2 = (int32_t*)malloc(sizeof(int32_t) * (cur->n ? cur->n : 1));
  for (int i = 0; i < nq; i++) {
    if (!q[i].valid) continue;
    const int nI = oo_features_in_area(cur, q[i].u, q[i].v, q[i].radius, q[i].min_level, q[i].max_level, vIndices2);
    if (nI == 0) continue;
    int bestDist = 256, bestIdx2 = -1;
    for (int k = 0; k < nI; k++) {
      const int i2 = vIndices2[k];
      if (mappoint_set[i2]) continue;                         /* :1453 */
      const int dist = oo_descriptor_distance(q[i].desc, cur->desc + (size_t)i2 * 32);
      if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
    }
    if (bestDist <= orb_dist) {                               /* :1466 */
      mappoint_set[bestIdx2] = 1;
      assigned[bestIdx2] = i;
      nmatches++;
      if (check_orientation) rh_push(&rh, rot_bin(q[i].angle, cur->keys_un[bestIdx2].angle), bestIdx2);
    }
  }
  if (check_orientation) {
    int i1, i2, i3;
    oo_three_maxima(rh.n, OO_HISTO_LENGTH, &i1, &i2, &i3);
    for (int i = 0; i < OO_HISTO_LENGTH; i++)
      if (i != i1 && i != i2 && i != i3)
        for (int j = 0; j < rh.n[i]; j++) { assigned[rh.v[i][j]] = -1; mappoint_set[rh.v[i][j]] = 0; nmatches--; }   /* :1496 = NULL */
  }
  rh_free(&rh);
  free(vIndices2);
  return nmatches;
}

static int featvec_lower_bound(const oo_featvec_node* nodes, int n, int key) {
  int lo = 0, hi = n;
  while (lo < hi) { int mid = (lo + hi) / 2; if (nodes[mid].node_id < key) lo = mid + 1; else hi = mid; }
  return lo;
}

/* SearchByBoW(KeyFrame*, Frame&, vector<MapPoint*>&): L/src/ORBmatcher.cc:161-273 */
int oo_search_by_bow(const uint8_t* descA, const float* angleA, const uint8_t* validA,
                     const oo_featvec_node* nodesA, int nA_nodes, const int32_t* idxA, const uint8_t* descB,
                     const float* angleB, int nB, const oo_featvec_node* nodesB, int nB_nodes,
                     const int32_t* idxB, float nnratio, int check_orientation, int32_t* matchB) {
  int nmatches = 0;
  rot_hist rh; memset(&rh, 0, sizeof(rh));
  for (int j = 0; j < nB; j++) matchB[j] = -1;
  int ia = 0, ib = 0;
  while (ia < nA_nodes && ib < nB_nodes) {
    if (nodesA[ia].node_id == nodesB[ib].node_id) {
      for (int iKF = 0; iKF < nodesA[ia].count; iKF++) {
        const int realIdxKF = idxA[nodesA[ia].start + iKF];
        if (!validA[realIdxKF]) continue;
        const uint8_t* dKF = descA + (size_t)realIdxKF * 32;
        int bestDist1 = 256, bestIdxF = -1, bestDist2 = 256;
        for (int iF = 0; iF < nodesB[ib].count; iF++) {
          const int realIdxF = idxB[nodesB[ib].start + iF];
          if (matchB[realIdxF] >= 0) continue;
          const int dist = oo_descriptor_distance(dKF, descB + (size_t)realIdxF * 32);
          if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdxF = realIdxF; }
          else if (dist < bestDist2) bestDist2 = dist;
        }
        if (bestDist1 <= OO_TH_LOW) {
          if ((float)bestDist1 < nnratio * (float)bestDist2) {
            matchB[bestIdxF] = realIdxKF;
            if (check_orientation) rh_push(&rh, rot_bin(angleA[realIdxKF], angleB[bestIdxF]), bestIdxF);
            nmatches++;
          }
        }
      }
      ia++; ib++;
    } else if (nodesA[ia].node_id < nodesB[ib].node_id) {
      ia = featvec_lower_bound(nodesA, nA_nodes, nodesB[ib].node_id);
    } else {
      ib = featvec_lower_bound(nodesB, nB_nodes, nodesA[ia].node_id);
    }
  }
  if (check_orientation) {
    int i1, i2, i3;
    oo_three_maxima(rh.n, OO_HISTO_LENGTH, &i1, &i2, &i3);
    for (int i = 0; i < OO_HISTO_LENGTH; i++) {
      if (i == i1 || i == i2 || i == i3) continue;
      for (int j = 0; j < rh.n[i]; j++) { matchB[rh.v[i][j]] = -1; nmatches--; }
    }
  }
  rh_free(&rh);
  return nmatches;
}

/* SearchByBoW(KeyFrame*, KeyFrame*, vector<MapPoint*>&): L/src/ORBmatcher.cc:494-612 */
int oo_search_by_bow_kf(const uint8_t* descA, const float* angleA, const uint8_t* validA, int nA,
                        const oo_featvec_node* nodesA, int nA_nodes, const int32_t* idxA, const uint8_t* descB,
                        const float* angleB, const uint8_t* validB, int nB, const oo_featvec_node* nodesB, int nB_nodes,
                        const int32_t* idxB, float nnratio, int check_orientation, int32_t* matchA) {
  int nmatches = 0;
  rot_hist rh; memset(&rh, 0, sizeof(rh));
  uint8_t* vbMatched2 = (uint8_t*)calloc(nB ? nB : 1, 1);
  for (int i = 0; i < nA; i++) matchA[i] = -1;
  int ia = 0, ib = 0;
  while (ia < nA_nodes && ib < nB_nodes) {
    if (nodesA[ia].node_id == nodesB[ib].node_id) {
      for (int i1 = 0; i1 < nodesA[ia].count; i1++) {
        const int idx1 = idxA[nodesA[ia].start + i1];
        if (!validA[idx1]) continue;
        const uint8_t* d1 = descA + (size_t)idx1 * 32;
        int bestDist1 = 256, bestIdx2 = -1, bestDist2 = 256;
        for (int i2 = 0; i2 < nodesB[ib].count; i2++) {
          const int idx2 = idxB[nodesB[ib].start + i2];
          if (vbMatched2[idx2] || !validB[idx2]) continue;
          const int dist = oo_descriptor_distance(d1, descB + (size_t)idx2 * 32);
          if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdx2 = idx2; }
          else if (dist < bestDist2) bestDist2 = dist;
        }
        if (bestDist1 < OO_TH_LOW) {
          if ((float)bestDist1 < nnratio * (float)bestDist2) {
            matchA[idx1] = bestIdx2;
            vbMatched2[bestIdx2] = 1;
            if (check_orientation) rh_push(&rh, rot_bin(angleA[idx1], angleB[bestIdx2]), idx1);
            nmatches++;
          }
        }
      }
      ia++; ib++;
    } else if (nodesA[ia].node_id < nodesB[ib].node_id) {
      ia = featvec_lower_bound(nodesA, nA_nodes, nodesB[ib].node_id);
    } else {
      ib = featvec_lower_bound(nodesB, nB_nodes, nodesA[ia].node_id);
    }
  }
  if (check_orientation) {
    int i1, i2, i3;
    oo_three_maxima(rh.n, OO_HISTO_LENGTH, &i1, &i2, &i3);
    for (int i = 0; i < OO_HISTO_LENGTH; i++) {
      if (i == i1 || i == i2 || i == i3) continue;
      for (int j = 0; j < rh.n[i]; j++) { matchA[rh.v[i][j]] = -1; nmatches--; }
    }
  }
  rh_free(&rh);
  free(vbMatched2);
  return nmatches;
}

/* Fuse: L/src/ORBmatcher.cc:818-868; Fuse(Sim3): :983-1009; SearchBySim3: :1118-1147, 1194-1223.  KeyFrame::GetFeaturesInArea
 * (L/src/KeyFrame.cc:526-567) is Frame's grid walk without the level filter; the level test follows inside the loop. */
void oo_proj_best(const oo_frame* kf, const oo_query* q, int nq, int gate, const float* inv_level_sigma2, int32_t* best_idx,
                  int32_t* best_dist) {
  int32_t* vIndices = (int32_t*)malloc(sizeof(int32_t) * (kf->n ? kf->n : 1));
  for (int i = 0; i < nq; i++) {
    best_idx[i] = -1; best_dist[i] = 256;
    if (!q[i].valid) continue;
    const float u = q[i].u, v = q[i].v, ur = q[i].u_r;
    const int nI = oo_features_in_area(kf, u, v, q[i].radius, -1, -1, vIndices);
    int bestDist = 256, bestIdx = -1;
    for (int k = 0; k < nI; k++) {
      const int idx = vIndices[k];
      const oo_keypoint* kp = &kf->keys_un[idx];
      const int kpLevel = kp->octave;
      if (kpLevel < q[i].min_level || kpLevel > q[i].max_level) continue;
      if (gate == 2) {
        if (kf->u_right && kf->u_right[idx] >= 0) {
          const float ex = u - kp->x, ey = v - kp->y, er = ur - kf->u_right[idx];
          const float e2 = ex * ex + ey * ey + er * er;
          if (e2 * inv_level_sigma2[kpLevel] > 7.8) continue;
        } else {
          const float ex = u - kp->x, ey = v - kp->y;
          const float e2 = ex * ex + ey * ey;
          if (e2 * inv_level_sigma2[kpLevel] > 5.99) continue;
        }
      }
      const int dist = oo_descriptor_distance(q[i].desc, kf->desc + (size_t)idx * 32);
      if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
    }
    best_idx[i] = bestIdx; best_dist[i] = bestIdx >= 0 ? bestDist : 256;
  }
  free(vIndices);
}

/* ---- whole-function restatements of the keyframe-rate projection searches -------------------------------------------------
 * Each function below walks its candidate points in order exactly as the reference loop does and returns, per point, the
 * keypoint the reference would act on (best_idx, -1 = none), its distance, the predicted level and the projection.  The pose
 * algebra in front of the loops (decomposing Scw, composing sR21 / t21) and the map bookkeeping behind them (Replace,
 * AddObservation, AddMapPoint, vpReplacePoint, the mutual check) are the caller's: they act on SLAM objects.
 * cv::Mat expressions as in oo_is_in_frustum: A*x + b = small-matrix gemm (float dot, double epilogue), cv::norm and
 * Mat::dot accumulate in double. */
static void kf_gemm3(const float* R, const float* x, const float* t, float* out) {
  for (int r = 0; r < 3; r++) {
    const float* a = R + 3 * r;
    const float d = a[0] * x[0] + a[1] * x[1] + a[2] * x[2];
    out[r] = (float)((double)d * 1.0 + (double)t[r] * 1.0);
  }
}
static float kf_norm3(const float* v) {
  double s = 0;
  for (int k = 0; k < 3; k++) { const double e = (double)v[k]; s += e * e; }
  return (float)sqrt(s);
}
static int kf_is_in_image(const oo_kf_camera* c, float x, float y) {   /* KeyFrame::IsInImage, L/src/KeyFrame.cc:569-571 */
  return x >= c->min_x && x < c->max_x && y >= c->min_y && y < c->max_y;
}
static void kf_res_init(oo_kf_result* r) { r->best_idx = -1; r->best_dist = 256; r->level = -1; r->u = r->v = r->u_r = 0.f; }

/* ORBmatcher::Fuse(KeyFrame*, const vector<MapPoint*>&, th): L/src/ORBmatcher.cc:781-884 (the loop body up to the map update) */
void oo_fuse(const oo_frame* kf, const float* inv_level_sigma2, const oo_kf_camera* cam, const oo_kf_point* pts, int n, oo_kf_result* res) {
  int32_t* vIndices = (int32_t*)malloc(sizeof(int32_t) * (kf->n ? kf->n : 1));
  for (int i = 0; i < n; i++) {
    kf_res_init(&res[i]);
    const oo_kf_point* pMP = &pts[i];
    if (pMP->skip) continue;                                  /* !pMP, isBad(), IsInKeyFrame(pKF)  :784-788 */
    float p3Dc[3];
    kf_gemm3(cam->R, pMP->pos, cam->t, p3Dc);                 /* Rcw * p3Dw + tcw  :791 */
    if (p3Dc[2] < 0.0f) continue;                             /* :794 */
    const float invz = 1 / p3Dc[2];                           /* :797 */
    const float x = p3Dc[0] * invz, y = p3Dc[1] * invz;
    const float u = cam->fx * x + cam->cx, v = cam->fy * y + cam->cy;
    if (!kf_is_in_image(cam, u, v)) continue;                 /* :805 */
    const float ur = u - cam->mbf * invz;                     /* :808 */
    const float maxDistance = 1.2f * pMP->max_distance, minDistance = 0.8f * pMP->min_distance;
    float PO[3];
    for (int k = 0; k < 3; k++) PO[k] = pMP->pos[k] - cam->Ow[k];
    const float dist3D = kf_norm3(PO);
    if (dist3D < minDistance || dist3D > maxDistance) continue;   /* :816 */
    double dot = 0;
    for (int k = 0; k < 3; k++) dot += (double)PO[k] * (double)pMP->normal[k];
    if (dot < 0.5 * dist3D) continue;                         /* :822 */
    const int nPredictedLevel = oo_predict_scale(pMP->max_distance, dist3D, cam->log_scale_factor, cam->n_levels);
    const float radius = cam->th * cam->scale_factors[nPredictedLevel];
    res[i].level = nPredictedLevel; res[i].u = u; res[i].v = v; res[i].u_r = ur;
    const int nI = oo_features_in_area(kf, u, v, radius, -1, -1, vIndices);   /* KeyFrame::GetFeaturesInArea: no level filter */
    int bestDist = 256, bestIdx = -1;
    for (int k = 0; k < nI; k++) {
      const int idx = vIndices[k];
      const oo_keypoint* kp = &kf->keys_un[idx];
      const int kpLevel = kp->octave;
      if (kpLevel < nPredictedLevel - 1 || kpLevel > nPredictedLevel) continue;
      if (kf->u_right && kf->u_right[idx] >= 0) {             /* stereo reprojection error :845-856 */
        const float ex = u - kp->x, ey = v - kp->y, er = ur - kf->u_right[idx];
        const float e2 = ex * ex + ey * ey + er * er;
        if (e2 * inv_level_sigma2[kpLevel] > 7.8) continue;
      } else {
        const float ex = u - kp->x, ey = v - kp->y;
        const float e2 = ex * ex + ey * ey;
        if (e2 * inv_level_sigma2[kpLevel] > 5.99) continue;
      }
      const int dist = oo_descriptor_distance(pMP->desc, kf->desc + (size_t)idx * 32);
      if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
    }
    res[i].best_idx = bestIdx; res[i].best_dist = bestIdx >= 0 ? bestDist : 256;
  }
  free(vIndices);
}

/* Fuse(KeyFrame*, cv::Mat Scw, ...) :932-1008 (mode 0) and one direction of SearchBySim3 :1063-1146 / :1149-1222 (mode 1) */
static void kf_independent(const oo_frame* kf, const oo_kf_camera* cam, const oo_kf_point* pts, int n, int sim3, oo_kf_result* res) {
  int32_t* vIndices = (int32_t*)malloc(sizeof(int32_t) * (kf->n ? kf->n : 1));
  for (int i = 0; i < n; i++) {
    kf_res_init(&res[i]);
    const oo_kf_point* pMP = &pts[i];
    if (pMP->skip) continue;
    float p3Dc[3];
    kf_gemm3(cam->R, pMP->pos, cam->t, p3Dc);
    if (sim3) {                                               /* p3Dc2 = sR21 * p3Dc1 + t21  :1076 */
      float p2[3];
      kf_gemm3(cam->R2, p3Dc, cam->t2, p2);
      p3Dc[0] = p2[0]; p3Dc[1] = p2[1]; p3Dc[2] = p2[2];
    }
    if (p3Dc[2] < 0.0f) continue;
    const float invz = (float)(1.0 / p3Dc[2]);                /* `1.0 / z`: double division  :945, :1083 */
    const float x = p3Dc[0] * invz, y = p3Dc[1] * invz;
    const float u = cam->fx * x + cam->cx, v = cam->fy * y + cam->cy;
    if (!kf_is_in_image(cam, u, v)) continue;
    const float maxDistance = 1.2f * pMP->max_distance, minDistance = 0.8f * pMP->min_distance;
    float dist3D;
    if (sim3) {
      dist3D = kf_norm3(p3Dc);                                /* cv::norm(p3Dc2)  :1097 */
    } else {
      float PO[3];
      for (int k = 0; k < 3; k++) PO[k] = pMP->pos[k] - cam->Ow[k];
      dist3D = kf_norm3(PO);
      if (dist3D < minDistance || dist3D > maxDistance) continue;
      double dot = 0;
      for (int k = 0; k < 3; k++) dot += (double)PO[k] * (double)pMP->normal[k];
      if (dot < 0.5 * dist3D) continue;                       /* :969 */
    }
    if (dist3D < minDistance || dist3D > maxDistance) continue;
    const int nPredictedLevel = oo_predict_scale(pMP->max_distance, dist3D, cam->log_scale_factor, cam->n_levels);
    const float radius = cam->th * cam->scale_factors[nPredictedLevel];
    res[i].level = nPredictedLevel; res[i].u = u; res[i].v = v; res[i].u_r = u - cam->mbf * invz;
    const int nI = oo_features_in_area(kf, u, v, radius, -1, -1, vIndices);
    int bestDist = 0x7fffffff, bestIdx = -1;                  /* INT_MAX  :988, :1120 */
    for (int k = 0; k < nI; k++) {
      const int idx = vIndices[k];
      const int kpLevel = kf->keys_un[idx].octave;
      if (kpLevel < nPredictedLevel - 1 || kpLevel > nPredictedLevel) continue;
      const int dist = oo_descriptor_distance(pMP->desc, kf->desc + (size_t)idx * 32);
      if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
    }
    res[i].best_idx = bestIdx; res[i].best_dist = bestIdx >= 0 ? bestDist : 256;
  }
  free(vIndices);
}
void oo_fuse_sim3(const oo_frame* kf, const oo_kf_camera* cam, const oo_kf_point* pts, int n, oo_kf_result* res) { kf_independent(kf, cam, pts, n, 0, res); }
void oo_search_by_sim3_dir(const oo_frame* kf, const oo_kf_camera* cam, const oo_kf_point* pts, int n, oo_kf_result* res) { kf_independent(kf, cam, pts, n, 1, res); }

/* SearchByProjection(KeyFrame*, cv::Mat Scw, vpPoints, vpMatched, th): L/src/ORBmatcher.cc:298-383.  matched[idx] != 0 <=>
 * vpMatched[idx] != NULL (updated).  res[i].best_idx = the keypoint point i was written to.  Returns nmatches. */
int oo_search_by_projection_loop(const oo_frame* kf, const oo_kf_camera* cam, const oo_kf_point* pts, int n, int th_low, uint8_t* matched,
                                 oo_kf_result* res) {
  int nmatches = 0;
  int32_t* vIndices = (int32_t*)malloc(sizeof(int32_t) * (kf->n ? kf->n : 1));
  for (int i = 0; i < n; i++) {
    kf_res_init(&res[i]);
    const oo_kf_point* pMP = &pts[i];
    if (pMP->skip) continue;                                  /* isBad() || spAlreadyFound.count(pMP)  :302 */
    float p3Dc[3];
    kf_gemm3(cam->R, pMP->pos, cam->t, p3Dc);
    if (p3Dc[2] < 0.0) continue;
    const float invz = 1 / p3Dc[2];                           /* :317 */
    const float x = p3Dc[0] * invz, y = p3Dc[1] * invz;
    const float u = cam->fx * x + cam->cx, v = cam->fy * y + cam->cy;
    if (!kf_is_in_image(cam, u, v)) continue;
    const float maxDistance = 1.2f * pMP->max_distance, minDistance = 0.8f * pMP->min_distance;
    float PO[3];
    for (int k = 0; k < 3; k++) PO[k] = pMP->pos[k] - cam->Ow[k];
    const float dist = kf_norm3(PO);
    if (dist < minDistance || dist > maxDistance) continue;
    double dot = 0;
    for (int k = 0; k < 3; k++) dot += (double)PO[k] * (double)pMP->normal[k];
    if (dot < 0.5 * dist) continue;
    const int nPredictedLevel = oo_predict_scale(pMP->max_distance, dist, cam->log_scale_factor, cam->n_levels);
    const float radius = cam->th * cam->scale_factors[nPredictedLevel];
    res[i].level = nPredictedLevel; res[i].u = u; res[i].v = v; res[i].u_r = u - cam->mbf * invz;
    const int nI = oo_features_in_area(kf, u, v, radius, -1, -1, vIndices);
    if (nI == 0) continue;
    int bestDist = 256, bestIdx = -1;
    for (int k = 0; k < nI; k++) {
      const int idx = vIndices[k];
      if (matched[idx]) continue;                             /* :358 */
      const int kpLevel = kf->keys_un[idx].octave;
      if (kpLevel < nPredictedLevel - 1 || kpLevel > nPredictedLevel) continue;
      const int d = oo_descriptor_distance(pMP->desc, kf->desc + (size_t)idx * 32);
      if (d < bestDist) { bestDist = d; bestIdx = idx; }
    }
    if (bestDist <= th_low) {                                 /* :377 */
      matched[bestIdx] = 1;
      res[i].best_idx = bestIdx; res[i].best_dist = bestDist;
      nmatches++;
    }
  }
  free(vIndices);
  return nmatches;
}

/* the projection part of SearchByProjection(Frame&, KeyFrame*, sAlreadyFound, th, ORBdist): L/src/ORBmatcher.cc:1403-1437
 * -> the query oo_search_by_projection_keyframe consumes (valid = 0 where the reference `continue`s) */
void oo_reloc_query(const oo_kf_camera* cam, const oo_kf_point* pMP, oo_query* q) {
  memset(q, 0, sizeof(*q));
  if (pMP->skip) return;
  float x3Dc[3];
  kf_gemm3(cam->R, pMP->pos, cam->t, x3Dc);
  const float xc = x3Dc[0], yc = x3Dc[1];
  const float invzc = (float)(1.0 / x3Dc[2]);
  const float u = cam->fx * xc * invzc + cam->cx;
  const float v = cam->fy * yc * invzc + cam->cy;
  if (u < cam->min_x || u > cam->max_x) return;
  if (v < cam->min_y || v > cam->max_y) return;
  float PO[3];
  for (int k = 0; k < 3; k++) PO[k] = pMP->pos[k] - cam->Ow[k];
  const float dist3D = kf_norm3(PO);
  const float maxDistance = 1.2f * pMP->max_distance, minDistance = 0.8f * pMP->min_distance;
  if (dist3D < minDistance || dist3D > maxDistance) return;
  const int nPredictedLevel = oo_predict_scale(pMP->max_distance, dist3D, cam->log_scale_factor, cam->n_levels);
  q->u = u; q->v = v; q->u_r = u - cam->mbf * invzc;
  q->radius = cam->th * cam->scale_factors[nPredictedLevel];
  q->min_level = nPredictedLevel - 1; q->max_level = nPredictedLevel + 1;
  q->valid = 1; q->blocks = 1; q->angle = pMP->angle;
  memcpy(q->desc, pMP->desc, 32);
}

/* ORBmatcher::CheckDistEpipolarLine: L/src/ORBmatcher.cc:137-159 */
static int check_dist_epipolar_line(const oo_keypoint* kp1, const oo_keypoint* kp2, const float* F12, const float* level_sigma2) {
  const float a = kp1->x * F12[0] + kp1->y * F12[3] + F12[6];
  const float b = kp1->x * F12[1] + kp1->y * F12[4] + F12[7];
  const float c = kp1->x * F12[2] + kp1->y * F12[5] + F12[8];
  const float num = a * kp2->x + b * kp2->y + c;
  const float den = a * a + b * b;
  if (den == 0) return 0;
  const float dsqr = num * num / den;
  return dsqr < 3.84 * level_sigma2[kp2->octave];
}

/* SearchForTriangulation: L/src/ORBmatcher.cc:614-764 (the epipole :622-630 arrives in ep) */
int oo_search_for_triangulation(const oo_keypoint* keysA, const uint8_t* descA, const float* u_rightA, const uint8_t* has_mpA, int nA,
                                const oo_featvec_node* nodesA, int nA_nodes, const int32_t* idxA, const oo_keypoint* keysB,
                                const uint8_t* descB, const float* u_rightB, const uint8_t* has_mpB, int nB,
                                const oo_featvec_node* nodesB, int nB_nodes, const int32_t* idxB, const oo_epipolar* ep,
                                int bOnlyStereo, int check_orientation, int32_t* vMatches12) {
  int nmatches = 0;
  rot_hist rh; memset(&rh, 0, sizeof(rh));
  uint8_t* vbMatched2 = (uint8_t*)calloc(nB ? nB : 1, 1);   /* declared and tested by the reference, never set (:635,679) */
  for (int i = 0; i < nA; i++) vMatches12[i] = -1;
  int ia = 0, ib = 0;
  while (ia < nA_nodes && ib < nB_nodes) {
    if (nodesA[ia].node_id == nodesB[ib].node_id) {
      for (int i1 = 0; i1 < nodesA[ia].count; i1++) {
        const int idx1 = idxA[nodesA[ia].start + i1];
        if (has_mpA[idx1]) continue;
        const int bStereo1 = u_rightA && u_rightA[idx1] >= 0;
        if (bOnlyStereo && !bStereo1) continue;
        const oo_keypoint* kp1 = &keysA[idx1];
        const uint8_t* d1 = descA + (size_t)idx1 * 32;
        int bestDist = OO_TH_LOW, bestIdx2 = -1;
        for (int i2 = 0; i2 < nodesB[ib].count; i2++) {
          const int idx2 = idxB[nodesB[ib].start + i2];
          if (vbMatched2[idx2] || has_mpB[idx2]) continue;
          const int bStereo2 = u_rightB && u_rightB[idx2] >= 0;
          if (bOnlyStereo && !bStereo2) continue;
          const int dist = oo_descriptor_distance(d1, descB + (size_t)idx2 * 32);
          if (dist > OO_TH_LOW || dist > bestDist) continue;
          const oo_keypoint* kp2 = &keysB[idx2];
          if (!bStereo1 && !bStereo2) {
            const float distex = ep->ex - kp2->x, distey = ep->ey - kp2->y;
            if (distex * distex + distey * distey < 100 * ep->scale_factors[kp2->octave]) continue;
          }
          if (check_dist_epipolar_line(kp1, kp2, ep->F12, ep->level_sigma2)) { bestIdx2 = idx2; bestDist = dist; }
        }
        if (bestIdx2 >= 0) {
          vMatches12[idx1] = bestIdx2;
          nmatches++;
          if (check_orientation) rh_push(&rh, rot_bin(kp1->angle, keysB[bestIdx2].angle), idx1);
        }
      }
      ia++; ib++;
    } else if (nodesA[ia].node_id < nodesB[ib].node_id) {
      ia = featvec_lower_bound(nodesA, nA_nodes, nodesB[ib].node_id);
    } else {
      ib = featvec_lower_bound(nodesB, nB_nodes, nodesA[ia].node_id);
    }
  }
  if (check_orientation) {
    int i1, i2, i3;
    oo_three_maxima(rh.n, OO_HISTO_LENGTH, &i1, &i2, &i3);
    for (int i = 0; i < OO_HISTO_LENGTH; i++) {
      if (i == i1 || i == i2 || i == i3) continue;
      for (int j = 0; j < rh.n[i]; j++) { vMatches12[rh.v[i][j]] = -1; nmatches--; }
    }
  }
  rh_free(&rh);
  free(vbMatched2);
  return nmatches;
}

/* SearchForInitialization: L/src/ORBmatcher.cc:388-492 */
int oo_search_for_initialization(const oo_keypoint* keys1, const uint8_t* desc1, int n1, const oo_frame* f2,
                                 float* prev_xy, int window, float nnratio, int check_orientation,
                                 int32_t* matches12) {
  int nmatches = 0;
  rot_hist rh; memset(&rh, 0, sizeof(rh));
  int n2 = f2->n;
  int* vMatchedDistance = (int*)malloc(sizeof(int) * (n2 ? n2 : 1));
  int* vnMatches21 = (int*)malloc(sizeof(int) * (n2 ? n2 : 1));
  int32_t* vIndices2 = (int32_t*)malloc(sizeof(int32_t) * (n2 ? n2 : 1));
  for (int i = 0; i < n1; i++) matches12[i] = -1;
  for (int i = 0; i < n2; i++) { vMatchedDistance[i] = INT_MAX; vnMatches21[i] = -1; }
  for (int i1 = 0; i1 < n1; i1++) {
    int level1 = keys1[i1].octave;
    if (level1 > 0) continue;
    int nI = oo_features_in_area(f2, prev_xy[2 * i1], prev_xy[2 * i1 + 1], (float)window, level1, level1, vIndices2);
    if (nI == 0) continue;
    const uint8_t* d1 = desc1 + (size_t)i1 * 32;
    int bestDist = INT_MAX, bestDist2 = INT_MAX, bestIdx2 = -1;
    for (int k = 0; k < nI; k++) {
      int i2 = vIndices2[k];
      int dist = oo_descriptor_distance(d1, f2->desc + (size_t)i2 * 32);
      if (vMatchedDistance[i2] <= dist) continue;
      if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx2 = i2; }
      else if (dist < bestDist2) bestDist2 = dist;
    }
    if (bestDist <= OO_TH_LOW) {
      if (bestDist < (float)bestDist2 * nnratio) {
        if (vnMatches21[bestIdx2] >= 0) { matches12[vnMatches21[bestIdx2]] = -1; nmatches--; }
        matches12[i1] = bestIdx2;
        vnMatches21[bestIdx2] = i1;
        vMatchedDistance[bestIdx2] = bestDist;
        nmatches++;
        if (check_orientation) rh_push(&rh, rot_bin(keys1[i1].angle, f2->keys_un[bestIdx2].angle), i1);
      }
    }
  }
  if (check_orientation) {
    int a, b, c;
    oo_three_maxima(rh.n, OO_HISTO_LENGTH, &a, &b, &c);
    for (int i = 0; i < OO_HISTO_LENGTH; i++) {
      if (i == a || i == b || i == c) continue;
      for (int j = 0; j < rh.n[i]; j++) {
        int idx1 = rh.v[i][j];
        if (matches12[idx1] >= 0) { matches12[idx1] = -1; nmatches--; }
      }
    }
  }
  for (int i1 = 0; i1 < n1; i1++)
    if (matches12[i1] >= 0) {
      prev_xy[2 * i1] = f2->keys_un[matches12[i1]].x;
      prev_xy[2 * i1 + 1] = f2->keys_un[matches12[i1]].y;
    }
  rh_free(&rh);
  free(vMatchedDistance); free(vnMatches21); free(vIndices2);
  return nmatches;
}

/* Frame::ComputeStereoMatches: L/src/Frame.cc:477-646 */
typedef struct { int dist, idx; } dist_idx;
static int dist_idx_cmp(const void* a, const void* b) {
  const dist_idx *x = (const dist_idx*)a, *y = (const dist_idx*)b;
  if (x->dist != y->dist) return x->dist < y->dist ? -1 : 1;
  return x->idx < y->idx ? -1 : (x->idx > y->idx ? 1 : 0);
}

int oo_compute_stereo_matches(const oo_keypoint* keysL, const uint8_t* descL, int N, const oo_keypoint* keysR,
                              const uint8_t* descR, int Nr, const oo_pyramid_view* pyrL,
                              const oo_pyramid_view* pyrR, const float* mvScaleFactors,
                              const float* mvInvScaleFactors, float mbf, float mb, float* mvuRight,
                              float* mvDepth) {
  for (int i = 0; i < N; i++) { mvuRight[i] = -1.0f; mvDepth[i] = -1.0f; }
  const int thOrbDist = (OO_TH_HIGH + OO_TH_LOW) / 2;
  const int nRows = pyrL->h[0];
  /* row table as CSR */
  int* rowCount = (int*)calloc(nRows + 1, sizeof(int));
  for (int iR = 0; iR < Nr; iR++) {
    const float kpY = keysR[iR].y;
    const float r = 2.0f * mvScaleFactors[keysR[iR].octave];
    const int maxr = (int)ceilf(kpY + r), minr = (int)floorf(kpY - r);
    for (int yi = minr; yi <= maxr; yi++)
      if (yi >= 0 && yi < nRows) rowCount[yi + 1]++; /* reference indexes unchecked (:501) */
  }
  for (int i = 0; i < nRows; i++) rowCount[i + 1] += rowCount[i];
  int* rowIdx = (int*)malloc(sizeof(int) * (rowCount[nRows] ? rowCount[nRows] : 1));
  int* fill = (int*)calloc(nRows, sizeof(int));
  for (int iR = 0; iR < Nr; iR++) {
    const float kpY = keysR[iR].y;
    const float r = 2.0f * mvScaleFactors[keysR[iR].octave];
    const int maxr = (int)ceilf(kpY + r), minr = (int)floorf(kpY - r);
    for (int yi = minr; yi <= maxr; yi++)
      if (yi >= 0 && yi < nRows) rowIdx[rowCount[yi] + fill[yi]++] = iR;
  }
  free(fill);
  const float minZ = mb, minD = 0, maxD = mbf / minZ;
  dist_idx* vDistIdx = (dist_idx*)malloc(sizeof(dist_idx) * (N ? N : 1));
  int nDist = 0;
  for (int iL = 0; iL < N; iL++) {
    const oo_keypoint* kpL = &keysL[iL];
    const int levelL = kpL->octave;
    const float vL = kpL->y, uL = kpL->x;
    const int row = (int)vL;
    if (row < 0 || row >= nRows) continue;
    const int c0 = rowCount[row], c1 = rowCount[row + 1];
    if (c0 == c1) continue;
    const float minU = uL - maxD, maxU = uL - minD;
    if (maxU < 0) continue;
    int bestDist = OO_TH_HIGH;
    int bestIdxR = 0;
    const uint8_t* dL = descL + (size_t)iL * 32;
    for (int iC = c0; iC < c1; iC++) {
      const int iR = rowIdx[iC];
      const oo_keypoint* kpR = &keysR[iR];
      if (kpR->octave < levelL - 1 || kpR->octave > levelL + 1) continue;
      const float uR = kpR->x;
      if (uR >= minU && uR <= maxU) {
        const int dist = oo_descriptor_distance(dL, descR + (size_t)iR * 32);
        if (dist < bestDist) { bestDist = dist; bestIdxR = iR; }
      }
    }
    if (bestDist < thOrbDist) {
      const float uR0 = keysR[bestIdxR].x;
      const float scaleFactor = mvInvScaleFactors[kpL->octave];
      const float scaleduL = roundf(kpL->x * scaleFactor);
      const float scaledvL = roundf(kpL->y * scaleFactor);
      const float scaleduR0 = roundf(uR0 * scaleFactor);
      const int w = 5, L = 5;
      const int oct = kpL->octave;
      const uint8_t* imL = pyrL->data[oct];
      const uint8_t* imR = pyrR->data[oct];
      const int sL = pyrL->stride[oct], sR = pyrR->stride[oct];
      int sadDist = INT_MAX; /* the reference's inner `int bestDist` that shadows the Hamming one (:573) */
      int bestincR = 0;
      float vDists[11];
      const float iniu = scaleduR0 + L - w;
      const float endu = scaleduR0 + L + w + 1;
      if (iniu < 0 || endu >= pyrR->w[oct]) continue;
      const int yL0 = (int)(scaledvL - w), xL0 = (int)(scaleduL - w);
      /* the reference takes these windows with rowRange/colRange, which throw when out of range; both this
       * restatement and the device kernel treat that as "no match" */
      if (yL0 < 0 || yL0 + 2 * w >= pyrL->h[oct] || xL0 < 0 || xL0 + 2 * w >= pyrL->w[oct] ||
          (int)scaleduR0 - L - w < 0 || (int)scaleduR0 + L + w >= pyrR->w[oct] || yL0 + 2 * w >= pyrR->h[oct])
        continue;
      const float cL = (float)imL[(size_t)(yL0 + w) * sL + xL0 + w];
      for (int incR = -L; incR <= +L; incR++) {
        const int xR0 = (int)(scaleduR0 + incR - w);
        const float cR = (float)imR[(size_t)(yL0 + w) * sR + xR0 + w];
        double acc = 0; /* cv::norm(NORM_L1) on CV_32F accumulates in double */
        for (int yy = 0; yy < 2 * w + 1; yy++)
          for (int xx = 0; xx < 2 * w + 1; xx++) {
            float a = (float)imL[(size_t)(yL0 + yy) * sL + xL0 + xx] - cL;
            float b = (float)imR[(size_t)(yL0 + yy) * sR + xR0 + xx] - cR;
            acc += fabsf(a - b);
          }
        float dist = (float)acc;
        if (dist < sadDist) { sadDist = (int)dist; bestincR = incR; }
        vDists[L + incR] = dist;
      }
      if (bestincR == -L || bestincR == L) continue;
      const float dist1 = vDists[L + bestincR - 1], dist2 = vDists[L + bestincR], dist3 = vDists[L + bestincR + 1];
      const float deltaR = (dist1 - dist3) / (2.0f * (dist1 + dist3 - 2.0f * dist2));
      if (deltaR < -1 || deltaR > 1) continue;
      float bestuR = mvScaleFactors[kpL->octave] * ((float)scaleduR0 + (float)bestincR + deltaR);
      float disparity = (uL - bestuR);
      if (disparity >= minD && disparity < maxD) {
        if (disparity <= 0) { disparity = 0.01; bestuR = uL - 0.01; }
        mvDepth[iL] = mbf / disparity;
        mvuRight[iL] = bestuR;
        vDistIdx[nDist].dist = sadDist; vDistIdx[nDist].idx = iL; nDist++;
      }
    }
  }
  int nkept = nDist;
  if (nDist > 0) { /* reference reads vDistIdx[size/2] unguarded (:635) */
    qsort(vDistIdx, nDist, sizeof(dist_idx), dist_idx_cmp);
    const float median = (float)vDistIdx[nDist / 2].dist;
    const float thDist = 1.5f * 1.4f * median;
    for (int i = nDist - 1; i >= 0; i--) {
      if (vDistIdx[i].dist < thDist) break;
      mvuRight[vDistIdx[i].idx] = -1;
      mvDepth[vDistIdx[i].idx] = -1;
      nkept--;
    }
  }
  free(vDistIdx); free(rowCount); free(rowIdx);
  return nkept;
}

/* ------------------------------------------------------------------------------------- vocabulary */
struct oo_vocab {
  int k, L, scoring, weighting;
  int n_nodes, n_words;
  int32_t* parent;
  int32_t* child_start; /* CSR of children in ascending id (= file) order */
  int32_t* child_idx;
  uint8_t* desc;
  double* weight;
  int32_t* word_id;
};

oo_vocab* oo_vocab_create(int k, int L, int scoring, int weighting, int n_nodes, const int32_t* parent,
                          const uint8_t* is_leaf, const uint8_t* desc, const double* weight) {
  if (n_nodes < 1) return NULL;
  oo_vocab* v = (oo_vocab*)calloc(1, sizeof(*v));
  v->k = k; v->L = L; v->scoring = scoring; v->weighting = weighting; v->n_nodes = n_nodes;
  v->parent = (int32_t*)calloc(n_nodes, sizeof(int32_t));
  v->child_start = (int32_t*)calloc(n_nodes + 1, sizeof(int32_t));
  v->child_idx = (int32_t*)calloc(n_nodes, sizeof(int32_t));
  v->desc = (uint8_t*)calloc((size_t)n_nodes, 32);
  v->weight = (double*)calloc(n_nodes, sizeof(double));
  v->word_id = (int32_t*)calloc(n_nodes, sizeof(int32_t));
  for (int i = 1; i < n_nodes; i++) {
    if (parent[i] < 0 || parent[i] >= i) { oo_vocab_destroy(v); return NULL; }
    v->parent[i] = parent[i];
    v->child_start[parent[i] + 1]++;
  }
  for (int i = 0; i < n_nodes; i++) v->child_start[i + 1] += v->child_start[i];
  int* fill = (int*)calloc(n_nodes, sizeof(int));
  for (int i = 1; i < n_nodes; i++) v->child_idx[v->child_start[parent[i]] + fill[parent[i]]++] = i;
  free(fill);
  memcpy(v->desc, desc, (size_t)n_nodes * 32);
  memcpy(v->weight, weight, sizeof(double) * n_nodes);
  for (int i = 1; i < n_nodes; i++)
    if (is_leaf[i]) v->word_id[i] = v->n_words++; /* ORBVocabulary.cc:115-120: word ids in file order */
  return v;
}

/* ORBVocabulary::loadFromTextFile, L/src/ORBVocabulary.cc:11-127 */
oo_vocab* oo_vocab_load_text(const char* path) {
  FILE* f = fopen(path, "r");
  if (!f) return NULL;
  int k, L, n1, n2;
  if (fscanf(f, "%d %d %d %d", &k, &L, &n1, &n2) != 4 || k < 0 || k > 20 || L < 1 || L > 10 || n1 < 0 || n1 > 5 ||
      n2 < 0 || n2 > 3) { fclose(f); return NULL; }
  int cap = 1024, n = 1;
  int32_t* parent = (int32_t*)malloc(sizeof(int32_t) * cap);
  uint8_t* leaf = (uint8_t*)malloc(cap);
  uint8_t* desc = (uint8_t*)malloc((size_t)cap * 32);
  double* weight = (double*)malloc(sizeof(double) * cap);
  parent[0] = 0; leaf[0] = 0; weight[0] = 0; memset(desc, 0, 32);
  for (;;) {
    int pid, isleaf;
    if (fscanf(f, "%d %d", &pid, &isleaf) != 2) break;
    if (n == cap) {
      cap *= 2;
      parent = (int32_t*)realloc(parent, sizeof(int32_t) * cap); leaf = (uint8_t*)realloc(leaf, cap);
      desc = (uint8_t*)realloc(desc, (size_t)cap * 32); weight = (double*)realloc(weight, sizeof(double) * cap);
    }
    parent[n] = pid; leaf[n] = isleaf > 0;
    for (int i = 0; i < 32; i++) { int b; if (fscanf(f, "%d", &b) != 1) b = 0; desc[(size_t)n * 32 + i] = (uint8_t)b; }
    if (fscanf(f, "%lf", &weight[n]) != 1) weight[n] = 0;
    n++;
  }
  fclose(f);
  oo_vocab* v = oo_vocab_create(k, L, n1, n2, n, parent, leaf, desc, weight);
  free(parent); free(leaf); free(desc); free(weight);
  return v;
}

/* ORBVocabulary::loadFromBinaryFile: L/src/ORBVocabulary.cc:152-213.  m_nodes is sized nb_nodes + 1 (:167) and the
 * `while (!f.eof())` loop (:185) body runs nb_nodes times for the nb_nodes - 1 records saveToBinaryFile wrote (:231): the
 * last pass re-uses the unchanged buffer, i.e. node nb_nodes is a copy of node nb_nodes - 1. */
oo_vocab* oo_vocab_load_binary(const char* path) {
  FILE* f = fopen(path, "rb");
  if (!f) return NULL;
  unsigned int nb_nodes, size_node;
  int m_k, m_L, m_scoring, m_weighting;
  if (fread(&nb_nodes, 4, 1, f) != 1 || fread(&size_node, 4, 1, f) != 1 || fread(&m_k, 4, 1, f) != 1 ||
      fread(&m_L, 4, 1, f) != 1 || fread(&m_scoring, 4, 1, f) != 1 || fread(&m_weighting, 4, 1, f) != 1 ||
      size_node < 41 || size_node > 4096) { fclose(f); return NULL; }
  const int cap = (int)nb_nodes + 1;
  int32_t* parent = (int32_t*)calloc(cap, sizeof(int32_t));
  uint8_t* leaf = (uint8_t*)calloc(cap, 1);
  uint8_t* desc = (uint8_t*)calloc((size_t)cap, 32);
  double* weight = (double*)calloc(cap, sizeof(double));
  char* buf = (char*)calloc(size_node, 1);
  int nid = 1, at_eof = 0;
  while (!at_eof && nid < cap) {
    if (fread(buf, 1, size_node, f) != size_node) at_eof = 1;   /* f.read fails, buf keeps the previous record */
    if (at_eof && nid == 1) break;
    parent[nid] = *(const int*)buf;
    memcpy(desc + (size_t)nid * 32, buf + 4, 32);
    weight[nid] = (double)*(const float*)(buf + 4 + 32);
    leaf[nid] = buf[8 + 32] != 0;
    nid++;
  }
  fclose(f);
  oo_vocab* v = oo_vocab_create(m_k, m_L, m_scoring, m_weighting, nid, parent, leaf, desc, weight);
  free(parent); free(leaf); free(desc); free(weight); free(buf);
  return v;
}

void oo_vocab_destroy(oo_vocab* v) {
  if (!v) return;
  free(v->parent); free(v->child_start); free(v->child_idx); free(v->desc); free(v->weight); free(v->word_id);
  free(v);
}
int oo_vocab_nodes(const oo_vocab* v) { return v->n_nodes; }
int oo_vocab_words(const oo_vocab* v) { return v->n_words; }

/* FORB::distance, src/FORB.cpp:77-100 (64-bit SWAR popcount; value == Hamming distance) */
static double forb_distance(const uint8_t* a, const uint8_t* b) {
  uint64_t ret = 0;
  for (int i = 0; i < 4; i++) {
    uint64_t pa, pb, x;
    memcpy(&pa, a + 8 * i, 8); memcpy(&pb, b + 8 * i, 8);
    x = pa ^ pb;
    x = x - ((x >> 1) & (uint64_t)~(uint64_t)0 / 3);
    x = (x & (uint64_t)~(uint64_t)0 / 15 * 3) + ((x >> 2) & (uint64_t)~(uint64_t)0 / 15 * 3);
    x = (x + (x >> 4)) & (uint64_t)~(uint64_t)0 / 255 * 15;
    ret += (uint64_t)(x * ((uint64_t)~(uint64_t)0 / 255)) >> (sizeof(uint64_t) - 1) * 8;
  }
  return (double)ret;
}

/* transform(feature, word_id, weight, nid, levelsup): TemplatedVocabulary.h:1216-1257 */
void oo_vocab_transform_feature(const oo_vocab* v, const uint8_t* d, int levelsup, int32_t* word_id, int32_t* node_id,
                                double* weight) {
  const int nid_level = v->L - levelsup;
  int32_t nid = 0; /* root when nid_level <= 0 */
  int final_id = 0, current_level = 0;
  if (v->child_start[1] == v->child_start[0]) { *word_id = v->word_id[0]; *weight = v->weight[0]; *node_id = 0; return; }
  do {
    ++current_level;
    const int c0 = v->child_start[final_id], c1 = v->child_start[final_id + 1];
    final_id = v->child_idx[c0];
    double best_d = forb_distance(d, v->desc + (size_t)final_id * 32);
    for (int c = c0 + 1; c < c1; c++) {
      const int id = v->child_idx[c];
      const double dd = forb_distance(d, v->desc + (size_t)id * 32);
      if (dd < best_d) { best_d = dd; final_id = id; }
    }
    if (current_level == nid_level) nid = final_id;
  } while (v->child_start[final_id + 1] != v->child_start[final_id]); /* !isLeaf() == has children */
  *word_id = v->word_id[final_id];
  *weight = v->weight[final_id];
  *node_id = nid;
}

typedef struct { int32_t key, idx; double w; } bow_tmp;
static int bow_tmp_cmp(const void* a, const void* b) { /* stable by (key, idx) */
  const bow_tmp *x = (const bow_tmp*)a, *y = (const bow_tmp*)b;
  if (x->key != y->key) return x->key < y->key ? -1 : 1;
  return x->idx < y->idx ? -1 : (x->idx > y->idx ? 1 : 0);
}

/* transform(features, BowVector, FeatureVector, levelsup): TemplatedVocabulary.h:1125-1192 */
int oo_vocab_transform(const oo_vocab* v, const uint8_t* desc, int n, int levelsup, int32_t* bow_ids, double* bow_vals,
                       int* n_bow, oo_featvec_node* fv_nodes, int32_t* fv_idx, int* n_fv) {
  *n_bow = 0; *n_fv = 0;
  if (v->n_nodes <= 1 || n <= 0) return 0; /* empty() */
  bow_tmp* wt = (bow_tmp*)malloc(sizeof(bow_tmp) * n);
  bow_tmp* nt = (bow_tmp*)malloc(sizeof(bow_tmp) * n);
  int m = 0;
  for (int i = 0; i < n; i++) {
    int32_t wid, nid; double w;
    oo_vocab_transform_feature(v, desc + (size_t)i * 32, levelsup, &wid, &nid, &w);
    if (w > 0) { wt[m].key = wid; wt[m].idx = i; wt[m].w = w; nt[m].key = nid; nt[m].idx = i; nt[m].w = 0; m++; }
  }
  qsort(wt, m, sizeof(bow_tmp), bow_tmp_cmp);
  qsort(nt, m, sizeof(bow_tmp), bow_tmp_cmp);
  const int tf = (v->weighting == 0 /*TF_IDF*/ || v->weighting == 1 /*TF*/);
  int nb = 0;
  for (int i = 0; i < m;) { /* std::map<WordId, double>: addWeight sums in feature order, addIfNotExist keeps the first */
    int j = i; double acc = wt[i].w;
    for (j = i + 1; j < m && wt[j].key == wt[i].key; j++) if (tf) acc += wt[j].w;
    bow_ids[nb] = wt[i].key; bow_vals[nb] = acc; nb++;
    i = j;
  }
  const int must = v->scoring != 5; /* every scoring but DOT_PRODUCT normalises (ScoringObject.h:72-88) */
  const int l2 = v->scoring == 1;   /* L2_NORM -> L2, all others L1 */
  if (tf && nb > 0 && !must) { const double nd = (double)nb; for (int i = 0; i < nb; i++) bow_vals[i] /= nd; }
  if (must) { /* BowVector::normalize */
    double norm = 0.0;
    if (!l2) for (int i = 0; i < nb; i++) norm += fabs(bow_vals[i]);
    else { for (int i = 0; i < nb; i++) norm += bow_vals[i] * bow_vals[i]; norm = sqrt(norm); }
    if (norm > 0.0) for (int i = 0; i < nb; i++) bow_vals[i] /= norm;
  }
  int nf = 0, pos = 0;
  for (int i = 0; i < m;) { /* FeatureVector: ascending node id, features in ascending index */
    int j = i;
    fv_nodes[nf].node_id = nt[i].key; fv_nodes[nf].start = pos;
    for (; j < m && nt[j].key == nt[i].key; j++) fv_idx[pos++] = nt[j].idx;
    fv_nodes[nf].count = pos - fv_nodes[nf].start; nf++;
    i = j;
  }
  free(wt); free(nt);
  *n_bow = nb; *n_fv = nf;
  return m;
}
