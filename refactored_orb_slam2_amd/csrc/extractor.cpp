// extractor.cpp -- host side of liborbfe's ORBextractor path: parameter tables, geometry plan, HBM work
// space, kernel sequencing, and the C ABI of include/orbfe.h.  No CPU fallback exists: every compute
// entry point needs a HIP device.
//
// Reference behaviour restated here (L/ = Source/Libraries/ORB_SLAM2/):
//   constructor tables          L/src/ORBextractor.cc:407-464
//   level sizes                 L/src/ORBextractor.cc:1043-1045
//   FAST cell geometry          L/src/ORBextractor.cc:740-771
//   DistributeOctTree set-up    L/src/ORBextractor.cc:535-537
//   resize coefficient set-up   cv::resize (OpenCV 4.5 imgproc/src/resize.cpp), see SURVEY.md §8(c)-P2
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <mutex>
#include <vector>

#include "orbfe_internal.h"

void orbfe_launch_copy0(const uint8_t* src, int sstride, size_t simg, uint8_t* dst, int dpitch, size_t dimg, int w,
                        int h, int n_images, int tiled, hipStream_t s);
int orbfe_set_octree_lds(size_t lds_bytes);
int orbfe_upload_pattern_floats();

// ------------------------------------------------------------------------------------------------ errors
static thread_local char g_err[512] = "";
void orbfe_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* orbfe_last_error(void) { return g_err; }

#define HIPCHK(expr)                                                                              \
  do {                                                                                            \
    hipError_t _e = (expr);                                                                       \
    if (_e != hipSuccess) {                                                                       \
      orbfe_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return ORBFE_ERR_HIP;                                                                       \
    }                                                                                             \
  } while (0)

extern "C" int orbfe_device_count(int* count) {
  if (!count) return ORBFE_ERR_INVALID;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    *count = 0;
    orbfe_set_error("hipGetDeviceCount: %s", hipGetErrorString(e));
    return ORBFE_ERR_NO_DEVICE;
  }
  *count = n;
  return ORBFE_OK;
}

// Device selection and plain device memory for hosts that do not link the HIP runtime themselves (a C or C++ caller of the
// batched mode keeps its records in HBM for the gather): thin wrappers, synchronous.
extern "C" int orbfe_set_device(int device) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n < 1) { orbfe_set_error("no HIP device available"); return ORBFE_ERR_NO_DEVICE; }
  if (device < 0 || device >= n) return ORBFE_ERR_INVALID;
  HIPCHK(hipSetDevice(device));
  return ORBFE_OK;
}
extern "C" int orbfe_device_malloc(size_t bytes, void** out) {
  if (!out) return ORBFE_ERR_INVALID;
  *out = nullptr;
  HIPCHK(hipMalloc(out, bytes ? bytes : 256));
  return ORBFE_OK;
}
extern "C" int orbfe_device_free(void* p) {
  if (p) HIPCHK(hipFree(p));
  return ORBFE_OK;
}
extern "C" int orbfe_device_upload(void* d_dst, const void* h_src, size_t bytes) {
  if ((!d_dst || !h_src) && bytes) return ORBFE_ERR_INVALID;
  if (bytes) HIPCHK(hipMemcpy(d_dst, h_src, bytes, hipMemcpyHostToDevice));
  return ORBFE_OK;
}
extern "C" int orbfe_device_download(void* h_dst, const void* d_src, size_t bytes) {
  if ((!h_dst || !d_src) && bytes) return ORBFE_ERR_INVALID;
  if (bytes) HIPCHK(hipMemcpy(h_dst, d_src, bytes, hipMemcpyDeviceToHost));
  return ORBFE_OK;
}

// ------------------------------------------------------------------------------------------------ handle
struct LevelGeom {
  int w, h, pitch;
  size_t plane;  // bytes of one image's plane (pitch*h rounded to 256)
  size_t off;    // offset of this level inside the per-level-major pyramid buffer
};

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
};

struct orbfe_extractor {
  orbfe_params prm{};
  int device = 0;
  hipStream_t stream = nullptr;
  // constructor tables
  float scale[ORBFE_MAX_LEVELS]{}, inv_scale[ORBFE_MAX_LEVELS]{}, sigma2[ORBFE_MAX_LEVELS]{}, inv_sigma2[ORBFE_MAX_LEVELS]{};
  int feat_per_level[ORBFE_MAX_LEVELS]{};
  // plan (depends on image geometry)
  int plan_w = 0, plan_h = 0;
  // latency path (one or two host images): the 13 dependent launches of a call, captured once per geometry as a hipGraph
  struct LaunchGraph {
    hipGraphExec_t exec = nullptr;
    int w = 0, h = 0, cap = 0, warm = 0;
    unsigned long long buffers = 0;   // buffer_signature() at capture time
  } graphs[2];
  // level 0 of the last device batch when the caller's images could be used in place (16-byte aligned rows): no pitched copy
  const uint8_t* ext0 = nullptr;
  int ext0_pitch = 0;
  size_t ext0_plane = 0;
  LevelGeom lg[ORBFE_MAX_LEVELS]{};
  std::vector<CellDesc> cells;
  std::vector<FastGroup> groups;   // runs of adjacent cells, one workgroup each
  int fc_rows = 0, fc_span = 0, fc_sc = 0, fc_bits = 0;   // wave-per-cell FAST: largest cell ROI rows, (x0 & 15) + 1 + cols, score plane bytes
  std::vector<BlurTile> tiles;
  int tile_first[ORBFE_MAX_LEVELS]{}, tile_count[ORBFE_MAX_LEVELS]{};   // a level's tiles are consecutive in `tiles`
  bool fuse_ok[ORBFE_MAX_LEVELS]{};    // [l]: the step l -> l + 1 can run inside level l's blur tiles (blur_level_kernel<true>)
  std::vector<BlurStrip> strips;       // gauss_blur7_mfma: one wave per (level, 48-column chunk)
  std::vector<uint8_t> blur_tab;       // its band matrices as MFMA operands (1 KB each)
  uint32_t blur_b_off[ORBFE_MAX_LEVELS]{}, blur_t_off = 0;
  bool blur_mfma = false;              // every coefficient fits int8, no level under eight pixels wide or high
  int blur_kind = 0;                   // orbfe_debug_blur_kernel: 0 = blur_level_kernel, level l's blur and the resize l -> l + 1 from one staged
                                       // window, one launch per level; 1 = resize chain + gauss_blur7_mfma_kernel; 2 = resize chain + one
                                       // blur_level_kernel<false> launch over all levels (the pipeline of rounds 1-4)
  OctLevel oct[ORBFE_MAX_LEVELS]{};
  int total_cells = 0;
  size_t slots_per_image = 0, gkeys_per_image = 0;
  int kp_per_image = 0;
  int max_nodes = 0, lds_keys = 0;
  size_t oct_lds = 0;
  // device tables
  DevBuf d_strips, d_blur_tab;
  DevBuf d_cells, d_groups, d_groups1, d_tiles, d_xt[ORBFE_MAX_LEVELS], d_yt[ORBFE_MAX_LEVELS];   // d_groups1: one cell per run (small batches)
  int resize_mode[ORBFE_MAX_LEVELS]{};  // 0: direct gathers; 1: every 256x16 destination tile's source window fits the LDS stage;
                                         // 2: and every aligned group of four destination pixels reads at most 8 adjacent source bytes;
                                         // 3: and every 256-pixel tile's window fits 21 chunks of 16 bytes (scale factors up to ~1.25)
  // work space for `cap_images`
  int cap_images = 0;
  DevBuf d_pyr, d_blur, d_cell_cnt, d_cell_off, d_slots, d_gkeys, d_lvl_kp, d_lvl_n, d_err;
  // host-API staging: one pinned buffer each way + device mirrors (single H2D, single D2H, one sync per call)
  DevBuf d_out_kps, d_out_desc, d_out_n, d_in_stage;
  void* h_in = nullptr;  size_t h_in_bytes = 0;    // pinned: caller images, packed
  void* h_out = nullptr; size_t h_out_bytes = 0;   // pinned: [n_out[B] | err | kps | desc] / pyramid planes
  int out_cap = 0;
  int32_t* h_err = nullptr;   // pinned: where the device error word is read to (a pageable destination makes the copy wait for every
                              // stream of the device, also the caller's: measured on the batched pipeline, 52 k -> 33 k frames/s)
  int last_images = 0;
  unsigned last_tiled = 0;    // PyrView::tiled of the pyramid the last call left in d_pyr (the download entry points un-tile by it)
  // profiling
  bool profile = false;
  unsigned profile_mask = ~0u;  // bit s: stage s is timed
  float stage_ms[ORBFE_STAGE_COUNT]{};
  int stage_launches[ORBFE_STAGE_COUNT]{};
  std::vector<hipEvent_t> ev_pool;
  struct EvPair { int stage; hipEvent_t a, b; };
  std::vector<EvPair> ev_pending;
  std::mutex mu;
};

static int dev_alloc(DevBuf& b, size_t bytes) {
  if (bytes <= b.bytes && b.p) return ORBFE_OK;
  if (b.p) HIPCHK(hipFree(b.p));
  b.p = nullptr;
  b.bytes = 0;
  if (bytes == 0) bytes = 256;
  HIPCHK(hipMalloc(&b.p, bytes));
  b.bytes = bytes;
  return ORBFE_OK;
}
static int pinned_alloc(void*& p, size_t& have, size_t bytes) {
  if (p && bytes <= have) return ORBFE_OK;
  if (p) HIPCHK(hipHostFree(p));
  p = nullptr;
  have = 0;
  HIPCHK(hipHostMalloc(&p, bytes ? bytes : 256, hipHostMallocDefault));
  have = bytes ? bytes : 256;
  return ORBFE_OK;
}
static void dev_free(DevBuf& b) {
  if (b.p) (void)hipFree(b.p);
  b.p = nullptr;
  b.bytes = 0;
}

static inline int cv_round_f(float v) { return (int)lrintf(v); }
static inline int16_t sat_short(float v) {
  int i = cv_round_f(v);
  return (int16_t)std::min(32767, std::max(-32768, i));
}

// cv::resize INTER_LINEAR coefficient set-up for one axis.  clamp_edges: the x axis forces fx=0 at the
// borders (sx<0, sx>=ssize-1); the y axis only clamps the row index when rows are read.
static void build_taps(int s, int d, bool x_axis, std::vector<ResizeTap>& out) {
  out.resize(d);
  const double scale = 1.0 / ((double)d / (double)s);
  for (int i = 0; i < d; i++) {
    float f = (float)((i + 0.5) * scale - 0.5);
    int si = (int)floorf(f);
    f -= si;
    if (x_axis) {
      if (si < 0) { f = 0; si = 0; }
      if (si >= s - 1) { f = 0; si = s - 1; }
    }
    ResizeTap t;
    int s0 = si, s1 = si + 1;
    s0 = std::min(std::max(s0, 0), s - 1);
    s1 = std::min(std::max(s1, 0), s - 1);
    t.s0 = (int16_t)s0;
    t.s1 = (int16_t)s1;
    t.c0 = sat_short((1.f - f) * 2048);
    t.c1 = sat_short(f * 2048);
    out[i] = t;
  }
}

static int upload(DevBuf& b, const void* src, size_t bytes, hipStream_t s) {
  int rc = dev_alloc(b, bytes);
  if (rc) return rc;
  if (bytes) HIPCHK(hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, s));
  return ORBFE_OK;
}

// cv::borderInterpolate(p, len, BORDER_REFLECT_101)
static int reflect101_host(int p, int len) {
  if (len == 1) return 0;
  while (p < 0 || p >= len) p = p < 0 ? -p : 2 * (len - 1) - p;
  return p;
}
// Coefficient of input sample `in` in output sample `out` of the 7-tap pass over a line of `len` samples: the taps whose
// REFLECT_101 source is `in` added up (L/src/ORBextractor.cc:1019, SURVEY P3: [18,34,48,56,48,34,18], folded taps <= 96 < 128)
static int blur_coef(int in, int out, int len) {
  static const int taps[7] = {18, 34, 48, 56, 48, 34, 18};
  if (out < 0 || out >= len || in < 0 || in >= len) return 0;
  int c = 0;
  for (int t = -3; t <= 3; t++)
    if (reflect101_host(out + t, len) == in) c += taps[t + 3];
  return c;   // two taps fold onto one sample at an edge (<= 96); on lines of two or three samples more do, and the sum leaves int8
}
// Band matrices of gauss_blur7_mfma_kernel (extract_kernels.hip) as v_mfma_i32_16x16x64_i8 operands: lane (q = lane >> 4,
// n = lane & 15) holds 16 bytes.  Horizontal, strip chunk c, column group g: byte j <-> input column 48c - 16 + 16q + j, output
// column 48c - 12 + 16g + n (strip c = output columns [48c - 12, 48c + 36): every tap inside the 64-byte segment); the level's
// left and right REFLECT_101 borders are folded into the first and last chunks' matrices.  Vertical, row group b of a window:
// byte j = 4ww + i <-> window row 16ww + 4q + i (the order the kernel's packed first-pass results come in), output window row
// 16 + 16b + n; plain Toeplitz -- the kernel fetches rows outside the level from their reflection.
// Returns false when a coefficient does not fit int8 or the kernel's single row reflection would not do (a level fewer than
// eight pixels wide or high): the plan then keeps the LDS kernel (gauss_blur7_kernel) for its blur.
static bool build_blur_tables(orbfe_extractor* e) {
  static const int taps[7] = {18, 34, 48, 56, 48, 34, 18};
  e->strips.clear();
  e->blur_tab.clear();
  bool fits = true;
  auto put = [&fits](int8_t& dst, int c) { fits = fits && c <= 127; dst = (int8_t)c; };
  e->blur_t_off = 0;
  e->blur_tab.resize(2 * 1024);
  {
    int8_t* tt = reinterpret_cast<int8_t*>(e->blur_tab.data());
    for (int b = 0; b < 2; b++)
      for (int lane = 0; lane < 64; lane++)
        for (int j = 0; j < 16; j++) {
          const int d = (16 * (j >> 2) + 4 * (lane >> 4) + (j & 3)) - (16 + 16 * b + (lane & 15));
          tt[((size_t)b * 64 + lane) * 16 + j] = (int8_t)(d >= -3 && d <= 3 ? taps[d + 3] : 0);
        }
  }
  for (int l = 0; l < e->prm.n_levels; l++) {
    const LevelGeom& g = e->lg[l];
    fits = fits && g.w >= 8 && g.h >= 8;
    const int nch = (g.w + 12 + ORBFE_BLUR_CHUNK - 1) / ORBFE_BLUR_CHUNK;
    for (int c = 0; c < nch; c++) e->strips.push_back(BlurStrip{(int16_t)l, (int16_t)c});
    e->blur_b_off[l] = (uint32_t)e->blur_tab.size();
    e->blur_tab.resize(e->blur_tab.size() + (size_t)nch * 3 * 1024);
    int8_t* bt = reinterpret_cast<int8_t*>(e->blur_tab.data() + e->blur_b_off[l]);
    for (int c = 0; c < nch; c++)
      for (int gg = 0; gg < 3; gg++)
        for (int lane = 0; lane < 64; lane++)
          for (int j = 0; j < 16; j++)
            put(bt[(((size_t)c * 3 + gg) * 64 + lane) * 16 + j],
                blur_coef(ORBFE_BLUR_CHUNK * c - 16 + 16 * (lane >> 4) + j, ORBFE_BLUR_CHUNK * c - 12 + 16 * gg + (lane & 15), g.w));
  }
  return fits;
}

// Builds everything that depends on the image size.

static int build_plan(orbfe_extractor* e, int w, int h) {
  if (e->plan_w == w && e->plan_h == h) return ORBFE_OK;
  if (w < 1 || h < 1 || w > 4095 || h > 4095) {
    orbfe_set_error("unsupported image size %dx%d (1..4095)", w, h);
    return ORBFE_ERR_INVALID;
  }
  const int nl = e->prm.n_levels;
  e->plan_w = e->plan_h = 0;   // the tables below are rewritten in place: a plan that fails half-way must not pass for the old one
  size_t off = 0;
  for (int l = 0; l < nl; l++) {
    LevelGeom& g = e->lg[l];
    g.w = cv_round_f((float)w * e->inv_scale[l]);  // L/src/ORBextractor.cc:1044-1045
    g.h = cv_round_f((float)h * e->inv_scale[l]);
    if (g.w < 1 || g.h < 1) {
      orbfe_set_error("pyramid level %d of a %dx%d image is empty", l, w, h);
      return ORBFE_ERR_INVALID;
    }
    g.pitch = (g.w + 63) & ~63;
    // rows padded to 8: the blurred planes are stored in 16 x 8 tiles; 256 spare bytes behind them: where gauss_blur7_mfma's
    // lanes outside the level put their dword (extract_kernels.hip)
    g.plane = ((size_t)g.pitch * ((g.h + 7) & ~7) + 256 + 255) & ~(size_t)255;
    g.off = off;  // per-image offsets; the buffer is level-major: level block = plane * cap_images
    off += g.plane;
  }
  // FAST cells and octree levels
  e->cells.clear();
  e->groups.clear();
  e->fc_rows = e->fc_span = e->fc_sc = e->fc_bits = 0;
  e->tiles.clear();
  size_t slot_off = 0, key_off = 0;
  int kp_off = 0, maxM = 0;
  for (int l = 0; l < nl; l++) {
    const LevelGeom& g = e->lg[l];
    OctLevel& o = e->oct[l];
    memset(&o, 0, sizeof(o));
    o.cell_begin = (int)e->cells.size();
    o.N = e->feat_per_level[l];
    const int minB = ORBFE_EDGE;
    const int maxBX = g.w - ORBFE_EDGE, maxBY = g.h - ORBFE_EDGE;
    const float width = (float)(maxBX - minB), height = (float)(maxBY - minB);
    const int nCols = (int)(width / 30.f), nRows = (int)(height / 30.f);
    o.n_ini = 1;
    o.hX = 1.f;
    o.width = std::max(maxBX - minB, 0);
    o.height = std::max(maxBY - minB, 0);
    size_t level_slots = 0;
    if (nCols >= 1 && nRows >= 1) {  // the reference would divide by zero otherwise (:753-754)
      const int wCell = (int)ceilf(width / nCols), hCell = (int)ceilf(height / nRows);
      for (int i = 0; i < nRows; i++) {
        const float iniY = (float)(minB + i * hCell);
        float maxY = iniY + hCell + 6;
        if (iniY >= maxBY - 3) continue;
        if (maxY > maxBY) maxY = (float)maxBY;
        const size_t row_first = e->cells.size();
        for (int j = 0; j < nCols; j++) {
          const float iniX = (float)(minB + j * wCell);
          float maxX = iniX + wCell + 6;
          if (iniX >= maxBX - 6) continue;
          if (maxX > maxBX) maxX = (float)maxBX;
          CellDesc c;
          c.level = (int16_t)l;
          c.x0 = (int16_t)iniX;
          c.y0 = (int16_t)iniY;
          c.cols = (int16_t)((int)maxX - (int)iniX);
          c.rows = (int16_t)((int)maxY - (int)iniY);
          c.shift_x = (int16_t)(j * wCell);
          c.shift_y = (int16_t)(i * hCell);
          if (c.cols > ORBFE_CELL_MAX || c.rows > ORBFE_CELL_MAX) {
            orbfe_set_error("FAST cell %dx%d exceeds the staged maximum %d", c.cols, c.rows, ORBFE_CELL_MAX);
            return ORBFE_ERR_INVALID;
          }
          const int tw = std::max(c.cols - 6, 0), th = std::max(c.rows - 6, 0);
          const int cap = std::max(((tw + 1) / 2) * ((th + 1) / 2), 1);  // NMS survivors are never 8-adjacent
          c.slot_cap = (int16_t)cap;
          c.ppr = c.ri = c.n_it = c.pad = 0; c.inv_ppr = 0.0f;   // filled below, once it is known which kernel variant stages the cells
          e->fc_rows = std::max(e->fc_rows, (int)c.rows);
          e->fc_span = std::max(e->fc_span, (c.x0 & 15) + 1 + (int)c.cols);
          e->fc_sc = std::max(e->fc_sc, (th + 2) * (tw + 2));
          e->fc_bits = std::max(e->fc_bits, th * (tw > 32 ? 64 : 32));   // bitmap: one or two words per tested row
          c.slot_off = (uint32_t)slot_off;
          slot_off += cap;
          level_slots += cap;
          e->cells.push_back(c);
        }
        // split this cell row into runs of <= G cells of near-equal length (cells of a row are contiguous: only
        // trailing ones are ever skipped)
        const int n_row = (int)(e->cells.size() - row_first);
        const int G = std::max(1, std::min(ORBFE_FG_MAX, (ORBFE_FG_MAX_WIDTH - 6) / wCell));
        const int n_grp = (n_row + G - 1) / G;
        int done = 0;
        for (int k = 0; k < n_grp; k++) {
          const int cnt = (n_row - done + (n_grp - k) - 1) / (n_grp - k);
          const CellDesc& a = e->cells[row_first + done];
          const CellDesc& b = e->cells[row_first + done + cnt - 1];
          FastGroup fg;
          fg.first_cell = (int32_t)(row_first + done);
          fg.n_cells = (int16_t)cnt; fg.level = (int16_t)l;
          fg.x0 = a.x0; fg.y0 = a.y0; fg.width = (int16_t)(b.x0 + b.cols - a.x0); fg.rows = a.rows;
          fg.wcell = (int16_t)wCell; fg.pad = 0;
          e->groups.push_back(fg);
          done += cnt;
        }
      }
      const int nIni = (int)roundf(width / height);  // :535
      if (nIni < 1 || nIni > ORBFE_MAX_INI) {
        orbfe_set_error("level %d aspect ratio gives nIni=%d (supported 1..%d)", l, nIni, ORBFE_MAX_INI);
        return ORBFE_ERR_INVALID;
      }
      o.n_ini = nIni;
      o.hX = width / nIni;
    }
    o.n_cells = (int)e->cells.size() - o.cell_begin;
    o.kp_cap = std::max(o.N + 3, 4 * o.n_ini) + 1;
    o.kp_off = kp_off;
    kp_off += o.kp_cap;
    maxM = std::max(maxM, o.kp_cap);
    o.key_off = key_off;
    o.key_cap = (int)std::min<size_t>(level_slots, 0xFFFFFF);
    key_off += level_slots;
  }
  // the quick-test loop's lane layout per cell (the tile is shifted by a column where that makes the first tested column even)
  for (CellDesc& c : e->cells) {
    const int tw = std::max(c.cols - 6, 0), th = std::max(c.rows - 6, 0);
    const int ppr = std::max((tw + 1) >> 1, 1), ri = std::max(64 / ppr, 1);
    c.ppr = (int16_t)ppr; c.ri = (int16_t)ri; c.n_it = (int16_t)((th + ri - 1) / ri);
    c.inv_ppr = 1.0f / (float)ppr;
  }
  // resize coefficient tables of every step l - 1 -> l, and which kernel can run it
  std::vector<ResizeTap> hxt[ORBFE_MAX_LEVELS], hyt[ORBFE_MAX_LEVELS];
  for (int l = 1; l < nl; l++) {
    std::vector<ResizeTap>&xt = hxt[l], &yt = hyt[l];
    build_taps(e->lg[l - 1].w, e->lg[l].w, true, xt);
    build_taps(e->lg[l - 1].h, e->lg[l].h, false, yt);
    // does the source window of every 256 x 16 destination tile fit the staged 34 rows x 560 bytes?
    bool ok = true;
    for (int x0 = 0; x0 < (int)xt.size() && ok; x0 += 256) {
      const int xl = std::min<int>(x0 + 255, (int)xt.size() - 1);
      const int sxa = xt[x0].s0 & ~15;
      if ((((xt[xl].s1 - sxa) >> 4) + 1) * 16 > 560) ok = false;
    }
    for (int y0 = 0; y0 < (int)yt.size() && ok; y0 += 16) {  // 256 x 16 tiles of the 4-rows-per-thread kernel: 34 rows
      const int yl = std::min<int>(y0 + 15, (int)yt.size() - 1);
      if (yt[yl].s1 - yt[y0].s0 + 1 > 34) ok = false;
    }
    bool win8 = ok;
    for (int y0 = 0; y0 < (int)yt.size() && win8; y0 += 32) {  // 256 x 32 tiles of pyr_resize_dot_kernel<32, 42>
      const int yl = std::min<int>(y0 + 31, (int)yt.size() - 1);
      if (yt[yl].s1 - yt[y0].s0 + 1 > 42) win8 = false;
    }
    for (int x4 = 0; x4 < (int)xt.size() && win8; x4 += 4) {
      const int xe = std::min<int>(x4 + 3, (int)xt.size() - 1);
      if (xt[xe].s0 + 1 - xt[x4].s0 > 7) win8 = false;
      for (int i = x4; i <= xe; i++)
        if (xt[i].c1 != 0 && xt[i].s1 != xt[i].s0 + 1) win8 = false;
    }
    bool narrow = win8;   // every 256-pixel tile's window within 21 chunks of 16 bytes: the kernel's 336-byte LDS rows (8 workgroups per CU)
    for (int x0 = 0; x0 < (int)xt.size() && narrow; x0 += 256) {
      const int xl = std::min<int>(x0 + 255, (int)xt.size() - 1);
      if (((xt[xl].s1 - (xt[x0].s0 & ~15)) >> 4) + 1 > 21) narrow = false;
    }
    e->resize_mode[l] = narrow ? 3 : win8 ? 2 : ok ? 1 : 0;
  }
  // blur tiles, and the part of level l + 1 each tile of level l owns when the resize step is fused into the blur: destination
  // dword j (pixels 4j .. 4j + 3) belongs to the tile column that holds the first source column of pixel 4j, destination row y
  // to the tile row that holds its upper source row.  Source indices ascend with the destination index, so a tile owns a
  // contiguous block; the step is fused only if every block fits the kernel's thread layout (16 dwords x 48 rows) and every tap
  // of an owned dword lies inside the staged window (columns ox - 4 .. ox + 71, rows oy - 3 .. oy + 58).
  for (int l = 0; l < nl; l++) {
    const LevelGeom& g = e->lg[l];
    const int tiles_x = (g.w + 63) / 64, tiles_y = (g.h + ORBFE_BLUR_TILE_H - 1) / ORBFE_BLUR_TILE_H;
    std::vector<int> jx(tiles_x + 1, 0), ry(tiles_y + 1, 0);
    bool fuse = l + 1 < nl && e->resize_mode[l + 1] >= 2;
    if (fuse) {
      const std::vector<ResizeTap>&xt = hxt[l + 1], &yt = hyt[l + 1];
      const int dw = (int)xt.size(), dh = (int)yt.size(), ndw = (dw + 3) / 4;
      for (int tx = 0, j = 0; tx <= tiles_x; tx++) {
        while (j < ndw && xt[4 * j].s0 < 64 * tx) j++;
        jx[tx] = tx == tiles_x ? ndw : j;
      }
      for (int ty = 0, y = 0; ty <= tiles_y; ty++) {
        while (y < dh && yt[y].s0 < ORBFE_BLUR_TILE_H * ty) y++;
        ry[ty] = ty == tiles_y ? dh : y;
      }
      for (int tx = 0; tx < tiles_x && fuse; tx++) {
        if (jx[tx + 1] - jx[tx] > ORBFE_FUSE_DWORDS) fuse = false;
        for (int j = jx[tx]; j < jx[tx + 1] && fuse; j++) {
          const int xe = std::min(4 * j + 3, dw - 1);
          if (xt[4 * j].s0 < 64 * tx || xt[xe].s0 + 1 > 64 * tx + 71) fuse = false;
        }
      }
      for (int ty = 0; ty < tiles_y && fuse; ty++) {
        if (ry[ty + 1] - ry[ty] > 16 * ORBFE_FUSE_ROWS) fuse = false;
        for (int y = ry[ty]; y < ry[ty + 1] && fuse; y++)
          if (yt[y].s0 < ORBFE_BLUR_TILE_H * ty || yt[y].s1 > ORBFE_BLUR_TILE_H * ty + ORBFE_BLUR_TILE_H + 2) fuse = false;
      }
    }
    e->fuse_ok[l] = fuse;
    e->tile_first[l] = (int)e->tiles.size();
    for (int ty = 0; ty < tiles_y; ty++)
      for (int tx = 0; tx < tiles_x; tx++) {
        BlurTile bt;
        memset(&bt, 0, sizeof(bt));
        bt.level = (int16_t)l; bt.tx = (int16_t)tx; bt.ty = (int16_t)ty;
        if (fuse) { bt.j0 = (int16_t)jx[tx]; bt.j1 = (int16_t)jx[tx + 1]; bt.r0 = (int16_t)ry[ty]; bt.r1 = (int16_t)ry[ty + 1]; }
        e->tiles.push_back(bt);
      }
    e->tile_count[l] = (int)e->tiles.size() - e->tile_first[l];
  }
  e->blur_mfma = build_blur_tables(e);
  e->total_cells = (int)e->cells.size();
  e->slots_per_image = slot_off;
  e->gkeys_per_image = key_off;
  e->kp_per_image = kp_off;
  e->max_nodes = (maxM + 63) & ~63;
  if (e->max_nodes > 8192) {
    orbfe_set_error("nFeatures too large for the octree node table (%d nodes)", e->max_nodes);
    return ORBFE_ERR_INVALID;
  }
  // LDS budget: node tables + as many keys as fit in 40 KiB, so that four workgroups share a CU (a 128-image batch =
  // 1024 workgroups is then resident at once); levels with more candidates spill their keys to HBM, same results
  {
    const size_t fixed = orbfe_octree_lds_bytes(e->max_nodes, 0);
    const size_t budget = 40 * 1024;
    e->lds_keys = fixed + 8 * 512 <= budget ? (int)((budget - fixed) / 8) : 512;
    e->oct_lds = orbfe_octree_lds_bytes(e->max_nodes, e->lds_keys);
    // a level whose keys leave LDS takes its first ff_depth subdivisions from count tables kept in the idle LDS key array
    // (octree_select_kernel): n_ini * (4 + .. + 4^D) counts, n_ini * 4^D cell -> leaf entries, two (depth, index) words per node
    for (int l = 0; l < e->prm.n_levels; l++) {
      OctLevel& o = e->oct[l];
      o.ff_depth = 0;
      for (int D = 1; D <= 5; D++) {
        const size_t cells = (size_t)o.n_ini << (2 * D);
        const size_t counts = (size_t)o.n_ini * (((size_t)1 << (2 * D + 2)) - 4) / 3;
        const size_t need = counts * 4 + ((cells + 1) & ~(size_t)1) * 2 + (size_t)e->max_nodes * 4;
        if (cells <= 4096 && need <= (size_t)e->lds_keys * 8) o.ff_depth = D;
      }
    }
    if (e->oct_lds > 160 * 1024) {
      orbfe_set_error("octree LDS %zu exceeds 160 KiB", e->oct_lds);
      return ORBFE_ERR_INVALID;
    }
    if (e->oct_lds > 64 * 1024) {
      int rc = orbfe_set_octree_lds(e->oct_lds);
      if (rc != 0) {
        orbfe_set_error("hipFuncSetAttribute(max dynamic LDS %zu) failed: %d", e->oct_lds, rc);
        return ORBFE_ERR_HIP;
      }
    }
  }
  // device tables
  int rc;
  if ((rc = upload(e->d_cells, e->cells.data(), e->cells.size() * sizeof(CellDesc), e->stream))) return rc;
  if ((rc = upload(e->d_groups, e->groups.data(), e->groups.size() * sizeof(FastGroup), e->stream))) return rc;
  {
    // one cell per run: with a handful of images the chip has more wave slots than cells, and a wave that walks four cells one
    // after the other is four times the latency of the stage (single-image FAST 40 -> 12 us)
    std::vector<FastGroup> g1(e->cells.size());
    for (size_t i = 0; i < e->cells.size(); i++) {
      memset(&g1[i], 0, sizeof(FastGroup));
      g1[i].first_cell = (int32_t)i;
      g1[i].n_cells = 1;
      g1[i].level = e->cells[i].level;
    }
    if ((rc = upload(e->d_groups1, g1.data(), g1.size() * sizeof(FastGroup), e->stream))) return rc;
    HIPCHK(hipStreamSynchronize(e->stream));   // g1 goes out of scope
  }
  if ((rc = upload(e->d_tiles, e->tiles.data(), e->tiles.size() * sizeof(BlurTile), e->stream))) return rc;
  if ((rc = upload(e->d_strips, e->strips.data(), e->strips.size() * sizeof(BlurStrip), e->stream))) return rc;
  if ((rc = upload(e->d_blur_tab, e->blur_tab.data(), e->blur_tab.size(), e->stream))) return rc;
  for (int l = 1; l < nl; l++) {
    if ((rc = upload(e->d_xt[l], hxt[l].data(), hxt[l].size() * sizeof(ResizeTap), e->stream))) return rc;
    if ((rc = upload(e->d_yt[l], hyt[l].data(), hyt[l].size() * sizeof(ResizeTap), e->stream))) return rc;
  }
  HIPCHK(hipStreamSynchronize(e->stream));
  e->plan_w = w;
  e->plan_h = h;
  e->cap_images = 0;  // force work-space re-allocation
  return ORBFE_OK;
}

static size_t pyr_bytes_per_image(const orbfe_extractor* e) {
  size_t s = 0;
  for (int l = 0; l < e->prm.n_levels; l++) s += e->lg[l].plane;
  return s;
}

static int ensure_workspace(orbfe_extractor* e, int n_images) {
  if (n_images <= e->cap_images) return ORBFE_OK;
  HIPCHK(hipStreamSynchronize(e->stream));
  const size_t B = (size_t)n_images;
  int rc;
  // + 4 KiB slack: kernels read whole aligned dwords / fixed-size patch rows that may end past the last plane
  if ((rc = dev_alloc(e->d_pyr, pyr_bytes_per_image(e) * B + 4096))) return rc;
  if ((rc = dev_alloc(e->d_blur, pyr_bytes_per_image(e) * B + 4096))) return rc;
  if ((rc = dev_alloc(e->d_cell_cnt, sizeof(int32_t) * e->total_cells * B))) return rc;
  if ((rc = dev_alloc(e->d_cell_off, sizeof(int32_t) * e->total_cells * B))) return rc;
  if ((rc = dev_alloc(e->d_slots, sizeof(uint32_t) * e->slots_per_image * B))) return rc;
  if ((rc = dev_alloc(e->d_gkeys, sizeof(uint64_t) * e->gkeys_per_image * B))) return rc;
  if ((rc = dev_alloc(e->d_lvl_kp, sizeof(uint32_t) * e->kp_per_image * B))) return rc;
  if ((rc = dev_alloc(e->d_lvl_n, sizeof(int32_t) * ORBFE_MAX_LEVELS * B))) return rc;  // 16 per image (aligned 64 B)
  if ((rc = dev_alloc(e->d_err, 256))) return rc;
  HIPCHK(hipMemsetAsync(e->d_err.p, 0, 256, e->stream));
  HIPCHK(hipMemsetAsync(e->d_lvl_n.p, 0, sizeof(int32_t) * ORBFE_MAX_LEVELS * B, e->stream));  // unused level slots stay 0
  HIPCHK(hipStreamSynchronize(e->stream));
  e->cap_images = n_images;
  return ORBFE_OK;
}

// level-major layout: level l of image i at d_pyr + cap_images*off_l + i*plane_l
static uint8_t* level_ptr(const orbfe_extractor* e, const DevBuf& buf, int level, int image) {
  return (uint8_t*)buf.p + (size_t)e->cap_images * e->lg[level].off + (size_t)image * e->lg[level].plane;
}

// which raw levels the level chain leaves tiled: those the fused kernel writes (kind 0, a fusable step in front of them)
static unsigned tiled_mask(const orbfe_extractor* e, int n_images = 1 << 30) {
  // a level is tiled when the kernel that writes it can (the fused level kernel; copy_level0 for a level 0 that is not the caller's
  // image in place) AND the launch that reads it as its source is a level kernel too (the stand-alone resize kernels read rows)
  unsigned m = 0;
  if (!ORBFE_TILED_LEVELS || e->blur_kind != 0) return 0;
  const int nl = e->prm.n_levels;
  for (int l = 0; l < nl; l++) {
    // (a level-0 copy is tiled for a handful of images only: the level kernel streams a tiled level 0 slower than a row-major one
    //  -- +0.06 ms per 256 images -- which a batch does not get back from the gathers; a one-image call does: 0.187 -> 0.179 ms)
    const bool writer = l == 0 ? (!e->ext0 && n_images <= 8) : e->fuse_ok[l - 1];
    const bool reader = l == nl - 1 || e->fuse_ok[l];
    if (writer && reader) m |= 1u << l;
  }
  return m;
}

static void make_view(const orbfe_extractor* e, const DevBuf& buf, PyrView& v, int n_images) {
  memset(&v, 0, sizeof(v));
  v.n_levels = e->prm.n_levels;
  v.tiled = &buf == &e->d_pyr ? tiled_mask(e, n_images) : ~0u;   // the blurred planes are always tiled
  for (int l = 0; l < v.n_levels; l++) {
    v.base[l] = level_ptr(e, buf, l, 0);
    v.img_stride[l] = e->lg[l].plane;
    v.pitch[l] = e->lg[l].pitch;
    v.w[l] = e->lg[l].w;
    v.h[l] = e->lg[l].h;
  }
  if (&buf == &e->d_pyr && e->ext0) {   // level 0 = the caller's images themselves
    v.base[0] = e->ext0;
    v.img_stride[0] = e->ext0_plane;
    v.pitch[0] = e->ext0_pitch;
  }
}
// Host copy of level `level` of image `image` (w x h bytes, rows dst_stride apart), whatever its layout in HBM: through the pinned
// block `stage` (>= pitch x padded rows bytes), synchronous on `s`
static int download_level(orbfe_extractor* e, int level, int image, uint8_t* dst, int dst_stride, uint8_t* stage, hipStream_t s);

// plane of image `image` at pyramid level `level` (raw pyramid) and its row pitch
static const uint8_t* pyr_plane(const orbfe_extractor* e, int level, int image, int* pitch) {
  if (level == 0 && e->ext0) {
    *pitch = e->ext0_pitch;
    return e->ext0 + (size_t)image * e->ext0_plane;
  }
  *pitch = e->lg[level].pitch;
  return level_ptr(e, e->d_pyr, level, image);
}

static size_t level_stage_bytes(const orbfe_extractor* e, int level) {
  return (size_t)e->lg[level].pitch * (size_t)((e->lg[level].h + 7) & ~7);
}
static int download_level(orbfe_extractor* e, int level, int image, uint8_t* dst, int dst_stride, uint8_t* stage, hipStream_t s) {
  const LevelGeom& g = e->lg[level];
  int sp = 0;
  const uint8_t* src = pyr_plane(e, level, image, &sp);
  const bool tiled = (e->last_tiled >> level) & 1u;
  if (tiled) {   // the whole plane as it lies in HBM, un-tiled here
    HIPCHK(hipMemcpyAsync(stage, src, level_stage_bytes(e, level), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    for (int y = 0; y < g.h; y++)
      for (int x0 = 0; x0 < g.w; x0 += 16)
        memcpy(dst + (size_t)y * dst_stride + x0, stage + orbfe_tiled_offset(x0, y, g.pitch), (size_t)std::min(16, g.w - x0));
  } else {
    HIPCHK(hipMemcpy2DAsync(stage, g.pitch, src, sp, g.w, g.h, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    for (int y = 0; y < g.h; y++) memcpy(dst + (size_t)y * dst_stride, stage + (size_t)y * g.pitch, (size_t)g.w);
  }
  return ORBFE_OK;
}

struct StageTimer {
  orbfe_extractor* e;
  hipStream_t s;
  int stage;
  hipEvent_t a = nullptr, b = nullptr;
  bool on = true;
  StageTimer(orbfe_extractor* e_, hipStream_t s_, int st) : e(e_), s(s_), stage(st) {
    if (!e->profile || !(e->profile_mask & (1u << st))) { on = false; return; }
    auto get = [&]() {
      hipEvent_t ev = nullptr;
      if (!e->ev_pool.empty()) { ev = e->ev_pool.back(); e->ev_pool.pop_back(); }
      else (void)hipEventCreate(&ev);
      return ev;
    };
    a = get();
    b = get();
    (void)hipEventRecord(a, s);
  }
  ~StageTimer() {
    if (!on) return;
    (void)hipEventRecord(b, s);
    e->ev_pending.push_back({stage, a, b});
  }
};

static void drain_events(orbfe_extractor* e) {
  for (auto& p : e->ev_pending) {
    float ms = 0;
    if (hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
      e->stage_ms[p.stage] += ms;
      e->stage_launches[p.stage] += 1;
    }
    e->ev_pool.push_back(p.a);
    e->ev_pool.push_back(p.b);
  }
  e->ev_pending.clear();
}

// A captured launch graph holds raw pointers into this handle's buffers (and into nobody else's): it stays valid exactly as long
// as none of them has been reallocated.  FNV-1a over the pointers; a process-wide allocation counter made two handles that warm
// up side by side (left and right extractor) throw each other's graphs away.
static unsigned long long buffer_signature(const orbfe_extractor* e) {
  const void* ptrs[] = {e->d_strips.p, e->d_blur_tab.p, e->d_cells.p, e->d_groups.p, e->d_groups1.p, e->d_tiles.p, e->d_pyr.p, e->d_blur.p,
                        e->d_cell_cnt.p, e->d_cell_off.p, e->d_slots.p, e->d_gkeys.p, e->d_lvl_kp.p, e->d_lvl_n.p, e->d_err.p,
                        e->d_out_kps.p, e->d_out_desc.p, e->d_out_n.p, e->d_in_stage.p, e->h_in, e->h_out};
  unsigned long long h = 1469598103934665603ull;
  auto mix = [&h](unsigned long long v) { for (int i = 0; i < 8; i++) { h ^= (v >> (8 * i)) & 0xff; h *= 1099511628211ull; } };
  for (const void* q : ptrs) mix((unsigned long long)(uintptr_t)q);
  for (int l = 0; l < ORBFE_MAX_LEVELS; l++) { mix((unsigned long long)(uintptr_t)e->d_xt[l].p); mix((unsigned long long)(uintptr_t)e->d_yt[l].p); }
  // a graph also bakes in the level-major layout (level l of image i at cap_images * off_l + i * plane_l) and the plan's geometry:
  // buffers that come back at their old addresses after a reallocation must not revive it
  mix((unsigned long long)e->cap_images); mix((unsigned long long)e->plan_w); mix((unsigned long long)e->plan_h);
  mix((unsigned long long)e->blur_kind);
  return h | 1ull;
}

// Enqueues the whole extractor pipeline for n_images whose level-0 planes are already in d_pyr.  No host-side state changes in
// here: a captured graph replays the launches without running this function.
static int enqueue_pipeline(orbfe_extractor* e, int n_images, orbfe_keypoint* d_kps, uint8_t* d_desc, int cap,
                            int32_t* d_n_out, hipStream_t s) {
  const int nl = e->prm.n_levels;
  PyrView pv, bv;
  make_view(e, e->d_pyr, pv, n_images);
  make_view(e, e->d_blur, bv, n_images);
  // Fused level chain (the default): launch l blurs level l and, from the same staged windows, writes level l + 1 -- every level
  // is read once by the two stages together and the separate resize launches disappear; a step the fused kernel cannot run
  // (scale factors outside its thread layout) falls back to resize + blur launches of its own.  orbfe_debug_blur_kernel(e, 1 | 2)
  // keeps the old order: resize chain here, ONE blur launch over all levels behind the quadtree.
  const bool fused = e->blur_kind == 0;
  auto resize_step = [&](int l) {   // level l - 1 -> l
    orbfe_launch_resize(pv.base[l - 1], pv.pitch[l - 1], pv.img_stride[l - 1], const_cast<uint8_t*>(pv.base[l]),
                        pv.pitch[l], e->lg[l].plane, pv.w[l], pv.h[l], (const ResizeTap*)e->d_xt[l].p,
                        (const ResizeTap*)e->d_yt[l].p, n_images, e->resize_mode[l], s);
  };
  {
    StageTimer t(e, s, ORBFE_STAGE_PYRAMID);
    for (int l = 0; l < nl; l++) {
      const BlurTile* lt = (const BlurTile*)e->d_tiles.p + e->tile_first[l];
      if (!fused) {
        if (l + 1 < nl) resize_step(l + 1);
      } else if (l + 1 < nl && e->fuse_ok[l]) {
        LevelResize rz;
        rz.dst = const_cast<uint8_t*>(pv.base[l + 1]);
        rz.dimg = e->lg[l + 1].plane;
        rz.dpitch = pv.pitch[l + 1]; rz.dw = pv.w[l + 1]; rz.dh = pv.h[l + 1];
        rz.dst_tiled = (int)((pv.tiled >> (l + 1)) & 1u);
        rz.xt = (const ResizeTap*)e->d_xt[l + 1].p;
        rz.yt = (const ResizeTap*)e->d_yt[l + 1].p;
        orbfe_launch_blur_level(pv, bv, lt, e->tile_count[l], &rz, n_images, s);
      } else {
        if (l + 1 < nl) resize_step(l + 1);
        orbfe_launch_blur_level(pv, bv, lt, e->tile_count[l], nullptr, n_images, s);
      }
    }
  }
  {
    StageTimer t(e, s, ORBFE_STAGE_FAST);
    const bool few = n_images <= 8;   // one cell per wave: 8 x 1220 cells = 9.8 k waves, ~1.4 rounds of the chip's resident waves
    orbfe_launch_fast_cells(pv, (const CellDesc*)e->d_cells.p, (const FastGroup*)(few ? e->d_groups1.p : e->d_groups.p),
                            few ? e->total_cells : (int)e->groups.size(), e->total_cells, e->fc_rows, e->fc_span, e->fc_sc,
                            e->fc_bits, (int32_t*)e->d_cell_cnt.p, (uint32_t*)e->d_slots.p, e->slots_per_image,
                            e->prm.ini_th_fast, e->prm.min_th_fast, n_images, s);
  }
  {
    StageTimer t(e, s, ORBFE_STAGE_OCTREE);
    OctParams op;
    memset(&op, 0, sizeof(op));
    for (int l = 0; l < nl; l++) op.lv[l] = e->oct[l];
    op.cells = (const CellDesc*)e->d_cells.p;
    op.cell_cnt = (const int32_t*)e->d_cell_cnt.p;
    op.slots = (const uint32_t*)e->d_slots.p;
    op.cell_off = (int32_t*)e->d_cell_off.p;
    op.gkeys = (unsigned long long*)e->d_gkeys.p;
    op.lvl_kp = (uint32_t*)e->d_lvl_kp.p;
    op.lvl_n = (int32_t*)e->d_lvl_n.p;
    op.err = (int32_t*)e->d_err.p;
    op.total_cells = e->total_cells;
    op.slots_per_image = e->slots_per_image;
    op.gkeys_per_image = e->gkeys_per_image;
    op.kp_per_image = e->kp_per_image;
    op.n_levels = nl;
    op.max_nodes = e->max_nodes;
    op.lds_keys = e->lds_keys;
    orbfe_launch_octree(op, n_images, e->oct_lds, s);
  }
  {
    StageTimer t(e, s, ORBFE_STAGE_BLUR);   // kept when the chain above has blurred already: the stage's event pairs stay one per batch
    BlurMfmaParams mf;
    mf.strips = (const BlurStrip*)e->d_strips.p;
    mf.n_strips = e->blur_mfma ? (int)e->strips.size() : 0;   // 0: the LDS kernel
    mf.tab = (const uint8_t*)e->d_blur_tab.p;
    memcpy(mf.b_off, e->blur_b_off, sizeof(mf.b_off));
    mf.t_off = e->blur_t_off;
    mf.use = e->blur_kind == 1;
    if (!fused) orbfe_launch_blur(pv, bv, (const BlurTile*)e->d_tiles.p, (int)e->tiles.size(), mf, n_images, s);
  }
  {
    StageTimer t(e, s, ORBFE_STAGE_DESCRIBE);
    DescribeParams dp;
    memset(&dp, 0, sizeof(dp));
    dp.pyr = pv;
    dp.blur = bv;
    dp.lvl_kp = (const uint32_t*)e->d_lvl_kp.p;
    dp.lvl_n = (const int32_t*)e->d_lvl_n.p;
    for (int l = 0; l < nl; l++) {
      dp.kp_off[l] = e->oct[l].kp_off;
      dp.kp_cap[l] = e->oct[l].kp_cap;
      dp.scale[l] = e->scale[l];
      dp.kp_size[l] = (float)(int)(31 * e->scale[l]);  // scaledPatchSize, :800
    }
    dp.kp_per_image = e->kp_per_image;
    dp.n_levels = nl;
    dp.out_kps = d_kps;
    dp.out_desc = d_desc;
    dp.out_n = d_n_out;
    dp.cap = cap;
    orbfe_launch_describe(dp, n_images, s);
  }
  hipError_t le = hipGetLastError();
  if (le != hipSuccess) {
    orbfe_set_error("kernel launch failed: %s", hipGetErrorString(le));
    return ORBFE_ERR_HIP;
  }
  return ORBFE_OK;
}

// ------------------------------------------------------------------------------------------------ C ABI
extern "C" int orbfe_extractor_create(const orbfe_params* p, int device, orbfe_extractor** out) {
  if (!p || !out) return ORBFE_ERR_INVALID;
  *out = nullptr;
  if (p->n_levels < 1 || p->n_levels > ORBFE_MAX_LEVELS || p->n_features < 1 || !(p->scale_factor > 1.0f) ||
      p->ini_th_fast < 1 || p->ini_th_fast > 255 || p->min_th_fast < 1 || p->min_th_fast > 255) {   // min > ini is legal: the
                                                          // reference runs FAST(ini) and, on an empty cell, FAST(min) whatever their order
    orbfe_set_error("invalid extractor parameters");
    return ORBFE_ERR_INVALID;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
    orbfe_set_error("no HIP device available (liborbfe has no CPU fallback)");
    return ORBFE_ERR_NO_DEVICE;
  }
  if (device < 0) {
    if (hipGetDevice(&device) != hipSuccess) device = 0;
  }
  if (device >= ndev) {
    orbfe_set_error("device %d out of range (%d visible)", device, ndev);
    return ORBFE_ERR_INVALID;
  }
  HIPCHK(hipSetDevice(device));
  orbfe_extractor* e = new orbfe_extractor();
  e->prm = *p;
  e->device = device;
  // L/src/ORBextractor.cc:411-441; scaleFactor is stored as double (L/include/ORBextractor.h:91)
  const double sf = (double)p->scale_factor;
  e->scale[0] = 1.0f;
  e->sigma2[0] = 1.0f;
  for (int i = 1; i < p->n_levels; i++) {
    e->scale[i] = (float)(e->scale[i - 1] * sf);
    e->sigma2[i] = e->scale[i] * e->scale[i];
  }
  for (int i = 0; i < p->n_levels; i++) {
    e->inv_scale[i] = 1.0f / e->scale[i];
    e->inv_sigma2[i] = 1.0f / e->sigma2[i];
  }
  const float factor = (float)(1.0f / sf);
  float nDesired = p->n_features * (1 - factor) / (1 - (float)pow((double)factor, (double)p->n_levels));
  int sum = 0;
  for (int l = 0; l < p->n_levels - 1; l++) {
    e->feat_per_level[l] = cv_round_f(nDesired);
    sum += e->feat_per_level[l];
    nDesired *= factor;
  }
  e->feat_per_level[p->n_levels - 1] = std::max(p->n_features - sum, 0);
  hipError_t he = hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking);
  if (he != hipSuccess) {
    orbfe_set_error("hipStreamCreate: %s", hipGetErrorString(he));
    delete e;
    return ORBFE_ERR_HIP;
  }
  {  // per-device constant: the sampling pattern as floats (the device symbol lives once per device / code object)
    const int prc = orbfe_upload_pattern_floats();
    if (prc != 0) {
      orbfe_set_error("uploading the BRIEF pattern failed: %d", prc);
      orbfe_extractor_destroy(e);
      return ORBFE_ERR_HIP;
    }
  }
  *out = e;
  return ORBFE_OK;
}

extern "C" int orbfe_extractor_destroy(orbfe_extractor* e) {
  if (!e) return ORBFE_OK;
  (void)hipSetDevice(e->device);
  if (e->stream) (void)hipStreamSynchronize(e->stream);
  drain_events(e);
  for (auto ev : e->ev_pool) (void)hipEventDestroy(ev);
  for (auto& g : e->graphs)
    if (g.exec) (void)hipGraphExecDestroy(g.exec);
  DevBuf* bufs[] = {&e->d_strips, &e->d_blur_tab, &e->d_cells, &e->d_groups, &e->d_groups1, &e->d_tiles, &e->d_pyr, &e->d_blur, &e->d_cell_cnt, &e->d_cell_off, &e->d_slots,
                    &e->d_gkeys, &e->d_lvl_kp, &e->d_lvl_n, &e->d_err, &e->d_out_kps, &e->d_out_desc, &e->d_out_n};
  for (auto b : bufs) dev_free(*b);
  dev_free(e->d_in_stage);
  if (e->h_err) (void)hipHostFree(e->h_err);
  if (e->h_in) (void)hipHostFree(e->h_in);
  if (e->h_out) (void)hipHostFree(e->h_out);
  for (int l = 0; l < ORBFE_MAX_LEVELS; l++) { dev_free(e->d_xt[l]); dev_free(e->d_yt[l]); }
  if (e->stream) (void)hipStreamDestroy(e->stream);
  delete e;
  return ORBFE_OK;
}

#define GETTER(name, field, type)                                                  \
  extern "C" int name(const orbfe_extractor* e, type* out) {                       \
    if (!e || !out) return ORBFE_ERR_INVALID;                                      \
    for (int i = 0; i < e->prm.n_levels; i++) out[i] = e->field[i];                \
    return ORBFE_OK;                                                               \
  }
GETTER(orbfe_extractor_scale_factors, scale, float)
GETTER(orbfe_extractor_inv_scale_factors, inv_scale, float)
GETTER(orbfe_extractor_sigma2, sigma2, float)
GETTER(orbfe_extractor_inv_sigma2, inv_sigma2, float)
GETTER(orbfe_extractor_features_per_level, feat_per_level, int32_t)

extern "C" int orbfe_extractor_levels(const orbfe_extractor* e, int* n) {
  if (!e || !n) return ORBFE_ERR_INVALID;
  *n = e->prm.n_levels;
  return ORBFE_OK;
}

extern "C" int orbfe_pyramid_level_size(const orbfe_extractor* e, int w0, int h0, int level, int* w, int* h) {
  if (!e || !w || !h || level < 0 || level >= e->prm.n_levels) return ORBFE_ERR_INVALID;
  *w = cv_round_f((float)w0 * e->inv_scale[level]);
  *h = cv_round_f((float)h0 * e->inv_scale[level]);
  return ORBFE_OK;
}

extern "C" int orbfe_extractor_max_keypoints(const orbfe_extractor* e, int w, int h, int* cap) {
  if (!e || !cap) return ORBFE_ERR_INVALID;
  int total = 0;
  for (int l = 0; l < e->prm.n_levels; l++) {
    const int lw = cv_round_f((float)w * e->inv_scale[l]), lh = cv_round_f((float)h * e->inv_scale[l]);
    const float width = (float)(lw - 2 * ORBFE_EDGE), height = (float)(lh - 2 * ORBFE_EDGE);
    int nIni = 1;
    if (width >= 30.f && height >= 30.f) nIni = std::max(1, (int)roundf(width / height));
    total += std::max(e->feat_per_level[l] + 3, 4 * nIni);
  }
  *cap = total;
  return ORBFE_OK;
}

extern "C" int orbfe_sync(orbfe_extractor* e) {
  if (!e) return ORBFE_ERR_INVALID;
  HIPCHK(hipStreamSynchronize(e->stream));
  return ORBFE_OK;
}

static int check_device_error(orbfe_extractor* e, hipStream_t s) {
  if (!e->h_err) HIPCHK(hipHostMalloc((void**)&e->h_err, 64, hipHostMallocDefault));
  *e->h_err = 0;
  HIPCHK(hipMemcpyAsync(e->h_err, e->d_err.p, sizeof(int32_t), hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  const int32_t err = *e->h_err;
  if (err) {
    orbfe_set_error("device-side capacity error word 0x%x", err);
    (void)hipMemsetAsync(e->d_err.p, 0, 4, s);
    return ORBFE_ERR_CAPACITY;
  }
  return ORBFE_OK;
}

extern "C" int orbfe_device_status(orbfe_extractor* e) {
  if (!e) return ORBFE_ERR_INVALID;
  std::lock_guard<std::mutex> lk(e->mu);
  HIPCHK(hipSetDevice(e->device));
  if (!e->d_err.p) return ORBFE_OK;
  return check_device_error(e, e->stream);
}

extern "C" int orbfe_extract_batch_device(orbfe_extractor* e, const uint8_t* d_imgs, int n_images, int w, int h,
                                          int stride, size_t image_pitch, orbfe_keypoint* d_kps, uint8_t* d_desc,
                                          int cap, int32_t* d_n_out, void* stream) {
  if (!e || !d_imgs || !d_kps || !d_desc || !d_n_out || n_images < 1 || stride < w || cap < 1) return ORBFE_ERR_INVALID;
  if (((uintptr_t)d_desc & 7) || ((uintptr_t)d_kps & 3)) {
    orbfe_set_error("d_desc must be 8-byte aligned, d_kps 4-byte aligned");
    return ORBFE_ERR_INVALID;
  }
  std::lock_guard<std::mutex> lk(e->mu);
  HIPCHK(hipSetDevice(e->device));
  int rc;
  if ((rc = build_plan(e, w, h))) return rc;
  if ((rc = ensure_workspace(e, n_images))) return rc;
  hipStream_t s = stream ? (hipStream_t)stream : e->stream;
  // Rows that start on 16-byte boundaries and are at least ceil16(w) bytes apart are what every kernel needs of a level (16-byte
  // loads of whole chunks): such images ARE level 0 -- no pitched copy (233 KB of traffic per KITTI image).  They must then stay
  // unchanged until this batch's results, and any later read of its pyramid (orbfe_stereo_match_device, orbfe_device_pyramid),
  // are done.  Anything else (odd widths packed tightly, as cv::Mat rows are) is copied into the pitched level-0 planes.
  const bool in_place = ((uintptr_t)d_imgs & 15) == 0 && (stride & 15) == 0 && (image_pitch & 15) == 0 && stride >= ((w + 15) & ~15) &&
                        image_pitch >= (size_t)stride * (size_t)h;
  {
    StageTimer t(e, s, ORBFE_STAGE_PYRAMID);   // kept when there is nothing to copy: the stage's event pairs stay two per batch
    if (in_place) {
      e->ext0 = d_imgs;
      e->ext0_pitch = stride;
      e->ext0_plane = image_pitch;
    } else {
      e->ext0 = nullptr;
      orbfe_launch_copy0(d_imgs, stride, image_pitch, level_ptr(e, e->d_pyr, 0, 0), e->lg[0].pitch, e->lg[0].plane, w, h,
                         n_images, (int)(tiled_mask(e, n_images) & 1u), s);
    }
  }
  if ((rc = enqueue_pipeline(e, n_images, d_kps, d_desc, cap, d_n_out, s))) return rc;
  e->last_images = n_images;
  e->last_tiled = tiled_mask(e, n_images);
  return ORBFE_OK;
}

extern "C" int orbfe_extract_batch(orbfe_extractor* e, const uint8_t* const* imgs, int n_images, int w, int h,
                                   int stride, orbfe_keypoint* kps, uint8_t* desc, int cap, int32_t* n_out) {
  if (!e || !imgs || !kps || !desc || !n_out || n_images < 1 || cap < 1) return ORBFE_ERR_INVALID;
  if (w < 1 || h < 1) return ORBFE_ERR_EMPTY;
  if (stride < w) return ORBFE_ERR_INVALID;
  std::lock_guard<std::mutex> lk(e->mu);
  HIPCHK(hipSetDevice(e->device));
  int rc;
  if ((rc = build_plan(e, w, h))) return rc;
  if ((rc = ensure_workspace(e, n_images))) return rc;
  hipStream_t s = e->stream;
  const size_t B = (size_t)n_images;
  const size_t img_bytes = (size_t)w * h;  // packed rows in the staging buffers
  const size_t kp_bytes = sizeof(orbfe_keypoint) * (size_t)cap * B, desc_bytes = (size_t)32 * cap * B;
  const size_t hdr_bytes = ((sizeof(int32_t) * (B + 1)) + 255) & ~(size_t)255;  // n_out[B], err
  if (cap != e->out_cap || kp_bytes > e->d_out_kps.bytes || desc_bytes > e->d_out_desc.bytes || hdr_bytes > e->d_out_n.bytes) {
    HIPCHK(hipStreamSynchronize(s));
    if ((rc = dev_alloc(e->d_out_kps, kp_bytes))) return rc;
    if ((rc = dev_alloc(e->d_out_desc, desc_bytes))) return rc;
    if ((rc = dev_alloc(e->d_out_n, hdr_bytes))) return rc;
    e->out_cap = cap;
  }
  if ((rc = dev_alloc(e->d_in_stage, img_bytes * B))) return rc;
  if ((rc = pinned_alloc(e->h_in, e->h_in_bytes, img_bytes * B))) return rc;
  if ((rc = pinned_alloc(e->h_out, e->h_out_bytes, hdr_bytes + kp_bytes + desc_bytes))) return rc;
  // pack the caller's rows into pinned memory, one asynchronous H2D, device-side pitch conversion
  for (int i = 0; i < n_images; i++) {
    if (!imgs[i]) return ORBFE_ERR_INVALID;
    uint8_t* dst = (uint8_t*)e->h_in + (size_t)i * img_bytes;
    if (stride == w) memcpy(dst, imgs[i], img_bytes);
    else for (int y = 0; y < h; y++) memcpy(dst + (size_t)y * w, imgs[i] + (size_t)y * stride, (size_t)w);
  }
  // Latency path (a frame or a stereo pair at a time, what the C++ drop-in does): no copy engine at all.  The level-0 kernel
  // reads the pinned staging buffer over PCIe itself and the descriptor kernel writes keypoints, descriptors and counts
  // straight into pinned host memory (hipHostMalloc memory is device-addressable); each memcpy on the stream costs a launch
  // and a dependency gap (~10 us) that a 120 KB result does not repay.  Larger batches keep the DMA copies.
  const bool zc = n_images <= 2;
  e->ext0 = nullptr;   // host images are staged and copied into the pitched level-0 planes
  uint8_t* ho = (uint8_t*)e->h_out;
  if (!zc) HIPCHK(hipMemcpyAsync(e->d_in_stage.p, e->h_in, img_bytes * B, hipMemcpyHostToDevice, s));
  int32_t* d_hdr = (int32_t*)e->d_out_n.p;
  // the latency path's launches: level-0 copy out of the pinned staging buffer, the pipeline, the device error word
  auto launch_zc = [&]() -> int {
    {
      StageTimer t(e, s, ORBFE_STAGE_PYRAMID);
      orbfe_launch_copy0((const uint8_t*)e->h_in, w, img_bytes, level_ptr(e, e->d_pyr, 0, 0), e->lg[0].pitch, e->lg[0].plane, w, h,
                         n_images, (int)(tiled_mask(e, n_images) & 1u), s);
    }
    int r = enqueue_pipeline(e, n_images, (orbfe_keypoint*)(ho + hdr_bytes), ho + hdr_bytes + kp_bytes, cap, (int32_t*)ho, s);
    if (r) return r;
    HIPCHK(hipMemcpyAsync(ho + sizeof(int32_t) * B, e->d_err.p, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    return ORBFE_OK;
  };
  if (zc) {
    // Every pointer these launches take belongs to the handle (pinned staging in and out, work space), so the whole sequence is
    // captured once per (geometry, capacity) into a hipGraph and replayed: one submission instead of thirteen dependent ones.
    // The first calls launch directly (module loading and attribute calls stay outside the capture); any allocation re-captures.
    orbfe_extractor::LaunchGraph& g = e->graphs[n_images - 1];
    const unsigned long long sig = buffer_signature(e);
    const bool same = g.w == w && g.h == h && g.cap == cap && g.buffers == sig;
    if (!same) {
      if (g.exec) { (void)hipGraphExecDestroy(g.exec); g.exec = nullptr; }
      g.w = w; g.h = h; g.cap = cap; g.buffers = sig; g.warm = 0;
    }
    bool launched = false;
    if (!e->profile && g.warm >= 2) {
      if (!g.exec) {
        hipGraph_t graph = nullptr;
        if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess) {
          const int r = launch_zc();
          const hipError_t ce = hipStreamEndCapture(s, &graph);
          if (r == ORBFE_OK && ce == hipSuccess && graph && hipGraphInstantiate(&g.exec, graph, nullptr, nullptr, 0) != hipSuccess)
            g.exec = nullptr;
          if (graph) (void)hipGraphDestroy(graph);
          if (r != ORBFE_OK || ce != hipSuccess) g.exec = nullptr;
        }
        (void)hipGetLastError();
        if (!g.exec) g.warm = -1000000;   // capture is not available here: direct launches from now on
      }
      if (g.exec) {
        HIPCHK(hipGraphLaunch(g.exec, s));
        launched = true;
      }
    }
    if (!launched) {
      if ((rc = launch_zc())) return rc;
      g.warm++;
    }
  } else {
    {
      StageTimer t(e, s, ORBFE_STAGE_PYRAMID);
      orbfe_launch_copy0((const uint8_t*)e->d_in_stage.p, w, img_bytes, level_ptr(e, e->d_pyr, 0, 0), e->lg[0].pitch, e->lg[0].plane, w,
                         h, n_images, (int)(tiled_mask(e, n_images) & 1u), s);
    }
    if ((rc = enqueue_pipeline(e, n_images, (orbfe_keypoint*)e->d_out_kps.p, (uint8_t*)e->d_out_desc.p, cap, d_hdr, s)))
      return rc;
    // results: header (counts + device error word) and the full padded records, one sync
    HIPCHK(hipMemcpyAsync(d_hdr + B, e->d_err.p, sizeof(int32_t), hipMemcpyDeviceToDevice, s));
    HIPCHK(hipMemcpyAsync(ho, d_hdr, sizeof(int32_t) * (B + 1), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(ho + hdr_bytes, e->d_out_kps.p, kp_bytes, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(ho + hdr_bytes + kp_bytes, e->d_out_desc.p, desc_bytes, hipMemcpyDeviceToHost, s));
  }
  e->last_images = n_images;   // host state of the pipeline: set here, not inside the (replayable) launches
  e->last_tiled = tiled_mask(e, n_images);
  HIPCHK(hipStreamSynchronize(s));
  if (e->profile) drain_events(e);
  const int32_t* hn = (const int32_t*)ho;
  if (hn[B] != 0) {
    orbfe_set_error("device-side capacity error word 0x%x", hn[B]);
    (void)hipMemsetAsync(e->d_err.p, 0, 4, s);
    return ORBFE_ERR_CAPACITY;
  }
  for (int i = 0; i < n_images; i++) {
    n_out[i] = hn[i];
    if (hn[i] > cap) {
      orbfe_set_error("image %d produced %d keypoints, capacity %d", i, hn[i], cap);
      return ORBFE_ERR_CAPACITY;
    }
    memcpy(kps + (size_t)i * cap, ho + hdr_bytes + sizeof(orbfe_keypoint) * (size_t)i * cap, sizeof(orbfe_keypoint) * hn[i]);
    memcpy(desc + (size_t)i * cap * 32, ho + hdr_bytes + kp_bytes + (size_t)32 * i * cap, (size_t)32 * hn[i]);
  }
  return ORBFE_OK;
}

extern "C" int orbfe_extract(orbfe_extractor* e, const uint8_t* img, int w, int h, int stride, orbfe_keypoint* kps,
                             uint8_t* desc, int cap, int* n_out) {
  if (!e || !kps || !desc || !n_out) return ORBFE_ERR_INVALID;
  if (!img || w < 1 || h < 1) return ORBFE_ERR_EMPTY;
  int32_t n = 0;
  const uint8_t* one[1] = {img};
  int rc = orbfe_extract_batch(e, one, 1, w, h, stride, kps, desc, cap, &n);
  *n_out = n;
  return rc;
}

extern "C" int orbfe_pyramid_level(orbfe_extractor* e, int level, uint8_t* dst, int dst_stride, int* w, int* h) {
  if (!e || level < 0 || level >= e->prm.n_levels || e->plan_w == 0 || e->last_images < 1) return ORBFE_ERR_INVALID;
  std::lock_guard<std::mutex> lk(e->mu);
  HIPCHK(hipSetDevice(e->device));
  const LevelGeom& g = e->lg[level];
  if (w) *w = g.w;
  if (h) *h = g.h;
  if (dst) {
    if (dst_stride < g.w) return ORBFE_ERR_INVALID;
    int rc;
    if ((rc = pinned_alloc(e->h_out, e->h_out_bytes, level_stage_bytes(e, level)))) return rc;
    if ((rc = download_level(e, level, 0, dst, dst_stride, (uint8_t*)e->h_out, e->stream))) return rc;
  }
  return ORBFE_OK;
}

extern "C" int orbfe_pyramid_levels(orbfe_extractor* e, uint8_t* const* dst, const int* dst_stride) {
  if (!e || !dst || !dst_stride || e->plan_w == 0 || e->last_images < 1) return ORBFE_ERR_INVALID;
  std::lock_guard<std::mutex> lk(e->mu);
  HIPCHK(hipSetDevice(e->device));
  const int nl = e->prm.n_levels;
  size_t total = 0;
  for (int l = 0; l < nl; l++) total += level_stage_bytes(e, l);
  int rc;
  if ((rc = pinned_alloc(e->h_out, e->h_out_bytes, total))) return rc;
  size_t off = 0;
  for (int l = 0; l < nl; l++) {  // image 0 of every level: its plane as it lies in HBM -> pinned, one sync for all
    int sp = 0;
    const uint8_t* src = pyr_plane(e, l, 0, &sp);
    const bool tiled = (e->last_tiled >> l) & 1u;
    if (tiled || sp == e->lg[l].pitch) HIPCHK(hipMemcpyAsync((uint8_t*)e->h_out + off, src, tiled ? level_stage_bytes(e, l) : (size_t)e->lg[l].pitch * e->lg[l].h, hipMemcpyDeviceToHost, e->stream));
    else HIPCHK(hipMemcpy2DAsync((uint8_t*)e->h_out + off, e->lg[l].pitch, src, sp, e->lg[l].w, e->lg[l].h, hipMemcpyDeviceToHost, e->stream));
    off += level_stage_bytes(e, l);
  }
  HIPCHK(hipStreamSynchronize(e->stream));
  off = 0;
  for (int l = 0; l < nl; l++) {
    const LevelGeom& g = e->lg[l];
    if (!dst[l] || dst_stride[l] < g.w) return ORBFE_ERR_INVALID;
    const uint8_t* st = (const uint8_t*)e->h_out + off;
    if ((e->last_tiled >> l) & 1u) {
      for (int y = 0; y < g.h; y++)
        for (int x0 = 0; x0 < g.w; x0 += 16)
          memcpy(dst[l] + (size_t)y * dst_stride[l] + x0, st + orbfe_tiled_offset(x0, y, g.pitch), (size_t)std::min(16, g.w - x0));
    } else {
      for (int y = 0; y < g.h; y++) memcpy(dst[l] + (size_t)y * dst_stride[l], st + (size_t)y * g.pitch, (size_t)g.w);
    }
    off += level_stage_bytes(e, l);
  }
  return ORBFE_OK;
}

extern "C" int orbfe_device_pyramid(const orbfe_extractor* e, int image, int level, const uint8_t** d_ptr, int* pitch,
                                    int* w, int* h) {
  if (!e || level < 0 || level >= e->prm.n_levels || image < 0 || image >= e->cap_images || e->plan_w == 0)
    return ORBFE_ERR_INVALID;
  int sp = 0;
  const uint8_t* src = pyr_plane(e, level, image, &sp);
  if (d_ptr) *d_ptr = src;
  if (pitch) *pitch = sp;
  if (w) *w = e->lg[level].w;
  if (h) *h = e->lg[level].h;
  return ORBFE_OK;
}

extern "C" int orbfe_device_pyramid_layout(const orbfe_extractor* e, int level, int* tiled) {
  if (!e || !tiled || level < 0 || level >= e->prm.n_levels) return ORBFE_ERR_INVALID;
  *tiled = (e->last_tiled >> level) & 1u ? 1 : 0;
  return ORBFE_OK;
}

// internal accessor for the stereo matcher (match side lives in matcher.cpp)
int orbfe_internal_pyr_view(const orbfe_extractor* e, PyrView* v, int* n_images) {
  if (!e || e->plan_w == 0 || e->cap_images < 1) return ORBFE_ERR_INVALID;
  make_view(e, e->d_pyr, *v, e->last_images);
  v->tiled = e->last_tiled;
  if (n_images) *n_images = e->last_images;
  return ORBFE_OK;
}
int orbfe_internal_tables(const orbfe_extractor* e, float* scale, float* inv_scale, int* n_levels, int* device) {
  for (int l = 0; l < e->prm.n_levels; l++) { scale[l] = e->scale[l]; inv_scale[l] = e->inv_scale[l]; }
  *n_levels = e->prm.n_levels;
  *device = e->device;
  return ORBFE_OK;
}
hipStream_t orbfe_internal_stream(const orbfe_extractor* e) { return e->stream; }

// ---- debug / parity accessors
extern "C" int orbfe_debug_candidates(orbfe_extractor* e, int image, int level, int32_t* x, int32_t* y, int32_t* score,
                                      int cap, int* n) {
  if (!e || !n || level < 0 || level >= e->prm.n_levels || image < 0 || image >= e->last_images) return ORBFE_ERR_INVALID;
  std::lock_guard<std::mutex> lk(e->mu);
  HIPCHK(hipSetDevice(e->device));
  const OctLevel& o = e->oct[level];
  std::vector<int32_t> cnt(std::max(o.n_cells, 1));
  HIPCHK(hipStreamSynchronize(e->stream));
  HIPCHK(hipMemcpy(cnt.data(), (int32_t*)e->d_cell_cnt.p + (size_t)image * e->total_cells + o.cell_begin,
                   sizeof(int32_t) * o.n_cells, hipMemcpyDeviceToHost));
  int total = 0;
  std::vector<uint32_t> tmp;
  for (int c = 0; c < o.n_cells; c++) {
    const CellDesc& cd = e->cells[o.cell_begin + c];
    tmp.resize(std::max(cnt[c], 1));
    if (cnt[c] > 0)
      HIPCHK(hipMemcpy(tmp.data(), (uint32_t*)e->d_slots.p + (size_t)image * e->slots_per_image + cd.slot_off,
                       sizeof(uint32_t) * cnt[c], hipMemcpyDeviceToHost));
    for (int j = 0; j < cnt[c]; j++) {
      if (total < cap) {
        x[total] = tmp[j] & 0xfff;
        y[total] = (tmp[j] >> 12) & 0xfff;
        score[total] = tmp[j] >> 24;
      }
      total++;
    }
  }
  *n = total;
  return total > cap ? ORBFE_ERR_CAPACITY : ORBFE_OK;
}

// 0: the fused level chain (default: blur of level l + resize l -> l + 1 per launch), 1: resize chain + the matrix-core blur (DESIGN
// lesson 31), 2: resize chain + one LDS blur launch -- same bytes, for the parity tests and A/B timing.
// A captured launch graph bakes the choice in: it is dropped here.
extern "C" int orbfe_debug_blur_kernel(orbfe_extractor* e, int kind) {
  if (!e || kind < 0 || kind > 2) return ORBFE_ERR_INVALID;
  std::lock_guard<std::mutex> lk(e->mu);
  if (kind != e->blur_kind)
    for (auto& g : e->graphs) {
      if (g.exec) { (void)hipGraphExecDestroy(g.exec); g.exec = nullptr; }
      g.warm = 0; g.buffers = 0;
    }
  e->blur_kind = kind;
  return ORBFE_OK;
}

extern "C" int orbfe_debug_blurred(orbfe_extractor* e, int image, int level, uint8_t* dst, int dst_stride) {
  if (!e || !dst || level < 0 || level >= e->prm.n_levels || image < 0 || image >= e->last_images) return ORBFE_ERR_INVALID;
  std::lock_guard<std::mutex> lk(e->mu);
  HIPCHK(hipSetDevice(e->device));
  const LevelGeom& g = e->lg[level];
  HIPCHK(hipStreamSynchronize(e->stream));
  // the blurred planes are stored in 16 x 8 pixel tiles (one 128-byte line each, extract_kernels.hip): untile on the host
  const int hp = (g.h + 7) & ~7;
  std::vector<uint8_t> tiled((size_t)g.pitch * hp);
  HIPCHK(hipMemcpy(tiled.data(), level_ptr(e, e->d_blur, level, image), tiled.size(), hipMemcpyDeviceToHost));
  for (int y = 0; y < g.h; y++)
    for (int x = 0; x < g.w; x++)
      dst[(size_t)y * dst_stride + x] = tiled[((size_t)(y >> 3) * (g.pitch >> 4) + (x >> 4)) * 128 + (y & 7) * 16 + (x & 15)];
  return ORBFE_OK;
}

extern "C" int orbfe_debug_pyramid(orbfe_extractor* e, int image, int level, uint8_t* dst, int dst_stride) {
  if (!e || !dst || level < 0 || level >= e->prm.n_levels || image < 0 || image >= e->last_images) return ORBFE_ERR_INVALID;
  std::lock_guard<std::mutex> lk(e->mu);
  HIPCHK(hipSetDevice(e->device));
  const LevelGeom& g = e->lg[level];
  if (dst_stride < g.w) return ORBFE_ERR_INVALID;
  HIPCHK(hipStreamSynchronize(e->stream));
  std::vector<uint8_t> stage(level_stage_bytes(e, level));
  return download_level(e, level, image, dst, dst_stride, stage.data(), e->stream);
}

extern "C" int orbfe_debug_level_keypoints(orbfe_extractor* e, int image, int level, int32_t* x, int32_t* y,
                                           int32_t* score, int cap, int* n) {
  if (!e || !n || level < 0 || level >= e->prm.n_levels || image < 0 || image >= e->last_images) return ORBFE_ERR_INVALID;
  std::lock_guard<std::mutex> lk(e->mu);
  HIPCHK(hipSetDevice(e->device));
  HIPCHK(hipStreamSynchronize(e->stream));
  int32_t cnt = 0;
  HIPCHK(hipMemcpy(&cnt, (int32_t*)e->d_lvl_n.p + (size_t)image * ORBFE_MAX_LEVELS + level, 4, hipMemcpyDeviceToHost));
  std::vector<uint32_t> tmp(std::max(cnt, 1));
  if (cnt > 0)
    HIPCHK(hipMemcpy(tmp.data(), (uint32_t*)e->d_lvl_kp.p + (size_t)image * e->kp_per_image + e->oct[level].kp_off,
                     sizeof(uint32_t) * cnt, hipMemcpyDeviceToHost));
  for (int i = 0; i < cnt && i < cap; i++) {
    x[i] = (tmp[i] & 0xfff) + ORBFE_EDGE;
    y[i] = ((tmp[i] >> 12) & 0xfff) + ORBFE_EDGE;
    score[i] = tmp[i] >> 24;
  }
  *n = cnt;
  return cnt > cap ? ORBFE_ERR_CAPACITY : ORBFE_OK;
}

extern "C" int orbfe_profile_enable(orbfe_extractor* e, int enable) {
  if (!e) return ORBFE_ERR_INVALID;
  std::lock_guard<std::mutex> lk(e->mu);
  e->profile = enable != 0;
  e->profile_mask = enable == 1 ? ~0u : ((unsigned)enable >> 1);
  return ORBFE_OK;
}

// The timed stage launches since the last drain as intervals on ref_event's clock (see include/orbfe.h): what a caller needs
// to tell how long a kernel ran from how long the chip worked on it when two handles' launches overlap.
extern "C" int orbfe_stage_intervals(orbfe_extractor* e, void* ref_event, int32_t* stage, float* start_ms, float* end_ms, int cap,
                                     int32_t* n) {
  if (!e || !ref_event || !n || cap < 0 || (cap > 0 && (!stage || !start_ms || !end_ms))) return ORBFE_ERR_INVALID;
  std::lock_guard<std::mutex> lk(e->mu);
  hipEvent_t ref = (hipEvent_t)ref_event;
  int cnt = 0;
  for (auto& p : e->ev_pending) {
    float ms = 0, ta = 0, tb = 0;
    if (hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
      e->stage_ms[p.stage] += ms;
      e->stage_launches[p.stage] += 1;
      if (cnt < cap && hipEventElapsedTime(&ta, ref, p.a) == hipSuccess && hipEventElapsedTime(&tb, ref, p.b) == hipSuccess) {
        stage[cnt] = p.stage; start_ms[cnt] = ta; end_ms[cnt] = tb;
        cnt++;
      }
    }
    e->ev_pool.push_back(p.a);
    e->ev_pool.push_back(p.b);
  }
  e->ev_pending.clear();
  *n = cnt;
  return ORBFE_OK;
}

extern "C" int orbfe_stage_times(orbfe_extractor* e, float* ms, int32_t* launches, int reset) {
  if (!e) return ORBFE_ERR_INVALID;
  std::lock_guard<std::mutex> lk(e->mu);
  drain_events(e);
  for (int i = 0; i < ORBFE_STAGE_COUNT; i++) {
    if (ms) ms[i] = e->stage_ms[i];
    if (launches) launches[i] = e->stage_launches[i];
    if (reset) { e->stage_ms[i] = 0; e->stage_launches[i] = 0; }
  }
  return ORBFE_OK;
}
