// pipeline.cpp -- the batched stereo front-end step behind one handle, for hosts that do not link the HIP runtime
// (include/orbfe.h: orbfe_pipeline_*).  Nothing here is a second code path: a chunk runs the public device entry points
// (orbfe_extract_batch_device x 2, orbfe_stereo_match_device, orbfe_track_queries_stereo_device,
// orbfe_proj_match_batch_device) on the handle's compute stream; what the handle adds is ownership -- pinned pitched host
// images, device buffers, three streams and the events that order copy-in / compute / copy-out of `slots` buffer sets -- i.e.
// what `bench.py` and `examples/stereo_kitti.py` borrow from torch.
//
// The loop it serves (Source/Examples/Stereo/stereo_kitti.cc:88-106 of the reference, one pair at a time there):
//   imread left / right -> Frame::Frame: ORBextractor x 2 (L/src/Frame.cc:87-94), ComputeStereoMatches (:477-646)
//   -> Tracking::TrackWithMotionModel: UpdateLastFrame / UnprojectStereo (:668-679), SearchByProjection(cur, last, th)
//      (L/src/ORBmatcher.cc:1247-1383)
#include <hip/hip_runtime.h>
#ifndef PIPE_CARRY_COPIES
#define PIPE_CARRY_COPIES 0
#endif
#ifndef PIPE_OUT_KERNEL
#define PIPE_OUT_KERNEL 4   // the output blocks leave by a copy kernel of at least that many workgroups (pipeline_kernels.hip); 0: hipMemcpyAsync (A/B)
#endif
#ifndef PIPE_OUT_KERNEL_MAX
#define PIPE_OUT_KERNEL_MAX 8
#endif
// Workgroups of a copy-out.  A workgroup moves ~5.5 GB/s over the link (posted 16-byte stores), and the FEWER of them the better for the
// kernels beside them -- 37 MB per 2.3 ms chunk (KITTI, every block): 1 / 2 / 3 / 4 / 8 / 16 / 64 workgroups = 53 / 92 / 108 / 109 / 105 /
// 97 / 89 k frames/s (one and two cannot keep up; from eight on the stores in flight slow the chip's other traffic); 2.5 MB: any count.
static inline int out_workgroups(size_t bytes) {
  const size_t w = bytes >> 23;   // one per 8 MB
  return (int)(w < PIPE_OUT_KERNEL ? PIPE_OUT_KERNEL : w > PIPE_OUT_KERNEL_MAX ? PIPE_OUT_KERNEL_MAX : w);
}
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <new>
#include <vector>

#include "../../include/orbfe.h"
#include "pipeline_internal.h"

void orbfe_set_error(const char* fmt, ...);

#define PCHK(expr)                                                                                \
  do {                                                                                            \
    hipError_t _e = (expr);                                                                       \
    if (_e != hipSuccess) {                                                                       \
      orbfe_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return ORBFE_ERR_HIP;                                                                       \
    }                                                                                             \
  } while (0)
#define RCHK(expr)             \
  do {                         \
    const int _rc = (expr);    \
    if (_rc != ORBFE_OK) return _rc; \
  } while (0)

namespace {
inline size_t up256(size_t v) { return (v + 255) & ~(size_t)255; }

// one output block: the same layout in device memory and in the slot's pinned host memory, moved by ONE copy
struct OutLayout {
  size_t n_left, n_right, n_stereo, n_tracked, kps, desc, u_right, depth, assigned, bytes;
  void build(int F, int cap) {
    size_t o = 0;
    auto take = [&o](size_t b) { const size_t at = o; o += up256(b); return at; };
    // ordered so that what a consumer typically asks for (orbfe_pipeline_config.output_mask) is ONE contiguous copy: the counts, then the
    // tracked assignments, then the stereo results, then the keypoint records and the descriptors
    n_left = take(sizeof(int32_t) * F);
    n_right = take(sizeof(int32_t) * F);
    n_stereo = take(sizeof(int32_t) * F);
    n_tracked = take(sizeof(int32_t) * F);
    assigned = take(sizeof(int32_t) * (size_t)F * cap);
    u_right = take(sizeof(float) * (size_t)F * cap);
    depth = take(sizeof(float) * (size_t)F * cap);
    kps = take(sizeof(orbfe_keypoint) * (size_t)F * cap);
    desc = take((size_t)32 * F * cap);
    bytes = o;
  }
};

struct Slot {
  // host (pinned): [left images | right images | cams | poses], and the output block
  uint8_t* h_in = nullptr;
  uint8_t* h_out = nullptr;
  // device
  uint8_t* d_in = nullptr;     // same layout as h_in
  uint8_t* d_out = nullptr;    // OutLayout
  orbfe_keypoint* d_kps_r = nullptr;
  uint8_t* d_desc_r = nullptr;
  uint8_t* d_blocked = nullptr;
  hipEvent_t ev_in = nullptr, ev_done = nullptr, ev_out = nullptr, ev_gather = nullptr, ev_l = nullptr, ev_r = nullptr;
  // the slot's own extractors and matcher: the matching half of a chunk reads the pyramids its extractors left in HBM (the stereo SAD
  // windows), so the NEXT chunk's extraction can only run beside it on other handles (DESIGN lesson 47)
  orbfe_extractor* ex_l = nullptr;
  orbfe_extractor* ex_r = nullptr;
  orbfe_matcher* mt = nullptr;
  bool pending = false;        // submitted, results not yet waited for
  bool gathering = false;      // a record gather of this slot has been enqueued and not yet waited for
  int frames = 0;
};
}  // namespace

// which stream -> priority / creation-order layout the next orbfe_pipeline_create uses (orbfe_debug_pipeline_streams: the A/B knob of
// tools/pipeline_rate.py; the table in pipeline_build says what each measured)
static int g_stream_layout = 6;
extern "C" int orbfe_debug_pipeline_streams(int layout) {
  if (layout < 0 || layout > 7) return ORBFE_ERR_INVALID;
  g_stream_layout = layout;
  return ORBFE_OK;
}

struct orbfe_pipeline {
  orbfe_pipeline_config cfg{};
  int device = 0, cap = 0, pitch = 0;
  size_t image_bytes = 0, in_left = 0, in_right = 0, in_cams = 0, in_poses = 0, in_bytes = 0;
  OutLayout lay{};
  // copy in | left extractor | right extractor | matching half (stereo match ... projection search) | copy out | record gather
  hipStream_t s_in = nullptr, s_l = nullptr, s_r = nullptr, s_cmp = nullptr, s_out = nullptr, s_gat = nullptr;
  std::vector<Slot> slots;
  // shared by the chunks (the matching halves run on ONE stream, strictly ordered): the last frame of the chunk before (descriptors |
  // keypoints | stereo depth | camera | count -- what the first frame of a chunk is searched with) and the queries of the chunk's frames
  uint8_t* d_carry = nullptr;
  size_t cy_kps = 0, cy_depth = 0, cy_cam = 0, cy_n = 0, cy_bytes = 0;
  orbfe_query* d_q = nullptr;          // [batch][cap]
  int32_t* d_nq = nullptr;             // [batch]
  std::mutex mu;
};

static void pipeline_free(orbfe_pipeline* p) {
  if (!p) return;
  (void)hipSetDevice(p->device);
  for (hipStream_t s : {p->s_in, p->s_l, p->s_r, p->s_cmp, p->s_out, p->s_gat})
    if (s) (void)hipStreamSynchronize(s);
  for (Slot& s : p->slots) {
    if (s.mt) (void)orbfe_matcher_destroy(s.mt);
    if (s.ex_l) (void)orbfe_extractor_destroy(s.ex_l);
    if (s.ex_r) (void)orbfe_extractor_destroy(s.ex_r);
    if (s.h_in) (void)hipHostFree(s.h_in);
    if (s.h_out) (void)hipHostFree(s.h_out);
    for (void* d : {(void*)s.d_in, (void*)s.d_out, (void*)s.d_kps_r, (void*)s.d_desc_r, (void*)s.d_blocked})
      if (d) (void)hipFree(d);
    for (hipEvent_t e : {s.ev_in, s.ev_done, s.ev_out, s.ev_gather, s.ev_l, s.ev_r})
      if (e) (void)hipEventDestroy(e);
  }
  for (void* d : {(void*)p->d_carry, (void*)p->d_q, (void*)p->d_nq})
    if (d) (void)hipFree(d);
  for (hipStream_t s : {p->s_in, p->s_l, p->s_r, p->s_cmp, p->s_out, p->s_gat})
    if (s) (void)hipStreamDestroy(s);
  delete p;
}

static int pipeline_build(orbfe_pipeline* p) {
  const orbfe_pipeline_config& c = p->cfg;
  const int F = c.batch;
  p->slots.resize((size_t)c.slots);
  for (Slot& s : p->slots) {
    RCHK(orbfe_extractor_create(&c.extractor, p->device, &s.ex_l));
    RCHK(orbfe_extractor_create(&c.extractor, p->device, &s.ex_r));
    RCHK(orbfe_matcher_create(p->device, &s.mt));
  }
  RCHK(orbfe_extractor_max_keypoints(p->slots[0].ex_l, c.width, c.height, &p->cap));
  const int cap = p->cap;
  p->pitch = (c.width + 63) & ~63;
  p->image_bytes = (size_t)p->pitch * c.height;
  p->in_left = 0;
  p->in_right = up256(p->image_bytes * F);
  p->in_cams = p->in_right + up256(p->image_bytes * F);
  p->in_poses = p->in_cams + up256(sizeof(orbfe_unproject_cam) * F);
  p->in_bytes = p->in_poses + up256(sizeof(orbfe_track_pose) * F);
  p->lay.build(F, cap);
  // Stream -> hardware queue: HIP maps streams onto a handful of hardware queues per priority level (three levels) in creation order,
  // and streams that share a queue run one after the other.  Measured on this handle (tools/pipeline_rate.py, 256-frame KITTI chunks,
  // three slots; frames/s with the images uploaded per chunk | resident in HBM with every block copied out | resident, assignments +
  // counts copied out | resident, counts only) -- round 6, profiles/r06_pipeline.md:
  //   0  all six at the highest priority, copy streams created first (round 5)      54.6 k | 87.7 k |    -   |  94.3 k
  //   1  compute highest, copies / gather middle                                    37.3 k | 71.1 k |    -   |  90.4 k
  //   2  compute highest, copies / gather lowest                                    56.4 k | 90.5 k | 97.7 k | 104.5 k
  //   3  compute middle (torch's level), copies highest                             34.0 k | 65.2 k |    -   |  83.8 k
  //   4  all at the middle priority, compute streams first                          34.4 k | 81.9 k |    -   |  94.4 k
  //   5  compute highest, copy out middle, copy in / gather lowest                  56.3 k | 89.2 k | 98.3 k | 104.6 k
  //   6  compute highest + copy out as the fourth stream of that level, rest lowest 55.9 k | 89.5 k | 99.7 k | 105.1 k   <- default
  //   7  compute middle, copies lowest                                              35.7 k | 87.5 k | 91.5 k |  97.2 k
  // (round 5 also: GPU_MAX_HW_QUEUES = 8 with layout 0: 83 k.)  What matters: the three compute streams alone on their priority
  // level's queues; a copy stream that lands on a compute stream's queue delays that eye's kernels by the copy (layout 0: the 37 MB
  // copy-out shared the right extractor's queue).  The torch-driven step of bench.py (three streams, nothing else) does 105-107 k.
  int prio_low = 0, prio_high = 0;
  PCHK(hipDeviceGetStreamPriorityRange(&prio_low, &prio_high));
  const int prio_mid = (prio_low + prio_high) / 2;
  switch (g_stream_layout) {
    default:
    case 0:   // round 5: all six at the highest priority, the copy streams created first
      PCHK(hipStreamCreateWithPriority(&p->s_in, hipStreamNonBlocking, prio_high));
      PCHK(hipStreamCreateWithPriority(&p->s_out, hipStreamNonBlocking, prio_high));
      PCHK(hipStreamCreateWithPriority(&p->s_gat, hipStreamNonBlocking, prio_high));
      PCHK(hipStreamCreateWithPriority(&p->s_cmp, hipStreamNonBlocking, prio_high));
      PCHK(hipStreamCreateWithPriority(&p->s_l, hipStreamNonBlocking, prio_high));
      PCHK(hipStreamCreateWithPriority(&p->s_r, hipStreamNonBlocking, prio_high));
      break;
    case 1:   // the three compute streams at the highest priority (a level of their own), the copy / gather streams at the middle one
      PCHK(hipStreamCreateWithPriority(&p->s_l, hipStreamNonBlocking, prio_high));
      PCHK(hipStreamCreateWithPriority(&p->s_r, hipStreamNonBlocking, prio_high));
      PCHK(hipStreamCreateWithPriority(&p->s_cmp, hipStreamNonBlocking, prio_high));
      PCHK(hipStreamCreateWithPriority(&p->s_in, hipStreamNonBlocking, prio_mid));
      PCHK(hipStreamCreateWithPriority(&p->s_out, hipStreamNonBlocking, prio_mid));
      PCHK(hipStreamCreateWithPriority(&p->s_gat, hipStreamNonBlocking, prio_mid));
      break;
    case 2:   // the same with the copy / gather streams at the lowest priority
      PCHK(hipStreamCreateWithPriority(&p->s_l, hipStreamNonBlocking, prio_high));
      PCHK(hipStreamCreateWithPriority(&p->s_r, hipStreamNonBlocking, prio_high));
      PCHK(hipStreamCreateWithPriority(&p->s_cmp, hipStreamNonBlocking, prio_high));
      PCHK(hipStreamCreateWithPriority(&p->s_in, hipStreamNonBlocking, prio_low));
      PCHK(hipStreamCreateWithPriority(&p->s_out, hipStreamNonBlocking, prio_low));
      PCHK(hipStreamCreateWithPriority(&p->s_gat, hipStreamNonBlocking, prio_low));
      break;
    case 3:   // compute streams at the middle priority (what torch's streams have), copies at the highest
      PCHK(hipStreamCreateWithPriority(&p->s_l, hipStreamNonBlocking, prio_mid));
      PCHK(hipStreamCreateWithPriority(&p->s_r, hipStreamNonBlocking, prio_mid));
      PCHK(hipStreamCreateWithPriority(&p->s_cmp, hipStreamNonBlocking, prio_mid));
      PCHK(hipStreamCreateWithPriority(&p->s_in, hipStreamNonBlocking, prio_high));
      PCHK(hipStreamCreateWithPriority(&p->s_out, hipStreamNonBlocking, prio_high));
      PCHK(hipStreamCreateWithPriority(&p->s_gat, hipStreamNonBlocking, prio_high));
      break;
    case 5:   // compute at the highest priority, copy out at the middle one, copy in / gather at the lowest
      PCHK(hipStreamCreateWithPriority(&p->s_l, hipStreamNonBlocking, prio_high));
      PCHK(hipStreamCreateWithPriority(&p->s_r, hipStreamNonBlocking, prio_high));
      PCHK(hipStreamCreateWithPriority(&p->s_cmp, hipStreamNonBlocking, prio_high));
      PCHK(hipStreamCreateWithPriority(&p->s_out, hipStreamNonBlocking, prio_mid));
      PCHK(hipStreamCreateWithPriority(&p->s_in, hipStreamNonBlocking, prio_low));
      PCHK(hipStreamCreateWithPriority(&p->s_gat, hipStreamNonBlocking, prio_low));
      break;
    case 6:   // compute at the highest priority and copy out as the FOURTH stream of that level, copy in / gather at the lowest
      PCHK(hipStreamCreateWithPriority(&p->s_l, hipStreamNonBlocking, prio_high));
      PCHK(hipStreamCreateWithPriority(&p->s_r, hipStreamNonBlocking, prio_high));
      PCHK(hipStreamCreateWithPriority(&p->s_cmp, hipStreamNonBlocking, prio_high));
      PCHK(hipStreamCreateWithPriority(&p->s_out, hipStreamNonBlocking, prio_high));
      PCHK(hipStreamCreateWithPriority(&p->s_in, hipStreamNonBlocking, prio_low));
      PCHK(hipStreamCreateWithPriority(&p->s_gat, hipStreamNonBlocking, prio_low));
      break;
    case 7:   // compute at the middle priority, every copy at the lowest
      PCHK(hipStreamCreateWithPriority(&p->s_l, hipStreamNonBlocking, prio_mid));
      PCHK(hipStreamCreateWithPriority(&p->s_r, hipStreamNonBlocking, prio_mid));
      PCHK(hipStreamCreateWithPriority(&p->s_cmp, hipStreamNonBlocking, prio_mid));
      PCHK(hipStreamCreateWithPriority(&p->s_in, hipStreamNonBlocking, prio_low));
      PCHK(hipStreamCreateWithPriority(&p->s_out, hipStreamNonBlocking, prio_low));
      PCHK(hipStreamCreateWithPriority(&p->s_gat, hipStreamNonBlocking, prio_low));
      break;
    case 4:   // all at the middle priority, compute streams first
      PCHK(hipStreamCreateWithPriority(&p->s_l, hipStreamNonBlocking, prio_mid));
      PCHK(hipStreamCreateWithPriority(&p->s_r, hipStreamNonBlocking, prio_mid));
      PCHK(hipStreamCreateWithPriority(&p->s_cmp, hipStreamNonBlocking, prio_mid));
      PCHK(hipStreamCreateWithPriority(&p->s_in, hipStreamNonBlocking, prio_mid));
      PCHK(hipStreamCreateWithPriority(&p->s_out, hipStreamNonBlocking, prio_mid));
      PCHK(hipStreamCreateWithPriority(&p->s_gat, hipStreamNonBlocking, prio_mid));
      break;
  }
  p->cy_kps = (size_t)cap * 32;
  p->cy_depth = p->cy_kps + (size_t)cap * sizeof(orbfe_keypoint);
  p->cy_cam = p->cy_depth + (size_t)cap * sizeof(float);
  p->cy_n = p->cy_cam + sizeof(orbfe_unproject_cam);
  p->cy_bytes = p->cy_n + sizeof(int32_t);
  PCHK(hipMalloc((void**)&p->d_carry, p->cy_bytes));
  PCHK(hipMalloc((void**)&p->d_q, sizeof(orbfe_query) * (size_t)F * cap));
  PCHK(hipMalloc((void**)&p->d_nq, sizeof(int32_t) * F));
  PCHK(hipMemset(p->d_carry, 0, p->cy_bytes));
  float sf[ORBFE_MAX_LEVELS] = {0};
  RCHK(orbfe_extractor_scale_factors(p->slots[0].ex_l, sf));
  for (Slot& s : p->slots) {
    PCHK(hipHostMalloc((void**)&s.h_in, p->in_bytes, hipHostMallocDefault));
    PCHK(hipHostMalloc((void**)&s.h_out, p->lay.bytes, hipHostMallocDefault));
    PCHK(hipMalloc((void**)&s.d_in, p->in_bytes));
    PCHK(hipMalloc((void**)&s.d_out, p->lay.bytes));
    PCHK(hipMalloc((void**)&s.d_kps_r, sizeof(orbfe_keypoint) * (size_t)F * cap));
    PCHK(hipMalloc((void**)&s.d_desc_r, (size_t)32 * F * cap));
    PCHK(hipMalloc((void**)&s.d_blocked, (size_t)F * cap));
    PCHK(hipMemset(s.d_in, 0, p->in_bytes));
    PCHK(hipMemset(s.d_out, 0, p->lay.bytes));
    memset(s.h_in, 0, p->in_bytes);
    memset(s.h_out, 0, p->lay.bytes);
    PCHK(hipEventCreateWithFlags(&s.ev_in, hipEventDisableTiming));
    PCHK(hipEventCreateWithFlags(&s.ev_done, hipEventDisableTiming));
    PCHK(hipEventCreateWithFlags(&s.ev_out, hipEventDisableTiming));
    PCHK(hipEventCreateWithFlags(&s.ev_gather, hipEventDisableTiming));
    PCHK(hipEventCreateWithFlags(&s.ev_l, hipEventDisableTiming));
    PCHK(hipEventCreateWithFlags(&s.ev_r, hipEventDisableTiming));
    // the records the examples use: identity pose (the constant-velocity prediction with zero velocity: Tcw = Tlw, expressed in
    // the last camera's frame), the configured intrinsics
    orbfe_unproject_cam* cams = reinterpret_cast<orbfe_unproject_cam*>(s.h_in + p->in_cams);
    orbfe_track_pose* poses = reinterpret_cast<orbfe_track_pose*>(s.h_in + p->in_poses);
    for (int f = 0; f < F; f++) {
      orbfe_unproject_cam& cm = cams[f];
      memset(&cm, 0, sizeof(cm));
      cm.Rwc[0] = cm.Rwc[4] = cm.Rwc[8] = 1.0f;
      cm.cx = c.cx; cm.cy = c.cy; cm.invfx = 1.0f / c.fx; cm.invfy = 1.0f / c.fy;
      orbfe_track_pose& ps = poses[f];
      memset(&ps, 0, sizeof(ps));
      ps.Rcw[0] = ps.Rcw[4] = ps.Rcw[8] = 1.0f;
      ps.fx = c.fx; ps.fy = c.fy; ps.cx = c.cx; ps.cy = c.cy; ps.mbf = c.bf;
      ps.min_x = 0.0f; ps.max_x = (float)c.width; ps.min_y = 0.0f; ps.max_y = (float)c.height;
      ps.th = c.th;
      for (int l = 0; l < c.extractor.n_levels; l++) ps.scale_factors[l] = sf[l];
    }
  }
  // plan, work space and code objects now, not inside the first chunk: one chunk of synthetic content through slot 0
  {
    Slot& s = p->slots[0];
    for (int f = 0; f < F; f++)
      for (int y = 0; y < c.height; y++) {
        uint8_t* rl = s.h_in + p->in_left + (size_t)f * p->image_bytes + (size_t)y * p->pitch;
        uint8_t* rr = s.h_in + p->in_right + (size_t)f * p->image_bytes + (size_t)y * p->pitch;
        for (int x = 0; x < c.width; x++) {
          uint32_t k = (uint32_t)((x + 2 * f) / 24) * 73856093u ^ (uint32_t)(y / 24) * 19349663u;
          k ^= k >> 13; k *= 0x5bd1e995u; k ^= k >> 15;
          int v = 40 + (int)(k % 176);
          if ((((x + 2 * f) % 24) == 7 && (y % 24) == 11) || (((x + 2 * f) % 24) == 17 && (y % 24) == 5)) v = (k & 1) ? 250 : 5;
          rl[x] = (uint8_t)v;
          rr[x > 12 ? x - 12 : 0] = (uint8_t)v;
        }
      }
    // every slot's handles: the same chunk through each of them
    for (int k = 0; k < c.slots; k++) {
      Slot& sk = p->slots[(size_t)k];
      if (k > 0) {
        memcpy(sk.h_in + p->in_left, s.h_in + p->in_left, p->image_bytes * F);
        memcpy(sk.h_in + p->in_right, s.h_in + p->in_right, p->image_bytes * F);
      }
      int rc = orbfe_pipeline_submit(p, k, F, 0);
      if (rc == ORBFE_OK) rc = orbfe_pipeline_wait(p, k);
      if (rc != ORBFE_OK) return rc;
    }
    for (Slot& sk : p->slots) {
      memset(sk.h_in + p->in_left, 0, p->image_bytes * F);
      memset(sk.h_in + p->in_right, 0, p->image_bytes * F);
    }
    PCHK(hipMemset(p->d_carry + p->cy_n, 0, sizeof(int32_t)));   // the warm-up chunk is nobody's predecessor
  }
  return ORBFE_OK;
}

extern "C" int orbfe_pipeline_create(const orbfe_pipeline_config* cfg, int device, orbfe_pipeline** out) {
  if (!cfg || !out) return ORBFE_ERR_INVALID;
  *out = nullptr;
  if (cfg->width < 1 || cfg->height < 1 || cfg->batch < 1 || cfg->batch > 4096 || cfg->slots < 1 || cfg->slots > 4 ||
      !(cfg->fx > 0.0f) || !(cfg->fy > 0.0f) || cfg->output_mask < 0 || cfg->output_mask > 31) {
    orbfe_set_error("invalid pipeline configuration");
    return ORBFE_ERR_INVALID;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
    orbfe_set_error("no HIP device available (liborbfe has no CPU fallback)");
    return ORBFE_ERR_NO_DEVICE;
  }
  if (device < 0 && hipGetDevice(&device) != hipSuccess) device = 0;
  if (device >= ndev) return ORBFE_ERR_INVALID;
  PCHK(hipSetDevice(device));
  orbfe_pipeline* p = new (std::nothrow) orbfe_pipeline();
  if (!p) return ORBFE_ERR_ALLOC;
  p->cfg = *cfg;
  p->device = device;
  int rc;
  try {
    rc = pipeline_build(p);
  } catch (...) {
    orbfe_set_error("orbfe_pipeline_create: out of host memory");
    rc = ORBFE_ERR_ALLOC;
  }
  if (rc != ORBFE_OK) {
    pipeline_free(p);
    return rc;
  }
  *out = p;
  return ORBFE_OK;
}

extern "C" int orbfe_pipeline_destroy(orbfe_pipeline* p) {
  pipeline_free(p);
  return ORBFE_OK;
}

extern "C" int orbfe_pipeline_input(orbfe_pipeline* p, int slot, orbfe_pipeline_input_view* in) {
  if (!p || !in || slot < 0 || slot >= (int)p->slots.size()) return ORBFE_ERR_INVALID;
  Slot& s = p->slots[(size_t)slot];
  in->left = s.h_in + p->in_left;
  in->right = s.h_in + p->in_right;
  in->pitch = p->pitch;
  in->image_bytes = p->image_bytes;
  in->cams = reinterpret_cast<orbfe_unproject_cam*>(s.h_in + p->in_cams);
  in->poses = reinterpret_cast<orbfe_track_pose*>(s.h_in + p->in_poses);
  return ORBFE_OK;
}

extern "C" int orbfe_pipeline_output(orbfe_pipeline* p, int slot, orbfe_pipeline_output_view* out) {
  if (!p || !out || slot < 0 || slot >= (int)p->slots.size()) return ORBFE_ERR_INVALID;
  const uint8_t* b = p->slots[(size_t)slot].h_out;
  const OutLayout& L = p->lay;
  out->cap = p->cap;
  out->n_left = reinterpret_cast<const int32_t*>(b + L.n_left);
  out->kps_left = reinterpret_cast<const orbfe_keypoint*>(b + L.kps);
  out->desc_left = b + L.desc;
  out->n_right = reinterpret_cast<const int32_t*>(b + L.n_right);
  out->u_right = reinterpret_cast<const float*>(b + L.u_right);
  out->depth = reinterpret_cast<const float*>(b + L.depth);
  out->n_stereo = reinterpret_cast<const int32_t*>(b + L.n_stereo);
  out->assigned = reinterpret_cast<const int32_t*>(b + L.assigned);
  out->n_tracked = reinterpret_cast<const int32_t*>(b + L.n_tracked);
  return ORBFE_OK;
}

extern "C" int orbfe_pipeline_device_records(orbfe_pipeline* p, int slot, const int32_t** d_n, const orbfe_keypoint** d_kps,
                                             const uint8_t** d_desc, int* cap) {
  if (!p || slot < 0 || slot >= (int)p->slots.size()) return ORBFE_ERR_INVALID;
  const uint8_t* b = p->slots[(size_t)slot].d_out;
  if (d_n) *d_n = reinterpret_cast<const int32_t*>(b + p->lay.n_left);
  if (d_kps) *d_kps = reinterpret_cast<const orbfe_keypoint*>(b + p->lay.kps);
  if (d_desc) *d_desc = b + p->lay.desc;
  if (cap) *cap = p->cap;
  return ORBFE_OK;
}

extern "C" void* orbfe_pipeline_stream(orbfe_pipeline* p) { return p ? (void*)p->s_cmp : nullptr; }

// One chunk through a slot.  Streams: copy in -> left | right extractor (two streams, as the reference's two threads) -> matching half
// (stereo match, unproject, track queries, projection search; ONE stream for all chunks: the carried last frame orders them) -> copy out.
// The extraction of this chunk runs beside the matching half of the chunk before it, which works on another slot's handles.
static int submit_chunk(orbfe_pipeline* p, int slot, int n, int has_predecessor, bool resident) {
  if (!p || slot < 0 || slot >= (int)p->slots.size() || n < 0 || n > p->cfg.batch) return ORBFE_ERR_INVALID;
  std::lock_guard<std::mutex> lk(p->mu);
  PCHK(hipSetDevice(p->device));
  Slot& s = p->slots[(size_t)slot];
  const orbfe_pipeline_config& c = p->cfg;
  const int F = c.batch, cap = p->cap;
  const OutLayout& L = p->lay;
  if (s.pending) PCHK(hipEventSynchronize(s.ev_out));   // the caller did not wait: the host block is about to be overwritten
  s.pending = true;
  s.frames = n;
  // ---- copy in: behind the kernels that still read this slot's images (level 0 of their pyramids in place) and its pyramids
  PCHK(hipStreamWaitEvent(p->s_in, s.ev_done, 0));
  if (n > 0) {
    if (!resident) {
      PCHK(hipMemcpyAsync(s.d_in + p->in_left, s.h_in + p->in_left, p->image_bytes * n, hipMemcpyHostToDevice, p->s_in));
      PCHK(hipMemcpyAsync(s.d_in + p->in_right, s.h_in + p->in_right, p->image_bytes * n, hipMemcpyHostToDevice, p->s_in));
    }
    PCHK(hipMemcpyAsync(s.d_in + p->in_cams, s.h_in + p->in_cams, p->in_bytes - p->in_cams, hipMemcpyHostToDevice, p->s_in));
  }
  PCHK(hipEventRecord(s.ev_in, p->s_in));
  // ---- every compute stream: behind the copy in, behind the copy out of this slot's previous results and a gather that still reads them
  hipStream_t cs = p->s_cmp;
  for (hipStream_t st : {p->s_l, p->s_r, cs}) {
    PCHK(hipStreamWaitEvent(st, s.ev_in, 0));
    PCHK(hipStreamWaitEvent(st, s.ev_out, 0));
    PCHK(hipStreamWaitEvent(st, s.ev_gather, 0));
  }
  int32_t* d_nl = reinterpret_cast<int32_t*>(s.d_out + L.n_left);
  int32_t* d_nr = reinterpret_cast<int32_t*>(s.d_out + L.n_right);
  int32_t* d_nst = reinterpret_cast<int32_t*>(s.d_out + L.n_stereo);
  int32_t* d_ntr = reinterpret_cast<int32_t*>(s.d_out + L.n_tracked);
  orbfe_keypoint* d_kl = reinterpret_cast<orbfe_keypoint*>(s.d_out + L.kps);
  uint8_t* d_dl = s.d_out + L.desc;
  float* d_ur = reinterpret_cast<float*>(s.d_out + L.u_right);
  float* d_depth = reinterpret_cast<float*>(s.d_out + L.depth);
  int32_t* d_assigned = reinterpret_cast<int32_t*>(s.d_out + L.assigned);
  if (n < F) {   // rows behind the chunk carry no keypoints (the gather pads short chunks with them)
    PCHK(hipMemsetAsync(d_nl + n, 0, sizeof(int32_t) * (F - n), cs));
    PCHK(hipMemsetAsync(d_nr + n, 0, sizeof(int32_t) * (F - n), cs));
    PCHK(hipMemsetAsync(d_nst + n, 0, sizeof(int32_t) * (F - n), cs));
    PCHK(hipMemsetAsync(d_ntr + n, 0, sizeof(int32_t) * (F - n), cs));
  }
  if (n > 0) {
    const orbfe_unproject_cam* d_cams = reinterpret_cast<const orbfe_unproject_cam*>(s.d_in + p->in_cams);
    const orbfe_track_pose* d_poses = reinterpret_cast<const orbfe_track_pose*>(s.d_in + p->in_poses);
    RCHK(orbfe_extract_batch_device(s.ex_l, s.d_in + p->in_left, n, c.width, c.height, p->pitch, p->image_bytes, d_kl, d_dl, cap,
                                    d_nl, p->s_l));
    PCHK(hipEventRecord(s.ev_l, p->s_l));
    RCHK(orbfe_extract_batch_device(s.ex_r, s.d_in + p->in_right, n, c.width, c.height, p->pitch, p->image_bytes, s.d_kps_r,
                                    s.d_desc_r, cap, d_nr, p->s_r));
    PCHK(hipEventRecord(s.ev_r, p->s_r));
    PCHK(hipStreamWaitEvent(cs, s.ev_l, 0));
    PCHK(hipStreamWaitEvent(cs, s.ev_r, 0));
    RCHK(orbfe_stereo_match_device(s.mt, s.ex_l, s.ex_r, n, d_kl, d_dl, d_nl, s.d_kps_r, s.d_desc_r, d_nr, cap, c.bf,
                                   c.bf / c.fx, d_ur, d_depth, d_nst, cs));
    // frame j is searched with the stereo points of frame j - 1 (frame 0: of the carried last frame of the chunk before); the points are
    // unprojected and projected in one pass, no point records in between
    RCHK(orbfe_track_queries_stereo_device(n, d_kl, d_dl, d_nl, d_depth, cap, d_cams, 1,
                                           reinterpret_cast<const orbfe_keypoint*>(p->d_carry + p->cy_kps), p->d_carry,
                                           reinterpret_cast<const int32_t*>(p->d_carry + p->cy_n),
                                           reinterpret_cast<const float*>(p->d_carry + p->cy_depth),
                                           reinterpret_cast<const orbfe_unproject_cam*>(p->d_carry + p->cy_cam), d_poses, 1, p->d_q,
                                           p->d_nq, cs));
    PCHK(hipMemsetAsync(s.d_blocked, 0, (size_t)n * cap, cs));
    PCHK(hipMemsetAsync(d_assigned, 0xff, sizeof(int32_t) * (size_t)n * cap, cs));
    RCHK(orbfe_proj_match_batch_device(s.mt, n, d_kl, d_dl, d_nl, d_ur, cap, 0.0f, (float)c.width, 0.0f, (float)c.height, p->d_q,
                                       p->d_nq, cap, 1, 0.9f, c.check_orientation, s.d_blocked, d_assigned, d_ntr, cs));
    if (!has_predecessor) {   // the first frame of a sequence (or of a rank's shard) is searched against nothing
      PCHK(hipMemsetAsync(d_ntr, 0, sizeof(int32_t), cs));
      PCHK(hipMemsetAsync(d_assigned, 0xff, sizeof(int32_t) * (size_t)cap, cs));
    }
    // carry: the last frame's keypoints, descriptors, depth, camera and count for the first frame of the next chunk (one launch)
    const size_t last = (size_t)(n - 1) * cap;
#if PIPE_CARRY_COPIES   // five device-to-device copies (A/B)
    PCHK(hipMemcpyAsync(p->d_carry, d_dl + last * 32, (size_t)cap * 32, hipMemcpyDeviceToDevice, cs));
    PCHK(hipMemcpyAsync(p->d_carry + p->cy_kps, d_kl + last, (size_t)cap * sizeof(orbfe_keypoint), hipMemcpyDeviceToDevice, cs));
    PCHK(hipMemcpyAsync(p->d_carry + p->cy_depth, d_depth + last, (size_t)cap * sizeof(float), hipMemcpyDeviceToDevice, cs));
    PCHK(hipMemcpyAsync(p->d_carry + p->cy_cam, d_cams + (n - 1), sizeof(orbfe_unproject_cam), hipMemcpyDeviceToDevice, cs));
    PCHK(hipMemcpyAsync(p->d_carry + p->cy_n, d_nl + (n - 1), sizeof(int32_t), hipMemcpyDeviceToDevice, cs));
#else
    orbfe_launch_carry_frame(d_dl + last * 32, d_kl + last, d_depth + last, d_cams + (n - 1), d_nl + (n - 1), cap, p->d_carry,
                             reinterpret_cast<orbfe_keypoint*>(p->d_carry + p->cy_kps), reinterpret_cast<float*>(p->d_carry + p->cy_depth),
                             reinterpret_cast<orbfe_unproject_cam*>(p->d_carry + p->cy_cam),
                             reinterpret_cast<int32_t*>(p->d_carry + p->cy_n), cs);
    PCHK(hipGetLastError());
#endif
  }
  PCHK(hipEventRecord(s.ev_done, cs));
  // ---- copy out: the whole block in one copy, or the blocks the configuration asks for (the counts always: they lead the block)
  PCHK(hipStreamWaitEvent(p->s_out, s.ev_done, 0));
  const int om = c.output_mask;
  if (om == 0) {
#if PIPE_OUT_KERNEL
    orbfe_launch_copy_block(s.d_out, s.h_out, L.bytes, out_workgroups(L.bytes), p->s_out);
    PCHK(hipGetLastError());
#else
    PCHK(hipMemcpyAsync(s.h_out, s.d_out, L.bytes, hipMemcpyDeviceToHost, p->s_out));
#endif
  } else {
    // block boundaries in layout order: counts | assigned | u_right, depth | keypoints | descriptors; adjacent wanted blocks go in one copy
    const size_t edge[6] = {0, L.assigned, L.u_right, L.kps, L.desc, L.bytes};
    const bool want[5] = {true, (om & ORBFE_PIPE_OUT_ASSIGNED) != 0, (om & ORBFE_PIPE_OUT_STEREO) != 0, (om & ORBFE_PIPE_OUT_KEYPOINTS) != 0,
                          (om & ORBFE_PIPE_OUT_DESCRIPTORS) != 0};
    for (int b = 0; b < 5;) {
      if (!want[b]) { b++; continue; }
      int e = b;
      while (e + 1 < 5 && want[e + 1]) e++;
#if PIPE_OUT_KERNEL   // (block boundaries are multiples of 256 bytes)
      orbfe_launch_copy_block(s.d_out + edge[b], s.h_out + edge[b], edge[e + 1] - edge[b], out_workgroups(edge[e + 1] - edge[b]), p->s_out);
      PCHK(hipGetLastError());
#else
      PCHK(hipMemcpyAsync(s.h_out + edge[b], s.d_out + edge[b], edge[e + 1] - edge[b], hipMemcpyDeviceToHost, p->s_out));
#endif
      b = e + 1;
    }
  }
  PCHK(hipEventRecord(s.ev_out, p->s_out));
  return ORBFE_OK;
}

extern "C" int orbfe_pipeline_submit(orbfe_pipeline* p, int slot, int n, int has_predecessor) {
  return submit_chunk(p, slot, n, has_predecessor, false);
}

extern "C" int orbfe_pipeline_submit_resident(orbfe_pipeline* p, int slot, int n, int has_predecessor) {
  return submit_chunk(p, slot, n, has_predecessor, true);
}

extern "C" int orbfe_pipeline_device_input(orbfe_pipeline* p, int slot, uint8_t** d_left, uint8_t** d_right, int* pitch,
                                           size_t* image_bytes) {
  if (!p || slot < 0 || slot >= (int)p->slots.size()) return ORBFE_ERR_INVALID;
  Slot& s = p->slots[(size_t)slot];
  if (d_left) *d_left = s.d_in + p->in_left;
  if (d_right) *d_right = s.d_in + p->in_right;
  if (pitch) *pitch = p->pitch;
  if (image_bytes) *image_bytes = p->image_bytes;
  return ORBFE_OK;
}

extern "C" int orbfe_pipeline_wait(orbfe_pipeline* p, int slot) {
  if (!p || slot < 0 || slot >= (int)p->slots.size()) return ORBFE_ERR_INVALID;
  std::lock_guard<std::mutex> lk(p->mu);
  PCHK(hipSetDevice(p->device));
  Slot& s = p->slots[(size_t)slot];
  if (!s.pending) return ORBFE_OK;
  PCHK(hipEventSynchronize(s.ev_out));
  s.pending = false;
  // an internal table overflow of either extractor (ORBFE_ERR_CAPACITY) surfaces here
  int rc = orbfe_device_status(s.ex_l);
  if (rc == ORBFE_OK) rc = orbfe_device_status(s.ex_r);
  return rc;
}

// The record gather of slot s (the path's only exchange) on a stream of its own, behind the slot's kernels: it
// overlaps the next chunk's kernels, and the slot's next submit waits for it.  Every rank calls it once per chunk, in chunk order.
extern "C" int orbfe_pipeline_gather(orbfe_pipeline* p, int slot, orbfe_gather* g, int mode, int32_t* d_n_all,
                                     orbfe_keypoint* d_kps_all, uint8_t* d_desc_all) {
  if (!p || !g || slot < 0 || slot >= (int)p->slots.size()) return ORBFE_ERR_INVALID;
  std::lock_guard<std::mutex> lk(p->mu);
  PCHK(hipSetDevice(p->device));
  Slot& s = p->slots[(size_t)slot];
  PCHK(hipStreamWaitEvent(p->s_gat, s.ev_done, 0));
  RCHK(orbfe_gather_records(g, reinterpret_cast<const int32_t*>(s.d_out + p->lay.n_left),
                            reinterpret_cast<const orbfe_keypoint*>(s.d_out + p->lay.kps), s.d_out + p->lay.desc, p->cfg.batch, p->cap,
                            mode, d_n_all, d_kps_all, d_desc_all, p->s_gat));
  PCHK(hipEventRecord(s.ev_gather, p->s_gat));
  s.gathering = true;
  return ORBFE_OK;
}

extern "C" int orbfe_pipeline_gather_wait(orbfe_pipeline* p, int slot) {
  if (!p || slot < 0 || slot >= (int)p->slots.size()) return ORBFE_ERR_INVALID;
  std::lock_guard<std::mutex> lk(p->mu);
  PCHK(hipSetDevice(p->device));
  Slot& s = p->slots[(size_t)slot];
  if (s.gathering) PCHK(hipEventSynchronize(s.ev_gather));
  s.gathering = false;
  return ORBFE_OK;
}
