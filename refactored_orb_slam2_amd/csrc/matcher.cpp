// matcher.cpp -- host side of liborbfe's ORBmatcher path: work-space handle, kernel sequencing, C ABI.
// No CPU fallback: everything that computes runs in match_kernels.hip.
//
// Reference behaviour (L/ = Source/Libraries/ORB_SLAM2/):
//   DescriptorDistance                         L/src/ORBmatcher.cc:1542-1556
//   SearchByProjection(Frame&, MapPoints)      L/src/ORBmatcher.cc:45-128
//   SearchByProjection(Frame& cur, last)       L/src/ORBmatcher.cc:1247-1383
//   SearchByBoW inner loops                    L/src/ORBmatcher.cc:201-222
//   Frame grid + GetFeaturesInArea             L/src/Frame.cc:250-263,341-410
//   Frame::ComputeStereoMatches                L/src/Frame.cc:477-646
#include <string.h>

#include <algorithm>
#include <mutex>
#include <vector>

#include "pipeline_internal.h"
#include "match_internal.h"

void orbfe_set_error(const char* fmt, ...);
int orbfe_internal_pyr_view(const orbfe_extractor* e, PyrView* v, int* n_images);
int orbfe_internal_tables(const orbfe_extractor* e, float* scale, float* inv_scale, int* n_levels, int* device);

#define HIPCHK(expr)                                                                              \
  do {                                                                                            \
    hipError_t _e = (expr);                                                                       \
    if (_e != hipSuccess) {                                                                       \
      orbfe_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return ORBFE_ERR_HIP;                                                                       \
    }                                                                                             \
  } while (0)

#define ORBFE_MAX_CAND 64  // stored candidates per query; longer lists are re-enumerated by the resolver

struct MBuf {
  void* p = nullptr;
  size_t bytes = 0;
};
static int mb_alloc(MBuf& b, size_t bytes) {
  if (b.p && bytes <= b.bytes) return ORBFE_OK;
  if (b.p) HIPCHK(hipFree(b.p));
  b.p = nullptr;
  b.bytes = 0;
  if (bytes < 256) bytes = 256;
  HIPCHK(hipMalloc(&b.p, bytes));
  b.bytes = bytes;
  return ORBFE_OK;
}

static int pin_alloc(void*& p, size_t& have, size_t bytes) {
  if (p && bytes <= have) return ORBFE_OK;
  if (p) HIPCHK(hipHostFree(p));
  p = nullptr;
  have = 0;
  bytes = (bytes + 65535) & ~(size_t)65535;
  HIPCHK(hipHostMalloc(&p, bytes, hipHostMallocDefault));
  have = bytes;
  return ORBFE_OK;
}
// lays parts out at 256-byte boundaries of a staging buffer
struct Layout {
  size_t off = 0;
  size_t add(size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; }
};
// Results of the one-frame host entry points (stereo association, projection searches) leave the device staging block for its
// pinned mirror by a copy kernel of four workgroups (pipeline_kernels.hip) instead of hipMemcpyAsync: the runtime's device-to-host
// path costs ~8 us more per call (ComputeStereoMatches 0.088 -> 0.080 ms, SearchByProjection(cur, last) 0.181 -> 0.179 per frame;
// the same kernel for the packed INPUT -- reads over the link -- measured no gain and stays a DMA copy).  Both blocks are 256-byte granular.
#ifndef HOST_D2H_KERNEL
#define HOST_D2H_KERNEL 1
#endif
static hipError_t packed_h2d(void* d, const void* h, size_t bytes, hipStream_t s) {
  return hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s);
}
static hipError_t packed_d2h(void* h, const void* d, size_t bytes, hipStream_t s) {
#if HOST_D2H_KERNEL
  orbfe_launch_copy_block(d, h, bytes, 4, s);
  return hipGetLastError();
#else
  return hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, s);
#endif
}

struct orbfe_matcher {
  int device = 0;
  hipStream_t stream = nullptr;
  // scratch
  MBuf cell_start, cell_idx, cell_rec, cand, n_cand, push_idx, push_bin, sad, bucket_start, bucket_idx;
  // staging for the host-pointer entry points
  MBuf h_keys, h_desc, h_ur, h_q, h_n, h_nq, h_blocked, h_assigned, h_nm;
  // SearchLocalPoints: generated queries (device) and staging of the host entry point
  MBuf lp_q, lp_pts, lp_fr, lp_track, lp_cnt;
  // the per-frame host-pointer entry points (orbfe_stereo_match, the SearchByProjection family): ONE packed upload and ONE packed
  // download per call through a pinned host mirror of a device staging buffer -- every hipMemcpyAsync costs 5-10 us of latency
  MBuf st_in, st_out;
  void* h_pin = nullptr;
  size_t h_pin_bytes = 0;
  std::mutex mu;
};

extern "C" int orbfe_matcher_create(int device, orbfe_matcher** out) {
  if (!out) return ORBFE_ERR_INVALID;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
    orbfe_set_error("no HIP device available (liborbfe has no CPU fallback)");
    return ORBFE_ERR_NO_DEVICE;
  }
  if (device < 0 && hipGetDevice(&device) != hipSuccess) device = 0;
  if (device >= ndev) return ORBFE_ERR_INVALID;
  HIPCHK(hipSetDevice(device));
  orbfe_matcher* m = new orbfe_matcher();
  m->device = device;
  hipError_t e = hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking);
  if (e != hipSuccess) {
    orbfe_set_error("hipStreamCreate: %s", hipGetErrorString(e));
    delete m;
    return ORBFE_ERR_HIP;
  }
  *out = m;
  return ORBFE_OK;
}

extern "C" int orbfe_matcher_destroy(orbfe_matcher* m) {
  if (!m) return ORBFE_OK;
  (void)hipSetDevice(m->device);
  if (m->stream) (void)hipStreamSynchronize(m->stream);
  MBuf* bufs[] = {&m->cell_start, &m->cell_idx, &m->cell_rec, &m->cand, &m->n_cand, &m->push_idx, &m->push_bin, &m->sad, &m->bucket_start, &m->bucket_idx, &m->h_keys,
                  &m->h_desc, &m->h_ur, &m->h_q, &m->h_n, &m->h_nq, &m->h_blocked, &m->h_assigned, &m->h_nm,
                  &m->lp_q, &m->lp_pts, &m->lp_fr, &m->lp_track, &m->lp_cnt, &m->st_in, &m->st_out};
  for (auto b : bufs)
    if (b->p) (void)hipFree(b->p);
  if (m->h_pin) (void)hipHostFree(m->h_pin);
  if (m->stream) (void)hipStreamDestroy(m->stream);
  delete m;
  return ORBFE_OK;
}

extern "C" int orbfe_matcher_sync(orbfe_matcher* m) {
  if (!m) return ORBFE_ERR_INVALID;
  HIPCHK(hipStreamSynchronize(m->stream));
  return ORBFE_OK;
}

static int launch_ok() {
  hipError_t le = hipGetLastError();
  if (le != hipSuccess) {
    orbfe_set_error("kernel launch failed: %s", hipGetErrorString(le));
    return ORBFE_ERR_HIP;
  }
  return ORBFE_OK;
}

// ------------------------------------------------------------------------------------------------ Hamming
extern "C" int orbfe_hamming_matrix_device(const uint8_t* d_A, int nA, const uint8_t* d_B, int nB, uint16_t* d_dist,
                                           void* stream) {
  if (!d_A || !d_B || !d_dist || nA < 0 || nB < 0) return ORBFE_ERR_INVALID;
  if (((uintptr_t)d_A & 15) || ((uintptr_t)d_B & 15)) {
    orbfe_set_error("descriptor matrices must be 16-byte aligned");
    return ORBFE_ERR_INVALID;
  }
  orbfe_launch_hamming_matrix(d_A, nA, d_B, nB, d_dist, (hipStream_t)stream);
  return launch_ok();
}

extern "C" int orbfe_hamming_bf_device(const uint8_t* d_A, const int32_t* d_nA, int strideA, int max_nA,
                                       const uint8_t* d_B, const int32_t* d_nB, int strideB, const int32_t* d_groupA,
                                       const int32_t* d_groupB, const uint8_t* d_maskB, int n_sets, orbfe_bf_match* d_out,
                                       void* stream) {
  if (!d_A || !d_B || !d_nA || !d_nB || !d_out || n_sets < 1 || max_nA < 0 || strideA < max_nA) return ORBFE_ERR_INVALID;
  if ((d_groupA == nullptr) != (d_groupB == nullptr) || strideB >= 65536) {
    orbfe_set_error("groupA/groupB must be given together; strideB must be < 65536");
    return ORBFE_ERR_INVALID;
  }
  if (((uintptr_t)d_A & 15) || ((uintptr_t)d_B & 15)) {
    orbfe_set_error("descriptor matrices must be 16-byte aligned");
    return ORBFE_ERR_INVALID;
  }
  HammingBfParams p{d_A, d_nA, strideA, d_B, d_nB, strideB, d_groupA, d_groupB, d_maskB, d_out};
  orbfe_launch_hamming_bf(p, max_nA, n_sets, (hipStream_t)stream);
  return launch_ok();
}

// ------------------------------------------------------------------------------------------------ projection
static int ensure_proj_scratch(orbfe_matcher* m, int n_frames, int cap, int q_cap) {
  int rc;
  const size_t F = (size_t)n_frames;
  if ((rc = mb_alloc(m->cell_start, F * (GRID_CELLS + 1) * sizeof(int32_t)))) return rc;
  if ((rc = mb_alloc(m->cell_idx, F * cap * sizeof(int32_t)))) return rc;
  if ((rc = mb_alloc(m->cell_rec, F * cap * 16))) return rc;
  if ((rc = mb_alloc(m->cand, F * q_cap * ORBFE_MAX_CAND * sizeof(orbfe_cand)))) return rc;
  if ((rc = mb_alloc(m->n_cand, F * q_cap * sizeof(int32_t)))) return rc;
  if ((rc = mb_alloc(m->push_idx, F * q_cap * sizeof(int32_t)))) return rc;
  if ((rc = mb_alloc(m->push_bin, F * q_cap))) return rc;
  return ORBFE_OK;
}

static void fill_frame_batch(orbfe_matcher* m, FrameBatch& fb, const orbfe_keypoint* d_kps, const uint8_t* d_desc,
                             const int32_t* d_n, const float* d_ur, int cap, float min_x, float max_x, float min_y,
                             float max_y) {
  fb.keys = d_kps;
  fb.desc = d_desc;
  fb.u_right = d_ur;
  fb.n = d_n;
  fb.cell_start = (int32_t*)m->cell_start.p;
  fb.cell_idx = (int32_t*)m->cell_idx.p;
  fb.cell_rec = (uint32_t*)m->cell_rec.p;
  fb.cap = cap;
  fb.min_x = min_x;
  fb.min_y = min_y;
  // mfGridElementWidthInv / HeightInv, L/src/Frame.cc:109-112
  fb.gw_inv = (float)ORBFE_GRID_COLS / (max_x - min_x);
  fb.gh_inv = (float)ORBFE_GRID_ROWS / (max_y - min_y);
}

static int proj_enqueue(orbfe_matcher* m, int n_frames, const orbfe_keypoint* d_kps, const uint8_t* d_desc,
                        const int32_t* d_n, const float* d_ur, int cap, float min_x, float max_x, float min_y,
                        float max_y, const orbfe_query* d_q, const int32_t* d_nq, int q_cap, int mode, float nnratio,
                        int check_ori, uint8_t* d_blocked, int32_t* d_assigned, int32_t* d_nm, bool resolve,
                        hipStream_t s, int th_high = ORBFE_TH_HIGH) {
  if (cap > 9500) {  // blocked[] + two claim buffers (9 bytes per keypoint) live in LDS next to the 64 KiB staging area
    orbfe_set_error("frame capacity %d too large for the LDS-resident resolver state (cap <= 9500)", cap);
    return ORBFE_ERR_INVALID;
  }
  if (((uintptr_t)d_desc & 15) || ((uintptr_t)d_q & 3) || ((uintptr_t)d_kps & 3)) {
    orbfe_set_error("descriptors must be 16-byte aligned, queries/keypoints 4-byte aligned");
    return ORBFE_ERR_INVALID;
  }
  int rc;
  if ((rc = ensure_proj_scratch(m, n_frames, cap, q_cap))) return rc;
  FrameBatch fb;
  fill_frame_batch(m, fb, d_kps, d_desc, d_n, d_ur, cap, min_x, max_x, min_y, max_y);
  QueryBatch qb{d_q, d_nq, q_cap};
  orbfe_launch_grid_build(fb, n_frames, s);
  orbfe_launch_proj_candidates(fb, qb, (orbfe_cand*)m->cand.p, (int32_t*)m->n_cand.p, ORBFE_MAX_CAND, n_frames, s);
  if (resolve)
    orbfe_launch_proj_resolve(fb, qb, (const orbfe_cand*)m->cand.p, (const int32_t*)m->n_cand.p, ORBFE_MAX_CAND, mode,
                              th_high, nnratio, check_ori, d_blocked, d_assigned, d_nm, (int32_t*)m->push_idx.p,
                              (uint8_t*)m->push_bin.p, n_frames, s);
  return launch_ok();
}

extern "C" int orbfe_proj_match_batch_device(orbfe_matcher* m, int n_frames, const orbfe_keypoint* d_kps,
                                             const uint8_t* d_desc, const int32_t* d_n, const float* d_u_right, int cap,
                                             float min_x, float max_x, float min_y, float max_y, const orbfe_query* d_q,
                                             const int32_t* d_nq, int q_cap, int mode, float nnratio,
                                             int check_orientation, uint8_t* d_blocked, int32_t* d_assigned,
                                             int32_t* d_n_matches, void* stream) {
  if (!m || !d_kps || !d_desc || !d_n || !d_q || !d_nq || !d_blocked || !d_assigned || !d_n_matches || n_frames < 1 ||
      cap < 1 || q_cap < 1 || (mode != 0 && mode != 1) || !(max_x > min_x) || !(max_y > min_y))
    return ORBFE_ERR_INVALID;
  std::lock_guard<std::mutex> lk(m->mu);
  HIPCHK(hipSetDevice(m->device));
  hipStream_t s = stream ? (hipStream_t)stream : m->stream;
  return proj_enqueue(m, n_frames, d_kps, d_desc, d_n, d_u_right, cap, min_x, max_x, min_y, max_y, d_q, d_nq, q_cap, mode,
                      nnratio, check_orientation, d_blocked, d_assigned, d_n_matches, true, s);
}

// Thread-local handle behind the handle-less host entry points (one HIP stream + scratch HBM + pinned staging per calling
// thread).  Not destroyed at thread exit: the HIP runtime may already be gone when thread_local destructors of the main thread
// run.  ORB-SLAM2's three long-lived threads (Tracking, LocalMapping, LoopClosing) hold three of them for the life of the
// process; a caller with transient threads gives the handle back with orbfe_thread_release() before the thread ends.
struct TlsMatcher {
  orbfe_matcher* m = nullptr;
};
static thread_local TlsMatcher t_tls_matcher;
static int tls_matcher(orbfe_matcher** out) {
  TlsMatcher& t = t_tls_matcher;
  if (!t.m) {
    int rc = orbfe_matcher_create(-1, &t.m);
    if (rc) return rc;
  }
  *out = t.m;
  return ORBFE_OK;
}
extern "C" int orbfe_thread_release(void) {
  TlsMatcher& t = t_tls_matcher;
  if (!t.m) return ORBFE_OK;
  orbfe_matcher* m = t.m;
  t.m = nullptr;
  return orbfe_matcher_destroy(m);
}

// uploads one host frame + queries; returns device pointers inside the handle's staging buffers
static int stage_host(orbfe_matcher* m, const orbfe_frame_view* f, const orbfe_query* q, int nq, hipStream_t s) {
  int rc;
  const int n = std::max(f->n, 1), nqq = std::max(nq, 1);
  if ((rc = mb_alloc(m->h_keys, sizeof(orbfe_keypoint) * n))) return rc;
  if ((rc = mb_alloc(m->h_desc, (size_t)32 * n))) return rc;
  if ((rc = mb_alloc(m->h_ur, sizeof(float) * n))) return rc;
  if ((rc = mb_alloc(m->h_q, sizeof(orbfe_query) * nqq))) return rc;
  if ((rc = mb_alloc(m->h_n, 16))) return rc;
  if ((rc = mb_alloc(m->h_nq, 16))) return rc;
  if ((rc = mb_alloc(m->h_blocked, (size_t)n + 16))) return rc;
  if ((rc = mb_alloc(m->h_assigned, sizeof(int32_t) * n))) return rc;
  if ((rc = mb_alloc(m->h_nm, 16))) return rc;
  if (f->n > 0) {
    HIPCHK(hipMemcpyAsync(m->h_keys.p, f->keys_un, sizeof(orbfe_keypoint) * f->n, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(m->h_desc.p, f->desc, (size_t)32 * f->n, hipMemcpyHostToDevice, s));
    if (f->u_right) HIPCHK(hipMemcpyAsync(m->h_ur.p, f->u_right, sizeof(float) * f->n, hipMemcpyHostToDevice, s));
  }
  if (nq > 0) HIPCHK(hipMemcpyAsync(m->h_q.p, q, sizeof(orbfe_query) * nq, hipMemcpyHostToDevice, s));
  int32_t nn = f->n, nnq = nq;
  HIPCHK(hipMemcpyAsync(m->h_n.p, &nn, 4, hipMemcpyHostToDevice, s));
  HIPCHK(hipMemcpyAsync(m->h_nq.p, &nnq, 4, hipMemcpyHostToDevice, s));
  HIPCHK(hipStreamSynchronize(s));  // nn / nnq live on this stack frame
  return ORBFE_OK;
}

static bool frame_ok(const orbfe_frame_view* f) {
  return f && f->n >= 0 && (f->n == 0 || (f->keys_un && f->desc)) && f->max_x > f->min_x && f->max_y > f->min_y;
}

extern "C" int orbfe_proj_candidates(const orbfe_frame_view* f, const orbfe_query* q, int nq, orbfe_cand* cand,
                                     int32_t* n_cand, int max_cand) {
  if (!frame_ok(f) || nq < 0 || (nq > 0 && (!q || !cand || !n_cand)) || max_cand < 1) return ORBFE_ERR_INVALID;
  if (nq == 0) return ORBFE_OK;
  orbfe_matcher* m;
  int rc;
  if ((rc = tls_matcher(&m))) return rc;
  std::lock_guard<std::mutex> lk(m->mu);
  HIPCHK(hipSetDevice(m->device));
  hipStream_t s = m->stream;
  if ((rc = stage_host(m, f, q, nq, s))) return rc;
  const int cap = std::max(f->n, 1);
  if ((rc = ensure_proj_scratch(m, 1, cap, nq))) return rc;
  // a caller-chosen max_cand needs its own candidate buffer size
  if ((rc = mb_alloc(m->cand, (size_t)nq * std::max(max_cand, ORBFE_MAX_CAND) * sizeof(orbfe_cand)))) return rc;
  FrameBatch fb;
  fill_frame_batch(m, fb, (const orbfe_keypoint*)m->h_keys.p, (const uint8_t*)m->h_desc.p, (const int32_t*)m->h_n.p,
                   f->u_right ? (const float*)m->h_ur.p : nullptr, cap, f->min_x, f->max_x, f->min_y, f->max_y);
  QueryBatch qb{(const orbfe_query*)m->h_q.p, (const int32_t*)m->h_nq.p, nq};
  orbfe_launch_grid_build(fb, 1, s);
  orbfe_launch_proj_candidates(fb, qb, (orbfe_cand*)m->cand.p, (int32_t*)m->n_cand.p, max_cand, 1, s);
  if ((rc = launch_ok())) return rc;
  HIPCHK(hipMemcpyAsync(cand, m->cand.p, sizeof(orbfe_cand) * (size_t)nq * max_cand, hipMemcpyDeviceToHost, s));
  HIPCHK(hipMemcpyAsync(n_cand, m->n_cand.p, sizeof(int32_t) * nq, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  // the kernel keeps the candidate's octave in bits 16..19 of dist for the resolver; the public field is the distance
  for (int i = 0; i < nq; i++)
    for (int c = 0; c < std::min(n_cand[i], max_cand); c++) cand[(size_t)i * max_cand + c].dist &= 0xffff;
  return ORBFE_OK;
}

static int search_host(const orbfe_frame_view* f, const orbfe_query* q, int nq, int mode, float nnratio, int check_ori,
                       uint8_t* blocked, int32_t* assigned, int* n_matches, int th_high = ORBFE_TH_HIGH,
                       bool stereo_gate = true) {
  if (!frame_ok(f) || nq < 0 || (nq > 0 && !q) || !blocked || !assigned || !n_matches) return ORBFE_ERR_INVALID;
  *n_matches = 0;
  if (nq == 0 || f->n == 0) return ORBFE_OK;
  orbfe_matcher* m;
  int rc;
  if ((rc = tls_matcher(&m))) return rc;
  std::lock_guard<std::mutex> lk(m->mu);
  HIPCHK(hipSetDevice(m->device));
  hipStream_t s = m->stream;
  // one packed upload [n, nq | keys | desc | mvuRight | queries | blocked | assigned | n_matches], the three kernels, one packed
  // download of the tail [blocked | assigned | n_matches]
  const size_t n = (size_t)f->n;
  const bool ur = f->u_right && stereo_gate;
  Layout L;
  const size_t o_hdr = L.add(16), o_keys = L.add(sizeof(orbfe_keypoint) * n), o_desc = L.add(32 * n);
  const size_t o_ur = L.add(ur ? sizeof(float) * n : 0), o_q = L.add(sizeof(orbfe_query) * (size_t)nq);
  const size_t o_out = L.off;
  const size_t o_blocked = L.add(n), o_assigned = L.add(sizeof(int32_t) * n), o_nm = L.add(16);
  if ((rc = pin_alloc(m->h_pin, m->h_pin_bytes, L.off)) || (rc = mb_alloc(m->st_in, L.off))) return rc;
  uint8_t* h = (uint8_t*)m->h_pin;
  uint8_t* d = (uint8_t*)m->st_in.p;
  ((int32_t*)(h + o_hdr))[0] = f->n;
  ((int32_t*)(h + o_hdr))[1] = nq;
  memcpy(h + o_keys, f->keys_un, sizeof(orbfe_keypoint) * n);
  memcpy(h + o_desc, f->desc, 32 * n);
  if (ur) memcpy(h + o_ur, f->u_right, sizeof(float) * n);
  memcpy(h + o_q, q, sizeof(orbfe_query) * (size_t)nq);
  memcpy(h + o_blocked, blocked, n);
  memcpy(h + o_assigned, assigned, sizeof(int32_t) * n);
  *(int32_t*)(h + o_nm) = 0;
  HIPCHK(packed_h2d(d, h, L.off, s));
  rc = proj_enqueue(m, 1, (const orbfe_keypoint*)(d + o_keys), d + o_desc, (const int32_t*)(d + o_hdr),
                    ur ? (const float*)(d + o_ur) : nullptr, f->n, f->min_x, f->max_x, f->min_y, f->max_y,
                    (const orbfe_query*)(d + o_q), (const int32_t*)(d + o_hdr) + 1, nq, mode, nnratio, check_ori, d + o_blocked,
                    (int32_t*)(d + o_assigned), (int32_t*)(d + o_nm), true, s, th_high);
  if (rc) { (void)hipStreamSynchronize(s); return rc; }
  HIPCHK(packed_d2h(h + o_out, d + o_out, L.off - o_out, s));
  HIPCHK(hipStreamSynchronize(s));
  memcpy(blocked, h + o_blocked, n);
  memcpy(assigned, h + o_assigned, sizeof(int32_t) * n);
  *n_matches = *(const int32_t*)(h + o_nm);
  return ORBFE_OK;
}

extern "C" int orbfe_search_by_projection_points(const orbfe_frame_view* f, const orbfe_query* q, int nq, float nnratio,
                                                 uint8_t* blocked, int32_t* assigned, int* n_matches) {
  return search_host(f, q, nq, 0, nnratio, 0, blocked, assigned, n_matches);
}
extern "C" int orbfe_search_by_projection_frame(const orbfe_frame_view* f, const orbfe_query* q, int nq,
                                                int check_orientation, uint8_t* blocked, int32_t* assigned,
                                                int* n_matches) {
  return search_host(f, q, nq, 1, 0.f, check_orientation, blocked, assigned, n_matches);
}

// ---- motion-model tracking: UnprojectStereo + the projection part of SearchByProjection(cur, last)
extern "C" int orbfe_unproject_stereo_device(int n_frames, const orbfe_keypoint* d_kps, const uint8_t* d_desc,
                                             const int32_t* d_n, const float* d_depth, int cap,
                                             const orbfe_unproject_cam* d_cam, int observed, orbfe_last_point* d_points,
                                             void* stream) {
  if (!d_kps || !d_desc || !d_n || !d_depth || !d_cam || !d_points || n_frames < 1 || cap < 1) return ORBFE_ERR_INVALID;
  if (((uintptr_t)d_desc & 15) || ((uintptr_t)d_kps & 3) || ((uintptr_t)d_points & 3) || ((uintptr_t)d_cam & 3)) {
    orbfe_set_error("descriptors must be 16-byte aligned, records 4-byte aligned");
    return ORBFE_ERR_INVALID;
  }
  orbfe_launch_unproject_stereo(d_kps, d_desc, d_n, d_depth, cap, d_cam, observed, d_points, n_frames, (hipStream_t)stream);
  return launch_ok();
}

extern "C" int orbfe_track_queries_device(int n_frames, const orbfe_track_pose* d_pose, const orbfe_last_point* d_points,
                                          const int32_t* d_n_points, int p_cap, int frame_shift, orbfe_query* d_queries,
                                          int32_t* d_nq, void* stream) {
  if (!d_pose || !d_points || !d_n_points || !d_queries || !d_nq || n_frames < 1 || p_cap < 1) return ORBFE_ERR_INVALID;
  if (((uintptr_t)d_pose & 3) || ((uintptr_t)d_points & 3) || ((uintptr_t)d_queries & 3)) return ORBFE_ERR_INVALID;
  orbfe_launch_track_queries(d_pose, d_points, d_n_points, p_cap, frame_shift, d_queries, d_nq, n_frames, (hipStream_t)stream);
  return launch_ok();
}

extern "C" int orbfe_track_queries_stereo_device(int n_frames, const orbfe_keypoint* d_kps, const uint8_t* d_desc, const int32_t* d_n,
                                                 const float* d_depth, int cap, const orbfe_unproject_cam* d_cam, int observed,
                                                 const orbfe_keypoint* d_carry_kps, const uint8_t* d_carry_desc,
                                                 const int32_t* d_carry_n, const float* d_carry_depth,
                                                 const orbfe_unproject_cam* d_carry_cam, const orbfe_track_pose* d_pose,
                                                 int frame_shift, orbfe_query* d_queries, int32_t* d_nq, void* stream) {
  if (!d_kps || !d_desc || !d_n || !d_depth || !d_cam || !d_pose || !d_queries || !d_nq || n_frames < 1 || cap < 1 || frame_shift < 0)
    return ORBFE_ERR_INVALID;
  const bool carry = d_carry_kps || d_carry_desc || d_carry_n || d_carry_depth || d_carry_cam;
  if (carry && !(d_carry_kps && d_carry_desc && d_carry_n && d_carry_depth && d_carry_cam)) {
    orbfe_set_error("the carry frame needs all five arrays (keypoints, descriptors, count, depth, camera) or none");
    return ORBFE_ERR_INVALID;
  }
  if (((uintptr_t)d_desc & 15) || ((uintptr_t)d_carry_desc & 15) || ((uintptr_t)d_kps & 3) || ((uintptr_t)d_carry_kps & 3) ||
      ((uintptr_t)d_cam & 3) || ((uintptr_t)d_carry_cam & 3) || ((uintptr_t)d_pose & 3) || ((uintptr_t)d_queries & 3)) {
    orbfe_set_error("descriptors must be 16-byte aligned, records 4-byte aligned");
    return ORBFE_ERR_INVALID;
  }
  orbfe_launch_track_queries_stereo(d_kps, d_desc, d_n, d_depth, cap, d_cam, observed, d_carry_kps, d_carry_desc, d_carry_n, d_carry_depth,
                                    d_carry_cam, d_pose, frame_shift, d_queries, d_nq, n_frames, (hipStream_t)stream);
  return launch_ok();
}

// ---- Tracking::SearchLocalPoints (L/src/Tracking.cc:1050-1078): isInFrustum -> queries (in HBM) -> SearchByProjection
static int local_points_enqueue(orbfe_matcher* m, int n_frames, const orbfe_keypoint* d_kps, const uint8_t* d_desc,
                                const int32_t* d_n, const float* d_ur, int cap, float min_x, float max_x, float min_y,
                                float max_y, const orbfe_frustum* d_fr, const orbfe_map_point* d_pts, const int32_t* d_np,
                                int p_cap, float th, float nnratio, orbfe_track* d_track, uint8_t* d_blocked,
                                int32_t* d_assigned, int32_t* d_ntm, int32_t* d_nm, hipStream_t s) {
  int rc;
  if ((rc = mb_alloc(m->lp_q, (size_t)n_frames * p_cap * sizeof(orbfe_query)))) return rc;
  HIPCHK(hipMemsetAsync(d_ntm, 0, sizeof(int32_t) * n_frames, s));
  orbfe_launch_frustum_queries(d_fr, d_pts, d_np, p_cap, th, 0.5f, d_track, (orbfe_query*)m->lp_q.p, d_ntm, n_frames, s);
  return proj_enqueue(m, n_frames, d_kps, d_desc, d_n, d_ur, cap, min_x, max_x, min_y, max_y, (const orbfe_query*)m->lp_q.p,
                      d_np, p_cap, 0, nnratio, 0, d_blocked, d_assigned, d_nm, true, s);
}

extern "C" int orbfe_search_local_points_batch_device(orbfe_matcher* m, int n_frames, const orbfe_keypoint* d_kps,
                                                      const uint8_t* d_desc, const int32_t* d_n, const float* d_u_right,
                                                      int cap, float min_x, float max_x, float min_y, float max_y,
                                                      const orbfe_frustum* d_frustum, const orbfe_map_point* d_points,
                                                      const int32_t* d_n_points, int p_cap, float th, float nnratio,
                                                      orbfe_track* d_track, uint8_t* d_blocked, int32_t* d_assigned,
                                                      int32_t* d_n_to_match, int32_t* d_n_matches, void* stream) {
  if (!m || !d_kps || !d_desc || !d_n || !d_frustum || !d_points || !d_n_points || !d_track || !d_blocked || !d_assigned ||
      !d_n_to_match || !d_n_matches || n_frames < 1 || cap < 1 || p_cap < 1 || !(max_x > min_x) || !(max_y > min_y))
    return ORBFE_ERR_INVALID;
  if (((uintptr_t)d_frustum & 3) || ((uintptr_t)d_points & 3) || ((uintptr_t)d_track & 3)) return ORBFE_ERR_INVALID;
  std::lock_guard<std::mutex> lk(m->mu);
  HIPCHK(hipSetDevice(m->device));
  hipStream_t s = stream ? (hipStream_t)stream : m->stream;
  return local_points_enqueue(m, n_frames, d_kps, d_desc, d_n, d_u_right, cap, min_x, max_x, min_y, max_y, d_frustum,
                              d_points, d_n_points, p_cap, th, nnratio, d_track, d_blocked, d_assigned, d_n_to_match,
                              d_n_matches, s);
}

extern "C" int orbfe_search_local_points(const orbfe_frame_view* f, const orbfe_frustum* fr, const orbfe_map_point* mp,
                                         int n_points, float th, float nnratio, orbfe_track* track, uint8_t* blocked,
                                         int32_t* assigned, int* n_to_match, int* n_matches) {
  if (!frame_ok(f) || !fr || n_points < 0 || (n_points > 0 && (!mp || !track)) || !blocked || !assigned || !n_to_match ||
      !n_matches || fr->n_levels < 1 || fr->n_levels > ORBFE_MAX_LEVELS)
    return ORBFE_ERR_INVALID;
  *n_to_match = 0;
  *n_matches = 0;
  if (n_points == 0) return ORBFE_OK;
  orbfe_matcher* m;
  int rc;
  if ((rc = tls_matcher(&m))) return rc;
  std::lock_guard<std::mutex> lk(m->mu);
  HIPCHK(hipSetDevice(m->device));
  hipStream_t s = m->stream;
  if ((rc = mb_alloc(m->lp_pts, sizeof(orbfe_map_point) * (size_t)n_points))) return rc;
  if ((rc = mb_alloc(m->lp_track, sizeof(orbfe_track) * (size_t)n_points))) return rc;
  if ((rc = mb_alloc(m->lp_fr, sizeof(orbfe_frustum)))) return rc;
  if ((rc = mb_alloc(m->lp_cnt, 16))) return rc;
  HIPCHK(hipMemcpyAsync(m->lp_pts.p, mp, sizeof(orbfe_map_point) * (size_t)n_points, hipMemcpyHostToDevice, s));
  HIPCHK(hipMemcpyAsync(m->lp_fr.p, fr, sizeof(orbfe_frustum), hipMemcpyHostToDevice, s));
  if ((rc = stage_host(m, f, nullptr, 0, s))) return rc;
  const int32_t np = n_points;
  HIPCHK(hipMemcpyAsync(m->h_nq.p, &np, 4, hipMemcpyHostToDevice, s));
  const int cap = std::max(f->n, 1);
  if (f->n > 0) {
    HIPCHK(hipMemcpyAsync(m->h_blocked.p, blocked, f->n, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(m->h_assigned.p, assigned, sizeof(int32_t) * f->n, hipMemcpyHostToDevice, s));
  }
  rc = local_points_enqueue(m, 1, (const orbfe_keypoint*)m->h_keys.p, (const uint8_t*)m->h_desc.p, (const int32_t*)m->h_n.p,
                            f->u_right ? (const float*)m->h_ur.p : nullptr, cap, f->min_x, f->max_x, f->min_y, f->max_y,
                            (const orbfe_frustum*)m->lp_fr.p, (const orbfe_map_point*)m->lp_pts.p, (const int32_t*)m->h_nq.p,
                            n_points, th, nnratio, (orbfe_track*)m->lp_track.p, (uint8_t*)m->h_blocked.p,
                            (int32_t*)m->h_assigned.p, (int32_t*)m->lp_cnt.p, (int32_t*)m->h_nm.p, s);
  if (rc) return rc;
  int32_t nm = 0, ntm = 0;
  HIPCHK(hipMemcpyAsync(track, m->lp_track.p, sizeof(orbfe_track) * (size_t)n_points, hipMemcpyDeviceToHost, s));
  if (f->n > 0) {
    HIPCHK(hipMemcpyAsync(blocked, m->h_blocked.p, f->n, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(assigned, m->h_assigned.p, sizeof(int32_t) * f->n, hipMemcpyDeviceToHost, s));
  }
  HIPCHK(hipMemcpyAsync(&nm, m->h_nm.p, 4, hipMemcpyDeviceToHost, s));
  HIPCHK(hipMemcpyAsync(&ntm, m->lp_cnt.p, 4, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  *n_matches = nm;
  *n_to_match = ntm;
  return ORBFE_OK;
}

// SearchByProjection(Frame&, KeyFrame*, const set<MapPoint*>&, th, ORBdist) (L/src/ORBmatcher.cc:1385-1504): the
// frame-to-frame walk without the stereo gate and with the caller's distance bound
extern "C" int orbfe_search_by_projection_keyframe(const orbfe_frame_view* f, const orbfe_query* q, int nq,
                                                   int check_orientation, int max_dist, uint8_t* blocked,
                                                   int32_t* assigned, int* n_matches) {
  if (max_dist < 0 || max_dist > 255) return ORBFE_ERR_INVALID;
  return search_host(f, q, nq, 1, 0.f, check_orientation, blocked, assigned, n_matches, max_dist, false);
}

// SearchByBoW(KeyFrame*, Frame&, vector<MapPoint*>&) (L/src/ORBmatcher.cc:161-273), host pointers, synchronous
static int bow_impl(const uint8_t* descA, const float* angleA, const uint8_t* validA, int nA,
                    const orbfe_featvec_node* nodesA, int n_nodesA, const int32_t* idxA, const uint8_t* descB,
                    const float* angleB, const uint8_t* validB, int nB, const orbfe_featvec_node* nodesB, int n_nodesB,
                    const int32_t* idxB, float nnratio, int check_orientation, int kf_mode, int32_t* matchA_out,
                    int32_t* matchB, int* n_matches) {
  if (nA < 0 || nB < 0 || n_nodesA < 0 || n_nodesB < 0 || !n_matches || (nB > 0 && !matchB)) return ORBFE_ERR_INVALID;
  *n_matches = 0;
  for (int j = 0; j < nB; j++) matchB[j] = -1;
  if (kf_mode)
    for (int i = 0; i < nA; i++) matchA_out[i] = -1;
  if (nA == 0 || nB == 0 || n_nodesA == 0 || n_nodesB == 0) return ORBFE_OK;
  if (!descA || !angleA || !validA || !nodesA || !idxA || !descB || !angleB || !nodesB || !idxB) return ORBFE_ERR_INVALID;
  // merge-join of the two FeatureVectors on NodeId (:183-253; lower_bound jumps == plain two-pointer walk on sorted ids)
  std::vector<BowPair> pairs;
  int ia = 0, ib = 0, totA = 0, totB = 0, maxB = 0;
  while (ia < n_nodesA && ib < n_nodesB) {
    if (nodesA[ia].node_id == nodesB[ib].node_id) {
      pairs.push_back(BowPair{nodesA[ia].start, nodesA[ia].count, nodesB[ib].start, nodesB[ib].count});
      maxB = std::max(maxB, nodesB[ib].count);
      ia++; ib++;
    } else if (nodesA[ia].node_id < nodesB[ib].node_id) ia++;
    else ib++;
  }
  for (int i = 0; i < n_nodesA; i++) totA = std::max(totA, nodesA[i].start + nodesA[i].count);
  for (int i = 0; i < n_nodesB; i++) totB = std::max(totB, nodesB[i].start + nodesB[i].count);
  if (pairs.empty()) return ORBFE_OK;
  if (maxB >= 65536 || maxB > 60000) return ORBFE_ERR_INVALID;
  // nodes are independent only if no frame feature is listed under two of them (always true for DBoW2 output)
  int sequential = 0;
  {
    std::vector<uint8_t> seen((size_t)nB, 0);
    for (const BowPair& pr : pairs) {   // a malformed FeatureVector must not index past the descriptor arrays on the device
      if (pr.startA < 0 || pr.countA < 0 || pr.startB < 0 || pr.countB < 0) return ORBFE_ERR_INVALID;
      for (int t = 0; t < pr.countA; t++)
        if (idxA[pr.startA + t] < 0 || idxA[pr.startA + t] >= nA) return ORBFE_ERR_INVALID;
      for (int t = 0; t < pr.countB; t++) {
        const int j = idxB[pr.startB + t];
        if (j < 0 || j >= nB) return ORBFE_ERR_INVALID;
        if (seen[j]) sequential = 1;
        seen[j] = 1;
      }
    }
  }
  orbfe_matcher* m;
  int rc;
  if ((rc = tls_matcher(&m))) return rc;
  std::lock_guard<std::mutex> lk(m->mu);
  HIPCHK(hipSetDevice(m->device));
  hipStream_t s = m->stream;
  // one packed upload: [pairs | descA | descB | angleA | angleB | idxA | idxB | validA]
  auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
  const size_t o_pairs = 0, o_dA = al(pairs.size() * sizeof(BowPair)), o_dB = o_dA + al((size_t)nA * 32),
               o_aA = o_dB + al((size_t)nB * 32), o_aB = o_aA + al((size_t)nA * 4), o_iA = o_aB + al((size_t)nB * 4),
               o_iB = o_iA + al((size_t)totA * 4), o_vA = o_iB + al((size_t)totB * 4), o_mB = o_vA + al((size_t)nA),
               o_cnt = o_mB + al((size_t)nB * 4), o_pi = o_cnt + 256, o_pb = o_pi + al((size_t)nA * 4),
               o_vB = o_pb + al((size_t)nA), o_mA = o_vB + al((size_t)nB), total = o_mA + al((size_t)nA * 4);
  if ((rc = mb_alloc(m->h_q, total))) return rc;
  uint8_t* d = (uint8_t*)m->h_q.p;
  HIPCHK(hipMemcpyAsync(d + o_pairs, pairs.data(), pairs.size() * sizeof(BowPair), hipMemcpyHostToDevice, s));
  HIPCHK(hipMemcpyAsync(d + o_dA, descA, (size_t)nA * 32, hipMemcpyHostToDevice, s));
  HIPCHK(hipMemcpyAsync(d + o_dB, descB, (size_t)nB * 32, hipMemcpyHostToDevice, s));
  HIPCHK(hipMemcpyAsync(d + o_aA, angleA, (size_t)nA * 4, hipMemcpyHostToDevice, s));
  HIPCHK(hipMemcpyAsync(d + o_aB, angleB, (size_t)nB * 4, hipMemcpyHostToDevice, s));
  HIPCHK(hipMemcpyAsync(d + o_iA, idxA, (size_t)totA * 4, hipMemcpyHostToDevice, s));
  HIPCHK(hipMemcpyAsync(d + o_iB, idxB, (size_t)totB * 4, hipMemcpyHostToDevice, s));
  HIPCHK(hipMemcpyAsync(d + o_vA, validA, (size_t)nA, hipMemcpyHostToDevice, s));
  HIPCHK(hipMemsetAsync(d + o_mB, 0xff, (size_t)nB * 4, s));
  HIPCHK(hipMemsetAsync(d + o_cnt, 0, 256, s));
  if (kf_mode) {
    HIPCHK(hipMemcpyAsync(d + o_vB, validB, (size_t)nB, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemsetAsync(d + o_mA, 0xff, (size_t)nA * 4, s));
  }
  BowParams p;
  p.pairs = (const BowPair*)(d + o_pairs);
  p.descA = d + o_dA; p.angleA = (const float*)(d + o_aA); p.validA = d + o_vA; p.idxA = (const int32_t*)(d + o_iA);
  p.descB = d + o_dB; p.angleB = (const float*)(d + o_aB); p.idxB = (const int32_t*)(d + o_iB);
  p.nnratio = nnratio; p.check_ori = check_orientation;
  p.sequential = sequential; p.n_pairs = (int)pairs.size();
  p.kf_mode = kf_mode; p.validB = kf_mode ? d + o_vB : nullptr; p.matchA = (int32_t*)(d + o_mA);
  p.matchB = (int32_t*)(d + o_mB); p.counters = (int32_t*)(d + o_cnt);
  p.push_idx = (int32_t*)(d + o_pi); p.push_bin = d + o_pb;
  orbfe_launch_bow(p, (int)pairs.size(), maxB, s);
  if ((rc = launch_ok())) return rc;
  int32_t cnt[2] = {0, 0};
  HIPCHK(hipMemcpyAsync(matchB, d + o_mB, (size_t)nB * 4, hipMemcpyDeviceToHost, s));
  if (kf_mode) HIPCHK(hipMemcpyAsync(matchA_out, d + o_mA, (size_t)nA * 4, hipMemcpyDeviceToHost, s));
  HIPCHK(hipMemcpyAsync(cnt, d + o_cnt, 8, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  *n_matches = cnt[1];
  return ORBFE_OK;
}

extern "C" int orbfe_search_by_bow(const uint8_t* descA, const float* angleA, const uint8_t* validA, int nA,
                                   const orbfe_featvec_node* nodesA, int n_nodesA, const int32_t* idxA,
                                   const uint8_t* descB, const float* angleB, int nB, const orbfe_featvec_node* nodesB,
                                   int n_nodesB, const int32_t* idxB, float nnratio, int check_orientation,
                                   int32_t* matchB, int* n_matches) {
  return bow_impl(descA, angleA, validA, nA, nodesA, n_nodesA, idxA, descB, angleB, nullptr, nB, nodesB, n_nodesB, idxB,
                  nnratio, check_orientation, 0, nullptr, matchB, n_matches);
}

// SearchByBoW(KeyFrame*, KeyFrame*, vector<MapPoint*>&) (L/src/ORBmatcher.cc:494-612)
extern "C" int orbfe_search_by_bow_kf(const uint8_t* descA, const float* angleA, const uint8_t* validA, int nA,
                                      const orbfe_featvec_node* nodesA, int n_nodesA, const int32_t* idxA,
                                      const uint8_t* descB, const float* angleB, const uint8_t* validB, int nB,
                                      const orbfe_featvec_node* nodesB, int n_nodesB, const int32_t* idxB, float nnratio,
                                      int check_orientation, int32_t* matchA, int* n_matches) {
  if (!matchA && nA > 0) return ORBFE_ERR_INVALID;
  if (nB > 0 && !validB) return ORBFE_ERR_INVALID;
  std::vector<int32_t> matchB((size_t)std::max(nB, 1));
  return bow_impl(descA, angleA, validA, nA, nodesA, n_nodesA, idxA, descB, angleB, validB, nB, nodesB, n_nodesB, idxB,
                  nnratio, check_orientation, 1, matchA, matchB.data(), n_matches);
}

// Independent arg-min searches (Fuse, Fuse(Sim3), SearchBySim3), host pointers, synchronous
extern "C" int orbfe_proj_best(const orbfe_frame_view* f, const orbfe_query* q, int nq, int gate, const float* inv_level_sigma2,
                               int n_levels, int32_t* best_idx, int32_t* best_dist) {
  if (!frame_ok(f) || nq < 0 || (nq > 0 && (!q || !best_idx || !best_dist)) || (gate != ORBFE_GATE_NONE && gate != ORBFE_GATE_FUSE_CHI2) ||
      (gate == ORBFE_GATE_FUSE_CHI2 && (!inv_level_sigma2 || n_levels < 1 || n_levels > ORBFE_MAX_LEVELS)))
    return ORBFE_ERR_INVALID;
  for (int i = 0; i < nq; i++) { best_idx[i] = -1; best_dist[i] = 256; }
  if (nq == 0 || f->n == 0) return ORBFE_OK;
  orbfe_matcher* m;
  int rc;
  if ((rc = tls_matcher(&m))) return rc;
  std::lock_guard<std::mutex> lk(m->mu);
  HIPCHK(hipSetDevice(m->device));
  hipStream_t s = m->stream;
  if ((rc = stage_host(m, f, q, nq, s))) return rc;
  if ((rc = ensure_proj_scratch(m, 1, f->n, nq))) return rc;
  if ((rc = mb_alloc(m->lp_cnt, 64))) return rc;
  float inv16[ORBFE_MAX_LEVELS] = {0};
  if (gate == ORBFE_GATE_FUSE_CHI2) {
    memcpy(inv16, inv_level_sigma2, sizeof(float) * n_levels);
    HIPCHK(hipMemcpyAsync(m->lp_cnt.p, inv16, sizeof(inv16), hipMemcpyHostToDevice, s));
  }
  FrameBatch fb;
  fill_frame_batch(m, fb, (const orbfe_keypoint*)m->h_keys.p, (const uint8_t*)m->h_desc.p, (const int32_t*)m->h_n.p,
                   f->u_right ? (const float*)m->h_ur.p : nullptr, f->n, f->min_x, f->max_x, f->min_y, f->max_y);
  QueryBatch qb{(const orbfe_query*)m->h_q.p, (const int32_t*)m->h_nq.p, nq};
  orbfe_launch_grid_build(fb, 1, s);
  // results land in the (otherwise unused here) n_cand / push_idx scratch: nq ints each
  orbfe_launch_proj_best(fb, qb, gate, (const float*)m->lp_cnt.p, (int32_t*)m->n_cand.p, (int32_t*)m->push_idx.p, 1, s);
  if ((rc = launch_ok())) return rc;
  HIPCHK(hipMemcpyAsync(best_idx, m->n_cand.p, sizeof(int32_t) * nq, hipMemcpyDeviceToHost, s));
  HIPCHK(hipMemcpyAsync(best_dist, m->push_idx.p, sizeof(int32_t) * nq, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  return ORBFE_OK;
}

// Fuse / Fuse(Sim3) / SearchBySim3 / SearchByProjection(KF,Scw) / SearchByProjection(Frame,KF,...) from the projection on
// (L/src/ORBmatcher.cc:766-1245, 275-386, 1385-1504): prologue kernel -> queries in HBM -> window search, host pointers
extern "C" int orbfe_kf_search(const orbfe_frame_view* f, const float* inv_level_sigma2, const orbfe_kf_camera* cam,
                               const orbfe_kf_point* points, int n_points, int mode, int check_orientation, int max_dist,
                               uint8_t* blocked, orbfe_kf_result* results, int* n_matches) {
  const bool seq = mode == ORBFE_KF_LOOP || mode == ORBFE_KF_RELOC;   // earlier points block later ones
  if (!frame_ok(f) || !cam || n_points < 0 || (n_points > 0 && (!points || !results)) || !n_matches || mode < ORBFE_KF_FUSE ||
      mode > ORBFE_KF_RELOC || cam->n_levels < 1 || cam->n_levels > ORBFE_MAX_LEVELS || (mode == ORBFE_KF_FUSE && !inv_level_sigma2) ||
      (seq && f->n > 0 && !blocked))
    return ORBFE_ERR_INVALID;
  *n_matches = 0;
  for (int i = 0; i < n_points; i++) {
    results[i].best_idx = -1; results[i].best_dist = 256; results[i].level = -1;
    results[i].u = results[i].v = results[i].u_r = 0.f;
  }
  if (n_points == 0) return ORBFE_OK;
  orbfe_matcher* m;
  int rc;
  if ((rc = tls_matcher(&m))) return rc;
  std::lock_guard<std::mutex> lk(m->mu);
  HIPCHK(hipSetDevice(m->device));
  hipStream_t s = m->stream;
  if ((rc = mb_alloc(m->lp_pts, sizeof(orbfe_kf_point) * (size_t)n_points))) return rc;
  if ((rc = mb_alloc(m->lp_track, sizeof(orbfe_kf_result) * (size_t)n_points))) return rc;
  if ((rc = mb_alloc(m->lp_fr, sizeof(orbfe_kf_camera)))) return rc;
  if ((rc = mb_alloc(m->lp_q, sizeof(orbfe_query) * (size_t)n_points))) return rc;
  if ((rc = mb_alloc(m->lp_cnt, 64))) return rc;
  HIPCHK(hipMemcpyAsync(m->lp_pts.p, points, sizeof(orbfe_kf_point) * (size_t)n_points, hipMemcpyHostToDevice, s));
  HIPCHK(hipMemcpyAsync(m->lp_fr.p, cam, sizeof(orbfe_kf_camera), hipMemcpyHostToDevice, s));
  if ((rc = stage_host(m, f, nullptr, 0, s))) return rc;
  const int32_t np = n_points;
  HIPCHK(hipMemcpyAsync(m->h_nq.p, &np, 4, hipMemcpyHostToDevice, s));
  orbfe_launch_kf_queries((const orbfe_kf_camera*)m->lp_fr.p, (const orbfe_kf_point*)m->lp_pts.p, n_points, mode,
                          (orbfe_query*)m->lp_q.p, (orbfe_kf_result*)m->lp_track.p, s);
  HIPCHK(hipMemcpyAsync(results, m->lp_track.p, sizeof(orbfe_kf_result) * (size_t)n_points, hipMemcpyDeviceToHost, s));
  if (f->n == 0) {
    HIPCHK(hipStreamSynchronize(s));
    return launch_ok();
  }
  const int cap = f->n;
  if (!seq) {
    if ((rc = ensure_proj_scratch(m, 1, cap, n_points))) return rc;
    float inv16[ORBFE_MAX_LEVELS] = {0};
    const int gate = mode == ORBFE_KF_FUSE ? ORBFE_GATE_FUSE_CHI2 : ORBFE_GATE_NONE;
    if (gate == ORBFE_GATE_FUSE_CHI2) {
      memcpy(inv16, inv_level_sigma2, sizeof(float) * cam->n_levels);
      HIPCHK(hipMemcpyAsync(m->lp_cnt.p, inv16, sizeof(inv16), hipMemcpyHostToDevice, s));
    }
    FrameBatch fb;
    fill_frame_batch(m, fb, (const orbfe_keypoint*)m->h_keys.p, (const uint8_t*)m->h_desc.p, (const int32_t*)m->h_n.p,
                     (f->u_right && gate == ORBFE_GATE_FUSE_CHI2) ? (const float*)m->h_ur.p : nullptr, cap, f->min_x, f->max_x,
                     f->min_y, f->max_y);
    QueryBatch qb{(const orbfe_query*)m->lp_q.p, (const int32_t*)m->h_nq.p, n_points};
    orbfe_launch_grid_build(fb, 1, s);
    orbfe_launch_proj_best(fb, qb, gate, (const float*)m->lp_cnt.p, (int32_t*)m->n_cand.p, (int32_t*)m->push_idx.p, 1, s);
    if ((rc = launch_ok())) return rc;
    std::vector<int32_t> bi((size_t)n_points), bd((size_t)n_points);
    HIPCHK(hipMemcpyAsync(bi.data(), m->n_cand.p, sizeof(int32_t) * n_points, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(bd.data(), m->push_idx.p, sizeof(int32_t) * n_points, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    int cnt = 0;
    for (int i = 0; i < n_points; i++) {
      results[i].best_idx = bi[i];
      results[i].best_dist = bd[i];
      cnt += bi[i] >= 0;
    }
    *n_matches = cnt;
    return ORBFE_OK;
  }
  // sequential modes: the greedy resolver of SearchByProjection(cur, last) with the caller's distance bound, no stereo gate
  std::vector<int32_t> assigned((size_t)cap, -2);
  HIPCHK(hipMemcpyAsync(m->h_blocked.p, blocked, cap, hipMemcpyHostToDevice, s));
  HIPCHK(hipMemcpyAsync(m->h_assigned.p, assigned.data(), sizeof(int32_t) * cap, hipMemcpyHostToDevice, s));
  rc = proj_enqueue(m, 1, (const orbfe_keypoint*)m->h_keys.p, (const uint8_t*)m->h_desc.p, (const int32_t*)m->h_n.p, nullptr, cap,
                    f->min_x, f->max_x, f->min_y, f->max_y, (const orbfe_query*)m->lp_q.p, (const int32_t*)m->h_nq.p, n_points, 1,
                    0.f, mode == ORBFE_KF_RELOC ? check_orientation : 0, (uint8_t*)m->h_blocked.p, (int32_t*)m->h_assigned.p,
                    (int32_t*)m->h_nm.p, true, s, max_dist);
  if (rc) return rc;
  int32_t nm = 0;
  HIPCHK(hipMemcpyAsync(blocked, m->h_blocked.p, cap, hipMemcpyDeviceToHost, s));
  HIPCHK(hipMemcpyAsync(assigned.data(), m->h_assigned.p, sizeof(int32_t) * cap, hipMemcpyDeviceToHost, s));
  HIPCHK(hipMemcpyAsync(&nm, m->h_nm.p, 4, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  for (int j = 0; j < cap; j++)
    if (assigned[j] >= 0 && assigned[j] < n_points) results[assigned[j]].best_idx = j;
  *n_matches = nm;
  return ORBFE_OK;
}

// SearchForTriangulation (L/src/ORBmatcher.cc:614-764), host pointers, synchronous
extern "C" int orbfe_search_for_triangulation(const orbfe_keypoint* keysA, const uint8_t* descA, const float* u_rightA,
                                              const uint8_t* has_mpA, int nA, const orbfe_featvec_node* nodesA, int n_nodesA,
                                              const int32_t* idxA, const orbfe_keypoint* keysB, const uint8_t* descB,
                                              const float* u_rightB, const uint8_t* has_mpB, int nB,
                                              const orbfe_featvec_node* nodesB, int n_nodesB, const int32_t* idxB,
                                              const orbfe_epipolar* ep, int only_stereo, int check_orientation,
                                              int32_t* matchA, int* n_matches) {
  if (nA < 0 || nB < 0 || n_nodesA < 0 || n_nodesB < 0 || !n_matches || (nA > 0 && !matchA) || !ep) return ORBFE_ERR_INVALID;
  *n_matches = 0;
  for (int i = 0; i < nA; i++) matchA[i] = -1;
  if (nA == 0 || nB == 0 || n_nodesA == 0 || n_nodesB == 0) return ORBFE_OK;
  if (!keysA || !descA || !has_mpA || !nodesA || !idxA || !keysB || !descB || !has_mpB || !nodesB || !idxB) return ORBFE_ERR_INVALID;
  std::vector<BowPair> pairs;
  int ia = 0, ib = 0, totA = 0, totB = 0;
  while (ia < n_nodesA && ib < n_nodesB) {
    if (nodesA[ia].node_id == nodesB[ib].node_id) {
      pairs.push_back(BowPair{nodesA[ia].start, nodesA[ia].count, nodesB[ib].start, nodesB[ib].count});
      if (nodesB[ib].count > 60000) return ORBFE_ERR_INVALID;
      ia++; ib++;
    } else if (nodesA[ia].node_id < nodesB[ib].node_id) ia++;
    else ib++;
  }
  for (int i = 0; i < n_nodesA; i++) totA = std::max(totA, nodesA[i].start + nodesA[i].count);
  for (int i = 0; i < n_nodesB; i++) totB = std::max(totB, nodesB[i].start + nodesB[i].count);
  if (pairs.empty()) return ORBFE_OK;
  // a pKF1 feature listed under two nodes (never produced by DBoW2) is visited twice by the reference: replay in order
  int sequential = 0;
  {
    std::vector<uint8_t> seen((size_t)nA, 0);
    for (const BowPair& pr : pairs) {
      if (pr.startA < 0 || pr.countA < 0 || pr.startB < 0 || pr.countB < 0) return ORBFE_ERR_INVALID;
      for (int t = 0; t < pr.countA; t++) {
        const int j = idxA[pr.startA + t];
        if (j < 0 || j >= nA) return ORBFE_ERR_INVALID;
        if (seen[j]) sequential = 1;
        seen[j] = 1;
      }
      for (int t = 0; t < pr.countB; t++)
        if (idxB[pr.startB + t] < 0 || idxB[pr.startB + t] >= nB) return ORBFE_ERR_INVALID;
    }
  }
  // candidate masks and stereo flags (:655-664, 677-686)
  std::vector<uint8_t> vA((size_t)nA), vB((size_t)nB), sA((size_t)nA), sB((size_t)nB);
  for (int i = 0; i < nA; i++) { sA[i] = u_rightA && u_rightA[i] >= 0; vA[i] = !has_mpA[i] && (!only_stereo || sA[i]); }
  for (int i = 0; i < nB; i++) { sB[i] = u_rightB && u_rightB[i] >= 0; vB[i] = !has_mpB[i] && (!only_stereo || sB[i]); }
  orbfe_matcher* m;
  int rc;
  if ((rc = tls_matcher(&m))) return rc;
  std::lock_guard<std::mutex> lk(m->mu);
  HIPCHK(hipSetDevice(m->device));
  hipStream_t s = m->stream;
  auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
  const size_t o_pairs = 0, o_dA = al(pairs.size() * sizeof(BowPair)), o_dB = o_dA + al((size_t)nA * 32),
               o_kA = o_dB + al((size_t)nB * 32), o_kB = o_kA + al((size_t)nA * sizeof(orbfe_keypoint)),
               o_iA = o_kB + al((size_t)nB * sizeof(orbfe_keypoint)), o_iB = o_iA + al((size_t)totA * 4),
               o_vA = o_iB + al((size_t)totB * 4), o_vB = o_vA + al((size_t)nA), o_sA = o_vB + al((size_t)nB),
               o_sB = o_sA + al((size_t)nA), o_mA = o_sB + al((size_t)nB), o_cnt = o_mA + al((size_t)nA * 4), o_pi = o_cnt + 256,
               o_pb = o_pi + al((size_t)std::max(totA, nA) * 4), total = o_pb + al((size_t)std::max(totA, nA));
  if ((rc = mb_alloc(m->h_q, total))) return rc;
  uint8_t* d = (uint8_t*)m->h_q.p;
  HIPCHK(hipMemcpyAsync(d + o_pairs, pairs.data(), pairs.size() * sizeof(BowPair), hipMemcpyHostToDevice, s));
  HIPCHK(hipMemcpyAsync(d + o_dA, descA, (size_t)nA * 32, hipMemcpyHostToDevice, s));
  HIPCHK(hipMemcpyAsync(d + o_dB, descB, (size_t)nB * 32, hipMemcpyHostToDevice, s));
  HIPCHK(hipMemcpyAsync(d + o_kA, keysA, (size_t)nA * sizeof(orbfe_keypoint), hipMemcpyHostToDevice, s));
  HIPCHK(hipMemcpyAsync(d + o_kB, keysB, (size_t)nB * sizeof(orbfe_keypoint), hipMemcpyHostToDevice, s));
  HIPCHK(hipMemcpyAsync(d + o_iA, idxA, (size_t)totA * 4, hipMemcpyHostToDevice, s));
  HIPCHK(hipMemcpyAsync(d + o_iB, idxB, (size_t)totB * 4, hipMemcpyHostToDevice, s));
  HIPCHK(hipMemcpyAsync(d + o_vA, vA.data(), (size_t)nA, hipMemcpyHostToDevice, s));
  HIPCHK(hipMemcpyAsync(d + o_vB, vB.data(), (size_t)nB, hipMemcpyHostToDevice, s));
  HIPCHK(hipMemcpyAsync(d + o_sA, sA.data(), (size_t)nA, hipMemcpyHostToDevice, s));
  HIPCHK(hipMemcpyAsync(d + o_sB, sB.data(), (size_t)nB, hipMemcpyHostToDevice, s));
  HIPCHK(hipMemsetAsync(d + o_mA, 0xff, (size_t)nA * 4, s));
  HIPCHK(hipMemsetAsync(d + o_cnt, 0, 256, s));
  TriParams t;
  memset(&t, 0, sizeof(t));
  t.b.pairs = (const BowPair*)(d + o_pairs);
  t.b.descA = d + o_dA; t.b.validA = d + o_vA; t.b.idxA = (const int32_t*)(d + o_iA);
  t.b.descB = d + o_dB; t.b.validB = d + o_vB; t.b.idxB = (const int32_t*)(d + o_iB);
  t.b.check_ori = check_orientation; t.b.sequential = sequential; t.b.n_pairs = (int)pairs.size();
  t.b.kf_mode = 1;   // bow_finish_kernel: rotation rejects clear matchA[push_idx]
  t.b.matchA = (int32_t*)(d + o_mA); t.b.matchB = nullptr;
  t.b.counters = (int32_t*)(d + o_cnt); t.b.push_idx = (int32_t*)(d + o_pi); t.b.push_bin = d + o_pb;
  t.keysA = (const orbfe_keypoint*)(d + o_kA); t.keysB = (const orbfe_keypoint*)(d + o_kB);
  t.stereoA = d + o_sA; t.stereoB = d + o_sB;
  t.ep = *ep;
  orbfe_launch_triangulation(t, (int)pairs.size(), s);
  if ((rc = launch_ok())) return rc;
  int32_t cnt[2] = {0, 0};
  HIPCHK(hipMemcpyAsync(matchA, d + o_mA, (size_t)nA * 4, hipMemcpyDeviceToHost, s));
  HIPCHK(hipMemcpyAsync(cnt, d + o_cnt, 8, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  *n_matches = cnt[1];
  return ORBFE_OK;
}

// SearchForInitialization (L/src/ORBmatcher.cc:388-492), host pointers, synchronous
extern "C" int orbfe_search_for_initialization(const orbfe_frame_view* f1, const orbfe_frame_view* f2, float* prev_matched_xy,
                                               int window_size, float nnratio, int check_orientation, int32_t* matches12,
                                               int* n_matches) {
  if (!frame_ok(f2) || !f1 || f1->n < 0 || (f1->n > 0 && (!f1->keys_un || !f1->desc || !prev_matched_xy || !matches12)) ||
      !n_matches || window_size < 0)
    return ORBFE_ERR_INVALID;
  *n_matches = 0;
  const int n1 = f1->n;
  for (int i = 0; i < n1; i++) matches12[i] = -1;
  if (n1 == 0 || f2->n == 0) return ORBFE_OK;
  // one query per F1 keypoint: window around vbPrevMatched[i1], level filter (level1, level1) (:403-410)
  std::vector<orbfe_query> q((size_t)n1);
  for (int i = 0; i < n1; i++) {
    orbfe_query& e = q[i];
    memset(&e, 0, sizeof(e));
    const int level1 = f1->keys_un[i].octave;
    if (level1 > 0) continue;
    e.u = prev_matched_xy[2 * i];
    e.v = prev_matched_xy[2 * i + 1];
    e.radius = (float)window_size;
    e.min_level = level1;
    e.max_level = level1;
    e.valid = 1;
    e.angle = f1->keys_un[i].angle;
    memcpy(e.desc, f1->desc + (size_t)i * 32, 32);
  }
  orbfe_matcher* m;
  int rc;
  if ((rc = tls_matcher(&m))) return rc;
  std::lock_guard<std::mutex> lk(m->mu);
  HIPCHK(hipSetDevice(m->device));
  hipStream_t s = m->stream;
  orbfe_frame_view f2n = *f2;
  f2n.u_right = nullptr;  // no stereo gate in this search
  if ((rc = stage_host(m, &f2n, q.data(), n1, s))) return rc;
  const int cap = f2->n;
  if ((rc = ensure_proj_scratch(m, 1, cap, n1))) return rc;
  if ((rc = mb_alloc(m->h_assigned, sizeof(int32_t) * (size_t)std::max(n1, cap)))) return rc;
  if ((rc = mb_alloc(m->h_ur, sizeof(float) * 2 * (size_t)std::max(n1, cap)))) return rc;  // prev_matched staging
  if ((size_t)cap * 8 > 60 * 1024) {
    orbfe_set_error("SearchForInitialization: frame with %d keypoints exceeds the LDS-resident tables", cap);
    return ORBFE_ERR_INVALID;
  }
  HIPCHK(hipMemcpyAsync(m->h_ur.p, prev_matched_xy, sizeof(float) * 2 * n1, hipMemcpyHostToDevice, s));
  FrameBatch fb;
  fill_frame_batch(m, fb, (const orbfe_keypoint*)m->h_keys.p, (const uint8_t*)m->h_desc.p, (const int32_t*)m->h_n.p, nullptr,
                   cap, f2->min_x, f2->max_x, f2->min_y, f2->max_y);
  QueryBatch qb{(const orbfe_query*)m->h_q.p, (const int32_t*)m->h_nq.p, n1};
  orbfe_launch_grid_build(fb, 1, s);
  orbfe_launch_proj_candidates(fb, qb, (orbfe_cand*)m->cand.p, (int32_t*)m->n_cand.p, ORBFE_MAX_CAND, 1, s);
  orbfe_launch_init_resolve(fb, qb, (const orbfe_cand*)m->cand.p, (const int32_t*)m->n_cand.p, ORBFE_MAX_CAND, nnratio,
                            check_orientation, (int32_t*)m->h_assigned.p, (float*)m->h_ur.p, (int32_t*)m->h_nm.p,
                            (int32_t*)m->push_idx.p, (uint8_t*)m->push_bin.p, s);
  if ((rc = launch_ok())) return rc;
  int32_t nm = 0;
  HIPCHK(hipMemcpyAsync(matches12, m->h_assigned.p, sizeof(int32_t) * n1, hipMemcpyDeviceToHost, s));
  HIPCHK(hipMemcpyAsync(prev_matched_xy, m->h_ur.p, sizeof(float) * 2 * n1, hipMemcpyDeviceToHost, s));
  HIPCHK(hipMemcpyAsync(&nm, m->h_nm.p, 4, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  *n_matches = nm;
  return ORBFE_OK;
}

// ------------------------------------------------------------------------------------------------ stereo
// the three stereo kernels for n_pairs pairs; the caller holds m->mu and has selected the device
static int stereo_enqueue(orbfe_matcher* m, orbfe_extractor* left, orbfe_extractor* right, int n_pairs,
                          const orbfe_keypoint* d_kps_l, const uint8_t* d_desc_l, const int32_t* d_n_l,
                          const orbfe_keypoint* d_kps_r, const uint8_t* d_desc_r, const int32_t* d_n_r, int cap, float mbf,
                          float mb, float* d_u_right, float* d_depth, int32_t* d_n_matched, hipStream_t stream) {
  StereoParams p;
  memset(&p, 0, sizeof(p));
  int nl = 0, nr = 0, il = 0, ir = 0, devl = 0, devr = 0;
  int rc;
  if ((rc = orbfe_internal_pyr_view(left, &p.pyrL, &il)) || (rc = orbfe_internal_pyr_view(right, &p.pyrR, &ir))) {
    orbfe_set_error("stereo match needs both extractors to have processed a device batch");
    return ORBFE_ERR_INVALID;
  }
  float sr[ORBFE_MAX_LEVELS], isr[ORBFE_MAX_LEVELS];
  orbfe_internal_tables(left, p.scale, p.inv_scale, &nl, &devl);
  orbfe_internal_tables(right, sr, isr, &nr, &devr);
  if (nl != nr || devl != m->device || devr != m->device || il < n_pairs || ir < n_pairs) {
    orbfe_set_error("stereo match: extractors must share levels/device and hold >= n_pairs images");
    return ORBFE_ERR_INVALID;
  }
  for (int l = 0; l < nl; l++)
    if (p.pyrL.w[l] != p.pyrR.w[l] || p.pyrL.h[l] != p.pyrR.h[l]) {
      orbfe_set_error("stereo match: left/right pyramids differ in size");
      return ORBFE_ERR_INVALID;
    }
  if ((rc = mb_alloc(m->sad, sizeof(int32_t) * (size_t)n_pairs * cap))) return rc;
  p.n_buckets = (p.pyrL.h[0] + 7) / 8;
  p.n_levels = nl;
  p.n_keys = p.n_buckets * nl;
  if (p.n_buckets > STEREO_MAX_BUCKETS || p.n_keys > STEREO_MAX_KEYS) return ORBFE_ERR_INVALID;
  if ((rc = mb_alloc(m->bucket_start, sizeof(int32_t) * (size_t)n_pairs * (p.n_keys + 1)))) return rc;
  if ((rc = mb_alloc(m->bucket_idx, 16 * (size_t)n_pairs * cap * STEREO_BUCKET_SPAN))) return rc;  // int4 records
  p.bucket_start = (int32_t*)m->bucket_start.p;
  p.bucket_idx = (int32_t*)m->bucket_idx.p;
  for (int l = 0; l < nl; l++)
    if (2.0f * p.scale[l] + 2.0f > 8.0f * (STEREO_BUCKET_SPAN - 1) / 2.0f) {
      orbfe_set_error("stereo match: pyramid scale %.2f too large for the row buckets", p.scale[l]);
      return ORBFE_ERR_INVALID;
    }
  p.kpsL = d_kps_l; p.descL = d_desc_l; p.nL = d_n_l;
  p.kpsR = d_kps_r; p.descR = d_desc_r; p.nR = d_n_r;
  p.cap = cap;
  p.mbf = mbf;
  p.maxD = mbf / mb;  // minZ = mb, maxD = mbf / minZ (L/src/Frame.cc:505-507)
  p.u_right = d_u_right;
  p.depth = d_depth;
  p.sad = (int32_t*)m->sad.p;
  p.n_matched = d_n_matched;
  orbfe_launch_stereo(p, n_pairs, stream);
  return launch_ok();
}

extern "C" int orbfe_stereo_match_device(orbfe_matcher* m, orbfe_extractor* left, orbfe_extractor* right, int n_pairs,
                                         const orbfe_keypoint* d_kps_l, const uint8_t* d_desc_l, const int32_t* d_n_l,
                                         const orbfe_keypoint* d_kps_r, const uint8_t* d_desc_r, const int32_t* d_n_r,
                                         int cap, float mbf, float mb, float* d_u_right, float* d_depth,
                                         int32_t* d_n_matched, void* stream) {
  if (!m || !left || !right || n_pairs < 1 || !d_kps_l || !d_desc_l || !d_n_l || !d_kps_r || !d_desc_r || !d_n_r ||
      cap < 1 || cap >= 65536 || !d_u_right || !d_depth || !d_n_matched || !(mb > 0))
    return ORBFE_ERR_INVALID;
  if (((uintptr_t)d_desc_l & 15) || ((uintptr_t)d_desc_r & 15)) {
    orbfe_set_error("descriptor matrices must be 16-byte aligned");
    return ORBFE_ERR_INVALID;
  }
  std::lock_guard<std::mutex> lk(m->mu);
  HIPCHK(hipSetDevice(m->device));
  return stereo_enqueue(m, left, right, n_pairs, d_kps_l, d_desc_l, d_n_l, d_kps_r, d_desc_r, d_n_r, cap, mbf, mb, d_u_right,
                        d_depth, d_n_matched, stream ? (hipStream_t)stream : m->stream);
}

// Frame::ComputeStereoMatches for the ONE pair the two extractors processed last (L/src/Frame.cc:91-99: the two ExtractORB
// threads, then ComputeStereoMatches): host keypoints / descriptors in, mvuRight / mvDepth out, the SAD refinement reads the
// pyramids that are still in HBM.  One packed upload, three kernels, one packed download on the calling thread's matcher.
extern "C" int orbfe_stereo_match(orbfe_extractor* left, orbfe_extractor* right, const orbfe_keypoint* kps_l,
                                  const uint8_t* desc_l, int n_l, const orbfe_keypoint* kps_r, const uint8_t* desc_r, int n_r,
                                  float mbf, float mb, float* u_right, float* depth, int* n_matched) {
  if (!left || !right || n_l < 0 || n_r < 0 || (n_l > 0 && (!kps_l || !desc_l || !u_right || !depth)) ||
      (n_r > 0 && (!kps_r || !desc_r)) || n_l >= 65536 || n_r >= 65536 || !(mb > 0))
    return ORBFE_ERR_INVALID;
  if (n_matched) *n_matched = 0;
  for (int i = 0; i < n_l; i++) u_right[i] = depth[i] = -1.0f;   // L/src/Frame.cc:478-479
  if (n_l == 0 || n_r == 0) return ORBFE_OK;
  orbfe_matcher* m;
  int rc;
  if ((rc = tls_matcher(&m))) return rc;
  std::lock_guard<std::mutex> lk(m->mu);
  HIPCHK(hipSetDevice(m->device));
  hipStream_t s = m->stream;
  const int cap = std::max(n_l, n_r);
  // one packed upload [n_l, n_r | keys_l | keys_r | desc_l | desc_r], three kernels, one packed download [u_right | depth | n]
  Layout L;
  const size_t o_hdr = L.add(16), o_kl = L.add(sizeof(orbfe_keypoint) * (size_t)cap), o_kr = L.add(sizeof(orbfe_keypoint) * (size_t)cap);
  const size_t o_dl = L.add((size_t)32 * cap), o_dr = L.add((size_t)32 * cap);
  const size_t o_out = L.off;
  const size_t o_ur = L.add(sizeof(float) * (size_t)cap), o_dp = L.add(sizeof(float) * (size_t)cap), o_nm = L.add(16);
  if ((rc = pin_alloc(m->h_pin, m->h_pin_bytes, L.off)) || (rc = mb_alloc(m->st_in, L.off))) return rc;
  uint8_t* h = (uint8_t*)m->h_pin;
  uint8_t* d = (uint8_t*)m->st_in.p;
  ((int32_t*)(h + o_hdr))[0] = n_l;
  ((int32_t*)(h + o_hdr))[1] = n_r;
  memcpy(h + o_kl, kps_l, sizeof(orbfe_keypoint) * (size_t)n_l);
  memcpy(h + o_kr, kps_r, sizeof(orbfe_keypoint) * (size_t)n_r);
  memcpy(h + o_dl, desc_l, (size_t)32 * n_l);
  memcpy(h + o_dr, desc_r, (size_t)32 * n_r);
  HIPCHK(packed_h2d(d, h, o_out, s));
  rc = stereo_enqueue(m, left, right, 1, (const orbfe_keypoint*)(d + o_kl), d + o_dl, (const int32_t*)(d + o_hdr),
                      (const orbfe_keypoint*)(d + o_kr), d + o_dr, (const int32_t*)(d + o_hdr) + 1, cap, mbf, mb, (float*)(d + o_ur),
                      (float*)(d + o_dp), (int32_t*)(d + o_nm), s);
  if (rc) { (void)hipStreamSynchronize(s); return rc; }
  HIPCHK(packed_d2h(h + o_out, d + o_out, L.off - o_out, s));
  HIPCHK(hipStreamSynchronize(s));
  memcpy(u_right, h + o_ur, sizeof(float) * (size_t)n_l);
  memcpy(depth, h + o_dp, sizeof(float) * (size_t)n_l);
  const int32_t nm = *(const int32_t*)(h + o_nm);
  if (n_matched) *n_matched = nm;
  return ORBFE_OK;
}
