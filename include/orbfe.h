/*
 * orbfe.h -- C ABI of liborbfe.so: the MI355X (gfx950) ORB front end.
 *
 * This is the drop-in boundary for ORB-SLAM2's per-frame feature front end.  The reference
 * (sjulier/Refactored_ORB_SLAM2) has no FFI for this path: the boundary there is two C++ classes,
 *   ORB_SLAM2::ORBextractor  (Source/Libraries/ORB_SLAM2/include/ORBextractor.h:43-104)
 *   ORB_SLAM2::ORBmatcher    (Source/Libraries/ORB_SLAM2/include/ORBmatcher.h:34-114)
 * and Frame's window query / stereo association (Source/Libraries/ORB_SLAM2/src/Frame.cc:250-263,
 * 341-410, 477-646).  The C++ classes shipped in refactored_orb_slam2_amd/csrc/host/ keep those exact
 * signatures and forward to the entry points below, so Tracking.cc / Frame.cc link unchanged
 * (INTEGRATION.md).  Everything here is plain C: opaque handles, POD structs, pointers and sizes; every
 * function returns ORBFE_OK (0) or a negative error code and never throws.  There is NO CPU fallback:
 * without a HIP device every compute entry point returns ORBFE_ERR_NO_DEVICE.
 *
 * L/ = Source/Libraries/ORB_SLAM2/ of the reference checkout.
 */
#ifndef ORBFE_H
#define ORBFE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORBFE_VERSION 100

enum {
  ORBFE_OK = 0,
  ORBFE_ERR_INVALID = -1,    /* bad argument (null pointer, unsupported size, bad parameter) */
  ORBFE_ERR_CAPACITY = -2,   /* an output buffer or an internal fixed-size table is too small */
  ORBFE_ERR_NO_DEVICE = -3,  /* no usable HIP device / HIP runtime failure at start-up */
  ORBFE_ERR_HIP = -4,        /* a HIP call failed; orbfe_last_error() has the text */
  ORBFE_ERR_EMPTY = -5,      /* empty input image: outputs untouched (L/src/ORBextractor.cc:981-982) */
  ORBFE_ERR_ALLOC = -6       /* a host memory allocation failed */
};

#define ORBFE_MAX_LEVELS 16
#define ORBFE_GRID_COLS 64   /* FRAME_GRID_COLS, L/include/Frame.h:37 */
#define ORBFE_GRID_ROWS 48   /* FRAME_GRID_ROWS, L/include/Frame.h:36 */
#define ORBFE_TH_HIGH 100    /* ORBmatcher::TH_HIGH, L/src/ORBmatcher.cc:38 */
#define ORBFE_TH_LOW 50      /* ORBmatcher::TH_LOW,  L/src/ORBmatcher.cc:39 */
#define ORBFE_HISTO_LENGTH 30

/* Constructor arguments of ORBextractor (L/src/ORBextractor.cc:407-410) */
typedef struct orbfe_params {
  int32_t n_features;
  float scale_factor;
  int32_t n_levels;
  int32_t ini_th_fast;
  int32_t min_th_fast;
} orbfe_params;

/* Layout-identical to cv::KeyPoint (pt.x, pt.y, size, angle, response, octave, class_id; 28 bytes) */
typedef struct orbfe_keypoint {
  float x, y;
  float size;
  float angle;
  float response;
  int32_t octave;
  int32_t class_id;
} orbfe_keypoint;

typedef struct orbfe_extractor orbfe_extractor;

const char* orbfe_last_error(void);   /* thread-local text of the last failure on this thread */
int orbfe_device_count(int* count);   /* number of visible HIP devices */
/* For hosts that do not link the HIP runtime themselves: the calling thread's current device (handles created with device = -1,
 * the per-thread matcher handle and the C++ drop-in classes use it), and plain device memory with synchronous copies (a C / C++
 * caller of the batched mode keeps its per-frame records in HBM for orbfe_gather_records). */
int orbfe_set_device(int device);
int orbfe_device_malloc(size_t bytes, void** out);
int orbfe_device_free(void* p);
int orbfe_device_upload(void* d_dst, const void* h_src, size_t bytes);
int orbfe_device_download(void* h_dst, const void* d_src, size_t bytes);

/* Hard limits (checked, ORBFE_ERR_INVALID beyond them):
 *   images              at most 4095 x 4095 pixels (keypoint coordinates travel between kernels as 12-bit fields) and at least
 *                       one FAST cell per level the caller wants keypoints from (a level narrower or lower than 2 x 16 + 30
 *                       pixels yields none, where the reference divides by zero, L/src/ORBextractor.cc:753-754)
 *   pyramid             n_levels <= ORBFE_MAX_LEVELS (16); FAST cells <= 66 x 66 pixels
 *   descriptor sets     fewer than 65 536 descriptors per frame / per set (orbfe_hamming_bf_device, orbfe_stereo_match*:
 *                       indices travel as 16-bit fields next to the distance)
 *   projection searches at most 9 500 keypoints per frame (orbfe_search_by_projection_*, orbfe_search_local_points*,
 *                       orbfe_proj_match_batch_device, orbfe_kf_search in its LOOP / RELOC modes: the ordered resolver keeps
 *                       nine bytes per keypoint in LDS); orbfe_proj_candidates / orbfe_proj_best alone take 65 535
 *   stereo matching     (image rows / 8, rounded up) x n_levels <= 8 192 row-bucket keys, rows <= 4 095
 *   inv_level_sigma2    orbfe_proj_best / orbfe_kf_search read n_levels floats (the caller states n_levels)
 * Threads: a handle serialises its own calls (internal mutex); different handles may be used from different threads at the same
 * time (Frame.cc:91-94 runs the two extractors on two threads).  The library holds no other mutable global state and reads no
 * environment variables.  The matcher entry points that take no handle (orbfe_search_*, orbfe_stereo_match, orbfe_kf_search,
 * ...) work on a handle the library creates per calling thread -- one HIP stream, scratch HBM and pinned staging that grow to
 * the largest call -- and keeps until the process ends: three for ORB-SLAM2's Tracking / LocalMapping / LoopClosing threads.
 * A thread that is about to end gives its handle back with orbfe_thread_release(). */
int orbfe_thread_release(void);   /* destroys the calling thread's implicit matcher handle, if it has one */

/* ------------------------------------------------------------------------------------- ORBextractor */
/* ORBextractor::ORBextractor (L/src/ORBextractor.cc:407-464).  device < 0 selects the current device. */
int orbfe_extractor_create(const orbfe_params* params, int device, orbfe_extractor** out);
int orbfe_extractor_destroy(orbfe_extractor* e);

/* Getters of L/include/ORBextractor.h:60-74; each writes n_levels floats. */
int orbfe_extractor_levels(const orbfe_extractor* e, int* n_levels);
int orbfe_extractor_scale_factors(const orbfe_extractor* e, float* out);
int orbfe_extractor_inv_scale_factors(const orbfe_extractor* e, float* out);
int orbfe_extractor_sigma2(const orbfe_extractor* e, float* out);
int orbfe_extractor_inv_sigma2(const orbfe_extractor* e, float* out);
int orbfe_extractor_features_per_level(const orbfe_extractor* e, int32_t* out);
/* Upper bound on keypoints per image: sum over levels of max(N_level + 3, 4 * nIni). */
int orbfe_extractor_max_keypoints(const orbfe_extractor* e, int w, int h, int* cap);

/* ORBextractor::operator() (L/src/ORBextractor.cc:978-1039) on ONE host image, synchronous.
 * img: CV_8UC1, h rows of w bytes, `stride` bytes between rows.  kps/desc: caller-owned host buffers of
 * `cap` entries (desc: cap x 32 bytes).  *n_out = number of keypoints.  Keypoints are in level order,
 * inside a level in DistributeOctTree list order, coordinates in level-0 pixels. */
int orbfe_extract(orbfe_extractor* e, const uint8_t* img, int w, int h, int stride, orbfe_keypoint* kps,
                  uint8_t* desc, int cap, int* n_out);

/* mvImagePyramid[level] of the last orbfe_extract call (L/include/ORBextractor.h:76; read by
 * Frame::ComputeStereoMatches, L/src/Frame.cc:483,567-589).  Copies the level (no border) to host. */
int orbfe_pyramid_level(orbfe_extractor* e, int level, uint8_t* dst, int dst_stride, int* w, int* h);
int orbfe_pyramid_level_size(const orbfe_extractor* e, int w0, int h0, int level, int* w, int* h);
/* All levels at once (one device synchronisation): dst[level] receives level `level` with row stride dst_stride[level]. */
int orbfe_pyramid_levels(orbfe_extractor* e, uint8_t* const* dst, const int* dst_stride);

/* Batched operator(): n_images host images of identical geometry, synchronous.  imgs[i] points to image i.
 * kps: n_images x cap, desc: n_images x cap x 32, n_out: n_images. */
int orbfe_extract_batch(orbfe_extractor* e, const uint8_t* const* imgs, int n_images, int w, int h, int stride,
                        orbfe_keypoint* kps, uint8_t* desc, int cap, int32_t* n_out);

/* Device-resident batch: all pointers are DEVICE pointers; asynchronous on `stream` (a hipStream_t, or
 * NULL for the handle's own stream).  d_imgs: image i at d_imgs + i*image_pitch, rows `stride` bytes
 * apart.  d_kps: n_images x cap, d_desc: n_images x cap x 32, d_n_out: n_images int32.  Work space is
 * (re)allocated when the geometry or batch size grows -- call once untimed before timing.
 * Level 0 in place: when d_imgs, stride and image_pitch are multiples of 16 and stride >= w rounded up to 16 (what
 * hipMemcpy2D into a pitched allocation or an image decoder delivers) the images ARE pyramid level 0 -- no copy.  They must
 * then stay unchanged until the results of this batch and every later read of its pyramid (orbfe_stereo_match_device,
 * orbfe_device_pyramid, orbfe_pyramid_level*) are complete.  Other layouts (tightly packed odd-width rows) are copied into
 * pitched planes first. */
int orbfe_extract_batch_device(orbfe_extractor* e, const uint8_t* d_imgs, int n_images, int w, int h, int stride,
                               size_t image_pitch, orbfe_keypoint* d_kps, uint8_t* d_desc, int cap,
                               int32_t* d_n_out, void* stream);
/* Device pointer + geometry of pyramid level `level` of image `image` of the last device batch.  Level 0 is row-major (rows
 * `pitch` bytes apart).  Levels >= 1 are row-major too when a stand-alone resize kernel wrote them, but the fused level chain (the
 * default) stores the levels it writes in TILES of 16 pixels x 8 rows = 128 bytes, tiles in raster order, pitch / 16 tiles per tile
 * row: pixel (x, y) at ((y >> 3) * (pitch >> 4) + (x >> 4)) * 128 + (y & 7) * 16 + (x & 15); orbfe_device_pyramid_layout says which.
 * orbfe_pyramid_level / orbfe_pyramid_levels always deliver row-major host copies. */
int orbfe_device_pyramid(const orbfe_extractor* e, int image, int level, const uint8_t** d_ptr, int* pitch,
                         int* w, int* h);
int orbfe_device_pyramid_layout(const orbfe_extractor* e, int level, int* tiled);   /* *tiled = 1: the 16 x 8 tiles described above */
int orbfe_sync(orbfe_extractor* e);   /* wait for the handle's stream(s) */
/* Waits for the handle's stream and returns ORBFE_ERR_CAPACITY if a kernel of an earlier (asynchronous)
 * batch flagged an internal table overflow; ORBFE_OK otherwise. */
int orbfe_device_status(orbfe_extractor* e);

/* Stage-level outputs of the last batch, for parity tests (host copies; synchronous).
 * candidates: FAST keypoints handed to DistributeOctTree (x, y relative to (16,16), score) in
 * vToDistributeKeys order; blurred: workingMat after GaussianBlur; level_keypoints: octree selection
 * before scaling (x, y in level coords, response). */
int orbfe_debug_candidates(orbfe_extractor* e, int image, int level, int32_t* x, int32_t* y, int32_t* score,
                           int cap, int* n);
int orbfe_debug_blurred(orbfe_extractor* e, int image, int level, uint8_t* dst, int dst_stride);
/* How pyramid and blur run from the next call on: 0 = the fused level chain (default: launch l blurs level l and writes level
 * l + 1 from the same staged windows), 1 = resize chain + the matrix-core blur (v_mfma_i32_16x16x64_i8 band products; DESIGN.md
 * lesson 31), 2 = resize chain + one LDS blur launch over all levels (rounds 1-4).  Same bytes; for parity tests and A/B timing. */
int orbfe_debug_blur_kernel(orbfe_extractor* e, int kind);
int orbfe_debug_pyramid(orbfe_extractor* e, int image, int level, uint8_t* dst, int dst_stride);
int orbfe_debug_level_keypoints(orbfe_extractor* e, int image, int level, int32_t* x, int32_t* y,
                                int32_t* score, int cap, int* n);

/* Per-stage HIP-event timing of the handle's stream.  enable != 0 records events around every kernel
 * stage of subsequent batches; orbfe_stage_times returns the accumulated milliseconds and launch counts
 * since the last reset.  Stage ids: */
enum {
  ORBFE_STAGE_PYRAMID = 0,
  ORBFE_STAGE_FAST = 1,
  ORBFE_STAGE_OCTREE = 2,
  ORBFE_STAGE_BLUR = 3,
  ORBFE_STAGE_DESCRIBE = 4,
  ORBFE_STAGE_COUNT = 5
};
int orbfe_profile_enable(orbfe_extractor* e, int enable);  /* 0 = off, 1 = every stage, otherwise a bit mask: bit (1 + stage) */
int orbfe_stage_times(orbfe_extractor* e, float* ms /*[ORBFE_STAGE_COUNT]*/, int32_t* launches, int reset);
/* The same events as intervals: for every stage launch timed since the last orbfe_stage_times / orbfe_stage_intervals call, its
 * stage index and the milliseconds from ref_event (a hipEvent_t the caller recorded earlier on the same device, timing enabled) to
 * the launch's first and last event.  Two handles that run side by side on two streams (left and right extractor) stretch each
 * other's launches; the union of their intervals is the time the chip spent on that kernel (bench.py's roofline).  The launches
 * also enter the totals of orbfe_stage_times.  At most cap intervals are written, *n receives their number. */
int orbfe_stage_intervals(orbfe_extractor* e, void* ref_event, int32_t* stage, float* start_ms, float* end_ms, int cap, int32_t* n);

/* --------------------------------------------------------------------------------------- ORBmatcher */
/* Work-space handle of the matcher kernels (scratch HBM + a HIP stream).  One handle serves one thread at
 * a time; the host-pointer entry points below that take no handle use a thread-local one, so they may be
 * called concurrently from Tracking / LocalMapping / LoopClosing threads (ORBmatcher objects are
 * stack-constructed per call site in the reference, L/src/Tracking.cc:781,1070). */
typedef struct orbfe_matcher orbfe_matcher;
int orbfe_matcher_create(int device, orbfe_matcher** out);
int orbfe_matcher_destroy(orbfe_matcher* m);
int orbfe_matcher_sync(orbfe_matcher* m);

/* ORBmatcher::DescriptorDistance (L/src/ORBmatcher.cc:1542-1556) for all pairs: dist[i*nB + j] =
 * Hamming(A[i], B[j]) as uint16.  DEVICE pointers (32-byte rows, 16-byte aligned), asynchronous on stream. */
int orbfe_hamming_matrix_device(const uint8_t* d_A, int nA, const uint8_t* d_B, int nB, uint16_t* d_dist,
                                void* stream);

/* Brute-force best / second-best (the inner loops of SearchByBoW, L/src/ORBmatcher.cc:201-222): for
 * every row i of A the first-minimum over j of B (strict <, index order) and the second-smallest
 * distance.  groupA/groupB (nullable, both or neither): compare only where groupA[i] == groupB[j]
 * (vocabulary node id); maskB (nullable): skip j with maskB[j] != 0 (already matched).  n_sets independent
 * problems: set s uses rows [s*strideA, s*strideA + nA[s]) of A and of out, [s*strideB, ..+nB[s]) of B;
 * max_nA >= every nA[s]; nB[s] < 65536.  All pointers DEVICE pointers; asynchronous on stream. */
typedef struct orbfe_bf_match {
  int32_t best_idx;    /* -1 when no candidate */
  int32_t best_dist;   /* 256 when none */
  int32_t second_dist; /* 256 when none */
} orbfe_bf_match;
int orbfe_hamming_bf_device(const uint8_t* d_A, const int32_t* d_nA, int strideA, int max_nA, const uint8_t* d_B,
                            const int32_t* d_nB, int strideB, const int32_t* d_groupA, const int32_t* d_groupB,
                            const uint8_t* d_maskB, int n_sets, orbfe_bf_match* d_out, void* stream);

/* View of the Frame members the projection searches read (L/include/Frame.h): mvKeysUn, mDescriptors,
 * mvuRight and the image bounds; the 64x48 grid (mGrid) is rebuilt on the device from them exactly as
 * Frame::AssignFeaturesToGrid / PosInGrid do (L/src/Frame.cc:250-263,399-410). */
typedef struct orbfe_frame_view {
  int32_t n;                        /* Frame::N */
  const orbfe_keypoint* keys_un;    /* mvKeysUn */
  const uint8_t* desc;              /* mDescriptors (n x 32) */
  const float* u_right;             /* mvuRight, nullable */
  float min_x, max_x, min_y, max_y; /* mnMinX, mnMaxX, mnMinY, mnMaxY */
} orbfe_frame_view;

/* One projected map point (A11: L/src/ORBmatcher.cc:52-71; A12: :1270-1308); 68 bytes */
typedef struct orbfe_query {
  float u, v;       /* projection */
  float u_r;        /* right-image coordinate of the projection */
  float radius;     /* window half-size, already multiplied by the level scale */
  int32_t min_level, max_level; /* GetFeaturesInArea level filter */
  int32_t valid;    /* 0 = the reference skips this point before the window query */
  int32_t blocks;   /* map point has Observations() > 0 */
  float angle;      /* keypoint angle (rotation histogram of A12) */
  uint8_t desc[32];
} orbfe_query;

typedef struct orbfe_cand {
  int32_t idx;   /* frame keypoint index */
  int32_t dist;  /* Hamming distance to the query descriptor */
} orbfe_cand;

/* Window query + distances, the data-parallel part of every SearchByProjection
 * (Frame::GetFeaturesInArea L/src/Frame.cc:341-397 + DescriptorDistance).  HOST pointers in, HOST
 * results out; synchronous.  For query q writes up to max_cand candidates in the reference's
 * enumeration order to cand[q*max_cand ..] and their total number to n_cand[q] (a count > max_cand
 * signals truncation).  The stereo gate |u_r - mvuRight| <= radius is applied on the device. */
int orbfe_proj_candidates(const orbfe_frame_view* frame, const orbfe_query* q, int nq, orbfe_cand* cand,
                          int32_t* n_cand, int max_cand);

/* SearchByProjection(Frame&, const vector<MapPoint*>&, th) (L/src/ORBmatcher.cc:45-128) on the device:
 * window query + distances in parallel, then the order-dependent assignment by one wave per frame.
 * blocked[idx] != 0 <=> F.mvpMapPoints[idx] has Observations() > 0 on entry (updated in place).
 * assigned[idx] = query index written to F.mvpMapPoints[idx]; entries not written keep their value.
 * *n_matches = return value of the reference.  HOST pointers, synchronous. */
int orbfe_search_by_projection_points(const orbfe_frame_view* frame, const orbfe_query* q, int nq, float nnratio,
                                      uint8_t* blocked, int32_t* assigned, int* n_matches);
/* SearchByProjection(Frame& cur, const Frame& last, th, bMono) (L/src/ORBmatcher.cc:1247-1383).  A slot the rotation check clears
 * (:1372 mvpMapPoints[...] = NULL) leaves with assigned = -1 and blocked = 0, as in the reference's frame. */
int orbfe_search_by_projection_frame(const orbfe_frame_view* cur, const orbfe_query* q, int nq,
                                     int check_orientation, uint8_t* blocked, int32_t* assigned, int* n_matches);

/* SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, const set<MapPoint*> &sAlreadyFound, th, ORBdist)
 * (L/src/ORBmatcher.cc:1385-1504), used by Tracking::Relocalization.  One query per keyframe map point that is not
 * bad and not in sAlreadyFound and that passes the projection / distance checks (:1408-1434): u, v, radius,
 * min_level = nPredictedLevel-1, max_level = nPredictedLevel+1, angle = pKF->mvKeysUn[i].angle, desc, valid = 1,
 * blocks = 1.  blocked[i2] on entry = CurrentFrame.mvpMapPoints[i2] != NULL (:1453).  No stereo gate (u_right of the
 * view is ignored); a match needs bestDist <= max_dist (:1466).  Outputs as orbfe_search_by_projection_frame. */
int orbfe_search_by_projection_keyframe(const orbfe_frame_view* cur, const orbfe_query* q, int nq, int check_orientation,
                                        int max_dist, uint8_t* blocked, int32_t* assigned, int* n_matches);

/* ---- Frame::isInFrustum + Tracking::SearchLocalPoints -------------------------------------------
 * The Frame members Frame::isInFrustum reads (L/src/Frame.cc:284-339). */
typedef struct orbfe_frustum {
  float Rcw[9], tcw[3], Ow[3];        /* mRcw (row-major), mtcw, mOw */
  float fx, fy, cx, cy, mbf;
  float min_x, max_x, min_y, max_y;   /* mnMinX, mnMaxX, mnMinY, mnMaxY */
  float log_scale_factor;             /* mfLogScaleFactor */
  int32_t n_levels;                   /* mnScaleLevels, 1..ORBFE_MAX_LEVELS (ORBFE_ERR_INVALID otherwise) */
  float scale_factors[ORBFE_MAX_LEVELS]; /* mvScaleFactors */
} orbfe_frustum;                      /* 168 bytes */

/* The MapPoint members the path reads (L/include/MapPoint.h); 72 bytes */
typedef struct orbfe_map_point {
  float pos[3], normal[3];            /* GetWorldPos(), GetNormal() */
  float min_distance, max_distance;   /* mfMinDistance, mfMaxDistance (un-scaled; 0.8 / 1.2 are applied as in MapPoint.cc:383-391) */
  int32_t skip;                       /* mnLastFrameSeen == mCurrentFrame.mnId || isBad()  (L/src/Tracking.cc:1057-1060) */
  int32_t observed;                   /* Observations() > 0 */
  uint8_t desc[32];                   /* GetDescriptor() */
} orbfe_map_point;

/* What isInFrustum leaves in the MapPoint (L/src/Frame.cc:329-335); 24 bytes.  in_view != 0 also means
 * pMP->IncreaseVisible() (L/src/Tracking.cc:1063). */
typedef struct orbfe_track {
  int32_t in_view;                    /* mbTrackInView */
  float proj_x, proj_y, proj_xr;      /* mTrackProjX, mTrackProjY, mTrackProjXR */
  int32_t level;                      /* mnTrackScaleLevel */
  float view_cos;                     /* mTrackViewCos */
} orbfe_track;

/* Second half of Tracking::SearchLocalPoints (L/src/Tracking.cc:1050-1078): isInFrustum(pMP, 0.5) for every local map
 * point, then ORBmatcher(nnratio).SearchByProjection(mCurrentFrame, mvpLocalMapPoints, th) (L/src/ORBmatcher.cc:45-128).
 * HOST pointers, synchronous.  track[i] is written for every point; blocked / assigned as in
 * orbfe_search_by_projection_points (assigned[idx] = index into mp[]); *n_to_match = nToMatch. */
int orbfe_search_local_points(const orbfe_frame_view* frame, const orbfe_frustum* frustum, const orbfe_map_point* mp,
                              int n_points, float th, float nnratio, orbfe_track* track, uint8_t* blocked,
                              int32_t* assigned, int* n_to_match, int* n_matches);

/* Batched, device-resident form: frame f owns keypoint rows [f*cap, f*cap + d_n[f]) and map points
 * d_points[f*p_cap .. f*p_cap + d_n_points[f]) with its own d_frustum[f].  The queries never leave HBM.  d_track
 * [n_frames][p_cap], d_blocked [n_frames][cap] (in/out), d_assigned [n_frames][cap] (in/out), d_n_to_match and
 * d_n_matches [n_frames].  Asynchronous on `stream` (NULL = the handle's stream). */
int orbfe_search_local_points_batch_device(orbfe_matcher* m, int n_frames, const orbfe_keypoint* d_kps,
                                           const uint8_t* d_desc, const int32_t* d_n, const float* d_u_right, int cap,
                                           float min_x, float max_x, float min_y, float max_y,
                                           const orbfe_frustum* d_frustum, const orbfe_map_point* d_points,
                                           const int32_t* d_n_points, int p_cap, float th, float nnratio,
                                           orbfe_track* d_track, uint8_t* d_blocked, int32_t* d_assigned,
                                           int32_t* d_n_to_match, int32_t* d_n_matches, void* stream);

/* ---- motion-model tracking on the device: Frame::UnprojectStereo (L/src/Frame.cc:668-679) and the projection part of
 * SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, th, bMono) (L/src/ORBmatcher.cc:1257-1308) ------------- */
typedef struct orbfe_unproject_cam {   /* the Frame members UnprojectStereo reads */
  float Rwc[9], Ow[3];                 /* mRwc (row-major), mOw */
  float cx, cy, invfx, invfy;
} orbfe_unproject_cam;                 /* 64 bytes */

typedef struct orbfe_last_point {      /* LastFrame keypoint i with its map point; 60 bytes */
  float pos[3];                        /* pMP->GetWorldPos() */
  int32_t valid;                       /* LastFrame.mvpMapPoints[i] != NULL && !LastFrame.mvbOutlier[i] */
  int32_t observed;                    /* pMP->Observations() > 0 */
  int32_t octave;                      /* LastFrame.mvKeys[i].octave */
  float angle;                         /* LastFrame.mvKeysUn[i].angle */
  uint8_t desc[32];                    /* pMP->GetDescriptor() */
} orbfe_last_point;

typedef struct orbfe_track_pose {      /* CurrentFrame members read by :1257-1308; 160 bytes */
  float Rcw[9], tcw[3];                /* CurrentFrame.mTcw */
  float fx, fy, cx, cy, mbf;
  float min_x, max_x, min_y, max_y;
  int32_t forward, backward;           /* bForward, bBackward (:1267-1268; tlc = Rlw*twc + tlw stays with the caller) */
  float th;
  float scale_factors[ORBFE_MAX_LEVELS]; /* CurrentFrame.mvScaleFactors */
} orbfe_track_pose;

/* One record per keypoint of every frame: map point = UnprojectStereo(i) when d_depth > 0 (valid = 0 otherwise), descriptor /
 * octave / angle of the keypoint itself (a map point created from this frame, L/src/MapPoint.cc:57-86).  DEVICE pointers,
 * asynchronous on `stream`.  d_points [n_frames][cap]. */
int orbfe_unproject_stereo_device(int n_frames, const orbfe_keypoint* d_kps, const uint8_t* d_desc, const int32_t* d_n,
                                  const float* d_depth, int cap, const orbfe_unproject_cam* d_cam, int observed,
                                  orbfe_last_point* d_points, void* stream);
/* Queries of frame f from the points of frame (f - frame_shift) mod n_frames (frame_shift = 1: the previous frame of a
 * sequence batch; 0: the caller arranged the points per frame): projection, invzc / bounds rejects, radius =
 * th * mvScaleFactors[octave], level range by bForward / bBackward, u_r = u - mbf * invzc.  d_nq[f] receives the point
 * count of the source frame.  Feed the result to orbfe_proj_match_batch_device(mode 1). */
int orbfe_track_queries_device(int n_frames, const orbfe_track_pose* d_pose, const orbfe_last_point* d_points,
                               const int32_t* d_n_points, int p_cap, int frame_shift, orbfe_query* d_queries,
                               int32_t* d_nq, void* stream);

/* The two calls above in one pass, for callers that need the stereo points only as the next frame's search queries (a tracking
 * loop that creates its temporal points from the stereo depth, L/src/Tracking.cc:877-936 UpdateLastFrame, then runs
 * SearchByProjection(cur, last)): queries of frame f from the KEYPOINTS of frame f - frame_shift -- UnprojectStereo with that frame's
 * camera (L/src/Frame.cc:668-679), then the projection with frame f's pose (L/src/ORBmatcher.cc:1270-1308) -- without the 60-byte
 * point record per keypoint going through memory.  Byte-equal to orbfe_unproject_stereo_device + orbfe_track_queries_device.
 * Frames in front of the batch (f < frame_shift): the carry frame when the five d_carry_* arrays are given (one frame: [cap]
 * keypoints / descriptors / depth, one count, one camera -- the last frame of the batch before), otherwise the batch's own tail
 * (index mod n_frames, as orbfe_track_queries_device).  d_nq[f] = keypoint count of the source frame. */
int orbfe_track_queries_stereo_device(int n_frames, const orbfe_keypoint* d_kps, const uint8_t* d_desc, const int32_t* d_n,
                                      const float* d_depth, int cap, const orbfe_unproject_cam* d_cam, int observed,
                                      const orbfe_keypoint* d_carry_kps, const uint8_t* d_carry_desc, const int32_t* d_carry_n,
                                      const float* d_carry_depth, const orbfe_unproject_cam* d_carry_cam,
                                      const orbfe_track_pose* d_pose, int frame_shift, orbfe_query* d_queries, int32_t* d_nq,
                                      void* stream);

/* SearchByBoW(KeyFrame*, Frame&, vector<MapPoint*>&) (L/src/ORBmatcher.cc:161-273), entirely on the device.
 * A DBoW2::FeatureVector is passed as its nodes sorted by id, each {node_id, start, count} into an index array
 * (nodesA/idxA = pKF->mFeatVec, nodesB/idxB = F.mFeatVec).  validA[i] != 0 <=> keyframe feature i has a map point
 * that is not bad.  matchB[j] = keyframe feature matched to frame feature j (the caller stores
 * vpMapPointsKF[matchB[j]] in vpMapPointMatches[j]) or -1; *n_matches = the reference's return value.
 * HOST pointers, synchronous. */
typedef struct orbfe_featvec_node { int32_t node_id, start, count; } orbfe_featvec_node;
int orbfe_search_by_bow(const uint8_t* descA, const float* angleA, const uint8_t* validA, int nA,
                        const orbfe_featvec_node* nodesA, int n_nodesA, const int32_t* idxA, const uint8_t* descB,
                        const float* angleB, int nB, const orbfe_featvec_node* nodesB, int n_nodesB,
                        const int32_t* idxB, float nnratio, int check_orientation, int32_t* matchB, int* n_matches);

/* SearchByBoW(KeyFrame*, KeyFrame*, vector<MapPoint*>&) (L/src/ORBmatcher.cc:494-612): both sides carry a validity mask
 * (map point present and not bad), the acceptance is bestDist < TH_LOW (strict), and the result is per feature of the
 * first keyframe: matchA[i] = index in the second keyframe (the caller stores vpMapPoints2[matchA[i]]) or -1. */
int orbfe_search_by_bow_kf(const uint8_t* descA, const float* angleA, const uint8_t* validA, int nA,
                           const orbfe_featvec_node* nodesA, int n_nodesA, const int32_t* idxA, const uint8_t* descB,
                           const float* angleB, const uint8_t* validB, int nB, const orbfe_featvec_node* nodesB,
                           int n_nodesB, const int32_t* idxB, float nnratio, int check_orientation, int32_t* matchA,
                           int* n_matches);

/* The candidate loop of the searches in which no query blocks another -- Fuse (L/src/ORBmatcher.cc:818-868), Fuse with Sim3
 * (:983-1009), SearchBySim3 (both directions, :1118-1147, 1194-1223): for every query the FIRST minimum-distance keypoint
 * of KeyFrame::GetFeaturesInArea(u, v, radius) (L/src/KeyFrame.cc:526-567) whose octave lies in [min_level, max_level]
 * (nPredictedLevel-1 .. nPredictedLevel).  gate: ORBFE_GATE_NONE, or ORBFE_GATE_FUSE_CHI2 = Fuse's reprojection test with
 * u_r = the projected right coordinate and inv_level_sigma2 = mvInvLevelSigma2, n_levels (<= ORBFE_MAX_LEVELS) floats
 * (e2 * invSigma2 > 7.8 stereo / 5.99 mono).
 * best_idx[q] = keypoint index or -1, best_dist[q] = its distance (256 if none): the caller applies TH_LOW / TH_HIGH and
 * the map bookkeeping (Replace / AddObservation / the mutual check) in query order.  HOST pointers, synchronous. */
enum { ORBFE_GATE_NONE = 1, ORBFE_GATE_FUSE_CHI2 = 2 };
int orbfe_proj_best(const orbfe_frame_view* keyframe, const orbfe_query* q, int nq, int gate, const float* inv_level_sigma2,
                    int n_levels, int32_t* best_idx, int32_t* best_dist);

/* ---- whole-function projection searches of the keyframe-rate callers ---------------------------------
 * Fuse (L/src/ORBmatcher.cc:766-907), Fuse with Sim3 (:909-1027), SearchBySim3 (:1029-1245, one direction per call),
 * SearchByProjection(KeyFrame*, Scw, ...) (:275-386) and SearchByProjection(Frame&, KeyFrame*, set, th, ORBdist)
 * (:1385-1504) share one shape: project every candidate map point into a camera, gate it (depth, image bounds, distance
 * range, viewing angle), predict its pyramid level, query the window and take the best descriptor.  orbfe_kf_search runs
 * all of that on the device; what stays with the caller is the pose algebra in front (decomposing Scw, composing
 * sR21 / t21: a handful of 3x3 products) and the map bookkeeping behind (Replace / AddObservation / AddMapPoint,
 * vpReplacePoint, the mutual check), which must replay in point order on the SLAM objects. */
enum {
  ORBFE_KF_FUSE = 1,       /* :766-907   float 1/z, KeyFrame::IsInImage, normal gate, chi-square gate, <= TH_LOW is the caller's */
  ORBFE_KF_FUSE_SIM3 = 2,  /* :909-1027  double 1.0/z, IsInImage, normal gate, no chi-square gate */
  ORBFE_KF_SIM3 = 3,       /* :1029-1245 second transform (sR21, t21), dist3D = |p3Dc2|, no normal gate */
  ORBFE_KF_LOOP = 4,       /* :275-386   float 1/z, IsInImage, normal gate; vpMatched blocks (sequential), accept <= max_dist */
  ORBFE_KF_RELOC = 5       /* :1385-1504 no depth gate, double 1.0/z, (fx*xc)*invzc, Frame bounds (< min || > max), levels
                              [l-1, l+1], occupied keypoints block, accept <= max_dist, rotation histogram */
};
typedef struct orbfe_kf_camera {
  float R[9], t[3];         /* p1 = R * p3Dw + t   (Rcw, tcw; SearchBySim3: R1w, t1w) */
  float R2[9], t2[3];       /* ORBFE_KF_SIM3 only: p2 = R2 * p1 + t2   (sR21, t21) */
  float Ow[3];              /* camera centre: dist3D = |p3Dw - Ow| (unused by ORBFE_KF_SIM3) */
  float fx, fy, cx, cy, mbf;
  float min_x, max_x, min_y, max_y;   /* mnMinX .. mnMaxY of the keyframe / frame searched in */
  float log_scale_factor;             /* mfLogScaleFactor */
  int32_t n_levels;                   /* mnScaleLevels, 1..ORBFE_MAX_LEVELS */
  float th;                           /* radius = th * mvScaleFactors[nPredictedLevel] */
  float scale_factors[ORBFE_MAX_LEVELS];
} orbfe_kf_camera;                    /* 220 bytes */
typedef struct orbfe_kf_point {       /* one candidate map point; 72 bytes */
  float pos[3], normal[3];            /* GetWorldPos(), GetNormal() */
  float min_distance, max_distance;   /* mfMinDistance, mfMaxDistance (un-scaled, see orbfe_map_point) */
  int32_t skip;                       /* the reference `continue`s before projecting (NULL, isBad(), already found ...) */
  float angle;                        /* ORBFE_KF_RELOC: pKF->mvKeysUn[i].angle for the rotation histogram */
  uint8_t desc[32];                   /* GetDescriptor() */
} orbfe_kf_point;
typedef struct orbfe_kf_result {      /* 24 bytes */
  int32_t best_idx;                   /* keypoint of the first minimum distance among the gated candidates, -1 = none.
                                         ORBFE_KF_LOOP / RELOC: the keypoint this point was ASSIGNED to (-1 = none / removed) */
  int32_t best_dist;                  /* its distance (256 = none); LOOP / RELOC: unspecified */
  int32_t level;                      /* nPredictedLevel, -1 when a gate rejected the point */
  float u, v, u_r;                    /* the projection (u_r = u - mbf * invz) */
} orbfe_kf_result;
/* keyframe = mvKeysUn / mDescriptors / mvuRight / bounds of the KeyFrame (Frame for ORBFE_KF_RELOC) searched in;
 * inv_level_sigma2 = mvInvLevelSigma2 (ORBFE_KF_FUSE, n_levels floats, else NULL).  blocked (LOOP: vpMatched[idx] != NULL;
 * RELOC: CurrentFrame.mvpMapPoints[idx] != NULL) is read and updated, NULL for the other modes.  check_orientation and
 * max_dist (TH_LOW / ORBdist) apply to LOOP / RELOC.  *n_matches: LOOP / RELOC = the reference's return value, otherwise the
 * number of points with a candidate.  HOST pointers, synchronous. */
int orbfe_kf_search(const orbfe_frame_view* keyframe, const float* inv_level_sigma2, const orbfe_kf_camera* cam,
                    const orbfe_kf_point* points, int n_points, int mode, int check_orientation, int max_dist,
                    uint8_t* blocked, orbfe_kf_result* results, int* n_matches);

/* SearchForTriangulation(KeyFrame *pKF1, KeyFrame *pKF2, cv::Mat F12, vMatchedPairs, bOnlyStereo) (L/src/ORBmatcher.cc:614-764)
 * with CheckDistEpipolarLine (:137-159).  The epipole (:622-630) is computed by the caller. */
typedef struct orbfe_epipolar {
  float F12[9];               /* row-major */
  float ex, ey;               /* epipole of pKF1's camera centre in pKF2's image */
  float scale_factors[ORBFE_MAX_LEVELS];  /* pKF2->mvScaleFactors */
  float level_sigma2[ORBFE_MAX_LEVELS];   /* pKF2->mvLevelSigma2 */
} orbfe_epipolar;             /* 172 bytes */
/* keys = mvKeysUn, u_right = mvuRight (NULL = monocular: all -1), has_mp[i] = (GetMapPoint(i) != NULL); FeatureVectors as in
 * orbfe_search_by_bow.  matchA[i] = index in pKF2 matched to feature i of pKF1, or -1 (vMatchedPairs = the pairs
 * (i, matchA[i]) in ascending i).  No match blocks another (vbMatched2 is never set in the reference); among the
 * candidates that pass the gates the smallest distance wins, the LAST one on ties (`dist > bestDist` rejects, :691). */
int orbfe_search_for_triangulation(const orbfe_keypoint* keysA, const uint8_t* descA, const float* u_rightA,
                                   const uint8_t* has_mpA, int nA, const orbfe_featvec_node* nodesA, int n_nodesA,
                                   const int32_t* idxA, const orbfe_keypoint* keysB, const uint8_t* descB,
                                   const float* u_rightB, const uint8_t* has_mpB, int nB, const orbfe_featvec_node* nodesB,
                                   int n_nodesB, const int32_t* idxB, const orbfe_epipolar* ep, int only_stereo,
                                   int check_orientation, int32_t* matchA, int* n_matches);

/* SearchForInitialization (L/src/ORBmatcher.cc:388-492), the monocular map-initialisation matcher: level-0
 * keypoints of F1 are searched in a window of `window_size` pixels around prev_matched_xy[2*i..2*i+1] in F2; a
 * closer later keypoint steals an earlier match (vMatchedDistance / vnMatches21).  matches12[i] = F2 index or -1;
 * prev_matched_xy is updated for the matched keypoints (:487-489).  HOST pointers, synchronous. */
int orbfe_search_for_initialization(const orbfe_frame_view* f1, const orbfe_frame_view* f2, float* prev_matched_xy,
                                    int window_size, float nnratio, int check_orientation, int32_t* matches12,
                                    int* n_matches);

/* Both searches for n_frames frames at once, DEVICE pointers, asynchronous on stream (NULL: the handle's).
 * Frame f: keypoints/descriptors/u_right rows [f*cap, f*cap + d_n[f]); queries [f*q_cap, f*q_cap + d_nq[f]).
 * mode 0 = A11 (points, ratio test nnratio), mode 1 = A12 (frame, rotation histogram if check_orientation).
 * d_blocked / d_assigned: n_frames x cap (in/out as above); d_n_matches: n_frames. */
int orbfe_proj_match_batch_device(orbfe_matcher* m, int n_frames, const orbfe_keypoint* d_kps, const uint8_t* d_desc,
                                  const int32_t* d_n, const float* d_u_right, int cap, float min_x, float max_x,
                                  float min_y, float max_y, const orbfe_query* d_q, const int32_t* d_nq, int q_cap,
                                  int mode, float nnratio, int check_orientation, uint8_t* d_blocked,
                                  int32_t* d_assigned, int32_t* d_n_matches, void* stream);

/* Frame::ComputeStereoMatches (L/src/Frame.cc:477-646) for n_pairs stereo frames, DEVICE pointers,
 * asynchronous.  Left/right keypoints+descriptors as produced by orbfe_extract_batch_device with the
 * two extractor handles, whose device pyramids (image p of each) are read for the 11x11 SAD refinement.
 * d_u_right / d_depth: n_pairs x cap floats (-1 = no match); d_n_matched: n_pairs (matches kept). */
int orbfe_stereo_match_device(orbfe_matcher* m, orbfe_extractor* left, orbfe_extractor* right, int n_pairs,
                              const orbfe_keypoint* d_kps_l, const uint8_t* d_desc_l, const int32_t* d_n_l,
                              const orbfe_keypoint* d_kps_r, const uint8_t* d_desc_r, const int32_t* d_n_r,
                              int cap, float mbf, float mb, float* d_u_right, float* d_depth, int32_t* d_n_matched,
                              void* stream);

/* Frame::ComputeStereoMatches (L/src/Frame.cc:477-646) for the ONE stereo pair the two extractors processed last -- the
 * per-frame calling pattern of Frame::Frame (L/src/Frame.cc:91-99: ExtractORB on two threads, then ComputeStereoMatches).
 * HOST pointers, synchronous.  kps / desc are what the two orbfe_extract calls returned (mvKeys, mDescriptors, mvKeysRight,
 * mDescriptorsRight); the row search, the descriptor match, the 11x11 SAD refinement on the pyramids still resident in HBM
 * (no mvImagePyramid download), the parabola and the median cut run on the device.  u_right / depth: n_l floats (mvuRight,
 * mvDepth; -1 = no match); *n_matched (optional) = matches kept.  mb = Frame::mb (minZ, :505), must be > 0.
 * Both extractors must have completed an extraction of the same image size on the calling thread's device. */
int orbfe_stereo_match(orbfe_extractor* left, orbfe_extractor* right, const orbfe_keypoint* kps_l, const uint8_t* desc_l,
                       int n_l, const orbfe_keypoint* kps_r, const uint8_t* desc_r, int n_r, float mbf, float mb,
                       float* u_right, float* depth, int* n_matched);

/* --------------------------------------------------------------------------------------- ORBVocabulary */
/* Frame::ComputeBoW (L/src/Frame.cc:412-417): ORBVocabulary::transform(descriptors, BowVector&, FeatureVector&, 4)
 * = DBoW2::TemplatedVocabulary::transform (Source/ThirdParty/DBoW2/DBoW2-local/include/DBoW2/
 * TemplatedVocabulary.h:1125-1257).  SURVEY.md §8(f) row 2: produces the FeatureVector SearchByBoW consumes. */
typedef struct orbfe_vocabulary orbfe_vocabulary;
/* Tree as arrays: node 0 is the root, parent[i] < i, children keep ascending id order; is_leaf[i] marks words
 * (word ids count leaves in id order); desc = n_nodes x 32 bytes; scoring / weighting = DBoW2::ScoringType /
 * WeightingType values (L1_NORM = 0, TF_IDF = 0). */
int orbfe_vocabulary_create(int k, int L, int scoring, int weighting, int n_nodes, const int32_t* parent,
                            const uint8_t* is_leaf, const uint8_t* desc, const double* weight, int device,
                            orbfe_vocabulary** out);
/* ORBVocabulary::loadFromTextFile (L/src/ORBVocabulary.cc:11-127), the ORBvoc.txt format. */
int orbfe_vocabulary_load_text(const char* path, int device, orbfe_vocabulary** out);
/* ORBVocabulary::loadFromBinaryFile (L/src/ORBVocabulary.cc:152-213): the format of saveToBinaryFile (:217-243), float
 * weights, including the duplicate of the last node the reference's eof loop appends. */
int orbfe_vocabulary_load_binary(const char* path, int device, orbfe_vocabulary** out);
int orbfe_vocabulary_destroy(orbfe_vocabulary* v);
int orbfe_vocabulary_info(const orbfe_vocabulary* v, int* k, int* L, int* n_nodes, int* n_words);
/* Tree descent only, DEVICE pointers, asynchronous: per descriptor the word id, the node at level L - levelsup and
 * the word weight.  A descent that ends in a leaf ABOVE level L - levelsup never reaches `*nid = final_id`
 * (TemplatedVocabulary.h:1249) and the reference's caller reads an uninitialised NodeId (:1149); this library reports node 0
 * (the root) for such a feature.  ORBvoc.txt with levelsup = 4 has no leaf above level 2, so the case does not arise there. */
int orbfe_bow_transform_device(orbfe_vocabulary* v, const uint8_t* d_desc, int n, int levelsup, int32_t* d_word,
                               int32_t* d_node, double* d_weight, void* stream);
/* The whole transform for one frame, HOST pointers, synchronous.  BowVector as ascending (bow_ids, bow_vals) pairs,
 * FeatureVector as ascending nodes {node_id, start, count} into fv_idx; every output array holds n entries at
 * most.  word_id / node_id / weight (per feature) are optional. */
int orbfe_compute_bow(orbfe_vocabulary* v, const uint8_t* desc, int n, int levelsup, int32_t* word_id,
                      int32_t* node_id, double* weight, int32_t* bow_ids, double* bow_vals, int* n_bow,
                      orbfe_featvec_node* fv_nodes, int32_t* fv_idx, int* n_fv_nodes);

/* ------------------------------------------------------------------------------- batched-sequence mode: record gather */
/* Independent frames shard over the GPUs of a node in contiguous chunks of the frame range (the reference
 * walks a sequence one frame at a time in one process, Source/Examples/Stereo/stereo_kitti.cc:88-106): one host thread or
 * process per GPU with its own extractor / matcher handles, no collective inside the step.  The only exchange is the gather
 * of the fixed-size padded records {n[frames]; keypoints[frames][cap]; descriptors[frames][cap][32]} over RCCL (librccl is
 * loaded on first use; ORBFE_ERR_NO_DEVICE without it):
 *   ORBFE_GATHER_ALL   ncclAllGather: every rank receives every rank's records, in rank order = frame order
 *   ORBFE_GATHER_ROOT  grouped ncclSend / ncclRecv: rank 0 alone receives them (the *_all pointers of other ranks may be NULL)
 * `frames` and `cap` are equal on every rank (orbfe_shard_range gives the chunk sizes; pad the short ones).  DEVICE pointers;
 * the collective is enqueued on `stream` (NULL: the handle's own stream, orbfe_gather_sync waits for it) and returns at once.
 * A handle belongs to one device; world = 1 is a valid communicator (the collective degenerates to a copy). */
typedef struct orbfe_gather orbfe_gather;
#define ORBFE_GATHER_ID_BYTES 128
enum { ORBFE_GATHER_ALL = 0, ORBFE_GATHER_ROOT = 1 };
/* [begin, end) of the frames rank `rank` owns: contiguous chunks whose sizes differ by at most one */
int orbfe_shard_range(int n_frames, int rank, int world, int* begin, int* end);
/* one process per GPU: rank 0 makes the id (ncclGetUniqueId) and hands it to the others out of band (MPI, a file, a socket) */
int orbfe_gather_unique_id(uint8_t id[ORBFE_GATHER_ID_BYTES]);
int orbfe_gather_create(const uint8_t* id, int rank, int world, int device, orbfe_gather** out);
/* one process, one host thread per GPU: n_devices handles at once (ncclCommInitAll); devices = NULL: 0 .. n_devices - 1 */
int orbfe_gather_create_all(int n_devices, const int* devices, orbfe_gather** out);
int orbfe_gather_destroy(orbfe_gather* g);
int orbfe_gather_rank(const orbfe_gather* g, int* rank, int* world);
int orbfe_gather_records(orbfe_gather* g, const int32_t* d_n, const orbfe_keypoint* d_kps, const uint8_t* d_desc, int frames,
                         int cap, int mode, int32_t* d_n_all, orbfe_keypoint* d_kps_all, uint8_t* d_desc_all, void* stream);
int orbfe_gather_sync(orbfe_gather* g);

/* ------------------------------------------------------------------------------------------------- warm-up */
/* The first call for an image size builds the plan and its device tables, allocates the work space and the pinned staging
 * buffers, loads the code objects; the third one- or two-image call captures the launch graph.  The reference constructs its
 * extractors once (L/src/Tracking.cc:112-127) and its first Track() already counts (initialisation): do that work when the
 * size is known instead of inside the first frames.  orbfe_extractor_prepare runs the host-API extraction of `n_images`
 * synthetic w x h images four times on handle e (n_images = 1 for ORBextractor::operator(), 2 for a stereo pair through
 * orbfe_extract_batch).  orbfe_frontend_prepare does that for both eyes (right may be NULL: monocular) and, ON THE CALLING
 * THREAD (the handle-less matcher entry points work on a per-thread handle), one stereo pair through orbfe_stereo_match and
 * the two SearchByProjection forms with max_queries queries (<= 0: one per keypoint).  Without them everything still happens
 * on first use.  HOST work only besides the calls themselves; synchronous. */
int orbfe_extractor_prepare(orbfe_extractor* e, int w, int h, int n_images);
int orbfe_frontend_prepare(orbfe_extractor* left, orbfe_extractor* right, int w, int h, int max_queries);

/* ------------------------------------------------------------------------------- batched stereo pipeline for a C / C++ host */
/* The batched-sequence mode (north_star) for a host that does not link the HIP runtime: the whole per-chunk step of
 * `examples/stereo_kitti.py` / `bench.py` -- images H2D, ORBextractor left + right (L/src/ORBextractor.cc:978-1039), Frame::
 * ComputeStereoMatches (L/src/Frame.cc:477-646), Frame::UnprojectStereo of every stereo point (:668-679), its projection into the
 * next frame and SearchByProjection(cur, last) (L/src/ORBmatcher.cc:1247-1383), results D2H -- behind one handle that owns the
 * extractors and matchers (one set per slot), `slots` sets of pinned host and device buffers and its streams -- copy in, left extractor,
 * right extractor (two streams, as the reference's two threads), the matching half (stereo match ... projection search), copy out; the
 * extraction of chunk k + 1 runs beside the matching half of chunk k (DESIGN lesson 47):
 *     orbfe_pipeline_input(p, s, &in)     pinned, pitched host images of slot s (+ the per-frame camera / pose records, pre-filled
 *                                         with the identity pose and the configured intrinsics); the host decodes into them
 *     orbfe_pipeline_submit(p, s, n, hp)  enqueues the chunk (n <= batch frames; hp = 0: frame 0 has no predecessor) and returns
 *     orbfe_pipeline_wait(p, s)           blocks until slot s's results are in its pinned output block
 *     orbfe_pipeline_output(p, s, &out)   the block: counts, keypoints, descriptors, mvuRight / mvDepth, tracked assignments
 * Chunks are processed in submit order; frame 0 of a chunk is searched with the points of the last frame of the previous one.  With
 * two slots the H2D copy of chunk k + 1 and the D2H copy of chunk k - 1 run beside the kernels of chunk k.  The left records of a
 * slot stay in HBM for orbfe_gather_records (orbfe_pipeline_device_records, enqueue it on orbfe_pipeline_stream) until the slot is
 * submitted again.  One host thread per handle at a time; one handle per GPU (orbfe_set_device before orbfe_pipeline_create, or
 * device >= 0). */
typedef struct orbfe_pipeline orbfe_pipeline;
typedef struct orbfe_pipeline_config {
  orbfe_params extractor;          /* both eyes */
  int32_t width, height;
  int32_t batch;                   /* stereo frames per chunk */
  int32_t slots;                   /* buffer sets, 1 .. 4 */
  float fx, fy, cx, cy, bf;        /* Camera.fx .. Camera.bf of the settings file (bf = baseline x fx) */
  float th;                        /* SearchByProjection window factor (7 for stereo, L/src/Tracking.cc:793-798) */
  int32_t check_orientation;       /* ORBmatcher(0.9, true) */
  int32_t output_mask;             /* which blocks of a chunk's results are copied to the slot's pinned host block: 0 = all of them, else
                                      an OR of ORBFE_PIPE_OUT_*; the per-frame counts (n_left, n_right, n_stereo, n_tracked) always are.
                                      A consumer that reads only the matches pays for 4 bytes per keypoint row instead of 72 (37 MB per
                                      256-frame KITTI chunk otherwise); everything stays available in HBM (orbfe_pipeline_device_records) */
} orbfe_pipeline_config;
#define ORBFE_PIPE_OUT_KEYPOINTS 1
#define ORBFE_PIPE_OUT_DESCRIPTORS 2
#define ORBFE_PIPE_OUT_STEREO 4        /* u_right, depth */
#define ORBFE_PIPE_OUT_ASSIGNED 8
#define ORBFE_PIPE_OUT_COUNTS 16       /* nothing but the counts */
typedef struct orbfe_pipeline_input_view {
  uint8_t* left; uint8_t* right;   /* frame f at + f * image_bytes, rows `pitch` bytes apart (pitch = width rounded up to 64) */
  int32_t pitch; size_t image_bytes;
  orbfe_unproject_cam* cams;       /* [batch]: Frame::UnprojectStereo's camera of frame f */
  orbfe_track_pose* poses;         /* [batch]: CurrentFrame members of frame f for the projection of frame f - 1's points */
} orbfe_pipeline_input_view;
typedef struct orbfe_pipeline_output_view {
  int32_t cap;                     /* keypoint rows per frame */
  const int32_t* n_left;           /* [batch] */
  const orbfe_keypoint* kps_left;  /* [batch][cap] */
  const uint8_t* desc_left;        /* [batch][cap][32] */
  const int32_t* n_right;          /* [batch] */
  const float* u_right;            /* [batch][cap] mvuRight (-1: none) */
  const float* depth;              /* [batch][cap] mvDepth */
  const int32_t* n_stereo;         /* [batch] */
  const int32_t* assigned;         /* [batch][cap]: index of the last frame's point matched to keypoint i, or -1 */
  const int32_t* n_tracked;        /* [batch]: SearchByProjection's return value */
} orbfe_pipeline_output_view;
int orbfe_pipeline_create(const orbfe_pipeline_config* cfg, int device, orbfe_pipeline** out);
int orbfe_pipeline_destroy(orbfe_pipeline* p);
int orbfe_pipeline_input(orbfe_pipeline* p, int slot, orbfe_pipeline_input_view* in);
int orbfe_pipeline_submit(orbfe_pipeline* p, int slot, int n_frames, int has_predecessor);
int orbfe_pipeline_wait(orbfe_pipeline* p, int slot);
int orbfe_pipeline_output(orbfe_pipeline* p, int slot, orbfe_pipeline_output_view* out);
/* For producers that already are on the device (a decoder on the GPU, frames from a neighbour over xGMI): the slot's DEVICE image
 * blocks (n-th image at + n * image_bytes, rows `pitch` bytes apart; write them on a stream of your own and synchronise it before the
 * submit, and not before orbfe_pipeline_wait of the slot's previous chunk), and a submit that leaves them as they are -- only the
 * camera / pose records go host to device.  Also what a host uses to run the handle at its kernels' rate on frames uploaded once. */
int orbfe_pipeline_device_input(orbfe_pipeline* p, int slot, uint8_t** d_left, uint8_t** d_right, int* pitch, size_t* image_bytes);
int orbfe_pipeline_submit_resident(orbfe_pipeline* p, int slot, int n_frames, int has_predecessor);
/* DEVICE pointers of slot s's left records ([batch] counts -- rows behind n_frames are zero --, [batch][cap] keypoints and
 * descriptors), valid from the slot's submit until its next submit, ordered on orbfe_pipeline_stream */
int orbfe_pipeline_device_records(orbfe_pipeline* p, int slot, const int32_t** d_n, const orbfe_keypoint** d_kps,
                                  const uint8_t** d_desc, int* cap);
void* orbfe_pipeline_stream(orbfe_pipeline* p);   /* the compute stream (a hipStream_t) */
/* A/B knob of tools/pipeline_rate.py: which stream -> priority level / creation order layout the NEXT orbfe_pipeline_create uses
 * (0 .. 4, see csrc/pipeline.cpp:pipeline_build; the default is the fastest measured).  Results do not depend on it. */
int orbfe_debug_pipeline_streams(int layout);
/* orbfe_gather_records of slot s's left records (`batch` frames of `cap` rows each, every rank alike) on a stream of the
 * pipeline's own, behind the slot's kernels: the collective overlaps the next chunk's kernels and the slot's next submit waits
 * for it.  Every rank calls it once per chunk, in chunk order (a rank whose shard has ended submits empty chunks).  The *_all
 * buffers are the caller's (orbfe_device_malloc); orbfe_pipeline_gather_wait blocks until they are filled. */
int orbfe_pipeline_gather(orbfe_pipeline* p, int slot, orbfe_gather* g, int mode, int32_t* d_n_all, orbfe_keypoint* d_kps_all,
                          uint8_t* d_desc_all);
int orbfe_pipeline_gather_wait(orbfe_pipeline* p, int slot);

/* --------------------------------------------------------------------------------------- sequence driver helpers */
/* Dependency-free PNG input for the dataset drivers (the reference reads with cv::imread(..., IMREAD_UNCHANGED),
 * Source/Examples/Stereo/stereo_kitti.cc:88-89, Source/Examples/RGB-D/rgbd_tum.cc, and converts colour frames in
 * Tracking::GrabImage*, L/src/Tracking.cc:164-178): 8-bit greyscale (+ alpha), or 8-bit RGB(A) converted with cvtColor's
 * RGB2GRAY weights; 16-bit greyscale (TUM depth maps) through orbfe_png_read_gray16; non-interlaced only; zlib only.
 * Images larger than 4095 x 4095 are refused (ORBFE_ERR_INVALID) before anything is allocated; no exception leaves these
 * functions (ORBFE_ERR_ALLOC when the decoder cannot get its memory).  Thread-safe: any number of calls may run at once. */
int orbfe_png_info(const char* path, int* w, int* h);
/* ... with the bit depth (8 / 16) and channel count (1 grey, 2 grey + alpha, 3 RGB, 4 RGBA); any output may be NULL */
int orbfe_png_info2(const char* path, int* w, int* h, int* depth, int* channels);
int orbfe_png_read_gray(const char* path, uint8_t* dst, int stride, int cap_rows, int* w, int* h);
/* Colour files and the settings file's Camera.RGB: the reference converts cv::imread's BGR data with RGB2GRAY when Camera.RGB is 1
 * -- every settings file it ships -- i.e. grey = 0.299 B + 0.587 G + 0.114 R, and with BGR2GRAY (the luminance) when it is 0
 * (L/src/Tracking.cc:164-178).  orbfe_png_read_gray is camera_rgb = 0; pass the file's flag here to get the reference's grey
 * levels (and so its keypoints) on colour input.  Grey files are unaffected. */
int orbfe_png_read_gray2(const char* path, uint8_t* dst, int stride, int cap_rows, int* w, int* h, int camera_rgb);
/* 16-bit greyscale: h rows of w uint16 samples (host byte order), `stride_elems` ELEMENTS apart */
int orbfe_png_read_gray16(const char* path, uint16_t* dst, int stride_elems, int cap_rows, int* w, int* h);

#ifdef __cplusplus
}
#endif
#endif /* ORBFE_H */
