/*
 * orb_oracle.h -- CPU restatement ("oracle") of the ORB-SLAM2 per-frame front end.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product path: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load liborboracle.so.  The product
 * (liborbfe.so, HIP) never links, loads or calls it.
 *
 * PARITY UNPINNED: the reference (sjulier/Refactored_ORB_SLAM2) ships no tests, fixtures or golden
 * vectors for this path and cannot be compiled in this image (needs OpenCV, which is neither vendored
 * nor installed).  The OpenCV primitives it calls (cv::FAST, cv::resize, cv::GaussianBlur,
 * cv::fastAtan2, cvRound) are restated here from the published OpenCV 4.5.x generic (non-IPP) algorithms
 * and cross-checked by an independent numpy restatement (tests/np_restatement.py), not by the real
 * reference binary.
 *
 * Every function cites the reference file:line it follows.  Abbreviation: L/ =
 * Source/Libraries/ORB_SLAM2/ of the reference checkout.
 */
#ifndef ORB_ORACLE_H
#define ORB_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* layout-identical to cv::KeyPoint (7 x 4 bytes) */
typedef struct oo_keypoint {
  float x, y;      /* pt */
  float size;
  float angle;     /* degrees [0,360) */
  float response;  /* FAST score */
  int32_t octave;
  int32_t class_id;
} oo_keypoint;

#define OO_MAX_LEVELS 16
#define OO_GRID_COLS 64 /* L/include/Frame.h:37 */
#define OO_GRID_ROWS 48 /* L/include/Frame.h:36 */
#define OO_TH_HIGH 100  /* L/src/ORBmatcher.cc:38 */
#define OO_TH_LOW 50    /* L/src/ORBmatcher.cc:39 */
#define OO_HISTO_LENGTH 30 /* L/src/ORBmatcher.cc:40 */

/* ---------------------------------------------------------------- primitives (OpenCV restatements) */
int oo_cvround(double v);                 /* P1: round half to even */
int oo_cvroundf(float v);
float oo_fast_atan2(float y, float x);    /* P5: cv::fastAtan2, degrees */
float oo_sinf(float x);                   /* restatement of glibc 2.35 sinf for |x| < 120 */
float oo_cosf(float x);
float oo_logf(float x);                   /* restatement of glibc 2.35 logf for positive normal x */
void oo_resize_linear_u8(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw, int dh,
                         int dstride);    /* P2 */
/* coefficient tables of P2 for one axis: ofs[d], coef[2*d..2*d+1] */
void oo_resize_tables(int s, int d, int32_t* ofs, int16_t* coef);
void oo_gauss_taps7(int taps[7]);         /* P3 fixed-point taps for ksize 7, sigma 2 */
void oo_gaussian_blur7_u8(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride); /* P3 */
void oo_copy_make_border_reflect101(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride,
                                    int border); /* P6 */
/* P4: cv::FAST(roi, th, nonmax=true), TYPE_9_16.  Returns number of keypoints written (x,y,score). */
int oo_fast9_16(const uint8_t* img, int stride, int cols, int rows, int threshold, int nonmax, int cap,
                int* out_x, int* out_y, int* out_score);
int oo_fast_corner_score(const uint8_t* p, int stride, int threshold);

/* ---------------------------------------------------------------- extractor (L/src/ORBextractor.cc) */
typedef struct oo_extractor oo_extractor;

oo_extractor* oo_extractor_create(int nfeatures, float scale_factor, int nlevels, int ini_th_fast,
                                  int min_th_fast);            /* ORBextractor.cc:407-464 */
void oo_extractor_destroy(oo_extractor* e);
int oo_extractor_levels(const oo_extractor* e);
const float* oo_extractor_scale_factors(const oo_extractor* e);     /* mvScaleFactor */
const float* oo_extractor_inv_scale_factors(const oo_extractor* e); /* mvInvScaleFactor */
const float* oo_extractor_sigma2(const oo_extractor* e);            /* mvLevelSigma2 */
const float* oo_extractor_inv_sigma2(const oo_extractor* e);        /* mvInvLevelSigma2 */
const int* oo_extractor_features_per_level(const oo_extractor* e);  /* mnFeaturesPerLevel */
const int* oo_extractor_umax(const oo_extractor* e);                /* umax[0..15] */
const int8_t* oo_pattern(void);                                     /* 1024 int8 */

/* operator(): ORBextractor.cc:978-1039.  Returns 0, or <0 on error (cap too small: -2).  Empty image:
 * returns 0 and leaves outputs untouched with *n_out = -1 (reference: silent return, :981-982). */
int oo_extract(oo_extractor* e, const uint8_t* img, int w, int h, int stride, oo_keypoint* kps, uint8_t* desc,
               int cap, int* n_out);

/* intermediate results of the last oo_extract call (for stage-level parity tests) */
int oo_level_size(const oo_extractor* e, int level, int* w, int* h);
const uint8_t* oo_level_pixels(const oo_extractor* e, int level, int* stride);     /* mvImagePyramid[level] */
const uint8_t* oo_level_blurred(const oo_extractor* e, int level, int* stride);    /* workingMat after blur */
/* FAST candidates handed to DistributeOctTree at this level (coords relative to (16,16)) */
int oo_level_candidates(const oo_extractor* e, int level, const int** x, const int** y, const int** score);
/* keypoints chosen by DistributeOctTree at this level, in list order (level coords incl. +16 border) */
int oo_level_keypoints(const oo_extractor* e, int level, const oo_keypoint** kps);

/* DistributeOctTree: ORBextractor.cc:529-731 (+DivideNode :475-527).  Inputs are candidate coordinates
 * relative to (minX,minY) and their responses, in vToDistributeKeys order.  Writes selected candidate
 * indices in list order; returns their count (<= n).  The reference's secondary sort key is a heap
 * pointer (allocator dependent); this restatement orders equal-size nodes by creation sequence, i.e.
 * behaves as a bump allocator would. */
int oo_distribute_octree(const int* x, const int* y, const int* score, int n, int minX, int maxX, int minY,
                         int maxY, int N, int* out_idx);

float oo_ic_angle(const uint8_t* img, int stride, int x, int y, const int* umax);   /* :76-100 */
void oo_orb_descriptor(const uint8_t* blurred, int stride, int x, int y, float angle_deg,
                       uint8_t desc[32]);                                          /* :103-146 */

/* ---------------------------------------------------------------- matcher (L/src/ORBmatcher.cc, Frame.cc) */
int oo_descriptor_distance(const uint8_t* a, const uint8_t* b); /* ORBmatcher.cc:1542-1556 */

typedef struct oo_frame {
  int n;                      /* Frame::N */
  const oo_keypoint* keys_un; /* mvKeysUn */
  const uint8_t* desc;        /* mDescriptors, n x 32 */
  const float* u_right;       /* mvuRight (may be NULL = all -1) */
  float min_x, max_x, min_y, max_y;       /* mnMinX.. */
  float grid_w_inv, grid_h_inv;           /* mfGridElementWidthInv/HeightInv */
  int n_levels;
  const float* scale_factors;             /* mvScaleFactors */
  /* grid in CSR form, cell = ix * OO_GRID_ROWS + iy (mGrid[ix][iy]); filled by oo_frame_build_grid */
  int32_t cell_start[OO_GRID_COLS * OO_GRID_ROWS + 1];
  int32_t* cell_idx;          /* n entries, caller-allocated */
} oo_frame;

void oo_frame_build_grid(oo_frame* f);  /* Frame.cc:250-263, 399-410 */
/* Frame.cc:341-397.  Returns count, indices in reference enumeration order. */
int oo_features_in_area(const oo_frame* f, float x, float y, float r, int min_level, int max_level,
                        int32_t* out_idx);

/* One projected map point / last-frame point, as the searches consume it */
typedef struct oo_query {
  float u, v;          /* projection (mTrackProjX/Y or u,v) */
  float u_r;           /* mTrackProjXR or u - mbf*invzc */
  float radius;        /* search radius already scaled (r*scale[level] or th*scale[octave]) */
  int32_t min_level, max_level;
  int32_t valid;       /* 0: skipped by the reference before the window query */
  int32_t blocks;      /* 1 if the map point has Observations()>0 (an assignment of it blocks later queries) */
  float angle;         /* keypoint angle for the rotation histogram (A12) */
  uint8_t desc[32];
} oo_query;

/* SearchByProjection(Frame&, vector<MapPoint*>&, th): ORBmatcher.cc:45-128.
 * blocked[idx] != 0 <=> F.mvpMapPoints[idx] has Observations()>0 on entry; updated in place.
 * assigned[idx] = query index written to F.mvpMapPoints[idx] (or left unchanged).  Returns nmatches. */
int oo_search_by_projection_points(const oo_frame* f, const oo_query* q, int nq, float nnratio, uint8_t* blocked,
                                   int32_t* assigned);
/* ---- Frame::isInFrustum + Tracking::SearchLocalPoints */
typedef struct oo_frustum {      /* the Frame members isInFrustum reads (L/include/Frame.h) */
  float Rcw[9], tcw[3], Ow[3];   /* mRcw (row-major), mtcw, mOw */
  float fx, fy, cx, cy, mbf;
  float min_x, max_x, min_y, max_y;   /* mnMinX .. mnMaxY */
  float log_scale_factor;        /* mfLogScaleFactor */
  int32_t n_levels;              /* mnScaleLevels */
  float scale_factors[OO_MAX_LEVELS];        /* mvScaleFactors */
} oo_frustum;
typedef struct oo_map_point {    /* the MapPoint members the path reads (L/include/MapPoint.h) */
  float pos[3], normal[3];       /* mWorldPos, mNormalVector */
  float min_distance, max_distance;   /* mfMinDistance, mfMaxDistance (GetMin/MaxDistanceInvariance scale them) */
  int32_t skip;                  /* mnLastFrameSeen == frame id || isBad()   (Tracking.cc:1057-1060) */
  int32_t observed;              /* Observations() > 0 */
  uint8_t desc[32];              /* GetDescriptor() */
} oo_map_point;
typedef struct oo_track {        /* what isInFrustum leaves in the MapPoint (Frame.cc:329-335) */
  int32_t in_view;               /* mbTrackInView */
  float proj_x, proj_y, proj_xr; /* mTrackProjX, mTrackProjY, mTrackProjXR */
  int32_t level;                 /* mnTrackScaleLevel */
  float view_cos;                /* mTrackViewCos */
} oo_track;
int oo_predict_scale(float max_distance, float current_dist, float log_scale_factor, int n_levels); /* MapPoint.cc:409-423 */
int oo_is_in_frustum(const oo_frustum* fr, const oo_map_point* mp, float viewing_cos_limit, oo_track* out); /* Frame.cc:284-339 */
/* query SearchByProjection(F, vpMapPoints, th) forms from one tracked point (ORBmatcher.cc:52-71) */
void oo_local_point_query(const oo_frustum* fr, const oo_map_point* mp, const oo_track* tr, float th, oo_query* q);
/* Tracking::SearchLocalPoints second half (Tracking.cc:1053-1078): isInFrustum(pMP, 0.5) for every local point that is
 * not skipped, then SearchByProjection(F, points, th) when anything is visible.  track[] per point; *n_to_match =
 * nToMatch.  Returns nmatches (0 when nToMatch == 0). */
int oo_search_local_points(const oo_frame* f, const oo_frustum* fr, const oo_map_point* mp, int n, float th, float nnratio,
                           oo_track* track, uint8_t* blocked, int32_t* assigned, int* n_to_match);

/* ---- keyframe-rate projection searches, whole loops (ORBmatcher.cc:766-1245, 275-386, 1385-1504) */
typedef struct oo_kf_camera {
  float R[9], t[3], R2[9], t2[3], Ow[3];
  float fx, fy, cx, cy, mbf, min_x, max_x, min_y, max_y, log_scale_factor;
  int32_t n_levels;
  float th, scale_factors[OO_MAX_LEVELS];
} oo_kf_camera;
typedef struct oo_kf_point { float pos[3], normal[3], min_distance, max_distance; int32_t skip; float angle; uint8_t desc[32]; } oo_kf_point;
typedef struct oo_kf_result { int32_t best_idx, best_dist, level; float u, v, u_r; } oo_kf_result;
void oo_fuse(const oo_frame* kf, const float* inv_level_sigma2, const oo_kf_camera* cam, const oo_kf_point* pts, int n, oo_kf_result* res);
void oo_fuse_sim3(const oo_frame* kf, const oo_kf_camera* cam, const oo_kf_point* pts, int n, oo_kf_result* res);
void oo_search_by_sim3_dir(const oo_frame* kf, const oo_kf_camera* cam, const oo_kf_point* pts, int n, oo_kf_result* res);
int oo_search_by_projection_loop(const oo_frame* kf, const oo_kf_camera* cam, const oo_kf_point* pts, int n, int th_low, uint8_t* matched,
                                 oo_kf_result* res);
void oo_reloc_query(const oo_kf_camera* cam, const oo_kf_point* pMP, oo_query* q);

/* ---- motion-model tracking: Frame::UnprojectStereo (Frame.cc:668-679) and the projection part of
 * SearchByProjection(cur, last) (ORBmatcher.cc:1257-1308) */
typedef struct oo_unproject_cam { float Rwc[9], Ow[3], cx, cy, invfx, invfy; } oo_unproject_cam;
typedef struct oo_last_point { float pos[3]; int32_t valid, observed, octave; float angle; uint8_t desc[32]; } oo_last_point;
typedef struct oo_track_pose {
  float Rcw[9], tcw[3], fx, fy, cx, cy, mbf, min_x, max_x, min_y, max_y;
  int32_t forward, backward;
  float th, scale_factors[OO_MAX_LEVELS];
} oo_track_pose;
/* record of keypoint kp with depth z: map point = UnprojectStereo when z > 0 */
void oo_unproject_stereo(const oo_unproject_cam* cam, const oo_keypoint* kp, float z, const uint8_t* desc, int observed,
                         oo_last_point* out);
/* query the reference forms from one last-frame point (valid = 0 where it `continue`s) */
void oo_track_query(const oo_track_pose* pose, const oo_last_point* lp, oo_query* q);
/* the same over arrays (n keypoints of one frame) */
void oo_unproject_stereo_n(const oo_unproject_cam* cam, const oo_keypoint* kps, const float* depth, const uint8_t* desc, int n,
                           int observed, oo_last_point* out);
void oo_track_queries_n(const oo_track_pose* pose, const oo_last_point* lp, int n, oo_query* q);

/* SearchByProjection(Frame& cur, const Frame& last, th, bMono): ORBmatcher.cc:1247-1383 (window,
 * argmin, TH_HIGH, rotation histogram).  Queries carry the projection.  Returns nmatches. */
int oo_search_by_projection_frame(const oo_frame* cur, const oo_query* q, int nq, int check_orientation,
                                  uint8_t* blocked, int32_t* assigned);
/* SearchByProjection(Frame&, KeyFrame*, const set<MapPoint*>&, th, ORBdist): ORBmatcher.cc:1385-1504 from the window
 * query onwards (q[i] = one keyframe map point that passed :1403-1434).  mappoint_set[i2] != 0 <=> the current frame's
 * mvpMapPoints[i2] is non-NULL; assigned[i2] = query index.  Returns nmatches. */
int oo_search_by_projection_keyframe(const oo_frame* cur, const oo_query* q, int nq, int check_orientation, int orb_dist,
                                     uint8_t* mappoint_set, int32_t* assigned);
/* SearchByBoW(KeyFrame*, Frame&, ...): ORBmatcher.cc:161-273 with feature vectors as sorted node lists.
 * nodeA/nodeB: n_nodes entries {node_id, start, count} into idxA/idxB.  validA[i]!=0 <=> KF feature i has
 * a good map point.  matchB[j] = KF index matched to frame feature j, or -1.  Returns nmatches. */
typedef struct oo_featvec_node { int32_t node_id, start, count; } oo_featvec_node;
int oo_search_by_bow(const uint8_t* descA, const float* angleA, const uint8_t* validA,
                     const oo_featvec_node* nodesA, int n_nodesA, const int32_t* idxA, const uint8_t* descB,
                     const float* angleB, int nB, const oo_featvec_node* nodesB, int n_nodesB,
                     const int32_t* idxB, float nnratio, int check_orientation, int32_t* matchB);
/* SearchByBoW(KeyFrame*, KeyFrame*, vector<MapPoint*>&): ORBmatcher.cc:494-612.  validA/validB: the keyframe feature has
 * a map point that is not bad.  matchA[i] = index in B matched to feature i of A, or -1.  Returns nmatches. */
int oo_search_by_bow_kf(const uint8_t* descA, const float* angleA, const uint8_t* validA, int nA,
                        const oo_featvec_node* nodesA, int n_nodesA, const int32_t* idxA, const uint8_t* descB,
                        const float* angleB, const uint8_t* validB, int nB, const oo_featvec_node* nodesB, int n_nodesB,
                        const int32_t* idxB, float nnratio, int check_orientation, int32_t* matchA);
/* Candidate loop of Fuse (ORBmatcher.cc:818-868; gate = 2: chi-square reprojection test) and of Fuse(Sim3) / SearchBySim3
 * (:983-1009, 1118-1147; gate = 1: none): first minimum over KeyFrame::GetFeaturesInArea(u, v, radius) with the octave in
 * [min_level, max_level].  best_idx / best_dist per query (-1 / 256 when nothing qualifies). */
void oo_proj_best(const oo_frame* kf, const oo_query* q, int nq, int gate, const float* inv_level_sigma2, int32_t* best_idx,
                  int32_t* best_dist);
/* SearchForTriangulation: ORBmatcher.cc:614-764 with CheckDistEpipolarLine :137-159.  has_mp: GetMapPoint(i) != NULL; u_right may
 * be NULL (monocular).  matchA[i] = vMatches12[i].  Returns nmatches. */
typedef struct oo_epipolar { float F12[9], ex, ey, scale_factors[OO_MAX_LEVELS], level_sigma2[OO_MAX_LEVELS]; } oo_epipolar;
int oo_search_for_triangulation(const oo_keypoint* keysA, const uint8_t* descA, const float* u_rightA, const uint8_t* has_mpA, int nA,
                                const oo_featvec_node* nodesA, int n_nodesA, const int32_t* idxA, const oo_keypoint* keysB,
                                const uint8_t* descB, const float* u_rightB, const uint8_t* has_mpB, int nB,
                                const oo_featvec_node* nodesB, int n_nodesB, const int32_t* idxB, const oo_epipolar* ep,
                                int only_stereo, int check_orientation, int32_t* matchA);
/* SearchForInitialization: ORBmatcher.cc:388-492 */
int oo_search_for_initialization(const oo_keypoint* keys1, const uint8_t* desc1, int n1, const oo_frame* f2,
                                 float* prev_matched_xy, int window, float nnratio, int check_orientation,
                                 int32_t* matches12);
void oo_three_maxima(const int* histo_sizes, int L, int* ind1, int* ind2, int* ind3); /* :1506-1538 */

/* Frame::ComputeStereoMatches: Frame.cc:477-646.  pyrL/pyrR: per-level plane pointers+strides+sizes of
 * the two extractors' mvImagePyramid.  Writes u_right[n], depth[n] (-1 when unmatched). */
typedef struct oo_pyramid_view {
  int n_levels;
  const uint8_t* data[OO_MAX_LEVELS];
  int stride[OO_MAX_LEVELS], w[OO_MAX_LEVELS], h[OO_MAX_LEVELS];
} oo_pyramid_view;
int oo_compute_stereo_matches(const oo_keypoint* keysL, const uint8_t* descL, int nL, const oo_keypoint* keysR,
                              const uint8_t* descR, int nR, const oo_pyramid_view* pyrL,
                              const oo_pyramid_view* pyrR, const float* scale_factors,
                              const float* inv_scale_factors, float mbf, float mb, float* u_right,
                              float* depth);

/* ---------------------------------------------------------------- vocabulary (DBoW2 + ORBVocabulary)
 * Frame::ComputeBoW (L/src/Frame.cc:412-417) -> TemplatedVocabulary::transform(features, BowVector&,
 * FeatureVector&, levelsup) (Source/ThirdParty/DBoW2/DBoW2-local/include/DBoW2/TemplatedVocabulary.h:1125-1257),
 * FORB::distance (src/FORB.cpp:77-100), BowVector::addWeight/normalize (src/BowVector.cpp:34-84),
 * FeatureVector::addFeature (src/FeatureVector.cpp:31-45); text format of L/src/ORBVocabulary.cc:11-127. */
typedef struct oo_vocab oo_vocab;
/* nodes 0..n_nodes-1, node 0 = root; parent[i] < i; children keep ascending id order (file order) */
oo_vocab* oo_vocab_create(int k, int L, int scoring, int weighting, int n_nodes, const int32_t* parent,
                          const uint8_t* is_leaf, const uint8_t* desc, const double* weight);
oo_vocab* oo_vocab_load_text(const char* path);
oo_vocab* oo_vocab_load_binary(const char* path);  /* ORBVocabulary.cc:152-213, incl. the node its eof loop duplicates */
void oo_vocab_destroy(oo_vocab* v);
int oo_vocab_nodes(const oo_vocab* v);
int oo_vocab_words(const oo_vocab* v);
void oo_vocab_transform_feature(const oo_vocab* v, const uint8_t* d, int levelsup, int32_t* word_id, int32_t* node_id,
                                double* weight);
/* full transform: BowVector as ascending (id, value) pairs, FeatureVector as ascending nodes + index lists */
int oo_vocab_transform(const oo_vocab* v, const uint8_t* desc, int n, int levelsup, int32_t* bow_ids, double* bow_vals,
                       int* n_bow, oo_featvec_node* fv_nodes, int32_t* fv_idx, int* n_fv);

#ifdef __cplusplus
}
#endif
#endif
