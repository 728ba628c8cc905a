// match_internal.h -- structs shared by matcher.cpp and match_kernels.hip
#pragma once
#include "orbfe_internal.h"

#define GRID_CELLS (ORBFE_GRID_COLS * ORBFE_GRID_ROWS)

// A batch of frames with identical capacity: frame f uses rows [f*cap, f*cap + n[f])
struct FrameBatch {
  const orbfe_keypoint* keys;
  const uint8_t* desc;
  const float* u_right;      // nullable
  const int32_t* n;          // [n_frames]
  int32_t* cell_start;       // [n_frames][GRID_CELLS + 1]
  int32_t* cell_idx;         // [n_frames][cap]
  uint32_t* cell_rec;        // [n_frames][cap] x 16 B, CSR order: (index | octave << 24, x, y, mvuRight or -1): what a window walk reads per entry
  int cap;
  float min_x, min_y, gw_inv, gh_inv;
};

struct QueryBatch {
  const orbfe_query* q;      // [n_frames][cap]
  const int32_t* n;          // [n_frames]
  int cap;
};

struct HammingBfParams {
  const uint8_t* A;
  const int32_t* nA;
  int strideA;
  const uint8_t* B;
  const int32_t* nB;
  int strideB;
  const int32_t* groupA;
  const int32_t* groupB;
  const uint8_t* maskB;
  orbfe_bf_match* out;
};

struct StereoParams {
  PyrView pyrL, pyrR;
  const orbfe_keypoint* kpsL;
  const uint8_t* descL;
  const int32_t* nL;
  const orbfe_keypoint* kpsR;
  const uint8_t* descR;
  const int32_t* nR;
  int cap;
  float scale[ORBFE_MAX_LEVELS], inv_scale[ORBFE_MAX_LEVELS];
  float mbf, maxD;
  float* u_right;
  float* depth;
  int32_t* sad;        // scratch [n_pairs][cap]
  int32_t* n_matched;  // [n_pairs]
  // row buckets of the right keypoints (bucket b = image rows [8b, 8b+8)): CSR per pair
  int32_t* bucket_start;  // [n_pairs][n_keys + 1], key = (row / 8) * n_levels + octave
  int32_t* bucket_idx;    // [n_pairs][cap * STEREO_BUCKET_SPAN]
  int n_buckets;          // row buckets
  int n_levels, n_keys;   // n_keys = n_buckets * n_levels
};
#define STEREO_MAX_BUCKETS 512      // rows / 8, rows <= 4095
#define STEREO_MAX_KEYS 8192        // (row bucket, octave) keys: 512 x 16; the bucket kernel's LDS holds two ints per key
#define STEREO_BUCKET_SPAN 8        // a band [y-r, y+r], r = 2*scale <= ~25 rows, overlaps at most this many buckets

struct BowPair { int32_t startA, countA, startB, countB; };
struct BowParams {
  const BowPair* pairs;
  const uint8_t* descA; const float* angleA; const uint8_t* validA; const int32_t* idxA;
  const uint8_t* descB; const float* angleB; const int32_t* idxB;
  float nnratio; int check_ori;
  int sequential, n_pairs;  // sequential != 0: a frame feature occurs under more than one node
  int kf_mode;              // SearchByBoW(KF,KF): validB mask, strict TH_LOW, result per A feature
  const uint8_t* validB;
  int32_t* matchA;          // [nA], pre-set to -1 (kf_mode)
  int32_t* matchB;      // [nB], pre-set to -1
  int32_t* counters;    // [0] pushes, [1] nmatches, [2..31] rotation histogram
  int32_t* push_idx; uint8_t* push_bin;
};
void orbfe_launch_unproject_stereo(const orbfe_keypoint* kps, const uint8_t* desc, const int32_t* n, const float* depth, int cap,
                                   const orbfe_unproject_cam* cams, int observed, orbfe_last_point* points, int n_frames,
                                   hipStream_t s);
void orbfe_launch_track_queries(const orbfe_track_pose* poses, const orbfe_last_point* points, const int32_t* n_points, int p_cap,
                                int frame_shift, orbfe_query* queries, int32_t* nq, int n_frames, hipStream_t s);
void orbfe_launch_track_queries_stereo(const orbfe_keypoint* kps, const uint8_t* desc, const int32_t* n, const float* depth, int cap,
                                       const orbfe_unproject_cam* cams, int observed, const orbfe_keypoint* c_kps, const uint8_t* c_desc,
                                       const int32_t* c_n, const float* c_depth, const orbfe_unproject_cam* c_cam,
                                       const orbfe_track_pose* poses, int frame_shift, orbfe_query* queries, int32_t* nq, int n_frames,
                                       hipStream_t s);
void orbfe_launch_frustum_queries(const orbfe_frustum* frustums, const orbfe_map_point* points, const int32_t* n_points,
                                  int p_cap, float th, float viewing_cos_limit, orbfe_track* track, orbfe_query* queries,
                                  int32_t* n_to_match, int n_frames, hipStream_t s);
void orbfe_launch_kf_queries(const orbfe_kf_camera* cam, const orbfe_kf_point* points, int n, int mode, orbfe_query* queries,
                             orbfe_kf_result* results, hipStream_t s);
void orbfe_launch_bow(const BowParams& p, int n_pairs, int max_countB, hipStream_t s);
struct TriParams {
  BowParams b;   // pairs, descriptors, idx arrays, matchA, counters, push arrays, check_ori, sequential, n_pairs (validA / validB: candidate masks)
  const orbfe_keypoint* keysA; const orbfe_keypoint* keysB;
  const uint8_t* stereoA; const uint8_t* stereoB;   // mvuRight >= 0
  orbfe_epipolar ep;
};
void orbfe_launch_triangulation(const TriParams& p, int n_pairs, hipStream_t s);
void orbfe_launch_proj_best(const FrameBatch& f, const QueryBatch& q, int gate, const float* inv_sigma2, int32_t* best_idx,
                            int32_t* best_dist, int n_frames, hipStream_t s);
void orbfe_launch_hamming_matrix(const uint8_t* A, int nA, const uint8_t* B, int nB, uint16_t* out, hipStream_t s);
void orbfe_launch_hamming_bf(const HammingBfParams& p, int max_nA, int n_sets, hipStream_t s);
void orbfe_launch_grid_build(const FrameBatch& f, int n_frames, hipStream_t s);
void orbfe_launch_proj_candidates(const FrameBatch& f, const QueryBatch& q, orbfe_cand* cand, int32_t* n_cand,
                                  int max_cand, int n_frames, hipStream_t s);
void orbfe_launch_proj_resolve(const FrameBatch& f, const QueryBatch& q, const orbfe_cand* cand, const int32_t* n_cand,
                               int max_cand, int mode, int th_high, float nnratio, int check_ori, uint8_t* blocked,
                               int32_t* assigned, int32_t* n_matches, int32_t* push_idx, uint8_t* push_bin, int n_frames,
                               hipStream_t s);
void orbfe_launch_init_resolve(const FrameBatch& f, const QueryBatch& q, const orbfe_cand* cand, const int32_t* n_cand,
                               int max_cand, float nnratio, int check_ori, int32_t* matches12, float* prev_xy,
                               int32_t* n_matches, int32_t* push_idx, uint8_t* push_bin, hipStream_t s);
void orbfe_launch_stereo(const StereoParams& p, int n_pairs, hipStream_t s);
