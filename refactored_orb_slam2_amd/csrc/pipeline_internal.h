// pipeline_internal.h -- the two copy kernels of csrc/pipeline.cpp (pipeline_kernels.hip).  Plumbing only: every computation of a
// chunk goes through the public device entry points of include/orbfe.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include "../../include/orbfe.h"

// `bytes` (a multiple of 16, both pointers 16-byte aligned) from device memory to any device-visible memory -- the pinned host block
// of a slot -- by `workgroups` workgroups of 16-byte loads / stores
void orbfe_launch_copy_block(const void* src, void* dst, size_t bytes, int workgroups, hipStream_t s);
// the carry frame of a chunked host: the last frame's descriptors | keypoints | depth | camera | count in one launch
void orbfe_launch_carry_frame(const uint8_t* desc, const orbfe_keypoint* kps, const float* depth, const orbfe_unproject_cam* cam,
                              const int32_t* n, int cap, uint8_t* c_desc, orbfe_keypoint* c_kps, float* c_depth,
                              orbfe_unproject_cam* c_cam, int32_t* c_n, hipStream_t s);
