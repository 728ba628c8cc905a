// pipeline_kernels.hip -- copies of the batched pipeline handle (csrc/pipeline.cpp) that are kernels instead of runtime copies.
//
// Copy-out: a chunk's selected output blocks go to the slot's pinned host block.  As hipMemcpyAsync(DeviceToHost) on the copy-out stream
// the 2.5 MB of tracked assignments cost the pipeline 4.5 % of its rate (106.6 k against 111.4 k frames/s, 256-frame chunks, three
// slots; with the counts alone -- 4 KB -- 111.7 k), every block (37 MB) 18 %: not the bytes, the runtime's copy path beside three busy
// compute streams.  A few workgroups storing 16 bytes per lane straight into the (device-visible, coherent) pinned block do the same
// at what the chunk needs of the link's rate: 111.9 k with the assignments, 109.3 k with every block (four workgroups; the count is
// chosen in pipeline.cpp: out_workgroups).  profiles/r06_pipeline.md.
#include "pipeline_internal.h"

__global__ __launch_bounds__(256) void copy_block_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
void orbfe_launch_copy_block(const void* src, void* dst, size_t bytes, int workgroups, hipStream_t s) {
  if (bytes < 16) return;
  hipLaunchKernelGGL(copy_block_kernel, dim3(workgroups), dim3(256), 0, s, reinterpret_cast<const uint4*>(src), reinterpret_cast<uint4*>(dst),
                     bytes / 16);
}

// Carry frame: what the first frame of the next chunk is searched with (orbfe_track_queries_stereo_device's d_carry_* arrays) -- one
// launch on the matching stream instead of five small device-to-device copies (+ 0.5 % on the pipeline's rate).  Dword copies; the
// first workgroup also moves the camera and the count.
__global__ __launch_bounds__(256) void carry_frame_kernel(const uint32_t* __restrict__ desc, const uint32_t* __restrict__ kps,
                                                          const uint32_t* __restrict__ depth, const uint32_t* __restrict__ cam,
                                                          const uint32_t* __restrict__ n, int cap, uint32_t* __restrict__ c_desc,
                                                          uint32_t* __restrict__ c_kps, uint32_t* __restrict__ c_depth,
                                                          uint32_t* __restrict__ c_cam, uint32_t* __restrict__ c_n) {
  const int i = blockIdx.x * 256 + threadIdx.x, step = gridDim.x * 256;
  for (int k = i; k < cap * 8; k += step) c_desc[k] = desc[k];
  for (int k = i; k < cap * (int)(sizeof(orbfe_keypoint) / 4); k += step) c_kps[k] = kps[k];
  for (int k = i; k < cap; k += step) c_depth[k] = depth[k];
  if (i < (int)(sizeof(orbfe_unproject_cam) / 4)) c_cam[i] = cam[i];
  if (i == 0) c_n[0] = n[0];
}
void orbfe_launch_carry_frame(const uint8_t* desc, const orbfe_keypoint* kps, const float* depth, const orbfe_unproject_cam* cam,
                              const int32_t* n, int cap, uint8_t* c_desc, orbfe_keypoint* c_kps, float* c_depth,
                              orbfe_unproject_cam* c_cam, int32_t* c_n, hipStream_t s) {
  hipLaunchKernelGGL(carry_frame_kernel, dim3(32), dim3(256), 0, s, (const uint32_t*)desc, (const uint32_t*)kps, (const uint32_t*)depth,
                     (const uint32_t*)cam, (const uint32_t*)n, cap, (uint32_t*)c_desc, (uint32_t*)c_kps, (uint32_t*)c_depth,
                     (uint32_t*)c_cam, (uint32_t*)c_n);
}
