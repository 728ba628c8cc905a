// vocab_kernels.hip -- DBoW2 vocabulary-tree descent on gfx950: the per-descriptor part of Frame::ComputeBoW
// (L/src/Frame.cc:412-417 -> TemplatedVocabulary::transform, Source/ThirdParty/DBoW2/DBoW2-local/include/DBoW2/
// TemplatedVocabulary.h:1216-1257, distance = FORB::distance, src/FORB.cpp:77-100).
#include "vocab_internal.h"

#define WAVE 64

// 32 lanes per descriptor (two descriptors per wave): at every tree level each lane takes one child of the current
// node, the half-wave min-reduces (distance << 8 | child position) -- the first minimum in child order, exactly the
// reference's strict "<" scan -- and descends until a node without children.
__global__ __launch_bounds__(256) void bow_transform_kernel(VocabDev V, const uint8_t* __restrict__ desc, int n, int levelsup,
                                                            int32_t* __restrict__ word, int32_t* __restrict__ node,
                                                            double* __restrict__ weight) {
  const int sub = threadIdx.x & 31;
  const int i = (blockIdx.x * 256 + threadIdx.x) >> 5;
  if (i >= n) return;  // uniform per half-wave
  const uint4* dp = reinterpret_cast<const uint4*>(desc + (size_t)i * 32);
  const uint4 a0 = dp[0], a1 = dp[1];
  const int nid_level = V.L - levelsup;
  int nid = 0, final_id = 0, current_level = 0;
  for (;;) {
    ++current_level;
    const int c0 = V.child_start[final_id], c1 = V.child_start[final_id + 1];
    unsigned best = 0xFFFFFFFFu;
    for (int c = c0 + sub; c < c1; c += 32) {  // k <= 32 in practice: one step
      const int id = V.child_idx[c];
      const uint4* q = reinterpret_cast<const uint4*>(V.desc + (size_t)id * 32);
      const uint4 b0 = q[0], b1 = q[1];
      const unsigned d = __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) +
                         __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
      const unsigned key = (d << 20) | (unsigned)(c - c0);
      best = min(best, key);
    }
#pragma unroll
    for (int s = 16; s >= 1; s >>= 1) best = min(best, (unsigned)__shfl_xor((int)best, s, 32));
    final_id = V.child_idx[c0 + (int)(best & 0xfffff)];
    if (current_level == nid_level) nid = final_id;
    if (V.child_start[final_id + 1] == V.child_start[final_id]) break;  // isLeaf()
  }
  if (sub == 0) {
    word[i] = V.word_id[final_id];
    node[i] = nid;
    weight[i] = V.weight[final_id];
  }
}

void orbfe_launch_bow_transform(const VocabDev& v, const uint8_t* desc, int n, int levelsup, int32_t* word, int32_t* node,
                                double* weight, hipStream_t s) {
  if (n < 1) return;
  hipLaunchKernelGGL(bow_transform_kernel, dim3((n * 32 + 255) / 256), dim3(256), 0, s, v, desc, n, levelsup, word, node, weight);
}
