// vocab_internal.h -- device view of a DBoW2 vocabulary tree
#pragma once
#include "orbfe_internal.h"

struct VocabDev {
  const uint8_t* desc;         // [n_nodes][32]
  const int32_t* child_start;  // [n_nodes + 1], CSR of children in ascending id (= file) order
  const int32_t* child_idx;    // [n_nodes - 1]
  const int32_t* word_id;      // [n_nodes]
  const double* weight;        // [n_nodes]
  int n_nodes, L;
};

void orbfe_launch_bow_transform(const VocabDev& v, const uint8_t* desc, int n, int levelsup, int32_t* word, int32_t* node,
                                double* weight, hipStream_t s);
