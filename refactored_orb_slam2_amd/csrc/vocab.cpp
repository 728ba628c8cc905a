// vocab.cpp -- ORBVocabulary on the device: loading (text format of L/src/ORBVocabulary.cc:11-127), tree upload, and
// Frame::ComputeBoW = TemplatedVocabulary::transform(features, BowVector&, FeatureVector&, levelsup)
// (Source/ThirdParty/DBoW2/DBoW2-local/include/DBoW2/TemplatedVocabulary.h:1125-1192).  The tree descent (all the
// Hamming work) runs in vocab_kernels.hip; assembling the two std::map-shaped results from the per-feature
// (word, node, weight) triples is host book-keeping that follows BowVector::addWeight / addIfNotExist / normalize
// (src/BowVector.cpp:34-84) and FeatureVector::addFeature (src/FeatureVector.cpp:31-45).
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <vector>

#include "vocab_internal.h"

void orbfe_set_error(const char* fmt, ...);

#define HIPCHK(expr)                                                                              \
  do {                                                                                            \
    hipError_t _e = (expr);                                                                       \
    if (_e != hipSuccess) {                                                                       \
      orbfe_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return ORBFE_ERR_HIP;                                                                       \
    }                                                                                             \
  } while (0)

struct orbfe_vocabulary {
  int device = 0;
  int k = 0, L = 0, scoring = 0, weighting = 0;
  int n_nodes = 0, n_words = 0;
  hipStream_t stream = nullptr;
  void *d_desc = nullptr, *d_cs = nullptr, *d_ci = nullptr, *d_word = nullptr, *d_weight = nullptr;
  // per-call scratch
  void *d_in = nullptr, *d_ow = nullptr, *d_on = nullptr, *d_owt = nullptr;
  int cap = 0;
  std::mutex mu;
};

static void vfree(orbfe_vocabulary* v) {
  void* p[] = {v->d_desc, v->d_cs, v->d_ci, v->d_word, v->d_weight, v->d_in, v->d_ow, v->d_on, v->d_owt};
  for (void* q : p)
    if (q) (void)hipFree(q);
  if (v->stream) (void)hipStreamDestroy(v->stream);
  delete v;
}

extern "C" int orbfe_vocabulary_create(int k, int L, int scoring, int weighting, int n_nodes, const int32_t* parent,
                                       const uint8_t* is_leaf, const uint8_t* desc, const double* weight, int device,
                                       orbfe_vocabulary** out) {
  if (!out) return ORBFE_ERR_INVALID;
  *out = nullptr;
  if (n_nodes < 2 || !parent || !is_leaf || !desc || !weight || L < 1 || scoring < 0 || scoring > 5 || weighting < 0 ||
      weighting > 3)
    return ORBFE_ERR_INVALID;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
    orbfe_set_error("no HIP device available (liborbfe has no CPU fallback)");
    return ORBFE_ERR_NO_DEVICE;
  }
  if (device < 0 && hipGetDevice(&device) != hipSuccess) device = 0;
  if (device >= ndev) return ORBFE_ERR_INVALID;
  // children lists in ascending id order == the order loadFromTextFile pushes them (:96)
  std::vector<int32_t> cs((size_t)n_nodes + 1, 0), ci((size_t)n_nodes, 0), wid((size_t)n_nodes, 0), fill((size_t)n_nodes, 0);
  for (int i = 1; i < n_nodes; i++) {
    if (parent[i] < 0 || parent[i] >= i) {
      orbfe_set_error("vocabulary node %d has parent %d (must precede it)", i, parent[i]);
      return ORBFE_ERR_INVALID;
    }
    cs[parent[i] + 1]++;
  }
  for (int i = 0; i < n_nodes; i++) cs[i + 1] += cs[i];
  for (int i = 1; i < n_nodes; i++) ci[cs[parent[i]] + fill[parent[i]]++] = i;
  int n_words = 0;
  for (int i = 1; i < n_nodes; i++)
    if (is_leaf[i]) wid[i] = n_words++;  // word ids in file order (:115-120)
  if (cs[1] == cs[0]) {
    orbfe_set_error("vocabulary root has no children");
    return ORBFE_ERR_INVALID;
  }
  HIPCHK(hipSetDevice(device));
  orbfe_vocabulary* v = new orbfe_vocabulary();
  v->device = device; v->k = k; v->L = L; v->scoring = scoring; v->weighting = weighting;
  v->n_nodes = n_nodes; v->n_words = n_words;
  auto up = [&](void** d, const void* h, size_t bytes) -> int {
    HIPCHK(hipMalloc(d, bytes ? bytes : 16));
    HIPCHK(hipMemcpy(*d, h, bytes, hipMemcpyHostToDevice));
    return ORBFE_OK;
  };
  int rc;
  if (hipStreamCreateWithFlags(&v->stream, hipStreamNonBlocking) != hipSuccess) { vfree(v); return ORBFE_ERR_HIP; }
  if ((rc = up(&v->d_desc, desc, (size_t)n_nodes * 32)) || (rc = up(&v->d_cs, cs.data(), sizeof(int32_t) * cs.size())) ||
      (rc = up(&v->d_ci, ci.data(), sizeof(int32_t) * ci.size())) || (rc = up(&v->d_word, wid.data(), sizeof(int32_t) * wid.size())) ||
      (rc = up(&v->d_weight, weight, sizeof(double) * n_nodes))) {
    vfree(v);
    return rc;
  }
  *out = v;
  return ORBFE_OK;
}

// ORBVocabulary::loadFromTextFile (L/src/ORBVocabulary.cc:11-127): "k L scoring weighting" then one node per line:
// "parent isLeaf d0 .. d31 weight"; node ids are line numbers + 1, node 0 is the root.
extern "C" int orbfe_vocabulary_load_text(const char* path, int device, orbfe_vocabulary** out) {
  if (!path || !out) return ORBFE_ERR_INVALID;
  *out = nullptr;
  FILE* f = fopen(path, "r");
  if (!f) {
    orbfe_set_error("cannot open vocabulary %s", path);
    return ORBFE_ERR_INVALID;
  }
  int k, L, n1, n2;
  if (fscanf(f, "%d %d %d %d", &k, &L, &n1, &n2) != 4 || k < 0 || k > 20 || L < 1 || L > 10 || n1 < 0 || n1 > 5 || n2 < 0 ||
      n2 > 3) {
    fclose(f);
    orbfe_set_error("Vocabulary loading failure: This is not a correct text file!");
    return ORBFE_ERR_INVALID;
  }
  std::vector<int32_t> parent(1, 0);
  std::vector<uint8_t> leaf(1, 0), desc(32, 0);
  std::vector<double> weight(1, 0.0);
  for (;;) {
    int pid, isleaf;
    if (fscanf(f, "%d %d", &pid, &isleaf) != 2) break;
    parent.push_back(pid);
    leaf.push_back(isleaf > 0);
    for (int i = 0; i < 32; i++) {
      int b = 0;
      if (fscanf(f, "%d", &b) != 1) b = 0;
      desc.push_back((uint8_t)b);
    }
    double w = 0;
    if (fscanf(f, "%lf", &w) != 1) w = 0;
    weight.push_back(w);
  }
  fclose(f);
  return orbfe_vocabulary_create(k, L, n1, n2, (int)parent.size(), parent.data(), leaf.data(), desc.data(), weight.data(),
                                 device, out);
}

// ORBVocabulary::loadFromBinaryFile (L/src/ORBVocabulary.cc:152-213), the format saveToBinaryFile writes (:217-243):
// u32 nb_nodes (root included), u32 size_node, i32 k, i32 L, i32 scoring, i32 weighting, then nb_nodes-1 records
// {u32 parent, u8 desc[32], f32 weight, u8 is_leaf}.  The reference's `while (!f.eof())` loop runs once more after the
// last record with the buffer unchanged, which appends a duplicate of the last node (same parent, descriptor, weight;
// one more word when it is a leaf); it can never win a descent (strict `<` against its twin, visited first) but it is
// part of m_nodes / m_words, so it is reproduced here.
extern "C" int orbfe_vocabulary_load_binary(const char* path, int device, orbfe_vocabulary** out) {
  if (!path || !out) return ORBFE_ERR_INVALID;
  *out = nullptr;
  FILE* f = fopen(path, "rb");
  if (!f) {
    orbfe_set_error("cannot open vocabulary %s", path);
    return ORBFE_ERR_INVALID;
  }
  uint32_t hdr[2];
  int32_t kl[4];
  if (fread(hdr, 4, 2, f) != 2 || fread(kl, 4, 4, f) != 4 || hdr[1] < 41 || hdr[1] > 4096 || kl[0] < 0 || kl[0] > 20 ||
      kl[1] < 1 || kl[1] > 10) {
    fclose(f);
    orbfe_set_error("Vocabulary loading failure: This is not a correct binary file!");
    return ORBFE_ERR_INVALID;
  }
  const uint32_t size_node = hdr[1];
  std::vector<int32_t> parent(1, 0);
  std::vector<uint8_t> leaf(1, 0), desc(32, 0), buf(size_node);
  std::vector<double> weight(1, 0.0);
  bool any = false;
  for (;;) {
    const bool got = fread(buf.data(), 1, size_node, f) == size_node;
    if (!got && !any) break;  // no record at all: nothing to duplicate
    int32_t pid;
    float w;
    memcpy(&pid, buf.data(), 4);
    memcpy(&w, buf.data() + 36, 4);
    if (pid < 0 || pid >= (int32_t)parent.size()) {
      fclose(f);
      orbfe_set_error("binary vocabulary: node %zu has parent %d", parent.size(), pid);
      return ORBFE_ERR_INVALID;
    }
    parent.push_back(pid);
    desc.insert(desc.end(), buf.begin() + 4, buf.begin() + 36);
    weight.push_back((double)w);
    leaf.push_back(buf[40] != 0);
    any = true;
    if (!got) break;  // that was the duplicate the reference's eof loop creates
  }
  fclose(f);
  return orbfe_vocabulary_create(kl[0], kl[1], kl[2], kl[3], (int)parent.size(), parent.data(), leaf.data(), desc.data(),
                                 weight.data(), device, out);
}

extern "C" int orbfe_vocabulary_destroy(orbfe_vocabulary* v) {
  if (!v) return ORBFE_OK;
  (void)hipSetDevice(v->device);
  if (v->stream) (void)hipStreamSynchronize(v->stream);
  vfree(v);
  return ORBFE_OK;
}

extern "C" int orbfe_vocabulary_info(const orbfe_vocabulary* v, int* k, int* L, int* n_nodes, int* n_words) {
  if (!v) return ORBFE_ERR_INVALID;
  if (k) *k = v->k;
  if (L) *L = v->L;
  if (n_nodes) *n_nodes = v->n_nodes;
  if (n_words) *n_words = v->n_words;
  return ORBFE_OK;
}

static VocabDev dev_view(const orbfe_vocabulary* v) {
  VocabDev d;
  d.desc = (const uint8_t*)v->d_desc;
  d.child_start = (const int32_t*)v->d_cs;
  d.child_idx = (const int32_t*)v->d_ci;
  d.word_id = (const int32_t*)v->d_word;
  d.weight = (const double*)v->d_weight;
  d.n_nodes = v->n_nodes;
  d.L = v->L;
  return d;
}

extern "C" int orbfe_bow_transform_device(orbfe_vocabulary* v, const uint8_t* d_desc, int n, int levelsup, int32_t* d_word,
                                          int32_t* d_node, double* d_weight, void* stream) {
  if (!v || !d_desc || n < 0 || !d_word || !d_node || !d_weight || ((uintptr_t)d_desc & 15)) return ORBFE_ERR_INVALID;
  HIPCHK(hipSetDevice(v->device));
  orbfe_launch_bow_transform(dev_view(v), d_desc, n, levelsup, d_word, d_node, d_weight, stream ? (hipStream_t)stream : v->stream);
  hipError_t le = hipGetLastError();
  if (le != hipSuccess) {
    orbfe_set_error("kernel launch failed: %s", hipGetErrorString(le));
    return ORBFE_ERR_HIP;
  }
  return ORBFE_OK;
}

extern "C" int orbfe_compute_bow(orbfe_vocabulary* v, const uint8_t* desc, int n, int levelsup, int32_t* word_id,
                                 int32_t* node_id, double* weight, int32_t* bow_ids, double* bow_vals, int* n_bow,
                                 orbfe_featvec_node* fv_nodes, int32_t* fv_idx, int* n_fv_nodes) {
  if (!v || n < 0 || (n > 0 && !desc) || !bow_ids || !bow_vals || !n_bow || !fv_nodes || !fv_idx || !n_fv_nodes)
    return ORBFE_ERR_INVALID;
  *n_bow = 0;
  *n_fv_nodes = 0;
  if (n == 0) return ORBFE_OK;
  std::lock_guard<std::mutex> lk(v->mu);
  HIPCHK(hipSetDevice(v->device));
  if (n > v->cap) {
    for (void** p : {&v->d_in, &v->d_ow, &v->d_on, &v->d_owt})
      if (*p) { HIPCHK(hipFree(*p)); *p = nullptr; }
    HIPCHK(hipMalloc(&v->d_in, (size_t)n * 32));
    HIPCHK(hipMalloc(&v->d_ow, sizeof(int32_t) * n));
    HIPCHK(hipMalloc(&v->d_on, sizeof(int32_t) * n));
    HIPCHK(hipMalloc(&v->d_owt, sizeof(double) * n));
    v->cap = n;
  }
  hipStream_t s = v->stream;
  std::vector<int32_t> w((size_t)n), nd((size_t)n);
  std::vector<double> wt((size_t)n);
  HIPCHK(hipMemcpyAsync(v->d_in, desc, (size_t)n * 32, hipMemcpyHostToDevice, s));
  orbfe_launch_bow_transform(dev_view(v), (const uint8_t*)v->d_in, n, levelsup, (int32_t*)v->d_ow, (int32_t*)v->d_on,
                             (double*)v->d_owt, s);
  HIPCHK(hipMemcpyAsync(w.data(), v->d_ow, sizeof(int32_t) * n, hipMemcpyDeviceToHost, s));
  HIPCHK(hipMemcpyAsync(nd.data(), v->d_on, sizeof(int32_t) * n, hipMemcpyDeviceToHost, s));
  HIPCHK(hipMemcpyAsync(wt.data(), v->d_owt, sizeof(double) * n, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  if (word_id) memcpy(word_id, w.data(), sizeof(int32_t) * n);
  if (node_id) memcpy(node_id, nd.data(), sizeof(int32_t) * n);
  if (weight) memcpy(weight, wt.data(), sizeof(double) * n);
  // ---- BowVector (std::map<WordId, double>) and FeatureVector (std::map<NodeId, vector<unsigned>>)
  std::vector<int> order;
  for (int i = 0; i < n; i++)
    if (wt[i] > 0) order.push_back(i);  // not stopped
  const bool tf = v->weighting == 0 /*TF_IDF*/ || v->weighting == 1 /*TF*/;
  std::vector<int> byw(order), byn(order);
  std::stable_sort(byw.begin(), byw.end(), [&](int a, int b) { return w[a] < w[b]; });
  std::stable_sort(byn.begin(), byn.end(), [&](int a, int b) { return nd[a] < nd[b]; });
  int nb = 0;
  for (size_t i = 0; i < byw.size();) {
    size_t j = i + 1;
    double acc = wt[byw[i]];  // addWeight sums in feature order; addIfNotExist keeps the first
    for (; j < byw.size() && w[byw[j]] == w[byw[i]]; j++)
      if (tf) acc += wt[byw[j]];
    bow_ids[nb] = w[byw[i]];
    bow_vals[nb] = acc;
    nb++;
    i = j;
  }
  const bool must = v->scoring != 5;  // DOT_PRODUCT is the only scoring that does not normalise
  const bool l2 = v->scoring == 1;
  if (tf && nb > 0 && !must) {
    const double ndv = (double)nb;
    for (int i = 0; i < nb; i++) bow_vals[i] /= ndv;
  }
  if (must) {
    double norm = 0.0;
    if (!l2) for (int i = 0; i < nb; i++) norm += fabs(bow_vals[i]);
    else { for (int i = 0; i < nb; i++) norm += bow_vals[i] * bow_vals[i]; norm = sqrt(norm); }
    if (norm > 0.0) for (int i = 0; i < nb; i++) bow_vals[i] /= norm;
  }
  int nf = 0, pos = 0;
  for (size_t i = 0; i < byn.size();) {
    size_t j = i;
    fv_nodes[nf].node_id = nd[byn[i]];
    fv_nodes[nf].start = pos;
    for (; j < byn.size() && nd[byn[j]] == nd[byn[i]]; j++) fv_idx[pos++] = byn[j];
    fv_nodes[nf].count = pos - fv_nodes[nf].start;
    nf++;
    i = j;
  }
  *n_bow = nb;
  *n_fv_nodes = nf;
  return ORBFE_OK;
}
