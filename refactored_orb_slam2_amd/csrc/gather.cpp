// gather.cpp -- the batched-sequence mode's only exchange, behind the C ABI: the gather of fixed-size padded per-frame
// records {n; keypoints[cap]; descriptors[cap][32]} over RCCL.
//
// The reference processes a sequence in one process, one frame after the other (Source/Examples/Stereo/stereo_kitti.cc:88-106);
// extraction carries no state between frames, so contiguous chunks of the frame range are independent units: one host thread or
// process per GPU, each with its own extractor / matcher handles and stream, no collective inside the step, and ONE exchange of
// the records -- ncclAllGather when every rank wants every record, grouped ncclSend / ncclRecv to rank 0 when only the consumer
// of the records does.  Equal counts per rank by construction (the last chunk is padded).
//
// librccl is loaded on first use (dlopen), so a single-GPU caller needs no RCCL at all, and a process that already holds an
// RCCL (PyTorch ships one under the same soname) shares it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <mutex>
#include <vector>

#include "../../include/orbfe.h"

void orbfe_set_error(const char* fmt, ...);

namespace {
// the part of rccl.h this file uses (rccl.h itself is not needed to build liborbfe)
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
enum { ncclSuccess = 0 };
enum { ncclUint8 = 1 };
struct Rccl {
  void* so = nullptr;
  int (*GetUniqueId)(ncclUniqueId*) = nullptr;
  int (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  int (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  int (*CommDestroy)(ncclComm_t) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
  int (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  int (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
Rccl g_rccl;
std::once_flag g_rccl_once;
bool g_rccl_ok = false;

bool load_rccl() {
  std::call_once(g_rccl_once, [] {
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names)
      if ((g_rccl.so = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
    if (!g_rccl.so) return;
#define SYM(field, name) (*(void**)&g_rccl.field = dlsym(g_rccl.so, name))
    SYM(GetUniqueId, "ncclGetUniqueId"); SYM(CommInitRank, "ncclCommInitRank"); SYM(CommInitAll, "ncclCommInitAll");
    SYM(CommDestroy, "ncclCommDestroy"); SYM(AllGather, "ncclAllGather"); SYM(Send, "ncclSend"); SYM(Recv, "ncclRecv");
    SYM(GroupStart, "ncclGroupStart"); SYM(GroupEnd, "ncclGroupEnd"); SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    g_rccl_ok = g_rccl.GetUniqueId && g_rccl.CommInitRank && g_rccl.CommInitAll && g_rccl.CommDestroy && g_rccl.AllGather &&
                g_rccl.Send && g_rccl.Recv && g_rccl.GroupStart && g_rccl.GroupEnd;
  });
  if (!g_rccl_ok) orbfe_set_error("librccl is not available (dlopen / dlsym failed): the record gather needs RCCL");
  return g_rccl_ok;
}
int nccl_fail(const char* what, int r) {
  orbfe_set_error("%s: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "RCCL error");
  return ORBFE_ERR_HIP;
}
}  // namespace

struct orbfe_gather {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1, device = 0;
  hipStream_t stream = nullptr;   // used when the caller passes no stream
};

extern "C" int orbfe_shard_range(int n_frames, int rank, int world, int* begin, int* end) {
  if (world < 1 || rank < 0 || rank >= world || n_frames < 0 || !begin || !end) return ORBFE_ERR_INVALID;
  const int base = n_frames / world, rem = n_frames % world;
  *begin = rank * base + (rank < rem ? rank : rem);
  *end = *begin + base + (rank < rem ? 1 : 0);
  return ORBFE_OK;
}

extern "C" int orbfe_gather_unique_id(uint8_t id[ORBFE_GATHER_ID_BYTES]) {
  if (!id) return ORBFE_ERR_INVALID;
  if (!load_rccl()) return ORBFE_ERR_NO_DEVICE;
  ncclUniqueId u;
  const int r = g_rccl.GetUniqueId(&u);
  if (r != ncclSuccess) return nccl_fail("ncclGetUniqueId", r);
  static_assert(sizeof(u) == ORBFE_GATHER_ID_BYTES, "ncclUniqueId size");
  memcpy(id, &u, sizeof(u));
  return ORBFE_OK;
}

static int make_handle(ncclComm_t comm, int rank, int world, int device, orbfe_gather** out) {
  orbfe_gather* g = new (std::nothrow) orbfe_gather();
  if (!g) return ORBFE_ERR_ALLOC;
  g->comm = comm; g->rank = rank; g->world = world; g->device = device;
  if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking) != hipSuccess) {
    orbfe_set_error("orbfe_gather: cannot create a stream on device %d", device);
    delete g;
    return ORBFE_ERR_HIP;
  }
  *out = g;
  return ORBFE_OK;
}

extern "C" int orbfe_gather_create(const uint8_t* id, int rank, int world, int device, orbfe_gather** out) {
  if (!id || !out || world < 1 || rank < 0 || rank >= world) return ORBFE_ERR_INVALID;
  *out = nullptr;
  if (!load_rccl()) return ORBFE_ERR_NO_DEVICE;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { orbfe_set_error("no HIP device available"); return ORBFE_ERR_NO_DEVICE; }
  if (device < 0 && hipGetDevice(&device) != hipSuccess) device = 0;
  if (device >= ndev) return ORBFE_ERR_INVALID;
  if (hipSetDevice(device) != hipSuccess) return ORBFE_ERR_HIP;
  ncclUniqueId u;
  memcpy(&u, id, sizeof(u));
  ncclComm_t comm = nullptr;
  const int r = g_rccl.CommInitRank(&comm, world, u, rank);
  if (r != ncclSuccess) return nccl_fail("ncclCommInitRank", r);
  const int rc = make_handle(comm, rank, world, device, out);
  if (rc) g_rccl.CommDestroy(comm);
  return rc;
}

extern "C" int orbfe_gather_create_all(int n_devices, const int* devices, orbfe_gather** out) {
  if (n_devices < 1 || !out) return ORBFE_ERR_INVALID;
  for (int i = 0; i < n_devices; i++) out[i] = nullptr;
  if (!load_rccl()) return ORBFE_ERR_NO_DEVICE;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { orbfe_set_error("no HIP device available"); return ORBFE_ERR_NO_DEVICE; }
  std::vector<int> devs((size_t)n_devices);
  for (int i = 0; i < n_devices; i++) {
    devs[(size_t)i] = devices ? devices[i] : i;
    if (devs[(size_t)i] < 0 || devs[(size_t)i] >= ndev) { orbfe_set_error("device %d of %d", devs[(size_t)i], ndev); return ORBFE_ERR_INVALID; }
  }
  std::vector<ncclComm_t> comms((size_t)n_devices, nullptr);
  const int r = g_rccl.CommInitAll(comms.data(), n_devices, devs.data());
  if (r != ncclSuccess) return nccl_fail("ncclCommInitAll", r);
  for (int i = 0; i < n_devices; i++) {
    const int rc = make_handle(comms[(size_t)i], i, n_devices, devs[(size_t)i], &out[i]);
    if (rc) {
      for (int j = 0; j < n_devices; j++) {
        if (out[j]) { (void)hipStreamDestroy(out[j]->stream); delete out[j]; out[j] = nullptr; }
        g_rccl.CommDestroy(comms[(size_t)j]);
      }
      return rc;
    }
  }
  return ORBFE_OK;
}

extern "C" int orbfe_gather_destroy(orbfe_gather* g) {
  if (!g) return ORBFE_OK;
  (void)hipSetDevice(g->device);
  if (g->stream) { (void)hipStreamSynchronize(g->stream); (void)hipStreamDestroy(g->stream); }
  if (g->comm && g_rccl_ok) g_rccl.CommDestroy(g->comm);
  delete g;
  return ORBFE_OK;
}

extern "C" int orbfe_gather_rank(const orbfe_gather* g, int* rank, int* world) {
  if (!g) return ORBFE_ERR_INVALID;
  if (rank) *rank = g->rank;
  if (world) *world = g->world;
  return ORBFE_OK;
}

extern "C" int orbfe_gather_records(orbfe_gather* g, const int32_t* d_n, const orbfe_keypoint* d_kps, const uint8_t* d_desc,
                                    int frames, int cap, int mode, int32_t* d_n_all, orbfe_keypoint* d_kps_all,
                                    uint8_t* d_desc_all, void* stream) {
  if (!g || !d_n || !d_kps || !d_desc || frames < 1 || cap < 1 || (mode != ORBFE_GATHER_ALL && mode != ORBFE_GATHER_ROOT))
    return ORBFE_ERR_INVALID;
  const bool receives = mode == ORBFE_GATHER_ALL || g->rank == 0;
  if (receives && (!d_n_all || !d_kps_all || !d_desc_all)) return ORBFE_ERR_INVALID;
  if (hipSetDevice(g->device) != hipSuccess) return ORBFE_ERR_HIP;
  hipStream_t s = stream ? (hipStream_t)stream : g->stream;
  const size_t bytes[3] = {sizeof(int32_t) * (size_t)frames, sizeof(orbfe_keypoint) * (size_t)frames * cap, (size_t)32 * frames * cap};
  const void* src[3] = {d_n, d_kps, d_desc};
  void* dst[3] = {d_n_all, d_kps_all, d_desc_all};
  int r = g_rccl.GroupStart();
  if (r != ncclSuccess) return nccl_fail("ncclGroupStart", r);
  for (int t = 0; t < 3 && r == ncclSuccess; t++) {
    if (mode == ORBFE_GATHER_ALL) {
      r = g_rccl.AllGather(src[t], dst[t], bytes[t], ncclUint8, g->comm, s);
    } else if (g->rank == 0) {
      for (int p = 1; p < g->world && r == ncclSuccess; p++)
        r = g_rccl.Recv((uint8_t*)dst[t] + bytes[t] * (size_t)p, bytes[t], ncclUint8, p, g->comm, s);
    } else {
      r = g_rccl.Send(src[t], bytes[t], ncclUint8, 0, g->comm, s);
    }
  }
  const int re = g_rccl.GroupEnd();
  if (r != ncclSuccess) return nccl_fail("RCCL gather", r);
  if (re != ncclSuccess) return nccl_fail("ncclGroupEnd", re);
  if (mode == ORBFE_GATHER_ROOT && g->rank == 0)   // rank 0's own records: a plain copy on the same stream
    for (int t = 0; t < 3; t++)
      if (hipMemcpyAsync(dst[t], src[t], bytes[t], hipMemcpyDeviceToDevice, s) != hipSuccess) {
        orbfe_set_error("orbfe_gather_records: device copy failed");
        return ORBFE_ERR_HIP;
      }
  return ORBFE_OK;
}

extern "C" int orbfe_gather_sync(orbfe_gather* g) {
  if (!g) return ORBFE_ERR_INVALID;
  if (hipSetDevice(g->device) != hipSuccess || hipStreamSynchronize(g->stream) != hipSuccess) return ORBFE_ERR_HIP;
  return ORBFE_OK;
}
