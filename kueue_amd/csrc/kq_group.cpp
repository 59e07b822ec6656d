// kq_group.cpp — include/kq_group.h on the device: the protocol of kq_group_core.hpp over the C ABI of include/kq_engine.h
// (kq_cycle_nominate_shard / kq_cycle_process_merged), HIP buffers and RCCL, loaded with dlopen. See the header for the protocol.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstdlib>
#include <new>
#include <string>
#include <vector>

#include "kq_group_core.hpp"

namespace {
struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool load(std::string* err) {
    if (lib) return true;
    lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!lib) { *err = std::string("dlopen librccl: ") + dlerror(); return false; }
    CommInitAll = (decltype(CommInitAll))dlsym(lib, "ncclCommInitAll");
    CommDestroy = (decltype(CommDestroy))dlsym(lib, "ncclCommDestroy");
    CommAbort = (decltype(CommAbort))dlsym(lib, "ncclCommAbort");
    GroupStart = (decltype(GroupStart))dlsym(lib, "ncclGroupStart");
    GroupEnd = (decltype(GroupEnd))dlsym(lib, "ncclGroupEnd");
    AllReduce = (decltype(AllReduce))dlsym(lib, "ncclAllReduce");
    GetErrorString = (decltype(GetErrorString))dlsym(lib, "ncclGetErrorString");
    if (!CommInitAll || !CommDestroy || !AllReduce || !GetErrorString || !GroupStart || !GroupEnd) { *err = "librccl lacks ncclCommInitAll / ncclAllReduce / ncclGroupStart"; return false; }
    return true;
  }
};

struct HipBackend {
  Rccl rccl;
  std::vector<hipStream_t> stream;
  std::vector<ncclComm_t> comm;
  int64_t n_allreduce = 0;   // ncclAllReduce groups issued (kq_group_collective_info)
  int rccl_ranks() const { int k = 0; for (auto c : comm) if (c) k++; return k; }
  bool set_device(int d) { return hipSetDevice(d) == hipSuccess; }
  int engine_create(const kq_config* c, void** e) { return kq_engine_create(c, (kq_engine**)e); }
  void engine_destroy(void* e) { kq_engine_destroy((kq_engine*)e); }
  int rank_init(int r, int) {
    if ((int)stream.size() <= r) stream.resize((size_t)r + 1, nullptr);
    return hipStreamCreateWithFlags(&stream[(size_t)r], hipStreamNonBlocking) == hipSuccess ? KQ_OK : KQ_EDEVICE;
  }
  void rank_fini(int r) { if (r < (int)stream.size() && stream[(size_t)r]) { (void)hipStreamDestroy(stream[(size_t)r]); stream[(size_t)r] = nullptr; } }
  int comm_init(int n, const int* devs, std::string* err) {
    if (!rccl.load(err)) return KQ_EDEVICE;
    comm.assign((size_t)n, nullptr);
    const ncclResult_t nr = rccl.CommInitAll(comm.data(), n, devs);
    if (nr != ncclSuccess) { *err = std::string("ncclCommInitAll: ") + rccl.GetErrorString(nr); comm.clear(); return KQ_EDEVICE; }
    return KQ_OK;
  }
  void comm_destroy() {
    for (auto c : comm) if (c && rccl.CommDestroy) (void)rccl.CommDestroy(c);
    comm.clear();
  }
  void* xalloc(size_t b) { void* p = nullptr; return hipMalloc(&p, b) == hipSuccess ? p : nullptr; }
  void xfree(void* p) { (void)hipFree(p); }
  void* host_alloc(size_t b) { void* p = nullptr; return hipHostMalloc(&p, b, hipHostMallocDefault) == hipSuccess ? p : nullptr; }
  void host_free(void* p) { (void)hipHostFree(p); }
  int d2h(void* h, const void* d, size_t b) { return hipMemcpy(h, d, b, hipMemcpyDeviceToHost) == hipSuccess ? KQ_OK : KQ_EDEVICE; }
  int h2d(void* d, const void* h, size_t b) { return hipMemcpy(d, h, b, hipMemcpyHostToDevice) == hipSuccess ? KQ_OK : KQ_EDEVICE; }
  // every rank's all-reduce from ONE thread inside one ncclGroup: an enqueue that fails cannot leave the other ranks waiting in theirs
  int allreduce_all(int n, const int* devs, void* const* xbuf, size_t words, std::string* err) {
    ncclResult_t bad = ncclSuccess;
    n_allreduce++;
    (void)rccl.GroupStart();
    for (int r = 0; r < n; r++) {
      (void)hipSetDevice(devs[r]);
      const ncclResult_t a = rccl.AllReduce(xbuf[r], xbuf[r], words, ncclInt64, ncclSum, comm[(size_t)r], stream[(size_t)r]);
      if (a != ncclSuccess && bad == ncclSuccess) bad = a;
    }
    const ncclResult_t e = rccl.GroupEnd();
    if (bad == ncclSuccess) bad = e;
    (void)hipSetDevice(devs[0]);
    if (bad != ncclSuccess) {
      *err = std::string("ncclAllReduce: ") + rccl.GetErrorString(bad);
      if (rccl.CommAbort) for (auto& c : comm) if (c) { (void)rccl.CommAbort(c); c = nullptr; }   // the communicator is unusable after a failed group
      return KQ_EDEVICE;
    }
    return KQ_OK;
  }
  int comm_wait(int r) { return hipStreamSynchronize(stream[(size_t)r]) == hipSuccess ? KQ_OK : KQ_EDEVICE; }
  int snapshot_put(void* e, const kq_snapshot* s) { return kq_snapshot_put((kq_engine*)e, s); }
  int cycle_run(void* e, const kq_heads* h, kq_decisions* o) { return kq_cycle_run((kq_engine*)e, h, o); }
  int shard_words(void* e, const kq_heads* h, const kq_decisions* o, int world, int64_t* w) { return kq_cycle_shard_words((kq_engine*)e, h, o, world, w); }
  int nominate_shard(void* e, const kq_heads* h, const uint8_t* mine, int world, int rank, void* x, kq_decisions* o) { return kq_cycle_nominate_shard((kq_engine*)e, h, mine, world, rank, x, o); }
  int process_merged(void* e, int world, int rank, const void* x, kq_decisions* o) { return kq_cycle_process_merged((kq_engine*)e, world, rank, x, o); }
  int commit(void* e, int32_t* n) { return kq_cycle_commit((kq_engine*)e, n); }
  int release(void* e, int32_t age) { return kq_cycle_release((kq_engine*)e, age); }
  int read_usage(void* e, int64_t* u) { return kq_snapshot_read_planes((kq_engine*)e, nullptr, u, nullptr); }
  const char* last_error(void* e) { return kq_last_error((kq_engine*)e); }
};
}  // namespace

struct kq_group { kqg::Group<HipBackend> g; };

extern "C" {

int kq_group_create_opts(const kq_config* cfg, int32_t n_dev, const int32_t* devices, uint32_t flags, kq_group** out) {
  if (!cfg || !out || n_dev < 1 || n_dev > 64 || !devices || (flags & ~(uint32_t)(KQ_GROUP_HOST_COLLECTIVE | KQ_GROUP_FORCE_SHARDED))) return KQ_EINVAL;
  try {
    kq_group* g = new (std::nothrow) kq_group();
    if (!g) return KQ_ENOMEM;
    const int rc = g->g.create(cfg, n_dev, devices, flags);
    if (rc != KQ_OK) {
      if (!g->g.last_error.empty()) fprintf(stderr, "kq_group_create: %s\n", g->g.last_error.c_str());
      kq_group_destroy(g);
      return rc;
    }
    *out = g;
    return KQ_OK;
  } catch (const std::bad_alloc&) { return KQ_ENOMEM; } catch (...) { return KQ_EINVAL; }
}
int kq_group_create(const kq_config* cfg, int32_t n_dev, const int32_t* devices, kq_group** out) {
  uint32_t flags = 0;
  if (const char* c = getenv("KQ_GROUP_COLLECTIVE")) if (std::string(c) == "host") flags |= KQ_GROUP_HOST_COLLECTIVE;
  if (const char* c = getenv("KQ_GROUP_FORCE_SHARDED")) if (c[0] && c[0] != '0') flags |= KQ_GROUP_FORCE_SHARDED;
  return kq_group_create_opts(cfg, n_dev, devices, flags, out);
}

void kq_group_destroy(kq_group* g) {
  if (!g) return;
  try { g->g.destroy(); delete g; } catch (...) {}
}

int kq_group_size(const kq_group* g) { return g ? g->g.n : 0; }
const char* kq_group_last_error(kq_group* g) { return g ? g->g.last_error.c_str() : "null group"; }

#define KQG_TRY(expr) try { return (expr); } catch (const std::bad_alloc&) { return g->g.fail(KQ_ENOMEM, "out of host memory"); } catch (const std::exception& x) { return g->g.fail(KQ_EINVAL, x.what()); } catch (...) { return g->g.fail(KQ_EINVAL, "exception"); }

int kq_group_snapshot_put(kq_group* g, const kq_snapshot* s) {
  if (!g || !s) return KQ_EINVAL;
  KQG_TRY(g->g.snapshot_put(s))
}
int kq_group_cycle_run(kq_group* g, const kq_heads* h, kq_decisions* out) {
  if (!g || !h || !out) return KQ_EINVAL;
  KQG_TRY(g->g.cycle_run(h, out))
}
int kq_group_cycle_commit(kq_group* g, int32_t* n_admitted) {
  if (!g) return KQ_EINVAL;
  KQG_TRY(g->g.cycle_commit(n_admitted))
}
int kq_group_cycle_release(kq_group* g, int32_t age) {
  if (!g) return KQ_EINVAL;
  KQG_TRY(g->g.cycle_release(age))
}
int kq_group_read_usage(kq_group* g, int32_t rank, int64_t* usage) {
  if (!g) return KQ_EINVAL;
  KQG_TRY(g->g.read_usage(rank, usage))
}
int kq_group_collective_info(kq_group* g, int32_t* rccl_ranks, int64_t* allreduce_calls, int64_t* host_sums) {
  if (!g) return KQ_EINVAL;
  if (rccl_ranks) *rccl_ranks = g->g.be.rccl_ranks();
  if (allreduce_calls) *allreduce_calls = g->g.be.n_allreduce;
  if (host_sums) *host_sums = g->g.host_sums;
  return KQ_OK;
}

}  // extern "C"
