// kq_group.cpp — include/kq_group.h: one root cohort tree over the GPUs of one process. Host code on top of the C ABI of
// include/kq_engine.h (kq_cycle_nominate_shard / kq_cycle_process_merged) + RCCL, loaded with dlopen. See the header for the protocol.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/kq_group.h"

namespace {
struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool load(std::string* err) {
    if (lib) return true;
    lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!lib) { *err = std::string("dlopen librccl: ") + dlerror(); return false; }
    CommInitAll = (decltype(CommInitAll))dlsym(lib, "ncclCommInitAll");
    CommDestroy = (decltype(CommDestroy))dlsym(lib, "ncclCommDestroy");
    AllReduce = (decltype(AllReduce))dlsym(lib, "ncclAllReduce");
    GetErrorString = (decltype(GetErrorString))dlsym(lib, "ncclGetErrorString");
    if (!CommInitAll || !CommDestroy || !AllReduce || !GetErrorString) { *err = "librccl lacks ncclCommInitAll / ncclAllReduce"; return false; }
    return true;
  }
};
// the decision buffers of the ranks other than 0: same capacities as the caller's
struct Scratch {
  std::vector<uint8_t> u8[10];
  std::vector<int32_t> i32[8];
  std::vector<int16_t> i16[2];
  std::vector<int64_t> i64[3];
  kq_decisions d;
  void size(const kq_heads* h, const kq_decisions* like, int nR) {
    const size_t n = (size_t)h->n, nps = (size_t)h->ps_off[h->n], cells = nps * (size_t)nR;
    const size_t tc = (size_t)std::max(like->tgt_cap, 0), rc = (size_t)std::max(like->rsn_cap, 0);
    auto a8 = [&](int i, size_t k) { u8[i].assign(std::max<size_t>(k, 1), 0); return u8[i].data(); };
    auto a32 = [&](int i, size_t k) { i32[i].assign(std::max<size_t>(k, 1), 0); return i32[i].data(); };
    d = *like;
    d.status = a8(0, n); d.action = a8(1, n); d.nominated_mode = a8(2, n); d.mode = a8(3, n); d.requeue_reason = a8(4, n); d.skip = a8(5, n);
    d.borrowing = a32(0, n); d.order = a32(1, n); d.flavor = a32(2, cells); d.res_mode = a8(6, cells); d.tried_idx = a32(3, cells);
    d.ps_count = a32(4, nps); d.tgt_off = a32(5, n + 1); d.tgt_adm = a32(6, tc); d.tgt_reason = a8(7, tc);
    if (like->rsn_cap > 0) {
      d.rsn_off = a32(7, n + 1);
      d.rsn_code = a8(8, rc); d.rsn_podset = a8(9, rc);
      i16[0].assign(rc, 0); i16[1].assign(rc, 0); d.rsn_flavor = i16[0].data(); d.rsn_resource = i16[1].data();
      for (int k = 0; k < 3; k++) i64[k].assign(rc, 0);
      d.rsn_a = i64[0].data(); d.rsn_b = i64[1].data(); d.rsn_c = i64[2].data();
    }
  }
};
}  // namespace

struct kq_group {
  int n = 0;
  std::vector<int> dev;
  std::vector<kq_engine*> eng;
  std::vector<void*> xbuf; std::vector<size_t> xwords;
  std::vector<hipStream_t> stream;
  std::vector<ncclComm_t> comm;
  std::vector<Scratch> scratch;
  Rccl rccl;
  int nR = 0;
  std::string last_error;
  int fail(int code, const std::string& m) { last_error = m; return code; }
};

template <class F> static int for_ranks(kq_group* g, F f) {  // f(rank) on one thread per device; the first error wins
  std::vector<int> rc((size_t)g->n, KQ_OK);
  std::vector<std::thread> th;
  for (int r = 1; r < g->n; r++) th.emplace_back([&, r] { (void)hipSetDevice(g->dev[r]); rc[r] = f(r); });
  (void)hipSetDevice(g->dev[0]);
  rc[0] = f(0);
  for (auto& t : th) t.join();
  for (int r = 0; r < g->n; r++) if (rc[r] != KQ_OK) { g->last_error = std::string("rank ") + std::to_string(r) + ": " + kq_last_error(g->eng[r]); return rc[r]; }
  return KQ_OK;
}

extern "C" {

int kq_group_create(const kq_config* cfg, int32_t n_dev, const int32_t* devices, kq_group** out) {
  if (!cfg || !out || n_dev < 1 || !devices) return KQ_EINVAL;
  try {
    kq_group* g = new (std::nothrow) kq_group();
    if (!g) return KQ_ENOMEM;
    g->n = n_dev;
    g->dev.assign(devices, devices + n_dev);
    for (int i = 0; i < n_dev; i++) for (int j = 0; j < i; j++) if (g->dev[i] == g->dev[j]) { delete g; return KQ_EINVAL; }
    g->eng.assign(n_dev, nullptr); g->xbuf.assign(n_dev, nullptr); g->xwords.assign(n_dev, 0); g->stream.assign(n_dev, nullptr);
    g->comm.assign(n_dev, nullptr); g->scratch.resize(n_dev);
    int rc = KQ_OK;
    for (int r = 0; r < n_dev && rc == KQ_OK; r++) {
      kq_config c = *cfg; c.device = g->dev[r];
      rc = kq_engine_create(&c, &g->eng[r]);
      if (rc == KQ_OK && hipSetDevice(g->dev[r]) == hipSuccess && hipStreamCreateWithFlags(&g->stream[r], hipStreamNonBlocking) != hipSuccess) rc = KQ_EDEVICE;
    }
    if (rc == KQ_OK && n_dev > 1) {
      std::string err;
      if (!g->rccl.load(&err)) { fprintf(stderr, "kq_group_create: %s\n", err.c_str()); rc = KQ_EDEVICE; }
      else {
        const ncclResult_t nr = g->rccl.CommInitAll(g->comm.data(), n_dev, g->dev.data());
        if (nr != ncclSuccess) { fprintf(stderr, "kq_group_create: ncclCommInitAll: %s\n", g->rccl.GetErrorString(nr)); rc = KQ_EDEVICE; }
      }
    }
    if (rc != KQ_OK) { kq_group_destroy(g); return rc; }
    *out = g;
    return KQ_OK;
  } catch (const std::bad_alloc&) { return KQ_ENOMEM; } catch (...) { return KQ_EINVAL; }
}

void kq_group_destroy(kq_group* g) {
  if (!g) return;
  try {
    for (int r = 0; r < g->n; r++) {
      (void)hipSetDevice(g->dev[r]);
      if (g->comm[r] && g->rccl.CommDestroy) (void)g->rccl.CommDestroy(g->comm[r]);
      if (g->xbuf[r]) (void)hipFree(g->xbuf[r]);
      if (g->stream[r]) (void)hipStreamDestroy(g->stream[r]);
      if (g->eng[r]) kq_engine_destroy(g->eng[r]);
    }
    delete g;
  } catch (...) {}
}

int kq_group_size(const kq_group* g) { return g ? g->n : 0; }
const char* kq_group_last_error(kq_group* g) { return g ? g->last_error.c_str() : "null group"; }

int kq_group_snapshot_put(kq_group* g, const kq_snapshot* s) {
  if (!g || !s) return KQ_EINVAL;
  try {
    g->nR = s->n_resource;
    return for_ranks(g, [&](int r) { return kq_snapshot_put(g->eng[r], s); });
  } catch (const std::bad_alloc&) { return g->fail(KQ_ENOMEM, "out of host memory"); } catch (const std::exception& x) { return g->fail(KQ_EINVAL, x.what()); }
}

int kq_group_cycle_run(kq_group* g, const kq_heads* h, kq_decisions* out) {
  if (!g || !h || !out) return KQ_EINVAL;
  try {
    const int n = g->n;
    if (n == 1) return kq_cycle_run(g->eng[0], h, out) == KQ_OK ? KQ_OK : g->fail(KQ_EDEVICE, kq_last_error(g->eng[0]));
    int64_t words = 0;
    int rc = kq_cycle_shard_words(g->eng[0], h, out, n, &words);
    if (rc != KQ_OK) return g->fail(rc, kq_last_error(g->eng[0]));
    std::vector<std::vector<uint8_t>> mine((size_t)n, std::vector<uint8_t>((size_t)std::max(h->n, 1), 0));
    for (int i = 0; i < h->n; i++) mine[(size_t)(i % n)][(size_t)i] = 1;   // dealt round-robin: consecutive heads are sibling ClusterQueues of similar cost
    for (int r = 1; r < n; r++) g->scratch[r].size(h, out, g->nR);
    // (1) sharded nominate -> exchange buffers
    rc = for_ranks(g, [&](int r) {
      if (g->xwords[r] < (size_t)words) {
        if (g->xbuf[r]) (void)hipFree(g->xbuf[r]);
        g->xbuf[r] = nullptr; g->xwords[r] = 0;
        const size_t cap = (size_t)words + (size_t)words / 8;
        if (hipMalloc(&g->xbuf[r], cap * 8) != hipSuccess) return (int)KQ_ENOMEM;
        g->xwords[r] = cap;
      }
      return kq_cycle_nominate_shard(g->eng[r], h, mine[(size_t)r].data(), n, r, g->xbuf[r], r == 0 ? out : &g->scratch[r].d);
    });
    if (rc != KQ_OK) return rc;
    // (2) the one collective of the cycle: sum of buffers with disjoint supports = gather of the nominations
    std::vector<int> nrc((size_t)n, 0);
    {
      std::vector<std::thread> th;
      auto one = [&](int r) {
        (void)hipSetDevice(g->dev[r]);
        const ncclResult_t a = g->rccl.AllReduce(g->xbuf[r], g->xbuf[r], (size_t)words, ncclInt64, ncclSum, g->comm[r], g->stream[r]);
        nrc[(size_t)r] = a != ncclSuccess ? (int)a : (hipStreamSynchronize(g->stream[r]) == hipSuccess ? 0 : -1);
      };
      for (int r = 1; r < n; r++) th.emplace_back(one, r);
      one(0);
      for (auto& t : th) t.join();
    }
    for (int r = 0; r < n; r++) if (nrc[(size_t)r] != 0) return g->fail(KQ_EDEVICE, std::string("ncclAllReduce on rank ") + std::to_string(r) + ": " + (nrc[(size_t)r] > 0 ? g->rccl.GetErrorString((ncclResult_t)nrc[(size_t)r]) : "stream error"));
    // (3) replicated order + processEntry over the merged batch
    return for_ranks(g, [&](int r) { return kq_cycle_process_merged(g->eng[r], n, r, g->xbuf[r], r == 0 ? out : &g->scratch[r].d); });
  } catch (const std::bad_alloc&) { return g->fail(KQ_ENOMEM, "out of host memory"); } catch (const std::exception& x) { return g->fail(KQ_EINVAL, x.what()); }
}

int kq_group_cycle_commit(kq_group* g, int32_t* n_admitted) {
  if (!g) return KQ_EINVAL;
  try {
    std::vector<int32_t> na((size_t)g->n, 0);
    const int rc = for_ranks(g, [&](int r) { return kq_cycle_commit(g->eng[r], &na[(size_t)r]); });
    if (n_admitted) *n_admitted = na[0];
    return rc;
  } catch (...) { return g->fail(KQ_EINVAL, "exception"); }
}
int kq_group_cycle_release(kq_group* g, int32_t age) {
  if (!g) return KQ_EINVAL;
  try { return for_ranks(g, [&](int r) { return kq_cycle_release(g->eng[r], age); }); } catch (...) { return g->fail(KQ_EINVAL, "exception"); }
}
int kq_group_read_usage(kq_group* g, int32_t rank, int64_t* usage) {
  if (!g || rank < 0 || rank >= g->n || !usage) return KQ_EINVAL;
  (void)hipSetDevice(g->dev[rank]);
  return kq_snapshot_read_planes(g->eng[rank], nullptr, usage, nullptr);
}

}  // extern "C"
