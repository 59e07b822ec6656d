// kq_group_core.hpp — the protocol of include/kq_group.h over an abstract backend: one engine per rank, the sharded nomination, ONE
// exchange of the nominations, processEntry replicated (scheduler.go:308-386 steps 3-5 over several devices of one process).
//
// The backend B supplies the engine calls (the C ABI of include/kq_engine.h on the device; the kqe_* twins of tests/emu in the CPU suite),
// device buffers and the device collective. What lives here is what both share and what round 4 shipped untested: the rank workers,
// the phase barrier, error containment, and the HOST collective seam (KQ_GROUP_HOST_COLLECTIVE: the exchange buffers are summed through
// host memory — no RCCL, the same device may appear twice — so that export -> reduce -> import -> kq_cycle_process_merged runs with two
// engines on ONE GPU and in the emulation).
//
// Threads: n - 1 PERSISTENT workers (rank r > 0), created with the group; the caller's thread is rank 0. A cycle is ONE job: every rank
// walks nominate -> exchange -> process on its own, meeting the others at the phase barrier only (round 4 created and joined n - 1
// threads three times per cycle). A rank that fails publishes its code before the barrier; every rank reads the verdict behind it and
// leaves together, so nobody waits in a collective for a rank that is gone.
#pragma once
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/kq_group.h"

namespace kqg {

// generation barrier for a fixed number of parties (C++17 has no std::barrier). The ranks of a cycle are a handful of threads that all
// arrive within microseconds of each other: spin first (a condition variable costs 10-20 us per wake-up, a cycle has five phases and lasts
// 0.4 ms), yield when the others are late.
class Barrier {
 public:
  void reset(int parties) { n_ = parties; count_.store(0); gen_.store(0); abort_.store(0); }
  // a rank that leaves its job any other way than through the barriers (an exception) calls abort(code): everybody waiting, or arriving
  // later, returns that code instead of waiting for a rank that is gone. reset() (the next job) clears it.
  void abort(int code) { abort_.store(code, std::memory_order_release); }
  int wait() {
    if (int a = abort_.load(std::memory_order_acquire)) return a;
    const uint64_t g = gen_.load(std::memory_order_acquire);
    if (count_.fetch_add(1, std::memory_order_acq_rel) + 1 == n_) {
      count_.store(0, std::memory_order_relaxed);
      gen_.store(g + 1, std::memory_order_release);
      return 0;
    }
    for (int spins = 0; gen_.load(std::memory_order_acquire) == g; spins++) {
      if (int a = abort_.load(std::memory_order_acquire)) return a;
      if (spins > 4000) std::this_thread::yield();
    }
    return 0;
  }
 private:
  std::atomic<int> count_{0}; std::atomic<uint64_t> gen_{0}; std::atomic<int> abort_{0}; int n_ = 1;
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

template <class B>
struct Group {
  B be;
  int n = 0;
  uint32_t flags = 0;
  std::vector<int> dev;
  std::vector<void*> eng;
  std::vector<void*> xbuf; std::vector<size_t> xwords;     // exchange buffers in device memory, one per rank
  std::vector<int64_t*> hbuf; size_t hwords = 0;           // host collective: one staging buffer per rank + the sum
  int64_t* hsum = nullptr;
  std::vector<Scratch> scratch;
  int nR = 0;
  int64_t host_sums = 0;   // exchanges that went through the host seam (kq_group_collective_info)
  std::string last_error;
  // ---- workers ----
  std::vector<std::thread> workers;
  std::mutex jm; std::condition_variable jcv, dcv;
  uint64_t job_gen = 0; int job_left = 0; bool quit = false;
  std::function<int(int)> job;
  std::vector<int> job_rc;
  Barrier bar;
  std::vector<int> phase_rc[2];   // what each rank publishes before a phase barrier (two sets: phase k + 1 is written while a late rank still reads phase k)
  std::vector<int> phase_no;      // phases this rank has been through in the running job
  std::atomic<int> origin{-1};    // the rank whose failure ended the running job at a phase barrier

  bool host_collective() const { return (flags & KQ_GROUP_HOST_COLLECTIVE) != 0; }
  int fail(int code, const std::string& m) { last_error = m; return code; }

  void worker(int r) {
    (void)be.set_device(dev[(size_t)r]);
    uint64_t seen = 0;
    for (;;) {
      std::function<int(int)> f;
      {
        std::unique_lock<std::mutex> lk(jm);
        jcv.wait(lk, [&] { return quit || job_gen != seen; });
        if (quit) return;
        seen = job_gen;
        f = job;
      }
      int rc;
      try { rc = f(r); } catch (const std::bad_alloc&) { rc = KQ_ENOMEM; bar.abort(rc); } catch (...) { rc = KQ_EINVAL; bar.abort(rc); }
      {
        std::lock_guard<std::mutex> lk(jm);
        job_rc[(size_t)r] = rc;
        if (--job_left == 0) dcv.notify_all();
      }
    }
  }
  // f(rank) on every rank — rank 0 on the caller's thread; the first failing rank's code wins. f must reach every phase barrier it
  // contains on every rank (see phase()).
  template <class F> int run(F f) {
    origin.store(-1);
    bar.reset(n);   // (nobody is inside the barrier between two jobs) — also clears an abort and the phase parity a broken job left behind
    std::fill(phase_no.begin(), phase_no.end(), 0);
    if (n > 1) {
      std::lock_guard<std::mutex> lk(jm);
      job = f; job_left = n - 1; job_gen++;
      std::fill(job_rc.begin(), job_rc.end(), KQ_OK);
      jcv.notify_all();
    }
    (void)be.set_device(dev[0]);
    int rc0;
    try { rc0 = f(0); } catch (const std::bad_alloc&) { rc0 = KQ_ENOMEM; bar.abort(rc0); } catch (...) { rc0 = KQ_EINVAL; bar.abort(rc0); }
    if (n > 1) {
      std::unique_lock<std::mutex> lk(jm);
      dcv.wait(lk, [&] { return job_left == 0; });
    }
    job_rc[0] = rc0;
    for (int r = 0; r < n; r++)
      if (job_rc[(size_t)r] != KQ_OK) {
        const int who = origin.load() >= 0 ? origin.load() : r;   // ranks that left at a barrier carry the failing rank's code, not their own
        const char* m = eng[(size_t)who] ? be.last_error(eng[(size_t)who]) : "";
        last_error = std::string("rank ") + std::to_string(who) + ": " + (m ? m : "");
        return job_rc[(size_t)r];
      }
    return KQ_OK;
  }
  // end of a phase: publish this rank's code, meet the others, return the first failure of ANY rank (so that all ranks leave together)
  int phase(int r, int rc) {
    std::vector<int>& pr = phase_rc[(size_t)(phase_no[(size_t)r]++ & 1)];
    pr[(size_t)r] = rc;
    if (const int gone = bar.wait()) return gone;   // a rank left through an exception: its code, nobody waits for it
    for (int q = 0; q < n; q++) if (pr[(size_t)q] != KQ_OK) { int none = -1; origin.compare_exchange_strong(none, q); return pr[(size_t)q]; }   // (every rank finds the same q; the first failing phase names the rank)
    return KQ_OK;
  }

  int create(const kq_config* cfg, int n_dev, const int32_t* devices, uint32_t fl) {
    n = n_dev; flags = fl;
    dev.assign(devices, devices + n_dev);
    if (!host_collective())
      for (int i = 0; i < n_dev; i++) for (int j = 0; j < i; j++) if (dev[(size_t)i] == dev[(size_t)j]) { n = 0; return fail(KQ_EINVAL, "one rank per device: a device is listed twice (KQ_GROUP_HOST_COLLECTIVE lifts this)"); }   // RCCL: one rank per device; n = 0: nothing was created, destroy() has nothing to walk
    eng.assign((size_t)n_dev, nullptr); xbuf.assign((size_t)n_dev, nullptr); xwords.assign((size_t)n_dev, 0);
    hbuf.assign((size_t)n_dev, nullptr); scratch.resize((size_t)n_dev); job_rc.assign((size_t)n_dev, KQ_OK); phase_rc[0].assign((size_t)n_dev, KQ_OK); phase_rc[1].assign((size_t)n_dev, KQ_OK); phase_no.assign((size_t)n_dev, 0);
    bar.reset(n_dev);
    for (int r = 0; r < n_dev; r++) {
      if (!be.set_device(dev[(size_t)r])) return fail(KQ_EDEVICE, std::string("device ") + std::to_string(dev[(size_t)r]) + " cannot be selected");
      kq_config c = *cfg; c.device = dev[(size_t)r];
      const int rc = be.engine_create(&c, &eng[(size_t)r]);
      if (rc != KQ_OK) return rc;
      const int rs = be.rank_init(r, dev[(size_t)r]);
      if (rs != KQ_OK) return rs;
    }
    if (n_dev > 1 && !host_collective()) {
      std::string err;
      const int rc = be.comm_init(n_dev, dev.data(), &err);
      if (rc != KQ_OK) return fail(rc, err);
    }
    for (int r = 1; r < n_dev; r++) workers.emplace_back([this, r] { worker(r); });
    return KQ_OK;
  }
  void destroy() {
    {
      std::lock_guard<std::mutex> lk(jm);
      quit = true;
      jcv.notify_all();
    }
    for (auto& t : workers) if (t.joinable()) t.join();
    workers.clear();
    be.comm_destroy();
    for (int r = 0; r < n && (size_t)r < eng.size(); r++) {   // (a create() that failed half-way leaves null entries, one that failed before sizing leaves none)
      (void)be.set_device(dev[(size_t)r]);
      if (xbuf[(size_t)r]) be.xfree(xbuf[(size_t)r]);
      if (hbuf[(size_t)r]) be.host_free(hbuf[(size_t)r]);
      be.rank_fini(r);
      if (eng[(size_t)r]) be.engine_destroy(eng[(size_t)r]);
    }
    if (hsum) be.host_free(hsum);
    hsum = nullptr;
  }

  int snapshot_put(const kq_snapshot* s) {
    nR = s->n_resource;
    last_error.clear();
    return run([&](int r) { return be.snapshot_put(eng[(size_t)r], s); });
  }

  int ensure_xbuf(int r, int64_t words) {
    if (xwords[(size_t)r] >= (size_t)words) return KQ_OK;
    if (xbuf[(size_t)r]) be.xfree(xbuf[(size_t)r]);
    xbuf[(size_t)r] = nullptr; xwords[(size_t)r] = 0;
    const size_t cap = (size_t)words + (size_t)words / 8;
    xbuf[(size_t)r] = be.xalloc(cap * 8);
    if (!xbuf[(size_t)r]) return KQ_ENOMEM;
    xwords[(size_t)r] = cap;
    return KQ_OK;
  }
  int ensure_host(int64_t words) {   // caller's thread, before the job
    if (hwords >= (size_t)words) return KQ_OK;
    const size_t cap = (size_t)words + (size_t)words / 8;
    for (int r = 0; r < n; r++) { if (hbuf[(size_t)r]) be.host_free(hbuf[(size_t)r]); hbuf[(size_t)r] = (int64_t*)be.host_alloc(cap * 8); if (!hbuf[(size_t)r]) { hwords = 0; return KQ_ENOMEM; } }
    if (hsum) be.host_free(hsum);
    hsum = (int64_t*)be.host_alloc(cap * 8);
    if (!hsum) { hwords = 0; return KQ_ENOMEM; }
    hwords = cap;
    return KQ_OK;
  }
  // SUM over the ranks of xbuf[0 .. words) into every rank's xbuf, through host memory: rank r adds slice r of all staging buffers
  int exchange_host(int r, int64_t words) {
    int rc = be.d2h(hbuf[(size_t)r], xbuf[(size_t)r], (size_t)words * 8);
    rc = phase(r, rc);
    if (rc != KQ_OK) return rc;
    const int64_t lo = words * r / n, hi = words * (r + 1) / n;
    for (int64_t i = lo; i < hi; i++) {
      uint64_t s = 0;   // wrap-around like ncclSum on int64 (the supports are disjoint: no carry ever happens)
      for (int q = 0; q < n; q++) s += (uint64_t)hbuf[(size_t)q][i];
      hsum[i] = (int64_t)s;
    }
    rc = phase(r, KQ_OK);
    if (rc != KQ_OK) return rc;
    return be.h2d(xbuf[(size_t)r], hsum, (size_t)words * 8);
  }
  // the device collective: issued for ALL ranks by rank 0 inside one group call (a rank whose enqueue fails cannot leave the others
  // waiting in theirs), then every rank waits for its own stream
  int exchange_device(int r, int64_t words, int* issue_rc, std::string* issue_err) {
    int rc = phase(r, KQ_OK);   // every xbuf is written (kq_cycle_nominate_shard is synchronous)
    if (rc != KQ_OK) return rc;
    if (r == 0) *issue_rc = be.allreduce_all(n, dev.data(), xbuf.data(), (size_t)words, issue_err);
    rc = phase(r, r == 0 ? *issue_rc : KQ_OK);
    if (rc != KQ_OK) return rc;
    (void)be.set_device(dev[(size_t)r]);
    return be.comm_wait(r);
  }

  int cycle_run(const kq_heads* h, kq_decisions* out) {
    last_error.clear();
    if (n == 1 && !(flags & KQ_GROUP_FORCE_SHARDED)) {
      const int rc = be.cycle_run(eng[0], h, out);   // the engine's own code: KQ_ECAPACITY stays KQ_ECAPACITY (grow and retry), as for n > 1
      return rc == KQ_OK ? KQ_OK : fail(rc, be.last_error(eng[0]));
    }
    int64_t words = 0;
    int rc = be.shard_words(eng[0], h, out, n, &words);
    if (rc != KQ_OK) return fail(rc, be.last_error(eng[0]));
    if (host_collective() && n > 1) { rc = ensure_host(words); if (rc != KQ_OK) return fail(rc, "out of host memory"); }
    std::vector<std::vector<uint8_t>> mine((size_t)n, std::vector<uint8_t>((size_t)std::max(h->n, 1), 0));
    for (int i = 0; i < h->n; i++) mine[(size_t)(i % n)][(size_t)i] = 1;   // dealt round-robin: consecutive heads are sibling ClusterQueues of similar cost
    for (int r = 1; r < n; r++) scratch[(size_t)r].size(h, out, nR);
    int issue_rc = KQ_OK; std::string issue_err;
    rc = run([&](int r) {
      kq_decisions* d = r == 0 ? out : &scratch[(size_t)r].d;
      // (1) sharded nominate -> exchange buffer of this rank (zero outside its own heads)
      int c = ensure_xbuf(r, words);
      if (c == KQ_OK) c = be.nominate_shard(eng[(size_t)r], h, mine[(size_t)r].data(), n, r, xbuf[(size_t)r], d);
      c = phase(r, c);
      if (c != KQ_OK) return c;
      // (2) the one collective of the cycle: sum of buffers with disjoint supports = gather of the nominations
      if (n > 1) {
        if (r == 0 && host_collective()) host_sums++;
        c = host_collective() ? exchange_host(r, words) : exchange_device(r, words, &issue_rc, &issue_err);
        c = phase(r, c);
        if (c != KQ_OK) return c;
      }
      // (3) replicated order + processEntry over the merged batch
      return be.process_merged(eng[(size_t)r], n, r, xbuf[(size_t)r], d);
    });
    if (rc != KQ_OK && issue_rc != KQ_OK) last_error = issue_err;
    return rc;
  }
  int cycle_commit(int32_t* n_admitted) {
    last_error.clear();
    std::vector<int32_t> na((size_t)n, 0);
    const int rc = run([&](int r) { return be.commit(eng[(size_t)r], &na[(size_t)r]); });
    if (n_admitted) *n_admitted = na[0];
    return rc;
  }
  int cycle_release(int32_t age) { last_error.clear(); return run([&](int r) { return be.release(eng[(size_t)r], age); }); }
  int read_usage(int rank, int64_t* usage) {
    if (rank < 0 || rank >= n || !usage) return KQ_EINVAL;
    (void)be.set_device(dev[(size_t)rank]);
    return be.read_usage(eng[(size_t)rank], usage);
  }
};

}  // namespace kqg
