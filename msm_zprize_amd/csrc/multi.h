// Multi-GPU context: one engine + one HIP stream + one host thread per device, inputs split over the
// devices, partial sums folded on the host.  This is what replaces the reference's worker pool
// (src/threads/threads.ts:132-359: startThreads(n) spawns n-1 workers that run the same msm body on
// their share of the work, src/parallel.ts:291-315) -- here startThreads(n) -> msmz_create(..., n_devices = n).
//
// Input split (MSM is additive over disjoint index sets): contiguous blocks of
// 2^MULTI_BLOCK_SHIFT entries are dealt round-robin, entry i lives on device (i >> shift) % G at local index
// ((i >> shift) / G << shift) | (i & mask).  Unlike one contiguous range per device this does not depend on the
// size of the set, so (1) a point set and a scalar set of different lengths are split consistently and (2) the
// first n entries of a set are a PREFIX of every device's local array -- msm(scalars, points, n) with
// n <= allocated (msm-batched-affine.ts:74-97; the warm-up of scripts/msm-weierstrass.ts:24) needs no data
// movement.  No inter-GPU traffic during an MSM; the G affine partial sums are added with msmz_point_add.
#pragma once
#include <condition_variable>
#include <cstring>
#include <map>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/msmz.h"

namespace msmz {

constexpr int MULTI_BLOCK_SHIFT = 16;

// how device-side generators map a local index to the global (seeded) index
struct GenMap {
  uint32_t nshards = 1, shard = 0;
  int blk_shift = MULTI_BLOCK_SHIFT;
};

class IEngine {
 public:
  virtual ~IEngine() {}
  // `split` (uploads and host-scalar MSMs): the engine is one shard of a multi-device context and `n` counts its LOCAL
  // records; the host buffer is the caller's whole array, from which the engine copies its own blocks (Engine::copy_h2d)
  virtual int upload_points(const uint8_t* xy, const uint8_t* inf, uint64_t n, uint64_t* h, const GenMap* split = nullptr) = 0;
  virtual int upload_scalars(const uint8_t* s, uint64_t n, uint64_t* h, const GenMap* split = nullptr) = 0;
  virtual int random_points(uint64_t n, uint64_t seed, const GenMap& map, uint64_t* h) = 0;
  virtual int random_scalars(uint64_t n, uint64_t seed, const GenMap& map, uint64_t* h) = 0;
  virtual int download_points(uint64_t h, uint64_t first, uint64_t count, uint8_t* xy, uint8_t* inf) = 0;
  virtual int download_scalars(uint64_t h, uint64_t first, uint64_t count, uint8_t* s) = 0;
  virtual int free_handle(uint64_t h) = 0;
  virtual int msm(uint64_t ph, const uint8_t* host_scalars, uint64_t sh, uint64_t n, const msmz_opts* o, uint8_t* out,
                  int* out_inf, msmz_log* log, const GenMap* split = nullptr) = 0;
  // stage-level test hooks (include/msmz_test.h)
  virtual int test_set_glv_bits(int) { return MSMZ_ERR_UNSUPPORTED; }
  virtual int test_retries() { return 0; }
  virtual int test_field(int, const uint8_t*, const uint8_t*, uint64_t, uint8_t*) { return MSMZ_ERR_UNSUPPORTED; }
  virtual int test_glv(const uint8_t*, uint64_t, uint8_t*, uint8_t*, uint8_t*) { return MSMZ_ERR_UNSUPPORTED; }
  virtual int test_digits(const uint8_t*, uint64_t, int, int, int, uint32_t*) { return MSMZ_ERR_UNSUPPORTED; }
  virtual int test_sort(const uint8_t*, uint64_t, int, int, int, uint32_t*, uint32_t*, uint64_t, uint32_t*, uint64_t) {
    return MSMZ_ERR_UNSUPPORTED;
  }
  virtual int test_point(int, const uint8_t*, const uint8_t*, const uint8_t*, const uint8_t*, uint64_t, uint8_t*) {
    return MSMZ_ERR_UNSUPPORTED;
  }
};

// entries of the first n that live on shard g of G
static inline uint64_t shard_count(uint64_t n, uint32_t g, uint32_t G, int shift = MULTI_BLOCK_SHIFT) {
  const uint64_t blk = 1ull << shift, cycle = blk * G;
  const uint64_t full = n / cycle, rem = n % cycle;
  uint64_t extra = rem > (uint64_t)g * blk ? rem - (uint64_t)g * blk : 0;
  if (extra > blk) extra = blk;
  return full * blk + extra;
}

class MultiEngine : public IEngine {
 public:
  // takes ownership of the engines (already initialised, one per device id; ids may repeat)
  MultiEngine(int curve_id, int fe_bytes, std::vector<IEngine*> engines)
      : curve_id_(curve_id), fb_(fe_bytes), G_((uint32_t)engines.size()) {
    for (IEngine* e : engines) workers_.push_back(new Worker(e));
  }
  ~MultiEngine() override {
    for (Worker* w : workers_) delete w;
  }

  int upload_points(const uint8_t* xy, const uint8_t* inf, uint64_t n, uint64_t* h, const GenMap* = nullptr) override {
    if (!xy || !h || n == 0) return MSMZ_ERR_ARG;
    MHandle mh{0, n, std::vector<uint64_t>(G_, 0)};
    int st = for_all([&](uint32_t g, IEngine* e) {
      const uint64_t cnt = shard_count(n, g, G_);
      if (cnt == 0) return (int)MSMZ_OK;
      // every device copies its own blocks straight out of the caller's buffer (no gathered host copy)
      const GenMap split{G_, g, MULTI_BLOCK_SHIFT};
      return e->upload_points(xy, inf, cnt, &mh.sub[g], &split);
    });
    return finish_handle(st, mh, h);
  }

  int upload_scalars(const uint8_t* s, uint64_t n, uint64_t* h, const GenMap* = nullptr) override {
    if (!s || !h || n == 0) return MSMZ_ERR_ARG;
    MHandle mh{1, n, std::vector<uint64_t>(G_, 0)};
    int st = for_all([&](uint32_t g, IEngine* e) {
      const uint64_t cnt = shard_count(n, g, G_);
      if (cnt == 0) return (int)MSMZ_OK;
      const GenMap split{G_, g, MULTI_BLOCK_SHIFT};
      return e->upload_scalars(s, cnt, &mh.sub[g], &split);
    });
    return finish_handle(st, mh, h);
  }

  int random_points(uint64_t n, uint64_t seed, const GenMap&, uint64_t* h) override {
    if (!h || n == 0) return MSMZ_ERR_ARG;
    MHandle mh{0, n, std::vector<uint64_t>(G_, 0)};
    int st = for_all([&](uint32_t g, IEngine* e) {
      const uint64_t cnt = shard_count(n, g, G_);
      if (cnt == 0) return (int)MSMZ_OK;
      return e->random_points(cnt, seed, GenMap{G_, g, MULTI_BLOCK_SHIFT}, &mh.sub[g]);
    });
    return finish_handle(st, mh, h);
  }

  int random_scalars(uint64_t n, uint64_t seed, const GenMap&, uint64_t* h) override {
    if (!h || n == 0) return MSMZ_ERR_ARG;
    MHandle mh{1, n, std::vector<uint64_t>(G_, 0)};
    int st = for_all([&](uint32_t g, IEngine* e) {
      const uint64_t cnt = shard_count(n, g, G_);
      if (cnt == 0) return (int)MSMZ_OK;
      return e->random_scalars(cnt, seed, GenMap{G_, g, MULTI_BLOCK_SHIFT}, &mh.sub[g]);
    });
    return finish_handle(st, mh, h);
  }

  int download_points(uint64_t hd, uint64_t first, uint64_t count, uint8_t* xy, uint8_t* inf) override {
    auto it = handles_.find(hd);
    if (it == handles_.end() || it->second.kind != 0 || !xy) return MSMZ_ERR_ARG;
    if (first > it->second.n || count > it->second.n - first) return MSMZ_ERR_ARG;   // (first + count can wrap)
    const size_t rec = 2 * (size_t)fb_;
    return for_range(it->second, first, count, [&](IEngine* e, uint64_t sub, uint64_t li, uint64_t gi, uint64_t len) {
      return e->download_points(sub, li, len, xy + (gi - first) * rec, inf ? inf + (gi - first) : nullptr);
    });
  }

  int download_scalars(uint64_t hd, uint64_t first, uint64_t count, uint8_t* s) override {
    auto it = handles_.find(hd);
    if (it == handles_.end() || it->second.kind != 1 || !s) return MSMZ_ERR_ARG;
    if (first > it->second.n || count > it->second.n - first) return MSMZ_ERR_ARG;
    return for_range(it->second, first, count, [&](IEngine* e, uint64_t sub, uint64_t li, uint64_t gi, uint64_t len) {
      return e->download_scalars(sub, li, len, s + (gi - first) * 32);
    });
  }

  int free_handle(uint64_t hd) override {
    auto it = handles_.find(hd);
    if (it == handles_.end()) return MSMZ_ERR_ARG;
    for (uint32_t g = 0; g < G_; g++)
      if (it->second.sub[g]) (void)workers_[g]->eng->free_handle(it->second.sub[g]);
    handles_.erase(it);
    return MSMZ_OK;
  }

  int msm(uint64_t ph, const uint8_t* host_scalars, uint64_t sh, uint64_t n, const msmz_opts* o, uint8_t* out,
          int* out_inf, msmz_log* log, const GenMap* = nullptr) override {
    if (!out || !out_inf || n == 0) return MSMZ_ERR_ARG;
    auto pit = handles_.find(ph);
    if (pit == handles_.end() || pit->second.kind != 0 || pit->second.n < n) return MSMZ_ERR_ARG;
    const MHandle* sc = nullptr;
    if (!host_scalars) {
      auto sit = handles_.find(sh);
      if (sit == handles_.end() || sit->second.kind != 1 || sit->second.n < n) return MSMZ_ERR_ARG;
      sc = &sit->second;
    }
    const size_t rec = 2 * (size_t)fb_;
    std::vector<std::vector<uint8_t>> part(G_, std::vector<uint8_t>(rec));
    std::vector<int> pinf(G_, 1), used(G_, 0);
    std::vector<msmz_log> logs(G_);
    const MHandle& pts = pit->second;
    int st = for_all([&](uint32_t g, IEngine* e) {
      const uint64_t cnt = shard_count(n, g, G_);
      if (cnt == 0) return (int)MSMZ_OK;
      used[g] = 1;
      if (sc) return e->msm(pts.sub[g], nullptr, sc->sub[g], cnt, o, part[g].data(), &pinf[g], &logs[g]);
      const GenMap split{G_, g, MULTI_BLOCK_SHIFT};   // host scalars: the device copies its own blocks of the caller's buffer
      return e->msm(pts.sub[g], host_scalars, 0, cnt, o, part[g].data(), &pinf[g], &logs[g], &split);
    });
    if (st) return st;
    // fold the partial sums (the reference's "partition sum" on the main thread, msm-batched-affine.ts:300-307)
    bool first = true;
    for (uint32_t g = 0; g < G_; g++) {
      if (!used[g]) continue;
      if (first) {
        memcpy(out, part[g].data(), rec);
        *out_inf = pinf[g];
        first = false;
        continue;
      }
      std::vector<uint8_t> acc(out, out + rec);
      const int ai = *out_inf;
      st = msmz_point_add(curve_id_, ai ? nullptr : acc.data(), ai, pinf[g] ? nullptr : part[g].data(), pinf[g], out,
                          out_inf);
      if (st) return st;
    }
    if (log) {   // stage times: the slowest device; counts: summed
      memset(log, 0, sizeof(*log));
      for (uint32_t g = 0; g < G_; g++) {
        if (!used[g]) continue;
        for (int i = 0; i < MSMZ_N_STAGES; i++)
          if (logs[g].stage_ms[i] > log->stage_ms[i]) log->stage_ms[i] = logs[g].stage_ms[i];
        for (int i = 0; i < 32; i++)
          if (logs[g].batch_add_ms[i] > log->batch_add_ms[i]) log->batch_add_ms[i] = logs[g].batch_add_ms[i];
        if (logs[g].scatter_kernel_ms > log->scatter_kernel_ms) log->scatter_kernel_ms = logs[g].scatter_kernel_ms;
        if (logs[g].max_bucket > log->max_bucket) log->max_bucket = logs[g].max_bucket;
        if (logs[g].rounds > log->rounds) log->rounds = logs[g].rounds;
        log->n_entries += logs[g].n_entries;
        log->n_pairs += logs[g].n_pairs;
        log->scatter_launches += logs[g].scatter_launches;
        log->c = logs[g].c;
        log->K = logs[g].K;
        log->glv = logs[g].glv;
      }
    }
    return MSMZ_OK;
  }

  int test_set_glv_bits(int bits) override {
    int st = MSMZ_OK;
    for (Worker* w : workers_) {
      const int s = w->eng->test_set_glv_bits(bits);
      if (s && !st) st = s;
    }
    return st;
  }
  int test_retries() override {
    int r = 0;
    for (Worker* w : workers_) r += w->eng->test_retries();
    return r;
  }
  int test_field(int op, const uint8_t* a, const uint8_t* b, uint64_t n, uint8_t* out) override {
    return workers_[0]->eng->test_field(op, a, b, n, out);
  }
  int test_glv(const uint8_t* s, uint64_t n, uint8_t* s0, uint8_t* s1, uint8_t* neg) override {
    return workers_[0]->eng->test_glv(s, n, s0, s1, neg);
  }
  int test_digits(const uint8_t* s, uint64_t n, int c, int K, int glv, uint32_t* d) override {
    return workers_[0]->eng->test_digits(s, n, c, K, glv, d);
  }
  int test_sort(const uint8_t* s, uint64_t n, int c, int glv, int fb, uint32_t* geom, uint32_t* off, uint64_t oc,
                uint32_t* refs, uint64_t rc) override {
    return workers_[0]->eng->test_sort(s, n, c, glv, fb, geom, off, oc, refs, rc);
  }
  int test_point(int op, const uint8_t* a, const uint8_t* ai, const uint8_t* b, const uint8_t* bi, uint64_t n,
                 uint8_t* out) override {
    return workers_[0]->eng->test_point(op, a, ai, b, bi, n, out);
  }

 private:
  struct MHandle {
    int kind;
    uint64_t n;
    std::vector<uint64_t> sub;   // per-device handle (0 = that device holds nothing)
  };

  // one persistent host thread per device: runs the tasks posted for its engine
  struct Worker {
    IEngine* eng;
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::function<int()> task;
    bool has_task = false, done = false, stop = false;
    int status = 0;
    explicit Worker(IEngine* e) : eng(e) {
      th = std::thread([this] {
        std::unique_lock<std::mutex> lk(mu);
        while (true) {
          cv.wait(lk, [this] { return has_task || stop; });
          if (stop) return;
          std::function<int()> f = std::move(task);
          has_task = false;
          lk.unlock();
          const int st = f();
          lk.lock();
          status = st;
          done = true;
          cv.notify_all();
        }
      });
    }
    ~Worker() {
      {
        std::lock_guard<std::mutex> lk(mu);
        stop = true;
      }
      cv.notify_all();
      th.join();
      delete eng;
    }
    void post(std::function<int()> f) {
      std::lock_guard<std::mutex> lk(mu);
      task = std::move(f);
      has_task = true;
      done = false;
      cv.notify_all();
    }
    int wait() {
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, [this] { return done; });
      return status;
    }
  };

  // run fn(g, engine) on every device's thread concurrently; first non-zero status wins
  int for_all(const std::function<int(uint32_t, IEngine*)>& fn) {
    for (uint32_t g = 0; g < G_; g++) {
      Worker* w = workers_[g];
      w->post([&fn, g, w] { return fn(g, w->eng); });
    }
    int st = MSMZ_OK;
    for (uint32_t g = 0; g < G_; g++) {
      const int s = workers_[g]->wait();
      if (s && !st) st = s;
    }
    return st;
  }

  // pieces of the global range [first, first + count): fn(engine, sub handle, local index, global index, length)
  int for_range(const MHandle& mh, uint64_t first, uint64_t count,
                const std::function<int(IEngine*, uint64_t, uint64_t, uint64_t, uint64_t)>& fn) {
    const uint64_t blk = 1ull << MULTI_BLOCK_SHIFT;
    uint64_t gi = first;
    const uint64_t end = first + count;
    while (gi < end) {
      const uint64_t b = gi >> MULTI_BLOCK_SHIFT;
      const uint32_t g = (uint32_t)(b % G_);
      uint64_t len = ((b + 1) << MULTI_BLOCK_SHIFT) - gi;
      if (len > end - gi) len = end - gi;
      const uint64_t li = ((b / G_) << MULTI_BLOCK_SHIFT) | (gi & (blk - 1));
      const int st = fn(workers_[g]->eng, mh.sub[g], li, gi, len);
      if (st) return st;
      gi += len;
    }
    return MSMZ_OK;
  }

  int finish_handle(int st, MHandle& mh, uint64_t* h) {
    if (st) {
      for (uint32_t g = 0; g < G_; g++)
        if (mh.sub[g]) (void)workers_[g]->eng->free_handle(mh.sub[g]);
      return st;
    }
    *h = next_handle_++;
    handles_[*h] = std::move(mh);
    return MSMZ_OK;
  }

  int curve_id_, fb_;
  uint32_t G_;
  std::vector<Worker*> workers_;
  std::map<uint64_t, MHandle> handles_;
  uint64_t next_handle_ = 1;
};

}  // namespace msmz
