// Host-side MSM engine: owns the device buffers of one GPU, sequences the kernels on one HIP stream
// and finishes the K window sums on the host.  This replaces the reference's SPMD worker runtime
// (src/threads/threads.ts:132-359, src/parallel.ts:291-320) and the JS orchestration inside
// `msm` (src/msm-batched-affine.ts:74-328): barriers between phases become stream order, the
// per-thread bucket split (msm-common.ts:88-188) becomes grid sizing.
#pragma once
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <type_traits>
#include <vector>

#include "../../include/msmz.h"
#include "kernels.h"
#include "gen_kernels.h"
#include "test_kernels.h"
#include "host64.h"
#include "multi.h"

namespace msmz {

#define MSMZ_HIP(x)                                                                        \
  do {                                                                                     \
    hipError_t e_ = (x);                                                                   \
    if (e_ != hipSuccess) {                                                                \
      fprintf(stderr, "msmz: HIP error '%s' from `%s` at %s:%d\n", hipGetErrorString(e_), #x, __FILE__, __LINE__); \
      return MSMZ_ERR_HIP;                                                                 \
    }                                                                                      \
  } while (0)

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  int ensure(size_t need) {   // grow-only
    if (need <= bytes) return MSMZ_OK;
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
    size_t sz = need + need / 8;
    if (hipMalloc(&p, sz) != hipSuccess) {
      if (hipMalloc(&p, need) != hipSuccess) return MSMZ_ERR_HIP;
      sz = need;
    }
    bytes = sz;
    return MSMZ_OK;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
  }
  template <class T>
  T* as() const {
    return reinterpret_cast<T*>(p);
  }
};

struct Handle {
  int kind;        // 0 = points, 1 = scalars
  uint64_t n;
  bool has_endo;   // points: records [n, 2n) hold the endomorphism images
  void* dev;
};

static inline int ceil_log2_u64(uint64_t x) {
  int r = 0;
  while (((uint64_t)1 << r) < x) r++;
  return r;
}

// default window size.  The reference's tables (msm-common.ts:8-57) were tuned for 16 CPU threads;
// on the GPU the accumulate phase costs ~N*K additions and the reduction ~2*K*2^(c-1), and the latter
// is latency-bound, so c stays well below log2(N).
static inline int default_window(uint64_t n_points) {
  int lg = ceil_log2_u64(n_points < 2 ? 2 : n_points);
  int c = lg - 3;
  if (c < 3) c = 3;
  if (c > 17) c = 17;   // 2^16 buckets per window: the largest the two-level LDS sort handles in one coarse pass
  return c;
}

constexpr int MSMZ_ERR_RETRY_BITS = 1000;   // internal: repeat the MSM with one more scalar bit (never leaves the engine)

// Host-side group addition of two canonical affine points (partial sums of index ranges / of GPUs):
// the reference's "partition sum" on the main thread (msm-batched-affine.ts:300-307).
template <class F, bool TE>
static int host_point_add(const uint8_t* a, int ai, const uint8_t* b, int bi, uint8_t* out, int* oi) {
  constexpr int NW = F::NW;
  if constexpr (TE) {
    auto load = [](TeExt<F>& p, const uint8_t* xy) {
      uint32_t w[2 * NW];
      memcpy(w, xy, sizeof(w));
      Fe<F> x, y;
      fe_unpack<F>(x, w);
      fe_unpack<F>(y, w + NW);
      fe_to_mont(p.X, x);
      fe_to_mont(p.Y, y);
      fe_set_const<F>(p.Z, F::ONE);
      fe_mul(p.T, p.X, p.Y);
    };
    if (!a || !b) return MSMZ_ERR_ARG;   // twisted Edwards has no infinity flag: the identity is (0, 1)
    TeExt<F> p, q, r;
    load(p, a);
    load(q, b);
    te_add(r, p, q);
    uint32_t w[2 * NW];
    te_to_affine_canon<F>(w, r);
    memcpy(out, w, sizeof(w));
    *oi = 0;
  } else {
    auto load = [](Xyzz<F>& p, const uint8_t* xy, int inf) {
      if (inf) {
        xyzz_set_inf(p);
        return;
      }
      uint32_t w[2 * NW];
      memcpy(w, xy, sizeof(w));
      Affine<F> t, m;
      fe_unpack<F>(t.x, w);
      fe_unpack<F>(t.y, w + NW);
      fe_to_mont(m.x, t.x);
      fe_to_mont(m.y, t.y);
      xyzz_from_affine(p, m);
    };
    Xyzz<F> p, q, r;
    load(p, a, ai);
    load(q, b, bi);
    xyzz_add(r, p, q);
    uint32_t w[2 * NW];
    bool inf = xyzz_to_affine_canon<F>(w, r);
    memcpy(out, w, sizeof(w));
    *oi = inf ? 1 : 0;
  }
  return MSMZ_OK;
}

template <class Cfg>
class Engine : public IEngine {
  using F = typename Cfg::F;
  using Fr = typename Cfg::Fr;
  static constexpr int NW = F::NW;
  static constexpr int RW = 2 * NW;      // affine record words
  static constexpr int XW = 4 * NW;      // XYZZ / extended record words
  static constexpr int FE_BYTES = NW * 4;
  static constexpr bool TE = Cfg::TE;
  static constexpr int PW_WORDS = TE ? 4 * NW : PointFmt<F>::STRIDE;   // words between the records of a resident point set

 public:
  explicit Engine(int device) : device_(device) {}

  int init() {
    MSMZ_HIP(hipSetDevice(device_));
    MSMZ_HIP(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
    for (auto& e : ev_) MSMZ_HIP(hipEventCreate(&e));
    MSMZ_HIP(hipHostMalloc(&h_meta_, sizeof(MsmMeta)));
    MSMZ_HIP(hipHostMalloc(&h_final_, (size_t)3 * kMaxWindows * XW * 4));   // [kMaxWindows | up to 2 * kMaxWindows results]
    // Kernels that stage more than the default dynamic-LDS allowance get their limit raised ONCE, here, right after
    // hipSetDevice -- not lazily inside the first MSM and not on every MSM.  static + dynamic LDS is checked against
    // the device's per-workgroup LDS, so a kernel that cannot launch fails context creation with its name.
    int st;
    if ((st = raise_lds_limit((const void*)k_fine, "k_fine", kFineLds))) return st;
    if ((st = raise_sort_limits<false, 0>()) || (st = raise_sort_limits<false, 16>()) || (st = raise_sort_limits<false, 17>())) return st;
    if constexpr (Fr::HAS_GLV) {
      if ((st = raise_sort_limits<true, 0>()) || (st = raise_sort_limits<true, 16>())) return st;
    }
    return meta_.ensure(sizeof(MsmMeta) + kTraceBytes * 65536);
  }
  template <bool GLV, int C>
  int raise_sort_limits() {
    int st;
    if ((st = raise_lds_limit((const void*)k_coarse<Fr, GLV, C>, GLV ? "k_coarse<glv>" : "k_coarse", kCoarseLdsMax))) return st;
    return raise_lds_limit((const void*)k_hist<Fr, GLV, C>, GLV ? "k_hist<glv>" : "k_hist", kHistLdsMax);
  }
  // dynamic LDS the sort kernels may be launched with (sort_phase never asks for more: SORT_MAX_BINS caps nbins)
  static constexpr size_t kFineLds = ((size_t)(1 << FINE_MAX_BITS) + FINE_STAGE) * 4;
  static constexpr size_t kCoarseLdsMax = (size_t)2 * SORT_MAX_BINS * 4;
  static constexpr size_t kHistLdsMax = (size_t)SORT_MAX_BINS * 4;
#ifdef MSMZ_TRACE
  static constexpr size_t kTraceBytes = 128;   // per workgroup, behind the buffers the sort kernels receive (tools/wg_timeline.py)
#else
  static constexpr size_t kTraceBytes = 0;
#endif

#ifdef MSMZ_TRACE
  // appends one section {name[32], n, n x 16 stamps} to the file MSMZ_TRACE_OUT names (`first` truncates it)
  int trace_dump(const char* name, const void* d_stamps, uint32_t n_wgs, bool first) {
    const char* path = getenv("MSMZ_TRACE_OUT");
    if (!path) return MSMZ_OK;
    MSMZ_HIP(hipStreamSynchronize(stream_));
    std::vector<uint64_t> t((size_t)n_wgs * 16);
    MSMZ_HIP(hipMemcpy(t.data(), d_stamps, t.size() * 8, hipMemcpyDeviceToHost));
    if (FILE* f = fopen(path, first ? "wb" : "ab")) {
      char nm[32] = {};
      strncpy(nm, name, 31);
      const uint64_t n = n_wgs;
      fwrite(nm, 1, 32, f);
      fwrite(&n, 8, 1, f);
      fwrite(t.data(), 8, t.size(), f);
      fclose(f);
    }
    return MSMZ_OK;
  }
#endif

  int raise_lds_limit(const void* fn, const char* name, size_t dyn_max) {
    hipFuncAttributes fa;
    memset(&fa, 0, sizeof(fa));
    MSMZ_HIP(hipFuncGetAttributes(&fa, fn));
    hipDeviceProp_t prop;
    MSMZ_HIP(hipGetDeviceProperties(&prop, device_));
    // per-workgroup LDS of the device: gfx950 reports 160 KiB as the opt-in maximum (64 KiB is the default allowance)
    size_t dev_max = prop.sharedMemPerBlock;
    if (prop.sharedMemPerBlockOptin > dev_max) dev_max = prop.sharedMemPerBlockOptin;
    if (prop.maxSharedMemoryPerMultiProcessor > dev_max) dev_max = prop.maxSharedMemoryPerMultiProcessor;
    if (fa.sharedSizeBytes + dyn_max > dev_max) {
      fprintf(stderr, "msmz: %s needs %zu B static + %zu B dynamic LDS, the device offers %zu B per workgroup\n", name,
              (size_t)fa.sharedSizeBytes, dyn_max, dev_max);
      return MSMZ_ERR_HIP;
    }
    MSMZ_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn_max));
    return MSMZ_OK;
  }

  ~Engine() override {
    (void)hipSetDevice(device_);
    for (auto& kv : handles_) (void)hipFree(kv.second.dev);
    for (DevBuf* b : {&bsum_, &f2desc_, &tilecnt_, &tileoff_, &final_, &desc_, &bfin_, &packed_, &bins_, &digits_, &counts_, &off_, &cursor_, &refs_, &rscan_, &partials_, &slots_, &red_[0], &red_[1],
                      &red_[2], &red_[3], &meta_, &stage_, &gen_table_})
      b->release();
    if (h_meta_) (void)hipHostFree(h_meta_);
    if (h_final_) (void)hipHostFree(h_final_);
    for (auto& e : ev_)
      if (e) (void)hipEventDestroy(e);
    if (stream_) (void)hipStreamDestroy(stream_);
  }

  // ------------------------------------------------------------------------------------------ data
  // Host -> device copy of this engine's `n` local records of `rec` bytes.  split == nullptr: one contiguous copy.
  // Otherwise the engine is shard `split->shard` of `split->nshards` inside a multi-device context (multi.h): its local
  // block b is global block b * nshards + shard of the caller's buffer, so every block is copied straight from where
  // the caller has it -- no gathered host copy in between.
  int copy_h2d(void* dst, const uint8_t* src, size_t rec, uint64_t n, const GenMap* split) {
    if (!split || split->nshards <= 1) {
      MSMZ_HIP(hipMemcpyAsync(dst, src, n * rec, hipMemcpyHostToDevice, stream_));
      return MSMZ_OK;
    }
    const uint64_t blk = 1ull << split->blk_shift;
    for (uint64_t li = 0; li < n; li += blk) {
      const uint64_t len = n - li < blk ? n - li : blk;
      const uint64_t gi = ((li >> split->blk_shift) * split->nshards + split->shard) << split->blk_shift;
      MSMZ_HIP(hipMemcpyAsync((uint8_t*)dst + li * rec, src + gi * rec, len * rec, hipMemcpyHostToDevice, stream_));
    }
    return MSMZ_OK;
  }

  int upload_points(const uint8_t* xy, const uint8_t* inf, uint64_t n, uint64_t* h, const GenMap* split = nullptr) override {
    if (!xy || !h || n == 0 || n >= (1ull << (Cfg::HAS_ENDO ? 29 : 30))) return MSMZ_ERR_ARG;   // record indices (incl. endomorphism images) fit 30 bits
    MSMZ_HIP(hipSetDevice(device_));
    int st = stage_.ensure(n * RW * 4 + n);
    if (st) return st;
    if ((st = copy_h2d(stage_.p, xy, (size_t)RW * 4, n, split))) return st;
    uint8_t* d_inf = nullptr;
    if (inf) {
      d_inf = stage_.as<uint8_t>() + n * RW * 4;
      if ((st = copy_h2d(d_inf, inf, 1, n, split))) return st;
    }
    const bool endo = Cfg::HAS_ENDO;
    void* dev = nullptr;
    MSMZ_HIP(hipMalloc(&dev, (size_t)n * PW_WORDS * 4 * (endo ? 2 : 1)));
    MsmMeta* d_meta = meta_.as<MsmMeta>();
    MSMZ_HIP(hipMemsetAsync(&d_meta->error, 0, 4, stream_));
    if constexpr (TE) {
      hipLaunchKernelGGL((k_te_points_to_niels<F>), dim3((n + 255) / 256), dim3(256), 0, stream_, (uint32_t*)dev,
                         stage_.as<uint32_t>(), (uint32_t)n, &d_meta->error);
    } else {
      hipLaunchKernelGGL((k_points_to_mont<F>), dim3((n + 255) / 256), dim3(256), 0, stream_, (uint32_t*)dev,
                         stage_.as<uint32_t>(), d_inf, (uint32_t)n, endo ? 1 : 0, &d_meta->error);
    }
    MSMZ_HIP(hipGetLastError());
    MSMZ_HIP(hipMemcpyAsync(&h_meta_->error, &d_meta->error, 4, hipMemcpyDeviceToHost, stream_));
    MSMZ_HIP(hipStreamSynchronize(stream_));
    if (h_meta_->error) {   // a coordinate >= p
      (void)hipFree(dev);
      return MSMZ_ERR_RANGE;
    }
    *h = next_handle_++;
    handles_[*h] = Handle{0, n, endo, dev};
    return MSMZ_OK;
  }

  int upload_scalars(const uint8_t* s, uint64_t n, uint64_t* h, const GenMap* split = nullptr) override {
    if (!s || !h || n == 0) return MSMZ_ERR_ARG;
    MSMZ_HIP(hipSetDevice(device_));
    void* dev = nullptr;
    MSMZ_HIP(hipMalloc(&dev, n * 32));
    if (int st = copy_h2d(dev, s, 32, n, split)) {
      (void)hipFree(dev);
      return st;
    }
    MsmMeta* d_meta = meta_.as<MsmMeta>();
    MSMZ_HIP(hipMemsetAsync(&d_meta->error, 0, 4, stream_));
    hipLaunchKernelGGL((k_check_scalars<Fr>), dim3((n + 255) / 256), dim3(256), 0, stream_, &d_meta->error,
                       (const uint32_t*)dev, (uint32_t)n);
    MSMZ_HIP(hipGetLastError());
    MSMZ_HIP(hipMemcpyAsync(&h_meta_->error, &d_meta->error, 4, hipMemcpyDeviceToHost, stream_));
    MSMZ_HIP(hipStreamSynchronize(stream_));
    if (h_meta_->error) {   // a scalar >= group order
      (void)hipFree(dev);
      return MSMZ_ERR_RANGE;
    }
    *h = next_handle_++;
    handles_[*h] = Handle{1, n, false, dev};
    return MSMZ_OK;
  }

  int random_points(uint64_t n, uint64_t seed, const GenMap& map, uint64_t* h) override {
    if (!h || n == 0 || n >= (1ull << (Cfg::HAS_ENDO ? 29 : 30))) return MSMZ_ERR_ARG;   // record indices (incl. endomorphism images) fit 30 bits
    MSMZ_HIP(hipSetDevice(device_));
    int st = ensure_gen_table();
    if (st) return st;
    const bool endo = Cfg::HAS_ENDO;
    void* dev = nullptr;
    MSMZ_HIP(hipMalloc(&dev, (size_t)n * PW_WORDS * 4 * (endo ? 2 : 1)));
    if constexpr (TE) {
      hipLaunchKernelGGL((k_te_gen_points<F>), dim3((n + 127) / 128), dim3(128), 0, stream_, (uint32_t*)dev,
                         gen_table_.as<uint32_t>(), (uint32_t)n, seed, map);
    } else {
      hipLaunchKernelGGL((k_gen_points<F>), dim3((n + 127) / 128), dim3(128), 0, stream_, (uint32_t*)dev,
                         gen_table_.as<uint32_t>(), (uint32_t)n, seed, endo ? 1 : 0, map);
    }
    MSMZ_HIP(hipGetLastError());
    MSMZ_HIP(hipStreamSynchronize(stream_));
    *h = next_handle_++;
    handles_[*h] = Handle{0, n, endo, dev};
    return MSMZ_OK;
  }

  int random_scalars(uint64_t n, uint64_t seed, const GenMap& map, uint64_t* h) override {
    if (!h || n == 0) return MSMZ_ERR_ARG;
    MSMZ_HIP(hipSetDevice(device_));
    void* dev = nullptr;
    MSMZ_HIP(hipMalloc(&dev, n * 32));
    hipLaunchKernelGGL((k_gen_scalars<Fr>), dim3((n + 255) / 256), dim3(256), 0, stream_, (uint32_t*)dev, (uint32_t)n,
                       seed, map);
    MSMZ_HIP(hipGetLastError());
    MSMZ_HIP(hipStreamSynchronize(stream_));
    *h = next_handle_++;
    handles_[*h] = Handle{1, n, false, dev};
    return MSMZ_OK;
  }

  int download_points(uint64_t hd, uint64_t first, uint64_t count, uint8_t* xy, uint8_t* inf) override {
    auto it = handles_.find(hd);
    if (it == handles_.end() || it->second.kind != 0 || !xy) return MSMZ_ERR_ARG;
    {
      const uint64_t have = it->second.n * (it->second.has_endo ? 2 : 1);   // the endomorphism images stay readable
      if (first > have || count > have - first) return MSMZ_ERR_ARG;        // (no first + count: it can wrap)
    }
    if (count == 0) return MSMZ_OK;
    MSMZ_HIP(hipSetDevice(device_));
    int st = stage_.ensure(count * RW * 4);
    if (st) return st;
    if constexpr (TE) {
      hipLaunchKernelGGL((k_te_points_from_niels<F>), dim3((count + 255) / 256), dim3(256), 0, stream_,
                         stage_.as<uint32_t>(), (const uint32_t*)it->second.dev + first * PW_WORDS, (uint32_t)count);
    } else {
      hipLaunchKernelGGL((k_points_from_mont<F>), dim3((count + 255) / 256), dim3(256), 0, stream_,
                         stage_.as<uint32_t>(), (const uint32_t*)it->second.dev + first * PW_WORDS, (uint32_t)count);
    }
    MSMZ_HIP(hipGetLastError());
    MSMZ_HIP(hipMemcpyAsync(xy, stage_.p, count * RW * 4, hipMemcpyDeviceToHost, stream_));
    MSMZ_HIP(hipStreamSynchronize(stream_));
    if (inf) {
      for (uint64_t i = 0; i < count; i++) {
        bool z = !TE;   // twisted Edwards has no point at infinity: the identity is the affine point (0, 1)
        for (int j = 0; j < RW * 4; j++) z = z && xy[i * RW * 4 + j] == 0;
        inf[i] = z ? 1 : 0;
      }
    }
    return MSMZ_OK;
  }

  int download_scalars(uint64_t hd, uint64_t first, uint64_t count, uint8_t* s) override {
    auto it = handles_.find(hd);
    if (it == handles_.end() || it->second.kind != 1 || !s) return MSMZ_ERR_ARG;
    if (first > it->second.n || count > it->second.n - first) return MSMZ_ERR_ARG;
    if (count == 0) return MSMZ_OK;
    MSMZ_HIP(hipSetDevice(device_));
    MSMZ_HIP(hipMemcpy(s, (const uint8_t*)it->second.dev + first * 32, count * 32, hipMemcpyDeviceToHost));
    return MSMZ_OK;
  }

  int free_handle(uint64_t hd) override {
    auto it = handles_.find(hd);
    if (it == handles_.end()) return MSMZ_ERR_ARG;
    (void)hipSetDevice(device_);
    (void)hipFree(it->second.dev);
    handles_.erase(it);
    return MSMZ_OK;
  }

  // ------------------------------------------------------------------------------------------ msm
  // Largest number of (half-)scalars one pass sorts: index + negate + fine bucket bits share a 32-bit word.
  static constexpr uint64_t kMaxEntriesPerPass = 1ull << 24;

  int msm(uint64_t ph, const uint8_t* host_scalars, uint64_t sh, uint64_t n, const msmz_opts* o, uint8_t* out,
          int* out_inf, msmz_log* log, const GenMap* split = nullptr) override {
    auto t_begin = std::chrono::steady_clock::now();
    if (!out || !out_inf || n == 0) return MSMZ_ERR_ARG;
    auto pit = handles_.find(ph);
    if (pit == handles_.end() || pit->second.kind != 0 || pit->second.n < n) return MSMZ_ERR_ARG;
    msmz_opts opt;
    memset(&opt, 0, sizeof(opt));
    if (o) opt = *o;
    // glv < 0: the engine's choice.  The split halves the windows but doubles the point set (index bits, gathers, tree
    // depth); since the two-dimensional bucket reduction made the reduction cheap per window it is only ahead on the
    // smallest inputs (profiles/r03_sweep.json: 2^14 0.85 vs 0.88 ms, 2^16 1.11 vs 1.07, 2^20 3.87 vs 3.64, 2^23 23.5 vs 20.5).
    if (opt.glv < 0) opt.glv = (!TE && Fr::HAS_GLV && pit->second.has_endo && n < (1ull << 15)) ? 1 : 0;
    MSMZ_HIP(hipSetDevice(device_));

    const uint32_t* d_scalars = nullptr;
    if (host_scalars) {
      // range (< group order) is checked on the device while the scalars are sliced
      int st = stage_.ensure(n * 32);
      if (st) return st;
      if ((st = copy_h2d(stage_.p, host_scalars, 32, n, split))) return st;
      d_scalars = stage_.as<uint32_t>();
    } else {
      auto sit = handles_.find(sh);
      if (sit == handles_.end() || sit->second.kind != 1 || sit->second.n < n) return MSMZ_ERR_ARG;
      d_scalars = (const uint32_t*)sit->second.dev;
    }
    if (log) memset(log, 0, sizeof(*log));
    // Inputs beyond what one pass sorts (2^24 entries; 2^23 points with GLV) run as consecutive index ranges whose
    // partial sums are added on the host -- the same additivity the multi-GPU split uses.
    const uint64_t per_pass = (opt.glv != 0 && !TE) ? kMaxEntriesPerPass / 2 : kMaxEntriesPerPass;
    const Handle& pts = pit->second;
    int st = MSMZ_OK;
    uint8_t part[RW * 4];
    for (uint64_t done = 0; done < n && st == MSMZ_OK; done += per_pass) {
      const uint64_t cnt = n - done < per_pass ? n - done : per_pass;
      const uint32_t* d_points = (const uint32_t*)pts.dev + done * PW_WORDS;
      const uint32_t* d_sc = d_scalars + done * 8;
      int pinf = 0;
      msmz_log plog;
      msmz_log* lp = log ? &plog : nullptr;
      if (lp) memset(lp, 0, sizeof(*lp));
      st = Cfg::run_msm(*this, pts, d_points, d_sc, cnt, opt, done == 0 ? out : part, done == 0 ? out_inf : &pinf, lp, 0);
      if (st == MSMZ_ERR_RETRY_BITS) {
        // a GLV half longer than the assumed 127 bits (k_hist flags it): redo with windows for the PROVEN bound
        // (Fr::GLV_PROVEN_BITS, tools/gen_constants.py), which no half can exceed -- a second flag is an internal error
        retries_++;
        st = Cfg::run_msm(*this, pts, d_points, d_sc, cnt, opt, done == 0 ? out : part, done == 0 ? out_inf : &pinf, lp, 1);
        if (st == MSMZ_ERR_RETRY_BITS) st = MSMZ_ERR_ARG;
      }
      if (st) break;
      if (done > 0) {
        uint8_t acc[RW * 4];
        memcpy(acc, out, sizeof(acc));
        const int ai = *out_inf;
        st = host_point_add<F, TE>(acc, ai, part, pinf, out, out_inf);
      }
      if (log) merge_log(log, plog, done == 0);
    }
    if (log) {
      log->stage_ms[MSMZ_ST_TOTAL] =
          std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    }
    return st;
  }

  static void merge_log(msmz_log* total, const msmz_log& part, bool first) {
    if (first) {
      *total = part;
      return;
    }
    for (int i = 0; i < MSMZ_N_STAGES; i++) total->stage_ms[i] += part.stage_ms[i];
    for (int i = 0; i < 32; i++) total->batch_add_ms[i] += part.batch_add_ms[i];
    total->n_entries += part.n_entries;
    total->n_pairs += part.n_pairs;
    total->scatter_kernel_ms += part.scatter_kernel_ms;
    total->scatter_launches += part.scatter_launches;
    if (part.max_bucket > total->max_bucket) total->max_bucket = part.max_bucket;
    if (part.rounds > total->rounds) total->rounds = part.rounds;
  }

  // ------------------------------------------------------------------------------------------ stage-level test hooks
  // (include/msmz_test.h: one device routine at a time, raw outputs)
  int test_buffers(size_t in_bytes, size_t out_bytes, uint8_t** d_in, uint8_t** d_out) {
    int st = stage_.ensure(in_bytes + out_bytes + 256);
    if (st) return st;
    *d_in = stage_.as<uint8_t>();
    *d_out = stage_.as<uint8_t>() + ((in_bytes + 255) & ~(size_t)255);
    return MSMZ_OK;
  }

  int test_set_glv_bits(int bits) override {
    if (!Fr::HAS_GLV) return MSMZ_ERR_UNSUPPORTED;
    if (bits != 0 && (bits < 8 || bits > Fr::GLV_BITS - 1)) return MSMZ_ERR_ARG;
    glv_bits_assumed_ = bits;
    return MSMZ_OK;
  }
  int test_retries() override { return retries_; }

  int test_field(int op, const uint8_t* a, const uint8_t* b, uint64_t n, uint8_t* out) override {
    if (!a || !b || !out || n == 0 || n > (1u << 22)) return MSMZ_ERR_ARG;
    MSMZ_HIP(hipSetDevice(device_));
    const size_t eb = (size_t)FE_BYTES * n;
    uint8_t *d_in, *d_out;
    int st = test_buffers(2 * eb, eb, &d_in, &d_out);
    if (st) return st;
    if ((st = slots_.ensure(((size_t)n + 64) * SlotFmt<F>::WORDS * 4))) return st;
    MSMZ_HIP(hipMemcpyAsync(d_in, a, eb, hipMemcpyHostToDevice, stream_));
    MSMZ_HIP(hipMemcpyAsync(d_in + eb, b, eb, hipMemcpyHostToDevice, stream_));
    hipLaunchKernelGGL((k_test_field<F>), dim3((n + 63) / 64), dim3(64), 0, stream_, (uint32_t*)d_out,
                       (const uint32_t*)d_in, (const uint32_t*)(d_in + eb), (uint32_t)n, op, slots_.as<uint32_t>());
    MSMZ_HIP(hipGetLastError());
    MSMZ_HIP(hipMemcpyAsync(out, d_out, eb, hipMemcpyDeviceToHost, stream_));
    MSMZ_HIP(hipStreamSynchronize(stream_));
    return MSMZ_OK;
  }

  int test_glv(const uint8_t* s, uint64_t n, uint8_t* s0, uint8_t* s1, uint8_t* neg) override {
    if (!Fr::HAS_GLV) return MSMZ_ERR_UNSUPPORTED;
    if (!s || !s0 || !s1 || !neg || n == 0 || n > (1u << 22)) return MSMZ_ERR_ARG;
    MSMZ_HIP(hipSetDevice(device_));
    uint8_t *d_in, *d_out;
    int st = test_buffers(32 * n, 34 * n, &d_in, &d_out);
    if (st) return st;
    MSMZ_HIP(hipMemcpyAsync(d_in, s, 32 * n, hipMemcpyHostToDevice, stream_));
    hipLaunchKernelGGL((k_test_glv<Fr>), dim3((n + 255) / 256), dim3(256), 0, stream_, (uint32_t*)d_out,
                       (uint32_t*)(d_out + 16 * n), d_out + 32 * n, (const uint32_t*)d_in, (uint32_t)n);
    MSMZ_HIP(hipGetLastError());
    MSMZ_HIP(hipMemcpyAsync(s0, d_out, 16 * n, hipMemcpyDeviceToHost, stream_));
    MSMZ_HIP(hipMemcpyAsync(s1, d_out + 16 * n, 16 * n, hipMemcpyDeviceToHost, stream_));
    MSMZ_HIP(hipMemcpyAsync(neg, d_out + 32 * n, 2 * n, hipMemcpyDeviceToHost, stream_));
    MSMZ_HIP(hipStreamSynchronize(stream_));
    return MSMZ_OK;
  }

  int test_digits(const uint8_t* s, uint64_t n, int c, int K, int glv, uint32_t* digits) override {
    if (!s || !digits || n == 0 || n > (1u << 22) || c < 2 || c > 24 || K < 1 || K > kMaxWindows) return MSMZ_ERR_ARG;
    if (glv && !Fr::HAS_GLV) return MSMZ_ERR_UNSUPPORTED;
    MSMZ_HIP(hipSetDevice(device_));
    const size_t ob = (size_t)(glv ? 2 : 1) * n * K * 4;
    uint8_t *d_in, *d_out;
    int st = test_buffers(32 * n, ob, &d_in, &d_out);
    if (st) return st;
    MSMZ_HIP(hipMemcpyAsync(d_in, s, 32 * n, hipMemcpyHostToDevice, stream_));
    if (glv) {
      if constexpr (Fr::HAS_GLV)
        hipLaunchKernelGGL((k_test_digits<Fr, true>), dim3((n + 255) / 256), dim3(256), 0, stream_, (uint32_t*)d_out,
                           (const uint32_t*)d_in, (uint32_t)n, c, K);
    } else {
      hipLaunchKernelGGL((k_test_digits<Fr, false>), dim3((n + 255) / 256), dim3(256), 0, stream_, (uint32_t*)d_out,
                         (const uint32_t*)d_in, (uint32_t)n, c, K);
    }
    MSMZ_HIP(hipGetLastError());
    MSMZ_HIP(hipMemcpyAsync(digits, d_out, ob, hipMemcpyDeviceToHost, stream_));
    MSMZ_HIP(hipStreamSynchronize(stream_));
    return MSMZ_OK;
  }

  int test_sort(const uint8_t* s, uint64_t n, int c, int glv, int force_fallback, uint32_t* geom, uint32_t* off,
                uint64_t off_cap, uint32_t* refs, uint64_t refs_cap) override {
    if (!s || !geom || n == 0 || n > (1u << 22)) return MSMZ_ERR_ARG;
    if (glv && !Fr::HAS_GLV) return MSMZ_ERR_UNSUPPORTED;
    MSMZ_HIP(hipSetDevice(device_));
    msmz_opts opt;
    memset(&opt, 0, sizeof(opt));
    opt.c = c;
    Plan pl;
    int st = make_plan(pl, n, glv != 0, opt, (uint32_t)n, !TE);
    if (st) return st;
    void* d_scalars = nullptr;
    MSMZ_HIP(hipMalloc(&d_scalars, 32 * n));
    MSMZ_HIP(hipMemcpyAsync(d_scalars, s, 32 * n, hipMemcpyHostToDevice, stream_));
    const bool saved = force_atomic_sort_;
    force_atomic_sort_ = saved || force_fallback != 0;
    st = sort_phase(pl, (const uint32_t*)d_scalars);
    force_atomic_sort_ = saved;
    if (!st) st = fetch_meta(pl);
    if (!st && (h_meta_->error & 4u)) st = MSMZ_ERR_RANGE;
    if (!st) {
      const uint32_t g8[8] = {(uint32_t)pl.c, (uint32_t)pl.K, (uint32_t)pl.Keff, pl.L, pl.nb, pl.n_entries, pl.max_bucket,
                              (uint32_t)pl.spread};
      memcpy(geom, g8, sizeof(g8));
      if (off) {
        if (off_cap < (uint64_t)pl.nb + 1) st = MSMZ_ERR_ARG;
        else if (hipMemcpy(off, off_.p, ((size_t)pl.nb + 1) * 4, hipMemcpyDeviceToHost) != hipSuccess) st = MSMZ_ERR_HIP;
      }
      if (!st && refs) {
        if (refs_cap < pl.n_entries) st = MSMZ_ERR_ARG;
        else if (pl.n_entries && hipMemcpy(refs, refs_.p, (size_t)pl.n_entries * 4, hipMemcpyDeviceToHost) != hipSuccess)
          st = MSMZ_ERR_HIP;
      }
    }
    (void)hipFree(d_scalars);
    return st;
  }

  int test_point(int op, const uint8_t* a, const uint8_t* a_inf, const uint8_t* b, const uint8_t* b_inf, uint64_t n,
                 uint8_t* out) override {
    if (!a || !b || !out || n == 0 || n > (1u << 20)) return MSMZ_ERR_ARG;
    MSMZ_HIP(hipSetDevice(device_));
    const size_t pb = (size_t)2 * FE_BYTES * n;
    uint8_t *d_in, *d_out;
    int st = test_buffers(2 * pb + 2 * n, pb, &d_in, &d_out);
    if (st) return st;
    MSMZ_HIP(hipMemcpyAsync(d_in, a, pb, hipMemcpyHostToDevice, stream_));
    MSMZ_HIP(hipMemcpyAsync(d_in + pb, b, pb, hipMemcpyHostToDevice, stream_));
    uint8_t* d_ai = nullptr;
    uint8_t* d_bi = nullptr;
    if (a_inf) {
      d_ai = d_in + 2 * pb;
      MSMZ_HIP(hipMemcpyAsync(d_ai, a_inf, n, hipMemcpyHostToDevice, stream_));
    }
    if (b_inf) {
      d_bi = d_in + 2 * pb + n;
      MSMZ_HIP(hipMemcpyAsync(d_bi, b_inf, n, hipMemcpyHostToDevice, stream_));
    }
    using P = typename std::conditional<TE, TePolicy<F>, WeierPolicy<F>>::type;
    const uint64_t threads = (op == TP_ADD_X4 || op == TP_DBL_X4) ? 4 * n : n;
    hipLaunchKernelGGL((k_test_point<P, TE>), dim3((threads + 63) / 64), dim3(64), 0, stream_, (uint32_t*)d_out,
                       (const uint32_t*)d_in, (const uint32_t*)(d_in + pb), d_ai, d_bi, (uint32_t)n, op);
    MSMZ_HIP(hipGetLastError());
    MSMZ_HIP(hipMemcpyAsync(out, d_out, pb, hipMemcpyDeviceToHost, stream_));
    MSMZ_HIP(hipStreamSynchronize(stream_));
    return MSMZ_OK;
  }

  // ------------------------------------------------------------------------------------------ shared phases
  struct Plan {
    uint32_t n, M, L, nb, nblocks;
    int c, K, b;
    int Keff, spread;           // bucket windows incl. the top window's 2^spread sub-windows
    int fold_shift = 0, fold_rows = 0;   // ... or the top window folded into its own bucket set (SortGeom)
    uint32_t top_range = 1;              // values the top window's digit can take
    bool glv, timing;
    uint32_t max_bucket = 0, n_entries = 0;
    uint32_t endo_delta = 0;    // GLV over a prefix of a set: half-1 entry i reads point record pts_n + i = (n + i) + endo_delta
    int ei = 0;                 // next event slot
    int ev_coarse = -1, ev_sort_end = -1;
  };

  void mark(Plan& pl) {
    if (pl.timing && pl.ei < kMaxEvents) (void)hipEventRecord(ev_[pl.ei], stream_);
    pl.ei++;
  }
  float elapsed(int a, int b2) {
    float ms = 0;
    if (a >= 0 && b2 >= 0 && a < kMaxEvents && b2 < kMaxEvents) (void)hipEventElapsedTime(&ms, ev_[a], ev_[b2]);
    return ms;
  }

  // Window geometry for window size c: K windows, L buckets each, significant bits t_top of the top window's
  // digit (from the largest scalar q - 1, or the typical GLV half), and the 2^spread sub-windows the top window
  // is spread over when it is sparse.
  struct Geometry {
    int c, K, t_top, spread, Keff;
    uint32_t L;
    uint32_t top_range = 0;   // number of values the top window's digit can take (<= L + 1)
    int fold_shift = 0, fold_rows = 0;   // thin top window folded into its own bucket set (sort_kernels.h SortGeom)
  };
  Geometry geometry(int c, bool glv, uint32_t M, int b, bool allow_fold = false) const {
    Geometry g;
    g.c = c;
    g.K = (b + 1 + c - 1) / c;                              // msm-batched-affine.ts:96
    g.L = 1u << (c - 1);
    const int pos = (g.K - 1) * c;
    g.t_top = b + 1 - pos;
    if (!glv) {
      uint64_t top = 0;
      for (int j = 0; j < 64 && pos + j < 256; j++)
        top |= (uint64_t)((Fr::Q[(pos + j) >> 5] >> ((pos + j) & 31)) & 1u) << j;
      top += 1;   // carry from the window below
      g.t_top = ceil_log2_u64(top + 1);
      g.top_range = (uint32_t)(top + 1 > g.L ? g.L : top + 1);
    } else if (Fr::GLV_TYP_BITS + 1 - pos < g.t_top) {
      g.t_top = Fr::GLV_TYP_BITS + 1 - pos;
      if (g.t_top < 1) g.t_top = 1;
    }
    g.spread = 0;
    {
      // a thin top window whose digit fits the COLUMN index of the two-dimensional reduction (l < D = 2^b2) is folded:
      // 2^(c-1-b2) copies of the digit's range fill the set's buckets as evenly as any other window's.  The bound on
      // the digit is the hard one (largest scalar; for GLV halves the bit length the windows were sized for).
      const int b2 = (c - 1) - (c - 1 + 1) / 2;                       // low bits of Split2d
      const int t_bound = glv ? b + 1 - pos : g.t_top;
      // (only where the two-level sort applies: the fallback sort numbers buckets by digit alone)
      const uint32_t ncb0 = g.L >> fine_bits(c, M);
      const bool sort2 = !force_atomic_sort_ && M <= (1u << 24) && ncb0 <= (uint32_t)COARSE_MAX_BINS &&
                         (uint64_t)g.K * ncb0 <= (uint64_t)SORT_MAX_BINS;
      if (allow_fold && !no_fold_ && sort2 && g.K > 1 && g.t_top <= c - 2 && b2 >= 1 && t_bound <= b2) {
        g.fold_shift = b2;
        g.fold_rows = c - 1 - b2;
      }
    }
    if (g.fold_shift == 0 && !no_spread_ && g.K > 1 && g.t_top <= c - 2) {
      g.spread = c - 1 - g.t_top;
      if (g.spread > 3) g.spread = 3;
      const int ib = ceil_log2_u64(M < 2 ? 2 : M);
      (void)ib;
      const int fbx = fine_bits(c, M);
      while (g.spread > 0 && ((g.L >> fbx) << g.spread) > (uint32_t)COARSE_MAX_BINS) g.spread--;
    }
    if (g.top_range == 0) g.top_range = g.t_top >= c - 1 ? g.L : 1u << g.t_top;
    g.Keff = g.K - 1 + (1 << g.spread);
    return g;
  }

  // Fine bits of the two-level sort = log2(buckets per coarse bin): as many as (1) the packed word leaves beside the
  // index and the sign, (2) k_fine's counters hold, and (3) keep an average bin inside k_fine's LDS staging (a bin of
  // 2^fb buckets holds ~M 2^fb / L entries; beyond FINE_STAGE it falls back to scattered stores: 3x slower).
  int fine_bits(int c, uint32_t M) const {
    const int idx_bits = ceil_log2_u64(M < 2 ? 2 : M);
    int fb = 31 - idx_bits;
    if (fb > FINE_MAX_BITS) fb = FINE_MAX_BITS;
    if (fb_cap_ > 0 && fb > fb_cap_) fb = fb_cap_;
    if (fb > c - 1) fb = c - 1;
    const uint64_t L = 1ull << (c - 1);
    while (fb > 0 && (((uint64_t)M << fb) / L) * 10 > (uint64_t)FINE_STAGE * 9) fb--;
    return fb;
  }

  // Fine bits of the TOP window's bins (SortGeom::fbt): its entries fall on top_range << spread buckets only (the largest
  // scalar bounds the top digit), so they are up to 2x denser than M / L; as many fine bits as keep such a bin inside
  // k_fine's staging, and no fewer than keep the window's bins inside k_coarse's 9-bit bin field.
  int fine_bits_top(const Plan& pl, int fb) const {
    if (pl.fold_shift != 0 || no_fbt_) return fb;
    const uint64_t slots = (uint64_t)pl.top_range << pl.spread;
    int fbt = fb;
    while (fbt > 0 && (((uint64_t)pl.M << fbt) / slots) * 10 > (uint64_t)FINE_STAGE * 9) fbt--;
    while (fbt < fb && ((pl.L >> fbt) << pl.spread) > (uint32_t)COARSE_MAX_BINS) fbt++;
    return fbt;
  }

  // Default window size.  Large inputs (M >= 2^18: profiles/r03_sweep.json) are throughput-bound: c = log2 M - 3 capped at 17, stepped
  // down while the top window would be nearly empty.  Smaller inputs are latency-bound -- every tree round costs
  // ~75 us whatever its size and the number of rounds is log2 of the LONGEST bucket, which usually sits in a
  // partly filled top window -- so they pick the c that minimizes a small cost model fitted to this GPU
  // (ms: rounds * 0.075 + additions / 4.5e6 + reduction levels * 0.065 + buckets * 0.8e-6).
  int choose_window(bool glv, uint32_t M, int b, bool tree_rounds) const {
    int c = default_window(M);
    if (M >= (1u << 18) || no_window_model_) {
      // measured optimum of the batched-affine path from 2^18 entries per window on (profiles/r03_sweep.json): 17 without
      // GLV (2^18: 1.60 ms against 1.83 at c = 15), 16 with it (128-bit halves = 8 windows exactly)
      if (tree_rounds && !no_window_model_) c = glv ? 16 : 17;
      for (int tries = 0; tries < 3 && c > 4; tries++) {
        const int K0 = (b + 1 + c - 1) / c;
        const int top_bits = b + 1 - (K0 - 1) * c;
        if (K0 == 1 || top_bits >= c - 4) break;
        c--;
      }
      return c;
    }
    const int lg = ceil_log2_u64(M < 2 ? 2 : M);
    int best_c = c;
    double best = 1e30;
    for (int cc = (lg - 6 < 3 ? 3 : lg - 6); cc <= (lg + 2 > 17 ? 17 : lg + 2); cc++) {
      const Geometry g = geometry(cc, glv, M, b);
      if (g.Keff > kMaxWindows) continue;
      const double lam = (double)M / g.L;
      const double conc = g.t_top < cc ? (double)(1u << (cc - g.t_top)) / (1 << g.spread) : 1.0;
      double maxb = 1.5 * lam + 12;
      if (g.K > 1 && conc * lam * 1.3 + 12 > maxb) maxb = conc * lam * 1.3 + 12;
      if (maxb > M) maxb = M;
      const int rounds = ceil_log2_u64((uint64_t)(maxb < 2 ? 2 : maxb));
      const double cost = (tree_rounds ? 0.075 * rounds : 0.0) + (double)g.K * M / 4.5e6 + 0.065 * ((cc - 1 + 1) / 2) +
                          0.8e-6 * g.Keff * g.L;
      if (cost < best) {
        best = cost;
        best_c = cc;
      }
    }
    return best_c;
  }

  int make_plan(Plan& pl, uint64_t n64, bool glv, const msmz_opts& opt, uint32_t pts_n, bool tree_rounds = true,
                int extra_bits = 0, bool allow_fold = false) {
    pl.n = (uint32_t)n64;
    pl.glv = glv;
    pl.M = glv ? 2 * pl.n : pl.n;
    // scalar bit length.  GLV halves: first attempt assumes |s_j| < 2^127 (every half seen so far; for BLS12-377 the
    // analytic bound is 2^126); k_hist flags a longer half and the MSM is redone (extra_bits = 1) with the proven bound
    // GLV_PROVEN_BITS <= 128, which also is what the 4-word halves of glv_decompose can hold.
    static_assert(!Fr::HAS_GLV || (Fr::GLV_PROVEN_BITS <= 128 && Fr::GLV_PROVEN_BITS <= Fr::GLV_BITS), "GLV halves must fit 4 words");
    if (!glv) {
      pl.b = Fr::BITS;
    } else if (extra_bits) {
      pl.b = Fr::GLV_PROVEN_BITS > Fr::GLV_BITS - 1 ? Fr::GLV_PROVEN_BITS : Fr::GLV_BITS - 1;
    } else {
      pl.b = glv_bits_assumed_ > 0 ? glv_bits_assumed_ : Fr::GLV_BITS - 1;
    }
    pl.c = opt.c > 0 ? opt.c : choose_window(glv, pl.M, pl.b, tree_rounds);
    if (pl.c < 2) pl.c = 2;
    if (pl.c > 24) pl.c = 24;
    const Geometry g = geometry(pl.c, glv, pl.M, pl.b, allow_fold);
    pl.K = g.K;
    pl.L = g.L;
    pl.spread = g.spread;
    pl.top_range = g.top_range;
    pl.fold_shift = g.fold_shift;
    pl.fold_rows = g.fold_rows;
    pl.Keff = g.Keff;
    const uint64_t nb64 = (uint64_t)pl.Keff * pl.L;
    if (nb64 + 1 >= (1ull << 31) || (uint64_t)pl.K * pl.M >= (1ull << 32) || pl.Keff > kMaxWindows) return MSMZ_ERR_ARG;
    pl.nb = (uint32_t)nb64;
    pl.nblocks = (pl.nb + SCAN_TILE - 1) / SCAN_TILE;
    pl.timing = opt.timing != 0;
    pl.endo_delta = glv ? pts_n - pl.n : 0u;
    return MSMZ_OK;
  }

  // scalars -> sorted references `refs_` + bucket offsets `off_` (+ meta->max_bucket); events 0..4.  No host round trip.
  int sort_phase(Plan& pl, const uint32_t* d_scalars) {
    const uint32_t n = pl.n, M = pl.M, L = pl.L, nb = pl.nb, nblocks = pl.nblocks;
    const int c = pl.c, K = pl.K;
    int st;
    if ((st = refs_.ensure((size_t)K * M * 4))) return st;
    if ((st = off_.ensure(((size_t)nb + 1) * 4))) return st;
    MsmMeta* d_meta = meta_.as<MsmMeta>();
    MSMZ_HIP(hipMemsetAsync(d_meta, 0, sizeof(MsmMeta), stream_));
    // two-level LDS-staged sort when the packed (fine | negate | index) word fits; else per-entry atomics.
    // packed word = fine bucket bits | negate | index: the narrower the index, the more fine bits fit, the
    // fewer (and longer) coarse runs the scatter writes
    const int idx_bits = ceil_log2_u64(M < 2 ? 2 : M);
    const int fb = fine_bits(c, M);
    const uint32_t ncb = L >> fb;
    const int fbt = fine_bits_top(pl, fb);
    const uint32_t ncbt = L >> fbt;
    const uint32_t top_bin = (uint32_t)(K - 1) * ncb;
    const uint32_t nbins = top_bin + (ncbt << pl.spread);
    const bool sort2 = !force_atomic_sort_ && fb >= 0 && M <= (1u << 24) && ncb <= (uint32_t)COARSE_MAX_BINS &&
                       (ncbt << pl.spread) <= (uint32_t)COARSE_MAX_BINS && nbins <= (uint32_t)SORT_MAX_BINS;
    const uint32_t n_half = pl.glv ? n : 0xffffffffu;
    if (sort2) {
      if ((st = packed_.ensure((size_t)K * M * 4))) return st;
      if ((st = bins_.ensure(((size_t)nbins + 2) * 4 + kTraceBytes * nbins))) return st;
      if ((st = counts_.ensure((size_t)nbins * 4))) return st;
      uint32_t* d_counts = counts_.as<uint32_t>();
      MSMZ_HIP(hipMemsetAsync(d_counts, 0, (size_t)nbins * 4, stream_));
      SortGeom g{n, M, c, K, fb, pl.spread, idx_bits, ncb, fbt, ncbt, pl.fold_shift, pl.fold_rows};
      mark(pl);  // 0
      const uint32_t per_tile = pl.glv ? COARSE_TILE / 2 : COARSE_TILE;   // scalars per workgroup (k_hist and k_coarse)
      const uint32_t tiles = (n + per_tile - 1) / per_tile;
      if ((st = tilecnt_.ensure((size_t)tiles * nbins * 2))) return st;
      if ((st = tileoff_.ensure((size_t)tiles * nbins * 4 + kTraceBytes * tiles))) return st;   // the tiles' runs inside the bins
      // kernels specialized for the window size (unrolled window loop) where one is compiled: 16 / 17, the defaults of
      // large inputs; any other window size takes the generic ones
      const int cspec = (no_sort_special_ || (c != 16 && c != 17) || (pl.glv && c != 16)) ? 0 : c;
      auto launch_sort = [&](auto glvc, auto cc, bool coarse) {
        constexpr bool G = decltype(glvc)::value;
        constexpr int C = decltype(cc)::value;
        if (!coarse)
          hipLaunchKernelGGL((k_hist<Fr, G, C>), dim3(tiles), dim3(COARSE_T), (size_t)nbins * 4, stream_, d_counts,
                             tilecnt_.as<uint16_t>(), tileoff_.as<uint32_t>(), d_meta, d_scalars, g, nbins);
        else   // dynamic LDS <= kCoarseLdsMax (nbins <= SORT_MAX_BINS): the limit init() raised
          hipLaunchKernelGGL((k_coarse<Fr, G, C>), dim3(tiles), dim3(COARSE_T), (size_t)2 * nbins * 4, stream_,
                             packed_.as<uint32_t>(), tileoff_.as<uint32_t>(), bins_.as<uint32_t>(), tilecnt_.as<uint16_t>(),
                             d_scalars, g, nbins);
      };
      auto dispatch_sort = [&](bool coarse) {
        using std::integral_constant;
        if (pl.glv) {
          if constexpr (Fr::HAS_GLV) {
            if (cspec == 16) launch_sort(std::true_type{}, integral_constant<int, 16>{}, coarse);
            else launch_sort(std::true_type{}, integral_constant<int, 0>{}, coarse);
          }
        } else if (cspec == 17) {
          launch_sort(std::false_type{}, integral_constant<int, 17>{}, coarse);
        } else if (cspec == 16) {
          launch_sort(std::false_type{}, integral_constant<int, 16>{}, coarse);
        } else {
          launch_sort(std::false_type{}, integral_constant<int, 0>{}, coarse);
        }
      };
      dispatch_sort(false);
      mark(pl);  // 1
      MSMZ_HIP(hipGetLastError());
      hipLaunchKernelGGL(k_bin_scan, dim3(1), dim3(1024), 0, stream_, bins_.as<uint32_t>(), d_counts, nbins,
                         &d_meta->n_entries);
      mark(pl);  // 2
      MSMZ_HIP(hipGetLastError());
      dispatch_sort(true);
      pl.ev_coarse = pl.ei;
      mark(pl);  // 3
      MSMZ_HIP(hipGetLastError());
      {
        const size_t lds = kFineLds;
        hipLaunchKernelGGL(k_fine, dim3(nbins), dim3(FINE_T), lds, stream_, refs_.as<uint32_t>(), off_.as<uint32_t>(),
                           &d_meta->max_bucket, packed_.as<uint32_t>(), bins_.as<uint32_t>(), fb, fbt, top_bin, nbins, idx_bits,
                           n_half, pl.endo_delta);
      }
#ifdef MSMZ_TRACE
      // development aid: workgroup time stamps of k_coarse / k_fine (tools/wg_timeline.py)
      if ((st = trace_dump("k_coarse", tileoff_.as<uint32_t>() + (size_t)tiles * nbins, tiles, true))) return st;
      if ((st = trace_dump("k_fine", bins_.as<uint32_t>() + ((nbins + 2) & ~1u), nbins, false))) return st;
#endif
    } else {
      // fallback (window sizes whose coarse bins do not fit the LDS staging): digits materialized, one global
      // atomic per entry
      if (pl.fold_shift != 0) return MSMZ_ERR_ARG;   // (make_plan only folds when the two-level sort applies)
      if ((st = digits_.ensure((size_t)K * M * 4))) return st;
      if ((st = counts_.ensure(((size_t)nb + 1) * 4))) return st;
      if ((st = cursor_.ensure((size_t)nb * 4))) return st;
      if ((st = partials_.ensure((size_t)32 * nblocks * 4))) return st;
      MSMZ_HIP(hipMemsetAsync(counts_.p, 0, ((size_t)nb + 1) * 4, stream_));
      MSMZ_HIP(hipMemsetAsync(cursor_.p, 0, (size_t)nb * 4, stream_));
      const uint32_t dgrid = (n + 256 * DIGITS_ITEMS - 1) / (256 * DIGITS_ITEMS);
      mark(pl);  // 0
      if (pl.glv) {
        if constexpr (Fr::HAS_GLV)
          hipLaunchKernelGGL((k_digits<Fr, true>), dim3(dgrid), dim3(256), 0, stream_, digits_.as<uint32_t>(),
                             counts_.as<uint32_t>(), d_meta, d_scalars, n, c, K, pl.spread);
      } else {
        hipLaunchKernelGGL((k_digits<Fr, false>), dim3(dgrid), dim3(256), 0, stream_, digits_.as<uint32_t>(),
                           counts_.as<uint32_t>(), d_meta, d_scalars, n, c, K, pl.spread);
      }
      mark(pl);  // 1
      MSMZ_HIP(hipGetLastError());
      hipLaunchKernelGGL(k_scan_partials, dim3(nblocks, 1), dim3(SCAN_T), 0, stream_, partials_.as<uint32_t>(),
                         counts_.as<uint32_t>(), nb, 0, nblocks);
      hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(SCAN_T), 0, stream_, partials_.as<uint32_t>(), nblocks,
                         &d_meta->n_entries);
      hipLaunchKernelGGL(k_scan_apply, dim3(nblocks, 1), dim3(SCAN_T), 0, stream_, off_.as<uint32_t>(),
                         partials_.as<uint32_t>(), counts_.as<uint32_t>(), nb, 0, nblocks, (size_t)0,
                         &d_meta->max_bucket);
      mark(pl);  // 2
      MSMZ_HIP(hipGetLastError());
      {
        dim3 grid((M + 256 * 4 - 1) / (256 * 4), K);
        hipLaunchKernelGGL(k_scatter, grid, dim3(256), 0, stream_, refs_.as<uint32_t>(), cursor_.as<uint32_t>(),
                           off_.as<uint32_t>(), digits_.as<uint32_t>(), M, c, pl.spread, n_half, pl.endo_delta);
      }
      pl.ev_coarse = pl.ei;
      mark(pl);  // 3
      MSMZ_HIP(hipGetLastError());
    }
    pl.ev_sort_end = pl.ei;
    mark(pl);  // 4
    MSMZ_HIP(hipGetLastError());
    return MSMZ_OK;
  }

  // read the device-side totals (one host round trip)
  int fetch_meta(Plan& pl) {
    MSMZ_HIP(hipMemcpyAsync(h_meta_, meta_.p, sizeof(MsmMeta), hipMemcpyDeviceToHost, stream_));
    MSMZ_HIP(hipStreamSynchronize(stream_));
    pl.max_bucket = h_meta_->max_bucket;
    pl.n_entries = h_meta_->n_entries;
    return MSMZ_OK;
  }

  // reduce levels on accumulator records: rows in red_[cur*2], C in red_[cur*2+1]; ends with one entry per window
  template <class P>
  int reduce_levels(const Plan& pl, int& cur, uint32_t n_in, uint32_t nprob = 0) {
    constexpr int AW = P::ACC_WORDS;
    int st;
    if (nprob == 0) nprob = (uint32_t)pl.Keff;   // independent weighted sums ("windows") the levels run side by side
    while (n_in > tail_n_) {
      const uint32_t S = 4;   // quads handle short tails too
      uint32_t g2 = (n_in + S - 1) / S;
      int nxt = cur ^ 1;
      if ((st = red_[nxt * 2].ensure((size_t)nprob * g2 * AW * 4))) return st;
      if ((st = red_[nxt * 2 + 1].ensure((size_t)nprob * g2 * AW * 4))) return st;
      uint32_t total = nprob * g2;
      if (total <= quad16_max_groups_) {
        // small level: latency-bound, one DPP quad per addition
        hipLaunchKernelGGL((k_reduce_quad16<P>), dim3((total * 16 + 63) / 64), dim3(64), 0, stream_,
                           red_[nxt * 2].as<uint32_t>(), red_[nxt * 2 + 1].as<uint32_t>(),
                           red_[cur * 2].as<uint32_t>(), red_[cur * 2 + 1].as<uint32_t>(), n_in, g2, total);
      } else {
        hipLaunchKernelGGL((k_reduce_quad<P>), dim3((total * 4 + 63) / 64), dim3(64), 0, stream_,
                           red_[nxt * 2].as<uint32_t>(), red_[nxt * 2 + 1].as<uint32_t>(),
                           red_[cur * 2].as<uint32_t>(), red_[cur * 2 + 1].as<uint32_t>(), n_in, g2, total);
      }
      n_in = g2;
      cur = nxt;
    }
    // the last levels (<= REDUCE_TAIL_N entries per window) in ONE launch, one workgroup per window; leaves the
    // window sums in final_
    {
      const int nxt = cur ^ 1;
      if ((st = red_[nxt * 2].ensure((size_t)nprob * n_in * AW * 4))) return st;
      if ((st = red_[nxt * 2 + 1].ensure((size_t)nprob * n_in * AW * 4))) return st;
      if ((st = final_.ensure((size_t)nprob * AW * 4))) return st;
      hipLaunchKernelGGL((k_reduce_tail<P>), dim3(nprob), dim3(REDUCE_TAIL_T), 0, stream_, red_[cur * 2].as<uint32_t>(),
                         red_[cur * 2 + 1].as<uint32_t>(), red_[nxt * 2].as<uint32_t>(), red_[nxt * 2 + 1].as<uint32_t>(),
                         final_.as<uint32_t>(), n_in, n_in);
    }
    return MSMZ_OK;
  }

  uint32_t first_group_size(const Plan& pl) const {
    if (s1_override_ > 0) return s1_override_ < pl.L ? s1_override_ : pl.L;
    uint32_t S1 = 2;
    // L / S1 a power of 4 saves one reduction level; with >= 2^20 buckets groups of 8 still fill the GPU
    // (2 waves per SIMD) and halve the levels above (measured: S1 = 4 -> 1.58 ms, 8 -> 1.45 ms, 16 -> 1.94 ms)
    if (pl.L >= 4 && (ceil_log2_u64(pl.L) & 1) == 0) S1 = 4;
    if ((uint64_t)pl.Keff * pl.L >= (1u << 20) && pl.L >= 8) S1 = 8;
    return S1 < pl.L ? S1 : pl.L;
  }

  // copy the K window sums (the C entries of the last level; its rows are multiples of L and not needed) to the host
  template <class P>
  int fetch_window_sums(const Plan& pl, int cur, uint32_t nprob = 0) {
    constexpr int AW = P::ACC_WORDS;
    MSMZ_HIP(hipGetLastError());
    (void)cur;
    if (nprob == 0) nprob = (uint32_t)pl.Keff;
    MSMZ_HIP(hipMemcpyAsync(h_final_ + (size_t)kMaxWindows * AW, final_.p, (size_t)nprob * AW * 4,
                            hipMemcpyDeviceToHost, stream_));
    MSMZ_HIP(hipMemcpyAsync(h_meta_, meta_.p, sizeof(MsmMeta), hipMemcpyDeviceToHost, stream_));
    MSMZ_HIP(hipStreamSynchronize(stream_));
    return MSMZ_OK;
  }

  // final sum on the host (msm-batched-affine.ts:300-322): W_k = C_k of the last level, Horner over windows
  void finalize_weierstrass(const Plan& pl, uint8_t* out, int* out_inf) {
    // ~K*c dependent doublings: on 64-bit limbs (host64.h), ~4x faster on a CPU core than the kernels' limb code
    using H = Host64<F>;
    typename H::Pt acc, w, t;
    host64_.set_inf(acc);
    for (int k = pl.Keff - 1; k >= 0; k--) {
      if (k < pl.K - 1)   // windows K-1 .. Keff-1 are the sub-windows of the top window: same weight
        for (int j = 0; j < pl.c; j++) {
          host64_.dbl(t, acc);
          acc = t;
        }
      host64_.load_pt(w, h_final_ + (size_t)(kMaxWindows + k) * XW);   // W_k = C of the last level
      host64_.add_pt(t, acc, w);
      acc = t;
    }
    Xyzz<F> fin;
    host64_.to_xyzz(fin, acc);
    uint32_t res[RW];
    bool inf = xyzz_to_affine_canon<F>(res, fin);
    memcpy(out, res, RW * 4);
    *out_inf = inf ? 1 : 0;
  }

  // Two-dimensional bucket reduction (reduce2d_kernels.h): line sums, then the weighted sums over H lines of 2 Keff
  // problems with the upper-level kernels.  Leaves result 2 kw (rows) / 2 kw + 1 (columns) of bucket set kw in final_.
  struct Split2d {
    int a, b;            // c - 1 = a + b: high / low bits of the bucket weight
    uint32_t H, D, NC;
  };
  Split2d split_2d(const Plan& pl) const {
    Split2d s;
    s.a = (pl.c - 1 + 1) / 2;
    s.b = pl.c - 1 - s.a;
    s.H = 1u << s.a;
    s.D = 1u << s.b;
    // chunks per line: so that the partial sums of all lines are ~256 K threads (measured at 2^20: 16 / 32 / 64 chunks ->
    // reduce stage 0.88 / 0.81 / 0.81 ms), at most 32 per line (5 pair-sum launches), and a chunk holds at least one
    // bucket along either direction
    uint32_t nc = 1;
    while (nc < 32 && nc * 2 <= s.D && (uint64_t)2 * pl.Keff * s.H * nc < (1u << 18)) nc *= 2;
    if (r2_nc_ > 0) {
      nc = 1;
      while (nc < r2_nc_ && nc * 2 <= s.D) nc *= 2;
    }
    s.NC = nc;
    return s;
  }
  // basic = true: the buckets are sums of partial accumulators (msmBasic path: slots_ + rscan_), else the affine bucket
  // sums of the tree rounds (bfin_)
  template <class P>
  int reduce_2d(const Plan& pl, const uint32_t* d_points, bool basic = false, bool summed = false) {
    const Split2d sp = split_2d(pl);
    R2Geom g;
    g.L = pl.L;
    g.H = sp.H;
    g.D = sp.D;
    g.NC = sp.NC;
    g.chr = sp.D / sp.NC;
    g.chc = sp.H / sp.NC;
    g.nprob = 2u * (uint32_t)pl.Keff;
    const uint32_t lines = g.nprob * g.H;
    const uint32_t total = lines * g.NC;
    int st;
    // ping-pong between red_[0] and red_[2] (rows of the level machinery); C inputs of the first level = infinity
    if ((st = red_[0].ensure((size_t)total * XW * 4))) return st;
    if ((st = red_[2].ensure((size_t)total * XW * 4))) return st;
    if (basic && summed) {   // one accumulator per bucket in bsum_ (k_bucket_sums)
      hipLaunchKernelGGL((k_reduce2d_partial_acc<P>), dim3((total + 127) / 128), dim3(128), 0, stream_,
                         red_[0].as<uint32_t>(), bsum_.as<uint32_t>(), (const uint32_t*)nullptr, g, total);
    } else if (basic) {
      hipLaunchKernelGGL((k_reduce2d_partial_acc<P>), dim3((total + 127) / 128), dim3(128), 0, stream_,
                         red_[0].as<uint32_t>(), slots_.as<uint32_t>(), rscan_.as<uint32_t>(), g, total);
    } else {
      if constexpr (!TE)
        hipLaunchKernelGGL((k_reduce2d_partial<F>), dim3((total + 127) / 128), dim3(128), 0, stream_,
                           red_[0].as<uint32_t>(), slots_.as<uint32_t>(), d_points, bfin_.as<uint4>(), g, total);
    }
    int src = 0;
    for (uint32_t n = total / 2; n >= lines && g.NC > 1; n /= 2) {
      const int dst = src ^ 2;
      if (n <= pairsum_x4_max_) {
        hipLaunchKernelGGL((k_pairsum_x4<P>), dim3((n * 4 + 63) / 64), dim3(64), 0, stream_, red_[dst].as<uint32_t>(),
                           red_[src].as<uint32_t>(), n);
      } else {
        hipLaunchKernelGGL((k_pairsum<P>), dim3((n + 127) / 128), dim3(128), 0, stream_, red_[dst].as<uint32_t>(),
                           red_[src].as<uint32_t>(), n);
      }
      src = dst;
      if (n == lines) break;
    }
    // upper levels: rows = line sums (weight unit 1), C = infinity (all-zero accumulator records)
    const int crow = src, ccol = src + 1;
    if ((st = red_[ccol].ensure((size_t)lines * XW * 4))) return st;
    hipLaunchKernelGGL((k_fill_neutral<P>), dim3((lines + 255) / 256), dim3(256), 0, stream_, red_[ccol].as<uint32_t>(), lines);
    int cur = crow >> 1;   // reduce_levels addresses rows as red_[cur * 2], C as red_[cur * 2 + 1]
    if ((st = reduce_levels<P>(pl, cur, g.H, g.nprob))) return st;
    MSMZ_HIP(hipGetLastError());
    return MSMZ_OK;
  }
  // Horner over the windows with the two results of every bucket set: acc = (acc * 2^(c-b) + A) * 2^b + B
  void finalize_weierstrass_2d(const Plan& pl, uint8_t* out, int* out_inf) {
    using H = Host64<F>;
    typename H::Pt acc, w, t;
    host64_.set_inf(acc);
    const Split2d sp = split_2d(pl);
    auto add_results = [&](int k, int which) {
      // bucket sets of window k: kw = k below the top window, K-1 .. Keff-1 (its sub-windows) for the top one
      const int lo = k, hi = (k == pl.K - 1) ? pl.Keff - 1 : k;
      if (which == 0 && k == pl.K - 1 && pl.fold_shift != 0) return;   // folded top window: its rows are copies, not weights
      for (int kw = lo; kw <= hi; kw++) {
        host64_.load_pt(w, h_final_ + (size_t)(kMaxWindows + 2 * kw + which) * XW);
        host64_.add_pt(t, acc, w);
        acc = t;
      }
    };
    for (int k = pl.K - 1; k >= 0; k--) {
      if (k < pl.K - 1)
        for (int j = 0; j < pl.c - sp.b; j++) {
          host64_.dbl(t, acc);
          acc = t;
        }
      add_results(k, 0);   // rows: weight 2^(c k + b)
      for (int j = 0; j < sp.b; j++) {
        host64_.dbl(t, acc);
        acc = t;
      }
      add_results(k, 1);   // columns: weight 2^(c k)
    }
    Xyzz<F> fin;
    host64_.to_xyzz(fin, acc);
    uint32_t res[RW];
    bool inf = xyzz_to_affine_canon<F>(res, fin);
    memcpy(out, res, RW * 4);
    *out_inf = inf ? 1 : 0;
  }

  void fill_log(msmz_log* log, const Plan& pl, int R, uint64_t n_pairs, int ev_plan0, int ev_plan1, int ev_acc_end,
                int ev_red_end, int round_ev0, float host_ms) {
    if (!log) return;
    log->c = pl.c;
    log->K = pl.K;
    log->rounds = R;
    log->glv = pl.glv ? 1 : 0;
    log->n_entries = pl.n_entries;
    log->n_pairs = n_pairs;
    log->max_bucket = pl.max_bucket;
    log->stage_ms[MSMZ_ST_FINAL] = host_ms;
    if (!pl.timing) return;
    log->stage_ms[MSMZ_ST_DIGITS] = elapsed(0, 1);
    log->stage_ms[MSMZ_ST_SCAN] = elapsed(1, 2);
    log->stage_ms[MSMZ_ST_SCATTER] = elapsed(2, pl.ev_sort_end);
    log->scatter_kernel_ms = elapsed(2, pl.ev_coarse);
    log->scatter_launches = 1;
    log->stage_ms[MSMZ_ST_PLAN] = elapsed(ev_plan0, ev_plan1);
    log->stage_ms[MSMZ_ST_ACCUMULATE] = elapsed(ev_plan1, ev_acc_end);
    log->stage_ms[MSMZ_ST_REDUCE] = elapsed(ev_acc_end, ev_red_end);
    int prev = ev_plan1, rr = 0;
    for (int r = 0; r < R && r < 32; r++) {
      if (h_round_pairs_[r] == 0) continue;
      int e = round_ev0 + rr;
      if (e >= kMaxEvents) break;
      log->batch_add_ms[r] = elapsed(prev, e);
      prev = e;
      rr++;
    }
  }

  // ------------------------------------------------------------------------------------------ Weierstrass, affine buckets
  int msm_weierstrass_affine(const Handle& pts, const uint32_t* d_points, const uint32_t* d_scalars, uint64_t n64,
                             const msmz_opts& opt, uint8_t* out, int* out_inf, msmz_log* log, int extra_bits = 0) {
    const bool glv = opt.glv != 0;
    if (glv && (!Fr::HAS_GLV || !pts.has_endo)) return MSMZ_ERR_UNSUPPORTED;
    Plan pl;
    const bool want_2d = opt.reserved[0] != 1 && reduce2d_;
    int st = make_plan(pl, n64, glv, opt, (uint32_t)pts.n, true, extra_bits, want_2d);
    if (st) return st;
    // location words hold a record index in 30 bits
    if ((uint64_t)pl.K * pl.M >= (1ull << 30)) return MSMZ_ERR_ARG;
    // whole groups of 64 records; + the records of the batched-affine first reduction level when it is selected
    const size_t f2_records = opt.reserved[0] == 1 ? (size_t)13 * pl.Keff * ((pl.L + 1) / 2) + 256 : 0;
    if ((st = slots_.ensure(((size_t)pl.K * pl.M + 64 + f2_records) * SlotFmt<F>::WORDS * 4))) return st;
    if ((st = sort_phase(pl, d_scalars))) return st;
    const uint32_t nb = pl.nb;
    MsmMeta* d_meta = meta_.as<MsmMeta>();

    // ---- plan: descriptors of every pair of every round + what is left of each bucket (plan_kernels.h)
    const int ev_plan0 = pl.ei;
    mark(pl);
    // buckets per plan workgroup: at most PLAN_CHUNK, fewer when the windows have few (long) buckets, so that the plan
    // still spreads over ~4 workgroups per CU
    uint32_t chunk = PLAN_CHUNK;
    while (chunk > 64 && (nb + chunk - 1) / chunk < 1024) chunk >>= 1;
    // the top window's bucket sets in half-size chunks when they are denser than the others (plan_kernels.h PlanChunks)
    PlanChunks pc;
    pc.chunk = chunk;
    pc.nb_main = nb;
    pc.chunk_top = chunk;
    if (!no_plan_top_ && pl.K > 1 && pl.fold_shift == 0 && chunk >= 128 &&
        (uint64_t)pl.L * 10 > ((uint64_t)pl.top_range << pl.spread) * 13) {
      pc.nb_main = (uint32_t)(pl.K - 1) * pl.L;
      pc.chunk_top = chunk / 2;
    }
    pc.n_main = (pc.nb_main + chunk - 1) / chunk;
    const uint32_t n_chunks = pc.n_main + (nb - pc.nb_main + pc.chunk_top - 1) / pc.chunk_top;
    // chunk totals per round, then the per-workgroup scratch of the rounds beyond PLAN_RL
    const size_t pair_words = (size_t)n_chunks * (PLAN_RMAX - PLAN_RL) * PLAN_T;
    if ((st = rscan_.ensure(((size_t)PLAN_RMAX * n_chunks + pair_words) * 4 + kTraceBytes * n_chunks))) return st;
    if ((st = desc_.ensure((size_t)pl.K * pl.M * 8))) return st;
    if ((st = bfin_.ensure((size_t)nb * 16))) return st;
    // the batched-affine first reduction level (opt.reserved[0] = 1) wants ONE sum per bucket: no rounds skipped
    const bool f2 = opt.reserved[0] == 1 && pl.L >= 2;
    const bool r2d = !f2 && reduce2d_ && pl.L >= 2;
    const int tail_skip = f2 ? 0 : (r2d ? tail_skip_2d_ : tail_skip_);
    hipLaunchKernelGGL(k_plan_count, dim3(n_chunks), dim3(PLAN_T), 0, stream_, rscan_.as<uint32_t>(), off_.as<uint32_t>(),
                       nb, n_chunks, d_meta, tail_skip, pc);
    hipLaunchKernelGGL(k_plan_emit, dim3(n_chunks), dim3(PLAN_T), 0, stream_, desc_.as<uint2>(), bfin_.as<uint4>(),
                       d_meta, rscan_.as<uint32_t>(), off_.as<uint32_t>(), refs_.as<uint32_t>(), nb, n_chunks,
                       tail_skip, rscan_.as<uint32_t>() + (size_t)PLAN_RMAX * n_chunks, pc);
    MSMZ_HIP(hipGetLastError());
#ifdef MSMZ_TRACE
    if ((st = trace_dump("k_plan_emit", rscan_.as<uint32_t>() + (size_t)PLAN_RMAX * n_chunks + pair_words, n_chunks, false))) return st;
#endif
    if ((st = fetch_meta(pl))) return st;      // the ONE host round trip before the final fetch
    if (h_meta_->error & 4u) return MSMZ_ERR_RANGE;
    if (h_meta_->error & 2u) return MSMZ_ERR_RETRY_BITS;
    if (pl.max_bucket > (1u << 24)) return MSMZ_ERR_ARG;
    const int R = (int)h_meta_->rounds;
    memcpy(h_round_pairs_, h_meta_->round_pairs, sizeof(h_round_pairs_));
    const int ev_plan1 = pl.ei;
    mark(pl);
    uint64_t n_pairs = 0;
    const int round_ev0 = pl.ei;
    for (int r = 0; r < R; r++) n_pairs += h_round_pairs_[r];
    for (int r = 0; r < R; r++) {
      const uint32_t pairs = h_round_pairs_[r];
      if (pairs == 0) continue;
      launch_batch_add(pairs, opt.safe != 0, d_points, desc_.as<uint2>() + h_meta_->round_base[r], h_meta_->round_base[r],
                       d_meta);
#ifdef MSMZ_TRACE
      {
        char nm[32];
        snprintf(nm, sizeof nm, "k_batch_add round %d", r);
        int B = 1;
        while (B < 16 && (uint64_t)pairs >= (uint64_t)MSMZ_BATCH_T * (B * 2) * batch_min_wgs_) B *= 2;
        const uint32_t wgs = (pairs + MSMZ_BATCH_T * B - 1) / (MSMZ_BATCH_T * B);
        if ((st = trace_dump(nm, d_meta + 1, wgs < 65536 ? wgs : 65536, false))) return st;
      }
#endif
      mark(pl);
    }
    const int ev_acc_end = pl.ei;
    mark(pl);

    // ---- bucket reduction
    using P = WeierPolicy<F>;
    if (r2d) {
      // two-dimensional: row / column sums of the buckets, then two half-length weighted sums per bucket set
      if ((st = reduce_2d<P>(pl, d_points))) return st;
      const int ev_red_end2 = pl.ei;
      mark(pl);
      if ((st = fetch_window_sums<P>(pl, 0, 2u * (uint32_t)pl.Keff))) return st;
      auto t_host2 = std::chrono::steady_clock::now();
      if (h_meta_->error & 1u) return MSMZ_ERR_DEGENERATE;
      finalize_weierstrass_2d(pl, out, out_inf);
      float host_ms2 = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_host2).count();
      fill_log(log, pl, R, n_pairs, ev_plan0, ev_plan1, ev_acc_end, ev_red_end2, round_ev0, host_ms2);
      return MSMZ_OK;
    }
    // level 1 from affine bucket sums, then XYZZ levels down to one entry per window
    uint32_t S1 = first_group_size(pl);
    if (f2) {   // the weight-L bucket is folded into element L/2, which must be the FIRST element of its group
      if (S1 > 8) S1 = 8;
      while (S1 > 1 && S1 * 2 > pl.L) S1 >>= 1;
    }
    const uint32_t groups = (pl.L + S1 - 1) / S1;   // elements are weights 0..L-1 (weight L folded into L/2)
    if ((st = red_[0].ensure((size_t)pl.Keff * groups * XW * 4))) return st;
    if ((st = red_[1].ensure((size_t)pl.Keff * groups * XW * 4))) return st;
    if (f2) {
      if ((st = reduce_first_affine(pl, d_points, S1, groups, n_pairs, d_meta))) return st;
    } else {
      uint32_t total = pl.Keff * groups;
      hipLaunchKernelGGL((k_reduce_first<F>), dim3((total + 127) / 128), dim3(128), 0, stream_,
                         red_[0].as<uint32_t>(), red_[1].as<uint32_t>(), slots_.as<uint32_t>(), d_points,
                         bfin_.as<uint4>(), pl.L, S1, groups, total);
    }
    int cur = 0;
    if ((st = reduce_levels<P>(pl, cur, groups))) return st;
    const int ev_red_end = pl.ei;
    mark(pl);
    if ((st = fetch_window_sums<P>(pl, cur))) return st;
    auto t_host0 = std::chrono::steady_clock::now();
    if (h_meta_->error & 1u) return MSMZ_ERR_DEGENERATE;
    finalize_weierstrass(pl, out, out_inf);
    float host_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_host0).count();
    fill_log(log, pl, R, n_pairs, ev_plan0, ev_plan1, ev_acc_end, ev_red_end, round_ev0, host_ms);
    return MSMZ_OK;
  }

  // ------------------------------------------------------------------------------------------ msmBasic: projective / extended buckets
  // (msm-basic.ts:45-176; Weierstrass "projective fallback" parallel.ts:69-87 and the twisted-Edwards MSM)
  template <class P>
  int msm_basic(const Handle& pts, const uint32_t* d_points, const uint32_t* d_scalars, uint64_t n64, const msmz_opts& opt,
                Plan& pl) {
    int st = make_plan(pl, n64, false, opt, (uint32_t)pts.n, false);
    if (st) return st;
    if ((st = sort_phase(pl, d_scalars))) return st;
    if ((st = fetch_meta(pl))) return st;
    if (h_meta_->error & 4u) return MSMZ_ERR_RANGE;
    if ((st = partials_.ensure((size_t)32 * pl.nblocks * 4))) return st;
    constexpr int AW = P::ACC_WORDS;
    const uint32_t nb = pl.nb, nblocks = pl.nblocks;
    MsmMeta* d_meta = meta_.as<MsmMeta>();
    const int ev_plan0 = pl.ei;
    mark(pl);
    // chunk offsets: cscan[g] = sum_{g' < g} ceil(size / 2^chunk_shift); chunks of 64 entries unless some bucket is
    // very long (then ~sqrt of it: bounds both the chunk and the number of partial sums one reduction thread adds)
    int chunk_shift = chunk_shift_override_ > 0 ? chunk_shift_override_ : ACC_CHUNK_SHIFT;
    while ((1ull << (2 * chunk_shift)) < pl.max_bucket) chunk_shift++;
    const int scan_mode = 2 | (chunk_shift << 4);
    if ((st = rscan_.ensure(((size_t)nb + 1) * 4))) return st;
    hipLaunchKernelGGL(k_scan_partials, dim3(nblocks, 1), dim3(SCAN_T), 0, stream_, partials_.as<uint32_t>(),
                       off_.as<uint32_t>(), nb, scan_mode, nblocks);
    hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(SCAN_T), 0, stream_, partials_.as<uint32_t>(), nblocks,
                       d_meta->round_pairs);
    hipLaunchKernelGGL(k_scan_apply, dim3(nblocks, 1), dim3(SCAN_T), 0, stream_, rscan_.as<uint32_t>(),
                       partials_.as<uint32_t>(), off_.as<uint32_t>(), nb, scan_mode, nblocks, (size_t)0, (uint32_t*)nullptr);
    MSMZ_HIP(hipMemcpyAsync(h_meta_, d_meta, sizeof(MsmMeta), hipMemcpyDeviceToHost, stream_));
    MSMZ_HIP(hipStreamSynchronize(stream_));
    const uint32_t n_chunks = h_meta_->round_pairs[0];
    const int ev_plan1 = pl.ei;
    mark(pl);
    if ((st = slots_.ensure((size_t)(n_chunks + 1) * AW * 4))) return st;
    if (n_chunks > 0) {
      hipLaunchKernelGGL((k_bucket_accumulate<P>), dim3((n_chunks + 127) / 128), dim3(128), 0, stream_,
                         slots_.as<uint32_t>(), d_points, refs_.as<uint32_t>(), off_.as<uint32_t>(),
                         rscan_.as<uint32_t>(), nb, n_chunks, chunk_shift);
    }
    const int ev_acc_end = pl.ei;
    mark(pl);
    basic_2d_ = reduce2d_ && pl.L >= 2;
    if (basic_2d_) {
      // every bucket is visited twice: buckets of several chunk accumulators (large inputs: Pallas 2^22 has 4,
      // ed-on-bls12-377 2^24 has 8) are first summed into one accumulator each, in bucket order
      const bool summed = (uint64_t)n_chunks * 2 > (uint64_t)nb * 3 && !no_bucket_sums_;
      if (summed) {
        if ((st = bsum_.ensure((size_t)nb * AW * 4))) return st;
        hipLaunchKernelGGL((k_bucket_sums<P>), dim3((nb + 127) / 128), dim3(128), 0, stream_, bsum_.as<uint32_t>(),
                           slots_.as<uint32_t>(), rscan_.as<uint32_t>(), nb);
      }
      if ((st = reduce_2d<P>(pl, d_points, true, summed))) return st;
      const int ev_red_end2 = pl.ei;
      mark(pl);
      if ((st = fetch_window_sums<P>(pl, 0, 2u * (uint32_t)pl.Keff))) return st;
      basic_ev_[0] = ev_plan0;
      basic_ev_[1] = ev_plan1;
      basic_ev_[2] = ev_acc_end;
      basic_ev_[3] = ev_red_end2;
      return MSMZ_OK;
    }
    const uint32_t S1 = first_group_size(pl);
    const uint32_t groups = (pl.L + S1 - 1) / S1;   // elements are weights 0..L-1 (weight L folded into L/2)
    if ((st = red_[0].ensure((size_t)pl.Keff * groups * AW * 4))) return st;
    if ((st = red_[1].ensure((size_t)pl.Keff * groups * AW * 4))) return st;
    {
      uint32_t total = pl.Keff * groups;
      hipLaunchKernelGGL((k_reduce_next<P>), dim3((total + 127) / 128), dim3(128), 0, stream_, red_[0].as<uint32_t>(),
                         red_[1].as<uint32_t>(), slots_.as<uint32_t>(), (const uint32_t*)nullptr,
                         rscan_.as<uint32_t>(), pl.L, S1, groups, total, pl.L);
    }
    int cur = 0;
    if ((st = reduce_levels<P>(pl, cur, groups))) return st;
    const int ev_red_end = pl.ei;
    mark(pl);
    if ((st = fetch_window_sums<P>(pl, cur))) return st;
    basic_ev_[0] = ev_plan0;
    basic_ev_[1] = ev_plan1;
    basic_ev_[2] = ev_acc_end;
    basic_ev_[3] = ev_red_end;
    return MSMZ_OK;
  }

  int msm_weierstrass_projective(const Handle& pts, const uint32_t* d_points, const uint32_t* d_scalars, uint64_t n64,
                                 const msmz_opts& opt, uint8_t* out, int* out_inf, msmz_log* log) {
    if (opt.glv) return MSMZ_ERR_UNSUPPORTED;   // msmProjective never uses the endomorphism (parallel.ts:69-87)
    Plan pl;
    int st = msm_basic<WeierPolicy<F>>(pts, d_points, d_scalars, n64, opt, pl);
    if (st) return st;
    auto t_host0 = std::chrono::steady_clock::now();
    if (basic_2d_) finalize_weierstrass_2d(pl, out, out_inf); else finalize_weierstrass(pl, out, out_inf);
    float host_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_host0).count();
    memset(h_round_pairs_, 0, sizeof(h_round_pairs_));
    fill_log(log, pl, 0, pl.n_entries, basic_ev_[0], basic_ev_[1], basic_ev_[2], basic_ev_[3], 0, host_ms);
    return MSMZ_OK;
  }

  // twisted Edwards MSM (parallel.ts:179-289 -> msm-basic.ts): extended buckets, no GLV
  int msm_twisted_edwards(const Handle& pts, const uint32_t* d_points, const uint32_t* d_scalars, uint64_t n64,
                          const msmz_opts& opt, uint8_t* out, int* out_inf, msmz_log* log) {
    if (opt.glv) return MSMZ_ERR_UNSUPPORTED;   // the reference's TE path has no endomorphism (msm-basic.ts:4)
    Plan pl;
    int st = msm_basic<TePolicy<F>>(pts, d_points, d_scalars, n64, opt, pl);
    if (st) return st;
    auto t_host0 = std::chrono::steady_clock::now();
    TeExt<F> acc;
    te_set_zero(acc);
    auto dbl_n = [&](int n) {
      for (int j = 0; j < n; j++) {
        TeExt<F> t;
        te_add(t, acc, acc);
        acc = t;
      }
    };
    if (basic_2d_) {
      // acc = (acc * 2^(c-b) + rows) * 2^b + columns, per window (see finalize_weierstrass_2d)
      const Split2d sp = split_2d(pl);
      auto add_results = [&](int k, int which) {
        const int lo = k, hi = (k == pl.K - 1) ? pl.Keff - 1 : k;
        for (int kw = lo; kw <= hi; kw++) {
          TeExt<F> w, t;
          host_load_te(w, h_final_ + (size_t)(kMaxWindows + 2 * kw + which) * XW);
          te_add(t, acc, w);
          acc = t;
        }
      };
      for (int k = pl.K - 1; k >= 0; k--) {
        if (k < pl.K - 1) dbl_n(pl.c - sp.b);
        add_results(k, 0);
        dbl_n(sp.b);
        add_results(k, 1);
      }
    } else {
      for (int k = pl.Keff - 1; k >= 0; k--) {
        if (k < pl.K - 1) dbl_n(pl.c);
        TeExt<F> w, t;
        host_load_te(w, h_final_ + (size_t)(kMaxWindows + k) * XW);
        te_add(t, acc, w);
        acc = t;
      }
    }
    uint32_t res[RW];
    te_to_affine_canon<F>(res, acc);
    memcpy(out, res, RW * 4);
    *out_inf = 0;
    float host_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_host0).count();
    memset(h_round_pairs_, 0, sizeof(h_round_pairs_));
    fill_log(log, pl, 0, pl.n_entries, basic_ev_[0], basic_ev_[1], basic_ev_[2], basic_ev_[3], 0, host_ms);
    return MSMZ_OK;
  }

  static void host_load_te(TeExt<F>& p, const uint32_t* w) {
    fe_unpack<F>(p.X, w);
    fe_unpack<F>(p.Y, w + NW);
    fe_unpack<F>(p.Z, w + 2 * NW);
    fe_unpack<F>(p.T, w + 3 * NW);
  }

  // Batched-affine first level of the bucket reduction (reduce_affine.h): S - 1 chain steps and
  // a short pair tree, every launch over Keff * groups (x pairs per group) additions; results behind the tree rounds'
  // records.  Leaves the scaled (row, tri) XYZZ records in red_[0] / red_[1] like k_reduce_first.
  int reduce_first_affine(const Plan& pl, const uint32_t* d_points, uint32_t S, uint32_t groups, uint64_t tree_pairs,
                          MsmMeta* d_meta) {
    F2Geom g;
    memset(&g, 0, sizeof(g));
    g.L = pl.L;
    g.S = S;
    g.groups = groups;
    g.NG = (uint32_t)pl.Keff * groups;
    g.inf_slot = (uint32_t)((tree_pairs + 63) / 64 * 64);
    g.out0 = g.inf_slot + 64;
    uint32_t n_launch = 0, dsum = 0, osum = 0;
    for (uint32_t step = 1; step < S; step++) {
      g.ppg[n_launch] = 1;
      g.desc_off[n_launch] = dsum;
      g.out_off[n_launch] = osum;
      dsum += g.NG;
      osum += g.NG;
      n_launch++;
    }
    for (uint32_t n = S - 1; n > 1; n = n / 2 + (n & 1)) {
      g.ppg[n_launch] = n / 2;
      g.desc_off[n_launch] = dsum;
      g.out_off[n_launch] = osum;
      dsum += g.NG * (n / 2);
      osum += g.NG * (n / 2);
      n_launch++;
    }
    g.n_launches = n_launch;
    g.desc_off[n_launch] = dsum;   // (row, tri) locations of every group
    dsum += g.NG;
    if ((uint64_t)g.out0 + osum + 64 >= (1ull << 30)) return MSMZ_ERR_ARG;
    int st;
    if ((st = f2desc_.ensure((size_t)dsum * 8))) return st;
    // slots_ may have to grow: its contents (the tree rounds' results) must survive -> it was sized for this in advance
    if (((size_t)g.out0 + osum + 64) * SlotFmt<F>::WORDS * 4 > slots_.bytes) return MSMZ_ERR_HIP;
    MSMZ_HIP(hipMemsetAsync(slots_.as<uint32_t>() + slot_words(g.inf_slot), 0, (size_t)64 * SlotFmt<F>::WORDS * 4, stream_));
    hipLaunchKernelGGL(k_reduce_affine_desc, dim3((g.NG + 255) / 256), dim3(256), 0, stream_, f2desc_.as<uint2>(),
                       bfin_.as<uint4>(), g);
    for (uint32_t i = 0; i < n_launch; i++)
      launch_batch_add(g.NG * g.ppg[i], true, d_points, f2desc_.as<uint2>() + g.desc_off[i], g.out0 + g.out_off[i], d_meta);
    hipLaunchKernelGGL((k_reduce_affine_finish<F>), dim3((g.NG + 127) / 128), dim3(128), 0, stream_, red_[0].as<uint32_t>(),
                       red_[1].as<uint32_t>(), slots_.as<uint32_t>(), d_points, f2desc_.as<uint2>() + g.desc_off[n_launch],
                       bfin_.as<uint4>(), g);
    MSMZ_HIP(hipGetLastError());
    return MSMZ_OK;
  }
  static size_t slot_words(uint32_t rec) {   // host twin of slot_offset<F> (rec a multiple of 64)
    return (size_t)(rec >> 6) * (SlotFmt<F>::CH * 64) * 4;
  }

  // one launch of batched-affine additions: pairs `dsc[0 .. pairs)`, results in slot records out_base + t
  void launch_batch_add(uint32_t pairs, bool safe, const uint32_t* d_points, const uint2* dsc, uint32_t out_base,
                        MsmMeta* d_meta) {
    constexpr int T = MSMZ_BATCH_T, OCC = MSMZ_BATCH_OCC, BMAX = MSMZ_BATCH_BMAX;
    // pairs per thread: as many as keep >= ~2 workgroups per CU in flight, capped at BMAX = 16 (measured per round at
    // 2^20: 7.6 M pairs B = 8..16, 3.7 M: 16, 1.8 M: 8, 0.9 M: 4, < 0.3 M: 2; 32 is slower everywhere)
    int B = 1;
    while (B < BMAX && (uint64_t)pairs >= (uint64_t)T * (B * 2) * batch_min_wgs_) B *= 2;
    if (batch_b_override_ > 0) B = batch_b_override_ < BMAX ? batch_b_override_ : BMAX;
    dim3 grid((pairs + T * B - 1) / (T * B)), block(T);
    if constexpr (!TE) {
      if (safe) {
        hipLaunchKernelGGL((k_batch_add<F, T, true, OCC, BMAX>), grid, block, 0, stream_, slots_.as<uint32_t>(),
                           d_points, dsc, out_base, pairs, B, d_meta);
      } else {
        hipLaunchKernelGGL((k_batch_add<F, T, false, OCC, BMAX>), grid, block, 0, stream_, slots_.as<uint32_t>(),
                           d_points, dsc, out_base, pairs, B, d_meta);
      }
    }
  }

  static void host_load_xyzz(Xyzz<F>& p, const uint32_t* w) {
    fe_unpack<F>(p.X, w);
    fe_unpack<F>(p.Y, w + NW);
    fe_unpack<F>(p.ZZ, w + 2 * NW);
    fe_unpack<F>(p.ZZZ, w + 3 * NW);
  }

  int ensure_gen_table() {
    if (gen_table_.p) return MSMZ_OK;
    int st = gen_table_.ensure((size_t)GEN_WINDOWS * GEN_TABLE * RW * 4);
    if (st) return st;
    // bases 2^(13 k) * G computed on the host, multiples on the device
    uint32_t bases[GEN_WINDOWS * RW];
    Affine<F> ga;
    fe_set_const<F>(ga.x, F::GX);
    fe_set_const<F>(ga.y, F::GY);
    if constexpr (TE) {
      TeExt<F> g;
      g.X = ga.x;
      g.Y = ga.y;
      fe_set_const<F>(g.Z, F::ONE);
      fe_mul(g.T, ga.x, ga.y);
      for (int k = 0; k < GEN_WINDOWS; k++) {
        Fe<F> zi, x, y;
        fe_inverse(zi, g.Z);
        fe_mul(x, g.X, zi);
        fe_mul(y, g.Y, zi);
        fe_store<F>(bases + k * RW, x);
        fe_store<F>(bases + k * RW + NW, y);
        for (int j = 0; j < GEN_BITS; j++) {
          TeExt<F> t;
          te_add(t, g, g);
          g = t;
        }
      }
    } else {
      Xyzz<F> g;
      xyzz_from_affine(g, ga);
      for (int k = 0; k < GEN_WINDOWS; k++) {
        Affine<F> a;
        host_xyzz_to_affine_mont(a, g);
        fe_store<F>(bases + k * RW, a.x);
        fe_store<F>(bases + k * RW + NW, a.y);
        for (int j = 0; j < GEN_BITS; j++) {
          Xyzz<F> t;
          xyzz_dbl(t, g);
          g = t;
        }
      }
    }
    st = stage_.ensure(sizeof(bases));
    if (st) return st;
    MSMZ_HIP(hipMemcpyAsync(stage_.p, bases, sizeof(bases), hipMemcpyHostToDevice, stream_));
    if constexpr (TE) {
      hipLaunchKernelGGL((k_te_gen_table<F>), dim3((GEN_WINDOWS * GEN_TABLE + 127) / 128), dim3(128), 0, stream_,
                         gen_table_.as<uint32_t>(), stage_.as<uint32_t>());
    } else {
      hipLaunchKernelGGL((k_gen_table<F>), dim3((GEN_WINDOWS * GEN_TABLE + 127) / 128), dim3(128), 0, stream_,
                         gen_table_.as<uint32_t>(), stage_.as<uint32_t>());
    }
    MSMZ_HIP(hipGetLastError());
    MSMZ_HIP(hipStreamSynchronize(stream_));
    return MSMZ_OK;
  }

  static void host_xyzz_to_affine_mont(Affine<F>& a, const Xyzz<F>& p) {
    Fe<F> zi3, t, zi2;
    fe_inverse(zi3, p.ZZZ);
    fe_mul(t, zi3, p.ZZ);
    fe_sqr(zi2, t);
    fe_mul(a.x, p.X, zi2);
    fe_mul(a.y, p.Y, zi3);
  }

  // shared state -------------------------------------------------------------------------------
  static constexpr int kMaxEvents = 64;
  static constexpr int kMaxWindows = 128;
  int device_;
  hipStream_t stream_ = nullptr;
  hipEvent_t ev_[kMaxEvents] = {};
  std::map<uint64_t, Handle> handles_;
  uint64_t next_handle_ = 1;
  // Tuning knobs.  A release build uses the constants; a development build (-DMSMZ_DEV, tools/build_variant.sh)
  // reads MSMZ_* environment variables when the context is created.  None of them changes a result.
#ifdef MSMZ_DEV
  static int env_int(const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; }
#else
  static int env_int(const char*, int dflt) { return dflt; }
#endif
  uint32_t coarse_wgs_ = (uint32_t)env_int("MSMZ_COARSE_WGS", 2048);
  uint32_t batch_min_wgs_ = (uint32_t)env_int("MSMZ_BATCH_WGS", 512);
  // rounds left to the reduction's loader: at most 2 (a bucket's final-location record holds 4 partial sums)
  int tail_skip_ = env_int("MSMZ_TAIL_SKIP", 2) > 2 ? 2 : env_int("MSMZ_TAIL_SKIP", 2);
  int chunk_shift_override_ = env_int("MSMZ_CHUNK_SHIFT", 0);
  int fb_cap_ = env_int("MSMZ_FB", 0);
  uint32_t s1_override_ = (uint32_t)env_int("MSMZ_S1", 0);
  uint32_t tail_n_ = (uint32_t)env_int("MSMZ_TAIL_N", REDUCE_TAIL_N);   // entries per window at which k_reduce_tail takes over
  uint32_t quad16_max_groups_ = (uint32_t)env_int("MSMZ_QUAD16", 8192);   // levels with at most this many groups use k_reduce_quad16
  Host64<F> host64_;
  bool no_spread_ = env_int("MSMZ_NO_SPREAD", 0) != 0;
  bool no_fold_ = env_int("MSMZ_NO_FOLD", 0) != 0;
  bool no_bucket_sums_ = env_int("MSMZ_NO_BUCKET_SUMS", 0) != 0;
  bool no_window_model_ = env_int("MSMZ_NO_WINDOW_MODEL", 0) != 0;
  bool force_atomic_sort_ = env_int("MSMZ_ATOMIC_SORT", 0) != 0;
  bool no_plan_top_ = env_int("MSMZ_NO_PLAN_TOP", 0) != 0;            // top-window bucket sets in full-size plan chunks
  bool no_fbt_ = env_int("MSMZ_NO_FBT", 0) != 0;                     // top window's bins as wide as the others
  bool no_sort_special_ = env_int("MSMZ_NO_SORT_SPECIAL", 0) != 0;   // generic sort kernels for every window size
  bool reduce2d_ = env_int("MSMZ_REDUCE2D", 1) != 0;          // two-dimensional bucket reduction (reduce2d_kernels.h); 0 = the grouped running sums
  int tail_skip_2d_ = env_int("MSMZ_TAIL_SKIP_2D", 1) > 2 ? 2 : env_int("MSMZ_TAIL_SKIP_2D", 1);
  uint32_t r2_nc_ = (uint32_t)env_int("MSMZ_R2_NC", 0);         // chunks per line (0 = automatic)
  uint32_t pairsum_x4_max_ = (uint32_t)env_int("MSMZ_PAIRSUM_X4", 16384);   // pair-sum levels with at most this many additions use DPP quads
  int batch_b_override_ = env_int("MSMZ_BATCH_B", 0);
  int retries_ = 0;            // MSMs redone with the proven GLV bound (test hook reads it)
  int glv_bits_assumed_ = 0;   // test hook (msmz_test_set_glv_bits): assumed bit length of a GLV half; 0 = GLV_BITS - 1
  DevBuf bsum_, f2desc_, tilecnt_, tileoff_, final_, desc_, bfin_, packed_, bins_, digits_, counts_, off_, cursor_, refs_, rscan_, partials_, slots_, red_[4], meta_, stage_, gen_table_;
  uint32_t h_round_pairs_[32] = {};
  uint32_t h_round_base_[32] = {};
  int basic_ev_[4] = {};
  bool basic_2d_ = false;   // the last msmBasic call reduced its buckets two-dimensionally (two results per bucket set)
  MsmMeta* h_meta_ = nullptr;
  uint32_t* h_final_ = nullptr;
};

}  // namespace msmz
