// C ABI of the MSM engine (include/msmz.h): curve dispatch + argument checking.
#include "instantiate.h"
namespace msmz {
#define X(F, Fr) MSMZ_INST_BATCH(F, Fr, MSMZ_EXTERN) MSMZ_INST_REDUCE(F, Fr, MSMZ_EXTERN) MSMZ_INST_MISC(F, Fr, MSMZ_EXTERN) MSMZ_INST_GEN(F, Fr, MSMZ_EXTERN)
MSMZ_WEIERSTRASS_FIELDS(X)
#undef X
}  // namespace msmz
#include "engine.h"

namespace msmz {

template <class Cfg>
static int run_weierstrass(Engine<Cfg>& e, const Handle& pts, const uint32_t* d_scalars, uint64_t n,
                           const msmz_opts& opt, uint8_t* out, int* out_inf, msmz_log* log) {
  if (opt.buckets == MSMZ_BUCKETS_PROJECTIVE) return MSMZ_ERR_UNSUPPORTED;
  return e.msm_weierstrass_affine(pts, d_scalars, n, opt, out, out_inf, log);
}

struct CfgBls377 {
  using F = Bls377Fp;
  using Fr = Bls377Fr;
  static constexpr bool HAS_ENDO = true;
  static constexpr int BATCH_T = MSMZ_BATCH_T;
  static int run_msm(Engine<CfgBls377>& e, const Handle& p, const uint32_t* s, uint64_t n, const msmz_opts& o,
                     uint8_t* out, int* oi, msmz_log* log) {
    return run_weierstrass(e, p, s, n, o, out, oi, log);
  }
};

}  // namespace msmz

using namespace msmz;

struct msmz_ctx {
  int curve_id;
  IEngine* engine;
};

template <class F>
static int point_add_w(const uint8_t* a, int ai, const uint8_t* b, int bi, uint8_t* out, int* oi) {
  constexpr int NW = F::NW;
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
  return MSMZ_OK;
}

extern "C" {

const char* msmz_strerror(int status) {
  switch (status) {
    case MSMZ_OK: return "ok";
    case MSMZ_ERR_ARG: return "bad argument";
    case MSMZ_ERR_NO_DEVICE: return "no usable HIP device (there is no CPU fallback)";
    case MSMZ_ERR_HIP: return "HIP runtime error";
    case MSMZ_ERR_UNSUPPORTED: return "option combination not supported for this curve";
    case MSMZ_ERR_DEGENERATE: return "unsafe batched addition hit a zero denominator (equal or opposite points); use safe=1";
    case MSMZ_ERR_RANGE: return "scalar >= group order or coordinate >= field modulus";
    default: return "unknown status";
  }
}

int msmz_curve_fe_bytes(int curve_id) {
  switch (curve_id) {
    case MSMZ_BLS12_377_G1:
    case MSMZ_BLS12_381_G1: return 48;
    case MSMZ_PALLAS:
    case MSMZ_ED_ON_BLS12_377: return 32;
    default: return -1;
  }
}

int msmz_create(msmz_ctx** out, int curve_id, const int* device_ids, int n_devices) {
  if (!out) return MSMZ_ERR_ARG;
  *out = nullptr;
  if (msmz_curve_fe_bytes(curve_id) < 0) return MSMZ_ERR_ARG;
  if (n_devices != 1 || !device_ids) return n_devices == 0 ? MSMZ_ERR_NO_DEVICE : MSMZ_ERR_ARG;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return MSMZ_ERR_NO_DEVICE;
  if (device_ids[0] < 0 || device_ids[0] >= count) return MSMZ_ERR_ARG;
  IEngine* eng = nullptr;
  int st = MSMZ_ERR_UNSUPPORTED;
  switch (curve_id) {
    case MSMZ_BLS12_377_G1: {
      auto* e = new Engine<CfgBls377>(device_ids[0]);
      st = e->init();
      eng = e;
      break;
    }
    default: break;
  }
  if (st != MSMZ_OK) {
    delete eng;
    return st;
  }
  *out = new msmz_ctx{curve_id, eng};
  return MSMZ_OK;
}

void msmz_destroy(msmz_ctx* ctx) {
  if (!ctx) return;
  delete ctx->engine;
  delete ctx;
}

int msmz_upload_points(msmz_ctx* c, const uint8_t* xy, const uint8_t* inf, uint64_t n, uint64_t* h) {
  return c ? c->engine->upload_points(xy, inf, n, h) : MSMZ_ERR_ARG;
}
int msmz_upload_scalars(msmz_ctx* c, const uint8_t* s, uint64_t n, uint64_t* h) {
  return c ? c->engine->upload_scalars(s, n, h) : MSMZ_ERR_ARG;
}
int msmz_random_points(msmz_ctx* c, uint64_t n, uint64_t seed, uint64_t* h) {
  return c ? c->engine->random_points(n, seed, h) : MSMZ_ERR_ARG;
}
int msmz_random_scalars(msmz_ctx* c, uint64_t n, uint64_t seed, uint64_t* h) {
  return c ? c->engine->random_scalars(n, seed, h) : MSMZ_ERR_ARG;
}
int msmz_download_points(msmz_ctx* c, uint64_t h, uint64_t first, uint64_t count, uint8_t* xy, uint8_t* inf) {
  return c ? c->engine->download_points(h, first, count, xy, inf) : MSMZ_ERR_ARG;
}
int msmz_download_scalars(msmz_ctx* c, uint64_t h, uint64_t first, uint64_t count, uint8_t* s) {
  return c ? c->engine->download_scalars(h, first, count, s) : MSMZ_ERR_ARG;
}
int msmz_free(msmz_ctx* c, uint64_t h) { return c ? c->engine->free_handle(h) : MSMZ_ERR_ARG; }

int msmz_msm(msmz_ctx* c, uint64_t ph, const uint8_t* scalars, uint64_t n, const msmz_opts* o, uint8_t* out,
             int* out_inf, msmz_log* log) {
  if (!c || !scalars) return MSMZ_ERR_ARG;
  return c->engine->msm(ph, scalars, 0, n, o, out, out_inf, log);
}
int msmz_msm_resident(msmz_ctx* c, uint64_t ph, uint64_t sh, uint64_t n, const msmz_opts* o, uint8_t* out,
                      int* out_inf, msmz_log* log) {
  if (!c) return MSMZ_ERR_ARG;
  return c->engine->msm(ph, nullptr, sh, n, o, out, out_inf, log);
}

int msmz_point_add(int curve_id, const uint8_t* a, int ai, const uint8_t* b, int bi, uint8_t* out, int* oi) {
  if (!out || !oi || (!a && !ai) || (!b && !bi)) return MSMZ_ERR_ARG;
  switch (curve_id) {
    case MSMZ_BLS12_377_G1: return point_add_w<Bls377Fp>(a, ai, b, bi, out, oi);
    default: return MSMZ_ERR_UNSUPPORTED;
  }
}

}  // extern "C"
