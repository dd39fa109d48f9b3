// C ABI of the MSM engine (include/msmz.h): curve dispatch + argument checking.
#include "instantiate.h"
namespace msmz {
#define X(F, Fr) MSMZ_INST_BATCH(F, Fr, MSMZ_EXTERN) MSMZ_INST_REDUCE(F, Fr, MSMZ_EXTERN) MSMZ_INST_MISC(F, Fr, MSMZ_EXTERN) MSMZ_INST_GEN(F, Fr, MSMZ_EXTERN)
MSMZ_WEIERSTRASS_FIELDS(X)
#undef X
#define X(F, Fr) MSMZ_INST_REDUCE_TE(F, Fr, MSMZ_EXTERN) MSMZ_INST_MISC_TE(F, Fr, MSMZ_EXTERN) MSMZ_INST_GEN_TE(F, Fr, MSMZ_EXTERN)
MSMZ_TE_FIELDS(X)
#undef X
}  // namespace msmz
#include "engine.h"
#include "../../include/msmz_test.h"

namespace msmz {

// Curve configurations: which field structs, which MSM paths.
template <class F_, class Fr_>
struct WeierCfg {
  using F = F_;
  using Fr = Fr_;
  static constexpr bool TE = false;
  static constexpr bool HAS_ENDO = true;
  static int run_msm(Engine<WeierCfg>& e, const Handle& p, const uint32_t* pts, const uint32_t* s, uint64_t n,
                     const msmz_opts& o, uint8_t* out, int* oi, msmz_log* log, int extra_bits) {
    if (o.buckets == MSMZ_BUCKETS_PROJECTIVE) return e.msm_weierstrass_projective(p, pts, s, n, o, out, oi, log);
    return e.msm_weierstrass_affine(p, pts, s, n, o, out, oi, log, extra_bits);
  }
};

template <class F_, class Fr_>
struct TeCfg {
  using F = F_;
  using Fr = Fr_;
  static constexpr bool TE = true;
  static constexpr bool HAS_ENDO = false;
  static int run_msm(Engine<TeCfg>& e, const Handle& p, const uint32_t* pts, const uint32_t* s, uint64_t n,
                     const msmz_opts& o, uint8_t* out, int* oi, msmz_log* log, int) {
    return e.msm_twisted_edwards(p, pts, s, n, o, out, oi, log);
  }
};

using CfgBls377 = WeierCfg<Bls377Fp, Bls377Fr>;
using CfgPallas = WeierCfg<PallasFp, PallasFr>;
using CfgBls381 = WeierCfg<Bls381Fp, Bls381Fr>;
using CfgEd377 = TeCfg<Ed377Fp, Ed377Fr>;

}  // namespace msmz

using namespace msmz;

struct msmz_ctx {
  int curve_id;
  IEngine* engine;
  int n_devices;
};

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

int msmz_ctx_fe_bytes(const msmz_ctx* c) { return c ? msmz_curve_fe_bytes(c->curve_id) : -1; }
int msmz_ctx_n_devices(const msmz_ctx* c) { return c ? c->n_devices : -1; }

static IEngine* make_engine(int curve_id, int device, int* st) {
  IEngine* eng = nullptr;
  *st = MSMZ_ERR_UNSUPPORTED;
  auto make = [&](auto* e) {
    *st = e->init();
    eng = e;
  };
  switch (curve_id) {
    case MSMZ_BLS12_377_G1: make(new Engine<CfgBls377>(device)); break;
    case MSMZ_PALLAS: make(new Engine<CfgPallas>(device)); break;
    case MSMZ_BLS12_381_G1: make(new Engine<CfgBls381>(device)); break;
    case MSMZ_ED_ON_BLS12_377: make(new Engine<CfgEd377>(device)); break;
    default: break;
  }
  if (*st != MSMZ_OK) {
    delete eng;
    eng = nullptr;
  }
  return eng;
}

int msmz_create(msmz_ctx** out, int curve_id, const int* device_ids, int n_devices) {
  if (!out) return MSMZ_ERR_ARG;
  *out = nullptr;
  if (msmz_curve_fe_bytes(curve_id) < 0) return MSMZ_ERR_ARG;
  if (n_devices == 0) return MSMZ_ERR_NO_DEVICE;   // there is no CPU backend
  if (n_devices < 0 || n_devices > MSMZ_MAX_DEVICES || !device_ids) return MSMZ_ERR_ARG;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return MSMZ_ERR_NO_DEVICE;
  for (int i = 0; i < n_devices; i++)
    if (device_ids[i] < 0 || device_ids[i] >= count) return MSMZ_ERR_ARG;
  int st = MSMZ_OK;
  if (n_devices == 1) {
    IEngine* eng = make_engine(curve_id, device_ids[0], &st);
    if (!eng) return st;
    *out = new msmz_ctx{curve_id, eng, 1};
    return MSMZ_OK;
  }
  // one engine (own HIP stream, own buffers) per listed device; the same id may be listed more than once
  std::vector<IEngine*> engines;
  for (int i = 0; i < n_devices; i++) {
    IEngine* eng = make_engine(curve_id, device_ids[i], &st);
    if (!eng) {
      for (IEngine* e : engines) delete e;
      return st;
    }
    engines.push_back(eng);
  }
  *out = new msmz_ctx{curve_id, new MultiEngine(curve_id, msmz_curve_fe_bytes(curve_id), engines), n_devices};
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
  return c ? c->engine->random_points(n, seed, GenMap{}, h) : MSMZ_ERR_ARG;
}
int msmz_random_scalars(msmz_ctx* c, uint64_t n, uint64_t seed, uint64_t* h) {
  return c ? c->engine->random_scalars(n, seed, GenMap{}, h) : MSMZ_ERR_ARG;
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

int msmz_test_set_glv_bits(msmz_ctx* c, int bits) { return c ? c->engine->test_set_glv_bits(bits) : MSMZ_ERR_ARG; }
int msmz_test_retries(msmz_ctx* c) { return c ? c->engine->test_retries() : -1; }
int msmz_test_field(msmz_ctx* c, int op, const uint8_t* a, const uint8_t* b, uint64_t n, uint8_t* out) {
  return c ? c->engine->test_field(op, a, b, n, out) : MSMZ_ERR_ARG;
}
int msmz_test_glv(msmz_ctx* c, const uint8_t* s, uint64_t n, uint8_t* s0, uint8_t* s1, uint8_t* neg) {
  return c ? c->engine->test_glv(s, n, s0, s1, neg) : MSMZ_ERR_ARG;
}
int msmz_test_digits(msmz_ctx* c, const uint8_t* s, uint64_t n, int cc, int K, int glv, uint32_t* digits) {
  return c ? c->engine->test_digits(s, n, cc, K, glv, digits) : MSMZ_ERR_ARG;
}
int msmz_test_sort(msmz_ctx* c, const uint8_t* s, uint64_t n, int cc, int glv, int force_fallback, uint32_t* geom,
                   uint32_t* off, uint64_t off_cap, uint32_t* refs, uint64_t refs_cap) {
  return c ? c->engine->test_sort(s, n, cc, glv, force_fallback, geom, off, off_cap, refs, refs_cap) : MSMZ_ERR_ARG;
}
int msmz_test_point(msmz_ctx* c, int op, const uint8_t* a, const uint8_t* ai, const uint8_t* b, const uint8_t* bi,
                    uint64_t n, uint8_t* out) {
  return c ? c->engine->test_point(op, a, ai, b, bi, n, out) : MSMZ_ERR_ARG;
}

int msmz_point_add(int curve_id, const uint8_t* a, int ai, const uint8_t* b, int bi, uint8_t* out, int* oi) {
  if (!out || !oi || (!a && !ai) || (!b && !bi)) return MSMZ_ERR_ARG;
  switch (curve_id) {
    case MSMZ_BLS12_377_G1: return host_point_add<Bls377Fp, false>(a, ai, b, bi, out, oi);
    case MSMZ_PALLAS: return host_point_add<PallasFp, false>(a, ai, b, bi, out, oi);
    case MSMZ_BLS12_381_G1: return host_point_add<Bls381Fp, false>(a, ai, b, bi, out, oi);
    case MSMZ_ED_ON_BLS12_377: return host_point_add<Ed377Fp, true>(a, 0, b, 0, out, oi);
    default: return MSMZ_ERR_UNSUPPORTED;
  }
}

}  // extern "C"
