/* Thin N-API addon over the C ABI of include/msmz.h -- the binding a TypeScript/JavaScript host uses
 * in place of the reference's wasm instance (src/field-msm.ts:42-133 exports + src/parallel.ts).
 * No arithmetic here: every function forwards to libmsmz.so.  N-API version 6 (BigInt) or later.
 *
 * Build: gcc -O2 -shared -fPIC -I/usr/include/node napi/msmz_napi.c -Lmsm_zprize_amd -lmsmz \
 *            -Wl,-rpath,'$ORIGIN/../msm_zprize_amd' -o js/msmz_napi.node
 */
#define NAPI_VERSION 6
#include <node_api.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/msmz.h"

#define NAPI_CALL(env, call)                                          \
  do {                                                                \
    napi_status s_ = (call);                                          \
    if (s_ != napi_ok) {                                              \
      napi_throw_error((env), NULL, "N-API call failed: " #call);     \
      return NULL;                                                    \
    }                                                                 \
  } while (0)

static napi_value throw_status(napi_env env, int st, const char* where) {
  char msg[256];
  strcpy(msg, where);
  strcat(msg, ": ");
  strncat(msg, msmz_strerror(st), sizeof(msg) - strlen(msg) - 1);
  char code[16];
  int n = 0, v = st;
  char tmp[16];
  do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
  for (int i = 0; i < n; i++) code[i] = tmp[n - 1 - i];
  code[n] = 0;
  napi_throw_error(env, code, msg);
  return NULL;
}

static int get_u64(napi_env env, napi_value v, uint64_t* out) {
  napi_valuetype t;
  if (napi_typeof(env, v, &t) != napi_ok) return 0;
  if (t == napi_bigint) {
    bool lossless;
    return napi_get_value_bigint_uint64(env, v, out, &lossless) == napi_ok;
  }
  double d;
  if (napi_get_value_double(env, v, &d) != napi_ok || d < 0) return 0;
  *out = (uint64_t)d;
  return 1;
}

static int get_ctx(napi_env env, napi_value v, msmz_ctx** ctx) {
  void* p = NULL;
  if (napi_get_value_external(env, v, &p) != napi_ok || !p) return 0;
  *ctx = *(msmz_ctx**)p;
  return *ctx != NULL;
}

static void ctx_finalize(napi_env env, void* data, void* hint) {
  (void)env; (void)hint;
  msmz_ctx** slot = (msmz_ctx**)data;
  if (*slot) msmz_destroy(*slot);
  free(slot);
}

/* create(curveId, deviceId | [deviceIds]) -> ctx   (an array = one engine per listed GPU, startThreads(n)) */
static napi_value Create(napi_env env, napi_callback_info info) {
  size_t argc = 2; napi_value argv[2];
  NAPI_CALL(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  int32_t curve = 0, devs[MSMZ_MAX_DEVICES] = {0};
  uint32_t ndev = 1;
  NAPI_CALL(env, napi_get_value_int32(env, argv[0], &curve));
  if (argc > 1) {
    bool is_arr = false;
    NAPI_CALL(env, napi_is_array(env, argv[1], &is_arr));
    if (is_arr) {
      NAPI_CALL(env, napi_get_array_length(env, argv[1], &ndev));
      if (ndev < 1 || ndev > MSMZ_MAX_DEVICES) return throw_status(env, MSMZ_ERR_ARG, "msmz_create");
      for (uint32_t i = 0; i < ndev; i++) {
        napi_value v;
        NAPI_CALL(env, napi_get_element(env, argv[1], i, &v));
        NAPI_CALL(env, napi_get_value_int32(env, v, &devs[i]));
      }
    } else {
      NAPI_CALL(env, napi_get_value_int32(env, argv[1], &devs[0]));
    }
  }
  msmz_ctx* ctx = NULL;
  int st = msmz_create(&ctx, curve, devs, (int)ndev);
  if (st) return throw_status(env, st, "msmz_create");
  msmz_ctx** slot = (msmz_ctx**)malloc(sizeof(*slot));
  *slot = ctx;
  napi_value ext;
  NAPI_CALL(env, napi_create_external(env, slot, ctx_finalize, NULL, &ext));
  return ext;
}

/* destroy(ctx) */
static napi_value Destroy(napi_env env, napi_callback_info info) {
  size_t argc = 1; napi_value argv[1];
  NAPI_CALL(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  void* p = NULL;
  if (napi_get_value_external(env, argv[0], &p) == napi_ok && p) {
    msmz_ctx** slot = (msmz_ctx**)p;
    if (*slot) msmz_destroy(*slot);
    *slot = NULL;
  }
  return NULL;
}

static napi_value make_handle(napi_env env, uint64_t h) {
  napi_value v;
  napi_create_double(env, (double)h, &v);
  return v;
}

/* uploadPoints(ctx, xyBuffer, infBufferOrNull, n) -> handle */
static napi_value UploadPoints(napi_env env, napi_callback_info info) {
  size_t argc = 4; napi_value argv[4];
  NAPI_CALL(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  msmz_ctx* ctx; if (!get_ctx(env, argv[0], &ctx)) return throw_status(env, MSMZ_ERR_ARG, "uploadPoints");
  void* xy; size_t xylen; NAPI_CALL(env, napi_get_buffer_info(env, argv[1], &xy, &xylen));
  void* inf = NULL; size_t inflen = 0; bool isbuf = false;
  napi_is_buffer(env, argv[2], &isbuf);
  if (isbuf) NAPI_CALL(env, napi_get_buffer_info(env, argv[2], &inf, &inflen));
  uint64_t n; if (!get_u64(env, argv[3], &n)) return throw_status(env, MSMZ_ERR_ARG, "uploadPoints");
  {
    int fbc = msmz_ctx_fe_bytes(ctx);   /* buffers must cover n records: 2 * fe_bytes each (+ one flag byte) */
    if (fbc <= 0 || n == 0 || xylen / (2 * (size_t)fbc) < n || (inf != NULL && inflen < n))
      return throw_status(env, MSMZ_ERR_ARG, "uploadPoints");
  }
  uint64_t h = 0;
  int st = msmz_upload_points(ctx, (const uint8_t*)xy, (const uint8_t*)inf, n, &h);
  if (st) return throw_status(env, st, "msmz_upload_points");
  return make_handle(env, h);
}

/* uploadScalars(ctx, buffer, n) -> handle */
static napi_value UploadScalars(napi_env env, napi_callback_info info) {
  size_t argc = 3; napi_value argv[3];
  NAPI_CALL(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  msmz_ctx* ctx; if (!get_ctx(env, argv[0], &ctx)) return throw_status(env, MSMZ_ERR_ARG, "uploadScalars");
  void* s; size_t slen; NAPI_CALL(env, napi_get_buffer_info(env, argv[1], &s, &slen));
  uint64_t n; if (!get_u64(env, argv[2], &n) || slen < 32 * n) return throw_status(env, MSMZ_ERR_ARG, "uploadScalars");
  uint64_t h = 0;
  int st = msmz_upload_scalars(ctx, (const uint8_t*)s, n, &h);
  if (st) return throw_status(env, st, "msmz_upload_scalars");
  return make_handle(env, h);
}

/* randomPoints(ctx, n, seed) / randomScalars(ctx, n, seed) -> handle */
static napi_value random_common(napi_env env, napi_callback_info info, int scalars) {
  size_t argc = 3; napi_value argv[3];
  NAPI_CALL(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  msmz_ctx* ctx; if (!get_ctx(env, argv[0], &ctx)) return throw_status(env, MSMZ_ERR_ARG, "random");
  uint64_t n, seed;
  if (!get_u64(env, argv[1], &n) || !get_u64(env, argv[2], &seed)) return throw_status(env, MSMZ_ERR_ARG, "random");
  uint64_t h = 0;
  int st = scalars ? msmz_random_scalars(ctx, n, seed, &h) : msmz_random_points(ctx, n, seed, &h);
  if (st) return throw_status(env, st, scalars ? "msmz_random_scalars" : "msmz_random_points");
  return make_handle(env, h);
}
static napi_value RandomPoints(napi_env env, napi_callback_info info) { return random_common(env, info, 0); }
static napi_value RandomScalars(napi_env env, napi_callback_info info) { return random_common(env, info, 1); }

/* downloadPoints(ctx, handle, first, count, feBytes) -> Buffer(xy) ; downloadScalars(ctx, handle, first, count) */
static napi_value DownloadPoints(napi_env env, napi_callback_info info) {
  size_t argc = 5; napi_value argv[5];
  NAPI_CALL(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  msmz_ctx* ctx; if (!get_ctx(env, argv[0], &ctx)) return throw_status(env, MSMZ_ERR_ARG, "downloadPoints");
  uint64_t h, first, count, fb;
  if (!get_u64(env, argv[1], &h) || !get_u64(env, argv[2], &first) || !get_u64(env, argv[3], &count) ||
      !get_u64(env, argv[4], &fb))
    return throw_status(env, MSMZ_ERR_ARG, "downloadPoints");
  void* data; napi_value buf;
  NAPI_CALL(env, napi_create_buffer(env, (size_t)(2 * fb * count), &data, &buf));
  int st = msmz_download_points(ctx, h, first, count, (uint8_t*)data, NULL);
  if (st) return throw_status(env, st, "msmz_download_points");
  return buf;
}
static napi_value DownloadScalars(napi_env env, napi_callback_info info) {
  size_t argc = 4; napi_value argv[4];
  NAPI_CALL(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  msmz_ctx* ctx; if (!get_ctx(env, argv[0], &ctx)) return throw_status(env, MSMZ_ERR_ARG, "downloadScalars");
  uint64_t h, first, count;
  if (!get_u64(env, argv[1], &h) || !get_u64(env, argv[2], &first) || !get_u64(env, argv[3], &count))
    return throw_status(env, MSMZ_ERR_ARG, "downloadScalars");
  void* data; napi_value buf;
  NAPI_CALL(env, napi_create_buffer(env, (size_t)(32 * count), &data, &buf));
  int st = msmz_download_scalars(ctx, h, first, count, (uint8_t*)data);
  if (st) return throw_status(env, st, "msmz_download_scalars");
  return buf;
}

static napi_value Free(napi_env env, napi_callback_info info) {
  size_t argc = 2; napi_value argv[2];
  NAPI_CALL(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  msmz_ctx* ctx; if (!get_ctx(env, argv[0], &ctx)) return throw_status(env, MSMZ_ERR_ARG, "free");
  uint64_t h; if (!get_u64(env, argv[1], &h)) return throw_status(env, MSMZ_ERR_ARG, "free");
  int st = msmz_free(ctx, h);
  if (st) return throw_status(env, st, "msmz_free");
  return NULL;
}

static int32_t opt_i32(napi_env env, napi_value obj, const char* key) {
  napi_valuetype t;
  if (napi_typeof(env, obj, &t) != napi_ok || t != napi_object) return 0;
  napi_value v; bool has = false;
  if (napi_has_named_property(env, obj, key, &has) != napi_ok || !has) return 0;
  if (napi_get_named_property(env, obj, key, &v) != napi_ok) return 0;
  if (napi_typeof(env, v, &t) != napi_ok) return 0;
  if (t == napi_boolean) { bool b; napi_get_value_bool(env, v, &b); return b ? 1 : 0; }
  int32_t r = 0;
  if (t == napi_number) napi_get_value_int32(env, v, &r);
  return r;
}

static void set_num(napi_env env, napi_value obj, const char* key, double v) {
  napi_value n; napi_create_double(env, v, &n); napi_set_named_property(env, obj, key, n);
}

/* msm(ctx, pointsHandle, scalars (handle number or Buffer), n, feBytes, opts) -> {xy, isInf, log} */
static napi_value Msm(napi_env env, napi_callback_info info) {
  size_t argc = 6; napi_value argv[6];
  NAPI_CALL(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  msmz_ctx* ctx; if (!get_ctx(env, argv[0], &ctx)) return throw_status(env, MSMZ_ERR_ARG, "msm");
  uint64_t ph, n, fb;
  if (!get_u64(env, argv[1], &ph) || !get_u64(env, argv[3], &n) || !get_u64(env, argv[4], &fb))
    return throw_status(env, MSMZ_ERR_ARG, "msm");
  msmz_opts o; memset(&o, 0, sizeof(o));
  if (argc > 5) {
    o.c = opt_i32(env, argv[5], "c");
    o.glv = opt_i32(env, argv[5], "glv");
    o.safe = opt_i32(env, argv[5], "safe");
    o.buckets = opt_i32(env, argv[5], "buckets");
    o.timing = opt_i32(env, argv[5], "timing");
    o.reserved[0] = opt_i32(env, argv[5], "reduceAffine");
  }
  void* data; napi_value xy;
  NAPI_CALL(env, napi_create_buffer(env, (size_t)(2 * fb), &data, &xy));
  int is_inf = 0; msmz_log log;
  bool isbuf = false; napi_is_buffer(env, argv[2], &isbuf);
  int st;
  if (isbuf) {
    void* s; size_t slen; NAPI_CALL(env, napi_get_buffer_info(env, argv[2], &s, &slen));
    if (slen < 32 * n) return throw_status(env, MSMZ_ERR_ARG, "msm");
    st = msmz_msm(ctx, ph, (const uint8_t*)s, n, &o, (uint8_t*)data, &is_inf, &log);
  } else {
    uint64_t sh; if (!get_u64(env, argv[2], &sh)) return throw_status(env, MSMZ_ERR_ARG, "msm");
    st = msmz_msm_resident(ctx, ph, sh, n, &o, (uint8_t*)data, &is_inf, &log);
  }
  if (st) return throw_status(env, st, "msmz_msm");
  napi_value res, jlog, inf, stages, rounds;
  NAPI_CALL(env, napi_create_object(env, &res));
  NAPI_CALL(env, napi_create_object(env, &jlog));
  napi_get_boolean(env, is_inf != 0, &inf);
  napi_set_named_property(env, res, "xy", xy);
  napi_set_named_property(env, res, "isInf", inf);
  static const char* names[MSMZ_N_STAGES] = {"digits", "scan", "scatter", "plan", "accumulate", "reduce", "final", "total"};
  NAPI_CALL(env, napi_create_object(env, &stages));
  for (int i = 0; i < MSMZ_N_STAGES; i++) set_num(env, stages, names[i], log.stage_ms[i]);
  napi_set_named_property(env, jlog, "stageMs", stages);
  set_num(env, jlog, "c", log.c); set_num(env, jlog, "K", log.K); set_num(env, jlog, "rounds", log.rounds);
  set_num(env, jlog, "glv", log.glv); set_num(env, jlog, "nEntries", (double)log.n_entries);
  set_num(env, jlog, "nPairs", (double)log.n_pairs); set_num(env, jlog, "maxBucket", log.max_bucket);
  set_num(env, jlog, "scatterKernelMs", log.scatter_kernel_ms);
  NAPI_CALL(env, napi_create_array_with_length(env, 32, &rounds));
  for (uint32_t i = 0; i < 32; i++) { napi_value v; napi_create_double(env, log.batch_add_ms[i], &v); napi_set_element(env, rounds, i, v); }
  napi_set_named_property(env, jlog, "batchAddMs", rounds);
  napi_set_named_property(env, res, "log", jlog);
  return res;
}

/* pointAdd(curveId, aXy|null, bXy|null, feBytes) -> {xy, isInf}  (null = infinity) */
static napi_value PointAdd(napi_env env, napi_callback_info info) {
  size_t argc = 4; napi_value argv[4];
  NAPI_CALL(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  int32_t curve; NAPI_CALL(env, napi_get_value_int32(env, argv[0], &curve));
  uint64_t fb; if (!get_u64(env, argv[3], &fb)) return throw_status(env, MSMZ_ERR_ARG, "pointAdd");
  void *a = NULL, *b = NULL; size_t la, lb; bool ia = false, ib = false;
  napi_is_buffer(env, argv[1], &ia); napi_is_buffer(env, argv[2], &ib);
  if (ia) NAPI_CALL(env, napi_get_buffer_info(env, argv[1], &a, &la));
  if (ib) NAPI_CALL(env, napi_get_buffer_info(env, argv[2], &b, &lb));
  void* data; napi_value xy;
  NAPI_CALL(env, napi_create_buffer(env, (size_t)(2 * fb), &data, &xy));
  int is_inf = 0;
  int st = msmz_point_add(curve, (const uint8_t*)a, a == NULL, (const uint8_t*)b, b == NULL, (uint8_t*)data, &is_inf);
  if (st) return throw_status(env, st, "msmz_point_add");
  napi_value res, inf;
  NAPI_CALL(env, napi_create_object(env, &res));
  napi_get_boolean(env, is_inf != 0, &inf);
  napi_set_named_property(env, res, "xy", xy);
  napi_set_named_property(env, res, "isInf", inf);
  return res;
}

static napi_value FeBytes(napi_env env, napi_callback_info info) {
  size_t argc = 1; napi_value argv[1];
  NAPI_CALL(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  int32_t curve; NAPI_CALL(env, napi_get_value_int32(env, argv[0], &curve));
  napi_value v; napi_create_int32(env, msmz_curve_fe_bytes(curve), &v);
  return v;
}

static napi_value Init(napi_env env, napi_value exports) {
  static const struct { const char* name; napi_callback fn; } fns[] = {
      {"create", Create}, {"destroy", Destroy}, {"uploadPoints", UploadPoints}, {"uploadScalars", UploadScalars},
      {"randomPoints", RandomPoints}, {"randomScalars", RandomScalars}, {"downloadPoints", DownloadPoints},
      {"downloadScalars", DownloadScalars}, {"free", Free}, {"msm", Msm}, {"pointAdd", PointAdd}, {"feBytes", FeBytes}};
  for (size_t i = 0; i < sizeof(fns) / sizeof(fns[0]); i++) {
    napi_value f;
    if (napi_create_function(env, fns[i].name, NAPI_AUTO_LENGTH, fns[i].fn, NULL, &f) != napi_ok) return NULL;
    napi_set_named_property(env, exports, fns[i].name, f);
  }
  return exports;
}

NAPI_MODULE(NODE_GYP_MODULE_NAME, Init)
