/* msmz -- C ABI of the MI355X-native Pippenger MSM engine.
 *
 * This is the drop-in boundary for the reference's MSM hot path: the functions below are what an
 * N-API / ctypes binding calls in place of the reference's wasm-backed
 *   Curve.Parallel.{msm, msmUnsafe, msmProjective, pointsFromBytes, scalarsFromBytes,
 *                   randomPointsFast, randomScalars}      (src/parallel.ts:89-158, 209-271)
 * and the result read-back   Projective.toAffine + Affine.toBigint   (scripts/msm-weierstrass.ts:90-92).
 * See INTEGRATION.md for the binding a maintainer would add.
 *
 * Conventions: every function returns an int status (0 = MSMZ_OK; msmz_strerror() explains the
 * rest); no C++ types cross the boundary; the caller owns all host buffers; a context drives the
 * GPU(s) it was created for and is not thread-safe (one caller at a time per context); there is no CPU fallback --
 * creating a context without a usable HIP device fails.
 *
 * Wire formats (same as the reference's byte route, parallel.ts:97-133, 209-249):
 *   point  = x || y, each coordinate little-endian canonical (non-Montgomery), fe_bytes = 48
 *            (BLS12-377 / BLS12-381) or 32 (Pallas, ed-on-bls12-377); optional per-point infinity flag
 *   scalar = 32 bytes little-endian, value < group order
 *   result = canonical affine x || y (both < p) + infinity flag -- "bit-exact" is defined on this.
 */
#ifndef MSMZ_H
#define MSMZ_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* curve ids: src/concrete/{bls12-377,pasta,bls12-381,ed-on-bls12-377}.params.ts */
enum {
  MSMZ_BLS12_377_G1 = 0,
  MSMZ_PALLAS = 1,
  MSMZ_BLS12_381_G1 = 2,
  MSMZ_ED_ON_BLS12_377 = 3
};

enum {
  MSMZ_OK = 0,
  MSMZ_ERR_ARG = 1,          /* bad argument (null pointer, unknown curve / handle, N mismatch) */
  MSMZ_ERR_NO_DEVICE = 2,    /* no usable HIP device: there is no CPU fallback */
  MSMZ_ERR_HIP = 3,          /* a HIP call failed (out of memory, launch failure) */
  MSMZ_ERR_UNSUPPORTED = 4,  /* option combination not available for this curve */
  MSMZ_ERR_DEGENERATE = 5,   /* msmUnsafe hit P + (+-P): a batch inversion saw a zero denominator
                                (the reference traps with wasm `unreachable`, inverse.ts:198-199) */
  MSMZ_ERR_RANGE = 6         /* a scalar is >= the group order / a coordinate is >= p */
};

enum { MSMZ_BUCKETS_AFFINE = 0, MSMZ_BUCKETS_PROJECTIVE = 1 };

/* Per-call options: the reference's `{c, useSafeAdditions}` (msm-batched-affine.ts:79-82) plus the
 * choices BASELINE.json's configs name (GLV on/off, affine vs projective buckets). 0 = default. */
typedef struct msmz_opts {
  int32_t c;        /* window size in bits; 0 = pick from N like windowSizeAffine (msm-common.ts:15-21) */
  int32_t glv;      /* 1 = GLV endomorphism split (the reference always splits on Weierstrass curves), 0 = off,
                     * -1 = the engine picks (GLV below 2^21 points, the measured crossover; the result is the same) */
  int32_t safe;     /* 1 = msm (handles equal / opposite / infinity points), 0 = msmUnsafe */
  int32_t buckets;  /* MSMZ_BUCKETS_AFFINE (batched-affine) or MSMZ_BUCKETS_PROJECTIVE (msmProjective) */
  int32_t timing;   /* 1 = fill msmz_log stage timings with HIP events (the reference's tic/toc log) */
  int32_t reserved[3]; /* reserved[0] = 1: first level of the bucket reduction by batched-affine additions
                        * (reduceBucketsAffine, msm-batched-affine-single-thread.ts:522-667) instead of XYZZ running
                        * sums; same result, measured slower on MI355X (profiles/r02_reduce_ab.txt): default 0 */
} msmz_opts;

/* Stage timings + counts, the analogue of the `log` array msm() returns (msm-common.ts:192-230). */
enum {
  MSMZ_ST_DIGITS = 0,      /* GLV + signed digits + histogram */
  MSMZ_ST_SCAN = 1,        /* bucket offsets */
  MSMZ_ST_SCATTER = 2,     /* counting-sort scatter of references (the HBM-bound kernel) */
  MSMZ_ST_PLAN = 3,        /* per-round pair offsets */
  MSMZ_ST_ACCUMULATE = 4,  /* all batched-affine tree rounds (or projective bucket accumulation) */
  MSMZ_ST_REDUCE = 5,      /* bucket reduction */
  MSMZ_ST_FINAL = 6,       /* window sums -> result (host) */
  MSMZ_ST_TOTAL = 7,       /* whole call, host wall clock */
  MSMZ_N_STAGES = 8
};

typedef struct msmz_log {
  float stage_ms[MSMZ_N_STAGES];
  int32_t c, K, rounds, glv;
  uint64_t n_entries;       /* non-zero digits = bucket insertions ("point-adds" of the metric) */
  uint64_t n_pairs;         /* affine additions performed in the tree rounds */
  uint32_t max_bucket;
  uint32_t scatter_launches;
  float scatter_kernel_ms;  /* duration of the scatter kernel alone (roofline numerator's time) */
  float batch_add_ms[32];   /* per tree round */
} msmz_log;

typedef struct msmz_ctx msmz_ctx;

/* device_ids / n_devices: the GPU(s) this context drives, 1 <= n_devices <= MSMZ_MAX_DEVICES (0 = "CPU backend":
 * refused with MSMZ_ERR_NO_DEVICE).  Replaces startThreads(n) (parallel.ts:291-315, threads.ts:132-359): with
 * n_devices > 1 the context owns one engine + HIP stream + host thread per device, every uploaded / generated set is
 * split over the devices in contiguous blocks of 2^16 entries dealt round-robin (so the first n entries of a set are a
 * prefix on every device), msmz_msm runs the whole pipeline on each device's share concurrently and adds the partial
 * sums on the host (no inter-GPU traffic).  The same device id may be listed more than once
 * (used to rehearse the scheduler on one GPU).  The other route to multi-GPU -- one process and one single-device
 * context per GPU, partial sums combined with msmz_point_add -- is what bench.py --gpus N uses. */
#define MSMZ_MAX_DEVICES 8
int msmz_create(msmz_ctx** ctx, int curve_id, const int* device_ids, int n_devices);
void msmz_destroy(msmz_ctx* ctx);            /* stopThreads() + frees every handle */
const char* msmz_strerror(int status);
int msmz_curve_fe_bytes(int curve_id);       /* 48 or 32; -1 for an unknown curve */
int msmz_ctx_fe_bytes(const msmz_ctx* ctx);  /* fe_bytes of the context's curve; -1 for a null context */
int msmz_ctx_n_devices(const msmz_ctx* ctx); /* number of engines (GPUs) the context drives */

/* Point sets live on the GPU across MSMs, like the reference keeps them in wasm memory
 * (scripts/msm-weierstrass.ts:19-35).  pointsFromBytes (parallel.ts:97-112 / 209-232). */
int msmz_upload_points(msmz_ctx* ctx, const uint8_t* xy_le, const uint8_t* is_inf /* nullable */, uint64_t n,
                       uint64_t* handle);
/* scalarsFromBytes (parallel.ts:114-133) */
int msmz_upload_scalars(msmz_ctx* ctx, const uint8_t* scalars_le32, uint64_t n, uint64_t* handle);
/* randomPointsFast / randomScalars (curve-random.ts:14-92, 151-194), seeded and generated on the GPU:
 * point i = a_i * G with a_i = splitmix64(seed, i) (64-bit), scalar i = rejection-sampled 32 bytes. */
int msmz_random_points(msmz_ctx* ctx, uint64_t n, uint64_t seed, uint64_t* handle);
int msmz_random_scalars(msmz_ctx* ctx, uint64_t n, uint64_t seed, uint64_t* handle);
int msmz_download_points(msmz_ctx* ctx, uint64_t handle, uint64_t first, uint64_t count, uint8_t* xy_le,
                         uint8_t* is_inf /* nullable */);
int msmz_download_scalars(msmz_ctx* ctx, uint64_t handle, uint64_t first, uint64_t count, uint8_t* scalars_le32);
int msmz_free(msmz_ctx* ctx, uint64_t handle);

/* The MSM: sum_i scalar_i * point_i over the first n entries.  Scalars either come from a host
 * buffer (copied to the GPU inside the call) or are already resident (scalars_handle).
 * out_xy_le: 2*fe_bytes.  Curve.Parallel.msm / msmUnsafe / msmProjective. */
int msmz_msm(msmz_ctx* ctx, uint64_t points_handle, const uint8_t* scalars_le32, uint64_t n, const msmz_opts* opts,
             uint8_t* out_xy_le, int* out_is_inf, msmz_log* log /* nullable */);
int msmz_msm_resident(msmz_ctx* ctx, uint64_t points_handle, uint64_t scalars_handle, uint64_t n,
                      const msmz_opts* opts, uint8_t* out_xy_le, int* out_is_inf, msmz_log* log /* nullable */);

/* Host-side group addition of two canonical affine results: combines per-GPU partial sums
 * (the reference's "partition sum" step, msm-batched-affine.ts:300-307). */
int msmz_point_add(int curve_id, const uint8_t* a_xy_le, int a_is_inf, const uint8_t* b_xy_le, int b_is_inf,
                   uint8_t* out_xy_le, int* out_is_inf);

#ifdef __cplusplus
}
#endif
#endif /* MSMZ_H */
