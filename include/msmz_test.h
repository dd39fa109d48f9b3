/* msmz -- stage-level test hooks of the C ABI.
 *
 * The reference tests every wasm routine against its bigint twin (src/field.test.ts:159-211,
 * src/curve-projective.test.ts:77-209, src/glv/glv-test.ts:83-125, src/testing/equivalent-wasm.ts:97-147).  These
 * entry points give the parity tests the same granularity on the device: each one runs ONE device routine of the
 * MSM pipeline on caller-supplied inputs and returns its raw output.  They are not part of the drop-in boundary and
 * never take part in an MSM.  Statuses and conventions as in msmz.h; on a multi-device context they use its first
 * engine.
 *
 * Field elements travel as fe_bytes little-endian bytes holding a Montgomery residue in the engine's memory format
 * (radix R = 2^392 for the 377/381-bit fields, 2^261 for the 255-bit ones; lazily reduced: any value in [0, 4p));
 * results are canonical (< p).
 */
#ifndef MSMZ_TEST_H
#define MSMZ_TEST_H

#include "msmz.h"

#ifdef __cplusplus
extern "C" {
#endif

enum {
  MSMZ_TF_MUL = 0,          /* a*b/R        (fe_mul, Montgomery product)                         */
  MSMZ_TF_SQR = 1,          /* a*a/R        (fe_sqr)                                             */
  MSMZ_TF_ADD = 2,          /* a+b          (lazy add, canonicalized on output)                  */
  MSMZ_TF_SUB = 3,          /* a-b                                                               */
  MSMZ_TF_INVERSE = 4,      /* R^2/a        (fe_inverse: per-lane binary GCD; 0 for a = 0)       */
  MSMZ_TF_INVERSE_WAVE = 5, /* R^2/a        (fe_inverse_wave: the wave-wide form of k_batch_add) */
  MSMZ_TF_ROUNDTRIP = 6,    /* a            (fe_store -> fe_unpack: the memory format)           */
  MSMZ_TF_IS_ZERO = 7,      /* a == b mod p ? 1 : 0 in byte 0  (fe_is_zero of a lazy difference) */
  MSMZ_TF_SLOT_ROUNDTRIP = 8 /* a+b after (a, b) went through a slot record of the tree rounds    */
};
enum {
  MSMZ_TP_ADD = 0,     /* accumulator + accumulator (XYZZ add / extended twisted-Edwards add), all edge cases */
  MSMZ_TP_ADD_X4 = 1,  /* the 4-lane form used by the upper reduction levels                                 */
  MSMZ_TP_DBL = 3,     /* doubling of the first operand                                                      */
  MSMZ_TP_DBL_X4 = 4   /* 2 (a + b): 4-lane addition, then the 4-lane doubling of that (general) accumulator  */
};

/* GLV half-scalar bound (src/wasm/glv.ts:216-226 `maxBits`).  The engine sizes the windows for halves below 2^127 and
 * lets the slicing kernel flag a longer half, in which case the MSM is redone with windows for the analytic bound
 * (GLV_PROVEN_BITS, tools/gen_constants.py).  No real scalar is known to take that path, so this hook shrinks the
 * ASSUMED bit length (8 .. 127; 0 restores the default): ordinary halves then overflow, the flag is raised and the
 * redone MSM must still equal the oracle.  msmz_test_retries = number of MSMs (per-engine passes) redone so far. */
int msmz_test_set_glv_bits(msmz_ctx* ctx, int bits);
int msmz_test_retries(msmz_ctx* ctx);
/* out[i] = op(a[i], b[i]) for i < n; a, b, out: n * fe_bytes */
int msmz_test_field(msmz_ctx* ctx, int op, const uint8_t* a, const uint8_t* b, uint64_t n, uint8_t* out);
/* GLV split of n 32-byte scalars: s0, s1 = magnitudes (16 bytes each), neg = 2 sign bytes per scalar
 * (glv.ts:68-169 `decompose`).  MSMZ_ERR_UNSUPPORTED on a curve without endomorphism. */
int msmz_test_glv(msmz_ctx* ctx, const uint8_t* scalars_le32, uint64_t n, uint8_t* s0_le16, uint8_t* s1_le16,
                  uint8_t* neg);
/* signed c-bit digits as the sort kernels slice them (msm-batched-affine.ts:180-199): K digits per (half) scalar,
 * digits[(h*n + i)*K + k] = l | negate << 31 */
int msmz_test_digits(msmz_ctx* ctx, const uint8_t* scalars_le32, uint64_t n, int c, int K, int glv, uint32_t* digits);
/* the bucket sort alone: scalars -> bucket offsets `off` (nb + 1 words) and sorted references `refs`
 * (index | negate << 31; GLV: index >= n = endomorphism half of point index - n).  Geometry in geom[8] =
 * {c, K, Keff, L, nb, n_entries, max_bucket, spread}.  off / refs may be NULL to query the geometry only;
 * force_fallback = 1 runs the one-pass atomic sort.  Capacities in words. */
int msmz_test_sort(msmz_ctx* ctx, const uint8_t* scalars_le32, uint64_t n, int c, int glv, int force_fallback,
                   uint32_t* geom, uint32_t* off, uint64_t off_cap, uint32_t* refs, uint64_t refs_cap);
/* point arithmetic on canonical affine inputs (x || y, 2*fe_bytes; infinity flags nullable, Weierstrass only):
 * out[i] = op(a[i], b[i]) as canonical affine, all-zero = infinity */
int msmz_test_point(msmz_ctx* ctx, int op, const uint8_t* a_xy, const uint8_t* a_inf, const uint8_t* b_xy,
                    const uint8_t* b_inf, uint64_t n, uint8_t* out_xy);

#ifdef __cplusplus
}
#endif
#endif /* MSMZ_TEST_H */
