// Explicit-instantiation lists: each (kernel family, curve) pair is compiled in its own translation
// unit (kern_<family>.hip with -DMSMZ_CURVE=<id>) so the build parallelizes; msmz.hip only sees
// `extern template` declarations.
#pragma once
#include "gen_kernels.h"
#include "kernels.h"
#include "test_kernels.h"

// curve ids as in include/msmz.h
#if !defined(MSMZ_CURVE) || MSMZ_CURVE == 0
#define MSMZ_W0(X) X(Bls377Fp, Bls377Fr)
#else
#define MSMZ_W0(X)
#endif
#if !defined(MSMZ_CURVE) || MSMZ_CURVE == 1
#define MSMZ_W1(X) X(PallasFp, PallasFr)
#else
#define MSMZ_W1(X)
#endif
#if !defined(MSMZ_CURVE) || MSMZ_CURVE == 2
#define MSMZ_W2(X) X(Bls381Fp, Bls381Fr)
#else
#define MSMZ_W2(X)
#endif
#if !defined(MSMZ_CURVE) || MSMZ_CURVE == 3
#define MSMZ_T3(X) X(Ed377Fp, Ed377Fr)
#else
#define MSMZ_T3(X)
#endif
#define MSMZ_WEIERSTRASS_FIELDS(X) MSMZ_W0(X) MSMZ_W1(X) MSMZ_W2(X)
#define MSMZ_TE_FIELDS(X) MSMZ_T3(X)

#ifndef MSMZ_BATCH_T
#define MSMZ_BATCH_T 256
#endif
#ifndef MSMZ_BATCH_OCC
#define MSMZ_BATCH_OCC 4
#endif
#ifndef MSMZ_BATCH_BMAX
#define MSMZ_BATCH_BMAX 16
#endif

#define MSMZ_INST_BATCH(F, Fr, PFX)                                                                               \
  PFX template __global__ void k_batch_add<F, MSMZ_BATCH_T, true, MSMZ_BATCH_OCC, MSMZ_BATCH_BMAX>(              \
      uint32_t*, const uint32_t*, const uint2*, uint32_t, uint32_t, int, MsmMeta*);                              \
  PFX template __global__ void k_batch_add<F, MSMZ_BATCH_T, false, MSMZ_BATCH_OCC, MSMZ_BATCH_BMAX>(             \
      uint32_t*, const uint32_t*, const uint2*, uint32_t, uint32_t, int, MsmMeta*);

#define MSMZ_INST_POLICY(P, PFX)                                                                                 \
  PFX template __global__ void k_reduce_quad<P>(uint32_t*, uint32_t*, const uint32_t*, const uint32_t*, uint32_t, \
                                                uint32_t, uint32_t);                                             \
  PFX template __global__ void k_reduce_quad16<P>(uint32_t*, uint32_t*, const uint32_t*, const uint32_t*,        \
                                                  uint32_t, uint32_t, uint32_t);                                 \
  PFX template __global__ void k_reduce_tail<P>(uint32_t*, uint32_t*, uint32_t*, uint32_t*, uint32_t*, uint32_t, \
                                                uint32_t);                                                       \
  PFX template __global__ void k_pairsum<P>(uint32_t*, const uint32_t*, uint32_t);                               \
  PFX template __global__ void k_fill_neutral<P>(uint32_t*, uint32_t);                                           \
  PFX template __global__ void k_bucket_sums<P>(uint32_t*, const uint32_t*, const uint32_t*, uint32_t);          \
  PFX template __global__ void k_reduce2d_partial_acc<P>(uint32_t*, const uint32_t*, const uint32_t*, R2Geom, uint32_t); \
  PFX template __global__ void k_pairsum_x4<P>(uint32_t*, const uint32_t*, uint32_t);                            \
  PFX template __global__ void k_reduce_next<P>(uint32_t*, uint32_t*, const uint32_t*, const uint32_t*,          \
                                                const uint32_t*, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t); \
  PFX template __global__ void k_bucket_accumulate<P>(uint32_t*, const uint32_t*, const uint32_t*,               \
                                                      const uint32_t*, const uint32_t*, uint32_t, uint32_t, int);

#define MSMZ_INST_REDUCE(F, Fr, PFX)                                                                              \
  PFX template __global__ void k_reduce_first<F>(uint32_t*, uint32_t*, const uint32_t*, const uint32_t*,          \
                                                 const uint4*, uint32_t, uint32_t, uint32_t, uint32_t);           \
  PFX template __global__ void k_reduce2d_partial<F>(uint32_t*, const uint32_t*, const uint32_t*, const uint4*,   \
                                                     R2Geom, uint32_t);                                          \
  PFX template __global__ void k_reduce_affine_finish<F>(uint32_t*, uint32_t*, const uint32_t*, const uint32_t*,  \
                                                         const uint2*, const uint4*, F2Geom);                     \
  MSMZ_INST_POLICY(WeierPolicy<F>, PFX)

#define MSMZ_INST_REDUCE_TE(F, Fr, PFX) MSMZ_INST_POLICY(TePolicy<F>, PFX)

#define MSMZ_INST_TEST(F, Fr, P, TE, PFX)                                                                         \
  PFX template __global__ void k_test_field<F>(uint32_t*, const uint32_t*, const uint32_t*, uint32_t, int, uint32_t*); \
  PFX template __global__ void k_test_glv<Fr>(uint32_t*, uint32_t*, uint8_t*, const uint32_t*, uint32_t);         \
  PFX template __global__ void k_test_digits<Fr, false>(uint32_t*, const uint32_t*, uint32_t, int, int);          \
  PFX template __global__ void k_test_point<P, TE>(uint32_t*, const uint32_t*, const uint32_t*, const uint8_t*,   \
                                                   const uint8_t*, uint32_t, int);

// sort kernels: window size 0 = any, 16 / 17 = the defaults of large inputs (window loop unrolled)
#define MSMZ_INST_SORT(Fr, GLV, C, PFX)                                                                           \
  PFX template __global__ void k_hist<Fr, GLV, C>(uint32_t*, uint16_t*, uint32_t*, MsmMeta*, const uint32_t*, SortGeom, uint32_t); \
  PFX template __global__ void k_coarse<Fr, GLV, C>(uint32_t*, const uint32_t*, const uint32_t*, const uint16_t*, const uint32_t*, SortGeom, uint32_t);

#define MSMZ_INST_SCALAR(Fr, PFX)                                                                                 \
  PFX template __global__ void k_digits<Fr, false>(uint32_t*, uint32_t*, MsmMeta*, const uint32_t*, uint32_t, int, int, int); \
  MSMZ_INST_SORT(Fr, false, 0, PFX)                                                                               \
  MSMZ_INST_SORT(Fr, false, 16, PFX)                                                                              \
  MSMZ_INST_SORT(Fr, false, 17, PFX)                                                                              \
  PFX template __global__ void k_check_scalars<Fr>(uint32_t*, const uint32_t*, uint32_t);                         \
  PFX template __global__ void k_gen_scalars<Fr>(uint32_t*, uint32_t, uint64_t, GenMap);

#define MSMZ_INST_MISC(F, Fr, PFX)                                                                                \
  PFX template __global__ void k_points_to_mont<F>(uint32_t*, const uint32_t*, const uint8_t*, uint32_t, int, uint32_t*); \
  PFX template __global__ void k_points_from_mont<F>(uint32_t*, const uint32_t*, uint32_t);                       \
  PFX template __global__ void k_digits<Fr, true>(uint32_t*, uint32_t*, MsmMeta*, const uint32_t*, uint32_t, int, int, int); \
  MSMZ_INST_SORT(Fr, true, 0, PFX)                                                                                \
  MSMZ_INST_SORT(Fr, true, 16, PFX)                                                                               \
  PFX template __global__ void k_test_digits<Fr, true>(uint32_t*, const uint32_t*, uint32_t, int, int);           \
  MSMZ_INST_TEST(F, Fr, WeierPolicy<F>, false, PFX)                                                               \
  MSMZ_INST_SCALAR(Fr, PFX)

#define MSMZ_INST_MISC_TE(F, Fr, PFX)                                                              \
  PFX template __global__ void k_te_points_to_niels<F>(uint32_t*, const uint32_t*, uint32_t, uint32_t*); \
  PFX template __global__ void k_te_points_from_niels<F>(uint32_t*, const uint32_t*, uint32_t);    \
  MSMZ_INST_TEST(F, Fr, TePolicy<F>, true, PFX)                                                    \
  MSMZ_INST_SCALAR(Fr, PFX)

#define MSMZ_INST_GEN(F, Fr, PFX)                                                \
  PFX template __global__ void k_gen_table<F>(uint32_t*, const uint32_t*);       \
  PFX template __global__ void k_gen_points<F>(uint32_t*, const uint32_t*, uint32_t, uint64_t, int, GenMap);

#define MSMZ_INST_GEN_TE(F, Fr, PFX)                                             \
  PFX template __global__ void k_te_gen_table<F>(uint32_t*, const uint32_t*);    \
  PFX template __global__ void k_te_gen_points<F>(uint32_t*, const uint32_t*, uint32_t, uint64_t, GenMap);

#define MSMZ_EXTERN extern
#define MSMZ_DEFINE
