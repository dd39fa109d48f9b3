// Explicit-instantiation lists: each kernel family is compiled in its own translation unit
// (kern_*.hip) so the build parallelizes; msmz.hip only sees `extern template` declarations.
#pragma once
#include "gen_kernels.h"
#include "kernels.h"

#define MSMZ_WEIERSTRASS_FIELDS(X) X(Bls377Fp, Bls377Fr)

#define MSMZ_BATCH_T 256

// (T, OCC = min waves per SIMD, BMAX = max pairs per thread) variants of the batch-add kernel
#define MSMZ_BATCH_VARIANTS(Y, F, PFX) \
  Y(F, 256, 2, 16, PFX) Y(F, 256, 3, 16, PFX) Y(F, 256, 4, 16, PFX) Y(F, 512, 2, 8, PFX) Y(F, 512, 4, 8, PFX)

#define MSMZ_INST_BATCH_ONE(F, T, OCC, BMAX, PFX)                                                                     \
  PFX template __global__ void k_batch_add<F, T, true, OCC, BMAX>(uint32_t*, const uint32_t*, const uint32_t*,        \
                                                                  const uint32_t*, const uint32_t*, uint32_t, int, int, \
                                                                  MsmMeta*);                                          \
  PFX template __global__ void k_batch_add<F, T, false, OCC, BMAX>(uint32_t*, const uint32_t*, const uint32_t*,       \
                                                                   const uint32_t*, const uint32_t*, uint32_t, int,    \
                                                                   int, MsmMeta*);

#define MSMZ_INST_BATCH(F, Fr, PFX) MSMZ_BATCH_VARIANTS(MSMZ_INST_BATCH_ONE, F, PFX)

#define MSMZ_INST_REDUCE(F, Fr, PFX)                                                                                   \
  PFX template __global__ void k_reduce_first<F>(uint32_t*, uint32_t*, const uint32_t*, const uint32_t*,               \
                                                 const uint32_t*, const uint32_t*, uint32_t, uint32_t, uint32_t,        \
                                                 uint32_t);                                                            \
  PFX template __global__ void k_reduce_next<F>(uint32_t*, uint32_t*, const uint32_t*, const uint32_t*, uint32_t,      \
                                                uint32_t, uint32_t, uint32_t, int);

#define MSMZ_INST_MISC(F, Fr, PFX)                                                                                     \
  PFX template __global__ void k_points_to_mont<F>(uint32_t*, const uint32_t*, const uint8_t*, uint32_t, int);         \
  PFX template __global__ void k_points_from_mont<F>(uint32_t*, const uint32_t*, uint32_t);                            \
  PFX template __global__ void k_digits<Fr, true>(uint32_t*, uint32_t*, const uint32_t*, uint32_t, int, int, int);          \
  PFX template __global__ void k_digits<Fr, false>(uint32_t*, uint32_t*, const uint32_t*, uint32_t, int, int, int);         \
  PFX template __global__ void k_gen_scalars<Fr>(uint32_t*, uint32_t, uint64_t);

#define MSMZ_INST_GEN(F, Fr, PFX)                                                                \
  PFX template __global__ void k_gen_table<F>(uint32_t*, const uint32_t*);                       \
  PFX template __global__ void k_gen_points<F>(uint32_t*, const uint32_t*, uint32_t, uint64_t, int);

#define MSMZ_EXTERN extern
#define MSMZ_DEFINE
