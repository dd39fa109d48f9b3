// Stage-level test hooks: kernels that run ONE device routine on caller-supplied inputs and return the raw
// outputs, so that tests/ can check every stage in isolation -- the build's version of the
// reference's per-operation checks (src/field.test.ts:159-211, src/curve-projective.test.ts:77-209,
// src/glv/glv-test.ts:83-125, src/testing/equivalent-wasm.ts:97-147).  Exposed through include/msmz_test.h; they
// never take part in an MSM.
#pragma once
#include <hip/hip_runtime.h>
#include "kernels.h"

namespace msmz {

enum {   // field ops (msmz_test_field)
  TF_MUL = 0, TF_SQR = 1, TF_ADD = 2, TF_SUB = 3, TF_INVERSE = 4, TF_INVERSE_WAVE = 5, TF_ROUNDTRIP = 6,
  TF_IS_ZERO = 7, TF_SLOT_ROUNDTRIP = 8
};
enum {   // point ops (msmz_test_point)
  TP_ADD = 0, TP_ADD_X4 = 1, TP_MADD = 2, TP_DBL = 3, TP_DBL_X4 = 4
};

// Operands / results are NW memory words per element (little endian).  Inputs are lazy Montgomery residues: any
// value in [0, 4p).  Results are the CANONICAL representative of the routine's output residue:
//   MUL a*b/R, SQR a*a/R, ADD, SUB, INVERSE(_WAVE)  R^2/a  (0 for a = 0 mod p), ROUNDTRIP a (store -> load),
//   IS_ZERO  1 / 0 in word 0, SLOT_ROUNDTRIP a through the slot-record format of the tree rounds.
template <class F>
__global__ void __launch_bounds__(64) k_test_field(uint32_t* out, const uint32_t* a_in, const uint32_t* b_in,
                                                   uint32_t n, int op, uint32_t* scratch) {
  constexpr int NW = F::NW;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  // whole waves stay alive: fe_inverse_wave spreads one value over the lanes of a wave
  const uint32_t ii = i < n ? i : n - 1;
  Fe<F> a, b, r;
  fe_unpack<F>(a, a_in + (size_t)ii * NW);
  fe_unpack<F>(b, b_in + (size_t)ii * NW);
  fe_zero(r);
  switch (op) {
    case TF_MUL: fe_mul(r, a, b); break;
    case TF_SQR: fe_sqr(r, a); break;
    case TF_ADD: fe_add(r, a, b); break;
    case TF_SUB: fe_sub(r, a, b); break;
    case TF_INVERSE: fe_inverse(r, a); break;
    case TF_INVERSE_WAVE: {
      // the routine inverts ONE wave-uniform value: feed it lane 0's element, then lane 1's, ...
      for (int l = 0; l < 64; l++) {
        Fe<F> x, y;
#pragma unroll
        for (int j = 0; j < F::N; j++) x.l[j] = __shfl(a.l[j], l, 64);
        fe_inverse_wave(y, x);
        if ((int)(threadIdx.x & 63) == l) r = y;
      }
      break;
    }
    case TF_ROUNDTRIP: {
      uint32_t w[NW];
      fe_store<F>(w, a);
      fe_unpack<F>(r, w);
      break;
    }
    case TF_IS_ZERO: {
      Fe<F> d;
      fe_sub(d, a, b);
      r.l[0] = fe_is_zero(d) ? 1 : 0;
      break;
    }
    case TF_SLOT_ROUNDTRIP: {
      // record i of a chunk-interleaved slot array: (a, b) stored as a point, a also as a parked product
      Affine<F> p, q;
      p.x = a;
      p.y = b;
      slot_store_point<F>(scratch + slot_offset<F>(ii), p, false);
      slot_load_point<F, false>(q, scratch + slot_offset<F>(ii));
      fe_add(r, q.x, q.y);   // a + b
      break;
    }
    default: break;
  }
  if (i >= n) return;
  uint32_t w[NW];
  if (op == TF_IS_ZERO) {
#pragma unroll
    for (int j = 0; j < NW; j++) w[j] = j == 0 ? (uint32_t)r.l[0] : 0u;
  } else {
    fe_to_canon_words<F>(w, r);
  }
#pragma unroll
  for (int j = 0; j < NW; j++) out[(size_t)i * NW + j] = w[j];
}

// GLV decomposition of n scalars: out0/out1 = |s0|, |s1| (4 words each), neg[2i], neg[2i+1] = their signs
template <class Fr>
__global__ void __launch_bounds__(256) k_test_glv(uint32_t* s0, uint32_t* s1, uint8_t* neg, const uint32_t* scalars,
                                                  uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if constexpr (Fr::HAS_GLV) {
    uint32_t s[8], h0[4], h1[4], n0, n1;
#pragma unroll
    for (int j = 0; j < 8; j++) s[j] = scalars[(size_t)i * 8 + j];
    glv_decompose<Fr>(h0, h1, n0, n1, s);
#pragma unroll
    for (int j = 0; j < 4; j++) {
      s0[(size_t)i * 4 + j] = h0[j];
      s1[(size_t)i * 4 + j] = h1[j];
    }
    neg[2 * i] = (uint8_t)n0;
    neg[2 * i + 1] = (uint8_t)n1;
  }
}

// signed digits of n scalars as the sort kernels slice them: digits[(h * n + i) * K + k] = l | (negate << 31)
template <class Fr, bool GLV>
__global__ void __launch_bounds__(256) k_test_digits(uint32_t* digits, const uint32_t* scalars, uint32_t n, int c, int K) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  constexpr int HALVES = GLV ? 2 : 1;
  const uint32_t L = 1u << (c - 1);
  DigitStream<Fr, GLV> ds;
  ds.load(scalars, i);
  for (int k = 0; k < K; k++) {
#pragma unroll
    for (int h = 0; h < HALVES; h++) {
      uint32_t ng;
      const uint32_t l = ds.next(h, k, c, L, ng);
      digits[((size_t)h * n + i) * K + k] = l | (ng << 31);
    }
  }
}

// point operations on pairs of canonical-affine inputs converted to the accumulator type of policy P:
//   out[i] = canonical affine (x | y) of op(a_i, b_i); all-zero = infinity (Weierstrass)
template <class P, bool TE>
__global__ void __launch_bounds__(64) k_test_point(uint32_t* out, const uint32_t* a_in, const uint32_t* b_in,
                                                   const uint8_t* a_inf, const uint8_t* b_inf, uint32_t n, int op) {
  using F = typename P::F;
  using Acc = typename P::Acc;
  constexpr int NW = F::NW;
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const bool x4 = op == TP_ADD_X4 || op == TP_DBL_X4;
  const uint32_t i = x4 ? t >> 2 : t;   // a DPP quad per pair for the 4-lane addition
  const uint32_t ii = i < n ? i : n - 1;
  auto load = [&](Acc& p, const uint32_t* in, const uint8_t* inf) {
    Fe<F> x, y, xm, ym;
    fe_unpack<F>(x, in + (size_t)ii * 2 * NW);
    fe_unpack<F>(y, in + (size_t)ii * 2 * NW + NW);
    fe_to_mont(xm, x);
    fe_to_mont(ym, y);
    if constexpr (TE) {
      p.X = xm;
      p.Y = ym;
      fe_set_const<F>(p.Z, F::ONE);
      fe_mul(p.T, xm, ym);
    } else {
      if (inf != nullptr && inf[ii]) {
        xyzz_set_inf(p);
      } else {
        Affine<F> a;
        a.x = xm;
        a.y = ym;
        xyzz_from_affine(p, a);
      }
    }
  };
  Acc a, b, r;
  load(a, a_in, a_inf);
  load(b, b_in, b_inf);
  switch (op) {
    case TP_ADD: P::add(r, a, b); break;
    case TP_ADD_X4: P::add_x4(r, a, b, (int)(threadIdx.x & 3), false); break;
    case TP_DBL_X4: {   // 2 (a + b): the doubling on a general accumulator (ZZ != 1)
      Acc sum;
      P::add_x4(sum, a, b, (int)(threadIdx.x & 3), false);
      P::add_x4(r, sum, sum, (int)(threadIdx.x & 3), true);
      break;
    }
    case TP_DBL: P::dbl(r, a); break;
    default: P::add(r, a, b); break;
  }
  if (i >= n || (x4 && (threadIdx.x & 3) != 0)) return;
  uint32_t w[2 * NW];
  if constexpr (TE) {
    te_to_affine_canon<F>(w, r);
  } else {
    (void)xyzz_to_affine_canon<F>(w, r);
  }
#pragma unroll
  for (int j = 0; j < 2 * NW; j++) out[(size_t)i * 2 * NW + j] = w[j];
}

}  // namespace msmz
