// Scalar-side arithmetic: GLV decomposition and signed-digit window slicing.
//
//   glv_decompose  -- restates `decompose` of the reference's scalar wasm module
//                     (src/wasm/glv.ts:68-169; bigint formula in src/glv/glv-test.ts:96-100):
//                         x_j = round(m_j * s / 2^256),  s0 = s + v00 x0 + v01 x1,  s1 = v10 x0 + v11 x1
//                     with the lattice basis from the truncated EGCD (src/glv/glv.ts:21-50), baked at
//                     build time by tools/gen_constants.py.  The reference truncates s to its top limbs
//                     before the multiply; here the whole 256-bit s is used (the cost is irrelevant on
//                     the GPU), so |s0|, |s1| < 2^127 -- any valid decomposition gives the same MSM.
//   signed_digits  -- msm-batched-affine.ts:180-199 / msm-basic.ts:80-93: digit l in [0, L], L = 2^(c-1),
//                     stored as  l | (negate << 31).
#pragma once
#include <cstdint>
#include "fp.h"

namespace msmz {

// r[0..na+nb) = a[0..na) * b[0..nb)   (schoolbook on saturated words; off the hot path)
template <int NA, int NB>
MSMZ_HD void words_mul(uint32_t* r, const uint32_t* a, const uint32_t* b) {
#pragma unroll
  for (int i = 0; i < NA + NB; i++) r[i] = 0;
#pragma unroll
  for (int i = 0; i < NA; i++) {
    uint64_t c = 0;
#pragma unroll
    for (int j = 0; j < NB; j++) {
      uint64_t t = (uint64_t)a[i] * b[j] + r[i + j] + c;
      r[i + j] = (uint32_t)t;
      c = t >> 32;
    }
    r[i + NB] = (uint32_t)c;
  }
}

// acc (8 words, mod 2^256) +/-= a (4 words) * b (4 words)
MSMZ_HD void acc256_muladd(uint32_t* acc, const uint32_t* a, const uint32_t* b, uint32_t negate) {
  uint32_t prod[8];
  words_mul<4, 4>(prod, a, b);
  if (negate) {
    words_sub<8>(acc, acc, prod);
  } else {
    words_add<8>(acc, acc, prod);
  }
}

// s (8 words, < q) -> |s0|, |s1| (4 words each) and sign flags; s = (+-)|s0| + (+-)|s1| * lambda (mod q)
template <class Fr>
MSMZ_HD void glv_decompose(uint32_t* s0, uint32_t* s1, uint32_t& neg0, uint32_t& neg1, const uint32_t* s) {
  // x_j = round(M_j * s / 2^256): 5 x 8 words -> 13 words, keep words 8..12, round on bit 255
  uint32_t t[13], x0[5], x1[5];
  words_mul<5, 8>(t, Fr::GLV_M0, s);
  {
    uint64_t c = t[7] >> 31;
#pragma unroll
    for (int i = 0; i < 5; i++) {
      uint64_t v = (uint64_t)t[8 + i] + c;
      x0[i] = (uint32_t)v;
      c = v >> 32;
    }
  }
  words_mul<5, 8>(t, Fr::GLV_M1, s);
  {
    uint64_t c = t[7] >> 31;
#pragma unroll
    for (int i = 0; i < 5; i++) {
      uint64_t v = (uint64_t)t[8 + i] + c;
      x1[i] = (uint32_t)v;
      c = v >> 32;
    }
  }
  // |x_j| < 2^128 (checked by the generator's bound test), so 4 words suffice
  uint32_t a0[8], a1[8];
#pragma unroll
  for (int i = 0; i < 8; i++) {
    a0[i] = s[i];
    a1[i] = 0;
  }
  acc256_muladd(a0, Fr::GLV_V00, x0, Fr::GLV_NEG[0]);
  acc256_muladd(a0, Fr::GLV_V01, x1, Fr::GLV_NEG[1]);
  acc256_muladd(a1, Fr::GLV_V10, x0, Fr::GLV_NEG[2]);
  acc256_muladd(a1, Fr::GLV_V11, x1, Fr::GLV_NEG[3]);
  // two's complement -> sign + magnitude
  uint32_t zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  neg0 = a0[7] >> 31;
  neg1 = a1[7] >> 31;
  if (neg0) words_sub<8>(a0, zero, a0);
  if (neg1) words_sub<8>(a1, zero, a1);
#pragma unroll
  for (int i = 0; i < 4; i++) {
    s0[i] = a0[i];
    s1[i] = a1[i];
  }
}

// c-bit window starting at bit `pos` of an NWORDS-word little-endian integer (bits past the end are 0)
template <int NWORDS>
MSMZ_HD uint32_t extract_bits(const uint32_t* w, int pos, int c) {
  int wi = pos >> 5, sh = pos & 31;
  uint64_t lo = (wi < NWORDS) ? w[wi] : 0u;
  uint64_t hi = (wi + 1 < NWORDS) ? w[wi + 1] : 0u;
  uint64_t v = (lo | (hi << 32)) >> sh;
  return (uint32_t)(v & ((1u << c) - 1u));
}

}  // namespace msmz
