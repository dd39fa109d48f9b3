// Prime-field arithmetic for the MSM kernels: Montgomery multiplication on signed, lazily
// reduced limbs.  Replaces the reference's run-time generated wasm field module
// (src/wasm/multiply-montgomery.ts:58-215, src/wasm/field-arithmetic.ts:32-166,
// src/wasm/inverse.ts:191-218, src/field-msm.ts:86-123) for gfx950.
//
// Representation
//   memory   : NW saturated 32-bit words, little endian ( = NW/2 64-bit limbs: 6x64 for the
//              377/381-bit fields, 4x64 for the 255-bit fields).  Values in memory are
//              Montgomery residues x*R mod p, *lazily* reduced: any value in [0, 3p).
//   registers: N signed limbs of W bits (radix 2^W, R = 2^(N*W)); a value is sum l[j]*2^(W*j),
//              limbs may be negative / wider than W bits between operations ("lazy").
//
// Why not saturated 64-bit-limb CIOS: on gfx950 v_mad_u64_u32, v_add_co_u32 and v_addc_co_u32
// all issue at the same half rate (profiles/r01_ubench_int_instr_rates.txt), so every carry
// instruction costs as much as a multiply.  With W = 28/29-bit limbs a whole column of products
// is summed in one 64-bit accumulator by back-to-back v_mad_i64_i32 with no carry chain --
// the same limb schedule idea as the reference's w=29 wasm code (multiply-montgomery.ts:47
// `nSafeSteps`), and the CIOS variant is kept in fp_cios.h for the A/B measurement.
//
// Bounds (checked in tests/test_fp_host.py, which compiles this file for the host):
//   mul/sqr inputs : |value| < 2^5 * p (2^3 * p for the 255-bit fields) and the limb magnitudes
//                    A, B of the two operands satisfy  N*A*B + N*2^(2W) < 2^63
//   mul/sqr output : value in (-1.5p, 0.5p), limbs 0..N-2 in [0, 2^W), top limb small signed.
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define MSMZ_HD __host__ __device__ __forceinline__
#else
#define MSMZ_HD inline
#endif

// Hide a limb's value range from the optimizer (device code only; no instruction is emitted).  Once LLVM has proved
// a limb non-negative (a masked product limb, also through a loop phi) it rewrites sext(limb) as zext(limb), and the
// AMDGPU backend then no longer recognises (int64)a * (int64)b as ONE v_mad_i64_i32: it emits two v_mad_u64_u32 plus
// two moves per product (seen in k_batch_add's backward pass: 832 + 728 extra instructions per addition).
#if defined(__HIP_DEVICE_COMPILE__)
#define MSMZ_OPAQUE_LIMB(x) asm("" : "+v"(x))
#define MSMZ_OPAQUE_ACC(x) asm("" : "+v"(x))
#else
#define MSMZ_OPAQUE_LIMB(x) (void)0
#define MSMZ_OPAQUE_ACC(x) (void)0
#endif
// MSMZ_FE_ILP = 1 (set per translation unit): a column of fe_mul / fe_sqr is summed in THREE independent chains that
// the compiler may not fuse back into one.  A back-to-back dependent v_mad_i64_i32 issues only every ~16 cycles, so a
// kernel that runs one or two waves per SIMD (the bucket reduction: long dependent point additions, 256+ VGPRs) is
// bound by that latency; kernels at four waves per SIMD (k_batch_add) hide it and keep the single chain, which has
// ~70 fewer instructions per product.
#ifndef MSMZ_FE_ILP
#define MSMZ_FE_ILP 0
#endif

namespace msmz {

template <class F>
struct Fe {
  int32_t l[F::N];
};

template <class F>
MSMZ_HD void fe_set_const(Fe<F>& r, const int32_t (&c)[F::N]) {
#pragma unroll
  for (int j = 0; j < F::N; j++) r.l[j] = c[j];
}

template <class F>
MSMZ_HD void fe_zero(Fe<F>& r) {
#pragma unroll
  for (int j = 0; j < F::N; j++) r.l[j] = 0;
}

// r = a + b, limb-wise, no carry (field-arithmetic.ts:32-58 `add`, without its reduction)
template <class F>
MSMZ_HD void fe_add(Fe<F>& r, const Fe<F>& a, const Fe<F>& b) {
#pragma unroll
  for (int j = 0; j < F::N; j++) r.l[j] = a.l[j] + b.l[j];
}

// r = a - b, limb-wise, no borrow (field-arithmetic.ts:60-110 `subtract`; sign lives in the limbs)
template <class F>
MSMZ_HD void fe_sub(Fe<F>& r, const Fe<F>& a, const Fe<F>& b) {
#pragma unroll
  for (int j = 0; j < F::N; j++) r.l[j] = a.l[j] - b.l[j];
}

template <class F>
MSMZ_HD void fe_neg(Fe<F>& r, const Fe<F>& a) {
#pragma unroll
  for (int j = 0; j < F::N; j++) r.l[j] = -a.l[j];
}

// r = neg ? -a : a  (mask form, no divergence)
template <class F>
MSMZ_HD void fe_cneg(Fe<F>& r, const Fe<F>& a, uint32_t neg) {
  int32_t m = -(int32_t)(neg & 1u);
#pragma unroll
  for (int j = 0; j < F::N; j++) r.l[j] = (a.l[j] ^ m) - m;
}

// One parallel carry pass: limbs 0..N-2 come back within [0, 2^W + small), value unchanged.
template <class F>
MSMZ_HD void fe_carry(Fe<F>& a) {
  constexpr int N = F::N, W = F::W;
  constexpr int32_t MASK = (1 << W) - 1;
  int32_t c[N];
#pragma unroll
  for (int j = 0; j < N - 1; j++) c[j] = a.l[j] >> W;
#pragma unroll
  for (int j = N - 2; j >= 1; j--) a.l[j] = (a.l[j] & MASK) + c[j - 1];
  a.l[0] &= MASK;
  a.l[N - 1] += c[N - 2];
}

// Full sequential carry: limbs 0..N-2 in [0, 2^W), top limb carries the sign.
template <class F>
MSMZ_HD void fe_normalize(Fe<F>& a) {
  constexpr int N = F::N, W = F::W;
  constexpr int32_t MASK = (1 << W) - 1;
  int32_t c = 0;
#pragma unroll
  for (int j = 0; j < N - 1; j++) {
    int32_t t = a.l[j] + c;
    a.l[j] = t & MASK;
    c = t >> W;
  }
  a.l[N - 1] += c;
}

// Montgomery product r = a*b/R (mod p), product-scanning with one signed 64-bit column
// accumulator (multiply-montgomery.ts:58-136, restated for signed lazy limbs).
// Column k:  acc += sum_{i+j=k} a_i*b_j  -  sum_{i+j=k, j>=1} m_i*p_j ;  for k < N the new
// quotient digit m_k = acc*p^-1 mod 2^W makes acc - m_k*p_0 divisible by 2^W.
template <class F>
MSMZ_HD void fe_mul(Fe<F>& r, const Fe<F>& a_in, const Fe<F>& b_in) {
  constexpr int N = F::N, W = F::W;
  constexpr uint32_t MASK = (1u << W) - 1;
  Fe<F> a = a_in, b = b_in;
#pragma unroll
  for (int j = 0; j < N; j++) {
    MSMZ_OPAQUE_LIMB(a.l[j]);
    MSMZ_OPAQUE_LIMB(b.l[j]);
  }
  int32_t m[N];
  int64_t acc = 0;
#pragma unroll
  for (int k = 0; k < 2 * N - 1; k++) {
    const int lo = k - (N - 1) > 0 ? k - (N - 1) : 0;
    const int hi = k < N - 1 ? k : N - 1;
    int64_t accp = 0;
#if MSMZ_FE_ILP
    int64_t acce = 0;
#pragma unroll
    for (int i = lo; i <= hi; i++) {
      if ((i - lo) & 1) acce += (int64_t)a.l[i] * (int64_t)b.l[k - i]; else acc += (int64_t)a.l[i] * (int64_t)b.l[k - i];
    }
#else
#pragma unroll
    for (int i = lo; i <= hi; i++) acc += (int64_t)a.l[i] * (int64_t)b.l[k - i];
#endif
#pragma unroll
    for (int i = lo; i <= hi; i++) {
      const int j = k - i;
      if (j >= 1 && F::PL[j] != 0) accp += (int64_t)m[i] * (int64_t)F::NPL[j];
    }
#if MSMZ_FE_ILP
    MSMZ_OPAQUE_ACC(acce);
    MSMZ_OPAQUE_ACC(accp);
    acc += acce;
#endif
    acc += accp;
    if (k < N) {
      uint32_t q = (uint32_t)acc;
      if (F::PINV != 1u) q *= F::PINV;
      m[k] = (int32_t)(q & MASK);
      if (F::PL[0] != 1) acc += (int64_t)m[k] * (int64_t)F::NPL[0];
      // with p_0 == 1 the low W bits of acc equal m_k and the floor shift drops them
      acc >>= W;
    } else {
      r.l[k - N] = (int32_t)((uint32_t)acc & MASK);
      acc >>= W;
    }
  }
  r.l[N - 1] = (int32_t)acc;
}

// Montgomery square (multiply-montgomery.ts:138-215): off-diagonal products once, doubled.
// The doubling is applied to the 64-bit column sum of the off-diagonal products, not to an operand: with a doubled
// 32-bit operand (a2 = 2 a) the compiler widens it to 64 bits (sext(2a) -> shl(sext a)) and every product becomes
// two v_mad_u64_u32 plus two moves instead of one v_mad_i64_i32 (seen in the ISA of k_batch_add: +1100 instructions
// per addition).
template <class F>
MSMZ_HD void fe_sqr(Fe<F>& r, const Fe<F>& a_in) {
  constexpr int N = F::N, W = F::W;
  constexpr uint32_t MASK = (1u << W) - 1;
  Fe<F> a = a_in;
#pragma unroll
  for (int j = 0; j < N; j++) MSMZ_OPAQUE_LIMB(a.l[j]);
  int32_t m[N];
  int64_t acc = 0;
#pragma unroll
  for (int k = 0; k < 2 * N - 1; k++) {
    const int lo = k - (N - 1) > 0 ? k - (N - 1) : 0;
    const int hi = k < N - 1 ? k : N - 1;
    int64_t off = 0;
#pragma unroll
    for (int i = lo; i <= hi; i++) {
      const int j = k - i;
      if (i < j) off += (int64_t)a.l[i] * (int64_t)a.l[j];
    }
    int64_t accp = 0;
    if ((k & 1) == 0) accp = (int64_t)a.l[k / 2] * (int64_t)a.l[k / 2];
#pragma unroll
    for (int i = lo; i <= hi; i++) {
      const int j = k - i;
      if (j >= 1 && F::PL[j] != 0) accp += (int64_t)m[i] * (int64_t)F::NPL[j];
    }
#if MSMZ_FE_ILP
    MSMZ_OPAQUE_ACC(off);
    MSMZ_OPAQUE_ACC(accp);
#endif
    acc += off * 2 + accp;
    if (k < N) {
      uint32_t q = (uint32_t)acc;
      if (F::PINV != 1u) q *= F::PINV;
      m[k] = (int32_t)(q & MASK);
      if (F::PL[0] != 1) acc += (int64_t)m[k] * (int64_t)F::NPL[0];
      acc >>= W;
    } else {
      r.l[k - N] = (int32_t)((uint32_t)acc & MASK);
      acc >>= W;
    }
  }
  r.l[N - 1] = (int32_t)acc;
}

// ---------------------------------------------------------------------------------- memory format
// words (saturated, value in [0, 2^(32*NW))) -> normalized limbs.  (field-helpers.ts:211-266
// `fromPackedBytes` does the same re-slicing from bytes to w-bit limbs.)
template <class F>
MSMZ_HD void fe_unpack(Fe<F>& r, const uint32_t* w) {
  constexpr int N = F::N, W = F::W, NW = F::NW;
  constexpr uint32_t MASK = (1u << W) - 1;
#pragma unroll
  for (int j = 0; j < N; j++) {
    const int bit = W * j;
    const int w0 = bit >> 5, sh = bit & 31;
    uint32_t v = w[w0] >> sh;
    if (sh + W > 32 && w0 + 1 < NW) v |= w[w0 + 1] << (32 - sh);
    r.l[j] = (int32_t)(v & MASK);
  }
}

// fully normalized, non-negative limbs (value < 2^(32*NW)) -> words
template <class F>
MSMZ_HD void fe_pack(uint32_t* w, const Fe<F>& a) {
  constexpr int N = F::N, W = F::W, NW = F::NW;
#pragma unroll
  for (int i = 0; i < NW; i++) {
    const int bit = 32 * i;
    const int j0 = bit / W, o = bit - W * j0;
    uint32_t v = (uint32_t)a.l[j0] >> o;
    if (j0 + 1 < N) v |= (uint32_t)a.l[j0 + 1] << (W - o);
    if (2 * W - o < 32 && j0 + 2 < N) v |= (uint32_t)a.l[j0 + 2] << (2 * W - o);
    w[i] = v;
  }
}

// Bring a lazy value |v| < 2^4 * p into [0, 3p) with normalized limbs: estimate the quotient
// from the two top limbs, subtract q*p, one sequential carry.  (The reference instead keeps
// everything in [0, 2p) with a conditional subtraction per operation: field-arithmetic.ts:112-166.)
template <class F>
MSMZ_HD void fe_reduce_small(Fe<F>& a) {
  constexpr int N = F::N, W = F::W;
  constexpr int64_t MASK = ((int64_t)1 << W) - 1;
  // est ~ floor(v / 2^(W*(N-1))) within +-1
  int32_t est = a.l[N - 1] + (a.l[N - 2] >> W);
  // q ~ v/p - 1 (+-0.5): float divide of the top bits by the top bits of p (|est >> SH| < 2^24)
  constexpr int SH = (F::PL[N - 1] >= (1 << 16)) ? 8 : 0;
  float x = (float)(est >> SH) * (1.0f / (float)((F::PL[N - 1] >> SH) + 1));
  int32_t q = (int32_t)__builtin_floorf(x) - 1;
  // a -= q*p fused with the sequential carry (64-bit temporaries: q*p_j may exceed 32 bits)
  int64_t c = 0;
#pragma unroll
  for (int j = 0; j < N - 1; j++) {
    int64_t t = (int64_t)a.l[j] + c + (int64_t)q * (int64_t)F::NPL[j];
    a.l[j] = (int32_t)(t & MASK);
    c = t >> W;
  }
  a.l[N - 1] = (int32_t)((int64_t)a.l[N - 1] + c + (int64_t)q * (int64_t)F::NPL[N - 1]);
}

// store a lazy value (|v| < 2^4 p) as a memory-format residue in [0, 3p)
template <class F>
MSMZ_HD void fe_store(uint32_t* w, const Fe<F>& a) {
  Fe<F> t = a;
  fe_reduce_small(t);
  fe_pack(w, t);
}

// store a direct mul/sqr output (value in (-1.5p, 0.5p)): add 2p, carry, pack
template <class F>
MSMZ_HD void fe_store_mulout(uint32_t* w, const Fe<F>& a) {
  Fe<F> t;
#pragma unroll
  for (int j = 0; j < F::N; j++) t.l[j] = a.l[j] + F::P2[j];
  fe_normalize(t);
  fe_pack(w, t);
}

// ---------------------------------------------------------------------------------- canonical form
// saturated-word helpers (used only off the hot path: inversion input, final outputs, equality)
template <int NW>
MSMZ_HD uint32_t words_sub(uint32_t* r, const uint32_t* a, const uint32_t* b) {  // returns borrow
  uint64_t br = 0;
#pragma unroll
  for (int i = 0; i < NW; i++) {
    uint64_t t = (uint64_t)a[i] - (uint64_t)b[i] - br;
    r[i] = (uint32_t)t;
    br = (t >> 32) & 1u;
  }
  return (uint32_t)br;
}

template <int NW>
MSMZ_HD uint32_t words_add(uint32_t* r, const uint32_t* a, const uint32_t* b) {  // returns carry
  uint64_t c = 0;
#pragma unroll
  for (int i = 0; i < NW; i++) {
    uint64_t t = (uint64_t)a[i] + (uint64_t)b[i] + c;
    r[i] = (uint32_t)t;
    c = t >> 32;
  }
  return (uint32_t)c;
}

template <int NW>
MSMZ_HD bool words_is_zero(const uint32_t* a) {
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < NW; i++) o |= a[i];
  return o == 0;
}

template <int NW>
MSMZ_HD bool words_eq(const uint32_t* a, const uint32_t* b) {
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < NW; i++) o |= a[i] ^ b[i];
  return o == 0;
}

template <int NW>
MSMZ_HD bool words_geq(const uint32_t* a, const uint32_t* b) {
  uint32_t t[NW];
  return words_sub<NW>(t, a, b) == 0;
}

// words (any value in [0, 4p)) -> canonical [0, p)   (field-arithmetic.ts:112-135 `reduce`)
template <class F>
MSMZ_HD void words_canon(uint32_t* w) {
  constexpr int NW = F::NW;
#pragma unroll 1
  for (int it = 0; it < 3; it++) {
    uint32_t t[NW];
    uint32_t br = words_sub<NW>(t, w, F::PW);
    if (!br) {
#pragma unroll
      for (int i = 0; i < NW; i++) w[i] = t[i];
    }
  }
}

// lazy register value -> canonical words in [0, p)
template <class F>
MSMZ_HD void fe_to_canon_words(uint32_t* w, const Fe<F>& a) {
  fe_store<F>(w, a);
  words_canon<F>(w);
}

template <class F>
MSMZ_HD bool fe_is_zero_mod_p(const Fe<F>& a) {
  uint32_t w[F::NW];
  fe_to_canon_words<F>(w, a);
  return words_is_zero<F::NW>(w);
}

// ---------------------------------------------------------------------------------- inversion
// Modular inversion by a limb-aligned "optimized binary GCD" (Pornin 2020, the same family as the
// reference's experimental src/inverse/faster-inverse.ts; the job of its Kaliski almost-inverse,
// inverse.ts:42-129 + 191-218).  Invariants  a = u*y, b = v*y (mod p)  with a = y, b = p, u = 1, v = 0.
// One outer iteration runs W binary-GCD steps on (low W bits | top ~29 bits) approximations of a and b,
// collecting them in a 2x2 matrix (f0 g0; f1 g1) with |f|+|g| <= 2^W, then applies the matrix:
//   (a, b) <- (a f0 + b g0, a f1 + b g1) / 2^W              exact: one limb shift
//   (u, v) <- (u f0 + v g0, u f1 + v g1) / 2^W  (mod p)     one Montgomery column step
// so W bits are retired per iteration with ~6 multiply-adds per limb instead of W long shifts/subtracts.
// A fixed iteration count (bounded by 2*BITS total steps) makes the loop branch-free at wave level.
// Input and output are *Montgomery* residues: given a*R it returns a^-1 * R.  Returns false (r = 0)
// when the input is 0 mod p.
template <class F>
MSMZ_HD void gcd_apply_exact(Fe<F>& ra, Fe<F>& rb, const Fe<F>& a, const Fe<F>& b, int32_t f0, int32_t g0, int32_t f1,
                             int32_t g1) {
  constexpr int N = F::N, W = F::W;
  constexpr int64_t MASK = ((int64_t)1 << W) - 1;
  int64_t ca = 0, cb = 0;
#pragma unroll
  for (int j = 0; j < N; j++) {
    int64_t ta = (int64_t)a.l[j] * f0 + (int64_t)b.l[j] * g0 + ca;
    int64_t tb = (int64_t)a.l[j] * f1 + (int64_t)b.l[j] * g1 + cb;
    if (j > 0) {
      ra.l[j - 1] = (int32_t)(ta & MASK);
      rb.l[j - 1] = (int32_t)(tb & MASK);
    }
    ca = ta >> W;   // j == 0: the low limb is 0 by construction
    cb = tb >> W;
  }
  ra.l[N - 1] = (int32_t)ca;
  rb.l[N - 1] = (int32_t)cb;
}

// r = (u f + v g) / 2^W mod p, result value in (-p, p), limbs normalized with a signed top limb
template <class F>
MSMZ_HD void gcd_apply_mod(Fe<F>& r, const Fe<F>& u, const Fe<F>& v, int32_t f, int32_t g) {
  constexpr int N = F::N, W = F::W;
  constexpr int64_t MASK = ((int64_t)1 << W) - 1;
  int64_t t0 = (int64_t)u.l[0] * f + (int64_t)v.l[0] * g;
  uint32_t q32 = (uint32_t)t0;
  if (F::PINV != 1u) q32 *= F::PINV;
  const int32_t q = (int32_t)(q32 & (uint32_t)MASK);   // t0 - q*p_0 = 0 mod 2^W
  int64_t c = 0;
#pragma unroll
  for (int j = 0; j < N; j++) {
    int64_t t = (int64_t)u.l[j] * f + (int64_t)v.l[j] * g + (int64_t)q * F::NPL[j] + c;
    if (j > 0) r.l[j - 1] = (int32_t)(t & MASK);
    c = t >> W;
  }
  r.l[N - 1] = (int32_t)c;
  // value in (-2p, p): add p when negative
  const int32_t neg = r.l[N - 1] >> 31;
#pragma unroll
  for (int j = 0; j < N; j++) r.l[j] += F::PL[j] & neg;
}

template <class F>
MSMZ_HD void fe_neg_normalized(Fe<F>& a) {   // a <- -a, limbs back to normalized form
  constexpr int N = F::N, W = F::W;
  constexpr int32_t MASK = (1 << W) - 1;
  int32_t c = 0;
#pragma unroll
  for (int j = 0; j < N - 1; j++) {
    int32_t t = c - a.l[j];
    a.l[j] = t & MASK;
    c = t >> W;
  }
  a.l[N - 1] = c - a.l[N - 1];
}

template <class F>
MSMZ_HD bool fe_inverse(Fe<F>& r, const Fe<F>& x) {
  constexpr int N = F::N, W = F::W;
  constexpr uint64_t LOWMASK = ((uint64_t)1 << W) - 1;
  Fe<F> a, b, u, v;
  {
    // canonical value of x as normalized limbs
    uint32_t w[F::NW];
    fe_to_canon_words<F>(w, x);
    fe_unpack<F>(a, w);
  }
#pragma unroll
  for (int j = 0; j < N; j++) {
    b.l[j] = F::PL[j];
    u.l[j] = (j == 0) ? 1 : 0;
    v.l[j] = 0;
  }
  constexpr int ITERS = (2 * F::BITS + W - 1) / W + 1;
#pragma unroll 1
  for (int it = 0; it < ITERS; it++) {
    // a == 0: converged (further iterations would be the identity: f0 = 1, g1 = 2^W)
    int32_t anyA = 0;
#pragma unroll
    for (int j = 0; j < N; j++) anyA |= a.l[j];
    if (anyA == 0) break;
    // ---- approximations: low W bits + the top bits at a common alignment
    int h = 0;   // highest limb where a or b is non-zero
#pragma unroll
    for (int j = 1; j < N; j++)
      if ((a.l[j] | b.l[j]) != 0) h = j;
    uint64_t ah = 0, bh = 0;
#pragma unroll
    for (int j = 1; j < N; j++) {
      if (j == h) {
        ah = ((uint64_t)(uint32_t)a.l[j] << W) | (uint32_t)a.l[j - 1];
        bh = ((uint64_t)(uint32_t)b.l[j] << W) | (uint32_t)b.l[j - 1];
      }
    }
    uint64_t xa, xb;
    if (h <= 1) {
      // both fit in 2W bits: exact values
      xa = h == 0 ? (uint64_t)(uint32_t)a.l[0] : ah;
      xb = h == 0 ? (uint64_t)(uint32_t)b.l[0] : bh;
    } else {
      const uint64_t mx = ah | bh;
      const int len = 64 - __builtin_clzll(mx | 1);      // <= 2W
      const int sh = len > (W + 1) ? len - (W + 1) : 0;   // keep the top W+1 bits
      xa = ((ah >> sh) << W) | ((uint64_t)(uint32_t)a.l[0] & LOWMASK);
      xb = ((bh >> sh) << W) | ((uint64_t)(uint32_t)b.l[0] & LOWMASK);
    }
    // ---- W binary steps on the approximations
    int32_t f0 = 1, g0 = 0, f1 = 0, g1 = 1;
#pragma unroll 1
    for (int s = 0; s < W; s++) {
      const bool odd = (xa & 1) != 0;
      const bool swap = odd && (xa < xb);
      // conditional swap of (xa, f0, g0) with (xb, f1, g1)
      const uint64_t ta = swap ? xb : xa, tb = swap ? xa : xb;
      const int32_t tf0 = swap ? f1 : f0, tg0 = swap ? g1 : g0;
      const int32_t tf1 = swap ? f0 : f1, tg1 = swap ? g0 : g1;
      xa = odd ? ta - tb : ta;
      f0 = odd ? tf0 - tf1 : tf0;
      g0 = odd ? tg0 - tg1 : tg0;
      xb = tb;
      xa >>= 1;
      f1 = tf1 * 2;
      g1 = tg1 * 2;
    }
    // ---- apply the matrix
    Fe<F> na, nb, nu, nv;
    gcd_apply_exact<F>(na, nb, a, b, f0, g0, f1, g1);
    gcd_apply_mod<F>(nu, u, v, f0, g0);
    gcd_apply_mod<F>(nv, u, v, f1, g1);
    if (na.l[N - 1] < 0) {
      fe_neg_normalized<F>(na);
      fe_neg<F>(nu, nu);
    }
    if (nb.l[N - 1] < 0) {
      fe_neg_normalized<F>(nb);
      fe_neg<F>(nv, nv);
    }
    a = na;
    b = nb;
    u = nu;
    v = nv;
  }
  // gcd ends up in b
  int32_t rest = b.l[0] ^ 1;
#pragma unroll
  for (int j = 1; j < N; j++) rest |= b.l[j];
  if (rest != 0) {
    fe_zero(r);
    return false;
  }
  Fe<F> r3;
  fe_carry(v);
  fe_set_const<F>(r3, F::R3);
  fe_mul<F>(r, v, r3);   // (x)^-1 = a^-1 R^-1  ->  * R^3 / R = a^-1 R
  return true;
}

// to / from Montgomery form (field-msm.ts:183-185 `toMontgomery` = multiply by R^2)
template <class F>
MSMZ_HD void fe_to_mont(Fe<F>& r, const Fe<F>& a) {
  Fe<F> r2;
  fe_set_const<F>(r2, F::R2);
  fe_mul<F>(r, a, r2);
}

template <class F>
MSMZ_HD void fe_from_mont(Fe<F>& r, const Fe<F>& a) {
  Fe<F> one;
  fe_zero(one);
  one.l[0] = 1;
  fe_mul<F>(r, a, one);
}

}  // namespace msmz
