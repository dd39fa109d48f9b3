// Seeded synthetic inputs generated on the GPU -- the analogue of the reference's
// `randomPointsFast` / `randomScalars` (src/curve-random.ts:14-92, 151-194), which draw from an
// unseeded `crypto.getRandomValues`.  Here everything is a pure function of (seed, index):
//
//   point i  = a_i * G,  a_i = splitmix64(seed, i)  -- a sum of table entries  T_k[w_k],  w_k the
//              k-th 13-bit window of a_i and T_k[w] = w * 2^(13k) * G  (same windowed construction as
//              curve-random.ts:24-91, but over multiples of the generator so that a_i is known and
//              an MSM over any N has the closed form (sum s_i a_i) * G).
//   scalar i = first of the 32-byte little-endian draws  u(seed, i, attempt)  that is < q after
//              masking to the bit length of q  (curve-random.ts:151-190 rejection sampling).
#pragma once
#include <hip/hip_runtime.h>
#include "kernels.h"

namespace msmz {

constexpr int GEN_BITS = 13;
constexpr int GEN_TABLE = 1 << GEN_BITS;
constexpr int GEN_WINDOWS = 5;   // 5 * 13 = 65 >= 64 bits

MSMZ_HD uint64_t splitmix64(uint64_t seed, uint64_t index) {
  uint64_t z = seed + (index + 1) * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

template <class F>
__device__ __forceinline__ void xyzz_to_affine_mont(Affine<F>& a, const Xyzz<F>& p) {
  Fe<F> zi3, t, zi2;
  fe_inverse(zi3, p.ZZZ);
  fe_mul(t, zi3, p.ZZ);
  fe_sqr(zi2, t);
  fe_mul(a.x, p.X, zi2);
  fe_mul(a.y, p.Y, zi3);
}

// table[k * GEN_TABLE + w] = w * base_k (affine record; w = 0 -> infinity record)
template <class F>
__global__ void __launch_bounds__(128) k_gen_table(uint32_t* table, const uint32_t* bases) {
  constexpr int RW = 2 * F::NW;
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= GEN_WINDOWS * GEN_TABLE) return;
  uint32_t k = t / GEN_TABLE, w = t % GEN_TABLE;
  Affine<F> base;
  load_affine<F>(base, bases + (size_t)k * RW, 0);
  Xyzz<F> acc, tmp;
  xyzz_set_inf(acc);
  for (int bit = GEN_BITS - 1; bit >= 0; bit--) {
    xyzz_dbl(tmp, acc);
    acc = tmp;
    if ((w >> bit) & 1u) {
      xyzz_madd(tmp, acc, base, false);
      acc = tmp;
    }
  }
  Affine<F> a;
  bool inf = xyzz_is_inf(acc);
  if (!inf) xyzz_to_affine_mont(a, acc);
  store_affine<F>(table + (size_t)t * RW, a, inf);
}

template <class F>
__global__ void __launch_bounds__(128) k_gen_points(uint32_t* out, const uint32_t* table, uint32_t n, uint64_t seed,
                                                    int endo) {
  constexpr int RW = 2 * F::NW;
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t a = splitmix64(seed, i);
  Xyzz<F> acc, tmp;
  xyzz_set_inf(acc);
#pragma unroll 1
  for (int k = 0; k < GEN_WINDOWS; k++) {
    uint32_t w = (uint32_t)(a >> (GEN_BITS * k)) & (GEN_TABLE - 1);
    if (w == 0) continue;
    Affine<F> p;
    load_affine<F>(p, table + ((size_t)k * GEN_TABLE + w) * RW, 0);
    xyzz_madd(tmp, acc, p, false);
    acc = tmp;
  }
  Affine<F> r;
  bool inf = xyzz_is_inf(acc);
  if (!inf) xyzz_to_affine_mont(r, acc);
  store_affine<F>(out + (size_t)i * RW, r, inf);
  if (endo) {
    Fe<F> beta, bx;
    fe_set_const<F>(beta, F::BETA);
    if (!inf) {
      fe_mul(bx, r.x, beta);
      r.x = bx;
    }
    store_affine<F>(out + ((size_t)n + i) * RW, r, inf);
  }
}

template <class Fr>
__global__ void __launch_bounds__(256) k_gen_scalars(uint32_t* out, uint32_t n, uint64_t seed) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  constexpr int TOP_BITS = Fr::BITS - 224;            // bits kept in the top 32-bit word
  constexpr uint32_t TOP_MASK = TOP_BITS >= 32 ? 0xffffffffu : ((1u << TOP_BITS) - 1u);
  uint32_t w[8];
  for (uint32_t attempt = 0; attempt < 64; attempt++) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      uint64_t v = splitmix64(seed ^ 0x5ca1a75ull, ((uint64_t)i * 64 + attempt) * 4 + j);
      w[2 * j] = (uint32_t)v;
      w[2 * j + 1] = (uint32_t)(v >> 32);
    }
    w[7] &= TOP_MASK;
    if (!words_geq<8>(w, Fr::Q)) break;
    if (attempt == 63) {
#pragma unroll
      for (int j = 0; j < 8; j++) w[j] = 0;   // unreachable in practice (p ~ 2^-64)
    }
  }
  uint4* o = reinterpret_cast<uint4*>(out + (size_t)i * 8);
  o[0] = make_uint4(w[0], w[1], w[2], w[3]);
  o[1] = make_uint4(w[4], w[5], w[6], w[7]);
}

}  // namespace msmz
