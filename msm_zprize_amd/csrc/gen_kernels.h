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
#include "multi.h"

namespace msmz {

constexpr int GEN_BITS = 13;
constexpr int GEN_TABLE = 1 << GEN_BITS;
constexpr int GEN_WINDOWS = 5;   // 5 * 13 = 65 >= 64 bits

// local index on one device of a multi-GPU context -> global (seeded) index; identity for a single device
MSMZ_HD uint64_t gen_global_index(uint32_t local, const GenMap& m) {
  const uint64_t blk = (uint64_t)local >> m.blk_shift;
  return ((blk * m.nshards + m.shard) << m.blk_shift) | ((uint64_t)local & ((1ull << m.blk_shift) - 1));
}

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
                                                    int endo, GenMap map) {
  constexpr int RW = 2 * F::NW;
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t a = splitmix64(seed, gen_global_index(i, map));
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
  store_affine<F>(out + (size_t)i * PointFmt<F>::STRIDE, r, inf);
  if (endo) {
    Fe<F> beta, bx;
    fe_set_const<F>(beta, F::BETA);
    if (!inf) {
      fe_mul(bx, r.x, beta);
      r.x = bx;
    }
    store_affine<F>(out + ((size_t)n + i) * PointFmt<F>::STRIDE, r, inf);
  }
}

template <class Fr>
__global__ void __launch_bounds__(256) k_gen_scalars(uint32_t* out, uint32_t n, uint64_t seed, GenMap map) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t gi = gen_global_index(i, map);
  constexpr int TOP_BITS = Fr::BITS - 224;            // bits kept in the top 32-bit word
  constexpr uint32_t TOP_MASK = TOP_BITS >= 32 ? 0xffffffffu : ((1u << TOP_BITS) - 1u);
  uint32_t w[8];
  for (uint32_t attempt = 0; attempt < 64; attempt++) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      uint64_t v = splitmix64(seed ^ 0x5ca1a75ull, (gi * 64 + attempt) * 4 + j);
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

// ------------------------------------------------------------------------------------------------ twisted Edwards
// Device records of twisted-Edwards input points are "Niels" form + x:  [y-x | y+x | 2d*x*y | x]
// (4*NW words).  The reference stores extended (X, Y, Z=1, T) instead (parallel.ts:209-232).
template <class F>
__device__ __forceinline__ void te_store_niels(uint32_t* rec, const Fe<F>& x, const Fe<F>& y) {
  Fe<F> ym, yp, t, k, kt;
  fe_sub(ym, y, x);
  fe_add(yp, y, x);
  fe_mul(t, x, y);
  fe_set_const<F>(k, F::K2D);
  fe_mul(kt, t, k);
  uint32_t w[2 * F::NW];
  fe_store<F>(w, ym);
  fe_store<F>(w + F::NW, yp);
  store_words<F>(rec, w);
  fe_store<F>(w, kt);
  fe_store<F>(w + F::NW, x);
  store_words<F>(rec + 2 * F::NW, w);
}

// canonical (x | y) -> Niels records
template <class F>
__global__ void __launch_bounds__(256) k_te_points_to_niels(uint32_t* out, const uint32_t* in, uint32_t n,
                                                            uint32_t* err) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  {
    uint32_t w[2 * F::NW];
    load_words<F>(w, in + (size_t)i * 2 * F::NW);
    if (words_geq<F::NW>(w, F::PW) || words_geq<F::NW>(w + F::NW, F::PW)) atomicOr(err, 4u);
  }
  Affine<F> p;
  load_affine<F>(p, in + (size_t)i * 2 * F::NW, 0);
  Fe<F> x, y;
  fe_to_mont(x, p.x);
  fe_to_mont(y, p.y);
  te_store_niels<F>(out + (size_t)i * 4 * F::NW, x, y);
}

// Niels records -> canonical (x | y):  x is stored, y = (y - x) + x
template <class F>
__global__ void __launch_bounds__(256) k_te_points_from_niels(uint32_t* out, const uint32_t* in, uint32_t n) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fe<F> ym, yp, kt, x, y, t;
  load_fe4<F>(ym, yp, kt, x, in + (size_t)i * 4 * F::NW);
  fe_add(y, ym, x);
  uint32_t w[2 * F::NW];
  fe_from_mont(t, x);
  fe_to_canon_words<F>(w, t);
  fe_from_mont(t, y);
  fe_to_canon_words<F>(w + F::NW, t);
  store_words<F>(out + (size_t)i * 2 * F::NW, w);
}

template <class F>
__device__ __forceinline__ void te_from_affine(TeExt<F>& p, const Affine<F>& a) {
  p.X = a.x;
  p.Y = a.y;
  fe_set_const<F>(p.Z, F::ONE);
  fe_mul(p.T, a.x, a.y);
}

template <class F>
__device__ __forceinline__ void te_to_affine_mont(Affine<F>& a, const TeExt<F>& p) {
  Fe<F> zi;
  fe_inverse(zi, p.Z);
  fe_mul(a.x, p.X, zi);
  fe_mul(a.y, p.Y, zi);
}

// table[k * GEN_TABLE + w] = w * base_k as affine [x | y] records (w = 0 -> the identity (0, 1))
template <class F>
__global__ void __launch_bounds__(128) k_te_gen_table(uint32_t* table, const uint32_t* bases) {
  constexpr int RW = 2 * F::NW;
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= GEN_WINDOWS * GEN_TABLE) return;
  uint32_t k = t / GEN_TABLE, w = t % GEN_TABLE;
  Affine<F> base;
  load_affine<F>(base, bases + (size_t)k * RW, 0);
  TeExt<F> b, acc, tmp;
  te_from_affine(b, base);
  te_set_zero(acc);
  for (int bit = GEN_BITS - 1; bit >= 0; bit--) {
    te_add(tmp, acc, acc);
    acc = tmp;
    if ((w >> bit) & 1u) {
      te_add(tmp, acc, b);
      acc = tmp;
    }
  }
  Affine<F> a;
  te_to_affine_mont(a, acc);
  store_affine<F>(table + (size_t)t * RW, a, false);
}

template <class F>
__global__ void __launch_bounds__(128) k_te_gen_points(uint32_t* out, const uint32_t* table, uint32_t n, uint64_t seed,
                                                       GenMap map) {
  constexpr int RW = 2 * F::NW;
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t a = splitmix64(seed, gen_global_index(i, map));
  TeExt<F> acc, tmp, q;
  te_set_zero(acc);
#pragma unroll 1
  for (int k = 0; k < GEN_WINDOWS; k++) {
    uint32_t w = (uint32_t)(a >> (GEN_BITS * k)) & (GEN_TABLE - 1);
    if (w == 0) continue;
    Affine<F> p;
    load_affine<F>(p, table + ((size_t)k * GEN_TABLE + w) * RW, 0);
    te_from_affine(q, p);
    te_add(tmp, acc, q);
    acc = tmp;
  }
  Affine<F> r;
  te_to_affine_mont(r, acc);
  te_store_niels<F>(out + (size_t)i * 4 * F::NW, r.x, r.y);
}

}  // namespace msmz
