// Two-dimensional bucket reduction.  Included by kernels.h.
//
// The reference's running-sum reduction (msm-batched-affine.ts:544-571) walks a window's L buckets as ONE chain; the
// grouped form of round 1-2 (k_reduce_first + quad levels) still leaves a dependency chain of 2 point additions per
// bit of the bucket index: 13 bits above the first level = 0.55 ms of latency at 2^20, and a first level that holds
// two XYZZ accumulators (256 VGPRs).  Here the bucket weight j in [0, L) is split into a high and a low half,
//     j = h * D + d,   h in [0, H), d in [0, D),   H = 2^ceil((c-1)/2), D = L / H,
//     sum_j j E_j = D * sum_h h R_h + sum_d d C_d,     R_h = sum_d E[h D + d],   C_d = sum_h E[h D + d]:
// the 2 L plain (unweighted) sums R, C are embarrassingly parallel -- any grouping, one accumulator per thread, mixed
// additions only -- and what is left are two weighted sums over H (<= 512) entries per bucket set instead of one over
// L: the chain is half as deep.  The host's Horner pass adds the row result at bit position c k + log2 D and the column
// result at c k (engine.h finalize_weierstrass_2d) -- no extra doublings.  The bucket of weight L = H D is added twice
// into row H/2, as before.
//
//   k_reduce2d_partial   thread = (problem, line, chunk): folds D / NC buckets of a row (or H / NC buckets of a column)
//                        into one XYZZ partial sum; problems 2 kw (rows) and 2 kw + 1 (columns) of
//                        bucket set kw, H lines each (the column problem's lines >= D are infinity), NC chunks per line
//   k_pairsum(_x4)       out[i] = in[2 i] + in[2 i + 1]: log2 NC launches turn the chunks of a line into its sum; the
//                        small late levels with one DPP quad per addition (xyzz_add_x4)
//   then the existing upper levels (k_reduce_quad16 / k_reduce_tail) run on 2 Keff problems of H entries with
//   rows = line sums, C = infinity.
#pragma once

namespace msmz {

struct R2Geom {
  uint32_t L, H, D;       // buckets per set, rows, columns (H * D = L, H >= D)
  uint32_t NC;            // chunks per line (power of two)
  uint32_t chr, chc;      // buckets per chunk along a row (D / NC) and along a column (H / NC)
  uint32_t nprob;         // 2 * Keff
};

template <class F>
__global__ void __launch_bounds__(128, MSMZ_REDUCE_OCC) k_reduce2d_partial(uint32_t* part, const uint32_t* slots,
                                                                           const uint32_t* points, const uint4* bfin,
                                                                           R2Geom g, uint32_t total) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const uint32_t per_prob = g.H * g.NC;
  const uint32_t prob = t / per_prob, u = t - prob * per_prob;
  const uint32_t kw = prob >> 1;
  const bool col = (prob & 1u) != 0;
  // the thread's buckets: weights j0, j0 + step, ... (count of them).  A plain sum may group buckets any way, so the
  // chunks of a ROW are interleaved (chunk q = buckets q, q + NC, q + 2 NC, ... of the row) and those of a COLUMN are
  // dealt so that neighbouring lanes hold neighbouring columns: either way the lanes of a wave read neighbouring
  // buckets in the same step (coalesced final-location words, neighbouring records).
  uint32_t line, chunk, j0 = 0, step = 0, count = 0;
  if (!col) {
    line = u / g.NC;
    chunk = u - line * g.NC;
    j0 = line * g.D + chunk;
    step = g.NC;
    count = g.chr;
  } else if (u < g.D * g.NC) {
    chunk = u / g.D;
    line = u - chunk * g.D;
    j0 = chunk * g.chc * g.D + line;
    step = g.D;
    count = g.chc;
  } else {
    // lines D .. H-1 of the column problem do not exist: infinity, so that both problems have H entries
    const uint32_t v = u - g.D * g.NC;
    line = g.D + v / g.NC;
    chunk = v - (v / g.NC) * g.NC;
  }
  Xyzz<F> acc;
  xyzz_set_inf(acc);
  // (A software pipeline -- final-location word two buckets ahead, record one bucket ahead -- measured no faster than
  // this plain loop once the chunks were interleaved: the kernel is bound by the multiplier at 2 waves per SIMD, and the
  // other wave covers the two dependent loads of a bucket.)
#pragma unroll 1
  for (uint32_t i = 0; i < count; i++) {
    const uint32_t j = j0 + i * step;
    if (j >= 1) add_bucket<F>(acc, kw * g.L + (j - 1), slots, points, bfin);
  }
  if (!col && line == g.H / 2 && chunk == 0) {
    // the single bucket of weight L = H D is folded in as 2 * (H/2) * D
    for (int twice = 0; twice < 2; twice++) add_bucket<F>(acc, kw * g.L + (g.L - 1), slots, points, bfin);
  }
  store_xyzz<F>(part + ((size_t)(prob * g.H + line) * g.NC + chunk) * 4 * F::NW, acc);
}

// The same partial sums for the msmBasic path (msm-basic.ts:106-128 buckets in XYZZ / extended coordinates): bucket g is
// the sum of the partial accumulators accs[cscan[g] .. cscan[g+1]) that k_bucket_accumulate left of it.
template <class P>
__global__ void __launch_bounds__(128, MSMZ_REDUCE_OCC) k_reduce2d_partial_acc(uint32_t* part, const uint32_t* accs,
                                                                               const uint32_t* cscan, R2Geom g,
                                                                               uint32_t total) {
  constexpr int XW = P::ACC_WORDS;
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const uint32_t per_prob = g.H * g.NC;
  const uint32_t prob = t / per_prob, u = t - prob * per_prob;
  const uint32_t kw = prob >> 1;
  const bool col = (prob & 1u) != 0;
  uint32_t line, chunk, j0 = 0, step = 0, count = 0;
  if (!col) {
    line = u / g.NC;
    chunk = u - line * g.NC;
    j0 = line * g.D + chunk;
    step = g.NC;
    count = g.chr;
  } else if (u < g.D * g.NC) {
    chunk = u / g.D;
    line = u - chunk * g.D;
    j0 = chunk * g.chc * g.D + line;
    step = g.D;
    count = g.chc;
  } else {
    const uint32_t v = u - g.D * g.NC;
    line = g.D + v / g.NC;
    chunk = v - (v / g.NC) * g.NC;
  }
  typename P::Acc acc, tmp, p;
  P::zero(acc);
  auto add_bucket_acc = [&](size_t gb) {
    const uint32_t q0 = cscan ? cscan[gb] : (uint32_t)gb, q1 = cscan ? cscan[gb + 1] : (uint32_t)gb + 1u;
    for (uint32_t q = q0; q < q1; q++) {
      P::load(p, accs + (size_t)q * XW);
      P::add(tmp, acc, p);
      acc = tmp;
    }
  };
  // the chunk ranges of the next bucket are requested one step ahead (cscan == nullptr: accs holds ONE accumulator
  // per bucket, in bucket order -- the sums k_bucket_sums formed)
  uint32_t qa = 0, qb = 0;
  auto range = [&](uint32_t i, uint32_t& a, uint32_t& b) {
    const uint32_t j = j0 + i * step;
    a = b = 0;
    if (i < count && j >= 1) {
      const size_t gb = (size_t)kw * g.L + (j - 1);
      a = cscan ? cscan[gb] : (uint32_t)gb;
      b = cscan ? cscan[gb + 1] : (uint32_t)gb + 1u;
    }
  };
  range(0, qa, qb);
#pragma unroll 1
  for (uint32_t i = 0; i < count; i++) {
    uint32_t na, nb;
    range(i + 1, na, nb);
    for (uint32_t q = qa; q < qb; q++) {
      P::load(p, accs + (size_t)q * XW);
      P::add(tmp, acc, p);
      acc = tmp;
    }
    qa = na;
    qb = nb;
  }
  if (!col && line == g.H / 2 && chunk == 0) {
    for (int twice = 0; twice < 2; twice++) add_bucket_acc((size_t)kw * g.L + (g.L - 1));
  }
  P::store(part + ((size_t)(prob * g.H + line) * g.NC + chunk) * XW, acc);
}

// Buckets of several partial accumulators (large msmBasic inputs): one accumulator per bucket, in bucket order, so that
// the two visits of the two-dimensional reduction read each bucket ONCE each instead of all its chunks twice.
template <class P>
__global__ void __launch_bounds__(128, MSMZ_REDUCE_OCC) k_bucket_sums(uint32_t* out, const uint32_t* accs, const uint32_t* cscan,
                                                                      uint32_t nb) {
  constexpr int XW = P::ACC_WORDS;
  const uint32_t gb = blockIdx.x * blockDim.x + threadIdx.x;
  if (gb >= nb) return;
  typename P::Acc acc, tmp, p;
  P::zero(acc);
  const uint32_t q0 = cscan[gb], q1 = cscan[gb + 1];
  if (q1 > q0) {
    P::load(acc, accs + (size_t)q0 * XW);
    for (uint32_t q = q0 + 1; q < q1; q++) {
      P::load(p, accs + (size_t)q * XW);
      P::add(tmp, acc, p);
      acc = tmp;
    }
  }
  P::store(out + (size_t)gb * XW, acc);
}

// n neutral accumulators (the C inputs of the first weighted level).  Not a memset: the twisted-Edwards identity is
// (0, 1, 1, 0), only the XYZZ infinity is an all-zero record.
template <class P>
__global__ void __launch_bounds__(256) k_fill_neutral(uint32_t* out, uint32_t n) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  typename P::Acc z;
  P::zero(z);
  P::store(out + (size_t)t * P::ACC_WORDS, z);
}

// out[i] = in[2 i] + in[2 i + 1], one thread per addition
template <class P>
__global__ void __launch_bounds__(128, MSMZ_REDUCE_OCC) k_pairsum(uint32_t* out, const uint32_t* in, uint32_t n_out) {
  constexpr int XW = P::ACC_WORDS;
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_out) return;
  typename P::Acc a, b, r;
  P::load(a, in + (size_t)(2 * t) * XW);
  P::load(b, in + (size_t)(2 * t + 1) * XW);
  P::add(r, a, b);
  P::store(out + (size_t)t * XW, r);
}

// the same with one DPP quad per addition (4 dependent field products instead of 14: the small, latency-bound levels)
template <class P>
__global__ void __launch_bounds__(64, MSMZ_Q16_OCC) k_pairsum_x4(uint32_t* out, const uint32_t* in, uint32_t n_out) {
  constexpr int XW = P::ACC_WORDS;
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t i = t >> 2;
  const int s = (int)(t & 3);
  const bool live = i < n_out;
  typename P::Acc a, b, r;
  P::zero(a);
  P::zero(b);
  if (live) {
    P::load(a, in + (size_t)(2 * i) * XW);
    P::load(b, in + (size_t)(2 * i + 1) * XW);
  }
  P::add_x4(r, a, b, s, false);
  if (live && s == 0) P::store(out + (size_t)i * XW, r);
}

}  // namespace msmz
