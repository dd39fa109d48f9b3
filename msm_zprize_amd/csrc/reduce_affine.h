// Batched-affine first level of the bucket reduction: the reference's `reduceBucketsAffine`
// (src/msm-batched-affine-single-thread.ts:522-667, doc/zprize22.md:317-358).  Included by kernels.h.
//
// The window's L bucket sums are cut into groups of S consecutive weights; inside every group the running-sum
// algorithm  R_t = R_{t-1} + E_{S-1-t},  P = R_0 + ... + R_{S-1}  (zprize22.md: "R = R + B_l; P = P + R") runs in
// lock-step over ALL groups of all windows, so that step t is ONE batch of affine additions (k_batch_add, 6 field
// products per addition instead of 10-14 in XYZZ) -- the doc's recursive splitting into independent sub-partitions is
// exactly this grouping.  The S-1 partial sums R_0 .. R_{S-2} (their sum is tri = sum_b b E_b) are then added up
// by a short pair tree, again in batches.  Launches: S-1 chain steps + ceil(log2(S-1)) tree rounds, each over
// (number of groups) pairs.  k_reduce_affine_finish converts (row, tri) to the scaled XYZZ records the upper levels
// (k_reduce_quad ...) expect -- the doc's "extra doublings don't have to be affine".
//
// Measured (profiles/r02_reduce_ab.txt): slower than the XYZZ first level on this GPU -- every launch costs the
// ~80 us latency floor of a batch inversion however small the batch -- so it is an option, not the default.
#pragma once

namespace msmz {

// Descriptor lists of all f2 launches for group `t` (one thread per group).
//   launch i = 0 .. S-2           chain step t = i+1:  R_t[a] = R_{t-1}[a] + E[a*S + S-1-t]
//   launch S-1 + u, u = 0 ..      tree round u over v_0 .. v_{S-2} (v_t = R_t): pairs (2i, 2i+1), odd one carried
// Result records: launch i writes records out0 + i * NG + a  (NG = number of groups; tree rounds use only the first
// `pairs_u * NG / NG`... every launch is given NG * ppg_i pairs, ppg_i = pairs per group of that launch).
// Empty elements (empty bucket, weight 0) are the all-zero "infinity" slot record inf_slot.
struct F2Geom {
  uint32_t L, S, groups, NG;      // buckets per window, group size, groups per window, Keff * groups
  uint32_t out0;                  // first slot record of the f2 results
  uint32_t inf_slot;              // a slot record of zeros (the host clears its whole group of 64 records)
  uint32_t n_launches;
  uint32_t desc_off[16];          // descriptor offset (in pairs) of every launch inside the f2 descriptor area
  uint32_t out_off[16];           // record offset of every launch's results (relative to out0)
  uint32_t ppg[16];               // pairs per group in every launch
};

__device__ __forceinline__ uint32_t f2_element_loc(const uint4* bfin, const F2Geom& g, uint32_t k, uint32_t j) {
  if (j == 0 || j >= g.L) return g.inf_slot;               // weight 0 is empty; weight L is added by the finish kernel
  const uint32_t loc = bfin[(size_t)k * g.L + (j - 1)].x;   // tail_skip = 0: one location per bucket
  return loc == LOC_NONE ? g.inf_slot : loc;
}

static __global__ void __launch_bounds__(256) k_reduce_affine_desc(uint2* desc, const uint4* bfin, F2Geom g) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= g.NG) return;
  const uint32_t k = t / g.groups, a = t - k * g.groups;
  const uint32_t S = g.S;
  // chain: location of R_t (t >= 1) = out0 + out_off[t-1] + group
  auto r_loc = [&](uint32_t step) -> uint32_t {
    return step == 0 ? f2_element_loc(bfin, g, k, a * S + S - 1) : g.out0 + g.out_off[step - 1] + t;
  };
  for (uint32_t step = 1; step < S; step++)
    desc[g.desc_off[step - 1] + t] = make_uint2(r_loc(step - 1), f2_element_loc(bfin, g, k, a * S + S - 1 - step));
  // tree over v_0 .. v_{S-2}
  uint32_t cur[16];                 // S <= 16: locations of the current list
  uint32_t n = S - 1;
  for (uint32_t i = 0; i < n; i++) cur[i] = r_loc(i);
  uint32_t launch = S - 1;
  while (n > 1) {
    const uint32_t np = n / 2;
    for (uint32_t i = 0; i < np; i++) {
      desc[g.desc_off[launch] + (size_t)t * np + i] = make_uint2(cur[2 * i], cur[2 * i + 1]);
      cur[i] = g.out0 + g.out_off[launch] + t * np + i;
    }
    if (n & 1) cur[np] = cur[n - 1];
    n = np + (n & 1);
    launch++;
  }
  // where row and tri of this group ended up: kept in the first two words of the group's descriptor of a pseudo-launch
  desc[g.desc_off[g.n_launches] + t] = make_uint2(r_loc(S - 1), S >= 2 ? cur[0] : g.inf_slot);
}

// (row, tri) of every group, affine -> the scaled XYZZ records of the upper levels:
//   rows[t] = S * (row + 2 B_L if the group holds weight L/2),   tris[t] = tri
template <class F>
__global__ void __launch_bounds__(128) k_reduce_affine_finish(uint32_t* rows, uint32_t* tris, const uint32_t* slots,
                                                              const uint32_t* points, const uint2* where,
                                                              const uint4* bfin, F2Geom g) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= g.NG) return;
  const uint32_t k = t / g.groups, a = t - k * g.groups;
  const uint2 w = where[t];
  Affine<F> p;
  Xyzz<F> row, tri, tmp;
  bool inf = load_operand<F, true>(p, w.x, slots, points);
  if (inf) xyzz_set_inf(row); else xyzz_from_affine(row, p);
  inf = load_operand<F, true>(p, w.y, slots, points);
  if (inf) xyzz_set_inf(tri); else xyzz_from_affine(tri, p);
  if (g.L >= 2 && a * g.S == g.L / 2) {
    // the single bucket of weight L is folded in as 2 * (L/2): keeps the element count a power of two
    const uint32_t loc = bfin[(size_t)k * g.L + (g.L - 1)].x;
    if (loc != LOC_NONE) {
      inf = load_operand<F, true>(p, loc, slots, points);
      for (int twice = 0; twice < 2; twice++) {
        xyzz_madd(tmp, row, p, inf);
        row = tmp;
      }
    }
  }
  for (uint32_t s = g.S; s > 1; s >>= 1) {
    xyzz_dbl(tmp, row);
    row = tmp;
  }
  store_xyzz<F>(rows + (size_t)t * 4 * F::NW, row);
  store_xyzz<F>(tris + (size_t)t * 4 * F::NW, tri);
}

}  // namespace msmz
