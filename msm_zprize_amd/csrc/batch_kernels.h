// One tree round of batched-affine bucket additions (rows a6 / a12 of SURVEY.md section 8).  Included by kernels.h.
//
// Pair t of a launch adds the two operands its descriptor dsc[t] names (plan_kernels.h; the schedule of the tree rounds
// is the reference's, msm-batched-affine.ts:232-247) and writes record out_base + t of the launch's compact,
// chunk-interleaved result array (kernels.h).  Round 0 gathers the original points.
//
// Batch inversion (Montgomery's trick) on two levels:
//   * each thread walks B pairs like batchAddUnsafeNew (curve-affine.ts:463-522): the forward pass reads only the x
//     coordinates, keeps a running product of the denominators and parks  z_i = prod_{j<i} d_j  in the pair's output
//     record; the backward pass reads the descriptor again, reloads both points and z_i, and  z_i * (running inverse)
//     is 1 / d_i;
//   * the T per-thread products of a workgroup are inverted together: product tree in LDS, ONE field
//     inversion (wave 0, fe_inverse_wave), down-sweep.
// 6 field products per addition + (3 T + inversion) per workgroup of T*B additions.
// Slope/sum formulas use P2 (y3 = m (x2 - x3) - y2, wasm/curve.ts:63-84 addAffinePacked).
//
// SAFE handles infinity operands, equal points (doubling, denominator 2y) and opposite points
// (result infinity) like batchAddNew (curve-affine.ts:376-458); the unsafe variant assumes distinct
// x like batchAddUnsafeNew and raises meta->error if a zero denominator poisons a batch.
#pragma once

namespace msmz {

#ifndef MSMZ_BATCH_BMAX
#define MSMZ_BATCH_BMAX 16
#endif
#ifdef MSMZ_EXP_NOMUL   // timing experiment (wrong results): the six products of an addition cost nothing
#define BM_MUL(r, a, b) fe_add(r, a, b)
#define BM_SQR(r, a) fe_add(r, a, a)
#else
#define BM_MUL(r, a, b) fe_mul(r, a, b)
#define BM_SQR(r, a) fe_sqr(r, a)
#endif
enum { PK_NONE = 0, PK_ADD = 1, PK_DBL = 2, PK_TAKE_A = 3, PK_TAKE_B = 4, PK_INF = 5 };

template <class F, int T, bool SAFE, int OCC, int BMAX>
__global__ void __launch_bounds__(T, OCC) k_batch_add(uint32_t* slots, const uint32_t* points, const uint2* dsc,
                                                      uint32_t out_base, uint32_t total, int B, MsmMeta* meta) {
  // product tree over the T per-thread products.  Level 1 (pairs of neighbouring lanes) is formed with
  // a lane shuffle, levels 1..log2(T) live in LDS limb-major: level d at offset T - (T >> (d-1)), T-1 nodes.
  __shared__ int32_t tree[F::N * T];
  __shared__ int32_t s_prev[F::N * T];   // limb-major: the prefix product before each thread's LAST pair (not parked in HBM)
  __shared__ uint8_t s_kind[SAFE ? BMAX * T : 1];
  constexpr int N = F::N;
  const uint32_t block_base = blockIdx.x * (uint32_t)(T * B);
#ifdef MSMZ_TRACE
  uint64_t* trace = reinterpret_cast<uint64_t*>(meta + 1);   // (the trace build allocates the stamps behind the meta block)
#endif
  MSMZ_STAMP(trace, 0);
  MSMZ_STAMP_HW(trace);

  Fe<F> prefix;
  fe_set_const<F>(prefix, F::ONE);
  // index of the thread's last pair (pairs t = block_base + i*T + threadIdx.x < total)
  int ilast = -1;
  if (block_base + threadIdx.x < total) {
    const uint32_t room = (total - 1u - block_base - threadIdx.x) / (uint32_t)T;
    ilast = room < (uint32_t)(B - 1) ? (int)room : B - 1;
  }
  // ---------------------------------------------------------------- forward pass
  // (the next pair's descriptor is requested one iteration ahead: operand addresses never wait for it)
  uint2 dnext = make_uint2(0, 0);
  if (block_base + threadIdx.x < total) dnext = dsc[block_base + threadIdx.x];
#pragma unroll 1
  for (int i = 0; i < B; i++) {
    const uint32_t t = block_base + (uint32_t)i * T + threadIdx.x;
    uint32_t kind = PK_NONE;
    uint2 dd = dnext;
    if (i + 1 < B && t + (uint32_t)T < total) dnext = dsc[t + (uint32_t)T];
    if (t < total) {
#if defined(MSMZ_EXP_LOCALMEM) || defined(MSMZ_EXP_LOCAL_FWD)   // timing experiment (wrong results): every operand from a small cache-resident set
      dd.x = LOC_ORIG | (t & 4095u);
      dd.y = LOC_ORIG | ((t + 1u) & 4095u);
#endif
      // the unsafe path only needs the denominators here: x coordinates (the numerator joins in the backward pass)
      Fe<F> d;
      kind = PK_ADD;
      if constexpr (SAFE && SlotFmt<F>::PACKED) {
        // x coordinates first; the y coordinates are fetched only where an edge case is possible: a zero x (the
        // record may be the all-zero infinity record) or equal x (doubling / opposite points)
        Fe<F> x1, x2;
        const uint32_t oa = load_operand_x_or<F>(x1, dd.x, slots, points);
        const uint32_t ob = load_operand_x_or<F>(x2, dd.y, slots, points);
        fe_sub(d, x2, x1);
        if (oa == 0 || ob == 0 || fe_is_zero(d)) {
          Affine<F> p1, p2;
          const bool infA = load_operand<F, true>(p1, dd.x, slots, points);
          const bool infB = load_operand<F, true>(p2, dd.y, slots, points);
          if (infA) {
            kind = infB ? PK_INF : PK_TAKE_B;
          } else if (infB) {
            kind = PK_TAKE_A;
          } else if (fe_is_zero(d)) {
            Fe<F> num;
            fe_sub(num, p2.y, p1.y);
            if (fe_is_zero(num) && !fe_is_zero(p1.y)) {
              kind = PK_DBL;
              fe_add(d, p2.y, p2.y);          // 2y
            } else {
              kind = PK_INF;
            }
          }
        }
      } else if constexpr (SAFE) {
        Affine<F> p1, p2;
        const bool infA = load_operand<F, true>(p1, dd.x, slots, points);
        const bool infB = load_operand<F, true>(p2, dd.y, slots, points);
        Fe<F> num;
        fe_sub(d, p2.x, p1.x);
        fe_sub(num, p2.y, p1.y);
        if (infA) {
          kind = infB ? PK_INF : PK_TAKE_B;
        } else if (infB) {
          kind = PK_TAKE_A;
        } else if (fe_is_zero(d)) {
          if (fe_is_zero(num) && !fe_is_zero(p1.y)) {
            kind = PK_DBL;
            fe_add(d, p2.y, p2.y);          // 2y
          } else {
            kind = PK_INF;
          }
        }
      } else {
        Fe<F> x1, x2;
        load_operand_x<F>(x1, dd.x, slots, points);
        load_operand_x<F>(x2, dd.y, slots, points);
        fe_sub(d, x2, x1);
      }
      if (kind == PK_ADD || kind == PK_DBL) {
        // park the running product of the denominators BEFORE this pair in the pair's output record
        // (a product's limbs are valid inputs as they are)
        // (the first pair's is the constant one and the last pair's stays in LDS: neither goes to HBM)
        if (i == ilast) {
#pragma unroll
          for (int j = 0; j < N; j++) s_prev[j * T + threadIdx.x] = prefix.l[j];
        } else if (i > 0) {
#if defined(MSMZ_EXP_LOCALMEM) || defined(MSMZ_EXP_LOCAL_Z)
          slot_store_mulout<F>(slots + slot_offset<F>(blockIdx.x * 64u + (t & 63u)), prefix);
#else
          slot_store_mulout<F>(slots + slot_offset<F>(out_base + t), prefix);
#endif
        }
        Fe<F> np;
        BM_MUL(np, prefix, d);
        prefix = np;
      }
    }
    if (SAFE) s_kind[i * T + threadIdx.x] = (uint8_t)kind;
  }

  MSMZ_STAMP(trace, 1);
  // product tree, inversion and down-sweep are few instructions on the critical path of the batch: at raised wave
  // priority they are not held up by the other workgroups' passes (accumulate 2.43 -> 2.35 ms at 2^20, 18.4 -> 18.1 at 2^23)
  __builtin_amdgcn_s_setprio(3);
  // ---------------------------------------------------------------- workgroup-wide inversion of the T products
  Fe<F> run;   // inverse of the product of this thread's denominators = inv(level-1 node) * partner's product
  {
    Fe<F> partner, node;
#pragma unroll
    for (int j = 0; j < N; j++) partner.l[j] = __shfl_xor(prefix.l[j], 1, 64);
    fe_mul(node, prefix, partner);
    if ((threadIdx.x & 1) == 0) {
#pragma unroll
      for (int j = 0; j < N; j++) tree[j * T + (threadIdx.x >> 1)] = node.l[j];
    }
  }
  __syncthreads();
  int lvl_off = 0;   // offset of the level being consumed (level 1 first)
#pragma unroll 1
  for (int width = T >> 2; width >= 1; width >>= 1) {
    const int child_off = lvl_off;
    lvl_off += width * 2;
    if ((int)threadIdx.x < width) {
      Fe<F> x, y, z;
#pragma unroll
      for (int j = 0; j < N; j++) {
        x.l[j] = tree[j * T + child_off + 2 * threadIdx.x];
        y.l[j] = tree[j * T + child_off + 2 * threadIdx.x + 1];
      }
      fe_mul(z, x, y);
#pragma unroll
      for (int j = 0; j < N; j++) tree[j * T + lvl_off + threadIdx.x] = z.l[j];
    }
    __syncthreads();
  }
  MSMZ_STAMP(trace, 2);
  if (threadIdx.x < 64) {
    Fe<F> root, inv;
#pragma unroll
    for (int j = 0; j < N; j++) root.l[j] = tree[j * T + lvl_off];
#ifdef MSMZ_EXP_NOINV   // timing experiment (wrong results): no inversion
    const bool ok = true;
    inv = root;
#else
    const bool ok = fe_inverse_wave(inv, root);
#endif
    if (threadIdx.x == 0) {
      if (!ok) atomicOr(&meta->error, 1u);
#pragma unroll
      for (int j = 0; j < N; j++) tree[j * T + lvl_off] = inv.l[j];
    }
  }
  __syncthreads();
  MSMZ_STAMP(trace, 3);
  // down-sweep: one thread per CHILD (inverse of a child = inverse of the parent times the sibling), so a level is one
  // field product deep, not two (the sweep is on the critical path of every batch: ~1.8 us per product when the
  // workgroup's other waves wait at the barrier)
#pragma unroll 1
  for (int width = 1; width <= T >> 2; width <<= 1) {
    const int parent_off = lvl_off;
    lvl_off -= width * 2;
    Fe<F> ci;
    const bool mine = (int)threadIdx.x < 2 * width;
    if (mine) {
      Fe<F> pi, sib;
#pragma unroll
      for (int j = 0; j < N; j++) {
        pi.l[j] = tree[j * T + parent_off + (threadIdx.x >> 1)];
        sib.l[j] = tree[j * T + lvl_off + (threadIdx.x ^ 1)];
      }
      fe_mul(ci, pi, sib);
    }
    __syncthreads();   // every sibling has been read before a child is overwritten
    if (mine) {
#pragma unroll
      for (int j = 0; j < N; j++) tree[j * T + lvl_off + threadIdx.x] = ci.l[j];
    }
    __syncthreads();
  }
  {
    Fe<F> partner, ninv;
#pragma unroll
    for (int j = 0; j < N; j++) {
      partner.l[j] = __shfl_xor(prefix.l[j], 1, 64);
      ninv.l[j] = tree[j * T + (threadIdx.x >> 1)];
    }
    fe_mul(run, ninv, partner);
  }

  MSMZ_STAMP(trace, 4);
  __builtin_amdgcn_s_setprio(0);   // (the backward pass at priority 1 instead: no difference)
  // ---------------------------------------------------------------- backward pass
  {
    const uint32_t tl = block_base + (uint32_t)(B - 1) * T + threadIdx.x;
    if (tl < total) dnext = dsc[tl];
  }
#pragma unroll 1
  for (int i = B - 1; i >= 0; i--) {
    const uint32_t t = block_base + (uint32_t)i * T + threadIdx.x;
    uint2 dd = dnext;
    if (i > 0 && t - (uint32_t)T < total) dnext = dsc[t - (uint32_t)T];
    if (t >= total) continue;
    const uint32_t kind = SAFE ? s_kind[i * T + threadIdx.x] : (uint32_t)PK_ADD;
#if defined(MSMZ_EXP_LOCALMEM) || defined(MSMZ_EXP_LOCAL_BWD)
    dd.x = LOC_ORIG | (t & 4095u);
    dd.y = LOC_ORIG | ((t + 1u) & 4095u);
#endif
#if defined(MSMZ_EXP_LOCALMEM) || defined(MSMZ_EXP_LOCAL_Z)
    uint32_t* out = slots + slot_offset<F>(blockIdx.x * 64u + (t & 63u));
#else
    uint32_t* out = slots + slot_offset<F>(out_base + t);
#endif
    if (kind == PK_ADD || kind == PK_DBL) {
      Affine<F> p2;
      load_operand<F, false>(p2, dd.y, slots, points);
      Fe<F> z, mm, ms, d, tt, s12, num;
      // product of the denominators before this pair: parked by the forward pass, except the last pair's (LDS)
      // and the first pair's (one)
      if (i == ilast) {
#pragma unroll
        for (int j = 0; j < N; j++) z.l[j] = s_prev[j * T + threadIdx.x];
      } else if (i > 0) {
        slot_load_fe<F>(z, out);
      } else {
        fe_set_const<F>(z, F::ONE);
      }
      if (kind == PK_ADD) {
        Affine<F> p1;
        load_operand<F, false>(p1, dd.x, slots, points);
        fe_sub(d, p2.x, p1.x);
        fe_sub(num, p2.y, p1.y);
        fe_add(s12, p2.x, p1.x);
      } else {
        fe_add(d, p2.y, p2.y);
        fe_add(s12, p2.x, p2.x);
        Fe<F> xx;
        fe_sqr(xx, p2.x);
        fe_add(num, xx, xx);
        fe_add(num, num, xx);               // 3x^2
        fe_carry(num);
      }
      BM_MUL(tt, z, run);                 // 1 / denominator of this pair
      BM_MUL(mm, num, tt);                // slope
      BM_MUL(tt, run, d);
      run = tt;
      BM_SQR(ms, mm);
      Affine<F> res;
      fe_sub(res.x, ms, s12);             // x3 = m^2 - x1 - x2
      fe_sub(tt, p2.x, res.x);
      fe_carry(tt);
      BM_MUL(ms, mm, tt);
      fe_sub(res.y, ms, p2.y);            // y3 = m (x2 - x3) - y2
      slot_store_point<F>(out, res, false);
    } else if (kind == PK_TAKE_A || kind == PK_TAKE_B) {
      Affine<F> p;
      load_operand<F, false>(p, kind == PK_TAKE_A ? dd.x : dd.y, slots, points);
      slot_store_point<F>(out, p, false);
    } else {                              // PK_INF
      Affine<F> dummy;
      fe_zero(dummy.x);
      fe_zero(dummy.y);
      slot_store_point<F>(out, dummy, true);
    }
  }
  MSMZ_STAMP(trace, 5);
}

}  // namespace msmz
