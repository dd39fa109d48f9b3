// HIP kernels of the Pippenger MSM pipeline (gfx950, wave64).  One kernel family per row of the
// hot-path table:
//
//   k_points_to_mont     upload: canonical bytes -> lazy Montgomery residues (+ endomorphism copy)
//                        (parallel.ts:97-112 pointsFromBytes, field-msm.ts:183-185, wasm/curve.ts:90-103)
//   k_hist, k_bin_scan,  (sort_kernels.h) GLV split + signed c-bit digits + coarse-bin histogram, bin offsets, and the
//   k_coarse, k_fine     counting sort of point *indices* into bucket order in two LDS-staged passes; the digits are
//                        re-sliced from the scalars by every pass, never stored
//                        (msm-batched-affine.ts:149,172-200 slicing; :411-435 integrateBucketCounts; :444-490 sortPoints
//                        -- which copies 116-byte points; here 4-byte references are sorted and points are gathered
//                        on first use)
//   k_digits, k_scan_*,  fallback sort for window sizes whose coarse bins do not fit the LDS staging (digits
//   k_scatter            materialized, one global atomic per entry); k_scan_* also serve the msmBasic path's chunk offsets
//   k_plan_count,        (plan_kernels.h) the schedule of the tree rounds as data: rounds decided on the device, one
//   k_plan_emit          {locA, locB} descriptor per addition (msm-batched-affine.ts:232-247), final locations per bucket
//   k_batch_add          (batch_kernels.h) one tree round of batched-affine additions, with a workgroup-wide Montgomery
//                        batch inversion (product tree in LDS, one wave-wide field inversion per workgroup,
//                        fe_inverse_wave); results in chunk-interleaved slot arrays
//                        (msm-batched-affine.ts:232-270; curve-affine.ts:376-522; inverse.ts:220-271)
//   k_reduce_first,      bucket reduction  sum_l l*B_l  by grouped running sums in XYZZ coordinates: first level
//   k_reduce_quad(16),   from the (partial) bucket sums, upper levels with a quad of lanes per group / per addition,
//   k_reduce_tail,       the last levels in one launch; k_reduce_next = first level of the msmBasic path
//   k_reduce_next        (msm-batched-affine.ts:544-571 reduceBucketsColumnProjective)
//   k_reduce_affine_*    (reduce_affine.h) optional batched-affine first level (reduceBucketsAffine,
//                        msm-batched-affine-single-thread.ts:522-667)
//   k_bucket_accumulate  msmBasic path: buckets in XYZZ / extended coordinates (msm-basic.ts:106-128)
//   k_test_*             (test_kernels.h) stage-level test hooks of include/msmz_test.h
//
// Bucket numbering: global bucket g = k*L + (l-1) for window k and digit l in [1, L], L = 2^(c-1).
// Sorted references: ref = point_index | (negate << 31).
#pragma once
#include <hip/hip_runtime.h>
#include "constants_gen.h"
#include "curve.h"
#include "scalar.h"

namespace msmz {

constexpr uint32_t REF_NEG = 0x80000000u;
constexpr uint32_t REF_IDX = 0x7fffffffu;

struct MsmMeta {               // small device-resident block of run-time totals
  uint32_t max_bucket;         // largest bucket size
  uint32_t n_entries;          // E = number of non-zero digits = point additions' inputs
  uint32_t error;              // bit 0: zero denominator hit in the unsafe batch add; bit 1: a scalar did not fit K windows
  uint32_t rounds;             // tree rounds the plan scheduled
  uint32_t round_pairs[32];    // number of pairs in tree round r
  uint32_t round_base[32];     // first record of round r's result array inside `slots` (prefix sum of round_pairs)
};

// ------------------------------------------------------------------------------------------------ loads
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// Resident point sets: record i of a set starts at word i * PointFmt<F>::STRIDE.  A 96-byte record [x | y] of the
// 377/381-bit curves is padded to 128 bytes: packed back to back, every second record starts in the middle of a 64-byte
// sector, so the x-only gathers of the tree rounds' forward pass (48 bytes) touched 1.5 sectors on average and a whole
// record 2.5 of 128-byte-line granularity; at 128-byte stride x is ONE sector and the record one line.  Memory is the
// cheap side (288 GB).  The 64-byte records of the 255-bit curves already are one sector.  Wire-format staging buffers
// (uploads / downloads) stay packed.
#ifndef MSMZ_POINT_PAD
#define MSMZ_POINT_PAD 1
#endif
template <class F>
struct PointFmt {
  static constexpr int STRIDE = (MSMZ_POINT_PAD && 2 * F::NW == 24) ? 32 : 2 * F::NW;   // 32-bit words
};

// NT = non-temporal access.  Measured (round 1): streaming the tree rounds' slot records past the caches makes
// k_batch_add 1.7x SLOWER (round 0: 1.9 -> 3.9 ms) -- the backward pass re-reads what the forward pass parked
// (z, x1) and lives off L2 / Infinity Cache hits -- so SLOT_NT stays false.
constexpr bool SLOT_NT = false;
// `cs` = distance between a record's consecutive 16-byte chunks, in chunks: 1 for ordinary records, SLOT_CS for
// the chunk-interleaved slot arrays (below).
template <class F, bool NT = false>
__device__ __forceinline__ void load_words(uint32_t* dst, const uint32_t* src, int cs = 1) {
  // records are 16-byte aligned: 2*NW words = 96 B / 64 B
  const u32x4* s4 = reinterpret_cast<const u32x4*>(src);
#pragma unroll
  for (int i = 0; i < (2 * F::NW) / 4; i++) {
    u32x4 v = NT ? __builtin_nontemporal_load(s4 + i * cs) : s4[i * cs];
    dst[4 * i] = v.x;
    dst[4 * i + 1] = v.y;
    dst[4 * i + 2] = v.z;
    dst[4 * i + 3] = v.w;
  }
}

template <class F, bool NT = false>
__device__ __forceinline__ void store_words(uint32_t* dst, const uint32_t* src, int cs = 1) {
  u32x4* d4 = reinterpret_cast<u32x4*>(dst);
#pragma unroll
  for (int i = 0; i < (2 * F::NW) / 4; i++) {
    u32x4 v = {src[4 * i], src[4 * i + 1], src[4 * i + 2], src[4 * i + 3]};
    if (NT) __builtin_nontemporal_store(v, d4 + i * cs); else d4[i * cs] = v;
  }
}

// load an affine point record; returns true if it is the point at infinity
template <class F, bool NT = false>
__device__ __forceinline__ bool load_affine(Affine<F>& p, const uint32_t* rec, uint32_t negate, int cs = 1) {
  uint32_t w[2 * F::NW];
  load_words<F, NT>(w, rec, cs);
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < 2 * F::NW; i++) o |= w[i];
  fe_unpack<F>(p.x, w);
  Fe<F> y;
  fe_unpack<F>(y, w + F::NW);
  fe_cneg(p.y, y, negate);
  return o == 0;
}

template <class F, bool NT = false>
__device__ __forceinline__ void store_affine(uint32_t* rec, const Affine<F>& p, bool inf, int cs = 1) {
  uint32_t w[2 * F::NW];
  if (inf) {
#pragma unroll
    for (int i = 0; i < 2 * F::NW; i++) w[i] = 0;
  } else {
    fe_store<F>(w, p.x);
    fe_store<F>(w + F::NW, p.y);
    // a finite point can never serialize to the all-zero record: x = y = 0 is not on the curve
  }
  store_words<F, NT>(rec, w, cs);
}

// Slot arrays (results of the tree rounds) hold affine points [x | y], values reduced to [0, 3p); the all-zero
// record is the point at infinity.  Two record formats (MSMZ_SLOT_PACKED):
//   packed (default)  2*NW saturated words like the resident points (96 / 64 bytes)
//   limb form         the N signed limbs of x, then of y, one word each (112 / 80 bytes): nothing to pack or unpack
//                     between two tree rounds (~400 instructions per addition) but 17-25 % more bytes
// The tree rounds are bound by memory traffic (measured: 2.3 ms with the field products removed, 1.85 ms with every
// operand served from cache), so the smaller record wins.
// The arrays are chunk-interleaved in groups of 64 records: chunk q (16 bytes) of record r sits at 16-byte unit
// (r / 64) * 64 * CH + q * 64 + r % 64, CH = chunks per record.  A wave whose lanes hold consecutive records (the
// writers) then moves one contiguous KB per load/store instruction, and the readers of the next round (lane i wants
// records 2i, 2i + 1) two KB, instead of 64 pieces a record apart.
#ifndef MSMZ_SLOT_PACKED
#define MSMZ_SLOT_PACKED 1
#endif
constexpr int SLOT_CS = 64;
template <class F>
struct SlotFmt {
  static constexpr bool PACKED = MSMZ_SLOT_PACKED != 0;
  static constexpr int FE_WORDS = PACKED ? F::NW : F::N;                  // words of one coordinate
  static constexpr int WORDS = PACKED ? 2 * F::NW : (2 * F::N + 3) / 4 * 4;   // words per record
  static constexpr int CH = WORDS / 4;                                    // 16-byte chunks per record
  static constexpr int XCH = (FE_WORDS + 3) / 4;                          // chunks that cover x (or a parked field element)
};
template <class F>
__device__ __forceinline__ size_t slot_offset(uint32_t rec) {   // in 32-bit words
  return ((size_t)(rec >> 6) * (SlotFmt<F>::CH * 64) + (rec & 63u)) * 4;
}

// first NCH chunks of a slot record -> words
template <int NCH>
__device__ __forceinline__ void slot_load_chunks(uint32_t* w, const uint32_t* rec) {
  const u32x4* s4 = reinterpret_cast<const u32x4*>(rec);
#pragma unroll
  for (int i = 0; i < NCH; i++) {
    const u32x4 v = s4[i * SLOT_CS];
    w[4 * i] = v.x;
    w[4 * i + 1] = v.y;
    w[4 * i + 2] = v.z;
    w[4 * i + 3] = v.w;
  }
}
template <int NCH>
__device__ __forceinline__ void slot_store_chunks(uint32_t* rec, const uint32_t* w) {
  u32x4* d4 = reinterpret_cast<u32x4*>(rec);
#pragma unroll
  for (int i = 0; i < NCH; i++) {
    const u32x4 v = {w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]};
    d4[i * SLOT_CS] = v;
  }
}

// whole record; returns true if it is the point at infinity (only evaluated when CHECK_INF)
template <class F, bool CHECK_INF>
__device__ __forceinline__ bool slot_load_point(Affine<F>& p, const uint32_t* rec) {
  using S = SlotFmt<F>;
  uint32_t w[S::WORDS];
  slot_load_chunks<S::CH>(w, rec);
  uint32_t o = 0;
  if (CHECK_INF) {
#pragma unroll
    for (int j = 0; j < 2 * S::FE_WORDS; j++) o |= w[j];
  }
  if constexpr (S::PACKED) {
    fe_unpack<F>(p.x, w);
    fe_unpack<F>(p.y, w + F::NW);
  } else {
#pragma unroll
    for (int j = 0; j < F::N; j++) {
      p.x.l[j] = (int32_t)w[j];
      p.y.l[j] = (int32_t)w[F::N + j];
    }
  }
  return CHECK_INF && o == 0;
}
// the x coordinate of a point record / a parked field element
template <class F>
__device__ __forceinline__ void slot_load_fe(Fe<F>& x, const uint32_t* rec) {
  using S = SlotFmt<F>;
  uint32_t w[S::XCH * 4];
  slot_load_chunks<S::XCH>(w, rec);
  if constexpr (S::PACKED) {
    fe_unpack<F>(x, w);
  } else {
#pragma unroll
    for (int j = 0; j < F::N; j++) x.l[j] = (int32_t)w[j];
  }
}
// park a direct mul/sqr output (value in (-1.5p, 0.5p), normalized limbs with a signed top limb)
template <class F>
__device__ __forceinline__ void slot_store_mulout(uint32_t* rec, const Fe<F>& x) {
  using S = SlotFmt<F>;
  uint32_t w[S::XCH * 4];
  if constexpr (S::PACKED) {
    fe_store_mulout<F>(w, x);
#pragma unroll
    for (int j = F::NW; j < S::XCH * 4; j++) w[j] = 0;
  } else {
#pragma unroll
    for (int j = 0; j < S::XCH * 4; j++) w[j] = j < F::N ? (uint32_t)x.l[j] : 0u;
  }
  slot_store_chunks<S::XCH>(rec, w);
}
// x, y: any lazy values |v| < 2^4 p; stored reduced to [0, 3p)
template <class F>
__device__ __forceinline__ void slot_store_point(uint32_t* rec, const Affine<F>& p, bool inf) {
  using S = SlotFmt<F>;
  uint32_t w[S::WORDS];
  if constexpr (S::PACKED) {
    fe_store<F>(w, p.x);
    fe_store<F>(w + F::NW, p.y);
  } else {
    Fe<F> x = p.x, y = p.y;
    fe_reduce_small(x);
    fe_reduce_small(y);
#pragma unroll
    for (int j = 0; j < S::WORDS; j++)
      w[j] = j < F::N ? (uint32_t)x.l[j] : (j < 2 * F::N ? (uint32_t)y.l[j - F::N] : 0u);
  }
  if (inf) {
#pragma unroll
    for (int j = 0; j < S::WORDS; j++) w[j] = 0;
  }
  slot_store_chunks<S::CH>(rec, w);
}

template <class F>
__device__ __forceinline__ void load_xyzz(Xyzz<F>& p, const uint32_t* rec) {
  uint32_t w[2 * F::NW];
  load_words<F>(w, rec);
  fe_unpack<F>(p.X, w);
  fe_unpack<F>(p.Y, w + F::NW);
  load_words<F>(w, rec + 2 * F::NW);
  fe_unpack<F>(p.ZZ, w);
  fe_unpack<F>(p.ZZZ, w + F::NW);
}

template <class F>
__device__ __forceinline__ void store_xyzz(uint32_t* rec, const Xyzz<F>& p) {
  uint32_t w[2 * F::NW];
  fe_store<F>(w, p.X);
  fe_store<F>(w + F::NW, p.Y);
  store_words<F>(rec, w);
  fe_store<F>(w, p.ZZ);
  fe_store<F>(w + F::NW, p.ZZZ);
  store_words<F>(rec + 2 * F::NW, w);
}

// ------------------------------------------------------------------------------------------------ upload
// in : n records of 2*NW canonical little-endian words (x | y), optional infinity flags
// out: n records in memory format; with `endo`, records [n, 2n) hold (beta*x, y)
template <class F>
__global__ void __launch_bounds__(256) k_points_to_mont(uint32_t* out, const uint32_t* in, const uint8_t* is_inf,
                                                        uint32_t n, int endo, uint32_t* err) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Affine<F> p, m;
  {
    // coordinates must be canonical (< p); flagged here instead of in a serial host loop
    uint32_t w[2 * F::NW];
    load_words<F>(w, in + (size_t)i * 2 * F::NW);
    if (words_geq<F::NW>(w, F::PW) || words_geq<F::NW>(w + F::NW, F::PW)) atomicOr(err, 4u);
  }
  bool inf = load_affine<F>(p, in + (size_t)i * 2 * F::NW, 0);
  (void)inf;
  bool flagged = is_inf != nullptr && is_inf[i] != 0;
  fe_to_mont(m.x, p.x);
  fe_to_mont(m.y, p.y);
  store_affine<F>(out + (size_t)i * PointFmt<F>::STRIDE, m, flagged);
  if (endo) {
    Fe<F> beta, bx;
    fe_set_const<F>(beta, F::BETA);
    fe_mul(bx, m.x, beta);
    m.x = bx;
    store_affine<F>(out + ((size_t)n + i) * PointFmt<F>::STRIDE, m, flagged);
  }
}

// memory-format records -> canonical affine words (for downloads / tests)
template <class F>
__global__ void __launch_bounds__(256) k_points_from_mont(uint32_t* out, const uint32_t* in, uint32_t n) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Affine<F> p;
  bool inf = load_affine<F>(p, in + (size_t)i * PointFmt<F>::STRIDE, 0);
  uint32_t w[2 * F::NW];
  if (inf) {
#pragma unroll
    for (int j = 0; j < 2 * F::NW; j++) w[j] = 0;
  } else {
    Fe<F> t;
    fe_from_mont(t, p.x);
    fe_to_canon_words<F>(w, t);
    fe_from_mont(t, p.y);
    fe_to_canon_words<F>(w + F::NW, t);
  }
  store_words<F>(out + (size_t)i * 2 * F::NW, w);
}

// ------------------------------------------------------------------------------------------------ digits (fallback sort)
// Used only when the window size leaves more coarse bins than the LDS-staged sort handles (sort_kernels.h): the
// digits are materialized, digits[k*M + i] for i in [0, M) -- M = N (no GLV) or 2N (GLV: entry N+i is the
// endomorphism half) -- and counts[] is the per-bucket histogram (one global atomic per entry).
// `spread` = sb > 0: the top window has few significant bits, so its entries are dealt over 2^sb
// sub-windows K-1 .. K-1+2^sb-1 by the low bits of the point index (every sub-window keeps the weight
// 2^(c(K-1))); this keeps bucket sizes balanced (the job of splitBuckets' special case for the sparse top
// window, msm-common.ts:105-112, 146-174).
constexpr int DIGITS_ITEMS = 8;

template <class Fr, bool GLV>
__global__ void __launch_bounds__(256) k_digits(uint32_t* digits, uint32_t* counts, MsmMeta* meta, const uint32_t* scalars,
                                                uint32_t n, int c, int K, int spread) {
  const uint32_t L = 1u << (c - 1);
  const uint32_t M = GLV ? 2 * n : n;
  const uint32_t smask = (1u << spread) - 1u;
  uint32_t bad = 0;
#pragma unroll 1
  for (int item = 0; item < DIGITS_ITEMS; item++) {
    const uint32_t i = (blockIdx.x * DIGITS_ITEMS + item) * 256 + threadIdx.x;
    if (i >= n) continue;
    uint32_t s[8];
    {
      const uint4* p4 = reinterpret_cast<const uint4*>(scalars + (size_t)i * 8);
      uint4 a = p4[0], b = p4[1];
      s[0] = a.x; s[1] = a.y; s[2] = a.z; s[3] = a.w;
      s[4] = b.x; s[5] = b.y; s[6] = b.z; s[7] = b.w;
    }
    if (words_geq<8>(s, Fr::Q)) bad |= 4u;
    constexpr int HALVES = GLV ? 2 : 1;
    constexpr int HW = GLV ? 4 : 8;
    uint32_t h[HALVES][HW], neg[HALVES];
    if constexpr (GLV) {
      glv_decompose<Fr>(h[0], h[1], neg[0], neg[1], s);
    } else {
#pragma unroll
      for (int j = 0; j < 8; j++) h[0][j] = s[j];
      neg[0] = 0;
    }
#pragma unroll
    for (int half = 0; half < HALVES; half++) {
      uint32_t carry = 0;
      for (int k = 0; k < K; k++) {
        uint32_t l = extract_bits<HW>(h[half], k * c, c) + carry;
        if (l > L) {
          l = 2 * L - l;
          carry = 1;
        } else {
          carry = 0;
        }
        // the half scalar's own sign flips every digit's sign
        const uint32_t ng = (carry ^ neg[half]) & (l != 0 ? 1u : 0u);
        const uint32_t entry = half * n + i;
        digits[(size_t)k * M + entry] = l | (ng << 31);
        if (l != 0) {
          const uint32_t kw = k == K - 1 ? (uint32_t)k + (entry & smask) : (uint32_t)k;
          atomicAdd(&counts[kw * L + (l - 1)], 1u);
        }
      }
      // does the scalar fit K windows?  (a carry out of the last one, or bits beyond it)
      if (carry) bad |= 2u;
      for (int pos = K * c; pos < 32 * HW; pos += 16)
        if (extract_bits<HW>(h[half], pos, 16) != 0) bad |= 2u;
    }
  }
  if (bad) atomicOr(&meta->error, bad);
}

// upload-time range check of resident scalars (scalarsFromBytes, parallel.ts:114-133: values < group order)
template <class Fr>
__global__ void __launch_bounds__(256) k_check_scalars(uint32_t* err, const uint32_t* scalars, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t s[8];
  const uint4* p4 = reinterpret_cast<const uint4*>(scalars + (size_t)i * 8);
  const uint4 a = p4[0], b = p4[1];
  s[0] = a.x; s[1] = a.y; s[2] = a.z; s[3] = a.w;
  s[4] = b.x; s[5] = b.y; s[6] = b.z; s[7] = b.w;
  if (words_geq<8>(s, Fr::Q)) atomicOr(err, 4u);
}

// ------------------------------------------------------------------------------------------------ scans
// Exclusive scan of n values v(g) in three launches; out has n + 1 entries (out[n] = total).
//   mode 0: v(g) = in[g]                                   (bucket sizes -> offsets)
//   mode 1: v(g) = pairs in round r of bucket g, r = blockIdx.y, from bucket offsets `in`
//           pairs_m(s) = floor((s + m - 1) / (2m)), m = 2^r  (number of j with j*2m + m < s)
constexpr int SCAN_T = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = SCAN_T * SCAN_ITEMS;

__device__ __forceinline__ uint32_t scan_value(const uint32_t* in, uint32_t g, uint32_t n, int mode, int r) {
  if (g >= n) return 0;
  if (mode == 0) return in[g];
  uint32_t s = in[g + 1] - in[g];
  if ((mode & 15) == 2) {                // chunks of 2^(mode >> 4) entries (msmBasic accumulation)
    const int sh = mode >> 4;
    return (s + (1u << sh) - 1u) >> sh;
  }
  uint32_t m = 1u << r;
  return (s + m - 1) >> (r + 1);
}

template <int T = 256>
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* total, uint32_t* lds) {
  // T threads: wave-level inclusive scan by shuffles, then across the T/64 waves through LDS
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t x = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t y = __shfl_up(x, d, 64);
    if (lane >= d) x += y;
  }
  if (lane == 63) lds[wave] = x;
  __syncthreads();
  uint32_t base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < T / 64; w++) {
    uint32_t t = lds[w];
    if (w < wave) base += t;
    tot += t;
  }
  __syncthreads();
  *total = tot;
  return base + x - v;
}

static __global__ void __launch_bounds__(SCAN_T) k_scan_partials(uint32_t* partials, const uint32_t* in, uint32_t n, int mode,
                                                          uint32_t nblocks) {
  __shared__ uint32_t lds[SCAN_T / 64];
  const int r = blockIdx.y;
  uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
  uint32_t sum = 0;
#pragma unroll
  for (int j = 0; j < SCAN_ITEMS; j++) sum += scan_value(in, base + j, n, mode, r);
  uint32_t total;
  block_exclusive_scan(sum, &total, lds);
  if (threadIdx.x == 0) partials[(size_t)r * nblocks + blockIdx.x] = total;
}

// one block per round: exclusive scan of the per-tile partials (in place); writes the grand total
static __global__ void __launch_bounds__(SCAN_T) k_scan_top(uint32_t* partials, uint32_t nblocks, uint32_t* totals) {
  __shared__ uint32_t lds[SCAN_T / 64];
  const int r = blockIdx.x;
  uint32_t* p = partials + (size_t)r * nblocks;
  uint32_t running = 0;
  for (uint32_t start = 0; start < nblocks; start += SCAN_T) {
    uint32_t idx = start + threadIdx.x;
    uint32_t v = idx < nblocks ? p[idx] : 0;
    uint32_t total;
    uint32_t ex = block_exclusive_scan(v, &total, lds);
    if (idx < nblocks) p[idx] = running + ex;
    running += total;
  }
  if (threadIdx.x == 0) totals[r] = running;
}

static __global__ void __launch_bounds__(SCAN_T) k_scan_apply(uint32_t* out, const uint32_t* partials, const uint32_t* in,
                                                       uint32_t n, int mode, uint32_t nblocks, size_t out_stride,
                                                       uint32_t* max_out) {
  __shared__ uint32_t lds[SCAN_T / 64];
  const int r = blockIdx.y;
  uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
  uint32_t v[SCAN_ITEMS], sum = 0, mx = 0;
#pragma unroll
  for (int j = 0; j < SCAN_ITEMS; j++) {
    v[j] = scan_value(in, base + j, n, mode, r);
    sum += v[j];
    mx = v[j] > mx ? v[j] : mx;
  }
  uint32_t total;
  uint32_t ex = block_exclusive_scan(sum, &total, lds) + partials[(size_t)r * nblocks + blockIdx.x];
  uint32_t* o = out + (size_t)r * out_stride;
#pragma unroll
  for (int j = 0; j < SCAN_ITEMS; j++) {
    if (base + j < n) o[base + j] = ex;
    ex += v[j];
    if (base + j + 1 == n) o[n] = ex;
  }
  if (max_out != nullptr && mx != 0) atomicMax(max_out, mx);
}

// ------------------------------------------------------------------------------------------------ scatter
// refs[off[g] + (arrival order within bucket g)] = i | negate<<31  for every non-zero digit.
// (This is the HBM-bound "bucket scatter": algorithmic bytes = 4 B digit read + 4 B reference write
// per entry)
// `n_half` / `endo_delta`: with GLV the entry index i >= n_half is the endomorphism half of point i - n_half; its
// record sits at index i + endo_delta of the point set (the images follow the whole set, which may be larger than
// the prefix this MSM covers: msm-batched-affine.ts:74-97 takes any N <= allocated).
static __global__ void __launch_bounds__(256) k_scatter(uint32_t* refs, uint32_t* cursor, const uint32_t* off,
                                                 const uint32_t* digits, uint32_t M, int c, int spread,
                                                 uint32_t n_half, uint32_t endo_delta) {
  constexpr int ITEMS = 4;
  const uint32_t L = 1u << (c - 1);
  const uint32_t k = blockIdx.y;
  const uint32_t* dk = digits + (size_t)k * M;
  const uint32_t base = blockIdx.x * (256 * ITEMS) + threadIdx.x;
#pragma unroll
  for (int j = 0; j < ITEMS; j++) {
    uint32_t i = base + j * 256;
    if (i >= M) break;
    uint32_t d = dk[i];
    uint32_t l = d & REF_IDX;
    if (l == 0) continue;
    const uint32_t kw = (k + 1 == gridDim.y) ? k + (i & ((1u << spread) - 1u)) : k;
    uint32_t g = kw * L + (l - 1);
    uint32_t pos = off[g] + atomicAdd(&cursor[g], 1u);
    refs[pos] = (i >= n_half ? i + endo_delta : i) | (d & REF_NEG);
  }
}

}  // namespace msmz

#include "sort_kernels.h"
#include "plan_kernels.h"

namespace msmz {

// ------------------------------------------------------------------------------------------------ operands
// A location word (plan_kernels.h) names an operand of a tree round / a partial bucket sum: an original point
// (bit 30; bit 31 = negate; record in the resident point set) or a slot record.  With packed slot records both are
// 2*NW saturated words and differ only in where their 16-byte chunks sit (contiguous / chunk-interleaved), so an
// operand is loaded WITHOUT a branch on its kind: all loads of a pair are issued back to back and cost one memory
// latency, not one per operand.
template <class F>
__device__ __forceinline__ const uint32_t* operand_address(uint32_t loc, const uint32_t* slots, const uint32_t* points,
                                                            int& cs, uint32_t& neg) {
  const bool orig = (loc & LOC_ORIG) != 0;
  cs = orig ? 1 : SLOT_CS;
  neg = orig ? loc >> 31 : 0u;
  return orig ? points + (size_t)(loc & 0x3fffffffu) * PointFmt<F>::STRIDE : slots + slot_offset<F>(loc);
}

template <class F, bool CHECK_INF>
__device__ __forceinline__ bool load_operand(Affine<F>& p, uint32_t loc, const uint32_t* slots, const uint32_t* points) {
  if constexpr (SlotFmt<F>::PACKED) {
    int cs;
    uint32_t neg;
    const uint32_t* rec = operand_address<F>(loc, slots, points, cs, neg);
    uint32_t w[2 * F::NW];
    load_words<F>(w, rec, cs);
    uint32_t o = 0;
    if (CHECK_INF) {
#pragma unroll
      for (int i = 0; i < 2 * F::NW; i++) o |= w[i];
    }
    fe_unpack<F>(p.x, w);
    Fe<F> y;
    fe_unpack<F>(y, w + F::NW);
    fe_cneg(p.y, y, neg);
    return CHECK_INF && o == 0;
  } else {
    if (loc & LOC_ORIG) {
      uint32_t w[2 * F::NW];
      load_words<F>(w, points + (size_t)(loc & 0x3fffffffu) * PointFmt<F>::STRIDE);
      uint32_t o = 0;
      if (CHECK_INF) {
#pragma unroll
        for (int i = 0; i < 2 * F::NW; i++) o |= w[i];
      }
      fe_unpack<F>(p.x, w);
      Fe<F> y;
      fe_unpack<F>(y, w + F::NW);
      fe_cneg(p.y, y, loc >> 31);
      return CHECK_INF && o == 0;
    }
    return slot_load_point<F, CHECK_INF>(p, slots + slot_offset<F>(loc));
  }
}
// x coordinate only, packed records; returns the OR of its words (zero: the record may be the all-zero infinity record)
template <class F>
__device__ __forceinline__ uint32_t load_operand_x_or(Fe<F>& x, uint32_t loc, const uint32_t* slots,
                                                      const uint32_t* points) {
  static_assert(SlotFmt<F>::PACKED, "packed slot records");
  int cs;
  uint32_t neg;
  const u32x4* s4 = reinterpret_cast<const u32x4*>(operand_address<F>(loc, slots, points, cs, neg));
  uint32_t w[F::NW];
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < F::NW / 4; i++) {
    const u32x4 v = s4[i * cs];
    w[4 * i] = v.x;
    w[4 * i + 1] = v.y;
    w[4 * i + 2] = v.z;
    w[4 * i + 3] = v.w;
    o |= v.x | v.y | v.z | v.w;
  }
  fe_unpack<F>(x, w);
  return o;
}

// x coordinate only
template <class F>
__device__ __forceinline__ void load_operand_x(Fe<F>& x, uint32_t loc, const uint32_t* slots, const uint32_t* points) {
  if constexpr (SlotFmt<F>::PACKED) {
    int cs;
    uint32_t neg;
    const u32x4* s4 = reinterpret_cast<const u32x4*>(operand_address<F>(loc, slots, points, cs, neg));
    uint32_t w[F::NW];
#pragma unroll
    for (int i = 0; i < F::NW / 4; i++) {
      const u32x4 v = s4[i * cs];
      w[4 * i] = v.x;
      w[4 * i + 1] = v.y;
      w[4 * i + 2] = v.z;
      w[4 * i + 3] = v.w;
    }
    fe_unpack<F>(x, w);
  } else {
    if (loc & LOC_ORIG) {
      uint32_t w[F::NW];
      const u32x4* s4 = reinterpret_cast<const u32x4*>(points + (size_t)(loc & 0x3fffffffu) * PointFmt<F>::STRIDE);
#pragma unroll
      for (int i = 0; i < F::NW / 4; i++) {
        const u32x4 v = s4[i];
        w[4 * i] = v.x;
        w[4 * i + 1] = v.y;
        w[4 * i + 2] = v.z;
        w[4 * i + 3] = v.w;
      }
      fe_unpack<F>(x, w);
    } else {
      slot_load_fe<F>(x, slots + slot_offset<F>(loc));
    }
  }
}

// ------------------------------------------------------------------------------------------------ wave-wide inversion
// fe_inverse (fp.h) for a wave-uniform input, with the four vectors of the binary-GCD state spread over the lanes:
// DPP row 0..3 = a, b, u, v, lane j of the row = limb j (N <= 14 of 16 lanes).  The W binary steps on the
// 64-bit approximations stay on the scalar unit (they are wave-uniform); the matrix application
//   (a, b) <- (a f0 + b g0, a f1 + b g1) / 2^W,   (u, v) <- the same mod p (one Montgomery column step)
// becomes two multiply-adds per lane, and the carry chain a 13-step ripple of `row_shr:1` DPP moves shared by all
// four rows.  Same algorithm and invariants as fe_inverse; ~4x shorter latency than running it limb by limb on
// the scalar unit (it is on the critical path of every batch: forward pass -> product tree -> inversion -> back).
template <int W, int N>
__device__ __forceinline__ int32_t limb_row_ripple(int32_t val, int j) {
  constexpr int32_t MASK = (1 << W) - 1;
  const bool low = j < N - 1;          // limbs below the (signed) top limb are reduced to [0, 2^W)
  // a carry moves up one limb per pass; after the second pass a further carry needs a limb within 4 of 2^W,
  // so the loop almost always ends after two or three passes (and always within N - 1)
#pragma unroll 1
  for (int pass = 0; pass < N - 1; pass++) {
    const int32_t c = low ? (val >> W) : 0;
    val = low ? (val & MASK) : val;
    val += __builtin_amdgcn_update_dpp(0, c, 0x111, 0xf, 0xf, true);   // row_shr:1: lane j takes lane j-1's carry
    if (__ballot(low && (uint32_t)val > (uint32_t)MASK) == 0) break;
  }
  return val;
}

template <class F>
__device__ __forceinline__ bool fe_inverse_wave(Fe<F>& r, const Fe<F>& x) {
  constexpr int N = F::N, W = F::W;
  constexpr uint32_t MASK = (1u << W) - 1u;
  static_assert(N <= 15, "one DPP row per vector");
  const int lane = (int)(threadIdx.x & 63u);
  const int row = lane >> 4, j = lane & 15;
  int32_t xl = 0, pl = 0, npl = 0;
  {
    Fe<F> xc;
    uint32_t w[F::NW];
    fe_to_canon_words<F>(w, x);
    fe_unpack<F>(xc, w);
#pragma unroll
    for (int k = 0; k < N; k++) {
      if (j == k) {
        xl = xc.l[k];
        pl = F::PL[k];
        npl = F::NPL[k];
      }
    }
  }
  int32_t val = row == 0 ? xl : row == 1 ? pl : (row == 2 && j == 0) ? 1 : 0;
  constexpr int ITERS = (2 * F::BITS + W - 1) / W + 1;
#pragma unroll 1
  for (int it = 0; it < ITERS; it++) {
    const uint64_t nz = __ballot(val != 0);
    const uint32_t ma = (uint32_t)nz & 0xffffu, mb = (uint32_t)(nz >> 16) & 0xffffu;
    if (ma == 0) break;   // a == 0: converged
    const int h = 31 - __builtin_clz(ma | mb | 1u);   // highest limb where a or b is non-zero
    const uint32_t a0 = (uint32_t)__builtin_amdgcn_readlane(val, 0), b0 = (uint32_t)__builtin_amdgcn_readlane(val, 16);
    uint64_t xa, xb;
    if (h == 0) {
      xa = a0;
      xb = b0;
    } else {
      const uint64_t ah = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane(val, h) << W) |
                          (uint32_t)__builtin_amdgcn_readlane(val, h - 1);
      const uint64_t bh = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane(val, 16 + h) << W) |
                          (uint32_t)__builtin_amdgcn_readlane(val, 16 + h - 1);
      if (h == 1) {
        xa = ah;
        xb = bh;
      } else {
        const uint64_t mx = ah | bh;
        const int len = 64 - __builtin_clzll(mx | 1);
        const int sh = len > (W + 1) ? len - (W + 1) : 0;
        xa = ((ah >> sh) << W) | (uint64_t)(a0 & MASK);
        xb = ((bh >> sh) << W) | (uint64_t)(b0 & MASK);
      }
    }
    // ---- W binary steps on the approximations (runs of trailing zeros retired at once)
    int32_t f0 = 1, g0 = 0, f1 = 0, g1 = 1;
    int rem = W;
    while (true) {
      int tz = xa == 0 ? rem : __builtin_ctzll(xa);
      if (tz > rem) tz = rem;
      xa >>= tz;
      f1 <<= tz;
      g1 <<= tz;
      rem -= tz;
      if (rem == 0) break;
      if (xa < xb) {
        const uint64_t tx = xa; xa = xb; xb = tx;
        int32_t ti = f0; f0 = f1; f1 = ti;
        ti = g0; g0 = g1; g1 = ti;
      }
      xa -= xb;
      f0 -= f1;
      g0 -= g1;
    }
    // ---- apply the matrix: own vector times Fc, partner vector (row ^ 1) times Gc
    const int32_t Fc = (row & 1) ? g1 : f0, Gc = (row & 1) ? f1 : g0;
    const int32_t partner = __shfl_xor(val, 16, 64);
    int64_t t = (int64_t)val * Fc + (int64_t)partner * Gc;
    uint32_t qu = (uint32_t)__builtin_amdgcn_readlane((int32_t)(uint32_t)t, 32);
    uint32_t qv = (uint32_t)__builtin_amdgcn_readlane((int32_t)(uint32_t)t, 48);
    if (F::PINV != 1u) {
      qu *= F::PINV;
      qv *= F::PINV;
    }
    const int32_t q = row == 2 ? (int32_t)(qu & MASK) : row == 3 ? (int32_t)(qv & MASK) : 0;
    t += (int64_t)q * npl;
    // divide by 2^W: limb i <- low part of limb i+1 plus the carry of limb i (limb 0's low part is 0)
    const int32_t lo = (int32_t)((uint32_t)t & MASK);
    const int32_t hi = (int32_t)(t >> W);
    val = __builtin_amdgcn_update_dpp(0, lo, 0x101, 0xf, 0xf, true) + hi;   // row_shl:1: lane j takes lane j+1
    if (j >= N) val = 0;
    val = limb_row_ripple<W, N>(val, j);
    // u, v in (-2p, p): add p when negative
    const int32_t top_u = __builtin_amdgcn_readlane(val, 32 + N - 1), top_v = __builtin_amdgcn_readlane(val, 48 + N - 1);
    if (row == 2 && top_u < 0) val += pl;
    if (row == 3 && top_v < 0) val += pl;
    // a, b back to non-negative (negating u, v along)
    const int32_t top_a = __builtin_amdgcn_readlane(val, N - 1), top_b = __builtin_amdgcn_readlane(val, 16 + N - 1);
    if ((top_a | top_b) < 0) {
      const bool neg = (row & 1) ? (top_b < 0) : (top_a < 0);
      val = neg ? -val : val;
      val = limb_row_ripple<W, N>(val, j);
    }
  }
  // gcd ends up in b: must be 1
  const uint64_t bad = __ballot(row == 1 && val != (j == 0 ? 1 : 0));
  if (bad != 0) {
    fe_zero(r);
    return false;
  }
  Fe<F> v, r3;
#pragma unroll
  for (int k = 0; k < N; k++) v.l[k] = __builtin_amdgcn_readlane(val, 48 + k);
  fe_carry(v);
  fe_set_const<F>(r3, F::R3);
  fe_mul<F>(r, v, r3);
  return true;
}

}  // namespace msmz

#include "batch_kernels.h"

namespace msmz {

// ------------------------------------------------------------------------------------------------ 4-lane point addition
// The upper reduction levels are latency-bound (a handful of waves, each alone on its SIMD, running 14
// dependent field products per XYZZ addition).  There one addition is spread over the 4 lanes of a DPP
// quad instead: every lane holds the same operands, each computes one of the (up to) four independent
// products of a dependency level, and the products are exchanged with quad_perm DPP moves (no LDS) --
// depth 4 products instead of 14 (twisted Edwards: 3 instead of 9).
template <class F>
__device__ __forceinline__ void fe_sel4(Fe<F>& r, int s, const Fe<F>& a0, const Fe<F>& a1, const Fe<F>& a2,
                                        const Fe<F>& a3) {
#pragma unroll
  for (int j = 0; j < F::N; j++) {
    int32_t lo = (s & 1) ? a1.l[j] : a0.l[j];
    int32_t hi = (s & 1) ? a3.l[j] : a2.l[j];
    r.l[j] = (s & 2) ? hi : lo;
  }
}

template <int J, class F>
__device__ __forceinline__ void fe_quad_bcast(Fe<F>& r, const Fe<F>& a) {   // r = a of sub-lane J of the quad
#pragma unroll
  for (int j = 0; j < F::N; j++) r.l[j] = __builtin_amdgcn_update_dpp(0, a.l[j], J * 0x55, 0xf, 0xf, false);
}

template <class F>
__device__ __forceinline__ void fe_pick(Fe<F>& r, bool c, const Fe<F>& a, const Fe<F>& b) {   // r = c ? a : b
#pragma unroll
  for (int j = 0; j < F::N; j++) r.l[j] = c ? a.l[j] : b.l[j];
}

// r = p + q, or (dbl, uniform over the quad, q == p) r = 2p, in four product levels of <= 4 independent products:
//   level   addition (add-2008-s)                         doubling (dbl-2008-s-1, a = 0)
//     1     U1 = X1 ZZ2, U2 = X2 ZZ1, S1 = Y1 ZZZ2, S2 = Y2 ZZZ1      -
//     2     PP = P^2, RR = R^2, ZZ1 ZZ2, ZZZ1 ZZZ2       V = U^2, XX = X^2            (U = 2Y, M = 3 XX)
//     3     PPP = P PP, Q = U1 PP, ZZ3 = (ZZ1 ZZ2) PP    W = U V, S = X V, M^2, ZZ3 = ZZ V
//     4     R (Q - X3), S1 PPP, ZZZ3 = (ZZZ1 ZZZ2) PPP   M (S - X3), Y W, ZZZ3 = ZZZ W
// The reduction levels double on purpose (2a, 2 row, 4 row); as the equal-operands edge case of the addition those
// doublings ran the sequential 9-product xyzz_dbl for the whole wave in three of the four steps of a level.
template <class F>
__device__ __forceinline__ void xyzz_add_x4(Xyzz<F>& r, const Xyzz<F>& p, const Xyzz<F>& q, int s, bool dbl) {
  const bool pinf = xyzz_is_inf(p), qinf = xyzz_is_inf(q);
  Fe<F> a, b, m, U1, U2, S1, S2, P, R, PP, RR, Zm, Zc, PPP, Q, t, u, U, M, zero;
  fe_zero(zero);
  fe_sel4(a, s, p.X, q.X, p.Y, q.Y);
  fe_sel4(b, s, q.ZZ, p.ZZ, q.ZZZ, p.ZZZ);
  fe_mul(m, a, b);
  fe_quad_bcast<0>(U1, m);
  fe_quad_bcast<1>(U2, m);
  fe_quad_bcast<2>(S1, m);
  fe_quad_bcast<3>(S2, m);
  fe_sub(P, U2, U1);
  fe_sub(R, S2, S1);
  fe_add(U, p.Y, p.Y);
  fe_carry(U);
  // level 2
  fe_sel4(a, s, P, R, p.ZZ, p.ZZZ);
  fe_sel4(b, s, P, R, q.ZZ, q.ZZZ);
  fe_pick(t, (s & 1) != 0, p.X, U);
  fe_pick(a, dbl, t, a);
  fe_pick(b, dbl, t, b);
  fe_mul(m, a, b);
  fe_quad_bcast<0>(PP, m);    // doubling: V
  fe_quad_bcast<1>(RR, m);    // doubling: XX
  fe_quad_bcast<2>(Zm, m);
  fe_quad_bcast<3>(Zc, m);
  fe_add(M, RR, RR);
  fe_add(M, M, RR);
  fe_carry(M);
  // level 3
  fe_sel4(a, s, P, U1, Zm, Zm);
  fe_sel4(t, s, U, p.X, M, p.ZZ);
  fe_pick(a, dbl, t, a);
  fe_pick(b, dbl && s == 2, M, PP);
  fe_mul(m, a, b);
  fe_quad_bcast<0>(PPP, m);   // doubling: W
  fe_quad_bcast<1>(Q, m);     // doubling: S
  fe_quad_bcast<2>(t, m);     // addition: ZZ3; doubling: M^2
  fe_quad_bcast<3>(u, m);     // doubling: ZZ3
  fe_pick(r.ZZ, dbl, u, t);
  fe_pick(RR, dbl, t, RR);
  fe_pick(u, dbl, zero, PPP);
  fe_sub(t, RR, u);
  fe_sub(t, t, Q);
  fe_sub(u, t, Q);       // X3
  fe_carry(u);
  fe_sub(t, Q, u);
  fe_carry(t);
  // level 4
  fe_sel4(a, s, R, S1, Zc, Zc);
  fe_sel4(m, s, M, p.Y, p.ZZZ, p.ZZZ);
  fe_pick(a, dbl, m, a);
  fe_pick(b, s == 0, t, PPP);
  fe_mul(m, a, b);
  fe_quad_bcast<0>(Q, m);
  fe_quad_bcast<1>(t, m);
  fe_quad_bcast<2>(r.ZZZ, m);
  fe_sub(r.Y, Q, t);
  fe_carry(r.Y);
  r.X = u;
  // edge cases exactly as xyzz_add / xyzz_dbl (uniform over the quad: all four lanes hold the same operands)
  if (pinf) {
    r = q;
  } else if (qinf) {
    r = p;
  } else if (!dbl && fe_is_zero(P)) {
    if (fe_is_zero(R)) xyzz_dbl(r, p); else xyzz_set_inf(r);
  }
}

template <class F>
__device__ __forceinline__ void te_add_x4(TeExt<F>& r, const TeExt<F>& p, const TeExt<F>& q, int s, bool) {   // the unified addition doubles as well
  Fe<F> a, b, m, A, B, C, D, E, Fv, G, H, t, u, v, w, k;
  fe_sub(t, p.Y, p.X);
  fe_sub(u, q.Y, q.X);
  fe_carry(t);
  fe_carry(u);
  fe_add(v, p.Y, p.X);
  fe_add(w, q.Y, q.X);
  fe_carry(v);
  fe_carry(w);
  fe_sel4(a, s, t, v, p.T, p.Z);
  fe_sel4(b, s, u, w, q.T, q.Z);
  fe_mul(m, a, b);
  fe_quad_bcast<0>(A, m);
  fe_quad_bcast<1>(B, m);
  fe_quad_bcast<2>(t, m);
  fe_quad_bcast<3>(D, m);
  fe_set_const<F>(k, F::K2D);
  fe_mul(C, t, k);
  fe_add(D, D, D);
  fe_sub(E, B, A);
  fe_sub(Fv, D, C);
  fe_add(G, D, C);
  fe_add(H, B, A);
  fe_carry(E);
  fe_carry(Fv);
  fe_carry(G);
  fe_carry(H);
  fe_sel4(a, s, E, G, E, Fv);
  fe_sel4(b, s, Fv, H, H, G);
  fe_mul(m, a, b);
  fe_quad_bcast<0>(r.X, m);
  fe_quad_bcast<1>(r.Y, m);
  fe_quad_bcast<2>(r.T, m);
  fe_quad_bcast<3>(r.Z, m);
}

// ------------------------------------------------------------------------------------------------ group policies
// The bucket accumulation of the msmBasic path and the bucket reduction are written once over a small
// "group policy": accumulator type + how to fold an input point record into it.
//   WeierPolicy : XYZZ accumulators, inputs = affine records [x | y] (2*NW words)
//   TePolicy    : extended twisted-Edwards accumulators, inputs = Niels records [y-x | y+x | 2dxy] (3*NW words)
template <class F_>
struct WeierPolicy {
  using F = F_;
  using Acc = Xyzz<F>;
  static constexpr int IN_WORDS = PointFmt<F>::STRIDE;   // words between the records of a resident point set
  static constexpr int ACC_WORDS = 4 * F::NW;
  static constexpr int ACC_OCC = 1;   // k_bucket_accumulate: no register cap (the XYZZ mixed addition spills below ~120)
  static __device__ __forceinline__ void zero(Acc& a) { xyzz_set_inf(a); }
  static __device__ __forceinline__ void add(Acc& r, const Acc& a, const Acc& b) { xyzz_add(r, a, b); }
  static __device__ __forceinline__ void dbl(Acc& r, const Acc& a) { xyzz_dbl(r, a); }
  static __device__ __forceinline__ void add_x4(Acc& r, const Acc& a, const Acc& b, int s, bool dbl) { xyzz_add_x4(r, a, b, s, dbl); }
  static __device__ __forceinline__ void madd(Acc& r, const Acc& a, const uint32_t* rec, uint32_t neg) {
    Affine<F> p;
    bool inf = load_affine<F>(p, rec, neg);
    xyzz_madd(r, a, p, inf);
  }
  static __device__ __forceinline__ void load(Acc& a, const uint32_t* rec) { load_xyzz<F>(a, rec); }
  static __device__ __forceinline__ void store(uint32_t* rec, const Acc& a) { store_xyzz<F>(rec, a); }
};

template <class F>
__device__ __forceinline__ void load_fe4(Fe<F>& a, Fe<F>& b, Fe<F>& c, Fe<F>& d, const uint32_t* rec) {
  uint32_t w[2 * F::NW];
  load_words<F>(w, rec);
  fe_unpack<F>(a, w);
  fe_unpack<F>(b, w + F::NW);
  load_words<F>(w, rec + 2 * F::NW);
  fe_unpack<F>(c, w);
  fe_unpack<F>(d, w + F::NW);
}

template <class F_>
struct TePolicy {
  using F = F_;
  using Acc = TeExt<F>;
  static constexpr int IN_WORDS = 4 * F::NW;   // Niels record padded to 4 field elements (16-byte aligned loads)
  static constexpr int ACC_WORDS = 4 * F::NW;
  // k_bucket_accumulate: no register cap (a cap for 5 waves per SIMD -- 91 registers, no spills -- measured 1 % slower
  // in a same-box A/B at 2^24, 6 waves spill)
  static constexpr int ACC_OCC = 1;
  static __device__ __forceinline__ void zero(Acc& a) { te_set_zero(a); }
  static __device__ __forceinline__ void add(Acc& r, const Acc& a, const Acc& b) { te_add(r, a, b); }
  static __device__ __forceinline__ void dbl(Acc& r, const Acc& a) { te_add(r, a, a); }
  static __device__ __forceinline__ void add_x4(Acc& r, const Acc& a, const Acc& b, int s, bool dbl) { te_add_x4(r, a, b, s, dbl); }
  static __device__ __forceinline__ void madd(Acc& r, const Acc& a, const uint32_t* rec, uint32_t neg) {
    TeNiels<F> n;
    Fe<F> pad;
    load_fe4<F>(n.ym, n.yp, n.kt, pad, rec);
    te_madd(r, a, n, neg);
  }
  static __device__ __forceinline__ void load(Acc& a, const uint32_t* rec) { load_fe4<F>(a.X, a.Y, a.Z, a.T, rec); }
  static __device__ __forceinline__ void store(uint32_t* rec, const Acc& a) {
    uint32_t w[2 * F::NW];
    fe_store<F>(w, a.X);
    fe_store<F>(w + F::NW, a.Y);
    store_words<F>(rec, w);
    fe_store<F>(w, a.Z);
    fe_store<F>(w + F::NW, a.T);
    store_words<F>(rec + 2 * F::NW, w);
  }
};

// ------------------------------------------------------------------------------------------------ msmBasic accumulation
// Bucket accumulation without batch inversion (msm-basic.ts:106-128: addMixed / subMixed into projective
// or extended buckets).  The sorted reference list of every bucket is cut into chunks of CH entries
// (cscan = exclusive scan of ceil(size / CH), scan mode 2); one thread folds one chunk into an accumulator.
// CH = 64 normally; for very long buckets (adversarial scalars) CH ~ sqrt(longest bucket), so that neither a chunk
// nor the list of a bucket's partial sums (added up by one thread of the first reduction level) gets long.
#ifndef MSMZ_REDUCE_OCC
#define MSMZ_REDUCE_OCC 2   // waves per SIMD the register budget of the big reduction kernels is capped for
#endif
constexpr int ACC_CHUNK_SHIFT = 5;   // normal chunk = 32 entries (64 measured 4 % slower on Pallas 2^22); the host raises it to ~sqrt(longest bucket)

template <class P>
__global__ void __launch_bounds__(128, P::ACC_OCC) k_bucket_accumulate(uint32_t* partial, const uint32_t* points,
                                                           const uint32_t* refs, const uint32_t* off,
                                                           const uint32_t* cscan, uint32_t nb, uint32_t n_chunks,
                                                           int chunk_shift) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_chunks) return;
  uint32_t lo = 0, hi = nb;   // cscan[lo] <= t < cscan[hi]
  while (hi - lo > 1) {
    uint32_t mid = (lo + hi) >> 1;
    if (cscan[mid] <= t) lo = mid; else hi = mid;
  }
  const uint32_t start = off[lo] + ((t - cscan[lo]) << chunk_shift);
  const uint32_t end = min(start + (1u << chunk_shift), off[lo + 1]);
  typename P::Acc acc, tmp;
  P::zero(acc);
  for (uint32_t p = start; p < end; p++) {
    const uint32_t rf = refs[p];
    P::madd(tmp, acc, points + (size_t)(rf & REF_IDX) * P::IN_WORDS, rf >> 31);
    acc = tmp;
  }
  P::store(partial + (size_t)t * P::ACC_WORDS, acc);
}

// ------------------------------------------------------------------------------------------------ reduce
// run += (sum of bucket g): the <= 4 partial sums the tree rounds left of it, at the locations the plan recorded
// (one for all but the longest buckets: the rounds stop two short of log2(longest bucket), because a round costs
// ~80 us of latency however few pairs it has).
template <class F>
__device__ __forceinline__ void add_bucket(Xyzz<F>& run, uint32_t g, const uint32_t* slots, const uint32_t* points,
                                           const uint4* bfin) {
  const uint4 fin = bfin[g];
  const uint32_t locs[4] = {fin.x, fin.y, fin.z, fin.w};
#pragma unroll 1
  for (int i = 0; i < 4; i++) {
    const uint32_t loc = locs[i];
    if (loc == LOC_NONE) break;
    Affine<F> p;
    const bool inf = load_operand<F, true>(p, loc, slots, points);
    Xyzz<F> tmp;
    xyzz_madd(tmp, run, p, inf);
    run = tmp;
  }
}

// Bucket reduction  W_k = sum_{l=1..L} l * B_l  (msm-batched-affine.ts:544-571) by grouped running sums.
// Elements are indexed by their weight j = l in [0, L) (element 0 is empty; the one bucket of weight L is
// folded into element L/2 twice), cut into groups of S = 2^s:
//   row_a = sum_b E[aS + b],   tri_a = sum_b b * E[aS + b]          (running-sum trick, :556-559)
//   sum_j j * E_j = sum_a tri_a + sum_a a * (S * row_a)
// so the next level runs the same computation on the *scaled* rows S*row_a (s doublings per group) and
// simply adds up the tri's:  C'_A = sum_b C[AS + b] + tri'_A.  After the last level (one entry per
// window) C is W_k.  No per-level power-of-two scaling of the partial sums is needed.
template <class F>
__global__ void __launch_bounds__(128, MSMZ_REDUCE_OCC) k_reduce_first(uint32_t* rows, uint32_t* tris, const uint32_t* slots,
                                                      const uint32_t* points, const uint4* bfin, uint32_t L, uint32_t S,
                                                      uint32_t groups, uint32_t total) {
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  uint32_t k = t / groups, a = t - k * groups;
  Xyzz<F> run, tri, tmp;
  xyzz_set_inf(run);
  xyzz_set_inf(tri);
  for (uint32_t b = S; b-- > 0;) {
    const uint32_t j = a * S + b;           // weight; bucket l = j, j in [0, L)
    if (j >= 1 && j < L) add_bucket<F>(run, k * L + (j - 1), slots, points, bfin);
    if (j == L / 2 && L >= 2) {
      // the single bucket of weight L is folded in as 2 * (L/2): keeps the element count a power of two
      for (int twice = 0; twice < 2; twice++)
        add_bucket<F>(run, k * L + (L - 1), slots, points, bfin);
    }
    if (b >= 1) {
      xyzz_add(tmp, tri, run);
      tri = tmp;
    }
  }
  for (uint32_t s = S; s > 1; s >>= 1) {
    xyzz_dbl(tmp, run);
    run = tmp;
  }
  store_xyzz<F>(rows + (size_t)t * 4 * F::NW, run);
  store_xyzz<F>(tris + (size_t)t * 4 * F::NW, tri);
}

// Level >= 2 on accumulator inputs (n_in entries per window), same recurrence.
// With c_in == nullptr this is the FIRST level of the msmBasic path: element j of window k (j in [0, L],
// n_in = L + 1) is then the sum of the partial accumulators rows_in[cscan[g] .. cscan[g+1]) of bucket
// g = k*L + j - 1  (cscan != nullptr), element 0 is empty and bucket L is folded into element L/2 twice (n_in = L).
template <class P>
__global__ void __launch_bounds__(128) k_reduce_next(uint32_t* rows_out, uint32_t* c_out, const uint32_t* rows_in,
                                                     const uint32_t* c_in, const uint32_t* cscan, uint32_t n_in,
                                                     uint32_t S, uint32_t groups, uint32_t total, uint32_t L) {
  constexpr int XW = P::ACC_WORDS;
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  uint32_t k = t / groups, A = t - k * groups;
  size_t base = (size_t)k * n_in + (size_t)A * S;
  typename P::Acc run, tri, cs, tmp, p;
  P::zero(run);
  P::zero(tri);
  P::zero(cs);
  for (uint32_t b = S; b-- > 0;) {
    const uint32_t e = A * S + b;
    if (e >= n_in) continue;
    if (cscan != nullptr) {
      if (e >= 1) {
        const size_t g = (size_t)k * L + (e - 1);
        for (uint32_t q = cscan[g]; q < cscan[g + 1]; q++) {
          P::load(p, rows_in + (size_t)q * XW);
          P::add(tmp, run, p);
          run = tmp;
        }
      }
      if (e == L / 2 && L >= 2) {   // weight-L bucket folded in as 2 * (L/2)
        const size_t g = (size_t)k * L + (L - 1);
        for (int twice = 0; twice < 2; twice++)
          for (uint32_t q = cscan[g]; q < cscan[g + 1]; q++) {
            P::load(p, rows_in + (size_t)q * XW);
            P::add(tmp, run, p);
            run = tmp;
          }
      }
    } else {
      P::load(p, c_in + (base + b) * XW);
      P::add(tmp, cs, p);
      cs = tmp;
      P::load(p, rows_in + (base + b) * XW);
      P::add(tmp, run, p);
      run = tmp;
    }
    if (b >= 1) {
      P::add(tmp, tri, run);
      tri = tmp;
    }
  }
  for (uint32_t s = S; s > 1; s >>= 1) {
    P::dbl(tmp, run);
    run = tmp;
  }
  P::add(tmp, cs, tri);
  P::store(rows_out + (size_t)t * XW, run);
  P::store(c_out + (size_t)t * XW, tmp);
}

// Same recurrence with S = 4, one group per QUAD of lanes: the 9 accumulator additions + 2 doublings of a
// group form a dependency graph of depth 4, so four lanes finish a group in 4 sequential point
// operations instead of 14 (the upper reduction levels are latency-bound: few groups, long chains).
//   step 1: L0 r0+r1        L1 b = r1+r3     L2 a = r2+r3     L3 c2+c3
//   step 2: L0 row = s01+a   L1 c0+c1         L2 2a            -
//   step 3: L0 2 row         L1 cs = c01+c23  L2 tri = b+2a    -
//   step 4: L0 4 row         L1 C' = cs+tri
// Operands travel between the lanes of a quad with ds_bpermute (lane shuffles of the limbs).
template <class P>
__device__ __forceinline__ void quad_fetch(typename P::Acc& got, const typename P::Acc& pub, int src_lane) {
  constexpr int NWORDS = sizeof(typename P::Acc) / 4;
  const int32_t* in = reinterpret_cast<const int32_t*>(&pub);
  int32_t* out = reinterpret_cast<int32_t*>(&got);
#pragma unroll
  for (int j = 0; j < NWORDS; j++) out[j] = __shfl(in[j], src_lane, 64);
}

template <class P>
__global__ void __launch_bounds__(64, MSMZ_REDUCE_OCC) k_reduce_quad(uint32_t* rows_out, uint32_t* c_out, const uint32_t* rows_in,
                                                    const uint32_t* c_in, uint32_t n_in, uint32_t groups,
                                                    uint32_t total) {
  constexpr int XW = P::ACC_WORDS;
  using Acc = typename P::Acc;
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t grp = t >> 2, q = t & 3;
  const bool live = grp < total;
  const uint32_t k = live ? grp / groups : 0, A = live ? grp - k * groups : 0;
  const uint32_t e = A * 4 + q;
  const int lane = threadIdx.x & 63, base_lane = lane & ~3;
  Acc r, c, v, w, got;
  P::zero(r);
  P::zero(c);
  if (live && e < n_in) {
    P::load(r, rows_in + ((size_t)k * n_in + e) * XW);
    P::load(c, c_in + ((size_t)k * n_in + e) * XW);
  }
  // step 1: pub = {-, r, c, r}; src = {1, 3, 3, 2}
  {
    const int src[4] = {1, 3, 3, 2};
    quad_fetch<P>(got, q == 2 ? c : r, base_lane + src[q]);
    P::add(v, q == 3 ? c : r, got);     // L0 s01, L1 b, L2 a, L3 c23
  }
  // step 2: pub = {c0, -, a, -}; src = {2, 0, 2, 3}
  {
    const int src[4] = {2, 0, 2, 3};
    quad_fetch<P>(got, q == 0 ? c : v, base_lane + src[q]);
    if (q == 3) P::zero(got);
    Acc lhs = (q == 1) ? c : v;
    if (q == 3) P::zero(lhs);
    P::add(w, lhs, got);                // L0 row, L1 c01, L2 2a, L3 0
  }
  // step 3: pub = {-, b, -, c23} (the step-1 results v); src = {0, 3, 1, 3}
  {
    const int src[4] = {0, 3, 1, 3};
    quad_fetch<P>(got, q == 0 ? w : v, base_lane + src[q]);
    P::add(r, w, got);                  // L0 2 row, L1 cs, L2 tri     (r is free now)
  }
  // step 4: pub = step-3 results; src = {0, 2, 2, 3}
  {
    const int src[4] = {0, 2, 2, 3};
    quad_fetch<P>(got, r, base_lane + src[q]);
    P::add(c, r, got);                  // L0 4 row, L1 C' = cs + tri
  }
  if (live && q == 0) P::store(rows_out + (size_t)grp * XW, c);
  if (live && q == 1) P::store(c_out + (size_t)grp * XW, c);
}

// k_reduce_quad with every lane replaced by a DPP quad running the 4-lane addition: 16 lanes per group of 4
// elements, 4 x 4 dependent field products per level.  Used for the small upper levels (latency-bound).
// One group: lanes (q, s) = (element 0..3, product slot 0..3) of a 16-lane row; rows/c of window k hold n_in entries.
template <class P>
__device__ __forceinline__ void reduce_group16(uint32_t* rows_out, uint32_t* c_out, const uint32_t* rows_in,
                                               const uint32_t* c_in, uint32_t n_in, uint32_t A, bool live, uint32_t q,
                                               int s, int base_lane) {
  constexpr int XW = P::ACC_WORDS;
  using Acc = typename P::Acc;
  const uint32_t e = A * 4 + q;
  Acc r, c, v, w, got;
  P::zero(r);
  P::zero(c);
  if (live && e < n_in) {
    P::load(r, rows_in + (size_t)e * XW);
    P::load(c, c_in + (size_t)e * XW);
  }
  {
    const int src[4] = {1, 3, 3, 2};
    quad_fetch<P>(got, q == 2 ? c : r, base_lane + 4 * src[q]);
    P::add_x4(v, q == 3 ? c : r, got, s, false);   // L0 s01, L1 b, L2 a, L3 c23
  }
  {
    const int src[4] = {2, 0, 2, 3};
    quad_fetch<P>(got, q == 0 ? c : v, base_lane + 4 * src[q]);
    if (q == 3) P::zero(got);
    Acc lhs = (q == 1) ? c : v;
    if (q == 3) P::zero(lhs);
    P::add_x4(w, lhs, got, s, q == 2);      // L0 row, L1 c01, L2 2a, L3 0
  }
  {
    const int src[4] = {0, 3, 1, 3};
    quad_fetch<P>(got, q == 0 ? w : v, base_lane + 4 * src[q]);
    P::add_x4(r, w, got, s, q == 0);        // L0 2 row, L1 cs, L2 tri
  }
  {
    const int src[4] = {0, 2, 2, 3};
    quad_fetch<P>(got, r, base_lane + 4 * src[q]);
    P::add_x4(c, r, got, s, q == 0);        // L0 4 row, L1 C' = cs + tri
  }
  if (live && q == 0 && s == 0) P::store(rows_out + (size_t)A * XW, c);
  if (live && q == 1 && s == 0) P::store(c_out + (size_t)A * XW, c);
}

#ifndef MSMZ_Q16_OCC
#define MSMZ_Q16_OCC 1
#endif
template <class P>
__global__ void __launch_bounds__(64, MSMZ_Q16_OCC) k_reduce_quad16(uint32_t* rows_out, uint32_t* c_out, const uint32_t* rows_in,
                                                         const uint32_t* c_in, uint32_t n_in, uint32_t groups,
                                                         uint32_t total) {
  constexpr int XW = P::ACC_WORDS;
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t grp = t >> 4, q = (t >> 2) & 3;
  const int s = (int)(t & 3);
  const bool live = grp < total;
  const uint32_t k = live ? grp / groups : 0, A = live ? grp - k * groups : 0;
  const int lane = threadIdx.x & 63, base_lane = (lane & ~15) + s;
  reduce_group16<P>(rows_out + (size_t)k * groups * XW, c_out + (size_t)k * groups * XW,
                    rows_in + (size_t)k * n_in * XW, c_in + (size_t)k * n_in * XW, n_in, A, live, q, s, base_lane);
}

// The LAST levels in one launch: one workgroup per window walks n_in -> ceil(n_in/4) -> ... -> 1 with a barrier
// between levels (each level is 4 x 4 dependent field products of latency, ~47 us, whatever its size; one launch
// saves the kernel boundaries, not the products).  The window's rows/c ping-pong between
// (r0, c0) and (r1, c1), n_in entries apart per window; the final C (the window sum) is written to c_final[k].
constexpr int REDUCE_TAIL_T = 256;   // one wave per SIMD: the 4-lane additions need ~370 VGPRs (at 256 they spill to scratch)
// entries per window at which the tail kernel takes over: its first level is then 8 groups = one pass of the
// workgroup (reduce stage at 2^20 with 128 / 64 / 32 / 16: 1.355 / 1.324 / 1.308 / 1.333 ms)
constexpr uint32_t REDUCE_TAIL_N = 32;
template <class P>
__global__ void __launch_bounds__(REDUCE_TAIL_T, 1) k_reduce_tail(uint32_t* r0, uint32_t* c0, uint32_t* r1, uint32_t* c1,
                                                                  uint32_t* c_final, uint32_t n_in, uint32_t stride) {
  constexpr int XW = P::ACC_WORDS;
  const uint32_t k = blockIdx.x;
  uint32_t* rin = r0 + (size_t)k * stride * XW;
  uint32_t* cin = c0 + (size_t)k * stride * XW;
  uint32_t* rout = r1 + (size_t)k * stride * XW;
  uint32_t* cout = c1 + (size_t)k * stride * XW;
  const uint32_t q = (threadIdx.x >> 2) & 3;
  const int s = (int)(threadIdx.x & 3);
  const int lane = threadIdx.x & 63, base_lane = (lane & ~15) + s;
  uint32_t n = n_in;
  while (n > 1) {
    const uint32_t g2 = (n + 3) / 4;
    for (uint32_t A0 = 0; A0 < g2; A0 += REDUCE_TAIL_T / 16) {
      const uint32_t A = A0 + (threadIdx.x >> 4);
      reduce_group16<P>(rout, cout, rin, cin, n, A, A < g2, q, s, base_lane);
    }
    __syncthreads();   // same workgroup = same CU: the level's stores are visible to the next level's loads
    uint32_t* t1 = rin; rin = rout; rout = t1;
    uint32_t* t2 = cin; cin = cout; cout = t2;
    n = g2;
  }
  // copy the window sum (C of the last level; for n_in == 1 the input C)
  for (uint32_t j = threadIdx.x; j < (uint32_t)XW; j += REDUCE_TAIL_T) c_final[(size_t)k * XW + j] = cin[j];
}


}  // namespace msmz

#include "reduce_affine.h"
#include "reduce2d_kernels.h"
