// Bucket sort of the (window, digit) entries -- rows a2-a5 of SURVEY.md section 8: GLV split, signed-digit
// slicing, histogram, offsets and the counting-sort scatter (msm-batched-affine.ts:149, 172-200, 411-435, 444-490).
// Included by kernels.h.
//
// The reference copies 116-byte points into bucket order; here only 4-byte references are sorted, and the digits
// are never written to memory: every pass that needs them recomputes them from the 32-byte scalar.
//
//   digit l in [1, L] of window k  ->  bucket index l - 1 = (coarse << FB) | fine,  FB <= FINE_MAX_BITS fine bits
//   coarse bin id = kw * NCB + coarse     (kw = bucket set: window k, or one of the top window's sub-windows, whose bins
//                                          may hold fewer buckets: SortGeom::fbt)
//
//   k_hist        reads scalars, counts entries per coarse bin (LDS histogram); one RETURNING global atomic per bin and
//                 workgroup adds the tile's count to the bin's and tells the tile where its run starts inside the bin
//   k_bin_scan    exclusive scan of the <= 8192 bin counts (one workgroup)
//   k_coarse      THE bucket scatter: reads scalars again (32 B each), slices all K windows, ranks a tile's entries
//                 per bin in LDS and writes (fine | negate | index) words in contiguous runs: 32 B in + 4 K B out per
//                 scalar, nothing else
//   k_fine        one workgroup per coarse bin (<= 2048 buckets, <= 37888 entries): entries pulled into registers
//                 with all loads in flight at once, LDS histogram -> bucket offsets `off`, every reference placed at
//                 its sorted position in LDS, streamed out coalesced; also the largest bucket size
#pragma once
#include <type_traits>
#include <utility>

namespace msmz {

constexpr int COARSE_T = 1024;
constexpr int COARSE_ITEMS = 2;                        // half-scalars (= entries per window) per thread
constexpr int COARSE_TILE = COARSE_T * COARSE_ITEMS;   // entries per window staged by one workgroup
constexpr int COARSE_MAX_BINS = 512;                   // bins per window (top window: incl. its sub-windows) the staging supports
constexpr int SORT_MAX_BINS = 8192;                    // all windows: k_coarse keeps 2 words per bin in LDS (64 KB)
constexpr int kMaxWindowsSort = 128;                   // windows a scalar can have (c >= 2)
constexpr int FINE_MAX_BITS = 11;
constexpr int FINE_T = 1024;
constexpr int FINE_PER = 37;                           // entries a thread holds in registers
constexpr int FINE_STAGE = FINE_T * FINE_PER;          // 37888 entries staged in LDS: 148 KB + 8 KB of counters (+ static) < 160 KB

// Development aid (-DMSMZ_TRACE builds only): thread 0 of every workgroup stamps the 100 MHz wall clock into 16 slots
// behind a buffer the kernel already receives; slot 15 = hardware id (which CU / XCD).  tools/wg_timeline.py reads them.
#ifdef MSMZ_TRACE
#define MSMZ_STAMP(tr, slot)                                                         \
  do {                                                                               \
    if (threadIdx.x == 0) (tr)[(size_t)blockIdx.x * 16 + (slot)] = wall_clock64();   \
  } while (0)
#define MSMZ_STAMP_HW(tr)                                                                                     \
  do {                                                                                                        \
    if (threadIdx.x == 0)                                                                                     \
      (tr)[(size_t)blockIdx.x * 16 + 15] =                                                                    \
          (uint64_t)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((uint64_t)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32); \
  } while (0)
#else
#define MSMZ_STAMP(tr, slot) do {} while (0)
#define MSMZ_STAMP_HW(tr) do {} while (0)
#endif

struct SortGeom {
  uint32_t n;          // scalars
  uint32_t M;          // entries per window: n, or 2 n with GLV (entry n + i = endomorphism half of scalar i)
  int c, K, fb, spread, idx_bits;
  uint32_t ncb;        // coarse bins per bucket set = L >> fb
  // The TOP window's bucket sets may use fewer fine bits (fbt <= fb, ncbt = L >> fbt bins each): its digit range is not a
  // power of two, so its buckets are up to 2x denser than the other windows' and a bin of 2^fb of them would not fit
  // k_fine's LDS staging.  Bins of windows 0..K-2 come first (ncb each), then the top window's sub-windows (ncbt each).
  int fbt;
  uint32_t ncbt;
  // A THIN top window (its digit has only a few significant bits) can be FOLDED into its own bucket set instead of
  // spread over sub-windows: bucket weight j = (entry mod 2^fold_rows) * 2^fold_shift + l, i.e. the L buckets of the set
  // hold 2^fold_rows copies ("rows") of the digit's small range, and the two-dimensional reduction's COLUMN sums are
  // exactly the per-digit sums (the row result of that set is not used).  fold_shift = 0: not folded.
  int fold_shift, fold_rows;
};

// bucket index (weight - 1) of digit l of window k inside its bucket set
__device__ __forceinline__ uint32_t bucket_index(const SortGeom& g, int k, uint32_t l, uint32_t entry) {
  if (g.fold_shift != 0 && k == g.K - 1) return ((entry & ((1u << g.fold_rows) - 1u)) << g.fold_shift) + l - 1u;
  return l - 1u;
}

template <class F, int... Ks>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Ks...>) {
  (f(std::integral_constant<int, Ks>{}), ...);
}
// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>)
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}
// windows a kernel specialized for window size CB unrolls: every position a (half-)scalar of WORDS words can have a digit at
template <int WORDS, int CB>
constexpr int windows_max() { return (WORDS * 32 + CB) / CB; }

// The (half-)scalars of one input scalar as little-endian words.  Windows are sliced with a word index that is the
// same for every lane (the window loop is wave-uniform), so a window costs two register moves out of a uniform
// switch plus one funnel shift instead of shifting the whole scalar down.
template <class Fr, bool GLV>
struct DigitStream {
  static constexpr int HALVES = GLV ? 2 : 1;
  static constexpr int WORDS = GLV ? 4 : 8;
  uint32_t w[HALVES][WORDS];
  uint32_t neg;     // bit h: half h is negative
  uint32_t carry;   // bit h: carry into the next window of half h

  // returns false when the scalar is not below the group order
  __device__ __forceinline__ bool load(const uint32_t* scalars, uint32_t i) {
    uint32_t s[8];
    const uint4* p4 = reinterpret_cast<const uint4*>(scalars + (size_t)i * 8);
    const uint4 a = p4[0], b = p4[1];
    s[0] = a.x; s[1] = a.y; s[2] = a.z; s[3] = a.w;
    s[4] = b.x; s[5] = b.y; s[6] = b.z; s[7] = b.w;
    carry = 0;
    if constexpr (GLV) {
      uint32_t n0, n1;
      glv_decompose<Fr>(w[0], w[1], n0, n1, s);
      neg = n0 | (n1 << 1);
    } else {
#pragma unroll
      for (int j = 0; j < 8; j++) w[0][j] = s[j];
      neg = 0;
    }
    return !words_geq<8>(s, Fr::Q);
  }
  __device__ __forceinline__ void clear() {
#pragma unroll
    for (int h = 0; h < HALVES; h++)
#pragma unroll
      for (int j = 0; j < WORDS; j++) w[h][j] = 0;
    neg = carry = 0;
  }
  // c bits at (wave-uniform) bit position pos of half h
  __device__ __forceinline__ uint32_t bits(int h, int pos, int c) const {
    const int wi = __builtin_amdgcn_readfirstlane(pos >> 5);
    uint32_t lo = 0, hi = 0;
    switch (wi) {   // uniform: one scalar branch
      case 0: lo = w[h][0]; hi = w[h][1]; break;
      case 1: lo = w[h][1]; hi = w[h][2]; break;
      case 2: lo = w[h][2]; hi = w[h][3]; break;
      case 3: lo = w[h][3]; hi = WORDS > 4 ? w[h][WORDS > 4 ? 4 : 0] : 0u; break;
      case 4: if (WORDS > 4) { lo = w[h][WORDS > 4 ? 4 : 0]; hi = w[h][WORDS > 5 ? 5 : 0]; } break;
      case 5: if (WORDS > 4) { lo = w[h][WORDS > 5 ? 5 : 0]; hi = w[h][WORDS > 6 ? 6 : 0]; } break;
      case 6: if (WORDS > 4) { lo = w[h][WORDS > 6 ? 6 : 0]; hi = w[h][WORDS > 7 ? 7 : 0]; } break;
      case 7: if (WORDS > 4) { lo = w[h][WORDS > 7 ? 7 : 0]; } break;
      default: break;
    }
    return __builtin_amdgcn_alignbit(hi, lo, (uint32_t)(pos & 31)) & ((1u << c) - 1u);
  }
  // signed digit of window k of half h (msm-batched-affine.ts:180-199): returns l in [0, L], sets `ng` to its sign.
  // Windows must be visited in order 0, 1, 2, ... (the carry).
  __device__ __forceinline__ uint32_t next(int h, int k, int c, uint32_t L, uint32_t& ng) {
    uint32_t l = bits(h, k * c, c) + ((carry >> h) & 1u);
    uint32_t cy = 0;
    if (l > L) {
      l = 2 * L - l;
      cy = 1;
    }
    carry = (carry & ~(1u << h)) | (cy << h);
    ng = (cy ^ (neg >> h)) & (l != 0 ? 1u : 0u);   // the half scalar's own sign flips every digit's sign
    return l;
  }
  // The same with the window index and the window size known at compile time (the kernels specialized for the default
  // window sizes unroll their window loop): the word index and the shift are immediates, a window costs one bit-field
  // extract (or one funnel shift + mask when it straddles two words) instead of the uniform switch above.
  template <int POS, int CB>
  __device__ __forceinline__ uint32_t bits_c(int h) const {
    constexpr int wi = POS >> 5, sh = POS & 31;
    constexpr uint32_t mask = (1u << CB) - 1u;
    if constexpr (wi >= WORDS) {
      return 0u;
    } else if constexpr (sh + CB <= 32 || wi + 1 >= WORDS) {
      return (w[h][wi] >> sh) & mask;
    } else {
      return __builtin_amdgcn_alignbit(w[h][wi + 1 < WORDS ? wi + 1 : wi], w[h][wi], (uint32_t)sh) & mask;
    }
  }
  template <int KW, int CB>
  __device__ __forceinline__ uint32_t next_c(int h, uint32_t& ng) {
    constexpr uint32_t L = 1u << (CB - 1);
    uint32_t l = bits_c<KW * CB, CB>(h) + ((carry >> h) & 1u);
    uint32_t cy = 0;
    if (l > L) {
      l = 2 * L - l;
      cy = 1;
    }
    carry = (carry & ~(1u << h)) | (cy << h);
    ng = (cy ^ (neg >> h)) & (l != 0 ? 1u : 0u);
    return l;
  }
  // does the scalar need more than K windows?  (a carry out of the last one, or bits beyond it)
  __device__ __forceinline__ bool overflows(int K, int c) const {
    uint32_t o = carry & ((1u << HALVES) - 1u);
#pragma unroll
    for (int h = 0; h < HALVES; h++)
      for (int pos = K * c; pos < 32 * WORDS; pos += 16) o |= bits(h, pos, 16);
    return o != 0;
  }
};

// (Measured: matching equal keys across the wave with ballots so that one lane per key issues the LDS atomic costs
// ~80 VALU instructions per wave instruction and made k_hist 2.2x and k_coarse 1.5x SLOWER than plain per-lane LDS
// atomics, which cost ~24 LDS cycles per wave instruction here.)

// bucket set (window, or sub-window of the sparse top window) and coarse bin of a non-zero digit
__device__ __forceinline__ uint32_t coarse_bin(const SortGeom& g, int k, uint32_t l, uint32_t entry) {
  const uint32_t bi = bucket_index(g, k, l, entry);
  if (k == g.K - 1) return (uint32_t)k * g.ncb + (entry & ((1u << g.spread) - 1u)) * g.ncbt + (bi >> g.fbt);
  return (uint32_t)k * g.ncb + (bi >> g.fb);
}

// ------------------------------------------------------------------------------------------------ histogram
// counts[bin] += entries; meta->error |= 2 when a scalar does not fit K windows (a GLV half above the assumed
// bound: the host then repeats the MSM with one more bit), |= 4 when a scalar is not below the group order
// (scalarsFromBytes' precondition, checked here instead of in a serial host loop).
// C > 0: specialized for window size C (window loop unrolled, DigitStream::next_c); C = 0: any window size.
template <class Fr, bool GLV, int C>
__global__ void __launch_bounds__(COARSE_T, 8) k_hist(uint32_t* counts, uint16_t* tile_counts, uint32_t* tile_offs, MsmMeta* meta,
                                                 const uint32_t* scalars, SortGeom g, uint32_t nbins) {
  extern __shared__ uint32_t s_hist[];
  constexpr int HALVES = GLV ? 2 : 1;
  constexpr int PER = COARSE_ITEMS / HALVES;   // scalars per thread: one workgroup = one tile of k_coarse
  const uint32_t L = 1u << (g.c - 1);
  for (uint32_t b = threadIdx.x; b < nbins; b += COARSE_T) s_hist[b] = 0;
  __syncthreads();
  uint32_t bad = 0;
  // all of the thread's scalars are requested before any is sliced (one memory latency, not PER)
  DigitStream<Fr, GLV> ds[PER];
  uint32_t idx[PER];
#pragma unroll
  for (int it = 0; it < PER; it++) {
    idx[it] = (blockIdx.x * PER + it) * COARSE_T + threadIdx.x;
    if (idx[it] < g.n) {
      if (!ds[it].load(scalars, idx[it])) bad |= 4u;
    } else {
      ds[it].clear();
    }
  }
  auto window = [&](auto kk) {
    const int k = kk;
#pragma unroll
    for (int it = 0; it < PER; it++) {
#pragma unroll
      for (int h = 0; h < HALVES; h++) {
        uint32_t ng, l;
        if constexpr (C > 0) l = ds[it].template next_c<decltype(kk)::value, C>(h, ng); else l = ds[it].next(h, k, g.c, L, ng);
        if (l != 0) atomicAdd(&s_hist[coarse_bin(g, k, l, (uint32_t)h * g.n + idx[it])], 1u);
      }
    }
  };
  if constexpr (C > 0) {
    static_for<windows_max<DigitStream<Fr, GLV>::WORDS, C>()>([&](auto kc) {
      if (decltype(kc)::value < g.K) window(kc);
    });
  } else {
#pragma unroll 1
    for (int k = 0; k < g.K; k++) window(k);
  }
#pragma unroll
  for (int it = 0; it < PER; it++)
    if (ds[it].overflows(g.K, g.c)) bad |= 2u;
  if (bad) atomicOr(&meta->error, bad);
  __syncthreads();
  // The tile's run inside every bin is reserved HERE: the atomic that builds the global counts returns where the
  // tile's entries start in the bin (the scatter kernel then needs no atomics: 512 tiles x 512 bins returning atomics on
  // 512 addresses cost it ~7 us of its 41).  All of a thread's atomics are in flight together.
  uint16_t* row = tile_counts + (size_t)blockIdx.x * nbins;
  uint32_t* roff = tile_offs + (size_t)blockIdx.x * nbins;
  constexpr int PB = SORT_MAX_BINS / COARSE_T;
  uint32_t r[PB];
#pragma unroll
  for (int q = 0; q < PB; q++) {
    const uint32_t b = (uint32_t)q * COARSE_T + threadIdx.x;
    r[q] = 0;
    if (b < nbins) {
      const uint32_t v = s_hist[b];
      row[b] = (uint16_t)v;             // <= COARSE_TILE entries of a tile fall into one bin
      if (v) r[q] = atomicAdd(&counts[b], v);
    }
  }
#pragma unroll
  for (int q = 0; q < PB; q++) {
    const uint32_t b = (uint32_t)q * COARSE_T + threadIdx.x;
    if (b < nbins) roff[b] = r[q];
  }
}

// exclusive scan of nbins <= SORT_MAX_BINS counts by one workgroup; base[nbins] = total = number of entries
static __global__ void __launch_bounds__(1024) k_bin_scan(uint32_t* base, const uint32_t* counts, uint32_t nbins,
                                                          uint32_t* total_out) {
  __shared__ uint32_t s_wave[16];
  constexpr int PER = SORT_MAX_BINS / 1024;
  uint32_t v[PER], sum = 0;
#pragma unroll
  for (int j = 0; j < PER; j++) {
    const uint32_t b = threadIdx.x * PER + j;
    v[j] = b < nbins ? counts[b] : 0;
    sum += v[j];
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t x = sum;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t y = __shfl_up(x, d, 64);
    if (lane >= d) x += y;
  }
  if (lane == 63) s_wave[wave] = x;
  __syncthreads();
  uint32_t wbase = 0, tot = 0;
#pragma unroll
  for (int w2 = 0; w2 < 16; w2++) {
    const uint32_t t = s_wave[w2];
    if (w2 < wave) wbase += t;
    tot += t;
  }
  uint32_t ex = wbase + x - sum;
#pragma unroll
  for (int j = 0; j < PER; j++) {
    const uint32_t b = threadIdx.x * PER + j;
    if (b < nbins) base[b] = ex;
    ex += v[j];
  }
  if (threadIdx.x == 0) {
    base[nbins] = tot;
    *total_out = tot;
  }
}

// ------------------------------------------------------------------------------------------------ coarse scatter
// One tile = COARSE_TILE half-scalars (2048 scalars, or 1024 scalars with GLV) = the tile k_hist counted.
//   setup    the tile's per-bin entry counts (2 bytes per bin) and the start of its run inside every bin (k_hist's
//            returning atomics) are read back, a block scan turns the counts into staging offsets: no global atomics here;
//   windows  one by one: slice, rank the entries per bin (one LDS atomic on a cursor that starts at the bin's staging
//            offset), stage them in LDS in bin order, write every bin's entries as one contiguous run.  Double-buffered
//            staging: 1 barrier per window.
// Algorithmic HBM bytes: 32 B read per scalar + 4 B written per entry.  Dynamic LDS: 2 * nbins words.
template <class Fr, bool GLV, int C>
__global__ void __launch_bounds__(COARSE_T, 8) k_coarse(uint32_t* packed_out, const uint32_t* tile_offs, const uint32_t* bin_base,
                                                        const uint16_t* tile_counts, const uint32_t* scalars, SortGeom g,
                                                        uint32_t nbins) {
  constexpr int HALVES = GLV ? 2 : 1;
  constexpr int SC = COARSE_ITEMS / HALVES;           // scalars per thread
  extern __shared__ uint32_t s_dyn[];
  uint32_t* s_cur = s_dyn;                  // [nbins] cursor: next staging position of the bin (starts at its staging offset)
  uint32_t* s_delta = s_dyn + nbins;        // [nbins] (global index of the tile's run in the bin) - (staging offset)
  static_assert(COARSE_TILE == 2048 && COARSE_MAX_BINS <= 512 && FINE_MAX_BITS <= 11, "staged word: 11 + 1 + 11 + 9 bits");
  __shared__ uint32_t s_stage[2][COARSE_TILE];   // staged words, in bin order
  __shared__ uint32_t s_wave[COARSE_T / 64];
  __shared__ uint32_t s_wstart[kMaxWindowsSort + 1];   // staging offset of every window's first bin; [K] = tile total
  const uint32_t L = 1u << (g.c - 1);
#ifdef MSMZ_TRACE
  uint64_t* trace = reinterpret_cast<uint64_t*>(const_cast<uint32_t*>(tile_offs) + (size_t)gridDim.x * nbins);
#endif
  MSMZ_STAMP(trace, 0);
  MSMZ_STAMP_HW(trace);

  DigitStream<Fr, GLV> ds[SC];
  uint32_t idx[SC];
#pragma unroll
  for (int s = 0; s < SC; s++) {
    idx[s] = (blockIdx.x * SC + s) * COARSE_T + threadIdx.x;
    if (idx[s] < g.n) ds[s].load(scalars, idx[s]); else ds[s].clear();
  }
  {
    const uint16_t* row = tile_counts + (size_t)blockIdx.x * nbins;
    const uint32_t* roff = tile_offs + (size_t)blockIdx.x * nbins;
    const uint32_t per = (nbins + COARSE_T - 1) / COARSE_T;   // consecutive bins per thread
    const uint32_t b0 = threadIdx.x * per;
    const int lncb = g.c - 1 - g.fb;
    uint32_t sum = 0, ex, total;
    if (per == 1) {
      // (<= 1024 bins: one bin per thread, three independent loads)
      const uint32_t cnt = b0 < nbins ? row[b0] : 0u;
      const uint32_t gb = b0 < nbins ? bin_base[b0] + roff[b0] : 0u;
      ex = block_exclusive_scan<COARSE_T>(cnt, &total, s_wave);
      if (b0 < nbins) {
        s_cur[b0] = ex;
        s_delta[b0] = gb - ex;
        if ((b0 & (g.ncb - 1u)) == 0 && (b0 >> lncb) < (uint32_t)g.K) s_wstart[b0 >> lncb] = ex;   // ncb is a power of two
      }
    } else {
      // more bins than threads (>= 2^21 entries per window): the three rows are read COALESCED with every load of a
      // thread in flight (bin = q * 1024 + thread), parked in LDS, and only then walked in the scan's order (consecutive
      // bins per thread) -- a thread reading its 4-8 consecutive bins straight from memory paid their latency in turn
      // (14.7 us of a 43 us workgroup life at 2^23)
      constexpr int PB = SORT_MAX_BINS / COARSE_T;
      uint32_t cnt[PB], gb[PB];
#pragma unroll
      for (int q = 0; q < PB; q++) {
        const uint32_t b = (uint32_t)q * COARSE_T + threadIdx.x;
        cnt[q] = b < nbins ? row[b] : 0u;
        gb[q] = b < nbins ? bin_base[b] + roff[b] : 0u;
      }
#pragma unroll
      for (int q = 0; q < PB; q++) {
        const uint32_t b = (uint32_t)q * COARSE_T + threadIdx.x;
        if (b < nbins) {
          s_cur[b] = cnt[q];
          s_delta[b] = gb[q];
        }
      }
      __syncthreads();
      for (uint32_t q = 0; q < per; q++) {
        const uint32_t b = b0 + q;
        if (b < nbins) sum += s_cur[b];
      }
      ex = block_exclusive_scan<COARSE_T>(sum, &total, s_wave);
      for (uint32_t q = 0; q < per; q++) {
        const uint32_t b = b0 + q;
        if (b < nbins) {
          const uint32_t c = s_cur[b];
          s_cur[b] = ex;
          s_delta[b] -= ex;
          if ((b & (g.ncb - 1u)) == 0 && (b >> lncb) < (uint32_t)g.K) s_wstart[b >> lncb] = ex;
          ex += c;
        }
      }
    }
    if (threadIdx.x == 0) s_wstart[g.K] = total;
  }
  __syncthreads();
  MSMZ_STAMP(trace, 1);
  // Stage window k into buffer `buf` in bin order.  A staged word is (fine | negate) << 20 | local index << 9 | bin: the
  // bin rides along (<= 512 bins per window, 2048 half-scalars per tile: 11 + 1 + 11 + 9 bits), so the copy-out finds the
  // run's address from the word itself and no second LDS array of destinations is written and read per entry.
  auto stage = [&](auto kk, const int buf, auto is_top) {
    constexpr bool TOP = decltype(is_top)::value;
    const int k = kk;
    const uint32_t wbase = s_wstart[k];
    uint32_t* cur_k = s_cur + (uint32_t)k * g.ncb;
#pragma unroll
    for (int s = 0; s < SC; s++) {
#pragma unroll
      for (int h = 0; h < HALVES; h++) {
        uint32_t ng, l;
        if constexpr (C > 0) l = ds[s].template next_c<decltype(kk)::value, C>(h, ng); else l = ds[s].next(h, k, g.c, L, ng);
        if (l != 0) {
          uint32_t bi = l - 1u, bin;
          if constexpr (TOP) {
            const uint32_t entry = (uint32_t)h * g.n + idx[s];
            bi = bucket_index(g, k, l, entry);
            bin = (entry & ((1u << g.spread) - 1u)) * g.ncbt + (bi >> g.fbt);
          } else {
            bin = bi >> g.fb;
          }
          const uint32_t fmask = (1u << (TOP ? g.fbt : g.fb)) - 1u;
          const uint32_t local = (uint32_t)(h * SC + s) * COARSE_T + threadIdx.x;   // < COARSE_TILE
          const uint32_t pa = atomicAdd(&cur_k[bin], 1u);   // staging position, tile-relative
          s_stage[buf][pa - wbase] = ((((bi & fmask) << 1) | ng) << 20) | (local << 9) | bin;
        }
      }
    }
  };
  // ... and, after a barrier, all threads copy it out: consecutive staged words of a bin go to consecutive addresses, so
  // every run is a contiguous, coalesced store
  auto copy_out = [&](const int k, const int buf) {
    const uint32_t wbase = s_wstart[k];
    const uint32_t wcnt = s_wstart[k == g.K - 1 ? g.K : k + 1] - wbase;
    const uint32_t* delta_k = s_delta + (uint32_t)k * g.ncb;
#pragma unroll
    for (int q = 0; q < COARSE_ITEMS; q++) {
      const uint32_t p = (uint32_t)q * COARSE_T + threadIdx.x;
      if (p < wcnt) {
        const uint32_t w = s_stage[buf][p];
        const uint32_t local = (w >> 9) & (COARSE_TILE - 1);
        uint32_t entry;
        if constexpr (GLV) entry = (local / COARSE_T) * g.n + blockIdx.x * COARSE_T + (local % COARSE_T);   // SC = 1
        else entry = blockIdx.x * COARSE_TILE + local;
        packed_out[delta_k[w & 511u] + wbase + p] = ((w >> 20) << g.idx_bits) | entry;
      }
    }
  };
  // one window per barrier, two buffers: the next window stages into the other buffer; the barrier of the window after
  // that orders this buffer's reuse behind these reads.  (Two windows per barrier with four buffers measured slower.)
  if constexpr (C > 0) {
    static_for<windows_max<DigitStream<Fr, GLV>::WORDS, C>()>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      if (k == 1) MSMZ_STAMP(trace, 2);
      if (k < g.K) {
        if (k == g.K - 1) {
          MSMZ_STAMP(trace, 3);
          stage(kc, k & 1, std::true_type{});
        } else {
          stage(kc, k & 1, std::false_type{});
        }
        __syncthreads();
        copy_out(k, k & 1);
      }
    });
  } else {
#pragma unroll 1
    for (int k = 0; k < g.K - 1; k++) {
      if (k == 1) MSMZ_STAMP(trace, 2);
      stage(k, k & 1, std::false_type{});
      __syncthreads();
      copy_out(k, k & 1);
    }
    MSMZ_STAMP(trace, 3);
    stage(g.K - 1, (g.K - 1) & 1, std::true_type{});
    __syncthreads();
    copy_out(g.K - 1, (g.K - 1) & 1);
  }
  MSMZ_STAMP(trace, 4);
}

// ------------------------------------------------------------------------------------------------ fine sort
// `n_half` / `endo_delta`: with GLV the entry index i >= n_half is the endomorphism half of point i - n_half; its
// record sits at index i + endo_delta of the point set (the images follow the whole set, which may be larger than
// the prefix this MSM covers: msm-batched-affine.ts:74-97 takes any N <= allocated).
// Bins below `top_bin` hold 2^fb buckets each, the top window's bins (from `top_bin` on) 2^fbt.
static __global__ void __launch_bounds__(FINE_T) k_fine(uint32_t* refs, uint32_t* off, uint32_t* max_bucket,
                                                 const uint32_t* packed, const uint32_t* bin_base, int fb, int fbt,
                                                 uint32_t top_bin, uint32_t n_bins, int idx_bits, uint32_t n_half,
                                                 uint32_t endo_delta) {
  extern __shared__ uint32_t s_dyn[];
  uint32_t* s_cnt = s_dyn;                                  // [1 << FINE_MAX_BITS] counts, then running cursors
  uint32_t* s_stage = s_dyn + (1 << FINE_MAX_BITS);          // [FINE_STAGE]
  __shared__ uint32_t s_wave[FINE_T / 64];
  __shared__ uint32_t s_wmax[FINE_T / 64];
  // last bins first: the top window's bins are the only ones that are structurally above average (its digit range is
  // not a power of two, so its buckets are up to 2x denser), and the workgroups that start first should be the long ones
  const uint32_t bin = n_bins - 1u - blockIdx.x;
  const bool top = bin >= top_bin;
  const uint32_t nfine = 1u << (top ? fbt : fb);
  // first bucket of the bin in `off`
  const size_t bucket0 = top ? ((size_t)top_bin << fb) + ((size_t)(bin - top_bin) << fbt) : (size_t)bin << fb;
  const uint32_t per = (nfine + FINE_T - 1) / FINE_T;        // consecutive buckets per thread (<= 2)
  const uint32_t begin = bin_base[bin], end = bin_base[bin + 1];
  const uint32_t cnt_bin = end - begin;
  if (cnt_bin == 0) {   // (the top window's bins beyond its digit range): every bucket is empty and starts at `begin`
    for (uint32_t f = threadIdx.x; f < nfine; f += FINE_T) off[bucket0 + f] = begin;
    if (bin + 1 == n_bins && threadIdx.x == 0) off[bucket0 + nfine] = end;
    return;
  }
  const bool staged = cnt_bin <= (uint32_t)FINE_STAGE;   // the bin fits the threads' registers (and the LDS staging)
  const uint32_t imask = (1u << idx_bits) - 1u;
#ifdef MSMZ_TRACE
  uint64_t* trace = reinterpret_cast<uint64_t*>(const_cast<uint32_t*>(bin_base) + ((n_bins + 2) & ~1u));
#endif
  MSMZ_STAMP(trace, 0);
  MSMZ_STAMP_HW(trace);
  for (uint32_t f = threadIdx.x; f < nfine; f += FINE_T) s_cnt[f] = 0;
  // the bin's entries: all loads of a thread are issued back to back (the bin is read ONCE)
  uint32_t v[FINE_PER];
  if (staged) {
#pragma unroll
    for (int j = 0; j < FINE_PER; j++) {
      const uint32_t p = (uint32_t)j * FINE_T + threadIdx.x;
      v[j] = p < cnt_bin ? packed[begin + p] : 0u;
    }
  }
  __syncthreads();
  MSMZ_STAMP(trace, 1);
  // histogram; on the staged path the atomic's return value IS the entry's rank inside its bucket (kept in a register),
  // so no second round of atomics is needed: position = bucket offset + rank
  uint32_t rank[FINE_PER];
  if (staged) {
#pragma unroll
    for (int j = 0; j < FINE_PER; j++) {
      rank[j] = 0;
      if ((uint32_t)j * FINE_T + threadIdx.x < cnt_bin) rank[j] = atomicAdd(&s_cnt[v[j] >> (idx_bits + 1)], 1u);
    }
  } else {
    // a bin beyond the staging: same register array, FINE_STAGE entries at a time (all of a chunk's loads in flight)
    for (uint32_t c0 = 0; c0 < cnt_bin; c0 += FINE_STAGE) {
#pragma unroll
      for (int j = 0; j < FINE_PER; j++) {
        const uint32_t p = c0 + (uint32_t)j * FINE_T + threadIdx.x;
        v[j] = p < cnt_bin ? packed[begin + p] : 0u;
      }
#pragma unroll
      for (int j = 0; j < FINE_PER; j++)
        if (c0 + (uint32_t)j * FINE_T + threadIdx.x < cnt_bin) atomicAdd(&s_cnt[v[j] >> (idx_bits + 1)], 1u);
    }
  }
  __syncthreads();
  MSMZ_STAMP(trace, 2);
  uint32_t mine = 0, mx = 0, cnts[2] = {0, 0};
  for (uint32_t j = 0; j < per; j++) {
    const uint32_t f = threadIdx.x * per + j;
    const uint32_t c = f < nfine ? s_cnt[f] : 0;
    cnts[j & 1] = c;
    mine += c;
    mx = c > mx ? c : mx;
  }
  // block-wide exclusive scan of `mine` over FINE_T threads
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t x = mine;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t y = __shfl_up(x, d, 64);
    if (lane >= d) x += y;
  }
  // (the largest bucket of the bin: reduced over the workgroup first -- one global atomic per workgroup, not per wave:
  // thousands of atomics on ONE address drain at ~90 per microsecond)
  uint32_t wmx = mx;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const uint32_t o = __shfl_xor(wmx, d, 64);
    wmx = o > wmx ? o : wmx;
  }
  if (lane == 63) {
    s_wave[wave] = x;
    s_wmax[wave] = wmx;
  }
  __syncthreads();
  uint32_t wbase = 0;
  for (int w2 = 0; w2 < wave; w2++) wbase += s_wave[w2];
  uint32_t ex = wbase + x - mine;                            // relative to the bin start
  for (uint32_t j = 0; j < per; j++) {
    const uint32_t f = threadIdx.x * per + j;
    if (f < nfine) {
      s_cnt[f] = ex;                                         // the bucket's offset (running cursor on the unstaged path)
      off[bucket0 + f] = begin + ex;
      ex += cnts[j & 1];
    }
  }
  if (threadIdx.x == 0) {
    uint32_t bmx = 0;
    for (int w2 = 0; w2 < FINE_T / 64; w2++) bmx = s_wmax[w2] > bmx ? s_wmax[w2] : bmx;
    if (bmx > 1) atomicMax(max_bucket, bmx);
  }
  if (bin + 1 == n_bins && threadIdx.x == 0) off[bucket0 + nfine] = end;   // = off[number of buckets]
  __syncthreads();
  MSMZ_STAMP(trace, 3);
  auto to_ref = [&](uint32_t pv) {
    uint32_t idx = pv & imask;
    if (idx >= n_half) idx += endo_delta;   // endomorphism half: record index in the point set
    return idx | (((pv >> idx_bits) & 1u) << 31);
  };
  if (staged) {
#pragma unroll
    for (int j = 0; j < FINE_PER; j++) {
      if ((uint32_t)j * FINE_T + threadIdx.x < cnt_bin) {
        const uint32_t pos = s_cnt[v[j] >> (idx_bits + 1)] + rank[j];
        s_stage[pos] = to_ref(v[j]);
      }
    }
    __syncthreads();
    MSMZ_STAMP(trace, 4);
    for (uint32_t p = threadIdx.x; p < cnt_bin; p += FINE_T) refs[begin + p] = s_stage[p];
    MSMZ_STAMP(trace, 5);
  } else {
    // a bin too large for the LDS staging (a dense top window, heavily repeated scalars): second read in the same
    // chunks, running cursors in LDS, scattered stores (the bin's range is L2-resident while it is written)
    for (uint32_t c0 = 0; c0 < cnt_bin; c0 += FINE_STAGE) {
#pragma unroll
      for (int j = 0; j < FINE_PER; j++) {
        const uint32_t p = c0 + (uint32_t)j * FINE_T + threadIdx.x;
        v[j] = p < cnt_bin ? packed[begin + p] : 0u;
      }
#pragma unroll
      for (int j = 0; j < FINE_PER; j++) {
        if (c0 + (uint32_t)j * FINE_T + threadIdx.x < cnt_bin) {
          const uint32_t pos = atomicAdd(&s_cnt[v[j] >> (idx_bits + 1)], 1u);
          refs[begin + pos] = to_ref(v[j]);
        }
      }
    }
    MSMZ_STAMP(trace, 4);
    MSMZ_STAMP(trace, 5);
  }
}

}  // namespace msmz
