// Schedule of the batched-affine tree rounds as explicit descriptor lists.  Included by kernels.h.
//
// Round r (m = 2^r) adds, inside every bucket, element j*2m + m into element j*2m (positions relative to the bucket
// start) for every j with j*2m + m < size -- exactly the reference's schedule (msm-batched-affine.ts:232-247).  The
// reference walks its sorted point array in place; here every round writes a COMPACT result array R_r inside `slots`
// (pairs numbered bucket by bucket), so the value of relative position `pos` of a bucket before round r lives at
//     R_rr[ P_rr(g) + pos / 2^(rr+1) ],   rr = min(r-1, floor(log2(size-pos-1)))     (size-pos >= 2)
//     the original point refs[start+pos]                                               (size-pos == 1 or r == 0)
// with P_r(g) = number of round-r pairs in the buckets before g.  Instead of letting every pair of every round
// rediscover its bucket (binary search over per-round prefix sums) and its operands, one pass over the buckets emits
// for every pair of every round the two operand locations:
//     desc[2 * (base_r + t)] = {locA, locB},   t = P_r(g) + j,   result record = base_r + t
// and for every bucket the <= 4 locations of what the rounds leave of it (`bfin`, read by the bucket reduction).
//
//   k_plan_count   per chunk of PLAN_CHUNK buckets: pairs per round            (tiny)
//   k_plan_emit    per chunk: round bases from the chunk totals, per-thread running pair numbers, descriptors
//
// The number of rounds is decided on the device from the largest bucket: R = ceil(log2 max) - tail_skip (the last
// rounds only touch the few longest buckets but cost a full round of latency, so the reduction's loader adds the <= 4
// partial sums such a bucket is left with).
#pragma once

namespace msmz {

constexpr int PLAN_T = 512;
constexpr int PLAN_PER = 2;                       // consecutive buckets per thread (counting / scan phases)
constexpr int PLAN_CHUNK = PLAN_T * PLAN_PER;     // buckets per workgroup
constexpr int PLAN_RMAX = 26;                     // rounds supported (bucket sizes < 2^26)
constexpr int PLAN_RL = 6;                        // rounds whose per-bucket pair numbers sit in LDS (one thread per PAIR);
                                                  // later rounds (buckets > 64 entries) are emitted bucket by bucket
constexpr int PLAN_TILE = 4096;                   // pairs whose owner buckets are expanded into LDS at a time
// (LDS: 6 x 4.1 KB of prefixes + 4.1 KB of offsets + 8 KB of owners = 37 KB: FOUR workgroups per CU, so that the 1024
// chunks of a 2^20-point MSM are resident together; at 53 KB three were, and the fourth quarter ran as a second pass:
// 95 us instead of ~50, profiles/r03_wg_timelines.txt)
constexpr uint32_t LOC_ORIG = 0x40000000u;        // location word: bit 30 = original point (bit 31 = negate), else slot record
constexpr uint32_t LOC_NONE = 0xffffffffu;

__device__ __forceinline__ int plan_rounds(uint32_t max_bucket, int tail_skip) {
  if (max_bucket <= 1) return 0;
  const int rfull = 32 - __builtin_clz(max_bucket - 1);   // rounds m = 1, 2, 4, ... < max_bucket
  if (rfull <= 1) return rfull;
  const int r = rfull - tail_skip;
  return r < 1 ? 1 : (r > PLAN_RMAX ? PLAN_RMAX : r);
}

__device__ __forceinline__ uint32_t pairs_in_round(uint32_t size, int r) {   // number of j with j*2m + m < size
  return (size + (1u << r) - 1u) >> (r + 1);
}

// chunk_pairs[r * n_chunks + chunk] = pairs of round r in the chunk's buckets, r < PLAN_RMAX
// `chunk` <= PLAN_CHUNK buckets per workgroup (the host picks it so that there are enough workgroups for the GPU even
// when a window has few, long buckets).
// The buckets from `nb_main` on (the top window's bucket sets, up to 2x denser than the others) are cut into chunks of
// `chunk_top` <= chunk buckets, so that their workgroups do not outlast the rest (a launch ends with its slowest one).
struct PlanChunks {
  uint32_t chunk, nb_main, n_main, chunk_top;
};
__device__ __forceinline__ void plan_chunk_range(const PlanChunks& pc, uint32_t nb, uint32_t& g0, uint32_t& nbk) {
  if (blockIdx.x < pc.n_main) {
    g0 = blockIdx.x * pc.chunk;
    nbk = pc.nb_main - g0 < pc.chunk ? pc.nb_main - g0 : pc.chunk;
  } else {
    g0 = pc.nb_main + (blockIdx.x - pc.n_main) * pc.chunk_top;
    nbk = nb - g0 < pc.chunk_top ? nb - g0 : pc.chunk_top;
  }
}

static __global__ void __launch_bounds__(PLAN_T) k_plan_count(uint32_t* chunk_pairs, const uint32_t* off, uint32_t nb,
                                                              uint32_t n_chunks, const MsmMeta* meta, int tail_skip,
                                                              PlanChunks pc) {
  __shared__ uint32_t s_tot[PLAN_RMAX];
  const int R = plan_rounds(meta->max_bucket, tail_skip);
  if (threadIdx.x < PLAN_RMAX) s_tot[threadIdx.x] = 0;
  __syncthreads();
  uint32_t c0, nbk;
  plan_chunk_range(pc, nb, c0, nbk);
  const uint32_t g0 = c0 + threadIdx.x * PLAN_PER;
  uint32_t size[PLAN_PER];
#pragma unroll
  for (int q = 0; q < PLAN_PER; q++)
    size[q] = threadIdx.x * PLAN_PER + q < nbk ? off[g0 + q + 1] - off[g0 + q] : 0u;
  for (int r = 0; r < R; r++) {
    uint32_t s = 0;
#pragma unroll
    for (int q = 0; q < PLAN_PER; q++) s += pairs_in_round(size[q], r);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
    if ((threadIdx.x & 63) == 0 && s) atomicAdd(&s_tot[r], s);
  }
  __syncthreads();
  if ((int)threadIdx.x < PLAN_RMAX) chunk_pairs[(size_t)threadIdx.x * n_chunks + blockIdx.x] = (int)threadIdx.x < R ? s_tot[threadIdx.x] : 0u;
}

static __global__ void __launch_bounds__(PLAN_T, 8) k_plan_emit(uint2* desc, uint4* bfin, MsmMeta* meta,
                                                             const uint32_t* chunk_pairs, const uint32_t* off,
                                                             const uint32_t* refs, uint32_t nb, uint32_t n_chunks,
                                                             int tail_skip, uint32_t* pair_scratch, PlanChunks pc) {
  __shared__ uint32_t s_before[PLAN_RMAX];              // pairs of round r in the chunks before this one
  __shared__ uint32_t s_total[PLAN_RMAX];               // pairs of round r
  __shared__ uint32_t s_rbase[PLAN_RMAX + 1];           // first record of round r's result array
  __shared__ uint32_t s_start[PLAN_CHUNK + 1];          // bucket offsets of the chunk
  __shared__ uint32_t s_pref[PLAN_RL][PLAN_CHUNK + 1];  // r < PLAN_RL: pairs of round r in the chunk's buckets before b
  // r >= PLAN_RL: running pair number of the thread's current bucket -- rarely needed (buckets > 256 entries), so it
  // lives in a per-workgroup slice of global scratch instead of 36 KB of LDS (which would cost occupancy)
  uint32_t* s_pair = pair_scratch + (size_t)blockIdx.x * (PLAN_RMAX - PLAN_RL) * PLAN_T;
  __shared__ uint32_t s_wave[PLAN_T / 64];
  __shared__ uint16_t s_owner[PLAN_TILE];              // chunk bucket of every pair of the current tile
#ifdef MSMZ_TRACE
  uint64_t* trace = reinterpret_cast<uint64_t*>(pair_scratch + (size_t)gridDim.x * (PLAN_RMAX - PLAN_RL) * PLAN_T);
#endif
  MSMZ_STAMP(trace, 0);
  MSMZ_STAMP_HW(trace);
  const int R = plan_rounds(meta->max_bucket, tail_skip);
  const int RL = R < PLAN_RL ? R : PLAN_RL;
  if (threadIdx.x < PLAN_RMAX) {
    s_before[threadIdx.x] = 0;
    s_total[threadIdx.x] = 0;
  }
  uint32_t g0, nbk;   // first bucket and number of buckets of this chunk
  plan_chunk_range(pc, nb, g0, nbk);
  for (uint32_t b = threadIdx.x; b <= nbk; b += PLAN_T) s_start[b] = off[g0 + b];
  __syncthreads();
  MSMZ_STAMP(trace, 1);
  if (blockIdx.x != 0 && s_start[nbk] == s_start[0]) {
    // a chunk of empty buckets (the top window's sets beyond its digit range): nothing to schedule, nothing left over;
    // leaves at once so that a waiting workgroup gets the slot (block 0 stays: it publishes the round totals)
    for (uint32_t b = threadIdx.x; b < nbk; b += PLAN_T) bfin[g0 + b] = make_uint4(LOC_NONE, LOC_NONE, LOC_NONE, LOC_NONE);
    return;
  }
  for (int r = 0; r < R; r++) {
    uint32_t tot = 0, pre = 0;
    for (uint32_t b = threadIdx.x; b < n_chunks; b += PLAN_T) {
      const uint32_t v = chunk_pairs[(size_t)r * n_chunks + b];
      tot += v;
      if (b < blockIdx.x) pre += v;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      tot += __shfl_xor(tot, d, 64);
      pre += __shfl_xor(pre, d, 64);
    }
    if ((threadIdx.x & 63) == 0) {
      if (tot) atomicAdd(&s_total[r], tot);
      if (pre) atomicAdd(&s_before[r], pre);
    }
  }
  __syncthreads();
  MSMZ_STAMP(trace, 2);
  if (threadIdx.x == 0) {
    uint32_t base = 0;
    for (int r = 0; r < PLAN_RMAX; r++) {
      s_rbase[r] = base;
      base += r < R ? s_total[r] : 0u;
    }
    s_rbase[PLAN_RMAX] = base;
    if (blockIdx.x == 0) {
      meta->rounds = (uint32_t)R;
      meta->n_entries = off[nb];
      for (int r = 0; r < 32; r++) {
        meta->round_pairs[r] = r < R ? s_total[r] : 0u;
        meta->round_base[r] = r < PLAN_RMAX ? s_rbase[r] : base;
      }
    }
  }
  // per round: exclusive scan of the buckets' pair counts (thread t owns PLAN_PER consecutive buckets)
  const uint32_t b0 = threadIdx.x * PLAN_PER;
  uint32_t size[PLAN_PER];
#pragma unroll
  for (int q = 0; q < PLAN_PER; q++) size[q] = b0 + q < nbk ? s_start[b0 + q + 1] - s_start[b0 + q] : 0u;
  for (int r = 0; r < R; r++) {
    uint32_t s = 0;
#pragma unroll
    for (int q = 0; q < PLAN_PER; q++) s += pairs_in_round(size[q], r);
    uint32_t total;
    uint32_t ex = block_exclusive_scan<PLAN_T>(s, &total, s_wave);
    if (r < PLAN_RL) {
#pragma unroll
      for (int q = 0; q < PLAN_PER; q++) {
        s_pref[r][b0 + q] = ex;
        ex += pairs_in_round(size[q], r);
      }
      if (threadIdx.x == PLAN_T - 1) s_pref[r][PLAN_CHUNK] = ex;
    } else {
      s_pair[(r - PLAN_RL) * PLAN_T + threadIdx.x] = ex;
    }
  }
  __syncthreads();
  MSMZ_STAMP(trace, 3);
  // location of the element at relative position `pos` of chunk bucket b before round r (see the header); prr = this
  // thread's running pair numbers for the rounds >= PLAN_RL (only meaningful on the bucket-by-bucket path)
  auto location = [&](uint32_t b, uint32_t st, uint32_t sz, uint32_t pos, int r) -> uint32_t {
    const uint32_t rem = sz - pos;
    if (r == 0 || rem == 1) {
      const uint32_t rf = refs[st + pos];
      return (rf & REF_IDX) | (rf & REF_NEG) | LOC_ORIG;
    }
    int rr = 31 - __builtin_clz(rem - 1);
    if (rr > r - 1) rr = r - 1;
    const uint32_t pr = rr < PLAN_RL ? s_pref[rr][b] : s_pair[(rr - PLAN_RL) * PLAN_T + threadIdx.x];
    return s_rbase[rr] + s_before[rr] + pr + (pos >> (rr + 1));
  };
  // rounds < PLAN_RL: one thread per pair, descriptors written in pair order (coalesced).  The pair -> bucket map
  // of a tile of pairs is expanded into LDS by the buckets' owner threads (a search per pair would be 10 dependent
  // LDS reads).
  for (int r = 0; r < RL; r++) {
    const uint32_t np = s_pref[r][PLAN_CHUNK];
    uint2* d = desc + s_rbase[r] + s_before[r];
    for (uint32_t tile0 = 0; tile0 < np; tile0 += PLAN_TILE) {
      const uint32_t tile1 = tile0 + PLAN_TILE < np ? tile0 + PLAN_TILE : np;
#pragma unroll
      for (int q = 0; q < PLAN_PER; q++) {
        const uint32_t b = b0 + q;
        uint32_t lo = s_pref[r][b], hi = s_pref[r][b + 1];
        lo = lo > tile0 ? lo : tile0;
        hi = hi < tile1 ? hi : tile1;
        for (uint32_t t = lo; t < hi; t++) s_owner[t - tile0] = (uint16_t)b;
      }
      __syncthreads();
      if (r == 0) {
        // round 0 (half of all pairs): both operands are original points, entries 2j and 2j + 1 of the bucket
#pragma unroll 4
        for (uint32_t t = tile0 + threadIdx.x; t < tile1; t += PLAN_T) {   // unrolled: four pairs' reference loads in flight
          const uint32_t b = s_owner[t - tile0];
          const uint32_t e = s_start[b] + ((t - s_pref[0][b]) << 1);
          const uint32_t r0 = refs[e], r1 = refs[e + 1];
          d[t] = make_uint2((r0 & (REF_IDX | REF_NEG)) | LOC_ORIG, (r1 & (REF_IDX | REF_NEG)) | LOC_ORIG);
        }
      } else {
#pragma unroll 4
        for (uint32_t t = tile0 + threadIdx.x; t < tile1; t += PLAN_T) {
          const uint32_t b = s_owner[t - tile0];
          const uint32_t st = s_start[b], sz = s_start[b + 1] - st;
          const uint32_t a = (t - s_pref[r][b]) << (r + 1);
          d[t] = make_uint2(location(b, st, sz, a, r), location(b, st, sz, a + (1u << r), r));
        }
      }
      __syncthreads();
    }
    if (r < 6) MSMZ_STAMP(trace, 4 + r);
  }
  // rounds >= PLAN_RL (very long buckets) and the per-bucket records: bucket by bucket
#pragma unroll 1
  for (int q = 0; q < PLAN_PER; q++) {
    const uint32_t b = b0 + q;
    if (b >= nbk) break;
    const uint32_t st = s_start[b], sz = s_start[b + 1] - st;
    for (int r = PLAN_RL; r < R; r++) {
      const uint32_t np = pairs_in_round(sz, r);
      if (np == 0) break;
      uint2* d = desc + s_rbase[r] + s_before[r] + s_pair[(r - PLAN_RL) * PLAN_T + threadIdx.x];
      for (uint32_t j = 0; j < np; j++) {
        const uint32_t a = j << (r + 1);
        d[j] = make_uint2(location(b, st, sz, a, r), location(b, st, sz, a + (1u << r), r));
      }
    }
    // what the R rounds leave of the bucket: ceil(size / 2^R) <= 4 partial sums
    uint4 fin = make_uint4(LOC_NONE, LOC_NONE, LOC_NONE, LOC_NONE);
    if (sz > 0) fin.x = location(b, st, sz, 0, R);
    if (sz > (1u << R)) fin.y = location(b, st, sz, 1u << R, R);
    if (sz > (2u << R)) fin.z = location(b, st, sz, 2u << R, R);
    if (sz > (3u << R)) fin.w = location(b, st, sz, 3u << R, R);
    bfin[g0 + b] = fin;
    for (int r = PLAN_RL; r < R; r++) s_pair[(r - PLAN_RL) * PLAN_T + threadIdx.x] += pairs_in_round(sz, r);   // -> next bucket
  }
  MSMZ_STAMP(trace, 10);
}

}  // namespace msmz
