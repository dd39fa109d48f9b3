// Saturated-limb CIOS Montgomery multiplication (NW x 32-bit words = NW/2 x 64-bit limbs) -- the
// schedule BASELINE.json's north_star names ("4x64-bit-limb CIOS").  Kept ONLY for the A/B
// measurement in tools/ubench_fp.hip: on gfx950 every carry instruction (v_add_co/v_addc_co)
// issues at the same half rate as v_mad_u64_u32 (profiles/r01_ubench_int_instr_rates.txt), so
// this runs slower than the carry-free lazy-limb product scanning of fp.h, which the kernels use.
#pragma once
#include <cstdint>
#include "fp.h"

namespace msmz {

// r = a*b/2^(32*NW) mod p, inputs/outputs canonical words in [0,p).  MU32 = -p^-1 mod 2^32.
template <class F, uint32_t MU32>
MSMZ_HD void cios_mul(uint32_t* r, const uint32_t* a, const uint32_t* b) {
  constexpr int NW = F::NW;
  uint32_t t[NW + 2];
#pragma unroll
  for (int i = 0; i < NW + 2; i++) t[i] = 0;
#pragma unroll
  for (int i = 0; i < NW; i++) {
    uint64_t c = 0;
#pragma unroll
    for (int j = 0; j < NW; j++) {
      uint64_t x = (uint64_t)a[i] * b[j] + t[j] + c;
      t[j] = (uint32_t)x;
      c = x >> 32;
    }
    uint64_t x = (uint64_t)t[NW] + c;
    t[NW] = (uint32_t)x;
    t[NW + 1] = (uint32_t)(x >> 32);
    uint32_t m = t[0] * MU32;
    c = ((uint64_t)m * F::PW[0] + t[0]) >> 32;
#pragma unroll
    for (int j = 1; j < NW; j++) {
      uint64_t y = (uint64_t)m * F::PW[j] + t[j] + c;
      t[j - 1] = (uint32_t)y;
      c = y >> 32;
    }
    x = (uint64_t)t[NW] + c;
    t[NW - 1] = (uint32_t)x;
    t[NW] = t[NW + 1] + (uint32_t)(x >> 32);
  }
  uint32_t d[NW];
  uint32_t br = words_sub<NW>(d, t, F::PW);
#pragma unroll
  for (int i = 0; i < NW; i++) r[i] = (br && !t[NW]) ? t[i] : d[i];
}

}  // namespace msmz
