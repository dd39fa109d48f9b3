// Field-multiplication throughput on gfx950: lazy-limb product scanning (fp.h, used by the
// kernels) vs saturated CIOS (fp_cios.h).  Prints modmul/s and cycles per wave-modmul.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../msm_zprize_amd/csrc/constants_gen.h"
#include "../msm_zprize_amd/csrc/fp.h"
#include "../msm_zprize_amd/csrc/fp_cios.h"
using namespace msmz;
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__);return 1;}}while(0)
constexpr int ITERS = 512;

template <class F, int MODE> __global__ void __launch_bounds__(256) k(uint32_t* io) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t w[F::NW];
  for (int i = 0; i < F::NW; i++) w[i] = io[i] + (i == 0 ? tid : 0);
  if (MODE == 0 || MODE == 2) {
    Fe<F> x, y; fe_unpack<F>(x, w); y = x; y.l[1] ^= 5;
    for (int it = 0; it < ITERS; it++) {
      if (MODE == 0) { fe_mul<F>(x, x, y); fe_mul<F>(y, y, x); }
      else { fe_sqr<F>(x, x); fe_sqr<F>(y, y); fe_add<F>(x, x, y); fe_carry<F>(x);} 
    }
    fe_add<F>(x, x, y); fe_store<F>(w, x);
  } else {
    uint32_t v[F::NW];
    for (int i = 0; i < F::NW; i++) v[i] = w[i] ^ 5;
    for (int it = 0; it < ITERS; it++) { cios_mul<F, 0xffffffffu>(w, w, v); cios_mul<F, 0xffffffffu>(v, v, w); }
    for (int i = 0; i < F::NW; i++) w[i] ^= v[i];
  }
  uint32_t s = 0; for (int i = 0; i < F::NW; i++) s ^= w[i];
  if (s == 0x12345) io[0] = s;
}

template <class F, int MODE> int run(const char* name, int bpc) {
  uint32_t* io; CK(hipMalloc(&io, 64)); CK(hipMemset(io, 0x11, 64));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  int grid = 256 * bpc;
  for (int i = 0; i < 3; i++) k<F, MODE><<<grid, 256>>>(io);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int r = 0; r < 5; r++) k<F, MODE><<<grid, 256>>>(io);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
  double muls = (double)grid * 256 * ITERS * 2;
  double wave_muls_per_simd = muls / 64 / 1024;
  printf("%-28s blocks/CU=%d  %.3f ms  %.2f Gmodmul/s  %.0f cycles/wave-modmul/SIMD @2.4GHz\n", name, bpc, ms, muls / ms / 1e6, ms * 1e-3 * 2.4e9 / wave_muls_per_simd);
  return 0;
}
int main() {
  for (int bpc : {2, 4, 6, 8}) {
    run<Bls377Fp, 0>("bls377 lazy14x28 mul", bpc);
    run<Bls377Fp, 2>("bls377 lazy14x28 sqr", bpc);
    run<Bls377Fp, 1>("bls377 cios 12x32 mul", bpc);
    run<PallasFp, 0>("pallas lazy9x29 mul", bpc);
    run<PallasFp, 1>("pallas cios 8x32 mul", bpc);
    run<Bls381Fp, 0>("bls381 lazy14x28 mul", bpc);
  }
  return 0;
}
