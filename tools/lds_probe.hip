// Prints the per-workgroup LDS figures HIP reports for the device and tries dynamic-LDS launches with and without
// hipFuncAttributeMaxDynamicSharedMemorySize (evidence for Engine::raise_lds_limit).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_probe(int* out) {
  extern __shared__ int s[];
  s[threadIdx.x] = threadIdx.x;
  __syncthreads();
  if (threadIdx.x == 0) *out = s[63];
}
int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  printf("%s: sharedMemPerBlock=%zu sharedMemPerBlockOptin=%zu maxSharedMemoryPerMultiProcessor=%zu\n", p.gcnArchName,
         (size_t)p.sharedMemPerBlock, (size_t)p.sharedMemPerBlockOptin, (size_t)p.maxSharedMemoryPerMultiProcessor);
  int* d;
  hipMalloc(&d, 4);
  const size_t sizes[] = {32768, 65536, 65540, 98304, 131072, 159744, 163840, 163844};
  for (int pass = 0; pass < 2; pass++) {
    for (size_t sz : sizes) {
      hipError_t ea = hipSuccess;
      if (pass == 1) ea = hipFuncSetAttribute((const void*)k_probe, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sz);
      (void)hipGetLastError();
      hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), sz, 0, d);
      hipError_t el = hipGetLastError();
      hipError_t es = hipDeviceSynchronize();
      printf("%s dynamic LDS %zu: setattr=%s launch=%s sync=%s\n", pass ? "with attribute" : "no attribute  ", sz,
             hipGetErrorName(ea), hipGetErrorName(el), hipGetErrorName(es));
    }
  }
  return 0;
}
