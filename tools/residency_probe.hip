// How many workgroups of T threads does a CU really hold at once?  Every workgroup stamps the wall clock, spins ~20 us
// and stamps again with its hardware id; the host counts the largest number of workgroups alive together on one CU.
// (Evidence for the sort kernels' geometry, DESIGN.md section 6: hipcc --offload-arch=gfx950 -O2 -o /tmp/rp tools/residency_probe.hip)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <map>
#include <vector>
template <int T>
__global__ void __launch_bounds__(T) k_spin(unsigned long long* out, int use_barrier) {
  extern __shared__ int s[];
  s[threadIdx.x] = threadIdx.x;
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < 2000) {
    if (use_barrier) __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[blockIdx.x * 4 + 0] = t0;
    out[blockIdx.x * 4 + 1] = wall_clock64();
    out[blockIdx.x * 4 + 2] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) |
                              ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
    out[blockIdx.x * 4 + 3] = s[(threadIdx.x + 1) % T];
  }
}
template <int T>
void run(int wgs, size_t lds, int use_barrier, unsigned long long* d) {
  hipFuncSetAttribute((const void*)k_spin<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  int api = 0;
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&api, k_spin<T>, T, lds);
  hipLaunchKernelGGL(k_spin<T>, dim3(wgs), dim3(T), lds, 0, d, use_barrier);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(wgs * 4);
  hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
  std::map<unsigned long long, std::vector<std::pair<unsigned long long, int>>> ev;
  unsigned long long tmin = ~0ull, tmax = 0;
  for (int i = 0; i < wgs; i++) {
    const unsigned long long hw = h[i * 4 + 2];
    const unsigned long long key = ((hw >> 32) & 0xf) * 4096 + ((hw >> 13) & 7) * 16 + ((hw >> 8) & 0xf);
    ev[key].push_back({h[i * 4], +1});
    ev[key].push_back({h[i * 4 + 1], -1});
    tmin = std::min(tmin, h[i * 4]);
    tmax = std::max(tmax, h[i * 4 + 1]);
  }
  int most = 0;
  for (auto& kv : ev) {
    std::sort(kv.second.begin(), kv.second.end());
    int cur = 0;
    for (auto& e : kv.second) {
      cur += e.second;
      most = std::max(most, cur);
    }
  }
  printf("T=%4d  LDS=%6zu  barrier=%d  workgroups=%4d  API says %d per CU   CUs seen %zu   most alive on one CU %d   span %.1f us\n", T, lds,
         use_barrier, wgs, api, ev.size(), most, (tmax - tmin) / 100.0);
}
int main() {
  unsigned long long* d;
  hipMalloc(&d, 8192 * 4 * 8);
  for (int b = 0; b < 2; b++) {
    run<1024>(512, 4096, b, d);
    run<1024>(512, 40000, b, d);
    run<1024>(1024, 40000, b, d);
    run<512>(1024, 40000, b, d);
    run<512>(2048, 20000, b, d);
    run<256>(2048, 20000, b, d);
  }
  return 0;
}
