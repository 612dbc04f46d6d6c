// Micro-benchmark: where and when the workgroups of a grid start.  One record per workgroup: XCC id, HW_ID (SE / CU /
// SIMD / wave slot), start and end time (s_memrealtime, 100 MHz), for a kernel with the halo-patch kernel's resources
// (256 or 512 threads, 77 or 144 KB of LDS), each workgroup spinning ~T us.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/census.hip -o tools/micro/census && tools/micro/census [grid] [threads] [lds_kb] [spin_us]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <map>
#include <vector>
#include <algorithm>

struct Rec { unsigned xcc, hwid; unsigned long long t0, t1; };

__global__ void census(Rec* out, int spin_ticks) {
  extern __shared__ float smem[];
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) smem[0] = 1.f;
  unsigned long long t1 = t0;
  while ((long long)(t1 - t0) < spin_ticks) t1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) {
    Rec r;
    r.xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11)) & 15;        // HW_REG_XCC_ID
    r.hwid = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));             // HW_REG_HW_ID
    r.t0 = t0, r.t1 = t1;
    out[blockIdx.x] = r;
  }
}

int main(int argc, char** argv) {
  const int grid = argc > 1 ? atoi(argv[1]) : 1024, threads = argc > 2 ? atoi(argv[2]) : 256;
  const int lds = (argc > 3 ? atoi(argv[3]) : 77) * 1024, spin = (argc > 4 ? atoi(argv[4]) : 20) * 100;
  Rec* d;
  hipMalloc(&d, grid * sizeof(Rec));
  hipFuncSetAttribute((const void*)census, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  for (int it = 0; it < 2; ++it) hipLaunchKernelGGL(census, dim3(grid), dim3(threads), lds, 0, d, spin);
  hipDeviceSynchronize();
  std::vector<Rec> h(grid);
  hipMemcpy(h.data(), d, grid * sizeof(Rec), hipMemcpyDeviceToHost);
  unsigned long long tmin = ~0ull;
  for (auto& r : h) tmin = std::min(tmin, r.t0);
  // HW_ID (gfx9): wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13] ...
  std::map<unsigned, std::vector<int>> bycu;
  for (int b = 0; b < grid; ++b) {
    const unsigned cu = (h[b].xcc << 8) | (((h[b].hwid >> 13) & 7) << 5) | (((h[b].hwid >> 12) & 1) << 4) | ((h[b].hwid >> 8) & 15);
    bycu[cu].push_back(b);
  }
  printf("grid %d threads %d lds %d KB spin %d us: %zu distinct CUs\n", grid, threads, lds / 1024, spin / 100, bycu.size());
  int shown = 0;
  for (auto& kv : bycu) {
    if (shown++ >= 12) break;
    printf("cu %04x (xcc %u):", kv.first, kv.first >> 8);
    for (int b : kv.second) printf("  b%-4d t0=%6.2f us", b, (h[b].t0 - tmin) / 100.0);
    printf("\n");
  }
  // how often are the co-resident first two blocks of a CU b and b + 256 / b and b + 8 / other?
  int d256 = 0, d8 = 0, other = 0;
  for (auto& kv : bycu) {
    std::vector<int> v = kv.second;
    std::sort(v.begin(), v.end(), [&](int a, int b) { return h[a].t0 < h[b].t0; });
    if (v.size() >= 2) {
      const int df = abs(v[1] - v[0]);
      if (df == 256) ++d256; else if (df == 8) ++d8; else ++other;
    }
  }
  printf("first two blocks on a CU differ by 256: %d, by 8: %d, other: %d\n", d256, d8, other);
  for (int b : {0, 1, 8, 255, 256, 257, 511, 512, 513, 1023})
    if (b < grid) printf("b%-4d xcc %u hwid %08x t0 %.2f t1 %.2f\n", b, h[b].xcc, h[b].hwid, (h[b].t0 - tmin) / 100.0, (h[b].t1 - tmin) / 100.0);
  return 0;
}
