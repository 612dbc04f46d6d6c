// What MFMA rate does the board sustain at its power cap?  Register-resident fp16 operands, no memory traffic inside
// the loop: 2 waves per SIMD on every CU issue v_mfma_f32_32x32x16_f16 (or 16x16x32) back to back on 8 independent
// accumulators, rotating through 2 x 4 distinct operand fragments so that consecutive instructions see different data
// (as in a real GEMM).  Operand data: random in [-1, 1] | random with half the A elements zero (post-ReLU) | constant |
// (round 5, `./mfma_power 2 new`) random with the low 5 / 6 / 8 mantissa bits of the B operand - or of both - zero: what a
// lo half rounded to 6 / 5 / 3 significant bits would cost the matrix pipe.
// Board power and shader clock are sampled from the amdgpu hwmon files of this process's GPU while it runs.
//     hipcc --offload-arch=gfx950 -O3 -o mfma_power tools/micro/mfma_power.hip && ./mfma_power [seconds per case]
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dirent.h>
#include <string>
#include <thread>
#include <unistd.h>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE, int ORDER = 0>
__global__ __launch_bounds__(512, 2) void mfma_loop(const f16x8* __restrict__ src, float* __restrict__ out, int iters) {
  const int tid = threadIdx.x + blockIdx.x * 512;
  f16x8 a[2], b[4];
  for (int i = 0; i < 2; ++i) a[i] = src[(size_t)(i * 6 + 0) * 131072 + (tid & 131071)];
  for (int i = 0; i < 4; ++i) b[i] = src[(size_t)(i + 2) * 131072 + (tid & 131071)];
  if (SHAPE == 0) {
    f32x16 acc[2][4];
    for (int i = 0; i < 2; ++i)
      for (int j = 0; j < 4; ++j)
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
      if (ORDER == 0) {                                        // consecutive MFMAs share the A operand
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i][j], 0, 0, 0);
      } else {                                                 // no operand shared between consecutive MFMAs
        constexpr int si[8] = {0, 1, 0, 1, 0, 1, 0, 1}, sj[8] = {0, 1, 2, 3, 1, 0, 3, 2};
#pragma unroll
        for (int q = 0; q < 8; ++q)
          acc[si[q]][sj[q]] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[si[q]], b[sj[q]], acc[si[q]][sj[q]], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i)
      for (int j = 0; j < 4; ++j)
        for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[tid] = s;
  } else {
    f32x4 acc[2][4];
    for (int i = 0; i < 2; ++i)
      for (int j = 0; j < 4; ++j)
        for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i)
      for (int j = 0; j < 4; ++j)
        for (int r = 0; r < 4; ++r) s += acc[i][j][r];
    out[tid] = s;
  }
}

static std::string hwmon_dir() {
  char bdf[64] = {0};
  if (hipDeviceGetPCIBusId(bdf, sizeof bdf, 0) != hipSuccess) return "";
  for (char* p = bdf; *p; ++p) *p = (char)tolower(*p);
  DIR* d = opendir("/sys/class/drm");
  if (!d) return "";
  std::string found;
  while (dirent* e = readdir(d)) {
    if (strncmp(e->d_name, "card", 4) || strchr(e->d_name, '-')) continue;
    std::string dev = std::string("/sys/class/drm/") + e->d_name + "/device";
    char real[512];
    if (!realpath(dev.c_str(), real)) continue;
    std::string r(real);
    for (auto& c : r) c = (char)tolower(c);
    if (r.find(bdf) == std::string::npos) continue;
    DIR* h = opendir((dev + "/hwmon").c_str());
    if (!h) continue;
    while (dirent* g = readdir(h))
      if (!strncmp(g->d_name, "hwmon", 5)) found = dev + "/hwmon/" + g->d_name;
    closedir(h);
  }
  closedir(d);
  return found;
}

static double read_num(const std::string& path) {
  FILE* f = fopen(path.c_str(), "r");
  if (!f) return -1;
  double v = -1;
  if (fscanf(f, "%lf", &v) != 1) v = -1;
  fclose(f);
  return v;
}

template <int SHAPE, int ORDER = 0>
static void run_case(const char* name, const f16x8* src, float* out, double seconds, const std::string& hw) {
  const int blocks = 256, iters = 20000;                       // one 8-wave workgroup per CU = 2 waves per SIMD
  hipLaunchKernelGGL((mfma_loop<SHAPE, ORDER>), dim3(blocks), dim3(512), 0, 0, src, out, iters);
  hipDeviceSynchronize();
  std::atomic<bool> stop{false};
  std::vector<double> pw, fq;
  std::thread th([&] {
    while (!stop.load()) {
      if (!hw.empty()) {
        pw.push_back(read_num(hw + "/power1_input"));
        fq.push_back(read_num(hw + "/freq1_input"));
      }
      usleep(20000);
    }
  });
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  auto t0 = std::chrono::steady_clock::now();
  long launches = 0;
  hipEventRecord(e0, 0);
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
    for (int i = 0; i < 4; ++i) hipLaunchKernelGGL((mfma_loop<SHAPE, ORDER>), dim3(blocks), dim3(512), 0, 0, src, out, iters);
    launches += 4;
    hipDeviceSynchronize();
  }
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  stop.store(true);
  th.join();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double flop = (double)launches * blocks * 8 /*waves*/ * iters * 8 /*mfma*/ * (SHAPE == 0 ? 32768.0 : 16384.0);
  double p = 0, f = 0;
  size_t n0 = pw.size() / 4, n = 0;
  for (size_t i = n0; i < pw.size(); ++i) { p += pw[i]; f += fq[i]; ++n; }
  printf("%-46s %8.1f TFLOP/s issued  power %7.1f W  sclk %7.1f MHz  (%ld launches, %.2f s)\n", name,
         flop / (ms * 1e-3) / 1e12, n ? p / n / 1e6 : -1, n ? f / n / 1e6 : -1, launches, ms * 1e-3);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const double seconds = argc > 1 ? atof(argv[1]) : 3.0;
  const std::string hw = hwmon_dir();
  printf("hwmon: %s  cap %.0f W\n", hw.c_str(), hw.empty() ? -1 : read_num(hw + "/power1_cap") / 1e6);
  const size_t n = (size_t)8 * 131072;                         // 8 planes of fragments
  std::vector<f16x8> h(n);
  f16x8* src;
  float* out;
  hipMalloc(&src, n * sizeof(f16x8));
  hipMalloc(&out, 256 * 512 * sizeof(float));
  const bool only_new = argc > 2 && !strcmp(argv[2], "new");
  for (int mode = only_new ? 5 : 0; mode < (only_new ? 10 : 5); ++mode) {
    srand(1234);
    // modes 5..9: 0 = reference (random), then the low `zb` mantissa bits cleared in B (planes 2..5) or in both operands
    const int zb = mode == 6 ? 5 : mode == 7 ? 6 : mode == 8 ? 8 : mode == 9 ? 6 : 0;
    for (size_t i = 0; i < n; ++i)
      for (int k = 0; k < 8; ++k) {
        float v = mode == 2 ? 0.5f : (float)rand() / RAND_MAX * 2.f - 1.f;
        if (mode == 1 && (i / 131072 == 0 || i / 131072 == 6) && (rand() & 1)) v = 0.f;    // the A fragments: half zeros
        if (mode == 3 && (i / 131072 >= 2 && i / 131072 <= 5) && (rand() & 1)) v = 0.f;    // the B fragments: half zeros
        if (mode == 4 && (rand() & 1)) v = 0.f;                                            // both
        _Float16 hv = (_Float16)v;
        if (zb && (mode == 9 || (i / 131072 >= 2 && i / 131072 <= 5))) {
          unsigned short bits;
          memcpy(&bits, &hv, 2);
          bits &= (unsigned short)~((1u << zb) - 1u);
          memcpy(&hv, &bits, 2);
        }
        h[i][k] = hv;
      }
    hipMemcpy(src, h.data(), n * sizeof(f16x8), hipMemcpyHostToDevice);
    const char* tag = mode == 0 ? "random operands" : mode == 1 ? "random, half of A zero" : mode == 2 ? "constant operands"
                      : mode == 3 ? "random, half of B zero" : mode == 4 ? "random, half of A and of B zero"
                      : mode == 5 ? "random operands (reference)" : mode == 6 ? "random, low 5 mantissa bits of B zero"
                      : mode == 7 ? "random, low 6 mantissa bits of B zero" : mode == 8 ? "random, low 8 mantissa bits of B zero"
                      : "random, low 6 mantissa bits of A and B zero";
    char name[128];
    snprintf(name, sizeof name, "v_mfma_f32_32x32x16_f16, %s", tag);
    run_case<0>(name, src, out, seconds, hw);
    snprintf(name, sizeof name, "v_mfma_f32_16x16x32_f16, %s", tag);
    run_case<1>(name, src, out, seconds, hw);
    if (mode == 0) {
      snprintf(name, sizeof name, "32x32x16, %s, no operand shared", tag);
      run_case<0, 1>(name, src, out, seconds, hw);
    }
  }
  return 0;
}
