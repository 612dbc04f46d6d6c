// Micro-benchmark: LDS-DMA streaming rate of halo patches out of a 268-MB S16 NHWC image (16 x 256 x 256 x 64 channels,
// 256 B per pixel + halo), persistent 512-thread workgroups, D patches in flight, nothing computed.
//   shape 0: 10 x 34 pixels x 128 B (one 32-channel block; the two blocks of a tile are separate patches)   43.5 KB
//   shape 1: 6 x 34 pixels x 256 B (both blocks, 4-row tile)                                                  52 KB
//   shape 2: 43.5 KB contiguous
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

constexpr int NT = 512;
__device__ __forceinline__ void wait_r6(int young) {
  if (young >= 2) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else if (young == 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
__device__ __forceinline__ void wait_r7(int young) {
  if (young >= 2) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
  else if (young == 1) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
__device__ __forceinline__ const float* unit_base(int shape, const float* x, int64_t x_bs, int64_t x_rs, int g, int pieces) {
  if (shape == 0) {
    const int tile = g >> 1, cc = g & 1;
    const int tx = tile & 7, ty = (tile >> 3) & 31, b = tile >> 8;
    return x + b * x_bs + (int64_t)(ty * 8) * x_rs + tx * 32 * 64 + cc * 32;
  } else if (shape == 1) {
    const int tx = g & 7, ty = (g >> 3) & 63, b = g >> 9;
    return x + b * x_bs + (int64_t)(ty * 4) * x_rs + tx * 32 * 64;
  }
  return x + (int64_t)g * (pieces * 4);
}
template <int SHAPE, int D>
__device__ __forceinline__ void stream_body(const float* __restrict__ x, int64_t x_bs, int64_t x_rs, int units, float* out) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int PIECES = SHAPE == 1 ? 6 * 34 * 16 : 340 * 8;        // 16-byte pieces of a patch
  constexpr int R = (PIECES + NT - 1) / NT;
  constexpr int STAGE = R * NT * 4;                                  // floats
  const int tid = threadIdx.x, wave = tid >> 6;
  int off[R];
#pragma unroll
  for (int j = 0; j < R; ++j) {
    int p = j * NT + tid;
    p = p < PIECES ? p : PIECES - 1;
    if (SHAPE == 0) {
      const int hp = p >> 3, hy = hp / 34, hx = hp - hy * 34;
      off[j] = (int)(hy * x_rs + hx * 64) + 4 * (p & 7);
    } else if (SHAPE == 1) {
      const int hp = p >> 4, hy = hp / 34, hx = hp - hy * 34;
      off[j] = (int)(hy * x_rs + hx * 64) + 4 * (p & 15);
    } else {
      off[j] = 4 * p;
    }
  }
  const int G = gridDim.x;
  int issued = 0, stage = 0;
#define ISSUE(v)                                                                                          \
  {                                                                                                       \
    const float* s_ = unit_base(SHAPE, x, x_bs, x_rs, blockIdx.x + (v) * G, PIECES);                                                                            \
    _Pragma("unroll") for (int j = 0; j < R; ++j)                                                         \
      __builtin_amdgcn_global_load_lds(s_ + off[j], smem + stage * STAGE + (j * NT + wave * 64) * 4, 16, 0, 0); \
    stage = stage + 1 == D ? 0 : stage + 1;                                                               \
    ++issued;                                                                                             \
  }
  for (int v = 0; v < D - 1 && v < units; ++v) ISSUE(v);
  float acc = 0.f;
  for (int u = 0; u < units; ++u) {
    if (issued < units) ISSUE(issued);
    // everything but the youngest min(D - 1, remaining) patches has landed
    const int young = issued - u - 1;
    if (R == 6) wait_r6(young); else wait_r7(young);
    __builtin_amdgcn_s_barrier();
    acc += smem[(u % D) * STAGE + tid];
  }
  if (acc == 12345.678f) out[0] = acc;
}

__global__ __launch_bounds__(NT, 1) void k01(const float* x, int64_t a, int64_t b, int u, float* o) { stream_body<0, 1>(x, a, b, u, o); }
__global__ __launch_bounds__(NT, 1) void k02(const float* x, int64_t a, int64_t b, int u, float* o) { stream_body<0, 2>(x, a, b, u, o); }
__global__ __launch_bounds__(NT, 1) void k03(const float* x, int64_t a, int64_t b, int u, float* o) { stream_body<0, 3>(x, a, b, u, o); }
__global__ __launch_bounds__(NT, 1) void k11(const float* x, int64_t a, int64_t b, int u, float* o) { stream_body<1, 1>(x, a, b, u, o); }
__global__ __launch_bounds__(NT, 1) void k12(const float* x, int64_t a, int64_t b, int u, float* o) { stream_body<1, 2>(x, a, b, u, o); }
__global__ __launch_bounds__(NT, 1) void k13(const float* x, int64_t a, int64_t b, int u, float* o) { stream_body<1, 3>(x, a, b, u, o); }
__global__ __launch_bounds__(NT, 1) void k21(const float* x, int64_t a, int64_t b, int u, float* o) { stream_body<2, 1>(x, a, b, u, o); }
__global__ __launch_bounds__(NT, 1) void k22(const float* x, int64_t a, int64_t b, int u, float* o) { stream_body<2, 2>(x, a, b, u, o); }
__global__ __launch_bounds__(NT, 1) void k23(const float* x, int64_t a, int64_t b, int u, float* o) { stream_body<2, 3>(x, a, b, u, o); }

template <int SHAPE, int D, typename K>
static void run(K kern, const float* x, float* out, const char* name) {
  constexpr int PIECES = SHAPE == 1 ? 6 * 34 * 16 : 340 * 8;
  constexpr int R = (PIECES + NT - 1) / NT;
  const size_t lds = (size_t)D * R * NT * 16;
  if (lds > 160 * 1024) { printf("%-40s D %d: LDS %zu KB too large\n", name, D, lds / 1024); return; }
  hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int64_t x_rs = 258 * 64, x_bs = 258 * x_rs;
  const int total = SHAPE == 0 ? 16 * 256 * 2 : (SHAPE == 1 ? 16 * 512 : (int)(16LL * 256 * 256 * 256 / (PIECES * 16)));
  const int G = 256, units = total / G;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(G), dim3(NT), lds, 0, x + x_rs + 64, x_bs, x_rs, units, out);
  hipEventRecord(e0, 0);
  const int reps = 10;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(G), dim3(NT), lds, 0, x + x_rs + 64, x_bs, x_rs, units, out);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms / reps * 1e3;
  printf("%-40s D %d: %7.1f us  %5.2f TB/s of patch bytes, %5.2f TB/s of image bytes\n", name, D, us,
         (double)units * G * PIECES * 16 / us / 1e6, 16.0 * 256 * 256 * 256 / us / 1e6);
}

int main() {
  float* x; float* out;
  const size_t n = (size_t)16 * 258 * 258 * 64 + 1024;
  hipMalloc(&x, n * 4 + (64 << 20)); hipMalloc(&out, 64);
  hipMemset(x, 0, n * 4 + (64 << 20));
  run<0, 1>(k01, x, out, "10x34 px x 128 B (two blocks apart)");
  run<0, 2>(k02, x, out, "10x34 px x 128 B (two blocks apart)");
  run<0, 3>(k03, x, out, "10x34 px x 128 B (two blocks apart)");
  run<1, 1>(k11, x, out, "6x34 px x 256 B");
  run<1, 2>(k12, x, out, "6x34 px x 256 B");
  run<1, 3>(k13, x, out, "6x34 px x 256 B");
  run<2, 1>(k21, x, out, "43.5 KB contiguous");
  run<2, 2>(k22, x, out, "43.5 KB contiguous");
  run<2, 3>(k23, x, out, "43.5 KB contiguous");
  return 0;
}
