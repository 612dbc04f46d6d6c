// Micro-benchmark: HBM write rate of the S16 epilogue's store shapes (hipcc --offload-arch=gfx950 -O3 -o store_patterns ...)
//   A  MFMA 32x32 epilogue: lane = pixel (256-B records), two 16-B stores per 32-B group, lane halves on neighbouring groups
//   B  MFMA 16x16 epilogue: 16 pixels x 4 groups per instruction
//   C  fully coalesced: 64 lanes x 16 B consecutive
//   D  as A, but the two stores of an instruction pair cover hi|lo of ONE group per lane pair (32 B contiguous per pair)
//   E  4 lanes cover 64 contiguous bytes (16 pixels x 64 B per instruction)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int P>
__global__ __launch_bounds__(256) void store_kernel(float* __restrict__ y, int tiles) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const f32x4 v = {(float)lane, 1.f, 2.f, (float)blockIdx.x};
  // a tile = 8 rows x 32 pixels x 256 B = 64 KB; wave w owns rows 2w, 2w+1
  for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
    float* base = y + (size_t)t * 16384;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      float* row = base + (wave * 2 + r) * 2048;          // 32 pixels x 64 floats
      if (P == 0) {
        const int px = lane & 31, h = lane >> 5;
#pragma unroll
        for (int o = 0; o < 4; ++o) {
          f32x4* yp = reinterpret_cast<f32x4*>(row + px * 64 + (2 * o + h) * 8);
          yp[0] = v; yp[1] = v;
        }
      } else if (P == 1) {
        const int l15 = lane & 15, g4 = lane >> 4;
#pragma unroll
        for (int pt = 0; pt < 2; ++pt)
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            f32x4* yp = reinterpret_cast<f32x4*>(row + (pt * 16 + l15) * 64 + (4 * u + g4) * 8);
            yp[0] = v; yp[1] = v;
          }
      } else if (P == 2) {
#pragma unroll
        for (int o = 0; o < 8; ++o) *reinterpret_cast<f32x4*>(row + o * 256 + lane * 4) = v;
      } else if (P == 3) {
        const int px = lane & 31, h = lane >> 5;
#pragma unroll
        for (int g = 0; g < 8; ++g) *reinterpret_cast<f32x4*>(row + px * 64 + g * 8 + h * 4) = v;
      } else {
        const int px = lane >> 2, q = lane & 3;
#pragma unroll
        for (int pp = 0; pp < 2; ++pp)
#pragma unroll
          for (int o = 0; o < 4; ++o) *reinterpret_cast<f32x4*>(row + (pp * 16 + px) * 64 + o * 16 + q * 4) = v;
      }
    }
  }
}

template <int P>
static void run(float* y, int tiles, int grid, const char* name) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(store_kernel<P>, dim3(grid), dim3(256), 0, 0, y, tiles);
  hipEventRecord(e0, 0);
  const int reps = 20;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(store_kernel<P>, dim3(grid), dim3(256), 0, 0, y, tiles);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)tiles * 65536;
  printf("%-34s grid %5d: %7.1f us  %6.2f TB/s\n", name, grid, ms / reps * 1e3, bytes / (ms / reps * 1e-3) / 1e12);
}

int main() {
  const int tiles = 4096;                          // 268 MB
  float* y; hipMalloc(&y, (size_t)tiles * 65536);
  for (int grid : {4096, 1024, 512}) {
    run<0>(y, tiles, grid, "A 32x32 epilogue (16 B @ 256 B)");
    run<1>(y, tiles, grid, "B 16x16 epilogue");
    run<2>(y, tiles, grid, "C coalesced 1 KB / instruction");
    run<3>(y, tiles, grid, "D lane pair = 32 B");
    run<4>(y, tiles, grid, "E 4 lanes = 64 B");
  }
  return 0;
}
