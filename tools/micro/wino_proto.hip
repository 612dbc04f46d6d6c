// Winograd F(2x2, 3x3) prototype, UNFUSED: the two transform kernels around sixteen 1x1 S16 GEMMs.
// Not part of libammc_hip.so: built by tools/micro/wino_proto.py into tools/micro/libwino_proto.so and timed there.
//
//   V[xi][t][c] = (B^T d B)[xi]     d = the 4x4 input window of tile t (output pixels (2ty..2ty+1, 2tx..2tx+1)), fp32
//                                   arithmetic on the joined (hi, lo) values, then split again: S16 in, S16 out
//   M[xi][t][n] = sum_c V[xi][t][c] U[xi][n][c]          sixteen GEMMs (ammc_conv_gemm_s16, ntaps 1, fp32 out)
//   y[2x2 of t][n] = act(scale * (A^T M A) + shift)      S16 NHWC out
// B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1], A^T = [1 1 1 0; 0 1 -1 -1]; U = G g G^T is made on the host in double.
#include "../../ammcnet_aaai2021_amd/csrc/ammc_common.h"
#include <hip/hip_fp16.h>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void join8(const float* p, float (&v)[8]) {
  const h8 hi = *reinterpret_cast<const h8*>(p), lo = *reinterpret_cast<const h8*>(p + 4);
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = (float)hi[i] + (float)lo[i] * (1.f / 2048.f);
}
__device__ __forceinline__ void store8(float* p, const float (&v)[8]) {
  ammc_u4 h, l;
  ammc_s16_split8(v, h, l);
  *reinterpret_cast<ammc_u4*>(p) = h;
  *reinterpret_cast<ammc_u4*>(p + 4) = l;
}

// x: S16 NHWC, halo corner of pixel (0,0); strides in elements.  V: [16][T][C] (S16 rows of C elements).
// thread = (tile t, channel group g): consecutive threads take consecutive groups of one tile (32-byte pieces, coalesced)
__global__ __launch_bounds__(256) void wino_in_kernel(const float* __restrict__ x, int64_t x_bs, int64_t x_rs, int64_t x_ps,
                                                      int B, int H, int W, int C, float* __restrict__ V) {
  const int G = C >> 3, TX = W >> 1, TY = H >> 1;
  const int64_t T = (int64_t)B * TY * TX;
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= T * G) return;
  const int g = (int)(gid % G);
  const int64_t t = gid / G;
  const int tx = (int)(t % TX);
  const int ty = (int)((t / TX) % TY);
  const int b = (int)(t / ((int64_t)TX * TY));
  const float* p0 = x + (int64_t)b * x_bs + (int64_t)(2 * ty) * x_rs + (int64_t)(2 * tx) * x_ps + 8 * g;
  float u[4][4][8];                                 // B^T d, one window column at a time (128 live values, not 256)
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    float d[4][8];
#pragma unroll
    for (int r = 0; r < 4; ++r) join8(p0 + r * x_rs + s * x_ps, d[r]);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      u[0][s][k] = d[0][k] - d[2][k];
      u[1][s][k] = d[1][k] + d[2][k];
      u[2][s][k] = d[2][k] - d[1][k];
      u[3][s][k] = d[1][k] - d[3][k];
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {                     // (B^T d) B
    float o[4][8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      o[0][k] = u[r][0][k] - u[r][2][k];
      o[1][k] = u[r][1][k] + u[r][2][k];
      o[2][k] = u[r][2][k] - u[r][1][k];
      o[3][k] = u[r][1][k] - u[r][3][k];
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) store8(V + ((int64_t)(4 * r + s) * T + t) * C + 8 * g, o[s]);
  }
}

// M: fp32 [16][T][N]; y: S16 NHWC, pixel (0,0)
__global__ __launch_bounds__(256) void wino_out_kernel(const float* __restrict__ M, int B, int H, int W, int N,
                                                       const float* __restrict__ scale, const float* __restrict__ shift, int relu,
                                                       float* __restrict__ y, int64_t y_bs, int64_t y_rs, int64_t y_ps) {
  const int G = N >> 3, TX = W >> 1, TY = H >> 1;
  const int64_t T = (int64_t)B * TY * TX;
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= T * G) return;
  const int g = (int)(gid % G);
  const int64_t t = gid / G;
  const int tx = (int)(t % TX);
  const int ty = (int)((t / TX) % TY);
  const int b = (int)(t / ((int64_t)TX * TY));
  float m[4][4][8];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const float* p = M + ((int64_t)(4 * r + s) * T + t) * N + 8 * g;
      const f32x4 a = *reinterpret_cast<const f32x4*>(p), c = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
      for (int k = 0; k < 4; ++k) m[r][s][k] = a[k], m[r][s][4 + k] = c[k];
    }
  float sc[8], sh[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) sc[k] = scale ? scale[8 * g + k] : 1.f, sh[k] = shift ? shift[8 * g + k] : 0.f;
  float q[2][4][8];                                 // A^T m
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      q[0][s][k] = m[0][s][k] + m[1][s][k] + m[2][s][k];
      q[1][s][k] = m[1][s][k] - m[2][s][k] - m[3][s][k];
    }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    float o[2][8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      o[0][k] = (q[i][0][k] + q[i][1][k] + q[i][2][k]) * sc[k] + sh[k];
      o[1][k] = (q[i][1][k] - q[i][2][k] - q[i][3][k]) * sc[k] + sh[k];
      if (relu) o[0][k] = fmaxf(o[0][k], 0.f), o[1][k] = fmaxf(o[1][k], 0.f);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j)
      store8(y + (int64_t)b * y_bs + (int64_t)(2 * ty + i) * y_rs + (int64_t)(2 * tx + j) * y_ps + 8 * g, o[j]);
  }
}

extern "C" int wino_input_transform(const float* x, int64_t x_bs, int64_t x_rs, int64_t x_ps, int B, int H, int W, int C,
                                    float* V, void* stream) {
  const int64_t n = (int64_t)B * (H / 2) * (W / 2) * (C / 8);
  hipLaunchKernelGGL(wino_in_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, x_bs, x_rs, x_ps,
                     B, H, W, C, V);
  return ammc_launch_status();
}

extern "C" int wino_output_transform(const float* M, int B, int H, int W, int N, const float* scale, const float* shift,
                                     int relu, float* y, int64_t y_bs, int64_t y_rs, int64_t y_ps, void* stream) {
  const int64_t n = (int64_t)B * (H / 2) * (W / 2) * (N / 8);
  hipLaunchKernelGGL(wino_out_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, M, B, H, W, N,
                     scale, shift, relu, y, y_bs, y_rs, y_ps);
  return ammc_launch_status();
}
