// Device half of the input pipeline: what the reference's loaders do per frame on the CPU
// (Code/dataset/two_stream_dataset.py:72-99, 503-506) done once per frame on the GPU, from the raw decoded bytes.
//
//   frames: uint8 RGB (or BGR) [n][h][w][3] -> cv2.resize INTER_LINEAR (8-bit fixed-point form) -> ToTensor (/255)
//           -> Normalize(0.5, 0.5) -> float32 [n][3][oh][ow]
//   flows : float32 [n][h][w][2] (.flo payload) -> cv2.resize INTER_LINEAR (float form) -> channel 0 / oh,
//           channel 1 = (scaled channel 0) / ow (the reference's `_load_op`, :94-95) -> float32 [n][2][oh][ow]
//
// One thread per output pixel; HBM-bound streaming (uploading 1 byte per sample instead of 4, and each frame once
// instead of once per clip that contains it, is the point).  The arithmetic follows oracle/pipeline_oracle.py
// operation by operation (no FMA contraction: see `rounded`), so the results are bit-identical.
#include "ammc_common.h"

// hipcc contracts a * b + c into an FMA by default (its __fmul_rn / __fadd_rn are plain operators and a file-scope
// `#pragma clang fp contract(off)` did not stop it): every product that feeds a sum goes through `rounded()`, an
// empty asm the optimiser cannot look through, so products and sums are rounded separately, as the CPU loaders
// (and the oracle) compute them
__device__ __forceinline__ float rounded(float v) {
  asm volatile("" : "+v"(v));
  return v;
}

namespace ammc_impl {

struct Coord { int s; float f; };

__device__ __forceinline__ Coord src_coord(int d, double scale, int src) {
  float f = (float)(((double)d + 0.5) * scale - 0.5);
  int s = (int)floorf(f);
  f = __fsub_rn(f, (float)s);
  if (s < 0) { s = 0; f = 0.f; }
  if (s >= src - 1) { s = src - 1; f = 0.f; }
  Coord c; c.s = s; c.f = f;
  return c;
}

__global__ __launch_bounds__(256) void frames_u8_kernel(const uint8_t* __restrict__ src, int n, int h, int w,
                                                        float* __restrict__ dst, int oh, int ow, int bgr,
                                                        double sx, double sy) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (int64_t)n * oh * ow) return;
  const int dx = (int)(gid % ow);
  const int dy = (int)((gid / ow) % oh);
  const int f = (int)(gid / ((int64_t)ow * oh));
  const Coord cx = src_coord(dx, sx, w), cy = src_coord(dy, sy, h);
  const int ax1 = (int)rintf(__fmul_rn(cx.f, 2048.f)), ax0 = (int)rintf(__fmul_rn(__fsub_rn(1.f, cx.f), 2048.f));
  const int by1 = (int)rintf(__fmul_rn(cy.f, 2048.f)), by0 = (int)rintf(__fmul_rn(__fsub_rn(1.f, cy.f), 2048.f));
  const int x1 = min(cx.s + 1, w - 1), y1 = min(cy.s + 1, h - 1);
  const uint8_t* img = src + (int64_t)f * h * w * 3;
  const uint8_t* r0 = img + (int64_t)cy.s * w * 3;
  const uint8_t* r1 = img + (int64_t)y1 * w * 3;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int s0 = (int)r0[cx.s * 3 + c] * ax0 + (int)r0[x1 * 3 + c] * ax1;
    const int s1 = (int)r1[cx.s * 3 + c] * ax0 + (int)r1[x1 * 3 + c] * ax1;
    int v = (((by0 * (s0 >> 4)) >> 16) + ((by1 * (s1 >> 4)) >> 16) + 2) >> 2;
    v = v < 0 ? 0 : (v > 255 ? 255 : v);
    const float t = __fdiv_rn((float)v, 255.f);
    const float o = __fdiv_rn(__fsub_rn(t, 0.5f), 0.5f);
    const int co = bgr ? 2 - c : c;
    dst[(((int64_t)f * 3 + co) * oh + dy) * ow + dx] = o;
  }
}

__global__ __launch_bounds__(256) void flows_kernel(const float* __restrict__ src, int n, int h, int w,
                                                    float* __restrict__ dst, int oh, int ow, double sx, double sy) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (int64_t)n * oh * ow) return;
  const int dx = (int)(gid % ow);
  const int dy = (int)((gid / ow) % oh);
  const int f = (int)(gid / ((int64_t)ow * oh));
  const Coord cx = src_coord(dx, sx, w), cy = src_coord(dy, sy, h);
  const float a1 = cx.f, a0 = __fsub_rn(1.f, cx.f), b1 = cy.f, b0 = __fsub_rn(1.f, cy.f);
  const int x1 = min(cx.s + 1, w - 1), y1 = min(cy.s + 1, h - 1);
  const float* img = src + (int64_t)f * h * w * 2;
  const float* r0 = img + (int64_t)cy.s * w * 2;
  const float* r1 = img + (int64_t)y1 * w * 2;
  // only channel 0 reaches the output: channel 1 is re-derived from it (two_stream_dataset.py:94-95)
  const float s0 = rounded(r0[cx.s * 2] * a0) + rounded(r0[x1 * 2] * a1);
  const float s1 = rounded(r1[cx.s * 2] * a0) + rounded(r1[x1 * 2] * a1);
  const float u = rounded(rounded(s0) * b0) + rounded(rounded(s1) * b1);
  const float c0 = rounded(u) / (float)oh;            // `img * 1.0 / image_height` (the * 1.0 is exact)
  const float c1 = rounded(c0) / (float)ow;
  dst[(((int64_t)f * 2 + 0) * oh + dy) * ow + dx] = c0;
  dst[(((int64_t)f * 2 + 1) * oh + dy) * ow + dx] = c1;
}

}  // namespace ammc_impl
using namespace ammc_impl;

extern "C" int ammc_frames_u8_to_f32(const uint8_t* src, int32_t n, int32_t h, int32_t w, float* dst, int32_t oh,
                                     int32_t ow, int32_t bgr, void* stream) {
  if (!src || !dst || n <= 0 || h <= 0 || w <= 0 || oh <= 0 || ow <= 0) return AMMC_EINVAL;
  const int64_t total = (int64_t)n * oh * ow;
  hipLaunchKernelGGL(frames_u8_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, n,
                     h, w, dst, oh, ow, bgr ? 1 : 0, (double)w / (double)ow, (double)h / (double)oh);
  return ammc_launch_status();
}

extern "C" int ammc_flows_to_f32(const float* src, int32_t n, int32_t h, int32_t w, float* dst, int32_t oh, int32_t ow,
                                 void* stream) {
  if (!src || !dst || n <= 0 || h <= 0 || w <= 0 || oh <= 0 || ow <= 0) return AMMC_EINVAL;
  const int64_t total = (int64_t)n * oh * ow;
  hipLaunchKernelGGL(flows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, n, h,
                     w, dst, oh, ow, (double)w / (double)ow, (double)h / (double)oh);
  return ammc_launch_status();
}
