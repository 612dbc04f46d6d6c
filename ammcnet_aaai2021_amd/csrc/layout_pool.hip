// HBM-bound helper kernels of the path: module-boundary layout changes, 2x2 max-pool,
// halo zeroing, weight pre-packing, eval-BN folding.  All are pure streaming kernels
// (16-B accesses per lane wherever the layout allows); their roofline is HBM.
#include "ammc_common.h"

namespace ammc_impl {

// ---- NCHW -> halo-padded NHWC -------------------------------------------------------
// One thread per (pixel, 4-channel group).  Lanes run along x, so the four NCHW reads
// are 256-B coalesced per wave and the NHWC write is one 16-B store per lane.
__global__ __launch_bounds__(256) void nchw_to_nhwc_halo_kernel(
    const float* __restrict__ x, int B, int C, int H, int W, float* __restrict__ y,
    int64_t y_bs, int64_t y_rs, int64_t y_ps, int Cp) {
  const int groups = Cp >> 2;
  const int64_t total = (int64_t)B * groups * H * W;
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= total) return;
  const int xw = (int)(gid % W);
  int64_t t = gid / W;
  const int yh = (int)(t % H);
  t /= H;
  const int g = (int)(t % groups);
  const int b = (int)(t / groups);
  f32x4 v;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = g * 4 + i;
    v[i] = c < C ? x[(((int64_t)b * C + c) * H + yh) * W + xw] : 0.f;
  }
  float* dst = y + (int64_t)b * y_bs + (int64_t)yh * y_rs + (int64_t)xw * y_ps + g * 4;
  *reinterpret_cast<f32x4*>(dst) = v;
}

// ---- NHWC (strided) -> NCHW ---------------------------------------------------------
// 32 pixels x 32 channels tile through LDS so both sides are coalesced.
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(
    const float* __restrict__ x, int64_t x_bs, int64_t x_rs, int64_t x_ps,
    int B, int C, int H, int W, float* __restrict__ y) {
  __shared__ float tile[32][33];
  const int HW = H * W;
  const int p0 = blockIdx.x * 32;        // pixel within image
  const int c0 = blockIdx.y * 32;
  const int b = blockIdx.z;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int p = p0 + ty + 8 * i;
    const int c = c0 + tx;
    float v = 0.f;
    if (p < HW && c < C) {
      const int yy = p / W, xx = p % W;
      v = x[(int64_t)b * x_bs + (int64_t)yy * x_rs + (int64_t)xx * x_ps + c];
    }
    tile[ty + 8 * i][tx] = v;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + 8 * i;
    const int p = p0 + tx;
    if (p < HW && c < C) y[((int64_t)b * C + c) * HW + p] = tile[tx][ty + 8 * i];
  }
}

// ---- halo zeroing -------------------------------------------------------------------
__global__ __launch_bounds__(256) void zero_halo_kernel(float* __restrict__ y, int B, int H, int W, int C4) {
  // border pixels per image: 2*(W+2) + 2*H
  const int nb = 2 * (W + 2) + 2 * H;
  const int64_t total = (int64_t)B * nb * C4;
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= total) return;
  const int c4 = (int)(gid % C4);
  int64_t t = gid / C4;
  const int e = (int)(t % nb);
  const int b = (int)(t / nb);
  int yy, xx;
  if (e < W + 2) { yy = 0; xx = e; }
  else if (e < 2 * (W + 2)) { yy = H + 1; xx = e - (W + 2); }
  else { const int r = e - 2 * (W + 2); yy = 1 + (r >> 1); xx = (r & 1) ? W + 1 : 0; }
  f32x4 z = {0.f, 0.f, 0.f, 0.f};
  *reinterpret_cast<f32x4*>(y + ((((int64_t)b * (H + 2) + yy) * (W + 2)) + xx) * (C4 * 4) + c4 * 4) = z;
}

// ---- MaxPool2d(2) on NHWC -----------------------------------------------------------
__global__ __launch_bounds__(256) void maxpool2x2_kernel(
    const float* __restrict__ x, int64_t x_bs, int64_t x_rs, int64_t x_ps,
    float* __restrict__ y, int64_t y_bs, int64_t y_rs, int64_t y_ps, int B, int h, int w, int C4) {
  const int64_t total = (int64_t)B * h * w * C4;
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= total) return;
  const int c4 = (int)(gid % C4);
  int64_t t = gid / C4;
  const int xx = (int)(t % w);
  t /= w;
  const int yy = (int)(t % h);
  const int b = (int)(t / h);
  const float* s = x + (int64_t)b * x_bs + (int64_t)(2 * yy) * x_rs + (int64_t)(2 * xx) * x_ps + c4 * 4;
  const f32x4 v00 = *reinterpret_cast<const f32x4*>(s);
  const f32x4 v01 = *reinterpret_cast<const f32x4*>(s + x_ps);
  const f32x4 v10 = *reinterpret_cast<const f32x4*>(s + x_rs);
  const f32x4 v11 = *reinterpret_cast<const f32x4*>(s + x_rs + x_ps);
  f32x4 o;
#pragma unroll
  for (int i = 0; i < 4; ++i) o[i] = fmaxf(fmaxf(v00[i], v01[i]), fmaxf(v10[i], v11[i]));
  *reinterpret_cast<f32x4*>(y + (int64_t)b * y_bs + (int64_t)yy * y_rs + (int64_t)xx * y_ps + c4 * 4) = o;
}

// ---- weight packing -----------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_conv_weight_kernel(
    const float* __restrict__ w, int cout, int cin, int ks2, int cin_p, int kpad, float* __restrict__ out) {
  const int64_t total = (int64_t)cout * kpad;
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= total) return;
  const int k = (int)(gid % kpad);
  const int n = (int)(gid / kpad);
  const int tap = k / cin_p, c = k % cin_p;
  out[gid] = (tap < ks2 && c < cin) ? w[((int64_t)n * cin + c) * ks2 + tap] : 0.f;
}

__global__ __launch_bounds__(256) void pack_convt_weight_kernel(
    const float* __restrict__ w, int cin, int co, float* __restrict__ out) {
  const int64_t total = (int64_t)4 * co * cin;
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= total) return;
  const int ci = (int)(gid % cin);
  const int row = (int)(gid / cin);      // (dy*2+dx)*co + c_out
  const int g = row / co, c_out = row % co;
  out[gid] = w[((int64_t)ci * co + c_out) * 4 + g];
}

__global__ __launch_bounds__(256) void bn_fold_kernel(
    const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ mean,
    const float* __restrict__ var, float eps, int c, float* __restrict__ scale, float* __restrict__ shift) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= c) return;
  const float s = gamma[i] / sqrtf(var[i] + eps);
  scale[i] = s;
  shift[i] = beta[i] - mean[i] * s;
}

__global__ __launch_bounds__(256) void pack_codebook_kernel(
    const float* __restrict__ e_dm, int d, int m, float* __restrict__ e_md, float* __restrict__ enorm) {
  // one thread per slot; reads are coalesced along m, writes are row-per-thread (tiny tensor)
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= m) return;
  float s = 0.f;
  for (int i = 0; i < d; ++i) {
    const float v = e_dm[(int64_t)i * m + j];
    e_md[(int64_t)j * d + i] = v;
    s += v * v;          // same left-to-right order as embed.pow(2).sum(0)
  }
  enorm[j] = s;
}

__global__ __launch_bounds__(256) void sum_partials_kernel(const float* __restrict__ p, int n, float inv, float* out) {
  __shared__ double red[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) s += (double)p[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = (float)(red[0] * (double)inv);
}

inline unsigned blocks_for(int64_t total) { return (unsigned)((total + 255) / 256); }

}  // namespace ammc_impl
using namespace ammc_impl;

extern "C" {

int ammc_nchw_to_nhwc_f32(const float* x, int32_t batch, int32_t c, int32_t h, int32_t w,
                          float* y, int64_t y_bs, int64_t y_rs, int64_t y_ps, int32_t cp, void* stream) {
  if (!x || !y || batch <= 0 || c <= 0 || h <= 0 || w <= 0 || cp < c || (cp & 3)) return AMMC_EINVAL;
  if (((uintptr_t)y & 15) || ((y_bs | y_rs | y_ps) & 3)) return AMMC_EINVAL;
  const int64_t total = (int64_t)batch * (cp >> 2) * h * w;
  hipLaunchKernelGGL(nchw_to_nhwc_halo_kernel, dim3(blocks_for(total)), dim3(256), 0,
                     (hipStream_t)stream, x, batch, c, h, w, y, y_bs, y_rs, y_ps, cp);
  return ammc_launch_status();
}

int ammc_nhwc_to_nchw_f32(const float* x, int64_t x_bs, int64_t x_rs, int64_t x_ps,
                          int32_t batch, int32_t c, int32_t h, int32_t w, float* y, void* stream) {
  if (!x || !y || batch <= 0 || c <= 0 || h <= 0 || w <= 0) return AMMC_EINVAL;
  dim3 grid((h * w + 31) / 32, (c + 31) / 32, batch);
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, grid, dim3(256), 0, (hipStream_t)stream,
                     x, x_bs, x_rs, x_ps, batch, c, h, w, y);
  return ammc_launch_status();
}

int ammc_zero_halo_f32(float* y, int32_t batch, int32_t h, int32_t w, int32_t c, void* stream) {
  if (!y || batch <= 0 || h <= 0 || w <= 0 || c <= 0 || (c & 3)) return AMMC_EINVAL;
  const int64_t total = (int64_t)batch * (2 * (w + 2) + 2 * h) * (c >> 2);
  hipLaunchKernelGGL(zero_halo_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream,
                     y, batch, h, w, c >> 2);
  return ammc_launch_status();
}

int ammc_maxpool2x2_f32(const float* x, int64_t x_bs, int64_t x_rs, int64_t x_ps,
                        float* y, int64_t y_bs, int64_t y_rs, int64_t y_ps,
                        int32_t batch, int32_t h, int32_t w, int32_t c, void* stream) {
  if (!x || !y || batch <= 0 || h <= 0 || w <= 0 || c <= 0 || (c & 3)) return AMMC_EINVAL;
  if ((x_bs | x_rs | x_ps | y_bs | y_rs | y_ps) & 3) return AMMC_EINVAL;
  const int64_t total = (int64_t)batch * h * w * (c >> 2);
  hipLaunchKernelGGL(maxpool2x2_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream,
                     x, x_bs, x_rs, x_ps, y, y_bs, y_rs, y_ps, batch, h, w, c >> 2);
  return ammc_launch_status();
}

int ammc_pack_conv_weight_f32(const float* w_oihw, int32_t cout, int32_t cin, int32_t ksize,
                              int32_t cin_p, float* out, void* stream) {
  if (!w_oihw || !out || cout <= 0 || cin <= 0 || cin_p < cin || (ksize != 1 && ksize != 3 && ksize != 4)) return AMMC_EINVAL;
  const int ks2 = ksize * ksize;
  const int kpad = ((ks2 * cin_p + 31) / 32) * 32;
  const int64_t total = (int64_t)cout * kpad;
  hipLaunchKernelGGL(pack_conv_weight_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream,
                     w_oihw, cout, cin, ks2, cin_p, kpad, out);
  return ammc_launch_status();
}

int ammc_pack_convt_weight_f32(const float* w_iohw, int32_t cin, int32_t co, float* out, void* stream) {
  if (!w_iohw || !out || cin <= 0 || co <= 0) return AMMC_EINVAL;
  const int64_t total = (int64_t)4 * co * cin;
  hipLaunchKernelGGL(pack_convt_weight_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream,
                     w_iohw, cin, co, out);
  return ammc_launch_status();
}

int ammc_bn_fold_f32(const float* gamma, const float* beta, const float* mean, const float* var,
                     float eps, int32_t c, float* scale, float* shift, void* stream) {
  if (!gamma || !beta || !mean || !var || !scale || !shift || c <= 0) return AMMC_EINVAL;
  hipLaunchKernelGGL(bn_fold_kernel, dim3(blocks_for(c)), dim3(256), 0, (hipStream_t)stream,
                     gamma, beta, mean, var, eps, c, scale, shift);
  return ammc_launch_status();
}

int ammc_pack_codebook_f32(const float* embed_dm, int32_t d, int32_t m, float* embed_md,
                           float* enorm, void* stream) {
  if (!embed_dm || !embed_md || !enorm || d <= 0 || m <= 0) return AMMC_EINVAL;
  hipLaunchKernelGGL(pack_codebook_kernel, dim3(blocks_for(m)), dim3(256), 0, (hipStream_t)stream,
                     embed_dm, d, m, embed_md, enorm);
  return ammc_launch_status();
}

int ammc_sum_partials_f32(const float* partial, int32_t nparts, float inv_count, float* out, void* stream) {
  if (!partial || !out || nparts <= 0) return AMMC_EINVAL;
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream,
                     partial, nparts, inv_count, out);
  return ammc_launch_status();
}

}  // extern "C"
