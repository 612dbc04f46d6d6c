// `outc` + tanh (unet.py:920, 998-1007): 3x3 convolution with 2 or 3 output channels,
// bias, tanh, NCHW store.  AI is 9-13 flop/B, so this layer is HBM-bound (SURVEY.md
// Appendix B): a direct VALU convolution, one output pixel per lane.  The filter taps are
// wave-uniform, so they are fetched with scalar loads and fed to v_fma as SGPR operands;
// the NHWC input is read 16 B per lane (every byte of a line is consumed by the c-loop, so
// the 3x3 re-reads are served by L1/L2) and the NCHW planes are written 256 B per wave.
#include "ammc_common.h"

namespace ammc_impl {

template <int COUT>
__global__ __launch_bounds__(256) void outc_tanh_kernel(
    const float* __restrict__ x, int64_t x_bs, int64_t x_rs, int64_t x_ps,
    const float* __restrict__ wp,   // [9][cin][4] (co padded to 4)
    const float* __restrict__ bias, int B, int H, int W, int cin, float* __restrict__ y) {
  const int64_t total = (int64_t)B * H * W;
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= total) return;
  const int xx = (int)(gid % W);
  const int64_t t = gid / W;
  const int yy = (int)(t % H);
  const int b = (int)(t / H);
  const float* px = x + (int64_t)b * x_bs + (int64_t)yy * x_rs + (int64_t)xx * x_ps;
  float acc[COUT];
#pragma unroll
  for (int o = 0; o < COUT; ++o) acc[o] = 0.f;
  for (int r = 0; r < 3; ++r) {
    for (int s = 0; s < 3; ++s) {
      const float* p = px + r * x_rs + s * x_ps;
      const float* wt = wp + (int64_t)(r * 3 + s) * cin * 4;
      for (int c = 0; c < cin; c += 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(p + c);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int o = 0; o < COUT; ++o) acc[o] = fmaf(v[i], wt[(c + i) * 4 + o], acc[o]);
      }
    }
  }
  const int64_t hw = (int64_t)H * W;
#pragma unroll
  for (int o = 0; o < COUT; ++o)
    y[((int64_t)b * COUT + o) * hw + (int64_t)yy * W + xx] = tanhf(acc[o] + bias[o]);
}

// OIHW [cout][cin][3][3] -> [9][cin][4]
__global__ __launch_bounds__(256) void pack_outc_kernel(const float* __restrict__ w, int cout, int cin,
                                                        float* __restrict__ out) {
  const int total = 9 * cin * 4;
  const int gid = blockIdx.x * 256 + threadIdx.x;
  if (gid >= total) return;
  const int o = gid & 3;
  const int c = (gid >> 2) % cin;
  const int tap = (gid >> 2) / cin;
  out[gid] = o < cout ? w[((int64_t)o * cin + c) * 9 + tap] : 0.f;
}

}  // namespace ammc_impl
using namespace ammc_impl;

extern "C" int ammc_pack_outc_weight_f32(const float* w_oihw, int32_t cout, int32_t cin, float* out, void* stream) {
  if (!w_oihw || !out || cout <= 0 || cout > 4 || cin <= 0) return AMMC_EINVAL;
  const int total = 9 * cin * 4;
  hipLaunchKernelGGL(pack_outc_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     w_oihw, cout, cin, out);
  return ammc_launch_status();
}

extern "C" int ammc_conv3x3_out_tanh_f32(const float* x, int64_t x_bs, int64_t x_rs, int64_t x_ps,
                                         const float* w_packed, const float* bias,
                                         int32_t batch, int32_t height, int32_t width,
                                         int32_t cin, int32_t cout, float* y_nchw, void* stream) {
  if (!x || !w_packed || !bias || !y_nchw) return AMMC_EINVAL;
  if (batch <= 0 || height <= 0 || width <= 0 || cin <= 0 || (cin & 3)) return AMMC_EINVAL;
  if ((x_bs | x_rs | x_ps) & 3) return AMMC_EINVAL;
  const int64_t total = (int64_t)batch * height * width;
  const dim3 grid((unsigned)((total + 255) / 256));
  hipStream_t s = (hipStream_t)stream;
  switch (cout) {
    case 1: hipLaunchKernelGGL(outc_tanh_kernel<1>, grid, dim3(256), 0, s, x, x_bs, x_rs, x_ps, w_packed, bias, batch, height, width, cin, y_nchw); break;
    case 2: hipLaunchKernelGGL(outc_tanh_kernel<2>, grid, dim3(256), 0, s, x, x_bs, x_rs, x_ps, w_packed, bias, batch, height, width, cin, y_nchw); break;
    case 3: hipLaunchKernelGGL(outc_tanh_kernel<3>, grid, dim3(256), 0, s, x, x_bs, x_rs, x_ps, w_packed, bias, batch, height, width, cin, y_nchw); break;
    case 4: hipLaunchKernelGGL(outc_tanh_kernel<4>, grid, dim3(256), 0, s, x, x_bs, x_rs, x_ps, w_packed, bias, batch, height, width, cin, y_nchw); break;
    default: return AMMC_EUNSUP;
  }
  return ammc_launch_status();
}
