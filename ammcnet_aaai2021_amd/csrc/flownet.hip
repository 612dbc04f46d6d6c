// Element-wise pieces of the FlowNet2-SD forward (reference models/flownet2/models.py:15-59); its
// convolutions and transposed convolutions run on conv_gemm_f32.  All HBM-bound streaming kernels.
#include "ammc_common.h"

namespace ammc_impl {

// inputs [B][3][2][H][W] in 0..rgb_max  ->  NHWC activation with 8 channels: (x - mean over (frame, H, W)) / rgb_max,
// channel order frame-major (models.py:16-18); channels 6, 7 are zero.  Two kernels (round 4: one workgroup per (sample,
// colour) doing both halves took 306 us at batch 32, 96 workgroups on 256 CUs): PREP_SLICES workgroups per (sample,
// colour) sum their slice in double, then every workgroup of the second kernel adds the slices' sums in a fixed order
// (the mean is the same bits everywhere, run to run) and writes its share of the pixels.
constexpr int PREP_SLICES = 32;
__global__ __launch_bounds__(256) void flownet_prep_sum_kernel(const float* __restrict__ in, int H, int W, double* __restrict__ part) {
  __shared__ double red[256];
  const int bc = blockIdx.x / PREP_SLICES, sl = blockIdx.x % PREP_SLICES;
  const int64_t n = (int64_t)2 * H * W;
  const int64_t per = (n + PREP_SLICES - 1) / PREP_SLICES;
  const int64_t lo = sl * per, hi = lo + per < n ? lo + per : n;
  const float* src = in + (int64_t)bc * n;
  double s = 0.0;
  for (int64_t i = lo + threadIdx.x; i < hi; i += 256) s += (double)src[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256) void flownet_prep_kernel(const float* __restrict__ in, int H, int W,
                                                           float* __restrict__ y, int64_t y_bs, int64_t y_rs, int64_t y_ps,
                                                           float rgb_max, const double* __restrict__ part) {
  const int bc = blockIdx.x / PREP_SLICES, sl = blockIdx.x % PREP_SLICES;
  const int b = bc / 3, c = bc % 3;
  const int64_t n = (int64_t)2 * H * W;
  const float* src = in + (int64_t)bc * n;
  double tot = 0.0;
  for (int i = 0; i < PREP_SLICES; ++i) tot += part[bc * PREP_SLICES + i];
  const float mean = (float)(tot / (double)n);
  const int64_t per = (n + PREP_SLICES - 1) / PREP_SLICES;
  const int64_t lo = sl * per, hi = lo + per < n ? lo + per : n;
  for (int64_t i = lo + threadIdx.x; i < hi; i += 256) {
    const int f = (int)(i / ((int64_t)H * W));
    const int64_t r = i - (int64_t)f * H * W;
    const int yy = (int)(r / W), xx = (int)(r - (int64_t)yy * W);
    y[(int64_t)b * y_bs + (int64_t)yy * y_rs + (int64_t)xx * y_ps + 3 * f + c] = (src[i] - mean) / rgb_max;
  }
}

// y = y > 0 ? y : slope * y over the first c channels of an NHWC activation (the LeakyReLU of a transposed conv whose
// input is a concatenation: the partial sums of the parts are activated once they are complete)
__global__ __launch_bounds__(256) void lrelu_kernel(float* __restrict__ y, int64_t y_bs, int64_t y_rs, int64_t y_ps, int B,
                                                    int H, int W, int C4, float slope) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (int64_t)B * H * W * C4) return;
  const int c4 = (int)(gid % C4);
  int64_t t = gid / C4;
  const int x = (int)(t % W);
  t /= W;
  const int yy = (int)(t % H), b = (int)(t / H);
  float* p = y + b * y_bs + yy * y_rs + x * y_ps + c4 * 4;
  f32x4 v = *reinterpret_cast<f32x4*>(p);
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = v[i] > 0.f ? v[i] : v[i] * slope;
  *reinterpret_cast<f32x4*>(p) = v;
}

// the same on an S16 activation (groups of 8 channels: 16 B of hi halves, 16 B of lo halves)
typedef _Float16 fl_h8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256) void lrelu_s16_kernel(float* __restrict__ y, int64_t y_bs, int64_t y_rs, int64_t y_ps, int B,
                                                        int H, int W, int C8, float slope) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (int64_t)B * H * W * C8) return;
  const int c8 = (int)(gid % C8);
  int64_t t = gid / C8;
  const int x = (int)(t % W);
  t /= W;
  const int yy = (int)(t % H), b = (int)(t / H);
  float* p = y + b * y_bs + yy * y_rs + x * y_ps + c8 * 8;
  const fl_h8 hi = *reinterpret_cast<const fl_h8*>(p), lo = *reinterpret_cast<const fl_h8*>(p + 4);
  float v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float f = (float)hi[i] + (float)lo[i] * (1.f / 2048.f);
    v[i] = f > 0.f ? f : f * slope;
  }
  ammc_u4 h, l;
  ammc_s16_split8(v, h, l);
  *reinterpret_cast<ammc_u4*>(p) = h;
  *reinterpret_cast<ammc_u4*>(p + 4) = l;
}

// nn.Upsample(scale_factor=4, mode='bilinear') of (x * premul): NHWC channels [0, c) -> NCHW [B][c][4H][4W]
// (align_corners=False: src = (dst + 0.5) / 4 - 0.5 clamped at 0; models.py:59, FlowNetSD.py:58)
__global__ __launch_bounds__(256) void upsample4_kernel(const float* __restrict__ x, int64_t x_bs, int64_t x_rs, int64_t x_ps,
                                                        int B, int H, int W, int c, float premul, float* __restrict__ out) {
  const int OH = 4 * H, OW = 4 * W;
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (int64_t)B * c * OH * OW) return;
  const int ox = (int)(gid % OW);
  int64_t t = gid / OW;
  const int oy = (int)(t % OH);
  t /= OH;
  const int ch = (int)(t % c), b = (int)(t / c);
  float fy = ((float)oy + 0.5f) * 0.25f - 0.5f, fx = ((float)ox + 0.5f) * 0.25f - 0.5f;
  fy = fy < 0.f ? 0.f : fy;
  fx = fx < 0.f ? 0.f : fx;
  const int y0 = (int)fy, x0 = (int)fx;
  const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
  const float ly = fy - (float)y0, lx = fx - (float)x0;
  const float* p = x + b * x_bs + ch;
  const float v00 = p[y0 * x_rs + x0 * x_ps] * premul, v01 = p[y0 * x_rs + x1 * x_ps] * premul;
  const float v10 = p[y1 * x_rs + x0 * x_ps] * premul, v11 = p[y1 * x_rs + x1 * x_ps] * premul;
  out[gid] = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
}

}  // namespace ammc_impl
using namespace ammc_impl;

extern "C" int ammc_flownet_prep_scratch_doubles(int32_t batch) { return batch > 0 ? batch * 3 * PREP_SLICES : 0; }

extern "C" int ammc_flownet_prep_f32(const float* in, int32_t batch, int32_t h, int32_t w, float* y, int64_t y_bs,
                                     int64_t y_rs, int64_t y_ps, float rgb_max, double* scratch, void* stream) {
  if (!in || !y || batch <= 0 || h <= 0 || w <= 0 || !(rgb_max > 0.f)) return AMMC_EINVAL;
  if (!scratch || ((uintptr_t)scratch & 7)) return AMMC_EINVAL;
  double* part = scratch;
  const int need = batch * 3 * PREP_SLICES;
  hipLaunchKernelGGL(flownet_prep_sum_kernel, dim3(need), dim3(256), 0, (hipStream_t)stream, in, h, w, part);
  hipLaunchKernelGGL(flownet_prep_kernel, dim3(need), dim3(256), 0, (hipStream_t)stream, in, h, w, y, y_bs, y_rs, y_ps,
                     rgb_max, part);
  return ammc_launch_status();
}

extern "C" int ammc_lrelu_f32(float* y, int64_t y_bs, int64_t y_rs, int64_t y_ps, int32_t batch, int32_t h, int32_t w,
                              int32_t c, float slope, void* stream) {
  if (!y || batch <= 0 || h <= 0 || w <= 0 || c <= 0 || (c & 3) || ((uintptr_t)y & 15) || ((y_bs | y_rs | y_ps) & 3))
    return AMMC_EINVAL;
  const int64_t total = (int64_t)batch * h * w * (c >> 2);
  hipLaunchKernelGGL(lrelu_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, y, y_bs, y_rs,
                     y_ps, batch, h, w, c >> 2, slope);
  return ammc_launch_status();
}

extern "C" int ammc_lrelu_s16(float* y, int64_t y_bs, int64_t y_rs, int64_t y_ps, int32_t batch, int32_t h, int32_t w,
                              int32_t c, float slope, void* stream) {
  if (!y || batch <= 0 || h <= 0 || w <= 0 || c <= 0 || (c & 7) || ((uintptr_t)y & 31) || ((y_bs | y_rs | y_ps) & 7))
    return AMMC_EINVAL;
  const int64_t total = (int64_t)batch * h * w * (c >> 3);
  hipLaunchKernelGGL(lrelu_s16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, y, y_bs,
                     y_rs, y_ps, batch, h, w, c >> 3, slope);
  return ammc_launch_status();
}

extern "C" int ammc_upsample4_bilinear_f32(const float* x, int64_t x_bs, int64_t x_rs, int64_t x_ps, int32_t batch,
                                           int32_t h, int32_t w, int32_t c, float premul, float* out, void* stream) {
  if (!x || !out || batch <= 0 || h <= 0 || w <= 0 || c <= 0) return AMMC_EINVAL;
  const int64_t total = (int64_t)batch * c * 16 * h * w;
  hipLaunchKernelGGL(upsample4_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, x_bs,
                     x_rs, x_ps, batch, h, w, c, premul, out);
  return ammc_launch_status();
}
