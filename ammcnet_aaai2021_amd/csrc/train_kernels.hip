// Training-mode pieces of the path that are not contractions: batch-statistics BatchNorm
// (forward statistics / apply, backward reductions / apply), max-pool and tanh backward, the
// commit-loss gradient, the EMA codebook update, per-channel bias-gradient sums and the
// filter re-layouts for the input-gradient convolutions.  All are streaming kernels over
// NHWC tensors with explicit strides (16 B per lane); their roofline is HBM.
//
// Reference semantics (Code/models/unet.py): nn.BatchNorm2d in training mode (:12,15: biased
// variance to normalise, unbiased for the running estimate, momentum 0.1), nn.MaxPool2d(2)
// (:36), torch.tanh (:1007), `diff` (:310), EMA update (:298-309).
#include "ammc_common.h"
#include <type_traits>
#include <stdlib.h>

namespace ammc_impl {

struct Tensor3 {          // NHWC view: element offset of pixel (b, y, x), channel 0
  int64_t bs, rs, ps;
};

// MaxPool2d(2) backward on the fly: the gradient a BatchNorm-backward kernel reads is  add + unpool(dpo)  - `add` through the
// kernel's dy argument, dpo[b][y >> 1][x >> 1] added where the forward recorded window position (y & 1) * 2 + (x & 1)
// (idx: a byte per pooled element, dense [B][ph][pw][C], ammc_maxpool2x2_s16_idx / ammc_scale_shift_act_s16_pool_f32).
// idx == nullptr: dy as it stands.  Pixels of a last odd row / column are in no window.
struct Unpool {
  const float* dpo;
  Tensor3 pt;
  const unsigned char* idx;
  int ph, pw;
};

__device__ __forceinline__ int64_t pix_off(int m, int H, int W, const Tensor3& t) {
  const int x = m % W;
  const int q = m / W;
  const int y = q % H;
  const int b = q / H;
  return (int64_t)b * t.bs + (int64_t)y * t.rs + (int64_t)x * t.ps;
}

// Pixels per workgroup in the per-channel reductions: about 2048 workgroups whatever the level (a fixed 2048 pixels
// left the 32x32 level with 16 workgroups on a 256-CU chip and the average of these kernels at 2.2 TB/s).  A function
// of the pixel count alone, so the partial sums - and the results - are the same on every launch.
// (Round 4: ~1024 workgroups instead of ~2048 - four workgroups of 256 threads per CU keep 64-128 KB of loads in flight,
// as many as eight did, and the finalizers that combine the partial rows - 64 launches of 8-64 workgroups per training
// step, pure latency - walk half as many.)
__host__ __device__ inline int red_pix_for(int M) {
  int p = ((M + 1023) / 1024 + 63) & ~63;
  return p < 64 ? 64 : (p > 4096 ? 4096 : p);
}

// ---- per-channel reductions ---------------------------------------------------------
// MODE 0: sum(x), sum(x^2)                         (BN forward statistics)
// MODE 1: sum(g), sum(g * xhat), g = dy * [pre>0]  (BN backward), xhat = (c - mean) * invstd
// MODE 2: sum(x)                                   (bias gradients)
// MODE 3: MODE 1 + max|g|, max|xhat| per channel    (Q = 4: bounds max|dc| before dc exists, bn_bwd_finalize_kernel)
// MODE 4: MODE 2 + max|x| of the whole tensor into the 256 `amax_bits` slots (as ammc_absmax_bits_f32 leaves it): the
//         bias-gradient pass of a ConvTranspose also finds the power of two for the S16 re-encoding of its gradient
// Layout: thread = (channel group of 4, pixel lane); partial[block][q][C].
// Workgroup b sums every gridDim.x-th chunk of U * PY pixels (a function of the shape alone: the same partial sums on
// every launch); a thread walks U pixel streams whose coordinates AND element offsets advance by additions and two
// carries per trip, and a trip issues all its loads before its first sum.  (Round 4.  The first form recomputed
// b * bs + y * rs + x * ps per load - 36 quarter-rate integer multiplies per trip of four loads - and guarded each load
// by a branch, which made hipcc wait for the loads in pairs.  Removing both did not move the kernel: on the 537-MB
// tensors of the 256x256 level it reads at 4.2-4.6 TB/s because the convolution that has just written the tensor
// left ~256 MB of it dirty in the Infinity Cache, and those lines go out to HBM under this kernel's loads - read plus
// write-back is the 6.1 TB/s HBM streams at.  Walking the tensor back to front, or as eight runs side by side the
// way the convolution's XCDs wrote it, to hit those lines instead of evicting them: measured, no gain.)
template <int MODE, bool UP = false>
__global__ __launch_bounds__(256) void chan_reduce_kernel(
    const float* __restrict__ x, Tensor3 xt, const float* __restrict__ dy, Tensor3 dt,
    const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ gamma,
    const float* __restrict__ beta, int relu, int M, int H, int W, int C, float* __restrict__ partial,
    int* __restrict__ amax_bits = nullptr, Unpool up = Unpool{nullptr, {0, 0, 0}, nullptr, 0, 0}) {
  __shared__ f32x4 red[MODE == 3 ? 4 : 2][256];
  constexpr bool BWD = MODE == 1 || MODE == 3;
  constexpr int U = BWD ? 4 : 8;                     // 16-byte loads in flight per thread: 8 either way
  const int C4 = C >> 2;
  const int tx = threadIdx.x % C4;
  const int ty = threadIdx.x / C4;
  const int PY = 256 / C4;
  const int step = U * PY;
  const int stride = (int)gridDim.x * step;
  const int sx = stride % W, sy = (stride / W) % H, sb = stride / (W * H);
  const int64_t x_adv = (int64_t)sb * xt.bs + (int64_t)sy * xt.rs + (int64_t)sx * xt.ps;
  const int64_t x_row = xt.rs - (int64_t)W * xt.ps, x_img = xt.bs - (int64_t)H * xt.rs;      // carries
  const int64_t d_adv = BWD ? (int64_t)sb * dt.bs + (int64_t)sy * dt.rs + (int64_t)sx * dt.ps : 0;
  const int64_t d_row = BWD ? dt.rs - (int64_t)W * dt.ps : 0, d_img = BWD ? dt.bs - (int64_t)H * dt.rs : 0;
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f}, m0v = {0.f, 0.f, 0.f, 0.f}, m1v = {0.f, 0.f, 0.f, 0.f};
  f32x4 mu, is, ga, be;
  if (BWD && ty < PY) {
    mu = *reinterpret_cast<const f32x4*>(mean + tx * 4);
    is = *reinterpret_cast<const f32x4*>(invstd + tx * 4);
    ga = *reinterpret_cast<const f32x4*>(gamma + tx * 4);
    be = *reinterpret_cast<const f32x4*>(beta + tx * 4);
  }
  if (ty < PY) {
    int px[U], py[U], pb[U];
    int64_t ox[U], od[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int m = blockIdx.x * step + ty + u * PY;
      px[u] = m % W;
      const int q = m / W;
      py[u] = q % H;
      pb[u] = q / H;
      ox[u] = (int64_t)(q / H) * xt.bs + (int64_t)py[u] * xt.rs + (int64_t)px[u] * xt.ps + tx * 4;
      od[u] = BWD ? (int64_t)(q / H) * dt.bs + (int64_t)py[u] * dt.rs + (int64_t)px[u] * dt.ps + tx * 4 : 0;
    }
    // one trip: all loads first, then the sums; GUARD = the last, partial trip (loads of the pixels past the end are
    // redirected to stream 0's pixel, which is in range, and their values dropped)
    auto trip = [&](const int mb, auto guard) {
      constexpr bool GUARD = decltype(guard)::value;
      f32x4 v[U], g[U], gp[U];
      unsigned code[U];
      bool ok[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        ok[u] = !GUARD || mb + u * PY < M;
        v[u] = *reinterpret_cast<const f32x4*>(x + (ok[u] ? ox[u] : ox[0]));
        if (BWD) g[u] = *reinterpret_cast<const f32x4*>(dy + (ok[u] ? od[u] : od[0]));
        if (BWD && UP) {                              // the pooled gradient of this pixel's window and where its maximum was
          const int uu = ok[u] ? u : 0;
          const int wy = py[uu] >> 1, wx = px[uu] >> 1;
          const bool in = wy < up.ph && wx < up.pw;
          const int cy = in ? wy : 0, cx = in ? wx : 0;
          gp[u] = *reinterpret_cast<const f32x4*>(up.dpo + (int64_t)pb[uu] * up.pt.bs + (int64_t)cy * up.pt.rs + (int64_t)cx * up.pt.ps + tx * 4);
          code[u] = *reinterpret_cast<const unsigned*>(up.idx + (((int64_t)pb[uu] * up.ph + cy) * up.pw + cx) * C + tx * 4);
          if (!in) code[u] = 0xffffffffu;             // no window: no position matches
        }
      }
      if (BWD && UP) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const unsigned pos = (unsigned)((py[u] & 1) * 2 + (px[u] & 1));
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (((code[u] >> (8 * i)) & 0xffu) == pos) g[u][i] += gp[u][i];
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (!GUARD || ok[u]) {
          if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { s0[i] += v[u][i]; s1[i] += v[u][i] * v[u][i]; }
          } else if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 4; ++i) s0[i] += v[u][i];
          } else if (MODE == 4) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { s0[i] += v[u][i]; m0v[i] = fmaxf(m0v[i], fabsf(v[u][i])); }
          } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float xh = (v[u][i] - mu[i]) * is[i];
              const float pre = v[u][i] * ga[i] + be[i];      // ga = folded scale, be = folded shift: the forward's expression
              const float gi = (!relu || pre > 0.f) ? g[u][i] : 0.f;
              s0[i] += gi;
              s1[i] += gi * xh;
              if (MODE == 3) {
                m0v[i] = fmaxf(m0v[i], fabsf(gi));
                m1v[i] = fmaxf(m1v[i], fabsf(xh));
              }
            }
          }
        }
        px[u] += sx;                                         // next chunk of this stream: + stride, two carries
        py[u] += sy;
        ox[u] += x_adv;
        if (BWD) od[u] += d_adv;
        if (px[u] >= W) {
          px[u] -= W;
          ++py[u];
          ox[u] += x_row;
          if (BWD) od[u] += d_row;
        }
        pb[u] += sb;
        if (py[u] >= H) {
          py[u] -= H;
          ++pb[u];
          ox[u] += x_img;
          if (BWD) od[u] += d_img;
        }
      }
    };
    int mb = blockIdx.x * step + ty;
    for (; mb + (U - 1) * PY < M; mb += stride) trip(mb, std::false_type{});
    if (mb < M) trip(mb, std::true_type{});
  }
  red[0][threadIdx.x] = s0;
  red[1][threadIdx.x] = MODE == 4 ? m0v : s1;
  if (MODE == 3) {
    red[2][threadIdx.x] = m0v;
    red[3][threadIdx.x] = m1v;
  }
  __syncthreads();
  if (ty == 0) {
    for (int j = 1; j < PY; ++j) {
      const f32x4 a0 = red[0][j * C4 + tx], a1 = red[1][j * C4 + tx];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        s0[i] += a0[i];
        if (MODE == 4) m0v[i] = fmaxf(m0v[i], a1[i]);
        else s1[i] += a1[i];
      }
      if (MODE == 3) {
        const f32x4 b0 = red[2][j * C4 + tx], b1 = red[3][j * C4 + tx];
#pragma unroll
        for (int i = 0; i < 4; ++i) { m0v[i] = fmaxf(m0v[i], b0[i]); m1v[i] = fmaxf(m1v[i], b1[i]); }
      }
    }
    float* p = partial + (int64_t)blockIdx.x * ((MODE == 2 || MODE == 4) ? 1 : (MODE == 3 ? 4 : 2)) * C;
    *reinterpret_cast<f32x4*>(p + tx * 4) = s0;
    if (MODE != 2 && MODE != 4) *reinterpret_cast<f32x4*>(p + C + tx * 4) = s1;
    if (MODE == 4) {                       // (ty == 0: threads 0 .. C4 - 1; one atomic per thread is a few dozen per workgroup)
      const float m = fmaxf(fmaxf(m0v[0], m0v[1]), fmaxf(m0v[2], m0v[3]));
      if (amax_bits && m > 0.f && m < INFINITY) atomicMax(amax_bits + (blockIdx.x & 255), __float_as_int(m));
    }
    if (MODE == 3) {
      *reinterpret_cast<f32x4*>(p + 2 * C + tx * 4) = m0v;
      *reinterpret_cast<f32x4*>(p + 3 * C + tx * 4) = m1v;
    }
  }
}

// partial[nblk][Q][C] -> out[Q][C], double accumulation, fixed order (deterministic)
// A workgroup of 1024 threads combines 8 columns: 128 slices of the partial rows per column (thread = (slice, column)),
// then a fixed-order LDS tree - deterministic, and short dependent chains: with ~2048 partial rows per layer and only
// C / 8 workgroups, the rows have to be spread over many threads or these finalizers cost more than the reductions.
constexpr int RP_COLS = 8, RP_SLICES = 128, RP_THREADS = RP_COLS * RP_SLICES;

__device__ __forceinline__ void column_sums(const float* __restrict__ partial, int nblk, int64_t row_stride, int col,
                                            bool valid, int second_off, double (&red)[2][RP_SLICES][RP_COLS], double& s,
                                            double& ss) {
  const int cl = threadIdx.x % RP_COLS, sl = threadIdx.x / RP_COLS;
  double a = 0.0, b = 0.0;
  if (valid) {
    // eight rows (sixteen loads) per trip, loads first (one dependent load per row left this latency-bound); the sums
    // keep row order whatever U is
    constexpr int U = 8;
    for (int r0 = sl; r0 < nblk; r0 += U * RP_SLICES) {
      float va[U], vb[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int r = r0 + u * RP_SLICES;
        const bool ok = r < nblk;
        va[u] = ok ? partial[(int64_t)r * row_stride + col] : 0.f;
        vb[u] = ok && second_off ? partial[(int64_t)r * row_stride + second_off + col] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        a += (double)va[u];
        b += (double)vb[u];
      }
    }
  }
  red[0][sl][cl] = a;
  red[1][sl][cl] = b;
  __syncthreads();
  for (int o = RP_SLICES / 2; o > 0; o >>= 1) {
    if (sl < o) {
      red[0][sl][cl] += red[0][sl + o][cl];
      red[1][sl][cl] += red[1][sl + o][cl];
    }
    __syncthreads();
  }
  s = red[0][0][cl];
  ss = red[1][0][cl];
}

// seg_rows > 0: blockIdx.y = a run of seg_rows rows -> out[blockIdx.y][QC] (the first of two stages when a convolution's
// statistics epilogue leaves thousands of rows: with QC / 8 workgroups alone, 8192 rows took 58 us)
// max_from (a multiple of 8, QC when there is none): columns from there on are combined by max instead of a sum (the
// max |g|, max |xhat| rows of a BatchNorm-backward partial)
__global__ __launch_bounds__(RP_THREADS) void reduce_partials_kernel(const float* __restrict__ partial, int nblk, int QC,
                                                              float scale, float* __restrict__ out, int seg_rows, int max_from) {
  __shared__ double red[2][RP_SLICES][RP_COLS];
  if (seg_rows > 0) {
    const int r0 = blockIdx.y * seg_rows;
    partial += (int64_t)r0 * QC;
    out += (int64_t)blockIdx.y * QC;
    nblk = min(seg_rows, nblk - r0);
  }
  if ((int)blockIdx.x * RP_COLS >= max_from) {
    float* redm = reinterpret_cast<float*>(&red[0][0][0]);
    const int cl = threadIdx.x % RP_COLS, sl = threadIdx.x / RP_COLS, col = blockIdx.x * RP_COLS + cl;
    float m = 0.f;
    if (col < QC)
      for (int r = sl; r < nblk; r += RP_SLICES) m = fmaxf(m, partial[(int64_t)r * QC + col]);
    redm[sl * RP_COLS + cl] = m;
    __syncthreads();
    for (int o = RP_SLICES / 2; o > 0; o >>= 1) {
      if (sl < o) redm[sl * RP_COLS + cl] = fmaxf(redm[sl * RP_COLS + cl], redm[(sl + o) * RP_COLS + cl]);
      __syncthreads();
    }
    if (sl == 0 && col < QC) out[col] = redm[cl];
    return;
  }
  const int i = blockIdx.x * RP_COLS + threadIdx.x % RP_COLS;
  double s, ss;
  column_sums(partial, nblk, QC, i, i < QC, 0, red, s, ss);
  if (threadIdx.x < RP_COLS && i < QC) out[i] = (float)(s * (double)scale);
}

// BN training finalize: batch mean / biased var -> invstd, folded scale/shift; running stats.
__global__ __launch_bounds__(RP_THREADS) void bn_finalize_kernel(
    const float* __restrict__ partial, int nblk, int C, float count, const float* __restrict__ gamma,
    const float* __restrict__ beta, float eps, float momentum, float* __restrict__ running_mean,
    float* __restrict__ running_var, float* __restrict__ mean, float* __restrict__ invstd,
    float* __restrict__ scale, float* __restrict__ shift) {
  __shared__ double red[2][RP_SLICES][RP_COLS];
  const int c = blockIdx.x * RP_COLS + threadIdx.x % RP_COLS;
  double s, ss;
  column_sums(partial, nblk, (int64_t)2 * C, c, c < C, C, red, s, ss);
  if (threadIdx.x >= RP_COLS || c >= C) return;
  const double mu = s / count;
  double var = ss / count - mu * mu;
  var = var > 0.0 ? var : 0.0;
  const float is = (float)(1.0 / sqrt(var + (double)eps));
  mean[c] = (float)mu;
  invstd[c] = is;
  const float sc = gamma[c] * is;
  scale[c] = sc;
  shift[c] = beta[c] - (float)mu * sc;
  const double unbiased = count > 1.f ? var * (double)count / ((double)count - 1.0) : var;
  running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mu;
  running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
}

// y = act(x * scale + shift) + res        (interior pixels only: halos stay zero)
__global__ __launch_bounds__(256) void scale_shift_act_kernel(
    const float* __restrict__ x, Tensor3 xt, const float* __restrict__ scale, const float* __restrict__ shift,
    const float* __restrict__ res, Tensor3 rt, float* __restrict__ y, Tensor3 yt, int relu, int M, int H, int W,
    int C4) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (int64_t)M * C4) return;
  const int c4 = (int)(gid % C4);
  const int m = (int)(gid / C4);
  const f32x4 v = *reinterpret_cast<const f32x4*>(x + pix_off(m, H, W, xt) + c4 * 4);
  const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + c4 * 4);
  const f32x4 sh = *reinterpret_cast<const f32x4*>(shift + c4 * 4);
  f32x4 o;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float t = v[i] * sc[i] + sh[i];
    if (relu) t = t > 0.f ? t : 0.f;
    o[i] = t;
  }
  if (res) {
    const f32x4 r = *reinterpret_cast<const f32x4*>(res + pix_off(m, H, W, rt) + c4 * 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] += r[i];
  }
  *reinterpret_cast<f32x4*>(y + pix_off(m, H, W, yt) + c4 * 4) = o;
}

// The same with the S16 image of y as output (fp32 y optional): one thread per (pixel, 8 channels).  A tensor that only
// split-fp16 convolutions read - the middle activation of a double_conv - never exists in fp32.
__global__ __launch_bounds__(256) void scale_shift_act_s16_kernel(
    const float* __restrict__ x, Tensor3 xt, const float* __restrict__ scale, const float* __restrict__ shift,
    const float* __restrict__ res, Tensor3 rt, float* __restrict__ y32, float* __restrict__ y16, Tensor3 yt, int relu,
    int M, int H, int W, int C8) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (int64_t)M * C8) return;
  const int c8 = (int)(gid % C8);
  const int m = (int)(gid / C8);
  const float* xp = x + pix_off(m, H, W, xt) + c8 * 8;
  float o[8];
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(xp + half * 4);
    const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + c8 * 8 + half * 4);
    const f32x4 sh = *reinterpret_cast<const f32x4*>(shift + c8 * 8 + half * 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float t = v[i] * sc[i] + sh[i];
      if (relu) t = t > 0.f ? t : 0.f;
      o[half * 4 + i] = t;
    }
  }
  if (res) {
    const float* rp = res + pix_off(m, H, W, rt) + c8 * 8;
    const f32x4 r0 = *reinterpret_cast<const f32x4*>(rp), r1 = *reinterpret_cast<const f32x4*>(rp + 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) { o[i] += r0[i]; o[4 + i] += r1[i]; }
  }
  const int64_t oo = pix_off(m, H, W, yt) + c8 * 8;
  if (y32) {
    *reinterpret_cast<f32x4*>(y32 + oo) = f32x4{o[0], o[1], o[2], o[3]};
    *reinterpret_cast<f32x4*>(y32 + oo + 4) = f32x4{o[4], o[5], o[6], o[7]};
  }
  typedef _Float16 ss_f16x8 __attribute__((ext_vector_type(8)));
  ss_f16x8 hi, lo;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const _Float16 hv = (_Float16)o[i];
    hi[i] = hv;
    lo[i] = (_Float16)((o[i] - (float)hv) * 2048.f);
  }
  *reinterpret_cast<ss_f16x8*>(y16 + oo) = hi;
  *reinterpret_cast<ss_f16x8*>(y16 + oo + 4) = lo;
}

// dc = gamma * invstd * (g - sum_g / M - xhat * sum_gx / M),  g = dy * [pre > 0]
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(
    const float* __restrict__ c, Tensor3 ct, const float* __restrict__ dy, Tensor3 dt,
    const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ gamma,
    const float* __restrict__ beta, const float* __restrict__ sums /* [2][C]: dbeta, dgamma */, float inv_count,
    int relu, float* __restrict__ dc, Tensor3 ot, int M, int H, int W, int C4, int* __restrict__ amax_bits) {
  int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool live = gid < (int64_t)M * C4;
  if (!live) {
    if (!amax_bits) return;
    gid = (int64_t)M * C4 - 1;               // (the block-wide max below needs every thread at the barrier)
  }
  const int c4 = (int)(gid % C4);
  const int m = (int)(gid / C4);
  const int C = C4 * 4;
  const f32x4 v = *reinterpret_cast<const f32x4*>(c + pix_off(m, H, W, ct) + c4 * 4);
  const f32x4 g = *reinterpret_cast<const f32x4*>(dy + pix_off(m, H, W, dt) + c4 * 4);
  const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c4 * 4);
  const f32x4 is = *reinterpret_cast<const f32x4*>(invstd + c4 * 4);
  const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c4 * 4);
  const f32x4 be = *reinterpret_cast<const f32x4*>(beta + c4 * 4);
  const f32x4 sg = *reinterpret_cast<const f32x4*>(sums + c4 * 4);
  const f32x4 sgx = *reinterpret_cast<const f32x4*>(sums + C + c4 * 4);
  f32x4 o;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float xh = (v[i] - mu[i]) * is[i];
    const float pre = v[i] * ga[i] + be[i];             // ga = folded scale (gamma*invstd), be = folded shift
    const float gi = (!relu || pre > 0.f) ? g[i] : 0.f;
    o[i] = ga[i] * (gi - sg[i] * inv_count - xh * sgx[i] * inv_count);
  }
  if (live) *reinterpret_cast<f32x4*>(dc + pix_off(m, H, W, ot) + c4 * 4) = o;
  if (amax_bits) {       // largest |dc| of the tensor, for the S16 re-encoding of this gradient (ammc_absmax_bits_f32)
    __shared__ float wmax[4];
    float mx = fmaxf(fmaxf(fabsf(o[0]), fabsf(o[1])), fmaxf(fabsf(o[2]), fabsf(o[3])));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {                                   // one atomic per workgroup, 256 slots
      mx = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
      if (mx > 0.f && mx < INFINITY) atomicMax(amax_bits + (blockIdx.x & 255), __float_as_int(mx));
    }
  }
}

// BN backward finalize for the fused S16 path: partial[nblk][4][C] of chan_reduce_kernel<3> -> sums[2][C] (dbeta, dgamma)
// and an upper bound of max |dc| into the 256 amax slots (as ammc_absmax_bits_f32 leaves the exact one):
//   |dc| = |scale| |g - sum_g / M - xhat sum_gx / M| <= |scale| (max|g| + |sum_g| / M + max|xhat| |sum_gx| / M)
// The S16 encoding only needs a power of two that brings the tensor into the half range; a bound that is a few times
// too large moves every value down a bit or two of an exponent range with 2^-24 to spare - and it is known BEFORE
// dc is written, so bn_bwd_apply_s16_kernel stores the S16 twin directly (no fp32 dc, no re-encoding pass).
__global__ __launch_bounds__(RP_THREADS) void bn_bwd_finalize_kernel(const float* __restrict__ partial, int nblk, int C,
                                                                     float inv_count, const float* __restrict__ scale,
                                                                     float* __restrict__ sums, int* __restrict__ amax_bits) {
  __shared__ double red[2][RP_SLICES][RP_COLS];
  __shared__ float redm[2][RP_SLICES][RP_COLS];
  const int cl = threadIdx.x % RP_COLS, sl = threadIdx.x / RP_COLS;
  const int c = blockIdx.x * RP_COLS + cl;
  double s, ss;
  column_sums(partial, nblk, (int64_t)4 * C, c, c < C, C, red, s, ss);
  float mg = 0.f, mx = 0.f;
  if (c < C) {
    constexpr int U = 8;                              // loads first, as column_sums
    for (int r0 = sl; r0 < nblk; r0 += U * RP_SLICES) {
      float vg[U], vx[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int r = r0 + u * RP_SLICES;
        const bool ok = r < nblk;
        vg[u] = ok ? partial[(int64_t)r * 4 * C + 2 * C + c] : 0.f;
        vx[u] = ok ? partial[(int64_t)r * 4 * C + 3 * C + c] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) mg = fmaxf(mg, vg[u]), mx = fmaxf(mx, vx[u]);
    }
  }
  redm[0][sl][cl] = mg;
  redm[1][sl][cl] = mx;
  __syncthreads();
  for (int o = RP_SLICES / 2; o > 0; o >>= 1) {
    if (sl < o) {
      redm[0][sl][cl] = fmaxf(redm[0][sl][cl], redm[0][sl + o][cl]);
      redm[1][sl][cl] = fmaxf(redm[1][sl][cl], redm[1][sl + o][cl]);
    }
    __syncthreads();
  }
  if (sl != 0 || c >= C) return;
  const float sg = (float)s, sgx = (float)ss;
  sums[c] = sg;
  sums[C + c] = sgx;
  const float bound = fabsf(scale[c]) * (redm[0][0][cl] + fabsf(sg) * inv_count + redm[1][0][cl] * fabsf(sgx) * inv_count);
  if (bound > 0.f && bound < INFINITY) atomicMax(amax_bits + (blockIdx.x & 255), __float_as_int(bound));
}

__device__ __forceinline__ float tk_pow2_to_1024(int amax_bits) {      // as pow2_to_1024 of conv_gemm_s16.hip
  int e = ((amax_bits >> 23) & 255) - 127;
  if (amax_bits == 0) e = 10;
  int fe = 10 - e;
  fe = fe < -60 ? -60 : (fe > 60 ? 60 : fe);
  return __int_as_float((fe + 127) << 23);
}

// fp32 NHWC channel slice -> its S16 twin, scaled by the power of two that brings max |x| (the 256 slots of MODE 4 /
// ammc_absmax_bits_f32) to 2^10; 2^-k goes to inv_scale[0..n) for the consumers' epilogues.  The strided form of
// ammc_split_rows_scaled_f32: the gradient of a ConvTranspose's output is one half of a concat buffer.
__global__ __launch_bounds__(256) void split_scaled_strided_kernel(const float* __restrict__ x, Tensor3 xt, float* __restrict__ y16,
                                                                   Tensor3 yt, int M, int H, int W, int C8,
                                                                   const int* __restrict__ amax_bits, float* __restrict__ inv_scale, int n) {
  __shared__ int red[256];
  red[threadIdx.x] = amax_bits[threadIdx.x];
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] = max(red[threadIdx.x], red[threadIdx.x + o]);
    __syncthreads();
  }
  const float f = tk_pow2_to_1024(red[0]);
  if (blockIdx.x == 0)
    for (int i = threadIdx.x; i < n; i += 256) inv_scale[i] = 1.f / f;        // a power of two: exact
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (int64_t)M * C8) return;
  const int c8 = (int)(gid % C8);
  const int m = (int)(gid / C8);
  const float* xp = x + pix_off(m, H, W, xt) + c8 * 8;
  const f32x4 a0 = *reinterpret_cast<const f32x4*>(xp), a1 = *reinterpret_cast<const f32x4*>(xp + 4);
  float v[8];
#pragma unroll
  for (int i = 0; i < 4; ++i) { v[i] = a0[i] * f; v[4 + i] = a1[i] * f; }
  ammc_u4 hi, lo;
  ammc_s16_split8(v, hi, lo);
  float* yp = y16 + pix_off(m, H, W, yt) + c8 * 8;
  *reinterpret_cast<ammc_u4*>(yp) = hi;
  *reinterpret_cast<ammc_u4*>(yp + 4) = lo;
}

// bn_bwd_apply with the S16 twin as output: one thread per (pixel, group of 8 channels); dc * f with the power of
// two f from the amax slots (inverse into inv_scale[0..n) for the consumers' epilogues), optionally dc in fp32 too
__global__ __launch_bounds__(256) void bn_bwd_apply_s16_kernel(
    const float* __restrict__ c, Tensor3 ct, const float* __restrict__ dy, Tensor3 dt,
    const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ gamma,
    const float* __restrict__ beta, const float* __restrict__ sums, float inv_count, int relu,
    float* __restrict__ dc16, float* __restrict__ dc32, Tensor3 ot, int M, int H, int W, int C8,
    const int* __restrict__ amax_bits, float* __restrict__ inv_scale, int n_inv) {
  __shared__ int red[256];
  red[threadIdx.x] = amax_bits[threadIdx.x];
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] = max(red[threadIdx.x], red[threadIdx.x + o]);
    __syncthreads();
  }
  const float f = tk_pow2_to_1024(red[0]);
  if (blockIdx.x == 0)
    for (int i = threadIdx.x; i < n_inv; i += 256) inv_scale[i] = 1.f / f;
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (int64_t)M * C8) return;
  const int c8 = (int)(gid % C8);
  const int m = (int)(gid / C8);
  const int C = C8 * 8;
  const float* cp = c + pix_off(m, H, W, ct) + c8 * 8;
  const float* gp = dy + pix_off(m, H, W, dt) + c8 * 8;
  float o[8];
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const int cc = c8 * 8 + half * 4;
    const f32x4 v = *reinterpret_cast<const f32x4*>(cp + half * 4);
    const f32x4 g = *reinterpret_cast<const f32x4*>(gp + half * 4);
    const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + cc);
    const f32x4 is = *reinterpret_cast<const f32x4*>(invstd + cc);
    const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + cc);
    const f32x4 be = *reinterpret_cast<const f32x4*>(beta + cc);
    const f32x4 sg = *reinterpret_cast<const f32x4*>(sums + cc);
    const f32x4 sgx = *reinterpret_cast<const f32x4*>(sums + C + cc);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float xh = (v[i] - mu[i]) * is[i];
      const float pre = v[i] * ga[i] + be[i];
      const float gi = (!relu || pre > 0.f) ? g[i] : 0.f;
      o[half * 4 + i] = ga[i] * (gi - sg[i] * inv_count - xh * sgx[i] * inv_count);
    }
  }
  const int64_t oo = pix_off(m, H, W, ot) + c8 * 8;
  if (dc32) {
    *reinterpret_cast<f32x4*>(dc32 + oo) = f32x4{o[0], o[1], o[2], o[3]};
    *reinterpret_cast<f32x4*>(dc32 + oo + 4) = f32x4{o[4], o[5], o[6], o[7]};
  }
  typedef _Float16 tk_f16x8 __attribute__((ext_vector_type(8)));
  tk_f16x8 hi, lo;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float v = o[i] * f;
    const _Float16 hv = (_Float16)v;
    hi[i] = hv;
    lo[i] = (_Float16)((v - (float)hv) * 2048.f);
  }
  *reinterpret_cast<tk_f16x8*>(dc16 + oo) = hi;
  *reinterpret_cast<tk_f16x8*>(dc16 + oo + 4) = lo;
}

// MaxPool2d(2) backward (+ the gradient that reaches the same tensor through the skip path):
// dx[2y+i][2x+j] = add[2y+i][2x+j] + (first max position in row-major window order ? dp[y][x] : 0)
// x / add / dx are fh x fw (fh = 2h or 2h+1): the last row / column of an odd size belongs to no window (MaxPool2d floors)
// and receives only `add` (or zero); the grid covers ceil(fh/2) x ceil(fw/2) window positions.
// XS16: x is the S16 twin of the pooled tensor (what the forward's ammc_maxpool2x2_s16 compared: v = hi + lo 2^-11) -
// the fp32 tensor then has no reader left in the step and is not written at all.
template <bool XS16>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(
    const float* __restrict__ x, Tensor3 xt, const float* __restrict__ dp, Tensor3 pt,
    const float* __restrict__ add, Tensor3 at, float* __restrict__ dx, Tensor3 ot, int M, int h, int w, int fh, int fw,
    int C4) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (int64_t)M * C4) return;
  const int hh = (fh + 1) >> 1, ww = (fw + 1) >> 1;
  const int c4 = (int)(gid % C4);
  const int m = (int)(gid / C4);
  const int xx = m % ww, q = m / ww, yy = q % hh, b = q / hh;
  f32x4 o[4];
  if (yy < h && xx < w) {
    const float* s = x + (int64_t)b * xt.bs + (int64_t)(2 * yy) * xt.rs + (int64_t)(2 * xx) * xt.ps + c4 * 4;
    const f32x4 g = *reinterpret_cast<const f32x4*>(dp + (int64_t)b * pt.bs + (int64_t)yy * pt.rs + (int64_t)xx * pt.ps + c4 * 4);
    f32x4 v[4];
    if (XS16) {
      // channels 4 c4 .. + 3 of S16 group c4 >> 1: four hi halves at byte 8 (c4 & 1) of the group, their lo halves 16 bytes on
      typedef _Float16 mp_h4 __attribute__((ext_vector_type(4)));
      const float* sg = x + (int64_t)b * xt.bs + (int64_t)(2 * yy) * xt.rs + (int64_t)(2 * xx) * xt.ps + (c4 >> 1) * 8 + (c4 & 1) * 2;
      const int64_t offs[4] = {0, xt.ps, xt.rs, xt.rs + xt.ps};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const mp_h4 hv = *reinterpret_cast<const mp_h4*>(sg + offs[j]), lv = *reinterpret_cast<const mp_h4*>(sg + offs[j] + 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[j][i] = (float)hv[i] + (float)lv[i] * (1.f / 2048.f);
      }
    } else {
      v[0] = *reinterpret_cast<const f32x4*>(s);
      v[1] = *reinterpret_cast<const f32x4*>(s + xt.ps);
      v[2] = *reinterpret_cast<const f32x4*>(s + xt.rs);
      v[3] = *reinterpret_cast<const f32x4*>(s + xt.rs + xt.ps);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int arg = 0;
      float best = v[0][i];
#pragma unroll
      for (int j = 1; j < 4; ++j)
        if (v[j][i] > best) { best = v[j][i]; arg = j; }
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j][i] = (j == arg) ? g[i] : 0.f;
    }
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int64_t po = (int64_t)(j >> 1);
    const int64_t qo = (int64_t)(j & 1);
    if (2 * yy + po >= fh || 2 * xx + qo >= fw) continue;
    if (add) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(add + (int64_t)b * at.bs + (2 * yy + po) * at.rs + (2 * xx + qo) * at.ps + c4 * 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) o[j][i] += a[i];
    }
    *reinterpret_cast<f32x4*>(dx + (int64_t)b * ot.bs + (2 * yy + po) * ot.rs + (2 * xx + qo) * ot.ps + c4 * 4) = o[j];
  }
}

// The same from the window positions the forward recorded (ammc_maxpool2x2_s16_idx: a byte per pooled element, dense
// [B][h][w][C]) instead of the pooled tensor itself: 1 byte read per 16 that maxpool_bwd_kernel<true> reads to find the
// maxima again.  Thread = (window, 8 channels).
__global__ __launch_bounds__(256) void maxpool_bwd_idx_kernel(
    const unsigned char* __restrict__ idx, const float* __restrict__ dp, Tensor3 pt, const float* __restrict__ add,
    Tensor3 at, float* __restrict__ dx, Tensor3 ot, int M, int h, int w, int fh, int fw, int C8) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (int64_t)M * C8) return;
  const int hh = (fh + 1) >> 1, ww = (fw + 1) >> 1;
  const int c8 = (int)(gid % C8);
  const int m = (int)(gid / C8);
  const int xx = m % ww, q = m / ww, yy = q % hh, b = q / hh;
  const bool win = yy < h && xx < w;
  unsigned long long args = 0;
  f32x4 g0 = {0.f, 0.f, 0.f, 0.f}, g1 = {0.f, 0.f, 0.f, 0.f};
  if (win) {
    args = *reinterpret_cast<const unsigned long long*>(idx + ((((int64_t)b * h + yy) * w + xx) * C8 + c8) * 8);
    const float* gp = dp + (int64_t)b * pt.bs + (int64_t)yy * pt.rs + (int64_t)xx * pt.ps + c8 * 8;
    g0 = *reinterpret_cast<const f32x4*>(gp);
    g1 = *reinterpret_cast<const f32x4*>(gp + 4);
  }
  f32x4 a[4][2];
  bool in[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    in[j] = 2 * yy + (j >> 1) < fh && 2 * xx + (j & 1) < fw;
    a[j][0] = a[j][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (in[j] && add) {
      const float* ap = add + (int64_t)b * at.bs + (int64_t)(2 * yy + (j >> 1)) * at.rs + (int64_t)(2 * xx + (j & 1)) * at.ps + c8 * 8;
      a[j][0] = *reinterpret_cast<const f32x4*>(ap);
      a[j][1] = *reinterpret_cast<const f32x4*>(ap + 4);
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (!in[j]) continue;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (win && (unsigned)((args >> (8 * i)) & 0xff) == (unsigned)j) a[j][0][i] += g0[i];
      if (win && (unsigned)((args >> (8 * (4 + i))) & 0xff) == (unsigned)j) a[j][1][i] += g1[i];
    }
    float* op = dx + (int64_t)b * ot.bs + (int64_t)(2 * yy + (j >> 1)) * ot.rs + (int64_t)(2 * xx + (j & 1)) * ot.ps + c8 * 8;
    *reinterpret_cast<f32x4*>(op) = a[j][0];
    *reinterpret_cast<f32x4*>(op + 4) = a[j][1];
  }
}

// d(pre-tanh) = dout * (1 - out^2), NCHW -> NHWC (channels >= C written as zero up to Cp)
__global__ __launch_bounds__(256) void tanh_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ out,
                                                       int B, int C, int H, int W, float* __restrict__ y, Tensor3 yt,
                                                       int Cp4) {
  const int64_t total = (int64_t)B * Cp4 * H * W;
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= total) return;
  const int xw = (int)(gid % W);
  int64_t t = gid / W;
  const int yh = (int)(t % H);
  t /= H;
  const int g = (int)(t % Cp4);
  const int b = (int)(t / Cp4);
  f32x4 v;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = g * 4 + i;
    float r = 0.f;
    if (c < C) {
      const int64_t o = (((int64_t)b * C + c) * H + yh) * W + xw;
      const float ov = out[o];
      r = dout[o] * (1.f - ov * ov);
    }
    v[i] = r;
  }
  *reinterpret_cast<f32x4*>(y + (int64_t)b * yt.bs + (int64_t)yh * yt.rs + (int64_t)xw * yt.ps + g * 4) = v;
}

// dz = ddiff * 2 (z - E[idx0]) / (N*D)  (+ dq)       (autograd of unet.py:310-311)
__global__ __launch_bounds__(256) void commit_bwd_kernel(const float* __restrict__ z, const float* __restrict__ e_md,
                                                         const int* __restrict__ idx, int k,
                                                         const float* __restrict__ ddiff, float coef,
                                                         const float* __restrict__ dq, float* __restrict__ dz,
                                                         int N, int D4) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (int64_t)N * D4) return;
  const int d4 = (int)(gid % D4);
  const int n = (int)(gid / D4);
  const int s = idx[(int64_t)n * k];
  const f32x4 zv = *reinterpret_cast<const f32x4*>(z + (int64_t)n * D4 * 4 + d4 * 4);
  const f32x4 ev = *reinterpret_cast<const f32x4*>(e_md + (int64_t)s * D4 * 4 + d4 * 4);
  const float g = (ddiff ? ddiff[0] : 0.f) * coef;
  f32x4 o;
#pragma unroll
  for (int i = 0; i < 4; ++i) o[i] = g * (zv[i] - ev[i]);
  if (dq) {
    const f32x4 q = *reinterpret_cast<const f32x4*>(dq + (int64_t)n * D4 * 4 + d4 * 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] += q[i];
  }
  *reinterpret_cast<f32x4*>(dz + (int64_t)n * D4 * 4 + d4 * 4) = o;
}

// EMA codebook update, step 1: one workgroup per slot.  Each of the EMA_NW waves scans its own share of the rows
// (ballot of `nearest slot == mine`), adds the hit rows feature-parallel in row order, and the partial sums are
// combined in wave order: deterministic (no atomics) and free of workgroup barriers inside the scan (the first form
// synchronised three times per 256 rows and took 1.8 ms per launch).  Round 4: 16 waves per slot instead of 4 - the scan
// of 32768 row indices by four waves was 240 us per launch with 256 workgroups of 256 threads on 256 CUs.
constexpr int EMA_NW = 16;
__global__ __launch_bounds__(64 * EMA_NW) void ema_accumulate_kernel(const float* __restrict__ x, const int* __restrict__ idx,
                                                             int k, int N, int D, int M, float decay, float omd,
                                                             float* __restrict__ cluster_size,
                                                             float* __restrict__ embed_avg /* [D][M] */, int raw) {
  const int slot = blockIdx.x;
  __shared__ float wsum[EMA_NW][256];
  __shared__ int wcount[EMA_NW];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int chunk = ((N + EMA_NW - 1) / EMA_NW + 63) / 64 * 64;
  const int n_lo = wave * chunk, n_hi = min(n_lo + chunk, N);
  float acc[4] = {0.f, 0.f, 0.f, 0.f};                   // features lane, lane + 64, ... (D <= 256)
  int cnt = 0;
  // Both loads of this loop used to be exposed: one dependent index load per 64 rows and one dependent row load per
  // hit (~0.5 ms per launch at 32768 rows, a few hundred serial ~1.5-us latencies per wave).  Now four index chunks are
  // in flight per trip and the hit rows are fetched four at a time; they are still ADDED one by one in row order, so the
  // sums are the same bits as before.
  for (int base = n_lo; base < n_hi; base += 256) {
    int id[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int n = base + 64 * u + lane;
      id[u] = n < n_hi ? idx[(int64_t)n * k] : -1;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      unsigned long long bal = __ballot(id[u] == slot);
      cnt += __popcll(bal);
      while (bal) {                                      // wave-uniform: rows in increasing order, four per trip
        int b[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          b[t] = bal ? __ffsll((long long)bal) - 1 : -1;
          bal &= bal - 1;                                // (0 & anything stays 0)
        }
        float v[4][4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const float* row = x + (int64_t)(base + 64 * u + (b[t] < 0 ? 0 : b[t])) * D;
#pragma unroll
          for (int j = 0; j < 4; ++j) v[t][j] = (b[t] >= 0 && lane + 64 * j < D) ? row[lane + 64 * j] : 0.f;
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
          if (b[t] >= 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] += v[t][j];
          }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) wsum[wave][lane + 64 * j] = acc[j];
  if (lane == 0) wcount[wave] = cnt;
  __syncthreads();
  int count = 0;
  float sum = 0.f;
#pragma unroll
  for (int wv = 0; wv < EMA_NW; ++wv) {                   // fixed order: the same bits run to run
    count += wcount[wv];
    if (threadIdx.x < D) sum += wsum[wv][threadIdx.x];
  }
  if (raw) {            // counts[M] / sums[D][M] only: the EMA is applied after a cross-rank all-reduce
    if (threadIdx.x == 0) cluster_size[slot] = (float)count;
    if (threadIdx.x < D) embed_avg[(int64_t)threadIdx.x * M + slot] = sum;
    return;
  }
  if (threadIdx.x == 0) cluster_size[slot] = decay * cluster_size[slot] + omd * (float)count;
  for (int d = threadIdx.x; d < D; d += 256) {
    // note: with D > 256 the loop above accumulates several features into one `sum`; guarded on the host (D <= 256)
    embed_avg[(int64_t)d * M + slot] = decay * embed_avg[(int64_t)d * M + slot] + omd * sum;
  }
}

// EMA from (all-reduced) raw counts / sums
__global__ __launch_bounds__(256) void ema_apply_kernel(const float* __restrict__ counts, const float* __restrict__ sums,
                                                        int D, int M, float decay, float omd,
                                                        float* __restrict__ cluster_size, float* __restrict__ embed_avg) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < M) cluster_size[i] = decay * cluster_size[i] + omd * counts[i];
  if (i < (int64_t)D * M) embed_avg[i] = decay * embed_avg[i] + omd * sums[i];
}

// step 2: n = sum(cluster_size); embed = embed_avg / ((cs + eps) / (n + M eps) * n)
__global__ __launch_bounds__(256) void ema_normalize_kernel(const float* __restrict__ cluster_size,
                                                            const float* __restrict__ embed_avg, int D, int M,
                                                            float eps, float* __restrict__ embed) {
  __shared__ float red[256];
  float s = 0.f;
  for (int i = threadIdx.x; i < M; i += 256) s += cluster_size[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  const float n = red[0];
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (int64_t)D * M; i += (int64_t)gridDim.x * 256) {
    const int slot = (int)(i % M);
    const float smoothed = (cluster_size[slot] + eps) / (n + M * eps) * n;
    embed[i] = embed_avg[i] / smoothed;
  }
}

// OIHW [cout][cin][3][3] -> input-gradient filter [cin][Kpad], k = (r*3+s)*cout_p + n,
// value W[n][c][2-r][2-s]   (transposed and flipped)
__global__ __launch_bounds__(256) void pack_conv_dgrad_weight_kernel(const float* __restrict__ w, int cout, int cin,
                                                                     int cout_p, int kpad, int rows,
                                                                     float* __restrict__ out) {
  const int64_t total = (int64_t)rows * kpad;
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= total) return;
  const int k = (int)(gid % kpad);
  const int c = (int)(gid / kpad);
  const int tap = k / cout_p, n = k % cout_p;
  float v = 0.f;
  if (tap < 9 && n < cout && c < cin) v = w[((int64_t)n * cin + c) * 9 + (8 - tap)];
  out[gid] = v;
}

// All 3x3 filters of a training step in ONE launch, straight to their S16 images (round 4: the step issued ~140 launches
// of 5-8 us for this - a pack and a split per layer and direction).  items: device table, one entry per filter image;
// group_end = running total of 8-element output groups (a thread takes one group = one 32-byte S16 store, finds its
// item by binary search).  kind 0: forward filter [cout][kpad], k = tap * cin_p + c (ammc_pack_conv_weight_f32);
// kind 1: input-gradient filter [rows][kpad], k = tap * cout_p + n, value W[n][c][2-r][2-s] (ammc_pack_conv_dgrad_weight_f32).
struct PackItem {
  const float* w;        // OIHW [cout][cin][3][3]
  float* out16;          // S16 image
  int32_t cout, cin, inner_p, kpad, kind, rows;     // inner_p = cin_p (kind 0) or cout_p (kind 1); rows of the image
  int64_t group_end;
};
__global__ __launch_bounds__(256) void pack_filters_s16_kernel(const PackItem* __restrict__ items, int n_items, int64_t total) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= total) return;
  int lo = 0, hi = n_items - 1;
  while (lo < hi) {                                   // first item whose group_end > gid
    const int mid = (lo + hi) >> 1;
    if (items[mid].group_end > gid) hi = mid; else lo = mid + 1;
  }
  const PackItem it = items[lo];
  const int64_t g = gid - (lo ? items[lo - 1].group_end : 0);
  const int k8 = it.kpad >> 3;
  const int row = (int)(g / k8), k0 = (int)(g % k8) * 8;
  float v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int k = k0 + i;
    const int tap = k / it.inner_p, in = k % it.inner_p;
    float t = 0.f;
    if (it.kind == 0) {
      if (tap < 9 && in < it.cin && row < it.cout) t = it.w[((int64_t)row * it.cin + in) * 9 + tap];
    } else {
      if (tap < 9 && in < it.cout && row < it.cin) t = it.w[((int64_t)in * it.cin + row) * 9 + (8 - tap)];
    }
    v[i] = t;
  }
  ammc_u4 h, l;
  ammc_s16_split8(v, h, l);
  float* op = it.out16 + ((int64_t)row * it.kpad + k0);
  *reinterpret_cast<ammc_u4*>(op) = h;
  *reinterpret_cast<ammc_u4*>(op + 4) = l;
}

// [rows][cols] -> [cols][rows_p] zero padded (1x1 conv input-gradient filter)
__global__ __launch_bounds__(256) void transpose_pad_kernel(const float* __restrict__ w, int rows, int cols,
                                                            int rows_p, float* __restrict__ out) {
  const int64_t total = (int64_t)cols * rows_p;
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= total) return;
  const int r = (int)(gid % rows_p);
  const int c = (int)(gid / rows_p);
  out[gid] = r < rows ? w[(int64_t)r * cols + c] : 0.f;
}

// Input-gradient filters of the PixelDiscriminator's 4x4 convolutions (pix2pix_networks.py:604-621, padding 2).
// stride 1: one [rows][16*cout_p] matrix, the window flipped (k = tap'*cout_p + n, tap' = 15 - tap).
// stride 2: four matrices [phase = py*2+px][rows][4*cout_p]; input pixel (2q'+py, 2r'+px) gathers the 2x2 block of
//           output gradients ((py+pad)/2 - 1 + q' + dr, ...) through filter tap ((py+pad) % 2 + 2(1-dr), ...);
//           pad = the convolution's padding (2: PixelDiscriminator; 1: ConvTranspose2d(k 4, s 2, p 1) of FlowNet2-SD,
//           whose forward IS this input gradient).
__global__ __launch_bounds__(256) void pack_conv4_dgrad_weight_kernel(const float* __restrict__ w, int cout, int cin,
                                                                      int cout_p, int kpad, int rows, int stride, int pad,
                                                                      float* __restrict__ out) {
  const int phases = stride == 2 ? 4 : 1;
  const int64_t total = (int64_t)phases * rows * kpad;
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= total) return;
  const int k = (int)(gid % kpad);
  const int c = (int)((gid / kpad) % rows);
  const int ph = (int)(gid / ((int64_t)kpad * rows));
  const int tap = k / cout_p, n = k % cout_p;
  float v = 0.f;
  if (n < cout && c < cin) {
    if (stride == 2) {
      if (tap < 4) {
        const int r = (((ph >> 1) + pad) & 1) + 2 * (1 - (tap >> 1));
        const int s = (((ph & 1) + pad) & 1) + 2 * (1 - (tap & 1));
        v = w[((int64_t)n * cin + c) * 16 + r * 4 + s];
      }
    } else if (tap < 16) {
      v = w[((int64_t)n * cin + c) * 16 + (15 - tap)];
    }
  }
  out[gid] = v;
}

// g *= (y > 0 ? 1 : slope): autograd of nn.LeakyReLU written on the layer OUTPUT (sign(y) == sign(pre-activation))
__global__ __launch_bounds__(256) void lrelu_bwd_kernel(const float* __restrict__ y, int64_t y_bs, int64_t y_rs,
                                                        int64_t y_ps, float* __restrict__ g, int64_t g_bs, int64_t g_rs,
                                                        int64_t g_ps, int B, int H, int W, int C4, float slope) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (int64_t)B * H * W * C4) return;
  const int c4 = (int)(gid % C4);
  int64_t t = gid / C4;
  const int x = (int)(t % W);
  t /= W;
  const int yy = (int)(t % H);
  const int b = (int)(t / H);
  const f32x4 yv = *reinterpret_cast<const f32x4*>(y + b * y_bs + yy * y_rs + x * y_ps + c4 * 4);
  float* gp = g + b * g_bs + yy * g_rs + x * g_ps + c4 * 4;
  f32x4 gv = *reinterpret_cast<const f32x4*>(gp);
#pragma unroll
  for (int i = 0; i < 4; ++i) gv[i] = yv[i] > 0.f ? gv[i] : gv[i] * slope;
  *reinterpret_cast<f32x4*>(gp) = gv;
}

inline unsigned nblk(int64_t total) { return (unsigned)((total + 255) / 256); }

// ---- row forms of the two big elementwise passes (round 4) ---------------------------------------------------------
// One workgroup per image row, thread = (pixel lane, group of 8 channels) walking the row in steps of 256 items: the
// row base is uniform (scalar arithmetic), a thread's channel group never changes (256 is a multiple of C8), so the
// per-channel constants are loaded ONCE per thread and an item's offsets are the previous item's plus a constant.
// The one-item-per-thread forms above pay, per 32-48 bytes of traffic, four integer divisions, nine 64-bit multiplies
// and twelve 16-byte loads of per-channel constants (and bn_bwd_apply_s16 an eight-barrier LDS reduction of the 256
// amax slots in front of its first load).  Need C8 = 2^csh <= 256 and pixel strides below 2^24; else the forms above.
// Measured in the batch-32 training step: bn_bwd_apply_s16 150 -> 129 us per launch on average (1.6 GB in 270 us =
// 5.9 TB/s on the 256x256 level), scale_shift_act_s16 unchanged (it already ran at 5.8-6.2 TB/s).
__global__ __launch_bounds__(256) void scale_shift_act_s16_rows_kernel(
    const float* __restrict__ x, Tensor3 xt, const float* __restrict__ scale, const float* __restrict__ shift,
    const float* __restrict__ res, Tensor3 rt, float* __restrict__ y32, float* __restrict__ y16, Tensor3 yt, int relu,
    int H, int W, int csh) {
  constexpr int U = 4;
  const int row = blockIdx.x, b = row / H, yy = row - b * H;
  const float* xr = x + (int64_t)b * xt.bs + (int64_t)yy * xt.rs;
  const float* rr = res ? res + (int64_t)b * rt.bs + (int64_t)yy * rt.rs : nullptr;
  const int64_t yrow = (int64_t)b * yt.bs + (int64_t)yy * yt.rs;
  const int n = W << csh;
  const int c8 = threadIdx.x & ((1 << csh) - 1), px0 = threadIdx.x >> csh, dpx = 256 >> csh;
  const f32x4 sc0 = *reinterpret_cast<const f32x4*>(scale + c8 * 8), sc1 = *reinterpret_cast<const f32x4*>(scale + c8 * 8 + 4);
  const f32x4 sh0 = *reinterpret_cast<const f32x4*>(shift + c8 * 8), sh1 = *reinterpret_cast<const f32x4*>(shift + c8 * 8 + 4);
  int xo = px0 * (int)xt.ps + c8 * 8, ro = px0 * (int)rt.ps + c8 * 8, yo = px0 * (int)yt.ps + c8 * 8;
  const int xd = dpx * (int)xt.ps, rd = dpx * (int)rt.ps, yd = dpx * (int)yt.ps;
  typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
  for (int e = threadIdx.x; e < n; e += 256 * U) {
    f32x4 v[U][2], r[U][2];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {                     // (items past the end of the row re-load item 0 and are dropped)
      ok[u] = e + u * 256 < n;
      const float* p = xr + (ok[u] ? xo + u * xd : xo);
      v[u][0] = *reinterpret_cast<const f32x4*>(p);
      v[u][1] = *reinterpret_cast<const f32x4*>(p + 4);
      if (rr) {
        const float* q = rr + (ok[u] ? ro + u * rd : ro);
        r[u][0] = *reinterpret_cast<const f32x4*>(q);
        r[u][1] = *reinterpret_cast<const f32x4*>(q + 4);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (!ok[u]) continue;
      float o[8];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float t0 = v[u][0][i] * sc0[i] + sh0[i], t1 = v[u][1][i] * sc1[i] + sh1[i];
        if (relu) { t0 = t0 > 0.f ? t0 : 0.f; t1 = t1 > 0.f ? t1 : 0.f; }
        if (rr) { t0 += r[u][0][i]; t1 += r[u][1][i]; }
        o[i] = t0;
        o[4 + i] = t1;
      }
      const int64_t oo = yrow + yo + u * yd;
      if (y32) {
        *reinterpret_cast<f32x4*>(y32 + oo) = f32x4{o[0], o[1], o[2], o[3]};
        *reinterpret_cast<f32x4*>(y32 + oo + 4) = f32x4{o[4], o[5], o[6], o[7]};
      }
      f16x8 hi, lo;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const _Float16 hv = (_Float16)o[i];
        hi[i] = hv;
        lo[i] = (_Float16)((o[i] - (float)hv) * 2048.f);
      }
      *reinterpret_cast<f16x8*>(y16 + oo) = hi;
      *reinterpret_cast<f16x8*>(y16 + oo + 4) = lo;
    }
    xo += U * xd;
    ro += U * rd;
    yo += U * yd;
  }
}

template <bool UP>
__global__ __launch_bounds__(256) void bn_bwd_apply_s16_rows_kernel(
    const float* __restrict__ c, Tensor3 ct, const float* __restrict__ dy, Tensor3 dt,
    const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ gamma,
    const float* __restrict__ beta, const float* __restrict__ sums, float inv_count, int relu,
    float* __restrict__ dc16, float* __restrict__ dc32, Tensor3 ot, int H, int W, int csh,
    const int* __restrict__ amax_bits, float* __restrict__ inv_scale, int n_inv, Unpool up) {
  constexpr int U = 4;
  __shared__ int red[4];
  const int row = blockIdx.x, b = row / H, yy = row - b * H;
  const float* cr = c + (int64_t)b * ct.bs + (int64_t)yy * ct.rs;
  const float* gr = dy + (int64_t)b * dt.bs + (int64_t)yy * dt.rs;
  // (up.idx: dy = add + unpool(dpo), see struct Unpool) this row's pooled row and its window-position bytes
  const bool up_row = UP && (yy >> 1) < up.ph;
  const float* pr = up_row ? up.dpo + (int64_t)b * up.pt.bs + (int64_t)(yy >> 1) * up.pt.rs : nullptr;
  const unsigned char* ir = up_row ? up.idx + (((int64_t)b * up.ph + (yy >> 1)) * up.pw << (csh + 3)) : nullptr;
  int px = threadIdx.x >> csh;
  const int64_t orow = (int64_t)b * ot.bs + (int64_t)yy * ot.rs;
  const int n = W << csh;
  const int C = 8 << csh;
  const int c8 = threadIdx.x & ((1 << csh) - 1), px0 = threadIdx.x >> csh, dpx = 256 >> csh;
  int co = px0 * (int)ct.ps + c8 * 8, go = px0 * (int)dt.ps + c8 * 8, oo = px0 * (int)ot.ps + c8 * 8;
  const int cd = dpx * (int)ct.ps, gd = dpx * (int)dt.ps, od = dpx * (int)ot.ps;
  // the power of two from the 256 amax slots: one slot per thread, a maximum per wave by shuffles, four values through LDS
  int am = amax_bits[threadIdx.x];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) am = max(am, __shfl_xor(am, off));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = am;
  // per-channel constants, loaded once (the arithmetic below is the one-item form's, operation for operation)
  f32x4 mu[2], is[2], ga[2], be[2], sg[2], sgx[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int cc = c8 * 8 + h * 4;
    mu[h] = *reinterpret_cast<const f32x4*>(mean + cc);
    is[h] = *reinterpret_cast<const f32x4*>(invstd + cc);
    ga[h] = *reinterpret_cast<const f32x4*>(gamma + cc);
    be[h] = *reinterpret_cast<const f32x4*>(beta + cc);
    sg[h] = *reinterpret_cast<const f32x4*>(sums + cc);
    sgx[h] = *reinterpret_cast<const f32x4*>(sums + C + cc);
  }
  __syncthreads();
  const float f = tk_pow2_to_1024(max(max(red[0], red[1]), max(red[2], red[3])));
  if (blockIdx.x == 0)
    for (int i = threadIdx.x; i < n_inv; i += 256) inv_scale[i] = 1.f / f;
  typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
  for (int e = threadIdx.x; e < n; e += 256 * U) {
    f32x4 v[U][2], g[U][2], gp[U][2];
    unsigned long long code[U];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      ok[u] = e + u * 256 < n;
      const float* p = cr + (ok[u] ? co + u * cd : co);
      const float* q = gr + (ok[u] ? go + u * gd : go);
      v[u][0] = *reinterpret_cast<const f32x4*>(p);
      v[u][1] = *reinterpret_cast<const f32x4*>(p + 4);
      g[u][0] = *reinterpret_cast<const f32x4*>(q);
      g[u][1] = *reinterpret_cast<const f32x4*>(q + 4);
      if (up_row) {
        const int wx = (px + (ok[u] ? u * dpx : 0)) >> 1;
        const bool in = wx < up.pw;
        const int cx = in ? wx : 0;
        const float* w = pr + cx * (int)up.pt.ps + c8 * 8;
        gp[u][0] = *reinterpret_cast<const f32x4*>(w);
        gp[u][1] = *reinterpret_cast<const f32x4*>(w + 4);
        code[u] = *reinterpret_cast<const unsigned long long*>(ir + (((int64_t)cx << csh) + c8) * 8);
        if (!in) code[u] = ~0ull;
      }
    }
    if (up_row) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const unsigned pos = (unsigned)((yy & 1) * 2 + ((px + u * dpx) & 1));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if ((unsigned)((code[u] >> (8 * i)) & 0xffu) == pos) g[u][0][i] += gp[u][0][i];
          if ((unsigned)((code[u] >> (8 * (4 + i))) & 0xffu) == pos) g[u][1][i] += gp[u][1][i];
        }
      }
    }
    px += U * dpx;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (!ok[u]) continue;
      float o[8];
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float xh = (v[u][h][i] - mu[h][i]) * is[h][i];
          const float pre = v[u][h][i] * ga[h][i] + be[h][i];
          const float gi = (!relu || pre > 0.f) ? g[u][h][i] : 0.f;
          o[h * 4 + i] = ga[h][i] * (gi - sg[h][i] * inv_count - xh * sgx[h][i] * inv_count);
        }
      const int64_t op = orow + oo + u * od;
      if (dc32) {
        *reinterpret_cast<f32x4*>(dc32 + op) = f32x4{o[0], o[1], o[2], o[3]};
        *reinterpret_cast<f32x4*>(dc32 + op + 4) = f32x4{o[4], o[5], o[6], o[7]};
      }
      f16x8 hi, lo;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float w = o[i] * f;
        const _Float16 hv = (_Float16)w;
        hi[i] = hv;
        lo[i] = (_Float16)((w - (float)hv) * 2048.f);
      }
      *reinterpret_cast<f16x8*>(dc16 + op) = hi;
      *reinterpret_cast<f16x8*>(dc16 + op + 4) = lo;
    }
    co += U * cd;
    go += U * gd;
    oo += U * od;
  }
}

// scale_shift_act_s16 of a tensor that a MaxPool2d(2) follows (the second unit of `inconv` / `down`, models/unet.py:23-37):
// one workgroup per PAIR of rows, thread = (window, group of 8 channels): the four activations of the window are written
// as S16 (and fp32 when asked), and so are their 2x2 maximum - the winner's (hi, lo) pair by the decoded values, the first
// maximum in row-major order: ammc_maxpool2x2_s16_idx's rule, bit for bit - and its window position, without the pass
// that would read the full-resolution S16 tensor back (537 MB of the 671 MB that pass moves on the 256x256 level).
__global__ __launch_bounds__(256) void scale_shift_act_s16_pool_rows_kernel(
    const float* __restrict__ x, Tensor3 xt, const float* __restrict__ scale, const float* __restrict__ shift,
    float* __restrict__ y32, float* __restrict__ y16, Tensor3 yt, float* __restrict__ p16, Tensor3 pt,
    unsigned char* __restrict__ idx, int relu, int H2, int W2, int csh) {
  constexpr int U = 2;
  const int row = blockIdx.x, b = row / H2, yy = row - b * H2;              // pooled row yy of sample b
  const float* xr = x + (int64_t)b * xt.bs + (int64_t)(2 * yy) * xt.rs;
  const int64_t yrow = (int64_t)b * yt.bs + (int64_t)(2 * yy) * yt.rs;
  float* pr = p16 + (int64_t)b * pt.bs + (int64_t)yy * pt.rs;
  unsigned char* ir = idx + (((int64_t)b * H2 + yy) * W2 << csh) * 8;
  const int n = W2 << csh;
  const int c8 = threadIdx.x & ((1 << csh) - 1), px0 = threadIdx.x >> csh, dpx = 256 >> csh;
  const f32x4 sc0 = *reinterpret_cast<const f32x4*>(scale + c8 * 8), sc1 = *reinterpret_cast<const f32x4*>(scale + c8 * 8 + 4);
  const f32x4 sh0 = *reinterpret_cast<const f32x4*>(shift + c8 * 8), sh1 = *reinterpret_cast<const f32x4*>(shift + c8 * 8 + 4);
  int xo = 2 * px0 * (int)xt.ps + c8 * 8, yo = 2 * px0 * (int)yt.ps + c8 * 8, po = px0 * (int)pt.ps + c8 * 8;
  const int xd = 2 * dpx * (int)xt.ps, yd = 2 * dpx * (int)yt.ps, pd = dpx * (int)pt.ps;
  const int xw[4] = {0, (int)xt.ps, (int)xt.rs, (int)xt.rs + (int)xt.ps};   // the window, row-major
  const int yw[4] = {0, (int)yt.ps, (int)yt.rs, (int)yt.rs + (int)yt.ps};
  typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
  for (int e = threadIdx.x; e < n; e += 256 * U) {
    f32x4 v[U][4][2];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      ok[u] = e + u * 256 < n;
      const float* p = xr + (ok[u] ? xo + u * xd : xo);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v[u][j][0] = *reinterpret_cast<const f32x4*>(p + xw[j]);
        v[u][j][1] = *reinterpret_cast<const f32x4*>(p + xw[j] + 4);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (!ok[u]) continue;
      f16x8 hi[4], lo[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float o[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float t0 = v[u][j][0][i] * sc0[i] + sh0[i], t1 = v[u][j][1][i] * sc1[i] + sh1[i];
          if (relu) { t0 = t0 > 0.f ? t0 : 0.f; t1 = t1 > 0.f ? t1 : 0.f; }
          o[i] = t0;
          o[4 + i] = t1;
        }
        const int64_t oo = yrow + yo + u * yd + yw[j];
        if (y32) {
          *reinterpret_cast<f32x4*>(y32 + oo) = f32x4{o[0], o[1], o[2], o[3]};
          *reinterpret_cast<f32x4*>(y32 + oo + 4) = f32x4{o[4], o[5], o[6], o[7]};
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const _Float16 hv = (_Float16)o[i];
          hi[j][i] = hv;
          lo[j][i] = (_Float16)((o[i] - (float)hv) * 2048.f);
        }
        *reinterpret_cast<f16x8*>(y16 + oo) = hi[j];
        *reinterpret_cast<f16x8*>(y16 + oo + 4) = lo[j];
      }
      f16x8 oh, ol;
      unsigned long long args = 0;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float best = (float)hi[0][i] + (float)lo[0][i] * (1.f / 2048.f);
        _Float16 bh = hi[0][i], bl = lo[0][i];
        unsigned arg = 0;
#pragma unroll
        for (int j = 1; j < 4; ++j) {
          const float d = (float)hi[j][i] + (float)lo[j][i] * (1.f / 2048.f);
          if (d > best) { best = d; bh = hi[j][i]; bl = lo[j][i]; arg = j; }
        }
        oh[i] = bh;
        ol[i] = bl;
        args |= (unsigned long long)arg << (8 * i);
      }
      float* pp = pr + po + u * pd;
      *reinterpret_cast<f16x8*>(pp) = oh;
      *reinterpret_cast<f16x8*>(pp + 4) = ol;
      *reinterpret_cast<unsigned long long*>(ir + (int64_t)(e + u * 256) * 8) = args;
    }
    xo += U * xd;
    yo += U * yd;
    po += U * pd;
  }
}

// csh = log2(C8) when the row forms apply (AMMC_ROW_KERNELS=0 switches them off for A/Bs), else -1
static int rows_csh(int c8, int64_t ps_a, int64_t ps_b, int64_t ps_c, int w) {
  static const int on = getenv("AMMC_ROW_KERNELS") ? atoi(getenv("AMMC_ROW_KERNELS")) != 0 : 1;
  if (!on || c8 <= 0 || (c8 & (c8 - 1)) || c8 > 256) return -1;
  const int64_t lim = 1 << 24;
  if (ps_a >= lim || ps_b >= lim || ps_c >= lim || ps_a < 0 || ps_b < 0 || ps_c < 0) return -1;
  if ((int64_t)w * (ps_a > ps_b ? (ps_a > ps_c ? ps_a : ps_c) : (ps_b > ps_c ? ps_b : ps_c)) >= (1LL << 31)) return -1;
  return __builtin_ctz((unsigned)c8);
}

}  // namespace ammc_impl
using namespace ammc_impl;

extern "C" {

int ammc_chan_reduce_blocks(int32_t pixels) { return pixels <= 0 ? 0 : (pixels + red_pix_for(pixels) - 1) / red_pix_for(pixels); }

static int check_nhwc(const void* p, int b, int h, int w, int c) {
  if (!p || b <= 0 || h <= 0 || w <= 0 || c <= 0 || (c & 3) || c > 1024) return AMMC_EINVAL;
  return AMMC_OK;
}

int ammc_bn_stats_f32(const float* x, int64_t x_bs, int64_t x_rs, int64_t x_ps, int32_t batch, int32_t h, int32_t w,
                      int32_t c, float* partial, void* stream) {
  if (check_nhwc(x, batch, h, w, c) || !partial) return AMMC_EINVAL;
  const int M = batch * h * w;
  Tensor3 xt{x_bs, x_rs, x_ps}, none{0, 0, 0};
  hipLaunchKernelGGL(chan_reduce_kernel<0>, dim3(ammc_chan_reduce_blocks(M)), dim3(256), 0, (hipStream_t)stream,
                     x, xt, nullptr, none, nullptr, nullptr, nullptr, nullptr, 0, M, h, w, c, partial);
  return ammc_launch_status();
}

int ammc_bn_finalize_f32(const float* partial, int32_t nblocks, int32_t c, float count, const float* gamma,
                         const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                         float* mean, float* invstd, float* scale, float* shift, void* stream) {
  if (!partial || !gamma || !beta || !running_mean || !running_var || !mean || !invstd || !scale || !shift ||
      nblocks <= 0 || c <= 0 || count <= 0.f) return AMMC_EINVAL;
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((c + RP_COLS - 1) / RP_COLS), dim3(RP_THREADS), 0, (hipStream_t)stream, partial, nblocks, c,
                     count, gamma, beta, eps, momentum, running_mean, running_var, mean, invstd, scale, shift);
  return ammc_launch_status();
}

int ammc_scale_shift_act_f32(const float* x, int64_t x_bs, int64_t x_rs, int64_t x_ps, const float* scale,
                             const float* shift, const float* res, int64_t r_bs, int64_t r_rs, int64_t r_ps,
                             float* y, int64_t y_bs, int64_t y_rs, int64_t y_ps, int32_t relu, int32_t batch,
                             int32_t h, int32_t w, int32_t c, void* stream) {
  if (check_nhwc(x, batch, h, w, c) || !scale || !shift || !y) return AMMC_EINVAL;
  const int M = batch * h * w;
  Tensor3 xt{x_bs, x_rs, x_ps}, rt{r_bs, r_rs, r_ps}, yt{y_bs, y_rs, y_ps};
  hipLaunchKernelGGL(scale_shift_act_kernel, dim3(nblk((int64_t)M * (c >> 2))), dim3(256), 0, (hipStream_t)stream,
                     x, xt, scale, shift, res, rt, y, yt, relu, M, h, w, c >> 2);
  return ammc_launch_status();
}

int ammc_scale_shift_act_s16_f32(const float* x, int64_t x_bs, int64_t x_rs, int64_t x_ps, const float* scale,
                                 const float* shift, const float* res, int64_t r_bs, int64_t r_rs, int64_t r_ps,
                                 float* y32, float* y16, int64_t y_bs, int64_t y_rs, int64_t y_ps, int32_t relu,
                                 int32_t batch, int32_t h, int32_t w, int32_t c, void* stream) {
  if (check_nhwc(x, batch, h, w, c) || !scale || !shift || !y16 || (c & 7) || ((uintptr_t)y16 & 31) ||
      ((y_bs | y_rs | y_ps) & 7))
    return AMMC_EINVAL;
  const int M = batch * h * w;
  Tensor3 xt{x_bs, x_rs, x_ps}, rt{r_bs, r_rs, r_ps}, yt{y_bs, y_rs, y_ps};
  const int csh = rows_csh(c >> 3, x_ps, res ? r_ps : 0, y_ps, w);
  if (csh >= 0)
    hipLaunchKernelGGL(scale_shift_act_s16_rows_kernel, dim3(batch * h), dim3(256), 0, (hipStream_t)stream, x, xt, scale, shift,
                       res, rt, y32, y16, yt, relu, h, w, csh);
  else
    hipLaunchKernelGGL(scale_shift_act_s16_kernel, dim3(nblk((int64_t)M * (c >> 3))), dim3(256), 0, (hipStream_t)stream,
                       x, xt, scale, shift, res, rt, y32, y16, yt, relu, M, h, w, c >> 3);
  return ammc_launch_status();
}

int ammc_scale_shift_act_s16_pool_f32(const float* x, int64_t x_bs, int64_t x_rs, int64_t x_ps, const float* scale,
                                      const float* shift, float* y32, float* y16, int64_t y_bs, int64_t y_rs, int64_t y_ps,
                                      float* pool16, int64_t p_bs, int64_t p_rs, int64_t p_ps, uint8_t* idx, int32_t relu,
                                      int32_t batch, int32_t h, int32_t w, int32_t c, void* stream) {
  if (check_nhwc(x, batch, h, w, c) || !scale || !shift || !y16 || !pool16 || !idx || (c & 7) || ((uintptr_t)y16 & 31) ||
      ((uintptr_t)pool16 & 31) || ((uintptr_t)idx & 7) || ((y_bs | y_rs | y_ps | p_bs | p_rs | p_ps) & 7))
    return AMMC_EINVAL;
  if ((h | w) & 1) return AMMC_EUNSUP;               // MaxPool2d floors: an odd size keeps the two separate passes
  const int csh = rows_csh(c >> 3, x_ps, y_ps, p_ps, w);
  if (csh < 0 || x_rs >= (1 << 30) || y_rs >= (1 << 30)) return AMMC_EUNSUP;
  Tensor3 xt{x_bs, x_rs, x_ps}, yt{y_bs, y_rs, y_ps}, pt{p_bs, p_rs, p_ps};
  hipLaunchKernelGGL(scale_shift_act_s16_pool_rows_kernel, dim3(batch * (h >> 1)), dim3(256), 0, (hipStream_t)stream, x, xt,
                     scale, shift, y32, y16, yt, pool16, pt, idx, relu, h >> 1, w >> 1, csh);
  return ammc_launch_status();
}

int ammc_bn_bwd_reduce_f32(const float* c_raw, int64_t c_bs, int64_t c_rs, int64_t c_ps, const float* dy,
                           int64_t d_bs, int64_t d_rs, int64_t d_ps, const float* mean, const float* invstd,
                           const float* gamma, const float* beta, int32_t relu, int32_t batch, int32_t h, int32_t w,
                           int32_t c, float* partial, void* stream) {
  if (check_nhwc(c_raw, batch, h, w, c) || !dy || !mean || !invstd || !gamma || !beta || !partial) return AMMC_EINVAL;
  const int M = batch * h * w;
  Tensor3 ct{c_bs, c_rs, c_ps}, dt{d_bs, d_rs, d_ps};
  hipLaunchKernelGGL(chan_reduce_kernel<1>, dim3(ammc_chan_reduce_blocks(M)), dim3(256), 0, (hipStream_t)stream,
                     c_raw, ct, dy, dt, mean, invstd, gamma, beta, relu, M, h, w, c, partial);
  return ammc_launch_status();
}

int ammc_bn_bwd_apply_f32(const float* c_raw, int64_t c_bs, int64_t c_rs, int64_t c_ps, const float* dy,
                          int64_t d_bs, int64_t d_rs, int64_t d_ps, const float* mean, const float* invstd,
                          const float* gamma, const float* beta, const float* sums, int32_t relu, float* dc,
                          int64_t o_bs, int64_t o_rs, int64_t o_ps, int32_t batch, int32_t h, int32_t w, int32_t c,
                          int32_t* amax_bits, void* stream) {
  if (check_nhwc(c_raw, batch, h, w, c) || !dy || !mean || !invstd || !gamma || !beta || !sums || !dc) return AMMC_EINVAL;
  const int M = batch * h * w;
  Tensor3 ct{c_bs, c_rs, c_ps}, dt{d_bs, d_rs, d_ps}, ot{o_bs, o_rs, o_ps};
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(nblk((int64_t)M * (c >> 2))), dim3(256), 0, (hipStream_t)stream,
                     c_raw, ct, dy, dt, mean, invstd, gamma, beta, sums, 1.f / (float)M, relu, dc, ot, M, h, w, c >> 2, amax_bits);
  return ammc_launch_status();
}

int ammc_bn_bwd_reduce_bound_f32(const float* c_raw, int64_t c_bs, int64_t c_rs, int64_t c_ps, const float* dy,
                                 int64_t d_bs, int64_t d_rs, int64_t d_ps, const float* mean, const float* invstd,
                                 const float* gamma, const float* beta, int32_t relu, int32_t batch, int32_t h,
                                 int32_t w, int32_t c, float* partial, void* stream) {
  if (check_nhwc(c_raw, batch, h, w, c) || !dy || !mean || !invstd || !gamma || !beta || !partial) return AMMC_EINVAL;
  const int M = batch * h * w;
  Tensor3 ct{c_bs, c_rs, c_ps}, dt{d_bs, d_rs, d_ps};
  hipLaunchKernelGGL(chan_reduce_kernel<3>, dim3(ammc_chan_reduce_blocks(M)), dim3(256), 0, (hipStream_t)stream,
                     c_raw, ct, dy, dt, mean, invstd, gamma, beta, relu, M, h, w, c, partial);
  return ammc_launch_status();
}

int ammc_bn_bwd_finalize_f32(const float* partial, int32_t nblocks, int32_t c, int32_t pixels, const float* gamma,
                             float* sums, int32_t* amax_bits, void* stream) {
  if (!partial || !gamma || !sums || !amax_bits || nblocks <= 0 || c <= 0 || pixels <= 0) return AMMC_EINVAL;
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((c + RP_COLS - 1) / RP_COLS), dim3(RP_THREADS), 0, (hipStream_t)stream,
                     partial, nblocks, c, 1.f / (float)pixels, gamma, sums, amax_bits);
  return ammc_launch_status();
}

int ammc_bn_bwd_apply_s16_f32(const float* c_raw, int64_t c_bs, int64_t c_rs, int64_t c_ps, const float* dy,
                              int64_t d_bs, int64_t d_rs, int64_t d_ps, const float* mean, const float* invstd,
                              const float* gamma, const float* beta, const float* sums, int32_t relu, float* dc16,
                              float* dc32, int64_t o_bs, int64_t o_rs, int64_t o_ps, int32_t batch, int32_t h,
                              int32_t w, int32_t c, const int32_t* amax_bits, float* inv_scale, int32_t n_inv,
                              void* stream) {
  if (check_nhwc(c_raw, batch, h, w, c) || !dy || !mean || !invstd || !gamma || !beta || !sums || !dc16 || !amax_bits ||
      !inv_scale || n_inv <= 0 || (c & 7) || ((uintptr_t)dc16 & 31) || ((o_bs | o_rs | o_ps) & 7))
    return AMMC_EINVAL;
  const int M = batch * h * w;
  Tensor3 ct{c_bs, c_rs, c_ps}, dt{d_bs, d_rs, d_ps}, ot{o_bs, o_rs, o_ps};
  const int csh = rows_csh(c >> 3, c_ps, d_ps, o_ps, w);
  if (csh >= 0)
    hipLaunchKernelGGL(bn_bwd_apply_s16_rows_kernel<false>, dim3(batch * h), dim3(256), 0, (hipStream_t)stream, c_raw, ct, dy, dt, mean,
                       invstd, gamma, beta, sums, 1.f / (float)M, relu, dc16, dc32, ot, h, w, csh, amax_bits, inv_scale, n_inv,
                       Unpool{nullptr, {0, 0, 0}, nullptr, 0, 0});
  else
    hipLaunchKernelGGL(bn_bwd_apply_s16_kernel, dim3(nblk((int64_t)M * (c >> 3))), dim3(256), 0, (hipStream_t)stream, c_raw,
                       ct, dy, dt, mean, invstd, gamma, beta, sums, 1.f / (float)M, relu, dc16, dc32, ot, M, h, w, c >> 3,
                       amax_bits, inv_scale, n_inv);
  return ammc_launch_status();
}

// The two passes above with dy = add + MaxPool2d(2)-backward(dpo) formed on the fly (struct Unpool): the unit's output was
// pooled in the forward, `add` is the gradient that reaches it through the skip path.
static int unpool_args_bad(const float* dpo, int64_t p_bs, int64_t p_rs, int64_t p_ps, const uint8_t* idx, int ph, int pw, int h,
                           int w, int c) {
  return !dpo || !idx || ph != (h >> 1) || pw != (w >> 1) || ph <= 0 || pw <= 0 || (c & 7) || ((uintptr_t)dpo & 15) ||
         ((uintptr_t)idx & 7) || ((p_bs | p_rs | p_ps) & 3);
}

int ammc_scale_shift_act_s16_pool_supported(int32_t c, int32_t h, int32_t w, int64_t x_rs, int64_t x_ps, int64_t y_rs,
                                            int64_t y_ps, int64_t p_ps) {
  if (c <= 0 || (c & 7) || ((h | w) & 1)) return 0;
  return rows_csh(c >> 3, x_ps, y_ps, p_ps, w) >= 0 && x_rs < (1 << 30) && y_rs < (1 << 30);
}

int ammc_bn_bwd_unpool_supported(int32_t c, int64_t c_ps, int64_t d_ps, int64_t o_ps, int32_t w) {
  return c > 0 && !(c & 7) && rows_csh(c >> 3, c_ps, d_ps, o_ps, w) >= 0;
}

int ammc_bn_bwd_reduce_bound_unpool_f32(const float* c_raw, int64_t c_bs, int64_t c_rs, int64_t c_ps, const float* add,
                                        int64_t d_bs, int64_t d_rs, int64_t d_ps, const float* dpo, int64_t p_bs, int64_t p_rs,
                                        int64_t p_ps, const uint8_t* idx, int32_t ph, int32_t pw, const float* mean,
                                        const float* invstd, const float* gamma, const float* beta, int32_t relu, int32_t batch,
                                        int32_t h, int32_t w, int32_t c, float* partial, void* stream) {
  if (check_nhwc(c_raw, batch, h, w, c) || !add || !mean || !invstd || !gamma || !beta || !partial) return AMMC_EINVAL;
  if (unpool_args_bad(dpo, p_bs, p_rs, p_ps, idx, ph, pw, h, w, c)) return AMMC_EINVAL;
  const int M = batch * h * w;
  Tensor3 ct{c_bs, c_rs, c_ps}, dt{d_bs, d_rs, d_ps};
  hipLaunchKernelGGL((chan_reduce_kernel<3, true>), dim3(ammc_chan_reduce_blocks(M)), dim3(256), 0, (hipStream_t)stream,
                     c_raw, ct, add, dt, mean, invstd, gamma, beta, relu, M, h, w, c, partial, nullptr,
                     Unpool{dpo, {p_bs, p_rs, p_ps}, idx, ph, pw});
  return ammc_launch_status();
}

int ammc_bn_bwd_apply_s16_unpool_f32(const float* c_raw, int64_t c_bs, int64_t c_rs, int64_t c_ps, const float* add,
                                     int64_t d_bs, int64_t d_rs, int64_t d_ps, const float* dpo, int64_t p_bs, int64_t p_rs,
                                     int64_t p_ps, const uint8_t* idx, int32_t ph, int32_t pw, const float* mean,
                                     const float* invstd, const float* gamma, const float* beta, const float* sums,
                                     int32_t relu, float* dc16, float* dc32, int64_t o_bs, int64_t o_rs, int64_t o_ps,
                                     int32_t batch, int32_t h, int32_t w, int32_t c, const int32_t* amax_bits,
                                     float* inv_scale, int32_t n_inv, void* stream) {
  if (check_nhwc(c_raw, batch, h, w, c) || !add || !mean || !invstd || !gamma || !beta || !sums || !dc16 || !amax_bits ||
      !inv_scale || n_inv <= 0 || (c & 7) || ((uintptr_t)dc16 & 31) || ((o_bs | o_rs | o_ps) & 7))
    return AMMC_EINVAL;
  if (unpool_args_bad(dpo, p_bs, p_rs, p_ps, idx, ph, pw, h, w, c)) return AMMC_EINVAL;
  const int csh = rows_csh(c >> 3, c_ps, d_ps, o_ps, w);
  if (csh < 0 || p_ps >= (1 << 24)) return AMMC_EUNSUP;
  const int M = batch * h * w;
  Tensor3 ct{c_bs, c_rs, c_ps}, dt{d_bs, d_rs, d_ps}, ot{o_bs, o_rs, o_ps};
  hipLaunchKernelGGL(bn_bwd_apply_s16_rows_kernel<true>, dim3(batch * h), dim3(256), 0, (hipStream_t)stream, c_raw, ct, add, dt, mean,
                     invstd, gamma, beta, sums, 1.f / (float)M, relu, dc16, dc32, ot, h, w, csh, amax_bits, inv_scale, n_inv,
                     Unpool{dpo, {p_bs, p_rs, p_ps}, idx, ph, pw});
  return ammc_launch_status();
}

int ammc_chan_sum_f32(const float* x, int64_t x_bs, int64_t x_rs, int64_t x_ps, int32_t batch, int32_t h, int32_t w,
                      int32_t c, float* partial, void* stream) {
  if (check_nhwc(x, batch, h, w, c) || !partial) return AMMC_EINVAL;
  const int M = batch * h * w;
  Tensor3 xt{x_bs, x_rs, x_ps}, none{0, 0, 0};
  hipLaunchKernelGGL(chan_reduce_kernel<2>, dim3(ammc_chan_reduce_blocks(M)), dim3(256), 0, (hipStream_t)stream,
                     x, xt, nullptr, none, nullptr, nullptr, nullptr, nullptr, 0, M, h, w, c, partial);
  return ammc_launch_status();
}

int ammc_chan_sum_absmax_f32(const float* x, int64_t x_bs, int64_t x_rs, int64_t x_ps, int32_t batch, int32_t h, int32_t w,
                             int32_t c, float* partial, int32_t* amax_bits, void* stream) {
  if (check_nhwc(x, batch, h, w, c) || !partial || !amax_bits) return AMMC_EINVAL;
  const int M = batch * h * w;
  Tensor3 xt{x_bs, x_rs, x_ps}, none{0, 0, 0};
  hipLaunchKernelGGL(chan_reduce_kernel<4>, dim3(ammc_chan_reduce_blocks(M)), dim3(256), 0, (hipStream_t)stream,
                     x, xt, nullptr, none, nullptr, nullptr, nullptr, nullptr, 0, M, h, w, c, partial, amax_bits);
  return ammc_launch_status();
}

int ammc_split_scaled_strided_f32(const float* x, int64_t x_bs, int64_t x_rs, int64_t x_ps, float* y16, int64_t y_bs,
                                  int64_t y_rs, int64_t y_ps, int32_t batch, int32_t h, int32_t w, int32_t c,
                                  const int32_t* amax_bits, float* inv_scale, int32_t n, void* stream) {
  if (check_nhwc(x, batch, h, w, c) || !y16 || !amax_bits || !inv_scale || n <= 0 || (c & 7)) return AMMC_EINVAL;
  if (((uintptr_t)y16 & 31) || ((y_bs | y_rs | y_ps) & 7) || ((uintptr_t)x & 15) || ((x_bs | x_rs | x_ps) & 3)) return AMMC_EINVAL;
  const int M = batch * h * w;
  Tensor3 xt{x_bs, x_rs, x_ps}, yt{y_bs, y_rs, y_ps};
  hipLaunchKernelGGL(split_scaled_strided_kernel, dim3(nblk((int64_t)M * (c >> 3))), dim3(256), 0, (hipStream_t)stream, x, xt,
                     y16, yt, M, h, w, c >> 3, amax_bits, inv_scale, n);
  return ammc_launch_status();
}

int ammc_reduce_partials_f32(const float* partial, int32_t nblocks, int32_t qc, float scale, float* out, void* stream) {
  if (!partial || !out || nblocks <= 0 || qc <= 0) return AMMC_EINVAL;
  hipLaunchKernelGGL(reduce_partials_kernel, dim3((qc + RP_COLS - 1) / RP_COLS), dim3(RP_THREADS), 0, (hipStream_t)stream, partial, nblocks, qc,
                     scale, out, 0, qc);
  return ammc_launch_status();
}

int ammc_reduce_partials_seg_f32(const float* partial, int32_t nblocks, int32_t qc, int32_t seg_rows, int32_t max_from,
                                 float* out, void* stream) {
  if (!partial || !out || nblocks <= 0 || qc <= 0 || seg_rows <= 0 || max_from < 0 || max_from > qc || (max_from & 7)) return AMMC_EINVAL;
  const int nseg = (nblocks + seg_rows - 1) / seg_rows;
  if (nseg > 65535) return AMMC_EUNSUP;
  hipLaunchKernelGGL(reduce_partials_kernel, dim3((qc + RP_COLS - 1) / RP_COLS, nseg), dim3(RP_THREADS), 0, (hipStream_t)stream, partial,
                     nblocks, qc, 1.f, out, seg_rows, max_from);
  return ammc_launch_status();
}

int ammc_maxpool2x2_bwd_f32(const float* x, int64_t x_bs, int64_t x_rs, int64_t x_ps, const float* dp, int64_t p_bs,
                            int64_t p_rs, int64_t p_ps, const float* add, int64_t a_bs, int64_t a_rs, int64_t a_ps,
                            float* dx, int64_t o_bs, int64_t o_rs, int64_t o_ps, int32_t batch, int32_t h, int32_t w,
                            int32_t in_h, int32_t in_w, int32_t c, void* stream) {
  if (!x || !dp || !dx || batch <= 0 || h <= 0 || w <= 0 || c <= 0 || (c & 3)) return AMMC_EINVAL;
  if ((in_h >> 1) != h || (in_w >> 1) != w) return AMMC_EINVAL;
  const int M = batch * ((in_h + 1) >> 1) * ((in_w + 1) >> 1);
  Tensor3 xt{x_bs, x_rs, x_ps}, pt{p_bs, p_rs, p_ps}, at{a_bs, a_rs, a_ps}, ot{o_bs, o_rs, o_ps};
  hipLaunchKernelGGL(maxpool_bwd_kernel<false>, dim3(nblk((int64_t)M * (c >> 2))), dim3(256), 0, (hipStream_t)stream, x, xt,
                     dp, pt, add, at, dx, ot, M, h, w, in_h, in_w, c >> 2);
  return ammc_launch_status();
}

int ammc_maxpool2x2_bwd_s16x_f32(const float* x16, int64_t x_bs, int64_t x_rs, int64_t x_ps, const float* dp, int64_t p_bs,
                                 int64_t p_rs, int64_t p_ps, const float* add, int64_t a_bs, int64_t a_rs, int64_t a_ps,
                                 float* dx, int64_t o_bs, int64_t o_rs, int64_t o_ps, int32_t batch, int32_t h, int32_t w,
                                 int32_t in_h, int32_t in_w, int32_t c, void* stream) {
  if (!x16 || !dp || !dx || batch <= 0 || h <= 0 || w <= 0 || c <= 0 || (c & 7)) return AMMC_EINVAL;
  if (((uintptr_t)x16 & 31) || ((x_bs | x_rs | x_ps) & 7)) return AMMC_EINVAL;
  if ((in_h >> 1) != h || (in_w >> 1) != w) return AMMC_EINVAL;
  const int M = batch * ((in_h + 1) >> 1) * ((in_w + 1) >> 1);
  Tensor3 xt{x_bs, x_rs, x_ps}, pt{p_bs, p_rs, p_ps}, at{a_bs, a_rs, a_ps}, ot{o_bs, o_rs, o_ps};
  hipLaunchKernelGGL(maxpool_bwd_kernel<true>, dim3(nblk((int64_t)M * (c >> 2))), dim3(256), 0, (hipStream_t)stream, x16, xt,
                     dp, pt, add, at, dx, ot, M, h, w, in_h, in_w, c >> 2);
  return ammc_launch_status();
}

int ammc_maxpool2x2_bwd_idx_f32(const uint8_t* idx, const float* dp, int64_t p_bs, int64_t p_rs, int64_t p_ps, const float* add,
                                int64_t a_bs, int64_t a_rs, int64_t a_ps, float* dx, int64_t o_bs, int64_t o_rs, int64_t o_ps,
                                int32_t batch, int32_t h, int32_t w, int32_t in_h, int32_t in_w, int32_t c, void* stream) {
  if (!idx || !dp || !dx || batch <= 0 || h <= 0 || w <= 0 || c <= 0 || (c & 7) || ((uintptr_t)idx & 7)) return AMMC_EINVAL;
  if ((in_h >> 1) != h || (in_w >> 1) != w) return AMMC_EINVAL;
  if (((uintptr_t)dp | (uintptr_t)dx | (uintptr_t)add) & 15) return AMMC_EINVAL;
  if ((p_bs | p_rs | p_ps | o_bs | o_rs | o_ps | (add ? (a_bs | a_rs | a_ps) : 0)) & 3) return AMMC_EINVAL;
  const int M = batch * ((in_h + 1) >> 1) * ((in_w + 1) >> 1);
  Tensor3 pt{p_bs, p_rs, p_ps}, at{a_bs, a_rs, a_ps}, ot{o_bs, o_rs, o_ps};
  hipLaunchKernelGGL(maxpool_bwd_idx_kernel, dim3(nblk((int64_t)M * (c >> 3))), dim3(256), 0, (hipStream_t)stream, idx, dp, pt,
                     add, at, dx, ot, M, h, w, in_h, in_w, c >> 3);
  return ammc_launch_status();
}

int ammc_tanh_bwd_nhwc_f32(const float* dout_nchw, const float* out_nchw, int32_t batch, int32_t c, int32_t h,
                           int32_t w, float* y, int64_t y_bs, int64_t y_rs, int64_t y_ps, int32_t cp, void* stream) {
  if (!dout_nchw || !out_nchw || !y || batch <= 0 || c <= 0 || h <= 0 || w <= 0 || cp < c || (cp & 3)) return AMMC_EINVAL;
  Tensor3 yt{y_bs, y_rs, y_ps};
  const int64_t total = (int64_t)batch * (cp >> 2) * h * w;
  hipLaunchKernelGGL(tanh_bwd_kernel, dim3(nblk(total)), dim3(256), 0, (hipStream_t)stream, dout_nchw, out_nchw, batch,
                     c, h, w, y, yt, cp >> 2);
  return ammc_launch_status();
}

int ammc_commit_bwd_f32(const float* z, const float* embed_md, const int32_t* idx_topk, int32_t k,
                        const float* ddiff, const float* dq, float* dz, int32_t n, int32_t d, void* stream) {
  if (!z || !embed_md || !idx_topk || !dz || n <= 0 || d <= 0 || (d & 3) || k <= 0) return AMMC_EINVAL;
  hipLaunchKernelGGL(commit_bwd_kernel, dim3(nblk((int64_t)n * (d >> 2))), dim3(256), 0, (hipStream_t)stream, z,
                     embed_md, idx_topk, k, ddiff, 2.f / ((float)n * (float)d), dq, dz, n, d >> 2);
  return ammc_launch_status();
}

int ammc_codebook_ema_f32(const float* x, const int32_t* idx_topk, int32_t k, int32_t n, int32_t d, int32_t m,
                          float decay, float one_minus_decay, float eps, float* cluster_size, float* embed_avg,
                          float* embed, void* stream) {
  if (!x || !idx_topk || !cluster_size || !embed_avg || !embed || n <= 0 || d <= 0 || m <= 0 || k <= 0) return AMMC_EINVAL;
  if (d > 256) return AMMC_EUNSUP;
  hipLaunchKernelGGL(ema_accumulate_kernel, dim3(m), dim3(64 * EMA_NW), 0, (hipStream_t)stream, x, idx_topk, k, n, d, m, decay,
                     one_minus_decay, cluster_size, embed_avg, 0);
  hipLaunchKernelGGL(ema_normalize_kernel, dim3(64), dim3(256), 0, (hipStream_t)stream, cluster_size, embed_avg, d, m,
                     eps, embed);
  return ammc_launch_status();
}

int ammc_codebook_count_f32(const float* x, const int32_t* idx_topk, int32_t k, int32_t n, int32_t d, int32_t m,
                            float* counts, float* sums, void* stream) {
  if (!x || !idx_topk || !counts || !sums || n <= 0 || d <= 0 || m <= 0 || k <= 0) return AMMC_EINVAL;
  if (d > 256) return AMMC_EUNSUP;
  hipLaunchKernelGGL(ema_accumulate_kernel, dim3(m), dim3(64 * EMA_NW), 0, (hipStream_t)stream, x, idx_topk, k, n, d, m, 0.f, 0.f,
                     counts, sums, 1);
  return ammc_launch_status();
}

int ammc_codebook_ema_apply_f32(const float* counts, const float* sums, int32_t d, int32_t m, float decay,
                                float one_minus_decay, float eps, float* cluster_size, float* embed_avg, float* embed,
                                void* stream) {
  if (!counts || !sums || !cluster_size || !embed_avg || !embed || d <= 0 || m <= 0) return AMMC_EINVAL;
  hipLaunchKernelGGL(ema_apply_kernel, dim3(nblk((int64_t)d * m)), dim3(256), 0, (hipStream_t)stream, counts, sums, d, m,
                     decay, one_minus_decay, cluster_size, embed_avg);
  hipLaunchKernelGGL(ema_normalize_kernel, dim3(64), dim3(256), 0, (hipStream_t)stream, cluster_size, embed_avg, d, m,
                     eps, embed);
  return ammc_launch_status();
}

int ammc_pack_conv_dgrad_weight_f32(const float* w_oihw, int32_t cout, int32_t cin, int32_t cout_p, int32_t rows,
                                    float* out, void* stream) {
  if (!w_oihw || !out || cout <= 0 || cin <= 0 || cout_p < cout || rows < cin) return AMMC_EINVAL;
  const int kpad = ((9 * cout_p + 31) / 32) * 32;
  hipLaunchKernelGGL(pack_conv_dgrad_weight_kernel, dim3(nblk((int64_t)rows * kpad)), dim3(256), 0,
                     (hipStream_t)stream, w_oihw, cout, cin, cout_p, kpad, rows, out);
  return ammc_launch_status();
}

int ammc_pack_filters_item_bytes(void) { return (int)sizeof(PackItem); }

int ammc_pack_filters_s16(const void* items_dev, int32_t n_items, int64_t total_groups, void* stream) {
  if (!items_dev || n_items <= 0 || total_groups <= 0 || ((uintptr_t)items_dev & 7)) return AMMC_EINVAL;
  hipLaunchKernelGGL(pack_filters_s16_kernel, dim3(nblk(total_groups)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const PackItem*>(items_dev), n_items, total_groups);
  return ammc_launch_status();
}

int ammc_pack_conv4_dgrad_weight_f32(const float* w_oihw, int32_t cout, int32_t cin, int32_t cout_p, int32_t rows,
                                     int32_t stride, int32_t pad, float* out, void* stream) {
  if (!w_oihw || !out || cout <= 0 || cin <= 0 || cout_p < cout || rows < cin || (stride != 1 && stride != 2) ||
      pad < 0 || pad > 2)
    return AMMC_EINVAL;
  const int taps = stride == 2 ? 4 : 16;
  const int kpad = ((taps * cout_p + 31) / 32) * 32;
  const int64_t total = (int64_t)(stride == 2 ? 4 : 1) * rows * kpad;
  hipLaunchKernelGGL(pack_conv4_dgrad_weight_kernel, dim3(nblk(total)), dim3(256), 0, (hipStream_t)stream, w_oihw, cout,
                     cin, cout_p, kpad, rows, stride, pad, out);
  return ammc_launch_status();
}

int ammc_lrelu_bwd_f32(const float* y, int64_t y_bs, int64_t y_rs, int64_t y_ps, float* g, int64_t g_bs, int64_t g_rs,
                       int64_t g_ps, int32_t batch, int32_t h, int32_t w, int32_t c, float slope, void* stream) {
  if (!y || !g || batch <= 0 || h <= 0 || w <= 0 || c <= 0 || (c & 3)) return AMMC_EINVAL;
  if (((uintptr_t)y | (uintptr_t)g) & 15) return AMMC_EINVAL;
  if ((y_bs | y_rs | y_ps | g_bs | g_rs | g_ps) & 3) return AMMC_EINVAL;
  hipLaunchKernelGGL(lrelu_bwd_kernel, dim3(nblk((int64_t)batch * h * w * (c >> 2))), dim3(256), 0, (hipStream_t)stream,
                     y, y_bs, y_rs, y_ps, g, g_bs, g_rs, g_ps, batch, h, w, c >> 2, slope);
  return ammc_launch_status();
}

int ammc_transpose_pad_f32(const float* w, int32_t rows, int32_t cols, int32_t rows_p, float* out, void* stream) {
  if (!w || !out || rows <= 0 || cols <= 0 || rows_p < rows) return AMMC_EINVAL;
  hipLaunchKernelGGL(transpose_pad_kernel, dim3(nblk((int64_t)cols * rows_p)), dim3(256), 0, (hipStream_t)stream, w,
                     rows, cols, rows_p, out);
  return ammc_launch_status();
}

}  // extern "C"
