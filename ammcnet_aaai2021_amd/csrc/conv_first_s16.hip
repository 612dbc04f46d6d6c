// The first layer of a stream (`inconv`'s first conv: Conv2d(12 | 6 -> 64, 3x3, pad 1, no bias) + BatchNorm(eval) + ReLU,
// reference models/unet.py:11-13, 23-30) straight from the module-boundary tensor: NCHW fp32 in, S16 NHWC out.
//
// Why its own kernel: with 12 (6) input channels the layer is pure HBM traffic - 50 MB of clips in, 268 MB of
// activations out at batch 16 - and it ran as two launches, a layout kernel (NCHW -> halo-padded S16 NHWC, 16 channels)
// and the implicit-GEMM kernel at ~2.1 TB/s: 183 us (rgb) / 145 us (flow) for ~65 us of traffic.  Here a workgroup
// reads the 10 x 34 halo patch of its 8 x 32 output pixels directly from the NCHW planes (zero outside the image: the
// layout kernel and its intermediate tensor are gone), splits it into S16 in registers, and keeps it in LDS next to
// ALL the filters of the layer (64 x 9 taps x 16 channels, S16: 40 KB): one barrier, then 240 MFMAs per wave with no
// further synchronisation, then the usual register epilogue.
//
// MFMA view (v_mfma_f32_16x16x32_f16, 3 per product as everywhere in the S16 kernels): K = 9 taps x 16 channels
// (12 or 6 real, the rest zero) = 144, padded to five 32-deep blocks; k-group q = 4 kb + g of block kb is channel
// group q & 1 of tap q >> 1, so the pixel fragment of lane (pixel l15, g) is 8 channels of the halo pixel shifted by
// that lane's tap - an im2col that costs one address per lane.  LDS images:
//   patch    pixel P holds four 16-byte slots (group 0 hi, group 0 lo, group 1 hi, group 1 lo) at chunk 4 P + (slot ^
//            ((P >> 2) & 1)): conflict free in every 16-lane group of ds_read_b128 for any patch offset and tap pair
//            (exhaustive search, as for the tap kernel's layout);
//   filters  [q][hi | lo][64 filters][16 B], filters in the tile order of conv_tap_s16.hip (tile t row r = filter
//            32 (t >> 1) + 8 (r >> 2) + 4 (t & 1) + (r & 3)): 16 lanes read 16 consecutive chunks; a lane's accumulators
//            of tiles 2 u, 2 u + 1 are 8 consecutive channels = one 32-byte S16 store.
#include "ammc_common.h"
#include <hip/hip_fp16.h>

namespace ammc_s16 {

typedef _Float16 f16x8v __attribute__((ext_vector_type(8)));

constexpr float F_LO_SCALE = 2048.f;
constexpr float F_LO_INV = 1.f / 2048.f;
constexpr int F_TH = 8, F_TW = 32, F_HW = F_TW + 2, F_HP = (F_TH + 2) * F_HW;      // 340 halo pixels
constexpr int F_NQ = 20;                                                            // k-groups of 8 (five blocks of four)
constexpr int F_WFLOATS = F_NQ * 2 * 64 * 4;                                        // filter image: 40 KB
constexpr int F_PFLOATS = F_HP * 16;                                                // patch image: 21.25 KB

struct FirstArgs {
  const float* x;          // NCHW fp32
  const float* w;          // filter image of ammc_pack_first_conv_f32
  const float* scale;
  const float* shift;
  float* y;                // S16 NHWC, pixel (0, 0)
  int32_t* overflow_flag;
  int64_t y_bs, y_rs, y_ps;
  int64_t x_bs;            // elements between consecutive batch entries of x (c * h * w when x is contiguous)
  int batch, c, h, w_, act, tiles_x, tiles_y;
};

// the halo pixels P = tid and 256 + tid of tile t: 16 channels each from the NCHW planes (zero outside the image / beyond
// c).  Branch free - every load goes to a clamped, valid address and is zeroed afterwards - so that the loop body stays
// one scheduling region and the compiler counts its vmcnt waits exactly (with a branch per load it fell back to
// vmcnt(0) at the loop head, i.e. waited for the previous tile's stores).  CT = the channel count when known (12, 6).
template <int CT>
__device__ __forceinline__ int first_load_patch(const FirstArgs& a, int t, int tid, float (&v)[2][16]) {
  int sp = t;
  const int tx = sp % a.tiles_x;
  sp /= a.tiles_x;
  const int ty = sp % a.tiles_y;
  const int b = sp / a.tiles_y;
  const int plane = a.h * a.w_;
  const int nc = CT ? CT : a.c;
  const float* xb = a.x + (int64_t)b * a.x_bs;
  int inb = 0;                                     // bit rnd: that halo pixel lies inside the image
#pragma unroll
  for (int rnd = 0; rnd < 2; ++rnd) {
    const int P = rnd * 256 + tid;
    const int hy = P / F_HW, hx = P - hy * F_HW;
    const int yy = ty * F_TH - 1 + hy, xx = tx * F_TW - 1 + hx;
    const bool in = P < F_HP && yy >= 0 && yy < a.h && xx >= 0 && xx < a.w_;
    const int pix = in ? yy * a.w_ + xx : 0;
    inb |= in ? (1 << rnd) : 0;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      if (CT && c >= CT) {
        v[rnd][c] = 0.f;
      } else {
        const int cc = CT ? c : (c < nc ? c : nc - 1);
        v[rnd][c] = xb[cc * plane + pix];           // zeroed (outside the image, c >= nc) in first_store_patch: the
      }                                             // values must not be touched before they are needed
    }
  }
  return inb;
}

// split into (hi, lo) and park in the patch image: four 16-byte slots per pixel
__device__ __forceinline__ void first_store_patch(float* Ps, int tid, const float (&v)[2][16], int inb, int nc) {
#pragma unroll
  for (int rnd = 0; rnd < 2; ++rnd) {
    const int P = rnd * 256 + tid;
    if (P < F_HP) {
      f16x8v hi[2], lo[2];
      const bool in = (inb >> rnd) & 1;
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        float val[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) val[c] = (in && 8 * g + c < nc) ? v[rnd][8 * g + c] : 0.f;
        ammc_u4 h, l;
        ammc_s16_split8(val, h, l);
        hi[g] = __builtin_bit_cast(f16x8v, h);
        lo[g] = __builtin_bit_cast(f16x8v, l);
      }
      const int sw = (P >> 2) & 1;
      float* pp = Ps + P * 16;
      *reinterpret_cast<f16x8v*>(pp + ((0 ^ sw) << 2)) = hi[0];
      *reinterpret_cast<f16x8v*>(pp + ((1 ^ sw) << 2)) = lo[0];
      *reinterpret_cast<f16x8v*>(pp + ((2 ^ sw) << 2)) = hi[1];
      *reinterpret_cast<f16x8v*>(pp + ((3 ^ sw) << 2)) = lo[1];
    }
  }
}

// Persistent over tiles (two workgroups per CU, tile t = blockIdx.x + k gridDim.x): the 40-KB filter image is loaded once
// per workgroup, the NEXT tile's patch is fetched into registers while the current one is contracted and stored, and
// the barriers are raw s_barriers (a __syncthreads would wait for the epilogue's stores).  One tile per workgroup, as
// first built: 107 us per stream at batch 16; this form 88 us.  Ablations (same box, per launch): without the stores
// 71 us, without the MFMAs 74, without both 40, and with every patch load a cache hit on top 38 - the three parts
// (VALU + LDS + barriers, 26 us of matrix pipe, 40 us of HBM stores) add up instead of overlapping at two 4-wave
// workgroups per CU, and more workgroups do not fit beside the 40-KB filter image.
template <int CT>
__global__ __launch_bounds__(256, CT ? 2 : 1) void conv_first_s16_kernel(FirstArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ws = smem;
  float* Ps = smem + F_WFLOATS;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g4 = lane >> 4;
  const int total = a.batch * a.tiles_y * a.tiles_x;

  // ---- filters: one linear DMA of the pre-arranged image (10 rounds of 256 x 16 B) ------------------------------------
#pragma unroll
  for (int j = 0; j < F_WFLOATS / 4 / 256; ++j)
    __builtin_amdgcn_global_load_lds(a.w + (j * 256 + tid) * 4, Ws + (j * 256 + wave * 64) * 4, 16, 0, 0);

  // BatchNorm scale / shift: 2 x 64 floats in LDS (as registers they were 32 VGPRs too many for two waves per SIMD; as
  // global loads inside the loop they would tie the epilogue to the vmcnt of the patch prefetch)
  float* Ss = Ps + F_PFLOATS;
  if (tid < 128) {
    const int c = tid & 63;
    Ss[tid] = tid < 64 ? (a.scale ? a.scale[c] : 1.f) : (a.shift ? a.shift[c] : 0.f);
  }

  float v[2][16];
  int t = blockIdx.x;
  const int nc = CT ? CT : a.c;
  int inb = first_load_patch<CT>(a, t, tid, v);
  first_store_patch(Ps, tid, v, inb, nc);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  float vmax = 0.f;
  for (; t < total; t += gridDim.x) {
  const int tn = t + gridDim.x;
  if (tn < total) inb = first_load_patch<CT>(a, tn, tid, v);
  int sp = t;
  const int tx = sp % a.tiles_x;
  sp /= a.tiles_x;
  const int ty = sp % a.tiles_y;
  const int b = sp / a.tiles_y;
  const int y0 = ty * F_TH, x0 = tx * F_TW;

  // ---- contraction: wave w owns image rows 2 w, 2 w + 1 of the patch; pixel tile pt = (row pt >> 1, 16-pixel half pt & 1)
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int kb = 0; kb < 5; ++kb) {
    // this lane's k-group: q = 4 kb + g4 -> tap q >> 1 (taps >= 9 are zero filters: read tap 8's pixels), group q & 1
    const int q = 4 * kb + g4;
    const int tap = (q >> 1) < 9 ? (q >> 1) : 8;
    const int off = (tap / 3) * F_HW + (tap % 3);
    const int cg = q & 1;
    f16x8v ah[4], ax[4], al2[4];
#pragma unroll
    for (int pt = 0; pt < 4; ++pt) {
      const int P = (2 * wave + (pt >> 1)) * F_HW + 16 * (pt & 1) + l15 + off;
      const int sw = (P >> 2) & 1;
      const float* pp = Ps + P * 16;
      ah[pt] = *reinterpret_cast<const f16x8v*>(pp + (((2 * cg) ^ sw) << 2));
      const f16x8v al = *reinterpret_cast<const f16x8v*>(pp + (((2 * cg + 1) ^ sw) << 2));
      ax[pt] = ah[pt] * (_Float16)F_LO_INV;
      al2[pt] = al * (_Float16)F_LO_INV;
    }
    const float* wq = Ws + q * (2 * 64 * 4);
#pragma unroll
    for (int f0 = 0; f0 < 4; f0 += 2) {
      f16x8v bh[2], bl[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        bh[j] = *reinterpret_cast<const f16x8v*>(wq + ((f0 + j) * 16 + l15) * 4);
        bl[j] = *reinterpret_cast<const f16x8v*>(wq + 64 * 4 + ((f0 + j) * 16 + l15) * 4);
      }
#pragma unroll
      for (int pt = 0; pt < 4; ++pt)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[pt][f0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j], ah[pt], acc[pt][f0 + j], 0, 0, 0);
#pragma unroll
      for (int pt = 0; pt < 4; ++pt)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[pt][f0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[j], ax[pt], acc[pt][f0 + j], 0, 0, 0);
#pragma unroll
      for (int pt = 0; pt < 4; ++pt)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[pt][f0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j], al2[pt], acc[pt][f0 + j], 0, 0, 0);
    }
  }

  // the next tile's patch replaces this one BEFORE the epilogue: the wait for its loads is a vmcnt(0) (the compiler
  // counts conservatively around the inline asm of the S16 split), which here only covers the previous tile's stores,
  // a whole contraction old - placed after the epilogue it would wait for the stores just issued
  if (tn < total) {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();                       // every wave is done with this tile's patch
    asm volatile("" ::: "memory");
    first_store_patch(Ps, tid, v, inb, nc);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }

  // ---- epilogue: lane = pixel l15 of tile pt; tiles 2 u, 2 u + 1 give it channels 32 u + 8 g4 .. + 7 -------------------
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int c0 = 32 * u + 8 * g4;
    float sc[8], sh[8];
    {
      const f32x4 s0 = *reinterpret_cast<const f32x4*>(Ss + c0), s1 = *reinterpret_cast<const f32x4*>(Ss + c0 + 4);
      const f32x4 h0 = *reinterpret_cast<const f32x4*>(Ss + 64 + c0), h1 = *reinterpret_cast<const f32x4*>(Ss + 64 + c0 + 4);
#pragma unroll
      for (int k = 0; k < 4; ++k) sc[k] = s0[k], sc[4 + k] = s1[k], sh[k] = h0[k], sh[4 + k] = h1[k];
    }
#pragma unroll
    for (int pt = 0; pt < 4; ++pt) {
      const int y = y0 + 2 * wave + (pt >> 1), x = x0 + 16 * (pt & 1) + l15;
      float tv[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        tv[k] = acc[pt][2 * u + (k >> 2)][k & 3] * sc[k] + sh[k];
        if (a.act == AMMC_ACT_RELU) tv[k] = fmaxf(tv[k], 0.f);
      }
#pragma unroll
      for (int k = 0; k < 8; k += 2) vmax = fmaxf(vmax, fmaxf(fabsf(tv[k]), fabsf(tv[k + 1])));
      ammc_u4 hi, lo;
      ammc_s16_split8(tv, hi, lo);
      ammc_u4* yp = reinterpret_cast<ammc_u4*>(a.y + ((int64_t)b * a.y_bs + (int64_t)y * a.y_rs + (int64_t)x * a.y_ps) + c0);
      yp[0] = hi;
      yp[1] = lo;
    }
  }
  }
  if (a.overflow_flag && !(vmax <= 65504.f)) atomicOr(a.overflow_flag, 1);
}

// OIHW [64][cin][3][3] -> the filter image [q][hi | lo][tile-ordered filter][8 halfs], q = 2 tap + channel group
__global__ __launch_bounds__(256) void pack_first_conv_kernel(const float* __restrict__ w, int cin, float* __restrict__ out) {
  const int gid = blockIdx.x * 256 + threadIdx.x;                 // one (q, filter position) per thread
  if (gid >= F_NQ * 64) return;
  const int q = gid / 64, pos = gid % 64;
  const int t = pos >> 4, r = pos & 15;
  const int f = 32 * (t >> 1) + 8 * (r >> 2) + 4 * (t & 1) + (r & 3);
  const int tap = q >> 1, cg = q & 1;
  f16x8v hi, lo;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = 8 * cg + i;
    const float v = (tap < 9 && c < cin) ? w[((int64_t)f * cin + c) * 9 + tap] : 0.f;
    const _Float16 hv = (_Float16)v;
    hi[i] = hv;
    lo[i] = (_Float16)((v - (float)hv) * F_LO_SCALE);
  }
  *reinterpret_cast<f16x8v*>(out + ((q * 2 + 0) * 64 + pos) * 4) = hi;
  *reinterpret_cast<f16x8v*>(out + ((q * 2 + 1) * 64 + pos) * 4) = lo;
}

}  // namespace ammc_s16
using namespace ammc_s16;

extern "C" int ammc_first_conv_image_floats(void) { return F_WFLOATS; }

extern "C" int ammc_pack_first_conv_f32(const float* w_oihw, int32_t cout, int32_t cin, float* out, void* stream) {
  if (!w_oihw || !out) return AMMC_EINVAL;
  if (cout != 64 || cin <= 0 || cin > 16) return AMMC_EUNSUP;
  hipLaunchKernelGGL(pack_first_conv_kernel, dim3((F_NQ * 64 + 255) / 256), dim3(256), 0, (hipStream_t)stream, w_oihw, cin, out);
  return ammc_launch_status();
}

extern "C" int ammc_conv_first_s16(const float* x_nchw, int32_t batch, int32_t c, int32_t h, int32_t w,
                                   const float* w_image, const float* scale, const float* shift, int32_t act,
                                   float* y, int64_t y_bs, int64_t y_rs, int64_t y_ps, int32_t* overflow_flag, void* stream) {
  return ammc_conv_first_s16_bs(x_nchw, (int64_t)c * h * w, batch, c, h, w, w_image, scale, shift, act, y, y_bs, y_rs, y_ps,
                                overflow_flag, stream);
}

extern "C" int ammc_conv_first_s16_bs(const float* x_nchw, int64_t x_bs, int32_t batch, int32_t c, int32_t h, int32_t w,
                                      const float* w_image, const float* scale, const float* shift, int32_t act,
                                      float* y, int64_t y_bs, int64_t y_rs, int64_t y_ps, int32_t* overflow_flag,
                                      void* stream) {
  if (!x_nchw || !w_image || !y || batch <= 0 || c <= 0 || h <= 0 || w <= 0 || x_bs < 0) return AMMC_EINVAL;
  if (c > 16 || w % F_TW || h % F_TH) return AMMC_EUNSUP;
  if (act != AMMC_ACT_NONE && act != AMMC_ACT_RELU) return AMMC_EUNSUP;
  if (((uintptr_t)w_image & 15) || ((uintptr_t)y & 31) || ((y_bs | y_rs | y_ps) & 7)) return AMMC_EINVAL;
  if ((int64_t)batch * y_bs >= (1LL << 31)) return AMMC_EUNSUP;
  constexpr size_t lds = (size_t)(F_WFLOATS + F_PFLOATS + 128) * sizeof(float);
  static_assert(lds <= 80 * 1024, "two workgroups per CU");
  FirstArgs a;
  a.x = x_nchw, a.w = w_image, a.scale = scale, a.shift = shift, a.y = y, a.overflow_flag = overflow_flag;
  a.y_bs = y_bs, a.y_rs = y_rs, a.y_ps = y_ps, a.x_bs = x_bs;
  a.batch = batch, a.c = c, a.h = h, a.w_ = w, a.act = act;
  a.tiles_x = w / F_TW, a.tiles_y = h / F_TH;
  if ((int64_t)c * h * w >= (1LL << 31)) return AMMC_EUNSUP;     // (plane offsets inside one batch entry are 32-bit)
  const int total = batch * a.tiles_y * a.tiles_x;
  const dim3 grid(total < 512 ? total : 512);
#define FIRST_LAUNCH(CT)                                                                                              \
  {                                                                                                                   \
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_first_s16_kernel<CT>),                     \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                         \
    if (e != hipSuccess) return (int)e;                                                                               \
    hipLaunchKernelGGL(conv_first_s16_kernel<CT>, grid, dim3(256), lds, (hipStream_t)stream, a);                     \
  }
  if (c == 12) FIRST_LAUNCH(12) else if (c == 6) FIRST_LAUNCH(6) else FIRST_LAUNCH(0)
#undef FIRST_LAUNCH
  return ammc_launch_status();
}
