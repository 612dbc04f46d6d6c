// The output layer of a stream (`outc`: Conv2d(64 -> 3 | 2, 3x3, pad 1) + tanh, reference models/unet.py:62-69 and the
// decoders' last line, :981-1007) as a streaming kernel: S16 NHWC in, fp32 NCHW out, optionally with the squared error
// against a target frame (the PSNR numerator of the scoring tail).
//
// Why its own kernel: with 3 filters the layer is a read of its input (268 MB at batch 16, 256x256) and little else,
// and the halo-patch kernel (conv_tap_s16.hip) ran it at 2.4 TB/s (113 us): one tile per workgroup, each 45-KB patch
// fetched with nothing else in flight, a barrier and a filter slice per tap.
// Here one persistent 8-wave workgroup per CU walks over its tiles with THREE patch stages: while unit u (one
// 32-channel block of one 8 x 32 tile) is contracted, the patches of units u + 1 and u + 2 are in flight, the filters
// of all taps and blocks stay in LDS (only four filter rows exist: MFMA rows 4..15 alias them, 9 KB), so a unit is nine
// taps without any barrier, and the only synchronisation is one counted vmcnt + one barrier per unit.  Every load of
// the loop is an LDS-DMA (patches, and the target tile of the fused squared error): with register loads in the loop the
// compiler's own vmcnt(0) waits serialised the pipeline.
// Measured at batch 16 (same box, per launch): 113 -> 77 us.  Ablations: DMA loop alone 59 us (a bare DMA loop of this
// shape on an idle chip: 43 us = 6.2 TB/s, tools/micro/dma_stream.hip), contraction alone 56 us (LDS fragment reads
// at two waves per SIMD: 1.75 us per unit against 0.4 us of MFMAs), epilogue 9 us.
//
// Arithmetic: the S16 contraction of conv_tap_s16_kernel<.., MF = 1> with two accumulator sets (hi x hi and the cross
// terms, joined in the epilogue like the two-accumulator variants of that kernel).  All four 16-lane groups of an
// accumulator hold the same four filters; group g stores channel g.
#include "ammc_common.h"
#include <hip/hip_fp16.h>
#include <stdio.h>
#include <stdlib.h>

namespace ammc_s16 {

typedef _Float16 f16x8o __attribute__((ext_vector_type(8)));

constexpr float O_LO_INV = 1.f / 2048.f;
constexpr int O_TH = 8, O_TW = 32, O_HW = O_TW + 2, O_HP = (O_TH + 2) * O_HW;     // 340 halo pixels
constexpr int O_APIECES = O_HP * 8;                                              // 2720 16-byte pieces
constexpr int O_NT = 512;
constexpr int O_R = (O_APIECES + O_NT - 1) / O_NT;                               // 6 DMA rounds per patch
constexpr int O_ASTAGE = ((O_APIECES + 63) / 64) * 64 * 4;                       // floats: 43 KB (whole wave-instructions)
constexpr int O_TG = 3 * 256;                                                    // target tile: <= 3 channels x 8 x 32 floats
constexpr int O_NS = 3;                                                          // patch stages
constexpr int O_MAXCC = 2;                                                       // <= 64 input channels (LDS: 144 + 9 KB)
static_assert(O_R == 6, "OUTC_WAIT below knows 0, 2, 6 and 8 outstanding operations");

struct OutcArgs {
  AmmcConvDesc d;
  int tiles_x, tiles_y, ncc, kpad, total, nstore;
};

#define OUTC_WAIT(n)                                                              \
  switch (n) {                                                                    \
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;               \
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;               \
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;               \
    default: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;              \
  }

__global__ __launch_bounds__(O_NT, 1) void conv_outc_s16_kernel(OutcArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;
  float* Fs = smem + O_NS * O_ASTAGE;                  // [chunk = tap * ncc + cc][4 filter rows][32 floats]
  float* Tg = Fs + 9 * O_MAXCC * 128;                  // [2][channel][row][32]: target tiles (alternating buffers)
  float* Scr = Tg + 2 * O_TG;                          // 1 KB: where the DMA instructions beyond an image land
  float* Sq = Scr + 256;                               // [2][8] per-wave squared error of a tile (alternating buffers)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g4 = lane >> 4;
  const AmmcConvDesc& d = a.d;
  const int G = gridDim.x;
  const int first = ammc_xcd_remap(blockIdx.x, G);     // tiles first + k G: a round of G tiles gives every XCD a
  const int ntile = (a.total - first + G - 1) / G;     // contiguous run (shared halo columns stay in its L2)
  const int U = ntile * a.ncc;                         // units of this workgroup

  // ---- filters: rows 0..3 of every (tap, block) slice, swizzled like the filter stages of conv_tap_s16 (MF = 1) -----
  for (int i = tid; i < 9 * a.ncc * 32; i += O_NT) {
    const int chunk = i >> 5, row = (i >> 3) & 3, q = i & 7;
    int sl = q ^ row;
    sl ^= (sl >> 1) & 1;
    const f32x4 w = *reinterpret_cast<const f32x4*>(d.w + (int64_t)row * a.kpad + chunk * 32 + 4 * sl);
    *reinterpret_cast<f32x4*>(Fs + chunk * 128 + row * 32 + 4 * q) = w;
  }

  const int ns = a.nstore;
  const int64_t ycs = d.y_cs > 0 ? d.y_cs : 1;
  const bool score = d.sq_target != nullptr;

  float sc = 1.f, sh = 0.f;                              // scale / bias of channel g4
  if (g4 < ns) {
    if (d.scale) sc = d.scale[g4];
    if (d.shift) sh = d.shift[g4];
  }
  asm volatile("" : "+v"(sc), "+v"(sh));                 // (the wait for these two loads belongs here, not into the loop)

  // halo-patch DMA pieces of this thread: piece p = j * NT + tid -> halo pixel p >> 3, physical slot p & 7
  int a_off[O_R];
#pragma unroll
  for (int j = 0; j < O_R; ++j) {
    int p = j * O_NT + tid;
    p = p < O_APIECES ? p : O_APIECES - 1;
    const int hp = p >> 3;
    int ls = (p & 7) ^ (hp & 7);
    ls ^= (ls >> 1) & 1;
    const int hy = hp / O_HW;
    const int hx = hp - hy * O_HW;
    a_off[j] = (int)((int64_t)hy * d.x_rs + (int64_t)hx * d.x_ps) + 4 * ls;
  }
  // (every wave issues every round - the vmcnt counts below are wave uniform; rounds beyond the stage land in Scr)
#define OUTC_ISSUE(tile_, cc_, stage_)                                                                            \
  {                                                                                                               \
    int sp_ = (tile_);                                                                                            \
    const int tx_ = sp_ % a.tiles_x;                                                                              \
    sp_ /= a.tiles_x;                                                                                             \
    const int ty_ = sp_ % a.tiles_y;                                                                              \
    const int b_ = sp_ / a.tiles_y;                                                                               \
    const float* xp_ = d.x + ((int64_t)b_ * d.x_bs + (int64_t)(ty_ * O_TH) * d.x_rs + (int64_t)(tx_ * O_TW) * d.x_ps) + (cc_) * 32; \
    float* dst_ = As + (stage_) * O_ASTAGE + wave * 256;                                                          \
    _Pragma("unroll") for (int j = 0; j < O_R; ++j)                                                               \
      __builtin_amdgcn_global_load_lds(xp_ + a_off[j], (j * O_NT + wave * 64) * 4 < O_ASTAGE ? dst_ + j * O_NT * 4 : Scr, 16, 0, 0); \
  }
  // the target tile (NCHW fp32: ns channels x 8 rows x 32 pixels, element e = 256 c + 32 row + col): two 4-byte DMA
  // rounds of the workgroup
  const int tg_e0 = tid < ns * 256 ? tid : 0, tg_e1 = 512 + tid < ns * 256 ? 512 + tid : 0;
  const int64_t tg_o0 = (tg_e0 >> 8) * ycs + ((tg_e0 >> 5) & 7) * d.y_rs + (tg_e0 & 31) * d.y_ps;
  const int64_t tg_o1 = (tg_e1 >> 8) * ycs + ((tg_e1 >> 5) & 7) * d.y_rs + (tg_e1 & 31) * d.y_ps;
#define OUTC_ISSUE_TG(tile_, buf_)                                                                                \
  {                                                                                                               \
    int sp_ = (tile_);                                                                                            \
    const int tx_ = sp_ % a.tiles_x;                                                                              \
    sp_ /= a.tiles_x;                                                                                             \
    const int ty_ = sp_ % a.tiles_y;                                                                              \
    const int b_ = sp_ / a.tiles_y;                                                                               \
    const float* tp_ = d.sq_target + ((int64_t)b_ * d.y_bs + (int64_t)(ty_ * O_TH) * d.y_rs + (int64_t)(tx_ * O_TW) * d.y_ps); \
    __builtin_amdgcn_global_load_lds(tp_ + tg_o0, wave * 64 < ns * 256 ? Tg + (buf_) * O_TG + wave * 64 : Scr, 4, 0, 0);   \
    __builtin_amdgcn_global_load_lds(tp_ + tg_o1, 512 + wave * 64 < ns * 256 ? Tg + (buf_) * O_TG + 512 + wave * 64 : Scr, 4, 0, 0); \
  }

  // issue side: the next unit to fetch.  The target tile of a unit that ends a tile is requested one unit ahead, just
  // before the patch of the unit after it: the wait of that unit then covers it
  int i_tile = first, i_cc = 0, i_stage = 0, issued = 0;
#define OUTC_ISSUE_NEXT()                                         \
  {                                                               \
    OUTC_ISSUE(i_tile, i_cc, i_stage);                            \
    ++issued;                                                     \
    i_stage = i_stage + 1 == O_NS ? 0 : i_stage + 1;              \
    if (++i_cc == a.ncc) i_cc = 0, i_tile += G;                   \
  }
  if (score && a.ncc == 1 && U > 0) OUTC_ISSUE_TG(first, 0);       // unit 0 ends a tile
  if (issued < U) OUTC_ISSUE_NEXT();
  if (issued < U) OUTC_ISSUE_NEXT();

  // this lane: pixel l15 of the two 16-pixel tiles of image row `wave`; filter row l15 & 3; S16 group g4
  const int slot_hi = 2 * g4 + (g4 & 1), slot_lo = slot_hi ^ 1;
  const int frow = l15 & 3;
  const float* f_hi = Fs + frow * 32 + ((slot_hi ^ frow) << 2);
  const float* f_lo = Fs + frow * 32 + ((slot_lo ^ frow) << 2);
  const int hpb = wave * O_HW + l15;
  f32x4 acc[2], acx[2];
  acc[0] = acc[1] = acx[0] = acx[1] = f32x4{0.f, 0.f, 0.f, 0.f};
  int c_tile = first, c_cc = 0, c_stage = 0, c_par = 0;  // c_par: parity of the tile index (target / Sq buffers)
  bool pend = false;                                     // the previous unit ended a tile: its stores are in flight
  int pend_b = 0;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // filter image written

  for (int u = 0; u < U; ++u) {
    // unit u (and the target tile requested before unit u + 1's patch) has landed once at most these are outstanding:
    // the DMA rounds of unit u + 1 and, issued after them, the two stores of the previous unit's epilogue (VMEM
    // operations retire in order; anything else still in flight only makes this wait longer, never shorter)
    const int nwait = (u + 1 < U ? O_R : 0) + (pend ? 2 : 0);
    OUTC_WAIT(nwait);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (pend && score && tid == 0) {                     // one atomic per tile: the eight partial sums of the waves
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) s += Sq[(c_par ^ 1) * 8 + w];
      unsafeAtomicAdd(d.sq_acc + pend_b, s);
    }
    const bool last_cc = c_cc + 1 == a.ncc;
    int sp = c_tile;
    const int tx = sp % a.tiles_x;
    sp /= a.tiles_x;
    const int ty = sp % a.tiles_y;
    const int b = sp / a.tiles_y;
    // unit u + 1 ends a tile: its target (tile c_tile, or the next one when this unit ends a tile itself)
    if (score && u + 1 < U && (a.ncc == 1 || c_cc + 2 == a.ncc)) {
      if (a.ncc == 1) OUTC_ISSUE_TG(c_tile + G, c_par ^ 1) else OUTC_ISSUE_TG(c_tile, c_par);
    }
    if (issued < U) OUTC_ISSUE_NEXT();                   // stage (u + 2) % 3: its last readers passed the barrier above

    // ---- nine taps of this 32-channel block, no synchronisation.  Two accumulator sets (hi x hi, cross terms: the
    // 2^-11 is applied once, in the epilogue) = four independent MFMA chains per wave, and the fragments of tap t + 1
    // are read before the MFMAs of tap t: at two waves per SIMD nothing else hides the LDS latency (the first form,
    // reads -> wait -> MFMAs tap by tap on one accumulator set, spent 2.2 us per unit here - more than the DMA) ---------
    const float* Ac = As + c_stage * O_ASTAGE;
    f16x8o fbh[2], fbl[2], fah[2][2], fal[2][2];
#define OUTC_LOAD(tap_, s_)                                                                   \
  {                                                                                           \
    const int chunk_ = (tap_) * a.ncc + c_cc;                                                 \
    fbh[s_] = *reinterpret_cast<const f16x8o*>(f_hi + chunk_ * 128);                          \
    fbl[s_] = *reinterpret_cast<const f16x8o*>(f_lo + chunk_ * 128);                          \
    _Pragma("unroll") for (int pt = 0; pt < 2; ++pt) {                                        \
      const int hp_ = hpb + 16 * pt + ((tap_) / 3) * O_HW + ((tap_) % 3);                     \
      const float* ap_ = Ac + hp_ * 32;                                                       \
      const int sw_ = hp_ & 7;                                                                \
      fah[s_][pt] = *reinterpret_cast<const f16x8o*>(ap_ + ((slot_hi ^ sw_) << 2));           \
      fal[s_][pt] = *reinterpret_cast<const f16x8o*>(ap_ + ((slot_lo ^ sw_) << 2));           \
    }                                                                                         \
  }
    OUTC_LOAD(0, 0);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int cur = tap & 1;
      if (tap < 8) OUTC_LOAD(tap + 1, cur ^ 1);
      __builtin_amdgcn_sched_barrier(0);           // (left alone the scheduler undoes the prefetch to save registers)
#pragma unroll
      for (int pt = 0; pt < 2; ++pt) acc[pt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fbh[cur], fah[cur][pt], acc[pt], 0, 0, 0);
#pragma unroll
      for (int pt = 0; pt < 2; ++pt) acx[pt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fbl[cur], fah[cur][pt], acx[pt], 0, 0, 0);
#pragma unroll
      for (int pt = 0; pt < 2; ++pt) acx[pt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fbh[cur], fal[cur][pt], acx[pt], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
#undef OUTC_LOAD

    // ---- epilogue of a tile: 16-lane group g holds filters 0..3 like every other group and stores channel g ----------
    pend = last_cc;
    if (last_cc) {
      const int y = ty * O_TH + wave, x0 = tx * O_TW;
      const int ch = g4 < ns ? g4 : 0;
      float sq0 = 0.f;
#pragma unroll
      for (int pt = 0; pt < 2; ++pt) {
        const f32x4 sum = acc[pt] + acx[pt] * O_LO_INV;
        const float r = g4 == 0 ? sum[0] : (g4 == 1 ? sum[1] : (g4 == 2 ? sum[2] : sum[3]));
        float t = r * sc + sh;
        if (d.act == AMMC_ACT_RELU) t = t > 0.f ? t : 0.f;
        else if (d.act == AMMC_ACT_TANH) t = tanhf(t);
        if (g4 < ns) {
          d.y[(int64_t)b * d.y_bs + (int64_t)y * d.y_rs + (int64_t)(x0 + 16 * pt + l15) * d.y_ps + (int64_t)ch * ycs] = t;
          if (score) {
            const float df = 0.5f * (Tg[c_par * O_TG + ch * 256 + wave * 32 + 16 * pt + l15] - t);
            sq0 += df * df;
          }
        }
        acc[pt] = acx[pt] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      if (score) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sq0 += __shfl_xor(sq0, off);
        if (lane == 0) Sq[c_par * 8 + wave] = sq0;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      pend_b = b;
    }
    c_stage = c_stage + 1 == O_NS ? 0 : c_stage + 1;
    if (++c_cc == a.ncc) c_cc = 0, c_tile += G, c_par ^= 1;
  }
  if (pend && score) {                                   // the last tile's sum
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (tid == 0) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) s += Sq[(c_par ^ 1) * 8 + w];
      unsafeAtomicAdd(d.sq_acc + pend_b, s);
    }
  }
#undef OUTC_ISSUE
#undef OUTC_ISSUE_TG
#undef OUTC_ISSUE_NEXT
}

// Called by conv_tap_s16_try for the 32-filter fp32-output case.  Returns OUTC_SKIP when the descriptor is not this
// kernel's (the halo-patch kernel then takes it), else the launch status.
int conv_outc_s16_try(const AmmcConvDesc& d, int kpad, hipStream_t stream, char* label, int label_len) {
  constexpr int OUTC_SKIP = -12345;
  if (!(d.outc_stream ? d.outc_stream - 1 : ammc_opt_outc_stream())) return OUTC_SKIP;   // per call, else the process default
  const int ns = d.n_store > 0 ? d.n_store : d.n;
  if (d.ntaps != 9 || d.up != 1 || d.x_step > 1 || !d.y_f32 || d.n != 32 || ns > 4 || d.res || d.pool_y) return OUTC_SKIP;
  if (d.cin % 32 || d.cin / 32 > O_MAXCC || d.width % O_TW || d.height % O_TH) return OUTC_SKIP;
  if (d.sq_target && !d.sq_acc) return AMMC_EINVAL;
  OutcArgs a;
  a.d = d;
  a.tiles_x = d.width / O_TW;
  a.tiles_y = d.height / O_TH;
  a.ncc = d.cin / 32;
  a.kpad = kpad;
  a.total = d.batch * a.tiles_y * a.tiles_x;
  a.nstore = ns;
  if (label) {
    snprintf(label, label_len, "conv_outc_s16");
    return AMMC_OK;
  }
  constexpr size_t lds = (size_t)(O_NS * O_ASTAGE + 9 * O_MAXCC * 128 + 2 * O_TG + 256 + 16) * sizeof(float);
  static_assert(lds <= 160 * 1024, "LDS budget");
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_outc_s16_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  const int grid = a.total < 256 ? a.total : 256;
  hipLaunchKernelGGL(conv_outc_s16_kernel, dim3(grid), dim3(O_NT), lds, stream, a);
  return ammc_launch_status();
}

}  // namespace ammc_s16
