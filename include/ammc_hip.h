/*
 * ammc_hip.h - C ABI of libammc_hip.so: the MI355X (gfx950) kernels behind the
 * appearance-motion memory-consistency network's forward()/backward().
 *
 * The reference (NjuHaoZhang/AMMCNet_AAAI2021) has no native code: its hot
 * path is `twostream.forward` (Code/models/unet.py:981-1007) executed by
 * torch.nn modules on cuDNN/ATen.  Every entry below replaces the ATen/cuDNN
 * call(s) named in its comment; the Python host
 * (ammcnet_aaai2021_amd/unet.py) keeps the reference's nn.Module interface and
 * state_dict layout and binds these symbols with ctypes.
 *
 * Conventions
 *   - plain pointers and sizes only; device pointers unless stated;
 *   - every entry takes the hipStream_t to launch on (as void*), launches
 *     asynchronously, allocates nothing and never synchronises;
 *   - return 0 on success, a negative AMMC_E* code on bad arguments, or the
 *     positive hipError_t of a failed launch; never abort();
 *   - activations inside the path are fp32 NHWC ("pixel-major"); tensors that
 *     feed a 3x3 convolution carry a one-pixel zero halo:
 *     [B][H+2][W+2][C].  Strides below are in ELEMENTS.
 */
#ifndef AMMC_HIP_H
#define AMMC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AMMC_OK 0
#define AMMC_EINVAL (-1)   /* bad shape / alignment / null pointer   */
#define AMMC_EUNSUP (-2)   /* shape not supported by any kernel tile */

#define AMMC_ACT_NONE 0
#define AMMC_ACT_RELU 1
#define AMMC_ACT_TANH 2
#define AMMC_ACT_LRELU 3   /* nn.LeakyReLU(0.1), pix2pix_networks.py:606 */

/* library / device identification ---------------------------------------- */
int ammc_abi_version(void);                    /* bumps on any signature change */
/* "file=digest,file=digest,..." (sha256[:12]) of the source files the library was compiled from; profiles/ files carry
 * the same map and bench.py quotes a profiled figure of a kernel only while the kernel's file, ammc_common.h and this
 * header are unchanged ("" for a library built outside ammcnet_aaai2021_amd/build.py) */
const char* ammc_source_digests(void);
const char* ammc_build_info(void);             /* "gfx950 ..." */
const char* ammc_error_string(int code);
/* Dispatch options, for A/B measurements and for tests that must reach every kernel instance:
 *   "s16_mf"  MFMA shape of the halo-patch kernel (conv_tap_s16<..., MF>): -1 = the measured faster one per
 *             variant (default), 1 = v_mfma_f32_16x16x32_f16, 0 = v_mfma_f32_32x32x16_f16.  Initial value: AMMC_S16_MF.
 *   "outc_stream"  1 = the output layer (32-filter tile, fp32 NCHW output) on the streaming kernel conv_outc_s16
 *             (default), 0 = on the halo-patch kernel.  Initial value: AMMC_OUTC_STREAM.
 *   "memory_rt"  feature rows per workgroup of ammc_memory_topk_fwd_s16: 0 = by size (64 from 16384 rows up, k <= 2,
 *             m <= 2048; default), 1 = 32, 2 = 64.  Results are bit-identical.  Initial value: AMMC_MEMORY_RT.
 *   "memory_split"  ammc_memory_topk_fwd_f16: -1 / 0 = one fused launch (default), 1 = the contraction runs in chunks of
 *             whole rounds of workgroups on the caller's stream and the HBM-bound gather / commit of each chunk on a
 *             second, library-owned stream beside the next chunk's contraction (the caller's stream waits for it before
 *             the call's successors run; measured 1 % faster at 262144 rows).  Initial value: AMMC_MEMORY_SPLIT.
 * Returns AMMC_EUNSUP for an unknown key, AMMC_EINVAL for a value out of range.  These are PROCESS defaults
 * (not thread safe against concurrent launches); a caller that needs a per-call choice sets the descriptor fields
 * `s16_mf` / `outc_stream` instead, which take precedence and touch no global state. */
int ammc_set_option(const char* key, int32_t value);

/*
 * Implicit-GEMM convolution on the fp32 MFMA pipe (v_mfma_f32_32x32x2_f32):
 *   y[m, n] = act( scale[n] * sum_k A[m, k] * Wp[n, k] + shift[n] ) + res[m, n]
 * m runs over the B*H*W pixels, k = tap*Cin + c over `ntaps` (1 or 9) window
 * taps and Cin channels; A is gathered on the fly from the NHWC input.
 *
 * Replaces, by mode:
 *   ntaps=9            nn.Conv2d(3x3, pad 1, bias=False) + BatchNorm2d(eval) + ReLU
 *                      (unet.py:11-16, `double_conv`), the AMFT residual add
 *                      (unet.py:962-965) through `res`
 *   ntaps=1            nn.Conv2d 1x1 + bias, `enc`/`dec` (unet.py:321-330) and
 *                      `out += x` (unet.py:386) through `res`
 *   ntaps=9, n=32, n_store=cout, act=TANH, NCHW strides
 *                      `outc` + torch.tanh (unet.py:920, 998-1007): the 2-3 output channels
 *                      ride in a 32-wide MFMA column tile
 *   ntaps=4, x_step=2  gradient of that ConvTranspose2d w.r.t. its input (autograd of unet.py:51)
 *   ntaps=9 on dY with the flipped/transposed filter: gradient of the 3x3 conv w.r.t. its input
 *   ntaps=1, up=2      nn.ConvTranspose2d(C, C/2, 2, stride 2) + bias (unet.py:47,51):
 *                      N = 4*cgroup columns, column n = (dy*2+dx)*cgroup + co is
 *                      scattered to pixel (2y+dy, 2x+dx); writing into a channel
 *                      slice of the concat buffer replaces torch.cat (unet.py:57)
 */
typedef struct AmmcConvDesc {
  const float* x;        /* input, offset to tap (0,0) of pixel (0,0,0), channel 0        */
  const float* w;        /* packed weights [N][Kpad], Kpad = roundup(ntaps*Cin, 32)       */
  float* y;              /* output, offset to pixel (0,0,0) (+halo) and channel offset     */
  const float* scale;    /* [N] or NULL (=1)                                              */
  const float* shift;    /* [N] or NULL (=0)                                              */
  const float* res;      /* residual added after the activation, or NULL (S16 kernels: S16, or fp32 NHWC when y_f32) */
  int32_t batch, height, width;        /* pixel space of m                               */
  int32_t cin;           /* channels per tap: power of two >= 4                           */
  int32_t ntaps;         /* 9 (3x3, pad 1 via the halo), 4 (2x2, see x_step), 16 (4x4) or 1 */
  int32_t n;             /* GEMM N: 32, or a multiple of 64                               */
  int32_t up;            /* 1, or 2 for the ConvTranspose scatter                         */
  int32_t cgroup;        /* channels per (dy,dx) group when up=2 (multiple of 32); else n */
  int32_t act;           /* AMMC_ACT_*                                                    */
  int32_t n_store;       /* columns actually stored (0 = all n); lets a small-N layer pad N to 32 */
  int64_t x_bs, x_rs, x_ps;            /* input batch / row / pixel strides              */
  int64_t y_bs, y_rs, y_ps;            /* output strides (of the OUTPUT resolution)      */
  int64_t r_bs, r_rs, r_ps;            /* residual strides                               */
  int64_t y_cs;          /* output channel stride: 0/1 = NHWC; H*W (with y_ps=1, y_rs=W) = NCHW    */
  int32_t x_step;        /* 0/1; 2 = the input is at twice the resolution of m (with ntaps=4: the  */
  int32_t y_f32;         /* (2x2 stride-2 gather of the ConvTranspose dgrad).  y_f32: ammc_conv_gemm_s16 only, 1 = fp32 output */
  int32_t s16_mf;        /* ammc_conv_gemm_s16 only, per CALL (re-entrant): 0 = the process default (ammc_set_option "s16_mf"), 1 = v_mfma_f32_32x32x16_f16, */
  int32_t outc_stream;   /* 2 = v_mfma_f32_16x16x32_f16 forced.  outc_stream: 0 = process default, 1 = halo-patch kernel, 2 = streaming kernel */
  int32_t* overflow_flag; /* ammc_conv_gemm_s16 only, may be NULL: set to 1 when an S16 output exceeds the half range */
  float* splitk_ws;      /* ammc_conv_gemm_s16 only, may be NULL: fp32 workspace that lets small-M layers split K    */
  int64_t splitk_ws_floats; /* over workgroups ([ksplit][M][N] partial tiles + a finishing kernel)                   */
  const float* sq_target; /* fp32-output epilogues only (outc), may be NULL: a tensor laid out like y; the kernel adds   */
  float* sq_acc;          /* sum(((t+1)/2 - (y+1)/2)^2) of every stored element to sq_acc[sample] (fp32 atomics): the  */
                          /* per-sample squared error of `psnr_error` (utils/utils.py:141-148) without re-reading y     */
  float* pool_y;          /* ammc_conv_gemm_s16, S16 3x3 layers that its halo-patch kernel serves (Cin % 32 == 0, W % 32  */
  int64_t pool_bs, pool_rs, pool_ps; /* == 0, H % 8 == 0, >= 192 patches; no residual), may be NULL: also store the 2x2   */
                          /* max-pooled output (nn.MaxPool2d(2) of the `down` block that follows, unet.py:36) at half     */
                          /* resolution; AMMC_EUNSUP when the layer is not that kernel's                                  */
  float* stats;           /* ammc_conv_gemm_s16, fp32 outputs, may be NULL: per-channel sum and sum of squares of the STORED    */
                          /* values of every output patch, stats[patch][2][n] - the partial rows ammc_bn_finalize_f32 combines */
                          /* (training-mode BatchNorm2d statistics, models/unet.py:12,15, without re-reading the tensor);      */
                          /* rows = ammc_conv_gemm_s16_stats_rows(desc), AMMC_EUNSUP when that is 0                            */
  const float* bn_c;      /* with `stats`, may be NULL: the output is the gradient of a BatchNorm(+ReLU) unit's output and the  */
  int64_t bn_bs, bn_rs, bn_ps; /* statistics are those of that unit's BACKWARD (autograd through models/unet.py:12-16):         */
  const float *bn_mean, *bn_invstd, *bn_scale, *bn_shift; /* stats[patch][4][n] = sum g, sum g xhat, max |g|, max |xhat| with   */
  int32_t bn_relu;        /* g = y [bn_relu == 0 or bn_c scale + shift > 0], xhat = (bn_c - mean) invstd; bn_c = the unit's     */
  int32_t reserved0;      /* saved convolution output at the same pixels (own strides), scale / shift its folded BatchNorm:    */
                          /* the rows ammc_bn_bwd_finalize_f32 combines, without ammc_bn_bwd_reduce_bound_f32's pass            */
} AmmcConvDesc;

int ammc_conv_gemm_f32(const AmmcConvDesc* desc, void* stream);

/* nn.MaxPool2d(2) (unet.py:36) on NHWC; input strides explicit (the skip tensor lives in
 * the concat buffer), output strides explicit (halo-padded). h,w are OUTPUT sizes. */
int ammc_maxpool2x2_f32(const float* x, int64_t x_bs, int64_t x_rs, int64_t x_ps,
                        float* y, int64_t y_bs, int64_t y_rs, int64_t y_ps,
                        int32_t batch, int32_t h, int32_t w, int32_t c, void* stream);

/* module-boundary layout change: NCHW fp32 [B,C,H,W] -> NHWC with explicit strides
 * (y points at pixel (0,0), i.e. inside the halo; channels c..cp-1 are written as zero,
 * cp % 4 == 0).  Writing into a channel slice of a wider buffer is allowed. */
int ammc_nchw_to_nhwc_f32(const float* x, int32_t batch, int32_t c, int32_t h, int32_t w,
                          float* y, int64_t y_bs, int64_t y_rs, int64_t y_ps, int32_t cp, void* stream);
/* NHWC (strided) -> NCHW, for tensors handed back across the module boundary */
int ammc_nhwc_to_nchw_f32(const float* x, int64_t x_bs, int64_t x_rs, int64_t x_ps,
                          int32_t batch, int32_t c, int32_t h, int32_t w, float* y, void* stream);
/* zero the one-pixel border of a [B][H+2][W+2][C] buffer */
int ammc_zero_halo_f32(float* y, int32_t batch, int32_t h, int32_t w, int32_t c, void* stream);

/* weight pre-packing (run once per load_state_dict / optimizer step) ------ */
/* OIHW [cout][cin][kh][kw] (kh=kw=3 or 1) -> [cout][Kpad], k = (r*kw+s)*cin_p + c */
int ammc_pack_conv_weight_f32(const float* w_oihw, int32_t cout, int32_t cin, int32_t ksize,
                              int32_t cin_p, float* out, void* stream);
/* ConvTranspose2d IOHW [cin][co][2][2] -> [4*co][cin], row (dy*2+dx)*co + c_out */
int ammc_pack_convt_weight_f32(const float* w_iohw, int32_t cin, int32_t co, float* out, void* stream);
/* eval-mode BatchNorm2d folded to y = scale*x + shift (unet.py:12,15) */
int ammc_bn_fold_f32(const float* gamma, const float* beta, const float* mean, const float* var,
                     float eps, int32_t c, float* scale, float* shift, void* stream);
/* codebook [D][M] -> slot-major [M][D] plus |E_m|^2 (unet.py:287, 315-316) */
int ammc_pack_codebook_f32(const float* embed_dm, int32_t d, int32_t m, float* embed_md,
                           float* enorm, void* stream);

/*
 * Memory addressing, forward (`Quantize_topk.forward`, unet.py:282-297, 310-313):
 * squared-L2 distance of each of n feature vectors x[n][d] to the m slots,
 * the k nearest slots (nearest first), their rows gathered and concatenated
 * into q_topk[n][k*d], q_one[n][d] = x + (E[i1]-x), and per-block partial
 * sums of (E[i1]-x)^2 written to diff_partial[ammc_memory_topk_blocks(n)].
 * embed_dm is the module's own buffer [d][m]; embed_md / enorm come from
 * ammc_pack_codebook_f32.  d in {64,128,192,256}, k <= 4, q_one may be NULL.
 * The n x m distance matrix never reaches HBM.
 */
int ammc_memory_topk_blocks(int32_t n);
int ammc_memory_topk_fwd_f32(const float* x, const float* embed_dm, const float* embed_md,
                             const float* enorm, int32_t n, int32_t d, int32_t m, int32_t k,
                             int32_t* idx_topk, float* q_topk, float* q_one,
                             float* diff_partial, void* stream);
/* The same memory addressing with the distance GEMM on the fp16 MFMA pipe in fp32-EQUIVALENT arithmetic (features and
 * slots as (hi, lo) half pairs, three MFMAs per product, fp32 accumulation; norms, gather, commit distance from the fp32
 * data as in ammc_memory_topk_fwd_f32): the inference default for the model's embed_dim = 64 (d != 64: AMMC_EUNSUP, use
 * the fp32 entry).  e_s16 = ammc_pack_codebook_s16(embed [d][m]): [d/8][hi | lo][mpad][8] halfs (opaque to callers), mpad = m rounded up
 * to 32; enorm / embed_md from ammc_pack_codebook_f32; diff_partial has ammc_memory_topk_blocks(n) entries. */
int ammc_pack_codebook_s16(const float* embed_dm, int32_t d, int32_t m, void* e_s16, void* stream);
/* ... and with a range verdict: *range_flag (device int32, zeroed by the caller, sticky) is raised when an entry of the
 * codebook does not fit the hi half (|v| > 65504 or not finite).  The reference's own EMA update (models/unet.py:298-309)
 * produces such entries from its initial state (:277-280: cluster_size = 0): a slot no row has hit yet sits at
 * embed = 0.99^t e0 / ~1e-5 ~ 1e5 x N(0, 1) for the first ~150 training steps.  A flagged codebook must be looked up with
 * ammc_memory_topk_fwd_f32 (an inf / NaN slot of the S16 image corrupts the ranking). */
int ammc_pack_codebook_s16_guarded(const float* embed_dm, int32_t d, int32_t m, void* e_s16, int32_t* range_flag, void* stream);
int ammc_memory_topk_fwd_s16(const float* x, const void* e_s16, const float* embed_md, const float* enorm, int32_t n,
                             int32_t d, int32_t m, int32_t k, int32_t* idx_topk, float* q_topk, float* q_one,
                             float* diff_partial, void* stream);

/* The whole memory block of the inference path as ONE launch (`enc_quan_dec_res_topk.forward`, models/unet.py:318-331,
 * 379-387): enc 1x1 (c -> d, + enc_b) -> distances + top-k (ammc_memory_topk_fwd_s16's arithmetic) -> gather -> dec 1x1
 * (k d -> c, + dec_b) -> += x, for 64-pixel tiles; z, the n x m distances and the gathered rows never leave the CU.
 * x / y: S16 activations [batch][h][w][c] (pointer to interior pixel 0, strides in floats); enc_w [d][c], dec_w [c][k d]:
 * ammc_pack_conv_weight_f32 (ksize 1) + ammc_split_rows_f32; e_s16 / embed_md / enorm as for ammc_memory_topk_fwd_s16.
 * Outputs: y, idx_topk [n][k], q_one [n][d] (may be NULL), q_topk [n][k d] fp32 (may be NULL), diff[0] = commit
 * distance (the LAST workgroup sums diff_partial[ammc_memory_topk_blocks(n)] in ammc_sum_partials_f32's order; *counter
 * is a device int32 the caller zeroes ONCE - the kernel leaves it at zero), *overflow_flag raised like the S16 epilogues
 * and ammc_split_rows_guarded_f32 do.  Bit-identical to the five-launch chain it replaces (same k order and expressions).
 * dec_wf is dec_w in FRAGMENT-major order (ammc_pack_frag_rows_s16 below).
 * Shapes: d = 64, k = 2, c = 512, m <= 2048 (the shipped block; otherwise AMMC_EUNSUP: use the chain). */
int ammc_memory_block_s16(const float* x, int64_t x_bs, int64_t x_rs, int64_t x_ps, float* y, int64_t y_bs, int64_t y_rs,
                          int64_t y_ps, int32_t batch, int32_t h, int32_t w, int32_t c, const float* enc_w, const float* enc_b,
                          const void* e_s16, const float* embed_md, const float* enorm, int32_t d, int32_t m, int32_t k,
                          const float* dec_wf, const float* dec_b, int32_t* idx_topk, float* q_topk, float* q_one,
                          float* diff_partial, float* diff, int32_t* counter, int32_t* overflow_flag, void* stream);
/* S16 filter rows [n][k] (ammc_pack_conv_weight_f32 + ammc_split_rows_f32; n % 32 == 0, k % 8 == 0) -> fragment-major
 * [k/8][hi | lo][n][8 halfs], the rows of every 32-row tile in MFMA order: what a lane of ammc_memory_block_s16's dec
 * phase reads is then 16 bytes of a 512-byte run shared with its 31 neighbours.  Same size in bytes. */
int ammc_pack_frag_rows_s16(const float* w_s16, int32_t n, int32_t k, float* out, void* stream);

/* diff = sum(partials) / count, fixed order (deterministic) */
int ammc_sum_partials_f32(const float* partial, int32_t nparts, float inv_count, float* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * "S16": fp32-equivalent arithmetic on the fp16 MFMA pipe.  A value v is the pair of halfs
 * hi = half(v), lo = half((v-hi)*2^11); tensors are NHWC with channels in groups of 8, each group
 * 32 bytes [8 hi | 8 lo] (4 B/element: strides in ELEMENTS are those of the fp32 layout).
 * ammc_conv_gemm_s16 takes the same descriptor as ammc_conv_gemm_f32 with x, w and res in S16
 * and computes a*b as hi*hi + (hi*lo + lo*hi)*2^-11 with three v_mfma_f32_32x32x16_f16 and fp32
 * accumulation; y is S16 unless y_f32 = 1 (then n_store / y_cs / TANH and an fp32 residual are available; up = 2
 * only without them).  Replaces the same reference calls as ammc_conv_gemm_f32.
 * ---------------------------------------------------------------------------------------- */
int ammc_conv_gemm_s16(const AmmcConvDesc* desc, void* stream);
/* `up.forward` of the decoder in inference (models/unet.py:50-59) as one launch: y = act(scale * conv3x3(cat([skip,
 * ConvTranspose2d(x2, k 2, s 2) + b])) + shift).  desc describes the 3x3 conv over the SKIP half only: x = the skip
 * tensor (S16, halo corner), cin = c skip channels, w = the S16 filter of the whole conv ([n][9 * 2c], k = tap * 2c +
 * ch: ammc_pack_conv_weight_f32 + ammc_split_rows_f32), scale, act, y (S16); desc->shift is ignored.  The other half
 * comes from x2 (S16, half resolution, halo 1, up_cin = 2c channels: what the transposed conv reads) through the
 * COMPOSED filters of ammc_pack_up_conv_f32 (the transposed conv folded into the 3x3 taps per output-pixel parity,
 * summed in double: no intermediate tensor; then ammc_split_rows_f32) and its border-class shifts shift9[3][3][n].
 * Needs W % 32 == 0, H % 8 == 0, c % 32 == 0, n = 64 or a multiple of 128; AMMC_EUNSUP otherwise (callers then run
 * the transposed conv and the 3x3 conv separately). */
int ammc_pack_up_conv_f32(const float* w3_oihw, const float* wt_iohw, const float* bt, const float* scale,
                          const float* shift, int32_t n, int32_t c, float* w2_out, float* shift9_out, void* stream);
int ammc_conv_up_s16(const AmmcConvDesc* desc, const float* up_x, int64_t up_bs, int64_t up_rs, int64_t up_ps,
                     int32_t up_cin, const float* up_w, const float* shift9, void* stream);
/* The first layer of a stream in inference (`inconv`'s first conv + BatchNorm(eval) + ReLU, models/unet.py:11-13, 23-30)
 * straight from the module-boundary tensor: x is NCHW fp32 [batch][c <= 16][h][w] (zero padding applied by the kernel),
 * y the S16 NHWC activation (64 channels, pixel (0,0), strides in elements).  Replaces ammc_nchw_to_s16_f32 +
 * ammc_conv_gemm_s16 for that layer.  w_image = ammc_pack_first_conv_f32(OIHW [64][c][3][3]) (ammc_first_conv_image_floats()
 * floats).  Needs w % 32 == 0, h % 8 == 0; AMMC_EUNSUP otherwise. */
int ammc_first_conv_image_floats(void);
int ammc_pack_first_conv_f32(const float* w_oihw, int32_t cout, int32_t cin, float* out, void* stream);
int ammc_conv_first_s16(const float* x_nchw, int32_t batch, int32_t c, int32_t h, int32_t w, const float* w_image,
                        const float* scale, const float* shift, int32_t act, float* y, int64_t y_bs, int64_t y_rs,
                        int64_t y_ps, int32_t* overflow_flag, void* stream);
/* The same with an explicit batch stride of x (elements; c * h * w = contiguous).  The evaluation loop's clips are
 * OVERLAPPING windows of one resident sub-video - clip b = frames [s + b, s + b + 4) of a [T][3][h][w] tensor
 * (test_helper.py:433-438: `view(B, t * c, H, W)` of consecutive frames) - so x = &frames[s], x_bs = 3 * h * w runs a batch
 * of 16 clips without first gathering 16 x 12 planes into a contiguous tensor (harness.score_batch_device). */
int ammc_conv_first_s16_bs(const float* x_nchw, int64_t x_bs, int32_t batch, int32_t c, int32_t h, int32_t w,
                           const float* w_image, const float* scale, const float* shift, int32_t act, float* y,
                           int64_t y_bs, int64_t y_rs, int64_t y_ps, int32_t* overflow_flag, void* stream);
/* Which kernel ammc_conv_gemm_s16 launches for this descriptor, as the NUL-terminated name rocprofv3 reports for it
 * (e.g. "conv_tap_s16<4, 1, 2, 4, 1>", "conv_gemm_s16<128x128>", "...+splitk4"), without launching anything: the same
 * argument checks and the same dispatch code run with the launch replaced by the label.  Tests pin kernel coverage on
 * it and bench.py labels its per-kernel timings with it. */
int ammc_conv_gemm_s16_variant(const AmmcConvDesc* desc, char* out, int32_t out_len);
/* Rows of `desc->stats` the kernel ammc_conv_gemm_s16 would launch for this descriptor writes (one per 8 x 32 output
 * patch), or 0 when that kernel has no statistics epilogue (the caller then runs ammc_bn_stats_f32 on the output as
 * before).  `desc->stats` itself is ignored here.  Same argument checks and dispatch code as the launch. */
int ammc_conv_gemm_s16_stats_rows(const AmmcConvDesc* desc);
/* fp32 -> S16, count elements (multiple of 8): packed filters, gathered codebook rows */
int ammc_split_rows_f32(const float* src, int64_t count, float* dst, void* stream);
/* ... and with a range verdict: *range_flag (device int32, sticky; the same flag the S16 epilogues raise through
 * AmmcConvDesc.overflow_flag) is raised when a value does not fit the hi half (|v| > 65504 or not finite): the S16 image
 * would hold inf there.  range_flag may be NULL (= ammc_split_rows_f32). */
int ammc_split_rows_guarded_f32(const float* src, int64_t count, float* dst, int32_t* range_flag, void* stream);
/* Gradient tensors as S16 operands (what autograd derives for the 3x3 convs, train_helper.py:337-339): the largest
 * |v| of the tensor as its fp32 bit pattern (atomicMax into one of the 256 slots out_bits[0..256), zeroed by the
 * caller; the consumer takes the maximum of the slots), then the split of
 * v * 2^k with k chosen so that the maximum lands at 2^10; inv_scale[0..n) = 2^-k is the epilogue scale of the
 * convolution that consumes the tensor. */
int ammc_absmax_bits_f32(const float* src, int64_t count, int32_t* out_bits, void* stream);
int ammc_split_rows_scaled_f32(const float* src, int64_t count, float* dst, const int32_t* amax_bits, float* inv_scale,
                               int32_t n, void* stream);
/* module-boundary layout: NCHW fp32 -> S16 NHWC (cp % 8 == 0) and back */
int ammc_nchw_to_s16_f32(const float* x, int32_t batch, int32_t c, int32_t h, int32_t w, float* y,
                         int64_t y_bs, int64_t y_rs, int64_t y_ps, int32_t cp, void* stream);
int ammc_s16_to_nchw_f32(const float* x, int64_t x_bs, int64_t x_rs, int64_t x_ps, int32_t batch, int32_t c,
                         int32_t h, int32_t w, float* y, void* stream);
/* nn.MaxPool2d(2) on S16 tensors */
int ammc_maxpool2x2_s16(const float* x, int64_t x_bs, int64_t x_rs, int64_t x_ps, float* y, int64_t y_bs,
                        int64_t y_rs, int64_t y_ps, int32_t batch, int32_t h, int32_t w, int32_t c, void* stream);

/* ... that also records, for every pooled element, which of the four window positions (row-major, the first maximum)
 * it came from: idx[batch][h][w][c], a byte each (8-byte aligned) - what ammc_maxpool2x2_bwd_idx_f32 routes gradients by */
int ammc_maxpool2x2_s16_idx(const float* x, int64_t x_bs, int64_t x_rs, int64_t x_ps, float* y, int64_t y_bs,
                            int64_t y_rs, int64_t y_ps, uint8_t* idx, int32_t batch, int32_t h, int32_t w, int32_t c,
                            void* stream);

/* fp16-operand form of the memory addressing for large memories (BASELINE.json config 5: 8192
 * slots x 512-d): distance GEMM on v_mfma_f32_32x32x16_f16 with fp32 accumulation, everything
 * else as ammc_memory_topk_fwd_f32 (gather / q_one / commit distance from the fp32 codebook).
 * Not the parity path: the ranking sees fp16-rounded operands.  d in {128,256,384,512}, k <= 4.
 * e_kblk_f16: [d/8][roundup(m,32)][8] halfs from ammc_pack_codebook_f16 (which also returns
 * |half(E_s)|^2); diff_partial has ammc_memory_topk_f16_blocks(n) entries. */
int ammc_pack_codebook_f16(const float* embed_dm, int32_t d, int32_t m, void* e_kblk_f16, float* enorm16,
                           void* stream);
int ammc_memory_topk_f16_blocks(int32_t n);
int ammc_memory_topk_fwd_f16(const float* x, const void* e_kblk_f16, const float* embed_md, const float* enorm16,
                             int32_t n, int32_t d, int32_t m, int32_t k, int32_t* idx_topk, float* q_topk,
                             float* q_one, float* diff_partial, void* stream);

/* The same operator with the feature ROWS resident in registers and the codebook streamed through LDS (round 5; what
 * config 5 runs): a wave keeps 3 x 32 rows x d features as fp16 MFMA operands in its VGPRs for a whole sweep of the
 * codebook, the codebook goes L2 -> LDS once per workgroup (tile images of ammc_pack_codebook_f16_tiles:
 * ammc_codebook_f16_tiles_bytes(d, m) bytes, 16-byte aligned) instead of L2 -> registers once per 128 rows.  Same
 * outputs and arithmetic contract as ammc_memory_topk_fwd_f16 (ranking from fp16-rounded operands with fp32
 * accumulation - key x.E_s - |E_s|^2 / 2; gather / q_one / commit from the fp32 codebook and features).  Exact ties
 * (k = 2, d >= 288: the packed-key form): two returned slots whose keys tie come in slot order; WHICH of several exactly
 * tied candidates takes the last place of the top-k is unspecified (other shapes: ties to the lower slot throughout);
 * q_one may be NULL; diff_partial has ammc_memory_topk_f16r_blocks(n) = ceil(n / 32) entries, each written
 * once (no atomics: deterministic).  d in {128,256,384,512}, k <= 4; pointers 16-byte aligned. */
int64_t ammc_codebook_f16_tiles_bytes(int32_t d, int32_t m);
int ammc_pack_codebook_f16_tiles(const float* embed_dm, int32_t d, int32_t m, void* tiles, void* stream);
int ammc_memory_topk_f16r_blocks(int32_t n);
int ammc_memory_topk_fwd_f16r(const float* x, const void* tiles, const float* embed_md, int32_t n, int32_t d, int32_t m,
                              int32_t k, int32_t* idx_topk, float* q_topk, float* q_one, float* diff_partial,
                              void* stream);

/* ------------------------------------------------------------------------------------------
 * Training mode (autograd of the same path; the reference derives these with torch.autograd)
 * ---------------------------------------------------------------------------------------- */

/* Weight gradient dWp[n][k] += sum_m G[m][n] * A[m, k], packed layout of ammc_conv_gemm_f32's
 * weights (the caller zeroes dw; partial tiles are combined with fp32 atomics).
 *   3x3 conv:      g = gradient of the raw conv output (pixel (0,0)), a = conv input (tap (0,0))
 *   1x1 conv:      ntaps 1
 *   ConvTranspose: g = layer input, a = output gradient at 2x resolution (a_step 2, ntaps 4),
 *                  rows = input channels, k = (dy*2+dx)*co + c_out
 * zeros: >= 128 floats of zeros (source of out-of-range pixels).  n % 32 == 0. */
typedef struct AmmcWgradDesc {
  const float* g;
  const float* a;
  float* dw;
  const float* zeros;
  int32_t batch, height, width;     /* pixel space of m (resolution of g)                     */
  int32_t n;                        /* rows of dWp                                            */
  int32_t cin;                      /* channels per tap of the a side                         */
  int32_t ntaps;                    /* 9, 4 or 1                                              */
  int32_t a_step;                   /* 0/1, or 2                                              */
  int32_t reserved;
  int64_t g_bs, g_rs, g_ps;
  int64_t a_bs, a_rs, a_ps;
} AmmcWgradDesc;
int ammc_conv_wgrad_f32(const AmmcWgradDesc* desc, void* stream);
/* the same with S16 operands (g, a: the S16 twins of the fp32 tensors, same strides): fp16 MFMA with transposed LDS
 * fragment reads; g_inv_scale (device, may be NULL): the power-of-two that ammc_split_rows_scaled_f32 left in g (or in
 * a: it is one scalar on the result), divided out.  ntaps 9 (the stride-1 3x3 layers: halo-patch kernels), 16 (4x4,
 * a_step 1 | 2: PixelDiscriminator) or 4 (2x2, a_step 2: ConvTranspose); cin a power of two >= 8, n % 32 == 0. */
int ammc_conv_wgrad_s16(const AmmcWgradDesc* desc, const float* g_inv_scale, void* stream);
/* The 3x3 form with the split partials SUMMED INTO THE PARAMETER GRADIENT by a second kernel (round 5): every workgroup of
 * the halo-patch weight-gradient kernel stores its tile of the packed gradient into the slab of its patch split
 * (slabs: [splits][n][kpad] floats of caller workspace, plain stores) and a reduce kernel writes dw_oihw [cout][cin][3][3]
 * (cout <= n, cin <= desc->cin: the unpadded channel counts) - replaces ammc_conv_wgrad_s16's fp32 atomics into a zeroed
 * packed buffer + ammc_unpack_conv_wgrad_f32; fixed summation order, so deterministic.  desc->dw is not used.
 * ammc_conv_wgrad_s16_slab_floats: the workspace this descriptor needs, or 0 when its kernel has no slab form (callers
 * then use ammc_conv_wgrad_s16). */
int64_t ammc_conv_wgrad_s16_slab_floats(const AmmcWgradDesc* desc);
int ammc_conv_wgrad_s16_slabs(const AmmcWgradDesc* desc, const float* g_inv_scale, float* slabs, int64_t slab_floats,
                              float* dw_oihw, int32_t cout, int32_t cin, void* stream);
/* packed gradient -> the module's parameter layout */
int ammc_unpack_conv_wgrad_f32(const float* packed, int32_t cout, int32_t cin, int32_t ksize, int32_t cin_p,
                               float* out_oihw, void* stream);
int ammc_unpack_convt_wgrad_f32(const float* packed, int32_t cin, int32_t co, float* out_iohw, void* stream);
/* filters of the input-gradient convolutions (run through ammc_conv_gemm_f32):
 * 3x3: [rows>=cin][Kpad], k = tap*cout_p + n, value W[n][c][2-r][2-s];  1x1: transpose, zero padded */
int ammc_pack_conv_dgrad_weight_f32(const float* w_oihw, int32_t cout, int32_t cin, int32_t cout_p, int32_t rows,
                                    float* out, void* stream);
int ammc_transpose_pad_f32(const float* w, int32_t rows, int32_t cols, int32_t rows_p, float* out, void* stream);

/* All 3x3 filters of a training step packed in ONE launch, straight to the S16 images the split-fp16 kernels read
 * (= ammc_pack_conv_weight_f32 / ammc_pack_conv_dgrad_weight_f32 followed by ammc_split_rows_f32, per layer).  items_dev: a
 * device array of n_items records of ammc_pack_filters_item_bytes() bytes each:
 *   { const float* w_oihw; float* out16; int32 cout, cin, inner_p, kpad, kind, rows; int64 group_end; }
 * kind 0 = forward filter [rows = cout][kpad], k = tap * inner_p + c (inner_p = cin_p); kind 1 = input-gradient filter
 * [rows >= cin][kpad], k = tap * inner_p + n (inner_p = cout_p); group_end = running total of rows * kpad / 8. */
int ammc_pack_filters_item_bytes(void);
int ammc_pack_filters_s16(const void* items_dev, int32_t n_items, int64_t total_groups, void* stream);

/* PixelDiscriminator (Code/models/pix2pix_networks.py:580-631): Conv2d(k 4, padding 2, stride 2|1, bias) + LeakyReLU(0.1).
 * Forward and weight gradient run through ammc_conv_gemm_f32 / ammc_conv_wgrad_f32 with ntaps 16 (x_step / a_step =
 * the stride, x / a = the corner of a 2-pixel halo).  Input gradient: stride 1 = one 16-tap conv with the flipped
 * filter; stride 2 = four 2x2-tap convs over the output gradient, one per input-pixel parity (py, px), written
 * through doubled y strides.  This packs those filters: stride 1 -> [rows][16*cout_p], stride 2 ->
 * [4][rows][4*cout_p] (phase = py*2+px). */
int ammc_pack_conv4_dgrad_weight_f32(const float* w_oihw, int32_t cout, int32_t cin, int32_t cout_p, int32_t rows,
                                     int32_t stride, int32_t pad /* of the convolution: 2 (D) or 1 */, float* out,
                                     void* stream);

/* FlowNet2-SD forward (Code/models/flownet2/models.py:15-59): the pieces that are not convolutions.  Its Conv2d(k 3,
 * stride 1|2) + LeakyReLU(0.1) layers are ammc_conv_gemm_f32 (ntaps 9, x_step = stride, AMMC_ACT_LRELU); its
 * ConvTranspose2d(k 4, s 2, p 1) layers are the four 2x2-tap parity convolutions of ammc_pack_conv4_dgrad_weight_f32
 * (pad 1); layers whose input is a concatenation run once per part and accumulate through `res`. */
/* scratch: ammc_flownet_prep_scratch_doubles(batch) doubles of device memory (per (sample, colour) slice sums of the mean:
 * summed by 32 workgroups each, combined in a fixed order) */
int ammc_flownet_prep_scratch_doubles(int32_t batch);
int ammc_flownet_prep_f32(const float* in /* [B][3][2][H][W], 0..rgb_max */, int32_t batch, int32_t h, int32_t w,
                          float* y /* NHWC, 8 channels */, int64_t y_bs, int64_t y_rs, int64_t y_ps, float rgb_max,
                          double* scratch, void* stream);
int ammc_lrelu_f32(float* y, int64_t y_bs, int64_t y_rs, int64_t y_ps, int32_t batch, int32_t h, int32_t w, int32_t c,
                   float slope, void* stream);
/* the same LeakyReLU in place on an S16 activation (c % 8 == 0): the split-fp16 form of the FlowNet2-SD forward */
int ammc_lrelu_s16(float* y, int64_t y_bs, int64_t y_rs, int64_t y_ps, int32_t batch, int32_t h, int32_t w, int32_t c,
                   float slope, void* stream);
int ammc_upsample4_bilinear_f32(const float* x, int64_t x_bs, int64_t x_rs, int64_t x_ps, int32_t batch, int32_t h,
                                int32_t w, int32_t c, float premul, float* out /* NCHW [B][c][4h][4w] */, void* stream);
/* g *= (y > 0 ? 1 : slope) in place: autograd of nn.LeakyReLU evaluated on the layer output */
int ammc_lrelu_bwd_f32(const float* y, int64_t y_bs, int64_t y_rs, int64_t y_ps, float* g, int64_t g_bs, int64_t g_rs,
                       int64_t g_ps, int32_t batch, int32_t h, int32_t w, int32_t c, float slope, void* stream);

/* Device half of the input pipeline (Code/dataset/two_stream_dataset.py:72-99, 503-506): per frame, once.
 * frames: uint8 [n][h][w][3] (RGB; bgr != 0: as decoded by cv2 / TurboJPEG, the cvtColor of :76 is folded in) ->
 *   cv2.resize INTER_LINEAR (8-bit fixed point) -> /255 -> (x - 0.5) / 0.5 -> float32 [n][3][oh][ow].
 * flows: float32 [n][h][w][2] (.flo payload, flowlib.py:589-611) -> cv2.resize INTER_LINEAR (float) ->
 *   c0 = u / oh, c1 = c0 / ow (the loader derives channel 1 from the scaled channel 0, :94-95) -> [n][2][oh][ow]. */
int ammc_frames_u8_to_f32(const uint8_t* src, int32_t n, int32_t h, int32_t w, float* dst, int32_t oh, int32_t ow,
                          int32_t bgr, void* stream);
int ammc_flows_to_f32(const float* src, int32_t n, int32_t h, int32_t w, float* dst, int32_t oh, int32_t ow,
                      void* stream);

/* nn.BatchNorm2d in training mode (unet.py:12,15).  Per-channel reductions write
 * partial[ammc_chan_reduce_blocks(B*H*W)][Q][C]; the finalizers combine them in fp64, fixed order. */
int ammc_chan_reduce_blocks(int32_t pixels);
int ammc_bn_stats_f32(const float* x, int64_t x_bs, int64_t x_rs, int64_t x_ps, int32_t batch, int32_t h, int32_t w,
                      int32_t c, float* partial /* Q=2: sum, sum of squares */, void* stream);
int ammc_bn_finalize_f32(const float* partial, int32_t nblocks, int32_t c, float count, const float* gamma,
                         const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                         float* mean, float* invstd, float* scale, float* shift, void* stream);
/* y = act(x*scale + shift) + res on interior pixels (BN apply + ReLU + AMFT residual) */
int ammc_scale_shift_act_f32(const float* x, int64_t x_bs, int64_t x_rs, int64_t x_ps, const float* scale,
                             const float* shift, const float* res, int64_t r_bs, int64_t r_rs, int64_t r_ps,
                             float* y, int64_t y_bs, int64_t y_rs, int64_t y_ps, int32_t relu, int32_t batch,
                             int32_t h, int32_t w, int32_t c, void* stream);
/* The same apply pass with the S16 image of y as its output (same strides; the fp32 y32 is optional): tensors that only
 * the split-fp16 convolutions read - the middle activation of a double_conv in training - never exist in fp32. */
int ammc_scale_shift_act_s16_f32(const float* x, int64_t x_bs, int64_t x_rs, int64_t x_ps, const float* scale,
                                 const float* shift, const float* res, int64_t r_bs, int64_t r_rs, int64_t r_ps,
                                 float* y32 /* may be NULL */, float* y16, int64_t y_bs, int64_t y_rs, int64_t y_ps,
                                 int32_t relu, int32_t batch, int32_t h, int32_t w, int32_t c, void* stream);
/* ... of a tensor that nn.MaxPool2d(2) follows (the second unit of `inconv` / `down`, models/unet.py:23-37): y as above, plus
 * its pooled S16 image pool16 (h/2 x w/2, own strides) and the window positions idx[batch][h/2][w/2][c] exactly as
 * ammc_maxpool2x2_s16_idx would produce them from y16 - in the same pass.  h, w even, c/8 a power of two <= 256; else
 * AMMC_EUNSUP (run the two passes). */
int ammc_scale_shift_act_s16_pool_f32(const float* x, int64_t x_bs, int64_t x_rs, int64_t x_ps, const float* scale,
                                      const float* shift, float* y32 /* may be NULL */, float* y16, int64_t y_bs, int64_t y_rs,
                                      int64_t y_ps, float* pool16, int64_t p_bs, int64_t p_rs, int64_t p_ps, uint8_t* idx,
                                      int32_t relu, int32_t batch, int32_t h, int32_t w, int32_t c, void* stream);
/* BN (+ReLU) backward: partial Q=2: sum g, sum g*xhat with g = dy*[c*scale+shift > 0]; then
 * dc = scale*(g - sums[0]/M - xhat*sums[1]/M); sums[0] = dbeta, sums[1] = dgamma.
 * NOTE: the `gamma`/`beta` arguments take the FOLDED scale (gamma*invstd) and shift produced by
 * ammc_bn_finalize_f32, so that the ReLU mask is evaluated exactly as in the forward. */
int ammc_bn_bwd_reduce_f32(const float* c_raw, int64_t c_bs, int64_t c_rs, int64_t c_ps, const float* dy,
                           int64_t d_bs, int64_t d_rs, int64_t d_ps, const float* mean, const float* invstd,
                           const float* gamma, const float* beta, int32_t relu, int32_t batch, int32_t h, int32_t w,
                           int32_t c, float* partial, void* stream);
int ammc_bn_bwd_apply_f32(const float* c_raw, int64_t c_bs, int64_t c_rs, int64_t c_ps, const float* dy,
                          int64_t d_bs, int64_t d_rs, int64_t d_ps, const float* mean, const float* invstd,
                          const float* gamma, const float* beta, const float* sums, int32_t relu, float* dc,
                          int64_t o_bs, int64_t o_rs, int64_t o_ps, int32_t batch, int32_t h, int32_t w, int32_t c,
                          int32_t* amax_bits /* may be NULL: [256] slots, max |dc| as in ammc_absmax_bits_f32 */,
                          void* stream);
/* The same backward with the S16 twin of dc as its output (training on the split-fp16 kernels, one rank):
 * reduce_bound: partial Q=4 = the two sums + per-channel max|g|, max|xhat|; finalize: sums[2][C] and an upper bound of
 * max|dc| (|scale| (max|g| + |sum_g|/M + max|xhat| |sum_gx|/M)) into the [256] amax slots (zeroed by the caller) - the
 * power-of-two rescaling of the gradient is known before dc exists; apply_s16: dc * 2^k as an S16 image (dc16; same
 * strides for the optional fp32 dc32), 2^-k into inv_scale[0..n_inv) for the consumers' epilogues. */
int ammc_bn_bwd_reduce_bound_f32(const float* c_raw, int64_t c_bs, int64_t c_rs, int64_t c_ps, const float* dy,
                                 int64_t d_bs, int64_t d_rs, int64_t d_ps, const float* mean, const float* invstd,
                                 const float* gamma, const float* beta, int32_t relu, int32_t batch, int32_t h,
                                 int32_t w, int32_t c, float* partial, void* stream);
int ammc_bn_bwd_finalize_f32(const float* partial, int32_t nblocks, int32_t c, int32_t pixels, const float* gamma,
                             float* sums, int32_t* amax_bits, void* stream);
int ammc_bn_bwd_apply_s16_f32(const float* c_raw, int64_t c_bs, int64_t c_rs, int64_t c_ps, const float* dy,
                              int64_t d_bs, int64_t d_rs, int64_t d_ps, const float* mean, const float* invstd,
                              const float* gamma, const float* beta, const float* sums, int32_t relu, float* dc16,
                              float* dc32 /* may be NULL */, int64_t o_bs, int64_t o_rs, int64_t o_ps, int32_t batch,
                              int32_t h, int32_t w, int32_t c, const int32_t* amax_bits, float* inv_scale,
                              int32_t n_inv, void* stream);
/* per-channel sum over pixels (bias gradients): partial Q=1 */
int ammc_chan_sum_f32(const float* x, int64_t x_bs, int64_t x_rs, int64_t x_ps, int32_t batch, int32_t h, int32_t w,
                      int32_t c, float* partial, void* stream);
/* the same pass with max |x| of the whole tensor left in the [256] amax slots (zeroed by the caller; as
 * ammc_absmax_bits_f32): the bias-gradient pass of a ConvTranspose also finds the power of two of its gradient's S16
 * re-encoding; then that re-encoding for a channel SLICE (strided form of ammc_split_rows_scaled_f32; c % 8 == 0) */
int ammc_chan_sum_absmax_f32(const float* x, int64_t x_bs, int64_t x_rs, int64_t x_ps, int32_t batch, int32_t h, int32_t w,
                             int32_t c, float* partial, int32_t* amax_bits, void* stream);
int ammc_split_scaled_strided_f32(const float* x, int64_t x_bs, int64_t x_rs, int64_t x_ps, float* y16, int64_t y_bs,
                                  int64_t y_rs, int64_t y_ps, int32_t batch, int32_t h, int32_t w, int32_t c,
                                  const int32_t* amax_bits, float* inv_scale, int32_t n, void* stream);
int ammc_reduce_partials_f32(const float* partial, int32_t nblocks, int32_t qc, float scale, float* out, void* stream);
/* the same per run of seg_rows rows: out[ceil(nblocks / seg_rows)][qc] - a first stage in front of ammc_bn_finalize_f32 /
 * ammc_bn_bwd_finalize_f32 when a convolution's statistics output (AmmcConvDesc.stats) has thousands of rows.  Columns
 * max_from .. qc - 1 (max_from % 8 == 0; = qc for none) are combined by max: the max |g|, max |xhat| rows of a
 * BatchNorm-backward partial (max_from = 2 c of its [4][c] rows). */
int ammc_reduce_partials_seg_f32(const float* partial, int32_t nblocks, int32_t qc, int32_t seg_rows, int32_t max_from,
                                 float* out, void* stream);
/* nn.MaxPool2d(2) backward (+ `add`, the gradient reaching the same tensor through the skip).  h, w: the pooled size;
 * in_h, in_w: the size of x / add / dx (2h or 2h+1: the last row / column of an odd size is in no window and gets `add`
 * alone, as MaxPool2d's floor does) */
int ammc_maxpool2x2_bwd_f32(const float* x, int64_t x_bs, int64_t x_rs, int64_t x_ps, const float* dp, int64_t p_bs,
                            int64_t p_rs, int64_t p_ps, const float* add, int64_t a_bs, int64_t a_rs, int64_t a_ps,
                            float* dx, int64_t o_bs, int64_t o_rs, int64_t o_ps, int32_t batch, int32_t h, int32_t w,
                            int32_t in_h, int32_t in_w, int32_t c, void* stream);
/* the same with the S16 twin of x as the pooled tensor (what ammc_maxpool2x2_s16 compared in the forward; c % 8 == 0): the
 * fp32 x then has no reader in the training step and is never written */
int ammc_maxpool2x2_bwd_s16x_f32(const float* x16, int64_t x_bs, int64_t x_rs, int64_t x_ps, const float* dp, int64_t p_bs,
                                 int64_t p_rs, int64_t p_ps, const float* add, int64_t a_bs, int64_t a_rs, int64_t a_ps,
                                 float* dx, int64_t o_bs, int64_t o_rs, int64_t o_ps, int32_t batch, int32_t h, int32_t w,
                                 int32_t in_h, int32_t in_w, int32_t c, void* stream);
/* the same from the window positions recorded by ammc_maxpool2x2_s16_idx in the forward (c % 8 == 0): the pooled tensor is
 * not read at all */
int ammc_maxpool2x2_bwd_idx_f32(const uint8_t* idx, const float* dp, int64_t p_bs, int64_t p_rs, int64_t p_ps, const float* add,
                                int64_t a_bs, int64_t a_rs, int64_t a_ps, float* dx, int64_t o_bs, int64_t o_rs, int64_t o_ps,
                                int32_t batch, int32_t h, int32_t w, int32_t in_h, int32_t in_w, int32_t c, void* stream);
/* BatchNorm (+ReLU) backward of a unit whose output was max-pooled in the forward, WITHOUT materialising its output
 * gradient: dy = add + MaxPool2d(2)-backward(dpo), formed on the fly from `add` (the gradient through the skip path, laid out
 * like dy), the pooled gradient dpo[batch][ph][pw][c] (own strides) and the window positions idx[batch][ph][pw][c] recorded
 * by the forward (ammc_maxpool2x2_s16_idx / ammc_scale_shift_act_s16_pool_f32); ph = h / 2, pw = w / 2 (floor).  Otherwise
 * ammc_bn_bwd_reduce_bound_f32 / ammc_bn_bwd_apply_s16_f32.  The apply form needs ammc_bn_bwd_unpool_supported(c, pixel
 * strides of c_raw / add / dc, w) != 0, else AMMC_EUNSUP (materialise dy with ammc_maxpool2x2_bwd_idx_f32). */
int ammc_bn_bwd_unpool_supported(int32_t c, int64_t c_ps, int64_t d_ps, int64_t o_ps, int32_t w);
/* 1 when ammc_scale_shift_act_s16_pool_f32 takes this geometry (even h, w; c / 8 a power of two <= 256; the row form's
 * stride limits; AMMC_ROW_KERNELS not 0), else 0: callers then run ammc_scale_shift_act_s16_f32 + ammc_maxpool2x2_s16_idx
 * (the same query shape as ammc_bn_bwd_unpool_supported for the backward side). */
int ammc_scale_shift_act_s16_pool_supported(int32_t c, int32_t h, int32_t w, int64_t x_rs, int64_t x_ps, int64_t y_rs,
                                            int64_t y_ps, int64_t p_ps);
int ammc_bn_bwd_reduce_bound_unpool_f32(const float* c_raw, int64_t c_bs, int64_t c_rs, int64_t c_ps, const float* add,
                                        int64_t d_bs, int64_t d_rs, int64_t d_ps, const float* dpo, int64_t p_bs, int64_t p_rs,
                                        int64_t p_ps, const uint8_t* idx, int32_t ph, int32_t pw, const float* mean,
                                        const float* invstd, const float* gamma, const float* beta, int32_t relu, int32_t batch,
                                        int32_t h, int32_t w, int32_t c, float* partial, void* stream);
int ammc_bn_bwd_apply_s16_unpool_f32(const float* c_raw, int64_t c_bs, int64_t c_rs, int64_t c_ps, const float* add,
                                     int64_t d_bs, int64_t d_rs, int64_t d_ps, const float* dpo, int64_t p_bs, int64_t p_rs,
                                     int64_t p_ps, const uint8_t* idx, int32_t ph, int32_t pw, const float* mean,
                                     const float* invstd, const float* gamma, const float* beta, const float* sums,
                                     int32_t relu, float* dc16, float* dc32, int64_t o_bs, int64_t o_rs, int64_t o_ps,
                                     int32_t batch, int32_t h, int32_t w, int32_t c, const int32_t* amax_bits,
                                     float* inv_scale, int32_t n_inv, void* stream);
/* torch.tanh backward at the module boundary: NCHW (dout, out) -> NHWC d(pre-tanh), cp channels */
int ammc_tanh_bwd_nhwc_f32(const float* dout_nchw, const float* out_nchw, int32_t batch, int32_t c, int32_t h,
                           int32_t w, float* y, int64_t y_bs, int64_t y_rs, int64_t y_ps, int32_t cp, void* stream);
/* gradient of the commit term and of q_one w.r.t. the encoder output (unet.py:310-311) */
int ammc_commit_bwd_f32(const float* z, const float* embed_md, const int32_t* idx_topk, int32_t k,
                        const float* ddiff, const float* dq, float* dz, int32_t n, int32_t d, void* stream);
/* the same update in two halves, for data-parallel training with synchronised statistics: per-slot counts [m]
 * and feature sums [d][m] of this rank (all-reduce them), then the EMA + renormalisation */
int ammc_codebook_count_f32(const float* x, const int32_t* idx_topk, int32_t k, int32_t n, int32_t d, int32_t m,
                            float* counts, float* sums, void* stream);
int ammc_codebook_ema_apply_f32(const float* counts, const float* sums, int32_t d, int32_t m, float decay,
                                float one_minus_decay, float eps, float* cluster_size, float* embed_avg, float* embed,
                                void* stream);
/* EMA codebook update (unet.py:298-309), deterministic (no atomics) */
int ammc_codebook_ema_f32(const float* x, const int32_t* idx_topk, int32_t k, int32_t n, int32_t d, int32_t m,
                          float decay, float one_minus_decay, float eps, float* cluster_size, float* embed_avg,
                          float* embed, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AMMC_HIP_H */
