/*
 * hrviton_hip.h -- C ABI of the MI355X (gfx950) HR-VITON hot path.
 *
 * The reference (sangyun884/HR-VITON) is pure Python/PyTorch: it has no FFI of
 * its own, so the "plugin boundary" for the hot path is the set of torch ops
 * its modules dispatch.  Every entry point below replaces
 * one such dispatch (file:line into /root/reference cited per function) and is
 * what a ctypes / cffi / pybind stub on the reference side would bind
 * (INTEGRATION.md shows the stub).  Conventions:
 *
 *   - plain C: raw device pointers + sizes, no torch types, no C++ in the
 *     signatures.  `hrv_stream_t` is a `hipStream_t` passed as void*.
 *   - activations are NHWC fp32 (or bf16 where the name says so); a tensor is
 *     addressed as base[pixel * cstride + coff + c] so channel-concatenation and
 *     channel-slicing never need a copy.  Channel counts handed to the conv
 *     engine are multiples of 4 (callers zero-pad, e.g. 9 -> 12).
 *   - all memory is caller-owned; kernels borrow pointers for the duration of
 *     the launch on `stream`; no hidden allocation, no global state but the
 *     thread-local last-error string.
 *   - return value: 0 (HRV_OK) or a negative hrv_status; hrv_last_error() gives
 *     the message.  Nothing here ever falls back to a CPU path.
 */
#ifndef HRVITON_HIP_H
#define HRVITON_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* hrv_stream_t; /* hipStream_t */

enum hrv_status {
  HRV_OK = 0,
  HRV_ERR_ARG = -1,    /* bad shape / alignment / null pointer */
  HRV_ERR_LAUNCH = -2, /* hipLaunch / hipGetLastError failure */
  HRV_ERR_NODEV = -3   /* no gfx950 device */
};

enum hrv_act { HRV_ACT_NONE = 0, HRV_ACT_RELU = 1, HRV_ACT_LRELU = 2, HRV_ACT_TANH = 3 };

const char* hrv_version(void);
const char* hrv_last_error(void);
/* 0 if a gfx950 device is visible to the process. */
int hrv_device_check(void);
/* Persistent kernels (one block per CU holding most of its LDS: the SPADE gamma|beta kernel, the patch tiles, the thin and
 * weight-gradient kernels) size their grids to the CURRENT device's CU count (cached per device) minus ``k`` reserved CUs,
 * so that kernels which must run concurrently with them -- RCCL's collectives during a data-parallel backward (the
 * reference's DataParallelWithCallback reduction, train_generator.py:171-178) -- find free CUs instead of queueing behind
 * 40 us .. 1 ms blocks.  Default 0; the environment variable HRV_RESERVE_CUS sets the initial value.
 * hrv_persistent_cus(): the resulting grid size. */
int hrv_set_reserved_cus(int32_t k);
int hrv_persistent_cus(void);
/* Diagnostics only (tools/gb_bench.py, tools/patch_timeline.py): a device buffer of 8 x uint64 per tile, OWNED BY THE TOOL,
 * into which the persistent tile kernels write per-tile phase timestamps while it is set; (NULL, 0) switches it off.  A
 * launch with more tiles than the buffer holds does not log. */
int hrv_diag_set_tlog(void* buf, int64_t tiles);
/* The library's A/B and diagnostic switches (environment variables HRV_*) are read from the environment once per
 * name and cached, so no launch path calls getenv() after a name's first use.  A process that changes one of them while it runs
 * (tests/, tools/conv_bench.py) calls this to drop the cache. */
int hrv_diag_reload_env(void);

/* ------------------------------------------------------------------------
 * Convolution engine (implicit GEMM on fp32 MFMA, v_mfma_f32_32x32x2_f32).
 * Replaces every nn.Conv2d on the path: networks.py:171-198 (ResBlock
 * 3x3 s1/s2, 1x1), :63-93 (conv1/conv2/flow_conv/bottleneck);
 * network_generator.py:97-99,132-143,184-186,201 (SPADE / resblock / stem
 * convs) and :263-272 (4x4 s2 PatchGAN), with the surrounding
 * eval-BatchNorm / bias / residual / ReLU|LeakyReLU|tanh fused as an epilogue:
 *     out = act( acc * scale[c] + shift[c] + residual )
 * and torch.cat(..., 1) / nn.Upsample(nearest, x2) of the *inputs* folded into
 * the gather (several sources, optional half-resolution source).
 * ---------------------------------------------------------------------- */
#define HRV_MAX_SRC 4

typedef struct hrv_src {
  const void* ptr;  /* NHWC base */
  int32_t C;        /* channels taken from this source (multiple of 4)       */
  int32_t cstride;  /* channels per pixel in memory (multiple of 4)          */
  int32_t coff;     /* first channel (multiple of 4)                         */
  int32_t up_shift; /* 0: source is H x W; 1: source is (H/2) x (W/2), read through
                       a nearest x2 upsample (network_generator.py:203);
                       -k: source is (H<<k) x (W<<k), read through a nearest
                       1/2^k downsample (F.interpolate nearest, :164,222)    */
  int32_t pre_act;  /* must be HRV_ACT_NONE for the MFMA engine (activations are
                       fused into the producer's epilogue); LRELU(0.2) is
                       honoured by the naive cross-check only               */
  int32_t C_real;   /* channels that exist in the raw OIHW weight for this
                       source (<= C; 0 means C).  Only the naive cross-check
                       reads it; the packed weight already has zero rows.    */
} hrv_src_t;

/* SPADE epilogue (network_generator.py:101-122 + the LeakyReLU of :170-171): when
 * hrv_conv2d_t.spade is set the convolution is the fused conv_gamma||conv_beta
 * (128 -> 2C, 3x3) and the epilogue modulates the tensor being normalised:
 *     v   = x + noise_z[n,w,h] * noise_scale[c]
 *     out = act( (v - mean[n,c]) * rstd[n,c] * (1 + gamma) + beta )
 * w_packed holds 2*ceil32(C) output columns ordered in (gamma[32] | beta[32])
 * pairs per 32-channel group; `shift` holds the two biases in the same column
 * order; `scale`/`residual` are ignored; `Cout` is that column count; `out`
 * has C channels.  mean/rstd come from hrv_instnorm_stats_nhwc_f32. */
typedef struct hrv_spade_epi {
  const float* x;      /* NHWC, same pixels as out */
  int32_t x_cstride, x_coff;
  int32_t C;           /* channels (multiple of 4; pad channels are zero)     */
  int32_t _pad;
  const float* mean;   /* [N][C] */
  const float* rstd;   /* [N][C] */
  const float* noise_z;     /* [N][W][H] N(0,1) draw, layout of torch.randn(b,w,h,1)
                               (network_generator.py:104-107), or NULL        */
  const float* noise_scale; /* [C] or NULL */
  float* g1p_out;      /* training: (1 + gamma) is also stored here, NHWC [.,C] dense
                          (needed by hrv_spade_norm_bwd_nhwc_f32), or NULL          */
} hrv_spade_epi_t;

typedef struct hrv_conv2d {
  int32_t N, H, W;   /* conv input extent (after any folded upsample)        */
  int32_t Ho, Wo;    /* output extent                                        */
  int32_t KH, KW, stride, pad;
  int32_t nsrc;
  hrv_src_t src[HRV_MAX_SRC];
  const void* w_packed; /* from hrv_conv2d_pack_weight_f32 (device copy)     */
  const float* w_oihw;  /* device copy of the raw [Cout][Cin][KH][KW] weight;
                           only read by hrv_conv2d_naive_nhwc_f32            */
  int32_t Cout;
  int32_t tile_cfg;     /* from hrv_conv2d_pick_tile                         */
  const float* scale;   /* [Cout] or NULL (=1)                               */
  const float* shift;   /* [Cout] or NULL (=0)                               */
  const void* residual; /* NHWC, same pixels as out, or NULL                 */
  int32_t res_cstride, res_coff;
  int32_t act;          /* hrv_act                                           */
  float act_slope;      /* LeakyReLU negative slope                          */
  void* out;            /* NHWC                                              */
  int32_t out_cstride, out_coff;
  const hrv_spade_epi_t* spade; /* NULL: standard epilogue                   */
  int32_t out_up_shift; /* 1: `out` is (2*Ho) x (2*Wo) and every result is stored
                           to its 2x2 block -- the nn.Upsample(nearest, x2) that
                           follows each SPADEResBlock (network_generator.py:226-
                           241) fused into the producer's store; 0: plain      */
  int32_t _pad2;
  void* workspace;          /* optional scratch for split-K (small-M / large-K layers):   */
  int64_t workspace_bytes;  /* >= hrv_conv2d_workspace_bytes(d); NULL => run unsplit     */
  /* --- backward-pass extensions (all 0 for an ordinary forward convolution) --- */
  int32_t pad_w_plus1;  /* 0: horizontal padding = pad; else horizontal padding + 1        */
  int32_t free_extent;  /* 1: Ho/Wo are taken as given; taps outside the input read zeros  */
  int32_t out_step;     /* 2: row (n,h,w) is stored at (2h+out_off_h, 2w+out_off_w) of an
                           out_H x out_W tensor: one phase of a stride-2 data gradient     */
  int32_t out_off_h, out_off_w, out_H, out_W;
  int32_t res_mode;     /* 0: epilogue adds `residual`; 1: multiplies by the activation
                           derivative (residual > 0 ? 1 : act_slope) -- LeakyReLU/ReLU
                           backward fused into the data-gradient convolution             */
  int32_t mixed_flags;  /* hrv_conv2d_nhwc_bf16 only: bit0 = `out` is fp32, bit1 = `residual` is fp32,
                           bit2 = the SPADE `x` is fp32 (tensors that feed an InstanceNorm stay fp32),
                           bit3 = the conv SOURCES are fp32 and are rounded to bf16 while staged
                           (mixed-precision training; needs a 128-byte-row tile, cfg 8/9) */
  int32_t _pad3;
} hrv_conv2d_t;

/* Tile configuration for (M = N*Ho*Wo output pixels, Cout): returns cfg id. */
int hrv_conv2d_pick_tile(int64_t M, int32_t Cout);
/* BN (output-channel tile) / BM (pixel tile) of a cfg; <0 if cfg is invalid. */
int hrv_conv2d_tile_bn(int32_t tile_cfg);
int hrv_conv2d_tile_bm(int32_t tile_cfg);
/* Bytes per packed K-tile row of the bf16 engine for a cfg (64 = 32 k-values, 128 = 64 k-values); <0 if invalid. */
int hrv_conv2d_tile_row_bytes(int32_t tile_cfg);
/* Number of floats of the packed weight for this layer. */
int64_t hrv_conv2d_packed_elems(int32_t Cout, int32_t KH, int32_t KW, int32_t nsrc,
                                const int32_t* srcC, int32_t tile_cfg);
/* HOST function: repack w[Cout][sum(srcC_real)][KH][KW] (torch layout) into the
 * K-tiled layout the MFMA kernel streams:  [kt][CoutPad][16] with
 * kt = ((kh*KW + kw) * chunks_total + chunk) and chunk running over the
 * sources' 16-channel groups.  srcC[i] are the padded per-source channel
 * counts used by the kernel (multiples of 4), srcC_real[i] the channels that
 * exist in w (<= srcC[i]); padded rows are zero. */
int hrv_conv2d_pack_weight_f32(const float* w_oihw, int32_t Cout, int32_t KH, int32_t KW, int32_t nsrc,
                               const int32_t* srcC, const int32_t* srcC_real, int32_t tile_cfg,
                               float* out);
/* ---- training side (conv_bwd.hip) ---------------------------------------
 * Device-side weight packing (weights change every optimiser step).  mode 0: the
 * forward layout of hrv_conv2d_pack_weight_f32; mode 1: the stride-1 data gradient
 * (rows = Cin, k over Cout, taps flipped; run the engine on dY with pad K-1-pad);
 * mode 2: one phase (phase_a, phase_b) of the stride-2 data gradient
 * (dX[2h'+a] = sum_j dY[h'+j-pad_p] W[a+pad-2(j-pad_p)]).  `wscale` multiplies
 * every weight; if sigma_dev != NULL the weights are also divided by the device
 * scalar sigma_dev[0] (spectral norm, no host sync).  out_geom[8] = {KHp, KWp, pad_h, pad_w,
 * rows, rows_pad, chunks_total, packed elems}; out_dev must hold that many floats
 * (<= KH*KW*chunks*rows_pad*16).
 * Weight gradient of one source of a (possibly concatenated) input:
 *   dW[co][ci_base+ci][kh][kw] (+)= sum_p dY[p][co] * X[p shifted by (kh,kw)][ci]
 * MFMA GEMM over pixel slabs + fixed-order reduction (deterministic).
 * colsum: out[c] (+)= sum_p x[p][c]  (bias gradient).
 * ---------------------------------------------------------------------- */
int hrv_conv2d_pack_weight_dev_f32(const float* w_oihw_dev, int32_t Cout, int32_t KH, int32_t KW, int32_t nsrc,
                                   const int32_t* srcC, const int32_t* srcC_real, int32_t tile_cfg, int32_t mode,
                                   int32_t stride, int32_t pad, int32_t phase_a, int32_t phase_b, float wscale,
                                   const float* sigma_dev, float* out_dev, int32_t* out_geom, hrv_stream_t stream);
/* Same packing, rounded to bf16, for the bf16 matrix-core engine (hrv_conv2d_nhwc_bf16): rows of 64 k-values
 * for the 128-byte-row tiles (tile_cfg 8/9), 32 otherwise.  Used by mixed-precision training, where the
 * activations stay fp32 in HBM (hrv_conv2d_t.mixed_flags bit 3) and only the MFMA operands are bf16. */
int hrv_conv2d_pack_weight_dev_bf16(const float* w_oihw_dev, int32_t Cout, int32_t KH, int32_t KW, int32_t nsrc,
                                    const int32_t* srcC, const int32_t* srcC_real, int32_t tile_cfg, int32_t mode,
                                    int32_t stride, int32_t pad, int32_t phase_a, int32_t phase_b, float wscale,
                                    const float* sigma_dev, uint16_t* out_dev, int32_t* out_geom, hrv_stream_t stream);
/* The same packing for a weight PAIR -- SPADE conv_gamma / conv_beta (network_generator.py:93-118), both
 * [rows_each][Cin][KH][KW] -- without materialising the combined matrix.  pair_mode 1: forward (mode 0), virtual
 * Cout = 64*ceil(rows_each/32) rows interleaved (gamma32 | beta32), the column order of the fused modulate epilogue;
 * pair_mode 2: stride-1 data gradient (mode 1) over dY = [dgamma | dbeta], Cout = 2 * (channels per half). */
int hrv_conv2d_pack_weight_pair_dev(const float* w_a_dev, const float* w_b_dev, int32_t rows_each, int32_t pair_mode,
                                    int32_t Cout, int32_t KH, int32_t KW, int32_t nsrc, const int32_t* srcC,
                                    const int32_t* srcC_real, int32_t tile_cfg, int32_t mode, int32_t pad, int32_t as_bf16,
                                    void* out_dev, int32_t* out_geom, hrv_stream_t stream);
/* Batched form of the three packs above: every weight of a network re-packed by ONE launch per training step (the
 * reference re-derives nothing -- cuDNN reads OIHW -- so this replaces ~180 small launches of this library's own).
 * hrv_conv2d_pack_weight_record fills a host record (hrv_conv2d_pack_record_bytes() bytes) for one pack -- same argument
 * meaning; as_bf16 selects the bf16 form; w2_dev / pair_mode / rows_each describe a pair, else null / 0 / 0 -- and
 * returns its geometry and block count; the caller keeps the records in a device array, their block prefix sums in
 * first_block_dev[n + 1], and launches hrv_conv2d_pack_weight_multi whenever the weights changed. */
int32_t hrv_conv2d_pack_record_bytes(void);
int hrv_conv2d_pack_weight_record(const float* w_oihw_dev, int32_t Cout, int32_t KH, int32_t KW, int32_t nsrc,
                                  const int32_t* srcC, const int32_t* srcC_real, int32_t tile_cfg, int32_t mode,
                                  int32_t stride, int32_t pad, int32_t phase_a, int32_t phase_b, float wscale,
                                  const float* sigma_dev, int32_t as_bf16, const float* w2_dev, int32_t pair_mode,
                                  int32_t rows_each, void* out_dev, int32_t* out_geom, void* record_host, int32_t* blocks);
int hrv_conv2d_pack_weight_multi(const void* records_dev, const int32_t* first_block_dev, int32_t n, int32_t blocks,
                                 hrv_stream_t stream);
int64_t hrv_conv2d_wgrad_workspace_bytes(int32_t Cout, int32_t CinTot, int32_t KH, int32_t KW, int64_t P);
int hrv_conv2d_wgrad_nhwc_f32(const float* dy, int32_t dy_cstride, int32_t dy_coff, int32_t Cout, const float* x,
                              int32_t x_C, int32_t x_cstride, int32_t x_coff, int32_t x_up_shift, int32_t x_C_real,
                              int32_t ci_base, int32_t CinTot, int32_t N, int32_t H, int32_t W, int32_t Ho, int32_t Wo,
                              int32_t KH, int32_t KW, int32_t stride, int32_t pad, float* workspace,
                              int64_t workspace_bytes, float* dw_oihw, int32_t accumulate, float* dbias,
                              int32_t dbias_accumulate, hrv_stream_t stream);
/* dbias (optional, [Cout]): the bias gradient sum_pixels dY[:, co], fused as one extra "ones" column of the same
 * MFMA reduction (no second pass over dY; deterministic).  Pass it with the FIRST source of a concatenation only.
 * Same contract with the operands rounded to bf16 while staged and multiplied on v_mfma_f32_32x32x16_bf16 (fp32
 * accumulate): the weight gradient of mixed-precision training.  Requires Wo % 4 == 0. */
int hrv_conv2d_wgrad_bf16mma_nhwc_f32(const float* dy, int32_t dy_cstride, int32_t dy_coff, int32_t Cout, const float* x,
                              int32_t x_C, int32_t x_cstride, int32_t x_coff, int32_t x_up_shift, int32_t x_C_real,
                              int32_t ci_base, int32_t CinTot, int32_t N, int32_t H, int32_t W, int32_t Ho, int32_t Wo,
                              int32_t KH, int32_t KW, int32_t stride, int32_t pad, float* workspace,
                              int64_t workspace_bytes, float* dw_oihw, int32_t accumulate, float* dbias,
                              int32_t dbias_accumulate, hrv_stream_t stream);
/* Same again with bf16-STORED operands: storage_flags bit0 = `dy` points at bf16 elements, bit1 = `x` does
 * (counts in elements, multiples of 4).  Tensors that only matrix cores read are kept in bf16 by the
 * mixed-precision training plan (same MMA operand bits, half the bytes).  Supported: 0, 2 (x), 3 (both). */
int hrv_conv2d_wgrad_bf16mma_st_nhwc_f32(const void* dy, int32_t dy_cstride, int32_t dy_coff, int32_t Cout,
                              const void* x, int32_t x_C, int32_t x_cstride, int32_t x_coff, int32_t x_up_shift,
                              int32_t x_C_real, int32_t ci_base, int32_t CinTot, int32_t N, int32_t H, int32_t W,
                              int32_t Ho, int32_t Wo, int32_t KH, int32_t KW, int32_t stride, int32_t pad,
                              float* workspace, int64_t workspace_bytes, float* dw_oihw, int32_t accumulate,
                              float* dbias, int32_t dbias_accumulate, int32_t storage_flags, hrv_stream_t stream);
int hrv_colsum_nhwc_f32(const float* x, int64_t P, int32_t C, int32_t cstride, int32_t coff, float* workspace,
                        int64_t workspace_bytes, float* out, int32_t accumulate, hrv_stream_t stream);

/* ---- training side, HBM-bound kernels (train.hip) -------------------------
 * SPADE / InstanceNorm backward (network_generator.py:101-122 + LeakyReLU :170-171;
 * PatchGAN IN+LeakyReLU :263-272) for
 *     v = x + noise_z*noise_scale;  nh = (v - mean)*rstd;  out = act(nh*g1p + beta)
 * given dout:  dpre = dout*act'(out); dnh = dpre*g1p; dgamma = dpre*nh; dbeta = dpre;
 *     dx = rstd*(dnh - mean_hw(dnh) - nh*mean_hw(dnh*nh));  dnoise_scale = sum dx*z.
 * g1p = 1+gamma (NULL: plain InstanceNorm, dgb must be NULL); dgb receives
 * [dgamma | dbeta] (2C channels); dnh is a C-channel scratch/output tensor.
 * workspace: hrv_norm_bwd_workspace_elems floats.  Deterministic (2-stage sums).
 * dbeta in place: when `dout` IS the dbeta half of `dgb` (same base pointer, same pixel stride, dout_coff == dgb_coff + C,
 * same storage type, act == NONE -- the data gradient that produced dout wrote it there with the activation derivative
 * applied in its own epilogue) the kernel reads it from there and does not store dbeta again. */
typedef struct hrv_norm_bwd {
  int32_t N, H, W, C;
  const float* x;     int32_t x_cstride, x_coff;
  const float* noise_z; const float* noise_scale;      /* nullable (together) */
  const float* mean;  const float* rstd;               /* [N][C] */
  const float* out;   int32_t out_cstride, out_coff;   /* activation output (its mask) */
  const float* g1p;   int32_t g1p_cstride, g1p_coff;   /* 1+gamma, nullable */
  const float* dout;  int32_t dout_cstride, dout_coff;
  float* dnh;         int32_t dnh_cstride, dnh_coff;
  float* dgb;         int32_t dgb_cstride, dgb_coff;   /* nullable */
  float* dx;          int32_t dx_cstride, dx_coff;
  int32_t dx_accumulate;                               /* 1: dx += ... */
  int32_t act;        float act_slope;
  int32_t dns_accumulate;
  float* dnoise_scale;                                 /* [C], nullable */
  float* workspace;
  int32_t dgb_bf16;   /* 1: `dgb` is stored as bf16 (element strides/offsets): a tensor only matrix cores read */
  int32_t out_bf16;   /* 1: `out` is stored as bf16 (only its sign is used here)                              */
  int32_t dx_bf16;    /* 1: `dx` is stored as bf16 (no accumulate): the gradient of a convolution output that only
                       *    that convolution's weight / data gradient (matrix cores) read                          */
  int32_t g1p_bf16;   /* 1: `g1p` is stored as bf16 (written by hrv_spade_gb_bf16; this kernel is its only reader) */
  int32_t dnh_bf16;   /* 1: the stage-1 -> stage-2 intermediate `dnh` is stored as bf16 (mixed precision: half its bytes) */
  int32_t dout_bf16;  /* 1: `dout` is stored as bf16 (mixed precision: the data gradient of a bf16-stored SPADE output, as
                       * autocast hands the gradient of a half-precision convolution input back in half precision) */
  /* x = cat(nearest_up2(lo), hi) along channels, never materialised (x_up_channels > 0; SPADEGenerator.up + the resized-input
   * concatenation, network_generator.py:203,226-242): channels [0, x_up_channels) are read from `x` = lo [N][H/2][W/2][x_cstride]
   * at (h >> 1, w >> 1), the remaining C - x_up_channels from `x2` = hi [N][H][W][x2_cstride] */
  int32_t x_up_channels;
  const float* x2;    int32_t x2_cstride, x2_coff;
} hrv_norm_bwd_t;
int64_t hrv_norm_bwd_workspace_elems(int32_t N, int32_t H, int32_t W, int32_t C);

/* Thin convolution (thin_conv.hip): 3x3 / 1x1 stride-1 'same' nn.Conv2d with <= 96 channels on either side over a
 * bf16-STORED NHWC source, forward (mode 0) or data gradient (mode 1: `src` is dY with Cout channels, the result has
 * Cin columns, taps flipped) -- SPADEResBlock.conv_0 / conv_1 / conv_s at 1024x768 and conv_img
 * (network_generator.py:141-143,201) in mixed-precision training.  `w_oihw` is the fp32 parameter on the device
 * (weight_orig for spectral-norm layers; the kernel multiplies it by wscale / sigma[0]).
 * out = act((conv + shift[c]) (+ residual | * act'(residual) for res_mode 1)), fp32 or bf16.
 * hrv_thin_conv_supported() says whether a (kernel, source-channel, column) combination is instantiated. */
typedef struct {
  const void* src; int32_t N, H, W, src_channels, src_cstride, src_coff;
  const float* w_oihw; int32_t Cout, Cin, KH, KW; const float* sigma; float wscale;
  int32_t mode;
  const float* shift;
  const void* residual; int32_t res_cstride, res_coff, res_bf16, res_mode;
  int32_t act; float act_slope;
  void* out; int32_t out_cstride, out_coff, out_bf16;
} hrv_thin_conv_t;
int hrv_thin_conv_supported(int32_t KH, int32_t KW, int32_t src_channels, int32_t out_columns);
int hrv_thin_conv_bf16(const hrv_thin_conv_t* d, hrv_stream_t stream);
int hrv_spade_norm_bwd_nhwc_f32(const hrv_norm_bwd_t* d, hrv_stream_t stream);
/* Two normalisations of the SAME x (norm_0 and norm_s of a learned-shortcut SPADEResBlock, network_generator.py:158-166) in one
 * pass per stage: x is read once, a->dx <- dx_a + dx_b is written once (b->dx is ignored); everything else per descriptor.
 * Bit-identical to hrv_spade_norm_bwd_nhwc_f32(a) followed by (b with dx = a->dx, dx_accumulate = 1). */
int hrv_spade_norm_bwd2_nhwc_f32(const hrv_norm_bwd_t* a, const hrv_norm_bwd_t* b, hrv_stream_t stream);
/* Loss value + gradient in one pass.  mode 0: L1 |a-b| (feature matching / VGG,
 * train_generator.py:300-312); 1: hinge-D fake max(1+a,0); 2: hinge-D real max(1-a,0);
 * 3: -a (generator hinge) (network_generator.py:365-376); 4: (a-b)^2 (LSGAN, networks.py:258-299).
 * loss_out[0] (+)= lscale*sum(l);  grad[i] = gscale*dl/da (grad may be NULL).
 * workspace: 1024 floats. */
int hrv_loss_f32(const float* a, const float* b, int64_t n, int32_t mode, float lscale, float gscale, float* grad,
                 float* workspace, float* loss_out, int32_t accumulate, hrv_stream_t stream);
int hrv_loss_bf16in_f32(const uint16_t* a, const uint16_t* b, int64_t n, int32_t mode, float lscale, float gscale, float* grad,
                        float* workspace, float* loss_out, int32_t accumulate, hrv_stream_t stream);
/* tanh backward through its output: out = dy*(1-y*y) (network_generator.py:245);
 * add_slice: out[.., out_coff:+C] (+)= a[.., a_coff:+C]  (gradient accumulation / channel split). */
/* x *= s_host * (s_dev ? s_dev[0] : 1): applies an upstream (device-resident) loss-gradient scalar. */
int hrv_scale_f32(float* x, int64_t n, float s_host, const float* s_dev, hrv_stream_t stream);
/* d *= act'(y) in place (ReLU / LeakyReLU derivative through the activation output y). */
int hrv_act_bwd_nhwc_f32(float* d, int32_t d_cstride, int32_t d_coff, const float* y, int32_t y_cstride, int32_t y_coff,
                         int32_t C, int64_t npix, int32_t act, float act_slope, hrv_stream_t stream);
int hrv_tanh_bwd_f32(const float* dy, const float* y, int64_t n, float* out, hrv_stream_t stream);
int hrv_add_slice_nhwc_f32(const float* a, int32_t a_cstride, int32_t a_coff, float* out, int32_t out_cstride,
                           int32_t out_coff, int32_t C, int64_t npix, int32_t accumulate, hrv_stream_t stream);
/* nn.Upsample(nearest, x2) backward: dlo (+)= 2x2 block sums of dhi.  accumulate: 0 write, 1 add, 2 write as bf16
 * (lo_cstride / lo_coff then count bf16 elements). */
int hrv_downsum2x2_nhwc_f32(const float* dhi, int32_t N, int32_t Hl, int32_t Wl, int32_t C, int32_t hi_cstride,
                            int32_t hi_coff, float* dlo, int32_t lo_cstride, int32_t lo_coff, int32_t accumulate,
                            hrv_stream_t stream);
int hrv_avgpool3x3s2_bwd_nhwc_f32(const float* dy, int32_t N, int32_t H, int32_t W, int32_t C, int32_t dy_cstride,
                                  int32_t dy_coff, float* dx, int32_t dx_cstride, int32_t dx_coff, int32_t accumulate,
                                  hrv_stream_t stream);
/* VGG19 2x2/2 max pool (networks.py:201-232 via torchvision cfg 'E'), dense NHWC. */
int hrv_maxpool2x2_nhwc_f32(const float* x, int32_t N, int32_t H, int32_t W, int32_t C, float* y, hrv_stream_t stream);
int hrv_maxpool2x2_bwd_nhwc_f32(const float* x, const float* dy, int32_t N, int32_t H, int32_t W, int32_t C, float* dx,
                                hrv_stream_t stream);
/* ... with the ReLU derivative of the pooled tensor fused (x = ReLU(pre); dx is the gradient w.r.t. pre). */
int hrv_maxpool2x2_bwd_relu_nhwc_f32(const float* x, const float* dy, int32_t N, int32_t H, int32_t W, int32_t C, float* dx,
                                     hrv_stream_t stream);
/* bf16-stored activations (mixed-precision VGG19: matrix cores, pools and the L1 taps are their only readers;
 * forward bf16 -> bf16 for NON-NEGATIVE inputs (ReLU outputs: integer max of the stored patterns, exact); backward with
 * bf16 x, fp32 dy / dx; loss over bf16 a, b. */
int hrv_maxpool2x2_nhwc_bf16(const uint16_t* x, int32_t N, int32_t H, int32_t W, int32_t C, uint16_t* y, hrv_stream_t stream);
/* mixed-precision VGG19 BACKWARD keeps its gradient tensors in bf16 too (their readers are the data-gradient matrix cores,
 * the pool routing and these sums): pool backward over bf16 x / dy / dx; out (+)= a over bf16 tensors (fp32 sum, one
 * rounding); hrv_loss_bf16in_f32 with mode | 32 stores its gradient as bf16 (`grad` then points at bf16 elements). */
int hrv_maxpool2x2_bwd_relu_nhwc_bf16(const uint16_t* x, const uint16_t* dy, int32_t N, int32_t H, int32_t W, int32_t C,
                                      uint16_t* dx, hrv_stream_t stream);
int hrv_add_slice_nhwc_bf16(const uint16_t* a, int32_t a_cstride, int32_t a_coff, uint16_t* out, int32_t out_cstride,
                            int32_t out_coff, int32_t C, int64_t npix, int32_t accumulate, hrv_stream_t stream);
int hrv_maxpool2x2_bwd_relu_nhwc_xbf16(const uint16_t* x, const float* dy, int32_t N, int32_t H, int32_t W, int32_t C,
                                       float* dx, hrv_stream_t stream);
/* torch.optim.Adam step over one flat buffer (train_generator.py:154-157,322,360);
 * g is multiplied by grad_scale first (1/world_size after a sum all-reduce). */
int hrv_adam_f32(float* w, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                 float weight_decay, int32_t step, float grad_scale, hrv_stream_t stream);
/* The same update with its step-dependent scalars on the device, for a hipGraph-captured training iteration (a replay must
 * not freeze the step count or the learning rate into launch arguments): hrv_adam_hyper_f32 increments step_dev[0] and
 * writes hyper_dev[3] = {lr_dev[0], 1 - beta1^t, sqrt(1 - beta2^t)}; hrv_adam_dev_f32 reads them. */
int hrv_adam_hyper_f32(int32_t* step_dev, const float* lr_dev, float beta1, float beta2, float* hyper_dev, hrv_stream_t stream);
int hrv_adam_dev_f32(float* w, const float* g, float* m, float* v, int64_t n, const float* hyper_dev, float beta1, float beta2,
                     float eps, float weight_decay, float grad_scale, hrv_stream_t stream);
/* torch spectral_norm (SpectralNorm.compute_weight): `power_iterations` rounds of
 * v<-normalize(W^T u), u<-normalize(W v) in place (eps 1e-12), then sigma = u.(W v).
 * W is [R][K] = weight_orig.reshape(Cout,-1); wv_scratch holds R floats.
 * bwd: dW_orig (+)= (G - <G,W_orig>/sigma * u v^T)/sigma; workspace 1024 floats. */
int hrv_spectral_norm_f32(const float* w, int32_t R, int32_t K, float* u, float* v, int32_t power_iterations, float eps,
                          float* wv_scratch, float* sigma_out, hrv_stream_t stream);
int hrv_spectral_norm_bwd_f32(const float* G, const float* w_orig, const float* u, const float* v, const float* sigma,
                              int32_t R, int32_t K, float* workspace, float* dw_orig, int32_t accumulate,
                              hrv_stream_t stream);
/* Per-step parameter staging of a SPADE norm / block, one launch each:
 *  vec_prep: bias of the fused gamma|beta convolution in its interleaved column order (64*ceil(C/32) floats) and the
 *            noise scale zero-padded to ceil4(C);
 *  shared_taps_prep: conv_shared (label_nc=c -> hid, 3x3) of the n <= 4 norms of a block as ONE 1x1 weight
 *            wt[n*hid][9*cp] over the tap-expanded label map (tap-major, cp channels per tap) + concatenated bias;
 *  shared_taps_grad: the inverse map of that weight's gradient into the n conv_shared weight / bias gradients. */
int hrv_spade_vec_prep_f32(const float* gamma_bias, const float* beta_bias, const float* noise_scale, int32_t C,
                           float* bias_interleaved, float* noise_scale_padded, hrv_stream_t stream);
/* ... of n <= 32 norms in one launch: job i writes its interleaved bias at bias_interleaved_all + off_bias[i] ((C[i] + 31) / 32 * 64
 * floats) and its padded noise scale at noise_scale_padded_all + off_noise[i] (ceil4(C[i]) floats); offsets are multiples of 4 */
int hrv_spade_vec_prep_multi_f32(int32_t n, const float* const* gamma_bias, const float* const* beta_bias, const float* const* noise_scale,
                                 const int32_t* C, const int32_t* off_bias, const int32_t* off_noise, float* bias_interleaved_all,
                                 float* noise_scale_padded_all, hrv_stream_t stream);
int hrv_shared_taps_prep_f32(const float* const* w, const float* const* b, int32_t n, int32_t hid, int32_t c, int32_t cp,
                             float* wt, float* bt, hrv_stream_t stream);
int hrv_shared_taps_grad_f32(const float* dw, const float* db, int32_t n, int32_t hid, int32_t c, int32_t cp,
                             float* const* gw, float* const* gb, hrv_stream_t stream);
/* The same power iteration for EVERY spectral-normalised convolution of a network in four launches (27 layers in the
 * SPADE generator, network_generator.py:121-143 via add_spectral_norm; per layer it is eight launches of microseconds).
 * u_keep / v_keep (optional): copies of the (u, v) that produced sigma, for the backward.  wv_scratch: sum of R floats. */
typedef struct {
  const float* w;         /* weight_orig as [R][K] */
  float *u, *v;           /* updated in place when power_iterations > 0 */
  float* sigma;           /* one float per job */
  float *u_keep, *v_keep; /* or NULL */
  int32_t R, K;
} hrv_sn_job_t;
int hrv_spectral_norm_batched_f32(const hrv_sn_job_t* jobs, int32_t n_jobs, int32_t power_iterations, float eps,
                                  float* wv_scratch, hrv_stream_t stream);

/* Scratch the engine would like for this launch (0: none).  Layers with fewer than
 * ~192 output tiles split their K range over up to 32 blocks per tile (partials in
 * the workspace, fixed-order reduction + epilogue in a second kernel: deterministic). */
int64_t hrv_conv2d_workspace_bytes(const hrv_conv2d_t* d);
int hrv_conv2d_nhwc_f32(const hrv_conv2d_t* d, hrv_stream_t stream);
/* bf16 storage / fp32 accumulate flavour of the same engine (v_mfma_f32_32x32x16_bf16): sources,
 * packed weights, residual, the SPADE x tensor and the output are bf16 (uint16_t bit patterns);
 * scale / shift / mean / rstd / noise stay fp32.  Channel counts, strides and offsets are multiples
 * of 8 (one 16-byte gather group).  Same descriptor; w_packed from hrv_conv2d_pack_weight_bf16. */
int64_t hrv_conv2d_packed_elems_bf16(int32_t Cout, int32_t KH, int32_t KW, int32_t nsrc, const int32_t* srcC,
                                     int32_t tile_cfg);
int hrv_conv2d_pack_weight_bf16(const float* w_oihw, int32_t Cout, int32_t KH, int32_t KW, int32_t nsrc,
                                const int32_t* srcC, const int32_t* srcC_real, int32_t tile_cfg, uint16_t* out);
int hrv_conv2d_nhwc_bf16(const hrv_conv2d_t* d, hrv_stream_t stream);
/* Same contract, one thread per output element, raw OIHW weights.  A device
 * side cross-check used by the tests to localise faults; never on the product
 * path. */
int hrv_conv2d_naive_nhwc_f32(const hrv_conv2d_t* d, hrv_stream_t stream);

/* Second half of a KxK stride-1 'same' convolution with a tiny Cout (the
 * 768->2 flow_conv, networks.py:85-92,122,137), run as a 1x1 convolution with
 * KH*KW*Cout "tap channels" on the conv engine (input read once, 16x fewer
 * padded MFMA columns) followed by this gather:
 *   out[p][co] = bias[co] + sum_tap y[p + off(tap)][tap*Cout + co] (+ residual[p][co]) */
int hrv_tapsum_nhwc_f32(const float* y, int32_t N, int32_t H, int32_t W, int32_t KH, int32_t KW, int32_t pad,
                        int32_t Cout, int32_t y_cstride, const float* bias, const float* residual,
                        int32_t res_cstride, float* out, int32_t out_cstride, hrv_stream_t stream);

/* ------------------------------------------------------------------------
 * InstanceNorm2d(affine=False) pieces (network_generator.py:86,427; eps 1e-5,
 * biased variance over H*W per (n,c)).  stats: mean/rstd of x (+ SPADE noise
 * z[n,w,h]*noise_scale[c], :104-110) -- two deterministic stages (shifted
 * partial sums, double-precision fixed-order finalise); `workspace` needs
 * hrv_instnorm_workspace_elems floats.  apply: out = act((x-mean)*rstd).
 * avgpool: F.avg_pool2d(3, stride 2, pad 1, count_include_pad=False) (:301-302).
 * ---------------------------------------------------------------------- */
int64_t hrv_instnorm_workspace_elems(int32_t N, int32_t H, int32_t W, int32_t C);
int hrv_instnorm_stats_nhwc_f32(const float* x, int32_t N, int32_t H, int32_t W, int32_t C, int32_t cstride,
                                int32_t coff, const float* noise_z, const float* noise_scale, float eps,
                                float* workspace, float* mean, float* rstd, hrv_stream_t stream);
/* Two normalisations of the SAME x with different noise terms (norm_0 and norm_s of a learned-shortcut SPADEResBlock,
 * network_generator.py:158-166) in one pass over x; workspace: 2 x hrv_instnorm_workspace_elems floats.  Results are
 * bit-identical to two hrv_instnorm_stats_nhwc_f32 calls. */
int hrv_instnorm_stats2_nhwc_f32(const float* x, int32_t N, int32_t H, int32_t W, int32_t C, int32_t cstride, int32_t coff,
                                 const float* z_a, const float* ns_a, const float* z_b, const float* ns_b, float eps, float* workspace,
                                 float* mean_a, float* rstd_a, float* mean_b, float* rstd_b, hrv_stream_t stream);
/* ... the same over x = cat(nearest_up2(lo), hi) (see hrv_norm_bwd_t: x_up_channels), C = up_channels + channels of hi */
int hrv_instnorm_stats2_up_nhwc_f32(const float* lo, int32_t lo_cstride, int32_t lo_coff, int32_t up_channels, const float* hi,
                                    int32_t hi_cstride, int32_t hi_coff, int32_t N, int32_t H, int32_t W, int32_t C, const float* z_a,
                                    const float* ns_a, const float* z_b, const float* ns_b, float eps, float* workspace, float* mean_a,
                                    float* rstd_a, float* mean_b, float* rstd_b, hrv_stream_t stream);
int hrv_instnorm_stats_nhwc_bf16(const uint16_t* x, int32_t N, int32_t H, int32_t W, int32_t C, int32_t cstride,
                                 int32_t coff, const float* noise_z, const float* noise_scale, float eps,
                                 float* workspace, float* mean, float* rstd, hrv_stream_t stream);
int hrv_instnorm_apply_nhwc_f32(const float* x, int32_t N, int32_t H, int32_t W, int32_t C, int32_t cstride,
                                int32_t coff, const float* mean, const float* rstd, int32_t act, float act_slope,
                                float* out, int32_t out_cstride, int32_t out_coff, hrv_stream_t stream);
int hrv_avgpool3x3s2_nhwc_f32(const float* x, int32_t N, int32_t H, int32_t W, int32_t C, int32_t cstride,
                              int32_t coff, float* out, int32_t out_cstride, int32_t out_coff,
                              hrv_stream_t stream);

/* ------------------------------------------------------------------------
 * Parse-map glue between the two networks (test_generator.py:161-217;
 * train_generator.py:217-275).
 *   mul_channel : fake_segmap[:,3] *= warped_cm (:167-176; binarize=1 is the
 *                 'detach' composition: (cm > 0.5))
 *   gauss_blur  : tgm.image.GaussianBlur((k,k),(s,s)) = depthwise conv with zero
 *                 padding, run as two separable passes; `taps_host` are the k
 *                 normalised 1-D weights (HOST pointer, copied into the launch);
 *                 `tmp` is a scratch tensor of the input's size (:179)
 *   parse_argmax: argmax(dim=1) (first maximum wins) -> int64 labels (nullable)
 *                 + the 13->7 merged one-hot map, NHWC [.,8] (:180-203)
 *   occlusion   : remove_overlap(softmax(gauss), cm) and cloth compositing
 *                 (:19-24,214-216); cm lives in channel cm_ch of an NHWC tensor
 * ---------------------------------------------------------------------- */
int hrv_mul_channel_nhwc_f32(float* x, int32_t x_cstride, int32_t ch, const float* m, int32_t m_cstride,
                             int32_t m_ch, int32_t binarize, int64_t npix, hrv_stream_t stream);
int hrv_gauss_blur_nhwc_f32(const float* in, int32_t N, int32_t H, int32_t W, int32_t C, int32_t cstride,
                            const float* taps_host, int32_t ksize, float* tmp, float* out, hrv_stream_t stream);
int hrv_parse_argmax_nhwc_f32(const float* g, int32_t cstride, int32_t nclass, int64_t npix, int64_t* labels,
                              float* parse7, int32_t parse_cstride, hrv_stream_t stream);
/* F.interpolate(size=(Ho,Wo), mode='bilinear'|'nearest') on contiguous NCHW planes
 * (planes = N*C): the input pre-processing of test_generator.py:144-150. */
int hrv_resize_nchw_f32(const float* in, int32_t planes, int32_t H, int32_t W, int32_t Ho, int32_t Wo,
                        int32_t nearest, float* out, hrv_stream_t stream);
int hrv_occlusion_nhwc_f32(const float* g, int32_t g_cstride, int32_t nclass, float* cloth, int32_t cloth_cstride,
                           float* cm, int32_t cm_cstride, int32_t cm_ch, int64_t npix, hrv_stream_t stream);

/* ------------------------------------------------------------------------
 * Layout converters at the module boundary (the reference's tensors are NCHW).
 * NCHW -> NHWC writes channels [out_coff, out_coff + C) of every pixel and, when the caller asks for it, ``zero_tail``
 * further channels [out_coff + C, out_coff + C + zero_tail) as zeros (the pad channels of a tensor the caller owns
 * whole: the conv engine requires zero pads, and this saves a fill of the whole tensor).  zero_tail == 0 touches nothing
 * outside the slice -- the form to use when ``out`` is one slice of a wider concatenation buffer.
 * ---------------------------------------------------------------------- */
int hrv_nchw_to_nhwc_f32(const float* in, int32_t N, int32_t C, int32_t H, int32_t W, float* out,
                         int32_t out_cstride, int32_t out_coff, int32_t zero_tail, hrv_stream_t stream);
int hrv_nhwc_to_nchw_f32(const float* in, int32_t in_cstride, int32_t in_coff, int32_t N, int32_t C,
                         int32_t H, int32_t W, float* out, hrv_stream_t stream);
/* Space-to-depth by 2 and its inverse over NHWC fp32: out[n][y/2][x/2][((y&1)*2 + (x&1))*C + c] = in[n][y][x][c].  Host side
 * (gen_train.S2DConv): PatchGAN's first convolution -- 4x4, stride 2, pad 2 over 10 channels (network_generator.py
 * NLayerDiscriminator model0) -- runs as a 2x2 stride-1 pad-1 convolution over the 4C-channel tensor. */
int hrv_space_to_depth2_nhwc_f32(const float* in, int32_t N, int32_t H, int32_t W, int32_t C, int32_t in_cstride,
                                 int32_t in_coff, float* out, hrv_stream_t stream);
/* Stride-1 convolutions with ONE output channel (K <= 4): PatchGAN's last layer -- Conv2d(nf, 1, kernel_size=4, stride=1,
 * padding=2) of network_generator.py NLayerDiscriminator and networks.py:389-393 -- forward, data gradient (+ an optional
 * tensor added to dx) and weight / bias gradient as dot-product / outer-product / reduction kernels (the implicit-GEMM engine
 * pads the single column to a 64-column tile).  fp32 FMAs; round_bf16: both operands rounded to bf16 first (the arithmetic
 * of the bf16 matrix-core engine in mixed-precision training).  y is the forward OUTPUT and the backward's dY:
 * [N][H+2*pad-K+1][W+2*pad-K+1][y_cstride], channel y_coff. */
typedef struct {
  const float* x; int32_t N, H, W, C, x_cstride, x_coff;
  const float* w_oihw;             /* [1][C][K][K] */
  const float* sigma; float wscale; /* weights are multiplied by wscale / sigma[0] (sigma may be null) */
  const float* bias;               /* [1] or null (forward) */
  int32_t K, pad;
  float* y; int32_t y_cstride, y_coff;
  float* dx; int32_t dx_cstride, dx_coff;            /* data gradient output [N][H][W][dx_cstride] */
  const float* add; int32_t add_cstride, add_coff;   /* optional: dx = dgrad + add */
  float* workspace;                /* weight gradient: hrv_conv_cout1_wgrad_slabs() * (C*K*K + 1) floats */
  int32_t round_bf16;
} hrv_conv_cout1_t;
int hrv_conv_cout1_fwd_f32(const hrv_conv_cout1_t* d, hrv_stream_t stream);
int hrv_conv_cout1_dgrad_f32(const hrv_conv_cout1_t* d, hrv_stream_t stream);
int32_t hrv_conv_cout1_wgrad_slabs(int32_t N, int32_t Ho, int32_t Wo);
int hrv_conv_cout1_wgrad_f32(const hrv_conv_cout1_t* d, float* dw, int32_t accumulate, float* dbias, int32_t dbias_accumulate,
                             hrv_stream_t stream);
/* cat((a, b), 1) for a NHWC (channels [a_coff, a_coff+Ca) of a_cstride) and b NCHW [N,Cb,H,W], written NHWC
 * [N,H,W,out_cstride] with zeroed pad channels: the PatchGAN input of train_generator.py:283-284. */
int hrv_concat_nhwc_nchw_f32(const float* a, int32_t Ca, int32_t a_cstride, int32_t a_coff, const float* b, int32_t Cb, int32_t N,
                             int32_t H, int32_t W, float* out, int32_t out_cstride, hrv_stream_t stream);
int hrv_depth_to_space2_nhwc_f32(const float* in, int32_t N, int32_t H, int32_t W, int32_t C, float* out, hrv_stream_t stream);
/* fp32 NCHW (module boundary) <-> bf16 NHWC (inside the bf16 generator path) */
int hrv_nchw_f32_to_nhwc_bf16(const float* in, int32_t N, int32_t C, int32_t H, int32_t W, uint16_t* out,
                              int32_t out_cstride, int32_t out_coff, int32_t zero_tail, hrv_stream_t stream);
int hrv_nhwc_bf16_to_nchw_f32(const uint16_t* in, int32_t in_cstride, int32_t in_coff, int32_t N, int32_t C,
                              int32_t H, int32_t W, float* out, hrv_stream_t stream);

/* ------------------------------------------------------------------------
 * Bilinear resize, align_corners=False (F.interpolate / nn.Upsample bilinear:
 * networks.py:130-133,150,181; test_generator.py:144-150,179,207), NHWC, with
 * an optional fused addend: out = resize(in) + addend   (networks.py:130-131).
 * rh, rw are the source-index ratios: 1/scale_factor, or in/out for size=.
 * ---------------------------------------------------------------------- */
int hrv_resize_bilinear_nhwc_f32(const float* in, int32_t N, int32_t H, int32_t W, int32_t C,
                                 int32_t in_cstride, int32_t in_coff, int32_t Ho, int32_t Wo, float rh,
                                 float rw, const float* addend, int32_t add_cstride, int32_t add_coff,
                                 float* out, int32_t out_cstride, int32_t out_coff, hrv_stream_t stream);

/* ------------------------------------------------------------------------
 * Appearance-flow warp: the fused form of
 *     flow  = F.interpolate(flow_prev, bilinear)            networks.py:133,150
 *     fnorm = flow / (norm_x, norm_y)                        networks.py:134,151
 *     grid  = make_grid(N, Ho, Wo) (= linspace(-1,1))        networks.py:161-168
 *     out   = F.grid_sample(src, fnorm + grid, 'bilinear',
 *                           padding_mode='border',
 *                           align_corners=False)             networks.py:135,152
 * (also test_generator.py:206-213 with size= resize and the hard-coded norms).
 * flow_prev is the reference's own [N,fh,fw,2] (x,y) layout.  flow_up
 * (optional) receives the un-normalised upsampled flow [N,Ho,Wo,2].
 * ---------------------------------------------------------------------- */
typedef struct hrv_flow_warp {
  const float* src;
  int32_t N, H, W, C; /* sampled tensor extent (C multiple of 4)             */
  int32_t src_cstride, src_coff;
  const float* flow;
  int32_t fh, fw;
  int32_t Ho, Wo;
  float rh, rw;
  float norm_x, norm_y;
  float* out;
  int32_t out_cstride, out_coff;
  float* flow_up; /* or NULL */
} hrv_flow_warp_t;
int hrv_flow_warp_nhwc_f32(const hrv_flow_warp_t* d, hrv_stream_t stream);

/* Tap expansion of a narrow NHWC tensor (im2col in 16-byte groups): out[n,h,w, t*G + g] = group g of the
 * source pixel under tap t of a stride-1 'same' KHxKW window (zero outside the image); the source may be
 * 2^down_shift larger (nearest down-sampling, F.interpolate(segmap, 'nearest'), network_generator.py:112).
 * With it the 7-channel conv_shared 3x3 of every SPADENorm (network_generator.py:97,113) runs as a 1x1
 * convolution over KH*KW*G*16 dense bytes per pixel.  Strides / offsets in 16-byte units; any element type. */
int hrv_tap_expand_nhwc(const void* src, int32_t N, int32_t H, int32_t W, int32_t group16_per_pixel,
                        int32_t src_stride16, int32_t src_off16, int32_t down_shift, int32_t KH, int32_t KW,
                        int32_t pad, void* out, hrv_stream_t stream);

/* ------------------------------------------------------------------------
 * Training of the condition generator (train_condition.py:113-286, tocg.train()).
 *
 * BatchNorm2d with batch statistics (networks.py:171-198 under train_condition.py:116):
 *   bn_finalize : folds the per-sample (mean, rstd) of hrv_instnorm_stats_nhwc_f32
 *                 (computed with eps_in) into batch mean / rstd (biased variance over
 *                 N*H*W, eps), the affine y = scale*x + shift (scale = gamma*rstd) and
 *                 nn.BatchNorm2d's running-statistics update (momentum, unbiased variance);
 *                 running_* may be NULL.  Writes entries c < C only.
 *   affine_act  : out = act(x*scale[c] + shift[c] (+ residual))   (BN apply + ReLU + skip)
 *   bn_bwd      : dx (+)= scale*(dy - mean(dy) - xhat*mean(dy*xhat)), dgamma (+)= sum dy*xhat,
 *                 dbeta (+)= sum dy; x is the convolution output the statistics were taken
 *                 of.  mean / rstd / scale hold ceil4(C) floats (zero in the pad channels);
 *                 workspace: hrv_bn_bwd_workspace_elems(C) floats.  Deterministic.
 * ---------------------------------------------------------------------- */
int hrv_bn_finalize_f32(const float* mean_nc, const float* rstd_nc, int32_t N, int32_t C, int32_t nc_stride,
                        float eps_in, int64_t HW, const float* gamma, const float* beta, float eps, float momentum,
                        float* running_mean, float* running_var, float* mean, float* rstd, float* scale, float* shift,
                        hrv_stream_t stream);
int hrv_affine_act_nhwc_f32(const float* x, int32_t x_cstride, int32_t x_coff, int32_t C, int64_t npix,
                            const float* scale, const float* shift, const float* residual, int32_t res_cstride,
                            int32_t res_coff, int32_t act, float act_slope, float* out, int32_t out_cstride,
                            int32_t out_coff, hrv_stream_t stream);
int64_t hrv_bn_bwd_workspace_elems(int32_t C);
int hrv_bn_bwd_nhwc_f32(const float* dy, int32_t dy_cstride, int32_t dy_coff, const float* x, int32_t x_cstride,
                        int32_t x_coff, int32_t C, int64_t npix, const float* mean, const float* rstd, const float* scale,
                        float* workspace, float* dx, int32_t dx_cstride, int32_t dx_coff, int32_t dx_accumulate,
                        float* dgamma, float* dbeta, int32_t dgb_accumulate, hrv_stream_t stream);
/* Adjoint of hrv_resize_bilinear_nhwc_f32 (F.interpolate / nn.Upsample backward,
 * networks.py:130-133,150,181; train_condition.py:242): dx[N,H,W,C] (+)= R^T dy[N,Ho,Wo,C],
 * gather form (deterministic).  Any C (an NCHW tensor is the C = 1 case with N*C planes). */
int hrv_resize_bilinear_bwd_nhwc_f32(const float* dy, int32_t N, int32_t Ho, int32_t Wo, int32_t C, int32_t dy_cstride,
                                     int32_t dy_coff, float rh, float rw, float* dx, int32_t H, int32_t W,
                                     int32_t dx_cstride, int32_t dx_coff, int32_t accumulate, hrv_stream_t stream);
/* F.interpolate(mode='nearest') over planes [planes][H][W] -> [planes][Ho][Wo] (an NCHW tensor: planes = N*C) and its adjoint
 * (gather form, deterministic): the inter-flow loss of train_condition.py:242 under --upsample nearest.  torch's legacy nearest:
 * src = min(floorf(dst * (float)in / out), in - 1). */
int hrv_resize_nearest_nchw_f32(const float* in, int32_t planes, int32_t H, int32_t W, int32_t Ho, int32_t Wo, float* out,
                                hrv_stream_t stream);
int hrv_resize_nearest_nchw_bwd_f32(const float* dout, int32_t planes, int32_t Ho, int32_t Wo, int32_t H, int32_t W, float* dx,
                                    hrv_stream_t stream);
/* The same nearest selection over NHWC activations (any C, channel slices), out = nearest(in) (+ addend), and its adjoint
 * dx (+)= N^T dout: ConditionGenerator.forward(upsample='nearest') -- networks.py:130-131 (T = up(T) + conv1x1(E)), :133 / :150 (the
 * flow upsampled by 2 in front of the warp; the warp kernel then reads it at ratio 1). */
int hrv_resize_nearest_nhwc_f32(const float* in, int32_t N, int32_t H, int32_t W, int32_t C, int32_t in_cstride, int32_t in_coff,
                                int32_t Ho, int32_t Wo, const float* addend, int32_t add_cstride, int32_t add_coff, float* out,
                                int32_t out_cstride, int32_t out_coff, hrv_stream_t stream);
int hrv_resize_nearest_bwd_nhwc_f32(const float* dout, int32_t N, int32_t Ho, int32_t Wo, int32_t C, int32_t d_cstride, int32_t d_coff,
                                    float* dx, int32_t H, int32_t W, int32_t dx_cstride, int32_t dx_coff, int32_t accumulate,
                                    hrv_stream_t stream);
/* Adjoint of hrv_flow_warp_nhwc_f32 (F.grid_sample backward, networks.py:135,152): given the
 * saved un-normalised flow at the output resolution (flow_up) and dout [N,Ho,Wo,C]:
 *   dsrc  [N,H,W,C]   += scatter of the four bilinear taps (fp32 atomics; zero-init or an
 *                        existing accumulator; may be NULL)
 *   dflow [N,Ho,Wo,2] (+)= d/d flow_up (coordinate gradient x W/2 (H/2) / norm; zero where the
 *                        sample position was clamped to the border, as torch does; may be NULL). */
typedef struct hrv_flow_warp_bwd {
  const float* src;
  int32_t N, H, W, C;
  int32_t src_cstride, src_coff;
  const float* flow_up;
  int32_t Ho, Wo;
  float norm_x, norm_y;
  const float* dout;
  int32_t dout_cstride, dout_coff;
  float* dsrc;
  int32_t dsrc_cstride, dsrc_coff;
  float* dflow;
  int32_t dflow_accumulate;
  int32_t _pad;
} hrv_flow_warp_bwd_t;
int hrv_flow_warp_bwd_nhwc_f32(const hrv_flow_warp_bwd_t* d, hrv_stream_t stream);
/* F.grid_sample(input NCHW, grid [N,Ho,Wo,2], 'bilinear', padding_mode='border',
 * align_corners=False) with an explicit grid, and its backward (train_condition.py:243-245;
 * train_generator.py:237-238).  din (+= atomics, zero-init) and dgrid may each be NULL. */
int hrv_grid_sample_nchw_f32(const float* in, int32_t N, int32_t C, int32_t H, int32_t W, const float* grid, int32_t Ho,
                             int32_t Wo, float* out, hrv_stream_t stream);
int hrv_grid_sample_nchw_bwd_f32(const float* in, int32_t N, int32_t C, int32_t H, int32_t W, const float* grid,
                                 int32_t Ho, int32_t Wo, const float* dout, float* din, float* dgrid,
                                 hrv_stream_t stream);
/* torch.softmax(x, 1) on NCHW (train_condition.py:175,246,260) and its backward
 * dx = y*(dy - sum_c dy*y); C <= 64. */
int hrv_softmax_nchw_f32(const float* x, int32_t N, int32_t C, int64_t HW, float* y, hrv_stream_t stream);
int hrv_softmax_nchw_bwd_f32(const float* y, const float* dy, int32_t N, int32_t C, int64_t HW, float* dx,
                             hrv_stream_t stream);
/* utils.cross_entropy2d (utils.py:29-42): mean over the valid pixels of logsumexp(x) - x[target]
 * (targets outside [0,C) are ignored like ignore_index=250); loss_out[0] = loss, loss_out[1] =
 * valid-pixel count; grad (optional, NCHW) = gscale*(softmax - onehot) -- the caller folds
 * 1/count into gscale.  workspace: 1024 floats. */
int hrv_cross_entropy_nchw_f32(const float* x, const int64_t* target, int32_t N, int32_t C, int64_t HW, float gscale,
                               float* grad, float* workspace, float* loss_out, hrv_stream_t stream);
/* Adjoint of hrv_tapsum_nhwc_f32 (the 768->2 flow_conv, networks.py:85-92, run as a
 * taps-as-channels 1x1 convolution): dy[q][tap*Cout+co] = dout[q - off(tap)][co]; dy has
 * ceil4(KH*KW*Cout) channels (padding zeroed). */
int hrv_tapsum_bwd_nhwc_f32(const float* dout, int32_t N, int32_t H, int32_t W, int32_t KH, int32_t KW, int32_t pad,
                            int32_t Cout, int32_t dout_cstride, float* dy, int32_t dy_cstride, hrv_stream_t stream);
/* out = x * m elementwise (dense buffers, n % 4 == 0): nn.Dropout(0.5) of the tocg discriminator with the keep
 * mask pre-scaled by 1/(1-p) (networks.py:363-368), forward and backward. */
int hrv_mul_f32(const float* x, const float* m, int64_t n, float* out, hrv_stream_t stream);
/* Flow total variation (train_condition.py:190-199) of one [N,H,W,2] flow:
 * loss = mean|f[:,1:]-f[:,:-1]| + mean|f[:,:,1:]-f[:,:,:-1]|, grad = d loss / d f (optional).
 * workspace: 1024 floats. */
int hrv_tv_loss_f32(const float* flow, int32_t N, int32_t H, int32_t W, float* grad, float* workspace, float* loss_out,
                    hrv_stream_t stream);

/* SPADENorm's conv_gamma || conv_beta (network_generator.py:117-121: two 3x3 convolutions 128 -> C over
 * actv = ReLU(conv_shared(seg))) fused with `normalized * (1 + gamma) + beta` (+ LeakyReLU, :170-171), and the data
 * gradient of that pair, d(actv) = conv^T([dgamma | dbeta]) * relu'(actv), on a dedicated kernel (csrc/spade_gb.hip):
 * one persistent 8-wave block per CU owns a 16x16-pixel tile and ALL of the layer's columns (no padded columns, the
 * halo patch loaded once), weights streamed in MFMA-fragment order through a 3-stage LDS ring.
 *   mode 0 (forward):  src = actv, bf16 NHWC [.,hid]; columns = (gamma | beta) of the C norm channels;
 *                      out = act((x + z*noise_scale - mean) * rstd * (1 + gamma + bias_gamma) + beta + bias_beta);
 *                      g1p (optional) receives (1 + gamma), dense [pixels][stat_stride], fp32 or bf16.
 *   mode 1 (data gradient): src = [dgamma (Cp) | dbeta (Cp)], bf16 NHWC; out[.., hid] = conv^T * (mask > 0 ? 1 :
 *                      act_slope), mask = actv (bf16) or NULL.
 * Shapes served: hid == 128, C % 32 in {0, 16} (32, 64, 80, 128, 144, 272, 528, ...; not 48: one pair + tail),
 * at least 256 tiles (hrv_spade_gb_supported); everything else keeps hrv_conv2d_nhwc_bf16's tiles. */
typedef struct hrv_spade_gb {
  int32_t mode, N, H, W;
  const void* src; int32_t src_cstride, src_coff;
  const void* w_packed;        /* hrv_spade_gb_pack_dev(mode, ...)                                   */
  int32_t C, Cp, hid;          /* norm channels, their padded count (layout of [dgamma|dbeta]), 128  */
  int32_t x_f32;               /* forward: x is fp32 (else bf16)                                     */
  const void* x; int32_t x_cstride, x_coff;
  int32_t stat_stride;         /* floats per image in mean / rstd, channels per pixel in g1p        */
  int32_t g1p_bf16;
  const float* mean; const float* rstd; const float* noise_z; const float* noise_scale;
  const float* bias_gamma; const float* bias_beta;
  void* g1p;
  int32_t act; float act_slope;
  void* out; int32_t out_cstride, out_coff, out_f32, _pad;
  const void* mask; int32_t mask_cstride, mask_coff;
} hrv_spade_gb_t;
int64_t hrv_spade_gb_packed_bytes(int32_t mode, int32_t C, int32_t Cp, int32_t hid);   /* -1: shape not served */
int hrv_spade_gb_supported(int32_t mode, int32_t C, int32_t Cp, int32_t hid, int32_t N, int32_t H, int32_t W);
/* fp32 conv_gamma.weight / conv_beta.weight [C][hid][3][3] (device) -> the bf16 fragment-order stream of `mode` */
int hrv_spade_gb_pack_dev(int32_t mode, const float* w_gamma, const float* w_beta, int32_t C, int32_t Cp, int32_t hid,
                          void* out, hrv_stream_t stream);
int hrv_spade_gb_bf16(const hrv_spade_gb_t* d, hrv_stream_t stream);

/* SPADENorm forward fused end to end (spade_fused.hip; network_generator.py:93-121): replaces, for one norm,
 *     actv  = F.relu(conv_shared(F.interpolate(segmap, size, 'nearest')))            (:115-116)
 *     out   = act(param_free_norm(x + noise) * (1 + conv_gamma(actv)) + conv_beta(actv))   (:104-110, :117-121)
 * `actv` is computed inside the kernel from the label patch (two blocks per CU; it never travels through HBM unless the
 * caller asks for it).
 *   seg      bf16 NHWC [N, H << seg_shift, W << seg_shift, 8] (label_nc <= 8 real channels, the rest zero): the level's
 *            nearest-downsampled label map is read in place (pixel (y << seg_shift, x << seg_shift));
 *   w_packed hrv_spade_fused_pack_dev(conv_shared.weight / .bias, conv_gamma.weight, conv_beta.weight);
 *   x        fp32 (x_f32 = 1) or bf16 NHWC slice; mean / rstd [N][C]; noise_z [N][W][H] + noise_scale [C] or both NULL;
 *   out      bf16 NHWC slice; g1p (optional) <- 1 + gamma, bf16 dense [N,H,W,C];
 *   actv     (optional) <- ReLU(conv_shared(seg)), bf16 NHWC slice of 128 channels (the training forward keeps it for
 *            the backward: ReLU mask of the data gradient, operand of the gamma|beta weight gradient).
 * Shapes served: hid == 128, C % 32 in {0, 16}, C >= 32, at least two tiles per CU (hrv_spade_fused_supported). */
typedef struct hrv_spade_fused {
  int32_t N, H, W, C;
  const void* seg; int32_t seg_H, seg_W, seg_shift, x_f32;
  const void* w_packed;
  const void* x; int32_t x_cstride, x_coff;
  const float* mean; const float* rstd; const float* noise_z; const float* noise_scale;
  const float* bias_gamma; const float* bias_beta;
  void* g1p;
  int32_t act; float act_slope;
  void* out; int32_t out_cstride, out_coff;
  void* actv; int32_t actv_cstride, actv_coff;
  /* x = cat(nearest_up2(lo), hi), never materialised (x_up_channels > 0, a multiple of 16; x fp32): channels [0, x_up_channels) are read
   * from `x` = lo [N][H/2][W/2][x_cstride] at (y >> 1, x >> 1), the rest from `x2` = hi [N][H][W][x2_cstride] (see hrv_norm_bwd_t) */
  int32_t x_up_channels;
  const void* x2; int32_t x2_cstride, x2_coff;
} hrv_spade_fused_t;
int64_t hrv_spade_fused_packed_bytes(int32_t C);   /* -1: norm width not served */
int hrv_spade_fused_supported(int32_t C, int32_t hid, int32_t label_nc, int32_t N, int32_t H, int32_t W);
/* conv_shared.weight [128][label_nc][3][3] + bias [128], conv_gamma.weight / conv_beta.weight [C][128][3][3] (fp32, device)
 * -> the bf16 fragment-order stream the kernel's LDS ring consumes */
int hrv_spade_fused_pack_dev(const float* w_shared, const float* b_shared, int32_t label_nc, const float* w_gamma,
                             const float* w_beta, int32_t C, void* out, hrv_stream_t stream);
int hrv_spade_fused_bf16(const hrv_spade_fused_t* d, hrv_stream_t stream);

/* 3x3 stride-1 'same' convolution over ONE bf16-stored NHWC source, any Cin / Cout (the source keeps its channels padded to 8 with
 * zeros, `out` / `mask` / `residual` theirs padded to 16 bytes: the pad lanes of `out` receive zeros), two blocks per CU
 * (conv_p2.hip): nn.Conv2d forward (VGG19 of the perceptual loss, networks.py:201-233) and data gradients -- of such a
 * convolution (mode 1) or of the SPADE (conv_gamma, conv_beta) pair over [dgamma | dbeta] (mode 2, network_generator.py:117-118).
 *   out = act(conv + bias[c] [+ residual]) [* (mask > 0 ? 1 : mask_slope)] [+ residual, if res_after_mask], bf16 or fp32 NHWC slice.
 * `Cin` = K (channels of `src`), `Cout` = columns, whatever the mode; w_packed from hrv_conv_p2_pack_dev of the same mode:
 *   mode 0: w = the layer's OIHW weight [Cout][Cin][3][3];
 *   mode 1: w = the FORWARD layer's OIHW weight [Cin][Cout][3][3] (its output channels are this call's K), taps flipped;
 *   mode 2: (w, w2) = [Cin / 2][Cout][3][3] each, K = [w rows | w2 rows], taps flipped.
 * `sigma` (optional device scalar) and `wscale`: packed weight = w * wscale / sigma[0].
 * hrv_conv_p2_supported: shape served and at least 1.5 tiles per CU (HRV_CONV_P2_MIN_TILES_X4: the threshold in quarter-tiles). */
typedef struct hrv_conv_p2 {
  int32_t N, H, W, Cin;
  const void* src; int32_t src_cstride, src_coff, Cout;
  const void* w_packed;
  const float* bias;
  int32_t act; float act_slope;
  const void* mask; int32_t mask_cstride, mask_coff; float mask_slope; int32_t out_f32;
  void* out; int32_t out_cstride, out_coff;
  int32_t res_f32;                                           /* residual dtype: 1 fp32, 0 bf16 */
  const void* residual; int32_t res_cstride, res_coff;       /* optional NHWC slice added BEFORE the activation (SPADEResBlock: x_s + dx) */
  int32_t res_after_mask;                                    /* 1: added behind activation and mask instead (a data gradient that meets
                                                              * another gradient of the same tensor: VGG19's tap gradients) */
} hrv_conv_p2_t;
int64_t hrv_conv_p2_packed_bytes(int32_t Cin, int32_t Cout);   /* -1: shape not served */
int hrv_conv_p2_supported(int32_t Cin, int32_t Cout, int32_t N, int32_t H, int32_t W);
int hrv_conv_p2_pack_dev(int32_t mode, const float* w, const float* w2, int32_t Cin, int32_t Cout, const float* sigma, float wscale,
                         void* out, hrv_stream_t stream);
int hrv_conv_p2_bf16(const hrv_conv_p2_t* d, hrv_stream_t stream);

/* PatchGAN's 4x4 stride-2 pad-2 convolution over a bf16-stored NHWC feature map, and its data gradient, as ONE launch family
 * on the two-blocks-per-CU skeleton of hrv_conv_p2_bf16 (conv_s2.hip; NLayerDiscriminator, network_generator.py:263-272:
 * nn.Conv2d(nf_prev, nf, kernel_size=4, stride=2, padding=2) and autograd's gradient w.r.t. its input).
 *   mode 0  forward: src [N][Hs][Ws][K] -> out [N][Ho = Hs/2+1][Wo = Ws/2+1][cols]; w = the layer's OIHW weight [cols][K][4][4].
 *   mode 1  data gradient: src = dY [N][Hs][Ws][K] (K = the forward layer's OUTPUT channels) -> out = dX [N][Ho][Wo][Cph]
 *           (Hs = Ho/2+1), cols = 4 * Cph (the four output phases are column passes); w = the forward OIHW weight [K][Cph][4][4].
 *   mode 2  a 2x2 stride-1 convolution, pad 1 on top / left only: src = a space-to-depth image [N][Hs][Ws][K] -> out
 *           [N][Ho in {Hs, Hs+1}][Wo ...][cols]; w = [cols][K][2][2] (PatchGAN's model0: 10 channels -> 4 x 12 per cell).
 *   out = act(conv + bias[c] [+ residual]) [* (mask > 0 ? 1 : mask_slope)]; residual / mask: slices shaped like `out`.
 * K % 8 == 0 (mode 0: % 32), cols % 64 == 0, Cph % 32 == 0.  w_packed from hrv_conv_s2_pack_dev of the same (mode, K, cols, Cph). */
typedef struct hrv_conv_s2 {
  int32_t mode;
  int32_t N, Hs, Ws, K;
  const void* src; int32_t src_cstride, src_coff;
  int32_t Ho, Wo, cols, Cph;
  const void* w_packed;
  const float* bias;
  int32_t act; float act_slope;
  void* out; int32_t out_cstride, out_coff, out_f32;
  const void* residual; int32_t res_cstride, res_coff, res_f32;
  const void* mask; int32_t mask_cstride, mask_coff; float mask_slope;
} hrv_conv_s2_t;
#define HRV_S2_SPLIT3 4   /* or-ed into hrv_conv_s2_pack_dev's mode (0 / 2): K = 3 K0 over a source laid out [hi | lo | hi]
                           * (hrv_split3_nhwc_bf16), w has K0 input channels and is packed as [hi(w) | hi(w) | w - hi(w)]: the
                           * convolution sums hi*hi + lo*hi + hi*lo -- fp32 operands to ~16 mantissa bits on the bf16 matrix cores
                           * (the half-resolution PatchGAN scale in D's own step, gen_train._d_f32) */
int64_t hrv_conv_s2_packed_bytes(int32_t mode, int32_t K, int32_t cols);   /* -1: shape not served */
int hrv_conv_s2_supported(int32_t mode, int32_t K, int32_t cols, int32_t Cph, int32_t N, int32_t Ho, int32_t Wo);
int hrv_conv_s2_pack_dev(int32_t mode, const float* w, int32_t K, int32_t cols, int32_t Cph, const float* sigma, float wscale,
                         void* out, hrv_stream_t stream);
/* up to 8 weights in one launch (a PatchGAN pass packs three forward / two data-gradient weights per scale) */
typedef struct hrv_s2_pack_job {
  int32_t mode_flags, K, cols, Cph;
  const float* w; const float* sigma; float wscale; int32_t _pad;
  void* out;
} hrv_s2_pack_job_t;
int hrv_conv_s2_pack_multi_dev(int32_t n, const hrv_s2_pack_job_t* jobs, hrv_stream_t stream);
int hrv_conv_s2_bf16(const hrv_conv_s2_t* d, hrv_stream_t stream);
/* bf16-storage companions of hrv_conv_s2_bf16 (the PatchGAN with bf16-stored feature maps): the space-to-depth image of model0's
 * input written in bf16 (cf. hrv_space_to_depth2_nhwc_f32), InstanceNorm2d(affine=False) + LeakyReLU written in bf16
 * (cf. hrv_instnorm_apply_nhwc_f32; network_generator.py:268-269), x *= s_host * (s_dev ? s_dev[0] : 1) over a bf16 loss gradient
 * (cf. hrv_scale_f32), and out[r][w] = w < W ? in[r][w] : 0 for a dY whose width the quad-staged weight gradient needs padded to 4. */
/* the space-to-depth image of model0's input (cf. hrv_space_to_depth2_nhwc_f32) in bf16 over Hp x Wp cells (>= H/2 x W/2; cells and
 * sub-pixels outside the image are zeros: with a one-cell border model0
 * is a 'same' 2x2 convolution, the shape hrv_conv2d_wgrad_bf16mma_st_nhwc_f32's LDS-DMA kernel serves), optionally as [hi | lo | hi] */
int hrv_space_to_depth2_cells_bf16(const float* in, int32_t N, int32_t H, int32_t W, int32_t C, int32_t in_cstride, int32_t in_coff,
                                   int32_t Hp, int32_t Wp, int32_t split3, uint16_t* out, hrv_stream_t stream);
int hrv_instnorm_apply_nhwc_bf16out(const float* x, int32_t N, int32_t H, int32_t W, int32_t C, int32_t cstride, int32_t coff,
                                    const float* mean, const float* rstd, int32_t act, float act_slope, uint16_t* out,
                                    int32_t out_cstride, int32_t out_coff, hrv_stream_t stream);
int hrv_scale_bf16(uint16_t* x, int64_t n, float s_host, const float* s_dev, hrv_stream_t stream);
/* out[p] = [hi(x[p]) | bf16(x[p] - hi(x[p])) | hi(x[p])] (3 C bf16 channels, dense) of an fp32 NHWC slice: see HRV_S2_SPLIT3 */
int hrv_split3_nhwc_bf16(const float* x, int64_t npix, int32_t C, int32_t cstride, int32_t coff, uint16_t* out, hrv_stream_t stream);
/* 1: hrv_conv2d_wgrad_bf16mma_st_nhwc_f32 serves this 4x4 stride-2 pad-2 layer (bf16-stored dY and X) with the LDS-DMA kernel of
 * wgrad_s2.hip -- any output width; 0: the quad-staged kernel, which needs Wo % 4 == 0 (hrv_pad_width_nhwc_bf16) */
int hrv_conv2d_wgrad_s2_supported(int32_t Cout, int32_t x_C, int32_t x_cstride, int32_t x_coff, int32_t dy_cstride, int32_t dy_coff,
                                  int32_t N, int32_t H, int32_t W);
int hrv_pad_width_nhwc_bf16(const uint16_t* in, int64_t rows, int32_t W, int32_t C, int32_t Wp, uint16_t* out, hrv_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* HRVITON_HIP_H */
