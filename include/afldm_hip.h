/*
 * afldm_hip.h — C ABI of libafldm_hip.so: the MI355X (gfx950) kernels behind the
 * alias-free latent-diffusion denoising path of SingleZombie/AFLDM.
 *
 * The reference has NO native boundary on this path: its operator API is PyTorch
 * nn.Module surgery (afldm/af_modules/af_api.py:70-83) over diffusers modules whose
 * arithmetic is dispatched to cuDNN / cuBLAS / cuFFT / SDPA.  This header is the native
 * layer a maintainer would bind underneath those modules; each entry point names the
 * reference computation (file:line) it replaces.  See INTEGRATION.md for the ctypes
 * binding that afldm_amd ships and the module-level hook a reference maintainer would add.
 *
 * Conventions
 *   - every pointer except `afldm_filter_matrix`'s output is a DEVICE pointer;
 *   - activations are NHWC ("channels last"), contiguous, dtype AFLDM_F32 or AFLDM_BF16;
 *     [B, H*W, C] token tensors for attention are the same memory;
 *   - conv / linear weights are OHWI ([Cout][KH][KW][Cin]) in the activation dtype
 *     (afldm_pack_weight converts from the reference's OIHW fp32 state-dict layout);
 *     biases, GroupNorm affine parameters and statistics are fp32;
 *   - all launches are stream-ordered on `stream` (a hipStream_t), never block, never
 *     allocate; the caller owns every buffer (graph-capturable);
 *   - return 0 on success, <0 = AFLDM_E*; afldm_last_error() gives the message of the last
 *     failure on the calling thread.  No entry point aborts.
 */
#ifndef AFLDM_HIP_H
#define AFLDM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* afldm_stream_t; /* hipStream_t */

enum { AFLDM_F32 = 0, AFLDM_BF16 = 1 };
enum {
  AFLDM_OK = 0,
  AFLDM_ESHAPE = -1,  /* unsupported / inconsistent shape */
  AFLDM_EDTYPE = -2,  /* unknown dtype code */
  AFLDM_EALIGN = -3,  /* pointer or leading dimension not 16-byte aligned */
  AFLDM_ELAUNCH = -4, /* HIP launch error */
  AFLDM_ENULL = -5    /* required pointer is NULL */
};

int afldm_version(void);
const char* afldm_last_error(void);
/* name[] receives gcnArchName; returns CU count (<0 on error). */
int afldm_device_info(char* name, int name_len);

/* ---- filter matrices (host) ------------------------------------------------------------
 * Builds the dense separable form of the reference's FFT-domain ideal filters from the
 * reference's mask rules (create_lpf_rect ideal_lpf.py:12-24, create_recon_rect :38-49):
 *   kind 0: U  [up*N x N], UpsampleRFFT(up)(X) == U X U^T      (ideal_lpf.py:148-158)
 *   kind 1: D  [N/2 x N],  LPF_RFFT(1/2)(Z)[::2,::2] == D Z D^T (ideal_lpf.py:69-93 + af_blocks.py:26)
 *   kind 2: L  [N x N],    LPF_RFFT(1/2)(Z) == L Z L^T           (ideal_lpf.py:69-93)
 * `out` is HOST memory, row-major fp32 (computed in fp64). */
int afldm_filter_matrix(int kind, int N, int up, float* out);

/* ---- layout / dtype plumbing ----------------------------------------------------------- */
/* src NCHW fp32 -> dst NHWC dtype (latent entry, randn_tensor ldm_pipeline.py:82-88) */
int afldm_nchw_to_nhwc(const float* src, void* dst, int B, int C, int H, int W, int dtype,
                       afldm_stream_t stream);
/* src NHWC dtype -> dst NCHW fp32 */
int afldm_nhwc_to_nchw(const void* src, float* dst, int B, int C, int H, int W, int dtype,
                       afldm_stream_t stream);
/* OIHW fp32 (state-dict layout) -> OHWI dtype.  KH*KW == 1 covers nn.Linear [O][I]. */
int afldm_pack_weight(const float* src, void* dst, int O, int I, int KH, int KW, int dtype,
                      afldm_stream_t stream);
/* dtype -> fp32 / fp32 -> dtype flat copies */
int afldm_cast(const void* src, int src_dtype, void* dst, int dst_dtype, size_t n,
               afldm_stream_t stream);

/* ---- timestep embedding ----------------------------------------------------------------
 * diffusers Timesteps / get_timestep_embedding as used by UNet2DModel.time_proj
 * (config configs/ldm/model_unet.json:28-29): out[r, :] = [cos(t w_i) | sin(t w_i)] when
 * flip_sin_to_cos, w_i = exp(-ln(10000) i / (dim/2 - freq_shift)).
 * `t` points to `rows` device floats (the timestep values).  out: [rows, dim] dtype. */
int afldm_timestep_embedding(const float* t, void* out, int rows, int dim, int flip_sin_to_cos,
                             float freq_shift, int dtype, afldm_stream_t stream);

/* y = silu(x), flat.  (TimestepEmbedding.act and ResnetBlock2D.nonlinearity on the 2-D temb:
 * af_blocks.py:20-21 keeps plain SiLU for <4-D tensors.) */
int afldm_silu(const void* x, void* y, size_t n, int dtype, afldm_stream_t stream);

/* ---- GroupNorm ----------------------------------------------------------------------------
 * torch.nn.GroupNorm(G, C, eps) over an NHWC tensor that is the virtual channel-concat of
 * x1 [B,HW,C1] and x2 [B,HW,C2] (x2 may be NULL with C2 = 0): the up-block skip
 * torch.cat([h, skip], 1) is never materialised.
 * Statistics are exchanged as PER-CHANNEL PARTIAL SUMS: stats[B][S][C][2] fp32 = (sum, sum of
 * squares) of channel c of ONE tensor over pixel-split s.  Producers: afldm_gn_stats (S =
 * afldm_gn_stats_splits(HW)) and afldm_conv2d, which emits the statistics of its output from the
 * GEMM epilogue (stats_out, S = afldm_conv2d_stats_splits(args)) so that most GroupNorms cost no
 * extra pass over the tensor.  Consumers (afldm_gn_apply, afldm_af_act, afldm_gn_table) take the
 * statistics of x1 and of x2 separately, add the partials of a group's channels in a fixed order
 * and finish mean / rstd in fp64: no finalize launch, no atomics, bit-reproducible; one tensor's
 * statistics serve both the next block and, later, a skip concatenation whose groups straddle
 * the two tensors. */
int afldm_gn_stats_splits(int HW);
int afldm_gn_stats(const void* x, int C, float* stats, int B, int HW, int dtype, afldm_stream_t stream);
/* y = act((x - mean) * rstd * gamma + beta); act: 0 none (Attention.group_norm),
 * 1 SiLU (conv_norm_out + conv_act, which make_af_unet does NOT wrap: af_api.py:70-83). */
int afldm_gn_apply(const void* x1, int C1, const void* x2, int C2, const float* stats1, int S1,
                   const float* stats2, int S2, const float* gamma, const float* beta, void* y, int B,
                   int HW, int G, float eps, int act, int dtype, afldm_stream_t stream);

/* Fold S_in row splits of per-channel partial sums [B][S_in][C][2] into [B][S_out][C][2] (S_in % S_out == 0):
 * the GEMM epilogue emits one split per 128-pixel tile, 512 per sample on the AF-VAE's 256^2 planes. */
int afldm_gn_fold(const float* stats_in, int S_in, float* stats_out, int S_out, int B, int C,
                  afldm_stream_t stream);

/* diagnostic: device buffer [workgroups][4 waves][2 items][10] of 64-bit shader-clock stamps that later afldm_af_act
 * launches on the plane kernel (N = 16 / 32) fill (phase boundaries of a workgroup's first two items; csrc/af.hip),
 * NULL = off (the default). */
int afldm_af_act_trace(void* buf);
/* ---- alias-free operators -----------------------------------------------------------------
 * afldm_af_act: [GroupNorm-apply ->] WarpedNonlinearity(SiLU) (af_blocks.py:19-28):
 *   y = D silu(U xn U^T) D^T per (b, c) plane,  xn = GN-applied x when stats1 != NULL
 *   (stats1 / stats2 = per-channel partial sums of x1 / x2, S1 / S2 their split counts).
 * x = virtual concat of x1/x2 as above, [B,N,N,C]; y [B,N,N,C].  N in {2,4,8,16,32}.
 * U: [2N x N], D: [N x 2N] device fp32 matrices from afldm_filter_matrix(0,N,2) / (1,2N,.). */
int afldm_af_act(const void* x1, int C1, const void* x2, int C2, const float* stats1, int S1,
                 const float* stats2, int S2, const float* gamma, const float* beta, int G, float eps,
                 const float* U, const float* D, const void* packed, void* y, int B, int N, int dtype,
                 afldm_stream_t stream);
/* The same with x1 and / or y in 8-channel blocks [B][C/8][N][N][8] (x_layout / y_layout = 1; afldm_conv_args.x_layout): N = 16 / 32,
 * bf16.  An item of the kernel - 8 (16) channels of one sample - is then one (two) contiguous 16 KB run instead of N^2 16-byte
 * pieces at a stride of 2 C bytes; the 3x3 convolutions of the block read / write the layout directly (afldm_conv2d_c8_ok). */
int afldm_af_act_c8(const void* x1, int C1, const void* x2, int C2, const float* stats1, int S1, const float* stats2,
                    int S2, const float* gamma, const float* beta, int G, float eps, const float* U, const float* D,
                    const void* packed, void* y, int B, int N, int dtype, int x_layout, int y_layout,
                    afldm_stream_t stream);
/* afldm_af_act on 2 x 2 planes, result stored ONCE per plane: y [B][C1+C2].  The reference's LPF_RFFT mask for the 4 x 4
 * upsampled plane is lpf(4) = [1,0,0,0] (ideal_lpf.py:17-21, M % 4 == 0 zeroes the Nyquist pair), so LPF_RFFT + [::2,::2] of
 * WarpedNonlinearity (af_blocks.py:19-28) leaves mean(silu(up(x))) in all four pixels: the activated 2 x 2 tensor is
 * plane-constant, bit for bit (the four outputs are one fmaf chain over equal operands).  A 3x3 'same' convolution of such
 * a tensor is a dense layer over Cin (not 4 Cin) columns whose weight is the sum of the taps that see each input pixel
 * (host: afldm_amd/models/blocks.py packed_conv_dense2x2_const) - a quarter of the weight bytes of the 2x2 level. */
int afldm_af_act_const2(const void* x1, int C1, const void* x2, int C2, const float* stats1, int S1, const float* stats2,
                        int S2, const float* gamma, const float* beta, int G, float eps, const float* U, const float* D,
                        void* y, int B, int dtype, afldm_stream_t stream);
/* conv1 -> (+ time embedding) -> norm2 -> WarpedNonlinearity of diffusers ResnetBlock2D.forward on 2 x 2 planes as ONE launch, for a
 * plane-constant input (the block's first WarpedNonlinearity, afldm_af_act_const2): a [B][Cin] is that activation, w [4 Cout][Cin] the
 * tap-summed dense form of conv1 with its rows ordered channel-major (row = 4 n + pixel, pixel = 2 oh + ow), bias [Cout], temb the
 * time_emb_proj row(s) (temb_stride elements between samples, 0 = one row for all), gamma / beta / G / eps norm2, U [4x2] / D [2x4]
 * the N = 2 filter matrices; y [B][Cout] = the plane-constant second activation (input of conv2's dense form).  A workgroup owns
 * one GroupNorm group x 16 samples, so statistics, normalisation and activation need no exchange (csrc/dense2.hip).  Rounding points
 * as afldm_conv2d + afldm_af_act_const2: the convolution output is rounded to `dtype` once before it is normalised.
 * _supported: 1 when there is a kernel (Cout / G in {8, 12, 24}, Cin a multiple of 8 MFMA K steps). */
int afldm_conv2x2_const_norm_act_supported(int Cin, int Cout, int G, int dtype);
int afldm_conv2x2_const_norm_act(const void* a, const void* w, const float* bias, const void* temb, int temb_stride,
                                 const float* gamma, const float* beta, int G, float eps, const float* U, const float* D, void* y,
                                 int B, int Cin, int Cout, int dtype, afldm_stream_t stream);
/* A convolution's split-K slabs straight into the GroupNorm that follows it, on the 2x2 / 4x4 planes (diffusers
 * resnet.py / attention_processor.py): the slabs a deferred afldm_conv2d left in its workspace ([nslab][B*N*N][C]
 * fp32) are summed in slab order, + bias[c] + temb[b*temb_stride + c] + residual, rounded to `dtype` (the value the
 * two-launch path stores; written to y_raw when other consumers need it), GroupNorm-ed with statistics formed
 * inside the launch (one workgroup holds whole groups of one sample: C / G channels x N*N pixels each, fp64
 * finish) and, with act = 1, passed through y = D silu(U xn U^T) D^T per plane:
 *   act = 1: conv1(...) + temb -> norm2 -> the WarpedNonlinearity af_api.py:70-83 wraps around `nonlinearity`
 *   act = 0: conv2(...) + shortcut -> Attention.group_norm of the attention block that follows the resnet
 *   act = 2: as act = 1 on 2 x 2 planes with the plane-constant result stored once, y [B][C] (see afldm_af_act_const2)
 * Replaces the reduction launch (and, for act = 1, the stored intermediate and its re-read).  N in {2, 4};
 * bias / temb / residual / y_raw may be NULL; temb_stride 0 = one row for all samples. */
int afldm_af_act_slabs(const float* slabs, int nslab, const float* bias, const void* temb, int temb_stride,
                       const void* residual, void* y_raw, const float* gamma, const float* beta, int G, float eps,
                       int act, const float* U, const float* D, void* y, int B, int C, int N, int dtype,
                       afldm_stream_t stream);
/* The tail of UNet2DModel.forward in one launch (bf16, 32x32 planes, C in {64, 128, 192}, Cout <= 4):
 * conv_norm_out (GroupNorm from the per-channel partial sums stats[B][S][C][2] of x) -> conv_act (plain SiLU:
 * af_api.py:70-83 leaves it unwrapped) -> conv_out (3x3 'same', weights packed OHWI [Cout][3][3][C], fp32 bias).
 * x NHWC [B,N,N,C]; y NHWC [B,N,N,Cout].  The activated tensor is never stored. */
int afldm_conv_out_fused(const void* x, const float* stats, int S, const float* gamma, const float* beta, int G,
                         float eps, const void* w, const float* bias, void* y, int B, int N, int C, int Cout,
                         int dtype, afldm_stream_t stream);
/* y = M x M^T per plane in ONE kernel (MFMA, no fp32 intermediate through HBM) for the two large
 * resampling sites of the UNet: AliasFreeUpsample2D at 16 -> 32 (M = U, af_blocks.py:92-93) and
 * AliasFreeDownsample2D at 32 -> 16 (M = D, af_blocks.py:149-150).  stats_out (optional):
 * per-channel GroupNorm partial sums of y, [B][1][C][2] (S = 1). */
int afldm_af_resample_plane(const void* x, const float* M, void* y, float* stats_out, int B, int N,
                            int C, int R, int dtype, afldm_stream_t stream);
/* One-time packing of U / D into the kernel's LDS image for the MFMA plane sizes (N = 16, 32):
 * `packed` = device buffer of afldm_af_pack_bytes(N, dtype) bytes, passed to afldm_af_act. */
size_t afldm_af_pack_bytes(int N, int dtype);
int afldm_af_pack(const float* U, const float* D, int N, int dtype, void* packed,
                  afldm_stream_t stream);
/* UpsampleRFFT(2) of AliasFreeUpsample2D (af_blocks.py:92-93): [B,N,N,C] -> [B,2N,2N,C].
 * workspace: device fp32 scratch of B*2N*N*C floats (row pass result). */
int afldm_af_up2(const void* x, const float* U, void* y, float* workspace, int B, int N, int C,
                 int dtype, afldm_stream_t stream);
/* LPF_RFFT(1/2) + [::2,::2] of AliasFreeDownsample2D (af_blocks.py:149-150):
 * [B,N,N,C] -> [B,N/2,N/2,C];  D: [N/2 x N];  workspace: B*(N/2)*N*C floats.
 * stats_out (optional, N a power of two <= 32, C % 4 == 0): per-channel GroupNorm partial sums of y with one
 * split per output row, [B][N/2][C][2]. */
int afldm_af_lpf_down2(const void* x, const float* D, void* y, float* workspace, float* stats_out, int B,
                       int N, int C, int dtype, afldm_stream_t stream);

/* Generic separable product y = M x M^T per (b, c) plane: [B,N,N,C] -> [B,R,R,C], M: [R x N]
 * device fp32.  Serves UpsampleRFFT(up) for any `up` (ImageShifter('ideal', 8):
 * shifters.py:163-170) and the same-size LPF_RFFT (kind 2 matrix).  workspace: B*R*N*C floats. */
int afldm_af_resample(const void* x, const float* M, void* y, float* workspace, int B, int N, int C,
                      int R, int dtype, afldm_stream_t stream);

/* y = Mh x Mw^T per plane with different matrices along H and W ([R][N] each): the phase-ramp shift
 * of shifters.py:103-132 (fourier_shift_batch) in dense circulant form.  workspace: B*R*N*C floats. */
int afldm_af_resample_hw(const void* x, const float* Mh, const float* Mw, void* y, float* workspace,
                         int B, int N, int C, int R, int dtype, afldm_stream_t stream);

/* ---- large-plane separable passes (alias-free VAE, planes 64^2 .. 256^2) ----------------------
 * y[line][r] = act(sum_k M[r][k] xn[line][k])            or, with M2 (chained in registers):
 * y[line][r2] = sum_r M2[r2][r] silu(sum_k M[r][k] xn[line][k])
 * A line is a strided vector: element k of line (outer, inner) sits at
 *   x + outer * in_outer_stride + inner + k * in_k_stride      (inner_count lines are contiguous).
 * xn = x * scale[b][c] + shift[b][c] when gn_table != NULL (b = outer / outer_per_sample,
 * c = inner % C; table from afldm_gn_table).  Used for UpsampleRFFT / LPF_RFFT on big planes and,
 * in three launches (up-H; up-W -> SiLU -> down-W chained; down-H), for WarpedNonlinearity when the
 * 2N x 2N plane does not fit LDS (af_blocks.py:19-28 at N = 64, 128 in the AF-VAE). */
typedef struct {
  const void* x;
  void* y;
  const float* M;        /* [R][K] device fp32 */
  const float* M2;       /* [R2][R] device fp32, or NULL */
  const float* gn_table; /* [B][C][2] device fp32, or NULL */
  long long outer_count, inner_count;
  long long in_outer_stride, in_k_stride, out_outer_stride, out_k_stride; /* elements */
  int K, R, R2;
  int C, outer_per_sample;
  int act;   /* 1: SiLU after M (only when R2 == 0) */
  int dtype;
  /* 1 (chained passes, R == 2 K): the caller guarantees M[2 i][k] = delta(i, k) - true of the reference's x2
   * periodic-sinc upsampler (ideal_lpf.py:96-121: the even phase of the zero-stuffed, recon-filtered signal is the
   * signal itself).  The even rows then need no product: M2 silu(M x) = M2[:, 0::2] silu(x) + M2[:, 1::2] silu(M[1::2] x).
   * Used where it pays (K = 128, bf16); 2 = also at K = 32 / 64, where the full product is faster (tests). */
  int up_identity;
} afldm_sep_args;
int afldm_sep_pass(const afldm_sep_args* args, afldm_stream_t stream);
/* table[b][c] = (rstd * gamma[c], beta[c] - mean * rstd * gamma[c]) from per-channel partial sums */
int afldm_gn_table(const float* stats, int S, const float* gamma, const float* beta, float* table, int B,
                   int C, int G, int HW, float eps, afldm_stream_t stream);
/* y[r][:] = softmax(x[r][:] * scale): the single-head d = 512 attention of the VAE mid block is
 * two GEMMs (afldm_conv2d with per-sample "weights") around this kernel. */
int afldm_softmax_rows(const void* x, void* y, long long rows, int cols, float scale, int dtype,
                       afldm_stream_t stream);

/* ---- convolution / linear as implicit GEMM on MFMA --------------------------------------
 * y[b,oh,ow,n] = bias[n] + temb[b*temb_stride + n] + residual[b,oh,ow,n]
 *              + sum_{kh,kw,ci} x[b, oh+kh-KS/2, ow+kw-KS/2, ci] * w[n,kh,kw,ci]
 * stride 1, zero padding KS/2, KS in {1,3}.  Replaces F.conv2d of ResnetBlock2D.conv1/conv2/
 * conv_shortcut, Downsample2D.conv with stride forced to 1 (af_blocks.py:129), Upsample2D.conv,
 * conv_in/conv_out, and nn.Linear (H = W = 1, B = rows) of Attention.to_q/k/v/to_out and the
 * time MLP.  x is the virtual concat of x1/x2.  out_mode 0: y NHWC with leading dim y_ld
 * (>= Cout, lets several GEMMs write column slices of one buffer); out_mode 1: channel-major
 * y[(b*Cout + n)*H*W + pix] (V^T for afldm_attention).
 * A pixel operand of 2 GiB or more (AF-VAE 256^2 levels at batch 128) is processed as B / c launches over c whole
 * samples each, c the largest divisor of B whose operand fits a buffer descriptor; afldm_conv2d_variant /
 * _stats_splits / _workspace answer for such a chunk. */
typedef struct {
  const void* x1;
  const void* x2;
  const void* w;
  const float* bias;
  const void* temb;     /* dtype T, may be NULL */
  const void* residual; /* dtype T, [B*H*W][res_ld], may be NULL */
  void* y;
  void* workspace; /* fp32 split-K slabs, may be NULL (=> no split-K) */
  size_t workspace_bytes;
  int C1, C2;
  int B, H, W, Cout, KS;
  int temb_stride; /* elements between samples in temb (0 = broadcast one row) */
  int res_ld, y_ld;
  int out_mode;
  int dtype;
  /* optional column split (fused Q|K|V projection): couts >= split_n go, channel-major
   * ([B][Cout - split_n][H*W]), to y2 instead of y; y then holds couts [0, split_n) with y_ld. */
  void* y2;
  int split_n;
  /* optional GroupNorm statistics of the OUTPUT (out_mode 0, no y2): per-channel partial sums
   * stats_out[B][S][Cout][2] fp32 with S = afldm_conv2d_stats_splits(args), computed on the stored
   * (rounded) values by the GEMM epilogue / the split-K reduction; NULL = none. */
  float* stats_out;
  /* 0, or the period of the time-embedding column: temb[b*temb_stride + n % temb_mod] (a 3x3
   * convolution on a 2x2 plane executed as ONE dense layer over the flattened plane: cout index =
   * pixel * C + c, every pixel takes the same per-channel time embedding; must divide Cout). */
  int temb_mod;
  /* optional: `sync_bytes` >= 40960 bytes of device memory that is ZERO before the first call and is left zero by
   * every call (one buffer serves all launches of a stream).  With it - and afldm_conv2d_fused_splitk(1) - a split-K
   * convolution whose workgroups all fit on the chip at once reduces its K slices inside the GEMM launch (arrival counter per output tile, slabs
   * written through, each slice finishing 1/splitk of the tile's rows) instead of in a second kernel; results are
   * bit-identical.  NULL = always the two-launch form. */
  unsigned int* sync;
  size_t sync_bytes;
  /* 1: when the call splits K (afldm_conv2d_variant(args) >> 8 & 255 = nslab > 1), leave the fp32 partial sums in
   * `workspace` ([nslab][B*H*W][Cout]: plain sums, no bias / temb / residual) and do NOT launch the reduction - the
   * caller hands them to a consumer that finishes them itself (afldm_af_act_slabs).  y / stats_out are not written.
   * Ignored (the call completes as usual) when K is not split. */
  int defer_reduce;
  /* 0, or the distance IN ELEMENTS between the weight tensors of consecutive samples: sample b is convolved with
   * w + b * w_batch_stride (KS = 1, out_mode 0, no split-K; a tile never spans two samples).  This is the "per-sample weights"
   * GEMM of the AF-VAE's single-head d = 512 attention (af_vae.py / diffusers AttnProcessor2_0): scores = Q_b K_b^T with K_b as
   * the weights, then P_b V_b with V_b^T as the weights - one launch for the batch instead of one per sample. */
  long long w_batch_stride;
  /* optional: the GroupNorm that FOLLOWS this convolution (diffusers Attention.group_norm behind ResnetBlock2D.conv2), applied by the
   * convolution's own epilogue where one tile holds a whole sample and whole groups (8x8 planes, bf16): y_norm [B*H*W][Cout] =
   * (y - mean_g) rstd_g gamma + beta with the statistics of the stored (rounded) y; y itself is written as usual.  Only honoured when
   * afldm_conv2d_norm_ok(args) == 1 (set y_norm = NULL otherwise and run afldm_gn_apply). */
  void* y_norm;
  const float* norm_gamma;
  const float* norm_beta;
  int norm_groups;
  float norm_eps;
  /* Operand layouts: 0 = NHWC (x [B,H,W,C], y [B,H,W,Cout]); 1 = 8-channel blocks [B][C/8][H][W][8] (bf16; 16 bytes per pixel and
   * block: what afldm_af_act writes / reads with y_layout / x_layout = 1).  The activation <-> 3x3 convolution tensors inside a
   * ResnetBlock2D travel in layout 1 where afldm_conv2d_c8_ok(args) == 1: an activation item (8 channels of a sample) is one
   * contiguous run instead of H*W 16-byte pieces.  Same values, same arithmetic. */
  int x_layout;
  int y_layout;
} afldm_conv_args;
int afldm_conv2d(const afldm_conv_args* args, afldm_stream_t stream);
/* 1: afldm_conv2d(args) accepts x_layout = 1 and / or y_layout = 1 (asked with both at 0 or 1: the answer does not depend on them). */
int afldm_conv2d_c8_ok(const afldm_conv_args* args);
/* 1: afldm_conv2d(args) will apply the GroupNorm described by norm_gamma / norm_beta / norm_groups / norm_eps into y_norm itself. */
int afldm_conv2d_norm_ok(const afldm_conv_args* args);
/* Tuning hook (benchmarks only): force tile/pipeline variant `variant` (>= 0) and/or a split-K
 * factor (>= 1) for subsequent afldm_conv2d calls; -1 restores the automatic choice. */
int afldm_conv2d_tune(int variant, int splitk);
/* Enable (1) / disable (0, the default) the in-kernel split-K reduction for calls that pass `sync` words: measured
 * slower than the two-launch form on MI355X, kept as a tested option. */
int afldm_conv2d_fused_splitk(int enable);
/* How afldm_conv2d will run this problem (tests / tuning tools): GEMM tile variant | split-K slices << 8 | in-kernel
 * reduction << 16; < 0 for the direct (non-GEMM) kernels. */
int afldm_conv2d_variant(const afldm_conv_args* args);
/* bytes of split-K workspace afldm_conv2d may use for this problem (0 if none). */
size_t afldm_conv2d_workspace(const afldm_conv_args* args);
/* split count S of the statistics afldm_conv2d writes to stats_out for this problem (> 0). */
int afldm_conv2d_stats_splits(const afldm_conv_args* args);

/* ---- attention ---------------------------------------------------------------------------
 * F.scaled_dot_product_attention as called by AttnProcessor2_0 / CrossFrameAttnProcessor
 * (cross_frame_attn.py:125,128): o = softmax(q k^T * scale) v per (batch, head).
 * q [B,Tq,ldq] k [Bk,Tk,ldk] token-major with head h at columns [h*d, (h+1)*d);
 * vt [Bk, heads*d, Tk] channel-major (written by afldm_conv2d out_mode 1); o [B,Tq,ldo].
 * Bk divides B: sample b reads K/V of kv-sample b / (B/Bk)  (the batch repeat of
 * cross_frame_attn.py:91-96). */
int afldm_attention(const void* q, int ldq, const void* k, int ldk, const void* vt, void* o, int ldo,
                    int B, int Bk, int heads, int Tq, int Tk, int d, float scale, int dtype,
                    afldm_stream_t stream);

/* ---- attention block front end, fused ------------------------------------------------------
 * group_norm -> to_q | to_k | to_v -> scaled_dot_product_attention of diffusers' AttnProcessor2_0 on the deprecated
 * attention-block configuration (the self-attention path of reference cross_frame_attn.py:66-77 in IDLE / STORE state;
 * diffusers attention_processor.py AttnProcessor2_0.__call__) as ONE launch: a workgroup owns a (sample, head), normalises
 * the raw tokens x [B,T,C] from the producer's per-channel partial sums `stats` [B,S,C,2] (as afldm_gn_apply does),
 * projects its head's q / k / v with w_qkv [3C,C] (to_q | to_k | to_v rows) + bias_qkv [3C] and runs the attention
 * against K / V^T resident in LDS; o [B,T,C] = the input of to_out.  q | k | v never exist in memory.
 * bf16 only; shapes: see afldm_attn_block_fused_supported (1 = there is a kernel for T tokens, C channels, head_dim,
 * G groups).  Rounding points: q, k, v and the softmax weights are rounded to bf16 as in afldm_conv2d + afldm_attention,
 * but the NORMALISED TOKENS ARE NEVER MATERIALISED: GroupNorm is folded into the head's weight rows, W' = bf16(W * rstd *
 * gamma) with the shift in the (fp32) bias, so bf16 rounds W' where the three-launch path rounds the normalised tokens, and
 * the softmax scale is applied to q before its one rounding.  Same error class (2.1e-3 against 1.9e-3 vs the fp32 form on
 * random data), different bits: a sample's bf16 result depends on which path its batch selected (policy: ops.py
 * _FUSED_ATTN_MIN_WGS; tests pin the path). */
int afldm_attn_block_fused_supported(int T, int C, int head_dim, int G);
/* diagnostic: device buffer [workgroups][waves][12] of 64-bit shader-clock stamps that later afldm_attn_block_fused
 * launches fill (phase boundaries per wave; csrc/attnf.hip), NULL = off (the default). */
int afldm_attn_block_fused_trace(void* buf);
int afldm_attn_block_fused(const void* x, const float* stats, int S, const float* gamma, const float* beta, int G,
                           float eps, const void* w_qkv, const float* bias_qkv, void* o, int B, int T, int C,
                           int heads, float scale, int dtype, afldm_stream_t stream);
/* The same launch carrying the rest of the attention block: y = to_out(o) + x (diffusers Attention.to_out[0] + the
 * residual connection of the attention-block configuration) and stats_out [B,heads,C,2] = per-channel partial sums of the
 * rounded y (the next GroupNorm's input; split h covers the token block [h T / heads, (h+1) T / heads)).  The `heads`
 * workgroups of a sample hand their o slices over inside the launch (one counter line per sample in `sync`, int32 words
 * [16384 + 32 b, ...), zero between launches; word 8193 = 1 when a workgroup gave up waiting, 2 when a sample's workgroups
 * were not on one XCD) and workgroup h then runs the to_out GEMM of its token block out of the XCD's L2.  w_out [C,C],
 * bias_out [C] fp32.  Shapes: afldm_attn_block_fused_out_supported (the 32 x 32 level, B a multiple of 8). */
int afldm_attn_block_fused_out_supported(int B, int T, int C, int head_dim, int G);
int afldm_attn_block_fused_out(const void* x, const float* stats, int S, const float* gamma, const float* beta, int G,
                               float eps, const void* w_qkv, const float* bias_qkv, void* o, const void* w_out,
                               const float* bias_out, void* y, float* stats_out, void* sync, long long sync_bytes, int B,
                               int T, int C, int heads, float scale, int dtype, afldm_stream_t stream);

/* ---- DDIM update -------------------------------------------------------------------------
 * DDIMScheduler.step, eta = 0, epsilon prediction, no clipping:
 *   x0 = (x - sqrt(1-a_t) eps) / sqrt(a_t);  x_prev = sqrt(a_prev) x0 + sqrt(1-a_prev) eps
 * coef: device float[4*nsteps] rows (sqrt(a_t), sqrt(1-a_t), sqrt(a_prev), sqrt(1-a_prev));
 * step_idx: device int, row to use; if advance != 0 the kernel increments it afterwards (so
 * a captured hipGraph of one step can be replayed 50x with no host involvement).
 * x, x_prev: NCHW fp32 latents [B,C,H,W]; eps: NHWC dtype (the UNet output layout). */
int afldm_ddim_step(const float* x, const void* eps, float* x_prev, const float* coef,
                    int* step_idx, int advance, int B, int C, int H, int W, int dtype,
                    afldm_stream_t stream);
/* Same update on flat fp32 tensors with the coefficients by value: the
 * DDIMScheduler.step(model_output, timestep, sample) API (ldm_pipeline.py:108-109). */
int afldm_ddim_step_flat(const float* x, const float* eps, float* x_prev, float sqrt_a_t,
                         float sqrt_1m_a_t, float sqrt_a_prev, float sqrt_1m_a_prev, size_t n,
                         afldm_stream_t stream);
/* tvals[step] -> t_out[0] (device->device), so the timestep also follows step_idx.  pre_advance != 0:
 * step_idx is incremented first (a sampler loop then starts from step_idx = -1 and needs no `advance`
 * launch behind afldm_ddim_step). */
int afldm_select_timestep(const float* tvals, int* step_idx, float* t_out, int pre_advance,
                          afldm_stream_t stream);
/* The same plus a table row: row_out[0 .. row_bytes) = table[step * row_bytes ..] (row_bytes % 16 == 0), one
 * launch.  A sampler knows its timesteps up front, so everything that depends on the timestep only - time_proj ->
 * TimestepEmbedding -> SiLU -> every ResnetBlock2D.time_emb_proj (diffusers UNet2DModel.forward) - is a table with
 * one row per step; the step then starts with this launch instead of seven. */
int afldm_select_step_row(const float* tvals, int* step_idx, float* t_out, int pre_advance, const void* table,
                          void* row_out, size_t row_bytes, afldm_stream_t stream);

/* ---- masked equivariance metrics (shift_utils/metrics.py:5-20) ------------------------------
 * One pass over a, b [B][n] (dtype) and mask [B][n] fp32: out[b] = { sum ((a - b) mask)^2, sum mask,
 * max(a mask), min(a mask), max(b mask), min(b mask) } (fp32 [B][6]).  mask_mse = mean_b out[b][0] / out[b][1];
 * mask_psnr = 10 log10(range^2 / mask_mse) with range = max over both - min over both. */
int afldm_masked_metrics(const void* a, const void* b, const float* mask, float* out, int B, size_t n,
                         int dtype, afldm_stream_t stream);

/* ---- upfirdn2d ---------------------------------------------------------------------------
 * Zero-stuffing up-sample (upx, upy) -> pad (negative = crop) -> 2-D FIR -> decimate (downx, downy) on
 * NCHW planes: torch_utils/ops/upfirdn2d.py:140-194 (`_upfirdn2d_ref`; the reference's fast path is the
 * vendored upfirdn2d.cu plugin).  x [planes][H][W], f [fh][fw] DEVICE fp32 taps, y [planes][outH][outW],
 * outW = (W*upx + padx0 + padx1 - fw) / downx + 1 (same for H).  flip_filter = 0 is a true convolution
 * (the filter is flipped), 1 a correlation; every tap is scaled by `gain` (the reference scales a 2-D
 * filter by gain and each pass of a separable one by sqrt(gain): callers pass the per-call factor).
 * Serves equivariance.py:68-103 (Lanczos fractional translation), shifters.py:158-161, :292-365. */
int afldm_upfirdn2d(const void* x, const float* f, void* y, int planes, int H, int W, int fh, int fw,
                    int upx, int upy, int downx, int downy, int padx0, int padx1, int pady0, int pady1,
                    int flip_filter, float gain, int dtype, afldm_stream_t stream);
/* Output extent along one axis for the arguments above, -1 when the plane is smaller than the filter. */
int afldm_upfirdn2d_out_size(int in, int up, int down, int pad0, int pad1, int ftaps);

/* ---- box calibration probes (measurement only: bench.py's `box` record) --------------------------------
 * afldm_probe_mfma: `workgroups` x 4 waves, each issuing iters x 4 independent v_mfma_f32_32x32x16_bf16
 * (flops = workgroups * 4 * iters * 4 * 32768); out [workgroups * 256] fp32 is never written.
 * afldm_probe_copy: 16-byte-per-lane grid-stride copy of `bytes` (a multiple of 16) device -> device.
 * Neither replaces a reference computation: they fingerprint the box (MFMA clock under load, HBM copy rate)
 * so that cross-box numbers can be normalised. */
int afldm_probe_mfma(float* out, int workgroups, int iters, afldm_stream_t stream);
/* the same loop on operands of random bf16 bit patterns rotating through four register sets (iters % 4 == 0): the
 * sustained MFMA rate of the chip is data-dependent (power): ~1.6 PFLOP/s here against ~2.4 on the near-constant
 * operands of afldm_probe_mfma. */
int afldm_probe_mfma_random(float* out, int workgroups, int iters, afldm_stream_t stream);
int afldm_probe_copy(const void* src, void* dst, size_t bytes, afldm_stream_t stream);
/* latency side of the fingerprint (round 4): afldm_probe_chase walks `steps` dependent loads through `buf` (uint32 indices
 * forming one cycle, built by the host with a stride beyond a cache line; nontemporal loads) on one lane and writes
 * out[0] = last index, out[1] = elapsed shader-clock ticks; afldm_probe_empty launches `workgroups` x 64 threads that do
 * nothing (captured N times into a graph: the kernel-to-kernel boundary of this box). */
int afldm_probe_chase(const void* buf, void* out, int steps, afldm_stream_t stream);
int afldm_probe_empty(int workgroups, afldm_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* AFLDM_HIP_H */
