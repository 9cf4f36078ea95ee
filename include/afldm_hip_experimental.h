/* afldm_hip_experimental.h - EXPERIMENTAL entry points (libafldm_exp.so), NOT part of the drop-in boundary.
 *
 * Structural designs that were built, validated bit-identical against the launches they replace and MEASURED SLOWER on
 * MI355X (profiles/r05/actconv_ab.txt, trunk_coop.txt, attn_small_ab.txt); they are kept, tested, for A/B work only.  A
 * binder of the reference must not use them: the product library (libafldm_hip.so, include/afldm_hip.h) neither exports
 * nor needs them, and the default execution path never loads libafldm_exp.so.  Same conventions as afldm_hip.h (device
 * pointers, stream-ordered, int status + afldm_last_error() of libafldm_hip.so, against which this library links).
 */
#ifndef AFLDM_HIP_EXPERIMENTAL_H
#define AFLDM_HIP_EXPERIMENTAL_H

#include "afldm_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- merged launch: [GroupNorm ->] WarpedNonlinearity -> 3x3 convolution ---------------------
 * `hidden_states = self.nonlinearity(self.norm1(x)); hidden_states = self.conv1(hidden_states)` of diffusers
 * ResnetBlock2D.forward with the reference's WarpedNonlinearity in place of SiLU (af_blocks.py:19-28), and the
 * norm2 -> nonlinearity -> conv2 pair after it, as ONE launch at the 32^2 / 16^2 levels (bf16; csrc/actconv.hip): the
 * workgroups that convolve a sample's tiles first run that sample's activation items and hand the activated tensor over
 * inside the launch (per-sample counter in `conv->sync`).  `act` = the arguments of afldm_af_act without its output;
 * `conv` = the arguments of afldm_conv2d whose x1 is BOTH the activation's output buffer ([B,N,N,C1+C2], written) and the
 * convolution's input (x2 = NULL, C1 = act->C1 + act->C2).  conv->sync needs >= (16384 + 64 * B) * 4 bytes (zero between
 * launches; words [16384, ...) hold one 128-byte counter line per sample, word 8193 an error flag: 1 = a workgroup gave up
 * waiting for its cluster, 2 = a workgroup did not run on the XCD its id implies).
 * Results are bit-identical to afldm_af_act(..., y = conv->x1) followed by afldm_conv2d(conv); shapes without a merged
 * kernel (afldm_af_act_conv2d_merged(...) == 0: other plane sizes, fp32, split-K plans, no sync words, AFLDM_NO_ACTCONV=1)
 * run as exactly those two launches. */
typedef struct {
  const void* x1;
  const void* x2;       /* second tensor of a virtual concat, or NULL */
  int C1, C2;
  const float* stats1;  /* per-channel GroupNorm partial sums of x1 [B][S1][C1][2], NULL = no normalisation */
  int S1;
  const float* stats2;
  int S2;
  const float* gamma;
  const float* beta;
  int G;
  float eps;
  const float* U;       /* as afldm_af_act */
  const float* D;
  const void* packed;
} afldm_af_act_args;
int afldm_af_act_conv2d_merged(const afldm_af_act_args* act, const afldm_conv_args* conv);
/* diagnostic: device buffer [workgroups][16] of 64-bit s_memtime stamps that later merged launches fill (0 start; 1 / 2 first
 * activation's prologue / items done; 3 / 4 / 5 first hand-over: stores acknowledged, cluster complete, left; 6 tile done;
 * 7 / 8 / 9 second hand-over; 10 / 11 second activation), NULL = off (the default). */
int afldm_af_act_conv2d_trace(void* buf);
/* 1: every merged launch uses the general (placement-independent) hand-over - write-through stores, agent-scope atomics,
 * acquire fence - instead of the XCD-local one it picks when every cluster sits inside one XCD; 0 (default): automatic.
 * Same results either way (tests; A/B of the two forms). */
int afldm_af_act_conv2d_mode(int general);
int afldm_af_act_conv2d(const afldm_af_act_args* act, const afldm_conv_args* conv, afldm_stream_t stream);
/* The general chain: [pre: norm1 -> nonlinearity ->] conv [-> post: norm2 -> nonlinearity] of ResnetBlock2D.forward as ONE
 * launch (pre or post may be NULL, not both).  `post` normalises the convolution's OWN output with the partial sums its
 * epilogue writes: post->x1 = conv->y, post->C1 = conv->Cout, post->C2 = 0, post->stats1 = conv->stats_out (or NULL: no
 * normalisation), post->S1 = afldm_conv2d_stats_splits(conv); the activated result goes to post_y [B,N,N,Cout].  The tile's
 * workgroups hand conv->y over per sample inside the launch and run that sample's activation items out of the XCD's L2;
 * conv->y itself is still written.  conv->sync needs >= (16384 + 64 * B) * 4 bytes.  Bit-identical to the two / three
 * launches, which is also how shapes without a merged kernel run (afldm_act_conv_act_merged == 0). */
int afldm_act_conv_act_merged(const afldm_af_act_args* pre, const afldm_conv_args* conv, const afldm_af_act_args* post);
int afldm_act_conv_act(const afldm_af_act_args* pre, const afldm_conv_args* conv, const afldm_af_act_args* post, void* post_y,
                       afldm_stream_t stream);

/* ---- the 2x2 level of the UNet as ONE cooperative launch ------------------------------------------
 * diffusers UNet2DModel.forward over down_blocks[-1] -> mid_block -> the resnets of up_blocks[0] (reference
 * configs/ldm/model_unet.json after the surgery of af_api.py:70-83): 7 ResnetBlock2D + the mid block's self-attention on
 * 2 x 2 planes, bf16 (csrc/trunk.hip).  One persistent workgroup per CU walks `phases` - an array of `nphases` records of
 * afldm_trunk_phase_bytes() bytes each, written by the host mirror (afldm_amd/trunk.py: GEMM partial products on 192 x 192
 * weight blocks / slab reduction + bias + time embedding + residual + GroupNorm + WarpedNonlinearity / 4-token attention) -
 * with a grid barrier between phases.  x_in / y_out [B,2,2,C] are the level's input and output, temb the step's
 * time-embedding row(s) at the level's first resnet (temb_stride elements between samples, 0 = shared); U [4x2], D [2x4] the
 * N = 2 filter matrices.  sync: >= 32832 bytes, zero between launches (words 8200 / 8201; word 8193 = 3 when a barrier
 * timed out).  Every workgroup must be resident at once: the launch needs the GPU to itself (one such launch at a time). */
int afldm_trunk_phase_bytes(void);
int afldm_trunk_trace(void* buf);      /* diagnostic: [nphases][2] 64-bit s_memtime stamps of workgroup 0 (phase start, barrier passed) */
int afldm_trunk_run(const void* phases, int nphases, const void* x_in, void* y_out, const void* temb, int temb_stride,
                    const float* U, const float* D, unsigned int* sync, size_t sync_bytes, afldm_stream_t stream);

/* ---- small planes: q | k | v projection + attention in one launch ------------------------------
 * to_q | to_k | to_v -> scaled_dot_product_attention of diffusers' AttnProcessor2_0 (reference cross_frame_attn.py:66-77, IDLE
 * branch) at the 8x8 (T = 64, C = 384) and 4x4 (T = 16, C = 768) levels, head_dim 24, bf16 (csrc/attns.hip): x [B,T,C] are the
 * tokens AFTER Attention.group_norm (the producing convolution / slab consumer applies it: afldm_conv_args.y_norm,
 * afldm_af_act_slabs), w_qkv [3C,C] / bias_qkv [3C] the packed projections, o [B,T,C] the input of to_out.  q | k | v are
 * rounded to bf16 as the three-launch path stores them and never leave the CU. */
int afldm_attn_small_fused_supported(int B, int T, int C, int heads);
int afldm_attn_small_fused(const void* x, const void* w_qkv, const float* bias_qkv, void* o, int B, int T, int C, int heads,
                           float scale, int dtype, afldm_stream_t stream);

#ifdef __cplusplus
}
#endif

#endif /* AFLDM_HIP_EXPERIMENTAL_H */
