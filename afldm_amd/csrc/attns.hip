// attns.hip - q | k | v projection + attention of the SMALL planes (8^2: 64 tokens, 4^2: 16 tokens) in ONE launch (round 5).
//
// Reference: diffusers AttnProcessor2_0 on the attention-block configuration (the self-attention branch of reference
// cross_frame_attn.py:66-77): to_q / to_k / to_v on the GroupNorm-ed tokens, softmax(q k^T * scale) v per (sample, head).
// At these levels the two launches it replaces - the fused projection GEMM (k_lin_wreg / k_igemm2, 17 us) and k_attn (6 - 8 us) -
// sit on their latency floors (10 launches, 0.25 ms of the batch-64 step for ~1 % of its flops), and q | k | v make a round trip
// through memory in between.  The normalised tokens are free at both levels: conv2's epilogue (8^2, afldm_conv_args.y_norm) or
// the slab consumer (4^2, afldm_af_act_slabs) has already applied Attention.group_norm.
//
// A workgroup owns ONE head and a group of samples (8^2: 4 samples, one per wave; 4^2: 8 samples, two per wave):
//   * the head's 72 weight rows (q | k | v, 24 each) go to LDS once (swizzled 16-byte pieces, rows padded to 80);
//   * every wave projects ITS samples' tokens: token rows straight from global memory as MFMA B fragments (the next 16-token
//     tile's loads in flight under the current tile's MFMAs), W fragments from LDS, + bias, ONE rounding to bf16 - the values
//     the three-launch path stores - into a wave-private LDS tile [token][80];
//   * the wave's lanes are its queries: 24-dim dot products against the sample's keys (LDS broadcasts), softmax in fp32 with
//     the weights rounded to bf16 before P V (k_attn's rounding points), one rounding of the output, 48 contiguous bytes per
//     token.  No workgroup barrier after the weights have landed; q | k | v never exist in memory.
#include "common.hpp"
#include "../../include/afldm_hip_experimental.h"      // (libafldm_exp.so: not part of the product library)

namespace afldm {

struct AttnSP {
  const bf16* x;       // [B][T][C] GroupNorm-ed tokens
  const bf16* w;       // [3C][C]: to_q | to_k | to_v rows
  const float* bias;   // [3C]
  bf16* o;             // [B][T][C]
  int B, C, heads;
  float scale_log2e;   // softmax scale * log2(e)
};

// T: tokens per sample (64 / 16); TW: tokens per wave (64 / 32); CK: C / 32
template <int T, int TW, int CK>
__global__ void __launch_bounds__(256) k_attn_small(AttnSP p) {
  constexpr int D = 24, NR = 80, C = CK * 32, RW = C * 2;        // rows padded to 5 MFMA tiles; weight row bytes
  constexpr int NTT = TW / 16, SPW = TW / T, QS = 80;              // token tiles per wave, samples per wave, staged row stride (elements)
  constexpr int W_BYTES = NR * RW, Q_BYTES = TW * QS * 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sW = smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
  bf16* sQ = reinterpret_cast<bf16*>(smem + W_BYTES + wave * Q_BYTES);        // this wave's [TW][QS]: q 0..23 | k 24..47 | v 48..71
  const int wi = xcd_remap(blockIdx.x, gridDim.x);
  const int grp = wi / p.heads, h = wi - grp * p.heads;
  const int tok0 = (grp * 4 + wave) * TW;                          // first token (global row of x / o) of this wave

  // ---- first token tile requested before anything waits
  const bf16* xr = p.x + (size_t)(tok0 + li) * C + lg * 8;
  bf16x8 xb[2][CK];
#pragma unroll
  for (int ks = 0; ks < CK; ++ks) xb[0][ks] = ld16<bf16x8>(xr + ks * 32);

  // ---- the head's weight rows -> LDS: piece c of row r at position c ^ (r & 7) inside its 8-piece group (all rows start on
  // the same bank: the swizzle spreads the 16 rows of a fragment read over 8 four-bank groups, twice)
  {
    constexpr int PPR = C / 8;                                     // 16-byte pieces per row
    for (int i = tid; i < NR * PPR; i += 256) {
      const int r = i / PPR, c = i - r * PPR;
      bf16x8 v = Mma<bf16>::zero();
      if (r < 3 * D) {
        const int m = r / D, rr = r - m * D;
        v = ld16<bf16x8>(p.w + ((size_t)m * C + h * D + rr) * C + c * 8);
      }
      st16<bf16x8>(sW + r * RW + (((c & ~7) | ((c ^ r) & 7)) << 4), v);
    }
  }
  // biases of the rows this lane's accumulators hold: rows 16 rt + 4 lg + e
  float bq[5][4];
#pragma unroll
  for (int rt = 0; rt < 5; ++rt)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int r = 16 * rt + 4 * lg + e;
      bq[rt][e] = r < 3 * D ? p.bias[(r / D) * C + h * D + (r % D)] : 0.f;
    }
  __syncthreads();

  // ---- projection: per 16-token tile, 5 row tiles x CK K steps
#pragma unroll
  for (int tt = 0; tt < NTT; ++tt) {
    if (tt + 1 < NTT) {
#pragma unroll
      for (int ks = 0; ks < CK; ++ks) xb[(tt + 1) & 1][ks] = ld16<bf16x8>(xr + (size_t)(tt + 1) * 16 * C + ks * 32);
    }
    f32x4 acc[5];
#pragma unroll
    for (int rt = 0; rt < 5; ++rt) acc[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < CK; ++ks) {
#pragma unroll
      for (int rt = 0; rt < 5; ++rt) {
        const int r = 16 * rt + li, c = ks * 4 + lg;
        const bf16x8 wf = ld16<bf16x8>(sW + r * RW + (((c & ~7) | ((c ^ r) & 7)) << 4));
        acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xb[tt & 1][ks], acc[rt], 0, 0, 0);
      }
    }
    // lane (token li, lg): rows 16 rt + 4 lg + e of token 16 tt + li -> 8 bytes of the staged row
#pragma unroll
    for (int rt = 0; rt < 5; ++rt) {
      bf16x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = (bf16)(acc[rt][e] + bq[rt][e]);
      *reinterpret_cast<bf16x4*>(sQ + (tt * 16 + li) * QS + 16 * rt + 4 * lg) = v;
    }
  }
  // (the wave's own LDS writes are ordered before its reads; no other wave touches this tile)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0)

  // ---- attention: lane = query token of this wave (TW <= 64 lanes live), keys = the T tokens of its sample
  if (lane < TW) {
    const int s0 = (lane / T) * T;                                  // first token (in the wave) of this lane's sample
    float q[D];
    {
      const bf16* qr = sQ + lane * QS;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const bf16x8 v = ld16<bf16x8>(qr + 8 * c);
#pragma unroll
        for (int e = 0; e < 8; ++e) q[8 * c + e] = (float)v[e] * p.scale_log2e;
      }
    }
    float sc[T], m = -3.0e38f;
#pragma unroll
    for (int j = 0; j < T; ++j) {
      const bf16* kr = sQ + (s0 + j) * QS + D;
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const bf16x8 v = ld16<bf16x8>(kr + 8 * c);
#pragma unroll
        for (int e = 0; e < 8; ++e) s = fmaf(q[8 * c + e], (float)v[e], s);
      }
      sc[j] = s;
      m = fmaxf(m, s);
    }
    float o[D], den = 0.f;
#pragma unroll
    for (int d = 0; d < D; ++d) o[d] = 0.f;
#pragma unroll
    for (int j = 0; j < T; ++j) {
      const float pw = (float)(bf16)__builtin_amdgcn_exp2f(sc[j] - m);
      den += pw;
      const bf16* vr = sQ + (s0 + j) * QS + 2 * D;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const bf16x8 v = ld16<bf16x8>(vr + 8 * c);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[8 * c + e] = fmaf(pw, (float)v[e], o[8 * c + e]);
      }
    }
    const float inv = 1.0f / den;
    bf16* dst = p.o + (size_t)(tok0 + lane) * C + h * D;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      bf16x8 v;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (bf16)(o[8 * c + e] * inv);
      st16<bf16x8>(dst + 8 * c, v);
    }
  }
}

template <int T, int TW, int CK>
static int launch_attn_small(const AttnSP& p, hipStream_t st) {
  constexpr int lds = 80 * CK * 64 + 4 * TW * 80 * 2;
  static_assert(lds <= 160 * 1024, "LDS");
  static unsigned long long attr_set = 0;
  if (first_on_device(attr_set))
    (void)hipFuncSetAttribute((const void*)k_attn_small<T, TW, CK>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  const int groups = p.B * T / (4 * TW);
  k_attn_small<T, TW, CK><<<groups * p.heads, 256, lds, st>>>(p);
  return check_launch("afldm_attn_small_fused");
}

}  // namespace afldm

using namespace afldm;

extern "C" int afldm_attn_small_fused_supported(int B, int T, int C, int heads) {
  if (heads <= 0 || C != heads * 24) return 0;
  if (T == 64 && C == 384 && B % 4 == 0) return 1;
  if (T == 16 && C == 768 && B % 8 == 0) return 1;
  return 0;
}

extern "C" int afldm_attn_small_fused(const void* x, const void* w_qkv, const float* bias_qkv, void* o, int B, int T, int C, int heads,
                                      float scale, int dtype, afldm_stream_t stream) {
  AFLDM_REQUIRE(x && w_qkv && bias_qkv && o, AFLDM_ENULL, "afldm_attn_small_fused: NULL pointer");
  AFLDM_REQUIRE(dtype == AFLDM_BF16 && afldm_attn_small_fused_supported(B, T, C, heads), AFLDM_ESHAPE,
                "afldm_attn_small_fused: no kernel for B=%d T=%d C=%d heads=%d dtype=%d (afldm_attn_small_fused_supported)", B, T, C, heads, dtype);
  AFLDM_REQUIRE(aligned16(x) && aligned16(w_qkv) && aligned16(o), AFLDM_EALIGN, "afldm_attn_small_fused: pointers must be 16-byte aligned");
  AttnSP p;
  p.x = (const bf16*)x;
  p.w = (const bf16*)w_qkv;
  p.bias = bias_qkv;
  p.o = (bf16*)o;
  p.B = B;
  p.C = C;
  p.heads = heads;
  p.scale_log2e = scale * 1.4426950408889634f;
  if (T == 64) return launch_attn_small<64, 64, 12>(p, (hipStream_t)stream);
  return launch_attn_small<16, 32, 24>(p, (hipStream_t)stream);
}
