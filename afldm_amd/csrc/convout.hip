// convout.hip — the tail of the UNet in one launch (gfx950, bf16): conv_norm_out (GroupNorm) -> conv_act (plain
// SiLU: make_af_unet does not wrap it, af_api.py:70-83) -> conv_out (3x3, C -> <= 4 channels).
//
// As three launches this was a GroupNorm + SiLU pass (25 MB read + 25 MB written at batch 64) and an implicit GEMM
// whose 64-cout tile carried 4 real couts (28 us: 16x the MFMA and weight-DMA work of the layer), 41 us together.
// Here a workgroup owns 8 image rows of one sample: it forms the GroupNorm scale / shift table from the producer's
// per-channel partial sums, loads its (8 + 2) x (32 + 2) x C halo patch ONCE, normalises + SiLUs it on the way into
// LDS (zero padding applies to the activated tensor: padded positions are written as zeros), and walks the 9 taps x
// C channels as 16x16x32 MFMAs with the pixels on the N side and the couts on the M side: lanes of output rows
// 4 .. 15 read a shared zero chunk instead of weights.  The activated tensor never exists in HBM.
//
// LDS patch layout = conv3h.hip's: one 128-byte row per (64-channel block, patch pixel), chunk positions swizzled by
// the pixel index so that the fragment reads of 16 consecutive pixels are conflict free for every tap shift.
#include "common.hpp"

namespace afldm {

struct ConvOutP {
  const bf16* x;
  GnStats gs;
  const float* gamma;
  const float* beta;
  const bf16* w;       // packed OHWI [Cout][3][3][C]
  const float* bias;
  bf16* y;             // NHWC [B][32][32][Cout]
  int B, G, Cout;
  float eps;
};

// (conv3h.hip, MF = 16) position of chunk c = kc * 4 + lg of a row with swizzle bits sw
__device__ __forceinline__ int co_pos(int c, int sw) {
  return (((c & 1) << 2) | ((c >> 2) << 1) | ((c >> 1) & 1)) ^ sw;
}

template <int C>
__global__ void __launch_bounds__(512) k_conv_out_fused(ConvOutP p) {
  constexpr int W_ = 32, ROWS = 8, PW = W_ + 2, PR = ROWS + 2, NPQ = PR * PW, NCB = C / 64, CH8 = C / 8;
  constexpr int PATCH = NPQ * 128;                       // bytes of one channel block's patch
  constexpr int W_BYTES = 4 * 9 * C * 2;                 // four weight rows
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sP = smem;                                       // [NCB][NPQ][128 B]
  bf16* sW = reinterpret_cast<bf16*>(smem + NCB * PATCH);                         // [4][9][C]
  char* sZ = smem + NCB * PATCH + W_BYTES;               // 16 zero bytes
  float* sc = reinterpret_cast<float*>(sZ + 16);         // [C] scale, [C] shift, [2 G] scratch
  float* sh = sc + C;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const int b = blockIdx.x / (W_ / ROWS), oh0 = (blockIdx.x - b * (W_ / ROWS)) * ROWS;
  const int HW = W_ * W_, G = p.G, cpg = C / G;

  // ---- GroupNorm table of this sample (as k_gn_apply: 8 lanes per group, fp64 finish)
  for (int g0 = 0; g0 < G; g0 += 64) {
    const int g = g0 + (tid >> 3), part = tid & 7;
    double s1 = 0.0, s2 = 0.0;
    if (g < G)
      for (int c = g * cpg + part; c < (g + 1) * cpg; c += 8) gn_channel_sums(p.gs, b, c, s1, s2);
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {
      s1 += __shfl_xor(s1, o, 64);
      s2 += __shfl_xor(s2, o, 64);
    }
    if (g < G && part == 0) {
      float mean, rstd;
      gn_mean_rstd(s1, s2, (double)HW * cpg, p.eps, mean, rstd);
      sc[2 * C + 2 * g] = mean;
      sc[2 * C + 2 * g + 1] = rstd;
    }
  }
  // ---- weights (rows >= Cout are zero) and the zero chunk
  for (int i = tid; i < 4 * 9 * C / 8; i += 512) {
    const int row = i / (9 * C / 8);
    bf16x8 v;
    if (row < p.Cout) v = *reinterpret_cast<const bf16x8*>(p.w + (size_t)i * 8);
    else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (bf16)0.0f;
    }
    *reinterpret_cast<bf16x8*>(sW + (size_t)i * 8) = v;
  }
  if (tid < 4) reinterpret_cast<float*>(sZ)[tid] = 0.f;
  __syncthreads();
  for (int c = tid; c < C; c += 512) {
    const float mean = sc[2 * C + 2 * (c / cpg)], rstd = sc[2 * C + 2 * (c / cpg) + 1];
    const float k = rstd * p.gamma[c];
    sc[c] = k;
    sh[c] = p.beta[c] - mean * k;
  }
  __syncthreads();

  // ---- the halo patch: GroupNorm + SiLU on the way into LDS, four independent 16-byte loads in flight per thread
  {
    constexpr int ITEMS = NPQ * CH8, U = 4;
    const bf16* xb = p.x + (size_t)b * HW * C;
    for (int i0 = tid; i0 < ITEMS; i0 += 512 * U) {
      bf16x8 v[U];
      bool ok[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = i0 + u * 512;
        const int q = i / CH8, ch8 = i - q * CH8;
        const int pr = q / PW, pc = q - pr * PW;
        const int ih = oh0 + pr - 1, iw = pc - 1;
        ok[u] = i < ITEMS && ih >= 0 && ih < W_ && iw >= 0 && iw < W_;
        if (ok[u]) v[u] = *reinterpret_cast<const bf16x8*>(xb + ((size_t)ih * W_ + iw) * C + ch8 * 8);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = i0 + u * 512;
        if (i < ITEMS) {
          const int q = i / CH8, ch8 = i - q * CH8;
          const int cb = ch8 >> 3, c = ch8 & 7;            // 64-channel block, chunk kc * 4 + lg inside it
          bf16x8 o;
          if (ok[u]) {
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (bf16)silu_f((float)v[u][e] * sc[ch8 * 8 + e] + sh[ch8 * 8 + e]);
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (bf16)0.0f;
          }
          *reinterpret_cast<bf16x8*>(sP + cb * PATCH + q * 128 + (co_pos(c, (q >> 1) & 3) << 4)) = o;
        }
      }
    }
  }
  __syncthreads();

  // ---- 9 taps x C channels: M side = couts (4 real rows), N side = 16 pixels; a wave owns two pixel tiles
  f32x4 acc[2];
  acc[0] = acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
  int qb[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int pt = wave * 2 + t;                          // 16 tiles of 16 pixels: image row pt / 2, column half pt % 2
    qb[t] = (pt >> 1) * PW + (pt & 1) * 16 + li;
  }
  const bool real = li < 4;
#pragma unroll 1
  for (int cb = 0; cb < NCB; ++cb) {
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int tapoff = (tap / 3) * PW + (tap - (tap / 3) * 3);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const char* aptr = real ? reinterpret_cast<const char*>(sW + ((size_t)li * 9 + tap) * C + cb * 64 + kk * 32 + lg * 8) : sZ;
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(aptr);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int q = qb[t] + tapoff;
          const bf16x8 bq = *reinterpret_cast<const bf16x8*>(sP + cb * PATCH + ((q * 128 + (co_pos(lg, (q >> 1) & 3) << 4)) ^ (kk << 5)));
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bq, acc[t], 0, 0, 0);
        }
      }
    }
  }
  // ---- epilogue: lane group 0 holds couts 0 .. 3 of pixel li
  if (lg == 0) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int pt = wave * 2 + t;
      const int oh = oh0 + (pt >> 1), ow = (pt & 1) * 16 + li;
      bf16* dst = p.y + ((size_t)(b * W_ + oh) * W_ + ow) * p.Cout;
      for (int n = 0; n < p.Cout; ++n) dst[n] = (bf16)(acc[t][n] + (p.bias ? p.bias[n] : 0.f));
    }
  }
}

}  // namespace afldm

using namespace afldm;

extern "C" int afldm_conv_out_fused(const void* x, const float* stats, int S, const float* gamma, const float* beta, int G,
                                    float eps, const void* w, const float* bias, void* y, int B, int N, int C, int Cout,
                                    int dtype, afldm_stream_t stream) {
  AFLDM_REQUIRE(x && stats && gamma && beta && w && y, AFLDM_ENULL, "afldm_conv_out_fused: NULL pointer");
  AFLDM_REQUIRE(dtype == AFLDM_BF16 && N == 32 && (C == 64 || C == 128 || C == 192) && Cout >= 1 && Cout <= 4 && B > 0 &&
                    G > 0 && G <= 64 && C % G == 0 && S > 0,
                AFLDM_ESHAPE, "afldm_conv_out_fused: bf16, 32x32 planes, C in {64,128,192}, Cout <= 4 only (N=%d C=%d Cout=%d dtype=%d)",
                N, C, Cout, dtype);
  ConvOutP p;
  p.x = (const bf16*)x; p.gs = GnStats{stats, nullptr, C, 0, S, 0}; p.gamma = gamma; p.beta = beta;
  p.w = (const bf16*)w; p.bias = bias; p.y = (bf16*)y; p.B = B; p.G = G; p.Cout = Cout; p.eps = eps;
  hipStream_t st = (hipStream_t)stream;
  const int grid = B * 4;
  auto lds_of = [](int c) { return (c / 64) * 340 * 128 + 4 * 9 * c * 2 + 16 + (2 * c + 2 * 64) * 4; };
#define AFLDM_CO(C_)                                                                                                   \
  if (C == C_) {                                                                                                       \
    static unsigned long long attr_set = 0;                                                                            \
    if (first_on_device(attr_set)) {                                                                                   \
      (void)hipFuncSetAttribute((const void*)k_conv_out_fused<C_>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_of(C_)); \
    }                                                                                                                  \
    k_conv_out_fused<C_><<<grid, 512, lds_of(C_), st>>>(p);                                                            \
  }
  AFLDM_CO(64) AFLDM_CO(128) AFLDM_CO(192)
#undef AFLDM_CO
  return check_launch("afldm_conv_out_fused");
}
