// attnf.hip — the attention block's front end as ONE launch (round 4):
//   GroupNorm-apply -> per-head q | k | v projection -> softmax(q k^T * scale) v
// for the self-attention blocks of the 32^2 / 16^2 levels (AttnProcessor2_0 on the deprecated attention-block
// configuration; reference cross_frame_attn.py:66-130 IDLE branch = diffusers AttnProcessor2_0).  bf16 only.
//
// The three-launch form (k_gn_apply, k_lin_wreg, k_attn) exists to move data: at batch 64 the projection writes 634 MB
// of q | k | v^T per step that attention reads straight back (641 MB), and the attention kernel re-stages every K / V^T
// chunk once per 256 queries behind a workgroup barrier.  Here a workgroup owns ONE (sample, head):
//
//   phase A  GroupNorm is FOLDED into the head's 72 weight rows once per workgroup (W' = bf16(W a), the shift goes into the
//            bias, formed with the rounded W': see the prologue), so each wave projects ITS tokens (32-token tiles) with
//            the raw token rows going from global memory straight into the MFMA as B / A fragments: q / k / v are three
//            32x32x16 MFMA chains against W' held in LDS.  K and V^T are written to LDS ONCE, in the fragment order
//            the attention loop reads; Q never leaves the registers (the projection's accumulator layout IS the
//            B-fragment layout after a fixed permutation of the weight rows).
//   phase B  flash attention over the resident K / V^T: no global loads, no staging, NO workgroup barrier - the waves
//            run free.  S^T = K Q^T so the softmax is lane-local, the running reference is subtracted by the MFMA
//            (K carries a 1, Q carries -m in the padding channel head_dim -> 32), the row sums come from a row of ones
//            in V^T, rescaling is lazy (only when a score outgrows the reference by 2^10).  The next tile's S^T MFMAs
//            are issued before the current tile's exponentials so that matrix and vector pipes overlap inside a wave.
//
// x is read once per head from the XCD's L2 (the heads of a sample run on one XCD), q | k | v never exist in HBM.
#include <stdlib.h>

#include "common.hpp"

namespace afldm {

typedef __attribute__((ext_vector_type(16))) float f32x16;

struct AttnFP {
  const bf16* x;       // [B][T][C] raw (pre-norm) tokens
  const bf16* w;       // [3C][C]: to_q | to_k | to_v rows
  const float* bias;   // [3C]
  const float* gamma;  // [C]
  const float* beta;   // [C]
  GnStats gs;          // per-channel partial sums of x (st1 [B][S][C][2])
  bf16* o;             // [B][T][C]
  int B, heads, C, G;
  float eps, qscale;   // qscale = softmax scale * log2(e)
  int force_slow;      // testing / A-B: always take the loop that tracks the row maxima
  int stagger;         // s_sleep units (64 cycles) the second half of the waves waits before the attention phase
  unsigned long long* trace;   // diagnostic: [workgroup][wave][8] s_memtime stamps (afldm_attn_block_fused_trace), or NULL
  // phase C (template OUT): to_out + residual + GroupNorm partial sums of the block's output inside the same launch
  const bf16* wo;      // [C][C] to_out rows
  const float* bias_o; // [C]
  bf16* y;             // [B][T][C] = to_out(o) + x
  float* stats_out;    // [B][heads][C][2] per-channel partial sums of y over the workgroup's token block
  unsigned* sync;      // hand-over counters: one 128-byte line per sample (word 0 arrivals, word 1 departures, word 2 XCD mask)
  unsigned* err;       // error word (1: a cluster timed out, 2: a cluster straddles XCDs)
};

__device__ __forceinline__ int attnf_xcc_id() {
  int v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 7;
}

static unsigned long long* g_attnf_trace = nullptr;

__device__ __forceinline__ f32x16 mfma32(const bf16x8& a, const bf16x8& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

constexpr float ATT_TAU = 10.0f;   // lazy rescale threshold (log2 units)

// D = head_dim (16 / 24), NW = waves, TPW = 32-token tiles per wave (T = 32 NW TPW), CK = C / 16 (K steps).
// LEAN: weight rows unpadded with an XOR swizzle of the 16-byte piece index (instead of 16 bytes of padding per row) and the
// GroupNorm vectors aliased onto the (not yet written) K region: the 16x16 level then needs <= 80 KiB and TWO four-wave
// workgroups share a CU - one's prologue / fold runs under the other's projection / attention.
template <int D, int NW, int TPW, int CK, bool LEAN = false>
struct AttnFCfg {
  static constexpr int T = NW * TPW * 32, NT = T / 32, C = CK * 16;
  static constexpr int RK = D * 2;                      // K row bytes (real channels only)
  static constexpr int K_BYTES = T * RK;
  static constexpr int V_BYTES = NT * 2 * 2 * D * 16;   // [tile][j][hi][d] x 16 B
  static constexpr int CST_BYTES = 64;                  // [0,16) K pad chunk {1,0,..}; [16,32) ones; [32,48) zeros
  static constexpr int RW = LEAN ? C * 2 : C * 2 + 16;  // weight row stride (bank-conflict-free ds_read_b128: padding, or the swizzle)
  static constexpr int W_BYTES = 3 * D * RW;
  static constexpr int AS_BYTES = C * 12;               // a[C], mu[C], beta[C] fp32 (GroupNorm folded into the weights)
  static constexpr int BIAS_BYTES = 3 * 32 * 4;         // folded q | k | v biases of this head, padded to 32, fp32
  static constexpr int KN_BYTES = 64;                   // per-wave max |k|^2 (fp32)
  static constexpr int OFF_K = 0, OFF_V = OFF_K + K_BYTES, OFF_CST = OFF_V + V_BYTES, OFF_W = OFF_CST + CST_BYTES,
                       OFF_AS = LEAN ? OFF_K : OFF_W + W_BYTES, OFF_BIAS = OFF_W + W_BYTES + (LEAN ? 0 : AS_BYTES),
                       OFF_KN = OFF_BIAS + BIAS_BYTES, LDS_BYTES = OFF_KN + KN_BYTES;
  static_assert(!LEAN || (AS_BYTES <= K_BYTES + V_BYTES && (C * 2) % 256 == 0), "aliased GroupNorm vectors / swizzled rows");
  static constexpr int NU = TPW >= 2 ? 2 : 1;           // query tiles per attention pass
  static constexpr int PADC = D / 16, PADHI = (D % 16) / 8;   // where slot D sits: chunk, lane half (element 0)
};

// DBG (AFLDM_ATTNF_DBG, timing decomposition, garbage results): 1 no attention phase, 2 no exponentials, 4 no projection MFMAs,
// 8 no token-tile reloads in the projection, 16 no W' fragment reads in the projection
template <int D, int NW, int TPW, int CK, int DBG = 0, bool LEAN = false, bool OUT = false>
__global__ void __launch_bounds__(NW * 64, LEAN ? 2 : 1) k_attn_fused(AttnFP p) {
  typedef AttnFCfg<D, NW, TPW, CK, LEAN> CF;
  constexpr int T = CF::T, NT = CF::NT, C = CF::C, RK = CF::RK, RW = CF::RW, NU = CF::NU, NTHR = NW * 64;
  static_assert(D == 16 || D == 24, "head_dim 16 / 24");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sK = smem + CF::OFF_K;
  char* sV = smem + CF::OFF_V;
  char* sC = smem + CF::OFF_CST;
  char* sW = smem + CF::OFF_W;
  float* sA = reinterpret_cast<float*>(smem + CF::OFF_AS);
  float* sB = reinterpret_cast<float*>(smem + CF::OFF_BIAS);
  float* sKN = reinterpret_cast<float*>(smem + CF::OFF_KN);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ln = lane & 31, hi = lane >> 5;
  // the heads of a sample (and neighbouring samples) on one XCD: x is re-read from that XCD's L2
  const int wi = xcd_remap(blockIdx.x, gridDim.x);
  const int b = wi / p.heads, h = wi - b * p.heads;
  auto stamp = [&](int i) {
    if (p.trace && lane == 0) p.trace[((size_t)blockIdx.x * NW + wave) * 12 + i] = __builtin_amdgcn_s_memtime();
  };
  stamp(0);

  // ------------------------------------------------------------------ prologue
  // GroupNorm is FOLDED into the head's weights, per workgroup (= per sample and head):
  //   y = W (a x + s) + bias,  a_c = rstd_g gamma_c,  s_c = beta_c - mu_g a_c
  //     = W' x + bias',        W'_oc = bf16(W_oc a_c),  bias'_o = bias_o + sum_c (W_oc beta_c - W'_oc mu_g(c))
  // so that the raw token rows go from global memory straight into the MFMA - no normalisation pass over the tile, no
  // staging through LDS (the second form of this kernel spent 16 of a workgroup's 60 us on them, LDS-bound).  The shift is
  // formed with the ROUNDED W': the rounding error of W' then multiplies the CENTRED x - the same error class as
  // rounding the normalised tokens (what afldm_gn_apply's bf16 output does), with no amplification by |mean| / std.
  //
  // Every global load of the prologue is ISSUED before anything waits on one (loads return in order): one memory round
  // trip instead of four dependent ones.
  // (a) GroupNorm partial sums: quarter-waves take one group each (cpg <= 16); consecutive lanes read consecutive
  //     channels of one split (one run), up to SV partials per lane in flight
  constexpr int SV = LEAN ? 8 : 16;
  const int cpg = C / p.G, S = p.gs.S1, q4 = lane >> 4, ql = lane & 15, nst = cpg * S;
  const float* stb = p.gs.st1 + (size_t)b * S * C * 2;
  float* sMu = sA + C;
  float* sBt = sA + 2 * C;
  f32x2 sv[SV];
  float gam = 0.f, bet = 0.f;
  auto stats_issue = [&](int g, int j0) {
#pragma unroll
    for (int u = 0; u < SV; ++u) {
      const int j = j0 + 16 * u;
      const int jj = (g < p.G && j < nst) ? j : 0, gg = g < p.G ? g : 0;
      const int sp = jj / cpg, cc = jj - sp * cpg;
      sv[u] = *reinterpret_cast<const f32x2*>(stb + ((size_t)sp * C + gg * cpg + cc) * 2);
    }
  };
  {
    const int g = wave * 4 + q4;
    if (g < p.G && ql < cpg) {
      gam = p.gamma[g * cpg + ql];
      bet = p.beta[g * cpg + ql];
    }
    stats_issue(g, ql);
  }
  // (b) the head's 3 D weight rows: LPR lanes per row, each PPT 16-byte pieces (interleaved) - the folded bias of a row is
  //     then a register sum + LPR-lane shuffle (no scratch, no extra barrier)
  constexpr int WPR = C / 8;                                   // 16-byte pieces per weight row
  constexpr int LPR = NTHR >= 3 * D * 4 ? 4 : NTHR >= 3 * D * 2 ? 2 : 1;
  constexpr int PPT = WPR / LPR;
  static_assert(WPR % LPR == 0 && 3 * D * LPR <= NTHR, "weight rows must divide among the threads");
  const int wrow = tid / LPR, wpart = tid % LPR;               // row 0 .. 3D-1 of (q | k | v), lane inside the row
  const bool wlive = wrow < 3 * D;
  const int wm = wlive ? wrow / D : 0, wrr = wlive ? wrow - wm * D : 0;
  bf16x8 wr[PPT];
  float bias_r = 0.f;
  if (wlive) {
    const bf16* wsrc = p.w + ((size_t)wm * C + h * D + wrr) * C;
#pragma unroll
    for (int i = 0; i < PPT; ++i) wr[i] = ld16<bf16x8>(wsrc + (wpart + LPR * i) * 8);
    bias_r = p.bias[wm * C + h * D + wrr];
  }
  // (c) this wave's first token tile, as MFMA fragments: lane (token, half) reads the 16 bytes of K step kk at
  //     channel 16 kk + 8 half (the two halves of a token are adjacent: 32-byte runs, 4 K steps per 128-byte line;
  //     the loads of a tile are issued back to back so that a line is re-used while it is still in the L1)
  //     (requested after the statistics have been consumed: at launch every workgroup of the chip is in its prologue
  //      and the burst is bandwidth-bound; the tile is not needed before the projection)
  constexpr int XPF = CK < 12 ? CK : 12;                       // K steps in flight (48 registers)
  const bf16* xrow0 = p.x + ((size_t)b * T + (size_t)wave * TPW * 32 + ln) * C + hi * 8;
  bf16x8 xr[XPF];

  // ---- consume: per-channel scale a, group mean mu, beta -> LDS
  for (int g0 = 0; g0 < p.G; g0 += NW * 4) {
    const int g = g0 + wave * 4 + q4;
    if (g0 > 0) {                                          // (more than 4 NW groups: the small test shapes)
      gam = bet = 0.f;
      if (g < p.G && ql < cpg) {
        gam = p.gamma[g * cpg + ql];
        bet = p.beta[g * cpg + ql];
      }
      stats_issue(g, ql);
    }
    double s1 = 0.0, s2 = 0.0;
    for (int j0 = ql;; j0 += 16 * SV) {
#pragma unroll
      for (int u = 0; u < SV; ++u) {
        if (g < p.G && j0 + 16 * u < nst) {
          s1 += (double)sv[u][0];
          s2 += (double)sv[u][1];
        }
      }
      if (j0 + 16 * SV >= nst) break;                       // (per lane: j0 depends on the lane; the shuffles below come after the loop)
      stats_issue(g, j0 + 16 * SV);
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
      s1 += __shfl_xor(s1, o, 64);
      s2 += __shfl_xor(s2, o, 64);
    }
    if (g < p.G && ql < cpg) {
      float mean, rstd;
      gn_mean_rstd(s1, s2, (double)T * cpg, p.eps, mean, rstd);
      const int c = g * cpg + ql;
      sA[c] = rstd * gam;
      sMu[c] = mean;
      sBt[c] = bet;
    }
  }
  if (tid < 3) {
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (bf16)((tid == 1 || (tid == 0 && e == 0)) ? 1.0f : 0.0f);
    st16<bf16x8>(sC + tid * 16, v);
  }
  stamp(1);
  __syncthreads();
  stamp(8);
#pragma unroll
  for (int kk = 0; kk < XPF; ++kk) xr[kk] = ld16<bf16x8>(xrow0 + kk * 16);
  // ---- W' = bf16(W a) -> LDS (padded rows); folded bias of the row
  if (wlive) {
    float part = 0.f;
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      const int pc = wpart + LPR * i;
      const f32x4 a0 = ld16<f32x4>(sA + pc * 8), a1 = ld16<f32x4>(sA + pc * 8 + 4);
      const f32x4 m0 = ld16<f32x4>(sMu + pc * 8), m1 = ld16<f32x4>(sMu + pc * 8 + 4);
      const f32x4 b0 = ld16<f32x4>(sBt + pc * 8), b1 = ld16<f32x4>(sBt + pc * 8 + 4);
      bf16x8 wf;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float w = (float)wr[i][e];
        const float av = e < 4 ? a0[e] : a1[e - 4], mv = e < 4 ? m0[e] : m1[e - 4], bv = e < 4 ? b0[e] : b1[e - 4];
        wf[e] = (bf16)(w * av);
        part += w * bv - (float)wf[e] * mv;
      }
      st16<bf16x8>(sW + wrow * RW + ((LEAN ? pc ^ (wrow & 15) : pc) << 4), wf);
    }
#pragma unroll
    for (int o = 1; o < LPR; o <<= 1) part += __shfl_xor(part, o, 64);       // (a row's LPR lanes are adjacent lanes of one wave)
    if (wpart == 0) sB[wm * 32 + wrr] = part + bias_r;
    if (wpart == LPR - 1 && wrr < 32 - D) sB[wm * 32 + D + wrr] = 0.f;        // the padding entries of the 32-wide bias rows
  }
  stamp(9);
  __syncthreads();
  stamp(2);

  // ------------------------------------------------------------------ phase A: projection
  // weight fragment rows: q / k use the row permutation sigma (swap bits 2, 3) so that accumulator register 8c + e of
  // lane (token, half) is channel 16c + 8 half + e: registers 8c .. 8c+7 ARE the B fragment chunk c of Q and one
  // 16-byte piece of K's row.  Rows >= D read another row (their results are never used).
  const int sig = (ln & 0x13) | ((ln & 4) << 1) | ((ln & 8) >> 1);
  // (padding lanes read row - 16: finite data, and a row no other lane of their ds_read_b128 group maps onto)
  const int wq_row = 0 * D + (sig < D ? sig : sig - 16), wk_row = 1 * D + (sig < D ? sig : sig - 16), wv_row = 2 * D + (ln < D ? ln : ln - 16);
  // fragment of K step kk = 16-byte piece 2 kk + half of the row (LEAN: at position piece ^ (row & 15))
  auto wfrag = [&](int row, int kk) {
    const int pc = 2 * kk + hi;
    return ld16<bf16x8>(sW + row * RW + ((LEAN ? pc ^ (row & 15) : pc) << 4));
  };

  bf16x8 qf[TPW][2];     // Q as B fragments (scaled by scale * log2 e), chunk c = channels 16c + 8 half + e
  float qn2[TPW];        // |q|^2 of this lane's query (of the bf16 values that enter the MFMA), per tile
  float kn2 = 0.f;       // max |k|^2 over this lane's keys
  const float bv = sB[2 * 32 + ln];                        // v: lane (channel d, half), every register is channel d
#pragma unroll
  for (int tt = 0; tt < TPW; ++tt) {
    const int tok0 = (wave * TPW + tt) * 32;               // first token of the tile (within the sample)
    f32x16 aq, ak, av;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      aq[r] = 0.f;
      ak[r] = 0.f;
      av[r] = bv;
    }
    // (W' fragments one K step ahead of their MFMAs: two waves per SIMD do not cover an LDS round trip per step)
    bf16x8 wq = wfrag(wq_row, 0), wk = wfrag(wk_row, 0), wv = wfrag(wv_row, 0);
#pragma unroll
    for (int kk = 0; kk < CK; ++kk) {
      const bf16x8 xb = xr[kk % XPF];
      const bf16x8 wq_c = wq, wk_c = wk, wv_c = wv;
      if (kk + 1 < CK && !(DBG & 16)) {
        wq = wfrag(wq_row, kk + 1);
        wk = wfrag(wk_row, kk + 1);
        wv = wfrag(wv_row, kk + 1);
      }
      // the freed register takes K step kk + XPF: of this tile, or of the next one
      if (!(DBG & 8)) {
        if (kk + XPF < CK) xr[kk % XPF] = ld16<bf16x8>(xrow0 + (size_t)tt * 32 * C + (kk + XPF) * 16);
        else if (tt + 1 < TPW) xr[kk % XPF] = ld16<bf16x8>(xrow0 + (size_t)(tt + 1) * 32 * C + (kk + XPF - CK) * 16);
      }
      if (!(DBG & 4)) {
        aq = mfma32(wq_c, xb, aq);        // [channel x token]
        ak = mfma32(wk_c, xb, ak);        // [channel x token]
        av = mfma32(xb, wv_c, av);        // [token x channel]: lane = channel, registers = tokens
      } else {
        aq[0] += (float)wq_c[0] + (float)xb[0];
        ak[0] += (float)wk_c[0];
        av[0] += (float)wv_c[0];
      }
    }
    // ---- tile epilogue: Q -> registers, K / V^T -> LDS; squared norms of the rounded rows (Cauchy-Schwarz bound on
    // the scores: decides, per wave, whether the attention loop has to track row maxima at all)
    float qs = 0.f, ks = 0.f;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      // folded biases: q / k register r = 8c + e -> channel 16c + 8 half + e
      const f32x4 bq0 = ld16<f32x4>(sB + 0 * 32 + 16 * c + 8 * hi), bq1 = ld16<f32x4>(sB + 0 * 32 + 16 * c + 8 * hi + 4);
      const f32x4 bk0 = ld16<f32x4>(sB + 1 * 32 + 16 * c + 8 * hi), bk1 = ld16<f32x4>(sB + 1 * 32 + 16 * c + 8 * hi + 4);
      bf16x8 qv, kv;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        qv[e] = (bf16)((aq[8 * c + e] + (e < 4 ? bq0[e] : bq1[e - 4])) * p.qscale);
        kv[e] = (bf16)(ak[8 * c + e] + (e < 4 ? bk0[e] : bk1[e - 4]));
      }
      const bool real = 16 * c + 8 * hi + 8 <= D;          // this lane's chunk c holds real channels
      if (!real) {
#pragma unroll
        for (int e = 0; e < 8; ++e) qv[e] = (bf16)0.0f;     // slot D (-m) is written by the attention loop
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          qs += (float)qv[e] * (float)qv[e];
          ks += (float)kv[e] * (float)kv[e];
        }
      }
      qf[tt][c] = qv;
      if (real) st16<bf16x8>(sK + (tok0 + ln) * RK + (2 * c + hi) * 16, kv);
    }
    qn2[tt] = qs + __shfl_xor(qs, 32, 64);
    kn2 = fmaxf(kn2, ks + __shfl_xor(ks, 32, 64));
    if (ln < D) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        bf16x8 vv;
#pragma unroll
        for (int e = 0; e < 8; ++e) vv[e] = (bf16)av[8 * j + e];
        st16<bf16x8>(sV + ((((tok0 >> 5) * 2 + j) * 2 + hi) * D + ln) * 16, vv);
      }
    }
  }
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) kn2 = fmaxf(kn2, __shfl_xor(kn2, o, 64));
  if (lane == 0) sKN[wave] = kn2;
  stamp(3);
  __syncthreads();      // K / V^T of every token resident
  stamp(4);
  float kmax2 = 0.f;
#pragma unroll
  for (int w = 0; w < NW; ++w) kmax2 = fmaxf(kmax2, sKN[w]);
  // (experiment switch AFLDM_ATTNF_STAGGER: the second-dispatched half of the waves starts the attention phase late so
  //  that the two waves of a SIMD do not run their matrix / vector segments in step - measured: no effect; the first pass
  //  of a wave pair looks 1.6x longer than the second in the stamps only because the older wave of a SIMD wins the
  //  arbitration and finishes both passes first, the sum is what it would be either way)
  if (wave >= NW / 2)
    for (int i = 0; i < p.stagger; ++i) __builtin_amdgcn_s_sleep(1);

  // ------------------------------------------------------------------ phase B: attention over the resident K / V^T
  // per-lane fragment addresses (constants for padding lanes: step 0)
  const char* k0p = sK + ln * RK + hi * 16;                       // chunk 0: channels 8 half + e (D >= 16)
  const bool k1real = 16 + 8 * hi + 8 <= D;
  const char* k1p = k1real ? sK + ln * RK + 32 : sC + (hi == CF::PADHI ? 0 : 32);
  const int k1step = k1real ? 32 * RK : 0;
  const bool vreal = ln < D;
  const char* vp = vreal ? sV + (hi * D + ln) * 16 : sC + (ln == D ? 16 : 32);
  const int vstep_j = vreal ? 2 * D * 16 : 0, vstep_t = 2 * vstep_j;
  constexpr int LR = (D / 8) * 4;                                  // accumulator register of row D (the row sums), half 0

#pragma unroll
  for (int pass = 0; pass < TPW / NU; ++pass) {
    bf16x8 q[NU][2];
    f32x16 oacc[NU];
    float m_run[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      q[u][0] = qf[pass * NU + u][0];
      q[u][1] = qf[pass * NU + u][1];
      m_run[u] = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[u][r] = 0.f;
    }
    auto qk = [&](int t, f32x16 (&s)[NU]) {
      const bf16x8 kf0 = ld16<bf16x8>(k0p + t * 32 * RK);
      const bf16x8 kf1 = ld16<bf16x8>(k1p + t * k1step);
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        f32x16 z;
#pragma unroll
        for (int r = 0; r < 16; ++r) z[r] = 0.f;
        z = mfma32(kf0, q[u][0], z);
        s[u] = mfma32(kf1, q[u][1], z);
      }
    };
    // move the reference of query tile u to m_new (rounded to bf16: it travels in Q).  `cur` / `nxt`: score tiles already
    // computed against the old reference (the next tile's MFMAs are issued one step ahead)
    auto rescale = [&](int u, float mnew_raw, f32x16& cur, f32x16& nxt, bool first) {
      const float m_new = (float)(bf16)mnew_raw;
      const float delta = m_new - m_run[u];
      m_run[u] = m_new;
      if (hi == CF::PADHI) q[u][CF::PADC][0] = (bf16)(-m_new);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        cur[r] -= delta;
        nxt[r] -= delta;                                 // (without a next tile: dead values)
      }
      if (!first) {                                     // (tile 0: O is still zero, and 2^-delta may overflow)
        const float alpha = __builtin_amdgcn_exp2f(-delta);
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[u][r] *= alpha;
      }
    };
    // O^T += V^T P^T for key tile t, P = 2^(S^T - m) of `cur`
    auto softmax_pv = [&](int t, f32x16 (&cur)[NU]) {
      const bf16x8 vf0 = ld16<bf16x8>(vp + t * vstep_t);
      const bf16x8 vf1 = ld16<bf16x8>(vp + t * vstep_t + vstep_j);
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        bf16x8 pb0, pb1;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          pb0[e] = (bf16)((DBG & 2) ? cur[u][e] : __builtin_amdgcn_exp2f(cur[u][e]));
          pb1[e] = (bf16)((DBG & 2) ? cur[u][8 + e] : __builtin_amdgcn_exp2f(cur[u][8 + e]));
        }
        oacc[u] = mfma32(vf0, pb0, oacc[u]);
        oacc[u] = mfma32(vf1, pb1, oacc[u]);
      }
    };
    // one key tile of the loop that tracks the row maxima: tile t + 1 is produced into `nxt` FIRST (matrix pipe), then
    // the lazy-rescale test of tile t, its exponentials and products
    auto step = [&](int t, f32x16 (&cur)[NU], f32x16 (&nxt)[NU]) {
      if (t + 1 < NT) qk(t + 1, nxt);
      float mx[NU];
      bool grow = false;
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        float m = cur[u][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) m = fmaxf(m, cur[u][r]);
        mx[u] = m;
        grow |= m > ATT_TAU;
      }
      if (__any(grow)) {                                 // wave-uniform, rare after tile 0
#pragma unroll
        for (int u = 0; u < NU; ++u) {
          const float m = fmaxf(mx[u], __shfl_xor(mx[u], 32, 64));
          rescale(u, m_run[u] + fmaxf(m, 0.f), cur[u], nxt[u], false);
        }
      }
      softmax_pv(t, cur);
    };
    f32x16 sa[NU], sb[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) sb[u][r] = 0.f;
    qk(0, sa);
    // tile 0: the reference is the row maximum of the first tile.  Scores can never exceed |q| max|k| (Cauchy-Schwarz on
    // the very bf16 rows the MFMAs contract): when that bound is within 2^80 of the reference for every query of the
    // wave - any realistic input - NOTHING can overflow and the loop needs neither maxima nor rescaling
    bool safe = !p.force_slow;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      float m = sa[u][0];
#pragma unroll
      for (int r = 1; r < 16; ++r) m = fmaxf(m, sa[u][r]);
      m = fmaxf(m, __shfl_xor(m, 32, 64));
      rescale(u, m, sa[u], sb[u], true);
      const float bound = __builtin_sqrtf(qn2[pass * NU + u] * kmax2) * 1.002f + 0.01f;
      safe = safe && (bound - m_run[u] <= 80.0f);
    }
    static_assert(NT % 2 == 0 && NT >= 2, "key tiles are walked in pairs (score registers ping-pong)");
    // The bounded loop, hand-ordered: per key tile 4 NU matrix instructions (S^T of tile t + 1, P V of tile t) of 32
    // cycles each against 16 NU exponentials (~8.5 cycles) + 8 NU conversions - the vector pipe is the longer one, so
    // every MFMA is followed by ITS share of vector work (independent of it: the exponentials of tile t, while the
    // MFMA runs) and the scheduler is told to keep that order (sched_group_barrier): left to itself it clusters the
    // S^T MFMAs right in front of their first use and the wave then waits out the matrix pipe with nothing to issue.
    auto fast_step = [&](int t, f32x16 (&cur)[NU], f32x16 (&nxt)[NU], bool more) {
      bf16x8 kf0, kf1;
      if (more) {
        kf0 = ld16<bf16x8>(k0p + (t + 1) * 32 * RK);
        kf1 = ld16<bf16x8>(k1p + (t + 1) * k1step);
      }
      const bf16x8 vf0 = ld16<bf16x8>(vp + t * vstep_t);
      const bf16x8 vf1 = ld16<bf16x8>(vp + t * vstep_t + vstep_j);
      auto expc = [&](const f32x16& sc, int j) {          // P chunk j of one query tile
        bf16x8 pb;
#pragma unroll
        for (int e = 0; e < 8; ++e) pb[e] = (bf16)((DBG & 2) ? sc[8 * j + e] : __builtin_amdgcn_exp2f(sc[8 * j + e]));
        return pb;
      };
      f32x16 z;
#pragma unroll
      for (int r = 0; r < 16; ++r) z[r] = 0.f;
#define AF_FENCE() __builtin_amdgcn_sched_barrier(0)
      // (each MFMA and the vector work issued under it form one scheduling region, in this order)
      if constexpr (NU == 2) {
        AF_FENCE();
        if (more) nxt[0] = mfma32(kf0, q[0][0], z);
        const bf16x8 p00 = expc(cur[0], 0);
        AF_FENCE();
        if (more) nxt[0] = mfma32(kf1, q[0][1], nxt[0]);
        const bf16x8 p01 = expc(cur[0], 1);
        AF_FENCE();
        oacc[0] = mfma32(vf0, p00, oacc[0]);
        const bf16x8 p10 = expc(cur[1], 0);
        AF_FENCE();
        oacc[0] = mfma32(vf1, p01, oacc[0]);
        if (more) nxt[1] = mfma32(kf0, q[1][0], z);
        const bf16x8 p11 = expc(cur[1], 1);
        AF_FENCE();
        oacc[1] = mfma32(vf0, p10, oacc[1]);
        if (more) nxt[1] = mfma32(kf1, q[1][1], nxt[1]);
        oacc[1] = mfma32(vf1, p11, oacc[1]);
      } else {
        AF_FENCE();
        if (more) nxt[0] = mfma32(kf0, q[0][0], z);
        const bf16x8 p00 = expc(cur[0], 0);
        AF_FENCE();
        if (more) nxt[0] = mfma32(kf1, q[0][1], nxt[0]);
        const bf16x8 p01 = expc(cur[0], 1);
        AF_FENCE();
        oacc[0] = mfma32(vf0, p00, oacc[0]);
        oacc[0] = mfma32(vf1, p01, oacc[0]);
      }
#undef AF_FENCE
      __builtin_amdgcn_sched_barrier(0);                   // a step is one scheduling region: nothing moves across
    };
    if (DBG & 1) {
      softmax_pv(0, sa);
    } else if (__all(safe)) {
      for (int t = 0; t + 2 < NT; t += 2) {
        fast_step(t, sa, sb, true);
        fast_step(t + 1, sb, sa, true);
      }
      fast_step(NT - 2, sa, sb, true);
      fast_step(NT - 1, sb, sa, false);
    } else {
      for (int t = 0; t < NT; t += 2) {
        step(t, sa, sb);
        step(t + 1, sb, sa);
      }
    }
    stamp(5 + (pass > 0 ? 1 : 0));
    // ---- finish: row D of O^T is the softmax denominator (half 0 holds it); normalise, store 4 channels per piece
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      float l = oacc[u][LR];
      const float lo = __shfl_xor(l, 32, 64);
      if (hi != 0) l = lo;
      const float inv = 1.0f / l;
      const int qrow = (wave * TPW + pass * NU + u) * 32 + ln;
      bf16* op = p.o + ((size_t)b * T + qrow) * C + h * D;
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int d0 = 8 * g4 + 4 * hi;
        if (d0 + 4 <= D)
          store4<bf16>(op + d0, oacc[u][4 * g4] * inv, oacc[u][4 * g4 + 1] * inv, oacc[u][4 * g4 + 2] * inv,
                       oacc[u][4 * g4 + 3] * inv);
      }
    }
  }
  stamp(7);

  // ------------------------------------------------------------------ phase C: to_out + residual + statistics
  // The heads of a sample are `heads` workgroups with consecutive work ids on ONE XCD (xcd_remap), dispatched together.  Each
  // has just stored its head's 24 channels of o for ALL tokens; once the sample's workgroups have all arrived (one L2 atomic
  // each; their stores are in the XCD's L2 when acknowledged), workgroup h owns the token block [h T / heads, (h + 1) T / heads)
  // of the sample and finishes the attention block there:  y = o Wo^T + bias + x,  with the per-channel partial sums of the
  // rounded y for the next GroupNorm (split h of `heads`).  What the stand-alone to_out launch did - a 75 MB pass at
  // 32 x 32 - becomes ~5 us at the end of a launch whose o rows are still in the L2.  A workgroup waits only for workgroups
  // of its own sample, which the dispatcher placed before or together with it: no cluster is ever partly resident for long.
  if constexpr (OUT) {
    constexpr int HEADS = C / D, TB = T / HEADS, NTT = TB / 32, NCH = NW / NTT, CW = C / NCH, NTL = CW / 32;
    constexpr int RWO = C * 2 + 16, WO_BYTES = C * RWO, OROW = C + 8;
    constexpr int WPT = (C * C / 8) / NTHR;                    // 16-byte pieces of Wo per thread
    static_assert(T % HEADS == 0 && TB % 32 == 0 && NTT * NCH == NW && CW % 32 == 0 && (C * C / 8) % NTHR == 0, "phase C tiling");
    static_assert(WO_BYTES + TB * OROW * 2 <= CF::LDS_BYTES, "Wo + staging tile fit the attention phase's LDS");
    char* sWo = smem;
    bf16* sO = reinterpret_cast<bf16*>(smem + WO_BYTES);
    // Wo -> registers while the other waves finish (independent of the siblings)
    bf16x8 wreg[WPT];
#pragma unroll
    for (int j = 0; j < WPT; ++j) wreg[j] = ld16<bf16x8>(p.wo + (size_t)(tid + NTHR * j) * 8);
    // ... and so are the residual rows (x left the L2 long ago: their latency runs under the wait for the siblings).
    // wave (token tile tt, cout slice chh): [channel x token] tiles, rows permuted by sigma so that registers 8c .. 8c+7 of
    // lane (token, half) are the 8 consecutive channels 16c + 8 half + e of the tile
    const int tt = wave % NTT, chh = wave / NTT;
    const int trow = h * TB + tt * 32 + ln;                    // this lane's token (within the sample)
    const bf16* xres = p.x + ((size_t)b * T + trow) * C;
    bf16x8 rx[NTL][2];
#pragma unroll
    for (int tl = 0; tl < NTL; ++tl)
#pragma unroll
      for (int c = 0; c < 2; ++c) rx[tl][c] = ld16<bf16x8>(xres + chh * CW + tl * 32 + 16 * c + 8 * hi);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this thread's o stores are acknowledged by the L2
    __syncthreads();                                           // K / V^T are dead, every wave's stores are out
    if (tid == 0) {
      unsigned* cnt = p.sync + (size_t)b * 32;
      __hip_atomic_fetch_or(cnt + 2, 1u << attnf_xcc_id(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
#pragma unroll
    for (int j = 0; j < WPT; ++j) {
      const int i = tid + NTHR * j, row = i / (C / 8), pc = i - row * (C / 8);
      st16<bf16x8>(sWo + row * RWO + (pc << 4), wreg[j]);
    }
    if (tid == 0) {
      unsigned* cnt = p.sync + (size_t)b * 32;
      unsigned spins = 0;
      while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)HEADS) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1u << 22)) {                            // never hang the device on a lost sibling
          __hip_atomic_store(p.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
      }
      const unsigned old = __hip_atomic_fetch_add(cnt + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (old == (unsigned)HEADS - 1) {                        // last one out: the line returns to zero for the next launch
        const unsigned xm = __hip_atomic_exchange(cnt + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (xm & (xm - 1)) __hip_atomic_store(p.err, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_exchange(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_exchange(cnt + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
    __syncthreads();
    stamp(10);
    // The siblings' o rows: published by plain stores acknowledged by the XCD's L2 and awaited through relaxed atomics.  The loads below
    // are the ACQUIRE side.  INVARIANT they rely on: nothing in this kernel reads p.o before this point (it is write-only up to the
    // hand-over), so this CU's vector L1 cannot hold a stale line of it and the lines come from the L2 the siblings wrote through.
    // A change that touches p.o earlier (a prefetch, a second token block per workgroup) MUST switch to the sc1 form
    // (-DAFLDM_ATTNF_ACQUIRE_SC1: agent-scope loads that are never served by the L1).  That form is not the default because it is
    // measurably slower: a lane's 12 pieces share their 128-byte lines with the other half-row's lanes, through the L1 one miss
    // serves eight pieces, with sc1 every piece is its own L2 request - 7.70 against 6.39 us for this phase, +2.8 us per launch,
    // +0.013 ms/step (profiles/r06/attnf_sc1_ab.txt).
    const bf16* orow = p.o + ((size_t)b * T + trow) * C + hi * 8;
    bf16x8 of[CK];
#if !defined(AFLDM_ATTNF_ACQUIRE_SC1)      // (default: plain loads - see the comment above)
#pragma unroll
    for (int kk = 0; kk < CK; ++kk) of[kk] = ld16<bf16x8>(orow + kk * 16);
#else
    typedef __attribute__((ext_vector_type(4))) unsigned int u4;
    u4 oraw[CK];
#pragma unroll
    for (int kk = 0; kk < CK; ++kk) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(oraw[kk]) : "v"(orow + kk * 16) : "memory");
#endif
    // (sc1 form: the bias loads go out behind the o rows and share their round trip; the wait sits after the accumulator set-up)
    f32x16 acc[NTL];
#pragma unroll
    for (int tl = 0; tl < NTL; ++tl) {
      const int c0 = chh * CW + tl * 32;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const f32x4 b0 = ld16<f32x4>(p.bias_o + c0 + 16 * c + 8 * hi), b1 = ld16<f32x4>(p.bias_o + c0 + 16 * c + 8 * hi + 4);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[tl][8 * c + e] = (float)rx[tl][c][e] + (e < 4 ? b0[e] : b1[e - 4]);
      }
    }
#if defined(AFLDM_ATTNF_ACQUIRE_SC1)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int kk = 0; kk < CK; ++kk) {
      asm volatile("" : "+v"(oraw[kk]));                      // (ties every use to the wait above)
      of[kk] = __builtin_bit_cast(bf16x8, oraw[kk]);
    }
#endif
#pragma unroll
    for (int kk = 0; kk < CK; ++kk) {
#pragma unroll
      for (int tl = 0; tl < NTL; ++tl) {
        const bf16x8 wf = ld16<bf16x8>(sWo + (chh * CW + tl * 32 + sig) * RWO + ((2 * kk + hi) << 4));
        acc[tl] = mfma32(wf, of[kk], acc[tl]);
      }
    }
#pragma unroll
    for (int tl = 0; tl < NTL; ++tl)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        bf16x8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (bf16)acc[tl][8 * c + e];
        st16<bf16x8>(sO + (tt * 32 + ln) * OROW + chh * CW + tl * 32 + 16 * c + 8 * hi, v);
      }
    __syncthreads();                                           // tile staged; Wo is dead
    // whole rows out, 16 bytes per lane; per-channel sums of the stored values by the thread that owns the column
    constexpr int CPR = C / 8, RPI = NTHR / CPR;
    const bool active = tid < RPI * CPR;
    const int ch = tid % CPR, tr = tid / CPR;
    float ss1[8], ss2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) ss1[e] = ss2[e] = 0.f;
    if (active) {
      bf16* yrow = p.y + ((size_t)b * T + h * TB) * C + ch * 8;
#pragma unroll 4
      for (int row = tr; row < TB; row += RPI) {
        const bf16x8 v = ld16<bf16x8>(sO + row * OROW + ch * 8);
        st16_wt<bf16x8>(yrow + (size_t)row * C, v);        // whole rows, write-through (nothing left dirty for the release)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float vr = (float)v[e];
          ss1[e] += vr;
          ss2[e] = fmaf(vr, vr, ss2[e]);
        }
      }
    }
    float* sR = reinterpret_cast<float*>(smem);                // [RPI][C][2] over the Wo region
    static_assert(RPI * C * 8 <= WO_BYTES, "statistics scratch");
    if (active) {
#pragma unroll
      for (int e = 0; e < 8; ++e) *reinterpret_cast<f32x2*>(sR + ((tr * C) + ch * 8 + e) * 2) = f32x2{ss1[e], ss2[e]};
    }
    __syncthreads();
    for (int c = tid; c < C; c += NTHR) {
      float a1 = 0.f, a2 = 0.f;
      for (int r = 0; r < RPI; ++r) {
        const f32x2 v = *reinterpret_cast<const f32x2*>(sR + ((r * C) + c) * 2);
        a1 += v[0];
        a2 += v[1];
      }
      *reinterpret_cast<f32x2*>(p.stats_out + (((size_t)b * HEADS + h) * C + c) * 2) = f32x2{a1, a2};
    }
    stamp(11);
  }
}

template <int D, int NW, int TPW, int CK, int DBG = 0, bool LEAN = false, bool OUT = false>
static int attnf_launch(const AttnFP& p, hipStream_t st) {
  typedef AttnFCfg<D, NW, TPW, CK, LEAN> CF;
  static unsigned long long once = 0;
  if (first_on_device(once)) {
    (void)hipFuncSetAttribute((const void*)k_attn_fused<D, NW, TPW, CK, DBG, LEAN, OUT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              CF::LDS_BYTES);
  }
  k_attn_fused<D, NW, TPW, CK, DBG, LEAN, OUT><<<p.B * p.heads, NW * 64, CF::LDS_BYTES, st>>>(p);
  return check_launch("afldm_attn_block_fused");
}

template <int D, int NW, int TPW, int CK, bool LEAN = false>
static int attnf_launch_dbg(const AttnFP& p, hipStream_t st) {
  static const int dbg = getenv("AFLDM_ATTNF_DBG") ? atoi(getenv("AFLDM_ATTNF_DBG")) : 0;
  switch (dbg) {
    case 1: return attnf_launch<D, NW, TPW, CK, 1, LEAN>(p, st);
    case 2: return attnf_launch<D, NW, TPW, CK, 2, LEAN>(p, st);
    case 4: return attnf_launch<D, NW, TPW, CK, 4, LEAN>(p, st);
    case 8: return attnf_launch<D, NW, TPW, CK, 8, LEAN>(p, st);
    case 16: return attnf_launch<D, NW, TPW, CK, 16, LEAN>(p, st);
    case 24: return attnf_launch<D, NW, TPW, CK, 24, LEAN>(p, st);
    default: return attnf_launch<D, NW, TPW, CK, 0, LEAN>(p, st);
  }
}

}  // namespace afldm

using namespace afldm;

// diagnostic: device buffer of [workgroups][waves][12] uint64 that the next launches fill with s_memtime stamps
// (0 start, 1 / 2 before / after the prologue barrier, 3 / 4 before / after the barrier that ends the projection,
//  5 / 6 end of the first / second attention pass); NULL switches it off
extern "C" int afldm_attn_block_fused_trace(void* buf) {
  g_attnf_trace = (unsigned long long*)buf;
  return AFLDM_OK;
}

// 1 when afldm_attn_block_fused has a kernel for this shape
extern "C" int afldm_attn_block_fused_supported(int T, int C, int head_dim, int G) {
  if (G <= 0 || C % G || C / G > 16 || C % head_dim) return 0;
  if (head_dim == 24) return (T == 1024 && C == 192) || (T == 256 && C == 384);
  if (head_dim == 16) return (T == 256 && C == 64) || (T == 64 && C == 128);
  return 0;
}

// 1 when afldm_attn_block_fused_out has a kernel for this shape: the 32 x 32 level, and every sample's `heads` workgroups
// inside one XCD (work ids are dealt to the 8 XCDs in equal contiguous runs: B a multiple of 8)
extern "C" int afldm_attn_block_fused_out_supported(int B, int T, int C, int head_dim, int G) {
  if (!afldm_attn_block_fused_supported(T, C, head_dim, G)) return 0;
  if (!(head_dim == 24 && T == 1024 && C == 192)) return 0;
  return B > 0 && B <= 512 && B % 8 == 0;
}

static int attn_block_fused_impl(const void* x, const float* stats, int S, const float* gamma, const float* beta,
                                 int G, float eps, const void* w_qkv, const float* bias_qkv, void* o, const void* w_out,
                                 const float* bias_out, void* y, float* stats_out, void* sync, long long sync_bytes, int B,
                                 int T, int C, int heads, float scale, int dtype, afldm_stream_t stream);

extern "C" int afldm_attn_block_fused_out(const void* x, const float* stats, int S, const float* gamma, const float* beta,
                                          int G, float eps, const void* w_qkv, const float* bias_qkv, void* o,
                                          const void* w_out, const float* bias_out, void* y, float* stats_out, void* sync,
                                          long long sync_bytes, int B, int T, int C, int heads, float scale, int dtype,
                                          afldm_stream_t stream) {
  AFLDM_REQUIRE(w_out && bias_out && y && stats_out && sync, AFLDM_ENULL, "afldm_attn_block_fused_out: NULL pointer");
  AFLDM_REQUIRE(heads > 0 && C % heads == 0 && afldm_attn_block_fused_out_supported(B, T, C, C / heads, G), AFLDM_ESHAPE,
                "afldm_attn_block_fused_out: no kernel for B=%d T=%d C=%d heads=%d groups=%d", B, T, C, heads, G);
  AFLDM_REQUIRE(sync_bytes >= (long long)(16384 + 32 * B) * 4, AFLDM_ESHAPE, "afldm_attn_block_fused_out: sync buffer too small");
  AFLDM_REQUIRE(aligned16(w_out) && aligned16(y) && aligned16(bias_out), AFLDM_EALIGN, "afldm_attn_block_fused_out: pointers must be 16-byte aligned");
  return attn_block_fused_impl(x, stats, S, gamma, beta, G, eps, w_qkv, bias_qkv, o, w_out, bias_out, y, stats_out, sync, sync_bytes,
                               B, T, C, heads, scale, dtype, stream);
}

extern "C" int afldm_attn_block_fused(const void* x, const float* stats, int S, const float* gamma, const float* beta,
                                      int G, float eps, const void* w_qkv, const float* bias_qkv, void* o, int B, int T,
                                      int C, int heads, float scale, int dtype, afldm_stream_t stream) {
  return attn_block_fused_impl(x, stats, S, gamma, beta, G, eps, w_qkv, bias_qkv, o, nullptr, nullptr, nullptr, nullptr, nullptr, 0,
                               B, T, C, heads, scale, dtype, stream);
}

static int attn_block_fused_impl(const void* x, const float* stats, int S, const float* gamma, const float* beta,
                                 int G, float eps, const void* w_qkv, const float* bias_qkv, void* o, const void* w_out,
                                 const float* bias_out, void* y, float* stats_out, void* sync, long long sync_bytes, int B,
                                 int T, int C, int heads, float scale, int dtype, afldm_stream_t stream) {
  AFLDM_REQUIRE(x && stats && gamma && beta && w_qkv && bias_qkv && o, AFLDM_ENULL, "afldm_attn_block_fused: NULL pointer");
  AFLDM_REQUIRE(dtype == AFLDM_BF16, AFLDM_EDTYPE, "afldm_attn_block_fused: bf16 only (fp32 runs gn_apply + linear + attention)");
  AFLDM_REQUIRE(B > 0 && heads > 0 && C % heads == 0 && S > 0, AFLDM_ESHAPE, "afldm_attn_block_fused: bad shape B=%d heads=%d C=%d S=%d", B, heads, C, S);
  const int d = C / heads;
  AFLDM_REQUIRE(afldm_attn_block_fused_supported(T, C, d, G), AFLDM_ESHAPE,
                "afldm_attn_block_fused: no kernel for T=%d C=%d head_dim=%d groups=%d", T, C, d, G);
  AFLDM_REQUIRE(aligned16(x) && aligned16(w_qkv) && aligned16(o), AFLDM_EALIGN, "afldm_attn_block_fused: pointers must be 16-byte aligned");
  AttnFP p;
  p.x = (const bf16*)x; p.w = (const bf16*)w_qkv; p.bias = bias_qkv; p.gamma = gamma; p.beta = beta;
  p.gs.st1 = stats; p.gs.st2 = nullptr; p.gs.C1 = C; p.gs.C2 = 0; p.gs.S1 = S; p.gs.S2 = 0;
  p.trace = g_attnf_trace;
  p.o = (bf16*)o; p.B = B; p.heads = heads; p.C = C; p.G = G; p.eps = eps;
  p.qscale = scale * 1.4426950408889634f;
  p.wo = (const bf16*)w_out; p.bias_o = bias_out; p.y = (bf16*)y; p.stats_out = stats_out;
  p.sync = sync ? (unsigned*)sync + 16384 : nullptr;
  p.err = sync ? (unsigned*)sync + 8193 : nullptr;
  (void)sync_bytes;
  {
    static const int stg = getenv("AFLDM_ATTNF_STAGGER") ? atoi(getenv("AFLDM_ATTNF_STAGGER")) : 0;      // (measured: no effect)
    p.stagger = stg;
  }
  {
    const char* e = getenv("AFLDM_ATTNF_SLOW");      // read per call: tests flip it (row-maxima loop instead of the bounded one)
    p.force_slow = e && atoi(e) != 0;
  }
  hipStream_t st = (hipStream_t)stream;
  if (d == 24 && T == 1024 && w_out) return attnf_launch<24, 8, 4, 12, 0, false, true>(p, st);
  if (d == 24 && T == 1024) return attnf_launch_dbg<24, 8, 4, 12>(p, st);
  if (d == 24 && T == 256) {
    // 16x16 level: two four-wave workgroups per CU (80 KiB each) unless AFLDM_ATTNF_L16=8 asks for the eight-wave form
    static const int l16 = getenv("AFLDM_ATTNF_L16") ? atoi(getenv("AFLDM_ATTNF_L16")) : 4;
    if (l16 == 8) return attnf_launch_dbg<24, 8, 1, 24>(p, st);
    return attnf_launch_dbg<24, 4, 2, 24, true>(p, st);
  }
  if (d == 16 && T == 256) return attnf_launch<16, 8, 1, 4>(p, st);
  if (d == 16 && T == 64) return attnf_launch<16, 2, 1, 8>(p, st);
  set_error("afldm_attn_block_fused: unreachable shape");
  return AFLDM_ESHAPE;
}
