// gn.hip — GroupNorm statistics and apply(+SiLU) over NHWC tensors that may be the virtual
// channel-concat of two tensors.  HBM-bound: algorithmic bytes = 1 read (stats; the second
// centred pass re-reads from L2) and 1 read + 1 write (apply) of the logical tensor.
#include <stdlib.h>

#include "common.hpp"

namespace afldm {

// ---- statistics -------------------------------------------------------------------------------
// Stand-alone producer of the per-channel partial sums (see GnStats in common.hpp) for tensors that
// do not come out of afldm_conv2d (which emits them from its epilogue): workgroup (b, s) reads the
// pixels of split s of sample b with FULL-ROW coalesced vector loads (4 channels per lane), reduces
// per channel across its pixel lanes in a fixed order through LDS and writes part[b][s][c][2].
template <typename T>
__global__ void __launch_bounds__(256) k_gn_partial(const T* __restrict__ x, int C, float* __restrict__ part, int HW, int S) {
  extern __shared__ float lds[];  // [ppl][C][2]
  const int nq = C / 4;                        // channel quads per pixel row
  const int tpr = nq < 256 ? nq : 256;         // threads per pixel row
  const int ppl = 256 / tpr;                   // pixel lanes
  const int b = blockIdx.x / S, sp = blockIdx.x % S;
  const int p0 = (int)(((long long)HW * sp) / S), p1 = (int)(((long long)HW * (sp + 1)) / S);
  const int tid = threadIdx.x;
  const int pl = tid / tpr;
  if (pl < ppl) {
    for (int q = tid - pl * tpr; q < nq; q += tpr) {
      float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
      const int c = 4 * q;
      const T* src = x + (size_t)b * HW * C + c;
      int pix = p0 + pl;
      // 4 independent loads in flight per lane
      for (; pix + 3 * ppl < p1; pix += 4 * ppl) {
        float v[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) load4<T>(src + (size_t)(pix + u * ppl) * C, v[u][0], v[u][1], v[u][2], v[u][3]);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            s1[e] += v[u][e];
            s2[e] = fmaf(v[u][e], v[u][e], s2[e]);
          }
      }
      for (; pix < p1; pix += ppl) {
        float v[4];
        load4<T>(src + (size_t)pix * C, v[0], v[1], v[2], v[3]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          s1[e] += v[e];
          s2[e] = fmaf(v[e], v[e], s2[e]);
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        lds[((size_t)pl * C + c + e) * 2 + 0] = s1[e];
        lds[((size_t)pl * C + c + e) * 2 + 1] = s2[e];
      }
    }
  }
  __syncthreads();
  for (int c = tid; c < C; c += 256) {
    float a1 = 0.f, a2 = 0.f;
    for (int l = 0; l < ppl; ++l) {
      a1 += lds[((size_t)l * C + c) * 2 + 0];
      a2 += lds[((size_t)l * C + c) * 2 + 1];
    }
    *reinterpret_cast<f32x2*>(part + (((size_t)b * S + sp) * C + c) * 2) = f32x2{a1, a2};
  }
}

template <typename T>
__global__ void __launch_bounds__(256) k_gn_apply(const T* __restrict__ x1, int C1, const T* __restrict__ x2, int C2,
                                                  GnStats gs, const float* __restrict__ gamma,
                                                  const float* __restrict__ beta, T* __restrict__ y, int HW, int G,
                                                  float eps, int act, int rows_per_block) {
  // one workgroup = `rows_per_block` pixels of ONE sample: per-channel scale/shift once in LDS,
  // then 4 channels per lane, coalesced
  extern __shared__ float lds[];  // [C] scale, [C] shift
  const int C = C1 + C2, cpg = C / G;
  const int blocks_per_sample = (HW + rows_per_block - 1) / rows_per_block;
  const int b = blockIdx.x / blocks_per_sample;
  const int r0 = (blockIdx.x % blocks_per_sample) * rows_per_block;
  float* sc = lds;
  float* sh = lds + C;
  typedef typename Mma<T>::Chunk Chunk;
  constexpr int EPC = Mma<T>::EPC;
  const int r1 = r0 + rows_per_block < HW ? r0 + rows_per_block : HW;
  // Everything that does not depend on the statistics is REQUESTED first - gamma / beta of the channels this thread
  // will tabulate and the first four rows of its channel column - so that the kernel is one load round trip deep
  // instead of three (partial sums -> gamma / beta -> data: ~1.5 us each on the launch-bound small levels).
  constexpr int GPT = 4;                                    // table entries per thread kept in registers (C <= 1024)
  const bool table_regs = C <= GPT * (int)blockDim.x;
  float gmr[GPT], btr[GPT];
  if (table_regs) {
#pragma unroll
    for (int k = 0; k < GPT; ++k) {
      const int c = (int)threadIdx.x + k * (int)blockDim.x;
      gmr[k] = c < C ? gamma[c] : 0.f;
      btr[k] = c < C ? beta[c] : 0.f;
    }
  }
  // Streaming form: a thread keeps ONE 16-byte channel column (its scale / shift live in registers)
  // and walks down the rows with four independent loads in flight - no index division, no LDS
  // reads in the loop (8-byte accesses run at 0.54-0.70x the 16-byte rate, MI355X_MICROARCH.md).
  const bool streaming = C1 % EPC == 0 && C2 % EPC == 0 && C / EPC <= (int)blockDim.x;
  const int nc = streaming ? C / EPC : 1, rl = (int)blockDim.x / nc;
  const bool mine = streaming && (int)threadIdx.x < rl * nc;
  const int col = (int)threadIdx.x % nc, lane_r = (int)threadIdx.x / nc, cc0 = EPC * col;
  const bool second0 = cc0 >= C1;
  const T* src = (second0 ? x2 : x1) + (size_t)b * HW * (second0 ? C2 : C1) + (second0 ? cc0 - C1 : cc0);
  const size_t Cs0 = second0 ? C2 : C1;
  int row = r0 + lane_r;
  Chunk v0[4];
  const bool have0 = mine && row + 3 * rl < r1;
  if (have0) {
#pragma unroll
    for (int u = 0; u < 4; ++u) v0[u] = ld16<Chunk>(src + (size_t)(row + u * rl) * Cs0);
  }
  // group statistics: 8 lanes per group (each adds every 8th channel of the group), G <= 32 per round
  for (int g0 = 0; g0 < G; g0 += 32) {
    const int g = g0 + (threadIdx.x >> 3), part = threadIdx.x & 7;
    double s1 = 0.0, s2 = 0.0;
    if (g < G)
      for (int c = g * cpg + part; c < (g + 1) * cpg; c += 8) gn_channel_sums(gs, b, c, s1, s2);
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {
      s1 += __shfl_xor(s1, o, 64);
      s2 += __shfl_xor(s2, o, 64);
    }
    if (g < G && part == 0) {
      float mean, rstd;
      gn_mean_rstd(s1, s2, (double)HW * cpg, eps, mean, rstd);
      sc[C + C + 2 * g] = mean;       // scratch behind the two tables
      sc[C + C + 2 * g + 1] = rstd;
    }
  }
  __syncthreads();
  if (table_regs) {
#pragma unroll
    for (int k = 0; k < GPT; ++k) {
      const int c = (int)threadIdx.x + k * (int)blockDim.x;
      if (c < C) {
        const float mean = sc[C + C + 2 * (c / cpg)], rstd = sc[C + C + 2 * (c / cpg) + 1];
        const float kk = rstd * gmr[k];
        sc[c] = kk;
        sh[c] = btr[k] - mean * kk;
      }
    }
  } else {
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
      const float mean = sc[C + C + 2 * (c / cpg)], rstd = sc[C + C + 2 * (c / cpg) + 1];
      const float k = rstd * gamma[c];
      sc[c] = k;
      sh[c] = beta[c] - mean * k;
    }
  }
  __syncthreads();
  if (streaming) {
    if (mine) {
      const int c = cc0;
      const size_t Cs = Cs0;
      T* dst = y + (size_t)b * HW * C + c;
      float ks[EPC], hs[EPC];
#pragma unroll
      for (int e = 0; e < EPC; ++e) {
        ks[e] = sc[c + e];
        hs[e] = sh[c + e];
      }
      auto emit = [&](const Chunk (&v)[4]) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          Chunk o;
#pragma unroll
          for (int e = 0; e < EPC; ++e) {
            float f = to_f32(v[u][e]) * ks[e] + hs[e];
            if (act == 1) f = silu_f(f);
            o[e] = from_f32<T>(f);
          }
          st16_out<Chunk>(dst + (size_t)(row + u * rl) * C, o);
        }
      };
      if (have0) {                                          // the batch requested in front of the statistics
        emit(v0);
        row += 4 * rl;
      }
      for (; row + 3 * rl < r1; row += 4 * rl) {
        Chunk v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = ld16<Chunk>(src + (size_t)(row + u * rl) * Cs);
        emit(v);
      }
      for (; row < r1; row += rl) {
        const Chunk v = ld16<Chunk>(src + (size_t)row * Cs);
        Chunk o;
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
          float f = to_f32(v[e]) * ks[e] + hs[e];
          if (act == 1) f = silu_f(f);
          o[e] = from_f32<T>(f);
        }
        st16_out<Chunk>(dst + (size_t)row * C, o);
      }
    }
    return;
  }
  const int nq = C / 4;
  const int total = (r1 - r0) * nq;
  for (int i = threadIdx.x; i < total; i += blockDim.x) {
    const int pix = r0 + i / nq, c = 4 * (i % nq);
    const bool second = c >= C1;
    const T* src = second ? x2 : x1;
    const int Cs = second ? C2 : C1, cs = second ? c - C1 : c;
    float v[4];
    load4<T>(src + ((size_t)b * HW + pix) * Cs + cs, v[0], v[1], v[2], v[3]);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[e] = v[e] * sc[c + e] + sh[c + e];
      if (act == 1) v[e] = silu_f(v[e]);
    }
    store4<T>(y + ((size_t)b * HW + pix) * C + c, v[0], v[1], v[2], v[3]);
  }
}

}  // namespace afldm

using namespace afldm;

static int gn_check(const char* fn, const void* x1, int C1, const void* x2, int C2, int B, int HW, int G) {
  AFLDM_REQUIRE(x1 != nullptr, AFLDM_ENULL, "%s: x1 is NULL", fn);
  AFLDM_REQUIRE(C1 > 0 && C2 >= 0 && (C2 == 0 || x2 != nullptr), AFLDM_ESHAPE, "%s: bad C1=%d C2=%d", fn, C1, C2);
  AFLDM_REQUIRE(B > 0 && HW > 0 && G > 0 && (C1 + C2) % G == 0, AFLDM_ESHAPE,
                "%s: C=%d not divisible by groups=%d (B=%d HW=%d)", fn, C1 + C2, G, B, HW);
  return AFLDM_OK;
}

// Fold S_in row splits of per-channel partial sums into S_out (S_in % S_out == 0, fixed order).  The GEMM
// epilogue emits one split per 128-pixel tile: 512 per sample on the 256^2 planes of the AF-VAE, and every
// consumer walks cpg x S partials per group in its prologue (k_gn_apply re-read 524 KB of partials per 65 KB
// of payload there).
__global__ void __launch_bounds__(256) k_gn_fold(const float* __restrict__ in, float* __restrict__ out, int C, int S_in,
                                                 int S_out, size_t total) {
  const int r = S_in / S_out;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const size_t bs = i / C;              // b * S_out + so
    const size_t b = bs / S_out, so = bs - b * S_out;
    const float* q = in + ((b * S_in + so * r) * C + c) * 2;
    float s1 = 0.f, s2 = 0.f;
    for (int k = 0; k < r; ++k) {
      const f32x2 v = *reinterpret_cast<const f32x2*>(q + (size_t)k * C * 2);
      s1 += v[0];
      s2 += v[1];
    }
    *reinterpret_cast<f32x2*>(out + i * 2) = f32x2{s1, s2};
  }
}

extern "C" int afldm_gn_fold(const float* stats_in, int S_in, float* stats_out, int S_out, int B, int C,
                             afldm_stream_t stream) {
  AFLDM_REQUIRE(stats_in && stats_out, AFLDM_ENULL, "afldm_gn_fold: NULL pointer");
  AFLDM_REQUIRE(B > 0 && C > 0 && S_in > 0 && S_out > 0 && S_in % S_out == 0, AFLDM_ESHAPE,
                "afldm_gn_fold: S_in=%d must be a multiple of S_out=%d", S_in, S_out);
  const size_t total = (size_t)B * S_out * C;
  const int grid = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
  k_gn_fold<<<grid, 256, 0, (hipStream_t)stream>>>(stats_in, stats_out, C, S_in, S_out, total);
  return check_launch("afldm_gn_fold");
}

extern "C" int afldm_gn_stats_splits(int HW) { return gn_splits(HW); }

extern "C" int afldm_gn_stats(const void* x, int C, float* part, int B, int HW, int dtype, afldm_stream_t stream) {
  AFLDM_REQUIRE(x != nullptr && part != nullptr, AFLDM_ENULL, "afldm_gn_stats: NULL pointer");
  AFLDM_REQUIRE(B > 0 && HW > 0 && C > 0 && C % 4 == 0, AFLDM_ESHAPE, "afldm_gn_stats: B=%d HW=%d C=%d (C must be a multiple of 4)", B, HW, C);
  hipStream_t st = (hipStream_t)stream;
  const int S = gn_splits(HW), nq = C / 4;
  const int ppl = 256 / (nq < 256 ? nq : 256);
  const size_t lds = (size_t)ppl * C * 2 * sizeof(float);
  AFLDM_REQUIRE(lds <= 64 * 1024, AFLDM_ESHAPE, "afldm_gn_stats: C=%d too wide", C);
  DISPATCH_T(dtype, (k_gn_partial<float><<<B * S, 256, lds, st>>>((const float*)x, C, part, HW, S)),
             (k_gn_partial<bf16><<<B * S, 256, lds, st>>>((const bf16*)x, C, part, HW, S)), "afldm_gn_stats");
  return check_launch("afldm_gn_stats");
}

extern "C" int afldm_gn_apply(const void* x1, int C1, const void* x2, int C2, const float* stats1, int S1,
                              const float* stats2, int S2, const float* gamma, const float* beta, void* y, int B,
                              int HW, int G, float eps, int act, int dtype, afldm_stream_t stream) {
  int rc = gn_check("afldm_gn_apply", x1, C1, x2, C2, B, HW, G);
  if (rc) return rc;
  AFLDM_REQUIRE(stats1 && S1 > 0 && (C2 == 0 || (stats2 && S2 > 0)) && gamma && beta && y, AFLDM_ENULL,
                "afldm_gn_apply: NULL pointer / bad split count");
  AFLDM_REQUIRE(act == 0 || act == 1, AFLDM_ESHAPE, "afldm_gn_apply: act %d not in {0,1}", act);
  AFLDM_REQUIRE(C1 % 4 == 0 && C2 % 4 == 0, AFLDM_ESHAPE, "afldm_gn_apply: C1=%d / C2=%d must be multiples of 4", C1, C2);
  hipStream_t st = (hipStream_t)stream;
  const int C = C1 + C2;
  static const int s_el = getenv("AFLDM_GN_ELEMS") ? atoi(getenv("AFLDM_GN_ELEMS")) : 0;
  int per_wg = s_el > 0 ? s_el : (HW >= 4096 ? 32768 : 16384);   // elements per workgroup (32K on the big VAE planes: amortises the statistics prologue)
  // a small tensor (small batch / the low levels) would be a handful of workgroups walking 64 elements per thread:
  // four times as many shorter ones (batch 1: 2.251 -> 2.226 ms/step; batch 64 unchanged)
  if (s_el <= 0 && (long long)B * HW * C < 128ll * per_wg) per_wg = 4096;
  int rows = per_wg / C;
  if (rows < 1) rows = 1;
  if (rows > HW) rows = HW;
  const int grid = B * ((HW + rows - 1) / rows);
  const size_t lds = (size_t)(2 * C + 2 * G) * sizeof(float);
  GnStats gs{stats1, stats2, C1, C2, S1, S2};
  DISPATCH_T(dtype,
             (k_gn_apply<float><<<grid, 256, lds, st>>>((const float*)x1, C1, (const float*)x2, C2, gs, gamma, beta,
                                                        (float*)y, HW, G, eps, act, rows)),
             (k_gn_apply<bf16><<<grid, 256, lds, st>>>((const bf16*)x1, C1, (const bf16*)x2, C2, gs, gamma, beta,
                                                       (bf16*)y, HW, G, eps, act, rows)),
             "afldm_gn_apply");
  return check_launch("afldm_gn_apply");
}
