// gn.hip — GroupNorm statistics and apply(+SiLU) over NHWC tensors that may be the virtual
// channel-concat of two tensors.  HBM-bound: algorithmic bytes = 1 read (stats; the second
// centred pass re-reads from L2) and 1 read + 1 write (apply) of the logical tensor.
#include "common.hpp"

namespace afldm {

template <typename T>
__device__ __forceinline__ float cat_load(const T* __restrict__ x1, int C1, const T* __restrict__ x2, int C2,
                                          size_t pixrow, int c) {
  return (c < C1) ? to_f32(x1[pixrow * C1 + c]) : to_f32(x2[pixrow * C2 + (c - C1)]);
}

__device__ __forceinline__ float block_sum_256(float v, float* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();  // protect `red` from a previous use
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float s = 0.f;
  const int nw = blockDim.x >> 6;
  for (int i = 0; i < nw; ++i) s += red[i];
  return s;
}

// ---- statistics -------------------------------------------------------------------------------
// Partial sums instead of finished statistics: workgroup (b, s) reads the pixels of split s of
// sample b with FULL-ROW coalesced vector loads (4 channels per lane), reduces per channel across
// its pixel lanes in a fixed order through LDS, and writes (sum, sum of squares) per group to
// part[b][s][g][2].  Consumers (k_gn_apply, the fused alias-free activation) add the S partials
// in order and finish mean / rstd themselves in fp64 — no finalize launch, bit-reproducible
// (no atomics).
template <typename T>
__global__ void __launch_bounds__(256) k_gn_partial(const T* __restrict__ x1, int C1, const T* __restrict__ x2, int C2,
                                                    float* __restrict__ part, int HW, int G, int S) {
  extern __shared__ float lds[];  // [ppl][C][2]
  const int C = C1 + C2, cpg = C / G;
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
      const bool second = c >= C1;
      const T* src = (second ? x2 : x1) + (size_t)b * HW * (second ? C2 : C1) + (second ? c - C1 : c);
      const size_t Cs = second ? C2 : C1;
      int pix = p0 + pl;
      // 4 independent loads in flight per lane
      for (; pix + 3 * ppl < p1; pix += 4 * ppl) {
        float v[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) load4<T>(src + (size_t)(pix + u * ppl) * Cs, v[u][0], v[u][1], v[u][2], v[u][3]);
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
        load4<T>(src + (size_t)pix * Cs, v[0], v[1], v[2], v[3]);
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
  for (int g = tid; g < G; g += 256) {
    float a1 = 0.f, a2 = 0.f;
    for (int l = 0; l < ppl; ++l)
      for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
        a1 += lds[((size_t)l * C + c) * 2 + 0];
        a2 += lds[((size_t)l * C + c) * 2 + 1];
      }
    float* q = part + (((size_t)b * S + sp) * G + g) * 2;
    q[0] = a1;
    q[1] = a2;
  }
}

template <typename T>
__global__ void __launch_bounds__(256) k_gn_apply(const T* __restrict__ x1, int C1, const T* __restrict__ x2, int C2,
                                                  const float* __restrict__ part, const float* __restrict__ gamma,
                                                  const float* __restrict__ beta, T* __restrict__ y, int HW, int G,
                                                  int S, float eps, int act, int rows_per_block) {
  // one workgroup = `rows_per_block` pixels of ONE sample: per-channel scale/shift once in LDS,
  // then 4 channels per lane, coalesced
  extern __shared__ float lds[];  // [C] scale, [C] shift
  const int C = C1 + C2, cpg = C / G;
  const int blocks_per_sample = (HW + rows_per_block - 1) / rows_per_block;
  const int b = blockIdx.x / blocks_per_sample;
  const int r0 = (blockIdx.x % blocks_per_sample) * rows_per_block;
  float* sc = lds;
  float* sh = lds + C;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float mean, rstd;
    gn_finalize(part, S, G, b, c / cpg, (double)HW * cpg, eps, mean, rstd);
    const float k = rstd * gamma[c];
    sc[c] = k;
    sh[c] = beta[c] - mean * k;
  }
  __syncthreads();
  const int nq = C / 4;
  const int r1 = r0 + rows_per_block < HW ? r0 + rows_per_block : HW;
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

extern "C" int afldm_gn_stats_splits(int HW) { return gn_splits(HW); }

extern "C" int afldm_gn_stats(const void* x1, int C1, const void* x2, int C2, float* part, int B, int HW, int G,
                              int dtype, afldm_stream_t stream) {
  int rc = gn_check("afldm_gn_stats", x1, C1, x2, C2, B, HW, G);
  if (rc) return rc;
  AFLDM_REQUIRE(part != nullptr, AFLDM_ENULL, "afldm_gn_stats: part is NULL");
  AFLDM_REQUIRE(C1 % 4 == 0 && C2 % 4 == 0, AFLDM_ESHAPE, "afldm_gn_stats: C1=%d / C2=%d must be multiples of 4", C1, C2);
  hipStream_t st = (hipStream_t)stream;
  const int S = gn_splits(HW), C = C1 + C2, nq = C / 4;
  const int ppl = 256 / (nq < 256 ? nq : 256);
  const size_t lds = (size_t)ppl * C * 2 * sizeof(float);
  DISPATCH_T(dtype,
             (k_gn_partial<float><<<B * S, 256, lds, st>>>((const float*)x1, C1, (const float*)x2, C2, part, HW, G, S)),
             (k_gn_partial<bf16><<<B * S, 256, lds, st>>>((const bf16*)x1, C1, (const bf16*)x2, C2, part, HW, G, S)),
             "afldm_gn_stats");
  return check_launch("afldm_gn_stats");
}

extern "C" int afldm_gn_apply(const void* x1, int C1, const void* x2, int C2, const float* part, const float* gamma,
                              const float* beta, void* y, int B, int HW, int G, float eps, int act, int dtype,
                              afldm_stream_t stream) {
  int rc = gn_check("afldm_gn_apply", x1, C1, x2, C2, B, HW, G);
  if (rc) return rc;
  AFLDM_REQUIRE(part && gamma && beta && y, AFLDM_ENULL, "afldm_gn_apply: NULL pointer");
  AFLDM_REQUIRE(act == 0 || act == 1, AFLDM_ESHAPE, "afldm_gn_apply: act %d not in {0,1}", act);
  AFLDM_REQUIRE(C1 % 4 == 0 && C2 % 4 == 0, AFLDM_ESHAPE, "afldm_gn_apply: C1=%d / C2=%d must be multiples of 4", C1, C2);
  hipStream_t st = (hipStream_t)stream;
  const int C = C1 + C2, S = gn_splits(HW);
  int rows = 8192 / C;            // ~8K elements per workgroup
  if (rows < 1) rows = 1;
  if (rows > HW) rows = HW;
  const int grid = B * ((HW + rows - 1) / rows);
  const size_t lds = (size_t)2 * C * sizeof(float);
  DISPATCH_T(dtype,
             (k_gn_apply<float><<<grid, 256, lds, st>>>((const float*)x1, C1, (const float*)x2, C2, part, gamma, beta,
                                                        (float*)y, HW, G, S, eps, act, rows)),
             (k_gn_apply<bf16><<<grid, 256, lds, st>>>((const bf16*)x1, C1, (const bf16*)x2, C2, part, gamma, beta,
                                                       (bf16*)y, HW, G, S, eps, act, rows)),
             "afldm_gn_apply");
  return check_launch("afldm_gn_apply");
}
