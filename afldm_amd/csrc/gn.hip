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

// one workgroup per (b, g): two-pass (mean, then centred variance) like torch's CPU kernel
template <typename T>
__global__ void __launch_bounds__(256) k_gn_stats(const T* __restrict__ x1, int C1, const T* __restrict__ x2, int C2,
                                                  float* __restrict__ stats, int HW, int G, float eps) {
  __shared__ float red[8];
  const int b = blockIdx.x / G, g = blockIdx.x % G;
  const int C = C1 + C2, cpg = C / G;
  const int n = HW * cpg;
  const int c0 = g * cpg;
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    int pix = i / cpg, cc = i - pix * cpg;
    s += cat_load(x1, C1, x2, C2, (size_t)b * HW + pix, c0 + cc);
  }
  const float mean = block_sum_256(s, red) / (float)n;
  float v = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    int pix = i / cpg, cc = i - pix * cpg;
    float d = cat_load(x1, C1, x2, C2, (size_t)b * HW + pix, c0 + cc) - mean;
    v += d * d;
  }
  const float var = block_sum_256(v, red) / (float)n;
  if (threadIdx.x == 0) {
    stats[2 * blockIdx.x + 0] = mean;
    stats[2 * blockIdx.x + 1] = rsqrtf(var + eps);
  }
}

template <typename T>
__global__ void __launch_bounds__(256) k_gn_apply(const T* __restrict__ x1, int C1, const T* __restrict__ x2, int C2,
                                                  const float* __restrict__ stats, const float* __restrict__ gamma,
                                                  const float* __restrict__ beta, T* __restrict__ y, int B, int HW,
                                                  int G, int act) {
  const int C = C1 + C2, cpg = C / G;
  const size_t n = (size_t)B * HW * C;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    int c = (int)(i % C);
    size_t p = i / C;
    int b = (int)(p / HW);
    int g = c / cpg;
    float mean = stats[2 * (b * G + g)], rstd = stats[2 * (b * G + g) + 1];
    float v = cat_load(x1, C1, x2, C2, p, c);
    v = (v - mean) * rstd * gamma[c] + beta[c];
    if (act == 1) v = silu_f(v);
    y[i] = from_f32<T>(v);
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

extern "C" int afldm_gn_stats(const void* x1, int C1, const void* x2, int C2, float* stats, int B, int HW, int G,
                              float eps, int dtype, afldm_stream_t stream) {
  int rc = gn_check("afldm_gn_stats", x1, C1, x2, C2, B, HW, G);
  if (rc) return rc;
  AFLDM_REQUIRE(stats != nullptr, AFLDM_ENULL, "afldm_gn_stats: stats is NULL");
  hipStream_t st = (hipStream_t)stream;
  DISPATCH_T(dtype,
             (k_gn_stats<float><<<B * G, 256, 0, st>>>((const float*)x1, C1, (const float*)x2, C2, stats, HW, G, eps)),
             (k_gn_stats<bf16><<<B * G, 256, 0, st>>>((const bf16*)x1, C1, (const bf16*)x2, C2, stats, HW, G, eps)),
             "afldm_gn_stats");
  return check_launch("afldm_gn_stats");
}

extern "C" int afldm_gn_apply(const void* x1, int C1, const void* x2, int C2, const float* stats, const float* gamma,
                              const float* beta, void* y, int B, int HW, int G, int act, int dtype,
                              afldm_stream_t stream) {
  int rc = gn_check("afldm_gn_apply", x1, C1, x2, C2, B, HW, G);
  if (rc) return rc;
  AFLDM_REQUIRE(stats && gamma && beta && y, AFLDM_ENULL, "afldm_gn_apply: NULL pointer");
  AFLDM_REQUIRE(act == 0 || act == 1, AFLDM_ESHAPE, "afldm_gn_apply: act %d not in {0,1}", act);
  hipStream_t st = (hipStream_t)stream;
  size_t n = (size_t)B * HW * (C1 + C2);
  int grid = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
  DISPATCH_T(dtype,
             (k_gn_apply<float><<<grid, 256, 0, st>>>((const float*)x1, C1, (const float*)x2, C2, stats, gamma, beta,
                                                      (float*)y, B, HW, G, act)),
             (k_gn_apply<bf16><<<grid, 256, 0, st>>>((const bf16*)x1, C1, (const bf16*)x2, C2, stats, gamma, beta,
                                                     (bf16*)y, B, HW, G, act)),
             "afldm_gn_apply");
  return check_launch("afldm_gn_apply");
}
