// misc.hip — layout/dtype plumbing, timestep embedding, SiLU, DDIM update.  All HBM-bound
// elementwise kernels: coalesced accesses, grid-stride loops, no LDS.
#include "common.hpp"

namespace afldm {

static inline int ew_grid(size_t n, int block = 256) {
  size_t g = (n + block - 1) / block;
  size_t cap = 256 * 8;  // 256 CUs x 8 blocks, grid-stride beyond (guide G11)
  return (int)(g < cap ? (g ? g : 1) : cap);
}

// ----------------------------------------------------------------------------- NCHW <-> NHWC
// One workgroup transposes a [C][64-pixel] slab through LDS so both sides stay coalesced
// when C is large; for C <= 8 (the 4-channel latents) the direct form below is already fine.
template <typename T>
__global__ void k_nchw_to_nhwc(const float* __restrict__ src, T* __restrict__ dst, int B, int C, int HW) {
  size_t n = (size_t)B * C * HW;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    int c = (int)(i % C);
    size_t p = i / C;  // b*HW + pix
    int pix = (int)(p % HW);
    int b = (int)(p / HW);
    dst[i] = from_f32<T>(src[((size_t)b * C + c) * HW + pix]);
  }
}
template <typename T>
__global__ void k_nhwc_to_nchw(const T* __restrict__ src, float* __restrict__ dst, int B, int C, int HW) {
  size_t n = (size_t)B * C * HW;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    int pix = (int)(i % HW);
    size_t p = i / HW;  // b*C + c
    int c = (int)(p % C);
    int b = (int)(p / C);
    dst[i] = to_f32(src[((size_t)b * HW + pix) * C + c]);
  }
}

// OIHW fp32 -> OHWI T
template <typename T>
__global__ void k_pack_weight(const float* __restrict__ src, T* __restrict__ dst, int O, int I, int KK) {
  size_t n = (size_t)O * I * KK;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    int ci = (int)(i % I);
    size_t r = i / I;
    int kk = (int)(r % KK);
    int o = (int)(r / KK);
    dst[i] = from_f32<T>(src[((size_t)o * I + ci) * KK + kk]);
  }
}

template <typename S, typename D>
__global__ void k_cast(const S* __restrict__ src, D* __restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    dst[i] = from_f32<D>(to_f32(src[i]));
}

// ----------------------------------------------------------------------------- timestep embedding
template <typename T>
__global__ void k_timestep_embedding(const float* __restrict__ t, T* __restrict__ out, int rows, int dim,
                                     int flip, float freq_shift) {
  int half = dim / 2;
  int n = rows * half;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    int r = i / half, k = i % half;
    // exponent = -ln(10000) * k / (half - freq_shift); emb = exp(exponent)  (fp32, like torch)
    float e = -9.210340371976184f * (float)k;
    e = e / ((float)half - freq_shift);
    float w = expf(e);
    float a = t[r] * w;
    float s = sinf(a), c = cosf(a);
    T* o = out + (size_t)r * dim;
    if (flip) {
      o[k] = from_f32<T>(c);
      o[half + k] = from_f32<T>(s);
    } else {
      o[k] = from_f32<T>(s);
      o[half + k] = from_f32<T>(c);
    }
  }
}

template <typename T>
__global__ void k_silu(const T* __restrict__ x, T* __restrict__ y, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    y[i] = from_f32<T>(silu_f(to_f32(x[i])));
}

// ----------------------------------------------------------------------------- DDIM
template <typename T>
// (x and xprev may alias: the engine updates the latents in place, every thread reads and writes index i only)
__global__ void k_ddim_step(const float* x, const T* __restrict__ eps, float* xprev,
                            const float* __restrict__ coef, int* __restrict__ step_idx, int advance, int B,
                            int C, int HW) {
  const int s = *step_idx;
  const float sa_t = coef[4 * s + 0], sb_t = coef[4 * s + 1], sa_p = coef[4 * s + 2], sb_p = coef[4 * s + 3];
  size_t n = (size_t)B * C * HW;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    int pix = (int)(i % HW);
    size_t p = i / HW;
    int c = (int)(p % C);
    int b = (int)(p / C);
    float e = to_f32(eps[((size_t)b * HW + pix) * C + c]);
    float xv = x[i];
    float x0 = (xv - sb_t * e) / sa_t;
    xprev[i] = sa_p * x0 + sb_p * e;
  }
}
// separate tiny kernel so that every block of k_ddim_step reads the same (old) index
__global__ void k_advance(int* step_idx) { *step_idx += 1; }

// pre_advance: the step counter is incremented HERE, at the head of the step (one thread, after every kernel of
// the previous step), instead of by a k_advance launch behind the previous step's DDIM update
__global__ void k_select_timestep(const float* tvals, int* step_idx, float* t_out, int pre_advance) {
  int s = *step_idx;
  if (pre_advance) {
    s += 1;
    *step_idx = s;
  }
  t_out[0] = tvals[s];
}

// Step prologue of the sampling loop in ONE launch (single workgroup): advance the step counter, publish the
// timestep value, and copy row `step` of a precomputed table (the 27 time_emb_proj outputs of that timestep:
// time_proj -> linear_1 -> SiLU -> linear_2 -> SiLU -> all time_emb_proj layers depend on the timestep only, and a
// sampler knows its timesteps in advance) into the buffer the convolutions read - instead of seven launches per step.
__global__ void __launch_bounds__(1024) k_select_step_row(const float* tvals, int* step_idx, float* t_out, int pre_advance,
                                                          const uint4* table, uint4* row_out, int row_chunks) {
  __shared__ int s_step;
  if (threadIdx.x == 0) {
    int s = *step_idx;
    if (pre_advance) {
      s += 1;
      *step_idx = s;
    }
    t_out[0] = tvals[s];
    s_step = s;
  }
  __syncthreads();
  const uint4* src = table + (size_t)s_step * row_chunks;
  for (int i = threadIdx.x; i < row_chunks; i += 1024) row_out[i] = src[i];
}

// Masked equivariance metrics in ONE pass (reference shift_utils/metrics.py:5-20): per sample
// out[b] = { sum ((a - b) m)^2, sum m, max(a m), min(a m), max(b m), min(b m) }.  One workgroup per sample,
// fixed summation order (strided per-thread partials, xor-shuffle tree, 4 wave partials in order).
template <typename T>
__global__ void __launch_bounds__(256) k_masked_metrics(const T* __restrict__ a, const T* __restrict__ b,
                                                        const float* __restrict__ m, float* __restrict__ out, size_t n) {
  __shared__ float red[4][6];
  const size_t base = (size_t)blockIdx.x * n;
  float v[6] = {0.f, 0.f, -3.4e38f, 3.4e38f, -3.4e38f, 3.4e38f};
  for (size_t i = threadIdx.x; i < n; i += 256) {
    const float mk = m[base + i];
    const float am = to_f32(a[base + i]) * mk, bm = to_f32(b[base + i]) * mk;
    const float d = am - bm;
    v[0] = fmaf(d, d, v[0]);
    v[1] += mk;
    v[2] = fmaxf(v[2], am);
    v[3] = fminf(v[3], am);
    v[4] = fmaxf(v[4], bm);
    v[5] = fminf(v[5], bm);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    v[0] += __shfl_xor(v[0], o, 64);
    v[1] += __shfl_xor(v[1], o, 64);
    v[2] = fmaxf(v[2], __shfl_xor(v[2], o, 64));
    v[3] = fminf(v[3], __shfl_xor(v[3], o, 64));
    v[4] = fmaxf(v[4], __shfl_xor(v[4], o, 64));
    v[5] = fminf(v[5], __shfl_xor(v[5], o, 64));
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0)
#pragma unroll
    for (int k = 0; k < 6; ++k) red[wave][k] = v[k];
  __syncthreads();
  if (threadIdx.x == 0) {
    float* o = out + (size_t)blockIdx.x * 6;
    o[0] = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);
    o[1] = (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]);
    o[2] = fmaxf(fmaxf(red[0][2], red[1][2]), fmaxf(red[2][2], red[3][2]));
    o[3] = fminf(fminf(red[0][3], red[1][3]), fminf(red[2][3], red[3][3]));
    o[4] = fmaxf(fmaxf(red[0][4], red[1][4]), fmaxf(red[2][4], red[3][4]));
    o[5] = fminf(fminf(red[0][5], red[1][5]), fminf(red[2][5], red[3][5]));
  }
}

}  // namespace afldm

using namespace afldm;

extern "C" int afldm_masked_metrics(const void* a, const void* b, const float* mask, float* out, int B, size_t n,
                                    int dtype, afldm_stream_t stream) {
  AFLDM_REQUIRE(a && b && mask && out, AFLDM_ENULL, "afldm_masked_metrics: NULL pointer");
  AFLDM_REQUIRE(B > 0 && n > 0, AFLDM_ESHAPE, "afldm_masked_metrics: bad shape");
  hipStream_t st = (hipStream_t)stream;
  DISPATCH_T(dtype, (k_masked_metrics<float><<<B, 256, 0, st>>>((const float*)a, (const float*)b, mask, out, n)),
             (k_masked_metrics<bf16><<<B, 256, 0, st>>>((const bf16*)a, (const bf16*)b, mask, out, n)), "afldm_masked_metrics");
  return check_launch("afldm_masked_metrics");
}

extern "C" int afldm_nchw_to_nhwc(const float* src, void* dst, int B, int C, int H, int W, int dtype,
                                  afldm_stream_t stream) {
  AFLDM_REQUIRE(src && dst, AFLDM_ENULL, "afldm_nchw_to_nhwc: NULL pointer");
  AFLDM_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, AFLDM_ESHAPE, "afldm_nchw_to_nhwc: bad shape");
  hipStream_t st = (hipStream_t)stream;
  size_t n = (size_t)B * C * H * W;
  DISPATCH_T(dtype, (k_nchw_to_nhwc<float><<<ew_grid(n), 256, 0, st>>>(src, (float*)dst, B, C, H * W)),
             (k_nchw_to_nhwc<bf16><<<ew_grid(n), 256, 0, st>>>(src, (bf16*)dst, B, C, H * W)), "afldm_nchw_to_nhwc");
  return check_launch("afldm_nchw_to_nhwc");
}

extern "C" int afldm_nhwc_to_nchw(const void* src, float* dst, int B, int C, int H, int W, int dtype,
                                  afldm_stream_t stream) {
  AFLDM_REQUIRE(src && dst, AFLDM_ENULL, "afldm_nhwc_to_nchw: NULL pointer");
  AFLDM_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, AFLDM_ESHAPE, "afldm_nhwc_to_nchw: bad shape");
  hipStream_t st = (hipStream_t)stream;
  size_t n = (size_t)B * C * H * W;
  DISPATCH_T(dtype, (k_nhwc_to_nchw<float><<<ew_grid(n), 256, 0, st>>>((const float*)src, dst, B, C, H * W)),
             (k_nhwc_to_nchw<bf16><<<ew_grid(n), 256, 0, st>>>((const bf16*)src, dst, B, C, H * W)), "afldm_nhwc_to_nchw");
  return check_launch("afldm_nhwc_to_nchw");
}

extern "C" int afldm_pack_weight(const float* src, void* dst, int O, int I, int KH, int KW, int dtype,
                                 afldm_stream_t stream) {
  AFLDM_REQUIRE(src && dst, AFLDM_ENULL, "afldm_pack_weight: NULL pointer");
  AFLDM_REQUIRE(O > 0 && I > 0 && KH > 0 && KW > 0, AFLDM_ESHAPE, "afldm_pack_weight: bad shape");
  hipStream_t st = (hipStream_t)stream;
  size_t n = (size_t)O * I * KH * KW;
  DISPATCH_T(dtype, (k_pack_weight<float><<<ew_grid(n), 256, 0, st>>>(src, (float*)dst, O, I, KH * KW)),
             (k_pack_weight<bf16><<<ew_grid(n), 256, 0, st>>>(src, (bf16*)dst, O, I, KH * KW)), "afldm_pack_weight");
  return check_launch("afldm_pack_weight");
}

extern "C" int afldm_cast(const void* src, int sd, void* dst, int dd, size_t n, afldm_stream_t stream) {
  AFLDM_REQUIRE(src && dst, AFLDM_ENULL, "afldm_cast: NULL pointer");
  hipStream_t st = (hipStream_t)stream;
  if (n == 0) return AFLDM_OK;
  if (sd == AFLDM_F32 && dd == AFLDM_F32)
    k_cast<float, float><<<ew_grid(n), 256, 0, st>>>((const float*)src, (float*)dst, n);
  else if (sd == AFLDM_F32 && dd == AFLDM_BF16)
    k_cast<float, bf16><<<ew_grid(n), 256, 0, st>>>((const float*)src, (bf16*)dst, n);
  else if (sd == AFLDM_BF16 && dd == AFLDM_F32)
    k_cast<bf16, float><<<ew_grid(n), 256, 0, st>>>((const bf16*)src, (float*)dst, n);
  else if (sd == AFLDM_BF16 && dd == AFLDM_BF16)
    k_cast<bf16, bf16><<<ew_grid(n), 256, 0, st>>>((const bf16*)src, (bf16*)dst, n);
  else {
    set_error("afldm_cast: unknown dtype %d/%d", sd, dd);
    return AFLDM_EDTYPE;
  }
  return check_launch("afldm_cast");
}

extern "C" int afldm_timestep_embedding(const float* t, void* out, int rows, int dim, int flip, float freq_shift,
                                        int dtype, afldm_stream_t stream) {
  AFLDM_REQUIRE(t && out, AFLDM_ENULL, "afldm_timestep_embedding: NULL pointer");
  AFLDM_REQUIRE(rows > 0 && dim > 0 && dim % 2 == 0, AFLDM_ESHAPE, "afldm_timestep_embedding: dim %d must be even", dim);
  hipStream_t st = (hipStream_t)stream;
  size_t n = (size_t)rows * dim / 2;
  DISPATCH_T(dtype, (k_timestep_embedding<float><<<ew_grid(n), 256, 0, st>>>(t, (float*)out, rows, dim, flip, freq_shift)),
             (k_timestep_embedding<bf16><<<ew_grid(n), 256, 0, st>>>(t, (bf16*)out, rows, dim, flip, freq_shift)),
             "afldm_timestep_embedding");
  return check_launch("afldm_timestep_embedding");
}

extern "C" int afldm_silu(const void* x, void* y, size_t n, int dtype, afldm_stream_t stream) {
  AFLDM_REQUIRE(x && y, AFLDM_ENULL, "afldm_silu: NULL pointer");
  if (n == 0) return AFLDM_OK;
  hipStream_t st = (hipStream_t)stream;
  DISPATCH_T(dtype, (k_silu<float><<<ew_grid(n), 256, 0, st>>>((const float*)x, (float*)y, n)),
             (k_silu<bf16><<<ew_grid(n), 256, 0, st>>>((const bf16*)x, (bf16*)y, n)), "afldm_silu");
  return check_launch("afldm_silu");
}

extern "C" int afldm_ddim_step(const float* x, const void* eps, float* x_prev, const float* coef, int* step_idx,
                               int advance, int B, int C, int H, int W, int dtype, afldm_stream_t stream) {
  AFLDM_REQUIRE(x && eps && x_prev && coef && step_idx, AFLDM_ENULL, "afldm_ddim_step: NULL pointer");
  AFLDM_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, AFLDM_ESHAPE, "afldm_ddim_step: bad shape");
  hipStream_t st = (hipStream_t)stream;
  size_t n = (size_t)B * C * H * W;
  DISPATCH_T(dtype,
             (k_ddim_step<float><<<ew_grid(n), 256, 0, st>>>(x, (const float*)eps, x_prev, coef, step_idx, advance, B, C, H * W)),
             (k_ddim_step<bf16><<<ew_grid(n), 256, 0, st>>>(x, (const bf16*)eps, x_prev, coef, step_idx, advance, B, C, H * W)),
             "afldm_ddim_step");
  if (advance) k_advance<<<1, 1, 0, st>>>(step_idx);
  return check_launch("afldm_ddim_step");
}

extern "C" int afldm_select_step_row(const float* tvals, int* step_idx, float* t_out, int pre_advance, const void* table,
                                     void* row_out, size_t row_bytes, afldm_stream_t stream) {
  AFLDM_REQUIRE(tvals && step_idx && t_out && table && row_out, AFLDM_ENULL, "afldm_select_step_row: NULL pointer");
  AFLDM_REQUIRE(row_bytes > 0 && row_bytes % 16 == 0 && aligned16(table) && aligned16(row_out), AFLDM_EALIGN,
                "afldm_select_step_row: rows must be whole 16-byte chunks");
  k_select_step_row<<<1, 1024, 0, (hipStream_t)stream>>>(tvals, step_idx, t_out, pre_advance, (const uint4*)table,
                                                          (uint4*)row_out, (int)(row_bytes / 16));
  return check_launch("afldm_select_step_row");
}

extern "C" int afldm_select_timestep(const float* tvals, int* step_idx, float* t_out, int pre_advance,
                                     afldm_stream_t stream) {
  AFLDM_REQUIRE(tvals && step_idx && t_out, AFLDM_ENULL, "afldm_select_timestep: NULL pointer");
  k_select_timestep<<<1, 1, 0, (hipStream_t)stream>>>(tvals, step_idx, t_out, pre_advance);
  return check_launch("afldm_select_timestep");
}

// Same update with the four coefficients passed by value and flat fp32 tensors (the
// DDIMScheduler.step(model_output, t, sample) API path, where both tensors are NCHW fp32).
__global__ void k_ddim_step_flat(const float* x, const float* __restrict__ eps, float* xprev, float sa_t, float sb_t, float sa_p, float sb_p,
                                 size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float e = eps[i];
    float x0 = (x[i] - sb_t * e) / sa_t;
    xprev[i] = sa_p * x0 + sb_p * e;
  }
}

extern "C" int afldm_ddim_step_flat(const float* x, const float* eps, float* x_prev, float sqrt_a_t,
                                    float sqrt_1m_a_t, float sqrt_a_prev, float sqrt_1m_a_prev, size_t n,
                                    afldm_stream_t stream) {
  AFLDM_REQUIRE(x && eps && x_prev, AFLDM_ENULL, "afldm_ddim_step_flat: NULL pointer");
  if (n == 0) return AFLDM_OK;
  size_t g = (n + 255) / 256;
  if (g > 2048) g = 2048;
  k_ddim_step_flat<<<(int)g, 256, 0, (hipStream_t)stream>>>(x, eps, x_prev, sqrt_a_t, sqrt_1m_a_t, sqrt_a_prev,
                                                            sqrt_1m_a_prev, n);
  return check_launch("afldm_ddim_step_flat");
}

// ----------------------------------------------------------------------------- box calibration probes
// bench.py's `box` record: a fixed MFMA loop and a stream copy, timed in every run, so that numbers measured on
// different MI355X boxes (clocks / power caps differ by ~10 %) can be normalised.
typedef __attribute__((ext_vector_type(8))) __bf16 probe_bf16x8;
typedef __attribute__((ext_vector_type(16))) float probe_f32x16;

__global__ __launch_bounds__(256) void k_probe_mfma(float* out, int iters) {
  probe_bf16x8 a, b;
  for (int i = 0; i < 8; ++i) {
    a[i] = (__bf16)(0.001f * (float)((threadIdx.x + i) & 7));
    b[i] = (__bf16)(0.5f + 0.001f * (float)((threadIdx.x * 3 + i) & 7));
  }
  probe_f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
  for (int it = 0; it < iters; ++it) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
  if (s == -1.2345f) out[blockIdx.x * 256 + threadIdx.x] = s;      // never true: keeps the loop alive
}

// The same loop with operands of RANDOM bf16 bit patterns (sign and mantissa random, 0.5 <= |x| < 2) rotating through four
// register sets: the matrix pipe's sustained rate depends on the data it multiplies (the chip holds 2.36 GHz on the
// near-constant operands above and ~1.55 GHz on these: 1.6 against 2.4 PFLOP/s, tools/proto/mfma_power.hip) - this figure,
// not the datasheet peak, is what a convolution on real activations can reach.
__device__ __forceinline__ unsigned probe_hash(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ probe_bf16x8 probe_rnd8(unsigned seed) {
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
  u32x4 v;
  for (int i = 0; i < 4; ++i) {
    const unsigned h = probe_hash(seed * 4 + i);
    const unsigned lo = ((h & 0x8000u) | (0x3f00u + (h & 0xffu))) & 0xffffu;
    const unsigned hi = (((h >> 16) & 0x8000u) | (0x3f00u + ((h >> 16) & 0xffu))) & 0xffffu;
    v[i] = lo | (hi << 16);
  }
  return __builtin_bit_cast(probe_bf16x8, v);
}
__global__ __launch_bounds__(256) void k_probe_mfma_random(float* out, int iters) {
  probe_bf16x8 a[4], b[4];
  for (int s = 0; s < 4; ++s) {
    a[s] = probe_rnd8(threadIdx.x * 8 + s + blockIdx.x * 4096);
    b[s] = probe_rnd8(threadIdx.x * 8 + s + 4 + blockIdx.x * 4096);
  }
  probe_f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
  for (int it = 0; it < iters; it += 4) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s], b[s], c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(s + 1) & 3], b[s], c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(s + 2) & 3], b[(s + 1) & 3], c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(s + 3) & 3], b[(s + 2) & 3], c3, 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
  if (s == -1.2345f) out[blockIdx.x * 256 + threadIdx.x] = s;      // never true: keeps the loop alive
}

__global__ __launch_bounds__(256) void k_probe_copy(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

extern "C" int afldm_probe_mfma(float* out, int workgroups, int iters, afldm_stream_t stream) {
  AFLDM_REQUIRE(out && workgroups > 0 && iters > 0, AFLDM_ESHAPE, "afldm_probe_mfma: out[workgroups * 256], workgroups, iters > 0");
  k_probe_mfma<<<workgroups, 256, 0, (hipStream_t)stream>>>(out, iters);
  return check_launch("afldm_probe_mfma");
}

extern "C" int afldm_probe_mfma_random(float* out, int workgroups, int iters, afldm_stream_t stream) {
  AFLDM_REQUIRE(out && workgroups > 0 && iters > 0 && iters % 4 == 0, AFLDM_ESHAPE,
                "afldm_probe_mfma_random: out[workgroups * 256], workgroups > 0, iters > 0 and a multiple of 4");
  k_probe_mfma_random<<<workgroups, 256, 0, (hipStream_t)stream>>>(out, iters);
  return check_launch("afldm_probe_mfma_random");
}

// dependent-load chain (one lane): next = buf[next]; `buf` holds ONE cycle over its n entries (host-built), the stride
// between consecutive elements of the cycle is what the host chose (>= a cache line: every step misses).  out[0] = the
// final index (keeps the chain alive), out[1] = elapsed shader-clock ticks (s_memtime).
__global__ void k_probe_chase(const unsigned* __restrict__ buf, unsigned* __restrict__ out, int steps) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  unsigned i = 0;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int s = 0; s < steps; ++s) i = __builtin_nontemporal_load(buf + i);
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[0] = i;
  out[1] = (unsigned)(t1 - t0);
}
__global__ void k_probe_empty(unsigned* out) {
  if (out == nullptr) return;
}

extern "C" int afldm_probe_chase(const void* buf, void* out, int steps, afldm_stream_t stream) {
  AFLDM_REQUIRE(buf && out && steps > 0, AFLDM_ENULL, "afldm_probe_chase: NULL pointer / no steps");
  k_probe_chase<<<1, 64, 0, (hipStream_t)stream>>>((const unsigned*)buf, (unsigned*)out, steps);
  return check_launch("afldm_probe_chase");
}

extern "C" int afldm_probe_empty(int workgroups, afldm_stream_t stream) {
  AFLDM_REQUIRE(workgroups > 0, AFLDM_ESHAPE, "afldm_probe_empty: no workgroups");
  k_probe_empty<<<workgroups, 64, 0, (hipStream_t)stream>>>(nullptr);
  return check_launch("afldm_probe_empty");
}

extern "C" int afldm_probe_copy(const void* src, void* dst, size_t bytes, afldm_stream_t stream) {
  AFLDM_REQUIRE(src && dst, AFLDM_ENULL, "afldm_probe_copy: NULL pointer");
  AFLDM_REQUIRE(bytes % 16 == 0 && aligned16(src) && aligned16(dst), AFLDM_EALIGN, "afldm_probe_copy: whole 16-byte chunks");
  if (bytes == 0) return AFLDM_OK;
  k_probe_copy<<<256 * 16, 256, 0, (hipStream_t)stream>>>((const uint4*)src, (uint4*)dst, bytes / 16);
  return check_launch("afldm_probe_copy");
}
