// exp_rate.hip - what does an exponential cost on gfx950?  (k_attn at 32x32 issues 537 M of them per launch.)
// Loops of (a) v_exp_f32, (b) plain v_fma_f32, (c) exp and fma interleaved 1:1, (d) a packed-fp32 polynomial 2^x
// (floor / fract, cubic in v_pk_fma_f32, exponent add) on 8 independent values per lane; 1024 workgroups x 256 threads.
#include <hip/hip_runtime.h>
typedef __attribute__((ext_vector_type(2))) float f2;

template <int MODE>
__global__ __launch_bounds__(256) void k_rate(float* out, int iters, float seed) {
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = seed * (float)(threadIdx.x + i + 1) * 1e-3f - 3.0f;
  float w[8];
  for (int i = 0; i < 8; ++i) w[i] = 1.0f + 1e-3f * i;
  for (int it = 0; it < iters; ++it) {
    if constexpr (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = __builtin_amdgcn_exp2f(v[i]) - 1.5f;      // 1 exp + 1 add per element (the add keeps the chain bounded)
    } else if constexpr (MODE == 1) {
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = fmaf(v[i], 0.999f, -1e-4f);
#pragma unroll
      for (int i = 0; i < 8; ++i) w[i] = fmaf(w[i], 0.999f, 1e-4f);
    } else if constexpr (MODE == 2) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        v[i] = __builtin_amdgcn_exp2f(v[i]) - 1.5f;
        w[i] = fmaf(w[i], 0.999f, 1e-4f);
        w[i] = fmaf(w[i], 1.001f, -1e-4f);
      }
    } else {
      // 2^x, x <= 0: i = floor(x), f = x - i in [0, 1), p(f) cubic, result = p * 2^i through the exponent field
#pragma unroll
      for (int i = 0; i < 8; i += 2) {
        f2 x = {v[i], v[i + 1]};
        f2 fl = {__builtin_floorf(x[0]), __builtin_floorf(x[1])};
        f2 f = x - fl;
        f2 p = f2{0.0790f, 0.0790f} * f + f2{0.2250f, 0.2250f};
        p = p * f + f2{0.6958f, 0.6958f};
        p = p * f + f2{1.0f, 1.0f};
        const int e0 = (int)fl[0], e1 = (int)fl[1];
        const float r0 = __builtin_bit_cast(float, __builtin_bit_cast(int, p[0]) + (e0 << 23));
        const float r1 = __builtin_bit_cast(float, __builtin_bit_cast(int, p[1]) + (e1 << 23));
        v[i] = r0 - 1.5f;
        v[i + 1] = r1 - 1.5f;
      }
    }
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += v[i] + w[i];
  if (s == 123.456f) out[blockIdx.x * 256 + threadIdx.x] = s;
}

extern "C" int er_run(float* out, int mode, int iters, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  switch (mode) {
    case 0: k_rate<0><<<1024, 256, 0, st>>>(out, iters, 1.0f); break;
    case 1: k_rate<1><<<1024, 256, 0, st>>>(out, iters, 1.0f); break;
    case 2: k_rate<2><<<1024, 256, 0, st>>>(out, iters, 1.0f); break;
    default: k_rate<3><<<1024, 256, 0, st>>>(out, iters, 1.0f); break;
  }
  return (int)hipGetLastError();
}
