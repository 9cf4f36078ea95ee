// Is the MFMA rate of this chip data-dependent?  The same loop of independent v_mfma_f32_32x32x16_bf16 (4 accumulator
// chains per wave, 4 waves per SIMD, every SIMD) with (mode 0) one constant low-entropy operand pair, (mode 1) operands
// of random bf16 bit patterns (|x| ~ 1) rotating through four register sets, (mode 2) as 1 plus one ds_read_b128 per MFMA
// (the halo kernel's LDS traffic per MFMA is 20 reads per 48 MFMAs).  Reports the clock the loop ran at (s_memtime).
#include <hip/hip_runtime.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__device__ __forceinline__ unsigned hash(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ bf16x8 rnd8(unsigned seed) {
  u32x4 v;
  for (int i = 0; i < 4; ++i) {
    // two bf16 per dword: sign random, exponent 126 / 127 (0.5 .. 2), mantissa random
    const unsigned h = hash(seed * 4 + i);
    const unsigned lo = ((h & 0x8000u) | (0x3f00u + (h & 0xffu))) & 0xffffu;
    const unsigned hi = (((h >> 16) & 0x8000u) | (0x3f00u + ((h >> 16) & 0xffu))) & 0xffffu;
    v[i] = lo | (hi << 16);
  }
  return __builtin_bit_cast(bf16x8, v);
}

template <int MODE>
__global__ __launch_bounds__(256) void k_mfma(float* out, unsigned long long* ticks, int iters) {
  __shared__ __attribute__((aligned(16))) char lds[16384];
  bf16x8 a[4], b[4];
  for (int s = 0; s < 4; ++s) {
    if (MODE == 0) {
      for (int i = 0; i < 8; ++i) {
        a[s][i] = (__bf16)(0.001f * (float)((threadIdx.x + i) & 7));
        b[s][i] = (__bf16)(0.5f + 0.001f * (float)((threadIdx.x * 3 + i) & 7));
      }
    } else {
      a[s] = rnd8(threadIdx.x * 8 + s + blockIdx.x * 4096);
      b[s] = rnd8(threadIdx.x * 8 + s + 4 + blockIdx.x * 4096);
    }
  }
  if (MODE >= 2) {
    for (int i = threadIdx.x; i < 1024; i += 256) reinterpret_cast<u32x4*>(lds)[i] = __builtin_bit_cast(u32x4, rnd8(i + 77));
    __syncthreads();
  }
  f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if (MODE >= 3) {
        // MODE - 2 fragment reads per 4 MFMAs, data discarded (operands stay the random register sets): the halo kernel
        // reads 20 fragments per 48 16x16x32 MFMAs = 3.3 per four 32x32x16 MFMAs' worth of flops
#pragma unroll
        for (int r = 0; r < MODE - 2; ++r) {
          const bf16x8 v = *reinterpret_cast<const bf16x8*>(lds + ((threadIdx.x * 16 + ((it * 4 + s) * 4 + r) * 1024) & 16383));
          asm volatile("" ::"v"(v));
        }
      }
      if (MODE == 2) {
        // one fragment read per MFMA (conflict-free: lane * 16 bytes), folded into the operands so that it stays live
        const bf16x8 r = *reinterpret_cast<const bf16x8*>(lds + ((threadIdx.x * 16 + (it * 4 + s) * 1024) & 16383));
        a[s] = r;
      }
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s], b[s], c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(s + 1) & 3], b[s], c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(s + 2) & 3], b[(s + 1) & 3], c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(s + 3) & 3], b[(s + 2) & 3], c3, 0, 0, 0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
  if (s == -1.2345f) out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}

extern "C" int mp_run(float* out, unsigned long long* ticks, int mode, int wgs, int iters, hipStream_t st) {
  if (mode == 0) k_mfma<0><<<wgs, 256, 0, st>>>(out, ticks, iters);
  else if (mode == 1) k_mfma<1><<<wgs, 256, 0, st>>>(out, ticks, iters);
  else if (mode == 2) k_mfma<2><<<wgs, 256, 0, st>>>(out, ticks, iters);
  else if (mode == 3) k_mfma<3><<<wgs, 256, 0, st>>>(out, ticks, iters);
  else if (mode == 4) k_mfma<4><<<wgs, 256, 0, st>>>(out, ticks, iters);
  else if (mode == 6) k_mfma<6><<<wgs, 256, 0, st>>>(out, ticks, iters);
  else k_mfma<5><<<wgs, 256, 0, st>>>(out, ticks, iters);
  return (int)hipGetLastError();
}
