// common.hpp — shared device/host helpers for libafldm_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/afldm_hip.h"

namespace afldm {

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

// ----------------------------------------------------------------------------- errors
void set_error(const char* fmt, ...);
int check_launch(const char* what);

#define AFLDM_REQUIRE(cond, code, ...) \
  do {                                 \
    if (!(cond)) {                     \
      afldm::set_error(__VA_ARGS__);   \
      return (code);                   \
    }                                  \
  } while (0)

#define DISPATCH_T(dtype, CALL_F32, CALL_BF16, name)            \
  do {                                                          \
    if ((dtype) == AFLDM_F32) {                                 \
      CALL_F32;                                                 \
    } else if ((dtype) == AFLDM_BF16) {                         \
      CALL_BF16;                                                \
    } else {                                                    \
      afldm::set_error("%s: unknown dtype %d", name, (int)(dtype));    \
      return AFLDM_EDTYPE;                                      \
    }                                                           \
  } while (0)

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// ----------------------------------------------------------------------------- scalar conversions
__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16 v) { return (float)v; }
template <typename T>
__device__ __forceinline__ T from_f32(float v);
template <>
__device__ __forceinline__ float from_f32<float>(float v) {
  return v;
}
template <>
__device__ __forceinline__ bf16 from_f32<bf16>(float v) {
  return (bf16)v;  // round-to-nearest-even (v_cvt_pk_bf16_f32 on gfx950)
}

// x * sigmoid(x) with the hardware exp2 / reciprocal (1 ulp class): an IEEE divide costs ~10
// instructions and SiLU is evaluated 4x per element inside the alias-free activation.
__device__ __forceinline__ float silu_f(float x) {
  const float e = __builtin_amdgcn_exp2f(x * -1.4426950408889634f);
  return x * __builtin_amdgcn_rcpf(1.0f + e);
}

// ----------------------------------------------------------------------------- 16-byte fragments
// A "chunk" is 16 bytes of a K-contiguous row: 4 fp32 or 8 bf16.  Both MFMA flavours below take
// one chunk per lane for A and one for B:
//   lane l = (i = l & 15, g = l >> 4)   A-chunk = A[i][k-set(g)],  B-chunk = B[k-set(g)][j = l & 15]
//   D (f32x4) on lane (j = l & 15, g):  D[i = 4 g + r][j],  r = 0..3      (same for both dtypes)
// f32 : v_mfma_f32_16x16x4_f32 x4, k-set(g) = {4g..4g+3}      -> one chunk pair covers K = 16
// bf16: v_mfma_f32_16x16x32_bf16,  k-set(g) = {8g..8g+7}      -> one chunk pair covers K = 32
// (the hardware pairs element e of A's lane-group g with element e of B's lane-group g, so any
//  K permutation applied identically to both operands is legal — used by the "chain" trick.)
template <typename T>
struct Mma;

template <>
struct Mma<float> {
  typedef f32x4 Chunk;
  static constexpr int EPC = 4;   // elements per chunk
  static constexpr int KPF = 16;  // K covered by one chunk pair
  static __device__ __forceinline__ void mma(f32x4& acc, const Chunk& a, const Chunk& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], acc, 0, 0, 0);
  }
  static __device__ __forceinline__ Chunk zero() { return Chunk{0.f, 0.f, 0.f, 0.f}; }
};

template <>
struct Mma<bf16> {
  typedef bf16x8 Chunk;
  static constexpr int EPC = 8;
  static constexpr int KPF = 32;
  static __device__ __forceinline__ void mma(f32x4& acc, const Chunk& a, const Chunk& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
  }
  static __device__ __forceinline__ Chunk zero() {
    Chunk z;
#pragma unroll
    for (int e = 0; e < 8; ++e) z[e] = (bf16)0.0f;
    return z;
  }
};

// 16-byte vector load/store through a generic pointer (global or LDS); p must be 16-B aligned.
template <typename V>
__device__ __forceinline__ V ld16(const void* p) {
  return *reinterpret_cast<const V*>(p);
}
template <typename V>
__device__ __forceinline__ void st16(void* p, const V& v) {
  *reinterpret_cast<V*>(p) = v;
}

// 16-byte store of a kernel's OUTPUT tensor.  With AFLDM_WT (a per-source build switch) it is a write-through (sc1)
// store: the line goes to memory as the kernel runs instead of staying dirty in the XCD's L2 until the end-of-kernel
// release writes it back in front of the next launch (cdna_hip_programming.md Guideline 16 / microarch "publish-large").
#ifndef AFLDM_WT
#define AFLDM_WT 0
#endif
template <typename V>
__device__ __forceinline__ void st16_out(void* p, const V& v) {
#if AFLDM_WT
  typedef __attribute__((ext_vector_type(4))) unsigned int u4;
  const u4 d = __builtin_bit_cast(u4, v);
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(d) : "memory");
#else
  *reinterpret_cast<V*>(p) = v;
#endif
}

// 16-byte write-through (sc1) store whatever the translation unit's AFLDM_WT: for outputs that leave as whole contiguous lines
template <typename V>
__device__ __forceinline__ void st16_wt(void* p, const V& v) {
  typedef __attribute__((ext_vector_type(4))) unsigned int u4;
  const u4 d = __builtin_bit_cast(u4, v);
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(d) : "memory");
}

// Pack 4 fp32 values as 4 consecutive T elements and store (8 B for bf16, 16 B for fp32).
template <typename T>
__device__ __forceinline__ void store4(T* p, float a, float b, float c, float d);
template <>
__device__ __forceinline__ void store4<float>(float* p, float a, float b, float c, float d) {
  *reinterpret_cast<f32x4*>(p) = f32x4{a, b, c, d};
}
template <>
__device__ __forceinline__ void store4<bf16>(bf16* p, float a, float b, float c, float d) {
  bf16x4 v;
  v[0] = (bf16)a; v[1] = (bf16)b; v[2] = (bf16)c; v[3] = (bf16)d;
  *reinterpret_cast<bf16x4*>(p) = v;
}
// store4 for a kernel's output tensor (see st16_out)
template <typename T>
__device__ __forceinline__ void store4_out(T* p, float a, float b, float c, float d) {
#if AFLDM_WT
  if constexpr (sizeof(T) == 2) {
    bf16x4 v;
    v[0] = (bf16)a; v[1] = (bf16)b; v[2] = (bf16)c; v[3] = (bf16)d;
    typedef __attribute__((ext_vector_type(2))) unsigned int u2;
    const u2 dd = __builtin_bit_cast(u2, v);
    asm volatile("global_store_dwordx2 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(dd) : "memory");
  } else {
    st16_out<f32x4>(p, f32x4{a, b, c, d});
  }
#else
  store4<T>(p, a, b, c, d);
#endif
}

template <typename T>
__device__ __forceinline__ void load4(const T* p, float& a, float& b, float& c, float& d);
template <>
__device__ __forceinline__ void load4<float>(const float* p, float& a, float& b, float& c, float& d) {
  f32x4 v = *reinterpret_cast<const f32x4*>(p);
  a = v[0]; b = v[1]; c = v[2]; d = v[3];
}
template <>
__device__ __forceinline__ void load4<bf16>(const bf16* p, float& a, float& b, float& c, float& d) {
  bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
  a = (float)v[0]; b = (float)v[1]; c = (float)v[2]; d = (float)v[3];
}

// XCD-aware work-item remap (guide T1, bijective form): consecutive work items land on the
// same XCD (private L2) instead of round-robin across the 8 XCDs.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int NX = 8;
  int xcd = bid % NX, q = nwg / NX, r = nwg % NX;
  int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + bid / NX;
}

// GroupNorm partial-sum format shared by gn.hip (producer / apply) and af.hip (fused consumer):
// part[b][s][g] = (sum, sum of squares) over split s of sample b; S = gn_splits(HW).
__host__ __device__ inline int gn_splits(int HW) {
  return HW >= 1024 ? 32 : HW >= 256 ? 16 : HW >= 64 ? 4 : 1;
}

// GroupNorm statistics travel as PER-CHANNEL partial sums: a producer (afldm_gn_stats, or the
// epilogue of afldm_conv2d / its split-K reduction) writes st[b][s][c] = (sum, sum of squares) of
// channel c over row-split s of sample b of ITS tensor; S is the producer's split count.  A
// consumer normalising the virtual channel-concat (x1 | x2) adds, for group g, the partials of the
// channels [g cpg, (g+1) cpg) over all splits in a fixed order (no atomics: bit-reproducible) and
// finishes mean / rstd itself - no finalize launch.  Per-channel (not per-group) partials let one
// tensor's statistics serve both as the next block's input and, later, as half of a skip
// concatenation whose groups straddle the two tensors.
struct GnStats {
  const float* st1;
  const float* st2;
  int C1, C2, S1, S2;
};

__device__ __forceinline__ void gn_channel_sums(const GnStats& s, int b, int c, double& s1, double& s2) {
  const bool second = c >= s.C1;
  const float* st = second ? s.st2 : s.st1;
  const int Cs = second ? s.C2 : s.C1, S = second ? s.S2 : s.S1, cc = second ? c - s.C1 : c;
  const float* q = st + ((size_t)b * S * Cs + cc) * 2;
  // eight independent loads in flight, then the adds in split order (a rolled loop waited out one load latency per
  // split: the prologue of every GroupNorm consumer at small batch)
  int i = 0;
  for (; i + 8 <= S; i += 8) {
    f32x2 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x2*>(q + (size_t)(i + u) * Cs * 2);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      s1 += (double)v[u][0];
      s2 += (double)v[u][1];
    }
  }
  for (; i < S; ++i) {
    const f32x2 v = *reinterpret_cast<const f32x2*>(q + (size_t)i * Cs * 2);
    s1 += (double)v[0];
    s2 += (double)v[1];
  }
}

// The sums are added in fp64 (the E[x^2] - mean^2 cancellation is harmless there); the rest is fp32.
__device__ __forceinline__ void gn_mean_rstd(double s1, double s2, double n, float eps, float& mean, float& rstd) {
  const double inv_n = 1.0 / n;
  const double m = s1 * inv_n;
  const float var = fmaxf((float)(s2 * inv_n - m * m), 0.f);
  mean = (float)m;
  rstd = rsqrtf(var + eps);
}

// Wave-cooperative group sums (ALL 64 lanes of the calling wave must be active): the cpg x S
// per-channel partials of group g are strided over the lanes (channel-major, fixed order) and
// combined with xor-shuffles; every lane returns the totals.
__device__ __forceinline__ void gn_group_sums_wave(const GnStats& s, int b, int g, int cpg, int lane, double& s1,
                                                   double& s2) {
  s1 = 0.0;
  s2 = 0.0;
  const int smax = s.S1 > s.S2 ? s.S1 : s.S2;
  for (int j = lane; j < cpg * smax; j += 64) {
    const int c = g * cpg + j / smax, sp = j - (j / smax) * smax;
    const bool second = c >= s.C1;
    const int S = second ? s.S2 : s.S1;
    if (sp < S) {
      const float* st = second ? s.st2 : s.st1;
      const int Cs = second ? s.C2 : s.C1, cc = second ? c - s.C1 : c;
      const f32x2 v = *reinterpret_cast<const f32x2*>(st + (((size_t)b * S + sp) * Cs + cc) * 2);
      s1 += (double)v[0];
      s2 += (double)v[1];
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s1 += __shfl_xor(s1, o, 64);
    s2 += __shfl_xor(s2, o, 64);
  }
}

// Per-lane GroupNorm (mean, rstd) for a wave whose lanes each own one (sample b, group g) key (`live` = the lane has
// one; ALL 64 lanes must call).  Up to four distinct keys are resolved per round: the keys are numbered with ballots,
// quarter q of the wave (16 lanes) walks the cpg x S partials of key q (channel-major, fixed order) and folds them
// with xor-shuffles inside the quarter - one load round trip per round instead of one per key (a wave of the N = 2
// activation spans 3-4 groups).
__device__ __forceinline__ void gn_wave_keys(const GnStats& s, bool live, int b, int g, int cpg, double n, float eps,
                                             int lane, float& mean, float& rstd) {
  bool done = !live;
  mean = 0.f;
  rstd = 1.f;
  const int smax = s.S1 > s.S2 ? s.S1 : s.S2;
  while (__any(!done)) {
    int slot = -1, nk = 0, kb[4], kg[4];                    // this lane's key number in the round, keys of the round
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const unsigned long long open = __ballot(!done && slot < 0);
      kb[k] = 0;
      kg[k] = 0;
      if (open) {                                           // (wave-uniform)
        const int leader = __ffsll(open) - 1;
        kb[k] = __shfl(b, leader, 64);
        kg[k] = __shfl(g, leader, 64);
        if (!done && slot < 0 && b == kb[k] && g == kg[k]) slot = k;
        nk = k + 1;
      }
    }
    const int q = lane >> 4, ql = lane & 15;
    const int qb = q == 0 ? kb[0] : q == 1 ? kb[1] : q == 2 ? kb[2] : kb[3];
    const int qg = q == 0 ? kg[0] : q == 1 ? kg[1] : q == 2 ? kg[2] : kg[3];
    double s1 = 0.0, s2 = 0.0;
    if (q < nk) {
      for (int j = ql; j < cpg * smax; j += 16) {
        const int c = qg * cpg + j / smax, sp = j - (j / smax) * smax;
        const bool second = c >= s.C1;
        const int S = second ? s.S2 : s.S1;
        if (sp < S) {
          const float* st = second ? s.st2 : s.st1;
          const int Cs = second ? s.C2 : s.C1, cc = second ? c - s.C1 : c;
          const f32x2 v = *reinterpret_cast<const f32x2*>(st + (((size_t)qb * S + sp) * Cs + cc) * 2);
          s1 += (double)v[0];
          s2 += (double)v[1];
        }
      }
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {                       // (xor < 16: stays inside the quarter)
      s1 += __shfl_xor(s1, o, 64);
      s2 += __shfl_xor(s2, o, 64);
    }
    const int src = (slot < 0 ? 0 : slot) * 16;
    const double t1 = __shfl(s1, src, 64), t2 = __shfl(s2, src, 64);
    if (slot >= 0) {
      gn_mean_rstd(t1, t2, n, eps, mean, rstd);
      done = true;
    }
  }
}

// (mean, rstd) of group g of sample b, serially by one thread.
__device__ __forceinline__ void gn_finalize(const GnStats& s, int b, int g, int cpg, double n, float eps, float& mean,
                                            float& rstd) {
  double s1 = 0.0, s2 = 0.0;
  for (int c = g * cpg; c < (g + 1) * cpg; ++c) gn_channel_sums(s, b, c, s1, s2);
  gn_mean_rstd(s1, s2, n, eps, mean, rstd);
}

// "Once per DEVICE" guard for hipFuncSetAttribute(MaxDynamicSharedMemorySize, ...): the attribute belongs to the
// (function, device) pair, so a process that drives a second GPU has to set it there too.  `done` is a
// bit mask over device ordinals; the caller keeps it in a function-local static.
static inline bool first_on_device(unsigned long long& done) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned long long bit = 1ull << (dev & 63);
  if (done & bit) return false;
  done |= bit;
  return true;
}

static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

}  // namespace afldm
