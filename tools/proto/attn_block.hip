// attn_block.hip - PROTOTYPE (not part of libafldm_hip.so): one self-attention block of the low-resolution levels
// (q|k|v projection -> attention -> to_out + bias + residual, GroupNorm partial sums of the result) as ONE persistent launch
// whose phases hand over XCD-locally.  Samples are partitioned across the 8 XCDs (B / 8 each); the 32 workgroups of an
// XCD cooperate on their samples and meet at per-XCD counter barriers (L2-scope atomics, sc1 loads of what other CUs wrote,
// no agent-scope fence: 1.3 us per phase, tools/proto/trunk_bounds.hip).  Unlike the resnet convolutions of VERDICT r02
// item 1 the block's weights are small (4.7 MB at 4x4, 1.2 MB at 8x8), so that every XCD streaming all of them costs 5 us, not 14-27.
// Replaces, per site: [k_gn_apply] + k_lin_wreg / k_igemm2 + k_attn + k_igemm2 (3 - 4 launches, 35 - 44 us at batch 64).
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((address_space(3))) void* lds_ptr_t;

struct ABP {
  const bf16* hn;      // [B][T][C] GroupNorm-ed input
  const bf16* x;       // [B][T][C] block input (residual)
  const bf16* wqkv;    // [3C][C] rows q | k | v
  const float* bqkv;   // [3C]
  const bf16* wo;      // [C][C]
  const float* bo;     // [C]
  bf16* qkv;           // scratch [B][T][3C]
  bf16* o;             // scratch [B][T][C]
  bf16* y;             // [B][T][C]
  float* stats;        // [B][C][2] (sum, sum of squares over the sample's T tokens) of y
  unsigned* sync;      // [8][32] per XCD: +0 ticket, +1 arrivals, +2 departures; word 256: error
  int B, C, heads;
  float scale_log2e;   // softmax scale x log2(e)
  unsigned long long* dbg;   // optional [256][8] wall-clock stamps (100 MHz) per workgroup
  int flags;                 // timing decomposition (garbage results): 1 no LDS-DMA, 2 no fragment reads / MFMAs, 4 no attention math
};

__device__ __forceinline__ int xcc_id() {
  int v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 7;
}

// 16-byte load that bypasses this CU's L1 (another CU of the XCD wrote the data during this launch): buffer load with sc1
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
__device__ __forceinline__ bf16x8 ld_sc1(__amdgpu_buffer_rsrc_t r, size_t elem_off) {
  return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r, (int)(elem_off * 2), 0, 16));
}

// per-XCD barrier: every thread's stores have left (vmcnt(0)), one lane arrives on the XCD's counter and polls it
__device__ __forceinline__ bool xcd_barrier(unsigned* cnt, unsigned target, unsigned* err, volatile int* ok_p) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    unsigned spins = 0;
    int ok = 1;
    while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1u << 21)) {
        ok = 0;
        __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
    }
    *ok_p = ok;
  }
  __syncthreads();
  return *ok_p != 0;
}

// 64 rows x (16 TN) columns of  out = A W^T :  A [rows][K] row-major (K contiguous), W [n][K].  4 waves, wave w = rows 16 w .. +15.
// BOTH operands stream through an NS-deep LDS ring in FRAGMENT order (LDS-DMA: per-lane gather address, lane-linear 1 KB
// destination per fragment), 64 K per stage = 2 TN weight fragments + 8 token fragments; NS - 1 stages (~100 KB per CU) are in
// flight, counted vmcnt, one workgroup barrier per stage.  (A first version with ONE stage in flight paid the L2 / HBM latency
// per stage: 60 us for the block.)  MFMA operands: A = W fragment (rows = couts), B = token fragment -> lane (li, lg) holds
// couts 4 lg + r of token li.  SC1: the token operand was written by other CUs of this XCD during this launch (L1 bypassed).
template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int TN, int NS, bool SC1>
__device__ __forceinline__ void gemm64(const bf16* A, size_t abytes, size_t a_row0, int K, const bf16* W, size_t wbytes, int n0,
                                       f32x4 (&acc)[TN], char* lds, int flags = 0) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  constexpr int FRAGS = 2 * TN + 8, STAGE = FRAGS * 1024, NPW = (FRAGS + 3) / 4;      // fragments per stage, DMA instructions per wave
  static_assert((NS - 2) * NPW < 64, "vmcnt is a 6-bit counter");
  __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, (int)wbytes, 0x00020000);
  __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (int)abytes, 0x00020000);
#pragma unroll
  for (int t = 0; t < TN; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nst = K / 64;
  // fragment f of a stage: f < 2 TN: weights (kk = f / TN, tile f % TN); else tokens (wave (f - 2 TN) / 2, kk = (f - 2 TN) % 2)
  auto issue = [&](int s) {
    char* base = lds + (s % NS) * STAGE;
    const bool live = s < nst;
    if (flags & 1) return;
#pragma unroll
    for (int j = 0; j < NPW; ++j) {
      int f = wave + 4 * j;
      if (f > FRAGS - 1) f = FRAGS - 1;                      // (waves whose share is one short re-issue the last fragment: uniform counts)
      lds_ptr_t dst = (lds_ptr_t)(base + f * 1024);
      if (f < 2 * TN) {
        // weights are packed FRAGMENT-major on the host: Wp[n / 16][k / 32][lane][8] - every fragment is 1 KB contiguous
        // (as 16 rows x 64-byte pieces, K * 2 bytes apart, the LDS-DMA gather ran at 18 GB/s per CU)
        const int kk = f / TN, t = f - kk * TN;
        const unsigned voff = live ? (unsigned)((((size_t)(n0 / 16 + t) * (K / 32) + 2 * s + kk) * 64 + lane) * 16) : 0x80000000u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, dst, 16, (int)voff, 0, 0, 0);
      } else {
        const int w = (f - 2 * TN) >> 1, kk = (f - 2 * TN) & 1;
        const unsigned voff = live ? (unsigned)(((a_row0 + 16 * w + li) * (size_t)K + 64 * s + 32 * kk + 8 * lg) * 2) : 0x80000000u;
        if constexpr (SC1) __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, dst, 16, (int)voff, 0, 0, 16);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, dst, 16, (int)voff, 0, 0, 0);
      }
    }
  };
#pragma unroll
  for (int s = 0; s < NS - 1; ++s) issue(s);
  for (int s = 0; s < nst; ++s) {
    wait_vm<(NS - 2) * NPW>();                               // stage s has landed (the NS - 2 younger ones may be in flight)
    __syncthreads();                                         // ... for every wave; the slot of stage s - 1 is free again
    issue(s + NS - 1);
    const char* base = lds + (s % NS) * STAGE;
    if (flags & 2) continue;
    const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(base + (2 * TN + 2 * wave) * 1024 + lane * 16);
    const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(base + (2 * TN + 2 * wave + 1) * 1024 + lane * 16);
#pragma unroll
    for (int t = 0; t < TN; ++t)
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(base + t * 1024 + lane * 16), a0, acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < TN; ++t)
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(base + (TN + t) * 1024 + lane * 16), a1, acc[t], 0, 0, 0);
  }
  wait_vm<0>();                                              // the zero-fill tail
  __syncthreads();
}

template <int T, int TNQ, int TNO, int NSQ, int NSO>
__global__ void __launch_bounds__(256) k_attn_block(ABP p) {
  extern __shared__ __attribute__((aligned(16))) char smem_all[];       // (no static LDS: it would shift the dynamic base off 16 bytes)
  volatile int* sh = reinterpret_cast<volatile int*>(smem_all);          // [0] XCD, [1] rank, [2] barrier verdict
  char* smem = smem_all + 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int C = p.C, C3 = 3 * C;
  const int nexp = gridDim.x / 8;                            // workgroups per XCD (checked by the barriers' targets)
  if (tid == 0) {
    const int x = xcc_id();
    sh[0] = x;
    sh[1] = (int)__hip_atomic_fetch_add(p.sync + x * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  const int x = sh[0], rank = sh[1];
  unsigned* sx = p.sync + x * 32;
  auto stamp = [&](int k) { if (p.dbg && tid == 0) p.dbg[blockIdx.x * 8 + k] = wall_clock64(); };
  stamp(0);
  const int spx = p.B / 8;                                   // samples of this XCD: [x spx, (x + 1) spx)
  const int M = spx * T;                                     // rows of this XCD (a multiple of 64)
  const size_t row_base = (size_t)x * M;

  // ---------------------------------------------------------------- phase Q: q | k | v = hn Wqkv^T + b
  {
    const int mb = M / 64, nb = (C3 / 16) / TNQ;
    for (int blk = rank; blk < mb * nb; blk += nexp) {
      const int bm = blk % mb, bn = blk / mb;
      const int n0 = bn * 16 * TNQ;
      const size_t r0 = row_base + (size_t)bm * 64;
      f32x4 acc[TNQ];
      gemm64<TNQ, NSQ, false>(p.hn, (size_t)p.B * T * C * 2, r0, C, p.wqkv, (size_t)C3 * C * 2, n0, acc, smem, p.flags);
      const size_t row = r0 + 16 * wave + li;
#pragma unroll
      for (int t = 0; t < TNQ; ++t) {
        const int n = n0 + 16 * t + 4 * lg;
        const f32x4 b = *reinterpret_cast<const f32x4*>(p.bqkv + n);
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (bf16)(acc[t][r] + b[r]);
        *reinterpret_cast<bf16x4*>(p.qkv + row * C3 + n) = o;
      }
    }
  }
  stamp(1);
  if (!xcd_barrier(sx + 1, (unsigned)nexp, p.sync + 256, sh + 2)) return;
  stamp(2);

  // ---------------------------------------------------------------- phase A: softmax(q k^T scale) v per (sample, head)
  {
    __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc((void*)p.qkv, 0, (int)((size_t)p.B * T * C3 * 2), 0x00020000);
    float* sK = reinterpret_cast<float*>(smem) + wave * (2 * T * 24);      // wave-private [T][24] K, then [T][24] V
    float* sV = sK + T * 24;
    const int ntask = spx * p.heads;
    for (int task = rank * 4 + wave; task < ntask; task += nexp * 4) {
      const int sl = task / p.heads, h = task - sl * p.heads;
      const size_t r0 = row_base + (size_t)sl * T;
      float q[24];
      if (lane < T) {
        const size_t qo = (r0 + lane) * C3 + h * 24;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const bf16x8 vq = ld_sc1(rq, qo + 8 * c), vk = ld_sc1(rq, qo + C + 8 * c), vv = ld_sc1(rq, qo + 2 * C + 8 * c);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            q[8 * c + e] = (float)vq[e] * p.scale_log2e;
            sK[lane * 24 + 8 * c + e] = (float)vk[e];
            sV[lane * 24 + 8 * c + e] = (float)vv[e];
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
      if (lane < T && !(p.flags & 4)) {
        auto score = [&](int j) {
          float a = 0.f;
#pragma unroll
          for (int d = 0; d < 24; d += 4) {
            const f32x4 kv = *reinterpret_cast<const f32x4*>(sK + j * 24 + d);
            a = fmaf(q[d], kv[0], a); a = fmaf(q[d + 1], kv[1], a); a = fmaf(q[d + 2], kv[2], a); a = fmaf(q[d + 3], kv[3], a);
          }
          return a;
        };
        // two passes over the keys (row maximum, then exp / sum / P V with the scores recomputed): T score registers would
        // push the T = 64 form over 256 VGPRs next to the GEMM phases' accumulators
        float m = -1e30f;
#pragma unroll 8
        for (int j = 0; j < T; ++j) m = fmaxf(m, score(j));
        float o[24];
#pragma unroll
        for (int d = 0; d < 24; ++d) o[d] = 0.f;
        float sum = 0.f;
#pragma unroll 4
        for (int j = 0; j < T; ++j) {
          const float pj = __builtin_amdgcn_exp2f(score(j) - m);
          sum += pj;
#pragma unroll
          for (int d = 0; d < 24; d += 4) {
            const f32x4 vv = *reinterpret_cast<const f32x4*>(sV + j * 24 + d);
            o[d] = fmaf(pj, vv[0], o[d]); o[d + 1] = fmaf(pj, vv[1], o[d + 1]); o[d + 2] = fmaf(pj, vv[2], o[d + 2]); o[d + 3] = fmaf(pj, vv[3], o[d + 3]);
          }
        }
        const float inv = 1.0f / sum;
        bf16* op = p.o + (r0 + lane) * C + h * 24;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          bf16x8 ov;
#pragma unroll
          for (int e = 0; e < 8; ++e) ov[e] = (bf16)(o[8 * c + e] * inv);
          *reinterpret_cast<bf16x8*>(op + 8 * c) = ov;
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
  stamp(3);
  if (!xcd_barrier(sx + 1, (unsigned)(2 * nexp), p.sync + 256, sh + 2)) return;
  stamp(4);

  // ---------------------------------------------------------------- phase O: y = o Wo^T + b + x, GroupNorm partial sums of y
  {
    float* sS = reinterpret_cast<float*>(smem + NSO * (2 * TNO + 8) * 1024);   // [4 waves][16 TNO couts][2] behind the ring
    const int mb = M / 64, nb = (C / 16) / TNO;
    for (int blk = rank; blk < mb * nb; blk += nexp) {
      const int bm = blk % mb, bn = blk / mb;
      const int n0 = bn * 16 * TNO;
      const size_t r0 = row_base + (size_t)bm * 64;
      f32x4 acc[TNO];
      gemm64<TNO, NSO, true>(p.o, (size_t)p.B * T * C * 2, r0, C, p.wo, (size_t)C * C * 2, n0, acc, smem, p.flags);
      const size_t row = r0 + 16 * wave + li;
#pragma unroll
      for (int t = 0; t < TNO; ++t) {
        const int n = n0 + 16 * t + 4 * lg;
        const f32x4 b = *reinterpret_cast<const f32x4*>(p.bo + n);
        const bf16x4 res = *reinterpret_cast<const bf16x4*>(p.x + row * C + n);
        bf16x4 ov;
        float s1[4], s2[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          ov[r] = (bf16)(acc[t][r] + b[r] + (float)res[r]);
          const float v = (float)ov[r];
          s1[r] = v;
          s2[r] = v * v;
        }
        *reinterpret_cast<bf16x4*>(p.y + row * C + n) = ov;
        // sums over the tokens of a sample: T >= 16: the wave's 16 rows belong to one sample (all 16 lanes li);
        // T = 4: groups of 4 lanes
        constexpr int RED = T >= 16 ? 16 : T;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
          for (int off = 1; off < RED; off <<= 1) {
            s1[r] += __shfl_xor(s1[r], off, 64);
            s2[r] += __shfl_xor(s2[r], off, 64);
          }
        }
        if constexpr (T <= 16) {
          if (li % RED == 0) {
            const size_t b_ = (r0 + 16 * wave + li) / T;
#pragma unroll
            for (int r = 0; r < 4; ++r) *reinterpret_cast<f32x2*>(p.stats + (b_ * C + n + r) * 2) = f32x2{s1[r], s2[r]};
          }
        } else {
          if (li == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) *reinterpret_cast<f32x2*>(sS + ((wave * TNO + t) * 16 + 4 * lg + r) * 2) = f32x2{s1[r], s2[r]};
          }
        }
      }
      if constexpr (T > 16) {       // T = 64: the block's 64 rows are one sample: the four waves' sums meet in LDS (fixed order)
        __syncthreads();
        for (int c = tid; c < 16 * TNO; c += 256) {
          float a1 = 0.f, a2 = 0.f;
          for (int w = 0; w < 4; ++w) {
            const f32x2 v = *reinterpret_cast<const f32x2*>(sS + ((w * TNO + c / 16) * 16 + c % 16) * 2);
            a1 += v[0];
            a2 += v[1];
          }
          *reinterpret_cast<f32x2*>(p.stats + ((r0 / T) * C + n0 + c) * 2) = f32x2{a1, a2};
        }
        __syncthreads();
      }
    }
  }
  stamp(5);
  // ---------------------------------------------------------------- leave: the last workgroup of the XCD re-zeroes its words
  if (tid == 0) {
    const unsigned old = __hip_atomic_fetch_add(sx + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old == (unsigned)nexp - 1) {
      __hip_atomic_store(sx + 0, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(sx + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(sx + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

template <int T, int TNQ, int TNO, int NSQ, int NSO>
static int launch(const ABP& p, hipStream_t st) {
  constexpr int lq = NSQ * (2 * TNQ + 8) * 1024, lo = NSO * (2 * TNO + 8) * 1024 + 4 * TNO * 16 * 2 * 4, la = 4 * 2 * T * 24 * 4;
  constexpr int lds = 64 + (lq > lo ? (lq > la ? lq : la) : (lo > la ? lo : la));
  static_assert(lds <= 160 * 1024, "LDS");
  static bool set = false;
  if (!set) {
    (void)hipFuncSetAttribute((const void*)k_attn_block<T, TNQ, TNO, NSQ, NSO>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    set = true;
  }
  k_attn_block<T, TNQ, TNO, NSQ, NSO><<<256, 256, lds, st>>>(p);
  return (int)hipGetLastError();
}

// T = 16 (4x4 planes, C = 768: 128 rows per XCD at batch 64) and T = 64 (8x8 planes, C = 384: 512 rows per XCD)
extern "C" int ab_attn_block(const void* hn, const void* x, const void* wqkv, const float* bqkv, const void* wo, const float* bo,
                             void* qkv, void* o, void* y, float* stats, unsigned* sync, int B, int T, int C, int heads, float scale,
                             void* stream, unsigned long long* dbg, int flags) {
  ABP p;
  p.hn = (const bf16*)hn; p.x = (const bf16*)x; p.wqkv = (const bf16*)wqkv; p.bqkv = bqkv; p.wo = (const bf16*)wo; p.bo = bo;
  p.qkv = (bf16*)qkv; p.o = (bf16*)o; p.y = (bf16*)y; p.stats = stats; p.sync = sync;
  p.B = B; p.C = C; p.heads = heads; p.scale_log2e = scale * 1.4426950408889634f; p.dbg = dbg; p.flags = flags;
  if (B % 8 || ((B / 8) * T) % 64 || C % 64) return -1;
  const int M = (B / 8) * T, mb = M / 64;
  if (32 % mb) return -2;
  const int nbq = 32 / mb, nbo = 32 / mb;                    // one block per workgroup in both GEMM phases
  const int tnq = (3 * C / 16) / nbq, tno = (C / 16) / nbo;
  if (T == 16 && tnq == 9 && tno == 3) return launch<16, 9, 3, 5, 8>(p, (hipStream_t)stream);
  if (T == 64 && tnq == 18 && tno == 6) return launch<64, 18, 6, 3, 6>(p, (hipStream_t)stream);
  return -3;
}
