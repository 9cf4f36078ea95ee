// trunk.hip - the 2x2 LEVEL of the UNet as ONE cooperative launch (round 5).
//
// Reference: diffusers UNet2DModel.forward (down_blocks[-1] -> mid_block -> up_blocks[0] of reference
// configs/ldm/model_unet.json) with the alias-free surgery of af_api.py:70-83: 7 ResnetBlock2D (norm -> WarpedNonlinearity
// (af_blocks.py:19-28) -> conv3x3 -> + time embedding -> norm -> WarpedNonlinearity -> conv3x3 -> + shortcut) and the mid block's
// self-attention (AttnProcessor2_0) on 2 x 2 planes.  As separate launches that is 51 kernels of 5 - 17 us = 0.36 ms of the
// 4.96 ms batch-64 step for < 1 % of its flops (and 0.3 ms of the 2.1 ms batch-1 step): every launch is at its latency floor.
//
// Here: one persistent workgroup per CU walks a PROGRAM of phases (built by the host, afldm_amd/trunk.py) separated by grid
// barriers.  A 3x3 convolution on a 2x2 plane is one dense layer over the flattened plane (models/blocks.py:
// packed_conv_dense2x2), so the level is three phase kinds:
//   GEMM  : partial products  slab[ki][rows][192 ni ..] = A[rows][192 ki ..] W[192 ni ..][192 ki ..]^T  - a unit is ONE 192 x 192
//           weight block (73.7 KB, stored contiguously by the host), up to 64 rows; 256 units for a 3072 x 3072 layer = one per
//           CU, so a CU moves 74 KB of weights + 25 KB of activations + 49 KB of partial sums per layer instead of the ~500 KB a
//           64 x 16 x 3072 tile would pull through its one L2 port.  The block is brought into LDS (swizzled LDS-DMA) DURING
//           THE PREVIOUS PHASE, i.e. before the barrier: the cold weight stream - what the dense layers are bound by - runs
//           under the other phases.  Virtual concats are two jobs of one phase (one per tensor), conv_shortcut is a third.
//   RED   : one thread per (sample, channel) plane sums the K-split slabs in slab order, adds bias + time embedding +
//           residual, rounds to bf16 (the value a stored tensor would have), forms the NEXT GroupNorm's statistics inside the
//           workgroup (192 channels = whole groups of one sample; fp64 finish), applies it and the N = 2 WarpedNonlinearity
//           (k_af_act_slabs' arithmetic), stores the activated operand of the next GEMM (and the raw tensor where a residual
//           / skip / shortcut reads it later).  A skip connection's half of a concatenated norm1 is a second job of the phase.
//   ATTN  : (sample, head) units: q / k / v from the slabs (+ bias, rounded), 4 x 4 scores, softmax, P V.
// Grid barrier: stores drained (vmcnt(0)) + agent-scope release, one arrival counter polled by one lane per workgroup,
// agent-scope acquire; the counter is zeroed by the last workgroup to leave the launch (graph replays start clean).  All
// workgroups must be resident at once (one per CU; the host launches min(CUs, 256)): the launch needs the GPU to itself -
// a second cooperative launch in flight on another stream could starve both (DenoiseEngine(cooperative=False) for
// concurrent engines).
#include "conv_common.hpp"
#include "../../include/afldm_hip_experimental.h"      // (libafldm_exp.so: not part of the product library)

namespace afldm {

constexpr int kTB = 192;                       // block edge: couts per unit = K per unit = channels per reduce unit
constexpr int kWBytes = kTB * kTB * 2;         // one weight block
constexpr unsigned long long kExtTag = 0xE;    // pointer fields with (v >> 60) == kExtTag are ext[(v >> 56) & 15] + (v & 0xFFFFFFFFFF)

enum { PH_GEMM = 1, PH_RED = 2, PH_ATTN = 3 };

struct GemmJob {
  unsigned long long A;        // bf16 [rows][a_ld]
  unsigned long long W;        // bf16 blocks [nsplit][ksplit][192][192]
  unsigned long long slab;     // fp32 [.][rows][slab_ld]
  int a_ld, rows, nsplit, ksplit, slab_ld, slab0;
  long long slab_stride;       // elements between slabs
};
struct RedJob {
  unsigned long long slab;     // fp32 slabs (nslab > 0) ...
  unsigned long long src;      // ... or an existing bf16 tensor [B][4][C] (nslab == 0)
  unsigned long long bias, bias2;    // fp32 [C] or 0 (bias2: conv_shortcut's)
  unsigned long long residual; // bf16 [B][4][C] or 0
  unsigned long long out_raw;  // bf16 [B][4][C] or 0
  unsigned long long gamma, beta;    // fp32 [C] (already offset to this tensor's channels) or 0
  unsigned long long out_act;  // bf16 [B][4][C] or 0
  long long slab_stride;
  int nslab, temb_off, cpg, mode;    // temb_off < 0: none; mode 0: raw only, 1: GroupNorm, 2: GroupNorm + WarpedNonlinearity
  int B, C;
  float eps;
  int pad;
};
struct AttnJob {
  unsigned long long slab, bias, out;      // slabs fp32 [nslab][4B][3C]; bias fp32 [3C]; out bf16 [4B][C]
  long long slab_stride;
  int nslab, B, C, heads;
  float scale;
  int pad;
};
struct Phase {
  int type, njobs;
  union {
    GemmJob g[3];
    RedJob r[2];
    AttnJob a;
  };
};
struct TrunkP {
  const Phase* phases;
  int nphases;
  const void* ext[4];          // 0: level input, 1: level output, 2: time-embedding row(s)
  int temb_stride;
  const float* U;              // [4][2]
  const float* D;              // [2][4]
  unsigned* sync;              // [0] arrivals, [1] departures
  unsigned* err;
  unsigned long long* trace;   // optional [nphases + 1][2] s_memtime stamps of workgroup 0 (phase start / barrier passed)
};

// Tensors that one phase writes and a later phase reads on ANOTHER XCD travel with agent-scope (sc1) accesses - stores written
// through to memory, loads that bypass this CU's L1 and are not served from a stale line of this XCD's L2 - so that the grid
// barrier needs NO cache maintenance: an agent-scope release / acquire pair (L2 write-back + invalidate) measured 9 - 12 us per
// barrier here and left every phase to start on a cold L2 (weights, descriptors, parameters included).
typedef __attribute__((ext_vector_type(4))) unsigned int tr_u4;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t tr_buf(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7FFFFFFF, 0x00020000);
}
__device__ __forceinline__ float tr_ld32(__amdgpu_buffer_rsrc_t r, size_t byte_off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)byte_off, 0, 16));
}
__device__ __forceinline__ float tr_ldbf(__amdgpu_buffer_rsrc_t r, size_t byte_off) {
  const unsigned short v = __builtin_amdgcn_raw_buffer_load_b16(r, (int)byte_off, 0, 16);
  return __builtin_bit_cast(float, (unsigned)v << 16);
}
__device__ __forceinline__ void tr_stbf(__amdgpu_buffer_rsrc_t r, size_t byte_off, bf16 v) {
  __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, v), r, (int)byte_off, 0, 16);
}
__device__ __forceinline__ bf16x8 tr_ld128(__amdgpu_buffer_rsrc_t r, size_t byte_off) {
  return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 16));
}
__device__ __forceinline__ void tr_st128(__amdgpu_buffer_rsrc_t r, size_t byte_off, const f32x4& v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(tr_u4, v), r, (int)byte_off, 0, 16);
}

template <typename P>
__device__ __forceinline__ P* tr_resolve(const TrunkP& a, unsigned long long v) {
  if ((v >> 60) == kExtTag) return reinterpret_cast<P*>(const_cast<char*>(static_cast<const char*>(a.ext[(v >> 56) & 15])) + (v & 0xFFFFFFFFFFull));
  return reinterpret_cast<P*>(v);
}

// unit u of a GEMM phase -> (job, row tile, n split, k split); false past the end
__device__ __forceinline__ bool tr_unit(const Phase& ph, int u, int& j, int& rt, int& ni, int& ki) {
  for (j = 0; j < ph.njobs; ++j) {
    const GemmJob& g = ph.g[j];
    const int nrt = (g.rows + 63) / 64, n = nrt * g.nsplit * g.ksplit;
    if (u < n) {
      rt = u / (g.nsplit * g.ksplit);
      const int r = u - rt * g.nsplit * g.ksplit;
      ni = r / g.ksplit;
      ki = r - ni * g.ksplit;
      return true;
    }
    u -= n;
  }
  return false;
}

// the weight block of unit (job j, ni, ki) -> LDS buffer `dst` by LDS-DMA (18 instructions per wave), rows of 24 16-byte
// chunks with chunk c of row r stored at position c ^ (r & 7): the 16 rows of a fragment read hit 8 distinct 4-bank groups twice
__device__ __forceinline__ void tr_issue_w(const TrunkP& a, const Phase& ph, int j, int ni, int ki, char* dst, int tid) {
  const GemmJob& g = ph.g[j];
  const bf16* w = tr_resolve<const bf16>(a, g.W) + ((size_t)ni * g.ksplit + ki) * (kTB * kTB);
  __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, kWBytes, 0x00020000);
  const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
  for (int i = 0; i < 18; ++i) {
    const int inst = wave * 18 + i;
    const int pos = inst * 64 + lane;                    // 16-byte slot in the LDS image
    const int r = pos / 24, cs = pos - r * 24;
    const int c = (cs & ~7) | ((cs ^ r) & 7);
    lds_ptr_t d = (lds_ptr_t)(dst + inst * 1024);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, d, 16, (r * 24 + c) * 16, 0, 0, 0);
  }
}

__device__ __forceinline__ void tr_gemm_unit(const TrunkP& a, const GemmJob& g, int rt, int ni, int ki, const char* wl, int tid) {
  const int wave = tid >> 6, lane = tid & 63, li = lane & 15, lg = lane >> 4;
  const __amdgpu_buffer_rsrc_t rA = tr_buf(tr_resolve<const bf16>(a, g.A));
  const int row0 = rt * 64;
  const int nr = g.rows - row0 >= 64 ? 4 : (g.rows - row0 + 15) / 16;     // row tiles of 16 in this unit (wave-uniform)
  bf16x8 bfr[4][6];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = row0 + 16 * r + li;
    const bool ok = r < nr && row < g.rows;
#pragma unroll
    for (int ks = 0; ks < 6; ++ks) {
      bf16x8 v = Mma<bf16>::zero();
      if (ok) v = tr_ld128(rA, ((size_t)row * g.a_ld + ki * kTB + ks * 32 + lg * 8) * 2);
      bfr[r][ks] = v;
    }
  }
  f32x4 acc[3][4];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[t][r] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < 6; ++ks) {
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const int n = wave * 48 + t * 16 + li;
      const int c = ks * 4 + lg, cs = (c & ~7) | ((c ^ n) & 7);
      const bf16x8 af = *reinterpret_cast<const bf16x8*>(wl + (n * 24 + cs) * 16);
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (r < nr) acc[t][r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfr[r][ks], acc[t][r], 0, 0, 0);
    }
  }
  const __amdgpu_buffer_rsrc_t rS = tr_buf(tr_resolve<float>(a, g.slab) + (size_t)(g.slab0 + ki) * g.slab_stride);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = row0 + 16 * r + li;
    if (r < nr && row < g.rows) {
#pragma unroll
      for (int t = 0; t < 3; ++t)
        tr_st128(rS, ((size_t)row * g.slab_ld + ni * kTB + wave * 48 + t * 16 + 4 * lg) * 4, acc[t][r]);
    }
  }
}

// one (sample, 192-channel chunk) unit of a reduce job; all 256 threads call (barriers inside), threads >= 192 idle
__device__ __forceinline__ void tr_red_unit(const TrunkP& a, const RedJob& q, int unit, float* sS, float* sM, int tid) {
  constexpr int P = 4;
  const int chunks = q.C / kTB;
  const int b = unit / chunks, c = (unit - b * chunks) * kTB + tid;
  const bool live = tid < kTB;
  float X[P], s1 = 0.f, s2 = 0.f, gm = 1.f, bt = 0.f;
  if (live) {
    const size_t e0 = ((size_t)b * P) * q.C + c;
    if (q.nslab > 0) {
      const __amdgpu_buffer_rsrc_t rs = tr_buf(tr_resolve<const float>(a, q.slab));
      // every slab term of the plane in flight at once (up to 32 slabs x 4 pixels), added in slab order
      float v[P];
#pragma unroll
      for (int px = 0; px < P; ++px) v[px] = 0.f;
      float res[P];
      const bool has_res = q.residual != 0;
      if (has_res) {
        const __amdgpu_buffer_rsrc_t rr = tr_buf(tr_resolve<const bf16>(a, q.residual));
#pragma unroll
        for (int px = 0; px < P; ++px) res[px] = tr_ldbf(rr, (e0 + (size_t)px * q.C) * 2);
      }
      for (int z0 = 0; z0 < q.nslab; z0 += 16) {
        float t[16][P];
#pragma unroll
        for (int k = 0; k < 16; ++k)
#pragma unroll
          for (int px = 0; px < P; ++px)
            t[k][px] = z0 + k < q.nslab ? tr_ld32(rs, ((size_t)(z0 + k) * q.slab_stride + e0 + (size_t)px * q.C) * 4) : 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k)
#pragma unroll
          for (int px = 0; px < P; ++px) v[px] += t[k][px];
      }
      float add = 0.f;
      if (q.bias) add = tr_resolve<const float>(a, q.bias)[c];
      if (q.bias2) add += tr_resolve<const float>(a, q.bias2)[c];
      float tv = 0.f;
      if (q.temb_off >= 0) tv = to_f32(static_cast<const bf16*>(a.ext[2])[(size_t)b * a.temb_stride + q.temb_off + c]);
      const __amdgpu_buffer_rsrc_t rw = tr_buf(q.out_raw ? tr_resolve<bf16>(a, q.out_raw) : nullptr);
#pragma unroll
      for (int px = 0; px < P; ++px) {
        float vv = v[px];
        if (q.bias || q.bias2) vv += add;
        if (q.temb_off >= 0) vv += tv;
        if (has_res) vv += res[px];
        const bf16 rt = from_f32<bf16>(vv);
        if (q.out_raw) tr_stbf(rw, (e0 + (size_t)px * q.C) * 2, rt);
        X[px] = to_f32(rt);                                 // the value a stored tensor would hand the normalisation
      }
    } else {
      const __amdgpu_buffer_rsrc_t rsrc = tr_buf(tr_resolve<const bf16>(a, q.src));
#pragma unroll
      for (int px = 0; px < P; ++px) X[px] = tr_ldbf(rsrc, (e0 + (size_t)px * q.C) * 2);
    }
    if (q.mode) {
      gm = tr_resolve<const float>(a, q.gamma)[c];
      bt = tr_resolve<const float>(a, q.beta)[c];
#pragma unroll
      for (int px = 0; px < P; ++px) {
        s1 += X[px];
        s2 = fmaf(X[px], X[px], s2);
      }
    }
  }
  if (q.mode == 0) return;                                  // (uniform)
  __syncthreads();                                          // the previous unit's readers of sS / sM are done
  if (live) {
    sS[2 * tid] = s1;
    sS[2 * tid + 1] = s2;
  }
  __syncthreads();
  const int ng = kTB / q.cpg;
  if (tid < ng) {
    double a1 = 0.0, a2 = 0.0;
    for (int k = 0; k < q.cpg; ++k) {
      a1 += (double)sS[2 * (tid * q.cpg + k)];
      a2 += (double)sS[2 * (tid * q.cpg + k) + 1];
    }
    float mean, rstd;
    gn_mean_rstd(a1, a2, (double)P * q.cpg, q.eps, mean, rstd);
    sM[2 * tid] = mean;
    sM[2 * tid + 1] = rstd;
  }
  __syncthreads();
  if (!live) return;
  {
    const float mean = sM[2 * (tid / q.cpg)], rstd = sM[2 * (tid / q.cpg) + 1];
    const float sc = rstd * gm, sh = bt - mean * sc;
#pragma unroll
    for (int px = 0; px < P; ++px) X[px] = X[px] * sc + sh;
  }
  const __amdgpu_buffer_rsrc_t ry = tr_buf(tr_resolve<bf16>(a, q.out_act));
  const size_t y0 = ((size_t)b * P) * q.C + c;
  if (q.mode == 1) {
#pragma unroll
    for (int px = 0; px < P; ++px) tr_stbf(ry, (y0 + (size_t)px * q.C) * 2, from_f32<bf16>(X[px]));
    return;
  }
  // WarpedNonlinearity on the 2 x 2 plane: D silu(U X U^T) D^T (k_af_act_small<2> / k_af_act_slabs' arithmetic and order)
  constexpr int N = 2, H2 = 4;
  const float* __restrict__ U = a.U;
  const float* __restrict__ D = a.D;
  float Y[N][N];
#pragma unroll
  for (int h = 0; h < N; ++h)
#pragma unroll
    for (int w = 0; w < N; ++w) Y[h][w] = 0.f;
#pragma unroll
  for (int hp = 0; hp < H2; ++hp) {
    float t1[N];
#pragma unroll
    for (int w = 0; w < N; ++w) {
      float s = 0.f;
#pragma unroll
      for (int h = 0; h < N; ++h) s = fmaf(U[hp * N + h], X[h * N + w], s);
      t1[w] = s;
    }
    float sz[H2];
#pragma unroll
    for (int wp = 0; wp < H2; ++wp) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < N; ++w) s = fmaf(U[wp * N + w], t1[w], s);
      sz[wp] = silu_f(s);
    }
#pragma unroll
    for (int w = 0; w < N; ++w) {
      float s = 0.f;
#pragma unroll
      for (int wp = 0; wp < H2; ++wp) s = fmaf(D[w * H2 + wp], sz[wp], s);
#pragma unroll
      for (int h = 0; h < N; ++h) Y[h][w] = fmaf(D[h * H2 + hp], s, Y[h][w]);
    }
  }
#pragma unroll
  for (int h = 0; h < N; ++h)
#pragma unroll
    for (int w = 0; w < N; ++w) tr_stbf(ry, (y0 + (size_t)(h * N + w) * q.C) * 2, from_f32<bf16>(Y[h][w]));
}

// attention of one sample's 4 tokens, all heads (a unit = a sample): the workgroup first reduces the sample's q | k | v rows from
// the slabs with coalesced loads (+ bias, rounded to bf16 as the projections' stored outputs are) into LDS, then thread
// (head, token) forms its 4 scores in fp32, rounds the softmax weights to bf16 before P V (k_attn's rounding points) and stores
// its d outputs with one rounding.
__device__ __forceinline__ void tr_attn(const TrunkP& a, const AttnJob& q, float* sQ, int tid, int wg, int nwg) {
  const int d = q.C / q.heads, W3 = 3 * q.C;                // d <= 32; sQ: [4][3C] floats
  const __amdgpu_buffer_rsrc_t rs = tr_buf(tr_resolve<const float>(a, q.slab));
  const float* bias = tr_resolve<const float>(a, q.bias);
  const __amdgpu_buffer_rsrc_t ro = tr_buf(tr_resolve<bf16>(a, q.out));
  for (int b = wg; b < q.B; b += nwg) {
    __syncthreads();                                        // the previous sample's readers are done
    for (int i = tid; i < 4 * W3; i += 256) {
      const int t = i / W3, col = i - t * W3;
      const size_t e = (size_t)(b * 4 + t) * W3 + col;
      float v = 0.f;
      for (int z = 0; z < q.nslab; ++z) v += tr_ld32(rs, ((size_t)z * q.slab_stride + e) * 4);
      sQ[i] = to_f32(from_f32<bf16>(v + bias[col]));
    }
    __syncthreads();
    for (int u = tid; u < q.heads * 4; u += 256) {
      const int h = u >> 2, t = u & 3;
      const float* qr = sQ + t * W3 + h * d;
      float sc[4], m = -3.0e38f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float* kj = sQ + j * W3 + q.C + h * d;
        float s = 0.f;
        for (int e = 0; e < d; ++e) s = fmaf(qr[e], kj[e], s);
        sc[j] = s * q.scale;
        m = fmaxf(m, sc[j]);
      }
      float den = 0.f, pw[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        pw[j] = to_f32(from_f32<bf16>(__builtin_amdgcn_exp2f((sc[j] - m) * 1.4426950408889634f)));
        den += pw[j];
      }
      const float inv = 1.0f / den;
      for (int e = 0; e < d; ++e) {
        float o = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) o = fmaf(pw[j], sQ[j * W3 + 2 * q.C + h * d + e], o);
        tr_stbf(ro, ((size_t)(b * 4 + t) * q.C + h * d + e) * 2, from_f32<bf16>(o * inv));
      }
    }
  }
}

__global__ void __launch_bounds__(256) k_trunk(TrunkP a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  auto wbuf = [&](int i) -> char* { return smem + i * kWBytes; };
  float* sS = reinterpret_cast<float*>(smem + 2 * kWBytes);       // [192][2]
  float* sM = sS + 2 * kTB;                                       // [8][2]
  float* sQ = reinterpret_cast<float*>(smem);                     // attention: [4][3C] floats (the weight buffers are idle then)
  volatile int* s_okp = reinterpret_cast<volatile int*>(sM + 32);
  const int tid = threadIdx.x, wg = blockIdx.x, nwg = gridDim.x;
  const bool hier = (nwg & 7) == 0;                                 // workgroup ids round-robin over the 8 XCDs
  const int xcd = wg & 7;
  const unsigned per_xcd = (unsigned)(nwg >> 3);
  bool w_ready = false;                                            // buffer 0 holds (or is receiving) this workgroup's first block of the current phase
  for (int pi = 0; pi < a.nphases; ++pi) {
    const Phase& ph = a.phases[pi];
    if (a.trace && wg == 0 && tid == 0) a.trace[4 * pi] = __builtin_amdgcn_s_memtime();
    if (ph.type == PH_GEMM) {
      int j, rt, ni, ki, buf = 0;
      bool have = tr_unit(ph, wg, j, rt, ni, ki);
      if (have && !w_ready) tr_issue_w(a, ph, j, ni, ki, wbuf(0), tid);
      for (int u = wg; have; u += nwg) {
        int j2, rt2, ni2, ki2;
        const bool more = tr_unit(ph, u + nwg, j2, rt2, ni2, ki2);
        if (more) tr_issue_w(a, ph, j2, ni2, ki2, wbuf(buf ^ 1), tid);
        if (more) wait_vmcnt<18>();                                // this unit's block has landed (the next one may be in flight)
        else wait_vmcnt<0>();
        __syncthreads();
        tr_gemm_unit(a, ph.g[j], rt, ni, ki, wbuf(buf), tid);
        __syncthreads();                                           // every wave is done with the buffer before a later block lands in it
        have = more;
        j = j2; rt = rt2; ni = ni2; ki = ki2;
        buf ^= 1;
      }
      w_ready = false;
    } else if (ph.type == PH_RED) {
      // the first weight block of the NEXT phase starts its way into LDS now (weights are not produced here): the cold
      // fetch runs under this phase's work and is over when the barrier's release fence drains the memory counters
      if (pi + 1 < a.nphases && a.phases[pi + 1].type == PH_GEMM) {
        int j, rt, ni, ki;
        if (tr_unit(a.phases[pi + 1], wg, j, rt, ni, ki)) {
          tr_issue_w(a, a.phases[pi + 1], j, ni, ki, wbuf(0), tid);
          w_ready = true;
        }
      }
      int base = 0;
      for (int jb = 0; jb < ph.njobs; ++jb) {
        const RedJob& q = ph.r[jb];
        const int units = q.B * (q.C / kTB);
        // (a workgroup's units of all jobs: unit index continues across the jobs so that the load spreads over the grid)
        for (int u = wg - base % nwg; u < units; u += nwg) {
          if (u >= 0) tr_red_unit(a, q, u, sS, sM, tid);
        }
        base += units;
      }
    } else if (ph.type == PH_ATTN) {
      tr_attn(a, ph.a, sQ, tid, wg, nwg);
    }
    // ---- grid barrier, two levels: 256 pollers of ONE memory-side counter serialise at its home channel (measured: ~20 us
    // per barrier).  The workgroups of an XCD (ids = x mod 8) meet on a counter in their own L2; the last one in carries the
    // XCD's arrival to the chip-wide counter (8 arrivals, 8 pollers) and then raises the XCD's flag, which the others poll.
    wait_vmcnt<0>();
    if (a.trace && wg == 0 && tid == 0) a.trace[4 * pi + 1] = __builtin_amdgcn_s_memtime();
    __syncthreads();
    if (a.trace && wg == 0 && tid == 0) a.trace[4 * pi + 2] = __builtin_amdgcn_s_memtime();
    if (tid == 0) {
      const unsigned k1 = (unsigned)(pi + 1);
      int ok = 1;
      unsigned spins = 0;
      if (hier) {
        unsigned* xc = a.sync + 32 * (1 + xcd);             // word 0: arrivals of this XCD, word 16: its flag
        const unsigned old = __hip_atomic_fetch_add(xc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (old == k1 * per_xcd - 1) {
          __hip_atomic_fetch_add(a.sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          while (__hip_atomic_load(a.sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 8u * k1) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 22)) { ok = 0; break; }
          }
          __hip_atomic_exchange(xc + 16, ok ? k1 : 0xFFFFFFFFu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
          unsigned f;
          while ((f = __hip_atomic_load(xc + 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < k1) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 22)) { ok = 0; break; }
          }
          if (f == 0xFFFFFFFFu) ok = 0;
        }
      } else {
        __hip_atomic_fetch_add(a.sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(a.sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < k1 * (unsigned)nwg) {
          __builtin_amdgcn_s_sleep(1);
          if (++spins > (1u << 22)) { ok = 0; break; }
        }
      }
      if (!ok) __hip_atomic_store(a.err, 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      *s_okp = ok;
    }
    __syncthreads();
    if (a.trace && wg == 0 && tid == 0) a.trace[4 * pi + 3] = __builtin_amdgcn_s_memtime();
    if (!*s_okp) break;
  }
  wait_vmcnt<0>();
  __syncthreads();
  if (tid == 0) {
    const unsigned old = __hip_atomic_fetch_add(a.sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old == (unsigned)nwg - 1) {                                 // last one out: every counter returns to zero
      __hip_atomic_store(a.sync, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(a.sync + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  // an XCD's own lines live in ITS L2: its workgroups zero them there (L2 atomics) once all of them have passed the last barrier
  if (tid == 0 && hier) {
    unsigned* xc = a.sync + 32 * (1 + xcd);
    const unsigned oldx = __hip_atomic_fetch_add(xc + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (oldx == per_xcd - 1) {
      __hip_atomic_exchange(xc, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_exchange(xc + 16, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_exchange(xc + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
}

static unsigned long long* g_trunk_trace = nullptr;

}  // namespace afldm

using namespace afldm;

extern "C" int afldm_trunk_phase_bytes(void) { return (int)sizeof(Phase); }

extern "C" int afldm_trunk_trace(void* buf) {
  g_trunk_trace = (unsigned long long*)buf;
  return AFLDM_OK;
}

extern "C" int afldm_trunk_run(const void* phases, int nphases, const void* x_in, void* y_out, const void* temb, int temb_stride,
                               const float* U, const float* D, unsigned int* sync, size_t sync_bytes, afldm_stream_t stream) {
  AFLDM_REQUIRE(phases && nphases > 0 && x_in && y_out && U && D && sync, AFLDM_ENULL, "afldm_trunk_run: NULL argument");
  AFLDM_REQUIRE(sync_bytes >= (size_t)(8448 + 9 * 32) * 4, AFLDM_ESHAPE, "afldm_trunk_run: sync buffer too small");
  static int cus = 0;
  if (!cus) {
    hipDeviceProp_t prop;
    int dev = 0;
    (void)hipGetDevice(&dev);
    cus = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
  }
  TrunkP a;
  a.phases = (const Phase*)phases;
  a.nphases = nphases;
  a.ext[0] = x_in;
  a.ext[1] = y_out;
  a.ext[2] = temb;
  a.ext[3] = nullptr;
  a.temb_stride = temb_stride;
  a.U = U;
  a.D = D;
  a.sync = sync + 8448;                 // (128-byte lines: chip-wide arrivals / departures, then one line per XCD; zero between launches)
  a.err = sync + 8193;
  a.trace = g_trunk_trace;
  constexpr int lds = 2 * kWBytes + (2 * kTB + 16 + 32 + 4) * 4;
  static unsigned long long attr_set = 0;
  if (first_on_device(attr_set)) (void)hipFuncSetAttribute((const void*)k_trunk, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  const int grid = cus < 256 ? cus : 256;
  k_trunk<<<grid, 256, lds, (hipStream_t)stream>>>(a);
  return check_launch("afldm_trunk_run");
}
