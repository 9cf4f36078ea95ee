// conv.hip — 3x3 / 1x1 convolution and nn.Linear as implicit GEMM on MFMA (gfx950).
//
//   Y[m, n] = sum_k  Xcol[m, k] * W[n, k]       m = (b, oh, ow),  n = cout,  k = (kh, kw, ci)
//
// NHWC activations make k contiguous for both operands, so every MFMA operand is one 16-byte
// chunk of a K-contiguous row (see Mma<T> in common.hpp).  The weight rows are the MFMA "A"
// operand (i = cout) and the pixels the "B" operand (j = pixel): the accumulator of lane
// (j, g) then holds 4 CONSECUTIVE couts of ONE pixel, so the epilogue reads bias / time-embedding
// / residual and writes the NHWC output as 8-byte (bf16) or 16-byte (fp32) vectors.
//
// Tile: BM pixels x BN couts per workgroup, WGM x WGN waves, K step = KCH 64-byte row pieces
// (KCH*32 bf16 or KCH*16 fp32 channels of one (kh, kw) tap).  Zero padding, the M tail and the
// Cout tail are handled by zero-filling the staged chunk.  Global -> VGPR -> LDS staging with the
// next tile's loads in flight during the MFMAs (guide T14), two LDS buffers, one barrier per
// K step.  LDS rows are 64 B with the 16-B chunk index XOR-swizzled by the row so that the
// fragment read (16 rows x 4 chunks per ds_read_b128 wave access) is bank-conflict free.
// Optional split-K (grid.z) writes fp32 slabs that k_splitk_reduce folds together with the
// epilogue terms — used for the 2x2 / 4x4 levels where M = B*HW is too small to fill 256 CUs.
#include <stdlib.h>

#include "conv_common.hpp"

namespace afldm {

__device__ __forceinline__ int swz(int row) { return (4 - ((row >> 2) & 3)) & 3; }

template <typename T>
__device__ __forceinline__ void epilogue_store(const ConvP& p, int m, int n, f32x4 v) {
  // adds bias / temb / residual for couts n..n+3 of pixel m and stores them
  const T* temb = (const T*)p.temb;
  const T* res = (const T*)p.residual;
  T* y = (T*)p.y;
  const int HW = p.H * p.W;
  const int b = m / HW;
  if (p.y2 && n >= p.split_n) {   // second output: channel-major [B][Cout - split_n][HW] (V^T of a fused QKV GEMM)
    T* y2 = (T*)p.y2;
    const int pix = m - b * HW, c2 = p.Cout - p.split_n;
    for (int r = 0; r < 4 && n + r < p.Cout; ++r) {
      float sv = v[r] + (p.bias ? p.bias[n + r] : 0.f);
      y2[((size_t)b * c2 + (n + r - p.split_n)) * HW + pix] = from_f32<T>(sv);
    }
    return;
  }
  if (n + 3 < p.Cout && p.vec_ok) {
    if (p.bias) {
      f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + n);
      v += bv;
    }
    if (temb) {
      float a0, a1, a2, a3;
      load4<T>(temb + (size_t)b * p.temb_stride + n % p.temb_mod, a0, a1, a2, a3);
      v[0] += a0; v[1] += a1; v[2] += a2; v[3] += a3;
    }
    if (res) {
      float a0, a1, a2, a3;
      load4<T>(res + (size_t)m * p.res_ld + n, a0, a1, a2, a3);
      v[0] += a0; v[1] += a1; v[2] += a2; v[3] += a3;
    }
    if (p.out_mode == 0) {
      store4<T>(y + (size_t)m * p.y_ld + n, v[0], v[1], v[2], v[3]);
    } else {
      const int pix = m - b * HW;
#pragma unroll
      for (int r = 0; r < 4; ++r) y[((size_t)b * p.Cout + n + r) * HW + pix] = from_f32<T>(v[r]);
    }
  } else {
    const int pix = m - b * HW;
    for (int r = 0; r < 4; ++r) {
      if (n + r >= p.Cout) break;
      float s = v[r];
      if (p.bias) s += p.bias[n + r];
      if (temb) s += to_f32(temb[(size_t)b * p.temb_stride + (n + r) % p.temb_mod]);
      if (res) s += to_f32(res[(size_t)m * p.res_ld + n + r]);
      if (p.out_mode == 0) y[(size_t)m * p.y_ld + n + r] = from_f32<T>(s);
      else y[((size_t)b * p.Cout + n + r) * HW + pix] = from_f32<T>(s);
    }
  }
}

template <typename T, int BM, int BN, int WGM, int WGN, int KCH>
__global__ void __launch_bounds__(WGM* WGN * 64) k_igemm(ConvP p) {
  typedef Mma<T> MM;
  typedef typename MM::Chunk Chunk;
  constexpr int NT = WGM * WGN * 64;
  constexpr int EPC = MM::EPC;       // elements per 16-B chunk
  constexpr int EPR = 4 * EPC;       // elements per 64-B row piece
  constexpr int KSTEP = KCH * EPR;   // channels consumed per K step
  constexpr int WMS = BM / WGM, WNS = BN / WGN;
  constexpr int TM = WMS / 16, TN = WNS / 16;
  constexpr int RPT = NT / 4;        // rows staged per pass (4 chunk-lanes per row)
  constexpr int RX = BM / RPT, RW = BN / RPT;
  static_assert(BM % RPT == 0 && BN % RPT == 0, "tile rows must be a multiple of NT/4");
  static_assert(WMS % 16 == 0 && WNS % 16 == 0, "wave tile must be a multiple of 16");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  // layout: [stage][kc][row][64 B]   X rows first, then W rows
  constexpr int X_STAGE = KCH * BM * 64, W_STAGE = KCH * BN * 64;
  char* sX = smem;
  char* sW = smem + 2 * X_STAGE;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  const int li = lane & 15, lg = lane >> 4;

  const int ntiles = gridDim.x;
  const int tile = xcd_remap(blockIdx.x, ntiles);
  const int tile_m = tile / p.tiles_n, tile_n = tile % p.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int Ct = p.C1 + p.C2;
  const int cblocks = Ct / KSTEP;
  const int HW = p.H * p.W;
  const int pad = p.KS >> 1;

  // K range of this split
  const int ks = blockIdx.z;
  const int kbeg = (int)(((long long)p.ksteps * ks) / p.splitk);
  const int kend = (int)(((long long)p.ksteps * (ks + 1)) / p.splitk);

  // ---- per-thread staging coordinates
  const int srow = tid >> 2, schunk = tid & 3;
  int xb[RX], xoh[RX], xow[RX];
  bool xv[RX];
#pragma unroll
  for (int r = 0; r < RX; ++r) {
    int m = m0 + r * RPT + srow;
    xv[r] = m < p.M;
    int mm = xv[r] ? m : 0;
    int b = mm / HW, pix = mm - b * HW;
    xb[r] = b;
    xoh[r] = pix / p.W;
    xow[r] = pix - xoh[r] * p.W;
  }
  Chunk rx[KCH][RX], rw[KCH][RW];

  auto load_stage = [&](int kt) {
    const int tap = kt / cblocks;
    const int ci0 = (kt - tap * cblocks) * KSTEP;
    const int kh = tap / p.KS, kw = tap - kh * p.KS;
    const bool second = ci0 >= p.C1;
    const T* xsrc = second ? (const T*)p.x2 : (const T*)p.x1;
    const int Cs = second ? p.C2 : p.C1;
    const int cs0 = second ? ci0 - p.C1 : ci0;
#pragma unroll
    for (int r = 0; r < RX; ++r) {
      const int ih = xoh[r] + kh - pad, iw = xow[r] + kw - pad;
      const bool ok = xv[r] && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
      const T* src = xsrc + ((size_t)(xb[r] * p.H + (ok ? ih : 0)) * p.W + (ok ? iw : 0)) * Cs + cs0 + schunk * EPC;
#pragma unroll
      for (int kc = 0; kc < KCH; ++kc) rx[kc][r] = ok ? ld16<Chunk>(src + kc * EPR) : MM::zero();
    }
    const T* wbase = (const T*)p.w;
#pragma unroll
    for (int r = 0; r < RW; ++r) {
      const int n = n0 + r * RPT + srow;
      const bool ok = n < p.Cout;
      const T* src = wbase + ((size_t)(ok ? n : 0) * p.KS * p.KS + tap) * Ct + ci0 + schunk * EPC;
#pragma unroll
      for (int kc = 0; kc < KCH; ++kc) rw[kc][r] = ok ? ld16<Chunk>(src + kc * EPR) : MM::zero();
    }
  };
  auto store_stage = [&](int buf) {
#pragma unroll
    for (int kc = 0; kc < KCH; ++kc) {
#pragma unroll
      for (int r = 0; r < RX; ++r) {
        const int row = r * RPT + srow;
        st16<Chunk>(sX + buf * X_STAGE + (kc * BM + row) * 64 + ((schunk ^ swz(row)) << 4), rx[kc][r]);
      }
#pragma unroll
      for (int r = 0; r < RW; ++r) {
        const int row = r * RPT + srow;
        st16<Chunk>(sW + buf * W_STAGE + (kc * BN + row) * 64 + ((schunk ^ swz(row)) << 4), rw[kc][r]);
      }
    }
  };

  f32x4 acc[TN][TM];
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (kbeg < kend) {
    load_stage(kbeg);
    store_stage(0);
    __syncthreads();
    for (int kt = kbeg; kt < kend; ++kt) {
      const int buf = (kt - kbeg) & 1;
      const bool more = kt + 1 < kend;
      if (more) load_stage(kt + 1);
#pragma unroll
      for (int kc = 0; kc < KCH; ++kc) {
        Chunk a[TN], b[TM];
#pragma unroll
        for (int t = 0; t < TN; ++t) {
          const int row = wn * WNS + t * 16 + li;
          a[t] = ld16<Chunk>(sW + buf * W_STAGE + (kc * BN + row) * 64 + ((lg ^ swz(row)) << 4));
        }
#pragma unroll
        for (int t = 0; t < TM; ++t) {
          const int row = wm * WMS + t * 16 + li;
          b[t] = ld16<Chunk>(sX + buf * X_STAGE + (kc * BM + row) * 64 + ((lg ^ swz(row)) << 4));
        }
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
          for (int tm = 0; tm < TM; ++tm) MM::mma(acc[tn][tm], a[tn], b[tm]);
      }
      if (more) store_stage(buf ^ 1);
      __syncthreads();
    }
  }

  // ---- epilogue: lane (j = li, g = lg) holds couts n..n+3 (n = .. + 4*lg) of pixel m (.. + li)
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      const int m = m0 + wm * WMS + tm * 16 + li;
      const int n = n0 + wn * WNS + tn * 16 + 4 * lg;
      if (m >= p.M || n >= p.Cout) continue;
      if (p.splitk > 1) {
        float* dst = p.ws + ((size_t)ks * p.M + m) * p.Cout + n;
        if (n + 3 < p.Cout) *reinterpret_cast<f32x4*>(dst) = acc[tn][tm];
        else
          for (int r = 0; r < 4 && n + r < p.Cout; ++r) dst[r] = acc[tn][tm][r];
      } else {
        epilogue_store<T>(p, m, n, acc[tn][tm]);
      }
    }
  }
}

// ----------------------------------------------------------------------------- in-kernel split-K reduction
// The K slices of one output tile used to meet in a second launch (k_splitk_reduce*): a kernel boundary plus a
// full re-read of every slab by a grid that starts cold.  Here the `splitk` workgroups of a tile reduce it
// themselves, reduce-scatter style: each writes its fp32 slab WRITE-THROUGH (sc1 stores: no release fence, the
// per-XCD L2s are not coherent), drains, arrives on the tile's counter, waits (one lane, relaxed sc1 poll) until all
// `splitk` slices have arrived, takes ONE agent-scope acquire, and then finishes rows [ks, ks + 1) * BM / splitk of
// the tile: slabs added in slice order with sc1 loads, ((sum + bias) + temb) + residual, one rounding, store,
// per-channel GroupNorm partial sums - the same arithmetic, in the same order, as k_splitk_reduce_stats, so the two
// paths are bit-identical.  The last workgroup to leave zeroes the tile's two words again (they are zero between
// launches; a graph replay needs no memset node).  Recipe: cdna_hip_programming.md section 5 item 2 / Guideline 16
// (counter form).  The host only enables it when tiles * splitk <= the CU count: every workgroup of the launch is
// resident, so the wait cannot deadlock; the spin is bounded all the same.
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <typename T, int BM, int BN, int NTC>
__device__ __forceinline__ void splitk_fused_reduce(const ConvP& p, int m0, int n0, int ks, int tile, int etid, float* sR) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // every storing wave: its write-through slab stores have left
  __syncthreads();
  unsigned* cnt = p.sync + 2 * tile;
  unsigned* flag = reinterpret_cast<unsigned*>(sR);
  if (etid == 0) {
    __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0, ok = 1;
    while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)p.splitk) {
      __builtin_amdgcn_s_sleep(4);
      if (++spins > (1u << 22)) {      // ~1 s: a slice that never arrives must not hang the device
        ok = 0;
        break;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    flag[0] = ok;
  }
  __syncthreads();
  const unsigned ok = flag[0];
  __syncthreads();                                      // flag read by everyone before sR is reused
  if (!ok) {
    // timed out: raise the error word (read + cleared by the host: ops.fused_splitk_error) and still DEPART, so that
    // the tile's two counters return to zero once every slice has left and later launches start clean
    if (etid == 0) {
      __hip_atomic_store(p.sync + 2 * 4096, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned old = __hip_atomic_fetch_add(cnt + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (old == (unsigned)p.splitk - 1) {
        __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(cnt + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    return;
  }
  constexpr int QPR = BN / 4;                           // fp32 quads per tile row
  constexpr int RL = NTC / QPR;                         // row lanes
  const int q = etid % QPR, tr = etid / QPR;
  const bool act = etid < RL * QPR && n0 + 4 * q < p.Cout;
  const int n = n0 + 4 * q;
  const int HW = p.H * p.W;
  const int RW = BM / p.splitk;                         // rows this slice finishes
  const int r0 = ks * RW;
  const int seg = RW < HW ? RW : HW;                    // rows of one statistics segment (inside one sample)
  const T* temb = (const T*)p.temb;
  const T* res = (const T*)p.residual;
  __amdgpu_buffer_rsrc_t rws = __builtin_amdgcn_make_buffer_rsrc((void*)p.ws, 0, (int)((long long)p.splitk * p.M * p.Cout * 4), 0x00020000);
  f32x4 bt = {0.f, 0.f, 0.f, 0.f};
  if (act && p.bias) bt = *reinterpret_cast<const f32x4*>(p.bias + n);
  for (int s0 = 0; s0 < RW; s0 += seg) {
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    if (act) {
      for (int row = s0 + tr; row < s0 + seg; row += RL) {
        const int m = m0 + r0 + row;
        if (m >= p.M) break;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        for (int sl = 0; sl < p.splitk; ++sl) {
          const unsigned off = (unsigned)((((size_t)sl * p.M + m) * p.Cout + n) * 4);
          v += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rws, (int)off, 0, 16));
        }
        if (p.bias) v += bt;
        if (temb) {
          float t0, t1, t2, t3;
          load4<T>(temb + (size_t)(m / HW) * p.temb_stride + n % p.temb_mod, t0, t1, t2, t3);
          v[0] += t0; v[1] += t1; v[2] += t2; v[3] += t3;
        }
        if (res) {
          float a0, a1, a2, a3;
          load4<T>(res + (size_t)m * p.res_ld + n, a0, a1, a2, a3);
          v[0] += a0; v[1] += a1; v[2] += a2; v[3] += a3;
        }
        float vr[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          vr[e] = to_f32(from_f32<T>(v[e]));            // statistics of what the consumer will read
          s1[e] += vr[e];
          s2[e] = fmaf(vr[e], vr[e], s2[e]);
        }
        store4<T>((T*)p.y + (size_t)m * p.y_ld + n, vr[0], vr[1], vr[2], vr[3]);
      }
    }
    if (p.stats_out) {
      // the RL row lanes of a column quad are added in a fixed order through LDS
      if (etid < RL * QPR) {
#pragma unroll
        for (int e = 0; e < 4; ++e) *reinterpret_cast<f32x2*>(sR + ((tr * QPR + q) * 4 + e) * 2) = f32x2{s1[e], s2[e]};
      }
      __syncthreads();
      const int mseg = m0 + r0 + s0;
      if (tr == 0 && act && mseg < p.M) {
        const int b = mseg / HW, sp = (mseg - b * HW) / seg;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float a1 = 0.f, a2 = 0.f;
          for (int r = 0; r < RL; ++r) {
            const f32x2 t = *reinterpret_cast<const f32x2*>(sR + ((r * QPR + q) * 4 + e) * 2);
            a1 += t[0];
            a2 += t[1];
          }
          *reinterpret_cast<f32x2*>(p.stats_out + (((size_t)b * p.stats_S + sp) * p.Cout + n + e) * 2) = f32x2{a1, a2};
        }
      }
      __syncthreads();
    }
  }
  // leave: the last of the tile's workgroups re-zeroes its words (every slice has passed its poll by then)
  if (etid == 0) {
    const unsigned old = __hip_atomic_fetch_add(cnt + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old == (unsigned)p.splitk - 1) {
      __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(cnt + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// ----------------------------------------------------------------------------- v2: LDS-DMA pipeline
// Same tiling / LDS image / epilogue as k_igemm, but the operands go HBM -> LDS directly
// (buffer_load_dwordx4 ... lds: no staging VGPRs, no ds_write pass) through a STAGES-deep ring
// with counted vmcnt waits and raw s_barrier, so STAGES-1 K steps of loads stay in flight across
// barriers (guide T3/T4).  The LDS destination of an LDS-DMA is lane-linear (base + lane*16), so
// the bank-conflict swizzle is applied to the per-lane SOURCE chunk instead (guide rule 21): lane
// l of a 16-row group writes physical chunk (l & 3) of row (l >> 2) and therefore fetches logical
// chunk (l & 3) ^ swz(row).  Zero padding, M / Cout tails and the "past the end" ring slots use
// the buffer descriptor's bounds check: an out-of-range voffset writes zeros to LDS.

// NPROD = 0: every wave both issues its share of the LDS-DMA and computes (v2).
// NPROD > 0: wave specialisation (v3) — the first WGM*WGN waves are CONSUMERS (LDS fragment reads +
// MFMA only), the last NPROD waves are PRODUCERS (LDS-DMA issue + counted vmcnt only).  An LDS-DMA
// wave-instruction costs ~60-185 issue cycles (MI355X_MICROARCH.md); at 10 of them per wave per
// K step that is as long as the 48 MFMAs, and an in-order wave cannot overlap the two.  Split
// across waves that share a SIMD, the DMA issue runs under the other wave's MFMAs.
// R128 (KCH = 2 only): an LDS row holds the whole 128-byte K step of one pixel / cout row, so every
// LDS-DMA wave-instruction fetches 8 full 128-byte lines (8 lanes x 16 B per row) instead of 16
// half lines.  Measured with AFLDM_CONV_DBG (profiles/r01/conv_dma_mfma_decomposition.log): with
// 64-byte pieces the DMA stream ALONE took 93 % of the kernel time at ~13 TB/s of requested bytes
// (each line requested twice, by the kc = 0 and kc = 1 instructions).  Row r keeps source chunk c
// at position c ^ ((r >> 1) & 7): conflict-free for the 16-lane groups of ds_read_b128.
// MINW: waves per SIMD the register allocator must leave room for (the 128x192 tile needs 151 + 96
// registers in its main loop = 2 waves per SIMD = 2 workgroups per CU; left alone the allocator
// spends 30 more on the epilogue and halves the occupancy).
template <typename T, int BM, int BN, int WGM, int WGN, int KCH, int STAGES, int NPROD = 0, bool R128 = false, int MINW = 1>
__global__ void __launch_bounds__((WGM * WGN + NPROD) * 64, MINW) k_igemm2(ConvP p) {
  typedef Mma<T> MM;
  typedef typename MM::Chunk Chunk;
  static_assert(!R128 || KCH == 2, "128-byte rows hold exactly two MFMA K chunks");
  constexpr int NWC = WGM * WGN;                    // consumer (compute) waves
  constexpr int NW = NPROD > 0 ? NPROD : NWC;       // waves that issue the LDS-DMA
  constexpr int EPC = MM::EPC, EPR = 4 * EPC, KSTEP = KCH * EPR;
  constexpr int ESZ = (int)sizeof(T);
  constexpr int WMS = BM / WGM, WNS = BN / WGN, TM = WMS / 16, TN = WNS / 16;
  constexpr int XG = BM / 16, WG_ = BN / 16;          // 16-row groups per plane
  constexpr int XI = KCH * XG, WI = KCH * WG_;         // wave-instructions per stage (X, W)
  static_assert(XI % NW == 0 && WI % NW == 0, "row groups must divide among the waves");
  constexpr int XPW = XI / NW, WPW = WI / NW;          // per wave
  constexpr int LPS = XPW + WPW;                        // LDS-DMA instructions per wave per stage
  constexpr int X_STAGE = KCH * BM * 64, W_STAGE = KCH * BN * 64, STAGE = X_STAGE + W_STAGE;
  constexpr unsigned OOB = 0x80000000u;

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool is_producer = NPROD > 0 ? wave_all >= NWC : true;
  const bool is_consumer = NPROD > 0 ? wave_all < NWC : true;
  const int wave = NPROD > 0 ? (is_producer ? wave_all - NWC : 0) : wave_all;   // index among the issuing waves
  const int cw = NPROD > 0 ? (is_consumer ? wave_all : 0) : wave_all;           // index among the compute waves
  const int wm = cw / WGN, wn = cw % WGN;
  const int li = lane & 15, lg = lane >> 4;

  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  // tile order: consecutive tiles run on one XCD (xcd_remap) and should share their LARGER operand
  // there: the pixel rows when X outweighs W (n fastest), the weight slice when W outweighs X (the
  // 4x4 / 2x2 levels: 10-21 MB of weights against 1-3 MB of pixels; m fastest).  With n fastest at
  // those levels every XCD's L2 fetched the whole weight tensor (PMC: 3.3x the algorithmic bytes).
  const int tiles_m_ = (p.M + BM - 1) / BM;
  const int tile_m = p.m_fast ? tile % tiles_m_ : tile / p.tiles_n;
  const int tile_n = p.m_fast ? tile / tiles_m_ : tile % p.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int Ct = p.C1 + p.C2;
  const int cblocks = Ct / KSTEP;
  const int HW = p.H * p.W;
  const int pad = p.KS >> 1;
  const int ks = blockIdx.z;
  const int kbeg = (int)(((long long)p.ksteps * ks) / p.splitk);
  const int kend = (int)(((long long)p.ksteps * (ks + 1)) / p.splitk);

  // buffer descriptors (wave-uniform by construction: kernel arguments only)
  const long long x1_bytes = (long long)p.M * p.C1 * ESZ, x2_bytes = (long long)p.M * p.C2 * ESZ;
  const long long w_bytes = (long long)p.Cout * p.KS * p.KS * Ct * ESZ;
  __amdgpu_buffer_rsrc_t rx1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.x1, 0, (int)x1_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rx2 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x2 ? p.x2 : p.x1), 0, (int)x2_bytes, 0x00020000);
  // (per-sample weights, afldm_conv_args.w_batch_stride: the tile lies inside sample m0 / HW)
  __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)((const T*)p.w + (size_t)(p.w_bstride ? m0 / HW : 0) * p.w_bstride), 0, (int)w_bytes, 0x00020000);

  // per-lane source coordinates.  Instruction j (0..XI-1) of a stage covers plane kc = j / XG, row
  // group g = j % XG; this wave issues j = wave + NW * i.  Lane l: row 16 g + (l >> 2), source
  // chunk (l & 3) ^ swz(row)  [swz(row) depends only on (l >> 4)].
  // R128: instruction j covers rows 8 j .. 8 j + 7; lane l: row 8 j + (l >> 3), position l & 7 holds
  // source chunk (l & 7) ^ ((row >> 1) & 7) = (l & 7) ^ ((4 j + (l >> 4)) & 7).
  const int lrow = R128 ? lane >> 3 : lane >> 2;
  const int lchunk = (lane & 3) ^ ((4 - (lane >> 4)) & 3);
  int xoh[XPW], xow[XPW], xcin[XPW];   // xcin: element offset of this lane's chunk inside the K step
  unsigned xbase[XPW];  // byte offset of pixel (b, oh, ow) channel 0 in a tensor with C channels = 1 (scaled later)
  bool xok[XPW];
#pragma unroll
  for (int i = 0; i < XPW; ++i) {
    const int j = wave + NW * i;
    const int g = j % XG;
    xcin[i] = R128 ? (((lane & 7) ^ ((4 * j + (lane >> 4)) & 7)) * EPC) : ((j / XG) * EPR + lchunk * EPC);
    const int m = R128 ? m0 + 8 * j + lrow : m0 + 16 * g + lrow;
    xok[i] = m < p.M;
    const int mm = xok[i] ? m : 0;
    const int b = mm / HW, pix = mm - b * HW;
    xoh[i] = pix / p.W;
    xow[i] = pix - xoh[i] * p.W;
    xbase[i] = (unsigned)mm;  // pixel index; multiplied by the channel count of the source tensor at issue time
  }
  unsigned wrow[WPW];
  int wcin[WPW];
  bool wok[WPW];
#pragma unroll
  for (int i = 0; i < WPW; ++i) {
    const int j = wave + NW * i;
    const int g = j % WG_;
    wcin[i] = R128 ? (((lane & 7) ^ ((4 * j + (lane >> 4)) & 7)) * EPC) : ((j / WG_) * EPR + lchunk * EPC);
    const int n = R128 ? n0 + 8 * j + lrow : n0 + 16 * g + lrow;
    wok[i] = n < p.Cout;
    wrow[i] = (unsigned)(wok[i] ? n : 0) * (unsigned)(p.KS * p.KS);
  }

  // K cursor of the NEXT step to issue (wave-uniform scalars, advanced incrementally: no division
  // in the loop).  Two K orders (p.tap_inner):
  //   0: tap outer, channel block inner.  Per-lane voffsets are recomputed only when the filter tap
  //      changes; inside a tap the channel block advances through the SGPR soffset of the buffer
  //      load, so a K step costs no VALU address arithmetic.
  //   1: channel block outer, tap inner: the KS*KS taps of one 128-byte channel block re-read the
  //      same pixels shifted by a row / column (L1 / L2 hits; the DMA-only time of the 32x32 level
  //      drops 126 -> 81-98 us) at the price of a retap per K step.  Measured a wash at 32x32 and a
  //      loss at 16x16 / 8x8 where the tile's pixels are L2-resident anyway
  //      (profiles/r01/conv_dma_mfma_decomposition.log), so the planner keeps order 0.
  const int taps = p.KS * p.KS;
  const bool tap_inner = p.tap_inner && taps > 1;
  int is_kt = kbeg;
  int is_tap = tap_inner ? kbeg % taps : kbeg / cblocks;
  int is_ci0 = tap_inner ? (kbeg / taps) * KSTEP : (kbeg - is_tap * cblocks) * KSTEP;
  int is_kh = is_tap / p.KS, is_kw = is_tap - is_kh * p.KS;
  unsigned xoff1[XPW], xoff2[XPW];   // byte voffset of (pixel + tap shift, channel kc*EPR + chunk) in x1 / x2, or OOB
  unsigned woff[WPW];                 // byte voffset of (cout row, tap 0, channel kc*EPR + chunk), or OOB
#pragma unroll
  for (int i = 0; i < WPW; ++i)
    woff[i] = wok[i] ? (wrow[i] * (unsigned)Ct + (unsigned)wcin[i]) * ESZ : OOB;
  auto retap = [&]() {
    const int dh = is_kh - pad, dw = is_kw - pad;
    const int dpix = dh * p.W + dw;
#pragma unroll
    for (int i = 0; i < XPW; ++i) {
      const int ih = xoh[i] + dh, iw = xow[i] + dw;
      const bool ok = xok[i] && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
      const unsigned pixel = (unsigned)((int)xbase[i] + dpix);
      const unsigned cin = (unsigned)xcin[i];
      xoff1[i] = ok ? (pixel * (unsigned)p.C1 + cin) * ESZ : OOB;
      xoff2[i] = ok ? (pixel * (unsigned)p.C2 + cin) * ESZ : OOB;
    }
  };
  retap();

  auto issue = [&](int slot) {
    char* sbase = smem + slot * STAGE;
    const bool live = is_kt < kend;
    const bool second = is_ci0 >= p.C1;
    const unsigned xs = live ? (unsigned)((second ? is_ci0 - p.C1 : is_ci0) * ESZ) : OOB;      // SGPR offsets
    const unsigned wsoff = live ? (unsigned)((is_tap * Ct + is_ci0) * ESZ) : OOB;
#pragma unroll
    for (int i = 0; i < XPW; ++i) {
      const int j = wave + NW * i;
      lds_ptr_t dst = (lds_ptr_t)(sbase + j * 1024);
      // (explicit int casts: hipcc 7.2 silently drops the whole kernel template when an element of a
      //  captured unsigned array is passed straight to this builtin)
      if (second) __builtin_amdgcn_raw_ptr_buffer_load_lds(rx2, dst, 16, (int)xoff2[i], (int)xs, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(rx1, dst, 16, (int)xoff1[i], (int)xs, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < WPW; ++i) {
      const int j = wave + NW * i;
      lds_ptr_t dst = (lds_ptr_t)(sbase + X_STAGE + j * 1024);
      if (p.w_nt) __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, dst, 16, (int)woff[i], (int)wsoff, 0, 2);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, dst, 16, (int)woff[i], (int)wsoff, 0, 0);
    }
    // advance the cursor
    ++is_kt;
    if (!tap_inner) {
      is_ci0 += KSTEP;
      if (is_ci0 >= Ct) {
        is_ci0 = 0;
        ++is_tap;
        if (++is_kw == p.KS) {
          is_kw = 0;
          ++is_kh;
        }
        if (is_kt < kend) retap();
      }
    } else {
      ++is_tap;
      if (++is_kw == p.KS) {
        is_kw = 0;
        ++is_kh;
      }
      if (is_tap == taps) {
        is_tap = 0;
        is_kh = 0;
        is_ci0 += KSTEP;
      }
      if (is_kt < kend) retap();
    }
  };

  f32x4 acc[TN][TM];
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto compute = [&](int slot) {
    const char* sX = smem + slot * STAGE;
    const char* sW = sX + X_STAGE;
#pragma unroll
    for (int kc = 0; kc < KCH; ++kc) {
      Chunk a[TN], b[TM];
#pragma unroll
      for (int t = 0; t < TN; ++t) {
        const int row = wn * WNS + t * 16 + li;
        a[t] = R128 ? ld16<Chunk>(sW + row * 128 + (((kc * 4 + lg) ^ ((row >> 1) & 7)) << 4))
                    : ld16<Chunk>(sW + (kc * BN + row) * 64 + ((lg ^ swz(row)) << 4));
      }
#pragma unroll
      for (int t = 0; t < TM; ++t) {
        const int row = wm * WMS + t * 16 + li;
        b[t] = R128 ? ld16<Chunk>(sX + row * 128 + (((kc * 4 + lg) ^ ((row >> 1) & 7)) << 4))
                    : ld16<Chunk>(sX + (kc * BM + row) * 64 + ((lg ^ swz(row)) << 4));
      }
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) MM::mma(acc[tn][tm], a[tn], b[tm]);
      // two compute waves per SIMD (8 consumers): the partner wave covers this wave's LDS latency,
      // so the fragments of the next K chunk are NOT prefetched (saves 40 registers: no spills at
      // the 170-register budget of 12 waves per CU)
      if constexpr (NPROD > 0 && NWC >= 8) __builtin_amdgcn_sched_barrier(0);
    }
  };

  // Protocol per K step kt (all waves meet at ONE s_barrier):
  //   issuing waves: wait until their LDS-DMA of step kt has landed (counted vmcnt, younger stages
  //                  stay in flight) -> barrier -> refill the slot of step kt-1 with step kt+STAGES-1
  //   compute waves: barrier -> fragments + MFMA of step kt
  // With NPROD > 0 the two roles run separate loops (disjoint register live ranges) that meet at
  // the same barrier count; with NPROD == 0 every wave does both.
  if (NPROD > 0) {
    if (is_producer) {
#pragma unroll
      for (int s = 0; s < STAGES - 1; ++s) issue(s);
      int slot = 0;
      for (int kt = kbeg; kt < kend; ++kt) {
        wait_vmcnt<(STAGES - 2) * LPS>();
        __builtin_amdgcn_s_barrier();
        int nslot = slot + STAGES - 1;
        if (nslot >= STAGES) nslot -= STAGES;
        if (!(p.dbg & 1)) issue(nslot);
        slot = slot + 1 == STAGES ? 0 : slot + 1;
      }
      wait_vmcnt<0>();  // drain the zero-fill tail before the workgroup's LDS can be re-assigned
      return;
    }
    int slot = 0;
    for (int kt = kbeg; kt < kend; ++kt) {
      __builtin_amdgcn_s_barrier();
      if (!(p.dbg & 2)) compute(slot);
      slot = slot + 1 == STAGES ? 0 : slot + 1;
    }
  } else {
    // prologue: STAGES-1 K steps in flight (slots past kend are filled with zeros: uniform counts)
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s) issue(s);
    int slot = 0;
    for (int kt = kbeg; kt < kend; ++kt) {
      wait_vmcnt<(STAGES - 2) * LPS>();
      __builtin_amdgcn_s_barrier();
      int nslot = slot + STAGES - 1;
      if (nslot >= STAGES) nslot -= STAGES;
      if (!(p.dbg & 1)) issue(nslot);
      if (!(p.dbg & 2)) compute(slot);
      slot = slot + 1 == STAGES ? 0 : slot + 1;
    }
    wait_vmcnt<0>();  // drain the zero-fill tail before the workgroup's LDS can be re-assigned
  }

  // ---- LDS-staged epilogue.  An accumulator lane holds 4 consecutive couts of ONE pixel, so direct
  // stores are 8-byte pieces of 16 different rows per instruction (and 2-byte scatters for the
  // channel-major V^T output): the short-K 1x1 GEMMs (q/k/v, to_out at 32x32) ran at ~1.5 TB/s of
  // output.  Instead the fp32 tile goes through the (now idle) pipeline buffers and leaves as full
  // rows: 16 B per lane, bias / temb / residual added in fp32 on the way (same order as
  // epilogue_store, so the result is bit-identical to the direct and split-K paths).
  {
    constexpr int SROW = BN + 8;                         // fp32 row stride: 8 mod 64 banks -> conflict-free 16-byte writes
    constexpr int LDS_TOTAL = STAGES * STAGE;
    constexpr int TMP = (WGM * TM * 16 * SROW * 4 <= LDS_TOTAL) ? TM
                      : (WGM * (TM / 2) * 16 * SROW * 4 <= LDS_TOTAL && TM % 2 == 0) ? TM / 2
                      : (WGM * (TM / 4) * 16 * SROW * 4 <= LDS_TOTAL && TM % 4 == 0) ? TM / 4 : 0;
    static_assert(TMP > 0, "staging tile does not fit the pipeline buffers");
    constexpr int PASSES = TM / TMP, PROWS = WGM * TMP * 16;   // rows staged per pass
    constexpr int NTC = NWC * 64;                        // threads in the epilogue (producer waves have exited)
    constexpr int EO = 16 / ESZ;                         // output elements per 16-byte store
    float* sC = reinterpret_cast<float*>(smem);
    const T* temb = (const T*)p.temb;
    const T* res = (const T*)p.residual;
    const int etid = cw * 64 + lane;
    constexpr int CPR = BN / EO;                         // 16-byte chunks per staged row
    constexpr int RPI = NTC / CPR;                       // rows the workgroup covers per sweep
    float ss1[EO], ss2[EO];
#pragma unroll
    for (int e = 0; e < EO; ++e) ss1[e] = ss2[e] = 0.f;
    // A tile that lies wholly in the channel-major output (the V^T third of a fused q|k|v GEMM,
    // out_mode 1) is staged TRANSPOSED ([cout][pixel]): the copy-out then reads 16-byte runs of
    // consecutive pixels instead of gathering 8 strided floats per store.
    constexpr int TROW = PROWS + 4;                      // transposed row stride (floats)
    constexpr bool T_FITS = BN * TROW * 4 <= LDS_TOTAL;
    const int cm_beg0 = p.out_mode == 1 ? 0 : (p.y2 ? p.split_n : p.Cout);
    const bool cm_tile = T_FITS && p.splitk == 1 && p.stage_ok && n0 >= cm_beg0 && !p.residual && (HW % EO) == 0;
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
      __syncthreads();   // pipeline buffers idle (first pass) / previous pass copied out
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int t = 0; t < TMP; ++t) {
          const int row = (wm * TMP + t) * 16 + li;      // staged row  <->  tile row wm*WMS + (ps*TMP + t)*16 + li
          if (cm_tile) {
#pragma unroll
            for (int r = 0; r < 4; ++r) sC[(wn * WNS + tn * 16 + 4 * lg + r) * TROW + row] = acc[tn][ps * TMP + t][r];
          } else {
            *reinterpret_cast<f32x4*>(sC + row * SROW + wn * WNS + tn * 16 + 4 * lg) = acc[tn][ps * TMP + t];
          }
          __builtin_amdgcn_sched_barrier(0);   // one tile at a time: keeps the accumulator -> VGPR copies from piling up
        }
      __syncthreads();
      if (cm_tile) {
        T* yc = p.out_mode == 1 ? (T*)p.y : (T*)p.y2;
        const int cch = p.Cout - cm_beg0;
        constexpr int RG = PROWS / EO;                   // 16-byte pixel runs per cout per pass
#pragma unroll 1
        for (int i = etid; i < BN * RG; i += NTC) {
          const int nl = i / RG, g = i - nl * RG;
          const int n = n0 + nl;
          if (n >= p.Cout) continue;
          const int row0 = g * EO;
          const int mt = row0 / (TMP * 16), rr = row0 - mt * (TMP * 16);
          const int m = m0 + mt * WMS + ps * TMP * 16 + rr;
          if (m >= p.M) continue;
          const int b = m / HW, pix = m - b * HW;
          float add = p.bias ? p.bias[n] : 0.f;
          const float tsv = temb ? to_f32(temb[(size_t)b * p.temb_stride + n % p.temb_mod]) : 0.f;
          Chunk o;
#pragma unroll
          for (int q = 0; q < EO / 4; ++q) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(sC + nl * TROW + row0 + 4 * q);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float sv = a[e] + add;
              if (temb) sv += tsv;
              o[4 * q + e] = from_f32<T>(sv);
            }
          }
          if (m + EO <= p.M) {
            st16<Chunk>(yc + ((size_t)b * cch + (n - cm_beg0)) * HW + pix, o);
          } else {
            for (int e = 0; e < EO && m + e < p.M; ++e) yc[((size_t)b * cch + (n - cm_beg0)) * HW + pix + e] = o[e];
          }
        }
        continue;
      }
      if (p.splitk > 1) {
        // split-K slab: fp32 rows, 16 bytes per lane
        constexpr int QPR = BN / 4;
        const bool v4 = (p.Cout & 3) == 0;
#pragma unroll 1
        for (int i = etid; i < PROWS * QPR; i += NTC) {
          const int row = i / QPR, q = i - row * QPR;
          const int n = n0 + 4 * q;
          const int mt = row / (TMP * 16), rr = row - mt * (TMP * 16);
          const int m = m0 + mt * WMS + ps * TMP * 16 + rr;
          if (m >= p.M || n >= p.Cout) continue;
          const f32x4 a = *reinterpret_cast<const f32x4*>(sC + row * SROW + 4 * q);
          float* dst = p.ws + ((size_t)ks * p.M + m) * p.Cout + n;
          if (p.sync) {     // in-kernel reduction: write-through (sc1) so that the other slices' workgroups can read it
            __amdgpu_buffer_rsrc_t rws = __builtin_amdgcn_make_buffer_rsrc((void*)p.ws, 0, (int)((long long)p.splitk * p.M * p.Cout * 4), 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, a), rws, (int)((((size_t)ks * p.M + m) * p.Cout + n) * 4), 0, 16);
          } else if (v4 && n + 3 < p.Cout) *reinterpret_cast<f32x4*>(dst) = a;
          else
            for (int r = 0; r < 4 && n + r < p.Cout; ++r) dst[r] = a[r];
        }
        continue;
      }
      // (a) NHWC part: couts [n0, min(n0 + BN, nhwc_end)), 16 bytes of one pixel per lane.  A thread
      // keeps ONE 16-byte column of the tile and walks down its rows, so the per-channel
      // GroupNorm sums of the stored values (stats_out) accumulate in its registers.
      const int nhwc_end = p.out_mode == 1 ? 0 : (p.y2 ? p.split_n : p.Cout);
      if (etid < RPI * CPR) {
        const int ch = etid % CPR;
        const int n = n0 + ch * EO;
        const bool fast = p.stage_ok && n + EO <= nhwc_end;
        // the thread's column is fixed: bias (and the time embedding, when the tile lies inside one
        // sample) are fetched ONCE, not per row - a global-load latency per row was most of the
        // epilogue of the short-K 1x1 GEMMs
        const bool one_sample = (HW % BM) == 0;
        float bvec[EO], tvec[EO];
#pragma unroll
        for (int e = 0; e < EO; ++e) bvec[e] = tvec[e] = 0.f;
        if (fast && p.bias) {
#pragma unroll
          for (int q = 0; q < EO / 4; ++q) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + n + 4 * q);
            bvec[4 * q] = bv[0]; bvec[4 * q + 1] = bv[1]; bvec[4 * q + 2] = bv[2]; bvec[4 * q + 3] = bv[3];
          }
        }
        const int b_tile = m0 / HW;
        if (fast && temb && one_sample) {
          const Chunk tv = ld16<Chunk>(temb + (size_t)b_tile * p.temb_stride + n % p.temb_mod);
#pragma unroll
          for (int e = 0; e < EO; ++e) tvec[e] = to_f32(tv[e]);
        }
        // rows of a thread: every RPI-th one; CONSECUTIVE ones when the tile spans several whole samples and
        // statistics are wanted (p.stats_multi), so that all rows of a thread lie in one sample
        constexpr int RPT = PROWS / RPI;
        const int row_first = p.stats_multi == 1 ? (etid / CPR) * RPT : etid / CPR;
        const int row_step = p.stats_multi == 1 ? 1 : RPI;
        const int row_end = p.stats_multi == 1 ? row_first + RPT : PROWS;
#pragma unroll 1
        for (int row = row_first; row < row_end; row += row_step) {
          const int mt = row / (TMP * 16), rr = row - mt * (TMP * 16);
          const int m = m0 + mt * WMS + ps * TMP * 16 + rr;
          if (m >= p.M || n >= nhwc_end) continue;
          float v[EO];
#pragma unroll
          for (int q = 0; q < EO / 4; ++q) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(sC + row * SROW + ch * EO + 4 * q);
            v[4 * q] = a[0]; v[4 * q + 1] = a[1]; v[4 * q + 2] = a[2]; v[4 * q + 3] = a[3];
          }
          if (fast) {
            if (p.bias) {
#pragma unroll
              for (int e = 0; e < EO; ++e) v[e] += bvec[e];
            }
            if (temb) {
              if (one_sample) {
#pragma unroll
                for (int e = 0; e < EO; ++e) v[e] += tvec[e];
              } else {
                const Chunk tv = ld16<Chunk>(temb + (size_t)(m / HW) * p.temb_stride + n % p.temb_mod);
#pragma unroll
                for (int e = 0; e < EO; ++e) v[e] += to_f32(tv[e]);
              }
            }
            if (res) {
              const Chunk rv = ld16<Chunk>(res + (size_t)m * p.res_ld + n);
#pragma unroll
              for (int e = 0; e < EO; ++e) v[e] += to_f32(rv[e]);
            }
            Chunk o;
#pragma unroll
            for (int e = 0; e < EO; ++e) o[e] = from_f32<T>(v[e]);
            st16_out<Chunk>((T*)p.y + (size_t)m * p.y_ld + n, o);
            if (p.stats_out && p.stats_multi == 2) {
              // H * W == 1 (a dense layer over a flattened plane): every row is a sample of its own, its "sums" are the
              // values themselves (S = 1)
#pragma unroll
              for (int e = 0; e < EO; e += 2) {
                const float v0 = to_f32(o[e]), v1 = to_f32(o[e + 1]);
                *reinterpret_cast<f32x4*>(p.stats_out + ((size_t)m * p.Cout + n + e) * 2) = f32x4{v0, v0 * v0, v1, v1 * v1};
              }
            } else if (p.stats_out) {
#pragma unroll
              for (int e = 0; e < EO; ++e) {
                const float vr = to_f32(o[e]);      // statistics of what the consumer will read
                ss1[e] += vr;
                ss2[e] = fmaf(vr, vr, ss2[e]);
              }
            }
          } else {   // ragged cout tail / odd leading dimensions
            const int b = m / HW;
            for (int e = 0; e < EO && n + e < nhwc_end; ++e) {
              float sv = v[e];
              if (p.bias) sv += p.bias[n + e];
              if (temb) sv += to_f32(temb[(size_t)b * p.temb_stride + (n + e) % p.temb_mod]);
              if (res) sv += to_f32(res[(size_t)m * p.res_ld + n + e]);
              ((T*)p.y)[(size_t)m * p.y_ld + n + e] = from_f32<T>(sv);
            }
          }
        }
      }
      // (b) channel-major part (out_mode 1, or the V^T third of a fused QKV GEMM): EO consecutive
      // pixels of one cout per lane; lanes = 16 couts x 4 pixel groups (64-byte runs per cout row)
      const int cm_beg = p.out_mode == 1 ? 0 : (p.y2 ? p.split_n : p.Cout);
      if (n0 + BN > cm_beg && cm_beg < p.Cout) {
        T* yc = p.out_mode == 1 ? (T*)p.y : (T*)p.y2;
        const int cch = p.Cout - cm_beg;                 // channels of the channel-major tensor
        constexpr int RG = PROWS / EO;                   // pixel groups per pass
#pragma unroll 1
        for (int i = etid; i < BN * RG; i += NTC) {
          const int nl = (i & 15) + 16 * (i / (16 * RG)), g = (i >> 4) % RG;
          const int n = n0 + nl;
          if (n < cm_beg || n >= p.Cout) continue;
          const int row0 = g * EO;
          const int mt = row0 / (TMP * 16), rr = row0 - mt * (TMP * 16);
          const int m = m0 + mt * WMS + ps * TMP * 16 + rr;
          if (m >= p.M) continue;
          const int b = m / HW, pix = m - b * HW;
          const float bsv = p.bias ? p.bias[n] : 0.f;
          const float tsv = temb ? to_f32(temb[(size_t)b * p.temb_stride + n % p.temb_mod]) : 0.f;
          Chunk o;
#pragma unroll
          for (int e = 0; e < EO; ++e) {
            float sv = sC[(row0 + e) * SROW + nl] + bsv;
            if (temb && m + e < p.M) sv += p.stage_ok ? tsv : to_f32(temb[(size_t)((m + e) / HW) * p.temb_stride + n % p.temb_mod]);
            if (res && m + e < p.M) sv += to_f32(res[(size_t)(m + e) * p.res_ld + n]);
            o[e] = from_f32<T>(sv);
          }
          if (p.stage_ok && m + EO <= p.M) {
            st16<Chunk>(yc + ((size_t)b * cch + (n - cm_beg)) * HW + pix, o);
          } else {
            for (int e = 0; e < EO && m + e < p.M; ++e) {   // (odd H*W: the run may cross into the next sample)
              const int me = m + e, be = me / HW;
              yc[((size_t)be * cch + (n - cm_beg)) * HW + (me - be * HW)] = o[e];
            }
          }
        }
      }
    }
    if (p.splitk > 1) {
      if (p.sync) splitk_fused_reduce<T, BM, BN, NTC>(p, m0, n0, ks, tile, etid, sC);
      return;
    }
    if (p.stats_out && p.stats_multi != 2) {
      // per-channel sums of this tile (BM rows of ONE sample: the host only asks when H*W % BM == 0):
      // the RPI row-interleaved partials of a column are added in a fixed order through LDS
      float* sR = sC;   // [RPI][BN][2]
      __syncthreads();
      if (etid < RPI * CPR) {
        const int ch = etid % CPR, tr = etid / CPR;
#pragma unroll
        for (int e = 0; e < EO; ++e)
          *reinterpret_cast<f32x2*>(sR + ((tr * BN) + ch * EO + e) * 2) = f32x2{ss1[e], ss2[e]};
      }
      __syncthreads();
      if (p.stats_multi == 1) {
        // the tile holds BM / HW whole samples (S = 1): row lane tr owns rows tr * RPT .. + RPT - 1
        constexpr int RPT = PROWS / RPI;
        const int ns = BM / HW, lanes_per_sample = HW / RPT;
        for (int idx = etid; idx < BN * ns; idx += NTC) {
          const int c = idx % BN, sl = idx / BN;
          const int b = m0 / HW + sl;
          if (n0 + c >= p.Cout || (size_t)b * HW >= (size_t)p.M) continue;
          float a1 = 0.f, a2 = 0.f;
          for (int tr = sl * lanes_per_sample; tr < (sl + 1) * lanes_per_sample; ++tr) {
            const f32x2 v = *reinterpret_cast<const f32x2*>(sR + ((tr * BN) + c) * 2);
            a1 += v[0];
            a2 += v[1];
          }
          *reinterpret_cast<f32x2*>(p.stats_out + ((size_t)b * p.Cout + n0 + c) * 2) = f32x2{a1, a2};
        }
      } else
      for (int c = etid; c < BN; c += NTC) {
        if (n0 + c >= p.Cout) continue;
        float a1 = 0.f, a2 = 0.f;
        for (int tr = 0; tr < RPI; ++tr) {
          const f32x2 v = *reinterpret_cast<const f32x2*>(sR + ((tr * BN) + c) * 2);
          a1 += v[0];
          a2 += v[1];
        }
        const int b = m0 / HW, sp = (m0 - b * HW) / BM;
        *reinterpret_cast<f32x2*>(p.stats_out + (((size_t)b * p.stats_S + sp) * p.Cout + n0 + c) * 2) = f32x2{a1, a2};
      }
    }
  }
}

// ----------------------------------------------------------------------------- v5: persistent tiles
// 128x192 tiles, 4 MFMA-only consumer waves + 4 LDS-DMA producer waves (as variant 33), but a workgroup
// walks SEVERAL tiles and the producers never stop at a tile boundary: while the consumers run the epilogue
// of tile i the ring already holds the first STAGES-1 K steps of tile i+1, and the epilogue's stores drain
// under the next tile's MFMAs.  (With one tile per workgroup slot nothing overlaps the fixed part: 14 of
// the 52 us of a 192->192 3x3 conv at 32x32 remain with both the DMA and the MFMA phase switched off.)
// The epilogue is wave-private - a consumer wave owns 64 rows x 96 couts, adds bias / time embedding /
// residual in fp32 in registers, rounds, stages 32 rows at a time through its own LDS patch and stores
// 192-byte row pieces; GroupNorm partial sums are written per (32-row pass) split, so no workgroup
// barrier exists outside the K loop and the producers may run ahead.  bf16, H*W % 128 == 0, Cout % 192 == 0,
// token-major output, no split-K.  Same K order and rounding as k_igemm2: bit-identical results.
template <typename T, int STAGES>
__global__ void __launch_bounds__(512, 1) k_igemm3(ConvP p) {
  typedef Mma<T> MM;
  typedef typename MM::Chunk Chunk;
  constexpr int BM = 128, BN = 192, WGN = 2, NWC = 4, NW = 4;
  constexpr int EPC = MM::EPC, KSTEP = 8 * EPC, ESZ = (int)sizeof(T);
  constexpr int WMS = 64, WNS = 96, TM = 4, TN = 6;
  constexpr int XI = BM / 8, WI = BN / 8, XPW = XI / NW, WPW = WI / NW, LPS = XPW + WPW;
  constexpr int X_STAGE = BM * 128, W_STAGE = BN * 128, STAGE = X_STAGE + W_STAGE;
  constexpr int SROW = WNS + 8, WSTG = 32 * SROW * ESZ;
  constexpr unsigned OOB = 0x80000000u;
  static_assert(sizeof(T) == 2, "bf16 only");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool is_producer = wave_all >= NWC;
  const int li = lane & 15, lg = lane >> 4;
  const int G = gridDim.x;
  const int tiles_m_ = p.M / BM, ntiles = tiles_m_ * p.tiles_n;
  const int first = xcd_remap(blockIdx.x, G);
  const int nmine = first < ntiles ? (ntiles - first + G - 1) / G : 0;
  const int Ct = p.C1 + p.C2, cblocks = Ct / KSTEP, HW = p.H * p.W, pad = p.KS >> 1;
  const int ksteps = p.ksteps;
  auto tile_origin = [&](int ti, int& m0, int& n0) {
    const int tile = first + ti * G;
    const int tm = p.m_fast ? tile % tiles_m_ : tile / p.tiles_n;
    const int tn = p.m_fast ? tile / tiles_m_ : tile % p.tiles_n;
    m0 = tm * BM;
    n0 = tn * BN;
  };

  if (is_producer) {
    const int wave = wave_all - NWC;
    const long long x1_bytes = (long long)p.M * p.C1 * ESZ, x2_bytes = (long long)p.M * p.C2 * ESZ;
    const long long w_bytes = (long long)p.Cout * p.KS * p.KS * Ct * ESZ;
    __amdgpu_buffer_rsrc_t rx1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.x1, 0, (int)x1_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rx2 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x2 ? p.x2 : p.x1), 0, (int)x2_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)w_bytes, 0x00020000);
    const int lrow = lane >> 3;
    int xoh[XPW], xow[XPW], xcin[XPW];
    unsigned xbase[XPW];
    unsigned xoff1[XPW], xoff2[XPW], woff[WPW];
    int cur = 0;                       // tile of the NEXT step to issue
    int is_kt = 0, is_tap = 0, is_ci0 = 0, is_kh = 0, is_kw = 0;
    auto retap = [&]() {
      const int dh = is_kh - pad, dw = is_kw - pad;
      const int dpix = dh * p.W + dw;
#pragma unroll
      for (int i = 0; i < XPW; ++i) {
        const int ih = xoh[i] + dh, iw = xow[i] + dw;
        const bool ok = (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
        const unsigned pixel = (unsigned)((int)xbase[i] + dpix);
        const unsigned cin = (unsigned)xcin[i];
        xoff1[i] = ok ? (pixel * (unsigned)p.C1 + cin) * ESZ : OOB;
        xoff2[i] = ok ? (pixel * (unsigned)p.C2 + cin) * ESZ : OOB;
      }
    };
    auto setup_tile = [&]() {
      int m0, n0;
      tile_origin(cur, m0, n0);
#pragma unroll
      for (int i = 0; i < XPW; ++i) {
        const int j = wave + NW * i;
        xcin[i] = ((lane & 7) ^ ((4 * j + (lane >> 4)) & 7)) * EPC;
        const int m = m0 + 8 * j + lrow;
        const int b = m / HW, pix = m - b * HW;
        xoh[i] = pix / p.W;
        xow[i] = pix - xoh[i] * p.W;
        xbase[i] = (unsigned)m;
      }
#pragma unroll
      for (int i = 0; i < WPW; ++i) {
        const int j = wave + NW * i;
        const int wc = ((lane & 7) ^ ((4 * j + (lane >> 4)) & 7)) * EPC;
        const int n = n0 + 8 * j + lrow;
        woff[i] = ((unsigned)n * (unsigned)(p.KS * p.KS) * (unsigned)Ct + (unsigned)wc) * ESZ;
      }
      is_kt = is_tap = is_ci0 = is_kh = is_kw = 0;
      retap();
    };
    if (nmine > 0) setup_tile();
    auto issue = [&](int slot) {
      char* sbase = smem + slot * STAGE;
      const bool live = cur < nmine;
      const bool second = is_ci0 >= p.C1;
      const unsigned xs = live ? (unsigned)((second ? is_ci0 - p.C1 : is_ci0) * ESZ) : OOB;
      const unsigned wsoff = live ? (unsigned)((is_tap * Ct + is_ci0) * ESZ) : OOB;
#pragma unroll
      for (int i = 0; i < XPW; ++i) {
        const int j = wave + NW * i;
        lds_ptr_t dst = (lds_ptr_t)(sbase + j * 1024);
        if (second) __builtin_amdgcn_raw_ptr_buffer_load_lds(rx2, dst, 16, (int)xoff2[i], (int)xs, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rx1, dst, 16, (int)xoff1[i], (int)xs, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < WPW; ++i) {
        const int j = wave + NW * i;
        lds_ptr_t dst = (lds_ptr_t)(sbase + X_STAGE + j * 1024);
        if (p.w_nt) __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, dst, 16, (int)woff[i], (int)wsoff, 0, 2);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, dst, 16, (int)woff[i], (int)wsoff, 0, 0);
      }
      if (!live) return;
      ++is_kt;
      is_ci0 += KSTEP;
      if (is_ci0 >= Ct) {
        is_ci0 = 0;
        ++is_tap;
        if (++is_kw == p.KS) {
          is_kw = 0;
          ++is_kh;
        }
        if (is_kt < ksteps) retap();
      }
      if (is_kt == ksteps) {          // next tile: its first K steps follow immediately (no drain)
        ++cur;
        if (cur < nmine) setup_tile();
      }
    };
#pragma unroll
    for (int q = 0; q < STAGES - 1; ++q) issue(q);
    int slot = 0;
    const int total = nmine * ksteps;
    for (int g = 0; g < total; ++g) {
      wait_vmcnt<(STAGES - 2) * LPS>();
      __builtin_amdgcn_s_barrier();
      int nslot = slot + STAGES - 1;
      if (nslot >= STAGES) nslot -= STAGES;
      issue(nslot);
      slot = slot + 1 == STAGES ? 0 : slot + 1;
    }
    wait_vmcnt<0>();
    return;
  }

  // ---------------- consumers
  const int cw = wave_all, wm = cw / WGN, wn = cw % WGN;
  T* ws = reinterpret_cast<T*>(smem + STAGES * STAGE + cw * WSTG);
  const T* temb = (const T*)p.temb;
  const T* res = (const T*)p.residual;
  int slot = 0;
  for (int ti = 0; ti < nmine; ++ti) {
    int m0, n0;
    tile_origin(ti, m0, n0);
    f32x4 acc[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
      for (int b = 0; b < TM; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int kt = 0; kt < ksteps; ++kt) {
      __builtin_amdgcn_s_barrier();
      const char* sX = smem + slot * STAGE;
      const char* sW = sX + X_STAGE;
#pragma unroll
      for (int kc = 0; kc < 2; ++kc) {
        Chunk a[TN], b[TM];
#pragma unroll
        for (int t = 0; t < TN; ++t) {
          const int row = wn * WNS + t * 16 + li;
          a[t] = ld16<Chunk>(sW + row * 128 + (((kc * 4 + lg) ^ ((row >> 1) & 7)) << 4));
        }
#pragma unroll
        for (int t = 0; t < TM; ++t) {
          const int row = wm * WMS + t * 16 + li;
          b[t] = ld16<Chunk>(sX + row * 128 + (((kc * 4 + lg) ^ ((row >> 1) & 7)) << 4));
        }
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
          for (int tm = 0; tm < TM; ++tm) MM::mma(acc[tn][tm], a[tn], b[tm]);
      }
      slot = slot + 1 == STAGES ? 0 : slot + 1;
    }
    // ---- wave-private epilogue: 2 passes of 32 rows
    const int bsm = m0 / HW;
    const int split0 = ((m0 - bsm * HW) / BM) * 4 + wm * 2;
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        const int n = n0 + wn * WNS + 16 * tn + 4 * lg;
        f32x4 bv = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 tv = f32x4{0.f, 0.f, 0.f, 0.f};
        if (temb) {
          const bf16x4 t4 = *reinterpret_cast<const bf16x4*>(temb + (size_t)bsm * p.temb_stride + n % p.temb_mod);
#pragma unroll
          for (int r = 0; r < 4; ++r) tv[r] = (float)t4[r];
        }
        float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tmi = 0; tmi < 2; ++tmi) {
          const int tm = 2 * ps + tmi;
          const int m = m0 + wm * WMS + 16 * tm + li;
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            v[r] = acc[tn][tm][r] + bv[r];
            if (temb) v[r] += tv[r];
          }
          if (res) {
            const bf16x4 rr = *reinterpret_cast<const bf16x4*>(res + (size_t)m * p.res_ld + n);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += (float)rr[r];
          }
          bf16x4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            o[r] = (bf16)v[r];
            const float q = (float)o[r];
            s1[r] += q;
            s2[r] = fmaf(q, q, s2[r]);
          }
          *reinterpret_cast<bf16x4*>(ws + (16 * tmi + li) * SROW + 16 * tn + 4 * lg) = o;
        }
        if (p.stats_out) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) {
              s1[r] += __shfl_xor(s1[r], o, 64);
              s2[r] += __shfl_xor(s2[r], o, 64);
            }
          }
          if (li == 0) {
            float* so = p.stats_out + (((size_t)bsm * p.stats_S + split0 + ps) * p.Cout + n) * 2;
#pragma unroll
            for (int r = 0; r < 4; ++r) *reinterpret_cast<f32x2*>(so + 2 * r) = f32x2{s1[r], s2[r]};
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
      constexpr int CPR = WNS / 8;     // 16-byte pieces per staged row
#pragma unroll 2
      for (int it = 0; it < 32 * CPR / 64; ++it) {
        const int idx = lane + 64 * it, row = idx / CPR, c = idx - row * CPR;
        st16<Chunk>((T*)p.y + (size_t)(m0 + wm * WMS + 32 * ps + row) * p.y_ld + n0 + wn * WNS + c * 8,
                    ld16<Chunk>(ws + row * SROW + c * 8));
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// split-K reduction that also emits the per-channel GroupNorm partial sums of the output: one
// workgroup = (sample, row split, 256 / RL channel quads); thread (quad, row lane) owns 4 channels
// and every RL-th row of the split; the row lanes are combined in a fixed order through LDS.
// RL = 16 where a split has 16 rows (4x4 and 8x8 levels), 4 for the 2x2 level.
template <typename T, int RL>
__global__ void __launch_bounds__(256) k_splitk_reduce_stats(ConvP p, int rows_per_split) {
  constexpr int QL = 256 / RL;
  __shared__ float red[RL][QL][8];
  const int HW = p.H * p.W, nq = p.Cout / 4, nqb = (nq + QL - 1) / QL;
  int bid = blockIdx.x;
  const int qb = bid % nqb;
  bid /= nqb;
  const int sp = bid % p.stats_S, b = bid / p.stats_S;
  const int tq = threadIdx.x % QL, tr = threadIdx.x / QL;
  const int q = qb * QL + tq;
  const bool live = q < nq;
  const int n = 4 * q;
  const int r0 = sp * rows_per_split, r1 = r0 + rows_per_split < HW ? r0 + rows_per_split : HW;
  const T* temb = (const T*)p.temb;
  const T* res = (const T*)p.residual;
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  if (live) {
    f32x4 bt = {0.f, 0.f, 0.f, 0.f};
    const bool has_b = p.bias != nullptr;
    if (has_b) bt = *reinterpret_cast<const f32x4*>(p.bias + n);
    float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
    if (temb) load4<T>(temb + (size_t)b * p.temb_stride + n % p.temb_mod, t0, t1, t2, t3);
    for (int pix = r0 + tr; pix < r1; pix += RL) {
      const size_t m = (size_t)b * HW + pix;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      for (int sl = 0; sl < p.splitk; ++sl) v += *reinterpret_cast<const f32x4*>(p.ws + ((size_t)sl * p.M + m) * p.Cout + n);
      // (same association as epilogue_store: ((acc + bias) + temb) + residual)
      if (has_b) v += bt;
      if (temb) { v[0] += t0; v[1] += t1; v[2] += t2; v[3] += t3; }
      if (res) {
        float a0, a1, a2, a3;
        load4<T>(res + m * p.res_ld + n, a0, a1, a2, a3);
        v[0] += a0; v[1] += a1; v[2] += a2; v[3] += a3;
      }
      float vr[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        vr[e] = to_f32(from_f32<T>(v[e]));      // statistics of what the consumer will read
        s1[e] += vr[e];
        s2[e] = fmaf(vr[e], vr[e], s2[e]);
      }
      store4<T>((T*)p.y + m * p.y_ld + n, vr[0], vr[1], vr[2], vr[3]);
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    red[tr][tq][2 * e] = s1[e];
    red[tr][tq][2 * e + 1] = s2[e];
  }
  __syncthreads();
  if (tr == 0 && live) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float a1 = red[0][tq][2 * e], a2 = red[0][tq][2 * e + 1];
#pragma unroll
      for (int r = 1; r < RL; ++r) {
        a1 += red[r][tq][2 * e];
        a2 += red[r][tq][2 * e + 1];
      }
      *reinterpret_cast<f32x2*>(p.stats_out + (((size_t)b * p.stats_S + sp) * p.Cout + n + e) * 2) = f32x2{a1, a2};
    }
  }
}

template <typename T>
__global__ void __launch_bounds__(256) k_splitk_reduce(ConvP p) {
  const int nq = (p.Cout + 3) / 4;
  const size_t total = (size_t)p.M * nq;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int m = (int)(i / nq);
    const int n = (int)(i - (size_t)m * nq) * 4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < p.splitk; ++s) {
      const float* src = p.ws + ((size_t)s * p.M + m) * p.Cout + n;
      if (n + 3 < p.Cout) v += *reinterpret_cast<const f32x4*>(src);
      else
        for (int r = 0; r < 4 && n + r < p.Cout; ++r) v[r] += src[r];
    }
    epilogue_store<T>(p, m, n, v);
  }
}

// ----------------------------------------------------------------------------- small direct forms
// conv_in (Cin = 4): one thread per (pixel, cout); K = KS*KS*Cin is tiny.
template <typename T>
__global__ void __launch_bounds__(256) k_conv_small_cin(ConvP p) {
  const int Ct = p.C1;
  const int HW = p.H * p.W, pad = p.KS >> 1;
  const size_t total = (size_t)p.M * p.Cout;
  const T* x = (const T*)p.x1;
  const T* w = (const T*)p.w;
  T* y = (T*)p.y;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int n = (int)(i % p.Cout);
    const int m = (int)(i / p.Cout);
    const int b = m / HW, pix = m - b * HW, oh = pix / p.W, ow = pix - oh * p.W;
    float acc = p.bias ? p.bias[n] : 0.f;
    for (int kh = 0; kh < p.KS; ++kh) {
      const int ih = oh + kh - pad;
      if (ih < 0 || ih >= p.H) continue;
      for (int kw = 0; kw < p.KS; ++kw) {
        const int iw = ow + kw - pad;
        if (iw < 0 || iw >= p.W) continue;
        const T* xs = x + ((size_t)(b * p.H + ih) * p.W + iw) * Ct;
        const T* wsrc = w + ((size_t)n * p.KS * p.KS + kh * p.KS + kw) * Ct;
        for (int c = 0; c < Ct; ++c) acc = fmaf(to_f32(xs[c]), to_f32(wsrc[c]), acc);
      }
    }
    if (p.temb) acc += to_f32(((const T*)p.temb)[(size_t)b * p.temb_stride + n % p.temb_mod]);
    if (p.residual) acc += to_f32(((const T*)p.residual)[(size_t)m * p.res_ld + n]);
    if (p.out_mode == 0) y[(size_t)m * p.y_ld + n] = from_f32<T>(acc);
    else y[((size_t)b * p.Cout + n) * HW + pix] = from_f32<T>(acc);
  }
}

// conv_in fast path (Cin = 4, KS = 3): weights as fp32 in LDS, one thread = one pixel x a quarter of
// the couts (interleaved in groups of 4 so a wave writes 32-byte runs), inputs in registers.
template <typename T, int CIN, int KS>
__global__ void __launch_bounds__(256) k_conv_cin4(ConvP p) {
  constexpr int KK = KS * KS * CIN;
  static_assert(KK % 4 == 0 && CIN == 4, "k_conv_cin4 assumes 4 input channels");
  extern __shared__ __attribute__((aligned(16))) float wl[];  // [Cout][KK]
  const T* w = (const T*)p.w;
  for (int i = threadIdx.x; i < p.Cout * KK; i += 256) wl[i] = to_f32(w[i]);
  __syncthreads();
  const int HW = p.H * p.W, pad = KS >> 1;
  const int m = blockIdx.x * 64 + (threadIdx.x >> 2), cg = threadIdx.x & 3;
  if (m >= p.M) return;
  const int b = m / HW, pix = m - b * HW, oh = pix / p.W, ow = pix - oh * p.W;
  const T* x = (const T*)p.x1;
  float xin[KK];
#pragma unroll
  for (int kh = 0; kh < KS; ++kh)
#pragma unroll
    for (int kw = 0; kw < KS; ++kw) {
      const int ih = oh + kh - pad, iw = ow + kw - pad, t = (kh * KS + kw) * CIN;
      if (ih >= 0 && ih < p.H && iw >= 0 && iw < p.W)
        load4<T>(x + ((size_t)(b * p.H + ih) * p.W + iw) * CIN, xin[t], xin[t + 1], xin[t + 2], xin[t + 3]);
      else
        xin[t] = xin[t + 1] = xin[t + 2] = xin[t + 3] = 0.f;
    }
  T* y = (T*)p.y + (size_t)m * p.y_ld;
  for (int n0 = 4 * cg; n0 < p.Cout; n0 += 16) {
    float acc[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      acc[e] = p.bias ? p.bias[n0 + e] : 0.f;
      const f32x4* wr = reinterpret_cast<const f32x4*>(wl + (n0 + e) * KK);
#pragma unroll
      for (int k4 = 0; k4 < KK / 4; ++k4) {
        const f32x4 wv = wr[k4];
        acc[e] = fmaf(xin[4 * k4 + 0], wv[0], acc[e]);
        acc[e] = fmaf(xin[4 * k4 + 1], wv[1], acc[e]);
        acc[e] = fmaf(xin[4 * k4 + 2], wv[2], acc[e]);
        acc[e] = fmaf(xin[4 * k4 + 3], wv[3], acc[e]);
      }
    }
    store4<T>(y + n0, acc[0], acc[1], acc[2], acc[3]);
  }
}

// conv_in on MFMA (bf16, Cin = 4, 3x3): a [pixels x 36] x [36 x Cout] GEMM that is purely write-bound
// (25 MB of output at batch 64 against 0.5 MB of input).  K = 36 is padded to two 16x16x32 steps; the
// im2col rows are gathered straight into the B fragments (lane group g of fragment 0 holds taps 2g and
// 2g + 1 of its pixel, fragment 1 only tap 8), the weight rows [Cout][36] are the A fragments read from
// global (13.8 KB, cache resident).  A wave owns 32 pixels x all couts; the tile leaves through a
// wave-private LDS patch as whole 16-byte rows, and the per-channel GroupNorm partial sums of the stored
// (rounded) values are reduced across the 16 pixel lanes with DPP shuffles, then across the 4 waves.
constexpr int CIN4_BM = 128;
__global__ void __launch_bounds__(256) k_conv_cin4_mfma(ConvP p) {
  typedef Mma<bf16> MM;
  typedef MM::Chunk Chunk;
  extern __shared__ __attribute__((aligned(16))) char smem_c4[];
  const int SROW = p.Cout + 8;
  bf16* stg = reinterpret_cast<bf16*>(smem_c4);
  float* ssum = reinterpret_cast<float*>(smem_c4 + (size_t)CIN4_BM * SROW * sizeof(bf16));   // [4][Cout][2]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
  const int HW = p.H * p.W;
  const int mb = blockIdx.x * CIN4_BM, mw = mb + wave * 32;
  const bf16* x = (const bf16*)p.x1;
  const bf16* w = (const bf16*)p.w;
  // ---- B fragments: im2col rows of the wave's 2 x 16 pixels
  Chunk bq[2][2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int m = mw + 16 * mt + li;
    const bool live = m < p.M;
    const int mm = live ? m : 0;
    const int b = mm / HW, pix = mm - b * HW, oh = pix / p.W, ow = pix - oh * p.W;
    bf16x4 tp[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int tap = q < 2 ? 2 * lg + q : 8;
      const int kh = tap / 3, kw = tap - 3 * kh;
      const int ih = oh + kh - 1, iw = ow + kw - 1;
      const bool ok = live && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W && (q < 2 || lg == 0);
      bf16x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = (bf16)0.0f;
      if (ok) v = *reinterpret_cast<const bf16x4*>(x + ((size_t)(b * p.H + ih) * p.W + iw) * 4);
      tp[q] = v;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      bq[mt][0][e] = tp[0][e];
      bq[mt][0][4 + e] = tp[1][e];
      bq[mt][1][e] = tp[2][e];
      bq[mt][1][4 + e] = (bf16)0.0f;
    }
  }
  const int NT = p.Cout / 16;
  bf16* wst = stg + (size_t)wave * 32 * SROW;
  for (int t = 0; t < NT; ++t) {
    // ---- A fragments: weight row 16 t + li, k = 8 lg .. 8 lg + 7 and (lg == 0) k = 32 .. 35
    const bf16* wr = w + (size_t)(16 * t + li) * 36;
    Chunk a0, a1 = MM::zero();
    {
      const bf16x4 lo = *reinterpret_cast<const bf16x4*>(wr + 8 * lg);
      const bf16x4 hi = *reinterpret_cast<const bf16x4*>(wr + 8 * lg + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { a0[e] = lo[e]; a0[4 + e] = hi[e]; }
      if (lg == 0) {
        const bf16x4 tl = *reinterpret_cast<const bf16x4*>(wr + 32);
#pragma unroll
        for (int e = 0; e < 4; ++e) a1[e] = tl[e];
      }
    }
    const int n0 = 16 * t + 4 * lg;
    float bv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[r] = p.bias ? p.bias[n0 + r] : 0.f;
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      MM::mma(acc, a0, bq[mt][0]);
      MM::mma(acc, a1, bq[mt][1]);
      bf16x4 o;
      const bool live = mw + 16 * mt + li < p.M;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        o[r] = (bf16)(acc[r] + bv[r]);
        const float vr = live ? (float)o[r] : 0.f;
        s1[r] += vr;
        s2[r] = fmaf(vr, vr, s2[r]);
      }
      *reinterpret_cast<bf16x4*>(wst + (size_t)(16 * mt + li) * SROW + n0) = o;
    }
    if (p.stats_out) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
          s1[r] += __shfl_xor(s1[r], o, 64);
          s2[r] += __shfl_xor(s2[r], o, 64);
        }
      }
      if (li == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) *reinterpret_cast<f32x2*>(ssum + ((size_t)wave * p.Cout + n0 + r) * 2) = f32x2{s1[r], s2[r]};
      }
    }
  }
  __syncthreads();
  // ---- copy out: whole rows, 16 bytes per lane (the patch is wave-private, the barrier is for ssum)
  {
    const int cpr = p.Cout / 8;                  // 16-byte chunks per row
    bf16* y = (bf16*)p.y;
    for (int i = lane; i < 32 * cpr; i += 64) {
      const int row = i / cpr, c = i - row * cpr;
      const int m = mw + row;
      if (m < p.M) st16<Chunk>(y + (size_t)m * p.y_ld + c * 8, ld16<Chunk>(wst + (size_t)row * SROW + c * 8));
    }
  }
  if (p.stats_out) {
    const int b = mb / HW, sidx = (mb - b * HW) / CIN4_BM;
    for (int c = tid; c < p.Cout; c += 256) {
      float a = 0.f, q = 0.f;
#pragma unroll
      for (int wv = 0; wv < 4; ++wv) {
        a += ssum[((size_t)wv * p.Cout + c) * 2];
        q += ssum[((size_t)wv * p.Cout + c) * 2 + 1];
      }
      *reinterpret_cast<f32x2*>(p.stats_out + (((size_t)b * p.stats_S + sidx) * p.Cout + c) * 2) = f32x2{a, q};
    }
  }
}

// conv_out (Cout = 4): one wave per pixel, lanes stride the channels, butterfly reduce.
template <typename T, int NOUT>
__global__ void __launch_bounds__(256) k_conv_small_cout(ConvP p) {
  const int Ct = p.C1;
  const int HW = p.H * p.W, pad = p.KS >> 1;
  const int lane = threadIdx.x & 63;
  const int wave_global = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  const T* x = (const T*)p.x1;
  const T* w = (const T*)p.w;
  T* y = (T*)p.y;
  for (int m = wave_global; m < p.M; m += nwaves) {
    const int b = m / HW, pix = m - b * HW, oh = pix / p.W, ow = pix - oh * p.W;
    float acc[NOUT];
#pragma unroll
    for (int n = 0; n < NOUT; ++n) acc[n] = 0.f;
    for (int kh = 0; kh < p.KS; ++kh) {
      const int ih = oh + kh - pad;
      if (ih < 0 || ih >= p.H) continue;
      for (int kw = 0; kw < p.KS; ++kw) {
        const int iw = ow + kw - pad;
        if (iw < 0 || iw >= p.W) continue;
        const T* xs = x + ((size_t)(b * p.H + ih) * p.W + iw) * Ct;
        const int tap = kh * p.KS + kw;
        for (int c = lane; c < Ct; c += 64) {
          const float xv = to_f32(xs[c]);
#pragma unroll
          for (int n = 0; n < NOUT; ++n)
            if (n < p.Cout) acc[n] = fmaf(xv, to_f32(w[((size_t)n * p.KS * p.KS + tap) * Ct + c]), acc[n]);
        }
      }
    }
#pragma unroll
    for (int n = 0; n < NOUT; ++n)
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) acc[n] += __shfl_xor(acc[n], o, 64);
    if (lane == 0) {
      for (int n = 0; n < p.Cout; ++n) {
        float s = acc[n] + (p.bias ? p.bias[n] : 0.f);
        if (p.temb) s += to_f32(((const T*)p.temb)[(size_t)b * p.temb_stride + n % p.temb_mod]);
        if (p.residual) s += to_f32(((const T*)p.residual)[(size_t)m * p.res_ld + n]);
        if (p.out_mode == 0) y[(size_t)m * p.y_ld + n] = from_f32<T>(s);
        else y[((size_t)b * p.Cout + n) * HW + pix] = from_f32<T>(s);
      }
    }
  }
}

// ----------------------------------------------------------------------------- host dispatch
// tuning overrides (afldm_conv2d_tune): -1 = automatic
static int g_force_variant = -1, g_force_splitk = -1, g_fused_splitk = 0;

struct Plan {
  int kind;  // 0 igemm, 1 small_cin, 2 small_cout
  int cfg;   // igemm tile config
  int splitk;
  int cfg_auto, splitk_auto;   // the automatic choice (fallback when a forced / preferred variant cannot run the shape)
};

constexpr int KCH_DEFAULT = 2;

struct Variant {
  int bm, bn, ver, stages;
};
// index = variant id used by afldm_conv2d_tune / the automatic chooser
static const Variant kVariants[] = {
    {128, 128, 1, 2},  // 0
    {128, 64, 1, 2},   // 1
    {64, 128, 1, 2},   // 2
    {64, 64, 1, 2},    // 3
    {128, 128, 2, 2},  // 4
    {128, 128, 2, 3},  // 5
    {128, 64, 2, 2},   // 6
    {128, 64, 2, 3},   // 7
    {128, 64, 2, 4},   // 8
    {64, 128, 2, 3},   // 9
    {64, 128, 2, 4},   // 10
    {64, 64, 2, 4},    // 11
    {128, 192, 2, 2},  // 12
    {256, 64, 2, 2},   // 13
    {256, 64, 2, 3},   // 14
    {64, 64, 2, 6},    // 15
    {128, 192, 2, 4},  // 16  (K step = one 64-byte row piece: KCH = 1)
    {128, 128, 2, 4},  // 17  (KCH = 1)
    {128, 192, 2, 3},  // 18  (KCH = 1)
    {128, 192, 2, 2},  // 19  (KCH = 1: 40 KB LDS -> 4 workgroups per CU)
    {128, 128, 2, 2},  // 20  (KCH = 1: 32 KB LDS)
    {128, 192, 3, 2},  // 21  wave-specialised: 4 consumer + 4 producer waves
    {128, 192, 3, 3},  // 22
    {128, 128, 3, 2},  // 23
    {128, 128, 3, 3},  // 24
    {128, 192, 3, 2},  // 25  4 consumers + 2 producers
    {256, 192, 3, 2},  // 26  8 consumers (4x2) + 4 producers, 112 KB LDS
    {256, 128, 3, 2},  // 27  8 consumers (4x2) + 4 producers, 96 KB LDS
    {128, 64, 3, 3},   // 28  4 consumers + 2 producers
    {128, 192, 4, 2},  // 29  ver 4 = 128-byte LDS rows (full-line LDS-DMA)
    {128, 128, 4, 2},  // 30
    {128, 64, 4, 2},   // 31
    {64, 64, 4, 4},    // 32
    {128, 192, 4, 3},  // 33  + 4 producer waves
    {128, 128, 4, 2},  // 34  + 4 producer waves
    {128, 128, 4, 3},  // 35
    {256, 64, 4, 2},   // 36
    {64, 192, 4, 2},   // 37  short-K GEMMs: twice the tiles of 128x192 (epilogues of one round overlap the next)
    {64, 128, 4, 2},   // 38
    {64, 192, 4, 3},   // 39
    {128, 192, 5, 3},  // 40  ver 5 = persistent tiles (k_igemm3): producers run ahead across tile boundaries
    {256, 192, 6, 3},  // 41  ver 6 = halo-patch 3x3 (conv3h.hip): 32x32 planes, 8 image rows per tile, 8 + 4 waves
    {128, 192, 6, 3},  // 42  16x16 planes, 8 rows per tile, 8 consumers (64x48) + 4 producers
    {128, 192, 6, 3},  // 43  16x16 planes, 4 consumers (64x96) + 4 producers
    {256, 192, 6, 3},  // 44  16x16 planes, one whole sample per tile
    {128, 192, 6, 3},  // 45  32x32 planes, 4 rows per tile, 8 + 4 waves
    {128, 192, 6, 3},  // 46  32x32 planes, 4 rows per tile, 4 + 4 waves
    {256, 192, 6, 3},  // 47  = 41 on 32x32x16 MFMAs
    {128, 192, 6, 3},  // 48  = 43 on 32x32x16 MFMAs
    {256, 192, 6, 3},  // 49  = 44 on 32x32x16 MFMAs
    {128, 192, 6, 3},  // 50  = 46 on 32x32x16 MFMAs
    {64, 96, 6, 3},    // 51  halo-patch, 8x8 planes: one sample x 96 couts per tile, 3 taps per K step, no split-K
    {64, 96, 6, 3},    // 52  4x4 planes: four samples x 96 couts per tile, split-K over the channel blocks
    {64, 96, 6, 3},    // 53  = 51
    {128, 96, 6, 3},   // 54  16x16 planes, 8 rows x 96 couts (small batches)
    {128, 96, 6, 3},   // 55  32x32 planes, 4 rows x 96 couts (small batches)
    {64, 32, 4, 12},   // 56  64x32 tiles, 12 stages (a skinny GEMM's K step costs DMA latency / ring depth): dense layers over <= 64 rows (the 2x2 level at batch 64) WITHOUT split-K
    {256, 192, 6, 2},  // 57  = 41 with a 2-deep weight ring (lookahead experiment)
    {256, 128, 6, 3},  // 58  halo-patch on planes of 64^2 and up (AF-VAE): 8 x 32 pixel blocks x 128 couts
    {256, 128, 6, 3},  // 59  = 58 with 4 consumer waves (128 x 64 each)
    {256, 128, 6, 3},  // 60  = 58 on 32x32x16 MFMAs
    {128, 96, 6, 2},   // 61  halo-patch, 32x32 planes, 128 x 96 tiles sized for TWO workgroups per CU (4 + 2 waves, 2-deep ring)
    {128, 96, 6, 2},   // 62  the same for 16x16 planes
    {256, 128, 6, 3},  // 63  = 58 with persistent workgroups (k_conv3h_pers: one workgroup per CU walks its tiles)
    {128, 192, 6, 3},  // 64  32x32 planes, 4 rows x 192 couts per tile, persistent workgroups (two tiles per CU at batch 64)
    {64, 96, 6, 3},    // 65  32x32 planes, 2 rows x 96 couts per tile (small batches)
    {64, 96, 6, 3},    // 66  16x16 planes, 4 rows x 96 couts per tile (small batches)
    {128, 48, 6, 3},   // 67  8x8 planes: TWO samples x 48 couts per tile (half the weight bytes per workgroup of 51; round 5)
    {128, 48, 6, 3},   // 68  4x4 planes: EIGHT samples x 48 couts per tile
};
constexpr int kNumVariants = (int)(sizeof(kVariants) / sizeof(kVariants[0]));

template <typename T>
static int epr() { return 4 * Mma<T>::EPC; }

static Plan make_plan(const afldm_conv_args* a, int elems_per_row) {
  Plan pl{0, 0, 1, 0, 1};
  const int Ct = a->C1 + a->C2;
  const int kstep = KCH_DEFAULT * elems_per_row;
  const bool gemm_ok = (Ct % kstep == 0) && (a->C2 == 0 || a->C1 % kstep == 0);
  if (!gemm_ok) {
    pl.kind = (a->Cout <= 8) ? 2 : 1;
    return pl;
  }
  const long long M = (long long)a->B * a->H * a->W;
  // Variant choice from the MI355X sweeps (tools/bench_kernels.py conv; profiles/r01/conv_variant_sweep*.log).
  // All picks use 128-byte LDS rows (full-line LDS-DMA, variants 29-33):
  //   short K (<= 16 steps) or M <= 4096 with <= 64 steps : 64x64, 4 stages, no split-K
  //   M >= 32768, Cout % 192 == 0                          : 128x192, 2 workgroups per CU      (~1000-1080 TF)
  //   1024 <= M < 32768, Cout % 192 == 0, >= 27 K steps    : 128x192 + 4 producer waves, 3 stages, 1 workgroup
  //                                                          per CU (256 tiles = one full wave at 16x16; +15-25 %)
  //   M >= 4096, Cout % 128 == 0                           : 128x128
  //   M >= 1024                                            : 128x64
  // and split-K so that tiles * splitk ~ one resident set of workgroups.
  int vid, bm, bn;
  const int ksteps_all = a->KS * a->KS * (Ct / kstep);
  const long long tiles64 = ((M + 63) / 64) * ((a->Cout + 63) / 64);
  if (M <= 4096 && ((ksteps_all <= 64 && tiles64 >= 256 && !(a->Cout % 192 == 0 && ksteps_all >= 27 && M >= 4096)) ||
                    (ksteps_all <= 16 && tiles64 >= 128))) {
    vid = 32; bm = 64; bn = 64;
    // more than two resident rounds of 64x64 tiles (the q|k|v projection of the 4x4 level at batch 64: 576): 128x64
    // tiles halve the rounds (in situ 5.180 -> 5.159 ms/step; 128x128 / 128x192 with K slices measured slower)
    if (a->KS == 1 && tiles64 >= 512 && M % 128 == 0) { vid = 31; bm = 128; bn = 64; }
  }
  else if (M >= 32768 && a->Cout % 192 == 0) { vid = 29; bm = 128; bn = 192; }
  else if (a->Cout % 192 == 0 && ((M >= 4096 && ksteps_all >= 27) || (M >= 1024 && ksteps_all >= 100))) { vid = 33; bm = 128; bn = 192; }
  else if (M >= 4096 && a->Cout % 128 == 0 && a->KS > 1) { vid = 30; bm = 128; bn = 128; }
  else if (M >= 1024) { vid = 31; bm = 128; bn = 64; }
  else { vid = 32; bm = 64; bn = 64; }
  const int ksteps = a->KS * a->KS * (Ct / kstep);
  auto splitk_for = [&](int v) {
    const int bm_ = kVariants[v].bm, bn_ = kVariants[v].bn;
    const long long tiles = ((M + bm_ - 1) / bm_) * ((a->Cout + bn_ - 1) / bn_);
    int sk = 1;
    if (v == 56) return 1;
    if (kVariants[v].ver == 6) {
      // halo-patch kernel: the small-tile variants (51+) may split the CHANNEL BLOCKS over up to 4 slices when the
      // tiles alone leave most CUs idle; the slice count must divide the number of 128-byte channel blocks
      if (v < 51 || (v >= 57 && v != 65 && v != 66 && v != 67 && v != 68)) return 1;
      const int ncb = Ct / (2 * elems_per_row);
      int z = 1;
      static const int s_zmax = getenv("AFLDM_CONV3H_ZMAX") ? atoi(getenv("AFLDM_CONV3H_ZMAX")) : 8;      // (4 -> 8: batch 8 2.410 -> 2.378, batch 1 2.158 -> 2.134 ms/step, same box)
      for (int c = 2; c <= s_zmax; ++c)
        if (ncb % c == 0 && tiles * c <= 320) z = c;
      // AFLDM_CONV3H_Z1=1: no slices from 128 tiles on (the 4x4 level at batch 64 through the per-sample epilogue of the
      // multi-sample tiles: 12.6 MB less slab traffic and one launch less per layer, conv3x3 family 1.52x -> 1.43x of
      // its algorithmic bytes) - but 128 workgroups walking 36 K steps lose to 256 walking 18 plus the reduction:
      // 5.37 -> 5.50 ms/step.  Off.
      static const int s_z1 = getenv("AFLDM_CONV3H_Z1") ? atoi(getenv("AFLDM_CONV3H_Z1")) : 0;
      if (s_z1 && tiles >= 128 && elems_per_row == 32) z = 1;
      return z;
    }
    if (kVariants[v].ver == 4 && kVariants[v].stages == 3 && bn_ == 192) {
      // one 8-wave workgroup per CU: aim at 256 workgroups (measured best: 64 tiles -> 4, 128 tiles -> 2)
      sk = (int)((256 + tiles / 2) / tiles);
      int maxsk = ksteps / 8;
      if (sk > maxsk) sk = maxsk;
      if (sk > 8) sk = 8;
      if (sk < 1) sk = 1;
    } else if (tiles < 256 && ksteps > 16) {   // (short K: a split only adds the reduction pass)   // measured: ~320 workgroups is the sweet spot (conv_variant_sweep2/4.log)
      const int cap = ksteps >= 100 ? 8 : 4;      // long K (the 4x4 / 2x2 levels): 8 slices measured best in situ
      sk = (int)(((ksteps >= 192 ? 768 : (cap == 8 ? 384 : 320)) + tiles - 1) / tiles);
      int maxsk = ksteps / 4;
      if (maxsk < 1) maxsk = 1;
      if (sk > maxsk) sk = maxsk;
      if (sk > cap) sk = cap;
      if (sk < 1) sk = 1;
    }
    return sk;
  };
  pl.cfg_auto = vid;
  pl.splitk_auto = splitk_for(vid);
  // halo-patch kernel for the 3x3 convolutions of the 32x32 / 16x16 levels (conv3h.hip); support is checked by resolve_exec
  {
    static const int s_h3 = getenv("AFLDM_CONV3H") ? atoi(getenv("AFLDM_CONV3H")) : 4;      // 0: off, 1: 32x32 / 16x16 only, 2: + variant 42 at 16x16, 3: + 8x8 / 4x4, 4: + small batches
    if (s_h3 && a->KS == 3 && a->C2 == 0 && a->H == a->W && a->Cout % 192 == 0) {
      // tile by plane size; the large tiles only while they still give every CU a workgroup
      const long long t256 = (M / 256) * (a->Cout / 192), t128 = (M / 128) * (a->Cout / 192);
      const bool bf = elems_per_row == 32;      // (64-byte row pieces: 32 bf16 / 16 fp32)
      static const int s_co = getenv("AFLDM_CONV3H_CO") ? atoi(getenv("AFLDM_CONV3H_CO")) : 0;      // bit 0: 32x32 planes, bit 1: 16x16 planes on the two-per-CU tiles (A/B)
      const long long t96 = (M / 128) * (a->Cout / 96);
      if (bf && (s_co & 1) && a->W == 32 && t96 >= 512 && a->Cout % 96 == 0) vid = 61;
      else if (bf && (s_co & 2) && a->W == 16 && t96 >= 512 && a->Cout % 96 == 0) vid = 62;
      else if (a->W == 32 && t256 >= 192) vid = 41;
      else if (a->W == 32 && t128 >= 192 && s_h3 >= 4 && bf) vid = 46;
      else if (a->W == 16 && t128 >= 192) vid = s_h3 == 2 ? 42 : 43;      // 4 consumer waves (64x96) measured best at 16x16
      else if (s_h3 >= 3 && a->W == 8 && M >= 2048 && a->Cout % 96 == 0) vid = 51;
      else if (s_h3 >= 3 && a->W == 4 && M >= 1024 && a->Cout % 96 == 0) vid = 52;
      else if (s_h3 >= 4 && bf && a->Cout % 96 == 0) {      // small batches (bf16; fp32 measured slower: 4.99 vs 4.38 ms/step at batch 1): the 96-cout tiles with split channel blocks
        // 64-pixel tiles (65 / 66) while the 128-pixel ones leave CUs without a workgroup: per launch 13.2 -> 10.2 us
        // (batch 8, 32^2), 18.2 -> 14.9 (batch 8, 16^2 incl. the reduction), 11.5 -> 9.7 (batch 1); in the step 2.463 -> 2.410
        // (batch 8), 2.198 -> 2.153 (batch 1), 2.80 -> 2.75 ms (batch 16), same box (profiles/r04/small_batch_tiles_ab.txt).
        // AFLDM_CONV3H_SB: bit 0: 32x32 planes, bit 1: 16x16 planes (0 = the 128-pixel tiles, for A/B)
        static const int s_sb = getenv("AFLDM_CONV3H_SB") ? atoi(getenv("AFLDM_CONV3H_SB")) : 3;
        const long long t128s = (M / 128) * (a->Cout / 96);
        if (a->W == 32 && M >= 1024) vid = ((s_sb & 1) && t128s < 256) ? 65 : 55;
        else if (a->W == 16 && M >= 256) vid = ((s_sb & 2) && t128s < 256) ? 66 : 54;
        else if (a->W == 8 && M >= 64) vid = 51;
        else if (a->W == 4 && M >= 64) vid = 52;
      }
      // long K at full batch: 128-pixel x 48-cout tiles (67 / 68) stream half the weight bytes per workgroup - bit-identical,
      // 26.0 vs 28.4 us (768 -> 384), 36.0 vs 41.3 (1152 -> 384), 48.9 vs 56.0 (768 -> 768) at 8^2; 30.3 vs 32.9 (1536 -> 768) at 4^2;
      // nothing for the 384-channel layers (profiles/r05/conv_small_tiles_ab.txt).  AFLDM_CONV3H_NARROW=0: off (A/B)
      static const int s_narrow = getenv("AFLDM_CONV3H_NARROW") ? atoi(getenv("AFLDM_CONV3H_NARROW")) : 1;
      if (s_narrow && bf && vid == 51 && M >= 4096 && M % 128 == 0 && Ct >= 768 && a->Cout % 48 == 0) vid = 67;
      if (s_narrow && bf && vid == 52 && M >= 1024 && M % 128 == 0 && Ct >= 1152 && a->Cout % 48 == 0) vid = 68;
    }
  }
  {
    // planes of 64^2 and up (the AF-VAE): the halo-patch kernel on 8 x 32 pixel blocks of the plane (variant 58)
    static const bool off = getenv("AFLDM_NO_CONV3H_SUB") != nullptr;
    if (!off && a->KS == 3 && a->C2 == 0 && a->W >= 64 && a->W % 32 == 0 && a->H % 8 == 0 && a->Cout % 128 == 0 &&
        Ct % (2 * elems_per_row) == 0 && M >= 256 * 192) {
      static const int s_subv = getenv("AFLDM_CONV3H_SUBV") ? atoi(getenv("AFLDM_CONV3H_SUBV")) : 63;     // 58 / 59 / 60 / 63 (A/B)
      vid = (s_subv >= 58 && s_subv <= 60) || s_subv == 63 ? s_subv : 58;
      if (vid == 63 && !(elems_per_row == 32 && M / 256 * (a->Cout / 128) >= 512)) vid = 58;      // bf16; persistent tiles pay from two tiles per CU
    }
  }
  {
    // a dense layer over <= 64 rows (the 3x3 convolutions of the 2x2 level in their flattened form at batch 64) on 96+
    // 64x32 tiles that walk the whole K themselves (no slabs, no reduction launch): MEASURED SLOWER - 28.7 us kernel-only
    // against 11.1 + 4.3 us for four slices of 64x64 tiles (a K step of this skinny tile costs ~0.6 us whatever the
    // ring depth: 4 and 12 stages measure the same), 5.39 vs 5.26 ms/step.  Off unless AFLDM_DENSE_NOSPLIT is set.
    static const bool off = getenv("AFLDM_DENSE_NOSPLIT") == nullptr;
    if (!off && a->KS == 1 && a->H * a->W == 1 && M <= 64 && a->Cout >= 2048 && a->Cout % 32 == 0) vid = 56;
  }
  // in-situ tuning hook (tools/tune_insitu.py): AFLDM_CONV_OVERRIDE="M:Cout:KS:Ct=variant/splitk;..."
  int ov_sk = -1;
  {
    static const char* ov = getenv("AFLDM_CONV_OVERRIDE");
    if (ov) {
      for (const char* q = ov; q && *q;) {
        long long m_ = 0; int co = 0, ks = 0, ct = 0, v = -1, sk_ = -1;
        if (sscanf(q, "%lld:%d:%d:%d=%d/%d", &m_, &co, &ks, &ct, &v, &sk_) == 6 && m_ == M && co == a->Cout && ks == a->KS &&
            ct == Ct && v >= 0 && v < kNumVariants) {
          vid = v; ov_sk = sk_;
        }
        q = strchr(q, ';');
        if (q) ++q;
      }
    }
  }
  if (g_force_variant >= 0 && g_force_variant < kNumVariants) vid = g_force_variant;
  pl.cfg = vid;
  pl.splitk = splitk_for(vid);
  if (ov_sk >= 1) pl.splitk = ov_sk;
  if (g_force_splitk >= 1) pl.splitk = g_force_splitk;
  return pl;
}

template <typename T, int BM, int BN, int WGM, int WGN>
static void launch_igemm(const ConvP& p0, hipStream_t st) {
  ConvP p = p0;
  constexpr int KCH = KCH_DEFAULT;
  p.tiles_n = (p.Cout + BN - 1) / BN;
  const int tiles_m = (p.M + BM - 1) / BM;
  dim3 grid(tiles_m * p.tiles_n, 1, p.splitk);
  constexpr int lds = 2 * KCH * (BM + BN) * 64;
  static unsigned long long attr_set = 0;
  if (first_on_device(attr_set)) {
    (void)hipFuncSetAttribute((const void*)k_igemm<T, BM, BN, WGM, WGN, KCH>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  }
  k_igemm<T, BM, BN, WGM, WGN, KCH><<<grid, WGM * WGN * 64, lds, st>>>(p);
}

template <typename T, int BM, int BN, int WGM, int WGN, int STAGES, int KCH = KCH_DEFAULT, int NPROD = 0, bool R128 = false, int MINW = 1>
static void launch_igemm2(const ConvP& p0, hipStream_t st) {
  ConvP p = p0;
  p.ksteps = p0.ksteps * KCH_DEFAULT / KCH;     // p0.ksteps counts KCH_DEFAULT-wide steps
  p.tiles_n = (p.Cout + BN - 1) / BN;
  const int tiles_m = (p.M + BM - 1) / BM;
  dim3 grid(tiles_m * p.tiles_n, 1, p.splitk);
  constexpr int lds = STAGES * KCH * (BM + BN) * 64;
  static unsigned long long attr_set = 0;
  if (first_on_device(attr_set)) {
    (void)hipFuncSetAttribute((const void*)k_igemm2<T, BM, BN, WGM, WGN, KCH, STAGES, NPROD, R128, MINW>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  }
  k_igemm2<T, BM, BN, WGM, WGN, KCH, STAGES, NPROD, R128, MINW><<<grid, (WGM * WGN + NPROD) * 64, lds, st>>>(p);
}


template <typename T>
static void launch_igemm3(const ConvP& p0, hipStream_t st) {
  if constexpr (sizeof(T) == 2) {
    constexpr int STAGES = 3;
    ConvP p = p0;
    p.tiles_n = p.Cout / 192;
    const int tiles = (p.M / 128) * p.tiles_n;
    constexpr int lds = STAGES * (128 + 192) * 128 + 4 * 32 * (96 + 8) * 2;
    static unsigned long long attr_set = 0;
    if (first_on_device(attr_set)) {
      (void)hipFuncSetAttribute((const void*)k_igemm3<T, STAGES>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    }
    k_igemm3<T, STAGES><<<tiles < 256 ? tiles : 256, 512, lds, st>>>(p);
  }
}

// shapes the persistent-tile kernel covers
template <typename T>
static bool igemm3_ok(const afldm_conv_args* a) {
  const int HW = a->H * a->W;
  const int Ct = a->C1 + a->C2;
  return sizeof(T) == 2 && a->out_mode == 0 && !a->y2 && HW % 128 == 0 && a->Cout % 192 == 0 && Ct % 64 == 0 &&
         (a->C2 == 0 || a->C1 % 64 == 0) && a->y_ld % 8 == 0 && (!a->residual || a->res_ld % 4 == 0) &&
         (!a->temb || (a->temb_stride % 4 == 0 && (a->temb_mod <= 0 || a->temb_mod % 4 == 0))) && aligned16(a->y) &&
         (!a->bias || aligned16(a->bias));
}

template <typename T>
static bool launch_variant(int id, const ConvP& p, hipStream_t st) {
  if (id == 56) {
    launch_igemm2<T, 64, 32, 2, 2, 12, 2, 0, true>(p, st);
    return true;
  }
  if (id >= kConv3hFirst) {
    conv3h_launch(id, (int)sizeof(T), p, st);
    return true;
  }
  switch (id) {
    case 40: launch_igemm3<T>(p, st); return true;
    case 0: launch_igemm<T, 128, 128, 2, 2>(p, st); return true;
    case 1: launch_igemm<T, 128, 64, 2, 2>(p, st); return true;
    case 2: launch_igemm<T, 64, 128, 2, 2>(p, st); return true;
    case 3: launch_igemm<T, 64, 64, 2, 2>(p, st); return true;
    case 4: launch_igemm2<T, 128, 128, 2, 2, 2>(p, st); return true;
    case 5: launch_igemm2<T, 128, 128, 2, 2, 3>(p, st); return true;
    case 6: launch_igemm2<T, 128, 64, 2, 2, 2>(p, st); return true;
    case 7: launch_igemm2<T, 128, 64, 2, 2, 3>(p, st); return true;
    case 8: launch_igemm2<T, 128, 64, 2, 2, 4>(p, st); return true;
    case 9: launch_igemm2<T, 64, 128, 2, 2, 3>(p, st); return true;
    case 10: launch_igemm2<T, 64, 128, 2, 2, 4>(p, st); return true;
    case 11: launch_igemm2<T, 64, 64, 2, 2, 4>(p, st); return true;
    case 12: launch_igemm2<T, 128, 192, 2, 2, 2, KCH_DEFAULT, 0, false, 2>(p, st); return true;
    case 13: launch_igemm2<T, 256, 64, 4, 1, 2>(p, st); return true;
    case 14: launch_igemm2<T, 256, 64, 4, 1, 3>(p, st); return true;
    case 15: launch_igemm2<T, 64, 64, 2, 2, 6>(p, st); return true;
    case 16: launch_igemm2<T, 128, 192, 2, 2, 4, 1>(p, st); return true;
    case 17: launch_igemm2<T, 128, 128, 2, 2, 4, 1>(p, st); return true;
    case 18: launch_igemm2<T, 128, 192, 2, 2, 3, 1>(p, st); return true;
    case 19: launch_igemm2<T, 128, 192, 2, 2, 2, 1>(p, st); return true;
    case 20: launch_igemm2<T, 128, 128, 2, 2, 2, 1>(p, st); return true;
    case 21: launch_igemm2<T, 128, 192, 2, 2, 2, 2, 4>(p, st); return true;
    case 22: launch_igemm2<T, 128, 192, 2, 2, 3, 2, 4>(p, st); return true;
    case 23: launch_igemm2<T, 128, 128, 2, 2, 2, 2, 4>(p, st); return true;
    case 24: launch_igemm2<T, 128, 128, 2, 2, 3, 2, 4>(p, st); return true;
    case 25: launch_igemm2<T, 128, 192, 2, 2, 2, 2, 2>(p, st); return true;
    case 26: launch_igemm2<T, 256, 192, 4, 2, 2, 2, 4, true, 3>(p, st); return true;
    case 27: launch_igemm2<T, 256, 128, 4, 2, 2, 2, 4>(p, st); return true;
    case 28: launch_igemm2<T, 128, 64, 2, 2, 3, 2, 2>(p, st); return true;
    case 29: launch_igemm2<T, 128, 192, 2, 2, 2, 2, 0, true, 2>(p, st); return true;
    case 30: launch_igemm2<T, 128, 128, 2, 2, 2, 2, 0, true, 2>(p, st); return true;
    case 31: launch_igemm2<T, 128, 64, 2, 2, 2, 2, 0, true, 3>(p, st); return true;
    case 32: launch_igemm2<T, 64, 64, 2, 2, 4, 2, 0, true>(p, st); return true;
    case 33: launch_igemm2<T, 128, 192, 2, 2, 3, 2, 4, true>(p, st); return true;
    case 34: launch_igemm2<T, 128, 128, 2, 2, 2, 2, 4, true>(p, st); return true;
    case 35: launch_igemm2<T, 128, 128, 2, 2, 3, 2, 0, true>(p, st); return true;
    case 36: launch_igemm2<T, 256, 64, 4, 1, 2, 2, 0, true>(p, st); return true;
    case 37: launch_igemm2<T, 64, 192, 2, 2, 2, 2, 0, true, 2>(p, st); return true;
    case 38: launch_igemm2<T, 64, 128, 2, 2, 2, 2, 0, true, 2>(p, st); return true;
    case 39: launch_igemm2<T, 64, 192, 2, 2, 3, 2, 0, true, 2>(p, st); return true;
  }
  return false;
}

// How afldm_conv2d will run a problem (shared by the dispatch and the host-side queries).
struct Exec {
  Plan pl;
  int vid;       // igemm variant actually launched
  int splitk;    // after the workspace check
  int fused;     // split-K slices reduced inside the GEMM launch (splitk_fused_reduce)
};

static int device_cus() {
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t pr;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) cus = pr.multiProcessorCount;
    else cus = -1;
  }
  return cus;
}
template <typename T>
static Exec resolve_exec(const afldm_conv_args* a) {
  Exec e;
  e.pl = make_plan(a, epr<T>());
  e.vid = e.pl.cfg;
  e.splitk = e.pl.kind == 0 ? e.pl.splitk : 1;
  if (e.pl.kind != 0) return e;
  const long long M = (long long)a->B * a->H * a->W;
  const int Ct = a->C1 + a->C2;
  if (e.splitk > 1) {
    const size_t need = (size_t)e.splitk * M * a->Cout * sizeof(float);
    if (!a->workspace || a->workspace_bytes < need) e.splitk = 1;  // no workspace -> no split
  }
  const bool v2_ok = M * (a->C1 > a->C2 ? a->C1 : a->C2) * (long long)sizeof(T) < (1ll << 31) &&
                     (long long)a->Cout * a->KS * a->KS * Ct * (long long)sizeof(T) < (1ll << 31);
  {
    // persistent tiles (variant 40) where the 128x192 one-tile-per-slot kernel would run >= 2 rounds of tiles
    static const int s_persist = getenv("AFLDM_PERSIST") ? atoi(getenv("AFLDM_PERSIST")) : 0;
    const long long tiles = (M / 128) * (a->Cout / 192);
    if (s_persist && e.vid == 29 && e.splitk == 1 && tiles >= 512 && igemm3_ok<T>(a)) e.vid = 40;
    if (e.vid == 40 && (e.splitk != 1 || !igemm3_ok<T>(a))) e.vid = 29;
  }
  if (kVariants[e.vid].ver == 6) {
    ConvP q;
    memset(&q, 0, sizeof(q));
    q.x1 = a->x1; q.w = a->w; q.y = a->y; q.y2 = a->y2; q.residual = a->residual; q.temb = a->temb;
    q.C1 = a->C1; q.C2 = a->C2; q.H = a->H; q.W = a->W; q.Cout = a->Cout; q.KS = a->KS; q.M = (int)M;
    q.out_mode = a->out_mode; q.y_ld = a->y_ld; q.res_ld = a->res_ld; q.temb_stride = a->temb_stride;
    q.temb_mod = a->temb_mod > 0 ? a->temb_mod : a->Cout;
    q.splitk = e.splitk;               // (after the workspace check: a shape that needs its slices cannot run without them)
    if (!conv3h_supported(e.vid, (int)sizeof(T), q)) {       // not this kernel's shape: the automatic choice
      e.vid = e.pl.cfg_auto;
      e.splitk = e.pl.splitk_auto;
      if (e.splitk > 1) {
        const size_t need = (size_t)e.splitk * M * a->Cout * sizeof(float);
        if (!a->workspace || a->workspace_bytes < need) e.splitk = 1;
      }
    }
  }
  if (kVariants[e.vid].ver >= 2 && kVariants[e.vid].ver != 6 && !v2_ok) e.vid = kVariants[e.vid].bm == 128 ? (kVariants[e.vid].bn >= 128 ? 0 : 1) : 3;
  e.fused = 0;
  {
    // in-kernel split-K reduction: every workgroup of the launch must be resident (tiles * splitk <= CUs), the slice
    // count must divide the tile's rows, and a slice's row range must be whole statistics segments
    // OFF by default: measured in the step it LOSES to the two-launch form (batch 64: 6.10 vs 5.54 ms/step, batch 8:
    // 3.26 vs 2.95, batch 1: 2.66 vs 2.55 - profiles/r02/fused_splitk_ab.txt): the 256 workgroups of the launch read
    // their 98 KB of slabs with far less memory-level parallelism than the 1536-workgroup reduction kernel, the
    // write-through stores are slower than plain ones, and every slice waits for the slowest.  The guide's verdict
    // for this seam (cut it) holds here; afldm_conv2d_fused_splitk(1) / AFLDM_FUSED_SPLITK=1 re-enable it.
    static const bool env_on = getenv("AFLDM_FUSED_SPLITK") && atoi(getenv("AFLDM_FUSED_SPLITK")) != 0;
    const Variant& v = kVariants[e.vid];
    const int HW = a->H * a->W, z = e.splitk;
    if ((env_on || g_fused_splitk) && z > 1 && v.ver >= 2 && v.ver <= 4 && a->sync && a->sync_bytes >= 40960 && (z == 2 || z == 4 || z == 8) &&
        v.bm % z == 0 && a->out_mode == 0 && !a->y2 && a->Cout % 4 == 0 && a->y_ld % 4 == 0 &&
        (!a->residual || a->res_ld % 4 == 0) && (!a->temb || a->temb_stride % 4 == 0) &&
        (size_t)z * M * a->Cout * 4 < ((size_t)1 << 31)) {
      const int rw = v.bm / z;
      const long long tiles = ((M + v.bm - 1) / v.bm) * ((a->Cout + v.bn - 1) / v.bn);
      const bool seg_ok = rw >= HW ? (rw % HW == 0) : (HW % rw == 0 && (HW % v.bm == 0 || v.bm % HW == 0));
      if (seg_ok && tiles <= 4096 && tiles * z <= device_cus()) e.fused = 1;
    }
  }
  return e;
}

// weights-in-registers kernel for the short-K attention projections (lin.hip)
int lin_wreg_bm(const afldm_conv_args* a);
int lin_wreg_launch(const afldm_conv_args* a, hipStream_t st);
// operands-straight-to-registers kernel for 1x1 convolutions / dense layers over few rows (skinny.hip)
int skinny_stats_splits(const afldm_conv_args* a);      // 0: does not apply; else the statistics splits of its output
int skinny_launch(const afldm_conv_args* a, hipStream_t st);

// conv_in on MFMA (k_conv_cin4_mfma): bf16, Cin = 4, 3x3, whole 128-pixel blocks inside one sample
template <typename T>
static bool cin4_mfma_ok(const afldm_conv_args* a) {
  static const bool off = getenv("AFLDM_NO_CIN4_MFMA") && atoi(getenv("AFLDM_NO_CIN4_MFMA")) != 0;
  return sizeof(T) == 2 && !off && a->C1 == 4 && a->C2 == 0 && a->KS == 3 && a->Cout % 16 == 0 && a->Cout <= 512 &&
         !a->temb && !a->residual && a->out_mode == 0 && !a->y2 && a->y_ld % 8 == 0 && (a->H * a->W) % CIN4_BM == 0 &&
         aligned16(a->y) && (reinterpret_cast<uintptr_t>(a->x1) & 7) == 0 && (reinterpret_cast<uintptr_t>(a->w) & 7) == 0;
}

// Where the GroupNorm partial sums of the output come from, and their split count S.
enum { ST_EPILOGUE = 1, ST_REDUCE = 2, ST_STANDALONE = 3, ST_FUSED = 4 };
static int reduce_stats_splits(int HW) { return HW >= 64 ? (HW / 16 > 32 ? 32 : HW / 16) : 1; }
template <typename T>
static int stats_mode(const afldm_conv_args* a, const Exec& e, int* S) {
  const int HW = a->H * a->W, eo = 16 / (int)sizeof(T);
  const bool vec4 = a->y_ld % 4 == 0 && (!a->residual || a->res_ld % 4 == 0) && (!a->temb || a->temb_stride % 4 == 0) &&
                    a->Cout % 4 == 0;
  const bool vec16 = a->y_ld % eo == 0 && (!a->residual || a->res_ld % eo == 0) && (!a->temb || a->temb_stride % eo == 0) &&
                     a->Cout % eo == 0;
  if (const int sk = skinny_stats_splits(a)) {
    *S = sk;
    return ST_EPILOGUE;
  }
  if (e.pl.kind == 0 && e.splitk > 1 && e.fused) {
    const int rw = kVariants[e.vid].bm / e.splitk;
    *S = rw >= HW ? 1 : HW / rw;
    return ST_FUSED;
  }
  if (e.pl.kind == 0 && e.splitk > 1 && vec4) {
    *S = reduce_stats_splits(HW);
    return ST_REDUCE;
  }
  if (e.pl.kind == 0 && e.splitk == 1 && e.vid == 40) {
    *S = (HW / 128) * 4;      // one split per (consumer row half, 32-row pass)
    return ST_EPILOGUE;
  }
  if (e.pl.kind == 0 && e.splitk == 1 && kVariants[e.vid].ver >= 2 && HW % kVariants[e.vid].bm == 0 && vec16 &&
      !getenv("AFLDM_CONV_NOSTAGE")) {
    *S = HW / kVariants[e.vid].bm;
    return ST_EPILOGUE;
  }
  if (sizeof(T) == 2 && e.pl.kind == 0 && e.splitk == 1 && kVariants[e.vid].ver == 6 && kVariants[e.vid].bm % HW == 0 && vec16) {
    *S = 1;           // halo-patch tiles of several whole samples (4x4 planes): one record per sample from the epilogue
    return ST_EPILOGUE;
  }
  if (e.pl.kind == 0 && e.splitk == 1 && kVariants[e.vid].ver >= 2 && kVariants[e.vid].ver <= 4 && HW == 1 && vec16 &&
      !getenv("AFLDM_CONV_NOSTAGE")) {
    *S = 1;           // H * W == 1: a row is a sample, written straight from the epilogue (stats_multi 2)
    return ST_EPILOGUE;
  }
  if (cin4_mfma_ok<T>(a)) {
    *S = HW / CIN4_BM;
    return ST_EPILOGUE;
  }
  // 64x64 tiles over planes smaller than a tile (to_out at the 4x4 / 2x2 levels): the tile holds 64 / HW whole
  // samples and a thread's two rows stay inside one of them (bf16: 32 row lanes x 2 rows, one staging pass)
  if (sizeof(T) == 2 && e.pl.kind == 0 && e.splitk == 1 && e.vid == 32 && HW < 64 && 64 % HW == 0 && HW % 2 == 0 && vec16 &&
      !getenv("AFLDM_CONV_NOSTAGE") && !getenv("AFLDM_NO_MULTI_STATS")) {
    *S = 1;
    return ST_EPILOGUE;
  }
  *S = gn_splits(HW);
  return ST_STANDALONE;
}

extern "C" int afldm_gn_stats(const void* x, int C, float* stats, int B, int HW, int dtype, afldm_stream_t stream);

// the epilogue of a halo-patch tile that is one whole 8x8 sample x 96 couts (variant 51, bf16, whole K, statistics from the
// epilogue) can apply the GroupNorm that follows: its couts must be whole groups
template <typename T>
static bool norm_fusable(const afldm_conv_args* a, const Exec& e, int smode) {
  if (sizeof(T) != 2 || e.pl.kind != 0 || e.splitk != 1 || smode != ST_EPILOGUE || !a->stats_out) return false;
  if (e.vid != 51 && e.vid != 53) return false;
  if (a->H != 8 || a->W != 8 || a->out_mode != 0 || a->y2 || a->y_ld != a->Cout) return false;
  if (a->norm_groups <= 0 || a->Cout % a->norm_groups || !a->norm_gamma || !a->norm_beta) return false;
  const int cpg = a->Cout / a->norm_groups;
  return 96 % cpg == 0;
}

// the argument block of every GEMM-path kernel, from the caller's arguments (plan-independent part)
template <typename T>
static void fill_convp(const afldm_conv_args* a, ConvP& p) {
  p.x1 = a->x1; p.x2 = a->x2; p.w = a->w; p.bias = a->bias; p.temb = a->temb; p.residual = a->residual;
  p.y = a->y; p.ws = (float*)a->workspace;
  p.y2 = a->y2; p.split_n = a->split_n;
  p.C1 = a->C1; p.C2 = a->C2; p.B = a->B; p.H = a->H; p.W = a->W; p.Cout = a->Cout; p.KS = a->KS;
  p.temb_stride = a->temb_stride; p.res_ld = a->res_ld; p.y_ld = a->y_ld; p.out_mode = a->out_mode;
  p.temb_mod = a->temb_mod > 0 ? a->temb_mod : a->Cout;
  p.M = a->B * a->H * a->W;
  p.splitk = 1; p.tiles_n = 1; p.ksteps = 0; p.sync = nullptr;
  p.w_bstride = a->w_batch_stride;
  static const int s_dbg = getenv("AFLDM_CONV_DBG") ? atoi(getenv("AFLDM_CONV_DBG")) : 0;
  p.dbg = s_dbg;
  static const int s_tapin = getenv("AFLDM_CONV_TAPINNER") ? atoi(getenv("AFLDM_CONV_TAPINNER")) : 0;
  p.tap_inner = s_tapin;
  p.vec_ok = ((a->out_mode == 1 || a->y_ld % 4 == 0) && (!a->residual || a->res_ld % 4 == 0) &&
              (!a->temb || a->temb_stride % 4 == 0)) ? 1 : 0;
  {
    // 16-byte epilogue accesses: every leading dimension a multiple of the 16-byte element count,
    // channel-major outputs need whole 16-byte pixel runs inside one sample
    const int eo = 16 / (int)sizeof(T);
    const int hw = a->H * a->W;
    const bool cm = a->out_mode == 1 || a->y2;
    p.stage_ok = ((a->out_mode == 1 || a->y_ld % eo == 0) && (!a->residual || a->res_ld % eo == 0) &&
                  (!a->temb || a->temb_stride % eo == 0) && (!cm || hw % eo == 0) && (!a->y2 || a->split_n % eo == 0) &&
                  !(cm && a->residual)) ? 1 : 0;
    static const int s_nostage = getenv("AFLDM_CONV_NOSTAGE") ? atoi(getenv("AFLDM_CONV_NOSTAGE")) : 0;
    if (s_nostage) p.stage_ok = 0;
  }
  {
    static const int s_mfast = getenv("AFLDM_CONV_MFAST") ? atoi(getenv("AFLDM_CONV_MFAST")) : -1;
    p.m_fast = s_mfast >= 0 ? s_mfast : ((long long)a->Cout * a->KS * a->KS > (long long)p.M ? 1 : 0);
  }
  p.xcd_gn = 0;
  p.y_norm = nullptr; p.ngamma = nullptr; p.nbeta = nullptr; p.ncpg = 1; p.neps = 0.f;
  p.x_c8 = a->x_layout == 1 ? 1 : 0;
  p.y_c8 = a->y_layout == 1 ? 1 : 0;
  p.w_nt = 0;       // set per plan (conv_dispatch: a single row tile)
}

// 8-channel-block operands (afldm_conv_args.x_layout / y_layout = 1): only where afldm_conv2d is ONE halo-patch launch with the
// whole K per workgroup, bf16, tiles inside one sample and the one-pass bf16 epilogue (no second output, NHWC residual is fine)
template <typename T>
static bool c8_ok(const afldm_conv_args* a) {
  if (sizeof(T) != 2 || a->C2 != 0 || a->x2 || a->y2 || a->out_mode != 0 || a->y_norm || a->y_ld != a->Cout || a->Cout % 8 || a->C1 % 64) return false;
  // (the conditions of plan_h3 below: one k_conv3h launch, statistics - if any - from its epilogue; callers pass whole batches:
  //  afldm_conv2d_c8_ok checks the batch chunking)
  if (a->w_batch_stride || a->defer_reduce || a->KS != 3 || lin_wreg_bm(a) || skinny_stats_splits(a)) return false;
  const Exec ex = resolve_exec<T>(a);
  if (ex.pl.kind != 0 || kVariants[ex.vid].ver != 6 || ex.splitk != 1 || ex.fused) return false;
  if (a->stats_out) {
    int S = 0;
    if (stats_mode<T>(a, ex, &S) != ST_EPILOGUE) return false;
  }
  return a->H == a->W && a->W <= 32 && a->H * a->W >= kVariants[ex.vid].bm;
}

template <typename T>
static int conv_dispatch(const afldm_conv_args* a, hipStream_t st) {
  AFLDM_REQUIRE((a->x_layout == 0 && a->y_layout == 0) || c8_ok<T>(a), AFLDM_ESHAPE,
                "afldm_conv2d: x_layout / y_layout = 1 (8-channel blocks) is not available for this problem (afldm_conv2d_c8_ok)");
  ConvP p;
  fill_convp<T>(a, p);
  const Exec ex = resolve_exec<T>(a);
  const Plan& pl = ex.pl;
  int smode = 0;
  p.stats_out = nullptr;
  p.stats_S = 1;
  p.stats_multi = 0;
  if (a->stats_out) smode = stats_mode<T>(a, ex, &p.stats_S);
  int rc = AFLDM_OK;
  AFLDM_REQUIRE(!a->w_batch_stride || pl.kind == 0, AFLDM_ESHAPE, "afldm_conv2d: w_batch_stride needs the GEMM path (Cin a multiple of the K step)");
  if (lin_wreg_bm(a)) return lin_wreg_launch(a, st);
  if (skinny_stats_splits(a)) return skinny_launch(a, st);
  if (pl.kind == 1 && cin4_mfma_ok<T>(a)) {
    if (smode == ST_EPILOGUE) p.stats_out = a->stats_out;
    const int lds = CIN4_BM * (a->Cout + 8) * 2 + 4 * a->Cout * 2 * (int)sizeof(float);
    static unsigned long long attr_set = 0;
    if (first_on_device(attr_set)) {
      (void)hipFuncSetAttribute((const void*)k_conv_cin4_mfma, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    k_conv_cin4_mfma<<<(p.M + CIN4_BM - 1) / CIN4_BM, 256, lds, st>>>(p);
    rc = check_launch("afldm_conv2d(cin4_mfma)");
  } else if (pl.kind == 1 && a->C1 == 4 && a->C2 == 0 && a->KS == 3 && a->Cout % 16 == 0 && a->Cout * 36 * 4 <= 64 * 1024 &&
      !a->temb && !a->residual && a->out_mode == 0 && a->y_ld % 4 == 0) {
    const int lds = a->Cout * 36 * (int)sizeof(float);
    k_conv_cin4<T, 4, 3><<<(p.M + 63) / 64, 256, lds, st>>>(p);
    rc = check_launch("afldm_conv2d(cin4)");
  } else if (pl.kind == 1) {
    AFLDM_REQUIRE(a->C2 == 0 && a->C1 <= 64, AFLDM_ESHAPE,
                  "afldm_conv2d: Cin=%d+%d is not a multiple of %d and too large for the direct kernel", a->C1,
                  a->C2, KCH_DEFAULT * epr<T>());
    size_t total = (size_t)p.M * p.Cout;
    int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    k_conv_small_cin<T><<<grid, 256, 0, st>>>(p);
    rc = check_launch("afldm_conv2d(small_cin)");
  } else if (pl.kind == 2) {
    AFLDM_REQUIRE(a->C2 == 0 && a->Cout <= 8, AFLDM_ESHAPE, "afldm_conv2d: unsupported small-Cout shape (Cout=%d, C2=%d)",
                  a->Cout, a->C2);
    int grid = (p.M + 3) / 4 < 8192 ? (p.M + 3) / 4 : 8192;
    k_conv_small_cout<T, 8><<<grid, 256, 0, st>>>(p);
    rc = check_launch("afldm_conv2d(small_cout)");
  } else {
    const int Ct = a->C1 + a->C2;
    p.ksteps = a->KS * a->KS * (Ct / (KCH_DEFAULT * epr<T>()));
    p.splitk = ex.splitk;
    if (smode == ST_EPILOGUE) {
      p.stats_out = a->stats_out;
      p.stats_multi = (a->H * a->W) == 1 ? 2 : (a->H * a->W) < kVariants[ex.vid].bm ? 1 : 0;
    }
    p.sync = ex.fused ? a->sync : nullptr;
    if (ex.fused && smode == ST_FUSED) p.stats_out = a->stats_out;
    if (a->y_norm) {
      AFLDM_REQUIRE(norm_fusable<T>(a, ex, smode), AFLDM_ESHAPE, "afldm_conv2d: y_norm given but this problem's epilogue cannot apply the GroupNorm (afldm_conv2d_norm_ok)");
      p.y_norm = a->y_norm; p.ngamma = a->norm_gamma; p.nbeta = a->norm_beta;
      p.ncpg = a->Cout / a->norm_groups; p.neps = a->norm_eps;
    }
    if (a->w_batch_stride) {      // per-sample weights: the LDS-DMA GEMM only, whole tiles inside a sample, K not split
      const int ver = kVariants[ex.vid].ver;
      AFLDM_REQUIRE(ver >= 2 && ver <= 4 && p.splitk == 1 && (a->H * a->W) % kVariants[ex.vid].bm == 0, AFLDM_ESHAPE,
                    "afldm_conv2d: w_batch_stride needs H*W a multiple of the %d-row tile of variant %d and no split-K",
                    kVariants[ex.vid].bm, ex.vid);
    }
    {
      // weights that exactly one workgroup per (cout tile, K slice) reads - a single row tile - stream non-temporally
      static const bool s_nt = !(getenv("AFLDM_NT_WEIGHTS") && atoi(getenv("AFLDM_NT_WEIGHTS")) == 0);
      p.w_nt = (s_nt && !a->w_batch_stride && p.M <= kVariants[ex.vid].bm) ? 1 : 0;
    }
    launch_variant<T>(ex.vid, p, st);
    rc = check_launch("afldm_conv2d(igemm)");
    if (rc) return rc;
    if (p.splitk > 1 && !ex.fused && a->defer_reduce) return AFLDM_OK;      // the caller's consumer finishes the slabs (afldm_af_act_slabs)
    if (p.splitk > 1 && !ex.fused) {
      if (smode == ST_REDUCE) {
        p.stats_out = a->stats_out;
        const int HW = a->H * a->W;
        const int rows = (HW + p.stats_S - 1) / p.stats_S;
        if (rows >= 16) {
          const int nqb = (p.Cout / 4 + 15) / 16;
          k_splitk_reduce_stats<T, 16><<<a->B * p.stats_S * nqb, 256, 0, st>>>(p, rows);
        } else {
          const int nqb = (p.Cout / 4 + 63) / 64;
          k_splitk_reduce_stats<T, 4><<<a->B * p.stats_S * nqb, 256, 0, st>>>(p, rows);
        }
      } else {
        size_t total = (size_t)p.M * ((p.Cout + 3) / 4);
        int grid = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
        k_splitk_reduce<T><<<grid, 256, 0, st>>>(p);
      }
      rc = check_launch("afldm_conv2d(splitk_reduce)");
    }
  }
  if (rc) return rc;
  if (smode == ST_STANDALONE)   // output of a kernel without a fused producer: one extra read of y
    rc = afldm_gn_stats(a->y, a->Cout, a->stats_out, a->B, a->H * a->W, a->dtype, (afldm_stream_t)st);
  return rc;
}

static int conv_validate(const afldm_conv_args* a) {
  AFLDM_REQUIRE(a != nullptr, AFLDM_ENULL, "afldm_conv2d: args is NULL");
  AFLDM_REQUIRE(a->x1 && a->w && a->y, AFLDM_ENULL, "afldm_conv2d: x1/w/y must be non-NULL");
  AFLDM_REQUIRE(a->KS == 1 || a->KS == 3, AFLDM_ESHAPE, "afldm_conv2d: KS=%d not in {1,3}", a->KS);
  AFLDM_REQUIRE(a->B > 0 && a->H > 0 && a->W > 0 && a->Cout > 0 && a->C1 > 0 && a->C2 >= 0, AFLDM_ESHAPE,
                "afldm_conv2d: bad shape B=%d H=%d W=%d Cout=%d C1=%d C2=%d", a->B, a->H, a->W, a->Cout, a->C1, a->C2);
  AFLDM_REQUIRE(a->C2 == 0 || a->x2 != nullptr, AFLDM_ENULL, "afldm_conv2d: C2>0 but x2 is NULL");
  AFLDM_REQUIRE(a->out_mode == 0 || a->out_mode == 1, AFLDM_ESHAPE, "afldm_conv2d: out_mode %d", a->out_mode);
  AFLDM_REQUIRE(!a->y2 || (a->split_n > 0 && a->split_n % 4 == 0 && a->split_n < a->Cout && a->out_mode == 0 &&
                           !a->temb && !a->residual),
                AFLDM_ESHAPE, "afldm_conv2d: y2 needs 0 < split_n < Cout, split_n %% 4 == 0, out_mode 0, no temb/residual");
  AFLDM_REQUIRE(a->out_mode == 1 || a->y_ld >= (a->y2 ? a->split_n : a->Cout), AFLDM_ESHAPE,
                "afldm_conv2d: y_ld=%d too small for Cout=%d", a->y_ld, a->Cout);
  AFLDM_REQUIRE(!a->residual || a->res_ld >= a->Cout, AFLDM_ESHAPE, "afldm_conv2d: res_ld=%d < Cout=%d", a->res_ld, a->Cout);
  AFLDM_REQUIRE((long long)a->B * a->H * a->W < (1ll << 30), AFLDM_ESHAPE, "afldm_conv2d: M too large");
  AFLDM_REQUIRE(a->w_batch_stride == 0 || (a->w_batch_stride > 0 && a->KS == 1 && a->out_mode == 0 && !a->y2 && !a->stats_out &&
                                            a->C2 == 0 && a->w_batch_stride % 8 == 0 && !a->defer_reduce),
                AFLDM_ESHAPE, "afldm_conv2d: w_batch_stride needs KS = 1, out_mode 0, no y2 / stats / concat, stride %% 8 == 0");
  AFLDM_REQUIRE(a->temb_mod == 0 || (a->temb_mod > 0 && a->temb_mod % 8 == 0 && a->Cout % a->temb_mod == 0), AFLDM_ESHAPE,
                "afldm_conv2d: temb_mod=%d must divide Cout=%d and be a multiple of 8", a->temb_mod, a->Cout);
  AFLDM_REQUIRE(!a->stats_out || (a->out_mode == 0 && !a->y2 && a->y_ld == a->Cout && a->Cout % 4 == 0), AFLDM_ESHAPE,
                "afldm_conv2d: stats_out needs a dense NHWC output (out_mode 0, no y2, y_ld == Cout, Cout %% 4 == 0)");
  AFLDM_REQUIRE(aligned16(a->x1) && aligned16(a->w) && aligned16(a->y) && aligned16(a->x2) && aligned16(a->residual) &&
                    aligned16(a->temb) && aligned16(a->bias),
                AFLDM_EALIGN, "afldm_conv2d: all pointers must be 16-byte aligned");
  return AFLDM_OK;
}

}  // namespace afldm

using namespace afldm;

extern "C" int afldm_conv2d_tune(int variant, int splitk) {
  AFLDM_REQUIRE(variant < kNumVariants, AFLDM_ESHAPE, "afldm_conv2d_tune: variant %d out of range (%d)", variant, kNumVariants);
  g_force_variant = variant;
  g_force_splitk = splitk;
  return AFLDM_OK;
}

extern "C" int afldm_conv2d_fused_splitk(int enable) {
  g_fused_splitk = enable ? 1 : 0;
  return AFLDM_OK;
}

// Batch chunking.  The LDS-DMA kernels address their pixel operand through a buffer descriptor (32-bit byte offsets, 2 GiB):
// the AF-VAE's 256^2 levels at batch 128 (128 x 65536 pixels x 128 channels x 2 B = 2.1 GB) used to fall back to the
// round-1 register-staged kernel at 0.43 PFLOP/s for that reason alone (profiles/r03/r03a_vae_kernel_stats.csv: 28 % of
// the C4 workload).  Samples are independent, so such a call is issued as B / c launches over c whole samples each
// (c the largest divisor of B whose operand fits), every one on the fast kernels; the queries below answer for the
// chunk, so plan, statistics splits and workspace agree with what is launched.
static int conv_batch_chunk(const afldm_conv_args* a) {
  const long long esz = a->dtype == AFLDM_F32 ? 4 : 2;
  const long long per = (long long)a->H * a->W * (a->C1 > a->C2 ? a->C1 : a->C2) * esz;
  if (a->B <= 1 || (long long)a->B * per < (1ll << 31) || per >= (1ll << 31)) return a->B;
  int c = (int)(((1ll << 31) - 1) / per);
  while (c > 1 && a->B % c) --c;
  return c;
}

static afldm_conv_args conv_chunk_args(const afldm_conv_args* a, int c) {
  afldm_conv_args q = *a;
  q.B = c;
  return q;
}

// conv3h_plan (conv_common.hpp): true when afldm_conv2d would run this problem as ONE halo-patch launch (k_conv3h, whole
// K per workgroup, statistics - if asked for - from its epilogue); then *p is exactly the argument block that launch
// takes and *variant its id.  The merged launches of actconv.hip run that tile as their last phase.
template <typename T>
static bool plan_h3(const afldm_conv_args* a, ConvP& p, int& vid) {
  if (conv_batch_chunk(a) != a->B || a->w_batch_stride || a->defer_reduce || a->KS != 3) return false;
  if (lin_wreg_bm(a) || skinny_stats_splits(a)) return false;
  fill_convp<T>(a, p);
  const Exec ex = resolve_exec<T>(a);
  if (ex.pl.kind != 0 || kVariants[ex.vid].ver != 6 || ex.splitk != 1 || ex.fused) return false;
  p.stats_out = nullptr;
  p.stats_S = 1;
  p.stats_multi = 0;
  if (a->stats_out) {
    if (stats_mode<T>(a, ex, &p.stats_S) != ST_EPILOGUE) return false;
    p.stats_out = a->stats_out;
    p.stats_multi = (a->H * a->W) == 1 ? 2 : (a->H * a->W) < kVariants[ex.vid].bm ? 1 : 0;
  }
  p.ksteps = 9 * ((a->C1 + a->C2) / (KCH_DEFAULT * epr<T>()));
  p.splitk = 1;
  p.sync = nullptr;
  vid = ex.vid;
  return true;
}
namespace afldm {
bool conv3h_plan(const afldm_conv_args* a, ConvP* p, int* variant) {
  if (!a || conv_validate(a)) return false;
  if (a->dtype == AFLDM_BF16) return plan_h3<bf16>(a, *p, *variant);
  if (a->dtype == AFLDM_F32) return plan_h3<float>(a, *p, *variant);
  return false;
}
}  // namespace afldm

extern "C" int afldm_conv2d_variant(const afldm_conv_args* a0) {
  if (!a0 || (a0->dtype != AFLDM_F32 && a0->dtype != AFLDM_BF16)) return -1;
  const afldm_conv_args ac = conv_chunk_args(a0, conv_batch_chunk(a0));
  const afldm_conv_args* a = &ac;
  const Exec e = a->dtype == AFLDM_F32 ? resolve_exec<float>(a) : resolve_exec<bf16>(a);
  if (skinny_stats_splits(a)) return -16;        // skinny.hip
  if (e.pl.kind != 0) return -1 - e.pl.kind;
  return e.vid | (e.splitk << 8) | (e.fused << 16);
}

extern "C" int afldm_conv2d_c8_ok(const afldm_conv_args* a) {
  if (!a || a->dtype != AFLDM_BF16 || conv_validate(a) || conv_batch_chunk(a) != a->B) return 0;
  return c8_ok<bf16>(a) ? 1 : 0;
}

extern "C" int afldm_conv2d_norm_ok(const afldm_conv_args* a) {
  if (!a || a->dtype != AFLDM_BF16 || conv_batch_chunk(a) != a->B || !a->stats_out) return 0;
  if (lin_wreg_bm(a) || skinny_stats_splits(a)) return 0;
  const Exec e = resolve_exec<bf16>(a);
  int S = 0;
  const int smode = stats_mode<bf16>(a, e, &S);
  return norm_fusable<bf16>(a, e, smode) ? 1 : 0;
}

extern "C" int afldm_conv2d_stats_splits(const afldm_conv_args* a0) {
  if (!a0 || (a0->dtype != AFLDM_F32 && a0->dtype != AFLDM_BF16)) return 0;
  const afldm_conv_args ac = conv_chunk_args(a0, conv_batch_chunk(a0));
  const afldm_conv_args* a = &ac;
  int S = 0;
  if (a->dtype == AFLDM_F32) stats_mode<float>(a, resolve_exec<float>(a), &S);
  else stats_mode<bf16>(a, resolve_exec<bf16>(a), &S);
  return S;
}

extern "C" size_t afldm_conv2d_workspace(const afldm_conv_args* a0) {
  if (!a0 || (a0->dtype != AFLDM_F32 && a0->dtype != AFLDM_BF16)) return 0;
  const afldm_conv_args ac = conv_chunk_args(a0, conv_batch_chunk(a0));
  const afldm_conv_args* a = &ac;
  if (lin_wreg_bm(a) || skinny_stats_splits(a)) return 0;
  Plan pl = make_plan(a, a->dtype == AFLDM_F32 ? 16 : 32);
  if (pl.kind != 0 || pl.splitk <= 1) return 0;
  return (size_t)pl.splitk * a->B * a->H * a->W * a->Cout * sizeof(float);
}

extern "C" int afldm_conv2d(const afldm_conv_args* a, afldm_stream_t stream) {
  int rc = conv_validate(a);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  if (a->dtype != AFLDM_F32 && a->dtype != AFLDM_BF16) {
    set_error("afldm_conv2d: unknown dtype %d", a->dtype);
    return AFLDM_EDTYPE;
  }
  const int c = conv_batch_chunk(a);
  if (c == a->B) return a->dtype == AFLDM_F32 ? conv_dispatch<float>(a, st) : conv_dispatch<bf16>(a, st);
  // whole-sample chunks (see conv_batch_chunk): every per-sample operand advances by its own stride
  const size_t esz = a->dtype == AFLDM_F32 ? 4 : 2;
  const size_t HW = (size_t)a->H * a->W;
  afldm_conv_args q = conv_chunk_args(a, c);
  const int S = a->stats_out ? afldm_conv2d_stats_splits(&q) : 0;
  for (int b0 = 0; b0 < a->B; b0 += c) {
    q.x1 = (const char*)a->x1 + (size_t)b0 * HW * a->C1 * esz;
    q.x2 = a->x2 ? (const char*)a->x2 + (size_t)b0 * HW * a->C2 * esz : nullptr;
    q.y = (char*)a->y + (size_t)b0 * (a->out_mode == 1 ? (size_t)a->Cout * HW : HW * a->y_ld) * esz;
    q.y2 = a->y2 ? (char*)a->y2 + (size_t)b0 * (size_t)(a->Cout - a->split_n) * HW * esz : nullptr;
    q.residual = a->residual ? (const char*)a->residual + (size_t)b0 * HW * a->res_ld * esz : nullptr;
    q.temb = a->temb ? (const char*)a->temb + (size_t)b0 * a->temb_stride * esz : nullptr;
    q.w = (const char*)a->w + (size_t)b0 * (size_t)a->w_batch_stride * esz;
    q.stats_out = a->stats_out ? a->stats_out + (size_t)b0 * S * a->Cout * 2 : nullptr;
    rc = a->dtype == AFLDM_F32 ? conv_dispatch<float>(&q, st) : conv_dispatch<bf16>(&q, st);
    if (rc) return rc;
  }
  return AFLDM_OK;
}
