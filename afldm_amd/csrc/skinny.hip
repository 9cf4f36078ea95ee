// skinny.hip — 1x1 convolutions / dense layers over FEW rows (M = B*H*W <= 1024, policy below) straight from global memory to MFMA
// operand registers (gfx950).
//
// The 3x3 convolutions of the 2x2 level run as dense layers over the flattened plane (64 rows x 3072 at batch 64,
// 19-38 MB of weights), and the 1x1 convolutions of the 8x8 / 4x4 / 2x2 levels (to_out, shortcuts) have 256-4096
// rows and K = 384-1536.  On the LDS-DMA implicit GEMM (k_igemm2, 64x64 tiles) such a layer is a chain of K steps of
// ~0.5 us each (barrier round + LDS round trip + a handful of MFMAs: profiles/r02/small_tile_step_decomposition.txt)
// and needs split-K slabs plus a reduction launch to occupy the chip at all: 10.5 + 5.3 us per dense layer.
//
// Here nothing is staged: a workgroup owns 16 * (8 / NS) couts x (up to) 64 rows and ALL of K; its 8 waves are NS K
// slices (8, 4 or 2: the largest that divides K into whole MFMA steps) x 8 / NS cout tiles and load both operands as
// MFMA fragments directly (16 bytes per lane: lane (i, g) = row i, K chunk g -
// the weights' 16 rows x 64 B and x's 16 rows x 64 B per instruction; x is tiny and L2 resident, every weight byte
// is read once per 64-row block).  All loads of a group of K steps are independent and in flight together (no ring,
// no barrier in the K loop); the eight partial tiles meet once in LDS, the first waves add them in a fixed order
// and finish: + bias + time embedding + residual, one rounding, 8 / 16-byte stores, GroupNorm partial sums of the
// stored values (per sample; H*W == 1: the values themselves).  One launch, no workspace.
#include "common.hpp"

namespace afldm {

struct SkP {
  const void* x1;
  const void* x2;
  const void* w;
  const float* bias;
  const void* temb;
  const void* residual;
  void* y;
  float* stats;      // [B][S][N][2] or NULL
  int M, K, C1, N, HW, temb_stride, temb_mod, res_ld, y_ld, stats_S;
  int ns;            // K slices per workgroup (8 / ns cout tiles of 16)
  int w_nt;          // 1: one 64-row block - every weight byte is read exactly once: non-temporal loads (ConvP::w_nt)
};

constexpr int SK_WAVES = 8, SK_ROWS = 64, SK_MT = SK_ROWS / 16, SK_U = 4;

// MTT: row tiles a workgroup can hold (4, or 1 for launches of <= 16 rows); U: K steps per register group - the
// one-tile form fetches 12 steps at once (a 3072-wide K slice of 384 = ONE load round trip instead of two or three;
// at batch 1 a dense layer is launch + load latency and little else).
template <typename T, int MTT, int U, bool WNT = false>
__global__ void __launch_bounds__(SK_WAVES * 64) k_skinny(SkP p) {
  typedef Mma<T> MM;
  typedef typename MM::Chunk Chunk;
  constexpr int EPC = MM::EPC, KPF = MM::KPF;
  __shared__ f32x4 sAcc[SK_WAVES][MTT][64];
  __shared__ float sOut[SK_WAVES / 2][SK_ROWS][16 + 1];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const int NS = p.ns, NT = SK_WAVES / NS;               // K slices, cout tiles of this workgroup
  const int wks = wave % NS, wnt = wave / NS;
  const int nblk = blockIdx.x * 16 * NT, m0 = blockIdx.y * SK_ROWS;
  const int n0 = nblk + wnt * 16;                        // this wave's cout tile in the K loop
  const int KW = p.K / NS, nks = KW / KPF;               // this wave's K slice and its fragment steps
  const int kbase = wks * KW;
  // operand bases (elements); a K slice lies inside one of the two inputs (host: C1 % KW == 0 when there are two)
  const bool second = kbase >= p.C1;
  const T* xs = second ? (const T*)p.x2 : (const T*)p.x1;
  const int ldx = second ? p.K - p.C1 : p.C1;
  const int kx = (second ? kbase - p.C1 : kbase) + lg * EPC;
  const T* wrow = (const T*)p.w + (size_t)(n0 + li) * p.K + kbase + lg * EPC;
  const T* xrow[MTT];
#pragma unroll
  for (int mt = 0; mt < MTT; ++mt) {
    int row = m0 + mt * 16 + li;
    row = row < p.M ? row : p.M - 1;                      // rows past the end repeat the last one (never stored)
    xrow[mt] = xs + (size_t)row * ldx + kx;
  }
  f32x4 acc[MTT];
#pragma unroll
  for (int mt = 0; mt < MTT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int mtn = (p.M - m0 + 15) / 16;                  // row tiles that exist in this block (wave-uniform): absent ones cost no loads

  // groups of U K steps, double buffered in registers: the loads of group g + 1 are issued before group g's MFMAs
  Chunk a[2][U], b[2][U][MTT];
  auto load_group = [&](int buf, int ks0) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (ks0 + u < nks) {                                // (wave-uniform)
        const int off = (ks0 + u) * KPF;
#if defined(AFLDM_SK_NOW)                                 // (timing decomposition builds: no weight / no x loads; garbage results)
        a[buf][u] = ld16<Chunk>(xrow[0] + off);
#else
        if constexpr (WNT) a[buf][u] = __builtin_nontemporal_load(reinterpret_cast<const Chunk*>(wrow + off));
        else a[buf][u] = ld16<Chunk>(wrow + off);
#endif
#pragma unroll
        for (int mt = 0; mt < MTT; ++mt) {
          if (mt < mtn) {
#if defined(AFLDM_SK_NOX)
            b[buf][u][mt] = a[buf][u];
#else
            b[buf][u][mt] = ld16<Chunk>(xrow[mt] + off);
#endif
          }
        }
      }
    }
  };
  auto mma_group = [&](int buf, int ks0) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (ks0 + u < nks) {
#pragma unroll
        for (int mt = 0; mt < MTT; ++mt)
          if (mt < mtn) MM::mma(acc[mt], a[buf][u], b[buf][u][mt]);
      }
    }
  };
  load_group(0, 0);
  for (int ks0 = 0; ks0 < nks; ks0 += 2 * U) {
    load_group(1, ks0 + U);
    mma_group(0, ks0);
    load_group(0, ks0 + 2 * U);
    mma_group(1, ks0 + U);
  }

  // the K slices meet in LDS (fragment layout, 16 bytes per lane: conflict free)
#pragma unroll
  for (int mt = 0; mt < MTT; ++mt) sAcc[wave][mt][lane] = acc[mt];
  __syncthreads();
  const bool want_stats = p.stats != nullptr;
  for (int pi = wave; pi < NT * MTT; pi += SK_WAVES) {  // (cout tile, row tile) pairs, fixed summation order
    const int nt = pi / MTT, mt = pi - nt * MTT;
    f32x4 v = sAcc[nt * NS][mt][lane];
    for (int s = 1; s < NS; ++s) v += sAcc[nt * NS + s][mt][lane];
    const int row = m0 + mt * 16 + li, n = nblk + nt * 16 + 4 * lg;
    float o[4] = {0.f, 0.f, 0.f, 0.f};
    if (row < p.M) {
      if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + n);
      if (p.temb) {
        float t0, t1, t2, t3;
        load4<T>((const T*)p.temb + (size_t)(row / p.HW) * p.temb_stride + n % p.temb_mod, t0, t1, t2, t3);
        v[0] += t0; v[1] += t1; v[2] += t2; v[3] += t3;
      }
      if (p.residual) {
        float r0, r1, r2, r3;
        load4<T>((const T*)p.residual + (size_t)row * p.res_ld + n, r0, r1, r2, r3);
        v[0] += r0; v[1] += r1; v[2] += r2; v[3] += r3;
      }
      store4<T>((T*)p.y + (size_t)row * p.y_ld + n, v[0], v[1], v[2], v[3]);
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = to_f32(from_f32<T>(v[e]));      // statistics of what the consumer will read
      if (want_stats && p.HW == 1) {
        // a row is a sample of its own: its "sums" are the values themselves (S = 1)
        float* st = p.stats + ((size_t)row * p.N + n) * 2;
        *reinterpret_cast<f32x4*>(st) = f32x4{o[0], o[0] * o[0], o[1], o[1] * o[1]};
        *reinterpret_cast<f32x4*>(st + 4) = f32x4{o[2], o[2] * o[2], o[3], o[3] * o[3]};
      }
    }
    if (want_stats && p.HW > 1) {
#pragma unroll
      for (int e = 0; e < 4; ++e) sOut[nt][mt * 16 + li][4 * lg + e] = o[e];     // (rows past M: zeros)
    }
  }
  if (want_stats && p.HW > 1) {
    // per-channel sums over the rows of each sample segment inside this 64-row block (fixed order)
    __syncthreads();
    const int seg = p.HW < SK_ROWS ? p.HW : SK_ROWS;      // rows per segment: a whole sample, or this block's part of one
    const int nseg = SK_ROWS / seg;
    for (int t = threadIdx.x; t < NT * nseg * 16; t += SK_WAVES * 64) {
      const int nt = t / (nseg * 16), r = t - nt * nseg * 16, sg = r / 16, c = r - sg * 16;
      const int r0 = m0 + sg * seg;
      if (r0 < p.M) {
        float s1 = 0.f, s2 = 0.f;
        for (int q = 0; q < seg; ++q) {
          const float vr = sOut[nt][sg * seg + q][c];
          s1 += vr;
          s2 = fmaf(vr, vr, s2);
        }
        const int b = r0 / p.HW, sp = (r0 - b * p.HW) / seg;
        *reinterpret_cast<f32x2*>(p.stats + (((size_t)b * p.stats_S + sp) * p.N + nblk + nt * 16 + c) * 2) = f32x2{s1, s2};
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------- host side
static bool sk_aligned16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

// K slices per workgroup: the largest of 8 / 4 / 2 that cuts K (and a two-pointer input at its seam) into whole MFMA
// steps while Cout is a whole number of 16 * (8 / slices) cout blocks; 0: none.
static int skinny_slices(const afldm_conv_args* a) {
  const int K = a->C1 + a->C2, kpf = a->dtype == AFLDM_F32 ? 16 : 32;
  for (int ns = SK_WAVES; ns >= 2; ns /= 2) {
    if (K % (ns * kpf) || a->Cout % (16 * (SK_WAVES / ns))) continue;
    if (a->C2 > 0 && a->C1 % (K / ns)) continue;
    return ns;
  }
  return 0;
}

// Statistics splits of the skinny kernel's output for this call (0: the kernel does not apply).
int skinny_stats_splits(const afldm_conv_args* a) {
  static const bool off = getenv("AFLDM_NO_SKINNY") && atoi(getenv("AFLDM_NO_SKINNY")) != 0;
  if (off || a->KS != 1 || a->out_mode != 0 || a->y2 || a->w_batch_stride) return 0;
  const long long M = (long long)a->B * a->H * a->W;
  const int HW = a->H * a->W, K = a->C1 + a->C2;
  const int esz = a->dtype == AFLDM_F32 ? 4 : 2;
  static const int s_maxm = getenv("AFLDM_SKINNY_MAXM") ? atoi(getenv("AFLDM_SKINNY_MAXM")) : 1024;
  static const int s_maxx = getenv("AFLDM_SKINNY_MAXX") ? atoi(getenv("AFLDM_SKINNY_MAXX")) : 0;
  const int ns = skinny_slices(a);
  if (M > s_maxm || ns == 0) return 0;
  // Every workgroup gathers the x fragments of its (up to 64) rows for ALL of K as 64-byte pieces from L2 (~30 GB/s per
  // CU measured: rows K * esz bytes apart camp on one or two L2 channels).  Fine while that is small, slower than the
  // LDS-DMA GEMM beyond: batch 64, 64 rows x 3072 = 393 KB per workgroup: 16.8 us kernel-only against 10.5 + 5.3; with
  // several 64-row blocks (M >= 256) the break-even is lower (in-step A/B at batch 1 ... 64, profiles/r02/skinny_ab.txt).
  {
    const long long rows = M < SK_ROWS ? ((M + 15) / 16) * 16 : SK_ROWS;
    const long long lim = s_maxx > 0 ? s_maxx : (M >= 256 ? 64 * 1024 : 256 * 1024);
    if (rows * K * esz > lim) return 0;
  }
  if ((long long)a->Cout * K * esz >= (1ll << 31)) return 0;
  if (!sk_aligned16(a->x1) || !sk_aligned16(a->x2) || !sk_aligned16(a->w) || !sk_aligned16(a->y) || !sk_aligned16(a->residual) ||
      !sk_aligned16(a->temb) || !sk_aligned16(a->bias))
    return 0;
  // 4-element vector loads / stores along the channel axis of y, the residual and the time-embedding rows
  if ((a->y_ld & 3) || (a->res_ld & 3) || (a->temb_stride & 3)) return 0;
  // the GroupNorm partial sums are formed per 64-row block: a block must hold whole samples (64 % HW == 0) or a sample
  // whole blocks (HW % 64 == 0); 3x3 / 6x6 / 12x12 planes take the generic path + stand-alone statistics
  if (a->stats_out && !(HW <= SK_ROWS ? SK_ROWS % HW == 0 : HW % SK_ROWS == 0)) return 0;
  return HW <= SK_ROWS ? 1 : HW / SK_ROWS;
}

int skinny_launch(const afldm_conv_args* a, hipStream_t st) {
  SkP p;
  p.x1 = a->x1; p.x2 = a->x2; p.w = a->w; p.bias = a->bias; p.temb = a->temb; p.residual = a->residual; p.y = a->y;
  p.stats = a->stats_out;
  p.M = a->B * a->H * a->W; p.K = a->C1 + a->C2; p.C1 = a->C1; p.N = a->Cout; p.HW = a->H * a->W;
  p.temb_stride = a->temb_stride; p.temb_mod = a->temb_mod ? a->temb_mod : a->Cout; p.res_ld = a->res_ld; p.y_ld = a->y_ld;
  p.stats_S = skinny_stats_splits(a);
  p.ns = skinny_slices(a);
  static const bool s_nt = !(getenv("AFLDM_NT_WEIGHTS") && atoi(getenv("AFLDM_NT_WEIGHTS")) == 0);
  p.w_nt = (s_nt && p.M <= SK_ROWS) ? 1 : 0;
  const dim3 grid(p.N / (16 * (SK_WAVES / p.ns)), (p.M + SK_ROWS - 1) / SK_ROWS);
  if (p.M <= 16) {
    if (a->dtype == AFLDM_F32) k_skinny<float, 1, 12><<<grid, SK_WAVES * 64, 0, st>>>(p);
    else if (p.w_nt) k_skinny<bf16, 1, 12, true><<<grid, SK_WAVES * 64, 0, st>>>(p);
    else k_skinny<bf16, 1, 12><<<grid, SK_WAVES * 64, 0, st>>>(p);
  } else {
    if (a->dtype == AFLDM_F32) k_skinny<float, SK_MT, SK_U><<<grid, SK_WAVES * 64, 0, st>>>(p);
    else if (p.w_nt) k_skinny<bf16, SK_MT, SK_U, true><<<grid, SK_WAVES * 64, 0, st>>>(p);
    else k_skinny<bf16, SK_MT, SK_U><<<grid, SK_WAVES * 64, 0, st>>>(p);
  }
  return check_launch("afldm_conv2d(skinny)");
}

}  // namespace afldm
