// af.hip — the alias-free operators of AFLDM as dense separable circulant products.
//
// The reference applies its ideal filters in the FFT domain (rfft2 -> 0/0.5/1 mask -> irfft2,
// reference afldm/af_libs/ideal_lpf.py:69-158).  Those operators are EXACTLY  Y = U X U^T
// (x2 periodic-sinc upsample) and  Y = D Z D^T  (brick-wall low-pass + decimate) with dense
// matrices built from the reference's masks (afldm_filter_matrix); WarpedNonlinearity
// (reference af_blocks.py:19-28) is  D silu(U X U^T) D^T  per (sample, channel) plane.
//
// k_af_act_plane (N = 16, 32): one workgroup = one sample x 16 channels, NHWC; every wave carries
// whole channel planes through the four separable passes on MFMA with the upsampled 2N x 2N plane
// living only in accumulator registers (see the kernel).  HBM traffic is one read + one write of
// the tensor (the reference: ~30x that).  GroupNorm-apply is fused into the load.
//
// k_af_act_small (N = 2, 4, 8): one thread per (sample, channel) plane held in registers;
// lanes = channels, so the matrix coefficients are wave-uniform scalar operands.  HBM-bound.
//
// k_axis_contract: generic one-axis product used by the 4+4 AliasFreeUp/Downsample2D sites.
#include "af_plane.hpp"

namespace afldm {

static unsigned long long* g_af_trace = nullptr;

// Builds the constant-fragment image once (host calls it once per (N, dtype)).
template <typename T, int N>
__global__ void k_af_pack(const float* __restrict__ U, const float* __restrict__ D, T* __restrict__ out) {
  typedef PlaneCfg<T, N> CF;
  constexpr int EPC = CF::EPC, H2 = CF::H2;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < CF::CONST_ELEMS; i += gridDim.x * blockDim.x) {
    const int frag = i / (64 * EPC), lane = (i / EPC) % 64, e = i % EPC;
    const int li = lane & 15, g = lane >> 4;
    float v = 0.f;
    if (frag < CF::F_UPA) {
      const int th = frag / CF::NKF1, kf = frag % CF::NKF1, k = af_kidx<T>(kf, g, e, false);
      if (k < N) v = U[(16 * th + li) * N + k];
    } else if (frag < CF::F_DPA) {
      // (P2's matrix carries log2(e) and P3's ln(2): the SiLU in between works on z' = z log2(e),
      //  z sigmoid(z) = ln(2) z' / (1 + exp2(-z')), one multiply and the negation free)
      const int q = frag - CF::F_UPA, t2 = q / CF::NKF1, f = q % CF::NKF1, k = af_kidx<T>(f, g, e, true);
      if (k < N) v = U[(16 * t2 + li) * N + k] * 1.4426950408889634f;
    } else if (frag < CF::F_DA) {
      const int q = frag - CF::F_DPA, t3 = q / CF::NKF3, f = q % CF::NKF3, k = af_kidx<T>(f, g, e, true);
      if (k < H2) v = D[(16 * t3 + li) * H2 + k] * 0.6931471805599453f;
    } else {
      const int q = frag - CF::F_DA, t4 = q / CF::NKF3, kf = q % CF::NKF3, k = af_kidx<T>(kf, g, e, false);
      if (k < H2) v = D[(16 * t4 + li) * H2 + k];
    }
    out[i] = from_f32<T>(v);
  }
}

// k_af_act_plane: one workgroup = 4 waves, one item = one sample x 16 channels (NHWC, so the 16
// channels of a pixel are one 32/64-byte run).  The tile is transposed into per-channel planes
// Xs[c][w][h] in LDS (GroupNorm applied on the way); then EACH WAVE OWNS WHOLE CHANNEL PLANES and
// carries a plane through all four passes without meeting the other waves:
//   P1  T1^T[w][h'] = sum_h  X^T[w][h]  U[h'][h]      A = X^T rows from LDS, B = U (registers)
//   P2  Z^T [w'][h'] = sum_w  U[w'][w]   T1^T[w][h']   A = U (chain-permuted), B = P1's accumulator
//       SiLU on the accumulator
//   P3  V^T [w][h']  = sum_w' D[w][w']   Z^T[w'][h']   A = D (chain-permuted), B = P2's accumulator
//       V^T -> wave-private LDS (the only transpose: h' must become the K index)
//   P4  Y   [h][w]   = sum_h' D[h][h']   V[h'][w]      A = D, B = V^T rows from LDS
// (an MFMA accumulator - rows 4g+r, column = lane - is a legal B operand for the next MFMA once the
// constant A matrix has its columns permuted to match: the 2N x 2N upsampled plane lives only in
// registers).  Barriers per item: 3 (tile staged / all planes done / output staged) instead of 6
// per slab; MFMA work of one wave overlaps the SiLU VALU work of another; 2 workgroups per CU.
template <typename T, int N, int CH>
__global__ void __launch_bounds__(256) k_af_act_plane(AfP<T> p) {
  typedef PlaneCfg<T, N, CH> CF;
  typedef Mma<T> MM;
  typedef typename MM::Chunk Chunk;
  constexpr int EPC = CF::EPC, KPF = CF::KPF, KH = CF::KH;
  constexpr int KHP = CF::KHP, H2P = CF::H2P, YRP = CF::YRP;
  constexpr int NT = CF::NW * 64, TN = CF::TN, TH = CF::TH, NKF1 = CF::NKF1, NKF3 = CF::NKF3, CPW = CF::CPW;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* Xs = reinterpret_cast<T*>(smem);
  T* Cs = Xs + CF::XS;                                   // constants (aliased by Vt when they live in registers)
  T* Vt = CF::CREG ? Cs : Cs + CF::CONST_ELEMS;
  float* gscb = reinterpret_cast<float*>(Cs + CF::CV);   // [2][16] per-channel GroupNorm scale of the current / next item (CH used)
  float* gshb = gscb + 32;                               // [2][16] shift
  float* gmsn = gshb + 32;                               // [16][2] (mean, rstd) of the NEXT item's groups

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int Ct = p.C1 + p.C2;
  const int ctiles = Ct / CH;
  const int nitems = p.B * ctiles;
  const int cpg = p.gs.st1 ? Ct / p.G : 1;
  T* Vw = Vt + wave * CF::VT;

  // ---- once per (persistent) workgroup: constant fragments -> registers (bf16: straight from the packed image, one
  // coalesced 1 KB read per fragment, no LDS round trip and no barrier) or -> LDS (fp32: 4x the registers)
  Chunk creg[CF::CREG ? CF::NFRAG : 1];
  if constexpr (CF::CREG) {
#pragma unroll
    for (int f = 0; f < CF::NFRAG; ++f) creg[f] = ld16<Chunk>(reinterpret_cast<const T*>(p.packed) + (f * 64 + lane) * EPC);
  } else {
    const Chunk* src = reinterpret_cast<const Chunk*>(p.packed);
    Chunk* dst = reinterpret_cast<Chunk*>(Cs);
    for (int i = tid; i < CF::CONST_ELEMS / EPC; i += NT) dst[i] = src[i];
    __syncthreads();
  }
  auto cfrag = [&](int f) -> Chunk {
    if constexpr (CF::CREG) return creg[f];
    else return ld16<Chunk>(Cs + (f * 64 + lane) * EPC);
  };

  // X tile staging units: unit u = (cq, w, hq) reads PX pixels (h = hq*PX + e) x EPC channels and
  // writes them transposed (h contiguous).  UPT units per thread live in registers so that the NEXT
  // item's tile is fetched from HBM while the current one is being computed.  On 16^2 planes PX is chosen so that EVERY
  // thread holds a unit: with 8-pixel units a 16^2 x 8-channel tile was 32 units - half a wave normalised and transposed
  // the whole tile (2.3-2.6 k clocks per item with the other 3.5 waves waiting at the barrier): 19.6 -> 18.6 us per launch.
  // (32^2 x 8 channels = 128 eight-pixel units: 4-pixel units for all 256 threads shorten the phase in clocks but the
  //  launch takes 34.7 us instead of 31.7, same box - kept at 8; profiles/r04/af_plane_trace.txt)
  constexpr int CQ = CH / EPC;
  constexpr int PXW = N * N * CQ / NT, PX = PXW >= 4 ? EPC : (PXW >= 1 ? PXW : 1);
  constexpr int HQ = N / PX, UNITS = N * CQ * HQ, UPT = (UNITS + NT - 1) / NT;
  static_assert(N % PX == 0 && (PX & (PX - 1)) == 0, "staging unit");
  Chunk pre[UPT][PX];
  auto fetch = [&](int item) {
    const int b = item / ctiles, c0 = (item - b * ctiles) * CH;
    const bool second = c0 >= p.C1;
    const T* xsrc = second ? p.x2 : p.x1;
    const int Cs_ = second ? p.C2 : p.C1, cs0 = second ? c0 - p.C1 : c0;
#pragma unroll
    for (int k = 0; k < UPT; ++k) {
      const int u = tid + k * NT;
      if (u < UNITS) {
        const int cq = u % CQ, w = (u / CQ) % N, hq = u / (CQ * N);
#pragma unroll
        for (int e = 0; e < PX; ++e)
          pre[k][e] = p.x_blocked ? ld16<Chunk>(xsrc + (((size_t)b * (Cs_ / EPC) + cs0 / EPC + cq) * N * N + (hq * PX + e) * N + w) * EPC)
                                  : ld16<Chunk>(xsrc + ((size_t)(b * N + hq * PX + e) * N + w) * Cs_ + cs0 + cq * EPC);
      }
    }
  };

  // contiguous item range per workgroup: the channel tiles that share a 128-byte line of a pixel
  // are processed back-to-back by one workgroup (and neighbouring workgroups sit on one XCD)
  const int wl = xcd_remap(blockIdx.x, gridDim.x);
  const int ipw = (nitems + (int)gridDim.x - 1) / (int)gridDim.x;
  const int item_end = (wl + 1) * ipw < nitems ? (wl + 1) * ipw : nitems;
  int item = wl * ipw;
  const int item_first = item;
  auto stamp = [&](int i) {
    if (p.trace && lane == 0 && item - item_first < 2)
      p.trace[(((size_t)blockIdx.x * CF::NW + wave) * 2 + (item - item_first)) * 10 + i] = __builtin_amdgcn_s_memtime();
  };
  stamp(0);
  if (p.stagger > 0) {
    // A/B: co-resident workgroups start their item loops `slot` x stagger x 64 clocks apart, so that one
    // stages / stores while another runs its passes.  slot = the wave's slot on its SIMD (HW_ID bits 3:0): the three (N = 32) /
    // four (N = 16) four-wave workgroups of a CU sit in consecutive slots.
    int hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    const int slot = hwid & 15;
    for (int i = 0; i < slot * p.stagger; ++i) __builtin_amdgcn_s_sleep(1);
  }

  // ---- GroupNorm scale / shift of an item's CH channels, buffered for two items (gscb / gshb[buf]).
  // One wave per group touched by the item's channels (usually 2-4): the cpg x S per-channel partial sums strided over
  // its lanes, folded with xor-shuffles, mean / rstd finished in fp64 by lane 0.  The FIRST item of a workgroup does this
  // in line (two barriers); every later item is PIPELINED: its partial sums (and gamma / beta) are requested into
  // registers before the current item's MFMA passes and folded after them, so the dependent load chain that used to
  // open every item (~2 us with nothing else to issue: 3 items per workgroup at N = 16) now runs under the passes.
  // Same loads, same order of additions: bit-identical statistics.
  constexpr int SPL = 2;                                   // partial-sum loads a lane keeps in flight
  const int smax = p.gs.S1 > p.gs.S2 ? p.gs.S1 : p.gs.S2;
  auto stats_inline = [&](int it, int buf) {               // (contains workgroup barriers: every thread calls it)
    const int b = it / ctiles, c0 = (it - b * ctiles) * CH;
    if (tid < CH) {
      gscb[buf * 16 + tid] = 1.f;
      gshb[buf * 16 + tid] = 0.f;
    }
    if (p.gs.st1) {
      const int g_first = c0 / cpg, g_last = (c0 + CH - 1) / cpg;
      for (int g = g_first + wave; g <= g_last; g += CF::NW) {
        double s1, s2;
        gn_group_sums_wave(p.gs, b, g, cpg, lane, s1, s2);
        if (lane == 0) {
          float mean, rstd;
          gn_mean_rstd(s1, s2, (double)N * N * cpg, p.eps, mean, rstd);
          gmsn[2 * (g - g_first)] = mean;
          gmsn[2 * (g - g_first) + 1] = rstd;
        }
      }
      __syncthreads();
      if (tid < CH) {
        const int gl = (c0 + tid) / cpg - g_first;
        const float sc = gmsn[2 * gl + 1] * p.gamma[c0 + tid];
        gscb[buf * 16 + tid] = sc;
        gshb[buf * 16 + tid] = p.beta[c0 + tid] - gmsn[2 * gl] * sc;
      }
    }
  };
  // (N = 32 with 8-channel items sits at 166 VGPRs - 3 waves per SIMD - and the ~10 registers this state keeps live across
  //  the passes would cost it the third wave: there every item takes the in-line path; 2 items per workgroup, 14 us each)
  //  likewise N = 16 with 16-channel items: 126 -> 140 VGPRs would drop it from 4 to 3 waves per SIMD)
  constexpr bool PIPE = !(N == 32 && CH == 8) && !(N == 16 && CH == 16);
  f32x2 sreg[SPL];
  float gmr = 1.f, btr = 0.f;
  int n_c0 = 0, n_gfirst = 0, n_g = 0;
  bool n_piped = false, n_mine = false;
  auto stats_request = [&](int it) {                       // the next item's partial sums -> registers
    const int b = it / ctiles;
    n_c0 = (it - b * ctiles) * CH;
    n_gfirst = n_c0 / cpg;
    const int g_last = (n_c0 + CH - 1) / cpg;
    n_piped = PIPE && p.gs.st1 != nullptr && g_last - n_gfirst + 1 <= CF::NW && cpg * smax <= 64 * SPL;     // (workgroup-uniform)
    n_g = n_gfirst + wave;
    n_mine = n_piped && n_g <= g_last;
    if (n_mine) {
#pragma unroll
      for (int u = 0; u < SPL; ++u) {
        const int j = lane + 64 * u;
        sreg[u] = f32x2{0.f, 0.f};
        if (j < cpg * smax) {
          const int c = n_g * cpg + j / smax, sp = j - (j / smax) * smax;
          const bool second = c >= p.gs.C1;
          const int S = second ? p.gs.S2 : p.gs.S1;
          if (sp < S) {
            const float* st = second ? p.gs.st2 : p.gs.st1;
            const int Cs = second ? p.gs.C2 : p.gs.C1, cc = second ? c - p.gs.C1 : c;
            sreg[u] = *reinterpret_cast<const f32x2*>(st + (((size_t)b * S + sp) * Cs + cc) * 2);
          }
        }
      }
    }
    if (n_piped && tid < CH) {
      gmr = p.gamma[n_c0 + tid];
      btr = p.beta[n_c0 + tid];
    }
  };
  auto stats_fold = [&]() {                                // after the passes, in front of the barrier that follows them
    if (!n_mine) return;
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int u = 0; u < SPL; ++u) {
      if (lane + 64 * u < cpg * smax) {                    // (exactly the terms, in the order, of gn_group_sums_wave)
        s1 += (double)sreg[u][0];
        s2 += (double)sreg[u][1];
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      s1 += __shfl_xor(s1, o, 64);
      s2 += __shfl_xor(s2, o, 64);
    }
    if (lane == 0) {
      float mean, rstd;
      gn_mean_rstd(s1, s2, (double)N * N * cpg, p.eps, mean, rstd);
      gmsn[2 * (n_g - n_gfirst)] = mean;
      gmsn[2 * (n_g - n_gfirst) + 1] = rstd;
    }
  };
  auto stats_publish = [&](int buf) {                      // behind that barrier: per-channel scale / shift of the next item
    if (n_piped && tid < CH) {
      const int gl = (n_c0 + tid) / cpg - n_gfirst;
      const float sc = gmsn[2 * gl + 1] * gmr;
      gscb[buf * 16 + tid] = sc;
      gshb[buf * 16 + tid] = btr - gmsn[2 * gl] * sc;
    }
  };

  // ---- the workgroup's FIRST TWO items at once, in front of everything else: they are consecutive channel tiles of one
  // sample (2 CH channels, <= NW groups), so one wave per group folds the partial sums while gamma / beta - requested
  // BEFORE the partial sums - and the first tile (requested right behind them) are in flight: one load round trip and one
  // barrier for both items.  Item by item the chain (partial sums -> barrier -> gamma / beta -> barrier, ~2.5 us with
  // every wave of the CU idle: the co-resident workgroups run in step) opened each of them.  Same terms in the same order
  // as gn_group_sums_wave: bit-identical statistics.
  int pre_done = 0;
  if (item < item_end) {
    const int b = item / ctiles, c0 = (item - b * ctiles) * CH;
    int nit = 1;
    if (item + 1 < item_end && (item + 1) / ctiles == b) nit = 2;
    const int nch = nit * CH;
    const int g_first = c0 / cpg, g_last = (c0 + nch - 1) / cpg;
    const bool fast = p.gs.st1 != nullptr && g_last - g_first + 1 <= CF::NW && cpg * smax <= 64 * SPL;    // (workgroup-uniform)
    if (p.gs.st1 == nullptr || fast) {
      float gm = 1.f, bt = 0.f;
      f32x2 sv[SPL];
      const int g = g_first + wave;
      if (fast) {
        if (tid < nch) {
          gm = p.gamma[c0 + tid];
          bt = p.beta[c0 + tid];
        }
#pragma unroll
        for (int u = 0; u < SPL; ++u) {
          const int j = lane + 64 * u;
          sv[u] = f32x2{0.f, 0.f};
          if (g <= g_last && j < cpg * smax) {
            const int c = g * cpg + j / smax, sp = j - (j / smax) * smax;
            const bool second = c >= p.gs.C1;
            const int S = second ? p.gs.S2 : p.gs.S1;
            if (sp < S) {
              const float* st = second ? p.gs.st2 : p.gs.st1;
              const int Cs_ = second ? p.gs.C2 : p.gs.C1, cc = second ? c - p.gs.C1 : c;
              sv[u] = *reinterpret_cast<const f32x2*>(st + (((size_t)b * S + sp) * Cs_ + cc) * 2);
            }
          }
        }
      }
      fetch(item);
      if (fast) {
        if (g <= g_last) {
          double s1 = 0.0, s2 = 0.0;
#pragma unroll
          for (int u = 0; u < SPL; ++u) {
            if (lane + 64 * u < cpg * smax) {
              s1 += (double)sv[u][0];
              s2 += (double)sv[u][1];
            }
          }
#pragma unroll
          for (int o = 32; o > 0; o >>= 1) {
            s1 += __shfl_xor(s1, o, 64);
            s2 += __shfl_xor(s2, o, 64);
          }
          if (lane == 0) {
            float mean, rstd;
            gn_mean_rstd(s1, s2, (double)N * N * cpg, p.eps, mean, rstd);
            gmsn[2 * (g - g_first)] = mean;
            gmsn[2 * (g - g_first) + 1] = rstd;
          }
        }
        __syncthreads();
      }
      if (tid < nch) {
        const int k = tid / CH, t = tid - k * CH;
        float sc = 1.f, sh = 0.f;
        if (fast) {
          const int gl = (c0 + tid) / cpg - g_first;
          sc = gmsn[2 * gl + 1] * gm;
          sh = bt - gmsn[2 * gl] * sc;
        }
        gscb[k * 16 + t] = sc;
        gshb[k * 16 + t] = sh;
      }
      pre_done = nit;
    } else {
      fetch(item);
    }
  }

  bool have_stats = false;                                 // the pipelined path already published this item's scale / shift
  int ibuf = 0;
  for (; item < item_end; ++item, ibuf ^= 1) {
    const int b = item / ctiles;
    const int c0 = (item - b * ctiles) * CH;
    const float* gsc = gscb + ibuf * 16;
    const float* gsh = gshb + ibuf * 16;
    stamp(1);
    if (!have_stats && item - item_first >= pre_done) stats_inline(item, ibuf);
    __syncthreads();  // gsc/gsh ready; also: the previous item's output copy out of the X region is done
    stamp(2);
    if constexpr (KH > N) {  // the output staging of the previous item overwrote the X region: re-zero its K padding
      for (int i = tid; i < N * CH * (KH - N); i += NT) {
        const int row = i / (KH - N), k = N + (i - row * (KH - N));
        Xs[row * KHP + k] = from_f32<T>(0.f);
      }
    }
    // ---- prefetched tile -> Xs[c][w][h] (h K-contiguous), GroupNorm applied
#pragma unroll
    for (int k = 0; k < UPT; ++k) {
      const int u = tid + k * NT;
      if (u < UNITS) {
        const int cq = u % CQ, w = (u / CQ) % N, hq = u / (CQ * N);
#pragma unroll
        for (int cc = 0; cc < EPC; ++cc) {
          const float sc = gsc[cq * EPC + cc], sh = gsh[cq * EPC + cc];
          T* dst = Xs + ((size_t)((cq * EPC + cc) * N + w)) * KHP + hq * PX;
          if constexpr (PX == 1) {
            *dst = from_f32<T>(to_f32(pre[k][0][cc]) * sc + sh);
          } else {
            typedef __attribute__((ext_vector_type(PX))) T Run;       // PX consecutive h of one (channel, w): 4 - 16 bytes
            Run o;
#pragma unroll
            for (int e = 0; e < PX; ++e) o[e] = from_f32<T>(to_f32(pre[k][e][cc]) * sc + sh);
            *reinterpret_cast<Run*>(dst) = o;
          }
        }
      }
    }
    stamp(3);
    __syncthreads();
    stamp(4);
    have_stats = false;
    n_piped = n_mine = false;
    if (item + 1 < item_end) {                 // in flight during the MFMA passes
      fetch(item + 1);
      if constexpr (PIPE) {
        if (item + 1 - item_first >= pre_done) stats_request(item + 1);
      }
    }

#include "af_plane_passes.inc"
    stamp(5);
    if constexpr (PIPE) stats_fold();
    __syncthreads();  // every wave has finished reading the X planes: the region becomes the output tile
    stamp(6);
    if constexpr (PIPE) {
      stats_publish(ibuf ^ 1);
      have_stats = n_piped;
    }

    // ---- output: stage the [N][N][16] tile through LDS -> coalesced 16-byte stores
    T* Ys = Xs;
#pragma unroll
    for (int pl = 0; pl < CPW; ++pl)
#pragma unroll
      for (int t4 = 0; t4 < TN; ++t4)
#pragma unroll
        for (int tw = 0; tw < TN; ++tw)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            Ys[(16 * t4 + 4 * lg + r) * YRP + (16 * tw + li) * CH + wave * CPW + pl] = from_f32<T>(yacc[pl][t4][tw][r]);
    stamp(7);
    __syncthreads();
    stamp(8);
    {
      constexpr int CPP = CH / EPC;  // 16-byte chunks per pixel
      for (int i = tid; i < N * N * CPP; i += NT) {
        const int pix = i / CPP, q = i - pix * CPP;
        const int h = pix / N, w = pix - h * N;
        // (blocks leave as whole contiguous lines: stored write-through, nothing of them is left dirty in the L2 for the
        //  end-of-kernel release to write back - 4.918 -> 4.898 ms/step; NHWC pieces of 16 bytes need the L2 to combine them)
        if (p.y_blocked) st16_wt<Chunk>(p.y + (((size_t)b * (Ct / EPC) + c0 / EPC + q) * N * N + pix) * EPC, ld16<Chunk>(Ys + h * YRP + w * CH + q * EPC));
        else st16_out<Chunk>(p.y + ((size_t)b * N * N + pix) * Ct + c0 + q * EPC, ld16<Chunk>(Ys + h * YRP + w * CH + q * EPC));
      }
    }
    stamp(9);
  }  // persistent item loop
}

// ----------------------------------------------------------------------------- Kronecker form (N = 4, 8)
// For small planes the whole 2-D operator is ONE dense matrix per direction:
//   Z[(h',w'), c] = sum_{(h,w)} KU[(h',w'),(h,w)] X[(h,w), c],   KU = U (x) U   [4N^2 x N^2]
//   Y[(h,w),  c]  = sum_{(h',w')} KD[(h,w),(h',w')] silu(Z)[(h',w'), c],  KD = D (x) D
// One wave = one (sample, 16 channels) item: GEMM1 (A = KU rows from LDS, B = the transposed X
// tile) leaves Z in the MFMA accumulator layout, which IS a legal B operand for GEMM2 once KD's
// columns are permuted to match (chain trick, see Mma<T>) — the upsampled plane never leaves
// registers.  The dense form costs more flops than the separable one (x N/3) but these levels are
// tiny and HBM-bound; what matters is that the item is 64 MFMAs instead of ~6000 VALU FMAs.
template <typename T, int N>
struct KronCfg {
  typedef Mma<T> MM;
  static constexpr int EPC = MM::EPC, KPF = MM::KPF;
  static constexpr int P = N * N, Q = 4 * N * N;
  static constexpr int PK = ((P + KPF - 1) / KPF) * KPF;   // K extent of GEMM1
  static constexpr int PR = ((P + 15) / 16) * 16;          // output rows of GEMM2, padded to tiles
  static constexpr bool PERM = sizeof(T) == 2;
  static constexpr int KU_ELEMS = Q * PK, KD_ELEMS = PR * Q;
  static constexpr int CONST_ELEMS = KU_ELEMS + KD_ELEMS;
  static constexpr int XK = 16 * PK;                        // per-wave transposed X tile
  static constexpr int YS = PR * 16;                        // per-wave output staging tile [px][16 c]
  static constexpr int LDS_BYTES = (CONST_ELEMS + 4 * (XK + YS)) * (int)sizeof(T);
};

template <typename T, int N>
__global__ void k_af_pack_kron(const float* __restrict__ U, const float* __restrict__ D, T* __restrict__ out) {
  typedef KronCfg<T, N> CF;
  constexpr int P = CF::P, Q = CF::Q, PK = CF::PK, H2 = 2 * N;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < CF::CONST_ELEMS; i += gridDim.x * blockDim.x) {
    float v = 0.f;
    if (i < CF::KU_ELEMS) {
      const int q = i / PK, pp = i - q * PK;
      if (pp < P) {
        const int hp = q / H2, wp = q - hp * H2, h = pp / N, w = pp - h * N;
        v = U[hp * N + h] * U[wp * N + w] * 1.4426950408889634f;   // SiLU works on z log2(e) (see silu_log2_x4)
      }
    } else {
      const int j = i - CF::KU_ELEMS;
      const int pr = j / Q;
      int k = j - pr * Q;
      if (CF::PERM) {  // column 32f + 8g + e  <-  column 32f + (e < 4 ? 4g + e : 16 + 4g + e - 4)
        const int f = k >> 5, g = (k >> 3) & 3, e = k & 7;
        k = 32 * f + (e < 4 ? 4 * g + e : 16 + 4 * g + (e - 4));
      }
      if (pr < P) {
        const int h = pr / N, w = pr - h * N, hp = k / H2, wp = k - hp * H2;
        v = D[h * H2 + hp] * D[w * H2 + wp] * 0.6931471805599453f;
      }
    }
    out[i] = from_f32<T>(v);
  }
}

template <typename T, int N>
__global__ void __launch_bounds__(256) k_af_act_kron(AfP<T> p) {
  typedef KronCfg<T, N> CF;
  typedef Mma<T> MM;
  typedef typename MM::Chunk Chunk;
  constexpr int EPC = CF::EPC, KPF = CF::KPF, P = CF::P, Q = CF::Q, PK = CF::PK, PR = CF::PR;
  constexpr int NKF1 = PK / KPF, NZ = Q / 16, NKF2 = Q / KPF, NY = PR / 16;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* KU = reinterpret_cast<T*>(smem);
  T* KD = KU + CF::KU_ELEMS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  T* Xk = KD + CF::KD_ELEMS + wave * (CF::XK + CF::YS);   // this wave's [16 c][PK] tile
  T* Ys = Xk + CF::XK;                                      // ... and its [px][16 c] output staging tile
  const int li = lane & 15, lg = lane >> 4;
  const int Ct = p.C1 + p.C2, ctiles = Ct / 16, nitems = p.B * ctiles;
  const int cpg = p.gs.st1 ? Ct / p.G : 1;

  // X tile of an item: [P px][16 c], unit = EPC pixels x EPC channels; lanes stride the units (the channel octet / quad
  // of a lane is fixed: 64 % CQ == 0).  The loads of the FIRST item are issued before the 64 KB constant image is
  // copied and before the GroupNorm partial sums are fetched, those of the next item while the current one is in its
  // GEMMs: the three latencies used to follow one another in every launch (constants, statistics, tile).
  constexpr int CQ = 16 / EPC, PQ = P / EPC, UNITS = CQ * PQ, UPL = (UNITS + 63) / 64;
  const int ngroups = (nitems + 3) / 4;
  Chunk xin[UPL][EPC];
  auto item_of = [&](int grp, int& b, int& c0) {
    const int item = grp * 4 + wave;
    const bool live = item < nitems;
    b = live ? item / ctiles : 0;
    c0 = live ? (item - b * ctiles) * 16 : 0;
    return live;
  };
  auto issue_x = [&](int grp) {
    int b, c0;
    if (grp >= ngroups || !item_of(grp, b, c0)) return;
    const bool second = c0 >= p.C1;
    const T* xsrc = second ? p.x2 : p.x1;
    const int Cs = second ? p.C2 : p.C1, cs0 = second ? c0 - p.C1 : c0;
#pragma unroll
    for (int k = 0; k < UPL; ++k) {
      const int u = lane + 64 * k;
      if (u < UNITS) {
        const int cq = u % CQ, pq = u / CQ;
#pragma unroll
        for (int e = 0; e < EPC; ++e) xin[k][e] = ld16<Chunk>(xsrc + ((size_t)b * P + pq * EPC + e) * Cs + cs0 + cq * EPC);
      }
    }
  };
  issue_x(blockIdx.x);
  {
    const Chunk* src = reinterpret_cast<const Chunk*>(p.packed);
    Chunk* dst = reinterpret_cast<Chunk*>(KU);
    for (int i = tid; i < CF::CONST_ELEMS / EPC; i += 256) dst[i] = src[i];
  }
  if constexpr (PK > P) {  // zero K padding of the X tile once (never overwritten)
    for (int i = lane; i < 16 * (PK - P); i += 64) {
      const int row = i / (PK - P), k = P + (i - row * (PK - P));
      Xk[row * PK + k] = from_f32<T>(0.f);
    }
  }
  __syncthreads();

  for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    int b, c0;
    const bool live = item_of(grp, b, c0);

    // ---- GroupNorm scale/shift of channel c0 + li: lane group lg adds every 4th channel of its group
    float sc = 1.f, sh = 0.f;
    if (p.gs.st1) {   // (wave-uniform branch; idle waves of the last group take part with their clamped item)
      const float gm = p.gamma[c0 + li], bt = p.beta[c0 + li];    // requested in front of the partial sums: one round trip, not two
      float mean, rstd;
      gn_wave_keys(p.gs, true, b, (c0 + li) / cpg, cpg, (double)P * cpg, p.eps, lane, mean, rstd);
      sc = rstd * gm;
      sh = bt - mean * sc;
    }

    // ---- X tile (already in registers) -> Xk[c][px] (pixels K-contiguous), GroupNorm applied; the scale / shift of a
    // lane's channel octet are fetched with full-wave shuffles up front.
    float usc[EPC], ush[EPC];
#pragma unroll
    for (int cc = 0; cc < EPC; ++cc) {
      usc[cc] = __shfl(sc, (lane % CQ) * EPC + cc, 64);
      ush[cc] = __shfl(sh, (lane % CQ) * EPC + cc, 64);
    }
    // (Xk / Ys are private to the wave and same-wave LDS operations are ordered: no workgroup barrier,
    //  the four waves drift apart and overlap their load / MFMA / store phases)
    if (live) {
#pragma unroll
      for (int k = 0; k < UPL; ++k) {
        const int u = lane + 64 * k;
        if (u < UNITS) {
          const int cq = u % CQ, pq = u / CQ;
#pragma unroll
          for (int cc = 0; cc < EPC; ++cc) {
            Chunk o;
#pragma unroll
            for (int e = 0; e < EPC; ++e) o[e] = from_f32<T>(to_f32(xin[k][e][cc]) * usc[cc] + ush[cc]);
            st16<Chunk>(Xk + (cq * EPC + cc) * PK + pq * EPC, o);
          }
        }
      }
    }
    issue_x(grp + (int)gridDim.x);      // the next item's tile travels during this item's GEMMs

    // ---- GEMM1 + SiLU
    Chunk xf[NKF1];
#pragma unroll
    for (int kf = 0; kf < NKF1; ++kf) xf[kf] = ld16<Chunk>(Xk + li * PK + kf * KPF + lg * EPC);
    f32x4 z[NZ];
#pragma unroll
    for (int t = 0; t < NZ; ++t) {
      z[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kf = 0; kf < NKF1; ++kf) MM::mma(z[t], ld16<Chunk>(KU + (16 * t + li) * PK + kf * KPF + lg * EPC), xf[kf]);
      z[t] = silu_log2_x4(z[t]);
    }
    // ---- GEMM2 (chained) + store: lane (c = li, g) holds pixels 16 t + 4 g + r
    Chunk pb[NKF2];
    if constexpr (CF::PERM) {
#pragma unroll
      for (int f = 0; f < NKF2; ++f) pb[f] = pack_chain<T>(z[2 * f], z[2 * f + 1]);
    } else {
#pragma unroll
      for (int f = 0; f < NKF2; ++f) pb[f] = z[f];
    }
#pragma unroll
    for (int t = 0; t < NY; ++t) {
      f32x4 y = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int f = 0; f < NKF2; ++f) MM::mma(y, ld16<Chunk>(KD + (16 * t + li) * Q + f * KPF + lg * EPC), pb[f]);
#pragma unroll
      for (int r = 0; r < 4; ++r) Ys[(16 * t + 4 * lg + r) * 16 + li] = from_f32<T>(y[r]);
    }
    if (live) {   // 16-byte stores: two (bf16) / four (fp32) lanes per pixel
      constexpr int CPP = 16 / EPC;
      for (int i = lane; i < P * CPP; i += 64) {
        const int px = i / CPP, q = i - px * CPP;
        st16<Chunk>(p.y + ((size_t)b * P + px) * Ct + c0 + q * EPC, ld16<Chunk>(Ys + px * 16 + q * EPC));
      }
    }
  }
}

// ----------------------------------------------------------------------------- 8 x 8 planes, separable (round 4)
// k_af_act_p8 (bf16): the Kronecker kernel above stages a 64 KB constant image per workgroup for 8 KB of planes and spends
// two thirds of its wave-cycles parked (profiles/r03/pmc_sq_r03j.txt); here the four separable passes of the plane kernel
// run on 16 x 16 x 32 MFMAs with TWO 8 x 8 planes sharing every tile, the upsampled 16 x 16 planes never leave the
// registers, and the constants are five fragments (20 VGPRs) built from U / D at kernel start - no constant image, no
// workgroup barrier, 4 KB of LDS per wave (input transpose + output staging).
//   pass 1  T1^T[(p,w)][h'] = sum_h X_p[h][w] U[h'][h]                         A = X^T rows (LDS), B = U^T        (K = 8)
//   pass 2  Z_p^T[w'][h']   = sum_w U[w'][w] T1^T[(p,w)][h']                   A = U (plane-selecting), B = pass 1's accumulator
//           SiLU on the accumulator
//   pass 3  V[h'][(p,w)]    = sum_w' S_p^T[w'][h'] D[w][w']                    A = pass 2's accumulators of BOTH planes, B = D (block diagonal)
//   pass 4  Y[h][(p,w)]     = sum_h' D[h][h'] V[h'][(p,w)]                     A = D (two channel pairs stacked), B = pass 3's accumulators of two pairs
// (an accumulator tile D[i][j] - lane = column j, registers = rows 4g + r - is a legal B operand of the next MFMA, contracting
//  its ROW index, and read as an A operand it is its own transpose, again contracting the row index: passes 2 - 4 alternate
//  the two so that w, then w', then h' are each the row index when their turn comes - no transpose through LDS.)
__global__ void __launch_bounds__(256) k_af_act_p8(AfP<bf16> p) {
  typedef bf16 T;
  typedef Mma<T> MM;
  __shared__ __attribute__((aligned(16))) char smem[4 * 4096];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  char* XT = smem + wave * 4096;                 // [16 c][8 w][8 h] bf16: 16 bytes per (c, w)
  T* Ys = reinterpret_cast<T*>(XT + 2048);       // [64 px][16 c]
  const int Ct = p.C1 + p.C2, ctiles = Ct / 16, nitems = p.B * ctiles;
  const int cpg = p.gs.st1 ? Ct / p.G : 1;
  const int item = blockIdx.x * 4 + wave;
  const bool live = item < nitems;
  const int b = live ? item / ctiles : 0, c0 = live ? (item - b * ctiles) * 16 : 0;
  const bool second = c0 >= p.C1;
  const T* xsrc = second ? p.x2 : p.x1;
  const int Cs = second ? p.C2 : p.C1, cs0 = second ? c0 - p.C1 : c0;

  // ---- input: lane (w = l & 7, channel octet cq = (l >> 3) & 1, row pair hh = l >> 4) reads rows 2 hh, 2 hh + 1
  const int xw = lane & 7, xcq = (lane >> 3) & 1, xhh = lane >> 4;
  bf16x8 xin[2];
#pragma unroll
  for (int e = 0; e < 2; ++e)
    xin[e] = ld16<bf16x8>(xsrc + ((size_t)b * 64 + (2 * xhh + e) * 8 + xw) * Cs + cs0 + xcq * 8);

  // ---- constants (fragments of U [16][8] and D [8][16]; log2 e / ln 2 folded in as in the other MFMA kernels)
  const float* __restrict__ U = p.U;
  const float* __restrict__ D = p.D;
  bf16x8 cB1, cA2[2], cB3, cA4;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    cB1[e] = (bf16)(lg == 0 ? U[li * 8 + e] : 0.f);                                              // B[k = h = e][j = h' = li]
    const int w2 = 4 * (lg & 1) + (e & 3);
#pragma unroll
    for (int t = 0; t < 2; ++t)
      cA2[t][e] = (bf16)((e < 4 && (lg >> 1) == t) ? U[li * 8 + w2] * 1.4426950408889634f : 0.f);   // A[i = w' = li][k = (p = g >> 1, w)]
    const int wp = 4 * lg + (e & 3);                                                               // w' (pass 3) / h' (pass 4) of element e
    cB3[e] = (bf16)(((e >> 2) == (li >> 3)) ? D[(li & 7) * 16 + wp] * 0.6931471805599453f : 0.f);  // B[k = (plane e >> 2, w')][j = (p_o, w_o) = li]
    cA4[e] = (bf16)(((e >> 2) == (li >> 3)) ? D[(li & 7) * 16 + wp] : 0.f);                       // A[i = (pair, h) = li][k = (pair e >> 2, h')]
  }

  // ---- GroupNorm scale / shift of channel c0 + li (all lanes take part), then of this lane's channel octet
  float sc = 1.f, sh = 0.f;
  if (p.gs.st1) {
    const float gm = p.gamma[c0 + li], bt = p.beta[c0 + li];      // requested in front of the partial sums
    float mean, rstd;
    gn_wave_keys(p.gs, true, b, (c0 + li) / cpg, cpg, 64.0 * cpg, p.eps, lane, mean, rstd);
    sc = rstd * gm;
    sh = bt - mean * sc;
  }
#pragma unroll
  for (int cc = 0; cc < 8; ++cc) {
    const float k = __shfl(sc, xcq * 8 + cc, 64), s0 = __shfl(sh, xcq * 8 + cc, 64);
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    bf16x2 v;
    v[0] = (bf16)((float)xin[0][cc] * k + s0);
    v[1] = (bf16)((float)xin[1][cc] * k + s0);
    *reinterpret_cast<bf16x2*>(XT + (((xcq * 8 + cc) * 8 + xw) << 4) + xhh * 4) = v;
  }
  // (XT / Ys are private to the wave; LDS operations of one wave execute in order)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int qq = 0; qq < 4; ++qq) {               // two channel pairs per round (pass 4 stacks them)
    f32x4 v3[2];
#pragma unroll
    for (int hp = 0; hp < 2; ++hp) {
      const int q = 2 * qq + hp;                  // channel pair: planes c0 + 2q, c0 + 2q + 1
      // pass 1: rows (p, w) = li, K = h (lane group 0 only)
      bf16x8 xa = ld16<bf16x8>(XT + (((2 * q + (li >> 3)) * 8 + (li & 7)) << 4));
      if (lg != 0) xa = MM::zero();
      f32x4 t1 = zero4;
      MM::mma(t1, xa, cB1);
      // pass 2 + SiLU: one tile per plane
      const bf16x8 b2 = pack_chain<T>(t1, zero4);
      f32x4 z0 = zero4, z1 = zero4;
      MM::mma(z0, cA2[0], b2);
      MM::mma(z1, cA2[1], b2);
      z0 = silu_log2_x4(z0);
      z1 = silu_log2_x4(z1);
      // pass 3: both planes in one product (the accumulators as A: rows h')
      const bf16x8 a3 = pack_chain<T>(z0, z1);
      v3[hp] = zero4;
      MM::mma(v3[hp], a3, cB3);
    }
    // pass 4: rows (pair, h), columns (p, w)
    const bf16x8 b4 = pack_chain<T>(v3[0], v3[1]);
    f32x4 y = zero4;
    MM::mma(y, cA4, b4);
    const int q = 2 * qq + (lg >> 1), c = 2 * q + (li >> 3), w = li & 7;
#pragma unroll
    for (int r = 0; r < 4; ++r) Ys[((4 * (lg & 1) + r) * 8 + w) * 16 + c] = (bf16)y[r];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  if (live) {
#pragma unroll
    for (int e = 0; e < 2; ++e)
      st16<bf16x8>(p.y + ((size_t)b * 64 + lane) * Ct + c0 + e * 8, ld16<bf16x8>(Ys + lane * 16 + e * 8));
  }
}

// ----------------------------------------------------------------------------- small planes
template <typename T, int N>
__global__ void __launch_bounds__(256) k_af_act_small(AfP<T> p) {
  constexpr int H2 = 2 * N;
  const int Ct = p.C1 + p.C2;
  const int total = p.B * Ct;
  const float* __restrict__ U = p.U;
  const float* __restrict__ D = p.D;
  {   // (the launch covers `total` exactly once; a wave works out its lanes' GroupNorm keys together)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < total;
    const int b = live ? i / Ct : 0, c = live ? i - b * Ct : 0;
    const bool second = c >= p.C1;
    const T* xs = second ? p.x2 : p.x1;
    const int Cs = second ? p.C2 : p.C1;
    const int cc = second ? c - p.C1 : c;
    // the plane itself is requested first (dead lanes re-read plane (0, 0)): its latency runs under the statistics chain
    T xr[N][N];
#pragma unroll
    for (int h = 0; h < N; ++h)
#pragma unroll
      for (int w = 0; w < N; ++w) xr[h][w] = xs[((size_t)(b * N + h) * N + w) * Cs + cc];
    float sc = 1.f, sh = 0.f;
    if (p.gs.st1) {
      const int cpg = Ct / p.G;
      const float gm = p.gamma[c], bt = p.beta[c];                // requested in front of the partial sums
      float mean, rstd;
      gn_wave_keys(p.gs, live, b, c / cpg, cpg, (double)N * N * cpg, p.eps, threadIdx.x & 63, mean, rstd);
      sc = rstd * gm;
      sh = bt - mean * sc;
    }
    if (!live) return;
    float X[N][N], Y[N][N];
#pragma unroll
    for (int h = 0; h < N; ++h)
#pragma unroll
      for (int w = 0; w < N; ++w) {
        X[h][w] = to_f32(xr[h][w]) * sc + sh;
        Y[h][w] = 0.f;
      }
    // hp stays a run-time loop: X/Y are indexed statically (registers), the matrix rows with a
    // wave-uniform run-time offset (scalar loads), which keeps N = 8 inside the VGPR budget.
    // (N = 2: fully unrolled - all 16 coefficients arrive in one batch of scalar loads instead of one dependent batch
    //  per row of the upsampled plane)
    constexpr int HP_UNROLL = N == 2 ? 4 : N == 4 ? 8 : 1;
#pragma unroll HP_UNROLL
    for (int hp = 0; hp < H2; ++hp) {
      float t1[N];
#pragma unroll
      for (int w = 0; w < N; ++w) {
        float a = 0.f;
#pragma unroll
        for (int h = 0; h < N; ++h) a = fmaf(U[hp * N + h], X[h][w], a);
        t1[w] = a;
      }
      float sz[H2];
#pragma unroll
      for (int wp = 0; wp < H2; ++wp) {
        float a = 0.f;
#pragma unroll
        for (int w = 0; w < N; ++w) a = fmaf(U[wp * N + w], t1[w], a);
        sz[wp] = silu_f(a);
      }
#pragma unroll
      for (int w = 0; w < N; ++w) {
        float a = 0.f;
#pragma unroll
        for (int wp = 0; wp < H2; ++wp) a = fmaf(D[w * H2 + wp], sz[wp], a);
#pragma unroll
        for (int h = 0; h < N; ++h) Y[h][w] = fmaf(D[h * H2 + hp], a, Y[h][w]);
      }
    }
    if constexpr (N == 2) {
      // plane-constant form (afldm_af_act_const2): D = C(lpf(4))[::2] is 1/4 everywhere (ideal_lpf.py:17-21: lpf(4) = [1,0,0,0]), so the
      // four outputs are the same fmaf chain over the same operands - bit-equal - and ONE value per plane is stored: y [B][C]
      if (p.y_blocked == 2) {
        p.y[(size_t)b * Ct + c] = from_f32<T>(Y[0][0]);
        return;
      }
    }
#pragma unroll
    for (int h = 0; h < N; ++h)
#pragma unroll
      for (int w = 0; w < N; ++w) p.y[((size_t)(b * N + h) * N + w) * Ct + c] = from_f32<T>(Y[h][w]);
  }
}

// ----------------------------------------------------------------------------- small planes fed by split-K slabs
// conv1 -> norm2 -> activation of a resnet block on the 2x2 / 4x4 planes (afldm_af_act_slabs): one thread per
// (sample, channel) plane sums the convolution's K slices, adds bias + time embedding and rounds to T exactly as the
// reduction kernel would have stored it; a workgroup holds GPB whole groups of one sample (C / G channels each), so the
// GroupNorm statistics of those rounded values are formed here (per-plane sums through LDS, added per group in
// channel order in fp64) - no reduction launch, no stored intermediate.
template <typename T>
struct AfSlabP {
  const float* slabs;
  const float* bias;
  const T* temb;
  const float* gamma;
  const float* beta;
  const float* U;
  const float* D;
  T* y;
  const T* residual;   // optional [B][N*N][C]: added after bias + temb (conv2 of a resnet block)
  T* y_raw;            // optional: the finished (rounded) convolution output itself, for its other consumers
  int nslab, temb_stride, B, C, G, gpb;
  float eps;
  int compact;         // N = 2, ACT = 1: store the plane-constant result once, y [B][C] (afldm_af_act_slabs act = 2)
};

// ACT: 1 = WarpedNonlinearity after the GroupNorm (a resnet's norm2 / norm1), 0 = GroupNorm only (Attention.group_norm)
template <typename T, int N, int ACT>
__global__ void __launch_bounds__(256) k_af_act_slabs(AfSlabP<T> p) {
  constexpr int H2 = 2 * N, P = N * N;
  __shared__ float sS[256][2];
  __shared__ float sM[32][2];
  const int cpg = p.C / p.G, bps = p.G / p.gpb;            // channels per group, blocks per sample
  const int b = blockIdx.x / bps, g0 = (blockIdx.x - b * bps) * p.gpb;
  const int tid = threadIdx.x, c = g0 * cpg + tid;         // blockDim.x == gpb * cpg
  const size_t slab = (size_t)p.B * P * p.C;
  float X[N][N], Y[N][N];
  float add = p.bias ? p.bias[c] : 0.f;
  const float tv = p.temb ? to_f32(p.temb[(size_t)b * p.temb_stride + c]) : 0.f;
  const float gm = p.gamma[c], bt = p.beta[c];             // (with the first batch of loads, not behind the two barriers)
  float s1 = 0.f, s2 = 0.f;
  {
    // slab order z = 0, 1, ... per pixel (the reduction kernel's order); the first four slabs as fully unrolled
    // batches of P independent loads (a rolled loop waited out one load latency per pixel and slab)
    const float* q0 = p.slabs + (size_t)b * P * p.C + c;
    float v[P];
#pragma unroll
    for (int px = 0; px < P; ++px) v[px] = 0.f;
#pragma unroll
    for (int z = 0; z < 4; ++z) {
      if (z < p.nslab) {
        float t[P];
#pragma unroll
        for (int px = 0; px < P; ++px) t[px] = q0[(size_t)z * slab + (size_t)px * p.C];
#pragma unroll
        for (int px = 0; px < P; ++px) v[px] += t[px];
      }
    }
    for (int z = 4; z < p.nslab; ++z) {
#pragma unroll
      for (int px = 0; px < P; ++px) v[px] += q0[(size_t)z * slab + (size_t)px * p.C];
    }
#pragma unroll
    for (int h = 0; h < N; ++h)
#pragma unroll
      for (int w = 0; w < N; ++w) {
        float vv = v[h * N + w];
        if (p.bias) vv += add;
        if (p.temb) vv += tv;
        if (p.residual) vv += to_f32(p.residual[((size_t)(b * P + h * N + w)) * p.C + c]);
        const T rt = from_f32<T>(vv);
        if (p.y_raw) p.y_raw[((size_t)(b * P + h * N + w)) * p.C + c] = rt;
        const float r = to_f32(rt);                         // the value the two-launch path stores and normalises
        X[h][w] = r;
        s1 += r;
        s2 = fmaf(r, r, s2);
      }
  }
  sS[tid][0] = s1;
  sS[tid][1] = s2;
  __syncthreads();
  if (tid < p.gpb) {
    double a1 = 0.0, a2 = 0.0;
    for (int k = 0; k < cpg; ++k) {
      a1 += (double)sS[tid * cpg + k][0];
      a2 += (double)sS[tid * cpg + k][1];
    }
    float mean, rstd;
    gn_mean_rstd(a1, a2, (double)P * cpg, p.eps, mean, rstd);
    sM[tid][0] = mean;
    sM[tid][1] = rstd;
  }
  __syncthreads();
  {
    const float mean = sM[tid / cpg][0], rstd = sM[tid / cpg][1];
    const float sc = rstd * gm, sh = bt - mean * sc;
#pragma unroll
    for (int h = 0; h < N; ++h)
#pragma unroll
      for (int w = 0; w < N; ++w) {
        X[h][w] = X[h][w] * sc + sh;
        Y[h][w] = 0.f;
      }
  }
  if constexpr (ACT == 0) {
#pragma unroll
    for (int h = 0; h < N; ++h)
#pragma unroll
      for (int w = 0; w < N; ++w) p.y[((size_t)(b * N + h) * N + w) * p.C + c] = from_f32<T>(X[h][w]);
    return;
  }
  const float* __restrict__ U = p.U;
  const float* __restrict__ D = p.D;
#pragma unroll
  for (int hp = 0; hp < H2; ++hp) {
    float t1[N];
#pragma unroll
    for (int w = 0; w < N; ++w) {
      float a = 0.f;
#pragma unroll
      for (int h = 0; h < N; ++h) a = fmaf(U[hp * N + h], X[h][w], a);
      t1[w] = a;
    }
    float sz[H2];
#pragma unroll
    for (int wp = 0; wp < H2; ++wp) {
      float a = 0.f;
#pragma unroll
      for (int w = 0; w < N; ++w) a = fmaf(U[wp * N + w], t1[w], a);
      sz[wp] = silu_f(a);
    }
#pragma unroll
    for (int w = 0; w < N; ++w) {
      float a = 0.f;
#pragma unroll
      for (int wp = 0; wp < H2; ++wp) a = fmaf(D[w * H2 + wp], sz[wp], a);
#pragma unroll
      for (int h = 0; h < N; ++h) Y[h][w] = fmaf(D[h * H2 + hp], a, Y[h][w]);
    }
  }
  if constexpr (N == 2) {
    if (p.compact) {                                        // act = 2: the plane-constant value once, y [B][C] (see k_af_act_small)
      p.y[(size_t)b * p.C + c] = from_f32<T>(Y[0][0]);
      return;
    }
  }
#pragma unroll
  for (int h = 0; h < N; ++h)
#pragma unroll
    for (int w = 0; w < N; ++w) p.y[((size_t)(b * N + h) * N + w) * p.C + c] = from_f32<T>(Y[h][w]);
}

// ----------------------------------------------------------------------------- one-axis product
// in  viewed as [B][A][Wd][C];  axis 0: out[b][a'][w][c] = sum_a M[a'][a] in[b][a][w][c]
//                               axis 1: out[b][a][w'][c] = sum_w M[w'][w] in[b][a][w][c]
template <typename TI, typename TO>
__global__ void __launch_bounds__(256) k_axis_contract(const TI* __restrict__ in, TO* __restrict__ out,
                                                       const float* __restrict__ M, int B, int A, int Wd, int C,
                                                       int Rout, int axis) {
  const int Ao = axis == 0 ? Rout : A, Wo = axis == 1 ? Rout : Wd;
  const size_t total = (size_t)B * Ao * Wo * C;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    size_t r = i / C;
    const int w = (int)(r % Wo);
    r /= Wo;
    const int a = (int)(r % Ao);
    const int b = (int)(r / Ao);
    float acc = 0.f;
    if (axis == 0) {
      const TI* src = in + ((size_t)b * A * Wd + w) * C + c;
      const float* m = M + (size_t)a * A;
      for (int k = 0; k < A; ++k) acc = fmaf(m[k], to_f32(src[(size_t)k * Wd * C]), acc);
    } else {
      const TI* src = in + ((size_t)(b * A + a) * Wd) * C + c;
      const float* m = M + (size_t)w * Wd;
      for (int k = 0; k < Wd; ++k) acc = fmaf(m[k], to_f32(src[(size_t)k * C]), acc);
    }
    out[i] = from_f32<TO>(acc);
  }
}

// Register-blocked form of the same product for the plane sizes of the UNet: one thread owns the
// whole contracted line (A inputs) of 4 adjacent channels, so inputs are read ONCE with 8/16-byte
// coalesced loads and the matrix coefficients are wave-uniform scalar operands.
//   in element (line, k, c) at  in  + base_in(line)  + k * stride_k + c
//   out element (line, r, c) at out + base_out(line) + r * stride_k + c       (same stride)
// axis 0: line = (b, w): base_in = (b*A*Wd + w)*C, base_out = (b*ROUT*Wd + w)*C, stride = Wd*C
// axis 1: line = (b, i): base_in = (b*Ad + i)*A*C, base_out = (b*Ad + i)*ROUT*C, stride = C
template <typename TI, typename TO, int A, int ROUT>
__global__ void __launch_bounds__(256) k_axis_contract_reg(const TI* __restrict__ in, TO* __restrict__ out,
                                                           const float* __restrict__ M, int B, int Jd, int C,
                                                           int axis, float* __restrict__ stats = nullptr) {
  // stats (axis 1 only): a thread owns one output row of 4 channels, so the per-channel GroupNorm partial
  // sums of the stored values come for free with one split per output row: stats[b][Jd][C][2]
  const int nq = C / 4;
  const size_t total = (size_t)B * Jd * nq;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = 4 * (int)(i % nq);
    const size_t line = i / nq;           // b * Jd + j
    const int j = (int)(line % Jd);
    const int b = (int)(line / Jd);
    size_t bin, bout, stride;
    if (axis == 0) {
      bin = ((size_t)b * A * Jd + j) * C;
      bout = ((size_t)b * ROUT * Jd + j) * C;
      stride = (size_t)Jd * C;
    } else {
      bin = line * A * C;
      bout = line * ROUT * C;
      stride = C;
    }
    float v[A][4];
    float ps[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < A; ++k) load4<TI>(in + bin + k * stride + c, v[k][0], v[k][1], v[k][2], v[k][3]);
#pragma unroll 4
    for (int r = 0; r < ROUT; ++r) {
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
      for (int k = 0; k < A; ++k) {
        const float m = M[r * A + k];
        a0 = fmaf(m, v[k][0], a0);
        a1 = fmaf(m, v[k][1], a1);
        a2 = fmaf(m, v[k][2], a2);
        a3 = fmaf(m, v[k][3], a3);
      }
      store4<TO>(out + bout + r * stride + c, a0, a1, a2, a3);
      if (stats) {
        const float q0 = to_f32(from_f32<TO>(a0)), q1 = to_f32(from_f32<TO>(a1));
        const float q2 = to_f32(from_f32<TO>(a2)), q3 = to_f32(from_f32<TO>(a3));
        ps[0] += q0; ps[1] = fmaf(q0, q0, ps[1]);
        ps[2] += q1; ps[3] = fmaf(q1, q1, ps[3]);
        ps[4] += q2; ps[5] = fmaf(q2, q2, ps[5]);
        ps[6] += q3; ps[7] = fmaf(q3, q3, ps[7]);
      }
    }
    if (stats) {
      float* so = stats + (line * C + c) * 2;
      *reinterpret_cast<f32x4*>(so) = f32x4{ps[0], ps[1], ps[2], ps[3]};
      *reinterpret_cast<f32x4*>(so + 4) = f32x4{ps[4], ps[5], ps[6], ps[7]};
    }
  }
}


// ----------------------------------------------------------------------------- fused resample (N = 16 -> 32, 32 -> 16)
// y = M x M^T per (sample, channel) plane in ONE kernel for the two large alias-free resampling sites
// of the UNet (AliasFreeUpsample2D at 16 -> 32, AliasFreeDownsample2D at 32 -> 16): the two-pass VALU
// form (k_axis_contract_reg) round-trips an fp32 intermediate through HBM (81 / 163 MB of traffic for
// 31 / 63 MB of tensor).  Same structure as k_af_act_plane: the item's 16 channels are transposed into
// per-channel planes in LDS, each wave carries whole planes through  T^T = X^T M^T  (A = X^T rows from
// LDS, B = M) and  Y^T = M T^T  (chained: the accumulator is the next B operand), and the tile leaves
// through LDS as 16-byte stores.  Optionally emits the per-channel GroupNorm partial sums (S = 1).
template <typename T, int N, int R>
struct RsCfg {
  typedef Mma<T> MM;
  static constexpr int EPC = MM::EPC, KPF = MM::KPF;
  static constexpr int KH = ((N + KPF - 1) / KPF) * KPF, NKF1 = KH / KPF;
  static constexpr int TNI = N / 16, TR = R / 16;
  static constexpr int NW = 4, CPW = 4;
  static constexpr bool PERM = sizeof(T) == 2;
  static constexpr int KHP = KH + EPC;
  static constexpr int YRP = R * 16 + 8;
  static constexpr int XS = 16 * N * KHP, YS = R * YRP;
  static constexpr int REG = XS > YS ? XS : YS;
  static constexpr int LDS_BYTES = REG * (int)sizeof(T);
  static_assert(N % 16 == 0 && R % 16 == 0, "16-row MFMA tiles");
};

template <typename T, int N, int R>
__global__ void __launch_bounds__(256) k_resample_plane(const T* __restrict__ x, const float* __restrict__ M,
                                                        T* __restrict__ y, float* __restrict__ stats_out, int B, int C) {
  typedef RsCfg<T, N, R> CF;
  typedef Mma<T> MM;
  typedef typename MM::Chunk Chunk;
  constexpr int EPC = CF::EPC, KPF = CF::KPF, KH = CF::KH, KHP = CF::KHP, YRP = CF::YRP;
  constexpr int NKF1 = CF::NKF1, TNI = CF::TNI, TR = CF::TR, CPW = CF::CPW, NT = 256;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* Xs = reinterpret_cast<T*>(smem);
  T* Ys = Xs;   // the X planes are dead once every wave has read its fragments
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int ctiles = C / 16, nitems = B * ctiles;

  // constant fragments of M [R][N] (fp32 in global), once per workgroup
  Chunk mB[TR][NKF1], mA[TR][NKF1];
#pragma unroll
  for (int t = 0; t < TR; ++t)
#pragma unroll
    for (int f = 0; f < NKF1; ++f)
#pragma unroll
      for (int e = 0; e < EPC; ++e) {
        const int kb = af_kidx<T>(f, lg, e, false), ka = af_kidx<T>(f, lg, e, true);
        mB[t][f][e] = from_f32<T>(kb < N ? M[(16 * t + li) * N + kb] : 0.f);
        mA[t][f][e] = from_f32<T>(ka < N ? M[(16 * t + li) * N + ka] : 0.f);
      }

  constexpr int CQ = 16 / EPC, HQ = N / EPC, UNITS = N * CQ * HQ, UPT = (UNITS + NT - 1) / NT;
  for (int item = xcd_remap(blockIdx.x, gridDim.x); item < nitems; item += gridDim.x) {
    const int b = item / ctiles, c0 = (item - b * ctiles) * 16;
    __syncthreads();   // previous item's copy out of the (aliased) region is done
    if constexpr (KH > N) {
      for (int i = tid; i < N * 16 * (KH - N); i += NT) {
        const int row = i / (KH - N), k = N + (i - row * (KH - N));
        Xs[row * KHP + k] = from_f32<T>(0.f);
      }
    }
    // ---- tile -> Xs[c][w][h] (h K-contiguous): unit = EPC pixels (along h) x EPC channels
#pragma unroll
    for (int k = 0; k < UPT; ++k) {
      const int u = tid + k * NT;
      if (u < UNITS) {
        const int cq = u % CQ, w = (u / CQ) % N, hq = u / (CQ * N);
        Chunk pre[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e)
          pre[e] = ld16<Chunk>(x + ((size_t)(b * N + hq * EPC + e) * N + w) * C + c0 + cq * EPC);
#pragma unroll
        for (int cc = 0; cc < EPC; ++cc) {
          Chunk o;
#pragma unroll
          for (int e = 0; e < EPC; ++e) o[e] = pre[e][cc];
          st16<Chunk>(Xs + ((size_t)((cq * EPC + cc) * N + w)) * KHP + hq * EPC, o);
        }
      }
    }
    __syncthreads();
    f32x4 yacc[CPW][TR][TR];
#pragma unroll
    for (int pl = 0; pl < CPW; ++pl) {
      const int c = wave * CPW + pl;
      Chunk xa[TNI][NKF1];
#pragma unroll
      for (int tw = 0; tw < TNI; ++tw)
#pragma unroll
        for (int kf = 0; kf < NKF1; ++kf)
          xa[tw][kf] = ld16<Chunk>(Xs + ((size_t)(c * N + 16 * tw + li)) * KHP + kf * KPF + lg * EPC);
#pragma unroll
      for (int th = 0; th < TR; ++th) {
        f32x4 t1[TNI];        // rows w, columns h' (tile th)
#pragma unroll
        for (int tw = 0; tw < TNI; ++tw) {
          t1[tw] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int kf = 0; kf < NKF1; ++kf) MM::mma(t1[tw], xa[tw][kf], mB[th][kf]);
        }
        Chunk b2[NKF1];
        if constexpr (CF::PERM) {
#pragma unroll
          for (int f = 0; f < NKF1; ++f)
            b2[f] = pack_chain<T>(t1[2 * f], 2 * f + 1 < TNI ? t1[2 * f + 1 < TNI ? 2 * f + 1 : 0] : f32x4{0.f, 0.f, 0.f, 0.f});
        } else {
#pragma unroll
          for (int f = 0; f < NKF1; ++f) b2[f] = t1[f];
        }
#pragma unroll
        for (int tr = 0; tr < TR; ++tr) {   // rows w' (tile tr), columns h'
          yacc[pl][tr][th] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int f = 0; f < NKF1; ++f) MM::mma(yacc[pl][tr][th], mA[tr][f], b2[f]);
        }
      }
    }
    __syncthreads();   // every wave has its X fragments: the region becomes the output tile
#pragma unroll
    for (int pl = 0; pl < CPW; ++pl) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int tr = 0; tr < TR; ++tr)
#pragma unroll
        for (int th = 0; th < TR; ++th)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const T o = from_f32<T>(yacc[pl][tr][th][r]);
            Ys[(16 * th + li) * YRP + (16 * tr + 4 * lg + r) * 16 + wave * CPW + pl] = o;
            const float vr = to_f32(o);
            s1 += vr;
            s2 = fmaf(vr, vr, s2);
          }
      if (stats_out) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
          s1 += __shfl_xor(s1, o, 64);
          s2 += __shfl_xor(s2, o, 64);
        }
        if (lane == 0) *reinterpret_cast<f32x2*>(stats_out + ((size_t)b * C + c0 + wave * CPW + pl) * 2) = f32x2{s1, s2};
      }
    }
    __syncthreads();
    {
      constexpr int CPP = 16 / EPC;
      for (int i = tid; i < R * R * CPP; i += NT) {
        const int pix = i / CPP, q = i - pix * CPP;
        const int h = pix / R, w = pix - h * R;
        st16<Chunk>(y + ((size_t)b * R * R + pix) * C + c0 + q * EPC, ld16<Chunk>(Ys + h * YRP + w * 16 + q * EPC));
      }
    }
  }
}

template <typename T, int N, int R>
static int launch_resample_plane(const void* x, const float* M, void* y, float* stats, int B, int C, hipStream_t st) {
  typedef RsCfg<T, N, R> CF;
  static unsigned long long attr_set = 0;
  if (first_on_device(attr_set)) {
    (void)hipFuncSetAttribute((const void*)k_resample_plane<T, N, R>, hipFuncAttributeMaxDynamicSharedMemorySize, CF::LDS_BYTES);
  }
  const int nitems = B * (C / 16);
  const int per_cu = (160 * 1024) / CF::LDS_BYTES > 3 ? 3 : (160 * 1024) / CF::LDS_BYTES;
  int grid = 256 * per_cu;
  if (grid > nitems) grid = nitems;
  k_resample_plane<T, N, R><<<grid, 256, CF::LDS_BYTES, st>>>((const T*)x, M, (T*)y, stats, B, C);
  return check_launch("afldm_af_resample_plane");
}

// ----------------------------------------------------------------------------- launchers
// resident (persistent) workgroups of the plane kernel per CU: LDS-bound, and at most 2 (N = 32) /
// 4 (N = 16) four-wave workgroups by the register budget
template <typename T, int N, int CH>
static int af_plane_wgs_per_cu() {
  const int by_lds = (160 * 1024) / PlaneCfg<T, N, CH>::LDS_BYTES;
  const int by_regs = N == 32 ? (CH == 8 && sizeof(T) == 2 ? 3 : 2) : 4;
  return by_lds < 1 ? 1 : (by_lds < by_regs ? by_lds : by_regs);
}
template <typename T, int N, int CH>
static int launch_af_plane(const AfP<T>& p, int cus, hipStream_t st) {
  typedef PlaneCfg<T, N, CH> CF;
  static unsigned long long attr_set = 0;
  if (first_on_device(attr_set)) {
    (void)hipFuncSetAttribute((const void*)k_af_act_plane<T, N, CH>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              CF::LDS_BYTES);
  }
  const int nitems = p.B * ((p.C1 + p.C2) / CH);
  int grid = cus * af_plane_wgs_per_cu<T, N, CH>();
  if (grid > nitems) grid = nitems;
  k_af_act_plane<T, N, CH><<<grid, CF::NW * 64, CF::LDS_BYTES, st>>>(p);
  return check_launch("afldm_af_act(plane)");
}

template <typename T, int N>
static int launch_af_mfma(const AfP<T>& p, hipStream_t st) {
  static int cus = 0;
  if (!cus) {
    hipDeviceProp_t prop;
    int dev = 0;
    (void)hipGetDevice(&dev);
    cus = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
  }
  // Items are indivisible, so a launch takes ceil(items / resident workgroups) item times: pick the
  // item size (16 or 8 channels) with the smaller makespan (an 8-channel item measured 0.62-0.82 of a 16-channel one).
  const int Ct = p.C1 + p.C2;
  const long long slots16 = (long long)cus * af_plane_wgs_per_cu<T, N, 16>(), slots8 = (long long)cus * af_plane_wgs_per_cu<T, N, 8>();
  const long long items16 = (long long)p.B * (Ct / 16), items8 = 2 * items16;
  const double t16 = (double)((items16 + slots16 - 1) / slots16), t8 = 0.65 * (double)((items8 + slots8 - 1) / slots8);
  // (8-channel items read 16-byte pieces of every pixel: only worth it while the tensor stays
  //  cache-resident between the two item passes over a line - measured 1.15x slower per round at 75 MB)
  const size_t bytes = (size_t)p.B * N * N * Ct * sizeof(T);
  static const int s_ch = getenv("AFLDM_AF_CH") ? atoi(getenv("AFLDM_AF_CH")) : 0;      // A/B: force the item size (8 / 16)
  if (s_ch == 8 && p.C1 % 8 == 0) return launch_af_plane<T, N, 8>(p, cus, st);
  if (s_ch == 16) return launch_af_plane<T, N, 16>(p, cus, st);
  if (t8 < t16 && p.C1 % 8 == 0 && bytes <= (40u << 20)) return launch_af_plane<T, N, 8>(p, cus, st);
  return launch_af_plane<T, N, 16>(p, cus, st);
}
template <typename T, int N>
static int launch_af_kron(const AfP<T>& p, hipStream_t st) {
  typedef KronCfg<T, N> CF;
  static unsigned long long attr_set = 0;
  if (first_on_device(attr_set)) {
    (void)hipFuncSetAttribute((const void*)k_af_act_kron<T, N>, hipFuncAttributeMaxDynamicSharedMemorySize, CF::LDS_BYTES);
  }
  const int nitems = p.B * ((p.C1 + p.C2) / 16);
  const int ngroups = (nitems + 3) / 4;
  const int per_cu = (160 * 1024) / CF::LDS_BYTES >= 2 ? 2 : 1;
  int grid = 256 * per_cu;
  if (grid > ngroups) grid = ngroups;
  k_af_act_kron<T, N><<<grid, 256, CF::LDS_BYTES, st>>>(p);
  return check_launch("afldm_af_act(kron)");
}

static int launch_af_p8(const AfP<bf16>& p, hipStream_t st) {
  const int nitems = p.B * ((p.C1 + p.C2) / 16);
  k_af_act_p8<<<(nitems + 3) / 4, 256, 0, st>>>(p);
  return check_launch("afldm_af_act(p8)");
}

template <typename T, int N>
static int launch_af_small(const AfP<T>& p, hipStream_t st) {
  const int total = p.B * (p.C1 + p.C2);
  const int grid = (total + 255) / 256;
  k_af_act_small<T, N><<<grid, 256, 0, st>>>(p);
  return check_launch("afldm_af_act(small)");
}

template <typename T>
static int af_act_dispatch(const void* x1, int C1, const void* x2, int C2, const GnStats& gs, const float* gamma,
                           const float* beta, int G, float eps, const float* U, const float* D, const void* packed,
                           void* y, int B, int N, hipStream_t st, int x_layout = 0, int y_layout = 0) {
  AfP<T> p;
  p.packed = packed;
  p.eps = eps;
  p.x1 = (const T*)x1; p.x2 = (const T*)x2; p.gs = gs; p.gamma = gamma; p.beta = beta;
  p.U = U; p.D = D; p.y = (T*)y; p.C1 = C1; p.C2 = C2; p.G = G; p.B = B;
  p.trace = g_af_trace;
  static const int s_stagger = getenv("AFLDM_AF_STAGGER") ? atoi(getenv("AFLDM_AF_STAGGER")) : 0;
  p.stagger = s_stagger;
  p.y_blocked = y_layout == 1 ? 1 : (y_layout == 2 && N == 2 ? 2 : 0);      // 2: plane-constant output of the 2 x 2 kernel
  p.x_blocked = x_layout == 1 ? 1 : 0;
  // bit mask 4 / 8: plane sizes run on the VALU kernel instead of the Kronecker MFMA kernel.  N = 4 (default): one thread
  // per plane with the loop over the upsampled rows fully unrolled (all 64 coefficients in SGPRs) beats the MFMA form,
  // whose workgroups each stage a 64 KB constant image: 5.251 -> 5.229 ms/step (same box).  N = 8 needs 256 coefficients
  // per row step and stays on MFMA (VALU: 5.25 -> 5.51).
  static const int s_small = getenv("AFLDM_AF_SMALL_N") ? atoi(getenv("AFLDM_AF_SMALL_N")) : 4;
  switch (N) {
    case 2: return launch_af_small<T, 2>(p, st);
    case 4:
      if (packed && p.C1 % 16 == 0 && p.C2 % 16 == 0 && !(s_small & 4)) return launch_af_kron<T, 4>(p, st);
      return launch_af_small<T, 4>(p, st);
    case 8:
      if constexpr (sizeof(T) == 2) {
        // bf16: the separable two-planes-per-tile kernel (no constant image); AFLDM_AF_P8=0 -> the Kronecker kernel (A/B)
        static const bool p8 = !(getenv("AFLDM_AF_P8") && atoi(getenv("AFLDM_AF_P8")) == 0);
        if (p8 && p.U && p.D && p.C1 % 16 == 0 && p.C2 % 16 == 0 && !(s_small & 8)) return launch_af_p8(p, st);
      }
      if (packed && p.C1 % 16 == 0 && p.C2 % 16 == 0 && !(s_small & 8)) return launch_af_kron<T, 8>(p, st);
      return launch_af_small<T, 8>(p, st);
    case 16: return launch_af_mfma<T, 16>(p, st);
    case 32: return launch_af_mfma<T, 32>(p, st);
  }
  set_error("afldm_af_act: plane size N=%d not in {2,4,8,16,32}", N);
  return AFLDM_ESHAPE;
}

// y = M x M^T per (sample, channel) plane for the SMALL resampling sites (2 <-> 4, 4 <-> 8, 8 <-> 16) in ONE launch (round 5):
// one thread per plane, lanes = channels (the coefficients are wave-uniform scalar operands).  Exactly the two passes of
// k_axis_contract_reg - H first into an fp32 intermediate, then W, every sum an fmaf chain over ascending k from 0 - with
// the intermediate in registers instead of an fp32 tensor in memory and a second launch: bit-identical results and
// statistics (one split per output row).  A <= 8: the plane lives in registers, output row by output row; A = 16 (-> 8): the
// output accumulates column by column (64 accumulators) - correct but 3x slower than the two passes (128 coefficients do not fit the
// scalar registers), not dispatched.
// SPLIT: threads per plane (A <= 8 form): thread `part` of a plane computes the output rows [part R / SPLIT, (part + 1) R / SPLIT) -
// rows are independent chains, so the values are the same bit for bit; the 8 -> 16 site (24 576 planes at batch 64, 3 072 fmas each)
// was 96 workgroups of one-wave-per-SIMD serial work (17.4 us)
template <typename T, int A, int R, int SPLIT = 1>
__global__ void __launch_bounds__(256) k_resample_small(const T* __restrict__ in, T* __restrict__ out, const float* __restrict__ M,
                                                        int B, int C, float* __restrict__ stats) {
  static_assert(SPLIT == 1 || (A <= 8 && R % SPLIT == 0), "row split");
  const size_t total = (size_t)B * C * SPLIT;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % C), bp = (int)(i / C), b = bp / SPLIT, part = bp - b * SPLIT;
  const T* xp = in + (size_t)b * A * A * C + c;
  T* yp = out + (size_t)b * R * R * C + c;
  if constexpr (A <= 8) {
    float x[A][A];
#pragma unroll
    for (int h = 0; h < A; ++h)
#pragma unroll
      for (int w = 0; w < A; ++w) x[h][w] = to_f32(xp[(size_t)(h * A + w) * C]);
#pragma unroll
    for (int rr = 0; rr < R / SPLIT; ++rr) {
      const int r = part * (R / SPLIT) + rr;
      float t[A];
#pragma unroll
      for (int w = 0; w < A; ++w) {
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < A; ++k) a = fmaf(M[r * A + k], x[k][w], a);
        t[w] = a;
      }
      float p1 = 0.f, p2 = 0.f;
#pragma unroll
      for (int s_ = 0; s_ < R; ++s_) {
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < A; ++k) a = fmaf(M[s_ * A + k], t[k], a);
        const T o = from_f32<T>(a);
        yp[(size_t)(r * R + s_) * C] = o;
        const float q = to_f32(o);
        p1 += q;
        p2 = fmaf(q, q, p2);
      }
      if (stats) *reinterpret_cast<f32x2*>(stats + (((size_t)b * R + r) * C + c) * 2) = f32x2{p1, p2};
    }
  } else {
    float yacc[R][R];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int s_ = 0; s_ < R; ++s_) yacc[r][s_] = 0.f;
#pragma unroll 2
    for (int w = 0; w < A; ++w) {
      float xc[A];
#pragma unroll
      for (int h = 0; h < A; ++h) xc[h] = to_f32(xp[(size_t)(h * A + w) * C]);
#pragma unroll
      for (int r = 0; r < R; ++r) {
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < A; ++k) a = fmaf(M[r * A + k], xc[k], a);
#pragma unroll
        for (int s_ = 0; s_ < R; ++s_) yacc[r][s_] = fmaf(M[s_ * A + w], a, yacc[r][s_]);
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      float p1 = 0.f, p2 = 0.f;
#pragma unroll
      for (int s_ = 0; s_ < R; ++s_) {
        const T o = from_f32<T>(yacc[r][s_]);
        yp[(size_t)(r * R + s_) * C] = o;
        const float q = to_f32(o);
        p1 += q;
        p2 = fmaf(q, q, p2);
      }
      if (stats) *reinterpret_cast<f32x2*>(stats + (((size_t)b * R + r) * C + c) * 2) = f32x2{p1, p2};
    }
  }
}

template <typename T>
static int resample_dispatch(const void* x, const float* M, void* y, float* ws, int B, int N, int C, int Rout,
                             hipStream_t st, float* stats = nullptr) {
  // pass 1 contracts H into the fp32 workspace [B][Rout][N][C]; pass 2 contracts W
  {
    // the small sites in one launch (k_resample_small; AFLDM_NO_RESAMPLE_SMALL=1: the two-pass form, for A/B)
    static const bool s_off = getenv("AFLDM_NO_RESAMPLE_SMALL") && atoi(getenv("AFLDM_NO_RESAMPLE_SMALL")) != 0;
    const int grid = (int)(((size_t)B * C + 255) / 256);
    static const int s_split = getenv("AFLDM_RESAMPLE_SPLIT") ? atoi(getenv("AFLDM_RESAMPLE_SPLIT")) : 4;      // threads per plane at 8 -> 16 (1 / 2 / 4; A/B)
    if (!s_off && N == 8 && Rout == 16 && s_split > 1) {
      if (s_split == 2) k_resample_small<T, 8, 16, 2><<<(int)(((size_t)B * C * 2 + 255) / 256), 256, 0, st>>>((const T*)x, (T*)y, M, B, C, stats);
      else k_resample_small<T, 8, 16, 4><<<(int)(((size_t)B * C * 4 + 255) / 256), 256, 0, st>>>((const T*)x, (T*)y, M, B, C, stats);
      return check_launch("afldm_af_resample(small)");
    }
#define AFLDM_RSS(A_, R_)                                                                                     \
  if (!s_off && N == A_ && Rout == R_) {                                                                      \
    k_resample_small<T, A_, R_><<<grid, 256, 0, st>>>((const T*)x, (T*)y, M, B, C, stats);                    \
    return check_launch("afldm_af_resample(small)");                                                          \
  }
    AFLDM_RSS(2, 4) AFLDM_RSS(4, 8) AFLDM_RSS(8, 16) AFLDM_RSS(4, 2) AFLDM_RSS(8, 4)      // (16 -> 8 measured 3x SLOWER this way: 47.6 vs 14.7 us)
#undef AFLDM_RSS
  }
  if (C % 4 == 0 && (Rout == 2 * N || 2 * Rout == N) && N >= 2 && N <= 32 && (N & (N - 1)) == 0) {
    const size_t t1 = (size_t)B * N * (C / 4), t2 = (size_t)B * Rout * (C / 4);
    const int g1 = (int)((t1 + 255) / 256 < 8192 ? (t1 + 255) / 256 : 8192);
    const int g2 = (int)((t2 + 255) / 256 < 8192 ? (t2 + 255) / 256 : 8192);
#define AFLDM_RS(A_, R_)                                                                                      \
  if (N == A_ && Rout == R_) {                                                                                \
    k_axis_contract_reg<T, float, A_, R_><<<g1, 256, 0, st>>>((const T*)x, ws, M, B, N, C, 0);                \
    k_axis_contract_reg<float, T, A_, R_><<<g2, 256, 0, st>>>(ws, (T*)y, M, B, Rout, C, 1, stats);            \
    return check_launch("afldm_af_resample(reg)");                                                            \
  }
    AFLDM_RS(2, 4) AFLDM_RS(4, 8) AFLDM_RS(8, 16) AFLDM_RS(16, 32)
    AFLDM_RS(4, 2) AFLDM_RS(8, 4) AFLDM_RS(16, 8) AFLDM_RS(32, 16)
#undef AFLDM_RS
  }
  if (stats) {
    set_error("afldm_af_lpf_down2: statistics are emitted by the register-blocked path only (N = %d, C = %d)", N, C);
    return AFLDM_ESHAPE;
  }
  size_t n1 = (size_t)B * Rout * N * C, n2 = (size_t)B * Rout * Rout * C;
  int g1 = (int)((n1 + 255) / 256 < 8192 ? (n1 + 255) / 256 : 8192);
  int g2 = (int)((n2 + 255) / 256 < 8192 ? (n2 + 255) / 256 : 8192);
  k_axis_contract<T, float><<<g1, 256, 0, st>>>((const T*)x, ws, M, B, N, N, C, Rout, 0);
  k_axis_contract<float, T><<<g2, 256, 0, st>>>(ws, (T*)y, M, B, Rout, N, C, Rout, 1);
  return check_launch("afldm_af_resample");
}

}  // namespace afldm

using namespace afldm;

template <typename T>
static int af_act_slabs_launch(const float* slabs, int nslab, const float* bias, const void* temb, int temb_stride,
                               const void* residual, void* y_raw, const float* gamma, const float* beta, int G, float eps,
                               int act, const float* U, const float* D, void* y, int B, int C, int N, hipStream_t st) {
  AfSlabP<T> p;
  p.slabs = slabs; p.bias = bias; p.temb = (const T*)temb; p.gamma = gamma; p.beta = beta; p.U = U; p.D = D; p.y = (T*)y;
  p.residual = (const T*)residual; p.y_raw = (T*)y_raw;
  p.nslab = nslab; p.temb_stride = temb_stride; p.B = B; p.C = C; p.G = G; p.eps = eps;
  p.compact = act == 2 ? 1 : 0;
  const int cpg = C / G;
  int gpb = 0;
  for (int k = 1; k <= G && k <= 32; ++k)
    if (G % k == 0 && k * cpg <= 256) gpb = k;           // whole groups per workgroup, as many as 256 threads hold
  AFLDM_REQUIRE(gpb > 0, AFLDM_ESHAPE, "afldm_af_act_slabs: C/G = %d channels per group do not fit a workgroup", cpg);
  p.gpb = gpb;
  const int grid = B * (G / gpb);
  if (N == 2 && act) k_af_act_slabs<T, 2, 1><<<grid, gpb * cpg, 0, st>>>(p);
  else if (N == 2) k_af_act_slabs<T, 2, 0><<<grid, gpb * cpg, 0, st>>>(p);
  else if (act) k_af_act_slabs<T, 4, 1><<<grid, gpb * cpg, 0, st>>>(p);
  else k_af_act_slabs<T, 4, 0><<<grid, gpb * cpg, 0, st>>>(p);
  return check_launch("afldm_af_act_slabs");
}

extern "C" int afldm_af_act_slabs(const float* slabs, int nslab, const float* bias, const void* temb, int temb_stride,
                                  const void* residual, void* y_raw, const float* gamma, const float* beta, int G, float eps,
                                  int act, const float* U, const float* D, void* y, int B, int C, int N, int dtype,
                                  afldm_stream_t stream) {
  AFLDM_REQUIRE(slabs && gamma && beta && y && (!act || (U && D)), AFLDM_ENULL, "afldm_af_act_slabs: NULL pointer");
  AFLDM_REQUIRE(N == 2 || N == 4, AFLDM_ESHAPE, "afldm_af_act_slabs: plane size N=%d not in {2,4}", N);
  AFLDM_REQUIRE(act >= 0 && act <= 2 && (act != 2 || N == 2), AFLDM_ESHAPE, "afldm_af_act_slabs: act=%d (2 = plane-constant output, N = 2 only)", act);
  AFLDM_REQUIRE(B > 0 && C > 0 && G > 0 && C % G == 0 && nslab >= 1 && nslab <= 64, AFLDM_ESHAPE,
                "afldm_af_act_slabs: bad shape B=%d C=%d G=%d nslab=%d", B, C, G, nslab);
  AFLDM_REQUIRE(!temb || temb_stride == 0 || temb_stride >= C, AFLDM_ESHAPE, "afldm_af_act_slabs: temb_stride=%d (0 = one row for all samples, else >= C=%d)", temb_stride, C);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == AFLDM_F32) return af_act_slabs_launch<float>(slabs, nslab, bias, temb, temb_stride, residual, y_raw, gamma, beta, G, eps, act, U, D, y, B, C, N, st);
  if (dtype == AFLDM_BF16) return af_act_slabs_launch<bf16>(slabs, nslab, bias, temb, temb_stride, residual, y_raw, gamma, beta, G, eps, act, U, D, y, B, C, N, st);
  set_error("afldm_af_act_slabs: unknown dtype %d", dtype);
  return AFLDM_EDTYPE;
}

extern "C" int afldm_af_act_trace(void* buf) {
  g_af_trace = (unsigned long long*)buf;
  return AFLDM_OK;
}

extern "C" int afldm_af_act(const void* x1, int C1, const void* x2, int C2, const float* stats1, int S1,
                            const float* stats2, int S2, const float* gamma, const float* beta, int G, float eps,
                            const float* U, const float* D, const void* packed, void* y, int B, int N, int dtype,
                            afldm_stream_t stream) {
  AFLDM_REQUIRE(x1 && U && D && y, AFLDM_ENULL, "afldm_af_act: NULL pointer");
  AFLDM_REQUIRE(N < 16 || packed, AFLDM_ENULL, "afldm_af_act: N=%d needs the packed filter image (afldm_af_pack)", N);
  if ((N == 4 || N == 8) && packed && C1 % 16 == 0 && C2 % 16 == 0)
    AFLDM_REQUIRE(aligned16(x1) && aligned16(x2) && aligned16(y), AFLDM_EALIGN, "afldm_af_act: pointers must be 16-byte aligned");
  AFLDM_REQUIRE(C1 > 0 && C2 >= 0 && (C2 == 0 || x2), AFLDM_ESHAPE, "afldm_af_act: bad C1=%d C2=%d", C1, C2);
  AFLDM_REQUIRE(B > 0, AFLDM_ESHAPE, "afldm_af_act: B=%d", B);
  AFLDM_REQUIRE(!stats1 || (gamma && beta && G > 0 && (C1 + C2) % G == 0 && S1 > 0 && (C2 == 0 || (stats2 && S2 > 0))),
                AFLDM_ESHAPE, "afldm_af_act: GroupNorm fusion needs gamma/beta, statistics of both tensors and C %% G == 0 (C=%d G=%d)",
                C1 + C2, G);
  if (N >= 16) {
    AFLDM_REQUIRE(C1 % 16 == 0 && C2 % 16 == 0, AFLDM_ESHAPE, "afldm_af_act: C1=%d / C2=%d must be multiples of 16 for N=%d",
                  C1, C2, N);
    AFLDM_REQUIRE(aligned16(x1) && aligned16(x2) && aligned16(y), AFLDM_EALIGN, "afldm_af_act: pointers must be 16-byte aligned");
  }
  hipStream_t st = (hipStream_t)stream;
  const GnStats gs{stats1, stats2, C1, C2, S1, S2};
  if (dtype == AFLDM_F32) return af_act_dispatch<float>(x1, C1, x2, C2, gs, gamma, beta, G, eps, U, D, packed, y, B, N, st);
  if (dtype == AFLDM_BF16) return af_act_dispatch<bf16>(x1, C1, x2, C2, gs, gamma, beta, G, eps, U, D, packed, y, B, N, st);
  set_error("afldm_af_act: unknown dtype %d", dtype);
  return AFLDM_EDTYPE;
}

// afldm_af_act with the tensors in 8-channel blocks (layout 1 of afldm_conv_args): N = 16 / 32 (the plane kernel), bf16
extern "C" int afldm_af_act_c8(const void* x1, int C1, const void* x2, int C2, const float* stats1, int S1,
                               const float* stats2, int S2, const float* gamma, const float* beta, int G, float eps,
                               const float* U, const float* D, const void* packed, void* y, int B, int N, int dtype,
                               int x_layout, int y_layout, afldm_stream_t stream) {
  AFLDM_REQUIRE(x1 && U && D && y && packed, AFLDM_ENULL, "afldm_af_act_c8: NULL pointer");
  AFLDM_REQUIRE(dtype == AFLDM_BF16 && (N == 16 || N == 32), AFLDM_ESHAPE, "afldm_af_act_c8: bf16 planes of 16^2 / 32^2 only (N=%d dtype=%d)", N, dtype);
  AFLDM_REQUIRE((x_layout == 0 || x_layout == 1) && (y_layout == 0 || y_layout == 1) && (x_layout == 0 || C2 == 0), AFLDM_ESHAPE,
                "afldm_af_act_c8: layouts are 0 / 1, a blocked input is a single tensor");
  AFLDM_REQUIRE(C1 > 0 && C2 >= 0 && (C2 == 0 || x2) && C1 % 16 == 0 && C2 % 16 == 0 && B > 0, AFLDM_ESHAPE, "afldm_af_act_c8: bad C1=%d C2=%d B=%d", C1, C2, B);
  AFLDM_REQUIRE(!stats1 || (gamma && beta && G > 0 && (C1 + C2) % G == 0 && S1 > 0 && (C2 == 0 || (stats2 && S2 > 0))), AFLDM_ESHAPE,
                "afldm_af_act_c8: GroupNorm fusion needs gamma/beta, statistics of both tensors and C %% G == 0 (C=%d G=%d)", C1 + C2, G);
  AFLDM_REQUIRE(aligned16(x1) && aligned16(x2) && aligned16(y), AFLDM_EALIGN, "afldm_af_act_c8: pointers must be 16-byte aligned");
  const GnStats gs{stats1, stats2, C1, C2, S1, S2};
  return af_act_dispatch<bf16>(x1, C1, x2, C2, gs, gamma, beta, G, eps, U, D, packed, y, B, N, (hipStream_t)stream, x_layout, y_layout);
}

// afldm_af_act on 2 x 2 planes with the plane-constant result stored once (y [B][C1 + C2]): see k_af_act_small
extern "C" int afldm_af_act_const2(const void* x1, int C1, const void* x2, int C2, const float* stats1, int S1,
                                   const float* stats2, int S2, const float* gamma, const float* beta, int G, float eps,
                                   const float* U, const float* D, void* y, int B, int dtype, afldm_stream_t stream) {
  AFLDM_REQUIRE(x1 && U && D && y, AFLDM_ENULL, "afldm_af_act_const2: NULL pointer");
  AFLDM_REQUIRE(C1 > 0 && C2 >= 0 && (C2 == 0 || x2) && B > 0, AFLDM_ESHAPE, "afldm_af_act_const2: bad C1=%d C2=%d B=%d", C1, C2, B);
  AFLDM_REQUIRE(!stats1 || (gamma && beta && G > 0 && (C1 + C2) % G == 0 && S1 > 0 && (C2 == 0 || (stats2 && S2 > 0))), AFLDM_ESHAPE,
                "afldm_af_act_const2: GroupNorm fusion needs gamma/beta, statistics of both tensors and C %% G == 0 (C=%d G=%d)", C1 + C2, G);
  const GnStats gs{stats1, stats2, C1, C2, S1, S2};
  hipStream_t st = (hipStream_t)stream;
  if (dtype == AFLDM_F32) return af_act_dispatch<float>(x1, C1, x2, C2, gs, gamma, beta, G, eps, U, D, nullptr, y, B, 2, st, 0, 2);
  if (dtype == AFLDM_BF16) return af_act_dispatch<bf16>(x1, C1, x2, C2, gs, gamma, beta, G, eps, U, D, nullptr, y, B, 2, st, 0, 2);
  set_error("afldm_af_act_const2: unknown dtype %d", dtype);
  return AFLDM_EDTYPE;
}

extern "C" size_t afldm_af_pack_bytes(int N, int dtype) {
  if (N == 4) return dtype == AFLDM_F32 ? KronCfg<float, 4>::CONST_ELEMS * 4 : KronCfg<bf16, 4>::CONST_ELEMS * 2;
  if (N == 8) return dtype == AFLDM_F32 ? KronCfg<float, 8>::CONST_ELEMS * 4 : KronCfg<bf16, 8>::CONST_ELEMS * 2;
  if (N == 16) return dtype == AFLDM_F32 ? PlaneCfg<float, 16>::CONST_ELEMS * 4 : PlaneCfg<bf16, 16>::CONST_ELEMS * 2;
  if (N == 32) return dtype == AFLDM_F32 ? PlaneCfg<float, 32>::CONST_ELEMS * 4 : PlaneCfg<bf16, 32>::CONST_ELEMS * 2;
  return 0;
}

extern "C" int afldm_af_pack(const float* U, const float* D, int N, int dtype, void* packed, afldm_stream_t stream) {
  AFLDM_REQUIRE(U && D && packed, AFLDM_ENULL, "afldm_af_pack: NULL pointer");
  AFLDM_REQUIRE(N == 4 || N == 8 || N == 16 || N == 32, AFLDM_ESHAPE, "afldm_af_pack: N=%d (only the MFMA plane sizes 4..32 are packed)", N);
  AFLDM_REQUIRE(dtype == AFLDM_F32 || dtype == AFLDM_BF16, AFLDM_EDTYPE, "afldm_af_pack: unknown dtype %d", dtype);
  hipStream_t st = (hipStream_t)stream;
  if (N == 4 && dtype == AFLDM_F32) k_af_pack_kron<float, 4><<<16, 256, 0, st>>>(U, D, (float*)packed);
  if (N == 8 && dtype == AFLDM_F32) k_af_pack_kron<float, 8><<<64, 256, 0, st>>>(U, D, (float*)packed);
  if (N == 4 && dtype == AFLDM_BF16) k_af_pack_kron<bf16, 4><<<16, 256, 0, st>>>(U, D, (bf16*)packed);
  if (N == 8 && dtype == AFLDM_BF16) k_af_pack_kron<bf16, 8><<<64, 256, 0, st>>>(U, D, (bf16*)packed);
  if (N == 16 && dtype == AFLDM_F32) k_af_pack<float, 16><<<16, 256, 0, st>>>(U, D, (float*)packed);
  if (N == 32 && dtype == AFLDM_F32) k_af_pack<float, 32><<<16, 256, 0, st>>>(U, D, (float*)packed);
  if (N == 16 && dtype == AFLDM_BF16) k_af_pack<bf16, 16><<<16, 256, 0, st>>>(U, D, (bf16*)packed);
  if (N == 32 && dtype == AFLDM_BF16) k_af_pack<bf16, 32><<<16, 256, 0, st>>>(U, D, (bf16*)packed);
  return check_launch("afldm_af_pack");
}

extern "C" int afldm_af_up2(const void* x, const float* U, void* y, float* workspace, int B, int N, int C, int dtype,
                            afldm_stream_t stream) {
  AFLDM_REQUIRE(x && U && y && workspace, AFLDM_ENULL, "afldm_af_up2: NULL pointer");
  AFLDM_REQUIRE(B > 0 && N > 0 && C > 0, AFLDM_ESHAPE, "afldm_af_up2: bad shape");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == AFLDM_F32) return resample_dispatch<float>(x, U, y, workspace, B, N, C, 2 * N, st);
  if (dtype == AFLDM_BF16) return resample_dispatch<bf16>(x, U, y, workspace, B, N, C, 2 * N, st);
  set_error("afldm_af_up2: unknown dtype %d", dtype);
  return AFLDM_EDTYPE;
}

extern "C" int afldm_af_lpf_down2(const void* x, const float* D, void* y, float* workspace, float* stats_out, int B, int N,
                                  int C, int dtype, afldm_stream_t stream) {
  AFLDM_REQUIRE(x && D && y && workspace, AFLDM_ENULL, "afldm_af_lpf_down2: NULL pointer");
  AFLDM_REQUIRE(B > 0 && N >= 2 && N % 2 == 0 && C > 0, AFLDM_ESHAPE, "afldm_af_lpf_down2: bad shape (N=%d must be even)", N);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == AFLDM_F32) return resample_dispatch<float>(x, D, y, workspace, B, N, C, N / 2, st, stats_out);
  if (dtype == AFLDM_BF16) return resample_dispatch<bf16>(x, D, y, workspace, B, N, C, N / 2, st, stats_out);
  set_error("afldm_af_lpf_down2: unknown dtype %d", dtype);
  return AFLDM_EDTYPE;
}

extern "C" int afldm_af_resample(const void* x, const float* M, void* y, float* workspace, int B, int N, int C, int R,
                                 int dtype, afldm_stream_t stream) {
  AFLDM_REQUIRE(x && M && y && workspace, AFLDM_ENULL, "afldm_af_resample: NULL pointer");
  AFLDM_REQUIRE(B > 0 && N > 0 && C > 0 && R > 0, AFLDM_ESHAPE, "afldm_af_resample: bad shape");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == AFLDM_F32) return resample_dispatch<float>(x, M, y, workspace, B, N, C, R, st);
  if (dtype == AFLDM_BF16) return resample_dispatch<bf16>(x, M, y, workspace, B, N, C, R, st);
  set_error("afldm_af_resample: unknown dtype %d", dtype);
  return AFLDM_EDTYPE;
}

extern "C" int afldm_af_resample_hw(const void* x, const float* Mh, const float* Mw, void* y, float* workspace, int B,
                                    int N, int C, int R, int dtype, afldm_stream_t stream) {
  AFLDM_REQUIRE(x && Mh && Mw && y && workspace, AFLDM_ENULL, "afldm_af_resample_hw: NULL pointer");
  AFLDM_REQUIRE(B > 0 && N > 0 && C > 0 && R > 0, AFLDM_ESHAPE, "afldm_af_resample_hw: bad shape");
  hipStream_t st = (hipStream_t)stream;
  const size_t n1 = (size_t)B * R * N * C, n2 = (size_t)B * R * R * C;
  const int g1 = (int)((n1 + 255) / 256 < 8192 ? (n1 + 255) / 256 : 8192);
  const int g2 = (int)((n2 + 255) / 256 < 8192 ? (n2 + 255) / 256 : 8192);
  if (dtype == AFLDM_F32) {
    k_axis_contract<float, float><<<g1, 256, 0, st>>>((const float*)x, workspace, Mh, B, N, N, C, R, 0);
    k_axis_contract<float, float><<<g2, 256, 0, st>>>(workspace, (float*)y, Mw, B, R, N, C, R, 1);
  } else if (dtype == AFLDM_BF16) {
    k_axis_contract<bf16, float><<<g1, 256, 0, st>>>((const bf16*)x, workspace, Mh, B, N, N, C, R, 0);
    k_axis_contract<float, bf16><<<g2, 256, 0, st>>>(workspace, (bf16*)y, Mw, B, R, N, C, R, 1);
  } else {
    set_error("afldm_af_resample_hw: unknown dtype %d", dtype);
    return AFLDM_EDTYPE;
  }
  return check_launch("afldm_af_resample_hw");
}

extern "C" int afldm_af_resample_plane(const void* x, const float* M, void* y, float* stats_out, int B, int N, int C,
                                       int R, int dtype, afldm_stream_t stream) {
  AFLDM_REQUIRE(x && M && y, AFLDM_ENULL, "afldm_af_resample_plane: NULL pointer");
  AFLDM_REQUIRE(B > 0 && C > 0 && C % 16 == 0 && ((N == 16 && R == 32) || (N == 32 && R == 16)), AFLDM_ESHAPE,
                "afldm_af_resample_plane: covers N=16->32 and N=32->16 with C %% 16 == 0 (got N=%d R=%d C=%d)", N, R, C);
  AFLDM_REQUIRE(aligned16(x) && aligned16(y), AFLDM_EALIGN, "afldm_af_resample_plane: pointers must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == AFLDM_BF16)
    return N == 16 ? launch_resample_plane<bf16, 16, 32>(x, M, y, stats_out, B, C, st)
                   : launch_resample_plane<bf16, 32, 16>(x, M, y, stats_out, B, C, st);
  if (dtype == AFLDM_F32)
    return N == 16 ? launch_resample_plane<float, 16, 32>(x, M, y, stats_out, B, C, st)
                   : launch_resample_plane<float, 32, 16>(x, M, y, stats_out, B, C, st);
  set_error("afldm_af_resample_plane: unknown dtype %d", dtype);
  return AFLDM_EDTYPE;
}
