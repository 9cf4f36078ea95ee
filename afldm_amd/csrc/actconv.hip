// actconv.hip - MERGED LAUNCH (round 5): [GroupNorm ->] WarpedNonlinearity -> 3x3 convolution of a ResnetBlock2D
// (reference af_blocks.py:19-28 inside diffusers ResnetBlock2D.forward: `hidden_states = self.nonlinearity(self.norm1(x))`
// followed by `self.conv1(...)`, and the norm2 -> nonlinearity -> conv2 pair) as ONE kernel at the 32^2 / 16^2 levels.
//
// Why: at batch 64 a 32^2 / 16^2 halo-patch convolution has exactly one tile per CU, so nothing inside the launch overlaps
// its fixed part (launch ramp + first patch from HBM, ~4 us) and nothing overlaps the activation launch's tail in front of
// it (profiles/r04/conv_fixed_cost.txt, af_plane_trace.txt; two independent jobs in flight recover 16 % of the step by
// filling exactly these holes, two_streams_ab.txt).  The two kernels exchange a tensor that only they touch: here the
// workgroups that will convolve sample b's tiles FIRST run that sample's activation items and hand the activated tensor
// over inside the launch.
//
//   cluster   : the CS workgroups that own the tiles of ONE sample (4 at 32^2: 4 row tiles x 1 cout tile; 4 at 16^2 with
//               384 couts: 2 x 2).  Tile ids of a sample are consecutive and xcd_remap() keeps consecutive tiles on one
//               XCD, so a cluster shares one L2.
//   phase 1   : the cluster's workgroups split the sample's Ct / 8 activation items (8 channels x N x N each); inside a
//               workgroup the 4-wave groups (3 at 32^2: 12 waves; 2 at 16^2: 8 waves) each run the plane kernel's item
//               loop (af.hip: transpose into per-channel planes with GroupNorm applied, four chained MFMA passes per wave,
//               staged 16-byte stores) in lock-step over the workgroup barrier - which is how co-resident workgroups of the
//               stand-alone kernel behave anyway.  The sample's group statistics are folded ONCE per workgroup.
//   hand-over : every wave waits for its stores (vmcnt(0)), one lane arrives on the sample's counter and polls it until
//               the CS workgroups are in.  XCD-local form (every cluster inside one XCD: tiles % (8 CS) == 0): plain
//               stores - the data is in the XCD's L2 when the store is acknowledged -, an L2 atomic, and the convolution's
//               patch LDS-DMA bypasses this CU's L1 (sc1).  General form (any placement): write-through stores, agent-scope
//               atomics, acquire fence.  The last workgroup to leave zeroes the counters (self-resetting: graph replays).
//   phase 2   : the halo-patch convolution tile, unchanged (conv3h_body.inc), reading the activated rows out of the L2.
//
// Results are bit-identical to afldm_af_act followed by afldm_conv2d (same arithmetic, same order); the activated tensor is
// still written (the caller may want it; it is 1/3 of the pair's bytes).  A workgroup spins only on workgroups of its own
// cluster, which have neighbouring ids (b, b + 8, b + 16, b + 24): with in-order dispatch a cluster is never partly
// resident for longer than it takes the dispatcher to reach its last member.
#include "af_plane.hpp"
#include "../../include/afldm_hip_experimental.h"      // (libafldm_exp.so: not part of the product library)
#include "conv3h_tile.hpp"

namespace afldm {

struct ClusterP {
  unsigned* sync;   // hand-over counters: one 128-byte line per sample (word 0 arrivals, word 1 departures), zero between launches
  unsigned* err;    // error word (sync[8193]; 1: a cluster timed out, 2: a workgroup is not on the XCD its id implies)
  int cs;           // workgroups per cluster (= tiles per sample)
  int flags;        // bit 0: general (agent-scope) hand-over instead of the XCD-local one
  unsigned long long* trace;   // diagnostic: [workgroup][8] s_memtime stamps (start, prologue done, items done, stores acknowledged,
                               // cluster complete, convolution K loop done is not visible from here: the host brackets the launch), or NULL
};
static unsigned long long* g_actconv_trace = nullptr;
static int g_actconv_general = 0;     // afldm_af_act_conv2d_mode(1): always the general (agent-scope) hand-over

__device__ __forceinline__ int hw_xcc_id() {
  int v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 7;
}


constexpr int kActConvMaxCh = 1024;     // channels of one workgroup's items (scale / shift table in LDS)
constexpr int kActConvMaxG = 64;

template <typename T, int N, int CH, int NG>
struct ActConvLds {
  typedef PlaneCfg<T, N, CH> CF;
  static constexpr int GROUP = CF::LDS_BYTES;
  static constexpr int TAB = NG * GROUP;                                  // [kActConvMaxG][2] mean / rstd, then scale / shift tables
  static constexpr int BYTES = TAB + kActConvMaxG * 8 + 2 * kActConvMaxCh * 4;
};

// PRE: the activation IN FRONT of the convolution (pa: raw input -> p.x1);  POST: the activation BEHIND it (pb: p.y -> pb.y,
// statistics = the partial sums the convolution's epilogue has just written).  The phases are lambdas so that the
// convolution tile's own `return`s (conv3h_body.inc) leave its phase, not the kernel.
template <typename T, int BM, int W_, int BN, int WGM, int WGN, int NPROD, int STAGES, int MINW, int MF, int TPS, int CH, bool PRE, bool POST>
__global__ void __launch_bounds__((WGM * WGN + NPROD) * 64, MINW) k_act_conv3h(ConvP p, AfP<T> pa, AfP<T> pb, ClusterP xx) {
  constexpr bool SUB = false;
  constexpr int N = W_;
  constexpr int NWALL = WGM * WGN + NPROD, NG = NWALL / 4, NTALL = NWALL * 64;
  static_assert(NWALL % 4 == 0, "4-wave groups");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const bool general = (xx.flags & 1) != 0;
  // the sample and this workgroup's rank in its cluster: conv3h_body.inc's default tile order (p.xcd_gn == 0)
  const int cl_tile = xcd_remap(blockIdx.x, gridDim.x);
  const int cl_b = cl_tile / xx.cs, cl_rank = cl_tile - cl_b * xx.cs;
  auto stamp = [&](int i) {
    if (xx.trace && threadIdx.x == 0) xx.trace[(size_t)blockIdx.x * 16 + i] = __builtin_amdgcn_s_memtime();
  };
  stamp(0);

  // ------------------------------------------------------------------------------------------- activation items of sample cl_b
  auto act_items = [&](const AfP<T>& q, const int st0) __attribute__((always_inline)) {
    typedef PlaneCfg<T, N, CH> CF;
    typedef ActConvLds<T, N, CH, NG> AL;
    typedef Mma<T> MM;
    typedef typename MM::Chunk Chunk;
    constexpr int EPC = CF::EPC, KPF = CF::KPF, KH = CF::KH;
    constexpr int KHP = CF::KHP, H2P = CF::H2P, YRP = CF::YRP;
    constexpr int NT = 256, TN = CF::TN, TH = CF::TH, NKF1 = CF::NKF1, NKF3 = CF::NKF3, CPW = CF::CPW;
    static_assert(CF::CREG && CF::NW == 4, "bf16 plane configuration");

    const int tid_all = threadIdx.x, lane = tid_all & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(tid_all >> 6);
    const int grp = wave_all >> 2, wave = wave_all & 3, tid = tid_all & 255;
    const int li = lane & 15, lg = lane >> 4;
    const int b = cl_b;

    char* gsm = smem + grp * AL::GROUP;
    T* Xs = reinterpret_cast<T*>(gsm);
    T* Cs = Xs + CF::XS;
    T* Vt = Cs;                                              // (bf16: the constants live in registers)
    T* Vw = Vt + wave * CF::VT;
    float* gtab = reinterpret_cast<float*>(smem + AL::TAB);  // [groups of this workgroup's channels][2]
    float* sctab = gtab + 2 * kActConvMaxG;                  // [channels of this workgroup's items]
    float* shtab = sctab + kActConvMaxCh;

    const int Ct = q.C1 + q.C2, ctiles = Ct / CH;
    const int ipw = (ctiles + xx.cs - 1) / xx.cs;
    const int first = cl_rank * ipw < ctiles ? cl_rank * ipw : ctiles;
    const int end = first + ipw < ctiles ? first + ipw : ctiles;
    const int rounds = (ipw + NG - 1) / NG;
    const int c_lo = first * CH, c_hi = end * CH;

    // ---- GroupNorm (mean, rstd) of the groups this workgroup's channels touch, ONCE: one wave per group, the terms and
    // the order of gn_group_sums_wave (what the stand-alone kernel adds per item): bit-identical statistics.  The small
    // loads (gamma / beta, partial sums of up to GR groups per wave) are requested FIRST - vmcnt retires in order, and
    // behind the first tile they would wait out its cold fetch - then the constants and the first tile.
    constexpr int SPL = 2, GR = (kActConvMaxG / 2 + NWALL - 1) / NWALL;     // partial-sum loads per lane and group; groups per wave (G <= 32 on the fast path)
    constexpr int CPT = (kActConvMaxCh + NTALL - 1) / NTALL;                // channels per thread of the scale / shift tables
    float gmr[CPT], btr[CPT];
    f32x2 sreg[GR][SPL];
    const bool norm = q.gs.st1 != nullptr && end > first;
    const int cpg = norm ? Ct / q.G : 1;
    const int g_lo = c_lo / cpg, g_hi = norm ? (c_hi - 1) / cpg : -1;
    const int smax = q.gs.S1 > q.gs.S2 ? q.gs.S1 : q.gs.S2;
    const bool fast = norm && cpg * smax <= 64 * SPL && g_hi - g_lo + 1 <= GR * NWALL;       // (workgroup-uniform)
    if (norm) {
#pragma unroll
      for (int k = 0; k < CPT; ++k) {
        const int c = c_lo + tid_all + k * NTALL;
        gmr[k] = c < c_hi ? q.gamma[c] : 0.f;
        btr[k] = c < c_hi ? q.beta[c] : 0.f;
      }
    }
    if (fast) {
#pragma unroll
      for (int qq = 0; qq < GR; ++qq) {
        const int g = g_lo + wave_all + qq * NWALL;
#pragma unroll
        for (int u = 0; u < SPL; ++u) {
          const int j = lane + 64 * u;
          sreg[qq][u] = f32x2{0.f, 0.f};
          if (g <= g_hi && j < cpg * smax) {
            const int c = g * cpg + j / smax, sp = j - (j / smax) * smax;
            const bool second = c >= q.gs.C1;
            const int S = second ? q.gs.S2 : q.gs.S1;
            if (sp < S) {
              const float* stp = second ? q.gs.st2 : q.gs.st1;
              const int Cs_ = second ? q.gs.C2 : q.gs.C1, cc = second ? c - q.gs.C1 : c;
              sreg[qq][u] = *reinterpret_cast<const f32x2*>(stp + (((size_t)b * S + sp) * Cs_ + cc) * 2);
            }
          }
        }
      }
    }

    stamp(12);
    Chunk creg[CF::NFRAG];
#pragma unroll
    for (int f = 0; f < CF::NFRAG; ++f) creg[f] = ld16<Chunk>(reinterpret_cast<const T*>(q.packed) + (f * 64 + lane) * EPC);
    auto cfrag = [&](int f) -> Chunk { return creg[f]; };

    // X tile staging units (as k_af_act_plane): unit u = (cq, w, hq) = PX pixels x EPC channels, transposed on the way in
    constexpr int CQ = CH / EPC;
    constexpr int PXW = N * N * CQ / NT, PX = PXW >= 4 ? EPC : (PXW >= 1 ? PXW : 1);
    constexpr int HQ = N / PX, UNITS = N * CQ * HQ, UPT = (UNITS + NT - 1) / NT;
    static_assert(N % PX == 0 && (PX & (PX - 1)) == 0, "staging unit");
    Chunk pre[UPT][PX];
    auto fetch = [&](int item) {
      const int c0 = item * CH;
      const bool second = c0 >= q.C1;
      const T* xsrc = second ? q.x2 : q.x1;
      const int Cs_ = second ? q.C2 : q.C1, cs0 = second ? c0 - q.C1 : c0;
#pragma unroll
      for (int k = 0; k < UPT; ++k) {
        const int u = tid + k * NT;
        if (u < UNITS) {
          const int cq = u % CQ, w = (u / CQ) % N, hq = u / (CQ * N);
#pragma unroll
          for (int e = 0; e < PX; ++e)
            pre[k][e] = ld16<Chunk>(xsrc + ((size_t)(b * N + hq * PX + e) * N + w) * Cs_ + cs0 + cq * EPC);
        }
      }
    };
    int item = first + grp;
    if (item < end) fetch(item);
    stamp(13);

    if (norm) {
      if (fast) {
#pragma unroll
        for (int qq = 0; qq < GR; ++qq) {
          const int g = g_lo + wave_all + qq * NWALL;
          if (g <= g_hi) {                                   // (wave-uniform)
            double s1 = 0.0, s2 = 0.0;
#pragma unroll
            for (int u = 0; u < SPL; ++u) {
              if (lane + 64 * u < cpg * smax) {              // (exactly the terms, in the order, of gn_group_sums_wave)
                s1 += (double)sreg[qq][u][0];
                s2 += (double)sreg[qq][u][1];
              }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
              s1 += __shfl_xor(s1, o, 64);
              s2 += __shfl_xor(s2, o, 64);
            }
            if (lane == 0) {
              float mean, rstd;
              gn_mean_rstd(s1, s2, (double)N * N * cpg, q.eps, mean, rstd);
              gtab[2 * (g - g_lo)] = mean;
              gtab[2 * (g - g_lo) + 1] = rstd;
            }
          }
        }
      } else {
        for (int g = g_lo + wave_all; g <= g_hi; g += NWALL) {
          double s1, s2;
          gn_group_sums_wave(q.gs, b, g, cpg, lane, s1, s2);
          if (lane == 0) {
            float mean, rstd;
            gn_mean_rstd(s1, s2, (double)N * N * cpg, q.eps, mean, rstd);
            gtab[2 * (g - g_lo)] = mean;
            gtab[2 * (g - g_lo) + 1] = rstd;
          }
        }
      }
      stamp(14);
      __syncthreads();
      stamp(15);
#pragma unroll
      for (int k = 0; k < CPT; ++k) {
        const int c = c_lo + tid_all + k * NTALL;
        if (c < c_hi) {
          const int gl = c / cpg - g_lo;
          const float sc = gtab[2 * gl + 1] * gmr[k];
          sctab[c - c_lo] = sc;
          shtab[c - c_lo] = btr[k] - gtab[2 * gl] * sc;
        }
      }
    } else {
      for (int c = c_lo + tid_all; c < c_hi; c += NTALL) {
        sctab[c - c_lo] = 1.f;
        shtab[c - c_lo] = 0.f;
      }
    }
    stamp(st0);

    // Every group runs every round (a group without an item in the last round computes on stale planes and stores
    // nothing): the workgroup barriers stay unconditional and the groups move in lock-step.
    for (int r = 0; r < rounds; ++r, item += NG) {
      const bool live = item < end;                          // (uniform inside the 4-wave group)
      const int c0 = (live ? item : first) * CH;
      __syncthreads();   // scale / shift tables ready (first round); the previous item's output has left the X region
      if constexpr (KH > N) {  // the output staging of the previous item overwrote the X region: re-zero its K padding
        for (int i = tid; i < N * CH * (KH - N); i += NT) {
          const int row = i / (KH - N), k = N + (i - row * (KH - N));
          Xs[row * KHP + k] = from_f32<T>(0.f);
        }
      }
      if (live) {
        const float* gsc = sctab + (c0 - c_lo);
        const float* gsh = shtab + (c0 - c_lo);
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
                typedef __attribute__((ext_vector_type(PX))) T Run;
                Run o;
#pragma unroll
                for (int e = 0; e < PX; ++e) o[e] = from_f32<T>(to_f32(pre[k][e][cc]) * sc + sh);
                *reinterpret_cast<Run*>(dst) = o;
              }
            }
          }
        }
      }
      __syncthreads();
      if (item + NG < end) fetch(item + NG);                 // in flight during the MFMA passes
#include "af_plane_passes.inc"
      __syncthreads();   // every wave has finished reading the X planes: the region becomes the output tile
      T* Ys = Xs;
#pragma unroll
      for (int pl = 0; pl < CPW; ++pl)
#pragma unroll
        for (int t4 = 0; t4 < TN; ++t4)
#pragma unroll
          for (int tw = 0; tw < TN; ++tw)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
              Ys[(16 * t4 + 4 * lg + rr) * YRP + (16 * tw + li) * CH + wave * CPW + pl] = from_f32<T>(yacc[pl][t4][tw][rr]);
      __syncthreads();
      if (live) {
        constexpr int CPP = CH / EPC;
        for (int i = tid; i < N * N * CPP; i += NT) {
          const int pix = i / CPP, qq = i - pix * CPP;
          const int h = pix / N, w = pix - h * N;
          st16<Chunk>(q.y + ((size_t)b * N * N + pix) * Ct + c0 + qq * EPC, ld16<Chunk>(Ys + h * YRP + w * CH + qq * EPC));
        }
      }
    }
    stamp(st0 + 1);
  };

  // ------------------------------------------------------------------------------------------- hand-over inside the cluster
  // Every thread's stores are acknowledged (XCD-local form: they are in the XCD's L2 then; general form: an agent-scope
  // release writes the L2 back), one lane arrives on the sample's counter line `line` and polls it until the cluster is
  // complete; the last workgroup to leave zeroes the line.  Readers behind it bypass their L1 (LDS-DMA: sc1) or have
  // not touched the tensor during this launch (plain loads of a line no CU has read since the kernel began).
  auto handover = [&](const int line, const int st0) __attribute__((always_inline)) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (general) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    stamp(st0);
    if (threadIdx.x == 0) {
      unsigned* cnt = xx.sync + ((size_t)line * gridDim.x / xx.cs + cl_b) * 32;
      if (general) {
        __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        // XCD-local: the atomics execute in this XCD's L2; word 2 collects the XCDs the cluster's members really run on
        __hip_atomic_fetch_or(cnt + 2, 1u << hw_xcc_id(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      unsigned spins = 0;
      while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)xx.cs) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1u << 24)) {                          // seconds: never hang the device on a lost cluster member
          __hip_atomic_store(xx.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
      }
      stamp(st0 + 1);
      if (general) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      unsigned old;
      if (general) old = __hip_atomic_fetch_add(cnt + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else old = __hip_atomic_fetch_add(cnt + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (old == (unsigned)xx.cs - 1) {                      // last one out: the counters return to zero for the next launch
        if (general) {
          __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(cnt + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
          // (a cluster split over XCDs would never get here - each L2 holds its own copy of the counter - and time out
          //  above; a mask with more than one bit can only mean the hardware id is not what we think it is)
          const unsigned xm = __hip_atomic_exchange(cnt + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          if (xm & (xm - 1)) __hip_atomic_store(xx.err, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_exchange(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          __hip_atomic_exchange(cnt + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      }
    }
    __syncthreads();
    stamp(st0 + 2);
  };

  // ------------------------------------------------------------------------------------------- the convolution tile
  auto conv_tile = [&]() __attribute__((always_inline)) {
#define H3_BX blockIdx.x
#define H3_NBX gridDim.x
#define H3_BZ 0
#define H3_PATCH_AUX 16
    // a producer wave leaves its K loop while the consumers still have the epilogue's barriers in front of them (bf16,
    // whole K, one sample per tile: 3, + 2 with statistics): with a phase behind the tile it must count them too
#define H3_PRODUCER_EXIT                                                  \
    if constexpr (POST) {                                                 \
      const int nb = 3 + (p.stats_out ? 2 : 0);                           \
      for (int i = 0; i < nb; ++i) __builtin_amdgcn_s_barrier();         \
    }                                                                     \
    return;
#include "conv3h_body.inc"
#undef H3_BX
#undef H3_NBX
#undef H3_BZ
#undef H3_PATCH_AUX
#undef H3_PRODUCER_EXIT
  };

  if constexpr (PRE) {
    act_items(pa, 1);
    handover(0, 3);
  }
  conv_tile();
  if constexpr (POST) {
    stamp(6);
    handover(1, 7);
    act_items(pb, 10);
  }
}

// ----------------------------------------------------------------------------------------------- host side
template <int BM, int W_, int BN, int WGM, int WGN, int MINW, int CH, bool PRE, bool POST>
static int launch_actconv(ConvP p, const AfP<bf16>& pa, const AfP<bf16>& pb, unsigned* sync, size_t sync_bytes, hipStream_t st) {
  typedef bf16 T;
  constexpr int NPROD = 4, STAGES = 3, MF = 16, TPS = 1;
  constexpr int NWALL = WGM * WGN + NPROD, NG = NWALL / 4;
  constexpr int ROWS = BM / W_, NPI = ((ROWS + 2) * (W_ + 2) + 7) / 8;
  constexpr int lds_conv = 2 * NPI * 1024 + STAGES * BN * 128;
  constexpr int lds_act = ActConvLds<T, W_, CH, NG>::BYTES;
  constexpr int lds = lds_conv > lds_act ? lds_conv : lds_act;
  static_assert(lds <= 160 * 1024, "LDS");
  p.tiles_n = p.Cout / BN;
  p.splitk = 1;
  p.xcd_gn = 0;
  const int tiles = (p.M / BM) * p.tiles_n;
  ClusterP xx;
  xx.cs = (W_ * W_ / BM) * p.tiles_n;
  xx.sync = sync + 16384;
  xx.err = sync + 8193;
  static const int s_mode = getenv("AFLDM_ACTCONV_MODE") ? atoi(getenv("AFLDM_ACTCONV_MODE")) : 0;      // 1: always the general hand-over
  xx.flags = (tiles % (8 * xx.cs) != 0 || s_mode == 1 || g_actconv_general) ? 1 : 0;
  xx.trace = g_actconv_trace;
  AFLDM_REQUIRE(sync_bytes >= (size_t)(16384 + 2 * 32 * p.B) * 4, AFLDM_ESHAPE, "afldm_af_act_conv2d: sync buffer too small for %d samples", p.B);
  auto kern = k_act_conv3h<T, BM, W_, BN, WGM, WGN, NPROD, STAGES, MINW, MF, TPS, CH, PRE, POST>;
  static unsigned long long attr_set = 0;
  if (first_on_device(attr_set)) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  kern<<<dim3(tiles, 1, 1), NWALL * 64, lds, st>>>(p, pa, pb, xx);
  return check_launch("afldm_af_act_conv2d(merged)");
}

// 0: no merged kernel for this chain; else the conv3h variant id the merged kernel embeds.  pre / post may be NULL (not both).
static int actconv_kernel_for(const afldm_af_act_args* pre, const afldm_conv_args* conv, const afldm_af_act_args* post, ConvP* p) {
  static const bool off = getenv("AFLDM_NO_ACTCONV") && atoi(getenv("AFLDM_NO_ACTCONV")) != 0;
  if (off || (!pre && !post) || !conv || conv->dtype != AFLDM_BF16 || !conv->sync) return 0;
  const int N = conv->H;
  if (conv->H != conv->W || (N != 32 && N != 16) || conv->x2 || conv->C2 || conv->Cout % 192) return 0;
  int vid = 0;
  if (!conv3h_plan(conv, p, &vid)) return 0;
  if (!((N == 32 && vid == kConv3hFirst + 0) || (N == 16 && vid == kConv3hFirst + 2))) return 0;
  const int bm = N == 32 ? 256 : 128;
  const int cs = (N * N / bm) * (conv->Cout / 192);
  if (conv->sync_bytes < (size_t)(16384 + 2 * 32 * conv->B) * 4) return 0;
  if (pre) {
    const int Ct = pre->C1 + pre->C2;
    if (conv->C1 != Ct || Ct % 8 || pre->C1 % 8 || !pre->packed || !pre->x1 || (pre->C2 && !pre->x2)) return 0;
    if (pre->stats1 && (pre->G <= 0 || pre->G > kActConvMaxG / 2 || Ct % pre->G || !pre->gamma || !pre->beta)) return 0;
    if (((Ct / 8 + cs - 1) / cs) * 8 > kActConvMaxCh) return 0;
  }
  if (post) {
    // the activation behind the tile normalises the convolution's OWN output with the partial sums its epilogue writes
    if (post->x1 != conv->y || post->x2 || post->C2 || post->C1 != conv->Cout || conv->y_ld != conv->Cout || conv->out_mode != 0 ||
        !post->packed || conv->Cout % 8)
      return 0;
    if (post->stats1 && (post->stats1 != conv->stats_out || post->S1 != p->stats_S || post->G <= 0 || post->G > kActConvMaxG / 2 ||
                         conv->Cout % post->G || !post->gamma || !post->beta))
      return 0;
    if (((conv->Cout / 8 + cs - 1) / cs) * 8 > kActConvMaxCh) return 0;
  }
  return vid;
}

static AfP<bf16> afp_of(const afldm_af_act_args* a, void* y, int B) {
  AfP<bf16> q;
  memset(&q, 0, sizeof(q));
  if (!a) return q;
  q.x1 = (const bf16*)a->x1;
  q.x2 = (const bf16*)a->x2;
  q.gs = GnStats{a->stats1, a->stats2, a->C1, a->C2, a->S1, a->S2};
  q.gamma = a->gamma;
  q.beta = a->beta;
  q.U = a->U;
  q.D = a->D;
  q.packed = a->packed;
  q.y = (bf16*)y;
  q.C1 = a->C1;
  q.C2 = a->C2;
  q.G = a->G;
  q.B = B;
  q.eps = a->eps;
  q.trace = nullptr;
  return q;
}

}  // namespace afldm

using namespace afldm;

extern "C" int afldm_af_act_conv2d_mode(int general) {
  g_actconv_general = general ? 1 : 0;
  return AFLDM_OK;
}

extern "C" int afldm_af_act_conv2d_trace(void* buf) {
  g_actconv_trace = (unsigned long long*)buf;
  return AFLDM_OK;
}

extern "C" int afldm_act_conv_act_merged(const afldm_af_act_args* pre, const afldm_conv_args* conv, const afldm_af_act_args* post) {
  ConvP p;
  return actconv_kernel_for(pre, conv, post, &p) != 0 ? 1 : 0;
}

extern "C" int afldm_act_conv_act(const afldm_af_act_args* pre, const afldm_conv_args* conv, const afldm_af_act_args* post,
                                  void* post_y, afldm_stream_t stream) {
  AFLDM_REQUIRE(conv && (pre || post), AFLDM_ENULL, "afldm_act_conv_act: NULL arguments");
  AFLDM_REQUIRE(!post || post_y, AFLDM_ENULL, "afldm_act_conv_act: the activation behind the convolution needs an output");
  AFLDM_REQUIRE(conv->x1 && !conv->x2 && conv->H == conv->W && (!pre || conv->C1 == pre->C1 + pre->C2), AFLDM_ESHAPE,
                "afldm_act_conv_act: conv->x1 must be the (square) activated tensor of the activation in front of it");
  AFLDM_REQUIRE(!post || (post->x1 == conv->y && !post->x2 && post->C1 == conv->Cout && post->C2 == 0), AFLDM_ESHAPE,
                "afldm_act_conv_act: the activation behind the convolution takes conv->y (%d channels)", conv->Cout);
  hipStream_t st = (hipStream_t)stream;
  ConvP p;
  const int vid = actconv_kernel_for(pre, conv, post, &p);
  if (!vid) {
    // no merged kernel for this shape / dtype: the launches it stands for
    int rc = AFLDM_OK;
    if (pre)
      rc = afldm_af_act(pre->x1, pre->C1, pre->x2, pre->C2, pre->stats1, pre->S1, pre->stats2, pre->S2, pre->gamma, pre->beta, pre->G,
                        pre->eps, pre->U, pre->D, pre->packed, const_cast<void*>(conv->x1), conv->B, conv->H, conv->dtype, stream);
    if (rc) return rc;
    rc = afldm_conv2d(conv, stream);
    if (rc || !post) return rc;
    return afldm_af_act(post->x1, post->C1, nullptr, 0, post->stats1, post->S1, nullptr, 0, post->gamma, post->beta, post->G,
                        post->eps, post->U, post->D, post->packed, post_y, conv->B, conv->H, conv->dtype, stream);
  }
  const AfP<bf16> pa = afp_of(pre, const_cast<void*>(conv->x1), conv->B), pb = afp_of(post, post_y, conv->B);
  const int sel = (pre ? 1 : 0) | (post ? 2 : 0);
  if (conv->H == 32) {
    if (sel == 1) return launch_actconv<256, 32, 192, 4, 2, 3, 8, true, false>(p, pa, pb, conv->sync, conv->sync_bytes, st);
    if (sel == 2) return launch_actconv<256, 32, 192, 4, 2, 3, 8, false, true>(p, pa, pb, conv->sync, conv->sync_bytes, st);
    return launch_actconv<256, 32, 192, 4, 2, 3, 8, true, true>(p, pa, pb, conv->sync, conv->sync_bytes, st);
  }
  if (sel == 1) return launch_actconv<128, 16, 192, 2, 2, 2, 8, true, false>(p, pa, pb, conv->sync, conv->sync_bytes, st);
  if (sel == 2) return launch_actconv<128, 16, 192, 2, 2, 2, 8, false, true>(p, pa, pb, conv->sync, conv->sync_bytes, st);
  return launch_actconv<128, 16, 192, 2, 2, 2, 8, true, true>(p, pa, pb, conv->sync, conv->sync_bytes, st);
}

extern "C" int afldm_af_act_conv2d_merged(const afldm_af_act_args* act, const afldm_conv_args* conv) {
  return afldm_act_conv_act_merged(act, conv, nullptr);
}

extern "C" int afldm_af_act_conv2d(const afldm_af_act_args* act, const afldm_conv_args* conv, afldm_stream_t stream) {
  AFLDM_REQUIRE(act && conv, AFLDM_ENULL, "afldm_af_act_conv2d: NULL arguments");
  return afldm_act_conv_act(act, conv, nullptr, nullptr, stream);
}
