// af.hip — the alias-free operators of AFLDM as dense separable circulant products.
//
// The reference applies its ideal filters in the FFT domain (rfft2 -> 0/0.5/1 mask -> irfft2,
// reference afldm/af_libs/ideal_lpf.py:69-158).  Those operators are EXACTLY  Y = U X U^T
// (x2 periodic-sinc upsample) and  Y = D Z D^T  (brick-wall low-pass + decimate) with dense
// matrices built from the reference's masks (afldm_filter_matrix); WarpedNonlinearity
// (reference af_blocks.py:19-28) is  D silu(U X U^T) D^T  per (sample, channel) plane.
//
// k_af_act_mfma (N = 16, 32): one workgroup = one sample x 16 channels, NHWC.  The MFMA column
// index j is the channel, so every pass is "constant matrix x data" (A = U or D rows from LDS,
// B = data).  Four passes:  P1 (h -> h', U)  -> LDS ->  P2 (w -> w', U), SiLU, P3 (w' -> w, D)
// chained IN REGISTERS (the accumulator layout of P2 is a legal B operand for P3 once D's
// columns are permuted to match, see Mma<T>)  -> LDS ->  P4 (h' -> h, D).  The 2N x 2N
// upsampled plane therefore never exists in memory: HBM traffic is one read + one write of the
// tensor (the reference: ~30x that).  GroupNorm-apply is fused into the load.
// The h' axis is processed in slabs of SL rows to bound LDS (P4 accumulates across slabs).
//
// k_af_act_small (N = 2, 4, 8): one thread per (sample, channel) plane held in registers;
// lanes = channels, so the matrix coefficients are wave-uniform scalar operands.  HBM-bound.
//
// k_axis_contract: generic one-axis product used by the 4+4 AliasFreeUp/Downsample2D sites.
#include "common.hpp"

namespace afldm {

template <typename T>
struct AfP {
  const T* x1;
  const T* x2;
  const float* stats;
  const float* gamma;
  const float* beta;
  const float* U;  // [2N][N]
  const float* D;  // [N][2N]
  T* y;
  int C1, C2, G, B;
};

template <typename T>
__device__ __forceinline__ typename Mma<T>::Chunk pack_chain(const f32x4& lo, const f32x4& hi);
template <>
__device__ __forceinline__ bf16x8 pack_chain<bf16>(const f32x4& lo, const f32x4& hi) {
  bf16x8 v;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    v[r] = (bf16)lo[r];
    v[4 + r] = (bf16)hi[r];
  }
  return v;
}

template <typename T, int N>
struct AfCfg {
  typedef Mma<T> MM;
  static constexpr int EPC = MM::EPC, KPF = MM::KPF;
  static constexpr int H2 = 2 * N;
  static constexpr int KH = ((N + KPF - 1) / KPF) * KPF;  // K extent for contractions over an N-long axis
  static constexpr int SL = (sizeof(T) == 2) ? 32 : 16;   // h' rows per slab
  static constexpr int NSLAB = H2 / SL;
  static constexpr int WPW = N / 4;                        // w columns per wave in P1 / P4
  static constexpr bool PERM = sizeof(T) == 2;             // P3's matrix needs chain-permuted columns
  // LDS carve (elements of T)
  static constexpr int XS = N * 16 * KH;
  static constexpr int US = H2 * KH;
  static constexpr int DS = N * H2;
  static constexpr int DPS = PERM ? N * H2 : 0;
  static constexpr int T1S = SL * 16 * KH;
  static constexpr int VS = N * 16 * SL;
  static constexpr int LDS_BYTES = (XS + US + DS + DPS + T1S + VS) * (int)sizeof(T) + 2 * 16 * (int)sizeof(float);
};

template <typename T, int N>
__global__ void __launch_bounds__(256) k_af_act_mfma(AfP<T> p) {
  typedef AfCfg<T, N> CF;
  typedef Mma<T> MM;
  typedef typename MM::Chunk Chunk;
  constexpr int EPC = CF::EPC, KPF = CF::KPF, H2 = CF::H2, KH = CF::KH, SL = CF::SL, WPW = CF::WPW;
  constexpr int NKF1 = KH / KPF;   // chunk pairs when contracting an N-long axis
  constexpr int NKF3 = H2 / KPF;   // ... a 2N-long axis
  constexpr int NKF4 = SL / KPF;   // ... one slab of h'

  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* Xs = reinterpret_cast<T*>(smem);
  T* Us = Xs + CF::XS;
  T* Ds = Us + CF::US;
  T* Dp = CF::PERM ? Ds + CF::DS : Ds;
  T* T1 = Ds + CF::DS + CF::DPS;
  T* Vs = T1 + CF::T1S;
  float* gsc = reinterpret_cast<float*>(Vs + CF::VS);
  float* gsh = gsc + 16;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int Ct = p.C1 + p.C2;
  const int ctiles = Ct / 16;
  const int item = xcd_remap(blockIdx.x, gridDim.x);
  const int b = item / ctiles;
  const int c0 = (item % ctiles) * 16;
  const bool second = c0 >= p.C1;
  const T* xsrc = second ? p.x2 : p.x1;
  const int Cs = second ? p.C2 : p.C1;
  const int cs0 = second ? c0 - p.C1 : c0;

  // ---- phase 0: constants into LDS
  for (int i = tid; i < H2 * KH; i += 256) {
    const int r = i / KH, k = i - r * KH;
    Us[i] = from_f32<T>(k < N ? p.U[r * N + k] : 0.f);
  }
  for (int i = tid; i < N * H2; i += 256) {
    const int r = i / H2, k = i - r * H2;
    Ds[i] = from_f32<T>(p.D[r * H2 + k]);
    if (CF::PERM) {
      // column 32f + 8g + e of Dp  <-  column 32f + (e < 4 ? 4g + e : 16 + 4g + e - 4) of D
      const int f = k >> 5, g = (k >> 3) & 3, e = k & 7;
      const int src = 32 * f + (e < 4 ? 4 * g + e : 16 + 4 * g + (e - 4));
      Dp[i] = from_f32<T>(p.D[r * H2 + src]);
    }
  }
  if (tid < 16) {
    float sc = 1.f, sh = 0.f;
    if (p.stats) {
      const int c = c0 + tid, g = c / (Ct / p.G);
      const float mean = p.stats[2 * (b * p.G + g)], rstd = p.stats[2 * (b * p.G + g) + 1];
      sc = rstd * p.gamma[c];
      sh = p.beta[c] - mean * sc;
    }
    gsc[tid] = sc;
    gsh[tid] = sh;
  }
  if (KH > N) {  // zero the K padding of the two B-operand arrays that contract an N-long axis
    for (int i = tid; i < N * 16 * (KH - N); i += 256) {
      const int row = i / (KH - N), k = N + (i - row * (KH - N));
      Xs[row * KH + k] = from_f32<T>(0.f);
    }
    for (int i = tid; i < SL * 16 * (KH - N); i += 256) {
      const int row = i / (KH - N), k = N + (i - row * (KH - N));
      T1[row * KH + k] = from_f32<T>(0.f);
    }
  }
  __syncthreads();

  // ---- phase 1: x tile -> Xs[w][c][h] (transposed so h is K-contiguous), GroupNorm applied
  {
    constexpr int CQ = 16 / EPC, HQ = N / EPC;
    for (int u = tid; u < N * CQ * HQ; u += 256) {
      const int cq = u % CQ;
      const int w = (u / CQ) % N;
      const int hq = u / (CQ * N);
      float v[EPC][EPC];  // [h offset][channel offset]
#pragma unroll
      for (int e = 0; e < EPC; ++e) {
        const int h = hq * EPC + e;
        Chunk ch = ld16<Chunk>(xsrc + ((size_t)(b * N + h) * N + w) * Cs + cs0 + cq * EPC);
#pragma unroll
        for (int cc = 0; cc < EPC; ++cc) v[e][cc] = to_f32(ch[cc]) * gsc[cq * EPC + cc] + gsh[cq * EPC + cc];
      }
#pragma unroll
      for (int cc = 0; cc < EPC; ++cc) {
        Chunk o;
#pragma unroll
        for (int e = 0; e < EPC; ++e) o[e] = from_f32<T>(v[e][cc]);
        st16<Chunk>(Xs + ((size_t)(w * 16 + cq * EPC + cc)) * KH + hq * EPC, o);
      }
    }
  }
  __syncthreads();

  f32x4 yacc[WPW][N / 16];
#pragma unroll
  for (int a = 0; a < WPW; ++a)
#pragma unroll
    for (int t = 0; t < N / 16; ++t) yacc[a][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int s = 0; s < CF::NSLAB; ++s) {
    // ---- P1: T1[h'][c][w] = sum_h U[h'][h] X[h][w][c]   (this wave: its WPW columns w)
#pragma unroll
    for (int ti = 0; ti < SL / 16; ++ti) {
      Chunk uf[NKF1];
#pragma unroll
      for (int kf = 0; kf < NKF1; ++kf) uf[kf] = ld16<Chunk>(Us + (s * SL + 16 * ti + li) * KH + kf * KPF + lg * EPC);
      f32x4 acc[WPW];
#pragma unroll
      for (int wi = 0; wi < WPW; ++wi) {
        acc[wi] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int w = wave * WPW + wi;
#pragma unroll
        for (int kf = 0; kf < NKF1; ++kf)
          MM::mma(acc[wi], uf[kf], ld16<Chunk>(Xs + (w * 16 + li) * KH + kf * KPF + lg * EPC));
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        T* dst = T1 + ((16 * ti + 4 * lg + r) * 16 + li) * KH + wave * WPW;
#pragma unroll
        for (int w4 = 0; w4 < WPW; w4 += 4) store4<T>(dst + w4, acc[w4][r], acc[w4 + 1][r], acc[w4 + 2][r], acc[w4 + 3][r]);
      }
    }
    __syncthreads();

    // ---- P2 -> SiLU -> P3 for this wave's SL/4 rows h' of the slab, chained in registers
    for (int hl = wave * (SL / 4); hl < (wave + 1) * (SL / 4); ++hl) {
      Chunk tf[NKF1];
#pragma unroll
      for (int kf = 0; kf < NKF1; ++kf) tf[kf] = ld16<Chunk>(T1 + (hl * 16 + li) * KH + kf * KPF + lg * EPC);
      f32x4 z[H2 / 16];
#pragma unroll
      for (int t2 = 0; t2 < H2 / 16; ++t2) {
        z[t2] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kf = 0; kf < NKF1; ++kf)
          MM::mma(z[t2], ld16<Chunk>(Us + (16 * t2 + li) * KH + kf * KPF + lg * EPC), tf[kf]);
#pragma unroll
        for (int r = 0; r < 4; ++r) z[t2][r] = silu_f(z[t2][r]);
      }
      Chunk pb[NKF3];
      if constexpr (CF::PERM) {
#pragma unroll
        for (int f = 0; f < NKF3; ++f) pb[f] = pack_chain<T>(z[2 * f], z[2 * f + 1]);
      } else {
#pragma unroll
        for (int f = 0; f < NKF3; ++f) pb[f] = z[f];
      }
#pragma unroll
      for (int t3 = 0; t3 < N / 16; ++t3) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int f = 0; f < NKF3; ++f) MM::mma(v, ld16<Chunk>(Dp + (16 * t3 + li) * H2 + f * KPF + lg * EPC), pb[f]);
#pragma unroll
        for (int r = 0; r < 4; ++r) Vs[((16 * t3 + 4 * lg + r) * 16 + li) * SL + hl] = from_f32<T>(v[r]);
      }
    }
    __syncthreads();

    // ---- P4: Y[h][w][c] += sum_{h' in slab} D[h][h'] V[h'][w][c]
#pragma unroll
    for (int wi = 0; wi < WPW; ++wi) {
      const int w = wave * WPW + wi;
#pragma unroll
      for (int kf = 0; kf < NKF4; ++kf) {
        Chunk vf = ld16<Chunk>(Vs + (w * 16 + li) * SL + kf * KPF + lg * EPC);
#pragma unroll
        for (int t4 = 0; t4 < N / 16; ++t4)
          MM::mma(yacc[wi][t4], ld16<Chunk>(Ds + (16 * t4 + li) * H2 + s * SL + kf * KPF + lg * EPC), vf);
      }
    }
    // no barrier needed here: the next slab's P1 only writes T1 (all waves passed the barrier
    // after P2/P3), and its P3 writes to Vs happen after the next barrier.
  }

  // ---- store: lane (c = li, g) holds rows h = 16 t4 + 4 g + r of column w
#pragma unroll
  for (int wi = 0; wi < WPW; ++wi) {
    const int w = wave * WPW + wi;
#pragma unroll
    for (int t4 = 0; t4 < N / 16; ++t4)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int h = 16 * t4 + 4 * lg + r;
        p.y[((size_t)(b * N + h) * N + w) * Ct + c0 + li] = from_f32<T>(yacc[wi][t4][r]);
      }
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
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int b = i / Ct, c = i - b * Ct;
    const bool second = c >= p.C1;
    const T* xs = second ? p.x2 : p.x1;
    const int Cs = second ? p.C2 : p.C1;
    const int cc = second ? c - p.C1 : c;
    float sc = 1.f, sh = 0.f;
    if (p.stats) {
      const int g = c / (Ct / p.G);
      const float mean = p.stats[2 * (b * p.G + g)], rstd = p.stats[2 * (b * p.G + g) + 1];
      sc = rstd * p.gamma[c];
      sh = p.beta[c] - mean * sc;
    }
    float X[N][N], Y[N][N];
#pragma unroll
    for (int h = 0; h < N; ++h)
#pragma unroll
      for (int w = 0; w < N; ++w) {
        X[h][w] = to_f32(xs[((size_t)(b * N + h) * N + w) * Cs + cc]) * sc + sh;
        Y[h][w] = 0.f;
      }
    // hp stays a run-time loop: X/Y are indexed statically (registers), the matrix rows with a
    // wave-uniform run-time offset (scalar loads), which keeps N = 8 inside the VGPR budget.
#pragma unroll 1
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
#pragma unroll
    for (int h = 0; h < N; ++h)
#pragma unroll
      for (int w = 0; w < N; ++w) p.y[((size_t)(b * N + h) * N + w) * Ct + c] = from_f32<T>(Y[h][w]);
  }
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

// ----------------------------------------------------------------------------- launchers
template <typename T, int N>
static int launch_af_mfma(const AfP<T>& p, hipStream_t st) {
  typedef AfCfg<T, N> CF;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)k_af_act_mfma<T, N>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              CF::LDS_BYTES);
    attr_set = true;
  }
  const int grid = p.B * ((p.C1 + p.C2) / 16);
  k_af_act_mfma<T, N><<<grid, 256, CF::LDS_BYTES, st>>>(p);
  return check_launch("afldm_af_act(mfma)");
}
template <typename T, int N>
static int launch_af_small(const AfP<T>& p, hipStream_t st) {
  const int total = p.B * (p.C1 + p.C2);
  const int grid = (total + 255) / 256;
  k_af_act_small<T, N><<<grid, 256, 0, st>>>(p);
  return check_launch("afldm_af_act(small)");
}

template <typename T>
static int af_act_dispatch(const void* x1, int C1, const void* x2, int C2, const float* stats, const float* gamma,
                           const float* beta, int G, const float* U, const float* D, void* y, int B, int N,
                           hipStream_t st) {
  AfP<T> p;
  p.x1 = (const T*)x1; p.x2 = (const T*)x2; p.stats = stats; p.gamma = gamma; p.beta = beta;
  p.U = U; p.D = D; p.y = (T*)y; p.C1 = C1; p.C2 = C2; p.G = G; p.B = B;
  switch (N) {
    case 2: return launch_af_small<T, 2>(p, st);
    case 4: return launch_af_small<T, 4>(p, st);
    case 8: return launch_af_small<T, 8>(p, st);
    case 16: return launch_af_mfma<T, 16>(p, st);
    case 32: return launch_af_mfma<T, 32>(p, st);
  }
  set_error("afldm_af_act: plane size N=%d not in {2,4,8,16,32}", N);
  return AFLDM_ESHAPE;
}

template <typename T>
static int resample_dispatch(const void* x, const float* M, void* y, float* ws, int B, int N, int C, int Rout,
                             hipStream_t st) {
  // pass 1 contracts H into the fp32 workspace [B][Rout][N][C]; pass 2 contracts W
  size_t n1 = (size_t)B * Rout * N * C, n2 = (size_t)B * Rout * Rout * C;
  int g1 = (int)((n1 + 255) / 256 < 8192 ? (n1 + 255) / 256 : 8192);
  int g2 = (int)((n2 + 255) / 256 < 8192 ? (n2 + 255) / 256 : 8192);
  k_axis_contract<T, float><<<g1, 256, 0, st>>>((const T*)x, ws, M, B, N, N, C, Rout, 0);
  k_axis_contract<float, T><<<g2, 256, 0, st>>>(ws, (T*)y, M, B, Rout, N, C, Rout, 1);
  return check_launch("afldm_af_resample");
}

}  // namespace afldm

using namespace afldm;

extern "C" int afldm_af_act(const void* x1, int C1, const void* x2, int C2, const float* stats, const float* gamma,
                            const float* beta, int G, const float* U, const float* D, void* y, int B, int N, int dtype,
                            afldm_stream_t stream) {
  AFLDM_REQUIRE(x1 && U && D && y, AFLDM_ENULL, "afldm_af_act: NULL pointer");
  AFLDM_REQUIRE(C1 > 0 && C2 >= 0 && (C2 == 0 || x2), AFLDM_ESHAPE, "afldm_af_act: bad C1=%d C2=%d", C1, C2);
  AFLDM_REQUIRE(B > 0, AFLDM_ESHAPE, "afldm_af_act: B=%d", B);
  AFLDM_REQUIRE(!stats || (gamma && beta && G > 0 && (C1 + C2) % G == 0), AFLDM_ESHAPE,
                "afldm_af_act: GroupNorm fusion needs gamma/beta and C %% G == 0 (C=%d G=%d)", C1 + C2, G);
  if (N >= 16) {
    AFLDM_REQUIRE(C1 % 16 == 0 && C2 % 16 == 0, AFLDM_ESHAPE, "afldm_af_act: C1=%d / C2=%d must be multiples of 16 for N=%d",
                  C1, C2, N);
    AFLDM_REQUIRE(aligned16(x1) && aligned16(x2) && aligned16(y), AFLDM_EALIGN, "afldm_af_act: pointers must be 16-byte aligned");
  }
  hipStream_t st = (hipStream_t)stream;
  if (dtype == AFLDM_F32) return af_act_dispatch<float>(x1, C1, x2, C2, stats, gamma, beta, G, U, D, y, B, N, st);
  if (dtype == AFLDM_BF16) return af_act_dispatch<bf16>(x1, C1, x2, C2, stats, gamma, beta, G, U, D, y, B, N, st);
  set_error("afldm_af_act: unknown dtype %d", dtype);
  return AFLDM_EDTYPE;
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

extern "C" int afldm_af_lpf_down2(const void* x, const float* D, void* y, float* workspace, int B, int N, int C,
                                  int dtype, afldm_stream_t stream) {
  AFLDM_REQUIRE(x && D && y && workspace, AFLDM_ENULL, "afldm_af_lpf_down2: NULL pointer");
  AFLDM_REQUIRE(B > 0 && N >= 2 && N % 2 == 0 && C > 0, AFLDM_ESHAPE, "afldm_af_lpf_down2: bad shape (N=%d must be even)", N);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == AFLDM_F32) return resample_dispatch<float>(x, D, y, workspace, B, N, C, N / 2, st);
  if (dtype == AFLDM_BF16) return resample_dispatch<bf16>(x, D, y, workspace, B, N, C, N / 2, st);
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
