// sep.hip — one-axis constant-matrix products for LARGE planes (the alias-free VAE: 64^2..256^2,
// reference afldm/models/af_vae.py + af_api.make_af_vae), where a whole 2N x 2N plane no longer
// fits in LDS and the alias-free operators become separable passes through HBM:
//
//   y[line][r] = act( sum_k M[r][k] * xn[line][k] )                    (R x K matrix M)
//   y[line][r2] = sum_r M2[r2][r] * silu( sum_k M[r][k] xn[line][k] )   (chained: up -> SiLU -> down
//                                                                       along one axis, in registers)
// A "line" is a strided vector of the tensor (stride = in_k_stride elements); 16 memory-adjacent
// lines (channels, or (w, c) pairs) form the MFMA column index j, so the matrix is always the A
// operand (rows from LDS) and the data the B operand after a 16 x K transposing stage through LDS.
// GroupNorm-apply (per (sample, channel) scale/shift table) can be fused into the load.
//
// Composition (host side, afldm_amd/ops.py):
//   AF activation N >= 64 : [GN +] up-H  ->  up-W/SiLU/down-W chained  ->  down-H      (3 launches)
//   UpsampleRFFT(2) / LPF+decimate at N >= 32 : pass over H, pass over W                (2 launches)
#include "common.hpp"

namespace afldm {

struct SepP {
  const void* x;
  void* y;
  const float* M;   // [R][K]
  const float* M2;  // [R2][R] or NULL
  const float* gn_table;  // [B][C][2] (scale, shift) or NULL
  long long outer_count, inner_count;
  long long in_outer_stride, in_k_stride, out_outer_stride, out_k_stride;
  int C, outer_per_sample, act;
};

// UID ("up identity", chained passes only): the caller guarantees M[2i][k] = delta(i, k) - the x2 periodic-sinc
// upsampler of the reference, U[::2, :] = I (ideal_lpf.py:96-121: the even phase of the
// zero-stuffed, recon-filtered signal is the signal itself).  Then  M2 silu(M x) = M2[:, 0::2] silu(x) + M2[:, 1::2]
// silu(M[1::2] x):  the even rows need no product, the first matrix shrinks to its K odd rows (69 -> 35 KB at K = 128),
// all three matrices are K wide and EIGHT waves fit next to them (104 + 43 KB) where four were alone with 137 KB.
// CT: 16-line column tiles per wave (plain passes only).  A wave owns 16 CT memory-adjacent lines: with CT = 4 (bf16) every
// row of a group is ONE 128-byte piece for loads and stores - at CT = 1 the 32-byte pieces of the batch-128 H passes (1 + 2 GB,
// beyond the MALL) reached 0.9 TB/s of HBM - and a matrix fragment read from LDS feeds CT MFMAs instead of one.
template <typename T, int K, int R, int R2, int NWV = 4, bool UID = false, int CT = 1>
struct SepCfg {
  typedef Mma<T> MM;
  static constexpr int EPC = MM::EPC, KPF = MM::KPF;
  static constexpr int KP = ((K + KPF - 1) / KPF) * KPF;   // K extent of the first product
  static constexpr int KPS = KP + EPC;                     // padded LDS row strides (bank conflicts)
  static constexpr int RP = ((R + KPF - 1) / KPF) * KPF;   // K extent of the chained product
  static constexpr int RPS = RP + EPC;
  static constexpr int RT = (R + 15) / 16, R2T = (R2 + 15) / 16;
  static constexpr int M_ELEMS = UID ? K * KPS : RT * 16 * KPS;              // UID: the K odd rows of M
  static constexpr int M2_ELEMS = R2 > 0 ? (UID ? 2 * R2T * 16 * KPS : R2T * 16 * RPS) : 0;   // UID: even | odd columns of M2
  static_assert(!UID || (R2 > 0 && R == 2 * K && K % KPF == 0), "identity form: chained x2 passes with whole K steps");
  static constexpr int LPW = 16 * CT;                      // lines per wave (group)
  static constexpr int TILE = LPW * KPS;
  static constexpr int OST = 32 * LPW;                     // per wave: output staging, 32 rows x LPW lines
  static_assert(CT == 1 || R2 == 0, "several column tiles per wave: plain passes only");
  static_assert(LPW <= 64, "one GroupNorm table entry per lane");
  static constexpr int LDS_BYTES = (M_ELEMS + M2_ELEMS + NWV * TILE + NWV * OST) * (int)sizeof(T);
  static constexpr bool PERM = sizeof(T) == 2;
};

// NWV waves per workgroup share one LDS image of the matrices: the large-plane configurations (matrices of 67 - 137 KB)
// hold one workgroup per CU, and with four waves that is ONE wave per SIMD whose load -> MFMA -> SiLU -> store chain
// nothing overlaps (the passes ran at 1 - 1.5 TB/s); eight waves where the tiles still fit give every SIMD a second
// wave to issue from.
template <typename T, int K, int R, int R2, int NWV, bool UID = false, int CT = 1>
__global__ void __launch_bounds__(NWV * 64) k_sep(SepP p) {
  typedef SepCfg<T, K, R, R2, NWV, UID, CT> CF;
  constexpr int LPW = CF::LPW;
  constexpr int NT = NWV * 64;
  typedef Mma<T> MM;
  typedef typename MM::Chunk Chunk;
  constexpr int EPC = CF::EPC, KPF = CF::KPF, KP = CF::KP, KPS = CF::KPS, RP = CF::RP, RPS = CF::RPS;
  constexpr int RT = CF::RT, R2T = CF::R2T, NKF1 = KP / KPF, NKF2 = R2 > 0 ? RP / KPF : 1;
  static_assert(R2 == 0 || (RT % 2 == 0 || !CF::PERM), "chained bf16 product packs row tiles in pairs");
  static_assert(!UID || ((K / 16) % 2 == 0 || !CF::PERM), "identity form: odd-row tiles pack in pairs");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* Ms = reinterpret_cast<T*>(smem);
  T* M2s = Ms + CF::M_ELEMS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  T* tile = M2s + CF::M2_ELEMS + wave * CF::TILE;
  T* sO = M2s + CF::M2_ELEMS + NWV * CF::TILE + wave * CF::OST;
  const int li = lane & 15, lg = lane >> 4;
  // Output rows leave through a wave-private LDS tile: an MFMA result holds ONE line per lane (2 / 4 bytes per row),
  // and storing it as such was 4 x 16 two-byte scattered stores per 16-row tile - the passes ran 5x off their HBM
  // time on store issue (profiles/r03/r03a_vae_kernel_stats.csv: 2.1 ms for a 2 x 1.07 GB pass).  Two tiles (32 rows
  // x 16 lines) are staged and leave as 16-byte pieces, a row's 16 lines (32 / 64 bytes) contiguous.
  constexpr int LPR = LPW / EPC;                           // lanes per row of LPW lines
  constexpr int RPI = 64 / LPR;                            // rows per store instruction
  const int orow = lane / LPR, ocol = (lane % LPR) * EPC;

  // ---- matrices -> LDS (zero padded; M2 columns chain-permuted for bf16), once per workgroup
  if constexpr (UID) {
    for (int i = tid; i < CF::M_ELEMS; i += NT) {            // odd rows of M
      const int r = i / KPS, k = i - r * KPS;
      Ms[i] = from_f32<T>(k < K ? p.M[(size_t)(2 * r + 1) * K + k] : 0.f);
    }
    constexpr int HALF = R2T * 16 * KPS;
    for (int i = tid; i < 2 * HALF; i += NT) {               // [even columns, standard order | odd columns, chain order]
      const int odd = i >= HALF, j = i - odd * HALF;
      const int r = j / KPS;
      int k = j - r * KPS;
      float v = 0.f;
      if (k < K) {
        if (odd && CF::PERM) {
          const int f = k >> 5, g = (k >> 3) & 3, e = k & 7;
          k = 32 * f + (e < 4 ? 4 * g + e : 16 + 4 * g + (e - 4));
        }
        if (r < R2) v = p.M2[(size_t)r * R + 2 * k + odd];
      }
      M2s[i] = from_f32<T>(v);
    }
  } else {
  for (int i = tid; i < CF::M_ELEMS; i += NT) {
    const int r = i / KPS, k = i - r * KPS;
    Ms[i] = from_f32<T>((r < R && k < K) ? p.M[(size_t)r * K + k] : 0.f);
  }
  }
  if constexpr (R2 > 0 && !UID) {
    for (int i = tid; i < CF::M2_ELEMS; i += NT) {
      const int r = i / RPS;
      int k = i - r * RPS;
      float v = 0.f;
      if (k < RP) {
        if (CF::PERM) {  // column 32f + 8g + e  <-  column 32f + (e < 4 ? 4g + e : 16 + 4g + e - 4)
          const int f = k >> 5, g = (k >> 3) & 3, e = k & 7;
          k = 32 * f + (e < 4 ? 4 * g + e : 16 + 4 * g + (e - 4));
        }
        if (r < R2 && k < R) v = p.M2[(size_t)r * R + k];
      }
      M2s[i] = from_f32<T>(v);
    }
  }
  if constexpr (KP > K) {  // K padding of this wave's tile (never overwritten)
    for (int i = lane; i < LPW * (KP - K); i += 64) {
      const int row = i / (KP - K), k = K + (i - row * (KP - K));
      tile[row * KPS + k] = from_f32<T>(0.f);
    }
  }
  __syncthreads();

  const long long groups_per_outer = p.inner_count / LPW;
  const long long ngroups = p.outer_count * groups_per_outer;
  const T* x = (const T*)p.x;
  T* y = (T*)p.y;
  // Software pipeline over this wave's groups: the NEXT group's lines (and its GroupNorm table entries) are fetched into
  // registers while the current group runs its MFMAs and stores.  The staging tile and the output tile are wave-private
  // and a wave's LDS operations execute in order, so the loop needs no workgroup barrier: with the matrices filling the
  // LDS there is ONE wave per SIMD, and the serial load -> barrier -> compute -> barrier chain left every pass at ~1 TB/s
  // (HBM latency per group un-hidden; profiles/r03/r03b_vae_kernel_stats.csv).
  constexpr int CQ = LPW / EPC, KQ = K / EPC, UNITS = CQ * KQ;
  constexpr int UPL = (UNITS + 63) / 64;                   // staging units per lane
  static_assert(K % EPC == 0 && 64 % CQ == 0, "staging units");
  const int ucq = lane % CQ, ukq = lane / CQ;              // unit j of this lane: (ucq, ukq + j * 64 / CQ)
  Chunk pre[UPL][EPC];
  float psc = 1.f, psh = 0.f;
  auto fetch = [&](long long grp) {
    if (grp >= ngroups) return;
    const long long outer = grp / groups_per_outer;
    const long long inner0 = (grp - outer * groups_per_outer) * LPW;
    const T* src = x + outer * p.in_outer_stride + inner0;
#pragma unroll
    for (int j = 0; j < UPL; ++j) {
      const int kq = ukq + j * (64 / CQ);
      if (kq < KQ) {
#pragma unroll
        for (int e = 0; e < EPC; ++e) pre[j][e] = ld16<Chunk>(src + (long long)(kq * EPC + e) * p.in_k_stride + ucq * EPC);
      }
    }
    psc = 1.f;
    psh = 0.f;
    if (p.gn_table) {    // per-line GroupNorm scale / shift: lane l holds line inner0 + l (channel = line % C)
      const int c = (int)((inner0 + (lane < LPW ? lane : 0)) % p.C);
      const long long b = outer / p.outer_per_sample;
      const f32x2 t = *reinterpret_cast<const f32x2*>(p.gn_table + ((size_t)b * p.C + c) * 2);
      psc = t[0];
      psh = t[1];
    }
  };
  const long long gstep = (long long)gridDim.x * NWV;
  fetch((long long)blockIdx.x * NWV + wave);
  for (long long g0 = (long long)blockIdx.x * NWV; g0 < ngroups; g0 += gstep) {
    const long long grp = g0 + wave;
    const bool live = grp < ngroups;
    const long long outer = live ? grp / groups_per_outer : 0;
    const long long inner0 = live ? (grp - outer * groups_per_outer) * LPW : 0;
    if (live) {
      float usc[EPC], ush[EPC];
#pragma unroll
      for (int cc = 0; cc < EPC; ++cc) {
        usc[cc] = __shfl(psc, ucq * EPC + cc, 64);
        ush[cc] = __shfl(psh, ucq * EPC + cc, 64);
      }
#pragma unroll
      for (int j = 0; j < UPL; ++j) {
        const int kq = ukq + j * (64 / CQ);
        if (kq < KQ) {
#pragma unroll
          for (int cc = 0; cc < EPC; ++cc) {
            Chunk o;
#pragma unroll
            for (int e = 0; e < EPC; ++e) o[e] = from_f32<T>(to_f32(pre[j][e][cc]) * usc[cc] + ush[cc]);
            st16<Chunk>(tile + (ucq * EPC + cc) * KPS + kq * EPC, o);
          }
        }
      }
    }
    fetch(grp + gstep);          // in flight during this group's products and stores

    // the tile is exchanged ACROSS LANES of this wave through LDS with no workgroup barrier: pin the order of the stores
    // above and the fragment reads below for the compiler (costs no instruction)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    Chunk xfa[CT][NKF1];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int kf = 0; kf < NKF1; ++kf) xfa[ct][kf] = ld16<Chunk>(tile + (16 * ct + li) * KPS + kf * KPF + lg * EPC);
    Chunk (&xf)[NKF1] = xfa[0];
    T* dst = y + outer * p.out_outer_stride + inner0;
    // tile t of the result -> staging slot t & 1; every second tile (and the last) the staged rows are stored
    auto putc = [&](int t, int ct, const f32x4& v) {
#pragma unroll
      for (int r = 0; r < 4; ++r) sO[((t & 1) * 16 + 4 * lg + r) * LPW + 16 * ct + li] = from_f32<T>(v[r]);
    };
    auto put = [&](int t, const f32x4& v) { putc(t, 0, v); };
    auto flush = [&](int t, int nt, int rows_total) {      // after tile t of nt
      if (!((t & 1) || t == nt - 1)) return;
      const int row0 = 16 * (t & ~1), nrows = 16 * ((t & 1) + 1);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // put() -> other lanes' reads of the staged rows
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int i = 0; i < 32 / RPI; ++i) {
        const int rr = i * RPI + orow;
        if (live && rr < nrows && row0 + rr < rows_total)
          st16<Chunk>(dst + (long long)(row0 + rr) * p.out_k_stride + ocol, ld16<Chunk>(sO + rr * LPW + ocol));
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // the staged rows are read before the next put() overwrites them
      __builtin_amdgcn_wave_barrier();
    };
    if constexpr (R2 == 0) {
      for (int t = 0; t < RT; ++t) {
        f32x4 z[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) z[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kf = 0; kf < NKF1; ++kf) {
          const Chunk a = ld16<Chunk>(Ms + (16 * t + li) * KPS + kf * KPF + lg * EPC);      // one matrix fragment, CT products
#pragma unroll
          for (int ct = 0; ct < CT; ++ct) MM::mma(z[ct], a, xfa[ct][kf]);
        }
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          if (p.act) {
#pragma unroll
            for (int r = 0; r < 4; ++r) z[ct][r] = silu_f(z[ct][r]);
          }
          putc(t, ct, z[ct]);
        }
        flush(t, RT, R);
      }
    } else if constexpr (UID) {
      constexpr int KT = K / 16, NKFO = K / KPF, HALF = R2T * 16 * KPS;
      f32x4 z[KT];                                           // odd rows of the up-sampled line, SiLU applied
#pragma unroll
      for (int t = 0; t < KT; ++t) {
        z[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kf = 0; kf < NKF1; ++kf) MM::mma(z[t], ld16<Chunk>(Ms + (16 * t + li) * KPS + kf * KPF + lg * EPC), xf[kf]);
#pragma unroll
        for (int r = 0; r < 4; ++r) z[t][r] = silu_f(z[t][r]);
      }
      Chunk pbo[NKFO];
      if constexpr (CF::PERM) {
#pragma unroll
        for (int f = 0; f < NKFO; ++f) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            pbo[f][r] = (bf16)z[2 * f][r];
            pbo[f][4 + r] = (bf16)z[2 * f + 1][r];
          }
        }
      } else {
#pragma unroll
        for (int f = 0; f < NKFO; ++f) pbo[f] = z[f];
      }
      Chunk xs[NKF1];                                        // even rows = the line itself: SiLU in place, standard K order
#pragma unroll
      for (int kf = 0; kf < NKF1; ++kf) {
#pragma unroll
        for (int e = 0; e < EPC; ++e) xs[kf][e] = from_f32<T>(silu_f(to_f32(xf[kf][e])));
      }
      for (int t = 0; t < R2T; ++t) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kf = 0; kf < NKF1; ++kf) MM::mma(v, ld16<Chunk>(M2s + (16 * t + li) * KPS + kf * KPF + lg * EPC), xs[kf]);
#pragma unroll
        for (int f = 0; f < NKFO; ++f) MM::mma(v, ld16<Chunk>(M2s + HALF + (16 * t + li) * KPS + f * KPF + lg * EPC), pbo[f]);
        put(t, v);
        flush(t, R2T, R2);
      }
    } else {
      f32x4 z[RT];
#pragma unroll
      for (int t = 0; t < RT; ++t) {
        z[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kf = 0; kf < NKF1; ++kf) MM::mma(z[t], ld16<Chunk>(Ms + (16 * t + li) * KPS + kf * KPF + lg * EPC), xf[kf]);
#pragma unroll
        for (int r = 0; r < 4; ++r) z[t][r] = silu_f(z[t][r]);
      }
      Chunk pb[NKF2];
      if constexpr (CF::PERM) {
#pragma unroll
        for (int f = 0; f < NKF2; ++f) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            pb[f][r] = (bf16)z[2 * f][r];
            pb[f][4 + r] = (bf16)z[2 * f + 1][r];
          }
        }
      } else {
#pragma unroll
        for (int f = 0; f < NKF2; ++f) pb[f] = z[f];
      }
      for (int t = 0; t < R2T; ++t) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int f = 0; f < NKF2; ++f) MM::mma(v, ld16<Chunk>(M2s + (16 * t + li) * RPS + f * KPF + lg * EPC), pb[f]);
        put(t, v);
        flush(t, R2T, R2);
      }
    }
  }
}

// (scale, shift) per (sample, channel) from GroupNorm partial sums: table[b][c] = (rstd*gamma, beta - mean*rstd*gamma)
__global__ void __launch_bounds__(256) k_gn_table(GnStats gs, const float* __restrict__ gamma,
                                                  const float* __restrict__ beta, float* __restrict__ table, int B, int C,
                                                  int G, double n, float eps) {
  // one wave per (sample, group): the cpg x S partials are summed across the lanes (a thread per channel
  // walked them serially: 100 us per call at S = 512)
  const int lane = threadIdx.x & 63;
  const int wg = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (wg >= B * G) return;
  const int b = wg / G, g = wg - b * G, cpg = C / G;
  double s1, s2;
  gn_group_sums_wave(gs, b, g, cpg, lane, s1, s2);
  float mean, rstd;
  gn_mean_rstd(s1, s2, n, eps, mean, rstd);
  for (int j = lane; j < cpg; j += 64) {
    const int c = g * cpg + j;
    const float k = rstd * gamma[c];
    const size_t i = (size_t)b * C + c;
    table[2 * i + 0] = k;
    table[2 * i + 1] = beta[c] - mean * k;
  }
}

// row softmax: y[r][:] = softmax(x[r][:] * scale), one workgroup per row (cols <= 16384)
template <typename T>
__global__ void __launch_bounds__(256) k_softmax_rows(const T* __restrict__ x, T* __restrict__ y, int cols, float scale) {
  __shared__ float red[8];
  const T* xr = x + (size_t)blockIdx.x * cols;
  T* yr = y + (size_t)blockIdx.x * cols;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float m = -1e30f;
  for (int i = threadIdx.x; i < cols; i += 256) m = fmaxf(m, to_f32(xr[i]) * scale);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if (lane == 0) red[wave] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float s = 0.f;
  for (int i = threadIdx.x; i < cols; i += 256) s += __expf(to_f32(xr[i]) * scale - m);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane == 0) red[wave] = s;
  __syncthreads();
  const float inv = 1.0f / (red[0] + red[1] + red[2] + red[3]);
  for (int i = threadIdx.x; i < cols; i += 256) yr[i] = from_f32<T>(__expf(to_f32(xr[i]) * scale - m) * inv);
}

template <typename T, int K, int R, int R2, int NWV, bool UID = false, int CT = 1>
static int launch_sep_w(const SepP& p, int per_cu, hipStream_t st) {
  typedef SepCfg<T, K, R, R2, NWV, UID, CT> CF;
  static_assert(CF::LDS_BYTES <= 160 * 1024, "LDS budget");
  static unsigned long long attr_set = 0;
  if (first_on_device(attr_set)) {
    (void)hipFuncSetAttribute((const void*)k_sep<T, K, R, R2, NWV, UID, CT>, hipFuncAttributeMaxDynamicSharedMemorySize, CF::LDS_BYTES);
  }
  const long long ngroups = p.outer_count * (p.inner_count / CF::LPW);
  long long grid = 256 * per_cu;
  if (grid * NWV > ngroups) grid = (ngroups + NWV - 1) / NWV;
  k_sep<T, K, R, R2, NWV, UID, CT><<<(int)grid, NWV * 64, CF::LDS_BYTES, st>>>(p);
  return check_launch("afldm_sep_pass");
}

// plain pass with CT column tiles per wave (inner_count a multiple of 16 CT); one or two 4-wave workgroups per CU
template <typename T, int K, int R, int CT>
static int launch_sep_ct(const SepP& p, hipStream_t st) {
  typedef SepCfg<T, K, R, 0, 4, false, CT> CF;
  constexpr int n4 = (160 * 1024) / CF::LDS_BYTES;
  return launch_sep_w<T, K, R, 0, 4, false, CT>(p, n4 >= 2 ? 2 : 1, st);
}

template <typename T, int K, int R, int R2, bool UID = false>
static int launch_sep(const SepP& p, hipStream_t st) {
  typedef SepCfg<T, K, R, R2, 4, UID> C4;
  typedef SepCfg<T, K, R, R2, 8, UID> C8;
  static_assert(C4::LDS_BYTES <= 160 * 1024, "LDS budget");
  constexpr int LDS = 160 * 1024;
  // waves per CU: as many 4-wave workgroups as fit (up to 4), or - when only one fits - one 8-wave workgroup if its
  // eight tiles still fit next to the matrices
  // Measured (profiles/r03/vae_sep_ab.txt): two 4-wave workgroups per CU is the sweet spot of the plain passes - three or
  // four per CU and an 8-wave workgroup for the 67 - 69 KB matrices all measured 5 - 40 % SLOWER (the passes sit at ~2.5 TB/s
  // of 32-byte row pieces; more waves only add contention); the 8-wave workgroup pays for the identity-form chained pass,
  // which is compute-bound (MFMA + SiLU per wave in series): 2.03 -> 1.25 ms at K = 128.
  constexpr int n4 = LDS / C4::LDS_BYTES;
  if constexpr (UID && n4 < 2 && C8::LDS_BYTES <= LDS) return launch_sep_w<T, K, R, R2, 8, UID>(p, 1, st);
  else return launch_sep_w<T, K, R, R2, 4, UID>(p, n4 >= 2 ? 2 : 1, st);
}

template <typename T>
static int sep_dispatch(const SepP& p, int K, int R, int R2, int up_identity, hipStream_t st) {
  if (up_identity) {      // (see SepCfg: chained x2 passes whose first matrix has identity even rows)
    // K = 128 is where it pays (the matrices drop under the 8-wave budget); at K = 64 the full product measured faster
    // (35.0 vs 41.7 ms over the C4 workload: the SiLU of the line itself costs more VALU than the MFMAs it saves) - the
    // small forms stay selectable for the tests (up_identity = 2)
    if (up_identity == 2 && K == 32 && R == 64 && R2 == 32) return launch_sep<T, 32, 64, 32, true>(p, st);
    if (up_identity == 2 && K == 64 && R == 128 && R2 == 64) return launch_sep<T, 64, 128, 64, true>(p, st);
    if constexpr (sizeof(T) == 2) {
      if (K == 128 && R == 256 && R2 == 128) return launch_sep<T, 128, 256, 128, true>(p, st);
    }
  }
  // plain passes over lines that come in memory-adjacent runs of 32 / 64: wider groups (see SepCfg, CT)
  static const bool no_ct = getenv("AFLDM_SEP_NO_CT") != nullptr;
  if (R2 == 0 && !no_ct) {
    if constexpr (sizeof(T) == 2) {
      if (K == 128 && R == 256 && p.inner_count % 64 == 0) return launch_sep_ct<T, 128, 256, 4>(p, st);
      if (K == 256 && R == 128 && p.inner_count % 32 == 0) return launch_sep_ct<T, 256, 128, 2>(p, st);
      if (K == 64 && R == 128 && p.inner_count % 64 == 0) return launch_sep_ct<T, 64, 128, 4>(p, st);
      if (K == 128 && R == 64 && p.inner_count % 32 == 0) return launch_sep_ct<T, 128, 64, 2>(p, st);
      if (K == 32 && R == 64 && p.inner_count % 64 == 0) return launch_sep_ct<T, 32, 64, 4>(p, st);
      if (K == 64 && R == 32 && p.inner_count % 64 == 0) return launch_sep_ct<T, 64, 32, 4>(p, st);
    }
  }
#define AFLDM_SEP(K_, R_, R2_) \
  if (K == K_ && R == R_ && R2 == R2_) return launch_sep<T, K_, R_, R2_>(p, st);
  // x2 upsampling passes (K = N, R = 2N), decimating passes (K = N, R = N/2) and the 2N -> N
  // passes of the large-plane activation, for the plane sizes of the AF-VAE (and small ones for tests)
  AFLDM_SEP(16, 32, 0) AFLDM_SEP(32, 64, 0) AFLDM_SEP(64, 128, 0)
  AFLDM_SEP(32, 16, 0) AFLDM_SEP(64, 32, 0) AFLDM_SEP(128, 64, 0)
  AFLDM_SEP(16, 32, 16) AFLDM_SEP(32, 64, 32) AFLDM_SEP(64, 128, 64)
  if constexpr (sizeof(T) == 2) {   // the 128^2 / 256^2 planes fit LDS in bf16 only
    AFLDM_SEP(128, 256, 0) AFLDM_SEP(256, 128, 0) AFLDM_SEP(128, 256, 128)
  }
#undef AFLDM_SEP
  set_error("afldm_sep_pass: no kernel for K=%d R=%d R2=%d in this dtype", K, R, R2);
  return AFLDM_ESHAPE;
}

}  // namespace afldm

using namespace afldm;

extern "C" int afldm_sep_pass(const afldm_sep_args* a, afldm_stream_t stream) {
  AFLDM_REQUIRE(a && a->x && a->y && a->M, AFLDM_ENULL, "afldm_sep_pass: NULL pointer");
  AFLDM_REQUIRE(a->outer_count > 0 && a->inner_count > 0 && a->inner_count % 16 == 0, AFLDM_ESHAPE,
                "afldm_sep_pass: inner_count=%lld must be a positive multiple of 16", (long long)a->inner_count);
  AFLDM_REQUIRE(a->in_k_stride % 8 == 0 && a->in_outer_stride % 8 == 0 && aligned16(a->x), AFLDM_EALIGN,
                "afldm_sep_pass: input strides must keep 16-byte chunks aligned");
  AFLDM_REQUIRE(a->out_k_stride % 8 == 0 && a->out_outer_stride % 8 == 0 && aligned16(a->y), AFLDM_EALIGN,
                "afldm_sep_pass: output strides must keep 16-byte chunks aligned");
  AFLDM_REQUIRE(!a->gn_table || (a->C > 0 && a->outer_per_sample > 0), AFLDM_ESHAPE, "afldm_sep_pass: GN table needs C and outer_per_sample");
  AFLDM_REQUIRE(a->R2 == 0 || a->M2, AFLDM_ENULL, "afldm_sep_pass: R2 > 0 needs M2");
  SepP p;
  p.x = a->x; p.y = a->y; p.M = a->M; p.M2 = a->R2 > 0 ? a->M2 : nullptr; p.gn_table = a->gn_table;
  p.outer_count = a->outer_count; p.inner_count = a->inner_count;
  p.in_outer_stride = a->in_outer_stride; p.in_k_stride = a->in_k_stride;
  p.out_outer_stride = a->out_outer_stride; p.out_k_stride = a->out_k_stride;
  p.C = a->C; p.outer_per_sample = a->outer_per_sample; p.act = a->act;
  hipStream_t st = (hipStream_t)stream;
  if (a->dtype == AFLDM_F32) return sep_dispatch<float>(p, a->K, a->R, a->R2, a->up_identity, st);
  if (a->dtype == AFLDM_BF16) return sep_dispatch<bf16>(p, a->K, a->R, a->R2, a->up_identity, st);
  set_error("afldm_sep_pass: unknown dtype %d", a->dtype);
  return AFLDM_EDTYPE;
}

extern "C" int afldm_gn_table(const float* stats, int S, const float* gamma, const float* beta, float* table, int B, int C,
                              int G, int HW, float eps, afldm_stream_t stream) {
  AFLDM_REQUIRE(stats && gamma && beta && table, AFLDM_ENULL, "afldm_gn_table: NULL pointer");
  AFLDM_REQUIRE(B > 0 && C > 0 && G > 0 && C % G == 0 && HW > 0 && S > 0, AFLDM_ESHAPE, "afldm_gn_table: bad shape");
  const GnStats gs{stats, nullptr, C, 0, S, 0};
  k_gn_table<<<(B * G + 3) / 4, 256, 0, (hipStream_t)stream>>>(gs, gamma, beta, table, B, C, G, (double)HW * (C / G), eps);
  return check_launch("afldm_gn_table");
}

extern "C" int afldm_softmax_rows(const void* x, void* y, long long rows, int cols, float scale, int dtype,
                                  afldm_stream_t stream) {
  AFLDM_REQUIRE(x && y, AFLDM_ENULL, "afldm_softmax_rows: NULL pointer");
  AFLDM_REQUIRE(rows > 0 && rows < (1ll << 31) && cols > 0, AFLDM_ESHAPE, "afldm_softmax_rows: bad shape");
  hipStream_t st = (hipStream_t)stream;
  DISPATCH_T(dtype, (k_softmax_rows<float><<<(int)rows, 256, 0, st>>>((const float*)x, (float*)y, cols, scale)),
             (k_softmax_rows<bf16><<<(int)rows, 256, 0, st>>>((const bf16*)x, (bf16*)y, cols, scale)), "afldm_softmax_rows");
  return check_launch("afldm_softmax_rows");
}
