// af_plane.hpp - argument block, LDS plan and small helpers of the plane form of the alias-free activation (N = 16, 32;
// design notes: af.hip), shared by af.hip (k_af_act_plane) and the merged launches (actconv.hip).
#pragma once
#include "common.hpp"

namespace afldm {

template <typename T>
struct AfP {
  const T* x1;
  const T* x2;
  GnStats gs;          // per-channel GroupNorm partial sums (gs.st1 == nullptr: no normalisation)
  const float* gamma;
  const float* beta;
  const float* U;  // [2N][N]   (small-plane kernel)
  const float* D;  // [N][2N]
  const void* packed;  // LDS image of the matrices for the MFMA kernel (afldm_af_pack)
  T* y;
  int C1, C2, G, B;
  float eps;
  unsigned long long* trace;   // diagnostic (k_af_act_plane): [workgroup][wave][item 0 / 1][10] s_memtime stamps, or NULL
  int stagger;                 // A/B (AFLDM_AF_STAGGER): s_sleep units (64 clocks) per co-residency slot a workgroup waits before its first item
  int y_blocked;               // output layout: 0 = NHWC [B][N][N][C]; 2 = one value per plane [B][C] (N = 2: the result is plane-constant); 1 = 16-byte channel blocks [B][C/EPC][N][N][EPC] (afldm_conv_args.x_layout = 1)
  int x_blocked;               // the same for the input x1 (no virtual concat then: C2 == 0)
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

// ----------------------------------------------------------------------------- plane kernel (N = 16, 32)
template <typename T, int N, int CH = 16 /* channels per item: 16, or 8 when that fills the CUs more evenly */>
struct PlaneCfg {
  typedef Mma<T> MM;
  static constexpr int EPC = MM::EPC, KPF = MM::KPF;
  static constexpr int H2 = 2 * N;
  static constexpr int KH = ((N + KPF - 1) / KPF) * KPF;  // K extent for contractions over an N-long axis
  static constexpr int NKF1 = KH / KPF;                    // K steps over an N-long axis (h, w)
  static constexpr int NKF3 = H2 / KPF;                    // K steps over a 2N-long axis (h', w')
  static constexpr int TN = N / 16, TH = H2 / 16;          // 16-row tiles of an N / 2N axis
  static constexpr int NW = 4;                             // waves per workgroup
  static constexpr int CPW = CH / NW;                      // channel planes per wave
  static constexpr bool PERM = sizeof(T) == 2;             // chained operands permute K (see Mma<T>)
  static constexpr bool CREG = sizeof(T) == 2;             // constant fragments cached in registers
  // LDS rows are K-contiguous runs read as 16-byte chunks by 16 lanes at a time; one chunk of
  // padding per row makes the 16 row starts hit 16 distinct 4-bank groups (guide G4).
  static constexpr int KHP = KH + EPC;                     // Xs row stride   [c][w][h]
  static constexpr int H2P = H2 + EPC;                     // Vt row stride   [w][h']
  static constexpr int YRP = N * CH + 8;                   // Ys row stride   [h][w*CH + c]
  // constant fragments, each 64 lanes x EPC elements in lane order (one conflict-free 16-byte read):
  //   ufB[th][kf]  U rows 16th+li, k = h standard          (B operand of P1)
  //   upA[t2][f]   U rows 16t2+li, k = w chain-permuted     (A operand of P2)
  //   dpA[t3][f]   D rows 16t3+li, k = w' chain-permuted    (A operand of P3)
  //   dA [t4][kf]  D rows 16t4+li, k = h' standard          (A operand of P4)
  static constexpr int NF_U = TH * NKF1, NF_D = TN * NKF3;
  static constexpr int F_UFB = 0, F_UPA = NF_U, F_DPA = 2 * NF_U, F_DA = 2 * NF_U + NF_D;
  static constexpr int NFRAG = 2 * NF_U + 2 * NF_D;
  static constexpr int CONST_ELEMS = NFRAG * 64 * EPC;
  static constexpr int XS = CH * N * KHP;                  // X tile; re-used as the output staging tile Ys
  static constexpr int VT = N * H2P;                       // per wave: V^T of one plane
  // bf16: the constant fragments are copied to registers once, so their LDS image shares the Vt
  // region (N = 32, 8-channel items: 54 -> 38 KB, three workgroups per CU instead of two)
  static constexpr int CV = CREG ? (CONST_ELEMS > NW * VT ? CONST_ELEMS : NW * VT) : CONST_ELEMS + NW * VT;
  static constexpr int LDS_BYTES = (XS + CV) * (int)sizeof(T) + 6 * 16 * (int)sizeof(float);   // + scale / shift (2 items) + group scratch
  static_assert(N * YRP <= XS, "output staging tile must fit in the X region");
  static_assert(NW * 16 * 2 * 8 <= NW * VT * (int)sizeof(T), "GroupNorm reduction scratch aliases Vt");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

// K index of element e of the chunk held by lane group g in K step f.  Standard operands (read
// from LDS / memory): consecutive.  Chained operands (an MFMA accumulator re-used as the B operand
// of the next MFMA): fp32 accumulators are already in standard order; two bf16 accumulator tiles
// pack into one chunk as (lo = tile 2f rows 4g..4g+3, hi = tile 2f+1 rows 4g..4g+3).
template <typename T>
__host__ __device__ constexpr int af_kidx(int f, int g, int e, bool chained) {
  return (chained && sizeof(T) == 2) ? 32 * f + (e < 4 ? 4 * g + e : 16 + 4 * g + (e - 4))
                                     : f * Mma<T>::KPF + g * Mma<T>::EPC + e;
}

// z' / (1 + exp2(-z')) on 4 accumulator values (z' = z log2(e); the caller's next matrix carries the
// ln(2)): two packed adds / multiplies and 8 transcendentals instead of 12 + 8 scalar operations.
__device__ __forceinline__ f32x4 silu_log2_x4(const f32x4& z) {
  f32x4 d;
#pragma unroll
  for (int r = 0; r < 4; ++r) d[r] = __builtin_amdgcn_exp2f(-z[r]);
  d = d + 1.0f;
#pragma unroll
  for (int r = 0; r < 4; ++r) d[r] = __builtin_amdgcn_rcpf(d[r]);
  return z * d;
}

}  // namespace afldm
