// lin.hip — the fused q|k|v projection of the attention blocks (K = 192 / 384) with the WEIGHTS IN
// REGISTERS.  y[m][n] = sum_k x[m][k] W[n][k] + bias[n]; couts < split_n leave token-major (q | k), couts
// >= split_n channel-major (the V^T operand of the attention kernel).
// Replaces nn.Linear inside diffusers' Attention (reference cross_frame_attn.py:66-130 runs
// AttnProcessor2_0: to_q / to_k / to_v / to_out[0]).
//
// These GEMMs are HBM-bound (K <= 384: 100 MB of tensors against 14.5 GFLOP for q|k|v at 32x32) and
// the general implicit-GEMM tile pipeline (conv.hip) spends its time in per-tile prologue / epilogue:
// 56 us for 100 MB.  Here a workgroup is persistent:
//   * wave w of the 4 compute waves owns NPW couts of the workgroup's BN = 4 NPW cout chunk and
//     keeps their K-complete weight rows as MFMA A fragments in registers for the whole kernel
//     (72 / 96 VGPRs) - weights are read from L2 once per workgroup, never through LDS;
//   * a 5th wave only issues the LDS-DMA (buffer_load ... lds) of the next BM-token x tile into a
//     two-slot ring (128-byte rows, XOR-swizzled source chunks as in conv.hip) and waits for it with
//     its own vmcnt, so the compute waves' loads / stores never sit in the same counter;
//   * ONE s_barrier per tile; the epilogue is wave-private (a wave owns whole cout columns: its
//     patch goes through a private LDS patch and leaves as 16-byte pieces of token rows or, for
//     V^T, of cout rows).
// Two workgroups per CU cover each other's latencies.  Summation order (K ascending in 32-wide MFMA
// steps, then bias, one rounding) is the one of conv.hip: results are bit-identical to the general
// kernel.  to_out (residual + GroupNorm sums) was measured 30-60 % slower in this form than on the
// general kernel and stays there; the q|k|v GEMM gains 15-20 % (35 vs 42 us at 32x32, batch 64).
#include "common.hpp"

namespace afldm {

typedef __attribute__((address_space(3))) void* lin_lds_ptr_t;

struct LinP {
  const bf16* x;
  const bf16* w;
  const float* bias;
  bf16* y;
  bf16* y2;
  int M, N, HW, split_n, y_ld, ntiles, nchunks, dbg;
};

template <int K, int NPW, int BM, int NST_, int NWC_>
struct LinCfg {
  static constexpr int NWC = NWC_, BN = NPW * NWC, TN = NPW / 16, TM = BM / 16, KF = K / 32, NPL = K / 64;
  static constexpr int PLANE = BM * 128, XSTAGE = NPL * PLANE, NST = NST_;
  static constexpr int RG = BM / 8, DMA_PER_TILE = NPL * RG;
  static constexpr int SROW = NPW + 8, TROW = BM + 8;   // bf16 elements; both row strides are multiples of 16 bytes
  static constexpr int WSTG = (BM * SROW > NPW * TROW ? BM * SROW : NPW * TROW) * 2;
  static constexpr int LDS = NST * XSTAGE + NWC * WSTG;
  static_assert(K % 64 == 0 && NPW % 16 == 0 && BM % 32 == 0, "tile shape");
  static_assert((SROW * 2) % 16 == 0 && (TROW * 2) % 16 == 0, "16-byte staged rows");
};

template <int K, int NPW, int BM, int NST, int NWC, int NPROD, int MINW>
__global__ void __launch_bounds__((NWC + NPROD) * 64, MINW) k_lin_wreg(LinP p) {
  typedef LinCfg<K, NPW, BM, NST, NWC> CF;
  typedef Mma<bf16> MM;
  typedef MM::Chunk Chunk;
  constexpr int TN = CF::TN, TM = CF::TM, KF = CF::KF;
  extern __shared__ __attribute__((aligned(16))) char lin_smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const int G = gridDim.x;
  const int l = xcd_remap(blockIdx.x, G);     // consecutive logical ids share an XCD (and there the x tile in L2)
  const int chunk = l % p.nchunks, t0 = l / p.nchunks, tstride = G / p.nchunks;
  const int nmine = t0 < p.ntiles ? (p.ntiles - t0 + tstride - 1) / tstride : 0;

  if (wave >= CF::NWC) {
    const int pw = wave - CF::NWC;                 // producer index: instruction j belongs to producer j % NPROD
    static_assert(CF::DMA_PER_TILE % NPROD == 0, "DMA instructions must divide among the producer waves");
    // ---------------- producer: x tile t -> ring slot.  Instruction j covers plane j / RG, rows 8 (j % RG) .. +7;
    // lane l: row + (l >> 3), LDS position l & 7 holds source chunk (l & 7) ^ ((row >> 1) & 7).
    __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)((long long)p.M * K * 2), 0x00020000);
    auto issue = [&](int slot, int t) {
      const int soff = t * BM * K * 2;
#pragma unroll
      for (int jj = 0; jj < CF::DMA_PER_TILE / NPROD; ++jj) {
        const int j = pw + NPROD * jj;
        const int plane = j / CF::RG, rg = j % CF::RG;
        const int row = 8 * rg + (lane >> 3);
        const int src = (lane & 7) ^ ((row >> 1) & 7);
        const int voff = (row * K + plane * 64 + src * 8) * 2;
        lin_lds_ptr_t dst = (lin_lds_ptr_t)(lin_smem + slot * CF::XSTAGE + plane * CF::PLANE + rg * 1024);
        if (!(p.dbg & 1)) __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, dst, 16, voff, soff, 0, 0);
      }
    };
    // NST - 1 tiles run ahead of the compute waves
#pragma unroll
    for (int q = 0; q < NST - 1; ++q)
      if (q < nmine) issue(q, t0 + q * tstride);
    int slot = NST - 1;
    for (int i = 0; i < ((p.dbg & 32) ? 1 : nmine); ++i) {
      // tile i has landed once at most the NST - 2 younger tiles are outstanding (fewer at the tail)
      if (i + NST - 1 <= nmine) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * CF::DMA_PER_TILE / NPROD) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                    // tile i visible; the slot of tile i - 1 is free again
      if (i + NST - 1 < nmine) issue(slot, t0 + (i + NST - 1) * tstride);
      slot = slot + 1 == NST ? 0 : slot + 1;
    }
    return;
  }

  // ---------------- compute waves
  const int n0w = chunk * CF::BN + wave * NPW;
  Chunk wreg[TN][KF];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn)
#pragma unroll
    for (int kf = 0; kf < KF; ++kf)
      wreg[tn][kf] = (p.dbg & 16) ? MM::zero() : ld16<Chunk>(p.w + (size_t)(n0w + 16 * tn + li) * K + 32 * kf + 8 * lg);
  // A workgroup's cout chunk lies wholly in the token-major (q | k) or in the channel-major (V^T) part.
  // An MFMA A / B fragment has the same per-lane layout, so the V^T chunks simply swap the operands:
  //   q|k : A = W, B = x  ->  a lane holds 4 consecutive COUTS of one token   (8-byte pieces of a token row)
  //   V^T : A = x, B = W  ->  a lane holds 4 consecutive TOKENS of one cout   (8-byte pieces of a cout row)
  const bool vt = n0w >= p.split_n;
  const int cv = p.N - p.split_n;
  f32x4 bq[TN];     // q|k: bias of couts 16 tn + 4 lg + r;  V^T: bias of cout 16 tn + li in every component
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    if (!p.bias) bq[tn] = f32x4{0.f, 0.f, 0.f, 0.f};
    else if (vt) { const float b1 = p.bias[n0w + 16 * tn + li]; bq[tn] = f32x4{b1, b1, b1, b1}; }
    else bq[tn] = *reinterpret_cast<const f32x4*>(p.bias + n0w + 16 * tn + 4 * lg);
  }
  bf16* ws = reinterpret_cast<bf16*>(lin_smem + CF::NST * CF::XSTAGE + wave * CF::WSTG);

  for (int i = 0; i < ((p.dbg & 32) ? 1 : nmine); ++i) {
    const int t = t0 + i * tstride;
    const int m0 = t * BM;
    __builtin_amdgcn_s_barrier();
    const char* sx = lin_smem + (i % NST) * CF::XSTAGE;
    // the tile is walked in 32-token halves: 2 x TN accumulator tiles live at a time next to the weights
#pragma unroll 1
    for (int h = 0; h < TM / 2; ++h) {
      f32x4 acc[TN][2];
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) acc[tn][tm] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (!(p.dbg & 2)) {
#pragma unroll
        for (int kf = 0; kf < KF; ++kf) {
          const char* pl = sx + (kf >> 1) * CF::PLANE;
          Chunk b[2];
#pragma unroll
          for (int tm = 0; tm < 2; ++tm) {
            const int row = 32 * h + 16 * tm + li;
            b[tm] = ld16<Chunk>(pl + row * 128 + ((((kf & 1) * 4 + lg) ^ ((row >> 1) & 7)) << 4));
          }
          if (vt) {
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
              for (int tm = 0; tm < 2; ++tm) MM::mma(acc[tn][tm], b[tm], wreg[tn][kf]);
          } else {
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
              for (int tm = 0; tm < 2; ++tm) MM::mma(acc[tn][tm], wreg[tn][kf], b[tm]);
          }
          if (kf & 1) __builtin_amdgcn_sched_barrier(0);   // at most two K steps of x fragments in flight (registers)
        }
      }
      // ---- wave-private staging
      if (!(p.dbg & 4)) {
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
          for (int tm = 0; tm < 2; ++tm) {
            bf16x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (bf16)(acc[tn][tm][r] + bq[tn][r]);
            if (vt) *reinterpret_cast<bf16x4*>(ws + (16 * tn + li) * CF::TROW + 32 * h + 16 * tm + 4 * lg) = o;
            else *reinterpret_cast<bf16x4*>(ws + (32 * h + 16 * tm + li) * CF::SROW + 16 * tn + 4 * lg) = o;
          }
      }
    }
    __builtin_amdgcn_wave_barrier();
    if (p.dbg & 8) continue;
    if (vt) {
      const int bsm = m0 / p.HW, tok0 = m0 - bsm * p.HW;
      constexpr int CPR = BM / 8;                        // 16-byte token runs per cout
#pragma unroll 2
      for (int it = 0; it < NPW * CPR / 64; ++it) {
        const int idx = lane + 64 * it, row = idx / CPR, c = idx - row * CPR;
        st16_out<Chunk>(p.y2 + ((size_t)bsm * cv + (n0w - p.split_n + row)) * p.HW + tok0 + c * 8,
                    ld16<Chunk>(ws + row * CF::TROW + c * 8));
      }
    } else {
      constexpr int CPR = NPW / 8;                       // 16-byte pieces of this wave's part of a row
#pragma unroll 2
      for (int it = 0; it < BM * CPR / 64; ++it) {
        const int idx = lane + 64 * it, row = idx / CPR, c = idx - row * CPR;
        st16_out<Chunk>(p.y + (size_t)(m0 + row) * p.y_ld + n0w + c * 8, ld16<Chunk>(ws + row * CF::SROW + c * 8));
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

template <int K, int NPW, int BM, int NST, int NWC, int NPROD, int MINW>
static int launch_lin(LinP p, hipStream_t st) {
  typedef LinCfg<K, NPW, BM, NST, NWC> CF;
  static unsigned long long attr_set = 0;
  if (first_on_device(attr_set)) {
    (void)hipFuncSetAttribute((const void*)k_lin_wreg<K, NPW, BM, NST, NWC, NPROD, MINW>, hipFuncAttributeMaxDynamicSharedMemorySize, CF::LDS);
  }
  p.ntiles = p.M / BM;
  p.nchunks = p.N / CF::BN;
  const long long items = (long long)p.ntiles * p.nchunks;
  long long g = items < 512 ? items : 512;
  g = (g / p.nchunks) * p.nchunks;
  k_lin_wreg<K, NPW, BM, NST, NWC, NPROD, MINW><<<(int)g, (NWC + NPROD) * 64, CF::LDS, st>>>(p);
  return check_launch("afldm_conv2d(lin_wreg)");
}

// Tile height of the weights-in-registers kernel for this call, 0 when it does not apply.
int lin_wreg_bm(const afldm_conv_args* a) {
  static const bool off = getenv("AFLDM_NO_LIN_WREG") && atoi(getenv("AFLDM_NO_LIN_WREG")) != 0;
  if (off || a->dtype != AFLDM_BF16 || a->KS != 1 || a->C2 != 0 || a->temb || a->out_mode != 0 || a->w_batch_stride) return 0;
  const int K = a->C1;
  if (K != 192 && K != 384) return 0;
  const int BM = 32, BN = K == 192 ? 192 : 128;
  const long long M = (long long)a->B * a->H * a->W;
  const int HW = a->H * a->W;
  if (M < 4096 || M % BM || HW % BM || a->Cout % BN || M * K * 2 >= (1ll << 31)) return 0;
  // the 8 x 8 level (its only remaining site at batch 64 since the 32^2 / 16^2 projections moved into k_attn_fused): the general
  // kernel's 128 x 64 tiles are ~2.5 us faster per launch there - 4.619 -> 4.606 ms/step, same box, four alternating rounds
  if (HW <= 64) return 0;
  // the fused q|k|v projection only: to_out (residual + statistics) measured faster on the general kernel
  if (!a->y2 || a->residual || a->stats_out || a->split_n % BN || a->split_n <= 0 || a->split_n >= a->Cout || HW % 8) return 0;
  if (a->y_ld % 8) return 0;
  if (!aligned16(a->x1) || !aligned16(a->w) || !aligned16(a->y) || !aligned16(a->y2) || (a->bias && !aligned16(a->bias)))
    return 0;
  return BM;
}

int lin_wreg_launch(const afldm_conv_args* a, hipStream_t st) {
  LinP p;
  p.x = (const bf16*)a->x1; p.w = (const bf16*)a->w; p.bias = a->bias;
  p.y = (bf16*)a->y; p.y2 = (bf16*)a->y2;
  p.M = a->B * a->H * a->W; p.N = a->Cout; p.HW = a->H * a->W;
  p.split_n = a->split_n;
  p.y_ld = a->y_ld;
  p.ntiles = p.nchunks = 0;
  static const int s_dbg = getenv("AFLDM_LIN_DBG") ? atoi(getenv("AFLDM_LIN_DBG")) : 0;   // timing decomposition (garbage results)
  p.dbg = s_dbg;
  static const int s_wide = getenv("AFLDM_LIN_WIDE") ? atoi(getenv("AFLDM_LIN_WIDE")) : 0;
  if (s_wide == 1) return a->C1 == 192 ? launch_lin<192, 48, 32, 4, 4, 1, 3>(p, st) : launch_lin<384, 32, 32, 2, 4, 1, 3>(p, st);
  if (s_wide == 2) return a->C1 == 192 ? launch_lin<192, 16, 32, 4, 12, 1, 7>(p, st) : launch_lin<384, 16, 32, 2, 8, 1, 5>(p, st);
  return a->C1 == 192 ? launch_lin<192, 16, 32, 4, 12, 2, 7>(p, st) : launch_lin<384, 16, 32, 2, 8, 2, 5>(p, st);
}

}  // namespace afldm
