// attn.hip — scaled-dot-product attention (softmax(q k^T * scale) v) on MFMA, flash-style
// (online softmax, S never materialised), tuned for the UNet's shape: head_dim 24, T <= 1024.
//
// Measured character (MI355X, T = 1024, 8 heads of 24, batch 64): NOT softmax/VALU bound.  With the
// exponentials, the maxima and every MFMA removed the first version still took 172 of its 150-210 us:
// the time goes into fetching K / V^T (K rows of one head are 48-byte runs at a 1152-byte stride: a
// 128-byte line fill per key for 48 useful bytes, ~60 B/clk/CU of L1 fill with 128 queries per
// workgroup) and into the per-chunk barrier.  Hence:
//
//  * a workgroup = NW waves (8 for T >= 256) x 32 queries of one (batch, head): every staged K / V^T
//    chunk is reused by 256 queries, halving the fill traffic and the barriers per query;
//  * K and V^T are staged per 64-key chunk through LDS (double-buffered) with a TWO-chunk register
//    prefetch (chunk c+2 is in flight while chunk c is computed), in fragment order: every A
//    fragment is one conflict-free 16-byte LDS read (64-B rows, chunk index XOR-swizzled by the row);
//  * one wave owns 32 queries (two 16-query B fragments sharing every K / V^T A fragment);
//  * S^T = K Q^T (A = K rows, B = Q rows): lane (j = query, g) holds the scores of ITS query for
//    keys {4g + r} of each 16-key tile -> the row max is 16 local values + 2 shuffles, and P^T is
//    already the B operand of the second MFMA, O^T = V^T P^T (A = V^T rows);
//  * scale * log2(e) is folded into Q once; the running reference max is subtracted BY THE MFMA
//    (K carries a constant 1, Q carries -m_run in the first padding channel of head_dim) and only
//    moved when a row max outgrows it by 2^10 (lazy rescale: exact up to rounding, numerator and
//    denominator share the reference); the row sums come out of the second MFMA through a row of
//    ones appended to V^T.  A score costs max + exp2 + cvt on the VALU;
//  * V arrives channel-major (vt[b][c][t], written by afldm_conv2d out_mode 1), so V^T rows are
//    contiguous key runs and O^T leaves 4 consecutive head channels of one query per lane.
#include <stdlib.h>

#include "common.hpp"

namespace afldm {

template <typename T>
struct AttnP {
  const T* q;
  const T* k;
  const T* vt;
  T* o;
  int ldq, ldk, ldo;
  int B, Bk, heads, Tq, Tk, d;
  float scale_log2e;
  int qblocks;  // query blocks (of 32 * waves) per (b, head)
};

__device__ __forceinline__ int aswz(int row) { return (4 - ((row >> 2) & 3)) & 3; }

constexpr int KC = 64;              // keys per chunk
constexpr float LAZY_TAU = 10.0f;   // rescale only when a row max grows by more than 2^10

template <typename T, int ND /* 16-row tiles of V^T: head_dim rows + the row of ones */,
          int NKF /* chunk pairs covering head_dim + the -m_run channel in QK^T */, int NW /* waves */,
          bool RAGGED /* Tk % 64 != 0: clamped / element-wise staging and key masking */,
          int DBG = 0 /* timing decomposition (AFLDM_ATTN_DBG): 1 no exp2, 2 no P V MFMA, 4 no S MFMA, 8 no max,
                         16 no per-chunk staging, 32 no per-chunk barrier; results are garbage */>
__global__ void __launch_bounds__(NW * 64) k_attn(AttnP<T> p) {
  typedef Mma<T> MM;
  typedef typename MM::Chunk Chunk;
  constexpr int EPC = MM::EPC, KPF = MM::KPF;
  constexpr bool BF = sizeof(T) == 2;
  constexpr int NPV = KC / KPF;                  // chunk pairs covering the 64 keys in P V (2 bf16 / 4 fp32)
  constexpr int KT_BYTES = NKF * KC * 64;        // K tile:  [kf][key][64 B]
  constexpr int VT_BYTES = NPV * ND * 16 * 64;   // V^T tile: [pv][d row][64 B] (fragment-ordered keys)
  constexpr int SMEM_BYTES = 2 * (KT_BYTES + VT_BYTES);
  constexpr int NT = NW * 64;
  __shared__ __attribute__((aligned(16))) char smem[SMEM_BYTES];
  char* sK = smem;
  char* sV = smem + 2 * KT_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;

  // XCD-aware order: the query blocks of one (batch, head) - and neighbouring heads, which share
  // K's cache lines - run on the same XCD and hit its L2
  int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int qb = bid % p.qblocks;
  bid /= p.qblocks;
  const int h = bid % p.heads;
  const int b = bid / p.heads;
  const int kb = b / (p.B / p.Bk);
  const int q0 = (qb * NW + wave) * 32;
  const int C = p.heads * p.d;

  const int kf_pad = p.d / KPF, lg_pad = (p.d % KPF) / EPC;

  // ---- Q fragments, two 16-query tiles (raw loads first: every global load of the prologue is
  // issued before anything waits on one)
  Chunk qf[2][NKF];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int qrow = q0 + 16 * u + li;
    const bool qok = qrow < p.Tq;
    const T* qptr = p.q + ((size_t)b * p.Tq + (qok ? qrow : 0)) * p.ldq + h * p.d;
#pragma unroll
    for (int kf = 0; kf < NKF; ++kf) {
      const int e0 = kf * KPF + lg * EPC;
      qf[u][kf] = (qok && e0 + EPC <= p.d) ? ld16<Chunk>(qptr + e0) : MM::zero();   // scaled below, after the K / V loads are out
    }
  }

  // ---- staging assignment: 16-byte pieces of the K chunk and of the V^T chunk.  Every predicate
  // and address component that does not depend on the chunk is computed here, once:
  //   K : NKF * KC rows x 4 pieces, live when the piece lies inside head_dim
  //   V^T: ND*16 d-rows x (KC*sizeof(T)/16) pieces, live when the row is a head channel
  // (with 8 waves the K pieces go to the lower half of the workgroup and the V^T pieces to the upper)
  constexpr int KPIECES = NKF * KC * 4;
  constexpr int VPR = KC * (int)sizeof(T) / 16;   // 16-B pieces per V^T row per chunk
  constexpr int VPIECES = ND * 16 * VPR;
  constexpr int KPT = (KPIECES + NT - 1) / NT, VPT = (VPIECES + NT - 1) / NT;
  constexpr int VROT = (KPIECES <= NT / 2 && VPIECES <= NT / 2) ? NT / 2 : 0;
  const T* kbase = p.k + (size_t)kb * p.Tk * p.ldk + h * p.d;
  const T* vbase = p.vt + ((size_t)kb * C + h * p.d) * p.Tk;
  const bool tiny = RAGGED && p.Tk < EPC;    // Tk = 4 in bf16: less than one 16-byte piece per V^T row
  bool kact[KPT], vact[VPT];
  int krow[KPT], ksrc[KPT], kdst[KPT], vkey[VPT], vsrc[VPT], vdst[VPT], vswz[VPT];
#pragma unroll
  for (int i = 0; i < KPT; ++i) {
    const int id = tid + i * NT;
    const int piece = id & 3, row = (id >> 2) % KC, kf = (id >> 2) / KC;
    const int e0 = kf * KPF + piece * EPC;
    kact[i] = id < KPIECES && e0 + EPC <= p.d;
    krow[i] = row;
    ksrc[i] = row * p.ldk + e0;
    kdst[i] = (kf * KC + row) * 64 + ((piece ^ aswz(row)) << 4);
  }
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const int id = (tid + VROT) % NT + i * NT;
    const int piece = id % VPR, drow = id / VPR;
    vact[i] = id < VPIECES && drow < p.d;
    vkey[i] = piece * EPC;
    vsrc[i] = drow * p.Tk + piece * EPC;
    vswz[i] = aswz(drow);
    if constexpr (BF) {
      // 8 consecutive keys 8j..8j+7 of a 32-key half: keys 8j..8j+3 -> group g = 2(j&1), keys
      // 8j+4..8j+7 -> g = 2(j&1)+1; element offset 0 for j < 2 (keys < 16), 4 otherwise.
      const int half = piece >> 2, j = piece & 3;
      vdst[i] = (half * ND * 16 + drow) * 64 + (j >> 1) * 8;
      vswz[i] = ((2 * (j & 1)) ^ aswz(drow)) << 4;          // byte offset of group g0; g0 + 1 is (that ^ 16)
    } else {
      // fp32: 4 consecutive keys = chunk g of 16-key tile t
      const int t = piece >> 2, g = piece & 3;
      vdst[i] = (t * ND * 16 + drow) * 64 + ((g ^ aswz(drow)) << 4);
    }
  }
  const T* kp[KPT];     // chunk-aligned levels: per-piece source pointers, advanced by one chunk per load
  const T* vp[VPT];
  // (lanes without a live piece still load - from a valid address - so that the chunk loop has
  //  no control flow around its global loads: with branches there the compiler's s_waitcnt
  //  insertion fell back to vmcnt(0) right after the loads and the prefetch was fully exposed)
#pragma unroll
  for (int i = 0; i < KPT; ++i) kp[i] = kbase + (kact[i] ? ksrc[i] : krow[i] * p.ldk);
#pragma unroll
  for (int i = 0; i < VPT; ++i) vp[i] = vbase + (vact[i] ? vsrc[i] : vkey[i]);

  auto load_chunk = [&](int key0, bool advance, Chunk (&rk)[KPT], Chunk (&rv)[VPT]) {
    if constexpr (!RAGGED) {
      const int kstep = advance ? KC * p.ldk : 0, vstep = advance ? KC : 0;   // past the end: reload the last chunk
#pragma unroll
      for (int i = 0; i < KPT; ++i) {
        rk[i] = ld16<Chunk>(kp[i]);
        kp[i] += kstep;
      }
#pragma unroll
      for (int i = 0; i < VPT; ++i) {
        rv[i] = ld16<Chunk>(vp[i]);
        vp[i] += vstep;
      }
      return;
    }
    if (key0 >= p.Tk) return;
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
      if (kact[i]) {
        // ragged last chunk: clamp to a valid key (its scores are masked below; the value only has to be finite)
        int off = key0 * p.ldk + ksrc[i];
        if (key0 + krow[i] >= p.Tk) off = (p.Tk - 1) * p.ldk + ksrc[i] - krow[i] * p.ldk;
        rk[i] = ld16<Chunk>(kbase + off);
      }
    }
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
      if (vact[i]) {
        if (tiny) {   // element-wise, zero tail
          rv[i] = MM::zero();
#pragma unroll
          for (int e = 0; e < EPC; ++e)
            if (key0 + vkey[i] + e < p.Tk) rv[i][e] = vbase[vsrc[i] + key0 + e];
        } else {
          int off = vsrc[i] + key0;
          if (key0 + vkey[i] + EPC > p.Tk) off = vsrc[i] - vkey[i] + p.Tk - EPC;   // masked keys: any finite data
          rv[i] = ld16<Chunk>(vbase + off);
        }
      }
    }
  };
  auto store_chunk = [&](int buf, const Chunk (&rk)[KPT], const Chunk (&rv)[VPT]) {
#pragma unroll
    for (int i = 0; i < KPT; ++i)
      if (kact[i]) st16<Chunk>(sK + buf * KT_BYTES + kdst[i], rk[i]);
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
      if (vact[i]) {
        char* base = sV + buf * VT_BYTES + vdst[i];
        if constexpr (BF) {
          bf16x4 lo, hi;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            lo[e] = rv[i][e];
            hi[e] = rv[i][4 + e];
          }
          *reinterpret_cast<bf16x4*>(base + vswz[i]) = lo;
          *reinterpret_cast<bf16x4*>(base + (vswz[i] ^ 16)) = hi;
        } else {
          st16<Chunk>(base, rv[i]);
        }
      }
    }
  };

  f32x4 oacc[2][ND];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int t = 0; t < ND; ++t) oacc[u][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  // m_run: the reference point the exponentials are taken against (log2 units, representable in T).
  float m_run[2] = {0.f, 0.f};

  auto compute = [&](int key0, int buf) {
    // ---- S^T - m_run: 4 key tiles x 2 query tiles
    f32x4 s[2][4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      Chunk kfz[NKF];
#pragma unroll
      for (int kf = 0; kf < NKF; ++kf) {
        const int row = 16 * t + li;
        kfz[kf] = ld16<Chunk>(sK + buf * KT_BYTES + (kf * KC + row) * 64 + ((lg ^ aswz(row)) << 4));
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        s[u][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kf = 0; kf < NKF; ++kf)
          if (!(DBG & 4)) MM::mma(s[u][t], kfz[kf], qf[u][kf]);
      }
    }
    if (RAGGED && key0 + KC > p.Tk) {  // ragged last chunk: mask keys >= Tk (wave-uniform branch)
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (key0 + 16 * t + 4 * lg + r >= p.Tk) s[u][t][r] = -1e30f;
    }
    // ---- row maxima of the shifted scores
    float mloc[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      float m = s[u][0][0];
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (!(DBG & 8)) m = fmaxf(m, s[u][t][r]);
      m = fmaxf(m, __shfl_xor(m, 16, 64));
      mloc[u] = fmaxf(m, __shfl_xor(m, 32, 64));
    }
    const bool first = key0 == 0;
    if (first || __any((mloc[0] > LAZY_TAU) || (mloc[1] > LAZY_TAU))) {   // wave-uniform, rare after chunk 0
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        // the new reference is rounded to T (it travels in Q); any value works as long as the
        // scores, the numerator and the denominator all use the same one
        const float m_new = to_f32(from_f32<T>(m_run[u] + (first ? mloc[u] : fmaxf(mloc[u], 0.f))));
        const float delta = m_new - m_run[u];
        const float alpha = __builtin_amdgcn_exp2f(-delta);
        m_run[u] = m_new;
#pragma unroll
        for (int kf = 0; kf < NKF; ++kf)
          if (kf == kf_pad && lg == lg_pad) qf[u][kf][0] = from_f32<T>(-m_new);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) s[u][t][r] -= delta;
#pragma unroll
        for (int t = 0; t < ND; ++t) oacc[u][t] *= alpha;
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (!(DBG & 1)) s[u][t][r] = __builtin_amdgcn_exp2f(s[u][t][r]);
    // ---- O^T += V^T P^T   (row d of V^T is ones: O^T row d = running softmax denominator)
#pragma unroll
    for (int pv = 0; pv < NPV; ++pv) {
      Chunk pb[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if constexpr (BF) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            pb[u][r] = (bf16)s[u][2 * pv][r];
            pb[u][4 + r] = (bf16)s[u][2 * pv + 1][r];
          }
        } else {
          pb[u] = s[u][pv];
        }
      }
#pragma unroll
      for (int td = 0; td < ND; ++td) {
        const int row = 16 * td + li;
        Chunk va = ld16<Chunk>(sV + buf * VT_BYTES + (pv * ND * 16 + row) * 64 + ((lg ^ aswz(row)) << 4));
#pragma unroll
        for (int u = 0; u < 2; ++u)
          if (!(DBG & 2)) MM::mma(oacc[u][td], va, pb[u]);
      }
    }
  };

  // ---- chunk loop, two register stages: while chunk c is computed from LDS buffer c & 1, chunk
  // c + 1 waits in registers (stored to the other buffer at the end of the step) and chunk c + 2 is
  // in flight from global memory.
  Chunk rkA[KPT], rvA[VPT], rkB[KPT], rvB[VPT];
  const int nchunks = (p.Tk + KC - 1) / KC;
  load_chunk(0, 1 < nchunks, rkA, rvA);     // pointers now at chunk 1 (if there is one)
  load_chunk(KC, 2 < nchunks, rkB, rvB);    // chunk 1 (or chunk 0 again when there is only one)
  // ---- LDS padding is written ONCE: K pieces beyond head_dim and V^T rows beyond head_dim stay
  // zero in both buffers, V^T row `d` is all ones (-> row d of O^T accumulates the softmax
  // denominator on the MFMA pipe, no VALU adds), K channel slot `d` is a constant 1 (Q carries
  // -m_run there); the chunk loop only rewrites the live pieces.
  for (int idx = tid; idx < SMEM_BYTES / 16; idx += NT) st16<Chunk>(smem + idx * 16, MM::zero());
  __syncthreads();
  for (int idx = tid; idx < 2 * NPV * 4 + 2 * KC; idx += NT) {
    if (idx < 2 * NPV * 4) {
      Chunk ones;
#pragma unroll
      for (int e = 0; e < EPC; ++e) ones[e] = from_f32<T>(1.0f);
      const int bsel = idx / (NPV * 4), rem = idx % (NPV * 4);
      st16<Chunk>(sV + bsel * VT_BYTES + ((rem >> 2) * ND * 16 + p.d) * 64 + ((rem & 3) << 4), ones);
    } else {
      Chunk one0 = MM::zero();
      one0[0] = from_f32<T>(1.0f);
      const int r2 = idx - 2 * NPV * 4;
      const int bsel = r2 / KC, row = r2 % KC;
      st16<Chunk>(sK + bsel * KT_BYTES + (kf_pad * KC + row) * 64 + ((lg_pad ^ aswz(row)) << 4), one0);
    }
  }

#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int kf = 0; kf < NKF; ++kf)
#pragma unroll
      for (int e = 0; e < EPC; ++e) qf[u][kf][e] = from_f32<T>(to_f32(qf[u][kf][e]) * p.scale_log2e);
  __syncthreads();       // zero fill / constant rows complete before the live pieces land
  store_chunk(0, rkA, rvA);
  __syncthreads();
  for (int c = 0; c < nchunks; c += 2) {
    // even step: B holds chunk c+1, A receives chunk c+2
    if (!(DBG & 16)) load_chunk((c + 2) * KC, c + 3 < nchunks, rkA, rvA);
    compute(c * KC, 0);
    if (!(DBG & 16) && c + 1 < nchunks) store_chunk(1, rkB, rvB);
    if (!(DBG & 32)) __syncthreads();
    if (c + 1 >= nchunks) break;
    // odd step: A holds chunk c+2, B receives chunk c+3
    if (!(DBG & 16)) load_chunk((c + 3) * KC, c + 4 < nchunks, rkB, rvB);
    compute((c + 1) * KC, 1);
    if (!(DBG & 16) && c + 2 < nchunks) store_chunk(0, rkA, rvA);
    if (!(DBG & 32)) __syncthreads();
  }

  // ---- finish: the denominator sits in O^T row d = tile d/16, lane group (d%16)/4, element 0
  // (head_dim is a multiple of 8); normalise, store 4 consecutive channels per lane
  const int ltile = p.d >> 4, lsrc = li + 16 * ((p.d & 15) >> 2);
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    float lsel = 0.f;
#pragma unroll
    for (int td = 0; td < ND; ++td)
      if (td == ltile) lsel = oacc[u][td][0];
    const float l = __shfl(lsel, lsrc, 64);
    const float inv = 1.0f / l;
    const int qrow = q0 + 16 * u + li;
    if (qrow < p.Tq) {
      T* op = p.o + ((size_t)b * p.Tq + qrow) * p.ldo + h * p.d;
#pragma unroll
      for (int td = 0; td < ND; ++td) {
        const int dch = 16 * td + 4 * lg;
        if (dch + 3 < p.d)
          store4_out<T>(op + dch, oacc[u][td][0] * inv, oacc[u][td][1] * inv, oacc[u][td][2] * inv, oacc[u][td][3] * inv);
      }
    }
  }
}

template <typename T, int NW, bool RAGGED>
static bool attn_launch_nw(const AttnP<T>& p, int nd, int nkf, int grid, hipStream_t st) {
  if (nd == 1 && nkf == 1) k_attn<T, 1, 1, NW, RAGGED><<<grid, NW * 64, 0, st>>>(p);
  else if (nd == 2 && nkf == 1) {
    static const int dbg = getenv("AFLDM_ATTN_DBG") ? atoi(getenv("AFLDM_ATTN_DBG")) : 0;
    if (sizeof(T) == 2 && NW == 8 && !RAGGED && dbg) {
      switch (dbg) {
        case 1: k_attn<T, 2, 1, NW, RAGGED, 1><<<grid, NW * 64, 0, st>>>(p); break;
        case 8: k_attn<T, 2, 1, NW, RAGGED, 8><<<grid, NW * 64, 0, st>>>(p); break;
        case 9: k_attn<T, 2, 1, NW, RAGGED, 9><<<grid, NW * 64, 0, st>>>(p); break;
        case 16: k_attn<T, 2, 1, NW, RAGGED, 16><<<grid, NW * 64, 0, st>>>(p); break;
        case 48: k_attn<T, 2, 1, NW, RAGGED, 48><<<grid, NW * 64, 0, st>>>(p); break;
        case 57: k_attn<T, 2, 1, NW, RAGGED, 57><<<grid, NW * 64, 0, st>>>(p); break;
        case 63: k_attn<T, 2, 1, NW, RAGGED, 63><<<grid, NW * 64, 0, st>>>(p); break;
        default: k_attn<T, 2, 1, NW, RAGGED><<<grid, NW * 64, 0, st>>>(p); break;
      }
    } else {
      k_attn<T, 2, 1, NW, RAGGED><<<grid, NW * 64, 0, st>>>(p);
    }
  }
  else if (nd == 2 && nkf == 2) k_attn<T, 2, 2, NW, RAGGED><<<grid, NW * 64, 0, st>>>(p);
  else if (nd == 3 && nkf == 2) k_attn<T, 3, 2, NW, RAGGED><<<grid, NW * 64, 0, st>>>(p);
  else if (nd == 3 && nkf == 3) k_attn<T, 3, 3, NW, RAGGED><<<grid, NW * 64, 0, st>>>(p);
  else return false;
  return true;
}

template <typename T>
static int attn_launch(const void* q, int ldq, const void* k, int ldk, const void* vt, void* o, int ldo, int B, int Bk,
                       int heads, int Tq, int Tk, int d, float scale, hipStream_t st) {
  AttnP<T> p;
  p.q = (const T*)q; p.k = (const T*)k; p.vt = (const T*)vt; p.o = (T*)o;
  p.ldq = ldq; p.ldk = ldk; p.ldo = ldo;
  p.B = B; p.Bk = Bk; p.heads = heads; p.Tq = Tq; p.Tk = Tk; p.d = d;
  p.scale_log2e = scale * 1.4426950408889634f;
  // 8 waves (256 queries share every staged chunk) once a (batch, head) has that many queries,
  // 4 waves below; a wave always owns 32 queries
  // (and 2 / 1 waves for the 8x8 / 4x4 planes: a 4-wave workgroup of the 4x4 level was one wave of 16 queries and
  //  three idle ones)
  static const bool small_ok = !(getenv("AFLDM_ATTN_NO_SMALL") && atoi(getenv("AFLDM_ATTN_NO_SMALL")) != 0);
  // only with >= 1024 workgroups: with few of them (small batch) four waves stage the K / V chunks faster
  // (same box, ms/step: batch 64 5.499 -> 5.473; batch 8 2.543 -> 2.562, batch 1 2.295 -> 2.308 without this limit)
  const bool many = (long long)B * heads >= 1024;
  const int waves = Tq >= 256 ? 8 : (Tq > 64 || !small_ok || !many) ? 4 : Tq > 32 ? 2 : 1;
  p.qblocks = (Tq + 32 * waves - 1) / (32 * waves);
  const int grid = B * heads * p.qblocks;
  constexpr int KPF = Mma<T>::KPF;
  const int nkf = d / KPF + 1, nd = d / 16 + 1;    // + 1: room for the -m_run channel / the row of ones
  const bool ragged = Tk % KC != 0;
  const bool ok = waves == 8 ? (ragged ? attn_launch_nw<T, 8, true>(p, nd, nkf, grid, st) : attn_launch_nw<T, 8, false>(p, nd, nkf, grid, st))
                  : waves == 4 ? (ragged ? attn_launch_nw<T, 4, true>(p, nd, nkf, grid, st) : attn_launch_nw<T, 4, false>(p, nd, nkf, grid, st))
                  : waves == 2 ? (ragged ? attn_launch_nw<T, 2, true>(p, nd, nkf, grid, st) : attn_launch_nw<T, 2, false>(p, nd, nkf, grid, st))
                               : (ragged ? attn_launch_nw<T, 1, true>(p, nd, nkf, grid, st) : attn_launch_nw<T, 1, false>(p, nd, nkf, grid, st));
  if (!ok) {
    set_error("afldm_attention: unsupported head_dim %d", d);
    return AFLDM_ESHAPE;
  }
  return check_launch("afldm_attention");
}

}  // namespace afldm

using namespace afldm;

extern "C" int afldm_attention(const void* q, int ldq, const void* k, int ldk, const void* vt, void* o, int ldo, int B,
                               int Bk, int heads, int Tq, int Tk, int d, float scale, int dtype,
                               afldm_stream_t stream) {
  AFLDM_REQUIRE(q && k && vt && o, AFLDM_ENULL, "afldm_attention: NULL pointer");
  AFLDM_REQUIRE(B > 0 && Bk > 0 && B % Bk == 0 && heads > 0 && Tq > 0 && Tk > 0, AFLDM_ESHAPE,
                "afldm_attention: bad shape B=%d Bk=%d heads=%d Tq=%d Tk=%d", B, Bk, heads, Tq, Tk);
  AFLDM_REQUIRE(d >= 8 && d <= 32 && d % 8 == 0, AFLDM_ESHAPE, "afldm_attention: head_dim %d must be 8, 16, 24 or 32", d);
  AFLDM_REQUIRE(Tk % 4 == 0 && (Tk % 8 == 0 || dtype == AFLDM_F32 || Tk < 8), AFLDM_ESHAPE,
                "afldm_attention: Tk=%d must be a multiple of 4 (fp32) / 8 (bf16), or < 8", Tk);
  AFLDM_REQUIRE(ldq >= heads * d && ldk >= heads * d && ldo >= heads * d && ldq % 8 == 0 && ldk % 8 == 0 && ldo % 8 == 0,
                AFLDM_ESHAPE, "afldm_attention: leading dims (%d,%d,%d) must be >= heads*d and multiples of 8", ldq, ldk, ldo);
  AFLDM_REQUIRE(aligned16(q) && aligned16(k) && aligned16(vt) && aligned16(o), AFLDM_EALIGN,
                "afldm_attention: pointers must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == AFLDM_F32) return attn_launch<float>(q, ldq, k, ldk, vt, o, ldo, B, Bk, heads, Tq, Tk, d, scale, st);
  if (dtype == AFLDM_BF16) return attn_launch<bf16>(q, ldq, k, ldk, vt, o, ldo, B, Bk, heads, Tq, Tk, d, scale, st);
  set_error("afldm_attention: unknown dtype %d", dtype);
  return AFLDM_EDTYPE;
}
