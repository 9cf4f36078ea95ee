// attn.hip — scaled-dot-product attention (softmax(q k^T * scale) v) on MFMA, flash-style
// (online softmax, S never materialised), tuned for the UNet's shape: head_dim 24, T <= 1024.
// With d = 24 the kernel is softmax(VALU)-bound, not MFMA-bound (T^2 exponentials per head vs
// 4 T^2 d flops), so the structure minimises VALU work per score and LDS/global traffic per MFMA:
//
//  * one wave owns 32 queries (two 16-query B fragments sharing every K / V^T A fragment);
//    a workgroup = up to 4 waves = 128 queries of one (batch, head);
//  * S^T = K Q^T (A = K rows, B = Q rows): lane (j = query, g) holds the scores of ITS query for
//    keys {4g + r} of each 16-key tile -> the row max / row sum are 16 local values + 2 shuffles,
//    and P^T is already the B operand of the second MFMA, O^T = V^T P^T (A = V^T rows);
//  * scale * log2(e) is folded into Q once, so a score costs max + sub + exp2 + add (+ cvt);
//  * K and V^T are staged per 64-key chunk through LDS (double-buffered, global -> VGPR -> LDS,
//    next chunk's loads in flight during the MFMAs) in fragment order: every A fragment is one
//    conflict-free 16-byte LDS read (64-B rows, chunk index XOR-swizzled by the row);
//  * V arrives channel-major (vt[b][c][t], written by afldm_conv2d out_mode 1), so V^T rows are
//    contiguous key runs and O^T leaves 4 consecutive head channels of one query per lane.
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

constexpr int KC = 64;  // keys per chunk

template <typename T, int ND /* 16-wide tiles of head_dim */, int NKF /* chunk pairs covering head_dim in QK^T */>
__global__ void __launch_bounds__(256) k_attn(AttnP<T> p) {
  typedef Mma<T> MM;
  typedef typename MM::Chunk Chunk;
  constexpr int EPC = MM::EPC, KPF = MM::KPF;
  constexpr bool BF = sizeof(T) == 2;
  constexpr int NPV = KC / KPF;                  // chunk pairs covering the 64 keys in P V (2 bf16 / 4 fp32)
  constexpr int KT_BYTES = NKF * KC * 64;        // K tile:  [kf][key][64 B]
  constexpr int VT_BYTES = NPV * ND * 16 * 64;   // V^T tile: [pv][d row][64 B] (fragment-ordered keys)
  __shared__ __attribute__((aligned(16))) char smem[2 * (KT_BYTES + VT_BYTES)];
  char* sK = smem;
  char* sV = smem + 2 * KT_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  constexpr int nthreads = 256, waves = 4;   // always 4 waves: idle ones still help staging

  int bid = blockIdx.x;
  const int qb = bid % p.qblocks;
  bid /= p.qblocks;
  const int h = bid % p.heads;
  const int b = bid / p.heads;
  const int kb = b / (p.B / p.Bk);
  const int q0 = (qb * waves + wave) * 32;
  const int C = p.heads * p.d;

  // ---- Q fragments (pre-scaled), two 16-query tiles
  Chunk qf[2][NKF];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int qrow = q0 + 16 * u + li;
    const bool qok = qrow < p.Tq;
    const T* qptr = p.q + ((size_t)b * p.Tq + (qok ? qrow : 0)) * p.ldq + h * p.d;
#pragma unroll
    for (int kf = 0; kf < NKF; ++kf) {
      const int e0 = kf * KPF + lg * EPC;
      Chunk c = (qok && e0 + EPC <= p.d) ? ld16<Chunk>(qptr + e0) : MM::zero();
#pragma unroll
      for (int e = 0; e < EPC; ++e) c[e] = from_f32<T>(to_f32(c[e]) * p.scale_log2e);
      qf[u][kf] = c;
    }
  }

  // ---- staging assignment: 16-byte pieces of the K chunk and of the V^T chunk
  //   K : NKF * KC rows x 4 pieces;   V^T: ND*16 d-rows x (KC*sizeof(T)/16) pieces
  constexpr int KPIECES = NKF * KC * 4;
  constexpr int VPR = KC * (int)sizeof(T) / 16;   // 16-B pieces per V^T row per chunk
  constexpr int VPIECES = ND * 16 * VPR;
  constexpr int MAXP = (KPIECES + 255) / 256 > (VPIECES + 255) / 256 ? (KPIECES + 255) / 256 : (VPIECES + 255) / 256;
  const T* kbase = p.k + (size_t)kb * p.Tk * p.ldk + h * p.d;
  const T* vbase = p.vt + ((size_t)kb * C + h * p.d) * p.Tk;
  Chunk rk[MAXP], rv[MAXP];

  auto load_chunk = [&](int key0) {
#pragma unroll
    for (int i = 0; i < MAXP; ++i) {
      const int id = tid + i * nthreads;
      rk[i] = MM::zero();
      rv[i] = MM::zero();
      if (id < KPIECES) {
        const int piece = id & 3, row = (id >> 2) % KC, kf = (id >> 2) / KC;
        const int key = key0 + row, e0 = kf * KPF + piece * EPC;
        if (key < p.Tk && e0 + EPC <= p.d) rk[i] = ld16<Chunk>(kbase + (size_t)key * p.ldk + e0);
      }
      if (id < VPIECES) {
        const int piece = id % VPR, drow = id / VPR;
        const int key = key0 + piece * EPC;
        if (drow < p.d && key < p.Tk) {
          const T* vsrc = vbase + (size_t)drow * p.Tk + key;
          if (key + EPC <= p.Tk) {
            rv[i] = ld16<Chunk>(vsrc);
          } else {  // Tk < one piece (the 2x2 level, Tk = 4): element-wise, zero tail
#pragma unroll
            for (int e = 0; e < EPC; ++e)
              if (key + e < p.Tk) rv[i][e] = vsrc[e];
          }
        }
      }
    }
  };
  auto store_chunk = [&](int buf) {
#pragma unroll
    for (int i = 0; i < MAXP; ++i) {
      const int id = tid + i * nthreads;
      if (id < KPIECES) {
        const int piece = id & 3, row = (id >> 2) % KC, kf = (id >> 2) / KC;
        st16<Chunk>(sK + buf * KT_BYTES + (kf * KC + row) * 64 + ((piece ^ aswz(row)) << 4), rk[i]);
      }
      if (id < VPIECES) {
        const int piece = id % VPR, drow = id / VPR;
        char* base = sV + buf * VT_BYTES;
        if constexpr (BF) {
          // 8 consecutive keys 8j..8j+7 of a 32-key half: keys 8j..8j+3 -> group g = 2(j&1), keys
          // 8j+4..8j+7 -> g = 2(j&1)+1; element offset 0 for j < 2 (keys < 16), 4 otherwise.
          const int half = piece >> 2, j = piece & 3;
          const int g0 = 2 * (j & 1), eoff = (j >> 1) * 4;
          bf16x4 lo, hi;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            lo[e] = rv[i][e];
            hi[e] = rv[i][4 + e];
          }
          char* rowp = base + (half * ND * 16 + drow) * 64;
          *reinterpret_cast<bf16x4*>(rowp + (((g0) ^ aswz(drow)) << 4) + eoff * 2) = lo;
          *reinterpret_cast<bf16x4*>(rowp + (((g0 + 1) ^ aswz(drow)) << 4) + eoff * 2) = hi;
        } else {
          // fp32: 4 consecutive keys = chunk g of 16-key tile t
          const int t = piece >> 2, g = piece & 3;
          st16<Chunk>(base + (t * ND * 16 + drow) * 64 + ((g ^ aswz(drow)) << 4), rv[i]);
        }
      }
    }
  };

  f32x4 oacc[2][ND];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int t = 0; t < ND; ++t) oacc[u][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run[2] = {-1e30f, -1e30f}, l_run[2] = {0.f, 0.f};

  load_chunk(0);
  store_chunk(0);
  __syncthreads();
  int buf = 0;
  for (int key0 = 0; key0 < p.Tk; key0 += KC, buf ^= 1) {
    const bool more = key0 + KC < p.Tk;
    if (more) load_chunk(key0 + KC);

    // ---- S^T: 4 key tiles x 2 query tiles
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
        for (int kf = 0; kf < NKF; ++kf) MM::mma(s[u][t], kfz[kf], qf[u][kf]);
      }
    }
    if (key0 + KC > p.Tk) {  // ragged last chunk: mask keys >= Tk (wave-uniform branch)
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (key0 + 16 * t + 4 * lg + r >= p.Tk) s[u][t][r] = -1e30f;
    }
    // ---- online softmax (scores are already in log2 units)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      float mloc = s[u][0][0];
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) mloc = fmaxf(mloc, s[u][t][r]);
      mloc = fmaxf(mloc, __shfl_xor(mloc, 16, 64));
      mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
      const float m_new = fmaxf(m_run[u], mloc);
      const float alpha = __builtin_amdgcn_exp2f(m_run[u] - m_new);
      m_run[u] = m_new;
      float psum = 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pv = __builtin_amdgcn_exp2f(s[u][t][r] - m_new);
          s[u][t][r] = pv;
          psum += pv;
        }
      l_run[u] = l_run[u] * alpha + psum;  // lane-partial; the 4 lane groups are combined at the end
#pragma unroll
      for (int t = 0; t < ND; ++t) oacc[u][t] *= alpha;
    }
    // ---- O^T += V^T P^T
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
        for (int u = 0; u < 2; ++u) MM::mma(oacc[u][td], va, pb[u]);
      }
    }
    if (more) store_chunk(buf ^ 1);
    __syncthreads();
  }

  // ---- finish: combine lane-partial row sums, normalise, store 4 consecutive channels per lane
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    float l = l_run[u];
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
    const int qrow = q0 + 16 * u + li;
    if (qrow < p.Tq) {
      T* op = p.o + ((size_t)b * p.Tq + qrow) * p.ldo + h * p.d;
#pragma unroll
      for (int td = 0; td < ND; ++td) {
        const int dch = 16 * td + 4 * lg;
        if (dch + 3 < p.d)
          store4<T>(op + dch, oacc[u][td][0] * inv, oacc[u][td][1] * inv, oacc[u][td][2] * inv, oacc[u][td][3] * inv);
      }
    }
  }
}

template <typename T>
static int attn_launch(const void* q, int ldq, const void* k, int ldk, const void* vt, void* o, int ldo, int B, int Bk,
                       int heads, int Tq, int Tk, int d, float scale, hipStream_t st) {
  AttnP<T> p;
  p.q = (const T*)q; p.k = (const T*)k; p.vt = (const T*)vt; p.o = (T*)o;
  p.ldq = ldq; p.ldk = ldk; p.ldo = ldo;
  p.B = B; p.Bk = Bk; p.heads = heads; p.Tq = Tq; p.Tk = Tk; p.d = d;
  p.scale_log2e = scale * 1.4426950408889634f;
  const int waves = 4;
  p.qblocks = (Tq + 32 * waves - 1) / (32 * waves);
  const int grid = B * heads * p.qblocks;
  constexpr int KPF = Mma<T>::KPF;
  const int nkf = (d + KPF - 1) / KPF, nd = (d + 15) / 16;
  if (nd == 1 && nkf == 1) k_attn<T, 1, 1><<<grid, waves * 64, 0, st>>>(p);
  else if (nd == 2 && nkf == 1) k_attn<T, 2, 1><<<grid, waves * 64, 0, st>>>(p);
  else if (nd == 2 && nkf == 2) k_attn<T, 2, 2><<<grid, waves * 64, 0, st>>>(p);
  else {
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
