// attn.hip — scaled-dot-product attention (softmax(q k^T * scale) v) on MFMA, flash-style
// (online softmax, S never materialised).  head_dim is small here (24; 16 in the tiny test
// config) and T <= 1024, so K (48 KB per (b, head) at T = 1024) and V^T stay L2-resident and
// the fragments are read straight from global memory (guide §5 common mistake 7: do not
// LDS-stage data that cache-fits).
//
// One wave owns 16 queries.  It computes S^T = K Q^T (A = K rows, B = Q rows), so that lane
// (j = query, g) holds the scores of ITS query for keys {4g + r}: the softmax reduction is 8
// local values + 2 cross-lane-group shuffles, and P^T is already laid out as the B operand of
// the second MFMA, O^T = V^T P^T (A = V^T rows straight from the channel-major vt tensor that
// afldm_conv2d(out_mode = 1) wrote).  O^T leaves 4 consecutive head channels of one query per
// lane -> vector stores into the token-major output.
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
  int qblocks;  // query blocks (of 16 * waves) per (b, head)
};

template <typename T, int ND /* 16-wide tiles of head_dim for O */, int NKF /* K-chunk pairs for QK */>
__global__ void __launch_bounds__(256) k_attn(AttnP<T> p) {
  typedef Mma<T> MM;
  typedef typename MM::Chunk Chunk;
  constexpr int EPC = MM::EPC, KPF = MM::KPF;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int waves = blockDim.x >> 6;

  int bid = blockIdx.x;
  const int qb = bid % p.qblocks;
  bid /= p.qblocks;
  const int h = bid % p.heads;
  const int b = bid / p.heads;
  const int kb = b / (p.B / p.Bk);
  const int q0 = (qb * waves + wave) * 16;
  if (q0 >= p.Tq) return;  // no block-level barriers below

  const int C = p.heads * p.d;
  const int qrow = q0 + li;
  const bool qok = qrow < p.Tq;
  const T* qptr = p.q + ((size_t)b * p.Tq + (qok ? qrow : 0)) * p.ldq + h * p.d;
  Chunk qf[NKF];
#pragma unroll
  for (int kf = 0; kf < NKF; ++kf) {
    const int e0 = kf * KPF + lg * EPC;
    qf[kf] = (qok && e0 + EPC <= p.d) ? ld16<Chunk>(qptr + e0) : MM::zero();
  }

  f32x4 oacc[ND];
#pragma unroll
  for (int t = 0; t < ND; ++t) oacc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = -1e30f, l_run = 0.f;

  const T* kbase = p.k + (size_t)kb * p.Tk * p.ldk + h * p.d;
  const T* vbase = p.vt + ((size_t)kb * C + h * p.d) * p.Tk;

  for (int key0 = 0; key0 < p.Tk; key0 += 32) {
    // ---- S^T for 32 keys: two 16-key tiles
    f32x4 s[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      const int key = key0 + 16 * t + li;
      const bool kok = key < p.Tk;
      const T* kp = kbase + (size_t)(kok ? key : 0) * p.ldk;
#pragma unroll
      for (int kf = 0; kf < NKF; ++kf) {
        const int e0 = kf * KPF + lg * EPC;
        Chunk kfz = (kok && e0 + EPC <= p.d) ? ld16<Chunk>(kp + e0) : MM::zero();
        MM::mma(s[t], kfz, qf[kf]);
      }
    }
    // ---- scale, mask, online softmax for query li (lane-group g holds keys 16t + 4g + r)
    float mloc = -1e30f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = key0 + 16 * t + 4 * lg + r;
        float v = s[t][r] * p.scale_log2e;
        v = key < p.Tk ? v : -1e30f;
        s[t][r] = v;
        mloc = fmaxf(mloc, v);
      }
    mloc = fmaxf(mloc, __shfl_xor(mloc, 16, 64));
    mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
    const float m_new = fmaxf(m_run, mloc);
    const float alpha = exp2f(m_run - m_new);
    m_run = m_new;
    float psum = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float pv = exp2f(s[t][r] - m_new);
        s[t][r] = pv;
        psum += pv;
      }
    l_run = l_run * alpha + psum;  // lane-partial; lane groups are combined at the end
#pragma unroll
    for (int t = 0; t < ND; ++t) oacc[t] *= alpha;

    // ---- O^T += V^T P^T
#pragma unroll
    for (int td = 0; td < ND; ++td) {
      const int drow = 16 * td + li;
      const bool dok = drow < p.d;
      const T* vp = vbase + (size_t)(dok ? drow : 0) * p.Tk + key0 + 4 * lg;
      const bool v0ok = dok && (key0 + 4 * lg) < p.Tk;
      const bool v1ok = dok && (key0 + 16 + 4 * lg) < p.Tk;
      if constexpr (sizeof(T) == 2) {
        // bf16: one K = 32 chunk pair; k-set(g) = {4g..4g+3} U {16+4g..16+4g+3}
        bf16x4 va = v0ok ? *reinterpret_cast<const bf16x4*>(vp) : bf16x4{(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
        bf16x4 vb = v1ok ? *reinterpret_cast<const bf16x4*>(vp + 16) : bf16x4{(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
        bf16x8 a, pb;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          a[r] = va[r];
          a[4 + r] = vb[r];
          pb[r] = (bf16)s[0][r];
          pb[4 + r] = (bf16)s[1][r];
        }
        MM::mma(oacc[td], a, pb);
      } else {
        // fp32: two K = 16 chunk pairs, one per 16-key tile; k-set(g) = {4g..4g+3}
        f32x4 va = v0ok ? *reinterpret_cast<const f32x4*>(vp) : f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 vb = v1ok ? *reinterpret_cast<const f32x4*>(vp + 16) : f32x4{0.f, 0.f, 0.f, 0.f};
        MM::mma(oacc[td], va, s[0]);
        MM::mma(oacc[td], vb, s[1]);
      }
    }
  }

  // ---- finish: combine the lane-partial row sums, normalise, store 4 consecutive channels
  l_run += __shfl_xor(l_run, 16, 64);
  l_run += __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_run;
  if (qok) {
    T* op = p.o + ((size_t)b * p.Tq + qrow) * p.ldo + h * p.d;
#pragma unroll
    for (int td = 0; td < ND; ++td) {
      const int dch = 16 * td + 4 * lg;
      if (dch + 3 < p.d)
        store4<T>(op + dch, oacc[td][0] * inv, oacc[td][1] * inv, oacc[td][2] * inv, oacc[td][3] * inv);
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
  int waves = (Tq + 15) / 16;
  if (waves > 4) waves = 4;
  p.qblocks = (Tq + 16 * waves - 1) / (16 * waves);
  const int grid = B * heads * p.qblocks;
  constexpr int KPF = Mma<T>::KPF;
  const int nkf = (d + KPF - 1) / KPF, nd = (d + 15) / 16;
  if (nd == 1 && nkf == 1) k_attn<T, 1, 1><<<grid, waves * 64, 0, st>>>(p);
  else if (nd == 2 && nkf == 1) k_attn<T, 2, 1><<<grid, waves * 64, 0, st>>>(p);
  else if (nd == 2 && nkf == 2) k_attn<T, 2, 2><<<grid, waves * 64, 0, st>>>(p);
  else if (nd == 4 && nkf == 2) k_attn<T, 4, 2><<<grid, waves * 64, 0, st>>>(p);
  else if (nd == 4 && nkf == 4) k_attn<T, 4, 4><<<grid, waves * 64, 0, st>>>(p);
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
  AFLDM_REQUIRE(d >= 8 && d <= 64 && d % 8 == 0, AFLDM_ESHAPE, "afldm_attention: head_dim %d must be a multiple of 8 in [8,64]", d);
  AFLDM_REQUIRE(Tk % 4 == 0, AFLDM_ESHAPE, "afldm_attention: Tk=%d must be a multiple of 4", Tk);
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
