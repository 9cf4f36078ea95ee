// conv.hip — 3x3 / 1x1 convolution and nn.Linear as implicit GEMM on MFMA (gfx950).
//
//   Y[m, n] = sum_k  Xcol[m, k] * W[n, k]       m = (b, oh, ow),  n = cout,  k = (kh, kw, ci)
//
// NHWC activations make k contiguous for both operands, so every MFMA operand is one 16-byte
// chunk of a K-contiguous row (see Mma<T> in common.hpp).  The weight rows are the MFMA "A"
// operand (i = cout) and the pixels the "B" operand (j = pixel): the accumulator of lane
// (j, g) then holds 4 CONSECUTIVE couts of ONE pixel, so the epilogue reads bias / time-embedding
// / residual and writes the NHWC output as 8-byte (bf16) or 16-byte (fp32) vectors.
//
// Tile: BM pixels x BN couts per workgroup, WGM x WGN waves, K step = KCH 64-byte row pieces
// (KCH*32 bf16 or KCH*16 fp32 channels of one (kh, kw) tap).  Zero padding, the M tail and the
// Cout tail are handled by zero-filling the staged chunk.  Global -> VGPR -> LDS staging with the
// next tile's loads in flight during the MFMAs (guide T14), two LDS buffers, one barrier per
// K step.  LDS rows are 64 B with the 16-B chunk index XOR-swizzled by the row so that the
// fragment read (16 rows x 4 chunks per ds_read_b128 wave access) is bank-conflict free.
// Optional split-K (grid.z) writes fp32 slabs that k_splitk_reduce folds together with the
// epilogue terms — used for the 2x2 / 4x4 levels where M = B*HW is too small to fill 256 CUs.
#include "common.hpp"

namespace afldm {

struct ConvP {
  const void* x1;
  const void* x2;
  const void* w;
  const float* bias;
  const void* temb;
  const void* residual;
  void* y;
  float* ws;
  int C1, C2, B, H, W, Cout, KS;
  int temb_stride, res_ld, y_ld, out_mode;
  int M;        // B*H*W
  int ksteps;   // total K steps = KS*KS * (C1+C2)/(KCH*EPR)
  int splitk;   // grid.z
  int tiles_n;
};

__device__ __forceinline__ int swz(int row) { return (4 - ((row >> 2) & 3)) & 3; }

template <typename T>
__device__ __forceinline__ void epilogue_store(const ConvP& p, int m, int n, f32x4 v) {
  // adds bias / temb / residual for couts n..n+3 of pixel m and stores them
  const T* temb = (const T*)p.temb;
  const T* res = (const T*)p.residual;
  T* y = (T*)p.y;
  const int HW = p.H * p.W;
  const int b = m / HW;
  if (n + 3 < p.Cout) {
    if (p.bias) {
      f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + n);
      v += bv;
    }
    if (temb) {
      float a0, a1, a2, a3;
      load4<T>(temb + (size_t)b * p.temb_stride + n, a0, a1, a2, a3);
      v[0] += a0; v[1] += a1; v[2] += a2; v[3] += a3;
    }
    if (res) {
      float a0, a1, a2, a3;
      load4<T>(res + (size_t)m * p.res_ld + n, a0, a1, a2, a3);
      v[0] += a0; v[1] += a1; v[2] += a2; v[3] += a3;
    }
    if (p.out_mode == 0) {
      store4<T>(y + (size_t)m * p.y_ld + n, v[0], v[1], v[2], v[3]);
    } else {
      const int pix = m - b * HW;
#pragma unroll
      for (int r = 0; r < 4; ++r) y[((size_t)b * p.Cout + n + r) * HW + pix] = from_f32<T>(v[r]);
    }
  } else {
    const int pix = m - b * HW;
    for (int r = 0; r < 4; ++r) {
      if (n + r >= p.Cout) break;
      float s = v[r];
      if (p.bias) s += p.bias[n + r];
      if (temb) s += to_f32(temb[(size_t)b * p.temb_stride + n + r]);
      if (res) s += to_f32(res[(size_t)m * p.res_ld + n + r]);
      if (p.out_mode == 0) y[(size_t)m * p.y_ld + n + r] = from_f32<T>(s);
      else y[((size_t)b * p.Cout + n + r) * HW + pix] = from_f32<T>(s);
    }
  }
}

template <typename T, int BM, int BN, int WGM, int WGN, int KCH>
__global__ void __launch_bounds__(WGM* WGN * 64) k_igemm(ConvP p) {
  typedef Mma<T> MM;
  typedef typename MM::Chunk Chunk;
  constexpr int NT = WGM * WGN * 64;
  constexpr int EPC = MM::EPC;       // elements per 16-B chunk
  constexpr int EPR = 4 * EPC;       // elements per 64-B row piece
  constexpr int KSTEP = KCH * EPR;   // channels consumed per K step
  constexpr int WMS = BM / WGM, WNS = BN / WGN;
  constexpr int TM = WMS / 16, TN = WNS / 16;
  constexpr int RPT = NT / 4;        // rows staged per pass (4 chunk-lanes per row)
  constexpr int RX = BM / RPT, RW = BN / RPT;
  static_assert(BM % RPT == 0 && BN % RPT == 0, "tile rows must be a multiple of NT/4");
  static_assert(WMS % 16 == 0 && WNS % 16 == 0, "wave tile must be a multiple of 16");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  // layout: [stage][kc][row][64 B]   X rows first, then W rows
  constexpr int X_STAGE = KCH * BM * 64, W_STAGE = KCH * BN * 64;
  char* sX = smem;
  char* sW = smem + 2 * X_STAGE;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  const int li = lane & 15, lg = lane >> 4;

  const int ntiles = gridDim.x;
  const int tile = xcd_remap(blockIdx.x, ntiles);
  const int tile_m = tile / p.tiles_n, tile_n = tile % p.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int Ct = p.C1 + p.C2;
  const int cblocks = Ct / KSTEP;
  const int HW = p.H * p.W;
  const int pad = p.KS >> 1;

  // K range of this split
  const int ks = blockIdx.z;
  const int kbeg = (int)(((long long)p.ksteps * ks) / p.splitk);
  const int kend = (int)(((long long)p.ksteps * (ks + 1)) / p.splitk);

  // ---- per-thread staging coordinates
  const int srow = tid >> 2, schunk = tid & 3;
  int xb[RX], xoh[RX], xow[RX];
  bool xv[RX];
#pragma unroll
  for (int r = 0; r < RX; ++r) {
    int m = m0 + r * RPT + srow;
    xv[r] = m < p.M;
    int mm = xv[r] ? m : 0;
    int b = mm / HW, pix = mm - b * HW;
    xb[r] = b;
    xoh[r] = pix / p.W;
    xow[r] = pix - xoh[r] * p.W;
  }
  Chunk rx[KCH][RX], rw[KCH][RW];

  auto load_stage = [&](int kt) {
    const int tap = kt / cblocks;
    const int ci0 = (kt - tap * cblocks) * KSTEP;
    const int kh = tap / p.KS, kw = tap - kh * p.KS;
    const bool second = ci0 >= p.C1;
    const T* xsrc = second ? (const T*)p.x2 : (const T*)p.x1;
    const int Cs = second ? p.C2 : p.C1;
    const int cs0 = second ? ci0 - p.C1 : ci0;
#pragma unroll
    for (int r = 0; r < RX; ++r) {
      const int ih = xoh[r] + kh - pad, iw = xow[r] + kw - pad;
      const bool ok = xv[r] && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
      const T* src = xsrc + ((size_t)(xb[r] * p.H + (ok ? ih : 0)) * p.W + (ok ? iw : 0)) * Cs + cs0 + schunk * EPC;
#pragma unroll
      for (int kc = 0; kc < KCH; ++kc) rx[kc][r] = ok ? ld16<Chunk>(src + kc * EPR) : MM::zero();
    }
    const T* wbase = (const T*)p.w;
#pragma unroll
    for (int r = 0; r < RW; ++r) {
      const int n = n0 + r * RPT + srow;
      const bool ok = n < p.Cout;
      const T* src = wbase + ((size_t)(ok ? n : 0) * p.KS * p.KS + tap) * Ct + ci0 + schunk * EPC;
#pragma unroll
      for (int kc = 0; kc < KCH; ++kc) rw[kc][r] = ok ? ld16<Chunk>(src + kc * EPR) : MM::zero();
    }
  };
  auto store_stage = [&](int buf) {
#pragma unroll
    for (int kc = 0; kc < KCH; ++kc) {
#pragma unroll
      for (int r = 0; r < RX; ++r) {
        const int row = r * RPT + srow;
        st16<Chunk>(sX + buf * X_STAGE + (kc * BM + row) * 64 + ((schunk ^ swz(row)) << 4), rx[kc][r]);
      }
#pragma unroll
      for (int r = 0; r < RW; ++r) {
        const int row = r * RPT + srow;
        st16<Chunk>(sW + buf * W_STAGE + (kc * BN + row) * 64 + ((schunk ^ swz(row)) << 4), rw[kc][r]);
      }
    }
  };

  f32x4 acc[TN][TM];
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (kbeg < kend) {
    load_stage(kbeg);
    store_stage(0);
    __syncthreads();
    for (int kt = kbeg; kt < kend; ++kt) {
      const int buf = (kt - kbeg) & 1;
      const bool more = kt + 1 < kend;
      if (more) load_stage(kt + 1);
#pragma unroll
      for (int kc = 0; kc < KCH; ++kc) {
        Chunk a[TN], b[TM];
#pragma unroll
        for (int t = 0; t < TN; ++t) {
          const int row = wn * WNS + t * 16 + li;
          a[t] = ld16<Chunk>(sW + buf * W_STAGE + (kc * BN + row) * 64 + ((lg ^ swz(row)) << 4));
        }
#pragma unroll
        for (int t = 0; t < TM; ++t) {
          const int row = wm * WMS + t * 16 + li;
          b[t] = ld16<Chunk>(sX + buf * X_STAGE + (kc * BM + row) * 64 + ((lg ^ swz(row)) << 4));
        }
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
          for (int tm = 0; tm < TM; ++tm) MM::mma(acc[tn][tm], a[tn], b[tm]);
      }
      if (more) store_stage(buf ^ 1);
      __syncthreads();
    }
  }

  // ---- epilogue: lane (j = li, g = lg) holds couts n..n+3 (n = .. + 4*lg) of pixel m (.. + li)
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      const int m = m0 + wm * WMS + tm * 16 + li;
      const int n = n0 + wn * WNS + tn * 16 + 4 * lg;
      if (m >= p.M || n >= p.Cout) continue;
      if (p.splitk > 1) {
        float* dst = p.ws + ((size_t)ks * p.M + m) * p.Cout + n;
        if (n + 3 < p.Cout) *reinterpret_cast<f32x4*>(dst) = acc[tn][tm];
        else
          for (int r = 0; r < 4 && n + r < p.Cout; ++r) dst[r] = acc[tn][tm][r];
      } else {
        epilogue_store<T>(p, m, n, acc[tn][tm]);
      }
    }
  }
}

template <typename T>
__global__ void __launch_bounds__(256) k_splitk_reduce(ConvP p) {
  const int nq = (p.Cout + 3) / 4;
  const size_t total = (size_t)p.M * nq;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int m = (int)(i / nq);
    const int n = (int)(i - (size_t)m * nq) * 4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < p.splitk; ++s) {
      const float* src = p.ws + ((size_t)s * p.M + m) * p.Cout + n;
      if (n + 3 < p.Cout) v += *reinterpret_cast<const f32x4*>(src);
      else
        for (int r = 0; r < 4 && n + r < p.Cout; ++r) v[r] += src[r];
    }
    epilogue_store<T>(p, m, n, v);
  }
}

// ----------------------------------------------------------------------------- small direct forms
// conv_in (Cin = 4): one thread per (pixel, cout); K = KS*KS*Cin is tiny.
template <typename T>
__global__ void __launch_bounds__(256) k_conv_small_cin(ConvP p) {
  const int Ct = p.C1;
  const int HW = p.H * p.W, pad = p.KS >> 1;
  const size_t total = (size_t)p.M * p.Cout;
  const T* x = (const T*)p.x1;
  const T* w = (const T*)p.w;
  T* y = (T*)p.y;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int n = (int)(i % p.Cout);
    const int m = (int)(i / p.Cout);
    const int b = m / HW, pix = m - b * HW, oh = pix / p.W, ow = pix - oh * p.W;
    float acc = p.bias ? p.bias[n] : 0.f;
    for (int kh = 0; kh < p.KS; ++kh) {
      const int ih = oh + kh - pad;
      if (ih < 0 || ih >= p.H) continue;
      for (int kw = 0; kw < p.KS; ++kw) {
        const int iw = ow + kw - pad;
        if (iw < 0 || iw >= p.W) continue;
        const T* xs = x + ((size_t)(b * p.H + ih) * p.W + iw) * Ct;
        const T* wsrc = w + ((size_t)n * p.KS * p.KS + kh * p.KS + kw) * Ct;
        for (int c = 0; c < Ct; ++c) acc = fmaf(to_f32(xs[c]), to_f32(wsrc[c]), acc);
      }
    }
    if (p.temb) acc += to_f32(((const T*)p.temb)[(size_t)b * p.temb_stride + n]);
    if (p.residual) acc += to_f32(((const T*)p.residual)[(size_t)m * p.res_ld + n]);
    if (p.out_mode == 0) y[(size_t)m * p.y_ld + n] = from_f32<T>(acc);
    else y[((size_t)b * p.Cout + n) * HW + pix] = from_f32<T>(acc);
  }
}

// conv_out (Cout = 4): one wave per pixel, lanes stride the channels, butterfly reduce.
template <typename T, int NOUT>
__global__ void __launch_bounds__(256) k_conv_small_cout(ConvP p) {
  const int Ct = p.C1;
  const int HW = p.H * p.W, pad = p.KS >> 1;
  const int lane = threadIdx.x & 63;
  const int wave_global = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  const T* x = (const T*)p.x1;
  const T* w = (const T*)p.w;
  T* y = (T*)p.y;
  for (int m = wave_global; m < p.M; m += nwaves) {
    const int b = m / HW, pix = m - b * HW, oh = pix / p.W, ow = pix - oh * p.W;
    float acc[NOUT];
#pragma unroll
    for (int n = 0; n < NOUT; ++n) acc[n] = 0.f;
    for (int kh = 0; kh < p.KS; ++kh) {
      const int ih = oh + kh - pad;
      if (ih < 0 || ih >= p.H) continue;
      for (int kw = 0; kw < p.KS; ++kw) {
        const int iw = ow + kw - pad;
        if (iw < 0 || iw >= p.W) continue;
        const T* xs = x + ((size_t)(b * p.H + ih) * p.W + iw) * Ct;
        const int tap = kh * p.KS + kw;
        for (int c = lane; c < Ct; c += 64) {
          const float xv = to_f32(xs[c]);
#pragma unroll
          for (int n = 0; n < NOUT; ++n)
            if (n < p.Cout) acc[n] = fmaf(xv, to_f32(w[((size_t)n * p.KS * p.KS + tap) * Ct + c]), acc[n]);
        }
      }
    }
#pragma unroll
    for (int n = 0; n < NOUT; ++n)
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) acc[n] += __shfl_xor(acc[n], o, 64);
    if (lane == 0) {
      for (int n = 0; n < p.Cout; ++n) {
        float s = acc[n] + (p.bias ? p.bias[n] : 0.f);
        if (p.temb) s += to_f32(((const T*)p.temb)[(size_t)b * p.temb_stride + n]);
        if (p.residual) s += to_f32(((const T*)p.residual)[(size_t)m * p.res_ld + n]);
        if (p.out_mode == 0) y[(size_t)m * p.y_ld + n] = from_f32<T>(s);
        else y[((size_t)b * p.Cout + n) * HW + pix] = from_f32<T>(s);
      }
    }
  }
}

// ----------------------------------------------------------------------------- host dispatch
struct Plan {
  int kind;  // 0 igemm, 1 small_cin, 2 small_cout
  int cfg;   // igemm tile config
  int splitk;
};

constexpr int KCH_DEFAULT = 2;

template <typename T>
static int epr() { return 4 * Mma<T>::EPC; }

static Plan make_plan(const afldm_conv_args* a, int elems_per_row) {
  Plan pl{0, 0, 1};
  const int Ct = a->C1 + a->C2;
  const int kstep = KCH_DEFAULT * elems_per_row;
  const bool gemm_ok = (Ct % kstep == 0) && (a->C2 == 0 || a->C1 % kstep == 0) && a->Cout >= 16;
  if (!gemm_ok) {
    pl.kind = (a->Cout <= 8) ? 2 : 1;
    return pl;
  }
  const long long M = (long long)a->B * a->H * a->W;
  // tile choice: 128x128 when Cout is a multiple of 128 (or large), 128x64 otherwise; 64-pixel
  // tiles when M is small
  int bm = (M >= 128 * 96) ? 128 : 64;
  int bn = (a->Cout % 128 == 0) ? 128 : 64;
  pl.cfg = (bm == 128 ? 0 : 2) + (bn == 128 ? 0 : 1);
  // split-K when the grid cannot fill the chip
  const long long tiles = ((M + bm - 1) / bm) * ((a->Cout + bn - 1) / bn);
  const int ksteps = a->KS * a->KS * (Ct / kstep);
  int sk = 1;
  if (tiles < 256) {
    sk = (int)((512 + tiles - 1) / tiles);
    int maxsk = ksteps / 4;
    if (maxsk < 1) maxsk = 1;
    if (sk > maxsk) sk = maxsk;
    if (sk > 32) sk = 32;
    if (sk < 1) sk = 1;
  }
  pl.splitk = sk;
  return pl;
}

template <typename T, int BM, int BN, int WGM, int WGN>
static void launch_igemm(const ConvP& p0, hipStream_t st) {
  ConvP p = p0;
  constexpr int KCH = KCH_DEFAULT;
  p.tiles_n = (p.Cout + BN - 1) / BN;
  const int tiles_m = (p.M + BM - 1) / BM;
  dim3 grid(tiles_m * p.tiles_n, 1, p.splitk);
  constexpr int lds = 2 * KCH * (BM + BN) * 64;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)k_igemm<T, BM, BN, WGM, WGN, KCH>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set = true;
  }
  k_igemm<T, BM, BN, WGM, WGN, KCH><<<grid, WGM * WGN * 64, lds, st>>>(p);
}

template <typename T>
static int conv_dispatch(const afldm_conv_args* a, hipStream_t st) {
  ConvP p;
  p.x1 = a->x1; p.x2 = a->x2; p.w = a->w; p.bias = a->bias; p.temb = a->temb; p.residual = a->residual;
  p.y = a->y; p.ws = (float*)a->workspace;
  p.C1 = a->C1; p.C2 = a->C2; p.B = a->B; p.H = a->H; p.W = a->W; p.Cout = a->Cout; p.KS = a->KS;
  p.temb_stride = a->temb_stride; p.res_ld = a->res_ld; p.y_ld = a->y_ld; p.out_mode = a->out_mode;
  p.M = a->B * a->H * a->W;
  p.splitk = 1; p.tiles_n = 1; p.ksteps = 0;
  Plan pl = make_plan(a, epr<T>());
  if (pl.kind == 1) {
    AFLDM_REQUIRE(a->C2 == 0 && a->C1 <= 64, AFLDM_ESHAPE,
                  "afldm_conv2d: Cin=%d+%d is not a multiple of %d and too large for the direct kernel", a->C1,
                  a->C2, KCH_DEFAULT * epr<T>());
    size_t total = (size_t)p.M * p.Cout;
    int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    k_conv_small_cin<T><<<grid, 256, 0, st>>>(p);
    return check_launch("afldm_conv2d(small_cin)");
  }
  if (pl.kind == 2) {
    AFLDM_REQUIRE(a->C2 == 0 && a->Cout <= 8, AFLDM_ESHAPE, "afldm_conv2d: unsupported small-Cout shape (Cout=%d, C2=%d)",
                  a->Cout, a->C2);
    int grid = (p.M + 3) / 4 < 8192 ? (p.M + 3) / 4 : 8192;
    k_conv_small_cout<T, 8><<<grid, 256, 0, st>>>(p);
    return check_launch("afldm_conv2d(small_cout)");
  }
  const int Ct = a->C1 + a->C2;
  p.ksteps = a->KS * a->KS * (Ct / (KCH_DEFAULT * epr<T>()));
  p.splitk = pl.splitk;
  if (p.splitk > 1) {
    size_t need = (size_t)p.splitk * p.M * p.Cout * sizeof(float);
    if (!a->workspace || a->workspace_bytes < need) p.splitk = 1;  // no workspace -> no split
  }
  switch (pl.cfg) {
    case 0: launch_igemm<T, 128, 128, 2, 2>(p, st); break;
    case 1: launch_igemm<T, 128, 64, 2, 2>(p, st); break;
    case 2: launch_igemm<T, 64, 128, 2, 2>(p, st); break;
    default: launch_igemm<T, 64, 64, 2, 2>(p, st); break;
  }
  int rc = check_launch("afldm_conv2d(igemm)");
  if (rc) return rc;
  if (p.splitk > 1) {
    size_t total = (size_t)p.M * ((p.Cout + 3) / 4);
    int grid = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    k_splitk_reduce<T><<<grid, 256, 0, st>>>(p);
    rc = check_launch("afldm_conv2d(splitk_reduce)");
  }
  return rc;
}

static int conv_validate(const afldm_conv_args* a) {
  AFLDM_REQUIRE(a != nullptr, AFLDM_ENULL, "afldm_conv2d: args is NULL");
  AFLDM_REQUIRE(a->x1 && a->w && a->y, AFLDM_ENULL, "afldm_conv2d: x1/w/y must be non-NULL");
  AFLDM_REQUIRE(a->KS == 1 || a->KS == 3, AFLDM_ESHAPE, "afldm_conv2d: KS=%d not in {1,3}", a->KS);
  AFLDM_REQUIRE(a->B > 0 && a->H > 0 && a->W > 0 && a->Cout > 0 && a->C1 > 0 && a->C2 >= 0, AFLDM_ESHAPE,
                "afldm_conv2d: bad shape B=%d H=%d W=%d Cout=%d C1=%d C2=%d", a->B, a->H, a->W, a->Cout, a->C1, a->C2);
  AFLDM_REQUIRE(a->C2 == 0 || a->x2 != nullptr, AFLDM_ENULL, "afldm_conv2d: C2>0 but x2 is NULL");
  AFLDM_REQUIRE(a->out_mode == 0 || a->out_mode == 1, AFLDM_ESHAPE, "afldm_conv2d: out_mode %d", a->out_mode);
  AFLDM_REQUIRE(a->out_mode == 1 || a->y_ld >= a->Cout, AFLDM_ESHAPE, "afldm_conv2d: y_ld=%d < Cout=%d", a->y_ld, a->Cout);
  AFLDM_REQUIRE(!a->residual || a->res_ld >= a->Cout, AFLDM_ESHAPE, "afldm_conv2d: res_ld=%d < Cout=%d", a->res_ld, a->Cout);
  AFLDM_REQUIRE((long long)a->B * a->H * a->W < (1ll << 30), AFLDM_ESHAPE, "afldm_conv2d: M too large");
  AFLDM_REQUIRE(aligned16(a->x1) && aligned16(a->w) && aligned16(a->y) && aligned16(a->x2) && aligned16(a->residual) &&
                    aligned16(a->temb) && aligned16(a->bias),
                AFLDM_EALIGN, "afldm_conv2d: all pointers must be 16-byte aligned");
  AFLDM_REQUIRE(a->Cout < 16 || ((a->out_mode == 1 || a->y_ld % 4 == 0) && (!a->residual || a->res_ld % 4 == 0) &&
                                 (!a->temb || a->temb_stride % 4 == 0)),
                AFLDM_EALIGN, "afldm_conv2d: y_ld/res_ld/temb_stride must be multiples of 4 elements");
  return AFLDM_OK;
}

}  // namespace afldm

using namespace afldm;

extern "C" size_t afldm_conv2d_workspace(const afldm_conv_args* a) {
  if (!a || (a->dtype != AFLDM_F32 && a->dtype != AFLDM_BF16)) return 0;
  Plan pl = make_plan(a, a->dtype == AFLDM_F32 ? 16 : 32);
  if (pl.kind != 0 || pl.splitk <= 1) return 0;
  return (size_t)pl.splitk * a->B * a->H * a->W * a->Cout * sizeof(float);
}

extern "C" int afldm_conv2d(const afldm_conv_args* a, afldm_stream_t stream) {
  int rc = conv_validate(a);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  if (a->dtype == AFLDM_F32) return conv_dispatch<float>(a, st);
  if (a->dtype == AFLDM_BF16) return conv_dispatch<bf16>(a, st);
  set_error("afldm_conv2d: unknown dtype %d", a->dtype);
  return AFLDM_EDTYPE;
}
