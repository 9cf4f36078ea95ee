// fir.hip — upfirdn2d: zero-stuffing upsample -> pad / crop -> 2-D FIR -> decimate, on NCHW planes.
// The reference keeps this operator as a vendored StyleGAN3 CUDA plugin with a PyTorch fallback
// (afldm/af_libs/torch_utils/ops/upfirdn2d.py:140-194 `_upfirdn2d_ref`, upfirdn2d.cu); it serves the
// Lanczos fractional translation (af_libs/equivariance.py:68-103), ImageShifter('lanczos')
// (shift_utils/shifters.py:158-161) and the 'blur' ImageUpsampler / ImageDownsampler (shifters.py:292-365).
//
// HBM-bound: every output is a short dot product over the taps whose up-sampling phase hits a real
// sample; the zero-stuffed and padded intermediate never exists.  A workgroup covers a 256 x 16 output
// patch of one plane (4 columns x 4 rows per thread; the tap re-reads are served by L1/L2), planes on
// the grid's z axis.  The flipped, gain-scaled taps live in LDS.  Accumulation is fp32 in any dtype.
#include "common.hpp"

namespace afldm {

struct FirP {
  int planes, H, W, outH, outW, fh, fw, upx, upy, downx, downy, padx0, pady0, flip;
  float gain;
};

// Each thread owns VX consecutive output columns on RY rows (rows 4 apart inside the workgroup's
// (64 VX) x (4 RY) patch): one output per thread makes the launch rate of tiny workgroups the bound
// (49 k workgroups for a 64 x 3 x 256^2 batch ran at 1.1 TB/s).  Without horizontal re-sampling the VX
// outputs slide a register window over the row, so a tap row costs VX + fw - 1 loads instead of VX fw.
template <typename T, int VX, int RY>
__global__ void __launch_bounds__(256) k_upfirdn2d(const T* __restrict__ x, const float* __restrict__ f,
                                                   T* __restrict__ y, FirP p) {
  extern __shared__ float taps[];   // correlation taps: out[oy][ox] = sum taps[fy][fx] * up[oy*dy + fy][ox*dx + fx]
  const int tid = threadIdx.y * 64 + threadIdx.x, ntaps = p.fh * p.fw;
  for (int i = tid; i < ntaps; i += 256) {
    const int fy = i / p.fw, fx = i - fy * p.fw;
    const float v = p.flip ? f[i] : f[(p.fh - 1 - fy) * p.fw + (p.fw - 1 - fx)];
    taps[i] = v * p.gain;
  }
  __syncthreads();
  const bool slide = p.upx == 1 && p.downx == 1;
  // column ownership: consecutive (ox0 .. ox0 + VX - 1) when the window slides, lane-interleaved
  // (ox0 + 64 v) otherwise so that one load instruction still covers neighbouring columns
  const int ox0 = slide ? (blockIdx.x * 64 + threadIdx.x) * VX : blockIdx.x * 64 * VX + threadIdx.x;
  const int oxs = slide ? 1 : 64;
  if (ox0 >= p.outW) return;
  // pure column filter on aligned rows: the VX columns are one 16-byte (fp32) / 8-byte (bf16) load per tap
  const bool vec = slide && p.fw == 1 && p.padx0 == 0 && p.W % VX == 0 &&
                   (reinterpret_cast<uintptr_t>(x) % (VX * sizeof(T))) == 0;
  for (int pl = blockIdx.z; pl < p.planes; pl += gridDim.z) {
    const T* xp = x + (size_t)pl * p.H * p.W;
#pragma unroll 1
    for (int r = 0; r < RY; ++r) {
      const int oy = (blockIdx.y * RY + r) * 4 + threadIdx.y;
      if (oy >= p.outH) break;
      // first tap (per axis) that lands on a real sample: up-sampled coordinate u = o*down - pad0 + tap
      const int u0 = oy * p.downy - p.pady0;
      const int fy0 = u0 < 0 ? -u0 : (p.upy - u0 % p.upy) % p.upy;
      const int iy0 = (u0 + fy0) / p.upy;
      const int ny = fy0 < p.fh ? min((p.fh - 1 - fy0) / p.upy + 1, p.H - iy0) : 0;
      float acc[VX];
#pragma unroll
      for (int v = 0; v < VX; ++v) acc[v] = 0.f;
      if (vec && ox0 + VX <= p.W) {
        typedef T VecT __attribute__((ext_vector_type(VX)));
        for (int a = 0; a < ny; ++a) {
          const VecT val = *reinterpret_cast<const VecT*>(xp + (size_t)(iy0 + a) * p.W + ox0);
          const float tp = taps[fy0 + a * p.upy];
#pragma unroll
          for (int v = 0; v < VX; ++v) acc[v] = fmaf(tp, to_f32((T)val[v]), acc[v]);
        }
      } else if (slide) {
        const int c0 = ox0 - p.padx0;   // input column under tap 0 of output ox0
        for (int a = 0; a < ny; ++a) {
          const T* row = xp + (size_t)(iy0 + a) * p.W;
          const float* trow = taps + (fy0 + a * p.upy) * p.fw;
          float win[VX];
#pragma unroll
          for (int v = 0; v < VX - 1; ++v) {
            const int c = c0 + v;
            win[v + 1] = (c >= 0 && c < p.W) ? to_f32(row[c]) : 0.f;
          }
          for (int fx = 0; fx < p.fw; ++fx) {
#pragma unroll
            for (int v = 0; v < VX - 1; ++v) win[v] = win[v + 1];
            const int c = c0 + fx + VX - 1;
            win[VX - 1] = (c >= 0 && c < p.W) ? to_f32(row[c]) : 0.f;
            const float tp = trow[fx];
#pragma unroll
            for (int v = 0; v < VX; ++v) acc[v] = fmaf(tp, win[v], acc[v]);
          }
        }
      } else {
#pragma unroll
        for (int v = 0; v < VX; ++v) {
          const int v0 = (ox0 + v * oxs) * p.downx - p.padx0;
          const int fx0 = v0 < 0 ? -v0 : (p.upx - v0 % p.upx) % p.upx;
          const int ix0 = (v0 + fx0) / p.upx;
          const int nx = fx0 < p.fw ? min((p.fw - 1 - fx0) / p.upx + 1, p.W - ix0) : 0;
          for (int a = 0; a < ny; ++a) {
            const T* row = xp + (size_t)(iy0 + a) * p.W + ix0;
            const float* trow = taps + (fy0 + a * p.upy) * p.fw + fx0;
            for (int b = 0; b < nx; ++b) acc[v] = fmaf(trow[b * p.upx], to_f32(row[b]), acc[v]);
          }
        }
      }
      T* yo = y + ((size_t)pl * p.outH + oy) * p.outW + ox0;
#pragma unroll
      for (int v = 0; v < VX; ++v)
        if (ox0 + v * oxs < p.outW) yo[v * oxs] = from_f32<T>(acc[v]);
    }
  }
}

template <typename T>
static void launch_upfirdn2d(const void* x, const float* f, void* y, const FirP& p, hipStream_t st) {
  constexpr int VX = 4;
  const size_t lds = (size_t)p.fh * p.fw * sizeof(float);
  const dim3 block(64, 4);
  const int gz = p.planes < 4096 ? p.planes : 4096;
  if (p.outH >= 16) {
    constexpr int RY = 4;
    const dim3 grid((p.outW + 64 * VX - 1) / (64 * VX), (p.outH + 4 * RY - 1) / (4 * RY), gz);
    k_upfirdn2d<T, VX, RY><<<grid, block, lds, st>>>((const T*)x, f, (T*)y, p);
  } else {
    const dim3 grid((p.outW + 64 * VX - 1) / (64 * VX), (p.outH + 3) / 4, gz);
    k_upfirdn2d<T, VX, 1><<<grid, block, lds, st>>>((const T*)x, f, (T*)y, p);
  }
}

}  // namespace afldm

using namespace afldm;

extern "C" int afldm_upfirdn2d(const void* x, const float* f, void* y, int planes, int H, int W, int fh, int fw,
                               int upx, int upy, int downx, int downy, int padx0, int padx1, int pady0, int pady1,
                               int flip_filter, float gain, int dtype, afldm_stream_t stream) {
  AFLDM_REQUIRE(x && f && y, AFLDM_ENULL, "afldm_upfirdn2d: NULL pointer");
  AFLDM_REQUIRE(planes > 0 && H > 0 && W > 0 && fh > 0 && fw > 0, AFLDM_ESHAPE,
                "afldm_upfirdn2d: bad shape planes=%d H=%d W=%d f=%dx%d", planes, H, W, fh, fw);
  AFLDM_REQUIRE(upx >= 1 && upy >= 1 && downx >= 1 && downy >= 1, AFLDM_ESHAPE,
                "afldm_upfirdn2d: up / down factors must be >= 1 (got up=%d,%d down=%d,%d)", upx, upy, downx, downy);
  const long upW = (long)W * upx + padx0 + padx1, upH = (long)H * upy + pady0 + pady1;
  // upfirdn2d.py:159-162: the padded / cropped up-sampled plane must not be smaller than the filter
  AFLDM_REQUIRE(upW >= fw && upH >= fh, AFLDM_ESHAPE,
                "afldm_upfirdn2d: up-sampled plane %ldx%ld smaller than the %dx%d filter", upH, upW, fh, fw);
  AFLDM_REQUIRE((size_t)fh * fw * sizeof(float) <= 64 * 1024, AFLDM_ESHAPE,
                "afldm_upfirdn2d: filter %dx%d exceeds the 64 KiB tap buffer", fh, fw);
  FirP p;
  p.planes = planes; p.H = H; p.W = W; p.fh = fh; p.fw = fw;
  p.upx = upx; p.upy = upy; p.downx = downx; p.downy = downy;
  p.padx0 = padx0; p.pady0 = pady0; p.flip = flip_filter ? 1 : 0; p.gain = gain;
  p.outW = (int)((upW - fw) / downx + 1);
  p.outH = (int)((upH - fh) / downy + 1);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == AFLDM_F32)
    launch_upfirdn2d<float>(x, f, y, p, st);
  else if (dtype == AFLDM_BF16)
    launch_upfirdn2d<bf16>(x, f, y, p, st);
  else {
    set_error("afldm_upfirdn2d: unknown dtype %d", dtype);
    return AFLDM_EDTYPE;
  }
  return check_launch("afldm_upfirdn2d");
}

extern "C" int afldm_upfirdn2d_out_size(int in, int up, int down, int pad0, int pad1, int ftaps) {
  const long u = (long)in * up + pad0 + pad1;
  return u >= ftaps ? (int)((u - ftaps) / down + 1) : -1;
}
