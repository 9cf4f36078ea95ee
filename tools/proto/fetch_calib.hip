// fetch_calib.hip - calibration of rocprofv3's FETCH_SIZE on gfx950 for the access widths the step's kernels use
// (MI355X_MICROARCH.md: "FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read ... other access
// widths are uncalibrated: calibrate on a known byte count in your own access pattern").  Every mode reads each byte of the
// buffer exactly once; what changes is the contiguous run a workgroup touches per pixel row (bytes of one channel tile).
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(4))) unsigned int u4;

// tensor [P][C] bf16 (row = C * 2 bytes); a workgroup owns channel tile `ct` (PIECE bytes per pixel) of a block of pixels
template <int PIECE>
__global__ void __launch_bounds__(256) k_pieces(const char* __restrict__ x, unsigned* __restrict__ out, int P, int rowbytes, int pix_per_wg) {
  constexpr int LPP = PIECE / 16;                  // lanes per pixel
  const int tiles = rowbytes / PIECE;
  const int ct = blockIdx.x % tiles, pb = blockIdx.x / tiles;
  unsigned acc = 0;
  for (int i = threadIdx.x; i < pix_per_wg * LPP; i += 256) {
    const int pix = pb * pix_per_wg + i / LPP, q = i % LPP;
    if (pix < P) {
      const u4 v = *reinterpret_cast<const u4*>(x + (size_t)pix * rowbytes + ct * PIECE + q * 16);
      acc += v[0] ^ v[1] ^ v[2] ^ v[3];
    }
  }
  if (acc == 0x12345678u) out[0] = acc;
}

__global__ void __launch_bounds__(256) k_stream(const char* __restrict__ x, unsigned* __restrict__ out, size_t bytes) {
  unsigned acc = 0;
  for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 16; i < bytes; i += (size_t)gridDim.x * 256 * 16) {
    const u4 v = *reinterpret_cast<const u4*>(x + i);
    acc += v[0] ^ v[1] ^ v[2] ^ v[3];
  }
  if (acc == 0x12345678u) out[0] = acc;
}

// ---- WRITE_SIZE (round 5): every byte of the buffer WRITTEN exactly once, in PIECE-byte runs per pixel row by
// different workgroups (the plane activation's stores: 16 / 32 bytes of a pixel's channels per item), plain or write-through (sc1)
template <int PIECE, bool SC1>
__global__ void __launch_bounds__(256) k_wpieces(char* __restrict__ x, int P, int rowbytes, int pix_per_wg) {
  constexpr int LPP = PIECE / 16;
  const int tiles = rowbytes / PIECE;
  const int ct = blockIdx.x % tiles, pb = blockIdx.x / tiles;
  for (int i = threadIdx.x; i < pix_per_wg * LPP; i += 256) {
    const int pix = pb * pix_per_wg + i / LPP, q = i % LPP;
    if (pix < P) {
      const u4 v = {(unsigned)pix, (unsigned)q, (unsigned)ct, 7u};
      char* dst = x + (size_t)pix * rowbytes + ct * PIECE + q * 16;
      if constexpr (SC1) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory");
      else *reinterpret_cast<u4*>(dst) = v;
    }
  }
}
template <bool SC1>
__global__ void __launch_bounds__(256) k_wstream(char* __restrict__ x, size_t bytes) {
  for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 16; i < bytes; i += (size_t)gridDim.x * 256 * 16) {
    const u4 v = {(unsigned)i, 1u, 2u, 3u};
    if constexpr (SC1) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(x + i), "v"(v) : "memory");
    else *reinterpret_cast<u4*>(x + i) = v;
  }
}

extern "C" int fc_wrun(void* x, int mode, int sc1, long long P, int rowbytes, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const int ppw = 1024;
  const int pbs = (int)((P + ppw - 1) / ppw);
#define WCASE(N)                                                                                             \
  case N:                                                                                                   \
    if (sc1) k_wpieces<N, true><<<pbs * (rowbytes / N), 256, 0, st>>>((char*)x, (int)P, rowbytes, ppw);    \
    else k_wpieces<N, false><<<pbs * (rowbytes / N), 256, 0, st>>>((char*)x, (int)P, rowbytes, ppw);       \
    break;
  switch (mode) {
    case 0:
      if (sc1) k_wstream<true><<<2048, 256, 0, st>>>((char*)x, (size_t)P * rowbytes);
      else k_wstream<false><<<2048, 256, 0, st>>>((char*)x, (size_t)P * rowbytes);
      break;
    WCASE(16) WCASE(32) WCASE(64) WCASE(128)
    default: return 1;
  }
#undef WCASE
  return (int)hipGetLastError();
}

extern "C" int fc_run(const void* x, void* out, int mode, long long P, int rowbytes, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const int ppw = 1024;
  const int pbs = (int)((P + ppw - 1) / ppw);
  switch (mode) {
    case 0: k_stream<<<2048, 256, 0, st>>>((const char*)x, (unsigned*)out, (size_t)P * rowbytes); break;
    case 16: k_pieces<16><<<pbs * (rowbytes / 16), 256, 0, st>>>((const char*)x, (unsigned*)out, (int)P, rowbytes, ppw); break;
    case 32: k_pieces<32><<<pbs * (rowbytes / 32), 256, 0, st>>>((const char*)x, (unsigned*)out, (int)P, rowbytes, ppw); break;
    case 64: k_pieces<64><<<pbs * (rowbytes / 64), 256, 0, st>>>((const char*)x, (unsigned*)out, (int)P, rowbytes, ppw); break;
    case 128: k_pieces<128><<<pbs * (rowbytes / 128), 256, 0, st>>>((const char*)x, (unsigned*)out, (int)P, rowbytes, ppw); break;
    default: return 1;
  }
  return (int)hipGetLastError();
}
