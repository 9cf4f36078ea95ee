// conv3h_tile.hpp - the halo-patch 3x3 convolution tile (see conv3h.hip for the design notes): MFMA flavours, the
// swizzled LDS image helpers and the tile body shared by k_conv3h (conv3h.hip) and the merged launches (actconv.hip).
#pragma once
#include "conv_common.hpp"


#include <type_traits>

namespace afldm {

typedef __attribute__((ext_vector_type(16))) float f32x16;

// MFMA flavour of the consumers.  MF = 16: v_mfma_f32_16x16x32_bf16 / 4 x 16x16x4_f32 (Mma<T>, common.hpp): lane
// (i = l & 15, g = l >> 4) feeds chunk kc * 4 + g of row i; accumulator = 4 consecutive couts of one pixel.
// MF = 32: v_mfma_f32_32x32x16_bf16 / 4 x 32x32x2_f32: lane (i = l & 31, g = l >> 5) feeds chunk 2 kk + g
// (kk = 0..3) of row i; the accumulator (16 floats) holds couts 8 rq + 4 g + e (rq, e = 0..3) of pixel l & 31.
// The 32x32 shape sustains ~15 % more matrix throughput on this chip (2382 vs 2075 TF in the guide's micro-benchmarks:
// 32 instead of 2 x ~19 issue cycles for the same 16 K multiply-adds) and the K loop of this kernel sits on the
// matrix pipe.
template <typename T>
struct Mma32;
template <>
struct Mma32<bf16> {
  static __device__ __forceinline__ void mma(f32x16& acc, const bf16x8& a, const bf16x8& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
  }
};
template <>
struct Mma32<float> {
  static __device__ __forceinline__ void mma(f32x16& acc, const f32x4& a, const f32x4& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], b[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], b[3], acc, 0, 0, 0);
  }
};

// LDS position of chunk c of row q and its inverse (see the header): MF = 16 keeps bit 2 for the lane-group parity and
// swizzles two bits, MF = 32 (the 16-lane read groups lie inside one 32-row half) swizzles all three.
// The row's swizzle bits sw: (q >> 1) & 3 (& 7 for MF = 32) of the row index for the weight rows and for patches of
// planes 16+ wide, where 16 consecutive tile pixels are 16 consecutive patch pixels.  On the 8x8 / 4x4 planes a
// fragment's 16 pixels span 2 / 4 image rows (patch rows are W + 2 pixels apart) and that rule put two of every 8
// equal-lane-group rows on the same banks - every patch read of those variants took two LDS cycles.  There the bits
// come from the patch COORDINATES (pr, pc): column pair (pc >> 1) & 3 on 8-wide planes (the 8 pixels of one lane
// group in a read group are columns c .. c+3 of one row and c+4 .. c+7 of the next); column pair and row parity on
// 4-wide planes (rows a, a+3 or a+1, a+2) - conflict free for all nine tap shifts (checked by enumeration).
template <int MF>
__device__ __forceinline__ int h_sw_rows(int q) {
  return MF == 16 ? (q >> 1) & 3 : (q >> 1) & 7;
}
template <int MF, int W_>
__device__ __forceinline__ int h_sw_patch(int q) {
  if constexpr (MF == 16 && W_ <= 8) {
    const int pr = q / (W_ + 2), pc = q - pr * (W_ + 2);
    return W_ == 8 ? (pc >> 1) & 3 : (((pc >> 1) & 1) | ((pr & 1) << 1));
  } else {
    return h_sw_rows<MF>(q);
  }
}
template <int MF>
__device__ __forceinline__ int h_pos(int c, int sw) {
  if constexpr (MF == 16) return (((c & 1) << 2) | ((c >> 2) << 1) | ((c >> 1) & 1)) ^ sw;
  else return c ^ sw;
}
template <int MF>
__device__ __forceinline__ int h_chunk_at(int pos, int sw) {   // source chunk that lives at position `pos` of a row with swizzle bits sw
  if constexpr (MF == 16) {
    const int x = pos ^ sw;
    return ((x >> 1) & 1) * 4 + (x & 1) * 2 + (x >> 2);
  } else {
    return pos ^ sw;
  }
}

// TPS: filter taps per K step (1, or 3 = one filter row): the small-plane variants (64 x 96 tiles over the 8x8 / 4x4
// levels, one MFMA wave per SIMD) do 3 taps between two barriers so that a step still carries 36 MFMAs per wave.
// Tiles may hold several whole samples (BM >= H * W: NSEG segments of SEG = H rows, each with its own halo rows) and
// the channel blocks may be split over blockIdx.z (fp32 slabs, reduced by k_splitk_reduce*, conv.hip).
// SUB: the plane is LARGER than the tile (the AF-VAE's 64^2 .. 256^2 planes): a tile is a ROWS x W_ block of an
// H x W plane, its patch the (ROWS + 2) x (W_ + 2) block around it - zero only where that leaves the image - and the
// tile's pixels are W_-long runs p.W pixels apart in memory.  (The implicit GEMM re-fetched the pixel tile for every
// tap there too: 0.69 PFLOP/s at 256^2 x 128 channels, profiles/r03.)

}  // namespace afldm
