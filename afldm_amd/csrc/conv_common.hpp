// conv_common.hpp — argument block and LDS-DMA helpers shared by conv.hip (implicit GEMM) and conv3h.hip
// (halo-patch 3x3 convolution).
#pragma once
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
  void* y2;      // optional second output (channel-major) for couts >= split_n
  float* ws;
  int split_n;
  int C1, C2, B, H, W, Cout, KS;
  int temb_stride, res_ld, y_ld, out_mode;
  int temb_mod;  // temb column = cout % temb_mod (a 3x3 conv on a 2x2 plane run as one dense layer: cout = pixel * C + c)
  int M;        // B*H*W
  int ksteps;   // total K steps = KS*KS * (C1+C2)/(KCH*EPR)
  int splitk;   // grid.z
  int tiles_n;
  int vec_ok;   // leading dims allow 4-element vector epilogue accesses
  int tap_inner;  // K order of the LDS-DMA kernel (see k_igemm2)
  float* stats_out;  // per-channel GroupNorm partial sums of the output [B][stats_S][Cout][2], or NULL
  int stats_S;
  int stats_multi;   // the tile spans BM / (H*W) whole samples: statistics per sample with S = 1 (64x64 tiles only)
  int m_fast;     // tile order of the LDS-DMA kernel: 1 = tile_m fastest (weights outweigh pixels)
  int stage_ok;   // leading dimensions / splits allow the LDS-staged 16-byte epilogue
  int dbg;      // AFLDM_CONV_DBG (timing decomposition only): bit 0 skip the LDS-DMA, bit 1 skip the MFMA phase
  int xcd_gn;    // conv3h tile order: the XCDs as a (8 / xcd_gn) x xcd_gn grid over (m, n) tiles; 0 = contiguous runs, n fastest
  unsigned* sync;   // in-kernel split-K reduction: 2 zero-initialised words per output tile (arrivals, departures), or NULL
  long long w_bstride;   // elements between the weight tensors of consecutive samples (0: shared weights); k_igemm2 only
  // the NEXT GroupNorm applied by the epilogue of a halo-patch tile that holds a whole sample (8x8 planes): y_norm [M][Cout] =
  // GroupNorm(y) with this tile's own statistics (groups of ncpg channels inside the tile's couts), or NULL
  void* y_norm;
  const float* ngamma;
  const float* nbeta;
  int ncpg;
  float neps;
  // 8-channel-block layout [B][C / EPC][H][W][EPC] (EPC = 16 bytes of elements) instead of NHWC, halo-patch kernel only, one whole
  // K per workgroup, tiles inside one sample: x_c8 for the pixel operand (the patch loader reads 16-byte chunks either way),
  // y_c8 for the output rows.  The tensors between the alias-free activations and the 3x3 convolutions of a ResnetBlock2D
  // travel in this layout: an activation item (8 / 16 channels of one sample) is then one / two contiguous runs.
  int x_c8, y_c8;
  // 1: the weight stream of this launch is read by exactly ONE workgroup per slice (a single row tile: the small batches, the
  // 2x2 / 4x4 levels) - requested non-temporal (LDS-DMA aux = 2 / `nt` loads) so that it does not displace the activations the
  // next launches re-read from the L2s (MI355X_MICROARCH "nt-weights"; AFLDM_NT_WEIGHTS=0: off, A/B)
  int w_nt;
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// halo-patch 3x3 convolution (conv3h.hip): variant ids >= kConv3hFirst of the conv variant table
constexpr int kConv3hFirst = 41;
bool conv3h_supported(int variant, int dtype_size, const ConvP& p);      // p.splitk = the planned K slices
int conv3h_tile(int variant, int* bm, int* bn);
void conv3h_launch(int variant, int dtype_size, const ConvP& p, hipStream_t st);
// conv.hip: true when afldm_conv2d(a) is ONE k_conv3h launch (whole K per workgroup, statistics from its epilogue); fills the
// argument block of that launch and its variant id (the merged launches of actconv.hip run the tile as their last phase)
bool conv3h_plan(const afldm_conv_args* a, ConvP* p, int* variant);

}  // namespace afldm
