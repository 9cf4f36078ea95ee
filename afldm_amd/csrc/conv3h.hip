// conv3h.hip — 3x3 'same' convolution on MFMA with a HALO PATCH staged once per channel block (gfx950).
//
// The implicit GEMM of conv.hip re-fetches its pixel tile for every one of the 9 filter taps: 9 x (BM x 128 B)
// of LDS-DMA per 64-channel block, and the PMC counters put its L2-miss traffic at 3x the algorithmic bytes
// (profiles/conv3x3_traffic.json, round 1).  At full MFMA rate that stream alone asks for ~53 B/cycle/CU of
// L2 -> LDS bandwidth, which is what the chip's L2s deliver in aggregate: the kernel was L2-bound as much as
// MFMA-bound.  Here a workgroup owns BM pixels = BM / W WHOLE image rows of one sample and, per 64-channel
// (128-byte) block, stages the (rows + 2) x (W + 2) pixel patch ONCE (zero padding written by the buffer
// descriptor's bounds check); the nine taps are nine shifted views of that patch, so only the weights
// (BN x 128 B per tap) stream per K step:
//
//   K order   : channel block outer, tap inner   (k_igemm2: tap outer, channel block inner)
//   LDS       : [patch 0][patch 1][weight ring: STAGES x BN x 128 B]
//   per step  : DMA  BN x 128 B (+ patch / 9)        against (BM + BN) x 128 B before
//   roles     : WGM x WGN consumer waves (LDS fragment reads + MFMA only) + NPROD producer waves
//               (LDS-DMA issue + counted vmcnt only), one s_barrier per K step
//
// LDS image: one 128-byte row per patch pixel / weight row holding the 8 16-byte chunks (chunk c = kc * 4 + lg:
// MFMA K-chunk kc, lane group lg) of the channel block at position
//       pos(c, q) = ((lg & 1) << 2 | kc << 1 | lg >> 1)  ^  sw(q),   sw(q) = (q >> 1) & 3,   q = row index
//       (patches of 8x8 / 4x4 planes: sw from the patch coordinates, see h_sw_patch)
// A ds_read_b128 is served in 16-lane groups made of 8 rows of lane group lg and 8 rows of lg ^ 1: bit 2 of the
// position separates the two halves, and 4 rows of equal parity inside ANY run of 16 consecutive rows differ in
// (q >> 1) & 3 - so the fragment reads are bank-conflict free for every tap shift of the patch (the swizzle of
// k_igemm2, (q >> 1) & 7 on all three bits, is only conflict free for 16-aligned runs).  With 32x32x16 MFMAs a 16-lane
// read group lies inside one 32-row fragment half (rows {0-3, 12-15, 20-27} + base, one chunk): there the plain
// position c ^ ((q >> 1) & 7) is conflict free for every shift.  An LDS-DMA writes lane-linearly, so the permutation
// is applied to the per-lane SOURCE chunk (guide rule 21).
//
// The residual enters through the ACCUMULATORS during the first K steps (fragment layout: its load latency runs under
// MFMA work).  Epilogue: (residual + sum) + (bias + temb) in fp32, one rounding; bf16 tiles are staged ONCE in bf16
// through the idle pipeline buffers and leave as whole rows, 16 bytes per lane (write-through stores), fp32 tiles and
// split-K slabs 128 rows at a time; per-channel GroupNorm partial sums of the stored values accumulated by the thread
// that owns the column (fixed order, no atomics); tiles of several whole samples (4x4 planes) carry one time-embedding
// vector and one statistics record per sample.
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
template <typename T, int BM, int W_, int BN, int WGM, int WGN, int NPROD, int STAGES, int MINW, int MF, int TPS, bool SUB = false>
__global__ void __launch_bounds__((WGM * WGN + NPROD) * 64, MINW) k_conv3h(ConvP p) {
  typedef Mma<T> MM;
  typedef typename MM::Chunk Chunk;
  constexpr int NWC = WGM * WGN;
  constexpr int EPC = MM::EPC, KSTEP = 8 * EPC, ESZ = (int)sizeof(T);
  constexpr int WMS = BM / WGM, WNS = BN / WGN, TM = WMS / MF, TN = WNS / MF;
  constexpr int ROWS = BM / W_, SEG = ROWS < W_ ? ROWS : W_, NSEG = ROWS / SEG;      // planes are square: H == W_
  constexpr int PW = W_ + 2, PR = NSEG * (SEG + 2), NPQ = PR * PW, NPI = (NPQ + 7) / 8;
  constexpr int PATCH = NPI * 1024;
  constexpr int SPC = 9 / TPS;                                   // K steps per channel block
  constexpr int WIT = BN / 8;                                    // weight instructions per tap
  constexpr int WI = TPS * WIT, WPW = WI / NPROD, PPW = (NPI + NPROD - 1) / NPROD;
  constexpr int W_TAP = BN * 128, W_STAGE = TPS * W_TAP;
  constexpr int LDS_TOTAL = 2 * PATCH + STAGES * W_STAGE;
  constexpr unsigned OOB = 0x80000000u;
  static_assert(MF == 16 || MF == 32, "MFMA flavour");
  static_assert(TPS == 1 || TPS == 3, "taps per step");
  static_assert(BM % W_ == 0 && ROWS % SEG == 0 && WI % NPROD == 0 && WMS % MF == 0 && WNS % MF == 0, "tile shape");
  static_assert((STAGES - 2) * WPW + PPW < 64, "vmcnt is a 6-bit counter");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool is_producer = wave_all >= NWC;
  const int li = lane & (MF - 1), lg = lane / MF;       // fragment row / lane group of the consumers

  // Tile order.  Default: each XCD (private L2) takes a contiguous run of tiles, n fastest - it reads 1/8 of the
  // pixels and ALL the weights.  p.xcd_gn > 0: the 8 XCDs form a (8 / gn) x gn grid over (m tiles, n tiles), so an
  // XCD reads gn / 8 of the weights and 1 / (8 / gn) of the pixels - the 4x4 level's 10-21 MB of weights were
  // fetched 8 times otherwise (launch_h3 picks gn from the byte counts).  Same time in the step (those fetches hit
  // the MALL), a third less fabric traffic for the family.
  int tile_m, tile_n;
  if (p.xcd_gn > 0) {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int gn = p.xcd_gn, tn_per = p.tiles_n / gn, tm_per = (gridDim.x / p.tiles_n) / (8 / gn);
    const int xm = xcd / gn, xn = xcd - xm * gn;
    const int lm = j / tn_per, ln = j - lm * tn_per;
    tile_m = xm * tm_per + lm;
    tile_n = xn * tn_per + ln;
  } else {
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    tile_m = tile / p.tiles_n;
    tile_n = tile - tile_m * p.tiles_n;
  }
  static_assert(!SUB || (NSEG == 1 && TPS == 1), "sub-tiled planes: one segment per tile");
  const int n0 = tile_n * BN;
  const int Ct = p.C1, HW = SUB ? p.H * p.W : W_ * W_;
  const int ncb = Ct / KSTEP;
  const int ks = blockIdx.z;
  const int cb_lo = (ncb * ks) / p.splitk, cb_hi = (ncb * (ks + 1)) / p.splitk;   // this slice's channel blocks
  const int G = (cb_hi - cb_lo) * SPC;
  int m0, b_tile, oh0, ow0 = 0, sp_tile;                         // first pixel, sample, tile origin, tile index inside the sample
  if constexpr (SUB) {
    const int tw = p.W / W_, tps = (p.H / ROWS) * tw;
    b_tile = tile_m / tps;
    sp_tile = tile_m - b_tile * tps;
    const int ty = sp_tile / tw;
    oh0 = ty * ROWS;
    ow0 = (sp_tile - ty * tw) * W_;
    m0 = b_tile * HW + oh0 * p.W + ow0;
  } else {
    m0 = tile_m * BM;
    b_tile = m0 / HW;
    oh0 = NSEG == 1 ? (m0 - b_tile * HW) / W_ : 0;               // first image row of the (single) segment
    sp_tile = (m0 - b_tile * HW) / BM;
  }
  // global pixel (row of x / y / residual) of tile pixel tp
  auto gpix = [&](int tp) -> int {
    if constexpr (SUB) return m0 + (tp / W_) * p.W + (tp % W_);
    else return m0 + tp;
  };

  if (is_producer) {
    const int wave = wave_all - NWC;
    __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x1, 0, (int)((long long)p.M * Ct * ESZ), 0x00020000);
    __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)((long long)p.Cout * 9 * Ct * ESZ), 0x00020000);
    // patch: instruction j covers patch pixels 8 j .. 8 j + 7, lane l = (pixel l >> 3, position l & 7).  Waves whose
    // share is one short re-issue the last instruction (same source, same destination): uniform vmcnt counts.
    unsigned poff[PPW], woff[WPW];
    int pj[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      int j = wave + NPROD * i;
      if (j > NPI - 1) j = NPI - 1;
      pj[i] = j;
      const int q = 8 * j + (lane >> 3);
      const int c = h_chunk_at<MF>(lane & 7, h_sw_patch<MF, W_>(q));
      const int pr = q / PW, pc = q - pr * PW;
      const int sg = pr / (SEG + 2), jj = pr - sg * (SEG + 2);            // segment (one sample's rows) and row inside it
      const int ih = oh0 + jj - 1;
      bool ok;
      int pixel;
      if constexpr (SUB) {
        const int iw = ow0 + pc - 1;
        ok = q < NPQ && iw >= 0 && iw < p.W && ih >= 0 && ih < p.H;
        pixel = b_tile * HW + ih * p.W + iw;
      } else {
        ok = q < NPQ && pc >= 1 && pc <= W_ && ih >= 0 && ih < W_;
        pixel = m0 + (sg * SEG + jj - 1) * W_ + (pc - 1);
      }
      poff[i] = ok ? ((unsigned)pixel * (unsigned)Ct + (unsigned)(c * EPC)) * ESZ : OOB;
    }
#pragma unroll
    for (int i = 0; i < WPW; ++i) {
      const int j = wave + NPROD * i;                            // instruction inside the stage: (tap in step, row group)
      const int tis = j / WIT, r = 8 * (j - tis * WIT) + (lane >> 3);
      const int c = h_chunk_at<MF>(lane & 7, h_sw_rows<MF>(r));
      woff[i] = (((unsigned)(n0 + r) * 9u + (unsigned)tis) * (unsigned)Ct + (unsigned)(c * EPC)) * ESZ;
    }
    // cursor of the next weight step to issue
    int is_g = 0, is_st = 0, is_cb = cb_lo, is_slot = 0;
    auto issue_weights = [&]() {
      char* sbase = smem + 2 * PATCH + is_slot * W_STAGE;
      const unsigned so = is_g < G ? (unsigned)((is_st * TPS * Ct + is_cb * KSTEP) * ESZ) : OOB;
#pragma unroll
      for (int i = 0; i < WPW; ++i) {
        lds_ptr_t dst = (lds_ptr_t)(sbase + (wave + NPROD * i) * 1024);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, dst, 16, (int)woff[i], (int)so, 0, 0);
      }
      ++is_g;
      if (++is_st == SPC) {
        is_st = 0;
        ++is_cb;
      }
      is_slot = is_slot + 1 == STAGES ? 0 : is_slot + 1;
    };
    auto issue_patch = [&](int cb) {
      char* sbase = smem + ((cb - cb_lo) & 1) * PATCH;
      const unsigned so = (unsigned)(cb * KSTEP * ESZ);
#pragma unroll
      for (int i = 0; i < PPW; ++i) {
        lds_ptr_t dst = (lds_ptr_t)(sbase + pj[i] * 1024);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, dst, 16, (int)poff[i], (int)so, 0, 0);
      }
    };
    issue_patch(cb_lo);
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s) issue_weights();
    int st = 0, cb = cb_lo, since = STAGES;                      // K steps since the last patch issue (in the loop)
    for (int g = 0; g < G; ++g) {
      // in flight behind the weights of step g: the weights of steps g+1 .. g+STAGES-2, and - for the STAGES-1 steps
      // after a patch issue - that patch (issued behind step g+STAGES-1's weights of its iteration)
      if (since <= STAGES - 1) wait_vmcnt<(STAGES - 2) * WPW + PPW>();
      else wait_vmcnt<(STAGES - 2) * WPW>();
      __builtin_amdgcn_s_barrier();
      ++since;
      if (!(p.dbg & 1)) {
        issue_weights();
        if (st == 0 && cb + 1 < cb_hi) {
          issue_patch(cb + 1);
          since = 1;
        }
      }
      if (++st == SPC) {
        st = 0;
        ++cb;
      }
    }
    wait_vmcnt<0>();   // drain the zero-fill tail before the workgroup's LDS can be re-assigned
    return;
  }

  // ------------------------------------------------------------------------------------------- consumers
  const int cw = wave_all, wm = cw / WGN, wn = cw - wm * WGN;
  constexpr int NKK = MF == 16 ? 2 : 4;                          // fragment K steps inside a 128-byte row
  constexpr int NACC = MF * MF / 64;                             // accumulator floats per lane and tile
  constexpr int RQ = NACC / 4;                                   // cout quads per lane and tile
  typedef __attribute__((ext_vector_type(NACC))) float AccT;
  int qb[TM];                                                    // patch pixel of (tile pixel, tap (0, 0))
#pragma unroll
  for (int t = 0; t < TM; ++t) {
    const int pix = wm * WMS + t * MF + li;
    const int r = pix / W_, c = pix - r * W_;
    const int sg = r / SEG, rr = r - sg * SEG;
    qb[t] = (sg * (SEG + 2) + rr) * PW + c;
  }
  // fragment address of (row q, K step kk) = (q * 128 + (h_pos(chunk(0, lg), q) << 4)) ^ (kk << 5) in both flavours
  auto frag_off = [&](int q) { return q * 128 + (h_pos<MF>(lg, h_sw_patch<MF, W_>(q)) << 4); };                      // patch rows
  const int a_off0 = (wn * WNS + li) * 128 + (h_pos<MF>(lg, h_sw_rows<MF>(wn * WNS + li)) << 4);                    // + t * MF * 128 for weight tile t ((row >> 1) & 7 depends on li only)

  // The residual enters through the ACCUMULATORS (fragment layout), a few tiles per K step during the first NRS
  // steps: requested at the top of step s, added at the top of step s + 1 - its latency runs under MFMA work instead
  // of in the epilogue (one dependent global load per copied row was most of the fixed cost of a residual
  // convolution) or in front of the first step.  residual + sum of products, fp32.
  // (split-K: the slabs carry plain partial sums, the reduction kernel adds the epilogue terms)
  auto cout_of = [&](int tn, int rq) { return n0 + wn * WNS + tn * MF + (MF == 16 ? 4 * lg : 8 * rq + 4 * lg); };
  AccT acc[TN][TM];
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b)
#pragma unroll
      for (int e = 0; e < NACC; ++e) acc[a][b][e] = 0.f;
  constexpr int NRT = TN * TM * RQ;                              // residual quads per lane
  constexpr int NRS = TPS == 1 ? 8 : 3;                          // steps they are spread over (G >= 9 / 3)
  constexpr int RCH = (NRT + NRS - 1) / NRS;
  typedef __attribute__((ext_vector_type(4))) T Quad;
  Quad rv[RCH];
  const bool use_res = p.residual && p.splitk == 1 && !(p.dbg & 16);
  // (fp32 on the 8 + 4 wave tiles: 16-byte residual quads do not fit next to 96 accumulators at 3 waves per SIMD -
  //  there the accumulators simply start from the residual, loaded in front of the first step)
  constexpr bool RSPREAD = !(sizeof(T) == 4 && NWC >= 8);
  if (!RSPREAD && use_res) {
    const T* res = (const T*)p.residual;
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
      for (int b = 0; b < TM; ++b)
#pragma unroll
        for (int rq = 0; rq < RQ; ++rq) {
          float r0, r1, r2, r3;
          load4<T>(res + (size_t)gpix(wm * WMS + b * MF + li) * p.res_ld + cout_of(a, rq), r0, r1, r2, r3);
          acc[a][b][4 * rq] = r0; acc[a][b][4 * rq + 1] = r1; acc[a][b][4 * rq + 2] = r2; acc[a][b][4 * rq + 3] = r3;
        }
  }
  auto res_issue = [&](int chunk) {
    const T* res = (const T*)p.residual;
#pragma unroll
    for (int k = 0; k < RCH; ++k) {
      const int idx = chunk * RCH + k;
      if (idx < NRT) {
        const int a = idx / (TM * RQ), b = (idx / RQ) % TM, rq = idx % RQ;
        rv[k] = *reinterpret_cast<const Quad*>(res + (size_t)gpix(wm * WMS + b * MF + li) * p.res_ld + cout_of(a, rq));
      }
    }
  };
  auto res_add = [&](int chunk) {
#pragma unroll
    for (int k = 0; k < RCH; ++k) {
      const int idx = chunk * RCH + k;
      if (idx < NRT) {
        const int a = idx / (TM * RQ), b = (idx / RQ) % TM, rq = idx % RQ;
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[a][b][4 * rq + e] += to_f32(rv[k][e]);
      }
    }
  };

  // TPS == 3 (the small tiles): the fragment offsets of all nine taps are computed ONCE (9 x TM registers) and the
  // step loop is unrolled over the three filter rows; computed per step (as the TPS == 1 tiles do) they were ~85 VALU
  // instructions in front of every step's first LDS read.
  int btab[TPS == 3 ? 9 : 1][TM];
  if constexpr (TPS == 3) {
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int t = 0; t < TM; ++t) btab[tap][t] = frag_off(qb[t] + (tap / 3) * PW + (tap - (tap / 3) * 3));
  }
  {
    int slot = 0, st = 0, pbuf = 0;
    auto step_end = [&]() {
      slot = slot + 1 == STAGES ? 0 : slot + 1;
      if (++st == SPC) {
        st = 0;
        pbuf ^= 1;
      }
    };
    // plain form of a K step (fragments of one phase, then its MFMAs; the compiler's own schedule): used for the
    // first NRS steps of a residual convolution, where the residual chunks need the registers the hand-ordered
    // pipeline below spends on fragments in flight
    auto k_step_plain = [&](auto ST) {                 // ST: the step's index inside its channel block when static (TPS == 3), else -1
      constexpr int sst = decltype(ST)::value;
      if (!(p.dbg & 2)) {
        const char* sP = smem + pbuf * PATCH;
        const char* sW = smem + 2 * PATCH + slot * W_STAGE;
#pragma unroll
        for (int ti = 0; ti < TPS; ++ti) {
          const int tap = st * TPS + ti;
          const int tapoff = (tap / 3) * PW + (tap - (tap / 3) * 3);
          int bo[TM];
#pragma unroll
          for (int t = 0; t < TM; ++t) {
            if constexpr (sst >= 0) bo[t] = btab[sst * TPS + ti][t];
            else bo[t] = frag_off(qb[t] + tapoff);
          }
#pragma unroll
          for (int kk = 0; kk < NKK; ++kk) {
            Chunk a[TN], b[TM];
#pragma unroll
            for (int t = 0; t < TN; ++t) a[t] = ld16<Chunk>(sW + ti * W_TAP + ((a_off0 + t * MF * 128) ^ (kk << 5)));
#pragma unroll
            for (int t = 0; t < TM; ++t) b[t] = ld16<Chunk>(sP + (bo[t] ^ (kk << 5)));
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
              for (int tm = 0; tm < TM; ++tm) {
                if constexpr (MF == 16) MM::mma(acc[tn][tm], a[tn], b[tm]);
                else Mma32<T>::mma(acc[tn][tm], a[tn], b[tm]);
              }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
      step_end();
    };
    auto k_step = [&](auto ST) {
      constexpr int sst = decltype(ST)::value;
      if (!(p.dbg & 2)) {
        const char* sP = smem + pbuf * PATCH;
        const char* sW = smem + 2 * PATCH + slot * W_STAGE;
        int boff[TPS][TM];
#pragma unroll
        for (int ti = 0; ti < TPS; ++ti) {
          const int tap = st * TPS + ti;                         // TPS == 3: kh = st, kw = ti
          const int tapoff = (tap / 3) * PW + (tap - (tap / 3) * 3);
#pragma unroll
          for (int t = 0; t < TM; ++t) {
            if constexpr (sst >= 0) boff[ti][t] = btab[sst * TPS + ti][t];
            else boff[ti][t] = frag_off(qb[t] + tapoff);
          }
        }
        // hand-ordered fragment pipeline over the step's TPS x NKK phases of TN MFMA groups (one weight fragment x TM
        // pixel fragments each): weight fragment i + AD is requested when group i starts, the pixel fragments of the
        // next phase in the middle of the current one.  (Left alone the compiler rotates two weight buffers with a
        // distance of ONE group: every group then waits out an LDS round trip.)
        constexpr int NPH = TPS * NKK, NG = NPH * TN, AD = 3;
        Chunk af[NG], bf[NPH][TM];
        auto lda = [&](int i) {
          const int ph = i / TN, ti = ph / NKK, kk = ph - ti * NKK;
#ifdef AFLDM_H3_NOLOAD                                                // (timing decomposition build: MFMAs without fragment reads)
          return Chunk{};
#endif
          return ld16<Chunk>(sW + ti * W_TAP + ((a_off0 + (i - ph * TN) * MF * 128) ^ (kk << 5)));
        };
        auto ldb = [&](int ph, int t) {
          const int ti = ph / NKK, kk = ph - ti * NKK;
#ifdef AFLDM_H3_NOLOAD
          return Chunk{};
#endif
          return ld16<Chunk>(sP + (boff[ti][t] ^ (kk << 5)));
        };
#pragma unroll
        for (int t = 0; t < TM; ++t) bf[0][t] = ldb(0, t);
#pragma unroll
        for (int i = 0; i < AD; ++i) af[i] = lda(i);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NG; ++i) {
          if (i + AD < NG) af[i + AD] = lda(i + AD);
          if (i % TN == TN / 2 && i / TN + 1 < NPH) {
#pragma unroll
            for (int t = 0; t < TM; ++t) bf[i / TN + 1][t] = ldb(i / TN + 1, t);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int tm = 0; tm < TM; ++tm) {
#ifdef AFLDM_H3_NOMMA                                                 // (timing decomposition build: fragment reads without MFMAs)
            asm volatile("" ::"v"(af[i]), "v"(bf[i / TN][tm]));
            continue;
#endif
            if constexpr (MF == 16) MM::mma(acc[i % TN][tm], af[i], bf[i / TN][tm]);
            else Mma32<T>::mma(acc[i % TN][tm], af[i], bf[i / TN][tm]);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      step_end();
    };
    typedef std::integral_constant<int, -1> RT;
    int g = 0;
    if (RSPREAD && use_res) {     // (wave-uniform) the first NRS steps, peeled: static accumulator indices for the residual chunks
      auto peeled = [&](auto C) {
        constexpr int c = decltype(C)::value;
        __builtin_amdgcn_s_barrier();
        if (c > 0) res_add(c - 1);
        res_issue(c);
        if constexpr (TPS == 3) k_step_plain(C);      // NRS == SPC == 3: step c of the first channel block
        else k_step_plain(RT{});
      };
      static_assert(TPS == 1 || NRS == SPC, "peeled steps = one channel block");
      peeled(std::integral_constant<int, 0>{});
      peeled(std::integral_constant<int, 1>{});
      peeled(std::integral_constant<int, 2>{});
      if constexpr (NRS > 3) {
        peeled(std::integral_constant<int, 3>{});
        peeled(std::integral_constant<int, 4>{});
        peeled(std::integral_constant<int, 5>{});
        peeled(std::integral_constant<int, 6>{});
        peeled(std::integral_constant<int, 7>{});
      }
      static_assert(NRS == 3 || NRS == 8, "peeled steps");
      res_add(NRS - 1);
      g = NRS;
    }
    if constexpr (TPS == 3) {
      for (; g < G; g += 3) {                            // G - g is a multiple of SPC = 3
        __builtin_amdgcn_s_barrier();
        k_step(std::integral_constant<int, 0>{});
        __builtin_amdgcn_s_barrier();
        k_step(std::integral_constant<int, 1>{});
        __builtin_amdgcn_s_barrier();
        k_step(std::integral_constant<int, 2>{});
      }
    } else {
      for (; g < G; ++g) {
        __builtin_amdgcn_s_barrier();
        k_step(RT{});
      }
    }
  }

  // ------------------------------------------------------------------------------------------- epilogue
  if (p.dbg & 8) return;       // (timing decomposition only)
  if constexpr (ESZ == 2) {
    if (p.splitk == 1) {
      // bf16, whole K in this workgroup: (residual + sum) + (bias + temb) is rounded in the accumulator layout and the
      // tile is staged in bf16 - ONE pass for all BM rows (half the LDS bytes of an fp32 tile, two barriers fewer),
      // then leaves as whole rows, 16 bytes per lane; per-channel GroupNorm partial sums of the stored values
      // accumulated by the thread that owns the column (fixed order, no atomics).
      // A tile of several whole samples (NSEG > 1: the 4x4 planes) carries one (bias + temb) vector and one statistics
      // record per sample.
      constexpr int NTC = NWC * 64, EO = 8, CPR = BN / EO, RPI = NTC / CPR, OROW = BN + 8;
      constexpr int SB_BYTES = ((NSEG * BN * 4 + 1023) / 1024) * 1024, HWT = BM / NSEG;      // rows of one sample segment
      static_assert(SB_BYTES + BM * OROW * 2 <= LDS_TOTAL && RPI * BN * 8 <= LDS_TOTAL, "staging tile");
      float* sB = reinterpret_cast<float*>(smem);        // [NSEG][BN] bias + time embedding per sample of this tile
      T* sO = reinterpret_cast<T*>(smem + SB_BYTES);     // [BM][OROW]
      const T* temb = (const T*)p.temb;
      const int etid = cw * 64 + lane;
      __syncthreads();                                    // pipeline buffers idle
      for (int c = etid; c < NSEG * BN; c += NTC) {
        const int sgi = c / BN, cc = c - sgi * BN;
        float v = p.bias ? p.bias[n0 + cc] : 0.f;
        if (temb) v += to_f32(temb[(size_t)(b_tile + sgi) * p.temb_stride + (n0 + cc) % p.temb_mod]);
        sB[c] = v;
      }
      __syncthreads();
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int rq = 0; rq < RQ; ++rq) {
          const int cl = cout_of(tn, rq) - n0;
#pragma unroll
          for (int t = 0; t < TM; ++t) {
            const int row = wm * WMS + t * MF + li;
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(sB + (NSEG == 1 ? 0 : row / HWT) * BN + cl);
            Quad o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = from_f32<T>(acc[tn][t][4 * rq + e] + b4[e]);
            *reinterpret_cast<Quad*>(sO + row * OROW + cl) = o;
          }
        }
      __syncthreads();
      if constexpr (NSEG > 1) {
        // row lanes are dealt per sample segment: a thread's rows (and its partial sums) belong to ONE sample
        constexpr int RPS = RPI / NSEG;                   // row lanes per segment
        static_assert(RPS >= 1, "row lanes");
        const int ch = etid % CPR, trs = etid / CPR;
        const int sgi = trs / RPS, lr = trs - sgi * RPS;
        const bool active = trs < RPS * NSEG;
        const int n = n0 + ch * EO;
        float ss1[EO], ss2[EO];
#pragma unroll
        for (int e = 0; e < EO; ++e) ss1[e] = ss2[e] = 0.f;
        if (active) {
          for (int r = lr; r < HWT; r += RPS) {
            const int row = sgi * HWT + r;
            const Chunk o = ld16<Chunk>(sO + row * OROW + ch * EO);
            if (!(p.dbg & 4)) st16_out<Chunk>((T*)p.y + (size_t)(m0 + row) * p.y_ld + n, o);
            if (p.stats_out) {
#pragma unroll
              for (int e = 0; e < EO; ++e) {
                const float vr = to_f32(o[e]);
                ss1[e] += vr;
                ss2[e] = fmaf(vr, vr, ss2[e]);
              }
            }
          }
        }
        if (p.stats_out) {
          float* sR = reinterpret_cast<float*>(smem);     // [NSEG * RPS][BN][2]
          __syncthreads();
          if (active) {
#pragma unroll
            for (int e = 0; e < EO; ++e) *reinterpret_cast<f32x2*>(sR + ((trs * BN) + ch * EO + e) * 2) = f32x2{ss1[e], ss2[e]};
          }
          __syncthreads();
          for (int c = etid; c < NSEG * BN; c += NTC) {
            const int sg2 = c / BN, cc = c - sg2 * BN;
            float a1 = 0.f, a2 = 0.f;
            for (int r = 0; r < RPS; ++r) {
              const f32x2 v = *reinterpret_cast<const f32x2*>(sR + (((sg2 * RPS + r) * BN) + cc) * 2);
              a1 += v[0];
              a2 += v[1];
            }
            *reinterpret_cast<f32x2*>(p.stats_out + (((size_t)(b_tile + sg2) * p.stats_S) * p.Cout + n0 + cc) * 2) = f32x2{a1, a2};   // S = 1
          }
        }
        return;
      }
      const bool active = etid < RPI * CPR;
      const int ch = etid % CPR, tr = etid / CPR;
      const int n = n0 + ch * EO;
      float ss1[EO], ss2[EO];
#pragma unroll
      for (int e = 0; e < EO; ++e) ss1[e] = ss2[e] = 0.f;
      if (active) {
#pragma unroll 4
        for (int row = tr; row < BM; row += RPI) {
          const Chunk o = ld16<Chunk>(sO + row * OROW + ch * EO);
          if (!(p.dbg & 4)) st16_out<Chunk>((T*)p.y + (size_t)gpix(row) * p.y_ld + n, o);      // write-through (build.py: AFLDM_WT)
          if (p.stats_out) {
#pragma unroll
            for (int e = 0; e < EO; ++e) {
              const float vr = to_f32(o[e]);
              ss1[e] += vr;
              ss2[e] = fmaf(vr, vr, ss2[e]);
            }
          }
        }
      }
      if (p.stats_out) {
        float* sR = reinterpret_cast<float*>(smem);       // [RPI][BN][2]
        __syncthreads();
        if (active) {
#pragma unroll
          for (int e = 0; e < EO; ++e) *reinterpret_cast<f32x2*>(sR + ((tr * BN) + ch * EO + e) * 2) = f32x2{ss1[e], ss2[e]};
        }
        __syncthreads();
        for (int c = etid; c < BN; c += NTC) {
          float a1 = 0.f, a2 = 0.f;
          for (int r = 0; r < RPI; ++r) {
            const f32x2 v = *reinterpret_cast<const f32x2*>(sR + ((r * BN) + c) * 2);
            a1 += v[0];
            a2 += v[1];
          }
          *reinterpret_cast<f32x2*>(p.stats_out + (((size_t)b_tile * p.stats_S + sp_tile) * p.Cout + n0 + c) * 2) = f32x2{a1, a2};
        }
      }
      return;
    }
  }
  {
    constexpr int SROW = BN + 8;                       // fp32 row stride: conflict-free 16-byte writes
    constexpr int PROWS = BM < 128 ? BM : 128;         // rows staged per pass
    constexpr int PASSES = BM / PROWS;
    constexpr int WPP = WGM / PASSES;                  // consumer wave rows (wm) per pass
    static_assert(PROWS * SROW * 4 <= LDS_TOTAL && WGM % PASSES == 0 && WPP * WMS == PROWS, "staging tile");
    constexpr int NTC = NWC * 64;
    constexpr int EO = 16 / ESZ;                       // output elements per 16-byte store
    constexpr int CPR = BN / EO;                       // 16-byte chunks per row
    constexpr int RPI = NTC / CPR;                     // rows the workgroup covers per sweep
    static_assert(RPI * BN * 2 * 4 <= LDS_TOTAL, "statistics scratch");
    float* sC = reinterpret_cast<float*>(smem);
    const T* temb = (const T*)p.temb;
    const int etid = cw * 64 + lane;
    const bool active = etid < RPI * CPR;
    const int ch = etid % CPR, tr = etid / CPR;
    const int n = n0 + ch * EO;
    // bias + time embedding of the thread's column (one sample per tile): added as ONE vector, acc + (bias + temb)
    float bvec[EO], ss1[EO], ss2[EO];
#pragma unroll
    for (int e = 0; e < EO; ++e) bvec[e] = ss1[e] = ss2[e] = 0.f;
    if (active && p.splitk == 1) {
      if (p.bias) {
#pragma unroll
        for (int q = 0; q < EO / 4; ++q) {
          const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + n + 4 * q);
          bvec[4 * q] = bv[0]; bvec[4 * q + 1] = bv[1]; bvec[4 * q + 2] = bv[2]; bvec[4 * q + 3] = bv[3];
        }
      }
      if (temb) {
        const Chunk tv = ld16<Chunk>(temb + (size_t)b_tile * p.temb_stride + n % p.temb_mod);
#pragma unroll
        for (int e = 0; e < EO; ++e) bvec[e] += to_f32(tv[e]);
      }
    }
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
      __syncthreads();   // pipeline buffers idle (first pass) / previous pass copied out
      if (wm / WPP == ps) {
        const int wml = wm - ps * WPP;
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
          for (int t = 0; t < TM; ++t) {
            const int row = wml * WMS + t * MF + li;
#pragma unroll
            for (int rq = 0; rq < RQ; ++rq) {
              const f32x4 v = f32x4{acc[tn][t][4 * rq], acc[tn][t][4 * rq + 1], acc[tn][t][4 * rq + 2], acc[tn][t][4 * rq + 3]};
              *reinterpret_cast<f32x4*>(sC + row * SROW + (cout_of(tn, rq) - n0)) = v;
            }
            __builtin_amdgcn_sched_barrier(0);
          }
      }
      __syncthreads();
      if (p.splitk > 1) {
        // split-K slab: fp32 rows, 16 bytes per lane
        constexpr int QPR = BN / 4;
#pragma unroll 1
        for (int i = etid; i < PROWS * QPR; i += NTC) {
          const int row = i / QPR, q = i - row * QPR;
          const f32x4 a = *reinterpret_cast<const f32x4*>(sC + row * SROW + 4 * q);
          *reinterpret_cast<f32x4*>(p.ws + ((size_t)ks * p.M + gpix(ps * PROWS + row)) * p.Cout + n0 + 4 * q) = a;
        }
        continue;
      }
      if (active) {
#pragma unroll 2
        for (int row = tr; row < PROWS; row += RPI) {
          const int m = gpix(ps * PROWS + row);
          float v[EO];
#pragma unroll
          for (int q = 0; q < EO / 4; ++q) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(sC + row * SROW + ch * EO + 4 * q);
            v[4 * q] = a[0]; v[4 * q + 1] = a[1]; v[4 * q + 2] = a[2]; v[4 * q + 3] = a[3];
          }
          Chunk o;
#pragma unroll
          for (int e = 0; e < EO; ++e) o[e] = from_f32<T>(v[e] + bvec[e]);       // (residual + sum) + (bias + temb)
          if (!(p.dbg & 4)) st16_out<Chunk>((T*)p.y + (size_t)m * p.y_ld + n, o);
          if (p.stats_out) {
#pragma unroll
            for (int e = 0; e < EO; ++e) {
              const float vr = to_f32(o[e]);      // statistics of what the consumer will read
              ss1[e] += vr;
              ss2[e] = fmaf(vr, vr, ss2[e]);
            }
          }
        }
      }
    }
    if (p.stats_out && p.splitk == 1) {
      // per-channel sums of this tile's BM rows (one split of one sample): the RPI row-interleaved partials of a
      // column are added in a fixed order through LDS
      float* sR = sC;   // [RPI][BN][2]
      __syncthreads();
      if (active) {
#pragma unroll
        for (int e = 0; e < EO; ++e) *reinterpret_cast<f32x2*>(sR + ((tr * BN) + ch * EO + e) * 2) = f32x2{ss1[e], ss2[e]};
      }
      __syncthreads();
      for (int c = etid; c < BN; c += NTC) {
        float a1 = 0.f, a2 = 0.f;
        for (int r = 0; r < RPI; ++r) {
          const f32x2 v = *reinterpret_cast<const f32x2*>(sR + ((r * BN) + c) * 2);
          a1 += v[0];
          a2 += v[1];
        }
        *reinterpret_cast<f32x2*>(p.stats_out + (((size_t)b_tile * p.stats_S + sp_tile) * p.Cout + n0 + c) * 2) = f32x2{a1, a2};
      }
    }
  }
}

// ----------------------------------------------------------------------------------------------- persistent tiles
// k_conv3h_pers (round 4): the halo-patch kernel with PERSISTENT workgroups for layers that give a CU several tiles (the
// AF-VAE's 64^2 .. 256^2 planes: 4 .. 64 tiles of 256 x 128 per CU; 128 -> 128 channels are only 18 K steps per tile, next
// to ~16 us of launch + first-patch latency + epilogue per one-tile workgroup).  One workgroup per CU walks tiles
// l, l + grid, ...; the LDS-DMA producers never stop at a tile boundary:
//   * the weight ring simply continues into the next tile's first steps (a workgroup keeps its n tile);
//   * the two patch buffers alternate by GLOBAL channel-block count, so the next tile's first patch goes into the buffer
//     the last-but-one block of this tile has left, 9 K steps before the tile ends;
//   * the epilogue stages the output tile through the LAST block's patch buffer only (two passes of 128 rows), which no
//     DMA touches before the consumers have passed the next tile's first barrier.
// The consumers therefore find their operands waiting when they come out of an epilogue, and an epilogue's stores drain
// under the next tile's K loop.  bf16, 16x16x32 MFMAs, one tap per step, whole K per workgroup (no split-K); same
// rounding points, statistics records and tile shapes as k_conv3h.
template <int BM, int W_, int BN, int WGM, int WGN, bool SUB>
__global__ void __launch_bounds__((WGM * WGN + 4) * 64, (WGM * WGN + 4 + 3) / 4) k_conv3h_pers(ConvP p) {
  typedef bf16 T;
  typedef Mma<T> MM;
  typedef typename MM::Chunk Chunk;
  constexpr int MF = 16, NPROD = 4, STAGES = 3;
  constexpr int NWC = WGM * WGN;
  constexpr int EPC = MM::EPC, KSTEP = 8 * EPC, ESZ = 2;
  constexpr int WMS = BM / WGM, WNS = BN / WGN, TM = WMS / MF, TN = WNS / MF;
  constexpr int ROWS = BM / W_;
  constexpr int PW = W_ + 2, PR = ROWS + 2, NPQ = PR * PW, NPI = (NPQ + 7) / 8;
  constexpr int PATCH = NPI * 1024;
  constexpr int SPC = 9;
  constexpr int WIT = BN / 8, WPW = WIT / NPROD, PPW = (NPI + NPROD - 1) / NPROD;
  constexpr int W_STAGE = BN * 128;
  constexpr unsigned OOB = 0x80000000u;
  static_assert(BM % W_ == 0 && WIT % NPROD == 0 && WMS % MF == 0 && WNS % MF == 0 && ROWS <= W_, "tile shape");
  static_assert((STAGES - 2) * WPW + PPW < 64, "vmcnt is a 6-bit counter");
  // epilogue staging (inside ONE patch buffer): [bias + temb: 1 KB][PROWS rows x (BN + 8) bf16]
  constexpr int OROW = BN + 8, SB_BYTES = 1024;
  constexpr int PROWS = WMS * (((PATCH - SB_BYTES) / (OROW * 2)) / WMS >= WGM ? WGM : ((PATCH - SB_BYTES) / (OROW * 2)) / WMS);
  constexpr int PASSES = BM / PROWS, WPP = WGM / PASSES;
  static_assert(PROWS >= WMS && BM % PROWS == 0 && WGM % PASSES == 0 && BN * 4 <= SB_BYTES, "staging passes");
  constexpr int NTC = NWC * 64, EO = 8, CPR = BN / EO, RPI = NTC / CPR;
  static_assert(RPI * BN * 8 <= PATCH, "statistics scratch inside a patch buffer");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool is_producer = wave_all >= NWC;
  const int li = lane & 15, lg = lane >> 4;

  const int tiles_m = p.M / BM, tiles = tiles_m * p.tiles_n;
  const int l = xcd_remap(blockIdx.x, gridDim.x);               // neighbouring tiles run at the same time on one XCD
  const int nmine = l < tiles ? (tiles - l + (int)gridDim.x - 1) / (int)gridDim.x : 0;
  const int tile_n = l % p.tiles_n;                             // (gridDim.x is a multiple of tiles_n: constant per workgroup)
  const int n0 = tile_n * BN;
  const int Ct = p.C1, HW = SUB ? p.H * p.W : W_ * W_;
  const int ncb = Ct / KSTEP, G = ncb * SPC;
  const bool has_stats = p.stats_out != nullptr;
  const int nepb = 1 + 2 * PASSES + (has_stats ? 2 : 0);        // workgroup barriers of one epilogue (both roles count them)

  struct TileGeo { int m0, b, oh0, ow0, sp; };
  auto geo = [&](int i) {                                        // i-th tile of this workgroup
    const int tile = l + i * (int)gridDim.x, tm = tile / p.tiles_n;
    TileGeo g;
    if constexpr (SUB) {
      const int tw = p.W / W_, tps = (p.H / ROWS) * tw;
      g.b = tm / tps;
      g.sp = tm - g.b * tps;
      const int ty = g.sp / tw;
      g.oh0 = ty * ROWS;
      g.ow0 = (g.sp - ty * tw) * W_;
      g.m0 = g.b * HW + g.oh0 * p.W + g.ow0;
    } else {
      g.m0 = tm * BM;
      g.b = g.m0 / HW;
      g.oh0 = (g.m0 - g.b * HW) / W_;
      g.ow0 = 0;
      g.sp = (g.m0 - g.b * HW) / BM;
    }
    return g;
  };

  if (is_producer) {
    const int wave = wave_all - NWC;
    __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x1, 0, (int)((long long)p.M * Ct * ESZ), 0x00020000);
    __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)((long long)p.Cout * 9 * Ct * ESZ), 0x00020000);
    unsigned woff[WPW];
#pragma unroll
    for (int i = 0; i < WPW; ++i) {
      const int j = wave + NPROD * i;
      const int r = 8 * j + (lane >> 3);
      const int c = h_chunk_at<MF>(lane & 7, h_sw_rows<MF>(r));
      woff[i] = ((unsigned)(n0 + r) * 9u * (unsigned)Ct + (unsigned)(c * EPC)) * ESZ;
    }
    // patch DMA of the i-th tile's channel block cb into patch buffer `buf`
    auto issue_patch = [&](int i, int cb, int buf) {
      const TileGeo g = geo(i);
      char* sbase = smem + buf * PATCH;
      const unsigned so = (unsigned)(cb * KSTEP * ESZ);
#pragma unroll
      for (int k = 0; k < PPW; ++k) {
        int j = wave + NPROD * k;
        if (j > NPI - 1) j = NPI - 1;                              // (short shares re-issue the last instruction: uniform vmcnt counts)
        const int q = 8 * j + (lane >> 3);
        const int c = h_chunk_at<MF>(lane & 7, h_sw_patch<MF, W_>(q));
        const int pr = q / PW, pc = q - pr * PW;
        const int ih = g.oh0 + pr - 1;
        bool ok;
        int pixel;
        if constexpr (SUB) {
          const int iw = g.ow0 + pc - 1;
          ok = q < NPQ && iw >= 0 && iw < p.W && ih >= 0 && ih < p.H;
          pixel = g.b * HW + ih * p.W + iw;
        } else {
          ok = q < NPQ && pc >= 1 && pc <= W_ && ih >= 0 && ih < W_;
          pixel = g.m0 + (pr - 1) * W_ + (pc - 1);
        }
        const unsigned off = ok ? ((unsigned)pixel * (unsigned)Ct + (unsigned)(c * EPC)) * ESZ : OOB;
        lds_ptr_t dst = (lds_ptr_t)(sbase + j * 1024);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, dst, 16, (int)off, (int)so, 0, 0);
      }
    };
    // weight cursor: the step to issue next, counted across tiles
    int ws_tile = 0, ws_st = 0, ws_cb = 0, ws_slot = 0;
    auto issue_weights = [&]() {
      char* sbase = smem + 2 * PATCH + ws_slot * W_STAGE;
      const unsigned so = ws_tile < nmine ? (unsigned)((ws_st * Ct + ws_cb * KSTEP) * ESZ) : OOB;
#pragma unroll
      for (int i = 0; i < WPW; ++i) {
        lds_ptr_t dst = (lds_ptr_t)(sbase + (wave + NPROD * i) * 1024);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, dst, 16, (int)woff[i], (int)so, 0, 0);
      }
      if (++ws_st == SPC) {
        ws_st = 0;
        if (++ws_cb == ncb) {
          ws_cb = 0;
          ++ws_tile;
        }
      }
      ws_slot = ws_slot + 1 == STAGES ? 0 : ws_slot + 1;
    };
    if (nmine > 0) issue_patch(0, 0, 0);
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s) issue_weights();
    int since = STAGES, gcb = 0;                                 // K steps since the last patch issue; global channel-block count
    for (int i = 0; i < nmine; ++i) {
      for (int cb = 0; cb < ncb; ++cb, ++gcb) {
        for (int st = 0; st < SPC; ++st) {
          if (since <= STAGES - 1) wait_vmcnt<(STAGES - 2) * WPW + PPW>();
          else wait_vmcnt<(STAGES - 2) * WPW>();
          __builtin_amdgcn_s_barrier();
          ++since;
          issue_weights();
          if (st == 0) {                                           // the next block's patch: of this tile, or the next tile's first
            if (cb + 1 < ncb) {
              issue_patch(i, cb + 1, (gcb + 1) & 1);
              since = 1;
            } else if (i + 1 < nmine) {
              issue_patch(i + 1, 0, (gcb + 1) & 1);
              since = 1;
            }
          }
        }
      }
      for (int e = 0; e < nepb; ++e) __builtin_amdgcn_s_barrier();   // the consumers' epilogue
    }
    wait_vmcnt<0>();
    return;
  }

  // ------------------------------------------------------------------------------------------- consumers
  const int cw = wave_all, wm = cw / WGN, wn = cw - wm * WGN;
  constexpr int NKK = 2;
  int qb[TM];
#pragma unroll
  for (int t = 0; t < TM; ++t) {
    const int pix = wm * WMS + t * MF + li;
    const int r = pix / W_, c = pix - r * W_;
    qb[t] = r * PW + c;
  }
  auto frag_off = [&](int q) { return q * 128 + (h_pos<MF>(lg, h_sw_patch<MF, W_>(q)) << 4); };
  const int a_off0 = (wn * WNS + li) * 128 + (h_pos<MF>(lg, h_sw_rows<MF>(wn * WNS + li)) << 4);
  auto cout_of = [&](int tn) { return n0 + wn * WNS + tn * MF + 4 * lg; };
  typedef __attribute__((ext_vector_type(4))) T Quad;
  const T* temb = (const T*)p.temb;
  const int etid = cw * 64 + lane;
  int slot = 0, gcb = 0;

  for (int i = 0; i < nmine; ++i) {
    const TileGeo tg = geo(i);
    auto gpix = [&](int tp) -> int {
      if constexpr (SUB) return tg.m0 + (tp / W_) * p.W + (tp % W_);
      else return tg.m0 + tp;
    };
    f32x4 acc[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
      for (int b = 0; b < TM; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    // residual through the accumulators during the first steps (fragment layout), as in k_conv3h
    constexpr int NRT = TN * TM, NRS = 8, RCH = (NRT + NRS - 1) / NRS;
    Quad rv[RCH];
    const bool use_res = p.residual != nullptr;
    auto res_issue = [&](int chunk) {
      const T* res = (const T*)p.residual;
#pragma unroll
      for (int k = 0; k < RCH; ++k) {
        const int idx = chunk * RCH + k;
        if (idx < NRT) {
          const int a = idx / TM, b = idx % TM;
          rv[k] = *reinterpret_cast<const Quad*>(res + (size_t)gpix(wm * WMS + b * MF + li) * p.res_ld + cout_of(a));
        }
      }
    };
    auto res_add = [&](int chunk) {
#pragma unroll
      for (int k = 0; k < RCH; ++k) {
        const int idx = chunk * RCH + k;
        if (idx < NRT) {
          const int a = idx / TM, b = idx % TM;
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[a][b][e] += to_f32(rv[k][e]);
        }
      }
    };
    auto k_step = [&](int st, int pbuf) {
      const char* sP = smem + pbuf * PATCH;
      const char* sW = smem + 2 * PATCH + slot * W_STAGE;
      const int tapoff = (st / 3) * PW + (st - (st / 3) * 3);
      int boff[TM];
#pragma unroll
      for (int t = 0; t < TM; ++t) boff[t] = frag_off(qb[t] + tapoff);
      constexpr int NG = NKK * TN, AD = 3;
      Chunk af[NG], bf[NKK][TM];
      auto lda = [&](int ii) { return ld16<Chunk>(sW + ((a_off0 + (ii % TN) * MF * 128) ^ ((ii / TN) << 5))); };
      auto ldb = [&](int kk, int t) { return ld16<Chunk>(sP + (boff[t] ^ (kk << 5))); };
#pragma unroll
      for (int t = 0; t < TM; ++t) bf[0][t] = ldb(0, t);
#pragma unroll
      for (int ii = 0; ii < AD; ++ii) af[ii] = lda(ii);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ii = 0; ii < NG; ++ii) {
        if (ii + AD < NG) af[ii + AD] = lda(ii + AD);
        if (ii % TN == TN / 2 && ii / TN + 1 < NKK) {
#pragma unroll
          for (int t = 0; t < TM; ++t) bf[ii / TN + 1][t] = ldb(ii / TN + 1, t);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) MM::mma(acc[ii % TN][tm], af[ii], bf[ii / TN][tm]);
        __builtin_amdgcn_sched_barrier(0);
      }
      slot = slot + 1 == STAGES ? 0 : slot + 1;
    };
    {
      int g = 0;
      for (int cb = 0; cb < ncb; ++cb, ++gcb)
        for (int st = 0; st < SPC; ++st, ++g) {
          __builtin_amdgcn_s_barrier();
          if (use_res && g <= NRS) {                              // (wave-uniform)
            if (g > 0) {
              // chunk g - 1 was requested one step ago
              switch (g - 1) {
                case 0: res_add(0); break;
                case 1: res_add(1); break;
                case 2: res_add(2); break;
                case 3: res_add(3); break;
                case 4: res_add(4); break;
                case 5: res_add(5); break;
                case 6: res_add(6); break;
                default: res_add(7); break;
              }
            }
            if (g < NRS) res_issue(g);
          }
          k_step(st, gcb & 1);
        }
    }
    // ---- epilogue: staged through the LAST block's patch buffer, PASSES passes of PROWS rows
    char* sE = smem + ((gcb - 1) & 1) * PATCH;
    float* sB = reinterpret_cast<float*>(sE);
    T* sO = reinterpret_cast<T*>(sE + SB_BYTES);
#define H3P_SYNC()                                  \
  do {                                              \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
    __builtin_amdgcn_s_barrier();                   \
    asm volatile("" ::: "memory");                   \
  } while (0)
    // (workgroup barriers of the epilogue: the wave's LDS traffic must have completed, its output STORES must not be waited
    //  for - __syncthreads() would drain vmcnt and serialise the store tail the next tile's K loop is meant to cover)
    H3P_SYNC();                                                  // (1) every consumer has left the last K step
    for (int c = etid; c < BN; c += NTC) {
      float v = p.bias ? p.bias[n0 + c] : 0.f;
      if (temb) v += to_f32(temb[(size_t)tg.b * p.temb_stride + (n0 + c) % p.temb_mod]);
      sB[c] = v;
    }
    const bool active = etid < RPI * CPR;
    const int ch = etid % CPR, tr = etid / CPR;
    const int n = n0 + ch * EO;
    float ss1[EO], ss2[EO];
#pragma unroll
    for (int e = 0; e < EO; ++e) ss1[e] = ss2[e] = 0.f;
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
      H3P_SYNC();                                                // bias vector written / previous pass copied out
      if (wm / WPP == ps) {
        const int wml = wm - ps * WPP;
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
          const int cl = cout_of(tn) - n0;
          const f32x4 b4 = *reinterpret_cast<const f32x4*>(sB + cl);
#pragma unroll
          for (int t = 0; t < TM; ++t) {
            const int row = wml * WMS + t * MF + li;
            Quad o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = from_f32<T>(acc[tn][t][e] + b4[e]);
            *reinterpret_cast<Quad*>(sO + row * OROW + cl) = o;
          }
        }
      }
      H3P_SYNC();
      if (active) {
        // a thread keeps the rows r = tr (mod RPI) of the WHOLE tile, pass by pass: the same rows in the same order as
        // k_conv3h's single sweep -> bit-identical statistics
        const int first = ((tr - ps * PROWS) % RPI + RPI) % RPI;
#pragma unroll 4
        for (int row = first; row < PROWS; row += RPI) {
          const Chunk o = ld16<Chunk>(sO + row * OROW + ch * EO);
          st16_out<Chunk>((T*)p.y + (size_t)gpix(ps * PROWS + row) * p.y_ld + n, o);
          if (has_stats) {
#pragma unroll
            for (int e = 0; e < EO; ++e) {
              const float vr = to_f32(o[e]);
              ss1[e] += vr;
              ss2[e] = fmaf(vr, vr, ss2[e]);
            }
          }
        }
      }
    }
    if (has_stats) {
      float* sR = reinterpret_cast<float*>(sE);                  // [RPI][BN][2]
      H3P_SYNC();
      if (active) {
#pragma unroll
        for (int e = 0; e < EO; ++e) *reinterpret_cast<f32x2*>(sR + ((tr * BN) + ch * EO + e) * 2) = f32x2{ss1[e], ss2[e]};
      }
      H3P_SYNC();
      for (int c = etid; c < BN; c += NTC) {
        float a1 = 0.f, a2 = 0.f;
        for (int r = 0; r < RPI; ++r) {
          const f32x2 v = *reinterpret_cast<const f32x2*>(sR + ((r * BN) + c) * 2);
          a1 += v[0];
          a2 += v[1];
        }
        *reinterpret_cast<f32x2*>(p.stats_out + (((size_t)tg.b * p.stats_S + tg.sp) * p.Cout + n0 + c) * 2) = f32x2{a1, a2};
      }
    }
  }
}

// ----------------------------------------------------------------------------------------------- host side
struct H3Variant {
  int bm, w, bn, wgm, wgn, mf, tps;
  int sub;     // 1: a tile is a (bm / w) x w block of a LARGER plane (p.W % w == 0, p.H % (bm / w) == 0)
};
// ids kConv3hFirst + index
static const H3Variant kH3[] = {
    {256, 32, 192, 4, 2, 16, 1},   // 41: 32x32 planes, 8 rows per tile; 8 consumer waves (64 x 96 each) + 4 producers
    {128, 16, 192, 2, 4, 16, 1},   // 42: 16x16 planes, 8 rows per tile; 8 consumers (64 x 48) + 4 producers
    {128, 16, 192, 2, 2, 16, 1},   // 43: 16x16 planes; 4 consumers (64 x 96) + 4 producers
    {256, 16, 192, 4, 2, 16, 1},   // 44: 16x16 planes, whole sample per tile
    {128, 32, 192, 2, 4, 16, 1},   // 45: 32x32 planes, 4 rows per tile
    {128, 32, 192, 2, 2, 16, 1},   // 46
    {256, 32, 192, 4, 2, 32, 1},   // 47: as 41 on 32x32x16 MFMAs
    {128, 16, 192, 2, 2, 32, 1},   // 48: as 43 on 32x32x16 MFMAs
    {256, 16, 192, 4, 2, 32, 1},   // 49: as 44 on 32x32x16 MFMAs
    {128, 32, 192, 2, 2, 32, 1},   // 50: as 46 on 32x32x16 MFMAs
    {64, 8, 96, 2, 2, 16, 3},      // 51: 8x8 planes, one sample x 96 couts per tile, 3 taps per step, no split-K
    {64, 4, 96, 2, 2, 16, 3},      // 52: 4x4 planes, four samples x 96 couts per tile, 3 taps per step, split-K 2
    {64, 8, 96, 2, 2, 16, 3},      // 53: (= 51; a 192-cout tile with 3 taps per step does not fit the LDS)
    {128, 16, 96, 2, 2, 16, 3},    // 54: 16x16 planes, 8 rows x 96 couts, 3 taps per step (small batches)
    {128, 32, 96, 2, 2, 16, 3},    // 55: 32x32 planes, 4 rows x 96 couts (small batches)
    {0, 0, 0, 0, 0, 0, 0},         // 56: (implicit-GEMM variant, conv.hip)
    {256, 32, 192, 4, 2, 16, 1},   // 57: as 41 with a 2-deep weight ring (lookahead experiment)
    {256, 32, 128, 4, 2, 16, 1, 1},   // 58: planes of 64^2 and up (AF-VAE): 8 x 32 pixel blocks x 128 couts, 8 + 4 waves
    {256, 32, 128, 2, 2, 16, 1, 1},   // 59: as 58 with 4 consumer waves (128 x 64 each) + 4 producers
    {256, 32, 128, 4, 2, 32, 1, 1},   // 60: as 58 on 32x32x16 MFMAs
    // Two workgroups in flight per CU: 128 x 96 tiles, 4 consumer waves (64 x 48) + 2 producers, 2-deep
    // weight ring: 72 - 78 KB of LDS and <= 168 VGPRs, so that one workgroup's prologue / epilogue runs under the other's K loop
    {128, 32, 96, 2, 2, 16, 1},       // 61: 32x32 planes, 4 rows x 96 couts
    {128, 16, 96, 2, 2, 16, 1},       // 62: 16x16 planes, 8 rows x 96 couts
    // Persistent workgroups (k_conv3h_pers, round 4): one workgroup per CU walks several tiles, the LDS-DMA producers run on
    // across tile boundaries (bf16, whole K per workgroup)
    {256, 32, 128, 4, 2, 16, 1, 1},   // 63: as 58 (AF-VAE planes of 64^2 and up)
    {128, 32, 192, 2, 2, 16, 1},      // 64: 32x32 planes, 4 rows per tile: two tiles per CU at batch 64
    // Small batches (round 4): 64-pixel tiles, so that batch 8 still gives every CU a workgroup at the 32^2 / 16^2 levels
    {64, 32, 96, 2, 2, 16, 3},        // 65: 32x32 planes, 2 rows x 96 couts, 3 taps per step
    {64, 16, 96, 2, 2, 16, 3},        // 66: 16x16 planes, 4 rows x 96 couts
};
constexpr int kNumH3 = (int)(sizeof(kH3) / sizeof(kH3[0]));

int conv3h_tile(int variant, int* bm, int* bn) {
  const int k = variant - kConv3hFirst;
  if (k < 0 || k >= kNumH3) return 0;
  *bm = kH3[k].bm;
  *bn = kH3[k].bn;
  return 1;
}

bool conv3h_supported(int variant, int dtype_size, const ConvP& p) {
  const int k = variant - kConv3hFirst;
  if (k < 0 || k >= kNumH3) return false;
  const H3Variant& v = kH3[k];
  if (v.bm == 0) return false;
  const int kstep = 128 / dtype_size;
  const int eo = 16 / dtype_size;
  const int HW = p.H * p.W;
  const int z = p.splitk > 0 ? p.splitk : 1;
  const bool tile_ok = HW % v.bm == 0 || (v.bm % HW == 0 && (z > 1 || dtype_size == 2));    // several samples per tile: as split-K slabs, or (bf16) through the per-sample epilogue
  const bool plane_ok = v.sub ? (p.W > v.w && p.W % v.w == 0 && p.H % (v.bm / v.w) == 0 && z == 1) : (p.W == v.w && p.H == p.W && tile_ok);
  if ((k == 22 || k == 23) && !(dtype_size == 2 && z == 1 && HW % v.bm == 0 && p.C1 / kstep >= 2 && (!p.temb || p.temb_mod > 0))) return false;   // persistent variants
  if (k >= 24 && dtype_size != 2) return false;                                      // 65 / 66: bf16 instantiations only
  return p.KS == 3 && p.C2 == 0 && plane_ok && p.M % v.bm == 0 && p.Cout % v.bn == 0 &&
         p.C1 % kstep == 0 && (p.C1 / kstep) % z == 0 && p.out_mode == 0 && !p.y2 && p.y_ld % eo == 0 &&
         (!p.residual || p.res_ld % eo == 0) && (!p.temb || (p.temb_stride % eo == 0 && p.temb_mod % eo == 0)) &&
         (long long)p.M * p.C1 * dtype_size < (1ll << 31) && (long long)p.Cout * 9 * p.C1 * dtype_size < (1ll << 31) &&
         p.Cout % 4 == 0 && aligned16(p.y) && aligned16(p.x1) && aligned16(p.w);
}

template <typename T, int BM, int W_, int BN, int WGM, int WGN, int MF, int TPS, int STAGES = 3, bool SUB = false, int NPROD = 4, int MINW_ = 0>
static void launch_h3(const ConvP& p0, hipStream_t st) {
  constexpr int NWC = WGM * WGN;
  constexpr int MINW = MINW_ > 0 ? MINW_ : (NWC + NPROD + 3) / 4;
  constexpr int ROWS = BM / W_, SEG = ROWS < W_ ? ROWS : W_, NSEG = ROWS / SEG;
  constexpr int NPI = (NSEG * (SEG + 2) * (W_ + 2) + 7) / 8;
  constexpr int lds = 2 * NPI * 1024 + STAGES * TPS * BN * 128;
  static_assert(lds <= 160 * 1024, "LDS");
  ConvP p = p0;
  p.tiles_n = p.Cout / BN;
  if (p.splitk < 1) p.splitk = 1;
  const int tiles_m = p.M / BM, tiles = tiles_m * p.tiles_n;
  {
    // XCD grid (see the kernel): bytes fetched beyond the L2s ~ pixels * gn + weights * (8 / gn)
    static const int s_gn = getenv("AFLDM_CONV3H_GN") ? atoi(getenv("AFLDM_CONV3H_GN")) : -1;
    const double xb = (double)p.M * p.C1, wb = 9.0 * p.Cout * p.C1;
    int best = 0;
    double best_cost = xb + 8.0 * wb;
    for (int gn = 2; gn <= 8; gn *= 2) {
      if (tiles % 8 || p.tiles_n % gn || tiles_m % (8 / gn)) continue;
      const double c = xb * gn + wb * (8 / gn);
      if (c < 0.9 * best_cost) best = gn, best_cost = c;
    }
    if (s_gn >= 0) best = (s_gn == 0 || (tiles % 8 == 0 && p.tiles_n % s_gn == 0 && 8 % s_gn == 0 && tiles_m % (8 / s_gn) == 0)) ? s_gn : 0;
    p.xcd_gn = best;
  }
  auto kern = k_conv3h<T, BM, W_, BN, WGM, WGN, NPROD, STAGES, MINW, MF, TPS, SUB>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set = true;
  }
  kern<<<dim3(tiles, 1, p.splitk), (NWC + NPROD) * 64, lds, st>>>(p);
}

template <typename T>
static void launch_h3_variant(int k, const ConvP& p, hipStream_t st) {
  switch (k) {
    case 0: launch_h3<T, 256, 32, 192, 4, 2, 16, 1>(p, st); break;
    case 1: launch_h3<T, 128, 16, 192, 2, 4, 16, 1>(p, st); break;
    case 2: launch_h3<T, 128, 16, 192, 2, 2, 16, 1>(p, st); break;
    case 3: launch_h3<T, 256, 16, 192, 4, 2, 16, 1>(p, st); break;
    case 4: launch_h3<T, 128, 32, 192, 2, 4, 16, 1>(p, st); break;
    case 5: launch_h3<T, 128, 32, 192, 2, 2, 16, 1>(p, st); break;
    case 6: launch_h3<T, 256, 32, 192, 4, 2, 32, 1>(p, st); break;
    case 7: launch_h3<T, 128, 16, 192, 2, 2, 32, 1>(p, st); break;
    case 8: launch_h3<T, 256, 16, 192, 4, 2, 32, 1>(p, st); break;
    case 9: launch_h3<T, 128, 32, 192, 2, 2, 32, 1>(p, st); break;
    case 10: launch_h3<T, 64, 8, 96, 2, 2, 16, 3>(p, st); break;
    case 11: launch_h3<T, 64, 4, 96, 2, 2, 16, 3>(p, st); break;
    case 12: launch_h3<T, 64, 8, 96, 2, 2, 16, 3>(p, st); break;
    case 13: launch_h3<T, 128, 16, 96, 2, 2, 16, 3>(p, st); break;
    case 14: launch_h3<T, 128, 32, 96, 2, 2, 16, 3>(p, st); break;
    case 15: break;                                                   // (id 56 is an implicit-GEMM variant, conv.hip)
    case 16: launch_h3<T, 256, 32, 192, 4, 2, 16, 1, 2>(p, st); break;  // 57: as 41 with a 2-deep weight ring (lookahead experiment)
    case 17: launch_h3<T, 256, 32, 128, 4, 2, 16, 1, 3, true>(p, st); break;   // 58: sub-tiled large planes
    case 18: launch_h3<T, 256, 32, 128, 2, 2, 16, 1, 3, true>(p, st); break;   // 59
    case 19: launch_h3<T, 256, 32, 128, 4, 2, 32, 1, 3, true>(p, st); break;   // 60
    case 20: launch_h3<T, 128, 32, 96, 2, 2, 16, 1, 2, false, 2, 3>(p, st); break;   // 61: two workgroups per CU
    case 21: launch_h3<T, 128, 16, 96, 2, 2, 16, 1, 2, false, 2, 3>(p, st); break;   // 62
  }
}

template <int BM, int W_, int BN, int WGM, int WGN, bool SUB>
static void launch_h3_pers(const ConvP& p0, hipStream_t st) {
  constexpr int ROWS = BM / W_, NPI = ((ROWS + 2) * (W_ + 2) + 7) / 8;
  constexpr int lds = 2 * NPI * 1024 + 3 * BN * 128;
  static_assert(lds <= 160 * 1024, "LDS");
  ConvP p = p0;
  p.tiles_n = p.Cout / BN;
  p.splitk = 1;
  const int tiles = (p.M / BM) * p.tiles_n;
  static int cus = 0;
  if (!cus) {
    hipDeviceProp_t prop;
    int dev = 0;
    (void)hipGetDevice(&dev);
    cus = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
  }
  int grid = tiles < cus ? tiles : cus;
  grid -= grid % p.tiles_n;                                    // a workgroup keeps its n tile (constant weight offsets)
  if (grid < p.tiles_n) grid = p.tiles_n;
  auto kern = k_conv3h_pers<BM, W_, BN, WGM, WGN, SUB>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set = true;
  }
  kern<<<dim3(grid, 1, 1), (WGM * WGN + 4) * 64, lds, st>>>(p);
}

void conv3h_launch(int variant, int dtype_size, const ConvP& p, hipStream_t st) {
  const int k = variant - kConv3hFirst;
  if (k == 22) { launch_h3_pers<256, 32, 128, 4, 2, true>(p, st); return; }
  if (k == 23) { launch_h3_pers<128, 32, 192, 2, 2, false>(p, st); return; }
  if (k == 24) { launch_h3<bf16, 64, 32, 96, 2, 2, 16, 3>(p, st); return; }
  if (k == 25) { launch_h3<bf16, 64, 16, 96, 2, 2, 16, 3>(p, st); return; }
  if (dtype_size == 2) launch_h3_variant<bf16>(k, p, st);
  else launch_h3_variant<float>(k, p, st);
}

}  // namespace afldm
