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
#include "conv3h_tile.hpp"

namespace afldm {

template <typename T, int BM, int W_, int BN, int WGM, int WGN, int NPROD, int STAGES, int MINW, int MF, int TPS, bool SUB = false>
__global__ void __launch_bounds__((WGM * WGN + NPROD) * 64, MINW) k_conv3h(ConvP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
#define H3_BX blockIdx.x
#define H3_NBX gridDim.x
#define H3_BZ blockIdx.z
#define H3_PATCH_AUX 0
#define H3_PRODUCER_EXIT return;
#include "conv3h_body.inc"
#undef H3_PRODUCER_EXIT
#undef H3_BX
#undef H3_NBX
#undef H3_BZ
#undef H3_PATCH_AUX
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
    // Fewer weight bytes per workgroup at the low levels (round 5): the 8^2 / 4^2 convolutions are bound by the ~100 GB/s a CU pulls
    // through LDS-DMA (741 KB per workgroup and launch, profiles/r02/small_tile_step_decomposition.txt); 128-pixel x 48-cout tiles
    // stream half the weights and twice the (small) patch: 91 instead of 128 KB per channel block
    {128, 8, 48, 4, 1, 16, 3},        // 67: 8x8 planes, two samples x 48 couts
    {128, 4, 48, 4, 1, 16, 3},        // 68: 4x4 planes, eight samples x 48 couts
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
  static unsigned long long attr_set = 0;
  if (first_on_device(attr_set)) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
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
  static unsigned long long attr_set = 0;
  if (first_on_device(attr_set)) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  }
  kern<<<dim3(grid, 1, 1), (WGM * WGN + 4) * 64, lds, st>>>(p);
}

void conv3h_launch(int variant, int dtype_size, const ConvP& p, hipStream_t st) {
  const int k = variant - kConv3hFirst;
  if (k == 22) { launch_h3_pers<256, 32, 128, 4, 2, true>(p, st); return; }
  if (k == 23) { launch_h3_pers<128, 32, 192, 2, 2, false>(p, st); return; }
  if (k == 24) { launch_h3<bf16, 64, 32, 96, 2, 2, 16, 3>(p, st); return; }
  if (k == 25) { launch_h3<bf16, 64, 16, 96, 2, 2, 16, 3>(p, st); return; }
  if (k == 26) { launch_h3<bf16, 128, 8, 48, 4, 1, 16, 3, 3, false, 2>(p, st); return; }
  if (k == 27) { launch_h3<bf16, 128, 4, 48, 4, 1, 16, 3, 3, false, 2>(p, st); return; }
  if (dtype_size == 2) launch_h3_variant<bf16>(k, p, st);
  else launch_h3_variant<float>(k, p, st);
}

}  // namespace afldm
