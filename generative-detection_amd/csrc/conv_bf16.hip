// 3x3 / 1x1 convolution, NHWC bf16 activations, fp32 accumulate on v_mfma_f32_32x32x16_bf16, gfx950.
// Mixed-precision path of BASELINE.json configs[4] (trainer `precision: bf16`; reference knobs: configs/autoencoder/pose/
// autoencoder_kl_16x16x16.yaml:139 `precision`, train.py:521).  Same layer set as conv3x3_f32.hip:
//   MODE 0  3x3 stride 1 pad 1 (ResnetBlock.conv1/conv2, conv_in, conv_out; data gradient = MODE 0 on the flipped pack)
//   MODE 1  Downsample: pad (0,1,0,1) + 3x3 stride 2
//   MODE 2  Upsample: nearest 2x + 3x3 stride 1 (the 4x intermediate is never formed: the gather reads x[y>>1][x>>1])
//   MODE 3  data gradient of MODE 1 (stride-2 transposed conv; taps that would hit an inserted zero are masked)
//   MODE 4  1x1 (nin_shortcut, AttnBlock q/k/v/proj_out, and their data gradients): pixels flattened to [M/16][16]
// [UPSTREAM] ldm/modules/diffusionmodules/model.py via src/modules/autoencodermodules/feat_encoder.py:4, feat_decoder.py:4.
//
// Implicit GEMM with the OUTPUT CHANNEL on the MFMA row and the PIXEL on the lane:  D[co][px] += W[co][ci] * X[ci][px].
//   A fragment (weights): lane (r, h) holds W[co0 + r][ci0 + 8h .. +7]  -- one coalesced 16-byte load from the pack
//                         [tap][CinP/16][CoutP/32][64 lanes][8], L2 -> registers, no LDS
//   B fragment (pixels):  lane (r, h) holds X[pixel r][ci0 + 8h .. +7]  -- one ds_read_b128 from the halo patch
//   D: lane = pixel, registers = 4-channel runs -> the epilogue packs to bf16 and writes 8 bytes per lane and run.
// Block = 8x16 output pixels x BCO output channels, 4 waves; per KC input channels the halo patch [halo px][KC + 8] is
// staged once (double-buffered, branch-free buffer loads with hardware zero fill), ONE barrier per chunk.
#include "bf16_common.h"
#include <stdlib.h>

namespace {

constexpr int TW = 16;   // tile width; the height TH is 8 (128 pixels) or 16 (256 pixels per block)

struct ConvB {
  const bf16_t* x;         // [N][Hi][Wi][Cin]
  const bf16_t* wpk;       // [taps][CinP/16][CoutP/32][64][8]
  const float* bias;       // [Cout] or null
  const bf16_t* residual;  // [N][Ho][Wo][Cout] bf16 or null
  void* y;                 // [N][Ho][Wo][Cout] bf16 (out_f32 = 0) or f32
  int N, Hi, Wi, Cin, Ho, Wo, Cout, CinP, CoutP;
  int tiles_x, tiles_y, out_f32;
  int xcd;                 // 1: XCD-contiguous tile order
};

template <int MODE, int TH> struct HaloB;
template <int TH> struct HaloB<0, TH> { static constexpr int H = TH + 2, W = TW + 2, TAPS = 9; };
template <int TH> struct HaloB<1, TH> { static constexpr int H = 2 * TH + 1, W = 2 * TW + 1, TAPS = 9; };
template <int TH> struct HaloB<2, TH> { static constexpr int H = TH / 2 + 2, W = TW / 2 + 2, TAPS = 9; };
template <int TH> struct HaloB<3, TH> { static constexpr int H = TH / 2 + 1, W = TW / 2 + 1, TAPS = 9; };
template <int TH> struct HaloB<4, TH> { static constexpr int H = TH, W = TW, TAPS = 1; };

template <int MODE, int TH>
__device__ __forceinline__ int halo_index_b(int r, int c, int kh, int kw, bool& ok) {
  ok = true;
  if (MODE == 0) return (r + kh) * HaloB<0, TH>::W + (c + kw);
  if (MODE == 1) return (2 * r + kh) * HaloB<1, TH>::W + (2 * c + kw);
  if (MODE == 2) return ((r + kh + 1) >> 1) * HaloB<2, TH>::W + ((c + kw + 1) >> 1);
  if (MODE == 4) return r * TW + c;
  ok = (((r + kh) | (c + kw)) & 1) == 0;
  return ((r + kh) >> 1) * HaloB<3, TH>::W + ((c + kw) >> 1);
}
template <int MODE>
__device__ __forceinline__ void halo_origin_b(int oy0, int ox0, int& iy0, int& ix0) {
  if (MODE == 0) { iy0 = oy0 - 1; ix0 = ox0 - 1; }
  else if (MODE == 1) { iy0 = 2 * oy0; ix0 = 2 * ox0; }
  else if (MODE == 4) { iy0 = oy0; ix0 = ox0; }
  else { iy0 = oy0 / 2 - 1; ix0 = ox0 / 2 - 1; }
}

// WCT x WPT MFMA tiles per wave (channel tiles x pixel tiles), WAVES_CO x WAVES_PX waves; WAVES_PX * WPT pixel tiles of 32 = the
// TH x 16 block tile.  The wide form (TH = 16: 64 co x 128 px per wave) halves the weight-fragment traffic per MFMA: with 2 x 2
// tiles every MFMA needs 512 B of weights through the vector L1 (64 B/clk per CU) -- as many cycles as the MFMAs themselves.
template <int MODE, int KC, int WCT, int WPT, int WAVES_CO, int WAVES_PX, int TH, int MINW = 1>
__global__ __launch_bounds__(256, MINW) void conv_bf16_kernel(ConvB p) {
  static_assert(WAVES_CO * WAVES_PX == 4 && WAVES_PX * WPT * 32 == TH * TW, "tile layout");
  constexpr int BCO = WAVES_CO * WCT * 32;
  constexpr int HS = KC + 8;                           // halo row stride in bf16 (16 bytes of padding: conflict-free b128 reads)
  constexpr int HPIX = HaloB<MODE, TH>::H * HaloB<MODE, TH>::W;
  constexpr int VC = KC / 8;                           // 16-byte vectors per halo pixel
  constexpr int HALO_V = HPIX * VC;
  constexpr int HALO_IT = (HALO_V + 255) / 256;
  constexpr int TAPS = HaloB<MODE, TH>::TAPS;
  constexpr int KS = KC / 16;                          // MFMA k-steps per tap and chunk
  constexpr int NIT = TAPS * KS;
  extern __shared__ __attribute__((aligned(16))) bf16_t smem[];   // 2 * HPIX * HS bf16 (two halo stages; 93 KB in the wide form)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wco = wave / WAVES_PX, wpx = wave % WAVES_PX;
  const int li = lane & 31, h = lane >> 5;

  int t = p.xcd ? xcd_contiguous(blockIdx.x, gridDim.x) : blockIdx.x;
  const int tx = t % p.tiles_x; t /= p.tiles_x;
  const int ty = t % p.tiles_y; const int n = t / p.tiles_y;
  const int oy0 = ty * TH, ox0 = tx * TW;
  const int co0 = blockIdx.y * BCO;
  int iy0, ix0;
  halo_origin_b<MODE>(oy0, ox0, iy0, ix0);
  static_assert(MODE == 0 || MODE == 4 || TH == 8, "the wide tile is only built for modes 0 and 4");

  const int esz = p.out_f32 ? 4 : 2;
  const unsigned OOB = 0x7FFFFFF0u;
  // this lane's pixel in each of its pixel tiles (column of the MFMA result)
  unsigned pixoff[WPT];      // element offset of the output pixel, or OOB
  int prr[WPT], pcc[WPT];
#pragma unroll
  for (int pt = 0; pt < WPT; ++pt) {
    const int pm = (wpx * WPT + pt) * 32 + li;
    prr[pt] = pm / TW; pcc[pt] = pm % TW;
    const int oy = oy0 + prr[pt], ox = ox0 + pcc[pt];
    pixoff[pt] = (oy < p.Ho && ox < p.Wo) ? (unsigned)((oy * p.Wo + ox) * p.Cout) : OOB;
  }

  // accumulators start at bias + residual: the epilogue is stores only
  f32x16 acc[WCT][WPT];
  {
    const int64_t img = (int64_t)n * p.Ho * p.Wo * p.Cout;
    const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16_t*>(p.residual ? p.residual + img : p.x), 0, p.residual ? p.Ho * p.Wo * p.Cout * 2 : 0, 0x00020000);
#pragma unroll
    for (int ct = 0; ct < WCT; ++ct)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int co = co0 + (wco * WCT + ct) * 32 + 8 * g + 4 * h;
        float bv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) bv[j] = (p.bias && co + j < p.Cout) ? p.bias[co + j] : 0.f;
#pragma unroll
        for (int pt = 0; pt < WPT; ++pt) {
          float rv[4] = {0.f, 0.f, 0.f, 0.f};
          if (p.residual) {   // Cout % 4 == 0 is checked on the host when a residual is given
            const unsigned off = (pixoff[pt] != OOB && co < p.Cout) ? (pixoff[pt] + (unsigned)co) * 2u : OOB;
            const u32x2 v = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rrsrc, off, 0, 0));
            rv[0] = bf16_lo(v.x); rv[1] = bf16_hi(v.x); rv[2] = bf16_lo(v.y); rv[3] = bf16_hi(v.y);
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[ct][pt][4 * g + j] = bv[j] + rv[j];
        }
      }
  }

  // ---- halo fetch: branch-free buffer loads, out-of-image / past-Cin lanes read 0 --------------------------------------
  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<bf16_t*>(p.x + (int64_t)n * p.Hi * p.Wi * p.Cin), 0, p.Hi * p.Wi * p.Cin * 2, 0x00020000);
  u32x4 hreg[HALO_IT];
  auto load_halo = [&](int c0) {
#pragma unroll
    for (int i = 0; i < HALO_IT; ++i) {
      const int f = tid + 256 * i;
      const int hp = f / VC, q = f % VC;
      const int iy = iy0 + hp / HaloB<MODE, TH>::W, ix = ix0 + hp % HaloB<MODE, TH>::W;
      const int c = c0 + 8 * q;
      const bool ok = f < HALO_V && iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi && c < p.Cin;
      const unsigned voff = ok ? (unsigned)(((iy * p.Wi + ix) * p.Cin + c) * 2) : OOB;
      hreg[i] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, voff, 0, 0);
    }
  };
  auto store_halo = [&](bf16_t* Hs) {
#pragma unroll
    for (int i = 0; i < HALO_IT; ++i) {
      const int f = tid + 256 * i;
      if (f < HALO_V) *reinterpret_cast<u32x4*>(Hs + (f / VC) * HS + 8 * (f % VC)) = hreg[i];
    }
  };

  // ---- operand fetch ------------------------------------------------------------------------------------------------------
  // Weight fragments: fragment index ((tap * KT + kt) * CT + ct), 1 KiB each, lane-linear.  Branch-free buffer loads: a step past
  // the last chunk reads another tap's (unused) fragment or, past the pack, zeros -- never a guarded load (hipcc branches around
  // those and waits vmcnt(0) at every join).
  const int KT = p.CinP / 16, CT = p.CoutP / 32;
  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<bf16_t*>(p.wpk), 0, TAPS * KT * CT * 1024, 0x00020000);
  const int ct0 = co0 / 32 + wco * WCT;
  const unsigned lane16 = lane * 16;
  auto load_a = [&](int ch, int it, bf16x8 (&a)[WCT]) {   // step `it` of chunk `ch`; it >= NIT runs into chunk ch + 1
    const int c2 = ch + it / NIT, i2 = it % NIT;
    const int tap = i2 / KS, ks = i2 % KS;
    const unsigned base = (unsigned)(((tap * KT + c2 * KS + ks) * CT + ct0) * 1024) + lane16;
#pragma unroll
    for (int ct = 0; ct < WCT; ++ct) a[ct] = frag_from_u32x4(__builtin_amdgcn_raw_buffer_load_b128(wrsrc, base + ct * 1024, 0, 0));
  };
  auto load_b = [&](const bf16_t* Hs, int it, bf16x8 (&b)[WPT]) {
    const int tap = it / KS, ks = it % KS;
    const int kh = TAPS == 1 ? 0 : tap / 3, kw = TAPS == 1 ? 0 : tap % 3;
#pragma unroll
    for (int pt = 0; pt < WPT; ++pt) {
      bool ok;
      const int hp = halo_index_b<MODE, TH>(prr[pt], pcc[pt], kh, kw, ok);
      u32x4 v = *reinterpret_cast<const u32x4*>(Hs + hp * HS + 16 * ks + 8 * h);
      if (MODE == 3 && !ok) v = u32x4{0u, 0u, 0u, 0u};
      b[pt] = frag_from_u32x4(v);
    }
  };

  // Register rings with compile-time slots (the step loop is fully unrolled): weights RA - 1 steps ahead (L2 latency), pixels one
  // step ahead (LDS); no rotation copies.  NIT % RA == 0 keeps the weight slots the same in every chunk.
  constexpr int RA = (NIT % 3 == 0) ? 3 : 2, PA = RA - 1;
  static_assert(NIT % RA == 0, "weight ring");
  const int nchunks = p.CinP / KC;
  bf16x8 abuf[RA][WCT], bbuf[2][WPT];
  load_halo(0);
#pragma unroll
  for (int j = 0; j < PA; ++j) load_a(0, j, abuf[j]);
  store_halo(smem);
  __syncthreads();

  for (int ch = 0; ch < nchunks; ++ch) {
    const bf16_t* Hs = smem + (ch & 1) * HPIX * HS;
    const bool more = ch + 1 < nchunks;
    if (more) load_halo((ch + 1) * KC);
    load_b(Hs, 0, bbuf[0]);
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      // pinned: hipcc otherwise sinks the prefetches down to their first use (a weight load followed a few instructions later by
      // the vmcnt(0) that waits for it: one L2 round trip per step)
      load_a(ch, it + PA, abuf[(it + PA) % RA]);
      if (it + 1 < NIT) load_b(Hs, it + 1, bbuf[(it + 1) % 2]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ct = 0; ct < WCT; ++ct)
#pragma unroll
        for (int pt = 0; pt < WPT; ++pt) acc[ct][pt] = mfma_bf16(abuf[it % RA][ct], bbuf[it % 2][pt], acc[ct][pt]);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (more) store_halo(smem + ((ch + 1) & 1) * HPIX * HS);
    __syncthreads();
  }

  // ---- epilogue: stores only ------------------------------------------------------------------------------------------------
  const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(
      static_cast<char*>(p.y) + (int64_t)n * p.Ho * p.Wo * p.Cout * esz, 0, p.Ho * p.Wo * p.Cout * esz, 0x00020000);
#pragma unroll
  for (int ct = 0; ct < WCT; ++ct)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int co = co0 + (wco * WCT + ct) * 32 + 8 * g + 4 * h;
#pragma unroll
      for (int pt = 0; pt < WPT; ++pt) {
        if (p.out_f32) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const unsigned off = (pixoff[pt] != OOB && co + j < p.Cout) ? (pixoff[pt] + (unsigned)(co + j)) * 4u : OOB;
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[ct][pt][4 * g + j]), yrsrc, off, 0, 0);
          }
        } else {   // Cout % 4 == 0 (host check): a 4-channel run is inside or outside as a whole
          const unsigned off = (pixoff[pt] != OOB && co < p.Cout) ? (pixoff[pt] + (unsigned)co) * 2u : OOB;
          u32x2 v;
          v.x = pack_bf16x2(acc[ct][pt][4 * g + 0], acc[ct][pt][4 * g + 1]);
          v.y = pack_bf16x2(acc[ct][pt][4 * g + 2], acc[ct][pt][4 * g + 3]);
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(__attribute__((ext_vector_type(2))) unsigned int, v), yrsrc, off, 0, 0);
        }
      }
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// Wide tile for the stride-1 3x3 convs with more than 64 output channels (every ResnetBlock conv and its data gradient):
// block = 16 x 16 output pixels x 128 output channels, 4 waves of 64 co x 128 px (2 x 4 MFMA tiles).
//
// Why.  In conv_bf16_kernel's 128-pixel tile every MFMA needs 512 B of weights through the vector L1 (64 B/clk per CU): as many
// cycles as the MFMAs themselves.  Here one 16-byte weight load feeds four MFMAs (256 B per MFMA).  The first attempt at this tile
// (64-channel chunks, halo staged through registers: 424 registers, 93 KB of LDS, ONE 4-wave block per CU) ran at 0.56x the narrow
// tile.  This form fits two blocks per CU: __launch_bounds__(256, 2) keeps it inside the 256 architectural registers (VGPR-form
// MFMAs, no accumulator-file copies), the halo takes no registers at all -- it arrives by LDS-DMA (asm, bf16_common.h) into a ring of
// three 21 KB stages, 32 channels per chunk, two chunks ahead, ONE barrier per chunk -- and the LDS image is unpadded [halo px][32 ch]
// with the 16-byte chunk index XOR-swizzled by (px >> 2) & 3, so that sixteen consecutive pixels read by one ds_read_b128 lane group
// fall on sixteen different chunk positions (the swizzle is applied to the DMA's per-lane source address and to every read).
// ------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void conv_bf16_wide_kernel(ConvB p) {
  constexpr int KC = 32, KS = 2, TAPS = 9, NIT = TAPS * KS, TH = 16;
  constexpr int HW = TW + 2, HPIX = (TH + 2) * HW;              // 18 x 18 halo pixels
  constexpr int PIECES = (HPIX * 4 + 63) / 64;                  // 21 LDS-DMA wave-instructions (1 KiB) per stage
  constexpr unsigned STAGEB = PIECES * 1024, NSTG = 3;
  constexpr int WCT = 2, WPT = 4;
  extern __shared__ __attribute__((aligned(1024))) bf16_t smem[];
  const unsigned lds0 = lds_addr_of(smem);
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wco = wave >> 1, wpx = wave & 1;

  int t = p.xcd ? xcd_contiguous(blockIdx.x, gridDim.x) : blockIdx.x;
  const int tx = t % p.tiles_x; t /= p.tiles_x;
  const int ty = t % p.tiles_y; const int n = t / p.tiles_y;
  const int oy0 = ty * TH, ox0 = tx * TW;
  const int co0 = blockIdx.y * 128;
  const int iy0 = oy0 - 1, ix0 = ox0 - 1;
  const int esz = p.out_f32 ? 4 : 2;
  const unsigned OOB = 0x7FFFFFF0u;

  // ---- halo fetch plan: piece j = wave + 4 k; lane fills 16-byte slot 64 j + lane = (halo pixel hp, chunk position chp) ----------
  const i32x4_t xw = rsrc_words(p.x + (int64_t)n * p.Hi * p.Wi * p.Cin, (unsigned)(p.Hi * p.Wi * p.Cin * 2));
  constexpr int NPW = (PIECES + 3) / 4;                           // pieces per wave (6; the last round only has piece 20)
  unsigned hvoff[NPW];
#pragma unroll
  for (int k = 0; k < NPW; ++k) {
    const int slot = 64 * (wave + 4 * k) + lane;
    const int hp = slot >> 2, chp = slot & 3;
    const int ch = chp ^ ((hp >> 2) & 3);
    const int iy = iy0 + hp / HW, ix = ix0 + hp % HW;
    const bool ok = hp < HPIX && iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi;
    hvoff[k] = ok ? (unsigned)(((iy * p.Wi + ix) * p.Cin + 8 * ch) * 2) : OOB;
  }
  auto issue = [&](int chunk, unsigned stage) {
#pragma unroll
    for (int k = 0; k < NPW; ++k) {
      const int j = wave + 4 * k;
      if (j < PIECES)      // wave-uniform
        lds_dma16(xw, (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + stage + 1024u * j)), hvoff[k] + (unsigned)(chunk * KC * 2));
    }
  };
  const int nchunks = p.CinP / KC;
  issue(0, 0);
  if (nchunks > 1) issue(1, STAGEB);

  // this lane's pixel in each of its four pixel tiles (column of the MFMA result)
  unsigned pixoff[WPT];
  int hp0[WPT];
#pragma unroll
  for (int pt = 0; pt < WPT; ++pt) {
    const int pm = (wpx * WPT + pt) * 32 + li;
    const int pr = pm / TW, pc = pm % TW;
    const int oy = oy0 + pr, ox = ox0 + pc;
    pixoff[pt] = (oy < p.Ho && ox < p.Wo) ? (unsigned)((oy * p.Wo + ox) * p.Cout) : OOB;
    hp0[pt] = pr * HW + pc;
  }

  // accumulators start at bias + residual: the epilogue is stores only
  f32x16 acc[WCT][WPT];
  {
    const int64_t img = (int64_t)n * p.Ho * p.Wo * p.Cout;
    const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16_t*>(p.residual ? p.residual + img : p.x), 0, p.residual ? p.Ho * p.Wo * p.Cout * 2 : 0, 0x00020000);
#pragma unroll
    for (int ct = 0; ct < WCT; ++ct)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int co = co0 + (wco * WCT + ct) * 32 + 8 * g + 4 * h;
        float bv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) bv[j] = (p.bias && co + j < p.Cout) ? p.bias[co + j] : 0.f;
#pragma unroll
        for (int pt = 0; pt < WPT; ++pt) {
          float rv[4] = {0.f, 0.f, 0.f, 0.f};
          if (p.residual) {
            const unsigned off = (pixoff[pt] != OOB && co < p.Cout) ? (pixoff[pt] + (unsigned)co) * 2u : OOB;
            const u32x2 v = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rrsrc, off, 0, 0));
            rv[0] = bf16_lo(v.x); rv[1] = bf16_hi(v.x); rv[2] = bf16_lo(v.y); rv[3] = bf16_hi(v.y);
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[ct][pt][4 * g + j] = bv[j] + rv[j];
        }
      }
  }

  // ---- operand fetch -------------------------------------------------------------------------------------------------------
  const int KT = p.CinP / 16, CT = p.CoutP / 32;
  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<bf16_t*>(p.wpk), 0, TAPS * KT * CT * 1024, 0x00020000);
  const int ct0 = co0 / 32 + wco * WCT;
  const unsigned lane16 = lane * 16;
  auto load_a = [&](int ch, int it, bf16x8 (&a)[WCT]) {   // step `it` of chunk `ch`; it >= NIT runs into chunk ch + 1 (or past the pack: zeros)
    const int c2 = ch + it / NIT, i2 = it % NIT;
    const int tap = i2 / KS, ks = i2 % KS;
    const unsigned base = (unsigned)(((tap * KT + c2 * KS + ks) * CT + ct0) * 1024) + lane16;
#pragma unroll
    for (int ct = 0; ct < WCT; ++ct) a[ct] = frag_from_u32x4(__builtin_amdgcn_raw_buffer_load_b128(wrsrc, base + ct * 1024, 0, 0));
  };
  // pixel fragment of step (tap, ks): lane (pixel, h) reads chunk 2 ks + h of halo pixel hp = hp0 + kh * 18 + kw, stored at chunk
  // position (2 ks + h) ^ ((hp >> 2) & 3)
  auto load_b = [&](unsigned stage_addr, int it, bf16x8 (&b)[WPT]) {
    const int tap = it / KS, ks = it % KS;
    const int toff = (tap / 3) * HW + (tap % 3);
#pragma unroll
    for (int pt = 0; pt < WPT; ++pt) {
      const unsigned hp = (unsigned)(hp0[pt] + toff);
      const unsigned pos = ((hp >> 2) ^ (unsigned)(2 * ks + h)) & 3u;
      b[pt] = frag_from_u32x4(lds_ld128(stage_addr + hp * 64u + pos * 16u));
    }
  };

  constexpr int RA = 3, PA = RA - 1;
  static_assert(NIT % RA == 0, "weight ring");
  bf16x8 abuf[RA][WCT], bbuf[2][WPT];
#pragma unroll
  for (int j = 0; j < PA; ++j) load_a(0, j, abuf[j]);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  unsigned cur = 0, nxt = STAGEB, fre = 2 * STAGEB;
  for (int ch = 0; ch < nchunks; ++ch) {
#ifndef ODVAE_CONVW_ABL_NODMA
    if (ch + 2 < nchunks) issue(ch + 2, fre);          // the stage chunk ch - 1 left: its last reads ended before the barrier above
#endif
    const unsigned sa = lds0 + cur;
    load_b(sa, 0, bbuf[0]);
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      load_a(ch, it + PA, abuf[(it + PA) % RA]);
      if (it + 1 < NIT) load_b(sa, it + 1, bbuf[(it + 1) % 2]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ct = 0; ct < WCT; ++ct)
#pragma unroll
        for (int pt = 0; pt < WPT; ++pt) acc[ct][pt] = mfma_bf16(abuf[it % RA][ct], bbuf[it % 2][pt], acc[ct][pt]);
      __builtin_amdgcn_sched_barrier(0);
    }
    { const unsigned o = cur; cur = nxt; nxt = fre; fre = o; }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // ---- epilogue: stores only ------------------------------------------------------------------------------------------------
  const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(
      static_cast<char*>(p.y) + (int64_t)n * p.Ho * p.Wo * p.Cout * esz, 0, p.Ho * p.Wo * p.Cout * esz, 0x00020000);
#pragma unroll
  for (int ct = 0; ct < WCT; ++ct)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int co = co0 + (wco * WCT + ct) * 32 + 8 * g + 4 * h;
#pragma unroll
      for (int pt = 0; pt < WPT; ++pt) {
        if (p.out_f32) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const unsigned off = (pixoff[pt] != OOB && co + j < p.Cout) ? (pixoff[pt] + (unsigned)(co + j)) * 4u : OOB;
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[ct][pt][4 * g + j]), yrsrc, off, 0, 0);
          }
        } else {
          const unsigned off = (pixoff[pt] != OOB && co < p.Cout) ? (pixoff[pt] + (unsigned)co) * 2u : OOB;
          u32x2 v;
          v.x = pack_bf16x2(acc[ct][pt][4 * g + 0], acc[ct][pt][4 * g + 1]);
          v.y = pack_bf16x2(acc[ct][pt][4 * g + 2], acc[ct][pt][4 * g + 3]);
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(__attribute__((ext_vector_type(2))) unsigned int, v), yrsrc, off, 0, 0);
        }
      }
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// All operands through LDS (stride-1 3x3 convs with more than 64 output channels): block = 16 x 16 output pixels x 128 output
// channels, 8 waves of 64 co x 64 px (2 x 2 MFMA tiles), 16 input channels per chunk.
//
// The lesson of the wide tile above: a wave's vector-memory operations complete in issue order, so a per-step weight load issued
// behind the fetch of a later halo chunk waits out that fetch's HBM latency.  Here NO global load is waited for inside a chunk: the
// chunk's weights (nine taps x four 32-channel tiles = 36 fragments of 1 KiB, which sit lane-linear in the pack exactly as an
// LDS-DMA piece writes them) and its halo (18 x 18 pixels x 16 channels = 11 pieces) arrive by asm LDS-DMA in a ring of three 47 KB
// stages, two chunks ahead; every MFMA operand is a ds_read_b128; one `s_waitcnt vmcnt(0)` + barrier per chunk (36 MFMAs per wave).
// Halo image: [halo px][2 x 16 B], the half index XOR-ed with (px >> 3) & 1 so that sixteen consecutive pixels of a lane group cover
// sixteen different 16-byte positions.
// ------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void conv_bf16_lds_kernel(ConvB p) {
  constexpr int TAPS = 9, TH = 16, HW = TW + 2, HPIX = (TH + 2) * HW;
  constexpr int WPIECES = TAPS * 4, HPIECES = (HPIX * 2 + 63) / 64, PIECES = WPIECES + HPIECES + 1;   // 36 + 11 + one dummy: six per wave
  constexpr unsigned STAGEB = PIECES * 1024, HALO_OFF = WPIECES * 1024;
  constexpr int WCT = 2, WPT = 2;
  extern __shared__ __attribute__((aligned(1024))) bf16_t smem[];
  const unsigned lds0 = lds_addr_of(smem);
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wco = wave >> 2, wpx = wave & 3;

  int t = p.xcd ? xcd_contiguous(blockIdx.x, gridDim.x) : blockIdx.x;
  const int tx = t % p.tiles_x; t /= p.tiles_x;
  const int ty = t % p.tiles_y; const int n = t / p.tiles_y;
  const int oy0 = ty * TH, ox0 = tx * TW;
  const int co0 = blockIdx.y * 128;
  const int iy0 = oy0 - 1, ix0 = ox0 - 1;
  const int esz = p.out_f32 ? 4 : 2;
  const unsigned OOB = 0x7FFFFFF0u;
  const int KT = p.CinP / 16, CT = p.CoutP / 32;
  const int nchunks = KT;

  // ---- fetch plan: piece j = wave + 8 k (k = 0 .. 5); j < 36: weight fragment (tap j / 4, channel tile j % 4), else halo piece j - 36 --
  const i32x4_t xw = rsrc_words(p.x + (int64_t)n * p.Hi * p.Wi * p.Cin, (unsigned)(p.Hi * p.Wi * p.Cin * 2));
  const i32x4_t ww = rsrc_words(p.wpk, (unsigned)(TAPS * KT * CT * 1024));
  unsigned hvoff[2];                 // this lane's source offsets of the wave's (up to) two halo pieces, chunk 0
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int j = wave + 8 * (4 + q);
    const int slot = 64 * (j - WPIECES) + lane;
    const int px = slot >> 1, half = (slot & 1) ^ ((px >> 3) & 1);
    const int iy = iy0 + px / HW, ix = ix0 + px % HW;
    const bool ok = j >= WPIECES && px < HPIX && iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi;   // piece 47 is all out of range: zeros
    hvoff[q] = ok ? (unsigned)(((iy * p.Wi + ix) * p.Cin + 8 * half) * 2) : OOB;
  }
  const unsigned lane16 = lane * 16;
  auto issue = [&](int kt, unsigned stage) {
    const unsigned st = lds0 + stage;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const int j = wave + 8 * k;                     // wave-uniform
      if (j < WPIECES) {
        const int tap = j >> 2, ct = j & 3;
        lds_dma16(ww, (unsigned)__builtin_amdgcn_readfirstlane((int)(st + 1024u * j)),
                  (unsigned)(((tap * KT + kt) * CT + co0 / 32 + ct) * 1024) + lane16);
      } else {      // (the last piece only pads every wave's batch to six pieces: the counted wait below needs one number)
        lds_dma16(xw, (unsigned)__builtin_amdgcn_readfirstlane((int)(st + 1024u * j)), hvoff[k >= 4 ? k - 4 : 0] + (unsigned)(kt * 32));
      }
    }
  };
  issue(0, 0);
  if (nchunks > 1) issue(1, STAGEB);

  // this lane's pixel in each of its two pixel tiles (column of the MFMA result)
  unsigned pixoff[WPT];
  int hp0[WPT];
#pragma unroll
  for (int pt = 0; pt < WPT; ++pt) {
    const int pm = (wpx * WPT + pt) * 32 + li;
    const int pr = pm / TW, pc = pm % TW;
    const int oy = oy0 + pr, ox = ox0 + pc;
    pixoff[pt] = (oy < p.Ho && ox < p.Wo) ? (unsigned)((oy * p.Wo + ox) * p.Cout) : OOB;
    hp0[pt] = pr * HW + pc;
  }

  // accumulators start at bias + residual: the epilogue is stores only
  f32x16 acc[WCT][WPT];
  {
    const int64_t img = (int64_t)n * p.Ho * p.Wo * p.Cout;
    const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16_t*>(p.residual ? p.residual + img : p.x), 0, p.residual ? p.Ho * p.Wo * p.Cout * 2 : 0, 0x00020000);
#pragma unroll
    for (int ct = 0; ct < WCT; ++ct)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int co = co0 + (wco * WCT + ct) * 32 + 8 * g + 4 * h;
        float bv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) bv[j] = (p.bias && co + j < p.Cout) ? p.bias[co + j] : 0.f;
#pragma unroll
        for (int pt = 0; pt < WPT; ++pt) {
          float rv[4] = {0.f, 0.f, 0.f, 0.f};
          if (p.residual) {
            const unsigned off = (pixoff[pt] != OOB && co < p.Cout) ? (pixoff[pt] + (unsigned)co) * 2u : OOB;
            const u32x2 v = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rrsrc, off, 0, 0));
            rv[0] = bf16_lo(v.x); rv[1] = bf16_hi(v.x); rv[2] = bf16_lo(v.y); rv[3] = bf16_hi(v.y);
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[ct][pt][4 * g + j] = bv[j] + rv[j];
        }
      }
  }

  // operand reads of step `tap` from the stage at LDS address sa: weight fragments are lane-linear, pixel fragments swizzled
  const unsigned a_lane = (unsigned)(wco * WCT) * 1024u + lane16;
  auto load_ab = [&](unsigned sa, int tap, bf16x8 (&a)[WCT], bf16x8 (&b)[WPT]) {
#pragma unroll
    for (int ct = 0; ct < WCT; ++ct) a[ct] = frag_from_u32x4(lds_ld128(sa + a_lane + (unsigned)((tap * 4 + ct) * 1024)));
    const int toff = (tap / 3) * HW + (tap % 3);
#pragma unroll
    for (int pt = 0; pt < WPT; ++pt) {
      const unsigned px = (unsigned)(hp0[pt] + toff);
      b[pt] = frag_from_u32x4(lds_ld128(sa + HALO_OFF + px * 32u + ((((px >> 3) ^ (unsigned)h) & 1u) << 4)));
    }
  };

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  unsigned cur = 0, nxt = STAGEB, fre = 2 * STAGEB;
  constexpr int RS = 3;                                 // steps in flight (operands of step s + 2 are requested before step s multiplies)
  for (int ch = 0; ch < nchunks; ++ch) {
    if (ch + 2 < nchunks) issue(ch + 2, fre);           // the stage chunk ch - 1 left: its last reads ended before the barrier above
    const unsigned sa = lds0 + cur;
    bf16x8 abuf[RS][WCT], bbuf[RS][WPT];
    load_ab(sa, 0, abuf[0], bbuf[0]);
    load_ab(sa, 1, abuf[1], bbuf[1]);
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {
      if (tap + 2 < TAPS) load_ab(sa, tap + 2, abuf[(tap + 2) % RS], bbuf[(tap + 2) % RS]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ct = 0; ct < WCT; ++ct)
#pragma unroll
        for (int pt = 0; pt < WPT; ++pt) acc[ct][pt] = mfma_bf16(abuf[tap % RS][ct], bbuf[tap % RS][pt], acc[ct][pt]);
      __builtin_amdgcn_sched_barrier(0);
    }
    { const unsigned o = cur; cur = nxt; nxt = fre; fre = o; }
    // The batch issued at the top of THIS chunk is for the chunk after next: only the previous batch has to have landed now.  A fetch
    // gets two chunk periods (HBM latency is longer than one).
    if (ch + 2 < nchunks) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // ---- epilogue: stores only ------------------------------------------------------------------------------------------------
  const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(
      static_cast<char*>(p.y) + (int64_t)n * p.Ho * p.Wo * p.Cout * esz, 0, p.Ho * p.Wo * p.Cout * esz, 0x00020000);
#pragma unroll
  for (int ct = 0; ct < WCT; ++ct)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int co = co0 + (wco * WCT + ct) * 32 + 8 * g + 4 * h;
#pragma unroll
      for (int pt = 0; pt < WPT; ++pt) {
        if (p.out_f32) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const unsigned off = (pixoff[pt] != OOB && co + j < p.Cout) ? (pixoff[pt] + (unsigned)(co + j)) * 4u : OOB;
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[ct][pt][4 * g + j]), yrsrc, off, 0, 0);
          }
        } else {
          const unsigned off = (pixoff[pt] != OOB && co < p.Cout) ? (pixoff[pt] + (unsigned)co) * 2u : OOB;
          u32x2 v;
          v.x = pack_bf16x2(acc[ct][pt][4 * g + 0], acc[ct][pt][4 * g + 1]);
          v.y = pack_bf16x2(acc[ct][pt][4 * g + 2], acc[ct][pt][4 * g + 3]);
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(__attribute__((ext_vector_type(2))) unsigned int, v), yrsrc, off, 0, 0);
        }
      }
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// 128 x 128 register tiles: block = 16 x 32 output pixels x 128 output channels, FOUR waves (one per SIMD; the 256 accumulator
// registers of a 4 x 4 grid of MFMA tiles each, in AGPRs), 16 input channels per chunk.
//
// Why this shape.  (1) Every LDS-staged variant above paid for its LDS-DMA pieces: a `buffer_load ... lds` costs the issuing wave
// 60-185 cycles (MI355X_MICROARCH.md), and weights + halo through LDS are 48-56 pieces per chunk -- 30-50 % of the chunk's MFMA time.
// (2) A weight fragment (1 KiB per tap, 16 ci, 32 co) fetched straight from L2 into registers costs one vector-memory instruction
// that hides in an MFMA gap, but with 64 x 64 wave tiles it feeds two MFMAs only and the weight stream of a whole chip nears the L2's
// bandwidth.  Here a fragment feeds FOUR MFMAs (four pixel tiles per wave), the chunk's 36 fragments live in 144 VGPRs and are
// refilled in place for the next chunk right behind their last use, and only the halo (18 x 34 pixels x 16 channels: five pieces per
// wave) goes through LDS, two chunks ahead in a ring of three 20 KiB stages.  (3) vmcnt retires in order: all vector memory of the
// loop is issued from inline asm and awaited by hand -- a fragment with exactly the 32 younger refills + 5 halo pieces in flight, the
// halo of the next chunk with the 36 refills + 5 pieces issued since.
// ------------------------------------------------------------------------------------------------------------------------
// -DODVAE_BIG_ABL=<bits>: timing-only ablations (wrong results): 1 no weight refills / waits, 2 no halo DMA / wait, 4 no barrier, 8 pixel fragments read once per chunk
#ifndef ODVAE_BIG_ABL
#define ODVAE_BIG_ABL 0
#endif
__global__ __launch_bounds__(256) void conv_bf16_big_kernel(ConvB p) {
  constexpr int TAPS = 9, TH = 16, TWB = 32, HWB = TWB + 2, HPIX = (TH + 2) * HWB;
  constexpr int HPIECES = (HPIX * 2 + 63) / 64;       // 20 halo pieces per chunk
  static_assert(HPIECES % 4 == 0, "whole pieces per wave");
  constexpr int HPW = HPIECES / 4;                    // five per wave
  constexpr unsigned HSTAGE = HPIECES * 1024;
  constexpr int WCT = 4, WPT = 4, NFR = TAPS * WCT;   // 36 weight fragments per chunk
  extern __shared__ __attribute__((aligned(1024))) bf16_t smem[];
  const unsigned lds0 = lds_addr_of(smem);
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, h = lane >> 5;
  const int wpx = __builtin_amdgcn_readfirstlane(tid >> 6);

  int t = p.xcd ? xcd_contiguous(blockIdx.x, gridDim.x) : blockIdx.x;
  const int tx = t % p.tiles_x; t /= p.tiles_x;
  const int ty = t % p.tiles_y; const int n = t / p.tiles_y;
  const int oy0 = ty * TH, ox0 = tx * TWB;
  const int co0 = blockIdx.y * 128;
  const int iy0 = oy0 - 1, ix0 = ox0 - 1;
  const int esz = p.out_f32 ? 4 : 2;
  const unsigned OOB = 0x7FFFFFF0u;
  const int KT = p.CinP / 16, CT = p.CoutP / 32;
  const int nchunks = KT;

  const i32x4_t xw = rsrc_words(p.x + (int64_t)n * p.Hi * p.Wi * p.Cin, (unsigned)(p.Hi * p.Wi * p.Cin * 2));
  const i32x4_t ww = rsrc_words(p.wpk, (unsigned)(TAPS * KT * CT * 1024));
  unsigned hvoff[HPW];
#pragma unroll
  for (int q = 0; q < HPW; ++q) {
    const int slot = 64 * (wpx + 4 * q) + lane;
    const int px = slot >> 1, half = (slot & 1) ^ ((px >> 3) & 1);
    const int iy = iy0 + px / HWB, ix = ix0 + px % HWB;
    const bool ok = px < HPIX && iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi;
    hvoff[q] = ok ? (unsigned)(((iy * p.Wi + ix) * p.Cin + 8 * half) * 2) : OOB;
  }
  const unsigned lane16 = lane * 16;
  // halo pieces of chunk kt -> stage hst; past the last chunk the same five instructions run against an empty descriptor (zeros land
  // in a stage nobody reads any more), so that the number of operations in flight is the same in every chunk
  auto issue_h = [&](int kt, unsigned hst) {
    i32x4_t r = xw;
    r.z = kt < nchunks ? xw.z : 0;
    const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane(kt < nchunks ? kt * 32 : 0);
#pragma unroll
    for (int q = 0; q < HPW; ++q)
      lds_dma16_s(r, (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + hst + 1024u * (wpx + 4 * q))), hvoff[q], soff);
  };
  // weight fragment (tap, ct) of chunk kt -> registers
  u32x4 wf[TAPS][WCT];
  auto load_w = [&](int kt, int tap, int ct, u32x4& dst) {
    const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane(((tap * KT + kt) * CT + co0 / 32 + ct) * 1024);
    // (no s_nop in front: `soff` is the result of scalar arithmetic here -- the kernel has no SGPR spills, checked in the build's
    // register statistics -- and the four MFMAs in front of every refill cover a VALU write anyway)
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(dst) : "v"(lane16), "s"(ww), "s"(soff) : "memory");
  };
  issue_h(0, 0);
#pragma unroll
  for (int tap = 0; tap < TAPS; ++tap)
#pragma unroll
    for (int ct = 0; ct < WCT; ++ct) load_w(0, tap, ct, wf[tap][ct]);
  issue_h(1, HSTAGE);

  // this lane's pixel in each of its four pixel tiles (one tile = one row of 32 pixels; column of the MFMA result)
  unsigned pixoff[WPT];
  unsigned hp0[WPT];
#pragma unroll
  for (int pt = 0; pt < WPT; ++pt) {
    const int pr = wpx * WPT + pt, pc = li;
    const int oy = oy0 + pr, ox = ox0 + pc;
    pixoff[pt] = (oy < p.Ho && ox < p.Wo) ? (unsigned)((oy * p.Wo + ox) * p.Cout) : OOB;
    hp0[pt] = (unsigned)(pr * HWB + pc);
  }

  // With one wave per SIMD nothing hides a long prologue: the accumulators start at zero (bias and residual join in the epilogue, whose
  // loads are requested in batches)
  f32x16 acc[WCT][WPT];
#pragma unroll
  for (int ct = 0; ct < WCT; ++ct)
#pragma unroll
    for (int pt = 0; pt < WPT; ++pt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ct][pt][r] = 0.f;

  auto load_px = [&](unsigned ha, int tap, bf16x8 (&b)[WPT]) {
    const unsigned toff = (unsigned)((tap / 3) * HWB + (tap % 3));
#pragma unroll
    for (int pt = 0; pt < WPT; ++pt) {
      const unsigned px = hp0[pt] + toff;
      b[pt] = frag_from_u32x4(lds_ld128(ha + px * 32u + ((((px >> 3) ^ (unsigned)h) & 1u) << 4)));
    }
  };

  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(HPW) : "memory");      // halo 0 and the 36 fragments (the five pieces of halo 1 stay in flight)
  __syncthreads();
  unsigned hcur = 0, hnxt = HSTAGE, hfre = 2 * HSTAGE;
  for (int ch = 0; ch < nchunks; ++ch) {
    const int chn = ch + 1 < nchunks ? ch + 1 : 0;               // past the last chunk: a harmless reload
    const unsigned ha = lds0 + hcur;
    bf16x8 bbuf[2][WPT];
    load_px(ha, 0, bbuf[0]);
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {
      if (tap + 1 < TAPS && !(ODVAE_BIG_ABL & 8)) load_px(ha, tap + 1, bbuf[(tap + 1) & 1]);
      if (ODVAE_BIG_ABL & 8) { for (int pt = 0; pt < WPT; ++pt) bbuf[(tap + 1) & 1][pt] = bbuf[tap & 1][pt]; }
      // fragments of this tap: requested one chunk ago; younger than them: 4 (8 - tap) refills + 5 halo pieces + 4 tap refills = 37
      if (!(ODVAE_BIG_ABL & 3))
      asm volatile("s_waitcnt vmcnt(%4)" : "+v"(wf[tap][0]), "+v"(wf[tap][1]), "+v"(wf[tap][2]), "+v"(wf[tap][3]) : "n"(NFR - WCT + HPW) : "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ct = 0; ct < WCT; ++ct) {      // four MFMAs per fragment, its refill for the next chunk right behind them
#pragma unroll
        for (int pt = 0; pt < WPT; ++pt) acc[ct][pt] = mfma_bf16(frag_from_u32x4(wf[tap][ct]), bbuf[tap & 1][pt], acc[ct][pt]);
        if (!(ODVAE_BIG_ABL & 1)) load_w(chn, tap, ct, wf[tap][ct]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (!(ODVAE_BIG_ABL & 2)) issue_h(ch + 2, hfre);          // the stage chunk ch - 1 read: its reads ended before the barrier above
    { const unsigned o = hcur; hcur = hnxt; hnxt = hfre; hfre = o; }
    if (ODVAE_BIG_ABL & 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NFR + HPW) : "memory");      // halo ch+1 (older than this chunk's 36 refills + 5 pieces) has landed
    if (!(ODVAE_BIG_ABL & 4)) __syncthreads();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the refills past the last chunk still write registers

  // ---- epilogue: bias (one 16-byte load per (channel tile, quad)) and residual requested in batches, then stores -----------------
  const int64_t img = (int64_t)n * p.Ho * p.Wo * p.Cout;
  const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(
      static_cast<char*>(p.y) + img * esz, 0, p.Ho * p.Wo * p.Cout * esz, 0x00020000);
  const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<bf16_t*>(p.residual ? p.residual + img : p.x), 0, p.residual ? p.Ho * p.Wo * p.Cout * 2 : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t brsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.bias ? p.bias : reinterpret_cast<const float*>(p.x)), 0, p.bias ? p.Cout * 4 : 0, 0x00020000);
#pragma unroll
  for (int ct = 0; ct < WCT; ++ct) {
    u32x4 bq[4];
    u32x2 rq[4][WPT];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int co = co0 + ct * 32 + 8 * g + 4 * h;
      bq[g] = __builtin_amdgcn_raw_buffer_load_b128(brsrc, co < p.Cout ? (unsigned)co * 4u : OOB, 0, 0);     // (Cout % 4 == 0: whole quads)
#pragma unroll
      for (int pt = 0; pt < WPT; ++pt) rq[g][pt] = u32x2{0u, 0u};
      if (p.residual) {
#pragma unroll
        for (int pt = 0; pt < WPT; ++pt)
          rq[g][pt] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rrsrc, (pixoff[pt] != OOB && co < p.Cout) ? (pixoff[pt] + (unsigned)co) * 2u : OOB, 0, 0));
      }
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int co = co0 + ct * 32 + 8 * g + 4 * h;
      const float b0 = __uint_as_float(bq[g].x), b1 = __uint_as_float(bq[g].y), b2 = __uint_as_float(bq[g].z), b3 = __uint_as_float(bq[g].w);
#pragma unroll
      for (int pt = 0; pt < WPT; ++pt) {
        const float v0 = acc[ct][pt][4 * g + 0] + b0 + bf16_lo(rq[g][pt].x), v1 = acc[ct][pt][4 * g + 1] + b1 + bf16_hi(rq[g][pt].x);
        const float v2 = acc[ct][pt][4 * g + 2] + b2 + bf16_lo(rq[g][pt].y), v3 = acc[ct][pt][4 * g + 3] + b3 + bf16_hi(rq[g][pt].y);
        if (p.out_f32) {
          const float vv[4] = {v0, v1, v2, v3};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const unsigned off = (pixoff[pt] != OOB && co + j < p.Cout) ? (pixoff[pt] + (unsigned)(co + j)) * 4u : OOB;
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(vv[j]), yrsrc, off, 0, 0);
          }
        } else {
          const unsigned off = (pixoff[pt] != OOB && co < p.Cout) ? (pixoff[pt] + (unsigned)co) * 2u : OOB;
          u32x2 v;
          v.x = pack_bf16x2(v0, v1);
          v.y = pack_bf16x2(v2, v3);
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(__attribute__((ext_vector_type(2))) unsigned int, v), yrsrc, off, 0, 0);
        }
      }
    }
  }
}

// OIHW f32 (kh x kw = 3x3 or 1x1) -> bf16 fragment packs.
//   fwd:   reduce over Cin, rows = Cout:   W[tap][co][ci]
//   dgrad: reduce over Cout, rows = Cin:   W'[tap][ci][co] = w[co][ci][flip(tap)]   (MODE 0 / 4 data gradient; MODE 3 uses
//          the same flipped pack: dx = transposed conv of dy)
// layout [tap][RP/16][OP/32][lane 64][8]: lane (r, h) element j = W[row 32*ot + r][k = 16*kt + 8h + j]
__global__ void conv_pack_bf16_kernel(const float* __restrict__ w, int Cout, int Cin, int taps,
                                      bf16_t* __restrict__ fwd, int RP_f, int OP_f, bf16_t* __restrict__ dgr, int RP_d, int OP_d) {
  const int64_t nf = fwd ? (int64_t)taps * RP_f * OP_f : 0;
  const int64_t nd = dgr ? (int64_t)taps * RP_d * OP_d : 0;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < nf + nd; idx += (int64_t)gridDim.x * blockDim.x) {
    const bool f = idx < nf;
    int64_t r = f ? idx : idx - nf;
    const int RP = f ? RP_f : RP_d, OP = f ? OP_f : OP_d;
    const int j = r & 7; r >>= 3;
    const int lane = r & 63; r >>= 6;
    const int ot = (int)(r % (OP / 32)); r /= (OP / 32);
    const int kt = (int)(r % (RP / 16)); const int tap = (int)(r / (RP / 16));
    const int row = 32 * ot + (lane & 31), k = 16 * kt + 8 * (lane >> 5) + j;
    float v = 0.f;
    if (f) { if (row < Cout && k < Cin) v = w[((int64_t)row * Cin + k) * taps + tap]; }
    else   { if (row < Cin && k < Cout) v = w[((int64_t)k * Cin + row) * taps + (taps - 1 - tap)]; }
    (f ? fwd : dgr)[f ? idx : idx - nf] = f32_to_bf16(v);
  }
}

int pad_to(int v, int m) { return (v + m - 1) / m * m; }

template <int MODE, int KC, int WCT, int WPT, int WAVES_CO, int WAVES_PX, int TH, int MINW = 1>
void launch_cfg(const ConvB& p, dim3 grid, hipStream_t st) {
  constexpr int bytes = 2 * HaloB<MODE, TH>::H * HaloB<MODE, TH>::W * (KC + 8) * 2;
  static bool once = false;
  if (!once) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bf16_kernel<MODE, KC, WCT, WPT, WAVES_CO, WAVES_PX, TH, MINW>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    once = true;
  }
  hipLaunchKernelGGL((conv_bf16_kernel<MODE, KC, WCT, WPT, WAVES_CO, WAVES_PX, TH, MINW>), grid, dim3(256), bytes, st, p);
}

template <int MODE, int KC>
void launch_by_cout(ConvB& p, hipStream_t st) {
  // A/B switch.  Measured (bench.py --bf16, B=32, 256x256): the wide tile runs the family at 365 TFLOP/s against 646 for the
  // 128-pixel tile -- one 4-wave block per CU (424 registers, 93 KB of LDS) has nothing to overlap its prologue, barriers and
  // epilogue with, which costs more than the halved weight traffic saves.  Off by default.
  static const bool wide = getenv("ODVAE_CONV_BF16_WIDE") != nullptr;
  if constexpr (MODE == 0 || MODE == 4) {
    if (p.Cout > 64 && p.Ho >= 16 && wide) {     // wide tile: 16 x 16 pixels x 128 channels, 64 co x 128 px per wave
      p.tiles_y = ceil_div(p.Ho, 16);
      launch_cfg<MODE, KC, 2, 4, 2, 2, 16>(p, dim3(p.N * p.tiles_x * p.tiles_y, ceil_div(p.Cout, 128)), st);
      return;
    }
  }
  const int tiles = p.N * p.tiles_x * p.tiles_y;
  if (p.Cout > 64)      launch_cfg<MODE, KC, 2, 2, 2, 2, 8>(p, dim3(tiles, ceil_div(p.Cout, 128)), st);
  else if (p.Cout > 32) launch_cfg<MODE, KC, 2, 1, 1, 4, 8>(p, dim3(tiles, 1), st);
  else                  launch_cfg<MODE, KC, 1, 1, 1, 4, 8>(p, dim3(tiles, 1), st);
}

}  // namespace

static int g_wide_tile = -1;   // -1: not chosen yet (environment decides at the first call)

extern "C" {

// Tile of the stride-1 3x3 convs with Cout > 64 and Ho >= 16: 0 = 8 x 16 pixels (conv_bf16_kernel), 1 = 16 x 16 with an LDS-DMA halo
// ring and weights from L2 (conv_bf16_wide_kernel, Cin % 32 == 0), 2 = 16 x 16 with weights AND halo through LDS
// (conv_bf16_lds_kernel, Cin % 16 == 0).  ODVAE_CONV_BF16_WIDE2 presets it.  Returns the previous setting (-1 = environment not read yet).
int odvae_conv_bf16_select_wide_tile(int on) { const int prev = g_wide_tile; g_wide_tile = on < 0 ? 0 : (on > 3 ? 3 : on); return prev; }

// reduction-channel padding (16 per MFMA k-step; the kernel walks chunks of 32 or 64) and output-channel padding of a pack
int odvae_conv_bf16_reduce_pad(int c) { return c % 64 == 0 ? c : pad_to(c, 32); }
int odvae_conv_bf16_out_pad(int c) { return c > 64 ? pad_to(c, 128) : (c > 32 ? 64 : 32); }
size_t odvae_conv_bf16_pack_elems(int reduce_c, int out_c, int taps) {
  return (size_t)taps * odvae_conv_bf16_reduce_pad(reduce_c) * odvae_conv_bf16_out_pad(out_c);
}

// w: OIHW f32 [Cout][Cin][k][k], taps = k*k in {1, 9}.  fwd_pack / dgrad_pack: bf16, odvae_conv_bf16_pack_elems(Cin, Cout, taps) /
// (Cout, Cin, taps) elements; either may be NULL.
int odvae_conv_pack_bf16(const float* w, int Cout, int Cin, int taps, void* fwd_pack, void* dgrad_pack, void* stream) {
  ODVAE_CHECK_ARG(w && Cout > 0 && Cin > 0 && (taps == 1 || taps == 9), "conv_pack_bf16: bad arguments");
  const int64_t total = (fwd_pack ? (int64_t)odvae_conv_bf16_pack_elems(Cin, Cout, taps) : 0) +
                        (dgrad_pack ? (int64_t)odvae_conv_bf16_pack_elems(Cout, Cin, taps) : 0);
  if (total == 0) return ODVAE_OK;
  const int blocks = (int)std::min<int64_t>(ceil_div64(total, 256), 4096);
  hipLaunchKernelGGL(conv_pack_bf16_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), w, Cout, Cin, taps,
                     static_cast<bf16_t*>(fwd_pack), odvae_conv_bf16_reduce_pad(Cin), odvae_conv_bf16_out_pad(Cout),
                     static_cast<bf16_t*>(dgrad_pack), odvae_conv_bf16_reduce_pad(Cout), odvae_conv_bf16_out_pad(Cin));
  ODVAE_LAUNCH_CHECK("conv_pack_bf16");
  return ODVAE_OK;
}

// y = conv(x) (+ bias) (+ residual).  x bf16 NHWC [N][Hi][Wi][Cin] (Cin % 8 == 0), pack from odvae_conv_pack_bf16 with
// (reduce = Cin, out = Cout), bias f32 [Cout] or NULL, residual bf16 [N][Ho][Wo][Cout] or NULL, y bf16 (out_f32 = 0; needs
// Cout % 4 == 0) or f32 (out_f32 = 1, any Cout).  mode 0..3 as in odvae_conv3x3_f32; mode 4 = 1x1 on [N][Hi][Wi] = [1][M/16][16].
int odvae_conv_bf16(int mode, const void* x, int N, int Hi, int Wi, int Cin, const void* pack, int Cout, const float* bias,
                    const void* residual, void* y, int Ho, int Wo, int out_f32, void* stream) {
  ODVAE_CHECK_ARG(mode >= 0 && mode <= 4, "conv_bf16: mode %d", mode);
  ODVAE_CHECK_ARG(x && pack && y && N > 0 && Hi > 0 && Wi > 0 && Cin > 0 && Cout > 0, "conv_bf16: null or empty operand");
  ODVAE_CHECK_ARG(Cin % 8 == 0, "conv_bf16: Cin = %d must be a multiple of 8 (16-byte channel vectors)", Cin);
  ODVAE_CHECK_ARG(out_f32 || Cout % 4 == 0, "conv_bf16: bf16 output needs Cout %% 4 == 0, got %d", Cout);
  ODVAE_CHECK_ARG(!residual || (!out_f32 && Cout % 4 == 0), "conv_bf16: residual needs a bf16 output with Cout %% 4 == 0");
  ODVAE_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)pack & 15) == 0 && ((uintptr_t)y & 7) == 0 && ((uintptr_t)residual & 7) == 0,
                  "conv_bf16: misaligned operand");
  int eh, ew;
  if (mode == 0 || mode == 4) { eh = Hi; ew = Wi; }
  else if (mode == 1) { eh = (Hi - 2) / 2 + 1; ew = (Wi - 2) / 2 + 1; }   // (H + 1 - 3)/2 + 1 with the (0,1,0,1) pad
  else { eh = 2 * Hi; ew = 2 * Wi; }
  ODVAE_CHECK_ARG(Ho == eh && Wo == ew, "conv_bf16(mode %d): output %dx%d does not match input %dx%d (expected %dx%d)", mode, Ho, Wo, Hi, Wi, eh, ew);
  ODVAE_CHECK_ARG((int64_t)Hi * Wi * Cin * 2 <= 0x7FFFFFF0ll && (int64_t)Ho * Wo * Cout * (out_f32 ? 4 : 2) <= 0x7FFFFFF0ll,
                  "conv_bf16: one image exceeds the 2 GiB buffer-descriptor range");
  ConvB p;
  p.x = static_cast<const bf16_t*>(x); p.wpk = static_cast<const bf16_t*>(pack); p.bias = bias;
  p.residual = static_cast<const bf16_t*>(residual); p.y = y;
  p.N = N; p.Hi = Hi; p.Wi = Wi; p.Cin = Cin; p.Ho = Ho; p.Wo = Wo; p.Cout = Cout;
  p.CinP = odvae_conv_bf16_reduce_pad(Cin); p.CoutP = odvae_conv_bf16_out_pad(Cout);
  p.tiles_x = ceil_div(Wo, TW); p.tiles_y = ceil_div(Ho, 8); p.out_f32 = out_f32;
  static const bool xcd = getenv("ODVAE_TILE_XCD") == nullptr || atoi(getenv("ODVAE_TILE_XCD")) != 0;
  p.xcd = xcd ? 1 : 0;
  ODVAE_CHECK_ARG((int64_t)N * p.tiles_x * p.tiles_y < 0x7FFFFFFFll, "conv_bf16: too many tiles");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bool k64 = p.CinP % 64 == 0;
  // ODVAE_CONV_BF16_WIDE2=1: stride-1 3x3 convs with more than 64 output channels and whole 32-channel chunks on the wide tile
  // (conv_bf16_wide_kernel).  Off by default: measured on the layers it is 3-7 % faster than the 128-pixel tile (795 vs 744 TFLOP/s at
  // 128 -> 128 @256x256, 951 vs 920 at 256 -> 256 @128x128, B=32), over a whole bf16 step the two are equal (405.6 vs 406.2 images/s at
  // 256x256, 70.6 vs 70.8 at 512x512).  What it showed (timing-only build without the in-loop halo DMA: +15 / +22 %): a wave's vector
  // memory operations complete in issue order, so every weight load issued behind the halo fetch of the chunk two ahead waits for that
  // fetch's HBM latency -- two steps of weight prefetch do not cover it.  The same holds for the register-staged halo loads of
  // conv_bf16_kernel.  The structural fix is weights through LDS as well (no per-step global loads); DESIGN.md 9.
  if (g_wide_tile < 0) g_wide_tile = getenv("ODVAE_CONV_BF16_WIDE2") ? atoi(getenv("ODVAE_CONV_BF16_WIDE2")) : 0;
  const int wide2 = g_wide_tile;
  if (mode == 0 && wide2 == 3 && Cout > 64 && Cout % 4 == 0 && Cin % 16 == 0 && Ho >= 16 && Wo >= 32) {
    p.tiles_x = ceil_div(Wo, 32);
    p.tiles_y = ceil_div(Ho, 16);
    constexpr int lds_bytes = 3 * 20 * 1024;
    static bool once_b = false;
    if (!once_b) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bf16_big_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
      once_b = true;
    }
    ODVAE_CHECK_ARG((int64_t)N * p.tiles_x * p.tiles_y < 0x7FFFFFFFll, "conv_bf16: too many tiles");
    hipLaunchKernelGGL(conv_bf16_big_kernel, dim3(N * p.tiles_x * p.tiles_y, ceil_div(Cout, 128)), dim3(256), lds_bytes, st, p);
    ODVAE_LAUNCH_CHECK("conv_bf16 (128 x 128 register tiles)");
    return ODVAE_OK;
  }
  if (mode == 0 && wide2 == 2 && Cout > 64 && Cin % 16 == 0 && Ho >= 16) {
    p.tiles_y = ceil_div(Ho, 16);
    constexpr int lds_bytes = 3 * 48 * 1024;
    static bool once_l = false;
    if (!once_l) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bf16_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
      once_l = true;
    }
    ODVAE_CHECK_ARG((int64_t)N * p.tiles_x * p.tiles_y < 0x7FFFFFFFll, "conv_bf16: too many tiles");
    hipLaunchKernelGGL(conv_bf16_lds_kernel, dim3(N * p.tiles_x * p.tiles_y, ceil_div(Cout, 128)), dim3(512), lds_bytes, st, p);
    ODVAE_LAUNCH_CHECK("conv_bf16 (all-LDS tile)");
    return ODVAE_OK;
  }
  if (mode == 0 && wide2 == 1 && Cout > 64 && Cin % 32 == 0 && Ho >= 16) {
    p.tiles_y = ceil_div(Ho, 16);
    constexpr int lds_bytes = 3 * 21 * 1024;
    static bool once = false;
    if (!once) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bf16_wide_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
      once = true;
    }
    ODVAE_CHECK_ARG((int64_t)N * p.tiles_x * p.tiles_y < 0x7FFFFFFFll, "conv_bf16: too many tiles");
    hipLaunchKernelGGL(conv_bf16_wide_kernel, dim3(N * p.tiles_x * p.tiles_y, ceil_div(Cout, 128)), dim3(256), lds_bytes, st, p);
    ODVAE_LAUNCH_CHECK("conv_bf16 (wide tile)");
    return ODVAE_OK;
  }
  switch (mode) {
    // (32-channel chunks at three blocks per CU -- __launch_bounds__(256, 3): 168 registers, 29 KB of LDS -- measured the same as
    // 64-channel chunks at two blocks per CU: 747 vs 740 TFLOP/s at 128 channels, B=32)
    case 0: if (k64) launch_by_cout<0, 64>(p, st); else launch_by_cout<0, 32>(p, st); break;
    case 1: launch_by_cout<1, 16>(p, st); break;   // 17x33 halo pixels: KC = 16 keeps the two stages under 64 KB
    case 2: launch_by_cout<2, 32>(p, st); break;
    case 3: launch_by_cout<3, 32>(p, st); break;
    default: if (k64) launch_by_cout<4, 64>(p, st); else launch_by_cout<4, 32>(p, st); break;
  }
  ODVAE_LAUNCH_CHECK("conv_bf16");
  return ODVAE_OK;
}

}  // extern "C"
