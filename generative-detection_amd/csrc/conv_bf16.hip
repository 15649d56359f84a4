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
  float* gn_partial;       // STATS launches: [N][tiles per image][gn_groups][2] = (sum, sum of squares) of y per tile and channel group
  int gn_groups, gn_cpg;   // channel groups of the GroupNorm that reads y; channels per group (a multiple of 4, <= 128)
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
// TH x 16 block tile (TH = 8 in every instantiation that is launched).
// STATS (stride-1 3x3, bf16 output): the epilogue also leaves the GroupNorm statistics of y -- of the ROUNDED bf16 values, the ones the
// GroupNorm reads -- per output tile and channel group: a lane owns a pixel and 4-channel runs, so a run's sums are a 32-lane
// butterfly, the two pixel halves of the block meet in LDS, and one thread per group writes its slot (no atomics; the consumer's
// finalize adds the tiles in f64 in a fixed order).  A template parameter: data-gradient launches and convs without a GroupNorm
// behind them run the plain build.
template <int MODE, int KC, int WCT, int WPT, int WAVES_CO, int WAVES_PX, int TH, int MINW = 1, bool STATS = false>
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
    // No guards: bias and residual come through buffer descriptors that are EMPTY when the operand is absent and end at its last element
    // otherwise (out-of-range dwords read 0).  With `p.bias ? p.bias[co] : 0` / `if (p.residual)` hipcc built 32 exec-masked loads and 16
    // residual loads each behind its own branch and each waited for with vmcnt(0) -- a chain of ~48 memory round trips in front of the
    // first MFMA of every block; now all of them are in flight together.
    const __amdgpu_buffer_rsrc_t brsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.bias ? p.bias : reinterpret_cast<const float*>(p.x)), 0, p.bias ? p.Cout * 4 : 0, 0x00020000);
    u32x4 bq[WCT][4];
    u32x2 rq[WCT][4][WPT];
#pragma unroll
    for (int ct = 0; ct < WCT; ++ct)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int co = co0 + (wco * WCT + ct) * 32 + 8 * g + 4 * h;
        bq[ct][g] = __builtin_amdgcn_raw_buffer_load_b128(brsrc, (unsigned)co * 4u, 0, 0);
#pragma unroll
        for (int pt = 0; pt < WPT; ++pt) {   // Cout % 4 == 0 is checked on the host when a residual is given
          const unsigned off = (pixoff[pt] != OOB && co < p.Cout) ? (pixoff[pt] + (unsigned)co) * 2u : OOB;
          rq[ct][g][pt] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rrsrc, off, 0, 0));
        }
      }
#pragma unroll
    for (int ct = 0; ct < WCT; ++ct)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float bv[4] = {__uint_as_float(bq[ct][g].x), __uint_as_float(bq[ct][g].y), __uint_as_float(bq[ct][g].z), __uint_as_float(bq[ct][g].w)};
#pragma unroll
        for (int pt = 0; pt < WPT; ++pt) {
          const u32x2 v = rq[ct][g][pt];
          const float rv[4] = {bf16_lo(v.x), bf16_hi(v.x), bf16_lo(v.y), bf16_hi(v.y)};
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
  float gs[WCT][4], gq[WCT][4];      // STATS: this lane's sums per (channel tile, 8-channel step) = per 4-channel run
#pragma unroll
  for (int ct = 0; ct < WCT; ++ct)
#pragma unroll
    for (int g = 0; g < 4; ++g) { gs[ct][g] = 0.f; gq[ct][g] = 0.f; }
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
          if (STATS && off != OOB) {
            const float r0 = bf16_lo(v.x), r1 = bf16_hi(v.x), r2 = bf16_lo(v.y), r3 = bf16_hi(v.y);
            gs[ct][g] += (r0 + r1) + (r2 + r3);
            gq[ct][g] += (r0 * r0 + r1 * r1) + (r2 * r2 + r3 * r3);
          }
        }
      }
    }
  if (STATS) {
    static_assert(!STATS || (WAVES_CO * WCT == 4 && WAVES_PX == 2), "statistics epilogue: 128 channels x two pixel halves per block");
    float* red = reinterpret_cast<float*>(smem);      // [wpx][run 0..31][2]; the main loop ended on a barrier, the halo stages are dead
#pragma unroll
    for (int ct = 0; ct < WCT; ++ct)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float a = gs[ct][g], b = gq[ct][g];
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }     // the 32 pixels of the lane group (h fixed)
        if (li == 0) {
          const int run = (wco * WCT + ct) * 8 + 2 * g + h;      // channels co0 + 4 run .. + 3
          red[(wpx * 32 + run) * 2 + 0] = a;
          red[(wpx * 32 + run) * 2 + 1] = b;
        }
      }
    __syncthreads();
    const int rpg = p.gn_cpg / 4;                     // 4-channel runs per group
    if (tid < 32 / rpg) {                             // thread = group of this block's 128 channels
      const int grp = co0 / p.gn_cpg + tid;
      if (grp < p.gn_groups) {
        float a = 0.f, b = 0.f;
        for (int j = 0; j < rpg; ++j)
#pragma unroll
          for (int w = 0; w < 2; ++w) { a += red[(w * 32 + tid * rpg + j) * 2]; b += red[(w * 32 + tid * rpg + j) * 2 + 1]; }
        float* dst = p.gn_partial + (((int64_t)n * (p.tiles_x * p.tiles_y) + ty * p.tiles_x + tx) * p.gn_groups + grp) * 2;
        dst[0] = a; dst[1] = b;
      }
    }
  }
}

// (Rounds 2 and 3 built three more forms of the stride-1 3x3 kernel and measured each against the one above: a 16 x 16-pixel tile with an
// LDS-DMA halo ring, +3-7 % per layer; weights AND halo through a three-stage LDS ring, -7 %; 128 x 128 register tiles with in-place
// weight refills, +2-9 % per layer -- all three equal to it over a whole step.  They were removed in round 4; what they showed -- a
// wave's vector-memory operations complete in issue order, the MFMA-only floor of this loop is 0.55 of the nominal bf16 peak at the
// clock the chip holds -- is kept in DESIGN.md 9 and profiles/r03_conv_bf16_{wide_pmc,big_tile}.txt.)

// OIHW f32 (kh x kw = 3x3 or 1x1) -> bf16 fragment packs.
//   fwd:   reduce over Cin, rows = Cout:   W[tap][co][ci]
//   dgrad: reduce over Cout, rows = Cin:   W'[tap][ci][co] = w[co][ci][flip(tap)]   (MODE 0 / 4 data gradient; MODE 3 uses
//          the same flipped pack: dx = transposed conv of dy)
// layout [tap][RP/16][OP/32][lane 64][8]: lane (r, h) element j = W[row 32*ot + r][k = 16*kt + 8h + j]
// (a batched form -- all 88 packs of a bf16 step in one launch right after the optimizer step, as the f32 Winograd packs are made -- was
// measured 0.3-0.65 ms per step SLOWER: a pack written just before its conv is read from L2, one written a whole forward earlier from HBM)
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

template <int MODE, int KC, int WCT, int WPT, int WAVES_CO, int WAVES_PX, int TH, int MINW = 1, bool STATS = false>
void launch_cfg(const ConvB& p, dim3 grid, hipStream_t st) {
  constexpr int bytes = 2 * HaloB<MODE, TH>::H * HaloB<MODE, TH>::W * (KC + 8) * 2;
  static bool once = false;
  if (!once) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bf16_kernel<MODE, KC, WCT, WPT, WAVES_CO, WAVES_PX, TH, MINW, STATS>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    once = true;
  }
  hipLaunchKernelGGL((conv_bf16_kernel<MODE, KC, WCT, WPT, WAVES_CO, WAVES_PX, TH, MINW, STATS>), grid, dim3(256), bytes, st, p);
}

template <int MODE, int KC>
void launch_by_cout(ConvB& p, hipStream_t st) {
  const int tiles = p.N * p.tiles_x * p.tiles_y;
  if constexpr (MODE == 0) {
    if (p.gn_partial) { launch_cfg<MODE, KC, 2, 2, 2, 2, 8, 1, true>(p, dim3(tiles, ceil_div(p.Cout, 128)), st); return; }
  }
  // (three blocks per CU where the kernel fits 168 registers without spilling: the stride-2, transposed and 1x1 forms, 76.2 -> 75.3 ms per
  // bf16 step.  The stride-1 3x3 form needs 176: with the hint it parks the next chunk's halo registers in scratch, 86.5 ms; in 32-channel
  // chunks it fits (162-167 registers, three blocks) and runs exactly as fast as 64-channel chunks at two blocks, 75.3-75.5 ms either way)
  constexpr int MW = (MODE == 2 || MODE == 3 || MODE == 4) ? 3 : 1;
  if (p.Cout > 64)      launch_cfg<MODE, KC, 2, 2, 2, 2, 8, MW>(p, dim3(tiles, ceil_div(p.Cout, 128)), st);
  else if (p.Cout > 32) launch_cfg<MODE, KC, 2, 1, 1, 4, 8>(p, dim3(tiles, 1), st);
  else                  launch_cfg<MODE, KC, 1, 1, 1, 4, 8>(p, dim3(tiles, 1), st);
}

}  // namespace

extern "C" {

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
static int conv_bf16_impl(int mode, const void* x, int N, int Hi, int Wi, int Cin, const void* pack, int Cout, const float* bias,
                          const void* residual, void* y, int Ho, int Wo, int out_f32, float* gn_partial, int gn_groups, void* stream);

int odvae_conv_bf16(int mode, const void* x, int N, int Hi, int Wi, int Cin, const void* pack, int Cout, const float* bias,
                    const void* residual, void* y, int Ho, int Wo, int out_f32, void* stream) {
  return conv_bf16_impl(mode, x, N, Hi, Wi, Cin, pack, Cout, bias, residual, y, Ho, Wo, out_f32, nullptr, 0, stream);
}

// tiles per image of the stride-1 3x3 kernel = the chunk count of its GroupNorm partials
int odvae_conv_bf16_stats_chunks(int H, int W) { return ceil_div(H, 8) * ceil_div(W, TW); }
// shapes whose stride-1 3x3 conv can leave the statistics: more than 64 output channels, whole groups per 128-channel block, groups of
// whole 4-channel runs
int odvae_conv_bf16_stats_supported(int Cout, int gn_groups) {
  if (gn_groups <= 0 || Cout <= 64 || Cout % gn_groups != 0) return 0;
  const int cpg = Cout / gn_groups;
  return cpg % 4 == 0 && cpg <= 128 && 128 % cpg == 0;
}

// The stride-1 3x3 conv (mode 0, bf16 output) whose epilogue also leaves the GroupNorm statistics of y for the layer that reads it:
// gn_partial [N][odvae_conv_bf16_stats_chunks(H, W)][gn_groups][2] = (sum, sum of squares) of the bf16-rounded y per tile and channel
// group, every slot written by exactly one block -- the input of odvae_groupnorm_fwd_partials_bf16 (no statistics pass).
int odvae_conv_bf16_stats(const void* x, int N, int H, int W, int Cin, const void* pack, int Cout, const float* bias, const void* residual,
                          void* y, float* gn_partial, int gn_groups, void* stream) {
  ODVAE_CHECK_ARG(gn_partial && odvae_conv_bf16_stats_supported(Cout, gn_groups), "conv_bf16_stats: Cout = %d with %d groups is not offered", Cout, gn_groups);
  return conv_bf16_impl(0, x, N, H, W, Cin, pack, Cout, bias, residual, y, H, W, 0, gn_partial, gn_groups, stream);
}

static int conv_bf16_impl(int mode, const void* x, int N, int Hi, int Wi, int Cin, const void* pack, int Cout, const float* bias,
                          const void* residual, void* y, int Ho, int Wo, int out_f32, float* gn_partial, int gn_groups, void* stream) {
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
  p.gn_partial = gn_partial; p.gn_groups = gn_groups; p.gn_cpg = gn_groups > 0 ? Cout / gn_groups : 0;
  static const bool xcd = getenv("ODVAE_TILE_XCD") == nullptr || atoi(getenv("ODVAE_TILE_XCD")) != 0;
  p.xcd = xcd ? 1 : 0;
  ODVAE_CHECK_ARG((int64_t)N * p.tiles_x * p.tiles_y < 0x7FFFFFFFll, "conv_bf16: too many tiles");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bool k64 = p.CinP % 64 == 0;
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
