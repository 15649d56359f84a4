// Weight (and bias) gradient of the 3x3 convolutions, NHWC, exact f32 on the matrix cores.
//
// dW[co][ci][kh][kw] = sum over (n, oy, ox) of xs(n,oy,ox;kh,kw)[ci] * dy[n][oy][ox][co], where xs
// is the input pixel the forward pass of that mode reads (zero outside the image):
//   MODE 0 stride 1 pad 1 | MODE 1 pad(0,1,0,1)+stride 2 | MODE 2 nearest-2x-upsample + stride 1 pad 1
// ([UPSTREAM] ldm/modules/diffusionmodules/model.py ResnetBlock / Downsample / Upsample convs; what
// loss.backward() computes for them in the reference, src/models/autoencoder.py:295-330).
//
// GEMM view per tap: M = ci, N = co, K = pixels (up to N*Ho*Wo = 2M).  A block owns a 64(ci) x 64(co)
// slice of all 9 taps (each wave 32x32 x 9 taps = 144 accumulator registers) and a contiguous range of
// 8x16 (MODE 1: 4x16) output-pixel tiles.  Per tile it stages the input halo patch [halo px][64 ci] and
// the dy tile [px][64 co] in LDS; both are read with lanes = consecutive channels (conflict-free b32),
// one dy read feeds nine MFMAs.  Blocks write partial slabs; a second kernel sums them in a fixed order
// (deterministic) and writes OIHW.  db = column sums of dy ride along in the ci-tile-0 blocks.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int TW = 16;
constexpr int BC = 64;  // channels per block on both axes

struct WgradParams {
  const float* x;    // [N][Hi][Wi][Cin]
  const float* dy;   // [N][Ho][Wo][Cout]
  float* slab;       // [nsplit][9][CinP][CoutP]
  float* bslab;      // [nsplit][CoutP] or null
  int N, Hi, Wi, Cin, Ho, Wo, Cout, CinP, CoutP;
  int tiles_x, tiles_y, ntiles, tiles_per_split, ci_tiles;
};

template <int MODE, int TH> struct Halo;
template <int TH> struct Halo<0, TH> { static constexpr int H = TH + 2, W = TW + 2; };
template <int TH> struct Halo<1, TH> { static constexpr int H = 2 * TH + 1, W = 2 * TW + 1; };
template <int TH> struct Halo<2, TH> { static constexpr int H = TH / 2 + 2, W = TW / 2 + 2; };

template <int MODE, int TH>
__device__ __forceinline__ int halo_index(int r, int c, int kh, int kw) {
  if (MODE == 0) return (r + kh) * Halo<0, TH>::W + (c + kw);
  if (MODE == 1) return (2 * r + kh) * Halo<1, TH>::W + (2 * c + kw);
  return ((r + kh + 1) >> 1) * Halo<2, TH>::W + ((c + kw + 1) >> 1);
}

template <int MODE, int TH>
__global__ __launch_bounds__(256) void conv3x3_wgrad_kernel(WgradParams p) {
  constexpr int HPIX = Halo<MODE, TH>::H * Halo<MODE, TH>::W;
  constexpr int NPIX = TH * TW;
  constexpr int HALO_F4 = HPIX * (BC / 4), HALO_IT = (HALO_F4 + 255) / 256;
  constexpr int DY_F4 = NPIX * (BC / 4), DY_IT = (DY_F4 + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Xs = smem;                 // [HPIX][BC]
  float* Ds = smem + HPIX * BC;     // [NPIX][BC]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, h = lane >> 5;
  const int split = blockIdx.x;
  const int ci_tile = blockIdx.y % p.ci_tiles, co_tile = blockIdx.y / p.ci_tiles;
  const int ci0 = ci_tile * BC, co0 = co_tile * BC;
  const bool xvec = (p.Cin & 3) == 0, dvec = (p.Cout & 3) == 0;
  const bool do_bias = p.bslab != nullptr && ci_tile == 0 && wm == 0;

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float bsum = 0.f;

  const int t_beg = split * p.tiles_per_split;
  const int t_end = min(p.ntiles, t_beg + p.tiles_per_split);
  for (int tile = t_beg; tile < t_end; ++tile) {
    int t = tile;
    const int tx = t % p.tiles_x; t /= p.tiles_x;
    const int ty = t % p.tiles_y; const int n = t / p.tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TW;
    int iy0, ix0;
    if (MODE == 0) { iy0 = oy0 - 1; ix0 = ox0 - 1; }
    else if (MODE == 1) { iy0 = 2 * oy0; ix0 = 2 * ox0; }
    else { iy0 = oy0 / 2 - 1; ix0 = ox0 / 2 - 1; }
    const float* xn = p.x + (int64_t)n * p.Hi * p.Wi * p.Cin;
    const float* dn = p.dy + (int64_t)n * p.Ho * p.Wo * p.Cout;

    __syncthreads();  // previous tile's reads are done
#pragma unroll
    for (int i = 0; i < HALO_IT; ++i) {
      const int f = tid + 256 * i;
      if (f < HALO_F4) {
        const int hp = f / (BC / 4), q = f % (BC / 4);
        const int iy = iy0 + hp / Halo<MODE, TH>::W, ix = ix0 + hp % Halo<MODE, TH>::W;
        const int c = ci0 + 4 * q;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi && c < p.Cin) {
          const float* src = xn + ((int64_t)iy * p.Wi + ix) * p.Cin + c;
          if (xvec) v = *reinterpret_cast<const float4*>(src);
          else {
            v.x = src[0];
            if (c + 1 < p.Cin) v.y = src[1];
            if (c + 2 < p.Cin) v.z = src[2];
            if (c + 3 < p.Cin) v.w = src[3];
          }
        }
        *reinterpret_cast<float4*>(Xs + hp * BC + 4 * q) = v;
      }
    }
#pragma unroll
    for (int i = 0; i < DY_IT; ++i) {
      const int f = tid + 256 * i;
      if (f < DY_F4) {
        const int px = f / (BC / 4), q = f % (BC / 4);
        const int oy = oy0 + px / TW, ox = ox0 + px % TW;
        const int c = co0 + 4 * q;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (oy < p.Ho && ox < p.Wo && c < p.Cout) {
          const float* src = dn + ((int64_t)oy * p.Wo + ox) * p.Cout + c;
          if (dvec) v = *reinterpret_cast<const float4*>(src);
          else {
            v.x = src[0];
            if (c + 1 < p.Cout) v.y = src[1];
            if (c + 2 < p.Cout) v.z = src[2];
            if (c + 3 < p.Cout) v.w = src[3];
          }
        }
        *reinterpret_cast<float4*>(Ds + px * BC + 4 * q) = v;
      }
    }
    __syncthreads();

    const float* xa = Xs + wm * 32 + li;
    const float* db = Ds + wn * 32 + li;
#pragma unroll 4
    for (int s = 0; s < NPIX / 2; ++s) {
      const int px = 2 * s + h;
      const int r = px / TW, c = px % TW;
      const float b = db[px * BC];
      if (do_bias) bsum += b;
#pragma unroll
      for (int t9 = 0; t9 < 9; ++t9) {
        const float a = xa[halo_index<MODE, TH>(r, c, t9 / 3, t9 % 3) * BC];
        acc[t9] = mfma32(a, b, acc[t9]);
      }
    }
  }

  // partial slab: [split][tap][ci][co], lane = co
  const int co = co0 + wn * 32 + li;
  float* sl = p.slab + (int64_t)split * 9 * p.CinP * p.CoutP;
#pragma unroll
  for (int t9 = 0; t9 < 9; ++t9)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ci = ci0 + wm * 32 + acc_row(r, lane);
      sl[((int64_t)t9 * p.CinP + ci) * p.CoutP + co] = acc[t9][r];
    }
  if (do_bias) {
    bsum += __shfl_xor(bsum, 32, 64);
    if (h == 0) p.bslab[(int64_t)split * p.CoutP + co] = bsum;
  }
}

// ---- v2: LDS-DMA staging, double-buffered -------------------------------------------------------------------
// Block = 512 threads = 8 waves (2 ci x 4 co sub-tiles of 32x32, all 9 taps each): 64 ci x 128 co per block.
// The halo patch [HPIX][64 ci] and the dy tile [NPIX][128 co] of the NEXT pixel tile are fetched by
// buffer_load ... lds (1 KiB per wave-instruction, straight into LDS, no staging registers; out-of-image lanes
// point past the buffer's num_records and therefore write zeros) while the current tile is multiplied:
// one barrier per tile, every byte of a tile in flight at once, 2 waves per SIMD.  Needs Cin % 4 == Cout % 4 == 0.
constexpr int BCI2 = 64, BCO2 = 128;

template <int MODE, int TH>
struct DmaGeom {
  static constexpr int HPIX = Halo<MODE, TH>::H * Halo<MODE, TH>::W;
  static constexpr int NPIX = TH * TW;
  static constexpr int NH = (HPIX * (BCI2 / 4) + 63) / 64;   // 1 KiB pieces of the halo image
  static constexpr int ND = NPIX * (BCO2 / 4) / 64;          // 1 KiB pieces of the dy image
  static constexpr int XS_F = NH * 256;                      // floats (halo image padded to whole pieces)
  static constexpr int BUF_F = XS_F + NPIX * BCO2;
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int MODE, int TH>
__global__ __launch_bounds__(512, 2) void conv3x3_wgrad_dma_kernel(WgradParams p) {
  using G = DmaGeom<MODE, TH>;
  extern __shared__ __attribute__((aligned(16))) float smem[];   // 2 * BUF_F floats

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int li = lane & 31, h = lane >> 5;
  const int split = blockIdx.x;
  const int ci_tile = blockIdx.y % p.ci_tiles, co_tile = blockIdx.y / p.ci_tiles;
  const int ci0 = ci_tile * BCI2, co0 = co_tile * BCO2;
  const bool do_bias = p.bslab != nullptr && ci_tile == 0 && wm == 0;

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float bsum = 0.f;

  auto issue = [&](int tile, int buf) {
    int t = tile;
    const int tx = t % p.tiles_x; t /= p.tiles_x;
    const int ty = t % p.tiles_y; const int n = t / p.tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TW;
    int iy0, ix0;
    if (MODE == 0) { iy0 = oy0 - 1; ix0 = ox0 - 1; }
    else if (MODE == 1) { iy0 = 2 * oy0; ix0 = 2 * ox0; }
    else { iy0 = oy0 / 2 - 1; ix0 = ox0 / 2 - 1; }
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.x + (int64_t)n * p.Hi * p.Wi * p.Cin), 0, p.Hi * p.Wi * p.Cin * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.dy + (int64_t)n * p.Ho * p.Wo * p.Cout), 0, p.Ho * p.Wo * p.Cout * 4, 0x00020000);
    float* base = smem + buf * G::BUF_F;
    const unsigned OOB = 0x7FFFFFF0u;   // >= num_records: the load returns 0
    for (int j = wave; j < G::NH + G::ND; j += 8) {
      if (j < G::NH) {
        const int f = j * 64 + lane;
        const int hp = f >> 4, q = f & 15;
        const int iy = iy0 + hp / Halo<MODE, TH>::W, ix = ix0 + hp % Halo<MODE, TH>::W;
        const int c = ci0 + 4 * q;
        const bool ok = hp < G::HPIX && iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi && c < p.Cin;
        const unsigned voff = ok ? (unsigned)(((iy * p.Wi + ix) * p.Cin + c) * 4) : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(base + j * 256), 16, voff, 0, 0, 0);
      } else {
        const int jj = j - G::NH;
        const int f = jj * 64 + lane;
        const int px = f >> 5, q = f & 31;
        const int oy = oy0 + px / TW, ox = ox0 + px % TW;
        const int c = co0 + 4 * q;
        const bool ok = oy < p.Ho && ox < p.Wo && c < p.Cout;
        const unsigned voff = ok ? (unsigned)(((oy * p.Wo + ox) * p.Cout + c) * 4) : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rd, (lds_ptr_t)(base + G::XS_F + jj * 256), 16, voff, 0, 0, 0);
      }
    }
  };

  const int t_beg = split * p.tiles_per_split;
  const int ntl = min(p.ntiles, t_beg + p.tiles_per_split) - t_beg;
  if (ntl > 0) issue(t_beg, 0);
  for (int it = 0; it < ntl; ++it) {
    const int cur = it & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of tile `it` have landed
    __syncthreads();                                    // everyone's have; buffer cur^1 is free again
    if (it + 1 < ntl) issue(t_beg + it + 1, cur ^ 1);
    // lane half h takes pixel 2s+h of k-step s; TW is even, so both halves sit in the same tile row
    const float* xa = smem + cur * G::BUF_F + wm * 32 + li + h * BCI2;            // [HPIX][64]
    const float* db = smem + cur * G::BUF_F + G::XS_F + wn * 32 + li + h * BCO2;  // [NPIX][128]
    auto load_step = [&](int s, float (&a)[9], float& b) {
      const int r = (2 * s) / TW, c = (2 * s) % TW;   // compile-time after unrolling
      b = db[2 * s * BCO2];
#pragma unroll
      for (int t9 = 0; t9 < 9; ++t9) {
        // halo_index is affine in c for every mode's even/odd pixel pair except MODE 2 (c >> 1): handle h there
        int idx;
        if (MODE == 2) idx = ((r + t9 / 3 + 1) >> 1) * Halo<2, TH>::W;   // column part added per lane below
        else idx = halo_index<MODE, TH>(r, c, t9 / 3, t9 % 3);
        if (MODE == 2) a[t9] = (xa - h * BCI2)[(idx + ((c + h + t9 % 3 + 1) >> 1)) * BCI2];
        else a[t9] = xa[idx * BCI2 + (MODE == 1 ? h * BCI2 : 0)];        // MODE 1: pixel step is 2 halo columns
      }
    };
    float ac[9], an[9], bc, bn = 0.f;
    load_step(0, ac, bc);
#pragma unroll
    for (int s = 0; s < G::NPIX / 2; ++s) {
      // software pipeline, pinned: k-step s+1's ten LDS reads are in flight behind step s's nine MFMAs
      if (s + 1 < G::NPIX / 2) load_step(s + 1, an, bn);
      __builtin_amdgcn_sched_barrier(0);
      if (do_bias) bsum += bc;
#pragma unroll
      for (int t9 = 0; t9 < 9; ++t9) acc[t9] = mfma32(ac[t9], bc, acc[t9]);
      __builtin_amdgcn_sched_barrier(0);
      bc = bn;
#pragma unroll
      for (int t9 = 0; t9 < 9; ++t9) ac[t9] = an[t9];
    }
  }

  // slab stores: 32-bit offsets into this split's slab; the tap stride goes into the scalar offset, so a store costs
  // no per-element address arithmetic (144 stores per lane)
  const int co = co0 + wn * 32 + li;
  float* sl = p.slab + (int64_t)split * 9 * p.CinP * p.CoutP;
  const __amdgpu_buffer_rsrc_t srsrc = __builtin_amdgcn_make_buffer_rsrc(sl, 0, 9 * p.CinP * p.CoutP * 4, 0x00020000);
  unsigned rowoff[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) rowoff[r] = (unsigned)((ci0 + wm * 32 + acc_row(r, lane)) * p.CoutP + co) * 4u;
  const int tap_stride = p.CinP * p.CoutP * 4;
#pragma unroll
  for (int t9 = 0; t9 < 9; ++t9)
#pragma unroll
    for (int r = 0; r < 16; ++r)
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[t9][r]), srsrc, rowoff[r], t9 * tap_stride, 0);
  if (do_bias) {
    bsum += __shfl_xor(bsum, 32, 64);
    if (h == 0) p.bslab[(int64_t)split * p.CoutP + co] = bsum;
  }
}

__global__ void conv3x3_wgrad_reduce_kernel(const float* __restrict__ slab, const float* __restrict__ bslab,
                                            int nsplit, int Cin, int Cout, int CinP, int CoutP,
                                            float* __restrict__ dw, float* __restrict__ dbias) {
  const int64_t per = (int64_t)9 * CinP * CoutP;
  const int64_t total = per + (dbias ? CoutP : 0);
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    if (idx < per) {
      const int co = (int)(idx % CoutP);
      const int ci = (int)((idx / CoutP) % CinP);
      const int tap = (int)(idx / ((int64_t)CoutP * CinP));
      if (co < Cout && ci < Cin) {
        float sk[4] = {0.f, 0.f, 0.f, 0.f};      // four independent chains: one chain of nsplit dependent round trips otherwise
        int sp = 0;
        for (; sp + 3 < nsplit; sp += 4) {
#pragma unroll
          for (int j = 0; j < 4; ++j) sk[j] += slab[(sp + j) * per + idx];
        }
        for (; sp < nsplit; ++sp) sk[0] += slab[sp * per + idx];
        dw[((int64_t)co * Cin + ci) * 9 + tap] = (sk[0] + sk[1]) + (sk[2] + sk[3]);
      }
    } else {
      const int co = (int)(idx - per);
      if (co < Cout) {
        float s = 0.f;
        for (int sp = 0; sp < nsplit; ++sp) s += bslab[(int64_t)sp * CoutP + co];
        dbias[co] = s;
      }
    }
  }
}

// ---- Upsample conv (nearest 2x + 3x3) by output parity class ---------------------------------------------------
// dWeff[cls][a][b] = sum over low-res pixels (y, x) of x[y + py - 1 + a][x + px - 1 + b]^T dy[2y + py][2x + px], cls = 2 py + px:
// 16 tap-products per low-res pixel instead of the 36 of the dense form (the forward pass multiplies the same
// pre-summed weights, conv3x3_f32.hip mode 5); dW[kh][kw] is the sum of the four dWeff entries whose pre-sum contains
// W[kh][kw].  Block = 64 ci x 64 co, 8 waves = 4 classes x 2 co sub-tiles, each wave both ci sub-tiles x 4 taps
// (128 accumulator registers); pixel tile = 2 x 16 low-res pixels: halo [4 x 18][64 ci] and the four parity planes of
// the 4 x 32 dy patch [cls][32 px][64 co] arrive by LDS-DMA, double-buffered, one barrier per tile.
constexpr int UP_TH = 2, UP_BC = 64;
struct UpGeom {
  static constexpr int HALO_W = TW + 2, HPIX = (UP_TH + 2) * HALO_W;   // 72
  static constexpr int NPIX = UP_TH * TW;                             // 32 low-res pixels
  static constexpr int NH = (HPIX * (UP_BC / 4) + 63) / 64;           // 1 KiB pieces of the halo image
  static constexpr int ND = 4 * NPIX * (UP_BC / 4) / 64;              // 1 KiB pieces of the dy planes
  static constexpr int XS_F = NH * 256;
  static constexpr int BUF_F = XS_F + 4 * NPIX * UP_BC;
};

__global__ __launch_bounds__(512, 2) void conv3x3_wgrad_up_kernel(WgradParams p) {
  using G = UpGeom;
  extern __shared__ __attribute__((aligned(16))) float smem[];   // 2 * BUF_F floats

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cls = wave >> 1, wn = wave & 1, py = cls >> 1, px = cls & 1;
  const int li = lane & 31, h = lane >> 5;
  const int split = blockIdx.x;
  const int ci_tile = blockIdx.y % p.ci_tiles, co_tile = blockIdx.y / p.ci_tiles;
  const int ci0 = ci_tile * UP_BC, co0 = co_tile * UP_BC;
  const bool do_bias = p.bslab != nullptr && ci_tile == 0;

  f32x16 acc[2][4];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][t][r] = 0.f;
  float bsum = 0.f;

  auto issue = [&](int tile, int buf) {
    int t = tile;
    const int tx = t % p.tiles_x; t /= p.tiles_x;
    const int ty = t % p.tiles_y; const int n = t / p.tiles_y;
    const int ly0 = ty * UP_TH, lx0 = tx * TW;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.x + (int64_t)n * p.Hi * p.Wi * p.Cin), 0, p.Hi * p.Wi * p.Cin * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.dy + (int64_t)n * p.Ho * p.Wo * p.Cout), 0, p.Ho * p.Wo * p.Cout * 4, 0x00020000);
    float* base = smem + buf * G::BUF_F;
    const unsigned OOB = 0x7FFFFFF0u;   // >= num_records: the load returns 0
    for (int j = wave; j < G::NH + G::ND; j += 8) {
      if (j < G::NH) {
        const int f = j * 64 + lane;
        const int hp = f >> 4, q = f & 15;
        const int iy = ly0 - 1 + hp / G::HALO_W, ix = lx0 - 1 + hp % G::HALO_W;
        const int c = ci0 + 4 * q;
        const bool ok = hp < G::HPIX && iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi && c < p.Cin;
        const unsigned voff = ok ? (unsigned)(((iy * p.Wi + ix) * p.Cin + c) * 4) : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(base + j * 256), 16, voff, 0, 0, 0);
      } else {
        const int jj = j - G::NH;
        const int f = jj * 64 + lane;
        const int pp = f >> 4, q = f & 15;              // pp = plane * 32 + low-res pixel of the tile
        const int plane = pp >> 5, pl = pp & 31;
        const int oy = 2 * (ly0 + pl / TW) + (plane >> 1), ox = 2 * (lx0 + pl % TW) + (plane & 1);
        const int c = co0 + 4 * q;
        const bool ok = oy < p.Ho && ox < p.Wo && c < p.Cout;
        const unsigned voff = ok ? (unsigned)(((oy * p.Wo + ox) * p.Cout + c) * 4) : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rd, (lds_ptr_t)(base + G::XS_F + jj * 256), 16, voff, 0, 0, 0);
      }
    }
  };

  const int t_beg = split * p.tiles_per_split;
  const int ntl = min(p.ntiles, t_beg + p.tiles_per_split) - t_beg;
  if (ntl > 0) issue(t_beg, 0);
  for (int it = 0; it < ntl; ++it) {
    const int cur = it & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of tile `it` have landed
    __syncthreads();                                    // everyone's have; buffer cur^1 is free again
    if (it + 1 < ntl) issue(t_beg + it + 1, cur ^ 1);
    // lane half h takes low-res pixel 2s+h of k-step s (same tile row: TW is even); the class offset is wave-uniform
    const float* xa = smem + cur * G::BUF_F + ((py * G::HALO_W + px) + h) * UP_BC + li;                 // [HPIX][64]
    const float* db = smem + cur * G::BUF_F + G::XS_F + (cls * G::NPIX + h) * UP_BC + wn * 32 + li;     // [4][32][64]
    auto load_step = [&](int s, float (&a)[2][4], float& b) {
      const int r = (2 * s) / TW, c = (2 * s) % TW;   // compile-time after unrolling
      b = db[2 * s * UP_BC];
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) a[mt][t] = xa[((r + t / 2) * G::HALO_W + c + t % 2) * UP_BC + mt * 32];
    };
    float ac[2][4], an[2][4], bc, bn = 0.f;
    load_step(0, ac, bc);
#pragma unroll
    for (int s = 0; s < G::NPIX / 2; ++s) {
      if (s + 1 < G::NPIX / 2) load_step(s + 1, an, bn);
      __builtin_amdgcn_sched_barrier(0);
      bsum += bc;
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) acc[mt][t] = mfma32(ac[mt][t], bc, acc[mt][t]);
      __builtin_amdgcn_sched_barrier(0);
      bc = bn;
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) ac[mt][t] = an[mt][t];
    }
  }

  // slab [split][cls * 4 + tap][CinP][CoutP]
  const int co = co0 + wn * 32 + li;
  float* sl = p.slab + (int64_t)split * 16 * p.CinP * p.CoutP;
  const __amdgpu_buffer_rsrc_t srsrc = __builtin_amdgcn_make_buffer_rsrc(sl, 0, 16 * p.CinP * p.CoutP * 4, 0x00020000);
  const int tap_stride = p.CinP * p.CoutP * 4;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const unsigned rowoff = (unsigned)((ci0 + mt * 32 + acc_row(r, lane)) * p.CoutP + co) * 4u;
#pragma unroll
      for (int t = 0; t < 4; ++t)
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[mt][t][r]), srsrc, rowoff, (cls * 4 + t) * tap_stride, 0);
    }
  if (do_bias) {
    bsum += __shfl_xor(bsum, 32, 64);
    if (h == 0) p.bslab[((int64_t)split * 4 + cls) * p.CoutP + co] = bsum;
  }
}

// dW[kh][kw] = sum of the dWeff entries (py, a) x (px, b) with kh in R(py, a), kw in R(px, b):
// kh = 0: (0,0) (1,0) | kh = 1: (0,1) (1,0) | kh = 2: (0,1) (1,1); splits summed in a fixed order
__global__ void conv3x3_wgrad_up_reduce_kernel(const float* __restrict__ slab, const float* __restrict__ bslab,
                                               int nsplit, int Cin, int Cout, int CinP, int CoutP,
                                               float* __restrict__ dw, float* __restrict__ dbias) {
  const int64_t mat = (int64_t)CinP * CoutP;
  const int64_t per = 9 * mat;
  const int64_t total = per + (dbias ? CoutP : 0);
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    if (idx < per) {
      const int co = (int)(idx % CoutP);
      const int ci = (int)((idx / CoutP) % CinP);
      const int tap = (int)(idx / mat);
      if (co < Cout && ci < Cin) {
        const int kh = tap / 3, kw = tap % 3;
        const int pa[2][2] = {{0, kh == 0 ? 0 : 1}, {1, kh == 2 ? 1 : 0}};   // (py, a) pairs for kh
        const int pb[2][2] = {{0, kw == 0 ? 0 : 1}, {1, kw == 2 ? 1 : 0}};   // (px, b) pairs for kw
        float sq[2][2] = {{0.f, 0.f}, {0.f, 0.f}};      // one chain per (i, j) term: four loads in flight per split instead of one chain
        for (int sp = 0; sp < nsplit; ++sp)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              const int t16 = (pa[i][0] * 2 + pb[j][0]) * 4 + pa[i][1] * 2 + pb[j][1];
              sq[i][j] += slab[((int64_t)sp * 16 + t16) * mat + (int64_t)ci * CoutP + co];
            }
        dw[((int64_t)co * Cin + ci) * 9 + tap] = (sq[0][0] + sq[0][1]) + (sq[1][0] + sq[1][1]);
      }
    } else {
      const int co = (int)(idx - per);
      if (co < Cout) {
        float s = 0.f;
        for (int sp = 0; sp < nsplit * 4; ++sp) s += bslab[(int64_t)sp * CoutP + co];
        dbias[co] = s;
      }
    }
  }
}

// ---- thin side: conv_in (3 -> C) and conv_out (C -> 3) ----------------------------------------------------------
// With <= 3 channels on one side the nine taps of that side fit ONE 32-wide MFMA operand: out[c][j] = sum over pixels
// of big[pixel][c] * small[pixel shifted by tap(j)][cs(j)], j = tap * Cs + cs < 27 -- a [C x pixels] x [pixels x 32]
// product that reads the wide tensor exactly once (HBM-bound: 1 GiB at 128 ch, 256^2, B=32).
//   conv_in : big = dy, small = x read at (y + kh - 1, x + kw - 1); column 27 multiplies 1.0, i.e. the bias gradient
//   conv_out: big = x,  small = dy read at (y - kh + 1, x - kw + 1)   (the same sum re-indexed by the input pixel)
// No LDS, no barriers: A fragments are coalesced 128-byte channel rows, B fragments gathers from the 25 MB tensor (cache
// resident); a wave owns whole image rows, eight k-steps (16 pixels) of loads in flight behind eight steps of MFMAs.
struct ThinParams {
  const float* big;     // [N][H][W][Cb]
  const float* small;   // [N][H][W][Cs]
  float* slab;          // [waves][CbP][32]
  int N, H, W, Cb, Cs, CbP, sign, ones_col;   // ones_col: column index that multiplies 1.0 (bias gradient), or -1
};

// One 16-byte load per lane is a pixel's channels 4i..4i+3: component t feeds channel tile t, whose row i is therefore
// channel c0 + 4i + t (any channel order will do, the slab store undoes it) -- a wave reads whole 512-byte pixel rows.
__global__ __launch_bounds__(256) void conv3x3_wgrad_thin_kernel(ThinParams p) {
  const int lane = threadIdx.x & 63, li = lane & 31, kk = lane >> 5;
  const int wave_in_grid = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
  const int c0 = blockIdx.y * 128;
  // this lane's B column: tap and channel of the thin side
  const int j = li, tap = j / p.Cs, cs = j % p.Cs;
  const bool jtap = j < 9 * p.Cs;
  const int dyj = p.sign * (tap / 3 - 1), dxj = p.sign * (tap % 3 - 1);
  const bool ones = j == p.ones_col;
  // channel quads past Cb (Cb not a multiple of 128): clamped address + select, never a guarded load
  const bool cok = c0 + 4 * li < p.Cb;
  const int coff = min(c0 + 4 * li, p.Cb - 4);
  f32x16 acc[4];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;

  const int rows = p.N * p.H;
  constexpr int CH = 8;   // k-steps per chunk (16 pixels)
  float4 a_cur[CH], a_nxt[CH];
  float b_cur[CH], b_nxt[CH];
  for (int row = wave_in_grid; row < rows; row += nwaves) {
    const int y = row % p.H;
    const int ys = y + dyj;
    const bool rowok = jtap && ys >= 0 && ys < p.H;
    const float* srow = p.small + ((int64_t)(row - y + min(max(ys, 0), p.H - 1)) * p.W) * p.Cs + cs;
    const float* brow = p.big + (int64_t)row * p.W * p.Cb + coff;
    auto load_chunk = [&](int x0, float4 (&a)[CH], float (&b)[CH]) {
#pragma unroll
      for (int s = 0; s < CH; ++s) {
        const int x = x0 + 2 * s + kk;
        const float4 v4 = *reinterpret_cast<const float4*>(brow + (int64_t)x * p.Cb);
        a[s] = cok ? v4 : make_float4(0.f, 0.f, 0.f, 0.f);
        const int xs = x + dxj;
        const float v = srow[(int64_t)min(max(xs, 0), p.W - 1) * p.Cs];
        b[s] = ones ? 1.f : ((rowok && xs >= 0 && xs < p.W) ? v : 0.f);
      }
    };
    load_chunk(0, a_cur, b_cur);
    for (int x0 = 0; x0 < p.W; x0 += 2 * CH) {
      if (x0 + 2 * CH < p.W) load_chunk(x0 + 2 * CH, a_nxt, b_nxt);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < CH; ++s) {
        acc[0] = mfma32(a_cur[s].x, b_cur[s], acc[0]);
        acc[1] = mfma32(a_cur[s].y, b_cur[s], acc[1]);
        acc[2] = mfma32(a_cur[s].z, b_cur[s], acc[2]);
        acc[3] = mfma32(a_cur[s].w, b_cur[s], acc[3]);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < CH; ++s) { b_cur[s] = b_nxt[s]; a_cur[s] = a_nxt[s]; }
    }
  }
  float* sl = p.slab + (int64_t)wave_in_grid * p.CbP * 32;
#pragma unroll
  for (int ct = 0; ct < 4; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int c = c0 + 4 * acc_row(r, lane) + ct;
      if (c < p.Cb) sl[c * 32 + li] = acc[ct][r];
    }
}

// partial column sums of the thin tensor (bias gradient of conv_out): part[block][Cs]
__global__ __launch_bounds__(256) void thin_colsum_kernel(const float* __restrict__ t, int64_t npix, int Cs, float* __restrict__ part) {
  __shared__ float red[4][3];
  float s[3] = {0.f, 0.f, 0.f};
  for (int64_t px = (int64_t)blockIdx.x * 256 + threadIdx.x; px < npix; px += (int64_t)gridDim.x * 256)
    for (int c = 0; c < Cs; ++c) s[c] += t[px * Cs + c];
  for (int c = 0; c < 3; ++c) s[c] = wave_sum(s[c]);
  if ((threadIdx.x & 63) == 0) for (int c = 0; c < 3; ++c) red[threadIdx.x >> 6][c] = s[c];
  __syncthreads();
  if (threadIdx.x < Cs) part[blockIdx.x * Cs + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// sum the per-wave slabs in a fixed order and scatter into OIHW; sign +1: c = co, thin = ci; sign -1: c = ci, thin = co
__global__ void conv3x3_wgrad_thin_reduce_kernel(const float* __restrict__ slab, int nwaves, int Cb, int Cs, int CbP, int sign,
                                                 const float* __restrict__ bpart, int nbpart, int Cin, int Cout,
                                                 float* __restrict__ dw, float* __restrict__ dbias) {
  // block = 16 outputs x 16 parts: part q sums slabs q, q+16, ... (eight loads in flight), then the 16 parts are added in
  // a fixed order -- 2048 slabs are far too long a chain for one thread per output
  __shared__ float red[16][17];
  const int part = threadIdx.x >> 4;
  const int idx = blockIdx.x * 16 + (threadIdx.x & 15);
  if (idx < Cb * 32) {
    const int c = idx / 32, j = idx % 32;
    float ps = 0.f;
    const float* src = slab + (int64_t)c * 32 + j;
    const int64_t wstride = (int64_t)CbP * 32;
    int w = part;
    for (; w + 7 * 16 < nwaves; w += 8 * 16) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = src[(w + 16 * u) * wstride];
#pragma unroll
      for (int u = 0; u < 8; ++u) ps += v[u];
    }
    for (; w < nwaves; w += 16) ps += src[w * wstride];
    red[part][threadIdx.x & 15] = ps;
  }
  __syncthreads();
  if (part == 0 && idx < Cb * 32) {
    const int c = idx / 32, j = idx % 32;
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) s += red[q][threadIdx.x & 15];
    if (j < 9 * Cs) {
      const int tap = j / Cs, cs = j % Cs;
      const int co = sign > 0 ? c : cs, ci = sign > 0 ? cs : c;
      dw[((int64_t)co * Cin + ci) * 9 + tap] = s;
    } else if (sign > 0 && dbias && j == 9 * Cs) dbias[c] = s;
  } else if (part == 0 && sign < 0 && dbias && idx >= Cb * 32 && idx - Cb * 32 < Cout) {
    const int co = idx - Cb * 32;
    float s = 0.f;
    for (int b = 0; b < nbpart; ++b) s += bpart[b * Cs + co];
    dbias[co] = s;
  }
}

struct Plan { int v2, up, th, tiles_x, tiles_y, ntiles, nsplit, tiles_per_split, CinP, CoutP, ci_tiles, co_tiles; };

// v2 (LDS-DMA) whenever both channel counts allow 16-byte pieces and the 128-wide co tile is not mostly padding;
// ODVAE_WGRAD_V1=1 forces the register-staged kernel (in-process A/B)
Plan make_plan(int mode, int N, int Ho, int Wo, int Cin, int Cout) {
  static const bool force_v1 = getenv("ODVAE_WGRAD_V1") != nullptr;
  Plan pl;
  pl.up = mode == 5 && !force_v1 && Cin % 4 == 0 && Cout % 4 == 0;
  if (pl.up) {   // tiles of 2 x 16 LOW-RES pixels (Ho, Wo are the output's: twice the input's)
    pl.v2 = 1; pl.th = UP_TH;
    pl.tiles_x = ceil_div(Wo / 2, TW); pl.tiles_y = ceil_div(Ho / 2, UP_TH);
    pl.ntiles = pl.tiles_x * pl.tiles_y * N;
    pl.CinP = ceil_div(Cin, UP_BC) * UP_BC; pl.CoutP = ceil_div(Cout, UP_BC) * UP_BC;
    pl.ci_tiles = pl.CinP / UP_BC; pl.co_tiles = pl.CoutP / UP_BC;
    int nsplit = ceil_div(256, pl.ci_tiles * pl.co_tiles);   // one 8-wave block per CU, one round: fewer slabs to write and sum
    if (nsplit > pl.ntiles) nsplit = pl.ntiles;
    if (nsplit < 1) nsplit = 1;
    pl.tiles_per_split = ceil_div(pl.ntiles, nsplit);
    pl.nsplit = ceil_div(pl.ntiles, pl.tiles_per_split);
    return pl;
  }
  if (mode == 5) mode = 2;   // channel counts the DMA kernel cannot take: dense form
  pl.v2 = !force_v1 && Cin % 4 == 0 && Cout % 4 == 0 && Cout > 64;
  const int bci = pl.v2 ? BCI2 : BC, bco = pl.v2 ? BCO2 : BC;
  pl.th = pl.v2 ? (mode == 1 ? 2 : 4) : (mode == 1 ? 4 : 8);
  pl.tiles_x = ceil_div(Wo, TW);
  pl.tiles_y = ceil_div(Ho, pl.th);
  pl.ntiles = pl.tiles_x * pl.tiles_y * N;
  pl.CinP = ceil_div(Cin, bci) * bci;
  pl.CoutP = ceil_div(Cout, bco) * bco;
  pl.ci_tiles = pl.CinP / bci;
  pl.co_tiles = pl.CoutP / bco;
  const int ctiles = pl.ci_tiles * pl.co_tiles;
  static const int v2_blocks = getenv("ODVAE_WGRAD_BLOCKS") ? atoi(getenv("ODVAE_WGRAD_BLOCKS")) : 256;   // one block per CU, one round (512: -3..-10 %)
  int nsplit = ceil_div(pl.v2 ? v2_blocks : 1024, ctiles);   // v2 runs one 8-wave block per CU
  if (nsplit > pl.ntiles) nsplit = pl.ntiles;
  if (nsplit < 1) nsplit = 1;
  pl.tiles_per_split = ceil_div(pl.ntiles, nsplit);
  pl.nsplit = ceil_div(pl.ntiles, pl.tiles_per_split);
  return pl;
}

// thin-side kernel: stride-1 conv whose input or output has <= 3 channels (conv_in / conv_out)
constexpr int THIN_BLOCKS = 1024, THIN_BIAS_BLOCKS = 256;
bool thin_applies(int mode, int Hi, int Wi, int Cin, int Cout) {
  static const bool off = getenv("ODVAE_WGRAD_V1") != nullptr;
  const int cb = Cin <= 3 ? Cout : Cin;
  return !off && mode == 0 && (Cin <= 3) != (Cout <= 3) && Wi % 16 == 0 && cb % 4 == 0 && cb >= 32 &&
         (int64_t)Hi * Wi * cb * 4 < 0x7FFFFFF0ll;
}
size_t thin_workspace_floats(int Cin, int Cout) {
  const int cb = Cin <= 3 ? Cout : Cin;
  return (size_t)THIN_BLOCKS * 4 * cb * 32 + (size_t)THIN_BIAS_BLOCKS * 3;
}

template <int MODE, int TH>
size_t wgrad_smem_bytes() { return (size_t)(Halo<MODE, TH>::H * Halo<MODE, TH>::W + TH * TW) * BC * sizeof(float); }

template <typename K>
hipError_t launch_dyn(K kernel, dim3 grid, dim3 block, size_t smem, hipStream_t st, const WgradParams& p) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kernel, grid, block, smem, st, p);
  return hipSuccess;
}

}  // namespace

extern "C" {

size_t odvae_conv3x3_wgrad_workspace_bytes(int mode, int N, int Ho, int Wo, int Cin, int Cout) {
  if (thin_applies(mode, Ho, Wo, Cin, Cout)) return thin_workspace_floats(Cin, Cout) * sizeof(float);
  const Plan pl = make_plan(mode, N, Ho, Wo, Cin, Cout);
  if (pl.up) return ((size_t)pl.nsplit * 16 * pl.CinP * pl.CoutP + (size_t)pl.nsplit * 4 * pl.CoutP) * sizeof(float);
  return ((size_t)pl.nsplit * 9 * pl.CinP * pl.CoutP + (size_t)pl.nsplit * pl.CoutP) * sizeof(float);
}

// dw: OIHW [Cout][Cin][3][3] (overwritten).  dbias: [Cout] or null.
int odvae_conv3x3_wgrad_f32(int mode, const float* x, const float* dy, int N, int Hi, int Wi, int Cin,
                            int Ho, int Wo, int Cout, float* dw, float* dbias,
                            void* workspace, size_t workspace_bytes, void* stream) {
  ODVAE_CHECK_ARG(x && dy && dw, "conv3x3_wgrad: null operand");
  ODVAE_CHECK_ARG((mode >= 0 && mode <= 2) || mode == 5, "conv3x3_wgrad: mode %d", mode);
  ODVAE_CHECK_ARG(N > 0 && Hi > 0 && Wi > 0 && Cin > 0 && Cout > 0, "conv3x3_wgrad: empty shape");
  if (mode == 0) ODVAE_CHECK_ARG(Ho == Hi && Wo == Wi, "conv3x3_wgrad mode 0: Ho,Wo must equal Hi,Wi");
  if (mode == 1) ODVAE_CHECK_ARG(Hi % 2 == 0 && Wi % 2 == 0 && Ho == Hi / 2 && Wo == Wi / 2, "conv3x3_wgrad mode 1: need even Hi,Wi");
  if (mode == 2 || mode == 5) ODVAE_CHECK_ARG(Ho == 2 * Hi && Wo == 2 * Wi, "conv3x3_wgrad mode %d: need Ho=2*Hi", mode);
  ODVAE_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)dy & 15) == 0, "conv3x3_wgrad: x/dy must be 16-byte aligned");
  ODVAE_CHECK_ARG((int64_t)Hi * Wi * Cin * 4 < 0x7FFFFFF0ll && (int64_t)Ho * Wo * Cout * 4 < 0x7FFFFFF0ll, "conv3x3_wgrad: one image must stay below 2 GiB");
  const Plan pl = make_plan(mode, N, Ho, Wo, Cin, Cout);
  const size_t need = odvae_conv3x3_wgrad_workspace_bytes(mode, N, Ho, Wo, Cin, Cout);
  if (!workspace || workspace_bytes < need) {
    odvae_set_error("conv3x3_wgrad: needs %zu workspace bytes, got %zu", need, workspace_bytes);
    return ODVAE_ERR_WORKSPACE;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (thin_applies(mode, Hi, Wi, Cin, Cout)) {
    ThinParams t;
    const bool in_thin = Cin <= 3;   // conv_in: big = dy, small = x; conv_out: big = x, small = dy
    t.big = in_thin ? dy : x; t.small = in_thin ? x : dy;
    t.slab = static_cast<float*>(workspace);
    t.N = N; t.H = Hi; t.W = Wi; t.Cb = in_thin ? Cout : Cin; t.Cs = in_thin ? Cin : Cout; t.CbP = t.Cb;
    t.sign = in_thin ? 1 : -1; t.ones_col = (in_thin && dbias) ? 9 * t.Cs : -1;
    const int groups = ceil_div(t.Cb, 128);
    const int blocks = std::max(1, std::min(THIN_BLOCKS / groups, ceil_div(N * Hi, 4)));
    hipLaunchKernelGGL(conv3x3_wgrad_thin_kernel, dim3(blocks, groups), dim3(256), 0, st, t);
    ODVAE_LAUNCH_CHECK("conv3x3_wgrad thin");
    float* bpart = t.slab + (size_t)THIN_BLOCKS * 4 * t.Cb * 32;
    if (!in_thin && dbias) {
      hipLaunchKernelGGL(thin_colsum_kernel, dim3(THIN_BIAS_BLOCKS), dim3(256), 0, st, dy, (int64_t)N * Hi * Wi, Cout, bpart);
      ODVAE_LAUNCH_CHECK("conv3x3_wgrad thin bias");
    }
    const int items = t.Cb * 32 + (in_thin ? 0 : Cout);
    hipLaunchKernelGGL(conv3x3_wgrad_thin_reduce_kernel, dim3(ceil_div(items, 16)), dim3(256), 0, st,
                       t.slab, blocks * 4, t.Cb, t.Cs, t.CbP, t.sign, bpart, THIN_BIAS_BLOCKS, Cin, Cout, dw, dbias);
    ODVAE_LAUNCH_CHECK("conv3x3_wgrad thin reduce");
    return ODVAE_OK;
  }
  WgradParams p;
  p.x = x; p.dy = dy;
  p.slab = static_cast<float*>(workspace);
  p.bslab = dbias ? p.slab + (size_t)pl.nsplit * (pl.up ? 16 : 9) * pl.CinP * pl.CoutP : nullptr;
  if (mode == 5 && !pl.up) mode = 2;
  p.N = N; p.Hi = Hi; p.Wi = Wi; p.Cin = Cin; p.Ho = Ho; p.Wo = Wo; p.Cout = Cout;
  p.CinP = pl.CinP; p.CoutP = pl.CoutP;
  p.tiles_x = pl.tiles_x; p.tiles_y = pl.tiles_y; p.ntiles = pl.ntiles;
  p.tiles_per_split = pl.tiles_per_split; p.ci_tiles = pl.ci_tiles;
  dim3 grid(pl.nsplit, pl.ci_tiles * pl.co_tiles);
  hipError_t e = hipSuccess;
  if (pl.up) {
    e = launch_dyn(conv3x3_wgrad_up_kernel, grid, dim3(512), (size_t)2 * UpGeom::BUF_F * sizeof(float), st, p);
    if (e != hipSuccess) {
      odvae_set_error("conv3x3_wgrad: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return ODVAE_ERR_HIP;
    }
    ODVAE_LAUNCH_CHECK("conv3x3_wgrad up");
    const int64_t total = (int64_t)9 * pl.CinP * pl.CoutP + pl.CoutP;
    hipLaunchKernelGGL(conv3x3_wgrad_up_reduce_kernel, dim3((int)std::min<int64_t>(ceil_div64(total, 256), 4096)), dim3(256), 0, st,
                       p.slab, p.bslab, pl.nsplit, Cin, Cout, pl.CinP, pl.CoutP, dw, dbias);
    ODVAE_LAUNCH_CHECK("conv3x3_wgrad up reduce");
    return ODVAE_OK;
  }
  if (pl.v2) {
    if (mode == 0)      e = launch_dyn(conv3x3_wgrad_dma_kernel<0, 4>, grid, dim3(512), (size_t)2 * DmaGeom<0, 4>::BUF_F * sizeof(float), st, p);
    else if (mode == 1) e = launch_dyn(conv3x3_wgrad_dma_kernel<1, 2>, grid, dim3(512), (size_t)2 * DmaGeom<1, 2>::BUF_F * sizeof(float), st, p);
    else                e = launch_dyn(conv3x3_wgrad_dma_kernel<2, 4>, grid, dim3(512), (size_t)2 * DmaGeom<2, 4>::BUF_F * sizeof(float), st, p);
  } else {
    if (mode == 0)      e = launch_dyn(conv3x3_wgrad_kernel<0, 8>, grid, dim3(256), wgrad_smem_bytes<0, 8>(), st, p);
    else if (mode == 1) e = launch_dyn(conv3x3_wgrad_kernel<1, 4>, grid, dim3(256), wgrad_smem_bytes<1, 4>(), st, p);
    else                e = launch_dyn(conv3x3_wgrad_kernel<2, 8>, grid, dim3(256), wgrad_smem_bytes<2, 8>(), st, p);
  }
  if (e != hipSuccess) {
    odvae_set_error("conv3x3_wgrad: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
    return ODVAE_ERR_HIP;
  }
  ODVAE_LAUNCH_CHECK("conv3x3_wgrad");
  const int64_t total = (int64_t)9 * pl.CinP * pl.CoutP + pl.CoutP;
  const int blocks = (int)std::min<int64_t>(ceil_div64(total, 256), 4096);
  hipLaunchKernelGGL(conv3x3_wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, st,
                     p.slab, p.bslab, pl.nsplit, Cin, Cout, pl.CinP, pl.CoutP, dw, dbias);
  ODVAE_LAUNCH_CHECK("conv3x3_wgrad reduce");
  return ODVAE_OK;
}

}  // extern "C"
