// Stride-1 3x3 convolution by Winograd F(2x2, 3x3), NHWC f32, fused transforms, gfx950.
//
// Same contract as mode 0 of conv3x3_f32.hip (ResnetBlock / conv_in / conv_out convs of [UPSTREAM]
// ldm/modules/diffusionmodules/model.py and their data gradients), 2.25x fewer multiply-adds:
//   Y = A^T [ sum_ci (G g G^T) (.) (B^T d B) ] A      per 2x2 output tile, d = its 4x4 input patch
//   B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1],  G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1],  A^T = [1 1 1 0; 0 1 -1 -1]
// i.e. sixteen independent [tiles x Cin] x [Cin x Cout] products M[xi] = V[xi] U[xi] (xi = position in the 4x4 tile)
// on the f32 MFMA, with U = G g G^T packed once per weight update (odvae_conv3x3_pack_wino_f32).
//
// Block = 4 waves, 8x16 output pixels = 32 tiles (4 x 8) x 64 output channels.  Per 16-channel chunk: the input halo
// (10 x 18 px) goes global -> registers -> LDS, every thread transforms its share into V [16 xi][32 tiles][16 ch] in
// LDS, then wave w multiplies row w of the 4x4 (xi = 4w .. 4w+3: A fragments from V, B fragments straight from the
// L2-resident U pack) into 4 xi x 2 co-tiles of accumulators.  The output transform needs all four rows of a tile:
// each wave folds its row (M A, registers only), the rows meet through LDS (A^T), and bias / residual / store follow.
// The arithmetic is f32 throughout; only the summation order differs from the direct form (error ~1e-6 relative).
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int TH = 8, TW = 16;           // output pixels per block
constexpr int NT = (TH / 2) * (TW / 2);  // 32 tiles of 2x2
constexpr int TXN = TW / 2;              // tiles per row of the block
constexpr int HH = TH + 2, HW = TW + 2;  // input halo
constexpr int KC = 16;                   // channels per chunk
constexpr int HS = KC + 4;               // halo / V row stride in floats (b128 reads land on distinct bank groups)
constexpr int BN = 64;                   // output channels per block
constexpr int HALO_F = HH * HW * HS;     // 3600 floats
constexpr int V_F = 16 * NT * HS;        // 10240 floats

struct WinoParams {
  const float* x;         // [N][H][W][Cin]
  const float* upk;       // [16][CinP/4][CoutP][4]
  const float* bias;      // [Cout] or null
  const float* residual;  // [N][H][W][Cout] or null
  float* y;               // [N][H][W][Cout]
  int N, H, W, Cin, Cout, CinP, CoutP, tiles_x, tiles_y, act;
  int xcd;                // 1: XCD-contiguous tile order (neighbouring tiles share an L2)
};

__global__ __launch_bounds__(256, 2) void conv3x3_wino_kernel(WinoParams p) {
  __shared__ __attribute__((aligned(16))) float smem[2 * HALO_F + V_F];   // 69 760 B: two blocks per CU
  float* Hs = smem;                  // two halo stages
  float* Vs = smem + 2 * HALO_F;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, h = lane >> 5;
  int t = blockIdx.x;
  const int tx = t % p.tiles_x; t /= p.tiles_x;
  const int ty = t % p.tiles_y; const int n = t / p.tiles_y;
  const int oy0 = ty * TH, ox0 = tx * TW;
  const int n0 = blockIdx.y * BN;
  const float* xn = p.x + (int64_t)n * p.H * p.W * p.Cin;
  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xn), 0, p.H * p.W * p.Cin * 4, 0x00020000);

  // ---- halo: 180 px x 4 channel quads = 720 float4 per chunk, 3 per thread (branch-free buffer loads) ----
  constexpr int HALO_F4 = HH * HW * (KC / 4);
  constexpr int HALO_IT = (HALO_F4 + 255) / 256;
  float4 hreg[HALO_IT];
  auto load_halo = [&](int c0) {
#pragma unroll
    for (int i = 0; i < HALO_IT; ++i) {
      const int f = tid + 256 * i;
      const int hp = f >> 2, q = f & 3;
      const int iy = oy0 - 1 + hp / HW, ix = ox0 - 1 + hp % HW;
      const int c = c0 + 4 * q;
      const bool ok = f < HALO_F4 && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W && c < p.Cin;
      const unsigned voff = ok ? (unsigned)(((iy * p.W + ix) * p.Cin + c) * 4) : 0x7FFFFFF0u;
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, voff, 0, 0);
      hreg[i] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    }
  };
  auto store_halo = [&](float* Hd) {
#pragma unroll
    for (int i = 0; i < HALO_IT; ++i) {
      const int f = tid + 256 * i;
      if (f < HALO_F4) *reinterpret_cast<float4*>(Hd + (f >> 2) * HS + 4 * (f & 3)) = hreg[i];
    }
  };
  // ---- input transform: thread = (tile, channel quad, row pair); V rows {0,1} need d rows 0..2, rows {2,3} rows 1..3 ----
  auto f4add = [](float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); };
  auto f4sub = [](float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); };
  auto transform = [&](const float* Hc) {
    const int tile = tid >> 3, q = (tid >> 1) & 3, rp = tid & 1;     // 32 tiles x 4 quads x 2 row pairs
    const int ty2 = 2 * (tile / TXN), tx2 = 2 * (tile % TXN);
    // rows of B^T d, one column of the patch at a time (keeps 3 float4 live instead of 12):
    //   rp 0: r0 = d0 - d2, r1 = d1 + d2 ;  rp 1 (rows 1,2,3 loaded): r2 = d2 - d1, r3 = d1 - d3
    float4 ra[4], rb[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float* src = Hc + ((ty2 + rp) * HW + tx2 + c) * HS + 4 * q;
      const float4 d0 = *reinterpret_cast<const float4*>(src);
      const float4 d1 = *reinterpret_cast<const float4*>(src + HW * HS);
      const float4 d2 = *reinterpret_cast<const float4*>(src + 2 * HW * HS);
      if (rp == 0) { ra[c] = f4sub(d0, d2); rb[c] = f4add(d1, d2); }
      else { ra[c] = f4sub(d1, d0); rb[c] = f4sub(d0, d2); }
    }
    // times B: columns  c0 = r0 - r2, c1 = r1 + r2, c2 = r2 - r1, c3 = r1 - r3
    const float4 va[4] = {f4sub(ra[0], ra[2]), f4add(ra[1], ra[2]), f4sub(ra[2], ra[1]), f4sub(ra[1], ra[3])};
    const float4 vb[4] = {f4sub(rb[0], rb[2]), f4add(rb[1], rb[2]), f4sub(rb[2], rb[1]), f4sub(rb[1], rb[3])};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      *reinterpret_cast<float4*>(Vs + (((2 * rp + 0) * 4 + c) * NT + tile) * HS + 4 * q) = va[c];
      *reinterpret_cast<float4*>(Vs + (((2 * rp + 1) * 4 + c) * NT + tile) * HS + 4 * q) = vb[c];
    }
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][nt][r] = 0.f;

  const float4* uq = reinterpret_cast<const float4*>(p.upk);
  const int QT = p.CinP / 4;
  const int nchunks = p.CinP / KC;
  // B fragments of step (chunk, group): xi = 4 wave + j, two co tiles
  auto load_b = [&](int ch, int g, float4 (&b)[4][2]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t base = ((int64_t)(wave * 4 + j) * QT + ch * (KC / 4) + 2 * g + h) * p.CoutP + n0 + li;
      b[j][0] = uq[base];
      b[j][1] = uq[base + 32];
    }
  };
  auto load_a = [&](int g, float4 (&a)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) a[j] = *reinterpret_cast<const float4*>(Vs + (((wave * 4 + j) * NT) + li) * HS + 8 * g + 4 * h);
  };
  auto mma = [&](const float4 (&a)[4], const float4 (&b)[4][2]) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        acc[j][nt] = mfma32(a[j].x, b[j][nt].x, acc[j][nt]);
        acc[j][nt] = mfma32(a[j].y, b[j][nt].y, acc[j][nt]);
        acc[j][nt] = mfma32(a[j].z, b[j][nt].z, acc[j][nt]);
        acc[j][nt] = mfma32(a[j].w, b[j][nt].w, acc[j][nt]);
      }
  };

  // Two barriers per chunk: the halo is double-buffered in LDS (chunk c+1 is stored while chunk c is multiplied), V is
  // single-buffered (transform -> barrier -> MFMAs -> barrier).  The two blocks of a CU run these phases out of step, so
  // one block's transform (VALU + LDS) hides under the other's MFMAs.
  static_assert(KC == 16, "two k-groups per chunk");
  float4 b[4][2], a[4];   // one set of weight fragments, reloaded right behind the MFMAs that read it (see the 8-wave kernel)
  load_halo(0);
  load_b(0, 0, b);
  store_halo(Hs);
  if (nchunks > 1) load_halo(KC);
  __syncthreads();
  for (int ch = 0; ch < nchunks; ++ch) {
    transform(Hs + (ch & 1) * HALO_F);
    __syncthreads();
    if (ch + 1 < nchunks) {
      store_halo(Hs + ((ch + 1) & 1) * HALO_F);
      if (ch + 2 < nchunks) load_halo((ch + 2) * KC);
    }
    load_a(0, a);
    mma(a, b);
    load_b(ch, 1, b);
    load_a(1, a);
    mma(a, b);
    if (ch + 1 < nchunks) load_b(ch + 1, 0, b);
    __syncthreads();
  }

  // ---- output transform.  Row fold in registers: R0 = m0 + m1 + m2, R1 = m1 - m2 - m3 (over j); then the four rows
  // (waves) meet in LDS: Y[0][c] = R(0) + R(1) + R(2), Y[1][c] = R(1) - R(2) - R(3) ----
  // V and the halo stages are dead (the loop ends on a barrier): reuse smem as X[wave][c][reg][lane] = 32 KB, one co
  // tile at a time
  float* X = smem;
  const int img_bytes = p.H * p.W * p.Cout * 4;
  const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(p.y + (int64_t)n * p.H * p.W * p.Cout, 0, img_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.residual ? p.residual : p.y) + (int64_t)n * p.H * p.W * p.Cout, 0, p.residual ? img_bytes : 0, 0x00020000);
  const bool relu = p.act != 0;
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float m0 = acc[0][nt][r], m1 = acc[1][nt][r], m2 = acc[2][nt][r], m3 = acc[3][nt][r];
      X[((wave * 2 + 0) * 16 + r) * 64 + lane] = m0 + m1 + m2;
      X[((wave * 2 + 1) * 16 + r) * 64 + lane] = m1 - m2 - m3;
    }
    // wave w finishes registers r = 4w .. 4w+3 (tiles (r&3) + 8(r>>2) + 4h): 4 output pixels each.  Bias + residual are
    // fetched before the exchange barrier; the stores follow with no load in between.
    const int co = n0 + nt * 32 + li;
    const float bv = (p.bias && co < p.Cout) ? p.bias[co] : 0.f;
    unsigned off[4][2][2];
    float seed[4][2][2];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int r = wave * 4 + rr;
      const int tile = (r & 3) + 8 * (r >> 2) + 4 * h;
      const int py = oy0 + 2 * (tile / TXN), px = ox0 + 2 * (tile % TXN);
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int a2 = 0; a2 < 2; ++a2) {
          const int oy = py + a2, ox = px + c;
          off[rr][c][a2] = (oy < p.H && ox < p.W && co < p.Cout) ? (unsigned)(((oy * p.W + ox) * p.Cout + co) * 4) : 0x7FFFFFF0u;
          seed[rr][c][a2] = bv + __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rrsrc, off[rr][c][a2], 0, 0));
        }
    }
    __syncthreads();
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int r = wave * 4 + rr;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const float r0 = X[((0 * 2 + c) * 16 + r) * 64 + lane], r1 = X[((1 * 2 + c) * 16 + r) * 64 + lane];
        const float r2 = X[((2 * 2 + c) * 16 + r) * 64 + lane], r3 = X[((3 * 2 + c) * 16 + r) * 64 + lane];
        const float y0 = r0 + r1 + r2 + seed[rr][c][0], y1 = r1 - r2 - r3 + seed[rr][c][1];
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(relu ? fmaxf(y0, 0.f) : y0), yrsrc, off[rr][c][0], 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(relu ? fmaxf(y1, 0.f) : y1), yrsrc, off[rr][c][1], 0, 0);
      }
    }
    __syncthreads();   // X is rewritten for the next co-tile
  }
}

// ---- 8-wave variant: 32 tiles x 128 output channels per block, one block per CU ----------------------------------
// With 64 output channels a block holds only 512 MFMAs per wave at Cin = 128, so its prologue (first halo fetch + first
// transform), the transform phases and the output exchange weigh 40 % (measured: 56 % MFMA utilisation).  Here eight
// waves (4 rows of the 4x4 x 2 co halves) share one V, V is double-buffered as well, and the transform of chunk c+1 sits
// between the two MFMA groups of chunk c: one barrier per chunk, the transform work per thread halves, and the input
// transform is done once per 128 output channels instead of once per 64.  111 KB of LDS: one block per CU.
constexpr int BN8 = 128;

// PERSIST: one block per CU walks a sequence of tiles and the chunk pipeline runs on across the tile boundary -- the halo
// fetches, halo stores and the first transform of the next tile take the slots the last three chunks of the current tile
// leave empty, and its first weight fragments are requested behind the last MFMAs.  The next tile then starts with its first
// MFMA: no prologue (fetch -> LDS -> transform -> LDS, 5.5k of a block's 95k cycles with nothing to overlap) except for a
// block's first tile.  Needs an even chunk count >= 4 (the two-stage rotation of halo and V then continues seamlessly) and
// the output exchange in LDS that the next tile's staged data does not use (X behind the second V stage: 135 KB in all).
constexpr int X_OFF_PERSIST = 2 * HALO_F + V_F;

template <bool PERSIST>
__global__ __launch_bounds__(512, 2) void conv3x3_wino8_kernel(WinoParams p) {
  extern __shared__ __attribute__((aligned(16))) float dsmem[];   // 2 halo stages + 2 V stages (+ the exchange area when PERSIST)
  float* Hs = dsmem;
  float* Vs = dsmem + 2 * HALO_F;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int xr = wave & 3, half = wave >> 2;          // row of the 4x4, co half
  const int li = lane & 31, h = lane >> 5;
  // Tile sequence of this block.  Not PERSIST: the one tile blockIdx.x names.  PERSIST (1-D grid of G blocks, G % (8 ny) == 0):
  // block b works on co block (b >> 3) % ny and on tiles i_b, i_b + TL, i_b + 2 TL, ... of the XCD-contiguous order, where
  // TL = G / ny blocks share a co block and i_b % 8 == b % 8, the XCD the block runs on.
  const int total_tiles = p.tiles_x * p.tiles_y * p.N;
  const int ny = p.CoutP / BN8;
  const int TL = PERSIST ? (int)gridDim.x / ny : 0;
  const int yblk = PERSIST ? ((int)blockIdx.x >> 3) % ny : (int)blockIdx.y;
  int s_cur = PERSIST ? ((int)blockIdx.x & 7) + 8 * (((int)blockIdx.x >> 3) / ny) : (int)blockIdx.x;
  struct Tile { int oy0, ox0, n; __amdgpu_buffer_rsrc_t xrsrc; };
  auto decode = [&](int sidx) {
    int t = (PERSIST || p.xcd) ? xcd_contiguous(sidx, PERSIST ? total_tiles : (int)gridDim.x) : sidx;
    Tile T;
    const int tx = t % p.tiles_x; t /= p.tiles_x;
    const int ty = t % p.tiles_y; T.n = t / p.tiles_y;
    T.oy0 = ty * TH; T.ox0 = tx * TW;
    T.xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x + (int64_t)T.n * p.H * p.W * p.Cin), 0, p.H * p.W * p.Cin * 4, 0x00020000);
    return T;
  };
  Tile cur = decode(s_cur);
  const int n0 = yblk * BN8 + half * 64;

  constexpr int HALO_F4 = HH * HW * (KC / 4);           // 720
  constexpr int HALO_IT = (HALO_F4 + 511) / 512;        // 2
  float4 hreg[HALO_IT];
  auto load_halo_to = [&](const Tile& T, int c0, float4 (&hr)[HALO_IT]) {
#pragma unroll
    for (int i = 0; i < HALO_IT; ++i) {
      const int f = tid + 512 * i;
      const int hp = f >> 2, q = f & 3;
      const int iy = T.oy0 - 1 + hp / HW, ix = T.ox0 - 1 + hp % HW;
      const int c = c0 + 4 * q;
      const bool ok = f < HALO_F4 && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W && c < p.Cin;
      const unsigned voff = ok ? (unsigned)(((iy * p.W + ix) * p.Cin + c) * 4) : 0x7FFFFFF0u;
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(T.xrsrc, voff, 0, 0);
      hr[i] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    }
  };
  auto store_halo_from = [&](float* Hd, const float4 (&hr)[HALO_IT]) {
#pragma unroll
    for (int i = 0; i < HALO_IT; ++i) {
      const int f = tid + 512 * i;
      if (f < HALO_F4) *reinterpret_cast<float4*>(Hd + (f >> 2) * HS + 4 * (f & 3)) = hr[i];
    }
  };
  auto load_halo = [&](const Tile& T, int c0) { load_halo_to(T, c0, hreg); };
  auto store_halo = [&](float* Hd) { store_halo_from(Hd, hreg); };
  auto f4add = [](float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); };
  auto f4sub = [](float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); };
  // thread = (tile, channel quad, row pair rp, column pair cp): rows {2rp, 2rp+1} x columns {2cp, 2cp+1} of V.  (rp, cp) is
  // wave-uniform (wave & 3): the four variants of the transform are scalar branches, so a wave executes only its own variant
  // (with per-lane roles both sides of every branch ran: twice the vector instructions).
  auto transform = [&](const float* Hc, float* Vd) {
    const int idx = ((wave >> 2) << 6) | lane;
    const int tile = idx >> 2, q = idx & 3, rp = (wave >> 1) & 1, cp = wave & 1;
    const int ty2 = 2 * (tile / TXN), tx2 = 2 * (tile % TXN);
    float4 ra[3], rb[3];   // rows of B^T d at patch columns cp, cp+1, cp+2
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float* src = Hc + ((ty2 + rp) * HW + tx2 + cp + k) * HS + 4 * q;
      const float4 d0 = *reinterpret_cast<const float4*>(src);
      const float4 d1 = *reinterpret_cast<const float4*>(src + HW * HS);
      const float4 d2 = *reinterpret_cast<const float4*>(src + 2 * HW * HS);
      if (rp == 0) { ra[k] = f4sub(d0, d2); rb[k] = f4add(d1, d2); }      // r0 = d0 - d2, r1 = d1 + d2
      else { ra[k] = f4sub(d1, d0); rb[k] = f4sub(d0, d2); }             // r2 = d2 - d1, r3 = d1 - d3 (rows 1,2,3 loaded)
    }
    // cp 0: c0 = col0 - col2, c1 = col1 + col2;  cp 1 (cols 1,2,3 loaded): c2 = col2 - col1, c3 = col1 - col3
    float4 va0, va1, vb0, vb1;
    if (cp == 0) { va0 = f4sub(ra[0], ra[2]); va1 = f4add(ra[1], ra[2]); vb0 = f4sub(rb[0], rb[2]); vb1 = f4add(rb[1], rb[2]); }
    else { va0 = f4sub(ra[1], ra[0]); va1 = f4sub(ra[0], ra[2]); vb0 = f4sub(rb[1], rb[0]); vb1 = f4sub(rb[0], rb[2]); }
    float* dst = Vd + tile * HS + 4 * q;
    *reinterpret_cast<float4*>(dst + ((2 * rp + 0) * 4 + 2 * cp + 0) * NT * HS) = va0;
    *reinterpret_cast<float4*>(dst + ((2 * rp + 0) * 4 + 2 * cp + 1) * NT * HS) = va1;
    *reinterpret_cast<float4*>(dst + ((2 * rp + 1) * 4 + 2 * cp + 0) * NT * HS) = vb0;
    *reinterpret_cast<float4*>(dst + ((2 * rp + 1) * 4 + 2 * cp + 1) * NT * HS) = vb1;
  };

  f32x16 acc[4][2];

  // Weight fragments come through a buffer descriptor: the lane part of the address (k half h, output channel) is one VGPR
  // computed here, the (xi, chunk, group) part is a scalar offset and the second co-tile an immediate, so a fetch costs no
  // vector instruction (64-bit per-lane address arithmetic was 13 % of the loop's issue cycles).
  const int QT = p.CinP / 4;
  const int nchunks = p.CinP / KC;
  const __amdgpu_buffer_rsrc_t ursrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.upk), 0, 16 * p.CinP * p.CoutP * 4, 0x00020000);
  const unsigned b_voff = (unsigned)((h * p.CoutP + n0 + li) * 16);
  const unsigned b_row = (unsigned)p.CoutP * 16u;      // bytes per k quad
  auto load_b = [&](int ch, int g, float4 (&b)[4][2]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane(((xr * 4 + j) * QT + ch * (KC / 4) + 2 * g) * (int)b_row);
      const u32x4 v0 = __builtin_amdgcn_raw_buffer_load_b128(ursrc, b_voff, soff, 0);
      const u32x4 v1 = __builtin_amdgcn_raw_buffer_load_b128(ursrc, b_voff + 512u, soff, 0);
      b[j][0] = make_float4(__uint_as_float(v0.x), __uint_as_float(v0.y), __uint_as_float(v0.z), __uint_as_float(v0.w));
      b[j][1] = make_float4(__uint_as_float(v1.x), __uint_as_float(v1.y), __uint_as_float(v1.z), __uint_as_float(v1.w));
    }
  };
  auto load_a = [&](const float* Vc, int g, float4 (&a)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) a[j] = *reinterpret_cast<const float4*>(Vc + (((xr * 4 + j) * NT) + li) * HS + 8 * g + 4 * h);
  };
  auto mma = [&](const float4 (&a)[4], const float4 (&b)[4][2]) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        acc[j][nt] = mfma32(a[j].x, b[j][nt].x, acc[j][nt]);
        acc[j][nt] = mfma32(a[j].y, b[j][nt].y, acc[j][nt]);
        acc[j][nt] = mfma32(a[j].z, b[j][nt].z, acc[j][nt]);
        acc[j][nt] = mfma32(a[j].w, b[j][nt].w, acc[j][nt]);
      }
  };

  // What the loop is built around (tools/mfma_mix_probe.hip, profiles/r02_wino8_loop.md): on gfx950 the f32 MFMA does not
  // overlap other vector work -- every VALU instruction, and the register write-back of every LDS / memory load, takes its
  // cycles away from the matrix pipe (SQ_VALU_MFMA_COEXEC_CYCLES = 0 for this kernel).  Latency is not the limiter (pinning
  // the weight refills 24 MFMAs ahead, staggering the SIMD partners by half a chunk, dropping the barrier: no change each),
  // instruction count is.  Hence: weight fetches with scalar offsets (no per-lane address arithmetic), wave-uniform
  // transform roles, one set of weight registers refilled in place.
  // vmcnt retires in order, so the halo fetch (HBM latency) is always the YOUNGEST load in flight when weights are awaited.
  float4 b[4][2], a[4];
  {   // prologue: the halos of chunks 0 and 1 are requested together (one HBM latency, not two in a row)
    float4 h1[HALO_IT];
    load_halo(cur, 0);
    if (nchunks > 1) load_halo_to(cur, KC, h1);
    load_b(0, 0, b);
    store_halo(Hs);
    if (nchunks > 1) store_halo_from(Hs + HALO_F, h1);
    if (nchunks > 2) load_halo(cur, 2 * KC);
  }
  __syncthreads();
  transform(Hs, Vs);
  __syncthreads();
  // pins the order {8 MFMAs of xi row j, the two weight fetches that refill b[j] for the next group} x 4: every fragment is
  // re-requested as soon as its last MFMA has issued, 24 MFMAs before it is needed again
  auto pin_reload = [&]() {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
    }
  };
  for (;;) {      // tiles of this block (exactly one unless PERSIST)
  const int s_nxt = s_cur + TL;
  const bool has_next = PERSIST && s_nxt < total_tiles;
  const Tile nxt = decode(has_next ? s_nxt : s_cur);
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][nt][r] = 0.f;
  for (int ch = 0; ch < nchunks; ++ch) {
    const float* Vc = Vs + (ch & 1) * V_F;
    load_a(Vc, 0, a);
    mma(a, b);
    load_b(ch, 1, b);
    pin_reload();
    // chunk indices past the end of this tile are the first chunks of the next one (PERSIST: nchunks is even, the stage
    // rotation carries over)
    if (ch + 1 < nchunks || has_next) transform(Hs + ((ch + 1) & 1) * HALO_F, Vs + ((ch + 1) & 1) * V_F);
    load_a(Vc, 1, a);
    mma(a, b);
    load_b(ch + 1 < nchunks ? ch + 1 : 0, 0, b);     // past the last chunk: the next tile's first fragments (same co block)
    pin_reload();
    if (ch + 2 < nchunks || has_next) store_halo(Hs + (ch & 1) * HALO_F);   // chunk ch+2; this stage was last read by transform(ch)
    if (ch + 3 < nchunks) load_halo(cur, (ch + 3) * KC);
    else if (has_next) load_halo(nxt, (ch + 3 - nchunks) * KC);
    __syncthreads();
  }

  // output transform: as in the 4-wave kernel, one exchange area per co half: X[half][row][c][reg][lane].  The residual
  // values of this wave's 16 output elements are requested BEFORE the exchange barrier (their latency hides under it) and
  // the stores then follow with no load in between (vmcnt retires loads and stores in order).
  float* X = dsmem + (PERSIST ? X_OFF_PERSIST : 0) + half * 8192;
  const int img_bytes = p.H * p.W * p.Cout * 4;
  const int n = cur.n, oy0 = cur.oy0, ox0 = cur.ox0;
  const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(p.y + (int64_t)n * p.H * p.W * p.Cout, 0, img_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.residual ? p.residual : p.y) + (int64_t)n * p.H * p.W * p.Cout, 0, p.residual ? img_bytes : 0, 0x00020000);
  const bool relu = p.act != 0;
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float m0 = acc[0][nt][r], m1 = acc[1][nt][r], m2 = acc[2][nt][r], m3 = acc[3][nt][r];
      X[((xr * 2 + 0) * 16 + r) * 64 + lane] = m0 + m1 + m2;
      X[((xr * 2 + 1) * 16 + r) * 64 + lane] = m1 - m2 - m3;
    }
    const int co = n0 + nt * 32 + li;
    const float bv = (p.bias && co < p.Cout) ? p.bias[co] : 0.f;
    unsigned off[4][2][2];
    float seed[4][2][2];      // bias + residual (a descriptor of 0 bytes returns 0 when there is no residual)
    // register r = 4 xr + rr of this wave is tile (rr + 4h) of tile row xr: output pixel (oy0 + 2 xr, ox0 + 8h + 2 rr) and its
    // three neighbours.  One multiply for the wave's first pixel, the other 15 offsets are adds of scalar steps; H and W are
    // even, so a 2x2 output tile is inside the image or outside as a whole (the launcher keeps OOB - steps >= num_records).
    const unsigned cstep = (unsigned)p.Cout * 4u, rstep = (unsigned)(p.W * p.Cout) * 4u;
    const unsigned oob = 0x7FFFFFF0u - rstep - cstep;
    const int py = oy0 + 2 * xr, px0 = ox0 + 8 * h;
    const unsigned base0 = (unsigned)(((py * p.W + px0) * p.Cout + co) * 4);
    const bool rowok = py < p.H && co < p.Cout;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const unsigned b0 = (rowok && px0 + 2 * rr < p.W) ? base0 + 2u * rr * cstep : oob;
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int a2 = 0; a2 < 2; ++a2) {
          off[rr][c][a2] = b0 + c * cstep + a2 * rstep;
          seed[rr][c][a2] = bv + __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rrsrc, off[rr][c][a2], 0, 0));
        }
    }
    __syncthreads();
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int r = xr * 4 + rr;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const float r0 = X[((0 * 2 + c) * 16 + r) * 64 + lane], r1 = X[((1 * 2 + c) * 16 + r) * 64 + lane];
        const float r2 = X[((2 * 2 + c) * 16 + r) * 64 + lane], r3 = X[((3 * 2 + c) * 16 + r) * 64 + lane];
        const float y0 = r0 + r1 + r2 + seed[rr][c][0], y1 = r1 - r2 - r3 + seed[rr][c][1];
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(relu ? fmaxf(y0, 0.f) : y0), yrsrc, off[rr][c][0], 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(relu ? fmaxf(y1, 0.f) : y1), yrsrc, off[rr][c][1], 0, 0);
      }
    }
    __syncthreads();
  }
  if (!has_next) break;
  cur = nxt;
  s_cur = s_nxt;
  }      // tiles
}

// U[xi = 4a + b][ci][co] = (G g G^T)[a][b];  dgrad: g taken with flipped taps and swapped channel roles.
// One thread per (co, ci) pair: reads the nine weights once and writes its 16 (+16) pack entries; padding entries of the
// packs are zero-filled by the launcher (hipMemsetAsync) beforehand.
// items != null: batched launch, blockIdx.y selects the weight (unpadded channel counts: the pads are the counts themselves)
__global__ void conv3x3_pack_wino_kernel(const float* __restrict__ w, int Cout, int Cin,
                                         float* __restrict__ fwd, int CinP_f, int CoutP_f,
                                         float* __restrict__ dgr, int CoutP_d, int CinP_d, const OdvaePackItem* __restrict__ items) {
  if (items) {
    const OdvaePackItem it = items[blockIdx.y];
    w = it.w; Cout = it.Cout; Cin = it.Cin; fwd = static_cast<float*>(it.fwd); dgr = static_cast<float*>(it.dgr);
    CinP_f = Cin; CoutP_f = Cout; CoutP_d = Cout; CinP_d = Cin;
  }
  const int64_t pairs = (int64_t)Cout * Cin;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < pairs; idx += (int64_t)gridDim.x * blockDim.x) {
    // co fastest: the forward pack is co-contiguous, so a wavefront writes 64 consecutive float4 slots' lanes
    const int co = (int)(idx % Cout), ci = (int)(idx / Cout);
    const float* g = w + ((int64_t)co * Cin + ci) * 9;
    float gg[3][3];
#pragma unroll
    for (int k = 0; k < 9; ++k) gg[k / 3][k % 3] = g[k];
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      float* dst = pass == 0 ? fwd : dgr;
      if (!dst) continue;
      // G x: rows (x0, (x0+x1+x2)/2, (x0-x1+x2)/2, x2)
      float t[4][3], u[4][4];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float x0 = pass == 0 ? gg[0][c] : gg[2][2 - c], x1 = pass == 0 ? gg[1][c] : gg[1][2 - c],
                    x2 = pass == 0 ? gg[2][c] : gg[0][2 - c];
        t[0][c] = x0; t[1][c] = 0.5f * (x0 + x1 + x2); t[2][c] = 0.5f * (x0 - x1 + x2); t[3][c] = x2;
      }
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        u[a][0] = t[a][0]; u[a][1] = 0.5f * (t[a][0] + t[a][1] + t[a][2]); u[a][2] = 0.5f * (t[a][0] - t[a][1] + t[a][2]);
        u[a][3] = t[a][2];
      }
      const int red = pass == 0 ? ci : co, out = pass == 0 ? co : ci;
      const int redP = pass == 0 ? CinP_f : CoutP_d, outP = pass == 0 ? CoutP_f : CinP_d;
#pragma unroll
      for (int xi = 0; xi < 16; ++xi)
        dst[(((int64_t)xi * (redP / 4) + red / 4) * outP + out) * 4 + (red & 3)] = u[xi >> 2][xi & 3];
    }
  }
}

constexpr int round_up_i(int a, int b) { return (a + b - 1) / b * b; }

}  // namespace

extern "C" {

// pack sizes: reduction axis padded to 16, output axis to 64
int odvae_conv3x3_wino_reduce_pad(int c_reduce) { return round_up_i(c_reduce, KC); }
int odvae_conv3x3_wino_out_pad(int c_out) { return round_up_i(c_out, BN); }
size_t odvae_conv3x3_wino_pack_floats(int c_reduce, int c_out) {
  return (size_t)16 * odvae_conv3x3_wino_reduce_pad(c_reduce) * odvae_conv3x3_wino_out_pad(c_out);
}

int odvae_conv3x3_pack_wino_f32(const float* w, int Cout, int Cin, float* fwd_pack, float* dgrad_pack, void* stream) {
  ODVAE_CHECK_ARG(w && Cout > 0 && Cin > 0, "conv3x3_pack_wino: bad arguments");
  const int CinP_f = odvae_conv3x3_wino_reduce_pad(Cin), CoutP_f = odvae_conv3x3_wino_out_pad(Cout);
  const int CoutP_d = odvae_conv3x3_wino_reduce_pad(Cout), CinP_d = odvae_conv3x3_wino_out_pad(Cin);
  const int64_t total = (fwd_pack ? (int64_t)16 * CinP_f * CoutP_f : 0) + (dgrad_pack ? (int64_t)16 * CoutP_d * CinP_d : 0);
  if (total == 0) return ODVAE_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  // padded rows / columns must read as zero; only shapes with padding need the fill
  if (fwd_pack && (CinP_f != Cin || CoutP_f != Cout)) {
    if (hipMemsetAsync(fwd_pack, 0, (size_t)16 * CinP_f * CoutP_f * sizeof(float), st) != hipSuccess) { odvae_set_error("conv3x3_pack_wino: memset failed"); return ODVAE_ERR_HIP; }
  }
  if (dgrad_pack && (CoutP_d != Cout || CinP_d != Cin)) {
    if (hipMemsetAsync(dgrad_pack, 0, (size_t)16 * CoutP_d * CinP_d * sizeof(float), st) != hipSuccess) { odvae_set_error("conv3x3_pack_wino: memset failed"); return ODVAE_ERR_HIP; }
  }
  const int blocks = (int)std::min<int64_t>(ceil_div64((int64_t)Cout * Cin, 256), 2048);
  hipLaunchKernelGGL(conv3x3_pack_wino_kernel, dim3(blocks), dim3(256), 0, st,
                     w, Cout, Cin, fwd_pack, CinP_f, CoutP_f, dgrad_pack, CoutP_d, CinP_d, (const OdvaePackItem*)nullptr);
  ODVAE_LAUNCH_CHECK("conv3x3_pack_wino");
  return ODVAE_OK;
}

// The same for n weights in ONE launch: items = device array of n OdvaePackItem (common.h: w, fwd pack, dgrad pack, Cout, Cin), every
// channel count its own pad (odvae_conv3x3_wino_reduce_pad(c) == odvae_conv3x3_wino_out_pad(c) == c).  A training step repacks ~20
// weights of this kind after every optimizer step: 20 launches of 30 us each become one.
int odvae_conv3x3_pack_wino_batch(const void* items, int n, void* stream) {
  ODVAE_CHECK_ARG(items && n > 0 && n <= 65535, "conv3x3_pack_wino_batch: bad arguments");
  hipLaunchKernelGGL(conv3x3_pack_wino_kernel, dim3(256, n), dim3(256), 0, static_cast<hipStream_t>(stream),
                     (const float*)nullptr, 0, 0, (float*)nullptr, 0, 0, (float*)nullptr, 0, 0, static_cast<const OdvaePackItem*>(items));
  ODVAE_LAUNCH_CHECK("conv3x3_pack_wino_batch");
  return ODVAE_OK;
}

// y = act(conv3x3_stride1_pad1(x) (+bias) (+residual)); upk = fwd or dgrad pack of odvae_conv3x3_pack_wino_f32.
// Needs even H and W and Cin % 4 == 0.
int odvae_conv3x3_wino_f32(const float* x, int N, int H, int W, int Cin, const float* upk, int Cout,
                           const float* bias, const float* residual, float* y, int act, void* stream) {
  ODVAE_CHECK_ARG(x && upk && y, "conv3x3_wino: null operand");
  ODVAE_CHECK_ARG(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "conv3x3_wino: empty shape");
  ODVAE_CHECK_ARG(H % 2 == 0 && W % 2 == 0 && Cin % 4 == 0, "conv3x3_wino: needs even H, W and Cin %% 4 == 0 (H=%d W=%d Cin=%d)", H, W, Cin);
  ODVAE_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)upk & 15) == 0, "conv3x3_wino: x/upk must be 16-byte aligned");
  ODVAE_CHECK_ARG((int64_t)H * W * Cin * 4 < 0x7FFFFFF0ll && ((int64_t)(H + 1) * W + 1) * Cout * 4 < 0x7FFFFFF0ll,
                  "conv3x3_wino: one input / output image must stay below 2 GiB");
  WinoParams p;
  p.xcd = 0;
  p.x = x; p.upk = upk; p.bias = bias; p.residual = residual; p.y = y;
  p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout;
  p.CinP = odvae_conv3x3_wino_reduce_pad(Cin); p.CoutP = odvae_conv3x3_wino_out_pad(Cout);
  p.tiles_x = ceil_div(W, TW); p.tiles_y = ceil_div(H, TH); p.act = act;
  const int64_t sp = (int64_t)p.tiles_x * p.tiles_y * N;
  ODVAE_CHECK_ARG(sp < (1ll << 31), "conv3x3_wino: too many tiles");
  static const bool no8 = getenv("ODVAE_WINO_4WAVE") != nullptr;
  if (Cout % BN8 == 0 && !no8) {
    // the persistent form needs >= 2 tiles per block, an even chunk count >= 4 and a grid that splits evenly over the co blocks
    static const bool no_persist = getenv("ODVAE_WINO_PERSIST") != nullptr && atoi(getenv("ODVAE_WINO_PERSIST")) == 0;
    static const int cus = [] {      // one persistent block per CU (256 on MI355X)
      int dev = 0, n = 0;
      if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
      return n;
    }();
    const int ny = p.CoutP / BN8, nchunks = p.CinP / KC, G = cus;
    const bool persist = !no_persist && nchunks >= 4 && (nchunks & 1) == 0 && G % (8 * ny) == 0 && sp >= 2 * (G / ny);
    const size_t smem = (size_t)(persist ? X_OFF_PERSIST + 16384 : 2 * HALO_F + 2 * V_F) * sizeof(float);
    auto kern = persist ? conv3x3_wino8_kernel<true> : conv3x3_wino8_kernel<false>;
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) {
      odvae_set_error("conv3x3_wino: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return ODVAE_ERR_HIP;
    }
    static const bool xcd = getenv("ODVAE_TILE_XCD") == nullptr || atoi(getenv("ODVAE_TILE_XCD")) != 0;
    p.xcd = xcd ? 1 : 0;
    if (persist) hipLaunchKernelGGL(kern, dim3(G), dim3(512), smem, static_cast<hipStream_t>(stream), p);
    else hipLaunchKernelGGL(kern, dim3((unsigned)sp, ny), dim3(512), smem, static_cast<hipStream_t>(stream), p);
  } else {
    hipLaunchKernelGGL(conv3x3_wino_kernel, dim3((unsigned)sp, p.CoutP / BN), dim3(256), 0, static_cast<hipStream_t>(stream), p);
  }
  ODVAE_LAUNCH_CHECK("conv3x3_wino");
  return ODVAE_OK;
}

}  // extern "C"
