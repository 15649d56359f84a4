// Stride-1 3x3 convolution by Winograd F(4x4, 3x3), NHWC f32, fused transforms, gfx950.
//
// Same contract as conv3x3_wino_f32.hip (the ResnetBlock convs of [UPSTREAM] ldm/modules/diffusionmodules/model.py and
// their data gradients) with 36 instead of 64 multiply-adds per 4x4 output pixels and (ci, co) -- 4x fewer than the
// direct form, 1.78x fewer than F(2x2, 3x3):
//   Y = A^T [ sum_ci (G g G^T) (.) (B^T d B) ] A        per 4x4 output tile, d = its 6x6 input patch
//   B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
//   G   = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1]
//   A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
// i.e. thirty-six independent [tiles x Cin] x [Cin x Cout] products M[xi] = V[xi] U[xi] on the f32 MFMA, U = G g G^T packed
// once per weight update.  The arithmetic is f32 throughout; the interpolation points 0, +-1, +-2 cost accuracy against
// F(2x2): measured max deviation from an f64 direct convolution 4e-6 of max|y| at Cin = 128 (direct f32: 2e-7).
//
// Block = 8 waves on 16 x 32 output pixels = 32 tiles (4 x 8) x 64 output channels: the 36 x 32 x 64 accumulators are 58 %
// of the CU's register file, which is what bounds the block (with 128 output channels they do not fit at all).  Wave
// (g, ct) owns xi = 9g .. 9g+8 for output channels 32 ct .. 32 ct + 31.  Per 8-channel chunk: the input halo (18 x 34 px) comes
// by LDS-DMA straight into LDS (no registers), in two planes (one per channel quad) whose rows are stored 4-way
// interleaved in x (x' = 9 (x & 3) + (x >> 2)) with a pitch of 38 slots: the 6x6 patches of eight neighbouring tiles then
// read as 128 contiguous bytes and those of the tile row below land on the other half of the banks -- every LDS access of
// the transform is conflict-free.  The input transform is split over the eight waves by ROWS of B^T d (rows {1,2}, {3,4}
// share their partial sums; 0 and 5 stand alone), one pair role and one single role per SIMD, two float channels per lane
// (packed f32 math); V is [xi][quad][tile][4], so an A fragment is 1 KB contiguous.  One barrier per chunk; V and the halo
// are double-buffered.  The output transform goes through LDS one tile row at a time (36 values per (tile, co) -> 16 pixels).
#include "bf16_common.h"
#include <stdlib.h>

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int TH = 16, TW = 32;            // output pixels per block
constexpr int TXN = TW / 4;                // 8 tiles per tile row, 4 tile rows
constexpr int KC = 8;                      // channels per chunk
constexpr int BN = 64;                     // output channels per block
constexpr int HROWS = TH + 2;              // 18 halo rows
constexpr int ROWP = 38;                   // 16-byte slots per halo row of one plane (36 used)
constexpr int PLANE = HROWS * ROWP;        // 684
constexpr int HALO_SLOTS = 2 * PLANE;      // 1368 slots = 21.4 KB
constexpr int HALO_DMA_PER_WAVE = 3;       // 24 LDS-DMA instructions of 1 KB per chunk (the last two carry no data)
constexpr unsigned HALO_B = 8 * HALO_DMA_PER_WAVE * 1024;   // 24 576
constexpr unsigned V_B = 36 * 1024;        // [36 xi][2 quads][32 tiles][4 floats]
constexpr unsigned LDS_B = 2 * HALO_B + 2 * V_B;            // 122 880
constexpr unsigned STATS_B = 4 * 512 * 8;                  // STATS build: (sum, sum of squares) per output-transform pass and thread, 16 KB behind the V stages
constexpr unsigned OOB = 0x7FFFFFF0u;

struct Wino4Params {
  const float* x;         // [N][H][W][Cin]
  const float* upk;       // [36][CinP/4][CoutP][4]
  const float* bias;      // [Cout] or null
  const float* residual;  // [N][H][W][Cout] or null
  float* y;               // [N][H][W][Cout]
  int N, H, W, Cin, Cout, CinP, CoutP, tiles_x, tiles_y, act, xcd;
  int persist;            // 1: 1-D grid of one block per CU, every block walks a sequence of tiles of one output-channel block
  int up;                 // 1: x is the LOW-resolution input [N][H/2][W/2][Cin] of an Upsample conv (nearest 2x + 3x3): the halo of the
                          // upsampled image is fetched from x[iy >> 1][ix >> 1]; H, W are the OUTPUT's
  float* gn_partial;      // [N][tiles per image][gn_groups][2] (sum, sum of squares of y per tile and channel group) or null
  int gn_groups, gn_cpg;  // channel groups of the GroupNorm that reads y, channels per group (a power of two <= 32)
  // EPI_GNBWD (this launch is the data gradient of a conv whose INPUT was a = swish(GroupNorm(gx)); y = da): the output transform also
  // leaves the first pass of that GroupNorm's backward -- per tile and channel the sums of du * xhat and du, du = da * swish'(u),
  // u = xhat * gamma + beta -- in gn_partial [N][tiles per image][2][Cout], the layout gn_bwd_reduce_kernel writes (groupnorm.hip)
  const float* gn_x;      // [N][H][W][Cout]: the GroupNorm's input
  const float* gn_mean;   // [N][gn_groups]
  const float* gn_rstd;   // [N][gn_groups]
  const float* gn_gamma;  // [Cout]
  const float* gn_beta;   // [Cout]
};
constexpr int EPI_NONE = 0, EPI_STATS = 1, EPI_GNBWD = 2, EPI_POOL = 3;
// EPI_POOL (the data gradient of an Upsample conv: y = the gradient w.r.t. the LOW-resolution input [N][H/2][W/2][Cout]): every thread
// of the output transform holds a 4x4 pixel block of one channel, so the 2x2 sums that nearest-2x upsampling's backward takes are formed
// in registers -- 4 stores instead of 16, and the full-resolution gradient never reaches HBM (upsample2x_bwd_kernel: 0.4 ms per step)

__device__ __forceinline__ f32x2 lds_ld64f(unsigned a) { return *(const __attribute__((address_space(3))) f32x2*)(uintptr_t)a; }
__device__ __forceinline__ void lds_st64f(unsigned a, f32x2 v) { *(__attribute__((address_space(3))) f32x2*)(uintptr_t)a = v; }
__device__ __forceinline__ float lds_ld32f(unsigned a) { return *(const __attribute__((address_space(3))) float*)(uintptr_t)a; }
__device__ __forceinline__ void lds_st32f(unsigned a, float v) { *(__attribute__((address_space(3))) float*)(uintptr_t)a = v; }
// -DODVAE_W4_ABL=<bits>: timing-only ablation builds (results are wrong): 1 no output transform, 2 no input transform in the loop,
// 4 no weight refills, 8 no A-fragment reads, 16 no halo DMA in the loop, 32 no wait for the DMA at the end of a chunk, 64 no barrier,
// 128 epilogue without the stores, 256 without the LDS exchange, 512 without its barriers, 1024 every other weight refill only,
// 2048 no output transform at all
#ifndef ODVAE_W4_ABL
#define ODVAE_W4_ABL 0
#endif
template <int N> __device__ __forceinline__ void wait_vm_but() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// one row of (B^T d) times B: six values t[0..5] along x -> the six V entries of that row, stored 1 KB apart
__device__ __forceinline__ void column_pass_store(const f32x2 (&t)[6], unsigned dst) {
  const f32x2 a = t[4] - 4.f * t[2], b = t[3] - 4.f * t[1];
  const f32x2 a2 = t[4] - t[2], e = t[3] - t[1];
  lds_st64f(dst + 0 * 1024, 4.f * t[0] + (t[4] - 5.f * t[2]));
  lds_st64f(dst + 1 * 1024, a + b);
  lds_st64f(dst + 2 * 1024, a - b);
  lds_st64f(dst + 3 * 1024, a2 + 2.f * e);
  lds_st64f(dst + 4 * 1024, a2 - 2.f * e);
  lds_st64f(dst + 5 * 1024, 4.f * t[1] + (t[5] - 5.f * t[3]));
}

// STATS: the output transform also leaves the GroupNorm statistics of y (p.gn_partial).  A template parameter, not a run-time branch: the
// two accumulators and the exchange live only in the instantiation the statistics launches use, so the data-gradient launches and every
// launch without a GroupNorm consumer run the plain build (255 registers, no scratch).
// UP: the Upsample form (x at half resolution, Wino4Params::up), likewise compiled in only where it is launched.
// EPI_GNBWD: the reduce pass of the GroupNorm backward that follows a data-gradient launch, from the same output transform (Wino4Params):
// that pass otherwise streams gx and da from HBM once more (5.7 of the f32 step's 232 ms); here da is in registers and only gx is read.
template <int EPI, bool UP>
__global__ __launch_bounds__(512) void conv3x3_wino4_kernel(Wino4Params p) {
  constexpr bool STATS = EPI == EPI_STATS, GNB = EPI == EPI_GNBWD, POOL = EPI == EPI_POOL;
  extern __shared__ __attribute__((aligned(16))) float dsmem[];
  const unsigned lds0 = lds_addr_of(dsmem);
  const unsigned halo0 = lds0, v0 = lds0 + 2 * HALO_B;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, h = lane >> 5;
  const int g = wave >> 1, ct = wave & 1;          // MFMA role: xi group, co tile
  // Tile sequence of this block.  Not persistent: the one tile blockIdx.x names, output-channel block blockIdx.y.  Persistent (1-D
  // grid of G blocks, G % (8 ny) == 0): block b works on co block (b >> 3) % ny and on tiles i_b, i_b + TL, ... of the
  // XCD-contiguous order, TL = G / ny, i_b % 8 == b % 8 (the XCD the block runs on) -- the scheme of the F(2x2) kernel.
  const int total_tiles = p.tiles_x * p.tiles_y * p.N;
  const int ny = p.CoutP / BN;
  const int TL = p.persist ? (int)gridDim.x / ny : 0;
  const int yblk = p.persist ? ((int)blockIdx.x >> 3) % ny : (int)blockIdx.y;
  int s_cur = p.persist ? ((int)blockIdx.x & 7) + 8 * (((int)blockIdx.x >> 3) / ny) : (int)blockIdx.x;
  struct Tile { int oy0, ox0, n; };
  auto decode = [&](int sidx) {
    int t = (p.persist || p.xcd) ? xcd_contiguous(sidx, p.persist ? total_tiles : (int)gridDim.x) : sidx;
    Tile T;
    const int tx = t % p.tiles_x; t /= p.tiles_x;
    const int ty = t % p.tiles_y; T.n = t / p.tiles_y;
    T.oy0 = ty * TH; T.ox0 = tx * TW;
    return T;
  };
  Tile cur = decode(s_cur);
  const int n0 = yblk * BN;

  // ---- halo by LDS-DMA: wave w issues instructions w, w + 8, w + 16 of a chunk (64 slots of 16 bytes each) ----
  i32x4_t xrs;
  unsigned hvoff[HALO_DMA_PER_WAVE];
  auto set_halo = [&](const Tile& T) {
    const int xh = UP ? p.H >> 1 : p.H, xw = UP ? p.W >> 1 : p.W;      // rows / columns of x in memory
    xrs = rsrc_words(p.x + (int64_t)T.n * xh * xw * p.Cin, (unsigned)(xh * xw * p.Cin) * 4u);
    // (the slot -> (plane, row, x) arithmetic depends on the lane only; from an opaque copy of the lane id it is redone per tile --
    // a dozen integer operations -- instead of being hoisted out of the tile loop and parked in scratch across the main loop)
    int lane_h = lane;
    asm volatile("" : "+v"(lane_h));
#pragma unroll
    for (int k = 0; k < HALO_DMA_PER_WAVE; ++k) {
      const int s = (wave + 8 * k) * 64 + lane_h;
      const int quad = s >= PLANE ? 1 : 0;
      const int rem = s - quad * PLANE;
      const int hr = rem / ROWP, xp = rem - hr * ROWP;
      const int hx = 4 * (xp % 9) + xp / 9;
      const int iy = T.oy0 - 1 + hr, ix = T.ox0 - 1 + hx;
      const bool ok = s < HALO_SLOTS && xp < 36 && hx < TW + 2 && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
      hvoff[k] = ok ? (unsigned)((((UP ? iy >> 1 : iy) * xw + (UP ? ix >> 1 : ix)) * p.Cin + 4 * quad) * 4) : OOB;
    }
  };
  set_halo(cur);
  auto dma_halo = [&](int ch, int stage) {
    const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane(ch * (KC * 4));
#pragma unroll
    for (int k = 0; k < HALO_DMA_PER_WAVE; ++k)
      lds_dma16_s(xrs, (unsigned)__builtin_amdgcn_readfirstlane((int)(halo0 + stage * HALO_B + (wave + 8 * k) * 1024u)), hvoff[k], soff);
  };

  // ---- input transform: wave -> (rows of B^T d, half of the tile rows); lane -> (channel pair, tile) ----
  const int role = wave >> 1, hh = wave & 1;       // 0: rows 1,2   1: rows 3,4   2: row 5   3: row 0
  const int pb = lane & 1, ttx = (lane >> 1) & 7, ty4 = 2 * hh + ((lane >> 4) & 1), tquad = lane >> 5;
  const unsigned t_rd = (unsigned)((tquad * PLANE + 4 * ty4 * ROWP + ttx) * 16 + pb * 8);
  const unsigned t_wr = (unsigned)(tquad * 512 + (ty4 * 8 + ttx) * 16 + pb * 8);
  constexpr unsigned RB = ROWP * 16;
#define ODVAE_W4_D(r, c) lds_ld64f(rd + (r) * RB + ((((c) & 3) * 9 + ((c) >> 2)) * 16))
  auto transform = [&](unsigned hsrc, unsigned vdst) {
    unsigned rd = hsrc + t_rd, wr = vdst + t_wr;
    asm volatile("" : "+v"(rd), "+v"(wr));      // opaque: the secondary base registers of the ds_read2 / ds_write pairs are re-derived per call, not kept
    if (role == 0) {
      f32x2 t1[6], t2[6];
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        const f32x2 d1 = ODVAE_W4_D(1, c), d2 = ODVAE_W4_D(2, c), d3 = ODVAE_W4_D(3, c), d4 = ODVAE_W4_D(4, c);
        const f32x2 a = d4 - 4.f * d2, b = d3 - 4.f * d1;
        t1[c] = a + b; t2[c] = a - b;
      }
      column_pass_store(t1, wr + 6 * 1024);
      column_pass_store(t2, wr + 12 * 1024);
    } else if (role == 1) {
      f32x2 t3[6], t4[6];
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        const f32x2 d1 = ODVAE_W4_D(1, c), d2 = ODVAE_W4_D(2, c), d3 = ODVAE_W4_D(3, c), d4 = ODVAE_W4_D(4, c);
        const f32x2 a = d4 - d2, e = d3 - d1;
        t3[c] = a + 2.f * e; t4[c] = a - 2.f * e;
      }
      column_pass_store(t3, wr + 18 * 1024);
      column_pass_store(t4, wr + 24 * 1024);
    } else if (role == 2) {
      f32x2 t5[6];
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        const f32x2 d1 = ODVAE_W4_D(1, c), d3 = ODVAE_W4_D(3, c), d5 = ODVAE_W4_D(5, c);
        t5[c] = 4.f * d1 + (d5 - 5.f * d3);
      }
      column_pass_store(t5, wr + 30 * 1024);
    } else {
      f32x2 t0[6];
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        const f32x2 d0 = ODVAE_W4_D(0, c), d2 = ODVAE_W4_D(2, c), d4 = ODVAE_W4_D(4, c);
        t0[c] = 4.f * d0 + (d4 - 5.f * d2);
      }
      column_pass_store(t0, wr);
    }
  };
#undef ODVAE_W4_D

  // ---- weight fragments: buffer loads with a scalar (xi, chunk) offset, as in the F(2x2) kernel -- but issued from inline asm, like
  // the halo DMA: vmcnt retires in order, and hipcc, counting only the loads it knows, asks for `vmcnt(8)` in front of every xi, which
  // with the three DMA instructions of the chunk in the queue means "the halo issued a moment ago must have landed" from xi 6 on
  // (measured: 8.6 % of the kernel).  With every vector-memory instruction of the loop hidden from it the waits are written by hand:
  // a fragment is awaited with exactly the loads younger than it left in flight.
  const int QT = p.CinP / 4;
  const int nchunks = p.CinP / KC;
  const i32x4_t urs = rsrc_words(p.upk, (unsigned)(36 * p.CinP * p.CoutP) * 4u);
  const unsigned b_voff = (unsigned)((h * p.CoutP + n0 + ct * 32 + li) * 16);
  const int b_row = p.CoutP * 16;
  auto load_b = [&](int ch, int j, f32x4& dst) {
    const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane(((9 * g + j) * QT + 2 * ch) * b_row);
    // (s_nop: the scalar offset may have been written by a VALU instruction -- v_readlane of a spilled SGPR -- which a vector-memory
    // instruction may read only five wait states later; hipcc does not look into inline asm for that hazard)
    asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(dst) : "v"(b_voff), "s"(urs), "s"(soff) : "memory");
  };
#define ODVAE_W4_AWAIT(N, reg) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(reg) : "n"(N) : "memory")

  f32x16 acc[9];

  // the last two chunks have no halo to fetch: their DMA slots run against an empty descriptor (nothing is read, zeros land in a
  // halo stage nobody reads any more), so that the number of loads in flight is the same in every chunk
  auto dma_halo_or_none = [&](int ch, int stage) {
    const bool live = ch < nchunks;
    i32x4_t r = xrs;
    r.z = live ? xrs.z : 0;
    const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane(live ? ch * (KC * 4) : 0);
#pragma unroll
    for (int k = 0; k < HALO_DMA_PER_WAVE; ++k)
      lds_dma16_s(r, (unsigned)__builtin_amdgcn_readfirstlane((int)(halo0 + stage * HALO_B + (wave + 8 * k) * 1024u)), hvoff[k], soff);
  };

  f32x4 b[9];
  dma_halo(0, 0);
#pragma unroll
  for (int j = 0; j < 9; ++j) load_b(0, j, b[j]);
  dma_halo_or_none(1, 1);
  wait_vm_but<9 + HALO_DMA_PER_WAVE>();
  __syncthreads();
  transform(halo0, v0);
  wait_vm_but<0>();
  __syncthreads();

  // Loop body per xi: the A fragment of xi+1 is requested before the four MFMAs of xi, the weight fragment of xi is re-requested for
  // the next chunk right behind them (past the last chunk: for chunk 0 of the next tile).  The
  // two waves of a SIMD run their share of the input transform at different points of the chunk (after xi 2 / after xi 6), so one
  // of them always has MFMAs to issue while the other waits on LDS.
  // Loads younger than fragment xi of this chunk when it is awaited: the fragments xi+1 .. 8 requested in the previous chunk, the
  // three DMA instructions from the top of this one, the refills 0 .. xi-1: always 11.
  // Persistent form: the next tile's first two halo chunks are requested before the output transform of this one and are awaited in the middle of it (counted: the
  // stores of the passes since are the only younger vector-memory instructions), so a tile after the first starts with its first
  // input transform instead of an HBM round trip.
  const unsigned a_off = (unsigned)(9 * g * 1024 + lane * 16);
  const int t_at = wave < 4 ? 2 : 6;
  for (;;) {      // tiles of this block
  const int s_nxt = s_cur + TL;
  const bool has_next = p.persist && s_nxt < total_tiles;
#pragma unroll
  for (int j = 0; j < 9; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  for (int ch = 0; ch < nchunks; ++ch) {
    const unsigned Vc = v0 + (ch & 1) * V_B + a_off;
    const bool more = ch + 1 < nchunks;
    const int chn = more ? ch + 1 : 0;      // past the last chunk: chunk 0 again, the next tile's first fragments (same output channels)
    if (!(ODVAE_W4_ABL & 16)) dma_halo_or_none(ch + 2, ch & 1);
    f32x4 a = lds_ld128f(Vc);
#pragma unroll
    for (int j = 0; j < 9; ++j) {
      f32x4 an = a;
      if (j < 8 && !(ODVAE_W4_ABL & 8)) an = lds_ld128f(Vc + (j + 1) * 1024);
      ODVAE_W4_AWAIT((ODVAE_W4_ABL & 20) ? 0 : (ODVAE_W4_ABL & 1024) ? 4 + HALO_DMA_PER_WAVE : 8 + HALO_DMA_PER_WAVE, b[j]);
      acc[j] = mfma32(a.x, b[j].x, acc[j]);
      acc[j] = mfma32(a.y, b[j].y, acc[j]);
      acc[j] = mfma32(a.z, b[j].z, acc[j]);
      acc[j] = mfma32(a.w, b[j].w, acc[j]);
      if (!(ODVAE_W4_ABL & 4) && (!(ODVAE_W4_ABL & 1024) || (j & 1) == 0)) load_b(chn, j, b[j]);
      a = an;
      if (!(ODVAE_W4_ABL & 2) && (j == 2 || j == 6) && j == t_at && more)
        transform(halo0 + ((ch + 1) & 1) * HALO_B, v0 + ((ch + 1) & 1) * V_B);
    }
    if (!(ODVAE_W4_ABL & 32)) wait_vm_but<(ODVAE_W4_ABL & 20) ? 0 : (ODVAE_W4_ABL & 1024) ? 5 : 9>();      // this wave's halo pieces of chunk ch+2 (older than the nine refills) have landed
    if (!(ODVAE_W4_ABL & 64)) __syncthreads();
  }
  wait_vm_but<0>();      // the last refills have landed: their registers stay live across the output transform
  const int n = cur.n, oy0 = cur.oy0, ox0 = cur.ox0;
  if (has_next) {
    cur = decode(s_nxt);
    set_halo(cur);
    dma_halo(0, 0);
    dma_halo_or_none(1, 1);
  }

  // ---- output transform, one tile row (= accumulator registers 4 rq .. 4 rq + 3) at a time through X[ct][xi][e][lane] ----
  // (its per-lane constants are derived from an opaque copy of the lane id: computed here, once per tile, instead of being hoisted
  // out of the tile loop and kept -- spilled -- across the main loop)
  int lane_e = lane;
  asm volatile("" : "+v"(lane_e));
  const int li_e = lane_e & 31, h_e = lane_e >> 5;
  const unsigned X0 = v0;
  const int img_bytes = POOL ? (p.H >> 1) * (p.W >> 1) * p.Cout * 4 : p.H * p.W * p.Cout * 4;
  const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(p.y + (int64_t)n * (img_bytes >> 2), 0, img_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.residual ? p.residual : p.y) + (int64_t)n * p.H * p.W * p.Cout, 0, p.residual ? img_bytes : 0, 0x00020000);
  const int ct2 = wave & 1, e2 = wave >> 1;        // the (co tile, register) this thread finishes in every pass
  const int co = n0 + ct2 * 32 + li_e;
  const float bv = (!GNB && p.bias && co < p.Cout) ? p.bias[co] : 0.f;      // (a data gradient has neither bias nor residual)
  const __amdgpu_buffer_rsrc_t gxrsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(GNB ? p.gn_x : p.y) + (int64_t)n * p.H * p.W * p.Cout, 0, GNB ? img_bytes : 0, 0x00020000);
  float g_mu = 0.f, g_rs = 0.f, g_ga = 0.f, g_be = 0.f;      // GNB: this thread's channel of the GroupNorm (one channel, one image per tile)
  if (GNB && co < p.Cout) {
    g_mu = p.gn_mean[n * p.gn_groups + co / p.gn_cpg]; g_rs = p.gn_rstd[n * p.gn_groups + co / p.gn_cpg];
    g_ga = p.gn_gamma[co]; g_be = p.gn_beta[co];
  }
  const int cstep = p.Cout * 4, rstep = (POOL ? p.W >> 1 : p.W) * p.Cout * 4;
  const unsigned x_wr = (unsigned)(X0 + ((ct * 36 + 9 * g) * 4 * 64 + lane_e) * 4);
  const unsigned x_rd = (unsigned)(X0 + ((ct2 * 36 * 4 + e2) * 64 + lane_e) * 4);
  if (ODVAE_W4_ABL & 2048) {      // timing only: no output transform at all (the accumulators are consumed by one guarded store)
    float sacc = 0.f;
#pragma unroll
    for (int j = 0; j < 9; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc += acc[j][r];
    if (sacc == 12345.678f) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(sacc), yrsrc, 0, 0, 0);
  }
#pragma unroll
  for (int rq = 0; rq < ((ODVAE_W4_ABL & 2048) ? 0 : 4); ++rq) {
#pragma unroll
    for (int j = 0; j < 9; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) if (!(ODVAE_W4_ABL & 256)) lds_st32f(x_wr + (j * 4 + e) * 256, acc[j][4 * rq + e]);
    // element (register 4 rq + e2, lane) is tile (row rq, column 4 h + e2): a 4x4 pixel block of one output channel
    const int py = oy0 + 4 * rq, px = ox0 + 4 * (4 * h_e + e2);
    const unsigned base = !(py < p.H && px < p.W && co < p.Cout) ? OOB
                        : POOL ? (unsigned)((((py >> 1) * (p.W >> 1) + (px >> 1)) * p.Cout + co) * 4) : (unsigned)(((py * p.W + px) * p.Cout + co) * 4);
    float seed[4][4];      // bias (+ residual, requested before the exchange barrier)
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int c = 0; c < 4; ++c) seed[a][c] = bv;
    if (!GNB && p.residual) {
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c)
          seed[a][c] += __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rrsrc, base, a * rstep + c * cstep, 0));
    }
    float gx[4][4];       // GNB: the GroupNorm input at this thread's 16 pixels (requested before the exchange barrier, like a residual)
    if (GNB) {
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c)
          gx[a][c] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(gxrsrc, base, a * rstep + c * cstep, 0));
    }
    if (!(ODVAE_W4_ABL & 512)) __syncthreads();
    float psum = 0.f, psq = 0.f;      // STATS: this pass's share of the GroupNorm statistics; parked in LDS at the end of the pass, so
                                      // that nothing of it stays in a register across the passes (with two accumulators live over
                                      // the whole output transform hipcc spilled nine registers into scratch)
    float pool[2] = {0.f, 0.f};
    float tt[6][4];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      float m[6];
#pragma unroll
      for (int j = 0; j < 6; ++j) m[j] = (ODVAE_W4_ABL & 256) ? acc[(6 * i + j) % 9][4 * rq + (i & 3)] : lds_ld32f(x_rd + (6 * i + j) * 1024);
      const float s1 = m[1] + m[2], d1 = m[1] - m[2], s2 = m[3] + m[4], d2 = m[3] - m[4];
      tt[i][0] = m[0] + s1 + s2;
      tt[i][1] = d1 + 2.f * d2;
      tt[i][2] = s1 + 4.f * s2;
      tt[i][3] = d1 + 8.f * d2 + m[5];
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float s1 = tt[1][c] + tt[2][c], d1 = tt[1][c] - tt[2][c], s2 = tt[3][c] + tt[4][c], d2 = tt[3][c] - tt[4][c];
      float yv[4];
      yv[0] = tt[0][c] + s1 + s2 + seed[0][c];
      yv[1] = d1 + 2.f * d2 + seed[1][c];
      yv[2] = s1 + 4.f * s2 + seed[2][c];
      yv[3] = d1 + 8.f * d2 + tt[5][c] + seed[3][c];
      if (POOL) {      // rows (0,1) and (2,3) of this column join the column pair's sums; stored after the second column of a pair
        if ((c & 1) == 0) { pool[0] = yv[0] + yv[1]; pool[1] = yv[2] + yv[3]; }
        else {
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(pool[0] + (yv[0] + yv[1])), yrsrc, base, (c >> 1) * cstep, 0);
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(pool[1] + (yv[2] + yv[3])), yrsrc, base, rstep + (c >> 1) * cstep, 0);
        }
      } else {
#pragma unroll
        for (int a = 0; a < 4; ++a)
          if (!(ODVAE_W4_ABL & 128) || yv[a] == 12345.678f)
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(yv[a]), yrsrc, base, a * rstep + c * cstep, 0);
      }
      if (STATS && base != OOB) {      // (a 4x4 tile is inside the image as a whole: H and W are multiples of 4)
#pragma unroll
        for (int a = 0; a < 4; ++a) { psum += yv[a]; psq += yv[a] * yv[a]; }
      }
      if (GNB && base != OOB) {        // psum: du * xhat, psq: du  (gn_bwd_reduce_kernel's a, b)
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          const float xh = (gx[a][c] - g_mu) * g_rs;
          const float u = xh * g_ga + g_be;
          const float sg = __builtin_amdgcn_rcpf(1.f + __expf(-u));
          const float du = yv[a] * (sg * (1.f + u * (1.f - sg)));
          psum += du * xh; psq += du;
        }
      }
    }
    if (STATS || GNB) {
      lds_st32f(lds0 + LDS_B + (unsigned)((rq * 512 + tid) * 8), psum);
      lds_st32f(lds0 + LDS_B + (unsigned)((rq * 512 + tid) * 8 + 4), psq);
    }
    // the next tile's halo: with a residual the 16 loads + 16 stores of pass 0 are younger than it, without one the 32 stores of passes 0, 1
    // (EPI_POOL writes 4 values per pass: its two halo chunks are awaited outright after pass 1 -- 8 stores younger)
    if (has_next && rq == ((!GNB && p.residual) || GNB ? 0 : 1)) { if (POOL) wait_vm_but<8>(); else wait_vm_but<32>(); }
    if (rq < 3 && !(ODVAE_W4_ABL & 512)) __syncthreads();     // X is rewritten by the next pass
  }
  // ---- GroupNorm statistics of this tile for the layer that reads y: (sum, sum of squares) per channel group, one slot per
  // (image, tile, group) written by exactly one block -- the consumer's finalize kernel adds the tiles up in f64, in fixed order ----
  if (STATS || GNB) {
    const int cpg = GNB ? 1 : p.gn_cpg;      // GNB: sums per channel, not per group
    float gsum = 0.f, gsq = 0.f;      // this thread's four passes, in pass order (each thread reads back what it wrote itself)
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      gsum += lds_ld32f(lds0 + LDS_B + (unsigned)((rq * 512 + tid) * 8));
      gsq += lds_ld32f(lds0 + LDS_B + (unsigned)((rq * 512 + tid) * 8 + 4));
    }
    gsum += __shfl_xor(gsum, 32, 64); gsq += __shfl_xor(gsq, 32, 64);          // the two tile columns of a lane pair
    for (int o = 1; o < cpg; o <<= 1) { gsum += __shfl_xor(gsum, o, 64); gsq += __shfl_xor(gsq, o, 64); }   // the channels of a group
    __syncthreads();                 // the last pass's reads of X are done: its first 2 KB become the exchange area S[e2][ct2][32][2]
    if (h_e == 0 && (li_e & (cpg - 1)) == 0) {
      lds_st32f(X0 + (unsigned)((((wave >> 1) * 2 + ct2) * 32 + li_e) * 8), gsum);
      lds_st32f(X0 + (unsigned)((((wave >> 1) * 2 + ct2) * 32 + li_e) * 8 + 4), gsq);
    }
    __syncthreads();
    int tid_e = tid;                 // (opaque, as lane_e above: the exchange addresses are made here, not parked in scratch across the main loop)
    asm volatile("" : "+v"(tid_e));
    if (tid_e < 64 && (tid_e & (cpg - 1)) == 0) {       // thread = (co tile tid >> 5, first lane of a group)
      const int ctq = tid_e >> 5, l = tid_e & 31, cg = n0 + ctq * 32 + l;
      if (cg < p.Cout) {
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          a += lds_ld32f(X0 + (unsigned)(((e * 2 + ctq) * 32 + l) * 8));
          b += lds_ld32f(X0 + (unsigned)(((e * 2 + ctq) * 32 + l) * 8 + 4));
        }
        const int tile_in_image = (oy0 / TH) * p.tiles_x + ox0 / TW;
        if (GNB) {
          float* dst = p.gn_partial + ((int64_t)n * (p.tiles_x * p.tiles_y) + tile_in_image) * 2 * p.Cout;
          dst[cg] = a; dst[p.Cout + cg] = b;
        } else {
          float* dst = p.gn_partial + (((int64_t)n * (p.tiles_x * p.tiles_y) + tile_in_image) * p.gn_groups + cg / cpg) * 2;
          dst[0] = a; dst[1] = b;
        }
      }
    }
  }
  if (!has_next) break;
  s_cur = s_nxt;
  __syncthreads();         // X (over both V stages) has been read
  transform(halo0, v0);
  __syncthreads();
  }      // tiles
}

// U[xi = 6a + b][ci][co] = (G g G^T)[a][b];  dgrad: g taken with flipped taps and swapped channel roles.  One thread per
// (co, ci) pair; padding entries of the packs are zero-filled by the launcher beforehand.
// items != null: batched launch, blockIdx.y selects the weight (unpadded channel counts: the pads are the counts themselves)
__global__ void conv3x3_pack_wino4_kernel(const float* __restrict__ w, int Cout, int Cin,
                                          float* __restrict__ fwd, int CinP_f, int CoutP_f,
                                          float* __restrict__ dgr, int CoutP_d, int CinP_d, const OdvaePackItem* __restrict__ items) {
  if (items) {
    const OdvaePackItem it = items[blockIdx.y];
    w = it.w; Cout = it.Cout; Cin = it.Cin; fwd = static_cast<float*>(it.fwd); dgr = static_cast<float*>(it.dgr);
    CinP_f = Cin; CoutP_f = Cout; CoutP_d = Cout; CinP_d = Cin;
  }
  const int64_t pairs = (int64_t)Cout * Cin;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < pairs; idx += (int64_t)gridDim.x * blockDim.x) {
    const int co = (int)(idx % Cout), ci = (int)(idx / Cout);
    const float* gp = w + ((int64_t)co * Cin + ci) * 9;
    float gg[3][3];
#pragma unroll
    for (int k = 0; k < 9; ++k) gg[k / 3][k % 3] = gp[k];
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      float* dst = pass == 0 ? fwd : dgr;
      if (!dst) continue;
      // G x = (x0/4, -(x0+x1+x2)/6, -(x0-x1+x2)/6, x0/24 + x1/12 + x2/6, x0/24 - x1/12 + x2/6, x2)
      float tv[6][3], u[6][6];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float x0 = pass == 0 ? gg[0][c] : gg[2][2 - c], x1 = pass == 0 ? gg[1][c] : gg[1][2 - c],
                    x2 = pass == 0 ? gg[2][c] : gg[0][2 - c];
        const float e = x0 + x2;
        tv[0][c] = 0.25f * x0;
        tv[1][c] = -(e + x1) * (1.f / 6.f);
        tv[2][c] = -(e - x1) * (1.f / 6.f);
        const float f = x0 * (1.f / 24.f) + x2 * (1.f / 6.f);
        tv[3][c] = f + x1 * (1.f / 12.f);
        tv[4][c] = f - x1 * (1.f / 12.f);
        tv[5][c] = x2;
      }
#pragma unroll
      for (int a = 0; a < 6; ++a) {
        const float x0 = tv[a][0], x1 = tv[a][1], x2 = tv[a][2];
        const float e = x0 + x2;
        u[a][0] = 0.25f * x0;
        u[a][1] = -(e + x1) * (1.f / 6.f);
        u[a][2] = -(e - x1) * (1.f / 6.f);
        const float f = x0 * (1.f / 24.f) + x2 * (1.f / 6.f);
        u[a][3] = f + x1 * (1.f / 12.f);
        u[a][4] = f - x1 * (1.f / 12.f);
        u[a][5] = x2;
      }
      const int red = pass == 0 ? ci : co, out = pass == 0 ? co : ci;
      const int redP = pass == 0 ? CinP_f : CoutP_d, outP = pass == 0 ? CoutP_f : CinP_d;
#pragma unroll
      for (int xi = 0; xi < 36; ++xi)
        dst[(((int64_t)xi * (redP / 4) + red / 4) * outP + out) * 4 + (red & 3)] = u[xi / 6][xi % 6];
    }
  }
}

constexpr int round_up_i(int a, int b) { return (a + b - 1) / b * b; }

}  // namespace

extern "C" {

// pack sizes: reduction axis padded to 8, output axis to 64
int odvae_conv3x3_wino4_reduce_pad(int c_reduce) { return round_up_i(c_reduce, KC); }
int odvae_conv3x3_wino4_out_pad(int c_out) { return round_up_i(c_out, BN); }
size_t odvae_conv3x3_wino4_pack_floats(int c_reduce, int c_out) {
  return (size_t)36 * odvae_conv3x3_wino4_reduce_pad(c_reduce) * odvae_conv3x3_wino4_out_pad(c_out);
}
// shapes the F(4x4) kernel takes (forward: reduce over Cin; the data gradient swaps the roles, so both must qualify)
int odvae_conv3x3_wino4_supported(int H, int W, int Cin, int Cout) {
  return H % 4 == 0 && W % 4 == 0 && H >= TH && W >= TW && Cin % KC == 0 && Cout % KC == 0 && Cin >= BN && Cout >= BN;
}

int odvae_conv3x3_pack_wino4_f32(const float* w, int Cout, int Cin, float* fwd_pack, float* dgrad_pack, void* stream) {
  ODVAE_CHECK_ARG(w && Cout > 0 && Cin > 0, "conv3x3_pack_wino4: bad arguments");
  const int CinP_f = odvae_conv3x3_wino4_reduce_pad(Cin), CoutP_f = odvae_conv3x3_wino4_out_pad(Cout);
  const int CoutP_d = odvae_conv3x3_wino4_reduce_pad(Cout), CinP_d = odvae_conv3x3_wino4_out_pad(Cin);
  if (!fwd_pack && !dgrad_pack) return ODVAE_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (fwd_pack && (CinP_f != Cin || CoutP_f != Cout)) {
    if (hipMemsetAsync(fwd_pack, 0, (size_t)36 * CinP_f * CoutP_f * sizeof(float), st) != hipSuccess) { odvae_set_error("conv3x3_pack_wino4: memset failed"); return ODVAE_ERR_HIP; }
  }
  if (dgrad_pack && (CoutP_d != Cout || CinP_d != Cin)) {
    if (hipMemsetAsync(dgrad_pack, 0, (size_t)36 * CoutP_d * CinP_d * sizeof(float), st) != hipSuccess) { odvae_set_error("conv3x3_pack_wino4: memset failed"); return ODVAE_ERR_HIP; }
  }
  const int blocks = (int)std::min<int64_t>(ceil_div64((int64_t)Cout * Cin, 256), 2048);
  hipLaunchKernelGGL(conv3x3_pack_wino4_kernel, dim3(blocks), dim3(256), 0, st,
                     w, Cout, Cin, fwd_pack, CinP_f, CoutP_f, dgrad_pack, CoutP_d, CinP_d, (const OdvaePackItem*)nullptr);
  ODVAE_LAUNCH_CHECK("conv3x3_pack_wino4");
  return ODVAE_OK;
}

// n weights in ONE launch: items = device array of n OdvaePackItem, channel counts their own pads (multiples of 64).  40-44 launches of
// ~15 us per f32 training step become one.
int odvae_conv3x3_pack_wino4_batch(const void* items, int n, void* stream) {
  ODVAE_CHECK_ARG(items && n > 0 && n <= 65535, "conv3x3_pack_wino4_batch: bad arguments");
  hipLaunchKernelGGL(conv3x3_pack_wino4_kernel, dim3(256, n), dim3(256), 0, static_cast<hipStream_t>(stream),
                     (const float*)nullptr, 0, 0, (float*)nullptr, 0, 0, (float*)nullptr, 0, 0, static_cast<const OdvaePackItem*>(items));
  ODVAE_LAUNCH_CHECK("conv3x3_pack_wino4_batch");
  return ODVAE_OK;
}

// tiles per image of the F(4x4) kernel = the chunk count of its GroupNorm partials
int odvae_conv3x3_wino4_stats_chunks(int H, int W) { return ceil_div(H, TH) * ceil_div(W, TW); }

struct Wino4GnBwd { const float *x, *mean, *rstd, *gamma, *beta; };
static int wino4_launch(const float* x, int N, int H, int W, int Cin, const float* upk, int Cout,
                        const float* bias, const float* residual, float* y, int act, float* gn_partial, int gn_groups, void* stream, int up = 0,
                        const Wino4GnBwd* gnb = nullptr, int pool = 0);

// y = conv3x3_stride1_pad1(x) (+bias) (+residual); upk = fwd or dgrad pack of odvae_conv3x3_pack_wino4_f32; act must be 0
// (a fused ReLU is not offered: ops.py keeps the ReLU convs of the VGG stack on F(2x2) for accuracy, see there).
int odvae_conv3x3_wino4_f32(const float* x, int N, int H, int W, int Cin, const float* upk, int Cout,
                            const float* bias, const float* residual, float* y, int act, void* stream) {
  return wino4_launch(x, N, H, W, Cin, upk, Cout, bias, residual, y, act, nullptr, 0, stream);
}

// The same, and the output transform also leaves the GroupNorm statistics of y for the layer that reads it:
// gn_partial [N][odvae_conv3x3_wino4_stats_chunks(H, W)][gn_groups][2] = (sum, sum of squares) of y per tile and channel group
// (every slot written, by exactly one block) -- the input of odvae_groupnorm_fwd_partials_f32, which then needs no statistics pass.
// Cout / gn_groups must be a power of two <= 32.
int odvae_conv3x3_wino4_stats_f32(const float* x, int N, int H, int W, int Cin, const float* upk, int Cout,
                                  const float* bias, const float* residual, float* y, float* gn_partial, int gn_groups, void* stream) {
  ODVAE_CHECK_ARG(gn_partial && gn_groups > 0 && Cout % gn_groups == 0, "conv3x3_wino4_stats: bad statistics arguments");
  const int cpg = Cout / gn_groups;
  ODVAE_CHECK_ARG(cpg <= 32 && (cpg & (cpg - 1)) == 0, "conv3x3_wino4_stats: %d channels per group (needs a power of two <= 32)", cpg);
  return wino4_launch(x, N, H, W, Cin, upk, Cout, bias, residual, y, 0, gn_partial, gn_groups, stream);
}

// Upsample conv ([UPSTREAM] Upsample.forward: F.interpolate(scale 2, nearest) then conv3x3) on the same kernel: x is the low-resolution
// input [N][H/2][W/2][Cin], y [N][H][W][Cout]; the 4x intermediate is never formed -- the halo DMA reads x[iy >> 1][ix >> 1].  2.25
// multiply-adds per output pixel and (ci, co) against the 4 of the parity-class kernels (conv3x3_f32.hip mode 5).  gn_partial /
// gn_groups as in odvae_conv3x3_wino4_stats_f32, or NULL / 0.
int odvae_conv3x3_wino4_up_f32(const float* x, int N, int H, int W, int Cin, const float* upk, int Cout,
                               const float* bias, const float* residual, float* y, float* gn_partial, int gn_groups, void* stream) {
  ODVAE_CHECK_ARG(H % 2 == 0 && W % 2 == 0, "conv3x3_wino4_up: output %dx%d is not twice an input size", H, W);
  if (gn_partial) {
    ODVAE_CHECK_ARG(gn_groups > 0 && Cout % gn_groups == 0, "conv3x3_wino4_up: bad statistics arguments");
    const int cpg = Cout / gn_groups;
    ODVAE_CHECK_ARG(cpg <= 32 && (cpg & (cpg - 1)) == 0, "conv3x3_wino4_up: %d channels per group (needs a power of two <= 32)", cpg);
  }
  return wino4_launch(x, N, H, W, Cin, upk, Cout, bias, residual, y, 0, gn_partial, gn_partial ? gn_groups : 0, stream, 1);
}

// Data gradient of a conv whose input was a = swish(GroupNorm(gn_x)) (x = the conv's dy, upk = its dgrad pack, y = da [N][H][W][Cout]), and
// the first pass of that GroupNorm's backward from the same output transform: gn_partial [N][odvae_conv3x3_wino4_stats_chunks(H, W)][2][Cout]
// = per tile and channel (sum du * xhat, sum du), every slot written by exactly one block -- the input of odvae_groupnorm_bwd_partials_f32.
int odvae_conv3x3_wino4_gnbwd_f32(const float* x, int N, int H, int W, int Cin, const float* upk, int Cout, float* y,
                                  const float* gn_x, const float* gn_mean, const float* gn_rstd, const float* gn_gamma, const float* gn_beta,
                                  int gn_groups, float* gn_partial, void* stream) {
  ODVAE_CHECK_ARG(gn_x && gn_mean && gn_rstd && gn_gamma && gn_beta && gn_partial, "conv3x3_wino4_gnbwd: null GroupNorm operand");
  ODVAE_CHECK_ARG(gn_groups > 0 && Cout % gn_groups == 0, "conv3x3_wino4_gnbwd: %d channels in %d groups", Cout, gn_groups);
  const Wino4GnBwd g{gn_x, gn_mean, gn_rstd, gn_gamma, gn_beta};
  return wino4_launch(x, N, H, W, Cin, upk, Cout, nullptr, nullptr, y, 0, gn_partial, gn_groups, stream, 0, &g);
}

// Data gradient of an Upsample conv w.r.t. its LOW-resolution input: x = the conv's dy [N][H][W][Cin], upk = its data-gradient pack,
// y [N][H/2][W/2][Cout] = the 2x2 sums of the full-resolution gradient (formed in the output transform; that gradient is never stored).
int odvae_conv3x3_wino4_pool_f32(const float* x, int N, int H, int W, int Cin, const float* upk, int Cout, float* y, void* stream) {
  return wino4_launch(x, N, H, W, Cin, upk, Cout, nullptr, nullptr, y, 0, nullptr, 0, stream, 0, nullptr, 1);
}

static int wino4_launch(const float* x, int N, int H, int W, int Cin, const float* upk, int Cout,
                        const float* bias, const float* residual, float* y, int act, float* gn_partial, int gn_groups, void* stream, int up,
                        const Wino4GnBwd* gnb, int pool) {
  ODVAE_CHECK_ARG(x && upk && y, "conv3x3_wino4: null operand");
  ODVAE_CHECK_ARG(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "conv3x3_wino4: empty shape");
  ODVAE_CHECK_ARG(act == 0, "conv3x3_wino4: no fused activation (act=%d); ReLU convs stay on odvae_conv3x3_wino_f32", act);
  ODVAE_CHECK_ARG(H % 4 == 0 && W % 4 == 0 && Cin % KC == 0, "conv3x3_wino4: needs H, W in multiples of 4 and Cin %% 8 == 0 (H=%d W=%d Cin=%d)", H, W, Cin);
  ODVAE_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)upk & 15) == 0, "conv3x3_wino4: x/upk must be 16-byte aligned");
  ODVAE_CHECK_ARG((int64_t)H * W * Cin * 4 < 0x7FFFFFF0ll && ((int64_t)(H + 3) * W + 3) * Cout * 4 < 0x7FFFFFF0ll,
                  "conv3x3_wino4: one input / output image must stay below 2 GiB");
  Wino4Params p;
  p.x = x; p.upk = upk; p.bias = bias; p.residual = residual; p.y = y;
  p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout;
  p.CinP = odvae_conv3x3_wino4_reduce_pad(Cin); p.CoutP = odvae_conv3x3_wino4_out_pad(Cout);
  p.tiles_x = ceil_div(W, TW); p.tiles_y = ceil_div(H, TH); p.act = act;
  p.gn_partial = gn_partial; p.gn_groups = gn_groups; p.gn_cpg = gn_groups > 0 ? Cout / gn_groups : 0;
  p.up = up;
  p.gn_x = gnb ? gnb->x : nullptr; p.gn_mean = gnb ? gnb->mean : nullptr; p.gn_rstd = gnb ? gnb->rstd : nullptr;
  p.gn_gamma = gnb ? gnb->gamma : nullptr; p.gn_beta = gnb ? gnb->beta : nullptr;
  ODVAE_CHECK_ARG((int64_t)36 * p.CinP * p.CoutP * 4 < 0x7FFFFFF0ll, "conv3x3_wino4: pack too large");
  const int64_t sp = (int64_t)p.tiles_x * p.tiles_y * N;
  ODVAE_CHECK_ARG(sp < (1ll << 31), "conv3x3_wino4: too many tiles");
  static const bool xcd = getenv("ODVAE_TILE_XCD") == nullptr || atoi(getenv("ODVAE_TILE_XCD")) != 0;
  p.xcd = xcd ? 1 : 0;
  const auto kern = gnb ? conv3x3_wino4_kernel<EPI_GNBWD, false>
                  : pool ? conv3x3_wino4_kernel<EPI_POOL, false>
                  : up ? (gn_partial ? conv3x3_wino4_kernel<EPI_STATS, true> : conv3x3_wino4_kernel<EPI_NONE, true>)
                       : (gn_partial ? conv3x3_wino4_kernel<EPI_STATS, false> : conv3x3_wino4_kernel<EPI_NONE, false>);
  const unsigned lds_bytes = gn_partial ? LDS_B + STATS_B : LDS_B;
  const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  if (e != hipSuccess) {
    odvae_set_error("conv3x3_wino4: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
    return ODVAE_ERR_HIP;
  }
  // persistent form: one block per CU; needs a grid that splits evenly over the output-channel blocks and >= 2 tiles per block
  static const bool no_persist = getenv("ODVAE_WINO_PERSIST") != nullptr && atoi(getenv("ODVAE_WINO_PERSIST")) == 0;
  static const int cus = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    return n;
  }();
  const int ny = p.CoutP / BN;
  p.persist = (!no_persist && cus % (8 * ny) == 0 && sp >= 2 * (cus / ny)) ? 1 : 0;
  if (p.persist) hipLaunchKernelGGL(kern, dim3(cus), dim3(512), lds_bytes, static_cast<hipStream_t>(stream), p);
  else hipLaunchKernelGGL(kern, dim3((unsigned)sp, ny), dim3(512), lds_bytes, static_cast<hipStream_t>(stream), p);
  ODVAE_LAUNCH_CHECK("conv3x3_wino4");
  return ODVAE_OK;
}

}  // extern "C"
