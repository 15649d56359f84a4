// Weight gradient of the stride-1 3x3 convolution in the Winograd F(2x2, 3x3) domain, fp32 on v_mfma_f32_32x32x2_f32.
//
// Forward: Y = A^T [ (G g G^T) .* (B^T d B) ] A per 2x2 output tile.  With V = B^T d B (the forward input transform of
// the 4x4 input patch) and dM = A dY A^T (the 2x2 output-gradient tile carried into the 4x4 domain),
//     dU[xi][ci][co] = sum over tiles  V[xi][tile][ci] * dM[xi][tile][co]        (16 independent products, K = tiles)
//     dg = G^T dU G                                                              (once, in the slab reduction)
// i.e. 16 instead of 36 multiply-adds per tile and (ci, co): autograd's `conv2d` weight gradient
// ([UPSTREAM] ldm/modules/diffusionmodules/model.py ResnetBlock.conv1/conv2, Encoder/Decoder conv layers; called from
// src/modules/autoencodermodules/feat_encoder.py:2, feat_decoder.py:2) in 0.44x the MFMA work of conv3x3_wgrad_f32.hip.
//
// Block = 8 waves, one ROW r of the 4x4 domain (xi = 4r + c, c = 0..3) x 128 ci x 128 co = 128 accumulator registers per
// wave (wave = 32-ci group x 64-co half), over a contiguous split of the tiles.  A block transforms only its own row:
// row r of B^T d needs two of the four patch rows, row r of A dY one or both tile rows -- no transform work is repeated
// across the four row blocks.  Per chunk of 16 tiles every thread fetches one (tile, channel quad) of x and of dy
// straight from HBM into registers (a wavefront reads whole 512-byte channel rows), transforms it and writes V and dM
// `[c][tile][128]` into LDS (double-buffered, 128 KB); the MFMA phase reads scalar fragments (lane = channel, the
// two k lanes = two tiles): 64 MFMAs per wave and barrier.  The fetch of chunk k+2 is in flight during the MFMAs of
// chunk k.  Each block writes its partial dU slab; the reduction sums the slabs in a fixed order (deterministic),
// applies G^T . G and writes OIHW.  The bias gradient rides along: dM[xi = (1,1)] is the sum of the tile's four dy pixels.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int CT = 16;               // tiles per chunk
constexpr int RS = 128;              // LDS row stride in floats ([c][tile][channel]); unpadded: every fragment offset is a multiple of 256 bytes, so
                                     // the reads are ds_read2st64_b32 off three base registers with no per-read address arithmetic
constexpr int OPF = 4 * CT * RS;     // floats per operand and stage (8 192)
constexpr unsigned OOB = 0xFFFFFFF0u;

struct WgwParams {
  const float* x; const float* dy; float* slabs; float* bias_part;
  int N, H, W, Cin, Cout, TY, TX;
  int total_tiles, total_chunks, chunks_per_split, nsplit;
};

__device__ __forceinline__ float4 f4(u32x4 v) {
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
// A * x + B * y with A, B in {-1, 0, 1} known at compile time: an add, a subtract, a negation or a copy
template <int A, int B>
__device__ __forceinline__ float4 f4lin(float4 x, float4 y) {
  static_assert(A >= -1 && A <= 1 && B >= -1 && B <= 1 && (A != 0 || B != 0), "coefficients are -1, 0 or 1");
  if constexpr (B == 0) return A > 0 ? x : make_float4(-x.x, -x.y, -x.z, -x.w);
  else if constexpr (A == 0) return B > 0 ? y : make_float4(-y.x, -y.y, -y.z, -y.w);
  else if constexpr (A > 0 && B > 0) return make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
  else if constexpr (A > 0) return make_float4(x.x - y.x, x.y - y.y, x.z - y.z, x.w - y.w);
  else if constexpr (B > 0) return make_float4(y.x - x.x, y.y - x.y, y.z - x.z, y.w - x.w);
  else return make_float4(-x.x - y.x, -x.y - y.y, -x.z - y.z, -x.w - y.w);
}
__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4sub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }

// WIDE: a tile row holds at least CT tiles, the per-chunk tile advance needs no division.  R = the block's row of the 4x4
// domain as a compile-time constant: the +-1 / 0 coefficients of its two transforms fold into adds, subtracts and nothing
// (with run-time coefficients the staging was 24 multiplies + 16 fused multiply-adds per chunk and thread; on gfx950 vector
// work is not hidden under the f32 MFMA, profiles/r02_wino8_loop.md).
template <bool WIDE, int R>
__device__ __forceinline__ void wgw_body(const WgwParams& p, float* dsm, const int split) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cig = wave & 3, coh = wave >> 2;
  const int li = lane & 31, h = lane >> 5;
  constexpr int r = R;
  const int cib = blockIdx.y, cob = blockIdx.z;
  // row r of B^T d = sa * d[ra] + sb * d[rb];  row r of A dY = ya * dy[0] + yb * dy[1]
  constexpr int ra = r == 0 ? 0 : 1, rb = r == 3 ? 3 : 2;
  constexpr int sa = r == 2 ? -1 : 1, sb = (r == 0 || r == 3) ? -1 : 1;
  constexpr int ya = r == 3 ? 0 : 1, yb = r == 0 ? 0 : (r == 1 ? 1 : -1);

  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.x), 0, (unsigned)((int64_t)p.N * p.H * p.W * p.Cin * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.dy), 0, (unsigned)((int64_t)p.N * p.H * p.W * p.Cout * 4), 0x00020000);

  const int chunk0 = split * p.chunks_per_split;
  const int nch = min(p.chunks_per_split, p.total_chunks - chunk0);   // >= 1 by construction of nsplit
  const int end_tile = min(p.total_tiles, (chunk0 + nch) * CT);         // fetches past it return zeros (no traffic)

  const int st = tid >> 5, sq = tid & 31;     // staging role: tile of the chunk, channel quad
  // this thread's tile of the next chunk to fetch, advanced by CT per fetch (one division at the start only when a
  // tile row holds at least CT tiles); all byte offsets fit 32 bits (tensors below 4 GiB, checked by the launcher)
  int tl = chunk0 * CT + st, tn, tty, ttx;
  {
    const int per_img = p.TY * p.TX;
    tn = tl / per_img;
    const int rem = tl - tn * per_img;
    tty = rem / p.TX; ttx = rem - tty * p.TX;
  }
  const unsigned cin4 = (unsigned)p.Cin * 4u, cout4 = (unsigned)p.Cout * 4u;
  const unsigned chx = (unsigned)(cib * 128 + 4 * sq) * 4u, chy = (unsigned)(cob * 128 + 4 * sq) * 4u;
  // byte offsets of this thread's patch origin (input row 2 tty - 1, column 2 ttx - 1: may "be negative", the masks below
  // cover exactly those lanes) and of its dy tile; unsigned arithmetic, consistent modulo 2^32
  unsigned xoff, yoff;
  const unsigned step_x = 2u * CT * cin4, step_x_wrap = step_x + (unsigned)p.W * cin4;
  const unsigned step_y = 2u * CT * cout4, step_y_wrap = step_y + (unsigned)p.W * cout4;
  auto set_bases = [&]() {
    xoff = (unsigned)((tn * p.H + 2 * tty - 1) * p.W + 2 * ttx - 1) * cin4 + chx;
    yoff = (unsigned)((tn * p.H + 2 * tty) * p.W + 2 * ttx) * cout4 + chy;
  };
  set_bases();
  float4 xr[2][4], yr[2][2];
  auto fetch_x = [&]() {
#ifdef ODVAE_WGW_NOFETCH   // ablation build: every fetch is answered with zeros by the descriptor, no memory traffic
    const bool valid = false;
#else
    const bool valid = tl < end_tile;
#endif
    const bool rok0 = valid && (ra != 0 || tty > 0), rok1 = valid && (rb != 3 || tty < p.TY - 1);
    const bool cok0 = ttx > 0, cok3 = ttx < p.TX - 1;
    const unsigned xo0 = xoff + (unsigned)(ra * p.W) * cin4, xo1 = xoff + (unsigned)(rb * p.W) * cin4;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const bool cok = c == 0 ? cok0 : (c == 3 ? cok3 : true);
      xr[0][c] = f4(__builtin_amdgcn_raw_buffer_load_b128(xrsrc, (rok0 && cok) ? xo0 + c * cin4 : OOB, 0, 0));
      xr[1][c] = f4(__builtin_amdgcn_raw_buffer_load_b128(xrsrc, (rok1 && cok) ? xo1 + c * cin4 : OOB, 0, 0));
    }
  };
  auto fetch_y = [&]() {     // and advance to this thread's tile of the following chunk
#ifdef ODVAE_WGW_NOFETCH   // ablation build: every fetch is answered with zeros by the descriptor, no memory traffic
    const bool valid = false;
#else
    const bool valid = tl < end_tile;
#endif
    const unsigned yo0 = yoff, yo1 = yoff + (unsigned)p.W * cout4;
    const bool u0 = valid && ya != 0, u1 = valid && yb != 0;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      yr[0][j] = f4(__builtin_amdgcn_raw_buffer_load_b128(yrsrc, u0 ? yo0 + j * cout4 : OOB, 0, 0));
      yr[1][j] = f4(__builtin_amdgcn_raw_buffer_load_b128(yrsrc, u1 ? yo1 + j * cout4 : OOB, 0, 0));
    }
    tl += CT;
    if constexpr (WIDE) {
      // W = 2 TX and H = 2 TY: a step of CT tiles moves the patch by 2 CT pixels, plus one pixel row when the tile row wraps
      // (tile rows of consecutive images are consecutive), so the offsets advance by adds -- no multiplies per chunk
      const int nx = ttx + CT;
      const bool wrap = nx >= p.TX;
      xoff += wrap ? step_x_wrap : step_x;
      yoff += wrap ? step_y_wrap : step_y;
      ttx = wrap ? nx - p.TX : nx;
      const int ny = tty + (wrap ? 1 : 0);
      tty = ny >= p.TY ? 0 : ny;        // tn itself is only needed by the non-WIDE path
    } else {
      const int per_img = p.TY * p.TX;
      tn = tl / per_img;
      const int rem = tl - tn * per_img;
      tty = rem / p.TX; ttx = rem - tty * p.TX;
      set_bases();
    }
  };
  float4 bias_acc = make_float4(0.f, 0.f, 0.f, 0.f);
  auto stage_x = [&](float* V) {
    float4 w[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) w[c] = f4lin<sa, sb>(xr[0][c], xr[1][c]);
    float* vd = V + st * RS + 4 * sq;
    *reinterpret_cast<float4*>(vd + 0 * CT * RS) = f4sub(w[0], w[2]);
    *reinterpret_cast<float4*>(vd + 1 * CT * RS) = f4add(w[1], w[2]);
    *reinterpret_cast<float4*>(vd + 2 * CT * RS) = f4sub(w[2], w[1]);
    *reinterpret_cast<float4*>(vd + 3 * CT * RS) = f4sub(w[1], w[3]);
  };
  auto stage_y = [&](float* M) {
    const float4 z0 = f4lin<ya, yb>(yr[0][0], yr[1][0]), z1 = f4lin<ya, yb>(yr[0][1], yr[1][1]);
    const float4 m1 = f4add(z0, z1);
    float* md = M + st * RS + 4 * sq;
    *reinterpret_cast<float4*>(md + 0 * CT * RS) = z0;
    *reinterpret_cast<float4*>(md + 1 * CT * RS) = m1;
    *reinterpret_cast<float4*>(md + 2 * CT * RS) = f4sub(z0, z1);
    *reinterpret_cast<float4*>(md + 3 * CT * RS) = make_float4(-z1.x, -z1.y, -z1.z, -z1.w);
    bias_acc = f4add(bias_acc, m1);      // meaningful in the r = 1 blocks: dy00 + dy01 + dy10 + dy11
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[c][nt][q] = 0.f;

  // fragments of k step t2 (tiles 2 t2, 2 t2 + 1) into set s; eight MFMAs per step
  float fa[2][4], fb[2][4][2];
  // The second co-tile is read through its own base register (its +32 is hidden from the compiler): otherwise hipcc pairs the
  // two co-tiles of one (c, k step) in a ds_read2_b32 and spends a v_add on a fresh base for every one of the 32 pairs.
  int nt1 = 32;
  asm volatile("" : "+v"(nt1));
  auto ldf = [&](const float* va, const float* mb, int t2, int s) {
    const float* mb1 = mb + nt1;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      fa[s][c] = va[(c * CT + 2 * t2) * RS];
      fb[s][c][0] = mb[(c * CT + 2 * t2) * RS];
      fb[s][c][1] = mb1[(c * CT + 2 * t2) * RS];
    }
  };
  auto mm = [&](int s) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      acc[c][0] = mfma32(fa[s][c], fb[s][c][0], acc[c][0]);
      acc[c][1] = mfma32(fa[s][c], fb[s][c][1], acc[c][1]);
    }
  };

  // One basic block per iteration, pieces pinned in this order (fragments of step t+1 are requested before the MFMAs of
  // step t; staging of chunk k+1 and the fetch of chunk k+2 ride under the MFMAs of steps 0-3; past the end of the split
  // both move zeros):  [f1] [mm0 | V writes] [f2] [mm1 | dM writes] [f3] [mm2 | x fetch] [f4] [mm3 | dy fetch] ...
  fetch_x(); fetch_y();
  stage_x(dsm); stage_y(dsm + OPF);
  fetch_x(); fetch_y();
  __syncthreads();
  for (int k = 0; k < nch; ++k) {
    const float* cur = dsm + (k & 1) * 2 * OPF;
    float* nxt = dsm + ((k + 1) & 1) * 2 * OPF;
    const float* va = cur + h * RS + cig * 32 + li;
    const float* mb = cur + OPF + h * RS + coh * 64 + li;
    ldf(va, mb, 0, 0);
    ldf(va, mb, 1, 1);
    __builtin_amdgcn_sched_barrier(0);
    mm(0); stage_x(nxt);
    __builtin_amdgcn_sched_barrier(0);
    ldf(va, mb, 2, 0);
    __builtin_amdgcn_sched_barrier(0);
    mm(1); stage_y(nxt + OPF);
    __builtin_amdgcn_sched_barrier(0);
    ldf(va, mb, 3, 1);
    __builtin_amdgcn_sched_barrier(0);
    mm(0); fetch_x();
    __builtin_amdgcn_sched_barrier(0);
    ldf(va, mb, 4, 0);
    __builtin_amdgcn_sched_barrier(0);
    mm(1); fetch_y();
    __builtin_amdgcn_sched_barrier(0);
    ldf(va, mb, 5, 1);
    __builtin_amdgcn_sched_barrier(0);
    mm(0);
    __builtin_amdgcn_sched_barrier(0);
    ldf(va, mb, 6, 0);
    __builtin_amdgcn_sched_barrier(0);
    mm(1);
    __builtin_amdgcn_sched_barrier(0);
    ldf(va, mb, 7, 1);
    __builtin_amdgcn_sched_barrier(0);
    mm(0);
    mm(1);
    __syncthreads();
  }

  // partial dU of this block: slabs[split][xi = 4r + c][ci][co]
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    float* sl = p.slabs + ((int64_t)(split * 16 + 4 * r + c) * p.Cin + cib * 128 + cig * 32) * p.Cout + cob * 128 + coh * 64 + li;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int q = 0; q < 16; ++q) sl[(int64_t)acc_row(q, lane) * p.Cout + nt * 32] = acc[c][nt][q];
  }
  if (r == 1 && cib == 0 && p.bias_part) {   // block-uniform
    *reinterpret_cast<float4*>(dsm + st * 128 + 4 * sq) = bias_acc;
    __syncthreads();
    if (tid < 128) {
      float s = 0.f;
#pragma unroll
      for (int t = 0; t < CT; ++t) s += dsm[t * 128 + tid];
      p.bias_part[(int64_t)split * p.Cout + cob * 128 + tid] = s;
    }
  }
}

template <bool WIDE>
__global__ __launch_bounds__(512, 2) void conv3x3_wgrad_wino_kernel(WgwParams p) {
  extern __shared__ __attribute__((aligned(16))) float dsm[];   // stage s: V at s*2*OPF, dM at s*2*OPF + OPF
  // the four row blocks of one tile split read the same x / dy: linear block ids go round-robin over the 8 XCDs, so
  // ids bx, bx + 8, bx + 16, bx + 24 (one XCD, one L2) take the four rows of one split when the split count allows it
  const int bx = blockIdx.x;
  const bool xcd_map = (p.nsplit & 7) == 0;
  const int r = xcd_map ? (bx >> 3) & 3 : bx & 3;
  const int split = xcd_map ? (bx & 7) + 8 * (bx >> 5) : bx >> 2;
  switch (r) {      // block-uniform
    case 0: wgw_body<WIDE, 0>(p, dsm, split); break;
    case 1: wgw_body<WIDE, 1>(p, dsm, split); break;
    case 2: wgw_body<WIDE, 2>(p, dsm, split); break;
    default: wgw_body<WIDE, 3>(p, dsm, split); break;
  }
}

// dw[co][ci][a][b] = (G^T (sum over splits of dU) G)[a][b];  G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]
__global__ __launch_bounds__(256) void conv3x3_wgrad_wino_reduce_kernel(const float* __restrict__ slabs, const float* __restrict__ bias_part,
                                                                        int nsplit, int Cin, int Cout, float* __restrict__ dw, float* __restrict__ db) {
  // block = 64 (ci, co) pairs x 4 wavefronts; wavefront k sums the slabs k, k + 4, ... (sixteen independent 256-byte row reads
  // per slab), the four partial sums meet in LDS in a fixed order.  One thread per pair over all slabs left 64 blocks for a
  // 128 x 128 layer: a quarter of the CUs, 2.2 TB/s.
  __shared__ float part[4][16][64];
  const int pl = threadIdx.x & 63, k = threadIdx.x >> 6;
  const int idx = blockIdx.x * 64 + pl;
  const bool live = idx < Cin * Cout;
  const int co = live ? idx % Cout : 0, ci = live ? idx / Cout : 0;
  float u[16];
#pragma unroll
  for (int xi = 0; xi < 16; ++xi) u[xi] = 0.f;
  if (live)
    for (int s = k; s < nsplit; s += 4)
#pragma unroll
      for (int xi = 0; xi < 16; ++xi) u[xi] += slabs[((int64_t)(s * 16 + xi) * Cin + ci) * Cout + co];
#pragma unroll
  for (int xi = 0; xi < 16; ++xi) part[k][xi][pl] = u[xi];
  __syncthreads();
  if (k != 0) return;
  if (live) {
#pragma unroll
    for (int xi = 0; xi < 16; ++xi) u[xi] = (part[0][xi][pl] + part[1][xi][pl]) + (part[2][xi][pl] + part[3][xi][pl]);
    float t[3][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float hs = 0.5f * (u[4 + j] + u[8 + j]), hd = 0.5f * (u[4 + j] - u[8 + j]);
      t[0][j] = u[j] + hs; t[1][j] = hd; t[2][j] = hs + u[12 + j];
    }
    float* o = dw + ((int64_t)co * Cin + ci) * 9;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float hs = 0.5f * (t[a][1] + t[a][2]), hd = 0.5f * (t[a][1] - t[a][2]);
      o[a * 3 + 0] = t[a][0] + hs; o[a * 3 + 1] = hd; o[a * 3 + 2] = hs + t[a][3];
    }
  }
  if (db && idx < Cout) {
    float sb = 0.f;
    for (int q = 0; q < nsplit; ++q) sb += bias_part[(int64_t)q * Cout + idx];
    db[idx] = sb;
  }
}

void plan(int N, int H, int W, int Cin, int Cout, WgwParams& p) {
  p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.TY = H / 2; p.TX = W / 2;
  p.total_tiles = N * p.TY * p.TX;
  p.total_chunks = ceil_div(p.total_tiles, CT);
  static const int target = getenv("ODVAE_WGRAD_WINO_BLOCKS") ? atoi(getenv("ODVAE_WGRAD_WINO_BLOCKS")) : 256;
  const int base = 4 * (Cin / 128) * (Cout / 128);
  int ns = std::max(1, std::min(target / base, p.total_chunks));
  p.chunks_per_split = ceil_div(p.total_chunks, ns);
  p.nsplit = ceil_div(p.total_chunks, p.chunks_per_split);
}

}  // namespace

extern "C" {

// 1 when the Winograd-domain weight gradient serves this shape (otherwise use odvae_conv3x3_wgrad_f32)
int odvae_conv3x3_wgrad_wino_supported(int N, int H, int W, int Cin, int Cout) {
  if (N <= 0 || H <= 0 || W <= 0 || H % 2 || W % 2 || Cin <= 0 || Cout <= 0 || Cin % 128 || Cout % 128) return 0;
  const int64_t px = (int64_t)N * H * W;
  if (px * Cin * 4 >= (int64_t)OOB || px * Cout * 4 >= (int64_t)OOB || px / 4 >= (1ll << 30)) return 0;
  return 1;
}

size_t odvae_conv3x3_wgrad_wino_workspace_bytes(int N, int H, int W, int Cin, int Cout) {
  if (!odvae_conv3x3_wgrad_wino_supported(N, H, W, Cin, Cout)) return 0;
  WgwParams p; plan(N, H, W, Cin, Cout, p);
  return ((size_t)p.nsplit * 16 * Cin * Cout + (size_t)p.nsplit * Cout) * sizeof(float);
}

// x [N][H][W][Cin], dy [N][H][W][Cout] (NHWC f32) -> dw OIHW [Cout][Cin][3][3] (overwritten), dbias [Cout] or NULL.
// Replaces autograd's weight/bias gradient of F.conv2d(x, w, b, stride=1, padding=1).
int odvae_conv3x3_wgrad_wino_f32(const float* x, const float* dy, int N, int H, int W, int Cin, int Cout,
                                 float* dw, float* dbias, void* workspace, size_t workspace_bytes, void* stream) {
  ODVAE_CHECK_ARG(x && dy && dw, "conv3x3_wgrad_wino: null operand");
  ODVAE_CHECK_ARG(odvae_conv3x3_wgrad_wino_supported(N, H, W, Cin, Cout),
                  "conv3x3_wgrad_wino: unsupported shape N=%d H=%d W=%d Cin=%d Cout=%d (even H, W; channels in multiples of 128; "
                  "tensors below 4 GiB)", N, H, W, Cin, Cout);
  ODVAE_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)dy & 15) == 0, "conv3x3_wgrad_wino: x/dy must be 16-byte aligned");
  WgwParams p; plan(N, H, W, Cin, Cout, p);
  const size_t need = odvae_conv3x3_wgrad_wino_workspace_bytes(N, H, W, Cin, Cout);
  if (!workspace || workspace_bytes < need) {
    odvae_set_error("conv3x3_wgrad_wino: workspace %zu < %zu bytes", workspace_bytes, need);
    return ODVAE_ERR_WORKSPACE;
  }
  p.x = x; p.dy = dy;
  p.slabs = static_cast<float*>(workspace);
  p.bias_part = dbias ? p.slabs + (size_t)p.nsplit * 16 * Cin * Cout : nullptr;
  const size_t smem = (size_t)4 * OPF * sizeof(float);
  auto kern = p.TX >= CT ? conv3x3_wgrad_wino_kernel<true> : conv3x3_wgrad_wino_kernel<false>;
  const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  if (e != hipSuccess) {
    odvae_set_error("conv3x3_wgrad_wino: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
    return ODVAE_ERR_HIP;
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(kern, dim3(4 * p.nsplit, Cin / 128, Cout / 128), dim3(512), smem, s, p);
  ODVAE_LAUNCH_CHECK("conv3x3_wgrad_wino");
  const int pairs = Cin * Cout;
  hipLaunchKernelGGL(conv3x3_wgrad_wino_reduce_kernel, dim3(ceil_div(pairs, 64)), dim3(256), 0, s,
                     p.slabs, p.bias_part, p.nsplit, Cin, Cout, dw, dbias);
  ODVAE_LAUNCH_CHECK("conv3x3_wgrad_wino_reduce");
  return ODVAE_OK;
}

}  // extern "C"
