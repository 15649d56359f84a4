// First form of DESIGN.md §7's next kernel: the stride-1 3x3 weight gradient in the F(4x4,3x3) domain (the arithmetic: tools/wgrad_wino4_math.py).
// NOT part of the library (written in the last GPU minutes of round 4): a standalone program that
// checks itself against a CPU loop (ragged H, W; passes at 3.0e-6 of max|dW|) and times the 128 -> 128 @256^2 layer (profiles/r04_wgrad_wino4_probe.txt).
// 256 VGPRs, one spilled register in this form.
//   dW[co][ci] = Aw^T [ sum_tiles (Gw dy Gw^T) .* (B^T x B) ] Aw       per 4x4 output tile, 36 products instead of 144
// Blocking (the forward kernel's, as sized in DESIGN.md): a block owns all 36 xi x 64 co x 32 ci of the transform-domain sum (8 waves: wave =
// (xi group of 9, co half of 32), nine 32 x 32 accumulators = 144 registers) over a contiguous range of tiles, 8 tiles per chunk:
//   stages of 4 tiles; in LDS two stages of U [36][4][64] and two 8-tile stages of V [36][8][32] (147.5 KB); waves 0-3 transform dy (4x4 -> 6x6,
//   one (tile, co) pair per thread and stage), waves 4-7 x (6x6 -> 6x6, one (tile, ci) pair per thread every second stage), every wave
//   multiplies: 2 k-steps x 9 xi of v_mfma_f32_32x32x2_f32 per stage (k = tile).
//   History (B = 8, 128 -> 128 @256^2, 32 splits): one 8-tile stage, fetch-transform-multiply in sequence 0.933 ms; operands of the next chunk
//   fetched before the MFMAs 0.771 ms (the best so far: 200 TFLOP/s direct-equivalent, the rate of the library's F(2x2)-domain kernel); two 4-tile
//   stages with x on waves 4-5 only 0.921 ms (two SIMDs carry 37 % more vector work); this form 0.827 ms -- putting one wave of a SIMD in the
//   transforms while the other multiplies did NOT overlap the two (as tools/mfma_valu_coexec_probe.hip found for bf16), so what is left is to
//   shorten the vector work itself: the two integer divisions per fetch (tile -> n, ty, tx), one v_cndmask per load, and the finish kernel
//   (64 blocks reading 75 MB: ~0.1 ms of every number above; 64 / 128 splits cost +0.21 / +0.72 ms through it).
// Grid = (Cout / 64) x (Cin / 32) x splits; a second kernel sums the splits and applies Aw^T . Aw.
// Build: hipcc --offload-arch=gfx950 -O3 tools/wgrad_wino4_probe.hip -o tools/bin/wgrad_wino4 ; run on the GPU box: tools/bin/wgrad_wino4
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
// raw buffer descriptors: offsets >= the byte count load 0
__device__ __forceinline__ float bload(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, int soff) {      // soff: wave-uniform part of the offset (a scalar register)
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, soff, 0));
}
constexpr unsigned OOB = 0xFFFFFFF0u;      // the host checks that both operands are smaller than this

constexpr int CO_B = 64, CI_B = 32, TS = 4, TCH = 8, NXI = 36;      // TS tiles per LDS stage, two stages; a split is a multiple of TCH tiles
constexpr int TV = 2 * TS;      // x is staged 8 tiles at a time (one pair per thread of waves 4-7 every second stage)
constexpr int LDS_U = 2 * NXI * TS * CO_B, LDS_V = 2 * NXI * TV * CI_B;      // floats

struct Params {
  const float* x;       // [N][H][W][Cin]
  const float* dy;      // [N][H][W][Cout]
  float* partial;       // [splits][36][Cout][Cin]
  int N, H, W, Cin, Cout, TY, TX, tiles, tiles_per_split;
};

// B^T d for one line of six (the forward kernel's input transform)
__device__ __forceinline__ void bt6(const float d[6], float o[6]) {
  o[0] = 4.f * d[0] - 5.f * d[2] + d[4];
  o[1] = -4.f * (d[1] + d[2]) + d[3] + d[4];
  o[2] = 4.f * (d[1] - d[2]) - d[3] + d[4];
  o[3] = 2.f * (d[3] - d[1]) - d[2] + d[4];
  o[4] = 2.f * (d[1] - d[3]) - d[2] + d[4];
  o[5] = 4.f * d[1] - 5.f * d[3] + d[5];
}
// Gw g for one line of four: the points' Vandermonde (powers 0..3) under the forward G's row factors
__device__ __forceinline__ void gw6(const float g[4], float o[6]) {
  const float e = g[0] + g[2], od = g[1] + g[3];
  const float e4 = g[0] + 4.f * g[2], o4 = 2.f * g[1] + 8.f * g[3];
  o[0] = 0.25f * g[0];
  o[1] = (-1.f / 6.f) * (e + od);
  o[2] = (-1.f / 6.f) * (e - od);
  o[3] = (1.f / 24.f) * (e4 + o4);
  o[4] = (1.f / 24.f) * (e4 - o4);
  o[5] = g[3];
}

__global__ __launch_bounds__(512) void wgrad_wino4_kernel(Params p) {
  extern __shared__ float smem[];
  float* U = smem;                 // [xi][tile][co]
  float* V = smem + LDS_U;         // [xi][tile][ci]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int co0 = blockIdx.x * CO_B, ci0 = blockIdx.y * CI_B, split = blockIdx.z;
  const int g = wave & 3, hco = wave >> 2;
  const int li = lane & 31, hk = lane >> 5;

  f32x16 acc[9];
#pragma unroll
  for (int j = 0; j < 9; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  const int tbeg = split * p.tiles_per_split, tend = tbeg + p.tiles_per_split;      // tiles_per_split is a multiple of TCH
  const int per_img = p.TY * p.TX;
  // x: the descriptor starts one row and one pixel BEFORE the tensor, so that the offset of a patch's first (halo) element is never negative:
  // the hardware's range check looks at the vector offset alone, a wrapped one reads as out of range even when offset + soffset is inside
  // (the first run of this program: every tile of the first tile row came back zero).  Halo elements are never fetched (OOB offset).
  const unsigned xlead = (unsigned)((p.W + 1) * p.Cin * 4);
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(const_cast<float*>(p.x)) - xlead, 0,
                                                                      (int)(unsigned)((int64_t)p.N * p.H * p.W * p.Cin * 4 + xlead), 0x00020000);
  const __amdgpu_buffer_rsrc_t dyrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dy), 0, (int)(unsigned)((int64_t)p.N * p.H * p.W * p.Cout * 4), 0x00020000);

  // Stages of TS = 4 tiles, two of them in LDS.  Waves 0-3 transform dy (one (tile, co) pair per thread), waves 4-5 transform x (one
  // (tile, ci) pair per thread); in every iteration waves 0-3 transform stage s + 1 FIRST and multiply stage s AFTER, waves 4-7 the other
  // way round, so that each SIMD (waves w and w + 4) has one wave in vector work while the other is in the matrix pipe.  The operands of
  // stage s + 2 are fetched into registers right after stage s + 1 left them.
  const bool is_u = wave < 4;
  const int ut = wave & 3, uco = lane;                         // waves 0-3: tile of the 4-tile stage, co
  const int vt = ((tid >> 5) & 7), vci = tid & 31;             // waves 4-7: tile of the 8-tile x stage, ci
  float d[4][4], xv[6][6];
  auto fetch_u = [&](int t0) {
    const int t = t0 + ut;
    const bool live = t < p.tiles;
    const int tc = live ? t : 0;
    const int n = tc / per_img, r = tc - n * per_img, ty = r / p.TX, tx = r - ty * p.TX;
    const unsigned base = (unsigned)((((n * p.H + 4 * ty) * p.W + 4 * tx) * p.Cout + co0 + uco) * 4);
    unsigned colbase[4];      // per column: the tile's first row, or out of range; the row's part of the offset is wave-uniform (soffset)
#pragma unroll
    for (int b = 0; b < 4; ++b) colbase[b] = (live && 4 * tx + b < p.W) ? base + (unsigned)(b * p.Cout * 4) : OOB;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const bool rowok = 4 * ty + a < p.H;
#pragma unroll
      for (int b = 0; b < 4; ++b) d[a][b] = bload(dyrs, rowok ? colbase[b] : OOB, a * p.W * p.Cout * 4);
    }
  };
  auto fetch_v = [&](int t0) {      // the 6 x 6 patch with its 1-pixel zero halo
    const int t = t0 + vt;
    const bool live = t < p.tiles;
    const int tc = live ? t : 0;
    const int n = tc / per_img, r = tc - n * per_img, ty = r / p.TX, tx = r - ty * p.TX;
    const unsigned base = (unsigned)((((n * p.H + 4 * ty) * p.W + 4 * tx) * p.Cin + ci0 + vci) * 4);      // of element (-1, -1) of the patch, relative to the shifted descriptor
#pragma unroll
    for (int b = 0; b < 6; ++b) {
      const int xx = 4 * tx - 1 + b;
      const unsigned colbase = (live && xx >= 0 && xx < p.W) ? base + (unsigned)(b * p.Cin * 4) : OOB;
#pragma unroll
      for (int a = 0; a < 6; ++a) {
        const bool rowok = 4 * ty - 1 + a >= 0 && 4 * ty - 1 + a < p.H;
        xv[a][b] = bload(xrs, rowok ? colbase : OOB, a * p.W * p.Cin * 4);
      }
    }
  };
  auto transform_u = [&](int st) {      // U = Gw dy Gw^T
    float tt[6][4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const float in[4] = {d[0][b], d[1][b], d[2][b], d[3][b]};
      float o[6];
      gw6(in, o);
#pragma unroll
      for (int a = 0; a < 6; ++a) tt[a][b] = o[a];
    }
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      float o[6];
      gw6(tt[a], o);
#pragma unroll
      for (int b = 0; b < 6; ++b) U[(((st * NXI) + a * 6 + b) * TS + ut) * CO_B + uco] = o[b];
    }
  };
  auto transform_v = [&](int st) {      // V = B^T x B
    float tt[6][6];
#pragma unroll
    for (int b = 0; b < 6; ++b) {
      const float in[6] = {xv[0][b], xv[1][b], xv[2][b], xv[3][b], xv[4][b], xv[5][b]};
      float o[6];
      bt6(in, o);
#pragma unroll
      for (int a = 0; a < 6; ++a) tt[a][b] = o[a];
    }
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      float o[6];
      bt6(tt[a], o);
#pragma unroll
      for (int b = 0; b < 6; ++b) V[(((st * NXI) + a * 6 + b) * TV + vt) * CI_B + vci] = o[b];
    }
  };
  auto multiply = [&](int s) {      // stage s of the split: U parity s & 1; V parity (s >> 1) & 1, its tiles 4 (s & 1) ..
    const int su = s & 1, sv = (s >> 1) & 1, tv0 = TS * (s & 1);
#pragma unroll
    for (int ks = 0; ks < TS / 2; ++ks) {
#pragma unroll
      for (int j = 0; j < 9; ++j) {
        const int xi = 9 * g + j;
        const float a = U[((su * NXI + xi) * TS + 2 * ks + hk) * CO_B + 32 * hco + li];      // A: row = co, k = tile
        const float b = V[((sv * NXI + xi) * TV + tv0 + 2 * ks + hk) * CI_B + li];           // B: k = tile, column = ci
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
      }
    }
  };

  // Stages past the split's end are transformed into the LDS stage nobody multiplies (tiles past the tensor's end fetch zeros).
  if (is_u) { fetch_u(tbeg); transform_u(0); fetch_u(tbeg + TS); }
  else      { fetch_v(tbeg); transform_v(0); fetch_v(tbeg + TV); }
  __syncthreads();
  const int stages = p.tiles_per_split / TS;      // even (a split is a multiple of 8 tiles)
  for (int s = 0; s < stages; ++s) {
    const int t0 = tbeg + s * TS;
    if (is_u) {
      transform_u((s + 1) & 1);
      fetch_u(t0 + 2 * TS);
      multiply(s);
    } else {
      multiply(s);
      if (s & 1) { transform_v(((s >> 1) + 1) & 1); fetch_v(t0 - TS + 2 * TV); }
    }
    __syncthreads();
  }

#pragma unroll
  for (int j = 0; j < 9; ++j) {
    const int xi = 9 * g + j;
    float* out = p.partial + (((int64_t)split * NXI + xi) * p.Cout + co0 + 32 * hco) * p.Cin + ci0 + li;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = 8 * (r >> 2) + 4 * hk + (r & 3);
      out[(int64_t)row * p.Cin] = acc[j][r];
    }
  }
}

// dW[co][ci][3][3] = Aw^T (sum over splits of M[.][co][ci]) Aw,  Aw^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 1]
__global__ void wgrad_wino4_finish_kernel(const float* partial, int splits, int Cout, int Cin, float* dw) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= Cout * Cin) return;
  float m[6][6];
#pragma unroll
  for (int a = 0; a < 6; ++a)
#pragma unroll
    for (int b = 0; b < 6; ++b) m[a][b] = 0.f;
  for (int s = 0; s < splits; ++s)
#pragma unroll
    for (int xi = 0; xi < NXI; ++xi) m[xi / 6][xi % 6] += partial[((int64_t)s * NXI + xi) * Cout * Cin + idx];
  float t[3][6];
#pragma unroll
  for (int b = 0; b < 6; ++b) {
    t[0][b] = m[0][b] + m[1][b] + m[2][b] + m[3][b] + m[4][b];
    t[1][b] = m[1][b] - m[2][b] + 2.f * (m[3][b] - m[4][b]);
    t[2][b] = m[1][b] + m[2][b] + 4.f * (m[3][b] + m[4][b]) + m[5][b];
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    dw[(int64_t)idx * 9 + a * 3 + 0] = t[a][0] + t[a][1] + t[a][2] + t[a][3] + t[a][4];
    dw[(int64_t)idx * 9 + a * 3 + 1] = t[a][1] - t[a][2] + 2.f * (t[a][3] - t[a][4]);
    dw[(int64_t)idx * 9 + a * 3 + 2] = t[a][1] + t[a][2] + 4.f * (t[a][3] + t[a][4]) + t[a][5];
  }
}

#define HIP_OK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

struct Run { float ms; std::vector<float> dw; };

static Run run(int N, int H, int W, int Cin, int Cout, int splits_want, const std::vector<float>& x, const std::vector<float>& dy, int reps) {
  if (Cin % CI_B || Cout % CO_B) { fprintf(stderr, "Cin %% 32 and Cout %% 64 required\n"); exit(1); }
  if (x.size() * 4 + (size_t)(W + 1) * Cin * 4 >= OOB || dy.size() * 4 >= OOB) { fprintf(stderr, "operands must stay below 4 GB (one buffer descriptor each)\n"); exit(1); }
  Params p{};
  p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout;
  p.TY = (H + 3) / 4; p.TX = (W + 3) / 4; p.tiles = N * p.TY * p.TX;
  int splits = splits_want;
  while (splits > 1 && (p.tiles + splits - 1) / splits < TCH) splits /= 2;
  p.tiles_per_split = (((p.tiles + splits - 1) / splits) + TCH - 1) / TCH * TCH;
  splits = (p.tiles + p.tiles_per_split - 1) / p.tiles_per_split;
  float *dx, *ddy, *dpart, *ddw;
  HIP_OK(hipMalloc(&dx, x.size() * 4)); HIP_OK(hipMalloc(&ddy, dy.size() * 4));
  HIP_OK(hipMalloc(&dpart, (size_t)splits * NXI * Cout * Cin * 4)); HIP_OK(hipMalloc(&ddw, (size_t)Cout * Cin * 9 * 4));
  HIP_OK(hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice)); HIP_OK(hipMemcpy(ddy, dy.data(), dy.size() * 4, hipMemcpyHostToDevice));
  p.x = dx; p.dy = ddy; p.partial = dpart;
  const unsigned lds = (LDS_U + LDS_V) * 4;
  HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_wino4_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1; HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
  Run out; out.ms = 0.f;
  for (int it = 0; it < reps + 1; ++it) {
    if (it == 1) HIP_OK(hipEventRecord(e0));
    hipLaunchKernelGGL(wgrad_wino4_kernel, dim3(Cout / CO_B, Cin / CI_B, splits), dim3(512), lds, 0, p);
    hipLaunchKernelGGL(wgrad_wino4_finish_kernel, dim3((Cout * Cin + 255) / 256), dim3(256), 0, 0, dpart, splits, Cout, Cin, ddw);
  }
  HIP_OK(hipGetLastError());
  HIP_OK(hipEventRecord(e1)); HIP_OK(hipEventSynchronize(e1));
  if (reps > 0) { HIP_OK(hipEventElapsedTime(&out.ms, e0, e1)); out.ms /= reps; }
  out.dw.resize((size_t)Cout * Cin * 9);
  HIP_OK(hipMemcpy(out.dw.data(), ddw, out.dw.size() * 4, hipMemcpyDeviceToHost));
  HIP_OK(hipFree(dx)); HIP_OK(hipFree(ddy)); HIP_OK(hipFree(dpart)); HIP_OK(hipFree(ddw));
  printf("  grid %d x %d x %d, %d tiles per split, LDS %u B\n", Cout / CO_B, Cin / CI_B, splits, p.tiles_per_split, lds);
  return out;
}

static void fill(std::vector<float>& v, unsigned seed, float scale) {
  unsigned s = seed * 2654435761u + 12345u;
  for (auto& f : v) { s = s * 1664525u + 1013904223u; f = scale * ((float)((s >> 8) & 0xFFFF) / 32768.f - 1.f); }
}

int main() {
  {      // check against a CPU loop (ragged H, W: the last tile row / column is partly outside)
    const int N = 2, H = 18, W = 27, Cin = 64, Cout = 128;
    std::vector<float> x((size_t)N * H * W * Cin), dy((size_t)N * H * W * Cout);
    fill(x, 1, 1.f); fill(dy, 2, 1.f);
    printf("check N=%d %dx%d %d->%d\n", N, H, W, Cin, Cout);
    Run r = run(N, H, W, Cin, Cout, 4, x, dy, 0);
    double worst = 0., big = 0.;
    for (int co = 0; co < Cout; ++co) for (int ci = 0; ci < Cin; ++ci) for (int ky = 0; ky < 3; ++ky) for (int kx = 0; kx < 3; ++kx) {
      double s = 0.;
      for (int n = 0; n < N; ++n) for (int y = 0; y < H; ++y) {
        const int yy = y + ky - 1; if (yy < 0 || yy >= H) continue;
        for (int xx = 0; xx < W; ++xx) {
          const int xs = xx + kx - 1; if (xs < 0 || xs >= W) continue;
          s += (double)dy[(((size_t)n * H + y) * W + xx) * Cout + co] * x[(((size_t)n * H + yy) * W + xs) * Cin + ci];
        }
      }
      big = fmax(big, fabs(s));
      worst = fmax(worst, fabs(s - r.dw[((size_t)co * Cin + ci) * 9 + ky * 3 + kx]));
    }
    printf("  worst |dev| / max|dW| = %.3e (%s)\n", worst / big, worst / big < 2e-5 ? "ok" : "MISMATCH");
    if (!(worst / big < 2e-5)) return 1;
  }
  {      // the step's largest layer: 128 -> 128 @256^2, B = 8 (1 GB of operands at B = 32 does not change the rate)
    const int N = 8, H = 256, W = 256, Cin = 128, Cout = 128;
    std::vector<float> x((size_t)N * H * W * Cin), dy((size_t)N * H * W * Cout);
    fill(x, 3, 1.f); fill(dy, 4, 1e-3f);
    for (int splits : {32, 64, 128}) {
      Run r = run(N, H, W, Cin, Cout, splits, x, dy, 10);
      const double flop = 2.0 * 9 * N * H * W * (double)Cin * Cout;
      printf("  splits %d: %.3f ms, %.1f TFLOP/s direct-equivalent (%.1f issued)\n", splits, r.ms, flop / r.ms * 1e-9, flop / 4 / r.ms * 1e-9);
    }
  }
  return 0;
}
