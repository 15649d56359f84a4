// The stride-1 3x3 weight gradient in the F(4x4,3x3) domain (the arithmetic: tools/wgrad_wino4_math.py) as a standalone, self-checking
// program -- a NEGATIVE RESULT, kept as the record of it (rounds 4-5); not part of the library.
//   dW[co][ci] = Aw^T [ sum_tiles (Gw dy Gw^T) .* (B^T x B) ] Aw       per 4x4 output tile, 36 products instead of 144
// Blocking (the forward kernel's): a block owns all 36 xi x 64 co x 32 ci of the transform-domain sum (8 waves: wave = (xi group of 9,
// co half of 32), nine 32 x 32 accumulators = 144 registers) over a contiguous range of tiles, 8 tiles per chunk: every thread transforms one
// (tile, co) pair of dy (4x4 -> 6x6) and threads 0..255 one (tile, ci) pair of x (6x6 -> 6x6) per chunk, U [36][8][64] and V [36][8][32] go
// to LDS (110.6 KB), then 4 k-steps x 9 xi of v_mfma_f32_32x32x2_f32 per wave (k = tile); the next chunk's operands are fetched into
// registers before the MFMAs.  Tile decode by a float reciprocal with a one-step correction; the sum over splits by its own kernel.
// Grid = (Cout / 64) x (Cin / 32) x splits; two more kernels sum the splits and apply Aw^T . Aw.
// Measured on one MI355X at B = 32 against the library's F(2x2)-domain kernel (odvae_conv3x3_wgrad_wino_f32), profiles/r05_wgrad_wino4_probe.txt:
//   128->128 @256^2  2.39 ms vs 2.14      128->128 @128^2  0.68 vs 0.57      256->256 @64^2  0.72 vs 0.55      512->512 @32^2 0.71 (library @16^2: 0.16)
// i.e. 10-30 % SLOWER on every layer shape of the step although it issues 1.78x fewer MFMAs: each transformed value is reused by only 32 (ci)
// or 64 (co) accumulator columns and BOTH operands are transformed per chunk (the forward kernel transforms one, its weights arrive
// pre-transformed), ~900 vector instructions per SIMD and chunk beside 72 MFMAs -- and on gfx950 the f32 MFMA and the vector ALU share
// their issue cycles (profiles/r02_wino8_loop.md), so the transform time ADDS to the matrix time.  Forms tried: fetch -> transform ->
// multiply in sequence 0.933 ms (B = 8), next chunk's operands fetched before the MFMAs 0.771-0.780, two 4-tile LDS stages with the two waves
// of a SIMD in opposite phases 0.827-0.921 (no overlap of matrix and vector work), reciprocal tile decode + parallel split sum 0.674.
// Build: hipcc --offload-arch=gfx950 -O3 tools/wgrad_wino4_probe.hip -o tools/bin/wgrad_wino4 ; run: tools/bin/wgrad_wino4 [N H W Cin Cout splits...]
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

constexpr int CO_B = 64, CI_B = 32, TCH = 8, NXI = 36;
constexpr int LDS_U = NXI * TCH * CO_B, LDS_V = NXI * TCH * CI_B;      // floats

struct Params {
  const float* x;       // [N][H][W][Cin]
  const float* dy;      // [N][H][W][Cout]
  float* partial;       // [splits][36][Cout][Cin]
  int N, H, W, Cin, Cout, TY, TX, tiles, tiles_per_split;
  float inv_per_img, inv_tx;      // reciprocals for the tile decode (tiles < 2^24, host-checked)
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

// tile index -> image, tile row, tile column
__device__ __forceinline__ void decode_tile(const Params& p, int t, int& n, int& ty, int& tx) {
  const int per_img = p.TY * p.TX;
  // t < 2^24 is exact in a float and the rounded product is off by at most one: one correction step each way
  int q = (int)((float)t * p.inv_per_img);
  int r = t - q * per_img;
  if (r < 0) { --q; r += per_img; }
  if (r >= per_img) { ++q; r -= per_img; }
  int q2 = (int)((float)r * p.inv_tx);
  int r2 = r - q2 * p.TX;
  if (r2 < 0) { --q2; r2 += p.TX; }
  if (r2 >= p.TX) { ++q2; r2 -= p.TX; }
  n = q; ty = q2; tx = r2;
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
  const int ut = tid >> 6, uco = tid & 63;          // this thread's (tile, co) of the dy transform
  const int vt = (tid >> 5) & 7, vci = tid & 31;    // and (tile, ci) of the x transform (threads 0..255)
  // x: the descriptor starts one row and one pixel BEFORE the tensor, so that the offset of a patch's first (halo) element is never negative:
  // the hardware's range check looks at the vector offset alone, a wrapped one reads as out of range even when offset + soffset is inside
  // (the first run of this program: every tile of the first tile row came back zero).  Halo elements are never fetched (OOB offset).
  const unsigned xlead = (unsigned)((p.W + 1) * p.Cin * 4);
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(const_cast<float*>(p.x)) - xlead, 0,
                                                                      (int)(unsigned)((int64_t)p.N * p.H * p.W * p.Cin * 4 + xlead), 0x00020000);
  const __amdgpu_buffer_rsrc_t dyrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dy), 0, (int)(unsigned)((int64_t)p.N * p.H * p.W * p.Cout * 4), 0x00020000);

  // The operands of chunk c + 1 are fetched into registers before the MFMAs of chunk c and transformed after them (one LDS stage: the
  // fetch latency hides behind the matrix work, the transforms do not).
  float d[4][4], xv[6][6];
  auto fetch = [&](int t0) {
    {
      const int t = t0 + ut;
      const bool live = t < p.tiles;
      const int tc = live ? t : 0;
      int n, ty, tx;
      decode_tile(p, tc, n, ty, tx);
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
    }
    if (tid < 256) {      // the 6 x 6 patch with its 1-pixel zero halo
      const int t = t0 + vt;
      const bool live = t < p.tiles;
      const int tc = live ? t : 0;
      int n, ty, tx;
      decode_tile(p, tc, n, ty, tx);
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
    }
  };
  auto transform = [&]() {
    {      // U = Gw dy Gw^T
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
        for (int b = 0; b < 6; ++b) U[((a * 6 + b) * TCH + ut) * CO_B + uco] = o[b];
      }
    }
    if (tid < 256) {      // V = B^T x B
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
        for (int b = 0; b < 6; ++b) V[((a * 6 + b) * TCH + vt) * CI_B + vci] = o[b];
      }
    }
  };

  fetch(tbeg);
  for (int t0 = tbeg; t0 < tend; t0 += TCH) {
    transform();
    __syncthreads();
    if (t0 + TCH < tend) fetch(t0 + TCH);
#pragma unroll 2
    for (int ks = 0; ks < TCH / 2; ++ks) {
#pragma unroll
      for (int j = 0; j < 9; ++j) {
        const int xi = 9 * g + j;
        const float a = U[(xi * TCH + 2 * ks + hk) * CO_B + 32 * hco + li];      // A: row = co, k = tile
        const float b = V[(xi * TCH + 2 * ks + hk) * CI_B + li];                 // B: k = tile, column = ci
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
      }
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

// the sum over splits on its own, one thread per (xi, co, ci), into split 0's slab (every thread reads and writes its own element only)
__global__ void wgrad_wino4_sum_kernel(float* partial, int splits, int64_t n36) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n36) return;
  float sk[4] = {0.f, 0.f, 0.f, 0.f};
  int s = 0;
  for (; s + 3 < splits; s += 4)
#pragma unroll
    for (int j = 0; j < 4; ++j) sk[j] += partial[(int64_t)(s + j) * n36 + idx];
  for (; s < splits; ++s) sk[0] += partial[(int64_t)s * n36 + idx];
  partial[idx] = (sk[0] + sk[1]) + (sk[2] + sk[3]);
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
  p.inv_per_img = 1.0f / (float)(p.TY * p.TX); p.inv_tx = 1.0f / (float)p.TX;
  if (p.tiles >= (1 << 24) - 64) { fprintf(stderr, "tile count %d too large for the float decode\n", p.tiles); exit(1); }
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
    const int64_t n36 = (int64_t)NXI * Cout * Cin;
    hipLaunchKernelGGL(wgrad_wino4_sum_kernel, dim3((unsigned)((n36 + 255) / 256)), dim3(256), 0, 0, dpart, splits, n36);
    hipLaunchKernelGGL(wgrad_wino4_finish_kernel, dim3((Cout * Cin + 255) / 256), dim3(256), 0, 0, dpart, 1, Cout, Cin, ddw);
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

int main(int argc, char** argv) {
  if (argc >= 6) {      // timing of one layer shape: N H W Cin Cout [splits...]; device-side fill is not needed: the host vectors are a few GB at most
    const int N = atoi(argv[1]), H = atoi(argv[2]), W = atoi(argv[3]), Cin = atoi(argv[4]), Cout = atoi(argv[5]);
    std::vector<float> x((size_t)N * H * W * Cin), dy((size_t)N * H * W * Cout);
    fill(x, 3, 1.f); fill(dy, 4, 1e-3f);
    for (int a = 6; a < argc || a == 6; ++a) {
      const int splits = a < argc ? atoi(argv[a]) : 64;
      Run r = run(N, H, W, Cin, Cout, splits, x, dy, 20);
      const double flop = 2.0 * 9 * N * H * W * (double)Cin * Cout;
      printf("B%d %d->%d @%dx%d splits %d: %.3f ms, %.1f TFLOP/s direct-equivalent (%.1f issued)\n", N, Cin, Cout, H, W, splits, r.ms, flop / r.ms * 1e-9, flop / 4 / r.ms * 1e-9);
      if (a >= argc) break;
    }
    return 0;
  }
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
