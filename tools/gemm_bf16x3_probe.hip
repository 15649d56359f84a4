// PROBE (not part of the library): an f32 product C = A B^T carried on the bf16 MFMA by splitting each f32 operand EXACTLY into three bf16
// planes (x = hi + mid + lo, 8 + 8 + 8 significand bits) and issuing six of the nine cross products (hi.hi, hi.mid, mid.hi, hi.lo, lo.hi,
// mid.mid; the three dropped ones are below 2^-24 of |a||b|) with f32 accumulation.  DESIGN.md 7 ("what is next", item 4) names this as the one idea
// with real headroom for the f32 step's attention products (66 of 224 ms at 0.80 of the f32 MFMA peak) and says why it is NOT in the product:
// its results are no longer the f32 MFMA's exact products.  This program measures what it would buy and what it costs in accuracy, so that the
// decision can be made on numbers:  error against an f64 product on a small case, then time at the attention shapes of the headline step
// (batch 32, T = 4096 tokens, C = 256: S = Q K^T is M = N = 4096, K = 256; the T-deep products are M = 4096, N = 256, K = 4096).
//   operands arrive PRE-SPLIT ([3 planes][rows][K] bf16: in a real integration the producing kernel's epilogue would write them), the split
//   kernel here is timed separately.
// Block = 128 x 128 outputs, 4 waves of 64 x 64 (2 x 2 tiles of v_mfma_f32_32x32x16_bf16), K in steps of 32 through a double-buffered LDS stage
// ([plane][row][32 + 8 pad] bf16: conflict-free ds_read_b128), operands staged through registers.
// Build: hipcc --offload-arch=gfx950 -O3 tools/gemm_bf16x3_probe.hip -o tools/bin/gemm_bf16x3 ; run on the GPU box: tools/bin/gemm_bf16x3
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef unsigned short bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define HIP_OK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ unsigned pack2(float lo, float hi) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}
__device__ __forceinline__ float bf16_f32(unsigned short v) { return __uint_as_float((unsigned)v << 16); }

// x [rows][K] f32 -> planes [3][rows][K] bf16: hi = rn(x), mid = rn(x - hi), lo = rn(x - hi - mid); the differences are exact in f32
__global__ void split3_kernel(const float* __restrict__ x, bf16_t* __restrict__ planes, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n / 2; i += (long)gridDim.x * blockDim.x) {
    const float2 v = reinterpret_cast<const float2*>(x)[i];
    const unsigned h = pack2(v.x, v.y);
    const float r1x = v.x - bf16_f32((unsigned short)(h & 0xFFFF)), r1y = v.y - bf16_f32((unsigned short)(h >> 16));
    const unsigned m = pack2(r1x, r1y);
    const float r2x = r1x - bf16_f32((unsigned short)(m & 0xFFFF)), r2y = r1y - bf16_f32((unsigned short)(m >> 16));
    const unsigned l = pack2(r2x, r2y);
    reinterpret_cast<unsigned*>(planes)[i] = h;
    reinterpret_cast<unsigned*>(planes + n)[i] = m;
    reinterpret_cast<unsigned*>(planes + 2 * n)[i] = l;
  }
}

#ifndef BKV
#define BKV 32
#endif
constexpr int BM = 128, BN = 128, BK = BKV, LDK = BK + 8;      // LDS row stride in bf16 (80 bytes at BK = 32, 48 at BK = 16: both conflict-free)
constexpr int CPR = BK / 8;                                    // 16-byte chunks per row of a plane tile
constexpr int CPT = BM * CPR / 256;                            // chunks per thread and plane (2 at BK = 32, 1 at BK = 16)
constexpr int PLANE_LDS = BM * LDK;                            // bf16 per plane tile
constexpr int STAGE_LDS = 6 * PLANE_LDS;                       // A: 3 planes, B: 3 planes

struct Params {
  const bf16_t* a;      // [batch][3][M][K]
  const bf16_t* b;      // [batch][3][N][K]
  float* c;             // [batch][M][N]
  int M, N, K;
  int terms;            // 6 (the probe's form), 3 (hi.hi, hi.mid, mid.hi) or 1 (plain bf16) for comparison
};

__global__ __launch_bounds__(256, BKV == 16 ? 2 : 1) void gemm_bf16x3_kernel(Params p) {
  extern __shared__ __attribute__((aligned(16))) bf16_t lds[];      // 2 stages
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;                          // wave's 64 x 64 quadrant
  const int r = lane & 31, h = lane >> 5;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN, z = blockIdx.z;
  const long MK = (long)p.M * p.K, NK = (long)p.N * p.K;
  const bf16_t* A = p.a + (long)z * 3 * MK;
  const bf16_t* B = p.b + (long)z * 3 * NK;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // staging: 6 planes x 128 rows x 32 k = 6 x 512 chunks of 16 bytes; thread t carries chunks t, t + 256 of every plane
  u32x4 st[6 * CPT];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int pl = 0; pl < 6; ++pl)
#pragma unroll
      for (int c = 0; c < CPT; ++c) {
        const int ch = tid + 256 * c, row = ch / CPR, kq = ch % CPR;
        const bf16_t* src = pl < 3 ? A + pl * MK + (long)(m0 + row) * p.K : B + (pl - 3) * NK + (long)(n0 + row) * p.K;
        st[pl * CPT + c] = *reinterpret_cast<const u32x4*>(src + k0 + 8 * kq);
      }
  };
  auto stash = [&](int stage) {
#pragma unroll
    for (int pl = 0; pl < 6; ++pl)
#pragma unroll
      for (int c = 0; c < CPT; ++c) {
        const int ch = tid + 256 * c, row = ch / CPR, kq = ch % CPR;
        *reinterpret_cast<u32x4*>(lds + stage * STAGE_LDS + pl * PLANE_LDS + row * LDK + 8 * kq) = st[pl * CPT + c];
      }
  };

  fetch(0);
  stash(0);
  __syncthreads();
  const int nk = p.K / BK;
  for (int ks = 0; ks < nk; ++ks) {
    const bf16_t* S = lds + (ks & 1) * STAGE_LDS;
    if (ks + 1 < nk) fetch((ks + 1) * BK);
#pragma unroll
    for (int sub = 0; sub < BK / 16; ++sub) {
      bf16x8 af[3][2], bfr[3][2];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          af[pl][t] = *reinterpret_cast<const bf16x8*>(S + pl * PLANE_LDS + (wm * 64 + t * 32 + r) * LDK + 16 * sub + 8 * h);
          bfr[pl][t] = *reinterpret_cast<const bf16x8*>(S + (3 + pl) * PLANE_LDS + (wn * 64 + t * 32 + r) * LDK + 16 * sub + 8 * h);
        }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          // smallest terms first
          if (p.terms >= 6) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2][i], bfr[0][j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bfr[2][j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][i], bfr[1][j], acc[i][j], 0, 0, 0);
          }
          if (p.terms >= 3) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][i], bfr[0][j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bfr[1][j], acc[i][j], 0, 0, 0);
          }
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bfr[0][j], acc[i][j], 0, 0, 0);
        }
    }
    if (ks + 1 < nk) stash((ks + 1) & 1);
    __syncthreads();
  }
  // D register e of lane (r, h): row (e & 3) + 8 (e >> 2) + 4 h, column r
  float* C = p.c + (long)z * p.M * p.N;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h, col = n0 + wn * 64 + j * 32 + r;
        C[(long)row * p.N + col] = acc[i][j][e];
      }
}

static void fill(std::vector<float>& v, unsigned seed, float scale) {
  unsigned s = seed * 2654435761u + 12345u;
  for (auto& f : v) { s = s * 1664525u + 1013904223u; f = scale * ((float)((s >> 8) & 0xFFFF) / 32768.f - 1.f) * (1.f + (float)((s >> 24) & 7)); }
}

static float time_gemm(Params p, int batch, int reps) {
  const unsigned ldsb = 2 * STAGE_LDS * sizeof(bf16_t);
  HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16x3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
  hipEvent_t e0, e1;
  HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
  for (int it = 0; it < reps + 3; ++it) {
    if (it == 3) HIP_OK(hipEventRecord(e0));
    hipLaunchKernelGGL(gemm_bf16x3_kernel, dim3(p.N / BN, p.M / BM, batch), dim3(256), ldsb, 0, p);
  }
  HIP_OK(hipGetLastError());
  HIP_OK(hipEventRecord(e1)); HIP_OK(hipEventSynchronize(e1));
  float ms = 0.f;
  HIP_OK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

int main() {
  {      // accuracy: one 128 x 128 x 256 product against f64, operands with a 1..8 x spread of magnitudes
    const int M = 128, N = 128, K = 256;
    std::vector<float> a((size_t)M * K), b((size_t)N * K);
    fill(a, 1, 1.f); fill(b, 2, 1.f);
    float *da, *db, *dc; bf16_t *pa, *pb;
    HIP_OK(hipMalloc(&da, a.size() * 4)); HIP_OK(hipMalloc(&db, b.size() * 4)); HIP_OK(hipMalloc(&dc, (size_t)M * N * 4));
    HIP_OK(hipMalloc(&pa, a.size() * 6)); HIP_OK(hipMalloc(&pb, b.size() * 6));
    HIP_OK(hipMemcpy(da, a.data(), a.size() * 4, hipMemcpyHostToDevice)); HIP_OK(hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(split3_kernel, dim3(64), dim3(256), 0, 0, da, pa, (long)a.size());
    hipLaunchKernelGGL(split3_kernel, dim3(64), dim3(256), 0, 0, db, pb, (long)b.size());
    std::vector<double> ref((size_t)M * N), mag((size_t)M * N);
    for (int i = 0; i < M; ++i) for (int j = 0; j < N; ++j) {
      double s = 0., t = 0.;
      for (int k = 0; k < K; ++k) { const double q = (double)a[(size_t)i * K + k] * b[(size_t)j * K + k]; s += q; t += fabs(q); }
      ref[(size_t)i * N + j] = s; mag[(size_t)i * N + j] = t;
    }
    // plain f32 accumulation in the order of k, for scale (what an exact-product, f32-accumulate unit does at best)
    double f32_worst = 0.;
    for (int i = 0; i < M; ++i) for (int j = 0; j < N; ++j) {
      float s = 0.f;
      for (int k = 0; k < K; ++k) s = fmaf(a[(size_t)i * K + k], b[(size_t)j * K + k], s);
      f32_worst = fmax(f32_worst, fabs((double)s - ref[(size_t)i * N + j]) / mag[(size_t)i * N + j]);
    }
    printf("accuracy, 128 x 128 x 256, error relative to sum |a_k b_k| (the natural scale of a dot product):\n  f32 fma chain (host)        %.3e\n", f32_worst);
    for (int terms : {1, 3, 6}) {
      Params p{pa, pb, dc, M, N, K, terms};
      const unsigned ldsb = 2 * STAGE_LDS * sizeof(bf16_t);
      HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16x3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
      hipLaunchKernelGGL(gemm_bf16x3_kernel, dim3(N / BN, M / BM, 1), dim3(256), ldsb, 0, p);
      std::vector<float> c((size_t)M * N);
      HIP_OK(hipMemcpy(c.data(), dc, c.size() * 4, hipMemcpyDeviceToHost));
      double worst = 0.;
      for (size_t i = 0; i < c.size(); ++i) worst = fmax(worst, fabs((double)c[i] - ref[i]) / mag[i]);
      printf("  bf16 MFMA, %d product(s)      %.3e\n", terms, worst);
    }
    HIP_OK(hipFree(da)); HIP_OK(hipFree(db)); HIP_OK(hipFree(dc)); HIP_OK(hipFree(pa)); HIP_OK(hipFree(pb));
  }
  const int batch = 32;
  struct Shape { int M, N, K; const char* what; };
  for (const Shape sh : {Shape{4096, 4096, 256, "S = Q K^T (T x T scores, K = C)"}, Shape{4096, 256, 4096, "O = P V / dQ = dS K (K = T)"}}) {
    const size_t an = (size_t)batch * sh.M * sh.K, bn = (size_t)batch * sh.N * sh.K;
    float *dx, *dc; bf16_t *pa, *pb;
    HIP_OK(hipMalloc(&dx, (an > bn ? an : bn) * 4)); HIP_OK(hipMalloc(&dc, (size_t)batch * sh.M * sh.N * 4));
    HIP_OK(hipMalloc(&pa, an * 6)); HIP_OK(hipMalloc(&pb, bn * 6));
    HIP_OK(hipMemset(dx, 0x3c, (an > bn ? an : bn) * 4));      // (some non-trivial bit pattern; timing only)
    hipEvent_t e0, e1; HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
    HIP_OK(hipEventRecord(e0));
    hipLaunchKernelGGL(split3_kernel, dim3(2048), dim3(256), 0, 0, dx, pa, (long)an);
    hipLaunchKernelGGL(split3_kernel, dim3(2048), dim3(256), 0, 0, dx, pb, (long)bn);
    HIP_OK(hipEventRecord(e1)); HIP_OK(hipEventSynchronize(e1));
    float split_ms = 0.f; HIP_OK(hipEventElapsedTime(&split_ms, e0, e1));
    // random-ish bf16 planes for the timed runs (zeros would let the chip clock higher than real data does)
    std::vector<bf16_t> rnd(1 << 20);
    unsigned s = 7; for (auto& v : rnd) { s = s * 1664525u + 1013904223u; v = (bf16_t)(0x3c00 + ((s >> 9) & 0x3ff) + ((s >> 31) << 15)); }
    for (size_t off = 0; off < an * 3; off += rnd.size()) HIP_OK(hipMemcpy(pa + off, rnd.data(), (off + rnd.size() <= an * 3 ? rnd.size() : an * 3 - off) * 2, hipMemcpyHostToDevice));
    for (size_t off = 0; off < bn * 3; off += rnd.size()) HIP_OK(hipMemcpy(pb + off, rnd.data(), (off + rnd.size() <= bn * 3 ? rnd.size() : bn * 3 - off) * 2, hipMemcpyHostToDevice));
    const double flop = 2.0 * batch * sh.M * (double)sh.N * sh.K;
    printf("batch %d, M %d N %d K %d -- %s; K step %d (%d block(s) per CU); splitting both operands: %.3f ms\n", batch, sh.M, sh.N, sh.K, sh.what, BK, BK == 16 ? 2 : 1, split_ms);
    for (int terms : {1, 3, 6}) {
      Params p{pa, pb, dc, sh.M, sh.N, sh.K, terms};
      const float ms = time_gemm(p, batch, 10);
      printf("  %d product(s): %.3f ms = %.1f TFLOP/s of f32-equivalent work (%.0f issued on the bf16 pipe); the f32 MFMA peak is 157.3, the library's f32 GEMM runs these at ~125\n",
             terms, ms, flop / ms * 1e-9, terms * flop / ms * 1e-9);
    }
    HIP_OK(hipFree(dx)); HIP_OK(hipFree(dc)); HIP_OK(hipFree(pa)); HIP_OK(hipFree(pb));
  }
  return 0;
}
