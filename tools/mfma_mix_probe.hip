// Micro-benchmark: how much v_mfma_f32_32x32x2_f32 throughput survives when other instruction classes share the SIMD
// (VALU work, LDS reads, L2-resident global loads), at one and two waves per SIMD; the same for v_mfma_f32_16x16x4_f32.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_mix_probe.hip -o tools/bin/mfma_mix_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// MODE 0: pure 32x32x2, 8 accumulators, 32 MFMAs per iteration
// MODE 1: + NV independent v_fma per MFMA (register-only vector work)
// MODE 2: + 8 ds_read_b128 per 32 MFMAs, consumed one iteration later
// MODE 3: + 8 global_load_dwordx4 (1 MB L2-resident window) per 32 MFMAs, consumed one iteration later
// MODE 4: 2 + 3
// MODE 5: pure 16x16x4, 32 accumulators (same 128 accumulator registers), 64 MFMAs per iteration (same flops as 32 of 32x32x2)
// MODE 6: 16x16x4 + NV v_fma per 2 MFMAs
template <int MODE, int NV>
__global__ __launch_bounds__(512) void probe(const float* __restrict__ g, float* out, int iters) {
  __shared__ __attribute__((aligned(16))) float sm[16384];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) sm[i] = g[i & 4095];
  __syncthreads();
  float4 a[8], b[8];
  for (int j = 0; j < 8; ++j) { a[j] = *reinterpret_cast<const float4*>(sm + ((lane + 64 * j) & 4095) * 4); b[j] = a[j]; }
  float v[8];
  for (int j = 0; j < 8; ++j) v[j] = (float)(lane + j);
  const float4* gq = reinterpret_cast<const float4*>(g) + lane + 64 * wave;
  if constexpr (MODE <= 4) {
    f32x16 acc[8];
    for (int t = 0; t < 8; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
      float4 na[8], nb[8];
      if (MODE == 2 || MODE == 4)
#pragma unroll
        for (int j = 0; j < 8; ++j) na[j] = *reinterpret_cast<const float4*>(sm + (((it + j * 5) * 64 + lane) & 4095) * 4);
      if (MODE == 3 || MODE == 4)
#pragma unroll
        for (int j = 0; j < 8; ++j) nb[j] = gq[((it * 8 + j) & 63) * 1024];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j].x, b[j].x, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j].y, b[j].y, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j].z, b[j].z, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j].w, b[j].w, acc[j], 0, 0, 0);
        if (MODE == 1)
#pragma unroll
          for (int q = 0; q < 4 * NV; ++q) v[(q + j) & 7] = __builtin_fmaf(v[(q + j) & 7], 1.0001f, 0.5f);
      }
      if (MODE == 2 || MODE == 4)
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = na[j];
      if (MODE == 3 || MODE == 4)
#pragma unroll
        for (int j = 0; j < 8; ++j) b[j] = nb[j];
    }
    float s = 0.f;
    for (int t = 0; t < 8; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    for (int j = 0; j < 8; ++j) s += v[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  } else {
    f32x4 acc[32];
    for (int t = 0; t < 32; ++t) for (int r = 0; r < 4; ++r) acc[t][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 32; ++j) {
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j & 7].x, b[j >> 2].x, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j & 7].y, b[j >> 2].y, acc[j], 0, 0, 0);
        if (MODE == 6)
#pragma unroll
          for (int q = 0; q < NV; ++q) v[(q + j) & 7] = __builtin_fmaf(v[(q + j) & 7], 1.0001f, 0.5f);
      }
    }
    float s = 0.f;
    for (int t = 0; t < 32; ++t) for (int r = 0; r < 4; ++r) s += acc[t][r];
    for (int j = 0; j < 8; ++j) s += v[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  }
}

template <int MODE, int NV>
void run(const char* name, int threads, const float* g, float* out) {
  const int iters = 4000, blocks = 256;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((probe<MODE, NV>), dim3(blocks), dim3(threads), 0, 0, g, out, 200);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((probe<MODE, NV>), dim3(blocks), dim3(threads), 0, 0, g, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  const double flop = (double)blocks * (threads / 64) * iters * 32 * 4096.0;
  printf("%-64s %d wave/SIMD: %8.3f ms  %6.1f TFLOP/s (%.3f of 157.3)\n", name, threads / 256, ms, flop / ms / 1e9, flop / ms / 1e9 / 157.3);
}

int main() {
  float *g, *out;
  hipMalloc(&g, 1 << 22); hipMalloc(&out, 256 * 512 * 4);
  std::vector<float> h(1 << 20);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) / 1000.f - 0.5f;
  hipMemcpy(g, h.data(), 1 << 22, hipMemcpyHostToDevice);
  for (int threads : {256, 512}) {
    run<0, 0>("32x32x2 pure", threads, g, out);
    run<1, 1>("32x32x2 + 1 v_fma per MFMA", threads, g, out);
    run<1, 2>("32x32x2 + 2 v_fma per MFMA", threads, g, out);
    run<1, 4>("32x32x2 + 4 v_fma per MFMA", threads, g, out);
    run<1, 8>("32x32x2 + 8 v_fma per MFMA", threads, g, out);
    run<2, 0>("32x32x2 + 8 ds_read_b128 per 32 MFMA", threads, g, out);
    run<3, 0>("32x32x2 + 8 global_load_dwordx4 (L2) per 32 MFMA", threads, g, out);
    run<4, 0>("32x32x2 + 8 ds_read_b128 + 8 global_load_dwordx4 per 32", threads, g, out);
    run<5, 0>("16x16x4 pure (same flops per iteration)", threads, g, out);
    run<6, 2>("16x16x4 + 2 v_fma per 2 MFMA (= 1 per 32x32x2-equivalent... x2)", threads, g, out);
    run<6, 8>("16x16x4 + 8 v_fma per 2 MFMA", threads, g, out);
  }
  return 0;
}
