// Micro-benchmark: what limits v_mfma_f32_32x32x2_f32 throughput when LDS / global loads share the instruction
// stream?  Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_probe.hip -o /tmp/mfma_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int VARIANT>
__global__ __launch_bounds__(512) void probe(const float* __restrict__ g, float* out, int iters) {
  __shared__ __attribute__((aligned(16))) float sm[16384];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) sm[i] = g[i & 4095];
  __syncthreads();
  f32x16 acc[9];
  for (int t = 0; t < 9; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float4 a0 = *reinterpret_cast<const float4*>(sm + lane * 4), a1 = a0, b0 = a0, b1 = a0;
  const float4* gq = reinterpret_cast<const float4*>(g) + lane;
  for (int it = 0; it < iters; ++it) {
    if (VARIANT == 0) {            // pure MFMA, 4 accumulators, 16 per iteration
      __builtin_amdgcn_sched_barrier(0);
    } else if (VARIANT == 1) {     // + 2 ds_read_b128 per 16 MFMAs (conv fwd A operand)
      const int o = ((it * 64 + lane) & 1023) * 4;
      a0 = *reinterpret_cast<const float4*>(sm + o);
      a1 = *reinterpret_cast<const float4*>(sm + o + 4096);
      __builtin_amdgcn_sched_barrier(0);
    } else if (VARIANT == 2) {     // + 2 ds_read_b128 + 2 global_load_dwordx4 (conv fwd v2 step)
      const int o = ((it * 64 + lane) & 1023) * 4;
      a0 = *reinterpret_cast<const float4*>(sm + o);
      a1 = *reinterpret_cast<const float4*>(sm + o + 4096);
      b0 = gq[(it & 15) * 64];
      b1 = gq[(it & 15) * 64 + 1024];
      __builtin_amdgcn_sched_barrier(0);
    }
    if (VARIANT <= 2) {
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          const float4 a = m ? a1 : a0, b = n ? b1 : b0;
          acc[m * 2 + n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[m * 2 + n], 0, 0, 0);
          acc[m * 2 + n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[m * 2 + n], 0, 0, 0);
          acc[m * 2 + n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc[m * 2 + n], 0, 0, 0);
          acc[m * 2 + n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc[m * 2 + n], 0, 0, 0);
        }
      __builtin_amdgcn_sched_barrier(0);
    } else {                       // wgrad step: 10 ds_read_b32 + 9 MFMAs on 9 accumulators
      const int o = (it * 64 + lane) & 2047;
      const float b = sm[o + 8192];
      float a[9];
#pragma unroll
      for (int t = 0; t < 9; ++t) a[t] = sm[o + t * 640];
      if (VARIANT == 4) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < 9; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b, acc[t], 0, 0, 0);
      if (VARIANT == 4) __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0.f;
  for (int t = 0; t < 9; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int VARIANT>
void run(const char* name, int threads, int mfma_per_iter, const float* g, float* out) {
  const int iters = 20000, blocks = 256;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((probe<VARIANT>), dim3(blocks), dim3(threads), 0, 0, g, out, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((probe<VARIANT>), dim3(blocks), dim3(threads), 0, 0, g, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  const double flop = (double)blocks * (threads / 64) * iters * mfma_per_iter * 4096.0;
  printf("%-44s threads/block %3d (%d wave/SIMD): %8.3f ms  %7.1f TFLOP/s\n", name, threads, threads / 256, ms, flop / ms / 1e9);
}

int main() {
  float *g, *out;
  hipMalloc(&g, 1 << 20); hipMalloc(&out, 256 * 512 * 4);
  std::vector<float> h(1 << 18);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) / 1000.f - 0.5f;
  hipMemcpy(g, h.data(), 1 << 20, hipMemcpyHostToDevice);
  for (int threads : {256, 512}) {
    run<0>("pure MFMA x16, 4 acc", threads, 16, g, out);
    run<1>("+2 ds_read_b128 per 16 MFMA", threads, 16, g, out);
    run<2>("+2 ds_read_b128 +2 global_load_x4 per 16", threads, 16, g, out);
    run<3>("wgrad step: 10 ds_read_b32 + 9 MFMA", threads, 9, g, out);
    run<4>("wgrad step, pinned (reads then MFMAs)", threads, 9, g, out);
  }
  return 0;
}
