// Micro-benchmark: sustained v_mfma_f32_32x32x16_bf16 rate on random data (clock under load), bare and with the operand
// traffic of a "three bf16 limbs per f32 operand, nine limb products" step (2x2 tiles: 36 MFMAs per 16-deep k-step with
// 6 A-limb fragments from LDS and 6 B-limb fragments from global).  Build on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_bf16_probe.hip -o /tmp/mfma_bf16_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int VARIANT>
__global__ __launch_bounds__(512) void probe(const uint4* __restrict__ g, float* out, int iters) {
  __shared__ __attribute__((aligned(16))) uint4 sm[4096];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) sm[i] = g[i & 1023];
  __syncthreads();
  f32x16 acc[4];
  for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  uint4 a[2][3], b[2][3];
  for (int m = 0; m < 2; ++m) for (int l = 0; l < 3; ++l) { a[m][l] = sm[(lane + 64 * (m * 3 + l)) & 4095]; b[m][l] = sm[(lane + 64 * (6 + m * 3 + l)) & 4095]; }
  for (int it = 0; it < iters; ++it) {
    if (VARIANT == 1) {   // operands re-fetched per step: A limbs from LDS, B limbs from global (L2-resident)
      const int o = (it * 64 + lane) & 511;
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int l = 0; l < 3; ++l) { a[m][l] = sm[o + 512 * (m * 3 + l)]; b[m][l] = g[((it & 7) * 64 + lane + 512 * (m * 3 + l)) & 8191]; }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int la = 0; la < 3; ++la)
#pragma unroll
          for (int lb = 0; lb < 3; ++lb)
            acc[m * 2 + n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[m][la]), __builtin_bit_cast(bf16x8, b[n][lb]),
                                                                     acc[m * 2 + n], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
  float s = 0.f;
  for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int VARIANT>
void run(const char* name, int threads, const uint4* g, float* out, int iters) {
  const int blocks = 256;
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
  const double flop = (double)blocks * (threads / 64) * iters * 36.0 * 2.0 * 32 * 32 * 16;
  printf("%-58s %d wave/SIMD: %8.2f ms  %7.1f TFLOP/s bf16 = %6.1f TFLOP/s of f32 products (x9)\n", name, threads / 256, ms,
         flop / ms / 1e9, flop / ms / 1e9 / 9.0);
}

int main() {
  std::vector<unsigned short> h(8192 * 8);
  uint4 *gz, *gr; float* out;
  hipMalloc(&gz, h.size() * 2); hipMalloc(&gr, h.size() * 2); hipMalloc(&out, 256 * 512 * 4);
  hipMemset(gz, 0, h.size() * 2);
  srand(1);
  for (auto& v : h) v = (unsigned short)(0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15));   // random bf16 around +-1
  hipMemcpy(gr, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  const int iters = 60000;
  run<0>("bare, zeros", 256, gz, out, iters);
  run<0>("bare, random data", 256, gr, out, iters);
  run<0>("bare, random data", 512, gr, out, iters);
  run<1>("6 ds_read_b128 + 6 global b128 per 36 MFMAs, random", 256, gr, out, iters);
  run<1>("6 ds_read_b128 + 6 global b128 per 36 MFMAs, random", 512, gr, out, iters);
  return 0;
}
