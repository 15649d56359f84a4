// Micro-benchmark: do v_mfma_f32_32x32x16_bf16 and ordinary vector instructions of ANOTHER wave on the same SIMD execute at the same time?
// 512-thread blocks = 2 waves per SIMD.  ROLE 0: all 8 waves issue MFMAs only.  ROLE 1: all 8 waves issue v_fma only.  ROLE 2: waves 0-3 (one
// per SIMD) MFMAs only, waves 4-7 v_fma only (the wave-pair layout of flash_attn_bf16.hip).  ROLE 3: every wave alternates 4 MFMAs and NV v_fma.
// ROLE 4 / 5: one class on four waves, the other four idle.  Prints the time of each and what ROLE 2 would take if the two classes serialised (t0/2 + t1/2) or overlapped perfectly (max(t0, t1)/2 ...).
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/mfma_valu_coexec_probe.hip -o /tmp/coexec && /tmp/coexec
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int ROLE, int NV>
__global__ __launch_bounds__(512) void probe(float* out, int iters) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x16 acc[4];
  for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = (float)(lane + r);
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.001f * (lane + j)); b[j] = (__bf16)(0.002f * (lane - j)); }
  float v[8];
  for (int j = 0; j < 8; ++j) v[j] = 1.0f + 0.001f * (lane + j);
  const bool do_mfma = ROLE == 0 || ROLE == 3 || ((ROLE == 2 || ROLE == 4 || ROLE == 7) && wave < 4);      // ROLE 4: waves 0-3 MFMAs, waves 4-7 idle
  const bool do_valu = ROLE == 1 || ROLE == 3 || ((ROLE == 2 || ROLE == 5) && wave >= 4);     // ROLE 5: waves 4-7 v_fma, waves 0-3 idle
  for (int it = 0; it < iters; ++it) {
    if (do_mfma) {
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[t], 0, 0, 0);
    }
    if (do_valu) {
#pragma unroll
      for (int k = 0; k < NV; ++k) v[k & 7] = __builtin_fmaf(v[k & 7], 1.0000001f, 0.5f);
    }
    if (ROLE == 6 || (ROLE == 7 && wave >= 4)) {      // the same multiply-adds as NV v_fma, issued as NV / 2 v_pk_fma_f32
      typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
      for (int k = 0; k < NV / 2; ++k) {
        f32x2 t = {v[(2 * k) & 7], v[(2 * k + 1) & 7]};
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(t) : "v"(f32x2{1.0000001f, 1.0000001f}), "v"(f32x2{0.5f, 0.5f}));
        v[(2 * k) & 7] = t.x; v[(2 * k + 1) & 7] = t.y;
      }
    }
  }
  float s = 0.f;
  for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
  for (int j = 0; j < 8; ++j) s += v[j];
  if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int ROLE, int NV>
float run(float* d, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((probe<ROLE, NV>), dim3(256), dim3(512), 0, 0, d, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((probe<ROLE, NV>), dim3(256), dim3(512), 0, 0, d, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}

int main() {
  float* d; hipMalloc(&d, 4096);
  const int iters = 20000;
  constexpr int NV = 32;      // 32 v_fma per 4 MFMAs (128 MFMA cycles): the ratio of the flash forward's softmax wave, roughly
  const float t0 = run<0, NV>(d, iters), t1 = run<1, NV>(d, iters), t2 = run<2, NV>(d, iters), t3 = run<3, NV>(d, iters);
  printf("8 waves MFMA only            %.3f ms (%.0f TFLOP/s)\n", t0, 256.0 * 8 * iters * 4 * 32768.0 / t0 / 1e9);
  printf("8 waves v_fma only (NV=%d)   %.3f ms\n", NV, t1);
  printf("4 waves MFMA + 4 waves v_fma %.3f ms   (serialised would be %.3f, perfectly overlapped %.3f)\n", t2, 0.5f * (t0 + t1), 0.5f * (t0 > t1 ? t0 : t1));
  const float t4 = run<4, NV>(d, iters), t5 = run<5, NV>(d, iters);
  printf("4 waves MFMA only (one per SIMD), 4 idle   %.3f ms (%.0f TFLOP/s)\n", t4, 256.0 * 4 * iters * 4 * 32768.0 / t4 / 1e9);
  printf("4 waves v_fma only, 4 idle                 %.3f ms\n", t5);
  const float t6 = run<6, NV>(d, iters), t7 = run<7, NV>(d, iters);
  printf("8 waves v_pk_fma_f32 only (NV / 2 = %d of them: the same multiply-adds)   %.3f ms\n", NV / 2, t6);
  printf("4 waves MFMA + 4 waves v_pk_fma_f32 (NV / 2)                              %.3f ms\n", t7);
  printf("8 waves, each 4 MFMA + %d v_fma alternating  %.3f ms   (serialised %.3f, overlapped %.3f)\n", NV, t3, t0 + t1, t0 > t1 ? t0 : t1);
  return 0;
}
