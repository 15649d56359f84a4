// Shared device/host helpers for the OD-VAE gfx950 kernels.
// Everything here is CDNA4-only: 64-lane wavefronts, v_mfma_f32_32x32x2_f32.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <algorithm>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define ODVAE_OK 0
#define ODVAE_ERR_ARG 1
#define ODVAE_ERR_WORKSPACE 2
#define ODVAE_ERR_HIP 3

// last error text, readable through odvae_last_error()
extern "C" void odvae_set_error(const char* fmt, ...);

#define ODVAE_CHECK_ARG(cond, ...)                 \
  do {                                             \
    if (!(cond)) {                                 \
      odvae_set_error(__VA_ARGS__);                \
      return ODVAE_ERR_ARG;                        \
    }                                              \
  } while (0)

#define ODVAE_LAUNCH_CHECK(name)                                              \
  do {                                                                        \
    hipError_t e_ = hipGetLastError();                                        \
    if (e_ != hipSuccess) {                                                   \
      odvae_set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); \
      return ODVAE_ERR_HIP;                                                   \
    }                                                                         \
  } while (0)

// One weight of a batched pack launch (odvae_*_pack_*_batch): OIHW f32 weight, its forward / data-gradient pack buffers (either may be null),
// channel counts, taps (9 or 1; bf16 packs only).  The channel counts of a batched item need no padding in either pack.
struct OdvaePackItem { const float* w; void* fwd; void* dgr; int Cout, Cin, taps, reserved; };

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// One exact-f32 matrix-core step: D(32x32) += A(32x2) * B(2x32).
// lane l supplies A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31];
// D register r of lane l is D[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31].
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
#else
  (void)a; (void)b; return c;  // host pass only parses kernels
#endif
}

__device__ __forceinline__ int acc_row(int r, int lane) {
  return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
}

// XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (each with its own L2), so with the plain order the
// eight neighbours of a tile run on eight different L2s and every halo / weight line is fetched once per XCD.  This bijection
// gives XCD k the k-th contiguous eighth of the tile range (MI355X_MICROARCH.md, workgroup dispatch; cdna_hip_programming.md T1).
__device__ __forceinline__ int xcd_contiguous(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_f64(double v) {      // fixed butterfly order: deterministic
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
