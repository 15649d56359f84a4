// Small-batch dense layers of the pose head on the f32 MFMA (v_mfma_f32_32x32x2_f32), gfx950: PoseDecoderSpatialVAE
// (4096 -> 500 -> 500 -> 27, tanh; src/modules/autoencodermodules/pose_decoder.py:60-97) and PoseEncoderSpatialVAE (coord_linear
// 512 -> 1024, latent_linear 19 -> 4, 1024 -> 500 -> 4096, swish; pose_encoder.py:59-131), forward, data gradient and weight gradient.
// M = batch (32) rows only, so the shapes are skinny and weight-bandwidth bound; the library GEMM (gemm_f32.hip) wants 128 x 128
// tiles and 16-byte aligned inner dimensions, which 19, 27 and 4 are not.  Here ONE wave owns a 32 x 32 output tile of one K
// split; operands are addressed through (row, k) strides, so the three layouts are one kernel:
//   C[i][j] = sum_k A[i * sa_i + k * sa_k] * B[k * sb_k + j * sb_j]
//   forward  y = x W^T      : i = m, j = n, k = in-feature    (sa_i = K, sa_k = 1; sb_k = 1, sb_j = K)
//   dgrad    dx = dpre W    : i = m, j = in-feature, k = n    (sa_i = N, sa_k = 1; sb_k = K, sb_j = 1)
//   wgrad    dW = dpre^T x  : i = n, j = in-feature, k = m    (sa_i = 1, sa_k = N; sb_k = K, sb_j = 1)
// Split-K partials [split][I][J] are summed in fixed order by the epilogue, which also adds the bias and applies the activation
// (and keeps the pre-activation for the backward): deterministic, no atomics.
#include "common.h"

namespace {

struct SmallGemm {
  const float* A; const float* B; float* C;   // C: [splits][I][J]
  int I, J, K, sa_i, sa_k, sb_k, sb_j;
  int tiles_i, kper;
};

__global__ __launch_bounds__(64) void small_gemm_kernel(SmallGemm p) {
  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  const int ti = blockIdx.x % p.tiles_i, tj = blockIdx.x / p.tiles_i, split = blockIdx.y;
  const int i = ti * 32 + r, j = tj * 32 + r;
  const int kbeg = split * p.kper, kend = min(p.K, kbeg + p.kper);
  const bool iok = i < p.I, jok = j < p.J;
  const float* Ai = p.A + (int64_t)(iok ? i : 0) * p.sa_i;
  const float* Bj = p.B + (int64_t)(jok ? j : 0) * p.sb_j;
  f32x16 acc;
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q] = 0.f;
  for (int k0 = kbeg; k0 < kend; k0 += 8) {      // lane half h takes k0 + 4h .. + 3 on both operands
    float a[4], b[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int k = k0 + 4 * h + s;
      const bool ok = k < kend;
      a[s] = (ok && iok) ? Ai[(int64_t)k * p.sa_k] : 0.f;
      b[s] = (ok && jok) ? Bj[(int64_t)k * p.sb_k] : 0.f;
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = mfma32(a[s], b[s], acc);
  }
  float* C = p.C + (int64_t)split * p.I * p.J;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int row = ti * 32 + acc_row(q, lane), col = tj * 32 + r;
    if (row < p.I && col < p.J) C[(int64_t)row * p.J + col] = acc[q];
  }
}

__device__ __forceinline__ float act_f(float x, int kind) {
  if (kind == 1) return tanhf(x);
  if (kind == 2) return x / (1.f + __expf(-x));
  if (kind == 3) return fmaxf(x, 0.f);
  return x;
}
__device__ __forceinline__ float act_grad(float x, int kind) {
  if (kind == 1) { const float t = tanhf(x); return 1.f - t * t; }
  if (kind == 2) { const float s = 1.f / (1.f + __expf(-x)); return s * (1.f + x * (1.f - s)); }
  if (kind == 3) return x > 0.f ? 1.f : 0.f;
  return 1.f;
}

// y = act(sum_s part[s] + bias[col]); pre (optional) keeps the argument of act
__global__ void linear_epilogue_kernel(const float* __restrict__ part, int splits, int I, int J, const float* __restrict__ bias, int act,
                                       float* __restrict__ y, float* __restrict__ pre) {
  const int64_t n = (int64_t)I * J;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * blockDim.x) {
    float s = bias ? bias[idx % J] : 0.f;
    for (int k = 0; k < splits; ++k) s += part[(int64_t)k * n + idx];
    if (pre) pre[idx] = s;
    y[idx] = act_f(s, act);
  }
}

__global__ void act_bwd_kernel(const float* __restrict__ pre, const float* __restrict__ dy, int act, int64_t n, float* __restrict__ dpre) {
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * blockDim.x)
    dpre[idx] = dy[idx] * act_grad(pre[idx], act);
}

int pick_splits(int I, int J, int K) {
  const int tiles = ceil_div(I, 32) * ceil_div(J, 32);
  int s = std::max(1, 512 / tiles);                       // ~2 waves per CU
  s = std::min(s, std::max(1, K / 64));                   // at least 64 reduction steps per split
  return s;
}

int run_gemm(const float* A, const float* B, int I, int J, int K, int sa_i, int sa_k, int sb_k, int sb_j, float* part, int splits, hipStream_t st) {
  SmallGemm p{A, B, part, I, J, K, sa_i, sa_k, sb_k, sb_j, ceil_div(I, 32), (ceil_div(K, splits) + 7) / 8 * 8};
  hipLaunchKernelGGL(small_gemm_kernel, dim3(ceil_div(I, 32) * ceil_div(J, 32), ceil_div(K, p.kper)), dim3(64), 0, st, p);
  return ceil_div(K, p.kper);
}

}  // namespace

extern "C" {

// floats of scratch for odvae_linear_fwd_f32 / odvae_linear_bwd_f32 (split-K partials of the largest product)
size_t odvae_linear_workspace_bytes(int M, int N, int K) {
  const size_t f = (size_t)pick_splits(M, N, K) * M * N, d = (size_t)pick_splits(M, K, N) * M * K;
  return std::max(f, d) * sizeof(float);
}

// y [M][N] = act(x [M][K] . w [N][K]^T + bias [N] or NULL); act: 0 none, 1 tanh, 2 swish, 3 relu; pre [M][N] (optional) receives the
// pre-activation for the backward
int odvae_linear_fwd_f32(const float* x, const float* w, const float* bias, int M, int N, int K, int act, float* y, float* pre,
                         void* workspace, size_t workspace_bytes, void* stream) {
  ODVAE_CHECK_ARG(x && w && y && M > 0 && N > 0 && K > 0 && act >= 0 && act <= 3, "linear_fwd: bad arguments");
  if (!workspace || workspace_bytes < odvae_linear_workspace_bytes(M, N, K)) {
    odvae_set_error("linear_fwd: needs %zu workspace bytes, got %zu", odvae_linear_workspace_bytes(M, N, K), workspace_bytes);
    return ODVAE_ERR_WORKSPACE;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  float* part = static_cast<float*>(workspace);
  const int splits = run_gemm(x, w, M, N, K, K, 1, 1, K, part, pick_splits(M, N, K), st);
  hipLaunchKernelGGL(linear_epilogue_kernel, dim3(std::min(ceil_div(M * N, 256), 1024)), dim3(256), 0, st, part, splits, M, N, bias, act, y, pre);
  ODVAE_LAUNCH_CHECK("linear_fwd");
  return ODVAE_OK;
}

// dy [M][N] -> dpre = dy * act'(pre) (in `dpre`, [M][N]); dx [M][K] = dpre . w (or NULL); dw [N][K] = dpre^T . x (or NULL).
// The bias gradient is the column sum of dpre (odvae_colsum_f32).
int odvae_linear_bwd_f32(const float* x, const float* w, const float* pre, const float* dy, int M, int N, int K, int act, float* dpre,
                         float* dx, float* dw, void* workspace, size_t workspace_bytes, void* stream) {
  ODVAE_CHECK_ARG(x && w && dy && dpre && M > 0 && N > 0 && K > 0 && (act == 0 || pre), "linear_bwd: bad arguments");
  if (!workspace || workspace_bytes < odvae_linear_workspace_bytes(M, N, K)) {
    odvae_set_error("linear_bwd: needs %zu workspace bytes, got %zu", odvae_linear_workspace_bytes(M, N, K), workspace_bytes);
    return ODVAE_ERR_WORKSPACE;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  float* part = static_cast<float*>(workspace);
  const int64_t n = (int64_t)M * N;
  hipLaunchKernelGGL(act_bwd_kernel, dim3((unsigned)std::min<int64_t>(ceil_div64(n, 256), 1024)), dim3(256), 0, st, act ? pre : dy, dy, act, n, dpre);
  if (dx) {
    const int splits = run_gemm(dpre, w, M, K, N, N, 1, K, 1, part, pick_splits(M, K, N), st);
    hipLaunchKernelGGL(linear_epilogue_kernel, dim3(std::min(ceil_div(M * K, 256), 1024)), dim3(256), 0, st, part, splits, M, K,
                       static_cast<const float*>(nullptr), 0, dx, static_cast<float*>(nullptr));
  }
  if (dw) run_gemm(dpre, x, N, K, M, 1, N, K, 1, dw, 1, st);      // reduction over the batch only: one split, written in place
  ODVAE_LAUNCH_CHECK("linear_bwd");
  return ODVAE_OK;
}

}  // extern "C"
