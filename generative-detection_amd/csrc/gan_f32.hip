// PatchGAN discriminator support kernels, NHWC f32, gfx950.  All HBM-bound.
//
// [UPSTREAM] taming/modules/discriminator/model.py NLayerDiscriminator (reference call sites
// src/modules/losses/contperceptual.py:285,355-356): Conv2d(4x4, stride 2|1, pad 1) -> BatchNorm2d -> LeakyReLU(0.2).
// The 4x4 convolutions run as im2col + the f32 MFMA GEMM (gemm_f32.hip) -- BASELINE.json's north_star allows the matrix
// cores for "im2col dense contractions"; 288 GB of HBM makes the column matrices (<= 1 GB at B = 32) a non-issue.
//   fwd   cols = im2col(x)           y    = cols . W'^T     (W' = weight reordered to [Cout][kh][kw][Cin])
//   dgrad dcols = dy . W'            dx   = col2im(dcols)   (gather form: no atomics, deterministic)
//   wgrad dW'  = dy^T . cols
// BatchNorm2d uses batch statistics in training (biased variance for normalisation, unbiased for the running
// estimate, momentum 0.1) and is fused with the LeakyReLU that follows it.
#include "common.h"

namespace {

int grid_1d(int64_t items, int cap = 8192) { return (int)std::min<int64_t>(std::max<int64_t>(ceil_div64(items, 256), 1), cap); }

// cols[(n,oy,ox)][(kh*4+kw)*C + c] = x[n][oy*S-1+kh][ox*S-1+kw][c]  (zero outside)
__global__ __launch_bounds__(256) void im2col4x4_kernel(const float* __restrict__ x, float* __restrict__ cols,
                                                         int N, int Hi, int Wi, int C, int Ho, int Wo, int S) {
  const int64_t total = (int64_t)N * Ho * Wo * 16 * C;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(idx % C);
    int64_t r = idx / C;
    const int tap = (int)(r % 16); r /= 16;
    const int ox = (int)(r % Wo); r /= Wo;
    const int oy = (int)(r % Ho); const int n = (int)(r / Ho);
    const int iy = oy * S - 1 + (tap >> 2), ix = ox * S - 1 + (tap & 3);
    float v = 0.f;
    if (iy >= 0 && iy < Hi && ix >= 0 && ix < Wi) v = x[(((int64_t)n * Hi + iy) * Wi + ix) * C + c];
    cols[idx] = v;
  }
}

// dx[n][iy][ix][c] = sum over taps (kh,kw) with (iy+1-kh) % S == 0 of dcols[(n,(iy+1-kh)/S,(ix+1-kw)/S)][(kh*4+kw)*C+c]
__global__ __launch_bounds__(256) void col2im4x4_kernel(const float* __restrict__ dcols, float* __restrict__ dx,
                                                         int N, int Hi, int Wi, int C, int Ho, int Wo, int S) {
  const int64_t total = (int64_t)N * Hi * Wi * C;
  const int K = 16 * C;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(idx % C);
    int64_t r = idx / C;
    const int ix = (int)(r % Wi); r /= Wi;
    const int iy = (int)(r % Hi); const int n = (int)(r / Hi);
    float s = 0.f;
#pragma unroll
    for (int kh = 0; kh < 4; ++kh) {
      const int ty = iy + 1 - kh;
      if (ty < 0 || ty % S != 0) continue;
      const int oy = ty / S;
      if (oy >= Ho) continue;
#pragma unroll
      for (int kw = 0; kw < 4; ++kw) {
        const int tx = ix + 1 - kw;
        if (tx < 0 || tx % S != 0) continue;
        const int ox = tx / S;
        if (ox >= Wo) continue;
        s += dcols[(((int64_t)n * Ho + oy) * Wo + ox) * K + (kh * 4 + kw) * C + c];
      }
    }
    dx[idx] = s;
  }
}

// OIHW [Cout][Cin][4][4] <-> GEMM layout [Cout][(kh*4+kw)*Cin + ci]
__global__ void weight4x4_reorder_kernel(const float* __restrict__ src, float* __restrict__ dst, int Cout, int Cin, int to_gemm) {
  const int64_t total = (int64_t)Cout * Cin * 16;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    // idx enumerates the OIHW side
    const int tap = (int)(idx % 16);
    const int ci = (int)((idx / 16) % Cin);
    const int co = (int)(idx / (16 * (int64_t)Cin));
    const int64_t g = ((int64_t)co * 16 + tap) * Cin + ci;
    if (to_gemm) dst[g] = src[idx]; else dst[idx] = src[g];
  }
}

// ---- per-channel statistics over rows of an [M][C] matrix (shared by BN forward and backward) ---------------
// part[blk][2][C]: sums of f1 and f2 per channel, where (f1,f2) = (x, x^2) for MODE 0 and
// (g, g*xhat) for MODE 1 with g = dy * lrelu'(u), u = xhat*gamma+beta
template <int MODE>
__global__ __launch_bounds__(256) void bn_colstats_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                          const float* __restrict__ mean, const float* __restrict__ rstd,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float slope, int64_t rows, int C, int rows_per_block,
                                                          float* __restrict__ part) {
  __shared__ float sh[2][256];
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = std::min<int64_t>(rows, r0 + rows_per_block);
  for (int cbase = 0; cbase < C; cbase += 256) {
    const int cw = min(256, C - cbase);
    const int lanes = 256 / cw;
    const int c = cbase + threadIdx.x % cw, rl = threadIdx.x / cw;
    float a = 0.f, b = 0.f;
    if (rl < lanes) {
      float mu = 0.f, rs = 0.f, ga = 0.f, be = 0.f;
      if (MODE == 1) { mu = mean[c]; rs = rstd[c]; ga = gamma[c]; be = beta[c]; }
      for (int64_t r = r0 + rl; r < r1; r += lanes) {
        const float v = x[r * C + c];
        if (MODE == 0) { a += v; b += v * v; }
        else {
          const float xh = (v - mu) * rs;
          const float u = xh * ga + be;
          const float g = dy[r * C + c] * (u > 0.f ? 1.f : slope);
          a += g; b += g * xh;
        }
      }
    }
    sh[0][threadIdx.x] = a; sh[1][threadIdx.x] = b;
    __syncthreads();
    if (threadIdx.x < cw) {
      float ta = 0.f, tb = 0.f;
      for (int k = 0; k < lanes; ++k) { ta += sh[0][k * cw + threadIdx.x]; tb += sh[1][k * cw + threadIdx.x]; }
      part[((int64_t)blockIdx.x * 2 + 0) * C + cbase + threadIdx.x] = ta;
      part[((int64_t)blockIdx.x * 2 + 1) * C + cbase + threadIdx.x] = tb;
    }
    __syncthreads();
  }
}

// training-mode forward statistics: mean, rstd and the running estimates (momentum update, unbiased variance)
__global__ void bn_finalize_kernel(const float* __restrict__ part, int nblk, int C, int64_t rows, float eps, float momentum,
                                   float* __restrict__ mean, float* __restrict__ rstd,
                                   float* __restrict__ running_mean, float* __restrict__ running_var) {
  // one wavefront per channel: the partials are summed lane-strided in f64, then by a fixed butterfly (deterministic).  One thread per
  // channel walking all nblk partials was a chain of ~1 000 dependent loads: 125 us per launch for 4 KB of output.
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (c >= C) return;
  double a = 0.0, b = 0.0;
  for (int k = lane; k < nblk; k += 64) { a += (double)part[((int64_t)k * 2 + 0) * C + c]; b += (double)part[((int64_t)k * 2 + 1) * C + c]; }
  a = wave_sum_f64(a); b = wave_sum_f64(b);
  if (lane != 0) return;
  const double m = (double)rows;
  const double mu = a / m;
  double var = b / m - mu * mu;
  if (var < 0.0) var = 0.0;
  mean[c] = (float)mu;
  rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (running_mean) {
    const double unbiased = rows > 1 ? var * m / (m - 1.0) : var;
    running_mean[c] = (float)((1.0 - momentum) * (double)running_mean[c] + momentum * mu);
    running_var[c] = (float)((1.0 - momentum) * (double)running_var[c] + momentum * unbiased);
  }
}

// sums[2][C] = column totals of the partials (backward: sum g, sum g*xhat)
__global__ void bn_sum_partials_kernel(const float* __restrict__ part, int nblk, int C, float* __restrict__ sums) {
  const int lane = threadIdx.x & 63;      // one wavefront per (which, channel), as bn_finalize_kernel
  const int idx = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (idx >= 2 * C) return;
  const int which = idx / C, c = idx % C;
  double a = 0.0;
  for (int k = lane; k < nblk; k += 64) a += (double)part[((int64_t)k * 2 + which) * C + c];
  a = wave_sum_f64(a);
  if (lane == 0) sums[idx] = (float)a;
}

// y = lrelu((x-mean)*rstd*gamma+beta)
__global__ __launch_bounds__(256) void bn_lrelu_apply_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                             const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float slope, int64_t total, int C,
                                                             float* __restrict__ y) {
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(idx % C);
    const float u = (x[idx] - mean[c]) * rstd[c] * gamma[c] + beta[c];
    y[idx] = u > 0.f ? u : slope * u;
  }
}

// training: dx = gamma*rstd/M * (M*g - sum_g - xhat*sum_gxhat); eval (train=0): dx = gamma*rstd*g
__global__ __launch_bounds__(256) void bn_lrelu_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                 const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                 const float* __restrict__ sums, float slope, int64_t rows, int C,
                                                                 int train, float* __restrict__ dx) {
  const int64_t total = rows * C;
  const float inv_m = 1.f / (float)rows;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(idx % C);
    const float xh = (x[idx] - mean[c]) * rstd[c];
    const float u = xh * gamma[c] + beta[c];
    const float g = dy[idx] * (u > 0.f ? 1.f : slope);
    float v = g;
    if (train) v = g - (sums[c] + xh * sums[C + c]) * inv_m;
    dx[idx] = gamma[c] * rstd[c] * v;
  }
}

__global__ __launch_bounds__(256) void lrelu_kernel(const float* __restrict__ x, float* __restrict__ y, float slope, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float v = x[i];
    y[i] = v > 0.f ? v : slope * v;
  }
}
// dx = dy * (x > 0 ? 1 : slope); with slope = 0 this is the ReLU backward given the pre- or post-activation
__global__ __launch_bounds__(256) void lrelu_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                        float* __restrict__ dx, float slope, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    dx[i] = dy[i] * (x[i] > 0.f ? 1.f : slope);
}

int stat_blocks(int64_t rows) { return (int)std::min<int64_t>(std::max<int64_t>(rows / 128, 1), 1024); }

}  // namespace

extern "C" {

int odvae_im2col4x4_f32(const float* x, float* cols, int N, int Hi, int Wi, int C, int Ho, int Wo, int stride, void* stream) {
  ODVAE_CHECK_ARG(x && cols && N > 0 && C > 0 && (stride == 1 || stride == 2), "im2col4x4: bad arguments");
  ODVAE_CHECK_ARG(Ho == (Hi + 2 - 4) / stride + 1 && Wo == (Wi + 2 - 4) / stride + 1, "im2col4x4: Ho/Wo do not match k=4, pad=1, stride=%d", stride);
  hipLaunchKernelGGL(im2col4x4_kernel, dim3(grid_1d((int64_t)N * Ho * Wo * 16 * C, 16384)), dim3(256), 0, static_cast<hipStream_t>(stream), x, cols, N, Hi, Wi, C, Ho, Wo, stride);
  ODVAE_LAUNCH_CHECK("im2col4x4");
  return ODVAE_OK;
}

int odvae_col2im4x4_f32(const float* dcols, float* dx, int N, int Hi, int Wi, int C, int Ho, int Wo, int stride, void* stream) {
  ODVAE_CHECK_ARG(dcols && dx && N > 0 && C > 0 && (stride == 1 || stride == 2), "col2im4x4: bad arguments");
  hipLaunchKernelGGL(col2im4x4_kernel, dim3(grid_1d((int64_t)N * Hi * Wi * C, 16384)), dim3(256), 0, static_cast<hipStream_t>(stream), dcols, dx, N, Hi, Wi, C, Ho, Wo, stride);
  ODVAE_LAUNCH_CHECK("col2im4x4");
  return ODVAE_OK;
}

// to_gemm = 1: OIHW -> [Cout][(kh,kw,ci)]; to_gemm = 0: the inverse (weight gradients back to OIHW)
int odvae_weight4x4_reorder_f32(const float* src, float* dst, int Cout, int Cin, int to_gemm, void* stream) {
  ODVAE_CHECK_ARG(src && dst && Cout > 0 && Cin > 0, "weight4x4_reorder: bad arguments");
  hipLaunchKernelGGL(weight4x4_reorder_kernel, dim3(grid_1d((int64_t)Cout * Cin * 16)), dim3(256), 0, static_cast<hipStream_t>(stream), src, dst, Cout, Cin, to_gemm);
  ODVAE_LAUNCH_CHECK("weight4x4_reorder");
  return ODVAE_OK;
}

size_t odvae_batchnorm_workspace_bytes(int64_t rows, int C) { return ((size_t)stat_blocks(rows) * 2 * C + 2 * C) * sizeof(float); }

// x,y: [rows][C] (NHWC flattened).  train=1: batch statistics into mean/rstd (+ running stats update when given);
// train=0: mean/rstd must already hold the running estimates (rstd = 1/sqrt(running_var+eps)).
int odvae_batchnorm_lrelu_fwd_f32(const float* x, int64_t rows, int C, const float* gamma, const float* beta, float eps,
                                  float momentum, float slope, int train, float* mean, float* rstd,
                                  float* running_mean, float* running_var, float* y,
                                  void* workspace, size_t workspace_bytes, void* stream) {
  ODVAE_CHECK_ARG(x && gamma && beta && mean && rstd && y && rows > 0 && C > 0, "batchnorm_lrelu_fwd: bad arguments");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (train) {
    const size_t need = odvae_batchnorm_workspace_bytes(rows, C);
    if (!workspace || workspace_bytes < need) { odvae_set_error("batchnorm_lrelu_fwd: needs %zu workspace bytes", need); return ODVAE_ERR_WORKSPACE; }
    float* part = static_cast<float*>(workspace);
    const int nblk = stat_blocks(rows);
    const int rpb = (int)ceil_div64(rows, nblk);
    const int nb = (int)ceil_div64(rows, rpb);
    hipLaunchKernelGGL((bn_colstats_kernel<0>), dim3(nb), dim3(256), 0, st, x, nullptr, nullptr, nullptr, nullptr, nullptr, slope, rows, C, rpb, part);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(ceil_div(C, 4)), dim3(256), 0, st, part, nb, C, rows, eps, momentum, mean, rstd, running_mean, running_var);
  }
  hipLaunchKernelGGL(bn_lrelu_apply_kernel, dim3(grid_1d(rows * C)), dim3(256), 0, st, x, mean, rstd, gamma, beta, slope, rows * C, C, y);
  ODVAE_LAUNCH_CHECK("batchnorm_lrelu_fwd");
  return ODVAE_OK;
}

int odvae_batchnorm_lrelu_bwd_f32(const float* x, const float* dy, int64_t rows, int C, const float* gamma, const float* beta,
                                  const float* mean, const float* rstd, float slope, int train,
                                  float* dx, float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes, void* stream) {
  ODVAE_CHECK_ARG(x && dy && gamma && beta && mean && rstd && dx && dgamma && dbeta && rows > 0 && C > 0, "batchnorm_lrelu_bwd: bad arguments");
  const size_t need = odvae_batchnorm_workspace_bytes(rows, C);
  if (!workspace || workspace_bytes < need) { odvae_set_error("batchnorm_lrelu_bwd: needs %zu workspace bytes", need); return ODVAE_ERR_WORKSPACE; }
  hipStream_t st = static_cast<hipStream_t>(stream);
  float* part = static_cast<float*>(workspace);
  const int nblk = stat_blocks(rows);
  const int rpb = (int)ceil_div64(rows, nblk);
  const int nb = (int)ceil_div64(rows, rpb);
  float* sums = part + (size_t)nblk * 2 * C;
  hipLaunchKernelGGL((bn_colstats_kernel<1>), dim3(nb), dim3(256), 0, st, x, dy, mean, rstd, gamma, beta, slope, rows, C, rpb, part);
  hipLaunchKernelGGL(bn_sum_partials_kernel, dim3(ceil_div(2 * C, 4)), dim3(256), 0, st, part, nb, C, sums);
  // dbeta = sum g ; dgamma = sum g*xhat
  hipMemcpyAsync(dbeta, sums, (size_t)C * sizeof(float), hipMemcpyDeviceToDevice, st);
  hipMemcpyAsync(dgamma, sums + C, (size_t)C * sizeof(float), hipMemcpyDeviceToDevice, st);
  hipLaunchKernelGGL(bn_lrelu_bwd_apply_kernel, dim3(grid_1d(rows * C)), dim3(256), 0, st, x, dy, mean, rstd, gamma, beta, sums, slope, rows, C, train, dx);
  ODVAE_LAUNCH_CHECK("batchnorm_lrelu_bwd");
  return ODVAE_OK;
}

int odvae_leaky_relu_f32(const float* x, float* y, float slope, int64_t n, void* stream) {
  ODVAE_CHECK_ARG(x && y && n > 0, "leaky_relu: bad arguments");
  hipLaunchKernelGGL(lrelu_kernel, dim3(grid_1d(n)), dim3(256), 0, static_cast<hipStream_t>(stream), x, y, slope, n);
  ODVAE_LAUNCH_CHECK("leaky_relu");
  return ODVAE_OK;
}

int odvae_leaky_relu_bwd_f32(const float* x, const float* dy, float* dx, float slope, int64_t n, void* stream) {
  ODVAE_CHECK_ARG(x && dy && dx && n > 0, "leaky_relu_bwd: bad arguments");
  hipLaunchKernelGGL(lrelu_bwd_kernel, dim3(grid_1d(n)), dim3(256), 0, static_cast<hipStream_t>(stream), x, dy, dx, slope, n);
  ODVAE_LAUNCH_CHECK("leaky_relu_bwd");
  return ODVAE_OK;
}

}  // extern "C"
