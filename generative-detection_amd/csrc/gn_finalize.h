// GroupNorm: the small reductions between the streaming passes (chunk partials -> mean / rstd; chunk partials -> per-channel and
// per-group backward sums; per-sample sums -> dgamma / dbeta), shared by the f32 kernels (groupnorm.hip) and the bf16 ones
// (bf16_ops.hip).  S is the file's shape struct (fields N, HW, C, G, cpg, chunks).  Include inside the file's anonymous namespace.
// All sums run in f64 in a fixed order (deterministic).  They used to be serial loops per thread (9-14 us per launch, 201
// launches per step); lane- / wavefront-strided partial sums bring them to a few microseconds.
#pragma once

// one wavefront per (n, g): the chunk partials are summed in f64, lane-strided then by a fixed butterfly (deterministic);
// a serial loop per thread took 14 us per launch for 64 chunks, 67 launches per step
template <typename S>
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* __restrict__ partial, S s, float eps,
                                                          float* __restrict__ mean, float* __restrict__ rstd) {
  const int lane = threadIdx.x & 63;
  const int idx = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (idx >= s.N * s.G) return;
  const int n = idx / s.G, g = idx % s.G;
  double a = 0.0, b = 0.0;
  for (int ch = lane; ch < s.chunks; ch += 64) {
    const float2 o = *reinterpret_cast<const float2*>(partial + (((int64_t)n * s.chunks + ch) * s.G + g) * 2);
    a += (double)o.x; b += (double)o.y;
  }
  a = wave_sum_f64(a); b = wave_sum_f64(b);
  if (lane == 0) {
    const double m = (double)s.HW * s.cpg;
    const double mu = a / m;
    double var = b / m - mu * mu;
    if (var < 0.0) var = 0.0;
    mean[idx] = (float)mu;
    rstd[idx] = (float)(1.0 / sqrt(var + (double)eps));
  }
}

// per (n, c): A = sum du*xhat, B = sum du (f64 over chunks).  Block (n, channel slice of CB = the largest multiple of cpg that
// fits 64): wavefront k sums the chunks k, k + 4, ... of the slice's channels (coalesced rows of the partials), the four partial
// sums meet in LDS in fixed order.  Writes chan[n][2][C] and the per-group ds1 = sum_c gamma*A, ds2 = sum_c gamma*B into
// grp[n][G][2].  cpg <= 64 (the launcher keeps the serial kernel below for wider groups).
template <typename S>
__global__ __launch_bounds__(256) void gn_bwd_finalize_kernel(const float* __restrict__ partial, S s, const float* __restrict__ gamma,
                                                              float* __restrict__ chan, float* __restrict__ grp) {
  __shared__ double part[4][2][64];
  __shared__ float sh[2][64];
  const int n = blockIdx.x, cl = threadIdx.x & 63, k = threadIdx.x >> 6;
  const int gpb = 64 / s.cpg, CB = gpb * s.cpg;     // groups / channels of this block's slice
  const int c = blockIdx.y * CB + cl;
  const bool live = cl < CB && c < s.C;
  double a = 0.0, b = 0.0;
  if (live)
    for (int ch = k; ch < s.chunks; ch += 4) {
      const float* o = partial + ((int64_t)n * s.chunks + ch) * 2 * s.C;
      a += (double)o[c]; b += (double)o[s.C + c];
    }
  part[k][0][cl] = a; part[k][1][cl] = b;
  __syncthreads();
  if (k == 0) {
    a = (part[0][0][cl] + part[1][0][cl]) + (part[2][0][cl] + part[3][0][cl]);
    b = (part[0][1][cl] + part[1][1][cl]) + (part[2][1][cl] + part[3][1][cl]);
    const double gm = live ? (double)gamma[c] : 0.0;
    if (live) {
      chan[((int64_t)n * 2 + 0) * s.C + c] = (float)a;
      chan[((int64_t)n * 2 + 1) * s.C + c] = (float)b;
    }
    sh[0][cl] = (float)(a * gm);
    sh[1][cl] = (float)(b * gm);
  }
  __syncthreads();
  if (threadIdx.x < gpb) {
    const int g = blockIdx.y * gpb + threadIdx.x;
    if (g < s.G) {
      float ga = 0.f, gb = 0.f;
      for (int j = 0; j < s.cpg; ++j) { ga += sh[0][threadIdx.x * s.cpg + j]; gb += sh[1][threadIdx.x * s.cpg + j]; }
      grp[((int64_t)n * s.G + g) * 2 + 0] = ga;
      grp[((int64_t)n * s.G + g) * 2 + 1] = gb;
    }
  }
}

// the same for groups wider than 64 channels: one block per sample, thread per channel, serial over the chunks
template <typename S>
__global__ void gn_bwd_finalize_wide_kernel(const float* __restrict__ partial, S s, const float* __restrict__ gamma,
                                            float* __restrict__ chan, float* __restrict__ grp) {
  extern __shared__ float shw[];  // [2][C]
  const int n = blockIdx.x;
  for (int c = threadIdx.x; c < s.C; c += blockDim.x) {
    double a = 0.0, b = 0.0;
    for (int ch = 0; ch < s.chunks; ++ch) {
      const float* o = partial + ((int64_t)n * s.chunks + ch) * 2 * s.C;
      a += (double)o[c]; b += (double)o[s.C + c];
    }
    chan[((int64_t)n * 2 + 0) * s.C + c] = (float)a;
    chan[((int64_t)n * 2 + 1) * s.C + c] = (float)b;
    shw[c] = (float)(a * (double)gamma[c]);
    shw[s.C + c] = (float)(b * (double)gamma[c]);
  }
  __syncthreads();
  for (int g = threadIdx.x; g < s.G; g += blockDim.x) {
    float a = 0.f, b = 0.f;
    for (int j = 0; j < s.cpg; ++j) { a += shw[g * s.cpg + j]; b += shw[s.C + g * s.cpg + j]; }
    grp[((int64_t)n * s.G + g) * 2 + 0] = a;
    grp[((int64_t)n * s.G + g) * 2 + 1] = b;
  }
}

// dgamma[c] = sum_n A[n][c], dbeta[c] = sum_n B[n][c]: block = 64 channels x 4 interleaved shares of the samples, f64
__global__ __launch_bounds__(256) void gn_bwd_param_kernel(const float* __restrict__ chan, int N, int C,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta) {
  __shared__ double part[4][2][64];
  const int cl = threadIdx.x & 63, k = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  double a = 0.0, b = 0.0;
  if (c < C)
    for (int n = k; n < N; n += 4) { a += (double)chan[((int64_t)n * 2 + 0) * C + c]; b += (double)chan[((int64_t)n * 2 + 1) * C + c]; }
  part[k][0][cl] = a; part[k][1][cl] = b;
  __syncthreads();
  if (k == 0 && c < C) {
    dgamma[c] = (float)((part[0][0][cl] + part[1][0][cl]) + (part[2][0][cl] + part[3][0][cl]));
    dbeta[c] = (float)((part[0][1][cl] + part[1][1][cl]) + (part[2][1][cl] + part[3][1][cl]));
  }
}

