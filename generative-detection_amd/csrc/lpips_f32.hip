// LPIPS-style perceptual distance support kernels, NHWC f32, gfx950.  All HBM-bound.
//
// [UPSTREAM] taming/modules/losses/lpips.py (reference call site src/modules/losses/contperceptual.py:143):
//   ScalingLayer  (x - shift) / scale                           -> scaling_layer_kernel
//   VGG16 slices  conv3x3+ReLU (conv3x3_f32.hip, act = 1), MaxPool2d(2,2) -> maxpool2x2_*_kernel
//   per tap k     normalize_tensor(f) = f / (sqrt(sum_c f^2) + 1e-10); lin_k((n0 - n1)^2); spatial mean
//                                                               -> lpips_distance_*_kernel (one wavefront per pixel)
#include "common.h"

namespace {

int grid_1d(int64_t items, int cap = 8192) { return (int)std::min<int64_t>(std::max<int64_t>(ceil_div64(items, 256), 1), cap); }

// y = (x - shift[c]) / scale[c];  bwd (INV): y = x / scale[c]
template <bool BWD>
__global__ __launch_bounds__(256) void scaling_layer_kernel(const float* __restrict__ x, const float* __restrict__ shift,
                                                            const float* __restrict__ scale, float* __restrict__ y, int64_t total, int C) {
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(idx % C);
    y[idx] = BWD ? x[idx] / scale[c] : (x[idx] - shift[c]) / scale[c];
  }
}

// y[n][oy][ox][c] = max over the 2x2 window of x (H, W even)
__global__ __launch_bounds__(256) void maxpool2x2_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int Ho, int Wo, int C) {
  const int q = C / 4;
  const int64_t total = (int64_t)N * Ho * Wo * q;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int cq = (int)(idx % q);
    int64_t r = idx / q;
    const int ox = (int)(r % Wo); r /= Wo;
    const int oy = (int)(r % Ho); const int n = (int)(r / Ho);
    const float* s = x + (((int64_t)n * 2 * Ho + 2 * oy) * 2 * Wo + 2 * ox) * C + 4 * cq;
    const float4 a = *reinterpret_cast<const float4*>(s), b = *reinterpret_cast<const float4*>(s + C);
    const float4 c = *reinterpret_cast<const float4*>(s + (int64_t)2 * Wo * C), d = *reinterpret_cast<const float4*>(s + (int64_t)2 * Wo * C + C);
    float4 o;
    o.x = fmaxf(fmaxf(a.x, b.x), fmaxf(c.x, d.x)); o.y = fmaxf(fmaxf(a.y, b.y), fmaxf(c.y, d.y));
    o.z = fmaxf(fmaxf(a.z, b.z), fmaxf(c.z, d.z)); o.w = fmaxf(fmaxf(a.w, b.w), fmaxf(c.w, d.w));
    *reinterpret_cast<float4*>(y + idx * 4) = o;
  }
}

// dx: dy goes to the first element of the window (row-major scan) that equals the max, zero elsewhere (torch rule)
__global__ __launch_bounds__(256) void maxpool2x2_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                             const float* __restrict__ dy, float* __restrict__ dx,
                                                             int N, int Ho, int Wo, int C) {
  const int64_t total = (int64_t)N * Ho * Wo * C;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(idx % C);
    int64_t r = idx / C;
    const int ox = (int)(r % Wo); r /= Wo;
    const int oy = (int)(r % Ho); const int n = (int)(r / Ho);
    const int64_t base = (((int64_t)n * 2 * Ho + 2 * oy) * 2 * Wo + 2 * ox) * C + c;
    const int64_t off[4] = {0, C, (int64_t)2 * Wo * C, (int64_t)2 * Wo * C + C};
    const float m = y[idx], g = dy[idx];
    bool done = false;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const bool hit = !done && x[base + off[k]] == m;
      dx[base + off[k]] = hit ? g : 0.f;
      done = done || hit;
    }
  }
}

// One wavefront per pixel, lanes over channels (C <= 512: <= 8 channels per lane).
// d(pixel) = sum_c w[c] * (f0/n0 - f1/n1)^2, n = sqrt(sum f^2) + 1e-10.  part[n][blk] = sum of d over the block's pixels
constexpr int MAXCPL = 8;
__global__ __launch_bounds__(256) void lpips_distance_kernel(const float* __restrict__ f0, const float* __restrict__ f1,
                                                             const float* __restrict__ w, int HW, int C, float* __restrict__ part) {
  __shared__ float sh[4];
  const int n = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cpl = C / 64 > 0 ? (C + 63) / 64 : 1;
  float acc = 0.f;
  for (int px = blockIdx.x * 4 + wave; px < HW; px += gridDim.x * 4) {
    const float* a = f0 + ((int64_t)n * HW + px) * C;
    const float* b = f1 + ((int64_t)n * HW + px) * C;
    float va[MAXCPL], vb[MAXCPL], sa = 0.f, sb = 0.f;
#pragma unroll
    for (int k = 0; k < MAXCPL; ++k) {
      const int c = lane + 64 * k;
      va[k] = (k < cpl && c < C) ? a[c] : 0.f;
      vb[k] = (k < cpl && c < C) ? b[c] : 0.f;
      sa += va[k] * va[k]; sb += vb[k] * vb[k];
    }
    sa = wave_sum(sa); sb = wave_sum(sb);
    const float ia = 1.f / (sqrtf(sa) + 1e-10f), ib = 1.f / (sqrtf(sb) + 1e-10f);
    float d = 0.f;
#pragma unroll
    for (int k = 0; k < MAXCPL; ++k) {
      const int c = lane + 64 * k;
      if (k < cpl && c < C) { const float t = va[k] * ia - vb[k] * ib; d += w[c] * t * t; }
    }
    acc += d;   // lane-partial; reduced once at the end
  }
  acc = wave_sum(acc);
  if (lane == 0) sh[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[(int64_t)n * gridDim.x + blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

__global__ void lpips_mean_final_kernel(const float* __restrict__ part, int nblk, float inv_hw, float* __restrict__ out) {
  const int n = blockIdx.x;
  double s = 0.0;
  for (int i = threadIdx.x; i < nblk; i += 64) s += (double)part[(int64_t)n * nblk + i];
  s = wave_sum_d(s);
  if (threadIdx.x == 0) out[n] = (float)(s * (double)inv_hw);
}

// gradient w.r.t. f1 (the reconstruction branch): g[n] = d loss / d out[n]
__global__ __launch_bounds__(256) void lpips_distance_bwd_kernel(const float* __restrict__ f0, const float* __restrict__ f1,
                                                                 const float* __restrict__ w, const float* __restrict__ g,
                                                                 int HW, int C, float* __restrict__ df1) {
  const int n = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cpl = (C + 63) / 64;
  const float gs = g[n] / (float)HW;
  for (int px = blockIdx.x * 4 + wave; px < HW; px += gridDim.x * 4) {
    const int64_t base = ((int64_t)n * HW + px) * C;
    float va[MAXCPL], vb[MAXCPL], sa = 0.f, sb = 0.f;
#pragma unroll
    for (int k = 0; k < MAXCPL; ++k) {
      const int c = lane + 64 * k;
      va[k] = (k < cpl && c < C) ? f0[base + c] : 0.f;
      vb[k] = (k < cpl && c < C) ? f1[base + c] : 0.f;
      sa += va[k] * va[k]; sb += vb[k] * vb[k];
    }
    sa = wave_sum(sa); sb = wave_sum(sb);
    const float s1 = sqrtf(sb);
    const float n1 = s1 + 1e-10f;
    const float ia = 1.f / (sqrtf(sa) + 1e-10f), ib = 1.f / n1;
    // gb[c] = d d / d b_c = -2 w (a_c - b_c);  df1_k = gb_k / n1 - (sum_c gb_c f1_c) f1_k / (n1^2 s1)
    float gb[MAXCPL], dot = 0.f;
#pragma unroll
    for (int k = 0; k < MAXCPL; ++k) {
      const int c = lane + 64 * k;
      gb[k] = (k < cpl && c < C) ? -2.f * w[c] * (va[k] * ia - vb[k] * ib) : 0.f;
      dot += gb[k] * vb[k];
    }
    dot = wave_sum(dot);
    const float coef = s1 > 0.f ? dot / (n1 * n1 * s1) : 0.f;   // torch gives NaN at an all-zero feature vector; 0 here
#pragma unroll
    for (int k = 0; k < MAXCPL; ++k) {
      const int c = lane + 64 * k;
      if (k < cpl && c < C) df1[base + c] = gs * (gb[k] * ib - coef * vb[k]);
    }
  }
}

}  // namespace

extern "C" {

// ScalingLayer: y = (x - shift[c]) / scale[c]; backward = 1: y = x / scale[c] (x = dy)
int odvae_scaling_layer_f32(const float* x, const float* shift, const float* scale, float* y, int64_t npix, int C, int backward, void* stream) {
  ODVAE_CHECK_ARG(x && shift && scale && y && npix > 0 && C > 0, "scaling_layer: bad arguments");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (backward) hipLaunchKernelGGL((scaling_layer_kernel<true>), dim3(grid_1d(npix * C)), dim3(256), 0, st, x, shift, scale, y, npix * C, C);
  else hipLaunchKernelGGL((scaling_layer_kernel<false>), dim3(grid_1d(npix * C)), dim3(256), 0, st, x, shift, scale, y, npix * C, C);
  ODVAE_LAUNCH_CHECK("scaling_layer");
  return ODVAE_OK;
}

// x [N][2Ho][2Wo][C] -> y [N][Ho][Wo][C]
int odvae_maxpool2x2_f32(const float* x, float* y, int N, int Ho, int Wo, int C, void* stream) {
  ODVAE_CHECK_ARG(x && y && N > 0 && Ho > 0 && Wo > 0 && C > 0 && C % 4 == 0, "maxpool2x2: need C %% 4 == 0");
  hipLaunchKernelGGL(maxpool2x2_kernel, dim3(grid_1d((int64_t)N * Ho * Wo * (C / 4))), dim3(256), 0, static_cast<hipStream_t>(stream), x, y, N, Ho, Wo, C);
  ODVAE_LAUNCH_CHECK("maxpool2x2");
  return ODVAE_OK;
}

int odvae_maxpool2x2_bwd_f32(const float* x, const float* y, const float* dy, float* dx, int N, int Ho, int Wo, int C, void* stream) {
  ODVAE_CHECK_ARG(x && y && dy && dx && N > 0 && Ho > 0 && Wo > 0 && C > 0, "maxpool2x2_bwd: bad arguments");
  hipLaunchKernelGGL(maxpool2x2_bwd_kernel, dim3(grid_1d((int64_t)N * Ho * Wo * C)), dim3(256), 0, static_cast<hipStream_t>(stream), x, y, dy, dx, N, Ho, Wo, C);
  ODVAE_LAUNCH_CHECK("maxpool2x2_bwd");
  return ODVAE_OK;
}

// out[n] = mean over pixels of sum_c w[c]*(normalize(f0) - normalize(f1))^2.  workspace: N*256 floats.  C <= 512
int odvae_lpips_distance_f32(const float* f0, const float* f1, const float* w, float* out, int N, int HW, int C,
                             void* workspace, size_t workspace_bytes, void* stream) {
  ODVAE_CHECK_ARG(f0 && f1 && w && out && N > 0 && HW > 0 && C > 0 && C <= 64 * MAXCPL, "lpips_distance: need C <= %d", 64 * MAXCPL);
  const int nblk = (int)std::min<int64_t>(256, std::max<int64_t>(1, HW / 4));
  if (!workspace || workspace_bytes < (size_t)N * nblk * sizeof(float)) {
    odvae_set_error("lpips_distance: needs %zu workspace bytes", (size_t)N * nblk * sizeof(float));
    return ODVAE_ERR_WORKSPACE;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  float* part = static_cast<float*>(workspace);
  hipLaunchKernelGGL(lpips_distance_kernel, dim3(nblk, N), dim3(256), 0, st, f0, f1, w, HW, C, part);
  hipLaunchKernelGGL(lpips_mean_final_kernel, dim3(N), dim3(64), 0, st, part, nblk, 1.f / (float)HW, out);
  ODVAE_LAUNCH_CHECK("lpips_distance");
  return ODVAE_OK;
}

int odvae_lpips_distance_bwd_f32(const float* f0, const float* f1, const float* w, const float* g, float* df1,
                                 int N, int HW, int C, void* stream) {
  ODVAE_CHECK_ARG(f0 && f1 && w && g && df1 && N > 0 && HW > 0 && C > 0 && C <= 64 * MAXCPL, "lpips_distance_bwd: bad arguments");
  const int nblk = (int)std::min<int64_t>(2048, std::max<int64_t>(1, HW / 4));
  hipLaunchKernelGGL(lpips_distance_bwd_kernel, dim3(nblk, N), dim3(256), 0, static_cast<hipStream_t>(stream), f0, f1, w, g, HW, C, df1);
  ODVAE_LAUNCH_CHECK("lpips_distance_bwd");
  return ODVAE_OK;
}

}  // extern "C"
