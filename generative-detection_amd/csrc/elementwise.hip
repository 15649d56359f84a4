// HBM-bound streaming / reduction kernels of the OD-VAE training step, f32, gfx950.
// All are float4-vectorised where the layout allows, wave64 shuffle reductions, fixed-order
// (deterministic) second stages.  Reference call sites are cited per kernel.
#include "common.h"

namespace {

int grid_1d(int64_t items, int cap = 8192) { return (int)std::min<int64_t>(std::max<int64_t>(ceil_div64(items, 256), 1), cap); }

// block-wide sum of one float per thread (256 threads); result valid in thread 0
__device__ __forceinline__ float block_sum(float v, float* sh) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  float r = 0.f;
  if (threadIdx.x == 0) for (int w = 0; w < (int)(blockDim.x >> 6); ++w) r += sh[w];
  __syncthreads();
  return r;
}

// ---- row softmax ---------------------------------------------------------------------------------
// AttnBlock.forward: w_ = softmax(bmm(q,k) * C^-0.5, dim=keys)  ([UPSTREAM] ldm .../model.py AttnBlock)
// one block per row; cols % 4 == 0.  y may alias x.
// Folded-softmax fallbacks taken since the library was loaded (odvae_device_health): a predicated softmax launch that found its flag set
// counts itself once.  Not an error -- the fallback is exact -- but a run that takes it often is paying three extra passes per block.
__device__ unsigned g_attn_fallbacks = 0;
typedef __attribute__((address_space(1))) unsigned ew_gu32_t;
__device__ __forceinline__ void count_fallback(const int* pred) {
  if (pred && blockIdx.x == 0 && threadIdx.x == 0)
    __hip_atomic_fetch_add((ew_gu32_t*)&g_attn_fallbacks, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// pred (nullable): the launch does nothing unless *pred != 0; ones (nullable): ones[row] = 1 for every row done (the folded-softmax
// fallback of ops.attention: the probabilities it writes are normalised, their row factor is 1)
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* x, float* y, int64_t rows, int cols, float scale,
                                                           const int* pred, float* ones) {
  __shared__ float sh[8];
  if (pred && *pred == 0) return;
  count_fallback(pred);
  for (int64_t row = blockIdx.x; row < rows; row += gridDim.x) {
    if (ones && threadIdx.x == 0) ones[row] = 1.f;
    const float* xr = x + row * cols;
    float* yr = y + row * cols;
    float mx = -INFINITY;
    for (int c = threadIdx.x * 4; c < cols; c += 1024) {
      const float4 v = *reinterpret_cast<const float4*>(xr + c);
      mx = fmaxf(mx, fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)));
    }
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
    __syncthreads();
    // softmax(scale*x): the max of scale*x is scale*max(x) for scale > 0
    float sum = 0.f;
    for (int c = threadIdx.x * 4; c < cols; c += 1024) {
      const float4 v = *reinterpret_cast<const float4*>(xr + c);
      sum += __expf((v.x - mx) * scale) + __expf((v.y - mx) * scale) + __expf((v.z - mx) * scale) + __expf((v.w - mx) * scale);
    }
    sum = wave_sum(sum);
    if ((threadIdx.x & 63) == 0) sh[4 + (threadIdx.x >> 6)] = sum;
    __syncthreads();
    const float inv = 1.f / (sh[4] + sh[5] + sh[6] + sh[7]);
    for (int c = threadIdx.x * 4; c < cols; c += 1024) {
      const float4 v = *reinterpret_cast<const float4*>(xr + c);
      float4 o;
      o.x = __expf((v.x - mx) * scale) * inv; o.y = __expf((v.y - mx) * scale) * inv;
      o.z = __expf((v.z - mx) * scale) * inv; o.w = __expf((v.w - mx) * scale) * inv;
      *reinterpret_cast<float4*>(yr + c) = o;
    }
    __syncthreads();
  }
}

// dS = scale * P * (dP - sum_j dP_j P_j);  ds may alias dp
__global__ __launch_bounds__(256) void softmax_rows_bwd_kernel(const float* p, const float* dp, float* ds, int64_t rows, int cols, float scale) {
  __shared__ float sh[4];
  for (int64_t row = blockIdx.x; row < rows; row += gridDim.x) {
    const float* pr = p + row * cols;
    const float* dr = dp + row * cols;
    float* sr = ds + row * cols;
    float dot = 0.f;
    for (int c = threadIdx.x * 4; c < cols; c += 1024) {
      const float4 a = *reinterpret_cast<const float4*>(pr + c);
      const float4 b = *reinterpret_cast<const float4*>(dr + c);
      dot += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
    }
    dot = wave_sum(dot);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = dot;
    __syncthreads();
    dot = sh[0] + sh[1] + sh[2] + sh[3];
    for (int c = threadIdx.x * 4; c < cols; c += 1024) {
      const float4 a = *reinterpret_cast<const float4*>(pr + c);
      const float4 b = *reinterpret_cast<const float4*>(dr + c);
      float4 o;
      o.x = scale * a.x * (b.x - dot); o.y = scale * a.y * (b.y - dot);
      o.z = scale * a.z * (b.z - dot); o.w = scale * a.w * (b.w - dot);
      *reinterpret_cast<float4*>(sr + c) = o;
    }
    __syncthreads();
  }
}

// Register-resident forms: the whole row (NV float4 per thread, cols <= 1024*NV) is read ONCE, reduced, and written
// once: 2 HBM passes instead of 4 (forward) and 3 instead of 5 (backward).  Used whenever the row fits.
template <int NV>
__global__ __launch_bounds__(256) void softmax_rows_reg_kernel(const float* x, float* y, int64_t rows, int cols, float scale,
                                                               const int* pred, float* ones) {
  __shared__ float sh[8];
  if (pred && *pred == 0) return;
  count_fallback(pred);
  for (int64_t row = blockIdx.x; row < rows; row += gridDim.x) {
    if (ones && threadIdx.x == 0) ones[row] = 1.f;
    const float* xr = x + row * cols;
    float4 v[NV];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (threadIdx.x + 256 * i) * 4;
      // branch-free: clamp the address, select afterwards (a guarded load makes hipcc wait vmcnt(0) per load)
      const float4 ld = *reinterpret_cast<const float4*>(xr + min(c, cols - 4));
      v[i] = c < cols ? ld : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
      mx = fmaxf(mx, fmaxf(fmaxf(v[i].x, v[i].y), fmaxf(v[i].z, v[i].w)));
    }
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      v[i].x = __expf((v[i].x - mx) * scale); v[i].y = __expf((v[i].y - mx) * scale);
      v[i].z = __expf((v[i].z - mx) * scale); v[i].w = __expf((v[i].w - mx) * scale);
      sum += v[i].x + v[i].y + v[i].z + v[i].w;      // exp(-inf) = 0 for the padding lanes
    }
    sum = wave_sum(sum);
    if ((threadIdx.x & 63) == 0) sh[4 + (threadIdx.x >> 6)] = sum;
    __syncthreads();
    const float inv = 1.f / (sh[4] + sh[5] + sh[6] + sh[7]);
    float* yr = y + row * cols;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (threadIdx.x + 256 * i) * 4;
      if (c < cols) *reinterpret_cast<float4*>(yr + c) = make_float4(v[i].x * inv, v[i].y * inv, v[i].z * inv, v[i].w * inv);
    }
    __syncthreads();
  }
}

template <int NV>
__global__ __launch_bounds__(256) void softmax_rows_bwd_reg_kernel(const float* p, const float* dp, float* ds, int64_t rows, int cols, float scale) {
  __shared__ float sh[4];
  for (int64_t row = blockIdx.x; row < rows; row += gridDim.x) {
    const float* pr = p + row * cols;
    const float* dr = dp + row * cols;
    float4 a[NV], b[NV];
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (threadIdx.x + 256 * i) * 4;
      const float4 la = *reinterpret_cast<const float4*>(pr + min(c, cols - 4));
      const float4 lb = *reinterpret_cast<const float4*>(dr + min(c, cols - 4));
      a[i] = c < cols ? la : make_float4(0.f, 0.f, 0.f, 0.f);
      b[i] = c < cols ? lb : make_float4(0.f, 0.f, 0.f, 0.f);
      dot += a[i].x * b[i].x + a[i].y * b[i].y + a[i].z * b[i].z + a[i].w * b[i].w;
    }
    dot = wave_sum(dot);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = dot;
    __syncthreads();
    dot = sh[0] + sh[1] + sh[2] + sh[3];
    float* sr = ds + row * cols;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (threadIdx.x + 256 * i) * 4;
      if (c < cols)
        *reinterpret_cast<float4*>(sr + c) = make_float4(scale * a[i].x * (b[i].x - dot), scale * a[i].y * (b[i].y - dot),
                                                         scale * a[i].z * (b[i].z - dot), scale * a[i].w * (b[i].w - dot));
    }
    __syncthreads();
  }
}

// ---- backward of nearest 2x upsample: dX[n][y][x][c] = sum of the 2x2 block of dU --------------------
// (Upsample.forward: F.interpolate(scale_factor=2.0, mode="nearest"), [UPSTREAM] ldm .../model.py)
__global__ __launch_bounds__(256) void upsample2x_bwd_kernel(const float* __restrict__ du, float* __restrict__ dx,
                                                             int N, int H, int W, int C) {
  const int quads = C / 4;
  const int64_t total = (int64_t)N * H * W * quads;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int q = (int)(idx % quads);
    int64_t r = idx / quads;
    const int xw = (int)(r % W); r /= W;
    const int yh = (int)(r % H); const int n = (int)(r / H);
    const float* s = du + (((int64_t)n * 2 * H + 2 * yh) * 2 * W + 2 * xw) * C + 4 * q;
    const float4 a = *reinterpret_cast<const float4*>(s);
    const float4 b = *reinterpret_cast<const float4*>(s + C);
    const float4 c = *reinterpret_cast<const float4*>(s + (int64_t)2 * W * C);
    const float4 d = *reinterpret_cast<const float4*>(s + (int64_t)2 * W * C + C);
    *reinterpret_cast<float4*>(dx + idx * 4) =
        make_float4(a.x + b.x + c.x + d.x, a.y + b.y + c.y + d.y, a.z + b.z + c.z + d.z, a.w + b.w + c.w + d.w);
  }
}

// ---- batch min/max rescale: 2(x-min)/(max-min)-1  (src/models/autoencoder.py:434-436) -----------------
__global__ __launch_bounds__(256) void minmax_partial_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ part) {
  __shared__ float sh[8];
  float mn = INFINITY, mx = -INFINITY;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float v = x[i];
    mn = fminf(mn, v); mx = fmaxf(mx, v);
  }
  mn = wave_min(mn); mx = wave_max(mx);
  if ((threadIdx.x & 63) == 0) { sh[threadIdx.x >> 6] = mn; sh[4 + (threadIdx.x >> 6)] = mx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    part[2 * blockIdx.x] = fminf(fminf(sh[0], sh[1]), fminf(sh[2], sh[3]));
    part[2 * blockIdx.x + 1] = fmaxf(fmaxf(sh[4], sh[5]), fmaxf(sh[6], sh[7]));
  }
}
__global__ void minmax_final_kernel(const float* __restrict__ part, int nblocks, float* __restrict__ out) {
  float mn = INFINITY, mx = -INFINITY;
  for (int i = threadIdx.x; i < nblocks; i += 64) { mn = fminf(mn, part[2 * i]); mx = fmaxf(mx, part[2 * i + 1]); }
  mn = wave_min(mn); mx = wave_max(mx);
  if (threadIdx.x == 0) { out[0] = mn; out[1] = mx; }
}
// x: NCHW [N][C][HW] -> y: NHWC [N][HW][C], y = 2(x-min)/(max-min)-1 (same operation order as the reference)
__global__ __launch_bounds__(256) void rescale_nchw_to_nhwc_kernel(const float* __restrict__ x, const float* __restrict__ mm,
                                                                   float* __restrict__ y, int N, int C, int HW) {
  const float mn = mm[0], range = mm[1] - mm[0];
  const int64_t total = (int64_t)N * HW * C;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(idx % C);
    const int64_t r = idx / C;
    const int hw = (int)(r % HW); const int n = (int)(r / HW);
    const float v = x[((int64_t)n * C + c) * HW + hw];
    y[idx] = 2.f * (v - mn) / range - 1.f;
  }
}

// ---- diagonal Gaussian posterior (src/util/distributions.py:5-41 + [UPSTREAM] ldm distributions) -----
// moments: NHWC [N][HW][2*Cz]; mean = channels [0,Cz), logvar = channels [Cz,2Cz) clamped to [-30,20].
// z = mean + exp(0.5*logvar)*eps  (eps: [N][HW][Cz], injected by the host)
__global__ __launch_bounds__(256) void gaussian_sample_kernel(const float* __restrict__ mom, const float* __restrict__ eps,
                                                              float* __restrict__ z, int64_t npix, int Cz) {
  const int64_t total = npix * Cz;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(idx % Cz); const int64_t px = idx / Cz;
    const float mu = mom[px * 2 * Cz + c];
    const float lv = fminf(fmaxf(mom[px * 2 * Cz + Cz + c], -30.f), 20.f);
    z[idx] = mu + expf(0.5f * lv) * eps[idx];
  }
}
// kl[n] = 0.5 * sum_{hw,c} (mean^2 + var - 1 - logvar); one block per sample
__global__ __launch_bounds__(256) void gaussian_kl_kernel(const float* __restrict__ mom, float* __restrict__ kl, int HW, int Cz) {
  __shared__ float sh[4];
  const int n = blockIdx.x;
  const int64_t per = (int64_t)HW * Cz;
  float s = 0.f;
  for (int64_t i = threadIdx.x; i < per; i += blockDim.x) {
    const int c = (int)(i % Cz); const int64_t px = (int64_t)n * HW + i / Cz;
    const float mu = mom[px * 2 * Cz + c];
    const float lv = fminf(fmaxf(mom[px * 2 * Cz + Cz + c], -30.f), 20.f);
    s += mu * mu + expf(lv) - 1.f - lv;
  }
  s = block_sum(s, sh);
  if (threadIdx.x == 0) kl[n] = 0.5f * s;
}
// dmom from dz (may be null), eps (null iff dz null) and dkl[n] (may be null)
__global__ __launch_bounds__(256) void gaussian_bwd_kernel(const float* __restrict__ mom, const float* __restrict__ eps,
                                                           const float* __restrict__ dz, const float* __restrict__ dkl,
                                                           float* __restrict__ dmom, int N, int HW, int Cz) {
  const int64_t total = (int64_t)N * HW * Cz;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(idx % Cz); const int64_t px = idx / Cz;
    const int n = (int)(px / HW);
    const float mu = mom[px * 2 * Cz + c];
    const float lraw = mom[px * 2 * Cz + Cz + c];
    const float lv = fminf(fmaxf(lraw, -30.f), 20.f);
    const bool pass = lraw >= -30.f && lraw <= 20.f;  // d clamp / d x
    float dmu = 0.f, dlv = 0.f;
    if (dz) { const float g = dz[idx]; dmu += g; dlv += g * eps[idx] * 0.5f * expf(0.5f * lv); }
    if (dkl) { const float g = dkl[n]; dmu += g * mu; dlv += g * 0.5f * (expf(lv) - 1.f); }
    dmom[px * 2 * Cz + c] = dmu;
    dmom[px * 2 * Cz + Cz + c] = pass ? dlv : 0.f;
  }
}

// ---- masked L1 reconstruction term (src/modules/losses/contperceptual.py:134-145,252-257) ------------
// s[n] = sum_{hw,c} | x*m - xr*m |, x/xr NHWC [N][HW][C], m [N][HW] or null.  Stage 1: partial[n][blk]
__global__ __launch_bounds__(256) void l1_partial_kernel(const float* __restrict__ x, const float* __restrict__ xr,
                                                         const float* __restrict__ m, float* __restrict__ part, int HW, int C) {
  __shared__ float sh[4];
  const int n = blockIdx.y;
  const int64_t per = (int64_t)HW * C;
  const float* xa = x + n * per; const float* xb = xr + n * per;
  const float* mk = m ? m + (int64_t)n * HW : nullptr;
  float s = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < per; i += (int64_t)gridDim.x * blockDim.x) {
    const float w = mk ? mk[i / C] : 1.f;
    s += fabsf(xa[i] * w - xb[i] * w);
  }
  s = block_sum(s, sh);
  if (threadIdx.x == 0) part[(int64_t)n * gridDim.x + blockIdx.x] = s;
}
__global__ void rowsum_final_kernel(const float* __restrict__ part, int nblk, float* __restrict__ out) {
  const int n = blockIdx.x;
  double s = 0.0;
  for (int i = threadIdx.x; i < nblk; i += 64) s += (double)part[(int64_t)n * nblk + i];
  s = wave_sum_d(s);
  if (threadIdx.x == 0) out[n] = (float)s;
}
// dxr = g[n] * sign(xr*m - x*m) * m
__global__ __launch_bounds__(256) void l1_bwd_kernel(const float* __restrict__ x, const float* __restrict__ xr,
                                                     const float* __restrict__ m, const float* __restrict__ g,
                                                     float* __restrict__ dxr, int N, int HW, int C) {
  const int64_t per = (int64_t)HW * C, total = per * N;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int n = (int)(idx / per);
    const float w = m ? m[idx / C] : 1.f;
    const float d = xr[idx] * w - x[idx] * w;
    const float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
    dxr[idx] = g[n] * sg * w;
  }
}

// ---- column sums (bias gradient of the 1x1 convolutions): out[c] = sum_rows x[row][c] ------------------
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ x, int64_t rows, int C,
                                                             int rows_per_block, float* __restrict__ part) {
  // thread t handles column (t % C) for C <= 256 (several row lanes), else strides over columns
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = std::min<int64_t>(rows, r0 + rows_per_block);
  if ((C & 3) == 0 && C <= 1024 && (((uintptr_t)x) & 15) == 0) {
    // channel quads: thread = (quad, row lane); 16-byte loads, four rows in flight per thread (the scalar form below runs
    // one dependent 4-byte load at a time: 2 TB/s on the 1x1 convolutions' 0.1-0.5 GB inputs)
    __shared__ float4 sh4[256];
    const int Q = C >> 2, lanes = 256 / Q;
    const int q = threadIdx.x % Q, rl = threadIdx.x / Q;
    float4 acc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (rl < lanes) {
      const float4* xq = reinterpret_cast<const float4*>(x) + q;
      int64_t r = r0 + rl;
      for (; r + 3 * lanes < r1; r += 4 * lanes) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = xq[(r + (int64_t)u * lanes) * Q];
#pragma unroll
        for (int u = 0; u < 4; ++u) { acc[u].x += v[u].x; acc[u].y += v[u].y; acc[u].z += v[u].z; acc[u].w += v[u].w; }
      }
      for (; r < r1; r += lanes) { const float4 v = xq[r * Q]; acc[0].x += v.x; acc[0].y += v.y; acc[0].z += v.z; acc[0].w += v.w; }
    }
    sh4[threadIdx.x] = make_float4((acc[0].x + acc[1].x) + (acc[2].x + acc[3].x), (acc[0].y + acc[1].y) + (acc[2].y + acc[3].y),
                                   (acc[0].z + acc[1].z) + (acc[2].z + acc[3].z), (acc[0].w + acc[1].w) + (acc[2].w + acc[3].w));
    __syncthreads();
    if (threadIdx.x < Q) {
      float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int k = 0; k < lanes; ++k) { const float4 v = sh4[k * Q + threadIdx.x]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
      *reinterpret_cast<float4*>(part + (int64_t)blockIdx.x * C + 4 * threadIdx.x) = t;
    }
  } else if (C <= 256) {
    __shared__ float sh[256];
    const int lanes = 256 / C;
    const int c = threadIdx.x % C, rl = threadIdx.x / C;
    float s = 0.f;
    if (rl < lanes) for (int64_t r = r0 + rl; r < r1; r += lanes) s += x[r * C + c];
    sh[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x < C) {
      float t = 0.f;
      for (int k = 0; k < lanes; ++k) t += sh[k * C + threadIdx.x];
      part[(int64_t)blockIdx.x * C + threadIdx.x] = t;
    }
  } else {
    for (int c = threadIdx.x; c < C; c += 256) {
      float s = 0.f;
      for (int64_t r = r0; r < r1; ++r) s += x[r * C + c];
      part[(int64_t)blockIdx.x * C + c] = s;
    }
  }
}
// one wavefront per column: lanes stride over the partial blocks, f64 wave reduction (fixed order => deterministic)
__global__ void colsum_final_kernel(const float* __restrict__ part, int nblk, int C, float* __restrict__ out) {
  const int c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (c >= C) return;
  double s = 0.0;
  for (int b = threadIdx.x & 63; b < nblk; b += 64) s += (double)part[(int64_t)b * C + c];
  s = wave_sum_d(s);
  if ((threadIdx.x & 63) == 0) out[c] = (float)s;
}

// ---- optimizer: global grad norm + Adam over one flat arena (src/models/autoencoder.py:365-377, yaml:140)
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ g, int64_t n, float* __restrict__ part) {
  __shared__ float sh[4];
  float s = 0.f;
  const int64_t n4 = n / 4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 v = reinterpret_cast<const float4*>(g)[i];
    s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  if (blockIdx.x == 0) for (int64_t i = n4 * 4 + threadIdx.x; i < n; i += blockDim.x) s += g[i] * g[i];
  s = block_sum(s, sh);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}
// out[0] = total norm, out[1] = clip coefficient min(1, max_norm/(norm+1e-6)) (torch clip_grad_norm_)
__global__ void norm_final_kernel(const float* __restrict__ part, int nblk, float max_norm, float* __restrict__ out) {
  double s = 0.0;
  for (int i = threadIdx.x; i < nblk; i += 64) s += (double)part[i];
  s = wave_sum_d(s);
  if (threadIdx.x == 0) {
    const float nrm = (float)sqrt(s);
    out[0] = nrm;
    float coef = 1.f;
    if (max_norm > 0.f) { coef = max_norm / (nrm + 1e-6f); if (coef > 1.f) coef = 1.f; }
    out[1] = coef;
  }
}
// torch.optim.Adam (no weight decay, no amsgrad); grads are scaled by *clip (device scalar) when non-null
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                   float lr, float beta1, float beta2, float eps,
                                                   float bc1, float bc2_sqrt, const float* __restrict__ clip) {
  const float cs = clip ? clip[1] : 1.f;
  const float step = lr / bc1;
  const int64_t n4 = n / 4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    float4 pv = reinterpret_cast<float4*>(p)[i];
    const float4 gv = reinterpret_cast<const float4*>(g)[i];
    float4 mv = reinterpret_cast<float4*>(m)[i];
    float4 vv = reinterpret_cast<float4*>(v)[i];
    float pp[4] = {pv.x, pv.y, pv.z, pv.w}, gg[4] = {gv.x, gv.y, gv.z, gv.w};
    float mm[4] = {mv.x, mv.y, mv.z, mv.w}, ww[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float gr = gg[j] * cs;
      mm[j] = beta1 * mm[j] + (1.f - beta1) * gr;
      ww[j] = beta2 * ww[j] + (1.f - beta2) * gr * gr;
      pp[j] -= step * mm[j] / (sqrtf(ww[j]) / bc2_sqrt + eps);
    }
    reinterpret_cast<float4*>(p)[i] = make_float4(pp[0], pp[1], pp[2], pp[3]);
    reinterpret_cast<float4*>(m)[i] = make_float4(mm[0], mm[1], mm[2], mm[3]);
    reinterpret_cast<float4*>(v)[i] = make_float4(ww[0], ww[1], ww[2], ww[3]);
  }
  if (blockIdx.x == 0)
    for (int64_t i = n4 * 4 + threadIdx.x; i < n; i += blockDim.x) {
      const float gr = g[i] * cs;
      const float mi = beta1 * m[i] + (1.f - beta1) * gr;
      const float vi = beta2 * v[i] + (1.f - beta2) * gr * gr;
      m[i] = mi; v[i] = vi;
      p[i] -= step * mi / (sqrtf(vi) / bc2_sqrt + eps);
    }
}

// ---- layout: NHWC -> NCHW copy (reconstructions handed back to NCHW callers) ---------------------------
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int C, int HW) {
  const int64_t total = (int64_t)N * C * HW;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int hw = (int)(idx % HW);
    const int64_t r = idx / HW;
    const int c = (int)(r % C); const int n = (int)(r / C);
    y[idx] = x[((int64_t)n * HW + hw) * C + c];
  }
}

// ---- y = x * m[n][hw] (mask_2d_bbox products, contperceptual.py:252-255); its own backward with x = dy ----
__global__ __launch_bounds__(256) void mul_mask_kernel(const float* __restrict__ x, const float* __restrict__ m,
                                                       float* __restrict__ y, int64_t npix, int C) {
  const int64_t total = npix * C;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x)
    y[idx] = x[idx] * m[idx / C];
}

// ---- latent mixing: out = z * mask + add (dropout on z_obj, + N(0,1) noise, + enc_pose; autoencoder.py:233-253)
__global__ __launch_bounds__(256) void latent_combine_kernel(const float* __restrict__ z, const float* __restrict__ mask,
                                                             const float* __restrict__ add, float* __restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float v = z[i];
    if (mask) v *= mask[i];
    if (add) v += add[i];
    out[i] = v;
  }
}

}  // namespace

extern "C" {

int odvae_mul_mask_f32(const float* x, const float* mask, float* y, int64_t npix, int C, void* stream) {
  ODVAE_CHECK_ARG(x && mask && y && npix > 0 && C > 0, "mul_mask: bad arguments");
  hipLaunchKernelGGL(mul_mask_kernel, dim3(grid_1d(npix * C)), dim3(256), 0, static_cast<hipStream_t>(stream), x, mask, y, npix, C);
  ODVAE_LAUNCH_CHECK("mul_mask");
  return ODVAE_OK;
}

int odvae_latent_combine_f32(const float* z, const float* mask, const float* add, float* out, int64_t n, void* stream) {
  ODVAE_CHECK_ARG(z && out && n > 0, "latent_combine: bad arguments");
  hipLaunchKernelGGL(latent_combine_kernel, dim3(grid_1d(n)), dim3(256), 0, static_cast<hipStream_t>(stream), z, mask, add, out, n);
  ODVAE_LAUNCH_CHECK("latent_combine");
  return ODVAE_OK;
}

static int softmax_rows_impl(const float* x, float* y, int64_t rows, int cols, float scale, const int* pred, float* ones, void* stream) {
  ODVAE_CHECK_ARG(x && y && rows > 0 && cols > 0 && cols % 4 == 0, "softmax_rows: need cols %% 4 == 0 (rows=%lld cols=%d)", (long long)rows, cols);
  ODVAE_CHECK_ARG(scale > 0.f, "softmax_rows: scale must be positive");
  // (under a predicate: a small grid that walks the rows, so that finding out there is nothing to do is a 2 048-block launch)
  const dim3 grid((unsigned)std::min<int64_t>(rows, pred ? 2048 : 65536 * 4)), block(256);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (cols <= 1024)       hipLaunchKernelGGL((softmax_rows_reg_kernel<1>), grid, block, 0, st, x, y, rows, cols, scale, pred, ones);
  else if (cols <= 2048)  hipLaunchKernelGGL((softmax_rows_reg_kernel<2>), grid, block, 0, st, x, y, rows, cols, scale, pred, ones);
  else if (cols <= 4096)  hipLaunchKernelGGL((softmax_rows_reg_kernel<4>), grid, block, 0, st, x, y, rows, cols, scale, pred, ones);
  else if (cols <= 16384) hipLaunchKernelGGL((softmax_rows_reg_kernel<16>), grid, block, 0, st, x, y, rows, cols, scale, pred, ones);
  else                    hipLaunchKernelGGL(softmax_rows_kernel, grid, block, 0, st, x, y, rows, cols, scale, pred, ones);
  ODVAE_LAUNCH_CHECK("softmax_rows");
  return ODVAE_OK;
}

int odvae_softmax_rows_f32(const float* x, float* y, int64_t rows, int cols, float scale, void* stream) {
  return softmax_rows_impl(x, y, rows, cols, scale, nullptr, nullptr, stream);
}

// The same under a device-side predicate (nothing happens unless *pred != 0), also writing ones[row] = 1: the fallback of the folded
// attention softmax (odvae_gemm_exp_bound_f32 / odvae_gemm_rownorm_f32) where a row's bound was too loose.
int odvae_softmax_rows_pred_f32(const float* x, float* y, int64_t rows, int cols, float scale, const int* pred, float* ones, void* stream) {
  ODVAE_CHECK_ARG(pred && ones, "softmax_rows_pred: null predicate / row-factor array");
  return softmax_rows_impl(x, y, rows, cols, scale, pred, ones, stream);
}

// Device-side counter of the folded-softmax fallbacks; add > 0 is a TEST HOOK that bumps it (synchronises the device).  -1 on a HIP error.
int odvae_attn_softmax_fallbacks(int add) {
  unsigned v = 0;
  if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_attn_fallbacks), sizeof(v)) != hipSuccess) return -1;
  if (add > 0) {
    v += (unsigned)add;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_attn_fallbacks), &v, sizeof(v)) != hipSuccess) return -1;
  }
  return (int)v;
}

// ---- folded attention softmax: the per-row bound of the scores, and the backward's row dot product with dO / l beside it ----
// qkv [rows][3C] (q | k | v per token).  nq[row] = |q_row|, nk[row] = |k_row|: one wavefront per token.
__global__ __launch_bounds__(256) void attn_rownorm_kernel(const float* __restrict__ qkv, int64_t rows, int C,
                                                           float* __restrict__ nq, float* __restrict__ nk) {
  const int lane = threadIdx.x & 63;
  for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += (int64_t)gridDim.x * 4) {
    const float4* pq = reinterpret_cast<const float4*>(qkv + row * 3 * C);
    const float4* pk = reinterpret_cast<const float4*>(qkv + row * 3 * C + C);
    float a = 0.f, b = 0.f;
    for (int i = lane; i < C / 4; i += 64) {
      const float4 x = pq[i], y = pk[i];
      a += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
      b += y.x * y.x + y.y * y.y + y.z * y.z + y.w * y.w;
    }
    a = wave_sum(a); b = wave_sum(b);
    if (lane == 0) { nq[row] = sqrtf(a); nk[row] = sqrtf(b); }
  }
}
// one block per image: bound[i] = |q_i| * max_j |k_j| (>= q_i . k_j for every j, Cauchy-Schwarz; a hair above for the rounding of the
// norms), in place over nq; block 0 clears the fallback flag
__global__ __launch_bounds__(256) void attn_bound_kernel(float* __restrict__ nq, const float* __restrict__ nk, int T, int* flag) {
  __shared__ float sh[4];
  const int n = blockIdx.x;
  float m = 0.f;
  for (int i = threadIdx.x; i < T; i += 256) m = fmaxf(m, nk[(int64_t)n * T + i]);
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3])) * 1.000001f;
  for (int i = threadIdx.x; i < T; i += 256) nq[(int64_t)n * T + i] *= m;
  if (n == 0 && threadIdx.x == 0 && flag) *flag = 0;
}

// out[row] = a[row] . b[row] and as[row][:] = a[row][:] * rs[row]  (attention backward with P = E / l: D_i = dO_i . O_i, dO_i / l_i)
__global__ __launch_bounds__(256) void rowdot_scale_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ rs,
                                                           int64_t rows, int cols, float* __restrict__ out, float* __restrict__ as) {
  const int lane = threadIdx.x & 63;
  for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += (int64_t)gridDim.x * 4) {
    const float4* pa = reinterpret_cast<const float4*>(a + row * cols);
    const float4* pb = reinterpret_cast<const float4*>(b + row * cols);
    float4* po = reinterpret_cast<float4*>(as + row * cols);
    const float r = rs[row];
    float s = 0.f;
    for (int q = lane; q < cols / 4; q += 64) {
      const float4 x = pa[q], y = pb[q];
      s += x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
      po[q] = make_float4(x.x * r, x.y * r, x.z * r, x.w * r);
    }
    s = wave_sum(s);
    if (lane == 0) out[row] = s;
  }
}

// bound[N * T] for odvae_gemm_exp_bound_f32 from qkv [N][T][3C]; nk_scratch [N * T]; *flag (nullable) is cleared
int odvae_attn_row_bound_f32(const float* qkv, int N, int T, int C, float* bound, float* nk_scratch, int* flag, void* stream) {
  ODVAE_CHECK_ARG(qkv && bound && nk_scratch && N > 0 && T > 0 && C > 0 && C % 4 == 0, "attn_row_bound: bad arguments (C %% 4 == 0 needed)");
  ODVAE_CHECK_ARG(((uintptr_t)qkv & 15) == 0, "attn_row_bound: qkv must be 16-byte aligned");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int64_t rows = (int64_t)N * T;
  hipLaunchKernelGGL(attn_rownorm_kernel, dim3((unsigned)std::min<int64_t>(ceil_div64(rows, 4), 65536)), dim3(256), 0, st, qkv, rows, C, bound, nk_scratch);
  ODVAE_LAUNCH_CHECK("attn_rownorm");
  hipLaunchKernelGGL(attn_bound_kernel, dim3(N), dim3(256), 0, st, bound, nk_scratch, T, flag);
  ODVAE_LAUNCH_CHECK("attn_bound");
  return ODVAE_OK;
}

int odvae_rowdot_scale_f32(const float* a, const float* b, const float* row_scale, int64_t rows, int cols, float* out, float* a_scaled, void* stream) {
  ODVAE_CHECK_ARG(a && b && row_scale && out && a_scaled && rows > 0 && cols > 0 && cols % 4 == 0, "rowdot_scale: need cols %% 4 == 0");
  const dim3 grid((unsigned)std::min<int64_t>(ceil_div64(rows, 4), 65536)), block(256);
  hipLaunchKernelGGL(rowdot_scale_kernel, grid, block, 0, static_cast<hipStream_t>(stream), a, b, row_scale, rows, cols, out, a_scaled);
  ODVAE_LAUNCH_CHECK("rowdot_scale");
  return ODVAE_OK;
}

// out[row] = sum_c a[row][c] * b[row][c]: one wavefront per row, float4 lanes (attention backward: dO[i] . O[i])
__global__ __launch_bounds__(256) void rowdot_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                     int64_t rows, int cols, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += (int64_t)gridDim.x * 4) {
    const float4* pa = reinterpret_cast<const float4*>(a + row * cols);
    const float4* pb = reinterpret_cast<const float4*>(b + row * cols);
    float s = 0.f;
    for (int q = lane; q < cols / 4; q += 64) {
      const float4 x = pa[q], y = pb[q];
      s += x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
    }
    s = wave_sum(s);
    if (lane == 0) out[row] = s;
  }
}

int odvae_rowdot_f32(const float* a, const float* b, int64_t rows, int cols, float* out, void* stream) {
  ODVAE_CHECK_ARG(a && b && out && rows > 0 && cols > 0 && cols % 4 == 0, "rowdot: need cols %% 4 == 0");
  const dim3 grid((unsigned)std::min<int64_t>(ceil_div64(rows, 4), 65536)), block(256);
  hipLaunchKernelGGL(rowdot_kernel, grid, block, 0, static_cast<hipStream_t>(stream), a, b, rows, cols, out);
  ODVAE_LAUNCH_CHECK("rowdot");
  return ODVAE_OK;
}

int odvae_softmax_rows_bwd_f32(const float* p, const float* dp, float* ds, int64_t rows, int cols, float scale, void* stream) {
  ODVAE_CHECK_ARG(p && dp && ds && rows > 0 && cols > 0 && cols % 4 == 0, "softmax_rows_bwd: need cols %% 4 == 0");
  const dim3 grid((unsigned)std::min<int64_t>(rows, 65536 * 4)), block(256);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (cols <= 1024)       hipLaunchKernelGGL((softmax_rows_bwd_reg_kernel<1>), grid, block, 0, st, p, dp, ds, rows, cols, scale);
  else if (cols <= 2048)  hipLaunchKernelGGL((softmax_rows_bwd_reg_kernel<2>), grid, block, 0, st, p, dp, ds, rows, cols, scale);
  else if (cols <= 4096)  hipLaunchKernelGGL((softmax_rows_bwd_reg_kernel<4>), grid, block, 0, st, p, dp, ds, rows, cols, scale);
  else if (cols <= 8192)  hipLaunchKernelGGL((softmax_rows_bwd_reg_kernel<8>), grid, block, 0, st, p, dp, ds, rows, cols, scale);
  else                    hipLaunchKernelGGL(softmax_rows_bwd_kernel, grid, block, 0, st, p, dp, ds, rows, cols, scale);
  ODVAE_LAUNCH_CHECK("softmax_rows_bwd");
  return ODVAE_OK;
}

int odvae_upsample2x_bwd_f32(const float* du, float* dx, int N, int H, int W, int C, void* stream) {
  ODVAE_CHECK_ARG(du && dx && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "upsample2x_bwd: need C %% 4 == 0");
  hipLaunchKernelGGL(upsample2x_bwd_kernel, dim3(grid_1d((int64_t)N * H * W * (C / 4))), dim3(256), 0, static_cast<hipStream_t>(stream), du, dx, N, H, W, C);
  ODVAE_LAUNCH_CHECK("upsample2x_bwd");
  return ODVAE_OK;
}

// workspace: >= 2*1024+2 floats.  minmax_out[2] receives (min, max).
int odvae_rescale_minmax_f32(const float* x_nchw, float* y_nhwc, int N, int C, int HW, float* minmax_out,
                             void* workspace, size_t workspace_bytes, void* stream) {
  ODVAE_CHECK_ARG(x_nchw && y_nhwc && minmax_out && N > 0 && C > 0 && HW > 0, "rescale_minmax: bad arguments");
  const int nblk = 1024;
  if (!workspace || workspace_bytes < (size_t)2 * nblk * sizeof(float)) {
    odvae_set_error("rescale_minmax: needs %zu workspace bytes", (size_t)2 * nblk * sizeof(float));
    return ODVAE_ERR_WORKSPACE;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  float* part = static_cast<float*>(workspace);
  const int64_t n = (int64_t)N * C * HW;
  hipLaunchKernelGGL(minmax_partial_kernel, dim3(nblk), dim3(256), 0, st, x_nchw, n, part);
  hipLaunchKernelGGL(minmax_final_kernel, dim3(1), dim3(64), 0, st, part, nblk, minmax_out);
  hipLaunchKernelGGL(rescale_nchw_to_nhwc_kernel, dim3(grid_1d(n)), dim3(256), 0, st, x_nchw, minmax_out, y_nhwc, N, C, HW);
  ODVAE_LAUNCH_CHECK("rescale_minmax");
  return ODVAE_OK;
}

int odvae_gaussian_sample_f32(const float* moments, const float* eps, float* z, int N, int HW, int Cz, void* stream) {
  ODVAE_CHECK_ARG(moments && eps && z && N > 0 && HW > 0 && Cz > 0, "gaussian_sample: bad arguments");
  const int64_t npix = (int64_t)N * HW;
  hipLaunchKernelGGL(gaussian_sample_kernel, dim3(grid_1d(npix * Cz)), dim3(256), 0, static_cast<hipStream_t>(stream), moments, eps, z, npix, Cz);
  ODVAE_LAUNCH_CHECK("gaussian_sample");
  return ODVAE_OK;
}

int odvae_gaussian_kl_f32(const float* moments, float* kl, int N, int HW, int Cz, void* stream) {
  ODVAE_CHECK_ARG(moments && kl && N > 0 && HW > 0 && Cz > 0, "gaussian_kl: bad arguments");
  hipLaunchKernelGGL(gaussian_kl_kernel, dim3(N), dim3(256), 0, static_cast<hipStream_t>(stream), moments, kl, HW, Cz);
  ODVAE_LAUNCH_CHECK("gaussian_kl");
  return ODVAE_OK;
}

// dmoments (overwritten) from dz/eps (both null or both set) and dkl (may be null)
int odvae_gaussian_bwd_f32(const float* moments, const float* eps, const float* dz, const float* dkl,
                           float* dmoments, int N, int HW, int Cz, void* stream) {
  ODVAE_CHECK_ARG(moments && dmoments && N > 0 && HW > 0 && Cz > 0, "gaussian_bwd: bad arguments");
  ODVAE_CHECK_ARG((dz == nullptr) == (eps == nullptr), "gaussian_bwd: dz and eps must both be set or both null");
  hipLaunchKernelGGL(gaussian_bwd_kernel, dim3(grid_1d((int64_t)N * HW * Cz)), dim3(256), 0, static_cast<hipStream_t>(stream), moments, eps, dz, dkl, dmoments, N, HW, Cz);
  ODVAE_LAUNCH_CHECK("gaussian_bwd");
  return ODVAE_OK;
}

// out[n] = sum |x*m - xr*m| over (hw, c).  workspace: N*256 floats
int odvae_l1_masked_sum_f32(const float* x, const float* xr, const float* mask, float* out, int N, int HW, int C,
                            void* workspace, size_t workspace_bytes, void* stream) {
  ODVAE_CHECK_ARG(x && xr && out && N > 0 && HW > 0 && C > 0, "l1_masked_sum: bad arguments");
  const int nblk = 256;
  if (!workspace || workspace_bytes < (size_t)N * nblk * sizeof(float)) {
    odvae_set_error("l1_masked_sum: needs %zu workspace bytes", (size_t)N * nblk * sizeof(float));
    return ODVAE_ERR_WORKSPACE;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  float* part = static_cast<float*>(workspace);
  hipLaunchKernelGGL(l1_partial_kernel, dim3(nblk, N), dim3(256), 0, st, x, xr, mask, part, HW, C);
  hipLaunchKernelGGL(rowsum_final_kernel, dim3(N), dim3(64), 0, st, part, nblk, out);
  ODVAE_LAUNCH_CHECK("l1_masked_sum");
  return ODVAE_OK;
}

int odvae_l1_masked_bwd_f32(const float* x, const float* xr, const float* mask, const float* g, float* dxr,
                            int N, int HW, int C, void* stream) {
  ODVAE_CHECK_ARG(x && xr && g && dxr && N > 0 && HW > 0 && C > 0, "l1_masked_bwd: bad arguments");
  hipLaunchKernelGGL(l1_bwd_kernel, dim3(grid_1d((int64_t)N * HW * C)), dim3(256), 0, static_cast<hipStream_t>(stream), x, xr, mask, g, dxr, N, HW, C);
  ODVAE_LAUNCH_CHECK("l1_masked_bwd");
  return ODVAE_OK;
}

size_t odvae_colsum_workspace_bytes(int64_t rows, int C) {
  const int nblk = (int)std::min<int64_t>(std::max<int64_t>(rows / 64, 1), 1024);
  return (size_t)nblk * C * sizeof(float);
}

int odvae_colsum_f32(const float* x, int64_t rows, int C, float* out, void* workspace, size_t workspace_bytes, void* stream) {
  ODVAE_CHECK_ARG(x && out && rows > 0 && C > 0, "colsum: bad arguments");
  const int nblk = (int)std::min<int64_t>(std::max<int64_t>(rows / 64, 1), 1024);
  if (!workspace || workspace_bytes < (size_t)nblk * C * sizeof(float)) {
    odvae_set_error("colsum: needs %zu workspace bytes", (size_t)nblk * C * sizeof(float));
    return ODVAE_ERR_WORKSPACE;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  float* part = static_cast<float*>(workspace);
  const int rpb = (int)ceil_div64(rows, nblk);
  const int nb = (int)ceil_div64(rows, rpb);
  hipLaunchKernelGGL(colsum_partial_kernel, dim3(nb), dim3(256), 0, st, x, rows, C, rpb, part);
  hipLaunchKernelGGL(colsum_final_kernel, dim3(ceil_div(C, 4)), dim3(256), 0, st, part, nb, C, out);
  ODVAE_LAUNCH_CHECK("colsum");
  return ODVAE_OK;
}

// out[0] = ||g||_2, out[1] = min(1, max_norm/(norm+1e-6)) (1 when max_norm <= 0).  workspace: 1024 floats
int odvae_grad_norm_f32(const float* g, int64_t n, float max_norm, float* out, void* workspace, size_t workspace_bytes, void* stream) {
  ODVAE_CHECK_ARG(g && out && n > 0, "grad_norm: bad arguments");
  ODVAE_CHECK_ARG(((uintptr_t)g & 15) == 0, "grad_norm: g must be 16-byte aligned");
  const int nblk = 1024;
  if (!workspace || workspace_bytes < (size_t)nblk * sizeof(float)) {
    odvae_set_error("grad_norm: needs %zu workspace bytes", (size_t)nblk * sizeof(float));
    return ODVAE_ERR_WORKSPACE;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  float* part = static_cast<float*>(workspace);
  hipLaunchKernelGGL(sumsq_partial_kernel, dim3(nblk), dim3(256), 0, st, g, n, part);
  hipLaunchKernelGGL(norm_final_kernel, dim3(1), dim3(64), 0, st, part, nblk, max_norm, out);
  ODVAE_LAUNCH_CHECK("grad_norm");
  return ODVAE_OK;
}

// one Adam step over a flat arena; step >= 1; clip = device pointer written by odvae_grad_norm_f32 or null
int odvae_adam_step_f32(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                        float eps, int step, const float* clip, void* stream) {
  ODVAE_CHECK_ARG(p && g && m && v && n > 0 && step >= 1, "adam_step: bad arguments");
  ODVAE_CHECK_ARG((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0, "adam_step: arenas must be 16-byte aligned");
  const float bc1 = (float)(1.0 - pow((double)beta1, (double)step));
  const float bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step));
  hipLaunchKernelGGL(adam_kernel, dim3(grid_1d(n / 4 + 1)), dim3(256), 0, static_cast<hipStream_t>(stream), p, g, m, v, n, lr, beta1, beta2, eps, bc1, bc2_sqrt, clip);
  ODVAE_LAUNCH_CHECK("adam_step");
  return ODVAE_OK;
}

int odvae_nhwc_to_nchw_f32(const float* x, float* y, int N, int C, int HW, void* stream) {
  ODVAE_CHECK_ARG(x && y && N > 0 && C > 0 && HW > 0, "nhwc_to_nchw: bad arguments");
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(grid_1d((int64_t)N * C * HW)), dim3(256), 0, static_cast<hipStream_t>(stream), x, y, N, C, HW);
  ODVAE_LAUNCH_CHECK("nhwc_to_nchw");
  return ODVAE_OK;
}

}  // extern "C"
