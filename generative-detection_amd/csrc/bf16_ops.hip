// HBM-bound pieces of the bf16 path (BASELINE.json configs[4]): GroupNorm(32, eps 1e-6) + swish forward / backward with bf16
// activations and f32 statistics ([UPSTREAM] ldm Normalize / nonlinearity; under torch.autocast group_norm itself runs in f32, so
// "read bf16, compute f32, round once on the way out" is the same arithmetic), dtype hand-offs at the f32 ends of the network
// (input image, latent, reconstruction), the 2x2 sum-pool of the Upsample data gradient and the bias-gradient column sum.
#include "bf16_common.h"

namespace {

#include "gn_finalize.h"

struct GnB {
  int N, HW, C, G, cpg, octs, pix_per_pass, chunks, pix_per_chunk;
};

// 1 / (1 + e^-u) with v_rcp_f32 (1 ulp) instead of an IEEE division (ten instructions): these kernels carry 18-30 vector instructions per
// element beside their loads, and the bf16 ones move two elements per 4 bytes -- the division alone was a third of the arithmetic
__device__ __forceinline__ float sigmoid_f(float u) { return __builtin_amdgcn_rcpf(1.f + __expf(-u)); }
__device__ __forceinline__ float swish_f(float u) { return u * sigmoid_f(u); }
__device__ __forceinline__ float act_grad_f(float u, bool swish) {
  if (!swish) return 1.f;
  const float sg = sigmoid_f(u);
  return sg * (1.f + u * (1.f - sg));
}
__device__ __forceinline__ void unpack8(const u32x4 v, float (&f)[8]) {
  f[0] = bf16_lo(v.x); f[1] = bf16_hi(v.x); f[2] = bf16_lo(v.y); f[3] = bf16_hi(v.y);
  f[4] = bf16_lo(v.z); f[5] = bf16_hi(v.z); f[6] = bf16_lo(v.w); f[7] = bf16_hi(v.w);
}
__device__ __forceinline__ u32x4 pack8(const float (&f)[8]) {
  u32x4 v;
  v.x = pack_bf16x2(f[0], f[1]); v.y = pack_bf16x2(f[2], f[3]); v.z = pack_bf16x2(f[4], f[5]); v.w = pack_bf16x2(f[6], f[7]);
  return v;
}

// partial [N][chunks][G][2] (sum, sum of squares), f32 per-thread sums over <= a few hundred values, LDS tree, f64 finalize
__global__ __launch_bounds__(256) void gnb_stats_kernel(const bf16_t* __restrict__ x, GnB s, float* __restrict__ partial) {
  __shared__ float red[2][256 * 8];
  const int tid = threadIdx.x, q = tid % s.octs, psub = tid / s.octs;
  const int n = blockIdx.y, chunk = blockIdx.x;
  const int p_beg = chunk * s.pix_per_chunk, p_end = min(s.HW, p_beg + s.pix_per_chunk);
  float sm[8] = {0, 0, 0, 0, 0, 0, 0, 0}, sq[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (psub < s.pix_per_pass) {
    const bf16_t* xn = x + (int64_t)n * s.HW * s.C + 8 * q;
    int px = p_beg + psub;
    for (; px + 3 * s.pix_per_pass < p_end; px += 4 * s.pix_per_pass) {     // four 16-byte loads in flight per thread
      const u32x4 v0 = *reinterpret_cast<const u32x4*>(xn + (int64_t)px * s.C);
      const u32x4 v1 = *reinterpret_cast<const u32x4*>(xn + (int64_t)(px + s.pix_per_pass) * s.C);
      const u32x4 v2 = *reinterpret_cast<const u32x4*>(xn + (int64_t)(px + 2 * s.pix_per_pass) * s.C);
      const u32x4 v3 = *reinterpret_cast<const u32x4*>(xn + (int64_t)(px + 3 * s.pix_per_pass) * s.C);
      float a[8], b[8], c[8], d[8];
      unpack8(v0, a); unpack8(v1, b); unpack8(v2, c); unpack8(v3, d);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        sm[j] += (a[j] + b[j]) + (c[j] + d[j]);
        sq[j] += (a[j] * a[j] + b[j] * b[j]) + (c[j] * c[j] + d[j] * d[j]);
      }
    }
    for (; px < p_end; px += s.pix_per_pass) {
      float a[8];
      unpack8(*reinterpret_cast<const u32x4*>(xn + (int64_t)px * s.C), a);
#pragma unroll
      for (int j = 0; j < 8; ++j) { sm[j] += a[j]; sq[j] += a[j] * a[j]; }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { red[0][psub * s.C + 8 * q + j] = sm[j]; red[1][psub * s.C + 8 * q + j] = sq[j]; }
  }
  __syncthreads();
  for (int c = tid; c < s.C; c += 256) {
    float a = 0.f, b = 0.f;
    for (int ps = 0; ps < s.pix_per_pass; ++ps) { a += red[0][ps * s.C + c]; b += red[1][ps * s.C + c]; }
    red[0][c] = a; red[1][c] = b;
  }
  __syncthreads();
  if (tid < s.G) {
    float a = 0.f, b = 0.f;
    for (int j = 0; j < s.cpg; ++j) { a += red[0][tid * s.cpg + j]; b += red[1][tid * s.cpg + j]; }
    float* o = partial + (((int64_t)n * s.chunks + chunk) * s.G + tid) * 2;
    o[0] = a; o[1] = b;
  }
}

struct Oct { float g[8], b[8], mu[8], rs[8], ds1[8], ds2[8]; };
__device__ __forceinline__ void load_oct(const GnB& s, int n, int q, const float* gamma, const float* beta, const float* mean,
                                         const float* rstd, const float* grp, Oct& k) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = 8 * q + j, g = c / s.cpg;
    k.g[j] = gamma[c]; k.b[j] = beta[c];
    k.mu[j] = mean[n * s.G + g]; k.rs[j] = rstd[n * s.G + g];
    if (grp) { k.ds1[j] = grp[((int64_t)n * s.G + g) * 2]; k.ds2[j] = grp[((int64_t)n * s.G + g) * 2 + 1]; }
  }
}

// grid (blocks, N); a thread keeps one channel octet (256 % octs == 0), so the affine constants are loop-invariant registers
// streaming accesses (`nt`): every byte of these passes is touched once here and next by another kernel a gigabyte later; measured on the
// f32 twins -4 % (backward apply) / -6 % (forward apply) per launch
typedef unsigned int gnb_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4 ldnt(const u32x4* p) { return __builtin_nontemporal_load(reinterpret_cast<const gnb_u4*>(p)); }
__device__ __forceinline__ void stnt(u32x4* p, u32x4 v) { __builtin_nontemporal_store(v, reinterpret_cast<gnb_u4*>(p)); }

__global__ __launch_bounds__(256) void gnb_apply_kernel(const bf16_t* __restrict__ x, GnB s, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, const float* __restrict__ mean,
                                                        const float* __restrict__ rstd, int swish, bf16_t* __restrict__ y) {
  const int n = blockIdx.y, per_n = s.HW * s.octs;
  const u32x4* xn = reinterpret_cast<const u32x4*>(x) + (int64_t)n * per_n;
  u32x4* yn = reinterpret_cast<u32x4*>(y) + (int64_t)n * per_n;
  const int stride = gridDim.x * 256;
  Oct k;
  load_oct(s, n, threadIdx.x % s.octs, gamma, beta, mean, rstd, nullptr, k);
  auto f = [&](const u32x4 v) {
    float a[8];
    unpack8(v, a);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float u = (a[j] - k.mu[j]) * k.rs[j] * k.g[j] + k.b[j];
      a[j] = swish ? swish_f(u) : u;
    }
    return pack8(a);
  };
  int i = blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * stride < per_n; i += 4 * stride) {
    const u32x4 v0 = ldnt(xn + i), v1 = ldnt(xn + i + stride), v2 = ldnt(xn + i + 2 * stride), v3 = ldnt(xn + i + 3 * stride);
    stnt(yn + i, f(v0)); stnt(yn + i + stride, f(v1)); stnt(yn + i + 2 * stride, f(v2)); stnt(yn + i + 3 * stride, f(v3));
  }
  for (; i < per_n; i += stride) stnt(yn + i, f(ldnt(xn + i)));
}

// partial [N][chunks][2][C]: per channel sums of du*xhat and du over the chunk's pixels
__global__ __launch_bounds__(256) void gnb_bwd_reduce_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy, GnB s,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             const float* __restrict__ mean, const float* __restrict__ rstd, int swish,
                                                             float* __restrict__ partial) {
  __shared__ float red[2][256 * 8];
  const int tid = threadIdx.x, q = tid % s.octs, psub = tid / s.octs;
  const int n = blockIdx.y, chunk = blockIdx.x;
  const int p_beg = chunk * s.pix_per_chunk, p_end = min(s.HW, p_beg + s.pix_per_chunk);
  if (psub < s.pix_per_pass) {
    Oct k;
    load_oct(s, n, q, gamma, beta, mean, rstd, nullptr, k);
    float a[8] = {0, 0, 0, 0, 0, 0, 0, 0}, b[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int64_t base = (int64_t)n * s.HW * s.C + 8 * q;
    auto accum = [&](const u32x4 xv, const u32x4 dv) {
      float xi[8], di[8];
      unpack8(xv, xi); unpack8(dv, di);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float xh = (xi[j] - k.mu[j]) * k.rs[j];
        const float du = di[j] * act_grad_f(xh * k.g[j] + k.b[j], swish);
        a[j] += du * xh; b[j] += du;
      }
    };
    int px = p_beg + psub;
    for (; px + s.pix_per_pass < p_end; px += 2 * s.pix_per_pass) {         // four 16-byte loads in flight per thread
      const u32x4 x0 = ldnt(reinterpret_cast<const u32x4*>(x + base + (int64_t)px * s.C));
      const u32x4 d0 = ldnt(reinterpret_cast<const u32x4*>(dy + base + (int64_t)px * s.C));
      const u32x4 x1 = ldnt(reinterpret_cast<const u32x4*>(x + base + (int64_t)(px + s.pix_per_pass) * s.C));
      const u32x4 d1 = ldnt(reinterpret_cast<const u32x4*>(dy + base + (int64_t)(px + s.pix_per_pass) * s.C));
      accum(x0, d0);
      accum(x1, d1);
    }
    for (; px < p_end; px += s.pix_per_pass)
      accum(*reinterpret_cast<const u32x4*>(x + base + (int64_t)px * s.C), *reinterpret_cast<const u32x4*>(dy + base + (int64_t)px * s.C));
#pragma unroll
    for (int j = 0; j < 8; ++j) { red[0][psub * s.C + 8 * q + j] = a[j]; red[1][psub * s.C + 8 * q + j] = b[j]; }
  }
  __syncthreads();
  float* o = partial + ((int64_t)n * s.chunks + chunk) * 2 * s.C;
  for (int cc = tid; cc < s.C; cc += 256) {
    float sa = 0.f, sb = 0.f;
    for (int ps = 0; ps < s.pix_per_pass; ++ps) { sa += red[0][ps * s.C + cc]; sb += red[1][ps * s.C + cc]; }
    o[cc] = sa; o[s.C + cc] = sb;
  }
}

// dx = rstd * (du*gamma - (ds2 + xhat*ds1)/m) (+ the skip connection's gradient, folded in)
__global__ __launch_bounds__(256) void gnb_bwd_apply_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy, GnB s,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ grp, int swish, const bf16_t* __restrict__ dx_add,
                                                            bf16_t* __restrict__ dx) {
  const int n = blockIdx.y, per_n = s.HW * s.octs;
  const u32x4* xn = reinterpret_cast<const u32x4*>(x) + (int64_t)n * per_n;
  const u32x4* dn = reinterpret_cast<const u32x4*>(dy) + (int64_t)n * per_n;
  const u32x4* an = dx_add ? reinterpret_cast<const u32x4*>(dx_add) + (int64_t)n * per_n : nullptr;
  u32x4* on = reinterpret_cast<u32x4*>(dx) + (int64_t)n * per_n;
  const float inv_m = 1.f / ((float)s.HW * (float)s.cpg);
  const int stride = gridDim.x * 256;
  Oct k;
  load_oct(s, n, threadIdx.x % s.octs, gamma, beta, mean, rstd, grp, k);
  auto f = [&](const u32x4 xv, const u32x4 dv, int at) {
    float xi[8], di[8], ad[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unpack8(xv, xi); unpack8(dv, di);
    if (an) unpack8(ldnt(an + at), ad);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float xh = (xi[j] - k.mu[j]) * k.rs[j];
      const float du = di[j] * act_grad_f(xh * k.g[j] + k.b[j], swish);
      xi[j] = k.rs[j] * (du * k.g[j] - (k.ds2[j] + xh * k.ds1[j]) * inv_m) + ad[j];
    }
    return pack8(xi);
  };
  int i = blockIdx.x * 256 + threadIdx.x;
  for (; i + stride < per_n; i += 2 * stride) {
    const u32x4 x0 = ldnt(xn + i), x1 = ldnt(xn + i + stride), d0 = ldnt(dn + i), d1 = ldnt(dn + i + stride);
    stnt(on + i, f(x0, d0, i)); stnt(on + i + stride, f(x1, d1, i + stride));
  }
  for (; i < per_n; i += stride) stnt(on + i, f(ldnt(xn + i), ldnt(dn + i), i));
}

bool make_shape(int N, int HW, int C, int G, GnB& s) {
  if (N <= 0 || HW <= 0 || C <= 0 || G <= 0 || C % G != 0 || C % 8 != 0) return false;
  s.N = N; s.HW = HW; s.C = C; s.G = G; s.cpg = C / G; s.octs = C / 8;
  if (s.octs > 256 || 256 % s.octs != 0 || G > 256 || (int64_t)HW * s.octs >= ((int64_t)1 << 31) || N > 65535) return false;
  s.pix_per_pass = 256 / s.octs;
  int chunks = ceil_div(2048, N);
  const int max_chunks = ceil_div(HW, 64);
  if (chunks > max_chunks) chunks = max_chunks;
  if (chunks < 1) chunks = 1;
  s.pix_per_chunk = ceil_div(HW, chunks);
  s.chunks = ceil_div(HW, s.pix_per_chunk);
  return true;
}
int apply_blocks(const GnB& s) { return (int)std::min<int64_t>(std::max<int64_t>(ceil_div64((int64_t)s.HW * s.octs, 256 * 8), 1), 65535); }

// ---- dtype hand-offs ---------------------------------------------------------------------------------------------------------
// y[row][0..CP) bf16 = x[row][0..C) f32, zero for c >= C   (CP % 8 == 0)
__global__ void cast_pad_kernel(const float* __restrict__ x, int64_t rows, int C, int CP, bf16_t* __restrict__ y) {
  const int octs = CP / 8;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < rows * octs; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / octs;
    const int c0 = (int)(i % octs) * 8;
    float f[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = c0 + j < C ? x[r * C + c0 + j] : 0.f;
    *reinterpret_cast<u32x4*>(y + r * CP + c0) = pack8(f);
  }
}
__global__ void cast_f32_kernel(const bf16_t* __restrict__ x, int64_t n8, float* __restrict__ y) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
    float f[8];
    unpack8(*reinterpret_cast<const u32x4*>(x + 8 * i), f);
    *reinterpret_cast<float4*>(y + 8 * i) = make_float4(f[0], f[1], f[2], f[3]);
    *reinterpret_cast<float4*>(y + 8 * i + 4) = make_float4(f[4], f[5], f[6], f[7]);
  }
}
// dx[n][y][x][c] = sum of the 2x2 block of du (data gradient of nearest-2x upsampling)
__global__ void sumpool2x2_bf16_kernel(const bf16_t* __restrict__ du, bf16_t* __restrict__ dx, int N, int H, int W, int C) {
  const int octs = C / 8;
  const int64_t total = (int64_t)N * H * W * octs;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int q = (int)(i % octs);
    int64_t r = i / octs;
    const int xx = (int)(r % W); r /= W;
    const int yy = (int)(r % H); const int n = (int)(r / H);
    const bf16_t* src = du + (((int64_t)n * 2 * H + 2 * yy) * 2 * W + 2 * xx) * C + 8 * q;
    float a[8], b[8], c[8], d[8];
    unpack8(*reinterpret_cast<const u32x4*>(src), a);
    unpack8(*reinterpret_cast<const u32x4*>(src + C), b);
    unpack8(*reinterpret_cast<const u32x4*>(src + (int64_t)2 * W * C), c);
    unpack8(*reinterpret_cast<const u32x4*>(src + (int64_t)2 * W * C + C), d);
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = (a[j] + b[j]) + (c[j] + d[j]);
    *reinterpret_cast<u32x4*>(dx + 8 * i) = pack8(a);
  }
}
// partial[block][C] column sums of a [rows][C] bf16 matrix; second kernel adds the blocks in order
__global__ __launch_bounds__(256) void colsum_bf16_kernel(const bf16_t* __restrict__ x, int64_t rows, int C, int rows_per_block, float* __restrict__ partial) {
  __shared__ float red[256 * 8];
  const int octs = C / 8, tid = threadIdx.x, q = tid % octs, psub = tid / octs, ppp = 256 / octs;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (psub < ppp)
    for (int64_t r = r0 + psub; r < r1; r += ppp) {
      float a[8];
      unpack8(*reinterpret_cast<const u32x4*>(x + r * C + 8 * q), a);
#pragma unroll
      for (int j = 0; j < 8; ++j) s[j] += a[j];
    }
  if (psub < ppp)
#pragma unroll
    for (int j = 0; j < 8; ++j) red[psub * C + 8 * q + j] = s[j];
  __syncthreads();
  for (int c = tid; c < C; c += 256) {
    float a = 0.f;
    for (int ps = 0; ps < ppp; ++ps) a += red[ps * C + c];
    partial[(int64_t)blockIdx.x * C + c] = a;
  }
}
// one wave per channel: lanes stride over the blocks, fixed-order tree (deterministic)
__global__ void colsum_finish_kernel(const float* __restrict__ partial, int blocks, int C, float* __restrict__ out) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (c >= C) return;
  float a = 0.f;
  for (int b = lane; b < blocks; b += 64) a += partial[(int64_t)b * C + c];
  a = wave_sum(a);
  if (lane == 0) out[c] = a;
}

int colsum_blocks(int64_t rows) { return (int)std::min<int64_t>(512, std::max<int64_t>(1, ceil_div64(rows, 512))); }

}  // namespace

extern "C" {

size_t odvae_groupnorm_bf16_workspace_bytes(int N, int HW, int C, int G) {
  GnB s;
  if (!make_shape(N, HW, C, G, s)) return 0;
  const size_t fwd = (size_t)N * s.chunks * G * 2;
  const size_t bwd = (size_t)N * s.chunks * 2 * C + (size_t)N * 2 * C + (size_t)N * G * 2;
  return (fwd > bwd ? fwd : bwd) * sizeof(float);
}

// y = act(GroupNorm(x)), x / y bf16 [N][HW][C]; gamma / beta f32; mean / rstd f32 [N][G] saved for the backward
int odvae_groupnorm_fwd_bf16(const void* x, int N, int HW, int C, int G, const float* gamma, const float* beta, float eps, int swish,
                             void* y, float* mean, float* rstd, void* workspace, size_t workspace_bytes, void* stream) {
  GnB s;
  ODVAE_CHECK_ARG(make_shape(N, HW, C, G, s), "groupnorm_fwd_bf16: unsupported shape N=%d HW=%d C=%d G=%d (need C%%G==0, C%%8==0, 256%%(C/8)==0)", N, HW, C, G);
  ODVAE_CHECK_ARG(x && gamma && beta && y && mean && rstd, "groupnorm_fwd_bf16: null operand");
  ODVAE_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0, "groupnorm_fwd_bf16: operands must be 16-byte aligned");
  const size_t need = (size_t)N * s.chunks * G * 2 * sizeof(float);
  if (!workspace || workspace_bytes < need) {
    odvae_set_error("groupnorm_fwd_bf16: needs %zu workspace bytes, got %zu", need, workspace_bytes);
    return ODVAE_ERR_WORKSPACE;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  float* partial = static_cast<float*>(workspace);
  hipLaunchKernelGGL(gnb_stats_kernel, dim3(s.chunks, N), dim3(256), 0, st, static_cast<const bf16_t*>(x), s, partial);
  hipLaunchKernelGGL(gn_finalize_kernel<GnB>, dim3(ceil_div(N * G, 4)), dim3(256), 0, st, partial, s, eps, mean, rstd);
  hipLaunchKernelGGL(gnb_apply_kernel, dim3(apply_blocks(s), N), dim3(256), 0, st, static_cast<const bf16_t*>(x), s, gamma, beta, mean, rstd,
                     swish, static_cast<bf16_t*>(y));
  ODVAE_LAUNCH_CHECK("groupnorm_fwd_bf16");
  return ODVAE_OK;
}

// The same without the statistics pass: partial [N][chunks][G][2] = (sum, sum of squares) of x per chunk and channel group as the conv
// that produced x left them (odvae_conv_bf16_stats: one chunk per output tile).  finalize (f64, fixed order over the chunks) + apply.
int odvae_groupnorm_fwd_partials_bf16(const void* x, int N, int HW, int C, int G, const float* gamma, const float* beta, float eps, int swish,
                                      void* y, float* mean, float* rstd, const float* partial, int chunks, void* stream) {
  GnB s;
  ODVAE_CHECK_ARG(make_shape(N, HW, C, G, s), "groupnorm_fwd_partials_bf16: unsupported shape N=%d HW=%d C=%d G=%d", N, HW, C, G);
  ODVAE_CHECK_ARG(x && gamma && beta && y && mean && rstd && partial && chunks > 0, "groupnorm_fwd_partials_bf16: null operand");
  ODVAE_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0 && ((uintptr_t)partial & 7) == 0, "groupnorm_fwd_partials_bf16: misaligned operand");
  hipStream_t st = static_cast<hipStream_t>(stream);
  GnB sf = s;
  sf.chunks = chunks;
  hipLaunchKernelGGL(gn_finalize_kernel<GnB>, dim3(ceil_div(N * G, 4)), dim3(256), 0, st, partial, sf, eps, mean, rstd);
  hipLaunchKernelGGL(gnb_apply_kernel, dim3(apply_blocks(s), N), dim3(256), 0, st, static_cast<const bf16_t*>(x), s, gamma, beta, mean, rstd,
                     swish, static_cast<bf16_t*>(y));
  ODVAE_LAUNCH_CHECK("groupnorm_fwd_partials_bf16");
  return ODVAE_OK;
}

// The apply pass alone with mean / rstd given (odvae_groupnorm_apply_f32's bf16 twin)
int odvae_groupnorm_apply_bf16(const void* x, int N, int HW, int C, int G, const float* gamma, const float* beta,
                               const float* mean, const float* rstd, int swish, void* y, void* stream) {
  GnB s;
  ODVAE_CHECK_ARG(make_shape(N, HW, C, G, s), "groupnorm_apply_bf16: unsupported shape N=%d HW=%d C=%d G=%d", N, HW, C, G);
  ODVAE_CHECK_ARG(x && gamma && beta && y && mean && rstd, "groupnorm_apply_bf16: null operand");
  ODVAE_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0, "groupnorm_apply_bf16: misaligned operand");
  hipLaunchKernelGGL(gnb_apply_kernel, dim3(apply_blocks(s), N), dim3(256), 0, static_cast<hipStream_t>(stream), static_cast<const bf16_t*>(x), s,
                     gamma, beta, mean, rstd, swish, static_cast<bf16_t*>(y));
  ODVAE_LAUNCH_CHECK("groupnorm_apply_bf16");
  return ODVAE_OK;
}

// dx bf16 (+ dx_add bf16, the folded skip gradient, or NULL), dgamma / dbeta f32 [C]
int odvae_groupnorm_bwd_bf16(const void* x, const void* dy, int N, int HW, int C, int G, const float* gamma, const float* beta,
                             const float* mean, const float* rstd, int swish, void* dx, float* dgamma, float* dbeta, const void* dx_add,
                             void* workspace, size_t workspace_bytes, void* stream) {
  GnB s;
  ODVAE_CHECK_ARG(make_shape(N, HW, C, G, s), "groupnorm_bwd_bf16: unsupported shape N=%d HW=%d C=%d G=%d", N, HW, C, G);
  ODVAE_CHECK_ARG(x && dy && gamma && beta && mean && rstd && dx && dgamma && dbeta, "groupnorm_bwd_bf16: null operand");
  ODVAE_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)dy & 15) == 0 && ((uintptr_t)dx & 15) == 0 && ((uintptr_t)dx_add & 15) == 0,
                  "groupnorm_bwd_bf16: operands must be 16-byte aligned");
  const size_t need = odvae_groupnorm_bf16_workspace_bytes(N, HW, C, G);
  if (!workspace || workspace_bytes < need) {
    odvae_set_error("groupnorm_bwd_bf16: needs %zu workspace bytes, got %zu", need, workspace_bytes);
    return ODVAE_ERR_WORKSPACE;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  float* partial = static_cast<float*>(workspace);
  float* chan = partial + (size_t)N * s.chunks * 2 * C;
  float* grp = chan + (size_t)N * 2 * C;
  const bf16_t* xb = static_cast<const bf16_t*>(x);
  const bf16_t* db = static_cast<const bf16_t*>(dy);
  hipLaunchKernelGGL(gnb_bwd_reduce_kernel, dim3(s.chunks, N), dim3(256), 0, st, xb, db, s, gamma, beta, mean, rstd, swish, partial);
  if (s.cpg <= 64) hipLaunchKernelGGL(gn_bwd_finalize_kernel<GnB>, dim3(N, ceil_div(C, (64 / s.cpg) * s.cpg)), dim3(256), 0, st, partial, s, gamma, chan, grp);
  else hipLaunchKernelGGL(gn_bwd_finalize_wide_kernel<GnB>, dim3(N), dim3(256), 2 * C * sizeof(float), st, partial, s, gamma, chan, grp);
  hipLaunchKernelGGL(gn_bwd_param_kernel, dim3(ceil_div(C, 64)), dim3(256), 0, st, chan, N, C, dgamma, dbeta);
  hipLaunchKernelGGL(gnb_bwd_apply_kernel, dim3(apply_blocks(s), N), dim3(256), 0, st, xb, db, s, gamma, beta, mean, rstd, grp, swish,
                     static_cast<const bf16_t*>(dx_add), static_cast<bf16_t*>(dx));
  ODVAE_LAUNCH_CHECK("groupnorm_bwd_bf16");
  return ODVAE_OK;
}

// y bf16 [rows][CP] = x f32 [rows][C], zero-padded channels (CP >= C, CP % 8 == 0): the f32 -> bf16 hand-off (input image,
// latent, reconstruction gradient; CP > C gives 3-channel tensors the 16-byte channel vectors the conv kernels read)
int odvae_cast_pad_bf16(const float* x, int64_t rows, int C, int CP, void* y, void* stream) {
  ODVAE_CHECK_ARG(x && y && rows > 0 && C > 0 && CP >= C && CP % 8 == 0, "cast_pad_bf16: bad arguments (rows=%lld C=%d CP=%d)", (long long)rows, C, CP);
  ODVAE_CHECK_ARG(((uintptr_t)y & 15) == 0, "cast_pad_bf16: y must be 16-byte aligned");
  const int64_t work = rows * (CP / 8);
  hipLaunchKernelGGL(cast_pad_kernel, dim3((unsigned)std::min<int64_t>(ceil_div64(work, 256), 8192)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), x, rows, C, CP, static_cast<bf16_t*>(y));
  ODVAE_LAUNCH_CHECK("cast_pad_bf16");
  return ODVAE_OK;
}

int odvae_cast_f32_from_bf16(const void* x, int64_t n, float* y, void* stream) {
  ODVAE_CHECK_ARG(x && y && n > 0 && n % 8 == 0, "cast_f32_from_bf16: n = %lld must be a positive multiple of 8", (long long)n);
  ODVAE_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0, "cast_f32_from_bf16: operands must be 16-byte aligned");
  hipLaunchKernelGGL(cast_f32_kernel, dim3((unsigned)std::min<int64_t>(ceil_div64(n / 8, 256), 8192)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), static_cast<const bf16_t*>(x), n / 8, y);
  ODVAE_LAUNCH_CHECK("cast_f32_from_bf16");
  return ODVAE_OK;
}

// dx [N][H][W][C] = 2x2 sum-pool of du [N][2H][2W][C] (bf16): data gradient of F.interpolate(scale 2, nearest)
int odvae_upsample2x_bwd_bf16(const void* du, void* dx, int N, int H, int W, int C, void* stream) {
  ODVAE_CHECK_ARG(du && dx && N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "upsample2x_bwd_bf16: bad arguments");
  const int64_t work = (int64_t)N * H * W * (C / 8);
  hipLaunchKernelGGL(sumpool2x2_bf16_kernel, dim3((unsigned)std::min<int64_t>(ceil_div64(work, 256), 16384)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), static_cast<const bf16_t*>(du), static_cast<bf16_t*>(dx), N, H, W, C);
  ODVAE_LAUNCH_CHECK("upsample2x_bwd_bf16");
  return ODVAE_OK;
}

size_t odvae_colsum_bf16_workspace_bytes(int64_t rows, int C) { return (size_t)colsum_blocks(rows) * C * sizeof(float); }

// out f32 [C] = column sums of x bf16 [rows][C] (bias gradients); deterministic
int odvae_colsum_bf16(const void* x, int64_t rows, int C, float* out, void* workspace, size_t workspace_bytes, void* stream) {
  ODVAE_CHECK_ARG(x && out && rows > 0 && C > 0 && C % 8 == 0 && C / 8 <= 256, "colsum_bf16: unsupported shape rows=%lld C=%d (C %% 8 == 0, C <= 2048)", (long long)rows, C);
  const int blocks = colsum_blocks(rows);
  if (!workspace || workspace_bytes < (size_t)blocks * C * sizeof(float)) {
    odvae_set_error("colsum_bf16: needs %zu workspace bytes, got %zu", (size_t)blocks * C * sizeof(float), workspace_bytes);
    return ODVAE_ERR_WORKSPACE;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  float* partial = static_cast<float*>(workspace);
  hipLaunchKernelGGL(colsum_bf16_kernel, dim3(blocks), dim3(256), 0, st, static_cast<const bf16_t*>(x), rows, C, (int)ceil_div64(rows, blocks), partial);
  hipLaunchKernelGGL(colsum_finish_kernel, dim3(ceil_div(C, 4)), dim3(256), 0, st, partial, blocks, C, out);
  ODVAE_LAUNCH_CHECK("colsum_bf16");
  return ODVAE_OK;
}

}  // extern "C"
