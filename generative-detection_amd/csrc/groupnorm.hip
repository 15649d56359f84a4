// GroupNorm (+ optional swish) forward/backward, NHWC f32, gfx950.  HBM-bound.
//
// [UPSTREAM] ldm/modules/diffusionmodules/model.py: Normalize = GroupNorm(32, C, eps=1e-6, affine=True),
// nonlinearity(x) = x*sigmoid(x); 67 sites in Encoder/Decoder (ResnetBlock.norm1/norm2, AttnBlock.norm,
// norm_out), reached from src/modules/autoencodermodules/feat_encoder.py:4 / feat_decoder.py:4.
//
// x is [N][HW][C].  Statistics: each block reduces a run of pixels of one sample with float4 loads
// (a wavefront reads 1 KiB of contiguous channels), per-thread f32 sums over <= a few hundred values,
// LDS tree to per-channel sums, then per-group partials; a finalize kernel combines the partials in
// f64 (fixed order => deterministic) into mean / rstd.  Apply is a pure streaming float4 pass.
// Backward recomputes xhat and the swish derivative from x (nothing but x, mean, rstd is saved).
#include "common.h"

namespace {

#include "gn_finalize.h"

struct GnShape {
  int N, HW, C, G, cpg;      // cpg = C / G
  int quads;                 // C / 4
  int pix_per_pass;          // 256 / quads
  int chunks, pix_per_chunk;
};

// 1 / (1 + e^-u) with v_rcp_f32 (1 ulp) instead of an IEEE division (ten instructions): these kernels carry 18-30 vector instructions per
// element beside their loads, and the bf16 ones move two elements per 4 bytes -- the division alone was a third of the arithmetic
__device__ __forceinline__ float sigmoid_f(float u) { return __builtin_amdgcn_rcpf(1.f + __expf(-u)); }
__device__ __forceinline__ float swish_f(float u) { return u * sigmoid_f(u); }

// ---- forward statistics -------------------------------------------------------------------------
// partial: [N][chunks][G][2]  (sum, sum of squares)
__global__ __launch_bounds__(256) void gn_stats_kernel(const float* __restrict__ x, GnShape s, float* __restrict__ partial) {
  __shared__ float red[2][256 * 4];  // [stat][psub][C] flattened: psub*C + c  (pix_per_pass*C = 1024)
  const int tid = threadIdx.x;
  const int q = tid % s.quads, psub = tid / s.quads;
  const int n = blockIdx.y, chunk = blockIdx.x;
  const int p_beg = chunk * s.pix_per_chunk;
  const int p_end = min(s.HW, p_beg + s.pix_per_chunk);
  const float* xn = x + (int64_t)n * s.HW * s.C + 4 * q;
  float4 sm = make_float4(0.f, 0.f, 0.f, 0.f), sq = sm;
  if (psub < s.pix_per_pass) {
    auto add = [&](const float4 v) {
      sm.x += v.x; sm.y += v.y; sm.z += v.z; sm.w += v.w;
      sq.x += v.x * v.x; sq.y += v.y * v.y; sq.z += v.z * v.z; sq.w += v.w * v.w;
    };
    const int64_t step = (int64_t)s.pix_per_pass * s.C;
    const float* ptr = xn + (int64_t)(p_beg + psub) * s.C;
    int px = p_beg + psub;
    for (; px + 3 * s.pix_per_pass < p_end; px += 4 * s.pix_per_pass, ptr += 4 * step) {   // four loads in flight
      const float4 v0 = *reinterpret_cast<const float4*>(ptr), v1 = *reinterpret_cast<const float4*>(ptr + step);
      const float4 v2 = *reinterpret_cast<const float4*>(ptr + 2 * step), v3 = *reinterpret_cast<const float4*>(ptr + 3 * step);
      add(v0); add(v1); add(v2); add(v3);
    }
    for (; px < p_end; px += s.pix_per_pass, ptr += step) add(*reinterpret_cast<const float4*>(ptr));
    *reinterpret_cast<float4*>(&red[0][psub * s.C + 4 * q]) = sm;
    *reinterpret_cast<float4*>(&red[1][psub * s.C + 4 * q]) = sq;
  }
  __syncthreads();
  // per-channel totals (threads stride over channels), written back to row 0
  for (int c = tid; c < s.C; c += 256) {
    float a = 0.f, b = 0.f;
    for (int ps = 0; ps < s.pix_per_pass; ++ps) { a += red[0][ps * s.C + c]; b += red[1][ps * s.C + c]; }
    red[0][c] = a; red[1][c] = b;   // row 0 only read by its own writer in this loop
  }
  __syncthreads();
  if (tid < s.G) {
    float a = 0.f, b = 0.f;
    for (int j = 0; j < s.cpg; ++j) { a += red[0][tid * s.cpg + j]; b += red[1][tid * s.cpg + j]; }
    float* o = partial + (((int64_t)n * s.chunks + chunk) * s.G + tid) * 2;
    o[0] = a; o[1] = b;
  }
}

// ---- forward apply -------------------------------------------------------------------------------
// grid (blocks, N): no 64-bit index arithmetic; with 256 % quads == 0 (C = 128/256/512) a thread keeps one channel quad,
// so gamma/beta/mean/rstd are loop-invariant registers and the loop is four independent 16-byte loads in flight
struct GnQuad { float g[4], b[4], mu[4], rs[4], ds1[4], ds2[4]; };

__device__ __forceinline__ void gn_load_quad(const GnShape& s, int n, int q, const float* gamma, const float* beta,
                                             const float* mean, const float* rstd, const float* grp, GnQuad& k) {
  const float4 ga = *reinterpret_cast<const float4*>(gamma + 4 * q);
  const float4 be = *reinterpret_cast<const float4*>(beta + 4 * q);
  k.g[0] = ga.x; k.g[1] = ga.y; k.g[2] = ga.z; k.g[3] = ga.w;
  k.b[0] = be.x; k.b[1] = be.y; k.b[2] = be.z; k.b[3] = be.w;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int g = (4 * q + j) / s.cpg;
    k.mu[j] = mean[n * s.G + g]; k.rs[j] = rstd[n * s.G + g];
    if (grp) { k.ds1[j] = grp[((int64_t)n * s.G + g) * 2 + 0]; k.ds2[j] = grp[((int64_t)n * s.G + g) * 2 + 1]; }
  }
}

template <bool SWISH>
__device__ __forceinline__ float4 gn_apply_quad(const float4 v, const GnQuad& k) {
  const float in[4] = {v.x, v.y, v.z, v.w};
  float out[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float u = (in[j] - k.mu[j]) * k.rs[j] * k.g[j] + k.b[j];
    out[j] = SWISH ? swish_f(u) : u;
  }
  return make_float4(out[0], out[1], out[2], out[3]);
}

template <bool SWISH>
__global__ __launch_bounds__(256) void gn_apply_kernel(const float* __restrict__ x, GnShape s,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ mean, const float* __restrict__ rstd,
                                                       float* __restrict__ y) {
  const int n = blockIdx.y;
  const int per_n = s.HW * s.quads;
  const float4* xn = reinterpret_cast<const float4*>(x) + (int64_t)n * per_n;
  float4* yn = reinterpret_cast<float4*>(y) + (int64_t)n * per_n;
  const int stride = gridDim.x * 256;
  int i = blockIdx.x * 256 + threadIdx.x;
  GnQuad k;
  if (256 % s.quads == 0) {
    gn_load_quad(s, n, threadIdx.x % s.quads, gamma, beta, mean, rstd, nullptr, k);
    for (; i + 3 * stride < per_n; i += 4 * stride) {
      const float4 v0 = xn[i], v1 = xn[i + stride], v2 = xn[i + 2 * stride], v3 = xn[i + 3 * stride];
      yn[i] = gn_apply_quad<SWISH>(v0, k); yn[i + stride] = gn_apply_quad<SWISH>(v1, k);
      yn[i + 2 * stride] = gn_apply_quad<SWISH>(v2, k); yn[i + 3 * stride] = gn_apply_quad<SWISH>(v3, k);
    }
    for (; i < per_n; i += stride) yn[i] = gn_apply_quad<SWISH>(xn[i], k);
  } else {
    for (; i < per_n; i += stride) {
      gn_load_quad(s, n, i % s.quads, gamma, beta, mean, rstd, nullptr, k);
      yn[i] = gn_apply_quad<SWISH>(xn[i], k);
    }
  }
}

// ---- backward ------------------------------------------------------------------------------------
// du = dy * d(act)/du with u = xhat*gamma + beta.  Pass 1: per-(n, chunk, c) sums of du*xhat and du.
template <bool SWISH>
__device__ __forceinline__ float act_grad(float u) {
  if (!SWISH) return 1.f;
  const float sg = sigmoid_f(u);
  return sg * (1.f + u * (1.f - sg));
}

// partial: [N][chunks][2][C]
template <bool SWISH>
__global__ __launch_bounds__(256) void gn_bwd_reduce_kernel(const float* __restrict__ x, const float* __restrict__ dy, GnShape s,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            float* __restrict__ partial) {
  __shared__ float red[2][256 * 4];
  const int tid = threadIdx.x;
  const int q = tid % s.quads, psub = tid / s.quads;
  const int n = blockIdx.y, chunk = blockIdx.x;
  const int p_beg = chunk * s.pix_per_chunk;
  const int p_end = min(s.HW, p_beg + s.pix_per_chunk);
  const int c = 4 * q;
  float a[4] = {0.f, 0.f, 0.f, 0.f}, b[4] = {0.f, 0.f, 0.f, 0.f};
  if (psub < s.pix_per_pass) {
    const float4 ga = *reinterpret_cast<const float4*>(gamma + c);
    const float4 be = *reinterpret_cast<const float4*>(beta + c);
    const float gg[4] = {ga.x, ga.y, ga.z, ga.w}, bb[4] = {be.x, be.y, be.z, be.w};
    float mu[4], rs[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { const int g = (c + j) / s.cpg; mu[j] = mean[n * s.G + g]; rs[j] = rstd[n * s.G + g]; }
    const int64_t base = (int64_t)n * s.HW * s.C + c;
    for (int px = p_beg + psub; px < p_end; px += s.pix_per_pass) {
      const float4 xv = *reinterpret_cast<const float4*>(x + base + (int64_t)px * s.C);
      const float4 dv = *reinterpret_cast<const float4*>(dy + base + (int64_t)px * s.C);
      const float xi[4] = {xv.x, xv.y, xv.z, xv.w}, di[4] = {dv.x, dv.y, dv.z, dv.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float xh = (xi[j] - mu[j]) * rs[j];
        const float du = di[j] * act_grad<SWISH>(xh * gg[j] + bb[j]);
        a[j] += du * xh; b[j] += du;
      }
    }
    *reinterpret_cast<float4*>(&red[0][psub * s.C + c]) = make_float4(a[0], a[1], a[2], a[3]);
    *reinterpret_cast<float4*>(&red[1][psub * s.C + c]) = make_float4(b[0], b[1], b[2], b[3]);
  }
  __syncthreads();
  float* o = partial + ((int64_t)n * s.chunks + chunk) * 2 * s.C;
  for (int cc = tid; cc < s.C; cc += 256) {
    float sa = 0.f, sb = 0.f;
    for (int ps = 0; ps < s.pix_per_pass; ++ps) { sa += red[0][ps * s.C + cc]; sb += red[1][ps * s.C + cc]; }
    o[cc] = sa; o[s.C + cc] = sb;
  }
}

// dx = rstd * (du*gamma - (ds2 + xhat*ds1)/m); same launch shape as gn_apply_kernel
template <bool SWISH>
__device__ __forceinline__ float4 gn_bwd_quad(const float4 xv, const float4 dv, const GnQuad& k, float inv_m) {
  const float xi[4] = {xv.x, xv.y, xv.z, xv.w}, di[4] = {dv.x, dv.y, dv.z, dv.w};
  float out[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float xh = (xi[j] - k.mu[j]) * k.rs[j];
    const float du = di[j] * act_grad<SWISH>(xh * k.g[j] + k.b[j]);
    out[j] = k.rs[j] * (du * k.g[j] - (k.ds2[j] + xh * k.ds1[j]) * inv_m);
  }
  return make_float4(out[0], out[1], out[2], out[3]);
}

template <bool SWISH>
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ dy, GnShape s,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           const float* __restrict__ grp, const float* __restrict__ dx_add,
                                                           float* __restrict__ dx) {
  const int n = blockIdx.y;
  const int per_n = s.HW * s.quads;
  const float4* xn = reinterpret_cast<const float4*>(x) + (int64_t)n * per_n;
  const float4* dn = reinterpret_cast<const float4*>(dy) + (int64_t)n * per_n;
  const float4* an = dx_add ? reinterpret_cast<const float4*>(dx_add) + (int64_t)n * per_n : nullptr;
  auto plus = [&](float4 v, int at) {
    if (an) { const float4 a = an[at]; v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w; }
    return v;
  };
  float4* on = reinterpret_cast<float4*>(dx) + (int64_t)n * per_n;
  const float inv_m = 1.f / ((float)s.HW * (float)s.cpg);
  const int stride = gridDim.x * 256;
  int i = blockIdx.x * 256 + threadIdx.x;
  GnQuad k;
  if (256 % s.quads == 0) {
    gn_load_quad(s, n, threadIdx.x % s.quads, gamma, beta, mean, rstd, grp, k);
    for (; i + stride < per_n; i += 2 * stride) {
      const float4 x0 = xn[i], x1 = xn[i + stride], d0 = dn[i], d1 = dn[i + stride];
      on[i] = plus(gn_bwd_quad<SWISH>(x0, d0, k, inv_m), i); on[i + stride] = plus(gn_bwd_quad<SWISH>(x1, d1, k, inv_m), i + stride);
    }
    for (; i < per_n; i += stride) on[i] = plus(gn_bwd_quad<SWISH>(xn[i], dn[i], k, inv_m), i);
  } else {
    for (; i < per_n; i += stride) {
      gn_load_quad(s, n, i % s.quads, gamma, beta, mean, rstd, grp, k);
      on[i] = plus(gn_bwd_quad<SWISH>(xn[i], dn[i], k, inv_m), i);
    }
  }
}

bool make_shape(int N, int HW, int C, int G, GnShape& s) {
  if (N <= 0 || HW <= 0 || C <= 0 || G <= 0 || C % G != 0 || C % 4 != 0) return false;
  s.N = N; s.HW = HW; s.C = C; s.G = G; s.cpg = C / G; s.quads = C / 4;
  if (s.quads > 256 || G > 256 || (int64_t)HW * s.quads >= ((int64_t)1 << 31) || N > 65535) return false;
  s.pix_per_pass = 256 / s.quads;
  // aim for >= ~2048 blocks overall, at least 64 pixels per block
  int chunks = ceil_div(2048, N);
  const int max_chunks = ceil_div(HW, 64);
  if (chunks > max_chunks) chunks = max_chunks;
  if (chunks < 1) chunks = 1;
  s.pix_per_chunk = ceil_div(HW, chunks);
  s.chunks = ceil_div(HW, s.pix_per_chunk);
  return true;
}

// x-dimension of the (blocks, N) apply grids: about eight float4 per thread
int apply_blocks(const GnShape& s) { return (int)std::min<int64_t>(std::max<int64_t>(ceil_div64((int64_t)s.HW * s.quads, 256 * 8), 1), 65535); }

}  // namespace

extern "C" {

// scratch for forward (stats partials) and backward (channel partials + chan + grp)
size_t odvae_groupnorm_workspace_bytes(int N, int HW, int C, int G) {
  GnShape s;
  if (!make_shape(N, HW, C, G, s)) return 0;
  const size_t fwd = (size_t)N * s.chunks * G * 2;
  const size_t bwd = (size_t)N * s.chunks * 2 * C + (size_t)N * 2 * C + (size_t)N * G * 2;
  return (fwd > bwd ? fwd : bwd) * sizeof(float);
}

// y = act(GroupNorm(x)); mean/rstd [N][G] are outputs (saved for backward).  swish: 0 = identity, 1 = x*sigmoid(x)
int odvae_groupnorm_fwd_f32(const float* x, int N, int HW, int C, int G, const float* gamma, const float* beta,
                            float eps, int swish, float* y, float* mean, float* rstd,
                            void* workspace, size_t workspace_bytes, void* stream) {
  GnShape s;
  ODVAE_CHECK_ARG(make_shape(N, HW, C, G, s), "groupnorm_fwd: unsupported shape N=%d HW=%d C=%d G=%d (need C%%G==0, C%%4==0, C<=1024)", N, HW, C, G);
  ODVAE_CHECK_ARG(x && gamma && beta && y && mean && rstd, "groupnorm_fwd: null operand");
  ODVAE_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0 && ((uintptr_t)gamma & 15) == 0 && ((uintptr_t)beta & 15) == 0,
                  "groupnorm_fwd: operands must be 16-byte aligned");
  const size_t need = (size_t)N * s.chunks * G * 2 * sizeof(float);
  if (!workspace || workspace_bytes < need) {
    odvae_set_error("groupnorm_fwd: needs %zu workspace bytes, got %zu", need, workspace_bytes);
    return ODVAE_ERR_WORKSPACE;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  float* partial = static_cast<float*>(workspace);
  hipLaunchKernelGGL(gn_stats_kernel, dim3(s.chunks, N), dim3(256), 0, st, x, s, partial);
  ODVAE_LAUNCH_CHECK("groupnorm stats");
  hipLaunchKernelGGL(gn_finalize_kernel<GnShape>, dim3(ceil_div(N * G, 4)), dim3(256), 0, st, partial, s, eps, mean, rstd);
  ODVAE_LAUNCH_CHECK("groupnorm finalize");
  const dim3 grid(apply_blocks(s), N);
  if (swish) hipLaunchKernelGGL((gn_apply_kernel<true>), grid, dim3(256), 0, st, x, s, gamma, beta, mean, rstd, y);
  else       hipLaunchKernelGGL((gn_apply_kernel<false>), grid, dim3(256), 0, st, x, s, gamma, beta, mean, rstd, y);
  ODVAE_LAUNCH_CHECK("groupnorm apply");
  return ODVAE_OK;
}

// The same with the statistics pass already done by the kernel that produced x: partial [N][chunks][G][2] = (sum, sum of squares) of x
// per chunk and channel group (odvae_conv3x3_wino4_stats_f32 writes one chunk per output tile).  Two launches instead of three: finalize
// (f64, fixed order over the chunks) + apply; x is read once.
int odvae_groupnorm_fwd_partials_f32(const float* x, int N, int HW, int C, int G, const float* gamma, const float* beta,
                                     float eps, int swish, float* y, float* mean, float* rstd,
                                     const float* partial, int chunks, void* stream) {
  GnShape s;
  ODVAE_CHECK_ARG(make_shape(N, HW, C, G, s), "groupnorm_fwd_partials: unsupported shape N=%d HW=%d C=%d G=%d (need C%%G==0, C%%4==0, C<=1024)", N, HW, C, G);
  ODVAE_CHECK_ARG(x && gamma && beta && y && mean && rstd && partial && chunks > 0, "groupnorm_fwd_partials: null operand");
  ODVAE_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0 && ((uintptr_t)gamma & 15) == 0 && ((uintptr_t)beta & 15) == 0 &&
                  ((uintptr_t)partial & 7) == 0, "groupnorm_fwd_partials: operands must be 16-byte aligned (partial: 8)");
  hipStream_t st = static_cast<hipStream_t>(stream);
  GnShape sf = s;
  sf.chunks = chunks;
  hipLaunchKernelGGL(gn_finalize_kernel<GnShape>, dim3(ceil_div(N * G, 4)), dim3(256), 0, st, partial, sf, eps, mean, rstd);
  ODVAE_LAUNCH_CHECK("groupnorm finalize (partials)");
  const dim3 grid(apply_blocks(s), N);
  if (swish) hipLaunchKernelGGL((gn_apply_kernel<true>), grid, dim3(256), 0, st, x, s, gamma, beta, mean, rstd, y);
  else       hipLaunchKernelGGL((gn_apply_kernel<false>), grid, dim3(256), 0, st, x, s, gamma, beta, mean, rstd, y);
  ODVAE_LAUNCH_CHECK("groupnorm apply");
  return ODVAE_OK;
}

// dx, dgamma[C], dbeta[C] from dy (gradient w.r.t. the activated output), x and the saved mean/rstd
int odvae_groupnorm_bwd_f32(const float* x, const float* dy, int N, int HW, int C, int G,
                            const float* gamma, const float* beta, const float* mean, const float* rstd, int swish,
                            float* dx, float* dgamma, float* dbeta, const float* dx_add,
                            void* workspace, size_t workspace_bytes, void* stream) {
  GnShape s;
  ODVAE_CHECK_ARG(make_shape(N, HW, C, G, s), "groupnorm_bwd: unsupported shape N=%d HW=%d C=%d G=%d", N, HW, C, G);
  ODVAE_CHECK_ARG(x && dy && gamma && beta && mean && rstd && dx && dgamma && dbeta, "groupnorm_bwd: null operand");
  ODVAE_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)dy & 15) == 0 && ((uintptr_t)dx & 15) == 0, "groupnorm_bwd: operands must be 16-byte aligned");
  ODVAE_CHECK_ARG(((uintptr_t)dx_add & 15) == 0, "groupnorm_bwd: dx_add must be 16-byte aligned");
  const size_t need = odvae_groupnorm_workspace_bytes(N, HW, C, G);
  if (!workspace || workspace_bytes < need) {
    odvae_set_error("groupnorm_bwd: needs %zu workspace bytes, got %zu", need, workspace_bytes);
    return ODVAE_ERR_WORKSPACE;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  float* partial = static_cast<float*>(workspace);
  float* chan = partial + (size_t)N * s.chunks * 2 * C;
  float* grp = chan + (size_t)N * 2 * C;
  if (swish) hipLaunchKernelGGL((gn_bwd_reduce_kernel<true>), dim3(s.chunks, N), dim3(256), 0, st, x, dy, s, gamma, beta, mean, rstd, partial);
  else       hipLaunchKernelGGL((gn_bwd_reduce_kernel<false>), dim3(s.chunks, N), dim3(256), 0, st, x, dy, s, gamma, beta, mean, rstd, partial);
  ODVAE_LAUNCH_CHECK("groupnorm bwd reduce");
  if (s.cpg <= 64) hipLaunchKernelGGL(gn_bwd_finalize_kernel<GnShape>, dim3(N, ceil_div(C, (64 / s.cpg) * s.cpg)), dim3(256), 0, st, partial, s, gamma, chan, grp);
  else hipLaunchKernelGGL(gn_bwd_finalize_wide_kernel<GnShape>, dim3(N), dim3(256), 2 * C * sizeof(float), st, partial, s, gamma, chan, grp);
  ODVAE_LAUNCH_CHECK("groupnorm bwd finalize");
  hipLaunchKernelGGL(gn_bwd_param_kernel, dim3(ceil_div(C, 64)), dim3(256), 0, st, chan, N, C, dgamma, dbeta);
  ODVAE_LAUNCH_CHECK("groupnorm bwd param");
  const dim3 grid(apply_blocks(s), N);
  if (swish) hipLaunchKernelGGL((gn_bwd_apply_kernel<true>), grid, dim3(256), 0, st, x, dy, s, gamma, beta, mean, rstd, grp, dx_add, dx);
  else       hipLaunchKernelGGL((gn_bwd_apply_kernel<false>), grid, dim3(256), 0, st, x, dy, s, gamma, beta, mean, rstd, grp, dx_add, dx);
  ODVAE_LAUNCH_CHECK("groupnorm bwd apply");
  return ODVAE_OK;
}

}  // extern "C"
