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
#include <type_traits>

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

// streaming accesses of the apply passes: every byte is touched once by this kernel and next by another one a gigabyte later
#ifndef ODVAE_GN_NT
#define ODVAE_GN_NT 1
#endif
typedef float gn_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld_stream(const float4* p) {
#if ODVAE_GN_NT
  const gn_f4 v = __builtin_nontemporal_load(reinterpret_cast<const gn_f4*>(p));
  return make_float4(v.x, v.y, v.z, v.w);
#else
  return *p;
#endif
}
__device__ __forceinline__ void st_stream(float4* p, float4 v) {
#if ODVAE_GN_NT
  __builtin_nontemporal_store(gn_f4{v.x, v.y, v.z, v.w}, reinterpret_cast<gn_f4*>(p));
#else
  *p = v;
#endif
}

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
      const float4 v0 = ld_stream(xn + i), v1 = ld_stream(xn + i + stride), v2 = ld_stream(xn + i + 2 * stride), v3 = ld_stream(xn + i + 3 * stride);
      st_stream(yn + i, gn_apply_quad<SWISH>(v0, k)); st_stream(yn + i + stride, gn_apply_quad<SWISH>(v1, k));
      st_stream(yn + i + 2 * stride, gn_apply_quad<SWISH>(v2, k)); st_stream(yn + i + 3 * stride, gn_apply_quad<SWISH>(v3, k));
    }
    for (; i < per_n; i += stride) st_stream(yn + i, gn_apply_quad<SWISH>(ld_stream(xn + i), k));
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
      const float4 xv = ld_stream(reinterpret_cast<const float4*>(x + base + (int64_t)px * s.C));
      const float4 dv = ld_stream(reinterpret_cast<const float4*>(dy + base + (int64_t)px * s.C));
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
    if (an) { const float4 a = ld_stream(an + at); v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w; }
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
      const float4 x0 = ld_stream(xn + i), x1 = ld_stream(xn + i + stride), d0 = ld_stream(dn + i), d1 = ld_stream(dn + i + stride);
      st_stream(on + i, plus(gn_bwd_quad<SWISH>(x0, d0, k, inv_m), i)); st_stream(on + i + stride, plus(gn_bwd_quad<SWISH>(x1, d1, k, inv_m), i + stride));
    }
    for (; i < per_n; i += stride) st_stream(on + i, plus(gn_bwd_quad<SWISH>(ld_stream(xn + i), ld_stream(dn + i), k, inv_m), i));
  } else {
    for (; i < per_n; i += stride) {
      gn_load_quad(s, n, i % s.quads, gamma, beta, mean, rstd, grp, k);
      on[i] = plus(gn_bwd_quad<SWISH>(xn[i], dn[i], k, inv_m), i);
    }
  }
}


// ---- backward, x and dy read ONCE (default where the shape allows; odvae_groupnorm_select_backward) -----------------------
// The two-kernel form above streams x and dy twice: once for the per-(sample, group) sums, once to apply them (5-6 tensor passes
// for 3-4 algorithmic ones; 17 of the f32 step's 237 ms, 10 of the bf16 step's 79).  Here a block KEEPS its share of a (sample,
// channel slab) in registers between the two phases, and the blocks that share a sample exchange their 64 partial sums through L2:
//   item  = (sample n, slab of 32 channels = 128 contiguous bytes per pixel, whole channel groups);
//   team  = the T = ceil(HW / 256) blocks that hold an item's pixels; the grid is a fixed number of resident blocks (2 per CU),
//           team k walks items k, k + teams, ...
//   A thread's 128 data registers are TWO halves of 8 (x, dy) slot pairs, and a block works on two items at once, one per half:
//     phase A (half h): xhat and du = dy * act'(u) replace x / dy in the registers; per-channel sums of du * xhat and du meet in LDS,
//             64 floats per block go to partial[n][member][2][C] (write-through stores), one arrival on the item's counter -- no wait;
//     finish  (half h): one lane polls the counter (relaxed, agent scope, bounded), ONE agent-scope acquire, the workgroup barrier
//             (cdna_hip_programming.md Guideline 16: placement-independent); every member adds the team's partials in the same fixed
//             order (f64: bit-identical ds1 / ds2 in all of them); dx = rstd * (du * gamma - (ds2 + xhat * ds1) / m) (+ skip gradient);
//             each slot, once stored, is refilled with the x / dy of the half's next item.
//   Order per iteration: A(0), A(1), finish(0), finish(1).  A block has published BOTH its items before it waits for the first, the
//   refill of half 0 flies under the wait, sums and stores of half 1, and that of half 1 under the next iteration's phase A of half 0:
//   the memory queue of a CU never drains at a barrier (a first version with one item per block and a wait right behind its own
//   arrival ran at 0.7x the two-kernel form).
// dgamma / dbeta come from the same partials through the finalize / param kernels of the two-kernel form.
// Residency: every member of a team must be running for the barrier to complete.  The grid is sized from the occupancy the runtime
// reports for this kernel (<= 2 blocks per CU, x CUs), blocks of the previous kernel on the stream retire on their own, and every
// spin is bounded (s_memrealtime, 0.2 s): a wait that gives up adds to g_gn_fused_timeouts and lets the block run on with whatever
// partials are there -- wrong numbers, reported by odvae_groupnorm_fused_timeouts(), never a hang.
constexpr int GF_THREADS = 256, GF_SLAB = 32, GF_PMAX = 256, GF_HSLOTS = 8;

__device__ unsigned g_gn_fused_timeouts = 0;

struct GnFusedParams {
  const float* x; const float* dy; const float* dskip; float* dx;
  const float* gamma; const float* beta; const float* mean; const float* rstd;
  float* partial;        // [N][T][2][C]
  unsigned* counters;    // [items], zeroed by the launcher
  int N, HW, C, G, cpg, slabs, T, P, teams, items;
};

typedef __attribute__((address_space(1))) unsigned gu32_t;
typedef __attribute__((address_space(1))) float gf32_t;

template <bool SWISH>
__global__ __launch_bounds__(GF_THREADS, 2) void gn_bwd_fused_kernel(GnFusedParams p) {
  __shared__ float red[2][32][GF_SLAB + 1];     // [stat][pixel row of the block][channel]
  __shared__ double parts[16][2][GF_SLAB];      // the team's sums, sixteen member shares
  __shared__ float gsum[2][GF_SLAB];            // per channel: ds1 / ds2 of its group
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = tid & 7, pr = tid >> 3;         // channel quad of the slab, pixel row 0..31
  const int team = blockIdx.x / p.T, member = blockIdx.x - team * p.T;
  if (team >= p.teams) return;
  const int p_beg = member * p.P, p_end = min(p.HW, p_beg + p.P);
  const int img_bytes = p.HW * p.C * 4;
  const float inv_m = 1.f / ((float)p.HW * (float)p.cpg);
  constexpr unsigned OOBV = 0x7FFFFFF0u;

  u32x4 xr[2][GF_HSLOTS], dr[2][GF_HSLOTS];     // per half: x / dy, then xhat / du
  float gg[2][4], rs[2][4];                     // per half: gamma and rstd of the thread's four channels (phase B needs them again)
  // addressing: the descriptor's base carries (sample, first channel of the slab); a thread's byte offset inside it is the same for
  // every item -- slot i of the thread is pixel p_beg + 32 i + pr, i.e. v0 + i * step with the step as the instruction's scalar offset --
  // and slots past the block's pixel range read as zeros / drop their stores through an out-of-range offset
  const unsigned v0 = (unsigned)(((p_beg + pr) * p.C + 4 * q) * 4);
  const int step = 32 * p.C * 4;
  const int nvalid = (p_end - p_beg - pr + 31) / 32;      // slots 0 .. nvalid-1 of this thread hold pixels
  auto voff_of = [&](int i) { return i < nvalid ? v0 : OOBV; };
  auto rsrc_of = [&](const float* base, int n, int c0, bool live) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base) + (int64_t)n * p.HW * p.C + c0, 0, live ? img_bytes - c0 * 4 : 0, 0x00020000);
  };
  auto item_n = [&](int it) { return it / p.slabs; };
  auto item_c0 = [&](int it) { return (it - (it / p.slabs) * p.slabs) * GF_SLAB; };

  auto load_half = [&](auto H, int it) {
    constexpr int h = decltype(H)::value;
    const bool live = it < p.items;
    const int n = live ? item_n(it) : 0, c0 = live ? item_c0(it) : 0;
    const __amdgpu_buffer_rsrc_t xs = rsrc_of(p.x, n, c0, live), ds = rsrc_of(p.dy, n, c0, live);
#pragma unroll
    for (int i = 0; i < GF_HSLOTS; ++i) {
      xr[h][i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(xs, voff_of(i), i * step, 0));
      dr[h][i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(ds, voff_of(i), i * step, 0));
    }
  };

  // phase A of one half: registers -> (xhat, du); the block's 64 sums -> partial; one arrival.  Ends with the block's barrier.
  auto phase_a = [&](auto H, int it) {
    constexpr int h = decltype(H)::value;
    const int n = item_n(it), c0 = item_c0(it), c = c0 + 4 * q;
    float bb[4], mu[4];
    {
      const float4 ga = *reinterpret_cast<const float4*>(p.gamma + c), be = *reinterpret_cast<const float4*>(p.beta + c);
      gg[h][0] = ga.x; gg[h][1] = ga.y; gg[h][2] = ga.z; gg[h][3] = ga.w; bb[0] = be.x; bb[1] = be.y; bb[2] = be.z; bb[3] = be.w;
#pragma unroll
      for (int j = 0; j < 4; ++j) { const int g = (c + j) / p.cpg; mu[j] = p.mean[n * p.G + g]; rs[h][j] = p.rstd[n * p.G + g]; }
    }
    float a[4] = {0.f, 0.f, 0.f, 0.f}, b[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < GF_HSLOTS; ++i) {
      float xv[4] = {__uint_as_float(xr[h][i].x), __uint_as_float(xr[h][i].y), __uint_as_float(xr[h][i].z), __uint_as_float(xr[h][i].w)};
      float dv[4] = {__uint_as_float(dr[h][i].x), __uint_as_float(dr[h][i].y), __uint_as_float(dr[h][i].z), __uint_as_float(dr[h][i].w)};
      const bool live = i < nvalid;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float xh = (xv[j] - mu[j]) * rs[h][j];
        const float du = live ? dv[j] * act_grad<SWISH>(xh * gg[h][j] + bb[j]) : 0.f;
        a[j] += du * xh; b[j] += du;
        xv[j] = xh; dv[j] = du;
      }
      xr[h][i] = u32x4{__float_as_uint(xv[0]), __float_as_uint(xv[1]), __float_as_uint(xv[2]), __float_as_uint(xv[3])};
      dr[h][i] = u32x4{__float_as_uint(dv[0]), __float_as_uint(dv[1]), __float_as_uint(dv[2]), __float_as_uint(dv[3])};
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { red[0][pr][4 * q + j] = a[j]; red[1][pr][4 * q + j] = b[j]; }
    __syncthreads();
    if (wave == 0) {      // thread = (stat, channel): the block's 64 sums, pixel rows added in fixed order
      const int which = lane >> 5, cc = lane & 31;
      float sacc = 0.f;
#pragma unroll 8
      for (int r = 0; r < 32; ++r) sacc += red[which][r][cc];
      float* dst = p.partial + (((int64_t)n * p.T + member) * 2 + which) * p.C + c0 + cc;
      __hip_atomic_store((gf32_t*)dst, sacc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // write-through (sc1)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the wave's refill loads of the other half are waited for here too)
      if (lane == 0) __hip_atomic_fetch_add((gu32_t*)(p.counters + it), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();      // red is free again
  };

  // the rest of one half: team barrier, team sums, dx, refill with item `nxt`
  auto finish = [&](auto H, int it, int nxt) {
    constexpr int h = decltype(H)::value;
    const int n = item_n(it), c0 = item_c0(it);
    if (tid == 0) {
      gu32_t* cnt = (gu32_t*)(p.counters + it);
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
      while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)p.T) {
        __builtin_amdgcn_s_sleep(4);
        if (__builtin_amdgcn_s_memrealtime() - t0 > 20000000ull) {       // 0.2 s at 100 MHz: give up, report, carry on
          __hip_atomic_fetch_add((gu32_t*)&g_gn_fused_timeouts, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    // the team's sums: thread = (member share 0..15, stat, channel quad); 16-byte loads, all of a thread's in flight together
    {
      const int ms = tid >> 4, which = (tid >> 3) & 1, qq = tid & 7;
      const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(p.partial + (int64_t)n * p.T * 2 * p.C, 0, p.T * 2 * p.C * 4, 0x00020000);
      const unsigned po = (unsigned)(((ms * 2 + which) * p.C + c0 + 4 * qq) * 4);
      const int pstep = 16 * 2 * p.C * 4;
      double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
      for (int k0 = 0; k0 * 16 < p.T; k0 += 8) {
        u32x4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k)      // aux 16 = sc1: served by L2, never by this CU's L1; members past T read as zeros (range check)
          v[k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(prs, (ms + 16 * (k0 + k)) < p.T ? po : OOBV, (k0 + k) * pstep, 16));
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          s0 += (double)__uint_as_float(v[k].x); s1 += (double)__uint_as_float(v[k].y);
          s2 += (double)__uint_as_float(v[k].z); s3 += (double)__uint_as_float(v[k].w);
        }
      }
      parts[ms][which][4 * qq + 0] = s0; parts[ms][which][4 * qq + 1] = s1; parts[ms][which][4 * qq + 2] = s2; parts[ms][which][4 * qq + 3] = s3;
    }
    __syncthreads();
    if (tid < 64) {       // per channel: the sixteen shares in fixed order, times gamma, then the channels of a group by a fixed butterfly
      const int which = tid >> 5, cc = tid & 31;
      double tot = 0.0;
#pragma unroll
      for (int m = 0; m < 16; ++m) tot += parts[m][which][cc];
      float v = (float)(tot * (double)p.gamma[c0 + cc]);
      for (int o = 1; o < p.cpg; o <<= 1) v += __shfl_xor(v, o, 64);
      gsum[which][cc] = v;
    }
    __syncthreads();
    float ds1[4], ds2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { ds1[j] = gsum[0][4 * q + j] * inv_m; ds2[j] = gsum[1][4 * q + j] * inv_m; }
    const bool more = nxt < p.items;
    const int n2 = more ? item_n(nxt) : 0, c02 = more ? item_c0(nxt) : 0;
    const __amdgpu_buffer_rsrc_t os = rsrc_of(p.dx, n, c0, true), ks = rsrc_of(p.dskip ? p.dskip : p.x, n, c0, p.dskip != nullptr);
    const __amdgpu_buffer_rsrc_t xs2 = rsrc_of(p.x, n2, c02, more), ds2r = rsrc_of(p.dy, n2, c02, more);
#pragma unroll
    for (int i0 = 0; i0 < GF_HSLOTS; i0 += 4) {
      u32x4 sk[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) sk[k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(ks, voff_of(i0 + k), (i0 + k) * step, 0));
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int i = i0 + k;
        const float xv[4] = {__uint_as_float(xr[h][i].x), __uint_as_float(xr[h][i].y), __uint_as_float(xr[h][i].z), __uint_as_float(xr[h][i].w)};
        const float dv[4] = {__uint_as_float(dr[h][i].x), __uint_as_float(dr[h][i].y), __uint_as_float(dr[h][i].z), __uint_as_float(dr[h][i].w)};
        const float kv[4] = {__uint_as_float(sk[k].x), __uint_as_float(sk[k].y), __uint_as_float(sk[k].z), __uint_as_float(sk[k].w)};
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = rs[h][j] * (dv[j] * gg[h][j] - (ds2[j] + xv[j] * ds1[j])) + kv[j];
        // (the slot offset rides in the VECTOR offset here, not in soffset: LLVM's hazard recogniser skips the "VMEM store of more than
        // 64 bits followed by a VALU write of its data registers" wait states when soffset is a register, and on gfx950 the hazard is
        // real -- with `..., voff, i * step` the second dword of lanes 12-15 of every 16 carried the NEXT slot's value)
        __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(o[0]), __float_as_uint(o[1]), __float_as_uint(o[2]), __float_as_uint(o[3])},
                                               os, voff_of(i) + (unsigned)(i * step), 0, 0);
      }
      if (more) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int i = i0 + k;
          xr[h][i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(xs2, voff_of(i), i * step, 0));
          dr[h][i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(ds2r, voff_of(i), i * step, 0));
        }
      }
    }
  };

  using H0 = std::integral_constant<int, 0>;
  using H1 = std::integral_constant<int, 1>;
  int it0 = team, it1 = team + p.teams;          // half 0 walks the team's even items, half 1 the odd ones
  load_half(H0{}, it0);
  load_half(H1{}, it1);
  const int stride2 = 2 * p.teams;
  for (; it0 < p.items; it0 += stride2, it1 += stride2) {
    const bool two = it1 < p.items;
    phase_a(H0{}, it0);
    if (two) phase_a(H1{}, it1);
    finish(H0{}, it0, it0 + stride2);
    if (two) finish(H1{}, it1, it1 + stride2);
  }
}

struct GnFusedPlan { bool ok; int T, P, teams, items, slabs, grid; size_t counter_bytes; };

// -1: fused where the shape allows (default), 0: always the two-kernel form, 1: fused or an error
int g_gn_bwd_mode = -1;

template <typename K>
int gn_fused_blocks_per_cu(K kernel) {
  int nb = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, GF_THREADS, 0) != hipSuccess) return 0;
  return std::min(nb, 2);
}

GnFusedPlan gn_fused_plan(const GnShape& s, bool swish) {
  GnFusedPlan pl{};
  static const int cus = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 0;
    return n;
  }();
  static const int bpc_t = gn_fused_blocks_per_cu(gn_bwd_fused_kernel<true>), bpc_f = gn_fused_blocks_per_cu(gn_bwd_fused_kernel<false>);
  const int resident = cus * (swish ? bpc_t : bpc_f);
  if (resident <= 0 || s.C % GF_SLAB != 0 || s.cpg > GF_SLAB || GF_SLAB % s.cpg != 0 || (int64_t)s.HW * s.C * 4 >= 0x7FFFFFF0ll) return pl;
  pl.T = ceil_div(s.HW, GF_PMAX);      // a block holds <= 256 pixels of an item per register half
  if (pl.T > resident) return pl;
  pl.P = ceil_div(s.HW, pl.T);
  pl.slabs = s.C / GF_SLAB;
  pl.items = s.N * pl.slabs;
  pl.teams = std::min(resident / pl.T, pl.items);
  pl.grid = pl.teams * pl.T;
  pl.counter_bytes = ((size_t)pl.items * 4 + 255) / 256 * 256;
  pl.ok = true;
  return pl;
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
  size_t bytes = (fwd > bwd ? fwd : bwd) * sizeof(float);
  const GnFusedPlan pl = gn_fused_plan(s, true);      // read-once backward: arrival counters + [N][T][2][C] partials + chan + grp
  if (pl.ok) bytes = std::max(bytes, pl.counter_bytes + ((size_t)N * pl.T * 2 * C + (size_t)N * 2 * C + (size_t)N * G * 2) * sizeof(float));
  return bytes;
}

// backward form: -1 (default) = the read-once kernel where one block holds a whole (sample, channel slab) -- HW <= 256 --, reduce + apply
// (two reads of x and dy) elsewhere; 0 = always reduce + apply; 1 = the read-once kernel with teams of blocks on every shape it takes,
// or ODVAE_ERR_ARG.  Returns the previous setting.
int odvae_groupnorm_select_backward(int mode) {
  const int prev = g_gn_bwd_mode;
  g_gn_bwd_mode = mode < -1 ? -1 : (mode > 1 ? 1 : mode);
  return prev;
}

// Waits of the fused kernels' team barriers that gave up since the library was loaded (synchronises the device).  Anything but 0 means
// a launch ran with team members not resident together: its results are wrong.
int odvae_groupnorm_fused_timeouts(void) {
  unsigned v = 0;
  if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_gn_fused_timeouts), sizeof(v)) != hipSuccess) return -1;
  return (int)v;
}

int odvae_attn_softmax_fallbacks(int add);   // elementwise.hip

// Both device-side health counters in one call (synchronises the device): gn_timeouts != 0 means a GroupNorm backward ran with a team
// barrier that gave up -- its gradients are WRONG and the run must stop; attn_fallbacks counts exact-softmax fallbacks (correct, slower).
// inject_gn_timeouts / inject_attn_fallbacks are TEST HOOKS: > 0 bumps the counter first, inject_gn_timeouts < 0 clears it.  ODVAE_ERR_HIP if a counter cannot be read.
int odvae_device_health(int* gn_timeouts, int* attn_fallbacks, int inject_gn_timeouts, int inject_attn_fallbacks) {
  unsigned v = 0;
  if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_gn_fused_timeouts), sizeof(v)) != hipSuccess) {
    odvae_set_error("device_health: cannot read the GroupNorm barrier counter");
    return ODVAE_ERR_HIP;
  }
  if (inject_gn_timeouts != 0) {      // > 0: add (a simulated timeout); < 0: clear the counter (tests restore the state they found)
    v = inject_gn_timeouts > 0 ? v + (unsigned)inject_gn_timeouts : 0u;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_gn_fused_timeouts), &v, sizeof(v)) != hipSuccess) {
      odvae_set_error("device_health: cannot write the GroupNorm barrier counter");
      return ODVAE_ERR_HIP;
    }
  }
  const int f = odvae_attn_softmax_fallbacks(inject_attn_fallbacks > 0 ? inject_attn_fallbacks : 0);
  if (f < 0) {
    odvae_set_error("device_health: cannot read the folded-softmax fallback counter");
    return ODVAE_ERR_HIP;
  }
  if (gn_timeouts) *gn_timeouts = (int)v;
  if (attn_fallbacks) *attn_fallbacks = f;
  return ODVAE_OK;
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

// The apply pass alone, with mean / rstd given (as a forward call left them): y = act(GroupNorm(x)) re-made from x -- the recompute of
// the "norm" activation-checkpoint policy (modules.py: the conv that consumed y keeps (x, mean, rstd) instead of y).
int odvae_groupnorm_apply_f32(const float* x, int N, int HW, int C, int G, const float* gamma, const float* beta,
                              const float* mean, const float* rstd, int swish, float* y, void* stream) {
  GnShape s;
  ODVAE_CHECK_ARG(make_shape(N, HW, C, G, s), "groupnorm_apply: unsupported shape N=%d HW=%d C=%d G=%d", N, HW, C, G);
  ODVAE_CHECK_ARG(x && gamma && beta && y && mean && rstd, "groupnorm_apply: null operand");
  ODVAE_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0 && ((uintptr_t)gamma & 15) == 0 && ((uintptr_t)beta & 15) == 0,
                  "groupnorm_apply: operands must be 16-byte aligned");
  const dim3 grid(apply_blocks(s), N);
  hipStream_t st = static_cast<hipStream_t>(stream);
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
  const GnFusedPlan pl = g_gn_bwd_mode != 0 ? gn_fused_plan(s, swish != 0) : GnFusedPlan{};
  ODVAE_CHECK_ARG(g_gn_bwd_mode != 1 || pl.ok, "groupnorm_bwd: the fused form does not take N=%d HW=%d C=%d G=%d", N, HW, C, G);
  // Default (-1): the read-once kernel only where ONE block holds an item (HW <= 256: no barrier between blocks at all; measured 1.2x
  // the two-kernel form at 512 ch @16x16).  With teams it LOSES on this chip: 0.53-0.8x on the 64x64 .. 256x256 levels (B = 32) -- a
  // block can hold 128 KB, which streams in ~12 us, and every team barrier costs ~25 us of store drain, counter round trip, acquire,
  // arrival skew and partial-sum reads that two blocks per CU cannot hide (profiles/r04_groupnorm_read_once.txt).
  if (pl.ok && (g_gn_bwd_mode == 1 || pl.T == 1)) {
    GnFusedParams fp;
    fp.x = x; fp.dy = dy; fp.dskip = dx_add; fp.dx = dx; fp.gamma = gamma; fp.beta = beta; fp.mean = mean; fp.rstd = rstd;
    fp.counters = static_cast<unsigned*>(workspace);
    fp.partial = reinterpret_cast<float*>(static_cast<char*>(workspace) + pl.counter_bytes);
    fp.N = N; fp.HW = HW; fp.C = C; fp.G = G; fp.cpg = s.cpg; fp.slabs = pl.slabs; fp.T = pl.T; fp.P = pl.P; fp.teams = pl.teams; fp.items = pl.items;
    if (hipMemsetAsync(fp.counters, 0, pl.counter_bytes, st) != hipSuccess) { odvae_set_error("groupnorm_bwd: memset of the arrival counters failed"); return ODVAE_ERR_HIP; }
    if (swish) hipLaunchKernelGGL((gn_bwd_fused_kernel<true>), dim3(pl.grid), dim3(GF_THREADS), 0, st, fp);
    else       hipLaunchKernelGGL((gn_bwd_fused_kernel<false>), dim3(pl.grid), dim3(GF_THREADS), 0, st, fp);
    ODVAE_LAUNCH_CHECK("groupnorm bwd fused");
    GnShape sf = s;
    sf.chunks = pl.T;      // the partials have the two-kernel form's layout with one chunk per team member
    float* chan_f = fp.partial + (size_t)N * pl.T * 2 * C;
    float* grp_f = chan_f + (size_t)N * 2 * C;
    hipLaunchKernelGGL(gn_bwd_finalize_kernel<GnShape>, dim3(N, ceil_div(C, (64 / s.cpg) * s.cpg)), dim3(256), 0, st, fp.partial, sf, gamma, chan_f, grp_f);
    ODVAE_LAUNCH_CHECK("groupnorm bwd finalize (fused)");
    hipLaunchKernelGGL(gn_bwd_param_kernel, dim3(ceil_div(C, 64)), dim3(256), 0, st, chan_f, N, C, dgamma, dbeta);
    ODVAE_LAUNCH_CHECK("groupnorm bwd param (fused)");
    return ODVAE_OK;
  }
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

// The same with the first pass already done: partial [N][chunks][2][C] = per chunk of pixels and channel (sum du * xhat, sum du), as the
// data-gradient launch that produced dy left it (odvae_conv3x3_wino4_gnbwd_f32: one chunk per output tile).  x and dy are read once.
int odvae_groupnorm_bwd_partials_f32(const float* x, const float* dy, int N, int HW, int C, int G,
                                     const float* gamma, const float* beta, const float* mean, const float* rstd, int swish,
                                     float* dx, float* dgamma, float* dbeta, const float* dx_add, const float* partial, int chunks,
                                     void* workspace, size_t workspace_bytes, void* stream) {
  GnShape s;
  ODVAE_CHECK_ARG(make_shape(N, HW, C, G, s), "groupnorm_bwd_partials: unsupported shape N=%d HW=%d C=%d G=%d", N, HW, C, G);
  ODVAE_CHECK_ARG(x && dy && gamma && beta && mean && rstd && dx && dgamma && dbeta && partial && chunks > 0, "groupnorm_bwd_partials: null operand");
  ODVAE_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)dy & 15) == 0 && ((uintptr_t)dx & 15) == 0 && ((uintptr_t)dx_add & 15) == 0,
                  "groupnorm_bwd_partials: operands must be 16-byte aligned");
  const size_t need = ((size_t)N * 2 * C + (size_t)N * G * 2) * sizeof(float);
  if (!workspace || workspace_bytes < need) {
    odvae_set_error("groupnorm_bwd_partials: needs %zu workspace bytes, got %zu", need, workspace_bytes);
    return ODVAE_ERR_WORKSPACE;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  float* chan = static_cast<float*>(workspace);
  float* grp = chan + (size_t)N * 2 * C;
  GnShape sp = s;
  sp.chunks = chunks;
  if (s.cpg <= 64) hipLaunchKernelGGL(gn_bwd_finalize_kernel<GnShape>, dim3(N, ceil_div(C, (64 / s.cpg) * s.cpg)), dim3(256), 0, st, partial, sp, gamma, chan, grp);
  else hipLaunchKernelGGL(gn_bwd_finalize_wide_kernel<GnShape>, dim3(N), dim3(256), 2 * C * sizeof(float), st, partial, sp, gamma, chan, grp);
  ODVAE_LAUNCH_CHECK("groupnorm bwd finalize (partials)");
  hipLaunchKernelGGL(gn_bwd_param_kernel, dim3(ceil_div(C, 64)), dim3(256), 0, st, chan, N, C, dgamma, dbeta);
  ODVAE_LAUNCH_CHECK("groupnorm bwd param (partials)");
  const dim3 grid(apply_blocks(s), N);
  if (swish) hipLaunchKernelGGL((gn_bwd_apply_kernel<true>), grid, dim3(256), 0, st, x, dy, s, gamma, beta, mean, rstd, grp, dx_add, dx);
  else       hipLaunchKernelGGL((gn_bwd_apply_kernel<false>), grid, dim3(256), 0, st, x, dy, s, gamma, beta, mean, rstd, grp, dx_add, dx);
  ODVAE_LAUNCH_CHECK("groupnorm bwd apply (partials)");
  return ODVAE_OK;
}

}  // extern "C"
