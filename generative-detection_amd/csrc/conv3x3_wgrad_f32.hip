// Weight (and bias) gradient of the 3x3 convolutions, NHWC, exact f32 on the matrix cores.
//
// dW[co][ci][kh][kw] = sum over (n, oy, ox) of xs(n,oy,ox;kh,kw)[ci] * dy[n][oy][ox][co], where xs
// is the input pixel the forward pass of that mode reads (zero outside the image):
//   MODE 0 stride 1 pad 1 | MODE 1 pad(0,1,0,1)+stride 2 | MODE 2 nearest-2x-upsample + stride 1 pad 1
// ([UPSTREAM] ldm/modules/diffusionmodules/model.py ResnetBlock / Downsample / Upsample convs; what
// loss.backward() computes for them in the reference, src/models/autoencoder.py:295-330).
//
// GEMM view per tap: M = ci, N = co, K = pixels (up to N*Ho*Wo = 2M).  A block owns a 64(ci) x 64(co)
// slice of all 9 taps (each wave 32x32 x 9 taps = 144 accumulator registers) and a contiguous range of
// 8x16 (MODE 1: 4x16) output-pixel tiles.  Per tile it stages the input halo patch [halo px][64 ci] and
// the dy tile [px][64 co] in LDS; both are read with lanes = consecutive channels (conflict-free b32),
// one dy read feeds nine MFMAs.  Blocks write partial slabs; a second kernel sums them in a fixed order
// (deterministic) and writes OIHW.  db = column sums of dy ride along in the ci-tile-0 blocks.
#include "common.h"

namespace {

constexpr int TW = 16;
constexpr int BC = 64;  // channels per block on both axes

struct WgradParams {
  const float* x;    // [N][Hi][Wi][Cin]
  const float* dy;   // [N][Ho][Wo][Cout]
  float* slab;       // [nsplit][9][CinP][CoutP]
  float* bslab;      // [nsplit][CoutP] or null
  int N, Hi, Wi, Cin, Ho, Wo, Cout, CinP, CoutP;
  int tiles_x, tiles_y, ntiles, tiles_per_split, ci_tiles;
};

template <int MODE, int TH> struct Halo;
template <int TH> struct Halo<0, TH> { static constexpr int H = TH + 2, W = TW + 2; };
template <int TH> struct Halo<1, TH> { static constexpr int H = 2 * TH + 1, W = 2 * TW + 1; };
template <int TH> struct Halo<2, TH> { static constexpr int H = TH / 2 + 2, W = TW / 2 + 2; };

template <int MODE, int TH>
__device__ __forceinline__ int halo_index(int r, int c, int kh, int kw) {
  if (MODE == 0) return (r + kh) * Halo<0, TH>::W + (c + kw);
  if (MODE == 1) return (2 * r + kh) * Halo<1, TH>::W + (2 * c + kw);
  return ((r + kh + 1) >> 1) * Halo<2, TH>::W + ((c + kw + 1) >> 1);
}

template <int MODE, int TH>
__global__ __launch_bounds__(256) void conv3x3_wgrad_kernel(WgradParams p) {
  constexpr int HPIX = Halo<MODE, TH>::H * Halo<MODE, TH>::W;
  constexpr int NPIX = TH * TW;
  constexpr int HALO_F4 = HPIX * (BC / 4), HALO_IT = (HALO_F4 + 255) / 256;
  constexpr int DY_F4 = NPIX * (BC / 4), DY_IT = (DY_F4 + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Xs = smem;                 // [HPIX][BC]
  float* Ds = smem + HPIX * BC;     // [NPIX][BC]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, h = lane >> 5;
  const int split = blockIdx.x;
  const int ci_tile = blockIdx.y % p.ci_tiles, co_tile = blockIdx.y / p.ci_tiles;
  const int ci0 = ci_tile * BC, co0 = co_tile * BC;
  const bool xvec = (p.Cin & 3) == 0, dvec = (p.Cout & 3) == 0;
  const bool do_bias = p.bslab != nullptr && ci_tile == 0 && wm == 0;

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float bsum = 0.f;

  const int t_beg = split * p.tiles_per_split;
  const int t_end = min(p.ntiles, t_beg + p.tiles_per_split);
  for (int tile = t_beg; tile < t_end; ++tile) {
    int t = tile;
    const int tx = t % p.tiles_x; t /= p.tiles_x;
    const int ty = t % p.tiles_y; const int n = t / p.tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TW;
    int iy0, ix0;
    if (MODE == 0) { iy0 = oy0 - 1; ix0 = ox0 - 1; }
    else if (MODE == 1) { iy0 = 2 * oy0; ix0 = 2 * ox0; }
    else { iy0 = oy0 / 2 - 1; ix0 = ox0 / 2 - 1; }
    const float* xn = p.x + (int64_t)n * p.Hi * p.Wi * p.Cin;
    const float* dn = p.dy + (int64_t)n * p.Ho * p.Wo * p.Cout;

    __syncthreads();  // previous tile's reads are done
#pragma unroll
    for (int i = 0; i < HALO_IT; ++i) {
      const int f = tid + 256 * i;
      if (f < HALO_F4) {
        const int hp = f / (BC / 4), q = f % (BC / 4);
        const int iy = iy0 + hp / Halo<MODE, TH>::W, ix = ix0 + hp % Halo<MODE, TH>::W;
        const int c = ci0 + 4 * q;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi && c < p.Cin) {
          const float* src = xn + ((int64_t)iy * p.Wi + ix) * p.Cin + c;
          if (xvec) v = *reinterpret_cast<const float4*>(src);
          else {
            v.x = src[0];
            if (c + 1 < p.Cin) v.y = src[1];
            if (c + 2 < p.Cin) v.z = src[2];
            if (c + 3 < p.Cin) v.w = src[3];
          }
        }
        *reinterpret_cast<float4*>(Xs + hp * BC + 4 * q) = v;
      }
    }
#pragma unroll
    for (int i = 0; i < DY_IT; ++i) {
      const int f = tid + 256 * i;
      if (f < DY_F4) {
        const int px = f / (BC / 4), q = f % (BC / 4);
        const int oy = oy0 + px / TW, ox = ox0 + px % TW;
        const int c = co0 + 4 * q;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (oy < p.Ho && ox < p.Wo && c < p.Cout) {
          const float* src = dn + ((int64_t)oy * p.Wo + ox) * p.Cout + c;
          if (dvec) v = *reinterpret_cast<const float4*>(src);
          else {
            v.x = src[0];
            if (c + 1 < p.Cout) v.y = src[1];
            if (c + 2 < p.Cout) v.z = src[2];
            if (c + 3 < p.Cout) v.w = src[3];
          }
        }
        *reinterpret_cast<float4*>(Ds + px * BC + 4 * q) = v;
      }
    }
    __syncthreads();

    const float* xa = Xs + wm * 32 + li;
    const float* db = Ds + wn * 32 + li;
#pragma unroll 4
    for (int s = 0; s < NPIX / 2; ++s) {
      const int px = 2 * s + h;
      const int r = px / TW, c = px % TW;
      const float b = db[px * BC];
      if (do_bias) bsum += b;
#pragma unroll
      for (int t9 = 0; t9 < 9; ++t9) {
        const float a = xa[halo_index<MODE, TH>(r, c, t9 / 3, t9 % 3) * BC];
        acc[t9] = mfma32(a, b, acc[t9]);
      }
    }
  }

  // partial slab: [split][tap][ci][co], lane = co
  const int co = co0 + wn * 32 + li;
  float* sl = p.slab + (int64_t)split * 9 * p.CinP * p.CoutP;
#pragma unroll
  for (int t9 = 0; t9 < 9; ++t9)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ci = ci0 + wm * 32 + acc_row(r, lane);
      sl[((int64_t)t9 * p.CinP + ci) * p.CoutP + co] = acc[t9][r];
    }
  if (do_bias) {
    bsum += __shfl_xor(bsum, 32, 64);
    if (h == 0) p.bslab[(int64_t)split * p.CoutP + co] = bsum;
  }
}

__global__ void conv3x3_wgrad_reduce_kernel(const float* __restrict__ slab, const float* __restrict__ bslab,
                                            int nsplit, int Cin, int Cout, int CinP, int CoutP,
                                            float* __restrict__ dw, float* __restrict__ dbias) {
  const int64_t per = (int64_t)9 * CinP * CoutP;
  const int64_t total = per + (dbias ? CoutP : 0);
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    if (idx < per) {
      const int co = (int)(idx % CoutP);
      const int ci = (int)((idx / CoutP) % CinP);
      const int tap = (int)(idx / ((int64_t)CoutP * CinP));
      if (co < Cout && ci < Cin) {
        float s = 0.f;
        for (int sp = 0; sp < nsplit; ++sp) s += slab[sp * per + idx];
        dw[((int64_t)co * Cin + ci) * 9 + tap] = s;
      }
    } else {
      const int co = (int)(idx - per);
      if (co < Cout) {
        float s = 0.f;
        for (int sp = 0; sp < nsplit; ++sp) s += bslab[(int64_t)sp * CoutP + co];
        dbias[co] = s;
      }
    }
  }
}

struct Plan { int th, tiles_x, tiles_y, ntiles, nsplit, tiles_per_split, CinP, CoutP, ci_tiles, co_tiles; };

Plan make_plan(int mode, int N, int Ho, int Wo, int Cin, int Cout) {
  Plan pl;
  pl.th = mode == 1 ? 4 : 8;
  pl.tiles_x = ceil_div(Wo, TW);
  pl.tiles_y = ceil_div(Ho, pl.th);
  pl.ntiles = pl.tiles_x * pl.tiles_y * N;
  pl.CinP = ceil_div(Cin, BC) * BC;
  pl.CoutP = ceil_div(Cout, BC) * BC;
  pl.ci_tiles = pl.CinP / BC;
  pl.co_tiles = pl.CoutP / BC;
  const int ctiles = pl.ci_tiles * pl.co_tiles;
  int nsplit = ceil_div(1024, ctiles);
  if (nsplit > pl.ntiles) nsplit = pl.ntiles;
  if (nsplit < 1) nsplit = 1;
  pl.tiles_per_split = ceil_div(pl.ntiles, nsplit);
  pl.nsplit = ceil_div(pl.ntiles, pl.tiles_per_split);
  return pl;
}

template <int MODE, int TH>
size_t wgrad_smem_bytes() { return (size_t)(Halo<MODE, TH>::H * Halo<MODE, TH>::W + TH * TW) * BC * sizeof(float); }

}  // namespace

extern "C" {

size_t odvae_conv3x3_wgrad_workspace_bytes(int mode, int N, int Ho, int Wo, int Cin, int Cout) {
  const Plan pl = make_plan(mode, N, Ho, Wo, Cin, Cout);
  return ((size_t)pl.nsplit * 9 * pl.CinP * pl.CoutP + (size_t)pl.nsplit * pl.CoutP) * sizeof(float);
}

// dw: OIHW [Cout][Cin][3][3] (overwritten).  dbias: [Cout] or null.
int odvae_conv3x3_wgrad_f32(int mode, const float* x, const float* dy, int N, int Hi, int Wi, int Cin,
                            int Ho, int Wo, int Cout, float* dw, float* dbias,
                            void* workspace, size_t workspace_bytes, void* stream) {
  ODVAE_CHECK_ARG(x && dy && dw, "conv3x3_wgrad: null operand");
  ODVAE_CHECK_ARG(mode >= 0 && mode <= 2, "conv3x3_wgrad: mode %d", mode);
  ODVAE_CHECK_ARG(N > 0 && Hi > 0 && Wi > 0 && Cin > 0 && Cout > 0, "conv3x3_wgrad: empty shape");
  if (mode == 0) ODVAE_CHECK_ARG(Ho == Hi && Wo == Wi, "conv3x3_wgrad mode 0: Ho,Wo must equal Hi,Wi");
  if (mode == 1) ODVAE_CHECK_ARG(Hi % 2 == 0 && Wi % 2 == 0 && Ho == Hi / 2 && Wo == Wi / 2, "conv3x3_wgrad mode 1: need even Hi,Wi");
  if (mode == 2) ODVAE_CHECK_ARG(Ho == 2 * Hi && Wo == 2 * Wi, "conv3x3_wgrad mode 2: need Ho=2*Hi");
  ODVAE_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)dy & 15) == 0, "conv3x3_wgrad: x/dy must be 16-byte aligned");
  const Plan pl = make_plan(mode, N, Ho, Wo, Cin, Cout);
  const size_t need = odvae_conv3x3_wgrad_workspace_bytes(mode, N, Ho, Wo, Cin, Cout);
  if (!workspace || workspace_bytes < need) {
    odvae_set_error("conv3x3_wgrad: needs %zu workspace bytes, got %zu", need, workspace_bytes);
    return ODVAE_ERR_WORKSPACE;
  }
  WgradParams p;
  p.x = x; p.dy = dy;
  p.slab = static_cast<float*>(workspace);
  p.bslab = dbias ? p.slab + (size_t)pl.nsplit * 9 * pl.CinP * pl.CoutP : nullptr;
  p.N = N; p.Hi = Hi; p.Wi = Wi; p.Cin = Cin; p.Ho = Ho; p.Wo = Wo; p.Cout = Cout;
  p.CinP = pl.CinP; p.CoutP = pl.CoutP;
  p.tiles_x = pl.tiles_x; p.tiles_y = pl.tiles_y; p.ntiles = pl.ntiles;
  p.tiles_per_split = pl.tiles_per_split; p.ci_tiles = pl.ci_tiles;
  hipStream_t st = static_cast<hipStream_t>(stream);
  dim3 grid(pl.nsplit, pl.ci_tiles * pl.co_tiles), block(256);
  hipError_t e = hipSuccess;
  if (mode == 0) {
    const size_t sm = wgrad_smem_bytes<0, 8>();
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wgrad_kernel<0, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
    if (e == hipSuccess) hipLaunchKernelGGL((conv3x3_wgrad_kernel<0, 8>), grid, block, sm, st, p);
  } else if (mode == 1) {
    const size_t sm = wgrad_smem_bytes<1, 4>();
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wgrad_kernel<1, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
    if (e == hipSuccess) hipLaunchKernelGGL((conv3x3_wgrad_kernel<1, 4>), grid, block, sm, st, p);
  } else {
    const size_t sm = wgrad_smem_bytes<2, 8>();
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wgrad_kernel<2, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
    if (e == hipSuccess) hipLaunchKernelGGL((conv3x3_wgrad_kernel<2, 8>), grid, block, sm, st, p);
  }
  if (e != hipSuccess) {
    odvae_set_error("conv3x3_wgrad: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
    return ODVAE_ERR_HIP;
  }
  ODVAE_LAUNCH_CHECK("conv3x3_wgrad");
  const int64_t total = (int64_t)9 * pl.CinP * pl.CoutP + pl.CoutP;
  const int blocks = (int)std::min<int64_t>(ceil_div64(total, 256), 4096);
  hipLaunchKernelGGL(conv3x3_wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, st,
                     p.slab, p.bslab, pl.nsplit, Cin, Cout, pl.CinP, pl.CoutP, dw, dbias);
  ODVAE_LAUNCH_CHECK("conv3x3_wgrad reduce");
  return ODVAE_OK;
}

}  // extern "C"
