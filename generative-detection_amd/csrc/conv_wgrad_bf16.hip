// Weight gradient of the 3x3 / 1x1 convolutions of the bf16 path: x, dy bf16 NHWC in HBM, dW f32 OIHW (master-weight
// precision), fp32 accumulation on v_mfma_f32_32x32x16_bf16, gfx950.  Modes as conv_bf16.hip (0 stride 1, 1 Downsample,
// 2 Upsample, 4 = 1x1); autograd's weight gradient of the F.conv2d calls of [UPSTREAM] ldm/modules/diffusionmodules/model.py
// (src/modules/autoencodermodules/feat_encoder.py:4, feat_decoder.py:4).
//
//   dW[tap][co][ci] = sum over pixels  dy[px][co] * x[src(px, tap)][ci]        (K = pixels)
// Both operands are needed K-major per lane (8 consecutive PIXELS of one channel) while HBM holds them channel-major.  The tiles
// are staged row-major ([px][channels], coalesced 16-byte rows) and read back with ds_read_b64_tr_b16, the hardware transpose
// read: per 16-lane group a 4-pixel x 16-channel block arrives with the pixel index in the register -- no transposing pass.
//   A = dy^T (rows = co), B = x (cols = ci), D[co][ci] per tap: 9 accumulator tiles per wave (144 registers).
// Block = COT x CIT waves (one 32-co x 32-ci tile each, all taps), looping over its share of the 8x16 (4x16 for Downsample)
// output-pixel tiles: next tile's dy rows and x halo are fetched into registers while the current one is multiplied.
// Partial sums go to f32 slabs [split][tap][co][ci]; a second kernel adds the slabs in fixed order (deterministic) into OIHW.
#include "bf16_common.h"

namespace {

constexpr int TW = 16;

struct WgradB {
  const bf16_t* x;    // [N][Hi][Wi][Cin]
  const bf16_t* dy;   // [N][Ho][Wo][Cout]
  float* slab;        // [splits][taps][CoutP][CinP]
  float* bslab;       // [splits][CoutP] per-split column sums of dy (bias gradient), or null
  int N, Hi, Wi, Cin, Ho, Wo, Cout, CinP, CoutP;
  int tiles_x, tiles_y, ntiles, splits;
};

template <int MODE, int TH> struct HaloW;
template <int TH> struct HaloW<0, TH> { static constexpr int H = TH + 2, W = TW + 2, TAPS = 9; };
template <int TH> struct HaloW<1, TH> { static constexpr int H = 2 * TH + 1, W = 2 * TW + 1, TAPS = 9; };
template <int TH> struct HaloW<2, TH> { static constexpr int H = TH / 2 + 2, W = TW / 2 + 2, TAPS = 9; };
template <int TH> struct HaloW<4, TH> { static constexpr int H = TH, W = TW, TAPS = 1; };

template <int MODE, int TH>
__device__ __forceinline__ int halo_index_w(int r, int c, int kh, int kw) {
  if (MODE == 0) return (r + kh) * HaloW<0, TH>::W + (c + kw);
  if (MODE == 1) return (2 * r + kh) * HaloW<1, TH>::W + (2 * c + kw);
  if (MODE == 2) return ((r + kh + 1) >> 1) * HaloW<2, TH>::W + ((c + kw + 1) >> 1);
  return r * TW + c;
}

template <int MODE, int TH, int COT, int CIT>
__global__ __launch_bounds__(COT * CIT * 64) void conv_wgrad_bf16_kernel(WgradB p) {
  constexpr int NT = COT * CIT * 64;
  constexpr int BCO = COT * 32, BCI = CIT * 32;
  constexpr int DS = BCO + 32, XS = BCI + 32;          // LDS row strides (bf16): + 64 bytes keeps the transposed reads conflict-free
  constexpr int HPIX = HaloW<MODE, TH>::H * HaloW<MODE, TH>::W;
  constexpr int TAPS = HaloW<MODE, TH>::TAPS;
  constexpr int TPX = TH * TW;
  constexpr int DV = TPX * (BCO / 8), XV = HPIX * (BCI / 8);    // 16-byte vectors per stage
  constexpr int D_IT = (DV + NT - 1) / NT, X_IT = (XV + NT - 1) / NT;
  extern __shared__ __attribute__((aligned(16))) bf16_t smem[];   // (TPX * DS + HPIX * XS) bf16: up to 77 KB, dynamic
  bf16_t* Ds = smem;
  bf16_t* Xs = smem + TPX * DS;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cot = wave % COT, cit = wave / COT;
  const int co0 = blockIdx.y * BCO, ci0 = blockIdx.z * BCI;
  const unsigned OOB = 0x7FFFFFF0u;

  f32x16 acc[TAPS];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

  // bias gradient rides along: the waves of the first ci tile already hold dy^T fragments; one more MFMA against a tile of ones
  // sums them over the pixels (a VALU sum of the fragment cost 16 instructions per k-step and 20 % of the kernel)
  const bool do_bias = p.bslab != nullptr && cit == 0 && blockIdx.z == 0;
  f32x16 bacc;
#pragma unroll
  for (int i = 0; i < 16; ++i) bacc[i] = 0.f;
  const bf16x8 ones = frag_from_u32x4(u32x4{0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u});   // eight bf16 1.0

  // Per-thread constants of the staging loads (which tile pixel / halo pixel and channel octet each of its vectors is): the
  // per-tile work is then a few compares and one add per 16-byte load instead of 64-bit index arithmetic (the first version
  // spent ~30 VALU instructions per load, 210 per tile against 72 MFMAs).
  int d_r[D_IT], d_c[D_IT], d_rel[D_IT], x_r[X_IT], x_c[X_IT], x_rel[X_IT];
#pragma unroll
  for (int i = 0; i < D_IT; ++i) {
    const int f = tid + NT * i, px = f / (BCO / 8), c = co0 + 8 * (f % (BCO / 8));
    d_r[i] = px / TW; d_c[i] = px % TW;
    d_rel[i] = (f < DV && c < p.Cout) ? ((d_r[i] * p.Wo + d_c[i]) * p.Cout + c) : -1;
  }
#pragma unroll
  for (int i = 0; i < X_IT; ++i) {
    const int f = tid + NT * i, hp = f / (BCI / 8), c = ci0 + 8 * (f % (BCI / 8));
    x_r[i] = hp / HaloW<MODE, TH>::W; x_c[i] = hp % HaloW<MODE, TH>::W;
    x_rel[i] = (f < XV && c < p.Cin) ? ((x_r[i] * p.Wi + x_c[i]) * p.Cin + c) : -1;
  }
  u32x4 dreg[D_IT], xreg[X_IT];
  auto fetch = [&](int tile) {
    int t = tile;
    const int tx = t % p.tiles_x; t /= p.tiles_x;
    const int ty = t % p.tiles_y; const int n = t / p.tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TW;
    int iy0, ix0;
    if (MODE == 0) { iy0 = oy0 - 1; ix0 = ox0 - 1; }
    else if (MODE == 1) { iy0 = 2 * oy0; ix0 = 2 * ox0; }
    else if (MODE == 4) { iy0 = oy0; ix0 = ox0; }
    else { iy0 = oy0 / 2 - 1; ix0 = ox0 / 2 - 1; }
    const __amdgpu_buffer_rsrc_t drsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16_t*>(p.dy + (int64_t)n * p.Ho * p.Wo * p.Cout), 0, p.Ho * p.Wo * p.Cout * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16_t*>(p.x + (int64_t)n * p.Hi * p.Wi * p.Cin), 0, p.Hi * p.Wi * p.Cin * 2, 0x00020000);
    const int dbase = (oy0 * p.Wo + ox0) * p.Cout, xbase = (iy0 * p.Wi + ix0) * p.Cin;
    const int rows_left = p.Ho - oy0, cols_left = p.Wo - ox0;
#pragma unroll
    for (int i = 0; i < D_IT; ++i) {
      const bool ok = d_rel[i] >= 0 && d_r[i] < rows_left && d_c[i] < cols_left;
      dreg[i] = __builtin_amdgcn_raw_buffer_load_b128(drsrc, ok ? (unsigned)((dbase + d_rel[i]) * 2) : OOB, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < X_IT; ++i) {
      const bool ok = x_rel[i] >= 0 && (unsigned)(iy0 + x_r[i]) < (unsigned)p.Hi && (unsigned)(ix0 + x_c[i]) < (unsigned)p.Wi;
      xreg[i] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, ok ? (unsigned)((xbase + x_rel[i]) * 2) : OOB, 0, 0);
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int i = 0; i < D_IT; ++i) {
      const int f = tid + NT * i;
      if (f < DV) *reinterpret_cast<u32x4*>(Ds + (f / (BCO / 8)) * DS + 8 * (f % (BCO / 8))) = dreg[i];
    }
#pragma unroll
    for (int i = 0; i < X_IT; ++i) {
      const int f = tid + NT * i;
      if (f < XV) *reinterpret_cast<u32x4*>(Xs + (f / (BCI / 8)) * XS + 8 * (f % (BCI / 8))) = xreg[i];
    }
  };

  // transposed-read roles of this lane: group g = lane >> 4 serves MFMA lanes (row/col 16(g&1) + i, k half h = g >> 1);
  // lane 4q + p of the group supplies the address of pixel (8h + 4*half + q), channels 4p .. 4p+3
  const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
  const int hk = g >> 1;
  const int dcol = cot * 32 + 16 * (g & 1) + 4 * pp;    // dy channel column this lane addresses
  const int xcol = cit * 32 + 16 * (g & 1) + 4 * pp;

  int tile = blockIdx.x;
  if (tile < p.ntiles) fetch(tile);
  for (; tile < p.ntiles; tile += p.splits) {
    stage();
    __syncthreads();
    if (tile + p.splits < p.ntiles) fetch(tile + p.splits);
#pragma unroll
    for (int r = 0; r < TH; ++r) {
      const int c_lo = 8 * hk + q, c_hi = c_lo + 4;     // tile column (= pixel within the 16-pixel k-step) of the two reads
      const bf16x8 a = frag_from_tr(lds_read_tr16(Ds + (r * TW + c_lo) * DS + dcol), lds_read_tr16(Ds + (r * TW + c_hi) * DS + dcol));
      if (do_bias) bacc = mfma_bf16(a, ones, bacc);   // wave-uniform; every column of bacc = sum over the 16 pixels of dy^T rows
#pragma unroll
      for (int t = 0; t < TAPS; ++t) {
        const int kh = TAPS == 1 ? 0 : t / 3, kw = TAPS == 1 ? 0 : t % 3;
        const bf16x8 b = frag_from_tr(lds_read_tr16(Xs + halo_index_w<MODE, TH>(r, c_lo, kh, kw) * XS + xcol),
                                      lds_read_tr16(Xs + halo_index_w<MODE, TH>(r, c_hi, kh, kw) * XS + xcol));
        acc[t] = mfma_bf16(a, b, acc[t]);
      }
    }
    __syncthreads();
  }

  // slab [split][tap][CoutP][CinP]: register i of lane (r, h) is dW[co = 32 cot + (i&3) + 8(i>>2) + 4h][ci = 32 cit + r]
  float* slab = p.slab + (int64_t)blockIdx.x * TAPS * p.CoutP * p.CinP;
  const int li = lane & 31, h = lane >> 5;
  if (do_bias && li == 0) {   // column 0 of the (all columns equal) sum tile: rows = channels
#pragma unroll
    for (int i = 0; i < 16; ++i) p.bslab[(int64_t)blockIdx.x * p.CoutP + co0 + cot * 32 + (i & 3) + 8 * (i >> 2) + 4 * h] = bacc[i];
  }
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int co = co0 + cot * 32 + (i & 3) + 8 * (i >> 2) + 4 * h, ci = ci0 + cit * 32 + li;
      slab[((int64_t)t * p.CoutP + co) * p.CinP + ci] = acc[t][i];
    }
}

// dw[co][ci][tap] (OIHW) = sum over splits of slab[s][tap][co][ci]
__global__ void wgrad_bf16_reduce_kernel(const float* __restrict__ slab, int splits, int taps, int Cout, int Cin, int CoutP, int CinP,
                                         float* __restrict__ dw, const float* __restrict__ bslab, float* __restrict__ db) {
  const int64_t per = (int64_t)taps * CoutP * CinP;
  if (db && blockIdx.x == gridDim.x - 1)      // (the LAST block: it has the smallest share of the grid-stride loop below, if any)
    for (int co = threadIdx.x; co < Cout; co += blockDim.x) {
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
      int k = 0;
      for (; k + 3 < splits; k += 4) {
        s0 += bslab[(int64_t)k * CoutP + co]; s1 += bslab[(int64_t)(k + 1) * CoutP + co];
        s2 += bslab[(int64_t)(k + 2) * CoutP + co]; s3 += bslab[(int64_t)(k + 3) * CoutP + co];
      }
      for (; k < splits; ++k) s0 += bslab[(int64_t)k * CoutP + co];
      db[co] = (s0 + s1) + (s2 + s3);
    }
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < per; idx += (int64_t)gridDim.x * blockDim.x) {
    const int ci = (int)(idx % CinP);
    const int co = (int)((idx / CinP) % CoutP);
    const int tap = (int)(idx / ((int64_t)CinP * CoutP));
    if (ci >= Cin || co >= Cout) continue;
    // eight independent partial sums: the loads of a thread are otherwise one dependent chain of `splits` L2 / HBM round trips (four in
    // flight per thread left the 75 MB of slabs at 3 TB/s)
    float sk[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int k = 0;
    for (; k + 7 < splits; k += 8) {
#pragma unroll
      for (int j = 0; j < 8; ++j) sk[j] += slab[(int64_t)(k + j) * per + idx];
    }
    for (; k < splits; ++k) sk[0] += slab[(int64_t)k * per + idx];
    dw[((int64_t)co * Cin + ci) * taps + tap] = ((sk[0] + sk[1]) + (sk[2] + sk[3])) + ((sk[4] + sk[5]) + (sk[6] + sk[7]));
  }
}

int pad_to(int v, int m) { return (v + m - 1) / m * m; }

template <int MODE, int TH, int COT, int CIT>
void launch_wgrad(const WgradB& p, dim3 grid, hipStream_t st) {
  constexpr int bytes = (TH * TW * (COT * 32 + 32) + HaloW<MODE, TH>::H * HaloW<MODE, TH>::W * (CIT * 32 + 32)) * 2;
  static bool once = false;   // more than 64 KB of LDS has to be requested per kernel
  if (!once) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_bf16_kernel<MODE, TH, COT, CIT>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    once = true;
  }
  hipLaunchKernelGGL((conv_wgrad_bf16_kernel<MODE, TH, COT, CIT>), grid, dim3(COT * CIT * 64), bytes, st, p);
}

struct Plan { int CoutP, CinP, splits, ntiles, tiles_x, tiles_y, th, cot, cit; };

bool make_plan(int mode, int N, int Ho, int Wo, int Cin, int Cout, Plan& pl) {
  if (!(mode == 0 || mode == 1 || mode == 2 || mode == 4)) return false;
  pl.th = mode == 1 ? 4 : 8;
  pl.cot = Cout > 32 ? 4 : 1;
  pl.cit = Cout > 32 ? 2 : 4;
  pl.CoutP = pad_to(Cout, pl.cot * 32);
  pl.CinP = pad_to(Cin, pl.cit * 32);
  pl.tiles_x = ceil_div(Wo, TW); pl.tiles_y = ceil_div(Ho, pl.th);
  const int64_t nt = (int64_t)N * pl.tiles_x * pl.tiles_y;
  if (nt >= (1ll << 31)) return false;
  pl.ntiles = (int)nt;
  const int pairs = (pl.CoutP / (pl.cot * 32)) * (pl.CinP / (pl.cit * 32));
  int s = std::max(1, 256 / pairs);       // one 8-wave block per CU in a single round; fewer slabs for the reduce to read
  pl.splits = (int)std::min<int64_t>(s, nt);
  return true;
}

}  // namespace

extern "C" {

size_t odvae_conv_wgrad_bf16_workspace_bytes(int mode, int N, int Ho, int Wo, int Cin, int Cout) {
  Plan pl;
  if (!make_plan(mode, N, Ho, Wo, Cin, Cout, pl)) return 0;
  return ((size_t)pl.splits * (mode == 4 ? 1 : 9) * pl.CoutP * pl.CinP + (size_t)pl.splits * pl.CoutP) * sizeof(float);
}

// dw f32 OIHW [Cout][Cin][k][k] (k*k = 9, or 1 for mode 4) from x bf16 [N][Hi][Wi][Cin] and dy bf16 [N][Ho][Wo][Cout];
// Cin % 8 == 0 and Cout % 8 == 0 (16-byte channel vectors).  db f32 [Cout] = per-channel sum of dy (bias gradient) or NULL: it rides
// along in the same pass over dy.  Deterministic (fixed slab order).
int odvae_conv_wgrad_bf16(int mode, const void* x, const void* dy, int N, int Hi, int Wi, int Cin, int Ho, int Wo, int Cout,
                          float* dw, float* db, void* workspace, size_t workspace_bytes, void* stream) {
  Plan pl;
  ODVAE_CHECK_ARG(x && dy && dw && N > 0 && Cin > 0 && Cout > 0, "conv_wgrad_bf16: null or empty operand");
  ODVAE_CHECK_ARG(make_plan(mode, N, Ho, Wo, Cin, Cout, pl), "conv_wgrad_bf16: unsupported mode %d / shape", mode);
  ODVAE_CHECK_ARG(Cin % 8 == 0 && Cout % 8 == 0, "conv_wgrad_bf16: Cin = %d and Cout = %d must be multiples of 8", Cin, Cout);
  int eh, ew;
  if (mode == 0 || mode == 4) { eh = Hi; ew = Wi; }
  else if (mode == 1) { eh = (Hi - 2) / 2 + 1; ew = (Wi - 2) / 2 + 1; }
  else { eh = 2 * Hi; ew = 2 * Wi; }
  ODVAE_CHECK_ARG(Ho == eh && Wo == ew, "conv_wgrad_bf16(mode %d): dy %dx%d does not match x %dx%d", mode, Ho, Wo, Hi, Wi);
  ODVAE_CHECK_ARG((int64_t)Hi * Wi * Cin * 2 < 0x7FFFFFF0ll && (int64_t)Ho * Wo * Cout * 2 < 0x7FFFFFF0ll,
                  "conv_wgrad_bf16: one image exceeds the 2 GiB buffer-descriptor range");
  ODVAE_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)dy & 15) == 0, "conv_wgrad_bf16: misaligned operand");
  const size_t need = odvae_conv_wgrad_bf16_workspace_bytes(mode, N, Ho, Wo, Cin, Cout);
  if (!workspace || workspace_bytes < need) {
    odvae_set_error("conv_wgrad_bf16: needs %zu workspace bytes, got %zu", need, workspace_bytes);
    return ODVAE_ERR_WORKSPACE;
  }
  WgradB p;
  p.x = static_cast<const bf16_t*>(x); p.dy = static_cast<const bf16_t*>(dy); p.slab = static_cast<float*>(workspace);
  p.bslab = db ? p.slab + (size_t)pl.splits * (mode == 4 ? 1 : 9) * pl.CoutP * pl.CinP : nullptr;
  p.N = N; p.Hi = Hi; p.Wi = Wi; p.Cin = Cin; p.Ho = Ho; p.Wo = Wo; p.Cout = Cout; p.CinP = pl.CinP; p.CoutP = pl.CoutP;
  p.tiles_x = pl.tiles_x; p.tiles_y = pl.tiles_y; p.ntiles = pl.ntiles; p.splits = pl.splits;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid(pl.splits, pl.CoutP / (pl.cot * 32), pl.CinP / (pl.cit * 32));
#define ODVAE_WG(MODE, TH)                                                                                                 \
  if (pl.cot == 4) launch_wgrad<MODE, TH, 4, 2>(p, grid, st);                                                             \
  else launch_wgrad<MODE, TH, 1, 4>(p, grid, st)
  switch (mode) {
    case 0: ODVAE_WG(0, 8); break;
    case 1: ODVAE_WG(1, 4); break;
    case 2: ODVAE_WG(2, 8); break;
    default: ODVAE_WG(4, 8); break;
  }
#undef ODVAE_WG
  ODVAE_LAUNCH_CHECK("conv_wgrad_bf16");
  const int taps = mode == 4 ? 1 : 9;
  const int64_t per = (int64_t)taps * pl.CoutP * pl.CinP;
  hipLaunchKernelGGL(wgrad_bf16_reduce_kernel, dim3((unsigned)std::min<int64_t>(ceil_div64(per, 256), 4096)), dim3(256), 0, st,
                     p.slab, pl.splits, taps, Cout, Cin, pl.CoutP, pl.CinP, dw, p.bslab, db);
  ODVAE_LAUNCH_CHECK("conv_wgrad_bf16 reduce");
  return ODVAE_OK;
}

}  // extern "C"
