// 3x3 convolution, NHWC, exact f32 on the matrix cores (v_mfma_f32_32x32x2_f32), gfx950.
//
// Replaces the F.conv2d calls of [UPSTREAM] ldm/modules/diffusionmodules/model.py:
//   ResnetBlock.conv1/conv2, Encoder/Decoder conv_in/conv_out      -> MODE 0 (stride 1, pad 1)
//   Downsample (F.pad(x,(0,1,0,1)) + conv stride 2 pad 0)            -> MODE 1
//   Upsample (F.interpolate(scale 2, nearest) + conv stride 1 pad 1) -> MODE 2 (no 4x intermediate)
//   data-gradient of MODE 1 (stride-2 transposed conv)               -> MODE 3
//   Upsample + conv by output parity class (four 2x2-tap convs on the low-res input, pre-summed weights:
//   16 instead of 36 tap-products per input pixel)                   -> MODE 5;  its data gradient, a 4x4-tap
//   stride-2 pad-1 conv over dy with pre-summed weights             -> MODE 6   (v2 kernel only)
// reached from src/modules/autoencodermodules/feat_encoder.py:4, feat_decoder.py:4 of the reference.
// The data-gradient of MODE 0 is MODE 0 itself on weights packed with flipped taps and swapped
// channel roles (odvae_conv3x3_pack_f32 makes both packs).
//
// Implicit GEMM: M = output pixels (block tile = 8x16 patch = 128 px), N = Cout, K = 9 taps x Cin.
// Per chunk of KC input channels the block stages the input halo patch once ([halo px][KC] in LDS,
// reused by all 9 taps) and streams the 9 per-tap weight slices [KC/4][BN][4] through a 2-deep LDS
// ring (loads for tap t+1 are issued into registers before tap t's MFMAs, written after them).
// A and B fragments are ds_read_b128: four k-steps per read, k order (8g + 4*half + j) on both.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int TH = 8, TW = 16;  // output pixels per block tile (TH*TW = 128)

struct ConvParams {
  const float* x;         // [N][Hi][Wi][Cin]
  const float* wpk;       // [9][CinP/4][CoutP][4]
  const float* bias;      // [Cout] or null
  const float* residual;  // [N][Ho][Wo][Cout] or null
  float* y;               // [N][Ho][Wo][Cout]
  int N, Hi, Wi, Cin, Ho, Wo, Cout, CinP, CoutP;
  int tiles_x, tiles_y;
  int act;                // 0 none, 1 ReLU (VGG slices of the LPIPS-style net)
};

template <int MODE> struct Halo;
template <> struct Halo<0> { static constexpr int H = TH + 2, W = TW + 2; };
template <> struct Halo<1> { static constexpr int H = 2 * TH + 1, W = 2 * TW + 1; };
template <> struct Halo<2> { static constexpr int H = TH / 2 + 2, W = TW / 2 + 2; };
template <> struct Halo<3> { static constexpr int H = TH / 2 + 1, W = TW / 2 + 1; };
template <> struct Halo<5> { static constexpr int H = TH + 2, W = TW + 2; };          // tile = low-res pixels
template <> struct Halo<6> { static constexpr int H = 2 * TH + 2, W = 2 * TW + 2; };  // halo over dy at 2x resolution

// halo pixel index for output pixel (r,c) of the tile and tap (kh,kw); ok=false -> operand is zero
template <int MODE>
__device__ __forceinline__ int halo_index(int r, int c, int kh, int kw, bool& ok) {
  ok = true;
  if (MODE == 0) return (r + kh) * Halo<0>::W + (c + kw);
  if (MODE == 1) return (2 * r + kh) * Halo<1>::W + (2 * c + kw);
  if (MODE == 2) return ((r + kh + 1) >> 1) * Halo<2>::W + ((c + kw + 1) >> 1);
  ok = (((r + kh) | (c + kw)) & 1) == 0;
  return ((r + kh) >> 1) * Halo<3>::W + ((c + kw) >> 1);
}

// top-left input pixel of the halo patch of the tile whose first output pixel is (oy0, ox0)
template <int MODE>
__device__ __forceinline__ void halo_origin(int oy0, int ox0, int& iy0, int& ix0) {
  if (MODE == 0) { iy0 = oy0 - 1; ix0 = ox0 - 1; }
  else if (MODE == 1) { iy0 = 2 * oy0; ix0 = 2 * ox0; }
  else if (MODE == 5) { iy0 = oy0 - 1; ix0 = ox0 - 1; }
  else if (MODE == 6) { iy0 = 2 * oy0 - 1; ix0 = 2 * ox0 - 1; }
  else { iy0 = oy0 / 2 - 1; ix0 = ox0 / 2 - 1; }
}

// WMT x WNT MFMA tiles per wave, WAVES_M x WAVES_N waves (4 waves; WAVES_M*WMT = 4 M-tiles of 32 px)
template <int MODE, int KC, int WMT, int WNT, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(256) void conv3x3_kernel(ConvParams p) {
  static_assert(WAVES_M * WAVES_N == 4 && WAVES_M * WMT == 4, "tile layout");
  constexpr int BN = WAVES_N * WNT * 32;
  constexpr int HS = KC + 4;                          // halo row stride (floats)
  constexpr int HPIX = Halo<MODE>::H * Halo<MODE>::W;
  constexpr int QC = KC / 4;                          // channel quads per chunk
  constexpr int HALO_F4 = HPIX * QC;                  // float4 per halo stage
  constexpr int HALO_IT = (HALO_F4 + 255) / 256;
  constexpr int W_F4 = QC * BN;                       // float4 per tap slice
  constexpr int W_IT = (W_F4 + 255) / 256;
  __shared__ __attribute__((aligned(16))) float smem[HPIX * HS + 2 * W_F4 * 4];
  float* Hs = smem;
  float* Ws = smem + HPIX * HS;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int li = lane & 31, h = lane >> 5;

  int t = blockIdx.x;
  const int tx = t % p.tiles_x; t /= p.tiles_x;
  const int ty = t % p.tiles_y; const int n = t / p.tiles_y;
  const int oy0 = ty * TH, ox0 = tx * TW;
  const int n0 = blockIdx.y * BN;
  int iy0, ix0;
  halo_origin<MODE>(oy0, ox0, iy0, ix0);
  const float* xn = p.x + (int64_t)n * p.Hi * p.Wi * p.Cin;
  const bool vec = (p.Cin & 3) == 0;

  f32x16 acc[WMT][WNT];
#pragma unroll
  for (int a = 0; a < WMT; ++a)
#pragma unroll
    for (int b = 0; b < WNT; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  float4 hreg[HALO_IT], wreg[W_IT];

  auto load_halo = [&](int c0) {
#pragma unroll
    for (int i = 0; i < HALO_IT; ++i) {
      const int f = tid + 256 * i;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (f < HALO_F4) {
        const int hp = f / QC, q = f % QC;
        const int iy = iy0 + hp / Halo<MODE>::W, ix = ix0 + hp % Halo<MODE>::W;
        const int c = c0 + 4 * q;
        if (iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi && c < p.Cin) {
          const float* src = xn + ((int64_t)iy * p.Wi + ix) * p.Cin + c;
          if (vec) v = *reinterpret_cast<const float4*>(src);
          else {
            v.x = src[0];
            if (c + 1 < p.Cin) v.y = src[1];
            if (c + 2 < p.Cin) v.z = src[2];
            if (c + 3 < p.Cin) v.w = src[3];
          }
        }
      }
      hreg[i] = v;
    }
  };
  auto store_halo = [&]() {
#pragma unroll
    for (int i = 0; i < HALO_IT; ++i) {
      const int f = tid + 256 * i;
      if (f < HALO_F4) *reinterpret_cast<float4*>(Hs + (f / QC) * HS + 4 * (f % QC)) = hreg[i];
    }
  };
  auto load_w = [&](int c0, int tap) {
    const float* src = p.wpk + ((int64_t)tap * (p.CinP / 4) + c0 / 4) * p.CoutP * 4;
#pragma unroll
    for (int i = 0; i < W_IT; ++i) {
      const int f = tid + 256 * i;
      if (f < W_F4) {
        const int q = f / BN, nn = f % BN;
        wreg[i] = *reinterpret_cast<const float4*>(src + ((int64_t)q * p.CoutP + n0 + nn) * 4);
      }
    }
  };
  auto store_w = [&](int buf) {
#pragma unroll
    for (int i = 0; i < W_IT; ++i) {
      const int f = tid + 256 * i;
      if (f < W_F4) *reinterpret_cast<float4*>(Ws + (buf * W_F4 + f) * 4) = wreg[i];
    }
  };

  const int nchunks = p.CinP / KC;
  load_halo(0);
  load_w(0, 0);
  for (int ch = 0; ch < nchunks; ++ch) {
    store_halo();
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {
      store_w(tap & 1);
      __syncthreads();
      // prefetch the next weight slice (and, near the end of the chunk, the next halo)
      if (tap < 8) load_w(ch * KC, tap + 1);
      else if (ch + 1 < nchunks) load_w((ch + 1) * KC, 0);
      if (tap == 7 && ch + 1 < nchunks) load_halo((ch + 1) * KC);

      const int kh = tap / 3, kw = tap % 3;
      int aoff[WMT]; bool aok[WMT];
#pragma unroll
      for (int mt = 0; mt < WMT; ++mt) {
        const int pm = (wm * WMT + mt) * 32 + li;
        aoff[mt] = halo_index<MODE>(pm / TW, pm % TW, kh, kw, aok[mt]) * HS + 4 * h;
      }
      const float* Wb = Ws + (tap & 1) * W_F4 * 4;
#pragma unroll
      for (int g = 0; g < KC / 8; ++g) {
        float4 a[WMT], b[WNT];
#pragma unroll
        for (int mt = 0; mt < WMT; ++mt) {
          a[mt] = *reinterpret_cast<const float4*>(Hs + aoff[mt] + g * 8);
          if (MODE == 3 && !aok[mt]) a[mt] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int nt = 0; nt < WNT; ++nt)
          b[nt] = *reinterpret_cast<const float4*>(Wb + ((g * 2 + h) * BN + (wn * WNT + nt) * 32 + li) * 4);
#pragma unroll
        for (int mt = 0; mt < WMT; ++mt)
#pragma unroll
          for (int nt = 0; nt < WNT; ++nt) {
            acc[mt][nt] = mfma32(a[mt].x, b[nt].x, acc[mt][nt]);
            acc[mt][nt] = mfma32(a[mt].y, b[nt].y, acc[mt][nt]);
            acc[mt][nt] = mfma32(a[mt].z, b[nt].z, acc[mt][nt]);
            acc[mt][nt] = mfma32(a[mt].w, b[nt].w, acc[mt][nt]);
          }
      }
    }
    __syncthreads();  // every wave is done with this chunk's halo before it is overwritten
  }

  // epilogue: lane = output channel, registers = pixels
  float* yn = p.y + (int64_t)n * p.Ho * p.Wo * p.Cout;
  const float* rn = p.residual ? p.residual + (int64_t)n * p.Ho * p.Wo * p.Cout : nullptr;
#pragma unroll
  for (int nt = 0; nt < WNT; ++nt) {
    const int co = n0 + (wn * WNT + nt) * 32 + li;
    if (co >= p.Cout) continue;
    const float bv = p.bias ? p.bias[co] : 0.f;
#pragma unroll
    for (int mt = 0; mt < WMT; ++mt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int pm = (wm * WMT + mt) * 32 + acc_row(r, lane);
        const int oy = oy0 + pm / TW, ox = ox0 + pm % TW;
        if (oy < p.Ho && ox < p.Wo) {
          const int64_t o = ((int64_t)oy * p.Wo + ox) * p.Cout + co;
          float v = acc[mt][nt][r] + bv;
          if (rn) v += rn[o];
          if (p.act) v = fmaxf(v, 0.f);
          yn[o] = v;
        }
      }
    }
  }
}

#ifdef ODVAE_STAMPS
// Diagnostic build only (tools/conv_stamps.py): s_memtime stamps of the v2 kernel's phases, per wave, for the
// first 4096 blocks.  Never compiled into libodvae_hip.so.
__device__ unsigned long long g_stamps[4096 * 4 * 4];
#define ODVAE_T() __builtin_amdgcn_s_memtime()
#endif

// ---- v2 main loop ------------------------------------------------------------------------------------
// Same tiling and epilogue, different feeding: weight fragments go global/L2 -> registers directly (the pack
// layout makes one coalesced 16-byte load per lane a whole B fragment for four k-steps; prefetched one k-group
// = 16 MFMAs ahead), so LDS holds only the input halo, double-buffered per KC-channel chunk.  One
// __syncthreads() per chunk (9*KC/8*16 = 576 MFMAs per wave at KC = 32) instead of one per tap (32 MFMAs):
// rocprof PMC on v1 showed 39 % of wave cycles parked at the per-tap barrier / waitcnt with 3 blocks per CU.
template <int MODE, int KC, int WMT, int WNT, int WAVES_M, int WAVES_N, int MINW = 1>
__global__ __launch_bounds__(256, MINW) void conv3x3_kernel_v2(ConvParams p) {
  static_assert(WAVES_M * WAVES_N == 4 && WAVES_M * WMT == 4, "tile layout");
  constexpr int BN = WAVES_N * WNT * 32;
  constexpr int HS = KC + 4;
  constexpr int HPIX = Halo<MODE>::H * Halo<MODE>::W;
  constexpr int QC = KC / 4;
  constexpr int NG = KC / 8;
  constexpr int HALO_F4 = HPIX * QC;
  constexpr int HALO_IT = (HALO_F4 + 255) / 256;
  constexpr int KW = MODE == 5 ? 2 : (MODE == 6 ? 4 : 3);   // taps per row; TAPS = KW * KW
  constexpr int TAPS = KW * KW;
  __shared__ __attribute__((aligned(16))) float smem[2 * HPIX * HS];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int li = lane & 31, h = lane >> 5;
  // MODE 5: blockIdx.z = output parity class (py, px); the tile is 8x16 LOW-RES pixels and pixel (y, x) of it produces
  // output pixel (2y + py, 2x + px) from input pixels (y + py - 1 + a, x + px - 1 + b), a, b in {0, 1}
  const int cls = MODE == 5 ? (int)blockIdx.z : 0, py = cls >> 1, px = cls & 1;

  int t = blockIdx.x;
  const int tx = t % p.tiles_x; t /= p.tiles_x;
  const int ty = t % p.tiles_y; const int n = t / p.tiles_y;
  const int oy0 = ty * TH, ox0 = tx * TW;
  const int n0 = blockIdx.y * BN;
  int iy0, ix0;
  halo_origin<MODE>(oy0, ox0, iy0, ix0);
  const float* xn = p.x + (int64_t)n * p.Hi * p.Wi * p.Cin;
  const bool vec = (p.Cin & 3) == 0;

  // Output addressing: 32-bit byte offsets into per-image buffer descriptors; a lane whose pixel or channel is
  // outside gets an offset past num_records, which the hardware turns into "load 0" / "drop the store": no compares,
  // no branches, no 64-bit math per element (the int64/branchy form cost ~25 VALU per store, 8k cycles per block).
  const int img_bytes = p.Ho * p.Wo * p.Cout * 4;
  const unsigned OOB = 0x7FFFFFF0u;
  unsigned cobyte[WNT];
#pragma unroll
  for (int nt = 0; nt < WNT; ++nt) {
    const int co = n0 + (wn * WNT + nt) * 32 + li;
    cobyte[nt] = co < p.Cout ? (unsigned)co * 4u : OOB;
  }
  // Tile pixel of row i of M-tile `mtg` (0..3).  MODE 3 (transposed stride-2 conv): an M-tile holds the 32 pixels of ONE
  // parity class (py, px) of the 8x16 patch, so whether tap (kh, kw) reads a real dy pixel or an inserted zero is the
  // same for the whole tile and the all-zero tap-tiles are skipped (9 of 36 remain).  Classes are paired 0:(0,0) 1:(1,1)
  // 2:(0,1) 3:(1,0) so that each wave of the 2x2 layout gets 5 resp. 4 live tap-tiles.
  auto tile_pixel = [&](int mtg, int i, int& r, int& c) {
    if (MODE == 3) { r = 2 * (i >> 3) + (mtg & 1); c = 2 * (i & 7) + (((mtg + 1) >> 1) & 1); }
    else { const int pm = mtg * 32 + i; r = pm / TW; c = pm % TW; }
  };
  auto tap_live = [&](int mtg, int kh, int kw) -> bool {
    return MODE != 3 || ((((mtg & 1) + kh) | ((((mtg + 1) >> 1) & 1) + kw)) & 1) == 0;
  };
  auto pix_byte = [&](int mt, int r) -> unsigned {
    int pr, pc;
    tile_pixel(wm * WMT + mt, acc_row(r, lane), pr, pc);
    if (MODE == 5) {
      const int ly = oy0 + pr, lx = ox0 + pc;
      return (ly < p.Hi && lx < p.Wi) ? (unsigned)(((2 * ly + py) * p.Wo + 2 * lx + px) * p.Cout) * 4u : OOB;
    }
    const int oy = oy0 + pr, ox = ox0 + pc;
    return (oy < p.Ho && ox < p.Wo) ? (unsigned)((oy * p.Wo + ox) * p.Cout) * 4u : OOB;
  };

  // The accumulators START at bias + residual, so the epilogue is stores only: vmcnt counts loads and stores in
  // order, and a load between stores (or hipcc's vmcnt(0) at the join of a guarded load) makes every store wait out
  // the previous store's write latency -- s_memtime stamps showed 89k cycles of epilogue per 147k of MFMA that way.
  f32x16 acc[WMT][WNT];
  {
    float bv[WNT];
#pragma unroll
    for (int nt = 0; nt < WNT; ++nt) bv[nt] = p.bias ? p.bias[min(n0 + (wn * WNT + nt) * 32 + li, p.Cout - 1)] : 0.f;
    if (p.residual) {
      const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<float*>(p.residual) + (int64_t)n * p.Ho * p.Wo * p.Cout, 0, img_bytes, 0x00020000);
#pragma unroll
      for (int mt = 0; mt < WMT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const unsigned pb = pix_byte(mt, r);
#pragma unroll
          for (int nt = 0; nt < WNT; ++nt)
            acc[mt][nt][r] = bv[nt] + __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rrsrc, pb + cobyte[nt], 0, 0));
        }
    } else {
#pragma unroll
      for (int mt = 0; mt < WMT; ++mt)
#pragma unroll
        for (int nt = 0; nt < WNT; ++nt)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[mt][nt][r] = bv[nt];
    }
  }

  float4 hreg[HALO_IT];
  // Halo fetch.  Vector path: branch-free buffer loads -- a lane outside the image (or past Cin) points its
  // offset beyond num_records and the hardware returns 0, so all HALO_IT loads are in flight together.  (A guarded
  // `if (inside) v = load` makes hipcc branch around every load and wait vmcnt(0) at each join: stamps showed the
  // prologue at 34k cycles.)  Scalar path only for Cin % 4 != 0 (the RGB input).
  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(xn), 0, p.Hi * p.Wi * p.Cin * 4, 0x00020000);
  auto load_halo = [&](int c0) {
    if (vec) {
#pragma unroll
      for (int i = 0; i < HALO_IT; ++i) {
        const int f = tid + 256 * i;
        const int hp = f / QC, q = f % QC;
        const int iy = iy0 + hp / Halo<MODE>::W, ix = ix0 + hp % Halo<MODE>::W;
        const int c = c0 + 4 * q;
        const bool ok = f < HALO_F4 && iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi && c < p.Cin;
        const unsigned voff = ok ? (unsigned)(((iy * p.Wi + ix) * p.Cin + c) * 4) : 0x7FFFFFF0u;
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, voff, 0, 0);
        hreg[i] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
      }
    } else {
#pragma unroll
      for (int i = 0; i < HALO_IT; ++i) {
        const int f = tid + 256 * i;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (f < HALO_F4) {
          const int hp = f / QC, q = f % QC;
          const int iy = iy0 + hp / Halo<MODE>::W, ix = ix0 + hp % Halo<MODE>::W;
          const int c = c0 + 4 * q;
          if (iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi && c < p.Cin) {
            const float* src = xn + ((int64_t)iy * p.Wi + ix) * p.Cin + c;
            v.x = src[0];
            if (c + 1 < p.Cin) v.y = src[1];
            if (c + 2 < p.Cin) v.z = src[2];
            if (c + 3 < p.Cin) v.w = src[3];
          }
        }
        hreg[i] = v;
      }
    }
  };
  auto store_halo = [&](float* Hs) {
#pragma unroll
    for (int i = 0; i < HALO_IT; ++i) {
      const int f = tid + 256 * i;
      if (f < HALO_F4) *reinterpret_cast<float4*>(Hs + (f / QC) * HS + 4 * (f % QC)) = hreg[i];
    }
  };

  // weight fragments: float4 index ((tap * QT + quad) * CoutP + cout)
  const float4* wq = reinterpret_cast<const float4*>(p.wpk);
  const int QT = p.CinP / 4;
  int ncol[WNT];
#pragma unroll
  for (int nt = 0; nt < WNT; ++nt) ncol[nt] = n0 + (wn * WNT + nt) * 32 + li;
  auto load_b = [&](int ch, int tap, int g, float4 (&b)[WNT]) {
    const int64_t base = ((int64_t)(cls * TAPS + tap) * QT + ch * QC + 2 * g + h) * p.CoutP;
#pragma unroll
    for (int nt = 0; nt < WNT; ++nt) b[nt] = wq[base + ncol[nt]];
  };

  const int nchunks = p.CinP / KC;
  constexpr int NIT = TAPS * NG;   // (tap, k-group) steps per chunk, 16 MFMAs per wave each
  float4 bc[WNT], bn[WNT], ac[WMT], an[WMT];
  // A fragment of step `it` of the current chunk: halo pixel of (tile pixel, tap), channels 8g + 4h .. +3
  auto load_a = [&](const float* Hs, int it, float4 (&a)[WMT]) {
    const int tap = it / NG, g = it % NG;
#pragma unroll
    for (int mt = 0; mt < WMT; ++mt) {
      int pr, pc;
      tile_pixel(wm * WMT + mt, li, pr, pc);
      int hp;
      if (MODE == 5) hp = (pr + py + tap / KW) * Halo<5>::W + (pc + px + tap % KW);
      else if (MODE == 6) hp = (2 * pr + tap / KW) * Halo<6>::W + (2 * pc + tap % KW);
      else { bool ok; hp = halo_index<(MODE <= 3 ? MODE : 0)>(pr, pc, tap / 3, tap % 3, ok); }
      const int off = hp * HS + 4 * h + 8 * g;
      a[mt] = *reinterpret_cast<const float4*>(Hs + off);   // MODE 3: dead taps read a harmless in-range pixel
    }
  };
#ifdef ODVAE_STAMPS
  const unsigned long long st_start = ODVAE_T();
  unsigned long long st_bar = 0;
#endif
  // weight fragments run TWO steps ahead (L2 latency under load exceeds one 16-MFMA step); bn2 is the far slot
  float4 bn2[WNT];
  auto load_b_step = [&](int ch, int it, bool valid, float4 (&b)[WNT]) {   // step `it` may spill into chunk ch+1
    if (it < NIT) load_b(ch, it / NG, it % NG, b);
    else if (valid) load_b(ch + 1, (it - NIT) / NG, (it - NIT) % NG, b);
  };
  load_halo(0);
  load_b(0, 0, 0, bc);
  load_b(0, 1 / NG, 1 % NG, bn);
  store_halo(smem);
  __syncthreads();
#ifdef ODVAE_STAMPS
  const unsigned long long st_loop = ODVAE_T();
#endif

  for (int ch = 0; ch < nchunks; ++ch) {
    const float* Hs = smem + (ch & 1) * HPIX * HS;
    const bool more = ch + 1 < nchunks;
    if (more) load_halo((ch + 1) * KC);
    load_a(Hs, 0, ac);
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      // software pipeline, pinned: B of step it+2 and A of step it+1 are requested before step it's MFMAs issue
      load_b_step(ch, it + 2, more, bn2);
      if (it + 1 < NIT) load_a(Hs, it + 1, an);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mt = 0; mt < WMT; ++mt) {
        // MODE 3 only; readfirstlane makes the predicate provably wave-uniform (a scalar branch, never an EXEC mask
        // around MFMAs)
        if (!tap_live(__builtin_amdgcn_readfirstlane(wm) * WMT + mt, (it / NG) / 3, (it / NG) % 3)) continue;
#pragma unroll
        for (int nt = 0; nt < WNT; ++nt) {
          acc[mt][nt] = mfma32(ac[mt].x, bc[nt].x, acc[mt][nt]);
          acc[mt][nt] = mfma32(ac[mt].y, bc[nt].y, acc[mt][nt]);
          acc[mt][nt] = mfma32(ac[mt].z, bc[nt].z, acc[mt][nt]);
          acc[mt][nt] = mfma32(ac[mt].w, bc[nt].w, acc[mt][nt]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int nt = 0; nt < WNT; ++nt) { bc[nt] = bn[nt]; bn[nt] = bn2[nt]; }
#pragma unroll
      for (int mt = 0; mt < WMT; ++mt) ac[mt] = an[mt];
    }
#ifdef ODVAE_STAMPS
    const unsigned long long st_b0 = ODVAE_T();
#endif
    if (more) store_halo(smem + ((ch + 1) & 1) * HPIX * HS);
    __syncthreads();
#ifdef ODVAE_STAMPS
    st_bar += ODVAE_T() - st_b0;
#endif
  }
#ifdef ODVAE_STAMPS
  const unsigned long long st_mma = ODVAE_T();
#endif

  // Epilogue: stores only (bias and residual were folded into the accumulators' initial value)
  const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(p.y + (int64_t)n * p.Ho * p.Wo * p.Cout, 0, img_bytes, 0x00020000);
  const bool relu = p.act != 0;
#pragma unroll
  for (int mt = 0; mt < WMT; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const unsigned pb = pix_byte(mt, r);
#pragma unroll
      for (int nt = 0; nt < WNT; ++nt) {
        const float v = acc[mt][nt][r];
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(relu ? fmaxf(v, 0.f) : v), yrsrc, pb + cobyte[nt], 0, 0);
      }
    }
#ifdef ODVAE_STAMPS
  if (blockIdx.y == 0 && blockIdx.x < 4096 && lane == 0) {
    const unsigned long long st_done = ODVAE_T();
    unsigned long long* o = g_stamps + ((size_t)blockIdx.x * 4 + wave) * 4;
    o[0] = st_loop - st_start; o[1] = st_mma - st_loop; o[2] = st_bar; o[3] = st_done - st_mma;
  }
#endif
}

// OIHW -> fwd pack [t][CinP/4][CoutP][4] and dgrad pack [t'][CoutP_d/4][CinP_d][4] (flipped taps)
__global__ void conv3x3_pack_kernel(const float* __restrict__ w, int Cout, int Cin,
                                    float* __restrict__ fwd, int CinP_f, int CoutP_f,
                                    float* __restrict__ dgr, int CoutP_d, int CinP_d) {
  // fwd pack: reduction channels = Cin (padded CinP_f), output channels = Cout (padded CoutP_f)
  const int64_t nf = fwd ? (int64_t)9 * CinP_f * CoutP_f : 0;
  // dgrad pack: reduction channels = Cout (padded CoutP_d), output channels = Cin (padded CinP_d)
  const int64_t nd = dgr ? (int64_t)9 * CoutP_d * CinP_d : 0;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < nf + nd;
       idx += (int64_t)gridDim.x * blockDim.x) {
    if (idx < nf) {
      int64_t r = idx;
      const int j = r & 3; r >>= 2;
      const int co = (int)(r % CoutP_f); r /= CoutP_f;
      const int q = (int)(r % (CinP_f / 4)); const int tap = (int)(r / (CinP_f / 4));
      const int ci = 4 * q + j;
      fwd[idx] = (co < Cout && ci < Cin) ? w[((int64_t)co * Cin + ci) * 9 + tap] : 0.f;
    } else {
      int64_t r = idx - nf;
      const int j = r & 3; r >>= 2;
      const int ci = (int)(r % CinP_d); r /= CinP_d;
      const int q = (int)(r % (CoutP_d / 4)); const int tap = (int)(r / (CoutP_d / 4));
      const int co = 4 * q + j;
      dgr[idx - nf] = (co < Cout && ci < Cin) ? w[((int64_t)co * Cin + ci) * 9 + (8 - tap)] : 0.f;
    }
  }
}

// Upsample (nearest 2x) + conv3x3 with the taps that hit the same low-res pixel pre-summed.
// fwd16 [cls*4 + a*2 + b][CinP/4][CoutP][4], cls = 2*py + px: weight of input pixel (y + py - 1 + a, x + px - 1 + b) for output
//   (2y + py, 2x + px) = sum of W[kh][kw] over kh in R(py, a), kw in R(px, b);  R(0,0)={0} R(0,1)={1,2} R(1,0)={0,1} R(1,1)={2}
// dgr16 [u*4 + v][CoutP_d/4][CinP_d][4]: weight of dy pixel (2y + u - 1, 2x + v - 1) for dx (y, x) = sum of W[kh][kw] over
//   kh in Q(u), kw in Q(v);  Q(0)={2} Q(1)={1,2} Q(2)={0,1} Q(3)={0}
__device__ __forceinline__ float up_row_sum(const float* w9, int lo_h, int hi_h, int lo_w, int hi_w) {
  float s = 0.f;
  for (int kh = lo_h; kh <= hi_h; ++kh)
    for (int kw = lo_w; kw <= hi_w; ++kw) s += w9[kh * 3 + kw];
  return s;
}
__global__ void conv3x3_pack_up_kernel(const float* __restrict__ w, int Cout, int Cin,
                                       float* __restrict__ fwd, int CinP_f, int CoutP_f,
                                       float* __restrict__ dgr, int CoutP_d, int CinP_d) {
  const int64_t nf = fwd ? (int64_t)16 * CinP_f * CoutP_f : 0;
  const int64_t nd = dgr ? (int64_t)16 * CoutP_d * CinP_d : 0;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < nf + nd;
       idx += (int64_t)gridDim.x * blockDim.x) {
    if (idx < nf) {
      int64_t r = idx;
      const int j = r & 3; r >>= 2;
      const int co = (int)(r % CoutP_f); r /= CoutP_f;
      const int q = (int)(r % (CinP_f / 4)); const int t16 = (int)(r / (CinP_f / 4));
      const int ci = 4 * q + j;
      const int cls = t16 >> 2, a = (t16 >> 1) & 1, b = t16 & 1, py = cls >> 1, px = cls & 1;
      // R(p, t): p=0,t=0 -> {0}; p=0,t=1 -> {1,2}; p=1,t=0 -> {0,1}; p=1,t=1 -> {2}
      const int lo_h = py == 0 ? (a == 0 ? 0 : 1) : (a == 0 ? 0 : 2), hi_h = py == 0 ? (a == 0 ? 0 : 2) : (a == 0 ? 1 : 2);
      const int lo_w = px == 0 ? (b == 0 ? 0 : 1) : (b == 0 ? 0 : 2), hi_w = px == 0 ? (b == 0 ? 0 : 2) : (b == 0 ? 1 : 2);
      fwd[idx] = (co < Cout && ci < Cin) ? up_row_sum(w + ((int64_t)co * Cin + ci) * 9, lo_h, hi_h, lo_w, hi_w) : 0.f;
    } else {
      int64_t r = idx - nf;
      const int j = r & 3; r >>= 2;
      const int ci = (int)(r % CinP_d); r /= CinP_d;
      const int q = (int)(r % (CoutP_d / 4)); const int t16 = (int)(r / (CoutP_d / 4));
      const int co = 4 * q + j;
      const int u = t16 >> 2, v = t16 & 3;
      const int lo_h = u == 0 ? 2 : (u == 1 ? 1 : 0), hi_h = u == 0 ? 2 : (u == 1 ? 2 : (u == 2 ? 1 : 0));
      const int lo_w = v == 0 ? 2 : (v == 1 ? 1 : 0), hi_w = v == 0 ? 2 : (v == 1 ? 2 : (v == 2 ? 1 : 0));
      dgr[idx - nf] = (co < Cout && ci < Cin) ? up_row_sum(w + ((int64_t)co * Cin + ci) * 9, lo_h, hi_h, lo_w, hi_w) : 0.f;
    }
  }
}

// ---- thin input: 3 channels in (conv_in on the RGB crop; the data gradient of conv_out, whose dy has 3 channels) -------
// K = 9 taps x 3 channels = 27 fits ONE 32-deep MFMA operand, so instead of padding Cin to a 32-channel chunk per tap
// (10x wasted MFMA work) a wave multiplies a 32-pixel row segment's im2col row block [32 px][27 -> 32] with the whole
// weight matrix [32][128 co], which it keeps in 64 registers for its lifetime.  No LDS, no barriers; A fragments are
// gathers from the 25 MB input (cache resident), the output is written once.
__global__ __launch_bounds__(256, 2) void conv3x3_thin_in_kernel(ConvParams p) {
  const int lane = threadIdx.x & 63, li = lane & 31, kk = lane >> 5;
  const int wave_in_grid = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
  const int c0 = blockIdx.y * 128;
  const int QT = p.CinP / 4;
  // k-step s of this lane is k = 2s + kk -> (tap, ci) = (k / 3, k % 3), taps past 8 multiply zeros
  float bw[4][16];
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    const int k = 2 * s + kk, tap = k / 3, ci = k - 3 * tap;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      const int co = c0 + ct * 32 + li;
      const float w = p.wpk[((int64_t)min(tap, 8) * QT * p.CoutP + min(co, p.CoutP - 1)) * 4 + ci];
      bw[ct][s] = (tap < 9 && co < p.CoutP) ? w : 0.f;
    }
  }
  float bv[4];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) bv[ct] = p.bias ? p.bias[min(c0 + ct * 32 + li, p.Cout - 1)] : 0.f;
  const int segs = p.Wo / 32;                 // host guarantees Wo % 32 == 0
  const int ntiles = p.N * p.Ho * segs;
  const bool relu = p.act != 0;
  for (int tile = __builtin_amdgcn_readfirstlane(wave_in_grid); tile < ntiles; tile += nwaves) {
    const int seg = tile % segs, row = tile / segs;       // row = n * Ho + y
    const int y = row % p.Ho, x0 = seg * 32;
    float a[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int k = 2 * s + kk, tap = k / 3, ci = k - 3 * tap;
      const int iy = y + tap / 3 - 1, ix = x0 + li + tap % 3 - 1;
      const bool ok = tap < 9 && iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi;
      const float v = p.x[((int64_t)(row - y + min(max(iy, 0), p.Hi - 1)) * p.Wi + min(max(ix, 0), p.Wi - 1)) * 3 + ci];
      a[s] = ok ? v : 0.f;
    }
    f32x16 acc[4];
    // the tile's 32 pixels x Cout channels are one contiguous run of y: a buffer descriptor over exactly that run (base wave-uniform),
    // ONE per-lane offset (row 4 (lane >> 5), channel c0 + li), the accumulator row as a scalar offset and the channel tile as an
    // immediate -- sixty-four 64-bit store addresses, loop-invariant and therefore hoisted and spilled (33 registers), are gone
    const int64_t obase = ((int64_t)row * p.Wo + x0) * p.Cout;
    const int run = 32 * p.Cout * 4, rstep = p.Cout * 4;
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(p.y + obase, 0, run, 0x00020000);
    const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.residual ? p.residual : p.y) + obase, 0,
                                                                         p.residual ? run : 0, 0x00020000);
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      const unsigned vo = (c0 + ct * 32 + li < p.Cout) ? (unsigned)((4 * kk * p.Cout + c0 + ct * 32 + li) * 4) : 0x7FFFFFF0u;
#pragma unroll
      for (int r = 0; r < 16; ++r)
        acc[ct][r] = bv[ct] + (p.residual ? __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rrs, vo, ((r & 3) + 8 * (r >> 2)) * rstep, 0)) : 0.f);
    }
#pragma unroll
    for (int s = 0; s < 16; ++s)
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) acc[ct] = mfma32(a[s], bw[ct][s], acc[ct]);
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      const unsigned vo = (c0 + ct * 32 + li < p.Cout) ? (unsigned)((4 * kk * p.Cout + c0 + ct * 32 + li) * 4) : 0x7FFFFFF0u;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = acc[ct][r];
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(relu ? fmaxf(v, 0.f) : v), yrs, vo, ((r & 3) + 8 * (r >> 2)) * rstep, 0);
      }
    }
  }
}

// ---- thin output: at most 3 channels out (decoder.conv_out, 128 -> 3: the reconstruction) ---------------------------------
// The implicit-GEMM kernel pads the output channels to 32: 10x wasted MFMA work (1.29 ms per step).  Here the nine taps of the three
// output channels form ONE 32-wide MFMA operand instead: Y27[px][tap * 3 + co] = sum_ci x[px][ci] w[co][ci][tap] is a 1x1 product
// over the tile's pixels INCLUDING its halo ring (27 of 32 columns useful, K = Cin), kept in LDS, and an output pixel is the sum of
// nine of its entries, one per tap, each taken at that tap's neighbour: y[p][co] = b[co] + sum_tap Y27[p + off(tap)][3 tap + co].
// Block = 8 x 32 output pixels; Y27 for the 10 x 34 halo region = eleven 32-pixel MFMA row tiles shared by four waves; the weight
// matrix W27 [Cin][32] sits in Cin / 2 registers per lane.  Out-of-image halo pixels read as zeros, so their Y27 rows are zero: the
// conv's zero padding.  No activation, no residual (conv_out has neither).
constexpr int TO_TH = 8, TO_TW = 32, TO_HW = TO_TW + 2, TO_HPIX = (TO_TH + 2) * TO_HW, TO_MT = (TO_HPIX + 31) / 32;
template <int CIN>
__global__ __launch_bounds__(256) void conv3x3_thin_out_kernel(ConvParams p) {
  __shared__ float Y[TO_MT * 32][32 + 1];     // [halo pixel][tap * 3 + co], +1: the nine-term gather walks rows
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, kk = lane >> 5;
  int t = blockIdx.x;
  const int tx = t % p.tiles_x; t /= p.tiles_x;
  const int ty = t % p.tiles_y; const int n = t / p.tiles_y;
  const int oy0 = ty * TO_TH, ox0 = tx * TO_TW;
  const int QT = p.CinP / 4;
  // B operand: lane (column nn = li, k parity kk) holds W27[2 s + kk][nn] for s = 0 .. CIN/2 - 1
  float bw[CIN / 2];
  {
    const int tap = li / 3, co = li - 3 * tap;
#pragma unroll
    for (int s = 0; s < CIN / 2; ++s) {
      const int ci = 2 * s + kk;
      bw[s] = (li < 27 && co < p.Cout) ? p.wpk[(((int64_t)tap * QT + (ci >> 2)) * p.CoutP + co) * 4 + (ci & 3)] : 0.f;
    }
  }
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x) + (int64_t)n * p.Hi * p.Wi * p.Cin, 0,
                                                                       p.Hi * p.Wi * p.Cin * 4, 0x00020000);
  for (int mt = wave; mt < TO_MT; mt += 4) {
    const int hp = mt * 32 + li;                         // halo pixel of this lane's MFMA row
    const int hy = hp / TO_HW, hx = hp - hy * TO_HW;
    const int iy = oy0 - 1 + hy, ix = ox0 - 1 + hx;
    const bool ok = hp < TO_HPIX && iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi;
    const unsigned voff = ok ? (unsigned)((iy * p.Wi + ix) * p.Cin) * 4u : 0x7FFFFFF0u;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // ALL of the pixel's channels requested before the first MFMA (CIN / 4 sixteen-byte loads per lane, 128 registers at CIN = 128): with four
    // loads in flight per step the kernel was eight dependent memory round trips per 32 halo pixels -- 533 us for the 1.07 GB it reads (2 TB/s)
    u32x4 v[CIN / 4];
#pragma unroll
    for (int j = 0; j < CIN / 4; ++j) v[j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, voff, j * 16, 0));
    __builtin_amdgcn_sched_barrier(0);      // (pinned: hipcc otherwise sinks each load back to its MFMA)
#pragma unroll
    for (int j = 0; j < CIN / 4; ++j) {
      // k-step s of this lane is channel 2 s + kk: of channels 4q .. 4q+3 the lane takes q*4 + kk (s = 2q) and q*4 + 2 + kk (s = 2q + 1)
      const float a0 = __uint_as_float(kk ? v[j].y : v[j].x), a1 = __uint_as_float(kk ? v[j].w : v[j].z);
      acc = mfma32(a0, bw[2 * j], acc);
      acc = mfma32(a1, bw[2 * j + 1], acc);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) Y[mt * 32 + acc_row(r, lane)][li] = acc[r];
  }
  __syncthreads();
  // one output pixel per thread: nine taps x Cout values out of its 3 x 3 neighbourhood of Y27 rows
  const int r = tid / TO_TW, c = tid - r * TO_TW;
  const int oy = oy0 + r, ox = ox0 + c;
  if (oy < p.Ho && ox < p.Wo) {
    float o[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const float* row = Y[(r + tap / 3) * TO_HW + c + tap % 3];
#pragma unroll
      for (int co = 0; co < 3; ++co) o[co] += row[3 * tap + co];
    }
    float* dst = p.y + (((int64_t)n * p.Ho + oy) * p.Wo + ox) * p.Cout;
    for (int co = 0; co < p.Cout; ++co) dst[co] = o[co] + (p.bias ? p.bias[co] : 0.f);
  }
}

constexpr int round_up(int a, int b) { return (a + b - 1) / b * b; }

}  // namespace

extern "C" {

#ifdef ODVAE_STAMPS
int odvae_debug_read_stamps(unsigned long long* host, size_t count) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_stamps), count * sizeof(unsigned long long));
}
#endif

// padded reduction / output channel counts of a weight pack (reduce = channels summed over)
int odvae_conv3x3_pack_reduce_pad(int c_reduce) { return round_up(c_reduce, 32); }
int odvae_conv3x3_pack_out_pad(int c_out) { return c_out <= 32 ? 32 : round_up(c_out, 128); }

// floats in a pack whose reduction axis has c_reduce channels and output axis c_out channels
size_t odvae_conv3x3_pack_floats(int c_reduce, int c_out) {
  return (size_t)9 * odvae_conv3x3_pack_reduce_pad(c_reduce) * odvae_conv3x3_pack_out_pad(c_out);
}

// w: OIHW [Cout][Cin][3][3].  fwd_pack (may be null) has odvae_conv3x3_pack_floats(Cin, Cout) floats,
// dgrad_pack (may be null) has odvae_conv3x3_pack_floats(Cout, Cin) floats.
int odvae_conv3x3_pack_f32(const float* w, int Cout, int Cin, float* fwd_pack, float* dgrad_pack, void* stream) {
  ODVAE_CHECK_ARG(w && Cout > 0 && Cin > 0, "conv3x3_pack: bad arguments");
  const int CinP_f = odvae_conv3x3_pack_reduce_pad(Cin), CoutP_f = odvae_conv3x3_pack_out_pad(Cout);
  const int CoutP_d = odvae_conv3x3_pack_reduce_pad(Cout), CinP_d = odvae_conv3x3_pack_out_pad(Cin);
  const int64_t total = (fwd_pack ? (int64_t)9 * CinP_f * CoutP_f : 0) + (dgrad_pack ? (int64_t)9 * CoutP_d * CinP_d : 0);
  if (total == 0) return ODVAE_OK;
  const int blocks = (int)std::min<int64_t>(ceil_div64(total, 256), 2048);
  hipLaunchKernelGGL(conv3x3_pack_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                     w, Cout, Cin, fwd_pack, CinP_f, CoutP_f, dgrad_pack, CoutP_d, CinP_d);
  ODVAE_LAUNCH_CHECK("conv3x3_pack");
  return ODVAE_OK;
}

// 16-tap packs of an Upsample conv (modes 5 / 6): fwd16 has 16 * reduce_pad(Cin) * out_pad(Cout) floats, dgrad16 has
// 16 * reduce_pad(Cout) * out_pad(Cin) floats; either may be null
size_t odvae_conv3x3_up_pack_floats(int c_reduce, int c_out) {
  return (size_t)16 * odvae_conv3x3_pack_reduce_pad(c_reduce) * odvae_conv3x3_pack_out_pad(c_out);
}
int odvae_conv3x3_pack_up_f32(const float* w, int Cout, int Cin, float* fwd16, float* dgrad16, void* stream) {
  ODVAE_CHECK_ARG(w && Cout > 0 && Cin > 0, "conv3x3_pack_up: bad arguments");
  const int CinP_f = odvae_conv3x3_pack_reduce_pad(Cin), CoutP_f = odvae_conv3x3_pack_out_pad(Cout);
  const int CoutP_d = odvae_conv3x3_pack_reduce_pad(Cout), CinP_d = odvae_conv3x3_pack_out_pad(Cin);
  const int64_t total = (fwd16 ? (int64_t)16 * CinP_f * CoutP_f : 0) + (dgrad16 ? (int64_t)16 * CoutP_d * CinP_d : 0);
  if (total == 0) return ODVAE_OK;
  const int blocks = (int)std::min<int64_t>(ceil_div64(total, 256), 2048);
  hipLaunchKernelGGL(conv3x3_pack_up_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                     w, Cout, Cin, fwd16, CinP_f, CoutP_f, dgrad16, CoutP_d, CinP_d);
  ODVAE_LAUNCH_CHECK("conv3x3_pack_up");
  return ODVAE_OK;
}

// y = act(conv3x3(x) (+bias) (+residual)), act: 0 none | 1 ReLU.  mode: 0 stride 1 pad 1 | 1 pad(0,1,0,1)+stride 2 |
// 2 nearest-2x-upsample then stride 1 pad 1 | 3 transposed stride 2 (data gradient of mode 1) |
// 5 = mode 2 computed per output parity class from a 16-tap pack (odvae_conv3x3_pack_up_f32 fwd16) |
// 6 = data gradient of mode 2 / 5: x = dy [N][2Ho][2Wo][Cin], y = dx [N][Ho][Wo][Cout], wpk = dgrad16.
// wpk: pack with reduction axis Cin and output axis Cout (for data gradients pass the dgrad pack,
// Cin = channels of x (= dy), Cout = channels of y (= dx)).
int odvae_conv3x3_f32(int mode, const float* x, int N, int Hi, int Wi, int Cin,
                      const float* wpk, int Cout, const float* bias, const float* residual,
                      float* y, int Ho, int Wo, int act, void* stream) {
  ODVAE_CHECK_ARG(x && wpk && y, "conv3x3: null operand");
  ODVAE_CHECK_ARG(N > 0 && Hi > 0 && Wi > 0 && Cin > 0 && Cout > 0, "conv3x3: empty shape");
  ODVAE_CHECK_ARG((mode >= 0 && mode <= 3) || mode == 5 || mode == 6, "conv3x3: mode %d", mode);
  if (mode == 0) ODVAE_CHECK_ARG(Ho == Hi && Wo == Wi, "conv3x3 mode 0: Ho,Wo must equal Hi,Wi");
  if (mode == 1) ODVAE_CHECK_ARG(Hi % 2 == 0 && Wi % 2 == 0 && Ho == Hi / 2 && Wo == Wi / 2, "conv3x3 mode 1: need even Hi,Wi and Ho=Hi/2");
  if (mode == 2 || mode == 3 || mode == 5) ODVAE_CHECK_ARG(Ho == 2 * Hi && Wo == 2 * Wi, "conv3x3 mode %d: need Ho=2*Hi", mode);
  if (mode == 6) ODVAE_CHECK_ARG(Hi == 2 * Ho && Wi == 2 * Wo, "conv3x3 mode 6: need Hi=2*Ho");
  ODVAE_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)wpk & 15) == 0, "conv3x3: x/wpk must be 16-byte aligned");
  ODVAE_CHECK_ARG((int64_t)Hi * Wi * Cin * 4 < 0x7FFFFFF0ll && (int64_t)Ho * Wo * Cout * 4 < 0x7FFFFFF0ll,
                  "conv3x3: one input / output image must stay below 2 GiB");

  ConvParams p;
  p.x = x; p.wpk = wpk; p.bias = bias; p.residual = residual; p.y = y;
  p.N = N; p.Hi = Hi; p.Wi = Wi; p.Cin = Cin; p.Ho = Ho; p.Wo = Wo; p.Cout = Cout;
  p.CinP = odvae_conv3x3_pack_reduce_pad(Cin); p.CoutP = odvae_conv3x3_pack_out_pad(Cout);
  p.tiles_x = ceil_div(mode == 5 ? Wi : Wo, TW); p.tiles_y = ceil_div(mode == 5 ? Hi : Ho, TH);   // mode 5 tiles the low-res grid
  p.act = act;
  const int64_t sp = (int64_t)p.tiles_x * p.tiles_y * N;
  ODVAE_CHECK_ARG(sp < (1ll << 31), "conv3x3: too many tiles");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bool narrow = Cout <= 32;
  dim3 block(256);
  static const bool thin_off = getenv("ODVAE_CONV_THIN_OFF") != nullptr;
  if (mode == 0 && Cin == 3 && Wo % 32 == 0 && !thin_off) {
    const int64_t ntiles = (int64_t)N * Ho * (Wo / 32);
    const int blocks = (int)std::min<int64_t>(1024, ceil_div64(ntiles, 4));
    hipLaunchKernelGGL(conv3x3_thin_in_kernel, dim3(blocks, ceil_div(Cout, 128)), block, 0, st, p);
    ODVAE_LAUNCH_CHECK("conv3x3 thin-in");
    return ODVAE_OK;
  }
  static const bool thin_out_off = getenv("ODVAE_CONV_THIN_OUT_OFF") != nullptr;
  if (mode == 0 && Cout <= 3 && Cin == 128 && !residual && act == 0 && !thin_out_off) {      // decoder.conv_out: the reconstruction
    p.tiles_x = ceil_div(Wo, TO_TW); p.tiles_y = ceil_div(Ho, TO_TH);
    ODVAE_CHECK_ARG((int64_t)p.tiles_x * p.tiles_y * N < (1ll << 31), "conv3x3: too many tiles");
    hipLaunchKernelGGL(conv3x3_thin_out_kernel<128>, dim3((unsigned)(p.tiles_x * p.tiles_y * N)), block, 0, st, p);
    ODVAE_LAUNCH_CHECK("conv3x3 thin-out");
    return ODVAE_OK;
  }
  // ODVAE_CONV_VARIANT=0 selects the first per-tap-barrier kernel (kept for in-process A/B timing and as the
  // "before" of profiles/r01_conv3x3_pmc.md); 1 (default) = v2
  static const int variant = getenv("ODVAE_CONV_VARIANT") ? atoi(getenv("ODVAE_CONV_VARIANT")) : 1;
  const bool use_v1 = variant == 0 && mode <= 3;
#define ODVAE_CONV_LAUNCH(KERNEL, MODE, KC)                                                                      \
  if (narrow) hipLaunchKernelGGL((KERNEL<MODE, KC, 1, 1, 4, 1>), dim3((unsigned)sp, p.CoutP / 32), block, 0, st, p); \
  else hipLaunchKernelGGL((KERNEL<MODE, KC, 2, 2, 2, 2>), dim3((unsigned)sp, p.CoutP / 128), block, 0, st, p)
  if (use_v1) {
    switch (mode) {
      case 0: ODVAE_CONV_LAUNCH(conv3x3_kernel, 0, 16); break;
      case 1: ODVAE_CONV_LAUNCH(conv3x3_kernel, 1, 8); break;
      case 2: ODVAE_CONV_LAUNCH(conv3x3_kernel, 2, 16); break;
      default: ODVAE_CONV_LAUNCH(conv3x3_kernel, 3, 16); break;
    }
  } else {
    switch (mode) {
      case 0: ODVAE_CONV_LAUNCH(conv3x3_kernel_v2, 0, 32); break;
      case 1: ODVAE_CONV_LAUNCH(conv3x3_kernel_v2, 1, 8); break;
      case 2: ODVAE_CONV_LAUNCH(conv3x3_kernel_v2, 2, 32); break;
      case 5:
        if (narrow) hipLaunchKernelGGL((conv3x3_kernel_v2<5, 32, 1, 1, 4, 1>), dim3((unsigned)sp, p.CoutP / 32, 4), block, 0, st, p);
        else hipLaunchKernelGGL((conv3x3_kernel_v2<5, 32, 2, 2, 2, 2>), dim3((unsigned)sp, p.CoutP / 128, 4), block, 0, st, p);
        break;
      case 6: ODVAE_CONV_LAUNCH(conv3x3_kernel_v2, 6, 8); break;
      default: ODVAE_CONV_LAUNCH(conv3x3_kernel_v2, 3, 32); break;
    }
  }
#undef ODVAE_CONV_LAUNCH
  ODVAE_LAUNCH_CHECK("conv3x3");
  return ODVAE_OK;
}

}  // extern "C"
