// Object-patch extraction on the device: crop (zero fill outside the camera image) -> Pillow-exact BILINEAR resize of the
// u8 crop to S x S -> ToTensor (u8 / 255 as f32), plus the NEAREST-resized 2-d box mask.  HBM-bound byte work, gfx950.
//
// Replaces the per-instance PIL path of the reference's dataset, src/data/datasets/nuscenes.py:159-192
//   patch = img_pil.crop((x1, y1, x2, y2))                                          (:159)
//   patch.resize((w, h), resample=BILINEAR, reducing_gap=1.0)                       (:176)
//   mask_bool[y1p:y2p, x1p:x2p] = True; Image.fromarray(mask_bool).resize(.., NEAREST)  (:178-189)
//   transforms.ToTensor()                                                           (:190-192)
// Pillow (third-party, 12.2.0 in this image) resamples 8-bit images in fixed point: per output index a window
// [xmin, xmin+n) of <= 5 taps with integer coefficients k = round(w * 2^22) (triangle filter, renormalised at the crop
// border), the horizontal pass rounded to u8 before the vertical pass, each as (2^21 + sum p*k) >> 22 clamped to 0..255.
// The coefficient / nearest-index tables depend only on (crop size, S); the host builds them once per crop size in f64
// exactly as Pillow's precompute_coeffs does (patches.py) and this kernel does integer arithmetic only, so the result is
// bit-identical to PIL.  Crops with size >= 2*S (where reducing_gap would add a box-reduce pass) are rejected on the host.
#include "common.h"

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;   // Pillow Resample.c, 8 bits per channel
constexpr int TAB_INTS = 8;                  // per output index: k[0..4], first source index, taps, nearest source index

struct PatchParams {
  const uint8_t* const* images;   // [B] device pointers, u8 HWC RGB
  const int32_t* geom;            // [B][8]: img_h, img_w, crop_x1, crop_y1, crop_size, table_slot, 0, 0
  const int32_t* mask_rect;       // [B][4]: x_start, x_stop, y_start, y_stop in crop coordinates (slice-normalised)
  const int32_t* tables;          // [n_slots][S][TAB_INTS]
  float* patch;                   // [B][S][S][3]  (NHWC; logical NCHW channels_last)
  float* mask;                    // [B][S][S]
  int B, S, n_slots;
};

__device__ __forceinline__ int clip8(int v) { return min(max(v >> PRECISION_BITS, 0), 255); }

// block = 64 x 4 output pixels (one wave per output row); 3 contiguous floats per lane -> coalesced stores; source bytes
// come through L1/L2; 32-bit index arithmetic only (the host checks H*W*3 < 2^31)
__global__ __launch_bounds__(256) void patch_crop_resize_kernel(PatchParams p) {
  const int ox = blockIdx.x * 64 + (threadIdx.x & 63);
  const int oy = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int b = blockIdx.z;
  if (ox >= p.S || oy >= p.S) return;
  const int4 g0 = reinterpret_cast<const int4*>(p.geom)[b * 2];
  const int4 g1 = reinterpret_cast<const int4*>(p.geom)[b * 2 + 1];
  const int H = g0.x, W = g0.y, cx1 = g0.z, cy1 = g0.w;
  const int slot = min(max(g1.y, 0), p.n_slots - 1);
  const int4* tab = reinterpret_cast<const int4*>(p.tables) + slot * p.S * 2;
  const int4 tx0 = tab[ox * 2], tx1 = tab[ox * 2 + 1], ty0 = tab[oy * 2], ty1 = tab[oy * 2 + 1];
  const int kx[5] = {tx0.x, tx0.y, tx0.z, tx0.w, tx1.x}, ky[5] = {ty0.x, ty0.y, ty0.z, ty0.w, ty1.x};
  const int xmin = tx1.y, nx = min(tx1.z, 5), ymin = ty1.y, ny = min(ty1.z, 5);
  const uint8_t* img = p.images[b];
  const int last_dword = H * W * 3 - 4;   // a pixel is fetched as one unaligned dword that never leaves the image
  int xoff[5], kxm[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const int sx = cx1 + xmin + j;
    xoff[j] = min(max(sx, 0), W - 1) * 3;
    kxm[j] = (j < nx && sx >= 0 && sx < W) ? kx[j] : 0;   // outside the camera image the crop is 0
  }
  int acc0 = 1 << (PRECISION_BITS - 1), acc1 = acc0, acc2 = acc0;
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    if (i < ny) {
      const int sy = cy1 + ymin + i;
      const int row = min(max(sy, 0), H - 1) * W * 3;
      int h0 = 0, h1 = 0, h2 = 0;
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        if (j < nx) {
          const int off = row + xoff[j];
          const int at = min(off, last_dword);
          uint32_t rgb;
          __builtin_memcpy(&rgb, img + at, 4);
          rgb >>= 8 * (off - at);
          h0 += (int)(rgb & 255u) * kxm[j]; h1 += (int)((rgb >> 8) & 255u) * kxm[j]; h2 += (int)((rgb >> 16) & 255u) * kxm[j];
        }
      }
      const int half = 1 << (PRECISION_BITS - 1);
      const int kyi = (sy >= 0 && sy < H) ? ky[i] : 0;       // a row outside the image is all zeros: clip8(half) = 0
      acc0 += clip8(h0 + half) * kyi; acc1 += clip8(h1 + half) * kyi; acc2 += clip8(h2 + half) * kyi;
    }
  }
  const int idx = (b * p.S + oy) * p.S + ox;
  float* o = p.patch + (int64_t)idx * 3;
  o[0] = __fdiv_rn((float)clip8(acc0), 255.0f);
  o[1] = __fdiv_rn((float)clip8(acc1), 255.0f);
  o[2] = __fdiv_rn((float)clip8(acc2), 255.0f);
  const int4 mr = reinterpret_cast<const int4*>(p.mask_rect)[b];
  const int nxs = tx1.w, nys = ty1.w;
  p.mask[idx] = (nxs >= mr.x && nxs < mr.y && nys >= mr.z && nys < mr.w) ? 1.0f : 0.0f;
}

}  // namespace

extern "C" int odvae_patch_table_ints(int S) { return S * TAB_INTS; }

extern "C" int odvae_patch_crop_resize_u8(const void* d_images, const void* d_geom, const void* d_mask_rect, const void* d_tables,
                                          int n_slots, int B, int S, void* patch, void* mask, hipStream_t stream) {
  ODVAE_CHECK_ARG(d_images && d_geom && d_mask_rect && d_tables && patch && mask, "patch_crop_resize: null pointer");
  ODVAE_CHECK_ARG(B > 0 && S > 0 && S <= 4096 && n_slots > 0, "patch_crop_resize: bad sizes");   // images: >= 2 pixels each
  PatchParams p;
  p.images = (const uint8_t* const*)d_images; p.geom = (const int32_t*)d_geom; p.mask_rect = (const int32_t*)d_mask_rect;
  p.tables = (const int32_t*)d_tables; p.patch = (float*)patch; p.mask = (float*)mask;
  p.B = B; p.S = S; p.n_slots = n_slots;
  ODVAE_CHECK_ARG(B <= 65535 && (int64_t)B * S * S < (int64_t)1 << 31, "patch_crop_resize: batch too large for one launch");
  hipLaunchKernelGGL(patch_crop_resize_kernel, dim3(ceil_div(S, 64), ceil_div(S, 4), B), dim3(256), 0, stream, p);
  ODVAE_LAUNCH_CHECK("patch_crop_resize_kernel");
  return 0;
}
