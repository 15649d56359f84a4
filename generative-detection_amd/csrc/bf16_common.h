// bf16 helpers for the mixed-precision path (BASELINE.json configs[4]: bf16 activations in HBM, fp32 master weights,
// fp32 accumulation on v_mfma_f32_32x32x16_bf16).  gfx950 only.
#pragma once
#include "common.h"

typedef unsigned short bf16_t;                                   // raw bits in HBM / LDS
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));      // one MFMA A/B fragment (4 VGPRs)
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float((unsigned)v << 16); }
// round to nearest even, NaN kept quiet (what torch's .to(bfloat16) does)
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  unsigned u = __float_as_uint(f);
  if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (bf16_t)((u >> 16) | 0x40);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}
// two floats -> one dword of two bf16 (lo in bits 0..15); v_cvt_pk_bf16_f32 rounds to nearest even
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
#if defined(__HIP_DEVICE_COMPILE__)
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
#else
  (void)lo; (void)hi; return 0;
#endif
}
__device__ __forceinline__ float bf16_lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16_hi(unsigned w) { return __uint_as_float(w & 0xFFFF0000u); }

// D(32x32) += A(32x16) * B(16x32), bf16 in, f32 accumulate.  lane l (r = l&31, h = l>>5) supplies A[row r][k = 8h+j] and
// B[k = 8h+j][col r], j = 0..7; D register i of lane l is D[row (i&3) + 8(i>>2) + 4h][col r]  (tools/tr_probe.hip)
__device__ __forceinline__ f32x16 mfma_bf16(bf16x8 a, bf16x8 b, f32x16 c) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
#else
  (void)a; (void)b; return c;
#endif
}
__device__ __forceinline__ bf16x8 frag_from_u32x4(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }

// ds_read_b64_tr_b16: per 16-lane group a 4-row x 16-column block of 16-bit elements, delivered column-major.  Lane 4q+p of
// the group passes the LDS address of (row q, column 4p); lane i receives column i of rows 0..3 (element q = row q).
// EXEC must be all ones.  Address must be 8-byte aligned.
__device__ __forceinline__ s16x4 lds_read_tr16(const bf16_t* lds_addr) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)lds_addr);
#else
  (void)lds_addr; return s16x4{0, 0, 0, 0};
#endif
}
__device__ __forceinline__ bf16x8 frag_from_tr(s16x4 lo, s16x4 hi) {
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

// ---- LDS in 32-bit byte addresses, LDS-DMA by inline asm (shared by the attention and the wide-tile conv kernels) ----------------
typedef __attribute__((address_space(3))) char lds_char_t;
__device__ __forceinline__ unsigned lds_addr_of(const void* generic) { return (unsigned)(uintptr_t)(lds_char_t*)generic; }
__device__ __forceinline__ u32x4 lds_ld128(unsigned a) { return *(const __attribute__((address_space(3))) u32x4*)(uintptr_t)a; }
__device__ __forceinline__ f32x4 lds_ld128f(unsigned a) { return *(const __attribute__((address_space(3))) f32x4*)(uintptr_t)a; }
__device__ __forceinline__ void lds_st128(unsigned a, u32x4 v) { *(__attribute__((address_space(3))) u32x4*)(uintptr_t)a = v; }
__device__ __forceinline__ void lds_st128f(unsigned a, f32x4 v) { *(__attribute__((address_space(3))) f32x4*)(uintptr_t)a = v; }
__device__ __forceinline__ s16x4 lds_ld_tr(unsigned a) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(uintptr_t)a);
#else
  (void)a; return s16x4{0, 0, 0, 0};
#endif
}

// LDS-DMA by inline asm: hipcc counts a `buffer_load ... lds` it knows about in vmcnt and, unable to tell the LDS-DMA's destination
// from the tiles being read, waits for it (vmcnt(0)) in front of the next ds_read -- the fetch of the tile two periods ahead would be
// waited for at the top of the period that issues it.  Hidden in an asm statement the load is invisible to that bookkeeping; the
// kernels wait for it themselves (one `s_waitcnt vmcnt(0)` in front of each period's barrier).  M0 carries the LDS destination
// (wave-uniform) and is restored; the s_nop covers the SALU-write-M0 -> LDS-DMA hazard.
typedef int i32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ i32x4_t rsrc_words(const void* base, unsigned bytes) {
  const uint64_t a = (uint64_t)base;
  i32x4_t r;
  r.x = (int)(unsigned)a; r.y = (int)((unsigned)(a >> 32) & 0xFFFFu); r.z = (int)bytes; r.w = 0x00020000;
  return r;
}
__device__ __forceinline__ void lds_dma16(i32x4_t rsrc, unsigned lds_dst, unsigned voff) {
#if defined(__HIP_DEVICE_COMPILE__)
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(lds_dst), "s"(rsrc) : "memory");
#else
  (void)rsrc; (void)lds_dst; (void)voff;
#endif
}
// the same with a scalar byte offset added to every lane's address (a per-chunk step costs no vector instruction)
__device__ __forceinline__ void lds_dma16_s(i32x4_t rsrc, unsigned lds_dst, unsigned voff, unsigned soff) {
#if defined(__HIP_DEVICE_COMPILE__)
  unsigned keep;
  // (s_nop 2: with the two s_mov in front five wait states between a VALU write of `soff` / `rsrc` -- v_readlane of a spilled SGPR --
  // and the vector-memory instruction that reads them; hipcc does not look into inline asm for that hazard)
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 2\n\tbuffer_load_dwordx4 %1, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(lds_dst), "s"(rsrc), "s"(soff) : "memory");
#else
  (void)rsrc; (void)lds_dst; (void)voff; (void)soff;
#endif
}
__device__ __forceinline__ void lds_dma4(i32x4_t rsrc, unsigned lds_dst, unsigned voff) {
#if defined(__HIP_DEVICE_COMPILE__)
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dword %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(lds_dst), "s"(rsrc) : "memory");
#else
  (void)rsrc; (void)lds_dst; (void)voff;
#endif
}
