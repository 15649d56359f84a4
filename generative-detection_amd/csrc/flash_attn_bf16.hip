// Fused single-head attention forward + backward, bf16 in HBM, fp32 softmax and accumulation, gfx950.
// softmax(q k^T * C^-1/2) v of [UPSTREAM] ldm AttnBlock.forward (src/modules/autoencodermodules/feat_encoder.py:2,
// feat_decoder.py:2) over T = H*W tokens with head dimension D = C (64 / 128 / 256 / 512): the T x T scores never leave
// the CU (the unfused f32 path keeps them in HBM: 1 GiB per image and block at T = 16 384).
//
// Input is the packed projection qkv [N][T][3C] (q | k | v), outputs o [N][T][C], lse2 [N][T] = log2 sum_j exp(s_ij * scale)
// (base-2 log-sum-exp of the scaled scores, what the backward needs to rebuild P without a second max/sum pass).
//
// Forward / dQ kernels: the QUERY is stationary.  A wave owns 32 query rows (Q, and dO in the backward, as MFMA B fragments in
// registers); K and V tiles of 32 keys stream through LDS, shared by the 4 waves (128 query rows per block).  Scores are
// computed transposed, S^T = K Q^T, so the query sits on the lane: row max / sum are in-register reductions plus ONE exchange
// with lane^32, and the probability tile is already the B operand of the next product (O^T += V^T P^T, dQ^T += K^T dS^T) --
// no LDS round trip for P.  V^T / K^T fragments come from the row-major tiles by ds_read_b64_tr_b16.
// dK/dV kernel: the KEY is stationary (K, V as B fragments; dK^T, dV^T accumulators), Q and dO tiles of 32 rows stream through
// LDS; S = Q K^T and dP = dO V^T have the key on the lane, P / dS feed dV^T += dO^T P and dK^T += Q^T dS directly.
// dQ comes from its own kernel instead of atomics: bit-reproducible, 7 instead of 5 products in the backward.
#include "bf16_common.h"

namespace {

constexpr int KB = 32;                     // keys (or query rows) per MFMA tile; a streamed LDS stage holds KBT = 32 or 64 of them
constexpr float LOG2E = 1.4426950408889634f;
constexpr float NEG_BIG = -1.0e30f;

struct FlashP {
  const bf16_t* qkv;    // [N][T][3C]
  const bf16_t* o;      // [N][T][C]      (backward: forward output; forward: written)
  const bf16_t* d_o;    // [N][T][C]      (backward)
  bf16_t* out;          // forward: o; backward: dqkv [N][T][3C]
  float* lse2;          // [N][T]
  const float* delta;   // [N][T] rowsum(dO * O)   (backward)
  int N, T, C;
  float scale;
};

__device__ __forceinline__ int acc_row32(int i, int h) { return (i & 3) + 8 * (i >> 2) + 4 * h; }

// v_exp_f32 as it is (exp2f() wraps it in range fix-ups worth five more VALU instructions per element; the arguments here are
// <= 8 after the running maximum is subtracted, and an underflow to 0 is the wanted result)
__device__ __forceinline__ float fast_exp2(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_exp2f(x);
#else
  return x;
#endif
}

// acc += A(32 rows of `tile`, all D columns) * B fragments bq[]; the row fragments are requested four k-steps ahead of their MFMAs
// (hipcc otherwise waits for every ds_read right before the MFMA that uses it: one LDS latency per MFMA at one wave per SIMD)
template <int D, int STRIDE>
__device__ __forceinline__ void rows_times_frags(const bf16_t* tile, int li, int h, const bf16x8 (&bq)[D / 16], f32x16& acc) {
  constexpr int G = 4, NG = D / 16 / G;
  const bf16_t* row = tile + li * STRIDE + 8 * h;
  u32x4 cur[G], nxt[G];
#pragma unroll
  for (int j = 0; j < G; ++j) cur[j] = *reinterpret_cast<const u32x4*>(row + 16 * j);
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    if (g + 1 < NG) {
#pragma unroll
      for (int j = 0; j < G; ++j) nxt[j] = *reinterpret_cast<const u32x4*>(row + 16 * (G * (g + 1) + j));
    }
#pragma unroll
    for (int j = 0; j < G; ++j) acc = mfma_bf16(frag_from_u32x4(cur[j]), bq[G * g + j], acc);
    if (g + 1 < NG) {
#pragma unroll
      for (int j = 0; j < G; ++j) cur[j] = nxt[j];
    }
  }
}
// same with two row tiles and two fragment sets at once (S and dP of the backward kernels): acc1 += A1 b1, acc2 += A2 b2
template <int D, int STRIDE>
__device__ __forceinline__ void rows_times_frags2(const bf16_t* t1, const bf16_t* t2, int li, int h, const bf16x8 (&b1)[D / 16],
                                                  const bf16x8 (&b2)[D / 16], f32x16& acc1, f32x16& acc2) {
  constexpr int G = 2, NG = D / 16 / G;
  const bf16_t* r1 = t1 + li * STRIDE + 8 * h;
  const bf16_t* r2 = t2 + li * STRIDE + 8 * h;
  u32x4 c1[G], c2[G], n1[G], n2[G];
#pragma unroll
  for (int j = 0; j < G; ++j) { c1[j] = *reinterpret_cast<const u32x4*>(r1 + 16 * j); c2[j] = *reinterpret_cast<const u32x4*>(r2 + 16 * j); }
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    if (g + 1 < NG) {
#pragma unroll
      for (int j = 0; j < G; ++j) {
        n1[j] = *reinterpret_cast<const u32x4*>(r1 + 16 * (G * (g + 1) + j));
        n2[j] = *reinterpret_cast<const u32x4*>(r2 + 16 * (G * (g + 1) + j));
      }
    }
#pragma unroll
    for (int j = 0; j < G; ++j) {
      acc1 = mfma_bf16(frag_from_u32x4(c1[j]), b1[G * g + j], acc1);
      acc2 = mfma_bf16(frag_from_u32x4(c2[j]), b2[G * g + j], acc2);
    }
    if (g + 1 < NG) {
#pragma unroll
      for (int j = 0; j < G; ++j) { c1[j] = n1[j]; c2[j] = n2[j]; }
    }
  }
}

// 8 consecutive accumulator registers 8s .. 8s+7 -> one bf16 B/A fragment (k order 16s + 8(j>>2) + 4h + (j&3))
__device__ __forceinline__ bf16x8 frag_from_acc(const f32x16& v, int s) {
  u32x4 w;
  w.x = pack_bf16x2(v[8 * s + 0], v[8 * s + 1]); w.y = pack_bf16x2(v[8 * s + 2], v[8 * s + 3]);
  w.z = pack_bf16x2(v[8 * s + 4], v[8 * s + 5]); w.w = pack_bf16x2(v[8 * s + 6], v[8 * s + 7]);
  return frag_from_u32x4(w);
}

// stage a [ROWS rows][W cols] bf16 tile (rows row0.., columns col0.. of a [T][ld] matrix behind `rsrc`) through registers
template <int W, int STRIDE, int NT, int ROWS = KB>
struct TileStage {
  static constexpr int V = ROWS * (W / 8), IT = (V + NT - 1) / NT;
  u32x4 reg[IT];
  __device__ __forceinline__ void fetch(const __amdgpu_buffer_rsrc_t& rsrc, int row0, int T, int ld, int col0, int tid) {
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const int f = tid + NT * i;
      const int r = f / (W / 8), q = f % (W / 8);
      const bool ok = f < V && row0 + r < T;
      reg[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, ok ? (unsigned)(((row0 + r) * ld + col0 + 8 * q) * 2) : 0x7FFFFFF0u, 0, 0);
    }
  }
  __device__ __forceinline__ void store(bf16_t* lds, int tid) {
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const int f = tid + NT * i;
      if (f < V) *reinterpret_cast<u32x4*>(lds + (f / (W / 8)) * STRIDE + 8 * (f % (W / 8))) = reg[i];
    }
  }
};

// V^T / K^T / Q^T / dO^T fragment (rows = 32 columns c0.. of the tile, k = the 16 tile rows of k-step s in accumulator order)
template <int STRIDE>
__device__ __forceinline__ bf16x8 tr_frag(const bf16_t* tile, int s, int c0, int lane) {
  const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3, hk = g >> 1;
  const bf16_t* base = tile + (16 * s + 4 * hk + q) * STRIDE + c0 + 16 * (g & 1) + 4 * pp;
  return frag_from_tr(lds_read_tr16(base), lds_read_tr16(base + 8 * STRIDE));
}

// ------------------------------------------------------------------------------------------------------------------------
// forward: grid (ceil(T/128), N, D/DV); slice z computes output columns [z*DV, (z+1)*DV)
// ------------------------------------------------------------------------------------------------------------------------
template <int D, int DV, int KBT>
__global__ __launch_bounds__(256) void flash_fwd_kernel(FlashP p) {
  constexpr int KSTR = D + 8, VSTR = DV + 32;
  constexpr int STAGE = KBT * KSTR + KBT * VSTR;            // bf16 per LDS stage (K tile then V tile)
  extern __shared__ __attribute__((aligned(16))) bf16_t smem[];   // 2 stages
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, h = lane >> 5;
  const int n = blockIdx.y, v0 = blockIdx.z * DV;
  const int q0 = blockIdx.x * 128 + wave * 32;
  const int C3 = 3 * p.C;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<bf16_t*>(p.qkv + (int64_t)n * p.T * C3), 0, p.T * C3 * 2, 0x00020000);
  const float c = p.scale * LOG2E;

  bf16x8 qf[D / 16];
  {
    const bool ok = q0 + li < p.T;
#pragma unroll
    for (int ks = 0; ks < D / 16; ++ks)
      qf[ks] = frag_from_u32x4(__builtin_amdgcn_raw_buffer_load_b128(rs, ok ? (unsigned)(((q0 + li) * C3 + 16 * ks + 8 * h) * 2) : 0x7FFFFFF0u, 0, 0));
  }
  f32x16 ot[DV / 32];
#pragma unroll
  for (int dt = 0; dt < DV / 32; ++dt)
#pragma unroll
    for (int i = 0; i < 16; ++i) ot[dt][i] = 0.f;
  float m = NEG_BIG, l = 0.f;

  // Pipeline: LDS stage (it & 1) holds tile `it`; the registers hold tile it + 1 (fetched during iteration it - 1).  At the top of
  // iteration `it` they are written to the other stage (last read in iteration it - 1, behind that iteration's barrier) and the
  // fetch of tile it + 2 is issued: a global load has a whole iteration to land, one barrier per iteration.
  TileStage<D, KSTR, 256, KBT> kst;
  TileStage<DV, VSTR, 256, KBT> vst;
  const int ntiles = (p.T + KBT - 1) / KBT;
  kst.fetch(rs, 0, p.T, C3, p.C, tid);
  vst.fetch(rs, 0, p.T, C3, 2 * p.C + v0, tid);
  kst.store(smem, tid);
  vst.store(smem + KBT * KSTR, tid);
  if (ntiles > 1) {
    kst.fetch(rs, KBT, p.T, C3, p.C, tid);
    vst.fetch(rs, KBT, p.T, C3, 2 * p.C + v0, tid);
  }
  __syncthreads();
  for (int it = 0; it < ntiles; ++it) {
    const bf16_t* Kt = smem + (it & 1) * STAGE;
    const bf16_t* Vt = Kt + KBT * KSTR;
    if (it + 1 < ntiles) {
      bf16_t* nx = smem + ((it + 1) & 1) * STAGE;
      kst.store(nx, tid);
      vst.store(nx + KBT * KSTR, tid);
      if (it + 2 < ntiles) {
        kst.fetch(rs, (it + 2) * KBT, p.T, C3, p.C, tid);
        vst.fetch(rs, (it + 2) * KBT, p.T, C3, 2 * p.C + v0, tid);
      }
    }
#pragma unroll
    for (int hf = 0; hf < KBT / KB; ++hf) {      // 32 keys at a time: one S^T tile in registers
      const int k0 = it * KBT + hf * KB;
      if (k0 >= p.T) break;                      // wave-uniform
      const bf16_t* Ks = Kt + hf * KB * KSTR;
      const bf16_t* Vs = Vt + hf * KB * VSTR;
      f32x16 st;
#pragma unroll
      for (int i = 0; i < 16; ++i) st[i] = 0.f;
      rows_times_frags<D, KSTR>(Ks, li, h, qf, st);   // S^T[key][q] = K Q^T
      const bool tail = k0 + KB > p.T;
      if (tail) {
#pragma unroll
        for (int i = 0; i < 16; ++i)
          if (k0 + acc_row32(i, h) >= p.T) st[i] = NEG_BIG;
      }
      float mx = fmaxf(fmaxf(fmaxf(st[0], st[1]), fmaxf(st[2], st[3])), fmaxf(fmaxf(st[4], st[5]), fmaxf(st[6], st[7])));
      mx = fmaxf(mx, fmaxf(fmaxf(fmaxf(st[8], st[9]), fmaxf(st[10], st[11])), fmaxf(fmaxf(st[12], st[13]), fmaxf(st[14], st[15]))));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * c;      // c > 0: the maximum of the scaled scores
      // Deferred maximum: the reference point m only moves when some row's scores exceed it by more than 2^8 (the exponentials
      // then stay below 256: harmless in f32 / bf16), so the 128-register rescale of O runs a few times per row, not per tile.
      if (__builtin_amdgcn_ballot_w64(mx > m + 8.f) != 0) {
        const float m_new = fmaxf(m, mx);
        const float alpha = fast_exp2(m - m_new);
        l *= alpha;
        m = m_new;
#pragma unroll
        for (int dt = 0; dt < DV / 32; ++dt)
#pragma unroll
          for (int i = 0; i < 16; ++i) ot[dt][i] *= alpha;
      }
      float ps = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) { st[i] = fast_exp2(fmaf(st[i], c, -m)); ps += st[i]; }
      l += ps;
#pragma unroll
      for (int s = 0; s < 2; ++s) {              // O^T[d][q] += V^T[d][key] P^T[key][q]
        const bf16x8 pb = frag_from_acc(st, s);
#pragma unroll
        for (int dt = 0; dt < DV / 32; ++dt) ot[dt] = mfma_bf16(tr_frag<VSTR>(Vs, s, dt * 32, lane), pb, ot[dt]);
      }
    }
    __syncthreads();
  }
  l += __shfl_xor(l, 32, 64);
  const float inv = 1.f / l;
  if (q0 + li < p.T) {
    bf16_t* orow = p.out + ((int64_t)n * p.T + q0 + li) * p.C + v0;
#pragma unroll
    for (int dt = 0; dt < DV / 32; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        u32x2 v;
        v.x = pack_bf16x2(ot[dt][4 * g + 0] * inv, ot[dt][4 * g + 1] * inv);
        v.y = pack_bf16x2(ot[dt][4 * g + 2] * inv, ot[dt][4 * g + 3] * inv);
        *reinterpret_cast<u32x2*>(orow + dt * 32 + 8 * g + 4 * h) = v;
      }
    if (blockIdx.z == 0 && h == 0) p.lse2[(int64_t)n * p.T + q0 + li] = m + log2f(l);
  }
}

// delta[row] = sum_c dO[row][c] * O[row][c]   (one wave per row)
__global__ void flash_delta_kernel(const bf16_t* __restrict__ d_o, const bf16_t* __restrict__ o, int64_t rows, int C, float* __restrict__ delta) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  float s = 0.f;
  for (int c0 = lane * 8; c0 < C; c0 += 512) {
    const u32x4 a = *reinterpret_cast<const u32x4*>(d_o + row * C + c0);
    const u32x4 b = *reinterpret_cast<const u32x4*>(o + row * C + c0);
    s += bf16_lo(a.x) * bf16_lo(b.x) + bf16_hi(a.x) * bf16_hi(b.x) + bf16_lo(a.y) * bf16_lo(b.y) + bf16_hi(a.y) * bf16_hi(b.y)
       + bf16_lo(a.z) * bf16_lo(b.z) + bf16_hi(a.z) * bf16_hi(b.z) + bf16_lo(a.w) * bf16_lo(b.w) + bf16_hi(a.w) * bf16_hi(b.w);
  }
  s = wave_sum(s);
  if (lane == 0) delta[row] = s;
}

// ------------------------------------------------------------------------------------------------------------------------
// dQ: query stationary.  grid (ceil(T/128), N, D/DA); slice z produces dQ columns [z*DA, (z+1)*DA)
// ------------------------------------------------------------------------------------------------------------------------
template <int D, int DA, int KBT>
__global__ __launch_bounds__(256) void flash_dq_kernel(FlashP p) {
  constexpr int KSTR = D + 8;
  constexpr int STAGE = 2 * KBT * KSTR;                      // K tile then V tile
  extern __shared__ __attribute__((aligned(16))) bf16_t smem[];   // 2 stages
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, h = lane >> 5;
  const int n = blockIdx.y, a0 = blockIdx.z * DA;
  const int q0 = blockIdx.x * 128 + wave * 32;
  const int C3 = 3 * p.C;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<bf16_t*>(p.qkv + (int64_t)n * p.T * C3), 0, p.T * C3 * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rdo = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<bf16_t*>(p.d_o + (int64_t)n * p.T * p.C), 0, p.T * p.C * 2, 0x00020000);
  const float c = p.scale * LOG2E;
  const bool qok = q0 + li < p.T;

  bf16x8 qf[D / 16], dof[D / 16];
#pragma unroll
  for (int ks = 0; ks < D / 16; ++ks) {
    qf[ks] = frag_from_u32x4(__builtin_amdgcn_raw_buffer_load_b128(rs, qok ? (unsigned)(((q0 + li) * C3 + 16 * ks + 8 * h) * 2) : 0x7FFFFFF0u, 0, 0));
    dof[ks] = frag_from_u32x4(__builtin_amdgcn_raw_buffer_load_b128(rdo, qok ? (unsigned)(((q0 + li) * p.C + 16 * ks + 8 * h) * 2) : 0x7FFFFFF0u, 0, 0));
  }
  const float lse = qok ? p.lse2[(int64_t)n * p.T + q0 + li] : 0.f;
  const float dl = qok ? p.delta[(int64_t)n * p.T + q0 + li] : 0.f;
  f32x16 dq[DA / 32];
#pragma unroll
  for (int dt = 0; dt < DA / 32; ++dt)
#pragma unroll
    for (int i = 0; i < 16; ++i) dq[dt][i] = 0.f;

  TileStage<D, KSTR, 256, KBT> kst, vst;       // pipeline as in flash_fwd_kernel
  const int ntiles = (p.T + KBT - 1) / KBT;
  kst.fetch(rs, 0, p.T, C3, p.C, tid);
  vst.fetch(rs, 0, p.T, C3, 2 * p.C, tid);
  kst.store(smem, tid);
  vst.store(smem + KBT * KSTR, tid);
  if (ntiles > 1) {
    kst.fetch(rs, KBT, p.T, C3, p.C, tid);
    vst.fetch(rs, KBT, p.T, C3, 2 * p.C, tid);
  }
  __syncthreads();
  for (int it = 0; it < ntiles; ++it) {
    const bf16_t* Kt = smem + (it & 1) * STAGE;
    const bf16_t* Vt = Kt + KBT * KSTR;
    if (it + 1 < ntiles) {
      bf16_t* nx = smem + ((it + 1) & 1) * STAGE;
      kst.store(nx, tid);
      vst.store(nx + KBT * KSTR, tid);
      if (it + 2 < ntiles) {
        kst.fetch(rs, (it + 2) * KBT, p.T, C3, p.C, tid);
        vst.fetch(rs, (it + 2) * KBT, p.T, C3, 2 * p.C, tid);
      }
    }
#pragma unroll
    for (int hf = 0; hf < KBT / KB; ++hf) {
      const int k0 = it * KBT + hf * KB;
      if (k0 >= p.T) break;
      const bf16_t* Ks = Kt + hf * KB * KSTR;
      const bf16_t* Vs = Vt + hf * KB * KSTR;
      f32x16 st, dpt;
#pragma unroll
      for (int i = 0; i < 16; ++i) { st[i] = 0.f; dpt[i] = 0.f; }
      rows_times_frags2<D, KSTR>(Ks, Vs, li, h, qf, dof, st, dpt);   // S^T[key][q] = K Q^T, dP^T[key][q] = V dO^T
      const bool tail = k0 + KB > p.T;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        float pr = fast_exp2(fmaf(st[i], c, -lse));
        if (tail && k0 + acc_row32(i, h) >= p.T) pr = 0.f;
        st[i] = pr * (dpt[i] - dl) * p.scale;  // dS^T
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8 db = frag_from_acc(st, s);
#pragma unroll
        for (int dt = 0; dt < DA / 32; ++dt) dq[dt] = mfma_bf16(tr_frag<KSTR>(Ks, s, a0 + dt * 32, lane), db, dq[dt]);   // dQ^T += K^T dS^T
      }
    }
    __syncthreads();
  }
  if (qok) {
    bf16_t* row = p.out + ((int64_t)n * p.T + q0 + li) * C3 + a0;
#pragma unroll
    for (int dt = 0; dt < DA / 32; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        u32x2 v;
        v.x = pack_bf16x2(dq[dt][4 * g + 0], dq[dt][4 * g + 1]);
        v.y = pack_bf16x2(dq[dt][4 * g + 2], dq[dt][4 * g + 3]);
        *reinterpret_cast<u32x2*>(row + dt * 32 + 8 * g + 4 * h) = v;
      }
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// dK, dV: key stationary.  grid (ceil(T/128), N, D/DA); slice z produces columns [z*DA, (z+1)*DA) of dK and dV
// ------------------------------------------------------------------------------------------------------------------------
template <int D, int DA, int KBT>
__global__ __launch_bounds__(256) void flash_dkv_kernel(FlashP p) {
  constexpr int QSTR = D + 8;
  constexpr int STAGE = 2 * KBT * QSTR + 4 * KBT;            // Q tile, dO tile, then lse2[KBT] and delta[KBT] as f32
  extern __shared__ __attribute__((aligned(16))) bf16_t smem[];   // 2 stages
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, h = lane >> 5;
  const int n = blockIdx.y, a0 = blockIdx.z * DA;
  const int key0 = blockIdx.x * 128 + wave * 32;
  const int C3 = 3 * p.C;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<bf16_t*>(p.qkv + (int64_t)n * p.T * C3), 0, p.T * C3 * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rdo = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<bf16_t*>(p.d_o + (int64_t)n * p.T * p.C), 0, p.T * p.C * 2, 0x00020000);
  const float c = p.scale * LOG2E;
  const bool kok = key0 + li < p.T;

  bf16x8 kf[D / 16], vf[D / 16];           // B fragments: lane (key, h) holds K[key][16ks + 8h ..], V[key][...]
#pragma unroll
  for (int ks = 0; ks < D / 16; ++ks) {
    kf[ks] = frag_from_u32x4(__builtin_amdgcn_raw_buffer_load_b128(rs, kok ? (unsigned)(((key0 + li) * C3 + p.C + 16 * ks + 8 * h) * 2) : 0x7FFFFFF0u, 0, 0));
    vf[ks] = frag_from_u32x4(__builtin_amdgcn_raw_buffer_load_b128(rs, kok ? (unsigned)(((key0 + li) * C3 + 2 * p.C + 16 * ks + 8 * h) * 2) : 0x7FFFFFF0u, 0, 0));
  }
  f32x16 dk[DA / 32], dv[DA / 32];
#pragma unroll
  for (int dt = 0; dt < DA / 32; ++dt)
#pragma unroll
    for (int i = 0; i < 16; ++i) { dk[dt][i] = 0.f; dv[dt][i] = 0.f; }

  TileStage<D, QSTR, 256, KBT> qst, ost;
  float lse_n = 0.f, dl_n = 0.f;
  auto fetch_rows = [&](int r0) {
    qst.fetch(rs, r0, p.T, C3, 0, tid);
    ost.fetch(rdo, r0, p.T, p.C, 0, tid);
    if (tid < KBT) {
      const bool ok = r0 + tid < p.T;
      lse_n = ok ? p.lse2[(int64_t)n * p.T + r0 + tid] : 0.f;
      dl_n = ok ? p.delta[(int64_t)n * p.T + r0 + tid] : 0.f;
    }
  };
  auto store_rows = [&](bf16_t* stage) {
    qst.store(stage, tid);
    ost.store(stage + KBT * QSTR, tid);
    if (tid < KBT) {
      float* rc = reinterpret_cast<float*>(stage + 2 * KBT * QSTR);
      rc[tid] = lse_n; rc[KBT + tid] = dl_n;
    }
  };
  const int ntiles = (p.T + KBT - 1) / KBT;      // pipeline as in flash_fwd_kernel
  fetch_rows(0);
  store_rows(smem);
  if (ntiles > 1) fetch_rows(KBT);
  __syncthreads();
  for (int it = 0; it < ntiles; ++it) {
    const bf16_t* Qt = smem + (it & 1) * STAGE;
    const bf16_t* Ot = Qt + KBT * QSTR;
    const float* rowc = reinterpret_cast<const float*>(Qt + 2 * KBT * QSTR);
    if (it + 1 < ntiles) {
      store_rows(smem + ((it + 1) & 1) * STAGE);
      if (it + 2 < ntiles) fetch_rows((it + 2) * KBT);
    }
#pragma unroll
    for (int hf = 0; hf < KBT / KB; ++hf) {
      const int r0 = it * KBT + hf * KB;
      if (r0 >= p.T) break;
      const bf16_t* Qs = Qt + hf * KB * QSTR;
      const bf16_t* Os = Ot + hf * KB * QSTR;
      // S[q][key] = Q K^T, dP[q][key] = dO V^T  (query row in the register, key on the lane)
      f32x16 sa, dpa;
#pragma unroll
      for (int i = 0; i < 16; ++i) { sa[i] = 0.f; dpa[i] = 0.f; }
      rows_times_frags2<D, QSTR>(Qs, Os, li, h, kf, vf, sa, dpa);
      const bool tail = r0 + KB > p.T;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int r = acc_row32(i, h);
        float pr = fast_exp2(fmaf(sa[i], c, -rowc[hf * KB + r]));
        if (!kok || (tail && r0 + r >= p.T)) pr = 0.f;
        sa[i] = pr;                                                 // P[q][key]
        dpa[i] = pr * (dpa[i] - rowc[KBT + hf * KB + r]) * p.scale;  // dS[q][key]
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8 pb = frag_from_acc(sa, s), db = frag_from_acc(dpa, s);
#pragma unroll
        for (int dt = 0; dt < DA / 32; ++dt) {
          dv[dt] = mfma_bf16(tr_frag<QSTR>(Os, s, a0 + dt * 32, lane), pb, dv[dt]);   // dV^T[d][key] += dO^T[d][q] P[q][key]
          dk[dt] = mfma_bf16(tr_frag<QSTR>(Qs, s, a0 + dt * 32, lane), db, dk[dt]);   // dK^T[d][key] += Q^T[d][q] dS[q][key]
        }
      }
    }
    __syncthreads();
  }
  if (kok) {
    bf16_t* row = p.out + ((int64_t)n * p.T + key0 + li) * C3 + a0;
#pragma unroll
    for (int dt = 0; dt < DA / 32; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        u32x2 a, b;
        a.x = pack_bf16x2(dk[dt][4 * g + 0], dk[dt][4 * g + 1]); a.y = pack_bf16x2(dk[dt][4 * g + 2], dk[dt][4 * g + 3]);
        b.x = pack_bf16x2(dv[dt][4 * g + 0], dv[dt][4 * g + 1]); b.y = pack_bf16x2(dv[dt][4 * g + 2], dv[dt][4 * g + 3]);
        *reinterpret_cast<u32x2*>(row + p.C + dt * 32 + 8 * g + 4 * h) = a;
        *reinterpret_cast<u32x2*>(row + 2 * p.C + dt * 32 + 8 * g + 4 * h) = b;
      }
  }
}

template <typename K>
void launch_dyn(K kernel, dim3 grid, int lds_bytes, hipStream_t st, const FlashP& p) {
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  hipLaunchKernelGGL(kernel, grid, dim3(256), lds_bytes, st, p);
}
constexpr int fwd_lds(int D, int DV, int KBT) { return 2 * (KBT * (D + 8) + KBT * (DV + 32)) * 2; }
constexpr int dq_lds(int D, int KBT) { return 2 * (2 * KBT * (D + 8)) * 2; }
constexpr int dkv_lds(int D, int KBT) { return 2 * (2 * KBT * (D + 8) + 4 * KBT) * 2; }

bool shape_ok(int N, int T, int C) {
  return N > 0 && T > 0 && (C == 64 || C == 128 || C == 256 || C == 512) && N <= 65535 && (int64_t)T * 3 * C * 2 < 0x7FFFFFF0ll;
}

}  // namespace

extern "C" {

int odvae_flash_attn_supported(int N, int T, int C) { return shape_ok(N, T, C) ? 1 : 0; }

// o = softmax(q k^T * scale) v, lse2 = log2 sum exp(scores * scale).  qkv bf16 [N][T][3C], o bf16 [N][T][C], lse2 f32 [N][T].
int odvae_flash_attn_fwd_bf16(const void* qkv, int N, int T, int C, float scale, void* o, float* lse2, void* stream) {
  ODVAE_CHECK_ARG(qkv && o && lse2, "flash_attn_fwd: null operand");
  ODVAE_CHECK_ARG(shape_ok(N, T, C), "flash_attn_fwd: unsupported shape N=%d T=%d C=%d (C in 64/128/256/512)", N, T, C);
  ODVAE_CHECK_ARG(((uintptr_t)qkv & 15) == 0 && ((uintptr_t)o & 15) == 0, "flash_attn_fwd: misaligned operand");
  FlashP p{};
  p.qkv = static_cast<const bf16_t*>(qkv); p.out = static_cast<bf16_t*>(o); p.lse2 = lse2; p.N = N; p.T = T; p.C = C; p.scale = scale;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int qb = ceil_div(T, 128);
  switch (C) {
    case 64:  launch_dyn(flash_fwd_kernel<64, 64, 64>, dim3(qb, N, 1), fwd_lds(64, 64, 64), st, p); break;
    case 128: launch_dyn(flash_fwd_kernel<128, 128, 64>, dim3(qb, N, 1), fwd_lds(128, 128, 64), st, p); break;
    case 256: launch_dyn(flash_fwd_kernel<256, 256, 32>, dim3(qb, N, 1), fwd_lds(256, 256, 32), st, p); break;
    default:  launch_dyn(flash_fwd_kernel<512, 128, 32>, dim3(qb, N, 4), fwd_lds(512, 128, 32), st, p); break;
  }
  ODVAE_LAUNCH_CHECK("flash_attn_fwd");
  return ODVAE_OK;
}

// dqkv [N][T][3C] (dq | dk | dv) from d_o, the forward's o and lse2; delta_ws: f32 [N*T] scratch.
int odvae_flash_attn_bwd_bf16(const void* qkv, const void* o, const void* d_o, const float* lse2, int N, int T, int C, float scale,
                              void* dqkv, float* delta_ws, void* stream) {
  ODVAE_CHECK_ARG(qkv && o && d_o && lse2 && dqkv && delta_ws, "flash_attn_bwd: null operand");
  ODVAE_CHECK_ARG(shape_ok(N, T, C), "flash_attn_bwd: unsupported shape N=%d T=%d C=%d", N, T, C);
  ODVAE_CHECK_ARG(((uintptr_t)qkv & 15) == 0 && ((uintptr_t)o & 15) == 0 && ((uintptr_t)d_o & 15) == 0 && ((uintptr_t)dqkv & 15) == 0,
                  "flash_attn_bwd: misaligned operand");
  FlashP p{};
  p.qkv = static_cast<const bf16_t*>(qkv); p.o = static_cast<const bf16_t*>(o); p.d_o = static_cast<const bf16_t*>(d_o);
  p.out = static_cast<bf16_t*>(dqkv); p.lse2 = const_cast<float*>(lse2); p.delta = delta_ws; p.N = N; p.T = T; p.C = C; p.scale = scale;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int64_t rows = (int64_t)N * T;
  hipLaunchKernelGGL(flash_delta_kernel, dim3((unsigned)ceil_div64(rows, 4)), dim3(256), 0, st, p.d_o, p.o, rows, C, delta_ws);
  ODVAE_LAUNCH_CHECK("flash_attn delta");
  const int qb = ceil_div(T, 128);
  switch (C) {
    case 64:
      launch_dyn(flash_dq_kernel<64, 64, 64>, dim3(qb, N, 1), dq_lds(64, 64), st, p);
      launch_dyn(flash_dkv_kernel<64, 64, 64>, dim3(qb, N, 1), dkv_lds(64, 64), st, p); break;
    case 128:
      launch_dyn(flash_dq_kernel<128, 128, 64>, dim3(qb, N, 1), dq_lds(128, 64), st, p);
      launch_dyn(flash_dkv_kernel<128, 128, 64>, dim3(qb, N, 1), dkv_lds(128, 64), st, p); break;
    case 256:
      launch_dyn(flash_dq_kernel<256, 256, 64>, dim3(qb, N, 1), dq_lds(256, 64), st, p);
      launch_dyn(flash_dkv_kernel<256, 128, 32>, dim3(qb, N, 2), dkv_lds(256, 32), st, p); break;
    default:
      launch_dyn(flash_dq_kernel<512, 128, 32>, dim3(qb, N, 4), dq_lds(512, 32), st, p);
      launch_dyn(flash_dkv_kernel<512, 64, 32>, dim3(qb, N, 8), dkv_lds(512, 32), st, p); break;
  }
  ODVAE_LAUNCH_CHECK("flash_attn_bwd");
  return ODVAE_OK;
}

}  // extern "C"
