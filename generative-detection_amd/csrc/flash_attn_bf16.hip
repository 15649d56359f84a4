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


// ------------------------------------------------------------------------------------------------------------------------
// Backward, second generation (D = 128 / 256): 512-thread blocks = 4 PAIRS of waves, <= 256 registers per wave.
//
// Why.  The kernels above keep a whole 32-row problem per wave: at D = 256 that is 330-430 registers, one wave per SIMD, hipcc
// parks part of it in the accumulator file and moves it back for every use (360 v_accvgpr moves per 96 MFMAs in flash_dq, spills in
// two instantiations), the dK/dV kernel only fits with the head dimension split over two blocks that BOTH recompute S and dP (9 matrix
// products per tile pair instead of 7), and the +8-element row padding that makes the row reads conflict-free leaves the transposed
// reads of the same tile 2.3-way conflicted (37-43 % of the LDS cycles, profiles/r02_flash_attn_pmc.txt).
//
// What.  Two waves share one 32-row problem and split it by ROLE, so that each half fits the 256 architectural registers (VGPR-form
// MFMAs, no accumulator-file traffic) and two waves live on every SIMD -- one wave's softmax arithmetic runs under its partner's
// MFMAs:
//   dK/dV (key stationary, 32 keys per pair, 128 per block):
//     wave A: S = Q K^T (K fragments in registers) -> P = exp2(c S - lse2) -> hands P to B through LDS -> dV^T += dO^T P   (all of D)
//     wave B: dP = dO V^T (V fragments in registers) -> dS = P (dP - delta) scale                      -> dK^T += Q^T dS   (all of D)
//     4 products per tile, none repeated; 32 MFMAs per wave and tile.
//   dQ (query stationary, 32 queries per pair): producer / consumer, the consumer one tile behind
//     wave A: S^T = K Q^T, dP^T = V dO^T (Q, dO fragments in registers) -> dS^T as packed bf16 fragments -> LDS
//     wave B: dQ^T += K^T dS^T (all of D).   3 products per tile.
//   One s_barrier per 32-row tile for everything: the pair hand-off (double-buffered), the landing of the tile two ahead, the release
//   of the stage two behind.
// Tiles arrive by LDS-DMA (buffer_load ... lds, 1 KiB per wave-instruction, no staging registers, rows past T read as zeros) into a
// ring of three stages, as unpadded rows of 2 D bytes whose 16-byte chunks are XOR-swizzled,
//     chunk' = chunk ^ (((row & 3) << 2) | ((row >> 2) & 3)),
// applied on the per-lane SOURCE address of the DMA (the LDS image of a DMA piece is lane-linear) and on every read: the
// ds_read_b128 row reads (16 different rows per lane group -> 16 different chunk positions) and the ds_read_b64_tr_b16 transposed
// reads (4 rows x 4 chunks per 32-lane half -> 16 different positions) of the SAME image are both conflict-free.
// Blocks of one image share an XCD (one L2 streams its q / k / v / dO rows once).
// ------------------------------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void* lds_ptr_t;
// wait until at most N of this wave's vector-memory operations are outstanding (N = the pieces of the newest fetch batch: a batch is
// given two periods to land -- it is issued for the tile after next -- and only the batch before it has to be complete at a barrier)
template <int N> __device__ __forceinline__ void wait_vm_all_but() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
#ifndef ODVAE_FLASH_PAIR_PRIO
#define ODVAE_FLASH_PAIR_PRIO 1
#endif
__device__ __forceinline__ bool flash_pair_prio() { return ODVAE_FLASH_PAIR_PRIO != 0; }
// -DODVAE_FLASH_STAMPS: s_memtime stamps in the dK/dV pair kernel (diagnostic build; tools/flash_stamps.py reads them with
// odvae_flash_debug_stamps).  Per role (A, B): cycles summed over the periods of block 0, pair 0, of the sections
//   0 tile fetch issue | 1 ring fill + (B) dS arithmetic | 2 second product (16 MFMAs) | 3 first product (16 MFMAs) | 4 (A) probabilities |
//   5 vmcnt wait | 6 barrier
#ifdef ODVAE_FLASH_STAMPS
__device__ unsigned long long g_flash_stamps[2][8];
#define FSTAMP(k) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); sacc_[k] += now_ - last_; last_ = now_; } while (0)   /* scalar registers only: no memory traffic inside the loop */
#else
#define FSTAMP(k) do { } while (0)
#endif
// timing-only ablation builds (tools/ab_build.py ... -DODVAE_FLASH_ABL=mask; results are WRONG): 1 = no tile fetch inside the loop,
// 2 = no barrier inside the loop, 4 = no softmax / dS arithmetic, 8 = no second product, 16 = no first product
#ifndef ODVAE_FLASH_ABL
#define ODVAE_FLASH_ABL 0
#endif

template <int D>
struct PairGeom {
  static constexpr int ROWB = 2 * D;              // bytes per tile row
  static constexpr int CPR = D / 8;               // 16-byte chunks per row (16 or 32: the swizzle permutes the low four bits)
  static constexpr int TILEB = 32 * ROWB;         // one 32-row tile
  static constexpr int PIECES = TILEB / 1024;     // LDS-DMA wave-instructions per tile (8 or 16)
  static constexpr int STAGEB = 2 * TILEB + 512;  // two tiles + two 256-byte row-constant slots
  static constexpr int BATCH = 2 * (PIECES / 8);  // LDS-DMA pieces a wave issues per tile pair
  static constexpr int NSTAGE = 3;                // dK/dV ring; the producer / consumer kernels (dQ, forward) run four stages
  static constexpr int NSTAGE_PC = 4;
};

__device__ __forceinline__ int swz16(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

// One LDS-DMA piece (1 KiB = RPP rows of 2 D bytes) of the [32][D] image of rows row0.. / columns col0.. of a [T][ld] matrix: piece j holds
// tile rows j RPP .. j RPP + RPP - 1; lane (r_in = lane / CPR, chp = lane % CPR) fills chunk position chp of row j RPP + r_in with source
// chunk chp ^ swz16(row).  The row part of the source offset is scalar arithmetic; per lane it is a shift, an OR, an XOR and an add
// (issuing a piece used to carry fourteen vector instructions, a 32-bit vector multiply among them: 160-180 cycles per piece, a fifth of
// a dK/dV period -- in-kernel stamps, tools/flash_stamps.py).  full = the tile lies inside the matrix (no per-lane range check).
template <int D>
__device__ __forceinline__ void dma_piece(i32x4_t rsrc, unsigned lds_tile, int j, int row0, int T, int ld, int col0, bool full, int lane) {
  using G = PairGeom<D>;
  constexpr int RPP = 64 / G::CPR;                        // rows per piece: 2 (D = 256) or 4 (D = 128)
  const unsigned r_in = (unsigned)lane / G::CPR, chp = (unsigned)lane % G::CPR;
  // swz16(j RPP + r_in): RPP = 2: ((2 j & 2) | r_in) << 2 | (j >> 1) & 3;  RPP = 4: r_in << 2 | j & 3
  const unsigned s0 = RPP == 2 ? ((((unsigned)j << 1) & 2u) << 2) | (((unsigned)j >> 1) & 3u) : ((unsigned)j & 3u);
  const unsigned ch = chp ^ (s0 | (r_in << 2));
  const unsigned rowbase = (unsigned)(((row0 + j * RPP) * ld + col0) * 2);          // scalar
  unsigned voff = rowbase + r_in * (unsigned)(ld * 2) + (ch << 4);
  if (!full) voff = row0 + j * RPP + (int)r_in < T ? voff : 0x7FFFFFF0u;
  lds_dma16(rsrc, (unsigned)__builtin_amdgcn_readfirstlane((int)(lds_tile + 1024u * j)), voff);
}
// this wave's share of one tile under the even split: pieces wave, wave + 8 (, ...)
template <int D>
__device__ __forceinline__ void dma_tile(i32x4_t rsrc, unsigned lds_tile, int row0, int T, int ld, int col0, int wave, int lane) {
  using G = PairGeom<D>;
  const bool full = row0 + 32 <= T;
#pragma unroll
  for (int k = 0; k < G::PIECES / 8; ++k) dma_piece<D>(rsrc, lds_tile, wave + 8 * k, row0, T, ld, col0, full, lane);
}
// 32 per-row f32 constants (lse2 / delta of rows row0..) into a 256-byte slot: lanes 32..63 write zeros behind them
__device__ __forceinline__ void dma_rowconst(i32x4_t rsrc, unsigned lds_slot, int row0, int T, int lane) {
  const unsigned voff = (lane < 32 && row0 + lane < T) ? (unsigned)((row0 + lane) * 4) : 0x7FFFFFF0u;
  lds_dma4(rsrc, (unsigned)__builtin_amdgcn_readfirstlane((int)lds_slot), voff);
}

// Row reads (A operand of k-step ks: lane (li, h) takes chunk 2 ks + h of row li):  address = rowv ^ (32 ks),
//   rowv = tile + li * ROWB + 16 * (swz16(li) ^ h)            (chunk (2 ks + h) ^ swz = (2 ks) ^ (swz ^ h))
__device__ __forceinline__ unsigned row_lane_off(int li, int h, int rowb) { return (unsigned)(li * rowb + 16 * (swz16(li) ^ h)); }
__device__ __forceinline__ bf16x8 row_frag(unsigned rowv, int ks) { return frag_from_u32x4(lds_ld128(rowv ^ (unsigned)(32 * ks))); }

// Transposed reads (rows of the fragment = columns 32 dt .. of the tile, k = tile rows 16 s .. 16 s + 15 in accumulator order): lane
// (g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3) reads 8 bytes of row r = 16 s + 4 (g >> 1) + q and of row r + 8, chunk
// 4 dt + j, j = 2 (g & 1) + (pp >> 1).  With swz16(r) = (q << 2) | hk and swz16(r + 8) = (q << 2) | (hk + 2) (hk = g >> 1):
//   address = (w ^ (64 dt)) + 8192 s,   w0 = tile + (4 hk + q) ROWB + 64 q + 16 (j ^ hk) + 8 (pp & 1),   w1 likewise for row + 8
struct TrLane { unsigned w0, w1; };
__device__ __forceinline__ TrLane tr_lane_off(int lane, int rowb) {
  const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3, hk = g >> 1, j = 2 * (g & 1) + (pp >> 1);
  TrLane t;
  t.w0 = (unsigned)((4 * hk + q) * rowb + 64 * q + 16 * (j ^ hk) + 8 * (pp & 1));
  t.w1 = (unsigned)((4 * hk + q + 8) * rowb + 64 * q + 16 * (j ^ (hk + 2)) + 8 * (pp & 1));
  return t;
}
template <int D>
__device__ __forceinline__ bf16x8 tr_frag_sw(unsigned tile, const TrLane& t, int s, int dt) {
  constexpr unsigned HALF = 16 * PairGeom<D>::ROWB;
  return frag_from_tr(lds_ld_tr(((tile + t.w0) ^ (unsigned)(64 * dt)) + HALF * s), lds_ld_tr(((tile + t.w1) ^ (unsigned)(64 * dt)) + HALF * s));
}

// blockIdx.x -> (image, 128-row block): the blocks of one image get ids that are equal mod 8, i.e. one XCD under round-robin placement
__device__ __forceinline__ void pair_block_coords(int N, int QB, int& n, int& qb) {
  const int id = blockIdx.x;
  if ((N & 7) == 0) {
    const int t = id >> 3;
    qb = t % QB;
    n = (t / QB) * 8 + (id & 7);
  } else {
    n = id / QB;
    qb = id % QB;
  }
}

// ---- dK / dV -------------------------------------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(512) void flash_dkv_pair_kernel(FlashP p) {
  using G = PairGeom<D>;
  constexpr unsigned XB = 2048;                            // one pair's P tile as two packed bf16 fragments: [2][64 lanes][16 bytes]
  constexpr int NST = 4;                                   // tile stages: a fetch is issued three tiles ahead and has two periods to land
  extern __shared__ __attribute__((aligned(1024))) char smem_c[];
  const unsigned smem = lds_addr_of(smem_c);
  const unsigned xbuf = smem + NST * G::STAGEB;           // [2 buffers][4 pairs][XB]
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int role = wave >> 2, pair = wave & 3;             // role 0 = A (S, P, dV), 1 = B (dP, dS, dK)
  // Waves w and w + 4 (a pair) share a SIMD.  A runs ahead: its MFMAs take the matrix pipe first, B's vector work (dS) hides under
  // them at the start of a period and B's MFMAs fill the pipe while A computes the next tile's probabilities at its end.
  if (ODVAE_FLASH_PAIR_PRIO == 1 && role == 0) __builtin_amdgcn_s_setprio(2);
  if (ODVAE_FLASH_PAIR_PRIO == 2 && role == 1) __builtin_amdgcn_s_setprio(2);
  const int QB = (p.T + 127) / 128;
  int n, kb;
  pair_block_coords(p.N, QB, n, kb);
  const int key0 = kb * 128 + pair * 32;
  const int C3 = 3 * p.C;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<bf16_t*>(p.qkv + (int64_t)n * p.T * C3), 0, p.T * C3 * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rdo = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<bf16_t*>(p.d_o + (int64_t)n * p.T * p.C), 0, p.T * p.C * 2, 0x00020000);
  const i32x4_t wqkv = rsrc_words(p.qkv + (int64_t)n * p.T * C3, (unsigned)(p.T * C3 * 2));      // the same buffers, for the asm LDS-DMA
  const i32x4_t wdo = rsrc_words(p.d_o + (int64_t)n * p.T * p.C, (unsigned)(p.T * p.C * 2));
  const i32x4_t wl = rsrc_words(p.lse2 + (int64_t)n * p.T, (unsigned)(p.T * 4));
  const i32x4_t wd = rsrc_words(p.delta + (int64_t)n * p.T, (unsigned)(p.T * 4));
  const float c = p.scale * LOG2E;
  const bool kok = key0 + li < p.T;
  const int ntiles = (p.T + 31) / 32;

  // Fetch duty (in-kernel stamps, tools/flash_stamps.py): a piece costs ~100-150 cycles of the issuing wave's time; issued by every wave
  // at the top of a period -- the first form -- all four SIMDs sat without an MFMA for ~800 of 3 600 cycles, and B (dS arithmetic in front
  // of its MFMAs, its MFMAs behind A's on the shared pipe) is the long pole while A waits a quarter of the period at the barrier.  So the A
  // waves alone fetch -- PIECES / 2 tile pieces each, waves 0 and 1 the two row-constant pieces as well -- and they do it BETWEEN their
  // MFMAs, one piece per four, where the cost lands on the wave with slack and B's MFMAs keep the pipe busy meanwhile.
  constexpr int NPA = G::PIECES / 2;
  auto fetch_piece = [&](int t, unsigned stage, int q) {     // A wave: its q-th piece (q < NPA; q == NPA: row constants) of tile t
    const unsigned st = smem + stage;
    int lane_v = lane;                                       // opaque copy: the source offsets are recomputed, not kept in registers
    asm volatile("" : "+v"(lane_v));
    const bool full = 32 * t + 32 <= p.T;
    if (q < NPA) {
      const int g = wave + 4 * q;                            // g in [0, 2 PIECES): Q tile pieces, then dO tile pieces
      if (g < G::PIECES) dma_piece<D>(wqkv, st, g, 32 * t, p.T, C3, 0, full, lane_v);
      else dma_piece<D>(wdo, st + G::TILEB, g - G::PIECES, 32 * t, p.T, p.C, 0, full, lane_v);
    } else {
      if (wave == 0) dma_rowconst(wl, st + 2 * G::TILEB, 32 * t, p.T, lane_v);
      if (wave == 1) dma_rowconst(wd, st + 2 * G::TILEB + 256, 32 * t, p.T, lane_v);
    }
  };
  auto issue = [&](int t, unsigned stage) {                 // prologue: a whole tile at once
    if (role == 0) {
#pragma unroll
      for (int q = 0; q <= NPA; ++q) fetch_piece(t, stage, q);
    }
  };
  issue(0, 0);
  if (ntiles > 1) issue(1, G::STAGEB);
  if (ntiles > 2) issue(2, 2 * G::STAGEB);

  // B fragments of this pair's 32 keys: K for role A, V for role B
  bf16x8 bf[D / 16];
  {
    const int col = (role == 0 ? p.C : 2 * p.C);
#pragma unroll
    for (int ks = 0; ks < D / 16; ++ks)
      bf[ks] = frag_from_u32x4(__builtin_amdgcn_raw_buffer_load_b128(rs, kok ? (unsigned)(((key0 + li) * C3 + col + 16 * ks + 8 * h) * 2) : 0x7FFFFFF0u, 0, 0));
  }
  f32x16 acc[D / 32];              // dV^T (A) or dK^T (B): [d][key]
#pragma unroll
  for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[dt][i] = 0.f;

  // lane parts of the LDS addresses; the first product reads Q (A) / dO (B) by rows, the second dO (A) / Q (B) transposed
  const unsigned rowl = smem + (role == 0 ? 0u : (unsigned)G::TILEB) + row_lane_off(li, h, G::ROWB);
  TrLane trl = tr_lane_off(lane, G::ROWB);
  trl.w0 += smem + (role == 0 ? (unsigned)G::TILEB : 0u);
  trl.w1 += smem + (role == 0 ? (unsigned)G::TILEB : 0u);
  const unsigned constl = smem + 2 * G::TILEB + 16 * h;    // + 256 for delta; row constants of register quad g at + 32 g
  const unsigned xl = xbuf + pair * XB + lane * 16;

  f32x16 s1;                       // A: S -> P;  B: dP, kept across the barrier
  bf16x8 pf[2];                    // A: P fragments of the tile whose dV product comes next period;  B: dS fragments
  // first product of the tile in `stage` alone (prologue only; inside the loop it is the second half of the MFMA stream below)
  auto first_product_only = [&](unsigned stage) {
    const unsigned rv = rowl + stage;
#pragma unroll
    for (int i = 0; i < 16; ++i) s1[i] = 0.f;
#pragma unroll
    for (int ks = 0; ks < D / 16; ++ks) s1 = mfma_bf16(row_frag(rv, ks), bf[ks], s1);
  };
  // A: P = exp2(c S - lse2) of the tile in `stage`, handed to B through exchange buffer `xsel`, and its fragments for the dV product
  auto probabilities = [&](unsigned stage, unsigned xsel) {
    if (ODVAE_FLASH_ABL & 4) { pf[0] = frag_from_acc(s1, 0); pf[1] = frag_from_acc(s1, 1); return; }
#pragma unroll
    for (int gp = 0; gp < 2; ++gp) {     // two row-constant reads in flight at a time (two LDS latencies per tile, not four; four do not fit the registers)
      const f32x4 la = lds_ld128f(constl + stage + 64 * gp), lb = lds_ld128f(constl + stage + 64 * gp + 32);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        s1[8 * gp + e] = fast_exp2(fmaf(s1[8 * gp + e], c, -la[e]));
        s1[8 * gp + 4 + e] = fast_exp2(fmaf(s1[8 * gp + 4 + e], c, -lb[e]));
      }
    }
    pf[0] = frag_from_acc(s1, 0);
    pf[1] = frag_from_acc(s1, 1);
    // B takes the bf16 fragments A multiplies with (its dS = P (dP - delta) scale is rounded to bf16 once more anyway)
    lds_st128(xl + xsel, __builtin_bit_cast(u32x4, pf[0]));
    lds_st128(xl + xsel + 1024, __builtin_bit_cast(u32x4, pf[1]));
  };

  unsigned cur = 0, nxt = G::STAGEB, nx2 = 2 * G::STAGEB, fre = 3 * G::STAGEB;    // stages of tiles j, j + 1, j + 2, j + 3 (= the one tile j - 1 leaves)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  first_product_only(0);
  if (role == 0) probabilities(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // One period = one MFMA stream per wave: NB second-product MFMAs of tile j (column block dt = i / 2, k-step i & 1), then NA
  // first-product MFMAs of tile j + 1.  Their LDS fragments travel in a ring RING MFMAs ahead, across the seam between the two products:
  // a wave waits out an LDS latency once per period, not once per product.  In the last period the first product runs on a stale
  // stage and its result is dropped.
  constexpr int NB = D / 16, NA = D / 16, RING = D >= 256 ? 4 : 6;   // D = 256 sits at 256 registers with four fragments in flight
#ifdef ODVAE_FLASH_STAMPS
  const bool stamp_on = blockIdx.x == 0 && pair == 0;
  unsigned long long sacc_[7] = {0, 0, 0, 0, 0, 0, 0};
  unsigned long long last_ = __builtin_amdgcn_s_memtime();
#endif
  for (int j = 0; j < ntiles; ++j) {
    const bool fetching = role == 0 && j + 3 < ntiles && !(ODVAE_FLASH_ABL & 1);   // into the stage tile j - 1 left (free since the barrier above)
    FSTAMP(0);
    const unsigned xsel = (j & 1) * 4 * XB;
    const unsigned rv = rowl + nxt;
    bf16x8 ring[RING];
    auto fetch = [&](int i) -> bf16x8 { return i < NB ? tr_frag_sw<D>(cur, trl, i & 1, i >> 1) : row_frag(rv, i - NB); };
#pragma unroll
    for (int i = 0; i < RING; ++i) ring[i] = fetch(i);
    if (role == 1 && !(ODVAE_FLASH_ABL & 4)) {   // dS = P (dP - delta) scale from A's probabilities (the ring's first reads are in flight)
#pragma unroll
      for (int sfr = 0; sfr < 2; ++sfr) {      // per fragment: its eight probabilities (one read) and the two row-constant quads, in flight together
        // P in fragment order: dword k of fragment s holds P of registers 8 s + 2 k (low half) and 8 s + 2 k + 1 (high half)
        const u32x4 pw = lds_ld128(xl + xsel + 1024 * sfr);
        const f32x4 da = lds_ld128f(constl + cur + 256 + 64 * sfr), db = lds_ld128f(constl + cur + 256 + 64 * sfr + 32);
        const float pv[8] = {bf16_lo(pw.x), bf16_hi(pw.x), bf16_lo(pw.y), bf16_hi(pw.y), bf16_lo(pw.z), bf16_hi(pw.z), bf16_lo(pw.w), bf16_hi(pw.w)};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          s1[8 * sfr + e] = pv[e] * fmaf(s1[8 * sfr + e], p.scale, -da[e] * p.scale);
          s1[8 * sfr + 4 + e] = pv[4 + e] * fmaf(s1[8 * sfr + 4 + e], p.scale, -db[e] * p.scale);
        }
      }
      pf[0] = frag_from_acc(s1, 0);
      pf[1] = frag_from_acc(s1, 1);
    }
    FSTAMP(1);
#pragma unroll
    for (int i = 0; i < NB + NA; ++i) {
      __builtin_amdgcn_sched_barrier(0);
#ifdef ODVAE_FLASH_STAMPS
      if (i == NB) FSTAMP(2);
#endif
      if (i < NB) {
        if (!(ODVAE_FLASH_ABL & 8)) acc[i >> 1] = mfma_bf16(ring[i % RING], pf[i & 1], acc[i >> 1]);
      } else {
        if (i == NB) {
#pragma unroll
          for (int r = 0; r < 16; ++r) s1[r] = 0.f;
        }
        if (!(ODVAE_FLASH_ABL & 16)) s1 = mfma_bf16(ring[i % RING], bf[i - NB], s1);
      }
      if (i + RING < NB + NA) ring[i % RING] = fetch(i + RING);
      if ((i & 3) == 1 && (i >> 2) < NPA && fetching) fetch_piece(j + 3, fre, i >> 2);      // A: one fetch piece per four MFMAs
      if (i == 3 && fetching) fetch_piece(j + 3, fre, NPA);                                  // (waves 0, 1: the row constants)
      __builtin_amdgcn_sched_barrier(0);
    }
    FSTAMP(3);
    if (role == 0) probabilities(nxt, xsel ^ (4 * XB));
    FSTAMP(4);
    const unsigned t = cur; cur = nxt; nxt = nx2; nx2 = fre; fre = t;
    // tile j + 2 is read in the next period (first product) and was fetched in the previous one: everything but the batch just issued
    // has to have landed (waves 0 and 1 carry one row-constant piece more)
    if (!fetching) wait_vm_all_but<0>();          // (B waves have no vector-memory operations in the loop at all)
    else if (wave < 2) wait_vm_all_but<NPA + 1>();
    else wait_vm_all_but<NPA>();
    FSTAMP(5);
    if (!(ODVAE_FLASH_ABL & 2)) __syncthreads();
    FSTAMP(6);
  }
#ifdef ODVAE_FLASH_STAMPS
  if (stamp_on && lane == 0)
    for (int k = 0; k < 7; ++k) g_flash_stamps[role][k] += sacc_[k];
#endif
  if (kok) {
    bf16_t* row = p.out + ((int64_t)n * p.T + key0 + li) * C3 + (role == 0 ? 2 * p.C : p.C);
#pragma unroll
    for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        u32x2 a;
        a.x = pack_bf16x2(acc[dt][4 * g + 0], acc[dt][4 * g + 1]); a.y = pack_bf16x2(acc[dt][4 * g + 2], acc[dt][4 * g + 3]);
        *reinterpret_cast<u32x2*>(row + dt * 32 + 8 * g + 4 * h) = a;
      }
  }
}

// ---- dQ ------------------------------------------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(512) void flash_dq_pair_kernel(FlashP p) {
  using G = PairGeom<D>;
  constexpr unsigned XB = 2048;                            // one pair's dS^T tile as two packed bf16 fragments: [2][64 lanes][16 bytes]
  extern __shared__ __attribute__((aligned(1024))) char smem_c[];
  const unsigned smem = lds_addr_of(smem_c);
  const unsigned xbuf = smem + G::NSTAGE_PC * G::STAGEB;  // [2 buffers][4 pairs][XB]
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int role = wave >> 2, pair = wave & 3;             // role 0 = A (S^T, dP^T, dS^T), 1 = B (dQ^T)
  if (role == 0 && flash_pair_prio()) __builtin_amdgcn_s_setprio(2);   // the producer is the long pole: the consumer's MFMAs fill the gaps it leaves
  const int QB = (p.T + 127) / 128;
  int n, qb;
  pair_block_coords(p.N, QB, n, qb);
  const int q0 = qb * 128 + pair * 32;
  const int C3 = 3 * p.C;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<bf16_t*>(p.qkv + (int64_t)n * p.T * C3), 0, p.T * C3 * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rdo = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<bf16_t*>(p.d_o + (int64_t)n * p.T * p.C), 0, p.T * p.C * 2, 0x00020000);
  const float c = p.scale * LOG2E;
  const bool qok = q0 + li < p.T;
  const int ntiles = (p.T + 31) / 32;

  const i32x4_t wqkv = rsrc_words(p.qkv + (int64_t)n * p.T * C3, (unsigned)(p.T * C3 * 2));      // the same buffer, for the asm LDS-DMA
  auto issue = [&](int t, unsigned stage) {   // tile t (keys 32 t ..): K rows, V rows
    const unsigned st = smem + stage;
    dma_tile<D>(wqkv, st, 32 * t, p.T, C3, p.C, wave, lane);
    dma_tile<D>(wqkv, st + G::TILEB, 32 * t, p.T, C3, 2 * p.C, wave, lane);
  };
  issue(0, 0);
  if (ntiles > 1) issue(1, G::STAGEB);
  const unsigned xl = xbuf + pair * XB + lane * 16;
  // period t (0 .. ntiles): the producer multiplies tile t (stage cur), the consumer tile t - 1 (stage prv); tile t + 1 lands in nxt
  // four stages: tile t + 2 is fetched during period t (into the stage tile t - 2 left) and has two periods to land
  unsigned prv = 3 * G::STAGEB, cur = 0, nxt = G::STAGEB, nn = 2 * G::STAGEB;

  if (role == 0) {
    // ---- producer ----
    bf16x8 qf[D / 16], dof[D / 16];
#pragma unroll
    for (int ks = 0; ks < D / 16; ++ks) {
      qf[ks] = frag_from_u32x4(__builtin_amdgcn_raw_buffer_load_b128(rs, qok ? (unsigned)(((q0 + li) * C3 + 16 * ks + 8 * h) * 2) : 0x7FFFFFF0u, 0, 0));
      dof[ks] = frag_from_u32x4(__builtin_amdgcn_raw_buffer_load_b128(rdo, qok ? (unsigned)(((q0 + li) * p.C + 16 * ks + 8 * h) * 2) : 0x7FFFFFF0u, 0, 0));
    }
    const float lse = qok ? p.lse2[(int64_t)n * p.T + q0 + li] : 0.f;
    const float dls = (qok ? p.delta[(int64_t)n * p.T + q0 + li] : 0.f) * p.scale;
    const unsigned rowl = smem + row_lane_off(li, h, G::ROWB);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int t = 0; t <= ntiles; ++t) {
      const bool fetch = t + 2 < ntiles;
      if (fetch) issue(t + 2, nn);
      if (t < ntiles) {
        const unsigned kr = rowl + cur;
        f32x16 sa, da;
#pragma unroll
        for (int i = 0; i < 16; ++i) { sa[i] = 0.f; da[i] = 0.f; }
        {   // K / V row fragments RING MFMAs ahead (order pinned); the V tile sits TILEB behind the K tile: same XOR, immediate offset
          constexpr int RING = 8, NM = 2 * (D / 16);
          bf16x8 ring[RING];
          auto fetch = [&](int i) -> bf16x8 { return frag_from_u32x4(lds_ld128((kr ^ (unsigned)(32 * (i >> 1))) + ((i & 1) ? (unsigned)G::TILEB : 0u))); };
#pragma unroll
          for (int i = 0; i < RING; ++i) ring[i] = fetch(i);
#pragma unroll
          for (int i = 0; i < NM; ++i) {
            __builtin_amdgcn_sched_barrier(0);
            if (i & 1) da = mfma_bf16(ring[i % RING], dof[i >> 1], da);      // dP^T[key][q]
            else sa = mfma_bf16(ring[i % RING], qf[i >> 1], sa);             // S^T[key][q]
            if (i + RING < NM) ring[i % RING] = fetch(i + RING);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        const bool tail = 32 * t + 32 > p.T;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          float pr = fast_exp2(fmaf(sa[i], c, -lse));
          if (tail && 32 * t + acc_row32(i, h) >= p.T) pr = 0.f;
          sa[i] = pr * fmaf(da[i], p.scale, -dls);             // dS^T = P (dP - delta) scale
        }
        const unsigned xw = xl + (t & 1) * 4 * XB;
        lds_st128(xw, __builtin_bit_cast(u32x4, frag_from_acc(sa, 0)));
        lds_st128(xw + 1024, __builtin_bit_cast(u32x4, frag_from_acc(sa, 1)));
      }
      { const unsigned o = prv; prv = cur; cur = nxt; nxt = nn; nn = o; }
      if (fetch) wait_vm_all_but<G::BATCH>(); else wait_vm_all_but<0>();     // the batch just issued stays in flight
      __syncthreads();
    }
  } else {
    // ---- consumer ----
    f32x16 dq[D / 32];
#pragma unroll
    for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
      for (int i = 0; i < 16; ++i) dq[dt][i] = 0.f;
    TrLane trl = tr_lane_off(lane, G::ROWB);
    trl.w0 += smem;
    trl.w1 += smem;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int t = 0; t <= ntiles; ++t) {
      const bool fetch = t + 2 < ntiles;
      if (fetch) issue(t + 2, nn);
      if (t >= 1) {
        const unsigned xr = xl + ((t - 1) & 1) * 4 * XB;
        constexpr int RING = 8, NM = D / 16;      // dQ^T += K^T dS^T; transposed fragments RING MFMAs ahead
        bf16x8 ring[RING];
        auto fetch = [&](int i) -> bf16x8 { return tr_frag_sw<D>(prv, trl, i & 1, i >> 1); };
#pragma unroll
        for (int i = 0; i < RING; ++i) ring[i] = fetch(i);
        bf16x8 db[2];
        db[0] = frag_from_u32x4(lds_ld128(xr));
        db[1] = frag_from_u32x4(lds_ld128(xr + 1024));
#pragma unroll
        for (int i = 0; i < NM; ++i) {
          __builtin_amdgcn_sched_barrier(0);
          dq[i >> 1] = mfma_bf16(ring[i % RING], db[i & 1], dq[i >> 1]);
          if (i + RING < NM) ring[i % RING] = fetch(i + RING);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      { const unsigned o = prv; prv = cur; cur = nxt; nxt = nn; nn = o; }
      if (fetch) wait_vm_all_but<G::BATCH>(); else wait_vm_all_but<0>();     // the batch just issued stays in flight
      __syncthreads();
    }
    if (qok) {
      bf16_t* row = p.out + ((int64_t)n * p.T + q0 + li) * C3;
#pragma unroll
      for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          u32x2 v;
          v.x = pack_bf16x2(dq[dt][4 * g + 0], dq[dt][4 * g + 1]);
          v.y = pack_bf16x2(dq[dt][4 * g + 2], dq[dt][4 * g + 3]);
          *reinterpret_cast<u32x2*>(row + dt * 32 + 8 * g + 4 * h) = v;
        }
    }
  }
}

// ---- forward -----------------------------------------------------------------------------------------------------------------------
// Same skeleton as the dQ kernel.  Producer A: S^T = K Q^T (Q fragments in registers), online softmax with the deferred maximum
// (query on the lane: max / sum are in-register plus one lane^32 exchange), P^T as packed bf16 fragments + the row's rescale factor ->
// LDS.  Consumer B, one tile behind: O^T += V^T P^T (all of D, V^T by transposed reads), rescaling O only in the rare periods in
// which some row's reference maximum moved.
template <int D>
__global__ __launch_bounds__(512) void flash_fwd_pair_kernel(FlashP p) {
  using G = PairGeom<D>;
  constexpr unsigned XB = 2048 + 256;                      // one pair's P^T tile (two packed bf16 fragments) + 64 rescale factors
  extern __shared__ __attribute__((aligned(1024))) char smem_c[];
  const unsigned smem = lds_addr_of(smem_c);
  const unsigned xbuf = smem + G::NSTAGE_PC * G::STAGEB;  // [2 buffers][4 pairs][XB]
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int role = wave >> 2, pair = wave & 3;             // role 0 = A (S^T, softmax), 1 = B (O^T)
  if (role == 0 && flash_pair_prio()) __builtin_amdgcn_s_setprio(2);
  const int QB = (p.T + 127) / 128;
  int n, qb;
  pair_block_coords(p.N, QB, n, qb);
  const int q0 = qb * 128 + pair * 32;
  const int C3 = 3 * p.C;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<bf16_t*>(p.qkv + (int64_t)n * p.T * C3), 0, p.T * C3 * 2, 0x00020000);
  const float c = p.scale * LOG2E;
  const bool qok = q0 + li < p.T;
  const int ntiles = (p.T + 31) / 32;
  const i32x4_t wqkv = rsrc_words(p.qkv + (int64_t)n * p.T * C3, (unsigned)(p.T * C3 * 2));
  auto issue = [&](int t, unsigned stage) {   // tile t (keys 32 t ..): K rows, V rows
    const unsigned st = smem + stage;
    dma_tile<D>(wqkv, st, 32 * t, p.T, C3, p.C, wave, lane);
    dma_tile<D>(wqkv, st + G::TILEB, 32 * t, p.T, C3, 2 * p.C, wave, lane);
  };
  issue(0, 0);
  if (ntiles > 1) issue(1, G::STAGEB);
  const unsigned xl = xbuf + pair * XB + lane * 16;        // fragment slot of this lane
  const unsigned al = xbuf + pair * XB + 2048 + lane * 4;  // rescale-factor slot of this lane
  // four stages: tile t + 2 is fetched during period t (into the stage tile t - 2 left) and has two periods to land
  unsigned prv = 3 * G::STAGEB, cur = 0, nxt = G::STAGEB, nn = 2 * G::STAGEB;

  if (role == 0) {
    // ---- producer ----
    bf16x8 qf[D / 16];
#pragma unroll
    for (int ks = 0; ks < D / 16; ++ks)
      qf[ks] = frag_from_u32x4(__builtin_amdgcn_raw_buffer_load_b128(rs, qok ? (unsigned)(((q0 + li) * C3 + 16 * ks + 8 * h) * 2) : 0x7FFFFFF0u, 0, 0));
    const unsigned rowl = smem + row_lane_off(li, h, G::ROWB);
    float m = NEG_BIG, l = 0.f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int t = 0; t <= ntiles; ++t) {
      const bool fetch = t + 2 < ntiles;
      if (fetch) issue(t + 2, nn);
      if (t < ntiles) {
        const unsigned kr = rowl + cur;
        f32x16 st;
#pragma unroll
        for (int i = 0; i < 16; ++i) st[i] = 0.f;
        {
          constexpr int RING = 8, NM = D / 16;
          bf16x8 ring[RING];
#pragma unroll
          for (int i = 0; i < RING; ++i) ring[i] = row_frag(kr, i);
#pragma unroll
          for (int i = 0; i < NM; ++i) {
            __builtin_amdgcn_sched_barrier(0);
            st = mfma_bf16(ring[i % RING], qf[i], st);           // S^T[key][q]
            if (i + RING < NM) ring[i % RING] = row_frag(kr, i + RING);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        if (32 * t + 32 > p.T) {
#pragma unroll
          for (int i = 0; i < 16; ++i)
            if (32 * t + acc_row32(i, h) >= p.T) st[i] = NEG_BIG;
        }
        float mx = fmaxf(fmaxf(fmaxf(st[0], st[1]), fmaxf(st[2], st[3])), fmaxf(fmaxf(st[4], st[5]), fmaxf(st[6], st[7])));
        mx = fmaxf(mx, fmaxf(fmaxf(fmaxf(st[8], st[9]), fmaxf(st[10], st[11])), fmaxf(fmaxf(st[12], st[13]), fmaxf(st[14], st[15]))));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * c;
        float alpha = 1.f;
        if (__builtin_amdgcn_ballot_w64(mx > m + 8.f) != 0) {    // deferred maximum, as in flash_fwd_kernel
          const float m_new = fmaxf(m, mx);
          alpha = fast_exp2(m - m_new);
          l *= alpha;
          m = m_new;
        }
        float ps = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) { st[i] = fast_exp2(fmaf(st[i], c, -m)); ps += st[i]; }
        l += ps;
        const unsigned xs = (t & 1) * 4 * XB;
        lds_st128(xl + xs, __builtin_bit_cast(u32x4, frag_from_acc(st, 0)));
        lds_st128(xl + xs + 1024, __builtin_bit_cast(u32x4, frag_from_acc(st, 1)));
        *(__attribute__((address_space(3))) float*)(uintptr_t)(al + xs) = alpha;
      }
      { const unsigned o = prv; prv = cur; cur = nxt; nxt = nn; nn = o; }
      if (fetch) wait_vm_all_but<G::BATCH>(); else wait_vm_all_but<0>();     // the batch just issued stays in flight
      __syncthreads();
    }
    l += __shfl_xor(l, 32, 64);
    *(__attribute__((address_space(3))) float*)(uintptr_t)al = l;        // buffer 0 is free: its last reader finished before the barrier above
    __syncthreads();
    if (qok && h == 0) p.lse2[(int64_t)n * p.T + q0 + li] = m + log2f(l);
  } else {
    // ---- consumer ----
    f32x16 ot[D / 32];
#pragma unroll
    for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
      for (int i = 0; i < 16; ++i) ot[dt][i] = 0.f;
    TrLane trl = tr_lane_off(lane, G::ROWB);
    trl.w0 += smem + G::TILEB;          // the V tile of a stage
    trl.w1 += smem + G::TILEB;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int t = 0; t <= ntiles; ++t) {
      const bool fetch = t + 2 < ntiles;
      if (fetch) issue(t + 2, nn);
      if (t >= 1) {
        const unsigned xs = ((t - 1) & 1) * 4 * XB;
        constexpr int RING = 8, NM = D / 16;
        bf16x8 ring[RING];
        auto fetch = [&](int i) -> bf16x8 { return tr_frag_sw<D>(prv, trl, i & 1, i >> 1); };
#pragma unroll
        for (int i = 0; i < RING; ++i) ring[i] = fetch(i);
        bf16x8 pb[2];
        pb[0] = frag_from_u32x4(lds_ld128(xl + xs));
        pb[1] = frag_from_u32x4(lds_ld128(xl + xs + 1024));
        const float alpha = *(const __attribute__((address_space(3))) float*)(uintptr_t)(al + xs);
        if (__builtin_amdgcn_ballot_w64(alpha != 1.f) != 0) {
#pragma unroll
          for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
            for (int i = 0; i < 16; ++i) ot[dt][i] *= alpha;
        }
#pragma unroll
        for (int i = 0; i < NM; ++i) {
          __builtin_amdgcn_sched_barrier(0);
          ot[i >> 1] = mfma_bf16(ring[i % RING], pb[i & 1], ot[i >> 1]);   // O^T[d][q] += V^T[d][key] P^T[key][q]
          if (i + RING < NM) ring[i % RING] = fetch(i + RING);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      { const unsigned o = prv; prv = cur; cur = nxt; nxt = nn; nn = o; }
      if (fetch) wait_vm_all_but<G::BATCH>(); else wait_vm_all_but<0>();     // the batch just issued stays in flight
      __syncthreads();
    }
    __syncthreads();
    const float inv = 1.f / *(const __attribute__((address_space(3))) float*)(uintptr_t)al;
    if (qok) {
      bf16_t* orow = p.out + ((int64_t)n * p.T + q0 + li) * p.C;
#pragma unroll
      for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          u32x2 v;
          v.x = pack_bf16x2(ot[dt][4 * g + 0] * inv, ot[dt][4 * g + 1] * inv);
          v.y = pack_bf16x2(ot[dt][4 * g + 2] * inv, ot[dt][4 * g + 3] * inv);
          *reinterpret_cast<u32x2*>(orow + dt * 32 + 8 * g + 4 * h) = v;
        }
    }
  }
}

template <int D> constexpr int fwd_pair_lds() { return PairGeom<D>::NSTAGE_PC * PairGeom<D>::STAGEB + 2 * 4 * (2048 + 256); }
template <int D> constexpr int dkv_pair_lds() { return 4 * PairGeom<D>::STAGEB + 2 * 4 * 2048; }
template <int D> constexpr int dq_pair_lds() { return PairGeom<D>::NSTAGE_PC * PairGeom<D>::STAGEB + 2 * 4 * 2048; }

thread_local hipError_t g_flash_attr_error = hipSuccess;
thread_local int g_flash_attr_lds = 0;
template <typename K>
void launch_dyn(K kernel, dim3 grid, int lds_bytes, hipStream_t st, const FlashP& p, int threads = 256) {
  // a refused LDS size (the pair kernels ask for up to ~148 KB) must not surface as an anonymous launch failure, or not at all:
  // the error text names the size, and the launch is skipped so that ODVAE_LAUNCH_CHECK reports this call, not a later one
  const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  if (e != hipSuccess) {
    g_flash_attr_error = e;
    g_flash_attr_lds = lds_bytes;
    return;
  }
  hipLaunchKernelGGL(kernel, grid, dim3(threads), lds_bytes, st, p);
}
#define ODVAE_FLASH_LAUNCH_CHECK(name)                                                                                   \
  do {                                                                                                                 \
    if (g_flash_attr_error != hipSuccess) {                                                                            \
      odvae_set_error("%s: hipFuncSetAttribute(MaxDynamicSharedMemorySize = %d bytes) failed: %s", name, g_flash_attr_lds, \
                      hipGetErrorString(g_flash_attr_error));                                                          \
      g_flash_attr_error = hipSuccess;                                                                                 \
      return ODVAE_ERR_HIP;                                                                                            \
    }                                                                                                                  \
    ODVAE_LAUNCH_CHECK(name);                                                                                          \
  } while (0)
// ODVAE_FLASH_BWD_V1=1 keeps the first-generation kernels (one 32-row problem per wave; forward and backward) for in-process A/B runs
bool flash_bwd_v1() {
  static const bool v = [] { const char* e = getenv("ODVAE_FLASH_BWD_V1"); return e && e[0] == '1'; }();
  return v;
}
constexpr int fwd_lds(int D, int DV, int KBT) { return 2 * (KBT * (D + 8) + KBT * (DV + 32)) * 2; }
constexpr int dq_lds(int D, int KBT) { return 2 * (2 * KBT * (D + 8)) * 2; }
constexpr int dkv_lds(int D, int KBT) { return 2 * (2 * KBT * (D + 8) + 4 * KBT) * 2; }

bool shape_ok(int N, int T, int C) {
  return N > 0 && T > 0 && (C == 64 || C == 128 || C == 256 || C == 512) && N <= 65535 && (int64_t)T * 3 * C * 2 < 0x7FFFFFF0ll;
}

}  // namespace

#ifdef ODVAE_FLASH_STAMPS
extern "C" int odvae_flash_debug_stamps(unsigned long long* out16, int reset) {
  (void)hipDeviceSynchronize();
  if (out16) (void)hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_flash_stamps), sizeof(unsigned long long) * 16);
  if (reset) { unsigned long long z[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_flash_stamps), z, sizeof(z)); }
  return 0;
}
#endif

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
  if (!flash_bwd_v1() && (C == 128 || C == 256) && (int64_t)qb * N < 0x7FFFFFFF) {
    if (C == 256) launch_dyn(flash_fwd_pair_kernel<256>, dim3(qb * N), fwd_pair_lds<256>(), st, p, 512);
    else launch_dyn(flash_fwd_pair_kernel<128>, dim3(qb * N), fwd_pair_lds<128>(), st, p, 512);
    ODVAE_FLASH_LAUNCH_CHECK("flash_attn_fwd (pair kernel)");
    return ODVAE_OK;
  }
  switch (C) {
    case 64:  launch_dyn(flash_fwd_kernel<64, 64, 64>, dim3(qb, N, 1), fwd_lds(64, 64, 64), st, p); break;
    case 128: launch_dyn(flash_fwd_kernel<128, 128, 64>, dim3(qb, N, 1), fwd_lds(128, 128, 64), st, p); break;
    case 256: launch_dyn(flash_fwd_kernel<256, 256, 32>, dim3(qb, N, 1), fwd_lds(256, 256, 32), st, p); break;
    default:  launch_dyn(flash_fwd_kernel<512, 128, 32>, dim3(qb, N, 4), fwd_lds(512, 128, 32), st, p); break;
  }
  ODVAE_FLASH_LAUNCH_CHECK("flash_attn_fwd");
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
  if (!flash_bwd_v1() && (C == 128 || C == 256) && (int64_t)qb * N < 0x7FFFFFFF) {
    if (C == 256) {
      launch_dyn(flash_dq_pair_kernel<256>, dim3(qb * N), dq_pair_lds<256>(), st, p, 512);
      launch_dyn(flash_dkv_pair_kernel<256>, dim3(qb * N), dkv_pair_lds<256>(), st, p, 512);
    } else {
      launch_dyn(flash_dq_pair_kernel<128>, dim3(qb * N), dq_pair_lds<128>(), st, p, 512);
      launch_dyn(flash_dkv_pair_kernel<128>, dim3(qb * N), dkv_pair_lds<128>(), st, p, 512);
    }
    ODVAE_FLASH_LAUNCH_CHECK("flash_attn_bwd (pair kernels)");
    return ODVAE_OK;
  }
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
  ODVAE_FLASH_LAUNCH_CHECK("flash_attn_bwd");
  return ODVAE_OK;
}

}  // extern "C"
