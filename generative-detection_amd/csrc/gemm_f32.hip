// Exact-f32 matrix-core GEMM for gfx950 (v_mfma_f32_32x32x2_f32).
//
// Serves the "dense contraction" rows of the hot path: 1x1 convolutions
// (nin_shortcut, attention q/k/v/proj_out, quant convs: [UPSTREAM] ldm/modules/diffusionmodules/model.py
// ResnetBlock/AttnBlock; src/models/autoencoder.py:88-90) in NHWC, their dgrad/wgrad, and the
// batched products of single-head attention (AttnBlock.forward: bmm(q,k), bmm(v,w_)).
//
// C[b] = alpha * op(A[b]) * op(B[b]) (+ bias[col]) (+ residual[b]),  row-major everywhere.
//   transA = 0: A stored [M][K] (k contiguous)   transA = 1: A stored [K][M]
//   transB = 0: B stored [K][N] (n contiguous)   transB = 1: B stored [N][K]
//
// Block = 256 threads = 4 waves (2x2), block tile 128x128, wave tile 64x64 = 2x2 MFMA tiles,
// BK = 32.  Operand tiles go through LDS; k-contiguous tiles are kept [row][k] (+4 pad, so the
// 16 rows of a ds_read_b128 lane group land on 16 disjoint 4-bank slots) and read as b128 =
// four k-steps per read; row-contiguous tiles are kept [k][row] and read as b32 (lanes =
// consecutive rows, conflict-free).  Both reads use the same k order inside a group of 8:
// step j of lane half h is k = 8g + 4h + j, so A and B always agree.
// Split-K (grid.y) covers the long-K/small-MN products (wgrad: K = B*H*W).
#include "bf16_common.h"   // LDS in 32-bit addresses, LDS-DMA by inline asm
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int LDS_KC = BK + 4;    // [row][k] tile row stride (floats)
constexpr int LDS_RC = BM + 4;    // [k][row] tile row stride (floats)
constexpr int TILE_FLOATS = BM * LDS_KC;  // 4608 >= BK*LDS_RC = 4224
// LDS-DMA form (default): operand tiles go HBM/L2 -> LDS directly (`buffer_load ... lds`: no registers, no ds_write), unpadded,
// two stages of {A, B} = 4 x 16 KB.  A k-contiguous tile [128 rows][8 slots of 16 B] is XOR-swizzled -- LDS slot (row, q') holds
// k quad q = q' ^ ((row >> 1) & 7), chosen by the per-lane GLOBAL address, the LDS side of the DMA being linear in the lane -- so the
// 16 rows of a ds_read_b128 phase land on 16 different bank groups (even rows on banks 0-31, odd rows on 32-63, eight slots each).
// A row-contiguous tile [32 k][128 rows] is read as b32 by 32 consecutive rows: conflict-free as it is.
// Step width BKT = 32: 64 KB, two blocks per CU; BKT = 16: 32 KB, four blocks per CU (short-K products, where a block is mostly
// prologue and store burst and more blocks per CU cover them); rows of 4 slots then swizzle with (row >> 2) & 3.
template <int BKT> struct DmaGeom {
  static constexpr unsigned TILE_B = BM * BKT * 4;      // 16 384 / 8 192
  static constexpr unsigned STAGE_B = 2 * TILE_B;       // A then B
  static constexpr unsigned LDS_B = 2 * STAGE_B;        // 65 536 / 32 768
  static constexpr int SLOTS = BKT / 4;                 // 16-byte slots per row of a k-contiguous tile
  static constexpr int PER_THREAD = BM * SLOTS / 256;   // DMA instructions per thread and operand: 4 / 2
  static constexpr int ROW_SHIFT = BKT == 32 ? 3 : 2, SWZ_SHIFT = BKT == 32 ? 1 : 2, SWZ_MASK = SLOTS - 1;
};
__device__ __forceinline__ float lds_ld32(unsigned a) { return *(const __attribute__((address_space(3))) float*)(uintptr_t)a; }

struct GemmParams {
  const float* A; const float* B; float* C;
  const float* bias; const float* residual;
  float* partial;      // split-K slabs [split][batch][M][N] or nullptr
  int M, N, K;
  int lda, ldb, ldc;
  int64_t sA, sB, sC;
  float alpha;
  int splits, k_per_split;
  int tiles_m;
  // softmax-backward epilogue (EPI_SMB kernels only): C = alpha * emul .* (A B^T - rowsub[row]) (* rowmul[row]); emul has C's layout
  const float* rowsub; const float* emul; int64_t sRow;
  const float* rowmul;   // EPI_SMB: optional second row factor (1 / l_i when emul holds unnormalised exponentials), or null
  // EPI_EXPB: C = exp(alpha * (A B^T - rowsub[row])) -- attention scores leave the QK^T product as exponentials relative to a per-row
  //           upper bound of the scores instead of the row maximum: no separate softmax pass over the T x T tensor
  // EPI_ROWNORM (A k-contiguous, unsplit): l[row] = sum_k A[row][k] is accumulated beside the products, C = A B / l[row];
  //           rowout[row] = 1 / l[row] (written by the first column tile); *flag |= 1 where l is not a normal number >= 1e-30
  float* rowout; int* flag;
  const int* pred;       // gemm_f32_pred_kernel: nothing happens unless *pred != 0
};
constexpr int EPI_NONE = 0, EPI_SMB = 1, EPI_EXPB = 2, EPI_ROWNORM = 3;

// Operand tiles are fetched through a buffer descriptor that starts at the tile's origin (first row of the block, first k of
// the split): the per-lane byte offsets are computed once, a step only changes a scalar offset, and rows / k beyond the
// operand fall past num_records and read as zero -- no per-step vector address arithmetic or compares (on gfx950 every
// vector instruction takes its cycles from the f32 MFMA, profiles/r02_wino8_loop.md).
template <bool KC>
struct TileFetch {
  __amdgpu_buffer_rsrc_t rsrc;
  unsigned voff[4];
  unsigned step_bytes;     // scalar advance per BK
  int kvalid;              // KC only: k values of this split that exist (tail masking when not a multiple of BK)

  // G: operand of this batch; rows [row0, nrows) x k [kbeg, kend) is what the block may touch
  __device__ __forceinline__ void init(const float* G, int ld, int row0, int nrows, int kbeg, int kend) {
    const int tid = threadIdx.x;
    const unsigned OOB = 0x7FFFFFF0u;
    kvalid = kend - kbeg;
    if (KC) {            // G[row][k]: rows past nrows start beyond the last valid byte
      const int rows = min(BM, nrows - row0);
      const int64_t bytes = ((int64_t)(rows - 1) * ld + kvalid) * 4;
      rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(G) + (int64_t)row0 * ld + kbeg, 0, (int)(bytes < 0x7FFFFFF0ll ? bytes : 0x7FFFFFF0ll), 0x00020000);
      step_bytes = BK * 4u;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int f = tid + 256 * i;
        voff[i] = (unsigned)(((f >> 3) * ld + 4 * (f & 7)) * 4);
      }
    } else {             // G[k][row]: k past kend starts beyond the last valid byte; rows are masked per lane
      const int64_t bytes = ((int64_t)(kvalid - 1) * ld + (nrows - row0)) * 4;
      rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(G) + (int64_t)kbeg * ld + row0, 0, (int)(bytes < 0x7FFFFFF0ll ? bytes : 0x7FFFFFF0ll), 0x00020000);
      step_bytes = (unsigned)(BK * ld) * 4u;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int f = tid + 256 * i;
        voff[i] = (row0 + 4 * (f & 31) < nrows) ? (unsigned)(((f >> 5) * ld + 4 * (f & 31)) * 4) : OOB;
      }
    }
  }
  // ---- LDS-DMA form ----
  i32x4_t words;           // the same descriptor as raw dwords (inline asm operand)
  template <int BKT>
  __device__ __forceinline__ void init_dma(const float* G, int ld, int row0, int nrows, int kbeg, int kend) {
    typedef DmaGeom<BKT> D;
    const int tid = threadIdx.x;
    const unsigned OOB = 0x7FFFFFF0u;
    kvalid = kend - kbeg;
    if (KC) {
      const int rows = min(BM, nrows - row0);
      const int64_t bytes = ((int64_t)(rows - 1) * ld + kvalid) * 4;
      words = rsrc_words(G + (int64_t)row0 * ld + kbeg, (unsigned)(bytes < 0x7FFFFFF0ll ? bytes : 0x7FFFFFF0ll));
      step_bytes = BKT * 4u;
#pragma unroll
      for (int i = 0; i < D::PER_THREAD; ++i) {
        const int f = tid + 256 * i;
        const int row = f >> D::ROW_SHIFT, q = (f & D::SWZ_MASK) ^ ((row >> D::SWZ_SHIFT) & D::SWZ_MASK);
        voff[i] = (unsigned)((row * ld + 4 * q) * 4);
      }
    } else {
      const int64_t bytes = ((int64_t)(kvalid - 1) * ld + (nrows - row0)) * 4;
      words = rsrc_words(G + (int64_t)kbeg * ld + row0, (unsigned)(bytes < 0x7FFFFFF0ll ? bytes : 0x7FFFFFF0ll));
      step_bytes = (unsigned)(BKT * ld) * 4u;
#pragma unroll
      for (int i = 0; i < D::PER_THREAD; ++i) {
        const int f = tid + 256 * i;
        voff[i] = (row0 + 4 * (f & 31) < nrows) ? (unsigned)(((f >> 5) * ld + 4 * (f & 31)) * 4) : OOB;
      }
    }
  }
  // tile of step `step` -> LDS at byte address `dst` (this operand's tile of the stage); slot f = tid + 256 i lands at dst + 16 f
  template <int BKT>
  __device__ __forceinline__ void dma(int step, unsigned dst) const {
    typedef DmaGeom<BKT> D;
    i32x4_t w;      // the descriptor is wave-uniform; say so (an "s" asm operand that hipcc believes divergent is passed in VGPRs)
    w.x = __builtin_amdgcn_readfirstlane(words.x); w.y = __builtin_amdgcn_readfirstlane(words.y);
    w.z = __builtin_amdgcn_readfirstlane(words.z); w.w = __builtin_amdgcn_readfirstlane(words.w);
    const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)step * step_bytes));
    const bool tail = KC && (step + 1) * BKT > kvalid;
    const unsigned wbase = (unsigned)__builtin_amdgcn_readfirstlane((int)(dst + (threadIdx.x >> 6) * 1024u));
#pragma unroll
    for (int i = 0; i < D::PER_THREAD; ++i) {
      unsigned vo = voff[i];
      if (tail) {
        const int f = threadIdx.x + 256 * i;
        const int q = (f & D::SWZ_MASK) ^ (((f >> D::ROW_SHIFT) >> D::SWZ_SHIFT) & D::SWZ_MASK);
        if (step * BKT + 4 * q >= kvalid) vo = 0x7FFFFFF0u;
      }
      lds_dma16_s(w, (unsigned)__builtin_amdgcn_readfirstlane((int)(wbase + 4096u * i)), vo, soff);
    }
  }
  // tile of step `step` (k = kbeg + step * BK ...)
  __device__ __forceinline__ void load(int step, float4 (&r)[4]) const {
    const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane(step) * step_bytes;
    const bool tail = KC && (step + 1) * BK > kvalid;      // uniform; only the last step of an odd-sized K
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      unsigned vo = voff[i];
      if (tail) {
        const int f = threadIdx.x + 256 * i;
        if (step * BK + 4 * (f & 7) >= kvalid) vo = 0x7FFFFFF0u;
      }
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, vo, soff, 0);
      r[i] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    }
  }
};

template <bool KC>
__device__ __forceinline__ void store_tile(float* S, const float4 (&r)[4]) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int f = tid + 256 * i;
    if (KC) *reinterpret_cast<float4*>(S + (f >> 3) * LDS_KC + 4 * (f & 7)) = r[i];
    else    *reinterpret_cast<float4*>(S + (f >> 5) * LDS_RC + 4 * (f & 31)) = r[i];
  }
}

// fragment for one 32-row MFMA tile, k-group g (8 k values), this lane's 4 steps
template <bool KC>
__device__ __forceinline__ void read_frag(const float* S, int row, int g, int h, float (&a)[4]) {
  if (KC) {
    const float4 v = *reinterpret_cast<const float4*>(S + row * LDS_KC + g * 8 + 4 * h);
    a[0] = v.x; a[1] = v.y; a[2] = v.z; a[3] = v.w;
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) a[j] = S[(g * 8 + 4 * h + j) * LDS_RC + row];
  }
}

template <bool A_KC, bool B_KC, int EPI, int DMA>     // DMA: 0 register-staged, else the step width of the LDS-DMA form
__device__ __forceinline__ void gemm_body(const GemmParams& p, float* smem, int tile, int batch);

// LDS-DMA form: 64 KB of dynamic LDS and two blocks per CU at BKT = 32, 32 KB and four at BKT = 16
template <bool A_KC, bool B_KC, int EPI = EPI_NONE, int BKT = 32>
__global__ __launch_bounds__(256, BKT == 32 ? 2 : (EPI == EPI_SMB ? 2 : 4)) void gemm_f32_dma_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) float dsm[];
  gemm_body<A_KC, B_KC, EPI, BKT>(p, dsm, blockIdx.x, blockIdx.z);
}

template <bool A_KC, bool B_KC, int EPI = EPI_NONE, bool OCC3 = (A_KC && B_KC)>
// OCC3: three blocks per CU (150 registers, the accumulators in VGPRs).  The form with both operands k-contiguous (QK^T-shaped
// and 1x1-forward products) gains 4-6 % from the third block covering prologue / store bursts, the unsplit batched TN products
// of the attention backward 3 %; the NN form and the split-K weight-gradient shapes lose 3-10 % with it and stay at two
// (tools/gemm_probe.py).
__global__ __launch_bounds__(256, OCC3 ? 3 : 1) void gemm_f32_kernel(GemmParams p) {
  __shared__ __attribute__((aligned(16))) float smem[2 * TILE_FLOATS];
  gemm_body<A_KC, B_KC, EPI, 0>(p, smem, blockIdx.x, blockIdx.z);
}

// The plain product under a device-side predicate, as ONE small grid: nothing happens unless *p.pred != 0 -- then the resident blocks walk
// the (tile, batch) pairs.  (A predicate inside gemm_f32_kernel would cost a 32 768-block launch ~50 us to find out it has nothing to do;
// here the no-op is a 1 024-block launch.)  The fallback of the folded attention softmax; unsplit shapes.
template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(256, 2) void gemm_f32_pred_kernel(GemmParams p, int tiles, int batches) {
  __shared__ __attribute__((aligned(16))) float smem[2 * TILE_FLOATS];
  if (*p.pred == 0) return;
  for (int t = blockIdx.x; t < tiles * batches; t += gridDim.x) {
    gemm_body<A_KC, B_KC, EPI_NONE, 0>(p, smem, t % tiles, t / tiles);
    __syncthreads();      // the next pair's first tile stores must not overtake this pair's last fragment reads
  }
}

template <bool A_KC, bool B_KC, int EPI, int DMA>
__device__ __forceinline__ void gemm_body(const GemmParams& p, float* smem, int tile, int batch) {
  constexpr bool SMB = EPI == EPI_SMB, EXPB = EPI == EPI_EXPB, ROWNORM = EPI == EPI_ROWNORM;
  static_assert(!ROWNORM || (A_KC && DMA == 0), "the row-sum form is built on the register-staged loop with A k-contiguous");
  float* As = smem;
  float* Bs = smem + TILE_FLOATS;

  const int tile_m = tile % p.tiles_m, tile_n = tile / p.tiles_m;
  const int split = blockIdx.y;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int kbeg = split * p.k_per_split;
  const int kend = min(p.K, kbeg + p.k_per_split);

  const float* A = p.A + batch * p.sA;
  const float* B = p.B + batch * p.sB;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, h = lane >> 5;

  // Output addressing: 32-bit byte offsets into a buffer descriptor that starts at this block's first row; rows >= M
  // and columns >= N get an offset past num_records (loads return 0, stores are dropped): no compares, no 64-bit
  // math per element.  The accumulators START at the residual (the host guarantees alpha == 1 with a residual), so
  // the epilogue is stores only -- vmcnt counts loads and stores in order; a load between stores serialises them.
  const bool to_partial = p.partial != nullptr;
  float* Cb = to_partial ? p.partial + ((int64_t)split * gridDim.z + batch) * (int64_t)p.M * p.N : p.C + batch * p.sC;
  const int ldc = to_partial ? p.N : p.ldc;
  const unsigned OOB = 0x7FFFFFF0u;
  const int rows_here = min(BM, p.M - m0);
  const int tile_bytes = ((rows_here - 1) * ldc + p.N) * 4;
  unsigned colbyte[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int col = n0 + wn * 64 + nt * 32 + li;
    colbyte[nt] = col < p.N ? (unsigned)col * 4u : OOB;
  }
  auto row_byte = [&](int mt, int r) -> unsigned {
    const int row = wm * 64 + mt * 32 + acc_row(r, lane);   // within the block tile
    return row < rows_here ? (unsigned)(row * ldc) * 4u : OOB;
  };

  f32x16 acc[2][2];
  // per-row vectors (rowsub, rowmul) of this tile: buffer loads whose descriptor ends at the tile's last row -- rows past it read 0, no
  // guard (32 guarded loads per lane, each waited for on its own, cost ~4 us per block: 8 % of a QK^T-shaped tile)
  float rmv[2][16];      // SMB: rowmul[row] (or 1)
  if constexpr (SMB || EXPB) {   // accumulators start at -rowsub[row]: the products then add up to dP - D (EXPB: to s - bound)
    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.rowsub) + batch * p.sRow + m0, 0, rows_here * 4, 0x00020000);
    const bool has_mul = SMB && p.rowmul != nullptr;
    const __amdgpu_buffer_rsrc_t mrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(has_mul ? p.rowmul : p.rowsub) + batch * p.sRow + m0, 0, has_mul ? rows_here * 4 : 0, 0x00020000);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * 64 + mt * 32 + acc_row(r, lane);
        const float d = -__uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(srs, row * 4, 0, 0));
        acc[mt][0][r] = d; acc[mt][1][r] = d;
        if constexpr (SMB) {
          const float m = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(mrs, row * 4, 0, 0));
          rmv[mt][r] = has_mul ? m : 1.f;
        }
      }
  } else if (!to_partial && p.residual) {
    const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.residual) + batch * p.sC + (int64_t)m0 * ldc, 0, tile_bytes, 0x00020000);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const unsigned rb_ = row_byte(mt, r);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
          acc[mt][nt][r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rrsrc, rb_ + colbyte[nt], 0, 0));
      }
  } else {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  }

  float rowl[2] = {0.f, 0.f};      // ROWNORM: this lane's share of the sums of rows wm * 64 + {0, 32} + li
  if constexpr (DMA != 0) {
    // One barrier per step: the tiles of step s+1 are requested (LDS-DMA, inline asm: invisible to hipcc's vmcnt bookkeeping, awaited by
    // hand) into the other stage at the top of step s and have its MFMAs to land.  Fragment addresses: one VGPR per (operand, k group)
    // computed once; stage and tile are immediates (the loop body is written out for both stages).
    constexpr int BKT = DMA == 0 ? 32 : DMA;
    typedef DmaGeom<BKT> D;
    const unsigned lds0 = lds_addr_of(smem);
    TileFetch<A_KC> fa;
    TileFetch<B_KC> fb;
    const int nsteps = kbeg < kend ? (kend - kbeg + BKT - 1) / BKT : 0;
    if (nsteps > 0) {
      fa.template init_dma<BKT>(A, p.lda, m0, p.M, kbeg, kend);
      fb.template init_dma<BKT>(B, p.ldb, n0, p.N, kbeg, kend);
      fa.template dma<BKT>(0, lds0);
      fb.template dma<BKT>(0, lds0 + D::TILE_B);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const unsigned hs = (unsigned)(h ^ ((li >> D::SWZ_SHIFT) & D::SWZ_MASK));
    constexpr int NG = BKT / 8;
    unsigned adrA[NG], adrB[NG];      // KC: row base + swizzled slot of k group g; row-contiguous: one base (adr[0])
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      adrA[g] = A_KC ? lds0 + (unsigned)((wm * 64 + li) * (BKT * 4)) + ((hs ^ (2u * g)) << 4) : lds0 + (unsigned)((4 * h * 128 + wm * 64 + li) * 4);
      adrB[g] = B_KC ? lds0 + D::TILE_B + (unsigned)((wn * 64 + li) * (BKT * 4)) + ((hs ^ (2u * g)) << 4)
                     : lds0 + D::TILE_B + (unsigned)((4 * h * 128 + wn * 64 + li) * 4);
    }
    auto frag = [&](auto kc, const unsigned (&adr)[NG], unsigned stage_b, int g, int t, float (&v)[4]) {
      if constexpr (decltype(kc)::value) {
        const f32x4 x = lds_ld128f(adr[g] + stage_b + (unsigned)(t * 32 * BKT * 4));
        v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = lds_ld32(adr[0] + stage_b + (unsigned)(((g * 8 + j) * 128 + t * 32) * 4));
      }
    };
    auto body = [&](auto stage, int step) {
      constexpr unsigned ST = decltype(stage)::value * D::STAGE_B;
      if (step + 1 < nsteps) {
        fa.template dma<BKT>(step + 1, lds0 + (D::STAGE_B - ST));
        fb.template dma<BKT>(step + 1, lds0 + (D::STAGE_B - ST) + D::TILE_B);
      }
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        float a[2][4], b[2][4];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          frag(std::integral_constant<bool, A_KC>{}, adrA, ST, g, t, a[t]);
          frag(std::integral_constant<bool, B_KC>{}, adrB, ST, g, t, b[t]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = mfma32(a[mt][j], b[nt][j], acc[mt][nt]);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    };
    int step = 0;
    for (; step + 1 < nsteps; step += 2) {
      body(std::integral_constant<unsigned, 0>{}, step);
      body(std::integral_constant<unsigned, 1>{}, step + 1);
    }
    if (step < nsteps) body(std::integral_constant<unsigned, 0>{}, step);
  } else {
  float4 ra[4], rb[4];
  TileFetch<A_KC> fa;
  TileFetch<B_KC> fb;
  if (kbeg < kend) {
    fa.init(A, p.lda, m0, p.M, kbeg, kend);
    fb.init(B, p.ldb, n0, p.N, kbeg, kend);
    fa.load(0, ra);
    fb.load(0, rb);
    store_tile<A_KC>(As, ra);
    store_tile<B_KC>(Bs, rb);
  }
  __syncthreads();

  int step = 0;
  for (int k0 = kbeg; k0 < kend; k0 += BK, ++step) {
    const bool has_next = (k0 + BK) < kend;
    if (has_next) {
      fa.load(step + 1, ra);
      fb.load(step + 1, rb);
    }
#pragma unroll
    for (int g = 0; g < BK / 8; ++g) {
      float a[2][4], b[2][4];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        read_frag<A_KC>(As, wm * 64 + t * 32 + li, g, h, a[t]);
        read_frag<B_KC>(Bs, wn * 64 + t * 32 + li, g, h, b[t]);
      }
      if constexpr (ROWNORM) {      // this lane's share of row li's sum: k = 8g + 4h + j (zero past K: out-of-range k reads as 0)
#pragma unroll
        for (int t = 0; t < 2; ++t) rowl[t] += (a[t][0] + a[t][1]) + (a[t][2] + a[t][3]);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = mfma32(a[mt][j], b[nt][j], acc[mt][nt]);
    }
    __syncthreads();
    if (has_next) {
      store_tile<A_KC>(As, ra);
      store_tile<B_KC>(Bs, rb);
      __syncthreads();
    }
  }
  }      // register-staged form

  // epilogue: lane owns column (n0 + wn*64 + nt*32 + li); rows come from the register index.  Stores only.
  const __amdgpu_buffer_rsrc_t crsrc = __builtin_amdgcn_make_buffer_rsrc(Cb + (int64_t)m0 * ldc, 0, tile_bytes, 0x00020000);
  float bv[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
    bv[nt] = (!to_partial && p.bias) ? p.bias[min(n0 + wn * 64 + nt * 32 + li, p.N - 1)] : 0.f;
  const float alpha = to_partial ? 1.f : p.alpha;
  if constexpr (EXPB) {  // exponentials of (score - row bound): exp2 of one product
    const float a2 = alpha * 1.44269504088896f;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const unsigned rb_ = row_byte(mt, r);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(__builtin_amdgcn_exp2f(acc[mt][nt][r] * a2)), crsrc, rb_ + colbyte[nt], 0, 0);
      }
    return;
  }
  if constexpr (ROWNORM) {
    // row sums: the two lane halves hold the k classes 4h .. 4h + 3 of every group of 8; the wn = 0 waves publish them in LDS (the
    // operand tiles are dead: the loop ended on a barrier), every lane then picks up the rows its accumulator registers hold
    float* rl = smem;      // [128]
#pragma unroll
    for (int t = 0; t < 2; ++t) rowl[t] += __shfl_xor(rowl[t], 32, 64);
    if (wn == 0 && h == 0) { rl[wm * 64 + li] = rowl[0]; rl[wm * 64 + 32 + li] = rowl[1]; }
    __syncthreads();
    bool bad = false;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * 64 + mt * 32 + acc_row(r, lane);
        const float l = rl[row];
        const float rinv = 1.f / l;
        bad |= row < rows_here && !(l >= 1e-30f && l < 3.0e38f);
        const unsigned rb_ = row_byte(mt, r);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[mt][nt][r] * rinv), crsrc, rb_ + colbyte[nt], 0, 0);
        if (tile_n == 0 && wn == 0 && li == 0 && row < rows_here) p.rowout[batch * p.sRow + m0 + row] = rinv;
      }
    if (p.flag && __any(bad) && lane == 0) atomicOr(p.flag, 1);
    return;
  }
  if constexpr (SMB) {   // per 32-row half: all loads of the multiplier tile first, then its stores (one half's 32 values live at a time: the
                         // kernel fits 168 registers and three blocks share a CU, so that a block's epilogue runs under two others' MFMAs)
    const __amdgpu_buffer_rsrc_t ersrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.emul) + batch * p.sC + (int64_t)m0 * ldc, 0, tile_bytes, 0x00020000);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      float pv[2][16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const unsigned rb_ = row_byte(mt, r);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
          pv[nt][r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ersrc, rb_ + colbyte[nt], 0, 0));
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const unsigned rb_ = row_byte(mt, r);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[mt][nt][r] * (alpha * rmv[mt][r]) * pv[nt][r]), crsrc, rb_ + colbyte[nt], 0, 0);
      }
    }
    return;
  }
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const unsigned rb_ = row_byte(mt, r);
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[mt][nt][r] * alpha + bv[nt]), crsrc, rb_ + colbyte[nt], 0, 0);
    }
}

// sums the split-K slabs, then alpha / bias / residual
__global__ void gemm_splitk_reduce_kernel(GemmParams p, int batches) {
  const int64_t mn = (int64_t)p.M * p.N;
  const int64_t total = mn * batches;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int batch = (int)(idx / mn);
    const int64_t rem = idx - (int64_t)batch * mn;
    const int row = (int)(rem / p.N), col = (int)(rem - (int64_t)row * p.N);
#ifdef ODVAE_OLD_REDUCE      // A/B build: one serial chain (the form before round 4's last week)
    float s_old = 0.f;
    for (int q = 0; q < p.splits; ++q) s_old += p.partial[((int64_t)q * batches + batch) * mn + rem];
#endif
    float sk[4] = {0.f, 0.f, 0.f, 0.f};      // four independent chains (one chain of `splits` dependent round trips otherwise)
    int sp = 0;
    for (; sp + 3 < p.splits; sp += 4) {
#pragma unroll
      for (int j = 0; j < 4; ++j) sk[j] += p.partial[((int64_t)(sp + j) * batches + batch) * mn + rem];
    }
    for (; sp < p.splits; ++sp) sk[0] += p.partial[((int64_t)sp * batches + batch) * mn + rem];
    float s = (sk[0] + sk[1]) + (sk[2] + sk[3]);
#ifdef ODVAE_OLD_REDUCE
    s = s_old;
#endif
    s *= p.alpha;
    if (p.bias) s += p.bias[col];
    const int64_t o = batch * p.sC + (int64_t)row * p.ldc + col;
    if (p.residual) s += p.residual[o];
    p.C[o] = s;
  }
}

// ODVAE_GEMM_DMA: unset = per shape (see odvae_gemm_f32), 1 = LDS-DMA form everywhere, 0 = register-staged form everywhere
int g_dma_mode = -2;      // -2: environment not read yet
int dma_mode() {
  if (g_dma_mode == -2) {
    const int v = getenv("ODVAE_GEMM_DMA") == nullptr ? -1 : atoi(getenv("ODVAE_GEMM_DMA"));
    g_dma_mode = v < 0 ? -1 : (v > 2 ? 2 : v);
  }
  return g_dma_mode;
}
template <typename K>
int launch_dma(K kern, dim3 grid, hipStream_t st, const GemmParams& p, unsigned lds_bytes = DmaGeom<32>::LDS_B) {
  const unsigned DMA_LDS_B = lds_bytes;
  const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)DMA_LDS_B);
  if (e != hipSuccess) {
    odvae_set_error("gemm_f32: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
    return ODVAE_ERR_HIP;
  }
  hipLaunchKernelGGL(kern, grid, dim3(256), DMA_LDS_B, st, p);
  return ODVAE_OK;
}

int choose_splits(int M, int N, int K, int batch) {
  const int64_t tiles = (int64_t)ceil_div(M, BM) * ceil_div(N, BN) * batch;
  static const int target = getenv("ODVAE_GEMM_SPLIT_BLOCKS") ? atoi(getenv("ODVAE_GEMM_SPLIT_BLOCKS")) : 512;   // two blocks per CU; 1024 / 256 measured 8-12 % slower
  if (tiles >= 512 || K <= 1024) return 1;
  int64_t s = target / tiles;
  const int64_t max_by_k = K / 512;  // at least 16 k-tiles per split
  if (s > max_by_k) s = max_by_k;
  if (s < 1) s = 1;
  return (int)s;
}

}  // namespace

extern "C" {

// Operand staging of odvae_gemm_f32 / odvae_gemm_softmax_bwd_f32: -1 per shape (default), 0 through registers (global -> VGPR -> ds_write,
// one LDS stage, up to three blocks per CU), 1 by LDS-DMA (two stages, one barrier per step).  ODVAE_GEMM_DMA presets it.  Returns the
// previous setting; 2 = LDS-DMA with 16-wide steps (32 KB, four blocks per CU).  Results are identical (same products, same summation order).
int odvae_gemm_select_staging(int mode) {
  const int prev = dma_mode();
  g_dma_mode = mode < 0 ? -1 : (mode > 2 ? 2 : mode);
  return prev;
}

// bytes of split-K scratch odvae_gemm_f32 needs for this shape (0 when it runs unsplit)
size_t odvae_gemm_f32_workspace_bytes(int M, int N, int K, int batch) {
  const int s = choose_splits(M, N, K, batch);
  return s > 1 ? (size_t)s * batch * M * N * sizeof(float) : 0;
}

static int gemm_f32_impl(int transA, int transB, int M, int N, int K, float alpha,
                         const float* A, int lda, int64_t strideA,
                         const float* B, int ldb, int64_t strideB,
                         float* C, int ldc, int64_t strideC,
                         const float* bias, const float* residual, int batch,
                         void* workspace, size_t workspace_bytes, void* stream);

int odvae_gemm_f32(int transA, int transB, int M, int N, int K, float alpha,
                   const float* A, int lda, int64_t strideA,
                   const float* B, int ldb, int64_t strideB,
                   float* C, int ldc, int64_t strideC,
                   const float* bias, const float* residual, int batch,
                   void* workspace, size_t workspace_bytes, void* stream) {
  return gemm_f32_impl(transA, transB, M, N, K, alpha, A, lda, strideA, B, ldb, strideB, C, ldc, strideC, bias, residual, batch,
                       workspace, workspace_bytes, stream);
}

// The same launch under a device-side predicate: every block returns at once unless *pred != 0 (the fallback products of
// odvae_attention_fwd_f32's folded softmax; unsplit shapes only).
int odvae_gemm_pred_f32(int transA, int transB, int M, int N, int K, float alpha,
                        const float* A, int lda, int64_t strideA,
                        const float* B, int ldb, int64_t strideB,
                        float* C, int ldc, int64_t strideC, int batch, const int* pred, void* stream) {
  ODVAE_CHECK_ARG(pred, "gemm_pred: null predicate");
  ODVAE_CHECK_ARG(odvae_gemm_f32_workspace_bytes(M, N, K, batch) == 0, "gemm_pred: split-K shapes are not offered (M=%d N=%d K=%d batch=%d)", M, N, K, batch);
  ODVAE_CHECK_ARG(M > 0 && N > 0 && K > 0 && batch > 0 && A && B && C, "gemm_pred: bad arguments");
  ODVAE_CHECK_ARG((int64_t)BM * ldc * 4 < 0x7FFFFFF0ll, "gemm_pred: ldc %d too large for 32-bit tile offsets", ldc);
  ODVAE_CHECK_ARG(lda % 4 == 0 && ldb % 4 == 0 && strideA % 4 == 0 && strideB % 4 == 0, "gemm_pred: lda/ldb/strides must be multiples of 4 floats");
  ODVAE_CHECK_ARG(((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0, "gemm_pred: A/B must be 16-byte aligned");
  if (!transA || transB) ODVAE_CHECK_ARG(K % 4 == 0, "gemm_pred: a k-contiguous operand needs K %% 4 == 0 (K=%d)", K);
  if (transA) ODVAE_CHECK_ARG(M % 4 == 0, "gemm_pred: transA needs M %% 4 == 0 (M=%d)", M);
  if (!transB) ODVAE_CHECK_ARG(N % 4 == 0, "gemm_pred: transB=0 needs N %% 4 == 0 (N=%d)", N);
  GemmParams p;
  p.A = A; p.B = B; p.C = C; p.bias = nullptr; p.residual = nullptr; p.partial = nullptr;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
  p.sA = strideA; p.sB = strideB; p.sC = strideC; p.alpha = alpha;
  p.rowsub = nullptr; p.emul = nullptr; p.sRow = 0; p.rowmul = nullptr; p.rowout = nullptr; p.flag = nullptr; p.pred = pred;
  p.splits = 1; p.k_per_split = ceil_div(K, BK) * BK; p.tiles_m = ceil_div(M, BM);
  const bool a_kc = !transA, b_kc = transB != 0;
  ODVAE_CHECK_ARG((a_kc ? (int64_t)BM * lda : (int64_t)p.k_per_split * lda) * 4 < 0x7FFFFFF0ll &&
                  (b_kc ? (int64_t)BN * ldb : (int64_t)p.k_per_split * ldb) * 4 < 0x7FFFFFF0ll, "gemm_pred: one block's operand window exceeds 2 GiB");
  const int tiles = p.tiles_m * ceil_div(N, BN);
  const dim3 grid((unsigned)std::min<int64_t>((int64_t)tiles * batch, 1024)), block(256);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (a_kc && b_kc)       hipLaunchKernelGGL((gemm_f32_pred_kernel<true, true>), grid, block, 0, st, p, tiles, batch);
  else if (a_kc)          hipLaunchKernelGGL((gemm_f32_pred_kernel<true, false>), grid, block, 0, st, p, tiles, batch);
  else if (b_kc)          hipLaunchKernelGGL((gemm_f32_pred_kernel<false, true>), grid, block, 0, st, p, tiles, batch);
  else                    hipLaunchKernelGGL((gemm_f32_pred_kernel<false, false>), grid, block, 0, st, p, tiles, batch);
  ODVAE_LAUNCH_CHECK("gemm_pred");
  return ODVAE_OK;
}

static int gemm_f32_impl(int transA, int transB, int M, int N, int K, float alpha,
                         const float* A, int lda, int64_t strideA,
                         const float* B, int ldb, int64_t strideB,
                         float* C, int ldc, int64_t strideC,
                         const float* bias, const float* residual, int batch,
                         void* workspace, size_t workspace_bytes, void* stream) {
  ODVAE_CHECK_ARG(M > 0 && N > 0 && K > 0 && batch > 0, "gemm_f32: empty shape M=%d N=%d K=%d batch=%d", M, N, K, batch);
  ODVAE_CHECK_ARG(A && B && C, "gemm_f32: null operand");
  ODVAE_CHECK_ARG(batch <= 65535, "gemm_f32: batch %d > 65535", batch);
  ODVAE_CHECK_ARG(!(residual && alpha != 1.f), "gemm_f32: a residual needs alpha == 1 (it seeds the accumulators)");
  ODVAE_CHECK_ARG((int64_t)BM * ldc * 4 < 0x7FFFFFF0ll, "gemm_f32: ldc %d too large for 32-bit tile offsets", ldc);
  // float4 staging: the contiguous axis of each operand must be a multiple of 4 and 16-byte aligned
  ODVAE_CHECK_ARG(lda % 4 == 0 && ldb % 4 == 0 && strideA % 4 == 0 && strideB % 4 == 0,
                  "gemm_f32: lda/ldb/strides must be multiples of 4 floats");
  ODVAE_CHECK_ARG(((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0, "gemm_f32: A/B must be 16-byte aligned");
  if (!transA || transB) ODVAE_CHECK_ARG(K % 4 == 0, "gemm_f32: a k-contiguous operand needs K %% 4 == 0 (K=%d)", K);
  if (transA) ODVAE_CHECK_ARG(M % 4 == 0, "gemm_f32: transA needs M %% 4 == 0 (M=%d)", M);
  if (!transB) ODVAE_CHECK_ARG(N % 4 == 0, "gemm_f32: transB=0 needs N %% 4 == 0 (N=%d)", N);

  GemmParams p;
  p.A = A; p.B = B; p.C = C; p.bias = bias; p.residual = residual; p.partial = nullptr;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
  p.sA = strideA; p.sB = strideB; p.sC = strideC; p.alpha = alpha;
  p.rowsub = nullptr; p.emul = nullptr; p.sRow = 0; p.rowmul = nullptr; p.rowout = nullptr; p.flag = nullptr; p.pred = nullptr;
  p.splits = choose_splits(M, N, K, batch);
  p.k_per_split = ceil_div(ceil_div(K, p.splits), BK) * BK;
  p.tiles_m = ceil_div(M, BM);
  if (p.splits > 1) {
    const size_t need = (size_t)p.splits * batch * M * N * sizeof(float);
    if (!workspace || workspace_bytes < need) {
      odvae_set_error("gemm_f32: split-K needs %zu workspace bytes, got %zu", need, workspace_bytes);
      return ODVAE_ERR_WORKSPACE;
    }
    p.partial = static_cast<float*>(workspace);
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  dim3 grid(p.tiles_m * ceil_div(N, BN), p.splits, batch), block(256);
  const bool a_kc = !transA, b_kc = transB != 0;
  // 32-bit byte offsets inside one block's operand window (128 rows x the split's k range)
  ODVAE_CHECK_ARG((a_kc ? (int64_t)BM * lda : (int64_t)p.k_per_split * lda) * 4 < 0x7FFFFFF0ll &&
                  (b_kc ? (int64_t)BN * ldb : (int64_t)p.k_per_split * ldb) * 4 < 0x7FFFFFF0ll,
                  "gemm_f32: one block's operand window exceeds 2 GiB (lda=%d ldb=%d K per split=%d)", lda, ldb, p.k_per_split);
  // Which form: measured on the step's shapes (tools/gemm_probe.py, TFLOP/s, LDS-DMA vs register-staged): TN (both operands
  // row-contiguous: dV = P^T dO, dK = dS^T Q, the 1x1 weight gradients) 140.6 vs 132.3 and 127.2 vs 120.7; NN 132.6 vs 133.5; NT (both
  // k-contiguous, K = 256: QK^T, dO V^T, 1x1 forward) 122.2 vs 129.3 -- eight steps per block there, and the 64 KB of the two DMA stages
  // cost the third block per CU that covers prologue and store bursts.  So: LDS-DMA where A is row-contiguous, ODVAE_GEMM_DMA=1 / 0
  // forces it on / off everywhere.
  const int dma_env = dma_mode();
  const int bkt = dma_env < 0 ? (!a_kc ? 32 : 0) : (dma_env == 0 ? 0 : (dma_env == 2 ? 16 : 32));
  if (bkt == 32) {
    const int rc = a_kc && b_kc ? launch_dma(gemm_f32_dma_kernel<true, true>, grid, st, p)
                 : a_kc         ? launch_dma(gemm_f32_dma_kernel<true, false>, grid, st, p)
                 : b_kc         ? launch_dma(gemm_f32_dma_kernel<false, true>, grid, st, p)
                                : launch_dma(gemm_f32_dma_kernel<false, false>, grid, st, p);
    if (rc != ODVAE_OK) return rc;
  } else if (bkt == 16) {
    const unsigned lb = DmaGeom<16>::LDS_B;
    const int rc = a_kc && b_kc ? launch_dma(gemm_f32_dma_kernel<true, true, false, 16>, grid, st, p, lb)
                 : a_kc         ? launch_dma(gemm_f32_dma_kernel<true, false, false, 16>, grid, st, p, lb)
                 : b_kc         ? launch_dma(gemm_f32_dma_kernel<false, true, false, 16>, grid, st, p, lb)
                                : launch_dma(gemm_f32_dma_kernel<false, false, false, 16>, grid, st, p, lb);
    if (rc != ODVAE_OK) return rc;
  }
  else if (a_kc && b_kc)   hipLaunchKernelGGL((gemm_f32_kernel<true, true>), grid, block, 0, st, p);
  else if (a_kc && !b_kc)  hipLaunchKernelGGL((gemm_f32_kernel<true, false>), grid, block, 0, st, p);
  else if (!a_kc && b_kc)  hipLaunchKernelGGL((gemm_f32_kernel<false, true>), grid, block, 0, st, p);
  else if (p.splits == 1 && batch > 1) hipLaunchKernelGGL((gemm_f32_kernel<false, false, false, true>), grid, block, 0, st, p);
  else                     hipLaunchKernelGGL((gemm_f32_kernel<false, false>), grid, block, 0, st, p);
  ODVAE_LAUNCH_CHECK("gemm_f32");
  if (p.splits > 1) {
    const int64_t total = (int64_t)M * N * batch;
    const int blocks = (int)std::min<int64_t>(ceil_div64(total, 256), 4096);
    hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, p, batch);
    ODVAE_LAUNCH_CHECK("gemm_f32 split-K reduce");
  }
  return ODVAE_OK;
}

// Attention backward, the product dP = dO V^T with the softmax backward folded into its epilogue:
//   dS = alpha * P .* (A B^T - rowdot[row]),   rowdot[i] = sum_j P[i][j] dP[i][j] = dO[i] . O[i]
// (the row sum of P .* dP equals the dot product of the output row and its gradient, so it is known before dP is).
// A [M][K], B [N][K] (both k-contiguous), P and dS [M][N] with leading dimension ldc and batch stride strideC (dS may
// alias P), rowdot [batch][M] with batch stride strideRow.  Replaces bmm + the softmax backward of
// [UPSTREAM] ldm AttnBlock.forward under autograd.
static int gemm_smb_impl(int M, int N, int K, float alpha, const float* A, int lda, int64_t strideA, const float* B, int ldb, int64_t strideB,
                         const float* P, const float* rowdot, const float* rowmul, int64_t strideRow,
                         float* dS, int ldc, int64_t strideC, int batch, void* stream);

int odvae_gemm_softmax_bwd_f32(int M, int N, int K, float alpha,
                               const float* A, int lda, int64_t strideA,
                               const float* B, int ldb, int64_t strideB,
                               const float* P, const float* rowdot, int64_t strideRow,
                               float* dS, int ldc, int64_t strideC, int batch, void* stream) {
  return gemm_smb_impl(M, N, K, alpha, A, lda, strideA, B, ldb, strideB, P, rowdot, nullptr, strideRow, dS, ldc, strideC, batch, stream);
}

// The same with the probabilities given as unnormalised exponentials E and 1 / l per row (odvae_attention_fwd_f32's folded softmax):
//   dS = alpha * E .* rinv[row] .* (A B^T - rowdot[row])
int odvae_gemm_softmax_bwd_scaled_f32(int M, int N, int K, float alpha,
                                      const float* A, int lda, int64_t strideA,
                                      const float* B, int ldb, int64_t strideB,
                                      const float* E, const float* rowdot, const float* rinv, int64_t strideRow,
                                      float* dS, int ldc, int64_t strideC, int batch, void* stream) {
  ODVAE_CHECK_ARG(rinv, "gemm_softmax_bwd_scaled: null row factor");
  return gemm_smb_impl(M, N, K, alpha, A, lda, strideA, B, ldb, strideB, E, rowdot, rinv, strideRow, dS, ldc, strideC, batch, stream);
}

static int gemm_smb_impl(int M, int N, int K, float alpha, const float* A, int lda, int64_t strideA, const float* B, int ldb, int64_t strideB,
                         const float* P, const float* rowdot, const float* rowmul, int64_t strideRow,
                         float* dS, int ldc, int64_t strideC, int batch, void* stream) {
  ODVAE_CHECK_ARG(M > 0 && N > 0 && K > 0 && batch > 0, "gemm_softmax_bwd: empty shape M=%d N=%d K=%d batch=%d", M, N, K, batch);
  ODVAE_CHECK_ARG(A && B && P && rowdot && dS, "gemm_softmax_bwd: null operand");
  ODVAE_CHECK_ARG(batch <= 65535, "gemm_softmax_bwd: batch %d > 65535", batch);
  ODVAE_CHECK_ARG((int64_t)BM * ldc * 4 < 0x7FFFFFF0ll, "gemm_softmax_bwd: ldc %d too large for 32-bit tile offsets", ldc);
  ODVAE_CHECK_ARG(lda % 4 == 0 && ldb % 4 == 0 && strideA % 4 == 0 && strideB % 4 == 0 && K % 4 == 0,
                  "gemm_softmax_bwd: lda/ldb/strides/K must be multiples of 4 floats");
  ODVAE_CHECK_ARG(((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0, "gemm_softmax_bwd: A/B must be 16-byte aligned");
  GemmParams p;
  p.A = A; p.B = B; p.C = dS; p.bias = nullptr; p.residual = nullptr; p.partial = nullptr;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
  p.sA = strideA; p.sB = strideB; p.sC = strideC; p.alpha = alpha;
  p.splits = 1; p.k_per_split = ceil_div(K, BK) * BK; p.tiles_m = ceil_div(M, BM);
  p.rowsub = rowdot; p.emul = P; p.sRow = strideRow; p.rowmul = rowmul; p.rowout = nullptr; p.flag = nullptr; p.pred = nullptr;
  dim3 grid(p.tiles_m * ceil_div(N, BN), 1, batch), block(256);
  if (dma_mode() == 2) {
    const int rc = launch_dma(gemm_f32_dma_kernel<true, true, true, 16>, grid, static_cast<hipStream_t>(stream), p, DmaGeom<16>::LDS_B);
    if (rc != ODVAE_OK) return rc;
  } else if (dma_mode() == 1) {
    const int rc = launch_dma(gemm_f32_dma_kernel<true, true, true>, grid, static_cast<hipStream_t>(stream), p);
    if (rc != ODVAE_OK) return rc;
  } else {
    hipLaunchKernelGGL((gemm_f32_kernel<true, true, true>), grid, block, 0, static_cast<hipStream_t>(stream), p);
  }
  ODVAE_LAUNCH_CHECK("gemm_softmax_bwd");
  return ODVAE_OK;
}

// Attention forward with the softmax folded into the two products ([UPSTREAM] AttnBlock.forward: bmm -> softmax -> bmm).
// (1) E = exp(alpha * (A B^T - rowbound[row])): A [M][K], B [N][K] k-contiguous; rowbound[i] >= max_j (A B^T)[i][j] (odvae_attn_row_bound_f32)
int odvae_gemm_exp_bound_f32(int M, int N, int K, float alpha,
                             const float* A, int lda, int64_t strideA, const float* B, int ldb, int64_t strideB,
                             const float* rowbound, int64_t strideRow, float* E, int ldc, int64_t strideC, int batch, void* stream) {
  ODVAE_CHECK_ARG(M > 0 && N > 0 && K > 0 && batch > 0 && batch <= 65535, "gemm_exp_bound: bad shape M=%d N=%d K=%d batch=%d", M, N, K, batch);
  ODVAE_CHECK_ARG(A && B && rowbound && E, "gemm_exp_bound: null operand");
  ODVAE_CHECK_ARG((int64_t)BM * ldc * 4 < 0x7FFFFFF0ll, "gemm_exp_bound: ldc %d too large for 32-bit tile offsets", ldc);
  ODVAE_CHECK_ARG(lda % 4 == 0 && ldb % 4 == 0 && strideA % 4 == 0 && strideB % 4 == 0 && K % 4 == 0,
                  "gemm_exp_bound: lda/ldb/strides/K must be multiples of 4 floats");
  ODVAE_CHECK_ARG(((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0, "gemm_exp_bound: A/B must be 16-byte aligned");
  GemmParams p;
  p.A = A; p.B = B; p.C = E; p.bias = nullptr; p.residual = nullptr; p.partial = nullptr;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
  p.sA = strideA; p.sB = strideB; p.sC = strideC; p.alpha = alpha;
  p.splits = 1; p.k_per_split = ceil_div(K, BK) * BK; p.tiles_m = ceil_div(M, BM);
  p.rowsub = rowbound; p.emul = nullptr; p.sRow = strideRow; p.rowmul = nullptr; p.rowout = nullptr; p.flag = nullptr; p.pred = nullptr;
  hipLaunchKernelGGL((gemm_f32_kernel<true, true, EPI_EXPB>), dim3(p.tiles_m * ceil_div(N, BN), 1, batch), dim3(256), 0, static_cast<hipStream_t>(stream), p);
  ODVAE_LAUNCH_CHECK("gemm_exp_bound");
  return ODVAE_OK;
}

// (2) C = (A B) / l[row], l[row] = sum_k A[row][k]: A [M][K] k-contiguous (the exponentials), B [K][N] n-contiguous (V);
//     rinv[batch][M] = 1 / l; *flag |= 1 if some l is not a normal number >= 1e-30 (the bound was too loose: every exponential of a
//     row underflowed -- the caller's predicated fallback then redoes the block with the row maximum)
int odvae_gemm_rownorm_f32(int M, int N, int K, const float* A, int lda, int64_t strideA, const float* B, int ldb, int64_t strideB,
                           float* C, int ldc, int64_t strideC, float* rinv, int64_t strideRow, int* flag, int batch, void* stream) {
  ODVAE_CHECK_ARG(M > 0 && N > 0 && K > 0 && batch > 0 && batch <= 65535, "gemm_rownorm: bad shape M=%d N=%d K=%d batch=%d", M, N, K, batch);
  ODVAE_CHECK_ARG(A && B && C && rinv, "gemm_rownorm: null operand");
  ODVAE_CHECK_ARG((int64_t)BM * ldc * 4 < 0x7FFFFFF0ll && (int64_t)BM * lda * 4 < 0x7FFFFFF0ll && (int64_t)K * ldb * 4 < 0x7FFFFFF0ll,
                  "gemm_rownorm: operand window too large for 32-bit offsets");
  ODVAE_CHECK_ARG(lda % 4 == 0 && ldb % 4 == 0 && strideA % 4 == 0 && strideB % 4 == 0 && K % 4 == 0 && N % 4 == 0,
                  "gemm_rownorm: lda/ldb/strides/K/N must be multiples of 4 floats");
  ODVAE_CHECK_ARG(((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0, "gemm_rownorm: A/B must be 16-byte aligned");
  GemmParams p;
  p.A = A; p.B = B; p.C = C; p.bias = nullptr; p.residual = nullptr; p.partial = nullptr;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
  p.sA = strideA; p.sB = strideB; p.sC = strideC; p.alpha = 1.f;
  p.splits = 1; p.k_per_split = ceil_div(K, BK) * BK; p.tiles_m = ceil_div(M, BM);
  p.rowsub = nullptr; p.emul = nullptr; p.sRow = strideRow; p.rowmul = nullptr; p.rowout = rinv; p.flag = flag; p.pred = nullptr;
  hipLaunchKernelGGL((gemm_f32_kernel<true, false, EPI_ROWNORM>), dim3(p.tiles_m * ceil_div(N, BN), 1, batch), dim3(256), 0, static_cast<hipStream_t>(stream), p);
  ODVAE_LAUNCH_CHECK("gemm_rownorm");
  return ODVAE_OK;
}

}  // extern "C"
