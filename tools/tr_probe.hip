// Probe: semantics of ds_read_b64_tr_b16 and the bf16 32x32x16 MFMA operand / accumulator maps on gfx950.
// hipcc --offload-arch=gfx950 -O2 tools/tr_probe.hip -o tools/bin/tr_probe && tools/bin/tr_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// LDS image: 16 rows x 64 cols of int16, value = row*100 + col.  Group g (16 lanes) reads the 4x16 block at rows 4g..4g+3,
// cols 0..15: lane 4q+p supplies the address of (row 4g+q, col 4p).
__global__ void tr_kernel(short* out) {
  __shared__ __attribute__((aligned(16))) short lds[16 * 64];
  for (int i = threadIdx.x; i < 16 * 64; i += 64) lds[i] = (short)((i / 64) * 100 + (i % 64));
  __syncthreads();
  int lane = threadIdx.x, g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  short* addr = lds + (4 * g + q) * 64 + 4 * p;
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)addr);
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = v[j];
}

// D = A * B with A[i][k] = i + 0.01 k?  use exact small integers: A[i][k] = (i == k), B[k][j] = k*32 + j  (K = 16)
__global__ void mfma_kernel(float* out) {
  int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) {
    int k = 8 * h + j;
    a[j] = (__bf16)((r == k) ? 1.0f : 0.0f);          // A[row r][k]
    b[j] = (__bf16)(float)(k * 4 + (r & 3) + 64 * (r >> 2) * 0 + (r >> 2));  // B[k][col r] = 4k + (r&3) + (r>>2)  (asymmetric, exact in bf16: < 256)
  }
  f32x16 c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  for (int i = 0; i < 16; ++i) out[lane * 16 + i] = c[i];
}

int main() {
  short* d; float* f;
  hipMalloc(&d, 64 * 4 * 2); hipMalloc(&f, 64 * 16 * 4);
  hipLaunchKernelGGL(tr_kernel, dim3(1), dim3(64), 0, 0, d);
  hipLaunchKernelGGL(mfma_kernel, dim3(1), dim3(64), 0, 0, f);
  short h[256]; float hf[1024];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost); hipMemcpy(hf, f, sizeof(hf), hipMemcpyDeviceToHost);
  // expected per the guide: lane i of group g receives column i of rows 4g..4g+3: element q = (4g+q)*100 + i
  int bad = 0;
  for (int lane = 0; lane < 64; ++lane) for (int q = 0; q < 4; ++q) {
    int g = lane >> 4, i = lane & 15, want = (4 * g + q) * 100 + i;
    if (h[lane * 4 + q] != want) { if (bad < 8) printf("tr: lane %d elem %d got %d want %d\n", lane, q, h[lane * 4 + q], want); ++bad; }
  }
  printf("tr_read: %s (%d mismatches)\n", bad ? "DIFFERENT FROM GUIDE" : "matches guide (lane i <- column i, element q <- row q)", bad);
  for (int lane = 0; lane < 20; lane += 1) printf("lane %2d: %5d %5d %5d %5d\n", lane, h[lane*4], h[lane*4+1], h[lane*4+2], h[lane*4+3]);
  // mfma: D[i][j] = sum_k A[i][k] B[k][j] = B[i][j] for i < 16 else 0 ; D reg r of lane l is row (r&3)+8*(r>>2)+4*(l>>5), col l&31
  bad = 0;
  for (int lane = 0; lane < 64; ++lane) for (int r = 0; r < 16; ++r) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = lane & 31;
    float want = row < 16 ? (float)(row * 4 + (col & 3) + (col >> 2)) : 0.0f;
    if (hf[lane * 16 + r] != want) { if (bad < 8) printf("mfma: lane %d reg %d (row %d col %d) got %g want %g\n", lane, r, row, col, hf[lane*16+r], want); ++bad; }
  }
  printf("mfma_f32_32x32x16_bf16 maps: %s (%d mismatches)\n", bad ? "DIFFERENT" : "A[r][8h+j], B[8h+j][r], D row (r&3)+8(r>>2)+4h col lane&31 confirmed", bad);
  return 0;
}
