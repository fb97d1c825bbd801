#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
// Probe ds_read_b64_tr_b16: LDS tile T[row][col] of 16-bit values = row*256 + col (rows 0..63, cols 0..63, stride 64 elements).
__global__ void probe(int* out) {
  __shared__ __attribute__((aligned(16))) unsigned short T[64 * 64];
  for (int i = threadIdx.x; i < 64 * 64; i += 64) T[i] = (unsigned short)((i / 64) * 256 + (i % 64));
  __syncthreads();
  int lane = threadIdx.x;
  int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
  // group g reads block rows (4g .. 4g+3), cols 0..15: lane 4q+p supplies address of row 4g+q, cols 4p..4p+3
  const unsigned short* addr = &T[(4 * g + q) * 64 + 4 * p];
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)addr);
  for (int e = 0; e < 4; ++e) out[lane * 4 + e] = (unsigned short)v[e];
}
int main() {
  int* d; hipMalloc(&d, 256 * 4);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
  int h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) { printf("lane %2d:", l); for (int e = 0; e < 4; ++e) printf(" (r%d,c%d)", h[l*4+e] / 256, h[l*4+e] % 256); printf("\n"); }
  return 0;
}
