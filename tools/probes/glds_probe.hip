#include <hip/hip_runtime.h>
#include <stdio.h>
// Probe __builtin_amdgcn_global_load_lds: does lane l land at lds_base + l*16 ?
__global__ void probe(const unsigned* g, unsigned* out) {
  __shared__ __attribute__((aligned(16))) unsigned T[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) T[i] = 0xdeadbeef;
  __syncthreads();
  // each lane points at its own 16 bytes of global memory, permuted: lane l reads chunk (l ^ 5)
  const unsigned* src = g + ((threadIdx.x ^ 5) * 4);
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)(T + 256), 16, 0, 0);
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  for (int i = threadIdx.x; i < 1024; i += 64) out[i] = T[i];
}
int main() {
  unsigned h[256]; for (int i = 0; i < 256; ++i) h[i] = i;
  unsigned *g, *o; hipMalloc(&g, sizeof(h)); hipMalloc(&o, 4096); hipMemcpy(g, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, g, o);
  unsigned r[1024]; hipMemcpy(r, o, 4096, hipMemcpyDeviceToHost);
  printf("T[252..259]: "); for (int i = 252; i < 260; ++i) printf("%x ", r[i]); printf("\n");
  int ok = 1; for (int l = 0; l < 64; ++l) for (int e = 0; e < 4; ++e) if (r[256 + l * 4 + e] != (unsigned)(((l ^ 5) * 4) + e)) ok = 0;
  printf("lane l -> lds_base + 16*l with its own source: %s\n", ok ? "YES" : "NO");
  printf("T[256..271]: "); for (int i = 256; i < 272; ++i) printf("%u ", r[i]); printf("\n");
  printf("after: %x\n", r[512]);
  return 0;
}
