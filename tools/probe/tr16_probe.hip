// Probe of ds_read_b64_tr_b16 semantics on gfx950: LDS holds a row-major [64 k][136] bf16 tile with t[k][m] = k*256 + m
// (as integers in 16-bit lanes); every lane passes the address of 4 contiguous elements of a 4x16 block and we print what it
// gets back.   hipcc --offload-arch=gfx950 -O2 tools/probe/tr16_probe.hip -o /tmp/tr16_probe && /tmp/tr16_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) short s16x4;
__global__ void k(uint16_t* out) {
  __shared__ uint16_t t[64 * 136];
  for (int i = threadIdx.x; i < 64 * 136; i += 64) t[i] = (uint16_t)((i / 136) * 256 + (i % 136));
  __syncthreads();
  const int lane = threadIdx.x, i = lane & 15, g = lane >> 4;
  // block of group g: rows k = g*8 .. g*8+3, cols m = 32 .. 47; lane i supplies row i/4, cols 32 + (i%4)*4 ..
  __attribute__((address_space(3))) s16x4* p = (__attribute__((address_space(3))) s16x4*)&t[(g * 8 + (i >> 2)) * 136 + 32 + (i & 3) * 4];
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p);
  for (int e = 0; e < 4; ++e) out[lane * 4 + e] = (uint16_t)v[e];
}
int main() {
  uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  uint16_t h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int lane = 0; lane < 64; ++lane) {
    printf("lane %2d:", lane);
    for (int e = 0; e < 4; ++e) printf(" (k=%d,m=%d)", h[lane * 4 + e] / 256, h[lane * 4 + e] % 256);
    printf("\n");
  }
  return 0;
}
