#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(2))) int i32x2;
typedef __attribute__((address_space(3))) i32x2 lds_i2;
// probe of ds_read_b64_tr_b8: LDS byte a holds (a & 0xff) in pass 0 and (a >> 8) in pass 1; lane l reads at byte address ADDR(l)
__global__ void k(uint8_t* out, int pass, int mode) {
  __shared__ __attribute__((aligned(16))) uint8_t s[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) s[i] = pass ? (uint8_t)(i >> 8) : (uint8_t)i;
  __syncthreads();
  const int l = threadIdx.x;
  int addr = mode == 0 ? l * 8 : mode == 1 ? l * 16 : (l & 15) * 16 + (l >> 4) * 8 + 1024 * 0;
  i32x2 v = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_i2*)(uintptr_t)(uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)(s + addr));
  reinterpret_cast<i32x2*>(out)[l] = v;
}
int main() {
  uint8_t *d, h0[512], h1[512];
  hipMalloc(&d, 512);
  for (int mode = 0; mode < 3; ++mode) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, 0, mode); hipMemcpy(h0, d, 512, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, 1, mode); hipMemcpy(h1, d, 512, hipMemcpyDeviceToHost);
    printf("mode %d: lane: source byte addresses of its 8 result bytes\n", mode);
    for (int l = 0; l < 64; ++l) { printf("%2d:", l); for (int b = 0; b < 8; ++b) printf(" %4d", h0[l * 8 + b] | (h1[l * 8 + b] << 8)); printf("\n"); }
  }
  return 0;
}
