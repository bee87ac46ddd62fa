// isa_probe.hip — semantics of the gfx950 instructions the 8-bit scratch path relies on, checked on the device:
//   (1) v_add_f32_sdwa dst_sel:BYTE_n dst_unused:UNUSED_PRESERVE : does the LOW byte of the f32 sum land in byte n?
//       (phase byte = round(t * 256) mod 256 via the magic-number add t + 1.5 * 2^15)
//   (2) v_cvt_scalef32_pk_f16_fp8 / v_cvt_pk_fp8_f32 round trip (OCP e4m3, saturation behaviour)
//   (3) v_cvt_f32_ubyteN decode
// build: hipcc -O3 --offload-arch=gfx950 scripts/probes/isa_probe.hip -o /tmp/isa_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;

__global__ void k_sdwa(const float* in, unsigned* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i * 4 + 3 >= n) return;
  const float a = in[4 * i], b = in[4 * i + 1], c = in[4 * i + 2], d = in[4 * i + 3];
  const float magic = 49152.0f;   // 1.5 * 2^15: ulp = 2^-8, so the low mantissa byte is round(t * 256) mod 256
  unsigned r = 0xdeadbeefu;
  asm volatile("v_add_f32_sdwa %0, %1, %2 dst_sel:BYTE_0 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(r) : "v"(a), "v"(magic));
  asm volatile("v_add_f32_sdwa %0, %1, %2 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(r) : "v"(b), "v"(magic));
  asm volatile("v_add_f32_sdwa %0, %1, %2 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(r) : "v"(c), "v"(magic));
  asm volatile("v_add_f32_sdwa %0, %1, %2 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(r) : "v"(d), "v"(magic));
  out[i] = r;
}

__global__ void k_fp8(const float* in, unsigned* packed, float* back, float scale, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i * 4 + 3 >= n) return;
  const float inv = 1.0f / scale;
  int p = __builtin_amdgcn_cvt_pk_fp8_f32(in[4 * i] * inv, in[4 * i + 1] * inv, 0, false);
  p = __builtin_amdgcn_cvt_pk_fp8_f32(in[4 * i + 2] * inv, in[4 * i + 3] * inv, p, true);
  packed[i] = (unsigned)p;
  const f16x2 lo = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(p, scale, false);
  const f16x2 hi = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(p, scale, true);
  back[4 * i] = (float)lo[0]; back[4 * i + 1] = (float)lo[1]; back[4 * i + 2] = (float)hi[0]; back[4 * i + 3] = (float)hi[1];
}

static float e4m3_ref(float x) {   // OCP e4m3fn, round-nearest-even, saturating at 448
  if (x != x) return x;
  float a = fabsf(x);
  if (a >= 448.f) a = 448.f;   // (values in (448, 464) round to 448; above: saturate is what we expect to see)
  if (a < ldexpf(1.f, -10)) return copysignf(0.f, x);
  int e; frexpf(a, &e); e -= 1;           // a = m * 2^e, m in [1,2)
  if (e < -6) e = -6;                       // subnormal: fixed quantum 2^-9
  const float q = ldexpf(1.f, e - 3);
  float r = nearbyintf(a / q) * q;
  if (r > 448.f) r = 448.f;
  return copysignf(r, x);
}

int main() {
  const int n = 1 << 16;
  float* h = (float*)malloc(n * 4);
  srand(1);
  for (int i = 0; i < n; ++i) h[i] = ((float)rand() / RAND_MAX - 0.5f) * 40.0f;   // phases in +-20 revolutions
  h[0] = 0.f; h[1] = 1.0f / 512.f; h[2] = -1.0f / 512.f; h[3] = 0.998046875f + 1.0f / 1024.f; h[4] = 3.0f / 512.f;
  float* din; unsigned* dout;
  hipMalloc(&din, n * 4); hipMalloc(&dout, n);
  hipMemcpy(din, h, n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_sdwa, dim3(n / 4 / 256), dim3(256), 0, 0, din, dout, n);
  unsigned* ho = (unsigned*)malloc(n);
  hipMemcpy(ho, dout, n, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < n; ++i) {
    const unsigned got = (ho[i / 4] >> (8 * (i & 3))) & 0xff;
    const float t = h[i] + 49152.0f;                       // the same f32 add on the host (RNE)
    unsigned bits; memcpy(&bits, &t, 4);
    const unsigned want = bits & 0xff;
    const long lr = lrintf(h[i] * 256.0f);                 // and its meaning: round(t*256) mod 256
    const unsigned want2 = (unsigned)(((lr % 256) + 256) % 256);
    if (got != want || want != want2) { if (bad < 8) printf("sdwa mismatch i=%d t=%g got %u want %u/%u\n", i, h[i], got, want, want2); ++bad; }
  }
  printf("sdwa byte-select add: %d / %d mismatches\n", bad, n);

  for (int i = 0; i < n; ++i) h[i] = ldexpf(((float)rand() / RAND_MAX - 0.5f) * 2.f, (rand() % 24) - 12);   // |x| up to ~2^11
  h[0] = 448.f; h[1] = 449.f; h[2] = 480.f; h[3] = 1e6f; h[4] = -1e6f; h[5] = 0.001953125f; h[6] = 0.0009765625f; h[7] = 0.0009f;
  hipMemcpy(din, h, n * 4, hipMemcpyHostToDevice);
  unsigned* dp; float* db;
  hipMalloc(&dp, n); hipMalloc(&db, n * 4);
  for (float scale : {1.0f, 0.25f}) {
    hipLaunchKernelGGL(k_fp8, dim3(n / 4 / 256), dim3(256), 0, 0, din, dp, db, scale, n);
    float* hb = (float*)malloc(n * 4);
    hipMemcpy(hb, db, n * 4, hipMemcpyDeviceToHost);
    int bad8 = 0, sat = 0;
    for (int i = 0; i < n; ++i) {
      const float want = e4m3_ref(h[i] / scale) * scale;
      if (fabsf(h[i] / scale) > 448.f) ++sat;
      if (hb[i] != want && !(hb[i] != hb[i] && want != want)) { if (bad8 < 8) printf("fp8 mismatch x=%g scale %g got %g want %g\n", h[i], scale, hb[i], want); ++bad8; }
    }
    printf("fp8 e4m3 pack/unpack (scale %g): %d / %d mismatches (%d inputs beyond +-448 -> saturate)\n", scale, bad8, n, sat);
    printf("  first values: %g->%g %g->%g %g->%g %g->%g %g->%g\n", h[0], hb[0], h[1], hb[1], h[2], hb[2], h[3], hb[3], h[7], hb[7]);
    free(hb);
  }
  return 0;
}
