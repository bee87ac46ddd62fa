// epi_probe.hip — semantics of the instructions the round-3 epilogue of k_bwd8 relies on, checked on the device:
//   (1) MODE.FP16_OVFL (bit 23) = 1: does v_cvt_pk_fp8_f32 SATURATE at +-448 instead of returning NaN?  (the saturation then
//       costs no v_med3_f32 per value); what happens to +-inf and NaN inputs; are in-range results unchanged?
//   (2) v_fma_mix_f32 op_sel:[1,0,0] op_sel_hi:[1,0,0]: f16 (high half of src0) * f32 + 0 in one instruction
//   (3) v_lshlrev_b32_sdwa src1_sel:BYTE_n: (byte n of a dword) << 2 in one instruction (table address of a phase byte)
// build: hipcc -O3 --offload-arch=gfx950 scripts/probes/epi_probe.hip -o scripts/probes/epi_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;

__global__ void k_ovfl(const float* in, unsigned* packed_ovfl, unsigned* packed_plain, unsigned* packed_med3, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i * 4 + 3 >= n) return;
  const float a = in[4 * i], b = in[4 * i + 1], c = in[4 * i + 2], d = in[4 * i + 3];
  int p = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
  p = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, p, true);
  packed_plain[i] = (unsigned)p;
  const float L = 448.0f;
  int m = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(a, -L, L), __builtin_amdgcn_fmed3f(b, -L, L), 0, false);
  m = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(c, -L, L), __builtin_amdgcn_fmed3f(d, -L, L), m, true);
  packed_med3[i] = (unsigned)m;
  unsigned q;
  // the conversions run between the two s_setreg: one asm statement, so nothing of the compiler's is scheduled in between
  asm volatile(
      "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1\n\t"
      "s_nop 2\n\t"
      "v_mov_b32 %0, 0\n\t"
      "v_cvt_pk_fp8_f32 %0, %1, %2\n\t"
      "v_cvt_pk_fp8_f32 %0, %3, %4 op_sel:[0,0,1]\n\t"
      "s_nop 2\n\t"
      "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 0\n\t"
      : "=&v"(q) : "v"(a), "v"(b), "v"(c), "v"(d));
  packed_ovfl[i] = q;
}

__global__ void k_mix(const unsigned* h2, const float* g, float* out_hi, float* out_lo, unsigned* out_shift, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned e = h2[i];
  const float gg = g[i];
  float r_hi, r_lo;
  asm volatile("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r_hi) : "v"(e), "v"(gg));
  asm volatile("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r_lo) : "v"(e), "v"(gg));
  out_hi[i] = r_hi; out_lo[i] = r_lo;
  unsigned s0, s1, s2, s3;
  const unsigned two = 2;
  asm volatile("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(s0) : "v"(two), "v"(e));
  asm volatile("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(s1) : "v"(two), "v"(e));
  asm volatile("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(s2) : "v"(two), "v"(e));
  asm volatile("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(s3) : "v"(two), "v"(e));
  out_shift[4 * i] = s0; out_shift[4 * i + 1] = s1; out_shift[4 * i + 2] = s2; out_shift[4 * i + 3] = s3;
}

int main() {
  const int n = 1 << 16;
  float* h = (float*)malloc(n * 4);
  srand(2);
  for (int i = 0; i < n; ++i) h[i] = ldexpf(((float)rand() / RAND_MAX - 0.5f) * 2.f, (rand() % 26) - 12);   // |x| up to ~2^13
  const float sp[] = {448.f, 449.f, 463.9f, 464.f, 465.f, 480.f, 1e6f, -1e6f, INFINITY, -INFINITY, NAN, -448.f, -449.f, -480.f, 447.9f, 0.f};
  for (int i = 0; i < 16; ++i) h[i] = sp[i];
  float* din; unsigned *d_o, *d_p, *d_m;
  hipMalloc(&din, n * 4); hipMalloc(&d_o, n); hipMalloc(&d_p, n); hipMalloc(&d_m, n);
  hipMemcpy(din, h, n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_ovfl, dim3(n / 4 / 256), dim3(256), 0, 0, din, d_o, d_p, d_m, n);
  unsigned char *ho = (unsigned char*)malloc(n), *hp = (unsigned char*)malloc(n), *hm = (unsigned char*)malloc(n);
  hipMemcpy(ho, d_o, n, hipMemcpyDeviceToHost); hipMemcpy(hp, d_p, n, hipMemcpyDeviceToHost); hipMemcpy(hm, d_m, n, hipMemcpyDeviceToHost);
  printf("special inputs:   x        plain  med3  ovfl\n");
  for (int i = 0; i < 16; ++i) printf("  %12g   0x%02x  0x%02x  0x%02x\n", h[i], hp[i], hm[i], ho[i]);
  int diff_med3 = 0, diff_inrange = 0, n_in = 0, n_out = 0;
  for (int i = 16; i < n; ++i) {
    const bool in_range = fabsf(h[i]) < 464.f;
    if (in_range) { ++n_in; if (ho[i] != hp[i]) ++diff_inrange; } else ++n_out;
    if (ho[i] != hm[i]) { if (diff_med3 < 8) printf("  ovfl != med3 at x = %g: 0x%02x vs 0x%02x\n", h[i], ho[i], hm[i]); ++diff_med3; }
  }
  printf("FP16_OVFL=1 v_cvt_pk_fp8_f32: %d in-range values differ from the plain conversion (of %d); %d of %d values differ from med3+convert (%d out of range)\n",
         diff_inrange, n_in, diff_med3, n - 16, n_out);

  // (2), (3)
  unsigned* hh = (unsigned*)malloc(n * 4);
  for (int i = 0; i < n; ++i) {
    const f16x2 v = {(_Float16)(((float)rand() / RAND_MAX - 0.5f) * 2.f), (_Float16)(((float)rand() / RAND_MAX - 0.5f) * 2.f)};
    memcpy(&hh[i], &v, 4);
    h[i] = ((float)rand() / RAND_MAX - 0.5f) * 600.f;
  }
  unsigned *d_h, *d_s; float *d_hi, *d_lo;
  hipMalloc(&d_h, n * 4); hipMalloc(&d_s, n * 16); hipMalloc(&d_hi, n * 4); hipMalloc(&d_lo, n * 4);
  hipMemcpy(d_h, hh, n * 4, hipMemcpyHostToDevice); hipMemcpy(din, h, n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_mix, dim3(n / 256), dim3(256), 0, 0, d_h, din, d_hi, d_lo, d_s, n);
  float *rhi = (float*)malloc(n * 4), *rlo = (float*)malloc(n * 4); unsigned* rs = (unsigned*)malloc(n * 16);
  hipMemcpy(rhi, d_hi, n * 4, hipMemcpyDeviceToHost); hipMemcpy(rlo, d_lo, n * 4, hipMemcpyDeviceToHost); hipMemcpy(rs, d_s, n * 16, hipMemcpyDeviceToHost);
  int bad_mix = 0, bad_sh = 0;
  for (int i = 0; i < n; ++i) {
    f16x2 v; memcpy(&v, &hh[i], 4);
    const float want_hi = (float)v[1] * h[i], want_lo = (float)v[0] * h[i];     // one rounding, as v_cvt + v_mul give
    if (rhi[i] != want_hi || rlo[i] != want_lo) { if (bad_mix < 8) printf("  fma_mix: got %g / %g want %g / %g\n", rhi[i], rlo[i], want_hi, want_lo); ++bad_mix; }
    for (int b = 0; b < 4; ++b) if (rs[4 * i + b] != (((hh[i] >> (8 * b)) & 0xffu) << 2)) ++bad_sh;
  }
  printf("v_fma_mix_f32 (f16 half * f32): %d / %d mismatches vs cvt+mul;  v_lshlrev_b32_sdwa BYTE_n: %d / %d mismatches\n", bad_mix, n, bad_sh, 4 * n);
  return 0;
}
