// siren_kernels.hip — hand-written gfx950 (CDNA4) kernels of the SIREN fitting hot path.
//
// Reference arithmetic being replaced (paths relative to the reference tree):
//   implicit_image/models/siren.py:56-68,123-134      Linear -> sin(omega*z) stack, (x-0.5)*2, out/2+0.5
//   implicit_image/utils/train_helper.py:147-161      F.mse_loss + autograd backward
//   torch.optim.Adam (train_helper.py:72-78)          + Masking.apply_mask (masking/core.py:271-279)
//
// Kernel set (see layout.h for the F-layout / weight-image definitions and DESIGN.md for the roofline
// of each kernel):
//   k_fwd      fused forward chain: coords -> layer 0 (f32 VALU) -> hidden layers on MFMA with the
//              activations resident in registers (accumulator-as-next-B-operand) -> last layer ->
//              residual, SSE partial, dL/dout; spills only the 16-bit PHASE of every sine (F-layout).
//   k_bwd      one layer of the backward pass: data gradient (delta_l -> delta_{l-1}, cos re-derived from the
//              phase) fused with the weight gradient dW_l = delta_l^T * sin(phase_{l-1}) (contraction over
//              pixels, operands read with ds_read_b64_tr_b16 from a DMA-fed LDS ring, 256x256 accumulators).
//   k_dw0      weight gradient of layer 0 against the coordinates.
//   k_reduce   fixed-order reduction of the per-workgroup dW slabs into the flat gradient.
//   k_adam     Adam + mask (torch.optim.Adam op order), k_images: rebuild the 16-bit weight images.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "layout.h"

namespace sf {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define DEV __device__ __forceinline__

struct OpBF16 {
  static DEV uint32_t pack2(float a, float b) {
    f32x2 v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
  }
  static DEV float sum2(uint32_t u) {
    return __builtin_bit_cast(float, u << 16) + __builtin_bit_cast(float, u & 0xffff0000u);
  }
  static DEV float lo(uint32_t u) { return __builtin_bit_cast(float, u << 16); }
  static DEV f32x16 mfma(u32x4 a, u32x4 b, f32x16 c) {
#ifdef SF_EXPERIMENT_NO_MFMA   // timing-only build: operands stay live, the matrix pipe stays idle
    asm volatile("" ::"v"(a), "v"(b));
    return c;
#endif
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0,
                                                   0, 0);
  }
};
struct OpF16 {
  static DEV uint32_t pack2(float a, float b) {
    f32x2 v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2));
  }
  static DEV float sum2(uint32_t u) {   // lo + hi element of a packed pair, in fp32
    const f16x2 v = __builtin_bit_cast(f16x2, u);
    return (float)v[0] + (float)v[1];
  }
  static DEV float lo(uint32_t u) { return (float)__builtin_bit_cast(f16x2, u)[0]; }
  static DEV f32x16 mfma(u32x4 a, u32x4 b, f32x16 c) {
#ifdef SF_EXPERIMENT_NO_MFMA
    asm volatile("" ::"v"(a), "v"(b));
    return c;
#endif
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0,
                                                  0);
  }
};

// streaming store of one 16-byte piece element (scratch tensors are written once and read once much later)
DEV void store_stream(u32x4* p, u32x4 v) {
#ifdef SF_EXPERIMENT_NO_STORE   // timing-only build
  asm volatile("" ::"v"(v));
  return;
#endif
#ifdef SF_EXPERIMENT_NT_STORE
  __builtin_nontemporal_store(v, p);
#else
  *p = v;
#endif
}

DEV uint32_t pack_phase2(float f0, float f1) {  // two fractions in [0,1) -> two unorm16 (x*65535, RNE)
  return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pknorm_u16(f0, f1));
}
DEV float bf16_lo(uint32_t u) { return __builtin_bit_cast(float, u << 16); }
DEV float bf16_hi(uint32_t u) { return __builtin_bit_cast(float, u & 0xffff0000u); }

constexpr float kInv65535 = 1.0f / 65535.0f;

// ---- 8-bit scratch formats (sf_config.scratch_format = 8) ------------------------------------------------------
// Phase byte: u = round(t * 256) mod 256 for a phase t in revolutions, decoded as u / 256 rev (|d sin| <= 1.3e-2,
// zero-mean over the pixels a gradient sums over).  t + 1.5 * 2^15 has ulp 2^-8, so the low mantissa byte of the
// sum IS u (two's complement wrap covers negative t); v_add_f32_sdwa writes that byte straight into byte n of the
// destination and preserves the other three: ONE instruction per value (verified on gfx950:
// scripts/probes/isa_probe.hip).  |t| must stay below 2^14 revolutions.
constexpr float kPhaseMagic = 49152.0f;
// `after` is a value the compiler itself computed FROM t (here sin(t)): it is not used by the instruction, it only
// orders it.  hipcc's hazard recognizer pads the MFMA-result -> VALU-read wait states for instructions it knows, not
// for the operands of an asm statement; if this add were the first reader of a fresh accumulator it would read
// it while the matrix pipe is still writing (seen as run-to-run different phase bytes in k_fwd<256> only).
template <int BYTE>
DEV void phase_byte(uint32_t& dst, float t, float after) {
  static_assert(BYTE >= 0 && BYTE < 4, "byte lane");
  // byte 0 is written first and pads the rest of the register with zeros (no dependence on an initial value)
  if constexpr (BYTE == 0) asm("v_add_f32_sdwa %0, %1, %2 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD" : "=v"(dst) : "v"(t), "v"(kPhaseMagic), "v"(after));
  if constexpr (BYTE == 1) asm("v_add_f32_sdwa %0, %1, %2 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(dst) : "v"(t), "v"(kPhaseMagic), "v"(after));
  if constexpr (BYTE == 2) asm("v_add_f32_sdwa %0, %1, %2 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(dst) : "v"(t), "v"(kPhaseMagic), "v"(after));
  if constexpr (BYTE == 3) asm("v_add_f32_sdwa %0, %1, %2 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(dst) : "v"(t), "v"(kPhaseMagic), "v"(after));
}
DEV uint32_t phase_byte4(const float* t, const float* after) {
  uint32_t r;
  phase_byte<0>(r, t[0], after[0]); phase_byte<1>(r, t[1], after[1]); phase_byte<2>(r, t[2], after[2]); phase_byte<3>(r, t[3], after[3]);
  return r;
}
// decode of byte n: u / 256 + 2^-16 revolutions (v_cvt_f32_ubyteN + one multiply-add).  The offset (1/256 of the byte
// step, 1e-4 rad) keeps a decoded phase off the zeros of sin and cos: a neuron whose phase stays within 1/512
// revolution of 0 for every pixel (no live input) would otherwise have sin = 0 EXACTLY, its outgoing weights an exactly
// zero gradient, and a topology update that ranks |gradient| (grow.py:86-95) could not tell such a candidate from the
// zeros it multiplies into the entries that are not candidates (it regrew 767 of 768 weights of a dense last layer).
#ifndef SF_PHASE_EPS
#define SF_PHASE_EPS (1.0f / 65536.0f)
#endif
constexpr float kPhaseEps = SF_PHASE_EPS;
template <int BYTE>
DEV float phase_rev8(uint32_t p) { return __builtin_fmaf((float)((p >> (8 * BYTE)) & 0xffu), 1.0f / 256.0f, kPhaseEps); }

// Delta byte: OCP fp8 e4m3 (3-bit significand, 2^-9 .. 448), SATURATING (the plain conversion returns NaN beyond
// +-448: isa_probe).  Deltas carry one power-of-two scale per pixel chunk, derived by k_bwd8<LAST> from the chunk's
// own residual so that rms(residual * G) lands in (kFp8Target / 2, kFp8Target]: e4m3's range then sits around the data at
// every stage of a fit (the hidden deltas are 0.1 - 5 x that, growing towards layer 0; 448 / 0.25 leaves the tails room),
// and the consumer's fp8 -> fp16 conversion needs no scale at all (e4m3's range is inside fp16's).
constexpr float kFp8Target = 0.5f;
// (min/max with literals, not v_med3_f32: a VOP3 instruction cannot take a literal, and the two bounds would each
//  occupy a register for the whole kernel)
DEV float sat448(float x) { return __builtin_fmaxf(__builtin_fminf(x, 448.0f), -448.0f); }
DEV uint32_t fp8x4_sat(float a, float b, float c, float d) {
  int r = __builtin_amdgcn_cvt_pk_fp8_f32(sat448(a), sat448(b), 0, false);
  r = __builtin_amdgcn_cvt_pk_fp8_f32(sat448(c), sat448(d), r, true);
  return (uint32_t)r;
}
// 8 fp8 (two dwords) -> 8 fp16 (one B-operand / piece element of 16 bytes)
DEV u32x4 fp8x8_to_f16(uint32_t lo, uint32_t hi) {
  typedef __attribute__((ext_vector_type(2))) _Float16 h2;
  const h2 a = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(lo, 1.0f, false), b = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(lo, 1.0f, true);
  const h2 c = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(hi, 1.0f, false), d = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(hi, 1.0f, true);
  return u32x4{__builtin_bit_cast(uint32_t, a), __builtin_bit_cast(uint32_t, b), __builtin_bit_cast(uint32_t, c),
               __builtin_bit_cast(uint32_t, d)};
}

// Phase decode for v_sin_f32 / v_cos_f32 (which take REVOLUTIONS and are 1-periodic): the unorm16 phase u
// is dropped into the mantissa of a float in [1,2): x = 1 + u/65536 (+ <2^-16 from the neighbour's bits in
// the high-half form).  Two integer ops per value instead of and/shift + cvt + mul; the 65535-vs-65536
// scale (<=1.5e-5 rev) is below the quantisation step of the phase itself.
#ifdef SF_EXPERIMENT_NO_TRANS   // timing-only build: no transcendentals in the backward epilogue
#define __builtin_amdgcn_cosf(x) (x)
#define __builtin_amdgcn_sinf(x) ((x) + 1.0f)
#endif
DEV float phase_rev_lo(uint32_t p) { return __builtin_bit_cast(float, ((p << 7) & 0x007fff80u) | 0x3f800000u); }
DEV float phase_rev_hi(uint32_t p) { return __builtin_bit_cast(float, (p >> 9) | 0x3f800000u); }

// workgroup barriers that do NOT drain in-flight global_load_lds (a plain __syncthreads() would)
DEV void bar_lds() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
DEV void bar_all() {
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
// barrier that waits for this wave's LDS-DMA but leaves its N YOUNGEST vector-memory operations (the
// epilogue stores issued AFTER the DMA) in flight: vmcnt counts loads, stores and LDS-DMA in issue order
template <int N>
DEV void bar_dma() {
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// LDS-DMA of one 1 KiB piece: lane i's 16 bytes at gsrc (per-lane address) -> LDS lds_wave_base + 16 i.
// Issued as inline asm, NOT through __builtin_amdgcn_global_load_lds: hipcc's waitcnt pass knows that the
// builtin writes LDS and, unable to tell which bytes, puts an s_waitcnt vmcnt(0) in front of the next LDS read
// it has a memory operand for (every ds_read_b64_tr_b16 intrinsic) - inside the ring loops that drained the
// prefetch issued a few instructions earlier on every iteration.  The kernels order DMA against LDS reads
// themselves (counted vmcnt + s_barrier in bar_dma / bar_all), which is the only synchronisation wanted.
DEV void glds16(const void* gsrc, char* lds_wave_base) {
#ifdef SF_EXPERIMENT_NO_DMA    // timing-only build: no LDS-DMA traffic (kernels compute on stale LDS)
  return;
#endif
#ifdef SF_EXPERIMENT_BUILTIN_DMA
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
#else
  const uint32_t lds = __builtin_amdgcn_readfirstlane(
      (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds_wave_base);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off"
               :: "s"(lds), "v"(gsrc) : "memory", "m0");
#endif
}
// same, with the address split into a wave-uniform base (SGPR pair) and a per-lane byte offset (one VGPR):
// no 64-bit per-lane address arithmetic, and nothing but the lane offset has to stay live in VGPRs
DEV void glds16s(const void* sbase, uint32_t lane_off_bytes, char* lds_wave_base) {
#ifdef SF_EXPERIMENT_NO_DMA
  return;
#endif
  const uint32_t lds = __builtin_amdgcn_readfirstlane(
      (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds_wave_base);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
               :: "s"(lds), "v"(lane_off_bytes), "s"(sbase) : "memory", "m0");
}


// same, with everything precomputed: wave-uniform image base (SGPR pair), byte offset inside the image (VGPR), LDS byte
// address of the destination piece (SGPR)
DEV void glds16o(const void* sbase, uint32_t off_bytes, uint32_t lds_addr) {
#ifdef SF_EXPERIMENT_NO_DMA
  return;
#endif
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
               :: "s"(lds_addr), "v"(off_bytes), "s"(sbase) : "memory", "m0");
}

// ---------------------------------------------------------------------------------------------
// k_fwd
// ---------------------------------------------------------------------------------------------
struct FwdArgs {
  const float* gh;         // [H] torch.linspace(0,1,H)   (data.py:82)
  const float* gw;         // [W] torch.linspace(0,1,W)   (data.py:83)
  int W;                   // image width
  int row_begin;           // first image row of this handle
  long pix0;               // first local pixel of this chunk
  long npix;               // local pixels of the handle (valid: pix < npix)
  int depth;               // number of Linear layers
  const f32x4* l0tab;      // [WD] {w00, w01, b0, 0}
  const u32x4* l0img;      // k_fwd_pipe: layer 0 as ONE MFMA k-step per 32-neuron tile (WD/32 pieces, see kL0Split)
  int n_super;             // k_fwd_pipe: 256-pixel groups of this chunk (workgroup w takes groups w, w + gridDim.x, ...)
  unsigned long long w_magic;   // k_fwd_pipe: ceil(2^40 / W): row = (p * w_magic) >> 40
  const u32x4* wf;         // (depth-2) forward images (FwdImg<WD>::PIECES pieces each, biases included)
  const u32x4* wf_last;    // forward image of the last layer padded to 32 rows + bias piece
  float sc_first;          // first_omega_0 / (2 pi)
  float sc_hidden;         // unused since the hidden forward images carry omega/(2 pi): the accumulator IS the phase
  float sc_last;           // 1 / weight-image scale
  float last_om, last_om_rev;  // outermost_linear=False (siren.py:110-117): the last layer is sin(omega z) too;
                               // omega and omega/(2 pi), both 0 for the (default) linear last layer
  u32x4* P;                // phases, layer l at P + l*p_stride.  16-bit scratch: unorm16, F-layout (KS pieces per
                           // pixel block); 8-bit scratch: phase bytes, one piece per (pixel block, 32-neuron tile):
                           // byte 8q+j of lane (h,m) = neuron 32*nt + 16*q + PI(h,j)  (NT pieces per pixel block)
  long p_stride;           // pieces per layer in the scratch
  u32x4* Dlast;            // dL/dout, F-layout, 16-bit float: 2 k-steps per block (16-bit scratch; the second is zero)
                           // or 1 k-step per block (8-bit scratch: k_bwd8 keeps the zero k-step in LDS)
  // (8-bit scratch: gscale is the STATIC 2^10 the residual is stored with; k_bwd8<LAST> applies the chunk's
  //  adaptive factor when it forms the first hidden delta)
  const float* img;        // [npix][nout] target
  int nout;                // out_features (1..3): channel count and stride of img / pred
  float gscale;            // 1/(nout*H*W): d(mse)/d(out) = (pred-img) * gscale   (the /2 of siren.py:131 folded in)
  float* pred;             // optional [npix][nout]
  float* sse_part;         // [gridDim.x] per-workgroup sum of squared residuals
  float* dbg;              // SF_EXPERIMENT_STAMP builds only
};

// Forward weight image of one hidden layer, as stored in HBM and copied verbatim into LDS:
//   [half 0: tiles 0..H0-1, KS pieces each][bias piece 0][half 1: tiles H0..NT-1][bias piece 1]
// (piece = 1 KiB = 64 lanes x 16 B; a bias piece holds the 32*tiles fp32 biases of its half, pre-scaled).
// The two halves are the double-buffering unit of k_fwd: while the waves run the MFMAs of one half, the
// other half of the NEXT layer is in flight (global_load_lds).
template <int WD>
struct FwdImg {
  static constexpr int NT = WD / 32, KS = WD / 16;
  static constexpr int H0 = NT >= 2 ? NT / 2 : 1, H1 = NT - H0;
  static constexpr int X_PIECES = H0 * KS + 1, Y_PIECES = H1 * KS + 1;
  static constexpr int PIECES = X_PIECES + Y_PIECES;            // per hidden layer
  static constexpr int LAST_PIECES = KS + 1;                     // last layer: one 32-row tile + bias piece
  static __host__ __device__ constexpr int tile_piece(int nt) { return nt < H0 ? nt * KS : X_PIECES + (nt - H0) * KS; }
  static __host__ __device__ constexpr int bias_piece(int nt) { return nt < H0 ? H0 * KS : X_PIECES + H1 * KS; }
  static __host__ __device__ constexpr int bias_off(int nt) { return (nt < H0 ? nt : nt - H0) * 32; }
};

// last-layer accumulator -> prediction, residual and dL/dout piece of one pixel block; returns this lane's squared residual
template <typename OP, bool TRAIN, bool S8>
DEV float fwd_residual(const FwdArgs& a, const f32x16& acc, const float (&tgt)[3], long pix, long pb, bool valid, int lane, int h) {
  float sse = 0.f;
  float d[3] = {0.f, 0.f, 0.f};
  const float gscale = a.gscale;
  if (h == 0 && valid) {  // rows 0..2 of the tile live in registers 0..2 of the lower lane half
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      if (c >= a.nout) break;   // rows >= out_features of the padded tile carry zero weights: no residual, no delta
      float o = acc[c] * a.sc_last, dfac = 1.0f;
      if (a.last_om_rev != 0.f) {   // sine output layer: d sin(om z)/dz = om cos(om z)
        const float tt = o * a.last_om_rev;
        o = __builtin_amdgcn_sinf(tt);
        dfac = a.last_om * __builtin_amdgcn_cosf(tt);
      }
      const float p = o * 0.5f + 0.5f;  // siren.py:131
      if (a.pred) a.pred[pix * a.nout + c] = p;
      if (a.img) {
        const float r = p - tgt[c];
        sse += r * r;
        d[c] = r * gscale * dfac;
      }
    }
  }
  if (TRAIN && S8) {
    a.Dlast[pb * 64 + lane] = u32x4{OP::pack2(d[0], d[1]), OP::pack2(d[2], 0.f), 0u, 0u};
  } else if (TRAIN) {
    // dL/dout in F-layout (k-step 0: neurons PI(0,j) = j for j < 4), second k-step zero
    a.Dlast[(pb * 2 + 0) * 64 + lane] = u32x4{OP::pack2(d[0], d[1]), OP::pack2(d[2], 0.f), 0u, 0u};
    a.Dlast[(pb * 2 + 1) * 64 + lane] = u32x4{0u, 0u, 0u, 0u};
  }
  return sse;
}
// workgroup SSE partial (fixed order: lanes by xor-shuffle, then waves 0..7); called by all 512 threads
DEV void fwd_sse_partial(const FwdArgs& a, float sse, int lane, int wave, int tid, float* sRed) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sse += __shfl_xor(sse, o);
  if (lane == 0) sRed[wave] = sse;
  __syncthreads();
  if (tid == 0 && a.sse_part) {
    float t = 0.f;
    for (int w = 0; w < kWavesFwd; ++w) t += sRed[w];
    a.sse_part[blockIdx.x] = t;
  }
}

template <int WD, typename OP, bool TRAIN, bool S8 = false>
__global__ __launch_bounds__(512) void k_fwd(FwdArgs a) {
  using IM = FwdImg<WD>;
  constexpr int NT = IM::NT, KS = IM::KS, H0 = IM::H0;
  constexpr int SPT = S8 ? 1 : 2;   // phase stores per tile epilogue (what the counted vmcnt waits leave in flight)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  u32x4* sW = reinterpret_cast<u32x4*>(smem);                                  // IM::PIECES pieces
  f32x4* sL0 = reinterpret_cast<f32x4*>(smem + (size_t)IM::PIECES * 1024);     // WD
  float* sRed = reinterpret_cast<float*>(sL0 + WD);                            // 8

  const int tid = threadIdx.x, lane = tid & 63, m = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // async copy of `n` image pieces (1 KiB each) into LDS pieces [dst, dst+n), spread over the 8 waves
  auto stage = [&](const u32x4* src, int dst, int n) {
    for (int pc = wave; pc < n; pc += kWavesFwd) glds16s(src + pc * 64, (uint32_t)lane * 16u, smem + (size_t)(dst + pc) * 1024);
  };
  for (int i = tid; i < WD; i += 512) sL0[i] = a.l0tab[i];
  const long pb = (long)blockIdx.x * kWavesFwd + wave;  // pixel block inside the chunk
  const long pix = a.pix0 + pb * 32 + m;
  const bool valid = pix < a.npix;
  const long pc = valid ? pix : a.npix - 1;
  const int row = (int)(pc / a.W), col = (int)(pc - (long)row * a.W);
  const float x0 = (a.gh[a.row_begin + row] - 0.5f) * 2.0f;  // siren.py:128
  const float x1 = (a.gw[col] - 0.5f) * 2.0f;
  float tgt[3] = {0.f, 0.f, 0.f};
  if (a.img && h == 0 && valid) {
#pragma unroll
    for (int c = 0; c < 3; ++c) if (c < a.nout) tgt[c] = a.img[pix * a.nout + c];
  }
  // the ordinary loads above are consumed before the first DMA is issued, so the compiler's
  // vmcnt(0) for them does not drain the weight prefetch
  asm volatile("" :: "v"(x0), "v"(x1), "v"(tgt[0]), "v"(tgt[1]), "v"(tgt[2]));
  if (a.depth > 2) stage(a.wf, 0, IM::X_PIECES);
  else stage(a.wf_last, 0, IM::LAST_PIECES);
  asm volatile("" ::: "memory");
  bar_lds();   // layer-0 table visible (does not wait for the weight DMA)

  u32x4 B[KS];
  // ---- layer 0: K = 2, f32 VALU, written straight into B-fragment order -----------------------
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    float av[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const f32x4 t = sL0[16 * s + pi_perm(h, j)];
      const float z = __builtin_fmaf(t.y, x1, __builtin_fmaf(t.x, x0, t.z));
      av[j] = __builtin_amdgcn_sinf(z * a.sc_first);
    }
    B[s] = u32x4{OP::pack2(av[0], av[1]), OP::pack2(av[2], av[3]), OP::pack2(av[4], av[5]), OP::pack2(av[6], av[7])};
    // layer-0 phases are NOT spilled: k_bwd re-derives them from the coordinates (2 FMAs per value)
  }

  // One 32-neuron output tile = bias-initialised accumulator + KS MFMAs (tile_mma), then the sine epilogue
  // (tile_epi).  The two are software-pipelined: the epilogue of tile k-1 (pure VALU + 2 stores) is issued
  // in the shadow of the MFMAs of tile k: per MFMA ~4 VALU (mul, fract, sin, half a pack + half a pknorm).
  auto tile_mma = [&](int nt) -> f32x16 {
    f32x16 acc;
    const float* bias = reinterpret_cast<const float*>(sW + IM::bias_piece(nt) * 64) + IM::bias_off(nt);
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      const f32x4 b = *reinterpret_cast<const f32x4*>(&bias[8 * q4 + 4 * h]);
      acc[4 * q4 + 0] = b.x; acc[4 * q4 + 1] = b.y; acc[4 * q4 + 2] = b.z; acc[4 * q4 + 3] = b.w;
    }
    const u32x4* wt = sW + IM::tile_piece(nt) * 64 + lane;
#pragma unroll
#ifdef SF_EXPERIMENT_FWD_LDS1   // timing-only build: one LDS fragment read per tile instead of KS
    for (int s = 0; s < KS; ++s) acc = OP::mfma(wt[0], B[s], acc);
#else
    for (int s = 0; s < KS; ++s) acc = OP::mfma(wt[s * 64], B[s], acc);
#endif
    return acc;
  };
  auto tile_epi = [&](const f32x16& acc, int nt, int l, u32x4* Bn) {
    if constexpr (S8) {   // 8-bit scratch: one 16-byte phase piece element per tile (both k-steps), 1 VALU per byte
      u32x4 pb8;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        float av[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) av[j] = __builtin_amdgcn_sinf(acc[8 * q + j]);
        Bn[2 * nt + q] =
            u32x4{OP::pack2(av[0], av[1]), OP::pack2(av[2], av[3]), OP::pack2(av[4], av[5]), OP::pack2(av[6], av[7])};
        if (TRAIN) {
          float tv[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) tv[j] = acc[8 * q + j];
          pb8[2 * q] = phase_byte4(tv, av);
          pb8[2 * q + 1] = phase_byte4(tv + 4, av + 4);
        }
      }
      if (TRAIN) store_stream(&a.P[(size_t)l * a.p_stride + (pb * NT + nt) * 64 + lane], pb8);
      return;
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      float av[8], ph[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float tt = acc[8 * q + j];   // revolutions: omega/(2 pi) is folded into the weight image and the bias
        av[j] = __builtin_amdgcn_sinf(tt);
        ph[j] = __builtin_amdgcn_fractf(tt);
      }
      Bn[2 * nt + q] =
          u32x4{OP::pack2(av[0], av[1]), OP::pack2(av[2], av[3]), OP::pack2(av[4], av[5]), OP::pack2(av[6], av[7])};
      if (TRAIN)
        store_stream(&a.P[(size_t)l * a.p_stride + (pb * KS + 2 * nt + q) * 64 + lane],
                     u32x4{pack_phase2(ph[0], ph[1]), pack_phase2(ph[2], ph[3]), pack_phase2(ph[4], ph[5]),
                           pack_phase2(ph[6], ph[7])});
    }
  };
#ifdef SF_EXPERIMENT_STAMP
  unsigned long long st_bar1 = 0, st_bar2 = 0;
  const unsigned long long st_t0 = __builtin_amdgcn_s_memtime();
#endif
  // ---- hidden layers: [WD x WD] on MFMA, activations stay in registers ------------------------
  for (int l = 1; l <= a.depth - 2; ++l) {
    const u32x4* img_l = a.wf + (size_t)(l - 1) * IM::PIECES * 64;
    u32x4 Bn[KS];
    // half X of layer l landed (the epilogue stores issued after that DMA may stay in flight);
    // everyone left half Y of layer l-1
#ifdef SF_EXPERIMENT_STAMP
    unsigned long long t_a = __builtin_amdgcn_s_memtime();
#endif
    if (l == 1) bar_dma<0>(); else bar_dma<TRAIN ? SPT * (IM::H1 + 1) : 0>();
#ifdef SF_EXPERIMENT_STAMP
    { const unsigned long long t_b = __builtin_amdgcn_s_memtime(); st_bar1 += t_b - t_a; }
#endif
    stage(img_l + IM::X_PIECES * 64, IM::X_PIECES, IM::Y_PIECES);
    asm volatile("" ::: "memory");
    f32x16 prev = tile_mma(0);
#pragma unroll
    for (int nt = 1; nt < H0; ++nt) {
      const f32x16 cur = tile_mma(nt);
      tile_epi(prev, nt - 1, l, Bn);
      __builtin_amdgcn_sched_barrier(0);   // keep every epilogue next to its MFMA tile (no deferred sin blobs)
      prev = cur;
    }
#ifdef SF_EXPERIMENT_STAMP
    t_a = __builtin_amdgcn_s_memtime();
#endif
    bar_dma<TRAIN ? SPT * (H0 - 1) : 0>();          // half Y landed; everyone left half X
#ifdef SF_EXPERIMENT_STAMP
    { const unsigned long long t_b = __builtin_amdgcn_s_memtime(); st_bar2 += t_b - t_a; }
#endif
    if (l < a.depth - 2) stage(img_l + IM::PIECES * 64, 0, IM::X_PIECES);
    else stage(a.wf_last, 0, IM::LAST_PIECES);
    asm volatile("" ::: "memory");
#pragma unroll
    for (int nt = H0; nt < NT; ++nt) {
      const f32x16 cur = tile_mma(nt);
      tile_epi(prev, nt - 1, l, Bn);
      __builtin_amdgcn_sched_barrier(0);
      prev = cur;
    }
    tile_epi(prev, NT - 1, l, Bn);
#pragma unroll
    for (int s = 0; s < KS; ++s) B[s] = Bn[s];
  }

#ifdef SF_EXPERIMENT_STAMP
  if (a.dbg && lane == 0 && (wave == 0 || wave == 5) && (blockIdx.x == 3 || blockIdx.x == 9000)) {
    float* o = a.dbg + ((blockIdx.x == 3 ? 0 : 2) + (wave == 0 ? 0 : 1)) * 4;
    o[0] = (float)(__builtin_amdgcn_s_memtime() - st_t0); o[1] = (float)st_bar1; o[2] = (float)st_bar2; o[3] = (float)(a.depth - 2);
  }
#endif
  // ---- last layer (out_features <= 3, padded to one 32-row tile) + residual ---------------------
  if (a.depth > 2) bar_dma<TRAIN ? SPT * (IM::H1 + 1) : 0>(); else bar_dma<0>();
  f32x16 acc;
  {
    const float* bias = reinterpret_cast<const float*>(sW + KS * 64);
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      const f32x4 b = *reinterpret_cast<const f32x4*>(&bias[8 * q4 + 4 * h]);
      acc[4 * q4 + 0] = b.x; acc[4 * q4 + 1] = b.y; acc[4 * q4 + 2] = b.z; acc[4 * q4 + 3] = b.w;
    }
  }
#pragma unroll
  for (int s = 0; s < KS; ++s) acc = OP::mfma(sW[s * 64 + lane], B[s], acc);

  fwd_sse_partial(a, fwd_residual<OP, TRAIN, S8>(a, acc, tgt, pix, pb, valid, lane, h), lane, wave, tid, sRed);
}

// ---------------------------------------------------------------------------------------------
// k_fwd_pipe: k_fwd for hidden = 256 and depth >= 3 as ONE software pipeline over all hidden layers.
//
// k_fwd leaves the order of a tile's 16 MFMAs, its 16 weight-fragment reads and the previous tile's sine epilogue to
// hipcc, which reads two fragments, waits for them, issues two MFMAs (ten times per tile, LDS latency exposed every
// time) and packs the whole epilogue behind the last six: the matrix pipe was busy 54 % of the time.  Here the order is
// fixed by hand, one "slot" per MFMA, and nothing crosses a slot boundary (sched_barrier):
//
//   slot g of a layer (g = 16 * tile + k-step):   read fragment g + PD (LDS -> registers, PD slots ahead)
//                                                 MFMA g
//                                                 1-2 values of the previous tile's epilogue (sin, phase byte, pack)
//                                                 [slot 12 of a tile: the next tile's bias = its accumulator]
//
// The pipeline never drains: the fragment ring, the bias-initialised accumulator and the epilogue of a layer's LAST tile
// carry over into the first tile of the next layer (that epilogue produces k-steps 14 and 15 of the activations, which
// tile 0 needs at its slots 14 and 15 only, so it runs in slots 1..13), and finally into the 16 MFMAs of the output
// layer.  Layer 0 enters the same way: its last 32 neurons are handed over as phases in `prev`, and the generic epilogue
// takes their sines (their phase bytes go to the unused layer-0 plane of the scratch, which keeps the store counts of
// the vmcnt waits uniform).  The two workgroup barriers per layer sit where the READS move from one half of the LDS
// weight image to the other (PD slots before the MFMAs do), so the wave still holds PD ready fragments after a barrier.
// Activations ping-pong between two register arrays (the layer loop is unrolled by two): no copies.
// The compiler still places every s_waitcnt lgkmcnt itself, from the fixed order.
// ---------------------------------------------------------------------------------------------
// Layer 0 on the matrix pipe (k_fwd_pipe).  z = (w0 x0 + w1 x1 + b) * first_omega_0 / (2 pi) needs f32 accuracy (the
// coordinates alone carry 12+ bits), so both factors are split into two 16-bit floats, a = ah + al, and the k-step sums
// the products that matter, each pair scaled by a power of two so that its small factor stays a normal number:
//   k:   0        1          2          3        4          5          6      7
//   A:   w0h      w0h / S    w0l * S    w1h      w1h / S    w1l * S    bh     bl * S        (lane half 0; half 1: zeros)
//   B:   x0h      x0l * S    x0h / S    x1h      x1l * S    x1h / S    1      1 / S
// (w, b pre-multiplied by omega/(2 pi)); the dropped wl*xl terms are < 2^-22 of |w x|.  Always fp16, also when the
// hidden layers run on bf16 operands.
constexpr float kL0Split = 64.0f;
DEV void split_f16(float x, float lo_scale, _Float16& hi, _Float16& lo) {
  hi = (_Float16)x;
  lo = (_Float16)((x - (float)hi) * lo_scale);
}
DEV uint32_t pack_h2(_Float16 a, _Float16 b) {
  typedef __attribute__((ext_vector_type(2))) _Float16 h2;
  return __builtin_bit_cast(uint32_t, h2{a, b});
}

#ifndef SF_FWD_PD
#define SF_FWD_PD 4
#endif
template <typename OP, bool TRAIN, bool S8, int PD = SF_FWD_PD>
__global__ __launch_bounds__(512) void k_fwd_pipe(FwdArgs a) {
  constexpr int WD = 256;
  using IM = FwdImg<WD>;
  constexpr int NT = IM::NT, KS = IM::KS, H0 = IM::H0, NG = NT * KS;
  constexpr int SPT = S8 ? 1 : 2;                       // phase stores per tile epilogue
  constexpr int ST_SLOT = KS - 3;                       // slot of a tile in which the previous tile's phases are stored
  // The weight DMA is spread out: a wave issues its (up to) 9 pieces of a half image one every DSP slots, starting
  // DMA0 slots into the half that follows the barrier which freed the LDS region (a burst of 8-9 global_load_lds
  // right behind the barrier cost each wave ~200 cycles per piece with both waves of a SIMD in it at once: the whole
  // staging was 25 % of the kernel).  NST = phase stores a wave issues between its last piece and the barrier that
  // waits for the image (slots 13 and 29 of a half lie behind the last DMA slot 26; slot 45 .. behind the barrier).
  constexpr int DMA0 = 2, DSP = 3, NPC = 9;
  constexpr int NST = TRAIN ? SPT * 2 : 0;
  static_assert(DMA0 + DSP * (NPC - 1) < ST_SLOT + KS && DMA0 + DSP * (NPC - 1) > ST_SLOT && (ST_SLOT - DMA0) % DSP != 0 &&
                NPC * kWavesFwd >= IM::X_PIECES && NPC * kWavesFwd >= IM::Y_PIECES, "DMA slot plan");
  static_assert(2 * H0 == NT && PD >= 3 && PD <= 8 && KS == 16, "slot plan: barrier slot KS - PD <= bias slot KS - 4 < store slot KS - 3");
#ifdef SF_EXPERIMENT_STAMP
  const unsigned long long st_entry = __builtin_amdgcn_s_memtime();
  unsigned long long st_mid = 0, st_end = 0, st_l0 = 0, st_x1 = 0, st_pipe = 0;
#endif
  extern __shared__ __attribute__((aligned(16))) char smem[];
  u32x4* sW = reinterpret_cast<u32x4*>(smem);
  u32x4* sL0 = reinterpret_cast<u32x4*>(smem + (size_t)IM::PIECES * 1024);     // layer-0 image: NT pieces
  float* sRed = reinterpret_cast<float*>(sL0 + NT * 64);
  static_assert(NT == kWavesFwd, "one layer-0 piece per wave");

  const int tid = threadIdx.x, lane = tid & 63, m = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int L = a.depth - 2;                      // hidden layers (>= 1: the host sends depth 2 to k_fwd)
  glds16s(a.l0img + wave * 64, (uint32_t)lane * 16u, reinterpret_cast<char*>(sL0) + (size_t)wave * 1024);
  asm volatile("" ::: "memory");

  // The workgroup is PERSISTENT: it walks the 256-pixel groups blockIdx.x, blockIdx.x + gridDim.x, ... of the chunk, so
  // that everything a group needs before its first MFMA overlaps the previous group's pipeline: its coordinates and
  // target are fetched one group ahead (raw values, five registers), the layer-0 image is staged once, and the SSE is
  // reduced once at the end.  (One workgroup per group spent 13 % of its life before the first and after the last MFMA,
  // with nothing else resident on the CU: the LDS holds one weight image.)
  long pb = 0, pix = 0;                           // pixel block of this wave / pixel of this lane in the current group
  bool valid = false;
  struct Fetch { float gh, gw, t[3]; };
  auto fetch = [&](int grp) -> Fetch {           // raw grid values and target of this lane's pixel in group grp
    Fetch f{0.f, 0.f, {0.f, 0.f, 0.f}};
    const long px = a.pix0 + ((long)grp * kWavesFwd + wave) * 32 + m;
    const bool ok = px < a.npix;
    const long pcl = ok ? px : a.npix - 1;
    const unsigned row = (unsigned)(((unsigned long long)pcl * a.w_magic) >> 40);
    const unsigned col = (unsigned)(pcl - (long)row * a.W);
    f.gh = a.gh[a.row_begin + (int)row];
    f.gw = a.gw[col];
    if (a.img && h == 0 && ok) {
#pragma unroll
      for (int c = 0; c < 3; ++c) if (c < a.nout) f.t[c] = a.img[px * a.nout + c];
    }
    return f;
  };
  Fetch nxt_f = fetch((int)blockIdx.x);
  float sse_acc = 0.f;
  bar_dma<0>();                                   // layer-0 image staged (and the first group's fetch has landed)

  auto slot_end = [&]() {   // nothing moves across: neither memory operations (compiler) nor instructions (scheduler)
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  // Explicit LDS addressing: five per-lane base registers, everything else is the 16-bit immediate of the read.  (Left
  // to itself hipcc gave every piece beyond the first 64 KiB its own loop-invariant address register - 56 of them - and
  // spilled what the pipeline needs.)  The empty asm makes a base opaque, so constants are not folded back into it.
  typedef __attribute__((address_space(3))) const u32x4 lds_cv4;
  typedef __attribute__((address_space(3))) const f32x4 lds_cf4;
  __builtin_assume(wave >= 0 && wave < kWavesFwd);
  const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem);
  const uint32_t lane16 = (uint32_t)lane * 16u;
  const uint32_t lds_wave = lds0 + (uint32_t)wave * 1024u;         // LDS-DMA: destination of piece `wave`
  uint32_t aXf = lds0 + lane16;                                               // fragments of half X / the output layer
  uint32_t aYf = lds0 + (uint32_t)IM::X_PIECES * 1024u + lane16;              // fragments of half Y
  uint32_t aXb = lds0 + (uint32_t)IM::bias_piece(0) * 1024u + (uint32_t)h * 16u;     // bias piece of half X
  uint32_t aYb = lds0 + (uint32_t)IM::bias_piece(H0) * 1024u + (uint32_t)h * 16u;    // bias piece of half Y
  uint32_t aLb = lds0 + (uint32_t)KS * 1024u + (uint32_t)h * 16u;                    // bias piece of the output layer
  uint32_t aL0 = lds0 + (uint32_t)IM::PIECES * 1024u + lane16;                       // fragments of the layer-0 image
  asm volatile("" : "+v"(aXf), "+v"(aYf), "+v"(aXb), "+v"(aYb), "+v"(aLb), "+v"(aL0));
  auto frag = [&](int nt, int s2) -> u32x4 {   // fragment (tile, k-step) of the hidden image in LDS
    return nt < H0 ? *(lds_cv4*)(uintptr_t)(aXf + (uint32_t)(nt * KS + s2) * 1024u)
                   : *(lds_cv4*)(uintptr_t)(aYf + (uint32_t)((nt - H0) * KS + s2) * 1024u);
  };
  auto tile_bias = [&](uint32_t ab, int q4, f32x16& acc) {
    const f32x4 b = *(lds_cf4*)(uintptr_t)(ab + (uint32_t)q4 * 32u);
    acc[4 * q4 + 0] = b.x; acc[4 * q4 + 1] = b.y; acc[4 * q4 + 2] = b.z; acc[4 * q4 + 3] = b.w;
  };
  auto hidden_bias = [&](int nt) -> uint32_t { return (nt < H0 ? aXb : aYb) + (uint32_t)IM::bias_off(nt) * 4u; };

  u32x4 Ba[KS], Bb[KS];
  f32x16 prev, cur, acc;
  u32x4 fr[PD];                                            // fragments 0..PD-1 of the upcoming tile 0
  // ---- layer 0 of the current group: one fp16 MFMA per tile (phases in revolutions), sines of tiles 0..6 -> k-steps
  // 0..13 of Ba; the last tile is handed to the pipeline as phases in `prev` (its sines are taken by the first slots of
  // layer 1).  Half X of layer 1 is requested on the way, one or two LDS-DMA pieces per tile (everyone left that LDS
  // region at the barrier in the previous group's output layer).
  auto layer0 = [&](float x0, float x1) {
    u32x4 bx = u32x4{0u, 0u, 0u, 0u};      // B operand (see kL0Split)
    {
      _Float16 x0h, x0l, x1h, x1l;
      split_f16(x0, kL0Split, x0h, x0l);
      split_f16(x1, kL0Split, x1h, x1l);
      const _Float16 x0s = (_Float16)((float)x0h * (1.0f / kL0Split)), x1s = (_Float16)((float)x1h * (1.0f / kL0Split));
      if (h == 0) bx = u32x4{pack_h2(x0h, x0l), pack_h2(x0s, x1h), pack_h2(x1l, x1s), pack_h2((_Float16)1.0f, (_Float16)(1.0f / kL0Split))};
    }
    auto l0_tile = [&](int nt) {
      return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, *(lds_cv4*)(uintptr_t)(aL0 + (uint32_t)nt * 1024u)),
                                                    __builtin_bit_cast(f16x8, bx), f32x16{}, 0, 0, 0);
    };
    auto x1_piece = [&](int i) {
      const int pc = wave + kWavesFwd * i;
      if (pc < IM::X_PIECES) glds16o(a.wf + (size_t)pc * 64, lane16, lds_wave + (uint32_t)(kWavesFwd * i) * 1024u);
    };
    f32x16 z = l0_tile(0);
#pragma unroll
    for (int nt = 1; nt < NT; ++nt) {
      const f32x16 zn = l0_tile(nt);
      if (nt <= 2) { x1_piece(2 * nt - 2); x1_piece(2 * nt - 1); } else x1_piece(nt + 1);
      uint32_t pb8[4];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        uint32_t w[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float s0 = __builtin_amdgcn_sinf(z[8 * q + 2 * j]), s1 = __builtin_amdgcn_sinf(z[8 * q + 2 * j + 1]);
          if constexpr (TRAIN && S8) {
            // phase bytes of layer 0 (round 3): the backward of layer 1 reads them like any hidden layer's, so it runs
            // the hidden-layer kernel (k_bwd8h) instead of the form that re-derives these phases from the coordinates
            if (j & 1) { phase_byte<2>(pb8[2 * q + (j >> 1)], z[8 * q + 2 * j], s0); phase_byte<3>(pb8[2 * q + (j >> 1)], z[8 * q + 2 * j + 1], s1); }
            else { phase_byte<0>(pb8[2 * q + (j >> 1)], z[8 * q + 2 * j], s0); phase_byte<1>(pb8[2 * q + (j >> 1)], z[8 * q + 2 * j + 1], s1); }
          }
          w[j] = OP::pack2(s0, s1);
          asm volatile("" : "+v"(w[j]));   // pinned to this tile (MachineSink would carry the phases down to their first use)
        }
        Ba[2 * (nt - 1) + q] = u32x4{w[0], w[1], w[2], w[3]};
      }
      if constexpr (TRAIN && S8) store_stream(a.P + ((size_t)pb * NT + (nt - 1)) * 64 + lane, u32x4{pb8[0], pb8[1], pb8[2], pb8[3]});
      slot_end();
      z = zn;
    }
    prev = z;
  };
  static_assert(NT + 1 == 9 && 9 * kWavesFwd >= IM::X_PIECES, "layer 0 issues pieces 0..8 of half X");

  // epilogue values e of accumulator `acc` -> sines into the two k-steps dst[0], dst[1] and phases into HBM.
  // Slot plan: none in slot 0 (the matrix pipe is still writing acc), two per slot in slots 1..3, one in 4..13.
  struct Epi { float av[16]; float ph[16]; uint32_t pb8[4]; u32x4 pw[2]; };
  auto epi_value = [&](Epi& E, const f32x16& acc, int e, u32x4* dst) {
    const float tt = acc[e];
#ifdef SF_EXPERIMENT_FWD_NO_EPI   // timing-only build: no sine, no phase byte (one pack per two values remains)
    E.av[e] = tt;
    if (e & 1) dst[e >> 3][(e & 7) >> 1] = OP::pack2(E.av[e - 1], E.av[e]);
    E.pb8[e >> 2] = __builtin_bit_cast(uint32_t, tt);
    return;
#endif
    E.av[e] = __builtin_amdgcn_sinf(tt);
    if constexpr (TRAIN && S8) {
      switch (e & 3) {
        case 0: phase_byte<0>(E.pb8[e >> 2], tt, E.av[e]); break;
        case 1: phase_byte<1>(E.pb8[e >> 2], tt, E.av[e]); break;
        case 2: phase_byte<2>(E.pb8[e >> 2], tt, E.av[e]); break;
        default: phase_byte<3>(E.pb8[e >> 2], tt, E.av[e]); break;
      }
    } else if constexpr (TRAIN) {
      E.ph[e] = __builtin_amdgcn_fractf(tt);
    }
    if (e & 1) {
      // the empty asm pins the pack to its slot: MachineSink would otherwise move it (and the two f32 sines it
      // frees) down to the next layer's first use, across every slot boundary
      uint32_t w = OP::pack2(E.av[e - 1], E.av[e]);
      asm volatile("" : "+v"(w));
      dst[e >> 3][(e & 7) >> 1] = w;
      if constexpr (TRAIN && !S8) {
        uint32_t p2 = pack_phase2(E.ph[e - 1], E.ph[e]);
        asm volatile("" : "+v"(p2));
        E.pw[e >> 3][(e & 7) >> 1] = p2;
      }
    }
  };
  auto epi_slot = [&](Epi& E, const f32x16& acc, int s, u32x4* dst, u32x4* ptile) {
    if (s >= 1 && s <= 3) { epi_value(E, acc, 2 * s - 2, dst); epi_value(E, acc, 2 * s - 1, dst); }
    if (s >= 4 && s <= 13) epi_value(E, acc, s + 2, dst);
    if (TRAIN && s == ST_SLOT) {
      if constexpr (S8) store_stream(ptile, u32x4{E.pb8[0], E.pb8[1], E.pb8[2], E.pb8[3]});
      else { store_stream(ptile, E.pw[0]); store_stream(ptile + 64, E.pw[1]); }
    }
  };
  constexpr int PPT = S8 ? 1 : 2;                         // phase pieces per (pixel block, tile)

  // one hidden layer l: reads activations Bi, writes Bo (k-steps 0..13; 14 and 15 follow in the next stage)
  auto layer = [&](u32x4 (&Bi)[KS], u32x4 (&Bo)[KS], int l) {
    const bool more = l < L;
    const u32x4* srcY = a.wf + ((size_t)(l - 1) * IM::PIECES + IM::X_PIECES) * 64;     // half Y of this layer
    const u32x4* srcX = more ? a.wf + (size_t)l * IM::PIECES * 64 : a.wf_last;        // half X of layer l + 1 / output layer
    const int nX = more ? IM::X_PIECES : IM::LAST_PIECES;
    u32x4* pl = a.P + (size_t)l * a.p_stride + (size_t)pb * (NT * PPT) * 64 + lane;   // phases of layer l, this pixel block
    const uint32_t bias_next0 = more ? hidden_bias(0) : aLb;
    u32x4 ring[NG + PD];
#pragma unroll
    for (int i = 0; i < PD; ++i) ring[i] = fr[i];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      Epi E;
      f32x16 nxt;
      u32x4* dst = nt == 0 ? &Bi[KS - 2] : &Bo[2 * (nt - 1)];
      u32x4* ptile = nt == 0 ? pl - a.p_stride + (NT - 1) * PPT * 64 : pl + (nt - 1) * PPT * 64;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const int g = nt * KS + s;
        if (g == NG / 2 - PD) {        // every read of half X is issued: half Y landed, everyone left half X
#ifdef SF_EXPERIMENT_STAMP
          const unsigned long long t_a = __builtin_amdgcn_s_memtime();
#endif
#ifndef SF_EXPERIMENT_FWD_NO_BAR
          bar_dma<NST>();
#endif
#ifdef SF_EXPERIMENT_STAMP
          st_mid += __builtin_amdgcn_s_memtime() - t_a;
#endif
        }
        if (g == NG - PD) {            // every read of half Y is issued: the next layer's half X landed
#ifdef SF_EXPERIMENT_STAMP
          const unsigned long long t_a = __builtin_amdgcn_s_memtime();
#endif
#ifndef SF_EXPERIMENT_FWD_NO_BAR
          bar_dma<NST>();
#endif
#ifdef SF_EXPERIMENT_STAMP
          st_end += __builtin_amdgcn_s_memtime() - t_a;
#endif
        }
        {   // one piece of the image that goes into the LDS half everyone left at the last barrier
          const int r = g % (NG / 2);          // slot within this half of the layer
          if (r >= DMA0 && (r - DMA0) % DSP == 0 && (r - DMA0) / DSP < NPC) {
            // piece wave + 8 i: the piece offset goes into the SGPR address pair and into M0 (SALU only); the one VGPR
            // operand is the lane offset every LDS access of the kernel already uses
            const int i8 = kWavesFwd * ((r - DMA0) / DSP), pc = wave + i8;
            if (g < NG / 2) {
              if (pc < IM::Y_PIECES) glds16o(srcY + (size_t)pc * 64, lane16, lds_wave + (uint32_t)(IM::X_PIECES + i8) * 1024u);
            } else {
              if (pc < nX) glds16o(srcX + (size_t)pc * 64, lane16, lds_wave + (uint32_t)i8 * 1024u);
            }
          }
        }
        const int f = g + PD;
        ring[f] = f < NG ? frag(f / KS, f % KS) : frag(0, f - NG);    // (f >= NG: tile 0 of the next layer / the output layer)
        cur = OP::mfma(ring[g], Bi[s], cur);
        epi_slot(E, prev, s, dst, ptile);
        if (s == KS - 4) {   // the next tile's bias = its accumulator: right behind the barrier slot, four MFMAs before its use
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4) tile_bias(nt + 1 < NT ? hidden_bias(nt + 1) : bias_next0, q4, nxt);
        }
        slot_end();
      }
      prev = cur;
      cur = nxt;
    }
#pragma unroll
    for (int i = 0; i < PD; ++i) fr[i] = ring[NG + i];
  };
  // output layer (one 32-row tile) under the epilogue of the last hidden tile
  auto output_layer = [&](u32x4 (&Bi)[KS]) {
    u32x4* ptile = a.P + (size_t)L * a.p_stride + ((size_t)pb * NT + (NT - 1)) * PPT * 64 + lane;
    u32x4 ring[KS + PD];
    Epi E;
#pragma unroll
    for (int i = 0; i < PD; ++i) ring[i] = fr[i];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      if (s + PD < KS) ring[s + PD] = frag(0, s + PD);
      if (s == KS - PD) bar_lds();     // every read of the output image is issued: the next group may overwrite half X
      cur = OP::mfma(ring[s], Bi[s], cur);
      epi_slot(E, prev, s, &Bi[KS - 2], ptile);
      slot_end();
    }
    acc = cur;
  };
  for (int grp = (int)blockIdx.x; grp < a.n_super; grp += (int)gridDim.x) {
    // this group's coordinates and target (fetched during the previous group)
    pb = (long)grp * kWavesFwd + wave;
    pix = a.pix0 + pb * 32 + m;
    valid = pix < a.npix;
    const float x0 = (nxt_f.gh - 0.5f) * 2.0f, x1 = (nxt_f.gw - 0.5f) * 2.0f;   // siren.py:128
    const float tgt[3] = {nxt_f.t[0], nxt_f.t[1], nxt_f.t[2]};
#ifdef SF_EXPERIMENT_STAMP
    const unsigned long long t_g = __builtin_amdgcn_s_memtime();
#endif
    layer0(x0, x1);
#ifdef SF_EXPERIMENT_STAMP
    const unsigned long long t_l0 = __builtin_amdgcn_s_memtime();
    st_l0 += t_l0 - t_g;
#endif
    bar_dma<0>();                                            // half X of layer 1 landed
#ifdef SF_EXPERIMENT_STAMP
    const unsigned long long t_x1 = __builtin_amdgcn_s_memtime();
    st_x1 += t_x1 - t_l0;
#endif
    if (grp + (int)gridDim.x < a.n_super) nxt_f = fetch(grp + (int)gridDim.x);   // lands under the pipeline
#pragma unroll
    for (int i = 0; i < PD; ++i) fr[i] = frag(0, i);
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) tile_bias(hidden_bias(0), q4, cur);
    slot_end();
    // (no exit between the two layers of a pair: the live ring would have to be spilled for it on every iteration)
    int l = 1;
    for (; l + 1 <= L; l += 2) {
      layer(Ba, Bb, l);
      layer(Bb, Ba, l + 1);
    }
    if (l == L) {
      layer(Ba, Bb, l);
      output_layer(Bb);
    } else {
      output_layer(Ba);
    }
#ifdef SF_EXPERIMENT_STAMP
    st_pipe += __builtin_amdgcn_s_memtime() - t_x1;
#endif
    sse_acc += fwd_residual<OP, TRAIN, S8>(a, acc, tgt, pix, pb, valid, lane, h);
  }
  fwd_sse_partial(a, sse_acc, lane, wave, tid, sRed);
#ifdef SF_EXPERIMENT_STAMP
  if (a.dbg && lane == 0 && (wave == 0 || wave == 5) && (blockIdx.x == 3 || blockIdx.x == 200)) {
    float* o = a.dbg + ((blockIdx.x == 3 ? 0 : 2) + (wave == 0 ? 0 : 1)) * 8;
    const unsigned long long st_exit = __builtin_amdgcn_s_memtime();
    const float ng = (float)((a.n_super - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x);
    o[0] = (float)st_l0 / ng; o[1] = (float)st_x1 / ng; o[2] = (float)st_pipe / ng; o[3] = (float)st_mid / ng;
    o[4] = (float)st_end / ng; o[5] = 0.f; o[6] = (float)(st_exit - st_entry) / ng; o[7] = ng;
  }
#endif
}

// ---------------------------------------------------------------------------------------------
// transposed LDS reads of F-layout blocks (operands of the pixel-contraction products)
// ---------------------------------------------------------------------------------------------
// address (in bytes, relative to an F-layout block of KS k-steps) that lane must supply to
// ds_read_b64_tr_b16 for fragment tile `tile` (32 neurons), pixel k-step kk (16 pixels), half-read u.
// Returned fragment: lane (r = lane&31 -> neuron 32*tile + r, hq = lane>>5), elements = pixels
// 16*kk + 8*hq + 4*u + {0..3}.
// Bank swizzle of every LDS block that is read transposed.  A piece copied verbatim (slot = register-image
// lane h*32 + m) puts the four 64-byte regions a 32-lane half of ds_read_b64_tr_b16 touches (two k-step pieces
// x two lane halves h) 512 / 1024 bytes apart, i.e. on the SAME 16 of the 64 banks: a 4-way conflict on every
// transposed read (SQ_LDS_BANK_CONFLICT = half of SQ_LDS_IDX_ACTIVE in the unswizzled k_bwd).  Stored at slot
// h*32 + (m ^ 4h ^ 8*(k-step parity)) the four regions land on the four different bank quarters, and the
// lane-linear ds_read_b128 of the same piece stays conflict-free (an XOR by a multiple of 4 only permutes the
// 4-lane clusters inside each of the instruction's lane groups).  global_load_lds writes LDS lane-linearly but
// takes a per-lane GLOBAL address, so the permutation costs nothing: LDS slot i fetches global slot sw_lane(i).
DEV int sw_lane(int l, int parity) { return l ^ ((l >> 5) << 2) ^ (parity << 3); }

DEV int tr_addr(int tile, int kk, int u, int lane) {
  const int g = lane >> 4, gs = g & 1, hq = g >> 1, li = lane & 15, q = li >> 2, p = li & 3;
  const int mpix = 16 * kk + 8 * hq + 4 * u + q;
  const int piece = (2 * tile + gs) * 64 + (p & 1) * 32 + (mpix ^ ((p & 1) << 2) ^ (gs << 3));
  return piece * 16 + 8 * (p >> 1);
}
// lane part of tr_addr (tile 0, k-step 0, half-read 0); tile adds 2048, k-step 256, and half-read 1 is the
// same address with bit 6 flipped (the swizzle XORs the bit the half-read index would otherwise add)
DEV int tr_lane_base(int lane) { return tr_addr(0, 0, 0, lane); }

DEV u32x4 ds_read_tr_pair(const char* base, int off0, int off1) {
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const s16x4 r0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + off0));
  const s16x4 r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + off1));
  const u32x2 a0 = __builtin_bit_cast(u32x2, r0), a1 = __builtin_bit_cast(u32x2, r1);
  return u32x4{a0.x, a0.y, a1.x, a1.y};
}
// fragment (tile, kk) of the F-layout block at `blk`: trb = tr_lane_base(lane)
DEV u32x4 ds_read_tr_frag(const char* blk, int trb, int tile, int kk) {
  const int a0 = trb + tile * 2048 + kk * 256;
  return ds_read_tr_pair(blk, a0, a0 ^ 64);
}

// ---------------------------------------------------------------------------------------------
// k_bwd: one layer of the backward pass, data-gradient and weight-gradient fused so that every
// scratch tensor is read exactly once:
//     reads   D_l     (delta of layer l, rows j)          F-layout bf16
//             P_{l-1} (phase of layer l-1, cols i)        F-layout unorm16
//     writes  D_{l-1} = (D_l W_l) * omega cos(P_{l-1})     F-layout bf16
//     accumulates dW_l[j][i] += D_l^T sin(P_{l-1}),  db_l[j] += sum D_l    (per-workgroup slab)
//   32-pixel blocks stream through a 4-slot LDS ring filled by global_load_lds two blocks ahead:
//     phase X: each wave keeps its rows of W_l^T in registers (stationary A operand); B = delta
//              pieces read lane-linear from LDS; epilogue reads the raw phase piece it owns, forms
//              cos (for delta_{l-1}, stored to HBM) and sin (bf16, written back IN PLACE over the phase);
//     phase W: dW MFMAs with both operands read transposed (ds_read_b64_tr_b16) from the two LDS images.
// ---------------------------------------------------------------------------------------------
struct BwdLayerArgs {
  const u32x4* D;       // delta image of layer l (JW/16 k-steps per pixel block)
  const u32x4* P;       // phase image of layer l-1 (IW/16 k-steps per pixel block)
  u32x4* Dout;          // delta image of layer l-1
  const u32x4* wb;      // backward weight image of layer l: tiles (IW/32) x k-steps (KSX) x 64 lanes
  long n_pb;            // pixel blocks in this chunk
  int pb_per_wg;        // multiple of 2
  float* slab;          // [gridDim.x][JW*IW + JW]
  // P0 variant (layer l-1 is layer 0): its phases are re-derived from the pixel coordinates instead of read
  const f32x4* l0tab;   // [IW] {w00, w01, b0, 0}
  long pix0, npix;      // first local pixel of the chunk, local pixel count
  int W, row_begin;
  unsigned long long w_magic;   // ceil(2^40 / W): row = (p * w_magic) >> 40
  float inv_hm1, inv_wm1;       // 1/(H-1), 1/(W-1) (0 when the extent is 1): torch.linspace(0,1,n)[i] ~ i/(n-1)
  float sc_first;               // first_omega_0 / (2 pi)
};

template <int JW, int IW, int WAVES_R, int WAVES_C, bool LAST, bool P0, typename OP, int NB>
__global__ __launch_bounds__(WAVES_R* WAVES_C * 64) void k_bwd(BwdLayerArgs a) {
  constexpr int NW = WAVES_R * WAVES_C;
  constexpr int JT = JW / 32, IT = IW / 32;
  constexpr int WJ = JT / WAVES_R, WI = IT / WAVES_C;
  constexpr int KSJ = JW / 16, KSI = IW / 16;
  constexpr int KSX = LAST ? 1 : KSJ;         // k-steps of the dX product (last layer: <= 3 real rows -> one step)
  constexpr int XT = IT / NW;                 // phase-X row tiles per wave
  static_assert(IT % NW == 0, "phase-X tiling needs NW <= IT");
  // NB = LDS ring depth in blocks of 32 pixels.  Block k is requested PD = NB-2 steps before it is used: the
  // loaded HBM latency is several microseconds, so the bytes in flight per CU (PD * BLK) set the streaming
  // rate long before the instruction schedule does (measured: 2 blocks in flight = 10 GB/s per CU).
  constexpr int PD = NB - 2;
  constexpr int BLK = (KSJ + KSI) * 1024;     // bytes per ring slot: delta pieces then phase pieces
  constexpr int G_MIN = KSJ / NW + (P0 ? 0 : KSI / NW);  // LDS-DMA instructions every wave issues per block (lower bound)
  constexpr int S_ST = 2 * XT;                // delta stores per wave per block
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / WAVES_C, wc = wave % WAVES_C;
  const int xit0 = wave * XT;
  const int trb = tr_lane_base(lane);
  const int lsw = sw_lane(lane, 0);      // slot of this lane's register image in an even piece; odd: lsw ^ 8
  const uint32_t lswb = (uint32_t)lsw * 16u;   // the same as a byte offset; odd: lswb ^ 128

  // stationary W^T rows of this wave: k-steps [0, KSR) in registers, [KSR, KSX) parked in LDS behind the ring
  // (the widest configuration needs the 32 registers: 256 accumulator + 128 weight registers leave too few)
  constexpr int KSR = (XT * KSX > 24) ? KSX - (P0 ? 3 : 4) : KSX;   // (P0: the layer-0 table takes 4 KiB of LDS)
  constexpr int WSP = KSX - KSR;                // spilled k-steps per row tile
  char* sWsp = smem + NB * BLK + (size_t)wave * XT * WSP * 1024;
  u32x4 wreg[XT][KSR];
#pragma unroll
  for (int x = 0; x < XT; ++x) {
#pragma unroll
    for (int s = 0; s < KSR; ++s) wreg[x][s] = a.wb[((xit0 + x) * KSX + s) * 64 + lane];
#pragma unroll
    for (int s = KSR; s < KSX; ++s)
      reinterpret_cast<u32x4*>(sWsp + (x * WSP + s - KSR) * 1024)[lane] = a.wb[((xit0 + x) * KSX + s) * 64 + lane];
  }

  f32x16 acc[WJ][WI];
#pragma unroll
  for (int x = 0; x < WJ; ++x)
#pragma unroll
    for (int y = 0; y < WI; ++y) acc[x][y] = f32x16{};
  float dbs[WJ];
#pragma unroll
  for (int x = 0; x < WJ; ++x) dbs[x] = 0.f;

  // block-cyclic assignment: at any moment the workgroups stream NEIGHBOURING 16 KiB pieces of each tensor
  // (contiguous ranges per workgroup would put all of them a power-of-two stride apart)
  const long pb_begin = blockIdx.x, pb_step = gridDim.x;
  const int nblk = (int)((a.n_pb - pb_begin + pb_step - 1) / pb_step);

  auto stage = [&](int k) {   // block k of this workgroup -> ring slot k % NB, one 1 KiB piece per wave-instruction
    char* base = smem + (k % NB) * BLK;
    const long pb = pb_begin + k * pb_step;
    for (int pc = wave; pc < KSJ; pc += NW) glds16s(a.D + (pb * KSJ + pc) * 64, lswb ^ ((pc & 1) << 7), base + pc * 1024);
    if (!P0)
      for (int pc = wave; pc < KSI; pc += NW) glds16s(a.P + (pb * KSI + pc) * 64, lswb ^ ((pc & 1) << 7), base + (KSJ + pc) * 1024);
  };
  // the same pieces one at a time, to be issued BETWEEN the MFMAs of a step (an LDS-DMA instruction costs its wave
  // ~150-200 issue cycles inside a burst of them, ~60 with MFMAs in flight): piece i of this wave for block k
  constexpr int GD = KSJ / NW, GP = GD + (P0 ? 0 : KSI / NW);
  // (k_wdw issues its pieces that way; here both placements tried - X chunks and W chunks - pushed the 256-register
  //  8-wave build into scratch spills INSIDE the steady loop, whose reloads then drain the DMA queue, so the burst
  //  at the top of the step stays)
  constexpr bool SPREAD = false && (KSJ % NW == 0) && (P0 || KSI % NW == 0) && GP <= 2 * WJ;
  auto stage_piece = [&](int k, int i) {
    // past the last block the SAME ring slot is refilled with the last block again: the slot is free (its block
    // was consumed NB - PD steps ago), the bytes are never read, and the steady loop stays branch-free
    char* base = smem + (k % NB) * BLK;
    const long pb = pb_begin + (long)(k < nblk ? k : nblk - 1) * pb_step;
    if (i < GD) {
      const int pc = wave + NW * i;
      glds16s(a.D + (pb * KSJ + pc) * 64, lswb ^ ((pc & 1) << 7), base + pc * 1024);
    } else {
      const int pc = wave + NW * (i - GD);
      glds16s(a.P + (pb * KSI + pc) * 64, lswb ^ ((pc & 1) << 7), base + (KSJ + pc) * 1024);
    }
  };
  // P0: layer-0 table in LDS (behind the parked weights) and this lane's pixel coordinates per block
  const f32x4* sL0 = reinterpret_cast<const f32x4*>(smem + NB * BLK + (size_t)NW * XT * WSP * 1024);
  if (P0) {
    f32x4* dst = reinterpret_cast<f32x4*>(smem + NB * BLK + (size_t)NW * XT * WSP * 1024);
    for (int i = tid; i < IW; i += NW * 64) dst[i] = a.l0tab[i];
  }
  auto pixel_xy = [&](int k, float& x0, float& x1) {
    long p = a.pix0 + (pb_begin + (long)k * pb_step) * 32 + (lane & 31);
    if (p >= a.npix) p = a.npix - 1;
    const unsigned row = (unsigned)(((unsigned long long)p * a.w_magic) >> 40);
    const unsigned col = (unsigned)(p - (long)row * a.W);
    x0 = ((float)(row + (unsigned)a.row_begin) * a.inv_hm1 - 0.5f) * 2.0f;
    x1 = ((float)col * a.inv_wm1 - 0.5f) * 2.0f;
  };

  for (int k = 0; k < PD && k < nblk; ++k) stage(k);
  // Software pipeline over 32-pixel blocks; step k runs phase X on block k and phase W on block k-1 in
  // one instruction stream (independent work: the VALU-heavy X epilogue hides under the W MFMAs).
  //   X(k): delta_{k} B-pieces * stationary W^T -> G; epilogue: cos/sin of the phase piece this wave owns,
  //         delta_{l-1} to HBM, sin (16-bit) written back IN PLACE over the phase piece
  //   W(k-1): dW += delta^T * act, both operands read transposed (ds_read_b64_tr_b16)
  // ---- one pipeline step (block kx in phase X, block kx-1 in phase W), hand-scheduled ----------------
  // One wave per SIMD: nothing but this wave can fill the shadow of its own MFMAs, and hipcc emits MFMA
  // chains, LDS reads and VALU epilogues back to back.  The step is therefore written as a linear list of
  // CHUNKS separated by sched_barrier(0); every chunk (a) issues the LDS reads of the NEXT chunk's operands
  // into a second register set, (b) issues a few MFMAs on operands read one chunk earlier, (c) runs a slice
  // of a VALU epilogue.  In-order issue then overlaps (b)'s matrix-pipe time with (a)'s latency and (c).
  constexpr int XC = KSX >= 2 ? KSX / 2 : 1;    // X chunks per row tile
  constexpr int XS = KSX / XC;                  // k-steps per X chunk (2, or 1 for the last layer)
  constexpr int ESUB = 4;                       // epilogue slices per row tile: (q, half) groups of 4 values
  auto x_load = [&](int k, int c, u32x4* dst) {
    const char* sD = smem + (k % NB) * BLK;
#pragma unroll
    for (int i = 0; i < XS; ++i)
      dst[i] = reinterpret_cast<const u32x4*>(sD + (c * XS + i) * 1024)[lsw ^ (((c * XS + i) & 1) << 3)];
  };
  auto x_mma_chunk = [&](int x, int c, const u32x4* b, f32x16& g) {
#pragma unroll
    for (int i = 0; i < XS; ++i) {
      const int s = c * XS + i;
      const u32x4 w = s < KSR ? wreg[x][s < KSR ? s : 0]
                              : reinterpret_cast<const u32x4*>(sWsp + (x * WSP + (s >= KSR ? s - KSR : 0)) * 1024)[lane];
      g = OP::mfma(w, b[i], g);
    }
  };
  // slice e of the epilogue of row tile x: e>>1 selects the phase piece (q), e&1 the half of its 8 values
  // (4 independent cos/sin chains per slice keep the VALU fed while only one wave runs on the SIMD)
  u32x4 ep_p, ep_d, ep_s;
  float ep_x0 = 0.f, ep_x1 = 0.f;
  auto x_epi_sub = [&](int k, int x, int e, const f32x16& g) {
    char* sP = smem + (k % NB) * BLK + KSJ * 1024;
    const int q = e >> 1, hf = e & 1, ks = 2 * (xit0 + x) + q;
    u32x4* pp = reinterpret_cast<u32x4*>(sP + ks * 1024) + (lsw ^ ((q & 1) << 3));   // ks = 2*tile + q
    if (!P0 && hf == 0) ep_p = *pp;
#pragma unroll
    for (int j2 = 2 * hf; j2 < 2 * hf + 2; ++j2) {
      float r0, r1;
      if (P0) {   // phase of layer 0 from the coordinates: neurons 16*ks + PI(h, 2*j2), +1
        const f32x4 t0 = sL0[16 * ks + pi_perm(lane >> 5, 2 * j2)], t1 = sL0[16 * ks + pi_perm(lane >> 5, 2 * j2 + 1)];
        r0 = __builtin_fmaf(t0.y, ep_x1, __builtin_fmaf(t0.x, ep_x0, t0.z)) * a.sc_first;
        r1 = __builtin_fmaf(t1.y, ep_x1, __builtin_fmaf(t1.x, ep_x0, t1.z)) * a.sc_first;
      } else {
        r0 = phase_rev_lo(ep_p[j2]); r1 = phase_rev_hi(ep_p[j2]);
      }
      ep_d[j2] = OP::pack2(g[8 * q + 2 * j2] * __builtin_amdgcn_cosf(r0), g[8 * q + 2 * j2 + 1] * __builtin_amdgcn_cosf(r1));
      ep_s[j2] = OP::pack2(__builtin_amdgcn_sinf(r0), __builtin_amdgcn_sinf(r1));
    }
    if (hf == 1) {
#ifndef SF_EXPERIMENT_NO_STORE
      store_stream(&a.Dout[((pb_begin + k * pb_step) * KSI + ks) * 64 + lane], ep_d);
#else
      asm volatile("" ::"v"(ep_d));
#endif
      *pp = ep_s;
    }
  };
  auto wa_load = [&](int k, int kk, int x) -> u32x4 {   // delta^T fragment (rows = neurons of this wave's tile x)
    const char* sD = smem + (k % NB) * BLK;
    return ds_read_tr_frag(sD, trb, wr * WJ + x, kk);
  };
  auto wb_load = [&](int k, int kk, u32x4* dst) {       // activation fragments (cols = this wave's WI tiles)
    const char* sP = smem + (k % NB) * BLK + KSJ * 1024;
#pragma unroll
    for (int y = 0; y < WI; ++y)
      dst[y] = ds_read_tr_frag(sP, trb, wc * WI + y, kk);
  };
  auto w_mma_chunk = [&](int x, const u32x4& fa, const u32x4* fb) {
#pragma unroll
    for (int y = 0; y < WI; ++y) acc[x][y] = OP::mfma(fa, fb[y], acc[x][y]);
    if (wc == 0) {
      float tsum = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) tsum += OP::sum2(fa[e]);
      dbs[x] += tsum;
    }
  };
  auto step = [&](int kx, bool do_x, bool do_w, bool do_s) {   // do_s: issue the LDS-DMA of block kx+PD inside the X chunks
    u32x4 xb[2][XS];            // X B-operand pieces, double-buffered
    u32x4 fb[2][WI], fa[2];     // W operands: activation fragments per k-step (2 sets), delta^T fragment (2 sets)
    f32x16 gp = {}, gc = {};
    if (P0 && do_x) pixel_xy(kx, ep_x0, ep_x1);
    if (do_x) x_load(kx, 0, xb[0]);
    else if (do_w) { wb_load(kx - 1, 0, fb[0]); fa[0] = wa_load(kx - 1, 0, 0); }
    __builtin_amdgcn_sched_barrier(0);
    // ---- X chunks: tile x, chunk c; epilogue slices of tile x-1 ride along ----
    if (do_x) {
#pragma unroll
      for (int x = 0; x < XT; ++x) {
        gc = f32x16{};
#pragma unroll
        for (int c = 0; c < XC; ++c) {
          const int i = x * XC + c;                  // linear X chunk index; operands sit in xb[i & 1]
          if (i + 1 < XT * XC) x_load(kx, (c + 1) % XC, xb[(i + 1) & 1]);
          else if (do_w) { wb_load(kx - 1, 0, fb[0]); fa[0] = wa_load(kx - 1, 0, 0); }
          x_mma_chunk(x, c, xb[i & 1], gc);
          if (x > 0) {   // ESUB slices of the previous tile spread over the XC chunks of this one
#pragma unroll
            for (int e = c * ESUB / XC; e < (c + 1) * ESUB / XC; ++e) x_epi_sub(kx, x - 1, e, gp);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        gp = gc;
      }
    }
    // ---- W chunks: k-step kk, row tile x; epilogue slices of the last X tile ride along in kk = 0 ----
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
      for (int x = 0; x < WJ; ++x) {
        const int i = kk * WJ + x;                    // linear W chunk index; delta^T fragment in fa[i & 1]
        if (do_w) {
          if (x + 1 < WJ) fa[(i + 1) & 1] = wa_load(kx - 1, kk, x + 1);
          else if (kk == 0) { wb_load(kx - 1, 1, fb[1]); fa[(i + 1) & 1] = wa_load(kx - 1, 1, 0); }
          w_mma_chunk(x, fa[i & 1], fb[kk]);
          if (SPREAD && do_s && i < GP) stage_piece(kx + PD, i);   // W chunks: fewest live registers of the step
        }
        if (do_x && kk == 0) {
#pragma unroll
          for (int e = x * ESUB / WJ; e < (x + 1) * ESUB / WJ; ++e) x_epi_sub(kx, XT - 1, e, gp);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

  if (nblk > 0) {
    bar_all();                          // block 0 landed
    if (PD < nblk) stage(PD);
    asm volatile("" ::: "memory");
    step(0, true, false, false);
    for (int k = 1; k < nblk; ++k) {
      // block k landed (requested PD steps ago; younger in the in-order vmcnt queue: the delta stores of the PD
      // steps since, and the DMA of blocks k+1 .. k+PD-1) and every wave finished step k-1
      if (k >= PD && k + PD - 1 < nblk) bar_dma<PD * S_ST + (PD - 1) * G_MIN>(); else bar_all();
      if (!SPREAD && k + PD < nblk) stage(k + PD);
      asm volatile("" ::: "memory");
      step(k, true, true, true);
    }
    bar_lds();
    step(nblk, false, true, false);
  }
  float* slab = a.slab + (size_t)blockIdx.x * (JW * IW + JW);
  const int cl = lane & 31, hh = lane >> 5;
#pragma unroll
  for (int x = 0; x < WJ; ++x)
#pragma unroll
    for (int y = 0; y < WI; ++y)
#pragma unroll
      for (int t = 0; t < 16; ++t)
        slab[(size_t)(32 * (wr * WJ + x) + rho(t, hh)) * IW + 32 * (wc * WI + y) + cl] = acc[x][y][t];
  if (wc == 0) {
#pragma unroll
    for (int x = 0; x < WJ; ++x) {
      const float tsum = dbs[x] + __shfl_xor(dbs[x], 32);
      if (hh == 0) slab[JW * IW + 32 * (wr * WJ + x) + cl] = tsum;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// k_dw0: weight gradient of layer 0 (no data gradient below it): dW_0[j][c] = sum_pixels delta_0[pix][j] * x_c[pix].
//   delta_0 blocks stream through a 4-slot LDS ring (LDS-DMA two blocks ahead, as in k_bwd) and are read
//   transposed; the B operand is built in registers from the pixel coordinates, split in two 16-bit terms
//   (columns x0_hi, x0_lo, x1_hi, x1_lo: hi + lo keeps 2x the significand of one 16-bit value), so the
//   kernel touches HBM for delta_0 only.  One 32-row tile of the slab per wave.
// ---------------------------------------------------------------------------------------------
struct Dw0Args {
  const u32x4* D;       // delta image of layer 0 (ks_total k-steps per pixel block; this launch reads JW/16 from ks_off)
  int ks_total, ks_off; // ks_total = JW/16, ks_off = 0 unless the layer is wider than one launch covers (siren_wide.hip)
  long n_pb;
  float* slab;          // [gridDim.x][JW*32 + JW]
  long pix0, npix;
  int W, row_begin;
  unsigned long long w_magic;
  float inv_hm1, inv_wm1;   // 1/(H-1), 1/(W-1): torch.linspace(0,1,n)[i] = i/(n-1) to within 1 ulp (enough for a
                            // gradient term; the forward uses the exact vectors).  No loads in the DMA loop.
};

template <int JW, typename OP>
__global__ __launch_bounds__(JW * 2) void k_dw0(Dw0Args a) {
  constexpr int NW = JW / 32, KSJ = JW / 16, NB = 8, PD = NB - 2, BLK = KSJ * 1024;   // 96 KiB in flight at JW = 256
  constexpr int G_MIN = KSJ / NW;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long pb_begin = blockIdx.x, pb_step = gridDim.x;
  const int nblk = (int)((a.n_pb - pb_begin + pb_step - 1) / pb_step);
  f32x16 acc = {};
  float dbs = 0.f;
  const int trb = tr_lane_base(lane);
  auto stage = [&](int k) {
    char* base = smem + (k % NB) * BLK;
    const long pb = pb_begin + k * pb_step;
    for (int pc = wave; pc < KSJ; pc += NW)
      glds16s(a.D + (pb * a.ks_total + a.ks_off + pc) * 64, (uint32_t)sw_lane(lane, pc & 1) * 16u, base + pc * 1024);
  };
  // B operand = coordinates of the block's 32 pixels as 16-bit columns {x0_hi, x0_lo, x1_hi, x1_lo}.  Wave 0
  // builds a [4][32] table per block (one pixel per lane, one integer division), double-buffered by block
  // parity behind the ring; every wave then reads its fragment with one ds_read_b128 per k-step.
  uint16_t* sXY = reinterpret_cast<uint16_t*>(smem + NB * BLK);          // [2][4][32]
  auto build_xy = [&](int k) {
    if (wave == 0 && lane < 32) {
      long p = a.pix0 + (pb_begin + (long)k * pb_step) * 32 + lane;
      if (p >= a.npix) p = a.npix - 1;
      const unsigned row = (unsigned)(((unsigned long long)p * a.w_magic) >> 40);
      const unsigned col = (unsigned)(p - (long)row * a.W);
      const float x0 = ((float)(row + (unsigned)a.row_begin) * a.inv_hm1 - 0.5f) * 2.0f;
      const float x1 = ((float)col * a.inv_wm1 - 0.5f) * 2.0f;
      const uint32_t h0 = OP::pack2(x0, 0.f), h1 = OP::pack2(x1, 0.f);
      uint16_t* t = sXY + (k & 1) * 128 + lane;
      t[0] = (uint16_t)h0;
      t[32] = (uint16_t)OP::pack2(x0 - OP::lo(h0), 0.f);
      t[64] = (uint16_t)h1;
      t[96] = (uint16_t)OP::pack2(x1 - OP::lo(h1), 0.f);
    }
  };
  auto coord_frag = [&](int k, int kk) -> u32x4 {   // lane (col c = lane&31, hq = lane>>5): pixels 16*kk + 8*hq + j
    const int c = lane & 31;
    u32x4 o = {0u, 0u, 0u, 0u};
    if (c < 4) o = *reinterpret_cast<const u32x4*>(sXY + (k & 1) * 128 + c * 32 + 16 * kk + 8 * (lane >> 5));
    return o;
  };
  for (int k = 0; k < PD && k < nblk; ++k) stage(k);
  build_xy(0);
  for (int k = 0; k < nblk; ++k) {
    if (k >= PD && k + PD - 1 < nblk) bar_dma<(PD - 1) * G_MIN>(); else bar_all();
    if (k + PD < nblk) stage(k + PD);
    asm volatile("" ::: "memory");
    if (k + 1 < nblk) build_xy(k + 1);
    const u32x4 fb0 = coord_frag(k, 0), fb1 = coord_frag(k, 1);
    const char* sD = smem + (k & (NB - 1)) * BLK;
    const u32x4 fa0 = ds_read_tr_frag(sD, trb, wave, 0);
    const u32x4 fa1 = ds_read_tr_frag(sD, trb, wave, 1);
    acc = OP::mfma(fa0, fb0, acc);
    acc = OP::mfma(fa1, fb1, acc);
    float t = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) t += OP::sum2(fa0[e]) + OP::sum2(fa1[e]);
    dbs += t;
  }
  float* slab = a.slab + (size_t)blockIdx.x * (JW * 32 + JW);
  const int cl = lane & 31, hh = lane >> 5;
#pragma unroll
  for (int t = 0; t < 16; ++t) slab[(size_t)(32 * wave + rho(t, hh)) * 32 + cl] = acc[t];
  const float tsum = dbs + __shfl_xor(dbs, 32);
  if (hh == 0) slab[JW * 32 + 32 * wave + cl] = tsum;
}

// ---------------------------------------------------------------------------------------------
// k_reduce: flat gradient from the slabs, fixed summation order (workgroup 0, 1, 2, ...).
//   mode 0: hidden / last layer: gW[j][i] = sum slab[j][i] (j < rows_out, i < cols_out), gb[j] = sum db[j]
//   mode 1: first layer: gW[j][0] = c0+c1, gW[j][1] = c2+c3 ; gb[j] = db[j]
// ---------------------------------------------------------------------------------------------
struct ReduceArgs {
  const float* slab;
  int n_wg, slab_rows, slab_cols;   // slab geometry (padded)
  int rows_out, cols_out;           // true layer shape
  int mode;
  float* gW;                        // [rows_out][cols_out]
  float* gb;                        // [rows_out]
  int accumulate;                   // add to the existing gradient (later chunks)
  float scale;                      // 1 / gradient pre-scale (power of two; fp16 backward operands)
  const float* scale_dev;           // 8-bit scratch: scale_dev[1] = 1 / (chunk pre-scale * n_values) overrides `scale`
  const float* scale2_dev;          // fp8 deltas: 1 / (cumulative per-layer delta scale of the layer's incoming deltas), or nullptr
};

// 256 threads = 16 outputs x 16 slab groups: group g sums slabs g, g+16, ... (independent loads, issued ahead),
// then the 16 partial sums are combined in fixed order 0..15 => deterministic, and the serial chain per output
// is n_wg/16 loads instead of n_wg (these launches are pure latency on small fits).
__global__ __launch_bounds__(256) void k_reduce(ReduceArgs a) {
  __shared__ float sh[16][17];
  const long slab_sz = (long)a.slab_rows * a.slab_cols + a.slab_rows;
  const int nW = a.rows_out * a.cols_out;
  const int o = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const int idx = blockIdx.x * 16 + o;
  const bool live = idx < nW + a.rows_out;
  float s = 0.f;
  if (live) {
    if (idx < nW) {
      const int j = idx / a.cols_out, i = idx % a.cols_out;
      if (a.mode == 0) {
        const float* p = a.slab + (long)j * a.slab_cols + i;
#pragma unroll 4
        for (int w = grp; w < a.n_wg; w += 16) s += p[w * slab_sz];
      } else {
        const float* p = a.slab + (long)j * a.slab_cols + 2 * i;
#pragma unroll 4
        for (int w = grp; w < a.n_wg; w += 16) s += p[w * slab_sz] + p[w * slab_sz + 1];
      }
    } else {
      const float* p = a.slab + (long)a.slab_rows * a.slab_cols + (idx - nW);
#pragma unroll 4
      for (int w = grp; w < a.n_wg; w += 16) s += p[w * slab_sz];
    }
  }
  sh[grp][o] = s;
  __syncthreads();
  if (grp == 0 && live) {
    float t = sh[0][o];
#pragma unroll
    for (int g = 1; g < 16; ++g) t += sh[g][o];
    t *= (a.scale_dev ? a.scale_dev[1] : a.scale) * (a.scale2_dev ? a.scale2_dev[0] : 1.0f);
    float* dst = idx < nW ? a.gW + idx : a.gb + (idx - nW);
    *dst = a.accumulate ? *dst + t : t;
  }
}

// Vectorised form for hidden layers, whose slab layout [W rows*cols | b rows] IS the flat-gradient
// layout: out[i] = sum_w slab[w][i].  256 threads = 8 float4 columns x 32 slab groups; group g sums
// slabs g, g+32, ... in order, groups are combined 0..31 in order (fixed order => deterministic).
__global__ __launch_bounds__(256) void k_reduce_vec(const float* slab, int n_wg, long stride, int n4, float* out,
                                                    int accumulate, float scale, const float* scale_dev, const float* scale2_dev) {
  if (scale_dev) scale = scale_dev[1];
  if (scale2_dev) scale *= scale2_dev[0];      // (powers of two: exact)
  __shared__ f32x4 sh[32][8];
  const int col = threadIdx.x & 7, grp = threadIdx.x >> 3;
  const int i4 = blockIdx.x * 8 + col;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (i4 < n4) {
    const f32x4* p = reinterpret_cast<const f32x4*>(slab) + i4;
    const long st4 = stride / 4;
#pragma unroll 4
    for (int w = grp; w < n_wg; w += 32) s += p[w * st4];
  }
  sh[grp][col] = s;
  __syncthreads();
  if (grp == 0 && i4 < n4) {
    f32x4 t = sh[0][col];
#pragma unroll
    for (int g = 1; g < 32; ++g) t += sh[g][col];
    t *= scale;
    f32x4* o = reinterpret_cast<f32x4*>(out) + i4;
    *o = accumulate ? *o + t : t;
  }
}

// sum of the per-workgroup SSE partials in double, fixed order
// (loss_tab / iter: graph replay mode of sf_step — the per-step value also goes to loss_tab[*iter])
__global__ void k_sse_reduce(const float* part, int n, double* out, double* loss_tab, const int* iter) {
  __shared__ double sh[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) s += (double)part[i];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    *out = sh[0];
    if (loss_tab) loss_tab[*iter] = sh[0];
  }
}

// graph replay mode: advances the device-side step index once per replayed step (last node of the graph)
__global__ void k_tick(int* iter) { *iter += 1; }

// ---------------------------------------------------------------------------------------------
// k_adam: torch.optim.Adam single-tensor op order (torch 2.x): lerp for exp_avg, mul/addcmul for
// exp_avg_sq, denom = sqrt(v)/sqrt(bc2) + eps, p -= step_size * m / denom; then Masking.apply_mask.
// ---------------------------------------------------------------------------------------------
struct AdamArgs {
  float* p; const float* g; float* m; float* v; const float* mask;
  long n;
  float beta1, beta2, eps, step_size, bc2_sqrt;
  float omb1, omb2;   // (float)(1 - (double)beta): torch.optim.Adam forms 1 - beta in double
  const float* tab;   // graph replay mode: {step_size, bc2_sqrt} of step *iter (precomputed on the host in double)
  const int* iter;
};
__global__ void k_adam(AdamArgs a) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  if (a.tab) {
    const int it = *a.iter;
    a.step_size = a.tab[2 * it];
    a.bc2_sqrt = a.tab[2 * it + 1];
  }
  const float g = a.g[i];
  float m = a.m[i], v = a.v[i];
  m = m + a.omb1 * (g - m);
  v = v * a.beta2 + a.omb2 * g * g;
  const float denom = __builtin_sqrtf(v) / a.bc2_sqrt + a.eps;
  float p = a.p[i] - a.step_size * (m / denom);
  if (a.mask) p *= a.mask[i];
  a.p[i] = p; a.m[i] = m; a.v[i] = v;
}

// ---------------------------------------------------------------------------------------------
// k_images: rebuild every low-precision weight image from the fp32 master parameters
// ---------------------------------------------------------------------------------------------
struct ImgArgs {
  const float* params;      // flat
  int depth, WD, out_features;
  long off_w[16], off_b[16];
  float wscale;             // scale of the LAST layer's forward image (power of two)
  float hscale;             // scale of the hidden forward images and biases: hidden_omega_0 / (2 pi), so that the
                            // MFMA accumulator of a hidden layer is the phase in revolutions (one VALU multiply
                            // per output value less in the issue-bound k_fwd epilogue)
  float om_first, om_hidden; // backward images carry omega of layer l-1: d sin(om z)/dz = om cos(om z)
  int fwd_is_f16;
  uint16_t* wf; uint16_t* wf_last; uint16_t* wb; uint16_t* wb_last;
  f32x4* l0tab;
  uint16_t* l0img;          // layer 0 as MFMA A fragments (k_fwd_pipe, see kL0Split), always fp16; nullptr: not built
  float sc_first;           // first_omega_0 / (2 pi)
  const float* link;        // fp8 deltas: link[l] = power-of-two scale folded into the backward image of layer l (k_fp8_links); nullptr: 1
};
DEV uint16_t to_bf16(float x) { return (uint16_t)(OpBF16::pack2(x, 0.f) & 0xffffu); }
DEV uint16_t to_f16(float x) { return (uint16_t)(OpF16::pack2(x, 0.f) & 0xffffu); }

// forward-image geometry for a run-time width (mirrors FwdImg<WD>)
struct FwdGeom {
  int NT, KS, H0, H1, XP, PIECES;
  __host__ __device__ explicit FwdGeom(int WD) {
    NT = WD / 32; KS = WD / 16; H0 = NT >= 2 ? NT / 2 : 1; H1 = NT - H0; XP = H0 * KS + 1; PIECES = XP + H1 * KS + 1;
  }
  __host__ __device__ int tile_piece(int nt) const { return nt < H0 ? nt * KS : XP + (nt - H0) * KS; }
  __host__ __device__ int bias_piece(int nt) const { return nt < H0 ? H0 * KS : XP + H1 * KS; }
  __host__ __device__ int bias_off(int nt) const { return (nt < H0 ? nt : nt - H0) * 32; }
};

// ---------------------------------------------------------------------------------------------
// k_fp8_norms + k_fp8_links: per-layer power-of-two scales of the fp8 deltas (scratch format 8), from the weights alone.
// delta_{l-1} = (W_l^T delta_l) * omega_{l-1} cos(phi_{l-1}) has, for uncorrelated delta components,
//   rms(delta_{l-1}) / rms(delta_l) = gain_l = omega_{l-1} * sqrt(0.5 * ||W_l||_F^2 / n_in)
// (1.0 at the SIREN initialisation; measured on fits: the real growth is this times 1.0 - 1.6 per layer, 10 - 25 x from the
// last hidden layer down to layer 0 where the prediction says 3 - 6 x: scripts/fp8_gain_probe.py).  One e4m3 scale per chunk
// cannot hold seven tensors that far apart between the 448 saturation and the 2^-6 subnormals (the non-smooth 1000-step
// fixture lost 1.6 dB to clipped layer-0 / layer-1 deltas), so every layer's outgoing deltas are divided by the predicted gain
// rounded to a power of two: link[l] = 2^-round(log2(gain_l)) goes into the backward image of layer l (k_images) - no
// instruction in any kernel - and the reduction of layer l's weight gradient multiplies by inv[l] = 1 / prod_{m > l} link[m].
// The last layer's link also normalises its own gain, so rms(first hidden delta) ~ the chunk's target.
// Sums in double, fixed order: a function of the parameters only (no state, bit-reproducible).
// ---------------------------------------------------------------------------------------------
struct Fp8ScaleArgs {
  const float* params;
  int depth, WD, out_features;
  long off_w[16];
  float om_first, om_hidden;
  float* link;     // [16]
  float* inv;      // [16]
  double* nrm;     // [16] sums of squares of the layers' weights
};
// (two launches: the sums of squares one workgroup per layer - a single workgroup spent 120 us waiting on its own loads -
//  then one thread for the seven links)
__global__ __launch_bounds__(1024) void k_fp8_norms(Fp8ScaleArgs a) {
  __shared__ double sh[1024];
  const int t = threadIdx.x, l = 1 + (int)blockIdx.x;
  const long n = (long)(l == a.depth - 1 ? a.out_features : a.WD) * a.WD;
  const float* W = a.params + a.off_w[l];
  double s = 0.0;
#pragma unroll 8
  for (long i = t; i < n; i += 1024) s += (double)W[i] * (double)W[i];
  sh[t] = s;
  __syncthreads();
  for (int st = 512; st > 0; st >>= 1) {          // fixed tree: the same sum on every run
    if (t < st) sh[t] += sh[t + st];
    __syncthreads();
  }
  if (t == 0) a.nrm[l] = sh[0];
}
__global__ void k_fp8_links(Fp8ScaleArgs a) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double S = 1.0;                                  // cumulative scale of the deltas ENTERING layer l's weight gradient
  for (int l = a.depth - 1; l >= 0; --l) {
    a.inv[l] = (float)(1.0 / S);
    if (l == 0) { a.link[0] = 1.0f; break; }
    const double om = l - 1 == 0 ? (double)a.om_first : (double)a.om_hidden;
    double gain = om * sqrt(0.5 * a.nrm[l] / (double)a.WD);
    if (!(gain > 1e-6)) gain = 1e-6;               // (all-zero layer, NaN)
    if (gain > 1e6) gain = 1e6;
    int e = 0;
    const double m = frexp(gain, &e);              // gain = m * 2^e, m in [0.5, 1): round(log2(gain)) = e - (m < sqrt(0.5))
    const int k = e - (m < 0.70710678118654752440 ? 1 : 0);
    const double lk = ldexp(1.0, -k);
    a.link[l] = (float)lk;
    S *= lk;
  }
}

__global__ void k_images(ImgArgs a) {
  const int WD = a.WD, NT = WD / 32, KS = WD / 16;
  const FwdGeom G(WD);
  const long per_layer = (long)WD * WD;
  const long n_hidden = (long)(a.depth - 2) * per_layer;
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  // element decode of an image [tile][s][lane][8]
  if (gid < n_hidden) {
    const int l = (int)(gid / per_layer) + 1;
    long e = gid % per_layer;
    const int j = e & 7; e >>= 3;
    const int lane = e & 63; e >>= 6;
    const int s = (int)(e % KS), tile = (int)(e / KS);
    const int r = lane & 31, h = lane >> 5;
    const float* Wl = a.params + a.off_w[l];
    const float wfwd = Wl[(long)(32 * tile + r) * WD + 16 * s + pi_perm(h, j)] * a.hscale;
    // (ONE factor: hipcc fuses W * factor and the conversion into v_fma_mixlo_f16, a single rounding; a second multiply would
    //  round the product to f32 first and change one image element in some thousands - formats 16 / 12 must not move)
    const float bsc = (l - 1 == 0 ? a.om_first : a.om_hidden) * (a.link ? a.link[l] : 1.0f);
    const float wbwd = Wl[(long)(16 * s + pi_perm(h, j)) * WD + 32 * tile + r] * bsc;
    const long dst = ((long)(l - 1) * G.PIECES + G.tile_piece(tile) + s) * 512 + lane * 8 + j;
    a.wf[dst] = a.fwd_is_f16 ? to_f16(wfwd) : to_bf16(wfwd);
    a.wb[gid] = a.fwd_is_f16 ? to_f16(wbwd) : to_bf16(wbwd);
  }
  if (gid < (long)(a.depth - 2) * WD) {  // hidden biases into the bias pieces (fp32, pre-scaled)
    const int l = (int)(gid / WD) + 1, n = (int)(gid % WD), nt = n / 32;
    float* piece = reinterpret_cast<float*>(a.wf + ((long)(l - 1) * G.PIECES + G.bias_piece(nt)) * 512);
    piece[G.bias_off(nt) + (n & 31)] = a.params[a.off_b[l] + n] * a.hscale;
  }
  const int L = a.depth - 1;
  if (gid < (long)KS * 64 * 8) {  // last layer forward image: one tile of 32 padded rows
    long e = gid;
    const int j = e & 7; e >>= 3;
    const int lane = e & 63; e >>= 6;
    const int s = (int)e;
    const int r = lane & 31, h = lane >> 5;
    float w = 0.f;
    if (r < a.out_features) w = a.params[a.off_w[L] + (long)r * WD + 16 * s + pi_perm(h, j)] * a.wscale;
    a.wf_last[gid] = a.fwd_is_f16 ? to_f16(w) : to_bf16(w);
  }
  if (gid < 32) {
    float* piece = reinterpret_cast<float*>(a.wf_last + (long)KS * 512);
    piece[gid] = (gid < a.out_features) ? a.params[a.off_b[L] + gid] * a.wscale : 0.f;
  }
  if (gid < (long)NT * 64 * 8) {  // last layer backward image: NT tiles x 1 k-step
    long e = gid;
    const int j = e & 7; e >>= 3;
    const int lane = e & 63; e >>= 6;
    const int tile = (int)e;
    const int r = lane & 31, h = lane >> 5;
    const int c = pi_perm(h, j);
    float w = 0.f;
    if (c < a.out_features)
      w = a.params[a.off_w[L] + (long)c * WD + 32 * tile + r] * ((L - 1 == 0 ? a.om_first : a.om_hidden) * (a.link ? a.link[L] : 1.0f));
    a.wb_last[gid] = a.fwd_is_f16 ? to_f16(w) : to_bf16(w);
  }
  if (gid < WD) {
    const float* W0 = a.params + a.off_w[0];
    a.l0tab[gid] = f32x4{W0[gid * 2], W0[gid * 2 + 1], a.params[a.off_b[0] + gid], 0.f};
  }
  if (a.l0img && gid < (long)NT * 64) {   // one lane (8 halves) of the layer-0 image per thread
    const int lane = (int)(gid & 63), tile = (int)(gid >> 6), r = lane & 31, n = 32 * tile + r;
    u32x4 v = u32x4{0u, 0u, 0u, 0u};
    if ((lane >> 5) == 0) {
      const float* W0 = a.params + a.off_w[0];
      _Float16 h0, l0, h1, l1, hb, lb;
      split_f16(W0[n * 2] * a.sc_first, kL0Split, h0, l0);
      split_f16(W0[n * 2 + 1] * a.sc_first, kL0Split, h1, l1);
      split_f16(a.params[a.off_b[0] + n] * a.sc_first, kL0Split, hb, lb);
      const _Float16 s0 = (_Float16)((float)h0 * (1.0f / kL0Split)), s1 = (_Float16)((float)h1 * (1.0f / kL0Split));
      v = u32x4{pack_h2(h0, s0), pack_h2(l0, h1), pack_h2(s1, l1), pack_h2(hb, lb)};
    }
    reinterpret_cast<u32x4*>(a.l0img)[gid] = v;
  }
}

}  // namespace sf
