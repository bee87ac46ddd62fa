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
//   k_bwdx     fused backward-data chain: delta_l -> delta_{l-1} on MFMA, cos recomputed from the phase.
//   k_dw       weight-gradient GEMM per layer, contraction over pixels: dW_l = delta_l^T * sin(phase_{l-1}),
//              operands staged in LDS and read with ds_read_b64_tr_b16, persistent 256x256 accumulators.
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
  static DEV f32x16 mfma(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0,
                                                   0, 0);
  }
};
struct OpF16 {
  static DEV uint32_t pack2(float a, float b) {
    f32x2 v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2));
  }
  static DEV f32x16 mfma(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0,
                                                  0);
  }
};

DEV uint32_t pack_phase2(float f0, float f1) {  // two fractions in [0,1) -> two unorm16 (x*65535, RNE)
  return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pknorm_u16(f0, f1));
}
DEV float bf16_lo(uint32_t u) { return __builtin_bit_cast(float, u << 16); }
DEV float bf16_hi(uint32_t u) { return __builtin_bit_cast(float, u & 0xffff0000u); }

constexpr float kInv65535 = 1.0f / 65535.0f;

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
  const u32x4* wf;         // (depth-2) forward images, WD*WD/8 pieces each
  const u32x4* wf_last;    // forward image of the last layer padded to 32 rows: WD/16 * 64 pieces
  const float* bias_h;     // (depth-2) x WD, pre-multiplied by the weight-image scale
  const float* bias_last;  // [32], pre-multiplied by the weight-image scale, rows >= out_features zero
  float sc_first;          // first_omega_0 / (2 pi)
  float sc_hidden;         // hidden_omega_0 / (2 pi) / weight-image scale
  float sc_last;           // 1 / weight-image scale
  u32x4* P;                // phases, layer l at P + l*p_stride, F-layout
  long p_stride;           // pieces per layer in the scratch
  u32x4* Dlast;            // delta of the last layer, F-layout with 2 k-steps (32 padded neurons), bf16
  const float* img;        // [npix][3] target
  float gscale;            // 1/(3*H*W): d(mse)/d(out) = (pred-img) * gscale   (the /2 of siren.py:131 folded in)
  float* pred;             // optional [npix][3]
  float* sse_part;         // [gridDim.x] per-workgroup sum of squared residuals
};

template <int WD, typename OP, bool TRAIN>
__global__ __launch_bounds__(512) void k_fwd(FwdArgs a) {
  constexpr int NT = WD / 32, KS = WD / 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  u32x4* sW = reinterpret_cast<u32x4*>(smem);                               // NT*KS*64 pieces
  f32x4* sL0 = reinterpret_cast<f32x4*>(smem + (size_t)NT * KS * 1024);     // WD
  float* sBias = reinterpret_cast<float*>(sL0 + WD);                        // WD (>= 32)
  float* sRed = sBias + (WD < 32 ? 32 : WD);                                // 8

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, m = lane & 31, h = lane >> 5;
  for (int i = tid; i < WD; i += 512) sL0[i] = a.l0tab[i];
  __syncthreads();

  const long pb = (long)blockIdx.x * kWavesFwd + wave;  // pixel block inside the chunk
  const long pix = a.pix0 + pb * 32 + m;
  const bool valid = pix < a.npix;
  const long pc = valid ? pix : a.npix - 1;
  const int row = (int)(pc / a.W), col = (int)(pc - (long)row * a.W);
  const float x0 = (a.gh[a.row_begin + row] - 0.5f) * 2.0f;  // siren.py:128
  const float x1 = (a.gw[col] - 0.5f) * 2.0f;

  u32x4 B[KS];
  // ---- layer 0: K = 2, f32 VALU, written straight into B-fragment order -----------------------
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    float av[8], ph[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const f32x4 t = sL0[16 * s + pi_perm(h, j)];
      const float z = __builtin_fmaf(t.y, x1, __builtin_fmaf(t.x, x0, t.z));
      const float tt = z * a.sc_first;
      av[j] = __builtin_amdgcn_sinf(tt);
      ph[j] = __builtin_amdgcn_fractf(tt);
    }
    B[s] = u32x4{OP::pack2(av[0], av[1]), OP::pack2(av[2], av[3]), OP::pack2(av[4], av[5]), OP::pack2(av[6], av[7])};
    if (TRAIN)
      a.P[(pb * KS + s) * 64 + lane] = u32x4{pack_phase2(ph[0], ph[1]), pack_phase2(ph[2], ph[3]),
                                             pack_phase2(ph[4], ph[5]), pack_phase2(ph[6], ph[7])};
  }

  // ---- hidden layers: [WD x WD] on MFMA, activations stay in registers ------------------------
  for (int l = 1; l <= a.depth - 2; ++l) {
    __syncthreads();
    {
      const u32x4* src = a.wf + (size_t)(l - 1) * NT * KS * 64;
#pragma unroll
      for (int i = tid; i < NT * KS * 64; i += 512) sW[i] = src[i];
      if (tid < WD) sBias[tid] = a.bias_h[(l - 1) * WD + tid];
    }
    __syncthreads();
    u32x4 Bn[KS];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      f32x16 acc;
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(&sBias[32 * nt + 8 * q4 + 4 * h]);
        acc[4 * q4 + 0] = b.x; acc[4 * q4 + 1] = b.y; acc[4 * q4 + 2] = b.z; acc[4 * q4 + 3] = b.w;
      }
#pragma unroll
      for (int s = 0; s < KS; ++s) acc = OP::mfma(sW[(nt * KS + s) * 64 + lane], B[s], acc);
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        float av[8], ph[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float tt = acc[8 * q + j] * a.sc_hidden;
          av[j] = __builtin_amdgcn_sinf(tt);
          ph[j] = __builtin_amdgcn_fractf(tt);
        }
        Bn[2 * nt + q] =
            u32x4{OP::pack2(av[0], av[1]), OP::pack2(av[2], av[3]), OP::pack2(av[4], av[5]), OP::pack2(av[6], av[7])};
        if (TRAIN)
          a.P[(size_t)l * a.p_stride + (pb * KS + 2 * nt + q) * 64 + lane] =
              u32x4{pack_phase2(ph[0], ph[1]), pack_phase2(ph[2], ph[3]), pack_phase2(ph[4], ph[5]),
                    pack_phase2(ph[6], ph[7])};
      }
    }
#pragma unroll
    for (int s = 0; s < KS; ++s) B[s] = Bn[s];
  }

  // ---- last layer (out_features <= 3, padded to one 32-row tile) + residual ---------------------
  __syncthreads();
  for (int i = tid; i < KS * 64; i += 512) sW[i] = a.wf_last[i];
  if (tid < 32) sBias[tid] = a.bias_last[tid];
  __syncthreads();
  f32x16 acc;
#pragma unroll
  for (int q4 = 0; q4 < 4; ++q4) {
    const f32x4 b = *reinterpret_cast<const f32x4*>(&sBias[8 * q4 + 4 * h]);
    acc[4 * q4 + 0] = b.x; acc[4 * q4 + 1] = b.y; acc[4 * q4 + 2] = b.z; acc[4 * q4 + 3] = b.w;
  }
#pragma unroll
  for (int s = 0; s < KS; ++s) acc = OP::mfma(sW[s * 64 + lane], B[s], acc);

  float sse = 0.f;
  float d[3] = {0.f, 0.f, 0.f};
  if (h == 0 && valid) {  // rows 0..2 of the tile live in registers 0..2 of the lower lane half
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float p = acc[c] * a.sc_last * 0.5f + 0.5f;  // siren.py:131
      if (a.pred) a.pred[pix * 3 + c] = p;
      if (a.img) {
        const float r = p - a.img[pix * 3 + c];
        sse += r * r;
        d[c] = r * a.gscale;
      }
    }
  }
  if (TRAIN) {
    // dL/dout in F-layout (k-step 0: neurons PI(0,j) = j for j < 4), second k-step zero
    a.Dlast[(pb * 2 + 0) * 64 + lane] = u32x4{OpBF16::pack2(d[0], d[1]), OpBF16::pack2(d[2], 0.f), 0u, 0u};
    a.Dlast[(pb * 2 + 1) * 64 + lane] = u32x4{0u, 0u, 0u, 0u};
  }
  // workgroup SSE partial (fixed order: lanes by xor-shuffle, then waves 0..7)
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

// ---------------------------------------------------------------------------------------------
// k_bwdx: delta chain.  delta_l = dL/dz_l (z = pre-activation).  delta_{l-1} = (delta_l W_l) * omega cos(omega z_{l-1})
// ---------------------------------------------------------------------------------------------
struct BwdArgs {
  int depth;
  const u32x4* wb;        // backward images of layers 1..depth-2 (WD*WD/8 pieces each), bf16
  const u32x4* wb_last;   // backward image of the last layer: NT tiles x 1 k-step x 64 pieces
  const u32x4* P;         // phases
  long p_stride;
  const u32x4* Dlast;     // F-layout, 2 k-steps
  u32x4* D;               // deltas of layers 0..depth-2, layer l at D + l*p_stride, F-layout bf16
  float om_first, om_hidden;  // radians: d sin(om z)/dz = om cos(om z)
};

template <int WD>
__global__ __launch_bounds__(512) void k_bwdx(BwdArgs a) {
  constexpr int NT = WD / 32, KS = WD / 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  u32x4* sW = reinterpret_cast<u32x4*>(smem);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const long pb = (long)blockIdx.x * kWavesFwd + wave;

  u32x4 B[KS];
  // ---- through the last layer: G = delta_last * W_last (one k-step, padded) ---------------------
  for (int i = tid; i < NT * 64; i += 512) sW[i] = a.wb_last[i];
  __syncthreads();
  {
    const u32x4 b0 = a.Dlast[(pb * 2) * 64 + lane];
    const int l = a.depth - 2;  // producing delta of layer depth-2
    const float om = (l == 0) ? a.om_first : a.om_hidden;
#pragma unroll
    for (int it = 0; it < NT; ++it) {
      f32x16 acc = {};
      acc = OpBF16::mfma(sW[it * 64 + lane], b0, acc);
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const u32x4 p = a.P[(size_t)l * a.p_stride + (pb * KS + 2 * it + q) * 64 + lane];
        float dv[8];
#pragma unroll
        for (int j2 = 0; j2 < 4; ++j2) {
          const float c0 = __builtin_amdgcn_cosf((float)(p[j2] & 0xffffu) * kInv65535);
          const float c1 = __builtin_amdgcn_cosf((float)(p[j2] >> 16) * kInv65535);
          dv[2 * j2] = acc[8 * q + 2 * j2] * (om * c0);
          dv[2 * j2 + 1] = acc[8 * q + 2 * j2 + 1] * (om * c1);
        }
        const u32x4 o = u32x4{OpBF16::pack2(dv[0], dv[1]), OpBF16::pack2(dv[2], dv[3]), OpBF16::pack2(dv[4], dv[5]),
                              OpBF16::pack2(dv[6], dv[7])};
        B[2 * it + q] = o;
        a.D[(size_t)l * a.p_stride + (pb * KS + 2 * it + q) * 64 + lane] = o;
      }
    }
  }
  // ---- hidden layers, from depth-2 down to 1: produces delta_{l-1} ------------------------------
  for (int l = a.depth - 2; l >= 1; --l) {
    __syncthreads();
    {
      const u32x4* src = a.wb + (size_t)(l - 1) * NT * KS * 64;
#pragma unroll
      for (int i = tid; i < NT * KS * 64; i += 512) sW[i] = src[i];
    }
    __syncthreads();
    const float om = (l - 1 == 0) ? a.om_first : a.om_hidden;
    u32x4 Bn[KS];
#pragma unroll
    for (int it = 0; it < NT; ++it) {
      f32x16 acc = {};
#pragma unroll
      for (int s = 0; s < KS; ++s) acc = OpBF16::mfma(sW[(it * KS + s) * 64 + lane], B[s], acc);
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const u32x4 p = a.P[(size_t)(l - 1) * a.p_stride + (pb * KS + 2 * it + q) * 64 + lane];
        float dv[8];
#pragma unroll
        for (int j2 = 0; j2 < 4; ++j2) {
          const float c0 = __builtin_amdgcn_cosf((float)(p[j2] & 0xffffu) * kInv65535);
          const float c1 = __builtin_amdgcn_cosf((float)(p[j2] >> 16) * kInv65535);
          dv[2 * j2] = acc[8 * q + 2 * j2] * (om * c0);
          dv[2 * j2 + 1] = acc[8 * q + 2 * j2 + 1] * (om * c1);
        }
        const u32x4 o = u32x4{OpBF16::pack2(dv[0], dv[1]), OpBF16::pack2(dv[2], dv[3]), OpBF16::pack2(dv[4], dv[5]),
                              OpBF16::pack2(dv[6], dv[7])};
        Bn[2 * it + q] = o;
        a.D[(size_t)(l - 1) * a.p_stride + (pb * KS + 2 * it + q) * 64 + lane] = o;
      }
    }
#pragma unroll
    for (int s = 0; s < KS; ++s) B[s] = Bn[s];
  }
}

// ---------------------------------------------------------------------------------------------
// k_dw: dW[j][i] = sum_pixels delta[pix][j] * act[pix][i],  db[j] = sum_pixels delta[pix][j]
//   A source: delta image (F-layout, KSA = rows/16 k-steps per block), rows = output neurons j
//   B source: BSRC 0: phase image of the previous layer (F-layout) -> sin -> bf16
//             BSRC 1: the two coordinates (layer 0): columns {x0_hi, x0_lo, x1_hi, x1_lo, 0...}
//                     (hi + lo bf16 split keeps 16 significant bits of the coordinate)
//   Each workgroup sweeps a contiguous range of pixel blocks and keeps its ROWS x COLS accumulator in
//   registers; the per-workgroup result goes to a slab that k_reduce sums in a fixed order.
// ---------------------------------------------------------------------------------------------
struct DwArgs {
  const u32x4* A;      // delta image
  const u32x4* Bp;     // phase image (BSRC 0)
  const float* gh;     // coords (BSRC 1)
  const float* gw;
  int W, row_begin;
  long pix0, npix;
  long n_pb;           // pixel blocks (of 32) in this chunk
  int pb_per_wg;       // blocks per workgroup (multiple of PBS)
  float* slab;         // [gridDim.x][ROWS*COLS + ROWS]
};

// address (in bytes, relative to an F-layout block of KS k-steps) that lane must supply to
// ds_read_b64_tr_b16 for fragment tile `tile` (32 neurons), pixel k-step kk (16 pixels), half-read u.
// Returned fragment: lane (r = lane&31 -> neuron 32*tile + r, hq = lane>>5), elements = pixels
// 16*kk + 8*hq + 4*u + {0..3}.
DEV int tr_addr(int tile, int kk, int u, int lane) {
  const int g = lane >> 4, gs = g & 1, hq = g >> 1, li = lane & 15, q = li >> 2, p = li & 3;
  const int mpix = 16 * kk + 8 * hq + 4 * u + q;
  const int piece = (2 * tile + gs) * 64 + (p & 1) * 32 + mpix;
  return piece * 16 + 8 * (p >> 1);
}

DEV u32x4 ds_read_tr_pair(const char* base, int off0, int off1) {
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const s16x4 r0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + off0));
  const s16x4 r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + off1));
  const u32x2 a0 = __builtin_bit_cast(u32x2, r0), a1 = __builtin_bit_cast(u32x2, r1);
  return u32x4{a0.x, a0.y, a1.x, a1.y};
}

template <int ROWS, int COLS, int WAVES_R, int WAVES_C, int BSRC>
__global__ __launch_bounds__(WAVES_R* WAVES_C * 64) void k_dw(DwArgs a) {
  constexpr int NW = WAVES_R * WAVES_C, NTHR = NW * 64;
  constexpr int JT = ROWS / 32, IT = COLS / 32;
  constexpr int WJ = JT / WAVES_R, WI = IT / WAVES_C;   // tiles per wave
  constexpr int KSA = ROWS / 16, KSB = COLS / 16;
  constexpr int PBS = 2;                                 // pixel blocks staged per iteration (64 px)
  static_assert(JT % WAVES_R == 0 && IT % WAVES_C == 0, "tiling");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sA = smem;                                       // PBS*KSA*1024
  char* sB = smem + PBS * KSA * 1024;                    // PBS*KSB*1024
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wr = wave / WAVES_C, wc = wave % WAVES_C;

  f32x16 acc[WJ][WI];
#pragma unroll
  for (int x = 0; x < WJ; ++x)
#pragma unroll
    for (int y = 0; y < WI; ++y) acc[x][y] = f32x16{};
  float dbs[WJ];
#pragma unroll
  for (int x = 0; x < WJ; ++x) dbs[x] = 0.f;

  const long pb_begin = (long)blockIdx.x * a.pb_per_wg;
  long pb_end = pb_begin + a.pb_per_wg;
  if (pb_end > a.n_pb) pb_end = a.n_pb;

  for (long pb = pb_begin; pb < pb_end; pb += PBS) {
    __syncthreads();
    // ---- stage A (deltas): straight copy of PBS*KSA pieces ----
    for (int i = tid; i < PBS * KSA * 64; i += NTHR) {
      const long blk = pb + i / (KSA * 64);
      u32x4 v = u32x4{0u, 0u, 0u, 0u};
      if (blk < pb_end) v = a.A[pb * KSA * 64 + i];
      reinterpret_cast<u32x4*>(sA)[i] = v;
    }
    // ---- stage B ----
    if (BSRC == 0) {
      for (int i = tid; i < PBS * KSB * 64; i += NTHR) {
        const long blk = pb + i / (KSB * 64);
        u32x4 o = u32x4{0u, 0u, 0u, 0u};
        if (blk < pb_end) {
          const u32x4 p = a.Bp[pb * KSB * 64 + i];
#pragma unroll
          for (int j2 = 0; j2 < 4; ++j2) {
            const float s0 = __builtin_amdgcn_sinf((float)(p[j2] & 0xffffu) * kInv65535);
            const float s1 = __builtin_amdgcn_sinf((float)(p[j2] >> 16) * kInv65535);
            o[j2] = OpBF16::pack2(s0, s1);
          }
        }
        reinterpret_cast<u32x4*>(sB)[i] = o;
      }
    } else {
      // one 32-neuron tile (2 k-steps): k-step 0, h=0, j=0..3 -> columns 0..3 = x0_hi,x0_lo,x1_hi,x1_lo
      for (int i = tid; i < PBS * KSB * 64; i += NTHR) {
        const int bi = i / (KSB * 64), r = i % (KSB * 64), s = r / 64, ln = r % 64;
        u32x4 o = u32x4{0u, 0u, 0u, 0u};
        if (s == 0 && ln < 32 && pb + bi < pb_end) {
          long pix = a.pix0 + (pb + bi) * 32 + ln;
          if (pix >= a.npix) pix = a.npix - 1;
          const int row = (int)(pix / a.W), col = (int)(pix - (long)row * a.W);
          const float x0 = (a.gh[a.row_begin + row] - 0.5f) * 2.0f, x1 = (a.gw[col] - 0.5f) * 2.0f;
          const float x0h = bf16_lo(OpBF16::pack2(x0, 0.f)), x1h = bf16_lo(OpBF16::pack2(x1, 0.f));
          o[0] = OpBF16::pack2(x0h, x0 - x0h);
          o[1] = OpBF16::pack2(x1h, x1 - x1h);
        }
        reinterpret_cast<u32x4*>(sB)[i] = o;
      }
    }
    __syncthreads();
    // ---- MFMA over the staged pixels: k-steps of 16 pixels ----
#pragma unroll
    for (int bi = 0; bi < PBS; ++bi) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        u32x4 fa[WJ], fb[WI];
#pragma unroll
        for (int x = 0; x < WJ; ++x)
          fa[x] = ds_read_tr_pair(sA + bi * KSA * 1024, tr_addr(wr * WJ + x, kk, 0, lane),
                                  tr_addr(wr * WJ + x, kk, 1, lane));
#pragma unroll
        for (int y = 0; y < WI; ++y)
          fb[y] = ds_read_tr_pair(sB + bi * KSB * 1024, tr_addr(wc * WI + y, kk, 0, lane),
                                  tr_addr(wc * WI + y, kk, 1, lane));
#pragma unroll
        for (int x = 0; x < WJ; ++x)
#pragma unroll
          for (int y = 0; y < WI; ++y) acc[x][y] = OpBF16::mfma(fa[x], fb[y], acc[x][y]);
        if (wc == 0) {
#pragma unroll
          for (int x = 0; x < WJ; ++x) {
            float t = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) t += bf16_lo(fa[x][e]) + bf16_hi(fa[x][e]);
            dbs[x] += t;
          }
        }
      }
    }
  }
  // ---- write the slab: dW tile (jt, it): lane col = lane&31, reg t row = rho(t, lane>>5) ----
  float* slab = a.slab + (size_t)blockIdx.x * (ROWS * COLS + ROWS);
  const int cl = lane & 31, hh = lane >> 5;
#pragma unroll
  for (int x = 0; x < WJ; ++x)
#pragma unroll
    for (int y = 0; y < WI; ++y)
#pragma unroll
      for (int t = 0; t < 16; ++t)
        slab[(size_t)(32 * (wr * WJ + x) + rho(t, hh)) * COLS + 32 * (wc * WI + y) + cl] = acc[x][y][t];
  if (wc == 0) {
#pragma unroll
    for (int x = 0; x < WJ; ++x) {
      const float t = dbs[x] + __shfl_xor(dbs[x], 32);
      if (hh == 0) slab[ROWS * COLS + 32 * (wr * WJ + x) + cl] = t;
    }
  }
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
};

__global__ void k_reduce(ReduceArgs a) {
  const long slab_sz = (long)a.slab_rows * a.slab_cols + a.slab_rows;
  const int nW = a.rows_out * a.cols_out;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= nW + a.rows_out) return;
  if (idx < nW) {
    const int j = idx / a.cols_out, i = idx % a.cols_out;
    float s = 0.f;
    if (a.mode == 0) {
      const float* p = a.slab + (long)j * a.slab_cols + i;
      for (int w = 0; w < a.n_wg; ++w) s += p[w * slab_sz];
    } else {
      const float* p = a.slab + (long)j * a.slab_cols + 2 * i;
      for (int w = 0; w < a.n_wg; ++w) s += p[w * slab_sz] + p[w * slab_sz + 1];
    }
    a.gW[idx] = a.accumulate ? a.gW[idx] + s : s;
  } else {
    const int j = idx - nW;
    const float* p = a.slab + (long)a.slab_rows * a.slab_cols + j;
    float s = 0.f;
    for (int w = 0; w < a.n_wg; ++w) s += p[w * slab_sz];
    a.gb[j] = a.accumulate ? a.gb[j] + s : s;
  }
}

// sum of the per-workgroup SSE partials in double, fixed order
__global__ void k_sse_reduce(const float* part, int n, double* out) {
  __shared__ double sh[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) s += (double)part[i];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) *out = sh[0];
}

// ---------------------------------------------------------------------------------------------
// k_adam: torch.optim.Adam single-tensor op order (torch 2.x): lerp for exp_avg, mul/addcmul for
// exp_avg_sq, denom = sqrt(v)/sqrt(bc2) + eps, p -= step_size * m / denom; then Masking.apply_mask.
// ---------------------------------------------------------------------------------------------
struct AdamArgs {
  float* p; const float* g; float* m; float* v; const float* mask;
  long n;
  float beta1, beta2, eps, step_size, bc2_sqrt;
};
__global__ void k_adam(AdamArgs a) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  const float g = a.g[i];
  float m = a.m[i], v = a.v[i];
  m = m + (1.0f - a.beta1) * (g - m);
  v = v * a.beta2 + (1.0f - a.beta2) * g * g;
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
  float wscale;             // forward-image scale (power of two)
  int fwd_is_f16;
  uint16_t* wf; uint16_t* wf_last; uint16_t* wb; uint16_t* wb_last;
  f32x4* l0tab; float* bias_h; float* bias_last;
};
DEV uint16_t to_bf16(float x) { return (uint16_t)(OpBF16::pack2(x, 0.f) & 0xffffu); }
DEV uint16_t to_f16(float x) { return (uint16_t)(OpF16::pack2(x, 0.f) & 0xffffu); }

__global__ void k_images(ImgArgs a) {
  const int WD = a.WD, NT = WD / 32, KS = WD / 16;
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
    const float wfwd = Wl[(long)(32 * tile + r) * WD + 16 * s + pi_perm(h, j)] * a.wscale;
    const float wbwd = Wl[(long)(16 * s + pi_perm(h, j)) * WD + 32 * tile + r];
    a.wf[gid] = a.fwd_is_f16 ? to_f16(wfwd) : to_bf16(wfwd);
    a.wb[gid] = to_bf16(wbwd);
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
  if (gid < (long)NT * 64 * 8) {  // last layer backward image: NT tiles x 1 k-step
    long e = gid;
    const int j = e & 7; e >>= 3;
    const int lane = e & 63; e >>= 6;
    const int tile = (int)e;
    const int r = lane & 31, h = lane >> 5;
    const int c = pi_perm(h, j);
    float w = 0.f;
    if (c < a.out_features) w = a.params[a.off_w[L] + (long)c * WD + 32 * tile + r];
    a.wb_last[gid] = to_bf16(w);
  }
  if (gid < WD) {
    const float* W0 = a.params + a.off_w[0];
    a.l0tab[gid] = f32x4{W0[gid * 2], W0[gid * 2 + 1], a.params[a.off_b[0] + gid], 0.f};
  }
  if (gid < (long)(a.depth - 2) * WD) {
    const int l = (int)(gid / WD) + 1, n = (int)(gid % WD);
    a.bias_h[gid] = a.params[a.off_b[l] + n] * a.wscale;
  }
  if (gid < 32) a.bias_last[gid] = (gid < a.out_features) ? a.params[a.off_b[L] + gid] * a.wscale : 0.f;
}

}  // namespace sf
