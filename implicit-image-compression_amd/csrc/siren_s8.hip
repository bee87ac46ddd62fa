// siren_s8.hip — backward kernels of the 8-BIT SCRATCH path (sf_config.scratch_format = 8, hidden <= 256).
//
// Same arithmetic as k_bwd / k_dw0 of siren_kernels.hip (the autograd backward of implicit_image/models/siren.py:56-68
// under F.mse_loss, implicit_image/utils/train_helper.py:147-161), but the two tensors a layer's backward re-reads
// from HBM are one byte per value instead of two:
//   phases  P_l   : u8 = round(t * 256) mod 256 (t in revolutions), written by k_fwd<.., S8>, one 1 KiB piece per
//                   (32-pixel block, 32-neuron tile): byte 8q+j of lane (h,m) = neuron 32*tile + 16*q + PI(h,j)
//   deltas  D_l   : OCP fp8 e4m3 in the same piece layout, in units of the step's power-of-two gradient pre-scale
//                   (k_bwd8<LAST> scales the chunk's residual rms to kFp8Target, so e4m3's 2^-9 .. 448 sits around the data)
// => k_bwd8 moves 768 B per pixel (delta and phase in, delta out) where k_bwd moves 1 536: the round-1 kernel ran
// at ~75 % of the box's achievable HBM rate, i.e. its time WAS its bytes.
//
// MFMA operands stay fp16: a byte piece is expanded ONCE per workgroup into the 16-bit F-layout image the round-1
// code already reads (lane-linear for the data-gradient B operand, ds_read_b64_tr_b16 for both weight-gradient
// operands), one block ahead of its use, by the wave that also issued its DMA (8 conversions per wave and block).
//
// LDS of one workgroup (256 x 256 layer, 160 KiB):
//   ring D   NB x  8 KiB   fp8 delta pieces   block k+NB is requested in step k, converted in step k+NB-1
//   ring P   NB x  8 KiB   phase-byte pieces  block k+NB-1 is requested in step k, decoded in step k+NB-1
//   D16      3  x 16 KiB   fp16 deltas        written in step k-1 (C), B operand in step k (X), tr-read in step k+1 (W)
//   S16      2  x 16 KiB   fp16 sin(phase)    written by the X epilogue in step k, tr-read in step k+1 (W)
// (included by siren_fit.hip after siren_kernels.hip)

#ifndef SF_BWD8_PF
#define SF_BWD8_PF 3      // B pieces read ahead in the 4-wave form (register ring of PF + 1)
#endif
#ifndef SF_BWD8_PF8
#define SF_BWD8_PF8 1     // the same in the 8-wave form
#endif
#ifndef SF_BWD8_FB2
#define SF_BWD8_FB2 0
#endif
#ifndef SF_BWD8_FA2
#define SF_BWD8_FA2 0      // 8-wave form: second delta^T fragment buffer (4 registers it does not have)
#endif

namespace sf {

struct Bwd8Args {
  const u32x4* D;       // LAST: dL/dout, 16-bit float, ONE piece (k-step 0) per pixel block; else fp8 deltas of layer l,
                        // JW/32 pieces per pixel block
  const u32x4* P;       // phase bytes of layer l-1, IW/32 pieces per pixel block (unused by the P0 variant)
  u32x4* Dout;          // fp8 deltas of layer l-1, IW/32 pieces per pixel block
  const u32x4* wb;      // backward weight image of layer l (fp16, omega folded in): tiles (IW/32) x k-steps x 64 lanes
  long n_pb;            // pixel blocks in this chunk
  float* slab;          // [gridDim.x][JW*IW + JW]
  // P0 variant (layer l-1 is layer 0): phases re-derived from the pixel coordinates
  const f32x4* l0tab;
  long pix0, npix;
  int W, row_begin;
  unsigned long long w_magic;
  float inv_hm1, inv_wm1;
  float sc_first;
  // LAST: adaptive delta scale of this chunk.  dL/dout arrives as residual * res_scale (static, fp16); the kernel sums
  // the chunk's SSE partials (every workgroup the same n_part floats in the same order => the same value everywhere),
  // picks G = 2^floor(log2(target / rms(residual))) and multiplies the first hidden delta by G / res_scale; workgroup 0
  // publishes {G / res_scale, 1 / (G * n_values)} for the slab reductions of the layers below.
  const float* sse_part; int n_part;
  double inv_chunk_values;     // 1 / (out_features * pixels of this chunk)
  double n_values;             // out_features * H * W of the full image
  float res_scale, target;
  float* scale_out;
  float* dbg;                  // SF_EXPERIMENT_STAMP builds only
  const u32x4* zeros;          // k_bwd8h: 1 KiB of zeros (delta pieces beyond the last block)
  u32x4* dump;                 // k_bwd8h: 8 KiB nobody reads (deltas of the blocks beyond the last)
};

// operand fragment as it sits in the register ring before the MFMA: 16 bytes of 16-bit floats, or 8 fp8 bytes (IN8)
template <bool BYTES> struct RawFrag { typedef u32x4 type; };
template <> struct RawFrag<true> { typedef u32x2 type; };
DEV u32x4 frag_of(u32x4 r) { return r; }
DEV u32x4 frag_of(u32x2 r) { return fp8x8_to_f16(r.x, r.y); }
// row r of an A fragment read with ds_read_b64_tr_b8 through tr8 lane bases = this neuron of the 32-neuron tile
SF_HOSTDEV int nu8(int r) { return 16 * (r >> 4) + pi_perm((r >> 3) & 1, r & 7); }

template <int JW, int IW, int WAVES_R, int WAVES_C, bool LAST, bool P0, typename OP, int NB, int PARK = 0, int NBP_ = 0, bool D8 = true>
__global__ __launch_bounds__(WAVES_R* WAVES_C * 64) void k_bwd8(Bwd8Args a) {
  constexpr int NW = WAVES_R * WAVES_C;
  constexpr int JT = JW / 32, IT = IW / 32;
  constexpr int WJ = JT / WAVES_R, WI = IT / WAVES_C;
  constexpr int KSJ = JW / 16, KSI = IW / 16;
  constexpr int KSX = LAST ? 1 : KSJ;
  constexpr int XT = IT / NW;
  static_assert(IT % NW == 0, "phase-X tiling needs NW <= IT");
  constexpr int NBP = NBP_ > 0 ? NBP_ : NB;     // slots of the phase ring (NB: slots of the delta ring)

  // D8 = false (scratch_format 12): deltas stay 16-bit floats (round-1 F-layout: one piece per k-step) - the ring slot
  // IS the image both products read (X lane-linear at step k, W transposed at step k+1), nothing is converted, and only
  // the phases are bytes.
  // D8 = true (scratch_format 8): the deltas a hidden layer READS are fp8 byte pieces (one piece per 32-neuron tile:
  // byte 8q+j of lane (h,m) = neuron 32 nt + 16 q + PI(h,j)), IN8; the ring slot holds the bytes and both products
  // read it (fp8 -> fp16 is exact): phase W takes one ds_read_b64_tr_b8 per delta^T fragment and converts in registers (every
  // wave has its own row tiles: nothing is converted twice); phase X needs every k-step in every wave, so the block is
  // expanded ONCE, one step ahead, into a 16-bit image (X16, two buffers) that all waves read lane-linearly - converting
  // in registers there cost 8 x 64 conversions per block instead of 64 and made the kernel VALU-bound.  The LAST layer
  // reads the 16-bit dL/dout piece k_fwd wrote, exactly as with D8 = false; only what it WRITES is fp8.
  constexpr bool IN8 = D8 && !LAST;
  constexpr int DPC = IN8 ? JT : (LAST ? 2 : KSJ);               // 1 KiB pieces per block in ring D
  constexpr int DDMA = IN8 ? JT : (LAST ? 1 : KSJ);              // of which the DMA fills (LAST: the second is the zero k-step)
  constexpr int TSTR = IN8 ? 1024 : 2048;                        // bytes of a 32-neuron tile inside a ring slot
  constexpr int PPC = P0 ? 0 : IT;            // 1 KiB pieces per block in ring P
  constexpr int GD = DDMA / NW, GP = PPC / NW; // LDS-DMA instructions EVERY wave issues per block (lower bounds)
  constexpr int S_ST = 2 * XT;                // delta stores per wave per block (one 8-byte half element per k-step of a row tile)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const rD = smem;
  char* const rP = rD + NB * DPC * 1024;
  char* const x16 = rP + NBP * PPC * 1024;                       // IN8: two 16-bit images of a block's deltas (B operand of phase X)
  char* const s16 = x16 + (IN8 ? 2 : 0) * KSJ * 1024;
  char* const wsp0 = s16 + 2 * KSI * 1024;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / WAVES_C, wc = wave % WAVES_C;
  const int xit0 = wave * XT;
  const int trb = tr_lane_base(lane);
  const int lsw = sw_lane(lane, 0);      // slot of this lane's register image in an even 16-bit piece; odd: lsw ^ 8
  // byte pieces (IN8): slot i of a ring piece holds the element of lane i ^ 8 (i >> 5) - the upper lane half is shifted by
  // eight pixels, i.e. by 32 banks, so that the even (h' = 0) and odd (h' = 1) row suppliers of a ds_read_b64_tr_b8 group
  // do not meet on the same banks (2-way conflict on every transposed byte read otherwise)
  const int lsw8 = lane ^ ((lane >> 5) << 3);

  // stationary W^T rows of this wave (as k_bwd): k-steps [0, KSR) in registers, the rest parked in LDS
  // (one wave per SIMD has 512 registers: nothing is parked then)
  constexpr int KSR = KSX - PARK;               // PARK k-steps of every row tile live in LDS behind the buffers
  constexpr int WSP = PARK;
  char* sWsp = wsp0 + (size_t)wave * XT * WSP * 1024;
  u32x4 wreg[XT][KSR];
#pragma unroll
  for (int x = 0; x < XT; ++x) {
#pragma unroll
    for (int s = 0; s < KSR; ++s) wreg[x][s] = a.wb[((xit0 + x) * KSX + s) * 64 + lane];
#pragma unroll
    for (int s = KSR; s < KSX; ++s)
      reinterpret_cast<u32x4*>(sWsp + (x * WSP + s - KSR) * 1024)[lane] = a.wb[((xit0 + x) * KSX + s) * 64 + lane];
  }
#ifndef SF_BWD8_TAB
#define SF_BWD8_TAB 1
#endif
  // A phase byte has 256 values: sin and cos of its decoded phase come from a 1 KiB table in LDS (one ds_read_b32 per value:
  // the pair as two 16-bit floats) instead of v_cvt_f32_ubyte + v_fma + v_sin + v_cos per value - the X epilogue is what
  // bounds phase W (VALU issue of two waves per SIMD).  The sine is the value the table-free form wrote to S16 bit for bit;
  // the cosine is rounded to 16 bits before it meets the accumulator (whose product is rounded to 8 or 16 bits anyway).
  constexpr bool TAB = SF_BWD8_TAB && !P0 && std::is_same<OP, OpF16>::value;
  uint32_t* const sTab = reinterpret_cast<uint32_t*>(wsp0 + (size_t)NW * XT * WSP * 1024);
  if (TAB) {
    for (int i = tid; i < 256; i += NW * 64) {
      const float r = __builtin_fmaf((float)i, 1.0f / 256.0f, kPhaseEps);
      sTab[i] = OP::pack2(__builtin_amdgcn_sinf(r), __builtin_amdgcn_cosf(r));
    }
  }
  const f32x4* sL0 = reinterpret_cast<const f32x4*>(wsp0 + (size_t)NW * XT * WSP * 1024);
  if (P0) {
    f32x4* dst = reinterpret_cast<f32x4*>(wsp0 + (size_t)NW * XT * WSP * 1024);
    for (int i = tid; i < IW; i += NW * 64) dst[i] = a.l0tab[i];
  }
  float dfac = 1.0f;   // LAST: chunk pre-scale / res_scale (power of two)
  if (LAST) {   // the zero k-step of dL/dout (padded neurons 16..31) lives in LDS only: second piece of every slot
    for (int b = wave; b < NB; b += NW) reinterpret_cast<u32x4*>(rD + (b * DPC + 1) * 1024)[lane] = u32x4{0u, 0u, 0u, 0u};
  }
  if (LAST && D8) {
    // chunk SSE: thread t sums partials t, t + NW*64, ... in double; threads are combined in index order
    if (D8) {
    double* red = reinterpret_cast<double*>(s16);     // NW*64 doubles <= 4 KiB, free until the first X epilogue
    double ps = 0.0;
    for (int i = tid; i < a.n_part; i += NW * 64) ps += (double)a.sse_part[i];
    red[tid] = ps;
    __syncthreads();
    if (tid == 0) {
      double t = 0.0;
      for (int i = 0; i < NW * 64; ++i) t += red[i];
      double rms = sqrt(t * a.inv_chunk_values);
      if (!(rms > 1e-12)) rms = 1e-12;                // also catches NaN
      if (rms > 4.0) rms = 4.0;
      const double G = exp2(floor(log2((double)a.target / rms)));
      red[0] = G;
    }
    __syncthreads();
    const double G = red[0];
    dfac = (float)(G / (double)a.res_scale);
    if (blockIdx.x == 0 && tid == 0) {
      a.scale_out[0] = dfac;
      a.scale_out[1] = (float)(1.0 / (G * a.n_values));
    }
    __syncthreads();
    }
  }

  f32x16 acc[WJ][WI];
#pragma unroll
  for (int x = 0; x < WJ; ++x)
#pragma unroll
    for (int y = 0; y < WI; ++y) acc[x][y] = f32x16{};
#ifndef SF_BWD8_DBSPLIT
#define SF_BWD8_DBSPLIT 1
#endif
  constexpr bool DBSPLIT = IN8 && SF_BWD8_DBSPLIT;   // bias-gradient row sums split over the column waves (IN8 form only: registers)
  float dbs[WJ];
#pragma unroll
  for (int x = 0; x < WJ; ++x) dbs[x] = 0.f;

  const long pb_begin = blockIdx.x, pb_step = gridDim.x;
  const int nblk = (int)((a.n_pb - pb_begin + pb_step - 1) / pb_step);

  auto stageD = [&](int k) {
    char* base = rD + (k % NB) * DPC * 1024;
    const long pb = pb_begin + k * pb_step;
    // (16-bit pieces are read transposed straight from the slot: bank swizzle applied on the SOURCE lane, see sw_lane)
    for (int pc = wave; pc < DDMA; pc += NW)
      glds16s(a.D + (pb * DDMA + pc) * 64, IN8 ? (uint32_t)lsw8 * 16u : ((uint32_t)lsw * 16u) ^ ((uint32_t)(pc & 1) << 7), base + pc * 1024);
  };
  auto stageP = [&](int k) {
    if (P0) return;
    char* base = rP + (k % NBP) * PPC * 1024;
    const long pb = pb_begin + k * pb_step;
    for (int pc = wave; pc < PPC; pc += NW) glds16s(a.P + (pb * IT + pc) * 64, (uint32_t)lane * 16u, base + pc * 1024);
  };
  // C(k): byte pieces of block k (ring D) -> 16-bit image X16[k & 1] (lane-linear pieces, one per k-step)
  auto convert = [&](int k) {
    if (!IN8) return;
    const char* src = rD + (k % NB) * DPC * 1024;
    char* dst = x16 + (k & 1) * KSJ * 1024;
    for (int pc = wave; pc < DPC; pc += NW) {
      const u32x4 raw = reinterpret_cast<const u32x4*>(src + pc * 1024)[lane];       // element of lane lsw8
      reinterpret_cast<u32x4*>(dst + (2 * pc) * 1024)[lsw8] = fp8x8_to_f16(raw.x, raw.y);
      reinterpret_cast<u32x4*>(dst + (2 * pc + 1) * 1024)[lsw8] = fp8x8_to_f16(raw.z, raw.w);
    }
  };
  auto pixel_xy = [&](int k, float& x0, float& x1) {
    long p = a.pix0 + (pb_begin + (long)k * pb_step) * 32 + (lane & 31);
    if (p >= a.npix) p = a.npix - 1;
    const unsigned row = (unsigned)(((unsigned long long)p * a.w_magic) >> 40);
    const unsigned col = (unsigned)(p - (long)row * a.W);
    x0 = ((float)(row + (unsigned)a.row_begin) * a.inv_hm1 - 0.5f) * 2.0f;
    x1 = ((float)col * a.inv_wm1 - 0.5f) * 2.0f;
  };

  constexpr int PF = NW <= 4 ? (KSX >= 4 ? SF_BWD8_PF : 1) : SF_BWD8_PF8;
  constexpr int NXB = PF + 1;
  constexpr int ESUB = 8;                       // epilogue slices per row tile: (q, quarter) groups of 2 values
  constexpr bool FB2 = SF_BWD8_FB2 && NW <= 4;  // both pixel k-steps of the activation fragments resident
#ifndef SF_BWD8_FA2_IN8
#define SF_BWD8_FA2_IN8 1
#endif
  constexpr bool FA2 = SF_BWD8_FA2 || NW <= 4 || (IN8 && SF_BWD8_FA2_IN8);  // delta^T fragments double-buffered
  constexpr int NWC = 2 * WJ;                   // W chunks per step
  constexpr int NE = XT * ESUB;                 // epilogue slices per step
  // ---- LDS addressing --------------------------------------------------------------------------------------
  // ds_* instructions take ONE address VGPR plus a 16-bit immediate.  Left to itself hipcc materialises a separate
  // loop-invariant VGPR for every (lane pattern, wave-uniform tile offset) pair it sees - 29 of them in this kernel,
  // which is what pushed the 8-wave form into scratch.  Here every access is written as
  //     per-step base VGPR (lane pattern + wave-uniform offset + buffer of this step)  +  compile-time immediate
  // with seven lane patterns kept across the loop and at most seven per-step bases alive at a time.
  const uint32_t aL1 = (uint32_t)lsw * 16u, aL1x = aL1 ^ 128u;       // piece element of this lane, even / odd k-step
  const uint32_t aLN = (uint32_t)lane * 16u;                         // lane-linear piece element
  const uint32_t aT1 = (uint32_t)trb, aT2 = aT1 ^ 64u;               // transposed-read lane bases (half-read 0 / 1)
  const uint32_t smem0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const uint32_t aTab = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem + (uint32_t)((NB * DPC + NBP * PPC + (IN8 ? 2 : 0) * KSJ + 2 * KSI + NW * XT * WSP) * 1024);
  const uint32_t oRP = NB * DPC * 1024, oX16 = oRP + NBP * PPC * 1024, oS16 = oX16 + (IN8 ? 2 : 0) * KSJ * 1024;
  // transposed byte reads (IN8): ds_read_b64_tr_b8 works on groups of 16 lanes; lane t of a group supplies the address of
  // an 8-byte row, result lane i < 8 receives byte i of the rows of lanes 0, 2, .., 14 and lane 8 + i byte i of the rows of
  // lanes 1, 3, .., 15 (scripts/probes/trb8.hip).  With lane t pointing at bytes 8q .. 8q+7 of piece lane (h' = t & 1,
  // pixel 8 hq + (t >> 1)) - q = group & 1, hq = group >> 1 - the wave receives an A fragment of the 16-pixel k-step whose
  // row r = lane & 31 is neuron nu8(r) of the tile and whose elements are the pixels 8 hq + 0..7 in order.
  const uint32_t aT8 = 16u * (32u * (uint32_t)(lane & 1) + ((8u * (uint32_t)(lane >> 5) + (uint32_t)((lane & 15) >> 1)) ^ (8u * (uint32_t)(lane & 1)))) +
                       8u * (uint32_t)((lane >> 4) & 1);   // (pixel slot ^ 8 h': the lsw8 swizzle of the byte pieces)
  typedef __attribute__((address_space(3))) const u32x4 lds_cv4;
  typedef __attribute__((address_space(3))) u32x4 lds_v4;
  typedef __attribute__((address_space(3))) const uint32_t lds_cu32;
  typedef __attribute__((address_space(3))) uint32_t lds_u32;
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  auto tr_pair = [&](uint32_t b1, uint32_t b2, int imm) -> u32x4 {     // fragment = two transposed 8-byte reads
    const s16x4 r0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(uintptr_t)(b1 + imm));
    const s16x4 r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(uintptr_t)(b2 + imm));
    const u32x2 a0 = __builtin_bit_cast(u32x2, r0), a1 = __builtin_bit_cast(u32x2, r1);
    return u32x4{a0.x, a0.y, a1.x, a1.y};
  };
  auto w_of = [&](int x, int s) -> u32x4 {
    return s < KSR ? wreg[x][s < KSR ? s : 0]
                   : reinterpret_cast<const u32x4*>(sWsp + (x * WSP + (s >= KSR ? s - KSR : 0)) * 1024)[lane];
  };
  const float lim448 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(0x43e00000));   // 448.0f, held in an SGPR
  uint32_t ones_h2 = 0x3c003c00u;   // (1.0h, 1.0h) in a register
  asm volatile("" : "+v"(ones_h2));
  u32x4 ep_d[XT];                   // deltas of one k-step: 8 fp8 (.x .y) or 8 fp16, stored as soon as they are complete
  u32x2 ep_pw2 = {0u, 0u};
  uint32_t ep_sn = 0;
  float ep_x0 = 0.f, ep_x1 = 0.f;
  // slice e of the epilogue of row tile x; bP / bS / bSx: per-step bases of the phase piece element and of the sine
  // piece elements (even / odd k-step) of row tile 0 of this wave
  auto x_epi_sub = [&](int k, int x, int e, const f32x16& g, uint32_t bP, uint32_t bS, uint32_t bSx) {
    // e = 4*q + v: values 2v, 2v+1 of k-step 2*it+q (bytes 8q + 2v, 8q + 2v + 1 of the tile's 16-byte piece element)
    const int q = e >> 2, v = e & 3, it = xit0 + x, ks = 2 * it + q;
    float r0 = 0.f, r1 = 0.f;
    if (P0) {
      const f32x4 ta = sL0[16 * ks + pi_perm(lane >> 5, 2 * v)], tb = sL0[16 * ks + pi_perm(lane >> 5, 2 * v + 1)];
      r0 = __builtin_fmaf(ta.y, ep_x1, __builtin_fmaf(ta.x, ep_x0, ta.z)) * a.sc_first;
      r1 = __builtin_fmaf(tb.y, ep_x1, __builtin_fmaf(tb.x, ep_x0, tb.z)) * a.sc_first;
    } else {
      // 8 phase bytes of k-step q in one 8-byte read per 4 slices (a dword per 2 slices is a 4-way bank conflict at
      // the 16-byte lane stride of a piece; 8 bytes are 2-way)
      if (v == 0) ep_pw2 = *(__attribute__((address_space(3))) const u32x2*)(uintptr_t)(bP + x * 1024 + 8 * q);
      if constexpr (!TAB) {
        const uint32_t pw = (v >> 1) ? ep_pw2.y : ep_pw2.x;
        if (v & 1) { r0 = phase_rev8<2>(pw); r1 = phase_rev8<3>(pw); }
        else { r0 = phase_rev8<0>(pw); r1 = phase_rev8<1>(pw); }
      }
    }
    const int t0 = 8 * q + 2 * v;
    float c0, c1;
    uint32_t sn;
    if constexpr (TAB) {
      typedef __attribute__((ext_vector_type(2))) _Float16 h2;
      const uint32_t pw = (v >> 1) ? ep_pw2.y : ep_pw2.x;
      const int b0 = (v & 1) ? 2 : 0;
      const uint32_t e0 = *(lds_cu32*)(uintptr_t)(aTab + (((pw >> (8 * b0)) & 0xffu) << 2));
      const uint32_t e1 = *(lds_cu32*)(uintptr_t)(aTab + (((pw >> (8 * b0 + 8)) & 0xffu) << 2));
      c0 = (float)__builtin_bit_cast(h2, e0)[1];
      c1 = (float)__builtin_bit_cast(h2, e1)[1];
      sn = __builtin_amdgcn_perm(e1, e0, 0x05040100u);      // (sin 0, sin 1)
    } else {
      c0 = __builtin_amdgcn_cosf(r0); c1 = __builtin_amdgcn_cosf(r1);
      sn = OP::pack2(__builtin_amdgcn_sinf(r0), __builtin_amdgcn_sinf(r1));
    }
    if (LAST) { c0 *= dfac; c1 *= dfac; }
    if (D8) {
      int w = (int)ep_d[x][v >> 1];
      // saturation as ONE v_med3_f32 per value: both bounds are the same SGPR (the negation is a source modifier), which is
      // the one constant-bus operand a VOP3 instruction may read
      const float s0 = __builtin_amdgcn_fmed3f(g[t0] * c0, -lim448, lim448), s1 = __builtin_amdgcn_fmed3f(g[t0 + 1] * c1, -lim448, lim448);
      w = (v & 1) ? __builtin_amdgcn_cvt_pk_fp8_f32(s0, s1, w, true) : __builtin_amdgcn_cvt_pk_fp8_f32(s0, s1, 0, false);
      ep_d[x][v >> 1] = (uint32_t)w;
    } else {
      ep_d[x][v] = OP::pack2(g[t0] * c0, g[t0 + 1] * c1);
    }
    // sines of two slices (4 values) leave in one 8-byte write
    if ((v & 1) == 0) ep_sn = sn;
    else *(__attribute__((address_space(3))) u32x2*)(uintptr_t)((q ? bSx : bS) + x * 2048 + q * 1024 + 4 * (v - 1)) = u32x2{ep_sn, sn};
    if (v == 3) {   // bytes 8q .. 8q+7 of this lane's piece element
#ifndef SF_EXPERIMENT_NO_STORE
      if (D8) reinterpret_cast<u32x2*>(&a.Dout[((pb_begin + k * pb_step) * IT + it) * 64 + lane])[q] = u32x2{ep_d[x].x, ep_d[x].y};
      else store_stream(&a.Dout[((pb_begin + k * pb_step) * KSI + ks) * 64 + lane], ep_d[x]);
#else
      asm volatile("" ::"v"(ep_d[x]));
#endif
    }
  };
  auto w_mma_chunk = [&](int x, const u32x4& fa, const u32x4* fb) {
#pragma unroll
    for (int y = 0; y < WI; ++y) acc[x][y] = OP::mfma(fa, fb[y], acc[x][y]);
    // bias gradient: row sums of delta^T.  One v_dot2_f32_f16 against (1, 1) per dword, computed by EVERY wave (only
    // the wc == 0 waves write theirs out): a wave-uniform branch here splits the step into basic blocks and costs
    // the register allocator more than the four instructions do
    // (element-wise: hipcc 7.2 folds a `for e: fdot2(bit_cast(fa[e]), ..)` loop into four uses of fa[0])
    typedef __attribute__((ext_vector_type(2))) _Float16 h2;
    const h2 one2 = __builtin_bit_cast(h2, ones_h2);
    const uint32_t f0 = fa.x, f1 = fa.y, f2 = fa.z, f3 = fa.w;
    if (!DBSPLIT || (x % WAVES_C) == wc) {   // DBSPLIT: the WAVES_C waves that hold the same delta^T fragments share the row tiles
      float d = dbs[x];
      d = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, f0), one2, d, false);
      d = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, f1), one2, d, false);
      d = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, f2), one2, d, false);
      d = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, f3), one2, d, false);
      dbs[x] = d;
    }
  };
  // ---- one pipeline step: C(kx+1), X(kx), W(kx-1) in one hand-cut instruction stream -------------------------
  // X: row tile outer; the delta piece of k-step c (lane-linear ds_read_b128 from D16) is the B operand and is read PF
  // chunks ahead into a small register ring: with one wave per SIMD (4-wave form) nothing else covers the LDS
  // latency; the 8-wave form has neither the registers nor the need (its SIMD partner fills the gap): PF = 1.
  // W: 2*WJ chunks of WI MFMAs; the VALU epilogue of the X tiles (XT*ESUB slices of 2 values) rides along.
  typedef typename RawFrag<IN8>::type raw_t;
#ifdef SF_EXPERIMENT_STAMP
  unsigned long long st_bar = 0, st_x = 0, st_w = 0, st_n = 0;
#endif
  auto step = [&](int kx, bool do_c, bool do_x, bool do_w) {
#ifdef SF_EXPERIMENT_STAMP
    const unsigned long long t_s0 = __builtin_amdgcn_s_memtime();
#endif
    u32x4 xb[NXB];
    u32x4 fb[FB2 ? 2 : 1][WI];
    raw_t fa[2];
    f32x16 g[XT];
    // wave-uniform buffer offsets of this step (SGPRs): ring slots of the deltas of block kx (X) and of block kx-1 (W, this
    // wave's row tiles)
    const uint32_t uX = IN8 ? smem0 + oX16 + (uint32_t)(kx & 1) * (KSJ * 1024) : smem0 + (uint32_t)(kx % NB) * (DPC * 1024);
    const uint32_t uWd = smem0 + (uint32_t)((kx + NB - 1) % NB) * (DPC * 1024) + (uint32_t)(wr * WJ) * (uint32_t)TSTR;
    const uint32_t uWs = smem0 + oS16 + (uint32_t)((kx + 1) & 1) * (KSI * 1024) + (uint32_t)(wc * WI) * 2048u;  // S16 of block kx-1, this wave's column tiles
    const uint32_t uEs = smem0 + oS16 + (uint32_t)(kx & 1) * (KSI * 1024) + (uint32_t)xit0 * 2048u;             // S16 of block kx, this wave's X tiles
    const uint32_t uEp = smem0 + oRP + (uint32_t)(kx % NBP) * (PPC * 1024) + (uint32_t)xit0 * 1024u;             // phase pieces of block kx
    // C(kx+1) rides in the shadow of phase X (sixteen dependent MFMAs with little else to issue): raw bytes read behind
    // chunk 1, the two converted pieces written behind chunks 6 and 10.  (At the top of the step, with both waves of every
    // SIMD in it at once, it was a serial LDS round trip.)
    constexpr bool CVX = IN8 && XT * KSX >= 12 && DPC == NW;
    if (do_c && !(CVX && do_x)) convert(kx + 1);
    u32x4 cv_raw;
    const uint32_t cv_src = smem0 + (uint32_t)((kx + 1) % NB) * (DPC * 1024) + (uint32_t)wave * 1024u + aLN;
    const uint32_t cv_dst = smem0 + oX16 + (uint32_t)((kx + 1) & 1) * (KSJ * 1024) + (uint32_t)wave * 2048u + (uint32_t)lsw8 * 16u;
    if (P0 && do_x) pixel_xy(kx, ep_x0, ep_x1);
    const uint32_t bX0 = (IN8 ? aLN : aL1) + uX, bX1 = (IN8 ? aLN : aL1x) + uX;   // (X16 is stored lane-linearly: no swizzle)
    auto x_load = [&](int c) -> u32x4 { return *(lds_cv4*)(uintptr_t)(((c & 1) ? bX1 : bX0) + c * 1024); };
    const uint32_t bWa1 = (IN8 ? aT8 : aT1) + uWd, bWa2 = aT2 + uWd, bWb1 = aT1 + uWs, bWb2 = aT2 + uWs;
    // A fragment (delta^T) of row tile x, pixel k-step kk (phase W)
    auto wa_load = [&](int kk, int x) -> raw_t {
      if constexpr (IN8) {
        typedef __attribute__((ext_vector_type(2))) int i32x2;
        const i32x2 r = __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) i32x2*)(uintptr_t)(bWa1 + x * TSTR + kk * 256));
        return u32x2{(uint32_t)r.x, (uint32_t)r.y};
      } else {
        return tr_pair(bWa1, bWa2, x * TSTR + kk * 256);
      }
    };
    auto wb_load = [&](int kk, u32x4* dst) {
#pragma unroll
      for (int y = 0; y < WI; ++y) dst[y] = tr_pair(bWb1, bWb2, y * 2048 + kk * 256);
    };
    const uint32_t bP = aLN + uEp, bS = aL1 + uEs, bSx = aL1x + uEs;
    if (do_x) {
#pragma unroll
      for (int c = 0; c < PF && c < XT * KSX; ++c) xb[c % NXB] = x_load(c % KSX);
    }
    if (do_w && (!do_x || XT * KSX <= PF)) {
      wb_load(0, fb[0]); fa[0] = wa_load(0, 0);
      if (FB2) wb_load(1, fb[FB2 ? 1 : 0]);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (do_x) {
      // linear chunk index i = x * KSX + c (row tile outer: one X accumulator tile is being built at a time);
      // the B piece of chunk i + PF is requested before the MFMA of chunk i
#pragma unroll
      for (int x = 0; x < XT; ++x) {
        g[x] = f32x16{};
#pragma unroll
        for (int c = 0; c < KSX; ++c) {
          const int i = x * KSX + c;
          if (i + PF < XT * KSX) xb[(i + PF) % NXB] = x_load((i + PF) % KSX);
          else if (do_w && i + PF == XT * KSX) {     // first W operands ride behind the last B piece
            wb_load(0, fb[0]); fa[0] = wa_load(0, 0);
            if (FB2) wb_load(1, fb[FB2 ? 1 : 0]);
          }
          g[x] = OP::mfma(w_of(x, c), xb[i % NXB], g[x]);
          if (CVX && do_c) {
            if (i == 1) cv_raw = *(lds_cv4*)(uintptr_t)cv_src;
            if (i == 6) *(lds_v4*)(uintptr_t)cv_dst = fp8x8_to_f16(cv_raw.x, cv_raw.y);
            if (i == 10) *(lds_v4*)(uintptr_t)(cv_dst + 1024u) = fp8x8_to_f16(cv_raw.z, cv_raw.w);
          }
          if (PF == 0 && do_w && i + 1 == XT * KSX) {   // no read-ahead: the first W operands follow the last X MFMA
            wb_load(0, fb[0]); fa[0] = wa_load(0, 0);
            if (FB2) wb_load(1, fb[FB2 ? 1 : 0]);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
#ifdef SF_EXPERIMENT_STAMP
    const unsigned long long t_s1 = __builtin_amdgcn_s_memtime();
#endif
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
      for (int x = 0; x < WJ; ++x) {
        const int i = kk * WJ + x;
        if (do_w) {
          if (FA2) {
            if (x + 1 < WJ) fa[(i + 1) & 1] = wa_load(kk, x + 1);
            else if (kk == 0) fa[(i + 1) & 1] = wa_load(1, 0);
          }
          w_mma_chunk(x, frag_of(fa[FA2 ? (i & 1) : 0]), fb[FB2 ? kk : 0]);
          if (!FA2) {   // single fragment buffer: the next one is requested once the MFMAs of this chunk have issued
            if (x + 1 < WJ) fa[0] = wa_load(kk, x + 1);
            else if (kk == 0) fa[0] = wa_load(1, 0);
          }
          // 8-wave form: the fragments of the second pixel k-step replace the first once their last MFMA has issued
          if (!FB2 && kk == 0 && x + 1 == WJ) wb_load(1, fb[0]);
        }
        if (do_x) {
#pragma unroll
          for (int eg = i * NE / NWC; eg < (i + 1) * NE / NWC; ++eg) x_epi_sub(kx, eg / ESUB, eg % ESUB, g[eg / ESUB], bP, bS, bSx);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#ifdef SF_EXPERIMENT_STAMP
    const unsigned long long t_s2 = __builtin_amdgcn_s_memtime();
    if (do_x && do_w) { st_x += t_s1 - t_s0; st_w += t_s2 - t_s1; st_n += 1; }
#endif
  };

  if (nblk > 0) {
    // Ring protocol.  Step k requests D(k + AD) and P(k + AP); the barrier at the top of step k (B_k) needs D(k + ND)
    // and P(k) landed:
    //   D  : AD = NB-2 (the slot of block k-2 was last read by W(k-2), step k-1), ND = 0 - IN8: ND = 1 (C(k+1) runs in step k)
    //   P  : AP = NBP-1 (the slot of block k-1 was decoded in step k-1)
    // so the request B_k waits for was issued AD-ND (resp. AP) steps earlier, and everything issued in the steps after
    // that one is younger in this wave's in-order vmcnt queue: YS = min(AD-ND, AP) - 1 full steps of GD + GP LDS-DMA
    // instructions and S_ST delta stores (full = those steps still had blocks to request).  Outside that steady state
    // the barrier drains the queue.  The first and the last block are peeled: the steady loop body has no conditional
    // phases.
    constexpr int AD = NB - 2, ND = IN8 ? 1 : 0, AP = NBP - 1;
    constexpr int YS = (P0 ? AD - ND : (AD - ND < AP ? AD - ND : AP)) - 1;
    constexpr int AX = P0 ? AD : (AD > AP ? AD : AP);       // a step is full while k + AX < nblk
    static_assert(YS >= 1, "ring depth");
    for (int k = 0; k < AD && k < nblk; ++k) stageD(k);
    for (int k = 0; k < AP && k < nblk; ++k) stageP(k);
    bar_all();                              // prologue only: everything requested so far has landed
    if (IN8) {
      convert(0);
      bar_all();
    }
    if (AD < nblk) stageD(AD);
    if (AP < nblk) stageP(AP);
    asm volatile("" ::: "memory");
    if (nblk > 1) step(0, true, true, false); else step(0, false, true, false);
    for (int k = 1; k + 1 < nblk; ++k) {
#ifdef SF_EXPERIMENT_STAMP
      const unsigned long long t_b0 = __builtin_amdgcn_s_memtime();
#endif
      if (k >= YS && k - 1 + AX < nblk) bar_dma<YS * (GD + GP + S_ST)>(); else bar_all();
#ifdef SF_EXPERIMENT_STAMP
      st_bar += __builtin_amdgcn_s_memtime() - t_b0;
#endif
      if (k + AD < nblk) stageD(k + AD);
      if (k + AP < nblk) stageP(k + AP);
      asm volatile("" ::: "memory");
      step(k, true, true, true);
    }
    if (nblk > 1) {
      bar_all();
      step(nblk - 1, false, true, true);
    }
    bar_lds();
    step(nblk, false, false, true);
  }
#if defined(SF_EXPERIMENT_STAMP) && !defined(SF_EXPERIMENT_STAMP2)
  if (a.dbg && !LAST && lane == 0 && (wave == 0 || wave == 5) && (blockIdx.x == 3 || blockIdx.x == 200) && st_n) {
    float* o = a.dbg + 32 + ((blockIdx.x == 3 ? 0 : 2) + (wave == 0 ? 0 : 1)) * 4 + (P0 ? 16 : 0);
    o[0] = (float)st_bar / (float)st_n; o[1] = (float)st_x / (float)st_n; o[2] = (float)st_w / (float)st_n; o[3] = (float)st_n;
  }
#endif
  float* slab = a.slab + (size_t)blockIdx.x * (JW * IW + JW);
  const int cl = lane & 31, hh = lane >> 5;
#pragma unroll
  for (int x = 0; x < WJ; ++x)
#pragma unroll
    for (int y = 0; y < WI; ++y)
#pragma unroll
      for (int t = 0; t < 16; ++t)
        slab[(size_t)(32 * (wr * WJ + x) + (IN8 ? nu8(rho(t, hh)) : rho(t, hh))) * IW + 32 * (wc * WI + y) + cl] = acc[x][y][t];
#pragma unroll
  for (int x = 0; x < WJ; ++x) {
    if (DBSPLIT ? (x % WAVES_C) == wc : wc == 0) {
      const float tsum = dbs[x] + __shfl_xor(dbs[x], 32);
      if (hh == 0) slab[JW * IW + 32 * (wr * WJ + x) + (IN8 ? nu8(cl) : cl)] = tsum;
    }
  }
}

// LDS bytes of k_bwd8 (host side of the same layout)
template <int JW, int IW, int NW, bool LAST, bool P0, int NB, int PARK = 0, int NBP_ = 0, bool D8 = true>
constexpr size_t bwd8_lds_bytes() {
  constexpr int JT = JW / 32, IT = IW / 32, KSJ = JW / 16, KSI = IW / 16, KSX = LAST ? 1 : KSJ, XT = IT / NW;
  constexpr int DPC = (D8 && !LAST) ? JT : (LAST ? 2 : KSJ), PPC = P0 ? 0 : IT;
  constexpr int WSP = PARK;
  constexpr int NBP = NBP_ > 0 ? NBP_ : NB;
  return (size_t)(NB * DPC + NBP * PPC + ((D8 && !LAST) ? 2 * KSJ : 0) + 2 * KSI + NW * XT * WSP) * 1024 + (P0 ? (size_t)IW * 16 : 1024);   // P0: layer-0 table; else the sin/cos table of the phase bytes
}

// ---------------------------------------------------------------------------------------------
// k_dw0_8: weight gradient of layer 0 from fp8 deltas (k_dw0 of siren_kernels.hip with the byte pieces expanded
// to the 16-bit image one block ahead).  Ring: 8 slots of JW/32 KiB; D16: 2 x JW/16 KiB.
// ---------------------------------------------------------------------------------------------
template <int JW, typename OP, int NB = 8>
__global__ __launch_bounds__(JW * 2) void k_dw0_8(Dw0Args a) {
  constexpr int NW = JW / 32, JT = JW / 32, KSJ = JW / 16;
  constexpr int GD = JT / NW;   // = 1
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const rD = smem;
  char* const d16 = rD + NB * JT * 1024;
  uint16_t* sXY = reinterpret_cast<uint16_t*>(d16 + 2 * KSJ * 1024);          // [2][4][32]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long pb_begin = blockIdx.x, pb_step = gridDim.x;
  const int nblk = (int)((a.n_pb - pb_begin + pb_step - 1) / pb_step);
  f32x16 acc = {};
  float dbs = 0.f;
  const int trb = tr_lane_base(lane);
  const int lsw = sw_lane(lane, 0);
  auto stage = [&](int k) {
    const long pb = pb_begin + k * pb_step;
    for (int pc = wave; pc < JT; pc += NW)
      glds16s(a.D + (pb * (a.ks_total / 2) + a.ks_off / 2 + pc) * 64, (uint32_t)lane * 16u, rD + ((k % NB) * JT + pc) * 1024);   // (ks_total / 2 tiles per pixel block; wider layers: a 256-neuron slice at tile ks_off / 2)
  };
  auto convert = [&](int k) {
    const char* src = rD + (k % NB) * JT * 1024;
    char* dst = d16 + (k & 1) * KSJ * 1024;
    for (int pc = wave; pc < JT; pc += NW) {
      const u32x4 raw = reinterpret_cast<const u32x4*>(src + pc * 1024)[lane];
      reinterpret_cast<u32x4*>(dst + (2 * pc) * 1024)[lsw] = fp8x8_to_f16(raw.x, raw.y);
      reinterpret_cast<u32x4*>(dst + (2 * pc + 1) * 1024)[lsw ^ 8] = fp8x8_to_f16(raw.z, raw.w);
    }
  };
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
  auto coord_frag = [&](int k, int kk) -> u32x4 {
    const int c = lane & 31;
    u32x4 o = {0u, 0u, 0u, 0u};
    if (c < 4) o = *reinterpret_cast<const u32x4*>(sXY + (k & 1) * 128 + c * 32 + 16 * kk + 8 * (lane >> 5));
    return o;
  };
  if (nblk > 0) {
    for (int k = 0; k < NB && k < nblk; ++k) stage(k);
    build_xy(0);
    bar_all();
    convert(0);
    for (int k = 0; k < nblk; ++k) {
      // B_k: block k+1 landed (requested in step k+1-NB); younger: the NB-2 requests of steps k+2-NB .. k-1
      if (k >= NB - 2 && k - 1 + NB < nblk) bar_dma<(NB - 2) * GD>(); else bar_all();
      if (k + NB < nblk) stage(k + NB);
      asm volatile("" ::: "memory");
      if (k + 1 < nblk) { convert(k + 1); build_xy(k + 1); }
      const u32x4 fb0 = coord_frag(k, 0), fb1 = coord_frag(k, 1);
      const char* sD = d16 + (k & 1) * KSJ * 1024;
      const u32x4 fa0 = ds_read_tr_frag(sD, trb, wave, 0);
      const u32x4 fa1 = ds_read_tr_frag(sD, trb, wave, 1);
      acc = OP::mfma(fa0, fb0, acc);
      acc = OP::mfma(fa1, fb1, acc);
      float t = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) t += OP::sum2(fa0[e]) + OP::sum2(fa1[e]);
      dbs += t;
    }
  }
  float* slab = a.slab + (size_t)blockIdx.x * (JW * 32 + JW);
  const int cl = lane & 31, hh = lane >> 5;
#pragma unroll
  for (int t = 0; t < 16; ++t) slab[(size_t)(32 * wave + rho(t, hh)) * 32 + cl] = acc[t];
  const float tsum = dbs + __shfl_xor(dbs, 32);
  if (hh == 0) slab[JW * 32 + 32 * wave + cl] = tsum;
}

}  // namespace sf
