// siren_wide.hip — layer-at-a-time kernels for hidden widths above 256 (512, 1024: BASELINE configs 3 and 5).
//
// The fused kernels of siren_kernels.hip keep a whole activation vector per pixel in registers and a
// whole 256x256 weight gradient per workgroup; neither fits beyond width 256.  Wide networks therefore run
// layer by layer with the activations (phases) and deltas in HBM, still in F-layout, through three kernels:
//   k_wlayer0   phases and activations of layer 0 from the coordinates (elementwise)
//   k_wgemm     Out^T[256-neuron block x 256 pixels per workgroup] = A-block * B, A (weights) streamed through
//               an LDS ring by LDS-DMA, B (input pieces) read from HBM/L2 into registers two chunks ahead.
//               MODE 0: forward hidden layer (B = activation, epilogue -> phase AND activation),
//               MODE 1: forward last layer + residual/loss, MODE 2: backward data (B = delta, epilogue x cos(phase))
//   k_wdw       weight gradient blocks: delta^T * activation, both operands read transposed from an LDS ring
// Unlike the fused width-256 path, each hidden layer keeps BOTH its phase (unorm16, for cos in the backward)
// and its activation sin(phase) (16-bit float, the GEMM operand) in HBM: every value is an operand of WD/256
// workgroups per product, so decoding phases inside the GEMM loops would repeat the transcendental WD/256 times
// and put as many VALU cycles as MFMA cycles into the inner loop.
// Arithmetic intensity grows with the width (FLOPs ~ W^2, bytes ~ W), so this path is less HBM-starved than
// the width-256 one even without fusion.  Reference arithmetic: same lines as siren_kernels.hip.
#pragma once

namespace sf {

#ifdef SF_WEXP_NOSTORE   // timing-only: the GEMM epilogues compute their values and drop them
#define WST(dst, val) do { auto v_ = (val); asm volatile("" ::"v"(v_)); } while (0)
#elif defined(SF_WEXP_NTSTORE)   // A/B: non-temporal epilogue stores
#define WST(dst, val) __builtin_nontemporal_store((val), &(dst))
#else
#define WST(dst, val) (dst) = (val)
#endif

// ---------------------------------------------------------------------------------------------------------
// blocked A image: [ob][chunk c][tile ot][s4][lane][8] ; chunk = 4 k-steps ; OT tiles of 32 rows per block
// element (ob, c, ot, s4, lane=(r,h), j) = M[(ob*OT + ot)*32 + r][16*(4c + s4) + PI(h, j)] * scale   (0 outside)
// with M = W (forward) or W^T (backward).
// ---------------------------------------------------------------------------------------------------------
struct WImgArgs {
  const float* W;       // [rows][cols] row-major fp32 master weights of one layer
  int rows, cols;       // nn.Linear layout: rows = out features, cols = in features
  int transpose;        // 0: M = W (M rows = out), 1: M = W^T (M rows = in)
  int OT;               // tiles per output block (8, or 1 for the 32-row padded last layer forward)
  int n_ob, n_chunk;    // output blocks, K chunks
  float scale;
  int f16;
  uint16_t* dst;
  const float* link;    // fp8 deltas: *link (a power of two, k_fp8_links) multiplies the scale; nullptr: 1
};
__global__ void k_wimage(WImgArgs a) {
  const long total = (long)a.n_ob * a.n_chunk * a.OT * 4 * 512;
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= total) return;
  long e = gid;
  const int j = e & 7; e >>= 3;
  const int lane = e & 63; e >>= 6;
  const int s4 = e & 3; e >>= 2;
  const int ot = (int)(e % a.OT); e /= a.OT;
  const int c = (int)(e % a.n_chunk);
  const int ob = (int)(e / a.n_chunk);
  const int r = lane & 31, h = lane >> 5;
  const int mrow = (ob * a.OT + ot) * 32 + r, mk = 16 * (4 * c + s4) + pi_perm(h, j);
  const int mrows = a.transpose ? a.cols : a.rows, mcols = a.transpose ? a.rows : a.cols;
  float v = 0.f;
  const float sc = a.scale * (a.link ? a.link[0] : 1.0f);      // (one factor: a single rounding into the 16-bit image)
  if (mrow < mrows && mk < mcols) v = (a.transpose ? a.W[(long)mk * a.cols + mrow] : a.W[(long)mrow * a.cols + mk]) * sc;
  a.dst[gid] = a.f16 ? to_f16(v) : to_bf16(v);
}

// layer-0 table {w00, w01, b0, 0} per neuron and the pre-scaled fp32 biases of the hidden layers 1..D-2
// ([D-2][WD]) followed by the last layer's padded to 32
struct WTabArgs {
  const float* params; int depth, WD, out_features; long off_w0, off_b0; long off_b[16]; float wscale, hscale;
  f32x4* l0tab; float* bias;
};
__global__ void k_wtables(WTabArgs a) {
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid < a.WD) {
    const float* W0 = a.params + a.off_w0;
    a.l0tab[gid] = f32x4{W0[gid * 2], W0[gid * 2 + 1], a.params[a.off_b0 + gid], 0.f};
  }
  const long nh = (long)(a.depth - 2) * a.WD;
  if (gid < nh) {
    const int l = (int)(gid / a.WD) + 1, n = (int)(gid % a.WD);
    a.bias[gid] = a.params[a.off_b[l] + n] * a.hscale;
  }
  if (gid < 32) a.bias[nh + gid] = gid < a.out_features ? a.params[a.off_b[a.depth - 1] + gid] * a.wscale : 0.f;
}

// ---------------------------------------------------------------------------------------------------------
// k_wlayer0: P_0 piece (pb, s) for every pixel block: phase of sin(omega0 * (W0 x + b0))
// ---------------------------------------------------------------------------------------------------------
struct WL0Args {
  const float* gh; const float* gw; int W, row_begin; long pix0, npix;
  const f32x4* l0tab; float sc_first; int KS; long n_pieces; u32x4* P; u32x4* Act;
};
// P8 (scratch_format 12 on the wide path, round 3): the phases are spilled as BYTES, one 1 KiB piece per (pixel block, 32-neuron
// tile) - byte 8 q + j of lane (h, m) = neuron 32 tile + 16 q + PI(h, j), the layout of the width-256 path - so a wave takes
// TWO k-steps (one tile): two activation pieces and one phase piece.  a.KS stays the number of 16-neuron k-steps.
template <typename OP2, bool P8 = false>
__global__ __launch_bounds__(256) void k_wlayer0(WL0Args a) {
  constexpr int SPW = P8 ? 2 : 1;                      // k-steps per wave
  const long unit = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (unit * SPW >= a.n_pieces) return;
  const int lane = threadIdx.x & 63, m = lane & 31, h = lane >> 5;
  const int upb = a.KS / SPW;                          // units per pixel block
  const long pb = unit / upb;
  const int s0 = (int)(unit % upb) * SPW;
  long pix = a.pix0 + pb * 32 + m;
  if (pix >= a.npix) pix = a.npix - 1;
  const int row = (int)(pix / a.W), col = (int)(pix - (long)row * a.W);
  const float x0 = (a.gh[a.row_begin + row] - 0.5f) * 2.0f, x1 = (a.gw[col] - 0.5f) * 2.0f;
  uint32_t pb8[4] = {0u, 0u, 0u, 0u};
#pragma unroll
  for (int q = 0; q < SPW; ++q) {
    const int s = s0 + q;
    float tv[8], ph[8], av[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const f32x4 t = a.l0tab[16 * s + pi_perm(h, j)];
      const float tt = __builtin_fmaf(t.y, x1, __builtin_fmaf(t.x, x0, t.z)) * a.sc_first;
      tv[j] = tt;
      ph[j] = __builtin_amdgcn_fractf(tt);
      av[j] = __builtin_amdgcn_sinf(tt);
    }
    const long piece = pb * a.KS + s;
    if constexpr (P8) {
      pb8[2 * q] = phase_byte4(tv, av);
      pb8[2 * q + 1] = phase_byte4(tv + 4, av + 4);
    } else {
      a.P[piece * 64 + lane] = u32x4{pack_phase2(ph[0], ph[1]), pack_phase2(ph[2], ph[3]), pack_phase2(ph[4], ph[5]),
                                     pack_phase2(ph[6], ph[7])};
    }
    a.Act[piece * 64 + lane] =
        u32x4{OP2::pack2(av[0], av[1]), OP2::pack2(av[2], av[3]), OP2::pack2(av[4], av[5]), OP2::pack2(av[6], av[7])};
  }
  if constexpr (P8) a.P[(pb * upb + s0 / 2) * 64 + lane] = u32x4{pb8[0], pb8[1], pb8[2], pb8[3]};
}

// ---------------------------------------------------------------------------------------------------------
// k_wgemm
// ---------------------------------------------------------------------------------------------------------
struct WGemmArgs {
  const u32x4* A;        // blocked image of this layer: block ob at A + ob * a_block_pieces * 64
  long a_block_pieces;   // pieces per output block = n_chunk * OT * 4
  int n_chunk;           // K chunks of 4 k-steps
  const u32x4* Bin;      // input tensor (F-layout, ks_in k-steps per pixel block): activations (MODE 0/1) or deltas (MODE 2)
  int ks_in;
  const float* bias;     // MODE 0/1: [out rows], pre-scaled; nullptr otherwise
  float sc;              // MODE 1: 1/wscale ; unused otherwise (hidden images carry omega/(2 pi))
  u32x4* Out;            // MODE 0: phases out, MODE 2: deltas out (F-layout, ks_out k-steps per block)
  u32x4* OutAct;         // MODE 0: activations out (same geometry as Out)
  int ks_out;
  int kp_out;            // P8: phase-byte pieces per pixel block (= width / 32) of Out (MODE 0) / Pprev (MODE 2)
  const u32x4* Pprev;    // MODE 2: phases of the layer whose delta is produced (same geometry as Out)
  u32x4* dump;           // k_wgemm3: 8 KiB the first tile's (empty) epilogue is stored to
  const float* fscale;   // MODE 2, fp8 out: *fscale multiplies the outgoing deltas (the chunk factor, last layer's launch); nullptr: 1
  // MODE 1 (last layer)
  const float* img; float* pred; float gscale; float* sse_part; u32x4* Dlast; long pix0, npix;
  int nout;              // out_features (1..3): channel count and stride of img / pred
  float last_om, last_om_rev;   // sine output layer (outermost_linear=False), 0 otherwise
  int n_super, n_ob;     // 256-pixel super-blocks of the chunk, output blocks; grid = roundup(n_super, 8) * n_ob
};

// k_wgemm<1>: the last layer (32 padded output rows) + residual.  Workgroup = 8 waves = 256 pixels, one
// 32-pixel block per wave; the [32 x WD] weight image streams through a 4-slot LDS ring (4 k-steps per chunk),
// the activations are prefetched two chunks ahead into registers.  (Hidden layers: k_wgemm2 below.)
template <int MODE, typename OP>
__global__ __launch_bounds__(512) void k_wgemm(WGemmArgs a) {
  static_assert(MODE == 1, "hidden layers and the data gradient run k_wgemm2");
  constexpr int OT = MODE == 1 ? 1 : 8;       // 32-row tiles per LDS chunk
  constexpr int TW = MODE == 1 ? 1 : 4;       // tiles per wave
  constexpr int PBW = MODE == 1 ? 1 : 2;      // pixel blocks per wave
  constexpr int NB = 4, CH = OT * 4 * 1024;   // ring slots, bytes per chunk
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sRed = reinterpret_cast<float*>(smem + NB * CH);
  const int tid = threadIdx.x, lane = tid & 63, m = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // XCD-aware 1-D grid: consecutive workgroup ids go to consecutive XCDs (8), each with its own L2.  The n_ob
  // workgroups that read the SAME 256 pixels (one per output block) are made consecutive on ONE XCD, so the
  // input pieces come from HBM once and from that XCD's L2 for the siblings.
  const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
  const int ob = q % a.n_ob;
  const int sb = (q / a.n_ob) * 8 + xcd;                                // 256-pixel super-block
  if (sb >= a.n_super) return;                                          // grid padding (whole workgroup leaves)
  const int t0 = MODE == 1 ? 0 : TW * (wave & 1);                       // first tile of this wave inside the block
  const long pb0 = (long)sb * kWavesFwd + (MODE == 1 ? wave : 2 * (wave >> 1));
  const u32x4* Ablk = a.A + (size_t)ob * a.a_block_pieces * 64;
  auto stage = [&](int c) {
    char* base = smem + (c % NB) * CH;
    for (int pc = wave; pc < OT * 4; pc += kWavesFwd) glds16s(Ablk + ((size_t)c * OT * 4 + pc) * 64, (uint32_t)lane * 16u, base + pc * 1024);
  };
  f32x16 acc[TW][PBW];
#pragma unroll
  for (int t = 0; t < TW; ++t) {
    f32x16 init = f32x16{};
    if (MODE != 2) {
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(&a.bias[(ob * OT + t0 + t) * 32 + 8 * q4 + 4 * h]);
        init[4 * q4 + 0] = b.x; init[4 * q4 + 1] = b.y; init[4 * q4 + 2] = b.z; init[4 * q4 + 3] = b.w;
      }
    }
#pragma unroll
    for (int p = 0; p < PBW; ++p) acc[t][p] = init;
  }
  float tgt[3] = {0.f, 0.f, 0.f};
  const long pix = a.pix0 + pb0 * 32 + m;
  const bool valid = MODE == 1 && pix < a.npix;
  if (MODE == 1 && a.img && h == 0 && valid) {
#pragma unroll
    for (int c = 0; c < 3; ++c) if (c < a.nout) tgt[c] = a.img[pix * a.nout + c];
  }
  asm volatile("" ::"v"(tgt[0]), "v"(tgt[1]), "v"(tgt[2]), "v"(acc[0][0][0]));
  // B pieces are prefetched TWO chunks ahead into registers, right behind the LDS-DMA of the same chunk: vmcnt
  // counts both in issue order, so "everything but the youngest G + 4*PBW operations" at the top of a step means
  // chunk c (A in LDS, B in registers) has landed while chunk c+1 stays in flight.
  constexpr int G = OT * 4 / kWavesFwd;      // LDS-DMA instructions per wave per chunk (0: uneven -> full waits)
  auto load_b = [&](int c, u32x4 (&raw)[PBW][4]) {
#pragma unroll
    for (int p = 0; p < PBW; ++p)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int sidx = 4 * c + i;
        sidx = sidx < a.ks_in ? sidx : a.ks_in - 1;
        raw[p][i] = a.Bin[((pb0 + p) * a.ks_in + sidx) * 64 + lane];
      }
  };
  u32x4 r0[PBW][4], r1[PBW][4], r2[PBW][4];
  stage(0);
  load_b(0, r0);
  if (a.n_chunk > 1) { stage(1); load_b(1, r1); }
  for (int c = 0; c < a.n_chunk; ++c) {
    if (G > 0 && c + 1 < a.n_chunk) bar_dma<G + 4 * PBW>(); else bar_all();
    if (c + 2 < a.n_chunk) { stage(c + 2); load_b(c + 2, r2); }
    asm volatile("" ::: "memory");
    const u32x4* sA = reinterpret_cast<const u32x4*>(smem + (c % NB) * CH) + lane;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      u32x4 b[PBW];
#pragma unroll
      for (int p = 0; p < PBW; ++p) b[p] = 4 * c + i < a.ks_in ? r0[p][i] : u32x4{0u, 0u, 0u, 0u};
#pragma unroll
      for (int t = 0; t < TW; ++t) {
        const u32x4 fa = sA[((t0 + t) * 4 + i) * 64];
#pragma unroll
        for (int p = 0; p < PBW; ++p) acc[t][p] = OP::mfma(fa, b[p], acc[t][p]);
      }
    }
#pragma unroll
    for (int p = 0; p < PBW; ++p)
#pragma unroll
      for (int i = 0; i < 4; ++i) { r0[p][i] = r1[p][i]; r1[p][i] = r2[p][i]; }
  }
  // ---- epilogue: residual, SSE partial, dL/dout ----
  {
    float sse = 0.f, d[3] = {0.f, 0.f, 0.f};
    if (h == 0 && valid) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        if (c >= a.nout) break;
        float o = acc[0][0][c] * a.sc, dfac = 1.0f;
        if (a.last_om_rev != 0.f) {
          const float tt = o * a.last_om_rev;
          o = __builtin_amdgcn_sinf(tt);
          dfac = a.last_om * __builtin_amdgcn_cosf(tt);
        }
        const float pr = o * 0.5f + 0.5f;
        if (a.pred) a.pred[pix * a.nout + c] = pr;
        if (a.img) {
          const float r = pr - tgt[c];
          sse += r * r;
          d[c] = r * a.gscale * dfac;
        }
      }
    }
    if (a.Dlast) {
      a.Dlast[(pb0 * 2 + 0) * 64 + lane] = u32x4{OP::pack2(d[0], d[1]), OP::pack2(d[2], 0.f), 0u, 0u};
      a.Dlast[(pb0 * 2 + 1) * 64 + lane] = u32x4{0u, 0u, 0u, 0u};
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sse += __shfl_xor(sse, o);
    if (lane == 0) sRed[wave] = sse;
    __syncthreads();
    if (tid == 0 && a.sse_part) {
      float t = 0.f;
      for (int w = 0; w < kWavesFwd; ++w) t += sRed[w];
      a.sse_part[sb] = t;
    }
  }
}

// k_wgemm2 (MODE 0 / 2): same product and epilogues as k_wgemm, but BOTH operands go through the LDS ring:
// a 256x256 tile needs 128 flop per operand byte, so at the MFMA peak the CUs would pull ~20 TB/s out of L2;
// loading the 256-pixel input chunk once per workgroup (instead of once per wave pair, in registers) cuts the
// L2 -> CU bytes by a third.  Chunk = 2 k-steps: 16 A pieces + 16 B pieces = 32 KiB per slot, 4 slots, staged
// three chunks ahead (4 LDS-DMA instructions per wave per chunk, counted vmcnt).
// P8: the phases are bytes (scratch_format 12, see k_wlayer0): one piece per (pixel block, 32-neuron tile) instead of two.
// NWV = 4 (round 3): the same tile plan with FOUR waves per workgroup - 256 neurons x 128 pixels, 24 KiB per chunk, a ring
// of three - so that TWO workgroups share a CU (2 x 72 KiB of LDS, 2 waves per SIMD as before): they drift apart, and the
// epilogue of one (128 sines per thread, matrix pipe idle) runs beside the main loop of the other.  a.n_super then counts
// 128-pixel units.
// IN8 / OUT8 (MODE 2, scratch_format 8 on the wide path, round 3): the incoming / outgoing deltas are fp8 e4m3 byte pieces, one
// per (pixel block, 32-neuron tile) in the phase-byte layout: a B chunk of two k-steps is ONE piece per pixel block (converted
// in registers, exact), a tile's sixteen outgoing values leave in one store.  Scales: csrc/siren_kernels.hip k_fp8_links.
template <int MODE, typename OP, bool P8 = false, int NWV = 8, bool IN8 = false, bool OUT8 = false>
__global__ __launch_bounds__(NWV * 64, NWV == 4 ? 2 : 1) void k_wgemm2(WGemmArgs a) {
  static_assert(MODE == 0 || MODE == 2, "last layer: k_wgemm<1>");
  static_assert(NWV == 8 || NWV == 4, "waves per workgroup");
  static_assert(MODE == 2 || (!IN8 && !OUT8), "fp8 deltas: the data-gradient product only");
  constexpr int OT = 8, TW = 4, PBW = 2, NPB = NWV;              // pixel blocks per workgroup: two per wave pair
  constexpr int NB = NWV == 8 ? 4 : 3, PD = NB - 1, SLOT = (16 + 2 * NPB) * 1024;
  constexpr int GA = 16 / NWV, GB = (IN8 ? NPB : 2 * NPB) / NWV, G = GA + GB;   // LDS-DMA instructions per wave and chunk
  constexpr int NEP = MODE == 0 ? TW * PBW * (P8 ? 3 : 4) : TW * PBW * (OUT8 ? 1 : 2);   // epilogue stores per wave
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // PERSISTENT workgroups: gridDim.x is a multiple of 8 * n_ob, so along t = blockIdx.x + k * gridDim.x the XCD
  // (t & 7) and the output block stay fixed and only the 256-pixel super-block advances.  The weight block
  // and the bias are loop invariants; the first chunks of the NEXT tile are requested before the epilogue of
  // the current one, so neither the DMA latency of a tile's prologue nor its store tail is exposed.
  const int xcd = blockIdx.x & 7, q0 = blockIdx.x >> 3;
  const int ob = q0 % a.n_ob;
  const int sb_step = (gridDim.x >> 3) / a.n_ob * 8;
  int sb = (q0 / a.n_ob) * 8 + xcd;
  if (sb >= a.n_super) return;
  const int t0 = TW * (wave & 1), pw = 2 * (wave >> 1);
  const u32x4* Ablk = a.A + (size_t)ob * a.a_block_pieces * 64;
  const int n2 = a.ks_in / 2;                            // chunks of 2 k-steps
  const bool pipe = n2 >= 4;                             // cross-tile prefetch (the K = 32 last-layer product: plain waits)
  static_assert(PD == 2 || PD == 3, "the wait plan below");
  auto stage = [&](int tsb, int c) {
#ifdef SF_WEXP_NODMA   // timing-only: nothing is staged (the products run on whatever the LDS holds)
    return;
#endif
    char* base = smem + (c % NB) * SLOT;
    const long pbg = (long)tsb * NPB;
#pragma unroll
    for (int i = 0; i < GA; ++i) {
      const int pc = wave + NWV * i, ot = pc >> 1, s2 = pc & 1;
      glds16s(Ablk + ((size_t)((c >> 1) * OT + ot) * 4 + 2 * (c & 1) + s2) * 64, (uint32_t)lane * 16u, base + pc * 1024);
    }
#pragma unroll
    for (int i = 0; i < GB; ++i) {
      const int pc = wave + NWV * i;
      if constexpr (IN8) {      // byte piece c of pixel block pc: k-steps 2 c, 2 c + 1
        glds16s(a.Bin + ((pbg + pc) * (a.ks_in / 2) + c) * 64, (uint32_t)lane * 16u, base + (16 + pc) * 1024);
      } else {
        const int p = pc >> 1, s2 = pc & 1;
        glds16s(a.Bin + ((pbg + p) * a.ks_in + 2 * c + s2) * 64, (uint32_t)lane * 16u, base + (16 + pc) * 1024);
      }
    }
  };
  // the same, one LDS-DMA instruction at a time (i-th of this wave's G): issued BETWEEN MFMAs an instruction costs its wave ~60
  // issue cycles, four in a burst behind a barrier 150-200 each (k_wdw's measurement)
  auto stage_piece = [&](int tsb, int c, int i) __attribute__((always_inline)) {
#ifdef SF_WEXP_NODMA
    return;
#endif
    char* base = smem + (c % NB) * SLOT;
    const long pbg = (long)tsb * NPB;
    if (i < GA) {
      const int pc = wave + NWV * i, ot = pc >> 1, s2 = pc & 1;
      glds16s(Ablk + ((size_t)((c >> 1) * OT + ot) * 4 + 2 * (c & 1) + s2) * 64, (uint32_t)lane * 16u, base + pc * 1024);
    } else {
      const int pc = wave + NWV * (i - GA);
      if constexpr (IN8) {
        glds16s(a.Bin + ((pbg + pc) * (a.ks_in / 2) + c) * 64, (uint32_t)lane * 16u, base + (16 + pc) * 1024);
      } else {
        const int p = pc >> 1, s2 = pc & 1;
        glds16s(a.Bin + ((pbg + p) * a.ks_in + 2 * c + s2) * 64, (uint32_t)lane * 16u, base + (16 + pc) * 1024);
      }
    }
  };
  f32x16 init[TW];
#pragma unroll
  for (int t = 0; t < TW; ++t) {
    init[t] = f32x16{};
    if (MODE == 0) {
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(&a.bias[(ob * OT + t0 + t) * 32 + 8 * q4 + 4 * h]);
        init[t][4 * q4 + 0] = b.x; init[t][4 * q4 + 1] = b.y; init[t][4 * q4 + 2] = b.z; init[t][4 * q4 + 3] = b.w;
      }
    }
  }
  asm volatile("" ::"v"(init[0][0]), "v"(init[TW - 1][15]));   // bias loads retire before the first DMA is issued
#ifdef SF_WEXP_STAGGER   // A/B: workgroups start 1/16 of a tile apart (breaks the convoy in which every CU stores at the same time)
  // (workgroup q = blockIdx.x >> 3 of an XCD goes to CU q mod 32: the second workgroup of a CU is q + 32 - it gets half a tile more)
  for (int i = 0; i < (int)((((blockIdx.x >> 3) & 15) + 8 * ((blockIdx.x >> 8) & 1)) & 15) * SF_WEXP_STAGGER; ++i) __builtin_amdgcn_s_sleep(32);
#endif
  for (int c = 0; c < PD && c < n2; ++c) stage(sb, c);
  bool first = true;
#ifdef SF_WEXP_STAMP   // timing-only build: where a wave's tile goes (waits at the chunk barriers / products / epilogue), in cycles
  unsigned long long st_wait = 0, st_epi = 0, st_tiles = 0;
  const unsigned long long st_begin = __builtin_amdgcn_s_memtime(), st_rt0 = __builtin_amdgcn_s_memrealtime();   // core cycles, 100 MHz ticks
#endif
  while (true) {
    const long pb0 = (long)sb * NPB + pw;
    f32x16 acc[TW][PBW];
#pragma unroll
    for (int t = 0; t < TW; ++t)
#pragma unroll
      for (int p = 0; p < PBW; ++p) acc[t][p] = init[t];
    u32x4 pv[TW][PBW][P8 ? 1 : 2];                         // MODE 2: phases for the cos factor, requested early
    // The chunk loop is SOFTWARE-PIPELINED over its operand fragments (round 3): the six fragments of a k-step are read from LDS
    // while the eight MFMAs of the k-step before run, and the barrier that admits chunk c + 1 sits in the MIDDLE of chunk c
    // (its fragments are requested under chunk c's second k-step).  hipcc's own order read two fragments, waited for them and
    // issued their MFMAs, sixteen chunk barriers per tile in between: stamps showed 27 k cycles of main loop per tile and wave
    // for 8 k cycles of MFMA per SIMD.
    auto admit = [&](int c, bool burst) __attribute__((always_inline)) {   // chunk c landed and its slot's predecessor is free; burst: request chunk c + PD here
      // vmcnt is in-order: "chunk c landed" = all but the younger operations done; younger are the DMA of chunks
      // c+1, c+2 and - in the first PD steps of a tile that follows another - that tile's NEP epilogue stores
#ifdef SF_WEXP_STAMP
      const unsigned long long st_w0 = __builtin_amdgcn_s_memtime();
#endif
      // (in flight behind chunk c: min(PD - 1, n2 - 1 - c) chunks of G instructions each)
      if (!pipe) bar_all();
#ifndef SF_WEXP_LAXWAIT
      else if (c + 1 >= n2) bar_all();
      else if (PD == 3 && c + 2 >= n2) bar_dma<G>();
#endif
#ifdef SF_WEXP_LAXWAIT   // timing-only (RACY): no wait ever requires the previous tile's stores to have retired
      else bar_dma<(PD - 1) * G + NEP>();
#else
      else if (first || c >= PD) bar_dma<(PD - 1) * G>();
      else bar_dma<(PD - 1) * G + NEP>();
#endif
#ifdef SF_WEXP_STAMP
      st_wait += __builtin_amdgcn_s_memtime() - st_w0;
#endif
      if (burst && c + PD < n2) stage(sb, c + PD);
      asm volatile("" ::: "memory");
      if (MODE == 2 && c == n2 - 1) {
#pragma unroll
        for (int t = 0; t < TW; ++t)
#pragma unroll
          for (int p = 0; p < PBW; ++p)
            if constexpr (P8) {
              pv[t][p][0] = a.Pprev[((pb0 + p) * a.kp_out + 8 * ob + (t0 + t)) * 64 + lane];
            } else {
#pragma unroll
              for (int qq = 0; qq < 2; ++qq)
                pv[t][p][qq] = a.Pprev[((pb0 + p) * a.ks_out + 16 * ob + 2 * (t0 + t) + qq) * 64 + lane];
            }
      }
    };
    struct Frags { u32x4 fa[TW]; u32x4 b[PBW]; };
    u32x4 braw[PBW];                                       // IN8: the byte pieces of the current chunk (both k-steps)
    auto frag_load = [&](int c, int s2, Frags& f) __attribute__((always_inline)) {
#ifdef SF_WEXP_NOLDS   // timing-only: the fragments are never read from LDS (opaque register values)
#pragma unroll
      for (int t = 0; t < TW; ++t) asm volatile("" : "+v"(f.fa[t]));
#pragma unroll
      for (int p = 0; p < PBW; ++p) asm volatile("" : "+v"(f.b[p]));
      return;
#endif
      const u32x4* sA = reinterpret_cast<const u32x4*>(smem + (c % NB) * SLOT) + lane;
      const u32x4* sB = sA + 16 * 64;
      if constexpr (IN8) {
        if (s2 == 0) {
#pragma unroll
          for (int p = 0; p < PBW; ++p) braw[p] = sB[(pw + p) * 64];
        }
#pragma unroll
        for (int p = 0; p < PBW; ++p) f.b[p] = s2 == 0 ? fp8x8_to_f16(braw[p].x, braw[p].y) : fp8x8_to_f16(braw[p].z, braw[p].w);
      } else {
#pragma unroll
        for (int p = 0; p < PBW; ++p) f.b[p] = sB[((pw + p) * 2 + s2) * 64];
      }
#pragma unroll
      for (int t = 0; t < TW; ++t) f.fa[t] = sA[((t0 + t) * 2 + s2) * 64];
    };
    // cstage >= 0: the G LDS-DMA instructions of chunk cstage are issued one after every second MFMA of this k-step
    auto mma_step = [&](const Frags& f, int cstage) __attribute__((always_inline)) {
      constexpr int PPS = (G + TW - 1) / TW;               // DMA instructions behind each row tile's MFMAs
#pragma unroll
      for (int t = 0; t < TW; ++t) {
#pragma unroll
        for (int p = 0; p < PBW; ++p) acc[t][p] = OP::mfma(f.fa[t], f.b[p], acc[t][p]);
        if (cstage >= 0) {
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = t * PPS; i < (t + 1) * PPS && i < G; ++i) stage_piece(sb, cstage, i);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    };
    // (the 16-bit-phase data gradient holds 64 registers of phases in its last chunks: no room for a second fragment set there)
    constexpr bool DB = !(MODE == 2 && !P8);
    Frags f0 = {}, f1 = {};
    admit(0, true);
    frag_load(0, 0, f0);
    if constexpr (DB) {
      // The LAST chunk is peeled off: with an `if (c + 1 < n2)` around the admit and the next fragment reads, hipcc's wait insertion
      // merged the two paths into the second k-step and put lgkmcnt(3..0) before its MFMAs - in the common path that waits for the
      // six fragment reads issued just before, i.e. the LDS latency of the NEXT k-step was exposed in every chunk.
      for (int c = 0; c + 1 < n2; ++c) {
        frag_load(c, 1, f1);                               // under the MFMAs of k-step (c, 0)
        __builtin_amdgcn_sched_barrier(0);
        mma_step(f0, -1);
        __builtin_amdgcn_sched_barrier(0);
        admit(c + 1, false);
        frag_load(c + 1, 0, f0);                           // under the MFMAs of k-step (c, 1)
        __builtin_amdgcn_sched_barrier(0);
        mma_step(f1, c + 1 + PD < n2 ? c + 1 + PD : -1);   // chunk c + 1 + PD goes into the slot chunk c has just left
        __builtin_amdgcn_sched_barrier(0);
      }
      frag_load(n2 - 1, 1, f1);
      __builtin_amdgcn_sched_barrier(0);
      mma_step(f0, -1);
      mma_step(f1, -1);
    } else {
      for (int c = 0; c < n2; ++c) {
        mma_step(f0, -1);
        frag_load(c, 1, f0);
        mma_step(f0, -1);
        if (c + 1 < n2) {
          admit(c + 1, true);
          frag_load(c + 1, 0, f0);
        }
      }
    }
    (void)f1;
#ifdef SF_WEXP_STAMP
    asm volatile("s_nop 0" ::"v"(acc[0][0][0]), "v"(acc[TW - 1][PBW - 1][15]));
    const unsigned long long st_e0 = __builtin_amdgcn_s_memtime();
#endif
    const int sbn = sb + sb_step;
    const bool more = sbn < a.n_super;
    if (pipe && more) {
      bar_lds();                                           // every wave is done reading this tile's last chunks
      for (int c = 0; c < PD; ++c) stage(sbn, c);
      asm volatile("" ::: "memory");
    }
#ifdef SF_WEXP_NOEPI   // timing-only: no epilogue at all (the accumulators are consumed by an empty asm)
    if (true) {
#pragma unroll
      for (int t = 0; t < TW; ++t)
#pragma unroll
        for (int p = 0; p < PBW; ++p) asm volatile("" ::"v"(acc[t][p]));
    } else
#endif
    if (MODE == 0) {
#pragma unroll
      for (int t = 0; t < TW; ++t)
#pragma unroll
        for (int p = 0; p < PBW; ++p) {
          uint32_t pb8[4] = {0u, 0u, 0u, 0u};
#pragma unroll
          for (int qq = 0; qq < 2; ++qq) {
            float tv[8], ph[8], av[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const float tt = acc[t][p][8 * qq + j];   // revolutions (scale folded into the image)
              tv[j] = tt;
              ph[j] = __builtin_amdgcn_fractf(tt);
              av[j] = __builtin_amdgcn_sinf(tt);
            }
            const long pidx = ((pb0 + p) * a.ks_out + 16 * ob + 2 * (t0 + t) + qq) * 64 + lane;
            if constexpr (P8) {
              pb8[2 * qq] = phase_byte4(tv, av);
              pb8[2 * qq + 1] = phase_byte4(tv + 4, av + 4);
            } else {
              WST(a.Out[pidx], (u32x4{pack_phase2(ph[0], ph[1]), pack_phase2(ph[2], ph[3]), pack_phase2(ph[4], ph[5]),
                                  pack_phase2(ph[6], ph[7])}));
            }
            WST(a.OutAct[pidx], (u32x4{OP::pack2(av[0], av[1]), OP::pack2(av[2], av[3]), OP::pack2(av[4], av[5]),
                                   OP::pack2(av[6], av[7])}));
          }
          if constexpr (P8) WST(a.Out[((pb0 + p) * a.kp_out + 8 * ob + (t0 + t)) * 64 + lane], (u32x4{pb8[0], pb8[1], pb8[2], pb8[3]}));
        }
    } else {
      float fsc = 1.0f;
      if constexpr (OUT8) fsc = a.fscale ? a.fscale[0] : 1.0f;
#pragma unroll
      for (int t = 0; t < TW; ++t)
#pragma unroll
        for (int p = 0; p < PBW; ++p) {
          uint32_t o8[4] = {0u, 0u, 0u, 0u};
#pragma unroll
          for (int qq = 0; qq < 2; ++qq) {
            const long pidx = ((pb0 + p) * a.ks_out + 16 * ob + 2 * (t0 + t) + qq) * 64 + lane;
            float c[8];
            if constexpr (P8) {
              const u32x4 pvv = pv[t][p][0];
              const uint32_t w0 = pvv[2 * qq], w1 = pvv[2 * qq + 1];     // bytes 8 qq .. 8 qq + 7
              c[0] = __builtin_amdgcn_cosf(phase_rev8<0>(w0)); c[1] = __builtin_amdgcn_cosf(phase_rev8<1>(w0));
              c[2] = __builtin_amdgcn_cosf(phase_rev8<2>(w0)); c[3] = __builtin_amdgcn_cosf(phase_rev8<3>(w0));
              c[4] = __builtin_amdgcn_cosf(phase_rev8<0>(w1)); c[5] = __builtin_amdgcn_cosf(phase_rev8<1>(w1));
              c[6] = __builtin_amdgcn_cosf(phase_rev8<2>(w1)); c[7] = __builtin_amdgcn_cosf(phase_rev8<3>(w1));
            } else {
              const u32x4 pvv = pv[t][p][qq];
#pragma unroll
              for (int j2 = 0; j2 < 4; ++j2) {
                c[2 * j2] = __builtin_amdgcn_cosf(phase_rev_lo(pvv[j2]));
                c[2 * j2 + 1] = __builtin_amdgcn_cosf(phase_rev_hi(pvv[j2]));
              }
            }
            if constexpr (OUT8) {
              float v[8];
#pragma unroll
              for (int j = 0; j < 8; ++j) v[j] = acc[t][p][8 * qq + j] * (c[j] * fsc);
              o8[2 * qq] = fp8x4_sat(v[0], v[1], v[2], v[3]);
              o8[2 * qq + 1] = fp8x4_sat(v[4], v[5], v[6], v[7]);
            } else {
              u32x4 o;
#pragma unroll
              for (int j2 = 0; j2 < 4; ++j2)
                o[j2] = OP::pack2(acc[t][p][8 * qq + 2 * j2] * c[2 * j2], acc[t][p][8 * qq + 2 * j2 + 1] * c[2 * j2 + 1]);
              WST(a.Out[pidx], (o));
            }
          }
          if constexpr (OUT8) WST(a.Out[((pb0 + p) * a.kp_out + 8 * ob + (t0 + t)) * 64 + lane], (u32x4{o8[0], o8[1], o8[2], o8[3]}));
        }
    }
#ifdef SF_WEXP_STAMP
    st_epi += __builtin_amdgcn_s_memtime() - st_e0;
    st_tiles += 1;
#endif
    if (!more) break;
    if (!pipe) {
      bar_lds();
      for (int c = 0; c < PD && c < n2; ++c) stage(sbn, c);
    }
    sb = sbn;
    first = false;
  }
#ifdef SF_WEXP_STAMP
  if (a.dump && lane == 0 && (wave == 0 || wave == 5) && (blockIdx.x == 3 || blockIdx.x == 200) && st_tiles) {
    float* o = reinterpret_cast<float*>(a.dump) + (MODE == 0 ? 0 : 32) + ((blockIdx.x == 3 ? 0 : 2) + (wave == 0 ? 0 : 1)) * 4;
    const float nt = (float)st_tiles;
    o[0] = (float)(__builtin_amdgcn_s_memtime() - st_begin) / nt; o[1] = (float)st_wait / nt; o[2] = (float)st_epi / nt; o[3] = nt;
    reinterpret_cast<float*>(a.dump)[(MODE == 0 ? 16 : 48) + (blockIdx.x == 3 ? 0 : 2) + (wave == 0 ? 0 : 1)] =
        (float)(__builtin_amdgcn_s_memtime() - st_begin) / (float)(__builtin_amdgcn_s_memrealtime() - st_rt0) * 100.0f;   // core MHz over the launch
  }
#endif
}

// ---------------------------------------------------------------------------------------------------------
// k_wgemm3 (round 3): the forward hidden GEMM of k_wgemm2<0, .., P8> with its epilogue PIPELINED UNDER THE NEXT TILE.
// An experiment that is kept because of what it rules out (opt-in: SIREN_FIT_WGEMM3=1).  Timing-only builds say the epilogue
// stores of the wide GEMMs cost their full drain time (15 of 57 ms at 512x8, profiles/r03_wide_store_ablation.txt); the
// suspicion was the BURST - with the epilogue at the end of a 256 x 256 tile every wave issues its 24 stores back to back,
// whereas k_fwd_pipe and k_bwd8h, which do not pay for their stores, let them leave one at a time between MFMAs.  So here:
//   tile = 256 neurons x 128 pixels (wave = 4 row tiles x ONE pixel block: 64 accumulator registers, two sets);
//   the operand ring streams CONTINUOUSLY across tiles (chunk g of the workgroup's whole sequence, 24 KiB: 16 A + 8 B pieces);
//   while tile T accumulates in one set, the other set - tile T-1 - is drained: VPK values per k-step (sine, phase byte, pack),
//   an activation store every 8 values, a phase-byte store every 16: 12 stores spread over the tile's k-steps.
//   Every step is the full step (k_bwd8h's rule): the first tile drains zeros into a.dump, requests beyond the last tile re-read
//   its chunks, the last tile is drained once more at the end - so the store pattern is periodic and every vmcnt is a constant.
// RESULT: bit-identical to the tile loop and no faster (forward 20.3 against 20.0 ms per step at 512x8; with the fragments of both
// loops double-buffered 20.5 against 18.6): it is not the burst, and the smaller tile pays more than the hidden epilogue returns.
// KSI = k-steps of the layer (32 at width 512, 64 at 1024).  fp16 operands, phase bytes (scratch formats 12 / 8).
// ---------------------------------------------------------------------------------------------------------
template <int V> struct IntC { static constexpr int value = V; };
template <int KSI>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_wgemm3(WGemmArgs a) {   // (two waves per SIMD, 256 VGPRs: left to itself hipcc spills the drained accumulator set to reach three)
  typedef OpF16 OP;
  constexpr int OT = 8, TW = 4, NPB = 4, NB = 4, PD = 3, SLOT = 24 * 1024, N2 = KSI / 2;
  constexpr int G = 3;                                   // LDS-DMA instructions per wave and chunk: 24 pieces / 8 waves
  constexpr int VPK = 64 / KSI;                          // epilogue values per k-step (2 at width 512, 1 at 1024)
  static_assert(KSI == 32 || KSI == 64, "widths 512 / 1024");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sBias = reinterpret_cast<float*>(smem + NB * SLOT);      // this output block's 256 biases
  const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int xcd = blockIdx.x & 7, q0 = blockIdx.x >> 3;
  const int ob = q0 % a.n_ob;
  const int sb_step = (gridDim.x >> 3) / a.n_ob * 8;
  const int sb0 = (q0 / a.n_ob) * 8 + xcd;               // first 128-pixel unit of this workgroup
  if (sb0 >= a.n_super) return;
  const int n_tiles = (a.n_super - sb0 + sb_step - 1) / sb_step;
  const int t0 = TW * (wave & 1), pbw = wave >> 1;
  const u32x4* Ablk = a.A + (size_t)ob * a.a_block_pieces * 64;
  if (tid < 256) sBias[tid] = a.bias[ob * 256 + tid];
  // chunk g of the workgroup's sequence = chunk g % N2 of tile g / N2 (beyond the last tile: the last tile again)
  auto stage = [&](int ti, int c, int slot) {
    if (ti >= n_tiles) ti = n_tiles - 1;
    char* base = smem + slot * SLOT;
    const long pbg = (long)(sb0 + ti * sb_step) * NPB;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int pc = wave + 8 * i, ot = pc >> 1, s2 = pc & 1;
      glds16s(Ablk + ((size_t)((c >> 1) * OT + ot) * 4 + 2 * (c & 1) + s2) * 64, (uint32_t)lane * 16u, base + pc * 1024);
    }
    glds16s(a.Bin + ((pbg + (wave >> 1)) * KSI + 2 * c + (wave & 1)) * 64, (uint32_t)lane * 16u, base + (16 + wave) * 1024);
  };
  f32x16 accA[TW], accB[TW];
#pragma unroll
  for (int t = 0; t < TW; ++t) { accA[t] = f32x16{}; accB[t] = f32x16{}; }
  static_assert(N2 % NB == 0 && PD < N2, "ring slots are compile-time inside a tile");
  for (int c = 0; c <= PD; ++c) stage(0, c, c);
  // the counted waits below assume the store pattern of a tile BEFORE the first one: with the first four chunks (and the bias
  // table) landed here they hold from step 0 (the later chunks of tile 0 are requested behind real - dumped - stores)
  bar_all();
  u32x4* const dumpw = a.dump + wave * 64 + lane;
  // epilogue value v (tile v >> 4, register v & 15) of accumulator set `prv`: sine -> 16-bit pair, phase byte; an activation
  // piece leaves after every 8 values, the tile's phase piece after 16.  (The running state is plain scalars selected by
  // switches on what are constants after unrolling: a struct with arrays here ended up in scratch memory, with a vmcnt(0)
  // in front of every reload.)
  float e_av = 0.f;
  uint32_t e_a0 = 0u, e_a1 = 0u, e_a2 = 0u, e_a3 = 0u, e_p0 = 0u, e_p1 = 0u, e_p2 = 0u, e_p3 = 0u;
  auto put_byte = [&](uint32_t& w, int byte, float tt, float sv) __attribute__((always_inline)) {
    switch (byte) {
      case 0: phase_byte<0>(w, tt, sv); break;
      case 1: phase_byte<1>(w, tt, sv); break;
      case 2: phase_byte<2>(w, tt, sv); break;
      default: phase_byte<3>(w, tt, sv); break;
    }
  };
  auto epi_value = [&](const f32x16 (&prv)[TW], int v, long pbp, bool to_dump) __attribute__((always_inline)) {
    const int t = v >> 4, e = v & 15;
    const float tt = prv[t][e];
    const float sv = __builtin_amdgcn_sinf(tt);
    switch (e >> 2) {
      case 0: put_byte(e_p0, e & 3, tt, sv); break;
      case 1: put_byte(e_p1, e & 3, tt, sv); break;
      case 2: put_byte(e_p2, e & 3, tt, sv); break;
      default: put_byte(e_p3, e & 3, tt, sv); break;
    }
    if (e & 1) {
      const uint32_t w = OP::pack2(e_av, sv);
      switch ((e & 7) >> 1) {
        case 0: e_a0 = w; break;
        case 1: e_a1 = w; break;
        case 2: e_a2 = w; break;
        default: e_a3 = w; break;
      }
    } else {
      e_av = sv;
    }
    if ((e & 7) == 7) {
      u32x4* dst = to_dump ? dumpw : a.OutAct + ((pbp * KSI + 16 * ob + 2 * (t0 + t) + (e >> 3)) * 64 + lane);
      *dst = u32x4{e_a0, e_a1, e_a2, e_a3};
    }
    if (e == 15) {
      u32x4* dst = to_dump ? dumpw : a.Out + ((pbp * (KSI / 2) + 8 * ob + (t0 + t)) * 64 + lane);
      *dst = u32x4{e_p0, e_p1, e_p2, e_p3};
    }
  };
  // stores issued in half-step hs = 2 c + s2 of a tile (taken modulo the tile: the pattern is the same in every tile)
  auto stores_half = [](int hs) constexpr -> int {
    const int hh = ((hs % (2 * N2)) + 2 * N2) % (2 * N2);
    int n = 0;
    for (int u = 0; u < VPK; ++u) {
      const int e = ((hh * VPK + u) & 15);
      n += ((e & 7) == 7) + (e == 15);
    }
    return n;
  };
  // The operand fragments are double-buffered across half-steps (as in k_wgemm2): f0 holds k-step (c, 0), f1 k-step (c, 1); the
  // barrier that admits chunk c + 1 sits between the two half-steps of chunk c, where chunk c + 1 + PD is requested too.
  struct Frags { u32x4 fa[TW]; u32x4 b; };
  Frags f0, f1;
  auto frag_load = [&](int c, int s2, Frags& f) __attribute__((always_inline)) {
    const u32x4* sA = reinterpret_cast<const u32x4*>(smem + (c % NB) * SLOT) + lane;
    f.b = sA[(16 + pbw * 2 + s2) * 64];
#pragma unroll
    for (int t = 0; t < TW; ++t) f.fa[t] = sA[((t0 + t) * 2 + s2) * 64];
  };
  // one tile: `cur` accumulates (its bias first), `prv` - the tile before - is drained into pixel block pbp (or the dump).
  // The chunk loop is unrolled sixteen chunks at a time through a compile-time part index (a 32-chunk body is not unrolled
  // by hipcc, and an accumulator register indexed at run time goes to scratch memory).
  auto tile_part = [&](auto part_tag, f32x16 (&cur)[TW], const f32x16 (&prv)[TW], int ti, long pbp, bool to_dump) __attribute__((always_inline)) {
    constexpr int PART = decltype(part_tag)::value;
#pragma unroll
    for (int ci = 0; ci < 16; ++ci) {
      const int c = 16 * PART + ci;
      frag_load(c, 1, f1);                               // under the MFMAs of half-step (c, 0)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < TW; ++t) cur[t] = OP::mfma(f0.fa[t], f0.b, cur[t]);
#pragma unroll
      for (int u = 0; u < VPK; ++u) epi_value(prv, (2 * c) * VPK + u, pbp, to_dump);
      __builtin_amdgcn_sched_barrier(0);
      {   // admit chunk c + 1 (of this tile or the first of the next): younger than its DMA are the DMA of the PD - 1 chunks
          // requested since and the stores of the 2 PD half-steps (c - PD, 1) .. (c, 0)
        int nst = 0;
        for (int hs = 2 * (c + 1) - 2 * PD - 1; hs <= 2 * c; ++hs) nst += stores_half(hs);
        switch (nst) {                                   // (a constant per c after unrolling)
          case 0: bar_dma<(PD - 1) * G + 0>(); break;
          case 1: bar_dma<(PD - 1) * G + 1>(); break;
          case 2: bar_dma<(PD - 1) * G + 2>(); break;
          case 3: bar_dma<(PD - 1) * G + 3>(); break;
          case 4: bar_dma<(PD - 1) * G + 4>(); break;
          case 5: bar_dma<(PD - 1) * G + 5>(); break;
          default: bar_dma<(PD - 1) * G + 6>(); break;
        }
        const int cn = c + 1 + PD;                       // chunk requested here: into the slot chunk c has just left
        if (cn < N2) stage(ti, cn, cn % NB); else stage(ti + 1, cn - N2, cn % NB);
        asm volatile("" ::: "memory");
        frag_load(c + 1, 0, f0);                         // under the MFMAs of half-step (c, 1); (chunk N2 = chunk 0 of the next tile: same slot)
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < TW; ++t) cur[t] = OP::mfma(f1.fa[t], f1.b, cur[t]);
#pragma unroll
      for (int u = 0; u < VPK; ++u) epi_value(prv, (2 * c + 1) * VPK + u, pbp, to_dump);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto tile = [&](f32x16 (&cur)[TW], const f32x16 (&prv)[TW], int ti, long pbp, bool to_dump) __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < TW; ++t)
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(&sBias[(t0 + t) * 32 + 8 * q4 + 4 * h]);
        cur[t][4 * q4 + 0] = b.x; cur[t][4 * q4 + 1] = b.y; cur[t][4 * q4 + 2] = b.z; cur[t][4 * q4 + 3] = b.w;
      }
    tile_part(IntC<0>{}, cur, prv, ti, pbp, to_dump);
    if constexpr (N2 > 16) tile_part(IntC<1>{}, cur, prv, ti, pbp, to_dump);
  };
  frag_load(0, 0, f0);
  // tiles alternate between the two accumulator sets
  long pb_prev = 0;
  int ti = 0;
  for (; ti + 1 < n_tiles; ti += 2) {
    tile(accA, accB, ti, pb_prev, ti == 0);
    pb_prev = (long)(sb0 + ti * sb_step) * NPB + pbw;
    tile(accB, accA, ti + 1, pb_prev, false);
    pb_prev = (long)(sb0 + (ti + 1) * sb_step) * NPB + pbw;
  }
  const bool odd = ti < n_tiles;
  if (odd) {
    tile(accA, accB, ti, pb_prev, ti == 0);
    pb_prev = (long)(sb0 + ti * sb_step) * NPB + pbw;
  }
  // drain the last tile (a burst, once per workgroup)
  auto drain = [&](const f32x16 (&prv)[TW]) __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < TW; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) epi_value(prv, 16 * t + e, pb_prev, false);
  };
  if (odd) drain(accA); else drain(accB);
  bar_all();      // no LDS-DMA of this workgroup is in flight when its LDS is handed on
}

// ---------------------------------------------------------------------------------------------------------
// k_wdw: [JW x 256] blocks of a weight gradient: dW[jb-block][ib-block] = delta[:, jb]^T * act[:, ib]
//   ring of 32-pixel blocks as in k_bwd; both operands are read transposed (ds_read_b64_tr_b16).
// ---------------------------------------------------------------------------------------------------------
struct WDwArgs {
  const u32x4* D; int ksd_total;            // delta tensor of layer l: k-steps per pixel block
  const u32x4* P; int ksp_total;            // activation tensor of layer l-1
  int nblk_i;                               // 256-column blocks of the phase tensor: blockIdx.y = jb * nblk_i + ib
  long n_pb;
  float* slab;                              // [gridDim.y][gridDim.x][JW*256 + JW]
};

// D8 (scratch_format 8 on the wide path): the deltas arrive as fp8 byte pieces (a.ksd_total then counts 32-neuron TILES per pixel
// block); a delta^T fragment is one ds_read_b64_tr_b8 + four conversions, its rows come out permuted by nu8 (undone when the slab
// is written) - the operand path of k_bwd8h (siren_s8h.hip), whose lane maps are used here.
template <int JW, typename OP, bool D8 = false>
__global__ __launch_bounds__(512) void k_wdw(WDwArgs a) {
  static_assert(!D8 || JW == 256, "byte deltas: hidden layers");
  constexpr int WAVES_R = JW == 256 ? 2 : 1, WAVES_C = 8 / WAVES_R, NW = 8;
  constexpr int JT = JW / 32, IT = 8, WJ = JT / WAVES_R, WI = IT / WAVES_C;
  constexpr int KSJ = D8 ? JW / 32 : JW / 16, KSI = 16, NB = 4, PD = 3, BLK = (KSJ + KSI) * 1024;   // (KSJ: delta pieces per block)
  constexpr int G = KSJ % NW == 0 ? KSJ / NW + KSI / NW : 0;   // LDS-DMA instructions per wave per block (0: uneven)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / WAVES_C, wc = wave % WAVES_C;
  const int trb = tr_lane_base(lane);
  // all blocks of one layer run in ONE launch; workgroups with the same blockIdx.x stream the same pixel blocks
  // (of different column blocks) at the same time and sit on the same XCD (gridDim.x is a multiple of 8), so
  // each delta / phase piece is pulled from HBM once and served to the other readers by that XCD's L2
  const int jb = blockIdx.y / a.nblk_i, ib = blockIdx.y % a.nblk_i;
  const int ksd_off = (D8 ? 8 : 16) * jb, ksp_off = 16 * ib;
  // byte pieces: LDS slot i holds the element of lane i ^ 8 (i >> 5); transposed byte reads: see k_bwd8h
  const uint32_t aL8 = (uint32_t)(lane ^ ((lane >> 5) << 3)) * 16u;
  const uint32_t aT8 = 16u * (32u * (uint32_t)(lane & 1) + ((8u * (uint32_t)(lane >> 5) + (uint32_t)((lane & 15) >> 1)) ^ (8u * (uint32_t)(lane & 1)))) +
                       8u * (uint32_t)((lane >> 4) & 1);
  const long pb_begin = blockIdx.x, pb_step = gridDim.x;
  const int nblk = (int)((a.n_pb - pb_begin + pb_step - 1) / pb_step);
  f32x16 acc[WJ][WI];
#pragma unroll
  for (int x = 0; x < WJ; ++x)
#pragma unroll
    for (int y = 0; y < WI; ++y) acc[x][y] = f32x16{};
  float dbs[WJ];
#pragma unroll
  for (int x = 0; x < WJ; ++x) dbs[x] = 0.f;
  auto stage = [&](int k) {
    char* base = smem + (k % NB) * BLK;
    const long pb = pb_begin + k * pb_step;
    for (int pc = wave; pc < KSJ; pc += NW)
      glds16s(a.D + (pb * a.ksd_total + ksd_off + pc) * 64, D8 ? aL8 : (uint32_t)sw_lane(lane, pc & 1) * 16u, base + pc * 1024);
    for (int pc = wave; pc < KSI; pc += NW)
      glds16s(a.P + (pb * a.ksp_total + ksp_off + pc) * 64, (uint32_t)sw_lane(lane, pc & 1) * 16u, base + (KSJ + pc) * 1024);
  };
  auto d8_frag = [&](const char* sD, int tile, int kk) -> u32x4 {
    typedef __attribute__((ext_vector_type(2))) int i32x2;
    const uint32_t adr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)sD + aT8 + (uint32_t)(tile * 1024 + kk * 256);
    const i32x2 r = __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) i32x2*)(uintptr_t)adr);
    return fp8x8_to_f16((uint32_t)r.x, (uint32_t)r.y);
  };
  for (int k = 0; k < PD && k < nblk; ++k) stage(k);
#ifdef SF_EXPERIMENT_STAMP   // timing-only build: where one wave's block step goes (wait / DMA issue / compute)
  unsigned long long st_wait = 0, st_stage = 0, st_comp = 0, st_t = __builtin_amdgcn_s_memtime();
#define SF_STAMP(acc_) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); acc_ += n_ - st_t; st_t = n_; } while (0)
#else
#define SF_STAMP(acc_) do {} while (0)
#endif
  for (int k = 0; k < nblk; ++k) {
    // block k landed (blocks k+1, k+2 stay in flight), everyone finished block k-1 (whose slot is refilled next)
    if (G > 0 && k + PD - 1 < nblk) bar_dma<(PD - 1) * G>(); else bar_all();
    SF_STAMP(st_wait);
    // The LDS-DMA of block k+PD is issued BETWEEN the MFMAs of this block, one piece per group of MFMAs: an
    // LDS-DMA instruction costs its wave ~150-200 issue cycles in a burst and ~60 among MFMAs.
    const bool do_stage = k + PD < nblk;
    char* nbase = smem + ((k + PD) % NB) * BLK;
    const long npb = pb_begin + (long)(k + PD) * pb_step;
    auto stage_piece = [&](int i) {     // i-th of this wave's G pieces of block k+PD
      if (!do_stage || i >= G) return;
      constexpr int GD = KSJ / NW;      // delta pieces per wave (D8: one)
      if (i < GD) {
        const int pc = wave + NW * i;
        glds16s(a.D + (npb * a.ksd_total + ksd_off + pc) * 64, D8 ? aL8 : (uint32_t)sw_lane(lane, pc & 1) * 16u, nbase + pc * 1024);
      } else {
        const int pc = wave + NW * (i - GD);
        glds16s(a.P + (npb * a.ksp_total + ksp_off + pc) * 64, (uint32_t)sw_lane(lane, pc & 1) * 16u, nbase + (KSJ + pc) * 1024);
      }
    };
    if (G == 0 && do_stage) stage(k + PD);
    asm volatile("" ::: "memory");
    SF_STAMP(st_stage);
    char* sD = smem + (k % NB) * BLK;
    char* sP = sD + KSJ * 1024;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      u32x4 fa[WJ], fb[WI];
#pragma unroll
      for (int x = 0; x < WJ; ++x) fa[x] = D8 ? d8_frag(sD, wr * WJ + x, kk) : ds_read_tr_frag(sD, trb, wr * WJ + x, kk);
#pragma unroll
      for (int y = 0; y < WI; ++y) fb[y] = ds_read_tr_frag(sP, trb, wc * WI + y, kk);
#pragma unroll
      for (int x = 0; x < WJ; ++x) {
#pragma unroll
        for (int y = 0; y < WI; ++y) acc[x][y] = OP::mfma(fa[x], fb[y], acc[x][y]);
        if (G > 0 && (x & 1) == 1) {      // after every second row tile: 2 * WI MFMAs in flight behind the DMA issue
          __builtin_amdgcn_sched_barrier(0);
          stage_piece(kk * (WJ / 2) + (x >> 1));
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (wc == 0) {
#pragma unroll
        for (int x = 0; x < WJ; ++x) {
          float t = 0.f;
#pragma unroll
          for (int e = 0; e < 4; ++e) t += OP::sum2(fa[x][e]);
          dbs[x] += t;
        }
      }
    }
#ifdef SF_EXPERIMENT_STAMP
    asm volatile("s_nop 0" :: "v"(acc[0][0][0]));   // MFMA results of this step consumed -> chain finished
#endif
    SF_STAMP(st_comp);
  }
#ifdef SF_EXPERIMENT_STAMP
  if (JW == 256 && lane == 0 && (wave == 0 || wave == 7) && (blockIdx.x == 0 || blockIdx.x == 5) && (blockIdx.y == 0 || blockIdx.y == 9)) {
    unsigned long long* dbg = reinterpret_cast<unsigned long long*>(a.slab + (size_t)gridDim.y * gridDim.x * (JW * 256 + JW));
    const int slot = ((blockIdx.x != 0) * 2 + (blockIdx.y != 0)) * 2 + (wave != 0);
    dbg[slot * 4 + 0] = st_wait; dbg[slot * 4 + 1] = st_stage; dbg[slot * 4 + 2] = st_comp; dbg[slot * 4 + 3] = nblk;
  }
#endif
  float* slab = a.slab + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (JW * 256 + JW);
  const int cl = lane & 31, hh = lane >> 5;
#pragma unroll
  for (int x = 0; x < WJ; ++x)
#pragma unroll
    for (int y = 0; y < WI; ++y)
#pragma unroll
      for (int t = 0; t < 16; ++t)
        slab[(size_t)(32 * (wr * WJ + x) + (D8 ? nu8(rho(t, hh)) : rho(t, hh))) * 256 + 32 * (wc * WI + y) + cl] = acc[x][y][t];
  if (wc == 0) {
#pragma unroll
    for (int x = 0; x < WJ; ++x) {
      const float tsum = dbs[x] + __shfl_xor(dbs[x], 32);
      if (hh == 0) slab[JW * 256 + 32 * (wr * WJ + x) + (D8 ? nu8(cl) : cl)] = tsum;
    }
  }
}

// fp8 deltas on the wide path: the chunk's power-of-two factor from its residual.  dL/dout is stored as resid * pre (pre = the
// 16-bit path's static gscale); F = 2^floor(log2(target / (rms(resid) * pre))) brings the first hidden delta to the target rms (its
// own gain is normalised by the last layer's link).  One workgroup, partials summed in double in fixed order; out[0] = F, out[1] = 1 / (F * gpre).
__global__ __launch_bounds__(256) void k_wchunk_scale(const float* sse_part, int n_part, double inv_values, float pre, float gpre, float target, float* out) {
  __shared__ double sh[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < n_part; i += 256) s += (double)sse_part[i];
  sh[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int i = 0; i < 256; ++i) t += sh[i];
    double rms = sqrt(t * inv_values);
    if (!(rms > 1e-12)) rms = 1e-12;
    if (rms > 4.0) rms = 4.0;
    const double F = exp2(floor(log2((double)target / (rms * (double)pre))));
    out[0] = (float)F; out[1] = (float)(1.0 / (F * (double)gpre));     // [1]: what a weight gradient formed from these deltas is multiplied by
  }
}

// block-wise slab reduction into a sub-block of a [rows_total x ldw] gradient matrix (fixed order)
struct WReduceArgs {
  const float* slab; int n_wg, slab_rows;       // slab = [gridDim.y blocks][n_wg][slab_rows*256 + slab_rows]
  int rows_out;                                 // valid rows of a block (256, or out_features for the last layer)
  int nblk_i;                                   // blockIdx.y = jb * nblk_i + ib
  float* gW; int ldw;                           // gradient matrix [rows][ldw]; block (jb, ib) starts at [256 jb][256 ib]
  float* gb;                                    // bias gradient (taken from the ib == 0 blocks)
  int accumulate; float scale;
  const float* s1; const float* s2;             // fp8 deltas: s1[1] = 1 / (chunk factor * gpre) REPLACES scale, s2[0] = 1 / cumulative layer scale
};
__global__ void k_wreduce(WReduceArgs a) {
  const float mul = (a.s1 ? a.s1[1] : a.scale) * (a.s2 ? a.s2[0] : 1.0f);      // (powers of two: exact)
  const long slab_sz = (long)a.slab_rows * 256 + a.slab_rows;
  const int jb = blockIdx.y / a.nblk_i, ib = blockIdx.y % a.nblk_i;
  const float* slab = a.slab + (size_t)blockIdx.y * a.n_wg * slab_sz;
  const int nW = a.rows_out * 256;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= nW + (ib == 0 ? a.rows_out : 0)) return;
  if (idx < nW) {
    const int j = idx >> 8, i = idx & 255;
    const float* p = slab + (long)j * 256 + i;
    float s = 0.f;
    for (int w = 0; w < a.n_wg; ++w) s += p[w * slab_sz];
    s *= mul;
    float* o = a.gW + (long)(256 * jb + j) * a.ldw + 256 * ib + i;
    *o = a.accumulate ? *o + s : s;
  } else {
    const int j = idx - nW;
    const float* p = slab + (long)a.slab_rows * 256 + j;
    float s = 0.f;
    for (int w = 0; w < a.n_wg; ++w) s += p[w * slab_sz];
    s *= mul;
    float* o = a.gb + 256 * jb + j;
    *o = a.accumulate ? *o + s : s;
  }
}

}  // namespace sf
