// siren_fwd16.hip — k_fwd_pipe16: k_fwd_pipe (siren_kernels.hip) re-tiled from v_mfma_f32_32x32x16 to v_mfma_f32_16x16x32
// (round 3).
//
// Arithmetic: that of k_fwd_pipe - implicit_image/models/siren.py:56-68,123-134 (Linear -> sin(omega z) stack, (x - 0.5) * 2,
// out / 2 + 0.5), train_helper.py:147-161 (mse) - on the same weights, with the same phase-byte scratch and the same
// dL/dout piece: nothing downstream of this kernel changes.
//
// Why: the forward is the most MFMA-dense kernel of the step and runs at the clock the package power cap leaves it (1.45
// GHz).  In a loop shaped like k_fwd_pipe's (every A fragment re-read from LDS, B in registers, the sine epilogue riding
// along) the 16x16x32 shape holds 1.9 GHz where the 32x32x16 shape holds 1.6 at equal cycles per FLOP
// (scripts/probes/mfma_shape.hip, profiles/r03_mfma_shape_probe.txt; MI355X_MICROARCH.md 'DVFS give-back' item 7).
//
// Geometry.  A wave still owns 32 pixels and all 256 neurons; a 32-neuron x 32-pixel tile is 2 x 2 accumulators of 16 x 16:
//   v_mfma_f32_16x16x32_f16:  D[16x16] += A[16x32] * B[32x16]
//     A: lane (r16 = l & 15, lg = l >> 4) holds row r16, k-slots 8 lg + j (j = 0..7);  B: lane (c, lg) holds column c, the
//     same k-slots;  C/D: lane (c, lg) holds column c, register r holds row 4 lg + r.
//   acc[rh][ph] (rh = row half, ph = pixel half):  lane (c, lg), register r = neuron 32 nt + 16 rh + 4 lg + r of pixel
//     16 ph + c.  The eight sines a lane holds for pixel half ph - (rh, r) - ARE its B fragment of k-step nt (32 neurons) of
//     the next layer, with k-slot 8 lg + j = neuron 32 nt + 16 (j >> 2) + 4 lg + (j & 3): the weight images carry that order.
//   Per tile: 16 A fragments (piece a = 2 k32 + rh, one ds_read_b128 each, feeding the two pixel halves) and 32 MFMAs; a
//     "slot" is one A fragment = two MFMAs, so the slot plan of k_fwd_pipe (16 slots per tile, the previous tile's sixteen
//     epilogue values spread over slots 1..13, barriers PD slots before the reads change halves, DMA pieces every third slot)
//     carries over unchanged.
//   Phase bytes: a lane's four dwords N[ph][rh] (bytes r = 0..3) hold other (pixel, neuron) pairs than the four dwords
//     O[q] of the 32x32 geometry the backward kernels read (lane (m, h): pixel m, neurons 8 q + 4 h + j).  With rows of 16
//     lanes, O[q].row(2 h + p) = N[p][q >> 1].row(2 (q & 1) + h): v_permlane16_swap + v_permlane32_swap on the register pair
//     (N[0][rh], N[1][rh]) yield (O[2 rh], O[2 rh + 1]) - four VALU instructions per tile, and the scratch layout stays.
//   The output layer's accumulator is brought into the 32x32 geometry's lower lane half the same way, so fwd_residual()
//     and the coordinate / target fetch of k_fwd_pipe are used as they are.
// Formats: phase bytes (TRAIN) or none (evaluation).  The 16-bit phase formats keep k_fwd_pipe.
// (included by siren_fit.hip after siren_kernels.hip)

namespace sf {

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;

struct OpF16x {   // 16x16x32 MFMA on fp16 operands
  static DEV f32x4 mfma(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
  }
};

// rows of 16 lanes: x.row1 <-> y.row0, x.row3 <-> y.row2  (the builtins, not inline asm: hipcc then pads the wait states
// between an MFMA result and these reads itself)
DEV void swap16(uint32_t& x, uint32_t& y) {
  const auto r = __builtin_amdgcn_permlane16_swap(x, y, false, false);
  x = r[0]; y = r[1];
}
// x.rows{2,3} <-> y.rows{0,1}
DEV void swap32(uint32_t& x, uint32_t& y) {
  const auto r = __builtin_amdgcn_permlane32_swap(x, y, false, false);
  x = r[0]; y = r[1];
}
// (__builtin_bit_cast applied DIRECTLY to an element of an ext_vector yields element 0 whatever the index with this hipcc -
//  seen as an output layer whose three channels were all channel 0: go through a scalar)
DEV uint32_t f32_bits(float f) { return __builtin_bit_cast(uint32_t, f); }

// ---- weight images of the 16x16x32 geometry (same piece placement as FwdImg<256>) ---------------------------------------
struct Img16Args {
  const float* params;
  int depth, out_features;
  long off_w[16], off_b[16];
  float wscale, hscale, sc_first;
  uint16_t* wf; uint16_t* wf_last; uint16_t* l0img;
};
__global__ void k_images16(Img16Args a) {
  constexpr int WD = 256, NT = 8;
  const FwdGeom G(WD);
  const long per_layer = (long)WD * WD;
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  auto kslot = [](int k32, int lg, int j) { return 32 * k32 + 16 * (j >> 2) + 4 * lg + (j & 3); };
  if (gid < (long)(a.depth - 2) * per_layer) {   // hidden images [tile][piece = 2 k32 + rh][lane][8]
    const int l = (int)(gid / per_layer) + 1;
    long e = gid % per_layer;
    const int j = e & 7; e >>= 3;
    const int lane = e & 63; e >>= 6;
    const int pc = (int)(e & 15), tile = (int)(e >> 4);
    const int r16 = lane & 15, lg = lane >> 4, k32 = pc >> 1, rh = pc & 1;
    const float w = a.params[a.off_w[l] + (long)(32 * tile + 16 * rh + r16) * WD + kslot(k32, lg, j)] * a.hscale;
    a.wf[((long)(l - 1) * G.PIECES + G.tile_piece(tile) + pc) * 512 + lane * 8 + j] = to_f16(w);
  }
  if (gid < (long)(a.depth - 2) * WD) {          // hidden biases (fp32, pre-scaled), as in k_images
    const int l = (int)(gid / WD) + 1, n = (int)(gid % WD), nt = n / 32;
    float* piece = reinterpret_cast<float*>(a.wf + ((long)(l - 1) * G.PIECES + G.bias_piece(nt)) * 512);
    piece[G.bias_off(nt) + (n & 31)] = a.params[a.off_b[l] + n] * a.hscale;
  }
  const int L = a.depth - 1;
  if (gid < 8L * 64 * 8) {                       // output layer: one 16-row tile, pieces k32 = 0..7, then the bias piece
    long e = gid;
    const int j = e & 7; e >>= 3;
    const int lane = e & 63; e >>= 6;
    const int k32 = (int)e, r16 = lane & 15, lg = lane >> 4;
    float w = 0.f;
    if (r16 < a.out_features) w = a.params[a.off_w[L] + (long)r16 * WD + kslot(k32, lg, j)] * a.wscale;
    a.wf_last[gid] = to_f16(w);
  }
  if (gid < 32) {
    float* piece = reinterpret_cast<float*>(a.wf_last + 8L * 512);
    piece[gid] = (gid < a.out_features) ? a.params[a.off_b[L] + gid] * a.wscale : 0.f;
  }
  if (gid < (long)NT * 64) {   // layer 0 (kL0Split): lanes 0..31 of piece `tile` = rows 16 rh + r16 (k-slots 0..7), lanes 32..63 zero
    const int lane = (int)(gid & 63), tile = (int)(gid >> 6), n = 32 * tile + (lane & 31);
    u32x4 v = u32x4{0u, 0u, 0u, 0u};
    if (lane < 32) {
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

#ifndef SF_FWD16_ORDER
#define SF_FWD16_ORDER 0    // placement of a slot's two MFMAs against its epilogue values: 0 = hipcc's
#endif
constexpr int kLast16Pieces = 9;    // output-layer image: 8 fragments + bias piece

template <bool TRAIN, int PD = SF_FWD_PD>
__global__ __launch_bounds__(512) void k_fwd_pipe16(FwdArgs a) {
  typedef OpF16 OP;
  constexpr int WD = 256;
  using IM = FwdImg<WD>;
  constexpr int NT = IM::NT, KS = IM::KS, H0 = IM::H0, NG = NT * KS, K32 = 8;
  constexpr int ST_SLOT = KS - 3;
  constexpr int DMA0 = 2, DSP = 3, NPC = 9;
  constexpr int NST = TRAIN ? 2 : 0;
  static_assert(DMA0 + DSP * (NPC - 1) < ST_SLOT + KS && DMA0 + DSP * (NPC - 1) > ST_SLOT && (ST_SLOT - DMA0) % DSP != 0 &&
                NPC * kWavesFwd >= IM::X_PIECES && NPC * kWavesFwd >= IM::Y_PIECES, "DMA slot plan");
  static_assert(2 * H0 == NT && PD >= 3 && PD <= 8 && KS == 16, "slot plan");
#ifdef SF_EXPERIMENT_STAMP
  const unsigned long long st_entry = __builtin_amdgcn_s_memtime();
  unsigned long long st_mid = 0, st_end = 0, st_l0 = 0, st_x1 = 0, st_pipe = 0;
#endif
  extern __shared__ __attribute__((aligned(16))) char smem[];
  u32x4* sW = reinterpret_cast<u32x4*>(smem);
  u32x4* sL0 = reinterpret_cast<u32x4*>(smem + (size_t)IM::PIECES * 1024);
  float* sRed = reinterpret_cast<float*>(sL0 + NT * 64);
  (void)sW;

  const int tid = threadIdx.x, lane = tid & 63, m = lane & 31, h = lane >> 5, c16 = lane & 15, lg = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int L = a.depth - 2;
  glds16s(a.l0img + wave * 64, (uint32_t)lane * 16u, reinterpret_cast<char*>(sL0) + (size_t)wave * 1024);
  asm volatile("" ::: "memory");

  long pb = 0, pix = 0;
  bool valid = false;
  struct Fetch { float gh, gw, t[3]; };
  auto fetch = [&](int grp) -> Fetch {           // raw grid values and target of pixel m = lane & 31 of group grp (as k_fwd_pipe)
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
  bar_dma<0>();

  auto slot_end = [&]() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  typedef __attribute__((address_space(3))) const u32x4 lds_cv4;
  typedef __attribute__((address_space(3))) const f32x4 lds_cf4;
  __builtin_assume(wave >= 0 && wave < kWavesFwd);
  const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem);
  const uint32_t lane16 = (uint32_t)lane * 16u;
  const uint32_t lds_wave = lds0 + (uint32_t)wave * 1024u;
  uint32_t aXf = lds0 + lane16;
  uint32_t aYf = lds0 + (uint32_t)IM::X_PIECES * 1024u + lane16;
  uint32_t aXb = lds0 + (uint32_t)IM::bias_piece(0) * 1024u + (uint32_t)lg * 16u;      // rows 4 lg .. 4 lg + 3 of a row half
  uint32_t aYb = lds0 + (uint32_t)IM::bias_piece(H0) * 1024u + (uint32_t)lg * 16u;
  uint32_t aLb = lds0 + (uint32_t)K32 * 1024u + (uint32_t)lg * 16u;                    // bias piece of the output layer
  // layer-0 image: lanes of lane group 0 read their row's 16 bytes, every other lane reads zeros (the piece's upper half)
  uint32_t aL0 = lds0 + (uint32_t)IM::PIECES * 1024u + (lg == 0 ? (uint32_t)c16 * 16u : 512u);
  asm volatile("" : "+v"(aXf), "+v"(aYf), "+v"(aXb), "+v"(aYb), "+v"(aLb), "+v"(aL0));
  auto frag = [&](int nt, int pc) -> u32x4 {   // fragment (tile, piece = 2 k32 + rh) of the hidden image in LDS
    return nt < H0 ? *(lds_cv4*)(uintptr_t)(aXf + (uint32_t)(nt * KS + pc) * 1024u)
                   : *(lds_cv4*)(uintptr_t)(aYf + (uint32_t)((nt - H0) * KS + pc) * 1024u);
  };
  struct Acc { f32x4 v[2][2]; };                 // [rh][ph]
  auto tile_bias = [&](uint32_t ab, int rh, Acc& t) {
    // (two reads of the same 16 bytes: cheaper than four v_mov)
    t.v[rh][0] = *(lds_cf4*)(uintptr_t)(ab + (uint32_t)rh * 64u);
    t.v[rh][1] = *(lds_cf4*)(uintptr_t)(ab + (uint32_t)rh * 64u);
  };
  auto hidden_bias = [&](int nt) -> uint32_t { return (nt < H0 ? aXb : aYb) + (uint32_t)IM::bias_off(nt) * 4u; };

  u32x4 Ba[2 * K32], Bb[2 * K32];                // activations: [ph * 8 + k32]
  Acc prev, cur;
  u32x4 fr[PD];

  // epilogue value e = 8 ph + 4 rh + r of accumulator tile `t` -> sine into dword (e & 7) >> 1 of dst[e >> 3], phase byte
  // r of N[ph][rh] = pb8[e >> 2]
  struct Epi { float av[16]; uint32_t pb8[4]; };
  auto epi_value = [&](Epi& E, const Acc& t, int e, u32x4* d0, u32x4* d1) {
    const float tt = t.v[(e >> 2) & 1][e >> 3][e & 3];
    E.av[e] = __builtin_amdgcn_sinf(tt);
    if constexpr (TRAIN) {
      switch (e & 3) {
        case 0: phase_byte<0>(E.pb8[e >> 2], tt, E.av[e]); break;
        case 1: phase_byte<1>(E.pb8[e >> 2], tt, E.av[e]); break;
        case 2: phase_byte<2>(E.pb8[e >> 2], tt, E.av[e]); break;
        default: phase_byte<3>(E.pb8[e >> 2], tt, E.av[e]); break;
      }
    }
    if (e & 1) {
      uint32_t w = OP::pack2(E.av[e - 1], E.av[e]);
      asm volatile("" : "+v"(w));              // pinned to its slot (see k_fwd_pipe)
      (e >> 3 ? d1 : d0)[0][(e & 7) >> 1] = w;
    }
  };
  // phase bytes of a finished tile into the 32x32 geometry (see the header) and out
#ifdef SF_EXP_NOSWAP   // timing-only: phase bytes stored in this kernel's own lane order (the backward then reads them wrongly)
  auto phase_pair = [&](Epi&, int) {};
#else
  auto phase_pair = [&](Epi& E, int rh) { swap16(E.pb8[rh], E.pb8[2 + rh]); swap32(E.pb8[rh], E.pb8[2 + rh]); };
#endif
  auto phase_store = [&](Epi& E, u32x4* ptile) { store_stream(ptile, u32x4{E.pb8[0], E.pb8[2], E.pb8[1], E.pb8[3]}); };
  // slot plan of the epilogue: none in slot 0, two values per slot in slots 1..3, one in 4..13 (value order: 0..15), the
  // first register pair is converted in slot 10 (values 0..3 and 8..11 are done), the second in the store slot 13
  auto epi_slot = [&](Epi& E, const Acc& t, int s, u32x4* d0, u32x4* d1, u32x4* ptile) {
    if (s >= 1 && s <= 3) { epi_value(E, t, 2 * s - 2, d0, d1); epi_value(E, t, 2 * s - 1, d0, d1); }
    if (s >= 4 && s <= 13) epi_value(E, t, s + 2, d0, d1);
    if constexpr (TRAIN) {
      if (s == 10) phase_pair(E, 0);
      if (s == ST_SLOT) { phase_pair(E, 1); phase_store(E, ptile); }
    }
  };

  // ---- layer 0: four 16x16x32 MFMAs per tile (k-slots 0..7 carry the split products, see kL0Split) ----------------------
  auto layer0 = [&](float x0, float x1) {
    u32x4 bx[2];
    {
      _Float16 x0h, x0l, x1h, x1l;
      split_f16(x0, kL0Split, x0h, x0l);
      split_f16(x1, kL0Split, x1h, x1l);
      const _Float16 x0s = (_Float16)((float)x0h * (1.0f / kL0Split)), x1s = (_Float16)((float)x1h * (1.0f / kL0Split));
      // every lane holds the values of pixel m = lane & 31; lane group 0 needs pixel c16 (its own) and pixel 16 + c16 (held
      // by lane group 1): x.row1 <-> y.row0 on copies
      u32x4 own = u32x4{pack_h2(x0h, x0l), pack_h2(x0s, x1h), pack_h2(x1l, x1s), pack_h2((_Float16)1.0f, (_Float16)(1.0f / kL0Split))};
      u32x4 oth = own;
#pragma unroll
      for (int i = 0; i < 4; ++i) { uint32_t xa = own[i], xb = oth[i]; swap16(xa, xb); oth[i] = xb; }
      const u32x4 z4 = u32x4{0u, 0u, 0u, 0u};
      bx[0] = lg == 0 ? own : z4;
      bx[1] = lg == 0 ? oth : z4;
    }
    auto l0_tile = [&](int nt, Acc& z) {
#pragma unroll
      for (int rh = 0; rh < 2; ++rh) {
        const u32x4 af = *(lds_cv4*)(uintptr_t)(aL0 + (uint32_t)nt * 1024u + (uint32_t)rh * 256u);
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) z.v[rh][ph] = OpF16x::mfma(af, bx[ph], f32x4{0.f, 0.f, 0.f, 0.f});
      }
    };
    auto x1_piece = [&](int i) {
      const int pc = wave + kWavesFwd * i;
      if (pc < IM::X_PIECES) glds16o(a.wf + (size_t)pc * 64, lane16, lds_wave + (uint32_t)(kWavesFwd * i) * 1024u);
    };
    Acc z;
    l0_tile(0, z);
#pragma unroll
    for (int nt = 1; nt < NT; ++nt) {
      Acc zn;
      l0_tile(nt, zn);
      if (nt <= 2) { x1_piece(2 * nt - 2); x1_piece(2 * nt - 1); } else x1_piece(nt + 1);
      Epi E;
#pragma unroll
      for (int e = 0; e < 16; ++e) epi_value(E, z, e, &Ba[nt - 1], &Ba[K32 + nt - 1]);
      if constexpr (TRAIN) {
        phase_pair(E, 0); phase_pair(E, 1);
        phase_store(E, a.P + ((size_t)pb * NT + (nt - 1)) * 64 + lane);
      }
      slot_end();
      z = zn;
    }
    prev = z;
  };
  static_assert(NT + 1 == 9 && 9 * kWavesFwd >= IM::X_PIECES, "layer 0 issues pieces 0..8 of half X");

  // one hidden layer l: reads activations Bi, writes Bo (k-steps 0..6; k-step 7 follows in the next stage)
  auto layer = [&](u32x4 (&Bi)[2 * K32], u32x4 (&Bo)[2 * K32], int l) {
    const bool more = l < L;
    const u32x4* srcY = a.wf + ((size_t)(l - 1) * IM::PIECES + IM::X_PIECES) * 64;
    const u32x4* srcX = more ? a.wf + (size_t)l * IM::PIECES * 64 : a.wf_last;
    const int nX = more ? IM::X_PIECES : kLast16Pieces;
    u32x4* pl = a.P + (size_t)l * a.p_stride + (size_t)pb * NT * 64 + lane;
    const uint32_t bias_next0 = more ? hidden_bias(0) : aLb;
    u32x4 ring[NG + PD];
#pragma unroll
    for (int i = 0; i < PD; ++i) ring[i] = fr[i];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      Epi E;
      Acc nxt;
      u32x4* d0 = nt == 0 ? &Bi[K32 - 1] : &Bo[nt - 1];
      u32x4* d1 = nt == 0 ? &Bi[2 * K32 - 1] : &Bo[K32 + nt - 1];
      u32x4* ptile = nt == 0 ? pl - a.p_stride + (NT - 1) * 64 : pl + (nt - 1) * 64;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const int g = nt * KS + s, k32 = s >> 1, rh = s & 1;
        if (g == NG / 2 - PD) {      // every read of half X is issued: half Y landed, everyone left half X
#ifdef SF_EXPERIMENT_STAMP
          const unsigned long long t_a = __builtin_amdgcn_s_memtime();
#endif
          bar_dma<NST>();
#ifdef SF_EXPERIMENT_STAMP
          st_mid += __builtin_amdgcn_s_memtime() - t_a;
#endif
        }
        if (g == NG - PD) {          // every read of half Y is issued: the next layer's half X landed
#ifdef SF_EXPERIMENT_STAMP
          const unsigned long long t_a = __builtin_amdgcn_s_memtime();
#endif
          bar_dma<NST>();
#ifdef SF_EXPERIMENT_STAMP
          st_end += __builtin_amdgcn_s_memtime() - t_a;
#endif
        }
        {
          const int r = g % (NG / 2);
          if (r >= DMA0 && (r - DMA0) % DSP == 0 && (r - DMA0) / DSP < NPC) {
            const int i8 = kWavesFwd * ((r - DMA0) / DSP), pc = wave + i8;
            if (g < NG / 2) {
              if (pc < IM::Y_PIECES) glds16o(srcY + (size_t)pc * 64, lane16, lds_wave + (uint32_t)(IM::X_PIECES + i8) * 1024u);
            } else {
              if (pc < nX) glds16o(srcX + (size_t)pc * 64, lane16, lds_wave + (uint32_t)i8 * 1024u);
            }
          }
        }
        const int f = g + PD;
        // (f >= NG: the first fragments of the next layer's tile 0 - or of the output layer, whose piece k32 feeds slots
        //  2 k32 and 2 k32 + 1: its fragments 0 .. PD/2 - 1 are read twice here, which keeps this loop uniform)
        ring[f] = f < NG ? frag(f / KS, f % KS) : (more ? frag(0, f - NG) : frag(0, (f - NG) >> 1));
#if SF_FWD16_ORDER == 2     // MFMA, epilogue, MFMA
        cur.v[rh][0] = OpF16x::mfma(ring[g], Bi[k32], cur.v[rh][0]);
        __builtin_amdgcn_sched_barrier(0);
        epi_slot(E, prev, s, d0, d1, ptile);
        __builtin_amdgcn_sched_barrier(0);
        cur.v[rh][1] = OpF16x::mfma(ring[g], Bi[K32 + k32], cur.v[rh][1]);
#else
        cur.v[rh][0] = OpF16x::mfma(ring[g], Bi[k32], cur.v[rh][0]);
        cur.v[rh][1] = OpF16x::mfma(ring[g], Bi[K32 + k32], cur.v[rh][1]);
#if SF_FWD16_ORDER == 1     // both MFMAs, then the epilogue
        __builtin_amdgcn_sched_barrier(0);
#endif
        epi_slot(E, prev, s, d0, d1, ptile);
#endif
        if (s == KS - 4) {   // the next tile's bias = its accumulator
          tile_bias(nt + 1 < NT ? hidden_bias(nt + 1) : bias_next0, 0, nxt);
          tile_bias(nt + 1 < NT ? hidden_bias(nt + 1) : bias_next0, 1, nxt);
        }
        slot_end();
      }
      prev = cur;
      cur = nxt;
    }
#pragma unroll
    for (int i = 0; i < PD; ++i) fr[i] = ring[NG + i];
  };
  // output layer (one 16-row tile: row half 0 only; slot s = (k32, ph)) under the epilogue of the last hidden tile
  auto output_layer = [&](u32x4 (&Bi)[2 * K32]) {
    u32x4* ptile = a.P + (size_t)L * a.p_stride + ((size_t)pb * NT + (NT - 1)) * 64 + lane;
    u32x4 ring[KS + PD];
    Epi E;
#pragma unroll
    for (int i = 0; i < PD; ++i) ring[i] = fr[i];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int k32 = s >> 1, ph = s & 1;
      if (s + PD < KS) ring[s + PD] = frag(0, (s + PD) >> 1);
      if (s == KS - PD) bar_lds();     // every read of the output image is issued: the next group may overwrite half X
      cur.v[0][ph] = OpF16x::mfma(ring[s], Bi[ph * K32 + k32], cur.v[0][ph]);
      epi_slot(E, prev, s, &Bi[K32 - 1], &Bi[2 * K32 - 1], ptile);
      slot_end();
    }
  };
  for (int grp = (int)blockIdx.x; grp < a.n_super; grp += (int)gridDim.x) {
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
    if (grp + (int)gridDim.x < a.n_super) nxt_f = fetch(grp + (int)gridDim.x);
#pragma unroll
    for (int i = 0; i < PD; ++i) fr[i] = frag(0, i);
    tile_bias(hidden_bias(0), 0, cur);
    tile_bias(hidden_bias(0), 1, cur);
    slot_end();
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
    // rows 0..2 of the output tile into the lower lane half of the 32x32 geometry (lane m <- pixel m): row 1 of the
    // register of pixel half 0 takes row 0 of the register of pixel half 1
    f32x16 accO = f32x16{};
    {
      uint32_t a0 = f32_bits(cur.v[0][0][0]), b0 = f32_bits(cur.v[0][1][0]);
      uint32_t a1 = f32_bits(cur.v[0][0][1]), b1 = f32_bits(cur.v[0][1][1]);
      uint32_t a2 = f32_bits(cur.v[0][0][2]), b2 = f32_bits(cur.v[0][1][2]);
      swap16(a0, b0); swap16(a1, b1); swap16(a2, b2);
      accO[0] = __builtin_bit_cast(float, a0); accO[1] = __builtin_bit_cast(float, a1); accO[2] = __builtin_bit_cast(float, a2);
    }
    sse_acc += fwd_residual<OP, TRAIN, true>(a, accO, tgt, pix, pb, valid, lane, h);
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

}  // namespace sf
