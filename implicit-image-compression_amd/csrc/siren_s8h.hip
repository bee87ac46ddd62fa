// siren_s8h.hip — k_bwd8h: the backward of one 256 x 256 HIDDEN layer from 8-bit scratch (phase bytes + fp8 deltas,
// sf_config.scratch_format = 8) as ONE hand-cut software pipeline, one "slot" per MFMA (round 3).
//
// Arithmetic: that of k_bwd8<256, 256, 2, 4, false, false, OpF16, .., D8 = true> (siren_s8.hip), bit for bit - the
// autograd backward of implicit_image/models/siren.py:56-68 for one hidden layer l:
//     delta_{l-1} = (W_l^T delta_l) * omega cos(phi_{l-1})        phase X + epilogue E  (fp8 out)
//     dW_l       += delta_l^T sin(phi_{l-1}),  db_l += sum delta_l   phase W
// What changed is the instruction stream.  k_bwd8 left the order inside a phase to hipcc and ran at the SUM of its matrix
// time and its vector time (32 MFMAs and ~205 VALU / 59 LDS instructions per wave and 32-pixel block, 4 100 cycles for
// 2 048 cycles of MFMA: DESIGN.md section 4a).  Here:
//   * fewer instructions per value.  The fp8 saturation is the hardware's (MODE.FP16_OVFL = 1 makes v_cvt_pk_fp8_f32 clamp
//     to +-448 instead of returning NaN: scripts/probes/epi_probe.hip) - no v_med3 per value; cos * g is ONE v_fma_mix_f32
//     (f16 half * f32); the sin/cos table sits at LDS address 0, so a table address is one v_lshlrev_b32_sdwa of the
//     phase byte; all sixteen phase bytes of a lane come from one ds_read_b128 and all sixteen fp8 results leave in one
//     16-byte store.
//   * nothing is converted twice.  k_bwd8 read the delta^T fragments of phase W as bytes (ds_read_b64_tr_b8) and expanded
//     them in registers - the four column waves of a row group each converting the same fragments (32 conversions per wave
//     and block).  Here the block is expanded ONCE into a 16-bit image (X16, three buffers) that phase X reads
//     lane-linearly and phase W reads transposed (ds_read_b64_tr_b16): 8 conversions per wave and block.  The byte ring
//     then only feeds the conversion: three slots hold the three blocks in flight.
//   * the bias-gradient row sums are taken by ONE of the four column waves that hold the same fragments; which one is a
//     compile-time constant of the wave's loop (the loop is instantiated per column wave), not a select per chunk.
//   * every LDS read is issued a fixed number of slots before its first use (operand fragments, phase bytes, table pairs),
//     the epilogue of block k is cut into sixteen single values, one per slot of phase W(k-1), and nothing crosses a
//     slot boundary (sched_barrier + compiler memory barrier, as in k_fwd_pipe).
//
// LDS of one workgroup (8 waves, 153 KiB with PARK = 2, NBP = 4):
//   T     1 KiB          sin/cos pairs of the 256 phase bytes (two 16-bit floats per entry), at address 0
//   RD    3 x  8 KiB     fp8 delta pieces of the blocks in flight (block k+3 is requested in step k, expanded in step k+2)
//   RP    NBP x 8 KiB    phase-byte pieces (block k+NBP-1 is requested in step k, decoded in step k+NBP-1)
//   X16   3 x 16 KiB     fp16 deltas: written in step k-1 (C), B operand in step k (X), read transposed in step k+1 (W)
//   S16   2 x 16 KiB     fp16 sin(phase): written by the epilogue in step k, read transposed in step k+1 (W)
//   WP    8 x PARK KiB   the last PARK k-steps of every wave's stationary W_l^T rows (the rest lives in registers)
// (included by siren_fit.hip after siren_s8.hip)

namespace sf {

#ifndef SF_BWD8H_PF
#define SF_BWD8H_PF 1      // B pieces of phase X read ahead (register ring of PF + 1)
#endif
#ifndef SF_BWD8H_LD
#define SF_BWD8H_LD 2      // slots between a table lookup and the value that uses it
#endif

template <int PARK, int NBP>
constexpr size_t bwd8h_lds_bytes() { return (size_t)(1 + 3 * 8 + NBP * 8 + 3 * 16 + 2 * 16 + 8 * PARK) * 1024; }

// DBX: the row tile (0..3) whose bias-gradient sums this instantiation takes (= the wave's column index wc)
template <int PARK, int NBP, int DBX>
__device__ __forceinline__ void bwd8h_body(const Bwd8Args& a, const int wave, const int lane) {
  typedef OpF16 OP;
  constexpr int KS = 16, NT = 8, NBD = 3, WJ = 4, WI = 2;
  constexpr int AD = NBD, AP = NBP - 1;          // blocks requested ahead (deltas / phases)
  constexpr int PF = SF_BWD8H_PF, NXB = PF + 1, LD = SF_BWD8H_LD, NTAB = LD + 2;   // (table ring: the pair of value t-1 is still read in slot t)
  constexpr int KSR = KS - PARK;
  constexpr uint32_t oRD = 1024, oRP = oRD + NBD * 8192, oX16 = oRP + NBP * 8192, oS16 = oX16 + 3 * 16384, oWP = oS16 + 2 * 16384;
  static_assert(PF >= 1 && PF <= 3 && LD >= 1 && LD <= 4 && PARK >= 0 && PARK <= 4 && NBP >= 3, "slot plan");
  typedef __attribute__((address_space(3))) const u32x4 lds_cv4;
  typedef __attribute__((address_space(3))) u32x4 lds_v4;
  typedef __attribute__((address_space(3))) const uint32_t lds_cu32;
  typedef __attribute__((address_space(3))) u32x2 lds_v2;
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  typedef __attribute__((ext_vector_type(2))) _Float16 h2;

  const int wr = wave >> 2, wc = wave & 3;
  // lane patterns of the LDS accesses (everything else is a wave-uniform offset or a 16-bit immediate)
  const uint32_t aLN = (uint32_t)lane * 16u;                                   // lane-linear piece element
  const uint32_t aL1 = (uint32_t)sw_lane(lane, 0) * 16u, aL1x = aL1 ^ 128u;     // swizzled piece element, even / odd k-step
  const uint32_t aT1 = (uint32_t)tr_lane_base(lane), aT2 = aT1 ^ 64u;           // transposed-read lane bases
  uint32_t aWP = aLN + oWP + (uint32_t)(wave * PARK) * 1024u;                   // parked W_l^T k-steps of this wave
  asm volatile("" : "+v"(aWP));

  // stationary W_l^T rows of this wave (row tile `wave` of the data-gradient product)
  u32x4 wreg[KSR];
#pragma unroll
  for (int s = 0; s < KSR; ++s) wreg[s] = a.wb[(wave * KS + s) * 64 + lane];
#pragma unroll
  for (int s = KSR; s < KS; ++s)
    *(lds_v4*)(uintptr_t)(aWP + (uint32_t)(s - KSR) * 1024u) = a.wb[(wave * KS + s) * 64 + lane];
  {   // sin/cos table of the phase bytes (decode: u / 256 + kPhaseEps revolutions), entry u at LDS byte 4 u
    const int tid = wave * 64 + lane;
    if (tid < 256) {
      const float r = __builtin_fmaf((float)tid, 1.0f / 256.0f, kPhaseEps);
      *(__attribute__((address_space(3))) uint32_t*)(uintptr_t)(4u * (uint32_t)tid) = OP::pack2(__builtin_amdgcn_sinf(r), __builtin_amdgcn_cosf(r));
    }
  }

  f32x16 acc[WJ][WI];
#pragma unroll
  for (int x = 0; x < WJ; ++x)
#pragma unroll
    for (int y = 0; y < WI; ++y) acc[x][y] = f32x16{};
  float dbs = 0.f;
  uint32_t ones_h2 = 0x3c003c00u;   // (1.0h, 1.0h)
  asm volatile("" : "+v"(ones_h2));

  const long pb_begin = blockIdx.x, pb_step = gridDim.x;
  const int nblk = (int)((a.n_pb - pb_begin + pb_step - 1) / pb_step);

  // LDS-DMA of this wave's piece of a block: deltas (tile `wave` of the 8 byte pieces) / phase bytes.  A request beyond the
  // last block re-reads the last one into a slot nobody reads any more: every step issues the same vector-memory
  // operations, so one counted vmcnt serves every barrier and the step has no conditional parts.
  auto stageD = [&](int k, uint32_t slot) {
    const int ks = k < nblk ? k : nblk - 1;
    glds16o(a.D + ((pb_begin + (long)ks * pb_step) * NT + wave) * 64, aLN, oRD + slot * 8192u + (uint32_t)wave * 1024u);
  };
  auto stageP = [&](int k, uint32_t slot) {
    const int ks = k < nblk ? k : nblk - 1;
    glds16o(a.P + ((pb_begin + (long)ks * pb_step) * NT + wave) * 64, aLN, oRP + slot * 8192u + (uint32_t)wave * 1024u);
  };
  auto slot_end = [&]() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  auto tr_pair = [&](uint32_t b1, uint32_t b2, int imm) -> u32x4 {
    const s16x4 r0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(uintptr_t)(b1 + imm));
    const s16x4 r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(uintptr_t)(b2 + imm));
    const u32x2 a0 = __builtin_bit_cast(u32x2, r0), a1 = __builtin_bit_cast(u32x2, r1);
    return u32x4{a0.x, a0.y, a1.x, a1.y};
  };

  // ---- one pipeline step: [DMA requests] C(kx+1) | X(kx) | W(kx-1) + E(kx), 32 slots ------------------------------
  u32x4 fa[2];
  u32x4 fb[2][WI];
  // i3 = kx mod 3, ip = kx mod NBP: carried by the caller (a modulo by a non-power of two is a dozen scalar instructions)
  auto step = [&](int kx, uint32_t i3, uint32_t ip) __attribute__((always_inline)) {
    constexpr bool do_c = true, do_x = true, do_w = true, dma_p = true, dma_d = true;   // (every step is the full step)
    const uint32_t i3n = i3 == 2u ? 0u : i3 + 1u, i3p = i3 == 0u ? 2u : i3 - 1u, i2 = (uint32_t)(kx & 1);
    const uint32_t ipp = ip == 0u ? (uint32_t)(NBP - 1) : ip - 1u;      // slot of block kx - 1 = slot of block kx + AP
    const uint32_t uX = oX16 + i3 * 16384u;                                           // X(kx): B operand
    const uint32_t uC = oX16 + i3n * 16384u + (uint32_t)wave * 2048u;                 // C(kx+1): destination (k-steps 2 wave, 2 wave + 1)
    const uint32_t uCs = oRD + i3n * 8192u + (uint32_t)wave * 1024u;                  // C(kx+1): byte piece
    const uint32_t uWd = oX16 + i3p * 16384u + (uint32_t)(wr * WJ) * 2048u;           // W(kx-1): delta^T tiles of this wave's row group
    const uint32_t uWs = oS16 + (i2 ^ 1u) * 16384u + (uint32_t)(wc * WI) * 2048u;     // W(kx-1): sine tiles of this wave's column group
    const uint32_t uEs = oS16 + i2 * 16384u + (uint32_t)wave * 2048u;                 // E(kx): sines of this wave's tile
    const uint32_t uEp = oRP + ip * 8192u + (uint32_t)wave * 1024u; // E(kx): phase bytes of this wave's tile
    // Per-step base registers = lane pattern + wave-uniform offset, formed where they are first needed and made opaque
    // (empty asm): a ds_* instruction takes one address register + a 16-bit immediate, and hipcc folds every constant it
    // can see into a 32-bit literal of a separate v_add per access otherwise (the LDS offsets here exceed 16 bits).
    auto base = [&](uint32_t lane_part, uint32_t uni) -> uint32_t {
      uint32_t b = lane_part + uni;
      asm volatile("" : "+v"(b));
      return b;
    };
    uint32_t bX0 = 0, bX1 = 0, bWa1 = 0, bWa2 = 0, bWb1 = 0, bWb2 = 0, bS = 0, bSx = 0;
    auto x_load = [&](int c) -> u32x4 { return *(lds_cv4*)(uintptr_t)(((c & 1) ? bX1 : bX0) + (uint32_t)c * 1024u); };
    auto wa_load = [&](int kk, int x) -> u32x4 { return tr_pair(bWa1, bWa2, x * 2048 + kk * 256); };
    auto wb_load = [&](int kk, u32x4* dst) {
#pragma unroll
      for (int y = 0; y < WI; ++y) dst[y] = tr_pair(bWb1, bWb2, y * 2048 + kk * 256);
    };
    u32x4 xb[NXB];
    u32x4 wt[PARK > 0 ? PARK : 1];
    u32x4 cv_raw = {0u, 0u, 0u, 0u}, pw = {0u, 0u, 0u, 0u}, ep_d = {0u, 0u, 0u, 0u};
    uint32_t tab[NTAB];
    uint32_t ep_sn = 0;
    float sv_even = 0.f;
    f32x16 g = f32x16{};
    if (do_x) {
      bX0 = base(aL1, uX); bX1 = base(aL1x, uX);
#pragma unroll
      for (int c = 0; c < PF; ++c) xb[c] = x_load(c);
    }
    slot_end();
    // phase-byte u of value t -> table address 4 u (the table starts at LDS address 0)
    auto lookup = [&](int t) -> uint32_t {
      const uint32_t w = pw[t >> 2];
      const uint32_t adr = (t & 3) == 3 ? (w >> 24) << 2 : (((w >> (8 * (t & 3))) & 0xffu) << 2);
      return *(lds_cu32*)(uintptr_t)adr;
    };
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      if (j == 0 && dma_p) stageP(kx + AP, ipp);
      if (j == 3 && dma_d) stageD(kx + AD, i3);
      if (j < KS) {
        if (do_x) {
          if (j + PF < KS) xb[(j + PF) % NXB] = x_load(j + PF);
          if (PARK > 0 && j + 3 >= KSR && j + 3 < KS) wt[j + 3 - KSR] = *(lds_cv4*)(uintptr_t)(aWP + (uint32_t)(j + 3 - KSR) * 1024u);
          if (do_c) {
            if (j == 1) cv_raw = *(lds_cv4*)(uintptr_t)base(aLN, uCs);
            if (j == 5) *(lds_v4*)(uintptr_t)base(aL1, uC) = fp8x8_to_f16(cv_raw.x, cv_raw.y);
            if (j == 8) *(lds_v4*)(uintptr_t)base(aL1x, uC + 1024u) = fp8x8_to_f16(cv_raw.z, cv_raw.w);
          }
          if (j == KS - LD - 3) pw = *(lds_cv4*)(uintptr_t)base(aLN, uEp);
          g = OP::mfma(j < KSR ? wreg[j < KSR ? j : 0] : wt[j >= KSR ? j - KSR : 0], xb[j % NXB], g);
        }
        if (do_w) {
          if (j == KS - 4) { bWb1 = base(aT1, uWs); bWb2 = base(aT2, uWs); wb_load(0, fb[0]); }
          if (j == KS - 2) { bWa1 = base(aT1, uWd); bWa2 = base(aT2, uWd); fa[0] = wa_load(0, 0); }
        }
      } else {
        const int i = (j - KS) >> 1, y = (j - KS) & 1, kk = i >> 2, x = i & 3;
        if (do_w) {
          if (y == 0 && i + 1 < 8) fa[(i + 1) & 1] = wa_load((i + 1) >> 2, (i + 1) & 3);
          if (i == 1 && y == 1) wb_load(1, fb[1]);
          acc[x][y] = OP::mfma(fa[i & 1], fb[kk][y], acc[x][y]);
          if (x == DBX) {   // bias gradient: row sums of delta^T (two v_dot2_f32_f16 against (1, 1) per slot)
            const h2 one2 = __builtin_bit_cast(h2, ones_h2);
            const uint32_t f0 = y == 0 ? fa[i & 1].x : fa[i & 1].z, f1 = y == 0 ? fa[i & 1].y : fa[i & 1].w;
            dbs = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, f0), one2, dbs, false);
            dbs = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, f1), one2, dbs, false);
          }
        }
      }
      if (do_x) {
        // E(kx): table pair of value t is looked up LD slots before the slot that forms the value
        const int tl = j - (KS - LD);
        if (tl >= 0 && tl < 16) tab[tl % NTAB] = lookup(tl);
        const int t = j - KS;
        if (t >= 0) {
          const uint32_t e = tab[t % NTAB];
          float sv = __builtin_fmaf((float)__builtin_bit_cast(h2, e)[1], g[t], 0.0f);      // cos * (W^T delta): v_fma_mix_f32
          if (t & 1) {
            const uint32_t e0 = tab[(t - 1) % NTAB];
            // (saturation at +-448 by MODE.FP16_OVFL)
            int w = (int)ep_d[t >> 2];
            w = (t & 2) ? __builtin_amdgcn_cvt_pk_fp8_f32(sv_even, sv, w, true) : __builtin_amdgcn_cvt_pk_fp8_f32(sv_even, sv, w, false);   // (low half first: what stays in the high half is overwritten next)
            ep_d[t >> 2] = (uint32_t)w;
            const uint32_t sn = __builtin_amdgcn_perm(e, e0, 0x05040100u);      // (sin t-1, sin t)
            if ((t & 3) == 1) ep_sn = sn;
            else {
              if (t == 3) bS = base(aL1, uEs);
              if (t == 11) bSx = base(aL1x, uEs + 1024u);
              *(lds_v2*)(uintptr_t)(((t >> 3) ? bSx : bS) + (uint32_t)(t & 4) * 2u) = u32x2{ep_sn, sn};
            }
          } else {
            asm volatile("" : "+v"(sv));   // formed in its own slot
            sv_even = sv;
          }
          if (t == 15) {
#ifndef SF_EXPERIMENT_NO_STORE
            a.Dout[((pb_begin + (long)(kx < nblk ? kx : nblk - 1) * pb_step) * NT + wave) * 64 + lane] = ep_d;
#else
            asm volatile("" ::"v"(ep_d));
#endif
          }
        }
      }
      slot_end();
    }
  };

  if (nblk > 0) {
    // Ring protocol (as k_bwd8, with the byte ring feeding only the conversion).  Step k requests P(k + AP) then D(k + AD);
    // the barrier at the top of step k needs D(k + 1) (expanded in step k) and P(k) landed.  D(k + 1) is the second request
    // of step k - 2; younger in this wave's in-order vmcnt queue: the store of step k - 2 and the three operations of step
    // k - 1 (requests beyond the last block are issued all the same, see stageD).
    // EVERY step is the full step - there is no peeled first or last block (peeled copies cost hipcc ~80 spills each):
    //   step 0 runs W(-1) on zero-filled images (adds exact zeros to dW and db);
    //   step nblk runs W(nblk - 1) beside X/E of a duplicate of the last block, whose deltas are stored a second time to
    //   the same place (same bytes).
    for (int k = 0; k < AP; ++k) stageP(k, (uint32_t)k);
    for (int k = 0; k < AD; ++k) stageD(k, (uint32_t)k);
    {   // images of "block -1": X16 buffer 2, S16 buffer 1
      const u32x4 z = {0u, 0u, 0u, 0u};
      const uint32_t t16 = (uint32_t)(wave * 64 + lane) * 16u;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        *(lds_v4*)(uintptr_t)(oX16 + 2u * 16384u + (uint32_t)i * 8192u + t16) = z;
        *(lds_v4*)(uintptr_t)(oS16 + 16384u + (uint32_t)i * 8192u + t16) = z;
      }
    }
    bar_all();
    {   // C(0)
      const u32x4 raw = *(lds_cv4*)(uintptr_t)(aLN + oRD + (uint32_t)wave * 1024u);
      *(lds_v4*)(uintptr_t)(aL1 + oX16 + (uint32_t)wave * 2048u) = fp8x8_to_f16(raw.x, raw.y);
      *(lds_v4*)(uintptr_t)(aL1x + oX16 + (uint32_t)wave * 2048u + 1024u) = fp8x8_to_f16(raw.z, raw.w);
    }
    uint32_t i3 = 0, ip = 0;
    for (int k = 0; k <= nblk; ++k) {
      if (k >= 2) bar_dma<4>(); else bar_all();
      step(k, i3, ip);
      i3 = i3 == 2u ? 0u : i3 + 1u;
      ip = ip == (uint32_t)(NBP - 1) ? 0u : ip + 1u;
    }
    bar_all();     // no LDS-DMA of this workgroup is in flight when its LDS is handed on
  }
  float* slab = a.slab + (size_t)blockIdx.x * (256 * 256 + 256);
  const int cl = lane & 31, hh = lane >> 5;
#pragma unroll
  for (int x = 0; x < WJ; ++x)
#pragma unroll
    for (int y = 0; y < WI; ++y)
#pragma unroll
      for (int t = 0; t < 16; ++t)
        slab[(size_t)(32 * (wr * WJ + x) + rho(t, hh)) * 256 + 32 * (wc * WI + y) + cl] = acc[x][y][t];
  {
    const float tsum = dbs + __shfl_xor(dbs, 32);
    if (hh == 0) slab[256 * 256 + 32 * (wr * WJ + DBX) + cl] = tsum;
  }
}

template <int PARK, int NBP>
__global__ __launch_bounds__(512) void k_bwd8h(Bwd8Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // the kernel addresses LDS absolutely (the table at address 0 is what makes a table address one instruction)
  if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem != 0u) __builtin_trap();
  // MODE.FP16_OVFL = 1: fp8 conversions saturate at +-448 instead of returning NaN (epi_probe.hip)
  __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#ifdef SF_BWD8H_ONEBODY
  bwd8h_body<PARK, NBP, 0>(a, wave, lane);
  return;
#endif
  switch (wave & 3) {
    case 0: bwd8h_body<PARK, NBP, 0>(a, wave, lane); break;
    case 1: bwd8h_body<PARK, NBP, 1>(a, wave, lane); break;
    case 2: bwd8h_body<PARK, NBP, 2>(a, wave, lane); break;
    default: bwd8h_body<PARK, NBP, 3>(a, wave, lane); break;
  }
}

}  // namespace sf
