// siren_s8h.hip — k_bwd8h: the backward of one 256 x 256 HIDDEN layer from 8-bit scratch (phase bytes + fp8 deltas,
// sf_config.scratch_format = 8) as a hand-cut software pipeline, one "slot" per MFMA (round 3).
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
//   * NO LDS STORE THAT IS NOT FORCED.  Measured on this kernel (timing-only builds, profiles/r03_*): an LDS store of
//     1 KiB costs ~30 cycles of the LDS pipe, a 1 KiB read 4.  The 16-bit image of the deltas that both products read
//     in the first form of this kernel (one conversion pass per block: 16 KiB of ds_write_b128) cost 13 % of the kernel.
//     Both products now read the fp8 BYTES the LDS-DMA left in the ring and convert in registers (v_cvt_scalef32_pk_f16_fp8,
//     exact): phase X one ds_read_b128 per two k-steps + 4 conversions per slot, phase W one ds_read_b64_tr_b8 per
//     delta^T fragment + 4 conversions (scripts/probes/trb8.hip has the lane map; rows come out permuted by nu8, undone
//     when the slab is written).  96 more conversions per wave and block for 16 KiB fewer stored and 16 KiB fewer read.
//     The only LDS stores left are the sines (the weight-gradient product contracts over pixels: one transposition
//     through LDS is forced), 16 bytes per lane and k-step.
//   * the bias-gradient row sums are taken by ONE of the four column waves that hold the same fragments; which one is a
//     compile-time constant of the wave's loop (the loop is instantiated per wave), not a select per chunk.
//   * every LDS read is issued a fixed number of slots before its first use, the epilogue of a block is cut into sixteen
//     single values, one per slot of a phase W, and nothing crosses a slot boundary (sched_barrier + compiler memory
//     barrier, as in k_fwd_pipe).
//   * THE TWO WAVES OF A SIMD RUN OPPOSITE PHASES (see the step loop).
//
// LDS of one workgroup (8 waves; PARK = 3: 145 KiB):
//   T     1 KiB          sin/cos pairs of the 256 phase bytes (two 16-bit floats per entry), at address 0
//   RD    5 x  8 KiB     fp8 delta pieces: block k+3 is requested in step k; read by X(k) and, two steps later, by W(k)
//   RP    4 x  8 KiB     phase-byte pieces: block k+3 is requested in step k, decoded at the end of X(k)
//   S16   3 x 16 KiB     fp16 sin(phase): written during X(k) (the sines need the phase bytes only), read transposed by W(k)
//   WP    8 x PARK KiB   the last PARK k-steps of every wave's stationary W_l^T rows (the rest lives in registers)
// (included by siren_fit.hip after siren_s8.hip)

namespace sf {

#ifndef SF_BWD8H_LD
#define SF_BWD8H_LD 2      // slots between a table lookup and the value that uses it
#endif

template <int PARK>
constexpr size_t bwd8h_lds_bytes() { return (size_t)(1 + 5 * 8 + 4 * 8 + 3 * 16 + 8 * PARK) * 1024; }

// DBX: the row tile (0..3) whose bias-gradient sums this instantiation takes (= the wave's column index wc)
// ROLE: 0 = waves 0-3 ("N": X | W + E), 1 = waves 4-7 ("S", their SIMD partners: W + E | X)
template <int PARK, int DBX, int ROLE>
__device__ __forceinline__ void bwd8h_body(const Bwd8Args& a, const int wave, const int lane) {
  typedef OpF16 OP;
  constexpr int KS = 16, NT = 8, NBD = 5, NBP = 4, WJ = 4, WI = 2;
  constexpr int AD = 3, AP = 3;                  // blocks requested ahead (deltas / phases)
  constexpr int LD = SF_BWD8H_LD, NTAB = 2 * (LD + 1);   // table ring of phase X: LD + 1 pairs in flight
  constexpr int KSR = KS - PARK;
  constexpr uint32_t oRD = 1024, oRP = oRD + NBD * 8192, oS16 = oRP + NBP * 8192, oWP = oS16 + 3 * 16384;
  static_assert(LD >= 1 && LD <= 4 && PARK >= 0 && PARK <= 4, "slot plan");
  typedef __attribute__((address_space(3))) const u32x4 lds_cv4;
  typedef __attribute__((address_space(3))) u32x4 lds_v4;
  typedef __attribute__((address_space(3))) const uint32_t lds_cu32;
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  typedef __attribute__((ext_vector_type(2))) _Float16 h2;
  typedef __attribute__((ext_vector_type(2))) int i32x2;

  const int wr = wave >> 2, wc = wave & 3;
  // lane patterns of the LDS accesses (everything else is a wave-uniform offset or a 16-bit immediate)
  const uint32_t aLN = (uint32_t)lane * 16u;                                   // lane-linear piece element
  // byte pieces of the delta ring: slot i of a piece holds the element of lane i ^ 8 (i >> 5) - the upper lane half is
  // shifted by eight pixels, i.e. by 32 banks, so that the even and odd row suppliers of a ds_read_b64_tr_b8 group do not
  // meet on the same banks (as k_bwd8); the lane-linear 16-byte reads of phase X stay conflict-free under it
  const uint32_t aL8 = (uint32_t)(lane ^ ((lane >> 5) << 3)) * 16u;
  const uint32_t aL1 = (uint32_t)sw_lane(lane, 0) * 16u, aL1x = aL1 ^ 128u;     // swizzled 16-bit piece element, even / odd k-step
  const uint32_t aT1 = (uint32_t)tr_lane_base(lane), aT2 = aT1 ^ 64u;           // transposed 16-bit reads (sines)
  // transposed byte reads: ds_read_b64_tr_b8 works on groups of 16 lanes; lane t of a group supplies the address of an
  // 8-byte row, result lane i < 8 receives byte i of the rows of lanes 0, 2, .., 14 and lane 8 + i byte i of the rows of
  // lanes 1, 3, .., 15.  With lane t pointing at bytes 8q .. 8q+7 of piece lane (h' = t & 1, pixel 8 hq + (t >> 1)) -
  // q = group & 1, hq = group >> 1 - the wave receives an A fragment of the 16-pixel k-step whose row r = lane & 31 is
  // neuron nu8(r) of the tile and whose elements are the pixels 8 hq + 0..7 in order.
  const uint32_t aT8 = 16u * (32u * (uint32_t)(lane & 1) + ((8u * (uint32_t)(lane >> 5) + (uint32_t)((lane & 15) >> 1)) ^ (8u * (uint32_t)(lane & 1)))) +
                       8u * (uint32_t)((lane >> 4) & 1);
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
  // last block reads 1 KiB of zeros (deltas) or the last block again (phase bytes): every step issues the same
  // vector-memory operations, so one counted vmcnt serves every barrier and the step has no conditional parts.
  // (the global addresses are carried from step to step - one scalar add and one select per request, no multiplies)
  const long blk_stride = pb_step * NT * 64;                       // u32x4 elements from a block of this workgroup to its next
  const u32x4* const pD0 = a.D + (pb_begin * NT + wave) * 64;
  const u32x4* const pP0 = a.P + (pb_begin * NT + wave) * 64;
  u32x4* const pO0 = a.Dout + (pb_begin * NT + wave) * 64;
  auto stageD = [&](const u32x4* src, uint32_t off) { glds16o(src, aL8, oRD + off + (uint32_t)wave * 1024u); };        // off = ring slot * 8 KiB
  auto stageP = [&](const u32x4* src, uint32_t off) { glds16o(src, aLN, oRP + off + (uint32_t)wave * 1024u); };
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

#ifdef SF_EXPERIMENT_STAMP
  unsigned long long st_bar = 0, st_x = 0, st_w = 0, st_n = 0;
#endif
#ifdef SF_EXPERIMENT_STAMP2
  unsigned st2[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) st2[i] = 0u;
  unsigned st2_n = 0;
#endif
  // ---- the two phases of a block, 16 slots each -----------------------------------------------------------------------
  //   phase X(k)            16 MFMAs of the data-gradient product (one dependent chain); B operand = the bytes of D(k),
  //                         converted in registers one slot ahead
  //   phase W(kw) + E(ke)   16 MFMAs of the weight-gradient product of block kw beside the sixteen epilogue values of block
  //                         ke (cos * g -> fp8, sines -> S16(ke)); ke is the block whose phase X this wave ran last
  // THE TWO WAVES OF A SIMD RUN OPPOSITE PHASES (a stagger of half a block).  Waves 0-3 ("N") run X(k) | W(k-1)+E(k) in the
  // two halves of step k; their SIMD partners, waves 4-7 ("S"), run W(k-2)+E(k-1) | X(k).  Run in step, both waves of a
  // SIMD were in the LDS-bound chain of phase X (all eight waves stream the same deltas) and then both in the VALU-heavy
  // phase W: the younger wave took 1 550-1 900 cycles for a phase X whose MFMAs need 1 024 per SIMD while the older one
  // waited 800-1 000 cycles at the barrier.  Staggered, a phase X always runs beside a phase W.  Price: a second workgroup
  // barrier per block (each half is closed by one), and E(k) of the S waves crosses a barrier (g, the phase bytes and the
  // first table pairs stay in registers).
  //   who needs what, with h = 2k (first half of step k) and h = 2k + 1 (second half):
  //     D(k)      read by X(k) at h = 2k (N), 2k+1 (S) and by W(k) at h = 2k+3 (N), 2k+4 (S); its ring slot is requested
  //               again at h = 2k+5 (D(k+5), second half of step k+2): five slots
  //     P(k)      read at the end of X(k): h = 2k (N), 2k+1 (S); its slot is requested again at h = 2k+2: four slots
  //     S16(k)    written at h = 2k+1 (N waves' tiles) and 2k+2 (S waves' tiles) - read at h = 2k+3 (N), 2k+4 (S); rewritten
  //               at h = 2k+5: two buffers
  u32x2 fa8[2];                      // delta^T fragments as bytes (ring of two)
  u32x4 fa;                          // the converted fragment of the current chunk
  u32x4 fb[2][WI];
  u32x4 raw[2];                      // byte pieces of phase X (two k-steps each), ring of two
  f32x16 g = f32x16{};
  uint32_t cp[4] = {0u, 0u, 0u, 0u};  // cosines of values 0..7 of the block whose phase X ran last, as 16-bit pairs
  u32x2 pwh = {0u, 0u};               // phase bytes 8..15 of that block (their table pairs are looked up by the epilogue)
  // Per-step base registers = lane pattern + wave-uniform offset, formed where they are first needed and made opaque
  // (empty asm): a ds_* instruction takes one address register + a 16-bit immediate, and hipcc folds every constant it
  // can see into a 32-bit literal of a separate v_add per access otherwise (the LDS offsets here exceed 16 bits).
  auto base = [&](uint32_t lane_part, uint32_t uni) -> uint32_t {
    uint32_t b = lane_part + uni;
    asm volatile("" : "+v"(b));
    return b;
  };
  const uint32_t two_s = __builtin_amdgcn_readfirstlane(2u);   // (the shift count of the SDWA form, in a scalar register)
  // phase-byte u of value t -> table address 4 u (the table starts at LDS address 0)
  auto lookup = [&](uint32_t w, int t) -> uint32_t {   // w = the dword that holds phase byte t
    uint32_t adr;
    if ((t & 3) == 0) {   // byte 0: hipcc emits v_lshlrev + v_and for (w & 0xff) << 2; the SDWA form is one instruction
      asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(adr) : "s"(two_s), "v"(w));
    } else {
      adr = (t & 3) == 3 ? (w >> 24) << 2 : (((w >> (8 * (t & 3))) & 0xffu) << 2);
    }
#ifdef SF_EXP_NOLOOKUP     // timing-only: no table read
    return adr | 0x3c000000u;
#endif
    return *(lds_cu32*)(uintptr_t)adr;
  };
  struct WOps { uint32_t a, b1, b2; };      // base registers of the operand fragments of a phase W
  auto wa_load = [&](const WOps& o, int kk, int x) -> u32x2 {
    const i32x2 r = __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) i32x2*)(uintptr_t)(o.a + (uint32_t)(x * 1024 + kk * 256)));
    return u32x2{(uint32_t)r.x, (uint32_t)r.y};
  };
  auto wb_load = [&](const WOps& o, int kk, u32x4* dst) {
#pragma unroll
    for (int y = 0; y < WI; ++y) dst[y] = tr_pair(o.b1, o.b2, y * 2048 + kk * 256);
  };
  // byte offsets of the operand images of W(b): delta tiles of this wave's row group in ring slot dslot, sine tiles of its
  // column group in S16 buffer sbuf
  auto w_uni_d = [&](uint32_t doff) -> uint32_t { return oRD + doff + (uint32_t)(wr * WJ) * 1024u; };
  auto w_uni_s = [&](uint32_t soff) -> uint32_t { return oS16 + soff + (uint32_t)(wc * WI) * 2048u; };
  // the first two byte pieces of X(b) (requested at the end of the phase before: D(b) has landed by then, see the loop)
  auto x_pre = [&](uint32_t doff) {
    const uint32_t bX = base(aL8, oRD + doff);
    raw[0] = *(lds_cv4*)(uintptr_t)bX;
    raw[1] = *(lds_cv4*)(uintptr_t)(bX + 1024u);
  };

  // X(k): dslot = ring slot of D(k).  dma: 0 none, 1 = a delta request, 2 = a phase-byte request (source dsrc, ring slot
  // rslot).  pslot: ring slot of the phase bytes of block k (read for the epilogue that follows).  pre: the operand
  // fragments of the NEXT phase W of this wave are requested at the end (S waves: both images are complete; N waves: only
  // the delta bytes are).
  auto phaseX = [&](uint32_t dslot, int dma, const u32x4* dsrc, uint32_t rslot, uint32_t pslot, uint32_t sbuf, uint32_t wdslot, uint32_t wsbuf, WOps& wo) __attribute__((always_inline)) {
    const uint32_t uEp = oRP + pslot + (uint32_t)wave * 1024u;
    const uint32_t uEs = oS16 + sbuf + (uint32_t)wave * 2048u;
    const uint32_t bX = base(aL8, oRD + dslot);
    u32x4 wt[PARK > 0 ? PARK : 1];
    u32x4 bq[2];                                   // converted B operands: slot j uses bq[j & 1]
    u32x4 pw;                                      // the sixteen phase bytes of this lane
    uint32_t tab[NTAB];                            // table pairs in flight
    uint32_t sp[4];                                // sines of a k-step, as 16-bit pairs
#ifdef SF_BWD8H_XPRIO
    __builtin_amdgcn_s_setprio(SF_BWD8H_XPRIO);
#endif
    bq[0] = fp8x8_to_f16(raw[0].x, raw[0].y);
    pw = *(lds_cv4*)(uintptr_t)base(aLN, uEp);
    slot_end();
#ifdef SF_EXPERIMENT_STAMP2
    unsigned st2_t = (unsigned)__builtin_amdgcn_s_memtime();
#endif
#pragma unroll
    for (int j = 0; j < KS; ++j) {
      if (j == 2 && dma == 1) stageD(dsrc, rslot);
      if (j == 2 && dma == 2) stageP(dsrc, rslot);
      // byte piece of k-steps j + 3, j + 4 (tile (j + 3) / 2), into the ring place its predecessor leaves in this slot
      if ((j & 1) == 1 && j + 3 < KS) raw[((j + 3) >> 1) & 1] = *(lds_cv4*)(uintptr_t)(bX + (uint32_t)((j + 3) >> 1) * 1024u);
      if (PARK > 0 && j + 2 >= KSR && j + 2 < KS) wt[j + 2 - KSR] = *(lds_cv4*)(uintptr_t)(aWP + (uint32_t)(j + 2 - KSR) * 1024u);
      if (j + 1 < KS) {   // the B operand of the NEXT slot: four conversions, ahead of this slot's MFMA (a conversion right in front
                          // of the MFMA that reads it costs two wait states: s_nop 1 in every slot)
        const u32x4 r = raw[((j + 1) >> 1) & 1];
        bq[(j + 1) & 1] = ((j + 1) & 1) ? fp8x8_to_f16(r.z, r.w) : fp8x8_to_f16(r.x, r.y);
      }
      __builtin_amdgcn_sched_barrier(0);
      g = OP::mfma(j < KSR ? wreg[j < KSR ? j : 0] : wt[j >= KSR ? j - KSR : 0], bq[j & 1], j == 0 ? f32x16{} : g);
      // The phase bytes are all the sines and cosines need, so HALF of that work is done here (values 0..7: the first
      // k-step of this wave's tile) and half in the epilogue (values 8..15): what bounds a phase is its count of vector
      // and LDS instructions per MFMA slot - about six per slot are hidden beside the MFMAs of two waves, every further one
      // costs ~5 cycles (timing-only builds, profiles/r03_*) - and this split levels the two phases.
      // Values 2i, 2i + 1 are looked up in slot i + 1 (i = 0..3) and split LD slots later: the sine pair goes into the pair
      // registers of the k-step (one 16-byte write), the cosine pair into cp[i] for the epilogue.
      if (j >= 1 && j <= 4) { tab[(2 * (j - 1)) % NTAB] = lookup(pw[(j - 1) >> 1], 2 * (j - 1)); tab[(2 * (j - 1) + 1) % NTAB] = lookup(pw[(j - 1) >> 1], 2 * (j - 1) + 1); }
      {
        const int i = j - 1 - LD;          // pair split in this slot
        if (i >= 0 && i < 4) {
          const uint32_t e0 = tab[(2 * i) % NTAB], e1 = tab[(2 * i + 1) % NTAB];
          uint32_t sn = __builtin_amdgcn_perm(e1, e0, 0x05040100u);      // (sin 2i, sin 2i+1)
          uint32_t cs = __builtin_amdgcn_perm(e1, e0, 0x07060302u);      // (cos 2i, cos 2i+1)
          asm volatile("" : "+v"(sn), "+v"(cs));
          cp[i] = cs;
          sp[i] = sn;
#ifndef SF_EXP_NOSINW
          if (i == 3) *(lds_v4*)(uintptr_t)base(aL1, uEs) = u32x4{sp[0], sp[1], sp[2], sp[3]};
#endif
        }
      }
      if (j == 5) pwh = u32x2{pw.z, pw.w};
      // (S waves only: the N waves' next phase W reads sines the S waves are still writing in this half)
#ifdef SF_EXP_NPRE   // timing-only (races with the S waves' sine writes)
      if (j == KS - 4)
#else
      if (ROLE == 1 && j == KS - 4)
#endif
      { wo.b1 = base(aT1, w_uni_s(wsbuf)); wo.b2 = base(aT2, w_uni_s(wsbuf)); wb_load(wo, 0, fb[0]); }
      if (j == KS - 2) { wo.a = base(aT8, w_uni_d(wdslot)); fa8[0] = wa_load(wo, 0, 0); fa8[1] = wa_load(wo, 0, 1); }
      slot_end();
#ifdef SF_EXPERIMENT_STAMP2
      { const unsigned tn = (unsigned)__builtin_amdgcn_s_memtime(); st2[j] += tn - st2_t; st2_t = tn; }
#endif
    }
  };
  // W(kw) + E(ke): odst = where this wave's piece of the deltas of block ke goes (the dump for blocks outside the chunk).
  // xslot: ring slot of the block whose phase X this wave runs next (its first byte pieces are requested at the end).
  auto phaseW = [&](u32x4* odst, uint32_t sbuf, uint32_t wsbuf, int dma, const u32x4* dsrc, uint32_t rslot, uint32_t xslot, WOps& wo) __attribute__((always_inline)) {
    const uint32_t uEs = oS16 + sbuf + (uint32_t)wave * 2048u;
    float sv_even = 0.f;
    u32x4 ep_d;                        // outgoing fp8 deltas (every byte is written before the store: no initial value)
    asm volatile("" : "=v"(ep_d));
    uint32_t tab[LD + 2];              // table pairs of values 8..15 in flight (the pair of value t-1 is still read in slot t)
    uint32_t sp[3] = {0u, 0u, 0u};
#ifdef SF_EXPERIMENT_STAMP2
    unsigned st2_t = (unsigned)__builtin_amdgcn_s_memtime();
#endif
#ifdef SF_BWD8H_WPRIO
    __builtin_amdgcn_s_setprio(SF_BWD8H_WPRIO);
#endif
#ifndef SF_EXP_NPRE
    if (ROLE == 0) { wo.b1 = base(aT1, w_uni_s(wsbuf)); wo.b2 = base(aT2, w_uni_s(wsbuf)); wb_load(wo, 0, fb[0]); }   // (N waves: S16(kw) was completed in the half before)
#endif
    fa = fp8x8_to_f16(fa8[0].x, fa8[0].y);
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int i = t >> 1, y = t & 1, kk = i >> 2, x = i & 3;
      if (t == 2 && dma == 1) stageD(dsrc, rslot);
      if (t == 2 && dma == 2) stageP(dsrc, rslot);
      if (y == 0 && i + 2 < 8) fa8[i & 1] = wa_load(wo, (i + 2) >> 2, (i + 2) & 3);   // (the bytes of chunk i + 2: three slots before their conversion)
      if (i == 1 && y == 1) wb_load(wo, 1, fb[1]);
      acc[x][y] = OP::mfma(fa, fb[kk][y], acc[x][y]);
#ifndef SF_EXP_NODOT
      if (x == DBX)
#else
      if (false)
#endif
      {   // bias gradient: row sums of delta^T (two v_dot2_f32_f16 against (1, 1) per slot)
        const h2 one2 = __builtin_bit_cast(h2, ones_h2);
        const uint32_t f0 = y == 0 ? fa.x : fa.z, f1 = y == 0 ? fa.y : fa.w;
        dbs = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, f0), one2, dbs, false);
        dbs = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, f1), one2, dbs, false);
      }
      if (y == 1 && i + 1 < 8) {   // the fragment of the next chunk (bytes requested three slots ago), behind this chunk's last MFMA
        u32x4 nf = fp8x8_to_f16(fa8[(i + 1) & 1].x, fa8[(i + 1) & 1].y);
        asm volatile("" : "+v"(nf));
        fa = nf;
      }
      if (t == 13) x_pre(xslot);
      // E: cos * (W^T delta) -> fp8 (saturation at +-448 by MODE.FP16_OVFL): v_fma_mix_f32 + half a v_cvt_pk_fp8_f32 per
      // value.  Values 0..7 take their cosines from cp[] (phase X); values 8..15 are looked up here, LD slots ahead, and
      // their sines leave in the 16-byte write of the tile's second k-step.
      if (t + LD >= 8 && t + LD < 16) tab[(t + LD) % (LD + 2)] = lookup((t + LD) < 12 ? pwh.x : pwh.y, t + LD);
      {
        const uint32_t e = t < 8 ? cp[t >> 1] : tab[t % (LD + 2)];
        float sv = __builtin_fmaf((float)__builtin_bit_cast(h2, e)[t < 8 ? (t & 1) : 1], g[t], 0.0f);
        if (t & 1) {
          int w = (int)ep_d[t >> 2];
          w = (t & 2) ? __builtin_amdgcn_cvt_pk_fp8_f32(sv_even, sv, w, true) : __builtin_amdgcn_cvt_pk_fp8_f32(sv_even, sv, w, false);   // (low half first: what stays in the high half is overwritten next)
          ep_d[t >> 2] = (uint32_t)w;
          if (t >= 9) {
            const uint32_t sn = __builtin_amdgcn_perm(e, tab[(t - 1) % (LD + 2)], 0x05040100u);      // (sin t-1, sin t)
            if (t < 15) sp[(t - 9) >> 1] = sn;
#ifndef SF_EXP_NOSINW
            else *(lds_v4*)(uintptr_t)base(aL1x, uEs + 1024u) = u32x4{sp[0], sp[1], sp[2], sn};
#endif
          }
        } else {
          asm volatile("" : "+v"(sv));   // formed in its own slot
          sv_even = sv;
        }
        if (t == 15) {
#ifndef SF_EXPERIMENT_NO_STORE
          odst[lane] = ep_d;
#else
          asm volatile("" ::"v"(ep_d));
#endif
        }
      }
      slot_end();
#ifdef SF_EXPERIMENT_STAMP2
      { const unsigned tn = (unsigned)__builtin_amdgcn_s_memtime(); st2[16 + t] += tn - st2_t; st2_t = tn; }
#endif
    }
  };

  if (nblk > 0) {
    // Ring protocol.  Step k (two halves, a barrier in front of each) requests P(k + 3) in its first half and D(k + 3) in
    // its second; requests beyond the last block read zeros (deltas: the block then adds exact zeros to dW and db, and its
    // outgoing deltas go to the dump) or the last block again (phase bytes: only their finiteness matters).  Every step is
    // the full step: steps 0 and 1 run their W phases on zero-filled images, the last two on zero deltas.
    // vmcnt: a wave's vector-memory operations per step, in issue order - N: P | D, store;  S: P, store | D.
    //   N waves request the first byte pieces of X(k + 1) at the end of the SECOND half of step k: D(k + 1), requested in
    //   the second half of step k - 2, must have landed at the barrier in front of that half; younger are store(k-2), the
    //   three operations of step k - 1 and P of step k => vmcnt(5).  P(k + 1), decoded at the end of X(k + 1), is older.
    //   S waves request the first byte pieces of X(k) at the end of the FIRST half of step k: D(k), requested in the second
    //   half of step k - 3; younger are P, store, D of steps k - 2 and k - 1 => vmcnt(6) at the first barrier.
    for (int k = 0; k < AP; ++k) stageP(pP0 + (k < nblk ? k : nblk - 1) * blk_stride, (uint32_t)k * 8192u);
    for (int k = 0; k < AD; ++k) stageD(k < nblk ? pD0 + k * blk_stride : a.zeros, (uint32_t)k * 8192u);
    {   // zero images for the W phases of steps 0 and 1: ring slots 3 and 4 (blocks -2 and -1), the S16 buffers
      const u32x4 z = {0u, 0u, 0u, 0u};
      const uint32_t t16 = (uint32_t)(wave * 64 + lane) * 16u;
#pragma unroll
      for (int i = 0; i < 2; ++i) *(lds_v4*)(uintptr_t)(oRD + 3u * 8192u + (uint32_t)i * 8192u + t16) = z;
#pragma unroll
      for (int i = 0; i < 6; ++i) *(lds_v4*)(uintptr_t)(oS16 + (uint32_t)i * 8192u + t16) = z;
    }
    bar_all();
    WOps wo = {0u, 0u, 0u};
    if (ROLE == 0) x_pre(0u);            // X(0)
    else {                               // the S waves' first W phase, W(-2): zero images
      wo.b1 = base(aT1, w_uni_s(16384u)); wo.b2 = base(aT2, w_uni_s(16384u)); wb_load(wo, 0, fb[0]);
      wo.a = base(aT8, w_uni_d(3u * 8192u)); fa8[0] = wa_load(wo, 0, 0); fa8[1] = wa_load(wo, 0, 1);
    }
    // Ring positions as BYTE OFFSETS in rotating scalar registers (a modulo or a compare-and-select per neighbour slot
    // was twenty scalar instructions per step, and this loop pays for every instruction it issues: DESIGN.md section 4a):
    // delta ring (5 slots of 8 KiB): blocks k-2 .. k+2;  sine buffers (3 of 16 KiB): blocks k-1, k, k+1
    uint32_t d_m2 = 3u * 8192u, d_m1 = 4u * 8192u, d_0 = 0u, d_p1 = 8192u, d_p2 = 2u * 8192u;
    uint32_t s_m1 = 2u * 16384u, s_0 = 0u, s_p1 = 16384u;
    uint32_t pc = 0u;                                                              // (k mod 4) * 8 KiB: phase ring slot of block k
    const u32x4* pD = pD0 + AD * blk_stride;                                       // D(k + 3)
    const u32x4* pP = pP0 + (AP < nblk ? AP : nblk - 1) * blk_stride;              // P(min(k + 3, nblk - 1))
    u32x4* pO = ROLE == 0 ? pO0 : pO0 - blk_stride;                                // deltas of block k (N waves) / k - 1 (S waves)
    u32x4* const dump = a.dump + wave * 64;
    // (the counted waits hold from step 0: what steps 0 to 2 read was requested above and has landed behind bar_all)
    for (int k = 0; k <= nblk + 1; ++k) {
      const uint32_t pm1 = (pc + 3u * 8192u) & 0x6000u;                  // phase ring slot of block k - 1 = of block k + 3
      const u32x4* const dsrc = k + AD < nblk ? pD : a.zeros;
      // E(k) of the N waves, E(k - 1) of the S waves: blocks outside the chunk go to the dump
      u32x4* const odst = (unsigned)(ROLE == 0 ? k : k - 1) < (unsigned)nblk ? pO : dump;
#ifdef SF_EXPERIMENT_STAMP
      const unsigned long long t_b0 = __builtin_amdgcn_s_memtime();
#endif
      if (ROLE == 1) bar_dma<6>(); else bar_lds();
#ifdef SF_EXPERIMENT_STAMP
      const unsigned long long t_b1 = __builtin_amdgcn_s_memtime();
#endif
      if (ROLE == 0) phaseX(d_0, 2, pP, pm1, pc, s_0, d_m1, s_m1, wo);                    // X(k) + sines(k); P(k+3); operands of W(k-1) requested
      else phaseW(odst, s_m1, 0u, 2, pP, pm1, d_0, wo);                                   // W(k-2) + E(k-1); P(k+3); bytes of X(k) requested
#ifdef SF_EXPERIMENT_STAMP
      const unsigned long long t_b2 = __builtin_amdgcn_s_memtime();
#endif
      if (ROLE == 0) bar_dma<5>(); else bar_lds();
#ifdef SF_EXPERIMENT_STAMP
      const unsigned long long t_b3 = __builtin_amdgcn_s_memtime();
#endif
      if (ROLE == 0) phaseW(odst, s_0, s_m1, 1, dsrc, d_m2, d_p1, wo);                    // W(k-1) + E(k); D(k+3); bytes of X(k+1) requested
      else phaseX(d_0, 1, dsrc, d_m2, pc, s_0, d_m1, s_m1, wo);                           // X(k) + sines(k); D(k+3); operands of W(k-1) requested
#ifdef SF_EXPERIMENT_STAMP
      { const unsigned long long t_b4 = __builtin_amdgcn_s_memtime(); st_bar += (t_b1 - t_b0) + (t_b3 - t_b2); st_x += t_b2 - t_b1; st_w += t_b4 - t_b3; st_n += 1; }
#endif
#ifdef SF_EXPERIMENT_STAMP2
      st2_n += 1;
#endif
      { const uint32_t t = d_m2; d_m2 = d_m1; d_m1 = d_0; d_0 = d_p1; d_p1 = d_p2; d_p2 = t; }
      { const uint32_t t = s_m1; s_m1 = s_0; s_0 = s_p1; s_p1 = t; }
      pc = (pc + 8192u) & 0x6000u;
      pD += blk_stride;
      pO += blk_stride;
      if (k + AP + 1 < nblk) pP += blk_stride;
    }
    bar_all();     // no LDS-DMA of this workgroup is in flight when its LDS is handed on
  }
#ifdef SF_EXPERIMENT_STAMP2
  if (a.dbg && lane == 0 && (wave == 0 || wave == 5) && blockIdx.x == 3 && st2_n) {
    float* o = a.dbg + (wave == 0 ? 0 : 32);
#pragma unroll
    for (int i = 0; i < 32; ++i) o[i] = (float)st2[i] / (float)st2_n;
  }
#endif
#if defined(SF_EXPERIMENT_STAMP) && !defined(SF_EXPERIMENT_STAMP2)
  if (a.dbg && lane == 0 && (wave == 0 || wave == 5) && (blockIdx.x == 3 || blockIdx.x == 200) && st_n) {
    float* o = a.dbg + 32 + ((blockIdx.x == 3 ? 0 : 2) + (wave == 0 ? 0 : 1)) * 4;
    o[0] = (float)st_bar / (float)st_n; o[1] = (float)st_x / (float)st_n; o[2] = (float)st_w / (float)st_n; o[3] = (float)st_n;
  }
#endif
  float* slab = a.slab + (size_t)blockIdx.x * (256 * 256 + 256);
  const int cl = lane & 31, hh = lane >> 5;
#pragma unroll
  for (int x = 0; x < WJ; ++x)
#pragma unroll
    for (int y = 0; y < WI; ++y)
#pragma unroll
      for (int t = 0; t < 16; ++t)
        slab[(size_t)(32 * (wr * WJ + x) + nu8(rho(t, hh))) * 256 + 32 * (wc * WI + y) + cl] = acc[x][y][t];
  {
    const float tsum = dbs + __shfl_xor(dbs, 32);
    if (hh == 0) slab[256 * 256 + 32 * (wr * WJ + DBX) + nu8(cl)] = tsum;
  }
}

template <int PARK>
__global__ __launch_bounds__(512) void k_bwd8h(Bwd8Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // the kernel addresses LDS absolutely (the table at address 0 is what makes a table address one instruction)
  if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem != 0u) __builtin_trap();
  // MODE.FP16_OVFL = 1: fp8 conversions saturate at +-448 instead of returning NaN (epi_probe.hip)
  __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  switch (wave) {   // (row group = role, column index = bias-gradient tile: every wave runs its own copy of the loop)
    case 0: bwd8h_body<PARK, 0, 0>(a, wave, lane); break;
    case 1: bwd8h_body<PARK, 1, 0>(a, wave, lane); break;
    case 2: bwd8h_body<PARK, 2, 0>(a, wave, lane); break;
    case 3: bwd8h_body<PARK, 3, 0>(a, wave, lane); break;
    case 4: bwd8h_body<PARK, 0, 1>(a, wave, lane); break;
    case 5: bwd8h_body<PARK, 1, 1>(a, wave, lane); break;
    case 6: bwd8h_body<PARK, 2, 1>(a, wave, lane); break;
    default: bwd8h_body<PARK, 3, 1>(a, wave, lane); break;
  }
}

}  // namespace sf
