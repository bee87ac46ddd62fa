// layout.h — data layouts shared by the gfx950 kernels of the SIREN fitting engine.
//
// Everything the hot path keeps in HBM between kernels is laid out in MFMA-fragment order
// ("F-layout") so that every global access is one fully coalesced 16 B/lane, 1 KiB/wave piece
// and no kernel needs a transpose pass:
//
//   MFMA v_mfma_f32_32x32x16_{bf16,f16}:  D[32x32] += A[32x16] * B[16x32]
//     A fragment: lane l holds row (l&31), k = 8*(l>>5) + j, j = 0..7      (8 x 16 bit = 16 B)
//     B fragment: lane l holds col (l&31), k = 8*(l>>5) + j
//     C/D:        lane l holds col (l&31), register t holds row rho(t, l>>5)
//   with rho(t,h) = (t&3) + 8*(t>>2) + 4*h.
//
// The engine runs the network "transposed": rows = neurons, columns = pixels, so a wave owns a
// block of 32 pixels (one per lane column) and all activations of those pixels.  The accumulator
// of layer l is then already the B operand of layer l+1 (registers 8q..8q+7 of neuron tile nt
// form k-step s = 2*nt+q), with the k order inside a step permuted:
//
//   k-step s covers neurons 16*s .. 16*s+15;  element j of lane half h is neuron 16*s + PI(h,j)
//   PI(h,j) = 8*(j>>2) + 4*h + (j&3)
//
// F-layout of an [n_pixels x width] 16-bit matrix X (phases, deltas):
//   piece(pb, s) = 64 lanes x 16 B, lane = h*32 + m   (pb = pixel block of 32, m = pixel in block)
//   element j of that lane = X[32*pb + m][16*s + PI(h,j)]
//   piece index = pb * (width/16) + s ; byte offset = piece index * 1024 + lane * 16
//
// Weight images (built from the fp32 master weights after every optimiser step):
//   forward  image of W[out][in]:  A fragment (nt, s): lane (r,h) elem j = W[32*nt + r][16*s + PI(h,j)]
//   backward image (for dX = delta * W): A fragment (it, s): lane (r,h) elem j = W[16*s + PI(h,j)][32*it + r]
//   both stored as [tile][s][lane][8].
#pragma once
#include <stdint.h>

#define SF_HOSTDEV __host__ __device__ __forceinline__

namespace sf {

constexpr int kPixBlock = 32;   // pixels per MFMA column tile (one wave)
constexpr int kWavesFwd = 8;    // waves per workgroup in the chain kernels
constexpr int kSuper = kPixBlock * kWavesFwd;  // pixels per workgroup pass (256)

SF_HOSTDEV int rho(int t, int h) { return (t & 3) + 8 * (t >> 2) + 4 * h; }
SF_HOSTDEV int pi_perm(int h, int j) { return 8 * (j >> 2) + 4 * h + (j & 3); }

}  // namespace sf
