// intra_pieces.h - piece-wise intra predictors: eight samples of a block row per lane and step, packed two to a register.
//
// Part of the reconstruction kernel (recon_kernel.hip; AV1 spec §7.11.2, SURVEY.md §8a row a12 - the mode search SVT-AV1 runs inside the
// worker that `run_av1an` spawns, /root/reference/crates/daemon/src/encode/av1an.rs:126-139).  Kept in a header of its own so that the same
// source also compiles for the host: tests/test_intra_pieces.py builds it with clang++ and checks every mode, angle delta and block size
// against the oracle's predictor without a GPU (the only device-specific line is the v_sad_u16 builtin).
//
// A block of N x N samples (N = 8, 16, 32, 64) is N * N / 8 pieces; piece q = row q / (N / 8), samples 8 (q % (N / 8)) .. + 7 of that row;
// lane sl of a group of G lanes takes the pieces q = sl, sl + G, ...
//  * A directional prediction is, row by row, a copy of an edge shifted by a whole number of elements and blended with its neighbour: with
//    idx = d (rho + 1), off = idx >> 6, sh = (idx >> 1) & 31, sample kappa of row rho is (E[kappa + off] (32 - sh) + E[kappa + off + 1] sh
//    + 16) >> 5 - "row form".  Angles below 90: E = above edge, d = dx; the edge is padded beyond 2 N - 1 with its last element, which is what
//    the prediction takes beyond max_base.  Angles above 180: E = left edge, d = dy, in the TRANSPOSED block (rho = column, kappa = row): the
//    candidates' SADs are taken against a transposed copy of the source, the final prediction is scattered.  Between 90 and 180 a sample
//    takes the above form (d = -dx) where (r + 1) dx <= (c + 1) 64 and the left form (d = -dy, transposed) elsewhere; in either layout the
//    samples that take the OTHER form are a prefix of the piece.
//  * DC / V / H / SMOOTH / SMOOTH_V / SMOOTH_H / PAETH: the formulas of the spec on eight columns.
// Edges: E[-1] is the corner, E[0 .. 2 N - 1] the edge, E[2 N .. 3 N + 8] copies of E[2 N - 1]; element -8 must be addressable.
// Vector arithmetic is written on whole 8-element vectors: bit-casting one dword of a 4-dword vector at a time to a 2-element vector
// inside a loop makes clang replicate the first dword's result (DESIGN.md §4.2 viii ran into the same).
#ifndef AV1MI_INTRA_PIECES_H
#define AV1MI_INTRA_PIECES_H
#include <stdint.h>

#if defined(__HIP_DEVICE_COMPILE__)
#define AV1MI_PIECE_FN static __device__ __forceinline__
#else
#define AV1MI_PIECE_FN static inline
#endif

// Dr_Intra_Derivative (AV1 spec 7.11.2.4) indexed by the angle itself (0 where the spec has no entry)
#define AV1MI_DR_DERIV_INIT { \
  0, 0, 0, 1023, 0, 0, 547, 0, 0, 372, 0, 0, 0, 0, 273, 0, 0, 215, 0, 0, 178, 0, 0, 151, 0, 0, 132, 0, 0, 116, 0, 0, \
  102, 0, 0, 0, 90, 0, 0, 80, 0, 0, 71, 0, 0, 64, 0, 0, 57, 0, 0, 51, 0, 0, 45, 0, 0, 0, 40, 0, 0, 35, 0, 0, \
  31, 0, 0, 27, 0, 0, 23, 0, 0, 19, 0, 0, 15, 0, 0, 0, 0, 11, 0, 0, 7, 0, 0, 3, 0, 0, 0 }
// magic[angle] = floor(2^22 / Dr_Intra_Derivative[angle]) + 1: (64 k * magic) >> 22 == floor(64 k / derivative) for k = 1 .. 64 in 64-bit
// arithmetic (checked exhaustively by tests/test_intra_pieces.py)
#define AV1MI_DR_MAGIC_INIT { \
  0, 0, 0, 4101, 0, 0, 7668, 0, 0, 11276, 0, 0, 0, 0, 15364, 0, 0, 19509, 0, 0, 23564, 0, 0, 27777, 0, 0, 31776, 0, 0, 36158, 0, 0, 41121, 0, 0, 0, 46604, \
  0, 0, 52429, 0, 0, 59075, 0, 0, 65537, 0, 0, 73585, 0, 0, 82242, 0, 0, 93207, 0, 0, 0, 104858, 0, 0, 119838, 0, 0, 135301, 0, 0, 155345, 0, 0, 182362, \
  0, 0, 220753, 0, 0, 279621, 0, 0, 0, 0, 381301, 0, 0, 599187, 0, 0, 1398102, 0, 0, 0 }

namespace av1mi_pieces {

typedef unsigned int pu4 __attribute__((ext_vector_type(4)));
typedef unsigned short pus8 __attribute__((ext_vector_type(8)));
typedef short pss8 __attribute__((ext_vector_type(8)));

enum { P_DC = 0, P_V = 1, P_H = 2, P_D67 = 8, P_SMOOTH = 9, P_SMOOTH_V = 10, P_SMOOTH_H = 11, P_PAETH = 12 };

AV1MI_PIECE_FN int p_abs(int v) { return v < 0 ? -v : v; }

// acc + |a.lo - b.lo| + |a.hi - b.hi| on packed 16-bit samples
AV1MI_PIECE_FN uint32_t sad2(uint32_t a, uint32_t b, uint32_t acc) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_sad_u16(a, b, acc);
#else
  return acc + (uint32_t)p_abs((int)(a & 0xFFFF) - (int)(b & 0xFFFF)) + (uint32_t)p_abs((int)(a >> 16) - (int)(b >> 16));
#endif
}

// E[st .. st + 8]: eight packed samples X and the same shifted by one element, Y.  Two 16-byte loads at 2-byte alignment: gfx950 runs with
// unaligned LDS access enabled and the compiler keeps each one ds_read_b128 (nine 2-byte loads and eight packing instructions before)
typedef pu4 pu4_any __attribute__((aligned(2)));
AV1MI_PIECE_FN void load_xy(const uint16_t *E, int st, pu4 &X, pu4 &Y) {
  X = *reinterpret_cast<const pu4_any *>(&E[st]);
  Y = *reinterpret_cast<const pu4_any *>(&E[st + 1]);
}

// 0xFFFF in every halfword j < kinv of a piece (the samples that take the other form)
AV1MI_PIECE_FN pu4 prefix_mask(int kinv) {
  const short kk = (short)(kinv < -64 ? -64 : (kinv > 64 ? 64 : kinv));
  const pss8 h = { 0, 1, 2, 3, 4, 5, 6, 7 };
  return __builtin_bit_cast(pu4, (pss8)((h - kk) >> (short)15));
}

// pv where the mask is clear, sv where it is set (a masked sample then adds nothing to a SAD against sv)
AV1MI_PIECE_FN pu4 merge(pu4 inv, pu4 sv, pu4 pv) { return (inv & sv) | (~inv & pv); }

// one piece of a directional prediction in row form; `off` goes back for the masks.  noblend: d is a multiple of 64 (45 / 135 / 225 degrees)
AV1MI_PIECE_FN pu4 dir_piece(const uint16_t *E, int d, int rho, int k0, bool noblend, int &off) {
  const int idx = d * (rho + 1), sh = (idx >> 1) & 31;
  off = idx >> 6;
  int st = k0 + off;
  st = st < -8 ? -8 : st;   // (a piece that starts further left holds no sample of this form)
  pu4 X, Y;
  load_xy(E, st, X, Y);
  if (noblend) return X;
  const unsigned short w1 = (unsigned short)sh, w0 = (unsigned short)(32 - sh);
  return __builtin_bit_cast(pu4, (pus8)((__builtin_bit_cast(pus8, X) * w0 + __builtin_bit_cast(pus8, Y) * w1 + (unsigned short)16) >> (unsigned short)5));
}

// one piece of a prediction that is not directional (or is V / H at exactly 90 / 180 degrees): raw edges A0 / L0, smooth weights smw.
// One loop per mode (a mode test inside the sample loop compiled to a scalar branch and an LDS wait per sample); the eight above
// samples of the piece come in one 16-byte load.
template <int N>
AV1MI_PIECE_FN pu4 plain_piece(int mode, int rho, int k0, int dcv, const uint16_t *A0, const uint16_t *L0, const uint8_t *smw) {
  if (mode == P_DC || mode == P_H) {
    const uint32_t c2 = (uint32_t)(mode == P_DC ? dcv : (int)L0[rho]) * 0x10001u;
    return (pu4){ c2, c2, c2, c2 };
  }
  const pu4 av = *reinterpret_cast<const pu4 *>(&A0[k0]);
  if (mode == P_V) return av;
  int a[8], px[8];
#pragma unroll
  for (int j = 0; j < 8; j++) a[j] = (int)((av[j >> 1] >> (16 * (j & 1))) & 0xFFFF);
  const int l = L0[rho];
  if (mode == P_PAETH) {
    const int tl = A0[-1], pt = p_abs(l - tl), l2 = l - 2 * tl;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int t = a[j], pl = p_abs(t - tl), ptl = p_abs(t + l2);
      px[j] = (pl <= pt && pl <= ptl) ? l : (pt <= ptl ? t : tl);
    }
  } else if (mode == P_SMOOTH_V) {
    const int wr = smw[rho], kr = (256 - wr) * (int)L0[N - 1] + 128;
#pragma unroll
    for (int j = 0; j < 8; j++) px[j] = (wr * a[j] + kr) >> 8;
  } else {
    int wc[8];
#pragma unroll
    for (int j = 0; j < 8; j++) wc[j] = smw[k0 + j];
    const int ar = A0[N - 1];
    if (mode == P_SMOOTH_H) {
#pragma unroll
      for (int j = 0; j < 8; j++) px[j] = (wc[j] * l + (256 - wc[j]) * ar + 128) >> 8;
    } else {
      const int wr = smw[rho], kr = (256 - wr) * (int)L0[N - 1] + 256;
#pragma unroll
      for (int j = 0; j < 8; j++) px[j] = (wr * a[j] + kr + wc[j] * l + (256 - wc[j]) * ar) >> 9;
    }
  }
  return (pu4){ (uint32_t)px[0] | ((uint32_t)px[1] << 16), (uint32_t)px[2] | ((uint32_t)px[3] << 16), (uint32_t)px[4] | ((uint32_t)px[5] << 16),
                (uint32_t)px[6] | ((uint32_t)px[7] << 16) };
}

// halfword j of a piece
AV1MI_PIECE_FN uint16_t piece_at(pu4 v, int j) { return (uint16_t)(v[j >> 1] >> (16 * (j & 1))); }

// is the prediction of `mode` at `ang` directional in the sense above (not V / H at exactly 90 / 180)?
AV1MI_PIECE_FN bool is_dir(int mode, int ang) { return mode >= P_V && mode <= P_D67 && ang != 90 && ang != 180; }

// ---- one lane's share of a pass over the block.  The prediction of `mode` at `ang` (dx / dy: Dr_Intra_Derivative values as the spec picks
// them for the angle; magic: c_dr_magic[180 - ang] for 90 < ang < 180) from the edges EA / EL (raw or filtered, never upsampled; A0 / L0 are
// the raw ones).  Two passes, in this order; a pass that does not apply to the mode does nothing:
//   pass_t: the left-edge part of a directional prediction above 90 degrees, in the transposed layout.  write == false: the lane's part of the
//           SAD against the transposed source tile `tsrc`; write == true: scattered into `pix` (row-major, stride N).
//   pass_n: everything else, in the block's own layout.  write == false: SAD against `src`; write == true: into `pix` - between 90 and 180
//           degrees merged over what pass_t left there (the caller orders the two passes' LDS accesses).
template <int N, int G>
AV1MI_PIECE_FN uint32_t pass_t(int sl, int mode, int ang, int dy, uint32_t magic, const uint16_t *EL, const uint16_t *tsrc, uint16_t *pix, bool write) {
  constexpr int CPR = N / 8, NPC = N * CPR, PK = (NPC + G - 1) / G;
  uint32_t acc = 0;
  if (!(is_dir(mode, ang) && ang > 90)) return 0;
  const int d = ang > 180 ? dy : -dy;
  const bool noblend = (d & 63) == 0;
#pragma unroll
  for (int k = 0; k < PK; k++) {
    const int q = sl + k * G;
    if (q >= NPC) break;
    const int rho = q / CPR, k0 = 8 * (q % CPR);   // rho = column, k0 = first row of the piece
    int off;
    const pu4 pv = dir_piece(EL, d, rho, k0, noblend, off);
    if (write) {   // (between 90 and 180 the samples of the above form are overwritten by pass_n)
#pragma unroll
      for (int j = 0; j < 8; j++) pix[(k0 + j) * N + rho] = piece_at(pv, j);
    } else {
      const pu4 sv = *reinterpret_cast<const pu4 *>(&tsrc[rho * N + k0]);
      pu4 inv = { 0, 0, 0, 0 };
      if (ang < 180) inv = prefix_mask((int)(((uint64_t)((uint32_t)(rho + 1) * 64u) * magic) >> 22) - k0);   // rows above floor(64 (c + 1) / dx) take the above form
      const pu4 mv = merge(inv, sv, pv);
      acc = sad2(sv[0], mv[0], acc); acc = sad2(sv[1], mv[1], acc); acc = sad2(sv[2], mv[2], acc); acc = sad2(sv[3], mv[3], acc);
    }
  }
  return acc;
}

template <int N, int G>
AV1MI_PIECE_FN uint32_t pass_n(int sl, int mode, int ang, int dx, int dcv, const uint16_t *EA, const uint16_t *A0, const uint16_t *L0, const uint8_t *smw,
                               const uint16_t *src, uint16_t *pix, bool write) {
  constexpr int CPR = N / 8, NPC = N * CPR, PK = (NPC + G - 1) / G;
  uint32_t acc = 0;
  const bool dirm = is_dir(mode, ang);
  if (dirm && ang > 180) return 0;
  const int d = ang < 90 ? dx : -dx;
  const bool noblend = (d & 63) == 0;
#pragma unroll
  for (int k = 0; k < PK; k++) {
    const int q = sl + k * G;
    if (q >= NPC) break;
    const int rho = q / CPR, k0 = 8 * (q % CPR);   // rho = row, k0 = first column of the piece
    pu4 pv, inv = { 0, 0, 0, 0 };
    if (dirm) {
      int off;
      pv = dir_piece(EA, d, rho, k0, noblend, off);
      if (ang > 90) inv = prefix_mask(-1 - off - k0);   // columns left of -1 - off take the left form
    } else {
      pv = plain_piece<N>(mode, rho, k0, dcv, A0, L0, smw);
    }
    if (write) {
      pu4 *dst = reinterpret_cast<pu4 *>(&pix[rho * N + k0]);
      if (dirm && ang > 90) pv = merge(inv, *dst, pv);
      *dst = pv;
    } else {
      const pu4 sv = *reinterpret_cast<const pu4 *>(&src[rho * N + k0]);
      const pu4 mv = merge(inv, sv, pv);
      acc = sad2(sv[0], mv[0], acc); acc = sad2(sv[1], mv[1], acc); acc = sad2(sv[2], mv[2], acc); acc = sad2(sv[3], mv[3], acc);
    }
  }
  return acc;
}

// the transposed copy of the source block the SADs of pass_t are taken against: one lane's share
template <int N, int G>
AV1MI_PIECE_FN void transpose_lane(int sl, const uint16_t *src, uint16_t *tsrc) {
  constexpr int CPR = N / 8, NPC = N * CPR, PK = (NPC + G - 1) / G;
#pragma unroll
  for (int k = 0; k < PK; k++) {
    const int q = sl + k * G;
    if (q >= NPC) break;
    const int r = q / CPR, k0 = 8 * (q % CPR);
    const pu4 sv = *reinterpret_cast<const pu4 *>(&src[r * N + k0]);
#pragma unroll
    for (int j = 0; j < 8; j++) tsrc[(k0 + j) * N + r] = piece_at(sv, j);
  }
}

}  // namespace av1mi_pieces
#endif
