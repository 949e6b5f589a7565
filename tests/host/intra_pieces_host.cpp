// Host build of the product's piece-wise intra predictors (av1-base_amd/csrc/intra_pieces.h) for tests/test_intra_pieces.py: the same
// source the reconstruction kernel compiles, driven lane by lane, so that every mode / angle / block size can be checked against the
// oracle's predictor without a GPU.  Test infrastructure only.
#include <cstdint>
#include <cstring>
#include "../../av1-base_amd/csrc/intra_pieces.h"

static const int16_t k_deriv[91] = AV1MI_DR_DERIV_INIT;
static const uint32_t k_magic[91] = AV1MI_DR_MAGIC_INIT;

namespace pc = av1mi_pieces;

template <int N, int G>
static long run(int mode, int ang, int dcv, const uint16_t *EA, const uint16_t *EL, const uint8_t *smw, const uint16_t *src, uint16_t *pix, int write) {
  alignas(16) static uint16_t tsrc[N * N];
  alignas(16) static uint16_t s[N * N];
  alignas(16) static uint16_t p[N * N];
  memcpy(s, src, sizeof(s));
  int dx = 0, dy = 0;
  if (pc::is_dir(mode, ang)) {
    if (ang < 90) dx = k_deriv[ang];
    else if (ang < 180) { dx = k_deriv[180 - ang]; dy = k_deriv[ang - 90]; }
    else dy = k_deriv[270 - ang];
  }
  const uint32_t magic = (pc::is_dir(mode, ang) && ang > 90 && ang < 180) ? k_magic[180 - ang] : 0u;
  long acc = 0;
  if (!write) for (int sl = 0; sl < G; sl++) pc::transpose_lane<N, G>(sl, s, tsrc);
  for (int sl = 0; sl < G; sl++) acc += pc::pass_t<N, G>(sl, mode, ang, dy, magic, EL, tsrc, p, write != 0);
  for (int sl = 0; sl < G; sl++) acc += pc::pass_n<N, G>(sl, mode, ang, dx, dcv, EA, EA, EL, smw, s, p, write != 0);
  if (write) memcpy(pix, p, sizeof(p));
  return acc;
}

// edges: pointers to element 0 of arrays with elements -8 .. 3 N + 8 addressable (element -1 the corner, 2 N .. padded)
extern "C" long pieces_run(int n, int lanes, int mode, int ang, int dcv, const uint16_t *above, const uint16_t *left, const uint8_t *smw,
                           const uint16_t *src, uint16_t *pix, int write) {
  // lanes: 64 = a whole wave on one block (luma), 32 = half a wave (a chroma plane of a U / V pair)
  switch (n * 100 + lanes) {
    case 864: return run<8, 64>(mode, ang, dcv, above, left, smw, src, pix, write);
    case 832: return run<8, 32>(mode, ang, dcv, above, left, smw, src, pix, write);
    case 1664: return run<16, 64>(mode, ang, dcv, above, left, smw, src, pix, write);
    case 1632: return run<16, 32>(mode, ang, dcv, above, left, smw, src, pix, write);
    case 3264: return run<32, 64>(mode, ang, dcv, above, left, smw, src, pix, write);
    case 3232: return run<32, 32>(mode, ang, dcv, above, left, smw, src, pix, write);
    case 6464: return run<64, 64>(mode, ang, dcv, above, left, smw, src, pix, write);
  }
  return -1;
}
extern "C" int pieces_deriv(int ang) { return k_deriv[ang]; }
extern "C" unsigned pieces_magic(int ang) { return k_magic[ang]; }
