/* av1o_lr.c - oracle loop restoration: Wiener filter (AV1 spec §7.17.3 loop_restore_block, §7.17.4 wiener filter
 * process, §7.17.6 get_source_sample) for the luma plane with 64x64 restoration units, plus the encoder-side choice
 * of each unit's filter (non-normative): minimum SSE against the source among {off, 3 fixed
 * symmetric filters}.  SURVEY.md §8a row a16.  Oracle code (test infrastructure): see av1o.h. */
#include "av1o.h"
#include <stdlib.h>
#include <string.h>

const int8_t av1o_wiener_candidates[3][3] = { { 0, 0, -4 }, { 1, -3, -6 }, { 3, -7, 15 } };
/* self-guided candidates { lr_sgr_set, xqd0, xqd1 } (enable_lr == 2): one parameter set (9: r0 = 2, eps 68; r1 = 1, eps 15), so
 * both box-filter passes are computed once per sample and the candidates differ only in the final blend - both passes
 * (w0, w1, w2) = (31, 31, 66); the r = 1 pass alone (0, 31, 97); the r = 2 pass almost alone (31, 95, 2).
 * xqd0 in [-96, 31], xqd1 in [-32, 95].  The other 15 sets are exercised by the fuzzed fixtures. */
const int8_t av1o_sgr_candidates[3][3] = { { 9, 31, 31 }, { 9, 0, 31 }, { 9, 31, 95 } };

static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

int av1o_lr_units(int size) { /* count_units_in_frame(64, size) */
  int n = (size + 32) / 64;
  return n < 1 ? 1 : n;
}

/* one sample of the Wiener-filtered luma plane.  cdef = UpscaledCdefFrame, pre = UpscaledCurrFrame (pre-CDEF). */
static int wiener_sample(const Av1oFrame *cdef, const Av1oFrame *pre, int W, int H, int bd, int x, int y, const int *vf, const int *hf) {
  /* W, H: the signalled frame size (PlaneEndX + 1, PlaneEndY + 1 of §7.17.6) */
  const int stripe = (y + 8) / 64, s0 = -8 + stripe * 64, s1 = s0 + 63;
  const int round0 = 3, round1 = 11;
  const int offset = 1 << (bd + 7 - round0 - 1), limit = (1 << (bd + 1 + 7 - round0)) - 1;
  int mid[7], r, t, s;
  for (r = 0; r < 7; r++) {
    int yy = clampi(y + r - 3, 0, H - 1);
    const Av1oFrame *srcf = cdef;
    if (yy < s0) { yy = yy > s0 - 2 ? yy : s0 - 2; srcf = pre; }
    else if (yy > s1) { yy = yy < s1 + 2 ? yy : s1 + 2; srcf = pre; }
    s = 0;
    for (t = 0; t < 7; t++) {
      int xx = clampi(x + t - 3, 0, W - 1);
      s += hf[t] * (int)srcf->p[0][(size_t)yy * srcf->stride[0] + xx];
    }
    s = (s + (1 << (round0 - 1))) >> round0;
    mid[r] = clampi(s, -offset, limit - offset);
  }
  s = 0;
  for (t = 0; t < 7; t++) s += vf[t] * mid[t];
  s = (s + (1 << (round1 - 1))) >> round1;
  return clampi(s, 0, (1 << bd) - 1);
}

/* ---- self-guided restoration (§7.17.3 self guided filter process, box filter process) -------------------------------
 * Sgr_Params[set] = { r0, eps0, r1, eps1 } */
const int av1o_sgr_params[16][4] = { { 2, 12, 1, 4 },  { 2, 15, 1, 6 },  { 2, 18, 1, 8 },  { 2, 21, 1, 9 },  { 2, 24, 1, 10 }, { 2, 29, 1, 11 },
                                     { 2, 36, 1, 12 }, { 2, 45, 1, 13 }, { 2, 56, 1, 14 }, { 2, 68, 1, 15 }, { 0, 0, 1, 5 },   { 0, 0, 1, 8 },
                                     { 0, 0, 1, 11 },  { 0, 0, 1, 14 },  { 2, 30, 0, 0 },  { 2, 75, 0, 0 } };

/* get_source_sample (§7.17.6) for luma */
static int lr_source(const Av1oFrame *cdef, const Av1oFrame *pre, int W, int H, int x, int y, int s0, int s1) {
  const Av1oFrame *f = cdef;
  x = clampi(x, 0, W - 1);
  y = clampi(y, 0, H - 1);
  if (y < s0) { y = y > s0 - 2 ? y : s0 - 2; f = pre; }
  else if (y > s1) { y = y < s1 + 2 ? y : s1 + 2; f = pre; }
  return (int)f->p[0][(size_t)y * f->stride[0] + x];
}

/* A and B of the box filter process at (x, y) for radius r, strength eps (the position may lie one sample outside the block) */
static void sgr_ab(const Av1oFrame *cdef, const Av1oFrame *pre, int W, int H, int bd, int x, int y, int s0, int s1, int r, int eps, int *A, int *B) {
  const int n = (2 * r + 1) * (2 * r + 1), n2e = n * n * eps;
  const uint32_t s = (uint32_t)(((1 << 20) + n2e / 2) / n2e), one_by_n = (uint32_t)(((1 << 12) + n / 2) / n);
  uint32_t a = 0, b = 0, d, p, z, a2;
  int dx, dy;
  for (dy = -r; dy <= r; dy++)
    for (dx = -r; dx <= r; dx++) {
      uint32_t c = (uint32_t)lr_source(cdef, pre, W, H, x + dx, y + dy, s0, s1);
      a += c * c; b += c;
    }
  a = (a + ((1u << (2 * (bd - 8))) >> 1)) >> (2 * (bd - 8));
  d = (b + ((1u << (bd - 8)) >> 1)) >> (bd - 8);
  p = a * (uint32_t)n > d * d ? a * (uint32_t)n - d * d : 0;
  z = (uint32_t)(((uint64_t)p * s + (1u << 19)) >> 20);
  a2 = z >= 255 ? 256 : (z == 0 ? 1 : ((z << 8) + z / 2) / (z + 1));
  *A = (int)a2;
  *B = (int)(((uint64_t)(256 - a2) * b * one_by_n + (1u << 11)) >> 12);
}

/* one sample of the self-guided-filtered luma plane */
static int sgr_sample(const Av1oFrame *cdef, const Av1oFrame *pre, int W, int H, int bd, int x, int y, int set, int w0, int w1) {
  const int stripe = (y + 8) / 64, s0 = -8 + stripe * 64, s1 = s0 + 63;
  const int *prm = av1o_sgr_params[set];
  const int cur = (int)cdef->p[0][(size_t)y * cdef->stride[0] + x];
  const int u = cur << 4, w2 = 128 - w0 - w1;
  int flt[2] = { 0, 0 }, pass, v;
  for (pass = 0; pass < 2; pass++) {
    const int r = prm[pass * 2], eps = prm[pass * 2 + 1];
    int a = 0, b = 0, dx, dy, shift = 5;
    if (!r) continue;
    if (pass == 0 && (y & 1)) shift = 4;
    for (dy = -1; dy <= 1; dy++)
      for (dx = -1; dx <= 1; dx++) {
        int weight, A, B;
        if (pass == 0) weight = ((y + dy) & 1) ? (dx == 0 ? 6 : 5) : 0;
        else weight = (dx == 0 || dy == 0) ? 4 : 3;
        if (!weight) continue;
        sgr_ab(cdef, pre, W, H, bd, x + dx, y + dy, s0, s1, r, eps, &A, &B);
        a += weight * A; b += weight * B;
      }
    v = a * cur + b;
    flt[pass] = (v + (1 << (8 + shift - 4 - 1))) >> (8 + shift - 4);
  }
  v = w1 * u;
  v += w0 * (prm[0] ? flt[0] : u);
  v += w2 * (prm[2] ? flt[1] : u);
  v = (v + (1 << 10)) >> 11;
  return clampi(v, 0, (1 << bd) - 1);
}

/* whole-plane helper for experiments and tests: out = SGR(cdef) with one parameter set everywhere */
void av1o_sgr_plane(const Av1oConfig *cfg, const Av1oFrame *pre, const Av1oFrame *cdef, Av1oFrame *out, int set, int w0, int w1) {
  const int W = cfg->true_width ? cfg->true_width : cfg->width, H = cfg->true_height ? cfg->true_height : cfg->height;
  int x, y;
  for (y = 0; y < H; y++)
    for (x = 0; x < W; x++) out->p[0][(size_t)y * out->stride[0] + x] = (uint16_t)sgr_sample(cdef, pre, W, H, cfg->bit_depth, x, y, set, w0, w1);
}

static void taps_of(const int8_t *c, int *f) {
  f[0] = f[6] = c[0]; f[1] = f[5] = c[1]; f[2] = f[4] = c[2];
  f[3] = 128 - 2 * (c[0] + c[1] + c[2]);
}

/* Decide every luma unit and produce the restored frame.  units[unit_rows * unit_cols]: type 0 = off, 1 = Wiener with
 * coef[pass][3] (pass 0 vertical, 1 horizontal).  `fuzz` != 0: pseudo-random types and coefficients (dav1d fuzzing). */
void av1o_lr_frame(const Av1oConfig *cfg, const Av1oFrame *pre, const Av1oFrame *cdef, const Av1oFrame *src, Av1oFrame *out,
                   Av1oLrUnit *units, unsigned fuzz) {
  /* units, stripes and sample clamps follow the SIGNALLED frame size; the frames themselves hold the padded size */
  const int W = cfg->true_width ? cfg->true_width : cfg->width, H = cfg->true_height ? cfg->true_height : cfg->height, bd = cfg->bit_depth;
  const int CW = cfg->width, CH = cfg->height;
  const int urows = av1o_lr_units(H), ucols = av1o_lr_units(W);
  static const int8_t tmin[3] = { -5, -23, -17 }, tmax[3] = { 10, 8, 46 };
  int ur, uc, p, x, y, k;
  unsigned rng = fuzz * 2654435761u + 12345u;
  for (p = 0; p < 3; p++) /* chroma: FrameRestorationType = NONE; luma: overwritten inside the signalled area below */
    for (y = 0; y < (CH >> (p > 0)); y++) memcpy(out->p[p] + (size_t)y * out->stride[p], cdef->p[p] + (size_t)y * cdef->stride[p], sizeof(uint16_t) * (size_t)(CW >> (p > 0)));
  for (ur = 0; ur < urows; ur++)
    for (uc = 0; uc < ucols; uc++) {
      /* unit rows are offset by 8 luma rows (§7.17.3): rows [64*ur - 8, 64*ur + 56), the last unit to the frame end */
      const int y0 = ur ? ur * 64 - 8 : 0, y1 = ur == urows - 1 ? H : ur * 64 + 56;
      const int x0 = uc * 64, x1 = uc == ucols - 1 ? W : uc * 64 + 64;
      Av1oLrUnit *u = &units[ur * ucols + uc];
      uint64_t best_sse = 0;
      int vf[7], hf[7];
      memset(u, 0, sizeof(*u));
      if (fuzz) {
        rng ^= rng << 13; rng ^= rng >> 17; rng ^= rng << 5;
        u->type = (rng & 3) != 0;
        if (cfg->enable_lr == 2 && u->type) {
          rng ^= rng << 13; rng ^= rng >> 17; rng ^= rng << 5;
          if (rng & 1) { /* self-guided with random set and weights (radius-0 passes take their implied weights) */
            int set, x0, x1;
            rng ^= rng << 13; rng ^= rng >> 17; rng ^= rng << 5;
            set = (int)(rng & 15);
            x0 = -96 + (int)((rng >> 4) % 128u);
            x1 = -32 + (int)((rng >> 12) % 128u);
            if (!av1o_sgr_params[set][0]) x0 = 0;
            if (!av1o_sgr_params[set][2]) x1 = clampi(128 - x0, -32, 95);
            u->type = 2; u->sgr_set = (int8_t)set; u->sgr_xqd[0] = (int8_t)x0; u->sgr_xqd[1] = (int8_t)x1;
          }
        }
        for (p = 0; p < 2; p++)
          for (k = 0; k < 3; k++) {
            rng ^= rng << 13; rng ^= rng >> 17; rng ^= rng << 5;
            u->coef[p][k] = (int8_t)(tmin[k] + (int)(rng % (unsigned)(tmax[k] - tmin[k] + 1)));
          }
      } else {
        for (y = y0; y < y1; y++)
          for (x = x0; x < x1; x++) {
            int d = (int)cdef->p[0][(size_t)y * cdef->stride[0] + x] - (int)src->p[0][(size_t)y * src->stride[0] + x];
            best_sse += (uint64_t)(d * d);
          }
        for (k = 0; k < 3; k++) {
          uint64_t sse = 0;
          taps_of(av1o_wiener_candidates[k], vf);
          for (y = y0; y < y1; y++)
            for (x = x0; x < x1; x++) {
              int d = wiener_sample(cdef, pre, W, H, bd, x, y, vf, vf) - (int)src->p[0][(size_t)y * src->stride[0] + x];
              sse += (uint64_t)(d * d);
            }
          if (sse < best_sse) { best_sse = sse; u->type = 1; memcpy(u->coef[0], av1o_wiener_candidates[k], 3); memcpy(u->coef[1], av1o_wiener_candidates[k], 3); }
        }
        for (k = 0; cfg->enable_lr == 2 && k < 3; k++) {
          const int8_t *c = av1o_sgr_candidates[k];
          uint64_t sse = 0;
          for (y = y0; y < y1; y++)
            for (x = x0; x < x1; x++) {
              int d = sgr_sample(cdef, pre, W, H, bd, x, y, c[0], c[1], c[2]) - (int)src->p[0][(size_t)y * src->stride[0] + x];
              sse += (uint64_t)(d * d);
            }
          if (sse < best_sse) { best_sse = sse; u->type = 2; u->sgr_set = c[0]; u->sgr_xqd[0] = c[1]; u->sgr_xqd[1] = c[2]; }
        }
      }
      if (u->type == 1) { taps_of(u->coef[0], vf); taps_of(u->coef[1], hf); }
      for (y = y0; y < y1; y++)
        for (x = x0; x < x1; x++)
          out->p[0][(size_t)y * out->stride[0] + x] = u->type == 1 ? (uint16_t)wiener_sample(cdef, pre, W, H, bd, x, y, vf, hf)
                                                      : u->type == 2 ? (uint16_t)sgr_sample(cdef, pre, W, H, bd, x, y, u->sgr_set, u->sgr_xqd[0], u->sgr_xqd[1])
                                                                     : cdef->p[0][(size_t)y * cdef->stride[0] + x];
    }
}
