/* av1o_lr.c - oracle loop restoration: Wiener filter (AV1 spec §7.17.3 loop_restore_block, §7.17.4 wiener filter
 * process, §7.17.6 get_source_sample) for the luma plane with 64x64 restoration units, plus the encoder-side choice
 * of each unit's filter (non-normative): minimum SSE against the source among {off, 3 fixed
 * symmetric filters}.  SURVEY.md §8a row a16.  Oracle code (test infrastructure): see av1o.h. */
#include "av1o.h"
#include <stdlib.h>
#include <string.h>

const int8_t av1o_wiener_candidates[3][3] = { { 0, 0, -4 }, { 1, -3, -6 }, { 3, -7, 15 } };

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
      }
      if (u->type) { taps_of(u->coef[0], vf); taps_of(u->coef[1], hf); }
      for (y = y0; y < y1; y++)
        for (x = x0; x < x1; x++)
          out->p[0][(size_t)y * out->stride[0] + x] = u->type ? (uint16_t)wiener_sample(cdef, pre, W, H, bd, x, y, vf, hf)
                                                              : cdef->p[0][(size_t)y * cdef->stride[0] + x];
    }
}
