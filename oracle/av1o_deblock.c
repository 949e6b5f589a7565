/* av1o_deblock.c - oracle deblocking filter: AV1 spec §7.14 (loop filter process) for this build's block structure
 * (square blocks, transform == block, no segmentation, no loop-filter deltas): §7.14.2 edge loop filter, §7.14.3 filter
 * size, §7.14.4 adaptive filter strength, §7.14.6 sample filtering (narrow 4-tap filter and the wide 6/8/14-tap
 * filters).  Per plane: every vertical edge of the frame, then every horizontal edge.  SURVEY.md §8a row a19.
 * Oracle code (test infrastructure): see av1o.h. */
#include "av1o.h"
#include <stdlib.h>

static int iabs(int v) { return v < 0 ? -v : v; }
static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* one sample position across an edge; px[-k] = p(k-1), px[k] = q(k); step = distance between taps */
static void filter_sample(uint16_t *px, int step, int plane, int limit, int blimit, int thresh, int size, int bd) {
  const int sh = bd - 8, one = 1 << sh;
  const int lim = limit << sh, blim = blimit << sh, thr = thresh << sh;
#define P(k) ((int)px[-((k) + 1) * step])
#define Q(k) ((int)px[(k) * step])
  int hev, mask, flat = 0, flat2 = 0;
  int p0 = P(0), p1 = P(1), q0 = Q(0), q1 = Q(1);
  hev = iabs(p1 - p0) > thr || iabs(q1 - q0) > thr;
  mask = iabs(p1 - p0) > lim || iabs(q1 - q0) > lim || iabs(p0 - q0) * 2 + iabs(p1 - q1) / 2 > blim;
  if (size >= 6) mask |= iabs(P(2) - p1) > lim || iabs(Q(2) - q1) > lim;
  if (size >= 8) mask |= iabs(P(3) - P(2)) > lim || iabs(Q(3) - Q(2)) > lim;
  if (mask) return; /* filterMask = 0 */
  if (size >= 6) {
    flat = iabs(p1 - p0) <= one && iabs(q1 - q0) <= one && iabs(P(2) - p0) <= one && iabs(Q(2) - q0) <= one;
    if (size >= 8) flat = flat && iabs(P(3) - p0) <= one && iabs(Q(3) - q0) <= one;
  }
  if (size >= 16) flat2 = iabs(P(4) - p0) <= one && iabs(Q(4) - q0) <= one && iabs(P(5) - p0) <= one && iabs(Q(5) - q0) <= one &&
                          iabs(P(6) - p0) <= one && iabs(Q(6) - q0) <= one;
  if (size == 4 || !flat) {
    /* narrow filter §7.14.6.3 */
    const int lo = -(1 << (bd - 1)), hi = (1 << (bd - 1)) - 1, half = 0x80 << sh;
    int ps1 = p1 - half, ps0 = p0 - half, qs0 = q0 - half, qs1 = q1 - half;
    int filter = hev ? clampi(ps1 - qs1, lo, hi) : 0, f1, f2;
    filter = clampi(filter + 3 * (qs0 - ps0), lo, hi);
    f1 = clampi(filter + 4, lo, hi) >> 3;
    f2 = clampi(filter + 3, lo, hi) >> 3;
    px[0] = (uint16_t)(clampi(qs0 - f1, lo, hi) + half);
    px[-step] = (uint16_t)(clampi(ps0 + f2, lo, hi) + half);
    if (!hev) {
      filter = (f1 + 1) >> 1;
      px[step] = (uint16_t)(clampi(qs1 - filter, lo, hi) + half);
      px[-2 * step] = (uint16_t)(clampi(ps1 + filter, lo, hi) + half);
    }
  } else {
    /* wide filter §7.14.6.4 */
    const int log2size = (size == 16 && flat2) ? 4 : 3;
    const int n = log2size == 4 ? 6 : (plane == 0 ? 3 : 2), n2 = (log2size == 3 && plane == 0) ? 0 : 1;  /* taps sum to 1 << log2size */
    int in[16], out[16], i, j;
    for (i = -(n + 1); i <= n; i++) in[i + 8] = i < 0 ? P(-i - 1) : Q(i);
    for (i = -n; i < n; i++) {
      int t = 0;
      for (j = -n; j <= n; j++) {
        int p = clampi(i + j, -(n + 1), n);
        t += in[p + 8] * (iabs(j) <= n2 ? 2 : 1);
      }
      out[i + 8] = (t + (1 << (log2size - 1))) >> log2size;
    }
    for (i = -n; i < n; i++) px[i * step] = (uint16_t)out[i + 8];
  }
#undef P
#undef Q
}

/* Encoder side: the levels for a frame.  deblock == 2: as configured.  deblock == 1: from the quantiser, the "pick from q"
 * rule of libaom (filt_guess = (q * 20723 + 1015158) >> 18 at 8-bit scale, 4 less on key frames), the same level for
 * all four filters. */
#include "../av1-base_amd/csrc/av1_tables.h"
void av1o_deblock_levels(const Av1oConfig *cfg, int is_key, int *levels) {
  int i;
  if (cfg->deblock == 2) { for (i = 0; i < 4; i++) levels[i] = cfg->lf_level[i]; return; }
  if (!cfg->deblock) { for (i = 0; i < 4; i++) levels[i] = 0; return; }
  {
    const int q = cfg->bit_depth == 8 ? av1_ac_q8[cfg->base_q_idx] : av1_ac_q10[cfg->base_q_idx];
    int g = cfg->bit_depth == 8 ? (q * 20723 + 1015158) >> 18 : (q * 20723 + 4060632) >> 20;
    if (is_key) g -= 4;
    g = clampi(g, 0, 63);
    for (i = 0; i < 4; i++) levels[i] = g;
  }
}

/* levels[4] = loop_filter_level[0] (luma vertical edges), [1] (luma horizontal), [2] (U), [3] (V).  frame: the coded
 * (padded) reconstruction, filtered in place; tw/th: the signalled frame size (edges beyond it are not filtered). */
void av1o_deblock_frame(const Av1oConfig *cfg, Av1oFrame *f, const uint8_t *mi_bsl, const uint8_t *mi_skip, const uint8_t *mi_is_inter,
                        int mi_stride, const int *levels, int sharpness) {
  const int bd = cfg->bit_depth;
  const int tw = cfg->true_width ? cfg->true_width : cfg->width, th = cfg->true_height ? cfg->true_height : cfg->height;
  const int mi_rows = cfg->height / 4, mi_cols = cfg->width / 4;
  int plane, pass, row, col, i;
  for (plane = 0; plane < 3; plane++) {
    const int ss = plane > 0;
    if (plane == 0 && !levels[0] && !levels[1]) continue;
    if (plane > 0 && !levels[plane + 1]) continue;
    for (pass = 0; pass < 2; pass++) {
      const int lvl = plane == 0 ? levels[pass] : levels[plane + 1];
      const int shift = sharpness > 4 ? 2 : (sharpness > 0 ? 1 : 0);
      const int limit = sharpness > 0 ? clampi(lvl >> shift, 1, 9 - sharpness) : (lvl >> shift > 1 ? lvl >> shift : 1);
      const int blimit = 2 * (lvl + 2) + limit, thresh = lvl >> 4;
      const int stride = f->stride[plane];
      if (!lvl) continue;
      for (row = 0; row < mi_rows; row += 1 << ss)
        for (col = 0; col < mi_cols; col += 1 << ss) {
          const int x = (col * 4) >> ss, y = (row * 4) >> ss; /* plane position of this 4x4 (chroma: of the 8x8 luma area) */
          const int prow = row - (pass ? 1 << ss : 0), pcol = col - (pass ? 0 : 1 << ss);
          int bsl, pbsl, txw, ptxw, skip, intra, is_block_edge, is_tx_edge, size;
          if (col * 4 >= tw || row * 4 >= th) continue;         /* onScreen */
          if (pass == 0 ? x == 0 : y == 0) continue;
          bsl = mi_bsl[row * mi_stride + col];
          pbsl = mi_bsl[prow * mi_stride + pcol];
          txw = (1 << (bsl > 6 ? 6 : bsl)) >> ss;                /* transform == block, at most 64 */
          ptxw = (1 << (pbsl > 6 ? 6 : pbsl)) >> ss;
          if (txw < 4) txw = 4;
          if (ptxw < 4) ptxw = 4;
          skip = mi_skip[row * mi_stride + col];
          intra = !mi_is_inter[row * mi_stride + col];
          is_tx_edge = ((pass == 0 ? x : y) % txw) == 0;       /* square blocks aligned to their size: block edge == tx edge */
          is_block_edge = is_tx_edge;
          if (!is_tx_edge) continue;
          if (!(is_block_edge || !skip || intra)) continue;
          size = txw < ptxw ? txw : ptxw;
          if (plane == 0) { if (size > 16) size = 16; }
          else { if (size > 8) size = 6; else if (size == 8) size = 6; }  /* chroma: Min(8, base) -> the 6-tap filter for 8 */
          if (plane > 0 && size < 6) size = 4;
          for (i = 0; i < 4; i++) {
            uint16_t *px = f->p[plane] + (size_t)(y + (pass ? 0 : i)) * stride + x + (pass ? i : 0);
            filter_sample(px, pass ? stride : 1, plane, limit, blimit, thresh, size, bd);
          }
        }
    }
  }
}
