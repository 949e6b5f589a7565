/* av1o_cdef.c - oracle CDEF, AV1 spec §7.15 (CDEF process): §7.15.1 cdef_block,
 * §7.15.2 direction search, §7.15.3 filter.  4:2:0 only (Cdef_Uv_Dir is the identity map).
 * One strength set per frame (cdef_bits = 0): cdef_idx_sb[] holds 0 (filter) or -1 (all blocks
 * of that 64x64 were skipped, so the decoder never read a cdef_idx and leaves it at -1).
 * Oracle code (test infrastructure): see av1o.h.
 */
#include "av1o.h"
#include <stdlib.h>
#include <string.h>

static const int cdef_directions[8][2][2] = {
  { { -1, 1 }, { -2, 2 } }, { { 0, 1 }, { -1, 2 } }, { { 0, 1 }, { 0, 2 } }, { { 0, 1 }, { 1, 2 } },
  { { 1, 1 }, { 2, 2 } },   { { 1, 0 }, { 2, 1 } },  { { 1, 0 }, { 2, 0 } }, { { 1, 0 }, { 2, -1 } } };
static const int cdef_pri_taps[2][2] = { { 4, 2 }, { 3, 3 } };
static const int cdef_sec_taps[2][2] = { { 2, 1 }, { 2, 1 } };
static const int div_table[9] = { 0, 840, 420, 280, 210, 168, 140, 120, 105 };

static int floor_log2(unsigned v) { return 31 - __builtin_clz(v); }

static int constrain(int diff, int threshold, int damping) {
  int adj, mag, lim;
  if (!threshold) return 0;
  adj = damping - floor_log2((unsigned)threshold);
  if (adj < 0) adj = 0;
  mag = abs(diff);
  lim = threshold - (mag >> adj);
  if (lim < 0) lim = 0;
  if (lim > mag) lim = mag;
  return diff < 0 ? -lim : lim;
}

static void cdef_direction(const uint16_t *p, int stride, int bd, int *ydir, int *var) {
  int cost[8] = { 0 }, partial[8][15], i, j, d, best = 0, bd_dir = 0;
  memset(partial, 0, sizeof(partial));
  for (i = 0; i < 8; i++)
    for (j = 0; j < 8; j++) {
      int x = (p[i * stride + j] >> (bd - 8)) - 128;
      partial[0][i + j] += x;
      partial[1][i + j / 2] += x;
      partial[2][i] += x;
      partial[3][3 + i - j / 2] += x;
      partial[4][7 + i - j] += x;
      partial[5][3 - i / 2 + j] += x;
      partial[6][j] += x;
      partial[7][i / 2 + j] += x;
    }
  for (i = 0; i < 8; i++) {
    cost[2] += partial[2][i] * partial[2][i];
    cost[6] += partial[6][i] * partial[6][i];
  }
  cost[2] *= div_table[8];
  cost[6] *= div_table[8];
  for (i = 0; i < 7; i++) {
    cost[0] += (partial[0][i] * partial[0][i] + partial[0][14 - i] * partial[0][14 - i]) * div_table[i + 1];
    cost[4] += (partial[4][i] * partial[4][i] + partial[4][14 - i] * partial[4][14 - i]) * div_table[i + 1];
  }
  cost[0] += partial[0][7] * partial[0][7] * div_table[8];
  cost[4] += partial[4][7] * partial[4][7] * div_table[8];
  for (i = 1; i < 8; i += 2) {
    for (j = 0; j < 5; j++) cost[i] += partial[i][3 + j] * partial[i][3 + j];
    cost[i] *= div_table[8];
    for (j = 0; j < 3; j++)
      cost[i] += (partial[i][j] * partial[i][j] + partial[i][10 - j] * partial[i][10 - j]) * div_table[2 * j + 2];
  }
  for (d = 0; d < 8; d++)
    if (cost[d] > best) { best = cost[d]; bd_dir = d; }
  *ydir = bd_dir;
  *var = (best - cost[(bd_dir + 4) & 7]) >> 10;
}

static void cdef_filter(const Av1oFrame *in, Av1oFrame *out, int plane, int mi_r, int mi_c, int pri, int sec,
                        int damping, int dir, int bd, int mi_rows, int mi_cols) {
  int ss = plane > 0, coeff_shift = bd - 8;
  int x0 = (mi_c * 4) >> ss, y0 = (mi_r * 4) >> ss, w = 8 >> ss, h = 8 >> ss, i, j, k, sg, d;
  const uint16_t *src = in->p[plane];
  int stride = in->stride[plane];
  for (i = 0; i < h; i++)
    for (j = 0; j < w; j++) {
      int sum = 0, x = src[(y0 + i) * stride + x0 + j], mx = x, mn = x, v;
      for (k = 0; k < 2; k++)
        for (sg = -1; sg <= 1; sg += 2) {
          int yy = y0 + i + sg * cdef_directions[dir][k][0], xx = x0 + j + sg * cdef_directions[dir][k][1];
          int cr = (yy << ss) >> 2, cc = (xx << ss) >> 2;
          if (yy >= 0 && xx >= 0 && cr < mi_rows && cc < mi_cols) {
            int p = src[yy * stride + xx];
            sum += cdef_pri_taps[(pri >> coeff_shift) & 1][k] * constrain(p - x, pri, damping);
            if (p > mx) mx = p;
            if (p < mn) mn = p;
          }
          for (d = -2; d <= 2; d += 4) {
            int dd = (dir + d) & 7;
            yy = y0 + i + sg * cdef_directions[dd][k][0];
            xx = x0 + j + sg * cdef_directions[dd][k][1];
            cr = (yy << ss) >> 2;
            cc = (xx << ss) >> 2;
            if (yy >= 0 && xx >= 0 && cr < mi_rows && cc < mi_cols) {
              int s = src[yy * stride + xx];
              sum += cdef_sec_taps[(pri >> coeff_shift) & 1][k] * constrain(s - x, sec, damping);
              if (s > mx) mx = s;
              if (s < mn) mn = s;
            }
          }
        }
      v = x + ((8 + sum - (sum < 0)) >> 4);
      if (v < mn) v = mn;
      if (v > mx) v = mx;
      out->p[plane][(y0 + i) * out->stride[plane] + x0 + j] = (uint16_t)v;
    }
}

void av1o_cdef_frame(const Av1oConfig *cfg, const Av1oFrame *in, Av1oFrame *out,
                     const uint8_t *skip_mi, int mi_stride, const int8_t *cdef_idx_sb) {
  int mi_rows = cfg->height / 4, mi_cols = cfg->width / 4, bd = cfg->bit_depth;
  int sb_cols = (cfg->width + 63) / 64, r, c, pl, y;
  int coeff_shift = bd - 8;
  for (pl = 0; pl < 3; pl++) {
    int ph = pl ? in->h / 2 : in->h, pw = pl ? in->w / 2 : in->w;
    for (y = 0; y < ph; y++) memcpy(out->p[pl] + y * out->stride[pl], in->p[pl] + y * in->stride[pl], pw * sizeof(uint16_t));
  }
  if (!cfg->enable_cdef) return;
  for (r = 0; r < mi_rows; r += 2)
    for (c = 0; c < mi_cols; c += 2) {
      int idx = cdef_idx_sb[(r >> 4) * sb_cols + (c >> 4)];
      int skip, ydir, var, pri, sec, dir, var_str, damping;
      if (idx < 0) continue;
      skip = skip_mi[r * mi_stride + c] && skip_mi[(r + 1) * mi_stride + c] && skip_mi[r * mi_stride + c + 1] &&
             skip_mi[(r + 1) * mi_stride + c + 1];
      if (skip) continue;
      cdef_direction(in->p[0] + (r * 4) * in->stride[0] + c * 4, in->stride[0], bd, &ydir, &var);
      pri = cfg->cdef_y_pri << coeff_shift;
      sec = (cfg->cdef_y_sec == 3 ? 4 : cfg->cdef_y_sec) << coeff_shift;
      dir = pri == 0 ? 0 : ydir;
      var_str = (var >> 6) ? (floor_log2((unsigned)(var >> 6)) < 12 ? floor_log2((unsigned)(var >> 6)) : 12) : 0;
      pri = var ? (pri * (4 + var_str) + 8) >> 4 : 0;
      damping = cfg->cdef_damping + coeff_shift;
      cdef_filter(in, out, 0, r, c, pri, sec, damping, dir, bd, mi_rows, mi_cols);
      pri = cfg->cdef_uv_pri << coeff_shift;
      sec = (cfg->cdef_uv_sec == 3 ? 4 : cfg->cdef_uv_sec) << coeff_shift;
      dir = pri == 0 ? 0 : ydir;
      damping = cfg->cdef_damping + coeff_shift - 1;
      cdef_filter(in, out, 1, r, c, pri, sec, damping, dir, bd, mi_rows, mi_cols);
      cdef_filter(in, out, 2, r, c, pri, sec, damping, dir, bd, mi_rows, mi_cols);
    }
}
