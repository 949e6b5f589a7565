/* av1o_pred.c - oracle intra prediction, AV1 spec §7.11.2 (intra prediction process) for
 * square blocks, with the intra edge filter and edge upsampling of §7.11.2.7 - §7.11.2.12 when
 * enable_intra_edge_filter is set (Av1oEdgeCtl), no filter-intra, palette or CfL.
 *   DC §7.11.2.5, V/H/directional §7.11.2.4, smooth §7.11.2.6, Paeth §7.11.2.2 (recursive
 *   intra = n/a).  Edge preparation (AboveRow/LeftCol incl. the 127/129 base values) is done by
 *   the caller (av1o_enc.c: prepare_edges) following §7.11.2 steps 1-7.
 * Oracle code (test infrastructure): see av1o.h.
 */
#include "av1o.h"
#include <stdlib.h>
#include <string.h>

static const uint8_t sm_weights_4[4] = { 255, 149, 85, 64 };
static const uint8_t sm_weights_8[8] = { 255, 197, 146, 105, 73, 50, 37, 32 };
static const uint8_t sm_weights_16[16] = { 255, 225, 196, 170, 145, 123, 102, 84, 68, 54, 43, 33, 26, 20, 17, 16 };
static const uint8_t sm_weights_32[32] = { 255, 240, 225, 210, 196, 182, 169, 157, 145, 133, 122, 111, 101, 92, 83, 74,
                                           66, 59, 52, 45, 39, 34, 29, 25, 21, 17, 14, 12, 10, 9, 8, 8 };
static const uint8_t sm_weights_64[64] = { 255, 248, 240, 233, 225, 218, 210, 203, 196, 189, 182, 176, 169, 163, 156, 150,
                                           144, 138, 133, 127, 121, 116, 111, 106, 101, 96, 91, 86, 82, 77, 73, 69,
                                           65, 61, 57, 54, 50, 47, 44, 41, 38, 35, 32, 29, 27, 25, 22, 20,
                                           18, 16, 15, 13, 12, 10, 9, 8, 7, 6, 6, 5, 5, 4, 4, 4 };
static const uint8_t *sm_weights(int log2n) {
  switch (log2n) {
    case 2: return sm_weights_4;
    case 3: return sm_weights_8;
    case 4: return sm_weights_16;
    case 5: return sm_weights_32;
    default: return sm_weights_64;
  }
}

/* Dr_Intra_Derivative (spec §7.11.2.4 table), indexed by angle in degrees (only the 3-degree
 * grid positions are populated) */
static int dr_intra_derivative(int angle) {
  static const int16_t tab[][2] = {
    { 3, 1023 }, { 6, 547 }, { 9, 372 }, { 14, 273 }, { 17, 215 }, { 20, 178 }, { 23, 151 }, { 26, 132 }, { 29, 116 },
    { 32, 102 }, { 36, 90 }, { 39, 80 }, { 42, 71 }, { 45, 64 }, { 48, 57 }, { 51, 51 }, { 54, 45 }, { 58, 40 },
    { 61, 35 }, { 64, 31 }, { 67, 27 }, { 70, 23 }, { 73, 19 }, { 76, 15 }, { 81, 11 }, { 84, 7 }, { 87, 3 } };
  unsigned i;
  for (i = 0; i < sizeof(tab) / sizeof(tab[0]); i++)
    if (tab[i][0] == angle) return tab[i][1];
  return 0;
}

static const int16_t mode_to_angle[9] = { 0, 90, 180, 45, 135, 113, 157, 203, 67 };

/* §7.11.2.9 intra edge filter strength selection */
static int ef_strength(int w, int h, int type, int delta) {
  const int d = abs(delta), wh = w + h;
  int s = 0;
  if (type == 0) {
    if (wh <= 8) { if (d >= 56) s = 1; }
    else if (wh <= 12) { if (d >= 40) s = 1; }
    else if (wh <= 16) { if (d >= 40) s = 1; }
    else if (wh <= 24) { if (d >= 8) s = 1; if (d >= 16) s = 2; if (d >= 32) s = 3; }
    else if (wh <= 32) { if (d >= 1) s = 1; if (d >= 4) s = 2; if (d >= 32) s = 3; }
    else { if (d >= 1) s = 3; }
  } else {
    if (wh <= 8) { if (d >= 40) s = 1; if (d >= 64) s = 2; }
    else if (wh <= 16) { if (d >= 20) s = 1; if (d >= 48) s = 2; }
    else if (wh <= 24) { if (d >= 4) s = 3; }
    else { if (d >= 1) s = 3; }
  }
  return s;
}
/* §7.11.2.10 intra edge upsample selection */
static int ef_use_upsample(int w, int h, int type, int delta) {
  const int d = abs(delta), wh = w + h;
  if (d <= 0 || d >= 40) return 0;
  return type ? wh <= 8 : wh <= 16;
}
/* §7.11.2.12 intra edge filter: p[0] is element -1 of the edge, sz elements; element -1 itself is only read */
static void ef_filter(uint16_t *p, int sz, int strength) {
  static const int K[3][5] = { { 0, 4, 8, 4, 0 }, { 0, 5, 6, 5, 0 }, { 2, 4, 4, 4, 2 } };
  uint16_t edge[2 * 64 + 8];
  int i, j;
  if (strength == 0) return;
  memcpy(edge, p, sizeof(uint16_t) * (size_t)sz);
  for (i = 1; i < sz; i++) {
    int s = 0;
    for (j = 0; j < 5; j++) {
      int k = i - 2 + j;
      k = k < 0 ? 0 : (k > sz - 1 ? sz - 1 : k);
      s += K[strength - 1][j] * edge[k];
    }
    p[i] = (uint16_t)((s + 8) >> 4);
  }
}
/* §7.11.2.11 intra edge upsample: buf[0] is element 0; elements -2 .. 2 * num_px - 2 are written */
static void ef_upsample(uint16_t *buf, int num_px, int bd) {
  int dup[64 + 3], i;
  const int maxv = (1 << bd) - 1;
  dup[0] = buf[-1];
  for (i = -1; i < num_px; i++) dup[i + 2] = buf[i];
  dup[num_px + 2] = buf[num_px - 1];
  buf[-2] = (uint16_t)dup[0];
  for (i = 0; i < num_px; i++) {
    int s = -dup[i] + 9 * dup[i + 1] + 9 * dup[i + 2] - dup[i + 3];
    s = (s + 8) >> 4;
    s = s < 0 ? 0 : (s > maxv ? maxv : s);
    buf[2 * i - 1] = (uint16_t)s;
    buf[2 * i] = (uint16_t)dup[i + 2];
  }
}

void av1o_predict_intra(uint16_t *dst, int stride, int log2n, int mode, int angle_delta,
                        const uint16_t *above_m1, const uint16_t *left_m1, int have_above, int have_left, int bd) {
  av1o_predict_intra_ef(dst, stride, log2n, mode, angle_delta, above_m1, left_m1, have_above, have_left, bd, NULL);
}

void av1o_predict_intra_ef(uint16_t *dst, int stride, int log2n, int mode, int angle_delta,
                           const uint16_t *above_m1, const uint16_t *left_m1, int have_above, int have_left, int bd,
                           const Av1oEdgeCtl *ef) {
  const int n = 1 << log2n;
  const uint16_t *A = above_m1 + 1; /* A[-1] .. A[2n-1] */
  const uint16_t *L = left_m1 + 1;
  int i, j;
  if (mode == DC_PRED) {
    int sum = 0, v;
    if (have_above && have_left) {
      for (i = 0; i < n; i++) sum += A[i] + L[i];
      v = (sum + n) >> (log2n + 1);
    } else if (have_left) {
      for (i = 0; i < n; i++) sum += L[i];
      v = (sum + (n >> 1)) >> log2n;
    } else if (have_above) {
      for (i = 0; i < n; i++) sum += A[i];
      v = (sum + (n >> 1)) >> log2n;
    } else {
      v = 1 << (bd - 1);
    }
    for (i = 0; i < n; i++)
      for (j = 0; j < n; j++) dst[i * stride + j] = (uint16_t)v;
    return;
  }
  if (mode == PAETH_PRED) {
    for (i = 0; i < n; i++)
      for (j = 0; j < n; j++) {
        int base = A[j] + L[i] - A[-1];
        int pl = abs(base - L[i]), pt = abs(base - A[j]), ptl = abs(base - A[-1]);
        dst[i * stride + j] = (pl <= pt && pl <= ptl) ? L[i] : (pt <= ptl ? A[j] : A[-1]);
      }
    return;
  }
  if (mode == SMOOTH_PRED || mode == SMOOTH_V_PRED || mode == SMOOTH_H_PRED) {
    const uint8_t *w = sm_weights(log2n);
    for (i = 0; i < n; i++)
      for (j = 0; j < n; j++) {
        int v;
        if (mode == SMOOTH_PRED)
          v = (w[i] * A[j] + (256 - w[i]) * L[n - 1] + w[j] * L[i] + (256 - w[j]) * A[n - 1] + 256) >> 9;
        else if (mode == SMOOTH_V_PRED)
          v = (w[i] * A[j] + (256 - w[i]) * L[n - 1] + 128) >> 8;
        else
          v = (w[j] * L[i] + (256 - w[j]) * A[n - 1] + 128) >> 8;
        dst[i * stride + j] = (uint16_t)v;
      }
    return;
  }
  /* directional (V_PRED..D67_PRED with angle delta), §7.11.2.4 */
  {
    int p_angle = mode_to_angle[mode] + angle_delta * 3;
    int up_a = 0, up_l = 0;   /* upsampleAbove, upsampleLeft */
    uint16_t ea[2 * 64 + 24], el[2 * 64 + 24];
    if (ef && ef->enable) {
      /* working copies with room for element -2 (upsampling): EA[i] = element i */
      uint16_t *EA = ea + 8, *EL = el + 8;
      memcpy(EA - 1, above_m1, sizeof(uint16_t) * (size_t)(2 * n + 1));
      memcpy(EL - 1, left_m1, sizeof(uint16_t) * (size_t)(2 * n + 1));
      if (p_angle != 90 && p_angle != 180) {
        if (p_angle > 90 && p_angle < 180 && 2 * n >= 24) {   /* §7.11.2.7 filter corner */
          const int c = (EL[0] * 5 + EA[-1] * 6 + EA[0] * 5 + 8) >> 4;
          EA[-1] = EL[-1] = (uint16_t)c;
        }
        if (have_above) ef_filter(EA - 1, ef->n_top + (p_angle < 90 ? n : 0) + 1, ef_strength(n, n, ef->filter_type, p_angle - 90));
        if (have_left) ef_filter(EL - 1, ef->n_left + (p_angle > 180 ? n : 0) + 1, ef_strength(n, n, ef->filter_type, p_angle - 180));
      }
      up_a = ef_use_upsample(n, n, ef->filter_type, p_angle - 90);
      if (up_a) ef_upsample(EA, n + (p_angle < 90 ? n : 0), bd);
      up_l = ef_use_upsample(n, n, ef->filter_type, p_angle - 180);
      if (up_l) ef_upsample(EL, n + (p_angle > 180 ? n : 0), bd);
      A = EA; L = EL;
    }
    if (p_angle == 90) {
      for (i = 0; i < n; i++)
        for (j = 0; j < n; j++) dst[i * stride + j] = A[j];
    } else if (p_angle == 180) {
      for (i = 0; i < n; i++)
        for (j = 0; j < n; j++) dst[i * stride + j] = L[i];
    } else if (p_angle < 90) {
      int dx = dr_intra_derivative(p_angle);
      int max_base = (2 * n - 1) << up_a;
      for (i = 0; i < n; i++)
        for (j = 0; j < n; j++) {
          int idx = (i + 1) * dx;
          int base = (idx >> (6 - up_a)) + (j << up_a);
          int shift = ((idx << up_a) >> 1) & 0x1F;
          if (base < max_base)
            dst[i * stride + j] = (uint16_t)((A[base] * (32 - shift) + A[base + 1] * shift + 16) >> 5);
          else
            dst[i * stride + j] = A[max_base];
        }
    } else if (p_angle < 180) {
      int dx = dr_intra_derivative(180 - p_angle);
      int dy = dr_intra_derivative(p_angle - 90);
      for (i = 0; i < n; i++)
        for (j = 0; j < n; j++) {
          int idx = (j << 6) - (i + 1) * dx;
          int base = idx >> (6 - up_a);
          if (base >= -(1 << up_a)) {
            int shift = ((idx << up_a) >> 1) & 0x1F;
            dst[i * stride + j] = (uint16_t)((A[base] * (32 - shift) + A[base + 1] * shift + 16) >> 5);
          } else {
            int shift;
            idx = (i << 6) - (j + 1) * dy;
            base = idx >> (6 - up_l);
            shift = ((idx << up_l) >> 1) & 0x1F;
            dst[i * stride + j] = (uint16_t)((L[base] * (32 - shift) + L[base + 1] * shift + 16) >> 5);
          }
        }
    } else {
      int dy = dr_intra_derivative(270 - p_angle);
      for (i = 0; i < n; i++)
        for (j = 0; j < n; j++) {
          int idx = (j + 1) * dy;
          int base = (idx >> (6 - up_l)) + (i << up_l);
          int shift = ((idx << up_l) >> 1) & 0x1F;
          dst[i * stride + j] = (uint16_t)((L[base] * (32 - shift) + L[base + 1] * shift + 16) >> 5);
        }
    }
  }
}
