/* av1o_pred.c - oracle intra prediction, AV1 spec §7.11.2 (intra prediction process) for
 * square blocks with enable_intra_edge_filter = 0 (so no edge filter / upsampling:
 * §7.11.2.4 with upsampleAbove = upsampleLeft = 0) and no filter-intra, palette or CfL.
 *   DC §7.11.2.5, V/H/directional §7.11.2.4, smooth §7.11.2.6, Paeth §7.11.2.2 (recursive
 *   intra = n/a).  Edge preparation (AboveRow/LeftCol incl. the 127/129 base values) is done by
 *   the caller (av1o_enc.c: prepare_edges) following §7.11.2 steps 1-7.
 * Oracle code (test infrastructure): see av1o.h.
 */
#include "av1o.h"
#include <stdlib.h>

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

void av1o_predict_intra(uint16_t *dst, int stride, int log2n, int mode, int angle_delta,
                        const uint16_t *above_m1, const uint16_t *left_m1, int have_above, int have_left, int bd) {
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
  /* directional (V_PRED..D67_PRED with angle delta) */
  {
    int p_angle = mode_to_angle[mode] + angle_delta * 3;
    if (p_angle == 90) {
      for (i = 0; i < n; i++)
        for (j = 0; j < n; j++) dst[i * stride + j] = A[j];
    } else if (p_angle == 180) {
      for (i = 0; i < n; i++)
        for (j = 0; j < n; j++) dst[i * stride + j] = L[i];
    } else if (p_angle < 90) {
      int dx = dr_intra_derivative(p_angle);
      int max_base = 2 * n - 1;
      for (i = 0; i < n; i++)
        for (j = 0; j < n; j++) {
          int idx = (i + 1) * dx;
          int base = (idx >> 6) + j;
          int shift = (idx >> 1) & 0x1F;
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
          int base = idx >> 6;
          if (base >= -1) {
            int shift = (idx >> 1) & 0x1F;
            dst[i * stride + j] = (uint16_t)((A[base] * (32 - shift) + A[base + 1] * shift + 16) >> 5);
          } else {
            int shift;
            idx = (i << 6) - (j + 1) * dy;
            base = idx >> 6;
            shift = (idx >> 1) & 0x1F;
            dst[i * stride + j] = (uint16_t)((L[base] * (32 - shift) + L[base + 1] * shift + 16) >> 5);
          }
        }
    } else {
      int dy = dr_intra_derivative(270 - p_angle);
      for (i = 0; i < n; i++)
        for (j = 0; j < n; j++) {
          int idx = (j + 1) * dy;
          int base = (idx >> 6) + i;
          int shift = (idx >> 1) & 0x1F;
          dst[i * stride + j] = (uint16_t)((L[base] * (32 - shift) + L[base + 1] * shift + 16) >> 5);
        }
    }
  }
}
