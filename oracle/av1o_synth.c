/* av1o_synth.c - `synthclip v1`: deterministic, integer-only synthetic clip so
 * that C, numpy and the HIP generator agree byte for byte.  Per plane sample
 *   clamp( tri(x+dx*t,P1)*A1/64 + tri(y+dy*t,P2)*A2/64 + rect_k(x,y,t) + noise*G, 0, 2^bd-1 )
 * with K = 6 opaque rectangles moving (+-3,+-2) px/frame, a (2,1) global pan, splitmix64 noise,
 * and a hard scene cut every scene_len frames (re-seed with seed + scene index).
 * Oracle code (test infrastructure): see av1o.h.
 */
#include "av1o.h"

static uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
/* triangle wave in [0,64] with period 2P */
static int tri(int v, int P) {
  int m = v % (2 * P);
  if (m < 0) m += 2 * P;
  if (m > P) m = 2 * P - m;
  return (m * 64) / P;
}

void av1o_synthclip_frame(Av1oFrame *f, int bit_depth, uint64_t seed, int t, int scene_len) {
  int scene = scene_len > 0 ? t / scene_len : 0;
  int tt = scene_len > 0 ? t % scene_len : t;
  uint64_t s = seed + (uint64_t)scene;
  int sh = bit_depth - 8, maxv = (1 << bit_depth) - 1, G = bit_depth == 8 ? 1 : 4;
  int pl, x, y, k;
  int rx[6], ry[6], rw[6], rh[6], rv[6][3];
  for (k = 0; k < 6; k++) {
    uint64_t h = splitmix64(s * 977 + (uint64_t)k);
    int vx = (h & 1) ? 3 : -3, vy = (h & 2) ? 2 : -2;
    rw[k] = 32 + (int)((h >> 8) % (uint64_t)(f->w / 4 + 1));
    rh[k] = 32 + (int)((h >> 24) % (uint64_t)(f->h / 4 + 1));
    rx[k] = (int)((h >> 40) % (uint64_t)f->w) + vx * tt;
    ry[k] = (int)((h >> 52) % (uint64_t)f->h) + vy * tt;
    rv[k][0] = (int)((splitmix64(h) >> 3) & 255);
    rv[k][1] = (int)((splitmix64(h) >> 13) & 255);
    rv[k][2] = (int)((splitmix64(h) >> 23) & 255);
  }
  for (pl = 0; pl < 3; pl++) {
    int ss = pl > 0, pw = f->w >> ss, ph = f->h >> ss;
    int A1 = pl ? 48 : 96, A2 = pl ? 32 : 64, P1 = pl ? 53 : 97, P2 = pl ? 41 : 61;
    int base = pl ? 128 - (A1 + A2) / 2 : 16;
    for (y = 0; y < ph; y++)
      for (x = 0; x < pw; x++) {
        int fx = x << ss, fy = y << ss; /* luma-domain position for the rectangles */
        int v = base + (tri(x + ((2 * tt) >> ss), P1) * A1) / 64 + (tri(y + (tt >> ss), P2) * A2) / 64;
        uint64_t nz;
        for (k = 0; k < 6; k++) {
          int px = ((rx[k] % f->w) + f->w) % f->w, py = ((ry[k] % f->h) + f->h) % f->h;
          if (fx >= px && fx < px + rw[k] && fy >= py && fy < py + rh[k]) v = rv[k][pl];
        }
        v <<= sh;
        nz = splitmix64(s ^ ((uint64_t)tt << 40) ^ ((uint64_t)(pl * 8192 + y) << 20) ^ (uint64_t)x);
        v += ((int)(nz & 15) - 8) * G;
        if (v < 0) v = 0;
        if (v > maxv) v = maxv;
        f->p[pl][y * f->stride[pl] + x] = (uint16_t)v;
      }
  }
}
