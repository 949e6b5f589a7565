/* av1o_ec.c - multi-symbol range ENCODER mirroring the AV1 symbol decoder (spec §8.2:
 * init_symbol / read_symbol / read_bool / read_literal / exit_symbol) with the CDF adaptation
 * of §8.2.6 ("symbol decoding process": rate = 3 + (cnt>15) + (cnt>31) + min(log2 N, 2)).
 * Oracle code: see av1o.h header note.  CDFs are held INVERTED (32768 - cumulative), n-1
 * values followed by a 0 terminator and the adaptation counter, so an n-symbol CDF is n+1
 * uint16_t.
 */
#include "av1o.h"
#include <string.h>

#define EC_PROB_SHIFT 6
#define EC_MIN_PROB 4

void av1o_ec_init(Av1oRangeEnc *e, uint8_t *buf, size_t cap) {
  e->low = 0;
  e->rng = 0x8000;
  e->cnt = -9;
  e->buf = buf;
  e->cap = cap;
  e->offs = 0;
  e->error = 0;
  e->nsym = 0;
}

static void put_byte(Av1oRangeEnc *e, unsigned v) {
  /* v may carry into the previous bytes (bit 8 set) */
  if (v & 0x100) {
    size_t i = e->offs;
    while (i > 0) {
      --i;
      if (e->buf[i] == 0xFF) {
        e->buf[i] = 0;
      } else {
        e->buf[i]++;
        break;
      }
    }
  }
  if (e->offs >= e->cap) {
    e->error = 1;
    return;
  }
  e->buf[e->offs++] = (uint8_t)v;
}

static int ilog_nz(uint32_t v) { /* number of bits needed: floor(log2 v)+1 */
  return 32 - __builtin_clz(v);
}

static void ec_normalize(Av1oRangeEnc *e, uint32_t low, uint32_t rng) {
  int c = e->cnt;
  int d = 16 - ilog_nz(rng);
  int s = c + d;
  if (s >= 0) {
    uint32_t m;
    c += 16;
    m = (1u << c) - 1;
    if (s >= 8) {
      put_byte(e, low >> c);
      low &= m;
      c -= 8;
      m >>= 8;
    }
    put_byte(e, low >> c);
    s = c + d - 24;
    low &= m;
  }
  e->low = low << d;
  e->rng = rng << d;
  e->cnt = s;
}

/* encode symbol s given fl = icdf[s-1] (32768 for s==0) and fh = icdf[s] */
static void ec_encode_q15(Av1oRangeEnc *e, unsigned fl, unsigned fh, int s, int nsyms) {
  uint32_t l = e->low, r = e->rng;
  const int N = nsyms - 1;
  if (fl < 32768) {
    uint32_t u = (((r >> 8) * (uint32_t)(fl >> EC_PROB_SHIFT)) >> (7 - EC_PROB_SHIFT)) + EC_MIN_PROB * (N - (s - 1));
    uint32_t v = (((r >> 8) * (uint32_t)(fh >> EC_PROB_SHIFT)) >> (7 - EC_PROB_SHIFT)) + EC_MIN_PROB * (N - s);
    l += r - u;
    r = u - v;
  } else {
    r -= (((r >> 8) * (uint32_t)(fh >> EC_PROB_SHIFT)) >> (7 - EC_PROB_SHIFT)) + EC_MIN_PROB * (N - s);
  }
  ec_normalize(e, l, r);
  e->nsym++;
}

void av1o_ec_encode_symbol(Av1oRangeEnc *e, int s, uint16_t *icdf, int nsyms) {
  unsigned fl = s > 0 ? icdf[s - 1] : 32768;
  unsigned fh = icdf[s];
  int i, rate;
  unsigned cnt;
  ec_encode_q15(e, fl, fh, s, nsyms);
  /* adaptation (spec §8.2.6), inverted-cdf form */
  cnt = icdf[nsyms];
  rate = 3 + (cnt > 15) + (cnt > 31) + (nsyms > 3 ? 2 : 1);
  for (i = 0; i < nsyms - 1; i++) {
    if (i < s)
      icdf[i] += (uint16_t)((32768 - icdf[i]) >> rate);
    else
      icdf[i] -= (uint16_t)(icdf[i] >> rate);
  }
  icdf[nsyms] = (uint16_t)(cnt + (cnt < 32));
}

void av1o_ec_encode_bool(Av1oRangeEnc *e, int val, unsigned f) {
  /* f = 32768 * P(val == 1) */
  uint32_t l = e->low, r = e->rng;
  uint32_t v = (((r >> 8) * (uint32_t)(f >> EC_PROB_SHIFT)) >> (7 - EC_PROB_SHIFT)) + EC_MIN_PROB;
  if (val) l += r - v;
  r = val ? v : r - v;
  ec_normalize(e, l, r);
  e->nsym++;
}

void av1o_ec_encode_literal(Av1oRangeEnc *e, unsigned v, int bits) {
  int i;
  for (i = bits - 1; i >= 0; i--) av1o_ec_encode_bool(e, (v >> i) & 1, 16384);
}

size_t av1o_ec_finish(Av1oRangeEnc *e) {
  /* minimum number of bits so that everything coded so far decodes regardless of what
   * follows (the spec's exit_symbol only requires the padding to be consumable). */
  uint32_t l = e->low;
  int c = e->cnt;
  int s = 10;
  uint32_t m = 0x3FFF;
  uint32_t v = ((l + m) & ~m) | (m + 1);
  s += c;
  if (s > 0) {
    uint32_t n = (1u << (c + 16)) - 1;
    do {
      put_byte(e, v >> (c + 16));
      v &= n;
      s -= 8;
      c -= 8;
      n >>= 8;
    } while (s > 0);
  }
  return e->offs;
}
