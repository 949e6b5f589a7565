// entropy_kernel.hip - AV1 tile entropy coding: one wavefront per tile (= 64x64 superblock),
// every tile of every frame of the chunk in one launch.
//
// Replaces the entropy-coding stage of the external SVT-AV1 worker behind `run_av1an`
// (/root/reference/crates/daemon/src/encode/av1an.rs:126-139).
// Bitstream syntax written: AV1 spec §5.11.4 decode_partition, §5.11.7 intra_frame_mode_info,
// §5.11.39 coeffs, §5.11.47 transform_type, contexts §8.3.2, symbol coder §8.2 (mirror).
//
// MI355X mapping: the tile's adaptive CDF set (11 KB) lives in LDS; the symbol
// sequence is inherently serial (adaptive CDFs + range coder), so the wave runs it in lock-step
// with wave-uniform control flow: lane i adapts CDF entry i (the per-symbol adaptation loop is
// one vector op), coefficient contexts for a whole transform block are computed lane-parallel
// before the serial pass, level/scan tables are staged in LDS, and output bytes are staged in
// LDS and flushed as coalesced 256-byte bursts.  Carry propagation uses a pending-byte/0xFF-run
// counter so output is append-only.
#include <hip/hip_runtime.h>
#include "av1mi_dev.h"

namespace {

typedef Av1miCdfLayout CL;

struct EcLds {
  uint16_t cdf[CL::TOTAL];
  int16_t lv[32 * 32];
  uint16_t scanpos[32 * 32];
  uint8_t ctx_base[32 * 32];
  uint8_t ctx_br[32 * 32];
  Av1miBlkInfo info[64];
  uint8_t above_lvl[3][16], above_dc[3][16], left_lvl[3][16], left_dc[3][16];
  uint32_t stage[64];  // 256 output bytes
};

struct Ec {
  EcLds *S;
  int lane;
  // range coder (identical in every lane)
  uint32_t low, rng;
  int cnt;
  // carry-free output: pending byte + run of 0xFF
  int pending;       // -1 = none yet
  int ff_run;
  int out_pos;       // bytes emitted so far (incl. staged)
  uint8_t *out;      // global slot
  int out_cap;
  int adapt;
  uint32_t nsym;
};

__device__ __forceinline__ void raw_byte(Ec &e, int b) {
  if (e.lane == 0) reinterpret_cast<volatile uint8_t *>(e.S->stage)[e.out_pos & 255] = (uint8_t)b;
  e.out_pos++;
  if ((e.out_pos & 255) == 0) {
    __syncthreads();
    const int base = e.out_pos - 256;
    if (base + 256 <= e.out_cap) reinterpret_cast<uint32_t *>(e.out + base)[e.lane] = e.S->stage[e.lane];
    __syncthreads();
  }
}
// byte with possible carry (bit 8) from the range coder
__device__ __forceinline__ void put_byte(Ec &e, unsigned v) {
  const int carry = (v >> 8) & 1, b = v & 0xFF;
  if (carry) {
    raw_byte(e, (e.pending + 1) & 0xFF);  // pending always exists when a carry arrives
    for (int i = 0; i < e.ff_run; i++) raw_byte(e, 0x00);
    e.ff_run = 0;
    e.pending = b;
  } else if (b == 0xFF) {
    if (e.pending < 0) e.pending = b; else e.ff_run++;
  } else {
    if (e.pending >= 0) raw_byte(e, e.pending);
    for (int i = 0; i < e.ff_run; i++) raw_byte(e, 0xFF);
    e.ff_run = 0;
    e.pending = b;
  }
}

__device__ __forceinline__ void ec_normalize(Ec &e, uint32_t low, uint32_t rng) {
  int c = e.cnt;
  const int d = __builtin_clz(rng) - 16;  // 16 - ilog(rng)
  int s = c + d;
  if (s >= 0) {
    c += 16;
    uint32_t m = (1u << c) - 1;
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
  e.low = low << d;
  e.rng = rng << d;
  e.cnt = s;
}

// encode symbol s with the n-symbol inverted CDF at LDS offset `off`; lane i adapts entry i.
__device__ __forceinline__ void write_sym(Ec &e, int s, int off, int n) {
  volatile uint16_t *cdf = e.S->cdf + off;
  const uint32_t fl = s > 0 ? cdf[s - 1] : 32768u, fh = cdf[s];
  const uint32_t cntr = cdf[n];
  uint32_t l = e.low, r = e.rng;
  const int N = n - 1;
  if (fl < 32768u) {
    const uint32_t u = (((r >> 8) * (fl >> 6)) >> 1) + 4 * (N - (s - 1));
    const uint32_t v = (((r >> 8) * (fh >> 6)) >> 1) + 4 * (N - s);
    l += r - u;
    r = u - v;
  } else {
    r -= (((r >> 8) * (fh >> 6)) >> 1) + 4 * (N - s);
  }
  if (e.adapt) {
    const int rate = 3 + (cntr > 15) + (cntr > 31) + (n > 3 ? 2 : 1);
    if (e.lane < n - 1) {
      uint32_t v = cdf[e.lane];
      if (e.lane < s) v += (32768u - v) >> rate; else v -= v >> rate;
      cdf[e.lane] = (uint16_t)v;
    } else if (e.lane == n) {
      cdf[n] = (uint16_t)(cntr + (cntr < 32));
    }
  }
  ec_normalize(e, l, r);
  e.nsym++;
}
__device__ __forceinline__ void write_bool(Ec &e, int val, uint32_t f) {
  uint32_t l = e.low, r = e.rng;
  const uint32_t v = (((r >> 8) * (f >> 6)) >> 1) + 4;
  if (val) l += r - v;
  r = val ? v : r - v;
  ec_normalize(e, l, r);
  e.nsym++;
}
__device__ __forceinline__ void write_literal(Ec &e, unsigned v, int bits) {
  for (int i = bits - 1; i >= 0; i--) write_bool(e, (v >> i) & 1, 16384);
}
__device__ __forceinline__ int ec_finish(Ec &e) {
  uint32_t l = e.low;
  int c = e.cnt, s = 10;
  const uint32_t m = 0x3FFF;
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
  if (e.pending >= 0) raw_byte(e, e.pending);
  for (int i = 0; i < e.ff_run; i++) raw_byte(e, 0xFF);
  // flush the partial staging burst
  __syncthreads();
  const int base = e.out_pos & ~255, rem = e.out_pos - base;
  if (rem > 0 && base + 256 <= e.out_cap) {
    if (e.lane * 4 < rem) reinterpret_cast<uint32_t *>(e.out + base)[e.lane] = e.S->stage[e.lane];
  }
  return e.out_pos;
}

__device__ __forceinline__ int iabs(int v) { return v < 0 ? -v : v; }
__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }
__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }
__device__ __forceinline__ int floor_log2(unsigned v) { return 31 - __builtin_clz(v); }

__device__ __forceinline__ int scan_index(int row, int col, int n) {
  int d = row + col;
  int before = d < n ? (d * (d + 1)) >> 1 : n * n - (((2 * n - 1 - d) * (2 * n - d)) >> 1);
  int lo = d - (n - 1) > 0 ? d - (n - 1) : 0;
  return before + ((d & 1) ? row - lo : col - lo);
}

__constant__ uint8_t c_base_ctx_off[5][5] = { { 0, 1, 6, 6, 21 }, { 1, 6, 6, 21, 21 }, { 6, 6, 21, 21, 21 }, { 6, 21, 21, 21, 21 }, { 21, 21, 21, 21, 21 } };
__constant__ uint8_t c_intra_mode_ctx[13] = { 0, 1, 2, 3, 4, 4, 4, 4, 3, 0, 1, 2, 0 };
__constant__ uint8_t c_mode_txfm[13] = { 0, 1, 2, 0, 3, 1, 2, 2, 1, 3, 1, 2, 3 };
// symbol of {DCT_DCT, ADST_DCT, DCT_ADST, ADST_ADST} in intra set 1 (7 symbols) / set 2 (5 symbols)
__constant__ uint8_t c_txsym_set1[4] = { 1, 5, 6, 4 };
__constant__ uint8_t c_txsym_set2[4] = { 1, 3, 4, 2 };

struct TileGeo {
  int sb_x, sb_y;   // luma pixel origin
  int max_x4[2], max_y4[2];  // frame limits in 4x4 units, superblock-local, per plane class (luma, chroma)
};

// coefficients of one transform block (spec §5.11.39); log2n = transform size, x4/y4 in plane 4x4
// units local to the superblock.
__device__ void write_coeffs(Ec &e, const Av1miDevParams &P, const TileGeo &tg, int plane, int log2n, int x4, int y4,
                             int eob, int ymode, int uvmode, const int16_t *lv_global) {
  EcLds *S = e.S;
  const int ptype = plane > 0, pc = ptype;
  const int txs = log2n - 2;
  const int n = 1 << log2n, w4 = n >> 2;
  int ctx;
  // all_zero context
  if (plane == 0) {
    ctx = 0;  // TX_MODE_LARGEST with square blocks: transform == block
  } else {
    int above = 0, left = 0;
    for (int k = 0; k < w4; k++) {
      if (x4 + k < tg.max_x4[pc]) above |= S->above_lvl[plane][x4 + k] | S->above_dc[plane][x4 + k];
      if (y4 + k < tg.max_y4[pc]) left |= S->left_lvl[plane][y4 + k] | S->left_dc[plane][y4 + k];
    }
    ctx = 7 + (above != 0) + (left != 0);
  }
  write_sym(e, eob == 0, CL::TXB_SKIP + (txs * 13 + ctx) * 3, 2);
  if (eob == 0) {
    __syncthreads();
    if (e.lane < w4) {
      if (x4 + e.lane < tg.max_x4[pc]) { S->above_lvl[plane][x4 + e.lane] = 0; S->above_dc[plane][x4 + e.lane] = 0; }
      if (y4 + e.lane < tg.max_y4[pc]) { S->left_lvl[plane][y4 + e.lane] = 0; S->left_dc[plane][y4 + e.lane] = 0; }
    }
    __syncthreads();
    return;
  }
  // stage levels + scan table + contexts (lane-parallel)
  {
    const uint32_t *g32 = reinterpret_cast<const uint32_t *>(lv_global);
    uint32_t *l32 = reinterpret_cast<uint32_t *>(S->lv);
    for (int i = e.lane; i < n * n / 2; i += 64) l32[i] = g32[i];
    __syncthreads();
    for (int p = e.lane; p < n * n; p += 64) {
      const int row = p >> log2n, col = p & (n - 1);
      S->scanpos[scan_index(row, col, n)] = (uint16_t)p;
#define LVA(r_, c_) (((r_) < n && (c_) < n) ? iabs(S->lv[((r_) << log2n) + (c_)]) : 0)
      const int a01 = LVA(row, col + 1), a10 = LVA(row + 1, col), a11 = LVA(row + 1, col + 1), a02 = LVA(row, col + 2), a20 = LVA(row + 2, col);
#undef LVA
      int mag = imin(a01, 3) + imin(a10, 3) + imin(a11, 3) + imin(a02, 3) + imin(a20, 3);
      int cb = p == 0 ? 0 : imin((mag + 1) >> 1, 4) + c_base_ctx_off[imin(row, 4)][imin(col, 4)];
      S->ctx_base[p] = (uint8_t)cb;
      int mb = imin(a01, 15) + imin(a10, 15) + imin(a11, 15);
      mb = imin((mb + 1) >> 1, 6);
      S->ctx_br[p] = (uint8_t)(p == 0 ? mb : ((row < 2 && col < 2) ? mb + 7 : mb + 14));
    }
    __syncthreads();
  }
  // transform_type (luma, sets with more than one type)
  if (plane == 0 && log2n <= 4) {
    const int tt = c_mode_txfm[ymode];
    if (log2n <= 3) write_sym(e, c_txsym_set1[tt], CL::TX_SET1 + ((log2n - 2) * 13 + ymode) * 8, 7);
    else write_sym(e, c_txsym_set2[tt], CL::TX_SET2 + ((log2n - 2) * 13 + ymode) * 6, 5);
  }
  (void)uvmode;
  // eob
  {
    const int eob_pt = eob <= 2 ? eob : floor_log2((unsigned)(eob - 1)) + 2;
    const int base = eob_pt < 2 ? eob_pt : ((1 << (eob_pt - 2)) + 1);
    const int extra = eob - base;
    const int msz = 2 * log2n - 4;
    const int nsy = 5 + msz;
    // EOB16..EOB1024 tables are consecutive: [2][2][nsy+1] each, nsy = 5..11
    const int eoff = CL::EOB16 + 4 * (msz * 6 + (msz * (msz - 1)) / 2);
    write_sym(e, eob_pt - 1, eoff + (ptype * 2 + 0) * (nsy + 1), nsy);
    if (eob_pt >= 3) {
      const int nbits = eob_pt - 2;
      write_sym(e, (extra >> (nbits - 1)) & 1, CL::EOB_EXTRA + ((txs * 2 + ptype) * 9 + (eob_pt - 3)) * 3, 2);
      for (int i = 1; i < nbits; i++) write_literal(e, (unsigned)((extra >> (nbits - 1 - i)) & 1), 1);
    }
  }
  // levels in reverse scan order
  const int br_txs = txs > 3 ? 3 : txs;
  for (int c = eob - 1; c >= 0; c--) {
    const int pos = S->scanpos[c];
    const int level = iabs(S->lv[pos]);
    if (c == eob - 1) {
      const int cctx = c == 0 ? 0 : (c <= (n * n) / 8 ? 1 : (c <= (n * n) / 4 ? 2 : 3));
      write_sym(e, imin(level, 3) - 1, CL::COEFF_BASE_EOB + ((txs * 2 + ptype) * 4 + cctx) * 4, 3);
    } else {
      write_sym(e, imin(level, 3), CL::COEFF_BASE + ((txs * 2 + ptype) * 42 + S->ctx_base[pos]) * 5, 4);
    }
    if (level > 2) {
      const int boff = CL::COEFF_BR + ((br_txs * 2 + ptype) * 21 + S->ctx_br[pos]) * 5;
      for (int idx = 0; idx < 4; idx++) {
        const int k3 = imin(level - 3 - idx * 3, 3);
        write_sym(e, k3, boff, 4);
        if (k3 < 3) break;
      }
    }
  }
  // signs / golomb in forward order, context bookkeeping
  int cul = 0, dc_cat = 0;
  for (int c = 0; c < eob; c++) {
    const int pos = S->scanpos[c];
    const int v = S->lv[pos], level = iabs(v);
    if (!level) continue;
    if (c == 0) {
      int dsum = 0;
      for (int k = 0; k < w4; k++) {
        if (x4 + k < tg.max_x4[pc]) { int s = S->above_dc[plane][x4 + k]; dsum += s == 1 ? -1 : (s == 2 ? 1 : 0); }
        if (y4 + k < tg.max_y4[pc]) { int s = S->left_dc[plane][y4 + k]; dsum += s == 1 ? -1 : (s == 2 ? 1 : 0); }
      }
      const int dctx = dsum < 0 ? 1 : (dsum > 0 ? 2 : 0);
      write_sym(e, v < 0, CL::DC_SIGN + (ptype * 3 + dctx) * 3, 2);
    } else {
      write_literal(e, v < 0, 1);
    }
    if (level > 14) {
      const unsigned g = (unsigned)(level - 15) + 1;
      const int len = floor_log2(g) + 1;
      for (int i = 0; i < len - 1; i++) write_literal(e, 0, 1);
      for (int i = len - 1; i >= 0; i--) write_literal(e, (g >> i) & 1, 1);
    }
    cul += level;
    if (pos == 0) dc_cat = v < 0 ? 1 : 2;
  }
  cul = imin(cul, 63);
  __syncthreads();
  if (e.lane < w4) {
    if (x4 + e.lane < tg.max_x4[pc]) { S->above_lvl[plane][x4 + e.lane] = (uint8_t)cul; S->above_dc[plane][x4 + e.lane] = (uint8_t)dc_cat; }
    if (y4 + e.lane < tg.max_y4[pc]) { S->left_lvl[plane][y4 + e.lane] = (uint8_t)cul; S->left_dc[plane][y4 + e.lane] = (uint8_t)dc_cat; }
  }
  __syncthreads();
}

__device__ __forceinline__ int icdf_prob(volatile uint16_t *c, int el) { return (el > 0 ? c[el - 1] : 32768) - c[el]; }

// split decision shared with the recon kernel
__device__ __forceinline__ bool node_split(const Av1miDevParams &P, int sb_x, int sb_y, int ox, int oy, int bsl) {
  const int n = 1 << bsl;
  bool split;
  if (bsl <= P.min_bs_log2 || bsl == 3) split = false;
  else if (bsl > P.max_bs_log2) split = true;
  else split = false;
  if (sb_y + oy + n > P.height || sb_x + ox + n > P.width) split = true;
  if (bsl == 3) split = false;
  return split;
}

__global__ void __launch_bounds__(64) entropy_tile_kernel(Av1miDevParams P, const uint16_t *__restrict__ cdf_init,
                                                         const int16_t *__restrict__ levels, const Av1miBlkInfo *__restrict__ blk,
                                                         uint8_t *__restrict__ slots, uint32_t *__restrict__ tile_bytes,
                                                         uint32_t *__restrict__ sym_count) {
  __shared__ EcLds S;
  const int sbs_per_frame = P.sb_rows * P.sb_cols;
  const int f = blockIdx.x / sbs_per_frame, sb = blockIdx.x % sbs_per_frame;
  const int sbr = sb / P.sb_cols, sbc = sb % P.sb_cols;
  const int lane = threadIdx.x;
  for (int i = lane; i < CL::TOTAL; i += 64) S.cdf[i] = cdf_init[i];
  {
    const Av1miBlkInfo *info = blk + (size_t)f * P.b8_rows * P.b8_cols + (size_t)(sbr * 8) * P.b8_cols + sbc * 8;
    const int r = lane >> 3, c = lane & 7;
    Av1miBlkInfo bi = {};
    if (sbr * 8 + r < P.b8_rows && sbc * 8 + c < P.b8_cols) bi = info[r * P.b8_cols + c];
    S.info[lane] = bi;
    if (lane < 48) { (&S.above_lvl[0][0])[lane] = 0; (&S.above_dc[0][0])[lane] = 0; (&S.left_lvl[0][0])[lane] = 0; (&S.left_dc[0][0])[lane] = 0; }
  }
  __syncthreads();
  Ec e;
  e.S = &S; e.lane = lane; e.low = 0; e.rng = 0x8000; e.cnt = -9; e.pending = -1; e.ff_run = 0; e.out_pos = 0;
  e.out = slots + (size_t)blockIdx.x * P.tile_slot_bytes; e.out_cap = P.tile_slot_bytes; e.adapt = !P.disable_cdf_update; e.nsym = 0;
  TileGeo tg;
  tg.sb_x = sbc * 64; tg.sb_y = sbr * 64;
  tg.max_x4[0] = P.mi_cols - sbc * 16; tg.max_y4[0] = P.mi_rows - sbr * 16;
  tg.max_x4[1] = (P.mi_cols >> 1) - sbc * 8; tg.max_y4[1] = (P.mi_rows >> 1) - sbr * 8;
  const int16_t *sb_levels = levels + ((size_t)f * sbs_per_frame + sb) * AV1MI_SB_LEVELS;

  for (int z = 0; z < 64; z++) {
    const int bx = (((z >> 0) & 1) | ((z >> 1) & 2) | ((z >> 2) & 4)) << 3;
    const int by = (((z >> 1) & 1) | ((z >> 2) & 2) | ((z >> 3) & 4)) << 3;
    if (tg.sb_y + by >= P.height || tg.sb_x + bx >= P.width) continue;
    // walk the nodes whose origin is (bx, by), largest first
    for (int bsl = 6; bsl >= 3; bsl--) {
      const int n = 1 << bsl;
      if ((bx | by) & (n - 1)) continue;
      // this node is reached iff every ancestor is split
      bool reached = true;
      for (int a = 6; a > bsl; a--) {
        const int an = 1 << a;
        if (!node_split(P, tg.sb_x, tg.sb_y, bx & ~(an - 1), by & ~(an - 1), a)) { reached = false; break; }
      }
      if (!reached) break;  // inside a larger leaf that was coded at its own origin
      const bool split = node_split(P, tg.sb_x, tg.sb_y, bx, by, bsl);
      // ---- partition symbol (spec §5.11.4)
      {
        const int half = n >> 1;
        const bool has_rows = tg.sb_y + by + half < P.height, has_cols = tg.sb_x + bx + half < P.width;
        const int b8x = bx >> 3, b8y = by >> 3;
        const int above = by > 0 && S.info[(b8y - 1) * 8 + b8x].bsl < bsl;
        const int left = bx > 0 && S.info[b8y * 8 + b8x - 1].bsl < bsl;
        const int off = CL::PARTITION + ((bsl - 3) * 4 + left * 2 + above) * 11;
        if (has_rows && has_cols) {
          write_sym(e, split ? 3 : 0, off, bsl == 3 ? 4 : 10);
        } else if (has_cols) {
          volatile uint16_t *pc = S.cdf + off;
          int p = icdf_prob(pc, 2) + icdf_prob(pc, 3);
          if (bsl != 3) p += icdf_prob(pc, 4) + icdf_prob(pc, 6) + icdf_prob(pc, 7) + icdf_prob(pc, 9);
          write_bool(e, 1, (uint32_t)p);
        } else if (has_rows) {
          volatile uint16_t *pc = S.cdf + off;
          int p = icdf_prob(pc, 1) + icdf_prob(pc, 3);
          if (bsl != 3) p += icdf_prob(pc, 4) + icdf_prob(pc, 5) + icdf_prob(pc, 6) + icdf_prob(pc, 8);
          write_bool(e, 1, (uint32_t)p);
        }
      }
      if (split) continue;
      // ---- leaf block: intra_frame_mode_info + residual
      {
        const int b8x = bx >> 3, b8y = by >> 3;
        const Av1miBlkInfo bi = S.info[b8y * 8 + b8x];
        const int ymode = bi.ymode, skip = bi.skip;
        const int avail_u = by > 0, avail_l = bx > 0;
        int sctx = 0;
        if (avail_u) sctx += S.info[(b8y - 1) * 8 + b8x].skip;
        if (avail_l) sctx += S.info[b8y * 8 + b8x - 1].skip;
        write_sym(e, skip, CL::SKIP + sctx * 3, 2);
        const int am = c_intra_mode_ctx[avail_u ? S.info[(b8y - 1) * 8 + b8x].ymode : 0];
        const int lm = c_intra_mode_ctx[avail_l ? S.info[b8y * 8 + b8x - 1].ymode : 0];
        write_sym(e, ymode, CL::KF_Y_MODE + (am * 5 + lm) * 14, 13);
        if (ymode >= 1 && ymode <= 8) write_sym(e, 3, CL::ANGLE_DELTA + (ymode - 1) * 8, 7);
        const int uvmode = ymode;
        const int cfl_allowed = n <= 32;
        write_sym(e, uvmode, CL::UV_MODE + (cfl_allowed * 13 + ymode) * 15, cfl_allowed ? 14 : 13);
        if (uvmode >= 1 && uvmode <= 8) write_sym(e, 3, CL::ANGLE_DELTA + (uvmode - 1) * 8, 7);
        const int w4 = n >> 2, w4c = imax(w4 >> 1, 1);
        const int log2c = bsl - 1;
        if (skip) {
          __syncthreads();
          if (lane < w4) { S.above_lvl[0][(bx >> 2) + lane] = 0; S.above_dc[0][(bx >> 2) + lane] = 0; S.left_lvl[0][(by >> 2) + lane] = 0; S.left_dc[0][(by >> 2) + lane] = 0; }
          if (lane < w4c) {
            for (int pl = 1; pl < 3; pl++) { S.above_lvl[pl][(bx >> 3) + lane] = 0; S.above_dc[pl][(bx >> 3) + lane] = 0; S.left_lvl[pl][(by >> 3) + lane] = 0; S.left_dc[pl][(by >> 3) + lane] = 0; }
          }
          __syncthreads();
        } else {
          write_coeffs(e, P, tg, 0, bsl, bx >> 2, by >> 2, bi.eob[0], ymode, uvmode, sb_levels + by * 64 + bx * n);
          write_coeffs(e, P, tg, 1, log2c, bx >> 3, by >> 3, bi.eob[1], ymode, uvmode, sb_levels + 4096 + (by >> 1) * 32 + (bx >> 1) * (n >> 1));
          write_coeffs(e, P, tg, 2, log2c, bx >> 3, by >> 3, bi.eob[2], ymode, uvmode, sb_levels + 5120 + (by >> 1) * 32 + (bx >> 1) * (n >> 1));
        }
      }
      break;
    }
  }
  const int nbytes = ec_finish(e);
  if (lane == 0) {
    tile_bytes[blockIdx.x] = (uint32_t)nbytes;
    sym_count[blockIdx.x] = e.nsym;
  }
}

}  // namespace

extern "C" hipError_t av1mi_launch_entropy(const Av1miDevParams *P, const uint16_t *cdf_init, const int16_t *levels,
                                           const Av1miBlkInfo *blk, uint8_t *slots, uint32_t *tile_bytes,
                                           uint32_t *sym_count, hipStream_t stream) {
  const int grid = P->n_frames * P->sb_rows * P->sb_cols;
  hipLaunchKernelGGL(entropy_tile_kernel, dim3(grid), dim3(64), 0, stream, *P, cdf_init, levels, blk, slots, tile_bytes, sym_count);
  return hipGetLastError();
}
