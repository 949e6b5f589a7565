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
  uint16_t cdf[CL::TOTAL + 64];   // +64: whole-row reads by 17 lanes may run past the last row
  int16_t lv[32 * 32];
  uint16_t scan[1024 + 256 + 64 + 16];  // scan index -> position, for n = 32, 16, 8, 4
  Av1miBlkInfo info[64];
  uint8_t above_lvl[3][16], above_dc[3][16], left_lvl[3][16], left_dc[3][16];
  uint8_t stage8[256];  // output staging
};

// Range-coder state.  Every member is wave-uniform (derived only from kernel arguments and
// readlane/readfirstlane results), so the compiler keeps it in SGPRs and the symbol loop runs on
// the scalar unit; the vector unit only touches the CDF rows.
struct Ec {
  uint32_t low, rng;
  int cnt;
  int pending;       // -1 = none yet
  int ff_run;
  int out_pos;       // bytes emitted so far (incl. staged)
  uint32_t nsym;
};

__shared__ EcLds g_ec;
#define S (&g_ec)
#define EC_ARGS Ec &e, const int lane, uint8_t *const out, const int out_cap, const int adapt

// make a wave-uniform value provably uniform for the compiler (the builtin, not inline asm: the
// hazard recogniser must see the VALU->SGPR write before a following v_readlane lane select)
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

__device__ __forceinline__ uint32_t stage_word(int lane) {
  return (uint32_t)S->stage8[4 * lane] | ((uint32_t)S->stage8[4 * lane + 1] << 8) | ((uint32_t)S->stage8[4 * lane + 2] << 16) |
         ((uint32_t)S->stage8[4 * lane + 3] << 24);
}
__device__ __forceinline__ void raw_byte(EC_ARGS, int b) {
  if (lane == 0) S->stage8[e.out_pos & 255] = (uint8_t)b;
  e.out_pos++;
  if ((e.out_pos & 255) == 0) {
    __syncthreads();
    const int base = e.out_pos - 256;
    if (base + 256 <= out_cap) reinterpret_cast<uint32_t *>(out + base)[lane] = stage_word(lane);
    __syncthreads();
  }
}
// byte with possible carry (bit 8): carry-free append-only output via pending byte + 0xFF run
__device__ __forceinline__ void put_byte(EC_ARGS, unsigned v) {
  const int carry = (v >> 8) & 1, b = v & 0xFF;
  if (carry) {
    raw_byte(e, lane, out, out_cap, adapt, (e.pending + 1) & 0xFF);
    for (int i = 0; i < e.ff_run; i++) raw_byte(e, lane, out, out_cap, adapt, 0x00);
    e.ff_run = 0;
    e.pending = b;
  } else if (b == 0xFF) {
    if (e.pending < 0) e.pending = b; else e.ff_run++;
  } else {
    if (e.pending >= 0) raw_byte(e, lane, out, out_cap, adapt, e.pending);
    for (int i = 0; i < e.ff_run; i++) raw_byte(e, lane, out, out_cap, adapt, 0xFF);
    e.ff_run = 0;
    e.pending = b;
  }
}

__device__ __forceinline__ void ec_normalize(EC_ARGS, uint32_t low, uint32_t rng) {
  int c = e.cnt;
  const int d = __builtin_clz(rng) - 16;  // 16 - ilog(rng)
  int s = c + d;
  if (s >= 0) {
    c += 16;
    uint32_t m = (1u << c) - 1;
    if (s >= 8) {
      put_byte(e, lane, out, out_cap, adapt, low >> c);
      low &= m;
      c -= 8;
      m >>= 8;
    }
    put_byte(e, lane, out, out_cap, adapt, low >> c);
    s = c + d - 24;
    low &= m;
  }
  e.low = low << d;
  e.rng = rng << d;
  e.cnt = s;
}

// range update for symbol s of an n-symbol CDF given fl = icdf[s-1] (32768 if s == 0), fh = icdf[s]
__device__ __forceinline__ void ec_code(EC_ARGS, uint32_t fl, uint32_t fh, int s, int n) {
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
  ec_normalize(e, lane, out, out_cap, adapt, l, r);
  e.nsym++;
}

// Encode symbol s (uniform) with the n-symbol inverted CDF row at LDS offset `off` (uniform).
// Lane j holds row entry j: fl/fh/counter come out through readlane, the adaptation of the whole
// row is one vector op + one LDS store.
__device__ __forceinline__ void write_sym(EC_ARGS, int s, int off, int n) {
  s = uni(s); off = uni(off); n = uni(n);
  const int v = S->cdf[off + (lane < 17 ? lane : 16)];
  const uint32_t fl = s > 0 ? (uint32_t)__builtin_amdgcn_readlane(v, s - 1) : 32768u;
  const uint32_t fh = (uint32_t)__builtin_amdgcn_readlane(v, s);
  if (adapt) {
    const int cntr = __builtin_amdgcn_readlane(v, n);
    const int rate = 3 + (cntr > 15) + (cntr > 31) + (n > 3 ? 2 : 1);
    int nv = lane < s ? v + ((32768 - v) >> rate) : v - (v >> rate);
    nv = lane == n ? cntr + (cntr < 32) : nv;
    if (lane <= n) S->cdf[off + lane] = (uint16_t)nv;
  }
  ec_code(e, lane, out, out_cap, adapt, fl, fh, s, n);
}
__device__ __forceinline__ void write_bool(EC_ARGS, int val, uint32_t f) {
  uint32_t l = e.low, r = e.rng;
  const uint32_t v = (((r >> 8) * (f >> 6)) >> 1) + 4;
  if (val) l += r - v;
  r = val ? v : r - v;
  ec_normalize(e, lane, out, out_cap, adapt, l, r);
  e.nsym++;
}
__device__ __forceinline__ void write_literal(EC_ARGS, unsigned v, int bits) {
  for (int i = bits - 1; i >= 0; i--) write_bool(e, lane, out, out_cap, adapt, (v >> i) & 1, 16384);
}
__device__ __forceinline__ int ec_finish(EC_ARGS) {
  uint32_t l = e.low;
  int c = e.cnt, s = 10;
  const uint32_t m = 0x3FFF;
  uint32_t v = ((l + m) & ~m) | (m + 1);
  s += c;
  if (s > 0) {
    uint32_t n = (1u << (c + 16)) - 1;
    do {
      put_byte(e, lane, out, out_cap, adapt, v >> (c + 16));
      v &= n;
      s -= 8;
      c -= 8;
      n >>= 8;
    } while (s > 0);
  }
  if (e.pending >= 0) raw_byte(e, lane, out, out_cap, adapt, e.pending);
  for (int i = 0; i < e.ff_run; i++) raw_byte(e, lane, out, out_cap, adapt, 0xFF);
  __syncthreads();
  const int base = e.out_pos & ~255, rem = e.out_pos - base;
  if (rem > 0 && base + 256 <= out_cap) {
    if (lane * 4 < rem) reinterpret_cast<uint32_t *>(out + base)[lane] = stage_word(lane);
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
__device__ __forceinline__ int scan_table_off(int log2n) { return log2n == 5 ? 0 : (log2n == 4 ? 1024 : (log2n == 3 ? 1280 : 1344)); }

__constant__ uint8_t c_base_ctx_off[5][5] = { { 0, 1, 6, 6, 21 }, { 1, 6, 6, 21, 21 }, { 6, 6, 21, 21, 21 }, { 6, 21, 21, 21, 21 }, { 21, 21, 21, 21, 21 } };
__constant__ uint8_t c_intra_mode_ctx[13] = { 0, 1, 2, 3, 4, 4, 4, 4, 3, 0, 1, 2, 0 };
__constant__ uint8_t c_mode_txfm[13] = { 0, 1, 2, 0, 3, 1, 2, 2, 1, 3, 1, 2, 3 };
// symbol of {DCT_DCT, ADST_DCT, DCT_ADST, ADST_ADST} in intra set 1 (7 symbols) / set 2 (5 symbols)
__constant__ uint8_t c_txsym_set1[4] = { 1, 5, 6, 4 };
__constant__ uint8_t c_txsym_set2[4] = { 1, 3, 4, 2 };

struct TileGeo {
  int sb_x, sb_y;   // luma pixel origin
  int max_x4_y, max_y4_y, max_x4_c, max_y4_c;  // frame limits in 4x4 units, superblock-local
};

// coefficients of one transform block (spec §5.11.39); x4/y4 in plane 4x4 units local to the SB.
__device__ __forceinline__ void write_coeffs(EC_ARGS, const TileGeo &tg, int plane, int log2n, int x4, int y4, int eob, int ymode,
                                             const int16_t *lv_global) {
  const int ptype = plane > 0;
  const int txs = log2n - 2;
  const int n = 1 << log2n, w4 = n >> 2;
  const int max_x4 = plane ? tg.max_x4_c : tg.max_x4_y, max_y4 = plane ? tg.max_y4_c : tg.max_y4_y;
#define a_lvl S->above_lvl[plane]
#define a_dc S->above_dc[plane]
#define l_lvl S->left_lvl[plane]
#define l_dc S->left_dc[plane]
  // all_zero context + dc sign context (lane-parallel over the w4 neighbours)
  int nb_or = 0, dsum = 0;
  if (lane < w4) {
    if (x4 + lane < max_x4) { nb_or |= (a_lvl[x4 + lane] | a_dc[x4 + lane]) ? 1 : 0; int sg = a_dc[x4 + lane]; dsum += sg == 1 ? -1 : (sg == 2 ? 1 : 0); }
    if (y4 + lane < max_y4) { nb_or |= (l_lvl[y4 + lane] | l_dc[y4 + lane]) ? 2 : 0; int sg = l_dc[y4 + lane]; dsum += sg == 1 ? -1 : (sg == 2 ? 1 : 0); }
  }
  for (int o = 4; o > 0; o >>= 1) { nb_or |= __shfl_xor(nb_or, o, 64); dsum += __shfl_xor(dsum, o, 64); }
  nb_or = uni(nb_or); dsum = uni(dsum);
  // luma: TX_MODE_LARGEST with square blocks => transform == block => ctx 0
  const int zctx = plane == 0 ? 0 : 7 + (nb_or & 1) + (nb_or >> 1);
  write_sym(e, lane, out, out_cap, adapt, eob == 0, CL::TXB_SKIP + (txs * 13 + zctx) * 3, 2);
  int cul = 0, dc_cat = 0;
  if (eob != 0) {
    // stage the block's levels in LDS (coalesced)
    {
      const uint32_t *g32 = reinterpret_cast<const uint32_t *>(lv_global);
      for (int i = lane; i < n * n / 2; i += 64) { const uint32_t w = g32[i]; S->lv[2 * i] = (int16_t)(w & 0xFFFF); S->lv[2 * i + 1] = (int16_t)(w >> 16); }
      __syncthreads();
    }
    if (plane == 0 && log2n <= 4) {
      const int tt = c_mode_txfm[ymode];
      if (log2n <= 3) write_sym(e, lane, out, out_cap, adapt, c_txsym_set1[tt], CL::TX_SET1 + ((log2n - 2) * 13 + ymode) * 8, 7);
      else write_sym(e, lane, out, out_cap, adapt, c_txsym_set2[tt], CL::TX_SET2 + ((log2n - 2) * 13 + ymode) * 6, 5);
    }
    {
      const int eob_pt = eob <= 2 ? eob : floor_log2((unsigned)(eob - 1)) + 2;
      const int base = eob_pt < 2 ? eob_pt : ((1 << (eob_pt - 2)) + 1);
      const int extra = eob - base;
      const int msz = 2 * log2n - 4;
      const int nsy = 5 + msz;
      const int eoff = CL::EOB16 + 4 * (msz * 6 + (msz * (msz - 1)) / 2);
      write_sym(e, lane, out, out_cap, adapt, eob_pt - 1, eoff + (ptype * 2 + 0) * (nsy + 1), nsy);
      if (eob_pt >= 3) {
        const int nbits = eob_pt - 2;
        write_sym(e, lane, out, out_cap, adapt, (extra >> (nbits - 1)) & 1, CL::EOB_EXTRA + ((txs * 2 + ptype) * 9 + (eob_pt - 3)) * 3, 2);
        for (int i = 1; i < nbits; i++) write_bool(e, lane, out, out_cap, adapt, (extra >> (nbits - 1 - i)) & 1, 16384);
      }
    }
    const int scan_off = scan_table_off(log2n);
#define scan(i_) S->scan[scan_off + (i_)]
    const int base_off0 = CL::COEFF_BASE + (txs * 2 + ptype) * 42 * 5;
    const int br_off0 = CL::COEFF_BR + ((txs > 3 ? 3 : txs) * 2 + ptype) * 21 * 5;
    // ---- levels, reverse scan order, 64 scan positions at a time: lane i prepares the item of
    // scan index c0 + i (level + both contexts) in registers, then the wave walks the items.
    for (int c0 = (eob - 1) & ~63; c0 >= 0; c0 -= 64) {
      const int c = c0 + lane;
      int item = 0;  // level(16) | base_ctx(8) | br_ctx(8)
      if (c < eob) {
        const int pos = scan(c);
        const int row = pos >> log2n, col = pos & (n - 1);
#define L S->lv
#define LVA(r_, c_) (((r_) < n && (c_) < n) ? iabs((int)L[((r_) << log2n) + (c_)]) : 0)
        const int a01 = LVA(row, col + 1), a10 = LVA(row + 1, col), a11 = LVA(row + 1, col + 1), a02 = LVA(row, col + 2), a20 = LVA(row + 2, col);
#undef LVA
        const int mag = imin(a01, 3) + imin(a10, 3) + imin(a11, 3) + imin(a02, 3) + imin(a20, 3);
        const int cb = pos == 0 ? 0 : imin((mag + 1) >> 1, 4) + c_base_ctx_off[imin(row, 4)][imin(col, 4)];
        int mb = imin(a01, 15) + imin(a10, 15) + imin(a11, 15);
        mb = imin((mb + 1) >> 1, 6);
        const int cbr = pos == 0 ? mb : ((row < 2 && col < 2) ? mb + 7 : mb + 14);
        item = (iabs((int)L[pos]) << 16) | (cb << 8) | cbr;
#undef L
      }
      const int top = imin(eob - 1 - c0, 63);
      for (int i = top; i >= 0; i--) {
        const int it = __builtin_amdgcn_readlane(item, uni(i));
        const int level = it >> 16;
        if (c0 + i == eob - 1) {
          const int cc = c0 + i;
          const int cctx = cc == 0 ? 0 : (cc <= (n * n) / 8 ? 1 : (cc <= (n * n) / 4 ? 2 : 3));
          write_sym(e, lane, out, out_cap, adapt, imin(level, 3) - 1, CL::COEFF_BASE_EOB + ((txs * 2 + ptype) * 4 + cctx) * 4, 3);
        } else {
          write_sym(e, lane, out, out_cap, adapt, imin(level, 3), base_off0 + ((it >> 8) & 0xFF) * 5, 4);
        }
        if (level > 2) {
          const int boff = br_off0 + (it & 0xFF) * 5;
          for (int idx = 0; idx < 4; idx++) {
            const int k3 = imin(level - 3 - idx * 3, 3);
            write_sym(e, lane, out, out_cap, adapt, k3, boff, 4);
            if (k3 < 3) break;
          }
        }
      }
    }
    // ---- signs / golomb in forward scan order; zero coefficients are skipped with a ballot
    for (int c0 = 0; c0 < eob; c0 += 64) {
      const int c = c0 + lane;
      int v = 0;
      if (c < eob) v = S->lv[scan(c)];
      unsigned long long nzmask = __ballot(v != 0);
      int lsum = iabs(v);
      for (int o = 32; o > 0; o >>= 1) lsum += __shfl_xor(lsum, o, 64);
      cul += uni(lsum);
      while (nzmask) {
        const int i = uni(__builtin_ctzll(nzmask));
        nzmask &= nzmask - 1;
        const int sv = __builtin_amdgcn_readlane(v, i);
        const int level = iabs(sv);
        if (c0 + i == 0) {
          const int dctx = dsum < 0 ? 1 : (dsum > 0 ? 2 : 0);
          write_sym(e, lane, out, out_cap, adapt, sv < 0, CL::DC_SIGN + (ptype * 3 + dctx) * 3, 2);
          dc_cat = sv < 0 ? 1 : 2;
        } else {
          write_bool(e, lane, out, out_cap, adapt, sv < 0, 16384);
        }
        if (level > 14) {
          const unsigned g = (unsigned)(level - 15) + 1;
          const int len = floor_log2(g) + 1;
          for (int k = 0; k < len - 1; k++) write_bool(e, lane, out, out_cap, adapt, 0, 16384);
          for (int k = len - 1; k >= 0; k--) write_bool(e, lane, out, out_cap, adapt, (g >> k) & 1, 16384);
        }
      }
    }
    cul = imin(cul, 63);
#undef scan
  }
  __syncthreads();
  if (lane < w4) {
    if (x4 + lane < max_x4) { a_lvl[x4 + lane] = (uint8_t)cul; a_dc[x4 + lane] = (uint8_t)dc_cat; }
    if (y4 + lane < max_y4) { l_lvl[y4 + lane] = (uint8_t)cul; l_dc[y4 + lane] = (uint8_t)dc_cat; }
  }
  __syncthreads();
#undef a_lvl
#undef a_dc
#undef l_lvl
#undef l_dc
}

__device__ __forceinline__ int icdf_prob(int off, int el) { return (el > 0 ? S->cdf[off + el - 1] : 32768) - S->cdf[off + el]; }

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
  const int sbs_per_frame = P.sb_rows * P.sb_cols;
  const int f = blockIdx.x / sbs_per_frame, sb = blockIdx.x % sbs_per_frame;
  const int sbr = sb / P.sb_cols, sbc = sb % P.sb_cols;
  const int lane = threadIdx.x;
  for (int i = lane; i < CL::TOTAL; i += 64) S->cdf[i] = cdf_init[i];
  if (lane < 64) S->cdf[CL::TOTAL + lane] = 0;
  // scan tables (scan index -> position) for n = 32, 16, 8, 4
  for (int l2 = 5; l2 >= 2; l2--) {
    const int n = 1 << l2;
    const int to = scan_table_off(l2);
    for (int p = lane; p < n * n; p += 64) S->scan[to + scan_index(p >> l2, p & (n - 1), n)] = (uint16_t)p;
  }
  {
    const Av1miBlkInfo *info = blk + (size_t)f * P.b8_rows * P.b8_cols + (size_t)(sbr * 8) * P.b8_cols + sbc * 8;
    const int r = lane >> 3, c = lane & 7;
    Av1miBlkInfo bi = {};
    if (sbr * 8 + r < P.b8_rows && sbc * 8 + c < P.b8_cols) bi = info[r * P.b8_cols + c];
    S->info[lane] = bi;
    if (lane < 48) { (&S->above_lvl[0][0])[lane] = 0; (&S->above_dc[0][0])[lane] = 0; (&S->left_lvl[0][0])[lane] = 0; (&S->left_dc[0][0])[lane] = 0; }
  }
  __syncthreads();
  Ec e;
  e.low = 0; e.rng = 0x8000; e.cnt = -9; e.pending = -1; e.ff_run = 0; e.out_pos = 0; e.nsym = 0;
  uint8_t *const out = slots + (size_t)blockIdx.x * P.tile_slot_bytes;
  const int out_cap = P.tile_slot_bytes;
  const int adapt = !P.disable_cdf_update;
  TileGeo tg;
  tg.sb_x = sbc * 64; tg.sb_y = sbr * 64;
  tg.max_x4_y = P.mi_cols - sbc * 16; tg.max_y4_y = P.mi_rows - sbr * 16;
  tg.max_x4_c = (P.mi_cols >> 1) - sbc * 8; tg.max_y4_c = (P.mi_rows >> 1) - sbr * 8;
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
      const int b8x = bx >> 3, b8y = by >> 3;
      // ---- partition symbol (spec §5.11.4)
      {
        const int half = n >> 1;
        const bool has_rows = tg.sb_y + by + half < P.height, has_cols = tg.sb_x + bx + half < P.width;
        const int above = by > 0 && uni(S->info[(b8y - 1) * 8 + b8x].bsl) < bsl;
        const int left = bx > 0 && uni(S->info[b8y * 8 + b8x - 1].bsl) < bsl;
        const int off = CL::PARTITION + ((bsl - 3) * 4 + left * 2 + above) * 11;
        if (has_rows && has_cols) {
          write_sym(e, lane, out, out_cap, adapt, split ? 3 : 0, off, bsl == 3 ? 4 : 10);
        } else if (has_cols) {
          int p = icdf_prob(off, 2) + icdf_prob(off, 3);
          if (bsl != 3) p += icdf_prob(off, 4) + icdf_prob(off, 6) + icdf_prob(off, 7) + icdf_prob(off, 9);
          write_bool(e, lane, out, out_cap, adapt, 1, (uint32_t)uni(p));
        } else if (has_rows) {
          int p = icdf_prob(off, 1) + icdf_prob(off, 3);
          if (bsl != 3) p += icdf_prob(off, 4) + icdf_prob(off, 5) + icdf_prob(off, 6) + icdf_prob(off, 8);
          write_bool(e, lane, out, out_cap, adapt, 1, (uint32_t)uni(p));
        }
      }
      if (split) continue;
      // ---- leaf block: intra_frame_mode_info + residual
      {
        const Av1miBlkInfo bi = S->info[b8y * 8 + b8x];
        const int ymode = uni(bi.ymode), skip = uni(bi.skip);
        const int eob0 = uni(bi.eob[0]), eob1 = uni(bi.eob[1]), eob2 = uni(bi.eob[2]);
        const int avail_u = by > 0, avail_l = bx > 0;
        int sctx = 0;
        if (avail_u) sctx += uni(S->info[(b8y - 1) * 8 + b8x].skip);
        if (avail_l) sctx += uni(S->info[b8y * 8 + b8x - 1].skip);
        write_sym(e, lane, out, out_cap, adapt, skip, CL::SKIP + sctx * 3, 2);
        const int am = uni(c_intra_mode_ctx[avail_u ? S->info[(b8y - 1) * 8 + b8x].ymode : 0]);
        const int lm = uni(c_intra_mode_ctx[avail_l ? S->info[b8y * 8 + b8x - 1].ymode : 0]);
        write_sym(e, lane, out, out_cap, adapt, ymode, CL::KF_Y_MODE + (am * 5 + lm) * 14, 13);
        if (ymode >= 1 && ymode <= 8) write_sym(e, lane, out, out_cap, adapt, 3, CL::ANGLE_DELTA + (ymode - 1) * 8, 7);
        const int uvmode = ymode;
        const int cfl_allowed = n <= 32;
        write_sym(e, lane, out, out_cap, adapt, uvmode, CL::UV_MODE + (cfl_allowed * 13 + ymode) * 15, cfl_allowed ? 14 : 13);
        if (uvmode >= 1 && uvmode <= 8) write_sym(e, lane, out, out_cap, adapt, 3, CL::ANGLE_DELTA + (uvmode - 1) * 8, 7);
        const int w4 = n >> 2, w4c = imax(w4 >> 1, 1);
        const int log2c = bsl - 1;
        if (skip) {
          __syncthreads();
          if (lane < w4) { S->above_lvl[0][(bx >> 2) + lane] = 0; S->above_dc[0][(bx >> 2) + lane] = 0; S->left_lvl[0][(by >> 2) + lane] = 0; S->left_dc[0][(by >> 2) + lane] = 0; }
          if (lane < w4c) {
            for (int pl = 1; pl < 3; pl++) { S->above_lvl[pl][(bx >> 3) + lane] = 0; S->above_dc[pl][(bx >> 3) + lane] = 0; S->left_lvl[pl][(by >> 3) + lane] = 0; S->left_dc[pl][(by >> 3) + lane] = 0; }
          }
          __syncthreads();
        } else {
          for (int pl = 0; pl < 3; pl++) {
            const int l2 = pl ? log2c : bsl;
            const int16_t *lvp = pl == 0 ? sb_levels + by * 64 + bx * n : sb_levels + 4096 + (pl - 1) * 1024 + (by >> 1) * 32 + (bx >> 1) * (n >> 1);
            write_coeffs(e, lane, out, out_cap, adapt, tg, pl, l2, pl ? bx >> 3 : bx >> 2, pl ? by >> 3 : by >> 2,
                         pl == 0 ? eob0 : (pl == 1 ? eob1 : eob2), ymode, lvp);
          }
        }
      }
      break;
    }
  }
  const int nbytes = ec_finish(e, lane, out, out_cap, adapt);
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
