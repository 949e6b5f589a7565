// entropy_kernel.hip - AV1 tile entropy coding in two kernels.
//
// Replaces the entropy-coding stage of the external SVT-AV1 worker behind `run_av1an`
// (/root/reference/crates/daemon/src/encode/av1an.rs:126-139).
// Bitstream syntax written: AV1 spec §5.11.4 decode_partition, §5.11.7 intra_frame_mode_info,
// §5.11.39 coeffs, §5.11.47 transform_type, contexts §8.3.2, symbol coder §8.2 (mirror).
//
// Why two kernels.  A tile's symbol sequence is serial twice over: adaptive CDFs
// and the range coder.  Run one tile per WAVE and the whole serial chain sits on the scalar unit
// (one per CU): measured 26 ms for 60 1080p frames (profiles/r01_a).  So:
//
//   K3 symbolize_tile_kernel   one wave per tile.  Everything that is parallel inside a tile:
//        coefficient contexts, the exact coding ORDER (wave prefix sums over per-coefficient symbol
//        counts) and the ~60 wide-alphabet symbols per tile (partition, modes, eob, ... adapted
//        cooperatively, lane j = CDF entry j).  Output: a flat stream of 32-bit entries per tile,
//        either RESOLVED (fl>>6, fh>>6, N-s: nothing left but range coding; literals are this kind
//        too) or NARROW (slot, symbol) for the 4-symbol coefficient CDFs that still must adapt.
//   K4 rangecode{2,4}_tiles_kernel  one LANE per tile, 64 tiles per wave.  Each lane keeps its tile's
//        adaptive coeff_base / coeff_br rows (2 x 63 rows x 8 B) in LDS laid out [slot][lane] -
//        bank = 2*lane mod 64 whatever the slot, i.e. conflict-free across lanes - plus its own
//        range coder in VGPRs, and walks its stream.  The serial chain now runs 64-wide on the
//        vector units of every CU instead of on 256 scalar units.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "av1mi_dev.h"

namespace {

typedef Av1miCdfLayout CL;

// ---- stream entry encoding ------------------------------------------------------------------
// narrow  : bit31 = 0 | slot (bits 2..10) | symbol (bits 0..1)
// resolved: bit31 = 1 | fl>>6 (bits 14..23, 512 = "s == 0") | fh>>6 (bits 4..13) | N - s (bits 0..3)
#define ENT_RESOLVED(fl6, fh6, ns) (0x80000000u | ((uint32_t)(fl6) << 14) | ((uint32_t)(fh6) << 4) | (uint32_t)(ns))
#define ENT_NARROW(slot, sym) (((uint32_t)(slot) << 2) | (uint32_t)(sym))
#define ENT_LITERAL(bit) ((bit) ? ENT_RESOLVED(256, 0, 0) : ENT_RESOLVED(512, 256, 1))
#define SLOTS_PER_COMBO 63   // 42 coeff_base + 21 coeff_br rows of one (tx size, plane type)
#define MAX_COMBOS 2

// The coefficient (narrow) rows are 57 % of the CDF set; a regular tile in adaptive mode never
// touches them here (they adapt per lane in K4), so only the FULL kernel variant holds them in LDS.
__shared__ uint16_t g_cdf_narrow[CL::INTRA_TOTAL - CL::COEFF_BASE + 64];
__shared__ uint16_t g_full_list[5 * 64];   // FULL variant: the (row, symbol) pairs of 64 coefficients in coding order
// inter-frame CDFs and the motion vector candidate list: only the INTER instantiations reference (and allocate) them
__shared__ uint16_t g_cdf_inter[CL::TOTAL - CL::INTER_BASE + 64];
struct InterLds {
  int stk_row[10], stk_col[10], stk_w[10];  // candidate list of the current block (spec §7.10.2)
  uint8_t newmv[256];                        // per 8x8 unit of the tile: its block was coded as NEWMV
};
__shared__ InterLds g_inter;
// The wide rows' layout of the REGULAR variants: a regular tile has one luma and one chroma transform size, so the tables the full
// layout (Av1miCdfLayout) keeps per size - a third of the wide rows - shrink to one slice per plane type: 3.2 instead of 4.9 KB, which
// takes the key-frame variant from 5 to 6 waves per SIMD and the inter-frame variant from 4 to 5.  PARTITION .. SKIP keep their offsets.
struct CC {
  enum {
    LEAD = CL::TX_SET1,
    TXSET = LEAD,                        // the luma size's slice of TX_SET1 ([13][8]) or TX_SET2 ([13][6]), if it has one
    TXB_SKIP = TXSET + 13 * 8,           // [2 plane types][13][3]
    EOB = TXB_SKIP + 2 * 13 * 3,         // [2][12]: the row (context 0) of the plane type's size
    EOB_EXTRA = EOB + 2 * 12,            // [2][9][3]
    DC_SIGN = EOB_EXTRA + 2 * 9 * 3,     // [2][3][3] as in the full layout
    COEFF_BASE_EOB = DC_SIGN + 18,       // [2][4][4]
    USE_WIENER = COEFF_BASE_EOB + 2 * 16, RESTORE_SW = USE_WIENER + 3, CFL_SIGN = RESTORE_SW + 4, CFL_ALPHA = CFL_SIGN + 9,
    TOTAL = CFL_ALPHA + 6 * 17
  };
};
static_assert(CL::COEFF_BASE - CL::USE_WIENER == CC::TOTAL - CC::USE_WIENER, "the trailing tables are copied as one piece");
__shared__ uint16_t g_cdfw_full[CL::COEFF_BASE + 18];   // wide rows; + 18: whole-row reads by 17 lanes may run past the last row
__shared__ uint16_t g_cdfw_compact[CC::TOTAL + 18];
struct SymLds {
  alignas(16) int16_t lv[32 * 32];
  uint8_t left_lvl[3][16], left_dc[3][16];   // per superblock row of the tile
};
__shared__ SymLds g_sym;
#define S (&g_sym)
// per-tile state whose size depends on the tile size (TSB x TSB superblocks): block info of the tile's 8x8 units and the
// above contexts; only the two-superblock instantiations reference (and allocate) the larger object
template <int TSB>
struct TileLds {
  Av1miBlkInfo info[64 * TSB * TSB];
  uint8_t above_lvl[3][16 * TSB], above_dc[3][16 * TSB];
};
__shared__ TileLds<1> g_tile1;
__shared__ TileLds<2> g_tile2;
template <int TSB> struct TileSel;
template <> struct TileSel<1> { static __device__ __forceinline__ TileLds<1> &get() { return g_tile1; } };
template <> struct TileSel<2> { static __device__ __forceinline__ TileLds<2> &get() { return g_tile2; } };
#define TI (TileSel<TSB>::get())
#define TW (8 * TSB)   // tile width in 8x8 units

struct Sym {
  uint16_t *cdf;     // the variant's wide rows (LDS)
  uint32_t *out;     // this tile's stream (global)
  int pos;           // entries written (uniform)
  int cap;
  int combo0, combo1;  // (txs * 2 + ptype) owning slot range 0 / 1, -1 = free
};

__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int iabs(int v) { return v < 0 ? -v : v; }
__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }
__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }
__device__ __forceinline__ int floor_log2(unsigned v) { return 31 - __builtin_clz(v); }

__device__ __forceinline__ void emit1(Sym &y, int lane, uint32_t ent) {
  if (lane == 0 && y.pos < y.cap) y.out[y.pos] = ent;
  y.pos++;
}

// Wide-alphabet / rare symbol: adapt the n-symbol row at LDS offset `off` cooperatively (lane j =
// entry j) and emit the RESOLVED entry.
template <int ARR = 0>  // 0: y.cdf (key-frame syntax), 1: g_cdf_inter (offsets relative to CL::INTER_BASE)
__device__ __forceinline__ void sym_wide(Sym &y, int lane, int adapt, int s, int off, int n) {
  s = uni(s); off = uni(off) - (ARR ? CL::INTER_BASE : 0); n = uni(n);
  const int v = ARR ? g_cdf_inter[off + (lane < 17 ? lane : 16)] : y.cdf[off + (lane < 17 ? lane : 16)];
  const uint32_t fl = s > 0 ? (uint32_t)__builtin_amdgcn_readlane(v, s - 1) : 32768u;
  const uint32_t fh = (uint32_t)__builtin_amdgcn_readlane(v, s);
  if (adapt) {
    const int cntr = __builtin_amdgcn_readlane(v, n);
    const int rate = 3 + (cntr > 15) + (cntr > 31) + (n > 3 ? 2 : 1);
    int nv = lane < s ? v + ((32768 - v) >> rate) : v - (v >> rate);
    nv = lane == n ? cntr + (cntr < 32) : nv;
    if (lane <= n) { if (ARR) g_cdf_inter[off + lane] = (uint16_t)nv; else y.cdf[off + lane] = (uint16_t)nv; }
  }
  emit1(y, lane, ENT_RESOLVED(fl >> 6, fh >> 6, n - 1 - s));
}
// same for a 4-symbol coefficient row kept in the narrow LDS array (FULL variant only)
__device__ __forceinline__ void sym_narrow_resolved(Sym &y, int lane, int adapt, int s, int off) {
  s = uni(s); off = uni(off) - CL::COEFF_BASE;
  const int v = g_cdf_narrow[off + (lane < 5 ? lane : 4)];
  const uint32_t fl = s > 0 ? (uint32_t)__builtin_amdgcn_readlane(v, s - 1) : 32768u;
  const uint32_t fh = (uint32_t)__builtin_amdgcn_readlane(v, s);
  if (adapt) {
    const int cntr = __builtin_amdgcn_readlane(v, 4);
    const int rate = 5 + (cntr > 15) + (cntr > 31);
    int nv = lane < s ? v + ((32768 - v) >> rate) : v - (v >> rate);
    nv = lane == 4 ? cntr + (cntr < 32) : nv;
    if (lane <= 4) g_cdf_narrow[off + lane] = (uint16_t)nv;
  }
  emit1(y, lane, ENT_RESOLVED(fl >> 6, fh >> 6, 3 - s));
}
__device__ __forceinline__ void sym_bool(Sym &y, int lane, int val, uint32_t f) {
  // P(val == 1) = f / 32768, no adaptation
  emit1(y, lane, val ? ENT_RESOLVED(f >> 6, 0, 0) : ENT_RESOLVED(512, f >> 6, 1));
}

__device__ __forceinline__ int scan_index(int row, int col, int n) {
  int d = row + col;
  int before = d < n ? (d * (d + 1)) >> 1 : n * n - (((2 * n - 1 - d) * (2 * n - d)) >> 1);
  int lo = d - (n - 1) > 0 ? d - (n - 1) : 0;
  return before + ((d & 1) ? row - lo : col - lo);
}
// position (row << log2n | col) of scan index c in the default zig-zag of an n x n block: the inverse of scan_index, from a table
// in device memory built at compile time (2.7 KB for 4x4 .. 32x32; L1 / L2 resident).  In LDS the table cost a quarter of the
// kernel's occupancy, computed (anti-diagonal by a float square root + integer fix-up) it was 30 instructions per lane and pass
// of a kernel that is bound by instruction issue.
struct ScanTab { uint16_t v[16 + 64 + 256 + 1024]; };
constexpr ScanTab make_scan_tab() {
  ScanTab t{};
  int off = 0;
  for (int l = 2; l <= 5; l++) {
    const int n = 1 << l;
    for (int row = 0; row < n; row++)
      for (int col = 0; col < n; col++) {
        const int d = row + col;
        const int before = d < n ? (d * (d + 1)) >> 1 : n * n - (((2 * n - 1 - d) * (2 * n - d)) >> 1);
        const int lo = d - (n - 1) > 0 ? d - (n - 1) : 0;
        t.v[off + before + ((d & 1) ? row - lo : col - lo)] = (uint16_t)((row << l) | col);
      }
    off += n * n;
  }
  return t;
}
__device__ const ScanTab c_scan_tab = make_scan_tab();
__device__ __forceinline__ int scan_pos(int c, int log2n) {
  const int off = log2n == 2 ? 0 : (log2n == 3 ? 16 : (log2n == 4 ? 80 : 336));
  return c_scan_tab.v[off + c];
}

// Small per-block tables as literals (four / two bits per entry): as `__constant__` byte arrays each read was a per-lane global load
// the wave waited for on the spot (the scalar unit has no byte loads) - a cache round trip per block and table on a serial chain.
// Intra_Mode_Context { 0, 1, 2, 3, 4, 4, 4, 4, 3, 0, 1, 2, 0 }, Mode_To_Txfm { 0, 1, 2, 0, 3, 1, 2, 2, 1, 3, 1, 2, 3 }
__device__ __forceinline__ int intra_mode_ctx(int mode) { return (int)((0x210344443210ull >> (4 * mode)) & 15u); }
__device__ __forceinline__ int mode_txfm(int mode) { return (int)((0x39da724u >> (2 * mode)) & 3u); }
// symbol of {DCT_DCT, ADST_DCT, DCT_ADST, ADST_ADST} in intra set 1 (7 symbols) / set 2 (5 symbols)
__device__ __forceinline__ int txsym_set1(int tt) { return (0x4651 >> (4 * tt)) & 15; }   // { 1, 5, 6, 4 }
__device__ __forceinline__ int txsym_set2(int tt) { return (0x2431 >> (4 * tt)) & 15; }   // { 1, 3, 4, 2 }

struct TileGeo {
  int tox, toy;     // origin of the current superblock inside the tile, luma pixels (0 or 64)
  int sb_x, sb_y;   // luma pixel origin
  int max_x4_y, max_y4_y, max_x4_c, max_y4_c;  // frame limits in 4x4 units, superblock-local
};

// exclusive prefix sum over the wave with DPP row shifts and broadcasts (seven additions; the shuffle form went through LDS six times)
__device__ __forceinline__ int wave_excl_scan(int v, int lane, int *total) {
  int x = v;
  x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, false);   // row_shr:1
  x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, false);   // row_shr:2
  x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, false);   // row_shr:4
  x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, false);   // row_shr:8   (inclusive within rows of 16)
  x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false);   // row_bcast:15 into rows 1 and 3
  x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false);   // row_bcast:31 into rows 2 and 3
  *total = __builtin_amdgcn_readlane(x, 63);
  (void)lane;
  return x - v;
}

// coefficients of one transform block (spec §5.11.39); x4/y4 in plane 4x4 units local to the SB.
template <bool FULL, int TSB>
__device__ __forceinline__ void sym_coeffs(Sym &y, const int lane, const int adapt, const TileGeo &tg, int plane, int log2n, int x4,
                                           int y4, int eob, int ymode, int is_inter, const int16_t *lv_global) {
  const int idtx = ymode >> 8;   // bit 8 of the mode argument: the luma block uses the identity transform
  ymode &= 255;
  const int ptype = plane > 0;
  const int txs = log2n - 2;
  const int w4 = (1 << log2n) >> 2;
  // a 64-point transform codes only its 32x32 low-frequency corner: contexts of the size class TX_64X64 (txs 4), scan, eob and
  // neighbourhoods of a 32x32 area
  const int bwl = log2n > 5 ? 5 : log2n;
  const int n = 1 << bwl;
  const int max_x4 = plane ? tg.max_x4_c : tg.max_x4_y, max_y4 = plane ? tg.max_y4_c : tg.max_y4_y;
  const int aoff = (plane ? tg.tox >> 3 : tg.tox >> 2);  // above contexts are indexed by tile-local 4x4 column
#define a_lvl (TI.above_lvl[plane] + aoff)
#define a_dc (TI.above_dc[plane] + aoff)
#define l_lvl S->left_lvl[plane]
#define l_dc S->left_dc[plane]
  // the block's levels, 16 bytes (8 levels) per lane and step: requested here, needed only after the block's first symbols (by then
  // they have arrived - issued where they are copied to LDS the wave waited a memory round trip per transform block); unconditional,
  // index clamped (a load under a condition is waited for on the spot)
  typedef unsigned int u4 __attribute__((ext_vector_type(4)));
  u4 lv_pre[2];
  {
    const u4 *g128 = reinterpret_cast<const u4 *>(lv_global);
    const int last = n * n / 8 - 1;
    lv_pre[0] = g128[imin(lane, last)];
    lv_pre[1] = g128[imin(lane + 64, last)];
  }
  int nb_or = 0, dsum = 0;
  if (lane < w4) {
    if (x4 + lane < max_x4) { nb_or |= (a_lvl[x4 + lane] | a_dc[x4 + lane]) ? 1 : 0; int sg = a_dc[x4 + lane]; dsum += sg == 1 ? -1 : (sg == 2 ? 1 : 0); }
    if (y4 + lane < max_y4) { nb_or |= (l_lvl[y4 + lane] | l_dc[y4 + lane]) ? 2 : 0; int sg = l_dc[y4 + lane]; dsum += sg == 1 ? -1 : (sg == 2 ? 1 : 0); }
  }
  // w4 <= 16 lanes hold values: an all-reduce inside the first row of 16 lanes by DPP rotations (row_ror 8, 4, 2, 1)
  nb_or |= __builtin_amdgcn_update_dpp(0, nb_or, 0x128, 0xF, 0xF, false); dsum += __builtin_amdgcn_update_dpp(0, dsum, 0x128, 0xF, 0xF, false);
  nb_or |= __builtin_amdgcn_update_dpp(0, nb_or, 0x124, 0xF, 0xF, false); dsum += __builtin_amdgcn_update_dpp(0, dsum, 0x124, 0xF, 0xF, false);
  nb_or |= __builtin_amdgcn_update_dpp(0, nb_or, 0x122, 0xF, 0xF, false); dsum += __builtin_amdgcn_update_dpp(0, dsum, 0x122, 0xF, 0xF, false);
  nb_or |= __builtin_amdgcn_update_dpp(0, nb_or, 0x121, 0xF, 0xF, false); dsum += __builtin_amdgcn_update_dpp(0, dsum, 0x121, 0xF, 0xF, false);
  nb_or = uni(nb_or); dsum = uni(dsum);
  // luma: TX_MODE_LARGEST with square blocks => transform == block => ctx 0
  const int zctx = plane == 0 ? 0 : 7 + (nb_or & 1) + (nb_or >> 1);
  sym_wide(y, lane, adapt, eob == 0, FULL ? CL::TXB_SKIP + (txs * 13 + zctx) * 3 : CC::TXB_SKIP + (ptype * 13 + zctx) * 3, 2);
  int cul = 0, dc_cat = 0;
  if (eob != 0) {
    {
      // ... into the LDS copy (same row-major layout)
      u4 *l128 = reinterpret_cast<u4 *>(S->lv);
      if (lane < n * n / 8) l128[lane] = lv_pre[0];
      if (lane + 64 < n * n / 8) l128[lane + 64] = lv_pre[1];
      __syncthreads();
    }
    if (plane == 0 && is_inter) {
      // inter_tx_type: every inter block is DCT_DCT = symbol 7 / 3 / 1 of TX_SET_INTER_1 / _2 / _3 (§5.11.47)
      if (log2n <= 3) sym_wide<1>(y, lane, adapt, 7, CL::INTER_TX1 + (log2n - 2) * 17, 16);
      else if (log2n == 4) sym_wide<1>(y, lane, adapt, 3, CL::INTER_TX2, 12);
      else if (log2n == 5) sym_wide<1>(y, lane, adapt, 1, CL::INTER_TX3 + 3 * 3, 2);
      // (TX_64X64: the transform set is DCT only, nothing is coded)
    } else if (plane == 0 && log2n <= 4) {
      // intra_tx_type: the mode's default type, or IDTX (symbol 0 of both intra sets) when the reconstruction chose it
      const int tt = mode_txfm(ymode);
      if (log2n <= 3) sym_wide(y, lane, adapt, idtx ? 0 : txsym_set1(tt), FULL ? CL::TX_SET1 + ((log2n - 2) * 13 + ymode) * 8 : CC::TXSET + ymode * 8, 7);
      else sym_wide(y, lane, adapt, idtx ? 0 : txsym_set2(tt), FULL ? CL::TX_SET2 + ((log2n - 2) * 13 + ymode) * 6 : CC::TXSET + ymode * 6, 5);
    }
    {
      const int eob_pt = eob <= 2 ? eob : floor_log2((unsigned)(eob - 1)) + 2;
      const int base = eob_pt < 2 ? eob_pt : ((1 << (eob_pt - 2)) + 1);
      const int extra = eob - base;
      const int msz = 2 * bwl - 4;
      const int nsy = 5 + msz;
      const int eoff = CL::EOB16 + 4 * (msz * 6 + (msz * (msz - 1)) / 2);
      sym_wide(y, lane, adapt, eob_pt - 1, FULL ? eoff + (ptype * 2 + 0) * (nsy + 1) : CC::EOB + ptype * 12, nsy);
      if (eob_pt >= 3) {
        const int nbits = eob_pt - 2;
        sym_wide(y, lane, adapt, (extra >> (nbits - 1)) & 1, FULL ? CL::EOB_EXTRA + ((txs * 2 + ptype) * 9 + (eob_pt - 3)) * 3 : CC::EOB_EXTRA + (ptype * 9 + (eob_pt - 3)) * 3, 2);
        for (int i = 1; i < nbits; i++) emit1(y, lane, ENT_LITERAL((extra >> (nbits - 1 - i)) & 1));
      }
    }
#define scan(i_) scan_pos((i_), bwl)
    // slot range of this (tx size, plane type): the first MAX_COMBOS combinations met in a tile
    // adapt per lane in K4; any further combination is resolved here (cooperatively, slowly).
    const int combo = txs * 2 + ptype;
    int slot0 = -1;
    if (adapt) {
      if (y.combo0 == combo) slot0 = 0;
      else if (y.combo0 < 0) { y.combo0 = combo; slot0 = 0; }
      else if (y.combo1 == combo) slot0 = SLOTS_PER_COMBO;
      else if (y.combo1 < 0) { y.combo1 = combo; slot0 = SLOTS_PER_COMBO; }
    }
    const int base_off0 = CL::COEFF_BASE + (txs * 2 + ptype) * 42 * 5;
    const int br_off0 = CL::COEFF_BR + ((txs > 3 ? 3 : txs) * 2 + ptype) * 21 * 5;
    // Scan positions come from a table in device memory: the load for the NEXT 64 coefficients is issued before the current ones are
    // worked on, with a clamped index instead of a condition (a load under `if (c >= 0)` is waited for on the spot, and on this serial
    // chain that was a cache round trip per 64 coefficients and pass).
    int pos_nx = scan(imax(eob - 1 - lane, 0));
    // ---- last coefficient: coeff_base_eob (3 symbols, resolved here)
    {
      const int cc = eob - 1;
      const int lvl_last = uni(iabs((int)S->lv[__builtin_amdgcn_readfirstlane(pos_nx)]));   // (lane 0 holds scan index eob - 1)
      const int cctx = cc == 0 ? 0 : (cc <= (n * n) / 8 ? 1 : (cc <= (n * n) / 4 ? 2 : 3));
      sym_wide(y, lane, adapt, imin(lvl_last, 3) - 1, FULL ? CL::COEFF_BASE_EOB + ((txs * 2 + ptype) * 4 + cctx) * 4 : CC::COEFF_BASE_EOB + (ptype * 4 + cctx) * 4, 3);
    }
    // ---- all coefficients in reverse scan order, 64 at a time: lane i owns scan index c_hi - i
    for (int c_hi = eob - 1; c_hi >= 0; c_hi -= 64) {
      const int c = c_hi - lane;
      int level = 0, cb = 0, cbr = 0, cnt = 0;
      const int pos = pos_nx;
      pos_nx = scan(imax(c - 64, 0));
      if (c >= 0) {
        const int row = pos >> bwl, col = pos & (n - 1);
#define LVA(r_, c_) (((r_) < n && (c_) < n) ? iabs((int)S->lv[((r_) << bwl) + (c_)]) : 0)
        const int a01 = LVA(row, col + 1), a10 = LVA(row + 1, col), a11 = LVA(row + 1, col + 1), a02 = LVA(row, col + 2), a20 = LVA(row + 2, col);
#undef LVA
        const int mag = imin(a01, 3) + imin(a10, 3) + imin(a11, 3) + imin(a02, 3) + imin(a20, 3);
        // Coeff_Base_Ctx_Offset of a square block depends on row + col only: 0, 1, 6, 6, 21, 21 ... (a table load per lane otherwise)
        const int rc = row + col;
        cb = pos == 0 ? 0 : imin((mag + 1) >> 1, 4) + (rc < 2 ? rc : (rc < 4 ? 6 : 21));
        int mb = imin(a01, 15) + imin(a10, 15) + imin(a11, 15);
        mb = imin((mb + 1) >> 1, 6);
        cbr = pos == 0 ? mb : ((row < 2 && col < 2) ? mb + 7 : mb + 14);
        level = iabs((int)S->lv[pos]);
        // symbols of this coefficient: base (except the last coefficient) + 0..4 range symbols
        const int nbr = level > 2 ? imin((level - 3) / 3 + 1, 4) : 0;
        cnt = (c != eob - 1) + nbr;
      }
      if (slot0 >= 0) {
        int total;
        const int off = wave_excl_scan(cnt, lane, &total);
        if (c >= 0 && y.pos + off + cnt <= y.cap) {
          uint32_t *o = y.out + y.pos + off;
          if (c != eob - 1) *o++ = ENT_NARROW(slot0 + cb, imin(level, 3));
          if (level > 2) {
            for (int idx = 0; idx < 4; idx++) {
              const int k3 = imin(level - 3 - idx * 3, 3);
              *o++ = ENT_NARROW(slot0 + 42 + cbr, k3);
              if (k3 < 3) break;
            }
          }
        }
        y.pos += total;
      } else if (FULL) {
        // Resolved path (static CDFs, or a third size class in an edge tile): the symbols adapt here, one after the other - a few
        // waves with a long serial chain, which last as long as the regular variant's thousands beside them.  So the chain is kept short:
        // the batch's (row, symbol) pairs go to an LDS list in coding order, lane-parallel like the slot path above; the serial loop
        // then takes 64 of them into a register, reads the NEXT pair's CDF row before it writes the current one back (forwarded when both
        // name the same row), builds the entries in a register and stores the 64 with one instruction.
        if constexpr (FULL) {
          int total;
          const int off = wave_excl_scan(cnt, lane, &total);
          if (c >= 0) {
            uint16_t *o = g_full_list + off;
            if (c != eob - 1) *o++ = (uint16_t)((cb << 2) | imin(level, 3));
            if (level > 2) {
              for (int idx = 0; idx < 4; idx++) {
                const int k3 = imin(level - 3 - idx * 3, 3);
                *o++ = (uint16_t)(((42 + cbr) << 2) | k3);
                if (k3 < 3) break;
              }
            }
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
          const int row0 = base_off0 - CL::COEFF_BASE, row1 = br_off0 - CL::COEFF_BASE - 42 * 5;   // g_cdf_narrow offset of row r: r < 42 ? row0 + 5 r : row1 + 5 r
          const int l5 = lane < 5 ? lane : 4;
          for (int k0 = 0; k0 < total; k0 += 64) {
            const int n_k = imin(64, total - k0);
            const int ents = k0 + lane < total ? (int)g_full_list[k0 + lane] : 0;
            int e = __builtin_amdgcn_readlane(ents, 0);
            int roff = ((e >> 2) < 42 ? row0 : row1) + (e >> 2) * 5;
            int v = g_cdf_narrow[roff + l5];
            uint32_t outv = 0;
            for (int k = 0; k < n_k; k++) {
              const int sy = e & 3;
              // the next pair and its row (k + 1 == n_k: lane 63's or a stale pair - a valid row either way, and unused)
              const int e_nx = __builtin_amdgcn_readlane(ents, (k + 1) & 63);
              const int roff_nx = ((e_nx >> 2) < 42 ? row0 : row1) + (e_nx >> 2) * 5;
              const int v_nx = g_cdf_narrow[roff_nx + l5];
              const uint32_t fl = sy > 0 ? (uint32_t)__builtin_amdgcn_readlane(v, sy - 1) : 32768u;
              const uint32_t fh = (uint32_t)__builtin_amdgcn_readlane(v, sy);
              int nv = v;
              if (adapt) {
                const int cntr = __builtin_amdgcn_readlane(v, 4);
                const int rate = 5 + (cntr > 15) + (cntr > 31);
                nv = lane < sy ? v + ((32768 - v) >> rate) : v - (v >> rate);
                nv = lane == 4 ? cntr + (cntr < 32) : nv;
                if (lane <= 4) g_cdf_narrow[roff + lane] = (uint16_t)nv;
              }
              outv = lane == k ? ENT_RESOLVED(fl >> 6, fh >> 6, 3 - sy) : outv;
              v = roff_nx == roff ? nv : v_nx;
              roff = roff_nx; e = e_nx;
            }
            if (lane < n_k && y.pos + lane < y.cap) y.out[y.pos + lane] = outv;
            y.pos += n_k;
          }
        }
      }
    }
    // ---- signs / golomb in forward scan order (all literals except the DC sign)
    int fpos_nx = scan(imin(lane, eob - 1));
    for (int c0 = 0; c0 < eob; c0 += 64) {
      const int c = c0 + lane;
      int v = 0;
      const int fpos = fpos_nx;
      fpos_nx = scan(imin(c + 64, eob - 1));
      if (c < eob) v = S->lv[fpos];
      const int level = iabs(v);
      { int lsum; (void)wave_excl_scan(level, lane, &lsum); cul += lsum; }   // (the DPP prefix sum's total)
      if (c0 == 0) {
        const int v0 = __builtin_amdgcn_readlane(v, 0);
        if (v0 != 0) {
          const int dctx = dsum < 0 ? 1 : (dsum > 0 ? 2 : 0);
          sym_wide(y, lane, adapt, v0 < 0, (FULL ? CL::DC_SIGN : CC::DC_SIGN) + (ptype * 3 + dctx) * 3, 2);
          dc_cat = v0 < 0 ? 1 : 2;
          const int l0 = iabs(v0);
          if (l0 > 14) {  // its golomb tail (rare) right behind it
            const unsigned g = (unsigned)(l0 - 15) + 1;
            const int len = floor_log2(g) + 1;
            for (int k = 0; k < len - 1; k++) emit1(y, lane, ENT_LITERAL(0));
            for (int k = len - 1; k >= 0; k--) emit1(y, lane, ENT_LITERAL((g >> k) & 1));
          }
        }
      }
      const bool mine = level != 0 && c != 0;
      const unsigned g = level > 14 ? (unsigned)(level - 15) + 1 : 0;
      const int glen = g ? floor_log2(g) + 1 : 0;
      const int cnt = mine ? 1 + (g ? 2 * glen - 1 : 0) : 0;
      int total;
      const int off = wave_excl_scan(cnt, lane, &total);
      if (mine && y.pos + off + cnt <= y.cap) {
        uint32_t *o = y.out + y.pos + off;
        *o++ = ENT_LITERAL(v < 0);
        if (g) {
          for (int k = 0; k < glen - 1; k++) *o++ = ENT_LITERAL(0);
          for (int k = glen - 1; k >= 0; k--) *o++ = ENT_LITERAL((g >> k) & 1);
        }
      }
      y.pos += total;
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

// ---- motion vector prediction (spec §7.10.2) inside a one-superblock tile ----------------------
// Everything is uniform across the wave (scalar control flow; the list lives in LDS).  Coordinates are tile-local
// 4x4 units (0..15); block info is per 8x8 unit; only LAST_FRAME is ever referenced, identity global motion,
// no temporal candidates.
struct MvScan {
  int num, new_count, found;
  int max_r4, max_c4;   // tile-local limits (frame edge)
  int cur_z;            // Morton index of the current block's origin unit: units with a smaller index are decoded
};
__device__ __forceinline__ int morton8(int ux, int uy) {
  return (ux & 1) | ((uy & 1) << 1) | ((ux & 2) << 1) | ((uy & 2) << 2) | ((ux & 4) << 2) | ((uy & 4) << 3);
}
// coding order of a tile-local 8x8 unit: superblocks in raster order, Morton order inside
template <int TSB>
__device__ __forceinline__ int unit_order(int ux, int uy) { return (((uy >> 3) * TSB + (ux >> 3)) << 6) | morton8(ux & 7, uy & 7); }
__device__ __forceinline__ bool mv_inside(const MvScan &m, int r4, int c4) { return r4 >= 0 && c4 >= 0 && r4 < m.max_r4 && c4 < m.max_c4; }
template <int TSB>
__device__ __forceinline__ void stack_add(MvScan &m, int r4, int c4, int weight) {
  const int u = (r4 >> 1) * TW + (c4 >> 1);
  if (!uni(TI.info[u].is_inter)) return;
  const int mr = uni(TI.info[u].mv_row), mc = uni(TI.info[u].mv_col);  // multiples of 8: lower_mv_precision is a no-op
  if (uni(g_inter.newmv[u])) m.new_count++;
  m.found = 1;
  int i = 0;
  for (; i < m.num; i++)
    if (uni(g_inter.stk_row[i]) == mr && uni(g_inter.stk_col[i]) == mc) break;
  if (i < m.num) g_inter.stk_w[i] = uni(g_inter.stk_w[i]) + weight;
  else if (m.num < 8) { g_inter.stk_row[m.num] = mr; g_inter.stk_col[m.num] = mc; g_inter.stk_w[m.num] = weight; m.num++; }
}
template <int TSB>
__device__ __forceinline__ int cand_n4(int r4, int c4) { return (1 << uni(TI.info[(r4 >> 1) * TW + (c4 >> 1)].bsl)) >> 2; }
template <int TSB>
__device__ __forceinline__ void scan_row(MvScan &m, int r4, int c4, int bw4, int delta_row) {
  int delta_col = 0, i = 0;
  const int end4 = imin(imin(bw4, m.max_c4 - c4), 16);
  if (iabs(delta_row) > 1) { delta_row += r4 & 1; delta_col = 1 - (c4 & 1); }
  while (i < end4) {
    const int r = r4 + delta_row, c = c4 + delta_col + i;
    if (!mv_inside(m, r, c)) break;
    int len = imin(cand_n4<TSB>(r, c), bw4);
    if (iabs(delta_row) > 1) len = imax(len, 2);
    if (bw4 >= 16) len = imax(len, 4);
    stack_add<TSB>(m, r, c, len * 2);
    i += len;
  }
}
template <int TSB>
__device__ __forceinline__ void scan_col(MvScan &m, int r4, int c4, int bh4, int delta_col) {
  int delta_row = 0, i = 0;
  const int end4 = imin(imin(bh4, m.max_r4 - r4), 16);
  if (iabs(delta_col) > 1) { delta_row = 1 - (r4 & 1); delta_col += c4 & 1; }
  while (i < end4) {
    const int r = r4 + delta_row + i, c = c4 + delta_col;
    if (!mv_inside(m, r, c)) break;
    int len = imin(cand_n4<TSB>(r, c), bh4);
    if (iabs(delta_col) > 1) len = imax(len, 2);
    if (bh4 >= 16) len = imax(len, 4);
    stack_add<TSB>(m, r, c, len * 2);
    i += len;
  }
}
template <int TSB>
__device__ __forceinline__ void scan_point(MvScan &m, int r4, int c4, int dr, int dc) {
  const int r = r4 + dr, c = c4 + dc;
  if (mv_inside(m, r, c) && unit_order<TSB>(c >> 1, r >> 1) < m.cur_z) stack_add<TSB>(m, r, c, 4);
}
__device__ __forceinline__ void sort_stack(int start, int end) {
  while (end > start) {
    int new_end = start;
    for (int i = start + 1; i < end; i++) {
      const int w0 = uni(g_inter.stk_w[i - 1]), w1 = uni(g_inter.stk_w[i]);
      if (w0 < w1) {
        const int r0 = uni(g_inter.stk_row[i - 1]), c0 = uni(g_inter.stk_col[i - 1]);
        g_inter.stk_row[i - 1] = uni(g_inter.stk_row[i]); g_inter.stk_col[i - 1] = uni(g_inter.stk_col[i]); g_inter.stk_w[i - 1] = w1;
        g_inter.stk_row[i] = r0; g_inter.stk_col[i] = c0; g_inter.stk_w[i] = w0;
        new_end = i;
      }
    }
    end = new_end;
  }
}
// returns NumMvFound; contexts through the references.  (r4, c4): block origin, n4: block size, all in 4x4 units
template <int TSB>
__device__ __forceinline__ int build_mv_stack(const Av1miDevParams &P, const TileGeo &tg, int r4, int c4, int n4, int &new_ctx, int &ref_ctx) {
  MvScan m;
  m.num = 0; m.new_count = 0; m.found = 0;
  // limits: the tile (TSB superblocks) clipped to the frame, in tile-local 4x4 units
  m.max_r4 = imin(16 * TSB, tg.max_y4_y + (tg.toy >> 2)); m.max_c4 = imin(16 * TSB, tg.max_x4_y + (tg.tox >> 2));
  m.cur_z = unit_order<TSB>(c4 >> 1, r4 >> 1);
  scan_row<TSB>(m, r4, c4, n4, -1);
  int found_above = m.found; m.found = 0;
  scan_col<TSB>(m, r4, c4, n4, -1);
  int found_left = m.found; m.found = 0;
  if (n4 <= 16) scan_point<TSB>(m, r4, c4, -1, n4);
  if (m.found) found_above = 1;
  m.found = 0;
  const int close_matches = found_above + found_left, num_nearest = m.num, num_new = m.new_count;
  for (int i = 0; i < num_nearest; i++) g_inter.stk_w[i] = uni(g_inter.stk_w[i]) + 640;  // REF_CAT_LEVEL
  scan_point<TSB>(m, r4, c4, -1, -1);
  if (m.found) found_above = 1;
  m.found = 0;
  scan_row<TSB>(m, r4, c4, n4, -3);
  if (m.found) found_above = 1;
  m.found = 0;
  scan_col<TSB>(m, r4, c4, n4, -3);
  if (m.found) found_left = 1;
  m.found = 0;
  scan_row<TSB>(m, r4, c4, n4, -5);
  if (m.found) found_above = 1;
  m.found = 0;
  scan_col<TSB>(m, r4, c4, n4, -5);
  if (m.found) found_left = 1;
  const int total_matches = found_above + found_left;
  sort_stack(0, num_nearest);
  sort_stack(num_nearest, m.num);
  // extra search (§7.10.2.12): with one reference in use it only meets vectors already in the list; pad with the
  // (zero) global vector
  for (int i = m.num; i < 2; i++) { g_inter.stk_row[i] = 0; g_inter.stk_col[i] = 0; g_inter.stk_w[i] = 0; }
  if (close_matches == 0) { new_ctx = imin(total_matches, 1); ref_ctx = total_matches; }
  else if (close_matches == 1) { new_ctx = 3 - imin(num_new, 1); ref_ctx = 2 + total_matches; }
  else { new_ctx = 5 - imin(num_new, 1); ref_ctx = 5; }
  // clamping (§7.10.2.14): the motion search keeps every coded vector within 16 samples of the frame, which is
  // inside the clamp range of every later block - nothing to do
  (void)P;
  return m.num;
}
__device__ __forceinline__ int drl_ctx(int idx) {
  const int w0 = uni(g_inter.stk_w[idx]), w1 = uni(g_inter.stk_w[idx + 1]);
  if (w0 >= 640 && w1 >= 640) return 0;
  if (w0 >= 640 && w1 < 640) return 1;
  if (w0 < 640 && w1 < 640) return 2;
  return 0;
}
// read_mv_component mirrored (§5.11.32), v != 0 in 1/8 samples (always a multiple of 8 here)
__device__ __forceinline__ void sym_mv_component(Sym &y, int lane, int adapt, int comp, int v) {
  const int base = CL::MV_COMP + comp * CL::MVC_SIZE;
  const int z = iabs(v) - 1;
  int cls = 0;
  while (cls < 10 && z >= (2 << (cls + 3))) cls++;
  const int o = z - (cls ? (2 << (cls + 2)) : 0);
  const int d = o >> 3, fr = (o >> 1) & 3;
  sym_wide<1>(y, lane, adapt, v < 0, base + CL::MVC_SIGN, 2);
  sym_wide<1>(y, lane, adapt, cls, base + CL::MVC_CLASS, 11);
  if (cls == 0) {
    sym_wide<1>(y, lane, adapt, d, base + CL::MVC_CLASS0, 2);
    sym_wide<1>(y, lane, adapt, fr, base + CL::MVC_CLASS0_FP + d * 5, 4);
  } else {
    for (int i = 0; i < cls; i++) sym_wide<1>(y, lane, adapt, (d >> i) & 1, base + CL::MVC_BITS + i * 3, 2);
    sym_wide<1>(y, lane, adapt, fr, base + CL::MVC_FP, 4);
  }
}

__device__ __forceinline__ int icdf_prob(const Sym &y, int off, int el) { return (el > 0 ? y.cdf[off + el - 1] : 32768) - y.cdf[off + el]; }

// Under a content-driven partition (P.part_map): does the superblock keep one block size throughout - no node between min_bs_log2
// and max_bs_log2 splits by its mask?  (Only then does a tile hold exactly two (transform size, plane type) classes.)
__device__ __forceinline__ bool sb_unsplit(const Av1miDevParams &P, int f, int sbr, int sbc) {
  if (!P.part_map || P.min_bs_log2 >= P.max_bs_log2) return true;
  const uint32_t m = P.part_map[((size_t)f * P.sb_rows + sbr) * P.sb_cols + sbc];
  return P.max_bs_log2 >= 6 ? !(m & 1u) : (P.max_bs_log2 == 5 ? !(m & 0x1Eu) : !(m & 0x1FFFE0u));
}

// ... and does the frame edge leave it alone?  A node of the leaf size whose origin lies inside the frame must not be forced to split
// (has_rows / has_cols of av1mi_node_split: a leaf may overhang the edge by less than half its size - the bottom superblock row of a
// 1080-row frame, 56 rows, keeps its four 32x32 leaves; a row of 40 would not).
__device__ __forceinline__ bool sb_uniform(const Av1miDevParams &P, int f, int sbr, int sbc) {
  if (!sb_unsplit(P, f, sbr, sbc)) return false;
  const int L = P.max_bs_log2, n = 1 << L;
  if (L <= 3) return true;
  for (int oy = 0; oy < 64; oy += n)
    for (int ox = 0; ox < 64; ox += n) {
      const int x = sbc * 64 + ox, yy = sbr * 64 + oy;
      if (x < P.width && yy < P.height && (yy + (n >> 1) >= P.height || x + (n >> 1) >= P.width)) return false;
    }
  return true;
}

// FULL = false: regular tiles in adaptive mode - every leaf of the tile has one size, i.e. exactly two (tx size, plane type)
// classes, no narrow rows in LDS (10.4 KB -> ~4 waves per SIMD).
// FULL = true: tiles that mix sizes (content-driven partition, forced splits at the frame edge) and static-CDF mode.  Each variant
// skips the other's tiles.
template <bool FULL, bool INTER, int TSB>
__global__ void __launch_bounds__(64) symbolize_tile_kernel(Av1miDevParams P, const uint16_t *__restrict__ cdf_init,
                                                           const int16_t *__restrict__ levels, const Av1miBlkInfo *__restrict__ blk,
                                                           uint32_t *__restrict__ streams, uint32_t *__restrict__ stream_len,
                                                           uint32_t *__restrict__ tile_combos, const uint8_t *__restrict__ lr_choice,
                                                           int tile0 /* chunk-wide index of this launch's first tile: launches cover whole frames */) {
  // one wave per TILE of TSB x TSB superblocks (raster order inside the tile)
  const int tiles_per_frame = P.tile_rows * P.tile_cols, sbs_per_frame = P.sb_rows * P.sb_cols;
  const int gt = (int)blockIdx.x + tile0;
  const int f = gt / tiles_per_frame, tile = gt % tiles_per_frame;
  const int tr = tile / P.tile_cols, tc = tile % P.tile_cols;
  const int lane = threadIdx.x;
  {
    bool regular = !P.disable_cdf_update;
    if (regular)   // (a tile whose superblocks mix block sizes holds more than two classes: the full variant's)
      for (int si = 0; si < TSB * TSB; si++) {
        const int sbr = tr * TSB + si / TSB, sbc = tc * TSB + si % TSB;
        if (sbr < P.sb_rows && sbc < P.sb_cols) regular = regular && sb_uniform(P, f, sbr, sbc);
      }
    if (regular == FULL) return;
  }
  if constexpr (FULL) {
    for (int i = lane; i < CL::COEFF_BASE; i += 64) g_cdfw_full[i] = cdf_init[i];
    if (lane < 18) g_cdfw_full[CL::COEFF_BASE + lane] = 0;
  } else {
    // the compact layout (struct CC): the tile's leaves have one size, so one luma and one chroma transform size
    const int l2y = P.max_bs_log2 > 6 ? 6 : P.max_bs_log2, l2c = l2y - 1 > 5 ? 5 : l2y - 1;
    uint16_t *const w = g_cdfw_compact;
    for (int i = lane; i < CC::LEAD; i += 64) w[i] = cdf_init[i];
    for (int i = lane; i < 13 * 8; i += 64)
      w[CC::TXSET + i] = l2y <= 3 ? cdf_init[CL::TX_SET1 + (l2y - 2) * 13 * 8 + i] : ((l2y == 4 && i < 13 * 6) ? cdf_init[CL::TX_SET2 + (l2y - 2) * 13 * 6 + i] : (uint16_t)0);
    for (int i = lane; i < 2 * 39; i += 64) { const int p = i / 39, txs = (p ? l2c : l2y) - 2; w[CC::TXB_SKIP + i] = cdf_init[CL::TXB_SKIP + txs * 39 + i % 39]; }
    if (lane < 24) {
      const int p = lane / 12, k = lane % 12, bwl = (p ? l2c : l2y) > 5 ? 5 : (p ? l2c : l2y), msz = 2 * bwl - 4, nsy = 5 + msz;
      const int eoff = CL::EOB16 + 4 * (msz * 6 + (msz * (msz - 1)) / 2) + (p * 2 + 0) * (nsy + 1);
      w[CC::EOB + lane] = k <= nsy ? cdf_init[eoff + k] : (uint16_t)0;
    }
    if (lane < 2 * 27) { const int p = lane / 27, txs = (p ? l2c : l2y) - 2; w[CC::EOB_EXTRA + lane] = cdf_init[CL::EOB_EXTRA + (txs * 2 + p) * 27 + lane % 27]; }
    if (lane < 18) w[CC::DC_SIGN + lane] = cdf_init[CL::DC_SIGN + lane];
    if (lane < 2 * 16) { const int p = lane / 16, txs = (p ? l2c : l2y) - 2; w[CC::COEFF_BASE_EOB + lane] = cdf_init[CL::COEFF_BASE_EOB + (txs * 2 + p) * 16 + lane % 16]; }
    for (int i = lane; i < CC::TOTAL - CC::USE_WIENER; i += 64) w[CC::USE_WIENER + i] = cdf_init[CL::USE_WIENER + i];
    if (lane < 18) w[CC::TOTAL + lane] = 0;
  }
  if (FULL) {
    for (int i = lane; i < CL::INTRA_TOTAL - CL::COEFF_BASE; i += 64) g_cdf_narrow[i] = cdf_init[CL::COEFF_BASE + i];
    g_cdf_narrow[CL::INTRA_TOTAL - CL::COEFF_BASE + lane] = 0;
  }
  if (INTER) {
    for (int i = lane; i < CL::TOTAL - CL::INTER_BASE; i += 64) g_cdf_inter[i] = cdf_init[CL::INTER_BASE + i];
    g_cdf_inter[CL::TOTAL - CL::INTER_BASE + lane] = 0;
  }
  {
    // block info of the whole tile (zero outside the frame), above contexts cleared (clear_above_context)
    const Av1miBlkInfo *finfo = blk + (size_t)f * P.b8_rows * P.b8_cols;
    for (int u = lane; u < 64 * TSB * TSB; u += 64) {
      const int r = tr * TW + u / TW, c = tc * TW + u % TW;
      Av1miBlkInfo bi = {};
      if (r < P.b8_rows && c < P.b8_cols) bi = finfo[(size_t)r * P.b8_cols + c];
      TI.info[u] = bi;
      if (INTER) g_inter.newmv[u] = 0;
    }
    for (int i = lane; i < 3 * 16 * TSB; i += 64) { (&TI.above_lvl[0][0])[i] = 0; (&TI.above_dc[0][0])[i] = 0; }
  }
  __syncthreads();
  Sym y;
  y.cdf = FULL ? g_cdfw_full : g_cdfw_compact;
  y.out = streams + (size_t)gt * P.stream_cap;
  y.pos = 0; y.cap = P.stream_cap;
  y.combo0 = -1; y.combo1 = -1;
  const int adapt = !P.disable_cdf_update;
  // INTER = false: key frames only (no inter syntax compiled in); INTER = true: the chunk's inter frames.  Each
  // instantiation skips the other's frames.
  const int inter_frame = INTER;
  if (av1mi_frame_is_inter(P, f) != (int)INTER) return;
  int lr_prev = 0;  // RefLrWiener of the tile: 0 = Wiener_Taps_Mid, k = candidate k-1 (the last unit coded with a Wiener filter)
  int sgr_prev = 0; // RefSgrXqd of the tile: 0 = Sgrproj_Xqd_Mid, k = self-guided candidate k-1 (the last self-guided unit)
#pragma nounroll
  for (int si = 0; si < TSB * TSB; si++) {
  const int sbr = tr * TSB + si / TSB, sbc = tc * TSB + si % TSB;
  if (sbr >= P.sb_rows || sbc >= P.sb_cols) continue;
  const int sb = sbr * P.sb_cols + sbc;
  TileGeo tg;
  tg.tox = (si % TSB) * 64; tg.toy = (si / TSB) * 64;
  tg.sb_x = sbc * 64; tg.sb_y = sbr * 64;
  tg.max_x4_y = P.mi_cols - sbc * 16; tg.max_y4_y = P.mi_rows - sbr * 16;
  tg.max_x4_c = (P.mi_cols >> 1) - sbc * 8; tg.max_y4_c = (P.mi_rows >> 1) - sbr * 8;
  const int tox = tg.tox, toy = tg.toy, tox8 = tox >> 3, toy8 = toy >> 3;
#define INFO(ux_, uy_) TI.info[((uy_) + toy8) * TW + (ux_) + tox8]
  const int16_t *sb_levels = levels + ((size_t)f * sbs_per_frame + sb) * AV1MI_SB_LEVELS;
  if (si % TSB == 0) {  // clear_left_context at the start of every superblock row of the tile
    __syncthreads();
    if (lane < 48) { (&S->left_lvl[0][0])[lane] = 0; (&S->left_dc[0][0])[lane] = 0; }
    __syncthreads();
  }

  if (P.enable_lr) {
    // read_lr (§5.11.57): the luma restoration unit whose origin lies in this superblock (units = 64x64, offset by 8
    // rows, so unit (r, c) starts in superblock (r, c)); its coefficients are coded against RefLrWiener, which starts
    // every tile at Wiener_Taps_Mid and follows the units coded with a filter
    const int urows = imax((P.true_h + 32) / 64, 1), ucols = imax((P.true_w + 32) / 64, 1);  // from the signalled size
    if (sbr < urows && sbc < ucols) {
      const int ch = uni(lr_choice[(size_t)f * urows * ucols + sbr * ucols + sbc]);
      // choice: 0 = off, 1..3 = Wiener candidate, 4..6 = self-guided candidate (RESTORE_SWITCHABLE frames only)
      if (P.enable_lr == 2) sym_wide(y, lane, adapt, ch == 0 ? 0 : (ch <= 3 ? 1 : 2), FULL ? CL::RESTORE_SW : CC::RESTORE_SW, 3);
      else sym_wide(y, lane, adapt, ch != 0, FULL ? CL::USE_WIENER : CC::USE_WIENER, 2);
      if (ch > 3) {
        const int len = P.sgr_code_len[sgr_prev][ch - 4];
        const unsigned long long bits = P.sgr_code_bits[sgr_prev][ch - 4];
        for (int i = len - 1; i >= 0; i--) emit1(y, lane, ENT_LITERAL((int)((bits >> i) & 1)));
        sgr_prev = ch - 3;
      } else if (ch) {
        const int len = P.lr_code_len[lr_prev][ch - 1];
        const unsigned long long bits = P.lr_code_bits[lr_prev][ch - 1];
        for (int i = len - 1; i >= 0; i--) emit1(y, lane, ENT_LITERAL((int)((bits >> i) & 1)));
        lr_prev = ch;
      }
    }
  }
#pragma nounroll
  for (int z = 0; z < 64; z++) {
    const int bx = (((z >> 0) & 1) | ((z >> 1) & 2) | ((z >> 2) & 4)) << 3;
    const int by = (((z >> 1) & 1) | ((z >> 2) & 2) | ((z >> 3) & 4)) << 3;
    if (tg.sb_y + by >= P.height || tg.sb_x + bx >= P.width) continue;
    const int b8x = bx >> 3, b8y = by >> 3;
    // leaf containing this 8x8 unit (written by the recon kernel); it is coded at its origin only
    const int leaf = uni(INFO(b8x, b8y).bsl);
    if ((bx | by) & ((1 << leaf) - 1)) continue;
    // nodes whose origin is (bx, by): every level above the leaf down to the leaf, largest first.
    // A node at level l > leaf with this origin contains a smaller leaf, so it is a SPLIT node.
    const int top = imin(6, __builtin_ctz((unsigned)(bx | by | 64)));
#pragma nounroll
    for (int bsl = top; bsl >= leaf; bsl--) {
      const int n = 1 << bsl;
      const bool split = bsl > leaf;
      {
        const int half = n >> 1;
        const bool has_rows = tg.sb_y + by + half < P.height, has_cols = tg.sb_x + bx + half < P.width;
        const int above = by + toy > 0 && uni(INFO(b8x, b8y - 1).bsl) < bsl;
        const int left = bx + tox > 0 && uni(INFO(b8x - 1, b8y).bsl) < bsl;
        const int off = CL::PARTITION + ((bsl - 3) * 4 + left * 2 + above) * 11;
        if (has_rows && has_cols) {
          sym_wide(y, lane, adapt, split ? 3 : 0, off, bsl == 3 ? 4 : 10);
        } else if (has_cols) {
          int p = icdf_prob(y, off, 2) + icdf_prob(y, off, 3);
          if (bsl != 3) p += icdf_prob(y, off, 4) + icdf_prob(y, off, 6) + icdf_prob(y, off, 7) + icdf_prob(y, off, 9);
          sym_bool(y, lane, 1, (uint32_t)uni(p));
        } else if (has_rows) {
          int p = icdf_prob(y, off, 1) + icdf_prob(y, off, 3);
          if (bsl != 3) p += icdf_prob(y, off, 4) + icdf_prob(y, off, 5) + icdf_prob(y, off, 6) + icdf_prob(y, off, 8);
          sym_bool(y, lane, 1, (uint32_t)uni(p));
        }
      }
      if (split) continue;
      {
        const int ymode = uni(INFO(b8x, b8y).ymode), skip = uni(INFO(b8x, b8y).skip);
        const int eob0 = uni(INFO(b8x, b8y).eob[0]), eob1 = uni(INFO(b8x, b8y).eob[1]), eob2 = uni(INFO(b8x, b8y).eob[2]);
        const int avail_u = by + toy > 0, avail_l = bx + tox > 0;  // inside the tile
        int sctx = 0;
        if (avail_u) sctx += uni(INFO(b8x, b8y - 1).skip);
        if (avail_l) sctx += uni(INFO(b8x - 1, b8y).skip);
        sym_wide(y, lane, adapt, skip, CL::SKIP + sctx * 3, 2);
        const int is_inter = inter_frame ? uni(INFO(b8x, b8y).is_inter) : 0;
        if (inter_frame) {
          // inter_frame_mode_info (§5.11.18): is_inter, context from the neighbours' intra-ness
          const int a_intra = avail_u ? !uni(INFO(b8x, b8y - 1).is_inter) : 0;
          const int l_intra = avail_l ? !uni(INFO(b8x - 1, b8y).is_inter) : 0;
          int ictx;
          if (avail_u && avail_l) ictx = (l_intra && a_intra) ? 3 : ((l_intra || a_intra) ? 1 : 0);
          else if (avail_u || avail_l) ictx = 2 * (avail_u ? a_intra : l_intra);
          else ictx = 0;
          sym_wide<1>(y, lane, adapt, is_inter, CL::IS_INTER + ictx * 3, 2);
        }
        if (is_inter) {
          // inter_block_mode_info (§5.11.23): LAST_FRAME = single_ref_p1 0, p3 0, p4 0; then the cheapest name of
          // the vector the recon kernel used: NEARESTMV, NEARMV, GLOBALMV, else NEWMV against the list's head
          const int n_last = (avail_u ? uni(INFO(b8x, b8y - 1).is_inter) : 0) + (avail_l ? uni(INFO(b8x - 1, b8y).is_inter) : 0);
          const int rctx = n_last > 0 ? 2 : 1;
          sym_wide<1>(y, lane, adapt, 0, CL::SINGLE_REF + (0 * 3 + rctx) * 3, 2);
          sym_wide<1>(y, lane, adapt, 0, CL::SINGLE_REF + (2 * 3 + rctx) * 3, 2);
          sym_wide<1>(y, lane, adapt, 0, CL::SINGLE_REF + (3 * 3 + rctx) * 3, 2);
          int new_ctx, ref_ctx;
          const int num = build_mv_stack<TSB>(P, tg, (by + toy) >> 2, (bx + tox) >> 2, n >> 2, new_ctx, ref_ctx);
          const int mvr = uni(INFO(b8x, b8y).mv_row), mvc = uni(INFO(b8x, b8y).mv_col);
          const int s0r = uni(g_inter.stk_row[0]), s0c = uni(g_inter.stk_col[0]), s1r = uni(g_inter.stk_row[1]), s1c = uni(g_inter.stk_col[1]);
          int mode;  // 0 NEARESTMV 1 NEARMV 2 GLOBALMV 3 NEWMV
          if (num >= 1 && mvr == s0r && mvc == s0c) mode = 0;
          else if (num >= 2 && mvr == s1r && mvc == s1c) mode = 1;
          else if ((mvr | mvc) == 0) mode = 2;
          else mode = 3;
          sym_wide<1>(y, lane, adapt, mode != 3, CL::NEWMV + new_ctx * 3, 2);
          if (mode != 3) {
            sym_wide<1>(y, lane, adapt, mode != 2, CL::GLOBALMV + 0 * 3, 2);
            if (mode != 2) sym_wide<1>(y, lane, adapt, mode == 1, CL::REFMV + ref_ctx * 3, 2);
          }
          if (mode == 3) { if (num > 1) sym_wide<1>(y, lane, adapt, 0, CL::DRL + drl_ctx(0) * 3, 2); }
          else if (mode == 1) { if (num > 2) sym_wide<1>(y, lane, adapt, 0, CL::DRL + drl_ctx(1) * 3, 2); }
          if (mode == 3) {
            const int dr = mvr - s0r, dc = mvc - s0c;  // list head, or the zero global vector when the list is empty
            sym_wide<1>(y, lane, adapt, (dr != 0 ? 2 : 0) | (dc != 0 ? 1 : 0), CL::MV_JOINT, 4);
            if (dr) sym_mv_component(y, lane, adapt, 0, dr);
            if (dc) sym_mv_component(y, lane, adapt, 1, dc);
          }
          {
            const int n8 = n >> 3;
            if (lane < n8 * n8) g_inter.newmv[(b8y + toy8 + lane / n8) * TW + b8x + tox8 + lane % n8] = (uint8_t)(mode == 3);
          }
        } else {
          if (inter_frame) {
            // intra_block_mode_info (§5.11.22): y_mode by block-size group
            sym_wide<1>(y, lane, adapt, ymode, CL::IF_Y_MODE + (bsl <= 3 ? 1 : (bsl == 4 ? 2 : 3)) * 14, 13);
          } else {
            const int am = intra_mode_ctx(uni(avail_u ? INFO(b8x, b8y - 1).ymode : 0));
            const int lm = intra_mode_ctx(uni(avail_l ? INFO(b8x - 1, b8y).ymode : 0));
            sym_wide(y, lane, adapt, ymode, CL::KF_Y_MODE + (am * 5 + lm) * 14, 13);
          }
          // chroma: the luma mode at luma's angle delta, or chroma from luma (`angle` bits 4-9 / 10-15: the alphas, not both zero)
          const int ainfo = uni(INFO(b8x, b8y).angle), adelta = ainfo & 7, cfl = (ainfo >> 4) != 0;
          if (ymode >= 1 && ymode <= 8) sym_wide(y, lane, adapt, adelta, CL::ANGLE_DELTA + (ymode - 1) * 8, 7);
          const int uvmode = cfl ? 13 : ymode;
          const int cfl_allowed = n <= 32;
          sym_wide(y, lane, adapt, uvmode, CL::UV_MODE + (cfl_allowed * 13 + ymode) * 15, cfl_allowed ? 14 : 13);
          if (cfl) {   // read_cfl_alphas (spec 5.11.45)
            const int au = ((ainfo >> 4) & 63) - 2 * ((ainfo >> 4) & 32), av = ((ainfo >> 10) & 63) - 2 * ((ainfo >> 10) & 32);   // 6-bit two's complement
            const int su = au == 0 ? 0 : (au < 0 ? 1 : 2), sv = av == 0 ? 0 : (av < 0 ? 1 : 2);
            sym_wide(y, lane, adapt, su * 3 + sv - 1, FULL ? CL::CFL_SIGN : CC::CFL_SIGN, 8);
            if (su) sym_wide(y, lane, adapt, iabs(au) - 1, (FULL ? CL::CFL_ALPHA : CC::CFL_ALPHA) + ((su - 1) * 3 + sv) * 17, 16);
            if (sv) sym_wide(y, lane, adapt, iabs(av) - 1, (FULL ? CL::CFL_ALPHA : CC::CFL_ALPHA) + ((sv - 1) * 3 + su) * 17, 16);
          }
          if (uvmode >= 1 && uvmode <= 8) sym_wide(y, lane, adapt, adelta, CL::ANGLE_DELTA + (uvmode - 1) * 8, 7);
        }
        const int w4 = n >> 2, w4c = imax(w4 >> 1, 1);
        const int log2c = bsl - 1;
        if (skip) {
          __syncthreads();
          if (lane < w4) { TI.above_lvl[0][((bx + tox) >> 2) + lane] = 0; TI.above_dc[0][((bx + tox) >> 2) + lane] = 0; S->left_lvl[0][(by >> 2) + lane] = 0; S->left_dc[0][(by >> 2) + lane] = 0; }
          if (lane < w4c) {
            for (int pl = 1; pl < 3; pl++) { TI.above_lvl[pl][((bx + tox) >> 3) + lane] = 0; TI.above_dc[pl][((bx + tox) >> 3) + lane] = 0; S->left_lvl[pl][(by >> 3) + lane] = 0; S->left_dc[pl][(by >> 3) + lane] = 0; }
          }
          __syncthreads();
        } else {
          for (int pl = 0; pl < 3; pl++) {
            const int l2 = pl ? log2c : bsl;
            const int16_t *lvp = sb_levels + av1mi_levels_off(pl, bx, by);
            sym_coeffs<FULL, TSB>(y, lane, adapt, tg, pl, l2, pl ? bx >> 3 : bx >> 2, pl ? by >> 3 : by >> 2, pl == 0 ? eob0 : (pl == 1 ? eob1 : eob2), ymode | (pl == 0 ? ((uni(INFO(b8x, b8y).angle) >> 3) & 1) << 8 : 0), is_inter, lvp);
          }
        }
      }
    }
  }
#undef INFO
  }  // superblocks of the tile
  if (lane == 0) {
    stream_len[gt] = (uint32_t)y.pos;
    tile_combos[gt] = (uint32_t)((y.combo0 & 0xFF) | ((y.combo1 & 0xFF) << 8));
  }
}
#undef S

// Tiles in the order of decreasing symbol-stream length: a counting sort in one workgroup (histogram over the length in LDS,
// exclusive scan, scatter).  Which tile of a bucket comes first does not matter -
// every tile writes only its own slot.
__global__ void __launch_bounds__(1024) tile_order_kernel(int n_tiles, const uint32_t *__restrict__ stream_len, uint32_t *__restrict__ order) {
  // One bucket per length (8192; longer streams share the first): with one per 16 entries most tiles of a chunk met in a few dozen
  // buckets and the LDS atomics of a wave serialised on them - 161 us for the 30 600 tiles of a 1080p x 60 chunk, a third of what the
  // four-stage range coder then takes.  The lengths are read eight per thread with all loads in flight before the first atomic; the
  // bucket bases come from a scan over eight buckets per thread, wave shuffles, and the 16 waves' sums.
  constexpr int NB = 8192, PER = NB / 1024, U = 8;
  __shared__ uint32_t hist[NB];
  __shared__ uint32_t wsum[16];
  const int t = threadIdx.x;
  auto bucket = [](uint32_t len) { return (uint32_t)(NB - 1) - (len > (uint32_t)(NB - 1) ? (uint32_t)(NB - 1) : len); };
  for (int k = 0; k < PER; k++) hist[k * 1024 + t] = 0;
  __syncthreads();
  for (int i0 = 0; i0 < n_tiles; i0 += 1024 * U) {
    uint32_t v[U];
#pragma unroll
    for (int u = 0; u < U; u++) { const int i = i0 + u * 1024 + t; v[u] = stream_len[i < n_tiles ? i : n_tiles - 1]; }   // (a load under a condition is waited for on the spot)
#pragma unroll
    for (int u = 0; u < U; u++) { const int i = i0 + u * 1024 + t; if (i < n_tiles) atomicAdd(&hist[bucket(v[u])], 1u); }
  }
  __syncthreads();
  // exclusive scan: thread t owns buckets PER t .. PER t + PER - 1
  uint32_t own[PER], tot = 0;
#pragma unroll
  for (int k = 0; k < PER; k++) { own[k] = hist[t * PER + k]; tot += own[k]; }
  uint32_t inc = tot;
  for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(inc, o, 64); if ((t & 63) >= o) inc += y; }
  if ((t & 63) == 63) wsum[t >> 6] = inc;
  __syncthreads();
  uint32_t base = inc - tot;
  for (int w = 0; w < (t >> 6); w++) base += wsum[w];
#pragma unroll
  for (int k = 0; k < PER; k++) { hist[t * PER + k] = base; base += own[k]; }
  __syncthreads();
  for (int i0 = 0; i0 < n_tiles; i0 += 1024 * U) {
    uint32_t v[U];
#pragma unroll
    for (int u = 0; u < U; u++) { const int i = i0 + u * 1024 + t; v[u] = stream_len[i < n_tiles ? i : n_tiles - 1]; }   // (a load under a condition is waited for on the spot)
#pragma unroll
    for (int u = 0; u < U; u++) { const int i = i0 + u * 1024 + t; if (i < n_tiles) order[atomicAdd(&hist[bucket(v[u])], 1u)] = (uint32_t)i; }
  }
}

// ================================================================================= K4
// One lane per tile, 64 tiles per workgroup, FOUR waves per workgroup - one per SIMD of the CU - working as a pipeline over batches of
// RC_BATCH entries (batch k is at stage j in trip k + j; double-buffered LDS rings between the stages, one barrier per trip):
//   wave 0 (adapt):   walks the tile's stream and adapts the tile's narrow CDF rows (LDS, [slot][lane] x 8 bytes {c0, c1, c2, counter});
//                     passes on every entry's row as it was BEFORE the entry;
//   wave 1 (resolve): turns (entry, row) into a RESOLVED entry {fl, fh, N - s};
//   wave 2 (range):   the range recurrence alone: rng -> (what the entry adds to `low`, the normalisation shift);
//   wave 3 (output):  accumulates `low` and writes the tile's bytes.
// The kernel's duration is the serial chain of its longest tile (~4.8 k symbols at 1080p) times the instructions per entry of the
// slowest stage: a wave alone on its SIMD issues one vector instruction per ~4.4 cycles, so the two-wave form (resolver 61, coder 45
// instructions per entry) was bound at 61; the stages here are 34 / 25 / 20 / 34.
#define RC_BATCH 16
#define RC_DUMMY (MAX_COMBOS * SLOTS_PER_COMBO)
struct RcLds {
  uint64_t row[RC_DUMMY + 1][64];   // + one dummy row: entries that need no resolving go through it, branch-free
  uint64_t ring_row[2][RC_BATCH][64];
  uint32_t ring_ent[2][RC_BATCH][64];
  uint32_t ring_upd[2][RC_BATCH][64];
};
__shared__ RcLds g_rc;

__global__ void __launch_bounds__(256) rangecode4_tiles_kernel(Av1miDevParams P, int n_tiles, const uint16_t *__restrict__ cdf_init,
                                                             const uint32_t *__restrict__ streams, const uint32_t *__restrict__ stream_len,
                                                             const uint32_t *__restrict__ tile_combos, uint8_t *__restrict__ slots,
                                                             uint32_t *__restrict__ tile_bytes, const uint32_t *__restrict__ order,
                                                             int tile0 /* chunk-wide index of the launch's first tile; n_tiles and `order` are launch-local */) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  // order != nullptr (more workgroups than the chip holds at once): a permutation of the launch's tiles, see the launcher
  const bool live = (int)(blockIdx.x * 64 + lane) < n_tiles;
  const int tile = live ? tile0 + (order ? (int)order[blockIdx.x * 64 + lane] : (int)(blockIdx.x * 64 + lane)) : 0;
  const int count_raw = live ? (int)stream_len[tile] : 0;
  const bool overflow = count_raw > P.stream_cap;
  const int count = overflow ? 0 : count_raw;
  int maxcount = count;
  for (int o = 32; o > 0; o >>= 1) { const int t = __shfl_xor(maxcount, o, 64); maxcount = t > maxcount ? t : maxcount; }
  const int nb = (maxcount + RC_BATCH - 1) / RC_BATCH;
  const uint32_t *st = streams + (size_t)(live ? tile : 0) * P.stream_cap;
  const int adapt = !P.disable_cdf_update;

  // ---- adapt stage: per-lane CDF rows from the defaults of this tile's two (tx size, plane type) classes
  if (wave == 0) {
    const uint32_t cm = live ? tile_combos[tile] : 0xFFFFu;
    for (int k = 0; k < MAX_COMBOS; k++) {
      const int combo = (cm >> (8 * k)) & 0xFF;
      if (combo == 0xFF) continue;
      const int txs = combo >> 1, ptype = combo & 1;
      const uint16_t *b = cdf_init + CL::COEFF_BASE + (txs * 2 + ptype) * 42 * 5;
      const uint16_t *r = cdf_init + CL::COEFF_BR + ((txs > 3 ? 3 : txs) * 2 + ptype) * 21 * 5;
      for (int j = 0; j < 42; j++) g_rc.row[k * SLOTS_PER_COMBO + j][lane] = (uint64_t)b[j * 5] | ((uint64_t)b[j * 5 + 1] << 16) | ((uint64_t)b[j * 5 + 2] << 32);
      for (int j = 0; j < 21; j++) g_rc.row[k * SLOTS_PER_COMBO + 42 + j][lane] = (uint64_t)r[j * 5] | ((uint64_t)r[j * 5 + 1] << 16) | ((uint64_t)r[j * 5 + 2] << 32);
    }
    g_rc.row[RC_DUMMY][lane] = 0;
  }
  // ---- range stage / output stage state
  uint32_t low = 0, rng = 0x8000;
  int cnt = -9, out_pos = 0;
  // Output: "pre-carry" entries, one 16-bit value per output byte holding the byte and, in bit 8, a carry that still
  // has to be added to the bytes before it (od_ec's precarry buffer).  Nothing already written is ever touched here;
  // pack_tiles_kernel resolves the carries of a whole tile with a wave-parallel carry-lookahead when it copies the
  // tile to its final place.
  // (the tile's slot is a per-lane 64-bit base - chunks whose slots exceed 4 GB are legal: 4K x 140 frames at capacity scale 2 -
  // and the entry's position inside it a 32-bit offset, one v_lshl_add_u64 per store; an entry beyond the slot's capacity goes to
  // the last one: the overflow is reported through tile_bytes, what the slot then holds does not matter)
  uint16_t *const out_tile = reinterpret_cast<uint16_t *>(slots) + (size_t)(live ? tile : 0) * (size_t)P.tile_slot_bytes;
  const int out_cap = P.tile_slot_bytes;  // entries
#define PUT(pos_, v_)                                                                                                    \
  do {                                                                                                                   \
    out_tile[(uint32_t)((pos_) < out_cap ? (pos_) : out_cap - 1)] = (uint16_t)((v_) & 0x1FFu);                           \
  } while (0)
#define EMIT(v_) do { PUT(out_pos, v_); out_pos++; } while (0)

  __syncthreads();
  // The stream reads of waves 0 and 1 are one 16-byte request per lane into 64 different streams (an HBM / L2 round trip each):
  // the next batch is loaded while the current one is worked on (handing the entries from wave 0 to wave 1 through LDS instead was
  // measured: no faster).  What lies beyond a tile's count inside the last 16 bytes is whatever an earlier chunk left there: wave 0
  // may adapt rows with it (the tile is over), wave 1 replaces it by an entry that codes nothing.
  // Three register buffers (batch mod 3): a buffer is refilled with batch + 3 at the END of the trip that used it - when its registers
  // are dead, so that the loaded registers ARE the loop-carried ones and nothing is copied (a copy waits for the load just issued) -
  // which gives a load two whole trips to arrive.  The loads are unconditional (a load under `i < count` costs a select per register
  // and makes the compiler wait for ALL loads in flight wherever one is used): a lane whose tile is shorter than the wave's longest
  // reads what an earlier chunk left in its slot - batch kb < nb lies inside the slot, whose capacity is a multiple of the batch -
  // and nothing uses it.
  typedef uint4 QBuf[RC_BATCH / 4];
  QBuf qa, qb, qc;
  auto load_batch = [&](QBuf &q, int kb) __attribute__((always_inline)) {
    kb = kb < nb ? kb : (nb > 0 ? nb - 1 : 0);
#pragma unroll
    for (int j = 0; j < RC_BATCH; j += 4) q[j / 4] = *reinterpret_cast<const uint4 *>(st + kb * RC_BATCH + j);
  };
  if (wave <= 1) { load_batch(qa, 0); load_batch(qb, 1); load_batch(qc, 2); }
  // Every wave runs a loop of its own with the same number of barriers (nb + 3): in one loop with a branch per wave the compiler's
  // wait-count pass loses track of the loads in flight at the joins and waits for all of them.
  auto trip_end = [&]() __attribute__((always_inline)) {
    // (the fences name the LDS address space only: the stream loads in flight and the output stores must not be waited for here)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
  };
  const int n_trips = (nb + 3 + 2) / 3 * 3;   // (a multiple of 3: the loops of waves 0 and 1 are unrolled by the three buffers)
  auto stage0 = [&](const int k, QBuf &q0) __attribute__((always_inline)) {
    {
      if (k < nb) {
        uint32_t ev[RC_BATCH];
#pragma unroll
        for (int j = 0; j < RC_BATCH; j += 4) { ev[j] = q0[j / 4].x; ev[j + 1] = q0[j / 4].y; ev[j + 2] = q0[j / 4].z; ev[j + 3] = q0[j / 4].w; }
        // The 16 entries of the batch, one after the other.  An entry that is already resolved runs the same instructions against the
        // dummy row: no divergent branch.  The row of entry i + 1 is read BEFORE entry i's row is written back, and replaced by that
        // new row if both entries name the same slot - otherwise every entry waited for an LDS round trip behind the previous
        // entry's write (read -> adapt -> write -> read ...).
        auto slot_of = [&](uint32_t e) { const uint32_t t = e >> 2; return (int)(t < (uint32_t)RC_DUMMY ? t : (uint32_t)RC_DUMMY); };
        int slot = slot_of(ev[0]);
        uint64_t rw = g_rc.row[slot][lane];
#pragma unroll
        for (int jj = 0; jj < RC_BATCH; jj++) {
          const int s = ev[jj] & 3;
          int slot_nx = 0;
          uint64_t rw_nx = 0;
          if (jj + 1 < RC_BATCH) { slot_nx = slot_of(ev[jj + 1]); rw_nx = g_rc.row[slot_nx][lane]; }
          g_rc.ring_row[k & 1][jj][lane] = rw;
          uint64_t nrow = rw;
          if (adapt) {
            // the three values move towards 32768 (index < s) or 0 by their distance >> rate: packed 16-bit arithmetic on {c0, c1}
            // and on {c2, counter} (the counter half is replaced afterwards)
            typedef unsigned short us2 __attribute__((ext_vector_type(2)));
            const uint32_t c01 = (uint32_t)rw, c2n = (uint32_t)(rw >> 32);  // {c0, c1}, {c2, counter}
            const uint32_t cn = c2n >> 16;
            const unsigned short rate = (unsigned short)(5 + (cn >> 4));   // counter <= 32: 5 + (cn > 15) + (cn > 31)
            const us2 rv = { rate, rate }, top = { 0x8000, 0x8000 };
            const us2 a = __builtin_bit_cast(us2, c01), b = __builtin_bit_cast(us2, c2n);
            const uint32_t up01 = __builtin_bit_cast(uint32_t, (us2)(a + ((top - a) >> rv))), dn01 = __builtin_bit_cast(uint32_t, (us2)(a - (a >> rv)));
            const uint32_t up2 = __builtin_bit_cast(uint32_t, (us2)(b + ((top - b) >> rv))), dn2 = __builtin_bit_cast(uint32_t, (us2)(b - (b >> rv)));
            const uint32_t m01 = s >= 2 ? 0xFFFFFFFFu : (s ? 0xFFFFu : 0u);   // halves with index < s
            const uint32_t n01 = (up01 & m01) | (dn01 & ~m01);
            const uint32_t cn1 = cn + 1 < 32u ? cn + 1 : 32u;
            const uint32_t n2 = ((s > 2 ? up2 : dn2) & 0xFFFFu) | (cn1 << 16);
            nrow = (uint64_t)n01 | ((uint64_t)n2 << 32);
            g_rc.row[slot][lane] = nrow;
          }
          if (jj + 1 < RC_BATCH) { rw = slot_nx == slot ? nrow : rw_nx; slot = slot_nx; }
        }
      }
    }
    load_batch(q0, k + 3);
    trip_end();
  };
  auto stage1 = [&](const int k, QBuf &q1) __attribute__((always_inline)) {
    {
      if (k >= 1 && k - 1 < nb) {
        const int kb = k - 1;
        uint32_t ev[RC_BATCH];
#pragma unroll
        for (int j = 0; j < RC_BATCH; j += 4) { ev[j] = q1[j / 4].x; ev[j + 1] = q1[j / 4].y; ev[j + 2] = q1[j / 4].z; ev[j + 3] = q1[j / 4].w; }
        uint64_t rws[RC_BATCH];
#pragma unroll
        for (int j = 0; j < RC_BATCH; j++) rws[j] = g_rc.ring_row[kb & 1][j][lane];
#pragma unroll
        for (int jj = 0; jj < RC_BATCH; jj++) {
          uint32_t ent = ev[jj];
          const int s = ent & 3;
          const uint32_t c01 = (uint32_t)rws[jj], c2 = (uint32_t)(rws[jj] >> 32) & 0xFFFFu;
          // fl = icdf[s-1] (32768 for s == 0), fh = icdf[s] (0 for s == 3): one 64-bit shift each of {32768, c0, c1, c2} / {c0, c1, c2, 0}
          const uint64_t vals = ((uint64_t)c2 << 32) | c01;
          const uint32_t fh = (uint32_t)(vals >> (16 * s)) & 0xFFFFu;
          const uint32_t fl = (uint32_t)(((vals << 16) | 0x8000u) >> (16 * s)) & 0xFFFFu;
          ent = (ent & 0x80000000u) ? ent : ENT_RESOLVED(fl >> 6, fh >> 6, 3 - s);
          // beyond this tile's count: "the whole range" (fl = 32768, fh = 0 of a one-symbol alphabet) changes nothing and emits nothing
          ent = kb * RC_BATCH + jj < count ? ent : ENT_RESOLVED(512, 0, 0);
          g_rc.ring_ent[kb & 1][jj][lane] = ent;
        }
      }
    }
    if (k >= 1) load_batch(q1, k - 1 + 3);
    trip_end();
  };
  auto stage2 = [&](const int k) __attribute__((always_inline)) {
    {
      if (k >= 2 && k - 2 < nb) {
        const int kb = k - 2;
        uint32_t ce[RC_BATCH];
#pragma unroll
        for (int j = 0; j < RC_BATCH; j++) ce[j] = g_rc.ring_ent[kb & 1][j][lane];
#pragma unroll
        for (int j = 0; j < RC_BATCH; j++) {
          const uint32_t ent = ce[j];
          const uint32_t fl6 = (ent >> 14) & 0x3FF, fh6 = (ent >> 4) & 0x3FF, ns = ent & 15;
          // range update (od_ec_encode_q15, the mirror of spec §8.2.6), branch-free: fl6 == 512 <=> s == 0
          const uint32_t r = rng, r8 = r >> 8;
          // (r8 < 2^8, fl6 / fh6 <= 512: 24-bit multiplies, full rate)
          const uint32_t v = (__umul24(r8, fh6) >> 1) + 4u * ns;
          const uint32_t u = fl6 >= 512 ? r : (__umul24(r8, fl6) >> 1) + 4u * ns + 4u;
          const uint32_t nr = u - v;
          const int d = __builtin_clz(nr) - 16;
          rng = nr << d;
          g_rc.ring_upd[kb & 1][j][lane] = (r - u) | ((uint32_t)d << 16);
        }
      }
    }
    trip_end();
  };
  auto stage3 = [&](const int k) __attribute__((always_inline)) {
    {
      if (k >= 3 && k - 3 < nb) {
        const int kb = k - 3;
        uint32_t cu[RC_BATCH];
#pragma unroll
        for (int j = 0; j < RC_BATCH; j++) cu[j] = g_rc.ring_upd[kb & 1][j][lane];
#pragma unroll
        for (int j = 0; j < RC_BATCH; j++) {
          const int d = (int)(cu[j] >> 16);
          uint32_t l = low + (cu[j] & 0xFFFFu);
          int s2 = cnt + d;
          if (s2 >= 0) {   // one byte, or two when at least 8 bits are ready (od_ec_enc_normalize)
            int c = cnt + 16;
            const bool two = s2 >= 8;
            // no inner branch: the first byte is stored in any case and, if it was not due, overwritten by the second at the same position
            PUT(out_pos, l >> c);
            out_pos += two;
            l = two ? l & ((1u << c) - 1) : l;
            c = two ? c - 8 : c;
            EMIT(l >> c);
            l &= (1u << c) - 1;
            s2 = c + d - 24;
          }
          low = l << d;
          cnt = s2;
        }
      }
    }
    trip_end();
  };
  if (wave == 0) {
    for (int k = 0; k < n_trips; k += 3) { stage0(k, qa); stage0(k + 1, qb); stage0(k + 2, qc); }
  } else if (wave == 1) {   // (its batch is k - 1)
    for (int k = 0; k < n_trips; k += 3) { stage1(k, qc); stage1(k + 1, qa); stage1(k + 2, qb); }
  } else if (wave == 2) {
    for (int k = 0; k < n_trips; k++) stage2(k);
  } else {
    for (int k = 0; k < n_trips; k++) stage3(k);
  }
  // ---- finish (od_ec_enc_done), output wave
  if (wave == 3) {
    if (live && !overflow) {
      uint32_t l = low;
      int c = cnt, s = 10;
      const uint32_t m = 0x3FFF;
      uint32_t v = ((l + m) & ~m) | (m + 1);
      s += c;
      if (s > 0) {
        uint32_t n = (1u << (c + 16)) - 1;
        do {
          EMIT(v >> (c + 16));
          v &= n;
          s -= 8;
          c -= 8;
          n >>= 8;
        } while (s > 0);
      }
    }
    if (live) tile_bytes[tile] = overflow ? 0xFFFFFFFFu : (uint32_t)out_pos;
  }
}
#undef EMIT
#undef PUT

// ---- K4, two-stage form (more than 256 workgroups: see the launcher)
// One lane per tile, 64 tiles per workgroup, TWO waves per workgroup working as a pipeline:
//   wave 0 (resolver): walks the tile's stream, adapts the tile's narrow CDF rows (LDS, [slot][lane] x
//           8 bytes {c0, c1, c2, counter}) and turns every entry into a RESOLVED one in an LDS ring;
//   wave 1 (coder):    runs the range coder over the ring and writes the tile's bytes.
// The kernel's duration is the serial chain of its longest tile (~4.8 k symbols at 1080p), so halving
// the instructions per link of that chain matters more than anything else here.
struct Rc2Lds {
  uint64_t row[MAX_COMBOS * SLOTS_PER_COMBO + 1][64];   // + one dummy row: entries that need no resolving go through it, branch-free
  uint32_t ring[2][RC_BATCH][64];
};
__shared__ Rc2Lds g_rc2;

__global__ void __launch_bounds__(128) rangecode2_tiles_kernel(Av1miDevParams P, int n_tiles, const uint16_t *__restrict__ cdf_init,
                                                             const uint32_t *__restrict__ streams, const uint32_t *__restrict__ stream_len,
                                                             const uint32_t *__restrict__ tile_combos, uint8_t *__restrict__ slots,
                                                             uint32_t *__restrict__ tile_bytes, const uint32_t *__restrict__ order,
                                                             int tile0 /* chunk-wide index of the launch's first tile; n_tiles and `order` are launch-local */) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  // order != nullptr (more tiles than the chip holds workgroups for at once): the 64 tiles of a workgroup are neighbours in
  // the order of decreasing stream length (tile_order_kernel) - a wave lasts as long as its longest tile, so similar lengths
  // waste the fewest lane-cycles, and the long ones start first
  const bool live = (int)(blockIdx.x * 64 + lane) < n_tiles;
  const int tile = live ? tile0 + (order ? (int)order[blockIdx.x * 64 + lane] : (int)(blockIdx.x * 64 + lane)) : 0;
  const int count_raw = live ? (int)stream_len[tile] : 0;
  const bool overflow = count_raw > P.stream_cap;
  const int count = overflow ? 0 : count_raw;
  int maxcount = count;
  for (int o = 32; o > 0; o >>= 1) { const int t = __shfl_xor(maxcount, o, 64); maxcount = t > maxcount ? t : maxcount; }
  const int nb = (maxcount + RC_BATCH - 1) / RC_BATCH;
  const uint32_t *st = streams + (size_t)(live ? tile : 0) * P.stream_cap;
  const int adapt = !P.disable_cdf_update;

  // ---- resolver state
  if (wave == 0) {
    // per-lane CDF rows from the defaults of this tile's two (tx size, plane type) classes
    const uint32_t cm = live ? tile_combos[tile] : 0xFFFFu;
    for (int k = 0; k < MAX_COMBOS; k++) {
      const int combo = (cm >> (8 * k)) & 0xFF;
      if (combo == 0xFF) continue;
      const int txs = combo >> 1, ptype = combo & 1;
      const uint16_t *b = cdf_init + CL::COEFF_BASE + (txs * 2 + ptype) * 42 * 5;
      const uint16_t *r = cdf_init + CL::COEFF_BR + ((txs > 3 ? 3 : txs) * 2 + ptype) * 21 * 5;
      for (int j = 0; j < 42; j++) g_rc2.row[k * SLOTS_PER_COMBO + j][lane] = (uint64_t)b[j * 5] | ((uint64_t)b[j * 5 + 1] << 16) | ((uint64_t)b[j * 5 + 2] << 32);
      for (int j = 0; j < 21; j++) g_rc2.row[k * SLOTS_PER_COMBO + 42 + j][lane] = (uint64_t)r[j * 5] | ((uint64_t)r[j * 5 + 1] << 16) | ((uint64_t)r[j * 5 + 2] << 32);
    }
    g_rc2.row[RC_DUMMY][lane] = 0;
  }
  // ---- coder state
  uint32_t low = 0, rng = 0x8000;
  int cnt = -9, out_pos = 0;
  // Output: "pre-carry" entries, one 16-bit value per output byte holding the byte and, in bit 8, a carry that still
  // has to be added to the bytes before it (od_ec's precarry buffer).  Nothing already written is ever touched here;
  // pack_tiles_kernel resolves the carries of a whole tile with a wave-parallel carry-lookahead when it copies the
  // tile to its final place.  This keeps the serial chain per symbol short: the kernel lasts as long as its longest tile.
  // (the tile's slot is a per-lane 64-bit base - chunks whose slots exceed 4 GB are legal: 4K x 140 frames at capacity scale 2 -
  // and the entry's position inside it a 32-bit offset, one v_lshl_add_u64 per store; an entry beyond the slot's capacity goes to
  // the last one: the overflow is reported through tile_bytes, what the slot then holds does not matter)
  uint16_t *const out_tile = reinterpret_cast<uint16_t *>(slots) + (size_t)(live ? tile : 0) * (size_t)P.tile_slot_bytes;
  const int out_cap = P.tile_slot_bytes;  // entries
#define PUT(pos_, v_)                                                                                                    \
  do {                                                                                                                   \
    out_tile[(uint32_t)((pos_) < out_cap ? (pos_) : out_cap - 1)] = (uint16_t)((v_) & 0x1FFu);                           \
  } while (0)
#define EMIT(v_) do { PUT(out_pos, v_); out_pos++; } while (0)

  __syncthreads();
  // batch k is resolved by wave 0 in trip k and coded by wave 1 in trip k + 1.  The resolver's stream reads are one
  // 16-byte request per lane into 64 different streams (an HBM round trip each): batch k + 1 is loaded while batch k is
  // being resolved, otherwise every trip would start with that latency.
  // (three register buffers refilled with batch + 3 at the end of the trip that used them, unconditional loads, a loop per wave with the
  // same number of barriers: see the four-stage form)
  typedef uint4 QBuf[RC_BATCH / 4];
  QBuf qa, qb, qd;
  auto load_batch = [&](QBuf &q, int kb) __attribute__((always_inline)) {
    kb = kb < nb ? kb : (nb > 0 ? nb - 1 : 0);
#pragma unroll
    for (int j = 0; j < RC_BATCH; j += 4) q[j / 4] = *reinterpret_cast<const uint4 *>(st + kb * RC_BATCH + j);
  };
  if (wave == 0) { load_batch(qa, 0); load_batch(qb, 1); load_batch(qd, 2); }
  auto trip_end = [&]() __attribute__((always_inline)) {
    // (the fences name the LDS address space only: the stream loads in flight and the output stores must not be waited for here)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
  };
  const int n_trips = (nb + 1 + 2) / 3 * 3;
  auto resolve_trip = [&](const int k, QBuf &q0) __attribute__((always_inline)) {
    {
      if (k < nb) {
        uint4 qc[RC_BATCH / 4];
#pragma unroll
        for (int j = 0; j < RC_BATCH / 4; j++) qc[j] = q0[j];
        // The 16 entries of the batch, one after the other.  An entry that is already resolved (or lies beyond this tile's count) runs
        // the same instructions against a dummy row and keeps its value: no divergent branch.  The row of entry i + 1 is read BEFORE
        // entry i's row is written back, and replaced by that new row if both entries name the same slot - otherwise every entry
        // waited for an LDS round trip behind the previous entry's write (read -> adapt -> write -> read ...).
        uint32_t ev[RC_BATCH];
#pragma unroll
        for (int j = 0; j < RC_BATCH; j += 4) { ev[j] = qc[j / 4].x; ev[j + 1] = qc[j / 4].y; ev[j + 2] = qc[j / 4].z; ev[j + 3] = qc[j / 4].w; }
        // (no test against the tile's count here: what lies beyond it may adapt rows - the tile is over - and the coder replaces it)
        auto slot_of = [&](uint32_t e) { const uint32_t t = e >> 2; return (int)(t < (uint32_t)RC_DUMMY ? t : (uint32_t)RC_DUMMY); };
        int slot = slot_of(ev[0]);
        uint64_t rw = g_rc2.row[slot][lane];
#pragma unroll
        for (int jj = 0; jj < RC_BATCH; jj++) {
          uint32_t ent = ev[jj];
          const bool nar = !(ent & 0x80000000u);
          const int s = ent & 3;
          int slot_nx = 0;
          uint64_t rw_nx = 0;
          if (jj + 1 < RC_BATCH) { slot_nx = slot_of(ev[jj + 1]); rw_nx = g_rc2.row[slot_nx][lane]; }
          const uint32_t c01 = (uint32_t)rw, c2n = (uint32_t)(rw >> 32);  // {c0, c1}, {c2, counter}
          // fl = icdf[s-1] (32768 for s == 0), fh = icdf[s] (0 for s == 3): one 64-bit shift each of {32768, c0, c1, c2} / {c0, c1, c2, 0}
          const uint64_t vals = ((uint64_t)(c2n & 0xFFFFu) << 32) | c01;
          const uint32_t fh = (uint32_t)(vals >> (16 * s)) & 0xFFFFu;
          const uint32_t fl = (uint32_t)(((vals << 16) | 0x8000u) >> (16 * s)) & 0xFFFFu;
          ent = nar ? ENT_RESOLVED(fl >> 6, fh >> 6, 3 - s) : ent;
          uint64_t nrow = rw;
          if (adapt) {
            // the three values move towards 32768 (index < s) or 0 by their distance >> rate: packed 16-bit arithmetic on {c0, c1}
            // and on {c2, counter} (the counter half is replaced afterwards)
            typedef unsigned short us2 __attribute__((ext_vector_type(2)));
            const uint32_t cn = c2n >> 16;
            const unsigned short rate = (unsigned short)(5 + (cn >> 4));   // counter <= 32: 5 + (cn > 15) + (cn > 31)
            const us2 rv = { rate, rate }, top = { 0x8000, 0x8000 };
            const us2 a = __builtin_bit_cast(us2, c01), b = __builtin_bit_cast(us2, c2n);
            const uint32_t up01 = __builtin_bit_cast(uint32_t, (us2)(a + ((top - a) >> rv))), dn01 = __builtin_bit_cast(uint32_t, (us2)(a - (a >> rv)));
            const uint32_t up2 = __builtin_bit_cast(uint32_t, (us2)(b + ((top - b) >> rv))), dn2 = __builtin_bit_cast(uint32_t, (us2)(b - (b >> rv)));
            const uint32_t m01 = s >= 2 ? 0xFFFFFFFFu : (s ? 0xFFFFu : 0u);   // halves with index < s
            const uint32_t n01 = (up01 & m01) | (dn01 & ~m01);
            const uint32_t cn1 = cn + 1 < 32u ? cn + 1 : 32u;
            const uint32_t n2 = ((s > 2 ? up2 : dn2) & 0xFFFFu) | (cn1 << 16);
            nrow = (uint64_t)n01 | ((uint64_t)n2 << 32);
            g_rc2.row[slot][lane] = nrow;
          }
          g_rc2.ring[k & 1][jj][lane] = ent;
          if (jj + 1 < RC_BATCH) { rw = slot_nx == slot ? nrow : rw_nx; slot = slot_nx; }
        }
      }
    }
    load_batch(q0, k + 3);
    trip_end();
  };
  auto code_trip = [&](const int k) __attribute__((always_inline)) {
    if (k > 0 && k <= nb) {
      const int base = (k - 1) * RC_BATCH;
      // the batch's 16 entries into registers at once (the reads are independent of the coder's state); an entry beyond this tile's
      // count is coded as "the whole range" (fl = 32768, fh = 0 of a one-symbol alphabet): it changes nothing and emits nothing
      uint32_t ce[RC_BATCH];
#pragma unroll
      for (int j = 0; j < RC_BATCH; j++) ce[j] = g_rc2.ring[(k - 1) & 1][j][lane];
#pragma unroll
      for (int j = 0; j < RC_BATCH; j++) {
        {
          const uint32_t ent = base + j < count ? ce[j] : ENT_RESOLVED(512, 0, 0);
          const uint32_t fl6 = (ent >> 14) & 0x3FF, fh6 = (ent >> 4) & 0x3FF, ns = ent & 15;
          // range update (od_ec_encode_q15, the mirror of spec §8.2.6), branch-free: fl6 == 512 <=> s == 0
          uint32_t l = low, r = rng;
          const uint32_t r8 = r >> 8;
          // (r8 < 2^8, fl6 / fh6 <= 512: 24-bit multiplies, full rate)
          const uint32_t v = (__umul24(r8, fh6) >> 1) + 4u * ns;
          const uint32_t u = fl6 >= 512 ? r : (__umul24(r8, fl6) >> 1) + 4u * ns + 4u;
          l += r - u;
          r = u - v;
          const int d = __builtin_clz(r) - 16;
          int s2 = cnt + d;
          if (s2 >= 0) {   // one byte, or two when at least 8 bits are ready (od_ec_enc_normalize)
            int c = cnt + 16;
            const bool two = s2 >= 8;
            // no inner branch: the first byte is stored in any case and, if it was not due, overwritten by the second at the same position
            PUT(out_pos, l >> c);
            out_pos += two;
            l = two ? l & ((1u << c) - 1) : l;
            c = two ? c - 8 : c;
            EMIT(l >> c);
            l &= (1u << c) - 1;
            s2 = c + d - 24;
          }
          low = l << d;
          rng = r << d;
          cnt = s2;
        }
      }
    }
    trip_end();
  };
  if (wave == 0) {
    for (int k = 0; k < n_trips; k += 3) { resolve_trip(k, qa); resolve_trip(k + 1, qb); resolve_trip(k + 2, qd); }
  } else {
    for (int k = 0; k < n_trips; k++) code_trip(k);
  }
  // ---- finish (od_ec_enc_done), coder wave
  if (wave == 1) {
    if (live && !overflow) {
      uint32_t l = low;
      int c = cnt, s = 10;
      const uint32_t m = 0x3FFF;
      uint32_t v = ((l + m) & ~m) | (m + 1);
      s += c;
      if (s > 0) {
        uint32_t n = (1u << (c + 16)) - 1;
        do {
          EMIT(v >> (c + 16));
          v &= n;
          s -= 8;
          c -= 8;
          n >>= 8;
        } while (s > 0);
      }
    }
    if (live) tile_bytes[tile] = overflow ? 0xFFFFFFFFu : (uint32_t)out_pos;
  }
}
#undef EMIT
#undef PUT

}  // namespace

extern "C" hipError_t av1mi_launch_entropy(const Av1miDevParams *P, const uint16_t *cdf_init, const int16_t *levels,
                                           const Av1miBlkInfo *blk, uint32_t *streams, uint32_t *stream_len, uint32_t *tile_combos,
                                           uint8_t *slots, uint32_t *tile_bytes, const uint8_t *lr_choice, uint32_t *tile_order /* n_tiles entries of scratch */,
                                           int frame0, int count /* frames [frame0, frame0 + count) of the chunk: all arrays are chunk-wide */,
                                           int phase /* bit 0: symbolize, bit 1: range-code (3 = both; the two may be given different frame ranges in two calls) */,
                                           hipStream_t stream, hipEvent_t mid,
                                           hipStream_t aux, hipEvent_t fork, hipEvent_t join /* aux != nullptr: the frame-edge tiles' variant runs there, beside the regular one */) {
  const int tpf = P->tile_rows * P->tile_cols;
  const int n_tiles = count * tpf, tile0 = frame0 * tpf;
  bool has_inter = false, has_key = false;
  for (int f = frame0; f < frame0 + count; f++) { if (av1mi_frame_is_inter(*P, f)) has_inter = true; else has_key = true; }
  // The two variants touch disjoint tiles.  The FULL one has few working waves (frame-edge tiles whose leaves the edge forces smaller)
  // with a long serial chain each - 0.14 ms per 60-frame chunk when it ran after the regular one; on a stream of its own it runs
  // beside it.
  hipStream_t fs = stream;
  if (!(phase & 1)) aux = nullptr;
  if (aux) {
    (void)hipEventRecord(fork, stream);
    (void)hipStreamWaitEvent(aux, fork, 0);
    fs = aux;
  }
#define SYM_LAUNCH(FULLV, INTERV, TSBV)                                                                                                   \
  hipLaunchKernelGGL((symbolize_tile_kernel<FULLV, INTERV, TSBV>), dim3(n_tiles), dim3(64), 0, (FULLV) ? fs : stream, *P, cdf_init, levels, blk, streams, \
                     stream_len, tile_combos, lr_choice, tile0)
  if (!(phase & 1)) {
  } else if (P->tile_sb == 1) {
    if (has_key) { SYM_LAUNCH(false, false, 1); SYM_LAUNCH(true, false, 1); }
    if (has_inter) { SYM_LAUNCH(false, true, 1); SYM_LAUNCH(true, true, 1); }
  } else {
    if (has_key) { SYM_LAUNCH(false, false, 2); SYM_LAUNCH(true, false, 2); }
    if (has_inter) { SYM_LAUNCH(false, true, 2); SYM_LAUNCH(true, true, 2); }
  }
#undef SYM_LAUNCH
  if (aux) {
    (void)hipEventRecord(join, aux);
    (void)hipStreamWaitEvent(stream, join, 0);
  }
  if (mid) (void)hipEventRecord(mid, stream);
  if (!(phase & 2)) return hipGetLastError();
  // Two forms of the range coder.  A wave of either wants a SIMD to itself (two workgroups' waves on one SIMD: twice as long,
  // measured), so a CU runs four waves at full speed: ONE workgroup of the four-stage form (~100 ns per entry of the longest tile) or
  // TWO of the two-stage form (~140 ns).  Measured range-coder times, two-stage / four-stage form, ms (tools/rc_probe.sh):
  //   1080p all-key x 30 (239 workgroups) 0.79 / 0.55     x 60 (478) 0.75 / 1.04     x 120 (956) 1.78 / 1.64     x 60 at CQ 8 2.57 / 2.88
  //   1080p IPPP x 60 (470)  0.68 / 0.53    at CQ 8  1.76 / 2.27    production point (CQ 8)  1.26 / 1.46    8K IPPP x 16 (510)  2.0 / 1.6
  //   4K x 30 (957) all-key  1.68 / 1.58    IPPP  0.84 / 0.74
  // Up to 256 workgroups the four-stage form runs them all at once and wins; beyond 512 both run in rounds and the faster workgroup
  // wins again.  Between the two the two-stage form still has everything resident while the four-stage form runs two rounds of
  // length-sorted workgroups, whose 64 equally long tiles cost more per entry (64 cache lines per stream load): that pays only where
  // few tiles are long - chunks with inter frames at a coarse quantiser (the P frames' tiles are short next to the key frame's).  The
  // quantiser index stands in for "few long tiles" here (the lengths themselves are on the device): >= 64.  AV1MI_RC_STAGES = 2 / 4
  // forces a form.
  // Order of the tiles: while all workgroups run at once the kernel lasts as long as its longest tile and the natural order is best
  // (a workgroup of 64 long tiles is slower per symbol than one long tile among short ones - measured 1.1 -> 2.1 ms).  In rounds
  // what counts is the sum over a CU's workgroups of their longest tile: tiles sorted by decreasing length (4K, 30 frames: 3.9 -> 2.9 ms).
  const char *const stages_str = getenv("AV1MI_RC_STAGES");   // (read per launch: the parity tests run both forms in one process)
  const int stages_env = stages_str ? atoi(stages_str) : 0;
  const int n_groups = (n_tiles + 63) / 64;
  const int stages = stages_env == 2 || stages_env == 4 ? stages_env : (n_groups <= 256 || n_groups > 512 || (has_inter && P->base_q_idx >= 64) ? 4 : 2);
  bool sorted = n_groups > (stages == 4 ? 256 : 512);
  if (const char *so = getenv("AV1MI_RC_SORT")) sorted = atoi(so) != 0;   // experiment knob: force the tile order on / off
  if (sorted) hipLaunchKernelGGL(tile_order_kernel, dim3(1), dim3(1024), 0, stream, n_tiles, stream_len + tile0, tile_order + tile0);
  if (stages == 4)
    hipLaunchKernelGGL(rangecode4_tiles_kernel, dim3(n_groups), dim3(256), 0, stream, *P, n_tiles, cdf_init, streams, stream_len,
                       tile_combos, slots, tile_bytes, sorted ? tile_order + tile0 : (uint32_t *)nullptr, tile0);
  else
    hipLaunchKernelGGL(rangecode2_tiles_kernel, dim3(n_groups), dim3(128), 0, stream, *P, n_tiles, cdf_init, streams, stream_len,
                       tile_combos, slots, tile_bytes, sorted ? tile_order + tile0 : (uint32_t *)nullptr, tile0);
  return hipGetLastError();
}
