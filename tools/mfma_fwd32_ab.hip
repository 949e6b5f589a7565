// A/B: forward 32x32 transform of a residual block, one wave per block, as (A) the butterfly networks the reconstruction kernel
// runs today (column pass, LDS transpose, row pass: tools/gen_txfm.py's av1_fdct32) and (B) an exact-integer matrix product on the
// matrix cores - Y = C * X * C^T with the 11-bit DCT matrix and the data both split into signed bytes, v_mfma_i32_32x32x32_i8.
// ("never measured".)  Standalone: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I av1-base_amd/csrc
// tools/mfma_fwd32_ab.hip -o av1-base_amd/ab/mfma_fwd32_ab ; run on the GPU box; prints ns per block for both and checks (B)
// against the same definition evaluated with 64-bit host arithmetic.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define AV1_TXFM_FN static __device__ __forceinline__
#define AV1_HALF_BTF_DEFINED
static __device__ __forceinline__ int32_t av1_half_btf(int32_t w0, int32_t a, int32_t w1, int32_t b) {
  return (__mul24(w0, a) + __mul24(w1, b) + 2048) >> 12;
}
#include "txfm_gen.h"

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

// ---------------------------------------------------------------- (A) butterflies, as tx_item does it
__global__ void __launch_bounds__(64) fwd_butterfly(const int16_t *__restrict__ resid, int32_t *__restrict__ out, int blocks_per_wave) {
  __shared__ int16_t scratch[32 * 33];
  const int lane = threadIdx.x;
  for (int b = 0; b < blocks_per_wave; b++) {
    const size_t blk = (size_t)blockIdx.x * blocks_per_wave + b;
    const int16_t *src = resid + blk * 1024;
    for (int p = lane; p < 1024; p += 64) scratch[(p >> 5) * 33 + (p & 31)] = src[p];
    __syncthreads();
    int32_t x[32];
    if (lane < 32) {
#pragma unroll
      for (int i = 0; i < 32; i++) x[i] = (int)scratch[i * 33 + lane] << 2;
      av1_fdct32(x);
#pragma unroll
      for (int i = 0; i < 32; i++) scratch[i * 33 + lane] = (int16_t)((x[i] + 8) >> 4);
    }
    __syncthreads();
    if (lane < 32) {
#pragma unroll
      for (int j = 0; j < 32; j++) x[j] = scratch[lane * 33 + j];
      av1_fdct32(x);
#pragma unroll
      for (int j = 0; j < 32; j++) out[blk * 1024 + lane * 32 + j] = x[j];
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------- (B) matrix cores
// Definition (what the host check evaluates): Cm[k][n] = round(4096 * o[k][n]), o = orthonormal 32-point DCT-II;
//   U[r][m] = (sum_c X[r][c] * Cm[m][c] + 512) >> 10          (row transform, 4 x orthonormal)
//   Y[k][m] = (sum_r Cm[k][r] * U[r][m] + 2048) >> 12         (column transform: 4 x the orthonormal 2-D DCT, the butterflies' scale)
// Signed byte split v = 256 * hi + lo, lo = (int8)(v & 255), hi = (v + 128) >> 8: exact for |v| < 2^15.
// Stage 1: A = X (lane = row r, k = column c), B = Cm^T.  Result U: column m on the lane, 16 rows in the accumulator registers.
// Stage 2: A = Cm with its k index in the order the accumulator registers hold U's rows, B = U straight from those registers.
struct MfmaTables { v4i s1_lo, s1_hi, s2_lo, s2_hi; };   // per lane: stage-1 B fragments (Cm^T), stage-2 A fragments (Cm, permuted k)

__device__ __forceinline__ int row_of_reg(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }

__global__ void __launch_bounds__(64) fwd_mfma(const int16_t *__restrict__ resid, int32_t *__restrict__ out, const MfmaTables *__restrict__ tab,
                                               int blocks_per_wave) {
  __shared__ __attribute__((aligned(16))) int16_t scratch[32 * 40];   // 80-byte rows: 16-byte aligned 128-bit reads
  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  const MfmaTables T = tab[lane];
  for (int b = 0; b < blocks_per_wave; b++) {
    const size_t blk = (size_t)blockIdx.x * blocks_per_wave + b;
    const int16_t *src = resid + blk * 1024;
    for (int p = lane; p < 1024; p += 64) scratch[(p >> 5) * 40 + (p & 31)] = src[p];
    __syncthreads();
    // ---- stage 1 operand: this lane's 16 residuals X[r][16h .. 16h+15], split into low and high bytes
    const v4i q0 = *reinterpret_cast<const v4i *>(&scratch[r * 40 + 16 * h]);
    const v4i q1 = *reinterpret_cast<const v4i *>(&scratch[r * 40 + 16 * h + 8]);
    v4i a_lo, a_hi;
    {
      const int d[8] = { q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w };
      int lo[4], hi[4];
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const unsigned e0 = (unsigned)d[2 * i], e1 = (unsigned)d[2 * i + 1];
        lo[i] = (int)__builtin_amdgcn_perm(e1, e0, 0x06040200u);          // low bytes of the four halfwords
        // (v + 128) >> 8 per halfword: add 128 to each halfword without carrying into the neighbour, take byte 1
        const unsigned g0 = ((e0 & 0x7FFF7FFFu) + 0x00800080u) ^ (e0 & 0x80008000u);
        const unsigned g1 = ((e1 & 0x7FFF7FFFu) + 0x00800080u) ^ (e1 & 0x80008000u);
        hi[i] = (int)__builtin_amdgcn_perm(g1, g0, 0x07050301u);
      }
      a_lo = (v4i){ lo[0], lo[1], lo[2], lo[3] };
      a_hi = (v4i){ hi[0], hi[1], hi[2], hi[3] };
    }
    v16i hh = {0}, mid = {0}, ll = {0};
    hh = __builtin_amdgcn_mfma_i32_32x32x32_i8(a_hi, T.s1_hi, hh, 0, 0, 0);
    mid = __builtin_amdgcn_mfma_i32_32x32x32_i8(a_hi, T.s1_lo, mid, 0, 0, 0);
    mid = __builtin_amdgcn_mfma_i32_32x32x32_i8(a_lo, T.s1_hi, mid, 0, 0, 0);
    ll = __builtin_amdgcn_mfma_i32_32x32x32_i8(a_lo, T.s1_lo, ll, 0, 0, 0);
    // ---- U = (hh << 16) + (mid << 8) + ll, rounded by 10 bits; split again; the 16 registers ARE the stage-2 B fragment
    int u_lo[4] = {0, 0, 0, 0}, u_hi[4] = {0, 0, 0, 0};
#pragma unroll
    for (int reg = 0; reg < 16; reg++) {
      const int u = ((hh[reg] << 16) + (mid[reg] << 8) + ll[reg] + 512) >> 10;
      u_lo[reg >> 2] |= (u & 255) << (8 * (reg & 3));
      u_hi[reg >> 2] |= (((u + 128) >> 8) & 255) << (8 * (reg & 3));
    }
    const v4i b_lo = { u_lo[0], u_lo[1], u_lo[2], u_lo[3] }, b_hi = { u_hi[0], u_hi[1], u_hi[2], u_hi[3] };
    v16i hh2 = {0}, mid2 = {0}, ll2 = {0};
    hh2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(T.s2_hi, b_hi, hh2, 0, 0, 0);
    mid2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(T.s2_hi, b_lo, mid2, 0, 0, 0);
    mid2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(T.s2_lo, b_hi, mid2, 0, 0, 0);
    ll2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(T.s2_lo, b_lo, ll2, 0, 0, 0);
#pragma unroll
    for (int reg = 0; reg < 16; reg++) {
      const int y = ((hh2[reg] << 16) + (mid2[reg] << 8) + ll2[reg] + 2048) >> 12;
      out[blk * 1024 + row_of_reg(reg, h) * 32 + r] = y;   // Y[k][m]: k = the register's row, m = the lane's column
    }
    __syncthreads();
  }
}

static int8_t lo8(int v) { return (int8_t)(v & 255); }
static int8_t hi8(int v) { return (int8_t)((v + 128) >> 8); }
static int pack4(const int8_t *b) { return (int)((uint32_t)(uint8_t)b[0] | ((uint32_t)(uint8_t)b[1] << 8) | ((uint32_t)(uint8_t)b[2] << 16) | ((uint32_t)(uint8_t)b[3] << 24)); }

int main(int argc, char **argv) {
  const int nblk = argc > 1 ? atoi(argv[1]) : 65536, bpw = 16, iters = 20;
  int Cm[32][32];
  for (int k = 0; k < 32; k++)
    for (int n = 0; n < 32; n++) {
      const double o = sqrt(2.0 / 32) * (k == 0 ? sqrt(0.5) : 1.0) * cos((2 * n + 1) * k * M_PI / 64);
      Cm[k][n] = (int)lrint(4096.0 * o);
    }
  std::vector<MfmaTables> tab(64);
  for (int lane = 0; lane < 64; lane++) {
    const int r = lane & 31, h = lane >> 5;
    int8_t l1[16], h1[16], l2[16], h2[16];
    for (int j = 0; j < 16; j++) {
      const int c1 = Cm[r][16 * h + j];                                                     // stage 1 B[k = c][col m = r] = Cm[m][c]
      l1[j] = lo8(c1); h1[j] = hi8(c1);
      const int kk = (j & 3) + 8 * (j >> 2) + 4 * h;                                         // stage 2: element j of half h sums over U's row kk
      const int c2 = Cm[r][kk];                                                             // A[row k = r][that k]
      l2[j] = lo8(c2); h2[j] = hi8(c2);
    }
    for (int i = 0; i < 4; i++) {
      tab[lane].s1_lo[i] = pack4(l1 + 4 * i); tab[lane].s1_hi[i] = pack4(h1 + 4 * i);
      tab[lane].s2_lo[i] = pack4(l2 + 4 * i); tab[lane].s2_hi[i] = pack4(h2 + 4 * i);
    }
  }
  std::vector<int16_t> X((size_t)nblk * 1024);
  uint64_t s = 88172645463325252ull;
  for (size_t i = 0; i < X.size(); i++) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    const int mag = (i / 1024) % 3 == 0 ? 1023 : ((i / 1024) % 3 == 1 ? 60 : 8);   // full-range, typical and small residuals
    X[i] = (int16_t)((int)(s % (2 * mag + 1)) - mag);
  }
  int16_t *dX; int32_t *dA, *dB; MfmaTables *dT;
  CHECK(hipMalloc(&dX, X.size() * 2)); CHECK(hipMalloc(&dA, X.size() * 4)); CHECK(hipMalloc(&dB, X.size() * 4)); CHECK(hipMalloc(&dT, sizeof(MfmaTables) * 64));
  CHECK(hipMemcpy(dX, X.data(), X.size() * 2, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dT, tab.data(), sizeof(MfmaTables) * 64, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  float msA = 0, msB = 0;
  const int grid = nblk / bpw;
  for (int which = 0; which < 2; which++) {
    for (int it = -2; it < iters; it++) {
      if (it == 0) CHECK(hipEventRecord(e0));
      if (which == 0) hipLaunchKernelGGL(fwd_butterfly, dim3(grid), dim3(64), 0, 0, dX, dA, bpw);
      else hipLaunchKernelGGL(fwd_mfma, dim3(grid), dim3(64), 0, 0, dX, dB, dT, bpw);
    }
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    CHECK(hipEventElapsedTime(which ? &msB : &msA, e0, e1));
  }
  std::vector<int32_t> A(X.size()), B(X.size());
  CHECK(hipMemcpy(A.data(), dA, A.size() * 4, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(B.data(), dB, B.size() * 4, hipMemcpyDeviceToHost));
  // host check of (B) on a sample of blocks, and how far (A) and (B) are from each other
  long bad = 0; int maxdiff = 0; double sumdiff = 0; long ndiff = 0;
  for (int blk = 0; blk < nblk; blk += (blk < 64 ? 1 : 257)) {
    const int16_t *x = &X[(size_t)blk * 1024];
    long long U[32][32];
    for (int r = 0; r < 32; r++)
      for (int m = 0; m < 32; m++) {
        long long acc = 0;
        for (int c = 0; c < 32; c++) acc += (long long)x[r * 32 + c] * Cm[m][c];
        U[r][m] = (acc + 512) >> 10;
      }
    for (int k = 0; k < 32; k++)
      for (int m = 0; m < 32; m++) {
        long long acc = 0;
        for (int r = 0; r < 32; r++) acc += (long long)Cm[k][r] * U[r][m];
        const int y = (int)((acc + 2048) >> 12);
        if (y != B[(size_t)blk * 1024 + k * 32 + m]) { if (bad < 5) fprintf(stderr, "blk %d (%d,%d): host %d gpu %d\n", blk, k, m, y, B[(size_t)blk * 1024 + k * 32 + m]); bad++; }
        const int d = abs(y - A[(size_t)blk * 1024 + k * 32 + m]);
        maxdiff = d > maxdiff ? d : maxdiff; sumdiff += d; ndiff++;
      }
  }
  printf("{\"blocks\": %d, \"butterfly_ns_per_block\": %.2f, \"mfma_ns_per_block\": %.2f, \"mfma_vs_host_mismatches\": %ld, "
         "\"max_abs_diff_butterfly_vs_matrix\": %d, \"mean_abs_diff\": %.4f}\n",
         nblk, msA * 1e6 / iters / nblk, msB * 1e6 / iters / nblk, bad, maxdiff, sumdiff / ndiff);
  return bad ? 1 : 0;
}
