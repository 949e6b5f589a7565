// A/B for the refinement kernel's SATD: the sum of the absolute 8x8 Hadamard coefficients of a 32x32 difference block,
// one wave per block, starting from where refine_eval's vertical pass leaves the difference - lane = column c, 16 rows 16h .. 16h+15
// in registers (av1-base_amd/csrc/me_kernel.hip refine_eval).
//   (A) the butterflies as shipped in round 1/2: difference -> LDS, rows (24 additions per 8 values) written back, columns, |.| sum;
//   (B) the matrix cores: Y = H' D^T H' with H' = I4 (x) H8 (entries +-1 / 0), the difference and the intermediate split into a
//       signed high byte and a biased low byte (v = 256 * (v >> 8) + ((v & 255) - 128) + 128; the + 128 is a constant matrix whose
//       transform is known and goes into the accumulator's initial value), two v_mfma_i32_32x32x32_i8 per side, no LDS, no barrier:
//       the vertical pass's registers ARE the first product's A fragment and the first product's accumulators ARE the second's B
//       fragment (k in the accumulator's row order).
// Both are exact integer arithmetic: (A) == (B) == the host's value, checked.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_satd_ab.hip -o av1-base_amd/ab/mfma_satd_ab ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

// candidates per staged block: the refinement evaluates 17 per leaf from one staged window; the difference of candidate `rep` here is
// the loaded one with `rep` XORed in, so that the global load (2 KB per block) does not bound the comparison
#define REPS 16

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int iabs(int v) { return v < 0 ? -v : v; }

__device__ __forceinline__ void hadamard8(int *v) {
  int t[8];
  t[0] = v[0] + v[4]; t[4] = v[0] - v[4]; t[1] = v[1] + v[5]; t[5] = v[1] - v[5]; t[2] = v[2] + v[6]; t[6] = v[2] - v[6]; t[3] = v[3] + v[7]; t[7] = v[3] - v[7];
  v[0] = t[0] + t[2]; v[2] = t[0] - t[2]; v[1] = t[1] + t[3]; v[3] = t[1] - t[3]; v[4] = t[4] + t[6]; v[6] = t[4] - t[6]; v[5] = t[5] + t[7]; v[7] = t[5] - t[7];
  t[0] = v[0] + v[1]; t[1] = v[0] - v[1]; t[2] = v[2] + v[3]; t[3] = v[2] - v[3]; t[4] = v[4] + v[5]; t[5] = v[4] - v[5]; t[6] = v[6] + v[7]; t[7] = v[6] - v[7];
#pragma unroll
  for (int i = 0; i < 8; i++) v[i] = t[i];
}

// the difference as the vertical pass holds it: d[j] = D[16h + j][c]
__device__ __forceinline__ void load_diff(const int16_t *__restrict__ src, int c, int h, int *d) {
#pragma unroll
  for (int j = 0; j < 16; j++) d[j] = src[(16 * h + j) * 32 + c];
}

__global__ void __launch_bounds__(64) satd_butterfly(const int16_t *__restrict__ diff, int *__restrict__ out, int blocks_per_wave) {
  constexpr int n = 32, NB = 4, HT = NB * NB * 8;
  __shared__ __attribute__((aligned(16))) int16_t dif[n * n];
  const int lane = threadIdx.x, c = lane & 31, h = lane >> 5;
  for (int b = 0; b < blocks_per_wave; b++) {
    const size_t blk = (size_t)blockIdx.x * blocks_per_wave + b;
    int d[16];
    load_diff(diff + blk * 1024, c, h, d);
    int total = 0;
    for (int rep = 0; rep < REPS; rep++) {
#pragma unroll
    for (int j = 0; j < 16; j++) dif[(16 * h + j) * n + c] = (int16_t)(d[j] ^ rep);
    __syncthreads();
    for (int task = lane; task < HT; task += 64) {
      const int sb = task >> 3, i = task & 7;
      int16_t *rowp = dif + ((sb / NB) * 8 + i) * n + (sb % NB) * 8;
      const uint4 q = *reinterpret_cast<const uint4 *>(rowp);
      int v[8] = { (int16_t)(q.x & 0xFFFF), (int16_t)(q.x >> 16), (int16_t)(q.y & 0xFFFF), (int16_t)(q.y >> 16),
                   (int16_t)(q.z & 0xFFFF), (int16_t)(q.z >> 16), (int16_t)(q.w & 0xFFFF), (int16_t)(q.w >> 16) };
      hadamard8(v);
      uint4 w;
      w.x = (uint32_t)(uint16_t)v[0] | ((uint32_t)(uint16_t)v[1] << 16); w.y = (uint32_t)(uint16_t)v[2] | ((uint32_t)(uint16_t)v[3] << 16);
      w.z = (uint32_t)(uint16_t)v[4] | ((uint32_t)(uint16_t)v[5] << 16); w.w = (uint32_t)(uint16_t)v[6] | ((uint32_t)(uint16_t)v[7] << 16);
      *reinterpret_cast<uint4 *>(rowp) = w;
    }
    __syncthreads();
    int satd = 0;
    for (int task = lane; task < HT; task += 64) {
      const int sb = task >> 3, j = task & 7;
      const int16_t *colp = dif + (sb / NB) * 8 * n + (sb % NB) * 8 + j;
      int v[8];
#pragma unroll
      for (int i = 0; i < 8; i++) v[i] = colp[i * n];
      hadamard8(v);
#pragma unroll
      for (int i = 0; i < 8; i++) satd += iabs(v[i]);
    }
    for (int o = 32; o > 0; o >>= 1) satd += __shfl_xor(satd, o, 64);
    __syncthreads();
    total += satd;
    }
    if (lane == 0) out[blk] = total;
  }
}

// 16 values of at most 15 bits + sign -> the fragment of their high bytes (v >> 8) and of their biased low bytes ((v & 255) - 128)
__device__ __forceinline__ void split16(const int *v, v4i &lo, v4i &hi) {
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const unsigned a = (unsigned)v[4 * i], b = (unsigned)v[4 * i + 1], c = (unsigned)v[4 * i + 2], d = (unsigned)v[4 * i + 3];
    const unsigned ab = __builtin_amdgcn_perm(b, a, 0x05040100u);   // a.b0 a.b1 b.b0 b.b1
    const unsigned cd = __builtin_amdgcn_perm(d, c, 0x05040100u);
    lo[i] = (int)(__builtin_amdgcn_perm(cd, ab, 0x06040200u) ^ 0x80808080u);
    hi[i] = (int)__builtin_amdgcn_perm(cd, ab, 0x07050301u);
  }
}

__device__ __forceinline__ int row_of_reg(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }

__global__ void __launch_bounds__(64) satd_mfma(const int16_t *__restrict__ diff, int *__restrict__ out, int blocks_per_wave) {
  const int lane = threadIdx.x, c = lane & 31, h = lane >> 5;
  // H' = I4 (x) H8, H8[i][j] = (-1)^popcount(i & j): symmetric.  Stage-1 B fragment: B[k = 16h + j][col = c] = H'[16h + j][c];
  // stage-2 A fragment: A[row = c][k <-> the accumulator's row row_of_reg(j, h)] = H'[c][row_of_reg(j, h)].
  v4i hb = {0, 0, 0, 0}, ha = {0, 0, 0, 0};
#pragma unroll
  for (int j = 0; j < 16; j++) {
    const int k1 = 16 * h + j, k2 = row_of_reg(j, h);
    const int e1 = (k1 >> 3) == (c >> 3) ? ((__builtin_popcount(k1 & c & 7) & 1) ? 0xFF : 0x01) : 0;
    const int e2 = (k2 >> 3) == (c >> 3) ? ((__builtin_popcount(k2 & c & 7) & 1) ? 0xFF : 0x01) : 0;
    hb[j >> 2] |= e1 << (8 * (j & 3));
    ha[j >> 2] |= e2 << (8 * (j & 3));
  }
  // the + 128 of the biased low bytes, transformed: stage 1 adds 128 * (column sum of H8) = 1024 where the lane's Hadamard index
  // is a multiple of 8; stage 2 adds 1024 on the accumulator rows that are multiples of 8 (registers 0, 4, 8, 12 of half 0)
  const int k1 = (c & 7) == 0 ? 1024 : 0, k2 = h == 0 ? 1024 : 0;
  for (int b = 0; b < blocks_per_wave; b++) {
    const size_t blk = (size_t)blockIdx.x * blocks_per_wave + b;
    int d[16];
    load_diff(diff + blk * 1024, c, h, d);
    int total = 0;
    for (int rep = 0; rep < REPS; rep++) {
    int dr[16];
#pragma unroll
    for (int j = 0; j < 16; j++) dr[j] = d[j] ^ rep;
    v4i lo, hi;
    split16(dr, lo, hi);
    v16i acc = {0};
    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(hi, hb, acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = (acc[r] << 8) + k1;
    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(lo, hb, acc, 0, 0, 0);
    int t[16];
#pragma unroll
    for (int r = 0; r < 16; r++) t[r] = acc[r];
    split16(t, lo, hi);
    v16i acc2 = {0};
    acc2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(ha, hi, acc2, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; r++) acc2[r] = (acc2[r] << 8) + ((r & 3) == 0 ? k2 : 0);
    acc2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(ha, lo, acc2, 0, 0, 0);
    int satd = 0;
#pragma unroll
    for (int r = 0; r < 16; r++) satd += iabs(acc2[r]);
    for (int o = 32; o > 0; o >>= 1) satd += __shfl_xor(satd, o, 64);
    total += satd;
    }
    if (lane == 0) out[blk] = total;
  }
}

int main(int argc, char **argv) {
  const int nblk = argc > 1 ? atoi(argv[1]) : 65536, bpw = 16, iters = 20;
  std::vector<int16_t> X((size_t)nblk * 1024);
  uint64_t s = 88172645463325252ull;
  for (size_t i = 0; i < X.size(); i++) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    const int mag = (i / 1024) % 3 == 0 ? 1023 : ((i / 1024) % 3 == 1 ? 60 : 8);   // full-range, typical and small differences
    X[i] = (int16_t)((int)(s % (2 * mag + 1)) - mag);
  }
  for (int i = 0; i < 1024; i++) X[i] = 1023;                                     // the extremes
  for (int i = 0; i < 1024; i++) X[1024 + i] = -1023;
  for (int i = 0; i < 1024; i++) X[2048 + i] = (int16_t)((((i >> 5) ^ i) & 1) ? 1023 : -1023);
  int16_t *dX; int *dA, *dB;
  CHECK(hipMalloc(&dX, X.size() * 2)); CHECK(hipMalloc(&dA, nblk * 4)); CHECK(hipMalloc(&dB, nblk * 4));
  CHECK(hipMemcpy(dX, X.data(), X.size() * 2, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  float msA = 0, msB = 0;
  const int grid = nblk / bpw;
  for (int which = 0; which < 2; which++) {
    for (int it = -2; it < iters; it++) {
      if (it == 0) CHECK(hipEventRecord(e0));
      if (which == 0) hipLaunchKernelGGL(satd_butterfly, dim3(grid), dim3(64), 0, 0, dX, dA, bpw);
      else hipLaunchKernelGGL(satd_mfma, dim3(grid), dim3(64), 0, 0, dX, dB, bpw);
    }
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    CHECK(hipEventElapsedTime(which ? &msB : &msA, e0, e1));
  }
  std::vector<int> A(nblk), B(nblk);
  CHECK(hipMemcpy(A.data(), dA, nblk * 4, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(B.data(), dB, nblk * 4, hipMemcpyDeviceToHost));
  long badA = 0, badB = 0;
  for (int blk = 0; blk < nblk; blk += (blk < 256 ? 1 : 61)) {
    const int16_t *x = &X[(size_t)blk * 1024];
    long want = 0;
    for (int rep = 0; rep < REPS; rep++)
    for (int by = 0; by < 4; by++)
      for (int bx = 0; bx < 4; bx++)
        for (int u = 0; u < 8; u++)
          for (int v = 0; v < 8; v++) {
            long acc = 0;
            for (int i = 0; i < 8; i++)
              for (int j = 0; j < 8; j++) {
                const int sgn = (__builtin_popcount(u & i) + __builtin_popcount(v & j)) & 1;
                const long xv = (int16_t)(x[(by * 8 + i) * 32 + bx * 8 + j] ^ rep);
                acc += sgn ? -xv : xv;
              }
            want += acc < 0 ? -acc : acc;
          }
    if (A[blk] != want) { if (badA < 3) fprintf(stderr, "butterfly blk %d: host %ld gpu %d\n", blk, want, A[blk]); badA++; }
    if (B[blk] != want) { if (badB < 3) fprintf(stderr, "mfma blk %d: host %ld gpu %d\n", blk, want, B[blk]); badB++; }
  }
  printf("{\"blocks\": %d, \"candidates_per_block\": %d, \"butterfly_ns_per_candidate\": %.2f, \"mfma_ns_per_candidate\": %.2f, \"butterfly_vs_host_mismatches\": %ld, \"mfma_vs_host_mismatches\": %ld}\n",
         nblk, REPS, msA * 1e6 / iters / nblk / REPS, msB * 1e6 / iters / nblk / REPS, badA, badB);
  return (badA || badB) ? 1 : 0;
}
