// recon64_kernel.hip - the reconstruction kernels for leaf blocks up to 64x64 (block_log2 = 6), 16-bit samples: the same source as
// recon_kernel.hip compiled with AV1MI_RECON_BIG (64-point transforms, 64x64 LDS tiles, 2 waves per SIMD) in a translation unit of its
// own, so that the 32x32 kernels keep their LDS footprint and occupancy.
#define AV1MI_RECON_BIG 1
#include "recon_kernel.hip"
