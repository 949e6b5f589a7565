// recon64_kernel.hip - the reconstruction kernels for block_log2 = 6: 64x64 luma blocks coded with the 64-point DCT (only the
// 32x32 low-frequency corner of the coefficients exists in AV1: spec §7.13.3), 32x32 chroma blocks - north_star's "4x4-64x64
// DCT/ADST"; SURVEY.md §8a row a10.  Same source as recon_kernel.hip, compiled with the larger LDS tiles and the 64-point
// networks (av1mi_launch_recon64); kept in a translation unit of its own so that the 32x32 kernels' occupancy stays what it is.
#define AV1MI_RECON_BIG 1
#include "recon_kernel.hip"
