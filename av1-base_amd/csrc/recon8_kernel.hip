// recon8_kernel.hip - the reconstruction kernels for 8-bit samples (leaf blocks up to 32x32): recon_kernel.hip compiled with
// AV1MI_RECON_PIX8 in a translation unit of its own - the four (block size, sample type) units compile side by side.
#define AV1MI_RECON_PIX8 1
#include "recon_kernel.hip"
