// recon64_8_kernel.hip - leaf blocks up to 64x64, 8-bit samples (see recon64_kernel.hip and recon8_kernel.hip)
#define AV1MI_RECON_BIG 1
#define AV1MI_RECON_PIX8 1
#include "recon_kernel.hip"
