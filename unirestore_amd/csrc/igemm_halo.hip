// Instantiation unit of the implicit-GEMM launchers (see igemm_impl.h); dispatched from igemm.hip.
#include "igemm_impl.h"

namespace urk {
int URK(halo_8x32_160)(void* kp, hipStream_t s) { ConvK& k = *static_cast<ConvK*>(kp); return launch_halo<8, 160, 8, 1>(k, s); }
int URK(halo_8x32_128)(void* kp, hipStream_t s) { ConvK& k = *static_cast<ConvK*>(kp); return launch_halo<8, 128, 4, 2>(k, s); }
int URK(halo_thin_32)(void* kp, hipStream_t s) { ConvK& k = *static_cast<ConvK*>(kp); return launch_halo_thin<8, 32, 8, 1>(k, s); }
int URK(himg_16x16)(void* kp, hipStream_t s) { ConvK& k = *static_cast<ConvK*>(kp); return launch_halo_img<16, 16, 1, 128, 4, 2>(k, s); }
int URK(himg_8x8x4)(void* kp, hipStream_t s) { ConvK& k = *static_cast<ConvK*>(kp); return launch_halo_img<8, 8, 4, 128, 4, 2>(k, s); }
}  // namespace urk
