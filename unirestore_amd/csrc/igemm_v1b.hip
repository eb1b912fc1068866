// Instantiation unit of the implicit-GEMM launchers (see igemm_impl.h); dispatched from igemm.hip.
#include "igemm_impl.h"

namespace urk {
int URK(v1_128x64)(void* kp, hipStream_t s) { ConvK& k = *static_cast<ConvK*>(kp); return launch_cfg<128, 64, 2, 2>(k, s); }
int URK(v1_256x32)(void* kp, hipStream_t s) { ConvK& k = *static_cast<ConvK*>(kp); return launch_cfg<256, 32, 4, 1>(k, s); }
int URK(v1_64x64)(void* kp, hipStream_t s) { ConvK& k = *static_cast<ConvK*>(kp); return launch_cfg<64, 64, 2, 2>(k, s); }
}  // namespace urk
