// Instantiation unit of the implicit-GEMM launchers (see igemm_impl.h); dispatched from igemm.hip.
#include "igemm_impl.h"

namespace urk {
int URK(v1_128x128)(void* kp, hipStream_t s) { ConvK& k = *static_cast<ConvK*>(kp); return launch_cfg<128, 128, 2, 2>(k, s); }
int URK(v1_128x160)(void* kp, hipStream_t s) { ConvK& k = *static_cast<ConvK*>(kp); return launch_cfg<128, 160, 4, 1>(k, s); }
}  // namespace urk
