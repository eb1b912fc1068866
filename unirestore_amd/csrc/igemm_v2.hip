// Instantiation unit of the implicit-GEMM launchers (see igemm_impl.h); dispatched from igemm.hip.
#include "igemm_impl.h"

namespace urk {
int URK(v2_256x32)(void* kp, hipStream_t s) { ConvK& k = *static_cast<ConvK*>(kp); return launch_glds<256, 32, 4, 1, 3>(k, s, 200); }
int URK(v2_128x64)(void* kp, hipStream_t s) { ConvK& k = *static_cast<ConvK*>(kp); return launch_glds<128, 64, 2, 2, 3>(k, s, 200); }
int URK(v2_256x160)(void* kp, hipStream_t s) { ConvK& k = *static_cast<ConvK*>(kp); return launch_glds<256, 160, 8, 1, 3>(k, s, 0); }
int URK(v2_256x128)(void* kp, hipStream_t s) { ConvK& k = *static_cast<ConvK*>(kp); return launch_glds<256, 128, 4, 2, 3>(k, s, 0); }
int URK(gemm_256x320_pair)(void* kp, hipStream_t s) { ConvK& k = *static_cast<ConvK*>(kp); return launch_gemm<256, 320, 8, 1, 2, false, true>(k, s); }
int URK(gemm_256x256)(void* kp, hipStream_t s) { ConvK& k = *static_cast<ConvK*>(kp); return launch_gemm<256, 256, 4, 2, 2>(k, s); }
}  // namespace urk
