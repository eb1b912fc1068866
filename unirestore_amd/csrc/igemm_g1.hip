// Instantiation unit: pure-GEMM LDS-DMA kernels at the small tile shapes (two-stage ring, two or more workgroups per CU).
#include "igemm_impl.h"

// ring depth per tile shape (stages of the LDS-DMA K ring); -D overrides are for A/B builds
#ifndef UR_NST_128x128
#define UR_NST_128x128 2
#endif
#ifndef UR_NST_128x160
#define UR_NST_128x160 2
#endif
#ifndef UR_NST_128x64
#define UR_NST_128x64 2
#endif
#ifndef UR_NST_64x64
#define UR_NST_64x64 2
#endif

namespace urk {
int URK(g1_128x128)(void* kp, hipStream_t s) { ConvK& k = *static_cast<ConvK*>(kp); return launch_gemm<128, 128, 2, 2, UR_NST_128x128, true>(k, s); }
int URK(g1_128x160)(void* kp, hipStream_t s) { ConvK& k = *static_cast<ConvK*>(kp); return launch_gemm<128, 160, 4, 1, UR_NST_128x160, true>(k, s); }
int URK(g1_128x64)(void* kp, hipStream_t s) { ConvK& k = *static_cast<ConvK*>(kp); return launch_gemm<128, 64, 2, 2, UR_NST_128x64, true>(k, s); }
int URK(g1_64x64)(void* kp, hipStream_t s) { ConvK& k = *static_cast<ConvK*>(kp); return launch_gemm<64, 64, 2, 2, UR_NST_64x64, true>(k, s); }
}  // namespace urk
