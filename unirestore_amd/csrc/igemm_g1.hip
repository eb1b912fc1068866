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
// deeper rings where the round-5 sweep (profiles/r5_wreg_ab.txt) found them: 64 x 64 / 4 stages for grids of <= 256 workgroups (the 8x8
// level: less than one workgroup per CU IS latency-bound: 512 x 1280 x 1280 12.9 -> 9.1 us), 128 x 64 / 3 stages unsplit for the long-K
// 16x16-level GEMMs (2048 x 1280 x 2560 25.9 -> 23.3 us, x 5120 49.1 -> 43.7 us against split-K + reduce)
int URK(g1_64x64_deep)(void* kp, hipStream_t s) { ConvK& k = *static_cast<ConvK*>(kp); return launch_gemm<64, 64, 2, 2, 4, true>(k, s); }
int URK(g1_128x64_deep)(void* kp, hipStream_t s) { ConvK& k = *static_cast<ConvK*>(kp); return launch_gemm<128, 64, 2, 2, 3, true>(k, s); }
#ifdef UR_AB_VARIANTS      // tools/ab_variants.sh: every tile shape x ring depth behind one entry point, chosen by UR_AB_ID at run time
int URK(g1_ab)(void* kp, hipStream_t s, int id) {
  ConvK& k = *static_cast<ConvK*>(kp);
  switch (id) {
    case 0: return launch_gemm<64, 64, 2, 2, 2, true>(k, s);
    case 1: return launch_gemm<64, 64, 2, 2, 3, true>(k, s);
    case 2: return launch_gemm<64, 64, 2, 2, 4, true>(k, s);
    case 3: return launch_gemm<128, 64, 2, 2, 2, true>(k, s);
    case 4: return launch_gemm<128, 64, 2, 2, 3, true>(k, s);
    case 5: return launch_gemm<128, 128, 2, 2, 2, true>(k, s);
    case 6: return launch_gemm<128, 128, 2, 2, 3, true>(k, s);
    case 7: return launch_gemm<128, 160, 4, 1, 2, true>(k, s);
    case 8: return launch_gemm<128, 160, 4, 1, 3, true>(k, s);
    case 9: return launch_gemm<128, 320, 4, 2, 2, true>(k, s);
    case 10: return launch_gemm<64, 128, 2, 2, 2, true>(k, s);
    case 11: return launch_gemm<64, 128, 2, 2, 4, true>(k, s);
    case 12: return launch_gemm<64, 320, 2, 2, 2, true>(k, s);
    case 13: return launch_gemm<64, 160, 2, 1, 2, true>(k, s);
    case 14: return launch_gemm<64, 160, 2, 1, 4, true>(k, s);
    case 15: return launch_gemm<64, 64, 2, 2, 6, true>(k, s);
    case 16: return launch_gemm<256, 128, 4, 2, 2, true>(k, s);
    case 17: return launch_gemm<128, 128, 2, 2, 4, true>(k, s);
  }
  return UR_E_UNSUPPORTED;
}
#endif
}  // namespace urk
