// Shared device helpers + host-side error / profiling plumbing for libunirestore_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

#include "../../include/unirestore_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// ------------------------------------------------------------------------------------------ bf16
__device__ __forceinline__ float bf2f(uint16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
// fp32 -> bf16 round-to-nearest-even in hardware: one v_cvt_pk_bf16_f32 per pair (gfx950)
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
  f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ uint16_t f2bf(float f) { return (uint16_t)(pack2bf(f, 0.f) & 0xffffu); }
__device__ __forceinline__ void unpack8(const uint4& v, float* f) {
  f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
  f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
  f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
  f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
}
__device__ __forceinline__ uint4 pack8(const float* f) {
  return make_uint4(pack2bf(f[0], f[1]), pack2bf(f[2], f[3]), pack2bf(f[4], f[5]), pack2bf(f[6], f[7]));
}

// ------------------------------------------------------------------------------------------ math
// SiLU with the hardware reciprocal (1 ulp) instead of an IEEE division (~10 instructions)
__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }
// erf by Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7): kept for callers that need erf itself
__device__ __forceinline__ float erf_fast(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  float poly = fmaf(1.061405429f, t, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  const float r = 1.0f - poly * t * __expf(-ax * ax);
  return copysignf(r, x);
}
// Exact (erf) GELU = x * Phi(x), with Phi(x) ~ sigmoid(a x + b x^3 + c x^5): minimax fit over [-8, 8] (argument clamped
// there, Phi is saturated beyond), max |abs err| 2.5e-5 - two orders below bf16 resolution - in 7 full-rate instructions
// + exp2 + rcp.  (The A-S erf above costs 16 + 2; the GEGLU epilogue of a 32768 x 2560 GEMM spent 22 us in it.)
__device__ __forceinline__ float gelu_f(float x) {
  const float xc = __builtin_amdgcn_fmed3f(x, -8.0f, 8.0f);
  const float t = xc * xc;
  // -log2(e) * (1.59501577, 7.40112920e-2, -7.03033577e-4)
  float pz = fmaf(1.0142631e-3f, t, -0.10677573f);
  pz = fmaf(pz, t, -2.3011214f);
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(pz * xc));
}
__device__ __forceinline__ float apply_act(float x, int act) {
  switch (act) {
    case UR_ACT_SILU: return silu_f(x);
    case UR_ACT_GELU: return gelu_f(x);
    case UR_ACT_TANH: return tanhf(x);
    case UR_ACT_RELU: return fmaxf(x, 0.f);
    default: return x;
  }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ------------------------------------------------------------------------------------------ host
namespace ur {
void set_error(const std::string& msg);
int fail(int code, const std::string& msg);
int check_launch(const char* what);
// Zero `bytes` (multiple of 4) with a kernel node: hipMemsetAsync memset nodes misbehaved under hipGraph
// replay on small private-pool buffers (ROCm 7.2), so the library never emits memset nodes.
void zero_async(void* ptr, size_t bytes, hipStream_t s);
// per-(image, channel) sum / sum-of-squares of a dense NHWC bf16 tensor, accumulated into fp64 stats[N][C][2] (norms.hip)
int gn_stats_launch(const void* x, double* stats, int N, int HW, int C, hipStream_t s);

// Live timing: one (start, stop) hipEvent pair around each launch, on the launch stream.
struct ProfScope {
  ProfScope(const char* family, double flops, double bytes, hipStream_t s);
  ~ProfScope();
  int slot;
  hipStream_t stream;
};
}  // namespace ur

#define UR_REQUIRE(cond, msg) \
  do {                        \
    if (!(cond)) return ur::fail(UR_E_INVALID, std::string(__func__) + ": " + (msg)); \
  } while (0)
