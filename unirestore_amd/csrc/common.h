// Shared device helpers + host-side error / profiling plumbing for libunirestore_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include <mutex>
#include <string>
#include <vector>

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


// ------------------------------------------------------------------------------------------ 16-bit activation types
// Every kernel that touches 16-bit activations / weights is a template on F16: false = bf16 (8-bit mantissa, fp32 range),
// true = IEEE fp16 (11-bit mantissa: ~8x smaller rounding error per stored tensor, range +-65504 - conversions overflow to inf).
// Both feed v_mfma_f32_32x32x16_{bf16,f16} at the same rate with fp32 accumulation.
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
template <bool F16> struct Act;
template <> struct Act<false> {
  static __device__ __forceinline__ float lo(uint32_t w) { return __uint_as_float(w << 16); }
  static __device__ __forceinline__ float hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
  static __device__ __forceinline__ float one(uint16_t v) { return bf2f(v); }
  static __device__ __forceinline__ uint32_t pack2(float a, float b) { return pack2bf(a, b); }
};
template <> struct Act<true> {
  static __device__ __forceinline__ float lo(uint32_t w) { return (float)__builtin_bit_cast(f16x2_t, w)[0]; }
  static __device__ __forceinline__ float hi(uint32_t w) { return (float)__builtin_bit_cast(f16x2_t, w)[1]; }
  static __device__ __forceinline__ float one(uint16_t v) { return (float)__builtin_bit_cast(_Float16, v); }
  // round-to-nearest-even, IEEE overflow: |v| > 65504 becomes +-inf (round 5; rounds 2-4 clamped to +-65504).  A clipped
  // activation is a silently wrong image; an inf reaches the next GroupNorm / softmax as NaN and the caller's finite check
  // (every public fp16 entry point of unirestore_amd.modules - DiffUIE.forward, Controller.forward, ControlledUNet.forward,
  // predict_z0, the adapters' operator-level forwards - checks its result and raises FloatingPointError) - overflow is loud, and the pack
  // is one instruction like bf16's.  NOT every inf survives to an output: -inf in an attention score becomes exp = 0, +-inf in front of the
  // 8-bit quantiser would clamp to 0 / 1 (the quantiser therefore propagates NaN and the check runs on its input's consumers); callers of
  // the raw C ABI get IEEE infs / NaNs in their tensors and check for themselves.
  static __device__ __forceinline__ uint32_t pack2(float a, float b) {
    f32x2_t v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2_t));
  }
};
template <bool F16> __device__ __forceinline__ uint16_t f2h16(float f) { return (uint16_t)(Act<F16>::pack2(f, 0.f) & 0xffffu); }
template <bool F16> __device__ __forceinline__ void unpack8t(const uint4& v, float* f) {
  f[0] = Act<F16>::lo(v.x); f[1] = Act<F16>::hi(v.x); f[2] = Act<F16>::lo(v.y); f[3] = Act<F16>::hi(v.y);
  f[4] = Act<F16>::lo(v.z); f[5] = Act<F16>::hi(v.z); f[6] = Act<F16>::lo(v.w); f[7] = Act<F16>::hi(v.w);
}
template <bool F16> __device__ __forceinline__ uint4 pack8t(const float* f) {
  return make_uint4(Act<F16>::pack2(f[0], f[1]), Act<F16>::pack2(f[2], f[3]), Act<F16>::pack2(f[4], f[5]), Act<F16>::pack2(f[6], f[7]));
}
// typed 8-element fragments (the element type reaches the IR: with raw uint4 fragments hipcc scheduled the halo conv's tap loop
// with 35 % more s_waitcnt instructions between the MFMAs and the kernel ran 7-10 % slower)
template <bool F16> struct Frag { typedef bf16x8 type; };
template <> struct Frag<true> { typedef f16x8 type; };
__device__ __forceinline__ f32x16 mfma16t(const bf16x8& a, const bf16x8& b, const f32x16& c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x16 mfma16t(const f16x8& a, const f16x8& b, const f32x16& c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
// one 32x32x16 MFMA step on 16-bit operands held as raw 128-bit fragments
template <bool F16> __device__ __forceinline__ f32x16 mfma16(const uint4& a, const uint4& b, const f32x16& c) {
  if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// host-side dispatch on a UR_DT_* value: BODY sees `constexpr bool F16`
#define UR_DT_SWITCH(dtype, ...)                                   \
  do {                                                             \
    if ((dtype) == UR_DT_F16) { constexpr bool F16 = true; __VA_ARGS__; } \
    else { constexpr bool F16 = false; __VA_ARGS__; }              \
  } while (0)
#define UR_REQUIRE_DT(dtype) UR_REQUIRE((dtype) == UR_DT_BF16 || (dtype) == UR_DT_F16, "dtype must be UR_DT_BF16 or UR_DT_F16")

// 16 / 8-byte WRITE-THROUGH stores (sc1): the bytes leave the XCD's L2 while the kernel runs instead of in the write-back at its
// end, and the line is dropped from that L2.  Site by site, same-box A/B of the whole forward (tools/ab_env.sh, ms per batch,
// every figure re-measured on correct data - a first pass ran on NaNs, see the s_nop below, and NaNs run 10 ms faster):
//   conv / GEMM staged epilogue (tile_copy) + GroupNorm apply output + split-K reduce (GN) output   281.8 -> 279.4   kept
//       (their consumers are halo convs / GEMMs on other CUs; staged epilogue for 3x3 launches only +3.7, for 1x1 only +0.9)
//   split-K partial planes                   273.8 -> 280.4   plain (the reduce pass re-reads them from L2 at once)
//   chain kernels' outputs (h0, q, k, y)     273.8 -> 275.7   plain
//   ping-pong attention output rows          273.8 -> 275.1   plain (the TAIL chain re-reads them at once)
//   `nt` instead of sc1 on the conv epilogue: no change.
typedef uint32_t ur_u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t ur_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void store16_wt(void* p, const uint4& v) {
#ifdef UR_NO_WT_STORES
  *reinterpret_cast<uint4*>(p) = v;
#else
  // (s_nop: a store of more than 8 bytes reads its data registers up to 2 wait states after issue, and hipcc's hazard
  //  recogniser does not see into the asm - without it the next VALU result can land in them first)
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(__builtin_bit_cast(ur_u32x4, v)) : "memory");
#endif
}
__device__ __forceinline__ void store8_wt(void* p, const uint2& v) {
#ifdef UR_NO_WT_STORES
  *reinterpret_cast<uint2*>(p) = v;
#else
  asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(__builtin_bit_cast(ur_u32x2, v)) : "memory");
#endif
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
int gn_stats_parts(int N, int HW, int C);
int gn_stats_launch(const void* x, float* part, int N, int HW, int C, int dtype, hipStream_t s);

// Once-per-DEVICE section: hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-device attribute, and one process may drive
// several GPUs from several host threads.  `if (auto once = x.first()) { ... }` runs the body in exactly one thread per device;
// every other thread that asks meanwhile BLOCKS until that body has finished (the guard releases the lock at the end of the
// if-statement), so nobody launches with > 64 KiB of dynamic LDS before the attribute is set.  Devices are keyed exactly (no
// aliasing above 63: one bit vector word per 64 devices).
struct DeviceOnce {
  std::mutex mu;
  std::atomic<unsigned long long> done[4] = {};           // one bit per device (256 devices): set once the guarded block has run there
  struct Guard {
    std::unique_lock<std::mutex> lk;
    std::atomic<unsigned long long>* word = nullptr;
    int dev = 0;
    bool run = false;
    Guard() = default;
    Guard(Guard&& o) noexcept : lk(std::move(o.lk)), word(o.word), dev(o.dev), run(o.run) { o.run = false; }
    explicit operator bool() const { return run; }
    ~Guard() {
      if (run) word->fetch_or(1ull << (dev & 63), std::memory_order_release);       // still under the lock: waiters see the finished state
    }
  };
  // Fast path (every launch after the first on a device): one acquire load, no lock - host threads driving different GPUs do not meet.
  Guard first() {
    int dev = 0;
    (void)hipGetDevice(&dev);
    Guard g;
    std::atomic<unsigned long long>& w = done[(dev >> 6) & 3];
    const unsigned long long bit = 1ull << (dev & 63);
    if (w.load(std::memory_order_acquire) & bit) return g;
    g.lk = std::unique_lock<std::mutex>(mu);
    g.run = !(w.load(std::memory_order_acquire) & bit);
    g.word = &w;
    g.dev = dev;
    if (!g.run) g.lk.unlock();
    return g;
  }
};

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
