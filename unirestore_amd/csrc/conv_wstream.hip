// Weight-streaming 3x3 convolution for the 8 x 8 feature maps (the UNet's bottom level: M = 64 * images rows, N = 1280, K = 9 * 1280
// or 9 * 2560).  At B = 8 such a layer multiplies 1.3 MB of activations with 29.5 - 59 MB of weights: it is a weight STREAM, and the
// tiled kernels (igemm_halo_img_kernel<8,8,4>) ran it at 0.8 - 1.0 TB/s because every tap's 16-KiB weight tile went through a 3-slot
// LDS ring behind a workgroup barrier - two tiles of lookahead against an HBM round trip (profiles/r5_e_per_shape_hip_events.txt:
// 34 us + a 7 - 12 us reduce pass over 26 MB of fp32 partial planes).
//
// Here nothing but the activations touches LDS and no wave ever waits for another one inside the K loop:
//   * workgroup = 4 waves (one per SIMD, 512 registers each) = 2 images (128 pixels) x 128 output channels x 4 input-channel chunks;
//     wave w owns chunk 4 * split + w (or a run of consecutive chunks, one after the other) COMPLETELY: its 2 x 10 x 10 x 64 halo patch (25 KiB, LDS-DMA'd once, private to the wave) and
//     its 144-KiB slice of the weights, which it reads straight into registers - fragment-major packing (ur_conv_desc.w_frag: one
//     coalesced 1-KiB global_load_dwordx4 = one MFMA A operand), eight (tap, k-step) units = 32 KiB per wave in flight, no barrier;
//   * the wave's 128 x 128 accumulator tile (16 MFMA tiles = 256 registers) sees each weight fragment once and each activation
//     fragment four times: 4 KiB of LDS reads per 16 MFMAs (1/8 of the LDS rate);
//   * the four chunk partials of a workgroup meet in LDS (two exchange rounds, fixed order), so the launch writes nchunk / 4 partial
//     planes instead of the tiled kernel's 10: 13 MB instead of 26 MB at 1280 -> 1280, B = 8;
//   * the 2-image tiles of one (channel tile, split) sit on ONE XCD in consecutive dispatch slots: the weights leave HBM once and the
//     other tiles read them from that XCD's L2.
// The split-K reduce pass of igemm_impl.h finishes the epilogue (bias rows, residual, GroupNorm partial sums) exactly as before.
// Reference layers: diffusers ResnetBlock2D.conv1 / conv2 of the SD-2.1 UNet's down_blocks[3], mid_block and up_blocks[0], called
// from /root/reference/src/modules/diffuie/base_model.py:137-160,184-198.
#include "igemm_impl.h"

namespace {

constexpr int WS_PW = 10, WS_HP = 100, WS_PIECES = 25, WS_PATCH = WS_PIECES * 1024;      // 2 images x 10 x 10 halo pixels x 128 B
constexpr int WS_LDS = 4 * 32768;                                                       // exchange round 1 (>= 4 patches = 102 400 B)
#ifndef UR_WS_TL
#define UR_WS_TL 0
#endif
constexpr int WS_D = 8;                                                                 // weight units (tap, k-step) in flight per wave

template <bool F16, int CPW>         // CPW = consecutive 64-channel chunks per wave (compile time: the chunk loop is unrolled, see below)
__global__ __launch_bounds__(256, 1) void conv3x3_wstream8_kernel(const ConvK p) {
  typedef typename Frag<F16>::type frag_t;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#if UR_WS_TL                                              // workgroup time line (A/B build): 100 MHz ticks -> ws + 32 Mi floats
  const unsigned long long tl0 = __builtin_amdgcn_s_memrealtime();
#endif
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  // blockIdx -> (image pair, channel tile, split): id % 8 = XCD; the tiles_m image pairs of one (channel tile, split) are consecutive
  // dispatch slots of one XCD
  const int id = blockIdx.x, xcd = id & 7, j = id >> 3;
  const int mt = j % p.tiles_m, g = (j / p.tiles_m) * 8 + xcd;
  if (g >= p.tiles_n * p.splitk) return;
  const int tn = g % p.tiles_n, sz = g / p.tiles_n;
  const int nchunk = p.nk / 9;
  const int c0 = (sz * 4 + wid) * CPW;                     // this wave's first chunk
  const int img0 = mt * 2, n0 = tn * 128;
  const int sperm = ((wid & 1) << 1) | (wid >> 1);         // accumulator index a holds channel fragment a ^ sperm (see the exchange)

  typedef __attribute__((address_space(3))) void* lptr_t;
  const unsigned patch_lds = (unsigned)(uintptr_t)(lptr_t)smem + wid * WS_PATCH;

  const uint4* __restrict__ wp = reinterpret_cast<const uint4*>(p.wf) + ((long long)(tn * nchunk + c0) * 144) * 64 + lane;
  const uint4* __restrict__ wpa[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) wpa[a] = wp + (a ^ sperm) * 64;
  uint4 wr[WS_D + 1][4];

  // ---- activations: the wave's halo patch of chunk c, LDS-DMA through a buffer descriptor (zero border = out-of-range lanes) ----
  auto load_patch = [&](int c) {
    int lane_o = lane;
    asm volatile("" : "+v"(lane_o));                       // opaque: a second call recomputes the 25 lane offsets instead of keeping them live (spills)
    const int lr = lane_o >> 3, ps = lane_o & 7;
    const int cb = c * 64;
    const bool second = cb >= p.C1;
    const uint16_t* src = second ? p.x2 : p.x;
    const int ld = second ? p.ldx2 : p.ldx;
    const ig_u32x4 rs = ig_make_rsrc(src, (unsigned long long)p.N * 64 * ld * 2);
    const unsigned soff = (unsigned)(second ? cb - p.C1 : cb) * 2u;
#pragma unroll
    for (int t = 0; t < WS_PIECES; ++t) {
      const int hr = t * 8 + lr;
      const int im = hr / WS_HP, rin = hr - im * WS_HP, hy = rin / WS_PW, hx = rin - hy * WS_PW;
      const int iy = hy - 1, ix = hx - 1;
      const bool v = img0 + im < p.N && (unsigned)iy < 8u && (unsigned)ix < 8u;
      const int chk = ps ^ (((rin >> 1) - hy) & 7);                  // slot swizzle of igemm_halo_img_kernel (conflict-free fragment reads)
      const unsigned vo = v ? (unsigned)(((img0 + im) * 8 + iy) * 8 + ix) * (unsigned)ld * 2u + chk * 16u : IG_OOB;
      ig_lds_dma16(patch_lds + t * 1024, vo, rs, soff);
    }
  };
  // the patch first, the first WS_D weight units behind it (plain loads: hipcc counts vmcnt for them): vmcnt(4 * WS_D) = the patch
  // has landed while the weights still fly
  load_patch(c0);
#pragma unroll
  for (int u = 0; u < WS_D; ++u)
#pragma unroll
    for (int a = 0; a < 4; ++a) wr[u][a] = wpa[a][u * 256];
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * WS_D) : "memory");
#if UR_WS_TL
  const unsigned long long tl1 = __builtin_amdgcn_s_memrealtime();
#endif

  f32x16 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

  const int frow = lane & 31, fhalf = lane >> 5;
  const unsigned char* patch = smem + wid * WS_PATCH;

  // ---- K loop: cpw chunks x (9 taps x 4 k-steps = 36 units, fully unrolled: the weight ring and the accumulators are registers).
  // The weight stream of consecutive chunks is contiguous and never drains; only the patch is switched between chunks.
  // (a run-time chunk loop carries the weight ring through phi nodes and hipcc answers with 592 B of scratch: CPW is a template argument)
#pragma unroll
  for (int cc = 0; cc < CPW; ++cc) {
    const bool more = cc + 1 < CPW;
    if (cc) {
      // every fragment read of the previous chunk has returned (the last unit's MFMAs consumed them); the DMA is invisible to hipcc
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      load_patch(c0 + cc);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    // fragment addresses from an opaque copy of the lane id per chunk: shared between the chunks they stay live (36 registers -> spills)
    int lane_o = lane;
    asm volatile("" : "+v"(lane_o));
    const int frow_o = lane_o & 31, fhalf_o = lane_o >> 5;
    int hrow0[4], hin0[4], hy0[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int pix = b * 32 + frow_o, im = pix >> 6, r = pix & 63, y = r >> 3, x = r & 7;
      hin0[b] = y * WS_PW + x;
      hrow0[b] = im * WS_HP + hin0[b];
      hy0[b] = y;
    }
    auto bfrag = [&](int tap, int ks, int b) -> frag_t {
      const int dy = tap / 3, dx = tap - dy * 3, sh = dy * WS_PW + dx;
      const int hsw = (((hin0[b] + sh) >> 1) - (hy0[b] + dy)) & 7;
      return *reinterpret_cast<const frag_t*>(patch + (hrow0[b] + sh) * 128 + (((ks * 2 + fhalf_o) ^ hsw) << 4));
    };
    frag_t bf[2][4];
#pragma unroll
    for (int b = 0; b < 4; ++b) bf[0][b] = bfrag(0, 0, b);
#pragma unroll
    for (int u = 0; u < 36; ++u) {
      if (u + WS_D < 36 || more) {                         // (the stream runs on into the next chunk's first units)
#pragma unroll
        for (int a = 0; a < 4; ++a) wr[(u + WS_D) % (WS_D + 1)][a] = wpa[a][(cc * 36 + u + WS_D) * 256];
      }
      if (u + 1 < 36) {
#pragma unroll
        for (int b = 0; b < 4; ++b) bf[(u + 1) & 1][b] = bfrag((u + 1) >> 2, (u + 1) & 3, b);
      }
      // (without the fences hipcc sinks every weight load to just before its first use - vmcnt(1) behind the load itself: the stream's
      //  whole latency exposed once per unit)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const frag_t af = __builtin_bit_cast(frag_t, wr[u % (WS_D + 1)][a]);
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = mfma16t(af, bf[u & 1][b], acc[a][b]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ---- the four chunk partials of the workgroup -> one: two exchange rounds through LDS, fixed order ((w + w^1) + (w^2 + w^3)) ------
  // accumulator a of wave w holds channel fragment a ^ sperm(w): every wave sends acc[2..3] to wave w ^ 1 (whose acc[0..1] are the same
  // fragments), then acc[1] to wave w ^ 2, and ends with the workgroup's sum of fragment sperm(w) in acc[0] - all indices static.
#if UR_WS_TL
  asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
  const unsigned long long tl2 = __builtin_amdgcn_s_memrealtime();
#endif
  __syncthreads();                                           // every wave is done with its patch
  {
    float4* mine = reinterpret_cast<float4*>(smem + wid * 32768);
#pragma unroll
    for (int a = 2; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          mine[(((a - 2) * 4 + b) * 4 + q) * 64 + lane] = make_float4(acc[a][b][4 * q], acc[a][b][4 * q + 1], acc[a][b][4 * q + 2], acc[a][b][4 * q + 3]);
    __syncthreads();
    const float4* theirs = reinterpret_cast<const float4*>(smem + (wid ^ 1) * 32768);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 t = theirs[((a * 4 + b) * 4 + q) * 64 + lane];
          acc[a][b][4 * q] += t.x; acc[a][b][4 * q + 1] += t.y; acc[a][b][4 * q + 2] += t.z; acc[a][b][4 * q + 3] += t.w;
        }
    __syncthreads();
    float4* mine2 = reinterpret_cast<float4*>(smem + wid * 16384);
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        mine2[(b * 4 + q) * 64 + lane] = make_float4(acc[1][b][4 * q], acc[1][b][4 * q + 1], acc[1][b][4 * q + 2], acc[1][b][4 * q + 3]);
    __syncthreads();
    const float4* theirs2 = reinterpret_cast<const float4*>(smem + (wid ^ 2) * 16384);
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 t = theirs2[(b * 4 + q) * 64 + lane];
        acc[0][b][4 * q] += t.x; acc[0][b][4 * q + 1] += t.y; acc[0][b][4 * q + 2] += t.z; acc[0][b][4 * q + 3] += t.w;
      }
  }

#if UR_WS_TL
  const unsigned long long tl3 = __builtin_amdgcn_s_memrealtime();
#endif
  // ---- partial plane `sz`: fp32 [M][Cout]; lane = pixel, 4 x 4 consecutive channels per accumulator tile ------------------------
  float* plane = p.ws + (long long)sz * p.M * p.Cout;
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const int m = img0 * 64 + b * 32 + frow;
    if (m < p.M) {
      float* row = plane + (long long)m * p.Cout + n0 + sperm * 32 + fhalf * 4;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<float4*>(row + q * 8) = make_float4(acc[0][b][4 * q], acc[0][b][4 * q + 1], acc[0][b][4 * q + 2], acc[0][b][4 * q + 3]);
    }
  }
#if UR_WS_TL
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lane == 0) {
    unsigned long long* o = reinterpret_cast<unsigned long long*>(p.ws + (32 << 20)) + (blockIdx.x * 4 + wid) * 8;
    o[0] = tl0; o[1] = tl1; o[2] = tl2; o[3] = tl3; o[4] = __builtin_amdgcn_s_memrealtime();
  }
#endif
}

}  // namespace

namespace urk {
// eligibility is decided by igemm.hip (dispatch_conv): 3x3 / stride 1 / pad 1 on 8 x 8 maps, chunk-major + fragment-major weights,
// Cout % 128 == 0, (C1 + C2) % 256 == 0, >= 8 chunks, a 16-bit staged output and a workspace for nchunk / 4 partial planes.
int URK(wstream_8x8)(void* kp, hipStream_t s) {
  ConvK& k = *static_cast<ConvK*>(kp);
  const int nchunk = k.nk / 9;
  k.tiles_m = (k.N + 1) / 2;
  k.tiles_n = k.Cout / 128;
  // one or two chunks per wave: two when that brings the launch down to one workgroup per CU (K = 9 x 2560 at B = 8: 400 -> 200)
  const long long wg1 = (long long)k.tiles_m * k.tiles_n * (nchunk / 4);
  const int cpw = (wg1 > 256 && nchunk % 8 == 0 && nchunk >= 16) ? 2 : 1;
  k.splitk = nchunk / (4 * cpw);
  k.nk_per_split = cpw * 36;                              // (the reduce pass does not read it)
  k.patch_tw = 0;
  k.prologue_ok = 0;
  set_gn_plan(k, false, 0);
  if (k.dry) { k.plan_tn = 1; return UR_OK; }
  static ur::DeviceOnce attr_once;      // the attribute is per device
  if (auto once_guard = attr_once.first())
  {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wstream8_kernel<UR_TU_F16 != 0, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, WS_LDS);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wstream8_kernel<UR_TU_F16 != 0, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, WS_LDS);
  }
  const int groups = (k.tiles_n * k.splitk + 7) / 8 * 8;
  if (cpw == 2) UR_F16_SWITCH(k, hipLaunchKernelGGL((conv3x3_wstream8_kernel<F16, 2>), dim3(groups * k.tiles_m), dim3(256), WS_LDS, s, k));
  else UR_F16_SWITCH(k, hipLaunchKernelGGL((conv3x3_wstream8_kernel<F16, 1>), dim3(groups * k.tiles_m), dim3(256), WS_LDS, s, k));
  launch_splitk_reduce(k, s);
  return ur::check_launch("ur_conv2d_nhwc");
}
}  // namespace urk
