// Flash attention for ONE wide head (d = 512: the VAE mid block, autoencoder.py:37-45 via diffusers' AttentionBlock) on gfx950.
//
// At d = 512 a wave cannot own both a query block's Q fragments and its whole output row (32 queries x 512 channels of fp32 is
// the entire register file), so the workgroup splits the two products differently:
//   * S^T = K.Q^T   is split over KEYS: wave w owns keys 16w..16w+15 of the 64-key tile for all 32 queries
//                   (v_mfma_f32_16x16x32: A = 16 key rows, B = Q^T held in registers for the whole kernel, 16 k-steps);
//   * O^T = V^T.P^T is split over CHANNELS: wave w owns channels 128w..128w+127 for all 32 queries
//                   (v_mfma_f32_32x32x16, 4 accumulator fragments); P^T goes through LDS once per tile (4 KB).
// The row maximum is exchanged through LDS (4 x 32 floats per tile); row sums stay per wave until the end.
// K (64 keys x 1 KB) and V^T (512 channels x 128 B) tiles are single LDS buffers filled by LDS-DMA: K(t+1) streams in under
// the softmax and the P.V product of tile t, V^T(t+1) under the K.Q^T product of tile t+1.  No T x T matrix ever exists.
#include "common.h"
#include "attention_params.h"

static __device__ uint4 g_attn512_zero_page[1];

namespace {

template <bool F16> struct Mfma32;
template <> struct Mfma32<false> {
  static __device__ __forceinline__ f32x4 run(const bf16x8& a, const bf16x8& b, const f32x4& c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};
template <> struct Mfma32<true> {
  static __device__ __forceinline__ f32x4 run(const f16x8& a, const f16x8& b, const f32x4& c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};

constexpr int A5_D = 512, A5_BK = 64, A5_BQ = 32;
constexpr int A5_KT = A5_BK * A5_D * 2;            // 64 KB: 64 key rows of 1 KB
constexpr int A5_VT = A5_D * A5_BK * 2;            // 64 KB: 512 channel rows of 128 B
constexpr int A5_P = A5_BQ * 128;                  // P^T tile: 32 query rows x 64 keys (16-bit)
constexpr int A5_LDS = A5_KT + A5_VT + A5_P + 4 * 32 * 4 + 32 * 4 + 4 * 32 * 4;

template <bool F16>
__global__ __launch_bounds__(256, 1) void attn512_kernel(const AttnP p) {
  constexpr int D = A5_D;
  typedef typename Frag<F16>::type frag_t;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* ks = smem;
  unsigned char* vs = smem + A5_KT;
  unsigned char* ps = vs + A5_VT;
  float* smax = reinterpret_cast<float*>(ps + A5_P);       // [4 waves][32 queries]
  float* salpha = smax + 128;                              // [32]
  float* ssum = salpha + 32;                               // [4][32]

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l15 = lane & 15, g4 = lane >> 4, l31 = lane & 31, hf = lane >> 5;
  const int bh = blockIdx.y, b = bh / p.H, h = bh - b * p.H;
  const int q0 = blockIdx.x * A5_BQ;
  const uint16_t* Q = p.q + b * p.bs_q + h * D;
  const uint16_t* K = p.k + b * p.bs_k + h * D;
  const uint16_t* V = p.vt + b * p.bs_vt + (long long)h * D * p.ldvt;

  // Q^T fragments (B operand of the 16x16x32 product): lane -> query qb*16 + l15, channels ds*32 + g4*8 .. +7
  frag_t qf[2][16];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const int qi = min(q0 + qb * 16 + l15, p.Tq - 1);
    const uint16_t* qp = Q + (long long)qi * p.ldq + g4 * 8;
#pragma unroll
    for (int ds = 0; ds < 16; ++ds) qf[qb][ds] = *reinterpret_cast<const frag_t*>(qp + ds * 32);
  }

  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  const uint16_t* zero = reinterpret_cast<const uint16_t*>(g_attn512_zero_page);
  // one LDS-DMA instruction = 1 KiB landing lane-linearly; the XOR swizzles live on the SOURCE address
  auto issue_k = [&](int kv0) {            // K: one key row (64 slots of 16 B) per instruction; this wave issues rows 16w..16w+15
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = wid * 16 + i;
      const int chunk = lane ^ (row & 15);
      const uint16_t* g = kv0 + row < p.Tk ? K + (long long)(kv0 + row) * p.ldk + chunk * 8 : zero;
      __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(ks + row * 1024), 16, 0, 0);
    }
  };
  const int lr = lane >> 3, pslot = lane & 7;
  auto issue_v = [&](int kv0) {            // V^T: 8 channel rows x 128 B per instruction; this wave issues pieces w, w+4, ...
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int piece = wid + 4 * i, row = piece * 8 + lr;
      const int chunk = pslot ^ ((row >> 1) & 7);
      const uint16_t* g = kv0 + chunk * 8 < p.ldvt ? V + (long long)row * p.ldvt + kv0 + chunk * 8 : zero;
      __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(vs + piece * 1024), 16, 0, 0);
    }
  };

  f32x16 oacc[4];
#pragma unroll
  for (int f = 0; f < 4; ++f)
#pragma unroll
    for (int e = 0; e < 16; ++e) oacc[f][e] = 0.f;
  float m_run[2] = {-1e30f, -1e30f}, l_part[2] = {0.f, 0.f};
  const float c = p.scale_log2e;

  const int ntiles = (p.Tk + A5_BK - 1) / A5_BK;
  issue_k(0);
  issue_v(0);
  asm volatile("s_waitcnt vmcnt(16)" ::: "memory");        // K(0) landed (V^T(0) may still be in flight)
  __builtin_amdgcn_s_barrier();
  for (int t = 0; t < ntiles; ++t) {
    const bool more = t + 1 < ntiles;
    // ---- S^T block: keys 16w..16w+15 of this tile x 32 queries ----------------------------------------------------
    f32x4 sacc[2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
      for (int e = 0; e < 4; ++e) sacc[qb][e] = 0.f;
    {
      const unsigned char* krow = ks + (wid * 16 + l15) * 1024;
#pragma unroll
      for (int ds = 0; ds < 16; ++ds) {
        const frag_t kf = *reinterpret_cast<const frag_t*>(krow + (((ds * 4 + g4) ^ l15) << 4));
        sacc[0] = Mfma32<F16>::run(kf, qf[0][ds], sacc[0]);
        sacc[1] = Mfma32<F16>::run(kf, qf[1][ds], sacc[1]);
      }
    }
    // accumulator register e of block qb: key t*64 + 16w + 4*g4 + e, query qb*16 + l15
    if ((t + 1) * A5_BK > p.Tk) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (t * A5_BK + wid * 16 + g4 * 4 + e >= p.Tk) { sacc[0][e] = -INFINITY; sacc[1][e] = -INFINITY; }
    }
    float mx[2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      float m = fmaxf(fmaxf(sacc[qb][0], sacc[qb][1]), fmaxf(sacc[qb][2], sacc[qb][3]));
      m = fmaxf(m, __shfl_xor(m, 16, 64));
      m = fmaxf(m, __shfl_xor(m, 32, 64));
      mx[qb] = m;
    }
    if (g4 == 0) { smax[wid * 32 + l15] = mx[0]; smax[wid * 32 + 16 + l15] = mx[1]; }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // K fragment reads retired, maxima written
    __builtin_amdgcn_s_barrier();                           // B1: nobody reads this K tile any more
    if (more) issue_k((t + 1) * A5_BK);
    float pr[2][4], alpha[2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      const int q = qb * 16 + l15;
      const float tm = fmaxf(fmaxf(smax[q], smax[32 + q]), fmaxf(smax[64 + q], smax[96 + q]));
      const float m_new = fmaxf(m_run[qb], tm);
      alpha[qb] = __builtin_amdgcn_exp2f((m_run[qb] - m_new) * c);
      m_run[qb] = m_new;
      const float mc = m_new * c;
      float sum = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        pr[qb][e] = __builtin_amdgcn_exp2f(fmaf(sacc[qb][e], c, -mc));
        sum += pr[qb][e];
      }
      l_part[qb] = l_part[qb] * alpha[qb] + sum;
      // P^T row of query q: keys 16w + 4*g4 .. +3 -> 8 bytes inside 16-B slot 2w + (g4 >> 1), rows XOR-swizzled like every 128-B row
      const int slot = (2 * wid + (g4 >> 1)) ^ ((q >> 1) & 7);
      *reinterpret_cast<uint2*>(ps + q * 128 + (slot << 4) + ((g4 & 1) << 3)) =
          make_uint2(Act<F16>::pack2(pr[qb][0], pr[qb][1]), Act<F16>::pack2(pr[qb][2], pr[qb][3]));
    }
    if (wid == 0 && g4 == 0) { salpha[l15] = alpha[0]; salpha[16 + l15] = alpha[1]; }
    if (more) asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");     // V^T(t) landed (K(t+1) may be in flight), P^T written
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                           // B2
    // ---- O^T slice (channels 128w..128w+127) += V^T . P^T -----------------------------------------------------------
    {
      const float a = salpha[l31];
      if (!__all(a == 1.f)) {
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
          for (int e = 0; e < 16; ++e) oacc[f][e] *= a;
      }
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const frag_t pf = *reinterpret_cast<const frag_t*>(ps + l31 * 128 + (((kk * 2 + hf) ^ ((l31 >> 1) & 7)) << 4));
#pragma unroll
        for (int f = 0; f < 4; ++f) {
          const int row = wid * 128 + f * 32 + l31;
          const frag_t vf = *reinterpret_cast<const frag_t*>(vs + row * 128 + (((kk * 2 + hf) ^ ((row >> 1) & 7)) << 4));
          oacc[f] = mfma16t(vf, pf, oacc[f]);
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // K(t+1) landed; V^T / P^T / alpha reads retired
    __builtin_amdgcn_s_barrier();                           // B3
    if (more) issue_v((t + 1) * A5_BK);
  }

  // ---- row sums: per wave partials -> LDS -> every lane's query; normalise and store ---------------------------------
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    float l = l_part[qb];
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    if (g4 == 0) ssum[wid * 32 + qb * 16 + l15] = l;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  const float inv = 1.f / (ssum[l31] + ssum[32 + l31] + ssum[64 + l31] + ssum[96 + l31]);
  const int qi = q0 + l31;
  if (qi < p.Tq) {
    uint16_t* op = p.o + b * p.bs_o + (long long)qi * p.ldo + h * D + wid * 128;
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int ch = f * 32 + 8 * g + 4 * hf;
        *reinterpret_cast<uint2*>(op + ch) =
            make_uint2(Act<F16>::pack2(oacc[f][g * 4] * inv, oacc[f][g * 4 + 1] * inv),
                       Act<F16>::pack2(oacc[f][g * 4 + 2] * inv, oacc[f][g * 4 + 3] * inv));
      }
  }
}

}  // namespace

#ifndef UR_TU_F16
#define UR_TU_F16 0
#endif
#if UR_TU_F16
#define UR_ATTN512_LAUNCH ur_attn512_launch_f16
#else
#define UR_ATTN512_LAUNCH ur_attn512_launch_bf16
#endif

int UR_ATTN512_LAUNCH(const void* pp, hipStream_t s) {
  const AttnP& p = *static_cast<const AttnP*>(pp);
  constexpr bool F16 = UR_TU_F16 != 0;
  static ur::DeviceOnce attr_once;      // the attribute is per device
  if (auto once_guard = attr_once.first()) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&attn512_kernel<F16>), hipFuncAttributeMaxDynamicSharedMemorySize, A5_LDS);
  }
  dim3 grid((p.Tq + A5_BQ - 1) / A5_BQ, p.B * p.H), block(256);
  hipLaunchKernelGGL((attn512_kernel<F16>), grid, block, A5_LDS, s, p);
  return ur::check_launch("ur_attention_fwd");
}
