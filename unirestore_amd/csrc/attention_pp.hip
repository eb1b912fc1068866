// Ping-pong flash attention forward for head dim 64 on gfx950 (round 4): the kernel behind ur_attention_fwd for the
// self-attention shapes (Tq % 256 == 0, Tk % 256 == 0); attention.hip keeps every other shape (cross-attention's 77 keys, d = 128).
//
// Why a second kernel.  attn_fwd_kernel (attention.hip) runs 3 independent waves per SIMD through the same per-tile program
// (QK^T MFMAs -> softmax VALU -> PV MFMAs -> barrier); its matrix pipe is busy 37.5 % of the time at the clock the part holds
// (profiles/r2_c_pmc_attention.txt): a wave's 16 MFMAs (512 cycles) and its ~207 other issue slots (x 4 cycles) ADD UP instead of
// overlapping, because the waves of a SIMD stay in phase.  This kernel fixes both terms:
//   * the two waves of a SIMD (wave w of group A, wave w + 4 of group B; one 8-wave workgroup per CU) are held HALF A TILE APART
//     by the workgroup barriers: while A multiplies (P.V of tile t, K.Q^T of tile t + 1) B runs the softmax of its tile t on the
//     VALU, then they swap - matrix pipe and VALU of a SIMD are both busy in every phase;
//   * the softmax of the common case is 2.6 VALU per score instead of 4.5: Q arrives pre-multiplied by scale*log2(e), the
//     "- max" is the MFMA's C operand (the S accumulator is initialised with -m instead of 0), and there is NO row-maximum pass:
//     the exponentials are taken against the running reference m and the lane's partial row sum doubles as the overflow test
//     (every p < 2^14 when the lane's sum of 32 is - the 16-bit P operand and the fp32 accumulators have that much head-room);
//     only a wave that fails the test (or is at tile 0) takes the slow path: maximum, new reference, O / l rescale, recompute.
// Layout (as attention.hip): S^T = K.Q^T so a lane owns one query column; K rows read through the bit-2/3 swap so that the S
// accumulator registers, packed pairwise, ARE the P^T B operand; V arrives transposed ([channel][key]); 64-key K / V^T tiles
// land in XOR-swizzled LDS by buffer-descriptor LDS-DMA (one 1-KiB piece of each per wave and tile), four stage buffers
// {V^T(t) | K(t+1)} filled three tiles ahead, counted by hand.  The tile loop and the epilogue are ONE generated asm statement
// with hand-owned registers (tools/gen_attn_asm.py -> attention_pp_asm.inc; its docstring has the schedule and the measurements);
// this file computes what enters it: descriptors, per-lane offsets, the Q fragments, fragment addresses, the first stages' DMA.
#include "common.h"
#include "attention_params.h"
#include <cstdlib>
#ifndef UR_ATTN_PP_INC
#define UR_ATTN_PP_INC "attention_pp_asm.inc"       // (A/B builds substitute a variant of the generated block: tools/ab_attn.sh)
#endif
#include UR_ATTN_PP_INC

namespace {

typedef uint32_t pp_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int pp_swap23(int i) { return (i & 0x13) | ((i & 4) << 1) | ((i & 8) >> 1); }
__device__ __forceinline__ pp_u32x4 pp_rsrc(const void* base) {
  const unsigned long long a = (unsigned long long)base;
  pp_u32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((unsigned)a);
  r[1] = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffffu);          // stride 0: raw buffer
  r[2] = 0xffffff00u;                                                             // shapes are exact multiples: no range check needed
  r[3] = 0x00020000u;
  return r;
}
// one 1-KiB LDS-DMA piece: LDS bytes [m0v, m0v + 1024) <- 64 lanes x 16 bytes at base + voff + soff
__device__ __forceinline__ void pp_dma(unsigned m0v, unsigned voff, const pp_u32x4& rs, unsigned soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(__builtin_amdgcn_readfirstlane(m0v)), "v"(voff), "s"(rs),
               "s"(__builtin_amdgcn_readfirstlane(soff)) : "memory");
}
#define ATTN_PP_ASM_SEL ATTN_PP_ASM

template <bool F16>
__global__ __launch_bounds__(512) void attn_pp64_kernel(const AttnP p) {
  constexpr int TILE = 8192;                  // 64 rows x 128 B
  constexpr int STAGE = 2 * TILE;             // V^T(t) | K(t + 1); four stage buffers
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef __attribute__((address_space(3))) void* lptr_t;
  typedef typename Frag<F16>::type frag_t;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wid >> 2;
  const int l31 = lane & 31, hf = lane >> 5;
  // work item: a (batch-head, query tile) with all keys, or - behind the n_full whole ones - one key half of such a tile
  const int wg = blockIdx.x, nq = p.Tq >> 8;
  const int half = wg < p.n_full ? -1 : ((wg - p.n_full) & 1);
  // workgroup b runs on XCD b % 8: every XCD gets a contiguous range of tiles, so the query tiles of one head share one L2
  // (141.9 -> 136.8 us on the 4-head launch against the plain order)
  const int tile = wg < p.n_full ? ((p.n_full & 7) == 0 ? (wg & 7) * (p.n_full >> 3) + (wg >> 3) : wg) : p.n_full + ((wg - p.n_full) >> 1);
  const int bh = tile / nq, qt = tile - bh * nq, b = bh / p.H, h = bh - b * p.H;
  const int q0 = qt * 256 + wid * 32;
  const int nt = half < 0 ? (p.Tk >> 6) : (p.Tk >> 7);
  const int kt0 = half > 0 ? nt : 0;                          // first key tile of this work item

  const uint16_t* Q = p.q + b * p.bs_q + h * 64;
  const uint16_t* K = p.k + b * p.bs_k + h * 64 + (long long)kt0 * 64 * p.ldk;
  const uint16_t* V = p.vt + b * p.bs_vt + (long long)h * 64 * p.ldvt + kt0 * 64;

  // ---- LDS-DMA bookkeeping: wave w moves piece w (8 rows) of every K and V^T tile -----------------------------------------
  const pp_u32x4 rs_k = pp_rsrc(K), rs_v = pp_rsrc(V);
  const unsigned smem_lds = (unsigned)(uintptr_t)(lptr_t)smem;
  const unsigned kstep = 128u * (unsigned)p.ldk;                     // bytes per 64-key tile of K
  unsigned kvo, vvo;
  {
    const int prow = wid * 8 + (lane >> 3), chunk = (lane & 7) ^ ((prow >> 1) & 7);
    kvo = ((unsigned)prow * (unsigned)p.ldk + chunk * 8u) * 2u;
    vvo = ((unsigned)prow * (unsigned)p.ldvt + chunk * 8u) * 2u;
  }
  auto issue = [&](int s) {                                          // stage s = V^T(s) | K(s + 1), buffer s & 3
    const unsigned base = smem_lds + (unsigned)(s & 3) * STAGE + (unsigned)wid * 1024u;
    if (s >= 0) pp_dma(base, vvo, rs_v, (unsigned)s * 128u);
    if (s + 1 < nt) pp_dma(base + TILE, kvo, rs_k, (unsigned)(s + 1) * kstep);
  };
#pragma unroll
  for (int s = -1; s < ATTN_PP_DIST; ++s) issue(s);

  // ---- Q fragments (B operand), pre-multiplied by scale * log2(e): lane -> query q0 + l31, dims ds*16 + hf*8 .. +7 --------
  frag_t qf[4];
  {
    const uint16_t* qp = Q + (long long)(q0 + l31) * p.ldq + hf * 8;
    const float c = p.scale_log2e;
#pragma unroll
    for (int ds = 0; ds < 4; ++ds) {
      uint4 raw = *reinterpret_cast<const uint4*>(qp + ds * 16);
      if (c != 1.0f) {
        float f[8];
        unpack8t<F16>(raw, f);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] *= c;
        raw = pack8t<F16>(f);
      }
      qf[ds] = __builtin_bit_cast(frag_t, raw);
    }
  }

  // fragment addresses inside a tile: k-step / key-step s is  base ^ (s << 5)  (the XOR swizzle is (row >> 1) & 7 on 16-byte slots)
  unsigned ka[2], va[2];
#pragma unroll
  for (int f = 0; f < 2; ++f) {
    const int kr = f * 32 + pp_swap23(l31), vr = f * 32 + l31;
    ka[f] = TILE + kr * 128 + ((hf ^ ((kr >> 1) & 7)) << 4);
    va[f] = vr * 128 + ((hf ^ ((vr >> 1) & 7)) << 4);
  }

  // ---- tile loop + epilogue: one hand-scheduled asm statement (tools/gen_attn_asm.py; the registers named in ATTN_PP_CLOBBERS are its own) ----------
  {
    // output row of the lane: the 16-bit tensor, or (key halves) a 68-float workspace row = 64 fp32 channels | m | l | pad
    uint16_t* optr = half < 0 ? p.o + b * p.bs_o + (long long)(q0 + l31) * p.ldo + h * 64 + 4 * hf
                              : reinterpret_cast<uint16_t*>(p.ws + ((long long)(wg - p.n_full) * 256 + wid * 32 + l31) * 68 + 4 * hf);
    const unsigned part_s = __builtin_amdgcn_readfirstlane(half < 0 ? 0u : 1u);
    const unsigned ldsb = __builtin_amdgcn_readfirstlane(smem_lds + (unsigned)wid * 1024u);
    const unsigned nt_s = __builtin_amdgcn_readfirstlane((unsigned)nt), grp_s = __builtin_amdgcn_readfirstlane((unsigned)grp);
    const unsigned kstep_s = __builtin_amdgcn_readfirstlane(kstep);
    // absolute LDS byte addresses of the fragments in stage buffer 0 (the block adds buffer / tile offsets as immediates and
    // derives the k-steps with v_xor: smem_lds is a multiple of 128, dynamic LDS starts the allocation)
    const unsigned ka_a[2] = {smem_lds + ka[0] - TILE, smem_lds + ka[1] - TILE}, va_a[2] = {smem_lds + va[0], smem_lds + va[1]};
    unsigned t0, t1, t2, t3, t4;
#ifdef UR_ATTN_PP_DBG          // phase-timer build (tools/ab_attn.sh with ATTN_DBG=1): five more SGPR temporaries
    unsigned d0, d1, d2, d3, d4, d5, d6;
#define PP_DBG_OUT , "=&s"(d0), "=&s"(d1), "=&s"(d2), "=&s"(d3), "=&s"(d4), "=&s"(d5), "=&s"(d6)
#else
#define PP_DBG_OUT
#endif
    if constexpr (F16) {
      asm volatile(ATTN_PP_ASM_SEL("v_mfma_f32_32x32x16_f16", "v_cvt_pk_f16_f32", "v_dot2c_f32_f16", "0x3c003c00")
                   : "=&s"(t0), "=&s"(t1), "=&s"(t2), "=&s"(t3), "=&s"(t4) PP_DBG_OUT
                   : "v"(qf[0]), "v"(qf[1]), "v"(qf[2]), "v"(qf[3]), "v"(ka_a[0]), "v"(ka_a[1]), "v"(va_a[0]), "v"(va_a[1]), "v"(kvo), "v"(vvo),
                     "v"(optr), "s"(rs_k), "s"(rs_v), "s"(kstep_s), "s"(nt_s), "s"(grp_s), "s"(ldsb), "s"(part_s)
                   : ATTN_PP_CLOBBERS);
    } else {
      asm volatile(ATTN_PP_ASM_SEL("v_mfma_f32_32x32x16_bf16", "v_cvt_pk_bf16_f32", "v_dot2c_f32_bf16", "0x3f803f80")
                   : "=&s"(t0), "=&s"(t1), "=&s"(t2), "=&s"(t3), "=&s"(t4) PP_DBG_OUT
                   : "v"(qf[0]), "v"(qf[1]), "v"(qf[2]), "v"(qf[3]), "v"(ka_a[0]), "v"(ka_a[1]), "v"(va_a[0]), "v"(va_a[1]), "v"(kvo), "v"(vvo),
                     "v"(optr), "s"(rs_k), "s"(rs_v), "s"(kstep_s), "s"(nt_s), "s"(grp_s), "s"(ldsb), "s"(part_s)
                   : ATTN_PP_CLOBBERS);
    }
  }
}

// The two key halves of a split tile -> the 16-bit output: o = (w1 O1 + w2 O2) / (w1 l1 + w2 l2), w_i = 2^(m_i - max(m1, m2)).
// One thread per (query row, 8 channels); fixed order, no atomics.
template <bool F16>
__global__ __launch_bounds__(256) void attn_pp_combine_kernel(const AttnP p, int n_split) {
  const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
  const int c8 = (int)(g & 7);
  const long long row = g >> 3;                              // tile-major: split tile index * 256 + row in tile
  if (row >= (long long)n_split * 256) return;
  const int st = (int)(row >> 8), r = (int)(row & 255), nq = p.Tq >> 8;
  const int tile = p.n_full + st, bh = tile / nq, qt = tile - bh * nq, b = bh / p.H, h = bh - b * p.H;
  const float* r1 = p.ws + ((long long)(2 * st) * 256 + r) * 68;
  const float* r2 = r1 + 256 * 68;
  const float m1 = r1[64], l1 = r1[65], m2 = r2[64], l2 = r2[65];
  const float m = fmaxf(m1, m2);
  const float w1 = __builtin_amdgcn_exp2f(m1 - m), w2 = __builtin_amdgcn_exp2f(m2 - m);
  const float inv = 1.f / (w1 * l1 + w2 * l2);
  const float4 a0 = *reinterpret_cast<const float4*>(r1 + c8 * 8), a1 = *reinterpret_cast<const float4*>(r1 + c8 * 8 + 4);
  const float4 b0 = *reinterpret_cast<const float4*>(r2 + c8 * 8), b1 = *reinterpret_cast<const float4*>(r2 + c8 * 8 + 4);
  const float s1 = w1 * inv, s2 = w2 * inv;
  uint4 o;
  o.x = Act<F16>::pack2(a0.x * s1 + b0.x * s2, a0.y * s1 + b0.y * s2);
  o.y = Act<F16>::pack2(a0.z * s1 + b0.z * s2, a0.w * s1 + b0.w * s2);
  o.z = Act<F16>::pack2(a1.x * s1 + b1.x * s2, a1.y * s1 + b1.y * s2);
  o.w = Act<F16>::pack2(a1.z * s1 + b1.z * s2, a1.w * s1 + b1.w * s2);
  *reinterpret_cast<uint4*>(p.o + b * p.bs_o + (long long)(qt * 256 + r) * p.ldo + h * 64 + c8 * 8) = o;
}

}  // namespace

// Built twice (-DUR_TU_F16=0 / 1), one object per 16-bit type (see attention.hip).
#ifndef UR_TU_F16
#define UR_TU_F16 0
#endif
#if UR_TU_F16
#define UR_ATTN_PP_LAUNCH ur_attn_pp_launch_f16
#else
#define UR_ATTN_PP_LAUNCH ur_attn_pp_launch_bf16
#endif

// Workgroups per launch and how many of the trailing query tiles are split in two key halves.  One workgroup per CU: a grid of
// n = whole rounds of 256 + r with 0 < r <= 128 leaves half of the chip idle for a whole tile time (B = 8, 5 heads, T = 4096:
// 640 = 2.5 rounds); splitting the keys of those r tiles gives 2r <= 256 half-length workgroups that fill the last round.
static void pp_plan(const AttnP& p, size_t ws_bytes, int& n_full, int& n_split) {
  const long long n = (long long)(p.Tq / 256) * p.B * p.H, r = attn_pp_split_tiles(p.B, p.H, p.Tq, p.Tk, 64);      // (attention_params.h)
  n_split = (r && p.ws && ws_bytes >= attn_pp_ws_bytes(r)) ? (int)r : 0;
  n_full = (int)(n - n_split);
}

int UR_ATTN_PP_LAUNCH(const void* pp, size_t ws_bytes, hipStream_t s) {
  AttnP p = *static_cast<const AttnP*>(pp);
  // the asm loop has no tails and its buffer descriptors no range check: refuse what the dispatcher should never send
  if (!attn_pp_shape_ok(p, 64) || (p.ws && ((unsigned long long)p.ws & 15ull)))
    return ur::fail(UR_E_INVALID, "ping-pong attention: needs Tq, Tk multiples of 256 and 16-byte aligned q / k / v^T rows");
  int n_split;
  pp_plan(p, ws_bytes, p.n_full, n_split);

  constexpr bool F16 = UR_TU_F16 != 0;
  dim3 grid(p.n_full + 2 * n_split), block(512);
  // UR_ATTN_PP_LDS (bytes, >= 32768): a larger request keeps further workgroups off the CU (occupancy experiments)
  static const int lds_env = getenv("UR_ATTN_PP_LDS") ? atoi(getenv("UR_ATTN_PP_LDS")) : 0;
  const int lds = lds_env > 4 * 16384 ? lds_env : 4 * 16384;
  static ur::DeviceOnce attr_once;
  if (lds > 65536)
  if (auto once_guard = attr_once.first())
    hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_pp64_kernel<F16>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipLaunchKernelGGL((attn_pp64_kernel<F16>), grid, block, lds, s, p);
  if (n_split) hipLaunchKernelGGL((attn_pp_combine_kernel<F16>), dim3(n_split * 8), dim3(256), 0, s, p, n_split);
  return ur::check_launch("ur_attention_fwd (ping-pong)");
}
