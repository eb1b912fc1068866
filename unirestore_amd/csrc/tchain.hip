// Token-stationary fused chains for gfx950 (hand-written HIP + hand-scheduled MFMA phases): a wave owns 32 tokens for a whole
// chain of token-wise layers.
//
// Why: at the UNet's 64x64 level (32 768 tokens x 320 channels at B = 8) every Linear of a BasicTransformerBlock is a short
// GEMM whose launch runs 2-5x above its streaming floor, and the tensors between them (84 MB of GEGLU hidden
// state per block) only exist to cross a launch boundary.  Here the token tile never leaves the register file:
//   * MFMA view: D[32 out-channels][32 tokens] += A[32 x 16] (weights, from LDS) . B[16 x 32] (activations, REGISTERS).
//     Lane (j = lane & 31, h = lane >> 5) owns token j; the accumulator of a stage - packed to 16 bit - IS the B operand of
//     the next stage because every weight matrix is stored with its output rows permuted by swap23 inside each 32-row block
//     (MFMA row i holds channel swap23(i)), so accumulator registers 8u..8u+7 of fragment f hold the 8 CONSECUTIVE channels
//     32f + 16u + 8h .. +7 = exactly the lane's share of k-step 2f + u.  No LDS round trip, no shuffles between layers.
//   * one workgroup = 4 waves (one per SIMD, the whole 512-entry register file each) = 128 consecutive tokens; the only
//     LDS traffic is the WEIGHT stream: the host packs every layer into 44-KiB tiles that are byte-for-byte the LDS image
//     ([row][128 B] blocks, XOR swizzle, fp32 vectors in the tail), so the loader is a linear LDS-DMA copy with scalar
//     addressing (buffer_load_dwordx4 ... lds, 11 KiB per wave per tile) into a 3-slot ring: one counted vmcnt + one raw
//     s_barrier per tile.  The DMA of tile t + 2 is issued from inside the MFMA phase of tile t (tchain_asm.inc).
//   * the MFMA phases are asm blocks with a software-pipelined fragment stream (tools/gen_chain_asm.py): at ~400 live registers
//     hipcc serialises ds_read -> wait -> mfma; the operands of the blocks stay compiler-allocated.
//   * LayerNorm is folded into the consuming weights (w' = W.gamma, bias' = W.beta + b) and applied to the accumulators as
//     rstd * (acc - mean * colsum) + bias' with mean / rstd computed in registers from the lane's own 16-bit values.
// Kinds (diffusers BasicTransformerBlock / Transformer2DModel as reached through
// /root/reference/src/modules/diffuie/base_model.py:137-160,184-198):
//   MLP   y = x + FF(LN3(x))                                                           (ur_ff_geglu_fused)
//   HEAD  h0 = proj_in(GroupNorm(x)); q, k, v^T = to_q/k/v(LN1(h0))                    (ur_transformer_head_fused)
//   TAIL  h1 = h0 + to_out(o1); h2 = h1 + to_out2(softmax(to_q2(LN2 h1) Kc^T) Vc); h3 = h2 + FF(LN3 h2);
//         y = x + proj_out(h3), + GroupNorm partial sums of y                         (ur_transformer_tail_fused)
#include "common.h"
#ifndef UR_CHAIN_ABL
#define UR_CHAIN_ABL 0      // timing-only ablations for A/B builds (tools/bench_chain.py): 1 = no MFMA phases, 2 = no weight DMA, 3 = no GELU,
#endif                      // 4 = MFMA phases without their fragment reads, 5 = without their MFMAs, 6 = cycle stamps of one FF chunk
#if UR_CHAIN_ABL == 4
#include "../../tools/ab/tchain_asm_abl4.inc"
#elif UR_CHAIN_ABL == 5
#include "../../tools/ab/tchain_asm_abl5.inc"
#else
#include "tchain_asm.inc"
#endif

namespace {

constexpr int TC_WB = 40960;                  // weight bytes of a tile: 320 rows x 128 B, or 5 x (64 rows x 128 B)
constexpr int TC_AUX = 4096;                  // fp32 vectors that ride with the tile (bias, LayerNorm column sums)
constexpr int TC_TILE = TC_WB + TC_AUX;       // 44 pieces of 1 KiB
constexpr int TC_SHARE = TC_TILE / 4;         // bytes each wave copies per tile (11 pieces)
constexpr int TC_NS = 3;                      // ring slots
constexpr int TC_TOK = 128;                   // tokens per workgroup
constexpr int TC_LDS = TC_NS * TC_TILE;

typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

#define TC_WAIT(N) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory")

__device__ __forceinline__ float f4e(const float4& v, int e) { return e == 0 ? v.x : e == 1 ? v.y : e == 2 ? v.z : v.w; }

template <int NF> __device__ __forceinline__ void zero_acc(f32x16 (&acc)[NF]) {
#pragma unroll
  for (int f = 0; f < NF; ++f)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[f][e] = 0.f;
}

template <bool F16>
struct TChain {
  typedef typename Frag<F16>::type frag_t;
  unsigned lds0;                            // LDS byte address of the ring
  u32x4 rs;                                 // buffer descriptor of the weight stream
  int lane, wid, h, voff;
  int ti, ntiles, islot, cslot;             // next tile to issue, its ring slot; ring slot of the next tile to consume
  int aoff[4];                              // lane's LDS offsets of the A fragment for the 4 k-steps of a [rows][128 B] block
#if UR_CHAIN_ABL == 6
  unsigned long long* dbg = nullptr;        // ablation 6: cycle stamps of one FF chunk (tools/bench_chain.py)
#endif

#define TC_MFMA_BLOCK(ASM, ...)                                                              \
  do {                                                                                       \
    if constexpr (UR_CHAIN_ABL == 1) break;                                                  \
    if constexpr (F16) asm volatile(ASM("v_mfma_f32_32x32x16_f16") __VA_ARGS__);              \
    else asm volatile(ASM("v_mfma_f32_32x32x16_bf16") __VA_ARGS__);                           \
  } while (0)

  __device__ __forceinline__ void init(unsigned char* sm, const unsigned char* stream, int nt) {
    lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lptr_t)sm);
    lane = threadIdx.x & 63;
    wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    h = lane >> 5;
    voff = lane * 16;
    ti = islot = cslot = 0;
    ntiles = nt;
    const unsigned long long a = (unsigned long long)stream;
    rs[0] = __builtin_amdgcn_readfirstlane((unsigned)a);
    rs[1] = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffffu);     // stride 0
    rs[2] = 0x7fffffffu;                                                        // num_records (bytes)
    rs[3] = 0x00020000u;
#pragma unroll
    for (int s = 0; s < 4; ++s) aoff[s] = (lane & 31) * 128 + ((((2 * s + h) ^ ((lane >> 1) & 7))) << 4);
    // tiles 0 and 1 start streaming before anything else happens
    dma_alone();
    dma_alone();
  }
  // (soffset, LDS base) of this wave's share of the next tile to issue.  Past the end of the stream the last tile is fetched
  // again (into a free slot): every phase issues exactly 11 pieces, so that one vmcnt immediate is right everywhere.
  __device__ __forceinline__ void dma_args(unsigned& so, unsigned& ld) {
    const int t = ti < ntiles ? ti : ntiles - 1;
    so = (unsigned)t * TC_TILE + (unsigned)wid * TC_SHARE;
    ld = lds0 + (unsigned)islot * TC_TILE + (unsigned)wid * TC_SHARE;
    ++ti;
    islot = islot == TC_NS - 1 ? 0 : islot + 1;
  }
  __device__ __forceinline__ void dma_alone() {
    unsigned so, ld;
    dma_args(so, ld);
    if constexpr (UR_CHAIN_ABL == 2) return;
    asm volatile(TC_ASM_DMA : "+s"(so), "+s"(ld) : "v"(voff), "s"(rs) : "memory", "scc");      // (s_add_u32 inside: SCC is clobbered)
  }
  // Tile `tc` is ready in its slot for every wave, and the slot of the tile before it is free again (the phase that follows
  // refills it with tile tc + 2).  In flight at this point: tile tc and tile tc + 1, 11 pieces each per wave, in issue order -
  // "at most 11 (+ EXTRA younger stores) outstanding" therefore means tile tc has landed.
  // lgkmcnt(0): this wave's fragment reads of the previous tile are retired before anyone overwrites its slot.
  template <int EXTRA = 0>
  __device__ __forceinline__ unsigned acquire() {
    if constexpr (UR_CHAIN_ABL == 2) TC_WAIT(0); else TC_WAIT(11 + EXTRA);
    __builtin_amdgcn_s_barrier();
    const unsigned slot = lds0 + (unsigned)cslot * TC_TILE;
    cslot = cslot == TC_NS - 1 ? 0 : cslot + 1;
    return slot;
  }
  // every DMA has landed and every wave is done with the ring: its memory may be reused
  __device__ __forceinline__ void drain() {
    TC_WAIT(0);
    __builtin_amdgcn_s_barrier();
  }

  // ---- MFMA phases (one weight tile each; `slot` = LDS byte address of the tile) ---------------------------------------
  // 64-deep k tile of an N = 320 stage: weights [320 rows][128 B], B fragments b0..b3; Z: first tile (accumulators start at 0)
  template <bool Z>
  __device__ __forceinline__ void gemm_tile(f32x16 (&acc)[10], const frag_t& b0, const frag_t& b1, const frag_t& b2, const frag_t& b3, unsigned slot) {
    const unsigned a0 = slot + aoff[0], a1 = slot + aoff[1], a2 = slot + aoff[2], a3 = slot + aoff[3];
    frag_t t0, t1, t2, t3, t4, t5, t6;
    unsigned so, ld;
    dma_args(so, ld);
#define TC_GEMM_OPERANDS                                                                                                        \
    : "+a"(acc[0]), "+a"(acc[1]), "+a"(acc[2]), "+a"(acc[3]), "+a"(acc[4]), "+a"(acc[5]), "+a"(acc[6]), "+a"(acc[7]), "+a"(acc[8]),  \
      "+a"(acc[9]), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "+s"(so), "+s"(ld)       \
    : "v"(b0), "v"(b1), "v"(b2), "v"(b3), "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(voff), "s"(rs), "s"(wid)                                  \
    : "memory", "scc"
    if constexpr (Z) TC_MFMA_BLOCK(TC_ASM_GEMM_N10_Z, TC_GEMM_OPERANDS);
    else TC_MFMA_BLOCK(TC_ASM_GEMM_N10, TC_GEMM_OPERANDS);
#undef TC_GEMM_OPERANDS
  }
  // a whole N = 320, K = 320 stage: 5 tiles, B fragments xb[0..19]; returns the LDS address of the last tile (its aux area
  // holds the stage's epilogue vectors and stays valid until the next acquire)
  // EXTRA: stores this wave issued just before the stage (they sit behind the DMA of the stage's second tile in the VMEM queue)
  template <int EXTRA = 0>
  __device__ __forceinline__ unsigned gemm_stage(f32x16 (&acc)[10], const frag_t (&xb)[20]) {
    unsigned slot = acquire<EXTRA>();
    gemm_tile<true>(acc, xb[0], xb[1], xb[2], xb[3], slot);
#pragma unroll
    for (int kt = 1; kt < 5; ++kt) {
      slot = kt == 1 ? acquire<EXTRA>() : acquire();
      gemm_tile<false>(acc, xb[4 * kt], xb[4 * kt + 1], xb[4 * kt + 2], xb[4 * kt + 3], slot);
    }
    return slot;
  }
  // GEGLU up-projection of 32 hidden units over K = 320: 5 blocks of [32 a rows | 32 g rows][128 B], B fragments xb[0..19]
  __device__ __forceinline__ void ff1_tile(f32x16 (&ag)[2], const frag_t (&xb)[20], unsigned slot) {
    const unsigned a0 = slot + aoff[0], a1 = slot + aoff[1], a2 = slot + aoff[2], a3 = slot + aoff[3];
    frag_t t0, t1, t2, t3, t4, t5, t6;
    unsigned so, ld;
    dma_args(so, ld);
    TC_MFMA_BLOCK(TC_ASM_FF1_A,
                  : "+a"(ag[0]), "+a"(ag[1]), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6),
                    "+s"(so), "+s"(ld)
                  : "v"(xb[0]), "v"(xb[1]), "v"(xb[2]), "v"(xb[3]), "v"(xb[4]), "v"(xb[5]), "v"(xb[6]), "v"(xb[7]), "v"(xb[8]), "v"(xb[9]),
                    "v"(xb[10]), "v"(xb[11]), "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(voff), "s"(rs), "s"(wid)
                  : "memory", "scc");
    TC_MFMA_BLOCK(TC_ASM_FF1_B,
                  : "+a"(ag[0]), "+a"(ag[1]), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6)
                  : "v"(xb[12]), "v"(xb[13]), "v"(xb[14]), "v"(xb[15]), "v"(xb[16]), "v"(xb[17]), "v"(xb[18]), "v"(xb[19]),
                    "v"(a0), "v"(a1), "v"(a2), "v"(a3)
                  : "memory");
  }

  // ---- epilogue vectors (aux area of a tile; asm reads: hipcc must not see LDS reads behind the DMA, see gen_chain_asm.py) ---
  // LDS address of the lane's share of a tile's aux area (fragment offsets are immediate operands of the aux blocks)
  __device__ __forceinline__ unsigned aux_base(unsigned slot) const { return slot + TC_WB + 32 * h; }
  // the 8 float4 of an FF1 tile for this lane's half-fragment U: ba | bg | colsum a | colsum g (2 float4 each)
  template <int U>
  __device__ __forceinline__ void aux_ff1(unsigned slot, float4 (&q)[8]) const {
    asm volatile(TC_ASM_AUX_FF1
                 : "=&v"(q[0]), "=&v"(q[1]), "=&v"(q[2]), "=&v"(q[3]), "=&v"(q[4]), "=&v"(q[5]), "=&v"(q[6]), "=&v"(q[7])
                 : "v"(aux_base(slot)), "i"(64 * U)
                 : "memory");
  }
  // 4 half-fragments (2 accumulator fragments from F0) of the fp32 vector at float offset VEC of the aux area:
  // q[2*i], q[2*i+1] = the lane's 8 values for half-fragment 2*F0 + i
  template <int VEC, int F0>
  __device__ __forceinline__ void aux_vec4(unsigned slot, float4 (&q)[8]) const {
    asm volatile(TC_ASM_AUX_8
                 : "=&v"(q[0]), "=&v"(q[1]), "=&v"(q[2]), "=&v"(q[3]), "=&v"(q[4]), "=&v"(q[5]), "=&v"(q[6]), "=&v"(q[7])
                 : "v"(aux_base(slot)), "i"((VEC + 32 * F0) * 4)
                 : "memory");
  }

  // cross-attention scores of one head: S^T[96 keys][32 tokens] = K_h (3 fragments of 32 key rows x 64 d at the tile start) . q^T
  __device__ __forceinline__ void att_s(f32x16& s0, f32x16& s1, f32x16& s2, const frag_t& q0, const frag_t& q1, const frag_t& q2, const frag_t& q3, unsigned slot) {
    const unsigned a0 = slot + aoff[0], a1 = slot + aoff[1], a2 = slot + aoff[2], a3 = slot + aoff[3];
    frag_t t0, t1, t2, t3, t4, t5, t6;
    unsigned so, ld;
    dma_args(so, ld);
    TC_MFMA_BLOCK(TC_ASM_ATT_S,
                  : "+a"(s0), "+a"(s1), "+a"(s2), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "+s"(so), "+s"(ld)
                  : "v"(q0), "v"(q1), "v"(q2), "v"(q3), "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(voff), "s"(rs), "s"(wid)
                  : "memory", "scc");
  }
  // O^T[64 d][32 tokens] = V^T_h (two [64 d rows][128 B] blocks at bytes 12288 / 20480: keys 0-63 | 64-127) . P^T, first 80 keys
  __device__ __forceinline__ void att_pv(f32x16 (&o)[2], const frag_t (&pf)[5], unsigned slot) {
    const unsigned v0 = slot + 12288, v1 = slot + 20480;
    const unsigned a0 = v0 + aoff[0], a1 = v0 + aoff[1], a2 = v0 + aoff[2], a3 = v0 + aoff[3], a4 = v1 + aoff[0];
    frag_t t0, t1, t2, t3, t4, t5, t6;
    TC_MFMA_BLOCK(TC_ASM_ATT_PV,
                  : "+a"(o[0]), "+a"(o[1]), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6)
                  : "v"(pf[0]), "v"(pf[1]), "v"(pf[2]), "v"(pf[3]), "v"(pf[4]), "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4)
                  : "memory");
  }
  // bias (floats 32 F..) and column sums (floats 512 + 32 F..) of ONE fragment F: q[0..3] bias (u = 0, 1), q[4..7] column sums
  template <int F>
  __device__ __forceinline__ void aux_bc(unsigned slot, float4 (&q)[8]) const {
    asm volatile(TC_ASM_AUX_BC8
                 : "=&v"(q[0]), "=&v"(q[1]), "=&v"(q[2]), "=&v"(q[3]), "=&v"(q[4]), "=&v"(q[5]), "=&v"(q[6]), "=&v"(q[7])
                 : "v"(aux_base(slot)), "i"(128 * F)
                 : "memory");
  }
  // LayerNorm-folded epilogue of ONE accumulator fragment: out[0..1] = 16-bit(rstd * (acc - mean * colsum) + bias'), vectors of
  // fragment FA of the aux area of `slot`
  template <int FA>
  __device__ __forceinline__ void ln_pack(const f32x16& acc, unsigned slot, float mean, float rstd, frag_t& o0, frag_t& o1) const {
    float4 q[8];
    aux_bc<FA>(slot, q);
    const float mr = mean * rstd;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = fmaf(rstd, acc[8 * u + e], fmaf(-mr, f4e(q[4 + 2 * u + (e >> 2)], e & 3), f4e(q[2 * u + (e >> 2)], e & 3)));
      if (u == 0) o0 = pack(v); else o1 = pack(v);
    }
  }

  static __device__ __forceinline__ void unpack(const frag_t& f, float (&o)[8]) { unpack8t<F16>(__builtin_bit_cast(uint4, f), o); }
  static __device__ __forceinline__ frag_t pack(const float (&v)[8]) { return __builtin_bit_cast(frag_t, pack8t<F16>(v)); }

  // out[2f+u] = 16-bit(act(acc + bias) [+ residual res[2f+u]]) for the 10 fragments of a bias-only stage; `res` may alias `out`
  template <bool RES, bool GELU = false>
  __device__ __forceinline__ void bias_res_pack(const f32x16 (&acc)[10], unsigned slot, const frag_t (&res)[20], frag_t (&out)[20]) const {
    bias_res_pair<RES, GELU, 0>(acc, slot, res, out);
    bias_res_pair<RES, GELU, 1>(acc, slot, res, out);
    bias_res_pair<RES, GELU, 2>(acc, slot, res, out);
    bias_res_pair<RES, GELU, 3>(acc, slot, res, out);
    bias_res_pair<RES, GELU, 4>(acc, slot, res, out);
  }
  template <bool RES, bool GELU, int FP>                    // two fragments at a time: 32 registers of bias values live
  __device__ __forceinline__ void bias_res_pair(const f32x16 (&acc)[10], unsigned slot, const frag_t (&res)[20], frag_t (&out)[20]) const {
    float4 q[8];
    aux_vec4<0, 2 * FP>(slot, q);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int f = 2 * FP + (i >> 1), u = i & 1;
      float r[8], v[8];
      if constexpr (RES) unpack(res[2 * f + u], r);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        v[e] = acc[f][8 * u + e] + f4e(q[2 * i + (e >> 2)], e & 3);
        if constexpr (GELU) v[e] = gelu_f(v[e]);
        if constexpr (RES) v[e] += r[e];
      }
      out[2 * f + u] = pack(v);
    }
  }
};

// mean and rstd of the token's C values (this lane's KS fragments + the other half's), exact two-pass variance
template <bool F16, int KS>
__device__ __forceinline__ void row_stats(const typename Frag<F16>::type (&x)[KS], float eps, float& mean, float& rstd) {
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < KS; ++k) {
    float v[8];
    TChain<F16>::unpack(x[k], v);
    s += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
  }
  s += __shfl_xor(s, 32, 64);
  mean = s * (1.0f / (16 * KS));
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < KS; ++k) {
    float v[8];
    TChain<F16>::unpack(x[k], v);
#pragma unroll
    for (int e = 0; e < 8; ++e) { const float d = v[e] - mean; q = fmaf(d, d, q); }
  }
  q += __shfl_xor(q, 32, 64);
  rstd = rsqrtf(q * (1.0f / (16 * KS)) + eps);
}

// GEGLU epilogue of half-fragment U of an FF1 tile: 16-bit((rstd*(a - mean*ca) + ba) * gelu(rstd*(g - mean*cg) + bg)), 8 hidden units
template <bool F16, int U>
__device__ __forceinline__ typename Frag<F16>::type ff1_epilogue(const TChain<F16>& tc, const f32x16 (&ag)[2], unsigned slot, float rstd, float mr) {
  float4 q[8];                                 // ba | bg | colsum a | colsum g of the lane's 8 channels
  tc.template aux_ff1<U>(slot, q);
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float ba = f4e(q[e >> 2], e & 3), bg = f4e(q[2 + (e >> 2)], e & 3);
    const float ca = f4e(q[4 + (e >> 2)], e & 3), cg = f4e(q[6 + (e >> 2)], e & 3);
    const float a = fmaf(rstd, ag[0][8 * U + e], fmaf(-mr, ca, ba));
    const float g = fmaf(rstd, ag[1][8 * U + e], fmaf(-mr, cg, bg));
    v[e] = UR_CHAIN_ABL == 3 ? a * g : a * gelu_f(g);
  }
  return TChain<F16>::pack(v);
}

// FeedForward(GEGLU) over the LayerNorm of xb (raw fragments; LayerNorm folded: mean / rstd given): acc += W2 . GEGLU(...) (no b2);
// the caller zeroes acc (ONE asm instance of the FF2 phase in the loop: with a second, zero-initialising one for the first chunk
// hipcc's register allocation of the whole kernel degraded to ~100 spilled registers).
// Stream per 64 hidden units: [FF1 tile: 5 blocks of (32 a rows | 32 g rows) x 64 k; aux = ba | bg | colsum_a | colsum_g]
//                             [FF1 tile of the next 32 units]  [FF2 tile: 320 rows x 64 k; aux of the LAST one = b2]
// Returns the LDS address of the last FF2 tile.
template <bool F16>
__device__ __forceinline__ unsigned ff_stage(TChain<F16>& tc, f32x16 (&acc)[10], f32x16 (&ag)[2], const typename Frag<F16>::type (&xb)[20],
                                             float mean, float rstd, int nchunk) {
  typedef typename Frag<F16>::type frag_t;
  const float mr = mean * rstd;
  unsigned last = 0;
#if UR_CHAIN_ABL == 6
  unsigned long long ts[12];
#define TC_TS(i) do { if (c == 10) ts[i] = __builtin_readcyclecounter(); } while (0)
#else
#define TC_TS(i) do {} while (0)
#endif
  for (int c = 0; c < nchunk; ++c) {
    frag_t hid[4];
    TC_TS(0);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const unsigned slot = tc.acquire();
      TC_TS(1 + 3 * half);
      tc.ff1_tile(ag, xb, slot);               // (zeroes ag: its first MFMAs take C = 0)
      TC_TS(2 + 3 * half);
      hid[half * 2] = ff1_epilogue<F16, 0>(tc, ag, slot, rstd, mr);
      __builtin_amdgcn_sched_barrier(0);       // (one half-fragment's temporaries at a time: interleaved, the two spill)
      hid[half * 2 + 1] = ff1_epilogue<F16, 1>(tc, ag, slot, rstd, mr);
      __builtin_amdgcn_sched_barrier(0);
      TC_TS(3 + 3 * half);
    }
    last = tc.acquire();
    TC_TS(7);
    tc.template gemm_tile<false>(acc, hid[0], hid[1], hid[2], hid[3], last);     // (acc starts at zero: see the callers)
    TC_TS(8);
  }
#if UR_CHAIN_ABL == 6
  if (tc.dbg && blockIdx.x == 17 && tc.lane == 0)
    for (int i = 0; i < 9; ++i) tc.dbg[tc.wid * 16 + i] = ts[i];
#endif
  return last;
}

template <bool F16>
__device__ __forceinline__ void load_frags(const uint16_t* __restrict__ base, long long tok, int ld, int h, typename Frag<F16>::type (&x)[20]) {
  const uint16_t* xp = base + tok * ld + 8 * h;
#pragma unroll
  for (int s = 0; s < 20; ++s) x[s] = *reinterpret_cast<const typename Frag<F16>::type*>(xp + 16 * s);
}
// (plain stores: the consumers re-read these rows from the same L2 at once - written through they cost 1.9 ms per forward, common.h)
#define TC_STORE16(ptr, val) (*reinterpret_cast<uint4*>(ptr) = __builtin_bit_cast(uint4, val))
template <bool F16>
__device__ __forceinline__ void store_frags(uint16_t* __restrict__ base, long long tok, int ld, int h, const typename Frag<F16>::type (&x)[20]) {
  uint16_t* xp = base + tok * ld + 8 * h;
#pragma unroll
  for (int s = 0; s < 20; ++s) TC_STORE16(xp + 16 * s, x[s]);
}

// ---------------------------------------------------------------------------------------------------------------------
struct MlpP {
  const unsigned char* stream;   // packed tiles (ntiles x TC_TILE bytes)
  const uint16_t* x;             // [T][ldx]
  uint16_t* y;                   // [T][ldy]
  int T, ldx, ldy, ntiles, hidden;
  float eps;
};

// kind MLP: y = x + W2 . GEGLU(W1 . LayerNorm(x) + b1) + b2, C = 320
template <bool F16>
__global__ __launch_bounds__(256, 1) void tchain_mlp_kernel(const MlpP p) {
  typedef typename Frag<F16>::type frag_t;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  TChain<F16> tc;
  tc.init(smem, p.stream, p.ntiles);
#if UR_CHAIN_ABL == 6
  tc.dbg = reinterpret_cast<unsigned long long*>(p.y + (long long)(p.T - 128) * p.ldy);      // (timing build: stamps land in the last rows of y)
#endif
  const long long tok = (long long)blockIdx.x * TC_TOK + tc.wid * 32 + (tc.lane & 31);
  frag_t xb[20];
  load_frags<F16>(p.x, tok, p.ldx, tc.h, xb);
  float mean, rstd;
  row_stats<F16, 20>(xb, p.eps, mean, rstd);
  f32x16 acc[10], ag[2];                         // declared once: one register tuple each for their whole lifetime
  zero_acc(acc);
  zero_acc(ag);
  const unsigned last = ff_stage<F16>(tc, acc, ag, xb, mean, rstd, p.hidden / 64);
  // y = ff + b2 + x: b2 rides in the aux area of the last FF2 tile; the residual is the raw input, still in registers
#pragma unroll
  for (int s = 0; s < 20; ++s) asm volatile("" : "+v"(xb[s]));      // opaque: hipcc would otherwise keep the 160 fp32 values row_stats unpacked alive (in scratch) across the loop
  tc.template bias_res_pack<true>(acc, last, xb, xb);
#if UR_CHAIN_ABL == 6
  if (blockIdx.x != gridDim.x - 1)               // (timing build: the stamps live in the last workgroup's rows)
#endif
  store_frags<F16>(p.y, tok, p.ldy, tc.h, xb);
  TC_WAIT(0);                                    // no LDS-DMA may outlive the workgroup
}

template <bool F16>
int launch_mlp(const MlpP& p, hipStream_t s) {
  static ur::DeviceOnce attr_once;      // the attribute is per device
  if (auto once_guard = attr_once.first()) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&tchain_mlp_kernel<F16>), hipFuncAttributeMaxDynamicSharedMemorySize, TC_LDS);
  }
  hipLaunchKernelGGL((tchain_mlp_kernel<F16>), dim3(p.T / TC_TOK), dim3(256), TC_LDS, s, p);
  return ur::check_launch("ur_ff_geglu_fused");
}


// the lane's token index, recomputed from the hardware ids behind an opaque barrier (keeping the value computed at kernel entry
// alive through the whole chain costs registers that end up in scratch)
__device__ __forceinline__ long long token_again() {
  unsigned t = threadIdx.x;
  asm volatile("" : "+v"(t));
  return (long long)blockIdx.x * TC_TOK + (t >> 6) * 32 + (t & 31);
}

template <bool F16>
__device__ __forceinline__ void launder(typename Frag<F16>::type (&x)[20]) {
  // opaque to the optimiser: hipcc otherwise keeps the 160 fp32 values a LayerNorm-statistics pass unpacked alive (spilled) until
  // the fragments' next unpack, many phases later
#pragma unroll
  for (int s = 0; s < 20; ++s) asm volatile("" : "+v"(x[s]));
}

// HEAD: LayerNorm-folded epilogue of fragment F of a q / k / v stage
template <bool F16, int F>
__device__ __forceinline__ void head_out(const TChain<F16>& tc, const f32x16 (&acc)[10], unsigned slot, float mean, float rstd, int part,
                                         uint16_t* op, typename Frag<F16>::type (&xb)[20]) {
  typedef typename Frag<F16>::type frag_t;
  frag_t o0, o1;
  tc.template ln_pack<F>(acc[F], slot, mean, rstd, o0, o1);
  if (part < 2) {                                                // q, k: straight out (two 16-byte stores per fragment)
    TC_STORE16(op + 32 * F, o0);
    TC_STORE16(op + 32 * F + 16, o1);
  } else {                                                       // v: h0's fragments are dead now - keep v in their registers
    xb[2 * F] = o0;
    xb[2 * F + 1] = o1;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// kind HEAD: h0 = proj_in(a_n * x + b_n) (GroupNorm applied from its per-image affine), q | k | v = to_q/k/v(LayerNorm1(h0)).
// Stream: proj_in (5 tiles, aux of the last = bias), to_q, to_k, to_v (5 tiles each, LayerNorm folded; aux of the last =
// bias' at floats 0.. and column sums at floats 512..).  v leaves TRANSPOSED ([image][channel][token]) for the attention kernel.
struct HeadP {
  const unsigned char* stream;
  const uint16_t* x;             // [T][C] transformer input (raw resnet output)
  const float* ab;               // [N][2][C] GroupNorm affine of x per image (ur_groupnorm_finalize)
  uint16_t* h0;                  // [T][C]
  uint16_t* q;                   // [T][C]
  uint16_t* k;                   // [T][C]
  uint16_t* vt;                  // [N][C][tok_per_img]
  int T, tok_per_img, ntiles;
  float eps;
};

template <bool F16>
__global__ __launch_bounds__(256, 1) void tchain_head_kernel(const HeadP p) {
  typedef typename Frag<F16>::type frag_t;
  constexpr int C = 320;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  TChain<F16> tc;
  tc.init(smem, p.stream, p.ntiles);
  const long long tok0 = (long long)blockIdx.x * TC_TOK;
  const long long tok = tok0 + tc.wid * 32 + (tc.lane & 31);
  const int img = (int)(tok0 / p.tok_per_img);
  frag_t xb[20];
  load_frags<F16>(p.x, tok, C, tc.h, xb);
  {
    const float* ap = p.ab + (long long)img * 2 * C + 8 * tc.h;
#pragma unroll
    for (int s = 0; s < 20; ++s) {
      const float4 a0 = *reinterpret_cast<const float4*>(ap + 16 * s), a1 = *reinterpret_cast<const float4*>(ap + 16 * s + 4);
      const float4 b0 = *reinterpret_cast<const float4*>(ap + C + 16 * s), b1 = *reinterpret_cast<const float4*>(ap + C + 16 * s + 4);
      float v[8];
      tc.unpack(xb[s], v);
      v[0] = fmaf(v[0], a0.x, b0.x); v[1] = fmaf(v[1], a0.y, b0.y); v[2] = fmaf(v[2], a0.z, b0.z); v[3] = fmaf(v[3], a0.w, b0.w);
      v[4] = fmaf(v[4], a1.x, b1.x); v[5] = fmaf(v[5], a1.y, b1.y); v[6] = fmaf(v[6], a1.z, b1.z); v[7] = fmaf(v[7], a1.w, b1.w);
      xb[s] = tc.pack(v);
      if ((s & 3) == 3) __builtin_amdgcn_sched_barrier(0);     // (hipcc otherwise hoists all 80 affine loads = 320 registers and spills them)
    }
  }
  f32x16 acc[10];
  zero_acc(acc);
  unsigned slot = tc.gemm_stage(acc, xb);
  tc.template bias_res_pack<false>(acc, slot, xb, xb);               // h0 (no residual)
  store_frags<F16>(p.h0, token_again(), C, tc.h, xb);                // 20 stores
  float mean, rstd;
  row_stats<F16, 20>(xb, p.eps, mean, rstd);
  launder<F16>(xb);
#pragma unroll
  for (int part = 0; part < 3; ++part) {
    slot = tc.template gemm_stage<20>(acc, xb);
    uint16_t* op = (part == 0 ? p.q : p.k) + token_again() * C + 8 * tc.h;
    head_out<F16, 0>(tc, acc, slot, mean, rstd, part, op, xb);
    head_out<F16, 1>(tc, acc, slot, mean, rstd, part, op, xb);
    head_out<F16, 2>(tc, acc, slot, mean, rstd, part, op, xb);
    head_out<F16, 3>(tc, acc, slot, mean, rstd, part, op, xb);
    head_out<F16, 4>(tc, acc, slot, mean, rstd, part, op, xb);
    head_out<F16, 5>(tc, acc, slot, mean, rstd, part, op, xb);
    head_out<F16, 6>(tc, acc, slot, mean, rstd, part, op, xb);
    head_out<F16, 7>(tc, acc, slot, mean, rstd, part, op, xb);
    head_out<F16, 8>(tc, acc, slot, mean, rstd, part, op, xb);
    head_out<F16, 9>(tc, acc, slot, mean, rstd, part, op, xb);
  }
  // v^T: transpose this wave's [32 tokens][320 channels] through LDS (the ring is free: the stream has ended)
  tc.drain();
  unsigned char* tr = smem + tc.wid * (C * 64);
  const int j = tc.lane & 31;
#pragma unroll
  for (int s = 0; s < 20; ++s) {
    const uint4 w = __builtin_bit_cast(uint4, xb[s]);
    const unsigned wv[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = 16 * s + 8 * tc.h + e;
      *reinterpret_cast<uint16_t*>(tr + c * 64 + j * 2) = (uint16_t)(wv[e >> 1] >> (16 * (e & 1)));
    }
  }
  __syncthreads();
  {
    const long long tin = tok0 - (long long)img * p.tok_per_img + tc.wid * 32;      // first token of this wave inside its image
    uint16_t* vp = p.vt + (long long)img * C * p.tok_per_img + tin + (tc.lane & 3) * 8;
#pragma unroll
    for (int i = 0; i < 20; ++i) {
      const int r = i * 16 + (tc.lane >> 2);
      *reinterpret_cast<uint4*>(vp + (long long)r * p.tok_per_img) = *reinterpret_cast<const uint4*>(tr + r * 64 + (tc.lane & 3) * 16);
    }
  }
}

// GroupNorm statistics of a workgroup's [128 tokens][320] tile of 16-bit values (fragments y of every wave): tile -> LDS
// [128][656 B] (row pad: conflict-free column reads), 160 threads add one channel pair each over the 128 rows in a fixed order ->
// this tile's slot of the partial plane [N][tokens_per_image / 128][320][2].  The ring must be drained (its memory is reused).
template <bool F16>
__device__ __forceinline__ void gn_partials_of_tile(const TChain<F16>& tc, unsigned char* smem, const typename Frag<F16>::type (&y)[20],
                                                    float* gn_part, long long tok0, int tok_per_img) {
  typedef typename Frag<F16>::type frag_t;
  constexpr int C = 320, ROW = 2 * C + 16;
  unsigned char* yt = smem + (tc.wid * 32 + (tc.lane & 31)) * ROW + 16 * tc.h;
#pragma unroll
  for (int s = 0; s < 20; ++s) *reinterpret_cast<frag_t*>(yt + 32 * s) = y[s];
  __syncthreads();
  if (threadIdx.x < C / 2) {
    float s0 = 0.f, s1 = 0.f, q0 = 0.f, q1 = 0.f;
    for (int r = 0; r < TC_TOK; ++r) {
      const uint32_t w = *reinterpret_cast<const uint32_t*>(smem + r * ROW + threadIdx.x * 4);
      const float a = Act<F16>::lo(w), b = Act<F16>::hi(w);
      s0 += a; q0 = fmaf(a, a, q0); s1 += b; q1 = fmaf(b, b, q1);
    }
    const int img = (int)(tok0 / tok_per_img), parts = tok_per_img / TC_TOK;
    const int part = (int)((tok0 - (long long)img * tok_per_img) / TC_TOK);
    float* st = gn_part + (((long long)img * parts + part) * C + 2 * threadIdx.x) * 2;
    *reinterpret_cast<float4*>(st) = make_float4(s0, q0, s1, q1);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// kind CSCE (SC-Tuner, /root/reference/src/modules/diffuie/scedit.py:24-38): s = x + proj(cond); out = tuner.2(GELU(tuner.0(s))) + s,
// x = a UNet skip [T][320], cond = the Controller feature of the same resolution [T][256]; + GroupNorm partial sums of out.
// Stream: proj (4 tiles of 320 rows x 64 k; aux of the last = bias) | tuner.0 (5, bias) | tuner.2 (5, bias).
struct CsceP {
  const unsigned char* stream;
  const uint16_t* x;             // [T][320]
  const uint16_t* cond;          // [T][256]
  uint16_t* y;                   // [T][320]
  float* gn_part;                // [N][tok_per_img / 128][320][2] or null
  int T, tok_per_img, ntiles;
};

template <bool F16>
__global__ __launch_bounds__(256, 1) void tchain_csce_kernel(const CsceP p) {
  typedef typename Frag<F16>::type frag_t;
  constexpr int C = 320, CC = 256;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  TChain<F16> tc;
  tc.init(smem, p.stream, p.ntiles);
  const long long tok0 = (long long)blockIdx.x * TC_TOK;
  const long long tok = tok0 + tc.wid * 32 + (tc.lane & 31);
  frag_t sb[20], hb[20];
  {                                                     // cond fragments: 16 k-steps, parked in hb until proj has consumed them
    const uint16_t* cp = p.cond + tok * CC + 8 * tc.h;
#pragma unroll
    for (int s = 0; s < 16; ++s) hb[s] = *reinterpret_cast<const frag_t*>(cp + 16 * s);
  }
  f32x16 acc[10];
  zero_acc(acc);
  // s = x + proj(cond) + b: 4 k tiles; the skip x is fetched behind the third (80 registers less during the others)
  unsigned slot = tc.acquire();
  tc.template gemm_tile<true>(acc, hb[0], hb[1], hb[2], hb[3], slot);
#pragma unroll
  for (int kt = 1; kt < 4; ++kt) {
    slot = tc.acquire();
    tc.template gemm_tile<false>(acc, hb[4 * kt], hb[4 * kt + 1], hb[4 * kt + 2], hb[4 * kt + 3], slot);
    if (kt == 2) load_frags<F16>(p.x, token_again(), C, tc.h, sb);
  }
  tc.template bias_res_pack<true>(acc, slot, sb, sb);
  // h = GELU(tuner.0(s) + b)
  slot = tc.gemm_stage(acc, sb);
  tc.template bias_res_pack<false, true>(acc, slot, hb, hb);
  // out = tuner.2(h) + b + s
  slot = tc.gemm_stage(acc, hb);
  tc.template bias_res_pack<true>(acc, slot, sb, sb);
  store_frags<F16>(p.y, token_again(), C, tc.h, sb);
  tc.drain();
  if (p.gn_part) gn_partials_of_tile<F16>(tc, smem, sb, p.gn_part, tok0, p.tok_per_img);
}

// ---------------------------------------------------------------------------------------------------------------------
// kind TAIL: everything of a BasicTransformerBlock behind the self-attention + Transformer2DModel.proj_out:
//   h1 = h0 + to_out1(o1) + b;  q2 = to_q2(LayerNorm2(h1));  o2 = softmax(q2 Kc^T / 8) Vc per head (constant context, <= 80 keys);
//   h2 = h1 + to_out2(o2) + b;  h3 = h2 + FF(LayerNorm3(h2));  y = x + proj_out(h3) + b;  GroupNorm partial sums of y.
// Stream: to_out1 (5 tiles, aux = bias) | per head: [to_q2 rows of the head, LayerNorm folded: 5 blocks of (32 | 32 rows) x 64 k,
// aux = bias' at floats 0..63 and column sums at 512..575] [K_h 96 x 64 at byte 0, V^T_h as two [64][128 B] blocks at bytes
// 12288 / 20480] [to_out2 K slice of the head: 320 rows x 64 k; aux of the last = bias] | FF (3 per 64 hidden units) | proj_out (5, bias).
struct TailP {
  const unsigned char* stream;
  const uint16_t* o1;            // [T][C] self-attention output
  const uint16_t* h0;            // [T][C] residual stream entering the block
  const uint16_t* xres;          // [T][C] transformer input (residual of proj_out)
  uint16_t* y;                   // [T][C]
  float* gn_part;                // [N][tok_per_img / 128][C][2] partial (sum, sum of squares) of y, or null
  int T, tok_per_img, ntiles, hidden, tk;
  float eps, scale_log2e;
};

template <bool F16>
__global__ __launch_bounds__(256, 1) void tchain_tail_kernel(const TailP p) {
  typedef typename Frag<F16>::type frag_t;
  constexpr int C = 320;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  TChain<F16> tc;
  tc.init(smem, p.stream, p.ntiles);
  const long long tok0 = (long long)blockIdx.x * TC_TOK;
  const long long tok = tok0 + tc.wid * 32 + (tc.lane & 31);
  frag_t hb[20], xq[20];
  load_frags<F16>(p.o1, tok, C, tc.h, xq);
  f32x16 acc[10], ag[2], sc2[1];                      // scores of a head: keys 0-63 in ag (q is packed by then), keys 64-95 in sc2
  zero_acc(acc);
  zero_acc(ag);
  zero_acc(sc2);
  // h1 = h0 + to_out1(o1) + b; the residual h0 is fetched behind the 4th of the stage's 5 tiles (80 registers less during the others)
  unsigned slot = tc.acquire();
  tc.template gemm_tile<true>(acc, xq[0], xq[1], xq[2], xq[3], slot);
#pragma unroll
  for (int kt = 1; kt < 5; ++kt) {
    slot = tc.acquire();
    tc.template gemm_tile<false>(acc, xq[4 * kt], xq[4 * kt + 1], xq[4 * kt + 2], xq[4 * kt + 3], slot);
    if (kt == 3) load_frags<F16>(p.h0, token_again(), C, tc.h, hb);
  }
  tc.template bias_res_pack<true>(acc, slot, hb, hb);
  // cross-attention, head by head: q_h = to_q2(LN2(h1))[64 channels of head h] (an FF1-shaped tile: 2 fragments over K = 320,
  // LayerNorm folded) -> scores over the constant keys -> softmax -> P.V -> o2_h (4 fragments) is at once the K slice h of
  // to_out2, accumulated into `acc`: no [token][320] intermediate of the attention exists, not even in registers.
  float mean, rstd;
  row_stats<F16, 20>(hb, p.eps, mean, rstd);
  launder<F16>(hb);
#pragma unroll
  for (int hd = 0; hd < 5; ++hd) {
    slot = tc.acquire();
    tc.ff1_tile(ag, hb, slot);
    frag_t qf[4];
    tc.template ln_pack<0>(ag[0], slot, mean, rstd, qf[0], qf[1]);
    tc.template ln_pack<1>(ag[1], slot, mean, rstd, qf[2], qf[3]);
    slot = tc.acquire();
    tc.att_s(ag[0], ag[1], sc2[0], qf[0], qf[1], qf[2], qf[3], slot);
    // softmax over the keys: the lane holds 48 of its token's 96 scores (key = 32 kf + 16 (r >> 3) + 8 h + (r & 7)), the other
    // half's lane the rest.  Two passes over the accumulator registers (maximum, then exp2 + sum + pack): no 48-float copy.
    int lim = p.tk - 8 * tc.h;                        // key index (without the lane half's + 8h) below which a key is real
    asm volatile("" : "+v"(lim));                     // opaque per head: hipcc otherwise hoists all 88 key comparisons of all heads to kernel entry (as 88 live masks)
    float mx = -INFINITY;
#pragma unroll
    for (int kf = 0; kf < 3; ++kf)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = 32 * kf + 16 * (r >> 3) + (r & 7);
        mx = fmaxf(mx, key < lim ? (kf < 2 ? ag[kf & 1][r] : sc2[0][r]) : -INFINITY);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float mc = mx * p.scale_log2e;
    float l = 0.f;
    frag_t pf[5];
#pragma unroll
    for (int ks = 0; ks < 5; ++ks) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int key = 16 * ks + e;
        const float sv = ks < 4 ? ag[(ks >> 1) & 1][8 * (ks & 1) + e] : sc2[0][e];
        v[e] = key < lim ? __builtin_amdgcn_exp2f(fmaf(sv, p.scale_log2e, -mc)) : 0.f;
        l += v[e];
      }
      pf[ks] = tc.pack(v);
    }
    l += __shfl_xor(l, 32, 64);
    tc.att_pv(ag, pf, slot);
    const float inv = __builtin_amdgcn_rcpf(l);
    frag_t of[4];
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = ag[f][8 * u + e] * inv;
        of[2 * f + u] = tc.pack(v);
      }
    slot = tc.acquire();
    if (hd == 0) tc.template gemm_tile<true>(acc, of[0], of[1], of[2], of[3], slot);
    else tc.template gemm_tile<false>(acc, of[0], of[1], of[2], of[3], slot);
  }
  // h2 = h1 + to_out2(o2) + b (bias: aux of the last to_out2 tile)
  tc.template bias_res_pack<true>(acc, slot, hb, hb);
  // h3 = h2 + FF(LN3(h2))
  row_stats<F16, 20>(hb, p.eps, mean, rstd);
  launder<F16>(hb);
  zero_acc(acc);
  slot = ff_stage<F16>(tc, acc, ag, hb, mean, rstd, p.hidden / 64);
  tc.template bias_res_pack<true>(acc, slot, hb, hb);
  // y = x + proj_out(h3) + b
  slot = tc.gemm_stage(acc, hb);
  const long long tok2 = token_again();
  load_frags<F16>(p.xres, tok2, C, tc.h, xq);        // (after the stage: 80 more live registers during it would spill)
  tc.template bias_res_pack<true>(acc, slot, xq, xq);
  store_frags<F16>(p.y, tok2, C, tc.h, xq);
  tc.drain();                                        // no LDS-DMA may outlive the workgroup; the ring memory is free now
  if (p.gn_part) gn_partials_of_tile<F16>(tc, smem, xq, p.gn_part, tok0, p.tok_per_img);
}

template <bool F16>
int launch_head(const HeadP& p, hipStream_t s) {
  static ur::DeviceOnce attr_once;      // the attribute is per device
  if (auto once_guard = attr_once.first()) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&tchain_head_kernel<F16>), hipFuncAttributeMaxDynamicSharedMemorySize, TC_LDS);
  }
  hipLaunchKernelGGL((tchain_head_kernel<F16>), dim3(p.T / TC_TOK), dim3(256), TC_LDS, s, p);
  return ur::check_launch("ur_transformer_head_fused");
}
template <bool F16>
int launch_csce(const CsceP& p, hipStream_t s) {
  static ur::DeviceOnce attr_once;      // the attribute is per device
  if (auto once_guard = attr_once.first()) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&tchain_csce_kernel<F16>), hipFuncAttributeMaxDynamicSharedMemorySize, TC_LDS);
  }
  hipLaunchKernelGGL((tchain_csce_kernel<F16>), dim3(p.T / TC_TOK), dim3(256), TC_LDS, s, p);
  return ur::check_launch("ur_csce_fused");
}
template <bool F16>
int launch_tail(const TailP& p, hipStream_t s) {
  static ur::DeviceOnce attr_once;      // the attribute is per device
  if (auto once_guard = attr_once.first()) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&tchain_tail_kernel<F16>), hipFuncAttributeMaxDynamicSharedMemorySize, TC_LDS);
  }
  hipLaunchKernelGGL((tchain_tail_kernel<F16>), dim3(p.T / TC_TOK), dim3(256), TC_LDS, s, p);
  return ur::check_launch("ur_transformer_tail_fused");
}

}  // namespace

extern "C" size_t ur_chain_tile_bytes(void) { return TC_TILE; }

extern "C" int ur_ff_geglu_fused(const void* x, const void* stream_w, size_t stream_bytes, void* y, long long T, int C, int hidden,
                                 int ldx, int ldy, float ln_eps, int dtype, ur_stream_t stream) {
  UR_REQUIRE(x && stream_w && y, "null pointer");
  UR_REQUIRE_DT(dtype);
  if (C != 320) return ur::fail(UR_E_UNSUPPORTED, "ur_ff_geglu_fused: C must be 320 (token-stationary chain: C/32 accumulator fragments per wave)");
  UR_REQUIRE(T > 0 && T % TC_TOK == 0 && T < (1ll << 31) / (long long)std::max(ldx, ldy), "T must be a positive multiple of 128");
  UR_REQUIRE(hidden > 0 && hidden % 64 == 0 && ldx % 8 == 0 && ldy % 8 == 0 && ldx >= C && ldy >= C, "hidden % 64, ld % 8");
  MlpP p = {};
  p.stream = (const unsigned char*)stream_w; p.x = (const uint16_t*)x; p.y = (uint16_t*)y;
  p.T = (int)T; p.ldx = ldx; p.ldy = ldy; p.hidden = hidden; p.ntiles = 3 * (hidden / 64); p.eps = ln_eps;
  UR_REQUIRE(stream_bytes >= (size_t)p.ntiles * TC_TILE, "weight stream too short");
  hipStream_t s = (hipStream_t)stream;
  const double flops = 2.0 * (double)T * C * hidden * 3.0;
  const double bytes = 2.0 * (double)T * C * 2 + (double)p.ntiles * TC_TILE;
  ur::ProfScope prof("chain_mlp", flops, bytes, s);
  UR_DT_SWITCH(dtype, return (launch_mlp<F16>(p, s)));
  return UR_OK;
}

extern "C" int ur_transformer_head_fused(const void* x, const float* gn_ab, const void* stream_w, size_t stream_bytes, void* h0, void* q,
                                         void* k, void* vt, long long T, int tokens_per_image, int C, float ln_eps, int dtype,
                                         ur_stream_t stream) {
  UR_REQUIRE(x && gn_ab && stream_w && h0 && q && k && vt, "null pointer");
  UR_REQUIRE_DT(dtype);
  if (C != 320) return ur::fail(UR_E_UNSUPPORTED, "ur_transformer_head_fused: C must be 320");
  UR_REQUIRE(T > 0 && tokens_per_image > 0 && tokens_per_image % TC_TOK == 0 && T % tokens_per_image == 0 && T * (long long)C < (1ll << 31),
             "tokens per image must be a multiple of 128 and divide T");
  HeadP p = {};
  p.stream = (const unsigned char*)stream_w; p.x = (const uint16_t*)x; p.ab = gn_ab; p.h0 = (uint16_t*)h0; p.q = (uint16_t*)q;
  p.k = (uint16_t*)k; p.vt = (uint16_t*)vt; p.T = (int)T; p.tok_per_img = tokens_per_image; p.ntiles = 20; p.eps = ln_eps;
  UR_REQUIRE(stream_bytes >= (size_t)p.ntiles * TC_TILE, "weight stream too short");
  hipStream_t s = (hipStream_t)stream;
  ur::ProfScope prof("chain_head", 2.0 * (double)T * C * C * 4.0, 2.0 * (double)T * C * 5 + (double)p.ntiles * TC_TILE, s);
  UR_DT_SWITCH(dtype, return (launch_head<F16>(p, s)));
  return UR_OK;
}

extern "C" int ur_transformer_tail_fused(const void* o1, const void* h0, const void* xres, const void* stream_w, size_t stream_bytes, void* y,
                                         float* gn_part, long long T, int tokens_per_image, int C, int hidden, int heads, int tk,
                                         float ln_eps, float attn_scale, int dtype, ur_stream_t stream) {
  UR_REQUIRE(o1 && h0 && xres && stream_w && y, "null pointer");
  UR_REQUIRE_DT(dtype);
  if (C != 320 || heads != 5 || tk < 1 || tk > 80)
    return ur::fail(UR_E_UNSUPPORTED, "ur_transformer_tail_fused: C must be 320 with 5 heads of 64 and at most 80 context tokens");
  UR_REQUIRE(T > 0 && tokens_per_image > 0 && tokens_per_image % TC_TOK == 0 && T % tokens_per_image == 0 && T * (long long)C < (1ll << 31),
             "tokens per image must be a multiple of 128 and divide T");
  UR_REQUIRE(hidden > 0 && hidden % 64 == 0, "hidden % 64");
  TailP p = {};
  p.stream = (const unsigned char*)stream_w; p.o1 = (const uint16_t*)o1; p.h0 = (const uint16_t*)h0; p.xres = (const uint16_t*)xres;
  p.y = (uint16_t*)y; p.gn_part = gn_part; p.T = (int)T; p.tok_per_img = tokens_per_image; p.hidden = hidden; p.tk = tk;
  p.ntiles = 25 + 3 * (hidden / 64); p.eps = ln_eps; p.scale_log2e = attn_scale * 1.4426950408889634f;
  UR_REQUIRE(stream_bytes >= (size_t)p.ntiles * TC_TILE, "weight stream too short");
  hipStream_t s = (hipStream_t)stream;
  const double flops = 2.0 * (double)T * C * (4.0 * C + 3.0 * hidden) + 4.0 * (double)T * tk * C;
  ur::ProfScope prof("chain_tail", flops, 2.0 * (double)T * C * 4 + (double)p.ntiles * TC_TILE, s);
  UR_DT_SWITCH(dtype, return (launch_tail<F16>(p, s)));
  return UR_OK;
}

extern "C" int ur_csce_fused(const void* x, const void* cond, const void* stream_w, size_t stream_bytes, void* y, float* gn_part, long long T,
                             int tokens_per_image, int C, int Ccond, int dtype, ur_stream_t stream) {
  UR_REQUIRE(x && cond && stream_w && y, "null pointer");
  UR_REQUIRE_DT(dtype);
  if (C != 320 || Ccond != 256) return ur::fail(UR_E_UNSUPPORTED, "ur_csce_fused: C must be 320 and the condition 256 channels wide");
  UR_REQUIRE(T > 0 && tokens_per_image > 0 && tokens_per_image % TC_TOK == 0 && T % tokens_per_image == 0 && T * (long long)C < (1ll << 31),
             "tokens per image must be a multiple of 128 and divide T");
  CsceP p = {};
  p.stream = (const unsigned char*)stream_w; p.x = (const uint16_t*)x; p.cond = (const uint16_t*)cond; p.y = (uint16_t*)y; p.gn_part = gn_part;
  p.T = (int)T; p.tok_per_img = tokens_per_image; p.ntiles = 14;
  UR_REQUIRE(stream_bytes >= (size_t)p.ntiles * TC_TILE, "weight stream too short");
  hipStream_t s = (hipStream_t)stream;
  ur::ProfScope prof("chain_csce", 2.0 * (double)T * C * (Ccond + 2.0 * C), 2.0 * (double)T * (2.0 * C + Ccond) + (double)p.ntiles * TC_TILE, s);
  UR_DT_SWITCH(dtype, return (launch_csce<F16>(p, s)));
  return UR_OK;
}
