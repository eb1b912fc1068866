// Flash-style attention forward for gfx950 (bf16 MFMA 32x32x16, fp32 online softmax).
//
// Layout choices (CDNA4-first):
//   * S^T = K.Q^T ("swapped" product): the accumulator lane owns ONE query column (q = lane&31), so the row
//     max / row sum are 32 in-register ops + one cross-half exchange, and the O rescale is a per-lane scalar.
//   * O^T = V^T.P^T with V stored TRANSPOSED in memory ([channel][key], written that way by the producing
//     GEMM's epilogue): both MFMA operands are then K-contiguous 16-byte LDS reads - no transpose anywhere.
//   * K rows are read from LDS through a bit-2/3 swap of the row index so that the S^T accumulator registers,
//     packed pairwise to bf16, ARE the P^T B-operand (8 consecutive keys per lane) - no shuffles for P.
//   * 4 waves x 32 queries per workgroup share 64-key K / V^T tiles staged through registers into
//     XOR-swizzled LDS (2 stages, one barrier per tile); swizzles verified by tools/lds_conflicts.py.
#include "common.h"
#include "attention_params.h"
#include <cstdlib>

// 128 B of zeros: source of out-of-range 16-byte pieces for the LDS-DMA loader
static __device__ uint4 g_attn_zero_page[8];

namespace {

__device__ __forceinline__ int swap23(int i) { return (i & 0x13) | ((i & 4) << 1) | ((i & 8) >> 1); }

template <int D, bool F16>
#ifndef UR_ATTN_WAVES
#define UR_ATTN_WAVES 3
#endif
#ifndef UR_ATTN_ABL
#define UR_ATTN_ABL 0      // timing-only ablations (profiles/r2_c_pmc_attention.txt): 1 = no exp2, 2 = one P.V MFMA per tile, 3 = no row maximum
#endif
__global__ __launch_bounds__(256, (D == 64 ? UR_ATTN_WAVES : 1)) void attn_fwd_kernel(const AttnP p) {
  constexpr int KROW = D * 2;                 // bytes per K row in LDS
  constexpr int KSLOTS = KROW / 16;           // 16-B slots per K row (8 or 16)
  constexpr int KT = 64 * KROW;               // K tile bytes
  constexpr int VT = D * 128;                 // V^T tile bytes: D rows x 64 keys
  constexpr int STAGE = KT + VT;
  constexpr int DS = D / 16;                  // QK^T k-steps
  constexpr int DF = D / 32;                  // O^T row fragments
  constexpr int KLD = KT / 4096;              // uint4 loads per thread for K tile (2 or 4)
  constexpr int VLD = VT / 4096;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, hf = lane >> 5;
  const int bh = blockIdx.y, b = bh / p.H, h = bh - b * p.H;
  const int q0 = blockIdx.x * 128 + wid * 32;

  const uint16_t* Q = p.q + b * p.bs_q + h * D;
  const uint16_t* K = p.k + b * p.bs_k + h * D;
  const uint16_t* V = p.vt + b * p.bs_vt + (long long)h * D * p.ldvt;

  // Q fragments (B operand): lane -> query q0 + l31, dims ds*16 + hf*8 .. +7
  typedef typename Frag<F16>::type frag_t;
  frag_t qf[DS];
  {
    const int qi = min(q0 + l31, p.Tq - 1);
    const uint16_t* qp = Q + (long long)qi * p.ldq + hf * 8;
#pragma unroll
    for (int ds = 0; ds < DS; ++ds) qf[ds] = *reinterpret_cast<const frag_t*>(qp + ds * 16);
  }

  // K / V^T tiles go global -> LDS by LDS-DMA (global_load_lds_dwordx4: one 1-KiB piece = 8 rows x 128 B per wave
  // instruction, landing lane-linearly), so the XOR swizzle lives on the SOURCE address: physical slot ps of row r gets
  // logical chunk ps ^ f(r).  No staging registers, no ds_write_b128 (each costs ~13 LDS cycles next to the fragment reads).
  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  const uint16_t* zero = reinterpret_cast<const uint16_t*>(g_attn_zero_page);
  constexpr bool DMA = (D == 64);
  const int lr = lane >> 3, ps = lane & 7;
  uint4 kr[DMA ? 1 : KLD], vr[DMA ? 1 : VLD];
  auto load_tile = [&](int kv0, int stage) {
    if (DMA) {
      unsigned char* ks = smem + stage * STAGE;
      unsigned char* vs = ks + KT;
#pragma unroll
      for (int i = 0; i < 2; ++i) {                       // K: 8 pieces of 8 key rows; this wave issues pieces wid and wid + 4
        const int piece = wid + 4 * i, row = piece * 8 + lr;
        const int chunk = ps ^ ((row >> 1) & 7);
        const uint16_t* g = kv0 + row < p.Tk ? K + (long long)(kv0 + row) * p.ldk + chunk * 8 : zero;
        __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(ks + piece * 1024), 16, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {                       // V^T: 8 pieces of 8 channel rows x 64 keys
        const int piece = wid + 4 * i, row = piece * 8 + lr;
        const int chunk = ps ^ ((row >> 1) & 7);
        const uint16_t* g = kv0 + chunk * 8 < p.ldvt ? V + (long long)row * p.ldvt + kv0 + chunk * 8 : zero;
        __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(vs + piece * 1024), 16, 0, 0);
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < (DMA ? 0 : KLD); ++i) {
      const int idx = i * 256 + tid, row = idx / KSLOTS, slot = idx % KSLOTS;
      const bool ok = kv0 + row < p.Tk;
      const uint4* ptr = reinterpret_cast<const uint4*>(ok ? K + (long long)(kv0 + row) * p.ldk + slot * 8 : K);
      uint4 v = *ptr;
      kr[i] = ok ? v : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < (DMA ? 0 : VLD); ++i) {
      const int idx = i * 256 + tid, row = idx >> 3, slot = idx & 7;
      const bool ok = kv0 + slot * 8 < p.ldvt;   // padding columns [Tk, ldvt) are zero by contract
      const uint4* ptr = reinterpret_cast<const uint4*>(ok ? V + (long long)row * p.ldvt + kv0 + slot * 8 : V);
      uint4 v = *ptr;
      vr[i] = ok ? v : make_uint4(0, 0, 0, 0);
    }
  };
  auto store_tile = [&](int stage) {
    if (DMA) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); return; }     // the DMA pieces of this stage have landed
    unsigned char* ks = smem + stage * STAGE;
    unsigned char* vs = ks + KT;
#pragma unroll
    for (int i = 0; i < (DMA ? 0 : KLD); ++i) {
      const int idx = i * 256 + tid, row = idx / KSLOTS, slot = idx % KSLOTS;
      const int sw = (D == 64) ? ((row >> 1) & 7) : (row & 15);
      *reinterpret_cast<uint4*>(ks + row * KROW + ((slot ^ sw) << 4)) = kr[i];
    }
#pragma unroll
    for (int i = 0; i < (DMA ? 0 : VLD); ++i) {
      const int idx = i * 256 + tid, row = idx >> 3, slot = idx & 7;
      *reinterpret_cast<uint4*>(vs + row * 128 + ((slot ^ ((row >> 1) & 7)) << 4)) = vr[i];
    }
  };

  f32x16 oacc[DF];
#pragma unroll
  for (int f = 0; f < DF; ++f)
#pragma unroll
    for (int e = 0; e < 16; ++e) oacc[f][e] = 0.f;
  float m_run = -1e30f, l_run = 0.f;

  const int ntiles = (p.Tk + 63) / 64;
  load_tile(0, 0);
  store_tile(0);
  __syncthreads();
  int stage = 0;
  for (int t = 0; t < ntiles; ++t) {
    const bool more = t + 1 < ntiles;
    if (more) load_tile((t + 1) * 64, stage ^ 1);
    const unsigned char* ks = smem + stage * STAGE;
    const unsigned char* vs = ks + KT;

    // ---- S^T = K Q^T : two 32-key fragments -----------------------------------------------------------
    f32x16 sacc[2];
#pragma unroll
    for (int f = 0; f < 2; ++f) {
#pragma unroll
      for (int e = 0; e < 16; ++e) sacc[f][e] = 0.f;
      const int row = f * 32 + swap23(l31);
      const int sw = (D == 64) ? ((row >> 1) & 7) : (row & 15);
#pragma unroll
      for (int ds = 0; ds < DS; ++ds) {
        const frag_t kf = *reinterpret_cast<const frag_t*>(ks + row * KROW + (((ds * 2 + hf) ^ sw) << 4));
        sacc[f] = mfma16t(kf, qf[ds], sacc[f]);
      }
    }
    // register r of fragment f, half hf  <->  key  t*64 + f*32 + 16*(r>>3) + 8*hf + (r&7)
    float s[32];
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int r = 0; r < 16; ++r) s[f * 16 + r] = sacc[f][r];
    if ((t + 1) * 64 > p.Tk) {
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = t * 64 + f * 32 + 16 * (r >> 3) + 8 * hf + (r & 7);
          if (key >= p.Tk) s[f * 16 + r] = -INFINITY;
        }
    }
    // online softmax in the exp2 domain on RAW scores: p = exp2(c*s - c*m), c = scale*log2(e) folded into one FMA.
    float mx = s[0];
#if UR_ATTN_ABL != 3
#pragma unroll
    for (int i = 1; i < 32; ++i) mx = fmaxf(mx, s[i]);
#endif
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float c = p.scale_log2e;
    // deferred rescale: while no row's maximum grew by more than 2^8 (in the exp2 domain) keep the old reference
    // maximum - P is then bounded by 256 instead of 1 (harmless in bf16/fp32) and the O / l rescale is skipped.
    if (!__all((mx - m_run) * c <= 8.0f)) {
      const float m_new = fmaxf(m_run, mx);
      const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
      l_run *= alpha;
      m_run = m_new;
#pragma unroll
      for (int f = 0; f < DF; ++f)
#pragma unroll
        for (int e = 0; e < 16; ++e) oacc[f][e] *= alpha;
    }
    const float mc = m_run * c;
    float psum = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
#if UR_ATTN_ABL == 1
      s[i] = fmaf(s[i], c, -mc);
#else
      s[i] = __builtin_amdgcn_exp2f(fmaf(s[i], c, -mc));
#endif
      psum += s[i];
    }
    l_run += psum;

    // ---- O^T += V^T P^T : 4 k-steps of 16 keys; P^T fragment = 8 consecutive accumulator registers ----
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      uint4 pk;
      const float* sp = s + kk * 8;
      pk.x = Act<F16>::pack2(sp[0], sp[1]); pk.y = Act<F16>::pack2(sp[2], sp[3]);
      pk.z = Act<F16>::pack2(sp[4], sp[5]); pk.w = Act<F16>::pack2(sp[6], sp[7]);
      const frag_t pf = __builtin_bit_cast(frag_t, pk);
#pragma unroll
      for (int f = 0; f < DF; ++f) {
        const int row = f * 32 + l31;
        const frag_t vf = *reinterpret_cast<const frag_t*>(vs + row * 128 + (((kk * 2 + hf) ^ ((row >> 1) & 7)) << 4));
#if UR_ATTN_ABL == 2
        if (f == 0 && kk == 0) oacc[f] = mfma16t(vf, pf, oacc[f]);
        else oacc[f][0] += __builtin_bit_cast(float, __builtin_bit_cast(uint4, vf).x) * s[kk * 8 + f];
#else
        oacc[f] = mfma16t(vf, pf, oacc[f]);
#endif
      }
    }
    if (more) store_tile(stage ^ 1);
    __syncthreads();
    stage ^= 1;
  }

  // ---- normalise and store: lane owns query q0+l31, channels f*32 + 8*g + 4*hf + e ------------------------
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.f / l_tot;
  const int qi = q0 + l31;
  if (qi < p.Tq) {
    uint16_t* op = p.o + b * p.bs_o + (long long)qi * p.ldo + h * D;
#pragma unroll
    for (int f = 0; f < DF; ++f)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c = f * 32 + 8 * g + 4 * hf;
        *reinterpret_cast<uint2*>(op + c) =
            make_uint2(Act<F16>::pack2(oacc[f][g * 4] * inv, oacc[f][g * 4 + 1] * inv),
                       Act<F16>::pack2(oacc[f][g * 4 + 2] * inv, oacc[f][g * 4 + 3] * inv));
      }
  }
}

}  // namespace

// Built twice (-DUR_TU_F16=0 / 1), one object per 16-bit type: instantiations of one kernel template that share a translation
// unit perturb each other's register allocation (the bf16 kernel measured 8 % slower next to its fp16 twin).
#ifndef UR_TU_F16
#define UR_TU_F16 0
#endif
#if UR_TU_F16
#define UR_ATTN_LAUNCH ur_attn_launch_f16
#else
#define UR_ATTN_LAUNCH ur_attn_launch_bf16
#endif
int ur_attn_launch_bf16(const void* pp, int D, hipStream_t s);
int ur_attn_launch_f16(const void* pp, int D, hipStream_t s);
int ur_attn512_launch_bf16(const void* pp, hipStream_t s);
int ur_attn512_launch_f16(const void* pp, hipStream_t s);
int ur_attn_pp_launch_bf16(const void* pp, size_t ws_bytes, hipStream_t s);      // attention_pp.hip
int ur_attn_pp_launch_f16(const void* pp, size_t ws_bytes, hipStream_t s);

int UR_ATTN_LAUNCH(const void* pp, int D, hipStream_t s) {
  const AttnP& p = *static_cast<const AttnP*>(pp);
  constexpr bool F16 = UR_TU_F16 != 0;
  dim3 grid((p.Tq + 127) / 128, p.B * p.H), block(256);
  if (D == 64) {
    constexpr int lds = 2 * (64 * 128 + 64 * 128);
    hipLaunchKernelGGL((attn_fwd_kernel<64, F16>), grid, block, lds, s, p);
  } else {
    constexpr int lds = 2 * (64 * 256 + 128 * 128);
    static ur::DeviceOnce attr_once;    // the attribute is per device
    if (auto once_guard = attr_once.first()) {
      hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_kernel<128, F16>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    }
    hipLaunchKernelGGL((attn_fwd_kernel<128, F16>), grid, block, lds, s, p);
  }
  return ur::check_launch("ur_attention_fwd");
}

#if !UR_TU_F16
extern "C" size_t ur_attention_workspace_bytes(int B, int H, int Tq, int Tk, int D) {
  return B > 0 && H > 0 && Tq > 0 && Tk > 0 ? attn_pp_ws_bytes(attn_pp_split_tiles(B, H, Tq, Tk, D)) : 0;      // (attention_params.h: the one rule)
}

extern "C" int ur_attention_fwd_ws(const void* q, const void* k, const void* vt, void* o, int B, int H, int Tq, int Tk, int D,
                                   int ldq, int ldk, int ldvt, int ldo, long long bs_q, long long bs_k, long long bs_vt,
                                   long long bs_o, float scale, void* ws, size_t ws_bytes, int dtype, ur_stream_t stream) {
  UR_REQUIRE(q && k && vt && o, "null pointer");
  UR_REQUIRE(D == 64 || D == 128 || D == 512, "head dim must be 64, 128 or 512 (use the GEMM path otherwise)");
  UR_REQUIRE_DT(dtype);
  UR_REQUIRE(B > 0 && H > 0 && Tq > 0 && Tk > 0, "empty problem");
  UR_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ldvt % 8 == 0 && ldo % 4 == 0 && ldvt >= Tk, "leading dims");
  UR_REQUIRE(!ws || ((uintptr_t)ws & 15) == 0, "workspace must be 16-byte aligned");
  AttnP p;
  p.q = (const uint16_t*)q; p.k = (const uint16_t*)k; p.vt = (const uint16_t*)vt; p.o = (uint16_t*)o;
  p.B = B; p.H = H; p.Tq = Tq; p.Tk = Tk; p.ldq = ldq; p.ldk = ldk; p.ldvt = ldvt; p.ldo = ldo;
  p.bs_q = bs_q; p.bs_k = bs_k; p.bs_vt = bs_vt; p.bs_o = bs_o;
  p.scale_log2e = (float)((double)scale * 1.4426950408889634);      // (scale = ln 2 gives exactly 1: q arrives pre-scaled)
  p.n_full = 0; p.ws = (float*)ws;
  hipStream_t s = (hipStream_t)stream;
  const double flops = 4.0 * B * H * (double)Tq * Tk * D;
  const double bytes = 2.0 * B * H * ((double)Tq * D * 2 + (double)Tk * D * 2);
  ur::ProfScope prof("attention", flops, bytes, s);
  if (D == 512) return dtype == UR_DT_F16 ? ur_attn512_launch_f16(&p, s) : ur_attn512_launch_bf16(&p, s);     // attention512.hip
  // self-attention shapes: the ping-pong kernel (attention_pp.hip: 256 queries per workgroup, one workgroup per CU) when its
  // grid fills whole rounds of the 256 CUs well enough - counting a split last round (workspace given) as half a round.
  // 640 workgroups = 2.5 rounds: 203 us unsplit against 229 for the 128-query kernel below; 320 = 1.25 rounds unsplit: 42 against
  // 39 us.  UR_ATTN_NOPP=1 keeps the round-1 kernel everywhere (A/B, tests).
  static const bool nopp = getenv("UR_ATTN_NOPP") && atoi(getenv("UR_ATTN_NOPP")) != 0;
  if (!nopp && attn_pp_shape_ok(p, D)) {          // (anything else - ragged tiles, unaligned views - takes the kernel below, which has tails)
    const long long wgs = (long long)(Tq / 256) * B * H;
    const long long rs = attn_pp_split_tiles(B, H, Tq, Tk, D);
    const long long r = ws && ws_bytes >= attn_pp_ws_bytes(rs) ? rs : 0;
    const double rounds = r ? (double)(wgs - r) / 256 + 0.5 : (double)((wgs + 255) / 256);
    if (wgs <= 256 || wgs >= rounds * 256 * 0.75)
      return dtype == UR_DT_F16 ? ur_attn_pp_launch_f16(&p, ws_bytes, s) : ur_attn_pp_launch_bf16(&p, ws_bytes, s);
  }
  return dtype == UR_DT_F16 ? ur_attn_launch_f16(&p, D, s) : ur_attn_launch_bf16(&p, D, s);
}

extern "C" int ur_attention_fwd(const void* q, const void* k, const void* vt, void* o, int B, int H, int Tq, int Tk, int D,
                                int ldq, int ldk, int ldvt, int ldo, long long bs_q, long long bs_k, long long bs_vt,
                                long long bs_o, float scale, int dtype, ur_stream_t stream) {
  return ur_attention_fwd_ws(q, k, vt, o, B, H, Tq, Tk, D, ldq, ldk, ldvt, ldo, bs_q, bs_k, bs_vt, bs_o, scale, nullptr, 0, dtype, stream);
}
#endif
