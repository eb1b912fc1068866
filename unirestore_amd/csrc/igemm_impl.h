#pragma once
// Kernel and launcher templates of the implicit GEMM (included by igemm.hip and the igemm_*.hip instantiation units;
// the launchers are instantiated in four translation units so that hipcc compiles them in parallel).
// Implicit-GEMM convolution / linear for gfx950: NHWC bf16 activations, [Cout][KH*KW*Cin] bf16 weights,
// v_mfma_f32_32x32x16_bf16 with fp32 accumulate, fused epilogues.  See include/unirestore_hip.h.
//
// Mapping (MI355X-first, not a cuDNN-style port):
//   * GEMM view: M = N*OH*OW output pixels, N = Cout, K = KH*KW*Cin; the K index runs (tap, cin) so every
//     16-byte vector a lane loads is 8 contiguous input channels of ONE pixel (NHWC) or of one weight row.
//   * MFMA "A" operand = weight rows (cout), "B" operand = pixels, so each lane of the 32x32 accumulator
//     holds 4 CONSECUTIVE output channels of one pixel -> 8-byte bf16 stores straight into NHWC.
//   * 256 threads = 4 waves; register-staged global->LDS pipeline (issue tile t+1 loads, compute tile t,
//     write LDS, one barrier per 64-deep K tile), 2 LDS stages.
//   * LDS tile = [row][128 B] with the 16-B slot XOR-swizzled by (row>>1)&7: conflict-free for both the
//     ds_write_b128 staging pattern and the ds_read_b128 fragment pattern of 32-row MFMA operands.
//   * zero padding, stride, nearest-2x upsample and channel concat are address arithmetic in the loader.
//   * block id -> tile map is XCD-aware (blocks that share an activation tile land on one XCD's L2).
#include <cstdio>
#include <cstdlib>
#include <map>
#include <string>

#include "common.h"

// 128 B of zeros: the global source of every padded / out-of-range 16-byte piece of the LDS-DMA loader.
static __device__ uint4 g_zero_page[8];
typedef uint32_t ig_u32x4 __attribute__((ext_vector_type(4)));

namespace {

struct ConvK {
  const uint16_t* x; const uint16_t* x2; const uint16_t* w; const float* bias; const uint16_t* res;
  void* y; uint16_t* yt; float* ws; float* gn_part; const float* gn_ab;
  float* row_stats; const float* ln_stats; const float* ln_colsum; float ln_eps; int ln_dim, ln_parts;
  int dry, plan_tn;
  int N, H, W, C1, ldx, C2, ldx2, Cin, Cout, ldw, ldy, ldr, KH, KW, stride, pad_t, pad_l, OH, OW, OHW;
  int ups, act, out_f32, n_split, t_rows, t_ld;
  float out_scale;
  int M, Ktot, nk, tiles_m, tiles_n, splitk, nk_per_split, nbatch;
  long long bs_x, bs_x2, bs_w, bs_bias, bs_y, bs_r, bias_img;
  size_t ws_bytes_;
  int dbg, gn_fused, gn_parts, gn_silu, f16, staged_ok_, prologue_ok;
  int patch_tw, patch_m0_unused;   // >0: tile rows are an (BM/patch_tw) x patch_tw pixel patch of one image (halo kernel)
  int kcm;   // 1: K runs (64-channel chunk, tap, channel) - the 9 taps of a chunk are consecutive K tiles (L2 reuse)
  const uint16_t* wf;   // fragment-major copy of w (ur_conv_desc.w_frag) or null
  int wmajor;           // 1: weight-major XCD map (igemm_halo_img_kernel: 1-D grid; GEMM kernels: an XCD owns a band of columns)
  int xgm, xbn;         // xgm > 0: 2-D XCD partition of the tile grid (xcd_tile_map): xgm row bands x (8 / xgm) column bands, xbn = tiles_n * xgm / 8
};

// blockIdx.x -> (row tile, column tile).  Workgroup id runs on XCD id % 8; every XCD has its own L2, so which tiles an XCD owns decides
// how often the operands cross the fabric: with xgm row bands x (8 / xgm) column bands the weights are fetched by xgm L2s and the
// activations by 8 / xgm (pick_xcd_grid chooses per launch).  xgm == 0: the rounds-1-5 map - a contiguous run of tiles per XCD, column
// tile fastest (= 8 row bands, no divisibility needed), or, wmajor, row tile fastest (= 8 column bands).
__device__ __forceinline__ void xcd_tile_map(const ConvK& p, int id, int& tm, int& tn) {
  const int xcd = id & 7, idx = id >> 3;
  if (p.xgm) {
    const int lm = idx / p.xbn, ln = idx - lm * p.xbn;
    tm = (xcd % p.xgm) * (p.tiles_m / p.xgm) + lm;
    tn = (xcd / p.xgm) * p.xbn + ln;
    return;
  }
  const int nt = p.tiles_m * p.tiles_n, q = nt >> 3, r = nt & 7;
  id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  if (p.wmajor) { tn = id / p.tiles_m; tm = id - tn * p.tiles_m; }
  else { tm = id / p.tiles_n; tn = id - tm * p.tiles_n; }
}

// Host: the XCD grid with the least operand traffic through the fabric, gm * W + (8 / gm) * A (W, A = weight / activation elements of
// the launch), among the shapes the tile grid divides into; kept only when it beats the contiguous-run map (8 * W + A) by >= 10 %.
// Measured where it matters - the 16x16 / 8x8 levels, whose launches move 2 - 4 TB/s through the fabric with the old map
// (profiles/r6_pmc_conv.json, profiles/r5_pmc_gemm.json: HBM-side fetch 3.7 - 7.3 x the algorithmic bytes).
static inline void pick_xcd_grid(ConvK& k, double W, double A) {
  k.xgm = 0; k.xbn = 0; k.wmajor = 0;
  static const bool off = getenv("UR_NOXCDGRID") != nullptr;
  if (off || k.nbatch != 1) return;
  const long long nt = (long long)k.tiles_m * k.tiles_n;
  double best = 0.9 * (8.0 * W + A);
  static const int force = getenv("UR_XCD_FORCE") ? atoi(getenv("UR_XCD_FORCE")) : 0;      // A/B: this row-band count wherever the grid divides
  if (force && nt % 8 == 0 && 8 % force == 0 && k.tiles_m % force == 0 && k.tiles_n % (8 / force) == 0) {
    if (force < 8) { k.xgm = force; k.xbn = k.tiles_n / (8 / force); }
    return;
  }
  if (nt % 8 == 0)
    for (int gm = 4; gm >= 1; gm >>= 1) {
      const int gn = 8 / gm;
      if (k.tiles_m % gm || k.tiles_n % gn) continue;
      const double c = gm * W + gn * A;
      if (c < best) { best = c; k.xgm = gm; k.xbn = k.tiles_n / gn; }
    }
  if (!k.xgm && k.tiles_n >= 8 && W + 8.0 * A < best) k.wmajor = 1;       // ragged grids: contiguous runs, row tile fastest
}

__device__ __forceinline__ bool is_pair_act(int act) { return act == UR_ACT_GEGLU || act == UR_ACT_GATE; }

// tile row r -> global output row.  Linear tiles: m0 + r.  Patch tiles (halo kernel): m0 is the patch's first pixel and
// rows run (py, px) over a patch_tw-wide window of an OW-wide image.
__device__ __forceinline__ int tile_row_to_m(const ConvK& p, int m0, int r) {
  return p.patch_tw ? m0 + (r / p.patch_tw) * p.OW + (r % p.patch_tw) : m0 + r;
}

// Final stage for 4 consecutive output channels [co, co+4) of pixel row m (values already activated/scaled).
template <bool F16>
__device__ __forceinline__ void epi_store(const ConvK& p, int gb, int m, int co, float v[4]) {
  if (p.res) {
    const uint16_t* r = p.res + gb * p.bs_r + (long long)m * p.ldr + co;
    uint2 rv = *reinterpret_cast<const uint2*>(r);
    v[0] += Act<F16>::lo(rv.x); v[1] += Act<F16>::hi(rv.x);
    v[2] += Act<F16>::lo(rv.y); v[3] += Act<F16>::hi(rv.y);
  }
  if (p.yt && co >= p.n_split) {
    int b = m / p.t_rows, t = m - b * p.t_rows;
    int cw = p.Cout - p.n_split;
    uint16_t* o = p.yt + ((long long)b * cw + (co - p.n_split)) * p.t_ld + t;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[(long long)e * p.t_ld] = f2h16<F16>(v[e]);
    return;
  }
  if (!p.y) return;
  if (p.out_f32) {
    float* o = reinterpret_cast<float*>(p.y) + gb * p.bs_y + (long long)m * p.ldy + co;
    *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
  } else {
    uint16_t* o = reinterpret_cast<uint16_t*>(p.y) + gb * p.bs_y + (long long)m * p.ldy + co;
    *reinterpret_cast<uint2*>(o) = make_uint2(Act<F16>::pack2(v[0], v[1]), Act<F16>::pack2(v[2], v[3]));
  }
}

// bias + activation + scale on a quad; `g` is the gate quad for pair activations. co_in = column in the
// GEMM's N space (pre-pairing); returns the output column.
__device__ __forceinline__ int epi_act(const ConvK& p, int gb, int m, int co_in, float a[4], const float g[4]) {
  const long long boff = gb * p.bs_bias + (p.bias_img ? (long long)(m / p.OHW) * p.bias_img : 0);
  if (p.bias) {
    const float* b = p.bias + boff + co_in;
    float4 bv = *reinterpret_cast<const float4*>(b);
    a[0] += bv.x; a[1] += bv.y; a[2] += bv.z; a[3] += bv.w;
  }
  int co = co_in;
  if (is_pair_act(p.act)) {
    float gg[4] = {g[0], g[1], g[2], g[3]};
    if (p.bias) {
      float4 bv = *reinterpret_cast<const float4*>(p.bias + boff + co_in + 32);
      gg[0] += bv.x; gg[1] += bv.y; gg[2] += bv.z; gg[3] += bv.w;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) a[e] *= (p.act == UR_ACT_GEGLU) ? gelu_f(gg[e]) : gg[e];
    co = (co_in >> 6) * 32 + (co_in & 31);
  } else if (p.act != UR_ACT_NONE) {
#pragma unroll
    for (int e = 0; e < 4; ++e) a[e] = apply_act(a[e], p.act);
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) a[e] *= p.out_scale;
  return co;
}

// Shared epilogue: acc[FN][FM] 32x32 fragments of the wave tile at (m0 + wm*WTM, n0 + wn*WTN).
// Row-contiguous tile <-> global copy with 16-byte accesses; NC (tile columns) is a compile-time constant so the
// (row, chunk) split is a multiply-shift, and LDS rows (stride SROW, 8-byte aligned) are touched with ds_*_b64.
// Loads are issued in batches of U before any is consumed (latency paid once per batch, not once per chunk).
template <int BM, int NC, int NT, int SROW, bool LOAD>
__device__ __forceinline__ void tile_copy(const ConvK& p, uint16_t* g, long long ld, unsigned char* smem, int m0, int M, int c0,
                                          int cmax) {
  constexpr int Q = NC / 8;                               // 16-byte chunks per row
  constexpr int TOT = BM * Q, U = 5;
  for (int i0 = threadIdx.x; i0 < TOT; i0 += NT * U) {
    uint4 v[U];
    uint16_t* gp[U];
    uint2* lp[U];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = i0 + u * NT;
      const int r = i / Q, c = (i - r * Q) * 8;
      const int m = tile_row_to_m(p, m0, i < TOT ? r : 0);
      ok[u] = i < TOT && m < M && c < cmax;
      gp[u] = g + (long long)(ok[u] ? m : m0) * ld + c0 + (ok[u] ? c : 0);
      lp[u] = reinterpret_cast<uint2*>(smem + (ok[u] ? r * SROW + c * 2 : 0));
      if (LOAD) v[u] = *reinterpret_cast<const uint4*>(gp[u]);
      else { const uint2 a = lp[u][0], b = lp[u][1]; v[u] = make_uint4(a.x, a.y, b.x, b.y); }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (!ok[u]) continue;
      if (LOAD) { lp[u][0] = make_uint2(v[u].x, v[u].y); lp[u][1] = make_uint2(v[u].z, v[u].w); }
      else {
store16_wt(gp[u], v[u]);      // write-through for 3x3 and 1x1 launches alike (3x3 only: +3.7 ms, 1x1 only: +0.9 ms; common.h)
      }
    }
  }
}

// Generic (unstaged) epilogue: fp32 outputs, transposed outputs, fused column sums, odd leading dimensions.
template <int FM, int FN, int WTM, int WTN, bool F16>
__device__ __forceinline__ void igemm_epilogue_direct(const ConvK& p, f32x16 (&acc)[FN][FM], int m0, int n0, int wm, int wn,
                                                      int lane, int gb) {
  const int fhalf = lane >> 5, mrow = lane & 31;
  const bool pair = is_pair_act(p.act);
#pragma unroll
  for (int a = 0; a < FN; ++a) {
    if (pair && (a & 1)) continue;
#pragma unroll
    for (int b = 0; b < FM; ++b) {
      int m = tile_row_to_m(p, m0, wm * WTM + b * 32 + mrow);          // (patch tiles of the halo kernels: rows run over an image window)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        int co_in = n0 + wn * WTN + a * 32 + rg * 8 + fhalf * 4;
        bool ok = m < p.M && co_in < p.Cout;
        float v[4], g[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[a][b][rg * 4 + e];
        if (pair) {
          constexpr int a1 = (FN > 1) ? 1 : 0;
#pragma unroll
          for (int e = 0; e < 4; ++e) g[e] = acc[(a + a1) % FN][b][rg * 4 + e];
        }
        int co = co_in;
        if (ok) co = epi_act(p, gb, m, co_in, v, g);
        if (ok) epi_store<F16>(p, gb, m, co, v);
      }
    }
  }
}

// Shared epilogue.  The common case (bf16 output) is STAGED: bias row and residual tile are brought into LDS with
// coalesced 16-byte loads (the K ring is free by then), the fragment pass is branch-free LDS-only math that
// overwrites the residual tile in place with the result, and the tile leaves with 16-byte row-contiguous stores.
// (The MFMA fragment layout would otherwise touch 16-byte runs per pixel per instruction, and a global bias load
// inside each quad's branch serialised ~20 memory latencies per lane.)
// Fragment pass of the staged epilogue: accumulators -> (LayerNorm transform) -> bias -> activation -> scale -> (+residual
// from LDS) -> bf16 into the staged tile.  The feature flags are template parameters (0 = off, 1 = on, 2 = decided at run
// time): the pass is unrolled FN x 4 x FM times, and with run-time flags every instance carries every feature - measured
// 2.2 us of instruction-fetch stalls per launch on a cold CU.  The caller picks a lean specialisation once per tile.
template <int FM, int FN, int WTM, int WTN, int BM, int BN, int SROW, int PAIR, int LN, int MULTI, int YT, int ACT, bool F16>
__device__ __forceinline__ void epi_frag_pass(const ConvK& p, f32x16 (&acc)[FN][FM], int m0, int n0, int c0, int wm, int wn, int lane,
                                              int gb, unsigned char* smem, const float* sbias, const float* scol, const float* srow) {
  const int fhalf = lane >> 5, mrow = lane & 31;
  const bool pair_ = PAIR == 2 ? is_pair_act(p.act) : PAIR != 0;
  const bool ln_ = LN == 2 ? p.ln_stats != nullptr : LN != 0;
  const bool multi_ = MULTI == 2 ? (p.bias_img && !p.patch_tw && p.OHW < BM) : MULTI != 0;
  const bool yt_ = YT == 2 ? p.yt != nullptr : YT != 0;
  const bool act_ = ACT == 2 ? p.act != UR_ACT_NONE : ACT != 0;
  const bool has_res = p.res != nullptr;
#pragma unroll
  for (int a = 0; a < FN; ++a) {
    if (pair_ && (a & 1)) continue;
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const int ln = wn * WTN + a * 32 + rg * 8 + fhalf * 4;          // column inside the tile (GEMM-N space)
      const int co_in = n0 + ln;
      float4 bv = *reinterpret_cast<const float4*>(sbias + ln);
      float4 gv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (pair_) gv = *reinterpret_cast<const float4*>(sbias + ln + 32);
      const int lco = pair_ ? ((ln >> 6) * 32 + (ln & 31)) : ln;        // column inside the OUTPUT tile
      const int co = c0 + lco;
      const bool col_ok = co_in < p.Cout;
      float4 sv = make_float4(0.f, 0.f, 0.f, 0.f), sg = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ln_) {
        sv = *reinterpret_cast<const float4*>(scol + ln);
        if (pair_) sg = *reinterpret_cast<const float4*>(scol + ln + 32);
      }
#pragma unroll
      for (int b = 0; b < FM; ++b) {
        const int lm = wm * WTM + b * 32 + mrow;
        if (multi_) {                                  // tile spans several images: this row's own bias row
          const int il = lm / p.OHW;
          bv = *reinterpret_cast<const float4*>(sbias + il * BN + ln);
          if (pair_) gv = *reinterpret_cast<const float4*>(sbias + il * BN + ln + 32);
        }
        float v[4], gq[4] = {0.f, 0.f, 0.f, 0.f};
        constexpr int a1 = (FN > 1) ? 1 : 0;
        if (ln_) {                                            // LayerNorm folded into this GEMM (see above)
          const float mean = srow[2 * lm], rstd = srow[2 * lm + 1];
          v[0] = rstd * (acc[a][b][rg * 4] - mean * sv.x) + bv.x;     v[1] = rstd * (acc[a][b][rg * 4 + 1] - mean * sv.y) + bv.y;
          v[2] = rstd * (acc[a][b][rg * 4 + 2] - mean * sv.z) + bv.z; v[3] = rstd * (acc[a][b][rg * 4 + 3] - mean * sv.w) + bv.w;
          if (pair_) {
            gq[0] = rstd * (acc[(a + a1) % FN][b][rg * 4] - mean * sg.x) + gv.x;     gq[1] = rstd * (acc[(a + a1) % FN][b][rg * 4 + 1] - mean * sg.y) + gv.y;
            gq[2] = rstd * (acc[(a + a1) % FN][b][rg * 4 + 2] - mean * sg.z) + gv.z; gq[3] = rstd * (acc[(a + a1) % FN][b][rg * 4 + 3] - mean * sg.w) + gv.w;
          }
        } else {
          v[0] = acc[a][b][rg * 4] + bv.x; v[1] = acc[a][b][rg * 4 + 1] + bv.y;
          v[2] = acc[a][b][rg * 4 + 2] + bv.z; v[3] = acc[a][b][rg * 4 + 3] + bv.w;
          if (pair_) {
            gq[0] = acc[(a + a1) % FN][b][rg * 4] + gv.x; gq[1] = acc[(a + a1) % FN][b][rg * 4 + 1] + gv.y;
            gq[2] = acc[(a + a1) % FN][b][rg * 4 + 2] + gv.z; gq[3] = acc[(a + a1) % FN][b][rg * 4 + 3] + gv.w;
          }
        }
        if (pair_) {
          const float g0 = gq[0], g1 = gq[1], g2 = gq[2], g3 = gq[3];
          if (p.act == UR_ACT_GEGLU) { v[0] *= gelu_f(g0); v[1] *= gelu_f(g1); v[2] *= gelu_f(g2); v[3] *= gelu_f(g3); }
          else { v[0] *= g0; v[1] *= g1; v[2] *= g2; v[3] *= g3; }
        } else if (act_) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], p.act);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= p.out_scale;
        const int mg = tile_row_to_m(p, m0, lm);
        const bool ok = col_ok && mg < p.M;
        if (yt_ && co >= p.n_split) {                                   // transposed columns (V^T) go out directly
          if (ok) epi_store<F16>(p, gb, mg, co, v);
          continue;
        }
        uint2* sp = reinterpret_cast<uint2*>(smem + lm * SROW + lco * 2);
        if (has_res) {
          const uint2 rv = *sp;
          v[0] += Act<F16>::lo(rv.x); v[1] += Act<F16>::hi(rv.x);
          v[2] += Act<F16>::lo(rv.y); v[3] += Act<F16>::hi(rv.y);
        }
        if (ok) *sp = make_uint2(Act<F16>::pack2(v[0], v[1]), Act<F16>::pack2(v[2], v[3]));
      }
    }
  }
}

// Staged epilogue body.  CLS 0 = the plain class (no pair activation, no LayerNorm consumer, one bias row per tile, no
// transposed columns): those features are compiled out, so the executed path is short and contiguous (skipping over
// feature blocks costs an instruction-cache miss per far branch on a cold CU).  CLS 1 = everything, decided at run time.
template <int FM, int FN, int WTM, int WTN, int BM, int BN, int NT, int CLS, bool PAIRC, bool F16>
__device__ __forceinline__ void staged_epilogue(const ConvK& p, f32x16 (&acc)[FN][FM], int m0, int n0, int wm, int wn, int lane,
                                                int gb, unsigned char* smem, bool owner, int nimg_tile_in) {
  // PAIRC: the kernel is only launched for pair activations (host-checked), whose output tile is BN/2 columns wide
  constexpr int SROW = (PAIRC ? BN : BN * 2) + 8;
  const bool pair = CLS ? is_pair_act(p.act) : false;
  const bool ln = CLS ? p.ln_stats != nullptr : false;
  const bool yt = CLS ? p.yt != nullptr : false;
  const int nimg_tile = CLS ? nimg_tile_in : 1;
  const int c0 = pair ? (n0 >> 1) : n0;                  // first output column of this tile
  const int ncols = pair ? BN / 2 : BN;
  const int cmax = min(yt ? p.n_split : (pair ? p.Cout / 2 : p.Cout), c0 + ncols) - c0;     // valid output columns here
  float* sbias = reinterpret_cast<float*>(smem + BM * SROW);                                   // [nimg_tile][BN] floats (GEMM-N order)
  float* scol = sbias + 4 * BN;                                                                 // [BN] LN fusion: column sums of W*gamma
  float* srow = scol + BN;                                                                     // [BM][2] LN fusion: mean, rstd per row
  const int img0 = m0 / p.OHW;
  for (int i = threadIdx.x; i < BN * nimg_tile; i += NT) {
    const int il = i / BN, col = i - il * BN;
    const long long boff = gb * p.bs_bias + (p.bias_img ? (long long)min(img0 + il, p.N - 1) * p.bias_img : 0);
    sbias[i] = (p.bias && n0 + col < p.Cout) ? p.bias[boff + n0 + col] : 0.f;
  }
  if (ln) {   // this GEMM consumes LayerNorm(x): out = rstd*(acc - mean*s[n]) + t[n]  (t arrives as the bias)
    for (int i = threadIdx.x; i < BN; i += NT) scol[i] = n0 + i < p.Cout ? p.ln_colsum[n0 + i] : 0.f;
    for (int r = threadIdx.x; r < BM; r += NT) {
      const int m = min(tile_row_to_m(p, m0, r), p.M - 1);
      float sx = 0.f, sq = 0.f;
      for (int q = 0; q < p.ln_parts; ++q) {               // one partial per N tile of the producer GEMM (no atomics)
        const float2 t2 = *reinterpret_cast<const float2*>(p.ln_stats + 2 * ((long long)q * p.M + m));
        sx += t2.x; sq += t2.y;
      }
      const float mean = sx / p.ln_dim;
      const float var = fmaxf(sq / p.ln_dim - mean * mean, 0.f);
      srow[2 * r] = mean;
      srow[2 * r + 1] = rsqrtf(var + p.ln_eps);
    }
  }
  if (p.res) {                                            // residual tile -> LDS, coalesced
    uint16_t* rb = const_cast<uint16_t*>(p.res) + gb * p.bs_r;
    if (pair) tile_copy<BM, (BN >= 16 ? BN / 2 : 8), NT, SROW, true>(p, rb, p.ldr, smem, m0, p.M, c0, cmax);
    else tile_copy<BM, BN, NT, SROW, true>(p, rb, p.ldr, smem, m0, p.M, c0, cmax);
  }
  __syncthreads();
  if (owner) {
    // lean specialisations for the common launches; everything else takes the all-run-time instance
    const bool multi = nimg_tile > 1, hasact = p.act != UR_ACT_NONE;
#define UR_EPI_PASS(PAIR, LN, MULTI, YT, ACT) \
    epi_frag_pass<FM, FN, WTM, WTN, BM, BN, SROW, PAIR, LN, MULTI, YT, ACT, F16>(p, acc, m0, n0, c0, wm, wn, lane, gb, smem, sbias, scol, srow)
    if constexpr (CLS == 0) { if (hasact) UR_EPI_PASS(0, 0, 0, 0, 1); else UR_EPI_PASS(0, 0, 0, 0, 0); }
    else if constexpr (PAIRC) {                                 // pair-only kernels (256-wide GEGLU / SimpleGate tiles, 128-160 accumulator registers):
      if (ln) UR_EPI_PASS(1, 1, 0, 0, 0); else UR_EPI_PASS(1, 0, 0, 0, 0);      // nothing else is compiled in - every extra instance moved their allocation (704 B of scratch once)
    } else {
      if (pair) { if (ln) UR_EPI_PASS(1, 1, 0, 0, 0); else UR_EPI_PASS(1, 0, 0, 0, 0); }        // never with multi / yt (host-checked)
      else if (yt) {                                                                            // fused QKV with transposed V
        if (hasact) UR_EPI_PASS(0, 2, 0, 1, 2);
        else if (ln) UR_EPI_PASS(0, 1, 0, 1, 0);                // (the two production forms with their flags compiled in: the all-run-time
        else UR_EPI_PASS(0, 0, 0, 1, 0);                        //  instance cost 3 us on 8192 x 1920 x 640 and 50 us on the Controller's 655 360-row QKV)
      }
      else if (multi) { if (hasact) UR_EPI_PASS(0, 0, 1, 0, 2); else UR_EPI_PASS(0, 0, 1, 0, 0); }     // per-image bias rows
      else if (hasact) UR_EPI_PASS(0, 1, 0, 0, 2);                                              // LayerNorm consumer
      else UR_EPI_PASS(0, 1, 0, 0, 0);
    }
#undef UR_EPI_PASS
  }
  if (p.dbg & 16) return;
  __syncthreads();
  if (p.gn_fused) {
    // Fused GroupNorm statistics of the tile just produced (exactly the 16-bit values the consumer will read), deterministic:
    // thread (g, cp) sums column pair cp over row group g from LDS into red[g][moment][cp]; one thread per column then adds
    // the row groups in order and stores the tile's partial (sum, sum of squares) - a plain store into this tile's own slot
    // of the partial plane [N][P][C][2] (no atomics; the host only sets gn_fused when a tile never straddles two images).
    float* red = sbias;                                   // bias / LayerNorm staging is dead by now: [NG][4][CP] floats (<= 4*NT)
    const int CP = ncols >> 1, NG = NT / CP, RGN = (BM + NG - 1) / NG;
    const int cp = threadIdx.x % CP, g = threadIdx.x / CP;
    if (g < NG) {
      float s0 = 0.f, s1 = 0.f, q0 = 0.f, q1 = 0.f;
      if (cp * 2 < cmax) {
        const int r0 = g * RGN, r1 = p.patch_tw ? min(BM, r0 + RGN) : min(min(BM, r0 + RGN), p.M - m0);
        for (int r = r0; r < r1; ++r) {
          const uint32_t w = *reinterpret_cast<const uint32_t*>(smem + r * SROW + cp * 4);
          const float a = Act<F16>::lo(w), b = Act<F16>::hi(w);
          s0 += a; q0 += a * a; s1 += b; q1 += b * b;
        }
      }
      red[(g * 4 + 0) * CP + cp] = s0; red[(g * 4 + 1) * CP + cp] = s1;
      red[(g * 4 + 2) * CP + cp] = q0; red[(g * 4 + 3) * CP + cp] = q1;
    }
    __syncthreads();
    const int ctot = pair ? p.Cout / 2 : p.Cout;
    const int img = m0 / p.OHW, rem = m0 - img * p.OHW;
    // partial slot of this tile inside its image: linear tiles = row block; halo patches = (patch row, patch column)
    const int pm = p.patch_tw ? ((rem / p.OW) / (BM / p.patch_tw)) * (p.OW / p.patch_tw) + (rem % p.OW) / p.patch_tw : rem / BM;
    float* st = p.gn_part + ((((long long)img * p.gn_parts + pm) * p.nbatch + gb) * ctot + c0) * 2;
    for (int i = threadIdx.x; i < cmax; i += NT) {
      const int c2 = i >> 1, odd = i & 1;
      float a = 0.f, b = 0.f;
      for (int gg = 0; gg < NG; ++gg) { a += red[(gg * 4 + odd) * CP + c2]; b += red[(gg * 4 + 2 + odd) * CP + c2]; }
      *reinterpret_cast<float2*>(st + 2 * i) = make_float2(a, b);
    }
  }
  if (p.row_stats) {
    // per-row (sum, sum of squares) over this tile's columns of the bf16 values just produced: the LayerNorm statistics
    // of the consumer GEMM.  NT/BM threads share a row and meet through wave shuffles; one plain store per row into this N tile's plane.
    constexpr int TPR = NT / BM >= 1 ? NT / BM : 1;
    const int r = threadIdx.x / TPR, part = threadIdx.x % TPR;
    if (r < BM) {
      const int m = tile_row_to_m(p, m0, r);
      const int npair = cmax >> 1, per = (npair + TPR - 1) / TPR;
      float sx = 0.f, sq = 0.f;
      for (int j = part * per; j < min(npair, (part + 1) * per); ++j) {
        const uint32_t w = *reinterpret_cast<const uint32_t*>(smem + r * SROW + j * 4);
        const float a = Act<F16>::lo(w), b = Act<F16>::hi(w);
        sx += a + b; sq += a * a + b * b;
      }
      if (TPR >= 2) { sx += __shfl_xor(sx, 1, 64); sq += __shfl_xor(sq, 1, 64); }
      if (TPR >= 4) { sx += __shfl_xor(sx, 2, 64); sq += __shfl_xor(sq, 2, 64); }
      static_assert(TPR == 1 || TPR == 2 || TPR == 4, "row-stat reduction: 1, 2 or 4 threads per row (same wave)");
      const int tn_idx = n0 / BN;
      if (m < p.M && part == 0)
        *reinterpret_cast<float2*>(p.row_stats + 2 * ((long long)tn_idx * p.M + m)) = make_float2(sx, sq);
    }
  }
  if (!p.y) return;                                      // statistics-only launch (pooled output): nothing to store
  uint16_t* yb = reinterpret_cast<uint16_t*>(p.y) + gb * p.bs_y;
  if (pair) tile_copy<BM, (BN >= 16 ? BN / 2 : 8), NT, SROW, false>(p, yb, p.ldy, smem, m0, p.M, c0, cmax);
  else tile_copy<BM, BN, NT, SROW, false>(p, yb, p.ldy, smem, m0, p.M, c0, cmax);
}

// LDS bytes the staged epilogue needs (tile + bias / LayerNorm staging, reused for the GroupNorm partial reduction)
template <int BM, int BN, int NT, bool PAIRC = false>
constexpr int epi_lds_bytes() {
  constexpr int tile = BM * ((PAIRC ? BN : BN * 2) + 8);
  constexpr int aux1 = 5 * BN * 4 + BM * 8, aux2 = 4 * NT * 4;
  return tile + (aux1 > aux2 ? aux1 : aux2);
}

// DIRECT_OK = false: the kernel is only ever launched with a staged-capable output (host-checked), so the unstaged
// store path is not compiled in (its FN x FM x 4 unrolled stores are pure code-size ballast there).
template <int FM, int FN, int WTM, int WTN, int BM, int BN, int NT, bool F16, bool DIRECT_OK = true, bool PAIRC = false>
__device__ __forceinline__ void igemm_epilogue(const ConvK& p, f32x16 (&acc)[FN][FM], int m0, int n0, int wm, int wn, int lane,
                                               int gb, int sz, unsigned char* smem, bool owner = true) {
  // owner: this wave holds accumulator fragments (false for the loader waves of the warp-specialised kernel,
  // which still take part in the barriers and the tile copies)
  const int fhalf = lane >> 5, mrow = lane & 31;
  if (p.splitk > 1) {
    if (!owner) return;
    float* ws = p.ws + ((long long)(sz * p.nbatch + gb) * p.M) * p.Cout;
#pragma unroll
    for (int a = 0; a < FN; ++a)
#pragma unroll
      for (int b = 0; b < FM; ++b) {
        int m = tile_row_to_m(p, m0, wm * WTM + b * 32 + mrow);
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          int co = n0 + wn * WTN + a * 32 + rg * 8 + fhalf * 4;
          if (m < p.M && co < p.Cout)
            // (plain stores: the reduce pass re-reads the planes from L2 at once - written through they cost 6.5 ms per forward)
            *reinterpret_cast<float4*>(ws + (long long)m * p.Cout + co) =
                make_float4(acc[a][b][rg * 4], acc[a][b][rg * 4 + 1], acc[a][b][rg * 4 + 2], acc[a][b][rg * 4 + 3]);
        }
      }
    return;
  }
  const bool pair = is_pair_act(p.act);
  constexpr int SROW = BN * 2 + 8;                       // staged row stride in bytes (+8 spreads the ds_write_b64 banks)
  // per-image bias rows (time embeddings of a schedule-batched Controller): a tile covers 1 image, or up to 4 whole ones
  const int nimg_tile = (p.bias_img && !p.patch_tw && p.OHW < BM) ? BM / p.OHW : 1;
  const bool bias_geom_ok = !p.bias_img || p.patch_tw || (p.OHW % BM) == 0 || ((BM % p.OHW) == 0 && nimg_tile <= 4);
  const bool staged = p.staged_ok_ && BN >= 32 && bias_geom_ok;   // host-evaluated part: bf16 y, 16-byte aligned rows, no colsum
  if (!staged) {
    if (DIRECT_OK && owner) igemm_epilogue_direct<FM, FN, WTM, WTN, F16>(p, acc, m0, n0, wm, wn, lane, gb);
    return;
  }
  const bool plain = !pair && !p.ln_stats && nimg_tile == 1 && !p.yt;
  if (PAIRC) { if (!plain) staged_epilogue<FM, FN, WTM, WTN, BM, BN, NT, 1, true, F16>(p, acc, m0, n0, wm, wn, lane, gb, smem, owner, nimg_tile); }
  else if (plain) staged_epilogue<FM, FN, WTM, WTN, BM, BN, NT, 0, false, F16>(p, acc, m0, n0, wm, wn, lane, gb, smem, owner, nimg_tile);
  else staged_epilogue<FM, FN, WTM, WTN, BM, BN, NT, 1, false, F16>(p, acc, m0, n0, wm, wn, lane, gb, smem, owner, nimg_tile);
}

// ---------------------------------------------------------------------------------------------------------------------
// One 64-deep K tile of a wave's FM x FN fragment tile out of LDS, hand-scheduled (igemm_asm.inc, tools/gen_igemm_asm.py):
// fragment reads run a register pool ahead of the MFMAs, every MFMA waits with a counted lgkmcnt for its own operands.
// ab[b] = LDS byte address of activation fragment b at k-step 0 (row * 128 + ((lane half ^ swizzle) << 4)), aw = the same for the
// wave's first weight fragment (fragment a: + a * 4096); rows are 128-byte, 128-aligned, slots XOR-swizzled with (row >> 1) & 7.
#if UR_IGASM_ABL == 4
#include "../../tools/ab/igemm_asm_abl4.inc"
#elif UR_IGASM_ABL == 5
#include "../../tools/ab/igemm_asm_abl5.inc"
#else
#include "igemm_asm.inc"
#endif
#ifndef UR_HALO_ABL
#define UR_HALO_ABL 0      // timing-only ablations of igemm_halo_kernel (A/B builds): 1 = no DMA waits, 2 = no MFMA body
#endif
#define IG_MN(ASM, ...)                                                                      \
  do {                                                                                       \
    if constexpr (F16) asm volatile(ASM("v_mfma_f32_32x32x16_f16") __VA_ARGS__);             \
    else asm volatile(ASM("v_mfma_f32_32x32x16_bf16") __VA_ARGS__);                          \
  } while (0)
template <int FM, int FN, bool F16>
__device__ __forceinline__ void ktile_mma(f32x16 (&acc)[FN][FM], const unsigned (&ab)[FM], unsigned aw) {
  ig_u32x4 t0, t1, t2, t3, t4, t5, t6, t7, t8, t9;
  unsigned x0, x1, xw;
#define KT_POOL6 "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5)
#define KT_POOL8 KT_POOL6, "=&v"(t6), "=&v"(t7)
#define KT_POOL10 KT_POOL8, "=&v"(t8), "=&v"(t9)
  if constexpr (FM == 1 && FN == 5) {
    IG_MN(IG_ASM_KT_1x5, : "+v"(acc[0][0]), "+v"(acc[1][0]), "+v"(acc[2][0]), "+v"(acc[3][0]), "+v"(acc[4][0]), KT_POOL10, "=&v"(x0), "=&v"(xw)
          : "v"(ab[0]), "v"(aw) : "memory");
  } else if constexpr (FM == 2 && FN == 2) {
    IG_MN(IG_ASM_KT_2x2, : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]), KT_POOL8, "=&v"(x0), "=&v"(x1), "=&v"(xw)
          : "v"(ab[0]), "v"(ab[1]), "v"(aw) : "memory");
  } else if constexpr (FM == 1 && FN == 1) {
    IG_MN(IG_ASM_KT_1x1, : "+v"(acc[0][0]), KT_POOL6, "=&v"(x0), "=&v"(xw) : "v"(ab[0]), "v"(aw) : "memory");
  } else if constexpr (FM == 2 && FN == 1) {
    IG_MN(IG_ASM_KT_2x1, : "+v"(acc[0][0]), "+v"(acc[0][1]), KT_POOL6, "=&v"(x0), "=&v"(x1), "=&v"(xw) : "v"(ab[0]), "v"(ab[1]), "v"(aw) : "memory");
  } else if constexpr (FM == 2 && FN == 4) {
    // (128 / 160 accumulator registers at two waves per SIMD: "a" lets them live in the AGPR half of the 256-register budget)
    IG_MN(IG_ASM_KT_2x4, : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(acc[2][0]), "+v"(acc[2][1]), "+v"(acc[3][0]), "+v"(acc[3][1]),
          KT_POOL6, "=&v"(x0), "=&v"(x1), "=&v"(xw) : "v"(ab[0]), "v"(ab[1]), "v"(aw) : "memory");
  } else if constexpr (FM == 1 && FN == 10) {
    IG_MN(IG_ASM_KT_1x10, : "+v"(acc[0][0]), "+v"(acc[1][0]), "+v"(acc[2][0]), "+v"(acc[3][0]), "+v"(acc[4][0]), "+v"(acc[5][0]), "+v"(acc[6][0]), "+v"(acc[7][0]),
          "+v"(acc[8][0]), "+v"(acc[9][0]), KT_POOL6, "=&v"(x0), "=&v"(xw) : "v"(ab[0]), "v"(aw) : "memory");
  } else {
    static_assert(FM == 0, "no hand-scheduled K tile for this fragment shape");
  }
#undef KT_POOL6
#undef KT_POOL8
#undef KT_POOL10
}
// (GEMM kernels) shapes with a generated K-tile body
template <int FM, int FN> constexpr bool ktile_gemm_ok() {
#if defined(UR_IGEMM_NOASM) || defined(UR_GEMM_NOASM)
  return false;
#else
  // (not the 1 x 10 / 2 x 4 tiles of the 256-wide kernels: 160 / 128 accumulator registers at two waves per SIMD leave no room for
  // the fragment pool - "+v" spilled 96-332 B and "+a" did not allocate at all; hipcc's own schedule stays there)
  return (FM == 1 && (FN == 1 || FN == 5 || FN == 10)) || (FM == 2 && (FN == 1 || FN == 2 || FN == 4));
#endif
}
template <int FM, int FN> constexpr bool ktile_asm_ok() {
#ifdef UR_IGEMM_NOASM
  return false;
#else
  return (FM == 1 && FN == 5) || (FM == 2 && FN == 2);
#endif
}


template <int BM, int BN, int WM, int WN, bool G1, bool F16>
__global__ __launch_bounds__(256) void igemm_kernel(const ConvK p) {
  // G1: pure GEMM (1x1, stride 1, one source, no padding): rows are plain offsets, no im2col state, no tap bookkeeping
  constexpr int WTM = BM / WM, WTN = BN / WN, FM = WTM / 32, FN = WTN / 32, XP = BM / 32, WP = BN / 32;
  constexpr int STAGE = (BM + BN) * 128;
  static_assert(WM * WN == 4 && WTM % 32 == 0 && WTN % 32 == 0, "tile/wave shape");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid % WM, wn = wid / WM;
  const int gb = blockIdx.y, sz = blockIdx.z;

  int tm, tn;
  xcd_tile_map(p, blockIdx.x, tm, tn);                  // XCD-aware bijective remap
  const int m0 = tm * BM, n0 = tn * BN;

  const uint16_t* __restrict__ X1 = p.x + gb * p.bs_x;
  const uint16_t* __restrict__ X2 = p.x2 ? p.x2 + gb * p.bs_x2 : nullptr;
  const uint16_t* __restrict__ Wt = p.w + gb * p.bs_w;

  // ---- loader geometry: thread -> (row = pass*32 + tid/8, 16-B chunk = tid%8) -----------------------
  const int chunk = tid & 7, lrow = tid >> 3;
  int ih0[XP], iw0[XP], nb[XP];
  bool xok[XP];
#pragma unroll
  for (int i = 0; i < XP; ++i) {
    int m = m0 + i * 32 + lrow;
    xok[i] = m < p.M;
    if (G1) { nb[i] = xok[i] ? m * p.ldx : 0; ih0[i] = iw0[i] = 0; continue; }    // nb = element offset of the row
    int mm = xok[i] ? m : 0;
    int n = mm / p.OHW, rem = mm - n * p.OHW;
    int oh = rem / p.OW, ow = rem - oh * p.OW;
    ih0[i] = oh * p.stride - p.pad_t;
    iw0[i] = ow * p.stride - p.pad_l;
    nb[i] = n * p.H;
  }
  long long woff[WP];
  bool wok[WP];
#pragma unroll
  for (int j = 0; j < WP; ++j) {
    int row = n0 + j * 32 + lrow;
    wok[j] = row < p.Cout;
    woff[j] = (long long)(wok[j] ? row : 0) * p.ldw;
  }
  const int Hlim = p.ups ? p.H * 2 : p.H, Wlim = p.ups ? p.W * 2 : p.W;

  const int kt_begin = sz * p.nk_per_split;
  const int kt_end = min(p.nk, kt_begin + p.nk_per_split);
  int kcur = kt_begin * 64 + chunk * 8;
  int tap = 0, cch = kcur;
  const int ntap = p.KH * p.KW;
  if (!G1) {
    tap = kcur / p.Cin; cch = kcur - tap * p.Cin;
    if (p.kcm) { tap = kt_begin % ntap; cch = (kt_begin / ntap) * 64 + chunk * 8; }
  }

  uint4 xr[XP], wr[WP];
  auto load_tile = [&]() {
    const bool kval = kcur < p.Ktot;
    if (G1) {
#pragma unroll
      for (int i = 0; i < XP; ++i) {
        const bool v = kval && xok[i];
        const uint4 val = *reinterpret_cast<const uint4*>(v ? X1 + nb[i] + kcur : X1);
        xr[i] = v ? val : make_uint4(0, 0, 0, 0);
      }
#pragma unroll
      for (int j = 0; j < WP; ++j) {
        const bool v = kval && wok[j];
        const uint4 val = *reinterpret_cast<const uint4*>(v ? Wt + woff[j] + kcur : Wt);
        wr[j] = v ? val : make_uint4(0, 0, 0, 0);
      }
      kcur += 64;
      return;
    }
    const int dy = (p.KW == 1) ? 0 : (tap * 11) >> 5;
    const int dx = tap - dy * p.KW;
    const uint16_t* src = X1;
    int ld = p.ldx, cc = cch;
    if (cch >= p.C1) { src = X2; ld = p.ldx2; cc = cch - p.C1; }
#pragma unroll
    for (int i = 0; i < XP; ++i) {
      int ih = ih0[i] + dy, iw = iw0[i] + dx;
      bool v = kval && xok[i] && (unsigned)ih < (unsigned)Hlim && (unsigned)iw < (unsigned)Wlim;
      if (p.ups) { ih >>= 1; iw >>= 1; }
      long long off = ((long long)(nb[i] + ih) * p.W + iw) * ld + cc;
      const uint4* ptr = reinterpret_cast<const uint4*>(v ? src + off : src);
      uint4 val = *ptr;
      xr[i] = v ? val : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < WP; ++j) {
      bool v = kval && wok[j];
      const uint4* ptr = reinterpret_cast<const uint4*>(v ? Wt + woff[j] + kcur : Wt);
      uint4 val = *ptr;
      wr[j] = v ? val : make_uint4(0, 0, 0, 0);
    }
    kcur += 64;
    if (p.kcm) { if (++tap == ntap) { tap = 0; cch += 64; } }
    else { cch += 64; while (cch >= p.Cin) { cch -= p.Cin; ++tap; } }
  };
  auto store_tile = [&](int stage) {
    unsigned char* xs = smem + stage * STAGE;
    unsigned char* wsm = xs + BM * 128;
#pragma unroll
    for (int i = 0; i < XP; ++i) {
      int row = i * 32 + lrow;
      *reinterpret_cast<uint4*>(xs + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4)) = xr[i];
    }
#pragma unroll
    for (int j = 0; j < WP; ++j) {
      int row = j * 32 + lrow;
      *reinterpret_cast<uint4*>(wsm + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4)) = wr[j];
    }
  };

  f32x16 acc[FN][FM];
#pragma unroll
  for (int a = 0; a < FN; ++a)
#pragma unroll
    for (int b = 0; b < FM; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

  const int frow = lane & 31, fhalf = lane >> 5;
  typedef __attribute__((address_space(3))) void* lptr_t;
  // LDS byte addresses (stage 0, k-step 0) of the wave's first activation / weight fragment for the hand-scheduled K tile
  const unsigned kt_x0 = (unsigned)(uintptr_t)(lptr_t)smem + (wm * WTM + frow) * 128 + ((fhalf ^ ((frow >> 1) & 7)) << 4);
  const unsigned kt_w0 = (unsigned)(uintptr_t)(lptr_t)smem + (BM + wn * WTN + frow) * 128 + ((fhalf ^ ((frow >> 1) & 7)) << 4);
  static_assert(WTM % 32 == 0 && ((WTN >> 1) & 7) == 0, "fragment rows must keep the swizzle phase of row 0");
  auto compute = [&](int stage) {
    if constexpr (ktile_gemm_ok<FM, FN>()) {
      unsigned ab[FM];
#pragma unroll
      for (int b = 0; b < FM; ++b) ab[b] = kt_x0 + stage * STAGE + b * 4096;
      ktile_mma<FM, FN, F16>(acc, ab, kt_w0 + stage * STAGE);
      return;
    }
    const unsigned char* xs = smem + stage * STAGE;
    const unsigned char* wsm = xs + BM * 128;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int slot = ks * 2 + fhalf;
      typename Frag<F16>::type bfr[FM], afr[FN];
#pragma unroll
      for (int b = 0; b < FM; ++b) {
        int row = wm * WTM + b * 32 + frow;
        bfr[b] = *reinterpret_cast<const typename Frag<F16>::type*>(xs + row * 128 + ((slot ^ ((row >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int a = 0; a < FN; ++a) {
        int row = wn * WTN + a * 32 + frow;
        afr[a] = *reinterpret_cast<const typename Frag<F16>::type*>(wsm + row * 128 + ((slot ^ ((row >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int a = 0; a < FN; ++a)
#pragma unroll
        for (int b = 0; b < FM; ++b)
          acc[a][b] = mfma16t(afr[a], bfr[b], acc[a][b]);
    }
  };

  // ---- main loop ---------------------------------------------------------------------------------------
  if (kt_begin < kt_end) {
    load_tile();
    store_tile(0);
    __syncthreads();
    int stage = 0;
    for (int kt = kt_begin; kt < kt_end; ++kt) {
      const bool more = kt + 1 < kt_end;
      if (more) load_tile();
      compute(stage);
      if (more) store_tile(stage ^ 1);
      __syncthreads();
      stage ^= 1;
    }
  }

  if constexpr (ktile_gemm_ok<FM, FN>()) asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");      // last asm MFMA -> VALU reads of the accumulators
  igemm_epilogue<FM, FN, WTM, WTN, BM, BN, 256, F16>(p, acc, m0, n0, wm, wn, lane, gb, sz, smem);
}

template <bool F16>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const ConvK p) {
  const bool pair = is_pair_act(p.act);
  const int qn = p.Cout / 4;  // quads per row in GEMM N space
  long long total = (long long)p.nbatch * p.M * qn;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    int q = (int)(i % qn);
    long long r = i / qn;
    int m = (int)(r % p.M), gb = (int)(r / p.M);
    int co_in = q * 4;
    if (pair && (co_in & 32)) continue;  // gate quads are consumed by their 'a' partner
    float v[4] = {0, 0, 0, 0}, g[4] = {0, 0, 0, 0};
    for (int s = 0; s < p.splitk; ++s) {
      const float* ws = p.ws + ((long long)(s * p.nbatch + gb) * p.M + m) * p.Cout + co_in;
      float4 t = *reinterpret_cast<const float4*>(ws);
      v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w;
      if (pair) {
        float4 u = *reinterpret_cast<const float4*>(ws + 32);
        g[0] += u.x; g[1] += u.y; g[2] += u.z; g[3] += u.w;
      }
    }
    int co = epi_act(p, gb, m, co_in, v, g);
    epi_store<F16>(p, gb, m, co, v);
  }
}

// Split-K reduce for GEMMs that take part in LayerNorm folding: one wave per output row, so the row owns its
// LayerNorm statistics - the consumer transform rstd*(acc - mean*s[n]) is applied to the reduced sums, and the
// producer's (sum, sumsq) of the bf16 values written goes out as ONE plane (ln_parts == 1 for the next GEMM).
template <bool F16>
__global__ __launch_bounds__(256) void splitk_reduce_rows_kernel(const ConvK p) {
  const bool pair = is_pair_act(p.act);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int qn = p.Cout / 4;
  for (int m = blockIdx.x * 4 + wv; m < p.M; m += gridDim.x * 4) {
    float mean = 0.f, rstd = 1.f;
    if (p.ln_stats) {
      float sx = 0.f, sq = 0.f;
      for (int q = 0; q < p.ln_parts; ++q) {
        const float2 t2 = *reinterpret_cast<const float2*>(p.ln_stats + 2 * ((long long)q * p.M + m));
        sx += t2.x; sq += t2.y;
      }
      mean = sx / p.ln_dim;
      rstd = rsqrtf(fmaxf(sq / p.ln_dim - mean * mean, 0.f) + p.ln_eps);
    }
    float rsx = 0.f, rsq = 0.f;
    for (int q = lane; q < qn; q += 64) {
      const int co_in = q * 4;
      if (pair && (co_in & 32)) continue;
      float v[4] = {0, 0, 0, 0}, g[4] = {0, 0, 0, 0};
      for (int s = 0; s < p.splitk; ++s) {
        const float* ws = p.ws + ((long long)s * p.M + m) * p.Cout + co_in;
        const float4 t = *reinterpret_cast<const float4*>(ws);
        v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w;
        if (pair) {
          const float4 u = *reinterpret_cast<const float4*>(ws + 32);
          g[0] += u.x; g[1] += u.y; g[2] += u.z; g[3] += u.w;
        }
      }
      if (p.ln_stats) {
        const float4 sc = *reinterpret_cast<const float4*>(p.ln_colsum + co_in);
        v[0] = rstd * (v[0] - mean * sc.x); v[1] = rstd * (v[1] - mean * sc.y);
        v[2] = rstd * (v[2] - mean * sc.z); v[3] = rstd * (v[3] - mean * sc.w);
        if (pair) {
          const float4 sg = *reinterpret_cast<const float4*>(p.ln_colsum + co_in + 32);
          g[0] = rstd * (g[0] - mean * sg.x); g[1] = rstd * (g[1] - mean * sg.y);
          g[2] = rstd * (g[2] - mean * sg.z); g[3] = rstd * (g[3] - mean * sg.w);
        }
      }
      const int co = epi_act(p, 0, m, co_in, v, g);
      if (p.yt && co >= p.n_split) {                        // transposed columns (V^T of a fused, LayerNorm-folded QKV GEMM) leave through
        epi_store<F16>(p, 0, m, co, v);                     // the transposing store - round 5: this branch was missing, and a split-K
        continue;                                           // LN-folded QKV (M <= 128 rows: B = 1-2 at the 8x8 level) never wrote V^T
      }
      if (p.res) {
        const uint2 rv = *reinterpret_cast<const uint2*>(p.res + (long long)m * p.ldr + co);
        v[0] += Act<F16>::lo(rv.x); v[1] += Act<F16>::hi(rv.x);
        v[2] += Act<F16>::lo(rv.y); v[3] += Act<F16>::hi(rv.y);
      }
      const uint2 o = make_uint2(Act<F16>::pack2(v[0], v[1]), Act<F16>::pack2(v[2], v[3]));
      *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(p.y) + (long long)m * p.ldy + co) = o;
      const float a0 = Act<F16>::lo(o.x), a1 = Act<F16>::hi(o.x);
      const float a2 = Act<F16>::lo(o.y), a3 = Act<F16>::hi(o.y);
      rsx += (a0 + a1) + (a2 + a3);
      rsq += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
    }
    if (p.row_stats) {
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) { rsx += __shfl_xor(rsx, o, 64); rsq += __shfl_xor(rsq, o, 64); }
      if (lane == 0) *reinterpret_cast<float2*>(p.row_stats + 2 * (long long)m) = make_float2(rsx, rsq);
    }
  }
}


// Split-K reduce that also leaves the GroupNorm statistics of what it writes (the 16x16 / 8x8 levels: every 3x3 conv
// there is split-K, and each used to be followed by a separate statistics pass).  Block = 16 column quads x 16 row
// lanes over 16 * RI consecutive rows of ONE image (host guarantees OHW % (16 * RI) == 0); column sums stay in registers, the
// row lanes meet in LDS and are added in row-lane order; one plain store per (column, moment) into the block's own slot of the partial plane.
template <int RI, bool F16>   // rows per block = 16 * RI (one image: host checks OHW % (16 * RI) == 0)
__global__ __launch_bounds__(256) void splitk_reduce_gn_kernel(const ConvK p) {
  __shared__ float red[16][16][9];
  const int qc = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int q = blockIdx.x * 16 + qc, qn = p.Cout / 4, co = q * 4;
  const int r0 = blockIdx.y * 16 * RI;
  float sm[4] = {0, 0, 0, 0}, sq[4] = {0, 0, 0, 0};
  if (q < qn) {
    float v[RI][4];
    uint2 rv[RI];
#pragma unroll
    for (int i = 0; i < RI; ++i) {
      const int m = r0 + rl + 16 * i;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[i][e] = 0.f;
      for (int sp = 0; sp < p.splitk; ++sp) {
        const float4 t = *reinterpret_cast<const float4*>(p.ws + ((long long)sp * p.M + m) * p.Cout + co);
        v[i][0] += t.x; v[i][1] += t.y; v[i][2] += t.z; v[i][3] += t.w;
      }
      rv[i] = p.res ? *reinterpret_cast<const uint2*>(p.res + (long long)m * p.ldr + co) : make_uint2(0, 0);
    }
    const float g[4] = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < RI; ++i) {
      const int m = r0 + rl + 16 * i;
      epi_act(p, 0, m, co, v[i], g);
      v[i][0] += Act<F16>::lo(rv[i].x); v[i][1] += Act<F16>::hi(rv[i].x);
      v[i][2] += Act<F16>::lo(rv[i].y); v[i][3] += Act<F16>::hi(rv[i].y);
      const uint2 o = make_uint2(Act<F16>::pack2(v[i][0], v[i][1]), Act<F16>::pack2(v[i][2], v[i][3]));
      store8_wt(reinterpret_cast<uint16_t*>(p.y) + (long long)m * p.ldy + co, o);
      const float a[4] = {Act<F16>::lo(o.x), Act<F16>::hi(o.x), Act<F16>::lo(o.y), Act<F16>::hi(o.y)};
#pragma unroll
      for (int e = 0; e < 4; ++e) { sm[e] += a[e]; sq[e] += a[e] * a[e]; }
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) { red[rl][qc][e] = sm[e]; red[rl][qc][4 + e] = sq[e]; }
  __syncthreads();
  // 128 (column quad, moment-lane) sums of 16 row lanes each, in row-lane order, stored into this block's own slot of the
  // partial plane [N][P][Cout][2] (P = OHW / (16 * RI); no atomics)
  if (threadIdx.x < 128) {
    const int c = threadIdx.x >> 3, e = threadIdx.x & 7;
    const int qq = blockIdx.x * 16 + c;
    if (qq < qn) {
      float a = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) a += red[r][c][e];
      const int img = r0 / p.OHW, pm = (r0 - img * p.OHW) / (16 * RI);
      float* st = p.gn_part + (((long long)img * p.gn_parts + pm) * p.Cout + qq * 4) * 2;
      st[2 * (e & 3) + (e >> 2)] = a;
    }
  }
}

// Each instantiation unit is compiled twice (-DUR_TU_F16=0 / 1): one object per 16-bit type, so that the two sets of kernels
// build in parallel.  URK(name) is the unit's exported launcher for its type; igemm.hip picks by ConvK::f16.
#ifndef UR_TU_F16
#define UR_TU_F16 0
#endif
#if UR_TU_F16
#define URK(name) name##_f16
#else
#define URK(name) name##_bf16
#endif
#define UR_F16_SWITCH(k, ...)                  \
  do {                                         \
    constexpr bool F16 = UR_TU_F16 != 0;       \
    __VA_ARGS__;                               \
  } while (0)

// reduce pass of a split-K launch: 0 = plain, 1 = row-wise (LayerNorm fusion), 2 = GroupNorm partials (16 * ri rows per block)
struct ReducePlan { int kind, ri; };
static ReducePlan plan_splitk_reduce(const ConvK& k) {
  const bool pair = k.act == UR_ACT_GEGLU || k.act == UR_ACT_GATE;
  if (k.row_stats || k.ln_stats) return {1, 0};
  static const bool no_gnred = getenv("UR_IGEMM_NOGNRED") != nullptr;
  if (!no_gnred && k.gn_part && k.y && !pair && k.staged_ok_ && k.nbatch == 1 && !k.yt && k.OHW % 64 == 0) {
    // rows per block 64 / 32 / 16: the largest that still gives >= 512 workgroups (M = 512 at the 8x8 level needs the
    // 16-row version: 160 workgroups of 64 rows left a third of the CUs idle in a latency-bound pass)
    const long long colb = (k.Cout / 4 + 15) / 16;
    if (colb * (k.M / 64) >= 512) return {2, 4};
    if (colb * (k.M / 32) >= 512) return {2, 2};
    return {2, 1};
  }
  return {0, 0};
}

// Where the GroupNorm partial plane of this launch comes from (decided BEFORE the launch so that the dry-run plan and the
// real launch agree): the conv epilogue (one partial per M tile of an image), the split-K reduce, or an extra pass over y.
static void set_gn_plan(ConvK& k, bool direct_ok, int direct_parts) {
  k.gn_fused = 0; k.gn_parts = 0;
  if (!k.gn_part) return;
  if (k.splitk == 1) {
    if (direct_ok) { k.gn_fused = 1; k.gn_parts = direct_parts; }
  } else {
    const ReducePlan r = plan_splitk_reduce(k);
    if (r.kind == 2) { k.gn_fused = 1; k.gn_parts = k.OHW / (16 * r.ri); }
  }
  if (!k.gn_fused) {
    const bool pair = k.act == UR_ACT_GEGLU || k.act == UR_ACT_GATE;
    k.gn_parts = ur::gn_stats_parts(k.N, k.OHW, (pair ? k.Cout / 2 : k.Cout) * k.nbatch);
  }
}

static void launch_splitk_reduce(ConvK& k, hipStream_t s) {
  const ReducePlan r = plan_splitk_reduce(k);
  if (r.kind == 1) {
    UR_F16_SWITCH(k, hipLaunchKernelGGL(splitk_reduce_rows_kernel<F16>, dim3(std::min((k.M + 3) / 4, 4096)), dim3(256), 0, s, k));
  } else if (r.kind == 2) {
    const long long colb = (k.Cout / 4 + 15) / 16;
    if (r.ri == 4) UR_F16_SWITCH(k, hipLaunchKernelGGL((splitk_reduce_gn_kernel<4, F16>), dim3(colb, k.M / 64), dim3(256), 0, s, k));
    else if (r.ri == 2) UR_F16_SWITCH(k, hipLaunchKernelGGL((splitk_reduce_gn_kernel<2, F16>), dim3(colb, k.M / 32), dim3(256), 0, s, k));
    else UR_F16_SWITCH(k, hipLaunchKernelGGL((splitk_reduce_gn_kernel<1, F16>), dim3(colb, k.M / 16), dim3(256), 0, s, k));
  } else {
    long long total = (long long)k.nbatch * k.M * (k.Cout / 4);
    int rb = (int)std::min<long long>((total + 255) / 256, 2048);
    UR_F16_SWITCH(k, hipLaunchKernelGGL(splitk_reduce_kernel<F16>, dim3(rb), dim3(256), 0, s, k));
  }
}

template <int BM, int BN, int WM, int WN>
int launch_cfg(ConvK& k, hipStream_t s) {
  k.tiles_m = (k.M + BM - 1) / BM;
  k.tiles_n = (k.Cout + BN - 1) / BN;
  const long long blocks = (long long)k.tiles_m * k.tiles_n * k.nbatch;
  int splitk = 1;
  if (blocks < 200 && k.nk >= 8 && k.ws && k.y) {      // (a statistics-only launch has no y for a reduce pass to write)
    long long want = (384 + blocks - 1) / blocks;
    long long cap_k = k.nk / 4;
    splitk = (int)std::min<long long>(std::min<long long>(want, cap_k), 16);
    long long need = (long long)splitk * k.nbatch * k.M * k.Cout * 4;
    while (splitk > 1 && need > (long long)k.ws_bytes_) { --splitk; need = (long long)splitk * k.nbatch * k.M * k.Cout * 4; }
    if (splitk < 1) splitk = 1;
  }
  k.splitk = splitk;
  k.nk_per_split = (k.nk + splitk - 1) / splitk;
  k.splitk = (k.nk + k.nk_per_split - 1) / k.nk_per_split;
  set_gn_plan(k, k.staged_ok_ && BN >= 32 && (k.OHW % BM) == 0, k.OHW / BM);
  if (k.dry) { k.plan_tn = k.splitk > 1 ? 1 : k.tiles_n; return UR_OK; }     // row-stat planes this launch writes
  constexpr int lds = 2 * (BM + BN) * 128;
  static_assert(epi_lds_bytes<BM, BN, 256>() <= lds, "epilogue staging must fit the K ring");
  static ur::DeviceOnce attr_once;      // the attribute is per device
  if (auto once_guard = attr_once.first()) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_kernel<BM, BN, WM, WN, false, UR_TU_F16 != 0>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_kernel<BM, BN, WM, WN, true, UR_TU_F16 != 0>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  }
  dim3 grid(k.tiles_m * k.tiles_n, k.nbatch, k.splitk);
  pick_xcd_grid(k, (double)k.Cout * k.Ktot, (double)k.N * k.H * k.W * k.Cin);
  static const bool no_g1 = getenv("UR_IGEMM_NOG1") != nullptr;
  const bool g1 = !no_g1 && k.KH == 1 && k.stride == 1 && !k.ups && k.C2 == 0 && k.pad_t == 0 && k.pad_l == 0 && k.OH == k.H && k.OW == k.W &&
                  (long long)k.M * k.ldx + k.Ktot < (1ll << 31);
  if (g1) UR_F16_SWITCH(k, hipLaunchKernelGGL((igemm_kernel<BM, BN, WM, WN, true, F16>), grid, dim3(256), lds, s, k));
  else UR_F16_SWITCH(k, hipLaunchKernelGGL((igemm_kernel<BM, BN, WM, WN, false, F16>), grid, dim3(256), lds, s, k));
  if (k.splitk > 1) launch_splitk_reduce(k, s);
  return ur::check_launch("ur_conv2d_nhwc");
}


// =====================================================================================================================
// v2 main loop: LDS-DMA (global_load_lds_dwordx4) into an NST-deep ring, counted vmcnt, one raw s_barrier per K tile.
//   * each wave-instruction lands 1 KiB = 8 rows x 128 B lane-linearly; the XOR swizzle therefore lives on the
//     SOURCE address (lane's physical slot ps reads logical chunk ps ^ ((row>>1)&7) of its row) - same 128-B line,
//     so coalescing is unchanged and the fragment reads stay conflict-free;
//   * padded / out-of-range pieces read a 16-byte zero page instead of being predicated (every wave issues the
//     same number of DMA ops per tile, so one immediate vmcnt(N) is right for all waves);
//   * no VGPR staging: the ring is NST-1 tiles ahead of the MFMAs (HBM/L2 latency hidden across barriers).
template <int BM, int BN, int WM, int WN, int NST, bool F16>
__global__ __launch_bounds__(WM* WN * 64) void igemm_glds_kernel(const ConvK p) {
  constexpr int NT = WM * WN * 64, RPP = NT / 8;                  // threads; tile rows covered by one loader pass
  constexpr int WTM = BM / WM, WTN = BN / WN, FM = WTM / 32, FN = WTN / 32;
  constexpr int XP = BM / RPP, WP = (BN + RPP - 1) / RPP, NLD = XP + WP;
  constexpr int STAGE = (BM + BN) * 128;
  static_assert(BM % RPP == 0 && WTM % 32 == 0 && WTN % 32 == 0 && RPP % 16 == 0, "tile/wave shape");
  static_assert(BN % RPP == 0 || (BN % RPP) * 2 == RPP, "partial W pass must be exactly half a pass");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid % WM, wn = wid / WM;
  const int gb = blockIdx.y, sz = blockIdx.z;
  int tm, tn;
  xcd_tile_map(p, blockIdx.x, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;

  const uint16_t* __restrict__ X1 = p.x + gb * p.bs_x;
  const uint16_t* __restrict__ X2 = p.x2 ? p.x2 + gb * p.bs_x2 : nullptr;
  const uint16_t* __restrict__ Wt = p.w + gb * p.bs_w;
  const uint16_t* zero = reinterpret_cast<const uint16_t*>(g_zero_page);

  const int lrow = tid >> 3;                                       // row inside a loader pass
  const int chunk = (tid & 7) ^ ((lrow >> 1) & 7);                 // logical 16-B K chunk this lane fetches
  int ih0[XP], iw0[XP], nb[XP];
  bool xok[XP];
#pragma unroll
  for (int i = 0; i < XP; ++i) {
    int m = m0 + i * RPP + lrow;
    xok[i] = m < p.M;
    int mm = xok[i] ? m : 0;
    int n = mm / p.OHW, rem = mm - n * p.OHW;
    int oh = rem / p.OW, ow = rem - oh * p.OW;
    ih0[i] = oh * p.stride - p.pad_t;
    iw0[i] = ow * p.stride - p.pad_l;
    nb[i] = n * p.H;
  }
  long long woff[WP];
  bool wok[WP];
  int wrow_lds[WP];
#pragma unroll
  for (int j = 0; j < WP; ++j) {
    int lr = j * RPP + lrow;
    if (lr >= BN) lr -= RPP / 2;          // half pass: the upper waves re-fetch the lower half (identical bytes)
    wrow_lds[j] = lr;
    int row = n0 + lr;
    wok[j] = row < p.Cout;
    woff[j] = (long long)(wok[j] ? row : 0) * p.ldw;
  }
  const int Hlim = p.ups ? p.H * 2 : p.H, Wlim = p.ups ? p.W * 2 : p.W;

  const int kt_begin = sz * p.nk_per_split;
  const int kt_end = min(p.nk, kt_begin + p.nk_per_split);
  int kcur = kt_begin * 64 + chunk * 8;
  int tap = kcur / p.Cin, cch = kcur - tap * p.Cin;
  const int ntap = p.KH * p.KW;
  if (p.kcm) { tap = kt_begin % ntap; cch = (kt_begin / ntap) * 64 + chunk * 8; }

  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  auto issue_tile = [&](int stage) {
    unsigned char* xs = smem + stage * STAGE;
    unsigned char* wsm = xs + BM * 128;
    const bool kval = kcur < p.Ktot;
    const int dy = (p.KW == 1) ? 0 : (tap * 11) >> 5;
    const int dx = tap - dy * p.KW;
    const uint16_t* src = X1;
    int ld = p.ldx, cc = cch;
    if (cch >= p.C1) { src = X2; ld = p.ldx2; cc = cch - p.C1; }
#pragma unroll
    for (int i = 0; i < XP; ++i) {
      int ih = ih0[i] + dy, iw = iw0[i] + dx;
      bool v = kval && xok[i] && (unsigned)ih < (unsigned)Hlim && (unsigned)iw < (unsigned)Wlim;
      if (p.ups) { ih >>= 1; iw >>= 1; }
      long long off = ((long long)(nb[i] + ih) * p.W + iw) * ld + cc;
      const uint16_t* g = v ? src + off : zero;
      __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(xs + (i * RPP + wid * 8) * 128), 16, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < WP; ++j) {
      bool v = kval && wok[j];
      const uint16_t* g = v ? Wt + woff[j] + kcur : zero;
      const int base_row = wrow_lds[j] - (lane >> 3);               // wave-uniform first row of this 1-KiB piece
      if (p.dbg & 128) __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(wsm + base_row * 128), 16, 0, 2);   // nt: stream past L1
      else __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(wsm + base_row * 128), 16, 0, 0);
    }
    kcur += 64;
    if (p.kcm) { if (++tap == ntap) { tap = 0; cch += 64; } }
    else { cch += 64; while (cch >= p.Cin) { cch -= p.Cin; ++tap; } }
  };

  f32x16 acc[FN][FM];
#pragma unroll
  for (int a = 0; a < FN; ++a)
#pragma unroll
    for (int b = 0; b < FM; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

  const int frow = lane & 31, fhalf = lane >> 5;
  // LDS byte addresses (stage 0, k-step 0) of the wave's first activation / weight fragment for the hand-scheduled K tile
  const unsigned kt_x0 = (unsigned)(uintptr_t)(lptr_t)smem + (wm * WTM + frow) * 128 + ((fhalf ^ ((frow >> 1) & 7)) << 4);
  const unsigned kt_w0 = (unsigned)(uintptr_t)(lptr_t)smem + (BM + wn * WTN + frow) * 128 + ((fhalf ^ ((frow >> 1) & 7)) << 4);
  static_assert(WTM % 32 == 0 && ((WTN >> 1) & 7) == 0, "fragment rows must keep the swizzle phase of row 0");
  auto compute = [&](int stage) {
    if constexpr (ktile_gemm_ok<FM, FN>()) {
      unsigned ab[FM];
#pragma unroll
      for (int b = 0; b < FM; ++b) ab[b] = kt_x0 + stage * STAGE + b * 4096;
      ktile_mma<FM, FN, F16>(acc, ab, kt_w0 + stage * STAGE);
      return;
    }
    const unsigned char* xs = smem + stage * STAGE;
    const unsigned char* wsm = xs + BM * 128;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int slot = ks * 2 + fhalf;
      typename Frag<F16>::type bfr[FM], afr[FN];
#pragma unroll
      for (int b = 0; b < FM; ++b) {
        int row = wm * WTM + b * 32 + frow;
        bfr[b] = *reinterpret_cast<const typename Frag<F16>::type*>(xs + row * 128 + ((slot ^ ((row >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int a = 0; a < FN; ++a) {
        int row = wn * WTN + a * 32 + frow;
        afr[a] = *reinterpret_cast<const typename Frag<F16>::type*>(wsm + row * 128 + ((slot ^ ((row >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int a = 0; a < FN; ++a)
#pragma unroll
        for (int b = 0; b < FM; ++b)
          acc[a][b] = mfma16t(afr[a], bfr[b], acc[a][b]);
    }
  };

  // ---- ring: tile t lives in stage t % NST; up to NST-1 tiles are in flight ahead of the MFMAs -----------------------
  // The wait in front of every barrier also retires this wave's LDS READS (lgkmcnt(0)): the tile issued right after the barrier
  // lands in the stage that was read in THIS iteration, and hipcc otherwise leaves the last fragment reads outstanding across
  // the barrier (it sinks the last MFMAs below it) - under LDS contention from a co-resident kernel the DMA then overwrote
  // fragments that had not been read yet (found in round 2 as run-to-run differences with the SC-Tuner side stream on).
  const int ntile = (p.dbg & 8) ? 0 : kt_end - kt_begin;
  if (ntile > 0) {
    int issued = 0;
#pragma unroll
    for (int s = 0; s < NST - 1; ++s)
      if (issued < ntile) { issue_tile(s); ++issued; }
    // wait for tile 0
    if (issued == NST - 1 && NST >= 3) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NLD * (NST - 2)) : "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int cs = 0, is = (NST - 1) % NST;
    for (int t = 0; t < ntile; ++t) {
      const bool more = issued < ntile;
      if (more) { if (!(p.dbg & 1)) issue_tile(is); ++issued; is = (is + 1 == NST) ? 0 : is + 1; }
      compute(cs);                                        // (no run-time condition around the asm K tile: hipcc spills its operands then)
      cs = (cs + 1 == NST) ? 0 : cs + 1;
      // tile t+1 must have landed before anyone reads it: all but the newest (NST-2) tiles' DMAs retired
      if (issued - (t + 1) >= NST - 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NLD * (NST - 2)) : "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  }
  if (p.dbg & 4) { if (acc[0][0][0] == 123.456f) reinterpret_cast<float*>(p.y)[0] = 1.f; return; }
  if constexpr (ktile_gemm_ok<FM, FN>()) asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");      // last asm MFMA -> VALU reads of the accumulators
  igemm_epilogue<FM, FN, WTM, WTN, BM, BN, NT, F16>(p, acc, m0, n0, wm, wn, lane, gb, sz, smem);
}

// LDS-DMA of one 1-KiB piece (64 lanes x 16 bytes -> LDS bytes [m0v, m0v + 1024)) through a raw buffer descriptor: lane address =
// base + voff + soff, and a lane whose voff is past num_records reads ZERO - the convolution's zero padding, ragged Cout rows and
// the K tail cost no select, no zero page and no 64-bit address arithmetic (the global_load_lds form needed ~10 VALU + a
// v_readfirstlane for M0 per piece: ~45 instructions per tap and wave in front of the MFMAs, 27 % of the dominant conv launch).
// Invisible to hipcc's s_waitcnt bookkeeping: the callers count vmcnt themselves (they did before, too).
constexpr unsigned IG_OOB = 0xffffff00u;       // voffset of a lane that must read zeros (>= every num_records used here)
__device__ __forceinline__ ig_u32x4 ig_make_rsrc(const void* base, unsigned long long bytes) {
  const unsigned long long a = (unsigned long long)base;
  ig_u32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((unsigned)a);
  r[1] = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffffu);         // stride 0: raw buffer
  r[2] = __builtin_amdgcn_readfirstlane((unsigned)(bytes < IG_OOB ? bytes : IG_OOB));
  r[3] = 0x00020000u;
  return r;
}
__device__ __forceinline__ void ig_lds_dma16(unsigned m0v, unsigned voff, const ig_u32x4& rs, unsigned soff) {
  // (readfirstlane: free when hipcc already knows the value is wave-uniform, and keeps an "s" operand out of a VGPR when it does not)
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(__builtin_amdgcn_readfirstlane(m0v)), "v"(voff), "s"(rs),
               "s"(__builtin_amdgcn_readfirstlane(soff)) : "memory");
}

// (the `nt` hint on these loads measured +5.2 ms per forward on the activation pieces and +12.4 ms on the weight pieces: both
// operands are re-read through L2 - halo rows by the neighbouring patches, weight tiles by every workgroup of the launch)

// Pure-GEMM specialisation of the LDS-DMA ring (1x1 / Linear, single source): row base offsets are 32-bit element
// offsets and nothing else is kept per row, which leaves room for 256 x 256 tiles (128 accumulator registers per wave).
// Large-N GEMMs (GEGLU, fused QKV) are L2->CU ingest bound: at 128 x 128 tiles every flop costs 1/64 B of ingest,
// at 256 x 256 half of that.
template <int BM, int BN, int WM, int WN, int NST, bool DIRECT, bool PAIRC, bool F16>
__global__ __launch_bounds__(WM* WN * 64) void gemm_glds_kernel(const ConvK p) {
  constexpr int NT = WM * WN * 64, RPP = NT / 8;
  constexpr int WTM = BM / WM, WTN = BN / WN, FM = WTM / 32, FN = WTN / 32;
  constexpr int XP = BM / RPP, WP = BN / RPP, NLD = XP + WP;
  constexpr int STAGE = (BM + BN) * 128;
  static_assert(BM % RPP == 0 && BN % RPP == 0 && WTM % 32 == 0 && WTN % 32 == 0 && RPP % 16 == 0, "tile/wave shape");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid % WM, wn = wid / WM;
  const int gb = blockIdx.y;
  int tm, tn;
  xcd_tile_map(p, blockIdx.x, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  const uint16_t* __restrict__ X1 = p.x + gb * p.bs_x;
  const uint16_t* __restrict__ Wt = p.w + gb * p.bs_w;
  const uint16_t* zero = reinterpret_cast<const uint16_t*>(g_zero_page);
  const int lrow = tid >> 3;
  const int chunk = (tid & 7) ^ ((lrow >> 1) & 7);
  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  int kbase = 0;
#ifdef UR_GEMM_DMA_BUILTIN                                  // A/B: the global_load_lds loader of rounds 1-2
  int xoff[XP], woff[WP];
#pragma unroll
  for (int i = 0; i < XP; ++i) { const int m = m0 + i * RPP + lrow; xoff[i] = m < p.M ? m * p.ldx + chunk * 8 : -1; }
#pragma unroll
  for (int j = 0; j < WP; ++j) { const int r = n0 + j * RPP + lrow; woff[j] = r < p.Cout ? r * p.ldw + chunk * 8 : -1; }
  auto issue_tile = [&](int stage) {
    unsigned char* xs = smem + stage * STAGE;
    unsigned char* wsm = xs + BM * 128;
    const bool kval = kbase + chunk * 8 < p.Ktot;
#pragma unroll
    for (int i = 0; i < XP; ++i) {
      const uint16_t* g = (kval && xoff[i] >= 0) ? X1 + xoff[i] + kbase : zero;
      __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(xs + (i * RPP + wid * 8) * 128), 16, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < WP; ++j) {
      const uint16_t* g = (kval && woff[j] >= 0) ? Wt + woff[j] + kbase : zero;
      __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(wsm + (j * RPP + wid * 8) * 128), 16, 0, 0);
    }
    kbase += 64;
  };
#else
  // buffer-descriptor LDS-DMA (ig_lds_dma16): rows past M / Cout read zero through the range check, the K tail through one select
  const ig_u32x4 rs_x = ig_make_rsrc(X1, (unsigned long long)p.M * p.ldx * 2);
  const ig_u32x4 rs_w = ig_make_rsrc(Wt, (unsigned long long)p.Cout * p.ldw * 2);
  const unsigned smem_lds = (unsigned)(uintptr_t)(lptr_t)smem;
  const int wid_s = __builtin_amdgcn_readfirstlane(wid);
  unsigned xvo[XP], wvo[WP];
#pragma unroll
  for (int i = 0; i < XP; ++i) { const int m = m0 + i * RPP + lrow; xvo[i] = m < p.M ? ((unsigned)m * (unsigned)p.ldx + chunk * 8u) * 2u : IG_OOB; }
#pragma unroll
  for (int j = 0; j < WP; ++j) { const int r = n0 + j * RPP + lrow; wvo[j] = r < p.Cout ? ((unsigned)r * (unsigned)p.ldw + chunk * 8u) * 2u : IG_OOB; }
  auto issue_tile = [&](int stage) {
    const unsigned xs = smem_lds + stage * STAGE, wsm = xs + BM * 128;
    const bool kval = kbase + chunk * 8 < p.Ktot;
    const unsigned so = (unsigned)kbase * 2u;
#pragma unroll
    for (int i = 0; i < XP; ++i) ig_lds_dma16(xs + (i * RPP + wid_s * 8) * 128, kval ? xvo[i] : IG_OOB, rs_x, so);
#pragma unroll
    for (int j = 0; j < WP; ++j) ig_lds_dma16(wsm + (j * RPP + wid_s * 8) * 128, kval ? wvo[j] : IG_OOB, rs_w, so);
    kbase += 64;
  };
#endif
  f32x16 acc[FN][FM];
#pragma unroll
  for (int a = 0; a < FN; ++a)
#pragma unroll
    for (int b = 0; b < FM; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
  const int frow = lane & 31, fhalf = lane >> 5;
  // LDS byte addresses (stage 0, k-step 0) of the wave's first activation / weight fragment for the hand-scheduled K tile
  const unsigned kt_x0 = (unsigned)(uintptr_t)(lptr_t)smem + (wm * WTM + frow) * 128 + ((fhalf ^ ((frow >> 1) & 7)) << 4);
  const unsigned kt_w0 = (unsigned)(uintptr_t)(lptr_t)smem + (BM + wn * WTN + frow) * 128 + ((fhalf ^ ((frow >> 1) & 7)) << 4);
  static_assert(WTM % 32 == 0 && ((WTN >> 1) & 7) == 0, "fragment rows must keep the swizzle phase of row 0");
  const int ntile = (p.dbg & 8) ? 0 : p.nk;               // (dbg bits: timing-only ablations, UR_IGEMM_DBG)
  int issued = 0;
#pragma unroll
  for (int s = 0; s < NST - 1; ++s)
    if (issued < ntile) { issue_tile(s); ++issued; }
  // (tile 0 has landed when at most the NST - 2 tiles behind it are outstanding - only if all NST - 1 were issued: a K shorter
  //  than the ring drains it instead)
  if (NST >= 3 && issued == NST - 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NLD * (NST >= 3 ? NST - 2 : 0)) : "memory");
  else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  int cs = 0, is = (NST - 1) % NST;
  for (int t = 0; t < ntile; ++t) {
    if (issued < ntile) { issue_tile(is); ++issued; is = (is + 1 == NST) ? 0 : is + 1; }
    const unsigned char* xs = smem + cs * STAGE;
    const unsigned char* wsm = xs + BM * 128;
    if constexpr (ktile_gemm_ok<FM, FN>()) {                // (no run-time condition around the asm block: hipcc spills its operands then)
      unsigned ab[FM];
#pragma unroll
      for (int b = 0; b < FM; ++b) ab[b] = kt_x0 + cs * STAGE + b * 4096;
      ktile_mma<FM, FN, F16>(acc, ab, kt_w0 + cs * STAGE);
    } else
    if (!(p.dbg & 2))
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int slot = ks * 2 + fhalf;
      typename Frag<F16>::type bfr[FM], afr[FN];
#pragma unroll
      for (int b = 0; b < FM; ++b) {
        const int row = wm * WTM + b * 32 + frow;
        bfr[b] = *reinterpret_cast<const typename Frag<F16>::type*>(xs + row * 128 + ((slot ^ ((row >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int a = 0; a < FN; ++a) {
        const int row = wn * WTN + a * 32 + frow;
        afr[a] = *reinterpret_cast<const typename Frag<F16>::type*>(wsm + row * 128 + ((slot ^ ((row >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int a = 0; a < FN; ++a)
#pragma unroll
        for (int b = 0; b < FM; ++b)
          acc[a][b] = mfma16t(afr[a], bfr[b], acc[a][b]);
    }
    cs = (cs + 1 == NST) ? 0 : cs + 1;
    if (NST >= 3 && issued - (t + 1) >= NST - 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NLD * (NST >= 3 ? NST - 2 : 0)) : "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  if (p.dbg & 4) { if (acc[0][0][0] == 123.456f) reinterpret_cast<float*>(p.y)[0] = 1.f; return; }
  if constexpr (ktile_gemm_ok<FM, FN>()) asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");      // last asm MFMA -> VALU reads of the accumulators
  igemm_epilogue<FM, FN, WTM, WTN, BM, BN, NT, F16, DIRECT, PAIRC>(p, acc, m0, n0, wm, wn, lane, gb, 0, smem);
}

template <int BM, int BN, int WM, int WN, int NST, bool DIRECT = false, bool PAIRC = false>
int launch_gemm(ConvK& k, hipStream_t s) {
  k.tiles_m = (k.M + BM - 1) / BM;
  k.tiles_n = (k.Cout + BN - 1) / BN;
  k.splitk = 1;
  k.nk_per_split = k.nk;
  set_gn_plan(k, k.staged_ok_ && (k.OHW % BM) == 0, k.OHW / BM);
  if (k.dry) { k.plan_tn = k.tiles_n; return UR_OK; }
  constexpr int lds_loop = NST * (BM + BN) * 128, lds_epi = epi_lds_bytes<BM, BN, WM * WN * 64, PAIRC>();
  constexpr int lds = lds_loop > lds_epi ? lds_loop : lds_epi;
  static_assert(lds <= 160 * 1024, "LDS budget");
  static ur::DeviceOnce attr_once;      // the attribute is per device
  if (auto once_guard = attr_once.first()) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_glds_kernel<BM, BN, WM, WN, NST, DIRECT, PAIRC, UR_TU_F16 != 0>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  }
  pick_xcd_grid(k, (double)k.Cout * k.Ktot, (double)k.M * k.Cin);
  UR_F16_SWITCH(k, hipLaunchKernelGGL((gemm_glds_kernel<BM, BN, WM, WN, NST, DIRECT, PAIRC, F16>), dim3(k.tiles_m * k.tiles_n, k.nbatch, 1), dim3(WM * WN * 64), lds, s, k));
  return ur::check_launch("ur_conv2d_nhwc");
}

template <int BM, int BN, int WM, int WN, int NST>
int launch_glds(ConvK& k, hipStream_t s, int min_blocks) {
  k.tiles_m = (k.M + BM - 1) / BM;
  k.tiles_n = (k.Cout + BN - 1) / BN;
  const long long blocks = (long long)k.tiles_m * k.tiles_n * k.nbatch;
  int splitk = 1;
  if (blocks < min_blocks && k.nk >= 8 && k.ws && k.y && !k.row_stats && !k.ln_stats) {
    long long want = (256 + blocks - 1) / blocks;
    splitk = (int)std::min<long long>(std::min<long long>(want, k.nk / 4), 16);
    while (splitk > 1 && (long long)splitk * k.nbatch * k.M * k.Cout * 4 > (long long)k.ws_bytes_) --splitk;
    if (splitk < 1) splitk = 1;
  }
  k.nk_per_split = (k.nk + splitk - 1) / splitk;
  k.splitk = (k.nk + k.nk_per_split - 1) / k.nk_per_split;
  set_gn_plan(k, k.staged_ok_ && BN >= 32 && (k.OHW % BM) == 0, k.OHW / BM);
  if (k.dry) { k.plan_tn = k.splitk > 1 ? 1 : k.tiles_n; return UR_OK; }
  constexpr int lds_loop = NST * (BM + BN) * 128, lds_epi = epi_lds_bytes<BM, BN, WM * WN * 64>();
  constexpr int lds = lds_loop > lds_epi ? lds_loop : lds_epi;
  static ur::DeviceOnce attr_once;      // the attribute is per device
  if (auto once_guard = attr_once.first()) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_glds_kernel<BM, BN, WM, WN, NST, UR_TU_F16 != 0>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  }
  dim3 grid(k.tiles_m * k.tiles_n, k.nbatch, k.splitk);
  pick_xcd_grid(k, (double)k.Cout * k.Ktot, (double)k.N * k.H * k.W * k.Cin);
  UR_F16_SWITCH(k, hipLaunchKernelGGL((igemm_glds_kernel<BM, BN, WM, WN, NST, F16>), grid, dim3(WM * WN * 64), lds, s, k));
  if (k.splitk > 1) launch_splitk_reduce(k, s);
  return ur::check_launch("ur_conv2d_nhwc");
}


// GroupNorm apply (+ SiLU) on one 16-byte piece of an LDS-resident input patch, in place: 8 channels of one pixel,
// v <- act(a[c] * v + b[c]) with (a, b) taken from the LDS copy of this chunk's affine table (fp32 a[64] | b[64], the table
// ur_groupnorm_finalize produced).  Zero-padding pieces are never touched (the convolution pads the NORMALISED tensor).
struct GnAB { float4 a0, a1, b0, b1; };
__device__ __forceinline__ GnAB gn_load_ab(const unsigned char* abuf, int chunk) {
  GnAB r;
  r.a0 = *reinterpret_cast<const float4*>(abuf + chunk * 32); r.a1 = *reinterpret_cast<const float4*>(abuf + chunk * 32 + 16);
  r.b0 = *reinterpret_cast<const float4*>(abuf + 256 + chunk * 32); r.b1 = *reinterpret_cast<const float4*>(abuf + 256 + chunk * 32 + 16);
  return r;
}
template <bool F16>
__device__ __forceinline__ void gn_piece_inplace(unsigned char* piece, const GnAB& ab, bool silu) {
  const float4 a0 = ab.a0, a1 = ab.a1, b0 = ab.b0, b1 = ab.b1;
  const uint4 v = *reinterpret_cast<const uint4*>(piece);
  float f[8];
  unpack8t<F16>(v, f);
  f[0] = fmaf(f[0], a0.x, b0.x); f[1] = fmaf(f[1], a0.y, b0.y); f[2] = fmaf(f[2], a0.z, b0.z); f[3] = fmaf(f[3], a0.w, b0.w);
  f[4] = fmaf(f[4], a1.x, b1.x); f[5] = fmaf(f[5], a1.y, b1.y); f[6] = fmaf(f[6], a1.z, b1.z); f[7] = fmaf(f[7], a1.w, b1.w);
  if (silu) {
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = silu_f(f[e]);
  }
  *reinterpret_cast<uint4*>(piece) = pack8t<F16>(f);
}

// Tap-crossing software pipeline of the halo convs (igemm_asm.inc, ktile_pipe in tools/gen_igemm_asm.py): the K tile's MFMAs are two
// asm blocks (PHASE 1 = first half, 2 = second half + the first fragment reads of the NEXT tile, 3 = second half of the last tile)
// with the workgroup's "next weight tile landed" wait + barrier between them; PHASE 0 issues the first tile's prefetch.  The
// fragment rings ra / rb live across the blocks.  Without it the matrix pipe drains at every per-tile barrier: the first MFMA
// behind the barrier waits a full LDS round trip (~150-250 of ~1300 cycles per tile, tools/probe + UR_IGASM_ABL timings).
template <int FM, int FN> struct KPipeRings { ig_u32x4 ra[FN == 5 ? 10 : 4], rb[4]; };      // (ring sizes of tools/gen_igemm_asm.py)
template <int FM, int FN, bool F16, int PHASE>
__device__ __forceinline__ void kpipe(f32x16 (&acc)[FN][FM], KPipeRings<FM, FN>& r, const unsigned (&ab)[FM], unsigned aw, const unsigned (&abn)[FM],
                                      unsigned awn) {
  unsigned x0, x1, xw;
#define KP_OPS_1x5 : "+v"(acc[0][0]), "+v"(acc[1][0]), "+v"(acc[2][0]), "+v"(acc[3][0]), "+v"(acc[4][0]),                                        \
      "+v"(r.ra[0]), "+v"(r.ra[1]), "+v"(r.ra[2]), "+v"(r.ra[3]), "+v"(r.ra[4]), "+v"(r.ra[5]), "+v"(r.ra[6]), "+v"(r.ra[7]), "+v"(r.ra[8]), "+v"(r.ra[9]), \
      "+v"(r.rb[0]), "+v"(r.rb[1]), "+v"(r.rb[2]), "+v"(r.rb[3]), "=&v"(x0), "=&v"(xw) : "v"(ab[0]), "v"(aw), "v"(abn[0]), "v"(awn) : "memory"
#define KP_OPS_2x2 : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]),                                                          \
      "+v"(r.ra[0]), "+v"(r.ra[1]), "+v"(r.ra[2]), "+v"(r.ra[3]),                                                                                 \
      "+v"(r.rb[0]), "+v"(r.rb[1]), "+v"(r.rb[2]), "+v"(r.rb[3]), "=&v"(x0), "=&v"(x1), "=&v"(xw)                                                   \
      : "v"(ab[0]), "v"(ab[1]), "v"(aw), "v"(abn[0]), "v"(abn[1]), "v"(awn) : "memory"
  if constexpr (FM == 1 && FN == 5) {
    if constexpr (PHASE == 0) IG_MN(IG_ASM_KP_1x5_PRE, KP_OPS_1x5);
    else if constexpr (PHASE == 1) IG_MN(IG_ASM_KP_1x5_H1, KP_OPS_1x5);
    else if constexpr (PHASE == 2) IG_MN(IG_ASM_KP_1x5_H2, KP_OPS_1x5);
    else IG_MN(IG_ASM_KP_1x5_H2L, KP_OPS_1x5);
  } else if constexpr (FM == 2 && FN == 2) {
    if constexpr (PHASE == 0) IG_MN(IG_ASM_KP_2x2_PRE, KP_OPS_2x2);
    else if constexpr (PHASE == 1) IG_MN(IG_ASM_KP_2x2_H1, KP_OPS_2x2);
    else if constexpr (PHASE == 2) IG_MN(IG_ASM_KP_2x2_H2, KP_OPS_2x2);
    else IG_MN(IG_ASM_KP_2x2_H2L, KP_OPS_2x2);
  } else {
    static_assert(FM == 0, "no pipelined K tile for this fragment shape");
  }
#undef KP_OPS_1x5
#undef KP_OPS_2x2
}
template <int FM, int FN> __device__ __forceinline__ void kpipe_drain(KPipeRings<FM, FN>& r) {
  if constexpr (FN == 5)
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r.ra[0]), "+v"(r.ra[1]), "+v"(r.ra[2]), "+v"(r.ra[3]), "+v"(r.ra[4]), "+v"(r.ra[5]), "+v"(r.ra[6]), "+v"(r.ra[7]),
                 "+v"(r.ra[8]), "+v"(r.ra[9]), "+v"(r.rb[0]), "+v"(r.rb[1]), "+v"(r.rb[2]), "+v"(r.rb[3]) : : "memory");
  else
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r.ra[0]), "+v"(r.ra[1]), "+v"(r.ra[2]), "+v"(r.ra[3]), "+v"(r.rb[0]), "+v"(r.rb[1]), "+v"(r.rb[2]), "+v"(r.rb[3]) : : "memory");
}
template <int FM, int FN> __device__ __forceinline__ void kpipe_init(KPipeRings<FM, FN>& r) {
#pragma unroll
  for (int i = 0; i < (FN == 5 ? 10 : 4); ++i) r.ra[i] = ig_u32x4{0u, 0u, 0u, 0u};
#pragma unroll
  for (int i = 0; i < 4; ++i) r.rb[i] = ig_u32x4{0u, 0u, 0u, 0u};
}

// =====================================================================================================================
// Halo-tile 3x3 convolution (stride 1, pad 1, optional nearest-2x upsampled input).
// The L2 -> CU ingest path, not the MFMA pipe, bounds the implicit GEMM (about 13 B/clk/CU: ~8 KB in flight against
// ~600 cycles of L2 latency; LDS-DMA does not allocate in L1), and the tap-major loaders re-fetch every input pixel
// 9 times.  Here a workgroup owns an 8 x 32 pixel output patch of ONE image: per 64-channel chunk the (8+2) x (32+2)
// input patch is DMA'd into LDS once (double-buffered across chunks, spread over the first six taps of the previous
// chunk) and the nine taps read their B fragments out of it at a constant row offset; only the weight tile streams per
// K tile (3-stage ring).  Ingest per K tile drops from (256+BN)*128 B to ~BN*128 B + 5 KB.
// A fragment is one 32-pixel patch row, so its LDS rows are consecutive for every tap and the (row>>1)&7 slot swizzle
// stays conflict-free (tools/lds_conflicts.py model; a 16x16 patch would be 2-way conflicted on every tap).
template <int TH, int BN, int WM, int WN, bool F16, bool GNP>
__global__ __launch_bounds__(WM* WN * 64) void igemm_halo_kernel(const ConvK p) {
  constexpr int NW = WM * WN, TW = 32, BM = TH * TW, PW = TW + 2, HPIX = (TH + 2) * PW;   // 340 (TH=8) / 204 (TH=4) halo pixels
  constexpr int HSLOTS = (HPIX + 8 * NW - 1) / (8 * NW);                              // tap slots that carry a halo piece
  constexpr int HPIECES = HSLOTS * NW, HBYTES = HPIECES * 1024;   // surplus pieces are never read (zero fill)
  // (ring slot = at least one piece per wave: in the thin tile, BN = 32, waves >= BN / 8 issue an all-zero piece behind the tile's rows)
  constexpr int WBYTES = BN * 128 > NW * 1024 ? BN * 128 : NW * 1024, WPIECES = BN / 8, WPW = (WPIECES + NW - 1) / NW;   // weight pieces per wave per tile
  constexpr int WTM = BM / WM, WTN = BN / WN, FM = WTM / 32, FN = WTN / 32, NT = NW * 64;
  static_assert((NW == 8 || NW == 4) && WTM % 32 == 0 && WTN % 32 == 0 && HSLOTS <= 8, "wave layout");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const hbuf = smem;                      // 2 halo patches
  unsigned char* const wring = smem + 2 * HBYTES;        // 3 weight tiles
  unsigned char* const abuf = wring + 3 * WBYTES;        // 2 x 1 KiB: GroupNorm affine (a[64] | b[64]) of the current / next chunk
  // GroupNorm apply (+ SiLU) of the input fused into the loader (ur_conv_desc.gn_ab): every wave rewrites the halo pieces IT
  // fetched, in LDS, two taps after issuing them (its own vmcnt covers them) - the normalised tensor never exists in HBM.
  // (GNP is a compile-time variant: with a run-time flag the plain kernel's tap loop scheduled 20 % slower)
  constexpr bool gnp = GNP;
  static_assert(HSLOTS <= 7, "the in-LDS GroupNorm pass needs taps 2 .. HSLOTS+1 <= 8");

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid % WM, wn = wid / WM;
  const int sz = blockIdx.y;
  int id = blockIdx.x;
  {
    const int nt = p.tiles_m * p.tiles_n, q = nt >> 3, r = nt & 7, xcd = id & 7, idx = id >> 3;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tn = id % p.tiles_n, tmi = id / p.tiles_n;
  const int tiles_x = p.OW / TW, tiles_y = p.OH / TH;
  const int tx = tmi % tiles_x, ty = (tmi / tiles_x) % tiles_y, img = tmi / (tiles_x * tiles_y);
  const int n0 = tn * BN;
  const int m0 = img * p.OHW + ty * TH * p.OW + tx * TW;         // first output pixel of the patch

  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  const uint16_t* __restrict__ X1 = p.x;
  const uint16_t* __restrict__ X2 = p.x2;
  const uint16_t* __restrict__ Wt = p.w;
  const uint16_t* zero = reinterpret_cast<const uint16_t*>(g_zero_page);
  const int lr = lane >> 3, ps = lane & 7;
  // Every piece this wave issues has index == wid (mod NW, NW even), so its rows share one parity pattern and the lane's logical
  // 16-byte K chunk is the same for all of them: physical slot ps holds chunk ps ^ ((row>>1)&7), row = 8*piece + lr.
  const int chunk = ps ^ ((((wid & 1) << 2) + (lr >> 1)) & 7);

  // input-pixel index of this lane for the six halo pieces (tap slots 0..5) this wave issues per chunk; -1 = zero fill
  int hpix[HSLOTS];
#pragma unroll
  for (int t = 0; t < HSLOTS; ++t) {
    const int hr = (t * NW + wid) * 8 + lr;
    const int hy = hr / PW, hx = hr - hy * PW;
    int iy = ty * TH - 1 + hy, ix = tx * TW - 1 + hx;               // coordinates in the (possibly upsampled) input
    const bool v = hr < HPIX && (unsigned)iy < (unsigned)p.OH && (unsigned)ix < (unsigned)p.OW;
    if (p.ups) { iy >>= 1; ix >>= 1; }
    hpix[t] = v ? (img * p.H + iy) * p.W + ix : -1;
  }
  int woff[WPW];
#pragma unroll
  for (int i = 0; i < WPW; ++i) {
    const int qq = (wid + NW * i < WPIECES) ? wid + NW * i : wid;     // surplus slot: repeat this wave's first piece
    const int row = n0 + qq * 8 + lr;
    woff[i] = row < p.Cout ? row * p.ldw : -1;
  }
  // blockIdx.y splits the channel-chunk range (few-tile layers: tiles * splits <= one round of CUs)
#if UR_HALO_ABL == 3                                          // timing-only: prologue + epilogue, no K loop
  const int cps = p.nk_per_split / 9, c_begin = sz * cps, nchunk = p.N < 0 ? 1 : 0;
#else
  const int cps = p.nk_per_split / 9, c_begin = sz * cps, nchunk = min(p.nk / 9, c_begin + cps) - c_begin;
#endif
  const int nk = nchunk * 9, kt0 = c_begin * 9;

#ifdef UR_HALO_DMA_BUILTIN                                  // A/B: the global_load_lds loaders of rounds 1-2
  auto issue_w = [&](int kt, int ring) {
    unsigned char* st = wring + ring * WBYTES;
#pragma unroll
    for (int i = 0; i < WPW; ++i) {
      const int qq = (wid + NW * i < WPIECES) ? wid + NW * i : wid;
      const uint16_t* g = woff[i] >= 0 ? Wt + woff[i] + (kt0 + kt) * 64 + chunk * 8 : zero;
      __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(st + qq * 1024), 16, 0, 0);
    }
  };
  auto issue_h = [&](int c, int t) {                                // halo piece (t*8 + wid) of chunk c
    const int q = t * NW + wid;
    int cc = (c_begin + c) * 64 + chunk * 8;
    const uint16_t* src = X1;
    int ld = p.ldx;
    if (cc >= p.C1) { src = X2; ld = p.ldx2; cc -= p.C1; }
    const uint16_t* g = hpix[t] >= 0 ? src + hpix[t] * ld + cc : zero;
    __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(hbuf + (c & 1) * HBYTES + q * 1024), 16, 0, 0);
  };
#else
  // buffer-descriptor LDS-DMA (ig_lds_dma16): per piece one scalar M0 value, one lane offset kept in a register, one scalar K offset
  const int wid_s = __builtin_amdgcn_readfirstlane(wid);
  const ig_u32x4 rs_w = ig_make_rsrc(Wt, (unsigned long long)p.Cout * p.ldw * 2);
  const ig_u32x4 rs_x1 = ig_make_rsrc(X1, (unsigned long long)p.N * p.H * p.W * p.ldx * 2);
  const ig_u32x4 rs_x2 = ig_make_rsrc(X2 ? X2 : X1, (unsigned long long)p.N * p.H * p.W * (X2 ? p.ldx2 : p.ldx) * 2);
  const unsigned wring_lds = (unsigned)(uintptr_t)(lptr_t)wring, hbuf_lds = (unsigned)(uintptr_t)(lptr_t)hbuf;
  unsigned wvo[WPW];
#pragma unroll
  for (int i = 0; i < WPW; ++i) wvo[i] = woff[i] >= 0 ? (unsigned)woff[i] * 2u + chunk * 16u : IG_OOB;
  auto issue_w = [&](int kt, int ring) {
#pragma unroll
    for (int i = 0; i < WPW; ++i) {
      const int qq = (wid_s + NW * i < WPIECES) ? wid_s + NW * i : wid_s;
      ig_lds_dma16(wring_lds + ring * WBYTES + qq * 1024, wvo[i], rs_w, (unsigned)(kt0 + kt) * 128u);
    }
  };
  auto issue_h = [&](int c, int t) {                                // halo piece (t*8 + wid) of chunk c
    const int cb = (c_begin + c) * 64;                              // first channel of the chunk: decides the source (C1 % 64 == 0)
    const bool second = cb >= p.C1;
    const unsigned ld2 = (unsigned)(second ? p.ldx2 : p.ldx) * 2u;
    const unsigned vo = hpix[t] >= 0 ? (unsigned)hpix[t] * ld2 + chunk * 16u : IG_OOB;
    const unsigned m0v = hbuf_lds + (c & 1) * HBYTES + (t * NW + wid_s) * 1024;
    if (second) ig_lds_dma16(m0v, vo, rs_x2, (unsigned)(cb - p.C1) * 2u);
    else ig_lds_dma16(m0v, vo, rs_x1, (unsigned)cb * 2u);
  };
#endif
  // affine table of chunk c -> abuf[c & 1]: lanes 0-15 fetch a[64], lanes 16-31 b[64] (fp32), the rest a zero page.  Every
  // wave issues the same piece to the same place (identical bytes), so each may read it back after its OWN vmcnt and the
  // per-iteration DMA count stays wave-uniform.
  auto issue_ab = [&](int c) {
    const float* t = p.gn_ab + ((long long)img * 2 + (lane >> 4 & 1)) * p.Cin + (c_begin + c) * 64 + (lane & 15) * 4;
    const void* g = lane < 32 ? (const void*)t : (const void*)zero;
    __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(abuf + (c & 1) * 1024), 16, 0, 0);
  };
  auto gn_slot = [&](int c, int t) {                      // this wave's halo piece t of chunk c, in place
    if (hpix[t] >= 0)
      gn_piece_inplace<F16>(hbuf + (c & 1) * HBYTES + (t * NW + wid) * 1024 + lane * 16, gn_load_ab(abuf + (c & 1) * 1024, chunk), p.gn_silu != 0);
  };

  f32x16 acc[FN][FM];
#pragma unroll
  for (int a = 0; a < FN; ++a)
#pragma unroll
    for (int b = 0; b < FM; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
  const int frow = lane & 31, fhalf = lane >> 5;
  // LDS byte addresses for the hand-scheduled tap body: halo buffer 0, and this wave's first weight fragment at k-step 0 in ring slot 0
  const unsigned hb_lds = (unsigned)(uintptr_t)(lptr_t)hbuf;
  const unsigned aw_lds = (unsigned)(uintptr_t)(lptr_t)wring + (wn * WTN + frow) * 128 + ((fhalf ^ ((frow >> 1) & 7)) << 4);
  static_assert(((WTN >> 1) & 7) == 0, "the wave's weight rows must keep the swizzle phase of fragment row 0");

  // ---- prologue: whole halo of chunk 0, weight tiles 0 and 1 ------------------------------------------------------------
  if (gnp) issue_ab(0);
#pragma unroll
  for (int t = 0; t < HSLOTS; ++t) issue_h(0, t);
  issue_w(0, 0);
  if (nk > 1) {
    issue_w(1, 1);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WPW) : "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  if (gnp) {                                              // chunk 0's patch is normalised before anyone reads it
#pragma unroll
    for (int t = 0; t < HSLOTS; ++t) gn_slot(0, t);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();

  // LDS addresses (k-step 0) of the fragments tap tp of chunk cc reads: activation rows shifted by (dy, dx), weight ring slot tp % 3
  auto tap_addr = [&](int cc, int tp, unsigned (&ab)[FM]) -> unsigned {
    const int dy = tp / 3, dx = tp % 3;
#pragma unroll
    for (int b = 0; b < FM; ++b) {
      const int hrow = (wm * FM + b + dy) * PW + dx + frow;
      ab[b] = hb_lds + (cc & 1) * HBYTES + hrow * 128 + ((fhalf ^ ((hrow >> 1) & 7)) << 4);
    }
    return aw_lds + (tp % 3) * WBYTES;
  };
#if defined(UR_HALO_DMA_BUILTIN) || !defined(UR_HALO_STAGGER)     // (staggering measured no gain: DESIGN.md 6c)
  constexpr bool stagger = false;
  const bool early_grp = true;
#else
  constexpr bool stagger = ktile_asm_ok<FM, FN>();
  const bool early_grp = wid_s < NW / 2;                   // (waves w and w + NW/2 share a SIMD)
#endif
#if UR_HALO_ABL == 6
  unsigned long long ts[63];
#pragma unroll
  for (int i = 0; i < 63; ++i) ts[i] = 0;
#endif
  KPipeRings<FM, FN> rings;
  if constexpr (ktile_asm_ok<FM, FN>()) {
    kpipe_init(rings);
    unsigned ab0[FM];
    const unsigned aw0 = tap_addr(0, 0, ab0);
    kpipe<FM, FN, F16, 0>(acc, rings, ab0, aw0, ab0, aw0);
  }

  // K loop as chunks x 9 UNROLLED taps: the tap decides everything that used to be run-time control in the loop body (which
  // halo piece to prefetch, the fragment row offset, the weight-ring slot = tap % 3 because 9 % 3 == 0), so each tap's body
  // is straight-line code with immediate LDS offsets instead of a branch ladder in front of every 20 MFMAs.
  for (int c = 0; c < nchunk; ++c) {
    const unsigned char* hb = hbuf + (c & 1) * HBYTES;
    const bool next_chunk = c + 1 < nchunk;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int kt = c * 9 + tap;
      const bool more_w = kt + 2 < nk;
      const bool more_h = tap < HSLOTS && next_chunk;
      // The DMA path of a CU takes ~24 cycles per 1-KiB piece and a wave sits in its buffer_load until the queue has room: with all
      // eight waves issuing at the top of the tap nobody feeds the matrix pipe meanwhile (0.37 us of a 1.2 us tap, UR_HALO_ABL=5).
      // So the two waves of a SIMD issue half a tap apart: waves 0..NW/2-1 here, the others between the two MFMA halves.
      auto issue_tap = [&]() {
#if UR_HALO_ABL != 5                                        // (5: timing-only, no DMA in the loop at all)
        if (more_w) issue_w(kt + 2, (tap + 2) % 3);
        if (tap < HSLOTS) { if (next_chunk) issue_h(c + 1, tap < HSLOTS ? tap : 0); }
#endif
      };
#if UR_HALO_ABL == 6                                        // in-kernel phase timers (shader cycles) of chunk 1, workgroup 0, waves 0 and NW/2
#define IG_TS(i) do { if (c == 1) ts[tap * 7 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define IG_TS(i) do { } while (0)
#endif
      IG_TS(0);
      if (!stagger || early_grp) issue_tap();
      IG_TS(1);
      const bool ab_now = gnp && tap == 0 && next_chunk;    // (+1 DMA op in this iteration, counted in the wait below)
      if (ab_now) issue_ab(c + 1);
      // piece (tap - 2) of the next chunk landed with the previous iteration's wait: normalise it in LDS next to this tap's MFMAs
      // (splitting the two waves of a SIMD to opposite sides of the MFMA block measured no gain and cost registers)
      if (gnp && tap >= 2 && tap - 2 < HSLOTS && next_chunk) gn_slot(c + 1, (tap >= 2 && tap - 2 < HSLOTS) ? tap - 2 : 0);
      // everything issued in EARLIER iterations has landed once only this iteration's pieces may still be in flight
      auto dma_wait = [&]() {
#if UR_HALO_ABL != 1 && UR_HALO_ABL != 5                    // (1: timing-only, never wait for the DMA)
        if (ab_now) {                                       // tap 0 with the next chunk's affine table in flight as well
          if (more_w) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WPW + 2) : "memory");
          else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        } else if (more_w && more_h) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WPW + 1) : "memory");
        else if (more_w) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WPW) : "memory");
        else if (more_h) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
      };
      if constexpr (ktile_asm_ok<FM, FN>()) {
        // hand-scheduled, tap-crossing pipeline (kpipe above).  Barrier 1 (mid-tap) publishes weight tile kt + 1 and every halo /
        // GroupNorm piece of earlier taps - the second half prefetches the next tap's first fragments behind it; barrier 2 (end)
        // tells everyone this tap's weight tile has been read: its ring slot is the next tap's DMA target.  This tap's own reads
        // are retired by then (its MFMAs consumed them); the in-place GroupNorm writes are older than those reads (LDS is in-order).
        unsigned ab[FM], abn[FM];
        const unsigned aw = tap_addr(c, tap, ab), awn = tap_addr(c + (tap == 8 ? 1 : 0), (tap + 1) % 9, abn);
        kpipe<FM, FN, F16, 1>(acc, rings, ab, aw, abn, awn);
        IG_TS(2);
        if (stagger && !early_grp) issue_tap();
        dma_wait();
        IG_TS(3);
        __builtin_amdgcn_s_barrier();
        IG_TS(4);
        kpipe<FM, FN, F16, 2>(acc, rings, ab, aw, abn, awn);      // (the last tap prefetches too - valid LDS, never used: one code path)
        IG_TS(5);
        __builtin_amdgcn_s_barrier();
        IG_TS(6);
      } else {
        const int dy = tap / 3, dx = tap % 3;
        const unsigned char* wsm = wring + (tap % 3) * WBYTES;
        typedef typename Frag<F16>::type frag_t;
        auto load_frags = [&](int ks, frag_t (&bfr)[FM], frag_t (&afr)[FN]) {
          const int slot = ks * 2 + fhalf;
#pragma unroll
          for (int b = 0; b < FM; ++b) {
            const int hrow = (wm * FM + b + dy) * PW + dx + frow;
            bfr[b] = *reinterpret_cast<const frag_t*>(hb + hrow * 128 + ((slot ^ ((hrow >> 1) & 7)) << 4));
          }
#pragma unroll
          for (int a = 0; a < FN; ++a) {
            const int row = wn * WTN + a * 32 + frow;
            afr[a] = *reinterpret_cast<const frag_t*>(wsm + row * 128 + ((slot ^ ((row >> 1) & 7)) << 4));
          }
        };
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          frag_t bfr[FM], afr[FN];
          load_frags(ks, bfr, afr);
#pragma unroll
          for (int a = 0; a < FN; ++a)
#pragma unroll
            for (int b = 0; b < FM; ++b)
              acc[a][b] = mfma16t(afr[a], bfr[b], acc[a][b]);
        }
        dma_wait();
        // lgkmcnt(0): (1) this wave's fragment reads of the ring slot / halo buffer that the NEXT tap's DMA overwrites are retired
        // before anyone passes the barrier (hipcc leaves the last reads outstanding across it otherwise - see igemm_glds_kernel);
        // (2) the in-place GroupNorm writes are published by the barrier
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
    }
  }
  if constexpr (ktile_asm_ok<FM, FN>()) kpipe_drain(rings);      // the last tap's prefetch lands in the rings: they stay reserved until here
  asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 7" ::: "memory");      // last MFMA of the asm tap body -> VALU reads of the accumulators
#if UR_HALO_ABL == 6                                          // phase timers -> the first bytes of y (no epilogue)
  if (acc[0][0][0] == 123.456f) reinterpret_cast<uint16_t*>(p.y)[0] = 1;
  if (blockIdx.x == 0 && blockIdx.y == 0 && lane == 0 && (wid == 0 || wid == NW / 2)) {
    unsigned long long* o = reinterpret_cast<unsigned long long*>(p.y) + (wid ? 64 : 0);
#pragma unroll
    for (int i = 0; i < 63; ++i) o[i] = ts[i];
  }
#elif UR_HALO_ABL == 4                                        // timing-only: no epilogue (one store keeps the loop alive)
  if (acc[0][0][0] == 123.456f) reinterpret_cast<uint16_t*>(p.y)[0] = 1;
#else
  // (BN == 32: the thin tile of the conv_out layers - fp32 output, 4-8 channels - leaves through the direct stores)
  igemm_epilogue<FM, FN, WTM, WTN, BM, BN, NT, F16, BN == 32>(p, acc, m0, n0, wm, wn, lane, 0, sz, smem);
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// Wave-specialised form of igemm_halo_kernel: NW compute waves + loader waves (two since round 6: one for the weight tiles, one for
// the patch pieces and affine tables; launched with one - UR_HALO_LD1 - the single loader does both, the rounds-3-5 structure).
// Phase timers in the kernel above (profiles/r3_halo_phase_timers.txt) show where the matrix pipe idles: a CU's DMA path takes
// ~25 cycles per 1-KiB piece and a wave sits in its buffer_load until the queue accepts it, so with every wave issuing its
// share at the top of a tap the first MFMA starts 400-660 cycles late (of ~2100), whatever the order or the instruction count.
// Here the compute waves never execute a memory instruction inside the K loop: the loader wave issues all pieces (half of the
// weight tile + the tap's halo pieces before the mid-tap barrier, the other half behind it), waits with a counted vmcnt for the
// previous tap's pieces and takes part in the two barriers, which publish them.  It always issues the same number of pieces per
// tap position (tile / chunk indices clamped at the end: a re-fetch into a slot nobody reads), so its counts are immediates.
template <int TH, int BN, int WM, int WN, bool F16, bool GNP>
__global__ __launch_bounds__((WM * WN + 2) * 64) void igemm_halo_ws_kernel(const ConvK p) {
  constexpr int NW = WM * WN, TW = 32, BM = TH * TW, PW = TW + 2, HPIX = (TH + 2) * PW;
  constexpr int HSLOTS = (HPIX + 8 * NW - 1) / (8 * NW), HNEED = (HPIX + 7) / 8;          // (LDS layout of the plain kernel); pieces actually read
  constexpr int HPIECES = HSLOTS * NW, HBYTES = HPIECES * 1024;
  // The loader spreads a chunk's HNEED patch pieces over the 16 half-taps of taps 0..7 of the previous chunk (HPER per half-tap:
  // a patch piece costs the loader ~85 cycles, a weight piece ~38 - profiles/r3_halo_phase_timers.txt - so each half-tap carries
  // about as much issue time as its 10 MFMAs per wave take); tap 8 carries none (its pieces would land after the prefetch).
  constexpr int HPER = (HNEED + 15) / 16, HTAP = 2 * HPER;
  constexpr int WBYTES = BN * 128, WPIECES = BN / 8, WH = WPIECES / 2;
  constexpr int WTM = BM / WM, WTN = BN / WN, FM = WTM / 32, FN = WTN / 32, NT = NW * 64;
  static_assert(NW == 8 && WTM % 32 == 0 && WTN % 32 == 0 && HTAP <= NW && ktile_asm_ok<FM, FN>(), "wave layout");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const hbuf = smem;                      // 2 halo patches
  unsigned char* const wring = smem + 2 * HBYTES;        // 3 weight tiles
  unsigned char* const abuf = wring + 3 * WBYTES;        // 2 x 1 KiB: GroupNorm affine (a[64] | b[64]) of the current / next chunk
  constexpr bool gnp = GNP;
  typedef __attribute__((address_space(3))) void* lptr_t;

#if UR_HALO_ABL == 7                                          // workgroup time line in 100 MHz ticks (A/B build): start, K loop, end
  const unsigned long long wg_t0 = __builtin_amdgcn_s_memrealtime();
#endif
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wid_s = __builtin_amdgcn_readfirstlane(wid);
  const int sz = blockIdx.y;
  int id = blockIdx.x;
  {
    const int nt = p.tiles_m * p.tiles_n, q = nt >> 3, r = nt & 7, xcd = id & 7, idx = id >> 3;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tn = id % p.tiles_n, tmi = id / p.tiles_n;
  const int tiles_x = p.OW / TW, tiles_y = p.OH / TH;
  const int tx = tmi % tiles_x, ty = (tmi / tiles_x) % tiles_y, img = tmi / (tiles_x * tiles_y);
  const int n0 = tn * BN;
  const int m0 = img * p.OHW + ty * TH * p.OW + tx * TW;         // first output pixel of the patch
  const int cps = p.nk_per_split / 9, c_begin = sz * cps, nchunk = min(p.nk / 9, c_begin + cps) - c_begin;
  const int nk = nchunk * 9, kt0 = c_begin * 9;
  const unsigned wring_lds = (unsigned)(uintptr_t)(lptr_t)wring, hbuf_lds = (unsigned)(uintptr_t)(lptr_t)hbuf;
  const unsigned abuf_lds = (unsigned)(uintptr_t)(lptr_t)abuf;
  const int lr = lane >> 3, ps = lane & 7;
  // halo row hr of the patch -> this lane's input pixel (or -1: zero padding / surplus row)
  auto halo_pixel = [&](int hr) -> int {
    const int hy = hr / PW, hx = hr - hy * PW;
    int iy = ty * TH - 1 + hy, ix = tx * TW - 1 + hx;               // coordinates in the (possibly upsampled) input
    const bool v = hr < HPIX && (unsigned)iy < (unsigned)p.OH && (unsigned)ix < (unsigned)p.OW;
    if (p.ups) { iy >>= 1; ix >>= 1; }
    return v ? (img * p.H + iy) * p.W + ix : -1;
  };

  if (wid_s >= NW) {
    // ================================================= loader wave(s) ================================================
    // One loader wave (576-thread launch) issues the weight tiles AND the patch pieces; with a second one (640-thread launch, round 6)
    // wave NW issues the weight tiles and wave NW + 1 the patch pieces (+ the affine tables): a lone loader is issue-bound on the taps
    // that carry patch pieces (16-20 weight pieces at ~38 cycles + 6 patch pieces at ~85 against 1 280 MFMA cycles), two waves feed the
    // CU's DMA path in parallel.  Each wave waits for its own pieces; both take part in both barriers of every tap.
    const bool two = blockDim.x > (NW + 1) * 64;
    const bool do_w = wid_s == NW, do_h = two ? wid_s == NW + 1 : true;
    // piece q = LDS rows 8q .. 8q+7; the lane's physical 16-byte slot ps holds logical K chunk ps ^ ((row >> 1) & 7): two parities
    const unsigned ck0 = (ps ^ ((lr >> 1) & 7)) * 16u, ck1 = (ps ^ ((4 + (lr >> 1)) & 7)) * 16u;
    const ig_u32x4 rs_w = ig_make_rsrc(p.w, (unsigned long long)p.Cout * p.ldw * 2);
    const ig_u32x4 rs_x1 = ig_make_rsrc(p.x, (unsigned long long)p.N * p.H * p.W * p.ldx * 2);
    const ig_u32x4 rs_x2 = ig_make_rsrc(p.x2 ? p.x2 : p.x, (unsigned long long)p.N * p.H * p.W * (p.x2 ? p.ldx2 : p.ldx) * 2);
    const ig_u32x4 rs_ab = ig_make_rsrc(gnp ? (const void*)p.gn_ab : (const void*)p.w, gnp ? (unsigned long long)p.N * 2 * p.Cin * 4 : 16ull);
    unsigned wvo[WPIECES];
#pragma unroll
    for (int q = 0; q < WPIECES; ++q) {
      const int row = n0 + q * 8 + lr;
      wvo[q] = row < p.Cout ? (unsigned)row * (unsigned)p.ldw * 2u + ((q & 1) ? ck1 : ck0) : IG_OOB;
    }
    int hpx[HNEED];
    auto dma_w = [&](int kt, int ring, int q0, int q1) {            // pieces [q0, q1) of weight tile kt (clamped) -> ring slot
      const unsigned so = (unsigned)(kt0 + min(kt, nk - 1)) * 128u;
#pragma unroll
      for (int q = q0; q < q1; ++q) ig_lds_dma16(wring_lds + ring * WBYTES + q * 1024, wvo[q], rs_w, so);
    };
    auto dma_h = [&](int c, int h) {                                // patch pieces of half-tap h (0..15), chunk c (clamped) -> halo buffer c & 1
      const int cb = (c_begin + min(c, nchunk - 1)) * 64;           // first channel of the chunk: decides the source (C1 % 64 == 0)
      const bool second = cb >= p.C1;
      const unsigned ld2 = (unsigned)(second ? p.ldx2 : p.ldx) * 2u;
      const unsigned so = (unsigned)(second ? cb - p.C1 : cb) * 2u;
#pragma unroll
      for (int j = 0; j < HPER; ++j) {
        const int q = h * HPER + j;
        if (h < 16 && q < HNEED) {
          const unsigned vo = hpx[q] >= 0 ? (unsigned)hpx[q] * ld2 + ((q & 1) ? ck1 : ck0) : IG_OOB;
          const unsigned m0v = hbuf_lds + (c & 1) * HBYTES + q * 1024;
          if (second) ig_lds_dma16(m0v, vo, rs_x2, so);
          else ig_lds_dma16(m0v, vo, rs_x1, so);
        }
      }
    };
    auto halo_count = [](int h) -> int { return (h >= 16 || h * HPER >= HNEED) ? 0 : (HNEED - h * HPER < HPER ? HNEED - h * HPER : HPER); };
    auto dma_ab = [&](int c) {                                      // affine table of chunk c: lanes 0-15 a[64], 16-31 b[64] (fp32)
      const unsigned vo = lane < 32 ? (unsigned)((img * 2 + (lane >> 4 & 1)) * p.Cin + (lane & 15) * 4) * 4u : IG_OOB;
      ig_lds_dma16(abuf_lds + (c & 1) * 1024, vo, rs_ab, (unsigned)((c_begin + min(c, nchunk - 1)) * 64) * 4u);
    };
#if UR_HALO_ABL == 6
    unsigned long long lts[45];
#pragma unroll
    for (int i = 0; i < 45; ++i) lts[i] = 0;
#define LD_TS(i) do { if (c == 1) lts[tap * 5 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define LD_TS(i) do { } while (0)
#endif
    // Prologue: the compute waves fetch patch 0 and weight tile 0 themselves (nine waves feed the DMA path 2-3x faster than one, and
    // they have nothing else to do yet); the loader issues the affine table and weight tile 1 - which its own tap-0 wait covers - and
    // builds its pixel table while those fly.
    if (gnp && do_h) dma_ab(0);
    if (do_w) dma_w(1, 1, 0, WPIECES);
    if (do_h) {
#pragma unroll
      for (int q = 0; q < HNEED; ++q) hpx[q] = halo_pixel(q * 8 + lr);
    }
    if (gnp && do_h) {                                                             // the table is in (tile 1 may still fly)
      if (do_w) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WPIECES) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (gnp) __builtin_amdgcn_s_barrier();                              // (the compute waves normalise patch 0 in between)
    for (int c = 0; c < nchunk; ++c) {
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int kt = c * 9 + tap;
        LD_TS(0);
        if (do_w) dma_w(kt + 2, (tap + 2) % 3, 0, WH);
        if (do_h) {
          dma_h(c + 1, 2 * tap);
          if (gnp && tap == 0) dma_ab(c + 1);
        }
        LD_TS(1);
        // everything this wave issued in EARLIER taps has landed once only this tap's first half may still be in flight
        if (kt + 1 == nk) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // last tap: nothing may land behind the barrier (the epilogue reuses LDS)
        else if (do_w && do_h) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WH + halo_count(2 * tap) + ((gnp && tap == 0) ? 1 : 0)) : "memory");
        else if (do_w) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WH) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(halo_count(2 * tap) + ((gnp && tap == 0) ? 1 : 0)) : "memory");
        LD_TS(2);
        __builtin_amdgcn_s_barrier();
        LD_TS(3);
        if (kt + 1 < nk) {
          if (do_w) dma_w(kt + 2, (tap + 2) % 3, WH, WPIECES);
          if (do_h) dma_h(c + 1, 2 * tap + 1);
        }
        LD_TS(4);
        __builtin_amdgcn_s_barrier();
      }
    }
#if UR_HALO_ABL == 6
    if (blockIdx.x == 0 && blockIdx.y == 0 && lane == 0) {
      unsigned long long* o = reinterpret_cast<unsigned long long*>(p.y) + 128;
#pragma unroll
      for (int i = 0; i < 45; ++i) o[i] = lts[i];
    }
#endif
    return;
  }

  // =================================================== compute waves ==================================================
  const int wm = wid % WM, wn = wid / WM;
  // GroupNorm apply (+ SiLU) of the input, in place in LDS: the pieces the loader issued during tap t (HTAP of them, piece
  // t * HTAP + wid for wave wid < HTAP) were published by the barriers of tap t + 1 and are rewritten at the top of tap t + 2.
  const int chunk = ps ^ ((((wid & 1) << 2) + (lr >> 1)) & 7);      // (HTAP is even: the pieces of a wave share the parity of wid)
  static_assert(HTAP % 2 == 0, "piece parity");
  int hpix[8];                                            // (GNP: kept for the whole kernel; otherwise only for the prologue fetch below)
#pragma unroll
  for (int t = 0; t < 8; ++t) hpix[t] = (wid < HTAP && t * HTAP + wid < HNEED) ? halo_pixel((t * HTAP + wid) * 8 + lr) : -1;
  {
    // prologue fetch: patch 0 (piece t * HTAP + wid, waves < HTAP) and weight tile 0 (pieces wid, wid + NW, ..) - see the loader
    const ig_u32x4 rs_w = ig_make_rsrc(p.w, (unsigned long long)p.Cout * p.ldw * 2);
    const int cb = c_begin * 64;
    const bool second = cb >= p.C1;
    const ig_u32x4 rs_x = second ? ig_make_rsrc(p.x2, (unsigned long long)p.N * p.H * p.W * p.ldx2 * 2) : ig_make_rsrc(p.x, (unsigned long long)p.N * p.H * p.W * p.ldx * 2);
    const unsigned ld2 = (unsigned)(second ? p.ldx2 : p.ldx) * 2u, so = (unsigned)(second ? cb - p.C1 : cb) * 2u;
    if (wid_s < HTAP) {
#pragma unroll
      for (int t = 0; t < 8; ++t)
        if (t * HTAP + wid_s < HNEED)
          ig_lds_dma16(hbuf_lds + (t * HTAP + wid_s) * 1024, hpix[t] >= 0 ? (unsigned)hpix[t] * ld2 + chunk * 16u : IG_OOB, rs_x, so);
    }
#pragma unroll
    for (int j = 0; j < (WPIECES + NW - 1) / NW; ++j) {
      const int q = wid_s + NW * j;
      if (q < WPIECES) {
        const int row = n0 + q * 8 + lr;
        ig_lds_dma16(wring_lds + q * 1024, row < p.Cout ? (unsigned)row * (unsigned)p.ldw * 2u + chunk * 16u : IG_OOB, rs_w, (unsigned)kt0 * 128u);
      }
    }
  }
  auto gn_slot = [&](int c, int t) {                      // this wave's piece of the ones issued in tap t, chunk c's patch
    if (hpix[t] >= 0)
      gn_piece_inplace<F16>(hbuf + (c & 1) * HBYTES + (t * HTAP + wid) * 1024 + lane * 16, gn_load_ab(abuf + (c & 1) * 1024, chunk), p.gn_silu != 0);
  };
  f32x16 acc[FN][FM];
#pragma unroll
  for (int a = 0; a < FN; ++a)
#pragma unroll
    for (int b = 0; b < FM; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
  const int frow = lane & 31, fhalf = lane >> 5;
  const unsigned aw_lds = wring_lds + (wn * WTN + frow) * 128 + ((fhalf ^ ((frow >> 1) & 7)) << 4);
  static_assert(((WTN >> 1) & 7) == 0, "the wave's weight rows must keep the swizzle phase of fragment row 0");
  // LDS addresses (k-step 0) of the fragments tap tp of chunk cc reads: activation rows shifted by (dy, dx), weight ring slot tp % 3
  auto tap_addr = [&](int cc, int tp, unsigned (&ab)[FM]) -> unsigned {
    const int dy = tp / 3, dx = tp % 3;
#pragma unroll
    for (int b = 0; b < FM; ++b) {
      const int hrow = (wm * FM + b + dy) * PW + dx + frow;
      ab[b] = hbuf_lds + (cc & 1) * HBYTES + hrow * 128 + ((fhalf ^ ((hrow >> 1) & 7)) << 4);
    }
    return aw_lds + (tp % 3) * WBYTES;
  };
#if UR_HALO_ABL == 6
  unsigned long long ts[45];
#pragma unroll
  for (int i = 0; i < 45; ++i) ts[i] = 0;
#define WS_TS(i) do { if (c == 1) ts[tap * 5 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define WS_TS(i) do { } while (0)
#endif
  KPipeRings<FM, FN> rings;
  kpipe_init(rings);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's pieces of patch 0 / weight tile 0
  __builtin_amdgcn_s_barrier();                           // ... and everybody else's (+ the loader's affine table)
  if (gnp) {                                              // chunk 0's patch is normalised before anyone reads it
#pragma unroll
    for (int t = 0; t < 8; ++t) gn_slot(0, t);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  {
    unsigned ab0[FM];
    const unsigned aw0 = tap_addr(0, 0, ab0);
    kpipe<FM, FN, F16, 0>(acc, rings, ab0, aw0, ab0, aw0);
  }
#if UR_HALO_ABL == 6
  const unsigned long long loop_c0 = __builtin_readcyclecounter(), loop_r0 = __builtin_amdgcn_s_memrealtime();
#endif
#if UR_HALO_ABL == 7
  const unsigned long long wg_t1 = __builtin_amdgcn_s_memrealtime();
#endif
  for (int c = 0; c < nchunk; ++c) {
    const bool next_chunk = c + 1 < nchunk;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      // the pieces issued two taps ago were published by the previous tap's barriers: normalise them next to this tap's MFMAs
      // (tap 0: the ones of tap 7 of the previous chunk - this chunk's own patch, first read by its tap 6)
      if (gnp && tap >= 2 && next_chunk) gn_slot(c + 1, tap >= 2 ? tap - 2 : 0);
      if (gnp && tap == 0 && c > 0) gn_slot(c, 7);
      unsigned ab[FM], abn[FM];
      const unsigned aw = tap_addr(c, tap, ab), awn = tap_addr(c + (tap == 8 ? 1 : 0), (tap + 1) % 9, abn);
      WS_TS(0);
      kpipe<FM, FN, F16, 1>(acc, rings, ab, aw, abn, awn);
      WS_TS(1);
      __builtin_amdgcn_s_barrier();                       // weight tile kt + 1 (and every older piece) is in LDS
      WS_TS(2);
      kpipe<FM, FN, F16, 2>(acc, rings, ab, aw, abn, awn);
      WS_TS(3);
      __builtin_amdgcn_s_barrier();                       // everyone has read weight tile kt: its slot takes tile kt + 3
      WS_TS(4);
    }
  }
  kpipe_drain(rings);
  asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");       // last MFMA of the asm tap body -> VALU reads of the accumulators
#if UR_HALO_ABL == 6                                          // phase timers -> the first bytes of y (no epilogue)
  if (acc[0][0][0] == 123.456f) reinterpret_cast<uint16_t*>(p.y)[0] = 1;
  if (blockIdx.x == 0 && blockIdx.y == 0 && lane == 0 && (wid == 0 || wid == NW / 2)) {
    unsigned long long* o = reinterpret_cast<unsigned long long*>(p.y) + (wid ? 64 : 0);
#pragma unroll
    for (int i = 0; i < 45; ++i) o[i] = ts[i];
  }
  if (lane == 0 && wid == 0 && blockIdx.x < 256) {            // K-loop duration of every workgroup in shader cycles and in 100 MHz ticks
    unsigned long long* o = reinterpret_cast<unsigned long long*>(p.y) + 256 + 2 * blockIdx.x;
    o[0] = __builtin_readcyclecounter() - loop_c0;
    o[1] = __builtin_amdgcn_s_memrealtime() - loop_r0;
  }
#elif UR_HALO_ABL == 7
  const unsigned long long wg_t2 = __builtin_amdgcn_s_memrealtime();
  igemm_epilogue<FM, FN, WTM, WTN, BM, BN, NT, F16, false>(p, acc, m0, n0, wm, wn, lane, 0, sz, smem);
  __syncthreads();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lane == 0 && wid == 0 && blockIdx.x < 256 && blockIdx.y == 0) {
    unsigned long long* o = reinterpret_cast<unsigned long long*>(p.y) + 4 * blockIdx.x;
    o[0] = wg_t0; o[1] = wg_t1; o[2] = wg_t2; o[3] = __builtin_amdgcn_s_memrealtime();
  }
#else
  igemm_epilogue<FM, FN, WTM, WTN, BM, BN, NT, F16, false>(p, acc, m0, n0, wm, wn, lane, 0, sz, smem);
#endif
}

// Thin-N halo launch (round 5): 3x3 convolutions onto <= 32 output channels - the three conv_out layers (VAE decoder 128 -> 3 at 512 x 512:
// 747 us at 52 TF/s on the generic 256 x 32 kernel, which gathers every pixel nine times; UNet 320 -> 4; VAE encoder 512 -> 8).  Same
// patch-in-LDS loader as every halo conv (self-loading kernel: with one weight piece per wave and tap there is nothing for a loader wave to
// do), tile = 8 x 32 pixels x 32 channel rows (rows >= Cout read zero through the descriptor's range check), hipcc-scheduled 1 x 1 fragment
// body.  The launch is bound by reading the input once (x 1.3 for the halo), not by the matrix pipe.
template <int TH, int BN, int WM, int WN>
int launch_halo_thin(ConvK& k, hipStream_t s) {
  constexpr int NW = WM * WN, HPIX = (TH + 2) * 34, HSLOTS = (HPIX + 8 * NW - 1) / (8 * NW);
  constexpr int HBYTES = HSLOTS * NW * 1024;
  constexpr int lds = 2 * HBYTES + 3 * (BN * 128 > NW * 1024 ? BN * 128 : NW * 1024) + 2048;
  static_assert(lds <= 160 * 1024, "LDS budget");
  k.prologue_ok = 0;
  k.tiles_m = k.N * (k.OH / TH) * (k.OW / 32);
  k.tiles_n = (k.Cout + BN - 1) / BN;
  k.splitk = 1;
  k.nk_per_split = k.nk;
  set_gn_plan(k, false, 0);
  if (k.dry) { k.plan_tn = k.tiles_n; return UR_OK; }
  k.patch_tw = 32;
  static ur::DeviceOnce attr_once;      // the attribute is per device
  if (auto once_guard = attr_once.first()) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_halo_kernel<TH, BN, WM, WN, UR_TU_F16 != 0, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  }
  UR_F16_SWITCH(k, hipLaunchKernelGGL((igemm_halo_kernel<TH, BN, WM, WN, F16, false>), dim3(k.tiles_m * k.tiles_n, 1), dim3(NW * 64), lds, s, k));
  return ur::check_launch("ur_conv2d_nhwc");
}

template <int TH, int BN, int WM, int WN>
int launch_halo(ConvK& k, hipStream_t s) {
  constexpr int NW = WM * WN, BM = TH * 32, HPIX = (TH + 2) * 34, HSLOTS = (HPIX + 8 * NW - 1) / (8 * NW);
  constexpr int HBYTES = HSLOTS * NW * 1024;
  constexpr int lds_loop = 2 * HBYTES + 3 * BN * 128 + 2048, lds_epi = epi_lds_bytes<BM, BN, NW * 64>();     // (+ 2 x 1 KiB affine tables)
  constexpr int lds = lds_loop > lds_epi ? lds_loop : lds_epi;
  static_assert(lds <= 160 * 1024, "LDS budget");
  // in-loader GroupNorm: the 128-wide tile only (the 160-wide variant's 5 x 16 accumulators leave no registers for it: it spilled)
  constexpr bool GN_OK = BN <= 128;
  k.prologue_ok = GN_OK ? 1 : 0;
  k.tiles_m = k.N * (k.OH / TH) * (k.OW / 32);
  k.tiles_n = (k.Cout + BN - 1) / BN;
  const int nchunk = k.nk / 9;
  const long long tiles = (long long)k.tiles_m * k.tiles_n;
  int splitk = 1;
  static const bool no_hsplit = getenv("UR_IGEMM_NOHSPLIT") != nullptr;
  if (!no_hsplit && tiles <= 128 && k.ws && k.y) {     // <= half a round of CUs: split the chunk range, reduce in a second pass
    splitk = (int)std::min<long long>(256 / tiles, std::max(1, nchunk / 2));
    while (splitk > 1 && (long long)splitk * k.M * k.Cout * 4 > (long long)k.ws_bytes_) --splitk;
  }
  const int cps = (nchunk + splitk - 1) / splitk;
  k.splitk = (nchunk + cps - 1) / cps;
  k.nk_per_split = cps * 9;
  set_gn_plan(k, true, (k.OH / TH) * (k.OW / 32));         // a patch never leaves its image: one partial per patch
  if (k.dry) { k.plan_tn = k.splitk > 1 ? 1 : k.tiles_n; return UR_OK; }
  k.patch_tw = 32;
  static ur::DeviceOnce attr_once;      // the attribute is per device
  if (auto once_guard = attr_once.first()) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_halo_kernel<TH, BN, WM, WN, UR_TU_F16 != 0, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_halo_kernel<TH, BN, WM, WN, UR_TU_F16 != 0, GN_OK>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_halo_ws_kernel<TH, BN, WM, WN, UR_TU_F16 != 0, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_halo_ws_kernel<TH, BN, WM, WN, UR_TU_F16 != 0, GN_OK>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  }
  static const bool no_ws = getenv("UR_HALO_NOWS") != nullptr;          // A/B: every wave loads for itself (rounds 1-2 structure)
  if (no_ws) {
    if (GN_OK && k.gn_ab) UR_F16_SWITCH(k, hipLaunchKernelGGL((igemm_halo_kernel<TH, BN, WM, WN, F16, GN_OK>), dim3(k.tiles_m * k.tiles_n, k.splitk), dim3(NW * 64), lds, s, k));
    else UR_F16_SWITCH(k, hipLaunchKernelGGL((igemm_halo_kernel<TH, BN, WM, WN, F16, false>), dim3(k.tiles_m * k.tiles_n, k.splitk), dim3(NW * 64), lds, s, k));
  } else {
    static const bool ld1 = getenv("UR_HALO_LD1") != nullptr;             // A/B: one loader wave for weights AND patch pieces (rounds 3-5)
    const dim3 blk((NW + (ld1 ? 1 : 2)) * 64);
    if (GN_OK && k.gn_ab) UR_F16_SWITCH(k, hipLaunchKernelGGL((igemm_halo_ws_kernel<TH, BN, WM, WN, F16, GN_OK>), dim3(k.tiles_m * k.tiles_n, k.splitk), blk, lds, s, k));
    else UR_F16_SWITCH(k, hipLaunchKernelGGL((igemm_halo_ws_kernel<TH, BN, WM, WN, F16, false>), dim3(k.tiles_m * k.tiles_n, k.splitk), blk, lds, s, k));
  }
  if (k.splitk > 1) {
    k.patch_tw = 0;                                        // the partial planes are plain [M][Cout]
    launch_splitk_reduce(k, s);
  }
  return ur::check_launch("ur_conv2d_nhwc");
}

// =====================================================================================================================
// Whole-image halo tiles for the small feature maps (16 x 16: one image per tile; 8 x 8: four images per tile; BM = 256).
// Same idea as igemm_halo_kernel - the input patch (with its zero border) is DMA'd into LDS once per 64-channel chunk and
// the nine taps read shifted fragments from it - but here the tile is TH x TW x NIMG pixels of WHOLE images, so tile rows
// are contiguous in M (plain epilogue / split-K paths apply), and the K loop is split over channel chunks (blockIdx.y)
// because these layers have few tiles and K = 9 x 1280..2560.  A 32-pixel fragment spans several image rows; the slot
// swizzle f(row) = ((row_in_image >> 1) - halo_y) & 7 keeps its ds_read_b128 conflict-free for both shapes
// (searched with tools/lds_conflicts.py), at the price of a per-piece source chunk on the DMA side.
template <int TH, int TW, int NIMG, int BN, int WM, int WN, bool F16, bool GNP>
__global__ __launch_bounds__(WM* WN * 64) void igemm_halo_img_kernel(const ConvK p) {
  constexpr int NW = WM * WN, BM = TH * TW * NIMG, PW = TW + 2, HP = (TH + 2) * PW, HPIX = HP * NIMG;
  constexpr int HSLOTS = (HPIX + 8 * NW - 1) / (8 * NW);
  constexpr int HPIECES = HSLOTS * NW, HBYTES = HPIECES * 1024;
  constexpr int WBYTES = BN * 128, WPIECES = BN / 8, WPW = (WPIECES + NW - 1) / NW;
  constexpr int WTM = BM / WM, WTN = BN / WN, FM = WTM / 32, FN = WTN / 32, NT = NW * 64;
  static_assert(BM == 256 && NW == 8 && WTM % 32 == 0 && WTN % 32 == 0 && HSLOTS <= 8, "tile shape");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const hbuf = smem;
  unsigned char* const wring = smem + 2 * HBYTES;
  unsigned char* const abuf = wring + 3 * WBYTES;        // GN_OK only: 2 x 1 KiB affine tables (see igemm_halo_kernel)
  // GroupNorm apply fused into the loader: one image per tile only (the 8x8x4 shape has no LDS left for the tables)
  constexpr bool GN_OK = NIMG == 1 && HSLOTS <= 7;
  constexpr bool gnp = GN_OK && GNP;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid % WM, wn = wid / WM;
  int sz = blockIdx.y, tn, tmi;
  if (p.wmajor) {
    // weight-major XCD map (round 6; 1-D grid, workgroup L runs on XCD L % 8): these layers multiply a few MB of activations with
    // 15 - 59 MB of weights, so every (channel tile, chunk split) slice of the weights belongs to ONE XCD - its tiles_m image tiles are
    // consecutive dispatch slots there, the slice leaves HBM / the Infinity Cache once and the other tiles hit that XCD's L2.  The
    // tile-major map below gave every XCD one image and ALL the weights: 291 MB through the fabric per 1280 -> 1280 @16x16 launch against
    // 40 MB algorithmic (profiles/r6_pmc_conv.json).
    const int L = blockIdx.x, xcd = L & 7, j = L >> 3;
    const int g = (j / p.tiles_m) * 8 + xcd;
    if (g >= p.tiles_n * p.splitk) return;
    tn = g % p.tiles_n; sz = g / p.tiles_n; tmi = j % p.tiles_m;
  } else {
    int id = blockIdx.x;
    const int nt = p.tiles_m * p.tiles_n, q = nt >> 3, r = nt & 7, xcd = id & 7, idx = id >> 3;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    tn = id % p.tiles_n; tmi = id / p.tiles_n;
  }
  const int n0 = tn * BN, img0 = tmi * NIMG, m0 = tmi * BM;

  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  const uint16_t* __restrict__ X1 = p.x;
  const uint16_t* __restrict__ X2 = p.x2;
  const uint16_t* __restrict__ Wt = p.w;
  const uint16_t* zero = reinterpret_cast<const uint16_t*>(g_zero_page);
  const int lr = lane >> 3, ps = lane & 7;

  // halo pieces this wave issues per chunk: input pixel (or -1 = zero border) and the lane's logical 16-byte chunk
  int hpix[HSLOTS], hchk[HSLOTS];
#pragma unroll
  for (int t = 0; t < HSLOTS; ++t) {
    const int hr = (t * NW + wid) * 8 + lr;
    const int im = hr / HP, rin = hr - im * HP, hy = rin / PW, hx = rin - hy * PW;
    const int iy = hy - 1, ix = hx - 1;
    const bool v = hr < HPIX && img0 + im < p.N && (unsigned)iy < (unsigned)TH && (unsigned)ix < (unsigned)TW;
    // fused nearest-2x upsample (p.ups): the patch is TH x TW pixels of the UPSAMPLED image, every piece reads its source pixel
    const int sy = p.ups ? iy >> 1 : iy, sx = p.ups ? ix >> 1 : ix;
    hpix[t] = v ? ((img0 + im) * p.H + sy) * p.W + sx : -1;
    hchk[t] = ps ^ (((rin >> 1) - hy) & 7);
  }
  const int wchunk = ps ^ ((((wid & 1) << 2) + (lr >> 1)) & 7);      // weight tile keeps the (row>>1)&7 swizzle
  int woff[WPW];
#pragma unroll
  for (int i = 0; i < WPW; ++i) {
    const int qq = (wid + NW * i < WPIECES) ? wid + NW * i : wid;
    const int row = n0 + qq * 8 + lr;
    woff[i] = row < p.Cout ? row * p.ldw : -1;
  }
  const int nchunk_all = p.nk / 9, cps = p.nk_per_split / 9;
  const int c_begin = sz * cps, c_end = min(nchunk_all, c_begin + cps);
  const int nchunk = c_end - c_begin, nk = nchunk * 9, kt0 = c_begin * 9;

  // buffer-descriptor LDS-DMA (ig_lds_dma16): zero border / ragged rows through the descriptor's range check
  const int wid_s = __builtin_amdgcn_readfirstlane(wid);
  const ig_u32x4 rs_w = ig_make_rsrc(Wt, (unsigned long long)p.Cout * p.ldw * 2);
  const ig_u32x4 rs_x1 = ig_make_rsrc(X1, (unsigned long long)p.N * p.H * p.W * p.ldx * 2);
  const ig_u32x4 rs_x2 = ig_make_rsrc(X2 ? X2 : X1, (unsigned long long)p.N * p.H * p.W * (X2 ? p.ldx2 : p.ldx) * 2);
  const unsigned wring_lds = (unsigned)(uintptr_t)(lptr_t)wring, hbuf_lds = (unsigned)(uintptr_t)(lptr_t)hbuf;
  unsigned wvo[WPW];
#pragma unroll
  for (int i = 0; i < WPW; ++i) wvo[i] = woff[i] >= 0 ? (unsigned)woff[i] * 2u + wchunk * 16u : IG_OOB;
  auto issue_w = [&](int kt, int ring) {
#pragma unroll
    for (int i = 0; i < WPW; ++i) {
      const int qq = (wid_s + NW * i < WPIECES) ? wid_s + NW * i : wid_s;
      ig_lds_dma16(wring_lds + ring * WBYTES + qq * 1024, wvo[i], rs_w, (unsigned)(kt0 + kt) * 128u);
    }
  };
  auto issue_h = [&](int c, int t) {                                // c = chunk index local to this split
    const int cb = (c_begin + c) * 64;                              // first channel of the chunk: decides the source (C1 % 64 == 0)
    const bool second = cb >= p.C1;
    const unsigned ld2 = (unsigned)(second ? p.ldx2 : p.ldx) * 2u;
    const unsigned vo = hpix[t] >= 0 ? (unsigned)hpix[t] * ld2 + hchk[t] * 16u : IG_OOB;
    const unsigned m0v = hbuf_lds + (c & 1) * HBYTES + (t * NW + wid_s) * 1024;
    if (second) ig_lds_dma16(m0v, vo, rs_x2, (unsigned)(cb - p.C1) * 2u);
    else ig_lds_dma16(m0v, vo, rs_x1, (unsigned)cb * 2u);
  };
  auto issue_ab = [&](int c) {                            // affine table of chunk c -> abuf[c & 1] (see igemm_halo_kernel)
    const float* t = p.gn_ab + ((long long)img0 * 2 + (lane >> 4 & 1)) * p.Cin + (c_begin + c) * 64 + (lane & 15) * 4;
    const void* g = lane < 32 ? (const void*)t : (const void*)zero;
    __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(abuf + (c & 1) * 1024), 16, 0, 0);
  };
  auto gn_slot = [&](int c, int t) {                      // this wave's halo piece t of chunk c, in place (per-piece chunk here)
    if (hpix[t] >= 0)
      gn_piece_inplace<F16>(hbuf + (c & 1) * HBYTES + (t * NW + wid) * 1024 + lane * 16, gn_load_ab(abuf + (c & 1) * 1024, hchk[t]), p.gn_silu != 0);
  };

  f32x16 acc[FN][FM];
#pragma unroll
  for (int a = 0; a < FN; ++a)
#pragma unroll
    for (int b = 0; b < FM; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
  const int frow = lane & 31, fhalf = lane >> 5;
  // this lane's pixel in each of the wave's FM fragments: halo row (tap 0,0) and halo y
  int hrow0[FM], hin0[FM], hy0[FM];
#pragma unroll
  for (int b = 0; b < FM; ++b) {
    const int pix = wm * WTM + b * 32 + frow;
    const int im = pix / (TH * TW), r = pix - im * (TH * TW), y = r / TW, x = r - y * TW;
    hin0[b] = y * PW + x;
    hrow0[b] = im * HP + hin0[b];
    hy0[b] = y;
  }
  const unsigned hb_lds = (unsigned)(uintptr_t)(lptr_t)hbuf;
  const unsigned aw_lds = (unsigned)(uintptr_t)(lptr_t)wring + (wn * WTN + frow) * 128 + ((fhalf ^ ((frow >> 1) & 7)) << 4);
  static_assert(((WTN >> 1) & 7) == 0, "the wave's weight rows must keep the swizzle phase of fragment row 0");

  if (gnp) issue_ab(0);
#pragma unroll
  for (int t = 0; t < HSLOTS; ++t) issue_h(0, t);
  issue_w(0, 0);
  if (nk > 1) {
    issue_w(1, 1);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WPW) : "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  if (gnp) {
#pragma unroll
    for (int t = 0; t < HSLOTS; ++t) gn_slot(0, t);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();

  for (int c = 0; c < nchunk; ++c) {                       // chunks x 9 unrolled taps (see igemm_halo_kernel)
    const unsigned char* hb = hbuf + (c & 1) * HBYTES;
    const bool next_chunk = c + 1 < nchunk;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int kt = c * 9 + tap;
      const bool more_w = kt + 2 < nk;
      const bool more_h = tap < HSLOTS && next_chunk;
      if (more_w) issue_w(kt + 2, (tap + 2) % 3);
      if (tap < HSLOTS) { if (next_chunk) issue_h(c + 1, tap < HSLOTS ? tap : 0); }
      const bool ab_now = gnp && tap == 0 && next_chunk;
      if (ab_now) issue_ab(c + 1);
      // piece (tap - 2) of the next chunk landed with the previous iteration's wait: normalise it in LDS next to this tap's MFMAs
      // (splitting the two waves of a SIMD to opposite sides of the MFMA block measured no gain and cost registers)
      if (gnp && tap >= 2 && tap - 2 < HSLOTS && next_chunk) gn_slot(c + 1, (tap >= 2 && tap - 2 < HSLOTS) ? tap - 2 : 0);
      if constexpr (ktile_asm_ok<FM, FN>()) {
        const int dy = tap / 3, dx = tap % 3;
        unsigned ab[FM];
#pragma unroll
        for (int b = 0; b < FM; ++b) {
          const int hsw = (((hin0[b] + dy * PW + dx) >> 1) - (hy0[b] + dy)) & 7;
          ab[b] = hb_lds + (c & 1) * HBYTES + (hrow0[b] + dy * PW + dx) * 128 + ((fhalf ^ hsw) << 4);
        }
        ktile_mma<FM, FN, F16>(acc, ab, aw_lds + (tap % 3) * WBYTES);
      } else {
        const int dy = tap / 3, dx = tap % 3;
        const unsigned char* wsm = wring + (tap % 3) * WBYTES;
        int hro[FM], hsw[FM];
#pragma unroll
        for (int b = 0; b < FM; ++b) {
          hro[b] = (hrow0[b] + dy * PW + dx) * 128;
          hsw[b] = (((hin0[b] + dy * PW + dx) >> 1) - (hy0[b] + dy)) & 7;
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const int slot = ks * 2 + fhalf;
          typename Frag<F16>::type bfr[FM], afr[FN];
#pragma unroll
          for (int b = 0; b < FM; ++b) bfr[b] = *reinterpret_cast<const typename Frag<F16>::type*>(hb + hro[b] + ((slot ^ hsw[b]) << 4));
#pragma unroll
          for (int a = 0; a < FN; ++a) {
            const int row = wn * WTN + a * 32 + frow;
            afr[a] = *reinterpret_cast<const typename Frag<F16>::type*>(wsm + row * 128 + ((slot ^ ((row >> 1) & 7)) << 4));
          }
#pragma unroll
          for (int a = 0; a < FN; ++a)
#pragma unroll
            for (int b = 0; b < FM; ++b)
              acc[a][b] = mfma16t(afr[a], bfr[b], acc[a][b]);
        }
      }
      if (ab_now) {                                       // tap 0 with the next chunk's affine table in flight as well
        if (more_w) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WPW + 2) : "memory");
        else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      } else if (more_w && more_h) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WPW + 1) : "memory");
      else if (more_w) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WPW) : "memory");
      else if (more_h) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      // lgkmcnt(0): (1) this wave's fragment reads of the ring slot / halo buffer that the NEXT tap's DMA overwrites are retired
      // before anyone passes the barrier (hipcc leaves the last reads outstanding across it otherwise - see igemm_glds_kernel);
      // (2) the in-place GroupNorm writes are published by the barrier
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  }
  asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");      // last MFMA of the asm tap body -> VALU reads of the accumulators
  igemm_epilogue<FM, FN, WTM, WTN, BM, BN, NT, F16, false>(p, acc, m0, n0, wm, wn, lane, 0, sz, smem);
}

// ---------------------------------------------------------------------------------------------------------------------
// Wave-specialised form of igemm_halo_img_kernel (round 6): 8 compute waves that execute no memory instruction inside the K loop +
// 2 loader waves - one issues the weight tile two taps ahead, the other the next chunk's halo pieces (spread over taps 0 .. 7) - the
// structure of igemm_halo_ws_kernel on the whole-image tile.  (With ONE loader wave the kernel was 6 % SLOWER than the self-loading one:
// 16 weight pieces at ~38 cycles + 8 patch pieces at ~85 exceed the 1 024 MFMA cycles of a tap; two waves feed the CU's DMA path in
// parallel: 71.2 -> 66.3 us in-graph at 1280 -> 1280 @16x16.)  Same LDS plan, same ring discipline (slot
// (tap + 2) % 3 was read in the previous tap: every compute wave passed that tap's barrier behind its own lgkmcnt(0)), same
// accumulation order and epilogue: results are bit-identical to the self-loading kernel.  No fused GroupNorm prologue here (the launches
// that use it stay on the self-loading kernel).  The prologue fetch (chunk 0's patch, weight tiles 0 and 1) is issued by nine of the ten waves.
template <int TH, int TW, int NIMG, int BN, int WM, int WN, bool F16>
__global__ __launch_bounds__((WM * WN + 2) * 64) void igemm_halo_img_ws_kernel(const ConvK p) {
  constexpr int NW = WM * WN, BM = TH * TW * NIMG, PW = TW + 2, HP = (TH + 2) * PW, HPIX = HP * NIMG;
  constexpr int HSLOTS = (HPIX + 8 * NW - 1) / (8 * NW);
  constexpr int HPIECES = HSLOTS * NW, HBYTES = HPIECES * 1024;
  constexpr int WBYTES = BN * 128, WPIECES = BN / 8;
  constexpr int WTM = BM / WM, WTN = BN / WN, FM = WTM / 32, FN = WTN / 32, NT = NW * 64;
  static_assert(BM == 256 && NW == 8 && WTM % 32 == 0 && WTN % 32 == 0 && HSLOTS <= 8 && ktile_asm_ok<FM, FN>() && WPIECES % NW == 0, "tile shape");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const hbuf = smem;
  unsigned char* const wring = smem + 2 * HBYTES;
  typedef __attribute__((address_space(3))) void* lptr_t;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wid_s = __builtin_amdgcn_readfirstlane(wid);
  int sz = blockIdx.y, tn, tmi;
  if (p.wmajor) {                                            // (see igemm_halo_img_kernel)
    const int L = blockIdx.x, xcd = L & 7, j = L >> 3;
    const int g = (j / p.tiles_m) * 8 + xcd;
    if (g >= p.tiles_n * p.splitk) return;
    tn = g % p.tiles_n; sz = g / p.tiles_n; tmi = j % p.tiles_m;
  } else {
    int id = blockIdx.x;
    const int nt = p.tiles_m * p.tiles_n, q = nt >> 3, r = nt & 7, xcd = id & 7, idx = id >> 3;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    tn = id % p.tiles_n; tmi = id / p.tiles_n;
  }
  const int n0 = tn * BN, img0 = tmi * NIMG, m0 = tmi * BM;
  const int lr = lane >> 3, ps = lane & 7;
  const int nchunk_all = p.nk / 9, cps = p.nk_per_split / 9;
  const int c_begin = sz * cps, c_end = min(nchunk_all, c_begin + cps);
  const int nchunk = c_end - c_begin, nk = nchunk * 9, kt0 = c_begin * 9;
  const ig_u32x4 rs_w = ig_make_rsrc(p.w, (unsigned long long)p.Cout * p.ldw * 2);
  const ig_u32x4 rs_x1 = ig_make_rsrc(p.x, (unsigned long long)p.N * p.H * p.W * p.ldx * 2);
  const ig_u32x4 rs_x2 = ig_make_rsrc(p.x2 ? p.x2 : p.x, (unsigned long long)p.N * p.H * p.W * (p.x2 ? p.ldx2 : p.ldx) * 2);
  const unsigned wring_lds = (unsigned)(uintptr_t)(lptr_t)wring, hbuf_lds = (unsigned)(uintptr_t)(lptr_t)hbuf;

  // halo piece q (LDS rows 8q .. 8q+7 of the patch): this lane's source pixel (or -1: zero border / surplus row) and its logical 16-byte chunk
  auto piece_of = [&](int q, int& pix, int& chk) {
    const int hr = q * 8 + lr;
    const int im = hr / HP, rin = hr - im * HP, hy = rin / PW, hx = rin - hy * PW;
    const int iy = hy - 1, ix = hx - 1;
    const bool v = hr < HPIX && img0 + im < p.N && (unsigned)iy < (unsigned)TH && (unsigned)ix < (unsigned)TW;
    const int sy = p.ups ? iy >> 1 : iy, sx = p.ups ? ix >> 1 : ix;
    pix = v ? ((img0 + im) * p.H + sy) * p.W + sx : -1;
    chk = ps ^ (((rin >> 1) - hy) & 7);
  };
  auto dma_piece = [&](int c, int q, int pix, int chk) {          // piece q of chunk c (local to this split) -> halo buffer c & 1
    const int cb = (c_begin + c) * 64;
    const bool second = cb >= p.C1;
    const unsigned ld2 = (unsigned)(second ? p.ldx2 : p.ldx) * 2u;
    const unsigned vo = pix >= 0 ? (unsigned)pix * ld2 + chk * 16u : IG_OOB;
    const unsigned m0v = hbuf_lds + (c & 1) * HBYTES + q * 1024;
    if (second) ig_lds_dma16(m0v, vo, rs_x2, (unsigned)(cb - p.C1) * 2u);
    else ig_lds_dma16(m0v, vo, rs_x1, (unsigned)cb * 2u);
  };
  auto wvo_of = [&](int q) -> unsigned {                           // weight piece q (tile rows 8q .. 8q+7): the (row>>1)&7 swizzle on the source
    const int row = n0 + q * 8 + lr;
    const int wchunk = ps ^ ((((q & 1) << 2) + (lr >> 1)) & 7);
    return row < p.Cout ? (unsigned)row * (unsigned)p.ldw * 2u + wchunk * 16u : IG_OOB;
  };

  if (wid_s == NW) {
    // ============================================ loader wave 1: the weight tiles ============================================
    unsigned wvo[WPIECES];
#pragma unroll
    for (int q = 0; q < WPIECES; ++q) wvo[q] = wvo_of(q);
    auto dma_w = [&](int kt, int ring) {
      const unsigned so = (unsigned)(kt0 + kt) * 128u;
#pragma unroll
      for (int q = 0; q < WPIECES; ++q) ig_lds_dma16(wring_lds + ring * WBYTES + q * 1024, wvo[q], rs_w, so);
    };
    if (nk > 1) dma_w(1, 1);                                       // prologue share: weight tile 1
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int kt = 0; kt < nk; ++kt) {
      // (9 % 3 == 0: the ring slot of tile kt is kt % 3 = tap % 3)
      if (kt + 2 < nk) {
        dma_w(kt + 2, (kt + 2) % 3);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WPIECES) : "memory");      // everything issued in earlier taps has landed
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
    }
    return;
  }
  if (wid_s == NW + 1) {
    // ====================================== loader wave 2: the next chunk's halo pieces ======================================
    int hpx[HPIECES];                                              // pix * 8 + chk (pix = -1 -> negative)
#pragma unroll
    for (int q = 0; q < HPIECES; ++q) { int pix, chk; piece_of(q, pix, chk); hpx[q] = pix >= 0 ? pix * 8 + chk : -1; }
    __builtin_amdgcn_s_barrier();
    constexpr int PPT = (HPIECES + 7) / 8;                         // pieces per tap over taps 0 .. 7 (tap 8 issues none: all landed before the chunk ends)
    for (int c = 0; c < nchunk; ++c) {
      const bool next_chunk = c + 1 < nchunk;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        if (next_chunk && tap < 8) {
#pragma unroll
          for (int j = 0; j < PPT; ++j) {
            const int q = tap * PPT + j;
            if (q < HPIECES) dma_piece(c + 1, q, hpx[q < HPIECES ? q : 0] >= 0 ? hpx[q < HPIECES ? q : 0] >> 3 : -1, hpx[q < HPIECES ? q : 0] & 7);
          }
        }
        if (tap == 8 || !next_chunk) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPT) : "memory");
        __builtin_amdgcn_s_barrier();
      }
    }
    return;
  }

  // =================================================== compute waves ==================================================
  const int wm = wid % WM, wn = wid / WM;
  {
    // prologue fetch: patch of chunk 0 (pieces t * NW + wid) and weight tile 0 (pieces wid, wid + NW, ..)
#pragma unroll
    for (int t = 0; t < HSLOTS; ++t) { int pix, chk; piece_of(t * NW + wid_s, pix, chk); dma_piece(0, t * NW + wid_s, pix, chk); }
#pragma unroll
    for (int i = 0; i < WPIECES / NW; ++i) ig_lds_dma16(wring_lds + (wid_s + NW * i) * 1024, wvo_of(wid_s + NW * i), rs_w, (unsigned)kt0 * 128u);
  }
  f32x16 acc[FN][FM];
#pragma unroll
  for (int a = 0; a < FN; ++a)
#pragma unroll
    for (int b = 0; b < FM; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
  const int frow = lane & 31, fhalf = lane >> 5;
  int hrow0[FM], hin0[FM], hy0[FM];
#pragma unroll
  for (int b = 0; b < FM; ++b) {
    const int pix = wm * WTM + b * 32 + frow;
    const int im = pix / (TH * TW), r = pix - im * (TH * TW), y = r / TW, x = r - y * TW;
    hin0[b] = y * PW + x;
    hrow0[b] = im * HP + hin0[b];
    hy0[b] = y;
  }
  const unsigned aw_lds = wring_lds + (wn * WTN + frow) * 128 + ((fhalf ^ ((frow >> 1) & 7)) << 4);
  static_assert(((WTN >> 1) & 7) == 0, "the wave's weight rows must keep the swizzle phase of fragment row 0");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  for (int c = 0; c < nchunk; ++c) {
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int dy = tap / 3, dx = tap % 3;
      unsigned ab[FM];
#pragma unroll
      for (int b = 0; b < FM; ++b) {
        const int hsw = (((hin0[b] + dy * PW + dx) >> 1) - (hy0[b] + dy)) & 7;
        ab[b] = hbuf_lds + (c & 1) * HBYTES + (hrow0[b] + dy * PW + dx) * 128 + ((fhalf ^ hsw) << 4);
      }
      ktile_mma<FM, FN, F16>(acc, ab, aw_lds + (tap % 3) * WBYTES);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // this wave's fragment reads of the slot / buffer the loader overwrites next are retired
      __builtin_amdgcn_s_barrier();
    }
  }
  asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");      // last MFMA of the asm tap body -> VALU reads of the accumulators
  igemm_epilogue<FM, FN, WTM, WTN, BM, BN, NT, F16, false>(p, acc, m0, n0, wm, wn, lane, 0, sz, smem);
}

template <int TH, int TW, int NIMG, int BN, int WM, int WN>
int launch_halo_img(ConvK& k, hipStream_t s) {
  constexpr int NW = WM * WN, BM = TH * TW * NIMG, HPIX = (TH + 2) * (TW + 2) * NIMG, HSLOTS = (HPIX + 8 * NW - 1) / (8 * NW);
  constexpr int HBYTES = HSLOTS * NW * 1024;
  constexpr bool GN_OK = NIMG == 1 && HSLOTS <= 7;          // must match the kernel's
  constexpr int lds_loop = 2 * HBYTES + 3 * BN * 128 + (GN_OK ? 2048 : 0), lds_epi = epi_lds_bytes<BM, BN, NW * 64>();
  constexpr int lds = lds_loop > lds_epi ? lds_loop : lds_epi;
  static_assert(lds <= 160 * 1024, "LDS budget");
  k.prologue_ok = GN_OK ? 1 : 0;
  k.tiles_m = (k.N + NIMG - 1) / NIMG;
  k.tiles_n = (k.Cout + BN - 1) / BN;
  const int nchunk = k.nk / 9;
  const long long tiles = (long long)k.tiles_m * k.tiles_n;
  int splitk = 1;
  if (tiles < 200 && k.ws && k.y) {      // one workgroup per CU: split the chunk range so that tiles * splits <= 256 (a single round)
    splitk = (int)std::min<long long>(std::max<long long>(256 / tiles, 1), std::max(1, nchunk / 2));
    while (splitk > 1 && (long long)splitk * k.M * k.Cout * 4 > (long long)k.ws_bytes_) --splitk;
  }
  const int cps = (nchunk + splitk - 1) / splitk;
  k.splitk = (nchunk + cps - 1) / cps;
  k.nk_per_split = cps * 9;
  set_gn_plan(k, (k.OHW % BM) == 0, k.OHW / BM);
  if (k.dry) { k.plan_tn = k.splitk > 1 ? 1 : k.tiles_n; return UR_OK; }
  static ur::DeviceOnce attr_once;      // the attribute is per device
  if (auto once_guard = attr_once.first()) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_halo_img_kernel<TH, TW, NIMG, BN, WM, WN, UR_TU_F16 != 0, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_halo_img_kernel<TH, TW, NIMG, BN, WM, WN, UR_TU_F16 != 0, GN_OK>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  }
  // weight-major XCD map when the weights outweigh the activations and the per-XCD share still fits one round of its 32 CUs
  static const bool no_wmajor = getenv("UR_HIMG_NOWMAJOR") != nullptr;
  const int slices = k.tiles_n * k.splitk, per_xcd = ((slices + 7) / 8) * k.tiles_m;
  k.wmajor = !no_wmajor && (long long)k.Cout * k.Ktot > (long long)k.N * k.H * k.W * k.Cin && per_xcd <= 32 && slices >= 8;
  const dim3 grid = k.wmajor ? dim3(8 * per_xcd, 1) : dim3(k.tiles_m * k.tiles_n, k.splitk);
  static const bool himg_ws = getenv("UR_HIMG_NOWS") == nullptr;        // wave-specialised form (8 compute + 2 loader waves); A/B: every wave loads for itself
  if (himg_ws && !k.gn_ab) {
    static ur::DeviceOnce ws_once;
    if (auto g2 = ws_once.first())
      hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_halo_img_ws_kernel<TH, TW, NIMG, BN, WM, WN, UR_TU_F16 != 0>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    UR_F16_SWITCH(k, hipLaunchKernelGGL((igemm_halo_img_ws_kernel<TH, TW, NIMG, BN, WM, WN, F16>), grid, dim3((NW + 2) * 64), lds, s, k));
  } else if (GN_OK && k.gn_ab) UR_F16_SWITCH(k, hipLaunchKernelGGL((igemm_halo_img_kernel<TH, TW, NIMG, BN, WM, WN, F16, GN_OK>), grid, dim3(NW * 64), lds, s, k));
  else UR_F16_SWITCH(k, hipLaunchKernelGGL((igemm_halo_img_kernel<TH, TW, NIMG, BN, WM, WN, F16, false>), grid, dim3(NW * 64), lds, s, k));
  if (k.splitk > 1) launch_splitk_reduce(k, s);
  return ur::check_launch("ur_conv2d_nhwc");
}

}  // namespace
