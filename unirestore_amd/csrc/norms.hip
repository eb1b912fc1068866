// HBM-bound normalisation kernels for gfx950: GroupNorm(+SiLU) / InstanceNorm over NHWC bf16,
// LayerNorm over the channel dim, row softmax.  16-byte vector accesses, fp32 math, fp64 global stats.
#include <cstdio>
#include <cstdlib>
#include <map>
#include <string>
#include "common.h"

namespace {

// ---- GroupNorm / InstanceNorm / average-pool statistics: DETERMINISTIC partial planes ---------------------------------
// part[N][P][C][2] fp32: partial (sum, sum of squares) of image n, pixel chunk p, channel c.  Every entry is written by
// exactly one workgroup with plain stores (no atomics anywhere), the finalize kernel adds the P partials of a group in a
// fixed order in fp64 - two replays of the same graph are bit-identical.  Conv / GEMM epilogues write the same layout
// (ur_conv_desc.gn_part), so a GroupNorm whose producer left partials needs no statistics pass at all.
//
// pass 1 (only for tensors without producer-side partials): grid = (pixel chunks, N, channel slabs).  A slab is
// CVS <= 32 channel vectors (8 channels each); thread (r, v) owns vector v of the slab and pixel rows r, r+R, ...
template <bool F16>
__global__ __launch_bounds__(256) void gn_stats_kernel(const uint16_t* __restrict__ x, float* __restrict__ part,
                                                       int HW, int C, int pix_per_block, int CVS) {
  __shared__ float lds[256 * 16];
  const int n = blockIdx.y, t = threadIdx.x, P = gridDim.x;
  const int CV = C >> 3, R = 256 / CVS;
  const int r = t / CVS, vl = t - r * CVS, v = blockIdx.z * CVS + vl;
  const int p_begin = blockIdx.x * pix_per_block, p_end = min(HW, p_begin + pix_per_block);
  float s[8] = {0, 0, 0, 0, 0, 0, 0, 0}, q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (r < R && v < CV) {
    const uint16_t* xi = x + (long long)n * HW * C + v * 8;
    for (int p = p_begin + r; p < p_end; p += 4 * R) {
      uint4 raw[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int pp = min(p + u * R, p_end - 1);
        raw[u] = *reinterpret_cast<const uint4*>(xi + (long long)pp * C);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (p + u * R >= p_end) break;
        float f[8];
        unpack8t<F16>(raw[u], f);
#pragma unroll
        for (int e = 0; e < 8; ++e) { s[e] += f[e]; q[e] += f[e] * f[e]; }
      }
    }
  }
  // row lanes meet in LDS and are added in row order by one thread per channel (fixed order)
  if (r < R) {
#pragma unroll
    for (int e = 0; e < 8; ++e) { lds[(r * 2) * CVS * 8 + vl * 8 + e] = s[e]; lds[(r * 2 + 1) * CVS * 8 + vl * 8 + e] = q[e]; }
  }
  __syncthreads();
  for (int i = t; i < CVS * 8; i += 256) {
    const int c = blockIdx.z * CVS * 8 + i;
    if (c >= C) continue;
    float a = 0.f, b = 0.f;
    for (int rr = 0; rr < R; ++rr) { a += lds[(rr * 2) * CVS * 8 + i]; b += lds[(rr * 2 + 1) * CVS * 8 + i]; }
    *reinterpret_cast<float2*>(part + (((long long)n * P + blockIdx.x) * C + c) * 2) = make_float2(a, b);
  }
}

// ---- finalize: partial planes of one or two (virtually concatenated) sources -> per-(image, channel) affine (a, b) ----
// grid = (G, N): one workgroup per (group, image) adds the group's cpg x P partial pairs in fp64 - per-thread in a fixed
// strided order, then a fixed LDS tree - and writes a = rstd*gamma, b = beta - mean*a for its channels ([N][2][C]).
// mean_out (optional, [N][G]): the group mean itself (G == C: AdaptiveAvgPool2d(1) of the tensor).
__global__ void gn_finalize_kernel(const float* __restrict__ p1, int P1, int C1, const float* __restrict__ p2, int P2, int C2,
                                   const float* __restrict__ gamma, const float* __restrict__ beta, int G, float eps,
                                   double inv_cnt, float* __restrict__ ab, float* __restrict__ mean_out) {
  __shared__ double red[2][256];
  const int g = blockIdx.x, n = blockIdx.y, t = threadIdx.x, NT = blockDim.x;
  const int C = C1 + C2, cpg = C / G, c_lo = g * cpg, Pmax = max(P1, P2);
  // gamma / beta of the lane's first channel: issued next to the plane loads (behind the reduction they were a second,
  // serialised memory latency in a kernel that is nothing but a latency chain)
  float ga0 = 1.f, be0 = 0.f;
  if (ab && t < cpg) { ga0 = gamma ? gamma[c_lo + t] : 1.f; be0 = beta ? beta[c_lo + t] : 0.f; }
  double s = 0.0, q = 0.0;
  for (int idx = t; idx < cpg * Pmax; idx += NT) {
    const int p = idx / cpg, c = c_lo + (idx - p * cpg);
    const bool first = c < C1;
    const int P = first ? P1 : P2;
    if (p < P) {
      const float* src = first ? p1 + (((long long)n * P1 + p) * C1 + c) * 2 : p2 + (((long long)n * P2 + p) * C2 + (c - C1)) * 2;
      const float2 v = *reinterpret_cast<const float2*>(src);
      s += (double)v.x; q += (double)v.y;
    }
  }
  if (NT == 64) {            // one wave: fixed butterfly through the lanes, no LDS round trips (every lane ends with the total)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
  } else {
    red[0][t] = s; red[1][t] = q;
    __syncthreads();
    for (int o = NT >> 1; o > 0; o >>= 1) {
      if (t < o) { red[0][t] += red[0][t + o]; red[1][t] += red[1][t + o]; }
      __syncthreads();
    }
  }
  const double S = NT == 64 ? s : red[0][0], Q = NT == 64 ? q : red[1][0];
  const double mean = S * inv_cnt;
  const double var = fma(Q, inv_cnt, -mean * mean);
  const float rstd = rsqrtf(fmaxf((float)var, 0.f) + eps);
  if (mean_out && t == 0) mean_out[(long long)n * G + g] = (float)mean;
  if (ab)
    for (int i = t; i < cpg; i += NT) {
      const int c = c_lo + i;
      const float a = rstd * (i == t ? ga0 : (gamma ? gamma[c] : 1.f));
      ab[((long long)n * 2) * C + c] = a;
      ab[((long long)n * 2 + 1) * C + c] = (i == t ? be0 : (beta ? beta[c] : 0.f)) - (float)mean * a;
    }
}

// ---- apply: y = act(a[c]*x + b[c]) ------------------------------------------------------------------------------------
// Same (pixel chunk, image, channel slab) decomposition; each thread keeps its 8 channels' (a, b) in registers.
// C = channels of THIS source tensor; it occupies channels [c_off, c_off+C) of the C_total-wide (virtually
// concatenated) normalisation domain; y has row stride C_total.
template <bool F16>
__device__ __forceinline__ void gn_apply_body(const uint16_t* __restrict__ x, uint16_t* __restrict__ y,
                                              const float* __restrict__ ab, int HW, int C, int silu,
                                              int pix_per_block, int c_off, int C_total, int CVS, int bx, int bz) {
  const int n = blockIdx.y, t = threadIdx.x;
  const int CV = C >> 3, R = 256 / CVS;
  const int r = t / CVS, v = bz * CVS + (t - r * CVS);
  if (r >= R || v >= CV) return;
  const int p_begin = bx * pix_per_block, p_end = min(HW, p_begin + pix_per_block);
  const uint16_t* xi = x + (long long)n * HW * C + v * 8;
  uint4 raw[4];
  if (p_begin + r < p_end) {            // first batch of loads issued ahead of the coefficient loads
#pragma unroll
    for (int u = 0; u < 4; ++u) raw[u] = *reinterpret_cast<const uint4*>(xi + (long long)min(p_begin + r + u * R, p_end - 1) * C);
  }
  const float* abn = ab + (long long)n * C_total * 2 + c_off + v * 8;
  float a[8], b[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { a[e] = abn[e]; b[e] = abn[C_total + e]; }
  uint16_t* yo = y + (long long)n * HW * C_total + c_off + v * 8;
  for (int p = p_begin + r; p < p_end; p += 4 * R) {        // 4 independent 16-byte loads in flight per thread
    if (p != p_begin + r) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int pp = min(p + u * R, p_end - 1);
        raw[u] = *reinterpret_cast<const uint4*>(xi + (long long)pp * C);
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int pp = p + u * R;
      if (pp >= p_end) break;
      float f[8];
      unpack8t<F16>(raw[u], f);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float o = f[e] * a[e] + b[e];
        f[e] = silu ? silu_f(o) : o;
      }
      store16_wt(yo + (long long)pp * C_total, pack8t<F16>(f));
    }
  }
}
template <bool F16>
__global__ __launch_bounds__(256) void gn_apply_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ y,
                                                       const float* __restrict__ ab, int HW, int C, int silu,
                                                       int pix_per_block, int c_off, int C_total, int CVS) {
  gn_apply_body<F16>(x, y, ab, HW, C, silu, pix_per_block, c_off, C_total, CVS, blockIdx.x, blockIdx.z);
}
// both sources of a virtually concatenated input in ONE launch: blockIdx.z < slabs1 -> source 1, else source 2 (own geometry)
template <bool F16>
__global__ __launch_bounds__(256) void gn_apply2_kernel(const uint16_t* __restrict__ x1, const uint16_t* __restrict__ x2, uint16_t* __restrict__ y,
                                                        const float* __restrict__ ab, int HW, int C1, int C2, int silu, int ppb1, int ppb2,
                                                        int cvs1, int cvs2, int slabs1, int chunks1, int chunks2) {
  if ((int)blockIdx.z < slabs1) {
    if ((int)blockIdx.x < chunks1) gn_apply_body<F16>(x1, y, ab, HW, C1, silu, ppb1, 0, C1 + C2, cvs1, blockIdx.x, blockIdx.z);
  } else if ((int)blockIdx.x < chunks2) {
    gn_apply_body<F16>(x2, y, ab, HW, C2, silu, ppb2, C1, C1 + C2, cvs2, blockIdx.x, blockIdx.z - slabs1);
  }
}

// ---- LayerNorm over C: one wave per row, R rows per wave in flight, exact two-pass variance in registers --------
template <int VPL, int R, bool F16>  // vectors (8 elems) per lane, rows batched per wave; 16-bit type
__global__ __launch_bounds__(256) void ln_rows_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ y,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      long long rows, int C, float eps) {
  const int lane = threadIdx.x & 63;
  const long long row0 = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * R;
  if (row0 >= rows) return;
  const int CV = C >> 3;
  uint4 raw[R][VPL];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
      const int v = lane + j * 64;
      const long long row = min(row0 + r, rows - 1);
      raw[r][j] = (v < CV) ? *reinterpret_cast<const uint4*>(x + row * C + v * 8) : make_uint4(0, 0, 0, 0);
    }
  float ga[VPL][8], be[VPL][8];
#pragma unroll
  for (int j = 0; j < VPL; ++j) {
    const int v = lane + j * 64;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      ga[j][e] = (gamma && v < CV) ? gamma[v * 8 + e] : 1.f;
      be[j][e] = (beta && v < CV) ? beta[v * 8 + e] : 0.f;
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if (row0 + r >= rows) break;
    float f[VPL][8];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
      unpack8t<F16>(raw[r][j], f[j]);
#pragma unroll
      for (int e = 0; e < 8; ++e) s += f[j][e];
    }
    const float mean = wave_sum(s) / C;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < VPL; ++j)
      if (lane + j * 64 < CV) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { float d = f[j][e] - mean; q += d * d; }
      }
    const float rstd = rsqrtf(wave_sum(q) / C + eps);
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
      const int v = lane + j * 64;
      if (v < CV) {
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (f[j][e] - mean) * rstd * ga[j][e] + be[j][e];
        *reinterpret_cast<uint4*>(y + (row0 + r) * C + v * 8) = pack8t<F16>(o);
      }
    }
  }
}

// ---- row softmax fp32 -> bf16 (one block per row; the row stays L2-resident across the 3 passes) ----
template <bool F16>
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ s, uint16_t* __restrict__ p, int cols,
                                                           int ldp) {
  __shared__ float red[8];
  const long long row = blockIdx.x;
  const float* sr = s + row * cols;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  float m = -INFINITY;
  for (int i = t; i < cols; i += 256) m = fmaxf(m, sr[i]);
  m = wave_max(m);
  if (lane == 0) red[w] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float l = 0.f;
  for (int i = t; i < cols; i += 256) l += __expf(sr[i] - m);
  l = wave_sum(l);
  if (lane == 0) red[4 + w] = l;
  __syncthreads();
  const float inv = 1.f / (red[4] + red[5] + red[6] + red[7]);
  uint16_t* pr = p + row * ldp;
  for (int i = t; i < cols; i += 256) pr[i] = f2h16<F16>(__expf(sr[i] - m) * inv);
  for (int i = cols + t; i < ldp; i += 256) pr[i] = 0;
}

}  // namespace

// shared geometry of the statistics / apply passes: channel vectors per slab, slabs, pixel chunks, pixels per chunk
static void gn_geom(int N, int HW, int C, int ppt, int& cvs, int& slabs, int& chunks, int& ppb) {
  const int cv = C / 8;
  cvs = cv < 32 ? cv : 32;                               // channel vectors per slab (<= 256 channels):
  while (cv % cvs) --cvs;                                // the largest divisor of CV that is <= 32 (no ragged slab)
  slabs = cv / cvs;
  const int R = 256 / cvs;
  // aim for >= ~2048 blocks (8 per CU) but keep >= ppt pixel rows per thread when the tensor is big enough
  static const int wantb = getenv("UR_GN_BLOCKS") ? atoi(getenv("UR_GN_BLOCKS")) : 2048;
  const long long want = std::max<long long>(1, wantb / ((long long)N * slabs));
  // (8x8 maps: 2 pixel rows per thread - 40 workgroups of 8-deep loops were pure latency)
  chunks = (int)std::min<long long>(want, std::max(1, HW / ((HW <= 64 ? std::min(ppt, 2) : ppt) * R)));
  ppb = (HW + chunks - 1) / chunks;
  chunks = (HW + ppb - 1) / ppb;
}

namespace ur {
int gn_stats_parts(int N, int HW, int C) {
  int cvs, slabs, chunks, ppb;
  gn_geom(N, HW, C, 8, cvs, slabs, chunks, ppb);
  return chunks;
}
int gn_stats_launch(const void* x, float* part, int N, int HW, int C, int dtype, hipStream_t s) {
  int cvs, slabs, chunks, ppb;
  gn_geom(N, HW, C, 8, cvs, slabs, chunks, ppb);
  UR_DT_SWITCH(dtype, hipLaunchKernelGGL(gn_stats_kernel<F16>, dim3(chunks, N, slabs), dim3(256), 0, s, (const uint16_t*)x, part, HW, C, ppb, cvs));
  return check_launch("gn_stats");
}
}  // namespace ur

extern "C" {

int ur_groupnorm_stats_parts(int N, int HW, int C) { return (N > 0 && HW > 0 && C > 0 && C % 8 == 0) ? ur::gn_stats_parts(N, HW, C) : UR_E_INVALID; }
size_t ur_groupnorm_ws_bytes(int N, int HW, int C) { return (size_t)N * (size_t)std::max(ur_groupnorm_stats_parts(N, HW, C), 1) * C * 2 * sizeof(float); }
size_t ur_groupnorm_ab_bytes(int N, int C) { return (size_t)N * C * 2 * sizeof(float); }

int ur_groupnorm_stats(const void* x, float* part, int N, int HW, int C, int dtype, ur_stream_t stream) {
  UR_REQUIRE(x && part && N > 0 && HW > 0 && C > 0 && C % 8 == 0, "bad args");
  UR_REQUIRE_DT(dtype);
  hipStream_t s = (hipStream_t)stream;
  ur::ProfScope prof("groupnorm_stats", 0.0, 2.0 * N * HW * (double)C, s);
  return ur::gn_stats_launch(x, part, N, HW, C, dtype, s);
}
int ur_instnorm_stats(const void* x, float* part, int N, int HW, int C, int dtype, ur_stream_t stream) {
  return ur_groupnorm_stats(x, part, N, HW, C, dtype, stream);
}

int ur_groupnorm_finalize(const float* part1, int parts1, int C1, const float* part2, int parts2, int C2, const float* gamma,
                          const float* beta, int N, int HW, int G, float eps, float* ab, float* mean_out, ur_stream_t stream) {
  const int C = C1 + (part2 ? C2 : 0);
  UR_REQUIRE(part1 && parts1 > 0 && C1 > 0 && (!part2 || (parts2 > 0 && C2 > 0)), "bad partial planes");
  UR_REQUIRE(G > 0 && C % G == 0 && N > 0 && HW > 0 && (ab || mean_out), "bad args");
  hipStream_t s = (hipStream_t)stream;
  const int cpg = C / G;
  const long long entries = (long long)cpg * std::max(parts1, part2 ? parts2 : 0);
  ur::ProfScope prof("groupnorm_finalize", 0.0, 8.0 * N * ((double)parts1 * C1 + (part2 ? (double)parts2 * C2 : 0.0)), s);
  const int nt = entries <= 1024 ? 64 : 256;            // <= 16 entries per lane: one wave, shuffle reduction
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(G, N), dim3(nt), 0, s, part1, parts1, C1, part2, part2 ? parts2 : 0, part2 ? C2 : 0, gamma, beta, G,
                     eps, 1.0 / ((double)cpg * HW), ab, mean_out);
  return ur::check_launch("ur_groupnorm_finalize");
}

int ur_groupnorm_apply_act(const void* x, const void* x2, void* y, const float* ab, int N, int HW, int C1, int C2, int silu,
                           int dtype, ur_stream_t stream) {
  UR_REQUIRE(x && y && ab && N > 0 && HW > 0, "null pointer / empty");
  UR_REQUIRE_DT(dtype);
  const int C = C1 + (x2 ? C2 : 0);
  UR_REQUIRE(C1 % 8 == 0 && (!x2 || C2 % 8 == 0), "C%8");
  hipStream_t s = (hipStream_t)stream;
  const char* fam = "groupnorm";
  static const bool prof_shapes = getenv("UR_PROF_SHAPES") != nullptr;
  if (prof_shapes) {                                     // per-shape families for offline analysis
    static std::map<std::string, int> interned;
    char buf[96];
    snprintf(buf, sizeof buf, "gn N%d HW%d C%d+%d", N, HW, C1, x2 ? C2 : 0);
    fam = interned.emplace(buf, 0).first->first.c_str();
  }
  ur::ProfScope prof(fam, 0.0, 4.0 * N * HW * (double)C, s);
  const uint16_t* src[2] = {(const uint16_t*)x, (const uint16_t*)x2};
  const int cs[2] = {C1, x2 ? C2 : 0}, off[2] = {0, C1};
  static const int ppt = getenv("UR_GN_PPT") ? atoi(getenv("UR_GN_PPT")) : 2;      // (8 -> 2 pixel rows per thread: -0.8 ms per forward, same-box A/B)
  static const bool one_launch = getenv("UR_GN_TWO_LAUNCHES") == nullptr;
  if (one_launch && cs[1] > 0) {                          // virtual concat: one launch for both sources
    int cvs1, slabs1, chunks1, ppb1, cvs2, slabs2, chunks2, ppb2;
    gn_geom(N, HW, cs[0], ppt, cvs1, slabs1, chunks1, ppb1);
    gn_geom(N, HW, cs[1], ppt, cvs2, slabs2, chunks2, ppb2);
    UR_DT_SWITCH(dtype, hipLaunchKernelGGL(gn_apply2_kernel<F16>, dim3(std::max(chunks1, chunks2), N, slabs1 + slabs2), dim3(256), 0, s, src[0], src[1],
                                           (uint16_t*)y, ab, HW, cs[0], cs[1], silu, ppb1, ppb2, cvs1, cvs2, slabs1, chunks1, chunks2));
    return ur::check_launch("ur_groupnorm_apply_act");
  }
  for (int i = 0; i < 2; ++i) {
    if (cs[i] <= 0) continue;
    int cvs, slabs, chunks, ppb;
    gn_geom(N, HW, cs[i], ppt, cvs, slabs, chunks, ppb);
    UR_DT_SWITCH(dtype, hipLaunchKernelGGL(gn_apply_kernel<F16>, dim3(chunks, N, slabs), dim3(256), 0, s, src[i], (uint16_t*)y, ab, HW, cs[i], silu,
                                           ppb, off[i], C, cvs));
  }
  return ur::check_launch("ur_groupnorm_apply_act");
}

int ur_groupnorm_nhwc(const void* x, const void* x2, void* y, const float* gamma, const float* beta, int N, int HW,
                      int C1, int C2, int G, float eps, int silu, float* ws, float* ab, const float* pre1, int parts1,
                      const float* pre2, int parts2, int dtype, ur_stream_t stream) {
  UR_REQUIRE(x && y && ab, "null pointer");
  UR_REQUIRE((pre1 && (!x2 || pre2)) || ws, "a source without producer-side partials needs the ws scratch");
  // statistics pass only for sources whose producer left no partial plane (ws holds x's plane, then x2's)
  if (!pre1) {
    parts1 = ur::gn_stats_parts(N, HW, C1);
    int rc = ur_groupnorm_stats(x, ws, N, HW, C1, dtype, stream);
    if (rc != UR_OK) return rc;
    pre1 = ws;
    ws += (size_t)N * parts1 * C1 * 2;
  }
  if (x2 && !pre2) {
    parts2 = ur::gn_stats_parts(N, HW, C2);
    int rc = ur_groupnorm_stats(x2, ws, N, HW, C2, dtype, stream);
    if (rc != UR_OK) return rc;
    pre2 = ws;
  }
  int rc = ur_groupnorm_finalize(pre1, parts1, C1, x2 ? pre2 : nullptr, parts2, C2, gamma, beta, N, HW, G, eps, ab, nullptr, stream);
  if (rc != UR_OK) return rc;
  return ur_groupnorm_apply_act(x, x2, y, ab, N, HW, C1, C2, silu, dtype, stream);
}

/* mean over HW -> fp32 [N][C] (nn.AdaptiveAvgPool2d(1)): the statistics pass + a finalize with one group per channel */
int ur_avgpool_hw(const void* x, float* out, int N, int HW, int C, float* ws, int dtype, ur_stream_t stream) {
  UR_REQUIRE(x && out && ws && C % 8 == 0, "bad args");
  int rc = ur_groupnorm_stats(x, ws, N, HW, C, dtype, stream);
  if (rc != UR_OK) return rc;
  return ur_groupnorm_finalize(ws, ur::gn_stats_parts(N, HW, C), C, nullptr, 0, 0, nullptr, nullptr, N, HW, C, 0.f, nullptr, out, stream);
}

int ur_layernorm_rows(const void* x, void* y, const float* gamma, const float* beta, long long rows, int C, float eps,
                      int dtype, ur_stream_t stream) {
  UR_REQUIRE(x && y && rows > 0, "null pointer / empty");
  UR_REQUIRE(C % 8 == 0 && C <= 2048, "C%8 and C<=2048");
  UR_REQUIRE_DT(dtype);
  hipStream_t s = (hipStream_t)stream;
  ur::ProfScope prof("layernorm", 0.0, 4.0 * rows * (double)C, s);
  const int vpl = (C / 8 + 63) / 64;
  constexpr int R = 4;
  dim3 grid((unsigned)((rows + 4 * R - 1) / (4 * R))), block(256);
  const uint16_t* xi = (const uint16_t*)x;
  uint16_t* yo = (uint16_t*)y;
  switch (vpl) {
    case 1: UR_DT_SWITCH(dtype, hipLaunchKernelGGL((ln_rows_kernel<1, R, F16>), grid, block, 0, s, xi, yo, gamma, beta, rows, C, eps)); break;
    case 2: UR_DT_SWITCH(dtype, hipLaunchKernelGGL((ln_rows_kernel<2, R, F16>), grid, block, 0, s, xi, yo, gamma, beta, rows, C, eps)); break;
    case 3: UR_DT_SWITCH(dtype, hipLaunchKernelGGL((ln_rows_kernel<3, R, F16>), grid, block, 0, s, xi, yo, gamma, beta, rows, C, eps)); break;
    default: UR_DT_SWITCH(dtype, hipLaunchKernelGGL((ln_rows_kernel<4, R, F16>), grid, block, 0, s, xi, yo, gamma, beta, rows, C, eps)); break;
  }
  return ur::check_launch("ur_layernorm_rows");
}

int ur_softmax_rows_f32(const float* sm, void* p, long long rows, int cols, int ldp, int dtype, ur_stream_t stream) {
  UR_REQUIRE(sm && p && rows > 0 && cols > 0 && ldp >= cols, "bad args");
  hipStream_t s = (hipStream_t)stream;
  ur::ProfScope prof("softmax_rows", 0.0, rows * (double)cols * 6.0, s);
  UR_DT_SWITCH(dtype, hipLaunchKernelGGL(softmax_rows_kernel<F16>, dim3((unsigned)rows), dim3(256), 0, s, sm, (uint16_t*)p, cols, ldp));
  return ur::check_launch("ur_softmax_rows_f32");
}

}  // extern "C"
