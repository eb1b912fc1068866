// HBM-bound normalisation kernels for gfx950: GroupNorm(+SiLU) / InstanceNorm over NHWC bf16,
// LayerNorm over the channel dim, row softmax.  16-byte vector accesses, fp32 math, fp64 global stats.
#include <cstdio>
#include <cstdlib>
#include <map>
#include <string>
#include "common.h"

namespace {

// ---- GroupNorm pass 1: per-(image, channel) sum / sum-of-squares -----------------------------------------------
// grid = (pixel chunks, N, channel slabs).  A slab is CVS <= 32 channel vectors (8 channels each); thread (r, v)
// owns vector v of the slab and pixel rows r, r+R, ... (R = 256/CVS), so a warp reads whole contiguous row pieces.
// Block partials are combined with LDS atomics and leave as ONE fp64 atomic per channel per block.
__global__ __launch_bounds__(256) void gn_stats_kernel(const uint16_t* __restrict__ x, double* __restrict__ stats,
                                                       int HW, int C, int pix_per_block, int c_off, int C_total, int CVS) {
  __shared__ float lds[2 * 256];
  const int n = blockIdx.y, t = threadIdx.x;
  const int CV = C >> 3, R = 256 / CVS;
  const int r = t / CVS, vl = t - r * CVS, v = blockIdx.z * CVS + vl;
  const int p_begin = blockIdx.x * pix_per_block, p_end = min(HW, p_begin + pix_per_block);
  for (int i = t; i < 2 * CVS * 8; i += 256) lds[i] = 0.f;
  __syncthreads();
  if (r < R && v < CV) {
    const uint16_t* xi = x + (long long)n * HW * C + v * 8;
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0}, q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
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
        unpack8(raw[u], f);
#pragma unroll
        for (int e = 0; e < 8; ++e) { s[e] += f[e]; q[e] += f[e] * f[e]; }
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      atomicAdd(&lds[vl * 8 + e], s[e]);
      atomicAdd(&lds[CVS * 8 + vl * 8 + e], q[e]);
    }
  }
  __syncthreads();
  double* st = stats + ((long long)n * C_total + c_off) * 2;
  for (int i = t; i < CVS * 8; i += 256) {
    const int c = blockIdx.z * CVS * 8 + i;
    if (c < C) {
      atomicAdd(&st[2 * c], (double)lds[i]);
      atomicAdd(&st[2 * c + 1], (double)lds[CVS * 8 + i]);
    }
  }
}

// ---- GroupNorm pass 2 (tiny): per-image group statistics -> per-channel affine (a, b); re-zeroes the sums ----
// grid = N, one block per image.  The fp64 sum buffer is zero at rest: this kernel consumes it and clears it, so
// no zero-fill launch is needed per call.
__global__ __launch_bounds__(256) void gn_finalize_kernel(double* __restrict__ stats, float* __restrict__ ab,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          int HW, int C, int G, float eps, const double* __restrict__ pre1,
                                                          const double* __restrict__ pre2, int C1) {
  extern __shared__ double gl[];  // gsum[G], gsq[G], then (float) mean[G], rstd[G]
  const int n = blockIdx.x, t = threadIdx.x, cpg = C / G;
  double* st = stats + (long long)n * C * 2;
  for (int g = t; g < 2 * G; g += 256) gl[g] = 0.0;
  __syncthreads();
  for (int c = t; c < C; c += 256) {                       // all threads, independent coalesced loads, LDS fp64 atomics
    // channel c comes from the producer's fused sums (pre1 / pre2, read-only) or from this call's own pass (st)
    const double* src = st + 2 * c;
    bool own = true;
    if (c < C1 && pre1) { src = pre1 + ((long long)n * C1 + c) * 2; own = false; }
    if (c >= C1 && pre2) { src = pre2 + ((long long)n * (C - C1) + (c - C1)) * 2; own = false; }
    const double s = src[0], q = src[1];
    atomicAdd(&gl[c / cpg], s);
    atomicAdd(&gl[G + c / cpg], q);
    if (own) { st[2 * c] = 0.0; st[2 * c + 1] = 0.0; }     // leave the internal sums zero for the next call
  }
  __syncthreads();
  float* mr = reinterpret_cast<float*>(gl + 2 * G);
  for (int g = t; g < G; g += 256) {
    const double cnt = (double)cpg * HW, mean = gl[g] / cnt;
    double var = gl[G + g] / cnt - mean * mean;
    var = var < 0.0 ? 0.0 : var;
    mr[g] = (float)mean;
    mr[G + g] = (float)(1.0 / sqrt(var + (double)eps));
  }
  __syncthreads();
  float* o = ab + (long long)n * C * 2;
  for (int c = t; c < C; c += 256) {
    const int g = c / cpg;
    const float ga = gamma ? gamma[c] : 1.f, be = beta ? beta[c] : 0.f, a = mr[G + g] * ga;
    o[c] = a;
    o[C + c] = be - mr[g] * a;
  }
}

// ---- GroupNorm pass 3: y = act(a[c]*x + b[c]) ------------------------------------------------------------------
// Same (pixel chunk, image, channel slab) decomposition; each thread keeps its 8 channels' (a, b) in registers.
// C = channels of THIS source tensor; it occupies channels [c_off, c_off+C) of the C_total-wide (virtually
// concatenated) normalisation domain; y has row stride C_total.
__global__ __launch_bounds__(256) void gn_apply_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ y,
                                                       const float* __restrict__ ab, int HW, int C, int silu,
                                                       int pix_per_block, int c_off, int C_total, int CVS) {
  const int n = blockIdx.y, t = threadIdx.x;
  const int CV = C >> 3, R = 256 / CVS;
  const int r = t / CVS, v = blockIdx.z * CVS + (t - r * CVS);
  if (r >= R || v >= CV) return;
  const float* abn = ab + (long long)n * C_total * 2 + c_off + v * 8;
  float a[8], b[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { a[e] = abn[e]; b[e] = abn[C_total + e]; }
  const int p_begin = blockIdx.x * pix_per_block, p_end = min(HW, p_begin + pix_per_block);
  const uint16_t* xi = x + (long long)n * HW * C + v * 8;
  uint16_t* yo = y + (long long)n * HW * C_total + c_off + v * 8;
  for (int p = p_begin + r; p < p_end; p += 4 * R) {        // 4 independent 16-byte loads in flight per thread
    uint4 raw[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int pp = min(p + u * R, p_end - 1);
      raw[u] = *reinterpret_cast<const uint4*>(xi + (long long)pp * C);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int pp = p + u * R;
      if (pp >= p_end) break;
      float f[8];
      unpack8(raw[u], f);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float o = f[e] * a[e] + b[e];
        f[e] = silu ? silu_f(o) : o;
      }
      *reinterpret_cast<uint4*>(yo + (long long)pp * C_total) = pack8(f);
    }
  }
}

// ---- GroupNorm apply with the finalize folded in (all sources carry producer-side sums) -------------------------------
// Each block rebuilds mean / rstd only for the groups its <=256-channel slab touches: cooperative fp64 LDS reduction over
// those groups' channel sums, then the same streaming pass as gn_apply_kernel.  Saves one launch per GroupNorm.
__global__ __launch_bounds__(256) void gn_apply_fused_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ y,
                                                             const double* __restrict__ pre1, const double* __restrict__ pre2,
                                                             int C1, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, int HW, int C, int silu,
                                                             int pix_per_block, int c_off, int C_total, int G, float eps, int CVS, double inv_cnt) {
  __shared__ double gs[2][264];
  __shared__ float gm[2][264];
  const int n = blockIdx.y, t = threadIdx.x;
  const int CV = C >> 3, R = 256 / CVS, cpg = C_total / G;
  // this thread's pixels / channel vector; the first batch of loads is issued BEFORE the statistics prologue (independent of it)
  const int r = t / CVS, v = blockIdx.z * CVS + (t - r * CVS);
  const bool worker = r < R && v < CV;
  const int p_begin = blockIdx.x * pix_per_block, p_end = min(HW, p_begin + pix_per_block);
  const uint16_t* xi = x + (long long)n * HW * C + (worker ? v : 0) * 8;
  uint4 raw[4];
  if (worker && p_begin + r < p_end) {
#pragma unroll
    for (int u = 0; u < 4; ++u) raw[u] = *reinterpret_cast<const uint4*>(xi + (long long)min(p_begin + r + u * R, p_end - 1) * C);
  }
  // affine parameters of this thread's 8 channels: also issued ahead of the prologue (one global round trip less on its tail)
  float ga[8], be[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = c_off + (worker ? v : 0) * 8 + e;
    ga[e] = gamma ? gamma[c] : 1.f;
    be[e] = beta ? beta[c] : 0.f;
  }
  const int cb = c_off + blockIdx.z * CVS * 8, ce = min(cb + CVS * 8, c_off + C);       // this slab in the C_total domain
  const int g_lo = cb / cpg, g_hi = (ce - 1) / cpg, ng = g_hi - g_lo + 1;
  for (int g = t; g < ng; g += 256) { gs[0][g] = 0.0; gs[1][g] = 0.0; }
  __syncthreads();
  const int C2 = C_total - C1;
  for (int c = g_lo * cpg + t; c < (g_hi + 1) * cpg; c += 256) {
    const double* src = c < C1 ? pre1 + ((long long)n * C1 + c) * 2 : pre2 + ((long long)n * C2 + (c - C1)) * 2;
    atomicAdd(&gs[0][c / cpg - g_lo], src[0]);
    atomicAdd(&gs[1][c / cpg - g_lo], src[1]);
  }
  __syncthreads();
  for (int g = t; g < ng; g += 256) {
    // mean / variance in fp64 (the subtraction cancels), 1/sqrt in fp32: inv_cnt comes from the host, so no fp64 divide / sqrt
    const double mean = gs[0][g] * inv_cnt;
    const double var = fma(gs[1][g], inv_cnt, -mean * mean);
    gm[0][g] = (float)mean;
    gm[1][g] = rsqrtf(fmaxf((float)var, 0.f) + eps);
  }
  __syncthreads();
  if (!worker) return;
  float a[8], b[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = c_off + v * 8 + e, g = c / cpg - g_lo;
    a[e] = gm[1][g] * ga[e];
    b[e] = be[e] - gm[0][g] * a[e];
  }
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
      unpack8(raw[u], f);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float o = f[e] * a[e] + b[e];
        f[e] = silu ? silu_f(o) : o;
      }
      *reinterpret_cast<uint4*>(yo + (long long)pp * C_total) = pack8(f);
    }
  }
}

// ---- LayerNorm over C: one wave per row, R rows per wave in flight, exact two-pass variance in registers --------
template <int VPL, int R>  // vectors (8 elems) per lane, rows batched per wave
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
      unpack8(raw[r][j], f[j]);
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
        *reinterpret_cast<uint4*>(y + (row0 + r) * C + v * 8) = pack8(o);
      }
    }
  }
}

// ---- row softmax fp32 -> bf16 (one block per row; the row stays L2-resident across the 3 passes) ----
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
  for (int i = t; i < cols; i += 256) pr[i] = f2bf(__expf(sr[i] - m) * inv);
  for (int i = cols + t; i < ldp; i += 256) pr[i] = 0;
}

}  // namespace

namespace ur {
int gn_stats_launch(const void* x, double* stats, int N, int HW, int C, hipStream_t s) {
  const int cv = C / 8;
  int cvs = cv < 32 ? cv : 32;
  while (cv % cvs) --cvs;
  const int slabs = cv / cvs, R = 256 / cvs;
  long long want = std::max<long long>(1, 2048 / ((long long)N * slabs));
  int chunks = (int)std::min<long long>(want, std::max(1, HW / (8 * R)));
  const int ppb = (HW + chunks - 1) / chunks;
  chunks = (HW + ppb - 1) / ppb;
  hipLaunchKernelGGL(gn_stats_kernel, dim3(chunks, N, slabs), dim3(256), 0, s, (const uint16_t*)x, stats, HW, C, ppb, 0, C, cvs);
  return check_launch("gn_stats");
}
}  // namespace ur

extern "C" {

size_t ur_groupnorm_ws_bytes(int N, int C) { return (size_t)N * C * 2 * sizeof(double); }
size_t ur_groupnorm_ab_bytes(int N, int C) { return (size_t)N * C * 2 * sizeof(float); }

int ur_groupnorm_nhwc(const void* x, const void* x2, void* y, const float* gamma, const float* beta, int N, int HW,
                      int C1, int C2, int G, float eps, int silu, void* ws, float* ab, const double* pre1, const double* pre2,
                      ur_stream_t stream) {
  UR_REQUIRE(x && y && ws && ab, "null pointer");
  const int C = C1 + (x2 ? C2 : 0);
  UR_REQUIRE(C1 % 8 == 0 && (!x2 || C2 % 8 == 0) && G > 0 && C % G == 0 && N > 0 && HW > 0, "C%8, C%G");
  UR_REQUIRE((size_t)C * 8 <= 64 * 1024, "C too large");
  hipStream_t s = (hipStream_t)stream;
  const double bytes = 2.0 * N * HW * (double)C;
  const char* fam = "groupnorm";
  static const bool prof_shapes = getenv("UR_PROF_SHAPES") != nullptr;
  if (prof_shapes) {                                     // per-shape families for tools/prof_shapes.py
    static std::map<std::string, int> interned;
    char buf[96];
    snprintf(buf, sizeof buf, "gn N%d HW%d C%d+%d pre%d", N, HW, C1, x2 ? C2 : 0, (pre1 ? 1 : 0) + (x2 && pre2 ? 1 : 0));
    fam = interned.emplace(buf, 0).first->first.c_str();
  }
  ur::ProfScope prof(fam, 0.0, 3.0 * bytes, s);
  // ws = [N][C][2] fp64 sums, ZERO AT REST (the caller zero-fills once, the finalize kernel re-zeroes what it read);
  // ab = [N][2][C] fp32 affine table (plain scratch; separate so it can never alias another call's sums)
  double* stats = (double*)ws;
  const uint16_t* src[2] = {(const uint16_t*)x, (const uint16_t*)x2};
  const int cs[2] = {C1, x2 ? C2 : 0}, off[2] = {0, C1};
  int chunks[2], ppb[2], cvs[2], slabs[2];
  for (int i = 0; i < 2; ++i) {
    if (cs[i] <= 0) continue;
    const int cv = cs[i] / 8;
    cvs[i] = cv < 32 ? cv : 32;                            // channel vectors per slab (<= 256 channels):
    while (cv % cvs[i]) --cvs[i];                          // the largest divisor of CV that is <= 32 (no ragged slab)
    slabs[i] = (cv + cvs[i] - 1) / cvs[i];
    const int R = 256 / cvs[i];
    // aim for >= ~2048 blocks (8 per CU) but keep >= 4 pixel rows per thread when the tensor is big enough
    long long want = std::max<long long>(1, 2048 / ((long long)N * slabs[i]));
    static const int ppt = getenv("UR_GN_PPT") ? atoi(getenv("UR_GN_PPT")) : 8;
    static const int wantb = getenv("UR_GN_BLOCKS") ? atoi(getenv("UR_GN_BLOCKS")) : 2048;
    want = std::max<long long>(1, wantb / ((long long)N * slabs[i]));
    // (8x8 maps: 2 pixel rows per thread - 40 workgroups of 8-deep loops were pure latency)
    chunks[i] = (int)std::min<long long>(want, std::max(1, HW / ((HW <= 64 ? std::min(ppt, 2) : ppt) * R)));
    ppb[i] = (HW + chunks[i] - 1) / chunks[i];
    chunks[i] = (HW + ppb[i] - 1) / ppb[i];
    const double* pre = i == 0 ? pre1 : pre2;
    if (!pre)
      hipLaunchKernelGGL(gn_stats_kernel, dim3(chunks[i], N, slabs[i]), dim3(256), 0, s, src[i], stats, HW, cs[i], ppb[i], off[i],
                         C, cvs[i]);
  }
  const bool all_pre = pre1 && (!x2 || pre2);
  const int cpg = C / G;
  if (all_pre && (256 / cpg + 2) <= 264) {
    for (int i = 0; i < 2; ++i)
      if (cs[i] > 0)
        hipLaunchKernelGGL(gn_apply_fused_kernel, dim3(chunks[i], N, slabs[i]), dim3(256), 0, s, src[i], (uint16_t*)y, pre1,
                           x2 ? pre2 : nullptr, C1, gamma, beta, HW, cs[i], silu, ppb[i], off[i], C, G, eps, cvs[i], 1.0 / ((double)cpg * HW));
    return ur::check_launch("ur_groupnorm_nhwc");
  }
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(N), dim3(256), (size_t)G * (2 * sizeof(double) + 2 * sizeof(float)), s, stats, ab,
                     gamma, beta, HW, C, G, eps, pre1, x2 ? pre2 : nullptr, C1);
  for (int i = 0; i < 2; ++i)
    if (cs[i] > 0)
      hipLaunchKernelGGL(gn_apply_kernel, dim3(chunks[i], N, slabs[i]), dim3(256), 0, s, src[i], (uint16_t*)y, ab, HW, cs[i], silu,
                         ppb[i], off[i], C, cvs[i]);
  return ur::check_launch("ur_groupnorm_nhwc");
}

int ur_layernorm_rows(const void* x, void* y, const float* gamma, const float* beta, long long rows, int C, float eps,
                      ur_stream_t stream) {
  UR_REQUIRE(x && y && rows > 0, "null pointer / empty");
  UR_REQUIRE(C % 8 == 0 && C <= 2048, "C%8 and C<=2048");
  hipStream_t s = (hipStream_t)stream;
  ur::ProfScope prof("layernorm", 0.0, 4.0 * rows * (double)C, s);
  const int vpl = (C / 8 + 63) / 64;
  constexpr int R = 4;
  dim3 grid((unsigned)((rows + 4 * R - 1) / (4 * R))), block(256);
  const uint16_t* xi = (const uint16_t*)x;
  uint16_t* yo = (uint16_t*)y;
  switch (vpl) {
    case 1: hipLaunchKernelGGL((ln_rows_kernel<1, R>), grid, block, 0, s, xi, yo, gamma, beta, rows, C, eps); break;
    case 2: hipLaunchKernelGGL((ln_rows_kernel<2, R>), grid, block, 0, s, xi, yo, gamma, beta, rows, C, eps); break;
    case 3: hipLaunchKernelGGL((ln_rows_kernel<3, R>), grid, block, 0, s, xi, yo, gamma, beta, rows, C, eps); break;
    default: hipLaunchKernelGGL((ln_rows_kernel<4, R>), grid, block, 0, s, xi, yo, gamma, beta, rows, C, eps); break;
  }
  return ur::check_launch("ur_layernorm_rows");
}

int ur_softmax_rows_f32(const float* sm, void* p, long long rows, int cols, int ldp, ur_stream_t stream) {
  UR_REQUIRE(sm && p && rows > 0 && cols > 0 && ldp >= cols, "bad args");
  hipStream_t s = (hipStream_t)stream;
  ur::ProfScope prof("softmax_rows", 0.0, rows * (double)cols * 6.0, s);
  hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)rows), dim3(256), 0, s, sm, (uint16_t*)p, cols, ldp);
  return ur::check_launch("ur_softmax_rows_f32");
}

}  // extern "C"
