// HBM-bound normalisation kernels for gfx950: GroupNorm(+SiLU) / InstanceNorm over NHWC bf16,
// LayerNorm over the channel dim, row softmax.  16-byte vector accesses, fp32 math, fp64 global stats.
#include "common.h"

namespace {

// ---- GroupNorm pass 1: per-(image, channel) sum / sum-of-squares -----------------------------------
// grid = (chunks, N).  Thread (r, v): channel vector v (8 channels), pixel rows r, r+R, ...  Block partials
// are combined in LDS and leave as one fp64 atomic per channel per block.
__global__ __launch_bounds__(256) void gn_stats_kernel(const uint16_t* __restrict__ x, double* __restrict__ stats,
                                                       int HW, int C, int pix_per_block, int c_off, int C_total) {
  extern __shared__ float lds[];  // [2][C] when CV <= 256
  const int n = blockIdx.y, t = threadIdx.x;
  const int CV = C >> 3;
  const int p_begin = blockIdx.x * pix_per_block, p_end = min(HW, p_begin + pix_per_block);
  const uint16_t* xi = x + (long long)n * HW * C;
  double* st = stats + ((long long)n * C_total + c_off) * 2;
  if (CV <= 256) {
    const int R = 256 / CV;
    for (int i = t; i < 2 * C; i += 256) lds[i] = 0.f;
    __syncthreads();
    const int r = t / CV, v = t - r * CV;
    if (r < R) {
      float s[8] = {0, 0, 0, 0, 0, 0, 0, 0}, q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int p = p_begin + r; p < p_end; p += R) {
        uint4 raw = *reinterpret_cast<const uint4*>(xi + (long long)p * C + v * 8);
        float f[8];
        unpack8(raw, f);
#pragma unroll
        for (int e = 0; e < 8; ++e) { s[e] += f[e]; q[e] += f[e] * f[e]; }
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        atomicAdd(&lds[v * 8 + e], s[e]);
        atomicAdd(&lds[C + v * 8 + e], q[e]);
      }
    }
    __syncthreads();
    for (int c = t; c < C; c += 256) {
      atomicAdd(&st[2 * c], (double)lds[c]);
      atomicAdd(&st[2 * c + 1], (double)lds[C + c]);
    }
  } else {
    for (int v = t; v < CV; v += 256) {
      float s[8] = {0, 0, 0, 0, 0, 0, 0, 0}, q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int p = p_begin; p < p_end; ++p) {
        uint4 raw = *reinterpret_cast<const uint4*>(xi + (long long)p * C + v * 8);
        float f[8];
        unpack8(raw, f);
#pragma unroll
        for (int e = 0; e < 8; ++e) { s[e] += f[e]; q[e] += f[e] * f[e]; }
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        atomicAdd(&st[2 * (v * 8 + e)], (double)s[e]);
        atomicAdd(&st[2 * (v * 8 + e) + 1], (double)q[e]);
      }
    }
  }
}

// ---- GroupNorm pass 2: y = act(a[c]*x + b[c]); a/b rebuilt per block from the fp64 channel sums ----
__global__ __launch_bounds__(256) void gn_apply_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ y,
                                                       const double* __restrict__ stats, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, int HW, int C, int G, float eps,
                                                       int silu, int pix_per_block, int c_off, int C_total) {
  // C = channels of THIS source tensor; it occupies channels [c_off, c_off+C) of the C_total-wide (virtually
  // concatenated) normalisation domain; y has row stride C_total.
  extern __shared__ float lds[];  // a[C], b[C]
  const int n = blockIdx.y, t = threadIdx.x;
  const int cpg = C_total / G;
  const double* st = stats + (long long)n * C_total * 2;
  for (int cl = t; cl < C; cl += 256) {
    const int c = c_off + cl;
    const int g0 = (c / cpg) * cpg;
    double s = 0.0, q = 0.0;
    for (int j = 0; j < cpg; ++j) { s += st[2 * (g0 + j)]; q += st[2 * (g0 + j) + 1]; }
    const double cnt = (double)cpg * HW;
    const double mean = s / cnt;
    double var = q / cnt - mean * mean;
    var = var < 0.0 ? 0.0 : var;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float ga = gamma ? gamma[c] : 1.f, be = beta ? beta[c] : 0.f;
    lds[cl] = rstd * ga;
    lds[C + cl] = be - (float)mean * rstd * ga;
  }
  __syncthreads();
  const int CV = C >> 3;
  const int p_begin = blockIdx.x * pix_per_block, p_end = min(HW, p_begin + pix_per_block);
  const long long base = (long long)n * HW * C;
  const long long obase = (long long)n * HW * C_total + c_off;
  const long long v_begin = (long long)p_begin * CV, v_end = (long long)p_end * CV;
  for (long long i = v_begin + t; i < v_end; i += 256) {
    const int v = (int)(i % CV);
    const long long pix = i / CV;
    uint4 raw = *reinterpret_cast<const uint4*>(x + base + i * 8);
    float f[8];
    unpack8(raw, f);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float o = f[e] * lds[v * 8 + e] + lds[C + v * 8 + e];
      f[e] = silu ? silu_f(o) : o;
    }
    *reinterpret_cast<uint4*>(y + obase + pix * C_total + v * 8) = pack8(f);
  }
}

// ---- LayerNorm over C: one wave per row, row held in registers, exact two-pass variance -----------
template <int VPL>  // vectors (8 elems) per lane
__global__ __launch_bounds__(256) void ln_rows_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ y,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      long long rows, int C, float eps) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int CV = C >> 3;
  float f[VPL][8];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < VPL; ++j) {
    const int v = lane + j * 64;
    if (v < CV) {
      uint4 raw = *reinterpret_cast<const uint4*>(x + row * C + v * 8);
      unpack8(raw, f[j]);
#pragma unroll
      for (int e = 0; e < 8; ++e) s += f[j][e];
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) f[j][e] = 0.f;
    }
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
      for (int e = 0; e < 8; ++e) {
        const int c = v * 8 + e;
        o[e] = (f[j][e] - mean) * rstd * (gamma ? gamma[c] : 1.f) + (beta ? beta[c] : 0.f);
      }
      *reinterpret_cast<uint4*>(y + row * C + v * 8) = pack8(o);
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

extern "C" {

size_t ur_groupnorm_ws_bytes(int N, int C) { return (size_t)N * C * 2 * sizeof(double); }

int ur_groupnorm_nhwc(const void* x, const void* x2, void* y, const float* gamma, const float* beta, int N, int HW,
                      int C1, int C2, int G, float eps, int silu, void* ws, ur_stream_t stream) {
  UR_REQUIRE(x && y && ws, "null pointer");
  const int C = C1 + (x2 ? C2 : 0);
  UR_REQUIRE(C1 % 8 == 0 && (!x2 || C2 % 8 == 0) && G > 0 && C % G == 0 && N > 0 && HW > 0, "C%8, C%G");
  UR_REQUIRE((size_t)C * 8 <= 64 * 1024, "C too large");
  hipStream_t s = (hipStream_t)stream;
  const double bytes = 2.0 * N * HW * (double)C;
  ur::ProfScope prof("groupnorm", 0.0, 3.0 * bytes, s);
  ur::zero_async(ws, ur_groupnorm_ws_bytes(N, C), s);
  // enough blocks to fill 256 CUs several times over, at least ~32 pixels per block
  int chunks = (int)std::min<long long>(std::max<long long>(1, (2048 + N - 1) / N), (HW + 31) / 32);
  int ppb = (HW + chunks - 1) / chunks;
  chunks = (HW + ppb - 1) / ppb;
  const uint16_t* src[2] = {(const uint16_t*)x, (const uint16_t*)x2};
  const int cs[2] = {C1, x2 ? C2 : 0}, off[2] = {0, C1};
  for (int i = 0; i < 2; ++i)
    if (cs[i] > 0)
      hipLaunchKernelGGL(gn_stats_kernel, dim3(chunks, N), dim3(256), (size_t)2 * cs[i] * sizeof(float), s, src[i],
                         (double*)ws, HW, cs[i], ppb, off[i], C);
  for (int i = 0; i < 2; ++i)
    if (cs[i] > 0)
      hipLaunchKernelGGL(gn_apply_kernel, dim3(chunks, N), dim3(256), (size_t)2 * cs[i] * sizeof(float), s, src[i],
                         (uint16_t*)y, (const double*)ws, gamma, beta, HW, cs[i], G, eps, silu, ppb, off[i], C);
  return ur::check_launch("ur_groupnorm_nhwc");
}

int ur_layernorm_rows(const void* x, void* y, const float* gamma, const float* beta, long long rows, int C, float eps,
                      ur_stream_t stream) {
  UR_REQUIRE(x && y && rows > 0, "null pointer / empty");
  UR_REQUIRE(C % 8 == 0 && C <= 2048, "C%8 and C<=2048");
  hipStream_t s = (hipStream_t)stream;
  ur::ProfScope prof("layernorm", 0.0, 4.0 * rows * (double)C, s);
  const int vpl = (C / 8 + 63) / 64;
  dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  const uint16_t* xi = (const uint16_t*)x;
  uint16_t* yo = (uint16_t*)y;
  switch (vpl) {
    case 1: hipLaunchKernelGGL(ln_rows_kernel<1>, grid, block, 0, s, xi, yo, gamma, beta, rows, C, eps); break;
    case 2: hipLaunchKernelGGL(ln_rows_kernel<2>, grid, block, 0, s, xi, yo, gamma, beta, rows, C, eps); break;
    case 3: hipLaunchKernelGGL(ln_rows_kernel<3>, grid, block, 0, s, xi, yo, gamma, beta, rows, C, eps); break;
    default: hipLaunchKernelGGL(ln_rows_kernel<4>, grid, block, 0, s, xi, yo, gamma, beta, rows, C, eps); break;
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
