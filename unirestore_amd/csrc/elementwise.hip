// HBM-bound stencil / reduction / elementwise kernels + tiny fp32 vector math for gfx950.
#include "common.h"

namespace {

// ---- depthwise 3x3 (pad 1) + bias (+ SimpleGate): one thread = 8 channels of one output pixel ------
template <bool F16>
__global__ __launch_bounds__(256) void dwconv3x3_kernel(const uint16_t* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, uint16_t* __restrict__ y, int N,
                                                        int H, int W, int C, int gate) {
  const int Cout = gate ? C / 2 : C;
  const int CV = Cout >> 3;
  const long long total = (long long)N * H * W * CV;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int v = (int)(i % CV);
    long long pix = i / CV;
    const int ow = (int)(pix % W);
    const int oh = (int)((pix / W) % H);
    const int n = (int)(pix / ((long long)W * H));
    float acc[2][8];
    const int halves = gate ? 2 : 1;
    for (int hf = 0; hf < halves; ++hf) {
      const int c0 = v * 8 + hf * Cout;
      float4 b0 = *reinterpret_cast<const float4*>(bias + c0), b1 = *reinterpret_cast<const float4*>(bias + c0 + 4);
      float* a = acc[hf];
      a[0] = b0.x; a[1] = b0.y; a[2] = b0.z; a[3] = b0.w; a[4] = b1.x; a[5] = b1.y; a[6] = b1.z; a[7] = b1.w;
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        const int ih = oh + dy - 1;
        if ((unsigned)ih >= (unsigned)H) continue;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const int iw = ow + dx - 1;
          if ((unsigned)iw >= (unsigned)W) continue;
          uint4 raw = *reinterpret_cast<const uint4*>(x + (((long long)n * H + ih) * W + iw) * C + c0);
          float f[8];
          unpack8t<F16>(raw, f);
          const float* wp = w + (dy * 3 + dx) * C + c0;
          float4 w0 = *reinterpret_cast<const float4*>(wp), w1 = *reinterpret_cast<const float4*>(wp + 4);
          a[0] += f[0] * w0.x; a[1] += f[1] * w0.y; a[2] += f[2] * w0.z; a[3] += f[3] * w0.w;
          a[4] += f[4] * w1.x; a[5] += f[5] * w1.y; a[6] += f[6] * w1.z; a[7] += f[7] * w1.w;
        }
      }
    }
    if (gate) {
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[0][e] *= acc[1][e];
    }
    *reinterpret_cast<uint4*>(y + pix * Cout + v * 8) = pack8t<F16>(acc[0]);
  }
}

// Strip form (round 5): one thread = 8 channels of FOUR consecutive output pixels of a row.  The 3 x 6 input window is loaded once
// (18 16-byte loads per channel half instead of 36) and the nine weights of the thread's channels sit in registers; per output pixel the
// taps are added in the same (dy, dx) order as in the one-pixel kernel above, so the results are bit-identical.  Needs W % 4 == 0.
template <bool F16>
__global__ __launch_bounds__(256) void dwconv3x3_strip_kernel(const uint16_t* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ bias, uint16_t* __restrict__ y, int N,
                                                              int H, int W, int C, int gate) {
  constexpr int S = 4;
  const int Cout = gate ? C / 2 : C;
  const int CV = Cout >> 3, WS = W / S;
  const long long total = (long long)N * H * WS * CV;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int v = (int)(i % CV);
    long long r = i / CV;
    const int ow0 = (int)(r % WS) * S;
    r /= WS;
    const int oh = (int)(r % H), n = (int)(r / H);
    float acc[2][S][8];
    const int halves = gate ? 2 : 1;
    for (int hf = 0; hf < halves; ++hf) {
      const int c0 = v * 8 + hf * Cout;
      const float4 b0 = *reinterpret_cast<const float4*>(bias + c0), b1 = *reinterpret_cast<const float4*>(bias + c0 + 4);
#pragma unroll
      for (int sx = 0; sx < S; ++sx) {
        float* a = acc[hf][sx];
        a[0] = b0.x; a[1] = b0.y; a[2] = b0.z; a[3] = b0.w; a[4] = b1.x; a[5] = b1.y; a[6] = b1.z; a[7] = b1.w;
      }
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        const int ih = oh + dy - 1;
        if ((unsigned)ih >= (unsigned)H) continue;
        uint4 raw[S + 2];
#pragma unroll
        for (int j = 0; j < S + 2; ++j) {
          const int iw = ow0 + j - 1;
          raw[j] = (unsigned)iw < (unsigned)W ? *reinterpret_cast<const uint4*>(x + (((long long)n * H + ih) * W + iw) * C + c0) : make_uint4(0, 0, 0, 0);
        }
        float wk[3][8];
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const float* wp = w + (dy * 3 + dx) * C + c0;
          const float4 w0 = *reinterpret_cast<const float4*>(wp), w1 = *reinterpret_cast<const float4*>(wp + 4);
          wk[dx][0] = w0.x; wk[dx][1] = w0.y; wk[dx][2] = w0.z; wk[dx][3] = w0.w; wk[dx][4] = w1.x; wk[dx][5] = w1.y; wk[dx][6] = w1.z; wk[dx][7] = w1.w;
        }
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
          for (int sx = 0; sx < S; ++sx) {
            const int iw = ow0 + sx + dx - 1;
            if ((unsigned)iw >= (unsigned)W) continue;          // (zero padding: the tap is skipped, as in the one-pixel kernel)
            float f[8];
            unpack8t<F16>(raw[sx + dx], f);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[hf][sx][e] += f[e] * wk[dx][e];
          }
      }
    }
#pragma unroll
    for (int sx = 0; sx < S; ++sx) {
      if (gate) {
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[0][sx][e] *= acc[1][sx][e];
      }
      *reinterpret_cast<uint4*>(y + ((((long long)n * H + oh) * W + ow0 + sx) * Cout) + v * 8) = pack8t<F16>(acc[0][sx]);
    }
  }
}

template <bool F16>
__global__ __launch_bounds__(256) void scale_channels_kernel(const uint16_t* __restrict__ x, const float* __restrict__ s,
                                                             const uint16_t* __restrict__ res, uint16_t* __restrict__ y,
                                                             long long HWCV, int CV, long long totalv) {
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < totalv; i += (long long)gridDim.x * 256) {
    const int v = (int)(i % CV);
    const long long n = i / HWCV;
    uint4 raw = *reinterpret_cast<const uint4*>(x + i * 8);
    float f[8];
    unpack8t<F16>(raw, f);
    const float* sp = s + (n * CV + v) * 8;
    float4 s0 = *reinterpret_cast<const float4*>(sp), s1 = *reinterpret_cast<const float4*>(sp + 4);
    f[0] *= s0.x; f[1] *= s0.y; f[2] *= s0.z; f[3] *= s0.w; f[4] *= s1.x; f[5] *= s1.y; f[6] *= s1.z; f[7] *= s1.w;
    if (res) {
      float r[8];
      unpack8t<F16>(*reinterpret_cast<const uint4*>(res + i * 8), r);
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] += r[e];
    }
    *reinterpret_cast<uint4*>(y + i * 8) = pack8t<F16>(f);
  }
}

template <bool F16>
__global__ __launch_bounds__(256) void axpy_channels_kernel(const uint16_t* __restrict__ a, const uint16_t* __restrict__ b,
                                                            const float* __restrict__ s, uint16_t* __restrict__ y, int CV,
                                                            long long totalv) {
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < totalv; i += (long long)gridDim.x * 256) {
    const int v = (int)(i % CV);
    float fa[8], fb[8];
    unpack8t<F16>(*reinterpret_cast<const uint4*>(a + i * 8), fa);
    unpack8t<F16>(*reinterpret_cast<const uint4*>(b + i * 8), fb);
    const float* sp = s + v * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) fa[e] += fb[e] * sp[e];
    *reinterpret_cast<uint4*>(y + i * 8) = pack8t<F16>(fa);
  }
}

// SPADE modulation: y = n * (1 + gamma) + beta (+ residual), gamma | beta from one fused conv output [rows][2C]
template <bool F16>
__global__ __launch_bounds__(256) void spade_modulate_kernel(const uint16_t* __restrict__ n, const uint16_t* __restrict__ gb, int ldgb,
                                                             const uint16_t* __restrict__ res, uint16_t* __restrict__ y, int CV,
                                                             long long totalv) {
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < totalv; i += (long long)gridDim.x * 256) {
    const long long r = i / CV;
    const int v = (int)(i - r * CV);
    float fn[8], fg[8], fb[8], fr[8];
    unpack8t<F16>(*reinterpret_cast<const uint4*>(n + i * 8), fn);
    unpack8t<F16>(*reinterpret_cast<const uint4*>(gb + r * ldgb + v * 8), fg);
    unpack8t<F16>(*reinterpret_cast<const uint4*>(gb + r * ldgb + (CV + v) * 8), fb);
    if (res) unpack8t<F16>(*reinterpret_cast<const uint4*>(res + i * 8), fr);
#pragma unroll
    for (int e = 0; e < 8; ++e) fn[e] = fmaf(fn[e], 1.f + fg[e], fb[e]) + (res ? fr[e] : 0.f);
    *reinterpret_cast<uint4*>(y + i * 8) = pack8t<F16>(fn);
  }
}

// ---- tiny fp32 linear: one wave per output column, all M rows (M small) ------------------------------
__global__ __launch_bounds__(256) void linear_f32_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, float* __restrict__ y, int M, int N,
                                                         int K, int groups, int act) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  const int Ng = N / groups, Kg = K / groups, g = n / Ng;
  const float* wr = w + (long long)n * Kg;
  for (int m0 = 0; m0 < M; m0 += 8) {
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // four k per lane and iteration with all 36 loads issued before the first use (one load per dependent step made this
    // kernel K / 64 serialised memory round trips: 24 us for a 512-wide gate MLP); same per-lane summation order
    for (int k0 = lane; k0 < Kg; k0 += 256) {
      float wv[4], xv[4][8];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int k = k0 + 64 * u;
        const bool kin = k < Kg;
        wv[u] = kin ? wr[k] : 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) xv[u][j] = (kin && m0 + j < M) ? x[(long long)(m0 + j) * K + g * Kg + k] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (k0 + 64 * u < Kg && m0 + j < M) acc[j] += wv[u] * xv[u][j];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float v = wave_sum(acc[j]);
      if (lane == 0 && m0 + j < M) y[(long long)(m0 + j) * N + n] = apply_act(v + (bias ? bias[n] : 0.f), act);
    }
  }
}

// ---- TFA prompt update: softmax_D(filter), softmax_D(info), tanh(content); upd = f*cond + i*c ---------
__global__ __launch_bounds__(256) void tfa_prompt_kernel(const float* __restrict__ pooled, const float* __restrict__ cond,
                                                         float* __restrict__ upd, int T, int D) {
  __shared__ float red[16];
  const int bt = blockIdx.x, b = bt / T, tt = bt - b * T, t = threadIdx.x, lane = t & 63, w = t >> 6;
  const float* pf = pooled + ((long long)b * 3 + 0) * T * D + (long long)tt * D;
  const float* pi = pooled + ((long long)b * 3 + 1) * T * D + (long long)tt * D;
  const float* pc = pooled + ((long long)b * 3 + 2) * T * D + (long long)tt * D;
  float mf = -INFINITY, mi = -INFINITY;
  for (int d = t; d < D; d += 256) { mf = fmaxf(mf, pf[d]); mi = fmaxf(mi, pi[d]); }
  mf = wave_max(mf); mi = wave_max(mi);
  if (lane == 0) { red[w] = mf; red[4 + w] = mi; }
  __syncthreads();
  mf = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  mi = fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7]));
  float lf = 0.f, li = 0.f;
  for (int d = t; d < D; d += 256) { lf += expf(pf[d] - mf); li += expf(pi[d] - mi); }
  lf = wave_sum(lf); li = wave_sum(li);
  if (lane == 0) { red[8 + w] = lf; red[12 + w] = li; }
  __syncthreads();
  lf = red[8] + red[9] + red[10] + red[11];
  li = red[12] + red[13] + red[14] + red[15];
  for (int d = t; d < D; d += 256) {
    const float f = expf(pf[d] - mf) / lf, iv = expf(pi[d] - mi) / li, c = tanhf(pc[d]);
    upd[(long long)bt * D + d] = f * cond[(long long)bt * D + d] + iv * c;
  }
}

__global__ void vec_mul_group_kernel(const float* a, const float* b, float* out, int N, int C, int G) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * C) return;
  const int n = i / C, c = i - n * C;
  out[i] = a[i] * b[n * G + c / (C / G)];
}

// ---- layout / boundary kernels --------------------------------------------------------------------------
template <bool F16>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ x, uint16_t* __restrict__ y, int C,
                                                           long long HW, int Cpad, float mul, float add, long long total) {
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long n = i / HW, p = i - n * HW;
    const float* xi = x + n * C * HW + p;
    uint16_t* yo = y + i * Cpad;
    for (int c = 0; c < Cpad; ++c) yo[c] = c < C ? f2h16<F16>(xi[(long long)c * HW] * mul + add) : (uint16_t)0;
  }
}

template <bool F16>
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const void* __restrict__ x, int is_f32, float* __restrict__ out,
                                                           int C, long long HW, int ld, float mul, float add,
                                                           long long total) {
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long n = i / HW, p = i - n * HW;
    for (int c = 0; c < C; ++c) {
      float v = is_f32 ? reinterpret_cast<const float*>(x)[i * ld + c] : Act<F16>::one(reinterpret_cast<const uint16_t*>(x)[i * ld + c]);
      out[(n * C + c) * HW + p] = v * mul + add;
    }
  }
}

// ---- image boundary: bicubic resize (PyTorch semantics: A = -0.75, align_corners=False, antialias=False, source
// index NOT clamped for cubic, tap indices clamped to the image) fused with reflect pad / crop and the layout pass ----
__device__ __forceinline__ void cubic_taps(int dst, float scale, int in_size, int idx[4], float wt[4]) {
  const float real = scale * (dst + 0.5f) - 0.5f;          // upsample_bicubic2d: area_pixel_compute_source_index(cubic=true)
  const float fl = floorf(real);
  const float t = fminf(fmaxf(real - fl, 0.f), 1.f);
  const int i0 = (int)fl;
  const float A = -0.75f;
  const float x1 = t, x2 = 1.f - t;
  wt[0] = ((A * (x1 + 1.f) - 5.f * A) * (x1 + 1.f) + 8.f * A) * (x1 + 1.f) - 4.f * A;
  wt[1] = ((A + 2.f) * x1 - (A + 3.f)) * x1 * x1 + 1.f;
  wt[2] = ((A + 2.f) * x2 - (A + 3.f)) * x2 * x2 + 1.f;
  wt[3] = ((A * (x2 + 1.f) - 5.f * A) * (x2 + 1.f) + 8.f * A) * (x2 + 1.f) - 4.f * A;
#pragma unroll
  for (int j = 0; j < 4; ++j) idx[j] = max(min(i0 + j - 1, in_size - 1), 0);
}

// img [N,C,H,W] fp32 -> (bicubic to RH x RW) -> reflect pad right/bottom -> v*mul+add -> y bf16 [N,RH+PH,RW+PW,Cpad]
template <bool F16>
__global__ __launch_bounds__(256) void image_resize_pad_kernel(const float* __restrict__ img, uint16_t* __restrict__ y, int C,
                                                               int H, int W, int RH, int RW, int PH, int PW, int Cpad,
                                                               float mul, float add, long long total) {
  const int OH = RH + PH, OW = RW + PW;
  const bool resize = RH != H || RW != W;
  const float sh = (float)H / RH, sw = (float)W / RW;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int ox = (int)(i % OW);
    const long long r = i / OW;
    const int oy = (int)(r % OH), n = (int)(r / OH);
    const int ry = oy < RH ? oy : 2 * (RH - 1) - oy, rx = ox < RW ? ox : 2 * (RW - 1) - ox;   // F.pad(mode="reflect")
    const float* base = img + (long long)n * C * H * W;
    uint16_t* yo = y + i * Cpad;
    if (!resize) {
      for (int c = 0; c < Cpad; ++c) yo[c] = c < C ? f2h16<F16>(base[((long long)c * H + ry) * W + rx] * mul + add) : (uint16_t)0;
      continue;
    }
    int iy[4], ix[4];
    float wy[4], wx[4];
    cubic_taps(ry, sh, H, iy, wy);
    cubic_taps(rx, sw, W, ix, wx);
    for (int c = 0; c < Cpad; ++c) {
      float v = 0.f;
      if (c < C) {
        const float* pc = base + (long long)c * H * W;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          const float* row = pc + (long long)iy[a] * W;
          v += wy[a] * (wx[0] * row[ix[0]] + wx[1] * row[ix[1]] + wx[2] * row[ix[2]] + wx[3] * row[ix[3]]);
        }
      }
      yo[c] = c < C ? f2h16<F16>(v * mul + add) : (uint16_t)0;
    }
  }
}

// x NHWC (bf16 | fp32) [N,XH,XW,ld] -> v*mul+add -> crop [0:CH, 0:CW] -> bicubic to OH x OW -> optional
// mul(255).round().clamp(0,255).div(255) -> out fp32 [N,C,OH,OW]
template <bool F16>
__global__ __launch_bounds__(256) void image_unpad_resize_kernel(const void* __restrict__ x, int is_f32, float* __restrict__ out,
                                                                 int C, int XH, int XW, int ld, int CH, int CW, int OH, int OW,
                                                                 float mul, float add, int quantize, long long total) {
  const bool resize = OH != CH || OW != CW;
  const float sh = (float)CH / OH, sw = (float)CW / OW;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int ox = (int)(i % OW);
    const long long r = i / OW;
    const int oy = (int)(r % OH), n = (int)(r / OH);
    int iy[4] = {oy, 0, 0, 0}, ix[4] = {ox, 0, 0, 0};
    float wy[4] = {1.f, 0.f, 0.f, 0.f}, wx[4] = {1.f, 0.f, 0.f, 0.f};
    if (resize) { cubic_taps(oy, sh, CH, iy, wy); cubic_taps(ox, sw, CW, ix, wx); }
    const long long nb = (long long)n * XH * XW;
    for (int c = 0; c < C; ++c) {
      float v = 0.f;
      const int taps = resize ? 4 : 1;
      for (int a = 0; a < taps; ++a) {
        float rowv = 0.f;
        for (int b = 0; b < taps; ++b) {
          const long long e = (nb + (long long)iy[a] * XW + ix[b]) * ld + c;
          const float t = is_f32 ? reinterpret_cast<const float*>(x)[e] : Act<F16>::one(reinterpret_cast<const uint16_t*>(x)[e]);
          rowv += wx[b] * (t * mul + add);
        }
        v += wy[a] * rowv;
      }
      // torch.round = half-to-even = rintf.  A non-finite value stays non-finite (torch.clamp propagates NaN; fmaxf would turn it into a
      // black pixel and hide an fp16 overflow from DiffUIE.forward's finite check): NaN and +-inf both leave as NaN.
      if (quantize) v = (fabsf(v) <= 3.0e38f) ? fminf(fmaxf(rintf(v * 255.f), 0.f), 255.f) / 255.f : __builtin_nanf("");
      out[(((long long)n * C + c) * OH + oy) * OW + ox] = v;
    }
  }
}

template <bool F16>
__global__ __launch_bounds__(256) void vae_sample_kernel(const float* __restrict__ mom, int ld, const float* __restrict__ noise,
                                                         float* __restrict__ z, uint16_t* __restrict__ zb, long long HW,
                                                         int Clat, int Cpad, float scale, long long total) {
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long n = i / HW, p = i - n * HW;
    for (int c = 0; c < Cpad; ++c) {
      float v = 0.f;
      if (c < Clat) {
        const float mean = mom[i * ld + c];
        const float lv = fminf(fmaxf(mom[i * ld + Clat + c], -30.f), 20.f);
        v = (mean + expf(0.5f * lv) * noise[(n * Clat + c) * HW + p]) * scale;
      }
      z[i * Cpad + c] = v;
      zb[i * Cpad + c] = f2h16<F16>(v);
    }
  }
}

template <bool F16>
__global__ __launch_bounds__(256) void add_noise_kernel(const float* __restrict__ z0, const float* __restrict__ noise,
                                                        float* __restrict__ zt, uint16_t* __restrict__ zb, long long HW,
                                                        int Clat, int Cpad, float sa, float sb, long long total) {
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long n = i / HW, p = i - n * HW;
    for (int c = 0; c < Cpad; ++c) {
      const float v = c < Clat ? sa * z0[i * Cpad + c] + sb * noise[(n * Clat + c) * HW + p] : 0.f;
      zt[i * Cpad + c] = v;
      zb[i * Cpad + c] = f2h16<F16>(v);
    }
  }
}

template <bool F16>
__global__ __launch_bounds__(256) void ddim_step_kernel(float* __restrict__ zt, const float* __restrict__ eps, int ld_eps,
                                                        uint16_t* __restrict__ zb, int Clat, int Cpad, float cx, float ce,
                                                        long long total) {
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    for (int c = 0; c < Cpad; ++c) {
      const float v = c < Clat ? cx * zt[i * Cpad + c] + ce * eps[i * ld_eps + c] : 0.f;
      zt[i * Cpad + c] = v;
      zb[i * Cpad + c] = f2h16<F16>(v);
    }
  }
}

template <bool F16>
__global__ __launch_bounds__(256) void f32_to_bf16_kernel(const float* __restrict__ x, int ld, uint16_t* __restrict__ y, int C,
                                                          int Cpad, float mul, long long total) {
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256)
    for (int c = 0; c < Cpad; ++c) y[i * Cpad + c] = c < C ? f2h16<F16>(x[i * ld + c] * mul) : (uint16_t)0;
}

inline int nblocks(long long total) { return (int)std::min<long long>((total + 255) / 256, 8192); }

}  // namespace

extern "C" {

int ur_dwconv3x3_nhwc(const void* x, const float* w9c, const float* bias, void* y, int N, int H, int W, int C, int gate,
                      int dtype, ur_stream_t stream) {
  UR_REQUIRE_DT(dtype);
  UR_REQUIRE(x && w9c && bias && y, "null pointer");
  UR_REQUIRE(C % (gate ? 16 : 8) == 0, "C must be a multiple of 8 (16 with gate)");
  hipStream_t s = (hipStream_t)stream;
  const double elems = (double)N * H * W * C;
  ur::ProfScope prof("dwconv3x3", 18.0 * elems, 2.0 * elems * (gate ? 1.5 : 2.0), s);
  const long long total = (long long)N * H * W * ((gate ? C / 2 : C) / 8);
  static const bool no_strip = getenv("UR_DW_NOSTRIP") != nullptr;
  if (!no_strip && W % 4 == 0) {
    UR_DT_SWITCH(dtype, hipLaunchKernelGGL(dwconv3x3_strip_kernel<F16>, dim3(nblocks(total / 4)), dim3(256), 0, s, (const uint16_t*)x, w9c, bias, (uint16_t*)y,
                       N, H, W, C, gate));
    return ur::check_launch("ur_dwconv3x3_nhwc");
  }
  UR_DT_SWITCH(dtype, hipLaunchKernelGGL(dwconv3x3_kernel<F16>, dim3(nblocks(total)), dim3(256), 0, s, (const uint16_t*)x, w9c, bias, (uint16_t*)y, N,
                     H, W, C, gate));
  return ur::check_launch("ur_dwconv3x3_nhwc");
}

int ur_scale_channels(const void* x, const float* sc, const void* residual, void* y, int N, int HW, int C,
                      int dtype, ur_stream_t stream) {
  UR_REQUIRE_DT(dtype);
  UR_REQUIRE(x && sc && y && C % 8 == 0, "bad args");
  hipStream_t s = (hipStream_t)stream;
  const long long totalv = (long long)N * HW * (C / 8);
  ur::ProfScope prof("elementwise", 0.0, (residual ? 6.0 : 4.0) * totalv * 8.0, s);
  UR_DT_SWITCH(dtype, hipLaunchKernelGGL(scale_channels_kernel<F16>, dim3(nblocks(totalv)), dim3(256), 0, s, (const uint16_t*)x, sc,
                     (const uint16_t*)residual, (uint16_t*)y, (long long)HW * (C / 8), C / 8, totalv));
  return ur::check_launch("ur_scale_channels");
}

int ur_axpy_channels(const void* a, const void* b, const float* sc, void* y, long long rows, int C, int dtype, ur_stream_t stream) {
  UR_REQUIRE_DT(dtype);
  UR_REQUIRE(a && b && sc && y && C % 8 == 0, "bad args");
  hipStream_t s = (hipStream_t)stream;
  const long long totalv = rows * (C / 8);
  ur::ProfScope prof("elementwise", 0.0, 6.0 * totalv * 8.0, s);
  UR_DT_SWITCH(dtype, hipLaunchKernelGGL(axpy_channels_kernel<F16>, dim3(nblocks(totalv)), dim3(256), 0, s, (const uint16_t*)a, (const uint16_t*)b, sc,
                     (uint16_t*)y, C / 8, totalv));
  return ur::check_launch("ur_axpy_channels");
}

int ur_spade_modulate(const void* n, const void* gb, int ldgb, const void* residual, void* y, long long rows, int C,
                      int dtype, ur_stream_t stream) {
  UR_REQUIRE_DT(dtype);
  UR_REQUIRE(n && gb && y && C % 8 == 0 && ldgb % 8 == 0 && ldgb >= 2 * C && rows > 0, "bad args");
  hipStream_t s = (hipStream_t)stream;
  const long long totalv = rows * (C / 8);
  ur::ProfScope prof("elementwise", 0.0, (residual ? 10.0 : 8.0) * totalv * 8.0, s);
  UR_DT_SWITCH(dtype, hipLaunchKernelGGL(spade_modulate_kernel<F16>, dim3(nblocks(totalv)), dim3(256), 0, s, (const uint16_t*)n, (const uint16_t*)gb, ldgb,
                     (const uint16_t*)residual, (uint16_t*)y, C / 8, totalv));
  return ur::check_launch("ur_spade_modulate");
}

int ur_linear_f32(const float* x, const float* w, const float* bias, float* y, int M, int N, int K, int groups, int act,
                  ur_stream_t stream) {
  UR_REQUIRE(x && w && y && M > 0 && N > 0 && K > 0 && groups > 0 && N % groups == 0 && K % groups == 0, "bad args");
  hipStream_t s = (hipStream_t)stream;
  ur::ProfScope prof("linear_f32", 2.0 * M * (double)N * K / groups, 4.0 * N * (double)K / groups, s);
  hipLaunchKernelGGL(linear_f32_kernel, dim3((N + 3) / 4), dim3(256), 0, s, x, w, bias, y, M, N, K, groups, act);
  return ur::check_launch("ur_linear_f32");
}

int ur_tfa_prompt_update(const float* pooled, const float* cond, float* upd, int B, int T, int D, ur_stream_t stream) {
  UR_REQUIRE(pooled && cond && upd && B > 0 && T > 0 && D > 0, "bad args");
  hipLaunchKernelGGL(tfa_prompt_kernel, dim3(B * T), dim3(256), 0, (hipStream_t)stream, pooled, cond, upd, T, D);
  return ur::check_launch("ur_tfa_prompt_update");
}

int ur_vec_mul_group(const float* a, const float* b, float* out, int N, int C, int G, ur_stream_t stream) {
  UR_REQUIRE(a && b && out && C % G == 0, "bad args");
  hipLaunchKernelGGL(vec_mul_group_kernel, dim3((N * C + 255) / 256), dim3(256), 0, (hipStream_t)stream, a, b, out, N, C, G);
  return ur::check_launch("ur_vec_mul_group");
}

int ur_nchw_f32_to_nhwc(const float* x, void* y, int N, int C, int H, int W, int Cpad, int dtype, ur_stream_t stream) {
  UR_REQUIRE_DT(dtype);
  UR_REQUIRE(x && y && Cpad >= C, "bad args");
  const long long total = (long long)N * H * W;
  UR_DT_SWITCH(dtype, hipLaunchKernelGGL(nchw_to_nhwc_kernel<F16>, dim3(nblocks(total)), dim3(256), 0, (hipStream_t)stream, x, (uint16_t*)y, C,
                     (long long)H * W, Cpad, 1.f, 0.f, total));
  return ur::check_launch("ur_nchw_f32_to_nhwc");
}

int ur_image_to_nhwc(const float* img, void* y, int N, int C, int H, int W, int Cpad, int dtype, ur_stream_t stream) {
  UR_REQUIRE_DT(dtype);
  UR_REQUIRE(img && y && Cpad >= C, "bad args");
  const long long total = (long long)N * H * W;
  UR_DT_SWITCH(dtype, hipLaunchKernelGGL(nchw_to_nhwc_kernel<F16>, dim3(nblocks(total)), dim3(256), 0, (hipStream_t)stream, img, (uint16_t*)y, C,
                     (long long)H * W, Cpad, 2.f, -1.f, total));
  return ur::check_launch("ur_image_to_nhwc");
}

int ur_nhwc_to_nchw_f32(const void* x, int x_is_f32, float* out, int N, int C, int H, int W, int ld, float mul, float add,
                        int dtype, ur_stream_t stream) {
  UR_REQUIRE_DT(dtype);
  UR_REQUIRE(x && out && ld >= C, "bad args");
  const long long total = (long long)N * H * W;
  UR_DT_SWITCH(dtype, hipLaunchKernelGGL(nhwc_to_nchw_kernel<F16>, dim3(nblocks(total)), dim3(256), 0, (hipStream_t)stream, x, x_is_f32, out, C,
                     (long long)H * W, ld, mul, add, total));
  return ur::check_launch("ur_nhwc_to_nchw_f32");
}

int ur_image_resize_pad_nhwc(const float* img, void* y, int N, int C, int H, int W, int RH, int RW, int PH, int PW, int Cpad,
                             float mul, float add, int dtype, ur_stream_t stream) {
  UR_REQUIRE_DT(dtype);
  UR_REQUIRE(img && y && Cpad >= C && N > 0 && H > 0 && W > 0 && RH > 0 && RW > 0, "bad args");
  UR_REQUIRE(PH >= 0 && PW >= 0 && PH < RH && PW < RW, "reflect padding must be smaller than the image");
  const long long total = (long long)N * (RH + PH) * (RW + PW);
  UR_DT_SWITCH(dtype, hipLaunchKernelGGL(image_resize_pad_kernel<F16>, dim3(nblocks(total)), dim3(256), 0, (hipStream_t)stream, img, (uint16_t*)y, C, H, W,
                     RH, RW, PH, PW, Cpad, mul, add, total));
  return ur::check_launch("ur_image_resize_pad_nhwc");
}

int ur_image_unpad_resize_nchw(const void* x, int x_is_f32, float* out, int N, int C, int XH, int XW, int ld, int CH, int CW,
                               int OH, int OW, float mul, float add, int quantize, int dtype, ur_stream_t stream) {
  UR_REQUIRE_DT(dtype);
  UR_REQUIRE(x && out && ld >= C && N > 0 && OH > 0 && OW > 0, "bad args");
  UR_REQUIRE(CH > 0 && CW > 0 && CH <= XH && CW <= XW, "crop window must lie inside the input");
  const long long total = (long long)N * OH * OW;
  UR_DT_SWITCH(dtype, hipLaunchKernelGGL(image_unpad_resize_kernel<F16>, dim3(nblocks(total)), dim3(256), 0, (hipStream_t)stream, x, x_is_f32, out, C, XH, XW,
                     ld, CH, CW, OH, OW, mul, add, quantize, total));
  return ur::check_launch("ur_image_unpad_resize_nchw");
}

int ur_vae_sample(const float* moments, int ld, const float* noise_nchw, float* z_nhwc, void* z_16, int N, int HW,
                  int Clat, int Cpad, float scale, int dtype, ur_stream_t stream) {
  UR_REQUIRE_DT(dtype);
  UR_REQUIRE(moments && noise_nchw && z_nhwc && z_16 && ld >= 2 * Clat && Cpad >= Clat, "bad args");
  const long long total = (long long)N * HW;
  UR_DT_SWITCH(dtype, hipLaunchKernelGGL(vae_sample_kernel<F16>, dim3(nblocks(total)), dim3(256), 0, (hipStream_t)stream, moments, ld, noise_nchw,
                     z_nhwc, (uint16_t*)z_16, (long long)HW, Clat, Cpad, scale, total));
  return ur::check_launch("ur_vae_sample");
}

int ur_add_noise(const float* z0, const float* noise_nchw, float* zt, void* zt_16, int N, int HW, int Clat, int Cpad,
                 float sa, float sb, int dtype, ur_stream_t stream) {
  UR_REQUIRE_DT(dtype);
  UR_REQUIRE(z0 && noise_nchw && zt && zt_16, "null pointer");
  const long long total = (long long)N * HW;
  UR_DT_SWITCH(dtype, hipLaunchKernelGGL(add_noise_kernel<F16>, dim3(nblocks(total)), dim3(256), 0, (hipStream_t)stream, z0, noise_nchw, zt,
                     (uint16_t*)zt_16, (long long)HW, Clat, Cpad, sa, sb, total));
  return ur::check_launch("ur_add_noise");
}

int ur_ddim_step(float* zt, const float* eps, int ld_eps, void* zt_16, long long M, int Clat, int Cpad, float c_x,
                 float c_e, int dtype, ur_stream_t stream) {
  UR_REQUIRE_DT(dtype);
  UR_REQUIRE(zt && eps && zt_16 && ld_eps >= Clat, "bad args");
  UR_DT_SWITCH(dtype, hipLaunchKernelGGL(ddim_step_kernel<F16>, dim3(nblocks(M)), dim3(256), 0, (hipStream_t)stream, zt, eps, ld_eps,
                     (uint16_t*)zt_16, Clat, Cpad, c_x, c_e, M));
  return ur::check_launch("ur_ddim_step");
}

int ur_f32_to_bf16_scaled(const float* x, int ld, void* y, long long M, int C, int Cpad, float mul, int dtype, ur_stream_t stream) {
  UR_REQUIRE_DT(dtype);
  UR_REQUIRE(x && y && ld >= C && Cpad >= C, "bad args");
  UR_DT_SWITCH(dtype, hipLaunchKernelGGL(f32_to_bf16_kernel<F16>, dim3(nblocks(M)), dim3(256), 0, (hipStream_t)stream, x, ld, (uint16_t*)y, C, Cpad,
                     mul, M));
  return ur::check_launch("ur_f32_to_bf16_scaled");
}

}  // extern "C"
