// Round-5 probe: is the L2 -> CU path of ordinary VGPR-destination loads (buffer_load_dwordx4 v[...]) a second path beside the
// LDS-DMA path (buffer_load_dwordx4 ... lds), or do both share one issue bound?   hipcc --offload-arch=gfx950 -O3 -o dma_vmem_probe ...
//
// Every wave streams 1-KiB pieces (64 lanes x 16 B, contiguous) out of a window of `foot` bytes that all workgroups share
// (L2-resident at 2-8 MB, MALL-resident at 64 MB), in batches of 8 pieces followed by one s_waitcnt vmcnt(0):
//   mode 0: 8 LDS-DMA pieces          mode 1: 8 VGPR pieces          mode 2: 4 + 4 interleaved          mode 3: 8 DMA + 8 VGPR per batch
// Reported: GB/s per CU and bytes per shader clock and CU (clock measured with s_memtime over s_memrealtime).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ u32x4 make_rsrc(const void* base, unsigned bytes) {
  const unsigned long long a = (unsigned long long)base;
  u32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((unsigned)a);
  r[1] = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffffu);
  r[2] = __builtin_amdgcn_readfirstlane(bytes);
  r[3] = 0x00020000u;
  return r;
}

#define DMA(ldsaddr, voff, soff) \
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(ldsaddr), "v"(voff), "s"(rs), "s"(soff) : "memory")
#define VLD(dst, voff, soff) asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(dst) : "v"(voff), "s"(rs), "s"(soff) : "memory")

template <int MODE>
__global__ __launch_bounds__(1024) void probe(const unsigned char* __restrict__ src, unsigned foot, int iters, unsigned* out, unsigned long long* clk) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const u32x4 rs = make_rsrc(src, foot);
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)smem + wid * 8192);
  const unsigned voff = lane * 16;
  // every wave walks the window with its own phase; consecutive pieces are consecutive KiB (what a packed weight stream looks like)
  unsigned pos = ((blockIdx.x * nw + wid) * 16384u * 7u) % foot;
  u32x4 acc = {0, 0, 0, 0};
  const unsigned long long t0 = __builtin_readcyclecounter();
  unsigned long long r0;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r0));
  for (int it = 0; it < iters; ++it) {
    u32x4 v0, v1, v2, v3, v4, v5, v6, v7;
    const unsigned s0 = __builtin_amdgcn_readfirstlane(pos);
    if (MODE == 0) {
#pragma unroll
      for (int j = 0; j < 8; ++j) DMA(lds0 + j * 1024, voff, s0 + j * 1024);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (MODE == 1) {
      VLD(v0, voff, s0); VLD(v1, voff, s0 + 1024); VLD(v2, voff, s0 + 2048); VLD(v3, voff, s0 + 3072);
      VLD(v4, voff, s0 + 4096); VLD(v5, voff, s0 + 5120); VLD(v6, voff, s0 + 6144); VLD(v7, voff, s0 + 7168);
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7)::"memory");
      acc ^= v0 ^ v1 ^ v2 ^ v3 ^ v4 ^ v5 ^ v6 ^ v7;
    } else if (MODE == 2) {
      DMA(lds0, voff, s0); VLD(v0, voff, s0 + 1024); DMA(lds0 + 1024, voff, s0 + 2048); VLD(v1, voff, s0 + 3072);
      DMA(lds0 + 2048, voff, s0 + 4096); VLD(v2, voff, s0 + 5120); DMA(lds0 + 3072, voff, s0 + 6144); VLD(v3, voff, s0 + 7168);
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3)::"memory");
      acc ^= v0 ^ v1 ^ v2 ^ v3;
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) DMA(lds0 + j * 1024, voff, s0 + j * 1024);
      const unsigned s1 = s0 + 8192;
      VLD(v0, voff, s1); VLD(v1, voff, s1 + 1024); VLD(v2, voff, s1 + 2048); VLD(v3, voff, s1 + 3072);
      VLD(v4, voff, s1 + 4096); VLD(v5, voff, s1 + 5120); VLD(v6, voff, s1 + 6144); VLD(v7, voff, s1 + 7168);
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7)::"memory");
      acc ^= v0 ^ v1 ^ v2 ^ v3 ^ v4 ^ v5 ^ v6 ^ v7;
    }
    pos += (MODE == 3 ? 16384u : 8192u);
    if (pos + 16384u > foot) pos -= (foot - 16384u) & ~1023u;
  }
  unsigned long long r1;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r1));
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (acc[0] == 0x12345678u && acc[1] == 77u) out[0] = acc[2] ^ acc[3];          // keep the VGPR loads alive
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int MODE>
static void run(const char* name, const unsigned char* src, unsigned foot, int waves, int wg_per_cu, unsigned* out, unsigned long long* clk) {
  const int iters = 2000, ncu = 256;
  const int grid = ncu * wg_per_cu, block = waves * 64;
  const size_t lds = (size_t)waves * 8192;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(block), lds, 0, src, foot, 50, out, clk);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(block), lds, 0, src, foot, iters, out, clk);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[2];
  hipMemcpy(h, clk, sizeof h, hipMemcpyDeviceToHost);
  const double bytes = (double)grid * waves * iters * (MODE == 3 ? 16384.0 : 8192.0);
  const double ghz = (double)h[0] / ((double)h[1] * 10.0);                       // s_memrealtime ticks at 100 MHz
  const double per_cu = bytes / ncu / (ms * 1e-3);
  printf("%-28s foot %3u MB  %2d waves x %d WG/CU : %7.1f us  %6.2f TB/s chip  %6.1f GB/s/CU  %5.1f B/clk/CU  (shader %.2f GHz)\n", name, foot >> 20, waves,
         wg_per_cu, ms * 1e3, bytes / (ms * 1e-3) / 1e12, per_cu / 1e9, per_cu / (ghz * 1e9), ghz);
}

int main() {
  const unsigned maxfoot = 64u << 20;
  unsigned char* src;
  unsigned* out;
  unsigned long long* clk;
  hipMalloc(&src, maxfoot + 65536);
  hipMemset(src, 1, maxfoot + 65536);
  hipMalloc(&out, 64);
  hipMalloc(&clk, 64);
  for (unsigned foot : {2u << 20, 16u << 20, 64u << 20})
    for (int waves : {4, 8, 16}) {
      const int wg = 1;
      run<0>("LDS-DMA only", src, foot, waves, wg, out, clk);
      run<1>("VGPR loads only", src, foot, waves, wg, out, clk);
      run<2>("4 DMA + 4 VGPR interleaved", src, foot, waves, wg, out, clk);
      run<3>("8 DMA + 8 VGPR per batch", src, foot, waves, wg, out, clk);
    }
  // two workgroups per CU (the short GEMMs' residency)
  run<0>("LDS-DMA only", src, 16u << 20, 4, 2, out, clk);
  run<1>("VGPR loads only", src, 16u << 20, 4, 2, out, clk);
  run<2>("4 DMA + 4 VGPR interleaved", src, 16u << 20, 4, 2, out, clk);
  hipDeviceSynchronize();
  return 0;
}
