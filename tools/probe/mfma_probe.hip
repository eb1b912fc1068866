// Stand-alone probe: issue rate of v_mfma_f32_32x32x16_bf16 from ONE wave per SIMD in the exact asm patterns tchain.hip uses,
// with the shader clock measured in-kernel (s_memtime = shader cycles, s_memrealtime = 100 MHz).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_probe tools/probe/mfma_probe.hip && ./mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define MF(acc, a, b) "v_mfma_f32_32x32x16_bf16 " acc ", " a ", " b ", " acc "\n\t"

template <int MODE>
__global__ __launch_bounds__(256, 1) void probe(unsigned long long* out, int iters, float* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 40960 / 4; i += 256) reinterpret_cast<unsigned*>(smem)[i] = i * 2654435761u;
  __syncthreads();
  f32x16 acc[10];
  for (int f = 0; f < 10; ++f) for (int e = 0; e < 16; ++e) acc[f][e] = 0.f;
  bf16x8 a, b, t0, t1, t2, t3, t4, t5, t6, t7;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.001f * (lane + e)); b[e] = (__bf16)(0.002f * (lane - e)); }
  t0 = t1 = t2 = t3 = t4 = t5 = t6 = t7 = a;
  const unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)smem + (lane & 31) * 128 + (((lane >> 5) ^ ((lane >> 1) & 7)) << 4);
  unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {          // 40 MFMAs, 10 accumulators in AGPRs, operands in VGPRs, nothing else
      asm volatile(
        MF("%0","%10","%11") MF("%1","%10","%11") MF("%2","%10","%11") MF("%3","%10","%11") MF("%4","%10","%11")
        MF("%5","%10","%11") MF("%6","%10","%11") MF("%7","%10","%11") MF("%8","%10","%11") MF("%9","%10","%11")
        MF("%0","%10","%11") MF("%1","%10","%11") MF("%2","%10","%11") MF("%3","%10","%11") MF("%4","%10","%11")
        MF("%5","%10","%11") MF("%6","%10","%11") MF("%7","%10","%11") MF("%8","%10","%11") MF("%9","%10","%11")
        MF("%0","%10","%11") MF("%1","%10","%11") MF("%2","%10","%11") MF("%3","%10","%11") MF("%4","%10","%11")
        MF("%5","%10","%11") MF("%6","%10","%11") MF("%7","%10","%11") MF("%8","%10","%11") MF("%9","%10","%11")
        MF("%0","%10","%11") MF("%1","%10","%11") MF("%2","%10","%11") MF("%3","%10","%11") MF("%4","%10","%11")
        MF("%5","%10","%11") MF("%6","%10","%11") MF("%7","%10","%11") MF("%8","%10","%11") MF("%9","%10","%11")
        : "+a"(acc[0]), "+a"(acc[1]), "+a"(acc[2]), "+a"(acc[3]), "+a"(acc[4]), "+a"(acc[5]), "+a"(acc[6]), "+a"(acc[7]), "+a"(acc[8]), "+a"(acc[9])
        : "v"(a), "v"(b));
    } else if (MODE == 1) {   // the same with accumulators in VGPRs
      asm volatile(
        MF("%0","%10","%11") MF("%1","%10","%11") MF("%2","%10","%11") MF("%3","%10","%11") MF("%4","%10","%11")
        MF("%5","%10","%11") MF("%6","%10","%11") MF("%7","%10","%11") MF("%8","%10","%11") MF("%9","%10","%11")
        MF("%0","%10","%11") MF("%1","%10","%11") MF("%2","%10","%11") MF("%3","%10","%11") MF("%4","%10","%11")
        MF("%5","%10","%11") MF("%6","%10","%11") MF("%7","%10","%11") MF("%8","%10","%11") MF("%9","%10","%11")
        MF("%0","%10","%11") MF("%1","%10","%11") MF("%2","%10","%11") MF("%3","%10","%11") MF("%4","%10","%11")
        MF("%5","%10","%11") MF("%6","%10","%11") MF("%7","%10","%11") MF("%8","%10","%11") MF("%9","%10","%11")
        MF("%0","%10","%11") MF("%1","%10","%11") MF("%2","%10","%11") MF("%3","%10","%11") MF("%4","%10","%11")
        MF("%5","%10","%11") MF("%6","%10","%11") MF("%7","%10","%11") MF("%8","%10","%11") MF("%9","%10","%11")
        : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7]), "+v"(acc[8]), "+v"(acc[9])
        : "v"(a), "v"(b));
    } else if (MODE == 2) {   // 2 alternating accumulators (FF1 pattern)
      asm volatile(
        MF("%0","%2","%3") MF("%1","%2","%3") MF("%0","%2","%3") MF("%1","%2","%3") MF("%0","%2","%3") MF("%1","%2","%3") MF("%0","%2","%3") MF("%1","%2","%3")
        MF("%0","%2","%3") MF("%1","%2","%3") MF("%0","%2","%3") MF("%1","%2","%3") MF("%0","%2","%3") MF("%1","%2","%3") MF("%0","%2","%3") MF("%1","%2","%3")
        MF("%0","%2","%3") MF("%1","%2","%3") MF("%0","%2","%3") MF("%1","%2","%3") MF("%0","%2","%3") MF("%1","%2","%3") MF("%0","%2","%3") MF("%1","%2","%3")
        MF("%0","%2","%3") MF("%1","%2","%3") MF("%0","%2","%3") MF("%1","%2","%3") MF("%0","%2","%3") MF("%1","%2","%3") MF("%0","%2","%3") MF("%1","%2","%3")
        MF("%0","%2","%3") MF("%1","%2","%3") MF("%0","%2","%3") MF("%1","%2","%3") MF("%0","%2","%3") MF("%1","%2","%3") MF("%0","%2","%3") MF("%1","%2","%3")
        : "+a"(acc[0]), "+a"(acc[1]) : "v"(a), "v"(b));
    } else if (MODE == 3) {   // 40 ds_read_b128 only, 8 in flight
#define RD(t, off) "ds_read_b128 " t ", %8 offset:" #off "\n\t"
      asm volatile(
        RD("%0",0) RD("%1",4096) RD("%2",8192) RD("%3",12288) RD("%4",16384) RD("%5",20480) RD("%6",24576) RD("%7",28672)
        "s_waitcnt lgkmcnt(7)\n\t" RD("%0",32768) "s_waitcnt lgkmcnt(7)\n\t" RD("%1",36864) "s_waitcnt lgkmcnt(7)\n\t" RD("%2",0) "s_waitcnt lgkmcnt(7)\n\t" RD("%3",4096)
        "s_waitcnt lgkmcnt(7)\n\t" RD("%4",8192) "s_waitcnt lgkmcnt(7)\n\t" RD("%5",12288) "s_waitcnt lgkmcnt(7)\n\t" RD("%6",16384) "s_waitcnt lgkmcnt(7)\n\t" RD("%7",20480)
        "s_waitcnt lgkmcnt(7)\n\t" RD("%0",24576) "s_waitcnt lgkmcnt(7)\n\t" RD("%1",28672) "s_waitcnt lgkmcnt(7)\n\t" RD("%2",32768) "s_waitcnt lgkmcnt(7)\n\t" RD("%3",36864)
        "s_waitcnt lgkmcnt(7)\n\t" RD("%4",0) "s_waitcnt lgkmcnt(7)\n\t" RD("%5",4096) "s_waitcnt lgkmcnt(7)\n\t" RD("%6",8192) "s_waitcnt lgkmcnt(7)\n\t" RD("%7",12288)
        "s_waitcnt lgkmcnt(7)\n\t" RD("%0",16384) "s_waitcnt lgkmcnt(7)\n\t" RD("%1",20480) "s_waitcnt lgkmcnt(7)\n\t" RD("%2",24576) "s_waitcnt lgkmcnt(7)\n\t" RD("%3",28672)
        "s_waitcnt lgkmcnt(7)\n\t" RD("%4",32768) "s_waitcnt lgkmcnt(7)\n\t" RD("%5",36864) "s_waitcnt lgkmcnt(7)\n\t" RD("%6",0) "s_waitcnt lgkmcnt(7)\n\t" RD("%7",4096)
        "s_waitcnt lgkmcnt(7)\n\t" RD("%0",8192) "s_waitcnt lgkmcnt(7)\n\t" RD("%1",12288) "s_waitcnt lgkmcnt(7)\n\t" RD("%2",16384) "s_waitcnt lgkmcnt(7)\n\t" RD("%3",20480)
        "s_waitcnt lgkmcnt(7)\n\t" RD("%4",24576) "s_waitcnt lgkmcnt(7)\n\t" RD("%5",28672) "s_waitcnt lgkmcnt(7)\n\t" RD("%6",32768) "s_waitcnt lgkmcnt(7)\n\t" RD("%7",36864)
        "s_waitcnt lgkmcnt(0)\n\t"
        : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7) : "v"(addr) : "memory");
    }
  }
  unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int f = 0; f < 10; ++f) s += acc[f][0];
  s += (float)t0[0] + (float)t1[0] + (float)t2[0] + (float)t3[0] + (float)t4[0] + (float)t5[0] + (float)t6[0] + (float)t7[0];
  if (s == 123.456f) sink[0] = s;
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = c1 - c0; out[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int MODE> void run(const char* name, int blocks) {
  unsigned long long* d; float* sink;
  hipMalloc(&d, blocks * 16); hipMalloc(&sink, 4);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  const int iters = 2000;
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 40960, 0, d, iters, sink);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 40960, 0, d, iters, sink);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(2 * blocks);
  hipMemcpy(h.data(), d, blocks * 16, hipMemcpyDeviceToHost);
  double cyc = 0, rt = 0;
  for (int i = 0; i < blocks; ++i) { cyc += h[2 * i]; rt += h[2 * i + 1]; }
  cyc /= blocks; rt /= blocks;
  printf("%-44s blocks %4d: %8.1f us  shader clock %.2f GHz  %.1f cycles per op (40 ops x %d iters)\n", name, blocks, ms * 1e3,
         cyc / (rt * 10.0) , cyc / (40.0 * iters), iters);
  hipFree(d); hipFree(sink);
}

int main() {
  for (int blocks : {256, 8}) {
    run<0>("40 MFMA, 10 acc in AGPR", blocks);
    run<1>("40 MFMA, 10 acc in VGPR", blocks);
    run<2>("40 MFMA, 2 alternating acc (AGPR)", blocks);
    run<3>("40 ds_read_b128, 8 in flight", blocks);
  }
  return 0;
}
