// Instruction-cache probe: the same VALU stream (v_dot2c_f32_bf16, 8 independent accumulators) as straight-line code of 8 KB .. 256 KB executed in a loop by 8 waves of one workgroup
// (and by `nwg` workgroups on as many CUs): cycles per wave-instruction.  A persistent stage kernel's block body is ~80 KB of straight-line code; is a VALU-bound phase of it fed by
// the instruction cache or by L2?
//   hipcc --offload-arch=gfx950 -O2 -o icache_probe icache_probe.hip && ./icache_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define I8 "v_dot2c_f32_bf16 %0, %9, %8\n v_dot2c_f32_bf16 %1, %9, %8\n v_dot2c_f32_bf16 %2, %9, %8\n v_dot2c_f32_bf16 %3, %9, %8\n v_dot2c_f32_bf16 %4, %9, %8\n v_dot2c_f32_bf16 %5, %9, %8\n v_dot2c_f32_bf16 %6, %9, %8\n v_dot2c_f32_bf16 %7, %9, %8\n"
#define A8 asm volatile(I8 I8 I8 I8 I8 I8 I8 I8 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(w), "v"(h));      // 64 instructions, 256 B (e32: 4 B each)
#define R4(x) x x x x
#define R8(x) x x x x x x x x
// KB = code bytes of the loop body / 1024; e32 VALU = 4 bytes per instruction: 64 instr = 256 B, so 4 x A8 = 1 KB
template <int KB>
__global__ void k(unsigned long long* out, float* sink, float seed, int iters) {
  float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7, w = 1.0001f;
  unsigned h = 0x3c003c00u;
  __builtin_amdgcn_s_barrier();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if constexpr (KB == 8) { R8(R4(A8)) }
    if constexpr (KB == 32) { R4(R8(R4(A8))) }
    if constexpr (KB == 64) { R8(R8(R4(A8))) }
    if constexpr (KB == 128) { R8(R8(R4(A8))) R8(R8(R4(A8))) }
    if constexpr (KB == 256) { R4(R8(R8(R4(A8)))) }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
  if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.f) sink[0] = a0;
}
template <int KB> void run(unsigned long long* out, float* sink) {
  for (int nwg : {1, 256}) for (int threads : {256, 512}) {
    const int iters = 2048 / KB;                                  // same instruction count for every size: 2048 KB / 4 B = 512 k instructions per wave
    hipLaunchKernelGGL(k<KB>, dim3(nwg), dim3(threads), 0, 0, out, sink, 0.5f, iters);
    hipLaunchKernelGGL(k<KB>, dim3(nwg), dim3(threads), 0, 0, out, sink, 0.5f, iters);
    unsigned long long h[16 * 256]; hipMemcpy(h, out, sizeof(unsigned long long) * 16 * nwg, hipMemcpyDeviceToHost);
    double mx = 0; for (int b = 0; b < nwg; ++b) mx = h[b * 16] > mx ? (double)h[b * 16] : mx;
    printf("body %3d KB, %3d workgroup(s) x %d waves: %.2f cycles per wave-instruction (slowest workgroup %.2f)\n", KB, nwg, threads / 64, h[0] / (double)(iters * KB * 256.0), mx / (double)(iters * KB * 256.0));
  }
}
int main() {
  unsigned long long* out; float* sink; hipMalloc(&out, 16 * 256 * 8); hipMalloc(&sink, 4);
  run<8>(out, sink); run<32>(out, sink); run<64>(out, sink); run<128>(out, sink); run<256>(out, sink);
}
