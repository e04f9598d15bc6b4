// Ceiling of the GEMM operand stream: workgroups of 256 threads pull [128 rows x BK] bf16 panels (A and B: 2 panels per k-tile)
// from an L2-resident matrix into LDS with global_load_lds_dwordx4, double-buffered with one barrier per k-tile exactly as
// gemm_kernel does -- but with no MFMAs and no epilogue.  Prints TB/s of L2 -> LDS traffic.
// usage: dma_probe <row_bytes: 64|128> <lds_bytes_per_wg> <wgs> <ktiles> <rows_total> <depth: loads in flight, 1|2>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((address_space(3))) void* lds_vp;
typedef __attribute__((address_space(1))) const void* glb_vp;

template <int ROWB, int DEPTH>
__global__ __launch_bounds__(256) void probe(const unsigned char* __restrict__ base, int64_t ld, int rows_total, int ktiles, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int PANEL_BYTES = 128 * ROWB, NSEG = PANEL_BYTES / 1024, CPR = ROWB / 16, RPS = 1024 / ROWB, BUF = 2 * PANEL_BYTES;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tiles = rows_total / 128;
  const int ta = blockIdx.x % tiles, tb = (blockIdx.x * 7 + 3) % tiles;
  auto issue = [&](unsigned char* buf, int kt) {
#pragma unroll
    for (int pnl = 0; pnl < 2; ++pnl) {
      const int t0 = (pnl ? tb : ta) * 128;
#pragma unroll
      for (int i = 0; i < NSEG / 4; ++i) {
        const int sg = wave + i * 4, r = sg * RPS + lane / CPR, p = lane % CPR;
        const unsigned char* src = base + (int64_t)(t0 + r) * ld + (int64_t)kt * ROWB + p * 16;
        __builtin_amdgcn_global_load_lds((glb_vp)src, (lds_vp)(buf + pnl * PANEL_BYTES + sg * 1024), 16, 0, 0);
      }
    }
  };
  constexpr int TPW = 2 * NSEG / 4;
  unsigned acc = 0;
  for (int d = 0; d < DEPTH; ++d) issue(smem + d * BUF, d);
  for (int kt = 0; kt < ktiles; ++kt) {
    if (DEPTH == 1 || kt + DEPTH - 1 >= ktiles) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(TPW * (DEPTH - 1)) : "memory");
    __builtin_amdgcn_s_barrier();
    acc += *reinterpret_cast<unsigned*>(smem + (kt % (DEPTH + 1)) * BUF + threadIdx.x * 4);
    if (kt + DEPTH < ktiles) issue(smem + ((kt + DEPTH) % (DEPTH + 1)) * BUF, kt + DEPTH);
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

int main(int argc, char** argv) {
  const int rowb = argc > 1 ? atoi(argv[1]) : 64, lds = argc > 2 ? atoi(argv[2]) : 32768, wgs = argc > 3 ? atoi(argv[3]) : 1024;
  const int ktiles = argc > 4 ? atoi(argv[4]) : 48, rows = argc > 5 ? atoi(argv[5]) : 4096, depth = argc > 6 ? atoi(argv[6]) : 1;
  const int64_t ld = (int64_t)ktiles * rowb;
  unsigned char* d; unsigned* sink;
  hipMalloc(&d, rows * ld); hipMemset(d, 1, rows * ld); hipMalloc(&sink, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&]() {
    if (rowb == 64 && depth == 1) hipLaunchKernelGGL((probe<64, 1>), dim3(wgs), dim3(256), lds, 0, d, ld, rows, ktiles, sink);
    else if (rowb == 64) hipLaunchKernelGGL((probe<64, 2>), dim3(wgs), dim3(256), lds, 0, d, ld, rows, ktiles, sink);
    else if (depth == 1) hipLaunchKernelGGL((probe<128, 1>), dim3(wgs), dim3(256), lds, 0, d, ld, rows, ktiles, sink);
    else hipLaunchKernelGGL((probe<128, 2>), dim3(wgs), dim3(256), lds, 0, d, ld, rows, ktiles, sink);
  };
  hipFuncSetAttribute(reinterpret_cast<const void*>(probe<128, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipFuncSetAttribute(reinterpret_cast<const void*>(probe<128, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipFuncSetAttribute(reinterpret_cast<const void*>(probe<64, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int i = 0; i < 3; ++i) run();
  hipEventRecord(e0);
  const int reps = 20;
  for (int i = 0; i < reps; ++i) run();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)wgs * ktiles * 2 * 128 * rowb;
  printf("row %3d B  lds/wg %6d  wgs %5d  ktiles %3d  matrix %6.1f MB  depth %d : %7.2f us/launch  %6.2f TB/s  (%.1f B/clk/CU at 2.4 GHz)\n", rowb, lds, wgs, ktiles,
         rows * ld / 1e6, depth, ms * 1e3 / reps, bytes / (ms * 1e-3 / reps) / 1e12, bytes / (ms * 1e-3 / reps) / 256 / 2.4e9);
  return 0;
}
