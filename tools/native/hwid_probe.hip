// Where do the workgroups of a launch land?  Records (XCC_ID, SE, CU) of every block of a grid whose blocks are all
// co-resident (32 KB LDS, 256 threads: the GEMM's footprint) -- input for the blockIdx -> tile mapping (speed only).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256) void probe(unsigned* out, int spin) {
  extern __shared__ unsigned char smem[];
  if (threadIdx.x == 0) {
    unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    out[blockIdx.x * 2] = hw; out[blockIdx.x * 2 + 1] = xcc;
  }
  volatile unsigned char* s = smem;
  unsigned acc = 0;
  for (int i = 0; i < spin; ++i) { s[threadIdx.x] = (unsigned char)i; acc += s[(threadIdx.x + 1) & 255]; }
  if (acc == 0xffffffffu) out[0] = acc;
}
int main(int argc, char** argv) {
  const int nblk = argc > 1 ? atoi(argv[1]) : 1024, lds = argc > 2 ? atoi(argv[2]) : 32768;
  unsigned* d; hipMalloc(&d, nblk * 8);
  hipLaunchKernelGGL(probe, dim3(nblk), dim3(256), lds, 0, d, 20000);
  hipDeviceSynchronize();
  std::vector<unsigned> h(nblk * 2);
  hipMemcpy(h.data(), d, nblk * 8, hipMemcpyDeviceToHost);
  printf("# blk xcc se sh cu   (XCD 0 .. by block order)\n");
  for (int x = 0; x < 2; ++x) {
    printf("XCD %d:", x);
    for (int b = x; b < nblk; b += 8) {
      const unsigned hw = h[b * 2], xcc = h[b * 2 + 1] & 15;
      printf(" %u:%u.%u.%u", xcc, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15);
    }
    printf("\n");
  }
  return 0;
}
