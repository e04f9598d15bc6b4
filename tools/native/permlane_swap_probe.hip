// Probe: sum over the 4 lane groups of a 16-lane row with v_permlane16_swap + v_permlane32_swap against the __shfl_xor form (lmv_xsum4, csrc/common.h).
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ float xsum4b(float v) {
  unsigned u = __float_as_uint(v);
  auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  u = __float_as_uint(v);
  auto q = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __uint_as_float(q[0]) + __uint_as_float(q[1]);
}
__device__ __forceinline__ float xsum4a(float v) { v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64); return v; }
__global__ void k(const float* a, float* o) { o[threadIdx.x] = xsum4b(a[threadIdx.x]); o[64 + threadIdx.x] = xsum4a(a[threadIdx.x]); }
int main() {
  float h[64], r[128]; for (int i = 0; i < 64; ++i) h[i] = (float)(1 << (i >> 4)) * 100.f + (i & 15);
  float *a, *o; (void)hipMalloc(&a, 256); (void)hipMalloc(&o, 512); (void)hipMemcpy(a, h, 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, o); (void)hipMemcpy(r, o, 512, hipMemcpyDeviceToHost);
  int bad = 0; for (int i = 0; i < 64; ++i) if (r[i] != r[64 + i]) ++bad;
  printf("bad %d; lane 0 %g %g, lane 17 %g %g, lane 63 %g %g\n", bad, r[0], r[64], r[17], r[81], r[63], r[127]);
  return 0;
}
