// Probe: v_dot2c_f32_bf16 / v_dot2_f32_bf16 / the builtin on (w, 0) and (0, w) pairs (the depth-wise taps of csrc/sstage.hip, dstage.hip); prints the three results beside the expected value.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
__global__ void k(const unsigned* a, const unsigned* b, float* o) {
  float acc = 1.0f, acc2 = 1.0f, acc3 = 1.0f;
  unsigned x = a[threadIdx.x], w = b[threadIdx.x];
  asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(acc) : "v"(x), "v"(w));
  asm volatile("v_dot2_f32_bf16 %0, %1, %2, %0" : "+v"(acc2) : "v"(x), "v"(w));
  acc3 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, x), __builtin_bit_cast(bf2, w), acc3, false);
  o[3 * threadIdx.x] = acc; o[3 * threadIdx.x + 1] = acc2; o[3 * threadIdx.x + 2] = acc3;
}
static unsigned short bf(float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)(u >> 16); }
int main() {
  float xs[4][2] = {{2.f, 3.f}, {2.f, 3.f}, {2.f, 3.f}, {1e30f, -5.f}};
  float ws[4][2] = {{10.f, 0.f}, {0.f, 10.f}, {10.f, 100.f}, {0.f, 2.f}};
  unsigned ha[4], hb[4];
  for (int i = 0; i < 4; ++i) { ha[i] = bf(xs[i][0]) | (bf(xs[i][1]) << 16); hb[i] = bf(ws[i][0]) | (bf(ws[i][1]) << 16); }
  unsigned *a, *b; float* o; hipMalloc(&a, 16); hipMalloc(&b, 16); hipMalloc(&o, 48);
  hipMemcpy(a, ha, 16, hipMemcpyHostToDevice); hipMemcpy(b, hb, 16, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(4), 0, 0, a, b, o);
  float ho[12]; hipMemcpy(ho, o, 48, hipMemcpyDeviceToHost);
  for (int i = 0; i < 4; ++i) printf("x=(%g,%g) w=(%g,%g): dot2c %g  dot2 %g  builtin %g   expect %g\n", xs[i][0], xs[i][1], ws[i][0], ws[i][1], ho[3 * i], ho[3 * i + 1], ho[3 * i + 2], 1 + xs[i][0] * ws[i][0] + xs[i][1] * ws[i][1]);
  return 0;
}
