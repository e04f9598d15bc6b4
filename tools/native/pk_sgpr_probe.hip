// Does a packed-fp32 VALU instruction honour op_sel_hi on an SGPR-pair source?  hipcc (ROCm 7.2, gfx950) emits  v_pk_fma_f32 v[a:b], v[c:d], s[n:n+1], v[e:f] op_sel_hi:[1,0,1] neg_lo:[1,0,0]
// neg_hi:[1,0,0]  with only s[n] holding the splat constant and s[n+1] holding unrelated live data (csrc/sstage.hip, LayerNorm combine, round 5).  This probe issues those forms.
//   hipcc --offload-arch=gfx950 -o pk_sgpr_probe pk_sgpr_probe.hip && ./pk_sgpr_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void probe(float* out, float lo, float hi) {
  float r[6];
  asm volatile(
      "s_mov_b32 s20, %6\n s_mov_b32 s21, %7\n"
      "v_mov_b32 v10, 1.0\n v_mov_b32 v11, 1.0\n v_mov_b32 v16, 2.0\n v_mov_b32 v17, 2.0\n v_mov_b32 v18, 4.0\n v_mov_b32 v19, 4.0\n"
      "s_nop 4\n"
      "v_pk_mul_f32 v[12:13], v[10:11], s[20:21] op_sel_hi:[1,0]\n"
      "v_pk_mul_f32 v[14:15], v[10:11], s[20:21]\n"
      "v_pk_fma_f32 v[16:17], v[18:19], s[20:21], v[16:17] op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n"
      "s_nop 4\n"
      "v_mov_b32 %0, v12\n v_mov_b32 %1, v13\n v_mov_b32 %2, v14\n v_mov_b32 %3, v15\n v_mov_b32 %4, v16\n v_mov_b32 %5, v17\n"
      : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]), "=v"(r[4]), "=v"(r[5]) : "s"(lo), "s"(hi)
      : "s20", "s21", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19");
  if (threadIdx.x == 0) for (int i = 0; i < 6; ++i) out[i] = r[i];
}
int main() {
  float* d; float h[6];
  (void)hipMalloc(&d, 24);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, 0.25f, 777.0f);
  (void)hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
  printf("s[20:21] = (0.25, 777)\n");
  printf("v_pk_mul_f32 1.0 * s[20:21] op_sel_hi:[1,0]                       -> lo %g hi %g   (compiler expects 0.25 0.25)\n", h[0], h[1]);
  printf("v_pk_mul_f32 1.0 * s[20:21] (default op_sel_hi)                   -> lo %g hi %g   (0.25 777)\n", h[2], h[3]);
  printf("v_pk_fma_f32 -4 * s[20:21] + 2 op_sel_hi:[1,0,1] neg_lo/hi:[1,0,0] -> lo %g hi %g   (compiler expects 1 1)\n", h[4], h[5]);
  return 0;
}
