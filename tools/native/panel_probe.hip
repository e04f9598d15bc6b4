// panel_probe.hip -- experiment behind the "panel" GEMM engine of round 2.
//
// Hypothesis (round-1 PMC evidence): the 128x128 / 2-barrier k-loop is bound by the L2 -> LDS operand stream at 64 flop per
// streamed byte.  Here a workgroup OWNS a panel of up to 128 rows of A for the whole launch (A stationary: in LDS, variant L, or
// in registers as MFMA fragments, variant R) and streams only W through a ring of 16 KB slots [128 n][64 k] with counted
// vmcnt and ONE raw s_barrier per slot: 128 flop per streamed byte, the W stream never stops across n-tiles, 8 waves
// (2 per SIMD) of 32 x 64 so that the epilogue of one wave hides under the MFMAs of its SIMD partner.
//
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 panel_probe.hip -o panel_probe     run: ./panel_probe [M N K]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef unsigned short bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((address_space(3))) void* lds_vp;
typedef const __attribute__((address_space(1))) void* glb_vp;

__device__ __forceinline__ unsigned pack_bf2(float lo, float hi) {
  const f32x2_t f = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2_t));
}

struct PArgs {
  const bf16_t* A; const bf16_t* W; const float* bias; bf16_t* C;
  int M, N, K;
};

constexpr int SLOT = 16384;

// one W slot [128 n][64 k]: 16 segments of 1 KiB, two per wave; chunk swizzle kc ^ ((row >> 1) & 7) on the SOURCE address
__device__ __forceinline__ void issue_slot(unsigned char* dst, const bf16_t* __restrict__ W, int N, int K, int nt, int ks, int lane, int wave) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int sg = wave + i * 8, r = sg * 8 + (lane >> 3), p = lane & 7, kc = p ^ ((r >> 1) & 7);
    const bf16_t* src = W + (int64_t)min(nt * 128 + r, N - 1) * K + ks * 64 + kc * 8;
    __builtin_amdgcn_global_load_lds((glb_vp)src, (lds_vp)(dst + sg * 1024), 16, 0, 0);
  }
}

__device__ __forceinline__ bf16x8_t bfrag(const unsigned char* sb, int r, int kc) {
  return *reinterpret_cast<const bf16x8_t*>(sb + r * 128 + ((kc ^ ((r >> 1) & 7)) << 4));
}

__device__ __forceinline__ void row_range(int M, int* r0, int* R) {
  const int T16 = (M + 15) >> 4, nwg = gridDim.x, base = T16 / nwg, rem = T16 % nwg, b = blockIdx.x;
  const int start = b * base + min(b, rem), cnt = base + (b < rem ? 1 : 0);
  *r0 = start * 16;
  *R = min(cnt * 16, M - start * 16);
}

template <int NS> __device__ __forceinline__ void wait_slot() {
  if constexpr (NS == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (NS == 3) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if constexpr (NS == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (NS == 5) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if constexpr (NS == 6) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");   // NS == 8
}

// bias through an inline-asm LDS read: hipcc orders every LDS read it can see behind the pending LDS-DMA writes of the W ring with
// s_waitcnt vmcnt(0) (it cannot prove that the bias array and the ring do not alias), which would drain the ring at every epilogue
__device__ __forceinline__ float4 lds_read16(const float* p) {
  float4 v;
  const unsigned a = (unsigned)(uintptr_t)p;
  asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
  return v;
}

template <int NS> __device__ __forceinline__ void wait_slot_st() {      // wait_slot + 8 younger stores allowed
  if constexpr (NS == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if constexpr (NS == 3) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  else if constexpr (NS == 4) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else if constexpr (NS == 5) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
  else if constexpr (NS == 6) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
}

__device__ __forceinline__ void store_tile(const f32x4_t (&acc)[2][4], const PArgs& g, const float* sBias, int r0, int R, int nt, int wm, int wn, int lane) {
  const bool full = (wm * 32 + 32 <= R) && (nt * 128 + 128 <= g.N);      // wave-uniform: no per-store exec masks on the common path
#pragma unroll
  for (int tj = 0; tj < 4; ++tj) {
    const int n = nt * 128 + wn * 64 + tj * 16 + (lane >> 4) * 4;
    const float4 b4 = lds_read16(sBias + min(n, g.N - 4));
#pragma unroll
    for (int ti = 0; ti < 2; ++ti) {
      const int ml = wm * 32 + ti * 16 + (lane & 15);
      const f32x4_t a = acc[ti][tj];
      uint2 o;
      o.x = pack_bf2(a[0] + b4.x, a[1] + b4.y);
      o.y = pack_bf2(a[2] + b4.z, a[3] + b4.w);
      if (full || (ml < R && n < g.N)) *reinterpret_cast<uint2*>(g.C + (int64_t)(r0 + ml) * g.N + n) = o;
    }
  }
}

// ---- variant L: A panel in LDS ---------------------------------------------------------------------------------------
template <int K, int NS, int ABL = 0>
__global__ __launch_bounds__(512, 2) void panel_fwd_lds(const PArgs g) {
  constexpr int KS = K / 64, ROWB = K * 2, A_BYTES = 128 * ROWB;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sA = smem;
  unsigned char* sB = smem + A_BYTES;
  float* sBias = reinterpret_cast<float*>(smem + A_BYTES + NS * SLOT);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  int r0, R;
  row_range(g.M, &r0, &R);
  if (R <= 0) return;
  for (int i = tid; i < g.N; i += 512) sBias[i] = g.bias[i];
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  // A panel: lane-linear LDS image, row stride ROWB, 16-byte chunk c of row r stored at c ^ (r & 15)
#pragma unroll
  for (int i = 0; i < A_BYTES / 1024 / 8; ++i) {
    const int sg = wave + i * 8, p = sg * 1024 + lane * 16, row = p / ROWB, pc = (p % ROWB) >> 4, kc = pc ^ (row & 15);
    const bf16_t* src = g.A + (int64_t)(r0 + min(row, R - 1)) * K + kc * 8;
    __builtin_amdgcn_global_load_lds((glb_vp)src, (lds_vp)(sA + sg * 1024), 16, 0, 0);
  }
  const int NT = (g.N + 127) >> 7, S = NT * KS;
#pragma unroll
  for (int s = 0; s < NS - 1; ++s) issue_slot(sB + s * SLOT, g.W, g.N, K, (s / KS) % NT, s % KS, lane, wave);
  int nxt_nt = ((NS - 1) / KS) % NT, nxt_ks = (NS - 1) % KS, nxt_ph = (NS - 1) % NS, cur_ph = 0;
  const int arow = wm * 32 + (lane & 15), akc = lane >> 4;
  for (int nt = 0; nt < NT; ++nt) {
    f32x4_t acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if constexpr ((ABL & 8) != 0) {
        // stores of the previous tile's epilogue (8 per wave) sit in the same in-order counter BEHIND the ring's older loads:
        // allow them to stay in flight for the first NS - 1 slots after an epilogue
        if (ks < NS - 1 && nt > 0) wait_slot_st<NS>(); else wait_slot<NS>();
      } else {
        wait_slot<NS>();
      }
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if constexpr ((ABL & 2) == 0) issue_slot(sB + nxt_ph * SLOT, g.W, g.N, K, nxt_nt, nxt_ks, lane, wave);
      nxt_ph = (nxt_ph + 1 == NS) ? 0 : nxt_ph + 1;
      if (++nxt_ks == KS) { nxt_ks = 0; nxt_nt = (nxt_nt + 1 == NT) ? 0 : nxt_nt + 1; }
      const unsigned char* sb = sB + cur_ph * SLOT;
      cur_ph = (cur_ph + 1 == NS) ? 0 : cur_ph + 1;
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        bf16x8_t af[2], bfr[4];
#pragma unroll
        for (int tj = 0; tj < 4; ++tj) bfr[tj] = bfrag(sb, wn * 64 + tj * 16 + (lane & 15), hh * 4 + (lane >> 4));
#pragma unroll
        for (int ti = 0; ti < 2; ++ti) {
          const int row = arow + ti * 16, kcl = ks * 8 + hh * 4 + akc;
          af[ti] = *reinterpret_cast<const bf16x8_t*>(sA + row * ROWB + ((kcl ^ (row & 15)) << 4));
        }
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
          for (int tj = 0; tj < 4; ++tj) {
            if constexpr ((ABL & 4) == 0) acc[ti][tj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[tj], af[ti], acc[ti][tj], 0, 0, 0);
            else { asm volatile("" :: "v"(bfr[tj]), "v"(af[ti])); }
          }
      }
    }
    if constexpr ((ABL & 1) == 0) store_tile(acc, g, sBias, r0, R, nt, wm, wn, lane);
    else if (g.M < 0) store_tile(acc, g, sBias, r0, R, nt, wm, wn, lane);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---- variant R: A panel in registers (MFMA fragments), LDS holds only the W ring --------------------------------------------
template <int K, int NS, int ABL = 0>
__global__ __launch_bounds__(512, 2) void panel_fwd_reg(const PArgs g) {
  constexpr int KS = K / 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sB = smem;
  float* sBias = reinterpret_cast<float*>(smem + NS * SLOT);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  int r0, R;
  row_range(g.M, &r0, &R);
  if (R <= 0) return;
  for (int i = tid; i < g.N; i += 512) sBias[i] = g.bias[i];
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  bf16x8_t af[2][KS * 2];
#pragma unroll
  for (int ti = 0; ti < 2; ++ti) {
    const int row = min(wm * 32 + ti * 16 + (lane & 15), R - 1);
#pragma unroll
    for (int kk = 0; kk < KS * 2; ++kk)
      af[ti][kk] = *reinterpret_cast<const bf16x8_t*>(g.A + (int64_t)(r0 + row) * K + kk * 32 + (lane >> 4) * 8);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int ti = 0; ti < 2; ++ti)
#pragma unroll
    for (int kk = 0; kk < KS * 2; ++kk) asm volatile("" : "+v"(af[ti][kk]));
  const int NT = (g.N + 127) >> 7;
#pragma unroll
  for (int s = 0; s < NS - 1; ++s) issue_slot(sB + s * SLOT, g.W, g.N, K, (s / KS) % NT, s % KS, lane, wave);
  int nxt_nt = ((NS - 1) / KS) % NT, nxt_ks = (NS - 1) % KS, nxt_ph = (NS - 1) % NS, cur_ph = 0;
  for (int nt = 0; nt < NT; ++nt) {
    f32x4_t acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if constexpr ((ABL & 8) != 0) {
        // stores of the previous tile's epilogue (8 per wave) sit in the same in-order counter BEHIND the ring's older loads:
        // allow them to stay in flight for the first NS - 1 slots after an epilogue
        if (ks < NS - 1 && nt > 0) wait_slot_st<NS>(); else wait_slot<NS>();
      } else {
        wait_slot<NS>();
      }
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if constexpr ((ABL & 2) == 0) issue_slot(sB + nxt_ph * SLOT, g.W, g.N, K, nxt_nt, nxt_ks, lane, wave);
      nxt_ph = (nxt_ph + 1 == NS) ? 0 : nxt_ph + 1;
      if (++nxt_ks == KS) { nxt_ks = 0; nxt_nt = (nxt_nt + 1 == NT) ? 0 : nxt_nt + 1; }
      const unsigned char* sb = sB + cur_ph * SLOT;
      cur_ph = (cur_ph + 1 == NS) ? 0 : cur_ph + 1;
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        bf16x8_t bfr[4];
#pragma unroll
        for (int tj = 0; tj < 4; ++tj) bfr[tj] = bfrag(sb, wn * 64 + tj * 16 + (lane & 15), hh * 4 + (lane >> 4));
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
          for (int tj = 0; tj < 4; ++tj) {
            if constexpr ((ABL & 4) == 0) acc[ti][tj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[tj], af[ti][ks * 2 + hh], acc[ti][tj], 0, 0, 0);
            else { asm volatile("" :: "v"(bfr[tj]), "v"(af[ti][ks * 2 + hh])); }
          }
      }
    }
    if constexpr ((ABL & 1) == 0) store_tile(acc, g, sBias, r0, R, nt, wm, wn, lane);
    else if (g.M < 0) store_tile(acc, g, sBias, r0, R, nt, wm, wn, lane);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}


// ---- family II: C-stationary (long reduction, narrow output: fc2 forward, dX of qkv / fc1) ------------------------------------
// The workgroup owns 128 rows x ALL N <= NSW * 128 output columns (accumulators: 32 x NSW * 64 per wave); a k-step streams one A
// slot [128 rows][64 k] and NSW W slots [128 n][64 k] (96 flop per streamed byte at N = 384), RD k-steps in the ring, one barrier
// per k-step (48 MFMAs per wave at NSW = 3).
template <int NSW, int RD>
__global__ __launch_bounds__(512, 2) void panel_fwd_cstat(const PArgs g) {
  constexpr int STEP = (1 + NSW) * SLOT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* sBias = reinterpret_cast<float*>(smem + RD * STEP);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  int r0, R;
  row_range(g.M, &r0, &R);
  if (R <= 0) return;
  for (int i = tid; i < g.N; i += 512) sBias[i] = g.bias[i];
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  const int K = g.K, KS = K >> 6;
  auto issue_step = [&](int ph, int ks) {
    unsigned char* dst = smem + ph * STEP;
#pragma unroll
    for (int i = 0; i < 2; ++i) {          // A slot: rows of this panel (clamped), same image as a W slot
      const int sg = wave + i * 8, r = sg * 8 + (lane >> 3), p = lane & 7, kc = p ^ ((r >> 1) & 7);
      const bf16_t* src = g.A + (int64_t)(r0 + min(r, R - 1)) * K + ks * 64 + kc * 8;
      __builtin_amdgcn_global_load_lds((glb_vp)src, (lds_vp)(dst + sg * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < NSW; ++j) issue_slot(dst + (1 + j) * SLOT, g.W, g.N, K, j, ks, lane, wave);
  };
  f32x4_t acc[2][NSW * 4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NSW * 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < RD - 1; ++s) issue_step(s, s % KS);
  int nxt = (RD - 1) % KS;
  for (int ks0 = 0; ks0 < KS; ks0 += RD) {
#pragma unroll
    for (int u = 0; u < RD; ++u) {
      if (ks0 + u < KS) {
        if constexpr (RD == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((RD - 2) * 2 * (1 + NSW)) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        issue_step((u + RD - 1) % RD, nxt);
        nxt = (nxt + 1 == KS) ? 0 : nxt + 1;
        const unsigned char* sa = smem + u * STEP;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          bf16x8_t af[2];
#pragma unroll
          for (int ti = 0; ti < 2; ++ti) af[ti] = bfrag(sa, wm * 32 + ti * 16 + (lane & 15), hh * 4 + (lane >> 4));
#pragma unroll
          for (int j = 0; j < NSW; ++j) {
            bf16x8_t bfr[4];
#pragma unroll
            for (int tj = 0; tj < 4; ++tj) bfr[tj] = bfrag(sa + (1 + j) * SLOT, wn * 64 + tj * 16 + (lane & 15), hh * 4 + (lane >> 4));
#pragma unroll
            for (int ti = 0; ti < 2; ++ti)
#pragma unroll
              for (int tj = 0; tj < 4; ++tj) acc[ti][j * 4 + tj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[tj], af[ti], acc[ti][j * 4 + tj], 0, 0, 0);
          }
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int j = 0; j < NSW; ++j) {
    f32x4_t t[2][4];
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
      for (int tj = 0; tj < 4; ++tj) t[ti][tj] = acc[ti][j * 4 + tj];
    store_tile(t, g, sBias, r0, R, j, wm, wn, lane);
  }
}

// ---- host ---------------------------------------------------------------------------------------------------------------
static bf16_t f2bf_h(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (bf16_t)(u >> 16); }
static float bf2f_h(bf16_t v) { unsigned u = ((unsigned)v) << 16; float f; memcpy(&f, &u, 4); return f; }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

struct Ctx {
  int M, N, K; std::vector<bf16_t> hA, hW; std::vector<float> hb; bf16_t *dA, *dW, *dC; float* db; hipEvent_t e0, e1;
};

template <typename F>
static void bench(const char* name, Ctx& c, int grid, int lds, F launch) {
  CK(hipMemset(c.dC, 0xff, (size_t)c.M * c.N * 2));
  launch();
  CK(hipDeviceSynchronize());
  CK(hipGetLastError());
  std::vector<bf16_t> hC((size_t)c.M * c.N);
  CK(hipMemcpy(hC.data(), c.dC, hC.size() * 2, hipMemcpyDeviceToHost));
  // check: 40 rows (first / last of the launch, panel borders, random) x all columns, fp64 reference
  double maxerr = 0, maxref = 0;
  std::vector<int> rows = {0, 1, 15, 16, 95, 96, 111, 112, 127, 128, c.M - 1, c.M - 2, c.M - 17, c.M / 2, c.M / 3};
  for (int i = 0; i < 25; ++i) rows.push_back((int)((1103515245u * (unsigned)(i + 7) + 12345u) % (unsigned)c.M));
  for (int m : rows) {
    if (m < 0 || m >= c.M) continue;
    for (int n = 0; n < c.N; ++n) {
      double s = c.hb[n];
      for (int k = 0; k < c.K; ++k) s += (double)bf2f_h(c.hA[(size_t)m * c.K + k]) * (double)bf2f_h(c.hW[(size_t)n * c.K + k]);
      const double got = bf2f_h(hC[(size_t)m * c.N + n]);
      maxerr = fmax(maxerr, fabs(got - s)); maxref = fmax(maxref, fabs(s));
    }
  }
  for (int i = 0; i < 5; ++i) launch();
  CK(hipEventRecord(c.e0));
  const int it = 30;
  for (int i = 0; i < it; ++i) launch();
  CK(hipEventRecord(c.e1));
  CK(hipEventSynchronize(c.e1));
  float ms; CK(hipEventElapsedTime(&ms, c.e0, c.e1));
  const double us = ms * 1e3 / it, tf = 2.0 * c.M * c.N * c.K / us / 1e6;
  printf("%-34s grid %4d lds %6d : %8.1f us %8.1f TFLOP/s   rel err %.2e %s\n", name, grid, lds, us, tf, maxerr / maxref, maxerr / maxref < 6e-3 ? "ok" : "WRONG");
  fflush(stdout);
}

template <int K, int NS, int ABL = 0>
static void run_lds(Ctx& c, int grid) {
  const int lds = 128 * K * 2 + NS * SLOT + ((c.N + 3) / 4 * 4) * 4;
  if (lds > 163840) { printf("panel_fwd_lds<%d,%d>: %d bytes of LDS do not fit\n", K, NS, lds); return; }
  auto kern = panel_fwd_lds<K, NS, ABL>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  PArgs g{c.dA, c.dW, c.db, c.dC, c.M, c.N, c.K};
  char nm[64]; snprintf(nm, sizeof nm, "L  A-in-LDS  K=%d ring=%d abl=%d", K, NS, ABL);
  bench(nm, c, grid, lds, [&] { hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, 0, g); });
}
template <int K, int NS, int ABL = 0>
static void run_reg(Ctx& c, int grid) {
  const int lds = NS * SLOT + ((c.N + 3) / 4 * 4) * 4;
  auto kern = panel_fwd_reg<K, NS, ABL>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  PArgs g{c.dA, c.dW, c.db, c.dC, c.M, c.N, c.K};
  char nm[64]; snprintf(nm, sizeof nm, "R  A-in-regs K=%d ring=%d abl=%d", K, NS, ABL);
  bench(nm, c, grid, lds, [&] { hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, 0, g); });
}


template <int NSW, int RD>
static void run_cstat(Ctx& c, int grid) {
  if (c.N > NSW * 128 || c.N <= (NSW - 1) * 128) return;
  const int lds = RD * (1 + NSW) * SLOT + ((c.N + 3) / 4 * 4) * 4;
  if (lds > 163840) { printf("panel_fwd_cstat<%d,%d>: %d bytes of LDS do not fit\n", NSW, RD, lds); return; }
  auto kern = panel_fwd_cstat<NSW, RD>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  PArgs g{c.dA, c.dW, c.db, c.dC, c.M, c.N, c.K};
  char nm[64]; snprintf(nm, sizeof nm, "C  C-stationary NSW=%d ring=%d", NSW, RD);
  bench(nm, c, grid, lds, [&] { hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, 0, g); });
}

int main(int argc, char** argv) {
  Ctx c;
  c.M = argc > 1 ? atoi(argv[1]) : 27136; c.N = argc > 2 ? atoi(argv[2]) : 1536; c.K = argc > 3 ? atoi(argv[3]) : 384;
  printf("panel probe: M=%d N=%d K=%d bf16\n", c.M, c.N, c.K);
  c.hA.resize((size_t)c.M * c.K); c.hW.resize((size_t)c.N * c.K); c.hb.resize(c.N);
  unsigned st = 12345u;
  auto rnd = [&] { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xffff) / 32768.0f - 1.0f; };
  for (auto& v : c.hA) v = f2bf_h(rnd());
  for (auto& v : c.hW) v = f2bf_h(rnd() * 0.05f);
  for (auto& v : c.hb) v = rnd();
  CK(hipMalloc(&c.dA, c.hA.size() * 2)); CK(hipMalloc(&c.dW, c.hW.size() * 2)); CK(hipMalloc(&c.dC, (size_t)c.M * c.N * 2)); CK(hipMalloc(&c.db, c.N * 4));
  CK(hipMemcpy(c.dA, c.hA.data(), c.hA.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(c.dW, c.hW.data(), c.hW.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(c.db, c.hb.data(), c.N * 4, hipMemcpyHostToDevice));
  CK(hipEventCreate(&c.e0)); CK(hipEventCreate(&c.e1));
  const int T16 = (c.M + 15) / 16;
  const int g128 = (T16 + 7) / 8;                                  // 128-row panels
  const int g256 = T16 >= 256 * 5 ? (T16 + 7) / 8 > 256 ? g128 : 256 : g128;      // <= 128 rows on exactly 256 workgroups when that fits
  if (c.N <= 512 && c.K % 64 == 0) {
    run_cstat<1, 2>(c, g128); run_cstat<1, 3>(c, g128); run_cstat<1, 3>(c, g256);
    run_cstat<2, 2>(c, g128); run_cstat<2, 3>(c, g128); run_cstat<2, 3>(c, g256);
    run_cstat<3, 2>(c, g128); run_cstat<3, 2>(c, g256);
    run_cstat<4, 2>(c, g128); run_cstat<4, 2>(c, g256);
  }
  if (c.K == 384) {
    run_lds<384, 3>(c, g256); run_lds<384, 3, 8>(c, g256); run_lds<384, 3, 1>(c, g256); run_lds<384, 3, 2>(c, g256); run_lds<384, 3, 4>(c, g256); run_lds<384, 3, 3>(c, g256);
    run_reg<384, 4>(c, g256); run_reg<384, 4, 8>(c, g256); run_reg<384, 6, 8>(c, g256); run_reg<384, 4, 1>(c, g256); run_reg<384, 4, 2>(c, g256); run_reg<384, 4, 4>(c, g256); run_reg<384, 4, 3>(c, g256);
    run_reg<384, 4, 5>(c, g256); run_reg<384, 4, 6>(c, g256);
  } else if (c.K == 192) {
    run_lds<192, 3>(c, g128); run_lds<192, 4>(c, g128); run_lds<192, 6>(c, g128);
    run_reg<192, 3>(c, g128); run_reg<192, 4>(c, g128); run_reg<192, 6>(c, g128);
  } else if (c.K == 512) {
    run_reg<512, 4>(c, g128); run_reg<512, 4>(c, g256);
  } else {
    printf("unsupported K\n");
  }
  return 0;
}
