// rs_probe.hip -- round-3 GEMM experiment: "register-stationary token panel" kernel for short reductions (K <= 384).
//
//   out[r, n] = sum_k A[r, k] W[n, k]  (+ bias, GELU / residual epilogues)
//
// A workgroup of 4 waves owns a 128-row token panel: every wave keeps ITS 32 rows x K as MFMA operand fragments in registers
// (2 x KS x 4 VGPRs, loaded once per panel straight from global memory).  Only the weights stream: 64 output columns x K per
// "slot", through a 3-deep LDS ring of 24 KB steps (6 k-steps of 32) filled by LDS-DMA; every W fragment read from LDS feeds two
// MFMAs, the token operand costs no LDS traffic at all.  A slot ends with its own small epilogue: the tile -> column mapping is
// chosen so that a lane holds 16 (or 2 x 8) CONSECUTIVE output columns -- 16-byte stores straight from the accumulators, no LDS
// transpose, 4 store instructions per wave every 96 MFMAs, i.e. the output leaves as a smooth stream instead of one burst per
// 128 x 128 tile.  Two independent 4-wave workgroups share a CU (2 waves per SIMD that are NOT coupled by a barrier).
// Work is a flat list of (panel, slot) pairs cut into equal contiguous ranges, one per workgroup (512 = 2 per CU).
//
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 rs_probe.hip -o rs_probe [-L../../lemevit_amd/csrc -llemevit_hip -DWITH_LMV]
// run:   ./rs_probe M N K
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <utility>
#ifdef WITH_LMV
#include "../../include/lemevit_hip.h"
#endif

typedef unsigned short bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((address_space(3))) void* lds_vp;
typedef const __attribute__((address_space(1))) void* glb_vp;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ unsigned pack_bf2(float lo, float hi) {
  const f32x2_t f = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2_t));
}
__device__ __forceinline__ float gelu_fast(float x) {
  const float z = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
  const float ex = __expf(-z * z);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  return 0.5f * x * (1.0f + copysignf(1.0f - poly * ex, x));
}

struct RSArgs {
  const bf16_t* A; const bf16_t* W; const float* bias; bf16_t* C; bf16_t* Cpre; const bf16_t* res; const float* row_scale;
  int M, N, K, rps;
  int npanels, nslots;     // 128-row panels, 64-column slots per panel
  long long total;         // npanels * nslots
  int nwg;
  unsigned long long* dbg;   // 8 stamps per workgroup (wave 0)
};

enum { EPI_BIAS = 0, EPI_GELU_PRE = 1, EPI_RES = 2, EPI_NOSTORE = 3 };

template <int N_> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory"); }
template <int N_> __device__ __forceinline__ void wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N_) : "memory"); }

// Vector-memory operations of the main loop go through inline asm: next to LDS-DMA requests hipcc waits vmcnt(0) for every load it can
// see (and at every loop back edge), which would drain the ring; the waits are counted by hand (loads and stores retire in order).
__device__ __forceinline__ f32x4_t gload16(const void* p) {
  f32x4_t v;
  asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ void gstore16(void* p, const f32x4_t& v) {
  asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ f32x4_t pack8(const float* v) {
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
  const u32x4_t u = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7])};
  return __builtin_bit_cast(f32x4_t, u);
}

// slot-local W row (= output column offset) that MFMA tile t reads at A-operand index i (lane & 15)
template <int CM> __device__ __forceinline__ int nmap(int t, int i) {
  if (CM == 0) return 16 * (i >> 2) + 4 * t + (i & 3);                                   // lane group g: columns 16 g .. 16 g + 15
  return 32 * (t >> 1) + 8 * (i >> 2) + 4 * (t & 1) + (i & 3);                           // lane group g: columns 8 g .. + 7 and 32 + 8 g .. + 7
}
template <int CM> __device__ __forceinline__ int swz(int n) {                           // 16-byte chunk swizzle of row n of a [64][128 B] k-tile
  if (CM == 0) return (((n >> 1) & 1) << 1) | (((n >> 4) & 1) << 2);
  return (((n >> 1) & 1) << 1) | (((n >> 3) & 1) << 2);
}

template <int KS, int EPI, int CM, int ABL = 0>
__global__ __launch_bounds__(256, 2) void rs_gemm_kernel(const RSArgs g) {
  constexpr int KST = 6;                    // k-steps (of 32) per ring step
  static_assert(KS % KST == 0, "K must be a multiple of 192");
  constexpr int SPS = KS / KST;             // ring steps per slot
  constexpr int STEP_BYTES = 64 * KST * 64; // 24 KB
  constexpr int NDMA = STEP_BYTES / 1024 / 4;
  constexpr int NL = (EPI == EPI_RES) ? 8 : 4;      // epilogue operand loads per slot (bias 4, residual 4)
  constexpr int NS_ = (EPI == EPI_NOSTORE) ? 0 : (EPI == EPI_GELU_PRE ? 8 : 4);   // stores per slot
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned lds0 = (unsigned)(uintptr_t)smem;
  const int K = KS * 32;

  const long long s_begin = g.total * blockIdx.x / g.nwg, s_end = g.total * (blockIdx.x + 1) / g.nwg;
  const int nsl = (int)(s_end - s_begin);
  if (nsl <= 0) return;
  const int T = nsl * SPS;

  // per-lane fragment read offsets inside a k64 tile: [t][h]
  unsigned roff[4][2];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int n = nmap<CM>(t, lane & 15);
#pragma unroll
    for (int h = 0; h < 2; ++h) roff[t][h] = n * 128 + (((4 * h + (lane >> 4)) ^ swz<CM>(n)) << 4);
  }
  // DMA: instruction i of a step covers rows 8 * ((wave + 4 i) & 7) .. + 7 of k64 tile (wave + 4 i) >> 3, i.e. row n_a = 8 wave + (lane >> 3)
  // (+ 32 for odd i) of k64 tile i >> 1; physical chunk pc = lane & 7 of row n holds logical chunk pc ^ swz(n)  (swz(n) == swz(n + 32))
  const int n_a = 8 * wave + (lane >> 3);
  const long long lane_off = (long long)n_a * K + (((lane & 7) ^ swz<CM>(n_a)) << 3);
  auto issue_dma = [&](const bf16_t* wsrc, int dbuf, int i) {      // wsrc = W + (slot * 64) * K + part * 192
    const bf16_t* src = wsrc + lane_off + (i & 1) * 32 * K + (i >> 1) * 64;
    const unsigned dst = dbuf * STEP_BYTES + (wave + 4 * i) * 1024;
    __builtin_amdgcn_global_load_lds((glb_vp)src, (lds_vp)(smem + dst), 16, 0, 0);
  };
  auto step_src = [&](int t) -> const bf16_t* {                   // (only for the two prologue steps)
    const int s = (int)((s_begin + t / SPS) % g.nslots), part = t % SPS;
    return g.W + (long long)s * 64 * K + part * (KST * 32);
  };

  bf16x8_t af[2][KS];
  float rsc[2] = {1.f, 1.f};
  auto load_panel = [&](int panel) {
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      const int row = min(panel * 128 + wave * 32 + rt * 16 + (lane & 15), g.M - 1);
      if (EPI == EPI_RES && g.row_scale) rsc[rt] = g.row_scale[row / g.rps];
      const bf16_t* p = g.A + (long long)row * K + (lane >> 4) * 8;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) af[rt][ks] = *reinterpret_cast<const bf16x8_t*>(p + ks * 32);
    }
  };

  unsigned long long tstamp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  tstamp[0] = __builtin_amdgcn_s_memtime();
  int panel = (int)(s_begin / g.nslots);
  load_panel(panel);
#pragma unroll
  for (int i = 0; i < NDMA; ++i) issue_dma(step_src(0), 0, i);
  if (T > 1) {
#pragma unroll
    for (int i = 0; i < NDMA; ++i) issue_dma(step_src(1), 1, i);
    wait_vm<NDMA>();
  } else {
    wait_vm<0>();
  }
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(af[rt][ks]));
  tstamp[1] = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  tstamp[2] = __builtin_amdgcn_s_memtime();

  f32x4_t acc[2][4];
  f32x4_t bias4[4], res4[2][2];
  int buf = 0, dbuf = 2;                    // ring buffer of the step being computed / of the step being requested (t + 2)
  int s2 = (int)((s_begin + 2 / SPS) % g.nslots);      // slot of step t + 2
  for (int sl = 0; sl < nsl; ++sl) {
    const long long sg = s_begin + sl;
    const int pnl = (int)(sg / g.nslots), s = (int)(sg % g.nslots);
    if (pnl != panel) {                     // next token panel: reload the register-resident rows (ring keeps its landed steps)
      panel = pnl;
      tstamp[3] = __builtin_amdgcn_s_memtime();
      load_panel(panel);
      wait_vm<0>();
      tstamp[4] = __builtin_amdgcn_s_memtime();
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(af[rt][ks]));
    }
    const int col0 = s * 64;
    // epilogue operands of this slot: bias of the lane's 16 columns (-> initial accumulators), residual chunks
    int cbase[2];
    if (CM == 0) { cbase[0] = col0 + 16 * (lane >> 4); cbase[1] = cbase[0] + 8; }
    else { cbase[0] = col0 + 8 * (lane >> 4); cbase[1] = cbase[0] + 32; }
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      if (ABL & 16) { bias4[2 * hh] = f32x4_t{0.f, 0.f, 0.f, 0.f}; bias4[2 * hh + 1] = f32x4_t{0.f, 0.f, 0.f, 0.f}; continue; }
      bias4[2 * hh] = gload16(g.bias + cbase[hh]);
      bias4[2 * hh + 1] = gload16(g.bias + cbase[hh] + 4);
    }
    const int row0 = pnl * 128 + wave * 32 + (lane & 15);
    if (EPI == EPI_RES) {
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) res4[rt][hh] = gload16(g.res + (long long)min(row0 + rt * 16, g.M - 1) * g.N + cbase[hh]);
    }
#pragma unroll
    for (int part = 0; part < SPS; ++part) {
      const int t = sl * SPS + part;
      const bool more = t + 2 < T;          // wave-uniform
      const unsigned bbase = lds0 + buf * STEP_BYTES;
      unsigned ra[4][2];
#pragma unroll
      for (int tt = 0; tt < 4; ++tt)
#pragma unroll
        for (int h = 0; h < 2; ++h) ra[tt][h] = bbase + roff[tt][h];
      bf16x8_t wf[2][4];
      auto rd = [&](int set, int ksl) {     // fragments of local k-step ksl (0..5): k64 tile ksl / 2, half ksl & 1
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
          const unsigned a = ra[tt][ksl & 1];
          switch (ksl >> 1) {
            case 0: asm volatile("ds_read_b128 %0, %1" : "=v"(wf[set][tt]) : "v"(a) : "memory"); break;
            case 1: asm volatile("ds_read_b128 %0, %1 offset:8192" : "=v"(wf[set][tt]) : "v"(a) : "memory"); break;
            default: asm volatile("ds_read_b128 %0, %1 offset:16384" : "=v"(wf[set][tt]) : "v"(a) : "memory"); break;
          }
        }
      };
      if (!(ABL & 2) || t == 0) rd(0, 0);
#pragma unroll
      for (int ksl = 0; ksl < KST; ++ksl) {
        const int set = ksl & 1;
        if (ksl + 1 < KST && (!(ABL & 2) || t == 0)) rd(set ^ 1, ksl + 1);
        if (more && !(ABL & 1)) issue_dma(g.W + (long long)s2 * 64 * K + ((SPS == 2 && part == 1) ? KST * 32 : 0), dbuf, ksl);    // NDMA == KST: one piece per k-step
        if (ksl + 1 < KST) wait_lgkm<4>(); else wait_lgkm<0>();
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) asm volatile("" : "+v"(wf[set][tt]));
        if (part == 0 && ksl == 0) {
#pragma unroll
          for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) acc[rt][tt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
          for (int tt = 0; tt < 4; ++tt)
            if (!(ABL & 8)) acc[rt][tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[set][tt], af[rt][part * KST + ksl], acc[rt][tt], 0, 0, 0);
      }
      // the next step's W pieces (this wave's) have landed; everybody is done reading this step's buffer
      if (ABL & 16) {
      } else if (ABL & 1) {
        wait_vm<0>();
      } else if (more) {
        if (SPS == 2 && part == 0 && sl != 0) wait_vm<NS_ + NL + NDMA>(); else wait_vm<NDMA>();
      } else {
        wait_vm<0>();
      }
      if (!(ABL & 4)) __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      buf = buf == 2 ? 0 : buf + 1;
      dbuf = dbuf == 2 ? 0 : dbuf + 1;
      if (SPS == 1 || part == 1) { s2 = s2 + 1 == g.nslots ? 0 : s2 + 1; }
    }
    // ---- slot epilogue: straight from the accumulators ------------------------------------------------------------------
#pragma unroll
    for (int q = 0; q < 4; ++q) asm volatile("" : "+v"(bias4[q]));          // (uses stay behind the counted wait above)
    if (EPI == EPI_RES) {
#pragma unroll
      for (int q = 0; q < 4; ++q) asm volatile("" : "+v"(res4[q >> 1][q & 1]));
    }
    if (EPI != EPI_NOSTORE) {
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        const int row = row0 + rt * 16;
        const float sc = rsc[rt];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          float v[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) { v[e] = acc[rt][2 * hh][e] + bias4[2 * hh][e]; v[4 + e] = acc[rt][2 * hh + 1][e] + bias4[2 * hh + 1][e]; }
          const long long o = (long long)((ABL & 32) ? (row & 127) : row) * g.N + cbase[hh];
          if (EPI == EPI_GELU_PRE) {
            if (row < g.M) gstore16(g.Cpre + o, pack8(v));
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = gelu_fast(v[e]);
          }
          if (EPI == EPI_RES) {
            typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
            const u32x4_t r = __builtin_bit_cast(u32x4_t, res4[rt][hh]);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              v[2 * e] = __uint_as_float(r[e] << 16) + sc * v[2 * e];
              v[2 * e + 1] = __uint_as_float(r[e] & 0xffff0000u) + sc * v[2 * e + 1];
            }
          }
          if (ABL & 64) {       // address-pattern experiment: 8 rows x 128 B per store instruction (data is garbage)
            const int r8 = pnl * 128 + wave * 32 + (rt * 2 + hh) * 8 + (lane >> 3);
            gstore16(g.C + (long long)r8 * g.N + col0 + (lane & 7) * 8, pack8(v));
          } else if (row < g.M) gstore16(g.C + o, pack8(v));
        }
      }
    } else {
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) asm volatile("" ::"v"(acc[rt][tt]));
    }
  }
  tstamp[5] = __builtin_amdgcn_s_memtime();
  if (g.dbg && tid == 0) {
#pragma unroll
    for (int i = 0; i < 6; ++i) g.dbg[blockIdx.x * 8 + i] = tstamp[i];
  }
}

// =====================================================================================================================================
// RSL: the same register-stationary scheme with a LOADER wave.  One workgroup of 8 waves per CU: waves 0..6 compute (32 token rows each,
// 224-row panels), wave 7 issues every LDS-DMA request of the W ring and is the only wave that waits for ring loads (vmcnt).  The compute
// waves' vector-memory queue then holds stores only (plus the panel / residual loads): on gfx9-class hardware loads and stores share
// vmcnt, and a wave that must wait for its ring loads also waits for every older store.
// v2: (a) the epilogue of slot s (bias, activation, conversion, 16-byte stores) is interleaved into the MFMA stream of slot s + 1
// (two accumulator sets); (b) the loader guarantees step t + 2 at the barrier that ends step t, so the first fragment reads of step
// t + 1 are issued BEFORE that barrier; (c) the next panel's rows are requested into the dead half of the panel registers during the
// last slot of a panel.  Ring: 4 steps of 24 KB.  Bias lives in LDS.
struct RSLArgs {
  const bf16_t* A; const bf16_t* W; const float* bias; bf16_t* C; bf16_t* Cpre; const bf16_t* res; const float* row_scale;
  int M, N, K, rps;
  int npanels, nslots;     // 224-row panels, 64-column slots per panel
  long long total;
  int nwg;
  unsigned long long* dbg;
};
constexpr int RSL_ROWS = 224, RSL_R = 4, RSL_STEP = 24576, RSL_BIAS_OFF = RSL_R * RSL_STEP;

template <int KS, int EPI, int CM, int ABL = 0>
__global__ __launch_bounds__(512, 2) void rsl_kernel(const RSLArgs g) {
  constexpr int KST = 6, SPS = KS / KST, K = KS * 32;
  static_assert(KS % KST == 0, "K must be a multiple of 192");
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned lds0 = (unsigned)(uintptr_t)smem;
  const long long s_begin = g.total * blockIdx.x / g.nwg, s_end = g.total * (blockIdx.x + 1) / g.nwg;
  const int nsl = (int)(s_end - s_begin);
  if (nsl <= 0) return;
  const int T = nsl * SPS;
  // bias table -> LDS (read by the epilogues through ds_read)
  for (int i = tid; i < g.N / 4; i += 512) reinterpret_cast<float4*>(smem + RSL_BIAS_OFF)[i] = reinterpret_cast<const float4*>(g.bias)[i];

  if (wave == 7) {
    // ---------------- loader ------------------------------------------------------------------------------------------------------
    int is = (int)(s_begin % g.nslots), ipart = 0, ibuf = 0;      // slot / part / ring buffer of the next step to request
    auto issue_step = [&]() {
      const bf16_t* wsrc = g.W + ipart * (KST * 32);
#pragma unroll
      for (int q = 0; q < 24; ++q) {
        const int n = 8 * (q & 7) + (lane >> 3), c = (lane & 7) ^ swz<CM>(n);
        const bf16_t* src = wsrc + (long long)min(is * 64 + n, g.N - 1) * K + (q >> 3) * 64 + c * 8;
        __builtin_amdgcn_global_load_lds((glb_vp)src, (lds_vp)(smem + ibuf * RSL_STEP + q * 1024), 16, 0, 0);
      }
      ibuf = (ibuf + 1) & (RSL_R - 1);
      if (++ipart == SPS) { ipart = 0; is = is + 1 == g.nslots ? 0 : is + 1; }
    };
    const int npre = T < 3 ? T : 3;
    for (int i = 0; i < npre; ++i) issue_step();
    if (npre >= 3) wait_vm<24>(); else wait_vm<0>();      // steps 0 and 1 have landed
    __builtin_amdgcn_s_barrier();
    for (int t = 0; t < T; ++t) {
      if (t + 3 < T) { issue_step(); wait_vm<24>(); } else { wait_vm<0>(); }      // step t + 2 has landed
      __builtin_amdgcn_s_barrier();
    }
    return;
  }

  // ---------------- compute waves ---------------------------------------------------------------------------------------------------
  unsigned roff[4][2];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int n = nmap<CM>(t, lane & 15);
#pragma unroll
    for (int h = 0; h < 2; ++h) roff[t][h] = n * 128 + (((4 * h + (lane >> 4)) ^ swz<CM>(n)) << 4);
  }
  bf16x8_t af[2][KS];
  float rsc[2] = {1.f, 1.f};
  int panel = (int)(s_begin / g.nslots);
  auto panel_ptr = [&](int pnl, int rt) -> const bf16_t* {
    const int row = min(pnl * RSL_ROWS + wave * 32 + rt * 16 + (lane & 15), g.M - 1);
    return g.A + (long long)row * K + (lane >> 4) * 8;
  };
  auto load_half = [&](int pnl, int part) {      // asm loads: counted by hand
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      const bf16_t* p = panel_ptr(pnl, rt);
#pragma unroll
      for (int ks = 0; ks < KST; ++ks) af[rt][part * KST + ks] = __builtin_bit_cast(bf16x8_t, gload16(p + (part * KST + ks) * 32));
    }
  };
  auto touch_half = [&](int part) {
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int ks = 0; ks < KST; ++ks) asm volatile("" : "+v"(af[rt][part * KST + ks]));
  };
  auto load_scale = [&](int pnl) {
    if (EPI == EPI_RES && g.row_scale) {
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) rsc[rt] = g.row_scale[min(pnl * RSL_ROWS + wave * 32 + rt * 16 + (lane & 15), g.M - 1) / g.rps];
    }
  };
  load_scale(panel);
#pragma unroll
  for (int part = 0; part < SPS; ++part) load_half(panel, part);
  wait_vm<0>();
#pragma unroll
  for (int part = 0; part < SPS; ++part) touch_half(part);
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  constexpr int RSETS = (KS / 6 == 1) ? 2 : 1;      // residual register sets (one-step slots load a slot ahead)
  f32x4_t acc[2][2][4], res4[RSETS][2][2];
  bf16x8_t wf[2][4];
  int buf = 0;
  bool fresh = false;                      // the panel registers were re-requested during the previous slot and have not been waited for
  // epilogue state of the previous slot (consumed inside the next slot's first step)
  bool have_prev = false, prev_live = false;
  int prev_cb[2] = {0, 0}, prev_rowb = 0;
  float prev_sc[2] = {1.f, 1.f};
  f32x4_t bias2[2];

  auto rd = [&](int set, unsigned bbase, int ksl) {
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
      const unsigned a = bbase + roff[tt][ksl & 1];
      switch (ksl >> 1) {
        case 0: asm volatile("ds_read_b128 %0, %1" : "=v"(wf[set][tt]) : "v"(a) : "memory"); break;
        case 1: asm volatile("ds_read_b128 %0, %1 offset:8192" : "=v"(wf[set][tt]) : "v"(a) : "memory"); break;
        default: asm volatile("ds_read_b128 %0, %1 offset:16384" : "=v"(wf[set][tt]) : "v"(a) : "memory"); break;
      }
    }
  };
  // one 8-value chunk of the previous slot's epilogue: accumulator set P, row tile rt, column half hh
  auto epi_chunk = [&](auto PC, int rt, int hh) {
    constexpr int P = decltype(PC)::value;
    if (EPI == EPI_NOSTORE) { asm volatile("" ::"v"(acc[P][rt][2 * hh]), "v"(acc[P][rt][2 * hh + 1])); return; }
    if (!prev_live) return;
    float v[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = acc[P][rt][2 * hh][e] + bias2[0][e]; v[4 + e] = acc[P][rt][2 * hh + 1][e] + bias2[1][e]; }
    const int row = prev_rowb + rt * 16;
    const long long o = (long long)row * g.N + prev_cb[hh];
    if (EPI == EPI_GELU_PRE) {
      if (row < g.M) gstore16(g.Cpre + o, pack8(v));
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = gelu_fast(v[e]);
    }
    if (EPI == EPI_RES) {
      typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
      const u32x4_t r = __builtin_bit_cast(u32x4_t, res4[RSETS == 2 ? P : 0][rt][hh]);
      const float sc = prev_sc[rt];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[2 * e] = __uint_as_float(r[e] << 16) + sc * v[2 * e];
        v[2 * e + 1] = __uint_as_float(r[e] & 0xffff0000u) + sc * v[2 * e + 1];
      }
    }
    if (row < g.M) gstore16(g.C + o, pack8(v));
  };
  auto load_bias = [&](int hh) {           // previous slot's bias of column half hh (LDS): older than the fragment reads whose counted wait precedes its use
    const unsigned bb = lds0 + RSL_BIAS_OFF + prev_cb[hh] * 4;
    asm volatile("ds_read_b128 %0, %1" : "=v"(bias2[0]) : "v"(bb) : "memory");
    asm volatile("ds_read_b128 %0, %1 offset:16" : "=v"(bias2[1]) : "v"(bb) : "memory");
  };

  rd(0, lds0, 0);                          // (steps 0 and 1 are guaranteed by the first barrier)
  auto slot_body = [&](auto PC, int sl) {
    constexpr int P = decltype(PC)::value;
    const long long sg = s_begin + sl;
    const int pnl = (int)(sg / g.nslots), s = (int)(sg % g.nslots);
    const bool last_of_panel = (s == g.nslots - 1) && (sl + 1 < nsl);      // wave-uniform: the next slot belongs to the next panel
    const int col0 = s * 64;
    int cbase[2];
    if (CM == 0) { cbase[0] = col0 + 16 * (lane >> 4); cbase[1] = cbase[0] + 8; }
    else { cbase[0] = col0 + 8 * (lane >> 4); cbase[1] = cbase[0] + 32; }
    const int rowb = pnl * RSL_ROWS + wave * 32 + (lane & 15);
    const bool live = pnl * RSL_ROWS + wave * 32 < g.M;      // wave-uniform: this wave's rows exist (last panel may be partial)
    constexpr int NLR = (EPI == EPI_RES) ? 4 : 0;      // residual loads of one slot
    constexpr int NSP = (EPI == EPI_GELU_PRE) ? 8 : 4; // stores of one slot
    if (EPI == EPI_RES && RSETS == 2) {     // one-step slots: this slot's residual, consumed by the epilogue inside the NEXT slot
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) res4[P][rt][hh] = gload16(g.res + (long long)min(rowb + rt * 16, g.M - 1) * g.N + cbase[hh]);
    }
    if (EPI == EPI_RES && RSETS == 1 && have_prev) {      // two-step slots: the PREVIOUS slot's residual, consumed one step later
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) res4[0][rt][hh] = gload16(g.res + (long long)min(prev_rowb + rt * 16, g.M - 1) * g.N + prev_cb[hh]);
    }
#pragma unroll
    for (int part = 0; part < SPS; ++part) {
      if (fresh) {                          // first use of re-requested panel registers: wait for the loads of THIS half (younger loads may fly)
        if (part == 0 && SPS == 2) wait_vm<2 * KST + NLR>(); else wait_vm<NLR>();
        touch_half(part);
        if (part == SPS - 1) { fresh = false; load_scale(pnl); }
      }
      const bool epi_here = part == SPS - 1 && have_prev;      // wave-uniform: the previous slot's epilogue rides in this step
      if (epi_here && EPI == EPI_RES) {
        // residual chunks of the previous slot: RSETS == 2: this slot's 4 loads are younger; RSETS == 1: the panel reload of part 0 may be
        if (RSETS == 2) wait_vm<NLR>(); else if (last_of_panel) wait_vm<2 * KST>(); else wait_vm<0>();
#pragma unroll
        for (int q = 0; q < 4; ++q) asm volatile("" : "+v"(res4[RSETS == 2 ? (P ^ 1) : 0][q >> 1][q & 1]));
      }
      const unsigned bbase = lds0 + buf * RSL_STEP, bnext = lds0 + ((buf + 1) & (RSL_R - 1)) * RSL_STEP;
#pragma unroll
      for (int ksl = 0; ksl < KST; ++ksl) {
        const int set = ksl & 1;
        if (!(ABL & 2)) { if (ksl + 1 < KST) rd(set ^ 1, bbase, ksl + 1); else rd(set ^ 1, bnext, 0); }      // last k-step: first fragments of the NEXT step
        wait_lgkm<4>();
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) asm volatile("" : "+v"(wf[set][tt]));
        if (part == 0 && ksl == 0) {
#pragma unroll
          for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) acc[P][rt][tt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
          for (int tt = 0; tt < 4; ++tt)
            acc[P][rt][tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[set][tt], af[rt][part * KST + ksl], acc[P][rt][tt], 0, 0, 0);
        if (part == SPS - 1 && ksl <= 4 && epi_here) {      // previous slot's epilogue, one chunk behind each of the MFMA groups 1..4
          if (ksl >= 1) {
            asm volatile("" : "+v"(bias2[0]), "+v"(bias2[1]));
            epi_chunk(std::integral_constant<int, P ^ 1>{}, (ksl - 1) >> 1, (ksl - 1) & 1);
          }
          if (ksl <= 3) load_bias(ksl & 1);             // bias of the chunk handled after the next group (older than that group's counted wait)
        }
      }
      if (last_of_panel) {                  // this half of the panel registers is dead: request the next panel's rows into it
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
          for (int ks = 0; ks < KST; ++ks) asm volatile("" : "+v"(acc[P][rt][0]), "+v"(af[rt][part * KST + ks]));      // (after the MFMAs that read it)
        load_half(pnl + 1, part);
        if (part == SPS - 1) fresh = true;
      }
      if (part == SPS - 1) {                // hand this slot's epilogue to the next slot's first step
        have_prev = true; prev_live = live; prev_cb[0] = cbase[0]; prev_cb[1] = cbase[1]; prev_rowb = rowb; prev_sc[0] = rsc[0]; prev_sc[1] = rsc[1];
      }
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      buf = (buf + 1) & (RSL_R - 1);
    }
  };
  for (int sl = 0; sl < nsl; sl += 2) {
    slot_body(std::integral_constant<int, 0>{}, sl);
    if (sl + 1 < nsl) slot_body(std::integral_constant<int, 1>{}, sl + 1);
  }
  // the last slot's epilogue
  wait_lgkm<0>();
  if (EPI == EPI_RES && RSETS == 1) {
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) res4[0][rt][hh] = gload16(g.res + (long long)min(prev_rowb + rt * 16, g.M - 1) * g.N + prev_cb[hh]);
  }
  if (EPI == EPI_RES) wait_vm<0>();
  auto tail = [&](auto PC) {
    constexpr int P = decltype(PC)::value;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      load_bias(q & 1);
      wait_lgkm<0>();
      asm volatile("" : "+v"(bias2[0]), "+v"(bias2[1]), "+v"(res4[RSETS == 2 ? P : 0][q >> 1][q & 1]));
      epi_chunk(PC, q >> 1, q & 1);
    }
  };
  if (nsl & 1) tail(std::integral_constant<int, 0>{}); else tail(std::integral_constant<int, 1>{});
}

// ---- reference: fp32 accumulation of the bf16 operands, one thread per output ---------------------------------------------------
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }
__global__ void ref_kernel(const bf16_t* A, const bf16_t* W, const float* bias, const bf16_t* res, const float* row_scale, int rps, float* out, float* out_pre, int M, int N, int K, int epi) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)M * N) return;
  const int r = (int)(idx / N), n = (int)(idx % N);
  float s = 0.f;
  for (int k = 0; k < K; ++k) s += bf2f(A[(long long)r * K + k]) * bf2f(W[(long long)n * K + k]);
  s += bias[n];
  if (epi == EPI_GELU_PRE) { out_pre[idx] = s; s = 0.5f * s * (1.0f + erff(s * 0.70710678118654752f)); }
  if (epi == EPI_RES) s = bf2f(res[idx]) + (row_scale ? row_scale[r / rps] : 1.f) * s;
  out[idx] = s;
}
__global__ void cmp_kernel(const bf16_t* got, const float* ref, long long n, float* maxerr, float* maxref) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  const float e = fabsf(bf2f(got[idx]) - ref[idx]);
  atomicMax(reinterpret_cast<int*>(maxerr), __float_as_int(e));
  atomicMax(reinterpret_cast<int*>(maxref), __float_as_int(fabsf(ref[idx])));
}

static bf16_t h_f2bf(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (bf16_t)(u >> 16); }

template <int KS, int EPI, int CM, int ABL = 0>
static float run_rs(const RSArgs& a, int iters, hipStream_t st) {
  auto kern = rs_gemm_kernel<KS, EPI, CM, ABL>;
  const int lds = 3 * 64 * 6 * 64;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(a.nwg), dim3(256), lds, st, a);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, st));
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, dim3(a.nwg), dim3(256), lds, st, a);
  CK(hipEventRecord(e1, st));
  CK(hipEventSynchronize(e1));
  CK(hipGetLastError());
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1000.f / iters;
}

template <int KS, int EPI, int CM, int ABL = 0>
static float run_rsl(const RSLArgs& a, int iters, hipStream_t st) {
  auto kern = rsl_kernel<KS, EPI, CM, ABL>;
  const int lds = RSL_BIAS_OFF + 32768;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(a.nwg), dim3(512), lds, st, a);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, st));
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, dim3(a.nwg), dim3(512), lds, st, a);
  CK(hipEventRecord(e1, st));
  CK(hipEventSynchronize(e1));
  CK(hipGetLastError());
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1000.f / iters;
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 27136, N = argc > 2 ? atoi(argv[2]) : 1536, K = argc > 3 ? atoi(argv[3]) : 384;
  if (K != 384 && K != 192) { printf("K must be 192 or 384\n"); return 0; }
  if (N % 64) { printf("N must be a multiple of 64\n"); return 0; }
  printf("rs probe: M=%d N=%d K=%d bf16\n", M, N, K);
  std::vector<bf16_t> hA((size_t)M * K), hW((size_t)N * K), hR((size_t)M * N);
  std::vector<float> hb(N), hs((M + 211) / 212);
  unsigned long long st_ = 0x9E3779B97F4A7C15ull;
  auto rnd = [&]() { st_ = st_ * 6364136223846793005ull + 1442695040888963407ull; return (float)((st_ >> 33) & 0xffffff) / 16777216.f * 2.f - 1.f; };
  for (auto& v : hA) v = h_f2bf(rnd());
  for (auto& v : hW) v = h_f2bf(rnd() * 0.08f);
  for (auto& v : hR) v = h_f2bf(rnd());
  for (auto& v : hb) v = rnd() * 0.5f;
  for (auto& v : hs) v = rnd() > 0 ? 1.1f : 0.f;
  bf16_t *dA, *dW, *dC, *dP, *dR; float *db, *dsc, *dref, *dref2, *derr;
  CK(hipMalloc(&dA, hA.size() * 2)); CK(hipMalloc(&dW, hW.size() * 2)); CK(hipMalloc(&dC, (size_t)M * N * 2)); CK(hipMalloc(&dP, (size_t)M * N * 2)); CK(hipMalloc(&dR, (size_t)M * N * 2));
  CK(hipMalloc(&db, N * 4)); CK(hipMalloc(&dsc, hs.size() * 4)); CK(hipMalloc(&dref, (size_t)M * N * 4)); CK(hipMalloc(&dref2, (size_t)M * N * 4)); CK(hipMalloc(&derr, 8));
  CK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dW, hW.data(), hW.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(dR, hR.data(), hR.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(db, hb.data(), N * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dsc, hs.data(), hs.size() * 4, hipMemcpyHostToDevice));
  hipStream_t st; CK(hipStreamCreate(&st));
  unsigned long long* ddbg; CK(hipMalloc(&ddbg, 1024 * 16 * 8)); CK(hipMemset(ddbg, 0, 1024 * 16 * 8));

  RSArgs a{};
  a.A = dA; a.W = dW; a.bias = db; a.C = dC; a.Cpre = dP; a.res = dR; a.row_scale = dsc; a.M = M; a.N = N; a.K = K; a.rps = 212;
  a.npanels = (M + 127) / 128; a.nslots = N / 64; a.total = (long long)a.npanels * a.nslots;
  const double fl = 2.0 * M * N * K;
  RSLArgs b{};
  b.A = dA; b.W = dW; b.bias = db; b.C = dC; b.Cpre = dP; b.res = dR; b.row_scale = dsc; b.M = M; b.N = N; b.K = K; b.rps = 212;
  b.npanels = (M + RSL_ROWS - 1) / RSL_ROWS; b.nslots = N / 64; b.total = (long long)b.npanels * b.nslots;
  auto check = [&](int epi, const char* tag, float us) {
    hipLaunchKernelGGL(ref_kernel, dim3((unsigned)(((long long)M * N + 255) / 256)), dim3(256), 0, st, dA, dW, db, dR, dsc, 212, dref, dref2, M, N, K, epi);
    float z[2] = {0.f, 0.f};
    CK(hipMemcpyAsync(derr, z, 8, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(cmp_kernel, dim3((unsigned)(((long long)M * N + 255) / 256)), dim3(256), 0, st, dC, dref, (long long)M * N, derr, derr + 1);
    float r[2]; CK(hipMemcpyAsync(r, derr, 8, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
    float rp = 0.f;
    if (epi == EPI_GELU_PRE) {
      CK(hipMemcpyAsync(derr, z, 8, hipMemcpyHostToDevice, st));
      hipLaunchKernelGGL(cmp_kernel, dim3((unsigned)(((long long)M * N + 255) / 256)), dim3(256), 0, st, dP, dref2, (long long)M * N, derr, derr + 1);
      float r2[2]; CK(hipMemcpyAsync(r2, derr, 8, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
      rp = r2[0] / r2[1];
    }
    printf("%-34s : %7.1f us  %7.1f TFLOP/s   rel err %.2e %s%s\n", tag, us, fl / us / 1e6, r[0] / r[1], (r[0] / r[1] < 6e-3 && rp < 6e-3) ? "ok" : "WRONG", epi == EPI_GELU_PRE ? " (+pre)" : "");
  };
#define RUN(KS_, EPI_, CM_, NWG_, TAG_) do { a.nwg = (int)std::min<long long>(NWG_, a.total); CK(hipMemsetAsync(dC, 0, (size_t)M * N * 2, st)); CK(hipMemsetAsync(dP, 0, (size_t)M * N * 2, st)); \
    const float us = run_rs<KS_, EPI_, CM_>(a, 20, st); char tag[96]; snprintf(tag, sizeof tag, "%s cm=%d nwg=%d", TAG_, CM_, a.nwg); \
    if (EPI_ == EPI_NOSTORE) printf("%-34s : %7.1f us  %7.1f TFLOP/s\n", tag, us, fl / us / 1e6); else check(EPI_, tag, us); } while (0)
#define RUNL(KS_, EPI_, CM_, ABL_, NWG_, TAG_) do { b.nwg = (int)std::min<long long>(NWG_, b.total); CK(hipMemsetAsync(dC, 0, (size_t)M * N * 2, st)); CK(hipMemsetAsync(dP, 0, (size_t)M * N * 2, st)); \
    const float us = run_rsl<KS_, EPI_, CM_, ABL_>(b, 20, st); char tag[96]; snprintf(tag, sizeof tag, "%s cm=%d nwg=%d", TAG_, CM_, b.nwg); \
    if (EPI_ == EPI_NOSTORE || ABL_) printf("%-34s : %7.1f us  %7.1f TFLOP/s\n", tag, us, fl / us / 1e6); else check(EPI_, tag, us); } while (0)
#define TLL(EPI_, TAG_) do { b.nwg = 256; b.dbg = ddbg; CK(hipMemset(ddbg, 0, 256 * 128)); run_rsl<12, EPI_, 1, 0>(b, 1, st); b.dbg = nullptr; \
      std::vector<unsigned long long> hd(256 * 16); CK(hipMemcpy(hd.data(), ddbg, 256 * 128, hipMemcpyDeviceToHost)); double a[13] = {0}; \
      for (int w = 0; w < 256; ++w) for (int i = 0; i < 13; ++i) a[i] += (double)hd[w * 16 + i] / 256; \
      printf("RSL timeline %-10s: compute wave 0: A-load %.0f  steps %.0f  barriers %.0f  epilogues %.0f  life %.0f | loader: issue %.0f  vmcnt %.0f  barriers %.0f  life %.0f  (T = %.1f steps; ticks)\n", \
             TAG_, a[0], a[1], a[2], a[3], a[4], a[8], a[9], a[10], a[11], a[12]); } while (0)
  if (K == 384 && N <= 8192) {
    RUNL(12, EPI_BIAS, 1, 0, 256, "RSL bias");
    RUNL(12, EPI_BIAS, 0, 0, 256, "RSL bias");
    RUNL(12, EPI_NOSTORE, 1, 0, 256, "RSL nostore");
    RUNL(12, EPI_NOSTORE, 1, 2, 256, "RSL nostore no ds_read");
    RUNL(12, EPI_GELU_PRE, 1, 0, 256, "RSL gelu+pre");
    RUNL(12, EPI_RES, 1, 0, 256, "RSL res+scale");
  }
  if (K == 192 && N <= 2048) {
    RUNL(6, EPI_BIAS, 1, 0, 256, "RSL bias");
    RUNL(6, EPI_NOSTORE, 1, 0, 256, "RSL nostore");
    RUNL(6, EPI_GELU_PRE, 1, 0, 256, "RSL gelu+pre");
    RUNL(6, EPI_RES, 1, 0, 256, "RSL res+scale");
  }
  if (K == 384) {
    RUN(12, EPI_BIAS, 0, 512, "rs bias");
    RUN(12, EPI_BIAS, 1, 512, "rs bias");
    RUN(12, EPI_BIAS, 1, 256, "rs bias");
    RUN(12, EPI_BIAS, 1, 1024, "rs bias");
    RUN(12, EPI_NOSTORE, 1, 512, "rs nostore");
#define RUNA(ABL_, NWG_, TAG_) do { a.nwg = NWG_; const float us = run_rs<12, EPI_NOSTORE, 1, ABL_>(a, 20, st); printf("%-34s nwg=%d : %7.1f us  %7.1f TFLOP/s\n", TAG_, a.nwg, us, fl / us / 1e6); } while (0)
#define RUNB(EPI_, ABL_, NWG_, TAG_) do { a.nwg = NWG_; const float us = run_rs<12, EPI_, 1, ABL_>(a, 20, st); printf("%-34s nwg=%d : %7.1f us  %7.1f TFLOP/s\n", TAG_, a.nwg, us, fl / us / 1e6); } while (0)
#define TIMELINE(EPI_, ABL_, NWG_, TAG_) do { a.nwg = NWG_; a.dbg = ddbg; CK(hipMemset(ddbg, 0, 1024 * 64)); run_rs<12, EPI_, 1, ABL_>(a, 1, st); a.dbg = nullptr; \
      std::vector<unsigned long long> hd(1024 * 8); CK(hipMemcpy(hd.data(), ddbg, 1024 * 64, hipMemcpyDeviceToHost)); \
      double sA = 0, sB = 0, sSw = 0, sTot = 0; int nsw = 0; unsigned long long t0min = ~0ull, t5max = 0, t0max = 0; \
      for (int w = 0; w < NWG_; ++w) { const unsigned long long* d = &hd[w * 8]; sA += d[1] - d[0]; sB += d[2] - d[1]; sTot += d[5] - d[0]; if (d[3]) { sSw += d[4] - d[3]; ++nsw; } \
        if (d[0] < t0min) t0min = d[0]; if (d[0] > t0max) t0max = d[0]; if (d[5] > t5max) t5max = d[5]; } \
      printf("timeline %-22s nwg=%d: A-load %.0f  barrier %.0f  switch %.0f (x%d)  wg-life %.0f  | launch spread %.0f  kernel span %.0f (ticks)\n", TAG_, NWG_, sA / NWG_, sB / NWG_, nsw ? sSw / nsw : 0., nsw, sTot / NWG_, (double)(t0max - t0min), (double)(t5max - t0min)); } while (0)
    TIMELINE(EPI_BIAS, 0, 512, "bias");
    TIMELINE(EPI_NOSTORE, 7, 512, "mfma only");
    TIMELINE(EPI_BIAS, 0, 256, "bias");
    RUNB(EPI_BIAS, 64 + 16 + 15, 512, "f&f stores only, 8 rows x 128 B");
    RUNB(EPI_BIAS, 64 + 16 + 7, 512, "MFMA + f&f stores, 8 rows x 128 B");
    RUNB(EPI_BIAS, 32 + 16 + 7, 512, "MFMA + f&f stores to L2 (1.5MB)");
    RUNB(EPI_BIAS, 32 + 16 + 15, 512, "f&f stores only, to L2");
    RUNB(EPI_BIAS, 16 + 7, 512, "MFMA + fire&forget stores");
    RUNB(EPI_BIAS, 16 + 7, 256, "MFMA + fire&forget stores");
    RUNB(EPI_GELU_PRE, 16 + 7, 512, "MFMA + f&f stores gelu+pre");
    RUNB(EPI_BIAS, 16 + 15, 512, "f&f stores only");
    RUNB(EPI_BIAS, 1, 512, "bias store, no DMA");
    RUNB(EPI_BIAS, 3, 512, "bias store, no DMA no ds_read");
    RUNB(EPI_BIAS, 15, 512, "bias store only (no MFMA)");
    RUNB(EPI_BIAS, 15, 256, "bias store only (no MFMA)");
    RUNB(EPI_GELU_PRE, 15, 512, "gelu+pre store only (no MFMA)");
    RUNB(EPI_BIAS, 8, 512, "bias store, DMA+reads, no MFMA");
    RUNA(1, 512, "nostore, no DMA");
    RUNA(2, 512, "nostore, no ds_read");
    RUNA(3, 512, "nostore, no DMA no ds_read");
    RUNA(7, 512, "nostore, MFMA only (no barrier)");
    RUNA(4, 512, "nostore, no barrier (racy)");
    RUNA(1, 256, "nostore, no DMA");
    RUNA(3, 256, "nostore, no DMA no ds_read");
    RUNA(7, 256, "nostore, MFMA only (no barrier)");
    RUN(12, EPI_GELU_PRE, 0, 512, "rs gelu+pre");
    RUN(12, EPI_GELU_PRE, 1, 512, "rs gelu+pre");
    RUN(12, EPI_RES, 0, 512, "rs res+scale");
    RUN(12, EPI_RES, 1, 512, "rs res+scale");
  } else {
    RUN(6, EPI_BIAS, 0, 512, "rs bias");
    RUN(6, EPI_BIAS, 1, 512, "rs bias");
    RUN(6, EPI_BIAS, 1, 1024, "rs bias");
    RUN(6, EPI_NOSTORE, 1, 512, "rs nostore");
    RUN(6, EPI_GELU_PRE, 1, 512, "rs gelu+pre");
    RUN(6, EPI_RES, 1, 512, "rs res+scale");
  }
#ifdef WITH_LMV
  {
    auto time_lmv = [&](int act, bool pre, bool res, const char* tag) {
      lmv_linear_problem p{};
      p.a = dA; p.w = dW; p.out = dC; p.rows = M; p.bias = db;
      if (pre) p.out_pre = dP;
      if (res) { p.res = dR; p.row_scale = dsc; p.rows_per_sample = 212; }
      for (int i = 0; i < 3; ++i) if (lmv_linear_fwd(&p, 1, N, K, act, LMV_BF16, st)) { printf("lmv error: %s\n", lmv_last_error()); return; }
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < 20; ++i) lmv_linear_fwd(&p, 1, N, K, act, LMV_BF16, st);
      CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      const float us = ms * 1000.f / 20;
      printf("%-34s : %7.1f us  %7.1f TFLOP/s\n", tag, us, fl / us / 1e6);
    };
    time_lmv(LMV_ACT_NONE, false, false, "lmv_linear_fwd bias");
    time_lmv(LMV_ACT_GELU, true, false, "lmv_linear_fwd gelu+pre");
    time_lmv(LMV_ACT_NONE, false, true, "lmv_linear_fwd res+scale");
  }
#endif
  return 0;
}
