// misc.hip -- error plumbing and the small memory-bound utilities (cast, DropPath row scale,
// fused flat AdamW).  All are grid-stride, 16-byte-per-lane kernels bounded by HBM bandwidth.
#include <stdarg.h>
#include <string.h>
#include <mutex>
#include <vector>
#include "common.h"

static thread_local char g_err[512] = "";

void lmv_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* lmv_last_error(void) { return g_err; }

// ---- error word of the persistent stage kernels ------------------------------------------------------------------------------------------
// One sticky word per device in PINNED, device-mapped host memory: a kernel whose bounded in-launch wait runs out stores 1 there (system scope), and the host reads
// it with a plain load -- no device synchronisation, no copy -- so every forward call can afford to look at it (lemevit_amd/model.py raises at the next call, bench.py and
// graph.try_graphed at their synchronisation points).  Allocated on the first stage call of a device under a mutex; never inside a stream capture (the first call of a
// process is an eager warm-up; a capture that would be the first call is refused).
namespace {
struct ErrWord { unsigned* host = nullptr; unsigned* dev = nullptr; };
ErrWord g_errword[64];
std::mutex g_errword_mu;
}
unsigned* lmv_stage_errword(void* stream) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lock(g_errword_mu);
  ErrWord& w = g_errword[dev & 63];
  if (!w.dev) {
    if (stream) {          // hipHostMalloc is not a capturable call
      hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
      if (hipStreamIsCapturing((hipStream_t)stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) return nullptr;
    }
    void* h = nullptr; void* d = nullptr;
    if (hipHostMalloc(&h, 256, hipHostMallocMapped) != hipSuccess) return nullptr;
    memset(h, 0, 256);
    if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess) { (void)hipHostFree(h); return nullptr; }
    w.host = (unsigned*)h; w.dev = (unsigned*)d;
  }
  return w.dev;
}
extern "C" int lmv_stage_error_count(int reset) {
  if (!lmv_stage_errword(nullptr)) LMV_FAIL(LMV_ERR_LAUNCH, "stage_error_count: cannot allocate the error word");
  int dev = 0;
  (void)hipGetDevice(&dev);
  volatile unsigned* h = g_errword[dev & 63].host;
  const unsigned v = *h;
  if (reset && v) *h = 0;
  return (int)(v & 0x7fffffffu);
}
// test hook: what a kernel's exhausted wait does, from the host (tests/test_model_gpu.py::test_lost_handoff_is_loud)
extern "C" int lmv_debug_stage_error_set(int value) {
  if (!lmv_stage_errword(nullptr)) LMV_FAIL(LMV_ERR_LAUNCH, "debug_stage_error_set: cannot allocate the error word");
  int dev = 0;
  (void)hipGetDevice(&dev);
  *(volatile unsigned*)g_errword[dev & 63].host = (unsigned)value;
  return LMV_OK;
}
// Workgroups of `kernel` (threads, dynamic LDS bytes) the device holds at once: the occupancy query x the CU count.  The stage kernels size their slot counts with it; their
// slot assignment (stage_ticket, stage_common.h) is correct for ANY residency, so an optimistic answer (MI355X_MICROARCH.md: the query can be one block per CU high) costs time, not results.
int lmv_stage_capacity(const void* kernel, int threads, size_t lds_bytes) {
  int dev = 0, cus = 0, per_cu = 0;
  (void)hipGetDevice(&dev);
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) return 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, lds_bytes) != hipSuccess || per_cu <= 0) return 0;
  return cus * per_cu;
}

// ---- launch timing probe ---------------------------------------------------------------------------------------------------------------
bool g_lmv_timing_on = false;
namespace {
struct TimingRec { hipEvent_t e0, e1; double flops, bytes; int kind; };
std::vector<TimingRec> g_timing;          // (single host thread: the probe is a measurement aid, enabled for a few steps by bench.py)
size_t g_timing_n = 0;
}
void lmv_timing_begin(void* stream, double flops, double bytes, int kind) {
  if (g_timing_n >= g_timing.size()) return;
  TimingRec& r = g_timing[g_timing_n];
  r.flops = flops; r.bytes = bytes; r.kind = kind;
  (void)hipEventRecord(r.e0, (hipStream_t)stream);
}
void lmv_timing_end(void* stream) {
  if (g_timing_n >= g_timing.size()) return;
  (void)hipEventRecord(g_timing[g_timing_n++].e1, (hipStream_t)stream);
}
void lmv_timing_set_kind(int kind) {          // the open record (between begin and end): the entry point learns which kernel takes the launch after it has opened the bracket
  if (g_timing_n < g_timing.size()) g_timing[g_timing_n].kind = kind;
}
extern "C" int lmv_debug_launch_timing(int capacity) {
  g_lmv_timing_on = false;
  for (auto& r : g_timing) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
  g_timing.clear(); g_timing_n = 0;
  if (capacity <= 0) return LMV_OK;
  g_timing.resize((size_t)capacity);
  for (auto& r : g_timing)
    if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) LMV_FAIL(LMV_ERR_LAUNCH, "launch timing: cannot create events");
  g_lmv_timing_on = true;
  return LMV_OK;
}
extern "C" int lmv_debug_launch_timing_read(float* ms, double* flops, double* bytes, int* kinds, int capacity) {
  g_lmv_timing_on = false;          // (the caller has synchronised the device)
  int n = 0;
  for (size_t i = 0; i < g_timing_n && n < capacity; ++i) {
    float t = 0.f;
    if (hipEventElapsedTime(&t, g_timing[i].e0, g_timing[i].e1) != hipSuccess) continue;
    ms[n] = t; flops[n] = g_timing[i].flops; bytes[n] = g_timing[i].bytes; kinds[n] = g_timing[i].kind; ++n;
  }
  return n;
}

extern "C" int lmv_abi_version(void) { return LMV_ABI_VERSION; }

// ---- A/B switches: lmv_config_set(key, value) at run time (tests, tools); five of them also from the environment, read once, here (round 6: the other ~25 LMV_* variables are gone) ------------------------------------------------------------------------------------
#include <stdlib.h>
#include <string.h>
namespace {
int env_int(const char* name, int dflt) { const char* e = name ? getenv(name) : nullptr; return e && *e ? atoi(e) : dflt; }
struct ConfigKey { const char* key; const char* env; int LmvConfig::*field; int dflt; };
const ConfigKey kConfigKeys[] = {
    {"gemm_bk", nullptr, &LmvConfig::gemm_bk, 0}, {"gemm_bk32_tiles", nullptr, &LmvConfig::gemm_bk32_tiles, 512},
    {"dw_bk", nullptr, &LmvConfig::dw_bk, 32}, {"dw_target_blocks", "LMV_DW_TARGET_BLOCKS", &LmvConfig::dw_target_blocks, 0},
    {"gemm_no_dma", nullptr, &LmvConfig::gemm_no_dma, 0}, {"gemm_w8", "LMV_GEMM_W8", &LmvConfig::gemm_w8, 1},
    {"gemm_cumap", nullptr, &LmvConfig::gemm_cumap, 1}, {"gemm_nst", nullptr, &LmvConfig::gemm_nst, 2},
    {"gemm_nst_dw", nullptr, &LmvConfig::gemm_nst_dw, 3}, {"gemm_rs", "LMV_GEMM_RS", &LmvConfig::gemm_rs, 1}, {"gemm_wn", "LMV_GEMM_WN", &LmvConfig::gemm_wn, 1},
    {"dwconv_v", nullptr, &LmvConfig::dwconv_v, 0},
    {"stage_ticket_skew", "LMV_STAGE_TICKET_SKEW", &LmvConfig::stage_ticket_skew, 0}, {"mlp_tm", nullptr, &LmvConfig::mlp_tm, 0}, {"mlp_rw96", nullptr, &LmvConfig::mlp_rw96, 1}, {"mlp_split384", nullptr, &LmvConfig::mlp_split384, 1}, {"dx_ln_fused", nullptr, &LmvConfig::dx_ln_fused, 1}, {"res_ln_fused", nullptr, &LmvConfig::res_ln_fused, 1}, {"ln_exact_fused", nullptr, &LmvConfig::ln_exact_fused, 1}, {"attn_pv16", nullptr, &LmvConfig::attn_pv16, 1},
    {"attn_fuse_dq", nullptr, &LmvConfig::attn_fuse_dq, 1}, {"attn_fused_bwd", nullptr, &LmvConfig::attn_fused_bwd, 2},
    {"attn_pair", nullptr, &LmvConfig::attn_pair, 1}, {"ln_bwd_blocks", nullptr, &LmvConfig::ln_bwd_blocks, 1024},
    {"ln_bwd_minrows", nullptr, &LmvConfig::ln_bwd_minrows, 2},
};
LmvConfig config_from_env() {
  LmvConfig c{};
  for (const ConfigKey& k : kConfigKeys) c.*(k.field) = env_int(k.env, k.dflt);
  if (c.ln_bwd_blocks <= 0 || c.ln_bwd_blocks > 2048) c.ln_bwd_blocks = 512;
  if (c.ln_bwd_minrows <= 0) c.ln_bwd_minrows = 2;
  return c;
}
LmvConfig g_config = config_from_env();      // static initialisation = library load
}  // namespace
LmvConfig& lmv_config() { return g_config; }
extern "C" int lmv_config_set(const char* key, int value) {
  if (!key) LMV_FAIL(LMV_ERR_SHAPE, "config_set: null key");
  for (const ConfigKey& k : kConfigKeys)
    if (!strcmp(k.key, key)) { g_config.*(k.field) = value; return LMV_OK; }
  LMV_FAIL(LMV_ERR_SHAPE, "config_set: unknown key '%s'", key);
}
extern "C" int lmv_config_get(const char* key, int* value) {
  if (!key || !value) LMV_FAIL(LMV_ERR_SHAPE, "config_get: null argument");
  for (const ConfigKey& k : kConfigKeys)
    if (!strcmp(k.key, key)) { *value = g_config.*(k.field); return LMV_OK; }
  LMV_FAIL(LMV_ERR_SHAPE, "config_get: unknown key '%s'", key);
}

namespace {

constexpr int TPB = 256;
inline int grid_for(int64_t work_items) {
  int64_t b = (work_items + TPB - 1) / TPB;
  if (b > 2048) b = 2048;
  if (b < 1) b = 1;
  return (int)b;
}

template <typename S, typename D>
__global__ __launch_bounds__(TPB) void cast_kernel(const S* __restrict__ src, D* __restrict__ dst, int64_t n) {
  const int64_t n4 = n >> 2;
  for (int64_t i = blockIdx.x * (int64_t)TPB + threadIdx.x; i < n4; i += (int64_t)gridDim.x * TPB) {
    float f[4];
    ld4(src + i * 4, f);
    st4(dst + i * 4, f);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const int64_t i = (n4 << 2) + threadIdx.x;
    DT<D>::st(dst + i, DT<S>::ld(src + i));
  }
}

// y[r, :] = x[r, :] * scale[r / rps] for up to TWO row segments per launch (the image-token and the meta-token gradient of a
// block are scaled by their own DropPath vectors at the same point of the backward pass); 16-byte chunks, 32-bit index math.
struct RowScaleSeg { const void* x; const float* scale; void* y; unsigned chunks, cpr, rps; };      // chunks = rows * cpr, cpr = C / EPC
struct RowScaleArgs { RowScaleSeg s[2]; unsigned total; };

template <typename T>
__global__ __launch_bounds__(TPB) void row_scale_kernel(const RowScaleArgs a) {
  constexpr int EPC = DT<T>::EPC;
  for (unsigned i = blockIdx.x * TPB + threadIdx.x; i < a.total; i += gridDim.x * TPB) {
    const bool second = i >= a.s[0].chunks;
    const RowScaleSeg& g = a.s[second ? 1 : 0];
    const unsigned j = second ? i - a.s[0].chunks : i;
    const float sc = g.scale[(j / g.cpr) / g.rps];
    float f[EPC];
    chunk_to_f<T>(reinterpret_cast<const uint4*>(g.x)[j], f);
#pragma unroll
    for (int e = 0; e < EPC; ++e) f[e] *= sc;
    reinterpret_cast<uint4*>(g.y)[j] = f_to_chunk<T>(f);
  }
}

// ---- classifier tail (models/lemevit.py:815-835): pooled[b, :] = mean_l x[b, l, :] + mean_m c[b, m, :] ----------------------------
// One thread owns 8 (bf16) / 4 (fp32) channels of one sample and walks the tokens of both segments: the sums stay in registers,
// the result is ONE fp32 row per sample (the operand of the head GEMM).  Backward is the broadcast dx[b, l, :] = g[b, :] / L.
template <typename T>
__global__ __launch_bounds__(TPB) void token_mean2_fwd_kernel(const T* __restrict__ x, int L, const T* __restrict__ c, int M, int C, int B,
                                                            T* __restrict__ out, const float* __restrict__ xs = nullptr, const float* __restrict__ xb = nullptr) {
  constexpr int EPC = DT<T>::EPC;
  const int cpr = C / EPC;
  const unsigned i = blockIdx.x * TPB + threadIdx.x;
  if (i >= (unsigned)B * cpr) return;
  const int b = i / cpr, ch = i % cpr;
  float acc[EPC], f[EPC];
#pragma unroll
  for (int e = 0; e < EPC; ++e) acc[e] = 0.f;
  const uint4* px = reinterpret_cast<const uint4*>(x) + (size_t)b * L * cpr + ch;
  // (8 token rows requested before the first is consumed: the one-load-per-iteration loop was a chain of exposed memory round trips -- 21 us for the 6 MB of a stage-4 map;
  //  the sums are still taken in token order)
  int l = 0;
  for (; l + 8 <= L; l += 8) {
    uint4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = px[(size_t)(l + j) * cpr];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      chunk_to_f<T>(v[j], f);
#pragma unroll
      for (int e = 0; e < EPC; ++e) acc[e] += f[e];
    }
  }
  for (; l < L; ++l) {
    chunk_to_f<T>(px[(size_t)l * cpr], f);
#pragma unroll
    for (int e = 0; e < EPC; ++e) acc[e] += f[e];
  }
  const float il = 1.f / (float)L;
#pragma unroll
  for (int e = 0; e < EPC; ++e) acc[e] *= il;
  if (xs) {      // eval-mode BatchNorm is affine per channel, so it commutes with the spatial mean: mean(BN(x)) = xs * mean(x) + xb
#pragma unroll
    for (int e = 0; e < EPC; ++e) acc[e] = fmaf(acc[e], xs[ch * EPC + e], xb[ch * EPC + e]);
  }
  if (c) {
    float a2[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) a2[e] = 0.f;
    const uint4* pc = reinterpret_cast<const uint4*>(c) + (size_t)b * M * cpr + ch;
    int m = 0;
    for (; m + 8 <= M; m += 8) {
      uint4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = pc[(size_t)(m + j) * cpr];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        chunk_to_f<T>(v[j], f);
#pragma unroll
        for (int e = 0; e < EPC; ++e) a2[e] += f[e];
      }
    }
    for (; m < M; ++m) {
      chunk_to_f<T>(pc[(size_t)m * cpr], f);
#pragma unroll
      for (int e = 0; e < EPC; ++e) a2[e] += f[e];
    }
    const float im = 1.f / (float)M;
#pragma unroll
    for (int e = 0; e < EPC; ++e) acc[e] += a2[e] * im;
  }
  reinterpret_cast<uint4*>(out)[i] = f_to_chunk<T>(acc);
}

template <typename T>
__global__ __launch_bounds__(TPB) void token_mean2_bwd_kernel(const T* __restrict__ g, T* __restrict__ dx, int L, T* __restrict__ dc, int M, int C, int B) {
  constexpr int EPC = DT<T>::EPC;
  const int cpr = C / EPC;
  const unsigned tot = (unsigned)B * (L + M) * cpr;
  for (unsigned i = blockIdx.x * TPB + threadIdx.x; i < tot; i += gridDim.x * TPB) {
    const unsigned row = i / cpr, ch = i % cpr;
    const unsigned b = row / (L + M), t = row % (L + M);
    float f[EPC];
    chunk_to_f<T>(reinterpret_cast<const uint4*>(g)[(size_t)b * cpr + ch], f);
    const float sc = t < (unsigned)L ? 1.f / (float)L : 1.f / (float)M;
#pragma unroll
    for (int e = 0; e < EPC; ++e) f[e] *= sc;
    if (t < (unsigned)L) reinterpret_cast<uint4*>(dx)[((size_t)b * L + t) * cpr + ch] = f_to_chunk<T>(f);
    else if (dc) reinterpret_cast<uint4*>(dc)[((size_t)b * M + (t - L)) * cpr + ch] = f_to_chunk<T>(f);
  }
}

// ---- dense 3x3 / stride-2 / pad-1 convolution on channels-last maps as im2col + the block GEMM (SURVEY section 8, row f1;
// models/lemevit.py:701-703 second stem convolution, :714-717 stage transitions) -------------------------------------------------
// patches[(b, ho, wo)][(ky * 3 + kx) * C + ci] = x[b, 2 ho - 1 + ky, 2 wo - 1 + kx, ci] (zero outside the map; columns 9 C .. KP - 1
// zero: KP pads the reduction to whole 64-deep k-steps so that the GEMM takes its LDS-DMA path).  One thread per 16-byte chunk:
// C % 8 == 0, so a chunk never straddles two taps.
template <typename T>
__global__ __launch_bounds__(TPB) void im2col_nhwc_kernel(const T* __restrict__ x, T* __restrict__ out, int B, int H, int W, int C, int Ho, int Wo, int KP) {
  constexpr int EPC = DT<T>::EPC;
  const int cpr = KP / EPC, cpt = C / EPC;                 // chunks per patch row / per tap
  const int64_t total = (int64_t)B * Ho * Wo * cpr;
  for (int64_t i = blockIdx.x * (int64_t)TPB + threadIdx.x; i < total; i += (int64_t)gridDim.x * TPB) {
    const int j = (int)(i % cpr);
    const int64_t r = i / cpr;
    const int wo = (int)(r % Wo), ho = (int)((r / Wo) % Ho), b = (int)(r / ((int64_t)Wo * Ho));
    const int tap = j / cpt, cc = j - tap * cpt, ky = tap / 3, kx = tap - ky * 3;
    const int h = 2 * ho - 1 + ky, w = 2 * wo - 1 + kx;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (tap < 9 && h >= 0 && h < H && w >= 0 && w < W) v = reinterpret_cast<const uint4*>(x)[(((int64_t)b * H + h) * W + w) * cpt + cc];
    reinterpret_cast<uint4*>(out)[i] = v;
  }
}

// data gradient: dx[b, h, w, ci] = sum over the taps (ky, kx) whose output pixel ((h + 1 - ky) / 2, (w + 1 - kx) / 2) exists of
// dpatches[that pixel][(ky * 3 + kx) * C + ci] -- a GATHER (1, 2 or 4 terms per input pixel, fixed order: no atomics)
template <typename T>
__global__ __launch_bounds__(TPB) void col2im_nhwc_kernel(const T* __restrict__ dp, T* __restrict__ dx, int B, int H, int W, int C, int Ho, int Wo, int KP) {
  constexpr int EPC = DT<T>::EPC;
  const int cpr = KP / EPC, cpt = C / EPC;
  const int64_t total = (int64_t)B * H * W * cpt;
  for (int64_t i = blockIdx.x * (int64_t)TPB + threadIdx.x; i < total; i += (int64_t)gridDim.x * TPB) {
    const int cc = (int)(i % cpt);
    const int64_t p = i / cpt;
    const int w = (int)(p % W), h = (int)((p / W) % H), b = (int)(p / ((int64_t)W * H));
    float acc[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) acc[e] = 0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int t = h + 1 - ky;
      if (t < 0 || (t & 1)) continue;
      const int ho = t >> 1;
      if (ho >= Ho) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int u = w + 1 - kx;
        if (u < 0 || (u & 1)) continue;
        const int wo = u >> 1;
        if (wo >= Wo) continue;
        float f[EPC];
        chunk_to_f<T>(reinterpret_cast<const uint4*>(dp)[(((int64_t)b * Ho + ho) * Wo + wo) * cpr + (ky * 3 + kx) * cpt + cc], f);
#pragma unroll
        for (int e = 0; e < EPC; ++e) acc[e] += f[e];
      }
    }
    reinterpret_cast<uint4*>(dx)[i] = f_to_chunk<T>(acc);
  }
}

__global__ __launch_bounds__(TPB) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, const float* __restrict__ wd_mask, bf16_t* __restrict__ shadow,
                                                   const int* __restrict__ step_dev, int64_t n, float lr, float b1, float b2, float eps, float wd,
                                                   float bc1, float bc2_sqrt) {
  if (step_dev) {                                   // step count lives on the device (captured graphs replay the launch unchanged)
    const float t = (float)*step_dev;
    bc1 = 1.f - powf(b1, t);
    bc2_sqrt = sqrtf(1.f - powf(b2, t));
  }
  const int64_t n4 = n >> 2;
  for (int64_t i = blockIdx.x * (int64_t)TPB + threadIdx.x; i < n4; i += (int64_t)gridDim.x * TPB) {
    float4 P = reinterpret_cast<float4*>(p)[i], G = reinterpret_cast<const float4*>(g)[i];
    float4 M = reinterpret_cast<float4*>(m)[i], V = reinterpret_cast<float4*>(v)[i];
    float4 W = wd_mask ? reinterpret_cast<const float4*>(wd_mask)[i] : make_float4(1.f, 1.f, 1.f, 1.f);
    float* pp = &P.x; const float* gg = &G.x; float* mm = &M.x; float* vv = &V.x; const float* ww = &W.x;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      pp[q] *= 1.f - lr * wd * ww[q];                       // decoupled weight decay
      mm[q] = b1 * mm[q] + (1.f - b1) * gg[q];
      vv[q] = b2 * vv[q] + (1.f - b2) * gg[q] * gg[q];
      const float denom = sqrtf(vv[q]) / bc2_sqrt + eps;
      pp[q] -= (lr / bc1) * mm[q] / denom;
    }
    reinterpret_cast<float4*>(p)[i] = P; reinterpret_cast<float4*>(m)[i] = M; reinterpret_cast<float4*>(v)[i] = V;
    if (shadow) reinterpret_cast<uint2*>(shadow)[i] = make_uint2(pack_bf2(P.x, P.y), pack_bf2(P.z, P.w));      // bf16 operand copy, refreshed in the same pass
  }
}

// exponential moving average of a flat fp32 parameter buffer: ema = decay * ema + (1 - decay) * param (timm.utils.ModelEmaV2, main.py:316 / engine.py: model_ema.update)
__global__ __launch_bounds__(TPB) void ema_kernel(float* __restrict__ ema, const float* __restrict__ p, int64_t n, float decay) {
  const int64_t n4 = n >> 2;
  const float w = 1.f - decay;
  for (int64_t i = blockIdx.x * (int64_t)TPB + threadIdx.x; i < n4; i += (int64_t)gridDim.x * TPB) {
    float4 E = reinterpret_cast<float4*>(ema)[i];
    const float4 P = reinterpret_cast<const float4*>(p)[i];
    E.x = fmaf(decay, E.x, w * P.x); E.y = fmaf(decay, E.y, w * P.y); E.z = fmaf(decay, E.z, w * P.z); E.w = fmaf(decay, E.w, w * P.w);
    reinterpret_cast<float4*>(ema)[i] = E;
  }
}

// out[map(i)] += sum_r partial[r][i]: the workgroup body lives in common.h (lmv_partial_reduce_block), shared with lmv_reduce_batch
__global__ __launch_bounds__(256) void partial_reduce_kernel(const float* __restrict__ partial, int nrows, int width, float* __restrict__ out_a, int na,
                                                            float* __restrict__ out_b, int mode) {
  __shared__ float4 red[256];
  lmv_partial_reduce_block(red, blockIdx.x, partial, nrows, width, out_a, na, out_b, mode);
}

}  // namespace

int lmv_launch_partial_reduce(const float* partial, int nrows, int width, float* out_a, int na, float* out_b, int mode, hipStream_t st) {
  if (width % 4) LMV_FAIL(LMV_ERR_SHAPE, "partial_reduce: width %d must be a multiple of 4", width);
  hipLaunchKernelGGL(partial_reduce_kernel, dim3((width / 4 + 7) / 8), dim3(256), 0, st, partial, nrows, width, out_a, na, out_b, mode);
  LMV_CHECK_LAUNCH("partial_reduce");
  return LMV_OK;
}

extern "C" int lmv_cast(const void* src, int sd, void* dst, int dd, int64_t n, void* stream) {
  if (n <= 0) return LMV_OK;
  if (!lmv_aligned16(src) || !lmv_aligned16(dst)) LMV_FAIL(LMV_ERR_SHAPE, "cast: pointers must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const int grid = grid_for(n / 4);
  if (sd == LMV_F32 && dd == LMV_BF16) hipLaunchKernelGGL((cast_kernel<float, bf16_t>), dim3(grid), dim3(TPB), 0, st, (const float*)src, (bf16_t*)dst, n);
  else if (sd == LMV_BF16 && dd == LMV_F32) hipLaunchKernelGGL((cast_kernel<bf16_t, float>), dim3(grid), dim3(TPB), 0, st, (const bf16_t*)src, (float*)dst, n);
  else if (sd == LMV_F32 && dd == LMV_F32) hipLaunchKernelGGL((cast_kernel<float, float>), dim3(grid), dim3(TPB), 0, st, (const float*)src, (float*)dst, n);
  else if (sd == LMV_BF16 && dd == LMV_BF16) hipLaunchKernelGGL((cast_kernel<bf16_t, bf16_t>), dim3(grid), dim3(TPB), 0, st, (const bf16_t*)src, (bf16_t*)dst, n);
  else LMV_FAIL(LMV_ERR_DTYPE, "cast: unsupported dtypes %d -> %d", sd, dd);
  LMV_CHECK_LAUNCH("cast");
  return LMV_OK;
}

// ---- stem, first convolution (models/lemevit.py:713: Conv2d(3, C/2, 3, stride 2, padding 1)) as a GEMM ------------
// patches[(b, ho, wo)][j] = x[b][ci][2 ho - 1 + ky][2 wo - 1 + kx],  j = ci * 9 + ky * 3 + kx  (the weight's own
// [Cout][3][3][3] order), j = 27..31 zero: a [B Ho Wo, 32] matrix that lmv_linear_fwd multiplies by the [Cout, 32]
// weight (forward) and lmv_linear_dw contracts with dY (weight gradient) -- the input needs no gradient.  The source is
// read through its strides (NCHW or channels-last, fp32 or bf16: the image batch as the data loader hands it over).
template <typename S, typename T>
__global__ __launch_bounds__(TPB) void im2col_c3_kernel(const S* __restrict__ x, T* __restrict__ out, int B, int H, int W, int Ho, int Wo,
                                                       int64_t sb, int64_t sc, int64_t sh, int64_t sw) {
  const unsigned p = blockIdx.x * TPB + threadIdx.x, total = (unsigned)B * Ho * Wo;
  if (p >= total) return;
  const unsigned wo = p % Wo, t = p / Wo, ho = t % Ho, b = t / Ho;
  float v[32];
#pragma unroll
  for (int ci = 0; ci < 3; ++ci)
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int h = 2 * (int)ho - 1 + ky, w = 2 * (int)wo - 1 + kx;
        const int hc = min(max(h, 0), H - 1), wc = min(max(w, 0), W - 1);      // clamped load, zeroed below
        const float val = DT<S>::ld(x + b * sb + ci * sc + hc * sh + wc * sw);
        v[ci * 9 + ky * 3 + kx] = (h == hc && w == wc) ? val : 0.f;
      }
#pragma unroll
  for (int j = 27; j < 32; ++j) v[j] = 0.f;
  T* o = out + (size_t)p * 32;
#pragma unroll
  for (int c = 0; c < 32; c += DT<T>::EPC) *reinterpret_cast<uint4*>(o + c) = f_to_chunk<T>(v + c);
}

extern "C" int lmv_im2col3x3s2_c3(const void* x, int x_dtype, void* patches, int dtype, int B, int H, int W, int64_t sb, int64_t sc, int64_t sh,
                                  int64_t sw, void* stream) {
  if (B <= 0 || H <= 0 || W <= 0) LMV_FAIL(LMV_ERR_SHAPE, "im2col: bad shape B=%d H=%d W=%d", B, H, W);
  if (!x || !patches || !lmv_aligned16(patches)) LMV_FAIL(LMV_ERR_SHAPE, "im2col: null or misaligned operand");
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  const int64_t total = (int64_t)B * Ho * Wo;
  if (total >= (int64_t)1 << 31) LMV_FAIL(LMV_ERR_SHAPE, "im2col: too many output pixels");
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)((total + TPB - 1) / TPB)), block(TPB);
#define IM2COL(S, T) hipLaunchKernelGGL((im2col_c3_kernel<S, T>), grid, block, 0, st, (const S*)x, (T*)patches, B, H, W, Ho, Wo, sb, sc, sh, sw)
  if (x_dtype == LMV_F32 && dtype == LMV_BF16) IM2COL(float, bf16_t);
  else if (x_dtype == LMV_F32 && dtype == LMV_F32) IM2COL(float, float);
  else if (x_dtype == LMV_BF16 && dtype == LMV_BF16) IM2COL(bf16_t, bf16_t);
  else if (x_dtype == LMV_BF16 && dtype == LMV_F32) IM2COL(bf16_t, float);
  else LMV_FAIL(LMV_ERR_DTYPE, "im2col: unsupported dtypes %d -> %d", x_dtype, dtype);
#undef IM2COL
  LMV_CHECK_LAUNCH("im2col3x3s2_c3");
  return LMV_OK;
}

extern "C" int lmv_row_scale_multi(const lmv_row_scale_segment* seg, int nseg, int C, int dtype, void* stream) {
  if (!seg || nseg < 1 || nseg > 2) LMV_FAIL(LMV_ERR_SHAPE, "row_scale: 1 or 2 segments");
  if (C <= 0 || (C % 8)) LMV_FAIL(LMV_ERR_SHAPE, "row_scale: C=%d must be a multiple of 8", C);
  if (dtype != LMV_BF16 && dtype != LMV_F32) LMV_FAIL(LMV_ERR_DTYPE, "row_scale: unsupported dtype %d", dtype);
  const unsigned cpr = (unsigned)(C / (dtype == LMV_BF16 ? 8 : 4));
  RowScaleArgs a{};
  int64_t total = 0;
  int k = 0;
  for (int i = 0; i < nseg; ++i) {
    if (seg[i].rows <= 0) continue;
    if (seg[i].rows_per_sample <= 0) LMV_FAIL(LMV_ERR_SHAPE, "row_scale: rows_per_sample must be > 0");
    if (!seg[i].x || !seg[i].y || !seg[i].scale || !lmv_aligned16(seg[i].x) || !lmv_aligned16(seg[i].y)) LMV_FAIL(LMV_ERR_SHAPE, "row_scale: null or misaligned operand");
    const int64_t chunks = seg[i].rows * cpr;
    total += chunks;
    if (total >= (int64_t)1 << 31) LMV_FAIL(LMV_ERR_SHAPE, "row_scale: tensor too large");
    a.s[k].x = seg[i].x; a.s[k].scale = seg[i].scale; a.s[k].y = seg[i].y;
    a.s[k].chunks = (unsigned)chunks; a.s[k].cpr = cpr; a.s[k].rps = (unsigned)seg[i].rows_per_sample;
    ++k;
  }
  if (k == 0) return LMV_OK;
  if (k == 1) { a.s[1] = a.s[0]; a.s[1].chunks = 0; }
  a.total = (unsigned)total;
  hipStream_t st = (hipStream_t)stream;
  const int grid = grid_for(total);
  if (dtype == LMV_BF16) hipLaunchKernelGGL((row_scale_kernel<bf16_t>), dim3(grid), dim3(TPB), 0, st, a);
  else hipLaunchKernelGGL((row_scale_kernel<float>), dim3(grid), dim3(TPB), 0, st, a);
  LMV_CHECK_LAUNCH("row_scale");
  return LMV_OK;
}

extern "C" int lmv_row_scale(const void* x, const float* scale, void* y, int64_t rows, int C, int rps, int dtype, void* stream) {
  lmv_row_scale_segment s{x, scale, y, rows, rps};
  return lmv_row_scale_multi(&s, 1, C, dtype, stream);
}

extern "C" int lmv_token_mean2_fwd(const void* x, int L, const void* c, int M, int C, int B, void* out, int dtype, void* stream) {
  if (B <= 0 || L <= 0 || C <= 0 || (c && M <= 0) || (C % 8)) LMV_FAIL(LMV_ERR_SHAPE, "token_mean2_fwd: bad shape B=%d L=%d M=%d C=%d", B, L, M, C);
  if (dtype != LMV_F32 && dtype != LMV_BF16) LMV_FAIL(LMV_ERR_DTYPE, "token_mean2_fwd: unsupported dtype %d", dtype);
  if (!x || !out || !lmv_aligned16(x) || !lmv_aligned16(c) || !lmv_aligned16(out)) LMV_FAIL(LMV_ERR_SHAPE, "token_mean2_fwd: null or misaligned operand");
  const int cpr = C / (dtype == LMV_BF16 ? 8 : 4);
  const int grid = (B * cpr + TPB - 1) / TPB;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == LMV_BF16) hipLaunchKernelGGL((token_mean2_fwd_kernel<bf16_t>), dim3(grid), dim3(TPB), 0, st, (const bf16_t*)x, L, (const bf16_t*)c, M, C, B, (bf16_t*)out);
  else hipLaunchKernelGGL((token_mean2_fwd_kernel<float>), dim3(grid), dim3(TPB), 0, st, (const float*)x, L, (const float*)c, M, C, B, (float*)out);
  LMV_CHECK_LAUNCH("token_mean2_fwd");
  return LMV_OK;
}

// out[b, :] = xscale * mean_l x[b, l, :] + xshift + mean_m c[b, m, :]: the inference tail with the final (eval-mode) BatchNorm folded into the pool
extern "C" int lmv_token_mean2_affine_fwd(const void* x, int L, const void* c, int M, int C, int B, const float* xscale, const float* xshift, void* out, int dtype,
                                          void* stream) {
  if (B <= 0 || L <= 0 || C <= 0 || (c && M <= 0) || (C % 8)) LMV_FAIL(LMV_ERR_SHAPE, "token_mean2_affine_fwd: bad shape B=%d L=%d M=%d C=%d", B, L, M, C);
  if (dtype != LMV_F32 && dtype != LMV_BF16) LMV_FAIL(LMV_ERR_DTYPE, "token_mean2_affine_fwd: unsupported dtype %d", dtype);
  if (!x || !out || !xscale || !xshift || !lmv_aligned16(x) || !lmv_aligned16(c) || !lmv_aligned16(out)) LMV_FAIL(LMV_ERR_SHAPE, "token_mean2_affine_fwd: null or misaligned operand");
  hipStream_t st = (hipStream_t)stream;
  const int cpr = C / (dtype == LMV_BF16 ? 8 : 4);
  const int grid = (int)(((int64_t)B * cpr + TPB - 1) / TPB);
  if (dtype == LMV_BF16) hipLaunchKernelGGL((token_mean2_fwd_kernel<bf16_t>), dim3(grid), dim3(TPB), 0, st, (const bf16_t*)x, L, (const bf16_t*)c, M, C, B, (bf16_t*)out, xscale, xshift);
  else hipLaunchKernelGGL((token_mean2_fwd_kernel<float>), dim3(grid), dim3(TPB), 0, st, (const float*)x, L, (const float*)c, M, C, B, (float*)out, xscale, xshift);
  LMV_CHECK_LAUNCH("token_mean2_affine_fwd");
  return LMV_OK;
}

extern "C" int lmv_token_mean2_bwd(const void* g, void* dx, int L, void* dc, int M, int C, int B, int dtype, void* stream) {
  if (B <= 0 || L <= 0 || C <= 0 || (dc && M <= 0) || (C % 8)) LMV_FAIL(LMV_ERR_SHAPE, "token_mean2_bwd: bad shape B=%d L=%d M=%d C=%d", B, L, M, C);
  if (dtype != LMV_F32 && dtype != LMV_BF16) LMV_FAIL(LMV_ERR_DTYPE, "token_mean2_bwd: unsupported dtype %d", dtype);
  if (!g || !dx || !lmv_aligned16(g) || !lmv_aligned16(dx) || !lmv_aligned16(dc)) LMV_FAIL(LMV_ERR_SHAPE, "token_mean2_bwd: null or misaligned operand");
  const int cpr = C / (dtype == LMV_BF16 ? 8 : 4);
  const int64_t tot = (int64_t)B * (L + (dc ? M : 0)) * cpr;
  hipStream_t st = (hipStream_t)stream;
  const int Mx = dc ? M : 0;
  if (dtype == LMV_BF16) hipLaunchKernelGGL((token_mean2_bwd_kernel<bf16_t>), dim3(grid_for(tot)), dim3(TPB), 0, st, (const bf16_t*)g, (bf16_t*)dx, L, (bf16_t*)dc, Mx, C, B);
  else hipLaunchKernelGGL((token_mean2_bwd_kernel<float>), dim3(grid_for(tot)), dim3(TPB), 0, st, (const float*)g, (float*)dx, L, (float*)dc, Mx, C, B);
  LMV_CHECK_LAUNCH("token_mean2_bwd");
  return LMV_OK;
}

static int conv_geom_check(const char* who, const void* a, const void* b, int B, int H, int W, int C, int KP, int dtype) {
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C % 8) || KP < 9 * C || (KP % 8)) LMV_FAIL(LMV_ERR_SHAPE, "%s: bad shape B=%d H=%d W=%d C=%d KP=%d", who, B, H, W, C, KP);
  if (dtype != LMV_F32 && dtype != LMV_BF16) LMV_FAIL(LMV_ERR_DTYPE, "%s: unsupported dtype %d", who, dtype);
  if (!a || !b || !lmv_aligned16(a) || !lmv_aligned16(b)) LMV_FAIL(LMV_ERR_SHAPE, "%s: null or misaligned operand", who);
  return LMV_OK;
}

extern "C" int lmv_im2col3x3s2_nhwc(const void* x, void* patches, int B, int H, int W, int C, int KP, int dtype, void* stream) {
  if (int rc = conv_geom_check("im2col3x3s2_nhwc", x, patches, B, H, W, C, KP, dtype)) return rc;
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  const int64_t total = (int64_t)B * Ho * Wo * (KP / (dtype == LMV_BF16 ? 8 : 4));
  hipStream_t st = (hipStream_t)stream;
  if (dtype == LMV_BF16) hipLaunchKernelGGL((im2col_nhwc_kernel<bf16_t>), dim3(grid_for(total)), dim3(TPB), 0, st, (const bf16_t*)x, (bf16_t*)patches, B, H, W, C, Ho, Wo, KP);
  else hipLaunchKernelGGL((im2col_nhwc_kernel<float>), dim3(grid_for(total)), dim3(TPB), 0, st, (const float*)x, (float*)patches, B, H, W, C, Ho, Wo, KP);
  LMV_CHECK_LAUNCH("im2col3x3s2_nhwc");
  return LMV_OK;
}

extern "C" int lmv_col2im3x3s2_nhwc(const void* dpatches, void* dx, int B, int H, int W, int C, int KP, int dtype, void* stream) {
  if (int rc = conv_geom_check("col2im3x3s2_nhwc", dpatches, dx, B, H, W, C, KP, dtype)) return rc;
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  const int64_t total = (int64_t)B * H * W * (C / (dtype == LMV_BF16 ? 8 : 4));
  hipStream_t st = (hipStream_t)stream;
  if (dtype == LMV_BF16) hipLaunchKernelGGL((col2im_nhwc_kernel<bf16_t>), dim3(grid_for(total)), dim3(TPB), 0, st, (const bf16_t*)dpatches, (bf16_t*)dx, B, H, W, C, Ho, Wo, KP);
  else hipLaunchKernelGGL((col2im_nhwc_kernel<float>), dim3(grid_for(total)), dim3(TPB), 0, st, (const float*)dpatches, (float*)dx, B, H, W, C, Ho, Wo, KP);
  LMV_CHECK_LAUNCH("col2im3x3s2_nhwc");
  return LMV_OK;
}

extern "C" int lmv_adamw_flat(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, const float* wd_mask, void* shadow_bf16,
                              int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay, int step, const int* step_dev,
                              void* stream) {
  if (n <= 0) return LMV_OK;
  if (n % 4) LMV_FAIL(LMV_ERR_SHAPE, "adamw_flat: n=%lld must be a multiple of 4 (pad the flat buffer)", (long long)n);
  if (!step_dev && step < 1) LMV_FAIL(LMV_ERR_SHAPE, "adamw_flat: step must be >= 1");
  if (!param || !grad || !exp_avg || !exp_avg_sq || !lmv_aligned16(param) || !lmv_aligned16(grad) || !lmv_aligned16(exp_avg) || !lmv_aligned16(exp_avg_sq) ||
      !lmv_aligned16(wd_mask) || ((uintptr_t)shadow_bf16 & 7u))
    LMV_FAIL(LMV_ERR_SHAPE, "adamw_flat: null or misaligned buffer");
  const float t = step_dev ? 1.f : (float)step;
  const float bc1 = 1.f - powf(beta1, t);
  const float bc2_sqrt = sqrtf(1.f - powf(beta2, t));
  hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n / 4)), dim3(TPB), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, wd_mask,
                     reinterpret_cast<bf16_t*>(shadow_bf16), step_dev, n, lr, beta1, beta2, eps, weight_decay, bc1, bc2_sqrt);
  LMV_CHECK_LAUNCH("adamw_flat");
  return LMV_OK;
}

extern "C" int lmv_ema_flat(float* ema, const float* param, int64_t n, float decay, void* stream) {
  if (!ema || !param || n <= 0 || (n % 4) || !lmv_aligned16(ema) || !lmv_aligned16(param)) LMV_FAIL(LMV_ERR_SHAPE, "ema_flat: null / misaligned buffer or n %% 4 != 0");
  if (!(decay >= 0.f && decay <= 1.f)) LMV_FAIL(LMV_ERR_SHAPE, "ema_flat: decay must be in [0, 1]");
  hipLaunchKernelGGL(ema_kernel, dim3(grid_for(n / 4)), dim3(TPB), 0, (hipStream_t)stream, ema, param, n, decay);
  LMV_CHECK_LAUNCH("ema_flat");
  return LMV_OK;
}

// ---- batched 2-byte matrix transpose: dst[c][r] = src[r][c] for up to LMV_TRANSPOSE_MAX_SEGS matrices in ONE launch ----------------------
// (the transposed bf16 copies of the mlp.3 weights that let the dX of fc2 run as a forward-form GEMM on the register-stationary kernel,
//  rsgemm.hip; refreshed once per optimizer step by FlatAdamW)
namespace {
struct TrSegDev { const unsigned short* src; unsigned short* dst; int rows, cols, tiles_c, tile0; };
struct TrBatch { TrSegDev s[LMV_TRANSPOSE_MAX_SEGS]; int n; };
__global__ __launch_bounds__(256) void transpose16_batch_kernel(const TrBatch tb) {
  __shared__ unsigned short tile[64][66];
  int si = 0;
#pragma unroll 1
  for (int k = 1; k < tb.n; ++k) if ((int)blockIdx.x >= tb.s[k].tile0) si = k;
  const TrSegDev g = tb.s[si];
  const int t = blockIdx.x - g.tile0, tr = t / g.tiles_c, tc = t % g.tiles_c;
  const int r0 = tr * 64, c0 = tc * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int r = r0 + ty * 16 + i, c = c0 + tx;
    tile[ty * 16 + i][tx] = (r < g.rows && c < g.cols) ? g.src[(int64_t)r * g.cols + c] : (unsigned short)0;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = c0 + ty * 16 + i, r = r0 + tx;
    if (r < g.rows && c < g.cols) g.dst[(int64_t)c * g.rows + r] = tile[tx][ty * 16 + i];
  }
}
}  // namespace
extern "C" int lmv_transpose_batch(const lmv_transpose_seg* segs, int nsegs, int dtype, void* stream) {
  if (nsegs <= 0) return LMV_OK;
  if (!segs) LMV_FAIL(LMV_ERR_SHAPE, "transpose_batch: segs is NULL");
  if (dtype != LMV_BF16) LMV_FAIL(LMV_ERR_DTYPE, "transpose_batch: 2-byte elements (bf16) only");
  hipStream_t st = (hipStream_t)stream;
  for (int s0 = 0; s0 < nsegs; s0 += LMV_TRANSPOSE_MAX_SEGS) {
    TrBatch tb{};
    tb.n = nsegs - s0 < LMV_TRANSPOSE_MAX_SEGS ? nsegs - s0 : LMV_TRANSPOSE_MAX_SEGS;
    int tiles = 0;
    for (int i = 0; i < tb.n; ++i) {
      const lmv_transpose_seg& q = segs[s0 + i];
      if (!q.src || !q.dst || q.rows <= 0 || q.cols <= 0 || q.src == q.dst) LMV_FAIL(LMV_ERR_SHAPE, "transpose_batch: bad segment %d", s0 + i);
      TrSegDev& d = tb.s[i];
      d.src = (const unsigned short*)q.src; d.dst = (unsigned short*)q.dst; d.rows = q.rows; d.cols = q.cols;
      d.tiles_c = (q.cols + 63) / 64; d.tile0 = tiles;
      tiles += ((q.rows + 63) / 64) * d.tiles_c;
    }
    hipLaunchKernelGGL(transpose16_batch_kernel, dim3(tiles), dim3(256), 0, st, tb);
  }
  LMV_CHECK_LAUNCH("transpose_batch");
  return LMV_OK;
}
