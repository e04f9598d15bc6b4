// stem.hip -- the stem of the backbone as ONE launch (round 4; inference form, BatchNorm folded into the convolutions):
//     Conv2d(3, C/2, 3, stride 2, pad 1) -> BatchNorm -> GELU -> Conv2d(C/2, C, 3, stride 2, pad 1) -> BatchNorm        (models/lemevit.py:698-704, used at :713)
// on [B, 3, H, W] images (any strides, fp32 or bf16) -> [B, H/4 * W/4, C] bf16 tokens.  The per-launch form materialises two patch matrices (im2col, 64 B and
// 9 C/2 x 2 B per output pixel) and the C/2-channel map between the convolutions: ~2 GB of HBM traffic for 256 images of 224 x 224 against 0.26 GB of input + output.
// Here a workgroup owns an 8 x 8 tile of OUTPUT pixels: it stages the 35 x 35 x 3 input patch in LDS, computes the 17 x 17 x C/2 tile of the first convolution from
// it on the matrix cores (K = 27 -> 32, operand gathered from the patch), keeps that tile in LDS (bf16, NHWC: what conv2's zero padding sees outside the map is
// written as zeros) and runs the second convolution on it: for a fixed kernel row the 3 x C/2 inputs of an output pixel are CONTIGUOUS in the NHWC tile, so a
// k-step of 32 is one ds_read_b128 per lane at a computed address -- no patch matrix anywhere.  conv2's weights (83 KB at C = 96) stay in LDS in MFMA-fragment
// order for the life of the persistent workgroup; the next tile's input patch is in flight (registers) while the current tile computes.
#include <atomic>
#include "stage_common.h"

namespace {

template <int CM, int CO> struct SG2 {
  static constexpr int KR = (3 * CM + 31) / 32 * 32, KSR = KR / 32;          // a kernel row of conv2: 3 CM inputs, padded to whole k-steps (zero weights behind 3 CM)
  static constexpr int NCO = CO / 16, NCM = CM / 16, NC2 = NCO / 2;          // conv2: a wave owns 2 pixel tiles x NC2 channel tiles
  static constexpr int W2_FRAGS = 3 * KSR * NCO;                             // [ky][ks][co tile]
  static constexpr int W1_FRAGS = 2 * NCM;                                   // [k-step][cm tile]: conv1 as K = 64 = 16 kernel rows (ci, ky) x 4 columns (kx 0..2, pad), 9 rows used
  static constexpr int L_W2 = 0, L_W2_BYTES = W2_FRAGS * 1024;
  // T1: the 17 x 17 x CM tile between the convolutions, bf16; pixel stride 2 CM + 16 B, row stride 17 pixels + a pad.  The pad is chosen for the lane groups a ds_read_b128
  // really has on gfx950 ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ...: MI355X_MICROARCH.md, LDS) -- round 5, tools/lds_conflicts.py: with the pads of round 4 (+80 / +112 B,
  // picked for 16 CONSECUTIVE lanes per group) every conv2 operand read took 8 LDS cycles instead of 4 (SQ_LDS_BANK_CONFLICT 0.39 of the kernel's LDS cycles); +16 B at
  // CM = 48 and +48 B at CM = 32 are conflict-free for these reads and leave conv1's 8-byte stores as they were.
  static constexpr int T1_PS = 2 * CM + 16, T1_RS = 17 * T1_PS + (CM == 48 ? 16 : 48);
  static constexpr int T1_PIX = 17 * 17, L_T1 = L_W2_BYTES, L_T1_BYTES = 17 * T1_RS + 256;      // (+ slack: the padded tail of a kernel row reads past the last pixel)
  static constexpr int P_ROW = 36, P_CH = 35 * P_ROW, L_P = L_T1 + (L_T1_BYTES + 15) / 16 * 16, L_P_BYTES = (3 * P_CH + 8) * 2;      // input patch, bf16 [3][35][36] (+ a zero word for k >= 27)
  static constexpr int L_TOTAL = L_P + (L_P_BYTES + 15) / 16 * 16;
  static constexpr int PF = (3 * 35 * 35 + 255) / 256;                       // patch elements per thread
  static_assert(NCO % 2 == 0 && CM % 16 == 0, "shape");
};

struct StemArgs {
  const void* x; int x_bf16; long long sb, sc, sh, sw;
  const uint4* wpk;                  // [W1: NCM fragments][W2: 3 KSR NCO fragments]
  const float* b1; const float* b2;
  bf16_t* y;
  int B, H, W, H2, W2, tiles_y, tiles_x, ntiles;
  unsigned long long* timing;          // optional: s_memtime stamps [workgroup][wave][8] of the workgroup's second tile (tools/stem_timeline.py)
};

template <int CM, int CO, bool XB>          // XB: bf16 images
__global__ __launch_bounds__(256, 2) void stem_kernel(const StemArgs a) {
  using G = SG2<CM, CO>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 4, li = lane & 15;
  {          // conv2's weights: resident for all tiles of this workgroup
    uint4* dst = reinterpret_cast<uint4*>(smem + G::L_W2);
    const uint4* src = a.wpk + G::W1_FRAGS * 64;
    for (int i = tid; i < G::W2_FRAGS * 64; i += 256) dst[i] = src[i];
    for (int i = tid; i < (G::L_TOTAL - G::L_P) / 4; i += 256) reinterpret_cast<unsigned*>(smem + G::L_P)[i] = 0u;          // (the pad column of the patch is read against zero weights)
    for (int i = tid; i < G::L_T1_BYTES / 4; i += 256) reinterpret_cast<unsigned*>(smem + G::L_T1)[i] = 0u;          // (pads and slack are read against zero weights: keep them finite)
  }
  bf16x8_t w1f[2][G::NCM];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int n = 0; n < G::NCM; ++n) w1f[s][n] = as_bf8(a.wpk[(s * G::NCM + n) * 64 + lane]);
  // conv1's operand: the lane's 8 k-slots of k-step s are TWO kernel rows (ci, ky) = row 8 s + 2 g + h, 4 consecutive patch columns each (kx 0..2 and a zero-weight pad):
  // two 8-byte LDS reads that ARE the packed fragment (byte offsets from the pixel's patch origin; rows >= 9 carry zero weights and read row 0)
  int roff[2][2];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int h = 0; h < 2; ++h) { const int row = 8 * s + 2 * g + h, rr = row < 9 ? row : 0; roff[s][h] = ((rr / 3 * 35 + rr % 3) * G::P_ROW) * 2; }
  // the wave's pixel tiles of the 17 x 17 tile (19 tiles of 16 over 4 waves): patch origin and T1 address of the lane's pixel
  int c1p[5], c1t[5];
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    const int q = 16 * (wave + 4 * k) + li, qq = q < G::T1_PIX ? q : G::T1_PIX - 1, ly = qq / 17, lx = qq - 17 * ly;
    c1p[k] = ((2 * ly) * G::P_ROW + 2 * lx) * 2 | (ly << 16) | (lx << 24);
    c1t[k] = q < G::T1_PIX ? ly * G::T1_RS + lx * G::T1_PS + 8 * g : -1;
  }
  unsigned short* const Pw = reinterpret_cast<unsigned short*>(smem + G::L_P);

  // biases in registers: a global load inside the tile loop would queue behind the next tile's patch loads (vmcnt counts in order) and stall the phase for a DRAM round trip
  float4 bias1[G::NCM], bias2[G::NC2];
#pragma unroll
  for (int n = 0; n < G::NCM; ++n) bias1[n] = *reinterpret_cast<const float4*>(a.b1 + 16 * n + 4 * g);
#pragma unroll
  for (int n = 0; n < G::NC2; ++n) bias2[n] = *reinterpret_cast<const float4*>(a.b2 + 16 * ((wave >> 1) * G::NC2 + n) + 4 * g);
  // the thread's patch elements: (channel, row, column) -> packed [LDS index | row << 16 | column << 24] and the offset inside the image's patch, once
  int pidx[G::PF]; long long poff[G::PF];
#pragma unroll
  for (int q = 0; q < G::PF; ++q) {
    const int e = tid + 256 * q, ee = e < 3 * 35 * 35 ? e : 0, ci = ee / 1225, rr = ee - ci * 1225, py = rr / 35, px = rr - py * 35;
    pidx[q] = e < 3 * 35 * 35 ? (ci * G::P_CH + py * G::P_ROW + px) | (py << 16) | (px << 24) : -1;
    poff[q] = ci * a.sc + py * a.sh + px * a.sw;
  }
  float pre[G::PF];
  unsigned pmask = 0u;
  auto fetch = [&](int tile) {          // the 35 x 35 x 3 input patch of a tile -> registers (zero outside the image)
    const int b = tile / (a.tiles_y * a.tiles_x), r = tile - b * (a.tiles_y * a.tiles_x), ty = r / a.tiles_x, tx = r - ty * a.tiles_x;
    const int y0 = 32 * ty - 3, x0 = 32 * tx - 3;
    const long long img0 = b * a.sb, base = img0 + y0 * a.sh + x0 * a.sw;
    // every load is issued unconditionally (an element outside the image reads the image's first pixel and is masked when the patch is written): a load inside a divergent
    // branch makes the compiler drain vmcnt at the join, one DRAM round trip per element
    pmask = 0u;
#pragma unroll
    for (int q = 0; q < G::PF; ++q) {
      const int iy = y0 + ((pidx[q] >> 16) & 255), ix = x0 + ((pidx[q] >> 24) & 255);
      const bool inb = pidx[q] >= 0 && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
      const long long off = inb ? base + poff[q] : img0;
      pmask |= inb ? 1u << q : 0u;
      if constexpr (XB) pre[q] = __uint_as_float((unsigned)reinterpret_cast<const bf16_t*>(a.x)[off] << 16);
      else pre[q] = reinterpret_cast<const float*>(a.x)[off];
    }
  };
  int tile = blockIdx.x;
  if (tile < a.ntiles) fetch(tile);
  __syncthreads();
#pragma unroll 1
  for (; tile < a.ntiles; tile += gridDim.x) {
#define ST_STAMP(k) do { if (a.timing && tile == (int)(blockIdx.x + gridDim.x) && lane == 0) a.timing[((size_t)blockIdx.x * 4 + wave) * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
    ST_STAMP(0);
    const int b = tile / (a.tiles_y * a.tiles_x), r = tile - b * (a.tiles_y * a.tiles_x), ty = r / a.tiles_x, tx = r - ty * a.tiles_x;
    // ---- patch -> LDS (bf16); the next tile's patch -> registers ----
#pragma unroll
    for (int q = 0; q < G::PF; ++q)
      if (pidx[q] >= 0) Pw[pidx[q] & 0xffff] = f2bf((pmask >> q) & 1u ? pre[q] : 0.f);
    __syncthreads();
    ST_STAMP(1);
    if (tile + (int)gridDim.x < a.ntiles) fetch(tile + gridDim.x);
    ST_STAMP(2);
    // ---- conv1 + bias + GELU on the 17 x 17 tile (19 pixel tiles of 16 over the 4 waves) -> T1 [17][17][CM] bf16; zeros where conv2 pads ----
    {
      const int H1 = a.H >> 1, W1 = a.W >> 1;
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        if (wave + 4 * k < 19) {
          const unsigned char* pp = smem + G::L_P + (c1p[k] & 0xffff);
          const int ly = (c1p[k] >> 16) & 255, lx = (c1p[k] >> 24) & 255;
          u32x4_t pk[2];
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) {
            const unsigned* r0 = reinterpret_cast<const unsigned*>(pp + roff[s2][0]);          // (4-byte aligned: two dwords, not one 8-byte read)
            const unsigned* r1 = reinterpret_cast<const unsigned*>(pp + roff[s2][1]);
            pk[s2] = u32x4_t{r0[0], r0[1], r1[0], r1[1]};
          }
          const bool inside = (unsigned)(16 * ty - 1 + ly) < (unsigned)H1 && (unsigned)(16 * tx - 1 + lx) < (unsigned)W1;
          f32x2_t h[2 * G::NCM + 2];
#pragma unroll
          for (int n = 0; n < G::NCM; ++n) {
            f32x4_t d = mfma_bf16(w1f[0][n], as_bf8(pk[0]), f32x4_t{0.f, 0.f, 0.f, 0.f});          // D[channel 16 n + 4 g + r][pixel li]
            d = mfma_bf16(w1f[1][n], as_bf8(pk[1]), d);
            h[2 * n] = f32x2_t{d[0] + bias1[n].x, d[1] + bias1[n].y}; h[2 * n + 1] = f32x2_t{d[2] + bias1[n].z, d[3] + bias1[n].w};
          }
          h[2 * G::NCM] = h[0]; h[2 * G::NCM + 1] = h[1];
          gelu4(h[0], h[1], h[2], h[3]);
          if constexpr (G::NCM > 2) gelu4(h[4], h[5], h[6], h[7]);
          if (c1t[k] >= 0) {
#pragma unroll
            for (int n = 0; n < G::NCM; ++n)
              *reinterpret_cast<uint2*>(smem + G::L_T1 + c1t[k] + 32 * n) = inside ? make_uint2(pack_bf2(h[2 * n][0], h[2 * n][1]), pack_bf2(h[2 * n + 1][0], h[2 * n + 1][1])) : make_uint2(0u, 0u);
          }
        }
      }
    }
    ST_STAMP(3);
    __syncthreads();
    ST_STAMP(4);
    // ---- conv2 + bias: wave w = (pixel-tile pair w & 1) x (channel-tile group w >> 1) ----
    {
      const int pp = wave & 1, cg = wave >> 1;
      f32x4_t acc[2][G::NC2];
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int n = 0; n < G::NC2; ++n) acc[p][n] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      int tbase[2], koff2[G::KSR];          // a lane's 8 inputs of a k-step lie inside one pixel (8 | CM): pixel (32 ks + 8 g) / CM of the run, channel (32 ks + 8 g) % CM
#pragma unroll
      for (int p = 0; p < 2; ++p) { const int o = 16 * (2 * pp + p) + li, oy = o >> 3, ox = o & 7; tbase[p] = (2 * oy) * G::T1_RS + (2 * ox) * G::T1_PS; }
#pragma unroll
      for (int ks = 0; ks < G::KSR; ++ks) { const int e = 32 * ks + 8 * g; koff2[ks] = (e / CM) * G::T1_PS + (e % CM) * 2; }
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int ks = 0; ks < G::KSR; ++ks) {
          bf16x8_t xf[2];
#pragma unroll
          for (int p = 0; p < 2; ++p) xf[p] = *reinterpret_cast<const bf16x8_t*>(smem + G::L_T1 + tbase[p] + ky * G::T1_RS + koff2[ks]);
#pragma unroll
          for (int n = 0; n < G::NC2; ++n) {
            const bf16x8_t wf = *reinterpret_cast<const bf16x8_t*>(smem + G::L_W2 + (((ky * G::KSR + ks) * G::NCO + cg * G::NC2 + n) * 64 + lane) * 16);
#pragma unroll
            for (int p = 0; p < 2; ++p) acc[p][n] = mfma_bf16(wf, xf[p], acc[p][n]);
          }
        }
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const int o = 16 * (2 * pp + p) + li, oy = 8 * ty + (o >> 3), ox = 8 * tx + (o & 7);
        bf16_t* dst = a.y + (((size_t)b * a.H2 + oy) * a.W2 + ox) * CO + 4 * g;
#pragma unroll
        for (int n = 0; n < G::NC2; ++n) {
          const int c0 = 16 * (cg * G::NC2 + n);
          const float4 bb = bias2[n];
          *reinterpret_cast<uint2*>(dst + c0) = make_uint2(pack_bf2(acc[p][n][0] + bb.x, acc[p][n][1] + bb.y), pack_bf2(acc[p][n][2] + bb.z, acc[p][n][3] + bb.w));
        }
      }
    }
    ST_STAMP(5);
    __syncthreads();          // T1 and the patch are rewritten by the next tile
    ST_STAMP(6);
  }
}

// wpk: [2 NCM fragments of W1, [k-step s][tile n]: lane (g, i) holds the conv1 weight of channel 16 n + i at kernel rows (ci, ky) = 8 s + 2 g + (j >> 2), column kx = j & 3 (zero for
// rows >= 9 and kx = 3)] [3 KSR NCO fragments of W2: lane (g, i) holds the conv2 weight of output channel
// 16 n + i at kernel row ky, inputs e = 32 ks + 8 g + j of the row's run (e = kx CM + ci; zero for e >= 3 CM)]
template <int CM, int CO>
__global__ __launch_bounds__(256) void stem_pack_kernel(const bf16_t* w1m, const bf16_t* w2m, int ld2, uint4* out) {
  using G = SG2<CM, CO>;
  const int f = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63, g = lane >> 4, i = lane & 15;
  if (f >= G::W1_FRAGS + G::W2_FRAGS) return;
  unsigned short v[8];
  if (f < G::W1_FRAGS) {
    const int s2 = f / G::NCM, n = f - s2 * G::NCM;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int row = 8 * s2 + 2 * g + (j >> 2), kx = j & 3;          // row = ci 3 + ky
      v[j] = row < 9 && kx < 3 ? w1m[(size_t)(16 * n + i) * 32 + row * 3 + kx] : (unsigned short)0;
    }
  } else {
    const int q = f - G::W1_FRAGS, n = q % G::NCO, ks = (q / G::NCO) % G::KSR, ky = q / (G::NCO * G::KSR);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int e = 32 * ks + 8 * g + j;
      v[j] = e < 3 * CM ? w2m[(size_t)(16 * n + i) * ld2 + ky * 3 * CM + e] : (unsigned short)0;
    }
  }
  out[(size_t)f * 64 + lane] = make_uint4(v[0] | (v[1] << 16), v[2] | (v[3] << 16), v[4] | (v[5] << 16), v[6] | (v[7] << 16));
}

}  // namespace

static int stem_variant(int Cm, int Co) { return Cm == 48 && Co == 96 ? 1 : Cm == 32 && Co == 64 ? 2 : 0; }
int lmv_stem_supported(int H, int W, int Cm, int Co, int dtype) { return dtype == LMV_BF16 && stem_variant(Cm, Co) != 0 && H > 0 && W > 0 && H % 32 == 0 && W % 32 == 0; }
size_t lmv_stem_wpk_bytes(int Cm, int Co) {
  const int v = stem_variant(Cm, Co);
  return v == 1 ? (size_t)(SG2<48, 96>::W1_FRAGS + SG2<48, 96>::W2_FRAGS) * 1024 : v == 2 ? (size_t)(SG2<32, 64>::W1_FRAGS + SG2<32, 64>::W2_FRAGS) * 1024 : 0;
}
int lmv_stem_pack(const void* w1m, const void* w2m, int ld2, int Cm, int Co, void* wpk_out, void* stream) {
  const int v = stem_variant(Cm, Co);
  if (!v) LMV_FAIL(LMV_ERR_DTYPE, "stem_pack: %d -> %d channels is not a supported stem", Cm, Co);
  if (!w1m || !w2m || !wpk_out || !lmv_aligned16(wpk_out) || ld2 < 9 * Cm) LMV_FAIL(LMV_ERR_SHAPE, "stem_pack: bad argument");
  hipStream_t st = (hipStream_t)stream;
  if (v == 1) hipLaunchKernelGGL((stem_pack_kernel<48, 96>), dim3((SG2<48, 96>::W1_FRAGS + SG2<48, 96>::W2_FRAGS + 3) / 4), dim3(256), 0, st, (const bf16_t*)w1m, (const bf16_t*)w2m, ld2, (uint4*)wpk_out);
  else hipLaunchKernelGGL((stem_pack_kernel<32, 64>), dim3((SG2<32, 64>::W1_FRAGS + SG2<32, 64>::W2_FRAGS + 3) / 4), dim3(256), 0, st, (const bf16_t*)w1m, (const bf16_t*)w2m, ld2, (uint4*)wpk_out);
  LMV_CHECK_LAUNCH("stem_pack");
  return LMV_OK;
}

template <int CM, int CO, bool XB> static int stem_launch(const StemArgs& a, hipStream_t st) {
  using G = SG2<CM, CO>;
  static std::atomic<unsigned long long> attr_done{0};
  int dev = 0, cus = 256;
  (void)hipGetDevice(&dev);
  const unsigned long long bit = 1ull << (dev & 63);
  if (!(attr_done.load(std::memory_order_acquire) & bit)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(stem_kernel<CM, CO, XB>), hipFuncAttributeMaxDynamicSharedMemorySize, G::L_TOTAL) != hipSuccess) LMV_FAIL(LMV_ERR_LAUNCH, "stem_fwd: cannot reserve LDS");
    attr_done.fetch_or(bit, std::memory_order_release);
  }
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  const int per = 160 * 1024 / G::L_TOTAL, grid = cus * (per < 1 ? 1 : per > 2 ? 2 : per);
  hipLaunchKernelGGL((stem_kernel<CM, CO, XB>), dim3(grid < a.ntiles ? grid : a.ntiles), dim3(256), G::L_TOTAL, st, a);
  LMV_CHECK_LAUNCH("stem_fwd");
  return LMV_OK;
}

static unsigned long long* g_stem_timing = nullptr;          // (debug: tools/stem_timeline.py)
extern "C" void lmv_stem_debug_timing(void* buf) { g_stem_timing = (unsigned long long*)buf; }

int lmv_stem_fwd(const void* x, int x_dtype, int64_t sb, int64_t sc, int64_t sh, int64_t sw, int B, int H, int W, int Cm, int Co, const void* wpk, const float* b1, const float* b2,
                 void* y, void* stream) {
  const int v = stem_variant(Cm, Co);
  if (!v || !lmv_stem_supported(H, W, Cm, Co, LMV_BF16)) LMV_FAIL(LMV_ERR_DTYPE, "stem_fwd: unsupported stem (%d -> %d channels, %d x %d)", Cm, Co, H, W);
  if (x_dtype != LMV_F32 && x_dtype != LMV_BF16) LMV_FAIL(LMV_ERR_DTYPE, "stem_fwd: images must be fp32 or bf16");
  if (!x || !wpk || !b1 || !b2 || !y || B <= 0 || !lmv_aligned16(wpk) || !lmv_aligned16(y) || !lmv_aligned16(b1) || !lmv_aligned16(b2)) LMV_FAIL(LMV_ERR_SHAPE, "stem_fwd: null or misaligned argument");
  StemArgs a{};
  a.x = x; a.x_bf16 = x_dtype == LMV_BF16; a.sb = sb; a.sc = sc; a.sh = sh; a.sw = sw; a.wpk = (const uint4*)wpk; a.b1 = b1; a.b2 = b2; a.y = (bf16_t*)y;
  a.B = B; a.H = H; a.W = W; a.H2 = H / 4; a.W2 = W / 4; a.tiles_y = a.H2 / 8; a.tiles_x = a.W2 / 8;
  const long long nt = (long long)B * a.tiles_y * a.tiles_x;
  if (nt >= (1ll << 31)) LMV_FAIL(LMV_ERR_SHAPE, "stem_fwd: too many tiles");
  a.ntiles = (int)nt; a.timing = g_stem_timing;
  LmvTimedLaunch timed(stream, (double)B * ((H / 2.0) * (W / 2.0) * 27.0 * Cm + (H / 4.0) * (W / 4.0) * 9.0 * Cm * Co) * 2.0,
                       (double)B * (3.0 * H * W * (a.x_bf16 ? 2.0 : 4.0) + (H / 4.0) * (W / 4.0) * Co * 2.0), 3);
  hipStream_t st = (hipStream_t)stream;
  if (v == 1) return a.x_bf16 ? stem_launch<48, 96, true>(a, st) : stem_launch<48, 96, false>(a, st);
  return a.x_bf16 ? stem_launch<32, 64, true>(a, st) : stem_launch<32, 64, false>(a, st);
}
