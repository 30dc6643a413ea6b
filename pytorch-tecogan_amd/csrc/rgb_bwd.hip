// Backward pass of the generator's output layer (conv 64 -> 3, code/models.py:77-79) for the batched G backward in ONE pass
// over its two big operands (16-bit element types):
//
//     dX[p][ci]     = relu'(x[p][ci]) * sum_{t, co} dpre[p - off(t)][co] * W[co][ci][t]          (input gradient, masked by x > 0)
//     dW[co][ci][t] = sum_p x[p][ci] * dpre[p - off(t)][co]                                      (weight gradient)
//
// with x = the layer input u4 [N,H,W,64] and dpre = d(loss)/d(pre-sigmoid) [N,H,W,4] (3 real channels + 1 pad: what
// tg_content_loss writes with dpre_channels = 4).  The generic path runs this layer as two launches padded to 32 gradient
// channels - tg_conv (K = 9 x 32, 27 of them real) and a half-empty 64 x 64 block of the weight-gradient work list - and reads
// the 84-MB input twice and a 42-MB padded dpre twice (336 MB per step at 40 samples of 128 x 128).  Here both products come from
// ONE im2col image Dm[p][k = co * 9 + t] = dpre[p - off(t)][co] (27 columns, padded to 32) built in LDS per 16 x 16 pixel tile:
//     dX^T[ci][p]  = W'[ci][k] * Dm^T[k][p]        one 16x16x32 MFMA per 16 pixels and 16 channels (K = 27)
//     dW'[k][ci]   = Dm^T[k][p] * x[p][ci]         transposed LDS reads of both images, K = the tile's 256 pixels
// so the launch moves x once in, dX once out and 8 bytes per pixel of dpre: 173 MB.  Workgroups are persistent (tile w, w + G,
// ...), keep dW' in registers and write one slab slot each in tg_wgrad's slab layout ([9][64][32] fp32, real columns only):
// the network's fold (tg_wgrad_fold_items) adds them into the PyTorch-layout gradient with the other layers' slabs.
//
// Replaces aten::convolution_backward of `output` behind code/train.py:336 (autograd of code/models.py:77-79).
#include "common.h"

namespace {

constexpr int kTile = 16, kPix = kTile * kTile;          // 16 x 16 pixels per tile
constexpr int kPatchW = kTile + 2;                       // dpre patch with a one-pixel halo
constexpr int kLdsX = kPix * 128;                        // x tile: [pixel][64 ch], 128-byte rows, 32-byte segments swizzled
constexpr int kLdsD = kPix * 64;                         // Dm: [pixel][32 k], 64-byte rows, the two segments swizzled
constexpr int kLdsP = kPatchW * kPatchW * 8;             // dpre patch, 8 bytes per pixel
constexpr int kLdsTotal = kLdsX + kLdsD + kLdsP;
constexpr int kSlot = 9 * 64 * 32;                       // floats per slab slot (tg_wgrad layout for Cx = 64, Cy = 32)

struct RgbBwdK {
  const char* dpre;
  const char* x;
  const float* w;
  char* dx;
  float* slab;
  int N, H, W, tiles_x, tiles_y, ntiles;
};

__device__ __attribute__((aligned(16))) unsigned int tg_rgbb_zero_page[4];

// LDS-DMA, one wave-instruction: lane l's 16 bytes land at lds_dst + 16 * l (see wgrad_group.hip)
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_dst)
      : "memory");
}

__device__ __forceinline__ bf16x8 tr_pair(const char* lo_addr, const char* hi_addr) {
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(lo_addr));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(hi_addr));
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  const s16x8 cat = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, cat);
}

template <typename T>
__global__ __launch_bounds__(256) void rgb_bwd_kernel(const RgbBwdK p) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  char* lds_x = smem;
  char* lds_d = smem + kLdsX;
  char* lds_p = smem + kLdsX + kLdsD;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int idx = lane & 15, g = lane >> 4, q = idx >> 2, pp = idx & 3;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  const char* zero = reinterpret_cast<const char*>(tg_rgbb_zero_page);

  // ---- W' fragments of the input gradient: row r = 4 g' + j of row tile mt is channel 16 g' + 4 mt + j, so that a lane ends up
  // with the 16 consecutive channels 16 g .. 16 g + 15 of its pixel; column k = co * 9 + t (27 real)
  bf16x8 wf[4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const int ci = 16 * (idx >> 2) + 4 * mt + (idx & 3);
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    s16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = 8 * g + e;
      const int co = k / 9, t = k - co * 9;
      const float f = k < 27 ? p.w[(co * 64 + ci) * 9 + t] : 0.f;
      v[e] = (short)f32_to_bits16<T>(f);
    }
    wf[mt] = __builtin_bit_cast(bf16x8, v);
  }

  f32x4 accw[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};   // dW'[k rows of tile mt][channels 16 wid + idx]

  // lane constants of the x DMA: chunk c (1 KiB = 8 pixel rows) of this wave, rows 8 c + lane / 8, physical piece lane % 8
  const int jp = lane & 7, r8 = lane >> 3;

  for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
    int r = tile;
    const int txb = r % p.tiles_x;
    r /= p.tiles_x;
    const int tyb = r % p.tiles_y;
    const int n = r / p.tiles_y;
    const int y0 = tyb * kTile, x0 = txb * kTile;
    // ---- x tile -> LDS by DMA (source-side swizzle: physical 32-byte segment s' of pixel k holds logical segment s' ^ key(k))
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int chunk = wid + 4 * c;
      const int k = chunk * 8 + r8;
      const int ty = k >> 4, tx = k & 15;
      const int key = (k >> 1) & 3;
      const int piece = (((jp >> 1) ^ key) * 2 + (jp & 1)) * 16;
      const bool ok = y0 + ty < p.H && x0 + tx < p.W;
      const char* src = p.x + (((size_t)n * p.H + (y0 + ty)) * p.W + (x0 + tx)) * 128 + piece;
      glds16(ok ? src : zero, lds0 + chunk * 1024);
    }
    // ---- dpre patch (one-pixel halo, zero outside the image)
    uint2 pv[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int e = tid + 256 * u;
      const int py = e / kPatchW, px = e - py * kPatchW;
      const int iy = y0 - 1 + py, ix = x0 - 1 + px;
      const bool ok = e < kPatchW * kPatchW && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
      const int cy = min(max(iy, 0), p.H - 1), cx = min(max(ix, 0), p.W - 1);
      pv[u] = *reinterpret_cast<const uint2*>(p.dpre + (((size_t)n * p.H + cy) * p.W + cx) * 8);
      if (!ok) pv[u] = uint2{0u, 0u};
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
      if (tid + 256 * u < kPatchW * kPatchW) *reinterpret_cast<uint2*>(lds_p + (tid + 256 * u) * 8) = pv[u];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // my DMA pieces have landed
    __syncthreads();

    // ---- Dm row of pixel tid: column co * 9 + t = dpre[pixel - off(t)][co]
    {
      const int ty = tid >> 4, tx = tid & 15;
      unsigned short h[32];
#pragma unroll
      for (int i = 27; i < 32; ++i) h[i] = 0;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int oy = t / 3 - 1, ox = t % 3 - 1;
        const uint2 d = *reinterpret_cast<const uint2*>(lds_p + ((ty + 1 - oy) * kPatchW + (tx + 1 - ox)) * 8);
        h[t] = (unsigned short)(d.x & 0xffffu);
        h[9 + t] = (unsigned short)(d.x >> 16);
        h[18 + t] = (unsigned short)(d.y & 0xffffu);
      }
      const int sw = (tid >> 2) & 1;
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          u32x4 v;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int b = s * 16 + hf * 8 + 2 * j;
            v[j] = (unsigned)h[b] | ((unsigned)h[b + 1] << 16);
          }
          *reinterpret_cast<u32x4*>(lds_d + tid * 64 + ((s ^ sw) * 32) + hf * 16) = v;
        }
    }
    __syncthreads();

    // ---- input gradient: wave `wid` takes tile rows 4 wid .. 4 wid + 3 (16 pixels each)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = (4 * wid + j) * 16 + idx;
      const bf16x8 bfr = *reinterpret_cast<const bf16x8*>(lds_d + k * 64 + (((g >> 1) ^ ((k >> 2) & 1)) * 32) + (g & 1) * 16);
      f32x4 acc[4];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) acc[mt] = Mma16<T>::run(wf[mt], bfr, f32x4{0.f, 0.f, 0.f, 0.f});
      const char* xr = lds_x + k * 128 + ((g ^ ((k >> 1) & 3)) * 32);
      const u32x4 m0 = *reinterpret_cast<const u32x4*>(xr), m1 = *reinterpret_cast<const u32x4*>(xr + 16);
      const int y = y0 + 4 * wid + j, x = x0 + idx;
      u32x4 o0, o1;
#pragma unroll
      for (int e = 0; e < 8; ++e) {   // channels 16 g + e (row tiles 0, 1) and 16 g + 8 + e (row tiles 2, 3)
        const unsigned w0 = m0[e >> 1], w1 = m1[e >> 1];
        const float x0v = bits16_to_f32<T>((unsigned short)((e & 1) ? (w0 >> 16) : (w0 & 0xffffu)));
        const float x1v = bits16_to_f32<T>((unsigned short)((e & 1) ? (w1 >> 16) : (w1 & 0xffffu)));
        const unsigned short a = f32_to_bits16<T>(x0v > 0.f ? acc[e >> 2][e & 3] : 0.f);
        const unsigned short b = f32_to_bits16<T>(x1v > 0.f ? acc[2 + (e >> 2)][e & 3] : 0.f);
        if (e & 1) {
          o0[e >> 1] |= (unsigned)a << 16;
          o1[e >> 1] |= (unsigned)b << 16;
        } else {
          o0[e >> 1] = a;
          o1[e >> 1] = b;
        }
      }
      if (y < p.H && x < p.W) {
        char* o = p.dx + (((size_t)n * p.H + y) * p.W + x) * 128 + g * 32;
        tg_store16(o, o0);
        tg_store16(o + 16, o1);
      }
    }
    // ---- weight gradient: wave `wid` owns channels 16 wid .. 16 wid + 15, both k-row tiles; K = the tile's 256 pixels
#pragma unroll
    for (int s = 0; s < kPix / 32; ++s) {
      const int k = 32 * s + 4 * g + q;
      const char* xa = lds_x + k * 128 + ((wid ^ ((k >> 1) & 3)) * 32) + 8 * pp;
      const bf16x8 bfr = tr_pair(xa, xa + 16 * 128);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const char* da = lds_d + k * 64 + ((mt ^ ((k >> 2) & 1)) * 32) + 8 * pp;
        accw[mt] = Mma16<T>::run(tr_pair(da, da + 16 * 64), bfr, accw[mt]);
      }
    }
    __syncthreads();   // every wave is done with the LDS images before the next tile's DMA
  }

  // ---- dW' -> this workgroup's slab slot, tg_wgrad layout [t][ci][co of 32]
  float* slot = p.slab + (size_t)blockIdx.x * kSlot;
  const int ci = 16 * wid + idx;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = 16 * mt + 4 * g + j;
      if (k < 27) {
        const int co = k / 9, t = k - co * 9;
        slot[(t * 64 + ci) * 32 + co] = accw[mt][j];
      }
    }
  if (g == 3) {   // column 3 of every (t, ci): the fold reads four columns at a time (only the real ones are used)
#pragma unroll
    for (int t = 0; t < 9; ++t) slot[(t * 64 + ci) * 32 + 3] = 0.f;
  }
}

}  // namespace

extern "C" int64_t tg_conv3x3_rgb_bwd_slot_floats(void) { return kSlot; }

extern "C" int tg_conv3x3_rgb_bwd(int dtype, const void* dpre4, const void* x, const float* w, void* dx, float* slab, int N, int H,
                                  int W, int Cin, int max_workgroups, void* stream) {
  if (!dpre4 || !x || !w || !dx || !slab || N <= 0 || H <= 0 || W <= 0 || max_workgroups <= 0) return TG_E_BADARG;
  if ((dtype != TG_BF16 && dtype != TG_F16) || Cin != 64) return TG_E_UNSUPPORTED;
  if (!tg_aligned16(x) || !tg_aligned16(dx) || !tg_aligned16(slab) || ((size_t)dpre4 & 7)) return TG_E_ALIGN;
  RgbBwdK k;
  k.dpre = (const char*)dpre4; k.x = (const char*)x; k.w = w; k.dx = (char*)dx; k.slab = slab;
  k.N = N; k.H = H; k.W = W;
  k.tiles_x = (W + kTile - 1) / kTile; k.tiles_y = (H + kTile - 1) / kTile;
  const long long nt = (long long)k.tiles_x * k.tiles_y * N;
  if (nt > 0x7fffffffLL) return TG_E_UNSUPPORTED;
  k.ntiles = (int)nt;
  const int nwg = (int)(nt < max_workgroups ? nt : max_workgroups);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == TG_F16) hipLaunchKernelGGL(rgb_bwd_kernel<F16>, dim3((unsigned)nwg), dim3(256), kLdsTotal, st, k);
  else hipLaunchKernelGGL(rgb_bwd_kernel<BF16>, dim3((unsigned)nwg), dim3(256), kLdsTotal, st, k);
  return tg_launch_status();
}
