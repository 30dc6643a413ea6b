// The generator's output layer: conv3x3 64 -> 3 + bias + sigmoid, stored as fp32 NCHW straight into the frame buffer the
// next recurrent pass warps (code/models.py:77-79; the store replaces the permute + float() of code/train.py:97-99).
//
// On the generic path (conv_mfma.hip) this layer is a 32-output-channel launch - the packed weights pad 3 channels to two
// 16-row MFMA tiles - through the general epilogue (sigmoid, strided NCHW store): 11.2 us per frame on 4 x 128x128 pixels,
// ten times per step on the serial recurrent chain.  Here ONE 16-row tile is multiplied (rows 0-2 live): a wave owns two
// image rows of 16 pixels, the 18 weight fragments (9 taps x 2 channel chunks, 18 KiB for the whole layer) stay in
// registers, LDS holds only the 18x10 pixel patch (conflict-free swizzled rows as in resblock.hip), and of the four lane
// groups only group 0 - whose accumulator rows are channels 0-3 - runs the epilogue: 16 lanes = 16 consecutive pixels of
// one channel plane = one 64-byte store.
#include "common.h"

namespace {

constexpr int kRow = 64;                   // bytes of one LDS row: 32 channels of one pixel
constexpr int kTW = 16, kTH = 8;           // output tile
constexpr int kPW = kTW + 2, kPH = kTH + 2;
constexpr int kPitch = 24;                 // patch pitch in rows (a multiple of 8: see lds_off)
constexpr int kImg = kPH * kPitch * kRow;  // one channel chunk of the patch
constexpr int kPieces = 2 * kPH * kPW * 4; // 16-byte pieces of the patch
constexpr int kNU = (kPieces + 255) / 256;

// 16-byte piece `piece` of row `row`: unpadded 64-byte rows, piece index XOR-ed with 2 * (bit 2 of the row) - a ds_read_b128
// of 16 lanes on 16 consecutive rows is then conflict-free at any base row (tools/lds_layout.py)
__device__ __forceinline__ int lds_off(int row, int piece) { return row * kRow + ((piece ^ ((row >> 1) & 2)) << 4); }

struct RgbK {
  const char* in;
  const char* w;
  const float* bias;
  float* out;
  long long n_stride;
  int N, H, W, tiles_x, tiles_y, c_real, act;
};

template <typename T>
__global__ __launch_bounds__(256) void conv_rgb_kernel(const RgbK p) {
  __shared__ __attribute__((aligned(16))) char lds[2 * kImg];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int idx = lane & 15, g = lane >> 4;
  int bx = blockIdx.x;
  const int txb = bx % p.tiles_x;
  bx /= p.tiles_x;
  const int tyb = bx % p.tiles_y;
  const int n = bx / p.tiles_y;
  const int y0 = tyb * kTH, x0 = txb * kTW;
  const char* in_n = p.in + (size_t)n * p.H * p.W * 128;

  // patch: unconditional loads from a clamped address, zeroed afterwards (a load under a divergent `if` serialises)
  u32x4 va[kNU];
  int da[kNU];
  bool ok[kNU];
#pragma unroll
  for (int u = 0; u < kNU; ++u) {
    const int i = min(tid + u * 256, kPieces - 1);
    const int s = i & 3, r = i >> 2;
    const int cc = r >= kPH * kPW ? 1 : 0, prow = r - cc * kPH * kPW;
    const int py = (prow * 57) >> 10, px = prow - py * kPW;  // prow / 18, exact for prow < 180
    const int iy = y0 - 1 + py, ix = x0 - 1 + px;
    da[u] = (tid + u * 256 < kPieces) ? cc * kImg + lds_off(py * kPitch + px, s) : -1;
    ok[u] = iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
    const int cy = min(max(iy, 0), p.H - 1), cx = min(max(ix, 0), p.W - 1);
    va[u] = *reinterpret_cast<const u32x4*>(in_n + ((size_t)cy * p.W + cx) * 128 + cc * 64 + s * 16);
  }
  // A-fragments: packed image [tap][chunk][32 rows][64 B]; lane (idx, g) feeds bytes 16g..16g+15 of row idx (row tile 0,
  // whose rows 0-2 are output channels 0-2: row_to_channel of common.h)
  bf16x8 wfr[18];
#pragma unroll
  for (int k = 0; k < 18; ++k)
    wfr[k] = *reinterpret_cast<const bf16x8*>(p.w + (size_t)(k >> 1) * 4096 + (k & 1) * 2048 + idx * 64 + g * 16);
  float bias[4] = {0.f, 0.f, 0.f, 0.f};
  if (g == 0) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (e < p.c_real) bias[e] = p.bias[e];
  }
#pragma unroll
  for (int u = 0; u < kNU; ++u)
    if (da[u] >= 0) *reinterpret_cast<u32x4*>(lds + da[u]) = ok[u] ? va[u] : u32x4{0u, 0u, 0u, 0u};
  __syncthreads();

  // wave `wid` owns tile rows 2*wid and 2*wid + 1: per tap and chunk one fragment read and one MFMA per row
  f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
  for (int tt = 0; tt < 9; ++tt) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int row = (2 * wid + r + tt / 3) * kPitch + idx + tt % 3;
        const bf16x8 xf = *reinterpret_cast<const bf16x8*>(lds + c * kImg + lds_off(row, g));
        acc[r] = Mma16<T>::run(wfr[tt * 2 + c], xf, acc[r]);
      }
    }
  }
  if (g == 0) {  // accumulator rows 0-3 = output channels 0-3 of pixel idx
    const size_t hw = (size_t)p.H * p.W;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int y = y0 + 2 * wid + r, x = x0 + idx;
      if (y < p.H && x < p.W) {
        float* o = p.out + (size_t)n * p.n_stride + (size_t)y * p.W + x;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (e < p.c_real) {
            float v = acc[r][e] + bias[e];
            if (p.act == TG_ACT_SIGMOID) v = 1.f / (1.f + __expf(-v));
            o[e * hw] = v;
          }
        }
      }
    }
  }
}

}  // namespace

extern "C" int tg_conv3x3_rgb(int dtype, const void* in, const void* w_packed, const float* bias, float* out,
                              long long out_n_stride, int c_real, int N, int H, int W, int Cin, int act, void* stream) {
  if (!in || !w_packed || !bias || !out || N <= 0 || H <= 0 || W <= 0 || c_real <= 0 || c_real > 4) return TG_E_BADARG;
  if (out_n_stride < (long long)c_real * H * W) return TG_E_BADARG;
  if (act != TG_ACT_NONE && act != TG_ACT_SIGMOID) return TG_E_UNSUPPORTED;
  if ((dtype != TG_BF16 && dtype != TG_F16) || Cin != 64) return TG_E_UNSUPPORTED;  // else: tg_conv with TG_OUT_NCHW_F32
  if (!tg_aligned16(in) || !tg_aligned16(w_packed)) return TG_E_ALIGN;
  RgbK k;
  k.in = (const char*)in; k.w = (const char*)w_packed; k.bias = bias; k.out = out; k.n_stride = out_n_stride;
  k.N = N; k.H = H; k.W = W; k.c_real = c_real; k.act = act;
  k.tiles_x = (W + kTW - 1) / kTW; k.tiles_y = (H + kTH - 1) / kTH;
  const long long blocks = (long long)k.tiles_x * k.tiles_y * N;
  if (blocks > 0x7fffffffLL) return TG_E_UNSUPPORTED;
  if (dtype == TG_F16) hipLaunchKernelGGL(conv_rgb_kernel<F16>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, k);
  else hipLaunchKernelGGL(conv_rgb_kernel<BF16>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, k);
  return tg_launch_status();
}
