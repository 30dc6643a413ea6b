// Input-gradient of the 4x4 stride-2 padding-1 convolutions (code/models.py:90-94, autograd of code/train.py:304-307,339-342) with
// CLASS-SPECIALISED waves - csrc/convt_cw.hip's structure for the other sub-pixel pattern (round 5):
//
//   din[n, 2y+oy, 2x+ox, ci] = act'(mask) * sum over the 4 taps (wy, wx, slot) of class (oy, ox), co:
//                                                   W'[slot][ci][co] * dout[n, y + wy - 1, x + wx - 1, co]
//     class 0 = (0,0): (1,1,5) (1,0,7) (0,1,13) (0,0,15)      class 1 = (0,1): (1,2,4) (1,1,6) (0,2,12) (0,1,14)
//     class 2 = (1,0): (2,1,1) (2,0,3) (1,1,9)  (1,0,11)      class 3 = (1,1): (2,2,0) (2,1,2) (1,2,8)  (1,1,10)
//   (the 16 (class, tap) pairs of csrc/convt_mfma.hip's PAT 1; W' = the role-swapped 16-slot packing)
//
// The sub-pixel launch of convt_mfma.hip re-stages the 16 slots' weights per 8 x 16 tile and runs thousands of workgroups: 75 us for
// the discriminator's first down-sampling layer (12 x 64 x 64 -> 128 x 128, 64 -> 64 channels: 31 MB of tensor traffic) in each
// half of the step.  Here the workgroups are persistent (min(tiles, cap / (Cin/64)) x Cin/64 over 4 x 16 tiles of dout) and each of
// the eight waves owns ONE class x 32 of the workgroup's 64 output channels: its 4 slots x Cout as A-fragments in registers (64 / 128
// VGPRs for 64 / 128 reduction channels), its k-loop over the LDS patch, its mask rows, and the epilogue straight from the
// accumulators (act'(mask), pack, 16-byte stores written through the L2).  Every class has 4 taps: the SIMDs are balanced.  The
// (4+2) x (16+2) patch is 9 one-KiB blocks per 32-channel chunk: wave w brings row block w of every chunk, waves 0 .. NCH-1 the ninth
// block of chunk w as well.  One barrier per tile; no accumulator image.
#ifndef TG_ST_AUX
#define TG_ST_AUX "sc1"   // results are written THROUGH the L2 (common.h, tg_store16; profiles/r05_u_write_through_ab.log)
#endif
#include "rbw_common.h"
#include <atomic>
#include <type_traits>

// out-of-image patch positions (and the pitch padding) are DMA'd from here; launches without a mask read their "mask rows" here
__device__ __attribute__((aligned(16))) unsigned int tg_c4d_zero_page[4];

namespace {

constexpr int kRow = 64, kPitch = 24;
constexpr int kTH = 4;                                   // dout rows per tile (x 16 columns): 64 dout = 256 din pixels
constexpr int kBlocks = ((kTH + 2) * kPitch) / 16;       // 1-KiB blocks of a chunk image (6 patch rows x pitch 24 = 144 image rows): 9
constexpr int kChunkBytes = kBlocks * 1024;
static_assert(kBlocks == 9, "waves 0-7 bring row blocks 0-7 of every chunk, waves 0 .. NCH-1 block 8 of chunk w");

__device__ __forceinline__ int swz(int row, int piece) { return row * kRow + ((piece ^ ((row >> 1) & 2)) << 4); }

// taps of a class: weight slot, window position (row, column) in the 3 x 3 window whose origin is (y - 1, x - 1)
template <int CLS> struct Taps;
template <> struct Taps<0> { static constexpr int slot[4] = {5, 7, 13, 15}, wy[4] = {1, 1, 0, 0}, wx[4] = {1, 0, 1, 0}; };
template <> struct Taps<1> { static constexpr int slot[4] = {4, 6, 12, 14}, wy[4] = {1, 1, 0, 0}, wx[4] = {2, 1, 2, 1}; };
template <> struct Taps<2> { static constexpr int slot[4] = {1, 3, 9, 11}, wy[4] = {2, 2, 1, 1}, wx[4] = {1, 0, 1, 0}; };
template <> struct Taps<3> { static constexpr int slot[4] = {0, 2, 8, 10}, wy[4] = {2, 2, 1, 1}, wx[4] = {2, 1, 2, 1}; };

template <typename T> __device__ __forceinline__ f32x4 mma(bf16x8 a, bf16x8 b, f32x4 c) { return Mma16<T>::run(a, b, c); }

struct C4dK {
  const char* in;     // dout [N][H][W][NCH * 32]
  const char* w;
  const char* zero;
  const char* mask;   // [N][2H][2W][Cout] or null
  char* out;          // din  [N][2H][2W][Cout]
  int H, W, Cout, tiles_x, tiles_y, ntiles, mask_mode;
};

// (arguments one by one: the first 16 dwords are preloaded into SGPRs with the wave - csrc/build.sh, -amdgpu-kernarg-preload-count)
template <int NCH, typename T>
__global__ __launch_bounds__(512) void conv4s2d_cw_kernel(const char* a_in, const char* a_w, const char* a_zero, int a_H, int a_W, int a_Cout,
                                                          int a_tiles_x, int a_tiles_y, int a_ntiles, int a_mask_mode, char* a_out,
                                                          const char* a_mask) {
  C4dK p;
  p.in = a_in; p.w = a_w; p.zero = a_zero; p.H = a_H; p.W = a_W; p.Cout = a_Cout; p.tiles_x = a_tiles_x; p.tiles_y = a_tiles_y;
  p.ntiles = a_ntiles; p.mask_mode = a_mask_mode; p.out = a_out; p.mask = a_mask;
  constexpr int kBufBytes = NCH * kChunkBytes;           // 18 KB / 36 KB
  constexpr int kDepth = NCH == 4 ? 2 : 3;               // fragment sets in flight (128 reduction channels: 128 VGPRs of weights)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int idx = lane & 15, g = lane >> 4;
  const int wc = wid & 1;                                     // channel half: packed rows 32 wc .. + 31
  // waves w and w + 4 share a SIMD; every class has 4 taps
  const int cls = ((wid >> 1) & 1) ? ((wid >> 2) ? 2 : 1) : ((wid >> 2) ? 0 : 3);
  const int co_base = blockIdx.y * 64;
  const int ntl = (p.ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  const size_t pix_bytes = (size_t)NCH * 64;

  // ---- LDS-DMA: row block rb = 16 image rows of a chunk; the lane's 16 bytes: row 16 rb + lane / 4, physical piece lane % 4 = logical
  // piece ^ swizzle.  Patch position (row, column) of image row r: (r / 24, r % 24), origin (ty0 - 1, tx0 - 1); columns 18 .. 23 are
  // pitch padding (a row outside every image: the zero page).  rb = wid for every chunk; rb = 8 of chunk wid for waves 0 .. NCH - 1
  int dpy[2], dpx[2], dof[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int row = (e ? 8 : wid) * 16 + (lane >> 2);
    const int py = row / kPitch, px = row - py * kPitch;
    dpy[e] = px < 18 ? py - 1 : -(1 << 20);
    dpx[e] = px - 1;
    dof[e] = ((lane & 3) ^ ((row >> 1) & 2)) * 16;
  }
  struct Tile { int n, ty0, tx0; };
  auto tile_of = [&](int t) {
    Tile r;
    const int txb = t % p.tiles_x;
    t /= p.tiles_x;
    const int tyb = t % p.tiles_y;
    r.n = t / p.tiles_y;
    r.ty0 = tyb * kTH;
    r.tx0 = txb * 16;
    return r;
  };
  auto dma_patch = [&](const Tile& tl, int buf) {   // asynchronous: vmcnt + barrier before anyone reads it
    const char* in_n = p.in + (size_t)tl.n * p.H * p.W * pix_bytes;
    {
      const int iy = tl.ty0 + dpy[0], ix = tl.tx0 + dpx[0];
      const bool ok = ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
      const char* src = in_n + (unsigned)((iy * p.W + ix) * (int)pix_bytes + dof[0]);
#pragma unroll
      for (int u = 0; u < NCH; ++u) glds16(ok ? src + u * 64 : p.zero, lds0 + buf * kBufBytes + u * kChunkBytes + wid * 1024);
    }
    if (wid < NCH) {   // wave-uniform: the ninth row block of chunk wid
      const int iy = tl.ty0 + dpy[1], ix = tl.tx0 + dpx[1];
      const bool ok = ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
      const char* src = in_n + (unsigned)((iy * p.W + ix) * (int)pix_bytes + dof[1]) + wid * 64;
      glds16(ok ? src : p.zero, lds0 + buf * kBufBytes + wid * kChunkBytes + 8 * 1024);
    }
  };
  int tile = (int)blockIdx.x;
  Tile cur = tile_of(tile);
  dma_patch(cur, 0);

  // fragment addresses of tile row b, window column wx inside one chunk image; window rows add multiples of the pitch
  int xb[4][3];
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int d = 0; d < 3; ++d) xb[b][d] = swz(b * kPitch + idx + d, g);
  // the lane's 8 output channels: co_base + 32 wc + 8 g .. + 7 (two row-interleaved MFMA tiles, common.h)
  const int ch0 = co_base + wc * 32 + 8 * g;
  const float neg = p.mask_mode == TG_MASK_LRELU ? 0.2f : 0.f;   // act'(mask) where the mask is not positive

  auto run = [&](auto CLS) {
    using TP = Taps<decltype(CLS)::value>;
    constexpr int NT = 4, NS = NCH * NT;   // k-steps per tile: chunk-major, the class's taps inside
    constexpr int oy = decltype(CLS)::value >> 1, ox = decltype(CLS)::value & 1;
    // A-fragments of packed rows 32 wc + 16 a + idx, in k-loop order.  Packed image [slot][chunk][Cout rows][64 B]
    bf16x8 wfr[NCH][NT][2];
    const char* const wl = p.w + ((size_t)co_base + wc * 32 + idx) * 64 + g * 16;
#pragma unroll
    for (int ci = 0; ci < NCH; ++ci)
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int a = 0; a < 2; ++a)
          wfr[ci][t][a] = *reinterpret_cast<const bf16x8*>(wl + ((size_t)(TP::slot[t] * NCH + ci) * p.Cout + a * 16) * 64);
    // the patch of the first tile is older than the weight fragments: it has landed when at most those are in flight
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NS * 2) : "memory");
    for (int i = 0; i < ntl; ++i) {
      lds_barrier();   // tile i's patch is in LDS (every wave waited for its own blocks), buffer (i + 1) & 1 is nobody's any more
      const Tile mine = cur;
      if (i + 1 < ntl) {
        tile += (int)gridDim.x;
        cur = tile_of(tile);
        dma_patch(cur, (i + 1) & 1);
      }
      // this tile's mask rows: output pixel (2 (ty0 + b) + oy, 2 (tx0 + idx) + ox), the lane's 8 channels.  UNCONDITIONAL loads (a launch
      // without a mask reads the zero page) behind the patch requests in the in-order counter: when they have arrived, the patch has
      u32x4 mk[4];
      {
        const char* m_n = p.mask + (size_t)mine.n * 4 * p.H * p.W * p.Cout * 2;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const int my = min(2 * (mine.ty0 + b) + oy, 2 * p.H - 1), mx = min(2 * (mine.tx0 + idx) + ox, 2 * p.W - 1);   // clamped: unused outside
          const char* a = p.mask ? m_n + (unsigned)(((my * 2 * p.W + mx) * p.Cout + ch0) * 2) : p.zero;
          mk[b] = *reinterpret_cast<const u32x4*>(a);
        }
      }
      const char* img = smem + (i & 1) * kBufBytes;
      f32x4 acc[2][4];
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
      // k-loop: NS steps of 4 B-fragment reads + 8 MFMAs
      bf16x8 xf[kDepth][4];
      auto frags = [&](int s_, int buf) {   // compile-time arguments after unrolling
        const int ci = s_ / NT, t = s_ % NT;
#pragma unroll
        for (int b = 0; b < 4; ++b)
          xf[buf][b] = *reinterpret_cast<const bf16x8*>(img + ci * kChunkBytes + xb[b][TP::wx[t]] + TP::wy[t] * kPitch * kRow);
      };
#pragma unroll
      for (int s_ = 0; s_ < kDepth - 1; ++s_) frags(s_, s_);
#pragma unroll
      for (int s_ = 0; s_ < NS; ++s_) {
        if (s_ + kDepth - 1 < NS) frags(s_ + kDepth - 1, (s_ + kDepth - 1) % kDepth);
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
          for (int a = 0; a < 2; ++a) acc[a][b] = mma<T>(wfr[s_ / NT][s_ % NT][a], xf[s_ % kDepth][b], acc[a][b]);
      }
      // epilogue from the accumulators.  The rows are pinned in registers BEFORE the first store goes out (the compiler's wait for a later
      // row would otherwise also wait for the earlier rows' stores: the counter is in order, the stores are inline asm)
#pragma unroll
      for (int b = 0; b < 4; ++b) asm volatile("" : "+v"(mk[b]));
      char* const out_t = p.out + ((((size_t)mine.n * 2 * p.H + 2 * mine.ty0 + oy) * 2 * p.W + 2 * mine.tx0 + ox) * p.Cout + ch0) * 2;
      const bool okx = mine.tx0 + idx < p.W;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = acc[0][b][e];
          v[4 + e] = acc[1][b][e];
        }
        if (p.mask_mode != TG_MASK_NONE) {
          // mask value > 0 on its 16-bit pattern (sign clear, not zero): the low half as the sign of word << 16, the high half as word > 0xffff
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int w_ = (int)mk[b][e];
            v[2 * e] = (int)((unsigned)w_ << 16) > 0 ? v[2 * e] : neg * v[2 * e];
            v[2 * e + 1] = w_ > 0xffff ? v[2 * e + 1] : neg * v[2 * e + 1];
          }
        }
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = pack2<T>(v[2 * e], v[2 * e + 1]);
        if (okx && mine.ty0 + b < p.H) tg_store16(out_t + (unsigned)((b * 4 * p.W + idx * 2) * p.Cout) * 2u, o);
      }
    }
  };
  switch (cls) {
    case 0: run(std::integral_constant<int, 0>{}); break;
    case 1: run(std::integral_constant<int, 1>{}); break;
    case 2: run(std::integral_constant<int, 2>{}); break;
    default: run(std::integral_constant<int, 3>{}); break;
  }
}

template <int NCH, typename T>
int launch_c4d(const C4dK& k, dim3 grid, hipStream_t st) {
  constexpr int lds = 2 * NCH * kChunkBytes;
  auto fn = conv4s2d_cw_kernel<NCH, T>;
  static std::atomic<bool> attr_done{false};  // one-time function attribute (benign race: idempotent)
  if (!attr_done) {
    TG_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    attr_done = true;
  }
  hipLaunchKernelGGL(fn, grid, dim3(512), lds, st, k.in, k.w, k.zero, k.H, k.W, k.Cout, k.tiles_x, k.tiles_y, k.ntiles, k.mask_mode, k.out, k.mask);
  return tg_launch_status();
}

}  // namespace

extern "C" int tg_conv4s2_dgrad_cw(int dtype, const void* dout, const void* w_dgrad_packed, void* din, int N, int OH, int OW, int Cout,
                                   int Cin, const void* mask, int mask_mode, int max_workgroups, void* stream) {
  // dout [N][OH][OW][Cout] -> din [N][2OH][2OW][Cin]; the kernel's reduction channels are Cout, its output channels Cin
  if (!dout || !w_dgrad_packed || !din || N <= 0 || OH <= 0 || OW <= 0) return TG_E_BADARG;
  if ((dtype != TG_BF16 && dtype != TG_F16) || (Cout != 64 && Cout != 128) || Cin <= 0 || Cin % 64) return TG_E_UNSUPPORTED;
  if (mask && (mask_mode < TG_MASK_NONE || mask_mode > TG_MASK_LRELU)) return TG_E_BADARG;
  if (!tg_aligned16(dout) || !tg_aligned16(w_dgrad_packed) || !tg_aligned16(din) || (mask && !tg_aligned16(mask))) return TG_E_ALIGN;
  if ((long long)OH * OW * Cout * 2 >= 0x7fffffffLL || (long long)4 * OH * OW * Cin * 2 >= 0x7fffffffLL) return TG_E_UNSUPPORTED;   // 32-bit offsets
  static const char* zero_page = [] {
    void* z = nullptr;
    return hipGetSymbolAddress(&z, HIP_SYMBOL(tg_c4d_zero_page)) == hipSuccess ? (const char*)z : (const char*)nullptr;
  }();
  if (!zero_page) return TG_E_BADARG;
  C4dK k;
  k.in = (const char*)dout; k.w = (const char*)w_dgrad_packed; k.zero = zero_page; k.mask = (const char*)mask; k.out = (char*)din;
  k.H = OH; k.W = OW; k.Cout = Cin; k.mask_mode = mask ? mask_mode : TG_MASK_NONE;
  k.tiles_x = (OW + 15) / 16; k.tiles_y = (OH + kTH - 1) / kTH;
  const long long nt = (long long)k.tiles_x * k.tiles_y * N;
  if (nt > 0x3fffffffLL) return TG_E_UNSUPPORTED;
  k.ntiles = (int)nt;
  // persistent grid: the cap's workgroups shared by the Cin/64 channel tiles, pixel tiles dealt evenly
  const int co_tiles = Cin / 64;
  const int cap = max_workgroups > 0 ? max_workgroups : 256;
  const int per = cap / co_tiles > 0 ? cap / co_tiles : 1;
  const int rounds = (k.ntiles + per - 1) / per;
  const int gx = (k.ntiles + rounds - 1) / rounds;
  dim3 grid((unsigned)gx, (unsigned)co_tiles);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == TG_F16) return Cout == 64 ? launch_c4d<2, F16>(k, grid, st) : launch_c4d<4, F16>(k, grid, st);
  return Cout == 64 ? launch_c4d<2, BF16>(k, grid, st) : launch_c4d<4, BF16>(k, grid, st);
}
