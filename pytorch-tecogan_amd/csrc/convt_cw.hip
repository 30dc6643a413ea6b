// Conv-transpose k3 s2 p1 op1 forward (code/ops.py:45-54 conv2_tran; code/models.py:72,74) with CLASS-SPECIALISED waves (round 5).
//
//   out[n, 2y+oy, 2x+ox, co] = act( bias[co] + sum over the taps (dy, dx, slot) of class (oy, ox), ci:  W[slot][co][ci] * in[n, y+dy, x+dx, ci] )
//     class 0 = (0,0): (0,0,4)            class 1 = (0,1): (0,1,3) (0,0,5)
//     class 2 = (1,0): (1,0,1) (0,0,7)    class 3 = (1,1): (1,1,0) (1,0,2) (0,1,6) (0,0,8)
//
// The layer writes FOUR times the pixels a 3x3 convolution of the same matrix work writes.  The sub-pixel launch of convt_mfma.hip
// re-stages the 9 slots' weights per tile (147 KB per 128 pixels at 128 -> 128: the CU's intake bounds it - 40 us at 256 x 256);
// the register-weights kernel's sub-pixel form (conv3_rw.hip, SP) keeps the weights but needs an accumulator image, a barrier and a
// producer epilogue per CLASS (8800 ticks per 64-pixel tile, tools/stamp_convt_rw.py: 31 us).  Here the workgroup is persistent
// (min(tiles, cap / (Cout/64)) x Cout/64 workgroups walk 4 x 16 INPUT tiles) and each of its eight waves owns ONE class of one half
// of the 64 output channels for the workgroup's lifetime:
//   * wave (wc, class): the class's 1 / 2 / 2 / 4 slots x Cin x 32 channels as A-fragments in registers (at most 128 VGPRs), its
//     k-loop over the tile's LDS patch (v_mfma_f32_16x16x32, full K: no split, no exchange), then the epilogue straight from the
//     accumulators - + bias, activation, 16-bit pack, one 16-byte store per lane and tile row (the lane's 8 channels of output pixel
//     (2y+oy, 2x+ox); the other channel half's wave completes the 128-byte line).  No accumulator image, no producer role;
//   * the two waves of a SIMD are classes 3 + 0 (5 taps) or 1 + 2 (4 taps): the epilogue arithmetic and the stores of one overlap
//     the matrix work of the other;
//   * the (4+1) x (16+1) patch is brought by LDS-DMA (all eight waves, 2 - 4 one-KiB blocks each; positions outside the image read
//     a page of zeros) into a double buffer one tile ahead; ONE barrier per tile.
// LDS image (64-byte rows per 32-channel chunk, pitch 24, piece XOR) and packed weights are those of conv3_rw.hip / conv_mfma.hip.
#ifndef TG_ST_AUX
#define TG_ST_AUX "sc1"   // this kernel's results are written THROUGH the L2 (common.h, tg_store16; profiles/r05_u_write_through_ab.log)
#endif
#include "rbw_common.h"
#include <atomic>
#include <type_traits>

#ifdef TG_STAMP
// diagnostic build (tools/stamp_convt_cw.py): waves 0 (class 3) and 4 (class 0) of workgroup 0: [role][0..3 prologue | 4 + 4 * tile + phase]
__device__ long long tg_cw_stamps[2 * 32];
#define CW_STAMP(i)                                                                                  \
  do {                                                                                               \
    if (blockIdx.x == 0 && blockIdx.y == 0 && (threadIdx.x & 255) == 0 && (i) < 32)                   \
      tg_cw_stamps[(threadIdx.x >> 8) * 32 + (i)] = (long long)__builtin_amdgcn_s_memtime();         \
  } while (0)
extern "C" int tg_debug_read_cw_stamps(long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(tg_cw_stamps), sizeof(long long) * n);
}
#else
#define CW_STAMP(i) do {} while (0)
#endif

// out-of-image patch positions (and the pitch padding) are DMA'd from here
__device__ __attribute__((aligned(16))) unsigned int tg_cw_zero_page[4];

namespace {

constexpr int kRow = 64, kPitch = 24;
constexpr int kTH = 4;                                   // input rows per tile (x 16 columns): 64 input = 256 output pixels
constexpr int kUsedBlocks = ((kTH + 1) * kPitch + 15) / 16;   // 1-KiB blocks of a chunk image that hold patch rows 0 .. kTH: 8
constexpr int kChunkBytes = kUsedBlocks * 1024;          // one 32-channel chunk of the patch
template <int NCH> struct CwGeo {
  static constexpr int kBufBytes = NCH * kChunkBytes;    // 16 KB / 32 KB
  static constexpr int NBW = NCH;                        // DMA instructions per wave and tile: row block `wave` of every chunk
  static constexpr int kLds = 2 * kBufBytes;
};

__device__ __forceinline__ int swz(int row, int piece) { return row * kRow + ((piece ^ ((row >> 1) & 2)) << 4); }

// taps of a class: weight slot, window position
template <int CLS> struct Taps;
template <> struct Taps<0> { static constexpr int N = 1; static constexpr int slot[4] = {4, 0, 0, 0}, dy[4] = {0, 0, 0, 0}, dx[4] = {0, 0, 0, 0}; };
template <> struct Taps<1> { static constexpr int N = 2; static constexpr int slot[4] = {3, 5, 0, 0}, dy[4] = {0, 0, 0, 0}, dx[4] = {1, 0, 0, 0}; };
template <> struct Taps<2> { static constexpr int N = 2; static constexpr int slot[4] = {1, 7, 0, 0}, dy[4] = {1, 0, 0, 0}, dx[4] = {0, 0, 0, 0}; };
template <> struct Taps<3> { static constexpr int N = 4; static constexpr int slot[4] = {0, 2, 6, 8}, dy[4] = {1, 1, 0, 0}, dx[4] = {1, 0, 1, 0}; };

template <typename T> __device__ __forceinline__ f32x4 mma(bf16x8 a, bf16x8 b, f32x4 c) { return Mma16<T>::run(a, b, c); }

struct CwK {
  const char* in;
  const char* w;
  const char* zero;
  const float* bias;
  char* out;
  unsigned char* bits;   // null, or [N][2H][2W][Cout / 8]: bit c % 8 of byte c / 8 = (stored value of channel c > 0) - the 1-bit ReLU mask
  int H, W, Cout, tiles_x, tiles_y, ntiles, act;
};

// (arguments one by one: the first 16 dwords are preloaded into SGPRs with the wave - csrc/build.sh, -amdgpu-kernarg-preload-count)
template <int NCH, typename T>
__global__ __launch_bounds__(512) void convt_cw_kernel(const char* a_in, const char* a_w, const char* a_zero, int a_H, int a_W, int a_Cout,
                                                       int a_tiles_x, int a_tiles_y, int a_ntiles, int a_act, char* a_out,
                                                       const float* a_bias, unsigned char* a_bits) {
  using G = CwGeo<NCH>;
  CwK p;
  p.bits = a_bits;
  p.in = a_in; p.w = a_w; p.zero = a_zero; p.H = a_H; p.W = a_W; p.Cout = a_Cout; p.tiles_x = a_tiles_x; p.tiles_y = a_tiles_y;
  p.ntiles = a_ntiles; p.act = a_act; p.out = a_out; p.bias = a_bias;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int idx = lane & 15, g = lane >> 4;
  const int wc = wid & 1;                                     // channel half: packed rows 32 wc .. + 31
  // waves w and w + 4 share a SIMD: SIMDs 0 / 1 run classes 3 (w < 4) and 0, SIMDs 2 / 3 classes 1 and 2
#ifndef CW_PAIRING
#define CW_PAIRING 0   // A/B (profiles/r05_r_convt_cw_ab.log, section 5): 1 = classes 3 + 1 / 2 + 0 per SIMD pair, 2 = 3 + 2 / 1 + 0
#endif
  const int cls = CW_PAIRING == 0 ? (((wid >> 1) & 1) ? ((wid >> 2) ? 2 : 1) : ((wid >> 2) ? 0 : 3))
                  : CW_PAIRING == 1 ? (((wid >> 1) & 1) ? ((wid >> 2) ? 0 : 2) : ((wid >> 2) ? 1 : 3))
                                    : (((wid >> 1) & 1) ? ((wid >> 2) ? 0 : 1) : ((wid >> 2) ? 2 : 3));
  const int co_base = blockIdx.y * 64;
  const int ntl = (p.ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  const size_t pix_bytes = (size_t)NCH * 64;
  CW_STAMP(0);

  // ---- LDS-DMA: block j = wid + 8 u of a patch buffer = rows 16 wid .. + 15 of chunk u (a chunk image is exactly 8 blocks); the lane's
  // 16 bytes: row 16 wid + lane / 4, physical piece lane % 4 = logical piece ^ swizzle.  The same for every tile and chunk: patch row /
  // column (the patch's origin is the tile's; positions nobody reads get a row outside every image), byte offset inside the pixel
  static_assert(kUsedBlocks == 8, "block j = wid + 8 u is row block wid of chunk u");
  const int drow = wid * 16 + (lane >> 2);
  const int dpx = drow % kPitch;
  const int dpy = (drow / kPitch <= kTH && dpx <= 16) ? drow / kPitch : -(1 << 20);
  const int dof = ((lane & 3) ^ ((drow >> 1) & 2)) * 16;
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
  auto dma_src = [&](const Tile& tl) -> const char* {   // the lane's 16 bytes of chunk 0 of tile tl's patch (the zero page outside the image)
    const int iy = tl.ty0 + dpy, ix = tl.tx0 + dpx;
    const bool ok = ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
    return ok ? p.in + (size_t)tl.n * p.H * p.W * pix_bytes + (unsigned)((iy * p.W + ix) * (int)pix_bytes + dof) : nullptr;
  };
  auto dma_chunk = [&](const char* src, int buf, int u) {   // asynchronous: vmcnt + barrier before anyone reads it
    glds16(src ? src + u * 64 : p.zero, lds0 + buf * G::kBufBytes + wid * 1024 + u * 8192);
  };
  int tile = (int)blockIdx.x;
  Tile cur = tile_of(tile);   // the tile this wave multiplies next
  {
    const char* s0 = dma_src(cur);
#pragma unroll
    for (int u = 0; u < NCH; ++u) dma_chunk(s0, 0, u);
  }
  CW_STAMP(1);

  // fragment addresses of tile row b, window column dx inside one chunk image; window rows add multiples of the pitch
  int xb[4][2];
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int d = 0; d < 2; ++d) xb[b][d] = swz(b * kPitch + idx + d, g);
  // the lane's 8 output channels: co_base + 32 wc + 8 g .. + 7 (two row-interleaved MFMA tiles, common.h)
  const int ch0 = co_base + wc * 32 + 8 * g;
  float bias_r[8];
#pragma unroll
  for (int e = 0; e < 8; e += 4) {
    f32x4 t = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) t = *reinterpret_cast<const f32x4*>(p.bias + ch0 + e);
    bias_r[e] = t[0]; bias_r[e + 1] = t[1]; bias_r[e + 2] = t[2]; bias_r[e + 3] = t[3];
  }
  const float act_slope = p.act == TG_ACT_RELU ? 0.f : p.act == TG_ACT_LRELU ? 0.2f : 1.f;   // activation as max(v, slope * v)
  const bool full = p.H % kTH == 0 && p.W % 16 == 0;   // every tile whole: every wave issues exactly kTH stores per tile

  auto run = [&](auto CLS) {
    using TP = Taps<decltype(CLS)::value>;
    constexpr int NT = TP::N, NS = NCH * NT;   // k-steps per tile: chunk-major, the class's taps inside
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
    CW_STAMP(2);
    // the patch of the first tile is older than the weight fragments: it has landed when at most those are in flight
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NS * 2) : "memory");
    for (int i = 0; i < ntl; ++i) {
      if (i < 6) CW_STAMP(4 + 4 * i + 0);
      lds_barrier();   // tile i's patch is in LDS (every wave waited for its own blocks), buffer (i + 1) & 1 is nobody's any more
      if (i < 6) CW_STAMP(4 + 4 * i + 1);
      const Tile mine = cur;
      if (i + 1 < ntl) {   // (requested inside the k-loop and two tiles ahead, the extra pointer and tile state spilled: 25.7 -> 27.5 us)
        tile += (int)gridDim.x;
        cur = tile_of(tile);
        const char* s1 = dma_src(cur);
#pragma unroll
        for (int u = 0; u < NCH; ++u) dma_chunk(s1, (i + 1) & 1, u);
      }
      const char* img = smem + (i & 1) * G::kBufBytes;
      f32x4 acc[2][4];   // start from the bias
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{bias_r[4 * a], bias_r[4 * a + 1], bias_r[4 * a + 2], bias_r[4 * a + 3]};
      // k-loop: NS steps of 4 B-fragment reads + 8 MFMAs, the fragments of steps s + 1 and s + 2 in flight while step s multiplies
#ifndef CW_DEPTH
#define CW_DEPTH 3   // fragment sets in flight (A/B: 2)
#endif
      bf16x8 xf[CW_DEPTH][4];
      auto frags = [&](int s_, int buf) {   // compile-time arguments after unrolling
        const int ci = s_ / NT, t = s_ % NT;
#pragma unroll
        for (int b = 0; b < 4; ++b)
          xf[buf][b] = *reinterpret_cast<const bf16x8*>(img + ci * kChunkBytes + xb[b][TP::dx[t]] + TP::dy[t] * kPitch * kRow);
      };
      frags(0, 0);
      if (NS > 1 && CW_DEPTH > 2) frags(1, 1);
#pragma unroll
      for (int s_ = 0; s_ < NS; ++s_) {
        if (s_ + CW_DEPTH - 1 < NS) frags(s_ + CW_DEPTH - 1, (s_ + CW_DEPTH - 1) % CW_DEPTH);
#ifdef CW_SCHED_BARRIER   // (A/B: fencing every k-step as conv3_rw does costs 1-5 % here - two waves per SIMD fill each other's gaps)
        __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
          for (int a = 0; a < 2; ++a) acc[a][b] = mma<T>(wfr[s_ / NT][s_ % NT][a], xf[s_ % CW_DEPTH][b], acc[a][b]);
#ifdef CW_SCHED_BARRIER   // (A/B: fencing every k-step as conv3_rw does costs 1-5 % here - two waves per SIMD fill each other's gaps)
        __builtin_amdgcn_sched_barrier(0);
#endif
      }
      if (i < 6) CW_STAMP(4 + 4 * i + 2);
      // epilogue from the accumulators: input pixel (ty0 + b, tx0 + idx) -> output pixel (2 y + oy, 2 x + ox), 8 channels = 16 bytes
      char* const out_t = p.out + ((((size_t)mine.n * 2 * p.H + 2 * mine.ty0 + oy) * 2 * p.W + 2 * mine.tx0 + ox) * p.Cout + ch0) * 2;
      const bool okx = mine.tx0 + idx < p.W;
      auto finish = [&](auto RELU, auto BITS) {
        // (BITS: + the 1-bit mask of the stored values - what the input-gradient of the layer ABOVE needs of this ReLU's output is its
        // sign pattern: 1 byte per lane instead of the 16 it would read back; LOG.md / DESIGN.md 'what a CU's vector-memory pipe takes')
        unsigned char* const bits_t = p.bits + ((((size_t)mine.n * 2 * p.H + 2 * mine.ty0 + oy) * 2 * p.W + 2 * mine.tx0 + ox) * p.Cout + ch0) / 8;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          u32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int a = e >> 1, k = (e & 1) * 2;
            float lo = acc[a][b][k], hi = acc[a][b][k + 1];
            if constexpr (decltype(RELU)::value) {
              lo = fmaxf(lo, 0.f);
              hi = fmaxf(hi, 0.f);
            } else {
              lo = fmaxf(lo, act_slope * lo);
              hi = fmaxf(hi, act_slope * hi);
            }
            o[e] = pack2<T>(lo, hi);
          }
          if (okx && mine.ty0 + b < p.H) {
            tg_store16(out_t + (unsigned)((b * 4 * p.W + idx * 2) * p.Cout) * 2u, o);
            if constexpr (decltype(BITS)::value) {
              unsigned m = 0;   // the test the masked input-gradients make on the 16-bit patterns: sign clear and not zero
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const int w_ = (int)o[e];
                m |= ((int)((unsigned)w_ << 16) > 0 ? 1u : 0u) << (2 * e);
                m |= (w_ > 0xffff ? 1u : 0u) << (2 * e + 1);
              }
              bits_t[(unsigned)((b * 4 * p.W + idx * 2) * p.Cout) / 8u] = (unsigned char)m;
            }
          }
        }
      };
#ifdef TG_EXPERIMENTS   // the 1-bit-mask form: bit-exact, and slower in the step (profiles/r06_k_mask_bits_ab.log); its mere presence costs
      const bool with_bits = p.bits != nullptr;   // the default kernel 0.016 ms per step (r06_n_mask_code_ab.log): experiments build only
#else
      constexpr bool with_bits = false;
#endif
      if (p.act == TG_ACT_RELU && with_bits) finish(std::true_type{}, std::true_type{});
      else if (p.act == TG_ACT_RELU) finish(std::true_type{}, std::false_type{});
      else finish(std::false_type{}, std::false_type{});
      // the next tile's patch must have landed before this wave arrives at the barrier.  It was requested before this tile's kTH stores
      // (the counter is in order; 2 kTH with the mask bytes): whole tiles - those may stay in flight; ragged images issue fewer stores: the plain wait
      if (full && !with_bits) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kTH) : "memory");
      else if (full && p.act == TG_ACT_RELU) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * kTH) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (i < 6) CW_STAMP(4 + 4 * i + 3);
    }
    CW_STAMP(28);
  };
  switch (cls) {
    case 0: run(std::integral_constant<int, 0>{}); break;
    case 1: run(std::integral_constant<int, 1>{}); break;
    case 2: run(std::integral_constant<int, 2>{}); break;
    default: run(std::integral_constant<int, 3>{}); break;
  }
}

template <int NCH, typename T>
int launch_cw(const CwK& k, dim3 grid, hipStream_t st) {
  constexpr int lds = CwGeo<NCH>::kLds;
  auto fn = convt_cw_kernel<NCH, T>;
  static std::atomic<bool> attr_done{false};  // one-time function attribute (benign race: idempotent)
  if (!attr_done) {
    TG_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    attr_done = true;
  }
  hipLaunchKernelGGL(fn, grid, dim3(512), lds, st, k.in, k.w, k.zero, k.H, k.W, k.Cout, k.tiles_x, k.tiles_y, k.ntiles, k.act, k.out, k.bias, k.bits);
  return tg_launch_status();
}

}  // namespace

extern "C" int tg_convt_fwd_cw(int dtype, const void* in, const void* w_packed, const float* bias, void* out, int N, int H, int W,
                               int Cin, int Cout, int act, void* relu_bits, int max_workgroups, void* stream) {
  if (!in || !w_packed || !out || N <= 0 || H <= 0 || W <= 0) return TG_E_BADARG;
  if (relu_bits && act != TG_ACT_RELU) return TG_E_BADARG;
  if (relu_bits && !kTgExperiments) return TG_E_UNSUPPORTED;   // (the 1-bit mask output is compiled into the experiments build only)
  if ((dtype != TG_BF16 && dtype != TG_F16) || (Cin != 64 && Cin != 128) || Cout <= 0 || Cout % 64) return TG_E_UNSUPPORTED;
  if (act != TG_ACT_NONE && act != TG_ACT_RELU && act != TG_ACT_LRELU) return TG_E_UNSUPPORTED;
  if (!tg_aligned16(in) || !tg_aligned16(w_packed) || !tg_aligned16(out) || (bias && !tg_aligned16(bias))) return TG_E_ALIGN;
  if ((long long)H * W * Cin * 2 >= 0x7fffffffLL || (long long)4 * 4 * W * Cout * 2 >= 0x7fffffffLL) return TG_E_UNSUPPORTED;   // 32-bit offsets
  static const char* zero_page = [] {
    void* z = nullptr;
    return hipGetSymbolAddress(&z, HIP_SYMBOL(tg_cw_zero_page)) == hipSuccess ? (const char*)z : (const char*)nullptr;
  }();
  if (!zero_page) return TG_E_BADARG;
  CwK k;
  k.in = (const char*)in; k.w = (const char*)w_packed; k.zero = zero_page; k.bias = bias; k.out = (char*)out;
  k.bits = (unsigned char*)relu_bits;
  k.H = H; k.W = W; k.Cout = Cout; k.act = act;
  k.tiles_x = (W + 15) / 16; k.tiles_y = (H + kTH - 1) / kTH;
  const long long nt = (long long)k.tiles_x * k.tiles_y * N;
  if (nt > 0x3fffffffLL) return TG_E_UNSUPPORTED;
  k.ntiles = (int)nt;
  // persistent grid: the cap's workgroups shared by the Cout/64 channel tiles, pixel tiles dealt evenly
  const int co_tiles = Cout / 64;
  const int cap = max_workgroups > 0 ? max_workgroups : 256;
  const int per = cap / co_tiles > 0 ? cap / co_tiles : 1;
  const int rounds = (k.ntiles + per - 1) / per;
  const int gx = (k.ntiles + rounds - 1) / rounds;
  dim3 grid((unsigned)gx, (unsigned)co_tiles);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == TG_F16) return Cin == 64 ? launch_cw<2, F16>(k, grid, st) : launch_cw<4, F16>(k, grid, st);
  return Cin == 64 ? launch_cw<2, BF16>(k, grid, st) : launch_cw<4, BF16>(k, grid, st);
}
