// One residual block of the generator trunk in ONE launch (bf16, 64 channels):
//
//     h   = relu(conv3x3(a, W1) + b1)          code/ops.py:45-54 (residual_block), code/models.py:66-69
//     out = a + conv3x3(h, W2)
//
// The recurrent pass runs 16 such blocks per frame on 4 x 32x32 pixels: as two launches each conv is ~6 us of which the
// MFMAs are a few hundred nanoseconds - the rest is the launch boundary, load latency and stores.  Fusing the pair halves
// the boundaries and loads `a` once.  What bounds the fused launch is moving 166 KB (147 KB of it weights) into the CU:
// a wave pays ~120 cycles to ISSUE one 1-KiB vector load (measured with s_memtime stamps, tools/stamp_resblock.py), and
// an in-order wave cannot issue MFMAs while it is stuck there.  Hence the shape of this kernel:
//   * a workgroup (8x8 output tile of one image) has EIGHT waves, two per SIMD: wave (w, kc) owns MFMA row tile w
//     (16 packed output-channel rows) for every pixel tile and the k-steps of channel chunk kc (split-K over the two
//     32-channel chunks).  While one wave of a SIMD waits on a load issue its partner issues MFMAs;
//   * the weights never touch LDS: each lane loads exactly the A-fragments it will feed to the MFMAs (9 x 16 B per conv
//     and wave, 1 KiB contiguous per wave-load) from the packed global image, a few k-steps ahead of their use, and the
//     k-loops wait for them one step at a time (vmcnt);
//   * LDS holds the 12x12 input patch, the 10x10 h region (conflict-free swizzled rows, below) and the exchange buffer
//     through which the two K-halves add their accumulators; each wave of a pair finalises half of the pixel tiles.
//   phase 1  patch loads and the head of the weight stream issued; patch -> LDS
//   phase 2  conv1 on the 10x10 halo region (7 MFMA pixel tiles), exchange, bias + relu, zero outside the image (conv2
//            pads h with zeros), h -> LDS (bf16, exactly what the unfused path would read back) and -> global (the
//            backward pass needs it: relu mask and weight-gradient operand)
//   phase 3  conv2 on the 8x8 tile from the LDS copy of h, exchange, + a (from the LDS patch), store
// MFMA operand roles and the packed-weight layout are those of conv_mfma.hip.
#include "common.h"
#include <cstdlib>
#include <type_traits>

#ifdef TG_STAMP
// Diagnostic build only (build.sh -DTG_STAMP): workgroup 0 records s_memtime at phase boundaries (tools/stamp_resblock.py)
__device__ long long tg_rb_stamps[16];
#define RB_STAMP(i)                                                                          \
  do {                                                                                       \
    if (blockIdx.x == 0 && threadIdx.x == 0) tg_rb_stamps[i] = (long long)__builtin_amdgcn_s_memtime(); \
  } while (0)
extern "C" int tg_debug_read_rb_stamps(long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(tg_rb_stamps), sizeof(long long) * n);
}
// ... and every wave of workgroup 0 at [0 kernel start | 1 conv1 begins | 2 conv1 done | 3 exchange barrier passed | 4 conv2 done]
__device__ long long tg_rb_wstamps[8 * 5];
#define RB_WSTAMP(i)                                                                         \
  do {                                                                                       \
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0)                                          \
      tg_rb_wstamps[(threadIdx.x >> 6) * 5 + (i)] = (long long)__builtin_amdgcn_s_memtime(); \
  } while (0)
extern "C" int tg_debug_read_rb_wstamps(long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(tg_rb_wstamps), sizeof(long long) * n);
}
#else
#define RB_STAMP(i) do {} while (0)
#define RB_WSTAMP(i) do {} while (0)
#endif

namespace {

// LDS images are rows of 64 bytes (32 bf16 channels of one pixel / one packed weight row), UNPADDED, with the 16-byte piece
// index XOR-swizzled by bit 2 of the row: piece' = piece ^ 2*((row >> 2) & 1).  ds_read_b128 services a wave in four fixed
// groups of 16 lanes ({0-3,12-15,20-27}, ...; MI355X_MICROARCH.md, LDS), each group needs 16 distinct 16-byte slots mod
// 256 B.  With this swizzle, a patch pitch of 18 rows for the 12-wide input patch and 16 for the 10-wide h region, every
// fragment read of both convolutions (all taps, all pixel tiles) is conflict-free; the 80-byte padded rows of conv_mfma.hip
// cost 2.6-3x on these pixel patterns (exhaustive count, tools/lds_layout.py).
constexpr int kRow = 64;
constexpr int kInW = 12, kInP = 18;  // input patch 12 pixels wide, stored with a pitch of 18 rows
constexpr int kHW = 10, kHP = 16;    // h region 10 pixels wide, stored with a pitch of 16 rows
constexpr int kLdsX = 8 * 4 * 1024;  // exchange: [wave][slot][lane][16 B]

// Output tile = 8 x TH pixels.  TH = 8: 12x12 patch, 10x10 h region (7 MFMA pixel tiles), 4 output pixel tiles.
// TH = 4: 12x8 patch, 10x6 region (4 tiles), 2 output tiles - twice the workgroups with ~55 % of the MFMA / LDS work each:
// the recurrent pass has only 4 x 32x32 pixels per launch (64 tiles of 8x8 on a 256-CU chip), and what a launch costs there
// is the serial time of ONE workgroup, of which the two convolutions are the larger half (stamps, tools/stamp_resblock.py:
// the fragment reads of conv1 alone are 1.3 of 6 us; the weight stream, once L2-resident, 0.5).
template <int TH> struct Geo {
  static constexpr int kInPix = (TH + 4) * kInW, kInRows = (TH + 4) * kInP;
  static constexpr int kHPix = (TH + 2) * kHW, kHRows = (TH + 2) * kHP;
  static constexpr int kLdsIn = 2 * kInRows * kRow;  // [chunk][row][64]
  static constexpr int kLdsH = 2 * kHRows * kRow;
  static constexpr int kLdsTotal = kLdsIn + kLdsH + kLdsX;
  static constexpr int NT1 = (kHPix + 15) / 16;  // pixel tiles of conv1 (the last one partial)
  static constexpr int NT2 = TH * 8 / 16;        // pixel tiles of conv2
  static constexpr int F1 = (NT1 + 1) / 2;       // conv1 tiles finalised by K half 0 (K half 1: the rest)
  static constexpr int NU = (2 * kInPix * 4 + 511) / 512;  // patch loads per thread
};

// byte offset of 16-byte piece `piece` of row `row` inside an image
__device__ __forceinline__ int lds_off(int row, int piece) { return row * kRow + ((piece ^ ((row >> 1) & 2)) << 4); }

struct ResblockK {
  const char* in;
  const char* w1;
  const float* b1;
  const char* w2;
  char* out_h;
  char* out_a;
  const char* pf1;  // packed weights the NEXT launch will stream (or null): pulled into this XCD's L2 ahead of time
  const char* pf2;
  int N, H, W, tiles_x, tiles_y, xcd_blocks;
  int skip;  // 1: out_a = in + conv2(h) (residual block); 0: out_a = conv2(h) (the conv-relu-conv pair of conv_trans.2)
  const char* hmask;  // BWD: the block's saved forward activation h (relu mask of the first stage)
};

template <typename T> __device__ __forceinline__ f32x4 mma(bf16x8 a, bf16x8 b, f32x4 c) { return Mma16<T>::run(a, b, c); }


// LDS-only barrier: __syncthreads() would also drain vmcnt, i.e. wait for the weight stream and the h stores
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// two 16-bit LDS offsets per register (the per-lane fragment addresses of every (pixel tile, tap) would otherwise be 63
// registers; two waves per SIMD leave 256 each)
template <int N> struct Packed16 {
  unsigned v[(N + 1) / 2];
  __device__ __forceinline__ void set(int i, int x) {  // i ascending from 0
    if (i & 1) v[i >> 1] |= (unsigned)x << 16; else v[i >> 1] = (unsigned)x;
  }
  __device__ __forceinline__ int get(int i) const { return (i & 1) ? (int)(v[i >> 1] >> 16) : (int)(v[i >> 1] & 0xffffu); }
};

template <typename T> __device__ __forceinline__ uint2 pack4(const float* v) {
  uint2 pk;
  pk.x = (unsigned)f32_to_bits16<T>(v[0]) | ((unsigned)f32_to_bits16<T>(v[1]) << 16);
  pk.y = (unsigned)f32_to_bits16<T>(v[2]) | ((unsigned)f32_to_bits16<T>(v[3]) << 16);
  return pk;
}

// BWD = the input-gradient of the same block, which has the same shape:
//     dH = relu'(h) * conv3x3^T(dOut, W2),   dIn = dOut + conv3x3^T(dH, W1)          (autograd of code/ops.py:45-54)
// i.e. stage 1 = transposed conv with the role-swapped packing of W2 (taps mirrored: weight slot 8 - t goes with spatial
// offset t), bias + relu replaced by the mask h > 0; stage 2 = transposed conv with W1, skip = dOut.  dH is stored like h
// (the weight-gradient launch of the first conv reads it; its channel sums are that conv's bias gradient).
template <bool BWD, typename T = BF16, int TH = 8>
__global__ __launch_bounds__(512) void resblock_kernel(const ResblockK p) {
  using G_ = Geo<TH>;
  constexpr int kInPix = G_::kInPix, kInRows = G_::kInRows, kHPix = G_::kHPix, kHRows = G_::kHRows;
  constexpr int kLdsIn = G_::kLdsIn, kLdsH = G_::kLdsH, NT1 = G_::NT1, NT2 = G_::NT2, F1 = G_::F1, NU = G_::NU;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* lds_in = smem;
  char* lds_h = smem + kLdsIn;
  char* lds_x = smem + kLdsIn + kLdsH;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform on purpose (SGPR)
  const int w = wid & 3, kc = wid >> 2;                      // MFMA row tile, channel chunk (K half)
  const int idx = lane & 15, g = lane >> 4;
  int bx = blockIdx.x;
  // XCD-aware tile order (TECOGAN_RB_XCD=1, A/B knob): workgroups go to the 8 XCDs round-robin, so with the plain order the eight
  // neighbours of a tile sit on other XCDs; this gives XCD x (blockIdx % 8 == x) a contiguous run of tiles (half an image at
  // 4 x 32 x 32) and keeps a tile's halo mostly inside its own XCD's L2
  if (p.xcd_blocks) bx = (bx & 7) * p.xcd_blocks + (bx >> 3);
  const int txb = bx % p.tiles_x;
  bx /= p.tiles_x;
  const int tyb = bx % p.tiles_y;
  const int n = bx / p.tiles_y;
  const int y0 = tyb * TH, x0 = txb * 8;
  const char* in_n = p.in + (size_t)n * p.H * p.W * 128;

  RB_STAMP(0);
  RB_WSTAMP(0);
  // ---- phase 1.  Patch loads are unconditional from a clamped address and zeroed afterwards: a load under a divergent
  // `if` makes the compiler wait for each one before issuing the next.
  u32x4 va[NU];
  int da[NU];
  bool ok[NU];
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int i = min(tid + u * 512, 2 * kInPix * 4 - 1);
    const int s = i & 3, r = i >> 2;
    const int cc = r >= kInPix ? 1 : 0, prow = r - cc * kInPix;
    const int py = (prow * 171) >> 11, px = prow - py * kInW;  // prow / 12, exact for prow <= 144
    const int iy = y0 - 2 + py, ix = x0 - 2 + px;
    da[u] = (tid + u * 512 < 2 * kInPix * 4) ? cc * kInRows * kRow + lds_off(py * kInP + px, s) : -1;
    ok[u] = iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
    const int cy = min(max(iy, 0), p.H - 1), cx = min(max(ix, 0), p.W - 1);
    va[u] = *reinterpret_cast<const u32x4*>(in_n + ((size_t)cy * p.W + cx) * 128 + cc * 64 + s * 16);
  }
  // A-fragments of row tile w for the 9 taps of chunk kc, W1 then W2: one stream of 18 loads, issued kAhead k-steps ahead
  // of their use.  Packed image: [tap][chunk][64 rows][64 B]; lane (idx, g) needs bytes 16g..16g+15 of row 16w + idx.
  const int wlane = ((kc * 64 + w * 16 + idx) * 64 + g * 16);
  bf16x8 wfr[18];
  auto issue_w = [&](int k) {  // compile-time k after unrolling
    const char* base = k < 9 ? p.w1 : p.w2;
#if defined(RB_DIAG_SAMEW)   // diagnostic: every weight load hits the same KiB (vector-L1 resident): the kernel without the stream
    wfr[k] = *reinterpret_cast<const bf16x8*>(base + wlane);
#else
    wfr[k] = *reinterpret_cast<const bf16x8*>(base + (size_t)(BWD ? 8 - k % 9 : k % 9) * 8192 + wlane);
#endif
  };
#ifndef RB_AHEAD
#define RB_AHEAD 9
#endif
#ifndef RB_AHEAD4
#define RB_AHEAD4 6
#endif
  // weight loads issued before the first MFMA (the rest: one per k-step).  8x8 tiles: all of W1 up front (the registers
  // exist anyway).  8x4 tiles have half the MFMAs per k-step to hide an issue behind, and every load issued up front
  // (~130 cycles of the wave's time each) delays the first MFMA: 6 measured best (7.07 vs 7.34 us at 9, 7.65 at 12)
  constexpr int kAhead = TH == 4 ? RB_AHEAD4 : RB_AHEAD;
#pragma unroll
  for (int k = 0; k < kAhead; ++k) issue_w(k);
  // lane (idx, g) of row tile w ends up with channels ch0 .. ch0+3 of pixel idx (row_to_channel<BF16> of common.h)
  const int chunk = w >> 1, half = w & 1;
  const int ch0 = 32 * chunk + 8 * g + 4 * half;
  f32x4 bias = {0.f, 0.f, 0.f, 0.f};
  if constexpr (!BWD) bias = *reinterpret_cast<const f32x4*>(p.b1 + ch0);
  // BWD: the relu mask of the (up to four) region tiles this wave finalises - 8 bytes per lane and tile, fetched now
  uint2 hm[F1];
  if constexpr (BWD) {
#pragma unroll
    for (int j = 0; j < F1; ++j) {
      const int hp = min((kc ? F1 + j : j) * 16 + idx, kHPix - 1);
      const int hy = (hp * 205) >> 11, hx = hp - hy * kHW;
      const int y = min(max(y0 - 1 + hy, 0), p.H - 1), x = min(max(x0 - 1 + hx, 0), p.W - 1);  // clamped: unused outside
      hm[j] = *reinterpret_cast<const uint2*>(p.hmask + (((size_t)n * p.H + y) * p.W + x) * 128 + ch0 * 2);
    }
  }
  RB_STAMP(1);

  // fragment offsets of conv1 (NT1 pixel tiles x 9 taps) inside one chunk image, computed while the loads are in flight.
  // Pixel tile t covers region pixels 16t .. 16t+15 (row-major, 10 wide); the last tile is partial: its spare lanes read a
  // clamped pixel.
  Packed16<NT1 * 9> xa;
#pragma unroll
  for (int t = 0; t < NT1; ++t) {
    int hp = t * 16 + idx;
    hp = hp < kHPix ? hp : kHPix - 1;
    const int hy = (hp * 205) >> 11, hx = hp - hy * kHW;  // hp / 10
#pragma unroll
    for (int tt = 0; tt < 9; ++tt) xa.set(t * 9 + tt, lds_off((hy + tt / 3) * kInP + hx + tt % 3, g));
  }
#pragma unroll
  for (int u = 0; u < NU; ++u)
    if (da[u] >= 0) *reinterpret_cast<u32x4*>(lds_in + da[u]) = ok[u] ? va[u] : u32x4{0u, 0u, 0u, 0u};
  RB_STAMP(2);
  lds_barrier();
  RB_STAMP(3);
  RB_WSTAMP(1);

  char* myx = lds_x + (wid * 4 * 64 + lane) * 16;               // exchange slots of this wave
  const char* px_ = lds_x + ((wid ^ 4) * 4 * 64 + lane) * 16;   // ... of the partner (same row tile, other K half)

  // ---- phase 2: conv1, k-steps of chunk kc.  Fragments of step st+1 are read from LDS before the MFMAs of step st.
  {
    f32x4 acc[NT1];
#pragma unroll
    for (int t = 0; t < NT1; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 xf[2][NT1];
    const char* img = lds_in + kc * kInRows * kRow;
    auto frags = [&](int tt, int buf) {
#pragma unroll
      for (int t = 0; t < NT1; ++t) xf[buf][t] = *reinterpret_cast<const bf16x8*>(img + xa.get(t * 9 + tt));
    };
    frags(0, 0);
#pragma unroll
    for (int tt = 0; tt < 9; ++tt) {
      if (tt + 1 < 9) frags(tt + 1, (tt + 1) & 1);
      if (tt + kAhead < 18) issue_w(tt + kAhead);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < NT1; ++t) acc[t] = mma<T>(wfr[tt], xf[tt & 1][t], acc[t]);
      __builtin_amdgcn_sched_barrier(0);
    }
    RB_STAMP(4);
    RB_WSTAMP(2);
    // exchange: K half 0 finalises tiles 0 .. F1-1, K half 1 the rest; each wave hands the other tiles to its partner
    auto finish1 = [&](auto T0, auto NT) {
      constexpr int t0 = decltype(T0)::value, nt = decltype(NT)::value, o0 = t0 ? 0 : F1, no = NT1 - nt;
#pragma unroll
      for (int j = 0; j < no; ++j) *reinterpret_cast<f32x4*>(myx + j * 1024) = acc[o0 + j];
      RB_STAMP(10);
      lds_barrier();
      RB_STAMP(11);
      RB_WSTAMP(3);
#pragma unroll
      for (int j = 0; j < nt; ++j) {
        const int t = t0 + j;
        const f32x4 other = *reinterpret_cast<const f32x4*>(px_ + j * 1024);
        const int hp = t * 16 + idx;
        if (hp < kHPix) {
          const int hy = (hp * 205) >> 11, hx = hp - hy * kHW;
          const int y = y0 - 1 + hy, x = x0 - 1 + hx;
          const bool inside = y >= 0 && y < p.H && x >= 0 && x < p.W;
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if constexpr (BWD) {
              const unsigned word = (e < 2) ? hm[j].x : hm[j].y;
              const float hv = bits16_to_f32<T>((unsigned short)((e & 1) ? (word >> 16) : (word & 0xffffu)));
              v[e] = hv > 0.f ? acc[t][e] + other[e] : 0.f;
            } else {
              v[e] = fmaxf(acc[t][e] + other[e] + bias[e], 0.f);
            }
            v[e] = inside ? v[e] : 0.f;
          }
          const uint2 pk = pack4<T>(v);
          *reinterpret_cast<uint2*>(lds_h + chunk * kHRows * kRow + lds_off(hy * kHP + hx, g) + half * 8) = pk;
          if (p.out_h && inside && hy >= 1 && hy <= TH && hx >= 1 && hx <= 8)   // (out_h null: inference, nobody reads h)
            *reinterpret_cast<uint2*>(p.out_h + (((size_t)n * p.H + y) * p.W + x) * 128 + ch0 * 2) = pk;
        }
      }
    };
    if (kc == 0) finish1(std::integral_constant<int, 0>{}, std::integral_constant<int, F1>{});
    else finish1(std::integral_constant<int, F1>{}, std::integral_constant<int, NT1 - F1>{});
  }
  RB_STAMP(5);
  lds_barrier();  // h complete
  RB_STAMP(6);

  // ---- phase 3: conv2 on the 8 x TH tile; pixel tile t = output rows 2t, 2t+1
  {
    Packed16<NT2 * 9> xb;
#pragma unroll
    for (int t = 0; t < NT2; ++t) {
      const int op = t * 16 + idx;
#pragma unroll
      for (int tt = 0; tt < 9; ++tt) xb.set(t * 9 + tt, lds_off(((op >> 3) + tt / 3) * kHP + (op & 7) + tt % 3, g));
    }
    f32x4 acc[NT2];
#pragma unroll
    for (int t = 0; t < NT2; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 xf[2][NT2];
    const char* img = lds_h + kc * kHRows * kRow;
    auto frags = [&](int tt, int buf) {
#pragma unroll
      for (int t = 0; t < NT2; ++t) xf[buf][t] = *reinterpret_cast<const bf16x8*>(img + xb.get(t * 9 + tt));
    };
    frags(0, 0);
    RB_STAMP(7);
#pragma unroll
    for (int tt = 0; tt < 9; ++tt) {
      if (tt + 1 < 9) frags(tt + 1, (tt + 1) & 1);
      if (9 + tt + kAhead < 18) issue_w(9 + tt + kAhead);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < NT2; ++t) acc[t] = mma<T>(wfr[9 + tt], xf[tt & 1][t], acc[t]);
      __builtin_amdgcn_sched_barrier(0);
    }
    RB_STAMP(8);
    RB_WSTAMP(4);
    // L2 prefetch for the next residual block: its 147 KB of weights were evicted from this XCD's L2 since the previous
    // frame used them (they come back from the Infinity Cache at ~1 us latency, which is what paces the weight stream
    // above).  The workgroups of one XCD (blockIdx.x % 8) each touch one eighth of the two images - two or three 1-KiB
    // loads per wave whose values are never used.
    // (plain volatile loads, consumed by an empty asm at the very end: the compiler then tracks the in-flight destination
    // registers; a hand-written asm load would let it reuse them while the data is still on its way)
    unsigned pfv0 = 0, pfv1 = 0, pfv2 = 0;
    if (p.pf1 && (TH == 8 || (blockIdx.x >> 3) < 8)) {
      const int slice = (blockIdx.x >> 3) & 7;
      const int off = slice * 9216 + (wid * 64 + lane) * 16;  // 9 KiB per slice and image: 8 waves x 1 KiB + 1 KiB
      pfv0 = *reinterpret_cast<const volatile unsigned*>(p.pf1 + off);
      pfv1 = *reinterpret_cast<const volatile unsigned*>(p.pf2 + off);
      if (wid < 2) pfv2 = *reinterpret_cast<const volatile unsigned*>((wid ? p.pf2 : p.pf1) + slice * 9216 + 8192 + lane * 16);
    }
    auto finish2 = [&](auto T0) {
      constexpr int t0 = decltype(T0)::value, o0 = t0 ? 0 : NT2 / 2;
#pragma unroll
      for (int j = 0; j < NT2 / 2; ++j) *reinterpret_cast<f32x4*>(myx + j * 1024) = acc[o0 + j];
      lds_barrier();
#pragma unroll
      for (int j = 0; j < NT2 / 2; ++j) {
        const int t = t0 + j;
        const f32x4 other = *reinterpret_cast<const f32x4*>(px_ + j * 1024);
        const int op = t * 16 + idx;
        const int oy = op >> 3, ox = op & 7;
        const int y = y0 + oy, x = x0 + ox;
        if (y < p.H && x < p.W) {
          // the skip connection comes from the LDS patch
          const uint2 rr = *reinterpret_cast<const uint2*>(lds_in + chunk * kInRows * kRow +
                                                           lds_off((oy + 2) * kInP + ox + 2, g) + half * 8);
          const float sk = p.skip ? 1.f : 0.f;
          float v[4];
          v[0] = acc[t][0] + other[0] + sk * bits16_to_f32<T>((unsigned short)(rr.x & 0xffffu));
          v[1] = acc[t][1] + other[1] + sk * bits16_to_f32<T>((unsigned short)(rr.x >> 16));
          v[2] = acc[t][2] + other[2] + sk * bits16_to_f32<T>((unsigned short)(rr.y & 0xffffu));
          v[3] = acc[t][3] + other[3] + sk * bits16_to_f32<T>((unsigned short)(rr.y >> 16));
          *reinterpret_cast<uint2*>(p.out_a + (((size_t)n * p.H + y) * p.W + x) * 128 + ch0 * 2) = pack4<T>(v);
        }
      }
    };
    if (kc == 0) finish2(std::integral_constant<int, 0>{});
    else finish2(std::integral_constant<int, NT2 / 2>{});
    asm volatile("" ::"v"(pfv0), "v"(pfv1), "v"(pfv2));
  }
  RB_STAMP(9);
}


// ---------------------------------------------------------------------------------------------------------------------------
// Three-way K split for the 8 x 4-tile forward launch (round 4; the recurrent pass: 160 launches per training step).
// Per-wave stamps of the kernel above (tools/stamp_resblock.py, profiles/r04_z_stamp_resblock.log): conv1 takes 2400 ticks for the
// older wave of each SIMD and 3950 for the younger one - with or without weight loads in the k-loop - for 576 cycles of MFMA: it
// is bound by its LDS fragment reads.  A wave there owns ONE 16-channel row tile and half of K, so each fragment it reads feeds one
// MFMA, and the four row-tile waves of a K half read the same 36 fragments (288 KB per workgroup).  Here a compute wave owns TWO
// row tiles (32 output channels) and ONE tap row (a third of K: 3 taps x 2 channel chunks = 6 k-steps): every fragment feeds two
// MFMAs, 144 KB instead of 288 for conv1 and 72 instead of 144 for conv2, the weight stream is the same 144 fragments but over six
// waves (24 each).  Waves 6 and 7 do not multiply; all eight share the patch load and the two finalising passes, which add the
// three K partials from the exchange buffer (conv1: 16 (row tile, pixel tile) pairs, two per wave; conv2: 8, one per wave).
// MEASURED SLOWER and not dispatched by default (-DRB_K3 builds it in): 6.58 vs 6.20 us per launch of the 16-launch trunk, chain
// 1.53 vs 1.47 ms, step 3.79 vs 3.70 ms (profiles/r04_z_rb_k3.log) at every depth of the weight stream (4 / 8 / 12 / 16 / 24 fragments
// ahead: 6.68 / 6.58 / 6.86 / 7.20 / 7.65).  The fragment reads are halved, but a compute wave now issues 24 weight loads instead of
// 18 (~130 ticks each, in order, nothing else can issue them for it) and two waves issue none: the longest per-wave issue chain
// grows from 2340 to 3120 ticks, which is what the workgroup then waits for.
constexpr int kLdsX3 = 6 * 8 * 1024;   // exchange: [compute wave][2 row tiles x 4 pixel tiles][lane][16 B]
#ifndef RB_K3_AHEAD
#define RB_K3_AHEAD 8                   // weight fragments in flight ahead of their k-step (two per k-step)
#endif
template <typename T>
__global__ __launch_bounds__(512) void resblock_k3_kernel(const ResblockK p) {
  constexpr int TH = 4;
  using G_ = Geo<TH>;
  constexpr int kInPix = G_::kInPix, kInRows = G_::kInRows, kHPix = G_::kHPix, kHRows = G_::kHRows;
  constexpr int kLdsIn = G_::kLdsIn, kLdsH = G_::kLdsH, NT1 = G_::NT1, NT2 = G_::NT2, NU = G_::NU;
  static_assert(NT1 == 4 && NT2 == 2, "8 x 4 tiles: 60 region pixels, 32 output pixels");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* lds_in = smem;
  char* lds_h = smem + kLdsIn;
  char* lds_x = smem + kLdsIn + kLdsH;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool comp = wid < 6;                 // compute waves
  const int rp = wid & 1, dy = wid >> 1;     // ... row-tile pair (row tiles 2rp, 2rp + 1), tap row
  const int fw = wid & 3, fq = wid >> 2;     // finalising role: row tile fw; conv1 pixel tiles 2fq, 2fq + 1; conv2 pixel tile fq
  const int idx = lane & 15, g = lane >> 4;
  int bx = blockIdx.x;
  if (p.xcd_blocks) bx = (bx & 7) * p.xcd_blocks + (bx >> 3);
  const int txb = bx % p.tiles_x;
  bx /= p.tiles_x;
  const int tyb = bx % p.tiles_y;
  const int n = bx / p.tiles_y;
  const int y0 = tyb * TH, x0 = txb * 8;
  const char* in_n = p.in + (size_t)n * p.H * p.W * 128;

  // ---- phase 1: patch loads (unconditional from a clamped address, zeroed afterwards), head of the weight stream
  u32x4 va[NU];
  int da[NU];
  bool ok[NU];
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int i = min(tid + u * 512, 2 * kInPix * 4 - 1);
    const int s_ = i & 3, r = i >> 2;
    const int cc = r >= kInPix ? 1 : 0, prow = r - cc * kInPix;
    const int py = (prow * 171) >> 11, px = prow - py * kInW;  // prow / 12, exact for prow <= 144
    const int iy = y0 - 2 + py, ix = x0 - 2 + px;
    da[u] = (tid + u * 512 < 2 * kInPix * 4) ? cc * kInRows * kRow + lds_off(py * kInP + px, s_) : -1;
    ok[u] = iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
    const int cy = min(max(iy, 0), p.H - 1), cx = min(max(ix, 0), p.W - 1);
    va[u] = *reinterpret_cast<const u32x4*>(in_n + ((size_t)cy * p.W + cx) * 128 + cc * 64 + s_ * 16);
  }
  // A-fragments of row tiles 2rp, 2rp + 1 for the k-steps of tap row dy: step s = (chunk s / 3, column tap s % 3); fragment
  // j = 12 * conv + 2 * s + row tile.  Packed image: [tap][chunk][64 rows][64 B]; lane (idx, g): bytes 16g .. 16g + 15 of its row.
  const char* const w1l = p.w1 + (size_t)dy * 3 * 8192 + (rp * 32 + idx) * 64 + g * 16;
  const char* const w2l = p.w2 + (size_t)dy * 3 * 8192 + (rp * 32 + idx) * 64 + g * 16;
  bf16x8 wfr[24];
  auto issue_w = [&](int j) {  // compile-time j after unrolling
    const int s_ = (j % 12) / 2, rtl = j & 1;
    wfr[j] = *reinterpret_cast<const bf16x8*>((j < 12 ? w1l : w2l) + (s_ % 3) * 8192 + ((s_ / 3) * 64 + rtl * 16) * 64);
  };
  constexpr int kAhead = RB_K3_AHEAD;
  if (comp) {
#pragma unroll
    for (int j = 0; j < kAhead; ++j) issue_w(j);
  }
  // finalising role: lane (idx, g) of row tile fw ends up with channels ch0 .. ch0 + 3 of pixel idx
  const int chunk = fw >> 1, half = fw & 1;
  const int ch0 = 32 * chunk + 8 * g + 4 * half;
  const f32x4 bias = *reinterpret_cast<const f32x4*>(p.b1 + ch0);
  // fragment offsets of conv1 inside one chunk image: pixel tile t (region pixels 16t .. 16t + 15, row-major, 10 wide; the last
  // tile is partial: its spare lanes read a clamped pixel) under column tap dx of this wave's tap row
  Packed16<NT1 * 3> xa;
#pragma unroll
  for (int t = 0; t < NT1; ++t) {
    int hp = t * 16 + idx;
    hp = hp < kHPix ? hp : kHPix - 1;
    const int hy = (hp * 205) >> 11, hx = hp - hy * kHW;  // hp / 10
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) xa.set(t * 3 + dx, lds_off((hy + dy) * kInP + hx + dx, g));
  }
#pragma unroll
  for (int u = 0; u < NU; ++u)
    if (da[u] >= 0) *reinterpret_cast<u32x4*>(lds_in + da[u]) = ok[u] ? va[u] : u32x4{0u, 0u, 0u, 0u};
  lds_barrier();

  // ---- phase 2: conv1, six k-steps per compute wave; the fragments of step s + 1 are read before the MFMAs of step s
  if (comp) {
    f32x4 acc[2][NT1];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int t = 0; t < NT1; ++t) acc[a][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 xf[2][NT1];
    auto frags = [&](int s_, int buf) {
      const char* img = lds_in + (s_ / 3) * kInRows * kRow;
#pragma unroll
      for (int t = 0; t < NT1; ++t) xf[buf][t] = *reinterpret_cast<const bf16x8*>(img + xa.get(t * 3 + s_ % 3));
    };
    frags(0, 0);
#pragma unroll
    for (int s_ = 0; s_ < 6; ++s_) {
      if (s_ + 1 < 6) frags(s_ + 1, (s_ + 1) & 1);
      if (kAhead + 2 * s_ < 24) issue_w(kAhead + 2 * s_);
      if (kAhead + 2 * s_ + 1 < 24) issue_w(kAhead + 2 * s_ + 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int t = 0; t < NT1; ++t) acc[a][t] = mma<T>(wfr[2 * s_ + a], xf[s_ & 1][t], acc[a][t]);
      __builtin_amdgcn_sched_barrier(0);
    }
    char* myx = lds_x + (wid * 8 * 64 + lane) * 16;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int t = 0; t < NT1; ++t) *reinterpret_cast<f32x4*>(myx + (a * 4 + t) * 1024) = acc[a][t];
  }
  lds_barrier();
  // finalise: sum of the three tap-row partials + bias, relu, zero outside the image (conv2 pads h with zeros), h -> LDS (bf16,
  // exactly what the unfused path would read back) and -> global (the backward pass needs it)
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int t = 2 * fq + j;
    const char* px_ = lds_x + (((fw >> 1) * 8 + (fw & 1) * 4 + t) * 64 + lane) * 16;
    const f32x4 s0 = *reinterpret_cast<const f32x4*>(px_);
    const f32x4 s1 = *reinterpret_cast<const f32x4*>(px_ + 2 * 8 * 1024);
    const f32x4 s2 = *reinterpret_cast<const f32x4*>(px_ + 4 * 8 * 1024);
    const int hp = t * 16 + idx;
    if (hp < kHPix) {
      const int hy = (hp * 205) >> 11, hx = hp - hy * kHW;
      const int y = y0 - 1 + hy, x = x0 - 1 + hx;
      const bool inside = y >= 0 && y < p.H && x >= 0 && x < p.W;
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] = fmaxf((s0[e] + s1[e]) + s2[e] + bias[e], 0.f);
        v[e] = inside ? v[e] : 0.f;
      }
      const uint2 pk = pack4<T>(v);
      *reinterpret_cast<uint2*>(lds_h + chunk * kHRows * kRow + lds_off(hy * kHP + hx, g) + half * 8) = pk;
      if (p.out_h && inside && hy >= 1 && hy <= TH && hx >= 1 && hx <= 8)   // (out_h null: inference, nobody reads h)
        *reinterpret_cast<uint2*>(p.out_h + (((size_t)n * p.H + y) * p.W + x) * 128 + ch0 * 2) = pk;
    }
  }
  lds_barrier();  // h complete (and every partial of conv1 has been read: the exchange buffer is free again)

  // ---- phase 3: conv2 on the 8 x 4 tile from the LDS copy of h; pixel tile t = output rows 2t, 2t + 1
  if (comp) {
    Packed16<NT2 * 3> xb;
#pragma unroll
    for (int t = 0; t < NT2; ++t) {
      const int op = t * 16 + idx;
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) xb.set(t * 3 + dx, lds_off(((op >> 3) + dy) * kHP + (op & 7) + dx, g));
    }
    f32x4 acc[2][NT2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int t = 0; t < NT2; ++t) acc[a][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 xf[2][NT2];
    auto frags = [&](int s_, int buf) {
      const char* img = lds_h + (s_ / 3) * kHRows * kRow;
#pragma unroll
      for (int t = 0; t < NT2; ++t) xf[buf][t] = *reinterpret_cast<const bf16x8*>(img + xb.get(t * 3 + s_ % 3));
    };
    frags(0, 0);
#pragma unroll
    for (int s_ = 0; s_ < 6; ++s_) {
      if (s_ + 1 < 6) frags(s_ + 1, (s_ + 1) & 1);
      if (12 + kAhead + 2 * s_ < 24) issue_w(12 + kAhead + 2 * s_);
      if (12 + kAhead + 2 * s_ + 1 < 24) issue_w(12 + kAhead + 2 * s_ + 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int t = 0; t < NT2; ++t) acc[a][t] = mma<T>(wfr[12 + 2 * s_ + a], xf[s_ & 1][t], acc[a][t]);
      __builtin_amdgcn_sched_barrier(0);
    }
    char* myx = lds_x + (wid * 4 * 64 + lane) * 16;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int t = 0; t < NT2; ++t) *reinterpret_cast<f32x4*>(myx + (a * 2 + t) * 1024) = acc[a][t];
  }
  lds_barrier();
  {
    const int t = fq;
    const char* px_ = lds_x + (((fw >> 1) * 4 + (fw & 1) * 2 + t) * 64 + lane) * 16;
    const f32x4 s0 = *reinterpret_cast<const f32x4*>(px_);
    const f32x4 s1 = *reinterpret_cast<const f32x4*>(px_ + 2 * 4 * 1024);
    const f32x4 s2 = *reinterpret_cast<const f32x4*>(px_ + 4 * 4 * 1024);
    const int op = t * 16 + idx;
    const int oy = op >> 3, ox = op & 7;
    const int y = y0 + oy, x = x0 + ox;
    if (y < p.H && x < p.W) {
      // the skip connection comes from the LDS patch
      const uint2 rr = *reinterpret_cast<const uint2*>(lds_in + chunk * kInRows * kRow + lds_off((oy + 2) * kInP + ox + 2, g) + half * 8);
      const float sk = p.skip ? 1.f : 0.f;
      float v[4];
      v[0] = (s0[0] + s1[0]) + s2[0] + sk * bits16_to_f32<T>((unsigned short)(rr.x & 0xffffu));
      v[1] = (s0[1] + s1[1]) + s2[1] + sk * bits16_to_f32<T>((unsigned short)(rr.x >> 16));
      v[2] = (s0[2] + s1[2]) + s2[2] + sk * bits16_to_f32<T>((unsigned short)(rr.y & 0xffffu));
      v[3] = (s0[3] + s1[3]) + s2[3] + sk * bits16_to_f32<T>((unsigned short)(rr.y >> 16));
      *reinterpret_cast<uint2*>(p.out_a + (((size_t)n * p.H + y) * p.W + x) * 128 + ch0 * 2) = pack4<T>(v);
    }
  }
}

}  // namespace

namespace {
template <typename T>
int resblock_launch(bool bwd, const void* in, const void* wa, const float* b1, const void* wb2, const void* hmask, void* out_h,
                    void* out_a, int N, int H, int W, int add_skip, const void* next_w1, const void* next_w2, void* stream) {
  ResblockK k;
  k.in = (const char*)in; k.w1 = (const char*)wa; k.b1 = b1; k.w2 = (const char*)wb2;
  k.out_h = (char*)out_h; k.out_a = (char*)out_a; k.hmask = (const char*)hmask;
  k.pf1 = (next_w1 && next_w2) ? (const char*)next_w1 : nullptr;
  k.pf2 = (const char*)next_w2;
  k.N = N; k.H = H; k.W = W; k.skip = add_skip ? 1 : 0;
  k.tiles_x = (W + 7) / 8;
  // 8x4 tiles while 8x8 tiles would leave half of the chip's CUs without a workgroup (TECOGAN_RB_TILE=8/4 forces one)
  static const int forced = [] { const char* e = kTgExperiments ? getenv("TECOGAN_RB_TILE") : nullptr; return e ? atoi(e) : 0; }();
  const long long blocks8 = (long long)k.tiles_x * ((H + 7) / 8) * N;
  const int th = forced == 4 || forced == 8 ? forced : (blocks8 <= 128 ? 4 : 8);
  k.tiles_y = (H + th - 1) / th;
  const long long blocks = (long long)k.tiles_x * k.tiles_y * N;
  if (blocks > 0x7fffffffLL) return TG_E_UNSUPPORTED;
  static std::atomic<bool> attr_done{false};
  if (!attr_done) {
    TG_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(resblock_kernel<false, T, 8>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, Geo<8>::kLdsTotal));
    TG_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(resblock_kernel<true, T, 8>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, Geo<8>::kLdsTotal));
    TG_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(resblock_kernel<false, T, 4>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, Geo<4>::kLdsTotal));
    TG_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(resblock_kernel<true, T, 4>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, Geo<4>::kLdsTotal));
    TG_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(resblock_k3_kernel<T>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, Geo<4>::kLdsIn + Geo<4>::kLdsH + kLdsX3));
    attr_done = true;
  }
  static const int xcd = [] { const char* e = kTgExperiments ? getenv("TECOGAN_RB_XCD") : nullptr; return e ? atoi(e) : 0; }();
  k.xcd_blocks = (xcd && blocks % 8 == 0) ? (int)(blocks / 8) : 0;
  const dim3 grid((unsigned)blocks), blk(512);
  hipStream_t st = (hipStream_t)stream;
  if (th == 4) {
    if (bwd) hipLaunchKernelGGL((resblock_kernel<true, T, 4>), grid, blk, Geo<4>::kLdsTotal, st, k);
#ifdef RB_K3   // opt-in build (tools/build_variant.sh): the three-way K split above - parity green, 6.58 vs 6.20 us per launch
    else hipLaunchKernelGGL((resblock_k3_kernel<T>), grid, blk, Geo<4>::kLdsIn + Geo<4>::kLdsH + kLdsX3, st, k);   // (no L2 prefetch hint)
#else
    else hipLaunchKernelGGL((resblock_kernel<false, T, 4>), grid, blk, Geo<4>::kLdsTotal, st, k);
#endif
  } else {
    if (bwd) hipLaunchKernelGGL((resblock_kernel<true, T, 8>), grid, blk, Geo<8>::kLdsTotal, st, k);
    else hipLaunchKernelGGL((resblock_kernel<false, T, 8>), grid, blk, Geo<8>::kLdsTotal, st, k);
  }
  return tg_launch_status();
}
}  // namespace

extern "C" int tg_resblock_fwd(int dtype, const void* in, const void* w1_packed, const float* b1, const void* w2_packed,
                               void* out_h, void* out_a, int N, int H, int W, int C, int add_skip,
                               const void* next_w1_packed, const void* next_w2_packed, void* stream) {
  if (!in || !w1_packed || !b1 || !w2_packed || !out_a || N <= 0 || H <= 0 || W <= 0) return TG_E_BADARG;   // (out_h may be null)
  if ((dtype != TG_BF16 && dtype != TG_F16) || C != 64) return TG_E_UNSUPPORTED;  // the trunk shape, 16-bit; else two tg_conv launches
  if (!tg_aligned16(in) || !tg_aligned16(w1_packed) || !tg_aligned16(w2_packed) || (out_h && !tg_aligned16(out_h)) ||
      !tg_aligned16(out_a) || !tg_aligned16(b1))
    return TG_E_ALIGN;
  if (dtype == TG_F16)
    return resblock_launch<F16>(false, in, w1_packed, b1, w2_packed, nullptr, out_h, out_a, N, H, W, add_skip, next_w1_packed,
                                next_w2_packed, stream);
  return resblock_launch<BF16>(false, in, w1_packed, b1, w2_packed, nullptr, out_h, out_a, N, H, W, add_skip, next_w1_packed,
                               next_w2_packed, stream);
}

extern "C" int tg_resblock_bwd(int dtype, const void* dout, const void* w2_dgrad_packed, const void* h, const void* w1_dgrad_packed,
                               void* out_dh, void* out_din, int N, int H, int W, int C, const void* next_wa_packed,
                               const void* next_wb_packed, void* stream) {
  if (!dout || !w2_dgrad_packed || !h || !w1_dgrad_packed || !out_dh || !out_din || N <= 0 || H <= 0 || W <= 0)
    return TG_E_BADARG;
  if ((dtype != TG_BF16 && dtype != TG_F16) || C != 64) return TG_E_UNSUPPORTED;
  if (!tg_aligned16(dout) || !tg_aligned16(w2_dgrad_packed) || !tg_aligned16(w1_dgrad_packed) || !tg_aligned16(out_dh) ||
      !tg_aligned16(out_din) || !tg_aligned16(h))
    return TG_E_ALIGN;
  if (dtype == TG_F16)
    return resblock_launch<F16>(true, dout, w2_dgrad_packed, nullptr, w1_dgrad_packed, h, out_dh, out_din, N, H, W, 1,
                                next_wa_packed, next_wb_packed, stream);
  return resblock_launch<BF16>(true, dout, w2_dgrad_packed, nullptr, w1_dgrad_packed, h, out_dh, out_din, N, H, W, 1,
                               next_wa_packed, next_wb_packed, stream);
}
