// 3x3 stride-1 convolution (forward, or input-gradient = taps mirrored) for the DENSE launches of the step - the batched
// generator backward (40 samples), the discriminator's residual stages (12 samples) - with the weights held in REGISTERS
// for the lifetime of a persistent workgroup:
//
//     out[n, y, x, co] = epilogue( sum_{tap, ci} W[tap][co][ci] * in[n, y + dy, x + dx, ci] )      bf16 / fp16, NHWC
//
// Round 4: WAVE-SPECIALISED.  The round-2/3 kernel ran all eight waves through the same phases - next patch's loads, k-loop,
// epilogue, patch store, barrier - in lockstep.  Its stamps (tools/stamp_rw.py, profiles/r04_b_stamp_rw_v1.log; ticks per 128-pixel
// tile of a 64 -> 64 launch): k-loop 2000-2600 of 6400, epilogue 1500-1800, issuing the next patch 1100-1700, barrier skew up to
// 1100 - the matrix pipes sat idle for two thirds of every tile; and for Cin = 128: k-loop 5500-6000 of 14 500, split-K exchange
// 600-2100, epilogue 2600-3000, LDS-DMA issue 4000-5000 (each of the 8 DMA instructions re-loaded the zero page's address from
// the GOT: a scalar-memory round trip).  Now the eight waves of a workgroup have two roles (waves w and w + 4 share a SIMD):
//   * CONSUMERS (waves 0-3, one per SIMD) hold the weights - 32 output channels x 9 taps x 64 input channels = 36 A-fragments,
//     144 VGPRs - and do nothing but the k-loop: B-fragments from the LDS patch image, 144 MFMAs per tile back to back (2304
//     cycles), then the 32 accumulator registers go to an LDS accumulator image ([pixel][64 channels] fp32, 272-byte pixel pitch: conflict-free
//     for the 16-byte stores).  Cin = 64: wave (wc, rg) = channel half x 4 of the tile's 8 rows.  Cin = 128: wave (wc, kh) =
//     channel half x K half over a 4-row tile; the two K halves land in two accumulator images and the epilogue adds them (the
//     split-K exchange, its barrier and its 32 KB of LDS are gone).
//   * PRODUCERS (waves 4-7) fill the NEXT tile's patch by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write
//     pass; positions outside the image read a page of zeros) and run the PREVIOUS tile's epilogue from the accumulator image:
//     +bias, +res, act, *act'(mask), 16-byte NHWC stores (8 lanes = one pixel's 64 channels: 1 KiB contiguous per wave-store),
//     per-channel statistics in registers.  The epilogue's mask / residual rows are fetched one tile ahead.
//   * ONE barrier per tile; patch and accumulator images are double-buffered, so consumers of tile i, the DMA of tile i + 1 and
//     the epilogue of tile i - 1 overlap.  A tile costs what its 144 MFMAs per SIMD cost (2304 cycles) plus the accumulator store.
// LDS: 2 x 30 KB patches + 2 x 34 KB accumulator images (Cin = 64); 2 x 36 KB + 2 x 34 KB (Cin = 128).
// LDS image and packed-weight layout are those of conv_mfma.hip's pipelined path (swizzled 64-byte rows, patch pitch 24).
//
// Replaces aten::conv2d / convolution_backward(input) of the 3x3 layers (code/models.py:54-58,68,73-76,102 via
// code/ops.py:57-63; autograd of code/train.py:336,340) where tg_conv is the general entry point.
#ifndef TG_ST_AUX
#define TG_ST_AUX "sc1"   // this kernel's results are written THROUGH the L2 (common.h, tg_store16; profiles/r05_u_write_through_ab.log)
#endif
#include "common.h"
#include <type_traits>

#ifndef RW_EARLY_STEPS
#define RW_EARLY_STEPS 8   // k-steps whose weight fragments the consumers fetch before the first barrier (18: all, the round-4 start)
#endif
#ifdef TG_STAMP
// Diagnostic build only (-DTG_STAMP, tools/stamp_rw.py): consumer wave 0 and producer wave 4 of workgroup 0 record s_memtime at
// the phase boundaries of the prologue and of their first four iterations: [role][0..3 prologue | 4 + 6 * iteration + phase]
__device__ long long tg_rw_stamps[2 * 32];
#define RW_STAMP(i)                                                                                         \
  do {                                                                                                      \
    if (blockIdx.x == 0 && blockIdx.y == 0 && (threadIdx.x & 255) == 0 && (i) < 32)                          \
      tg_rw_stamps[(threadIdx.x >> 8) * 32 + (i)] = (long long)__builtin_amdgcn_s_memtime();                \
  } while (0)
extern "C" int tg_debug_read_rw_stamps(long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(tg_rw_stamps), sizeof(long long) * n);
}
#else
#define RW_STAMP(i) do {} while (0)
#endif

#ifndef RW_PRODUCER_PRIO
#define RW_PRODUCER_PRIO 2
#endif

// out-of-image patch positions (and the 6 pitch-padding rows per patch row) are DMA'd from here
__device__ __attribute__((aligned(16))) unsigned int tg_rw_zero_page[4];

namespace {

constexpr int kRow = 64, kPitch = 24, kPW = 18;
constexpr int kRedBytes = 1024;   // statistics: [2][64] fp32 sums of the four producer waves + a ticket
constexpr int kAccPitch = 272;  // bytes per pixel of an accumulator image (64 fp32 + 16: 8 lanes' 16-byte stores hit 32 distinct banks)

template <int NCH> struct Geo {
  static constexpr int TH = NCH == 2 ? 8 : 4;                    // output rows per tile (x 16 columns)
  static constexpr int PH = TH + 2;                              // patch rows
  static constexpr int kChunkBytes = PH * kPitch * kRow;         // one 32-channel chunk of the patch: 15 KiB / 9 KiB
  static constexpr int kBufBytes = NCH * kChunkBytes;            // 30 KiB / 36 KiB
  static constexpr int kBlocks = kBufBytes / 1024;               // 1-KiB DMA blocks per patch
  static constexpr int kBlocksPerChunk = kChunkBytes / 1024;
  static constexpr int NB = (kBlocks + 3) / 4;                   // DMA instructions per producer wave and tile
  static constexpr int kPix = TH * 16;                           // 128 / 64
  static constexpr int kHalves = NCH == 4 ? 2 : 1;               // K halves that land in separate accumulator images
  static constexpr int kAccHalf = kPix * kAccPitch;
  static constexpr int kAccBytes = kHalves * kAccHalf;           // 34 816 either way
  static constexpr int KS = kPix / 32;                           // 8-pixel store instructions per producer wave and tile
  static constexpr int kLds = 2 * kBufBytes + 2 * kAccBytes;       // (+ kRedBytes of statistics scratch behind it)
  // Cin = 64 (round 5): the workgroup's 72 KB of weights are staged ONCE by LDS-DMA - in the accumulator images, which nobody uses
  // before the first tile is through, the statistics scratch and 3 KB more - and the consumers take their fragments from there
  static constexpr bool kStageW = NCH == 2;
  static constexpr int kStage = 2 * kBufBytes, kStageBytes = 18 * 4096;
  static constexpr int kLdsAll = kStageW ? (kStage + kStageBytes > kLds + 1024 ? kStage + kStageBytes : kLds + 1024) : kLds + 1024;
  static_assert(kChunkBytes % 1024 == 0, "a DMA block must not straddle two chunk images");
};

__device__ __forceinline__ int swz(int row, int piece) { return row * kRow + ((piece ^ ((row >> 1) & 2)) << 4); }

struct RwK {
  const char* in;
  const char* w;
  const float* bias;
  const char* res;
  const char* mask;
  char* out;
  float* stats;
  const char* zero;   // 16 bytes of zeros (tg_rw_zero_page): a kernel argument, so that its address sits in SGPRs for the whole kernel
  int N, H, W, Cout;
  int tiles_x, tiles_y, ntiles;
  int flip, act, mask_mode, stats_groups, stats_mode, stats_replicas;
};

template <typename T> __device__ __forceinline__ f32x4 mma(bf16x8 a, bf16x8 b, f32x4 c) { return Mma16<T>::run(a, b, c); }

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gsrc), "s"(lds_dst) : "memory", "m0");
}

// LDS-only barrier: __syncthreads() would also wait for every vector-memory operation in flight
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <typename T> __device__ __forceinline__ void unpack8(const u32x4 r, float* v) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    v[2 * i] = bits16_to_f32<T>((unsigned short)(r[i] & 0xffffu));
    v[2 * i + 1] = bits16_to_f32<T>((unsigned short)(r[i] >> 16));
  }
}

// two floats -> one register of two 16-bit values, round to nearest even (ONE v_cvt_pk_* instruction; element-wise conversion +
// shift + or costs three)
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
template <typename T> __device__ __forceinline__ unsigned pack2(float lo, float hi);
template <> __device__ __forceinline__ unsigned pack2<BF16>(float lo, float hi) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{lo, hi}, bf16x2_t));
}
template <> __device__ __forceinline__ unsigned pack2<F16>(float lo, float hi) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{lo, hi}, f16x2_t));
}

// SM: statistics of the stored values - 0 none, 1 per-channel sums (a bias gradient), 2 sums and sums of squares (batch norm).
template <int NCH, int SM, typename T = BF16>
// (arguments one by one, the ones a wave needs first in front: the first 16 dwords are preloaded into SGPRs with the wave -
// csrc/build.sh, -amdgpu-kernarg-preload-count; profiles/r05_k_kernarg_preload_ab.log)
__global__ __launch_bounds__(512) void conv3_rw_kernel(const char* a_in, const char* a_w, const char* a_zero, int a_H, int a_W, int a_Cout,
                                                       int a_tiles_x, int a_tiles_y, int a_ntiles, int a_flip, int a_N, char* a_out,
                                                       const char* a_mask, const char* a_res, const float* a_bias, float* a_stats,
                                                       int a_act, int a_mask_mode, int a_stats_groups, int a_stats_mode, int a_stats_replicas) {
  RwK p;
  p.in = a_in; p.w = a_w; p.zero = a_zero; p.H = a_H; p.W = a_W; p.Cout = a_Cout; p.tiles_x = a_tiles_x; p.tiles_y = a_tiles_y;
  p.ntiles = a_ntiles; p.flip = a_flip; p.N = a_N; p.out = a_out; p.mask = a_mask; p.res = a_res; p.bias = a_bias; p.stats = a_stats;
  p.act = a_act; p.mask_mode = a_mask_mode; p.stats_groups = a_stats_groups; p.stats_mode = a_stats_mode; p.stats_replicas = a_stats_replicas;
  using G = Geo<NCH>;
  static_assert(NCH == 2 || NCH == 4, "Cin = 64 or 128");
  constexpr bool STATS = SM > 0;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const accb = smem + 2 * G::kBufBytes;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int co_base = blockIdx.y * 64;
  // this workgroup's tiles: blockIdx.x, + gridDim.x, ...  (the grid never exceeds the tile count)
  const int ntl = (p.ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  auto decode = [&](int t, int& n, int& ty0, int& tx0) {
    const int txb = t % p.tiles_x;
    t /= p.tiles_x;
    const int tyb = t % p.tiles_y;
    n = t / p.tiles_y;
    ty0 = tyb * G::TH;
    tx0 = txb * 16;
  };
  RW_STAMP(0);
#ifndef RW_NO_STAGE
  if constexpr (G::kStageW) {
    // Every consumer wave used to load its 36 A-fragments itself - the two waves of a channel half the same 36 KB: 144 KB through
    // the CU's intake (~22 B/clk) in front of the first tile, of which 16 fragments up front and 20 inside the first tile's k-loop.
    // Now all eight waves DMA the 72 KB once: block (tap so, chunk ci) = 64 rows x 64 B; wave (u = wid % 4, ci = wid / 4) brings rows
    // 16 u .. + 15 of chunk ci for every tap; the lane's 16 bytes are physical piece lane % 4 of its row (swz is an involution).
    const unsigned lds0w = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const int u = wid & 3, ci = wid >> 2, row = 16 * u + (lane >> 2);
    const char* src = p.w + ((size_t)co_base + row) * 64 + (((lane & 3) ^ ((row >> 1) & 2)) << 4);
#pragma unroll
    for (int so = 0; so < 9; ++so) {
      const int slot = p.flip ? 8 - so : so;
      glds16(src + (size_t)(slot * NCH + ci) * p.Cout * 64, lds0w + G::kStage + (so * 2 + ci) * 4096 + u * 1024);
    }
  }
#endif

  if (wid < 4) {
    // ============================================================ CONSUMER: weights in registers, k-loop, accumulators -> LDS
    // (v_mfma_f32_16x16x32.  The 32x32x16 form was built too - profiles/r04_f_*_mfma32.log: same k-loop time, the producers no
    // faster, the launches 3-10 % slower - so the producers' pace is not set by the matrix instructions' hold on the vector issue.)
    const int idx = lane & 15, g = lane >> 4;
    const int wc = wid & 1;                                   // channel half: packed rows 32*wc .. 32*wc + 31
    const int kh = NCH == 4 ? wid >> 1 : 0;                   // input-channel half (Cin = 128)
    const int r0 = NCH == 4 ? 0 : (wid >> 1) * 4;             // first of the wave's 4 tile rows
    // A-fragments of packed rows 32*wc + 16*a + idx for 9 taps x 2 chunks, in k-loop order.  Packed image
    // [tap][chunk][Cout rows][64 B]; the input-gradient launch pairs spatial offset `so` with slot 8 - so.
    // Only the fragments of k-steps 0 .. kEarly - 1 are fetched here; the rest go out INSIDE the first tile's k-loop, two per
    // step.  A wave pays ~130 ticks to issue one of these 1-KiB loads, 36 of them were 4300-5500 ticks of prologue during
    // which the matrix pipe idled and the producers (patch 0 in LDS after ~3100 ticks) waited at the first barrier; most of the
    // step's 190 launches of this kernel walk 2-6 tiles, so that was a quarter of their time (profiles/r04_t_stamp_*).
    bf16x8 wfr[2][9][2];
    const char* const wl = p.w + ((size_t)co_base + wc * 32 + idx) * 64 + g * 16;
    auto wload = [&](int s_) {   // both fragments of k-step s_ (compile-time after unrolling)
      const int ci = s_ / 9, so = s_ % 9;
      const int slot = p.flip ? 8 - so : so;
      const int chunk = kh * 2 + ci;
#pragma unroll
      for (int a = 0; a < 2; ++a)
        wfr[ci][so][a] = *reinterpret_cast<const bf16x8*>(wl + ((size_t)(slot * NCH + chunk) * p.Cout + a * 16) * 64);
    };
#ifndef RW_NO_STAGE
    constexpr bool kStaged = G::kStageW;
#else
    constexpr bool kStaged = false;
#endif
    constexpr int kEarly = kStaged ? 18 : RW_EARLY_STEPS;
    if constexpr (!kStaged) {
#pragma unroll
      for (int s_ = 0; s_ < kEarly; ++s_) wload(s_);
    }
    RW_STAMP(1);
    // fragment addresses of the wave's rows inside one chunk image (column taps 0..2); row taps add multiples of the pitch
    // (two 16-bit offsets per register)
    unsigned xbase[6];
#pragma unroll
    for (int q = 0; q < 12; q += 2)
      xbase[q / 2] = (unsigned)swz((r0 + q / 3) * kPitch + idx + q % 3, g) |
                     ((unsigned)swz((r0 + (q + 1) / 3) * kPitch + idx + (q + 1) % 3, g) << 16);
    auto xoff = [&](int b, int c) {  // compile-time (b, c) after unrolling
      const int q = b * 3 + c;
      return (int)((q & 1) ? xbase[q / 2] >> 16 : xbase[q / 2] & 0xffffu);
    };
    // lane (idx, g) ends with channels 32*wc + 8g .. + 7 (two row-interleaved MFMA tiles, common.h) of pixel (r0 + b, idx)
    const int acc_off = kh * G::kAccHalf + (r0 * 16 + idx) * kAccPitch + (wc * 32 + 8 * g) * 4;
    RW_STAMP(2);
    if constexpr (kStaged) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's part of the weights has landed
      lds_barrier();                                     // ... everybody's
#pragma unroll
      for (int ci = 0; ci < 2; ++ci)
#pragma unroll
        for (int so = 0; so < 9; ++so)
#pragma unroll
          for (int a = 0; a < 2; ++a)
            wfr[ci][so][a] = *reinterpret_cast<const bf16x8*>(smem + G::kStage + (so * 2 + ci) * 4096 + swz(wc * 32 + a * 16 + idx, g));
    }
    lds_barrier();   // patch 0 is in LDS (staged weights: and every consumer has its fragments - the area is the accumulators' again)
    RW_STAMP(3);
    for (int i = 0; i <= ntl; ++i) {
      RW_STAMP(4 + 6 * i + 0);
      if (i < ntl) {
        f32x4 acc[2][4];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        const char* img = smem + (i & 1) * G::kBufBytes + kh * 2 * G::kChunkBytes;
        // k-loop: 18 steps (2 chunks x 9 taps) of 4 B-fragment reads + 8 MFMAs.  The wave is alone on its SIMD's matrix pipe, so
        // the fragments of steps s + 1 and s + 2 are in flight while step s multiplies (left to itself the compiler reads one
        // fragment, waits lgkmcnt(0), issues two MFMAs: the whole LDS latency per 32 cycles of matrix work)
        bf16x8 xf[3][4];
        auto frags = [&](int s_, int buf) {   // compile-time arguments after unrolling
          const int ci = s_ / 9, so = s_ % 9;
#pragma unroll
          for (int b = 0; b < 4; ++b)
            xf[buf][b] = *reinterpret_cast<const bf16x8*>(img + ci * G::kChunkBytes + xoff(b, so % 3) + (so / 3) * kPitch * kRow);
        };
        frags(0, 0);
        frags(1, 1);
        auto kloop = [&](auto first) {   // first tile: the late weight fragments are fetched on the way (see kEarly)
#pragma unroll
          for (int s_ = 0; s_ < 18; ++s_) {
            if (s_ + 2 < 18) frags(s_ + 2, (s_ + 2) % 3);
            if constexpr (decltype(first)::value) {
              if (kEarly + s_ < 18) wload(kEarly + s_);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
              for (int a = 0; a < 2; ++a) {
#ifdef RW_DIAG_NOMFMA   // diagnostic: the consumer without its matrix instructions (what do the producers cost then?)
                acc[a][b][0] += __builtin_bit_cast(f32x4, xf[s_ % 3][b])[a] + __builtin_bit_cast(f32x4, wfr[s_ / 9][s_ % 9][a])[0];
#else
                acc[a][b] = mma<T>(wfr[s_ / 9][s_ % 9][a], xf[s_ % 3][b], acc[a][b]);
#endif
              }
            __builtin_amdgcn_sched_barrier(0);
          }
        };
        if (i == 0) kloop(std::true_type{});
        else kloop(std::false_type{});
        RW_STAMP(4 + 6 * i + 1);
        char* dst = accb + (i & 1) * G::kAccBytes + acc_off;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          *reinterpret_cast<f32x4*>(dst + b * 16 * kAccPitch) = acc[0][b];
          *reinterpret_cast<f32x4*>(dst + b * 16 * kAccPitch + 16) = acc[1][b];
        }
      }
      RW_STAMP(4 + 6 * i + 2);
      lds_barrier();
      RW_STAMP(4 + 6 * i + 3);
    }
    RW_STAMP(28);
    return;
  }

  // ================================================================ PRODUCER: patch DMA one tile ahead, epilogue one tile behind
  const int pw = wid - 4;
  // (static priority: the consumer on this SIMD issues MFMAs back to back and, being the older wave, would win every arbitration -
  // the producer's few hundred vector instructions per tile then crawl; MI355X_MICROARCH.md, 'Two waves per SIMD', item 4)
  __builtin_amdgcn_s_setprio(RW_PRODUCER_PRIO);
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  const size_t pix_bytes = (size_t)NCH * 64;
  // ---- DMA: block j = pw + 4u of a patch buffer is 16 rows of one chunk image; the lane's 16 bytes: row 16 (j % bpc) + lane / 4,
  // physical piece lane % 4 = logical piece ^ swizzle (swz is an involution).  Per block and lane, the same for every tile: patch
  // row / column - 1 (invalid positions - pitch padding, blocks beyond the buffer - get a row that is outside every image) and the
  // byte offset inside the pixel.  (Unpacked: the producers have registers to spare and every vector instruction they issue competes
  // with the consumer's MFMA stream on the same SIMD.)
  int dpy[G::NB], dpx[G::NB], dof[G::NB];
#pragma unroll
  for (int u = 0; u < G::NB; ++u) {
    const int j = pw + 4 * u;
    const int cc = j / G::kBlocksPerChunk, row = (j - cc * G::kBlocksPerChunk) * 16 + (lane >> 2);
    const int py = row / kPitch, px = row - py * kPitch;
    const int piece = (lane & 3) ^ ((row >> 1) & 2);
    const bool valid = j < G::kBlocks && px < kPW;
    dpy[u] = valid ? py - 1 : -(1 << 20);
    dpx[u] = px - 1;
    dof[u] = cc * 64 + piece * 16;
  }
  struct Tile { int n, ty0, tx0; };
  auto tile_of = [&](int t) {
    Tile r;
    decode(t, r.n, r.ty0, r.tx0);
    return r;
  };
  auto dma_patch = [&](const Tile& tl, int buf) {   // asynchronous: vmcnt + barrier before anyone reads it
    const char* in_n = p.in + (size_t)tl.n * p.H * p.W * pix_bytes;
    const unsigned dst0 = lds0 + buf * G::kBufBytes + pw * 1024;
#pragma unroll
    for (int u = 0; u < G::NB; ++u) {
      if (pw + 4 * u < G::kBlocks) {  // wave-uniform
        const int iy = tl.ty0 + dpy[u], ix = tl.tx0 + dpx[u];
        const bool ok = ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);   // (no short-circuit branches)
        const char* src = in_n + (unsigned)((iy * p.W + ix) * (int)pix_bytes + dof[u]);    // (an image is < 4 GB)
#ifdef RW_DMA_NT_BYTES   // A/B only (profiles/r05_u_write_through_ab.log, section 6): inputs too large for the L2s are streamed past them
        if ((size_t)p.N * p.H * p.W * pix_bytes >= (size_t)RW_DMA_NT_BYTES)
          asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off nt" : : "v"(ok ? src : p.zero), "s"(dst0 + u * 4096) : "memory", "m0");
        else
#endif
        glds16(ok ? src : p.zero, dst0 + u * 4096);
      }
    }
  };
  // ---- epilogue: store instruction k of this wave covers pixels pw * kPix/4 + 8k .. + 7 of the tile (8 consecutive columns of one
  // row); the lane's pixel is + lane / 8, its channels 8 (lane % 8) .. + 7 of the workgroup's 64
  const int q8 = lane & 7;
  const int ch0 = co_base + 8 * q8;
  const int pl0 = pw * (G::kPix / 4) + (lane >> 3);
  float bias_r[8];
#pragma unroll
  for (int e = 0; e < 8; e += 4) {
    f32x4 t = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) t = *reinterpret_cast<const f32x4*>(p.bias + ch0 + e);
    bias_r[e] = t[0]; bias_r[e + 1] = t[1]; bias_r[e + 2] = t[2]; bias_r[e + 3] = t[3];
  }
  // the epilogue's mask rows (a launch without a mask: its residual rows), fetched one tile ahead
  const char* pre_src = p.mask_mode != TG_MASK_NONE ? p.mask : p.res;
  u32x4 pre[G::KS];
#pragma unroll
  for (int k = 0; k < G::KS; ++k) pre[k] = u32x4{0u, 0u, 0u, 0u};
  auto issue_pre = [&](const Tile& tl) {
#ifdef TG_EXPERIMENTS   // (bit-exact and slower in the step; even unused it cost the default kernel 0.03 ms per step: profiles/r06_n_mask_code_ab.log)
    if (p.mask_mode == TG_MASK_RELU_BITS) {
      // the 1-bit form of a ReLU mask ([N][H][W][Cout / 8] bytes, written by the forward launch that produced the activation): the
      // lane's 8 channels are ONE byte - a sixteenth of the 16-bit rows' bytes through the CU's vector-memory pipe
      const unsigned char* src_n = reinterpret_cast<const unsigned char*>(p.mask) + (size_t)tl.n * p.H * p.W * (p.Cout / 8);
#pragma unroll
      for (int k = 0; k < G::KS; ++k) {
        const int pl = pl0 + 8 * k;
        const int cy = min(tl.ty0 + (pl >> 4), p.H - 1), cx = min(tl.tx0 + (pl & 15), p.W - 1);
        pre[k][0] = src_n[(unsigned)((cy * p.W + cx) * p.Cout + ch0) / 8u];
      }
    } else
#endif
    if (pre_src) {
      const char* src_n = pre_src + (size_t)tl.n * p.H * p.W * p.Cout * 2;
#pragma unroll
      for (int k = 0; k < G::KS; ++k) {
        const int pl = pl0 + 8 * k;
        const int cy = min(tl.ty0 + (pl >> 4), p.H - 1), cx = min(tl.tx0 + (pl & 15), p.W - 1);  // clamped: unused outside the image
        pre[k] = *reinterpret_cast<const u32x4*>(src_n + (unsigned)(((cy * p.W + cx) * p.Cout + ch0) * 2));
      }
    }
  };
  float s1[STATS ? 8 : 1], s2[SM == 2 ? 8 : 1];
  int cur_grp = -1;
  if constexpr (STATS) {
#pragma unroll
    for (int e = 0; e < 8; ++e) s1[e] = 0.f;
  }
  if constexpr (SM == 2) {
#pragma unroll
    for (int e = 0; e < 8; ++e) s2[e] = 0.f;
  }
  // per-channel statistics of the stored values: lanes -> wave (over the 8 pixels of a store instruction) -> the four producer
  // waves through an LDS accumulator -> ONE global atomic per channel and workgroup, issued by whichever wave arrives last (a ticket
  // in LDS; a wave's LDS operations are served in order, so every wave's sums are in before its ticket is).  A persistent workgroup
  // flushes once per statistics group it touches; flushes of consecutive groups are an iteration's barrier apart.
  // (Per-wave global atomics - 8 instructions of 8 lanes each - took 147 us instead of 34 on c20's input-gradient: 5120 atomic
  // instructions on the same four cache lines serialise at ~25 ns each; profiles/r04_i_g_bwd_anatomy_v2e.log.)
  float* const red = reinterpret_cast<float*>(smem + G::kLds);   // [2][64] sums, [128] the ticket
#ifndef RW_NO_STAGE
  constexpr bool kStagedP = G::kStageW;
#else
  constexpr bool kStagedP = false;
#endif
  if constexpr (STATS && !kStagedP) {
    if (tid - 256 < 132) red[tid - 256] = 0.f;   // published by the first barrier
  }
  auto flush_stats = [&](int grp) {
    if constexpr (STATS) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
#pragma unroll
        for (int m = 8; m < 64; m <<= 1) {
          s1[e] += __shfl_xor(s1[e], m);
          if constexpr (SM == 2) s2[e] += __shfl_xor(s2[e], m);
        }
      }
      if (lane < 8) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          atomicAdd(&red[8 * q8 + e], s1[e]);
          if constexpr (SM == 2) atomicAdd(&red[64 + 8 * q8 + e], s2[e]);
        }
      }
      unsigned ticket = 0;
      if (lane == 0) ticket = atomicAdd(reinterpret_cast<unsigned*>(red + 128), 1u);
      ticket = __builtin_amdgcn_readfirstlane(ticket);
      if ((ticket & 3u) == 3u) {   // the last of the four producer waves
        const size_t rep = (size_t)(blockIdx.x & (p.stats_replicas - 1)) * p.stats_groups * 2 * p.Cout;
        float* dst = p.stats + rep + (size_t)grp * 2 * p.Cout + co_base + lane;
        atomicAdd(dst, red[lane]);
        red[lane] = 0.f;
        if constexpr (SM == 2) {
          atomicAdd(dst + p.Cout, red[64 + lane]);
          red[64 + lane] = 0.f;
        }
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        s1[e] = 0.f;
        if constexpr (SM == 2) s2[e] = 0.f;
      }
    }
  };
  // (Round 5, profiles/r05_x_rw_producer_diag.log: every byte the producers move goes through the CU's vector-memory pipe at ~25 B/clk -
  // 64 KB per tile of a masked input-gradient = 2600 ticks in the issue queue, THEN 1300 of arithmetic: 4350 against the consumers'
  // 3300-3600; without the stores, the DMA or the row loads the launch is consumer-bound (147 -> 116 us on c6's input-gradient).  Two
  // re-orderings that would overlap the arithmetic with the queueing were built and measured SLOWER: group-wise interleaving of store /
  // DMA / row load / arithmetic (203 us: a dozen uniform branches and their joins per group) and two waves computing first and
  // storing last from a second register set (163 us).)
  // The epilogue of a tile is split over two iterations: COMPUTE (accumulator image -> 16-bit results in registers) while the
  // consumers multiply the next tile, and the STORES of those registers at the top of the following iteration.  That way the only
  // vector-memory operations an iteration issues behind its DMA are loads whose data is needed an iteration later, so the wait in
  // front of the barrier never waits for a store and the DMA - issued first - has the whole iteration to land.  (One fused loop made
  // the compiler wait vmcnt(0) for the mask rows in each of its KS bodies, which also waited for the store the previous body had just
  // issued: 5400 ticks per tile; with the stores behind the bodies and the DMA last, the DMA's landing was exposed instead: 1000-3300
  // ticks; tools/stamp_rw.py, profiles/r04_c_*, r04_e_*.)
  u32x4 ov[G::KS];
  unsigned okm = 0;
  [[maybe_unused]] int st_it = 0;
  Tile st_tile = {0, 0, 0};
  // The epilogue's options are launch-uniform.  Tested inside the unrolled body they became 72 scalar branches and ~270 register
  // moves at their joins per tile (2500 ticks for ~150 useful vector instructions: tools/stamp_rw.py, profiles/r04_s_*); so the
  // body is instantiated for the three combinations the step's launches use, chosen once per tile, plus the general form:
  //   0 forward: + bias, activation, no mask / residual      1 input-gradient under a ReLU mask (no bias / activation / residual)
  //   2 input-gradient + residual (no bias / activation / mask)      3 anything else (every option tested at run time)
  //   4 input-gradient under a 1-BIT ReLU mask (no bias / activation / residual; the entry point refuses other combinations)
  const int emode =
#ifdef TG_EXPERIMENTS
                    (p.mask_mode == TG_MASK_RELU_BITS) ? 4 :
#endif
                    (p.mask_mode == TG_MASK_NONE && !p.res) ? 0
                    : (p.mask_mode == TG_MASK_RELU && !p.res && !p.bias && p.act == TG_ACT_NONE) ? 1
                    : (p.mask_mode == TG_MASK_NONE && p.res && !p.bias && p.act == TG_ACT_NONE) ? 2 : 3;
  // activation as max(v, slope * v): slope 1 = none, 0 = ReLU, 0.2 = LeakyReLU (mode 0: no branch on p.act)
  const float act_slope = p.act == TG_ACT_RELU ? 0.f : p.act == TG_ACT_LRELU ? 0.2f : 1.f;
  auto compute = [&](const Tile& tl, int buf, const u32x4* mk, auto MODE) {
    constexpr int M = decltype(MODE)::value;
    const int n = tl.n, ty0 = tl.ty0, tx0 = tl.tx0;
    if constexpr (STATS) {
      const int grp = n / (p.N / p.stats_groups);
      if (grp != cur_grp) {
        if (cur_grp >= 0) flush_stats(cur_grp);
        cur_grp = grp;
      }
    }
    const char* ab = accb + buf * G::kAccBytes + q8 * 32;
    f32x4 a0[G::KS], a1[G::KS];
#pragma unroll
    for (int k = 0; k < G::KS; ++k) {
      const int pl = pl0 + 8 * k;
      a0[k] = *reinterpret_cast<const f32x4*>(ab + pl * kAccPitch);
      a1[k] = *reinterpret_cast<const f32x4*>(ab + pl * kAccPitch + 16);
      if constexpr (NCH == 4) {   // the other K half
        a0[k] += *reinterpret_cast<const f32x4*>(ab + G::kAccHalf + pl * kAccPitch);
        a1[k] += *reinterpret_cast<const f32x4*>(ab + G::kAccHalf + pl * kAccPitch + 16);
      }
    }
#ifdef TG_STAMP
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (diagnostic: stamp 5 of the iteration = the accumulator reads have landed)
    RW_STAMP(4 + 6 * st_it + 5);
#endif
    const char* const res_t = (M == 3 && p.res) ? p.res + ((((size_t)n * p.H + ty0) * p.W + tx0) * p.Cout) * 2 : nullptr;
    okm = 0;
    st_tile = tl;
#pragma unroll
    for (int k = 0; k < G::KS; ++k) {
      const int pl = pl0 + 8 * k;
      const int ry = pl >> 4, rx = pl & 15;
      const bool ok = (ty0 + ry < p.H) & (tx0 + rx < p.W);
      okm |= ok ? (1u << k) : 0u;
      float v[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] = a0[k][e];
        v[4 + e] = a1[k][e];
      }
      if constexpr (M == 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          v[e] += bias_r[e];                         // (zeros without a bias)
          v[e] = fmaxf(v[e], act_slope * v[e]);
        }
      } else if constexpr (M == 1) {
        // mask value > 0 on its 16-bit pattern (sign clear, not zero): the low half as the sign of word << 16, the high half as
        // word > 0xffff - no unpacking to float
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int w_ = (int)mk[k][e];
          v[2 * e] = (int)((unsigned)w_ << 16) > 0 ? v[2 * e] : 0.f;
          v[2 * e + 1] = w_ > 0xffff ? v[2 * e + 1] : 0.f;
        }
#ifdef TG_EXPERIMENTS
      } else if constexpr (M == 4) {
        const unsigned bm = mk[k][0];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (bm >> e) & 1u ? v[e] : 0.f;
#endif
      } else if constexpr (M == 2) {
        float r[8];
        unpack8<T>(mk[k], r);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += r[e];
      } else {
        if (p.bias) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += bias_r[e];
        }
        if (p.res) {
          float r[8];
          if (p.mask_mode == TG_MASK_NONE) unpack8<T>(mk[k], r);
          else if (ok) Vec<T>::load(res_t + (unsigned)((ry * p.W + rx) * p.Cout + ch0) * 2u, r);   // (res AND mask: the trunk's first block only)
          else {
#pragma unroll
            for (int e = 0; e < 8; ++e) r[e] = 0.f;
          }
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += r[e];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], act_slope * v[e]);
        if (p.mask_mode != TG_MASK_NONE) {
          const float neg = p.mask_mode == TG_MASK_LRELU ? 0.2f : 0.f;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int w_ = (int)mk[k][e];
            v[2 * e] = (int)((unsigned)w_ << 16) > 0 ? v[2 * e] : neg * v[2 * e];
            v[2 * e + 1] = w_ > 0xffff ? v[2 * e + 1] : neg * v[2 * e + 1];
          }
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) ov[k][e] = pack2<T>(v[2 * e], v[2 * e + 1]);
      if constexpr (STATS) {
        if (ok) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            s1[e] += v[e];
            if constexpr (SM == 2) s2[e] += v[e] * v[e];
          }
        }
      }
    }
  };
  auto store_results = [&]() {   // of the tile `compute` ran on last: the tile's first output pixel (uniform) + a 32-bit lane offset
    char* const out_t = p.out + ((((size_t)st_tile.n * p.H + st_tile.ty0) * p.W + st_tile.tx0) * p.Cout) * 2;
#pragma unroll
    for (int k = 0; k < G::KS; ++k) {
      const int pl = pl0 + 8 * k;
      if ((okm >> k) & 1u) tg_store16(out_t + (unsigned)(((pl >> 4) * p.W + (pl & 15)) * p.Cout + ch0) * 2u, ov[k]);
    }
  };

  Tile cur = tile_of((int)blockIdx.x), prev = cur, st_prev = cur;   // the consumers' tile of iteration i, of i - 1; st_prev: see below
  dma_patch(cur, 0);
  RW_STAMP(1);
  RW_STAMP(2);
  if constexpr (kStagedP) {
    // this wave's 9 weight blocks are older than its patch blocks (8, or 7 for the waves behind the buffer's last block)
    if (pw < ((G::kBlocks & 3) ? (G::kBlocks & 3) : 4)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G::NB) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G::NB - 1) : "memory");
    lds_barrier();   // the weights are staged; the consumers take their fragments while patch 0 is still landing
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // patch 0 has landed
  lds_barrier();
  if constexpr (STATS && kStagedP) {
    if (tid - 256 < 132) red[tid - 256] = 0.f;   // (the scratch was part of the staging area) published by the first tile's barrier
  }
  RW_STAMP(3);
  int tile = (int)blockIdx.x;
  for (int i = 0; i <= ntl; ++i, tile += (int)gridDim.x) {
    st_it = i;
    RW_STAMP(4 + 6 * i + 0);
#ifndef RW_DIAG_NOSTORE   // (diagnostic builds, tools/stamp_rw.py + profiles/r05_x_rw_producer_diag.log: what does each part of the producers' issue phase cost?)
    if (i >= 2) store_results();   // tile i - 2, computed during iteration i - 1
#endif
    // the previous tile's mask / residual rows arrived during iteration i - 1 (its vmcnt(0)); taken over into registers the compiler
    // does not connect with a load any more, so that nothing below waits on the vector-memory counter
    u32x4 mk[G::KS];
#pragma unroll
    for (int k = 0; k < G::KS; ++k) {
      mk[k] = pre[k];
      asm volatile("" : "+v"(mk[k]));
    }
    // the next tile's patch (its buffer was last read in iteration i - 1) and this tile's mask / residual rows
    prev = cur;
    if (i + 1 < ntl) {
      cur = tile_of(tile + (int)gridDim.x);
#ifndef RW_DIAG_NODMA
      dma_patch(cur, (i + 1) & 1);
#endif
    }
#ifndef RW_DIAG_NOROWS
    if (i < ntl) issue_pre(prev);
#endif
    RW_STAMP(4 + 6 * i + 1);
    // the previous tile's accumulators were published by the barrier that ended iteration i - 1
    if (i >= 1) {
      if (emode == 0) compute(st_prev, (i - 1) & 1, mk, std::integral_constant<int, 0>{});
      else if (emode == 1) compute(st_prev, (i - 1) & 1, mk, std::integral_constant<int, 1>{});
      else if (emode == 2) compute(st_prev, (i - 1) & 1, mk, std::integral_constant<int, 2>{});
#ifdef TG_EXPERIMENTS
      else if (emode == 4) compute(st_prev, (i - 1) & 1, mk, std::integral_constant<int, 4>{});
#endif
      else compute(st_prev, (i - 1) & 1, mk, std::integral_constant<int, 3>{});
    }
    st_prev = prev;
    RW_STAMP(4 + 6 * i + 2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the DMA (and the rows) have landed before the barrier publishes the patch
    RW_STAMP(4 + 6 * i + 3);
    lds_barrier();
    RW_STAMP(4 + 6 * i + 4);
  }
  if (ntl >= 1) store_results();   // the last tile
  if constexpr (STATS) {
    if (cur_grp >= 0) flush_stats(cur_grp);
  }
  RW_STAMP(28);
}

template <int NCH, int SM, typename T>
int launch_rw(const RwK& k, dim3 grid, hipStream_t st) {
  constexpr int lds = Geo<NCH>::kLdsAll;   // (kLds + kRedBytes; Cin = 64: + the tail of the weight staging area)
  auto fn = conv3_rw_kernel<NCH, SM, T>;
  static std::atomic<bool> attr_done{false};  // one-time function attribute (benign race: idempotent)
  if (!attr_done) {
    TG_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    attr_done = true;
  }
  hipLaunchKernelGGL(fn, grid, dim3(512), lds, st, k.in, k.w, k.zero, k.H, k.W, k.Cout, k.tiles_x, k.tiles_y, k.ntiles, k.flip, k.N, k.out,
                     k.mask, k.res, k.bias, k.stats, k.act, k.mask_mode, k.stats_groups, k.stats_mode, k.stats_replicas);
  return tg_launch_status();
}

}  // namespace

extern "C" int tg_conv3x3_rw(int dtype, const void* in, const void* w_packed, const float* bias, const void* res,
                             const void* mask, void* out, float* stats, int N, int H, int W, int Cin, int Cout, int flip,
                             int act, int mask_mode, int stats_mode, int stats_groups, int stats_replicas, int max_workgroups,
                             void* stream) {
  if (!in || !w_packed || !out || N <= 0 || H <= 0 || W <= 0) return TG_E_BADARG;
  if ((dtype != TG_BF16 && dtype != TG_F16) || (Cin != 64 && Cin != 128) || Cout <= 0 || Cout % 64) return TG_E_UNSUPPORTED;
  if (act != TG_ACT_NONE && act != TG_ACT_RELU && act != TG_ACT_LRELU) return TG_E_UNSUPPORTED;
  if (mask_mode != TG_MASK_NONE && !mask) return TG_E_BADARG;
  if (mask_mode < TG_MASK_NONE || mask_mode > TG_MASK_RELU_BITS || mask_mode == TG_MASK_BNZ) return TG_E_BADARG;
  if (mask_mode == TG_MASK_RELU_BITS && (res || bias || act != TG_ACT_NONE || !kTgExperiments)) return TG_E_UNSUPPORTED;   // the plain masked input-gradient only, experiments build only
  if (stats && (stats_groups <= 0 || N % stats_groups || stats_mode < 1 || stats_mode > 2)) return TG_E_BADARG;
  if (stats && (stats_replicas < 1 || (stats_replicas & (stats_replicas - 1)))) return TG_E_BADARG;  // a power of two
  if (!tg_aligned16(in) || !tg_aligned16(w_packed) || !tg_aligned16(out) || (res && !tg_aligned16(res)) ||
      (mask && !tg_aligned16(mask)) || (bias && !tg_aligned16(bias)))
    return TG_E_ALIGN;
  static const char* zero_page = [] {
    void* z = nullptr;
    return hipGetSymbolAddress(&z, HIP_SYMBOL(tg_rw_zero_page)) == hipSuccess ? (const char*)z : (const char*)nullptr;
  }();
  if (!zero_page) return TG_E_BADARG;
  RwK k;
  k.in = (const char*)in; k.w = (const char*)w_packed; k.bias = bias; k.res = (const char*)res;
  k.mask = (const char*)mask; k.out = (char*)out; k.stats = stats; k.zero = zero_page;
  k.N = N; k.H = H; k.W = W; k.Cout = Cout;
  const int th = Cin == 64 ? Geo<2>::TH : Geo<4>::TH;
  k.tiles_x = (W + 15) / 16; k.tiles_y = (H + th - 1) / th;
  const long long nt = (long long)k.tiles_x * k.tiles_y * N;
  if (nt > 0x7fffffffLL) return TG_E_UNSUPPORTED;
  k.ntiles = (int)nt;
  k.flip = flip ? 1 : 0; k.act = act; k.mask_mode = mask ? mask_mode : TG_MASK_NONE;
  k.stats_groups = stats ? stats_groups : 1;
  k.stats_mode = stats ? stats_mode : 0;
  k.stats_replicas = stats ? stats_replicas : 1;
  // persistent grid: one workgroup per CU (256 CUs shared by the Cout/64 channel tiles), pixel tiles dealt evenly
  const int co_tiles = Cout / 64;
  int cap = max_workgroups > 0 ? max_workgroups : 256;
  int per = cap / co_tiles > 0 ? cap / co_tiles : 1;
  const int rounds = (k.ntiles + per - 1) / per;
  const int gx = (k.ntiles + rounds - 1) / rounds;
  dim3 grid((unsigned)gx, (unsigned)co_tiles);
  hipStream_t st = (hipStream_t)stream;
  const int sm = k.stats_mode;
#define RW_GO(NCH, TAG) (sm == 2 ? launch_rw<NCH, 2, TAG>(k, grid, st) : sm == 1 ? launch_rw<NCH, 1, TAG>(k, grid, st) \
                                 : launch_rw<NCH, 0, TAG>(k, grid, st))
  if (dtype == TG_F16) return Cin == 64 ? RW_GO(2, F16) : RW_GO(4, F16);
  return Cin == 64 ? RW_GO(2, BF16) : RW_GO(4, BF16);
#undef RW_GO
}
