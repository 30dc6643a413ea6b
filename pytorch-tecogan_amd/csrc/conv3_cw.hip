// 3x3 stride-1 convolution (forward, or input-gradient = taps mirrored), 64 reduction channels, with EIGHT EQUAL WAVES (round 5):
//
//     out[n, y, x, co] = epilogue( sum_{tap, ci} W[tap][co][ci] * in[n, y + dy, x + dx, ci] )      bf16 / fp16, NHWC, Cin = 64
//
// conv3_rw.hip splits a workgroup into four consumer waves (weights in registers, k-loop, accumulators -> an LDS image) and four
// producer waves (patch DMA, epilogue from the image, stores).  Its long launches are PRODUCER-bound: a tile's 64 KB (patch, mask
// rows, results) take 2600 ticks of the CU's vector-memory pipe during which the producers sit in their issue queue, and only then do
// they start 1300 ticks of epilogue arithmetic - 4350 per tile against the consumers' 3300-3600 (profiles/r05_x_rw_producer_diag.log);
// two re-orderings of the producer iteration were slower.  convt_cw.hip showed the way out: no roles.  Here every wave
//   * owns 32 of the workgroup's 64 output channels x 2 of the tile's 8 rows (wave = (wc, rg)) with the 9 taps' weights for its channel
//     half in registers (144 VGPRs, staged once per workgroup by LDS-DMA, as conv3_rw does for Cin = 64),
//   * brings its share of the next tile's patch by LDS-DMA (4 one-KiB blocks) and its own mask / residual rows (2 loads),
//   * runs its k-loop (18 steps of 2 fragment reads + 4 MFMAs: the two waves of a SIMD fill each other's gaps), and
//   * finishes its 32 pixels x 32 channels straight from the accumulators: + bias / residual, activation, act'(mask), statistics,
//     16-byte stores (written through the L2).
// No accumulator image (68 KB of LDS and 64 KB of LDS traffic per tile less), ONE barrier per tile, and the waves' memory
// instructions, arithmetic and matrix work overlap because the waves drift apart between barriers.
// (A second form - waves 4-7 carrying their accumulators over the barrier and finishing tile i - 1 at the START of iteration i, so that
// one wave of a SIMD multiplies while the other does arithmetic - was built, parity-green, and measured slower: 6.3 vs 5.3 us on the
// chain's launches, 164 vs 146 us on c6's input-gradient.  A tile costs its 2304 matrix cycles PLUS the ~300 vector instructions per
// SIMD of addresses and epilogue, which the matrix stream does not hide - in either arrangement, and in conv3_rw's.)
// LDS image, packed weights, epilogue semantics and the persistent grid are conv3_rw.hip's; Cin = 128 stays there (its weights
// do not fit eight full copies of a channel half).
//
// Replaces aten::conv2d / convolution_backward(input) of the 64-channel 3x3 layers (code/models.py:54-58,68,73-76,102 via
// code/ops.py:57-63; autograd of code/train.py:336,340).
#ifndef TG_ST_AUX
#define TG_ST_AUX "sc1"   // results are written THROUGH the L2 (common.h, tg_store16; profiles/r05_u_write_through_ab.log)
#endif
#include "rbw_common.h"
#include <atomic>
#include <type_traits>

#ifdef TG_STAMP
// diagnostic build (tools/stamp_conv3_cw.py): waves 0 and 4 of workgroup 0: [wave][0..3 prologue | 4 + 4 * tile + phase]
__device__ long long tg_c3cw_stamps[2 * 32];
#define C3_STAMP(i)                                                                                  \
  do {                                                                                               \
    if (blockIdx.x == 0 && blockIdx.y == 0 && (threadIdx.x & 255) == 0 && (i) < 32)                   \
      tg_c3cw_stamps[(threadIdx.x >> 8) * 32 + (i)] = (long long)__builtin_amdgcn_s_memtime();       \
  } while (0)
extern "C" int tg_debug_read_c3cw_stamps(long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(tg_c3cw_stamps), sizeof(long long) * n);
}
#else
#define C3_STAMP(i) do {} while (0)
#endif

// out-of-image patch positions (and the pitch padding) are DMA'd from here
__device__ __attribute__((aligned(16))) unsigned int tg_c3cw_zero_page[4];

namespace {

constexpr int kRow = 64, kPitch = 24, kPW = 18;
constexpr int kTH = 8, kPH = kTH + 2;                    // 8 x 16 output pixels per tile, 10 x 18 patch
constexpr int kChunkBytes = 16 * 1024;                   // 240 image rows of a 32-channel chunk, padded to 16 one-KiB blocks: block
                                                         // j = wave + 8 u of a buffer is row block wave + 8 (u & 1) of chunk u >> 1
constexpr int kBufBytes = 2 * kChunkBytes;               // 32 KB
constexpr int kStage = 2 * kBufBytes;                    // the workgroup's 72 KB of weights, staged once: [tap][chunk][64 rows][64 B]
constexpr int kStageBytes = 18 * 4096;
constexpr int kRed = kStage + kStageBytes;               // statistics: [2][64] fp32 sums of the eight waves + a ticket
constexpr int kLds = kRed + 1024;                        // 137 KB
constexpr int kDepth = 3;                                // fragment sets in flight (4: the statistics instantiations spill)

__device__ __forceinline__ int swz(int row, int piece) { return row * kRow + ((piece ^ ((row >> 1) & 2)) << 4); }

template <typename T> __device__ __forceinline__ f32x4 mma(bf16x8 a, bf16x8 b, f32x4 c) { return Mma16<T>::run(a, b, c); }

template <typename T> __device__ __forceinline__ void unpack8(const u32x4 r, float* v) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    v[2 * i] = bits16_to_f32<T>((unsigned short)(r[i] & 0xffffu));
    v[2 * i + 1] = bits16_to_f32<T>((unsigned short)(r[i] >> 16));
  }
}

struct C3K {
  const char* in;
  const char* w;
  const float* bias;
  const char* res;
  const char* mask;
  char* out;
  float* stats;
  const char* zero;
  int N, H, W, Cout;
  int tiles_x, tiles_y, ntiles;
  int flip, act, mask_mode, stats_groups, stats_mode, stats_replicas;
};

// SM: statistics of the stored values - 0 none, 1 per-channel sums (a bias gradient), 2 sums and sums of squares (batch norm).
// (arguments one by one, the ones a wave needs first in front: the first 16 dwords are preloaded into SGPRs with the wave)
template <int SM, typename T, int NCH = 2>   // NCH: 32-channel chunks of the reduction (2: 64 channels; 1: 32 - the discriminator's first layer)
__global__ __launch_bounds__(512) void conv3_cw_kernel(const char* a_in, const char* a_w, const char* a_zero, int a_H, int a_W, int a_Cout,
                                                       int a_tiles_x, int a_tiles_y, int a_ntiles, int a_flip, int a_N, char* a_out,
                                                       const char* a_mask, const char* a_res, const float* a_bias, float* a_stats,
                                                       int a_act, int a_mask_mode, int a_stats_groups, int a_stats_replicas) {
  C3K p;
  p.in = a_in; p.w = a_w; p.zero = a_zero; p.H = a_H; p.W = a_W; p.Cout = a_Cout; p.tiles_x = a_tiles_x; p.tiles_y = a_tiles_y;
  p.ntiles = a_ntiles; p.flip = a_flip; p.N = a_N; p.out = a_out; p.mask = a_mask; p.res = a_res; p.bias = a_bias; p.stats = a_stats;
  p.act = a_act; p.mask_mode = a_mask_mode; p.stats_groups = a_stats_groups; p.stats_replicas = a_stats_replicas;
  constexpr bool STATS = SM > 0;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int idx = lane & 15, g = lane >> 4;
  const int wc = wid & 1;              // channel half: packed rows 32 wc .. + 31
  const int r0 = (wid >> 1) * 2;       // first of the wave's 2 tile rows
  const int co_base = blockIdx.y * 64;
  const int ntl = (p.ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  C3_STAMP(0);

  // ---- the workgroup's 72 KB of weights, once: block (tap so, chunk ci) = 64 rows x 64 B; wave (u = wid % 4, ci = wid / 4) brings rows
  // 16 u .. + 15 of chunk ci for every tap; the lane's 16 bytes are physical piece lane % 4 of its row (swz is an involution)
  {
    const int u = wid & 3, ci = wid >> 2, row = 16 * u + (lane >> 2);
    const char* src = p.w + ((size_t)co_base + row) * 64 + (((lane & 3) ^ ((row >> 1) & 2)) << 4);
    if (ci < NCH) {   // (wave-uniform; one chunk: waves 0-3 bring it)
#pragma unroll
      for (int so = 0; so < 9; ++so) {
        const int slot = p.flip ? 8 - so : so;
        glds16(src + (size_t)(slot * NCH + ci) * p.Cout * 64, lds0 + kStage + (so * 2 + ci) * 4096 + u * 1024);
      }
    }
  }
  // ---- patch DMA: block j = wid + 8 u of a buffer = 16 rows (row block wid + 8 (u & 1)) of chunk u >> 1; the lane's 16 bytes: row
  // 16 rb + lane / 4, physical piece lane % 4 = logical piece ^ swizzle (the same for both row blocks: they are 128 rows apart).
  // Per lane and row block, the same for every tile: patch row / column - 1 (pitch padding and the 16 rows beyond the image get a
  // row outside every image) and the byte offset inside the pixel's chunk
  int dpy[2], dpx[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int row = (wid + 8 * e) * 16 + (lane >> 2);
    const int py = row / kPitch, px = row - py * kPitch;
    const bool valid = py < kPH && px < kPW;
    dpy[e] = valid ? py - 1 : -(1 << 20);
    dpx[e] = px - 1;
  }
  const int dof = ((lane & 3) ^ ((((wid * 16 + (lane >> 2))) >> 1) & 2)) * 16;
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
  auto dma_patch = [&](const Tile& tl, int buf) {   // asynchronous: vmcnt + barrier before anyone reads it.  EXACTLY 4 requests per wave
    const char* in_n = p.in + (size_t)tl.n * p.H * p.W * (NCH * 64);
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int iy = tl.ty0 + dpy[e], ix = tl.tx0 + dpx[e];
      const bool ok = ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);   // (no short-circuit branches)
      const char* src = in_n + (unsigned)((iy * p.W + ix) * (NCH * 64) + dof);             // (an image is < 4 GB)
      const unsigned dst = lds0 + buf * kBufBytes + (wid + 8 * e) * 1024;
      glds16(ok ? src : p.zero, dst);
      if constexpr (NCH == 2) glds16(ok ? src + 64 : p.zero, dst + kChunkBytes);
    }
  };
  int tile = (int)blockIdx.x;
  Tile cur = tile_of(tile);
  dma_patch(cur, 0);
  C3_STAMP(1);

  // fragment addresses of tile rows r0, r0 + 1 under column taps 0..2 inside one chunk image; row taps add multiples of the pitch
  int xb[2][3];
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int c = 0; c < 3; ++c) xb[b][c] = swz((r0 + b) * kPitch + idx + c, g);
  // the lane's 8 output channels: co_base + 32 wc + 8 g .. + 7 (two row-interleaved MFMA tiles, common.h)
  const int ch0 = co_base + wc * 32 + 8 * g;
  float bias_r[8];
#pragma unroll
  for (int e = 0; e < 8; e += 4) {
    f32x4 t = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) t = *reinterpret_cast<const f32x4*>(p.bias + ch0 + e);
    bias_r[e] = t[0]; bias_r[e + 1] = t[1]; bias_r[e + 2] = t[2]; bias_r[e + 3] = t[3];
  }
  const char* pre_src = p.mask_mode != TG_MASK_NONE ? p.mask : p.res;   // the epilogue's mask rows (without a mask: its residual rows)
  // launch-uniform epilogue forms (conv3_rw.hip): 0 forward (+ bias, activation)   1 input-gradient under a ReLU mask
  // 2 input-gradient + residual   3 anything else (every option tested at run time)
  const int emode = (p.mask_mode == TG_MASK_NONE && !p.res) ? 0
                    : (p.mask_mode == TG_MASK_RELU && !p.res && !p.bias && p.act == TG_ACT_NONE) ? 1
                    : (p.mask_mode == TG_MASK_NONE && p.res && !p.bias && p.act == TG_ACT_NONE) ? 2 : 3;
  const float act_slope = p.act == TG_ACT_RELU ? 0.f : p.act == TG_ACT_LRELU ? 0.2f : 1.f;   // activation as max(v, slope * v)
  float acc0[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc0[e] = (!STATS && emode == 0) ? bias_r[e] : 0.f;   // (the statistics instantiations have no registers to spare)

  // per-channel statistics of the stored values: lanes (the 16 pixels of a row) -> wave -> the eight waves through an LDS accumulator
  // -> ONE global atomic per channel and workgroup, issued by whichever wave arrives last (a ticket in LDS)
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
  float* const red = reinterpret_cast<float*>(smem + kRed);   // [2][64] sums, [128] the ticket
  if constexpr (STATS) {
    if (tid < 132) red[tid] = 0.f;   // published by the first barrier
  }
  auto flush_stats = [&](int grp) {
    if constexpr (STATS) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
#pragma unroll
        for (int m = 1; m < 16; m <<= 1) {
          s1[e] += __shfl_xor(s1[e], m);
          if constexpr (SM == 2) s2[e] += __shfl_xor(s2[e], m);
        }
      }
      if (idx == 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          atomicAdd(&red[wc * 32 + 8 * g + e], s1[e]);
          if constexpr (SM == 2) atomicAdd(&red[64 + wc * 32 + 8 * g + e], s2[e]);
        }
      }
      unsigned ticket = 0;
      if (lane == 0) ticket = atomicAdd(reinterpret_cast<unsigned*>(red + 128), 1u);
      ticket = __builtin_amdgcn_readfirstlane(ticket);
      if ((ticket & 7u) == 7u) {   // the last of the eight waves (a wave's LDS operations are served in order)
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

  // this wave's weight blocks are older than its 2 NCH patch blocks
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NCH) : "memory");
  lds_barrier();   // the weights are staged
  // A-fragments of packed rows 32 wc + 16 a + idx for 9 taps x 2 chunks
  bf16x8 wfr[NCH][9][2];
#pragma unroll
  for (int ci = 0; ci < NCH; ++ci)
#pragma unroll
    for (int so = 0; so < 9; ++so)
#pragma unroll
      for (int a = 0; a < 2; ++a)
        wfr[ci][so][a] = *reinterpret_cast<const bf16x8*>(smem + kStage + (so * 2 + ci) * 4096 + swz(wc * 32 + a * 16 + idx, g));
  C3_STAMP(2);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // patch 0 has landed

  for (int i = 0; i < ntl; ++i) {
    if (i < 6) C3_STAMP(4 + 4 * i + 0);
    lds_barrier();   // tile i's patch is in LDS (every wave waited for its own blocks), buffer (i + 1) & 1 is nobody's any more
    if (i < 6) C3_STAMP(4 + 4 * i + 1);
    const Tile mine = cur;
    if (i + 1 < ntl) {
      tile += (int)gridDim.x;
      cur = tile_of(tile);
      dma_patch(cur, (i + 1) & 1);
    }
    // this tile's mask / residual rows (behind the DMA in the in-order counter: when they have arrived, the next patch has)
    // (UNCONDITIONAL loads - a launch without rows reads the zero page: under `if (pre_src)` the compiler waited for them, and with them
    // for the patch requests, at the end of the branch, in front of the k-loop)
    u32x4 mk[2];
    {
      const char* src_n = pre_src + (size_t)mine.n * p.H * p.W * p.Cout * 2;
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int cy = min(mine.ty0 + r0 + b, p.H - 1), cx = min(mine.tx0 + idx, p.W - 1);   // clamped: unused outside the image
        const char* a = pre_src ? src_n + (unsigned)(((cy * p.W + cx) * p.Cout + ch0) * 2) : p.zero;
        mk[b] = *reinterpret_cast<const u32x4*>(a);
      }
    }
    const char* img = smem + (i & 1) * kBufBytes;
    f32x4 acc[2][2];   // the forward form starts from the bias (acc0: zeros in every other form)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[a][b] = f32x4{acc0[4 * a], acc0[4 * a + 1], acc0[4 * a + 2], acc0[4 * a + 3]};
    // k-loop: 18 steps (2 chunks x 9 taps) of 2 B-fragment reads + 4 MFMAs, kDepth - 1 steps' fragments in flight
    bf16x8 xf[kDepth][2];
    auto frags = [&](int s_, int buf) {   // compile-time arguments after unrolling
      const int ci = s_ / 9, so = s_ % 9;
#pragma unroll
      for (int b = 0; b < 2; ++b)
        xf[buf][b] = *reinterpret_cast<const bf16x8*>(img + ci * kChunkBytes + xb[b][so % 3] + (so / 3) * kPitch * kRow);
    };
#pragma unroll
    for (int s_ = 0; s_ < kDepth - 1; ++s_) frags(s_, s_);
#pragma unroll
    for (int s_ = 0; s_ < 9 * NCH; ++s_) {
      if (s_ + kDepth - 1 < 9 * NCH) frags(s_ + kDepth - 1, (s_ + kDepth - 1) % kDepth);
#ifndef C3_NO_SCHED_BARRIER   // (left to itself the compiler reads a fragment pair, waits for it, issues two MFMAs: one step in flight)
      __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int a = 0; a < 2; ++a) acc[a][b] = mma<T>(wfr[s_ / 9][s_ % 9][a], xf[s_ % kDepth][b], acc[a][b]);
#ifndef C3_NO_SCHED_BARRIER
      __builtin_amdgcn_sched_barrier(0);
#endif
    }
    if (i < 6) C3_STAMP(4 + 4 * i + 2);
    // ---- epilogue from the accumulators: pixel (ty0 + r0 + b, tx0 + idx), channels ch0 .. + 7
    if constexpr (STATS) {
      const int grp = mine.n / (p.N / p.stats_groups);
      if (grp != cur_grp) {
        if (cur_grp >= 0) flush_stats(cur_grp);
        cur_grp = grp;
      }
    }
    // the rows are taken over into registers the compiler does not connect with a load any more BEFORE the first store goes out: its wait
    // for the second row would otherwise also wait for the first row's store (the counter is in order; the stores are inline asm)
    u32x4 mkc[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      mkc[b] = mk[b];
      asm volatile("" : "+v"(mkc[b]));
    }
    char* const out_t = p.out + ((((size_t)mine.n * p.H + mine.ty0 + r0) * p.W + mine.tx0) * p.Cout + ch0) * 2;
    const bool okx = mine.tx0 + idx < p.W;
    auto finish = [&](auto MODE) {
      constexpr int M = decltype(MODE)::value;
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const bool ok = okx && mine.ty0 + r0 + b < p.H;
        float v[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = acc[0][b][e];
          v[4 + e] = acc[1][b][e];
        }
        if constexpr (M < 0) {    // (C3_DIAG_NOEPI)
        } else if constexpr (M == 0) {   // (the bias is in the accumulators already; the vector instructions of an epilogue are not hidden by
          if constexpr (STATS) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += bias_r[e];
          }
          if (p.act == TG_ACT_RELU) {   //  the matrix stream: one max instead of multiply + max per value where the activation allows)
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
          } else if (p.act == TG_ACT_LRELU) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.2f * v[e]);
          }
        } else if constexpr (M == 1) {
          // mask value > 0 on its 16-bit pattern (sign clear, not zero): the low half as the sign of word << 16, the high half as word > 0xffff
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int w_ = (int)mkc[b][e];
            v[2 * e] = (int)((unsigned)w_ << 16) > 0 ? v[2 * e] : 0.f;
            v[2 * e + 1] = w_ > 0xffff ? v[2 * e + 1] : 0.f;
          }
        } else if constexpr (M == 2) {
          float r[8];
          unpack8<T>(mkc[b], r);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += r[e];
        } else {
          if (p.bias) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += bias_r[e];
          }
          if (p.res) {
            float r[8];
            if (p.mask_mode == TG_MASK_NONE) unpack8<T>(mkc[b], r);
            else if (ok) Vec<T>::load(p.res + ((((size_t)mine.n * p.H + mine.ty0 + r0 + b) * p.W + mine.tx0 + idx) * p.Cout + ch0) * 2, r);   // (res AND mask)
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
              const int w_ = (int)mkc[b][e];
              v[2 * e] = (int)((unsigned)w_ << 16) > 0 ? v[2 * e] : neg * v[2 * e];
              v[2 * e + 1] = w_ > 0xffff ? v[2 * e + 1] : neg * v[2 * e + 1];
            }
          }
        }
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = pack2<T>(v[2 * e], v[2 * e + 1]);
        if constexpr (STATS) {
          if (ok) {   // (of the fp32 values, as conv3_rw.hip)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              s1[e] += v[e];
              if constexpr (SM == 2) s2[e] += v[e] * v[e];
            }
          }
        }
        if (ok) tg_store16(out_t + (unsigned)((b * p.W + idx) * p.Cout) * 2u, o);
      }
    };
#ifdef C3_DIAG_NOEPI   // diagnostic (timing only, results wrong): no epilogue arithmetic - what would fewer vector instructions buy?
    if (emode < 0) finish(std::integral_constant<int, 0>{});
    else finish(std::integral_constant<int, -1>{});
#else
    if (emode == 0) finish(std::integral_constant<int, 0>{});
    else if (emode == 1) finish(std::integral_constant<int, 1>{});
    else if (emode == 2) finish(std::integral_constant<int, 2>{});
    else finish(std::integral_constant<int, 3>{});
#endif
    // (the next tile's patch has landed: it was requested before this tile's rows, and those have been waited for - the counter is in order)
    if (i < 6) C3_STAMP(4 + 4 * i + 3);
  }
  if constexpr (STATS) {
    // a barrier between the loop's last flush and this one: when a workgroup's LAST tile is the first of a new statistics group, that
    // tile's epilogue has just flushed the old group, and without the barrier a fast wave could add its new-group sums into `red` and
    // take a second ticket before a slow wave has contributed to (or the last wave has read and zeroed) the first flush
    lds_barrier();
    if (cur_grp >= 0) flush_stats(cur_grp);
  }
  C3_STAMP(28);
}

template <int SM, typename T, int NCH = 2>
int launch_c3cw(const C3K& k, dim3 grid, hipStream_t st) {
  auto fn = conv3_cw_kernel<SM, T, NCH>;
  static std::atomic<bool> attr_done{false};  // one-time function attribute (benign race: idempotent)
  if (!attr_done) {
    TG_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, kLds));
    attr_done = true;
  }
  hipLaunchKernelGGL(fn, grid, dim3(512), kLds, st, k.in, k.w, k.zero, k.H, k.W, k.Cout, k.tiles_x, k.tiles_y, k.ntiles, k.flip, k.N, k.out,
                     k.mask, k.res, k.bias, k.stats, k.act, k.mask_mode, k.stats_groups, k.stats_replicas);
  return tg_launch_status();
}

}  // namespace

extern "C" int tg_conv3x3_cw(int dtype, const void* in, const void* w_packed, const float* bias, const void* res,
                             const void* mask, void* out, float* stats, int N, int H, int W, int Cin, int Cout, int flip,
                             int act, int mask_mode, int stats_mode, int stats_groups, int stats_replicas, int max_workgroups,
                             void* stream) {
  if (!in || !w_packed || !out || N <= 0 || H <= 0 || W <= 0) return TG_E_BADARG;
  if ((dtype != TG_BF16 && dtype != TG_F16) || (Cin != 64 && Cin != 32) || Cout <= 0 || Cout % 64) return TG_E_UNSUPPORTED;
  if (Cin == 32 && stats) return TG_E_UNSUPPORTED;   // (the 32-channel form is built without statistics: the discriminator's first layer has none)
  if (act != TG_ACT_NONE && act != TG_ACT_RELU && act != TG_ACT_LRELU) return TG_E_UNSUPPORTED;
  if (mask_mode != TG_MASK_NONE && !mask) return TG_E_BADARG;
  if (stats && (stats_groups <= 0 || N % stats_groups || stats_mode < 1 || stats_mode > 2)) return TG_E_BADARG;
  if (stats && (stats_replicas < 1 || (stats_replicas & (stats_replicas - 1)))) return TG_E_BADARG;  // a power of two
  if (!tg_aligned16(in) || !tg_aligned16(w_packed) || !tg_aligned16(out) || (res && !tg_aligned16(res)) ||
      (mask && !tg_aligned16(mask)) || (bias && !tg_aligned16(bias)))
    return TG_E_ALIGN;
  if ((long long)H * W * Cout * 2 >= 0x7fffffffLL) return TG_E_UNSUPPORTED;   // 32-bit offsets inside an image
  static const char* zero_page = [] {
    void* z = nullptr;
    return hipGetSymbolAddress(&z, HIP_SYMBOL(tg_c3cw_zero_page)) == hipSuccess ? (const char*)z : (const char*)nullptr;
  }();
  if (!zero_page) return TG_E_BADARG;
  C3K k;
  k.in = (const char*)in; k.w = (const char*)w_packed; k.bias = bias; k.res = (const char*)res;
  k.mask = (const char*)mask; k.out = (char*)out; k.stats = stats; k.zero = zero_page;
  k.N = N; k.H = H; k.W = W; k.Cout = Cout;
  k.tiles_x = (W + 15) / 16; k.tiles_y = (H + kTH - 1) / kTH;
  const long long nt = (long long)k.tiles_x * k.tiles_y * N;
  if (nt > 0x7fffffffLL) return TG_E_UNSUPPORTED;
  k.ntiles = (int)nt;
  k.flip = flip ? 1 : 0; k.act = act; k.mask_mode = mask ? mask_mode : TG_MASK_NONE;
  k.stats_groups = stats ? stats_groups : 1;
  k.stats_mode = stats ? stats_mode : 0;
  k.stats_replicas = stats ? stats_replicas : 1;
  // persistent grid: the cap's workgroups shared by the Cout/64 channel tiles, pixel tiles dealt evenly
  const int co_tiles = Cout / 64;
  const int cap = max_workgroups > 0 ? max_workgroups : 256;
  const int per = cap / co_tiles > 0 ? cap / co_tiles : 1;
  const int rounds = (k.ntiles + per - 1) / per;
  const int gx = (k.ntiles + rounds - 1) / rounds;
  dim3 grid((unsigned)gx, (unsigned)co_tiles);
  hipStream_t st = (hipStream_t)stream;
  const int sm = k.stats_mode;
#define C3_GO(TAG) (sm == 2 ? launch_c3cw<2, TAG>(k, grid, st) : sm == 1 ? launch_c3cw<1, TAG>(k, grid, st) : launch_c3cw<0, TAG>(k, grid, st))
  if (Cin == 32) return dtype == TG_F16 ? launch_c3cw<0, F16, 1>(k, grid, st) : launch_c3cw<0, BF16, 1>(k, grid, st);
  if (dtype == TG_F16) return C3_GO(F16);
  return C3_GO(BF16);
#undef C3_GO
}
