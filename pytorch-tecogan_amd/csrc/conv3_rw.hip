// 3x3 stride-1 convolution (forward, or input-gradient = taps mirrored) for the DENSE launches of the step - the batched
// generator backward (40 samples), the discriminator's residual stages (12 samples) - with the weights held in REGISTERS
// for the lifetime of a persistent workgroup:
//
//     out[n, y, x, co] = epilogue( sum_{tap, ci} W[tap][co][ci] * in[n, y + dy, x + dx, ci] )      bf16, NHWC
//
// Why another conv kernel.  conv_mfma.hip stages, per 64-channel x 128-pixel output tile, 23 KB of activations AND the
// 73 KB of weights of a 64->64 layer into LDS, then reads both operands back from LDS: (i) every workgroup pays the weight
// staging again (320-5120 times per launch), serialised in front of ~1 us of MFMAs; (ii) with 2x2 MFMA tiles per wave the
// fragment reads of both operands need 256 B/clk - the whole LDS bandwidth (MI355X_MICROARCH.md, LDS) - so the matrix
// pipe cannot be kept busy.  Here
//   * a workgroup is PERSISTENT: it walks pixel tiles blockIdx.x, blockIdx.x + gridDim.x, ... of its 64 output channels;
//   * each of its 8 waves loads the A-fragments it needs - 32 output channels x 9 taps x 64 input channels = 36 fragments,
//     144 VGPRs - ONCE; the k-loop feeds them to the MFMAs straight from registers, so LDS serves only the activation
//     fragments (half the LDS traffic per MFMA);
//   * the activation patch of tile i+1 is fetched into registers while tile i computes and written to the other of two LDS
//     patch buffers afterwards: one barrier per tile, the fetch latency is hidden behind a whole tile of MFMAs;
//   * two waves per SIMD (8 waves: 2 channel halves x 4 row pairs), so one wave's LDS reads / epilogue overlap its
//     partner's MFMAs (MI355X_MICROARCH.md, 'Two waves per SIMD').
// Cin = 128 (NCH = 4) splits K over the two waves of a pair (input-channel halves), which add their accumulators through
// an LDS exchange buffer - the split-K scheme of resblock.hip - so that the weight registers stay at 144 per wave.
// LDS image and packed-weight layout are those of conv_mfma.hip's pipelined path (swizzled 64-byte rows, patch pitch 24).
//
// Replaces aten::conv2d / convolution_backward(input) of the 3x3 layers (code/models.py:54-58,68,73-76,102 via
// code/ops.py:57-63; autograd of code/train.py:336,340) where tg_conv is the general entry point.
#include "common.h"
#include <type_traits>

namespace {

constexpr int kRow = 64, kPitch = 24;
constexpr int kTH = 8;                                // output rows per tile (x 16 columns)
constexpr int kPH = kTH + 2, kPW = 18;                // patch pixels
constexpr int kChunkBytes = kPH * kPitch * kRow;      // one 32-channel chunk of the patch: 15360 B
constexpr int kXchgBytes = 8 * 4 * 64 * 16;           // NCH = 4: [wave][slot][lane][16 B]
constexpr int kRedBytes = 4 * 2 * 64 * 4;             // statistics scratch [wave group][sum, sumsq][64 channels]

__device__ __forceinline__ int swz(int row, int piece) { return row * kRow + ((piece ^ ((row >> 1) & 2)) << 4); }

struct RwK {
  const char* in;
  const char* w;
  const float* bias;
  const char* res;
  const char* mask;
  char* out;
  float* stats;
  int N, H, W, Cout;
  int tiles_x, tiles_y, ntiles;
  int flip, act, mask_mode, stats_groups, stats_mode, stats_replicas;
};

template <typename T> __device__ __forceinline__ f32x4 mma(bf16x8 a, bf16x8 b, f32x4 c) { return Mma16<T>::run(a, b, c); }

// The activation patch of the NEXT tile goes straight from global memory into the other LDS buffer (global_load_lds_dwordx4, as in
// wgrad_group.hip): no staging registers (24 per thread at Cin = 128, where the wave's 144 weight registers already spill) and no
// ds_write pass.  One wave-instruction fills 1 KiB = 16 rows of a chunk image; the swizzle moves to the per-lane SOURCE address,
// patch positions outside the image (and the 6 pitch-padding rows per patch row) read a page of zeros.  -DRW_NO_DMA: the
// register-staged path of round 2 (A/B builds).
#ifndef RW_NO_DMA
#define RW_DMA 1
#else
#define RW_DMA 0
#endif
__device__ __attribute__((aligned(16))) unsigned int tg_rw_zero_page[4];
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_dst)
      : "memory");
}

// LDS-only barrier: __syncthreads() would also wait for the epilogue's global stores and the next tile's patch loads
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <typename T> __device__ __forceinline__ void unpack8(const u32x4 r, float* v) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    v[2 * i] = bits16_to_f32<T>((unsigned short)(r[i] & 0xffffu));
    v[2 * i + 1] = bits16_to_f32<T>((unsigned short)(r[i] >> 16));
  }
}

// SM: statistics of the stored values - 0 none, 1 per-channel sums (a bias gradient), 2 sums and sums of squares (batch norm).
// EARLY: the epilogue's mask (else residual) rows are fetched BEFORE the k-loop (8 registers live across it) instead of
// behind it: a persistent workgroup walks ~40 tiles of ~1 us of MFMAs each, and a cold 16-byte-per-lane read behind every one
// of them was a full HBM round trip in which all eight waves sat (c6's input-gradient, 128x128 x 40 samples: 164 us with the
// relu mask against 106 without).  Off where the registers are not there (Cin = 128 spills already: 8 more registers cost
// the discriminator's stage-2 launches more than the fetch saved).
template <int NCH, int SM, typename T = BF16>
__global__ __launch_bounds__(512) void conv3_rw_kernel(const RwK p) {
  constexpr bool STATS = SM > 0;
#ifndef RW_EARLY
#define RW_EARLY (NCH == 2)
#endif
  constexpr bool EARLY = RW_EARLY;
  // DMA staging where the registers are short (Cin = 128: 21-59 spilled VGPRs -> 0-28, c32's input-gradient 112 -> 78 us);
  // Cin = 64 keeps the register-staged patch (no spills there, and ~1 us per launch faster: tools/mb_rw.py)
  constexpr bool kDma = RW_DMA && NCH == 4;
  // Cin = 128 with statistics: the 16 per-lane accumulators would be live across the k-loop on top of 144 weight registers (28
  // VGPRs spilled).  There the sums of each tile are reduced over the wave right behind its stores and added into an LDS
  // accumulator; the flush reads that.
  constexpr bool kLdsStats = STATS && NCH == 4;
  static_assert(NCH == 2 || NCH == 4, "Cin = 64 or 128");
  constexpr int PT = NCH == 2 ? 2 : 4;          // output rows per wave in the k-loop
  constexpr int PF = 2;                         // output rows per wave in the epilogue (NCH = 4: half of PT after the exchange)
  constexpr int NLD = (NCH * kPH * kPW * 4 + 511) / 512;  // patch pieces (16 B) per thread: 3 or 6
  constexpr int kBufBytes = NCH * kChunkBytes;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* lds_x = smem + 2 * kBufBytes;                                  // exchange (NCH = 4 only)
  float* red = reinterpret_cast<float*>(smem + 2 * kBufBytes + (NCH == 4 ? kXchgBytes : 0));

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int idx = lane & 15, g = lane >> 4;
  const int wc = wid & 1;                                   // channel half: packed rows 32*wc .. 32*wc + 31
  const int kh = NCH == 4 ? (wid >> 1) & 1 : 0;             // input-channel half (split-K)
  const int wp = NCH == 4 ? wid >> 2 : wid >> 1;            // row group
  const int r0 = wp * PT;
  const int co_base = blockIdx.y * 64;
  const size_t pix_bytes = (size_t)NCH * 64;

  // ---- per-thread patch pieces: coordinates inside the patch are the same for every tile.  One packed word per piece
  // (py | px << 4 | (chunk*4 + piece) << 9 | LDS offset/16 << 13 | in-range << 25): registers are what this kernel is short of
  constexpr int NDI = (NCH * (kChunkBytes / 1024) + 7) / 8;   // DMA wave-instructions per wave and tile (15 KiB per chunk image)
  unsigned dinfo[kDma ? NDI : 1];  // instruction wid + 8 u: py | px << 4 | byte offset inside the pixel << 9 | valid << 25
  if constexpr (kDma) {
#pragma unroll
    for (int u = 0; u < NDI; ++u) {
      const int j = wid + 8 * u;                       // 1-KiB block j of the buffer: chunk j / 15, rows 16 (j % 15) + lane / 4
      const int cc = j / (kChunkBytes / 1024), row = (j - cc * (kChunkBytes / 1024)) * 16 + (lane >> 2);
      const int py = row / kPitch, px = row - py * kPitch;
      const int piece = (lane & 3) ^ ((row >> 1) & 2);  // the logical piece this physical slot holds (swz is an involution)
      const bool valid = j < NCH * (kChunkBytes / 1024) && px < kPW;
      dinfo[u] = (unsigned)py | ((unsigned)px << 4) | ((unsigned)(cc * 64 + piece * 16) << 9) | ((valid ? 1u : 0u) << 25);
    }
  }
  unsigned pinfo[kDma ? 1 : NLD];
#pragma unroll
  for (int u = 0; u < (kDma ? 0 : NLD); ++u) {
    const int i = tid + u * 512;
    const bool in_range = i < NCH * kPH * kPW * 4;
    const int ic = in_range ? i : NCH * kPH * kPW * 4 - 1;
    const int s = ic & 3, r = ic >> 2;
    const int cc = r / (kPH * kPW), prow = r - cc * (kPH * kPW);
    const int py = prow / kPW, px = prow - py * kPW;
    const int dst = cc * kChunkBytes + swz(py * kPitch + px, s);
    pinfo[u] = (unsigned)py | ((unsigned)px << 4) | ((unsigned)(cc * 4 + s) << 9) | ((unsigned)(dst >> 4) << 13) |
               ((in_range ? 1u : 0u) << 25);
  }
  u32x4 va[kDma ? 1 : NLD];
  unsigned vok = 0;  // bit u: piece u of the patch in flight lies inside the image
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  const int wid_u = wid;
  auto decode = [&](int t, int& n, int& ty0, int& tx0) {
    const int txb = t % p.tiles_x;
    t /= p.tiles_x;
    const int tyb = t % p.tiles_y;
    n = t / p.tiles_y;
    ty0 = tyb * kTH;
    tx0 = txb * 16;
  };
  // loads are unconditional from a clamped address and zeroed at the LDS store: a load under a divergent `if` makes the
  // compiler wait for each one before it issues the next
  // DMA form: the whole patch of tile t into buffer `buf` (asynchronous: s_waitcnt vmcnt(0) + barrier before it is read)
  auto dma_patch = [&](int t, int buf) {
    int n, ty0, tx0;
    decode(t, n, ty0, tx0);
    const char* in_n = p.in + (size_t)n * p.H * p.W * pix_bytes;
    const char* zero = reinterpret_cast<const char*>(tg_rw_zero_page);
#pragma unroll
    for (int u = 0; u < NDI; ++u) {
      if (wid_u + 8 * u < NCH * (kChunkBytes / 1024)) {  // wave-uniform
        const int iy = ty0 - 1 + (int)(dinfo[u] & 15u), ix = tx0 - 1 + (int)((dinfo[u] >> 4) & 31u);
        const bool ok = (dinfo[u] >> 25) && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        const char* src = in_n + ((long long)iy * p.W + ix) * (long long)pix_bytes + ((dinfo[u] >> 9) & 0xffffu);
        glds16(ok ? src : zero, lds0 + buf * kBufBytes + (wid_u + 8 * u) * 1024);
      }
    }
  };
  auto issue_patch = [&](int t) {
    int n, ty0, tx0;
    decode(t, n, ty0, tx0);
    const char* in_n = p.in + (size_t)n * p.H * p.W * pix_bytes;
    vok = 0;
#pragma unroll
    for (int u = 0; u < (kDma ? 0 : NLD); ++u) {
      const int iy = ty0 - 1 + (int)(pinfo[u] & 15u), ix = tx0 - 1 + (int)((pinfo[u] >> 4) & 31u);
      vok |= (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) ? (1u << u) : 0u;
      const int cy = min(max(iy, 0), p.H - 1), cx = min(max(ix, 0), p.W - 1);
      va[u] = *reinterpret_cast<const u32x4*>(in_n + ((size_t)cy * p.W + cx) * pix_bytes + ((pinfo[u] >> 9) & 15u) * 16);
    }
  };
  auto store_patch = [&](int buf) {
    char* dst = smem + buf * kBufBytes;
#pragma unroll
    for (int u = 0; u < (kDma ? 0 : NLD); ++u)
      if (pinfo[u] >> 25)
        *reinterpret_cast<u32x4*>(dst + ((pinfo[u] >> 13) & 4095u) * 16) = ((vok >> u) & 1u) ? va[u] : u32x4{0u, 0u, 0u, 0u};
  };

  int tile = blockIdx.x;
  if constexpr (kDma) {
    if (tile < p.ntiles) dma_patch(tile, 0);
  } else {
    if (tile < p.ntiles) issue_patch(tile);
  }

  // ---- the wave's weights: A-fragments of packed rows 32*wc + 16*a + idx for 9 taps x 2 chunks, in k-loop order.
  // Packed image [tap][chunk][Cout rows][64 B]; the input-gradient launch pairs spatial offset `so` with slot 8 - so.
  bf16x8 wfr[2][9][2];
  {
    const char* wl = p.w + ((size_t)co_base + wc * 32 + idx) * 64 + g * 16;
#pragma unroll
    for (int ci = 0; ci < 2; ++ci)
#pragma unroll
      for (int so = 0; so < 9; ++so)
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          const int slot = p.flip ? 8 - so : so;
          const int chunk = kh * 2 + ci;
          wfr[ci][so][a] = *reinterpret_cast<const bf16x8*>(wl + ((size_t)(slot * NCH + chunk) * p.Cout + a * 16) * 64);
        }
  }
  // lane (idx, g) ends up with channels ch0 .. ch0 + 7 of pixel idx (two row-interleaved MFMA tiles, common.h)
  const int ch0 = co_base + wc * 32 + 8 * g;
  float bias_r[8];
#pragma unroll
  for (int e = 0; e < 8; e += 4) {
    f32x4 t = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) t = *reinterpret_cast<const f32x4*>(p.bias + ch0 + e);
    bias_r[e] = t[0]; bias_r[e + 1] = t[1]; bias_r[e + 2] = t[2]; bias_r[e + 3] = t[3];
  }
  // fragment addresses of the wave's rows inside one chunk image (column taps 0..2); row taps add multiples of the pitch
  // (two 16-bit offsets per register)
  unsigned xbase[PT * 3 / 2];
#pragma unroll
  for (int q = 0; q < PT * 3; q += 2)
    xbase[q / 2] = (unsigned)swz((r0 + q / 3) * kPitch + idx + q % 3, g) |
                   ((unsigned)swz((r0 + (q + 1) / 3) * kPitch + idx + (q + 1) % 3, g) << 16);
  auto xoff = [&](int b, int c) {  // compile-time (b, c) after unrolling
    const int q = b * 3 + c;
    return (int)((q & 1) ? xbase[q / 2] >> 16 : xbase[q / 2] & 0xffffu);
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
  if constexpr (kLdsStats) {
    if (tid < 128) red[tid] = 0.f;   // [sum, sum of squares][64 channels]; published by the barrier behind the first patch
  }
  // per-channel statistics of the stored values: lanes -> wave -> workgroup -> one atomic per channel (uniform control flow)
  auto flush_stats = [&](int grp) {
    if constexpr (kLdsStats) {
      lds_barrier();   // every wave's tile sums are in
      if (tid < 64 * SM) {
        const int which = tid >> 6, chn = tid & 63;
        const size_t rep = (size_t)(blockIdx.x & (p.stats_replicas - 1)) * p.stats_groups * 2 * p.Cout;
        atomicAdd(p.stats + rep + ((size_t)grp * 2 + which) * p.Cout + co_base + chn, red[which * 64 + chn]);
        red[which * 64 + chn] = 0.f;
      }
      lds_barrier();
    } else if constexpr (STATS) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
#pragma unroll
        for (int m = 1; m < 16; m <<= 1) {
          s1[e] += __shfl_xor(s1[e], m);
          if constexpr (SM == 2) s2[e] += __shfl_xor(s2[e], m);
        }
      }
      const int slot = NCH == 4 ? wp * 2 + kh : wp;  // the four waves that share a channel half
      if (idx == 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          red[(slot * 2 + 0) * 64 + wc * 32 + 8 * g + e] = s1[e];
          if constexpr (SM == 2) red[(slot * 2 + 1) * 64 + wc * 32 + 8 * g + e] = s2[e];
        }
      }
      lds_barrier();
      if (tid < 64 * SM) {  // SM = 1: sums only ([groups][2][Cout] layout either way)
        const int which = tid >> 6, chn = tid & 63;
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) s += red[(w * 2 + which) * 64 + chn];
        const size_t rep = (size_t)(blockIdx.x & (p.stats_replicas - 1)) * p.stats_groups * 2 * p.Cout;
        atomicAdd(p.stats + rep + ((size_t)grp * 2 + which) * p.Cout + co_base + chn, s);
      }
      lds_barrier();
#pragma unroll
      for (int e = 0; e < 8; ++e) s1[e] = 0.f;
      if constexpr (SM == 2) {
#pragma unroll
        for (int e = 0; e < 8; ++e) s2[e] = 0.f;
      }
    }
  };

  if constexpr (kDma) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the first patch (and the weights) have landed
  else if (tile < p.ntiles) store_patch(0);
  lds_barrier();

  for (int buf = 0; tile < p.ntiles; tile += gridDim.x, buf ^= 1) {
    const int next = tile + gridDim.x;
    int n, ty0, tx0;
    decode(tile, n, ty0, tx0);
    if constexpr (kDma) {
      if (next < p.ntiles) dma_patch(next, buf ^ 1);  // lands in the other buffer during this tile's MFMAs
    } else {
      if (next < p.ntiles) issue_patch(next);  // in flight during this tile's MFMAs
    }
    // the epilogue's mask rows (a launch without a mask: its residual rows) of the two output rows this wave finalises
    const int frow0 = NCH == 4 ? r0 + 2 * kh : r0;
    const char* pre_src = p.mask_mode != TG_MASK_NONE ? p.mask : p.res;
    u32x4 pre[PF];
    auto issue_pre = [&]() {
      if (pre_src) {
#pragma unroll
        for (int j = 0; j < PF; ++j) {
          const int cy = min(ty0 + frow0 + j, p.H - 1), cx = min(tx0 + idx, p.W - 1);  // clamped: unused outside the image
          pre[j] = *reinterpret_cast<const u32x4*>(pre_src + ((((size_t)n * p.H + cy) * p.W + cx) * p.Cout + ch0) * 2);
        }
      }
    };
    if constexpr (EARLY) issue_pre();

    // ---- k-loop: 2 chunks x 9 taps, weights from registers, activation fragments from the patch image
    f32x4 acc[2][PT];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < PT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    const char* img = smem + buf * kBufBytes + kh * 2 * kChunkBytes;
#pragma unroll
    for (int ci = 0; ci < 2; ++ci) {
#pragma unroll
      for (int so = 0; so < 9; ++so) {
#pragma unroll
        for (int b = 0; b < PT; ++b) {
          const bf16x8 xf = *reinterpret_cast<const bf16x8*>(img + ci * kChunkBytes + xoff(b, so % 3) + (so / 3) * kPitch * kRow);
#pragma unroll
          for (int a = 0; a < 2; ++a) acc[a][b] = mma<T>(wfr[ci][so][a], xf, acc[a][b]);
        }
      }
    }

    // DMA form: the next tile's pieces of this wave have landed by now (issued a whole k-loop ago); behind this point the only
    // vector-memory operations in flight are the epilogue's own, so the barrier at the end of the tile publishes the buffer
    if constexpr (kDma) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // ---- NCH = 4: the two K halves add their accumulators; each wave finalises two of the pair's four rows
    f32x4 fin[2][PF];
    if constexpr (NCH == 4) {
      char* myx = lds_x + (wid * 4 * 64 + lane) * 16;
      const char* px_ = lds_x + ((wid ^ 2) * 4 * 64 + lane) * 16;
      auto exchange = [&](auto KEEP) {
        constexpr int keep = decltype(KEEP)::value, give = 2 - keep;  // first row kept / handed over
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int j = 0; j < 2; ++j) *reinterpret_cast<f32x4*>(myx + (a * 2 + j) * 1024) = acc[a][give + j];
        lds_barrier();
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const f32x4 o = *reinterpret_cast<const f32x4*>(px_ + (a * 2 + j) * 1024);
            fin[a][j] = acc[a][keep + j] + o;
          }
      };
      if (kh == 0) exchange(std::integral_constant<int, 0>{});
      else exchange(std::integral_constant<int, 2>{});
    } else {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int j = 0; j < PF; ++j) fin[a][j] = acc[a][j];
    }

    // ---- epilogue: +bias, +res, act, *act'(mask), store, statistics (the order of conv_mfma.hip)
    if constexpr (STATS) {
      const int grp = n / (p.N / p.stats_groups);
      if (grp != cur_grp) {
        if (cur_grp >= 0) flush_stats(cur_grp);
        cur_grp = grp;
      }
    }
#pragma unroll
    for (int j = 0; j < PF; ++j) {
      const int cy = ty0 + frow0 + j, cx = tx0 + idx;
      if (cy < p.H && cx < p.W) {
        const size_t eoff = ((((size_t)n * p.H + cy) * p.W + cx) * p.Cout + ch0) * 2;
        float v[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = fin[0][j][e] + bias_r[e];
          v[4 + e] = fin[1][j][e] + bias_r[4 + e];
        }
        if (p.res) {
          float r[8];
          if (EARLY && p.mask_mode == TG_MASK_NONE) unpack8<T>(pre[j], r);
          else Vec<T>::load(p.res + eoff, r);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += r[e];
        }
        if (p.act == TG_ACT_RELU) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
        } else if (p.act == TG_ACT_LRELU) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : 0.2f * v[e];
        }
        if (p.mask_mode != TG_MASK_NONE) {
          float m[8];
          if constexpr (EARLY) unpack8<T>(pre[j], m);
          else Vec<T>::load(p.mask + eoff, m);
          const float neg = p.mask_mode == TG_MASK_LRELU ? 0.2f : 0.f;
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] *= (m[e] > 0.f ? 1.f : neg);
        }
        Vec<T>::store(p.out + eoff, v);
        if constexpr (STATS) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            s1[e] += v[e];
            if constexpr (SM == 2) s2[e] += v[e] * v[e];
          }
        }
      }
    }
    if constexpr (kLdsStats) {   // this tile's sums: over the 16 pixels of the wave's rows, then into the LDS accumulator
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
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        s1[e] = 0.f;
        if constexpr (SM == 2) s2[e] = 0.f;
      }
    }

    if constexpr (!kDma) {
      if (next < p.ntiles) store_patch(buf ^ 1);
    }
    lds_barrier();  // the other buffer is complete; everyone has finished reading this one (and the exchange slots)
  }
  if constexpr (STATS) {
    if (cur_grp >= 0) flush_stats(cur_grp);
  }
}

template <int NCH, int SM, typename T>
int launch_rw(const RwK& k, dim3 grid, hipStream_t st) {
  constexpr int lds = 2 * NCH * kChunkBytes + (NCH == 4 ? kXchgBytes : 0) + kRedBytes;
  auto fn = conv3_rw_kernel<NCH, SM, T>;
  static std::atomic<bool> attr_done{false};  // one-time function attribute (benign race: idempotent)
  if (!attr_done) {
    TG_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    attr_done = true;
  }
  hipLaunchKernelGGL(fn, grid, dim3(512), lds, st, k);
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
  if (stats && (stats_groups <= 0 || N % stats_groups || stats_mode < 1 || stats_mode > 2)) return TG_E_BADARG;
  if (stats && (stats_replicas < 1 || (stats_replicas & (stats_replicas - 1)))) return TG_E_BADARG;  // a power of two
  if (!tg_aligned16(in) || !tg_aligned16(w_packed) || !tg_aligned16(out) || (res && !tg_aligned16(res)) ||
      (mask && !tg_aligned16(mask)) || (bias && !tg_aligned16(bias)))
    return TG_E_ALIGN;
  RwK k;
  k.in = (const char*)in; k.w = (const char*)w_packed; k.bias = bias; k.res = (const char*)res;
  k.mask = (const char*)mask; k.out = (char*)out; k.stats = stats;
  k.N = N; k.H = H; k.W = W; k.Cout = Cout;
  k.tiles_x = (W + 15) / 16; k.tiles_y = (H + kTH - 1) / kTH;
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
