// Stride-2 gather convolutions with the weights in registers (round 6): the discriminator's 4x4 stride-2 padding-1 forward
// (code/models.py:90-94) and the input-gradient of the generator's k3 s2 conv-transposes (autograd of code/ops.py:45-54,
// code/models.py:72,74) - both are
//
//   out[n, y, x, co] = sum_{ky, kx in 0..KS-1} sum_ci  W[ky*KS + kx][co][ci] * in[n, 2y + ky - 1, 2x + kx - 1, ci]        KS = 4 / 3
//
// conv4s2_mfma.hip re-stages the KS*KS slots' weights (131 / 262 KB per 64 output channels) for every 8 x 16 tile: 112 TFLOP/s on the
// discriminator's four layers, 343 on the conv-transposes' input-gradients.  Here the workgroups are persistent over 4 x 16 OUTPUT
// tiles and the weights never move again: the reduction (taps x channels) is split between FOUR wave groups - a wave keeps
// (its quarter of the k-steps) x (32 of the workgroup's 64 output channels) as A-fragments in registers (64 VGPRs per 64 reduction
// channels for KS = 4) - and the four partial sums of a tile meet through an LDS exchange: wave (channel half, group kg) writes the three
// tile rows it does not own, and finishes row kg (sum in group order, + bias, statistics, one 16-byte store per lane).
//
// The patch: input rows 2 ty0 - 1 .. + 2*4 + KS - 3, columns 2 tx0 - 1 .. + 2*16 + KS - 3, by LDS-DMA.  Lanes of a B-fragment are 16
// consecutive OUTPUT pixels = input pixels two apart, so the patch's columns are DE-INTERLEAVED by parity into two sub-images per input
// row (column j = 2 jj + P at image row  i * 40 + 20 P + jj): a tap kx reads 16 consecutive image rows starting at 20 (kx & 1) + (kx >> 1),
// conflict-free under the usual piece swizzle at any base row (tools/lds_layout.py).  Reduction channels travel in PHASES of two
// 32-channel chunks (50 / 46 KB): a 64-channel layer is one phase per tile, a 128-channel layer two, the next phase's patch in
// flight during the current one's k-loop (two buffers).  LDS: 2 x 50 KB + 48 KB exchange.
//
// NWC (round 6, second form): a tile costs what its PATCH costs to take in - 50 KB per phase through the CU's ~25 B/clk, 2000 ticks for
// ~1150 of matrix work - and with 128 output channels as two 64-channel workgroups the patch was taken in twice.  Where the registers
// allow (KS^2 x Cin / 32 <= 36 k-steps: every layer but the 128 -> 128 4x4 one) ONE workgroup now makes all 128 channels of a tile:
// NWC = 4 channel quarters x 2 reduction halves instead of 2 halves x 4 quarters - half the patch bytes per output, a 2-way exchange.
#ifndef TG_ST_AUX
#define TG_ST_AUX "sc1"   // results are written THROUGH the L2 (common.h, tg_store16; profiles/r05_u_write_through_ab.log)
#endif
#include "rbw_common.h"
#include <atomic>
#include <type_traits>

// out-of-image patch positions (and the pitch padding) are DMA'd from here
__device__ __attribute__((aligned(16))) unsigned int tg_s2cw_zero_page[4];

namespace {

constexpr int kRow = 64, kPitch = 40, kOdd = 20;
constexpr int kTH = 4;                                   // output rows per tile (x 16 columns)
constexpr int kXBytes = 2 * 4 * 3 * 2048;                // exchange: [channel half][tile row][source slot][2 x 1 KiB]
static_assert(kPitch % 8 == 0, "row taps must be immediates: multiples of 8 image rows keep the swizzle key");

__device__ __forceinline__ int swz(int row, int piece) { return row * kRow + ((piece ^ ((row >> 1) & 2)) << 4); }

template <int KS> struct Geo {
  static constexpr int NT = KS * KS;
  static constexpr int PR = 2 * kTH + KS - 2;            // patch rows: 10 / 9
  static constexpr int kRows = PR * kPitch;              // image rows of a chunk: 400 / 360
  static constexpr int KB = (kRows + 15) / 16;           // 1-KiB blocks of a chunk image: 25 / 23
  static constexpr int kChunkBytes = KB * 1024;
  static constexpr int kPhaseBytes = 2 * kChunkBytes;
  static constexpr int NS = 2 * NT;                      // k-steps of a phase: step s = (tap s / 2, chunk s % 2)
  // wave group kg of NKG owns steps [first(kg), first(kg + 1)).  NKG = 4: 8 each (KS = 4); 5, 4, 4, 5 (KS = 3: groups kg and kg + 2 share
  // a SIMD).  NKG = 2: halves
  template <int NKG> static constexpr int first(int kg) {
    return NKG == 2 ? (NS / 2) * kg : KS == 4 ? 8 * kg : (kg == 0 ? 0 : kg == 1 ? 5 : kg == 2 ? 9 : kg == 3 ? 13 : 18);
  }
  static constexpr int kLds = 2 * kPhaseBytes + kXBytes + 2048;
};

template <typename T> __device__ __forceinline__ f32x4 mma(bf16x8 a, bf16x8 b, f32x4 c) { return Mma16<T>::run(a, b, c); }

struct S2K {
  const char* in;     // [N][IH][IW][NCH * 32]
  const char* w;      // packed [slot][chunk][Cout rows][64 B]
  const char* zero;
  char* out;          // [N][OH][OW][Cout]
  const float* bias;
  float* stats;
  int N, IH, IW, OH, OW, Cout, tiles_x, tiles_y, ntiles, stats_groups, stats_replicas;
};

// (arguments one by one: the first 16 dwords are preloaded into SGPRs with the wave - csrc/build.sh, -amdgpu-kernarg-preload-count)
template <int KS, int NCH, bool STATS, typename T, int NWC>
__global__ __launch_bounds__(512) void conv_s2_cw_kernel(const char* a_in, const char* a_w, const char* a_zero, int a_IH, int a_IW, int a_OH,
                                                         int a_OW, int a_Cout, int a_tiles_x, int a_tiles_y, int a_ntiles, char* a_out,
                                                         const float* a_bias, float* a_stats, int a_N, int a_groups, int a_replicas) {
  using GG = Geo<KS>;
  constexpr int NPH = NCH / 2, KB = GG::KB, kChunkBytes = GG::kChunkBytes, kPhaseBytes = GG::kPhaseBytes;
  constexpr int kX = 2 * kPhaseBytes, kRed = kX + kXBytes;
  S2K p;
  p.in = a_in; p.w = a_w; p.zero = a_zero; p.IH = a_IH; p.IW = a_IW; p.OH = a_OH; p.OW = a_OW; p.Cout = a_Cout; p.tiles_x = a_tiles_x;
  p.tiles_y = a_tiles_y; p.ntiles = a_ntiles; p.out = a_out; p.bias = a_bias; p.stats = a_stats; p.N = a_N; p.stats_groups = a_groups;
  p.stats_replicas = a_replicas;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int idx = lane & 15, g = lane >> 4;
  constexpr int NKG = 8 / NWC;                                // reduction groups: 4 (two channel halves) or 2 (four channel quarters)
  constexpr int RPW = 4 / NKG;                                // tile rows a wave finishes: kg, kg + NKG, ...
  const int wc = wid % NWC;                                   // channel group: packed rows 32 wc .. + 31 of the workgroup's 32 NWC
  const int kgr = wid / NWC;                                  // reduction group
  const int co_base = blockIdx.y * (32 * NWC);
  const int ntl = (p.ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  constexpr int pix_bytes = NCH * 64;

  // ---- LDS-DMA: block rb = 16 image rows of a chunk; the lane's 16 bytes: image row 16 rb + lane / 4, physical piece lane % 4 = logical
  // piece ^ swizzle.  Image row r = patch row r / 40, sub-image column r % 40: parity P = (r % 40 >= 20), jj = r % 40 - 20 P (jj >= 17: pitch
  // padding, the zero page) = input pixel (2 ty0 - 1 + r / 40, 2 tx0 - 1 + 2 jj + P).  Wave w brings blocks w, w + 8, w + 16 of both
  // chunks of a phase; KS = 4: waves 0 / 1 also block 24 of chunk 0 / 1.
  constexpr int NB = KS == 4 ? 4 : 3;
  int dpy[NB], dpx[NB];
  const int dof = ((lane & 3) ^ ((lane >> 3) & 2)) * 16;      // ((row >> 1) & 2 with row = 16 rb + lane / 4)
#pragma unroll
  for (int e = 0; e < NB; ++e) {
    const int rb = e < 3 ? wid + 8 * e : 24;
    const int row = rb * 16 + (lane >> 2);
    const int py = row / kPitch, c = row - py * kPitch;
    const int par = c >= kOdd ? 1 : 0, jj = c - kOdd * par;
    const bool in_img = (rb < KB) & (py < GG::PR) & (jj < 17);
    dpy[e] = in_img ? py - 1 : -(1 << 20);
    dpx[e] = 2 * jj + par - 1;
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
  auto dma_phase = [&](const Tile& tl, int ph, int buf) {   // asynchronous: vmcnt + barrier before anyone reads it
    const char* in_n = p.in + (size_t)tl.n * p.IH * p.IW * pix_bytes + ph * 128;
    const unsigned dst = lds0 + buf * kPhaseBytes;
#pragma unroll
    for (int e = 0; e < 3; ++e) {
      if (wid + 8 * e < KB) {   // wave-uniform (KS = 3: wave 7 has two blocks per chunk)
        const int iy = 2 * tl.ty0 + dpy[e], ix = 2 * tl.tx0 + dpx[e];
        const bool ok = ((unsigned)iy < (unsigned)p.IH) & ((unsigned)ix < (unsigned)p.IW);
        const char* src = in_n + (unsigned)((iy * p.IW + ix) * pix_bytes + dof);
        glds16(ok ? src : p.zero, dst + (wid + 8 * e) * 1024);
        glds16(ok ? src + 64 : p.zero, dst + kChunkBytes + (wid + 8 * e) * 1024);
      }
    }
    if constexpr (KS == 4) {
      if (wid < 2) {            // block 24 of chunk wid
        const int iy = 2 * tl.ty0 + dpy[NB - 1], ix = 2 * tl.tx0 + dpx[NB - 1];
        const bool ok = ((unsigned)iy < (unsigned)p.IH) & ((unsigned)ix < (unsigned)p.IW);
        const char* src = in_n + (unsigned)((iy * p.IW + ix) * pix_bytes + dof) + wid * 64;
        glds16(ok ? src : p.zero, dst + wid * kChunkBytes + 24 * 1024);
      }
    }
  };
  int tile = (int)blockIdx.x;
  Tile cur = tile_of(tile);
  dma_phase(cur, 0, 0);

  // fragment addresses under column tap kx inside one chunk image: 16 consecutive image rows from 20 (kx & 1) + (kx >> 1); tile row b and
  // row tap ky add (2 b + ky) * 40 image rows (an immediate)
  int xa[KS];
#pragma unroll
  for (int kx = 0; kx < KS; ++kx) xa[kx] = swz(kOdd * (kx & 1) + (kx >> 1) + idx, g);
  // the lane's 8 output channels: co_base + 32 wc + 8 g .. + 7 (two row-interleaved MFMA tiles, common.h)
  const int ch0 = co_base + wc * 32 + 8 * g;
  float bias_r[8];
#pragma unroll
  for (int e = 0; e < 8; e += 4) {
    f32x4 t = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) t = *reinterpret_cast<const f32x4*>(p.bias + ch0 + e);
    bias_r[e] = t[0]; bias_r[e + 1] = t[1]; bias_r[e + 2] = t[2]; bias_r[e + 3] = t[3];
  }

  // per-channel statistics of the finished values (fp32, before the 16-bit store - as conv4s2_mfma.hip): lanes -> wave -> the eight waves
  // through an LDS accumulator -> ONE global atomic per channel and workgroup, issued by whichever wave arrives last (conv3_cw.hip)
  float s1[STATS ? 8 : 1], s2[STATS ? 8 : 1];
  int cur_grp = -1;
  constexpr int CW = 32 * NWC;                                // the workgroup's output channels
  float* const red = reinterpret_cast<float*>(smem + kRed);   // [2][CW] sums, [2 CW] the ticket
  if constexpr (STATS) {
#pragma unroll
    for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
    if (tid < 2 * CW + 4) red[tid] = 0.f;   // published by the first barrier
  }
  auto flush_stats = [&](int grp) {
    if constexpr (STATS) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
#pragma unroll
        for (int m = 1; m < 16; m <<= 1) {
          s1[e] += __shfl_xor(s1[e], m);
          s2[e] += __shfl_xor(s2[e], m);
        }
      }
      if (idx == 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          atomicAdd(&red[wc * 32 + 8 * g + e], s1[e]);
          atomicAdd(&red[CW + wc * 32 + 8 * g + e], s2[e]);
        }
      }
      unsigned ticket = 0;
      if (lane == 0) ticket = atomicAdd(reinterpret_cast<unsigned*>(red + 2 * CW), 1u);
      ticket = __builtin_amdgcn_readfirstlane(ticket);
      if ((ticket & 7u) == 7u) {   // the last of the eight waves (a wave's LDS operations are served in order)
        const size_t rep = (size_t)(blockIdx.x & (p.stats_replicas - 1)) * p.stats_groups * 2 * p.Cout;
#pragma unroll
        for (int c = lane; c < CW; c += 64) {
          float* dst = p.stats + rep + (size_t)grp * 2 * p.Cout + co_base + c;
          atomicAdd(dst, red[c]);
          red[c] = 0.f;
          atomicAdd(dst + p.Cout, red[CW + c]);
          red[CW + c] = 0.f;
        }
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
    }
  };

  auto run = [&](auto KGT) {
    constexpr int KG = decltype(KGT)::value;
    constexpr int S0 = GG::template first<NKG>(KG), CNT = GG::template first<NKG>(KG + 1) - S0;
    constexpr int kDepth = (NPH * CNT >= 16) ? 2 : 3;         // fragment sets in flight (the weights take 8 VGPRs per k-step)
    // A-fragments of packed rows 32 wc + 16 a + idx for this wave's k-steps of every phase.  Packed image [slot][chunk][Cout rows][64 B]
    bf16x8 wfr[NPH][CNT][2];
    const char* const wl = p.w + ((size_t)co_base + wc * 32 + idx) * 64 + g * 16;
#pragma unroll
    for (int ph = 0; ph < NPH; ++ph)
#pragma unroll
      for (int s_ = 0; s_ < CNT; ++s_)
#pragma unroll
        for (int a = 0; a < 2; ++a)
          wfr[ph][s_][a] = *reinterpret_cast<const bf16x8*>(wl + ((size_t)(((S0 + s_) >> 1) * NCH + 2 * ph + ((S0 + s_) & 1)) * p.Cout + a * 16) * 64);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the first phase's patch blocks and the weights
    int buf = 0;
    for (int i = 0; i < ntl; ++i) {
      const Tile mine = cur;
      f32x4 acc[2][4];
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ph = 0; ph < NPH; ++ph) {
        // this phase's patch is in LDS (every wave waited for its own blocks), the other buffer and the exchange image are nobody's any more
        lds_barrier();
        if (ph + 1 < NPH) {
          dma_phase(mine, ph + 1, buf ^ 1);
        } else if (i + 1 < ntl) {
          tile += (int)gridDim.x;
          cur = tile_of(tile);
          dma_phase(cur, 0, buf ^ 1);
        }
        const char* img = smem + buf * kPhaseBytes;
        bf16x8 xf[kDepth][4];
        auto frags = [&](int s_, int fb) {   // compile-time arguments after unrolling
          const int st = S0 + s_, tap = st >> 1, cc = st & 1, ky = tap / KS, kx = tap % KS;
#pragma unroll
          for (int b = 0; b < 4; ++b)
            xf[fb][b] = *reinterpret_cast<const bf16x8*>(img + cc * kChunkBytes + xa[kx] + (2 * b + ky) * kPitch * kRow);
        };
#pragma unroll
        for (int s_ = 0; s_ < kDepth - 1; ++s_) frags(s_, s_);
#pragma unroll
        for (int s_ = 0; s_ < CNT; ++s_) {
          if (s_ + kDepth - 1 < CNT) frags(s_ + kDepth - 1, (s_ + kDepth - 1) % kDepth);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int a = 0; a < 2; ++a) acc[a][b] = mma<T>(wfr[ph][s_][a], xf[s_ % kDepth][b], acc[a][b]);
          __builtin_amdgcn_sched_barrier(0);
        }
        // the next phase's blocks (requested a k-loop ago) and the previous tile's store have landed: nothing is waited for long here,
        // and the barrier that follows never waits for memory
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        buf ^= 1;
      }
      // ---- exchange: the tile rows this wave does not finish (it finishes rows KG, KG + NKG, ...), slot = the source group's rank among
      //      the other groups.  Image: [channel group][tile row][source slot][2 x 1 KiB]
      {
        char* const xw = smem + kX + lane * 16;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          if (b % NKG == KG) continue;
          const int owner = b % NKG, slot = KG < owner ? KG : KG - 1;
#pragma unroll
          for (int a = 0; a < 2; ++a)
            *reinterpret_cast<f32x4*>(xw + (((wc * 4 + b) * (NKG - 1) + slot) * 2 + a) * 1024) = acc[a][b];
        }
      }
      lds_barrier();
      if constexpr (STATS) {
        const int grp = mine.n / (p.N / p.stats_groups);
        if (grp != cur_grp) {
          if (cur_grp >= 0) flush_stats(cur_grp);
          cur_grp = grp;
        }
      }
#pragma unroll
      for (int j = 0; j < RPW; ++j) {
        const int row = KG + NKG * j;
        f32x4 tot[2];
        {
          const char* const xr = smem + kX + lane * 16;
          f32x4 part[NKG][2];
#pragma unroll
          for (int k = 0; k < NKG; ++k) {
            if (k == KG) continue;
            const int slot = k < KG ? k : k - 1;
#pragma unroll
            for (int a = 0; a < 2; ++a) part[k][a] = *reinterpret_cast<const f32x4*>(xr + (((wc * 4 + row) * (NKG - 1) + slot) * 2 + a) * 1024);
          }
#pragma unroll
          for (int a = 0; a < 2; ++a) {
            part[KG][a] = acc[a][row];
            tot[a] = part[0][a];   // group order: the same sum whichever wave finishes the row
#pragma unroll
            for (int k = 1; k < NKG; ++k) tot[a] += part[k][a];
          }
        }
        // ---- epilogue: pixel (ty0 + row, tx0 + idx), channels ch0 .. + 7
        const bool ok = (mine.tx0 + idx < p.OW) & (mine.ty0 + row < p.OH);
        float v[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = tot[0][e] + bias_r[e];
          v[4 + e] = tot[1][e] + bias_r[4 + e];
        }
        if constexpr (STATS) {
          if (ok) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              s1[e] += v[e];
              s2[e] += v[e] * v[e];
            }
          }
        }
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = pack2<T>(v[2 * e], v[2 * e + 1]);
        char* const dst = p.out + ((((size_t)mine.n * p.OH + mine.ty0 + row) * p.OW + mine.tx0 + idx) * p.Cout + ch0) * 2;
        if (ok) tg_store16(dst, o);
      }
    }
    if constexpr (STATS) {
      lds_barrier();   // (a flush in the last tile's epilogue and this one must not overlap: conv3_cw.hip)
      if (cur_grp >= 0) flush_stats(cur_grp);
    }
  };
  if constexpr (NKG == 2) {
    if (kgr == 0) run(std::integral_constant<int, 0>{});
    else run(std::integral_constant<int, 1>{});
  } else {
    switch (kgr) {
      case 0: run(std::integral_constant<int, 0>{}); break;
      case 1: run(std::integral_constant<int, 1>{}); break;
      case 2: run(std::integral_constant<int, 2>{}); break;
      default: run(std::integral_constant<int, 3>{}); break;
    }
  }
}

template <int KS, int NCH, bool STATS, typename T, int NWC>
int launch_s2cw(const S2K& k, dim3 grid, hipStream_t st) {
  constexpr int lds = Geo<KS>::kLds;
  auto fn = conv_s2_cw_kernel<KS, NCH, STATS, T, NWC>;
  static std::atomic<bool> attr_done{false};  // one-time function attribute (benign race: idempotent)
  if (!attr_done) {
    TG_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    attr_done = true;
  }
  hipLaunchKernelGGL(fn, grid, dim3(512), lds, st, k.in, k.w, k.zero, k.IH, k.IW, k.OH, k.OW, k.Cout, k.tiles_x, k.tiles_y, k.ntiles, k.out,
                     k.bias, k.stats, k.N, k.stats_groups, k.stats_replicas);
  return tg_launch_status();
}

template <int KS>
int go_s2cw(int dtype, const void* in, const void* w_packed, const float* bias, void* out, float* stats, int stats_groups,
            int stats_replicas, int N, int IH, int IW, int Cin, int Cout, int max_workgroups, void* stream) {
  if (!in || !w_packed || !out || N <= 0 || IH <= 0 || IW <= 0) return TG_E_BADARG;
  if ((dtype != TG_BF16 && dtype != TG_F16) || (Cin != 64 && Cin != 128) || Cout <= 0 || Cout % 64) return TG_E_UNSUPPORTED;
  if ((IH & 1) || (IW & 1)) return TG_E_UNSUPPORTED;
  if (stats && (stats_groups <= 0 || N % stats_groups)) return TG_E_BADARG;
  if (stats && (stats_replicas < 1 || (stats_replicas & (stats_replicas - 1)))) return TG_E_BADARG;  // a power of two
  if (!tg_aligned16(in) || !tg_aligned16(w_packed) || !tg_aligned16(out) || (bias && !tg_aligned16(bias))) return TG_E_ALIGN;
  if ((long long)IH * IW * Cin * 2 >= 0x7fffffffLL) return TG_E_UNSUPPORTED;   // 32-bit offsets inside an image
  static const char* zero_page = [] {
    void* z = nullptr;
    return hipGetSymbolAddress(&z, HIP_SYMBOL(tg_s2cw_zero_page)) == hipSuccess ? (const char*)z : (const char*)nullptr;
  }();
  if (!zero_page) return TG_E_BADARG;
  S2K k;
  k.in = (const char*)in; k.w = (const char*)w_packed; k.zero = zero_page; k.out = (char*)out; k.bias = bias; k.stats = stats;
  k.N = N; k.IH = IH; k.IW = IW; k.OH = IH / 2; k.OW = IW / 2; k.Cout = Cout;
  k.stats_groups = stats ? stats_groups : 1;
  k.stats_replicas = stats && stats_replicas > 1 ? stats_replicas : 1;
  k.tiles_x = (k.OW + 15) / 16; k.tiles_y = (k.OH + kTH - 1) / kTH;
  const long long nt = (long long)k.tiles_x * k.tiles_y * N;
  if (nt > 0x3fffffffLL) return TG_E_UNSUPPORTED;
  k.ntiles = (int)nt;
  // 128 output channels per workgroup (the patch taken in once for all of them) where the weights of a reduction HALF fit a wave's
  // registers: KS^2 * Cin / 32 k-steps / 2 * 8 VGPRs <= 144; A/B hook TECOGAN_S2_NWC=2 forces the 64-channel form
  static const int env_nwc = [] { const char* e = getenv("TECOGAN_S2_NWC"); return e ? atoi(e) : 0; }();
  const bool wide = Cout % 128 == 0 && KS * KS * (Cin / 32) <= 36 && env_nwc != 2;
  // persistent grid: the cap's workgroups shared by the channel tiles, pixel tiles dealt evenly
  const int co_tiles = Cout / (wide ? 128 : 64);
  const int cap = max_workgroups > 0 ? max_workgroups : 256;
  const int per = cap / co_tiles > 0 ? cap / co_tiles : 1;
  const int rounds = (k.ntiles + per - 1) / per;
  const int gx = (k.ntiles + rounds - 1) / rounds;
  dim3 grid((unsigned)gx, (unsigned)co_tiles);
  hipStream_t st = (hipStream_t)stream;
#define S2_GO(NCH, TAG, NWC) (stats ? launch_s2cw<KS, NCH, true, TAG, NWC>(k, grid, st) : launch_s2cw<KS, NCH, false, TAG, NWC>(k, grid, st))
  if (wide) {
    if constexpr (KS == 3) {
      if (dtype == TG_F16) return Cin == 64 ? S2_GO(2, F16, 4) : S2_GO(4, F16, 4);
      return Cin == 64 ? S2_GO(2, BF16, 4) : S2_GO(4, BF16, 4);
    } else {   // (KS = 4: 64 reduction channels only - `wide` excludes 128)
      return dtype == TG_F16 ? S2_GO(2, F16, 4) : S2_GO(2, BF16, 4);
    }
  }
  if (dtype == TG_F16) return Cin == 64 ? S2_GO(2, F16, 2) : S2_GO(4, F16, 2);
  return Cin == 64 ? S2_GO(2, BF16, 2) : S2_GO(4, BF16, 2);
#undef S2_GO
}

}  // namespace

extern "C" int tg_conv4s2_fwd_cw(int dtype, const void* in, const void* w_packed, const float* bias, void* out, float* stats,
                                 int stats_groups, int stats_replicas, int N, int IH, int IW, int Cin, int Cout, int max_workgroups,
                                 void* stream) {
  return go_s2cw<4>(dtype, in, w_packed, bias, out, stats, stats_groups, stats_replicas, N, IH, IW, Cin, Cout, max_workgroups, stream);
}

extern "C" int tg_convt_dgrad_cw(int dtype, const void* dout, const void* w_dgrad_packed, void* din, int N, int OH, int OW, int Cout,
                                 int Cin, int max_workgroups, void* stream) {
  // din[y][x][ci] = sum_{dy,dx in -1..1} dout[2y+dy][2x+dx][co] * W[slot][ci][co]: the kernel's reduction channels are Cout, its rows Cin
  return go_s2cw<3>(dtype, dout, w_dgrad_packed, nullptr, din, nullptr, 1, 1, N, OH, OW, Cout, Cin, max_workgroups, stream);
}
