// TWO consecutive residual blocks of the generator trunk in ONE launch, stream-first (round 5; forward, 16-bit, 64 channels, 8 x 4 tiles):
//
//     h1 = relu(conv3x3(a0, W1a) + b1a)    a1 = a0 + conv3x3(h1, W2a)          code/ops.py:45-54, code/models.py:54-58,66-69,80-81
//     h2 = relu(conv3x3(a1, W1b) + b1b)    a2 = a1 + conv3x3(h2, W2b)
//
// resblock_ws.hip (one block per launch) is bound by what a workgroup has to take in - 156 KB through its CU at ~30 B/clk - plus the
// launch boundary (~2.1 us) and the start-up of its stream (~1 us until the first bytes have landed): 5.05 us per block.  Here a
// workgroup computes both blocks of its 8 x 4 output tile with the halo RECOMPUTED (h1 on 14 x 10, a1 on 12 x 8, h2 on 10 x 6 pixels
// from a 16 x 12 patch), so that its stream - patch, W1a, W2a, W1b, W2b - runs without a gap across the two blocks and one boundary
// and one start-up disappear per pair.  (Round 3's resblock2.hip did the same with the unified-wave structure of resblock.hip and
// lost: 21 instead of 12 pixel tiles on a workgroup's serial path.  Here the matrix work - 6900 instead of 3450 cycles per SIMD -
// runs beside a stream of 10 000 ticks.)
//
// Roles as in resblock_ws.hip: waves 0-3 run the first convolution of each block (32x32x16 tiles, whole K in one accumulator chain, A
// from the LDS image of W1, B from the LDS image of the block's input), waves 4-7 hold the second convolution's weights in registers
// and run it (K halves per wave, meeting through LDS).  The W1 image is ONE 72-KB region: W1b is DMA'd into it by the conv1 waves as
// soon as they are through with W1a (they wait for a1 anyway); W2b goes into the conv2 waves' registers once W2a has been used.
// Results are bit-identical to two tg_resblock_fwd_ws launches: every output value is the same sum in the same order (gated in
// tests/test_kernels_gpu.py).
//
// LDS (147 KB): W1 72 KB | patch 28 KB | h1 35 KB (then the K-half exchange of block A, then h2) | a1 12 KB.
// Images (conflict-free for the 32x32x16 reads, see resblock_ws.hip): a reader whose lanes are consecutive pixels n of a region WR
// wide, under a tap, wants the source's row index = n + const (mod 16):
//   * patch (16 wide, read by the 14-wide h1 region) and a1 (12 wide, read by the 10-wide h2 region): columns 0 .. WR - 1 at row
//     WR py + px, the two extra columns in 16-row blocks at the row whose low bits are (WR py + px) & 15 (WR / 2 odd: eight source rows
//     give sixteen residues); 8 pixels x (chunk 0 | chunk 1) per KiB, so that a DMA instruction reads whole pixels;
//   * h1 (14 wide, read by the 12-wide a1 region): pitch 28 = 12 (mod 16); h2 (10 wide, read by the 8-wide output tile): pitch 24.
// tests/test_resblock_ws_maps_cpu.py restates these maps on the CPU.
#include "common.h"
#include "rbw_common.h"
#include <type_traits>

#ifdef TG_STAMP
// Diagnostic build only: every wave of workgroup 0 records s_memtime at up to 12 points (tools/stamp_resblock.py, RB_PAIR=1).  The
// stamps stay in registers until the wave's end: a store per stamp would sit in the vector-memory queue behind the weight stream and
// every later vmcnt wait of the wave would wait for it.
__device__ long long tg_rbw2_stamps[8 * 12];
#define RB2_STAMP_DECL long long rb2_st[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define RB2_STAMP(i) rb2_st[i] = (long long)__builtin_amdgcn_s_memtime()
#define RB2_STAMP_FLUSH                                                                                     \
  do {                                                                                                      \
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) {                                                       \
      _Pragma("unroll") for (int i_ = 0; i_ < 12; ++i_) tg_rbw2_stamps[(threadIdx.x >> 6) * 12 + i_] = rb2_st[i_]; \
    }                                                                                                       \
  } while (0)
extern "C" int tg_debug_read_rbw2_stamps(long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(tg_rbw2_stamps), sizeof(long long) * n);
}
#else
#define RB2_STAMP_DECL do {} while (0)
#define RB2_STAMP(i) do {} while (0)
#define RB2_STAMP_FLUSH do {} while (0)
#endif

__device__ __attribute__((aligned(16))) unsigned int tg_rbw2_zero_page[4];

namespace {

constexpr int TH = 4;                      // output tile 8 x 4
// regions (width x height) and their 32-pixel tiles
constexpr int kP0W = 16, kP0H = TH + 8;    // patch of a0
constexpr int kH1W = 14, kH1H = TH + 6, kH1Pix = kH1W * kH1H, kNT1A = (kH1Pix + 31) / 32;   // 140 px, 5 tiles
constexpr int kA1W = 12, kA1H = TH + 4, kA1Pix = kA1W * kA1H, kNT2A = (kA1Pix + 31) / 32;   // 96 px, 3 tiles
constexpr int kH2W = 10, kH2H = TH + 2, kH2Pix = kH2W * kH2H, kNT1B = (kH2Pix + 31) / 32;   // 60 px, 2 tiles
constexpr int kNT2B = TH * 8 / 32;                                                            // 32 px, 1 tile
static_assert(kA1Pix % 32 == 0 && kNT2B == 1, "conv2's tiles are full");
// images
constexpr int kW1Bytes = 18 * 4096;
constexpr int kP0Main = (kH1W * kP0H + 15) / 16 * 16;          // 176
constexpr int kP0Rows = 224;                                   // 176 + 2 x 16, padded so that every wave of a role issues the same number of blocks
constexpr int kNPD = kP0Rows / 8, kNPC = (kNPD + 7) / 8, kNPH = (kNPD + 3) / 8;   // 28 DMA blocks: 4 per conv1 wave, 3 per conv2 wave
static_assert(kP0Main + 16 * ((kP0H + 7) / 8) <= kP0Rows, "patch rows");
static_assert(kNPD <= 8 * kNPC && kNPD > 8 * (kNPC - 1) + 3 && kNPD <= 8 * kNPH + 4 && kNPD > 8 * (kNPH - 1) + 7, "uniform DMA counts");
constexpr int kH1P = 28, kH1Chunk = kH1P * kH1H * 64;          // 17 920
constexpr int kA1Main = (kH2W * kA1H + 15) / 16 * 16;          // 80
constexpr int kA1Rows = kA1Main + 16 * ((kA1H + 7) / 8);       // 96
constexpr int kH2P = 24, kH2Chunk = kH2P * kH2H * 64;          // 9 216
constexpr int kP0 = kW1Bytes, kH1 = kP0 + kP0Rows * 128, kA1 = kH1 + 2 * kH1Chunk, kLds = kA1 + kA1Rows * 128;   // 150 528
constexpr int kXA = kH1;    // K-half exchange of block A: 4 waves x 3 tiles x 2 KiB (h1 is dead by then)
constexpr int kH2 = kH1;    // h2 (the exchange is dead by then)
constexpr int kXB = kP0;    // K-half exchange of block B (the patch is dead by then)
static_assert(4 * kNT2A * 2048 <= 2 * kH1Chunk && 2 * kH2Chunk <= 2 * kH1Chunk && 4 * kNT2B * 2048 <= kP0Rows * 128, "aliases fit");
// From [H1] on the two wave groups synchronise through LDS counters, not s_barrier: a wave that is stuck in the vector-memory issue
// queue behind 72 KB of W1b (both groups are, in turn) cannot come to a barrier, and everybody else would wait for it there
// (first measurement of this file: 11.6 instead of 7 us per pair).  One counter per event, used once per launch, zeroed by wave 0
// before the first s_barrier: [0] conv2 waves through with h1 | [1 + rt] exchange A written | [3] a1 parts written | [4 + j] W1b
// instalment j landed (conv1 waves) | [7] h2 parts written | [8 + rt] exchange B written
constexpr int kSync = kLds, kLdsAll = kLds + 64;
static_assert(kLdsAll <= 160 * 1024, "LDS");

template <int WR> __device__ __forceinline__ int div_wr(int n);   // n / WR for the pixel counts above
template <> __device__ __forceinline__ int div_wr<8>(int n) { return n >> 3; }
template <> __device__ __forceinline__ int div_wr<10>(int n) { return (n * 205) >> 11; }
template <> __device__ __forceinline__ int div_wr<12>(int n) { return (n * 171) >> 11; }
template <> __device__ __forceinline__ int div_wr<14>(int n) { return (n * 4682) >> 16; }
// row of source pixel (py, px) for a reader region WR wide (the source is WR + 2 wide): = WR py + px (mod 16)
template <int WR> __device__ __forceinline__ int src_row(int py, int px, int kMain) {
  const int v = WR * py + px;
  return px < WR ? v : kMain + 16 * (py >> 3) + (v & 15);
}

struct Rb2K {
  const char* in;
  const char* w1a; const float* b1a; const char* w2a;
  const char* w1b; const float* b1b; const char* w2b;
  char* out_h1; char* out_a1; char* out_h2; char* out_a2;
  const char* zero;
  int N, H, W, tiles_x, tiles_y;
};

template <typename T>
__global__ __launch_bounds__(512) void resblock2_ws_kernel(const Rb2K p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bx = blockIdx.x;
  const int txb = bx % p.tiles_x;
  bx /= p.tiles_x;
  const int tyb = bx % p.tiles_y;
  const int n = bx / p.tiles_y;
  const int y0 = tyb * TH, x0 = txb * 8;
  const size_t img = (size_t)n * p.H * p.W * 128;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  const int l32 = lane & 31, hi = lane >> 5;
  RB2_STAMP_DECL;
  RB2_STAMP(0);
  if (wid == 0 && lane < 16) reinterpret_cast<unsigned*>(smem + kSync)[lane] = 0u;   // (visible to all behind the first s_barrier)
  // arrive at / wait for counter c reaching `need` (LDS operations of a wave execute in order: what it wrote before `arrive` is there
  // when the counter says so; the spin is bounded - every arrival below is unconditional - so that a bug could not hang the GPU)
  auto arrive = [&](int c) {
    if (lane == 0) atomicAdd(reinterpret_cast<unsigned*>(smem + kSync) + c, 1u);
  };
  auto await = [&](int c, unsigned need) {
    // (an asm read: a volatile C++ load makes hipcc wait for vmcnt(0) - the whole weight stream - in front of every poll)
    const unsigned a = lds0 + kSync + 4 * c;
    for (int spin = 0; spin < (1 << 20); ++spin) {
      unsigned v;
      asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
      if ((unsigned)__builtin_amdgcn_readfirstlane((int)v) >= need) break;
      __builtin_amdgcn_s_sleep(1);
    }
  };

  // ---- DMA of the patch (a0 on (y0 - 4 .. , x0 - 4 ..), 16 x 12): instruction j = wid + 8 k = image rows 16 j .. = pixels 8 j .. 8 j + 7,
  // both chunks; the lane's 16 bytes: physical piece lane % 4 of row 16 j + lane / 4
  auto issue_patch = [&](auto NP) {
#pragma unroll
    for (int k = 0; k < decltype(NP)::value; ++k) {
      const int j = wid + 8 * k;
      const int lrow = lane >> 2, cc = lrow >> 3;
      const int row = 8 * j + (lrow & 7);
      int py, px;
      bool valid;
      if (row < kP0Main) {
        py = div_wr<14>(row);
        px = row - 14 * py;
        valid = row < kH1W * kP0H;
      } else {
        const int e = row - kP0Main, e4 = e & 15;
        py = 8 * (e >> 4) + ((7 * (e4 >> 1) + 7) & 7);   // the patch row whose (14 py + 14) & 15 is e4 & 14
        px = 14 + (e4 & 1);
        valid = py < kP0H;
      }
      const int piece = (lane & 3) ^ ((row >> 2) & 3);
      const int iy = y0 - 4 + py, ix = x0 - 4 + px;
      const bool ok = valid & ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
      const char* src = p.in + img + (unsigned)((iy * p.W + ix) * 128 + cc * 64 + piece * 16);
      glds16(ok ? src : p.zero, lds0 + kP0 + j * 1024);
    }
  };
  // W1 image: [tap][chunk][64 rows in matrix order][64 B] (resblock_ws.hip): the lane's source offset inside a (tap, chunk) block for
  // quarter u (16 matrix rows) of it
  auto w1_lane_off = [&](int u) {
    const int mp = 16 * u + (lane >> 2);
    const int rt_ = mp >> 5, m = mp & 31, j = m >> 3, hm = (m >> 2) & 1, e = m & 3;
    const int R = 16 * (2 * rt_ + (j & 1)) + 4 * (2 * hm + (j >> 1)) + e;
    return R * 64 + (((lane & 3) ^ ((mp >> 2) & 3)) << 4);
  };

  // ================================================================================================ conv1 phase (waves 0-3)
  // WR x RH region with origin (ry0, rx0) in the image, read from the source image at `src` (rows by src_row<WR>); the wave's NTW tiles
  // start at tile t0; h -> LDS image at `hb` (pitch HP); returns the packed values and the lane's pixels for the global store.
  // W0 / W1_ / W2_: vmcnt values under which the three instalments of the W1 image have landed.
  // (a macro-like generic lambda: everything compile-time after inlining)
#define RB2_CONV1(WR, NPIX, HP, HCHUNK, NTW, T0, SRC, SRCMAIN, HB, RY0, RX0, BIAS, W0, W1_, W2_, SYNC, PK, HYV, HXV, INS)                      \
  {                                                                                                                                    \
    const int a0_ = img_off(32 * rt + l32, hi);                                                                                        \
    int xa_[NTW][9];                                                                                                                   \
    _Pragma("unroll") for (int tw = 0; tw < NTW; ++tw) {                                                                               \
      const int n0 = 32 * ((T0) + tw) + l32;                                                                                           \
      const int nn = n0 < (NPIX) ? n0 : n0 - 32;                                                                                       \
      const int hy = div_wr<WR>(nn), hx = nn - (WR) * hy;                                                                              \
      HYV[tw] = hy;                                                                                                                    \
      HXV[tw] = hx;                                                                                                                    \
      _Pragma("unroll") for (int t = 0; t < 9; ++t) xa_[tw][t] = (SRC) + patch_off(src_row<WR>(hy + t / 3, hx + t % 3, SRCMAIN), 0, hi); \
    }                                                                                                                                  \
    f32x16 acc_[NTW];                                                                                                                  \
    _Pragma("unroll") for (int tw = 0; tw < NTW; ++tw) _Pragma("unroll") for (int i = 0; i < 16; ++i) acc_[tw][i] = 0.f;               \
    bf16x8 af_[3], bf_[3][NTW];                                                                                                        \
    auto frags_ = [&](int s, int buf) {                                                                                                \
      const int t = s >> 2, kc_ = (s >> 1) & 1, x32 = (s & 1) * 32;                                                                    \
      af_[buf] = *reinterpret_cast<const bf16x8*>(smem + (2 * t + kc_) * 4096 + (a0_ ^ x32));                                          \
      _Pragma("unroll") for (int tw = 0; tw < NTW; ++tw)                                                                               \
        bf_[buf][tw] = *reinterpret_cast<const bf16x8*>(smem + kc_ * 512 + (xa_[tw][t] ^ x32));                                        \
    };                                                                                                                                 \
    auto run_ = [&](auto S0, auto S1) {                                                                                                \
      constexpr int s0 = decltype(S0)::value, s1 = decltype(S1)::value;                                                                \
      frags_(s0, s0 % 3);                                                                                                              \
      frags_(s0 + 1, (s0 + 1) % 3);                                                                                                    \
      _Pragma("unroll") for (int s = s0; s < s1; ++s) {                                                                                \
        if (s + 2 < s1) frags_(s + 2, (s + 2) % 3);                                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                                             \
        _Pragma("unroll") for (int tw = 0; tw < NTW; ++tw) acc_[tw] = Mma32<T>::run(af_[s % 3], bf_[s % 3][tw], acc_[tw]);             \
        __builtin_amdgcn_sched_barrier(0);                                                                                             \
      }                                                                                                                                \
    };                                                                                                                                 \
    wait_vm<W0>();                                                                                                                     \
    SYNC(0);                                                                                                                           \
    run_(std::integral_constant<int, 0>{}, std::integral_constant<int, 12>{});                                                         \
    wait_vm<W1_>();                                                                                                                    \
    SYNC(1);                                                                                                                           \
    run_(std::integral_constant<int, 12>{}, std::integral_constant<int, 24>{});                                                        \
    wait_vm<W2_>();                                                                                                                    \
    SYNC(2);                                                                                                                           \
    run_(std::integral_constant<int, 24>{}, std::integral_constant<int, 36>{});                                                        \
    _Pragma("unroll") for (int tw = 0; tw < NTW; ++tw) {                                                                               \
      const int n0 = 32 * ((T0) + tw) + l32;                                                                                           \
      const int hy = HYV[tw], hx = HXV[tw];                                                                                            \
      const int y = (RY0) + hy, x = (RX0) + hx;                                                                                        \
      const bool inside = ((unsigned)y < (unsigned)p.H) & ((unsigned)x < (unsigned)p.W);                                               \
      _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                                                  \
        float v[4];                                                                                                                    \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                                                \
          v[e] = fmaxf(acc_[tw][4 * q + e] + BIAS[q][e], 0.f);                                                                         \
          v[e] = inside ? v[e] : 0.f;                                                                                                  \
        }                                                                                                                              \
        PK[tw][q >> 1][2 * (q & 1)] = pack2<T>(v[0], v[1]);                                                                            \
        PK[tw][q >> 1][2 * (q & 1) + 1] = pack2<T>(v[2], v[3]);                                                                        \
      }                                                                                                                                \
      INS[tw] = n0 < (NPIX) && inside;                                                                                                 \
      if (n0 < (NPIX)) {                                                                                                               \
        char* hrow = smem + (HB) + rt * (HCHUNK);                                                                                      \
        *reinterpret_cast<u32x4*>(hrow + img_off(hy * (HP) + hx, 2 * hi)) = PK[tw][0];                                                 \
        *reinterpret_cast<u32x4*>(hrow + img_off(hy * (HP) + hx, 2 * hi + 1)) = PK[tw][1];                                             \
      }                                                                                                                                \
    }                                                                                                                                  \
  }

  if (wid < 4) {
    // =============================================================================================== CONV1 WAVES
    const int rt = wid & 1, pg = wid >> 1;
    issue_patch(std::integral_constant<int, kNPC>{});
    {
      const int u = wid & 3, cw = wid >> 2;   // (all eight waves share W1a: quarter u of chunk cw per tap)
      const char* src = p.w1a + cw * 4096 + w1_lane_off(u);
      const unsigned dst = lds0 + cw * 4096 + u * 1024;
#pragma unroll
      for (int t = 0; t < 9; ++t) glds16(src + t * 8192, dst + t * 8192);
    }
    f32x4 biasA[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) biasA[q] = *reinterpret_cast<const f32x4*>(p.b1a + 32 * rt + 16 * hi + 4 * q);
    RB2_STAMP(1);
    // ---- block A, conv1: h1 on the 14 x 10 region at (y0 - 3, x0 - 3); tiles 0-2 (pg 0) / 3-4 (pg 1)
    // vmcnt (oldest first): kNPC patch blocks, 9 taps of W1a, 4 bias loads
    u32x4 pk1[3][2];
    int hy1[3], hx1[3];
    bool in1[3];
#define RB2_SYNC_A(j) lds_barrier()
    if (pg == 0) {
      RB2_CONV1(14, kH1Pix, kH1P, kH1Chunk, 3, 0, kP0, kP0Main, kH1, y0 - 3, x0 - 3, biasA, 4 + 6, 4 + 3, 4, RB2_SYNC_A, pk1, hy1, hx1, in1)
    } else {
      RB2_CONV1(14, kH1Pix, kH1P, kH1Chunk, 2, 3, kP0, kP0Main, kH1, y0 - 3, x0 - 3, biasA, 4 + 6, 4 + 3, 4, RB2_SYNC_A, pk1, hy1, hx1, in1)
      in1[2] = false;
    }
    RB2_STAMP(2);
    lds_barrier();   // [H1] h1 complete; nobody reads W1a any more
    RB2_STAMP(3);
    // h1 -> global (interior of the region: the workgroup's own 8 x 4 pixels), then W1b into the W1 image: this wave brings quarter
    // wid of BOTH chunks of every tap; then the second block's bias
    if (p.out_h1) {
#pragma unroll
      for (int tw = 0; tw < 3; ++tw) {
        if (in1[tw] && hy1[tw] >= 3 && hy1[tw] < 3 + TH && hx1[tw] >= 3 && hx1[tw] < 11) {
          char* dst = p.out_h1 + img + ((size_t)(y0 - 3 + hy1[tw]) * p.W + (x0 - 3 + hx1[tw])) * 128 + (32 * rt + 16 * hi) * 2;
          *reinterpret_cast<u32x4*>(dst) = pk1[tw][0];
          *reinterpret_cast<u32x4*>(dst + 16) = pk1[tw][1];
        }
      }
    }
    {
      const char* src = p.w1b + w1_lane_off(wid);
      const unsigned dst = lds0 + wid * 1024;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        glds16(src + t * 8192, dst + t * 8192);
        glds16(src + t * 8192 + 4096, dst + t * 8192 + 4096);
      }
    }
    f32x4 biasB[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) biasB[q] = *reinterpret_cast<const f32x4*>(p.b1b + 32 * rt + 16 * hi + 4 * q);
    RB2_STAMP(4);
    // ---- block B, conv1: h2 on the 10 x 6 region at (y0 - 1, x0 - 1), one tile per wave, from the a1 image, once a1 is complete.
    // Its instalments of W1b are this group's own business: each wave counts its own vmcnt - (the h1 stores, older: counted as if
    // already gone,) 18 blocks of W1b, 4 bias loads - and the four meet at a counter.
    await(3, 4);
    RB2_STAMP(8);
#define RB2_SYNC_B(j) { arrive(4 + (j)); await(4 + (j), 4); }
    u32x4 pk2[1][2];
    int hy2[1], hx2[1];
    bool in2[1];
    RB2_CONV1(10, kH2Pix, kH2P, kH2Chunk, 1, pg, kA1, kA1Main, kH2, y0 - 1, x0 - 1, biasB, 4 + 12, 4 + 6, 4, RB2_SYNC_B, pk2, hy2, hx2, in2)
    arrive(7);   // h2 part written
    RB2_STAMP(5);
    RB2_STAMP(6);
    if (p.out_h2 && in2[0] && hy2[0] >= 1 && hy2[0] <= TH && hx2[0] >= 1 && hx2[0] <= 8) {
      char* dst = p.out_h2 + img + ((size_t)(y0 - 1 + hy2[0]) * p.W + (x0 - 1 + hx2[0])) * 128 + (32 * rt + 16 * hi) * 2;
      *reinterpret_cast<u32x4*>(dst) = pk2[0][0];
      *reinterpret_cast<u32x4*>(dst + 16) = pk2[0][1];
    }
    RB2_STAMP(7);
    RB2_STAMP_FLUSH;
    return;
  }

  // ================================================================================================= CONV2 WAVES: (rt, kc)
  const int rt = wid & 1, kc = (wid >> 1) & 1;
  issue_patch(std::integral_constant<int, kNPH>{});
  {
    const int u = wid & 3, cw = wid >> 2;
    const char* src = p.w1a + cw * 4096 + w1_lane_off(u);
    const unsigned dst = lds0 + cw * 4096 + u * 1024;
#pragma unroll
    for (int t = 0; t < 9; ++t) glds16(src + t * 8192, dst + t * 8192);
  }
  RB2_STAMP(1);
  // A-fragments of the second convolution: per tap the chunk's two halves; lane (l32, hi): matrix row l32 of row tile rt, bytes 16 (2 half + hi)
  bf16x8 wx[9], wy[9];
  int w2off;
  {
    const int j = l32 >> 3, hm = (l32 >> 2) & 1, e = l32 & 3;
    const int R = 16 * (2 * rt + (j & 1)) + 4 * (2 * hm + (j >> 1)) + e;
    w2off = kc * 4096 + R * 64 + hi * 16;
  }
  auto load_w2 = [&](const char* w2, auto T0, auto T1) {   // taps [T0, T1)
#pragma unroll
    for (int t = decltype(T0)::value; t < decltype(T1)::value; ++t) {
      wx[t] = *reinterpret_cast<const bf16x8*>(w2 + w2off + t * 8192);
      wy[t] = *reinterpret_cast<const bf16x8*>(w2 + w2off + t * 8192 + 32);
    }
  };
  using I0 = std::integral_constant<int, 0>;
  using I3 = std::integral_constant<int, 3>;
  using I6 = std::integral_constant<int, 6>;
  using I9 = std::integral_constant<int, 9>;
  // vmcnt (oldest first): kNPH patch blocks, 9 taps of W1a, then 2 loads per tap of W2a
  wait_vm<6>();
  lds_barrier();   // [S0A]
  load_w2(p.w2a, I0{}, I3{});
  wait_vm<3 + 6>();
  lds_barrier();   // [S1A]
  load_w2(p.w2a, I3{}, I6{});
  wait_vm<12>();
  lds_barrier();   // [S2A]
  load_w2(p.w2a, I6{}, I9{});
  RB2_STAMP(2);

  // the second convolution of a block on NT 32-pixel tiles of a WR-wide region: B from the h image at HB (pitch HP), K half kc
#define RB2_CONV2(WR, NT, HP, HB, ACC, WAITH)                                                                                                   \
  {                                                                                                                                    \
    int xb_[NT][9];                                                                                                                    \
    _Pragma("unroll") for (int t = 0; t < NT; ++t) {                                                                                   \
      const int nn = 32 * t + l32, oy = div_wr<WR>(nn), ox = nn - (WR) * oy;                                                           \
      _Pragma("unroll") for (int tt = 0; tt < 9; ++tt) xb_[t][tt] = (HB) + img_off((oy + tt / 3) * (HP) + ox + tt % 3, hi);            \
    }                                                                                                                                  \
    WAITH; /* the h image is complete */                                                                                               \
    _Pragma("unroll") for (int t = 0; t < NT; ++t) _Pragma("unroll") for (int i = 0; i < 16; ++i) ACC[t][i] = 0.f;                     \
    bf16x8 xf_[3][NT];                                                                                                                 \
    auto frags2_ = [&](int s2, int buf) {                                                                                              \
      _Pragma("unroll") for (int t = 0; t < NT; ++t)                                                                                   \
        xf_[buf][t] = *reinterpret_cast<const bf16x8*>(smem + (xb_[t][s2 >> 1] ^ ((s2 & 1) * 32)));                                    \
    };                                                                                                                                 \
    frags2_(0, 0);                                                                                                                     \
    frags2_(1, 1);                                                                                                                     \
    _Pragma("unroll") for (int s2 = 0; s2 < 18; ++s2) {                                                                                \
      if (s2 + 2 < 18) frags2_(s2 + 2, (s2 + 2) % 3);                                                                                  \
      __builtin_amdgcn_sched_barrier(0);                                                                                               \
      _Pragma("unroll") for (int t = 0; t < NT; ++t)                                                                                   \
        ACC[t] = Mma32<T>::run((s2 & 1) ? wy[s2 >> 1] : wx[s2 >> 1], xf_[s2 % 3][t], ACC[t]);                                          \
      __builtin_amdgcn_sched_barrier(0);                                                                                               \
    }                                                                                                                                  \
  }
  // the K halves meet: wave kc keeps registers 8 kc .. 8 kc + 7 (channels 32 rt + 16 hi + 8 kc ..) and hands the other eight to its
  // partner: [rt][kc][tile][2][lane] 16 B at XB
#define RB2_XWRITE(NT, XB, ACC)                                                                                                         \
  {                                                                                                                                    \
    char* const xw = smem + (XB) + ((rt * 2 + kc) * (NT) * 2) * 1024 + lane * 16;                                                      \
    _Pragma("unroll") for (int t = 0; t < NT; ++t) _Pragma("unroll") for (int q = 0; q < 2; ++q) {                                     \
      f32x4 v;                                                                                                                         \
      _Pragma("unroll") for (int e = 0; e < 4; ++e) v[e] = kc ? ACC[t][4 * q + e] : ACC[t][8 + 4 * q + e];                             \
      *reinterpret_cast<f32x4*>(xw + (t * 2 + q) * 1024) = v;                                                                          \
    }                                                                                                                                  \
  }
  // v[0..7] = own half + the partner's, for tile t
#define RB2_XREAD(NT, XB, ACC, t, v)                                                                                                    \
  {                                                                                                                                    \
    const char* const xr = smem + (XB) + ((rt * 2 + (kc ^ 1)) * (NT) * 2) * 1024 + lane * 16;                                          \
    const f32x4 o0 = *reinterpret_cast<const f32x4*>(xr + ((t) * 2) * 1024);                                                           \
    const f32x4 o1 = *reinterpret_cast<const f32x4*>(xr + ((t) * 2 + 1) * 1024);                                                       \
    _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                                                    \
      v[e] = (kc ? ACC[t][8 + e] : ACC[t][e]) + o0[e];                                                                                 \
      v[4 + e] = (kc ? ACC[t][12 + e] : ACC[t][4 + e]) + o1[e];                                                                        \
    }                                                                                                                                  \
  }

  // ---- block A, conv2: a1 = a0 + conv(h1, W2a) on the 12 x 8 region at (y0 - 2, x0 - 2), three tiles
  u32x4 oa1[kNT2A];
  {
    f32x16 accA[kNT2A];
    RB2_CONV2(12, kNT2A, kH1P, kH1 + kc * kH1Chunk, accA, lds_barrier() /* [H1], the last s_barrier */)
    RB2_STAMP(3);
    arrive(0);
    await(0, 4);   // every conv2 wave is through with h1: its space takes the exchange
    RB2_STAMP(8);
    RB2_XWRITE(kNT2A, kXA, accA)
    arrive(1 + rt);
    await(1 + rt, 2);
    RB2_STAMP(9);
#pragma unroll
    for (int t = 0; t < kNT2A; ++t) {
      const int nn = 32 * t + l32, oy = div_wr<12>(nn), ox = nn - 12 * oy;
      const int y = y0 - 2 + oy, x = x0 - 2 + ox;
      const bool inside = ((unsigned)y < (unsigned)p.H) & ((unsigned)x < (unsigned)p.W);
      float v[8];
      RB2_XREAD(kNT2A, kXA, accA, t, v)
      // skip: channels 16 hi + 8 kc .. + 7 of chunk rt = piece 2 hi + kc of patch pixel (oy + 2, ox + 2)
      const u32x4 rr = *reinterpret_cast<const u32x4*>(smem + kP0 + patch_off(14 * (oy + 2) + ox + 2, rt, 2 * hi + kc));
      u32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float s0 = bits16_to_f32<T>((unsigned short)(rr[e] & 0xffffu)), s1 = bits16_to_f32<T>((unsigned short)(rr[e] >> 16));
        o[e] = inside ? pack2<T>(v[2 * e] + s0, v[2 * e + 1] + s1) : 0u;   // (a1 outside the image is the next conv's zero padding)
      }
      *reinterpret_cast<u32x4*>(smem + kA1 + patch_off(src_row<10>(oy, ox, kA1Main), rt, 2 * hi + kc)) = o;
      oa1[t] = o;
    }
    arrive(3);   // a1 part written
  }
  RB2_STAMP(4);
  // W2a is used up: the registers take W2b, behind W1b in the stream (nobody waits for these waves until h2 is there); a1 goes to
  // global memory behind them (the conv1 waves only need the LDS copy; with the stores inside the loop above these waves sat in the
  // issue queue behind W1b for ~4000 ticks before a1 was complete)
  load_w2(p.w2b, I0{}, I9{});
#pragma unroll
  for (int t = 0; t < kNT2A; ++t) {
    const int nn = 32 * t + l32, oy = div_wr<12>(nn), ox = nn - 12 * oy;
    const int y = y0 - 2 + oy, x = x0 - 2 + ox;
    if (((unsigned)y < (unsigned)p.H) & ((unsigned)x < (unsigned)p.W) && oy >= 2 && oy < 2 + TH && ox >= 2 && ox < 10)
      *reinterpret_cast<u32x4*>(p.out_a1 + img + ((size_t)y * p.W + x) * 128 + (32 * rt + 16 * hi + 8 * kc) * 2) = oa1[t];
  }
  RB2_STAMP(5);
  // ---- block B, conv2: a2 = a1 + conv(h2, W2b) on the 8 x 4 tile
  {
    f32x16 accB[kNT2B];
    RB2_CONV2(8, kNT2B, kH2P, kH2 + kc * kH2Chunk, accB, await(7, 4))
    RB2_STAMP(6);
    RB2_XWRITE(kNT2B, kXB, accB)
    arrive(8 + rt);
    await(8 + rt, 2);
    const int oy = l32 >> 3, ox = l32 & 7;
    const int y = y0 + oy, x = x0 + ox;
    float v[8];
    RB2_XREAD(kNT2B, kXB, accB, 0, v)
    const u32x4 rr = *reinterpret_cast<const u32x4*>(smem + kA1 + patch_off(10 * (oy + 2) + ox + 2, rt, 2 * hi + kc));
    if (y < p.H && x < p.W) {
      u32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float s0 = bits16_to_f32<T>((unsigned short)(rr[e] & 0xffffu)), s1 = bits16_to_f32<T>((unsigned short)(rr[e] >> 16));
        o[e] = pack2<T>(v[2 * e] + s0, v[2 * e + 1] + s1);
      }
      *reinterpret_cast<u32x4*>(p.out_a2 + img + ((size_t)y * p.W + x) * 128 + (32 * rt + 16 * hi + 8 * kc) * 2) = o;
    }
  }
  RB2_STAMP(7);
  RB2_STAMP_FLUSH;
#undef RB2_SYNC_A
#undef RB2_SYNC_B
#undef RB2_CONV1
#undef RB2_CONV2
#undef RB2_XWRITE
#undef RB2_XREAD
}

template <typename T>
int launch_rb2(const Rb2K& k, unsigned blocks, hipStream_t st) {
  auto fn = resblock2_ws_kernel<T>;
  static std::atomic<bool> attr_done{false};
  if (!attr_done) {
    TG_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsAll));
    attr_done = true;
  }
  hipLaunchKernelGGL(fn, dim3(blocks), dim3(512), kLdsAll, st, k);
  return tg_launch_status();
}

}  // namespace

extern "C" int tg_resblock2_fwd_ws(int dtype, const void* in, const void* w1a_packed, const float* b1a, const void* w2a_packed,
                                   const void* w1b_packed, const float* b1b, const void* w2b_packed, void* out_h1, void* out_a1,
                                   void* out_h2, void* out_a2, int N, int H, int W, int C, void* stream) {
  if (!in || !w1a_packed || !b1a || !w2a_packed || !w1b_packed || !b1b || !w2b_packed || !out_a1 || !out_a2 || N <= 0 || H <= 0 || W <= 0)
    return TG_E_BADARG;   // (out_h1 / out_h2 may be null: inference)
  if ((dtype != TG_BF16 && dtype != TG_F16) || C != 64) return TG_E_UNSUPPORTED;
  const void* ptrs[] = {in, w1a_packed, b1a, w2a_packed, w1b_packed, b1b, w2b_packed, out_h1, out_a1, out_h2, out_a2};
  for (const void* q : ptrs)
    if (q && !tg_aligned16(q)) return TG_E_ALIGN;
  if ((long long)H * W * 128 > 0x7fffffffLL) return TG_E_UNSUPPORTED;
  static const char* zero_page = [] {
    void* z = nullptr;
    return hipGetSymbolAddress(&z, HIP_SYMBOL(tg_rbw2_zero_page)) == hipSuccess ? (const char*)z : (const char*)nullptr;
  }();
  if (!zero_page) return TG_E_BADARG;
  Rb2K k;
  k.in = (const char*)in;
  k.w1a = (const char*)w1a_packed; k.b1a = b1a; k.w2a = (const char*)w2a_packed;
  k.w1b = (const char*)w1b_packed; k.b1b = b1b; k.w2b = (const char*)w2b_packed;
  k.out_h1 = (char*)out_h1; k.out_a1 = (char*)out_a1; k.out_h2 = (char*)out_h2; k.out_a2 = (char*)out_a2; k.zero = zero_page;
  k.N = N; k.H = H; k.W = W;
  k.tiles_x = (W + 7) / 8;
  k.tiles_y = (H + TH - 1) / TH;
  const long long blocks = (long long)k.tiles_x * k.tiles_y * N;
  if (blocks > 0x7fffffffLL) return TG_E_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  return dtype == TG_F16 ? launch_rb2<F16>(k, (unsigned)blocks, st) : launch_rb2<BF16>(k, (unsigned)blocks, st);
}
