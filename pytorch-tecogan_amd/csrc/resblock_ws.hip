// One residual block of the generator trunk in ONE launch, wave-specialised and stream-first (round 5; forward, 16-bit, 64 channels):
//
//     h   = relu(conv3x3(a, W1) + b1)          code/ops.py:45-54 (residual_block), code/models.py:54-58,66-69,80-81
//     out = a + conv3x3(h, W2)
//
// resblock.hip (rounds 1-4) ran all eight waves through the same phases with a split-K exchange between them; its stamps
// (profiles/r04_z_stamp_resblock.log) showed a workgroup of 9900 ticks for 1728 cycles of MFMA: 2300 ticks until the patch is in
// LDS, conv1 2400 / 3950 ticks (bound by 288 KB of LDS fragment reads - the four row-tile waves of a K half read the SAME
// fragments), exchange + finalise 1300, and a weight stream (147 KB through the CU's vector-memory path at ~33 B/clk ~ 4500
// ticks) that only moves while a compute wave is free to issue it.  This kernel is built the other way round: what bounds a
// workgroup is the STREAM (166 KB at ~33 B/clk = 5000 ticks), so the stream starts at tick 0, runs in the order its bytes are
// needed - patch, W1 tap by tap, W2 tap by tap - and never waits for a wave that computes:
//   * all eight waves first issue the whole input of conv1 by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write
//     pass): the 12 x (TH + 4) pixel patch (out-of-image positions read a zero page) and the 72 KB of W1, tap by tap, in the
//     row order the 32x32 matrix tiles want (the DMA's per-lane source address does the permutation);
//   * waves 0-3 (one per SIMD) run conv1: wave (rt, pg) = 32 output channels x the pg-th half of the region's 32-pixel tiles,
//     v_mfma_f32_32x32x16, the WHOLE K in one accumulator chain - no split-K, no exchange; per matrix instruction one A and one
//     B fragment from LDS (8x8 tiles: two B per A), both images conflict-free (below), software-pipelined two steps ahead.
//     conv1 starts when the patch and the first three taps of W1 have landed and meets the later taps at two more barriers
//     (every DMA issuer counts its own vmcnt); then bias + relu, h -> LDS, and - after the barrier, off the critical path - h ->
//     global for the backward pass;
//   * waves 4-7 (their SIMD partners) own conv2.  Its weights never touch LDS: wave (rt, kc) loads the 18 A-fragments of its 32
//     rows and its half of K straight into registers - six behind each of conv1's three barriers, so that they queue BEHIND the
//     W1 taps still on their way (first measurement of this file: with W2 issued up front by the conv1 waves the two streams
//     interleaved and conv1 could not start before tick 5100).  conv2 = 32x32x16 tiles over the output tile's pixels, B from
//     the LDS copy of h (one read per matrix instruction), the two K halves meet through 8 registers each in LDS (the dead W1
//     image), + skip from the LDS patch, 16-byte stores.
// LDS: W1 72 KB + patch 12 / 20 KB + h 18 / 30 KB (8x4 / 8x8 tiles).
//
// LDS images.  ds_read_b128 serves a wave in four fixed 16-lane groups ({0-3,12-15,20-27}, ...; MI355X_MICROARCH.md, LDS); a
// group takes one LDS cycle when its lanes hit 16 distinct 16-byte slots mod 256 B.  In a 32x32x16 operand read the 32 lanes of
// a half-wave hold 32 different rows (pixels / channels) and the SAME 16-byte piece, so a group is conflict-free iff its 16
// rows are distinct mod 16 once the piece index is XOR-ed with bits 2-3 of the row: slot = 4 row + (piece ^ (row >> 2 & 3)).
//   * W1: rows = output channels in matrix-row order, consecutive: trivially distinct;
//   * patch: region pixel n = 10 hy + hx under tap (dy, dx) reads patch pixel (hy + dy, hx + dx); the patch is stored so that
//     its row index is congruent to 10 py + px mod 16: columns 0-9 at row 10 py + px, columns 10-11 in extra 16-row blocks at
//     the row whose low four bits are (10 py + px) & 15 (eight patch rows give sixteen different residues: no row is wasted
//     for 8x4 tiles).  Then row = n + const (mod 16) for every tap, and any 16 lanes with distinct n mod 16 are conflict-free;
//   * h: conv2's lane l reads output pixel (l / 8, l % 8) of its 32-pixel tile under a tap; with a row pitch of 24 (= 8 mod 16)
//     the row is 8 (l / 8) + l % 8 + const = l + const (mod 16): conflict-free the same way.
// tests/test_resblock_ws_maps_cpu.py restates these maps on the CPU and checks every read and every lane group.
#include "common.h"
#include "rbw_common.h"
#include <cstdlib>
#include <type_traits>

#ifdef TG_STAMP
// Diagnostic build only (-DTG_STAMP, tools/stamp_resblock.py): every wave of workgroup 0 records s_memtime at
// [0 start | 1 DMA issued | 2 first stage passed | 3 conv1 done / last stage passed | 4 h barrier passed | 5 h stored / conv2 done | 6 end]
__device__ long long tg_rbw_stamps[8 * 8];
#define RBW_STAMP(i)                                                                         \
  do {                                                                                       \
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0)                                          \
      tg_rbw_stamps[(threadIdx.x >> 6) * 8 + (i)] = (long long)__builtin_amdgcn_s_memtime(); \
  } while (0)
extern "C" int tg_debug_read_rbw_stamps(long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(tg_rbw_stamps), sizeof(long long) * n);
}
#else
#define RBW_STAMP(i) do {} while (0)
#endif

// out-of-image patch positions and the unused rows of the patch image are DMA'd from here
// (every lane of an out-of-image position reads the SAME 16 bytes; a page with 16 bytes per lane measured the same: r05_b log)
__device__ __attribute__((aligned(16))) unsigned int tg_rbw_zero_page[4];

namespace {

constexpr int kW1Bytes = 18 * 4096;   // [tap][chunk][64 rows in matrix order][64 B]

template <int TH> struct GeoW {
  static constexpr int kPH = TH + 4;                          // patch rows (12 columns)
  static constexpr int kMain = (kPH * 10 + 15) / 16 * 16;     // rows of columns 0-9, rounded to a DMA block: 80 / 128
  static constexpr int kXB = (kPH + 7) / 8;                   // 16-row blocks for columns 10-11: 1 / 2
  static constexpr int kPRows = kMain + 16 * kXB;             // 96 / 160
  static constexpr int kPBytes = 2 * kPRows * 64;             // the patch image: 8 pixels x (chunk 0 rows | chunk 1 rows) per KiB
  static constexpr int kNPD = kPRows / 8;                     // 1-KiB DMA instructions (8 whole pixels each): 12 / 20
  static constexpr int kHPix = (TH + 2) * 10;                 // h region: 10 x (TH + 2)
  static constexpr int kHP = 24;                               // row pitch of the h image (= 8 mod 16, see the header)
  static constexpr int kHRows = (TH + 2) * kHP;
  static constexpr int kHChunk = kHRows * 64;
  static constexpr int kPatch0 = kW1Bytes, kH0 = kW1Bytes + kPBytes;
  static constexpr int kLds = kH0 + 2 * kHChunk;              // 104 448 / 124 928
  static constexpr int NT1 = (kHPix + 31) / 32;               // 32-pixel tiles of conv1: 2 / 4 (the last one partial)
  static constexpr int NTW = NT1 / 2;                         // ... per compute wave
  static constexpr int NT2 = TH * 8 / 32;                     // 32-pixel tiles of conv2: 1 / 2
  static constexpr int kNPC = (kNPD + 7) / 8, kNPH = (kNPD + 3) / 8;   // patch DMA instructions of a compute / helper wave
  // block j = wave + 8 k: every wave of a role issues the same number of blocks (the vmcnt arithmetic below is per role)
  static_assert(kNPD <= 8 * kNPC && kNPD > 8 * (kNPC - 1) + 3, "waves 0-3 issue kNPC patch blocks each");
  static_assert(kNPD <= 8 * kNPH + 4 && kNPD > 8 * (kNPH - 1) + 7, "waves 4-7 issue kNPH patch blocks each");
};

// row of patch pixel (py, px) inside a chunk image: congruent to 10 py + px mod 16 (see the header)
__device__ __forceinline__ int patch_row(int py, int px, int kMain) {
  const int v = 10 * py + px;
  return px < 10 ? v : kMain + 16 * (py >> 3) + (v & 15);
}
struct RbwK {
  const char* in;
  const char* w1;
  const float* b1;
  const char* w2;
  char* out_h;
  char* out_a;
  const char* zero;   // 16 bytes of zeros (tg_rbw_zero_page): a kernel argument, so that its address sits in SGPRs
  int N, H, W, tiles_x, tiles_y;
  int skip;           // 1: out_a = in + conv2(h); 0: out_a = conv2(h)
};

// conv1 meets W1 in three instalments: taps [0, 3) (with the patch), [3, 6), [6, 9)
constexpr int kStageTaps[3] = {3, 6, 9};
// (Measured and dropped: every third workgroup of an XCD starting an instalment with another tap, so that the workgroups of an XCD
// do not all ask its L2 for the same two 4-KB blocks at once - 9.1 instead of 5.0 us per launch: requests for the same lines at the
// same time are what the L2 serves best.  profiles/r05_a_resblock_ws_ab.log)

// The arguments are passed one by one, the ones a wave needs first in front: with -mllvm -amdgpu-kernarg-preload-count=16 (csrc/build.sh)
// the first 16 dwords arrive in SGPRs with the wave instead of through an s_load round trip at its first instruction - which sits in
// front of the very first DMA address of every workgroup of every launch (profiles/r05_k_kernarg_preload_ab.log).
template <typename T, int TH>
__global__ __launch_bounds__(512) void resblock_ws_kernel(const char* a_in, const char* a_w1, const char* a_zero, int a_H, int a_W,
                                                          int a_tiles_x, int a_tiles_y, const char* a_w2, const float* a_b1,
                                                          char* a_out_h, char* a_out_a, int a_N, int a_skip) {
  RbwK p;
  p.in = a_in; p.w1 = a_w1; p.zero = a_zero; p.H = a_H; p.W = a_W; p.tiles_x = a_tiles_x; p.tiles_y = a_tiles_y;
  p.w2 = a_w2; p.b1 = a_b1; p.out_h = a_out_h; p.out_a = a_out_a; p.N = a_N; p.skip = a_skip;
  using G = GeoW<TH>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bx = blockIdx.x;
  const int txb = bx % p.tiles_x;
  bx /= p.tiles_x;
  const int tyb = bx % p.tiles_y;
  const int n = bx / p.tiles_y;
  const int y0 = tyb * TH, x0 = txb * 8;
  const char* const in_n = p.in + (size_t)n * p.H * p.W * 128;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  RBW_STAMP(0);

  // ---- DMA, all eight waves.  Patch: instruction j = wid + 8 k covers image rows 16 j .. + 15 = pixels 8 j .. 8 j + 7, both chunks;
  // the lane's 16 bytes are physical piece lane % 4 of row 16 j + lane / 4.
  auto issue_patch = [&](auto NP) {
#pragma unroll
    for (int k = 0; k < decltype(NP)::value; ++k) {
#ifdef RBW_ISSUE4   // diagnostic (profiles/r05_b_resblock2_ws_ab.log): only the four conv1 waves issue the patch and W1
      const int j = wid + 4 * k;
#else
      const int j = wid + 8 * k;   // wave-uniform
#endif
      const int lrow = lane >> 2, cc = lrow >> 3;
      const int row = 8 * j + (lrow & 7);   // patch_row of the lane's pixel
      int py, px;
      bool valid;
      if (row < G::kMain) {
        py = (row * 205) >> 11;   // row / 10, exact below 1024
        px = row - 10 * py;
        valid = row < G::kPH * 10;
      } else {
        const int e = row - G::kMain, e4 = e & 15;
        py = 8 * (e >> 4) + ((5 * (e4 >> 1) + 7) & 7);   // the patch row whose (10 py + 10) & 15 is e4 & 14
        px = 10 + (e4 & 1);
        valid = py < G::kPH;
      }
      const int piece = (lane & 3) ^ ((row >> 2) & 3);
      const int iy = y0 - 2 + py, ix = x0 - 2 + px;
      const bool ok = valid & ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
      const char* src = in_n + (unsigned)((iy * p.W + ix) * 128 + cc * 64 + piece * 16);   // (an image is < 4 GB)
      glds16(ok ? src : p.zero, lds0 + G::kPatch0 + j * 1024);
    }
  };
  // W1: wave (u = wid % 4, chunk = wid / 4) brings quarter u (16 matrix rows) of block (tap, chunk) for every tap.  LDS row
  // m' = 32 rt + m holds the output channel that makes a lane's 16 accumulator registers 16 CONSECUTIVE channels:
  // D row m = 8 j + 4 hi + e (reg = 4 j + e, hi = lane / 32)  <->  channel 32 rt + 16 hi + 4 j + e, i.e. packed row
  // 16 (2 rt + j % 2) + 4 (2 hi + j / 2) + e of the [tap][chunk][64 rows][64 B] image (row_to_channel<BF16>, common.h).
  auto issue_w1 = [&]() {
    const int u = wid & 3, cw = wid >> 2;
    const int mp = 16 * u + (lane >> 2);
    const int rt = mp >> 5, m = mp & 31, j = m >> 3, hm = (m >> 2) & 1, e = m & 3;
    const int R = 16 * (2 * rt + (j & 1)) + 4 * (2 * hm + (j >> 1)) + e;
    const int piece = (lane & 3) ^ ((mp >> 2) & 3);
    const char* src = p.w1 + cw * 4096 + R * 64 + piece * 16;
    const unsigned dst = lds0 + cw * 4096 + u * 1024;
#ifdef RBW_ISSUE4
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      glds16(src - cw * 4096 + t * 8192, dst - cw * 4096 + t * 8192);
      glds16(src - cw * 4096 + t * 8192 + 4096, dst - cw * 4096 + t * 8192 + 4096);
    }
#else
#pragma unroll
    for (int t = 0; t < 9; ++t) glds16(src + t * 8192, dst + t * 8192);
#endif
  };

  if (wid >= 4) {
    // ============================================================== CONV2 WAVES: (rt, kc) = 32 output channels x K half (input chunk)
    const int rt = wid & 1, kc = (wid >> 1) & 1;
    const int l32 = lane & 31, hi = lane >> 5;
#ifndef RBW_ISSUE4
    issue_patch(std::integral_constant<int, G::kNPH>{});
    issue_w1();
    constexpr int kD0 = 6, kD1 = 3;   // W1 taps of this wave still on their way at the first / second barrier
#else
    constexpr int kD0 = 0, kD1 = 0;
#endif
    RBW_STAMP(1);
    // A-fragments of conv2: per tap the chunk's two halves; lane (l32, hi): matrix row l32 of row tile rt - the packed row that makes
    // its 16 accumulator registers 16 consecutive channels (as W1 above) - bytes 16 (2 half + hi) ..
    // (Built and measured slower, -DRBW_W2_ROWS: per tap two loads of WHOLE 64-byte rows - X: lane <- matrix row l32 % 16, piece
    // 2 (l32 / 16) + hi; Y: the same of row 16 + l32 % 16 - and v_permlane16_swap to make the two fragments of them: half the
    // 128-byte lines per instruction, +0.12 us per launch.  profiles/r05_a_resblock_ws_ab.log)
    bf16x8 wx[9], wy[9];
    const char* w2l;
#ifdef RBW_W2_ROWS
    {
      const int m = l32 & 15, j = m >> 3, hm = (m >> 2) & 1, e = m & 3;
      const int R = 16 * (2 * rt + (j & 1)) + 4 * (2 * hm + (j >> 1)) + e;   // (matrix rows m and m + 16: j and j + 2, i.e. R and R + 4)
      w2l = p.w2 + kc * 4096 + R * 64 + (2 * (l32 >> 4) + hi) * 16;
    }
    constexpr int kYOff = 4 * 64;
#else
    {
      const int j = l32 >> 3, hm = (l32 >> 2) & 1, e = l32 & 3;
      const int R = 16 * (2 * rt + (j & 1)) + 4 * (2 * hm + (j >> 1)) + e;
      w2l = p.w2 + kc * 4096 + R * 64 + hi * 16;
    }
    constexpr int kYOff = 32;
#endif
    auto load_w2 = [&](auto T0, auto T1) {   // taps [T0, T1)
#pragma unroll
      for (int t = decltype(T0)::value; t < decltype(T1)::value; ++t) {
        wx[t] = *reinterpret_cast<const bf16x8*>(w2l + t * 8192);
        wy[t] = *reinterpret_cast<const bf16x8*>(w2l + t * 8192 + kYOff);
      }
    };
    // W2 goes out in four batches around conv1's barriers, so that these waves are never stuck in an issue queue when a barrier
    // is due; vmcnt arithmetic (oldest first): kNPH patch blocks, 9 W1 taps, then 2 loads per tap of W2
#ifndef RBW_KA   // (2/5/8, 1/4/7, 3/6/9: each tap of W2 that goes out earlier costs conv1 more than it gives conv2; r05_a log)
#define RBW_KA 0
#define RBW_KB 3
#define RBW_KC 6
#endif
    constexpr int kA = RBW_KA, kB = RBW_KB, kC = RBW_KC;   // taps [0, kA) before the first barrier, [kA, kB) behind it, [kB, kC), [kC, 9)
    load_w2(std::integral_constant<int, 0>{}, std::integral_constant<int, kA>{});
    wait_vm<kD0 + 2 * kA>();    // patch + taps 0-2 of W1
    lds_barrier();
    RBW_STAMP(2);
    load_w2(std::integral_constant<int, kA>{}, std::integral_constant<int, kB>{});
    wait_vm<kD1 + 2 * kB>();   // taps 3-5
    lds_barrier();
    load_w2(std::integral_constant<int, kB>{}, std::integral_constant<int, kC>{});
    wait_vm<2 * kC>();   // taps 6-8
    lds_barrier();
    RBW_STAMP(3);
    load_w2(std::integral_constant<int, kC>{}, std::integral_constant<int, 9>{});
    // B addresses: lane = output pixel (l32 / 8, l32 % 8) of 32-pixel tile t (4 tile rows) under each tap, piece hi
    int xb[G::NT2][9];
#pragma unroll
    for (int t = 0; t < G::NT2; ++t)
#pragma unroll
      for (int tt = 0; tt < 9; ++tt)
        xb[t][tt] = G::kH0 + kc * G::kHChunk + img_off((4 * t + (l32 >> 3) + tt / 3) * G::kHP + (l32 & 7) + tt % 3, hi);
    lds_barrier();   // h complete
    RBW_STAMP(4);
    f32x16 acc2[G::NT2];
#pragma unroll
    for (int t = 0; t < G::NT2; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc2[t][i] = 0.f;
    bf16x8 xf[3][G::NT2];
    auto frags2 = [&](int s2, int buf) {
#pragma unroll
      for (int t = 0; t < G::NT2; ++t)
        xf[buf][t] = *reinterpret_cast<const bf16x8*>(smem + (xb[t][s2 >> 1] ^ ((s2 & 1) * 32)));
    };
    frags2(0, 0);
    frags2(1, 1);
#pragma unroll
    for (int tp = 0; tp < 9; ++tp) {
#ifdef RBW_W2_ROWS
      // the tap's two fragments out of its two row loads
      u32x4 fx = __builtin_bit_cast(u32x4, wx[tp]), fy = __builtin_bit_cast(u32x4, wy[tp]);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const auto r = __builtin_amdgcn_permlane16_swap(fx[e], fy[e], false, false);
        fx[e] = r[0];
        fy[e] = r[1];
      }
      const bf16x8 f0 = __builtin_bit_cast(bf16x8, fx), f1 = __builtin_bit_cast(bf16x8, fy);
#else
      const bf16x8 f0 = wx[tp], f1 = wy[tp];
#endif
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {
        const int s2 = 2 * tp + h2;
        if (s2 + 2 < 18) frags2(s2 + 2, (s2 + 2) % 3);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < G::NT2; ++t) acc2[t] = Mma32<T>::run(h2 ? f1 : f0, xf[s2 % 3][t], acc2[t]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    RBW_STAMP(5);
    // the two K halves meet: wave kc finalises registers 8 kc .. 8 kc + 7 (channels 32 rt + 16 hi + 8 kc ..) and hands the other
    // eight to its partner through the dead W1 image: [rt][kc][tile][2][lane] 16 B
    char* const xw = smem + ((rt * 2 + kc) * G::NT2 * 2) * 1024 + lane * 16;
    const char* const xr = smem + ((rt * 2 + (kc ^ 1)) * G::NT2 * 2) * 1024 + lane * 16;
#pragma unroll
    for (int t = 0; t < G::NT2; ++t)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = kc ? acc2[t][4 * q + e] : acc2[t][8 + 4 * q + e];
        *reinterpret_cast<f32x4*>(xw + (t * 2 + q) * 1024) = v;
      }
    lds_barrier();
    const float sk = p.skip ? 1.f : 0.f;
#pragma unroll
    for (int t = 0; t < G::NT2; ++t) {
      const int oy = 4 * t + (l32 >> 3), ox = l32 & 7;
      const int y = y0 + oy, x = x0 + ox;
      const f32x4 o0 = *reinterpret_cast<const f32x4*>(xr + (t * 2) * 1024);
      const f32x4 o1 = *reinterpret_cast<const f32x4*>(xr + (t * 2 + 1) * 1024);
      // skip connection: channels 16 hi + 8 kc .. + 7 of chunk rt = piece 2 hi + kc of patch pixel (oy + 2, ox + 2)
      const u32x4 rr = *reinterpret_cast<const u32x4*>(smem + G::kPatch0 + patch_off(10 * (oy + 2) + ox + 2, rt, 2 * hi + kc));
      if (y < p.H && x < p.W) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = (kc ? acc2[t][8 + e] : acc2[t][e]) + o0[e];
          v[4 + e] = (kc ? acc2[t][12 + e] : acc2[t][4 + e]) + o1[e];
        }
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float s0 = bits16_to_f32<T>((unsigned short)(rr[e] & 0xffffu)), s1 = bits16_to_f32<T>((unsigned short)(rr[e] >> 16));
          o[e] = pack2<T>(v[2 * e] + sk * s0, v[2 * e + 1] + sk * s1);
        }
        tg_store16(p.out_a + (((size_t)n * p.H + y) * p.W + x) * 128 + (32 * rt + 16 * hi + 8 * kc) * 2, o);
      }
    }
    RBW_STAMP(6);
    return;
  }

  // ================================================================ CONV1 WAVES: (rt, pg) = 32 output channels x half of the pixel tiles
  const int rt = wid & 1, pg = wid >> 1;
  const int l32 = lane & 31, hi = lane >> 5;
#ifdef RBW_ISSUE4
  issue_patch(std::integral_constant<int, G::kNPD / 4>{});
  constexpr int kPerTap = 2;
#else
  issue_patch(std::integral_constant<int, G::kNPC>{});
  constexpr int kPerTap = 1;
#endif
  issue_w1();
  f32x4 bias4[4];   // bias of the lane's 16 channels
#pragma unroll
  for (int q = 0; q < 4; ++q) bias4[q] = *reinterpret_cast<const f32x4*>(p.b1 + 32 * rt + 16 * hi + 4 * q);
  RBW_STAMP(1);

  // fragment addresses while the stream is on its way.  A: row 32 rt + l32 of a (tap, chunk) block, piece hi (+ 2 for the second
  // half of the chunk: XOR 32 on the byte offset); B: the lane's region pixel under each tap
  const int a0 = img_off(32 * rt + l32, hi);
  int xa[G::NTW][9];
  int hyv[G::NTW], hxv[G::NTW];
#pragma unroll
  for (int tw = 0; tw < G::NTW; ++tw) {
    // (spare lanes of the last tile read the pixel 32 below: same residue mod 16, so the group stays conflict-free)
    const int n0 = 32 * (pg * G::NTW + tw) + l32;
    const int nn = n0 < G::kHPix ? n0 : n0 - 32;
    const int hy = (nn * 205) >> 11, hx = nn - 10 * hy;
    hyv[tw] = hy;
    hxv[tw] = hx;
#pragma unroll
    for (int t = 0; t < 9; ++t) xa[tw][t] = G::kPatch0 + patch_off(patch_row(hy + t / 3, hx + t % 3, G::kMain), 0, hi);
  }

  // ---- conv1: 36 steps (tap, chunk, half of the chunk) of one A fragment, NTW B fragments, NTW matrix instructions
  f32x16 acc[G::NTW];
#pragma unroll
  for (int tw = 0; tw < G::NTW; ++tw)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[tw][i] = 0.f;
  bf16x8 af[3], bfr[3][G::NTW];
  auto frags = [&](int s, int buf) {   // compile-time arguments after unrolling
    const int t = s >> 2, kc = (s >> 1) & 1, x32 = (s & 1) * 32;
    af[buf] = *reinterpret_cast<const bf16x8*>(smem + (2 * t + kc) * 4096 + (a0 ^ x32));
#pragma unroll
    for (int tw = 0; tw < G::NTW; ++tw)
      bfr[buf][tw] = *reinterpret_cast<const bf16x8*>(smem + kc * 512 + (xa[tw][t] ^ x32));
  };
  auto run_steps = [&](auto S0, auto S1) {
    constexpr int s0 = decltype(S0)::value, s1 = decltype(S1)::value;
    frags(s0, s0 % 3);
    frags(s0 + 1, (s0 + 1) % 3);
#pragma unroll
    for (int s = s0; s < s1; ++s) {
      if (s + 2 < s1) frags(s + 2, (s + 2) % 3);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int tw = 0; tw < G::NTW; ++tw) acc[tw] = Mma32<T>::run(af[s % 3], bfr[s % 3][tw], acc[tw]);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // vmcnt arithmetic (oldest first): kNPC patch blocks, 9 W1 taps, 4 bias loads
  wait_vm<4 + kPerTap * (9 - kStageTaps[0])>();
  lds_barrier();
  RBW_STAMP(2);
  run_steps(std::integral_constant<int, 0>{}, std::integral_constant<int, 4 * kStageTaps[0]>{});
  wait_vm<4 + kPerTap * (9 - kStageTaps[1])>();
  lds_barrier();
  run_steps(std::integral_constant<int, 4 * kStageTaps[0]>{}, std::integral_constant<int, 4 * kStageTaps[1]>{});
  wait_vm<4 + kPerTap * (9 - kStageTaps[2])>();
  lds_barrier();
  run_steps(std::integral_constant<int, 4 * kStageTaps[1]>{}, std::integral_constant<int, 4 * kStageTaps[2]>{});
  RBW_STAMP(3);

  // ---- h = relu(conv1 + b1), zero outside the image (conv2 pads h with zeros), rounded to 16 bits exactly as the unfused path
  // would read it back: -> LDS (pieces 2 hi, 2 hi + 1 of chunk rt) now, -> global (the backward pass needs it; null in inference)
  // behind the barrier, where nobody waits for it
  u32x4 pk[G::NTW][2];
  bool st_h[G::NTW];
#pragma unroll
  for (int tw = 0; tw < G::NTW; ++tw) {
    const int nn = 32 * (pg * G::NTW + tw) + l32;
    const int hy = hyv[tw], hx = hxv[tw];
    const int y = y0 - 1 + hy, x = x0 - 1 + hx;
    const bool inside = ((unsigned)y < (unsigned)p.H) & ((unsigned)x < (unsigned)p.W);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] = fmaxf(acc[tw][4 * q + e] + bias4[q][e], 0.f);
        v[e] = inside ? v[e] : 0.f;
      }
      pk[tw][q >> 1][2 * (q & 1)] = pack2<T>(v[0], v[1]);
      pk[tw][q >> 1][2 * (q & 1) + 1] = pack2<T>(v[2], v[3]);
    }
    st_h[tw] = nn < G::kHPix && inside && hy >= 1 && hy <= TH && hx >= 1 && hx <= 8;
    if (nn < G::kHPix) {
      char* hrow = smem + G::kH0 + rt * G::kHChunk;
      *reinterpret_cast<u32x4*>(hrow + img_off(hy * G::kHP + hx, 2 * hi)) = pk[tw][0];
      *reinterpret_cast<u32x4*>(hrow + img_off(hy * G::kHP + hx, 2 * hi + 1)) = pk[tw][1];
    }
  }
  lds_barrier();   // h complete
  RBW_STAMP(4);
  if (p.out_h) {
#pragma unroll
    for (int tw = 0; tw < G::NTW; ++tw) {
      if (st_h[tw]) {
        char* dst = p.out_h + (((size_t)n * p.H + (y0 - 1 + hyv[tw])) * p.W + (x0 - 1 + hxv[tw])) * 128 + (32 * rt + 16 * hi) * 2;
        tg_store16(dst, pk[tw][0]);
        tg_store16(dst + 16, pk[tw][1]);
      }
    }
  }
  RBW_STAMP(5);
  lds_barrier();   // (the conv2 waves' exchange barrier)
  RBW_STAMP(6);
}

template <typename T, int TH>
int launch_rbw(const RbwK& k, unsigned blocks, hipStream_t st) {
  auto fn = resblock_ws_kernel<T, TH>;
  static std::atomic<bool> attr_done{false};  // one-time function attribute (benign race: idempotent)
  if (!attr_done) {
    TG_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, GeoW<TH>::kLds));
    attr_done = true;
  }
  hipLaunchKernelGGL(fn, dim3(blocks), dim3(512), GeoW<TH>::kLds, st, k.in, k.w1, k.zero, k.H, k.W, k.tiles_x, k.tiles_y, k.w2, k.b1,
                     k.out_h, k.out_a, k.N, k.skip);
  return tg_launch_status();
}

}  // namespace

extern "C" int tg_resblock_fwd_ws(int dtype, const void* in, const void* w1_packed, const float* b1, const void* w2_packed,
                                  void* out_h, void* out_a, int N, int H, int W, int C, int add_skip, void* stream) {
  if (!in || !w1_packed || !b1 || !w2_packed || !out_a || N <= 0 || H <= 0 || W <= 0) return TG_E_BADARG;   // (out_h may be null)
  if ((dtype != TG_BF16 && dtype != TG_F16) || C != 64) return TG_E_UNSUPPORTED;
  if (!tg_aligned16(in) || !tg_aligned16(w1_packed) || !tg_aligned16(w2_packed) || (out_h && !tg_aligned16(out_h)) ||
      !tg_aligned16(out_a) || !tg_aligned16(b1))
    return TG_E_ALIGN;
  if ((long long)H * W * 128 > 0x7fffffffLL) return TG_E_UNSUPPORTED;   // 32-bit byte offsets inside an image
  static const char* zero_page = [] {
    void* z = nullptr;
    return hipGetSymbolAddress(&z, HIP_SYMBOL(tg_rbw_zero_page)) == hipSuccess ? (const char*)z : (const char*)nullptr;
  }();
  if (!zero_page) return TG_E_BADARG;
  RbwK k;
  k.in = (const char*)in; k.w1 = (const char*)w1_packed; k.b1 = b1; k.w2 = (const char*)w2_packed;
  k.out_h = (char*)out_h; k.out_a = (char*)out_a; k.zero = zero_page;
  k.N = N; k.H = H; k.W = W; k.skip = add_skip ? 1 : 0;
  k.tiles_x = (W + 7) / 8;
  // 8x4 tiles while 8x8 tiles would leave half of the chip's CUs without a workgroup (resblock.hip's rule)
  const long long blocks8 = (long long)k.tiles_x * ((H + 7) / 8) * N;
#ifdef RBW_TH8_MIN   // A/B build: 8 x 8 tiles from this many 8 x 8 tiles on (default rule: above 128)
  const int th = blocks8 < RBW_TH8_MIN ? 4 : 8;
#else
  const int th = blocks8 <= 128 ? 4 : 8;
#endif
  k.tiles_y = (H + th - 1) / th;
  const long long blocks = (long long)k.tiles_x * k.tiles_y * N;
  if (blocks > 0x7fffffffLL) return TG_E_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == TG_F16) return th == 4 ? launch_rbw<F16, 4>(k, (unsigned)blocks, st) : launch_rbw<F16, 8>(k, (unsigned)blocks, st);
  return th == 4 ? launch_rbw<BF16, 4>(k, (unsigned)blocks, st) : launch_rbw<BF16, 8>(k, (unsigned)blocks, st);
}
