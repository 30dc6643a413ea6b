// Weight gradients of MANY 3x3 stride-1 convolutions in ONE persistent launch (16-bit element types):
//
//   dW_j[t][a][b] = sum over (n, y, x) of  X_j[n, y + dy[t], x + dx[t]][a] * Y_j[n, y, x][b]        (zero padding)
//
// for a list of jobs j of possibly different image size and channel counts (the generator's 40 plain 3x3 layers, the
// 8 residual convs of a discriminator stage ...; channel counts are multiples of 32, a 32-channel remainder runs as a
// half-empty 64-channel block whose missing channels hold don't-care values the fold never reads).  The unit of work is (job, 64x64 channel block, 128-pixel tile); the
// units of all jobs form one list, and workgroup w takes the contiguous range [w * per, (w + 1) * per) of it.  A
// workgroup keeps the partial dW of its current (job, block) in accumulators and writes one fp32 slab whenever its
// range leaves that block - so the number of slabs is  workgroups + channel blocks  instead of
// launches x workgroups (tg_wgrad writes one slab per workgroup and LAUNCH: 23.5 MB per layer at 160 workgroups;
// 10 generator layers wrote - and the fold read back - 235 MB per step, this launch 47 MB).  Slab slot of a
// segment = workgroup index + global ordinal of the channel block, so the slots of one block are contiguous and
// tg_wgrad_finalize_multi folds them unchanged (one fold job per channel block).
//
// Inside a tile the structure follows wgrad_mfma.hip - both MFMA operands are transposed reads
// (ds_read_b64_tr_b16) of [pixel][channel] LDS images - with the staging rebuilt around it:
//   * LDS-DMA (global_load_lds_dwordx4) fills the images: no staging registers, no ds_write pass (930 of 7440 cycles
//     per tile in the register-staged kernel, tools/stamp_wgrad.py), and the loads of tile i+1 are issued into the
//     second LDS buffer at the head of tile i's k-loop: ONE barrier per tile, the matrix pipe never waits for a store
//     pass.  Out-of-image pixels read a 16-byte page of zeros.
//   * the images are unpadded 128-byte rows (DMA writes 1 KiB per wave-instruction, lane-linear) whose 32-byte
//     segment index is XOR-ed with bits 1-2 of the pixel's COLUMN (X) / row (Y): a 32-lane group of a transposed read
//     takes 32 bytes of each of 8 consecutive pixels -> 8 distinct segments of the 256-byte bank window at any tap
//     shift.  The swizzle lives on the DMA's per-lane SOURCE address.
//   * the key depends on the column only and every tile geometry is a template parameter, so the whole k-loop reads
//     through 3 (X: one per column tap) + 2 (Y) lane addresses with immediate offsets: no address arithmetic in the
//     loop (the old k-loop spent 4 v_mad_u64 + 26 v_add per 18 MFMAs and waited lgkmcnt(0) in front of every pair).
//
// Replaces aten::convolution_backward (weight path) behind code/train.py:336,340 for the layers of
// code/models.py:54-58,68-76 (generator) and :90-94 (discriminator residual blocks).
#include "common.h"

namespace {

constexpr int GJ = 12;  // int64 per job row (include/tecogan_hip.h, tg_wgrad_group)

#ifdef TG_STAMP
// Diagnostic build only (-DTG_STAMP, tools/stamp_wgroup.py): waves 0 and 7 of workgroup 0 record s_memtime around the phases of
// their first 8 tiles: [wave][tile][0 loop head | 1 DMA landed | 2 barrier passed | 3 next DMA issued | 4 k-loop done]
}  // namespace
__device__ long long tg_wg_stamps[2 * 8 * 5];
extern "C" int tg_debug_read_wg_stamps(long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(tg_wg_stamps), sizeof(long long) * n);
}
namespace {
#define WG_STAMP(t, ph)                                                                                        \
  do {                                                                                                         \
    if (blockIdx.x == 0 && (threadIdx.x == 0 || threadIdx.x == 448) && (t) < 8)                                \
      tg_wg_stamps[((threadIdx.x ? 1 : 0) * 8 + (t)) * 5 + (ph)] = (long long)__builtin_amdgcn_s_memtime();    \
  } while (0)
#else
#define WG_STAMP(t, ph) do {} while (0)
#endif

struct WgGroupK {
  const long long* jobs;
  float* slab;
  int njobs, units_total, per_wg;
};

// 16 bytes of zeros: the DMA source of every out-of-image piece
__device__ __attribute__((aligned(16))) unsigned int tg_wg_zero_page[4];

// One LDS-DMA wave-instruction: lane l's 16 bytes at `gsrc` land at LDS byte address lds_dst + 16 * l (lds_dst wave-uniform,
// in M0).  Inline asm on purpose: hipcc orders every ds_read behind ALL outstanding LDS-DMA it knows of (s_waitcnt vmcnt(0)
// in front of the k-loop, which drains the prefetch of the next tile); an asm load is absent from its bookkeeping, so the
// kernel counts it itself - dma_wait() in front of the barrier that precedes the reads.
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_dst)
      : "memory");
}
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// ... all but the n most recent vector-memory instructions of this wave have completed (n is wave-uniform, <= 7)
__device__ __forceinline__ void dma_wait_le(int n) {
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
  }
}

// Tile geometry.  S = 1 (3x3 stride-1 convs): tile TW x 128/TW output pixels, X patch (TW + 2) x (TH + 2) of pitch TW + 2.
// S = 2 (NTS = 3: conv-transpose k3 s2, X = the output gradient on the 2h grid; NTS = 4: conv k4 s2, X = the input): tile
// 16 x 4 pixels of Y, X patch rows 2 ty + dy, columns 2 tx + dx - the columns are DE-INTERLEAVED by parity into two sub-images
// of pitch 18, so that the 8 pixels a 32-lane group of a transposed read takes (consecutive tx under one tap) are 8 consecutive
// rows of one sub-image again and the column-keyed swizzle stays conflict-free.
template <int TW, int S, int NTS>
struct WgGeom {
  static constexpr int TH = S == 1 ? 128 / TW : 4;
  static constexpr int TPIX = TW * TH, KSTEPS = TPIX / 32;
  static constexpr int PITCH = S == 1 ? TW + 2 : 18;            // rows of the X image per patch row (even: swizzle parity)
  static constexpr int NR = S * (TH - 1) + NTS;                 // patch rows
  static constexpr int NC = S * (TW - 1) + NTS;                 // patch columns
  static constexpr int SUB = NR * PITCH;                        // rows of one sub-image
  static constexpr int XROWS = S * SUB;
  static constexpr int XCH = (XROWS * 128 + 1023) / 1024, YCH = TPIX / 8;
  static constexpr int XCW = (XCH + 7) / 8, YCW = (YCH + 7) / 8;
  static constexpr int XBYTES = XCH * 1024, BUF = XBYTES + YCH * 1024;
  // pixel k + 16 of the tile: 16 columns further (TW 32), else the next tile row = S patch rows down
  static constexpr int HI = (TW == 32) ? 16 * 128 : S * PITCH * 128;
  static constexpr int KROW = (TW == 32) ? 1 : 2;               // tile rows per 32-pixel k-step
  static_assert(PITCH % 2 == 0 && SUB % 2 == 0, "the swizzle key needs even pitches");
  static_assert(S == 1 || TW == 16, "stride-2 tiles are 16 x 4");
  // image row -> patch (row, column), swizzle key, validity
  __device__ static void decode(int row, int& py, int& px, int& key, bool& valid) {
    if constexpr (S == 1) {
      py = row / PITCH;
      px = row - py * PITCH;
      key = (px >> 1) & 3;
      valid = row < XROWS;
    } else {
      const int sub = row / SUB, rr = row - sub * SUB;
      py = rr / PITCH;
      const int c = rr - py * PITCH;
      px = 2 * c + sub;
      key = (c >> 1) & 3;
      valid = row < XROWS && px < NC;
    }
  }
  // lane address (bytes into the X image, without the 32-byte segment) of tile column tx under column tap d, and its key
  __device__ static void lane_col(int tx, int d, int& rowbytes, int& key) {
    if constexpr (S == 1) {
      const int px = tx + d;
      rowbytes = px * 128;
      key = (px >> 1) & 3;
    } else {
      const int c = tx + (d >> 1);
      rowbytes = ((d & 1) * SUB + c) * 128;
      key = (c >> 1) & 3;
    }
  }
  // byte offset of k-step s, row tap dy
  __device__ static constexpr int tap_row(int s, int dy) { return (S * KROW * s + dy) * PITCH * 128; }
};

// NB = 64-channel sub-blocks of Y per workgroup.  NB = 2 (64 x 128 channel blocks, layers with >= 128 Y channels): a wave owns
// 16 x 64 of the block (144 accumulator registers with 9 taps), a 32-pixel k-step is 8 Y + 18 X transposed reads for 36 MFMAs
// instead of 4 + 18 for 18, and X is fetched once per 128 Y channels instead of once per 64.  The Y image has 256-byte rows
// then: 8 segments, swizzle key = the pixel's low 3 bits (8 consecutive pixels -> 8 distinct segments of the bank window).
template <typename T, int TW, int S, int NTS, int NB>
__global__ __launch_bounds__(512) void wgrad_group_kernel(const WgGroupK p) {
  using Gm = WgGeom<TW, S, NTS>;
  constexpr int TH = Gm::TH, NT = NTS * NTS, XCH = Gm::XCH, XCW = Gm::XCW;
  constexpr int XBYTES = Gm::XBYTES, HI = Gm::HI, TPIX = Gm::TPIX;
  constexpr int CB = 64 * NB, YROW = 128 * NB, NF = 2 * NB;       // Y channels per block, bytes per Y image row, B fragments per wave
  constexpr int YCH = TPIX * NB / 8, YCW = (YCH + 7) / 8, BUF = XBYTES + YCH * 1024;
  constexpr int YPR = 8 * NB, YRC = 64 / YPR;                     // 16-byte pieces per Y row, rows per 1-KiB DMA chunk
  constexpr int kSlot = NT * 64 * CB + CB;  // floats per slab slot: [NT][64][CB] + channel sums of Y
  auto ykey = [](int k) { return NB == 1 ? (k >> 1) & 3 : k & 7; };

  // NBUF buffers of BUF bytes: two.  -DWG_NBUF3 (tools/build_variant.sh) builds the stride-1 geometries with three (3 x 39-42 KB),
  // the DMA of tile i + 2 issued at the head of tile i: measured 2-5 % SLOWER on every layer set (c6 91.6 -> 95.9 us at 256
  // workgroups, step 3.76 -> 3.77 ms; profiles/r04_z_wgrad_diag.log) - the kernel does run at the sum of its MFMA time (38 us for
  // c6) and its memory time (62 us without the matrix instructions = 400 MB of DMA at 6.4 TB/s), but DMA latency is not why.
  // Stamps (tools/stamp_wgroup.py, profiles/r04_z_stamp_wgroup_before.log): a tile is ~4800 ticks = wait 450 + barrier skew up to
  // 1200 + 1300-1700 in which all eight waves issue the next tile's 5 DMA instructions each and no MFMA runs + k-loop 1800-2600.
  // Spreading that issue over the k-loop (three buffers, waves 0-3 in front of k-steps 0 / 1, their SIMD partners in front of 2 / 3)
  // was built: the issue phase shrinks to 470 ticks and the k-loop GROWS by 1700 (5600 per tile, c6 94.8 -> 109.8 us, step 3.74 ->
  // 3.80 ms; profiles/r04_z_wgrad_inloop.log) - a DMA instruction between the transposed LDS reads costs more than one at the head.
#ifdef WG_NBUF3
  constexpr int NBUF = 3 * BUF <= 160 * 1024 ? 3 : 2;
#else
  constexpr int NBUF = 2;
#endif
  extern __shared__ __attribute__((aligned(1024))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wa = wid & 3, wb = wid >> 2;
  const int idx = lane & 15, g = lane >> 4, q = idx >> 2, pp = idx & 3;

  int u = blockIdx.x * p.per_wg;
  const int u_end = min(u + p.per_wg, p.units_total);
  if (u >= u_end) return;

  // ---- lane constants of the fragment reads (tile- and job-independent): one X address per column tap, two Y addresses
  int xa[NTS], ya[NF];
#pragma unroll
  for (int d = 0; d < NTS; ++d) {
    int rb, key;
    Gm::lane_col(4 * g + q, d, rb, key);
    xa[d] = rb + ((wa ^ key) * 32) + 8 * pp;
  }
#pragma unroll
  for (int b = 0; b < NF; ++b) {
    const int k = 4 * g + q;
    ya[b] = XBYTES + k * YROW + (((wb * NF + b) ^ ykey(k)) * 32) + 8 * pp;
  }
  // ---- lane constants of the DMA: chunk c of a wave covers image rows 8 * chunk + lane / 8, physical piece lane % 8
  const int jp = lane & 7, r8 = lane >> 3;
  int xpy[XCW], xpx[XCW], xch[XCW];  // patch row / column of my piece, byte offset of its channels in the pixel
#pragma unroll
  for (int c = 0; c < XCW; ++c) {
    int py, px, key;
    bool valid;
    Gm::decode((wid + 8 * c) * 8 + r8, py, px, key, valid);
    xpy[c] = valid ? py : -100000;  // rows behind the image (last chunk, pitch padding) load zeros
    xpx[c] = px;
    xch[c] = ((((jp >> 1) ^ key) * 2) + (jp & 1)) * 16;
  }
  int yty[YCW], ytx[YCW], ych[YCW];
#pragma unroll
  for (int c = 0; c < YCW; ++c) {
    const int k = (wid + 8 * c) * YRC + lane / YPR, yp = lane % YPR;
    yty[c] = k / TW;
    ytx[c] = k - yty[c] * TW;
    ych[c] = ((((yp >> 1) ^ ykey(k)) * 2) + (yp & 1)) * 16;
  }
  const char* zero = reinterpret_cast<const char*>(tg_wg_zero_page);
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  const int wid_u = __builtin_amdgcn_readfirstlane(wid);
  int dma_per_tile = 0;   // DMA instructions this wave issues per tile (wave-uniform): what may stay in flight behind a tile
#pragma unroll
  for (int c = 0; c < XCW; ++c) dma_per_tile += wid_u + 8 * c < XCH ? 1 : 0;
#pragma unroll
  for (int c = 0; c < YCW; ++c) dma_per_tile += wid_u + 8 * c < YCH ? 1 : 0;

  f32x4 acc[NT][NF];
  constexpr int E = 8;
  float bsum[E];

  // ---- job lookup
  int job = 0;
  while (job + 1 < p.njobs && (int)p.jobs[GJ * (job + 1) + 2] <= u) ++job;

  while (u < u_end) {
    const long long* jr = p.jobs + GJ * job;
    const char* xbase = reinterpret_cast<const char*>(jr[0]);
    const char* ybase = reinterpret_cast<const char*>(jr[1]);
    // H x W: the grid of Y (and of the tiles); X lives on the S*H x S*W grid
    const int ubeg = (int)jr[2], N = (int)jr[3], H = (int)jr[4], W = (int)jr[5], Cx = (int)jr[6], Cy = (int)jr[7];
    const int tiles_x = (int)jr[8], tiles_y = (int)jr[9], want_ysum = (int)jr[10], gb0 = (int)jr[11];
    const int XH = S * H, XW = S * W;
    const int tiles = tiles_x * tiles_y * N;
    const int b_blocks = (Cy + CB - 1) / CB, blocks = ((Cx + 63) >> 6) * b_blocks;  // a channel remainder is a part-empty block
    const int local = u - ubeg;
    const int blk = local / tiles;
    int tile = local - blk * tiles;
    const int seg_n = min(u_end - u, tiles - tile);  // tiles of this segment
    const int a0 = (blk / b_blocks) * 64, b0 = (blk % b_blocks) * CB;
    const bool ysum = want_ysum && a0 == 0;
    const int xpixb = Cx * 2, ypixb = Cy * 2;

#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int b = 0; b < NF; ++b) acc[t][b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < E; ++e) bsum[e] = 0.f;

    // per-segment DMA constants: byte offset of my pieces relative to the patch / tile origin (32-bit: patch-relative)
    int xrel[XCW], yrel[YCW];
    // A 32-channel remainder runs as a half-empty block: the pieces of its missing channels are CLAMPED onto the pixel's last
    // real 16 bytes instead of being masked off.  What they produce - rows a >= Cx - a0 / columns b >= Cy - b0 of the slab -
    // is never read (the fold takes real channels only) and cannot reach a real entry: dW[a][b] involves channels a, b alone.
    const int xlast = xpixb - a0 * 2 - 16, ylast = ypixb - b0 * 2 - 16;
#pragma unroll
    for (int c = 0; c < XCW; ++c) xrel[c] = (max(xpy[c], 0) * XW + xpx[c]) * xpixb + min(xch[c], xlast);
#pragma unroll
    for (int c = 0; c < YCW; ++c) yrel[c] = (yty[c] * W + ytx[c]) * ypixb + min(ych[c], ylast);
    // coordinates of the next tile to issue, advanced incrementally (no division per tile)
    int i_txb, i_tyb, i_n;
    {
      int r = tile;
      i_txb = r % tiles_x;
      r /= tiles_x;
      i_tyb = r % tiles_y;
      i_n = r / tiles_y;
    }

    auto issue = [&](int bufoff) {
      const int ty0 = i_tyb * TH, tx0 = i_txb * TW;
      const char* xo = xbase + ((long long)i_n * XH * XW + (long long)(S * ty0 - 1) * XW + (S * tx0 - 1)) * xpixb + a0 * 2;
      const unsigned lx = lds0 + bufoff;
#pragma unroll
      for (int c = 0; c < XCW; ++c) {
        if (wid_u + 8 * c < XCH) {  // wave-uniform
          const int iy = S * ty0 - 1 + xpy[c], ix = S * tx0 - 1 + xpx[c];
          const bool ok = (unsigned)iy < (unsigned)XH && (unsigned)ix < (unsigned)XW;
          glds16(ok ? xo + xrel[c] : zero, lx + (wid_u + 8 * c) * 1024);
        }
      }
      const char* yo = ybase + ((long long)i_n * H * W + (long long)ty0 * W + tx0) * ypixb + b0 * 2;
#pragma unroll
      for (int c = 0; c < YCW; ++c) {
        if (wid_u + 8 * c < YCH) {  // wave-uniform
          const bool ok = ty0 + yty[c] < H && tx0 + ytx[c] < W;
          glds16(ok ? yo + yrel[c] : zero, lx + XBYTES + (wid_u + 8 * c) * 1024);
        }
      }
      if (++i_txb == tiles_x) {
        i_txb = 0;
        if (++i_tyb == tiles_y) {
          i_tyb = 0;
          ++i_n;
        }
      }
    };

    int buf = 0;
    if (NBUF == 3) dma_wait();   // (the previous segment's slab stores: vmcnt counts DMA pieces only from here on)
    issue(0);
    if (NBUF == 3 && seg_n > 1) issue(BUF);
    for (int i = 0; i < seg_n; ++i, ++tile) {
      // my DMA pieces of this tile have landed (vmcnt: the next tile's may still be in flight) and every wave is past the
      // previous tile's reads (barrier) - whose buffer the DMA issued below overwrites
      WG_STAMP(i, 0);
      if (NBUF == 3) dma_wait_le(i + 1 < seg_n ? dma_per_tile : 0);
      else dma_wait();
      WG_STAMP(i, 1);
      __syncthreads();
      WG_STAMP(i, 2);
      const int bo = buf * BUF;
      if (NBUF == 3) {
        if (i + 2 < seg_n) issue((buf == 0 ? 2 : buf - 1) * BUF);
      } else if (i + 1 < seg_n) {
        issue(BUF - bo);
      }
      WG_STAMP(i, 3);
      if (ysum) {
        // bias gradient = sum over pixels of Y: thread (row tid / YPR [+ 512 / YPR ...], logical piece tid % YPR) adds its 8 channels
#pragma unroll
        for (int h = 0; h < TPIX / (512 / YPR); ++h) {
          const int k = tid / YPR + (512 / YPR) * h, lp = tid % YPR;
          const int phys = ((((lp >> 1) ^ ykey(k)) * 2) + (lp & 1)) * 16;
          const u32x4 v = *reinterpret_cast<const u32x4*>(smem + bo + XBYTES + k * YROW + phys);
          float f[8];
          Vec<T>::load(&v, f);
#pragma unroll
          for (int e = 0; e < E; ++e) bsum[e] += f[e];
        }
      }
      const char* base = smem + bo;
#pragma unroll
      for (int s = 0; s < Gm::KSTEPS; ++s) {
        bf16x8 bfr[NF];
#pragma unroll
        for (int b = 0; b < NF; ++b) {
          const char* yp = base + ya[b] + s * 32 * YROW;
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(yp));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(yp + 16 * YROW));
          typedef __attribute__((ext_vector_type(8))) short s16x8;
          const s16x8 cat = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          bfr[b] = __builtin_bit_cast(bf16x8, cat);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int dy = t / NTS, dx = t % NTS;
#ifdef WG_DIAG_NOAREAD   // diagnostic: every tap reads tap 0's fragment address (the loads collapse to one: wrong results)
          const char* xp = base + xa[0] + Gm::tap_row(s, 0);
#else
          const char* xp = base + xa[dx] + Gm::tap_row(s, dy);
#endif
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(xp));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(xp + HI));
          typedef __attribute__((ext_vector_type(8))) short s16x8;
          const s16x8 cat = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          const bf16x8 af = __builtin_bit_cast(bf16x8, cat);
#ifdef WG_DIAG_NOMFMA   // diagnostic builds (tools/build_variant.sh): LDS reads without the matrix instructions (wrong results)
#pragma unroll
          for (int b = 0; b < NF; ++b) acc[t][b][0] += __builtin_bit_cast(f32x4, af)[b & 3];
#else
#pragma unroll
          for (int b = 0; b < NF; ++b) acc[t][b] = Mma16<T>::run(af, bfr[b], acc[t][b]);
#endif
        }
      }
      WG_STAMP(i, 4);
      buf = NBUF == 3 ? (buf == 2 ? 0 : buf + 1) : buf ^ 1;
    }

    // ---- the segment's partial dW -> slab slot (workgroup + global ordinal of the channel block)
    float* slab = p.slab + (size_t)(blockIdx.x + gb0 + blk) * kSlot;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int b = 0; b < NF; ++b) {
        const int bch = (wb * NF + b) * 16 + idx;
#pragma unroll
        for (int j = 0; j < 4; ++j) tg_store4(&slab[(t * 64 + wa * 16 + 4 * g + j) * CB + bch], acc[t][b][j]);
      }
    __syncthreads();  // every wave is done with the LDS images (the next segment's DMA, or the sums below, overwrite them)
    if (want_ysum) {
      float* red = reinterpret_cast<float*>(smem);  // [512 / YPR rows][CB channels]
      if (ysum) {
#pragma unroll
        for (int e = 0; e < E; ++e) red[(tid / YPR) * CB + (tid % YPR) * 8 + e] = bsum[e];
      }
      __syncthreads();
      if (tid < CB) {
        float s = 0.f;
        if (ysum)
          for (int r = 0; r < 512 / YPR; ++r) s += red[r * CB + tid];
        slab[NT * 64 * CB + tid] = s;  // blocks with a0 != 0 write zeros: the fold reads the sums of block row 0 only
      }
      __syncthreads();
    }
    u += seg_n;
    if (u < u_end && u >= ubeg + blocks * tiles) ++job;  // (a segment never crosses a job; ranges may)
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Wave-specialised form for the plain 3x3 layers (round 4; S = 1, 64 x 64 channel blocks - same units, slabs and fold as above).
// Stamps of the kernel above (tools/stamp_wgroup.py, profiles/r04_z_stamp_wgroup_before.log): a 128-pixel tile takes ~4800 ticks
// for 2304 of MFMA - in 1300-1700 of them all eight waves issue the next tile's DMA (5-6 instructions each through the CU's one
// address path) and the matrix pipes idle; spreading those instructions over the k-loop is worse (r04_z_wgrad_inloop.log).  Here
// the roles are split as in conv3_rw.hip:
//   * waves 0-3, CONSUMERS (one per SIMD): wave w owns rows 16w .. 16w + 15 of the block and ALL 64 Y channels - 9 taps x 4 = 36
//     accumulator tiles (144 registers, which the unified wave could not afford beside its DMA addresses: r03_m) - so a k-step is
//     8 Y + 18 X transposed reads for 36 MFMAs instead of 4 + 18 for 18, and the matrix pipe of a SIMD is fed by one wave that
//     never issues a vector-memory instruction;
//   * waves 4-7, PRODUCERS: wait for their DMA pieces of tile i, pass the barrier, issue tile i + 1 into the other buffer
//     (11 instructions per wave, nobody to compete with) and add up the bias gradient from the Y image.
template <typename T, int TW, int S>
__global__ __launch_bounds__(512) void wgrad_group_ws_kernel(const WgGroupK p) {
  using Gm = WgGeom<TW, S, 3>;   // S = 1: plain 3x3 layers; S = 2: conv-transpose k3 s2 (X = the output gradient on the fine grid)
  static_assert(S == 1 || TW == 16, "stride-2 tiles are 16 x 4");
  constexpr int NTS = 3, TH = Gm::TH, NT = 9, XCH = Gm::XCH, XBYTES = Gm::XBYTES, TPIX = Gm::TPIX;
  constexpr int CB = 64, YROW = 128, NF = 4;
  constexpr int YCH = TPIX / 8, BUF = XBYTES + YCH * 1024;
  constexpr int XCW = (XCH + 3) / 4, YCW = (YCH + 3) / 4;   // DMA instructions per PRODUCER wave and tile
  constexpr int YPR = 8, YRC = 8, PROWS = 256 / YPR;         // bias sums: 256 producer threads = 32 rows x 8 pieces
  constexpr int kSlot = NT * 64 * CB + CB;
  constexpr int E = 8;
  auto ykey = [](int k) { return (k >> 1) & 3; };
  extern __shared__ __attribute__((aligned(1024))) char smem[];   // two buffers of BUF bytes

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool consumer = wid < 4;
  const int idx = lane & 15, g = lane >> 4, q = idx >> 2, pp = idx & 3;

  int u = blockIdx.x * p.per_wg;
  const int u_end = min(u + p.per_wg, p.units_total);
  if (u >= u_end) return;
  const char* zero = reinterpret_cast<const char*>(tg_wg_zero_page);
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

  int job = 0;
  while (job + 1 < p.njobs && (int)p.jobs[GJ * (job + 1) + 2] <= u) ++job;

  while (u < u_end) {
    const long long* jr = p.jobs + GJ * job;
    const char* xbase = reinterpret_cast<const char*>(jr[0]);
    const char* ybase = reinterpret_cast<const char*>(jr[1]);
    const int ubeg = (int)jr[2], N = (int)jr[3], H = (int)jr[4], W = (int)jr[5], Cx = (int)jr[6], Cy = (int)jr[7];
    const int tiles_x = (int)jr[8], tiles_y = (int)jr[9], want_ysum = (int)jr[10], gb0 = (int)jr[11];
    (void)N;
    const int XH = S * H, XW = S * W;
    const int tiles = tiles_x * tiles_y * N;
    const int b_blocks = (Cy + CB - 1) / CB, blocks = ((Cx + 63) >> 6) * b_blocks;
    const int local = u - ubeg;
    const int blk = local / tiles;
    int tile = local - blk * tiles;
    const int seg_n = min(u_end - u, tiles - tile);
    const int a0 = (blk / b_blocks) * 64, b0 = (blk % b_blocks) * CB;
    const bool ysum = want_ysum && a0 == 0;
    float* slab = p.slab + (size_t)(blockIdx.x + gb0 + blk) * kSlot;
    float bsum[E];
#pragma unroll
    for (int e = 0; e < E; ++e) bsum[e] = 0.f;

    if (consumer) {
      // ======================================================================================================= CONSUMER
      const int wa = wid;
      int xa[NTS], ya[NF];
#pragma unroll
      for (int d = 0; d < NTS; ++d) {
        int rb, key;
        Gm::lane_col(4 * g + q, d, rb, key);
        xa[d] = rb + ((wa ^ key) * 32) + 8 * pp;
      }
#pragma unroll
      for (int b = 0; b < NF; ++b) {
        const int k = 4 * g + q;
        ya[b] = XBYTES + k * YROW + ((b ^ ykey(k)) * 32) + 8 * pp;
      }
      f32x4 acc[NT][NF];
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int b = 0; b < NF; ++b) acc[t][b] = f32x4{0.f, 0.f, 0.f, 0.f};
      int buf = 0;
      for (int i = 0; i < seg_n; ++i) {
        WG_STAMP(i, 0);
        WG_STAMP(i, 1);
        __syncthreads();   // tile i is in LDS (the producers waited for it), everybody is past tile i - 1
        WG_STAMP(i, 2);
        WG_STAMP(i, 3);
        const char* base = smem + buf * BUF;
        // k-loop, software-pipelined by hand (left to itself the compiler hoists every fragment read of the tile above the first
        // MFMA: 316 spilled registers).  A k-step is 32 pixels; its X fragments are pairs of HALF fragments ("slots"): TW = 16:
        // slot j = patch row 2s + j (4 per k-step, tap row dy uses slots dy, dy + 1); TW = 32: slot 2r + h = half h of row s + r (6 per
        // k-step, tap row dy uses slots 2dy, 2dy + 1); stride 2: slot j = patch row 4s + j (5 per k-step, tap row dy uses slots dy,
        // dy + 2).  Order per k-step: [reads for dy 1 | 12 MFMAs of dy 0] [reads for dy 2 | MFMAs
        // of dy 1] [Y fragments and dy-0 slots of the NEXT k-step | MFMAs of dy 2] - every read has 12 MFMAs (~200 cycles) to land.
        constexpr int NS = S == 2 ? 5 : TW == 16 ? 4 : 6;
        auto slot_off = [](int s_, int j) {   // byte offset of slot j of k-step s_ (compile-time after unrolling)
          return S == 2 ? (4 * s_ + j) * Gm::PITCH * 128
                        : TW == 16 ? (2 * s_ + j) * Gm::PITCH * 128 : (s_ + j / 2) * Gm::PITCH * 128 + (j % 2) * 16 * 128;
        };
        auto lo_slot = [](int dy) { return (S == 2 || TW == 16) ? dy : 2 * dy; };
        auto hi_slot = [](int dy) { return S == 2 ? dy + 2 : TW == 16 ? dy + 1 : 2 * dy + 1; };
        s16x4 xs[2][NS][NTS];
        s16x4 ys[2][NF][2];
        auto read_y = [&](int s_) {
#pragma unroll
          for (int b = 0; b < NF; ++b) {
            const char* yp = base + ya[b] + s_ * 32 * YROW;
            ys[s_ & 1][b][0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(yp));
            ys[s_ & 1][b][1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(yp + 16 * YROW));
          }
        };
        auto read_slot = [&](int s_, int j) {
#pragma unroll
          for (int d = 0; d < NTS; ++d)
            xs[s_ & 1][j][d] =
                __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(base + xa[d] + slot_off(s_, j)));
        };
        typedef __attribute__((ext_vector_type(8))) short s16x8;
        auto cat8 = [](s16x4 lo, s16x4 hi) {
          const s16x8 c = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          return __builtin_bit_cast(bf16x8, c);
        };
        read_y(0);
        read_slot(0, lo_slot(0));
        read_slot(0, hi_slot(0));
#pragma unroll
        for (int s_ = 0; s_ < Gm::KSTEPS; ++s_) {
#pragma unroll
          for (int dy = 0; dy < NTS; ++dy) {
            // reads that the NEXT group of MFMAs needs
            if (dy + 1 < NTS) {
              if (S == 2) {
                if (dy == 0) read_slot(s_, 1);   // tap row 1: slots 1, 3; tap row 2: slots 2 (in registers since row 0), 4
                read_slot(s_, dy == 0 ? 3 : 4);
              } else if (TW == 16) {
                read_slot(s_, dy + 2);
              } else {
                read_slot(s_, 2 * dy + 2);
                read_slot(s_, 2 * dy + 3);
              }
            } else if (s_ + 1 < Gm::KSTEPS) {
              read_y(s_ + 1);
              read_slot(s_ + 1, lo_slot(0));
              read_slot(s_ + 1, hi_slot(0));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int dx = 0; dx < NTS; ++dx) {
              const bf16x8 af = cat8(xs[s_ & 1][lo_slot(dy)][dx], xs[s_ & 1][hi_slot(dy)][dx]);
#pragma unroll
              for (int b = 0; b < NF; ++b)
                acc[dy * NTS + dx][b] = Mma16<T>::run(af, cat8(ys[s_ & 1][b][0], ys[s_ & 1][b][1]), acc[dy * NTS + dx][b]);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        WG_STAMP(i, 4);
        buf ^= 1;
      }
      // the segment's partial dW -> slab slot (workgroup + global ordinal of the channel block).  (The lane's offset goes through
      // an empty asm: otherwise the 144 store addresses are computed - and spilled - in front of the tile loop.  Transposing the
      // tiles through LDS for 1-KB coalesced stores was built too and measured equal: profiles/r04_z_wgrad_ws.log.)
      int lane_off = (wa * 16 + 4 * g) * CB + idx;
      asm volatile("" : "+v"(lane_off));
      float* const sl = slab + lane_off;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int b = 0; b < NF; ++b) {
#pragma unroll
          for (int j = 0; j < 4; ++j) tg_store4(&sl[(t * 64 + j) * CB + b * 16], acc[t][b][j]);
        }
    } else {
      // ======================================================================================================= PRODUCER
      const int pw = wid - 4, ptid = tid - 256;
      const int xpixb = Cx * 2, ypixb = Cy * 2;
      const int jp = lane & 7, r8 = lane >> 3;
      // A 32-channel remainder runs as a half-empty block: the pieces of its missing channels are CLAMPED onto the pixel's last
      // real 16 bytes (what they produce is never read by the fold, see the kernel above)
      const int xlast = xpixb - a0 * 2 - 16, ylast = ypixb - b0 * 2 - 16;
      int xpy[XCW], xpx[XCW], xrel[XCW];
#pragma unroll
      for (int c = 0; c < XCW; ++c) {
        int py, px, key;
        bool valid;
        Gm::decode((pw + 4 * c) * 8 + r8, py, px, key, valid);
        xpy[c] = valid ? py : -100000;
        xpx[c] = px;
        const int xch = ((((jp >> 1) ^ key) * 2) + (jp & 1)) * 16;
        xrel[c] = (max(xpy[c], 0) * XW + xpx[c]) * xpixb + min(xch, xlast);
      }
      int yty[YCW], ytx[YCW], yrel[YCW];
#pragma unroll
      for (int c = 0; c < YCW; ++c) {
        const int k = (pw + 4 * c) * YRC + lane / YPR, yp = lane % YPR;
        yty[c] = k / TW;
        ytx[c] = k - yty[c] * TW;
        const int ych = ((((yp >> 1) ^ ykey(k)) * 2) + (yp & 1)) * 16;
        yrel[c] = (yty[c] * W + ytx[c]) * ypixb + min(ych, ylast);
      }
      int i_txb, i_tyb, i_n;
      {
        int r = tile;
        i_txb = r % tiles_x;
        r /= tiles_x;
        i_tyb = r % tiles_y;
        i_n = r / tiles_y;
      }
      auto issue = [&](int bufoff) {
        const int ty0 = i_tyb * TH, tx0 = i_txb * TW;
        const char* xo = xbase + ((long long)i_n * XH * XW + (long long)(S * ty0 - 1) * XW + (S * tx0 - 1)) * xpixb + a0 * 2;
        const unsigned lx = lds0 + bufoff;
#pragma unroll
        for (int c = 0; c < XCW; ++c) {
          if (pw + 4 * c < XCH) {  // wave-uniform
            const int iy = S * ty0 - 1 + xpy[c], ix = S * tx0 - 1 + xpx[c];
            const bool ok = (unsigned)iy < (unsigned)XH && (unsigned)ix < (unsigned)XW;
            glds16(ok ? xo + xrel[c] : zero, lx + (pw + 4 * c) * 1024);
          }
        }
        const char* yo = ybase + ((long long)i_n * H * W + (long long)ty0 * W + tx0) * ypixb + b0 * 2;
#pragma unroll
        for (int c = 0; c < YCW; ++c) {
          if (pw + 4 * c < YCH) {  // wave-uniform
            const bool ok = ty0 + yty[c] < H && tx0 + ytx[c] < W;
            glds16(ok ? yo + yrel[c] : zero, lx + XBYTES + (pw + 4 * c) * 1024);
          }
        }
        if (++i_txb == tiles_x) {
          i_txb = 0;
          if (++i_tyb == tiles_y) {
            i_tyb = 0;
            ++i_n;
          }
        }
      };
      int buf = 0;
      issue(0);
      for (int i = 0; i < seg_n; ++i) {
        WG_STAMP(i, 0);
        dma_wait();        // my pieces of tile i have landed
        WG_STAMP(i, 1);
        __syncthreads();   // ... everybody's have; the consumers are past tile i - 1, whose buffer the next DMA overwrites
        WG_STAMP(i, 2);
        const int bo = buf * BUF;
        if (i + 1 < seg_n) issue(BUF - bo);
        WG_STAMP(i, 3);
        if (ysum) {
          // bias gradient = sum over pixels of Y: thread (row ptid / 8 [+ 32 ...], logical piece ptid % 8) adds its 8 channels
#pragma unroll
          for (int h = 0; h < TPIX / PROWS; ++h) {
            const int k = ptid / YPR + PROWS * h, lp = ptid % YPR;
            const int phys = ((((lp >> 1) ^ ykey(k)) * 2) + (lp & 1)) * 16;
            const u32x4 v = *reinterpret_cast<const u32x4*>(smem + bo + XBYTES + k * YROW + phys);
            float f[8];
            Vec<T>::load(&v, f);
#pragma unroll
            for (int e = 0; e < E; ++e) bsum[e] += f[e];
          }
        }
        WG_STAMP(i, 4);
        buf ^= 1;
      }
    }
    __syncthreads();  // every wave is done with the LDS images (the next segment's DMA, or the sums below, overwrite them)
    if (want_ysum) {
      float* red = reinterpret_cast<float*>(smem);  // [PROWS rows][CB channels]
      if (ysum && !consumer) {
        const int ptid = tid - 256;
#pragma unroll
        for (int e = 0; e < E; ++e) red[(ptid / YPR) * CB + (ptid % YPR) * 8 + e] = bsum[e];
      }
      __syncthreads();
      if (tid < CB) {
        float sum = 0.f;
        if (ysum)
          for (int r = 0; r < PROWS; ++r) sum += red[r * CB + tid];
        slab[NT * 64 * CB + tid] = sum;  // blocks with a0 != 0 write zeros: the fold reads the sums of block row 0 only
      }
      __syncthreads();
    }
    u += seg_n;
    if (u < u_end && u >= ubeg + blocks * tiles) ++job;  // (a segment never crosses a job; ranges may)
  }
}

template <typename T, int TW, int S = 1>
int launch_group_ws(const WgGroupK& k, int nwg, hipStream_t st) {
  auto fn = wgrad_group_ws_kernel<T, TW, S>;
  constexpr int lds = 2 * (WgGeom<TW, S, 3>::XBYTES + WgGeom<TW, S, 3>::TPIX * 128);
  static_assert(lds <= 160 * 1024, "two LDS buffers must fit");
  static std::atomic<bool> attr_done{false};
  if (!attr_done) {
    TG_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    attr_done = true;
  }
  hipLaunchKernelGGL(fn, dim3((unsigned)nwg), dim3(512), lds, st, k);
  return tg_launch_status();
}

template <typename T, int TW, int S, int NTS, int NB = 1>
int launch_group(const WgGroupK& k, int nwg, hipStream_t st) {
  auto fn = wgrad_group_kernel<T, TW, S, NTS, NB>;
  constexpr int one = WgGeom<TW, S, NTS>::XBYTES + WgGeom<TW, S, NTS>::TPIX * NB * 128;
#ifdef WG_NBUF3
  constexpr int lds = (3 * one <= 160 * 1024 ? 3 : 2) * one;   // (NBUF of the kernel)
#else
  constexpr int lds = 2 * one;
#endif
  static_assert(lds <= 160 * 1024, "the LDS buffers must fit");
  static std::atomic<bool> attr_done{false};
  if (!attr_done) {
    TG_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    attr_done = true;
  }
  hipLaunchKernelGGL(fn, dim3((unsigned)nwg), dim3(512), lds, st, k);
  return tg_launch_status();
}

template <typename T>
int dispatch_group(int variant, int tile_w, const WgGroupK& k, int nwg, hipStream_t st) {
#ifdef WG_NO_WS   // A/B build (tools/build_variant.sh): the unified-wave kernel for the plain 3x3 layers too
  if (variant == TG_WGROUP_C3) return tile_w == 32 ? launch_group<T, 32, 1, 3>(k, nwg, st) : launch_group<T, 16, 1, 3>(k, nwg, st);
#else
  // the wave-specialised kernel from 12 tiles per workgroup (below that its four consumer waves' serial slab write and first-tile
  // wait cost more than the tile loop gains: the discriminator's 16 x 16 lists, 8 tiles each, 17.8 -> 19.9 us)
  if (variant == TG_WGROUP_C3 && k.per_wg >= 12)
    return tile_w == 32 ? launch_group_ws<T, 32>(k, nwg, st) : launch_group_ws<T, 16>(k, nwg, st);
  if (variant == TG_WGROUP_C3) return tile_w == 32 ? launch_group<T, 32, 1, 3>(k, nwg, st) : launch_group<T, 16, 1, 3>(k, nwg, st);
#ifdef WG_CT_WS   // the conv-transpose kind (S = 2, 64-pixel tiles of two k-steps) on the wave-specialised kernel: parity green, measured
                  // slightly SLOWER than the unified kernel (G backward alone 1.316 -> 1.335 ms, profiles/r04_z_wgrad_ws.log) - opt-in build
  if (variant == TG_WGROUP_CT && k.per_wg >= 24) return launch_group_ws<T, 16, 2>(k, nwg, st);
#endif
#endif
#ifdef TG_EXPERIMENTS   // 64 x 128 channel blocks: 1.1-2.5x slower (spills), profiles/r03_m_wgrad_b128.log
  if (variant == TG_WGROUP_C3_B128)
    return tile_w == 32 ? launch_group<T, 32, 1, 3, 2>(k, nwg, st) : launch_group<T, 16, 1, 3, 2>(k, nwg, st);
  if (variant == TG_WGROUP_CT_B128) return launch_group<T, 16, 2, 3, 2>(k, nwg, st);
#endif
  if (variant == TG_WGROUP_CT) return launch_group<T, 16, 2, 3>(k, nwg, st);
  return launch_group<T, 16, 2, 4>(k, nwg, st);
}

}  // namespace

extern "C" int64_t tg_wgrad_group_slot_floats_v(int variant) {
  if (variant == TG_WGROUP_C3 || variant == TG_WGROUP_CT) return 9 * 64 * 64 + 64;
  if (kTgExperiments && (variant == TG_WGROUP_C3_B128 || variant == TG_WGROUP_CT_B128)) return 9 * 64 * 128 + 128;
  if (variant == TG_WGROUP_C4S2) return 16 * 64 * 64 + 64;
  return TG_E_BADARG;
}

namespace {
int group_launch(int dtype, int variant, int tile_w, const int64_t* jobs_dev, int njobs, int units_total, int workgroups,
                 float* slab, void* stream) {
  if (!jobs_dev || !slab || njobs <= 0 || units_total <= 0 || workgroups <= 0) return TG_E_BADARG;
  if (!tg_aligned16(slab)) return TG_E_ALIGN;
  if (dtype != TG_BF16 && dtype != TG_F16) return TG_E_UNSUPPORTED;
  if (!kTgExperiments && (variant == TG_WGROUP_C3_B128 || variant == TG_WGROUP_CT_B128)) return TG_E_UNSUPPORTED;
  if (variant == TG_WGROUP_C3 || variant == TG_WGROUP_C3_B128) {
    if (tile_w != 32 && tile_w != 16) return TG_E_UNSUPPORTED;
  } else if (variant == TG_WGROUP_CT || variant == TG_WGROUP_C4S2 || variant == TG_WGROUP_CT_B128) {
    if (tile_w != 16) return TG_E_UNSUPPORTED;
  } else {
    return TG_E_BADARG;
  }
  WgGroupK k;
  k.jobs = reinterpret_cast<const long long*>(jobs_dev);
  k.slab = slab;
  k.njobs = njobs;
  k.units_total = units_total;
  k.per_wg = (units_total + workgroups - 1) / workgroups;
  const int nwg = (units_total + k.per_wg - 1) / k.per_wg;
  hipStream_t st = (hipStream_t)stream;
  return dtype == TG_BF16 ? dispatch_group<BF16>(variant, tile_w, k, nwg, st) : dispatch_group<F16>(variant, tile_w, k, nwg, st);
}
}  // namespace

extern "C" int tg_wgrad_group_v(int dtype, int variant, int tile_w, const int64_t* jobs_dev, int njobs, int units_total,
                                int workgroups, float* slab, void* stream) {
  return group_launch(dtype, variant, tile_w, jobs_dev, njobs, units_total, workgroups, slab, stream);
}
