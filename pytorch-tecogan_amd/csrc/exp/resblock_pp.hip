// The residual block's two input-gradients in ONE persistent launch, tile-pipelined (round 5; 16-bit, 64 channels, 8 x 4 tiles):
//
//     dH  = relu'(h) * conv3x3^T(dOut, W2)          dIn = dOut + conv3x3^T(dH, W1)          (autograd of code/ops.py:45-54;
//                                                                                            code/train.py:336: loss.backward())
// In the batched generator backward (40 samples = 1280 tiles) the trunk's 32 input-gradients ran as 32 launches of the
// register-weights kernel (9.2 us each: 2-3 tiles per workgroup behind a 2-us weight prologue).  resblock_ws.hip's structure - W1 in
// an LDS image for 32x32x16 tiles on four conv1 waves, W2 in the registers of four conv2 waves - is bound by the 156 KB a workgroup
// takes in; a launch with MANY tiles per workgroup amortises them: here a persistent workgroup keeps both weight sets for all its
// tiles and streams only the 12 x 8 patch (12 KB) and the relu mask per tile, and the two wave groups work on DIFFERENT tiles:
// conv1 waves on tile i while the conv2 waves finish tile i - 1 (patch, h and the K-half exchange are double-buffered).
// The groups meet through monotonic LDS counters (resblock2_ws.hip's finding: s_barrier would make each group wait for the other's
// vector-memory queue): [0] patch landed (conv2 waves, 4 per tile) | [1] h written (conv1 waves, 4 per tile) | [2] conv2 waves
// through with a tile | [3 + rt] exchange written (2 per tile).
// Operand roles follow resblock.hip's BWD form: stage 1 = transposed conv with the role-swapped packing of W2 (weight slot 8 - t
// goes with spatial offset t), relu mask = the forward's h; stage 2 = transposed conv with W1's, skip = dOut.
// LDS images, fragment maps and the row order of the W1 image are resblock_ws.hip's (tests/test_resblock_ws_maps_cpu.py).
#include "common.h"
#include "rbw_common.h"
#include <type_traits>

__device__ __attribute__((aligned(16))) unsigned int tg_rbp_zero_page[4];

#ifdef TG_STAMP
// Diagnostic build only: waves 0 (conv1) and 4 (conv2) of workgroup 0 record s_memtime at 8 points of tiles 2 .. 4 (kept in registers
// until the wave's end: a store per stamp would sit in front of the kernel's own vmcnt waits)
__device__ long long tg_rbp_stamps[2 * 3 * 8];
#define RBP_DECL long long rbp_st[24]; _Pragma("unroll") for (int q_ = 0; q_ < 24; ++q_) rbp_st[q_] = 0
#define RBP_STAMP(it, k) do { if ((it) >= 2 && (it) < 5) { const long long t_ = (long long)__builtin_amdgcn_s_memtime(); \
    _Pragma("unroll") for (int q_ = 0; q_ < 3; ++q_) if ((it) - 2 == q_) rbp_st[q_ * 8 + (k)] = t_; } } while (0)
#define RBP_FLUSH(role) do { if (blockIdx.x == 0 && (threadIdx.x & 63) == 0 && ((threadIdx.x >> 6) & 3) == 0) { \
    _Pragma("unroll") for (int q_ = 0; q_ < 24; ++q_) tg_rbp_stamps[(role) * 24 + q_] = rbp_st[q_]; } } while (0)
extern "C" int tg_debug_read_rbp_stamps(long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(tg_rbp_stamps), sizeof(long long) * n);
}
#else
#define RBP_DECL do {} while (0)
#define RBP_STAMP(it, k) do {} while (0)
#define RBP_FLUSH(role) do {} while (0)
#endif

namespace {

constexpr int TH = 4;
constexpr int kPH = TH + 4, kMain = 80, kPRows = 96, kPBytes = 2 * kPRows * 64, kNPD = kPRows / 8;   // patch 12 x 8: 12 KB, 12 DMA blocks
constexpr int kHPix = (TH + 2) * 10, kHP = 24, kHChunk = kHP * (TH + 2) * 64, kHBytes = 2 * kHChunk;    // h region 10 x 6: 18 KB
constexpr int kW1Bytes = 18 * 4096;
// THREE patch buffers: tile i + 2's patch is issued while tile i is being worked on (with two, a buffer comes free only when the conv2
// waves are through with tile i - 1, i.e. one tile ahead, and the DMA's ~2000 ticks of latency sat in every tile: 5800 ticks per tile in
// the first measurement of this file); two h buffers; ONE exchange buffer (a wave writes it again only behind counter [2] of the tile before)
constexpr int kNPB = 3;
constexpr int kP0 = kW1Bytes, kH0 = kP0 + kNPB * kPBytes, kX0 = kH0 + 2 * kHBytes, kXBytes = 8192, kSync = kX0 + kXBytes;
constexpr int kLds = kSync + 64;   // 155 712
static_assert(kLds <= 160 * 1024 && kNPD == 12, "LDS / patch blocks");

__device__ __forceinline__ int patch_row(int py, int px) {
  const int v = 10 * py + px;
  return px < 10 ? v : kMain + (v & 15);   // (8 patch rows: one extra block)
}

struct RbpK {
  const char* in;     // dOut
  const char* w1;     // stage 1: dgrad packing of W2
  const char* w2;     // stage 2: dgrad packing of W1
  const char* mask;   // h of the forward pass
  char* out_h;        // dH
  char* out_a;        // dIn
  const char* zero;
  int N, H, W, tiles_x, tiles_y, ntiles;
};

template <typename T>
__global__ __launch_bounds__(512) void resblock_pp_bwd_kernel(const RbpK p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  const int l32 = lane & 31, hi = lane >> 5;
  RBP_DECL;
  const int ntl = (p.ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;   // this workgroup's tiles: blockIdx.x, + gridDim.x, ...
  if (wid == 0 && lane < 16) reinterpret_cast<unsigned*>(smem + kSync)[lane] = 0u;   // (visible to all behind the s_barrier below)
  auto arrive = [&](int c) {
    if (lane == 0) atomicAdd(reinterpret_cast<unsigned*>(smem + kSync) + c, 1u);
  };
  auto await = [&](int c, unsigned need) {   // bounded: every arrival is unconditional, a bug must not hang the GPU
    const unsigned a = lds0 + kSync + 4 * c;
    for (int spin = 0; spin < (1 << 20); ++spin) {
      unsigned v;
      asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
      if ((unsigned)__builtin_amdgcn_readfirstlane((int)v) >= need) break;
      __builtin_amdgcn_s_sleep(1);
    }
  };
  struct Tile { int n, y0, x0; };
  auto tile_of = [&](int i) {
    int t = (int)blockIdx.x + i * (int)gridDim.x;
    Tile r;
    const int txb = t % p.tiles_x;
    t /= p.tiles_x;
    r.y0 = (t % p.tiles_y) * TH;
    r.n = t / p.tiles_y;
    r.x0 = txb * 8;
    return r;
  };

  // ---- W1 image, all eight waves: wave (u = wid % 4, chunk = wid / 4) brings quarter u of block (tap so, chunk) for every tap; the
  // block of spatial offset so holds weight slot 8 - so.  Row order / swizzle: resblock_ws.hip.
  {
    const int u = wid & 3, cw = wid >> 2;
    const int mp = 16 * u + (lane >> 2);
    const int rt_ = mp >> 5, m = mp & 31, j = m >> 3, hm = (m >> 2) & 1, e = m & 3;
    const int R = 16 * (2 * rt_ + (j & 1)) + 4 * (2 * hm + (j >> 1)) + e;
    const char* src = p.w1 + cw * 4096 + R * 64 + (((lane & 3) ^ ((mp >> 2) & 3)) << 4);
    const unsigned dst = lds0 + cw * 4096 + u * 1024;
#pragma unroll
    for (int so = 0; so < 9; ++so) glds16(src + (8 - so) * 8192, dst + so * 8192);
  }

  if (wid < 4) {
    // =============================================================================================== CONV1 WAVES: (rt, pg), one 32-pixel tile each
    const int rt = wid & 1, pg = wid >> 1;
    const int a0 = img_off(32 * rt + l32, hi);
    const int n0 = 32 * pg + l32, nn = n0 < kHPix ? n0 : n0 - 32;
    const int hy = (nn * 205) >> 11, hx = nn - 10 * hy;
    int xa[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) xa[t] = patch_off(patch_row(hy + t / 3, hx + t % 3), 0, hi);
    const int hoff = rt * kHChunk + img_off(hy * kHP + hx, 2 * hi);
    const int hoff1 = rt * kHChunk + img_off(hy * kHP + hx, 2 * hi + 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();   // W1, patch 0 and the conv2 waves' W2 fragments are in
    // the relu mask of the lane's pixel, fetched ONE TILE AHEAD (clamped address: unused outside the image)
    auto mask_of = [&](int i, u32x4& m0, u32x4& m1, size_t& pix, bool& inside) {
      const Tile tl = tile_of(i);
      const int y = tl.y0 - 1 + hy, x = tl.x0 - 1 + hx;
      inside = ((unsigned)y < (unsigned)p.H) & ((unsigned)x < (unsigned)p.W);
      pix = ((size_t)tl.n * p.H + min(max(y, 0), p.H - 1)) * p.W + min(max(x, 0), p.W - 1);
      m0 = *reinterpret_cast<const u32x4*>(p.mask + pix * 128 + (32 * rt + 16 * hi) * 2);
      m1 = *reinterpret_cast<const u32x4*>(p.mask + pix * 128 + (32 * rt + 16 * hi) * 2 + 16);
    };
    u32x4 nm0, nm1;
    size_t npix;
    bool ninside;
    mask_of(0, nm0, nm1, npix, ninside);
    for (int i = 0; i < ntl; ++i) {
      const int buf = i & 1, pbuf = i % kNPB;
      const u32x4 hm0 = nm0, hm1 = nm1;
      const size_t pix = npix;
      const bool inside = ninside;
      RBP_STAMP(i, 0);
      if (i + 1 < ntl) mask_of(i + 1, nm0, nm1, npix, ninside);
      if (i > 0) await(0, 4u * i);            // patch i has landed
      RBP_STAMP(i, 1);
      if (i >= 2) await(2, 4u * (i - 1));     // the conv2 waves are through with tile i - 2: its h buffer is free
      RBP_STAMP(i, 2);
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      bf16x8 af[3], bf[3];
      const char* const pb = smem + kP0 + pbuf * kPBytes;
      auto frags = [&](int s, int b) {
        const int t = s >> 2, kc = (s >> 1) & 1, x32 = (s & 1) * 32;
        af[b] = *reinterpret_cast<const bf16x8*>(smem + (2 * t + kc) * 4096 + (a0 ^ x32));
        bf[b] = *reinterpret_cast<const bf16x8*>(pb + kc * 512 + (xa[t] ^ x32));
      };
      frags(0, 0);
      frags(1, 1);
#pragma unroll
      for (int s = 0; s < 36; ++s) {
        if (s + 2 < 36) frags(s + 2, (s + 2) % 3);
        __builtin_amdgcn_sched_barrier(0);
        acc = Mma32<T>::run(af[s % 3], bf[s % 3], acc);
        __builtin_amdgcn_sched_barrier(0);
      }
      // The next tile's mask was requested before the k-loop: it has landed by now, and so has the previous tile's dH store.  Waiting
      // HERE - in front of this tile's stores - means no wait of this wave ever has a fresh store in front of it (a vmcnt wait behind
      // a store is the store's whole round trip: 5800 instead of ~1700 ticks per tile in the first measurement of this file).
      RBP_STAMP(i, 3);
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(nm0), "+v"(nm1)::"memory");
      RBP_STAMP(i, 4);
      // dH = (h > 0) ? acc : 0, zero outside the image; -> LDS (conv2's operand) and -> global (the weight gradient's operand)
      u32x4 pk[2];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const u32x4 hm = q < 2 ? hm0 : hm1;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const unsigned word = hm[2 * (q & 1) + (e >> 1)];
          const float hv = bits16_to_f32<T>((unsigned short)((e & 1) ? (word >> 16) : (word & 0xffffu)));
          v[e] = (inside && hv > 0.f) ? acc[4 * q + e] : 0.f;
        }
        pk[q >> 1][2 * (q & 1)] = pack2<T>(v[0], v[1]);
        pk[q >> 1][2 * (q & 1) + 1] = pack2<T>(v[2], v[3]);
      }
      if (n0 < kHPix) {
        char* hb = smem + kH0 + buf * kHBytes;
        *reinterpret_cast<u32x4*>(hb + hoff) = pk[0];
        *reinterpret_cast<u32x4*>(hb + hoff1) = pk[1];
      }
      arrive(1);
      RBP_STAMP(i, 5);
      if (n0 < kHPix && inside && hy >= 1 && hy <= TH && hx >= 1 && hx <= 8) {
        char* dst = p.out_h + pix * 128 + (32 * rt + 16 * hi) * 2;
        *reinterpret_cast<u32x4*>(dst) = pk[0];
        *reinterpret_cast<u32x4*>(dst + 16) = pk[1];
      }
      RBP_STAMP(i, 6);
    }
    RBP_FLUSH(0);
    return;
  }

  // ================================================================================================= CONV2 WAVES: (rt, kc)
  const int rt = wid & 1, kc = (wid >> 1) & 1, pw = wid - 4;
  // patch DMA: block j = pw + 4 k covers image rows 16 j .. = pixels 8 j .. 8 j + 7, both chunks; per block and lane, the same for every tile:
  // patch row / column, validity, byte offset inside the pixel
  int dpy[3], dpx[3], dof[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int j = pw + 4 * k;
    const int lrow = lane >> 2, cc = lrow >> 3;
    const int row = 8 * j + (lrow & 7);
    int py, px;
    bool valid;
    if (row < kMain) {
      py = (row * 205) >> 11;
      px = row - 10 * py;
      valid = true;
    } else {
      const int e4 = (row - kMain) & 15;
      py = (5 * (e4 >> 1) + 7) & 7;
      px = 10 + (e4 & 1);
      valid = true;
    }
    dpy[k] = valid ? py - 2 : -(1 << 20);
    dpx[k] = px - 2;
    dof[k] = cc * 64 + (((lane & 3) ^ ((row >> 2) & 3)) << 4);
  }
  auto dma_patch = [&](const Tile& tl, int buf) {
    const char* in_n = p.in + (size_t)tl.n * p.H * p.W * 128;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int iy = tl.y0 + dpy[k], ix = tl.x0 + dpx[k];
      const bool ok = ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
      const char* src = in_n + (unsigned)((iy * p.W + ix) * 128 + dof[k]);
      glds16(ok ? src : p.zero, lds0 + kP0 + buf * kPBytes + (pw + 4 * k) * 1024);
    }
  };
  dma_patch(tile_of(0), 0);
  if (ntl > 1) dma_patch(tile_of(1), 1);
  // stage 2's A-fragments, resident: per tap the chunk's two halves; lane (l32, hi): matrix row l32 of row tile rt (the packed row that
  // makes its 16 accumulator registers 16 consecutive channels), bytes 16 (2 half + hi); spatial offset t goes with weight slot 8 - t
  bf16x8 wx[9], wy[9];
  {
    const int j = l32 >> 3, hm = (l32 >> 2) & 1, e = l32 & 3;
    const int R = 16 * (2 * rt + (j & 1)) + 4 * (2 * hm + (j >> 1)) + e;
    const char* w2l = p.w2 + kc * 4096 + R * 64 + hi * 16;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      wx[t] = *reinterpret_cast<const bf16x8*>(w2l + (8 - t) * 8192);
      wy[t] = *reinterpret_cast<const bf16x8*>(w2l + (8 - t) * 8192 + 32);
    }
  }
  int xb[9];
  {
    const int oy = l32 >> 3, ox = l32 & 7;
#pragma unroll
    for (int tt = 0; tt < 9; ++tt) xb[tt] = kc * kHChunk + img_off((oy + tt / 3) * kHP + ox + tt % 3, hi);
  }
  const int oy = l32 >> 3, ox = l32 & 7;
  const int skip_off = patch_off(10 * (oy + 2) + ox + 2, rt, 2 * hi + kc);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  lds_barrier();   // W1, patches 0 and 1 and W2 are in
  if (ntl > 1) arrive(0);   // (patch 1)
  for (int i = 0; i < ntl; ++i) {
    const Tile tl = tile_of(i);
    const int buf = i & 1, pbuf = i % kNPB;
    RBP_STAMP(i, 0);
    if (i > 0) await(2, 4u * i);   // every conv2 wave is through with tile i - 1: its patch buffer and the exchange buffer are free
    if (i + 2 < ntl) dma_patch(tile_of(i + 2), (i + 2) % kNPB);   // (the buffer tile i - 1 has just left)
    RBP_STAMP(i, 1);
    await(1, 4u * (i + 1));   // h of tile i is complete
    RBP_STAMP(i, 2);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    bf16x8 xf[3];
    const char* const hb = smem + kH0 + buf * kHBytes;
    auto frags2 = [&](int s2, int b) { xf[b] = *reinterpret_cast<const bf16x8*>(hb + (xb[s2 >> 1] ^ ((s2 & 1) * 32))); };
    frags2(0, 0);
    frags2(1, 1);
#pragma unroll
    for (int s2 = 0; s2 < 18; ++s2) {
      if (s2 + 2 < 18) frags2(s2 + 2, (s2 + 2) % 3);
      __builtin_amdgcn_sched_barrier(0);
      acc = Mma32<T>::run((s2 & 1) ? wy[s2 >> 1] : wx[s2 >> 1], xf[s2 % 3], acc);
      __builtin_amdgcn_sched_barrier(0);
    }
    RBP_STAMP(i, 3);
    // the K halves meet: wave kc keeps registers 8 kc .. 8 kc + 7 and hands the other eight to its partner
    char* const xw = smem + kX0 + ((rt * 2 + kc) * 2) * 1024 + lane * 16;
    const char* const xr = smem + kX0 + ((rt * 2 + (kc ^ 1)) * 2) * 1024 + lane * 16;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = kc ? acc[4 * q + e] : acc[8 + 4 * q + e];
      *reinterpret_cast<f32x4*>(xw + q * 1024) = v;
    }
    arrive(3 + rt);
    await(3 + rt, 2u * (i + 1));
    RBP_STAMP(i, 4);
    const f32x4 o0 = *reinterpret_cast<const f32x4*>(xr);
    const f32x4 o1 = *reinterpret_cast<const f32x4*>(xr + 1024);
    const u32x4 rr = *reinterpret_cast<const u32x4*>(smem + kP0 + pbuf * kPBytes + skip_off);   // skip = dOut, from the patch
    // patch i + 2, issued at the top of this tile, has landed (and the previous tile's store is long done): say so BEFORE this tile's
    // store goes out, so that this wait never has a fresh store in front of it
    if (i + 2 < ntl) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      arrive(0);
    }
    RBP_STAMP(i, 5);
    const int y = tl.y0 + oy, x = tl.x0 + ox;
    if (y < p.H && x < p.W) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] = (kc ? acc[8 + e] : acc[e]) + o0[e];
        v[4 + e] = (kc ? acc[12 + e] : acc[4 + e]) + o1[e];
      }
      u32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float s0 = bits16_to_f32<T>((unsigned short)(rr[e] & 0xffffu)), s1 = bits16_to_f32<T>((unsigned short)(rr[e] >> 16));
        o[e] = pack2<T>(v[2 * e] + s0, v[2 * e + 1] + s1);
      }
      *reinterpret_cast<u32x4*>(p.out_a + (((size_t)tl.n * p.H + y) * p.W + x) * 128 + (32 * rt + 16 * hi + 8 * kc) * 2) = o;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the skip / exchange reads are done before the buffers are handed on)
    arrive(2);
    RBP_STAMP(i, 6);
  }
  RBP_FLUSH(1);
}

template <typename T>
int launch_rbp(const RbpK& k, unsigned blocks, hipStream_t st) {
  auto fn = resblock_pp_bwd_kernel<T>;
  static std::atomic<bool> attr_done{false};
  if (!attr_done) {
    TG_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, kLds));
    attr_done = true;
  }
  hipLaunchKernelGGL(fn, dim3(blocks), dim3(512), kLds, st, k);
  return tg_launch_status();
}

}  // namespace

extern "C" int tg_resblock_bwd_pp(int dtype, const void* dout, const void* w2_dgrad_packed, const void* h, const void* w1_dgrad_packed,
                                  void* out_dh, void* out_din, int N, int H, int W, int C, int max_workgroups, void* stream) {
  if (!dout || !w2_dgrad_packed || !h || !w1_dgrad_packed || !out_dh || !out_din || N <= 0 || H <= 0 || W <= 0) return TG_E_BADARG;
  if ((dtype != TG_BF16 && dtype != TG_F16) || C != 64) return TG_E_UNSUPPORTED;
  const void* ptrs[] = {dout, w2_dgrad_packed, h, w1_dgrad_packed, out_dh, out_din};
  for (const void* q : ptrs)
    if (!tg_aligned16(q)) return TG_E_ALIGN;
  if ((long long)H * W * 128 > 0x7fffffffLL) return TG_E_UNSUPPORTED;
  static const char* zero_page = [] {
    void* z = nullptr;
    return hipGetSymbolAddress(&z, HIP_SYMBOL(tg_rbp_zero_page)) == hipSuccess ? (const char*)z : (const char*)nullptr;
  }();
  if (!zero_page) return TG_E_BADARG;
  RbpK k;
  k.in = (const char*)dout; k.w1 = (const char*)w2_dgrad_packed; k.w2 = (const char*)w1_dgrad_packed; k.mask = (const char*)h;
  k.out_h = (char*)out_dh; k.out_a = (char*)out_din; k.zero = zero_page;
  k.N = N; k.H = H; k.W = W;
  k.tiles_x = (W + 7) / 8;
  k.tiles_y = (H + TH - 1) / TH;
  const long long nt = (long long)k.tiles_x * k.tiles_y * N;
  if (nt > 0x7fffffffLL) return TG_E_UNSUPPORTED;
  k.ntiles = (int)nt;
  const int cap = max_workgroups > 0 ? max_workgroups : 256;
  const unsigned blocks = (unsigned)(nt < cap ? nt : cap);
  hipStream_t st = (hipStream_t)stream;
  return dtype == TG_F16 ? launch_rbp<F16>(k, blocks, st) : launch_rbp<BF16>(k, blocks, st);
}
