// One residual block of the generator trunk in ONE launch (bf16, 64 channels):
//
//     h   = relu(conv3x3(a, W1) + b1)          code/ops.py:45-54 (residual_block), code/models.py:66-69
//     out = a + conv3x3(h, W2)
//
// The recurrent pass runs 16 such blocks per frame on 4 x 32x32 pixels: as two launches each conv is ~6 us of which the
// MFMAs are a few hundred nanoseconds - the rest is the launch boundary, load latency and stores.  Fusing the pair halves
// the boundaries and loads `a` once.  A workgroup owns an 8x8 output tile of one image; wave w owns output channels of MFMA
// row tile w (16 packed rows) for EVERY pixel tile, so
//   * the weights never touch LDS: each lane loads exactly the A-fragments it will feed to the MFMAs (18 x 16 B per conv,
//     1 KiB contiguous per wave-load) straight from the packed global image, in k-step order, a few steps ahead of their
//     use, and the k-loops wait for them one step at a time (vmcnt) - the weight stream runs underneath the MFMAs;
//   * LDS holds only the 12x12 input patch and the 10x10 h region (48 KB), in conflict-free swizzled rows (below).
//   phase 1  patch loads and the head of the weight stream issued; patch -> LDS
//   phase 2  conv1 on the 10x10 halo region (7 MFMA pixel tiles), bias + relu, zero outside the image (conv2 pads h with
//            zeros), h -> LDS (bf16, exactly what the unfused path would read back) and -> global (the backward pass
//            needs it: relu mask and weight-gradient operand)
//   phase 3  conv2 on the 8x8 tile from the LDS copy of h; + a (from the LDS patch); store
// MFMA operand roles and the packed-weight layout are those of conv_mfma.hip.
#include "common.h"

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
#else
#define RB_STAMP(i) do {} while (0)
#endif

namespace {

// LDS images are rows of 64 bytes (32 bf16 channels of one pixel / one packed weight row), UNPADDED, with the 16-byte piece
// index XOR-swizzled by bit 2 of the row: piece' = piece ^ 2*((row >> 2) & 1).  ds_read_b128 services a wave in four fixed
// groups of 16 lanes ({0-3,12-15,20-27}, ...; MI355X_MICROARCH.md, LDS), each group needs 16 distinct 16-byte slots mod
// 256 B.  With this swizzle, a patch pitch of 18 rows for the 12-wide input patch and 16 for the 10-wide h region, every
// fragment read of both convolutions (all taps, all pixel tiles) is conflict-free; the 80-byte padded rows of conv_mfma.hip
// cost 2.6-3x on these pixel patterns (exhaustive count, tools/lds_layout.py).
constexpr int kRow = 64;
constexpr int kInW = 12, kInPix = 144;      // input patch 12 x 12 pixels ...
constexpr int kInP = 18, kInRows = 12 * kInP;  // ... stored with a pitch of 18 rows
constexpr int kHW = 10, kHPix = 100;        // h region 10 x 10 ...
constexpr int kHP = 16, kHRows = 10 * kHP;  // ... stored with a pitch of 16 rows
constexpr int kLdsIn = 2 * kInRows * kRow;  // [chunk][row][64]
constexpr int kLdsH = 2 * kHRows * kRow;
constexpr int kLdsTotal = kLdsIn + kLdsH;

// byte offset of 16-byte piece `piece` of row `row` inside an image
__device__ __forceinline__ int lds_off(int row, int piece) { return row * kRow + ((piece ^ ((row >> 1) & 2)) << 4); }

struct ResblockK {
  const char* in;
  const char* w1;
  const float* b1;
  const char* w2;
  char* out_h;
  char* out_a;
  int N, H, W, tiles_x, tiles_y;
};

__device__ __forceinline__ f32x4 mma(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ u32x4 pack8(const float* v) {
  u32x4 t;
#pragma unroll
  for (int i = 0; i < 4; ++i)
    t[i] = (unsigned)f32_to_bf16_bits(v[2 * i]) | ((unsigned)f32_to_bf16_bits(v[2 * i + 1]) << 16);
  return t;
}

// LDS-only barrier: __syncthreads() would also drain vmcnt, i.e. wait for the weight stream and the h stores
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__global__ __launch_bounds__(256) void resblock_fwd_kernel(const ResblockK p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* lds_in = smem;
  char* lds_h = smem + kLdsIn;

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave = MFMA row tile (wave-uniform on purpose)
  const int idx = lane & 15, g = lane >> 4;
  int bx = blockIdx.x;
  const int txb = bx % p.tiles_x;
  bx /= p.tiles_x;
  const int tyb = bx % p.tiles_y;
  const int n = bx / p.tiles_y;
  const int y0 = tyb * 8, x0 = txb * 8;
  const char* in_n = p.in + (size_t)n * p.H * p.W * 128;

  RB_STAMP(0);
  // ---- phase 1.  Patch loads are unconditional from a clamped address and zeroed afterwards: a load under a divergent
  // `if` makes the compiler wait for each one before issuing the next.
  u32x4 va[5];
  int da[5];
  bool ok[5];
#pragma unroll
  for (int u = 0; u < 5; ++u) {
    const int i = min(tid + u * 256, 2 * kInPix * 4 - 1);
    const int s = i & 3, r = i >> 2;
    const int cc = r >= kInPix ? 1 : 0, prow = r - cc * kInPix;
    const int py = (prow * 171) >> 11, px = prow - py * kInW;  // prow / 12, exact for prow < 144
    const int iy = y0 - 2 + py, ix = x0 - 2 + px;
    da[u] = (tid + u * 256 < 2 * kInPix * 4) ? cc * kInRows * kRow + lds_off(py * kInP + px, s) : -1;
    ok[u] = iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
    const int cy = min(max(iy, 0), p.H - 1), cx = min(max(ix, 0), p.W - 1);
    va[u] = *reinterpret_cast<const u32x4*>(in_n + ((size_t)cy * p.W + cx) * 128 + cc * 64 + s * 16);
  }
  // A-fragments of this wave's row tile, in k-step order st = chunk*9 + tap.  Packed image: [tap][chunk][64 rows][64 B];
  // lane (idx, g) needs bytes 16g..16g+15 of row 16w + idx.  The 36 fragment loads (W1 then W2) form one stream that is
  // issued kAhead steps ahead of its use: a CU accepts ~33 B/clk of vector loads (measured: the 41 loads of this kernel
  // issued back to back block the wave for 5000 cycles), so issuing everything up front serialises load and compute,
  // while one load per k-step lets every issue stall overlap the previous step's MFMAs.
  const int wlane = ((w * 16 + idx) * 64 + g * 16);
  bf16x8 wfr[36];
  auto issue_w = [&](int k) {  // k < 18: W1 step k; else W2 step k-18 (compile-time after unrolling)
    const int st = k < 18 ? k : k - 18;
    const char* base = k < 18 ? p.w1 : p.w2;
    wfr[k] = *reinterpret_cast<const bf16x8*>(base + (size_t)((st % 9) * 2 + st / 9) * 4096 + wlane);
  };
  constexpr int kAhead = 8;
#pragma unroll
  for (int k = 0; k < kAhead; ++k) issue_w(k);
  // lane (idx, g) of row tile w ends up with channels ch0 .. ch0+3 of pixel idx (row_to_channel<BF16> of common.h)
  const int chunk = w >> 1, half = w & 1;
  const int ch0 = 32 * chunk + 8 * g + 4 * half;
  const f32x4 bias = *reinterpret_cast<const f32x4*>(p.b1 + ch0);
  RB_STAMP(1);

  // fragment addresses of conv1 (7 pixel tiles x 9 taps), computed while the loads are in flight.  Pixel tile t covers
  // region pixels 16t .. 16t+15 (row-major, 10 wide); tile 6 is partial: its spare lanes read a clamped pixel.
  int xa[7][9];
#pragma unroll
  for (int t = 0; t < 7; ++t) {
    int hp = t * 16 + idx;
    hp = hp < kHPix ? hp : kHPix - 1;
    const int hy = (hp * 205) >> 11, hx = hp - hy * kHW;  // hp / 10
#pragma unroll
    for (int tt = 0; tt < 9; ++tt) xa[t][tt] = lds_off((hy + tt / 3) * kInP + hx + tt % 3, g);
  }
#pragma unroll
  for (int u = 0; u < 5; ++u)
    if (da[u] >= 0) *reinterpret_cast<u32x4*>(lds_in + da[u]) = ok[u] ? va[u] : u32x4{0u, 0u, 0u, 0u};
  RB_STAMP(2);
  lds_barrier();
  RB_STAMP(3);

  // ---- phase 2: conv1.  Fragments of k-step st+2 are read from LDS before the MFMAs of step st are issued (one wave per
  // SIMD: nothing else hides the LDS latency; sched_barrier keeps the scheduler from sinking the reads back).
  {
    f32x4 acc[7];
#pragma unroll
    for (int t = 0; t < 7; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 xf[3][7];
    auto frags = [&](int st, int buf) {
      const int cc = st / 9, tt = st - cc * 9;
#pragma unroll
      for (int t = 0; t < 7; ++t) xf[buf][t] = *reinterpret_cast<const bf16x8*>(lds_in + cc * kInRows * kRow + xa[t][tt]);
    };
    frags(0, 0);
    frags(1, 1);
#pragma unroll
    for (int st = 0; st < 18; ++st) {
      if (st + 2 < 18) frags(st + 2, (st + 2) % 3);
      issue_w(st + kAhead);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < 7; ++t) acc[t] = mma(wfr[st], xf[st % 3][t], acc[t]);
      __builtin_amdgcn_sched_barrier(0);
    }
    RB_STAMP(4);
#pragma unroll
    for (int t = 0; t < 7; ++t) {
      const int hp = t * 16 + idx;
      if (hp < kHPix) {
        const int hy = (hp * 205) >> 11, hx = hp - hy * kHW;
        const int y = y0 - 1 + hy, x = x0 - 1 + hx;
        const bool inside = y >= 0 && y < p.H && x >= 0 && x < p.W;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v[j] = acc[t][j] + bias[j];
          v[j] = (inside && v[j] > 0.f) ? v[j] : 0.f;
        }
        uint2 pk;
        pk.x = (unsigned)f32_to_bf16_bits(v[0]) | ((unsigned)f32_to_bf16_bits(v[1]) << 16);
        pk.y = (unsigned)f32_to_bf16_bits(v[2]) | ((unsigned)f32_to_bf16_bits(v[3]) << 16);
        *reinterpret_cast<uint2*>(lds_h + chunk * kHRows * kRow + lds_off(hy * kHP + hx, g) + half * 8) = pk;
        if (inside && hy >= 1 && hy <= 8 && hx >= 1 && hx <= 8)
          *reinterpret_cast<uint2*>(p.out_h + (((size_t)n * p.H + y) * p.W + x) * 128 + ch0 * 2) = pk;
      }
    }
  }
  RB_STAMP(5);
  lds_barrier();  // h complete
  RB_STAMP(6);

  // ---- phase 3: conv2 on the 8x8 tile; pixel tile t = output rows 2t, 2t+1
  {
    int xb[4][9];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int op = t * 16 + idx;
#pragma unroll
      for (int tt = 0; tt < 9; ++tt) xb[t][tt] = lds_off(((op >> 3) + tt / 3) * kHP + (op & 7) + tt % 3, g);
    }
    f32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 xf[3][4];
    auto frags = [&](int st, int buf) {
      const int cc = st / 9, tt = st - cc * 9;
#pragma unroll
      for (int t = 0; t < 4; ++t) xf[buf][t] = *reinterpret_cast<const bf16x8*>(lds_h + cc * kHRows * kRow + xb[t][tt]);
    };
    frags(0, 0);
    frags(1, 1);
    RB_STAMP(7);
#pragma unroll
    for (int st = 0; st < 18; ++st) {
      if (st + 2 < 18) frags(st + 2, (st + 2) % 3);
      if (18 + st + kAhead < 36) issue_w(18 + st + kAhead);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = mma(wfr[18 + st], xf[st % 3][t], acc[t]);
      __builtin_amdgcn_sched_barrier(0);
    }
    RB_STAMP(8);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int op = t * 16 + idx;
      const int oy = op >> 3, ox = op & 7;
      const int y = y0 + oy, x = x0 + ox;
      if (y < p.H && x < p.W) {
        // the skip connection comes from the LDS patch
        const uint2 rr = *reinterpret_cast<const uint2*>(lds_in + chunk * kInRows * kRow +
                                                         lds_off((oy + 2) * kInP + ox + 2, g) + half * 8);
        float v[4];
        v[0] = acc[t][0] + __uint_as_float(rr.x << 16);
        v[1] = acc[t][1] + __uint_as_float(rr.x & 0xffff0000u);
        v[2] = acc[t][2] + __uint_as_float(rr.y << 16);
        v[3] = acc[t][3] + __uint_as_float(rr.y & 0xffff0000u);
        uint2 pk;
        pk.x = (unsigned)f32_to_bf16_bits(v[0]) | ((unsigned)f32_to_bf16_bits(v[1]) << 16);
        pk.y = (unsigned)f32_to_bf16_bits(v[2]) | ((unsigned)f32_to_bf16_bits(v[3]) << 16);
        *reinterpret_cast<uint2*>(p.out_a + (((size_t)n * p.H + y) * p.W + x) * 128 + ch0 * 2) = pk;
      }
    }
  }
  RB_STAMP(9);
}

}  // namespace

extern "C" int tg_resblock_fwd(int dtype, const void* in, const void* w1_packed, const float* b1, const void* w2_packed,
                               void* out_h, void* out_a, int N, int H, int W, int C, void* stream) {
  if (!in || !w1_packed || !b1 || !w2_packed || !out_h || !out_a || N <= 0 || H <= 0 || W <= 0) return TG_E_BADARG;
  if (dtype != TG_BF16 || C != 64) return TG_E_UNSUPPORTED;  // the trunk shape; anything else runs as two tg_conv launches
  if (!tg_aligned16(in) || !tg_aligned16(w1_packed) || !tg_aligned16(w2_packed) || !tg_aligned16(out_h) ||
      !tg_aligned16(out_a) || !tg_aligned16(b1))
    return TG_E_ALIGN;
  ResblockK k;
  k.in = (const char*)in; k.w1 = (const char*)w1_packed; k.b1 = b1; k.w2 = (const char*)w2_packed;
  k.out_h = (char*)out_h; k.out_a = (char*)out_a;
  k.N = N; k.H = H; k.W = W;
  k.tiles_x = (W + 7) / 8; k.tiles_y = (H + 7) / 8;
  const long long blocks = (long long)k.tiles_x * k.tiles_y * N;
  if (blocks > 0x7fffffffLL) return TG_E_UNSUPPORTED;
  static bool attr_done = false;
  if (!attr_done) {
    TG_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(resblock_fwd_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, kLdsTotal));
    attr_done = true;
  }
  hipLaunchKernelGGL(resblock_fwd_kernel, dim3((unsigned)blocks), dim3(256), kLdsTotal, (hipStream_t)stream, k);
  return tg_launch_status();
}
