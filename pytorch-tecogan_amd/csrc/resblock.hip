// One residual block of the generator trunk in ONE launch (bf16, 64 channels):
//
//     h   = relu(conv3x3(a, W1) + b1)          code/ops.py:45-54 (residual_block), code/models.py:66-69
//     out = a + conv3x3(h, W2)
//
// The recurrent pass runs 16 such blocks per frame on 4 x 32x32 pixels: each conv is a ~6 us launch of which the MFMAs
// are a few hundred nanoseconds - the rest is the launch boundary, the weight/activation load latency and the store.
// Fusing the pair halves the boundaries and loads `a` once.  A workgroup owns an 8x8 output tile of one image:
//   phase 1  stage the 12x12 input patch and W1 (73.7 KB, all taps) in LDS; start W2's global loads into registers
//   phase 2  conv1 on the 10x10 halo region (7 MFMA pixel tiles), bias + relu, zero outside the image (conv2 pads h with
//            zeros), h -> LDS (bf16, exactly what the unfused path would read back) and -> global (the backward pass
//            needs it: relu mask and weight-gradient operand)
//   phase 3  W2 registers -> the LDS region W1 occupied; conv2 on the 8x8 tile from the LDS copy of h; + a; store
// MFMA operand roles and the packed-weight layout are those of conv_mfma.hip; the LDS images are swizzled (below).
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
// fragment read of both convolutions (all taps, all pixel tiles) and of the weights is conflict-free; the 80-byte padded
// rows of conv_mfma.hip cost 2x on the weight reads and 2.6-3x on these pixel patterns (exhaustive count, tools/lds_layout.py).
constexpr int kRow = 64;
constexpr int kInW = 12, kInPix = 144;      // input patch 12 x 12 pixels ...
constexpr int kInP = 18, kInRows = 12 * kInP;  // ... stored with a pitch of 18 rows
constexpr int kHW = 10, kHPix = 100;        // h region 10 x 10 ...
constexpr int kHP = 16, kHRows = 10 * kHP;  // ... stored with a pitch of 16 rows
constexpr int kLdsIn = 2 * kInRows * kRow;  // [chunk][row][64]
constexpr int kLdsH = 2 * kHRows * kRow;
constexpr int kLdsW = 2 * 9 * 64 * kRow;    // [chunk][tap][row][64]
constexpr int kLdsTotal = kLdsIn + kLdsH + kLdsW;

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

__global__ __launch_bounds__(256) void resblock_fwd_kernel(const ResblockK p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* lds_in = smem;
  char* lds_h = smem + kLdsIn;
  char* lds_w = smem + kLdsIn + kLdsH;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform on purpose: keeps wc / wp in SGPRs
  const int idx = lane & 15, g = lane >> 4;
  const int wc = wid & 1, wp = wid >> 1;  // wave = 32-channel half x pixel-tile half
  int bx = blockIdx.x;
  const int txb = bx % p.tiles_x;
  bx /= p.tiles_x;
  const int tyb = bx % p.tiles_y;
  const int n = bx / p.tiles_y;
  const int y0 = tyb * 8, x0 = txb * 8;
  const char* in_n = p.in + (size_t)n * p.H * p.W * 128;

  RB_STAMP(0);
  // ---- phase 1: every global load of the phase is issued before the first LDS store
  // (the patch loads are unconditional from a clamped address and zeroed afterwards: a load under a divergent `if` makes
  // the compiler wait for each one before issuing the next - five dependent round trips)
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
  // packed weights: [tap][chunk][64 rows][64 B]; one (tap, chunk) block is exactly 256 16-byte pieces
  u32x4 vw[18];
#pragma unroll
  for (int u = 0; u < 18; ++u) vw[u] = *reinterpret_cast<const u32x4*>(p.w1 + ((size_t)u * 256 + tid) * 16);
  // this lane's bias (channels 32*wc + 8g .. +7), in flight under the staging
  float bias[8];
  {
    const f32x4 t0 = *reinterpret_cast<const f32x4*>(p.b1 + 32 * wc + 8 * g);
    const f32x4 t1 = *reinterpret_cast<const f32x4*>(p.b1 + 32 * wc + 8 * g + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) { bias[j] = t0[j]; bias[4 + j] = t1[j]; }
  }
  auto store_w = [&]() {
#pragma unroll
    for (int u = 0; u < 18; ++u) {
      const int tt = u >> 1, cc = u & 1;
      *reinterpret_cast<u32x4*>(lds_w + (cc * 9 + tt) * 64 * kRow + lds_off(tid >> 2, tid & 3)) = vw[u];
    }
  };
  RB_STAMP(1);
#pragma unroll
  for (int u = 0; u < 5; ++u)
    if (da[u] >= 0) *reinterpret_cast<u32x4*>(lds_in + da[u]) = ok[u] ? va[u] : u32x4{0u, 0u, 0u, 0u};
  store_w();
  RB_STAMP(2);
  __syncthreads();
  RB_STAMP(3);
  // W2 travels in registers while conv1 computes
#pragma unroll
  for (int u = 0; u < 18; ++u) vw[u] = *reinterpret_cast<const u32x4*>(p.w2 + ((size_t)u * 256 + tid) * 16);

  // ---- phase 2: conv1 over the 10x10 region.  Pixel tile t covers region pixels 16t .. 16t+15 (row-major, 10 wide)
  // this wave's two weight tiles: packed rows 32wc .. 32wc+31 (row bit 2 == idx bit 2 in both tiles)
  const int wrow = (wc * 2) * 16 * kRow + lds_off(idx, g);
  {
    // tiles 0-3 / 4-7; tile 7 does not exist (and tile 6 is partial): those lanes read a clamped pixel and their results are
    // dropped - cheaper than a branch around two of the eight MFMAs of every k-step
    int xa[4][9], hp_l[4];  // LDS byte offset of every (pixel tile, tap) fragment of this lane: no address math in the k-loop
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      int hp = (wp * 4 + b) * 16 + idx;
      hp_l[b] = hp;
      hp = hp < kHPix ? hp : kHPix - 1;
      const int hy = (hp * 205) >> 11, hx = hp - hy * kHW;  // hp / 10 for hp < 128
#pragma unroll
      for (int tt = 0; tt < 9; ++tt) xa[b][tt] = lds_off((hy + tt / 3) * kInP + hx + tt % 3, g);
    }
    f32x4 acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    // 18 k-steps (chunk, tap); the fragments of step s+1 are read from LDS before the MFMAs of step s are issued - with
    // one wave per SIMD nothing else hides the LDS latency
    bf16x8 wf[3][2], xf[3][4];
    auto frags1 = [&](int st, int buf) {
      const int cc = st / 9, tt = st - cc * 9;
      const char* lw = lds_w + (cc * 9 + tt) * 64 * kRow + wrow;
      wf[buf][0] = *reinterpret_cast<const bf16x8*>(lw);
      wf[buf][1] = *reinterpret_cast<const bf16x8*>(lw + 16 * kRow);
#pragma unroll
      for (int b = 0; b < 4; ++b) xf[buf][b] = *reinterpret_cast<const bf16x8*>(lds_in + cc * kInRows * kRow + xa[b][tt]);
    };
    // (sched_barrier: without it the scheduler sinks every read back next to its MFMA to save registers)
    frags1(0, 0);
    frags1(1, 1);
#pragma unroll
    for (int st = 0; st < 18; ++st) {
      const int cur = st % 3;
      if (st + 2 < 18) frags1(st + 2, (st + 2) % 3);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        acc[0][b] = mma(wf[cur][0], xf[cur][b], acc[0][b]);
        acc[1][b] = mma(wf[cur][1], xf[cur][b], acc[1][b]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    RB_STAMP(4);
    // lane (idx, g): region pixel hp_l[b], channels 32wc + 8g .. +7 = rows 4g..4g+3 of the wave's two tiles
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int hp = hp_l[b];
      if (hp < kHPix) {
        const int hy = (hp * 205) >> 11, hx = hp - hy * kHW;
        const int y = y0 - 1 + hy, x = x0 - 1 + hx;
        const bool inside = y >= 0 && y < p.H && x >= 0 && x < p.W;
        float v[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v[j] = acc[0][b][j] + bias[j];
          v[4 + j] = acc[1][b][j] + bias[4 + j];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (inside && v[e] > 0.f) ? v[e] : 0.f;
        const u32x4 pk = pack8(v);
        *reinterpret_cast<u32x4*>(lds_h + wc * kHRows * kRow + lds_off(hy * kHP + hx, g)) = pk;
        if (inside && hy >= 1 && hy <= 8 && hx >= 1 && hx <= 8)
          *reinterpret_cast<u32x4*>(p.out_h + (((size_t)n * p.H + y) * p.W + x) * 128 + wc * 64 + g * 16) = pk;
      }
    }
  }
  RB_STAMP(5);
  __syncthreads();  // W1 reads done, h complete
  RB_STAMP(6);
  store_w();
  __syncthreads();
  RB_STAMP(7);

  // ---- phase 3: conv2 on the 8x8 tile; pixel tile t = output rows 2t, 2t+1
  {
    int xa[2][9];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int op = (wp * 2 + b) * 16 + idx;
#pragma unroll
      for (int tt = 0; tt < 9; ++tt) xa[b][tt] = lds_off(((op >> 3) + tt / 3) * kHP + (op & 7) + tt % 3, g);
    }
    f32x4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 wf[3][2], xf[3][2];
    auto frags2 = [&](int st, int buf) {
      const int cc = st / 9, tt = st - cc * 9;
      const char* lw = lds_w + (cc * 9 + tt) * 64 * kRow + wrow;
      wf[buf][0] = *reinterpret_cast<const bf16x8*>(lw);
      wf[buf][1] = *reinterpret_cast<const bf16x8*>(lw + 16 * kRow);
#pragma unroll
      for (int b = 0; b < 2; ++b) xf[buf][b] = *reinterpret_cast<const bf16x8*>(lds_h + cc * kHRows * kRow + xa[b][tt]);
    };
    frags2(0, 0);
    frags2(1, 1);
#pragma unroll
    for (int st = 0; st < 18; ++st) {
      const int cur = st % 3;
      if (st + 2 < 18) frags2(st + 2, (st + 2) % 3);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        acc[0][b] = mma(wf[cur][0], xf[cur][b], acc[0][b]);
        acc[1][b] = mma(wf[cur][1], xf[cur][b], acc[1][b]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    RB_STAMP(8);
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int op = (wp * 2 + b) * 16 + idx;
      const int oy = op >> 3, ox = op & 7;
      const int y = y0 + oy, x = x0 + ox;
      if (y < p.H && x < p.W) {
        float r[8], v[8];
        Vec<BF16>::load(lds_in + wc * kInRows * kRow + lds_off((oy + 2) * kInP + ox + 2, g), r);  // the skip connection
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v[j] = acc[0][b][j] + r[j];
          v[4 + j] = acc[1][b][j] + r[4 + j];
        }
        *reinterpret_cast<u32x4*>(p.out_a + (((size_t)n * p.H + y) * p.W + x) * 128 + wc * 64 + g * 16) = pack8(v);
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
