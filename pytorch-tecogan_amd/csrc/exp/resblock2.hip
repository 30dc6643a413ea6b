// TWO consecutive residual blocks of the generator trunk in ONE launch (16-bit element types, 64 channels, 8 x 4 output tiles):
//
//     h1 = relu(conv3x3(a0, W1a) + b1a)   a1 = a0 + conv3x3(h1, W2a)          code/ops.py:45-54 (residual_block) twice,
//     h2 = relu(conv3x3(a1, W1b) + b1b)   a2 = a1 + conv3x3(h2, W2b)          code/models.py:66-69
//
// The recurrent pass is a chain of ~25 dependent launches per frame on 4 x 32 x 32 pixels; a fused block (resblock.hip) costs
// 6.8 us alone and 8-9 us beside the other lane, of which ~3 us are the launch boundary and ~1.3 us the patch + first-weights round
// trip in front of the first MFMA - the MFMAs themselves are a fraction.  This kernel pays boundary and round trip once per TWO
// blocks and recomputes the halo instead of exchanging it: a workgroup owns an 8 x 4 tile of a2 and computes h2 on 10 x 6, a1 on
// 12 x 8 and h1 on 14 x 10 pixels from a 16 x 12 patch of a0 (21 MFMA pixel tiles instead of 2 x 6 - the matrix pipes are ~6 %
// busy in this chain, the extra work is cheaper than a boundary).  No cross-workgroup traffic: neighbours recompute what they
// need; every workgroup stores the part of h1 / a1 / h2 that lies in its own tile (the backward pass needs all three).
// Structure per stage = resblock.hip's: eight waves (4 row tiles x 2 K halves), weights streamed from the packed global images
// straight into registers (an 18-fragment ring: W1b / W2b arrive while conv2a / conv1b run), LDS images of 64-byte rows with the
// XOR swizzle, pitches 22 / 20 / 18 / 16 for the 16 / 14 / 12 / 10 pixel wide regions (conflict counts: tools/lds_layout.py).
// Results are bit-identical to two tg_resblock_fwd launches (same products, same summation order).
#include "common.h"
#include <atomic>
#include <type_traits>

namespace {

constexpr int kRow = 64;
constexpr int kTH = 4;                                   // output tile 8 x 4
// regions (width, height, LDS pitch in rows) and their offset from the tile origin
constexpr int kW0 = 16, kH0 = 12, kP0 = 22;              // a0 patch, origin -4
constexpr int kW1 = 14, kH1 = 10, kP1 = 20;              // h1, origin -3
constexpr int kW2 = 12, kH2 = 8, kP2 = 18;               // a1, origin -2
constexpr int kW3 = 10, kH3 = 6, kP3 = 16;               // h2, origin -1
constexpr int kChunk0 = kH0 * kP0 * kRow, kChunk1 = kH1 * kP1 * kRow, kChunk2 = kH2 * kP2 * kRow, kChunk3 = kH3 * kP3 * kRow;
constexpr int kOff0 = 0, kOff1 = kOff0 + 2 * kChunk0, kOff2 = kOff1 + 2 * kChunk1, kOff3 = kOff2 + 2 * kChunk2;
constexpr int kOffX = kOff3 + 2 * kChunk3;
constexpr int kLdsX = 8 * 4 * 1024;                      // exchange: [wave][slot][lane][16 B]
constexpr int kLdsTotal = kOffX + kLdsX;
static_assert(kLdsTotal <= 160 * 1024, "LDS budget");

__device__ __forceinline__ int lds_off(int row, int piece) { return row * kRow + ((piece ^ ((row >> 1) & 2)) << 4); }

struct Rb2K {
  const char* in;
  const char* w[4];      // W1a, W2a, W1b, W2b (forward packings)
  const float* b1a;
  const float* b1b;
  char* out_h1;
  char* out_a1;
  char* out_h2;
  char* out_a2;
  const char* pf[4];     // packed weights of the NEXT launch (or null): pulled into this XCD's L2
  int N, H, W, tiles_x, tiles_y;
};

template <typename T> __device__ __forceinline__ f32x4 mma(bf16x8 a, bf16x8 b, f32x4 c) { return Mma16<T>::run(a, b, c); }
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int N> struct Packed16 {
  unsigned v[(N + 1) / 2];
  __device__ __forceinline__ void set(int i, int x) {  // i ascending from 0
    if (i & 1) v[i >> 1] |= (unsigned)x << 16; else v[i >> 1] = (unsigned)x;
  }
  __device__ __forceinline__ int get(int i) const { return (i & 1) ? (int)(v[i >> 1] >> 16) : (int)(v[i >> 1] & 0xffffu); }
};

template <typename T> __device__ __forceinline__ uint2 pack4(const float* v) {
  uint2 pk;
  pk.x = (unsigned)f32_to_bits16<T>(v[0]) | ((unsigned)f32_to_bits16<T>(v[1]) << 16);
  pk.y = (unsigned)f32_to_bits16<T>(v[2]) | ((unsigned)f32_to_bits16<T>(v[3]) << 16);
  return pk;
}

// geometry of stage S (0: conv1a -> h1, 1: conv2a -> a1, 2: conv1b -> h2, 3: conv2b -> a2)
template <int S> struct Stage {
  static constexpr int WOUT = S == 0 ? kW1 : S == 1 ? kW2 : S == 2 ? kW3 : 8;
  static constexpr int HOUT = S == 0 ? kH1 : S == 1 ? kH2 : S == 2 ? kH3 : kTH;
  static constexpr int NPIX = WOUT * HOUT, NT = (NPIX + 15) / 16;
  static constexpr int PIN = S == 0 ? kP0 : S == 1 ? kP1 : S == 2 ? kP2 : kP3;        // pitch of the image it reads
  static constexpr int IN_OFF = S == 0 ? kOff0 : S == 1 ? kOff1 : S == 2 ? kOff2 : kOff3;
  static constexpr int IN_CHUNK = S == 0 ? kChunk0 : S == 1 ? kChunk1 : S == 2 ? kChunk2 : kChunk3;
  static constexpr int POUT = S == 0 ? kP1 : S == 1 ? kP2 : kP3;                       // pitch of the image it writes (S < 3)
  static constexpr int OUT_OFF = S == 0 ? kOff1 : S == 1 ? kOff2 : kOff3;
  static constexpr int OUT_CHUNK = S == 0 ? kChunk1 : S == 1 ? kChunk2 : kChunk3;
  static constexpr int ORG = 3 - S;                                                    // region origin = tile origin - ORG
};

template <typename T>
__global__ __launch_bounds__(512) void resblock2_kernel(const Rb2K p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int w = wid & 3, kc = wid >> 2;                      // MFMA row tile, channel chunk (K half)
  const int idx = lane & 15, g = lane >> 4;
  int bx = blockIdx.x;
  const int txb = bx % p.tiles_x;
  bx /= p.tiles_x;
  const int tyb = bx % p.tiles_y;
  const int n = bx / p.tiles_y;
  const int y0 = tyb * kTH, x0 = txb * 8;
  const char* in_n = p.in + (size_t)n * p.H * p.W * 128;

  // ---- a0 patch: 16 x 12 pixels x 2 chunks x 4 pieces = 1536 pieces, 3 per thread; unconditional loads from clamped addresses
  constexpr int NU = 3;
  u32x4 va[NU];
  int da[NU];
  bool ok[NU];
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int i = tid + u * 512;
    const int s = i & 3, r = i >> 2;
    const int cc = r >= kW0 * kH0 ? 1 : 0, prow = r - cc * kW0 * kH0;
    const int py = prow >> 4, px = prow & 15;
    const int iy = y0 - 4 + py, ix = x0 - 4 + px;
    da[u] = kOff0 + cc * kChunk0 + lds_off(py * kP0 + px, s);
    ok[u] = iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
    const int cy = min(max(iy, 0), p.H - 1), cx = min(max(ix, 0), p.W - 1);
    va[u] = *reinterpret_cast<const u32x4*>(in_n + ((size_t)cy * p.W + cx) * 128 + cc * 64 + s * 16);
  }
  // ---- weight stream: fragment ws = 9 * conv + tap lives in ring slot ws % 18.  Packed image: [tap][chunk][64 rows][64 B];
  // lane (idx, g) needs bytes 16 g .. 16 g + 15 of row 16 w + idx of chunk kc.
  const int wlane = ((kc * 64 + w * 16 + idx) * 64 + g * 16);
  bf16x8 wfr[18];
  auto issue_w = [&](auto WS) {
    constexpr int ws = decltype(WS)::value;
    if constexpr (ws < 36) wfr[ws % 18] = *reinterpret_cast<const bf16x8*>(p.w[ws / 9] + (size_t)(ws % 9) * 8192 + wlane);
  };
#define RB2_ISSUE(ws) issue_w(std::integral_constant<int, (ws)>{})
  RB2_ISSUE(0); RB2_ISSUE(1); RB2_ISSUE(2); RB2_ISSUE(3); RB2_ISSUE(4); RB2_ISSUE(5);

  const int chunk = w >> 1, half = w & 1;
  const int ch0 = 32 * chunk + 8 * g + 4 * half;   // lane (idx, g) of row tile w ends with channels ch0 .. ch0 + 3 of pixel idx
  const f32x4 bias_a = *reinterpret_cast<const f32x4*>(p.b1a + ch0);
  const f32x4 bias_b = *reinterpret_cast<const f32x4*>(p.b1b + ch0);

#pragma unroll
  for (int u = 0; u < NU; ++u) *reinterpret_cast<u32x4*>(smem + da[u]) = ok[u] ? va[u] : u32x4{0u, 0u, 0u, 0u};
  lds_barrier();

  char* myx = smem + kOffX + (wid * 4 * 64 + lane) * 16;               // exchange slots of this wave
  const char* px_ = smem + kOffX + ((wid ^ 4) * 4 * 64 + lane) * 16;   // ... of the partner (same row tile, other K half)

  // One group of pixel tiles [T0, T0 + NG) of stage S: k-loop over the 9 taps of chunk kc (weights from ring slots 9 * S .. + 8),
  // K-half exchange, epilogue.  NEXT: first weight-stream index this group's k-steps issue (one per k-step, < LIMIT).
  auto group = [&](auto S_, auto T0_, auto NG_, auto NEXT_, auto LIMIT_) {
    constexpr int S = decltype(S_)::value, T0 = decltype(T0_)::value, NG = decltype(NG_)::value;
    constexpr int NEXT = decltype(NEXT_)::value, LIMIT = decltype(LIMIT_)::value;
    using St = Stage<S>;
    constexpr int F = (NG + 1) / 2;   // tiles finalised by K half 0 (K half 1: the rest)
    Packed16<NG * 9> xa;
#pragma unroll
    for (int t = 0; t < NG; ++t) {
      int hp = (T0 + t) * 16 + idx;
      hp = hp < St::NPIX ? hp : St::NPIX - 1;
      const int hy = hp / St::WOUT, hx = hp - hy * St::WOUT;
#pragma unroll
      for (int tt = 0; tt < 9; ++tt) xa.set(t * 9 + tt, lds_off((hy + tt / 3) * St::PIN + hx + tt % 3, g));
    }
    f32x4 acc[NG];
#pragma unroll
    for (int t = 0; t < NG; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 xf[2][NG];
    const char* img = smem + St::IN_OFF + kc * St::IN_CHUNK;
    auto frags = [&](int tt, int buf) {
#pragma unroll
      for (int t = 0; t < NG; ++t) xf[buf][t] = *reinterpret_cast<const bf16x8*>(img + xa.get(t * 9 + tt));
    };
    frags(0, 0);
#pragma unroll
    for (int tt = 0; tt < 9; ++tt) {
      if (tt + 1 < 9) frags(tt + 1, (tt + 1) & 1);
      if (NEXT + tt < LIMIT) {
        // (compile-time after unrolling; spelled out because a lambda template argument cannot depend on the loop variable)
        switch (NEXT + tt) {
#define RB2_CASE(k) case k: RB2_ISSUE(k); break;
          RB2_CASE(6) RB2_CASE(7) RB2_CASE(8) RB2_CASE(9) RB2_CASE(10) RB2_CASE(11) RB2_CASE(12) RB2_CASE(13) RB2_CASE(14)
          RB2_CASE(15) RB2_CASE(16) RB2_CASE(17) RB2_CASE(18) RB2_CASE(19) RB2_CASE(20) RB2_CASE(21) RB2_CASE(22) RB2_CASE(23)
          RB2_CASE(24) RB2_CASE(25) RB2_CASE(26) RB2_CASE(27) RB2_CASE(28) RB2_CASE(29) RB2_CASE(30) RB2_CASE(31) RB2_CASE(32)
          RB2_CASE(33) RB2_CASE(34) RB2_CASE(35)
#undef RB2_CASE
          default: break;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < NG; ++t) acc[t] = mma<T>(wfr[(9 * S + tt) % 18], xf[tt & 1][t], acc[t]);
      __builtin_amdgcn_sched_barrier(0);
    }
    // exchange: K half 0 finalises tiles 0 .. F-1 of the group, K half 1 the rest; each wave hands the other tiles to its partner
    auto finish = [&](auto J0_, auto NJ_) {
      constexpr int j0 = decltype(J0_)::value, nj = decltype(NJ_)::value, o0 = j0 ? 0 : F, no = NG - nj;
#pragma unroll
      for (int j = 0; j < no; ++j) *reinterpret_cast<f32x4*>(myx + j * 1024) = acc[o0 + j];
      lds_barrier();
#pragma unroll
      for (int j = 0; j < nj; ++j) {
        const int t = j0 + j;
        const f32x4 other = *reinterpret_cast<const f32x4*>(px_ + j * 1024);
        const int hp = (T0 + t) * 16 + idx;
        if (hp < St::NPIX) {
          const int hy = hp / St::WOUT, hx = hp - hy * St::WOUT;
          const int y = y0 - St::ORG + hy, x = x0 - St::ORG + hx;
          const bool inside = y >= 0 && y < p.H && x >= 0 && x < p.W;
          const bool own = inside && hy >= St::ORG && hy < St::ORG + kTH && hx >= St::ORG && hx < St::ORG + 8;
          float v[4];
          if constexpr (S == 0 || S == 2) {   // conv1: + bias, relu
            const f32x4 bias = S == 0 ? bias_a : bias_b;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(acc[t][e] + other[e] + bias[e], 0.f);
          } else {                            // conv2: + the block's input (two pixels further in, in the image conv1 read)
            using Sk = Stage<S - 1>;
            const uint2 rr = *reinterpret_cast<const uint2*>(smem + Sk::IN_OFF + chunk * Sk::IN_CHUNK +
                                                             lds_off((hy + 2) * Sk::PIN + hx + 2, g) + half * 8);
            v[0] = acc[t][0] + other[0] + bits16_to_f32<T>((unsigned short)(rr.x & 0xffffu));
            v[1] = acc[t][1] + other[1] + bits16_to_f32<T>((unsigned short)(rr.x >> 16));
            v[2] = acc[t][2] + other[2] + bits16_to_f32<T>((unsigned short)(rr.y & 0xffffu));
            v[3] = acc[t][3] + other[3] + bits16_to_f32<T>((unsigned short)(rr.y >> 16));
          }
          if constexpr (S < 3) {   // the next conv pads this tensor with zeros outside the image
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = inside ? v[e] : 0.f;
          }
          const uint2 pk = pack4<T>(v);
          if constexpr (S < 3)
            *reinterpret_cast<uint2*>(smem + St::OUT_OFF + chunk * St::OUT_CHUNK + lds_off(hy * St::POUT + hx, g) + half * 8) = pk;
          char* dst = S == 0 ? p.out_h1 : S == 1 ? p.out_a1 : S == 2 ? p.out_h2 : p.out_a2;
          if (own) *reinterpret_cast<uint2*>(dst + (((size_t)n * p.H + y) * p.W + x) * 128 + ch0 * 2) = pk;
        }
      }
    };
    if (kc == 0) finish(std::integral_constant<int, 0>{}, std::integral_constant<int, F>{});
    else finish(std::integral_constant<int, F>{}, std::integral_constant<int, NG - F>{});
  };
#define RB2_GROUP(S, T0, NG, NEXT, LIMIT)                                                                   \
  group(std::integral_constant<int, S>{}, std::integral_constant<int, T0>{}, std::integral_constant<int, NG>{}, \
        std::integral_constant<int, NEXT>{}, std::integral_constant<int, LIMIT>{})

  // conv1a: 9 pixel tiles in two groups (5 + 4: accumulators + double-buffered fragments of 9 tiles would not fit beside the
  // 72 weight registers); the k-steps of the first group issue the rest of W1a and W2a[0..5], those of the second W2a[6..8]
  RB2_GROUP(0, 0, 5, 6, 15);
  lds_barrier();                         // every partner has read its exchange slots before they are written again
  RB2_GROUP(0, 5, 4, 15, 18);
  lds_barrier();                         // h1 complete
  RB2_GROUP(1, 0, 6, 18, 27);            // conv2a; its k-steps stream W1b into the slots W1a has left
  lds_barrier();                         // a1 complete
  RB2_GROUP(2, 0, 4, 27, 36);            // conv1b; W2b into W2a's slots
  lds_barrier();                         // h2 complete
  // L2 prefetch for the next launch's four weight images (see resblock.hip): the workgroups of one XCD (blockIdx.x % 8) share
  // the work - the first eight of them touch one eighth of each image; values are never used
  unsigned pfv[4] = {0u, 0u, 0u, 0u};
  if (p.pf[0] && (blockIdx.x >> 3) < 8) {
    const int slice = (blockIdx.x >> 3) & 7;
    const int off = slice * 9216 + (wid * 64 + lane) * 16;  // 9 KiB per slice and image: 8 waves x 1 KiB + 1 KiB
#pragma unroll
    for (int i = 0; i < 4; ++i) pfv[i] = *reinterpret_cast<const volatile unsigned*>(p.pf[i] + off);
    if (wid < 4) pfv[wid & 3] ^= *reinterpret_cast<const volatile unsigned*>(p.pf[wid] + slice * 9216 + 8192 + lane * 16);
  }
  RB2_GROUP(3, 0, 2, 36, 36);            // conv2b -> a2
  asm volatile("" ::"v"(pfv[0]), "v"(pfv[1]), "v"(pfv[2]), "v"(pfv[3]));
#undef RB2_GROUP
#undef RB2_ISSUE
}

}  // namespace

extern "C" int tg_resblock2_fwd(int dtype, const void* in, const void* w1a, const float* b1a, const void* w2a, const void* w1b,
                                const float* b1b, const void* w2b, void* out_h1, void* out_a1, void* out_h2, void* out_a2, int N,
                                int H, int W, int C, const void* const* next_w4, void* stream) {
  if (!in || !w1a || !b1a || !w2a || !w1b || !b1b || !w2b || !out_h1 || !out_a1 || !out_h2 || !out_a2 || N <= 0 || H <= 0 ||
      W <= 0)
    return TG_E_BADARG;
  if ((dtype != TG_BF16 && dtype != TG_F16) || C != 64) return TG_E_UNSUPPORTED;
  const void* al[] = {in, w1a, w2a, w1b, w2b, out_h1, out_a1, out_h2, out_a2, b1a, b1b};
  for (const void* q : al)
    if (!tg_aligned16(q)) return TG_E_ALIGN;
  Rb2K k;
  k.in = (const char*)in;
  k.w[0] = (const char*)w1a; k.w[1] = (const char*)w2a; k.w[2] = (const char*)w1b; k.w[3] = (const char*)w2b;
  k.b1a = b1a; k.b1b = b1b;
  k.out_h1 = (char*)out_h1; k.out_a1 = (char*)out_a1; k.out_h2 = (char*)out_h2; k.out_a2 = (char*)out_a2;
  for (int i = 0; i < 4; ++i) k.pf[i] = next_w4 ? (const char*)next_w4[i] : nullptr;
  if (next_w4 && (!next_w4[0] || !next_w4[1] || !next_w4[2] || !next_w4[3])) k.pf[0] = nullptr;
  k.N = N; k.H = H; k.W = W;
  k.tiles_x = (W + 7) / 8; k.tiles_y = (H + kTH - 1) / kTH;
  const long long blocks = (long long)k.tiles_x * k.tiles_y * N;
  if (blocks > 0x7fffffffLL) return TG_E_UNSUPPORTED;
  static std::atomic<bool> attr_done{false};
  if (!attr_done) {
    TG_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(resblock2_kernel<BF16>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, kLdsTotal));
    TG_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(resblock2_kernel<F16>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, kLdsTotal));
    attr_done = true;
  }
  const dim3 grid((unsigned)blocks), blk(512);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == TG_F16) hipLaunchKernelGGL(resblock2_kernel<F16>, grid, blk, kLdsTotal, st, k);
  else hipLaunchKernelGGL(resblock2_kernel<BF16>, grid, blk, kLdsTotal, st, k);
  return tg_launch_status();
}
