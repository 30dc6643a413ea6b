// 4x4 stride-2 padding-1 convolution forward (the discriminator's down-sampling convs, code/models.py:90-94) on MFMA:
//
//   out[y][x][co] = sum_{ky,kx in 0..3} in[2y-1+ky][2x-1+kx][ci] * W[ky*4+kx][co][ci]     (+bias) and per-channel
//   sum / sum of squares of the stored output (the BatchNorm batch statistics of the layer that follows).
//
// tg_conv runs this shape through its generic tap-table path, where in-kernel stamps showed ~8600 cycles of staging and ~4300
// cycles of k-loop per 32-channel stage for 1000 cycles of MFMA (per-load address arithmetic, a dependent LDS table read per
// tap, no overlap between stages): 26-48 us per launch whatever the layer size.  This kernel is the 3x3 pipelined path's
// structure for the 16-tap stride-2 pattern: compile-time taps (every pixel-fragment read is one lane register + an immediate),
// the next chunk's global loads in flight during the current chunk's 128 MFMAs per wave, weights in swizzled conflict-free
// 64-byte rows.  The input patch keeps 80-byte rows: for lanes 2 pixels apart that padding IS conflict-free
// (tools/lds_layout.py).  A workgroup owns 8 x 16 output pixels x 64 channels.
#include "common.h"

namespace {

constexpr int kWRow = 64;                  // weights: unpadded rows, piece index XOR 2*(bit 2 of row)
__device__ __forceinline__ int swz(int row, int piece) { return row * kWRow + ((piece ^ ((row >> 1) & 2)) << 4); }
constexpr int kARow = 80;                  // patch rows: 64 data bytes + 16 pad
// 8 waves (two per SIMD - the staging loads of one overlap the MFMAs of its partner): WC = 2 channel halves x WP = 4 row pairs
constexpr int CT = 2, WC = 2, PT = 2, WP = 4, TH = PT * WP, CO_TILE = 16 * CT * WC, NTHR = 64 * WC * WP;
// KS = 4: the 4x4 stride-2 forward.  KS = 3: the same gather with a 3x3 window (taps dy,dx in -1..1) - that is the
// input-gradient of the conv-transpose layers (code/ops.py:45-54, autograd), with the role-swapped weight packing.
template <int KS> struct Geo {
  static constexpr int NT = KS * KS;
  static constexpr int IH_P = 2 * TH + KS - 2, IW_P = 2 * 16 + KS - 2;  // 18 x 34 (KS = 4) / 17 x 33 input pixels
  static constexpr int kPatchBytes = IH_P * IW_P * kARow;
  static constexpr int kWBytes = NT * CO_TILE * kWRow;
  static constexpr int kLds = kPatchBytes + kWBytes;
};

struct C4K {
  const char* in;
  const char* w;
  const float* bias;
  char* out;
  float* stats;
  int N, IH, IW, Cin, OH, OW, Cout, tiles_x, tiles_y, nchunks, stats_groups, stats_replicas;
  int gx, ny;  // pixel tiles, output-channel tiles (ny > 1: the grid is one-dimensional, see the kernel)
  int total;   // units (pixel tile x channel tile, padded to groups of 8 x ny) a workgroup may walk
};

template <typename T> struct MmaT;
template <> struct MmaT<BF16> {
  using Frag = bf16x8;
  __device__ __forceinline__ static f32x4 run(Frag a, Frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct MmaT<F16> {
  using Frag = f16x8;
  __device__ __forceinline__ static f32x4 run(Frag a, Frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
};
template <> struct MmaT<F32> {
  using Frag = f32x4;
  __device__ __forceinline__ static f32x4 run(Frag a, Frag b, f32x4 c) {
#pragma unroll
    for (int i = 0; i < 4; ++i) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[i], c, 0, 0, 0);
    return c;
  }
};

template <typename T, int KS>
__global__ __launch_bounds__(NTHR) void conv_s2_gather_kernel(const C4K p) {
  constexpr int NT = Geo<KS>::NT, IH_P = Geo<KS>::IH_P, IW_P = Geo<KS>::IW_P, kPatchBytes = Geo<KS>::kPatchBytes;
  using TR = ElemTraits<T>;
  using Frag = typename MmaT<T>::Frag;
  constexpr int E = TR::kVec;
  constexpr int NG = (TR::kBytes == 2) ? CT / 2 : CT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* lds_a = smem;
  char* lds_w = smem + kPatchBytes;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = wid % WC, wp = wid / WC;
  const int idx = lane & 15, g = lane >> 4;
  // Workgroups go to the 8 XCDs round-robin by linear id, and each XCD has its own L2.  The ny output-channel tiles of one
  // pixel tile read the SAME input patch: as grid.y they were gx workgroups apart - same XCD, but launched long after each
  // other, so a 168-MB input (conv_trans.4's input-gradient, 40 samples) came from HBM once per channel tile (PMC: 351 MB
  // for 210 MB algorithmic).  One-dimensional grid instead: workgroups b and b + 8 - same XCD, dispatched together -
  // are the channel tiles of one pixel tile.
  // (round 5: the grid may be SMALLER than the number of (pixel tile, channel tile) units - max_workgroups of the entry points; a
  // workgroup then walks units vb = blockIdx.x, + gridDim.x, ...: a capped launch leaves the step's other lane its CUs)
  for (int vb = (int)blockIdx.x; vb < p.total; vb += (int)gridDim.x) {
  if (vb != (int)blockIdx.x) __syncthreads();   // the previous unit's LDS reads (fragments, statistics scratch) are done
  int bx = vb, by = 0;
  if (p.ny > 1) {
    const int r = bx % (8 * p.ny), grp = bx / (8 * p.ny);
    by = r >> 3;
    bx = grp * 8 + (r & 7);
    if (bx >= p.gx) continue;  // padding of the last group (uniform per workgroup)
  }
  const int txb = bx % p.tiles_x;
  bx /= p.tiles_x;
  const int tyb = bx % p.tiles_y;
  const int n = bx / p.tiles_y;
  const int ty0 = tyb * TH, tx0 = txb * 16;
  const int iy0 = 2 * ty0 - 1, ix0 = 2 * tx0 - 1;
  const int co_base = by * CO_TILE;
  const size_t in_pix = (size_t)p.Cin * TR::kBytes;
  const char* in_n = p.in + (size_t)n * p.IH * p.IW * in_pix;

  float bias_r[NG][E];
#pragma unroll
  for (int a = 0; a < NG; ++a) {
    const int ch0 = (TR::kBytes == 2) ? co_base + (wc * CT + 2 * a) * 16 + 8 * g : co_base + (wc * CT + a) * 16 + 4 * g;
#pragma unroll
    for (int e = 0; e < E; e += 4) {
      f32x4 t = {0.f, 0.f, 0.f, 0.f};
      if (p.bias) t = *reinterpret_cast<const f32x4*>(p.bias + ch0 + e);
      bias_r[a][e] = t[0]; bias_r[a][e + 1] = t[1]; bias_r[a][e + 2] = t[2]; bias_r[a][e + 3] = t[3];
    }
  }

  f32x4 acc[CT][PT];
#pragma unroll
  for (int a = 0; a < CT; ++a)
#pragma unroll
    for (int b = 0; b < PT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  // patch: 18 x 34 pixels x 4 pieces = 2448 pieces -> 10 per thread (KS = 4); weights: NT blocks of 256 pieces -> NT per thread
  constexpr int NPIECE = IH_P * IW_P * 4, UA = (NPIECE + NTHR - 1) / NTHR;
  constexpr int WPIECES = CO_TILE * 4, UW = (NT * WPIECES + NTHR - 1) / NTHR;  // weight pieces of one chunk, per thread
  constexpr int kDivMul = (65536 + IW_P - 1) / IW_P;  // prow / IW_P == (prow * kDivMul) >> 16, exact for prow < 612 (IW_P 33, 34)
  u32x4 va[UA], vw[UW];
  int da[UA];
  bool ok[UA];
  auto issue = [&](int c0) {
#pragma unroll
    for (int u = 0; u < UA; ++u) {
      const int i = min(tid + u * NTHR, NPIECE - 1);
      const int s = i & 3, prow = i >> 2;
      const int py = (prow * kDivMul) >> 16, px = prow - py * IW_P;
      const int iy = iy0 + py, ix = ix0 + px;
      da[u] = (tid + u * NTHR < NPIECE) ? prow * kARow + s * 16 : -1;
      ok[u] = iy >= 0 && iy < p.IH && ix >= 0 && ix < p.IW;
      const int cy = min(max(iy, 0), p.IH - 1), cx = min(max(ix, 0), p.IW - 1);  // clamped load, zeroed at the store
      va[u] = *reinterpret_cast<const u32x4*>(in_n + ((size_t)cy * p.IW + cx) * in_pix + (size_t)c0 * 64 + s * 16);
    }
#pragma unroll
    for (int k = 0; k < UW; ++k) {  // packed weights [slot][chunk][Cout rows][64 B]: CO_TILE consecutive rows per slot
      const int i = min(tid + k * NTHR, NT * WPIECES - 1), t = i / WPIECES, piece = i - t * WPIECES;
      vw[k] = *reinterpret_cast<const u32x4*>(p.w + (((size_t)t * p.nchunks + c0) * p.Cout + co_base) * 64 + piece * 16);
    }
  };
  auto store = [&]() {
#pragma unroll
    for (int u = 0; u < UA; ++u)
      if (da[u] >= 0) *reinterpret_cast<u32x4*>(lds_a + da[u]) = ok[u] ? va[u] : u32x4{0u, 0u, 0u, 0u};
#pragma unroll
    for (int k = 0; k < UW; ++k) {
      const int i = tid + k * NTHR, t = i / WPIECES, piece = i - t * WPIECES;
      if (i < NT * WPIECES) *reinterpret_cast<u32x4*>(lds_w + t * CO_TILE * kWRow + swz(piece >> 2, piece & 3)) = vw[k];
    }
  };

  int xb[PT];  // lane address of input pixel (2*(wp*PT+b), 2*idx) of the patch = tap (0,0) of output pixel (b, idx)
#pragma unroll
  for (int b = 0; b < PT; ++b) xb[b] = ((wp * PT + b) * 2 * IW_P + 2 * idx) * kARow + g * 16;
  const int wbase = wc * CT * 16 * kWRow + swz(idx, g);  // + multiples of 16 rows: bit 2 unchanged

  issue(0);
  for (int c0 = 0; c0 < p.nchunks; ++c0) {
    __syncthreads();  // the previous chunk's fragment reads are done
    store();
    __syncthreads();
    if (c0 + 1 < p.nchunks) issue(c0 + 1);  // in flight during the MFMAs below
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int toff = ((t / KS) * IW_P + (t % KS)) * kARow;  // compile-time after unrolling
      Frag wf[CT];
#pragma unroll
      for (int a = 0; a < CT; ++a) wf[a] = *reinterpret_cast<const Frag*>(lds_w + (t * CO_TILE + a * 16) * kWRow + wbase);
#pragma unroll
      for (int b = 0; b < PT; ++b) {
        const Frag xf = *reinterpret_cast<const Frag*>(lds_a + xb[b] + toff);
#pragma unroll
        for (int a = 0; a < CT; ++a) acc[a][b] = MmaT<T>::run(wf[a], xf, acc[a][b]);
      }
    }
  }

  // epilogue: +bias, store, per-channel statistics of what was stored
  float s1[NG][E], s2[NG][E];
#pragma unroll
  for (int a = 0; a < NG; ++a)
#pragma unroll
    for (int e = 0; e < E; ++e) s1[a][e] = s2[a][e] = 0.f;
#pragma unroll
  for (int b = 0; b < PT; ++b) {
    const int oy = ty0 + wp * PT + b, ox = tx0 + idx;
    const bool valid = oy < p.OH && ox < p.OW;
    const size_t pix = ((size_t)n * p.OH + oy) * p.OW + ox;
#pragma unroll
    for (int a = 0; a < NG; ++a) {
      float v[E];
      int ch0;
      if constexpr (TR::kBytes == 2) {
        ch0 = co_base + (wc * CT + 2 * a) * 16 + 8 * g;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v[j] = acc[2 * a][b][j];
          v[4 + j] = acc[2 * a + 1][b][j];
        }
      } else {
        ch0 = co_base + (wc * CT + a) * 16 + 4 * g;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = acc[a][b][j];
      }
      if (valid) {
#pragma unroll
        for (int e = 0; e < E; ++e) v[e] += bias_r[a][e];
        Vec<T>::store(p.out + (pix * p.Cout + ch0) * TR::kBytes, v);
        if (p.stats) {
#pragma unroll
          for (int e = 0; e < E; ++e) {
            s1[a][e] += v[e];
            s2[a][e] += v[e] * v[e];
          }
        }
      }
    }
  }
  if (p.stats) {  // uniform
#pragma unroll
    for (int a = 0; a < NG; ++a)
#pragma unroll
      for (int e = 0; e < E; ++e) {
#pragma unroll
        for (int m = 1; m < 16; m <<= 1) {
          s1[a][e] += __shfl_xor(s1[a][e], m);
          s2[a][e] += __shfl_xor(s2[a][e], m);
        }
      }
    __syncthreads();  // all fragment reads finished: LDS becomes the cross-wave reduction scratch
    float* red = reinterpret_cast<float*>(smem);  // [WP][2][CO_TILE]
    if (idx == 0) {
#pragma unroll
      for (int a = 0; a < NG; ++a) {
        const int cl0 = (TR::kBytes == 2) ? (wc * CT + 2 * a) * 16 + 8 * g : (wc * CT + a) * 16 + 4 * g;
#pragma unroll
        for (int e = 0; e < E; ++e) {
          red[(wp * 2 + 0) * CO_TILE + cl0 + e] = s1[a][e];
          red[(wp * 2 + 1) * CO_TILE + cl0 + e] = s2[a][e];
        }
      }
    }
    __syncthreads();
    const int grp = n / (p.N / p.stats_groups);
    if (tid < 2 * CO_TILE) {
      const int which = tid / CO_TILE, chn = tid - which * CO_TILE;
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < WP; ++w) s += red[(w * 2 + which) * CO_TILE + chn];
      const size_t rep = (size_t)(vb & (p.stats_replicas - 1)) * p.stats_groups * 2 * p.Cout;
      atomicAdd(p.stats + rep + ((size_t)grp * 2 + which) * p.Cout + co_base + chn, s);
    }
  }
  }   // units of this workgroup
}

}  // namespace

namespace {
template <int KS>
int launch_s2(int dtype, const void* in, const void* w_packed, const float* bias, void* out, float* stats, int stats_groups,
              int stats_replicas, int N, int IH, int IW, int Cin, int Cout, void* stream, int max_workgroups = 0) {
  C4K k;
  k.in = (const char*)in; k.w = (const char*)w_packed; k.bias = bias; k.out = (char*)out; k.stats = stats;
  k.N = N; k.IH = IH; k.IW = IW; k.Cin = Cin; k.OH = IH / 2; k.OW = IW / 2; k.Cout = Cout;
  k.stats_groups = stats ? stats_groups : 1;
  k.stats_replicas = stats && stats_replicas > 1 ? stats_replicas : 1;
  k.nchunks = Cin / (dtype == TG_F32 ? 16 : 32);
  k.tiles_x = (k.OW + 15) / 16; k.tiles_y = (k.OH + TH - 1) / TH;
  const long long gx = (long long)k.tiles_x * k.tiles_y * N;
  k.ny = Cout / CO_TILE;
  const long long total = k.ny > 1 ? (gx + 7) / 8 * 8 * k.ny : gx;
  if (total > 0x7fffffffLL) return TG_E_UNSUPPORTED;
  k.gx = (int)gx;
  k.total = (int)total;
  // max_workgroups (tg_conv4s2_fwd_capped: the discriminator's cap) or the A/B hooks TECOGAN_S2_CAP_CT (KS 3: the conv-transposes'
  // input-gradients, lane A - every cap costs) / TECOGAN_S2_CAP (KS 4) (profiles/r05_zz_s2_cap_ab.log); 0: one workgroup per unit
  static const int env_cap = [] { const char* e = getenv(KS == 3 ? "TECOGAN_S2_CAP_CT" : "TECOGAN_S2_CAP"); return e ? atoi(e) : 0; }();
  const int cap = env_cap > 0 ? env_cap : max_workgroups;
  const long long ngrid = cap > 0 && cap < total ? (long long)((cap + 7) / 8 * 8) : total;   // (a multiple of 8: units b and b + 8 stay on one XCD)
  dim3 grid((unsigned)ngrid, 1);
  hipStream_t st = (hipStream_t)stream;
  constexpr int lds = Geo<KS>::kLds;
  static std::atomic<bool> attr_done{false};
  if (!attr_done) {
    TG_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_s2_gather_kernel<BF16, KS>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    TG_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_s2_gather_kernel<F32, KS>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    TG_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_s2_gather_kernel<F16, KS>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    attr_done = true;
  }
  if (dtype == TG_BF16) hipLaunchKernelGGL((conv_s2_gather_kernel<BF16, KS>), grid, dim3(NTHR), lds, st, k);
  else if (dtype == TG_F16) hipLaunchKernelGGL((conv_s2_gather_kernel<F16, KS>), grid, dim3(NTHR), lds, st, k);
  else hipLaunchKernelGGL((conv_s2_gather_kernel<F32, KS>), grid, dim3(NTHR), lds, st, k);
  return tg_launch_status();
}

int check_s2(int dtype, const void* in, const void* w_packed, const float* bias, void* out, int N, int IH, int IW, int Cin,
             int Cout) {
  if (!in || !w_packed || !out || N <= 0 || IH <= 0 || IW <= 0 || Cin <= 0 || Cout <= 0) return TG_E_BADARG;
  if (dtype != TG_BF16 && dtype != TG_F32 && dtype != TG_F16) return TG_E_BADARG;
  if ((IH & 1) || (IW & 1)) return TG_E_UNSUPPORTED;
  if (Cin % 32 || Cout % 32) return TG_E_ALIGN;
  if (Cout % CO_TILE) return TG_E_UNSUPPORTED;  // run tg_conv instead
  if (!tg_aligned16(in) || !tg_aligned16(w_packed) || !tg_aligned16(out) || (bias && !tg_aligned16(bias))) return TG_E_ALIGN;
  return TG_OK;
}
}  // namespace

extern "C" int tg_conv4s2_fwd(int dtype, const void* in, const void* w_packed, const float* bias, void* out, float* stats,
                              int stats_groups, int stats_replicas, int N, int IH, int IW, int Cin, int Cout, void* stream) {
  const int rc = check_s2(dtype, in, w_packed, bias, out, N, IH, IW, Cin, Cout);
  if (rc != TG_OK) return rc;
  if (stats && (stats_groups <= 0 || N % stats_groups)) return TG_E_BADARG;
  if (stats && (stats_replicas < 1 || (stats_replicas & (stats_replicas - 1)))) return TG_E_BADARG;  // a power of two
  return launch_s2<4>(dtype, in, w_packed, bias, out, stats, stats_groups, stats_replicas, N, IH, IW, Cin, Cout, stream);
}

extern "C" int tg_conv4s2_fwd_capped(int dtype, const void* in, const void* w_packed, const float* bias, void* out, float* stats,
                                     int stats_groups, int stats_replicas, int N, int IH, int IW, int Cin, int Cout,
                                     int max_workgroups, void* stream) {
  const int rc = check_s2(dtype, in, w_packed, bias, out, N, IH, IW, Cin, Cout);
  if (rc != TG_OK) return rc;
  if (stats && (stats_groups <= 0 || N % stats_groups)) return TG_E_BADARG;
  if (stats && (stats_replicas < 1 || (stats_replicas & (stats_replicas - 1)))) return TG_E_BADARG;  // a power of two
  return launch_s2<4>(dtype, in, w_packed, bias, out, stats, stats_groups, stats_replicas, N, IH, IW, Cin, Cout, stream,
                      max_workgroups > 0 ? max_workgroups : 0);
}

extern "C" int tg_convt_dgrad(int dtype, const void* dout, const void* w_dgrad_packed, void* din, int N, int OH, int OW,
                              int Cout, int Cin, void* stream) {
  // din[y][x][ci] = sum_{dy,dx in -1..1} dout[2y+dy][2x+dx][co] * W[slot][ci][co]: the 3x3-window stride-2 gather
  const int rc = check_s2(dtype, dout, w_dgrad_packed, nullptr, din, N, OH, OW, Cout, Cin);
  if (rc != TG_OK) return rc;
  return launch_s2<3>(dtype, dout, w_dgrad_packed, nullptr, din, nullptr, 1, 1, N, OH, OW, Cout, Cin, stream);
}
