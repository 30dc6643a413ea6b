// Conv-transpose (kernel 3, stride 2, padding 1, output_padding 1) forward as ONE sub-pixel launch.
//
//   out[2y+oy][2x+ox][co] = sum over the taps of sub-pixel class (oy,ox) of  in[y+dy][x+dx][ci] * W[slot][co][ci]
//     (0,0): (0,0,4)              (0,1): (0,1,3) (0,0,5)              code/ops.py:45-54 (conv2_tran),
//     (1,0): (1,0,1) (0,0,7)      (1,1): (1,1,0) (1,0,2) (0,1,6) (0,0,8)       code/models.py:72,74
//
// tg_conv runs the four classes as four sets of workgroups (blockIdx.z): every class re-stages the same input patch and the
// launch is four times as many workgroups of 1-4 taps each - on the generator's recurrent pass (4 x 64x64 pixels,
// 128 -> 128 channels) that was 1024 latency-bound workgroups and 29 us, the most expensive forward layer of the chain.
// Here a workgroup owns an 8 x 16 INPUT pixel tile x 64 output channels and all four classes: the (8+1) x (16+1) patch is
// staged once per 32-channel chunk, the 9 (class, tap) weight blocks are exactly the 9 kernel slots, and the k-loop is a
// 3x3-conv-sized loop (9 MFMA groups per chunk) into four accumulator sets.  Staging, swizzled conflict-free LDS rows and
// the chunk pipeline are those of the 3x3 path of conv_mfma.hip; bf16 and fp32 share the code.
#include "common.h"

namespace {

constexpr int kRow = 64, kPitch = 24;  // unpadded 64-byte rows, piece index XOR 2*(bit 2 of row); patch pitch 24 rows
__device__ __forceinline__ int swz(int row, int piece) { return row * kRow + ((piece ^ ((row >> 1) & 2)) << 4); }

// 8 waves (two per SIMD - the staging loads of one overlap the MFMAs of its partner): WC = 2 channel halves x WP = 4 row pairs
constexpr int CT = 2, WC = 2, PT = 2, WP = 4, TH = PT * WP, CO_TILE = 16 * CT * WC, NTHR = 64 * WC * WP;  // 64 ch x (8 x 16 px)
// Sub-pixel patterns: (class, tap) pair s uses weight slot s, window position (dy, dx) relative to the window origin.
//   PAT 0  conv-transpose k3 s2 p1 op1 forward: window 2x2 at origin (0,0), 9 pairs (table in the header comment)
//   PAT 1  input-gradient of the 4x4 stride-2 conv (code/models.py:90-94, autograd): window 3x3 at origin (-1,-1), 16 pairs:
//            (0,0): (0,0,5) (0,-1,7) (-1,0,13) (-1,-1,15)      (0,1): (0,1,4) (0,0,6) (-1,1,12) (-1,0,14)
//            (1,0): (1,0,1) (1,-1,3) (0,0,9)  (0,-1,11)        (1,1): (1,1,0) (1,0,2) (0,1,8)   (0,0,10)
template <int PAT> struct Pat;
template <> struct Pat<0> {
  static constexpr int NT = 9, WIN = 2, ORG = 0;
  static constexpr int cls[9] = {3, 2, 3, 1, 0, 1, 3, 2, 3};
  static constexpr int dy[9] = {1, 1, 1, 0, 0, 0, 0, 0, 0};
  static constexpr int dx[9] = {1, 0, 0, 1, 0, 0, 1, 0, 0};
};
template <> struct Pat<1> {
  static constexpr int NT = 16, WIN = 3, ORG = -1;
  static constexpr int cls[16] = {3, 2, 3, 2, 1, 0, 1, 0, 3, 2, 3, 2, 1, 0, 1, 0};
  static constexpr int dy[16] = {2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0};   // window row = tap dy + 1
  static constexpr int dx[16] = {2, 1, 1, 0, 2, 1, 1, 0, 2, 1, 1, 0, 2, 1, 1, 0};
};
template <int PAT> struct PGeo {
  static constexpr int IH_P = TH + Pat<PAT>::WIN - 1, IW_P = 16 + Pat<PAT>::WIN - 1;
  static constexpr int kPatchBytes = IH_P * kPitch * kRow;            // one chunk
  static constexpr int kWBytes = Pat<PAT>::NT * CO_TILE * kRow;       // one chunk: NT slots x 64 rows
  static constexpr int kLds = kPatchBytes + kWBytes;
};

struct ConvtK {
  const char* in;
  const char* w;
  const float* bias;
  char* out;
  const char* mask;  // optional: multiply by act'(mask) (the saved activation of the layer below), TG_MASK_*
  int N, IH, IW, Cin, Cout, tiles_x, tiles_y, nchunks, act, mask_mode;
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

template <typename T, int PAT>
__global__ __launch_bounds__(NTHR) void subpixel_kernel(const ConvtK p) {
  using P = Pat<PAT>;
  constexpr int NT = P::NT, WIN = P::WIN, ORG = P::ORG, IH_P = PGeo<PAT>::IH_P, IW_P = PGeo<PAT>::IW_P;
  constexpr int kPatchBytes = PGeo<PAT>::kPatchBytes;
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
  int bx = blockIdx.x;
  const int txb = bx % p.tiles_x;
  bx /= p.tiles_x;
  const int tyb = bx % p.tiles_y;
  const int n = bx / p.tiles_y;
  const int ty0 = tyb * TH, tx0 = txb * 16;
  const int co_base = blockIdx.y * CO_TILE;
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

  f32x4 acc[4][CT][PT];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int a = 0; a < CT; ++a)
#pragma unroll
      for (int b = 0; b < PT; ++b) acc[c][a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  // patch: IH_P x IW_P pixels x 4 pieces (612 for the 9 x 17 patch) -> 3 per thread; weights: NT blocks of 256 pieces
  constexpr int NPIECE = IH_P * IW_P * 4, UA = (NPIECE + NTHR - 1) / NTHR;
  constexpr int WPIECES = CO_TILE * 4, UW = (NT * WPIECES + NTHR - 1) / NTHR;  // weight pieces of one chunk, per thread
  constexpr int kDivMul = (65536 + IW_P - 1) / IW_P;  // prow / IW_P == (prow * kDivMul) >> 16 (exact for prow < 256)
  u32x4 va[UA], vw[UW];
  int da[UA];
  bool ok[UA];
  auto issue = [&](int c0) {
#pragma unroll
    for (int u = 0; u < UA; ++u) {
      const int i = min(tid + u * NTHR, NPIECE - 1);
      const int s = i & 3, prow = i >> 2;
      const int py = (prow * kDivMul) >> 16, px = prow - py * IW_P;
      const int iy = ty0 + ORG + py, ix = tx0 + ORG + px;
      da[u] = (tid + u * NTHR < NPIECE) ? swz(py * kPitch + px, s) : -1;
      ok[u] = iy >= 0 && iy < p.IH && ix >= 0 && ix < p.IW;
      const int cy = min(max(iy, 0), p.IH - 1), cx = min(max(ix, 0), p.IW - 1);  // clamped load, zeroed at the store
      va[u] = *reinterpret_cast<const u32x4*>(in_n + ((size_t)cy * p.IW + cx) * in_pix + (size_t)c0 * 64 + s * 16);
    }
#pragma unroll
    for (int k = 0; k < UW; ++k) {  // packed weights [slot][chunk][Cout rows][64 B]: CO_TILE consecutive rows per slot
      const int i = min(tid + k * NTHR, NT * WPIECES - 1), s9 = i / WPIECES, piece = i - s9 * WPIECES;
      vw[k] = *reinterpret_cast<const u32x4*>(p.w + (((size_t)s9 * p.nchunks + c0) * p.Cout + co_base) * 64 + piece * 16);
    }
  };
  auto store = [&]() {
#pragma unroll
    for (int u = 0; u < UA; ++u)
      if (da[u] >= 0) *reinterpret_cast<u32x4*>(lds_a + da[u]) = ok[u] ? va[u] : u32x4{0u, 0u, 0u, 0u};
#pragma unroll
    for (int k = 0; k < UW; ++k) {
      const int i = tid + k * NTHR, s9 = i / WPIECES, piece = i - s9 * WPIECES;
      if (i < NT * WPIECES) *reinterpret_cast<u32x4*>(lds_w + s9 * CO_TILE * kRow + swz(piece >> 2, piece & 3)) = vw[k];
    }
  };

  // lane addresses: pixel (row wp*PT+b [+dy], column idx [+dx]); the pitch of 24 keeps bit 2 of the row independent of dy
  int xb[PT][WIN];
#pragma unroll
  for (int b = 0; b < PT; ++b)
#pragma unroll
    for (int c = 0; c < WIN; ++c) xb[b][c] = swz((wp * PT + b) * kPitch + idx + c, g);
  const int wbase = wc * CT * 16 * kRow + swz(idx, g);  // + multiples of 16 rows: bit 2 unchanged

  issue(0);
  for (int c0 = 0; c0 < p.nchunks; ++c0) {
    __syncthreads();  // the previous chunk's fragment reads are done
    store();
    __syncthreads();
    if (c0 + 1 < p.nchunks) issue(c0 + 1);  // in flight during the MFMAs below
    Frag xf[WIN][WIN][PT];
#pragma unroll
    for (int r = 0; r < WIN; ++r)
#pragma unroll
      for (int c = 0; c < WIN; ++c)
#pragma unroll
        for (int b = 0; b < PT; ++b)
          xf[r][c][b] = *reinterpret_cast<const Frag*>(lds_a + xb[b][c] + r * kPitch * kRow);
#pragma unroll
    for (int s9 = 0; s9 < NT; ++s9) {
      Frag wf[CT];
#pragma unroll
      for (int a = 0; a < CT; ++a) wf[a] = *reinterpret_cast<const Frag*>(lds_w + (s9 * CO_TILE + a * 16) * kRow + wbase);
#pragma unroll
      for (int b = 0; b < PT; ++b)
#pragma unroll
        for (int a = 0; a < CT; ++a)
          acc[P::cls[s9]][a][b] = MmaT<T>::run(wf[a], xf[P::dy[s9]][P::dx[s9]][b], acc[P::cls[s9]][a][b]);
    }
  }

  // epilogue: class (oy, ox) of input pixel (cy, cx) is output pixel (2cy+oy, 2cx+ox)
  const int OH = 2 * p.IH, OW = 2 * p.IW;
  // all mask vectors of the epilogue up front, from clamped addresses: under the divergent bounds test below every load
  // would be waited for before the next is issued (PT * 4 * NG dependent round trips)
  u32x4 mpre[PT][4][NG];
  if (p.mask_mode != TG_MASK_NONE) {  // uniform
#pragma unroll
    for (int b = 0; b < PT; ++b) {
      const int cy = min(ty0 + wp * PT + b, p.IH - 1), cx = min(tx0 + idx, p.IW - 1);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const size_t pix = ((size_t)n * OH + 2 * cy + (c >> 1)) * OW + 2 * cx + (c & 1);
#pragma unroll
        for (int a = 0; a < NG; ++a) {
          const int ch0 = (TR::kBytes == 2) ? co_base + (wc * CT + 2 * a) * 16 + 8 * g : co_base + (wc * CT + a) * 16 + 4 * g;
          mpre[b][c][a] = *reinterpret_cast<const u32x4*>(p.mask + (pix * p.Cout + ch0) * TR::kBytes);
        }
      }
    }
  }
#pragma unroll
  for (int b = 0; b < PT; ++b) {
    const int cy = ty0 + wp * PT + b, cx = tx0 + idx;
    if (cy < p.IH && cx < p.IW) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const size_t pix = ((size_t)n * OH + 2 * cy + (c >> 1)) * OW + 2 * cx + (c & 1);
#pragma unroll
        for (int a = 0; a < NG; ++a) {
          float v[E];
          int ch0;
          if constexpr (TR::kBytes == 2) {
            ch0 = co_base + (wc * CT + 2 * a) * 16 + 8 * g;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              v[j] = acc[c][2 * a][b][j];
              v[4 + j] = acc[c][2 * a + 1][b][j];
            }
          } else {
            ch0 = co_base + (wc * CT + a) * 16 + 4 * g;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = acc[c][a][b][j];
          }
#pragma unroll
          for (int e = 0; e < E; ++e) {
            v[e] += bias_r[a][e];
            if (p.act == TG_ACT_RELU) v[e] = v[e] > 0.f ? v[e] : 0.f;
            else if (p.act == TG_ACT_LRELU) v[e] = v[e] > 0.f ? v[e] : 0.2f * v[e];
          }
          if (p.mask_mode != TG_MASK_NONE) {
            float m[E];
            Vec<T>::load(&mpre[b][c][a], m);
            const float neg = p.mask_mode == TG_MASK_LRELU ? 0.2f : 0.f;
#pragma unroll
            for (int e = 0; e < E; ++e) v[e] *= (m[e] > 0.f ? 1.f : neg);
          }
          Vec<T>::store(p.out + (pix * p.Cout + ch0) * TR::kBytes, v);
        }
      }
    }
  }
}

}  // namespace

namespace {
template <int PAT>
int launch_subpixel(int dtype, const void* in, const void* w_packed, const float* bias, void* out, int N, int IH, int IW,
                    int Cin, int Cout, int act, void* stream, const void* mask = nullptr, int mask_mode = TG_MASK_NONE) {
  ConvtK k;
  k.in = (const char*)in; k.w = (const char*)w_packed; k.bias = bias; k.out = (char*)out;
  k.mask = (const char*)mask; k.mask_mode = mask ? mask_mode : TG_MASK_NONE;
  k.N = N; k.IH = IH; k.IW = IW; k.Cin = Cin; k.Cout = Cout; k.act = act;
  k.nchunks = Cin / (dtype == TG_F32 ? 16 : 32);
  k.tiles_x = (IW + 15) / 16; k.tiles_y = (IH + TH - 1) / TH;
  const long long gx = (long long)k.tiles_x * k.tiles_y * N;
  if (gx > 0x7fffffffLL) return TG_E_UNSUPPORTED;
  dim3 grid((unsigned)gx, (unsigned)(Cout / CO_TILE));
  hipStream_t st = (hipStream_t)stream;
  constexpr int lds = PGeo<PAT>::kLds;
  static std::atomic<bool> attr_done{false};
  if (!attr_done) {
    TG_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(subpixel_kernel<BF16, PAT>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    TG_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(subpixel_kernel<F32, PAT>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    TG_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(subpixel_kernel<F16, PAT>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    attr_done = true;
  }
  if (dtype == TG_BF16) hipLaunchKernelGGL((subpixel_kernel<BF16, PAT>), grid, dim3(NTHR), lds, st, k);
  else if (dtype == TG_F16) hipLaunchKernelGGL((subpixel_kernel<F16, PAT>), grid, dim3(NTHR), lds, st, k);
  else hipLaunchKernelGGL((subpixel_kernel<F32, PAT>), grid, dim3(NTHR), lds, st, k);
  return tg_launch_status();
}

int check_subpixel(int dtype, const void* in, const void* w_packed, const float* bias, void* out, int N, int IH, int IW,
                   int Cin, int Cout, int act) {
  if (!in || !w_packed || !out || N <= 0 || IH <= 0 || IW <= 0 || Cin <= 0 || Cout <= 0) return TG_E_BADARG;
  if (dtype != TG_BF16 && dtype != TG_F32 && dtype != TG_F16) return TG_E_BADARG;
  if (act != TG_ACT_NONE && act != TG_ACT_RELU && act != TG_ACT_LRELU) return TG_E_UNSUPPORTED;
  if (Cin % 32 || Cout % 32) return TG_E_ALIGN;
  if (Cout % CO_TILE) return TG_E_UNSUPPORTED;  // run tg_conv with the four-class descriptor instead
  if (!tg_aligned16(in) || !tg_aligned16(w_packed) || !tg_aligned16(out) || (bias && !tg_aligned16(bias))) return TG_E_ALIGN;
  return TG_OK;
}
}  // namespace

extern "C" int tg_convt_fwd(int dtype, const void* in, const void* w_packed, const float* bias, void* out, int N, int IH,
                            int IW, int Cin, int Cout, int act, void* stream) {
  const int rc = check_subpixel(dtype, in, w_packed, bias, out, N, IH, IW, Cin, Cout, act);
  return rc != TG_OK ? rc : launch_subpixel<0>(dtype, in, w_packed, bias, out, N, IH, IW, Cin, Cout, act, stream);
}

extern "C" int tg_conv4s2_dgrad(int dtype, const void* dout, const void* w_dgrad_packed, void* din, int N, int OH, int OW,
                                int Cout, int Cin, const void* mask, int mask_mode, void* stream) {
  // dout [N][OH][OW][Cout] -> din [N][2OH][2OW][Cin]: the four sub-pixel classes of the 4x4 stride-2 conv's input-gradient
  const int rc = check_subpixel(dtype, dout, w_dgrad_packed, nullptr, din, N, OH, OW, Cout, Cin, TG_ACT_NONE);
  if (rc != TG_OK) return rc;
  if (mask && (mask_mode < TG_MASK_NONE || mask_mode > TG_MASK_LRELU || !tg_aligned16(mask))) return TG_E_BADARG;
  return launch_subpixel<1>(dtype, dout, w_dgrad_packed, nullptr, din, N, OH, OW, Cout, Cin, TG_ACT_NONE, stream, mask,
                            mask_mode);
}
