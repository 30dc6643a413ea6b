// Gather convolution on the CDNA4 matrix cores (implicit GEMM, no im2col buffer).
//
//   D[co][px] += W[co][k] * X[k][px]      k = (tap, input channel)
//
// The MFMA "A" operand is the weight tile (rows = output channels) and the "B" operand the activation tile
// (columns = 16 consecutive output pixels of one image row), so that one lane's accumulator registers are
// consecutive CHANNELS of one pixel: the NHWC epilogue store is one 16-byte store per lane per tile (pair).
//
// Work decomposition: one 256-thread workgroup (4 waves) owns CO_TILE output channels x (TH rows x 16 px) of
// one output sub-lattice ("class") of one image.  K is walked in 64-byte channel chunks (32 bf16 / 16 f32
// channels): per chunk the activation halo patch is staged ONCE into LDS and reused by every tap; the packed
// weight rows of a group of taps are staged next to it.  Both element types share this code; only the MFMA
// issue (1 x 16x16x32 bf16 vs 4 x 16x16x4 f32 per 16-byte fragment) and the accumulator->channel map differ.
//
// Replaces aten::conv2d / conv_transpose2d / convolution_backward(input) call sites of the reference:
// code/models.py:54-58,68,72-76,90-94,102 (via code/ops.py:45-63) and the autograd of code/train.py:336,340.
#include "common.h"

#ifdef TG_STAMP
// Diagnostic build only (build.sh -DTG_STAMP): workgroup 0 records s_memtime at phase boundaries into a buffer of its own
// that nothing else reads; the production library contains no stamp.
__device__ long long tg_stamps[64];
#define TG_STAMP_AT(i)                                                                 \
  do {                                                                                 \
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) {   \
      tg_stamps[i] = (long long)__builtin_amdgcn_s_memtime();                         \
    }                                                                                  \
  } while (0)
extern "C" int tg_debug_read_stamps(long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(tg_stamps), sizeof(long long) * n);
}
#else
#define TG_STAMP_AT(i) do {} while (0)
#endif

namespace {

constexpr int kRowBytes = 80;  // generic path: 64 data bytes + 16 pad, ds_read_b128 fragment reads at most 2-way conflicted
// STD3 path: unpadded 64-byte rows whose 16-byte piece index is XOR-ed with 2*(bit 2 of the row), patch pitch 24 rows.
// ds_read_b128 serves a wave in four fixed groups of 16 lanes ({0-3,12-15,20-27}, ...: MI355X_MICROARCH.md, LDS); with
// this swizzle 16 lanes on 16 consecutive rows - at any base row, i.e. for every tap - hit 16 distinct 16-byte slots, so
// every fragment read takes the minimum 4 LDS cycles instead of 8 (tools/lds_layout.py counts them exhaustively).  The
// pitch of 24 (a multiple of 8) keeps bit 2 of a row independent of the tap's row offset, so the three column taps need
// three precomputed lane addresses and everything else is an immediate offset.
constexpr int kSwzRow = 64, kSwzPitch = 24;
__device__ __forceinline__ int swz_off(int row, int piece) { return row * kSwzRow + ((piece ^ ((row >> 1) & 2)) << 4); }

// All fields are 32-bit on purpose: the struct lives in the kernarg segment and is indexed by blockIdx.z / the tap index;
// with 8/16-bit members hipcc lowers every such access to a VECTOR global_load_ubyte/ushort + v_readfirstlane (a ~600-cycle
// dependent round trip per access), with dwords it emits s_load_dword.
struct ConvClassK {
  int ooy, oox, ntaps, dymin, dxmin, ih, iw;
  int dy[TG_MAX_TAPS];
  int dx[TG_MAX_TAPS];
  int widx[TG_MAX_TAPS];
};

struct ConvK {
  const char* in;
  const char* w;
  const float* bias;
  const char* res;
  const char* mask;
  char* out;
  float* stats;
  int N, IH, IW, Cin, OH, OW, Cout, S, OS;
  int tiles_x, tiles_y, nchunks, tg, cg;
  int tx_log2, ty_log2;  // log2 of the tile counts, or -1
  int act, mask_mode, stats_mode, stats_groups, stats_replicas, out_mode, c_real;
  long long out_n_stride;
  int a_rows_max;
  int flip;           // STD3: taps mirrored (input-gradient launch)
  int std3;           // host-side: launch the compile-time 3x3 variant
  int slim;           // host-side: the launch qualifies for the slim epilogue (NHWC, <= LeakyReLU, no statistics)
  ConvClassK cls[TG_MAX_CLASSES];
};

template <typename T> struct Mma;
template <> struct Mma<BF16> {
  using Frag = bf16x8;
  __device__ __forceinline__ static f32x4 run(Frag a, Frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct Mma<F16> {
  using Frag = f16x8;
  __device__ __forceinline__ static f32x4 run(Frag a, Frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
};
template <> struct Mma<F32> {
  using Frag = f32x4;
  __device__ __forceinline__ static f32x4 run(Frag a, Frag b, f32x4 c) {
    // k index of (lane group g, element i) is channel 4g+i for BOTH operands, so any consistent order is a valid GEMM
#pragma unroll
    for (int i = 0; i < 4; ++i) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[i], c, 0, 0, 0);
    return c;
  }
};

__device__ __forceinline__ float apply_act(float v, int act) {
  if (act == TG_ACT_RELU) return v > 0.f ? v : 0.f;
  if (act == TG_ACT_LRELU) return v > 0.f ? v : 0.2f * v;
  if (act == TG_ACT_SIGMOID) return 1.f / (1.f + __expf(-v));
  if (act == TG_ACT_TANH24) {  // f_net output 24*tanh(v), code/models.py:50
    const float t = __expf(-2.f * fabsf(v));
    return copysignf(24.f * (1.f - t) / (1.f + t), v);
  }
  return v;
}

// STD3: the launch is a plain 3x3 stride-1 conv (forward, or dgrad = taps mirrored) whose packed slot order is the tap
// order.  Tap offsets are then compile-time, the 9-tap loop is fully unrolled and the compiler can keep many LDS fragment
// reads in flight; the generic path (tap table per class) pays a dependent table lookup per k-step and is kept for the
// stride-2 / sub-pixel launches.
// SLIM (pipelined path only): the launch stores NHWC, applies at most ReLU / LeakyReLU and takes no statistics - the common
// case of the step.  The general epilogue carries sigmoid / 24*tanh, the strided fp32-NCHW store and the statistics
// reduction behind uniform branches: 3900 instructions (31 KB) for the 64x128 tile, and workgroup 0 spent 6250 cycles in
// it against 2200 in its k-loop (TG_STAMP) - instruction fetch, not work.
template <typename T, int CT, int PT, int WC, int WP, bool STD3, bool SLIM = false>
__global__ __launch_bounds__(64 * WC * WP) void conv_gather_kernel(const ConvK p) {
  // NTHR = 512 (pipelined 3x3 path only): two waves per SIMD from ONE workgroup, for launches that cannot give a CU two
  // workgroups (<= 256 workgroups): one wave's staging-load issue (~120 cycles per 1-KiB load) overlaps its partner's MFMAs
  constexpr int NTHR = 64 * WC * WP;
  static_assert(NTHR == 256 || (NTHR == 512 && STD3), "8-wave workgroups exist for the pipelined 3x3 path");
  using TR = ElemTraits<T>;
  using Frag = typename Mma<T>::Frag;
  constexpr int CO_TILE = 16 * CT * WC;
  constexpr int TH = PT * WP;
  constexpr int E = TR::kVec;                              // channels per 16-byte epilogue vector
  constexpr int NG = (TR::kBytes == 2) ? CT / 2 : CT;      // epilogue vectors per px-tile per lane
  static_assert(WC * WP == 4 || WC * WP == 8, "4 or 8 waves per workgroup");
  static_assert(TR::kBytes == 4 || CT % 2 == 0, "bf16 pairs co-tiles");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* lds_a = smem;
  char* lds_w = smem + (size_t)p.cg * p.a_rows_max * (STD3 ? kSwzRow : kRowBytes);

  const ConvClassK& cl = p.cls[blockIdx.z];
  const int tid = threadIdx.x;
  TG_STAMP_AT(0);
  const int lane = tid & 63, wid = tid >> 6;
  const int wc = wid % WC, wp = wid / WC;
  const int idx = lane & 15, g = lane >> 4;

  // block decode without integer division when the tile counts are powers of two (every shape of the TecoGAN step)
  int bx = blockIdx.x, txb, tyb, n;
  if (p.tx_log2 >= 0 && p.ty_log2 >= 0) {
    txb = bx & (p.tiles_x - 1);
    tyb = (bx >> p.tx_log2) & (p.tiles_y - 1);
    n = bx >> (p.tx_log2 + p.ty_log2);
  } else {
    txb = bx % p.tiles_x;
    bx /= p.tiles_x;
    tyb = bx % p.tiles_y;
    n = bx / p.tiles_y;
  }
  const int co_base = blockIdx.y * CO_TILE;

  const int os_sh = p.OS - 1;  // OS is 1 or 2
  const int OHc = (p.OH - cl.ooy + p.OS - 1) >> os_sh;
  const int OWc = (p.OW - cl.oox + p.OS - 1) >> os_sh;
  const int ty0 = tyb * TH, tx0 = txb * 16;
  if (ty0 >= OHc || tx0 >= OWc) return;  // whole tile outside this class's grid (uniform per workgroup)

  const int iy0 = ty0 * p.S + cl.dymin, ix0 = tx0 * p.S + cl.dxmin;
  const int prow_n = cl.ih * cl.iw;
  const int ntaps = cl.ntaps;

  f32x4 acc[CT][PT];
#pragma unroll
  for (int a = 0; a < CT; ++a)
#pragma unroll
    for (int b = 0; b < PT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  const size_t in_pix_bytes = (size_t)p.Cin * TR::kBytes;
  const char* in_n = p.in + (size_t)n * p.IH * p.IW * in_pix_bytes;

  // bias for this lane's output channels, fetched now so that its latency hides under the K loop
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

  // K loop: stages of (chunk group x tap group).  Each stage issues ALL of its global loads before the first LDS store
  // (UA/UW loads in flight per thread): with one workgroup per CU on the small recurrent-pass layers nothing else hides
  // the L2/HBM latency, and a load->store->load chain costs one round trip per 16 bytes.
  // UA activation pieces and UW (chunk,tap) weight blocks per thread are loaded in ONE issue phase before any LDS store.
  constexpr int UA = 6;
  constexpr int PIECES = CO_TILE * 4, PPT = (PIECES + 255) / 256;
  // 512 threads and 256-piece weight blocks: the two 256-thread halves take alternate blocks (half = wave-uniform)
  constexpr bool HALVES = NTHR == 512;
  static_assert(!HALVES || PIECES == 256, "8-wave configs use 64-channel tiles");
  const int half = HALVES ? __builtin_amdgcn_readfirstlane(tid >> 8) : 0;
  // small tiles: the whole K of a 64-channel 3x3 layer in flight at once; big tiles: one 3x3 chunk (pipelined kernel) or
  // fewer blocks in the generic kernel, whose 128-channel tiles otherwise lose occupancy to the staging registers
  constexpr int UW = (64 * WC * WP == 512) ? 9 : ((CT * PT <= 4) ? 18 : ((STD3 || PPT == 1) ? 9 : 5));
  const int a_stride = p.a_rows_max * (STD3 ? kSwzRow : kRowBytes);  // LDS bytes of one chunk's activation patch
  const float inv_iw = 1.0f / (float)cl.iw, inv_prow = 1.0f / (float)prow_n;
  if constexpr (STD3) {
    // ---- 3x3 fast path, software-pipelined over chunk groups: the global loads of group i+1 are issued right after
    // group i has been written to LDS and stay in flight (in registers) while the MFMAs of group i run.  The host
    // guarantees that one group fits one issue: cg*prow_n*4 <= 256*UA pieces and cg*9 <= UW weight blocks.
    constexpr int pitch = kSwzPitch * kSwzRow;
    int xbase[PT][3];  // lane address of pixel (row wp*PT+b, column idx + c) of the patch, c = column tap
#pragma unroll
    for (int b = 0; b < PT; ++b)
#pragma unroll
      for (int c = 0; c < 3; ++c) xbase[b][c] = swz_off((wp * PT + b) * kSwzPitch + idx + c, g);
    const int wbase = swz_off(wc * CT * 16 + idx, g);  // + multiples of 16 rows: bit 2 unchanged
    u32x4 va[UA];
    int da[UA];
    u32x4 vw[UW][PPT];
    auto issue = [&](int c0, int cn) {
      const int total_a = cn * prow_n * 4;
#pragma unroll
      for (int u = 0; u < UA; ++u) {
        const int i = tid + u * NTHR;
        va[u] = u32x4{0u, 0u, 0u, 0u};
        da[u] = -1;
        if (i < total_a) {
          const int s = i & 3, r = i >> 2;
          const int cc = (int)(((float)r + 0.5f) * inv_prow), prow = r - cc * prow_n;
          const int py = (int)(((float)prow + 0.5f) * inv_iw), px = prow - py * cl.iw;
          const int iy = iy0 + py, ix = ix0 + px;
          da[u] = cc * a_stride + swz_off(py * kSwzPitch + px, s);
          if (iy >= 0 && iy < p.IH && ix >= 0 && ix < p.IW)
            va[u] = *reinterpret_cast<const u32x4*>(in_n + ((size_t)iy * p.IW + ix) * in_pix_bytes +
                                                    (size_t)(c0 + cc) * 64 + s * 16);
        }
      }
#pragma unroll
      for (int u = 0; u < UW; ++u) {
        const int blk = HALVES ? 2 * u + half : u;  // (chunk, tap) block: compile-time unless the halves alternate
        if (blk < cn * 9) {  // wave-uniform; slot == tap
          const int cc = blk / 9, tt = blk - cc * 9;
          const char* src = p.w + (((size_t)tt * p.nchunks + c0 + cc) * p.Cout + co_base) * 64;
#pragma unroll
          for (int k = 0; k < PPT; ++k) {
            const int piece = (HALVES ? (tid & 255) : tid) + k * 256;
            if (PIECES % 256 == 0 || piece < PIECES) vw[u][k] = *reinterpret_cast<const u32x4*>(src + piece * 16);
          }
        }
      }
    };
    auto store = [&](int cn) {
#pragma unroll
      for (int u = 0; u < UA; ++u)
        if (da[u] >= 0) *reinterpret_cast<u32x4*>(lds_a + da[u]) = va[u];
#pragma unroll
      for (int u = 0; u < UW; ++u) {
        const int blk = HALVES ? 2 * u + half : u;
        if (blk < cn * 9) {
          char* dstw = lds_w + blk * CO_TILE * kSwzRow;  // [cc][tt] with tg == 9
#pragma unroll
          for (int k = 0; k < PPT; ++k) {
            const int piece = (HALVES ? (tid & 255) : tid) + k * 256;
            if (PIECES % 256 == 0 || piece < PIECES)
              *reinterpret_cast<u32x4*>(dstw + swz_off(piece >> 2, piece & 3)) = vw[u][k];
          }
        }
      }
    };
    issue(0, min(p.cg, p.nchunks));
    for (int c0 = 0; c0 < p.nchunks; c0 += p.cg) {
      const int cn = min(p.cg, p.nchunks - c0);
      __syncthreads();  // previous fragment reads are done before LDS is overwritten
      TG_STAMP_AT(1);
      store(cn);
      TG_STAMP_AT(3);
      __syncthreads();
      TG_STAMP_AT(4);
      const int c1 = c0 + p.cg;
      if (c1 < p.nchunks) issue(c1, min(p.cg, p.nchunks - c1));
      for (int cc = 0; cc < cn; ++cc) {
        const char* la = lds_a + cc * a_stride;
        const char* lw = lds_w + cc * 9 * CO_TILE * kSwzRow + wbase;
        // spatial offsets in compile-time order (so every pixel read is register + immediate); the input-gradient launch
        // pairs offset `so` with weight slot 8 - so instead (taps mirrored)
#pragma unroll
        for (int so = 0; so < 9; ++so) {
          const int tt = p.flip ? 8 - so : so;
          Frag wf[CT];
#pragma unroll
          for (int a = 0; a < CT; ++a) wf[a] = *reinterpret_cast<const Frag*>(lw + (tt * CO_TILE + a * 16) * kSwzRow);
#pragma unroll
          for (int b = 0; b < PT; ++b) {
            const Frag xf = *reinterpret_cast<const Frag*>(la + xbase[b][so % 3] + (so / 3) * pitch);
#pragma unroll
            for (int a = 0; a < CT; ++a) acc[a][b] = Mma<T>::run(wf[a], xf, acc[a][b]);
          }
        }
      }
    }
  } else {
    for (int c0 = 0; c0 < p.nchunks; c0 += p.cg) {
      const int cn = min(p.cg, p.nchunks - c0);
      for (int t0 = 0; t0 < ntaps; t0 += p.tg) {
        const int tn = min(p.tg, ntaps - t0);
        __syncthreads();  // previous fragment reads are done before LDS is overwritten
        const int total_a = (t0 == 0) ? cn * prow_n * 4 : 0;
        const int nq = cn * tn;
        int a_done = 0, q_done = 0, q_cc = 0, q_tt = 0;
        while (a_done < total_a || q_done < nq) {
          u32x4 va[UA];
          int da[UA];
          u32x4 vw[UW][PPT];
          // ---- issue phase.  Patch coordinates use exact float reciprocals (no hardware integer divide; ~35 VALU each
          // would be microseconds at one wave per SIMD).
  #pragma unroll
          for (int u = 0; u < UA; ++u) {
            const int i = a_done + tid + u * 256;
            va[u] = u32x4{0u, 0u, 0u, 0u};
            da[u] = -1;
            if (i < total_a) {
              const int s = i & 3, r = i >> 2;
              const int cc = (int)(((float)r + 0.5f) * inv_prow), prow = r - cc * prow_n;
              const int py = (int)(((float)prow + 0.5f) * inv_iw), px = prow - py * cl.iw;
              const int iy = iy0 + py, ix = ix0 + px;
              da[u] = cc * a_stride + prow * kRowBytes + s * 16;
              if (iy >= 0 && iy < p.IH && ix >= 0 && ix < p.IW)
                va[u] = *reinterpret_cast<const u32x4*>(in_n + ((size_t)iy * p.IW + ix) * in_pix_bytes +
                                                        (size_t)(c0 + cc) * 64 + s * 16);
            }
          }
          // weights: for one (chunk, tap) the CO_TILE packed rows are contiguous in global memory: a plain block copy.
          // (chunk, tap) of block qq = q_done + u is carried as two wave-uniform counters: `qq / tn` by a runtime tn is a
          // ~50-instruction emulated division per block - in VALU - and that address arithmetic, not the loads, was the
          // bulk of the ~2000 instructions a staging phase of the stride-2 launches executed (in-kernel stamps:
          // 12500 cycles per phase).
          int w_cc = q_cc, w_tt = q_tt;
  #pragma unroll
          for (int u = 0; u < UW; ++u) {
            const int qq = q_done + u;  // wave-uniform
            if (qq < nq) {
              const int cc = w_cc, tt = w_tt;
              if (++w_tt == tn) { w_tt = 0; ++w_cc; }
              const int slot = cl.widx[t0 + tt];
              const char* src = p.w + (((size_t)slot * p.nchunks + c0 + cc) * p.Cout + co_base) * 64;
  #pragma unroll
              for (int k = 0; k < PPT; ++k) {
                const int piece = tid + k * 256;
                if (PIECES % 256 == 0 || piece < PIECES) vw[u][k] = *reinterpret_cast<const u32x4*>(src + piece * 16);
              }
            }
          }
          // ---- store phase
          TG_STAMP_AT(1);
  #pragma unroll
          for (int u = 0; u < UA; ++u)
            if (da[u] >= 0) *reinterpret_cast<u32x4*>(lds_a + da[u]) = va[u];
          TG_STAMP_AT(2);
          w_cc = q_cc;
          w_tt = q_tt;
  #pragma unroll
          for (int u = 0; u < UW; ++u) {
            const int qq = q_done + u;
            if (qq < nq) {
              const int cc = w_cc, tt = w_tt;
              if (++w_tt == tn) { w_tt = 0; ++w_cc; }
              char* dstw = lds_w + (cc * p.tg + tt) * CO_TILE * kRowBytes;
  #pragma unroll
              for (int k = 0; k < PPT; ++k) {
                const int piece = tid + k * 256;
                if (PIECES % 256 == 0 || piece < PIECES)
                  *reinterpret_cast<u32x4*>(dstw + (piece >> 2) * kRowBytes + (piece & 3) * 16) = vw[u][k];
              }
            }
          }
          a_done += 256 * UA;
          q_done += UW;
          q_cc = w_cc;  // (w_cc, w_tt) now describe block q_done
          q_tt = w_tt;
          TG_STAMP_AT(3);
        }
        __syncthreads();
        TG_STAMP_AT(4);

        {
          // lane-invariant parts of the fragment addresses; the tap offset comes from the kernarg tap list (scalar loads
          // the compiler can issue several taps ahead) - the LDS tap table cost a dependent ds_read per tap, which with one
          // wave per SIMD was ~250 cycles per tap for 64 cycles of MFMA
          int xrow[PT];
#pragma unroll
          for (int b = 0; b < PT; ++b) xrow[b] = ((wp * PT + b) * cl.iw + idx) * p.S * kRowBytes + g * 16;
          const int wrow = (wc * CT * 16 + idx) * kRowBytes + g * 16;
          for (int cc = 0; cc < cn; ++cc) {
            const char* la = lds_a + cc * a_stride;
  #pragma unroll 4
            for (int tt = 0; tt < tn; ++tt) {
              const int toff = ((cl.dy[t0 + tt] - cl.dymin) * cl.iw + (cl.dx[t0 + tt] - cl.dxmin)) * kRowBytes;
              const char* lw = lds_w + (cc * p.tg + tt) * CO_TILE * kRowBytes + wrow;
              Frag wf[CT];
  #pragma unroll
              for (int a = 0; a < CT; ++a) wf[a] = *reinterpret_cast<const Frag*>(lw + a * 16 * kRowBytes);
  #pragma unroll
              for (int b = 0; b < PT; ++b) {
                const Frag xf = *reinterpret_cast<const Frag*>(la + xrow[b] + toff);
  #pragma unroll
                for (int a = 0; a < CT; ++a) acc[a][b] = Mma<T>::run(wf[a], xf, acc[a][b]);
              }
            }
          }
        }
      }
    }

  }

  // ---------------------------------------------------------------- epilogue
  TG_STAMP_AT(5);
  const int q = g;  // accumulator rows 4q..4q+3 of each 16-row tile live in this lane
  float s1[NG][E], s2[NG][E];
#pragma unroll
  for (int a = 0; a < NG; ++a)
#pragma unroll
    for (int e = 0; e < E; ++e) s1[a][e] = s2[a][e] = 0.f;

  // The mask vectors of the epilogue (a launch without a mask: its residual vectors), ALL fetched up front from clamped
  // addresses: a load under the divergent `valid` test below makes the compiler wait for each one before it issues the next -
  // PT * NG dependent round trips to cold data at the end of every workgroup
  const char* pre_src = p.mask_mode != TG_MASK_NONE ? p.mask : p.res;
  u32x4 pre[PT][NG];
  if (pre_src) {  // uniform
#pragma unroll
    for (int b = 0; b < PT; ++b) {
      const int cy = min(ty0 + wp * PT + b, OHc - 1), cx = min(tx0 + idx, OWc - 1);
      const size_t pix = ((size_t)n * p.OH + cy * p.OS + cl.ooy) * p.OW + cx * p.OS + cl.oox;
#pragma unroll
      for (int a = 0; a < NG; ++a) {
        const int ch0 = (TR::kBytes == 2) ? co_base + (wc * CT + 2 * a) * 16 + 8 * q : co_base + (wc * CT + a) * 16 + 4 * q;
        pre[b][a] = *reinterpret_cast<const u32x4*>(pre_src + (pix * p.Cout + ch0) * TR::kBytes);
      }
    }
  }

#pragma unroll
  for (int b = 0; b < PT; ++b) {
    const int cy = ty0 + wp * PT + b, cx = tx0 + idx;
    const bool valid = (cy < OHc) && (cx < OWc);
    const int oy = cy * p.OS + cl.ooy, ox = cx * p.OS + cl.oox;
    const size_t pix = ((size_t)n * p.OH + oy) * p.OW + ox;
#pragma unroll
    for (int a = 0; a < NG; ++a) {
      float v[E];
      int ch0;
      if constexpr (TR::kBytes == 2) {
        ch0 = co_base + (wc * CT + 2 * a) * 16 + 8 * q;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v[j] = acc[2 * a][b][j];
          v[4 + j] = acc[2 * a + 1][b][j];
        }
      } else {
        ch0 = co_base + (wc * CT + a) * 16 + 4 * q;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = acc[a][b][j];
      }
      if (valid) {
#pragma unroll
        for (int e = 0; e < E; ++e) v[e] += bias_r[a][e];
        const size_t eoff = (pix * p.Cout + ch0) * TR::kBytes;
        if (p.res) {
          float r[E];
          if (p.mask_mode == TG_MASK_NONE) Vec<T>::load(&pre[b][a], r);
          else Vec<T>::load(p.res + eoff, r);
#pragma unroll
          for (int e = 0; e < E; ++e) v[e] += r[e];
        }
        if constexpr (SLIM) {
          if (p.act != TG_ACT_NONE) {
            const float neg = p.act == TG_ACT_LRELU ? 0.2f : 0.f;
#pragma unroll
            for (int e = 0; e < E; ++e) v[e] = v[e] > 0.f ? v[e] : neg * v[e];
          }
        } else if (p.act != TG_ACT_NONE) {
#pragma unroll
          for (int e = 0; e < E; ++e) v[e] = apply_act(v[e], p.act);
        }
        if (p.mask_mode == TG_MASK_RELU || p.mask_mode == TG_MASK_LRELU) {
          float m[E];
          Vec<T>::load(&pre[b][a], m);
          const float neg = p.mask_mode == TG_MASK_LRELU ? 0.2f : 0.f;
#pragma unroll
          for (int e = 0; e < E; ++e) v[e] *= (m[e] > 0.f ? 1.f : neg);
        }
        if (SLIM || p.out_mode == TG_OUT_NHWC) {
          Vec<T>::store(p.out + eoff, v);
        } else if (ch0 == 0) {
          float* o = reinterpret_cast<float*>(p.out) + (size_t)n * p.out_n_stride + (size_t)oy * p.OW + ox;
          for (int e = 0; e < p.c_real; ++e) o[(size_t)e * p.OH * p.OW] = v[e];
        }
        if (kTgExperiments && !SLIM && p.stats_mode == 3) {
          // batch-norm backward sums for the layer this output is the gradient of (TG_MASK_BNZ: `mask` is that layer's
          // pre-normalisation tensor z): sum dy and sum dy * z per channel, of the values as STORED (what tg_bn_bwd_reduce,
          // which this replaces, would read back)
          float zz[E], vr[E];
          Vec<T>::load(&pre[b][a], zz);
          u32x4 rt;
          Vec<T>::pack(&rt, v);
          Vec<T>::load(&rt, vr);
#pragma unroll
          for (int e = 0; e < E; ++e) {
            s1[a][e] += vr[e];
            s2[a][e] += vr[e] * zz[e];
          }
        } else if (!SLIM && p.stats_mode) {
#pragma unroll
          for (int e = 0; e < E; ++e) {
            s1[a][e] += v[e];
            s2[a][e] += v[e] * v[e];
          }
        }
      }
    }
  }

  TG_STAMP_AT(6);
  if (!SLIM && p.stats_mode) {  // uniform branch
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
        const int cl0 = (TR::kBytes == 2) ? (wc * CT + 2 * a) * 16 + 8 * q : (wc * CT + a) * 16 + 4 * q;
#pragma unroll
        for (int e = 0; e < E; ++e) {
          red[(wp * 2 + 0) * CO_TILE + cl0 + e] = s1[a][e];
          red[(wp * 2 + 1) * CO_TILE + cl0 + e] = s2[a][e];
        }
      }
    }
    __syncthreads();
    const int grp = n / (p.N / p.stats_groups);
    for (int i = tid; i < (p.stats_mode == 1 ? 1 : 2) * CO_TILE; i += NTHR) {
      const int which = i / CO_TILE, chn = i - which * CO_TILE;
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < WP; ++w) s += red[(w * 2 + which) * CO_TILE + chn];
      // replica = low bits of the pixel-tile index: thousands of workgroups adding to the same words serialise at the
      // memory side (a 5120-workgroup dgrad spent 240 us on this); tg_reduce_replicas folds the replicas afterwards
      const size_t rep = (size_t)(blockIdx.x & (p.stats_replicas - 1)) * p.stats_groups * 2 * p.Cout;
      atomicAdd(p.stats + rep + ((size_t)grp * 2 + which) * p.Cout + co_base + chn, s);
    }
  }
}

// ------------------------------------------------------------------------------------------------ packing
template <typename T>
__global__ void pack_weights_kernel(const float* __restrict__ w, char* __restrict__ packed, int cout, int cin,
                                    int cout_p, int cin_p, long long s_co, long long s_ci, int nslots,
                                    const int* __restrict__ slot_off) {
  using TR = ElemTraits<T>;
  const int nchunks = cin_p / TR::kChunk;
  const long long total = (long long)nslots * nchunks * cout_p * TR::kChunk;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int kc = (int)(i % TR::kChunk);
    long long r = i / TR::kChunk;
    const int row = (int)(r % cout_p);
    r /= cout_p;
    const int c = (int)(r % nchunks);
    const int slot = (int)(r / nchunks);
    const int co = row_to_channel<T>(row);
    const int ci = c * TR::kChunk + kc;
    float v = 0.f;
    if (co < cout && ci < cin) v = w[co * s_co + ci * s_ci + slot_off[slot]];
    store_elem<T>(packed, i, v);
  }
}

struct TileCfg {
  int co_tile, th;
};

template <typename T, int CT, int PT, int WC, int WP, bool STD3, bool SLIM = false>
int launch_conv_impl(const ConvK& k, dim3 grid, size_t lds, hipStream_t st) {
  auto fn = conv_gather_kernel<T, CT, PT, WC, WP, STD3, SLIM>;
  static std::atomic<bool> attr_done{false};  // one-time function attribute (benign race: idempotent)
  if (!attr_done) {
    TG_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     160 * 1024));
    attr_done = true;
  }
  hipLaunchKernelGGL(fn, grid, dim3(64 * WC * WP), lds, st, k);
  return tg_launch_status();
}

template <typename T, int CT, int PT, int WC, int WP>
int launch_conv(const ConvK& k, dim3 grid, size_t lds, hipStream_t st) {
  if (k.std3) return k.slim ? launch_conv_impl<T, CT, PT, WC, WP, true, true>(k, grid, lds, st)
                            : launch_conv_impl<T, CT, PT, WC, WP, true, false>(k, grid, lds, st);
  return launch_conv_impl<T, CT, PT, WC, WP, false>(k, grid, lds, st);
}

template <typename T>
int dispatch_conv(int cfg, const ConvK& k, dim3 grid, size_t lds, hipStream_t st) {
  switch (cfg) {
    case TG_TILE_64x256: return launch_conv<T, 4, 4, 1, 4>(k, grid, lds, st);
    case TG_TILE_64x64: return launch_conv<T, 2, 2, 2, 2>(k, grid, lds, st);
    case TG_TILE_128x128: return launch_conv<T, 4, 4, 2, 2>(k, grid, lds, st);
    case TG_TILE_32x128: return launch_conv<T, 2, 2, 1, 4>(k, grid, lds, st);
    case TG_TILE_32x64: return launch_conv<T, 2, 1, 1, 4>(k, grid, lds, st);
    case TG_TILE_64x128: return launch_conv<T, 4, 2, 1, 4>(k, grid, lds, st);
    case TG_TILE_64x128_8W:  // pipelined 3x3 path only (prepare_conv falls back to 64x128 otherwise)
      if (!k.std3) return TG_E_UNSUPPORTED;
      return k.slim ? launch_conv_impl<T, 2, 2, 2, 4, true, true>(k, grid, lds, st)
                    : launch_conv_impl<T, 2, 2, 2, 4, true, false>(k, grid, lds, st);
    case TG_TILE_64x64_8W:
      if (!k.std3) return TG_E_UNSUPPORTED;
      return k.slim ? launch_conv_impl<T, 2, 1, 2, 4, true, true>(k, grid, lds, st)
                    : launch_conv_impl<T, 2, 1, 2, 4, true, false>(k, grid, lds, st);
  }
  return TG_E_UNSUPPORTED;
}

TileCfg tile_cfg(int cfg) {
  switch (cfg) {
    case TG_TILE_64x256: return {64, 16};
    case TG_TILE_64x64: return {64, 4};
    case TG_TILE_128x128: return {128, 8};
    case TG_TILE_32x128: return {32, 8};
    case TG_TILE_32x64: return {32, 4};
    case TG_TILE_64x128: return {64, 8};
    case TG_TILE_64x128_8W: return {64, 8};
    case TG_TILE_64x64_8W: return {64, 4};
  }
  return {0, 0};
}

}  // namespace

extern "C" int64_t tg_packed_weight_bytes(int dtype, int nslots, int cout_p, int cin_p) {
  if (nslots <= 0 || cout_p <= 0 || cin_p <= 0) return TG_E_BADARG;
  return (int64_t)nslots * cout_p * cin_p * (dtype == TG_F32 ? 4 : 2);
}

extern "C" int tg_pack_conv_weights(int dtype, const float* w, void* packed, int cout, int cin, int cout_p, int cin_p,
                                    int64_t s_co, int64_t s_ci, int nslots, const int32_t* slot_off_dev,
                                    void* stream) {
  if (!w || !packed || !slot_off_dev || cout <= 0 || cin <= 0 || nslots <= 0) return TG_E_BADARG;
  if (cout_p % 32 || cin_p % 32 || cout > cout_p || cin > cin_p) return TG_E_ALIGN;
  const long long total = (long long)nslots * cout_p * cin_p;
  const int blocks = (int)std::min<long long>((total + 255) / 256, 2048);
  hipStream_t st = (hipStream_t)stream;
#define TG_PACK(TAG)                                                                                                  \
  hipLaunchKernelGGL(pack_weights_kernel<TAG>, dim3(blocks), dim3(256), 0, st, w, (char*)packed, cout, cin, cout_p, \
                     cin_p, (long long)s_co, (long long)s_ci, nslots, slot_off_dev)
  TG_DISPATCH_DTYPE(dtype, TG_PACK(BF16), TG_PACK(F16), TG_PACK(F32));
#undef TG_PACK
  return tg_launch_status();
}

// tile configuration of a launch (TG_TILE_AUTO resolved from the per-layer measurements)
static int pick_tile(const tg_conv_desc* d) {
  int cfg = d->tile_cfg;
  if (cfg == TG_TILE_AUTO) {
    long long px = 0;
    for (int c = 0; c < d->ncls; ++c) {
      const long long ohc = (d->OH - d->cls[c].ooy + d->OS - 1) / d->OS, owc = (d->OW - d->cls[c].oox + d->OS - 1) / d->OS;
      px += (long long)d->N * ohc * owc;
    }
    // measured per layer shape under hipGraph replay (tools/microbench.py, profiles/r01_*_microbench*.log)
    const bool plain3x3 = d->ncls == 1 && d->S == 1 && d->OS == 1 && d->cls[0].ntaps == 9;
    if (d->Cout % 64) cfg = TG_TILE_32x128;
    else if (d->S > 1) cfg = (d->Cout % 128 == 0 && px >= 16384) ? TG_TILE_128x128 : (px >= 32768 ? TG_TILE_64x128 : TG_TILE_64x64);
    else if (plain3x3 && px < 16384) cfg = TG_TILE_32x64;                    // recurrent-pass and deep-D layers
    else if (plain3x3 && px <= 16384 && d->Cin <= 64 && d->Cout <= 64) cfg = TG_TILE_32x64;  // 7.0 vs 7.6 us (64x64)
    else if (plain3x3 && (px > 262144 || (d->Cin >= 128 && d->Cout >= 128 && px > 65536)))
      cfg = TG_TILE_64x256;                                                  // 150 vs 214 us (c6 dgrad), 99 vs 105 us
    else if (plain3x3) {
      cfg = TG_TILE_64x128;                                                  // tools/microbench.py tiles: 10-30 % faster
      // <= 256 workgroups: every CU gets at most one, so take two waves per SIMD from the workgroup itself
      // (tools/microbench.py w8: 9.8 -> 8.6, 13.5 -> 11.6 us at 256 workgroups; slower on larger grids, which co-schedule)
      const long long wgs = (long long)d->N * ((d->OH + 7) / 8) * ((d->OW + 15) / 16) * (d->Cout / 64);
      if (wgs <= 256) cfg = TG_TILE_64x128_8W;
    }
    else if (d->ncls == 4 && d->Cout % 128 == 0 && px >= 16384) cfg = TG_TILE_64x128;  // conv-transpose forward: 22.7 vs 27.2 us
    else if (d->Cout % 128 == 0 && px >= 16384) cfg = TG_TILE_128x128;
    else if (px >= 32768) cfg = TG_TILE_64x256;
    else cfg = TG_TILE_64x64;
  }
  return cfg;
}

// validates the descriptor and derives the launch plan (tile config, LDS split, grid); shared by tg_conv and tg_conv_pick_tile
static int prepare_conv(const tg_conv_desc* d, const void* in, const void* w_packed, const float* bias, const void* res,
                        const void* mask, void* out, float* stats, bool check_ptrs, ConvK& k, dim3& grid, size_t& lds_out,
                        int& cfg_out) {
  if (!d) return TG_E_BADARG;
  if (check_ptrs && (!in || !w_packed || !out)) return TG_E_BADARG;
  if (d->dtype != TG_F32 && d->dtype != TG_BF16 && d->dtype != TG_F16) return TG_E_BADARG;
  if (d->N <= 0 || d->IH <= 0 || d->IW <= 0 || d->OH <= 0 || d->OW <= 0 || d->S <= 0 || d->OS <= 0) return TG_E_BADARG;
  if (d->ncls <= 0 || d->ncls > TG_MAX_CLASSES) return TG_E_BADARG;
  if (d->Cin <= 0 || d->Cout <= 0 || d->Cin % 32 || d->Cout % 32) return TG_E_ALIGN;
  if (check_ptrs) {
    if (!tg_aligned16(in) || !tg_aligned16(w_packed) || !tg_aligned16(out) || (res && !tg_aligned16(res)) ||
        (mask && !tg_aligned16(mask)))
      return TG_E_ALIGN;
    if (d->mask_mode != TG_MASK_NONE && !mask) return TG_E_BADARG;
    if (d->stats_mode && !stats) return TG_E_BADARG;
  }
  if (d->stats_mode < 0 || d->stats_mode > 3) return TG_E_BADARG;
  if (!kTgExperiments && (d->stats_mode == 3 || d->mask_mode == TG_MASK_BNZ)) return TG_E_UNSUPPORTED;  // experiments build only
  if ((d->stats_mode == 3) != (d->mask_mode == TG_MASK_BNZ)) return TG_E_BADARG;  // the BN sums need z in the mask slot
  if (d->mask_mode == TG_MASK_BNZ && !mask && check_ptrs) return TG_E_BADARG;
  if (d->stats_mode && (d->stats_groups <= 0 || d->N % d->stats_groups)) return TG_E_BADARG;
  if (d->out_mode == TG_OUT_NCHW_F32 && (d->c_real <= 0 || d->c_real > 4 || d->out_n_stride <= 0)) return TG_E_BADARG;
  if (d->out_mode != TG_OUT_NHWC && d->out_mode != TG_OUT_NCHW_F32) return TG_E_BADARG;
  if (d->out_mode == TG_OUT_NCHW_F32 && (res || d->mask_mode)) return TG_E_UNSUPPORTED;

  int cfg = pick_tile(d);
  const TileCfg tc = tile_cfg(cfg);
  if (!tc.co_tile || d->Cout % tc.co_tile) return TG_E_UNSUPPORTED;

  k.in = (const char*)in; k.w = (const char*)w_packed; k.bias = bias; k.res = (const char*)res;
  k.mask = (const char*)mask; k.out = (char*)out; k.stats = stats;
  k.N = d->N; k.IH = d->IH; k.IW = d->IW; k.Cin = d->Cin; k.OH = d->OH; k.OW = d->OW; k.Cout = d->Cout;
  k.S = d->S; k.OS = d->OS;
  k.act = d->act; k.mask_mode = d->mask_mode; k.stats_mode = d->stats_mode; k.stats_groups = d->stats_groups;
  k.out_mode = d->out_mode; k.c_real = d->c_real; k.out_n_stride = d->out_n_stride;
  k.stats_replicas = d->stats_replicas > 1 ? d->stats_replicas : 1;
  if (k.stats_replicas & (k.stats_replicas - 1)) return TG_E_BADARG;  // power of two
  const int chunk = d->dtype == TG_F32 ? 16 : 32;
  k.nchunks = d->Cin / chunk;

  int max_ohc = 0, max_owc = 0, max_rows = 0, max_taps = 0;
  for (int c = 0; c < d->ncls; ++c) {
    const tg_conv_class& s = d->cls[c];
    if (s.ntaps <= 0 || s.ntaps > TG_MAX_TAPS || s.ooy < 0 || s.oox < 0 || s.ooy >= d->OS || s.oox >= d->OS)
      return TG_E_BADARG;
    ConvClassK& o = k.cls[c];
    o.ooy = s.ooy; o.oox = s.oox; o.ntaps = s.ntaps;
    int dymin = 127, dymax = -128, dxmin = 127, dxmax = -128;
    for (int t = 0; t < s.ntaps; ++t) {
      o.dy[t] = s.dy[t]; o.dx[t] = s.dx[t]; o.widx[t] = s.widx[t];
      if (s.widx[t] < 0) return TG_E_BADARG;
      dymin = std::min<int>(dymin, s.dy[t]); dymax = std::max<int>(dymax, s.dy[t]);
      dxmin = std::min<int>(dxmin, s.dx[t]); dxmax = std::max<int>(dxmax, s.dx[t]);
    }
    o.dymin = dymin; o.dxmin = dxmin;
    o.ih = (tc.th - 1) * d->S + (dymax - dymin) + 1;
    o.iw = 15 * d->S + (dxmax - dxmin) + 1;
    max_rows = std::max(max_rows, o.ih * o.iw);
    max_taps = std::max(max_taps, s.ntaps);
    max_ohc = std::max(max_ohc, (d->OH - s.ooy + d->OS - 1) / d->OS);
    max_owc = std::max(max_owc, (d->OW - s.oox + d->OS - 1) / d->OS);
  }
  k.a_rows_max = max_rows;
  size_t a_bytes = (size_t)max_rows * kRowBytes;
  size_t w_tap = (size_t)tc.co_tile * kRowBytes;
  k.tiles_x = (max_owc + 15) / 16;
  k.tiles_y = (max_ohc + tc.th - 1) / tc.th;
  auto ilog2 = [](int v) { int l = 0; while ((1 << l) < v) ++l; return (1 << l) == v ? l : -1; };
  k.tx_log2 = ilog2(k.tiles_x);
  k.ty_log2 = ilog2(k.tiles_y);
  if (d->OS > 2) return TG_E_UNSUPPORTED;
  const long long gx = (long long)k.tiles_x * k.tiles_y * d->N;
  if (gx > 0x7fffffffLL) return TG_E_UNSUPPORTED;
  const long long wgs = gx * (d->Cout / tc.co_tile) * d->ncls;
  // LDS budget: a grid that cannot give every CU two workgroups anyway may use (almost) the whole 160 KB so that all
  // of K is staged in one or two stages; big grids keep two workgroups per CU for cross-workgroup latency hiding.
  // plain 3x3 stride-1 pattern (forward: dy=t/3-1, dx=t%3-1; dgrad: mirrored), slots in tap order -> pipelined STD3 kernel
  bool pat_fwd = false, pat_bwd = false;
  if (d->ncls == 1 && d->S == 1 && d->OS == 1 && d->cls[0].ntaps == 9) {
    pat_fwd = pat_bwd = true;
    for (int t = 0; t < 9; ++t) {
      const tg_conv_class& s0 = d->cls[0];
      if (s0.widx[t] != t) pat_fwd = pat_bwd = false;
      if (s0.dy[t] != t / 3 - 1 || s0.dx[t] != t % 3 - 1) pat_fwd = false;
      if (s0.dy[t] != 1 - t / 3 || s0.dx[t] != 1 - t % 3) pat_bwd = false;
    }
  }
  // measured (profiles/r01_*_microbench*.log): the pipelined kernel wins up to ~1000 workgroups (latency-bound launches);
  // beyond that two co-resident workgroups of the plain kernel hide latency better than one pipelined workgroup per CU
  const bool pattern = (pat_fwd || pat_bwd) && (wgs <= 1024 || (k.nchunks >= 4 && wgs <= 2048));
  const bool std3_ok = pattern && max_taps == 9 && max_rows * 4 <= 256 * 6;
  if (std3_ok) {  // the pipelined 3x3 kernel keeps its LDS images in swizzled 64-byte rows, patch pitch kSwzPitch
    a_bytes = (size_t)k.cls[0].ih * kSwzPitch * kSwzRow;
    w_tap = (size_t)tc.co_tile * kSwzRow;
  }
  // LDS budget: small grids (<= 2 workgroups per CU anyway) and the software-pipelined 3x3 kernel (it hides its own load
  // latency) may use almost all 160 KB; the generic kernel on big grids keeps two workgroups per CU.
  const size_t budget = (wgs <= 512 || pattern) ? 150 * 1024 : 72 * 1024;
  int tg = 0, cg = 0;
  auto plan = [&]() {
    tg = max_taps;
    while (tg > 1 && a_bytes + tg * w_tap > budget) --tg;
    cg = 1;
    if (tg == max_taps)
      while (cg < k.nchunks && (size_t)(cg + 1) * (a_bytes + tg * w_tap) <= budget) ++cg;
  };
  plan();
  if (std3_ok && tg != 9) {  // cannot happen with the tile configurations above; stay correct if one is added
    a_bytes = (size_t)max_rows * kRowBytes;
    w_tap = (size_t)tc.co_tile * kRowBytes;
    plan();
  }
  k.std3 = 0;
  k.flip = 0;
#ifdef TG_NO_SLIM  // A/B builds (tools): always the general epilogue
  k.slim = 0;
#else
  k.slim = (d->out_mode == TG_OUT_NHWC && d->stats_mode == 0 &&
            (d->act == TG_ACT_NONE || d->act == TG_ACT_RELU || d->act == TG_ACT_LRELU)) ? 1 : 0;
#endif
  if (std3_ok && tg == 9) {
    k.std3 = 1;
    k.a_rows_max = k.cls[0].ih * kSwzPitch;
    k.flip = (!pat_fwd && pat_bwd) ? 1 : 0;
    // one chunk group must fit one issue phase of the pipelined kernel: UA = 6 pieces, UW = 18 / 9 weight blocks per thread
    const bool small_cfg = (cfg == TG_TILE_64x64 || cfg == TG_TILE_32x128 || cfg == TG_TILE_32x64 || cfg == TG_TILE_64x128_8W ||
                            cfg == TG_TILE_64x64_8W);
    const int uw = small_cfg ? 18 : 9;
    while (cg > 1 && (cg * max_rows * 4 > 256 * 6 || cg * 9 > uw)) --cg;
  }
  if (cfg == TG_TILE_64x128_8W && !k.std3) cfg = TG_TILE_64x128;  // same tile geometry, 4 waves, generic path
  if (cfg == TG_TILE_64x64_8W && !k.std3) cfg = TG_TILE_64x64;
  size_t lds = (size_t)cg * (a_bytes + tg * w_tap);
  if (lds > 160 * 1024) return TG_E_UNSUPPORTED;
  lds = std::max<size_t>(lds, 4 * 2 * tc.co_tile * sizeof(float));  // stats scratch
  k.tg = tg;
  k.cg = cg;
  grid = dim3((unsigned)gx, (unsigned)(d->Cout / tc.co_tile), (unsigned)d->ncls);
  lds_out = lds;
  cfg_out = cfg;
  return TG_OK;
}

extern "C" int tg_conv_pick_tile(const tg_conv_desc* d) {
  ConvK k;
  dim3 grid;
  size_t lds;
  int cfg;
  const int rc = prepare_conv(d, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, false, k, grid, lds, cfg);
  return rc != TG_OK ? rc : (cfg | (k.std3 << 8));
}

extern "C" int tg_conv(const tg_conv_desc* d, const void* in, const void* w_packed, const float* bias, const void* res,
                       const void* mask, void* out, float* stats, void* stream) {
  ConvK k;
  dim3 grid;
  size_t lds;
  int cfg;
  const int rc = prepare_conv(d, in, w_packed, bias, res, mask, out, stats, true, k, grid, lds, cfg);
  if (rc != TG_OK) return rc;
  hipStream_t st = (hipStream_t)stream;
  if (d->dtype == TG_BF16) return dispatch_conv<BF16>(cfg, k, grid, lds, st);
  if (d->dtype == TG_F16) return dispatch_conv<F16>(cfg, k, grid, lds, st);
  return dispatch_conv<F32>(cfg, k, grid, lds, st);
}

// ---------------------------------------------------------------------------------------- table-driven repack
// One launch repacks every conv of a network (forward and dgrad copies): blockIdx.y selects the job.
// Job = 9 x int64: w ptr, packed ptr, s_row, s_k, rows, K, rows_p, K_p, nslots  (slot t reads kernel offset t).
namespace {
template <typename T>
__global__ void pack_multi_kernel(const long long* __restrict__ jobs) {
  using TR = ElemTraits<T>;
  const long long* j = jobs + 9 * blockIdx.y;
  const float* w = reinterpret_cast<const float*>(j[0]);
  char* packed = reinterpret_cast<char*>(j[1]);
  const long long s_co = j[2], s_ci = j[3];
  const int cout = (int)j[4], cin = (int)j[5], cout_p = (int)j[6], cin_p = (int)j[7], nslots = (int)j[8];
  const int nchunks = cin_p / TR::kChunk;
  // one K-chunk row (kChunk elements) per group of kChunk threads; the (slot, chunk, row) split of the row index uses float
  // reciprocals - the first version divided 64-bit element indices four times per element (no hardware divide): 58 us per step
  constexpr int KC = TR::kChunk, RPB = 256 / KC;
  const int nrows = nslots * nchunks * cout_p;
  const float inv_cp = 1.0f / (float)cout_p, inv_nc = 1.0f / (float)nchunks;
  const int kc = threadIdx.x % KC;
  for (int o = blockIdx.x * RPB + threadIdx.x / KC; o < nrows; o += gridDim.x * RPB) {
    const int r = (int)(((float)o + 0.5f) * inv_cp), row = o - r * cout_p;
    const int slot = (int)(((float)r + 0.5f) * inv_nc), c = r - slot * nchunks;
    const long long i = (long long)o * KC + kc;
    const int co = row_to_channel<T>(row);
    const int ci = c * TR::kChunk + kc;
    float v = 0.f;
    if (co < cout && ci < cin) v = w[co * s_co + ci * s_ci + slot];
    store_elem<T>(packed, i, v);
  }
}
}  // namespace

extern "C" int tg_pack_conv_weights_multi(int dtype, const int64_t* jobs_dev, int njobs, int blocks_per_job,
                                          void* stream) {
  if (!jobs_dev || njobs <= 0 || blocks_per_job <= 0) return TG_E_BADARG;
  dim3 grid((unsigned)blocks_per_job, (unsigned)njobs);
#define TG_PACKM(TAG) \
  hipLaunchKernelGGL(pack_multi_kernel<TAG>, grid, dim3(256), 0, (hipStream_t)stream, (const long long*)jobs_dev)
  TG_DISPATCH_DTYPE(dtype, TG_PACKM(BF16), TG_PACKM(F16), TG_PACKM(F32));
#undef TG_PACKM
  return tg_launch_status();
}
