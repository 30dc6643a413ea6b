// Weight gradient of a gather convolution on the matrix cores:
//
//   dW[t][a][b] = sum over (n, y, x) of  X[n, y*S + dy[t], x*S + dx[t]][a] * Y[n, y, x][b]
//
// The reduction dimension is the PIXEL index, but both operands are NHWC (channel-contiguous), so each MFMA
// operand is a transposed read of a [pixel][channel] LDS image: ds_read_b64_tr_b16 for bf16 (4 pixels x 16
// channels per 16-lane group, row addresses supplied per lane, which also makes tap shifts and stride-2
// gathers free), plain ds_read_b32 for f32 (v_mfma_f32_16x16x4_f32 takes one value per lane).
//
// One workgroup owns a (16*AW)-channel slice of X times a (16*BT*BW)-channel slice of Y for ALL taps and
// walks its share of the pixel tiles (split-K over pixels) with the partial dW kept in accumulators; it
// writes one fp32 slab, and tg_wgrad_finalize sums the slabs into the PyTorch-layout gradient.
//
// Replaces aten::convolution_backward (weight path) behind code/train.py:336,340.
#include "common.h"
#include <cstdlib>

#ifdef TG_STAMP
// Diagnostic build only (build.sh -DTG_STAMP): wave 0 of workgroup 0 accumulates s_memtime differences per phase of the tile
// loop into a buffer nothing else reads (slot 0 barrier wait, 1 LDS stores, 2 second barrier, 3 issue of the next tile's
// loads, 4 k-loop, 5 tiles); the production library contains no stamp.  tools/stamp_wgrad.py prints them.
__device__ long long tg_wg_stamps[8];
#define TG_WG_MARK(i)                                                                          \
  do {                                                                                         \
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) {           \
      const long long _n = (long long)__builtin_amdgcn_s_memtime();                            \
      tg_wg_stamps[i] += _n - _wg_t;                                                           \
      _wg_t = _n;                                                                              \
    }                                                                                          \
  } while (0)
extern "C" int tg_debug_wgrad_stamps(long long* out, int reset) {
  long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (reset) return (int)hipMemcpyToSymbol(HIP_SYMBOL(tg_wg_stamps), z, sizeof(z));
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(tg_wg_stamps), sizeof(z));
}
#else
#define TG_WG_MARK(i) do {} while (0)
#endif

namespace {

struct WgradK {
  const char* x;
  const char* y;
  float* slab;
  const long long* jobs;  // non-null: blockIdx.x = job * nsplit + split, and x / y / slab come from jobs[3*job ..]
  int N, XH, XW, Cx, YH, YW, Cy, S;
  int ntaps;
  int dy[TG_MAX_TAPS];  // 32-bit: dynamically indexed kernarg bytes would become vector loads (see conv_mfma.hip)
  int dx[TG_MAX_TAPS];
  int dymin, dxmin, ih, iw;
  int tw_log2, th;
  int tiles_x, tiles_y, tiles_total, nsplit;
  int b_blocks;
  int ysum;  // 1: also write sum over pixels of Y (the conv's bias gradient) behind the taps of every slab
};

template <typename T> __device__ __forceinline__ void unpack16(const u32x4& v, float* f);
template <> __device__ __forceinline__ void unpack16<BF16>(const u32x4& v, float* f) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[2 * i] = __uint_as_float(v[i] << 16);
    f[2 * i + 1] = __uint_as_float(v[i] & 0xffff0000u);
  }
}
template <> __device__ __forceinline__ void unpack16<F16>(const u32x4& v, float* f) {
  const f16x8 h = __builtin_bit_cast(f16x8, v);
#pragma unroll
  for (int i = 0; i < 8; ++i) f[i] = (float)h[i];
}
template <> __device__ __forceinline__ void unpack16<F32>(const u32x4& v, float* f) {
#pragma unroll
  for (int i = 0; i < 4; ++i) f[i] = __uint_as_float(v[i]);
}

// TPW = taps per workgroup (blockIdx.z selects the tap group).  With TPW == NTAPS a workgroup owns every tap (best when it
// walks many pixel tiles); TPW = 3 (4 for 4x4) cuts the fp32 slab each workgroup writes - and the fold reads back - by
// NTAPS/TPW for the same number of workgroups: small layers were bound by exactly that traffic (256 x 147 KB per layer).
// NW = waves per workgroup.  8 (two per SIMD): each wave holds half the accumulators (72 for 9 taps), so the tile-staging
// loads of one wave (~120 cycles of issue each) overlap its partner's MFMAs instead of serialising with them.
template <typename T, int NTAPS, int TPW, int AW, int BT, int NW = 4>
__global__ __launch_bounds__(64 * NW) void wgrad_kernel(const WgradK p) {
  using TR = ElemTraits<T>;
  constexpr int NTHR = 64 * NW;
  constexpr int BW = NW / AW;
  constexpr int A_BLK = 16 * AW, B_BLK = 16 * BT * BW;
  // bf16 rows are padded to an ODD number of 32-byte segments (128 + 32, 64 + 32): a 32-lane group of ds_read_b64_tr_b16
  // reads 32 contiguous bytes of each of 8 consecutive pixel rows (k = 4g + q below), and 8 consecutive multiples of an odd
  // segment count are 8 distinct segments of the 256-byte bank window -> conflict-free at any base row / tap shift.
  // (With +16-byte padding and k = 8g + q the rows r and r+8 overlapped by half a segment: 2-way conflicts.)
  constexpr int PADB = (TR::kBytes == 2) ? 32 : 16;
  constexpr int XROW = A_BLK * TR::kBytes + PADB, YROW = B_BLK * TR::kBytes + PADB;
  constexpr int XV = A_BLK * TR::kBytes / 16, YV = B_BLK * TR::kBytes / 16;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tw = 1 << p.tw_log2;
  const int ypix = tw * p.th;
  char* lds_y = smem;
  char* lds_x = smem + (size_t)ypix * YROW;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wa = wid % AW, wb = wid / AW;
  const int idx = lane & 15, g = lane >> 4;
  const int a_blk = blockIdx.y / p.b_blocks, b_blk = blockIdx.y % p.b_blocks;
  const int a0 = a_blk * A_BLK, b0 = b_blk * B_BLK;

  // several same-shaped layers in one grid (tg_wgrad_multi): the layer index is the slow part of blockIdx.x
  int split = blockIdx.x;
  const char* xbase = p.x;
  const char* ybase = p.y;
  float* sbase = p.slab;
  if (p.jobs) {
    const int job = blockIdx.x / p.nsplit;
    split -= job * p.nsplit;
    xbase = reinterpret_cast<const char*>(p.jobs[3 * job]);
    ybase = reinterpret_cast<const char*>(p.jobs[3 * job + 1]);
    sbase = reinterpret_cast<float*>(p.jobs[3 * job + 2]);
  }

  const int tap0 = blockIdx.z * TPW;
  // bias gradient = sum over pixels of Y: the Y tiles pass through this thread's registers anyway, and a thread always
  // holds the same 16-byte channel piece (NTHR % YV == 0), so it keeps E running sums (one X block / tap group does it)
  constexpr int E = 16 / TR::kBytes;
  const bool ysum = p.ysum && a_blk == 0 && blockIdx.z == 0;
  float bsum[E];
#pragma unroll
  for (int e = 0; e < E; ++e) bsum[e] = 0.f;
  f32x4 acc[TPW][BT];
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int b = 0; b < BT; ++b) acc[t][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  const size_t xpix_bytes = (size_t)p.Cx * TR::kBytes, ypix_bytes = (size_t)p.Cy * TR::kBytes;
  const int prow_n = p.ih * p.iw;

  // Software pipeline over this workgroup's pixel tiles: the global loads of tile i+1 are issued (into registers) right
  // after tile i has been written to LDS and stay in flight while the MFMAs of tile i run.  The kernel runs one
  // workgroup per CU (its accumulators fill the register file), so no other workgroup would hide that latency.
  constexpr int UX = ((TR::kBytes == 2) ? 11 : 22) * 4 / NW + (NW == 8 ? 1 : 0);  // 16-byte pieces per thread: X patch up to 10x34 pixels x A_BLK channels
  constexpr int UY = ((TR::kBytes == 2) ? 4 : 8) * 4 / NW;                           //                            Y tile 128 pixels x B_BLK channels
  u32x4 vx[UX], vy[UY];
  const int nx = prow_n * XV, ny = ypix * YV;

  // Per-thread piece constants.  Which patch pixel / channel piece a thread stages does not depend on the tile, so the
  // global offset relative to the patch origin, the LDS offset and the (row, column) for the bounds test are computed ONCE:
  // with the divisions and multiplies inside the per-tile loops, issuing a tile's 8 loads cost ~1700 cycles and the tap
  // address arithmetic doubled the k-loop (in-kernel stamps, tools/stamp_wgrad.py).
  int xg[UX], xl[UX], xrc[UX], yg[UY], yl[UY], yrc[UY];
#pragma unroll
  for (int u = 0; u < UX; ++u) {
    const int i = tid + u * NTHR;
    const int prow = i / XV, s2 = i - prow * XV;  // XV is a power of two
    const int py = prow / p.iw, px = prow - py * p.iw;
    xg[u] = (int)(((size_t)py * p.XW + px) * xpix_bytes) + s2 * 16;
    xl[u] = i < nx ? prow * XROW + s2 * 16 : -1;
    xrc[u] = py | (px << 16);
  }
#pragma unroll
  for (int u = 0; u < UY; ++u) {
    const int i = tid + u * NTHR;
    const int prow = i / YV, s2 = i - prow * YV;
    const int ry = prow >> p.tw_log2, rx = prow & (tw - 1);
    yg[u] = (int)(((size_t)ry * p.YW + rx) * ypix_bytes) + s2 * 16;
    yl[u] = i < ny ? prow * YROW + s2 * 16 : -1;
    yrc[u] = ry | (rx << 16);
  }
  // LDS byte offset of tap t relative to the tap-less row of a pixel (uniform: lives in scalar registers)
  int tapb[TPW];
#pragma unroll
  for (int t = 0; t < TPW; ++t) tapb[t] = ((p.dy[tap0 + t] - p.dymin) * p.iw + (p.dx[tap0 + t] - p.dxmin)) * XROW;

  auto issue = [&](int tile) {
    int r = tile;
    const int txb = r % p.tiles_x;
    r /= p.tiles_x;
    const int tyb = r % p.tiles_y;
    const int n = r / p.tiles_y;
    const int ty0 = tyb * p.th, tx0 = txb * tw;
    const int iy0 = ty0 * p.S + p.dymin, ix0 = tx0 * p.S + p.dxmin;
    // origin pixel of the patch (may lie outside the image: only in-bounds pieces are dereferenced)
    const char* xo = xbase + ((size_t)n * p.XH * p.XW + (long long)iy0 * p.XW + ix0) * (long long)xpix_bytes + (size_t)a0 * TR::kBytes;
#pragma unroll
    for (int u = 0; u < UX; ++u) {
      const int iy = iy0 + (xrc[u] & 0xffff), ix = ix0 + (xrc[u] >> 16);
      vx[u] = u32x4{0u, 0u, 0u, 0u};
      if (xl[u] >= 0 && (unsigned)iy < (unsigned)p.XH && (unsigned)ix < (unsigned)p.XW)
        vx[u] = *reinterpret_cast<const u32x4*>(xo + xg[u]);
    }
    const char* yo = ybase + ((size_t)n * p.YH * p.YW + (size_t)ty0 * p.YW + tx0) * ypix_bytes + (size_t)b0 * TR::kBytes;
#pragma unroll
    for (int u = 0; u < UY; ++u) {
      const int yy = ty0 + (yrc[u] & 0xffff), xx = tx0 + (yrc[u] >> 16);
      vy[u] = u32x4{0u, 0u, 0u, 0u};
      if (yl[u] >= 0 && yy < p.YH && xx < p.YW) vy[u] = *reinterpret_cast<const u32x4*>(yo + yg[u]);
    }
  };

  int tile = split;
  if (tile < p.tiles_total) issue(tile);
#ifdef TG_STAMP
  long long _wg_t = (long long)__builtin_amdgcn_s_memtime();
#endif
  for (; tile < p.tiles_total; tile += p.nsplit) {
    __syncthreads();  // the previous tile's fragment reads are done
    TG_WG_MARK(0);
#pragma unroll
    for (int u = 0; u < UX; ++u)
      if (xl[u] >= 0) *reinterpret_cast<u32x4*>(lds_x + xl[u]) = vx[u];
#pragma unroll
    for (int u = 0; u < UY; ++u) {
      if (yl[u] >= 0) {
        *reinterpret_cast<u32x4*>(lds_y + yl[u]) = vy[u];
        if (ysum) {
          float f[E];
          unpack16<T>(vy[u], f);
#pragma unroll
          for (int e = 0; e < E; ++e) bsum[e] += f[e];
        }
      }
    }
    TG_WG_MARK(1);
    __syncthreads();
    TG_WG_MARK(2);
    if (tile + p.nsplit < p.tiles_total) issue(tile + p.nsplit);  // in flight during the MFMAs below
    TG_WG_MARK(3);

    for (int k0 = 0; k0 < ypix; k0 += 32) {
      if constexpr (TR::kBytes == 2) {
        // lane (q,pp) of each 16-lane group supplies row q, columns 4pp..4pp+3 of a 4-pixel x 16-channel block
        const int q = idx >> 2, pp = idx & 3;
        const int k_lo = k0 + 4 * g + q, k_hi = k_lo + 16;  // any bijection lane -> k works if X and Y use the same one
        bf16x8 bf[BT];
#pragma unroll
        for (int b = 0; b < BT; ++b) {
          const int ch = ((wb * BT + b) * 16 + 4 * pp) * 2;
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (s16x4 __attribute__((address_space(3)))*)(lds_y + k_lo * YROW + ch));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (s16x4 __attribute__((address_space(3)))*)(lds_y + k_hi * YROW + ch));
          typedef __attribute__((ext_vector_type(8))) short s16x8;
          const s16x8 cat = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          bf[b] = __builtin_bit_cast(bf16x8, cat);
        }
        const int ty_lo = k_lo >> p.tw_log2, tx_lo = k_lo & (tw - 1);
        const int ty_hi = k_hi >> p.tw_log2, tx_hi = k_hi & (tw - 1);
        const int cha = (wa * 16 + 4 * pp) * 2;
        const char* x_lo = lds_x + ((ty_lo * p.iw + tx_lo) * p.S) * XROW + cha;  // the pixel's row without a tap offset
        const char* x_hi = lds_x + ((ty_hi * p.iw + tx_hi) * p.S) * XROW + cha;
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (s16x4 __attribute__((address_space(3)))*)(x_lo + tapb[t]));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (s16x4 __attribute__((address_space(3)))*)(x_hi + tapb[t]));
          typedef __attribute__((ext_vector_type(8))) short s16x8;
          const s16x8 cat = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          const bf16x8 af = __builtin_bit_cast(bf16x8, cat);
#pragma unroll
          for (int b = 0; b < BT; ++b) acc[t][b] = Mma16<T>::run(af, bf[b], acc[t][b]);
        }
      } else {
#pragma unroll 2
        for (int sub = 0; sub < 8; ++sub) {
          const int k = k0 + 4 * sub + g;  // lane group g carries pixel k of this 4-pixel MFMA
          float bv[BT];
#pragma unroll
          for (int b = 0; b < BT; ++b)
            bv[b] = *reinterpret_cast<const float*>(lds_y + k * YROW + ((wb * BT + b) * 16 + idx) * 4);
          const int ty = k >> p.tw_log2, tx = k & (tw - 1);
          const char* x_k = lds_x + ((ty * p.iw + tx) * p.S) * XROW + (wa * 16 + idx) * 4;
#pragma unroll
          for (int t = 0; t < TPW; ++t) {
            const float av = *reinterpret_cast<const float*>(x_k + tapb[t]);
#pragma unroll
            for (int b = 0; b < BT; ++b) acc[t][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[b], acc[t][b], 0, 0, 0);
          }
        }
      }
    }
    TG_WG_MARK(4);
#ifdef TG_STAMP
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) tg_wg_stamps[5] += 1;
#endif
  }

  // slab[split][t][a][b] (+ [Cy] channel sums of Y when ysum); accumulator rows 4g+j are the X channel, column idx the Y channel
  const size_t slab_sz = (size_t)NTAPS * p.Cx * p.Cy + (p.ysum ? p.Cy : 0);
  float* slab = sbase + (size_t)split * slab_sz;
  if (ysum) {  // combine the NTHR/YV threads that hold the same channel piece
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);  // [NTHR/YV][YV*E] ; LDS tiles are dead now
    constexpr int CH = YV * E;                    // == B_BLK
    const int s2 = tid % YV, r = tid / YV;
#pragma unroll
    for (int e = 0; e < E; ++e) red[r * CH + s2 * E + e] = bsum[e];
    __syncthreads();
    if (tid < CH) {
      float t = 0.f;
      for (int q = 0; q < NTHR / YV; ++q) t += red[q * CH + tid];
      slab[(size_t)NTAPS * p.Cx * p.Cy + b0 + tid] = t;
    }
  }
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int b = 0; b < BT; ++b) {
      const int bch = b0 + (wb * BT + b) * 16 + idx;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int ach = a0 + wa * 16 + 4 * g + j;
        slab[((size_t)(tap0 + t) * p.Cx + ach) * p.Cy + bch] = acc[t][b][j];
      }
    }
}

// Fold of the per-split slabs into the PyTorch-layout gradient.  The fold is pure HBM streaming (nsplit x the weight
// volume, fp32), so it is laid out for wide contiguous reads: a thread owns 4 consecutive slab elements (one 16-byte
// load per slab), a wavefront therefore reads 1 KiB contiguous per slab, and the 4 wavefronts of a workgroup take every
// 4th slab each with 4 independent loads in flight; the 4 partial sums meet in LDS.  (The first version read 128-byte
// pieces - 32 elements x 8 split lanes - and reached 2.3 TB/s.)
__device__ __forceinline__ void fold_job(const float* __restrict__ slab, float* __restrict__ grad, long long s_a,
                                         long long s_b, int nsplit, int ntaps, int ca_p, int cb_p, int ca, int cb,
                                         const int* __restrict__ slot_off, int accumulate, int blk, int nblk,
                                         f32x4 (*sh)[64], float* __restrict__ bias_grad, size_t slab_sz) {
  const int wunits = ntaps * ca_p * (cb_p / 4);  // float4 units of the weight part of one slab
  const int units = wunits + (bias_grad ? cb_p / 4 : 0);  // + the channel sums of Y (bias gradient), always accumulated
  const int col = threadIdx.x & 63, sl = threadIdx.x >> 6;
  for (int u0 = blk * 64; u0 < units; u0 += nblk * 64) {
    const int u = u0 + col;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    int a = 0, b = 0, t = 0;
    bool live = false;
    const bool is_bias = u >= wunits;
    if (u < units) {
      b = (u % (cb_p / 4)) * 4;
      const int r = u / (cb_p / 4);
      a = r % ca_p;
      t = r / ca_p;
      live = is_bias ? b < cb : (a < ca && b < cb);  // padded rows / columns are never read
    }
    if (live) {
      const float* src = slab + (is_bias ? (size_t)ntaps * ca_p * cb_p + b : ((size_t)t * ca_p + a) * cb_p + b);
      f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = s1, s3 = s1;
      int k = sl;
      for (; k + 12 < nsplit; k += 16) {
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(src + (size_t)k * slab_sz);
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(src + (size_t)(k + 4) * slab_sz);
        const f32x4 v2 = *reinterpret_cast<const f32x4*>(src + (size_t)(k + 8) * slab_sz);
        const f32x4 v3 = *reinterpret_cast<const f32x4*>(src + (size_t)(k + 12) * slab_sz);
        s += v0; s1 += v1; s2 += v2; s3 += v3;
      }
      for (; k < nsplit; k += 4) s += *reinterpret_cast<const f32x4*>(src + (size_t)k * slab_sz);
      s = (s + s1) + (s2 + s3);
    }
    sh[sl][col] = s;
    __syncthreads();
    if (sl == 0 && live) {
      const f32x4 tot = (sh[0][col] + sh[1][col]) + (sh[2][col] + sh[3][col]);
      if (is_bias) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (b + e < cb) bias_grad[b + e] += tot[e];
      } else {
        float* dst = grad + a * s_a + (slot_off ? slot_off[t] : t);
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (b + e < cb) {
            float* d = dst + (b + e) * s_b;
            *d = accumulate ? *d + tot[e] : tot[e];
          }
      }
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void wgrad_finalize_kernel(const float* __restrict__ slab, int nsplit, int ntaps,
                                                            int ca_p, int cb_p, int ca, int cb, float* __restrict__ grad,
                                                            long long s_a, long long s_b,
                                                            const int* __restrict__ slot_off, int accumulate,
                                                            float* __restrict__ bias_grad, long long slab_stride) {
  __shared__ f32x4 sh[4][64];
  fold_job(slab, grad, s_a, s_b, nsplit, ntaps, ca_p, cb_p, ca, cb, slot_off, accumulate, blockIdx.x, gridDim.x, sh,
           bias_grad, (size_t)slab_stride);
}

// All weight gradients of a network folded in ONE launch: blockIdx.y selects the job.
// Job = 12 x int64: slab ptr, grad ptr, s_a, s_b, nsplit, ntaps, ca_p, cb_p, ca, cb, bias-grad ptr or 0, slab stride
// in floats (slot t adds kernel offset t).
//
// Second version.  The first one (fold_job above, still behind tg_wgrad_finalize) gave every thread 4 consecutive b of one
// (t, a) and wrote them straight to grad[b][a][t]: reads were wide, but the WRITES were 4-byte read-modify-writes 2304 bytes
// apart - PMC: 66 MB written + 66 MB re-read per launch for 7-13 MB of gradients, and 2.4 TB/s overall.  Here a workgroup
// owns a tile of AB a-rows x BB b-columns x all taps for a chunk of KC splits: 16-byte loads with NT independent loads in
// flight per split, the tile summed in registers, transposed through LDS, and added to the gradient in runs of AB*NT
// consecutive floats per b (576 bytes for a 3x3 layer).  Split chunks of one tile meet in the gradient through float atomics
// (the buffers are zeroed at the start of the step and every producer accumulates), so the summation order over chunks is
// not fixed: weight gradients are reproducible to rounding, like the bias / BN sums already were.
template <int NT, int BB>
__device__ __forceinline__ void fold2_job(const float* __restrict__ slab, float* __restrict__ grad, long long s_a, long long s_b,
                                          int nsplit, int ca_p, int cb_p, int ca, int cb, float* __restrict__ bias_grad,
                                          size_t slab_sz, float* __restrict__ sh, int u0, int ustep) {
  constexpr int AB = 1024 / BB, NB4 = BB / 4, KC = 8, ROW = AB * NT, PITCH = ROW + 1;
  const int tid = threadIdx.x, ai = tid / NB4, bj = tid % NB4;
  const int tiles_b = cb_p / BB, tiles = (ca_p / AB) * tiles_b;
  const int nchunks = (nsplit + KC - 1) / KC;
  for (int u = u0; u < tiles * nchunks; u += ustep) {
    const int chunk = u % nchunks, tile = u / nchunks;
    const int tb = tile % tiles_b, ta = tile / tiles_b;
    const int a = ta * AB + ai, b = tb * BB + 4 * bj;
    const int k0 = chunk * KC, k1 = min(nsplit, k0 + KC);
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (a < ca && b < cb) {  // padded rows / columns are never read
      const float* src = slab + (size_t)a * cb_p + b;
      for (int k = k0; k < k1; ++k) {
        const float* sk = src + (size_t)k * slab_sz;
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] += *reinterpret_cast<const f32x4*>(sk + (size_t)t * ca_p * cb_p);
      }
    }
    __syncthreads();  // the previous tile's readers are done with the LDS image
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e) sh[(4 * bj + e) * PITCH + ai * NT + t] = acc[t][e];
    __syncthreads();
    for (int i = tid; i < BB * ROW; i += 256) {
      const int lb = i / ROW, r = i - lb * ROW;
      const int la = r / NT, t = r - la * NT;
      const int aa = ta * AB + la, bb = tb * BB + lb;
      if (aa < ca && bb < cb) atomicAdd(grad + aa * s_a + t + bb * s_b, sh[lb * PITCH + r]);
    }
    if (bias_grad && ta == 0 && tid < NB4) {  // channel sums of Y behind the taps of every slab (the conv's bias gradient)
      const int b4 = tb * BB + 4 * tid;
      if (b4 < cb) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int k = k0; k < k1; ++k)
          s += *reinterpret_cast<const f32x4*>(slab + (size_t)k * slab_sz + (size_t)NT * ca_p * cb_p + b4);
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (b4 + e < cb) atomicAdd(bias_grad + b4 + e, s[e]);
      }
    }
  }
}

constexpr int kFold2Lds = (64 * (16 * 16 + 1)) * 4;  // the largest image: 16 taps, BB = 64 -> 65792 bytes

__device__ __forceinline__ void fold2_dispatch(const long long* __restrict__ j, float* fold_sh, int u0, int ustep) {
  const float* slab = reinterpret_cast<const float*>(j[0]);
  float* grad = reinterpret_cast<float*>(j[1]);
  const int nsplit = (int)j[4], ntaps = (int)j[5], ca_p = (int)j[6], cb_p = (int)j[7], ca = (int)j[8], cb = (int)j[9];
  float* bias = reinterpret_cast<float*>(j[10]);
  const size_t sz = (size_t)j[11];
  const bool wide = (cb_p % 64) == 0;
  if (ntaps == 9) {
    if (wide) fold2_job<9, 64>(slab, grad, j[2], j[3], nsplit, ca_p, cb_p, ca, cb, bias, sz, fold_sh, u0, ustep);
    else fold2_job<9, 32>(slab, grad, j[2], j[3], nsplit, ca_p, cb_p, ca, cb, bias, sz, fold_sh, u0, ustep);
  } else if (ntaps == 16) {
    if (wide) fold2_job<16, 64>(slab, grad, j[2], j[3], nsplit, ca_p, cb_p, ca, cb, bias, sz, fold_sh, u0, ustep);
    else fold2_job<16, 32>(slab, grad, j[2], j[3], nsplit, ca_p, cb_p, ca, cb, bias, sz, fold_sh, u0, ustep);
  }
}

__global__ __launch_bounds__(256) void wgrad_finalize_multi_kernel(const long long* __restrict__ jobs) {
  extern __shared__ __attribute__((aligned(16))) float fold_sh[];
  fold2_dispatch(jobs + 12 * blockIdx.y, fold_sh, blockIdx.x, gridDim.x);
}

// The same fold with ONE workgroup per work item (16-row tile x 8-slab chunk) over all jobs: the grid above is
// blocks_per_job x jobs, and a job of a few slabs has 4-8 items - with ~80 jobs per discriminator pass, 4500 of 5000
// workgroups (each reserving the 66-KB transposition image) were dispatched only to exit (103-147 us for 123 MB of slabs).
// Job rows carry a 13th entry here: the job's first item index in the launch.
__global__ __launch_bounds__(256) void wgrad_fold_items_kernel(const long long* __restrict__ jobs, int njobs, int nitems) {
  extern __shared__ __attribute__((aligned(16))) float fold_sh[];
  // (the grid may be smaller than the item count: TECOGAN_FOLD_WGS, a workgroup then walks items blockIdx.x, + gridDim.x, ...)
  for (int it = blockIdx.x; it < nitems; it += gridDim.x) {
    int lo = 0, hi = njobs - 1;  // last job whose first item is <= it
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if ((int)jobs[13 * mid + 12] <= it) lo = mid; else hi = mid - 1;
    }
    fold2_dispatch(jobs + 13 * lo, fold_sh, it - (int)jobs[13 * lo + 12], 1 << 30);
  }
}

struct WgCfg {
  int a_blk, b_blk;
};

template <typename T, int NTAPS, int TPW, int AW, int BT, int NW = 4>
int launch_wgrad(const WgradK& k, dim3 grid, size_t lds, hipStream_t st) {
  auto fn = wgrad_kernel<T, NTAPS, TPW, AW, BT, NW>;
  static std::atomic<bool> attr_done{false};
  if (!attr_done) {
    TG_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     160 * 1024));
    attr_done = true;
  }
  hipLaunchKernelGGL(fn, grid, dim3(64 * NW), lds, st, k);
  return tg_launch_status();
}

int pick_cfg(const tg_wgrad_desc* d, WgCfg* c) {  // returns config id
  if (d->ntaps == 16) {
    if (d->Cx % 64 || d->Cy % 32) return -1;
    *c = {64, 32};
    return 3;
  }
  if (d->ntaps != 9) return -1;
  if (d->Cx % 64) {
    if (d->Cx % 32 || d->Cy % 32) return -1;
    if (d->Cy % 64) {  // 32 x 32 blocks (f_net's 3 -> 32, 32 -> 32 and 32 -> 2 layers, padded to 32 channels)
      *c = {32, 32};
      return 4;
    }
    *c = {32, 64};
    return 1;
  }
  if (d->Cy % 64) {
    if (d->Cy % 32) return -1;
    *c = {64, 32};
    return 2;
  }
  *c = {64, 64};
  return 0;
}

}  // namespace

extern "C" int64_t tg_wgrad_slab_floats(const tg_wgrad_desc* d) {
  if (!d || d->nsplit <= 0 || d->ntaps <= 0) return TG_E_BADARG;
  return (int64_t)d->nsplit * ((int64_t)d->ntaps * d->Cx * d->Cy + (d->y_sum ? d->Cy : 0));
}

namespace {
int wgrad_launch(const tg_wgrad_desc* d, const void* x, const void* y, float* slab, const int64_t* jobs, int njobs,
                 void* stream);
}

extern "C" int tg_wgrad(const tg_wgrad_desc* d, const void* x, const void* y, float* slab, void* stream) {
  if (!d || !x || !y || !slab) return TG_E_BADARG;
  if (!tg_aligned16(x) || !tg_aligned16(y) || !tg_aligned16(slab)) return TG_E_ALIGN;
  return wgrad_launch(d, x, y, slab, nullptr, 1, stream);
}

extern "C" int tg_wgrad_multi(const tg_wgrad_desc* d, const int64_t* jobs_dev, int njobs, void* stream) {
  if (!d || !jobs_dev || njobs <= 0) return TG_E_BADARG;
  if ((long long)njobs * d->nsplit > 65535LL * 16) return TG_E_UNSUPPORTED;
  return wgrad_launch(d, nullptr, nullptr, nullptr, jobs_dev, njobs, stream);
}

namespace {
int wgrad_launch(const tg_wgrad_desc* d, const void* x, const void* y, float* slab, const int64_t* jobs, int njobs,
                 void* stream) {
  if (d->dtype != TG_F32 && d->dtype != TG_BF16 && d->dtype != TG_F16) return TG_E_BADARG;
  if (d->N <= 0 || d->XH <= 0 || d->XW <= 0 || d->YH <= 0 || d->YW <= 0 || d->S <= 0 || d->S > 2 || d->nsplit <= 0)
    return TG_E_BADARG;
  if (d->Cx % 32 || d->Cy % 32 || d->Cx <= 0 || d->Cy <= 0) return TG_E_ALIGN;
  WgCfg c;
  const int cfg = pick_cfg(d, &c);
  if (cfg < 0) return TG_E_UNSUPPORTED;

  WgradK k;
  k.x = (const char*)x; k.y = (const char*)y; k.slab = slab;
  k.jobs = reinterpret_cast<const long long*>(jobs);
  k.N = d->N; k.XH = d->XH; k.XW = d->XW; k.Cx = d->Cx; k.YH = d->YH; k.YW = d->YW; k.Cy = d->Cy; k.S = d->S;
  k.ntaps = d->ntaps;
  int dymin = 127, dymax = -128, dxmin = 127, dxmax = -128;
  for (int t = 0; t < d->ntaps; ++t) {
    k.dy[t] = d->dy[t]; k.dx[t] = d->dx[t];
    dymin = std::min<int>(dymin, d->dy[t]); dymax = std::max<int>(dymax, d->dy[t]);
    dxmin = std::min<int>(dxmin, d->dx[t]); dxmax = std::max<int>(dxmax, d->dx[t]);
  }
  k.dymin = dymin; k.dxmin = dxmin;
  int tw, th;
  if (d->S == 2) { tw = 16; th = 4; }
  else if (d->YW > 16) { tw = 32; th = 4; }
  else { tw = 16; th = 8; }
  k.tw_log2 = tw == 32 ? 5 : 4; k.th = th;
  k.ih = (th - 1) * d->S + (dymax - dymin) + 1;
  k.iw = (tw - 1) * d->S + (dxmax - dxmin) + 1;
  k.tiles_x = (d->YW + tw - 1) / tw;
  k.tiles_y = (d->YH + th - 1) / th;
  const long long tt = (long long)k.tiles_x * k.tiles_y * d->N;
  if (tt > 0x7fffffffLL) return TG_E_UNSUPPORTED;
  k.tiles_total = (int)tt;
  k.nsplit = d->nsplit;
  k.ysum = d->y_sum ? 1 : 0;
  k.b_blocks = d->Cy / c.b_blk;
  const bool is16 = d->dtype != TG_F32;
  const int eb = is16 ? 2 : 4;
  const int padb = is16 ? 32 : 16;  // row padding of the kernel (PADB)
  const size_t lds = (size_t)tw * th * (c.b_blk * eb + padb) + (size_t)k.ih * k.iw * (c.a_blk * eb + padb);
  if (lds > 160 * 1024) return TG_E_UNSUPPORTED;
  {  // register staging capacity of the kernel (UX / UY pieces per thread)
    const int xv = c.a_blk * eb / 16, yv = c.b_blk * eb / 16;
    const int ux = is16 ? 11 : 22, uy = is16 ? 4 : 8;
    if (k.ih * k.iw * xv > 256 * ux || tw * th * yv > 256 * uy) return TG_E_UNSUPPORTED;
  }
  const int tpw = d->taps_per_wg > 0 ? d->taps_per_wg : d->ntaps;
  if (!((d->ntaps == 9 && (tpw == 9 || tpw == 3)) || (d->ntaps == 16 && (tpw == 16 || tpw == 4)))) return TG_E_UNSUPPORTED;
  dim3 grid((unsigned)(d->nsplit * njobs), (unsigned)((d->Cx / c.a_blk) * k.b_blocks), (unsigned)(d->ntaps / tpw));
  hipStream_t st = (hipStream_t)stream;
  const bool split = tpw != d->ntaps;
#define TG_WG(T_, NT_, TP_, AW_, BT_) return launch_wgrad<T_, NT_, TP_, AW_, BT_>(k, grid, lds, st)
  if (d->dtype == TG_BF16) {
    switch (cfg) {
      case 0:
        if (split) TG_WG(BF16, 9, 3, 4, 4);
        else return launch_wgrad<BF16, 9, 9, 4, 2, 8>(k, grid, lds, st);  // 8 waves: 2-11 % faster (tools/microbench.py wgrad)
      case 1: if (split) TG_WG(BF16, 9, 3, 2, 2); else TG_WG(BF16, 9, 9, 2, 2);
      case 2: if (split) TG_WG(BF16, 9, 3, 4, 2); else return launch_wgrad<BF16, 9, 9, 4, 1, 8>(k, grid, lds, st);
      case 3: if (split) TG_WG(BF16, 16, 4, 4, 2); else return launch_wgrad<BF16, 16, 16, 4, 1, 8>(k, grid, lds, st);
      case 4: if (split) TG_WG(BF16, 9, 3, 2, 1); else TG_WG(BF16, 9, 9, 2, 1);
    }
  } else if (d->dtype == TG_F16) {
    switch (cfg) {
      case 0:
        if (split) TG_WG(F16, 9, 3, 4, 4);
        else return launch_wgrad<F16, 9, 9, 4, 2, 8>(k, grid, lds, st);
      case 1: if (split) TG_WG(F16, 9, 3, 2, 2); else TG_WG(F16, 9, 9, 2, 2);
      case 2: if (split) TG_WG(F16, 9, 3, 4, 2); else return launch_wgrad<F16, 9, 9, 4, 1, 8>(k, grid, lds, st);
      case 3: if (split) TG_WG(F16, 16, 4, 4, 2); else return launch_wgrad<F16, 16, 16, 4, 1, 8>(k, grid, lds, st);
      case 4: if (split) TG_WG(F16, 9, 3, 2, 1); else TG_WG(F16, 9, 9, 2, 1);
    }
  } else {
    switch (cfg) {
      case 0: if (split) TG_WG(F32, 9, 3, 4, 4); else TG_WG(F32, 9, 9, 4, 4);
      case 1: if (split) TG_WG(F32, 9, 3, 2, 2); else TG_WG(F32, 9, 9, 2, 2);
      case 2: if (split) TG_WG(F32, 9, 3, 4, 2); else TG_WG(F32, 9, 9, 4, 2);
      case 3: if (split) TG_WG(F32, 16, 4, 4, 2); else TG_WG(F32, 16, 16, 4, 2);
      case 4: if (split) TG_WG(F32, 9, 3, 2, 1); else TG_WG(F32, 9, 9, 2, 1);
    }
  }
#undef TG_WG
  return TG_E_UNSUPPORTED;
}
}  // namespace

extern "C" int tg_wgrad_finalize(const float* slab, int nsplit, int ntaps, int ca_p, int cb_p, int ca, int cb,
                                 float* grad, int64_t s_a, int64_t s_b, const int32_t* slot_off_dev, int accumulate,
                                 float* bias_grad, int64_t slab_stride, void* stream) {
  if (!slab || !grad || !slot_off_dev || nsplit <= 0 || ntaps <= 0 || ca <= 0 || cb <= 0 || ca > ca_p || cb > cb_p)
    return TG_E_BADARG;
  if (cb_p % 4 || slab_stride % 4 || !tg_aligned16(slab)) return TG_E_ALIGN;
  if (slab_stride < (int64_t)ntaps * ca_p * cb_p + (bias_grad ? cb_p : 0)) return TG_E_BADARG;
  const long long units = (long long)ntaps * ca_p * (cb_p / 4) + (bias_grad ? cb_p / 4 : 0);
  const int blocks = (int)std::min<long long>((units + 63) / 64, 4096);
  hipLaunchKernelGGL(wgrad_finalize_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, slab, nsplit, ntaps, ca_p,
                     cb_p, ca, cb, grad, (long long)s_a, (long long)s_b, slot_off_dev, accumulate, bias_grad,
                     (long long)slab_stride);
  return tg_launch_status();
}

extern "C" int tg_wgrad_finalize_multi(const int64_t* jobs_dev, int njobs, int blocks_per_job, void* stream) {
  if (!jobs_dev || njobs <= 0 || blocks_per_job <= 0) return TG_E_BADARG;
  static std::atomic<bool> attr_done{false};  // one-time function attribute (benign race: idempotent)
  if (!attr_done) {
    TG_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_finalize_multi_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, kFold2Lds));
    attr_done = true;
  }
  hipLaunchKernelGGL(wgrad_finalize_multi_kernel, dim3((unsigned)blocks_per_job, (unsigned)njobs), dim3(256), kFold2Lds,
                     (hipStream_t)stream, (const long long*)jobs_dev);
  return tg_launch_status();
}

extern "C" int tg_wgrad_fold_items(const int64_t* jobs_dev, int njobs, int nitems, int max_taps, void* stream) {
  if (!jobs_dev || njobs <= 0 || nitems <= 0 || (max_taps != 9 && max_taps != 16)) return TG_E_BADARG;
  const int lds = (64 * (16 * max_taps + 1)) * 4;
  static std::atomic<bool> attr_done{false};  // one-time function attribute (benign race: idempotent)
  if (!attr_done) {
    TG_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_fold_items_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, kFold2Lds));
    attr_done = true;
  }
  // (experiments build: TECOGAN_FOLD_WGS makes a smaller grid walk the items - nothing to gain, profiles/r03_r_rw_dma_ab.log)
  static const int cap = [] { const char* e = kTgExperiments ? getenv("TECOGAN_FOLD_WGS") : nullptr; return e ? atoi(e) : 0; }();   // 0: one workgroup per item
  const int grid = cap > 0 && cap < nitems ? cap : nitems;
  hipLaunchKernelGGL(wgrad_fold_items_kernel, dim3((unsigned)grid), dim3(256), lds, (hipStream_t)stream,
                     (const long long*)jobs_dev, njobs, nitems);
  return tg_launch_status();
}
