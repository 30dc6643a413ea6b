// Resampling kernels of the flow network f_net (code/models.py:9-50): MaxPool2d(2) after every encoder block and
// nn.Upsample(scale_factor=2, bilinear, align_corners=False) after every decoder block, on NHWC activations.
// Both are HBM-bound channel-wise maps: one thread per 16-byte channel vector, consecutive lanes walk the channel
// dimension so every load/store is a fully coalesced 1 KiB wavefront access.
#include "common.h"
#include <algorithm>

namespace {

inline int grid_for(long long total, int block = 256, int cap = 8192) {
  return (int)std::max<long long>(1, std::min<long long>((total + block - 1) / block, cap));
}

template <typename T>
__global__ void maxpool2_kernel(const char* __restrict__ src, char* __restrict__ dst, int N, int H, int W, int C) {
  using TR = ElemTraits<T>;
  const int nvec = C / TR::kVec, OH = H / 2, OW = W / 2;
  const long long total = (long long)N * OH * OW * nvec;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int vc = (int)(i % nvec);
    long long r = i / nvec;
    const int ox = (int)(r % OW);
    r /= OW;
    const int oy = (int)(r % OH);
    const int n = (int)(r / OH);
    const long long base = ((((long long)n * H + 2 * oy) * W + 2 * ox) * C + vc * TR::kVec) * TR::kBytes;
    const long long row = (long long)W * C * TR::kBytes, col = (long long)C * TR::kBytes;
    float a[TR::kVec], b[TR::kVec], c[TR::kVec], d[TR::kVec];
    Vec<T>::load(src + base, a);
    Vec<T>::load(src + base + col, b);
    Vec<T>::load(src + base + row, c);
    Vec<T>::load(src + base + row + col, d);
#pragma unroll
    for (int e = 0; e < TR::kVec; ++e) a[e] = fmaxf(fmaxf(a[e], b[e]), fmaxf(c[e], d[e]));
    Vec<T>::store(dst + ((((long long)n * OH + oy) * OW + ox) * C + vc * TR::kVec) * TR::kBytes, a);
  }
}

// src = 0.5*(dst+0.5)-0.5 clamped at 0 (area_pixel_compute_source_index, align_corners=False)
__device__ __forceinline__ void up2_coord(int d, int in_size, int& i0, int& i1, float& l1) {
  float s = __fsub_rn(__fmul_rn(0.5f, __fadd_rn((float)d, 0.5f)), 0.5f);
  s = s < 0.f ? 0.f : s;
  i0 = (int)s;
  i1 = i0 + ((i0 < in_size - 1) ? 1 : 0);
  l1 = __fsub_rn(s, (float)i0);
}

template <typename T>
__global__ void up2_bilinear_kernel(const char* __restrict__ src, char* __restrict__ dst, int N, int H, int W, int C) {
  using TR = ElemTraits<T>;
  const int nvec = C / TR::kVec, OH = 2 * H, OW = 2 * W;
  const long long total = (long long)N * OH * OW * nvec;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int vc = (int)(i % nvec);
    long long r = i / nvec;
    const int ox = (int)(r % OW);
    r /= OW;
    const int oy = (int)(r % OH);
    const int n = (int)(r / OH);
    int y0, y1, x0, x1;
    float ly, lx;
    up2_coord(oy, H, y0, y1, ly);
    up2_coord(ox, W, x0, x1, lx);
    const float hy = 1.f - ly, hx = 1.f - lx;
    const char* img = src + ((long long)n * H * W * C + vc * TR::kVec) * TR::kBytes;
    const long long col = (long long)C * TR::kBytes, row = (long long)W * col;
    float a[TR::kVec], b[TR::kVec], c[TR::kVec], d[TR::kVec];
    Vec<T>::load(img + y0 * row + x0 * col, a);
    Vec<T>::load(img + y0 * row + x1 * col, b);
    Vec<T>::load(img + y1 * row + x0 * col, c);
    Vec<T>::load(img + y1 * row + x1 * col, d);
#pragma unroll
    for (int e = 0; e < TR::kVec; ++e) {
      const float top = __fmaf_rn(hx, a[e], lx * b[e]);
      const float bot = __fmaf_rn(hx, c[e], lx * d[e]);
      a[e] = __fmaf_rn(hy, top, ly * bot);
    }
    Vec<T>::store(dst + ((((long long)n * OH + oy) * OW + ox) * C + vc * TR::kVec) * TR::kBytes, a);
  }
}

// Backward of nn.Upsample(scale_factor=2, bilinear, align_corners=False): dsrc[i][j] = sum of ddst over the <= 4 x 4 output
// pixels whose bilinear footprint holds (i, j).  Output row 2i and 2i + 1 give source row i the weight 0.75, rows 2i - 1 and
// 2i + 2 the weight 0.25; at the border the clamped neighbour's quarter comes back to the edge row (rows 0 and 2H - 1 give it
// weight 1).  mask (nullable, [N][H][W][C]): the tensor that was up-sampled is a LeakyReLU(0.2) output; the result is multiplied
// by its derivative (code/models.py:14-19: conv, lrelu, conv, lrelu, upsample).
template <typename T>
__global__ void up2_bilinear_bwd_kernel(const char* __restrict__ ddst, const char* __restrict__ mask, char* __restrict__ dsrc,
                                        int N, int H, int W, int C) {
  using TR = ElemTraits<T>;
  const int nvec = C / TR::kVec, OH = 2 * H, OW = 2 * W;
  const long long total = (long long)N * H * W * nvec;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int vc = (int)(i % nvec);
    long long r = i / nvec;
    const int x = (int)(r % W);
    r /= W;
    const int y = (int)(r % H);
    const int n = (int)(r / H);
    // output rows / columns that touch source row y, and their weights (out-of-range ones folded into the edge)
    int oy[4] = {2 * y - 1, 2 * y, 2 * y + 1, 2 * y + 2}, ox[4] = {2 * x - 1, 2 * x, 2 * x + 1, 2 * x + 2};
    float wy[4] = {0.25f, 0.75f, 0.75f, 0.25f}, wx[4] = {0.25f, 0.75f, 0.75f, 0.25f};
    if (y == 0) { wy[0] = 0.f; wy[1] = 1.f; }
    if (y == H - 1) { wy[3] = 0.f; wy[2] = 1.f; }
    if (x == 0) { wx[0] = 0.f; wx[1] = 1.f; }
    if (x == W - 1) { wx[3] = 0.f; wx[2] = 1.f; }
    float acc[TR::kVec];
#pragma unroll
    for (int e = 0; e < TR::kVec; ++e) acc[e] = 0.f;
    const char* img = ddst + ((long long)n * OH * OW * C + vc * TR::kVec) * TR::kBytes;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      if (wy[a] == 0.f) continue;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        if (wx[b] == 0.f) continue;
        float v[TR::kVec];
        Vec<T>::load(img + ((long long)oy[a] * OW + ox[b]) * C * TR::kBytes, v);
        const float w = wy[a] * wx[b];
#pragma unroll
        for (int e = 0; e < TR::kVec; ++e) acc[e] += w * v[e];
      }
    }
    const long long off = ((((long long)n * H + y) * W + x) * C + vc * TR::kVec) * TR::kBytes;
    if (mask) {
      float m[TR::kVec];
      Vec<T>::load(mask + off, m);
#pragma unroll
      for (int e = 0; e < TR::kVec; ++e) acc[e] *= (m[e] > 0.f ? 1.f : 0.2f);
    }
    Vec<T>::store(dsrc + off, acc);
  }
}

// d(pre-activation) of f_net's output layer: out = 24 tanh(p) (code/models.py:49-50), so dp = dout * (24 - out^2 / 24);
// dout / out are fp32 [N][2][H][W] (the layout of the estimator's result), dpre is NHWC [N][H][W][32] with channels 2.. zero
template <typename T>
__global__ void tanh24_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ out, char* __restrict__ dpre,
                                  long long npix, int HW) {
  using TR = ElemTraits<T>;
  for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < npix; p += (long long)gridDim.x * blockDim.x) {
    const long long n = p / HW, q = p % HW;
    float v[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) v[c] = 0.f;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const long long i = (n * 2 + c) * HW + q;
      const float o = out[i];
      v[c] = dout[i] * (24.f - o * o * (1.f / 24.f));
    }
    char* d = dpre + p * 32 * TR::kBytes;
#pragma unroll
    for (int k = 0; k < 32 / TR::kVec; ++k) Vec<T>::store(d + k * 16, v + k * TR::kVec);
  }
}

}  // namespace

#define FN_DISPATCH(dtype, KERNEL, grid, st, ...)                                                    \
  do {                                                                                               \
    if ((dtype) == TG_BF16) hipLaunchKernelGGL(KERNEL<BF16>, grid, dim3(256), 0, st, __VA_ARGS__);   \
    else if ((dtype) == TG_F16) hipLaunchKernelGGL(KERNEL<F16>, grid, dim3(256), 0, st, __VA_ARGS__); \
    else if ((dtype) == TG_F32) hipLaunchKernelGGL(KERNEL<F32>, grid, dim3(256), 0, st, __VA_ARGS__); \
    else return TG_E_BADARG;                                                                         \
  } while (0)

extern "C" int tg_up2_bilinear_bwd(int dtype, const void* ddst, const void* lrelu_mask, void* dsrc, int N, int H, int W, int C,
                                   void* stream) {
  if (!ddst || !dsrc || N <= 0 || H <= 0 || W <= 0 || C <= 0) return TG_E_BADARG;
  if (C % 32 || !tg_aligned16(ddst) || !tg_aligned16(dsrc) || (lrelu_mask && !tg_aligned16(lrelu_mask))) return TG_E_ALIGN;
  const long long total = (long long)N * H * W * (C / (dtype == TG_F32 ? 4 : 8));
  FN_DISPATCH(dtype, up2_bilinear_bwd_kernel, dim3(grid_for(total)), (hipStream_t)stream, (const char*)ddst,
              (const char*)lrelu_mask, (char*)dsrc, N, H, W, C);
  return tg_launch_status();
}

extern "C" int tg_tanh24_bwd(int dtype, const float* dout_nchw, const float* out_nchw, void* dpre_nhwc32, int N, int H, int W,
                             void* stream) {
  if (!dout_nchw || !out_nchw || !dpre_nhwc32 || N <= 0 || H <= 0 || W <= 0) return TG_E_BADARG;
  if (!tg_aligned16(dpre_nhwc32)) return TG_E_ALIGN;
  const long long npix = (long long)N * H * W;
  FN_DISPATCH(dtype, tanh24_bwd_kernel, dim3(grid_for(npix)), (hipStream_t)stream, dout_nchw, out_nchw, (char*)dpre_nhwc32, npix,
              H * W);
  return tg_launch_status();
}

extern "C" int tg_maxpool2(int dtype, const void* src, void* dst, int N, int H, int W, int C, void* stream) {
  if (!src || !dst || N <= 0 || H <= 0 || W <= 0 || C <= 0 || (H & 1) || (W & 1)) return TG_E_BADARG;
  if (C % 32 || !tg_aligned16(src) || !tg_aligned16(dst)) return TG_E_ALIGN;
  const long long total = (long long)N * (H / 2) * (W / 2) * (C / (dtype == TG_F32 ? 4 : 8));
  if (dtype == TG_BF16)
    hipLaunchKernelGGL(maxpool2_kernel<BF16>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                       (const char*)src, (char*)dst, N, H, W, C);
  else if (dtype == TG_F16)
    hipLaunchKernelGGL(maxpool2_kernel<F16>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                       (const char*)src, (char*)dst, N, H, W, C);
  else if (dtype == TG_F32)
    hipLaunchKernelGGL(maxpool2_kernel<F32>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                       (const char*)src, (char*)dst, N, H, W, C);
  else
    return TG_E_BADARG;
  return tg_launch_status();
}

extern "C" int tg_up2_bilinear(int dtype, const void* src, void* dst, int N, int H, int W, int C, void* stream) {
  if (!src || !dst || N <= 0 || H <= 0 || W <= 0 || C <= 0) return TG_E_BADARG;
  if (C % 32 || !tg_aligned16(src) || !tg_aligned16(dst)) return TG_E_ALIGN;
  const long long total = (long long)N * (2 * H) * (2 * W) * (C / (dtype == TG_F32 ? 4 : 8));
  if (dtype == TG_BF16)
    hipLaunchKernelGGL(up2_bilinear_kernel<BF16>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                       (const char*)src, (char*)dst, N, H, W, C);
  else if (dtype == TG_F16)
    hipLaunchKernelGGL(up2_bilinear_kernel<F16>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                       (const char*)src, (char*)dst, N, H, W, C);
  else if (dtype == TG_F32)
    hipLaunchKernelGGL(up2_bilinear_kernel<F32>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                       (const char*)src, (char*)dst, N, H, W, C);
  else
    return TG_E_BADARG;
  return tg_launch_status();
}
