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

}  // namespace

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
