// HBM-bound pieces of the opt-in VGG feature loss (code/train.py:30-45,124-127,253-273; code/ops.py:144-213).  The VGG-19
// convolutions themselves run on tg_conv / tg_conv3x3_rw; what is here is what sits between them:
//   tg_vgg_input        fp32 NCHW frames -> NHWC (32 channels, 3 live) with the reference's input arithmetic
//                       deprocess(x) * 255 - VGG_MEAN = 127.5 * x + 127.5 - mean[c]          (code/train.py:31-32)
//   tg_cosine_loss      per pixel: cos = <g, t> / (|g| |t|), |v| = sqrt(sum_c v^2 + 1e-12); accumulates sum(cos) and writes
//                       d(coef * sum cos)/dg, optionally masked by g > 0 (g is a ReLU output)   (code/train.py:258-266,
//                       with the channel-wise L2 norm the reference's torch.min line was written to be)
//   tg_maxpool2_bwd     gradient of MaxPool2d(2,2) routed to the first maximum of each window, + an optional second
//                       gradient (the feature tap on the same tensor), x relu'(a) of the pooled tensor (a ReLU output)
//   tg_vgg_input_grad   d(loss)/d(pre-sigmoid) += d(loss)/d(vgg input) * 127.5 * g (1 - g)
// All bounded by HBM: one pass over the tensors named, 16-byte vectors, a wavefront per pixel in the cosine kernel.
#include "common.h"

namespace {

inline int vgg_grid(long long total, int per_block = 256) {
  long long g = (total + per_block - 1) / per_block;
  return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

template <typename T>
__global__ void vgg_input_kernel(const float* __restrict__ src, char* __restrict__ dst, long long npix, int HW, float scale,
                                 float s0, float s1, float s2) {
  using TR = ElemTraits<T>;
  constexpr int NV = 32 / TR::kVec;  // 16-byte vectors per 32-channel pixel
  const long long total = npix * NV;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int v = (int)(i % NV);
    const long long p = i / NV;
    float o[TR::kVec];
#pragma unroll
    for (int e = 0; e < TR::kVec; ++e) o[e] = 0.f;
    if (v == 0) {
      const long long n = p / HW, q = p % HW;
      const float* s = src + n * 3 * HW + q;
      o[0] = __fmaf_rn(scale, s[0], s0);
      o[1] = __fmaf_rn(scale, s[HW], s1);
      o[2] = __fmaf_rn(scale, s[2 * (long long)HW], s2);
    }
    Vec<T>::store(dst + i * 16, o);
  }
}

// one wavefront per pixel; C/64 channels per lane (C in {64,128,256,512} -> 1 element or 2..8 as 16-byte vectors is not
// uniform, so lanes walk vectors of kVec elements with a stride of 64 vectors)
template <typename T>
__global__ __launch_bounds__(256) void cosine_loss_kernel(const char* __restrict__ fg, const char* __restrict__ ft,
                                                          char* __restrict__ dg, long long npix, int C, float coef,
                                                          int relu_mask, float* __restrict__ acc,
                                                          const float* __restrict__ loss_scale) {
  using TR = ElemTraits<T>;
  if (loss_scale) coef *= *loss_scale;  // fp16 mode: backward seeds carry the dynamic loss scale
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nvec = C / TR::kVec;
  float wsum = 0.f;
  for (long long p = blockIdx.x * 4LL + wave; p < npix; p += gridDim.x * 4LL) {
    const char* a = fg + p * C * TR::kBytes;
    const char* b = ft + p * C * TR::kBytes;
    float saa = 0.f, sbb = 0.f, sab = 0.f;
    for (int v = lane; v < nvec; v += 64) {
      float x[TR::kVec], y[TR::kVec];
      Vec<T>::load(a + v * 16, x);
      Vec<T>::load(b + v * 16, y);
#pragma unroll
      for (int e = 0; e < TR::kVec; ++e) {
        saa = __fmaf_rn(x[e], x[e], saa);
        sbb = __fmaf_rn(y[e], y[e], sbb);
        sab = __fmaf_rn(x[e], y[e], sab);
      }
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
      saa += __shfl_xor(saa, m);
      sbb += __shfl_xor(sbb, m);
      sab += __shfl_xor(sab, m);
    }
    const float na = sqrtf(saa + 1e-12f), nb = sqrtf(sbb + 1e-12f);
    const float inv = 1.f / (na * nb);
    const float cosv = sab * inv;
    wsum += cosv;  // identical in every lane
    // d cos / d a_c = b_c / (na nb) - a_c * <a,b> / (na^3 nb)
    const float k1 = coef * inv, k2 = coef * cosv / (na * na);
    for (int v = lane; v < nvec; v += 64) {
      float x[TR::kVec], y[TR::kVec];
      Vec<T>::load(a + v * 16, x);
      Vec<T>::load(b + v * 16, y);
#pragma unroll
      for (int e = 0; e < TR::kVec; ++e) {
        const float g = k1 * y[e] - k2 * x[e];
        y[e] = (relu_mask && !(x[e] > 0.f)) ? 0.f : g;
      }
      Vec<T>::store(dg + (p * C * TR::kBytes + v * 16), y);
    }
  }
  __shared__ float part[4];
  if (lane == 0) part[wave] = wsum;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(acc, part[0] + part[1] + part[2] + part[3]);
}

template <typename T>
__global__ void maxpool2_bwd_kernel(const char* __restrict__ a, const char* __restrict__ dpool, const char* __restrict__ res,
                                    char* __restrict__ out, int N, int H, int W, int C, int relu_mask) {
  using TR = ElemTraits<T>;
  const int nvec = C / TR::kVec, PH = H / 2, PW = W / 2;
  const long long total = (long long)N * PH * PW * nvec;
  const long long col = (long long)C * TR::kBytes, row = (long long)W * col;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int vc = (int)(i % nvec);
    long long r = i / nvec;
    const int px = (int)(r % PW);
    r /= PW;
    const int py = (int)(r % PH);
    const int n = (int)(r / PH);
    const long long base = ((long long)n * H * W * C + vc * TR::kVec) * TR::kBytes + (2 * py) * row + (2 * px) * col;
    float v[4][TR::kVec], d[TR::kVec];
    Vec<T>::load(a + base, v[0]);
    Vec<T>::load(a + base + col, v[1]);
    Vec<T>::load(a + base + row, v[2]);
    Vec<T>::load(a + base + row + col, v[3]);
    Vec<T>::load(dpool + ((((long long)n * PH + py) * PW + px) * C + vc * TR::kVec) * TR::kBytes, d);
    int arg[TR::kVec];
#pragma unroll
    for (int e = 0; e < TR::kVec; ++e) {  // first maximum in scan order (aten::max_pool2d_with_indices)
      float m = v[0][e];
      int k = 0;
#pragma unroll
      for (int q = 1; q < 4; ++q)
        if (v[q][e] > m) { m = v[q][e]; k = q; }
      arg[e] = k;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const long long off = base + (q >> 1) * row + (q & 1) * col;
      float o[TR::kVec];
      if (res) Vec<T>::load(res + off, o);
#pragma unroll
      for (int e = 0; e < TR::kVec; ++e) {
        float g = (res ? o[e] : 0.f) + (arg[e] == q ? d[e] : 0.f);
        if (relu_mask && !(v[q][e] > 0.f)) g = relu_mask == 2 ? 0.2f * g : 0.f;  // 1: ReLU, 2: LeakyReLU(0.2)
        o[e] = g;
      }
      Vec<T>::store(out + off, o);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void vgg_input_grad_kernel(const char* __restrict__ dx, const float* __restrict__ gen,
                                                            char* __restrict__ dpre, long long npix, int HW, float scale,
                                                            float* __restrict__ bias_acc) {
  using TR = ElemTraits<T>;
  float cs[3] = {0.f, 0.f, 0.f};
  for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < npix; p += (long long)gridDim.x * blockDim.x) {
    const long long n = p / HW, q = p % HW;
    const float* g = gen + n * 3 * HW + q;
    float d[TR::kVec], o[TR::kVec];
    Vec<T>::load(dx + p * 32 * TR::kBytes, d);
    Vec<T>::load(dpre + p * 32 * TR::kBytes, o);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float s = g[c * (long long)HW];
      const float add = d[c] * scale * s * (1.f - s);
      o[c] += add;
      cs[c] += add;
    }
    Vec<T>::store(dpre + p * 32 * TR::kBytes, o);
  }
  if (bias_acc) {  // the output layer's bias gradient is the channel sum of d(loss)/d(pre-sigmoid): add this part of it
    __shared__ float sh[3][4];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) cs[c] += __shfl_xor(cs[c], m);
      if ((threadIdx.x & 63) == 0) sh[c][threadIdx.x >> 6] = cs[c];
    }
    __syncthreads();
    if (threadIdx.x < 3) atomicAdd(bias_acc + threadIdx.x, sh[threadIdx.x][0] + sh[threadIdx.x][1] + sh[threadIdx.x][2] + sh[threadIdx.x][3]);
  }
}

}  // namespace

#define VGG_DISPATCH(dtype, KERNEL, grid, block, st, ...)                                        \
  do {                                                                                           \
    if ((dtype) == TG_BF16) hipLaunchKernelGGL(KERNEL<BF16>, grid, block, 0, st, __VA_ARGS__);   \
    else if ((dtype) == TG_F16) hipLaunchKernelGGL(KERNEL<F16>, grid, block, 0, st, __VA_ARGS__); \
    else if ((dtype) == TG_F32) hipLaunchKernelGGL(KERNEL<F32>, grid, block, 0, st, __VA_ARGS__); \
    else return TG_E_BADARG;                                                                     \
  } while (0)

extern "C" int tg_vgg_input(int dtype, const float* src_nchw, void* dst_nhwc32, int N, int H, int W, float scale,
                            const float* shift3, void* stream) {
  if (!src_nchw || !dst_nhwc32 || !shift3 || N <= 0 || H <= 0 || W <= 0) return TG_E_BADARG;
  if (!tg_aligned16(dst_nhwc32)) return TG_E_ALIGN;
  const long long npix = (long long)N * H * W;
  VGG_DISPATCH(dtype, vgg_input_kernel, dim3(vgg_grid(npix * (dtype == TG_F32 ? 8 : 4))), dim3(256), (hipStream_t)stream,
               src_nchw, (char*)dst_nhwc32, npix, H * W, scale, shift3[0], shift3[1], shift3[2]);
  return tg_launch_status();
}

extern "C" int tg_cosine_loss(int dtype, const void* fg, const void* ft, void* dg, int64_t npix, int C, float coef,
                              int relu_mask, float* acc, const float* loss_scale, void* stream) {
  if (!fg || !ft || !dg || !acc || npix <= 0 || C <= 0) return TG_E_BADARG;
  if (C % 32 || !tg_aligned16(fg) || !tg_aligned16(ft) || !tg_aligned16(dg)) return TG_E_ALIGN;
  VGG_DISPATCH(dtype, cosine_loss_kernel, dim3(vgg_grid(npix, 4)), dim3(256), (hipStream_t)stream, (const char*)fg,
               (const char*)ft, (char*)dg, (long long)npix, C, coef, relu_mask, acc, loss_scale);
  return tg_launch_status();
}

extern "C" int tg_maxpool2_bwd(int dtype, const void* a, const void* dpool, const void* res, void* out, int N, int H, int W,
                               int C, int relu_mask, void* stream) {
  if (!a || !dpool || !out || N <= 0 || H <= 0 || W <= 0 || C <= 0 || (H & 1) || (W & 1)) return TG_E_BADARG;
  if (C % 32 || !tg_aligned16(a) || !tg_aligned16(dpool) || !tg_aligned16(out) || (res && !tg_aligned16(res)))
    return TG_E_ALIGN;
  const long long total = (long long)N * (H / 2) * (W / 2) * (C / (dtype == TG_F32 ? 4 : 8));
  VGG_DISPATCH(dtype, maxpool2_bwd_kernel, dim3(vgg_grid(total)), dim3(256), (hipStream_t)stream, (const char*)a,
               (const char*)dpool, (const char*)res, (char*)out, N, H, W, C, relu_mask);
  return tg_launch_status();
}

extern "C" int tg_vgg_input_grad(int dtype, const void* dx_nhwc32, const float* gen_nchw, void* dpre_nhwc32, int N, int H,
                                 int W, float scale, float* bias_acc3, void* stream) {
  if (!dx_nhwc32 || !gen_nchw || !dpre_nhwc32 || N <= 0 || H <= 0 || W <= 0) return TG_E_BADARG;
  if (!tg_aligned16(dx_nhwc32) || !tg_aligned16(dpre_nhwc32)) return TG_E_ALIGN;
  const long long npix = (long long)N * H * W;
  VGG_DISPATCH(dtype, vgg_input_grad_kernel, dim3(vgg_grid(npix, 1024)), dim3(256), (hipStream_t)stream, (const char*)dx_nhwc32,
               gen_nchw, (char*)dpre_nhwc32, npix, H * W, scale, bias_acc3);
  return tg_launch_status();
}
