// HBM-bound NHWC kernels: training-mode batch norm (code/ops.py:75-77, eps 1e-3, momentum 0.1) forward/backward,
// the fc head (code/models.py:143-145), loss reductions (code/train.py:205-333) and fused Adam (main.py:239-243).
// All kernels move 16 bytes per lane per access; per-channel quantities are kept in registers because a thread's
// channel vector is fixed for its whole grid-stride loop.
#include "common.h"
#ifdef TG_VEC_LD_NT   // A/B only: the tensors these kernels stream are read once (profiles/r05_u_write_through_ab.log, section 6)
#define TG_LDG16(p) __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p))
#else
#define TG_LDG16(p) (*reinterpret_cast<const u32x4*>(p))
#endif
#include <type_traits>

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  return v;
}

// Per-channel accumulators ([groups][2][C] floats: the statistics a conv epilogue adds up, the two sums of the backward pass)
// come in R replica blocks: producer workgroup b adds into block b mod R, because a few hundred workgroups adding into the
// same 128 floats serialise in L2 (the discriminator's stage-1 conv: 10.3 us without statistics, 19.9 with them at R = 1,
// 13.0 at R = 4: tools/mb_stats.py).  The consumer folds the blocks first: sh[i] = sum_r acc[r][i], in replica order.
// Every thread reads the E consecutive channels it owns from all R blocks: independent 16-byte loads, ALL issued before the
// first use (R as a compile-time constant: a run-time loop waits for each block's loads before it issues the next block's -
// R round trips, measured slower than a fold through LDS, which in turn put a barrier in front of the kernel's first tensor
// loads: +1.2 us per launch).
template <int E, int RR> __device__ __forceinline__ void load_folded_n(const float* __restrict__ acc, size_t block, float* out) {
  f32x4 t[RR][E / 4];
#pragma unroll
  for (int r = 0; r < RR; ++r)
#pragma unroll
    for (int e = 0; e < E / 4; ++e) t[r][e] = *reinterpret_cast<const f32x4*>(acc + r * block + 4 * e);
#pragma unroll
  for (int e = 0; e < E; ++e) out[e] = 0.f;
#pragma unroll
  for (int r = 0; r < RR; ++r)
#pragma unroll
    for (int e = 0; e < E; ++e) out[e] += t[r][e / 4][e % 4];
}
template <int E> __device__ __forceinline__ void load_folded(const float* __restrict__ acc, int R, size_t block, float* out) {
  switch (R) {
    case 1: load_folded_n<E, 1>(acc, block, out); return;
    case 2: load_folded_n<E, 2>(acc, block, out); return;
    case 4: load_folded_n<E, 4>(acc, block, out); return;
    case 8: load_folded_n<E, 8>(acc, block, out); return;
  }
#pragma unroll
  for (int e = 0; e < E; ++e) out[e] = 0.f;
  for (int r = 0; r < R; ++r) {
#pragma unroll
    for (int e = 0; e < E; ++e) out[e] += acc[r * block + e];
  }
}
__device__ __forceinline__ float load_folded1(const float* __restrict__ acc, int R, size_t block) {
  float s = 0.f;
  for (int r = 0; r < R; ++r) s += acc[r * block];
  return s;
}

// Threads are arranged [rows = 256/VPP][VPP] where VPP = C / kVec vectors per pixel (a power of two <= 32).
template <typename T>
__global__ __launch_bounds__(256) void bn_apply_kernel(const char* __restrict__ z, const float* __restrict__ stats_rep, int R,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      const char* __restrict__ skip, char* __restrict__ y,
                                                      float* __restrict__ rmean, float* __restrict__ rvar,
                                                      float* __restrict__ save, int N, int HW, int C, int groups,
                                                      int act, float eps, float momentum,
                                                      long long* __restrict__ nbt) {
  using TR = ElemTraits<T>;
  constexpr int E = TR::kVec;
  const int vpp = C / E;
  const int vec = threadIdx.x % vpp, prow = threadIdx.x / vpp, rows = 256 / vpp;
  const int grp = blockIdx.y;
  const long long npix = (long long)(N / groups) * HW;
  const float cnt = (float)npix;
  const size_t rblock = (size_t)groups * 2 * C;
  // The first trip's tensor loads go out BEFORE the per-channel parameters are fetched (raw, from clamped addresses): the two
  // do not depend on each other, and most workgroups make exactly one trip - their run time was two dependent round trips
  const long long base = (long long)grp * npix;
  const long long step = (long long)gridDim.x * rows;
  long long p = (long long)blockIdx.x * rows + prow;
  u32x4 rz[4], rs[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const long long pix = p + u * step < npix ? p + u * step : npix - 1;
    const long long off = ((base + pix) * C + vec * E) * TR::kBytes;
    rz[u] = TG_LDG16(z + off);
    if (skip) rs[u] = TG_LDG16(skip + off);
  }
  float scale[E], shift[E], sum1[E], sum2[E];
  load_folded<E>(stats_rep + (grp * 2 + 0) * C + vec * E, R, rblock, sum1);
  load_folded<E>(stats_rep + (grp * 2 + 1) * C + vec * E, R, rblock, sum2);
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int c = vec * E + e;
    const float mean = sum1[e] / cnt;
    float var = sum2[e] / cnt - mean * mean;
    var = var < 0.f ? 0.f : var;
    const float invstd = rsqrtf(var + eps);
    scale[e] = gamma[c] * invstd;
    shift[e] = beta[c] - mean * scale[e];
  }
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < C) {  // single writer for running stats and saved mean/invstd
    const int c = threadIdx.x;
    float rm = rmean ? rmean[c] : 0.f, rv = rvar ? rvar[c] : 0.f;
    for (int g2 = 0; g2 < groups; ++g2) {  // sequential updates, one per forward call of the reference
      const float mean = load_folded1(stats_rep + (g2 * 2 + 0) * C + c, R, rblock) / cnt;
      float var = load_folded1(stats_rep + (g2 * 2 + 1) * C + c, R, rblock) / cnt - mean * mean;
      var = var < 0.f ? 0.f : var;
      save[(g2 * 2 + 0) * C + c] = mean;
      save[(g2 * 2 + 1) * C + c] = rsqrtf(var + eps);
      rm = (1.f - momentum) * rm + momentum * mean;
      rv = (1.f - momentum) * rv + momentum * var * (cnt / (cnt - 1.f));
    }
    if (rmean) rmean[c] = rm;
    if (rvar) rvar[c] = rv;
    if (nbt && c == 0) *nbt += groups;  // num_batches_tracked: one per forward call of the reference
  }
  auto finish = [&](float* v, const float* sk, long long off) {
#pragma unroll
    for (int e = 0; e < E; ++e) {
      v[e] = v[e] * scale[e] + shift[e];
      if (act == TG_ACT_LRELU) v[e] = v[e] > 0.f ? v[e] : 0.2f * v[e];
      else if (act == TG_ACT_RELU) v[e] = v[e] > 0.f ? v[e] : 0.f;
      if (skip) v[e] += sk[e];
    }
    Vec<T>::store(y + off, v);
  };
#pragma unroll
  for (int u = 0; u < 4; ++u) {  // the trip fetched above
    if (p + u * step < npix) {
      float v[E], sk[E];
      Vec<T>::load(&rz[u], v);
      if (skip) Vec<T>::load(&rs[u], sk);
      finish(v, sk, ((base + p + u * step) * C + vec * E) * TR::kBytes);
    }
  }
  p += 4 * step;
  for (; p + 3 * step < npix; p += 4 * step) {  // four pixels per trip, all loads issued before the first use
    float v[4][E], sk[4][E];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long long off = ((base + p + u * step) * C + vec * E) * TR::kBytes;
      Vec<T>::load(z + off, v[u]);
      if (skip) Vec<T>::load(skip + off, sk[u]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) finish(v[u], sk[u], ((base + p + u * step) * C + vec * E) * TR::kBytes);
  }
  for (; p < npix; p += step) {
    const long long off = ((base + p) * C + vec * E) * TR::kBytes;
    float v[E], sk[E];
    Vec<T>::load(z + off, v);
    if (skip) Vec<T>::load(skip + off, sk);
    finish(v, sk, off);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const char* __restrict__ dy, const char* __restrict__ yact,
                                                           const char* __restrict__ z, const float* __restrict__ save,
                                                           float* __restrict__ red, int R, int N, int HW, int C, int groups,
                                                           int act) {
  using TR = ElemTraits<T>;
  constexpr int E = TR::kVec;
  __shared__ float sh[256 * 2 * E];
  const int vpp = C / E;
  const int vec = threadIdx.x % vpp, prow = threadIdx.x / vpp, rows = 256 / vpp;
  const int grp = blockIdx.y;
  const long long npix = (long long)(N / groups) * HW;
  float mean[E], invstd[E], s1[E], s2[E];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    mean[e] = save[(grp * 2 + 0) * C + vec * E + e];
    invstd[e] = save[(grp * 2 + 1) * C + vec * E + e];
    s1[e] = s2[e] = 0.f;
  }
  const long long base = (long long)grp * npix;
  const long long step = (long long)gridDim.x * rows;
  auto accum = [&](const float* d, const float* zz, const float* a) {
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const float dd = (act == TG_ACT_LRELU) ? d[e] * (a[e] > 0.f ? 1.f : 0.2f) : d[e];
      s1[e] += dd;
      s2[e] += dd * (zz[e] - mean[e]) * invstd[e];
    }
  };
  long long p = (long long)blockIdx.x * rows + prow;
  // four pixels per trip with all their loads issued before the first use: a workgroup walks its pixels serially, and with
  // one pixel per trip every trip paid a full memory round trip (11.5 us for two 6 MB tensors on 96 workgroups)
  for (; p + 3 * step < npix; p += 4 * step) {
    float d[4][E], zz[4][E], a[4][E];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long long off = ((base + p + u * step) * C + vec * E) * TR::kBytes;
      Vec<T>::load(dy + off, d[u]);
      Vec<T>::load(z + off, zz[u]);
      if (act == TG_ACT_LRELU) Vec<T>::load(yact + off, a[u]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) accum(d[u], zz[u], a[u]);
  }
  for (; p < npix; p += step) {
    const long long off = ((base + p) * C + vec * E) * TR::kBytes;
    float d[E], zz[E], a[E];
    Vec<T>::load(dy + off, d);
    Vec<T>::load(z + off, zz);
    if (act == TG_ACT_LRELU) Vec<T>::load(yact + off, a);
    accum(d, zz, a);
  }
#pragma unroll
  for (int e = 0; e < E; ++e) {
    sh[(threadIdx.x * 2 + 0) * E + e] = s1[e];
    sh[(threadIdx.x * 2 + 1) * E + e] = s2[e];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += 256) {
    const int which = i / C, c = i % C;
    const int v = c / E, e = c % E;
    float s = 0.f;
    for (int r = 0; r < rows; ++r) s += sh[((r * vpp + v) * 2 + which) * E + e];
    atomicAdd(red + (size_t)(blockIdx.x & (R - 1)) * groups * 2 * C + (grp * 2 + which) * C + c, s);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const char* __restrict__ dy, const char* __restrict__ yact,
                                                          const char* __restrict__ z, const float* __restrict__ save,
                                                          const float* __restrict__ red_rep, int R, const float* __restrict__ gamma,
                                                          char* __restrict__ dz, float* __restrict__ dgamma,
                                                          float* __restrict__ dbeta, int N, int HW, int C, int groups,
                                                          int act, int red_raw) {
  using TR = ElemTraits<T>;
  constexpr int E = TR::kVec;
  const int vpp = C / E;
  const int vec = threadIdx.x % vpp, prow = threadIdx.x / vpp, rows = 256 / vpp;
  const int grp = blockIdx.y;
  const long long npix = (long long)(N / groups) * HW;
  const float inv_cnt = 1.f / (float)npix;
  const size_t rblock = (size_t)groups * 2 * C;
  // the first trip's tensor loads before the per-channel parameters (see bn_apply_kernel)
  const long long base = (long long)grp * npix;
  const long long step = (long long)gridDim.x * rows;
  long long p = (long long)blockIdx.x * rows + prow;
  u32x4 rd[4], rz[4], ra[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const long long pix = p + u * step < npix ? p + u * step : npix - 1;
    const long long off = ((base + pix) * C + vec * E) * TR::kBytes;
    rd[u] = TG_LDG16(dy + off);
    rz[u] = TG_LDG16(z + off);
    if (act == TG_ACT_LRELU) ra[u] = TG_LDG16(yact + off);
  }
  float mean[E], invstd[E], k0[E], m1[E], m2[E];
  load_folded<E>(red_rep + (grp * 2 + 0) * C + vec * E, R, rblock, m1);
  load_folded<E>(red_rep + (grp * 2 + 1) * C + vec * E, R, rblock, m2);
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int c = vec * E + e;
    mean[e] = save[(grp * 2 + 0) * C + c];
    invstd[e] = save[(grp * 2 + 1) * C + c];
    k0[e] = gamma[c] * invstd[e];
    // red_raw: the second sum is sum dy * z as the producing conv's epilogue left it (tg_conv, stats_mode 3), not yet
    // sum dy * (z - mean) * invstd
    if (red_raw) m2[e] = (m2[e] - mean[e] * m1[e]) * invstd[e];
    m1[e] *= inv_cnt;
    m2[e] *= inv_cnt;
  }
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < C) {
    const int c = threadIdx.x;
    float dg = 0.f, db = 0.f;
    for (int g2 = 0; g2 < groups; ++g2) {
      const float b1 = load_folded1(red_rep + (g2 * 2 + 0) * C + c, R, rblock);
      float g1 = load_folded1(red_rep + (g2 * 2 + 1) * C + c, R, rblock);
      if (red_raw) g1 = (g1 - save[(g2 * 2 + 0) * C + c] * b1) * save[(g2 * 2 + 1) * C + c];
      db += b1;
      dg += g1;
    }
    dgamma[c] += dg;
    dbeta[c] += db;
  }
  auto finish = [&](float* d, const float* zz, const float* a, long long off) {
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const float dd = (act == TG_ACT_LRELU) ? d[e] * (a[e] > 0.f ? 1.f : 0.2f) : d[e];
      const float xh = (zz[e] - mean[e]) * invstd[e];
      d[e] = k0[e] * (dd - m1[e] - xh * m2[e]);
    }
    Vec<T>::store(dz + off, d);
  };
#pragma unroll
  for (int u = 0; u < 4; ++u) {  // the trip fetched above
    if (p + u * step < npix) {
      float d[E], zz[E], a[E];
      Vec<T>::load(&rd[u], d);
      Vec<T>::load(&rz[u], zz);
      if (act == TG_ACT_LRELU) Vec<T>::load(&ra[u], a);
      finish(d, zz, a, ((base + p + u * step) * C + vec * E) * TR::kBytes);
    }
  }
  p += 4 * step;
  for (; p + 3 * step < npix; p += 4 * step) {  // four pixels per trip, all loads issued before the first use
    float d[4][E], zz[4][E], a[4][E];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long long off = ((base + p + u * step) * C + vec * E) * TR::kBytes;
      Vec<T>::load(dy + off, d[u]);
      Vec<T>::load(z + off, zz[u]);
      if (act == TG_ACT_LRELU) Vec<T>::load(yact + off, a[u]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) finish(d[u], zz[u], a[u], ((base + p + u * step) * C + vec * E) * TR::kBytes);
  }
  for (; p < npix; p += step) {
    const long long off = ((base + p) * C + vec * E) * TR::kBytes;
    float d[E], zz[E], a[E];
    Vec<T>::load(dy + off, d);
    Vec<T>::load(z + off, zz);
    if (act == TG_ACT_LRELU) Vec<T>::load(yact + off, a);
    finish(d, zz, a, off);
  }
}

#ifdef TG_EXPERIMENTS   // measured slower, alone and in the step (profiles/r04_y_bn_bwd_coop_ab.log)
// Batch-norm backward in ONE launch (round 4): the reduction, a grid-wide wait, the apply.  Same pixel / channel mapping as the
// two launches above; a thread keeps its U pixels' raw dy / z (/ yact) vectors in registers across the wait, so the tensors are
// read once.  The wait is a counter in the step's zeroed accumulator arena: every workgroup of the launch must be co-resident,
// which the host guarantees by size (tg_bn_bwd_coop refuses more than kBcMaxWgs = 192 workgroups: a CU holds 3-4 of them, so 64
// free CUs are enough, and two processes on one GPU fit twice over) - the other lane's persistent launches only delay the wait, nothing they run depends
// on this kernel.  A wait of more than 50 ms (it never should take 50 us) poisons the result with NaNs instead of hanging.
constexpr int kBcMaxWgs = 192;
template <typename T, int U>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 8))) void bn_bwd_coop_kernel(const char* __restrict__ dy, const char* __restrict__ yact,
                                                         const char* __restrict__ z, const float* __restrict__ save,
                                                         float* __restrict__ red, int R, const float* __restrict__ gamma,
                                                         char* __restrict__ dz, float* __restrict__ dgamma,
                                                         float* __restrict__ dbeta, int N, int HW, int C, int groups, int act,
                                                         unsigned* __restrict__ bar) {
  using TR = ElemTraits<T>;
  constexpr int E = TR::kVec;
  __shared__ float sh[256 * 2 * E];
  __shared__ float tot[2 * 256];
  const int vpp = C / E;
  const int vec = threadIdx.x % vpp, prow = threadIdx.x / vpp, rows = 256 / vpp;
  const int grp = blockIdx.y;
  const long long npix = (long long)(N / groups) * HW;
  const float inv_cnt = 1.f / (float)npix;
  const size_t rblock = (size_t)groups * 2 * C;
  const long long base = (long long)grp * npix;
  const long long step = (long long)gridDim.x * rows;
  const long long p = (long long)blockIdx.x * rows + prow;
  u32x4 rd[U], rz[U], ra[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {   // every load of the launch goes out before anything is used (raw, from clamped addresses)
    const long long pix = p + u * step < npix ? p + u * step : npix - 1;
    const long long off = ((base + pix) * C + vec * E) * TR::kBytes;
    rd[u] = TG_LDG16(dy + off);
    rz[u] = TG_LDG16(z + off);
    if (act == TG_ACT_LRELU) ra[u] = TG_LDG16(yact + off);
  }
  float mean[E], invstd[E], k0[E], s1[E], s2[E];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int c = vec * E + e;
    mean[e] = save[(grp * 2 + 0) * C + c];
    invstd[e] = save[(grp * 2 + 1) * C + c];
    k0[e] = gamma[c] * invstd[e];
    s1[e] = s2[e] = 0.f;
  }
  unsigned pos[U];   // LeakyReLU: bit e = activation e was positive (the raw yact vectors do not survive the wait: registers)
#pragma unroll
  for (int u = 0; u < U; ++u) {
    pos[u] = 0xffffffffu;
    if (act == TG_ACT_LRELU) {
      float a[E];
      Vec<T>::load(&ra[u], a);
      pos[u] = 0;
#pragma unroll
      for (int e = 0; e < E; ++e) pos[u] |= (a[e] > 0.f ? 1u : 0u) << e;
    }
    if (p + u * step < npix) {
      float d[E], zz[E];
      Vec<T>::load(&rd[u], d);
      Vec<T>::load(&rz[u], zz);
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const float dd = d[e] * ((pos[u] >> e) & 1u ? 1.f : 0.2f);
        s1[e] += dd;
        s2[e] += dd * (zz[e] - mean[e]) * invstd[e];
      }
    }
  }
#pragma unroll
  for (int e = 0; e < E; ++e) {
    sh[(threadIdx.x * 2 + 0) * E + e] = s1[e];
    sh[(threadIdx.x * 2 + 1) * E + e] = s2[e];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += 256) {
    const int which = i / C, c = i % C;
    const int v = c / E, e = c % E;
    float s = 0.f;
    for (int r = 0; r < rows; ++r) s += sh[((r * vpp + v) * 2 + which) * E + e];
    atomicAdd(red + (size_t)(blockIdx.x & (R - 1)) * rblock + (grp * 2 + which) * C + c, s);
  }
  // ---- grid-wide wait: this workgroup's sums are out (fence), one arrival per workgroup, thread 0 polls
  __threadfence();
  __syncthreads();
  bool late = false;
  if (threadIdx.x == 0) {
    const unsigned nwg = gridDim.x * gridDim.y;
    __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();   // 100 MHz
    while (__hip_atomic_load(bar, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < nwg) {
      __builtin_amdgcn_s_sleep(2);
      if (__builtin_amdgcn_s_memrealtime() - t0 > 5000000ull) { late = true; break; }
    }
    tot[0] = late ? 1.f : 0.f;
  }
  __syncthreads();
  late = tot[0] != 0.f;
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += 256) {   // fold the replica blocks (agent-scope loads: other XCDs' atomics)
    float s = 0.f;
    for (int r = 0; r < R; ++r)
      s += __hip_atomic_load(red + (size_t)r * rblock + (size_t)grp * 2 * C + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    tot[i] = late ? __builtin_nanf("") : s;
  }
  __syncthreads();
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < C) {   // single writer of the parameter gradients (all groups)
    const int c = threadIdx.x;
    float dg = 0.f, db = 0.f;
    for (int g2 = 0; g2 < groups; ++g2)
      for (int r = 0; r < R; ++r) {
        db += __hip_atomic_load(red + (size_t)r * rblock + (g2 * 2 + 0) * C + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        dg += __hip_atomic_load(red + (size_t)r * rblock + (g2 * 2 + 1) * C + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    dgamma[c] += dg;
    dbeta[c] += db;
  }
  float m1[E], m2[E];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    m1[e] = tot[vec * E + e] * inv_cnt;
    m2[e] = tot[C + vec * E + e] * inv_cnt;
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    if (p + u * step < npix) {
      float d[E], zz[E];
      Vec<T>::load(&rd[u], d);
      Vec<T>::load(&rz[u], zz);
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const float dd = d[e] * ((pos[u] >> e) & 1u ? 1.f : 0.2f);
        const float xh = (zz[e] - mean[e]) * invstd[e];
        d[e] = k0[e] * (dd - m1[e] - xh * m2[e]);
      }
      Vec<T>::store(dz + ((base + p + u * step) * C + vec * E) * TR::kBytes, d);
    }
  }
}


#endif  // TG_EXPERIMENTS

#ifdef TG_EXPERIMENTS   // measured slower than the two coalesced launches (profiles/r03_l_bn_bwd_fused_ab.log)
// Batch-norm backward of a SMALL tensor (<= kBfThreads * kBfTrips pixels per group) in ONE launch: workgroup v owns the E
// channels of 16-byte piece v of every pixel, keeps its share of dy / z (/ yact) in registers between the reduction and the
// apply, and is the only writer of those channels' dgamma / dbeta.  The two-launch path (tg_bn_bwd_reduce + tg_bn_bwd_apply)
// costs two ~6 us launches on the discriminator's 16x16 ... 4x4 layers whatever the tensor size (<= 0.8 MB: all of it latency);
// here the tensors are read once (16-byte pieces, one 2*C-byte row apart: L2-resident) and nothing goes through atomics.
constexpr int kBfThreads = 1024, kBfTrips = 4;
template <typename T>
__global__ __launch_bounds__(kBfThreads) void bn_bwd_fused_kernel(const char* __restrict__ dy, const char* __restrict__ yact,
                                                                  const char* __restrict__ z, const float* __restrict__ save,
                                                                  const float* __restrict__ gamma, char* __restrict__ dz,
                                                                  float* __restrict__ dgamma, float* __restrict__ dbeta, int N,
                                                                  int HW, int C, int groups, int act) {
  using TR = ElemTraits<T>;
  constexpr int E = TR::kVec, NW = kBfThreads / 64;
  __shared__ float sh[NW][2 * E];
  __shared__ float tot[2 * E];
  const int vec = blockIdx.x, grp = blockIdx.y;
  const int npix = (N / groups) * HW;
  const long long base = (long long)grp * npix;
  u32x4 rd[kBfTrips], rz[kBfTrips], ra[kBfTrips];
#pragma unroll
  for (int u = 0; u < kBfTrips; ++u) {  // every load of the launch goes out before anything is used
    const int pix = threadIdx.x + u * kBfThreads;
    const long long off = ((base + (pix < npix ? pix : npix - 1)) * C + vec * E) * TR::kBytes;
    rd[u] = TG_LDG16(dy + off);
    rz[u] = TG_LDG16(z + off);
    if (act == TG_ACT_LRELU) ra[u] = TG_LDG16(yact + off);
  }
  float mean[E], invstd[E], k0[E], s[2 * E];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int c = vec * E + e;
    mean[e] = save[(grp * 2 + 0) * C + c];
    invstd[e] = save[(grp * 2 + 1) * C + c];
    k0[e] = gamma[c] * invstd[e];
    s[e] = s[E + e] = 0.f;
  }
  float dd[kBfTrips][E], xh[kBfTrips][E];
#pragma unroll
  for (int u = 0; u < kBfTrips; ++u) {
    float zz[E], a[E];
    Vec<T>::load(&rd[u], dd[u]);
    Vec<T>::load(&rz[u], zz);
    if (act == TG_ACT_LRELU) Vec<T>::load(&ra[u], a);
    const bool live = threadIdx.x + u * kBfThreads < npix;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      if (act == TG_ACT_LRELU) dd[u][e] *= (a[e] > 0.f ? 1.f : 0.2f);
      xh[u][e] = (zz[e] - mean[e]) * invstd[e];
      if (live) {
        s[e] += dd[u][e];
        s[E + e] += dd[u][e] * xh[u][e];
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 2 * E; ++i) s[i] = wave_sum(s[i]);
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int i = 0; i < 2 * E; ++i) sh[threadIdx.x >> 6][i] = s[i];
  }
  __syncthreads();
  if (threadIdx.x < 2 * E) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) t += sh[w][threadIdx.x];
    tot[threadIdx.x] = t;
    // sole owner of these channels; the groups (the reference's D calls) of one launch add in turn
    float* dst = (threadIdx.x < E ? dbeta : dgamma) + vec * E + (threadIdx.x < E ? threadIdx.x : threadIdx.x - E);
    if (groups == 1) *dst += t; else atomicAdd(dst, t);
  }
  __syncthreads();
  const float inv_cnt = 1.f / (float)npix;
#pragma unroll
  for (int u = 0; u < kBfTrips; ++u) {
    const int pix = threadIdx.x + u * kBfThreads;
    if (pix < npix) {
      float o[E];
#pragma unroll
      for (int e = 0; e < E; ++e) o[e] = k0[e] * (dd[u][e] - tot[e] * inv_cnt - xh[u][e] * tot[E + e] * inv_cnt);
      Vec<T>::store(dz + ((base + pix) * C + vec * E) * TR::kBytes, o);
    }
  }
}

#endif  // TG_EXPERIMENTS

template <typename T>
__global__ void fc_head_fwd_kernel(const char* __restrict__ feat, const float* __restrict__ w,
                                   const float* __restrict__ b, float* __restrict__ prob, int N, int HW, int C, int Cp) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float s = b[0];
  for (int c = 0; c < C; ++c)
    for (int p = 0; p < HW; ++p) s += w[c * HW + p] * load_elem<T>(feat, ((long long)n * HW + p) * Cp + c);
  prob[n] = 1.f / (1.f + __expf(-s));
}

template <typename T>
__global__ void fc_head_bwd_kernel(const char* __restrict__ feat, const float* __restrict__ w,
                                   const float* __restrict__ dlogit, char* __restrict__ dfeat, float* __restrict__ dw,
                                   float* __restrict__ db, int N, int HW, int C, int Cp) {
  // single block; feature gradient for every (n, p, c) including zeroed padding channels
  for (int i = threadIdx.x; i < N * HW * Cp; i += blockDim.x) {
    const int c = i % Cp, p = (i / Cp) % HW, n = i / (Cp * HW);
    store_elem<T>(dfeat, i, c < C ? dlogit[n] * w[c * HW + p] : 0.f);
  }
  for (int i = threadIdx.x; i < C * HW; i += blockDim.x) {
    const int c = i / HW, p = i % HW;
    float s = 0.f;
    for (int n = 0; n < N; ++n) s += dlogit[n] * load_elem<T>(feat, ((long long)n * HW + p) * Cp + c);
    dw[i] += s;
  }
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int n = 0; n < N; ++n) s += dlogit[n];
    db[0] += s;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void absdiff_sum_kernel(const char* __restrict__ a, const char* __restrict__ b,
                                                         float* __restrict__ acc, long long npix, int C, int Cp) {
  using TR = ElemTraits<T>;
  constexpr int E = TR::kVec;
  __shared__ float sh[4];
  const int vpp = Cp / E;
  const long long total = npix * vpp;
  float s = 0.f;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += 256LL * gridDim.x) {
    const int vec = (int)(i % vpp);
    float x[E], y[E];
    Vec<T>::load(a + i * 16, x);
    Vec<T>::load(b + i * 16, y);
#pragma unroll
    for (int e = 0; e < E; ++e)
      if (vec * E + e < C) s += fabsf(x[e] - y[e]);
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(acc, sh[0] + sh[1] + sh[2] + sh[3]);
}

// Several absdiff sums in ONE launch (the four layer losses at the end of the discriminator's fake-half forward, which sits on
// the step's critical path: four 7-us launches for 5 MB each).  blockIdx.y selects the job: 6 x int64 = a, b, acc, npix, C, Cp.
template <typename T>
__global__ __launch_bounds__(256) void absdiff_sum_multi_kernel(const long long* __restrict__ jobs) {
  using TR = ElemTraits<T>;
  constexpr int E = TR::kVec;
  __shared__ float sh[4];
  const long long* j = jobs + 6 * blockIdx.y;
  const char* a = reinterpret_cast<const char*>(j[0]);
  const char* b = reinterpret_cast<const char*>(j[1]);
  float* acc = reinterpret_cast<float*>(j[2]);
  const int C = (int)j[4], vpp = (int)j[5] / E;
  const long long total = j[3] * vpp;
  float s = 0.f;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += 256LL * gridDim.x) {
    const int vec = (int)(i % vpp);
    float x[E], y[E];
    Vec<T>::load(a + i * 16, x);
    Vec<T>::load(b + i * 16, y);
#pragma unroll
    for (int e = 0; e < E; ++e)
      if (vec * E + e < C) s += fabsf(x[e] - y[e]);
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(acc, sh[0] + sh[1] + sh[2] + sh[3]);
}

// 1024 threads per block: the block count - and with it the number of atomics that serialise on the same four words at the
// end - is a quarter of what 256-thread blocks give for the same number of threads in flight
constexpr int kClThreads = 1024;
template <typename T>
__global__ __launch_bounds__(kClThreads) void content_loss_kernel(const float* __restrict__ gen, const float* __restrict__ y,
                                                          char* __restrict__ dpre, float* __restrict__ acc, int B,
                                                          int T_, int H, int W, float gscale, int t0, int t1, int pp_T,
                                                          float pp_coef, const float* __restrict__ loss_scale,
                                                          float* __restrict__ bias_acc, int dpre_c) {
  using TR = ElemTraits<T>;
  constexpr int NW = kClThreads / 64;
  if (loss_scale) {  // fp16 mode: every backward seed carries the dynamic loss scale (tg_adam_scaled divides it out)
    gscale *= *loss_scale;
    pp_coef *= *loss_scale;
  }
  const long long HW = (long long)H * W;
  const long long total = (long long)B * (t1 - t0) * HW;  // frames [t0,t1); dpre holds only those, frame-major
  float s = 0.f, cs[3] = {0.f, 0.f, 0.f}, pps = 0.f;
  for (long long i = blockIdx.x * (long long)kClThreads + threadIdx.x; i < total; i += (long long)kClThreads * gridDim.x) {
    const long long pos = i % HW;
    const long long r = i / HW;
    const int b = (int)(r % B);
    const int t = t0 + (int)(r / B);  // destination order is (t, b): the batched backward sees frame-major samples
    const long long src = ((long long)b * T_ + t) * 3 * HW + pos;
    // ping-pong term (code/train.py:275-283): frames t and 2(T-1)-t of the doubled sequence should agree; the loss
    // mean|gen_t - gen_partner| enters the generator loss with weight 2*pp_scaling (aliased gen_loss/fnet_loss tensors)
    const bool pp = pp_T > 0 && t != pp_T - 1;
    const long long psrc = pp ? ((long long)b * T_ + (2 * (pp_T - 1) - t)) * 3 * HW + pos : 0;
    float v[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) v[c] = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float g = gen[src + c * HW], d = g - y[src + c * HW];
      s += d * d;
      float dg = gscale * 2.f * d;
      if (pp) {
        const float e = g - gen[psrc + c * HW];
        dg += pp_coef * (e > 0.f ? 1.f : (e < 0.f ? -1.f : 0.f));
        if (t < pp_T - 1) pps += fabsf(e);  // every pair once
      }
      v[c] = dg * g * (1.f - g);
      cs[c] += v[c];
    }
    if (dpre && dpre_c == 4) {  // compact rows for tg_conv3x3_rgb_bwd: 3 channels + 1 pad, 16-bit
      if constexpr (!std::is_same<T, F32>::value) {
        uint2 pk;
        pk.x = (unsigned)f32_to_bits16<T>(v[0]) | ((unsigned)f32_to_bits16<T>(v[1]) << 16);
        pk.y = (unsigned)f32_to_bits16<T>(v[2]);
        *reinterpret_cast<uint2*>(dpre + i * 8) = pk;
      }
    } else if (dpre) {
      char* o = dpre + i * 32 * TR::kBytes;
#pragma unroll
      for (int k = 0; k < 32 / TR::kVec; ++k) Vec<T>::store(o + k * 16, v + k * TR::kVec);
    }
  }
  // block-level reduction, then ONE atomic per block and quantity: thousands of waves adding to the same four words
  // serialise at the memory side (measured: 428 us with per-wave atomics)
  __shared__ float shc[5][NW];  // rows: channel sums 0-2, ping-pong sum, squared error
  s = wave_sum(s);
  pps = wave_sum(pps);
#pragma unroll
  for (int c = 0; c < 3; ++c) cs[c] = wave_sum(cs[c]);
  if ((threadIdx.x & 63) == 0) {
    shc[4][threadIdx.x >> 6] = s;
    shc[3][threadIdx.x >> 6] = pps;
#pragma unroll
    for (int c = 0; c < 3; ++c) shc[c][threadIdx.x >> 6] = cs[c];
  }
  __syncthreads();
  if (threadIdx.x < 5) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) t += shc[threadIdx.x][w];
    if (threadIdx.x == 4) atomicAdd(acc, t);
    else if (threadIdx.x == 3) { if (pp_T > 0) atomicAdd(acc + 6, t); }
    else atomicAdd(bias_acc + threadIdx.x, t);  // output-layer bias gradient
  }
}

// cfg layout (floats): 0 content_div, 1 warp_div, 2..5 layer_div, 6 EPS, 7 ratio, 8 dt_ratio, 9 use_layerloss,
//                      10 pp_div (0 = no pingpang), 11 pp_scaling, 12..15 layer_norm
//   cfg[9] is a flag word: bit 0 D_LAYERLOSS, bit 1 VGG feature loss.  With bit 1 the block continues behind the two Adam
//   hyper-parameter rows of the step's parameter buffer: cfg[32] vgg_scaling, cfg[33..35] pixels per VGG layer.
// acc layout: 0 content sumsq, 1 warp sumsq, 2..5 layer absdiff sums, 6 pingpang abs sum, (8..10 output-bias gradient),
//             11..13 sum of per-pixel cosines of the three VGG layers
// scalars out (64 floats): 0..3 layer losses, 4 layer_sum, 5 gen_loss total (aliased tensor), 6 warp loss, 7 t_adv,
//              8 d_loss, 9 mean p_real, 10 mean p_fake, 11 content, 12 t_balance, 13 pingpang, 14 tb, 15 len(update_list),
//              16..39 update_list, 40..63 update_list_avg
__global__ void loss_finalize_kernel(const float* __restrict__ prob, const float* __restrict__ acc,
                                     float* __restrict__ sc, float* __restrict__ dlogit, int tb,
                                     const float* __restrict__ cfg, const float* __restrict__ loss_scale) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const float eps = cfg[6];
  const float S = loss_scale ? *loss_scale : 1.f;
  const int flags = (int)cfg[9];
  const bool layerloss = flags & 1, vgg = flags & 2;
  float layer_sum = 0.f;
  for (int i = 0; i < 4; ++i) {
    const float l = layerloss ? acc[2 + i] / cfg[2 + i] : 0.f;
    sc[i] = l;
    layer_sum += 0.02f * l / cfg[12 + i];
  }
  sc[4] = layer_sum;
  const float content = acc[0] / cfg[0];
  sc[11] = content;
  sc[6] = acc[1] / cfg[1];
  float t_adv = 0.f, d_loss = 0.f, mr = 0.f, mf = 0.f, real_l = 0.f;
  const float inv = 1.f / (float)tb;
  for (int n = 0; n < tb; ++n) {
    const float pr = prob[n], pf = prob[tb + n];
    t_adv += -logf(pf + eps);
    const float lr_ = logf(pr + eps), lf_ = logf(1.f - pf + eps);
    d_loss += -(lf_ + lr_);
    real_l += lr_;
    mr += pr;
    mf += pf;
    dlogit[n] = -S * inv * (1.f / (pr + eps)) * pr * (1.f - pr);
    dlogit[tb + n] = S * inv * (1.f / (1.f - pf + eps)) * pf * (1.f - pf);
  }
  t_adv *= inv; d_loss *= inv; mr *= inv; mf *= inv; real_l *= inv;
  float total = content;
  // VGG terms (code/train.py:253-273 as fixed in DESIGN.md): per layer 1 - mean cosine; the sum enters the aliased
  // gen_loss / fnet_loss tensor twice (once through each name, code/train.py:268-269)
  float vl[3] = {0.f, 0.f, 0.f}, vgg_all = 0.f;
  if (vgg) {
    for (int i = 0; i < 3; ++i) {
      vl[i] = 1.f - acc[11 + i] / cfg[33 + i];
      vgg_all += vl[i];
    }
    total += 2.f * cfg[32] * vgg_all;
  }
  float pp = 0.f;
  if (cfg[10] != 0.f) {
    pp = acc[6] / cfg[10];
    if (cfg[11] > 0.f) total += 2.f * pp * cfg[11];
  }
  sc[13] = pp;
  total += 2.f * cfg[7] * t_adv;
  if (layerloss) total += layer_sum * cfg[8];
  sc[5] = total; sc[7] = t_adv; sc[8] = d_loss; sc[9] = mr; sc[10] = mf;
  sc[12] = real_l + t_adv;
  sc[14] = 0.99f * sc[12];  // tb: a fresh EMA(0.99) seeded with zero every call (code/train.py:324-327)
  // update_list in the reference's order and its running average avg_k = 0.99*u_k + 0.01*avg_{k-1} (code/train.py:329-333):
  // sc[16..] = update_list values, sc[40..] = update_list_avg
  int n = 0;
  float* ul = sc + 16;
  if (layerloss) { for (int i = 0; i < 5; ++i) ul[n++] = sc[i]; }
  ul[n++] = total; ul[n++] = sc[6];
  if (vgg) { for (int i = 0; i < 3; ++i) ul[n++] = vl[i]; ul[n++] = vgg_all; }
  if (cfg[10] != 0.f) ul[n++] = pp;
  ul[n++] = t_adv; ul[n++] = d_loss; ul[n++] = mr; ul[n++] = mf; ul[n++] = total;
  float shadow = 0.f;
  for (int i = 0; i < n; ++i) { shadow = 0.99f * ul[i] + 0.01f * shadow; sc[40 + i] = shadow; }
  sc[15] = (float)n;
}

// hyper (device): 0 lr, 1 beta1, 2 beta2, 3 eps, 4 1-beta1^t, 5 1-beta2^t, 6 grad scale (1/world for data parallel), 7 t.
// Read from memory, not kernel arguments, so that a captured hipGraph replays with the current step's values.
// scaler (fp16 mode, else null): the dynamic loss-scale state of tg_scaler_update; found_inf[which] != 0 skips the whole
// update (torch.cuda.amp.GradScaler.step, code/train.py:336-341) and the gradients are divided by the scale.
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, long long n, const float* __restrict__ hyper,
                            const float* __restrict__ scaler, int which) {
  if (scaler && scaler[2 + which] != 0.f) return;
  const float lr = hyper[0], b1 = hyper[1], b2 = hyper[2], eps = hyper[3],
              gscale = scaler ? hyper[6] * scaler[4] : hyper[6];
  float bc1 = hyper[4], bc2 = hyper[5];
  // fp16 mode: GradScaler.step() does not call optimizer.step() on overflow, so torch's step count - and with it the bias
  // corrections - does not advance on a skipped update.  The host counts calls (hyper[7]); the skips are counted on the
  // device (scaler[5 + which], tg_scaler_update), and the corrections are re-derived here for the steps actually taken.
  if (scaler && scaler[5 + which] != 0.f) {
    const double t = (double)hyper[7] - (double)scaler[5 + which];
    bc1 = (float)(1.0 - pow((double)b1, t));
    bc2 = (float)(1.0 - pow((double)b2, t));
  }
  const float step = lr / bc1, sq2 = sqrtf(bc2);
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float gi = g[i] * gscale;
    const float mi = m[i] + (gi - m[i]) * (1.f - b1);        // exp_avg.lerp_(grad, 1-beta1)
    const float vi = v[i] * b2 + (1.f - b2) * gi * gi;       // exp_avg_sq.mul_(beta2).addcmul_(g, g, 1-beta2)
    m[i] = mi;
    v[i] = vi;
    p[i] -= step * (mi / (sqrtf(vi) / sq2 + eps));              // denom = sqrt(v)/sqrt(bc2) + eps
  }
}

inline int grid_for(long long total, int per_block, int cap) {
  return (int)std::max<long long>(1, std::min<long long>((total + per_block - 1) / per_block, cap));
}

// A/B hook (profiles/r05_zz_bn_cap_ab.log): TECOGAN_BN_WGS caps the batch-norm launches' grids (0 / unset: the built-in caps)
inline int bn_grid_cap(int builtin) {
  static const int cap = [] { const char* e = getenv("TECOGAN_BN_WGS"); return e ? atoi(e) : 0; }();
  return cap > 0 && cap < builtin ? cap : builtin;
}

inline bool bn_shape_ok(int dtype, int C) {
  const int e = dtype == TG_F32 ? 4 : 8;
  const int vpp = C / e;
  return C % 32 == 0 && vpp <= 256 && (256 % vpp) == 0 && C <= 256;
}

}  // namespace

#define TG_DISPATCH(dtype, KERNEL, grid, block, st, ...)                                         \
  do {                                                                                           \
    if ((dtype) == TG_BF16) hipLaunchKernelGGL(KERNEL<BF16>, grid, block, 0, st, __VA_ARGS__);   \
    else if ((dtype) == TG_F16) hipLaunchKernelGGL(KERNEL<F16>, grid, block, 0, st, __VA_ARGS__); \
    else if ((dtype) == TG_F32) hipLaunchKernelGGL(KERNEL<F32>, grid, block, 0, st, __VA_ARGS__); \
    else return TG_E_BADARG;                                                                     \
  } while (0)

extern "C" int tg_bn_apply(int dtype, const void* z, const float* stats, int stats_replicas, const float* gamma, const float* beta,
                           const void* skip, void* y, float* running_mean, float* running_var, float* save, int N,
                           int HW, int C, int groups, int act, float eps, float momentum, int64_t* num_batches_tracked,
                           void* stream) {
  if (!z || !stats || !gamma || !beta || !y || !save || N <= 0 || HW <= 0 || groups <= 0 || N % groups) return TG_E_BADARG;
  if (stats_replicas < 1 || (stats_replicas & (stats_replicas - 1))) return TG_E_BADARG;
  if (!bn_shape_ok(dtype, C)) return TG_E_UNSUPPORTED;
  const int rows = 256 / (C / (dtype == TG_F32 ? 4 : 8));
  dim3 grid(grid_for((long long)(N / groups) * HW, rows * 4, bn_grid_cap(1024)), groups);
  TG_DISPATCH(dtype, bn_apply_kernel, grid, dim3(256), (hipStream_t)stream, (const char*)z, stats, stats_replicas, gamma, beta,
              (const char*)skip, (char*)y, running_mean, running_var, save, N, HW, C, groups, act, eps, momentum,
              (long long*)num_batches_tracked);
  return tg_launch_status();
}

extern "C" int tg_bn_bwd_reduce(int dtype, const void* dy, const void* yact, const void* z, const float* save,
                                float* red, int red_replicas, int N, int HW, int C, int groups, int act, void* stream) {
  if (!dy || !z || !save || !red || N <= 0 || HW <= 0 || groups <= 0 || N % groups) return TG_E_BADARG;
  if (red_replicas < 1 || (red_replicas & (red_replicas - 1))) return TG_E_BADARG;
  if (act == TG_ACT_LRELU && !yact) return TG_E_BADARG;
  if (!bn_shape_ok(dtype, C)) return TG_E_UNSUPPORTED;
  const int rows = 256 / (C / (dtype == TG_F32 ? 4 : 8));
  dim3 grid(grid_for((long long)(N / groups) * HW, rows * 8, bn_grid_cap(512)), groups);  // two trips of four pixels per thread
  TG_DISPATCH(dtype, bn_bwd_reduce_kernel, grid, dim3(256), (hipStream_t)stream, (const char*)dy, (const char*)yact,
              (const char*)z, save, red, red_replicas, N, HW, C, groups, act);
  return tg_launch_status();
}

extern "C" int tg_bn_bwd_apply(int dtype, const void* dy, const void* yact, const void* z, const float* save,
                               const float* red, int red_replicas, const float* gamma, void* dz, float* dgamma, float* dbeta,
                               int N, int HW, int C, int groups, int act, int red_raw, void* stream) {
  if (!dy || !z || !save || !red || !gamma || !dz || !dgamma || !dbeta || N <= 0 || HW <= 0 || groups <= 0 || N % groups)
    return TG_E_BADARG;
  if (act == TG_ACT_LRELU && !yact) return TG_E_BADARG;
  if (red_replicas < 1 || (red_replicas & (red_replicas - 1))) return TG_E_BADARG;
  if (!bn_shape_ok(dtype, C)) return TG_E_UNSUPPORTED;
  const int rows = 256 / (C / (dtype == TG_F32 ? 4 : 8));
  dim3 grid(grid_for((long long)(N / groups) * HW, rows * 4, bn_grid_cap(1024)), groups);
  TG_DISPATCH(dtype, bn_bwd_apply_kernel, grid, dim3(256), (hipStream_t)stream, (const char*)dy, (const char*)yact,
              (const char*)z, save, red, red_replicas, gamma, (char*)dz, dgamma, dbeta, N, HW, C, groups, act, red_raw);
  return tg_launch_status();
}

#ifdef TG_EXPERIMENTS
// reduce + wait + apply in one launch; TG_E_UNSUPPORTED when the tensor needs more than kBcMaxWgs workgroups (the caller then
// takes tg_bn_bwd_reduce + tg_bn_bwd_apply).  `barrier` is one zeroed 32-bit word per call (the step's accumulator arena).
extern "C" int tg_bn_bwd_coop_max_workgroups(void) { return kBcMaxWgs; }

extern "C" int tg_bn_bwd_coop(int dtype, const void* dy, const void* yact, const void* z, const float* save, float* red,
                              int red_replicas, const float* gamma, void* dz, float* dgamma, float* dbeta, int N, int HW, int C,
                              int groups, int act, unsigned* barrier, void* stream) {
  if (!dy || !z || !save || !red || !gamma || !dz || !dgamma || !dbeta || !barrier || N <= 0 || HW <= 0 || groups <= 0 || N % groups)
    return TG_E_BADARG;
  if (act == TG_ACT_LRELU && !yact) return TG_E_BADARG;
  if (red_replicas < 1 || (red_replicas & (red_replicas - 1))) return TG_E_BADARG;
  if (!bn_shape_ok(dtype, C)) return TG_E_UNSUPPORTED;
  const int rows = 256 / (C / (dtype == TG_F32 ? 4 : 8));
  const long long npix = (long long)(N / groups) * HW;
  const long long g4 = (npix + rows * 4 - 1) / (rows * 4), g8 = (npix + rows * 8 - 1) / (rows * 8);
  if (g4 * groups <= kBcMaxWgs) {
    dim3 grid((unsigned)g4, groups);
    if (dtype == TG_BF16) hipLaunchKernelGGL((bn_bwd_coop_kernel<BF16, 4>), grid, dim3(256), 0, (hipStream_t)stream, (const char*)dy, (const char*)yact, (const char*)z, save, red, red_replicas, gamma, (char*)dz, dgamma, dbeta, N, HW, C, groups, act, barrier);
    else if (dtype == TG_F16) hipLaunchKernelGGL((bn_bwd_coop_kernel<F16, 4>), grid, dim3(256), 0, (hipStream_t)stream, (const char*)dy, (const char*)yact, (const char*)z, save, red, red_replicas, gamma, (char*)dz, dgamma, dbeta, N, HW, C, groups, act, barrier);
    else if (dtype == TG_F32) hipLaunchKernelGGL((bn_bwd_coop_kernel<F32, 4>), grid, dim3(256), 0, (hipStream_t)stream, (const char*)dy, (const char*)yact, (const char*)z, save, red, red_replicas, gamma, (char*)dz, dgamma, dbeta, N, HW, C, groups, act, barrier);
    else return TG_E_BADARG;
  } else if (g8 * groups <= kBcMaxWgs) {
    dim3 grid((unsigned)g8, groups);
    if (dtype == TG_BF16) hipLaunchKernelGGL((bn_bwd_coop_kernel<BF16, 8>), grid, dim3(256), 0, (hipStream_t)stream, (const char*)dy, (const char*)yact, (const char*)z, save, red, red_replicas, gamma, (char*)dz, dgamma, dbeta, N, HW, C, groups, act, barrier);
    else if (dtype == TG_F16) hipLaunchKernelGGL((bn_bwd_coop_kernel<F16, 8>), grid, dim3(256), 0, (hipStream_t)stream, (const char*)dy, (const char*)yact, (const char*)z, save, red, red_replicas, gamma, (char*)dz, dgamma, dbeta, N, HW, C, groups, act, barrier);
    else if (dtype == TG_F32) hipLaunchKernelGGL((bn_bwd_coop_kernel<F32, 8>), grid, dim3(256), 0, (hipStream_t)stream, (const char*)dy, (const char*)yact, (const char*)z, save, red, red_replicas, gamma, (char*)dz, dgamma, dbeta, N, HW, C, groups, act, barrier);
    else return TG_E_BADARG;
  } else {
    return TG_E_UNSUPPORTED;
  }
  return tg_launch_status();
}

#endif  // TG_EXPERIMENTS

#ifdef TG_EXPERIMENTS
extern "C" int tg_bn_bwd_fused_max_pixels(void) { return kBfThreads * kBfTrips; }

extern "C" int tg_bn_bwd_fused(int dtype, const void* dy, const void* yact, const void* z, const float* save,
                               const float* gamma, void* dz, float* dgamma, float* dbeta, int N, int HW, int C, int groups,
                               int act, void* stream) {
  if (!dy || !z || !save || !gamma || !dz || !dgamma || !dbeta || N <= 0 || HW <= 0 || groups <= 0 || N % groups)
    return TG_E_BADARG;
  if (act == TG_ACT_LRELU && !yact) return TG_E_BADARG;
  if (act != TG_ACT_NONE && act != TG_ACT_LRELU) return TG_E_UNSUPPORTED;
  if (!bn_shape_ok(dtype, C)) return TG_E_UNSUPPORTED;
  if ((long long)(N / groups) * HW > (long long)kBfThreads * kBfTrips) return TG_E_UNSUPPORTED;
  dim3 grid(C / (dtype == TG_F32 ? 4 : 8), groups);
  TG_DISPATCH(dtype, bn_bwd_fused_kernel, grid, dim3(kBfThreads), (hipStream_t)stream, (const char*)dy, (const char*)yact,
              (const char*)z, save, gamma, (char*)dz, dgamma, dbeta, N, HW, C, groups, act);
  return tg_launch_status();
}

#endif  // TG_EXPERIMENTS

extern "C" int tg_fc_head_fwd(int dtype, const void* feat, const float* w, const float* b, float* prob, int N, int HW,
                              int C, int Cp, void* stream) {
  if (!feat || !w || !b || !prob || N <= 0 || HW <= 0 || C <= 0 || C > Cp) return TG_E_BADARG;
  TG_DISPATCH(dtype, fc_head_fwd_kernel, dim3((N + 63) / 64), dim3(64), (hipStream_t)stream, (const char*)feat, w, b,
              prob, N, HW, C, Cp);
  return tg_launch_status();
}

extern "C" int tg_fc_head_bwd(int dtype, const void* feat, const float* w, const float* dlogit, void* dfeat, float* dw,
                              float* db, int N, int HW, int C, int Cp, void* stream) {
  if (!feat || !w || !dlogit || !dfeat || !dw || !db || N <= 0 || HW <= 0 || C <= 0 || C > Cp) return TG_E_BADARG;
  TG_DISPATCH(dtype, fc_head_bwd_kernel, dim3(1), dim3(256), (hipStream_t)stream, (const char*)feat, w, dlogit,
              (char*)dfeat, dw, db, N, HW, C, Cp);
  return tg_launch_status();
}

extern "C" int tg_absdiff_sum(int dtype, const void* a, const void* b, float* acc, int64_t npix, int C, int Cp,
                              void* stream) {
  if (!a || !b || !acc || npix <= 0 || C <= 0 || C > Cp || Cp % 32) return TG_E_BADARG;
  const long long total = npix * (Cp / (dtype == TG_F32 ? 4 : 8));
  TG_DISPATCH(dtype, absdiff_sum_kernel, dim3(grid_for(total, 1024, 512)), dim3(256), (hipStream_t)stream,
              (const char*)a, (const char*)b, acc, (long long)npix, C, Cp);
  return tg_launch_status();
}

extern "C" int tg_absdiff_sum_multi(int dtype, const int64_t* jobs_dev, int njobs, int blocks_per_job, void* stream) {
  if (!jobs_dev || njobs <= 0 || blocks_per_job <= 0) return TG_E_BADARG;
  TG_DISPATCH(dtype, absdiff_sum_multi_kernel, dim3((unsigned)blocks_per_job, (unsigned)njobs), dim3(256), (hipStream_t)stream,
              (const long long*)jobs_dev);
  return tg_launch_status();
}

extern "C" int tg_content_loss(int dtype, const float* gen, const float* y, void* dpre, float* acc, int B, int T,
                               int H, int W, float gscale, int t0, int t1, int pp_T, float pp_coef,
                               const float* loss_scale, float* bias_acc, int dpre_channels, void* stream) {
  if (!gen || !y || !acc || B <= 0 || T <= 0 || H <= 0 || W <= 0 || t0 < 0 || t1 > T || t0 >= t1) return TG_E_BADARG;
  if (dpre_channels != 32 && !(dpre_channels == 4 && dtype != TG_F32)) return TG_E_UNSUPPORTED;
  if (pp_T != 0 && T != 2 * pp_T - 1) return TG_E_BADARG;  // ping-pong: the sequence is x followed by reverse(x)[1:]
  const long long total = (long long)B * (t1 - t0) * H * W;
  TG_DISPATCH(dtype, content_loss_kernel, dim3(grid_for(total, kClThreads, 256)), dim3(kClThreads), (hipStream_t)stream, gen, y,
              (char*)dpre, acc, B, T, H, W, gscale, t0, t1, pp_T, pp_coef, loss_scale, bias_acc ? bias_acc : acc + 8,
              dpre_channels);
  return tg_launch_status();
}

extern "C" int tg_loss_finalize(const float* prob, const float* acc, float* scalars, float* dlogit, int tb,
                                const float* cfg, const float* loss_scale, void* stream) {
  if (!prob || !acc || !scalars || !dlogit || !cfg || tb <= 0) return TG_E_BADARG;
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, prob, acc, scalars, dlogit, tb,
                     cfg, loss_scale);
  return tg_launch_status();
}

namespace {
__global__ void dlogit_real_kernel(const float* __restrict__ prob, float* __restrict__ dlogit, int tb,
                                   const float* __restrict__ cfg, const float* __restrict__ loss_scale) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= tb) return;
  const float eps = cfg[6], inv = 1.f / (float)tb, pr = prob[n];
  const float S = loss_scale ? *loss_scale : 1.f;
  dlogit[n] = -S * inv * (1.f / (pr + eps)) * pr * (1.f - pr);  // same expression as loss_finalize_kernel
}
}  // namespace

extern "C" int tg_dlogit_real(const float* prob, float* dlogit, int tb, const float* cfg, const float* loss_scale,
                              void* stream) {
  if (!prob || !dlogit || !cfg || tb <= 0) return TG_E_BADARG;
  hipLaunchKernelGGL(dlogit_real_kernel, dim3((tb + 63) / 64), dim3(64), 0, (hipStream_t)stream, prob, dlogit, tb, cfg,
                     loss_scale);
  return tg_launch_status();
}

extern "C" int tg_adam(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper_dev, void* stream) {
  if (!p || !g || !m || !v || !hyper_dev || n <= 0) return TG_E_BADARG;
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n, 1024, 2048)), dim3(256), 0, (hipStream_t)stream, p, g, m, v,
                     (long long)n, hyper_dev, (const float*)nullptr, 0);
  return tg_launch_status();
}

namespace {
// *flag = 1 when any of g[0..n) is inf or NaN (GradScaler's found_inf, torch/amp/grad_scaler.py unscale_)
__global__ void check_finite_kernel(const float* __restrict__ g, long long n, float* __restrict__ flag) {
  bool bad = false;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float x = g[i];
    bad |= !(fabsf(x) <= 3.402823466e38f);  // false for inf and NaN
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) *flag = 1.f;  // benign race: every writer stores the same value
}

// state: 0 scale, 1 growth tracker, 2 found_inf (generator), 3 found_inf (discriminator), 4 1/scale, 5 / 6 skipped updates of
// the generator / discriminator (subtracted from the host's call count in adam_kernel).  The reference shares
// ONE GradScaler between both optimisers and calls update() after each step() (code/train.py:9,337-341): two updates per
// training step, the generator's first.  update(): found_inf -> scale *= backoff, tracker = 0; else tracker += 1 and at
// `interval` scale *= growth, tracker = 0.
__global__ void scaler_update_kernel(float* __restrict__ s, float growth, float backoff, int interval) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float scale = s[0], tracker = s[1];
  for (int which = 0; which < 2; ++which) {
    if (s[2 + which] != 0.f) {
      scale *= backoff;
      tracker = 0.f;
      s[5 + which] += 1.f;  // updates this network has skipped so far (adam_kernel's step count)
    } else {
      tracker += 1.f;
      if (tracker >= (float)interval) {
        scale *= growth;
        tracker = 0.f;
      }
    }
    s[2 + which] = 0.f;
  }
  s[0] = scale;
  s[1] = tracker;
  s[4] = 1.f / scale;
}
}  // namespace

extern "C" int tg_adam_scaled(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper_dev,
                              const float* scaler_state, int which, void* stream) {
  if (!p || !g || !m || !v || !hyper_dev || !scaler_state || n <= 0 || which < 0 || which > 1) return TG_E_BADARG;
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n, 1024, 2048)), dim3(256), 0, (hipStream_t)stream, p, g, m, v,
                     (long long)n, hyper_dev, scaler_state, which);
  return tg_launch_status();
}

extern "C" int tg_check_finite(const float* g, int64_t n, float* flag, void* stream) {
  if (!g || !flag || n <= 0) return TG_E_BADARG;
  hipLaunchKernelGGL(check_finite_kernel, dim3(grid_for(n, 2048, 1024)), dim3(256), 0, (hipStream_t)stream, g, (long long)n,
                     flag);
  return tg_launch_status();
}

extern "C" int tg_scaler_update(float* scaler_state, float growth, float backoff, int interval, void* stream) {
  if (!scaler_state || growth < 1.f || backoff <= 0.f || backoff > 1.f || interval <= 0) return TG_E_BADARG;
  hipLaunchKernelGGL(scaler_update_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, scaler_state, growth, backoff, interval);
  return tg_launch_status();
}

namespace {
// 16 columns x 16 replica lanes per workgroup: the replica loop of one column is spread over 16 threads whose loads are
// independent (one thread per column walked <= 64 replicas serially: 22 us for 256 columns in ONE workgroup)
__global__ __launch_bounds__(256) void reduce_replicas_kernel(const float* __restrict__ src, int replicas, int stride, int n,
                                                              float* __restrict__ dst, int accumulate) {
  __shared__ float part[16][17];
  const int c = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int i = blockIdx.x * 16 + c;
  float s = 0.f;
  if (i < n)
    for (int r = rl; r < replicas; r += 16) s += src[(size_t)r * stride + i];
  part[rl][c] = s;
  __syncthreads();
  if (rl == 0 && i < n) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += part[k][c];
    dst[i] = accumulate ? dst[i] + t : t;
  }
}
}  // namespace

extern "C" int tg_reduce_replicas(const float* src, int replicas, int stride, int n, float* dst, int accumulate,
                                  void* stream) {
  if (!src || !dst || replicas <= 0 || n <= 0 || stride < n) return TG_E_BADARG;
  hipLaunchKernelGGL(reduce_replicas_kernel, dim3((n + 15) / 16), dim3(256), 0, (hipStream_t)stream, src, replicas,
                     stride, n, dst, accumulate);
  return tg_launch_status();
}

extern "C" int tg_abi_version(void) { return TG_ABI_VERSION; }
extern "C" int tg_has_experiments(void) { return kTgExperiments ? 1 : 0; }

extern "C" const char* tg_error_string(int code) {
  switch (code) {
    case TG_OK: return "ok";
    case TG_E_BADARG: return "invalid argument";
    case TG_E_UNSUPPORTED: return "unsupported shape";
    case TG_E_ALIGN: return "misaligned pointer or channel count not a multiple of 32";
  }
  return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown error";
}
