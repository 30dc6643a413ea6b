// The discriminator's TAIL in ONE launch per direction (round 6) - everything behind block4's 4x4 stride-2 convolution and in front of
// its input-gradient (/root/reference/code/models.py:119-123,137-146; code/ops.py:75-77,85-88; code/train.py:304-307):
//
//   forward   z4 -> BN(block4.1, batch statistics from the conv's epilogue) + LeakyReLU = n4 -> conv 4x4 s2 p1 64 -> 3 (block5.0) = z5
//             -> BN(block5.1) + LeakyReLU = n5 -> flatten (C, H, W order) -> Linear(3 hw, 1) -> sigmoid = prob
//   backward  [real half: d(loss)/d(logit) from prob] -> fc (dW, db, d n5) -> LeakyReLU', BN(block5.1) backward = d z5 -> block5.0's
//             input-gradient = d n4 -> LeakyReLU', BN(block4.1) backward = d z4                  (block4.0's input-gradient: conv4s2d_cw)
//
// As separate launches this was 5 (+1) and 6 kernels on 0.3 M elements: bn_apply, tg_conv (20 us for 1.2 MFLOP), bn_apply, fc head,
// [loss seed], fc backward, 2 x (bn_bwd_reduce, bn_bwd_apply), tg_conv - 58 + 44 us per half of the step, all of it launch latency.
// ONE WORKGROUP PER SAMPLE (a first version with a single workgroup for everything took 31 + 36 us: 0.6 MB through one CU's ~25 B/clk).
// A sample's block4 image (8 KB) stays in LDS between the BatchNorm and the convolution.  What couples the samples - block5.1's batch
// statistics, block4.1's backward sums - goes through a few floats of scratch and a TICKET: every workgroup publishes its part
// (__threadfence, atomic counter) and exits; the one that draws the last ticket finishes the job for all samples (BN(block5.1), fc and
// sigmoid: 192 x 3 values; in backward the d z4 sweep).  Nobody waits for anybody: no co-residency assumption, no spin.
// Every tensor another launch reads (z4 n4 z5 n5 d z5 d z4: the weight gradients' operands, the layer loss, block4.0's input-gradient)
// is written with the rounding points of the separate launches.  BatchNorm groups (two halves in one tensor) are handled side by side.
#include "common.h"
#include <type_traits>

namespace {

constexpr int kThreads = 512, kWaves = kThreads / 64;   // (eight waves of up to 256 registers: sixteen spilled)
constexpr int kWsHead = 32;                             // scratch: [0] the ticket, then the regions of ws_floats()

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  return v;
}
__device__ __forceinline__ float fold1(const float* acc, int R, size_t block) {   // replica blocks of a per-channel accumulator
  float s = 0.f;
  for (int r = 0; r < R; ++r) s += acc[r * block];
  return s;
}
template <typename T> __device__ __forceinline__ float rnd(float v);   // round to the element type, as a store + load would
template <> __device__ __forceinline__ float rnd<F32>(float v) { return v; }
template <> __device__ __forceinline__ float rnd<BF16>(float v) { return bf16_bits_to_f32(f32_to_bf16_bits(v)); }
template <> __device__ __forceinline__ float rnd<F16>(float v) { return f16_bits_to_f32(f32_to_f16_bits(v)); }

struct TailK {
  // tensors (NHWC, channels padded): z4 n4 dn4 dz4 [N][H4][H4][C4]; z5 n5 dz5 [N][H5][H5][Cp5]
  const char* z4; char* n4; char* z5; char* n5;
  const float *stats4, *gamma4, *beta4, *gamma5, *beta5, *w5, *fc_w, *fc_b;   // w5: fp32 master [C5][C4][4][4] (rounded here as the packer does)
  float *rm4, *rv4, *save4, *rm5, *rv5, *save5, *prob;
  long long *nbt4, *nbt5;
  int N, H4, C4, C5, Cp5, groups, R4;
  float eps, momentum;
  float* ws;   // scratch, zero when the launch starts and when it ends
  // backward only
  char *dn4, *dz4, *dz5;
  float *dlogit, *g_fc_w, *g_fc_b, *dgamma4, *dbeta4, *dgamma5, *dbeta5;
  const float *cfg, *loss_scale;
  int seed_real;
};

// LDS (dynamic): w5s [16][C5 * C4 + 4] floats (tap stride + 16 B: the 16 taps' rows start on 16 distinct 16-byte slots) | par [8][C4]
// floats (scale / shift / mean / invstd ...) | zb, nb [per * hw5][4] floats | red [waves][2][C4] floats | dl [per] floats | flag |
// the sample's n4 image [hw4][C4 elements + 16 B] (forward)
struct Lds {
  float *w5s, *par, *zb, *nb, *red, *dl;
  int* flag;
  char* img;
};
__device__ __forceinline__ Lds carve(char* smem, const TailK& p, int per) {
  Lds l;
  const int hw5 = (p.H4 / 2) * (p.H4 / 2);
  float* f = reinterpret_cast<float*>(smem);
  l.w5s = f; f += 16 * (p.C5 * p.C4 + 4);
  l.par = f; f += 8 * p.C4;
  l.zb = f; f += per * hw5 * 4;
  l.nb = f; f += per * hw5 * 4;
  l.red = f; f += kWaves * 2 * p.C4;
  l.dl = f; f += (per + 3) / 4 * 4;
  l.flag = reinterpret_cast<int*>(f); f += 4;
  l.img = reinterpret_cast<char*>(f);
  return l;
}
static size_t lds_bytes(int per, int H4, int C4, int C5, int elem_bytes) {
  const int hw5 = (H4 / 2) * (H4 / 2);
  return sizeof(float) * ((size_t)16 * (C5 * C4 + 4) + 8 * C4 + 2 * (size_t)per * hw5 * 4 + kWaves * 2 * C4 + (per + 3) / 4 * 4 + 4) +
         (size_t)H4 * H4 * (C4 * elem_bytes + 16);
}
static size_t ws_floats(int N, int H4, int groups) {
  // [0] the ticket | [kWsHead ...] block4.1's backward sums [groups][2][64] (zero between launches) | z5 as fp32 [N][hw5][4] (overwritten)
  return kWsHead + (size_t)groups * 2 * 64 + (size_t)N * (H4 / 2) * (H4 / 2) * 4;
}

template <typename T>
__device__ __forceinline__ void stage_w5(const TailK& p, const Lds& l) {   // w5s[tap][co][ci] = round(W[co][ci][ky][kx])
  // (reads the master in ITS order - consecutive threads, consecutive floats - and scatters into LDS)
  for (int i = threadIdx.x; i < 16 * p.C5 * p.C4; i += kThreads) {
    const int tap = i & 15, ci = (i >> 4) % p.C4, co = (i >> 4) / p.C4;
    l.w5s[tap * (p.C5 * p.C4 + 4) + co * p.C4 + ci] = rnd<T>(p.w5[i]);
  }
}

// publishes this workgroup's global writes and draws a ticket; true for the workgroup that drew the last one (which then sees every
// other workgroup's writes).  All threads call it.
__device__ __forceinline__ bool last_ticket(const TailK& p, const Lds& l) {
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) *l.flag = atomicAdd(reinterpret_cast<unsigned*>(p.ws), 1u) == (unsigned)(gridDim.x - 1) ? 1 : 0;
  __syncthreads();
  const bool last = *l.flag != 0;
  if (last) __threadfence();
  return last;
}

template <typename T>
__global__ __launch_bounds__(kThreads) void d_tail_fwd_kernel(const TailK p) {
  using TR = ElemTraits<T>;
  constexpr int E = TR::kVec;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int per = p.N / p.groups, H5 = p.H4 / 2, hw4 = p.H4 * p.H4, hw5 = H5 * H5;
  const Lds l = carve(smem, p, per);
  const int C4 = p.C4, vpp = C4 / E, PS = C4 * TR::kBytes + 16;   // pixel stride of the LDS image (+16 B: neighbouring pixels on other slots)
  const int n = blockIdx.x, mygrp = n / per;
  float* const zs = p.ws + kWsHead + p.groups * 2 * 64;
  stage_w5<T>(p, l);
  // ---- BN(block4.1): batch statistics of this sample's group from the conv epilogue's sums (tg_bn_apply's arithmetic).  Workgroup 0
  //      also leaves every group's saved statistics and the running statistics (one update per group, in order, as the reference's
  //      two forward calls make them)
  if (tid < C4) {
    const float cnt = (float)(per * hw4);
    const size_t rblock = (size_t)p.groups * 2 * C4;
    float rm = (n == 0 && p.rm4) ? p.rm4[tid] : 0.f, rv = (n == 0 && p.rv4) ? p.rv4[tid] : 0.f;
    for (int g = (n == 0 ? 0 : mygrp); g < (n == 0 ? p.groups : mygrp + 1); ++g) {
      const float mean = fold1(p.stats4 + (g * 2 + 0) * C4 + tid, p.R4, rblock) / cnt;
      float var = fold1(p.stats4 + (g * 2 + 1) * C4 + tid, p.R4, rblock) / cnt - mean * mean;
      var = var < 0.f ? 0.f : var;
      const float invstd = rsqrtf(var + p.eps);
      if (g == mygrp) {
        const float scale = p.gamma4[tid] * invstd;
        l.par[tid] = scale;
        l.par[C4 + tid] = p.beta4[tid] - mean * scale;
      }
      if (n == 0) {
        p.save4[(g * 2 + 0) * C4 + tid] = mean;
        p.save4[(g * 2 + 1) * C4 + tid] = invstd;
        rm = (1.f - p.momentum) * rm + p.momentum * mean;
        rv = (1.f - p.momentum) * rv + p.momentum * var * (cnt / (cnt - 1.f));
      }
    }
    if (n == 0) {
      if (p.rm4) p.rm4[tid] = rm;
      if (p.rv4) p.rv4[tid] = rv;
      if (tid == 0 && p.nbt4) *p.nbt4 += p.groups;   // num_batches_tracked: one per forward call of the reference
    }
  }
  __syncthreads();
  {   // n4 of this sample: to memory and into the LDS image
    const int vec = tid % vpp;   // (the stride is a multiple of vpp: a thread's channel vector is fixed)
    float sc[E], sh[E];
#pragma unroll
    for (int e = 0; e < E; ++e) { sc[e] = l.par[vec * E + e]; sh[e] = l.par[C4 + vec * E + e]; }
    const int total = hw4 * vpp;
    const char* zg = p.z4 + (size_t)n * hw4 * vpp * 16;
    char* ng = p.n4 + (size_t)n * hw4 * vpp * 16;
    for (int i0 = tid; i0 < total; i0 += 4 * kThreads) {   // four vectors per trip, every load issued before the first use
      u32x4 raw[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) raw[u] = *reinterpret_cast<const u32x4*>(zg + (size_t)min(i0 + u * kThreads, total - 1) * 16);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + u * kThreads;
        if (i < total) {
          float v[E];
          Vec<T>::load(&raw[u], v);
#pragma unroll
          for (int e = 0; e < E; ++e) {
            v[e] = v[e] * sc[e] + sh[e];
            v[e] = v[e] > 0.f ? v[e] : 0.2f * v[e];
          }
          Vec<T>::pack(l.img + (i / vpp) * PS + vec * 16, v);
          Vec<T>::store(ng + (size_t)i * 16, v);
        }
      }
    }
  }
  __syncthreads();
  // ---- block5.0: conv 4x4 s2 p1, 64 -> C5, from the LDS image.  Lane = (tap, quarter of the input channels) keeps its 16 x C5 weights
  //      in registers; a wave takes one output pixel per step and adds its 64 partial sums up through shuffles
  {
    constexpr int NV = 16 / E;                               // 16-byte vectors of the lane's 16 channels
    const int tap = lane >> 2, q = lane & 3, ky = tap >> 2, kx = tap & 3;
    float w[3][16], w3[16];
#pragma unroll
    for (int co = 0; co < 3; ++co)
#pragma unroll
      for (int j = 0; j < 16; j += 4) {
        f32x4 t = {0.f, 0.f, 0.f, 0.f};
        if (co < p.C5) t = *reinterpret_cast<const f32x4*>(l.w5s + tap * (p.C5 * C4 + 4) + co * C4 + q * 16 + j);
        w[co][j] = t[0]; w[co][j + 1] = t[1]; w[co][j + 2] = t[2]; w[co][j + 3] = t[3];
      }
#pragma unroll
    for (int j = 0; j < 16; ++j) w3[j] = p.C5 > 3 ? l.w5s[tap * (p.C5 * C4 + 4) + 3 * C4 + q * 16 + j] : 0.f;   // (a fourth channel, if any)
    for (int opx = wave; opx < hw5; opx += kWaves) {
      const int oy = opx / H5, ox = opx - oy * H5;
      const int iy = 2 * oy + ky - 1, ix = 2 * ox + kx - 1;
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
      if (iy >= 0 && iy < p.H4 && ix >= 0 && ix < p.H4) {
        const char* src = l.img + (iy * p.H4 + ix) * PS + q * 16 * TR::kBytes;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          float x[E];
          Vec<T>::load(src + v * 16, x);
#pragma unroll
          for (int e = 0; e < E; ++e) {
            acc[0] += x[e] * w[0][v * E + e];
            acc[1] += x[e] * w[1][v * E + e];
            acc[2] += x[e] * w[2][v * E + e];
            acc[3] += x[e] * w3[v * E + e];
          }
        }
      }
#pragma unroll
      for (int co = 0; co < 4; ++co) acc[co] = wave_sum(acc[co]);
      if (lane < 4) zs[((size_t)n * hw5 + opx) * 4 + lane] = lane == 0 ? acc[0] : lane == 1 ? acc[1] : lane == 2 ? acc[2] : acc[3];
    }
  }
  if (!last_ticket(p, l)) return;
  // ======== the last workgroup: BN(block5.1) over each group's samples, LeakyReLU, flatten, Linear, sigmoid
  float rm5 = 0.f, rv5 = 0.f;
  if (tid < p.Cp5) { rm5 = p.rm5 ? p.rm5[tid] : 0.f; rv5 = p.rv5 ? p.rv5[tid] : 0.f; }
  for (int grp = 0; grp < p.groups; ++grp) {
    const int n0 = grp * per;
    for (int i = tid; i < per * hw5 * 4; i += kThreads) l.zb[i] = zs[(size_t)n0 * hw5 * 4 + i];
    __syncthreads();
    // statistics of the fp32 values (as the conv epilogues take them), one wave per real channel
    if (wave < p.C5) {
      float s1 = 0.f, s2 = 0.f;
      for (int px = lane; px < per * hw5; px += 64) {
        const float z = l.zb[px * 4 + wave];
        s1 += z;
        s2 += z * z;
      }
      s1 = wave_sum(s1);
      s2 = wave_sum(s2);
      if (lane == 0) { l.red[wave * 2] = s1; l.red[wave * 2 + 1] = s2; }
    }
    __syncthreads();
    if (tid < p.Cp5) {   // padded channels: sums 0 -> mean 0, invstd rsqrt(eps); gamma / beta padding is 0
      const float cnt = (float)(per * hw5);
      const float mean = tid < p.C5 ? l.red[tid * 2] / cnt : 0.f;
      float var = tid < p.C5 ? l.red[tid * 2 + 1] / cnt - mean * mean : 0.f;
      var = var < 0.f ? 0.f : var;
      const float invstd = rsqrtf(var + p.eps), scale = p.gamma5[tid] * invstd;
      l.par[2 * C4 + tid] = scale;
      l.par[3 * C4 + tid] = p.beta5[tid] - mean * scale;
      p.save5[(grp * 2 + 0) * p.Cp5 + tid] = mean;
      p.save5[(grp * 2 + 1) * p.Cp5 + tid] = invstd;
      rm5 = (1.f - p.momentum) * rm5 + p.momentum * mean;
      rv5 = (1.f - p.momentum) * rv5 + p.momentum * var * (cnt / (cnt - 1.f));
    }
    __syncthreads();
    for (int i = tid; i < per * hw5 * p.Cp5; i += kThreads) {   // z5 and n5 with their padding channels (zeros)
      const int c = i % p.Cp5, px = i / p.Cp5;
      float zr = 0.f, nr = 0.f;
      if (c < p.C5) {
        zr = rnd<T>(l.zb[px * 4 + c]);
        float v = zr * l.par[2 * C4 + c] + l.par[3 * C4 + c];
        v = v > 0.f ? v : 0.2f * v;
        nr = rnd<T>(v);
        l.nb[px * 4 + c] = nr;
      }
      store_elem<T>(p.z5, (size_t)n0 * hw5 * p.Cp5 + i, zr);
      store_elem<T>(p.n5, (size_t)n0 * hw5 * p.Cp5 + i, nr);
    }
    __syncthreads();
    for (int s_ = wave; s_ < per; s_ += kWaves) {   // flatten (channel-major) + Linear + sigmoid: one wave per sample
      float s = 0.f;
      for (int i = lane; i < p.C5 * hw5; i += 64) {
        const int c = i / hw5, q = i - c * hw5;
        s += p.fc_w[i] * l.nb[(s_ * hw5 + q) * 4 + c];
      }
      s = wave_sum(s);
      if (lane == 0) p.prob[n0 + s_] = 1.f / (1.f + __expf(-(s + p.fc_b[0])));
    }
    __syncthreads();   // the next group reuses par / zb / nb
  }
  if (tid < p.Cp5) {
    if (p.rm5) p.rm5[tid] = rm5;
    if (p.rv5) p.rv5[tid] = rv5;
  }
  if (tid == 0) {
    if (p.nbt5) *p.nbt5 += p.groups;
    *reinterpret_cast<unsigned*>(p.ws) = 0u;   // the ticket, for the next launch (the fp32 z5 scratch is overwritten, not accumulated)
  }
}

template <typename T>
__global__ __launch_bounds__(kThreads) void d_tail_bwd_kernel(const TailK p) {
  using TR = ElemTraits<T>;
  constexpr int E = TR::kVec;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int per = p.N / p.groups, H5 = p.H4 / 2, hw4 = p.H4 * p.H4, hw5 = H5 * H5;
  const Lds l = carve(smem, p, per);
  const int C4 = p.C4, vpp = C4 / E;
  const int n = blockIdx.x, grp = n / per, n0 = grp * per, nl = n - n0;
  const bool first = nl == 0;   // the group's first workgroup adds the parameter gradients every workgroup of the group computes
  float* const red4 = p.ws + kWsHead;   // [groups][2][C4]: block4.1's backward sums, zero at launch
  stage_w5<T>(p, l);
  // ---- d(loss)/d(logit) of the group's samples: given, or (real half) seeded here with tg_dlogit_real's expression
  if (tid < per) {
    float d;
    if (p.seed_real) {
      const float eps = p.cfg[6], inv = 1.f / (float)p.N, pr = p.prob[n0 + tid];
      const float S = p.loss_scale ? *p.loss_scale : 1.f;
      d = -S * inv * (1.f / (pr + eps)) * pr * (1.f - pr);
      if (first) p.dlogit[n0 + tid] = d;
    } else {
      d = p.dlogit[n0 + tid];
    }
    l.dl[tid] = d;
  }
  if (tid < p.Cp5) {   // BN(block5.1): saved statistics, k0 = gamma * invstd
    l.par[tid] = p.save5[(grp * 2 + 0) * p.Cp5 + tid];
    l.par[C4 + tid] = p.save5[(grp * 2 + 1) * p.Cp5 + tid];
    l.par[2 * C4 + tid] = p.gamma5[tid] * p.save5[(grp * 2 + 1) * p.Cp5 + tid];
  }
  if (tid < C4) {      // BN(block4.1)
    l.par[3 * C4 + tid] = p.save4[(grp * 2 + 0) * C4 + tid];
    l.par[4 * C4 + tid] = p.save4[(grp * 2 + 1) * C4 + tid];
  }
  __syncthreads();
  // ---- fc backward (tg_fc_head_bwd) over the whole group (192 x 3 values: every workgroup of the group repeats it): d n5 = dlogit * W
  //      (rounded as its store would), LeakyReLU', the two sums of BN(block5.1)'s backward - one wave per real channel.  zb = dd, nb = x-hat
  if (wave < p.C5) {
    const int c = wave;
    float s1 = 0.f, s2 = 0.f;
    for (int px = lane; px < per * hw5; px += 64) {
      const int s_ = px / hw5, q = px - s_ * hw5;
      const size_t gi = ((size_t)(n0 + s_) * hw5 + q) * p.Cp5 + c;
      const float dfeat = rnd<T>(l.dl[s_] * p.fc_w[c * hw5 + q]);
      const float dd = dfeat * (load_elem<T>(p.n5, gi) > 0.f ? 1.f : 0.2f);
      const float xh = (load_elem<T>(p.z5, gi) - l.par[c]) * l.par[C4 + c];
      l.zb[px * 4 + c] = dd;
      l.nb[px * 4 + c] = xh;
      s1 += dd;
      s2 += dd * xh;
    }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    if (lane == 0) { l.red[c * 2] = s1; l.red[c * 2 + 1] = s2; }
  }
  if (first && tid >= 256 && tid < 256 + p.C5 * hw5) {   // dW += dlogit . n5 (waves 4..: beside the channel waves)
    const int i = tid - 256, c = i / hw5, q = i - c * hw5;
    float s = 0.f;
    for (int s_ = 0; s_ < per; ++s_) s += l.dl[s_] * load_elem<T>(p.n5, ((size_t)(n0 + s_) * hw5 + q) * p.Cp5 + c);
    atomicAdd(p.g_fc_w + i, s);   // (the groups' first workgroups run side by side)
  }
  if (first && tid == 255) {
    float s = 0.f;
    for (int s_ = 0; s_ < per; ++s_) s += l.dl[s_];
    atomicAdd(p.g_fc_b, s);
  }
  __syncthreads();
  if (first && tid < p.C5) {
    atomicAdd(p.dbeta5 + tid, l.red[tid * 2]);
    atomicAdd(p.dgamma5 + tid, l.red[tid * 2 + 1]);
  }
  {
    const float inv_cnt = 1.f / (float)(per * hw5);
    for (int i = tid; i < per * hw5 * p.Cp5; i += kThreads) {   // d z5 = k0 (dd - mean dd - x-hat mean(dd x-hat)), padding channels 0
      const int c = i % p.Cp5, px = i / p.Cp5;
      float v = 0.f;
      if (c < p.C5) v = rnd<T>(l.par[2 * C4 + c] * (l.zb[px * 4 + c] - l.red[c * 2] * inv_cnt - l.nb[px * 4 + c] * l.red[c * 2 + 1] * inv_cnt));
      if (c < 4) l.zb[px * 4 + c] = v;                                        // (this thread is the only reader of that dd)
      if (px / hw5 == nl) store_elem<T>(p.dz5, (size_t)n0 * hw5 * p.Cp5 + i, v);   // this sample's rows
    }
  }
  __syncthreads();
  // ---- block5.0's input-gradient for THIS sample: d n4[iy][ix][ci] = sum over (ky, kx) with 2 oy + ky - 1 = iy, 2 ox + kx - 1 = ix, and co,
  //      of W[co][ci][ky][kx] * d z5[oy][ox][co]; then LeakyReLU'(n4) and the two sums of BN(block4.1)'s backward (a thread's channel
  //      vector is fixed: the stride is a multiple of vpp).  d n4 goes to memory rounded, as the separate launches left it
  float s1[E], s2[E];
#pragma unroll
  for (int e = 0; e < E; ++e) s1[e] = s2[e] = 0.f;
  const int vec = tid % vpp;
  {
    const int total = hw4 * vpp;
    const size_t g0 = (size_t)n * hw4 * vpp * 16;
    for (int i = tid; i < total; i += kThreads) {
      const u32x4 ra = *reinterpret_cast<const u32x4*>(p.n4 + g0 + (size_t)i * 16), rz = *reinterpret_cast<const u32x4*>(p.z4 + g0 + (size_t)i * 16);
      const int pix = i / vpp, iy = pix / p.H4, ix = pix - iy * p.H4;
      float d[E];
#pragma unroll
      for (int e = 0; e < E; ++e) d[e] = 0.f;
      for (int ky = (iy + 1) & 1; ky < 4; ky += 2) {
        const int oy = (iy + 1 - ky) >> 1;
        if (oy < 0 || oy >= H5) continue;
        for (int kx = (ix + 1) & 1; kx < 4; kx += 2) {
          const int ox = (ix + 1 - kx) >> 1;
          if (ox < 0 || ox >= H5) continue;
          const float* wt = l.w5s + (ky * 4 + kx) * (p.C5 * C4 + 4) + vec * E;
          const f32x4 dzv = *reinterpret_cast<const f32x4*>(l.zb + ((nl * hw5) + oy * H5 + ox) * 4);
#pragma unroll
          for (int co = 0; co < 4; ++co) {
            if (co < p.C5) {
              const float g = dzv[co];
#pragma unroll
              for (int e = 0; e < E; e += 4) {
                const f32x4 wv = *reinterpret_cast<const f32x4*>(wt + co * C4 + e);
                d[e] += wv[0] * g; d[e + 1] += wv[1] * g; d[e + 2] += wv[2] * g; d[e + 3] += wv[3] * g;
              }
            }
          }
        }
      }
      Vec<T>::store(p.dn4 + g0 + (size_t)i * 16, d);
      float a[E], z[E];
      Vec<T>::load(&ra, a);
      Vec<T>::load(&rz, z);
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const float dd = rnd<T>(d[e]) * (a[e] > 0.f ? 1.f : 0.2f);
        s1[e] += dd;
        s2[e] += dd * (z[e] - l.par[3 * C4 + vec * E + e]) * l.par[4 * C4 + vec * E + e];
      }
    }
  }
  // lanes of a wave with the same channel vector (lane % vpp), the waves through LDS, then ONE atomic per channel and sum
#pragma unroll
  for (int e = 0; e < E; ++e) {
    for (int m = vpp; m < 64; m <<= 1) {
      s1[e] += __shfl_xor(s1[e], m);
      s2[e] += __shfl_xor(s2[e], m);
    }
  }
  if (lane < vpp) {
#pragma unroll
    for (int e = 0; e < E; ++e) {
      l.red[(wave * 2 + 0) * C4 + lane * E + e] = s1[e];
      l.red[(wave * 2 + 1) * C4 + lane * E + e] = s2[e];
    }
  }
  __syncthreads();
  if (tid < 2 * C4) {
    const int which = tid / C4, c = tid - which * C4;
    float s = 0.f;
    for (int w = 0; w < kWaves; ++w) s += l.red[(w * 2 + which) * C4 + c];
    atomicAdd(red4 + (grp * 2 + which) * C4 + c, s);
  }
  if (!last_ticket(p, l)) return;
  // ======== the last workgroup: d z4 = k0 (dd - mean dd - x-hat mean(dd x-hat)) for every sample, group by group
  for (int g = 0; g < p.groups; ++g) {
    if (tid < 2 * C4) l.par[(6 + tid / C4) * C4 + tid % C4] = red4[g * 2 * C4 + tid];
    if (tid < C4) {
      l.par[3 * C4 + tid] = p.save4[(g * 2 + 0) * C4 + tid];
      l.par[4 * C4 + tid] = p.save4[(g * 2 + 1) * C4 + tid];
      l.par[5 * C4 + tid] = p.gamma4[tid] * p.save4[(g * 2 + 1) * C4 + tid];
    }
    __syncthreads();
    if (tid < C4) {   // (this workgroup alone: plain read-modify-write)
      p.dbeta4[tid] += l.par[6 * C4 + tid];
      p.dgamma4[tid] += l.par[7 * C4 + tid];
    }
    {
      const float inv_cnt = 1.f / (float)(per * hw4);
      float mean[E], invstd[E], k0[E], m1[E], m2[E];
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const int c = vec * E + e;
        mean[e] = l.par[3 * C4 + c]; invstd[e] = l.par[4 * C4 + c]; k0[e] = l.par[5 * C4 + c];
        m1[e] = l.par[6 * C4 + c] * inv_cnt; m2[e] = l.par[7 * C4 + c] * inv_cnt;
      }
      const int total = per * hw4 * vpp;
      const size_t g0 = (size_t)g * per * hw4 * vpp * 16;
      for (int i0 = tid; i0 < total; i0 += 2 * kThreads) {   // two vectors of each tensor per trip, every load issued before the first use
        u32x4 rd[2], ra[2], rz[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const size_t off = g0 + (size_t)min(i0 + u * kThreads, total - 1) * 16;
          rd[u] = *reinterpret_cast<const u32x4*>(p.dn4 + off);
          ra[u] = *reinterpret_cast<const u32x4*>(p.n4 + off);
          rz[u] = *reinterpret_cast<const u32x4*>(p.z4 + off);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          if (i0 + u * kThreads < total) {
            float d[E], a[E], z[E];
            Vec<T>::load(&rd[u], d);
            Vec<T>::load(&ra[u], a);
            Vec<T>::load(&rz[u], z);
#pragma unroll
            for (int e = 0; e < E; ++e) {
              const float dd = d[e] * (a[e] > 0.f ? 1.f : 0.2f);
              d[e] = k0[e] * (dd - m1[e] - (z[e] - mean[e]) * invstd[e] * m2[e]);
            }
            Vec<T>::store(p.dz4 + g0 + (size_t)(i0 + u * kThreads) * 16, d);
          }
        }
      }
    }
    __syncthreads();   // the next group reuses par
  }
  if (tid < p.groups * 2 * C4) red4[tid] = 0.f;   // scratch back to zero for the next launch
  if (tid == 0) *reinterpret_cast<unsigned*>(p.ws) = 0u;
}

int check_tail(int dtype, int N, int H4, int C4, int C5, int Cp5, int groups) {
  if (dtype != TG_BF16 && dtype != TG_F16 && dtype != TG_F32) return TG_E_BADARG;
  if (N <= 0 || groups <= 0 || N % groups || H4 <= 0 || (H4 & 1)) return TG_E_BADARG;
  const int E = dtype == TG_F32 ? 4 : 8;
  // block4's 64 output channels (the forward conv's lane = tap x a quarter of them); the fc gradient and the per-sample seeds have one thread each
  if (C4 != 64 || C5 < 1 || C5 > 4 || Cp5 < C5 || Cp5 % 32 || Cp5 > C4) return TG_E_UNSUPPORTED;
  const int per = N / groups, hw5 = (H4 / 2) * (H4 / 2);
  if (per > 256 || C5 * hw5 > 256 || (long long)per * H4 * H4 > tg_d_tail_max_pixels()) return TG_E_UNSUPPORTED;
  if (lds_bytes(per, H4, C4, C5, dtype == TG_F32 ? 4 : 2) > 160 * 1024) return TG_E_UNSUPPORTED;
  return TG_OK;
}

template <typename K> int launch_tail(K kernel, const TailK& k, size_t lds, hipStream_t st) {
  if (lds > 64 * 1024) TG_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kernel, dim3((unsigned)k.N), dim3(kThreads), lds, st, k);   // one workgroup per sample
  return tg_launch_status();
}

}  // namespace

// per BatchNorm group, at block4's output: beyond, the last workgroup's sweeps are the wrong shape (run the separate launches)
extern "C" long long tg_d_tail_max_pixels(void) { return 4096; }
extern "C" long long tg_d_tail_scratch_floats(int N, int H4, int groups) {
  return (N > 0 && H4 > 0 && groups > 0) ? (long long)ws_floats(N, H4, groups) : -1;
}

extern "C" int tg_d_tail_fwd(int dtype, const void* z4, const float* stats4, int stats_replicas, const float* gamma4, const float* beta4,
                             float* rmean4, float* rvar4, int64_t* nbt4, float* save4, void* n4, const float* w5, void* z5,
                             const float* gamma5, const float* beta5, float* rmean5, float* rvar5, int64_t* nbt5, float* save5, void* n5,
                             const float* fc_w, const float* fc_b, float* prob, int N, int H4, int C4, int C5, int Cp5, int groups,
                             float eps, float momentum, float* scratch, void* stream) {
  if (!scratch) return TG_E_BADARG;
  if (!z4 || !stats4 || !gamma4 || !beta4 || !save4 || !n4 || !w5 || !z5 || !gamma5 || !beta5 || !save5 || !n5 || !fc_w || !fc_b || !prob)
    return TG_E_BADARG;
  if (stats_replicas < 1) return TG_E_BADARG;
  const int rc = check_tail(dtype, N, H4, C4, C5, Cp5, groups);
  if (rc != TG_OK) return rc;
  if (!tg_aligned16(z4) || !tg_aligned16(n4) || !tg_aligned16(z5) || !tg_aligned16(n5)) return TG_E_ALIGN;
  TailK k = {};
  k.z4 = (const char*)z4; k.n4 = (char*)n4; k.z5 = (char*)z5; k.n5 = (char*)n5;
  k.stats4 = stats4; k.gamma4 = gamma4; k.beta4 = beta4; k.gamma5 = gamma5; k.beta5 = beta5; k.w5 = w5; k.fc_w = fc_w; k.fc_b = fc_b;
  k.rm4 = rmean4; k.rv4 = rvar4; k.save4 = save4; k.rm5 = rmean5; k.rv5 = rvar5; k.save5 = save5; k.prob = prob;
  k.nbt4 = (long long*)nbt4; k.nbt5 = (long long*)nbt5;
  k.N = N; k.H4 = H4; k.C4 = C4; k.C5 = C5; k.Cp5 = Cp5; k.groups = groups; k.R4 = stats_replicas; k.eps = eps; k.momentum = momentum;
  k.ws = scratch;
  const size_t lds = lds_bytes(N / groups, H4, C4, C5, dtype == TG_F32 ? 4 : 2);
  hipStream_t st = (hipStream_t)stream;
  TG_DISPATCH_DTYPE(dtype, return launch_tail(d_tail_fwd_kernel<BF16>, k, lds, st), return launch_tail(d_tail_fwd_kernel<F16>, k, lds, st),
                    return launch_tail(d_tail_fwd_kernel<F32>, k, lds, st));
  return TG_OK;
}

extern "C" int tg_d_tail_bwd(int dtype, const float* dlogit_in_out, const float* prob, const float* cfg, const float* loss_scale,
                             int seed_real, const void* n5, const void* z5, const float* save5, const float* gamma5, const float* fc_w,
                             const float* w5, const void* n4, const void* z4, const float* save4, const float* gamma4, void* dz5, void* dn4,
                             void* dz4, float* g_fc_w, float* g_fc_b, float* dgamma5, float* dbeta5, float* dgamma4, float* dbeta4, int N,
                             int H4, int C4, int C5, int Cp5, int groups, float* scratch, void* stream) {
  if (!scratch || (seed_real && groups != 1)) return TG_E_BADARG;
  if (!dlogit_in_out || !n5 || !z5 || !save5 || !gamma5 || !fc_w || !w5 || !n4 || !z4 || !save4 || !gamma4 || !dz5 || !dn4 || !dz4 ||
      !g_fc_w || !g_fc_b || !dgamma5 || !dbeta5 || !dgamma4 || !dbeta4)
    return TG_E_BADARG;
  if (seed_real && (!prob || !cfg)) return TG_E_BADARG;
  const int rc = check_tail(dtype, N, H4, C4, C5, Cp5, groups);
  if (rc != TG_OK) return rc;
  if (!tg_aligned16(z4) || !tg_aligned16(n4) || !tg_aligned16(dn4) || !tg_aligned16(dz4)) return TG_E_ALIGN;
  TailK k = {};
  k.z4 = (const char*)z4; k.n4 = (char*)const_cast<void*>(n4); k.z5 = (char*)const_cast<void*>(z5); k.n5 = (char*)const_cast<void*>(n5);
  k.gamma4 = gamma4; k.gamma5 = gamma5; k.w5 = w5; k.fc_w = fc_w;
  k.save4 = const_cast<float*>(save4); k.save5 = const_cast<float*>(save5); k.prob = const_cast<float*>(prob);
  k.N = N; k.H4 = H4; k.C4 = C4; k.C5 = C5; k.Cp5 = Cp5; k.groups = groups; k.R4 = 1;
  k.dn4 = (char*)dn4; k.dz4 = (char*)dz4; k.dz5 = (char*)dz5; k.dlogit = const_cast<float*>(dlogit_in_out);
  k.g_fc_w = g_fc_w; k.g_fc_b = g_fc_b; k.dgamma4 = dgamma4; k.dbeta4 = dbeta4; k.dgamma5 = dgamma5; k.dbeta5 = dbeta5;
  k.cfg = cfg; k.loss_scale = loss_scale; k.seed_real = seed_real ? 1 : 0;
  k.ws = scratch;
  const size_t lds = lds_bytes(N / groups, H4, C4, C5, dtype == TG_F32 ? 4 : 2);
  hipStream_t st = (hipStream_t)stream;
  TG_DISPATCH_DTYPE(dtype, return launch_tail(d_tail_bwd_kernel<BF16>, k, lds, st), return launch_tail(d_tail_bwd_kernel<F16>, k, lds, st),
                    return launch_tail(d_tail_bwd_kernel<F32>, k, lds, st));
  return TG_OK;
}
