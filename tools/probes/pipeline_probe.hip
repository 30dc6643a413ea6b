// Probe (diagnostic, not product): a LAYER-PIPELINED, weights-stationary residual trunk.
//
// Today one recurrent pass runs its 16 residual blocks as 16 dependent launches of 128 workgroups, 6.7 us each = 108 us:
// ~3 us of every launch is the kernel boundary, ~1.3 us the 147-KB weight stream.  The spatial persistent trunk (every workgroup
// keeps its tile and exchanges halos with 8 neighbours per block) was closed in round 3: a neighbour round costs what the boundary
// costs (profiles/r03_h_handoff_probe.log).  THIS decomposition is different: stage k (= residual block k, its weights resident in
// registers for the whole step) owns a fixed group of CUs; the frame (4 images x 32 x 32 pixels x 64 channels) streams through the
// stages in BANDS of 4 rows, handed from stage to stage through L2 with tagged flags.
//
//   grid = STAGES x 8 workgroups of 512 threads (one per CU): slot c of a stage = (image c / 2, column half c % 2), i.e. 16 columns
//   x 32 rows, processed as 8 bands of 4 rows.  Band b of stage k needs rows 4b - 2 .. 4b + 5 and columns - 2 .. + 17 of stage
//   k - 1's output (two 3 x 3 convolutions): bands b - 1, b, b + 1 of BOTH column halves of that image.  Bands arrive in order, so
//   the wait is for band b + 1 of two producers (a "2 -> 1" hand-off; the last band waits for band 7).
//   Per band a workgroup: polls the two flags (one lane each, relaxed sc1 loads, bounded spin), loads its (4 + 4) x (16 + 4)-pixel
//   input patch (20 KB, sc1 loads), runs `mfma` dependent-free MFMAs per wave standing in for conv1 on the (4 + 2) x (16 + 2)
//   region + conv2 on 4 x 16 (7 + 4 pixel tiles x 4 channel tiles x 18 k-steps = 792 MFMAs per band = 99 per wave), stores its
//   8-KB output band (sc1 write-through stores), drains them, publishes the band number.
//
// Reported: the FRAME LATENCY - first poll of stage 0 to the last band of the last stage published (s_memrealtime, the chip-wide
// 100-MHz counter) - for STAGES = 16, with and without the arithmetic stand-in, alone and beside a memory-streaming
// neighbour on the other 128 CUs; and the same 16 x 8 bands as launches would do them (for scale).
// Kill criterion (VERDICT): >= 90 us => not built.  Every spin is bounded.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x)                                                                           \
  do {                                                                                  \
    hipError_t e_ = (x);                                                                \
    if (e_ != hipSuccess) {                                                             \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__);     \
      exit(1);                                                                          \
    }                                                                                   \
  } while (0)

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
constexpr int IMG = 4, HW = 32, PIXB = 128, SLOTS = 8, BANDS = 8, BR = 4, CW = 16;

__device__ __forceinline__ void st_sc1(void* p, u32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ u32x4 ld_sc1(const void* p) {
  u32x4 a;
  asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(a) : "v"(p) : "memory");
  return a;
}

// bufs: STAGES + 1 frame buffers (stage k reads k, writes k + 1); flags[(stage) * SLOTS + slot] = bands published (+ base)
__global__ __launch_bounds__(512) void pipeline_kernel(char* bufs, unsigned* flags, unsigned* err, long long* t_first,
                                                       long long* t_last, int stages, int mfma, unsigned base) {
  const int stage = blockIdx.x / SLOTS, slot = blockIdx.x % SLOTS, tid = threadIdx.x;
  const int img = slot >> 1, xh = slot & 1;
  const size_t frame = (size_t)IMG * HW * HW * PIXB;
  const char* in = bufs + (size_t)stage * frame;
  char* out = bufs + (size_t)(stage + 1) * frame;
  __shared__ int bad;
  if (tid == 0) bad = 0;
  __syncthreads();
  bf16x8 a = {}, b = {};
  f32x4 acc[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  unsigned keep = 0;
  if (stage == 0 && slot == 0 && tid == 0) t_first[0] = (long long)__builtin_amdgcn_s_memrealtime();   // (100 MHz, chip-wide)
  for (int band = 0; band < BANDS; ++band) {
    // ---- wait: band min(band + 1, 7) of both column halves of this image at the previous stage
    if (stage > 0 && tid < 2) {
      const unsigned want = base + (unsigned)min(band + 2, BANDS);
      const unsigned* f = flags + (stage - 1) * SLOTS + img * 2 + tid;
      int spins = 0;
      while ((int)(__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) < 0) {
        if (++spins > 4000000) {
          bad = 1;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    __syncthreads();
    if (bad) {
      if (tid == 0) atomicAdd(err, 1u);
      return;
    }
    // ---- input patch: (BR + 4) x (CW + 4) pixels x 8 pieces of 16 B = 1280 pieces: 2.5 per thread (clamped at the border)
    {
      u32x4 v[3];
#pragma unroll
      for (int u = 0; u < 3; ++u) {   // (unconditional from a clamped piece index: all three loads in flight together)
        const int i = min(tid + 512 * u, (BR + 4) * (CW + 4) * 8 - 1);
        const int pp = i >> 3, pj = i & 7;
        const int py = min(max(band * BR - 2 + pp / (CW + 4), 0), HW - 1), px = min(max(xh * CW - 2 + pp % (CW + 4), 0), HW - 1);
        v[u] = ld_sc1(in + ((size_t)img * HW * HW + (size_t)py * HW + px) * PIXB + pj * 16);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      keep += v[0].x + v[1].x + v[2].x;
    }
    // ---- the block's arithmetic: `mfma` MFMAs per wave on four independent accumulators
    for (int k = 0; k < mfma; k += 4) {
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[j], 0, 0, 0);
    }
    // ---- output band: BR x CW pixels x 8 pieces = 512 pieces, one per thread; drain, publish
    {
      const int pp = tid >> 3, pj = tid & 7;
      const int py = band * BR + pp / CW, px = xh * CW + pp % CW;
      u32x4 v = {(unsigned)tid, keep, (unsigned)acc[0][0], (unsigned)band};
      st_sc1(out + ((size_t)img * HW * HW + (size_t)py * HW + px) * PIXB + pj * 16, v);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_store(flags + stage * SLOTS + slot, base + (unsigned)band + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (stage == stages - 1 && tid == 0) t_last[slot] = (long long)__builtin_amdgcn_s_memrealtime();
}

// the same work as dependent launches: one launch per stage, 8 workgroups x 8 bands (for scale only)
__global__ __launch_bounds__(512) void stage_kernel(const char* in, char* out, int mfma) {
  const int slot = blockIdx.x % SLOTS, band = blockIdx.x / SLOTS, tid = threadIdx.x;
  const int img = slot >> 1, xh = slot & 1;
  bf16x8 a = {}, b = {};
  f32x4 acc[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  unsigned keep = 0;
  for (int u = 0; u < 3; ++u) {
    const int i = tid + 512 * u;
    if (i < (BR + 4) * (CW + 4) * 8) {
      const int pp = i >> 3, pj = i & 7;
      const int py = min(max(band * BR - 2 + pp / (CW + 4), 0), HW - 1), px = min(max(xh * CW - 2 + pp % (CW + 4), 0), HW - 1);
      keep += reinterpret_cast<const u32x4*>(in + ((size_t)img * HW * HW + (size_t)py * HW + px) * PIXB + pj * 16)->x;
    }
  }
  for (int k = 0; k < mfma; k += 4) {
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[j], 0, 0, 0);
  }
  const int pp = tid >> 3, pj = tid & 7;
  const int py = band * BR + pp / CW, px = xh * CW + pp % CW;
  u32x4 v = {(unsigned)tid, keep, (unsigned)acc[0][0], (unsigned)band};
  *reinterpret_cast<u32x4*>(out + ((size_t)img * HW * HW + (size_t)py * HW + px) * PIXB + pj * 16) = v;
}

__global__ __launch_bounds__(256) void stream_kernel(const u32x4* src, u32x4* dst, size_t n, int passes) {
  for (int p = 0; p < passes; ++p)
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

int main() {
  const int stages = 16, reps = 50;
  const size_t frame = (size_t)IMG * HW * HW * PIXB;
  char* bufs;
  unsigned *flags, *err;
  long long *t_first, *t_last;
  CK(hipMalloc(&bufs, frame * (stages + 1)));
  CK(hipMemset(bufs, 1, frame * (stages + 1)));
  CK(hipMalloc(&flags, stages * SLOTS * sizeof(unsigned)));
  CK(hipMemset(flags, 0, stages * SLOTS * sizeof(unsigned)));
  CK(hipMalloc(&err, 4));
  CK(hipMemset(err, 0, 4));
  CK(hipMalloc(&t_first, 8));
  CK(hipMalloc(&t_last, 8 * SLOTS));
  const double ticks_per_us = 100.0;   // s_memrealtime runs at 100 MHz on every XCD
  const size_t sn = 64u << 20;   // neighbour: 1 GiB source / destination pair, streamed by 128 workgroups
  u32x4 *ssrc, *sdst;
  CK(hipMalloc(&ssrc, sn * 16));
  CK(hipMalloc(&sdst, sn * 16));
  hipStream_t s1, s2;
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  unsigned base = 0;
  for (int neighbour = 0; neighbour < 2; ++neighbour) {
    for (int mfma : {0, 100}) {
      std::vector<double> lat;
      for (int r = 0; r < reps; ++r) {
        if (neighbour) hipLaunchKernelGGL(stream_kernel, dim3(128), dim3(256), 0, s2, ssrc, sdst, sn / 8, 1);
        hipLaunchKernelGGL(pipeline_kernel, dim3(stages * SLOTS), dim3(512), 0, s1, bufs, flags, err, t_first, t_last, stages, mfma, base);
        CK(hipStreamSynchronize(s1));
        if (neighbour) CK(hipStreamSynchronize(s2));
        base += BANDS + 8;
        long long tf, tl[SLOTS];
        CK(hipMemcpy(&tf, t_first, 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(tl, t_last, 8 * SLOTS, hipMemcpyDeviceToHost));
        long long last = tl[0];
        for (int i = 1; i < SLOTS; ++i) last = std::max(last, tl[i]);
        if (r >= 5) lat.push_back((double)(last - tf) / ticks_per_us);
      }
      unsigned e;
      CK(hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost));
      std::sort(lat.begin(), lat.end());
      printf("pipeline, %2d stages x 8 CUs x 8 bands, %3d MFMAs per wave and band, %s: frame latency median %.1f us (min %.1f, p90 %.1f)%s\n",
             stages, mfma, neighbour ? "beside a streaming neighbour on 128 CUs" : "alone", lat[lat.size() / 2], lat.front(),
             lat[lat.size() * 9 / 10], e ? "  [BOUNDED SPIN EXPIRED]" : "");
    }
  }
  // for scale: the same per-stage work as 16 dependent launches of 64 workgroups (hipGraph replay)
  for (int mfma : {0, 100}) {
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s1, hipStreamCaptureModeThreadLocal));
    for (int k = 0; k < stages; ++k)
      hipLaunchKernelGGL(stage_kernel, dim3(SLOTS * BANDS), dim3(512), 0, s1, bufs + (size_t)k * frame, bufs + (size_t)(k + 1) * frame, mfma);
    CK(hipStreamEndCapture(s1, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int r = 0; r < 5; ++r) CK(hipGraphLaunch(ge, s1));
    CK(hipEventRecord(e0, s1));
    for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, s1));
    CK(hipEventRecord(e1, s1));
    CK(hipStreamSynchronize(s1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("launches, 16 dependent launches of 64 workgroups (hipGraph), %3d MFMAs per wave: %.1f us per frame\n", mfma, ms * 1e3 / reps);
  }
  return 0;
}
