// Probe: (1) does HW_REG_XCC_ID identify the XCD a workgroup runs on (blockIdx round-robin)?  (2) throughput of fp32
// atomic adds into per-XCD replicas at workgroup scope (executed in the XCD-local L2) vs agent scope vs plain slab stores.
// build: hipcc --offload-arch=gfx950 -O3 -o atomics_probe atomics_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ int xcc_id() {
  // s_getreg_b32 hwreg(HW_REG_XCC_ID = 20, offset 0, width 4)
  return __builtin_amdgcn_s_getreg(20 | (0 << 6) | ((4 - 1) << 11)) & 15;
}

__global__ void probe_ids(int* out) {
  if (threadIdx.x == 0) out[blockIdx.x] = xcc_id();
}

// mode 0: plain stores into slab[blockIdx.x]; 1: workgroup-scope atomics into replica[xcc]; 2: agent-scope atomics into
// replica[blockIdx % 8]
template <int MODE, int ROT>
__global__ __launch_bounds__(256) void accum(float* buf, int n_per_wg) {
  float* dst;
  if (MODE == 0) dst = buf + (size_t)blockIdx.x * n_per_wg;
  else if (MODE == 1) dst = buf + (size_t)(xcc_id() & 7) * n_per_wg;
  else dst = buf + (size_t)(blockIdx.x & 7) * n_per_wg;
  const int rot = ROT ? (int)(((blockIdx.x >> 3) * 1184u) % (unsigned)n_per_wg) : 0;  // stagger the walk per workgroup
  for (int i0 = threadIdx.x; i0 < n_per_wg; i0 += 256) {
    int i = i0 + rot;
    if (i >= n_per_wg) i -= n_per_wg;
    const float v = 1.0f;
    if (MODE == 0) dst[i] = v;
    else if (MODE == 1) __hip_atomic_fetch_add(dst + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else __hip_atomic_fetch_add(dst + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

int main() {
  const int WGS = 256, N = 36864;  // one 64x64x9 weight-gradient tile set per workgroup
  int* ids;
  CK(hipMalloc(&ids, WGS * 4));
  hipLaunchKernelGGL(probe_ids, dim3(WGS), dim3(64), 0, 0, ids);
  std::vector<int> h(WGS);
  CK(hipMemcpy(h.data(), ids, WGS * 4, hipMemcpyDeviceToHost));
  int match = 0;
  for (int i = 0; i < WGS; ++i) match += (h[i] == (i & 7));
  printf("xcc ids of blocks 0..15:");
  for (int i = 0; i < 16; ++i) printf(" %d", h[i]);
  printf("\nblocks whose xcc == blockIdx %% 8: %d / %d\n", match, WGS);

  float* buf;
  CK(hipMalloc(&buf, (size_t)WGS * N * 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int mode = 0; mode < 5; ++mode) {
    CK(hipMemset(buf, 0, (size_t)WGS * N * 4));
    CK(hipDeviceSynchronize());
    const int iters = 50;
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0, 0));
      for (int it = 0; it < iters; ++it) {
        if (mode == 0) hipLaunchKernelGGL((accum<0, 0>), dim3(WGS), dim3(256), 0, 0, buf, N);
        if (mode == 1) hipLaunchKernelGGL((accum<1, 0>), dim3(WGS), dim3(256), 0, 0, buf, N);
        if (mode == 2) hipLaunchKernelGGL((accum<2, 0>), dim3(WGS), dim3(256), 0, 0, buf, N);
        if (mode == 3) hipLaunchKernelGGL((accum<1, 1>), dim3(WGS), dim3(256), 0, 0, buf, N);
        if (mode == 4) hipLaunchKernelGGL((accum<2, 1>), dim3(WGS), dim3(256), 0, 0, buf, N);
      }
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
    }
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<float> hb(8 * N);
    CK(hipMemcpy(hb.data(), buf, 8 * N * 4, hipMemcpyDeviceToHost));
    double sum = 0;
    for (int i = 0; i < 8 * N; ++i) sum += hb[i];
    // modes 1/2: total adds = 2 reps * iters * WGS * N ones spread over 8 replicas
    printf("mode %d: %.2f us per launch (%.1f MB of element traffic each); sum over first 8 blocks = %.0f (expected %s %.0f)\n",
           mode, ms * 1000.0 / iters, WGS * N * 4 / 1e6, sum, mode ? "" : "stores:", mode ? 2.0 * iters * WGS * N : 8.0 * N);
  }
  return 0;
}
