// Probe (diagnostic, not product): what does ONE CU's vector-memory pipe accept per clock - bytes or instructions?
//
// conv3_rw's producer waves issue 16 one-KiB vector-memory instructions per wave and tile (8 LDS-DMA, 4 loads, 4 stores) and spend
// ~165 ticks on each (profiles/r05_x_rw_producer_diag.log); resblock_ws streams 156 KiB per workgroup in ~7000 ticks.  Both are
// "~25 B/clk per CU" if the cost is per byte and "~41 ticks per wave-instruction per CU" if it is per instruction.  This probe
// runs one workgroup per CU; every wave issues NI back-to-back loads (or LDS-DMA requests, or stores) of 4 / 8 / 16 bytes per lane
// from an L2-resident 64-KiB window, with all 64 lanes or only 32 / 16 active, 1 - 8 waves per workgroup, and reports
// ticks per wave-instruction and bytes per clock of the CU (workgroup 0's waves, s_memtime).
//   hipcc --offload-arch=gfx950 -O3 -o vmem_rate_probe vmem_rate_probe.hip && ./vmem_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                           \
  do {                                                                                  \
    hipError_t e_ = (x);                                                                \
    if (e_ != hipSuccess) {                                                             \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__);     \
      exit(1);                                                                          \
    }                                                                                   \
  } while (0)

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
constexpr int NI = 64;   // instructions per wave

// MODE 0: global_load (W bytes per lane), 1: global_load_lds_dwordx4 (LDS-DMA, 16 B per lane), 2: global_store (W bytes per lane)
template <int MODE, int W>
__global__ __launch_bounds__(512) void probe(const char* src, char* dst, long long* ticks, int active_lanes) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  // every wave walks its own 64 x 1 KiB of the window (L2 hits after the first pass); lanes 16 bytes apart
  const char* p = src + (size_t)(blockIdx.x & 7) * 65536 + wid * 8192 + lane * 16;
  char* q = dst + (size_t)blockIdx.x * 65536 * 8 + wid * 65536 + lane * 16;
  u32x4 acc = {0u, 0u, 0u, 0u};
  unsigned long long acc2 = 0;
  unsigned acc1 = 0;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + wid * 1024);
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  if (lane < active_lanes) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const char* a = p + (i & 7) * 1024;
      if constexpr (MODE == 0) {
        // (the destination registers stay allocated for the whole kernel: the loads are asynchronous, and a register the compiler
        // believed free again would be overwritten by late data - an address, for one)
        if constexpr (W == 16) asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(acc) : "v"(a) : "memory");
        if constexpr (W == 8) asm volatile("global_load_dwordx2 %0, %1, off" : "+v"(acc2) : "v"(a) : "memory");
        if constexpr (W == 4) asm volatile("global_load_dword %0, %1, off" : "+v"(acc1) : "v"(a) : "memory");
      } else if constexpr (MODE == 1) {
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(a), "s"(lds0) : "memory", "m0");
      } else {
        char* b = q + (i & 7) * 1024;
        if constexpr (W == 16) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(b), "v"(acc) : "memory");
        if constexpr (W == 4) asm volatile("global_store_dword %0, %1, off" ::"v"(b), "v"(acc[0]) : "memory");
      }
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();   // all NI instructions ISSUED
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const long long t2 = __builtin_amdgcn_s_memtime();   // ... and returned
  if (blockIdx.x == 0 && lane == 0) {
    ticks[wid * 2] = t1 - t0;
    ticks[wid * 2 + 1] = t2 - t0;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (acc[0] + (unsigned)acc2 + acc1 == 0x12345u) dst[0] = 1;
}

template <int MODE, int W>
void run(const char* name, const char* src, char* dst, long long* ticks, int waves, int lanes, int grid) {
  CK(hipMemset(ticks, 0, 16 * sizeof(long long)));
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((probe<MODE, W>), dim3(grid), dim3(64 * waves), 8192, 0, src, dst, ticks, lanes);
  CK(hipDeviceSynchronize());
  long long h[16];
  CK(hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost));
  long long issue = 0, done = 0;
  for (int w = 0; w < waves; ++w) {
    issue = h[2 * w] > issue ? h[2 * w] : issue;
    done = h[2 * w + 1] > done ? h[2 * w + 1] : done;
  }
  const double bytes = (double)waves * NI * lanes * W;
  printf("%-34s waves %d lanes %2d grid %3d: issue %6lld ticks (%5.1f per wave-instruction), returned %6lld ticks: %5.1f B/clk, %5.3f wave-instr/clk per CU\n",
         name, waves, lanes, grid, issue, (double)issue / NI, done, bytes / done, (double)waves * NI / done);
}

int main() {
  char *src, *dst;
  long long* ticks;
  CK(hipMalloc(&src, 8 * 65536));
  CK(hipMalloc(&dst, (size_t)256 * 65536 * 8));
  CK(hipMalloc(&ticks, 16 * sizeof(long long)));
  CK(hipMemset(src, 1, 8 * 65536));
  for (int grid : {1, 256}) {
    for (int waves : {1, 4, 8}) {
      run<0, 16>("global_load_dwordx4", src, dst, ticks, waves, 64, grid);
      run<0, 8>("global_load_dwordx2", src, dst, ticks, waves, 64, grid);
      run<0, 4>("global_load_dword", src, dst, ticks, waves, 64, grid);
      run<0, 16>("global_load_dwordx4 (32 lanes)", src, dst, ticks, waves, 32, grid);
      run<0, 16>("global_load_dwordx4 (16 lanes)", src, dst, ticks, waves, 16, grid);
      run<1, 16>("global_load_lds_dwordx4 (LDS-DMA)", src, dst, ticks, waves, 64, grid);
      run<1, 16>("global_load_lds_dwordx4 (32 lanes)", src, dst, ticks, waves, 32, grid);
      run<2, 16>("global_store_dwordx4", src, dst, ticks, waves, 64, grid);
      run<2, 4>("global_store_dword", src, dst, ticks, waves, 64, grid);
    }
  }
  return 0;
}
