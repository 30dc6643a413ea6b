// Probe (diagnostic, not product): can a latency-critical chain of small launches share the chip with dense launches?
//   chain  = NCHAIN dependent launches of 64 workgroups x 512 threads, ~88 KB LDS, a few microseconds each
//   dense  = NDENSE launches of 2048 workgroups x 256 threads, ~70 KB LDS, ~20 us per workgroup
// Variants: streams of equal priority / chain on a high-priority stream / chain and dense on CU-masked streams with
// disjoint masks; each eager, as ONE captured graph (fork/join inside the capture) and as one graph per stream.
// Also prints which (XCC, SE, CU) the workgroups of a masked stream land on, i.e. the layout of the CU mask bits.
// Every loop is bounded; nothing spins on memory.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>
#include <vector>

#define CK(x)                                                                           \
  do {                                                                                  \
    hipError_t e_ = (x);                                                                \
    if (e_ != hipSuccess) {                                                             \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__);     \
      exit(1);                                                                          \
    }                                                                                   \
  } while (0)

__global__ __launch_bounds__(512) void small_kernel(float* buf, unsigned* where, int iters) {
  extern __shared__ float lds[];
  float v = buf[blockIdx.x * 512 + threadIdx.x];
  lds[threadIdx.x] = v;
  __syncthreads();
  for (int i = 0; i < iters; ++i) v = fmaf(v, 1.0001f, lds[(threadIdx.x + i) & 511]);
  buf[blockIdx.x * 512 + threadIdx.x] = v;
  if (where && threadIdx.x == 0) {
    const unsigned hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));   // HW_REG_HW_ID
    const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));  // HW_REG_XCC_ID
    where[blockIdx.x] = (xcc << 16) | (hw & 0xffffu);
  }
}

__global__ __launch_bounds__(256) void dense_kernel(float* buf, unsigned* where, int iters) {
  extern __shared__ float lds[];
  float v = buf[(blockIdx.x * 256 + threadIdx.x) & 0xfffff];
  lds[threadIdx.x] = v;
  __syncthreads();
  for (int i = 0; i < iters; ++i) v = fmaf(v, 1.0001f, lds[(threadIdx.x + i) & 255]);
  buf[(blockIdx.x * 256 + threadIdx.x) & 0xfffff] = v;
  if (where && threadIdx.x == 0) {
    const unsigned hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));
    const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));
    where[blockIdx.x] = (xcc << 16) | (hw & 0xffffu);
  }
}

static int NCHAIN = 400, NDENSE = 40, IT_SMALL = 60, IT_DENSE = 400;
static float *bufA, *bufB;
static unsigned *whereA, *whereB;

static void chain(hipStream_t s, unsigned* where = nullptr) {
  for (int i = 0; i < NCHAIN; ++i)
    hipLaunchKernelGGL(small_kernel, dim3(64), dim3(512), 88 * 1024, s, bufA, i == NCHAIN - 1 ? where : nullptr, IT_SMALL);
}
static void dense(hipStream_t s, unsigned* where = nullptr) {
  for (int i = 0; i < NDENSE; ++i)
    hipLaunchKernelGGL(dense_kernel, dim3(2048), dim3(256), 70 * 1024, s, bufB, i == NDENSE - 1 ? where : nullptr, IT_DENSE);
}

static double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

struct Times { double chain_ms, all_ms; };

// eager: both streams start together; chain end is taken from an event on its stream
static Times run_eager(hipStream_t sa, hipStream_t sb, bool do_chain, bool do_dense) {
  hipEvent_t e0, ea;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&ea));
  CK(hipDeviceSynchronize());
  const double t0 = now_ms();
  CK(hipEventRecord(e0, sa));
  if (do_dense) dense(sb);
  if (do_chain) chain(sa);
  CK(hipEventRecord(ea, sa));
  CK(hipDeviceSynchronize());
  const double t1 = now_ms();
  float c = 0;
  CK(hipEventElapsedTime(&c, e0, ea));
  CK(hipEventDestroy(e0)); CK(hipEventDestroy(ea));
  return {c, t1 - t0};
}

// one graph, fork/join inside the capture; origin = so
static double run_one_graph(hipStream_t so, hipStream_t sa, hipStream_t sb, unsigned flags, const char* tag) {
  hipGraph_t g;
  hipGraphExec_t ge;
  hipEvent_t fork, ja, jb;
  CK(hipEventCreate(&fork)); CK(hipEventCreate(&ja)); CK(hipEventCreate(&jb));
  CK(hipStreamBeginCapture(so, hipStreamCaptureModeThreadLocal));
  CK(hipEventRecord(fork, so));
  CK(hipStreamWaitEvent(sa, fork, 0));
  CK(hipStreamWaitEvent(sb, fork, 0));
  dense(sb);
  chain(sa);
  CK(hipEventRecord(ja, sa));
  CK(hipEventRecord(jb, sb));
  CK(hipStreamWaitEvent(so, ja, 0));
  CK(hipStreamWaitEvent(so, jb, 0));
  CK(hipStreamEndCapture(so, &g));
  hipError_t e = flags ? hipGraphInstantiateWithFlags(&ge, g, flags) : hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  if (e != hipSuccess) {
    printf("%s: instantiate failed: %s\n", tag, hipGetErrorString(e));
    (void)hipGetLastError();
    return -1;
  }
  CK(hipGraphLaunch(ge, so));
  CK(hipDeviceSynchronize());
  double best = 1e9;
  for (int r = 0; r < 3; ++r) {
    const double t0 = now_ms();
    CK(hipGraphLaunch(ge, so));
    CK(hipDeviceSynchronize());
    best = std::min(best, now_ms() - t0);
  }
  CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  return best;
}

// one linear graph per stream, launched on that stream
static Times run_two_graphs(hipStream_t sa, hipStream_t sb) {
  hipGraph_t ga, gb;
  hipGraphExec_t gea, geb;
  CK(hipStreamBeginCapture(sa, hipStreamCaptureModeThreadLocal));
  chain(sa);
  CK(hipStreamEndCapture(sa, &ga));
  CK(hipStreamBeginCapture(sb, hipStreamCaptureModeThreadLocal));
  dense(sb);
  CK(hipStreamEndCapture(sb, &gb));
  CK(hipGraphInstantiate(&gea, ga, nullptr, nullptr, 0));
  CK(hipGraphInstantiate(&geb, gb, nullptr, nullptr, 0));
  hipEvent_t e0, ea;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&ea));
  Times best{1e9, 1e9};
  for (int r = 0; r < 4; ++r) {
    CK(hipDeviceSynchronize());
    const double t0 = now_ms();
    CK(hipEventRecord(e0, sa));
    CK(hipGraphLaunch(geb, sb));
    CK(hipGraphLaunch(gea, sa));
    CK(hipEventRecord(ea, sa));
    CK(hipDeviceSynchronize());
    const double t1 = now_ms();
    float c = 0;
    CK(hipEventElapsedTime(&c, e0, ea));
    if (r > 0 && t1 - t0 < best.all_ms) best = {c, t1 - t0};
  }
  CK(hipGraphExecDestroy(gea)); CK(hipGraphExecDestroy(geb)); CK(hipGraphDestroy(ga)); CK(hipGraphDestroy(gb));
  return best;
}

static void census(const char* tag, unsigned* where_dev, int n) {
  std::vector<unsigned> h(n);
  CK(hipMemcpy(h.data(), where_dev, n * sizeof(unsigned), hipMemcpyDeviceToHost));
  std::set<unsigned> cus;
  int per_xcc[16] = {0};
  for (unsigned v : h) {
    const unsigned xcc = v >> 16, hw = v & 0xffffu, cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    cus.insert((xcc << 12) | (se << 8) | (sh << 4) | cu);
    per_xcc[xcc & 15]++;
  }
  printf("%s: %d workgroups on %zu distinct CUs; per XCC:", tag, n, cus.size());
  for (int i = 0; i < 8; ++i) printf(" %d", per_xcc[i]);
  printf("\n   (xcc.se.sh.cu):");
  int k = 0;
  for (unsigned c : cus) {
    if (k++ < 48) printf(" %u.%u.%u.%u", c >> 12, (c >> 8) & 15, (c >> 4) & 15, c & 15);
  }
  printf("%s\n", cus.size() > 48 ? " ..." : "");
}

int main(int argc, char** argv) {
  setvbuf(stdout, nullptr, _IONBF, 0);
  if (argc > 1) NCHAIN = atoi(argv[1]);
  if (argc > 2) NDENSE = atoi(argv[2]);
  const int phase = argc > 3 ? atoi(argv[3]) : 7;  // bit 0: eager/priority, bit 1: CU masks, bit 2: graphs without masks
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("device %s, %d CUs, streamPrioritiesSupported %d\n", prop.name, prop.multiProcessorCount, prop.streamPrioritiesSupported);
  int lo = 0, hi = 0;
  CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  printf("priority range: least %d greatest %d\n", lo, hi);
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(small_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(dense_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipMalloc(&bufA, 64 * 512 * 4)); CK(hipMalloc(&bufB, (1 << 20) * 4));
  CK(hipMemset(bufA, 0, 64 * 512 * 4)); CK(hipMemset(bufB, 0, (1 << 20) * 4));
  CK(hipMalloc(&whereA, 64 * 4)); CK(hipMalloc(&whereB, 2048 * 4));

  hipStream_t so, sa, sb, sa_hi, sb_lo;
  CK(hipStreamCreateWithFlags(&so, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  CK(hipStreamCreateWithPriority(&sa_hi, hipStreamNonBlocking, hi));
  CK(hipStreamCreateWithPriority(&sb_lo, hipStreamNonBlocking, lo));

  // warm-up
  chain(sa); dense(sb);
  CK(hipDeviceSynchronize());

  Times t;
  if (phase & 1) {
  t = run_eager(sa, sb, true, false);  printf("eager  chain alone                : chain %.3f ms\n", t.chain_ms);
  t = run_eager(sa, sb, false, true);  printf("eager  dense alone                : all %.3f ms\n", t.all_ms);
  t = run_eager(sa, sb, true, true);   printf("eager  both, equal priority       : chain %.3f ms, all %.3f ms\n", t.chain_ms, t.all_ms);
  t = run_eager(sa_hi, sb_lo, true, true); printf("eager  both, chain high priority  : chain %.3f ms, all %.3f ms\n", t.chain_ms, t.all_ms);
  t = run_eager(sa_hi, sb_lo, true, false); printf("eager  chain alone (hi stream)    : chain %.3f ms\n", t.chain_ms);
  }

  // CU masks.  Try several layouts of a 1/4 : 3/4 split and look at where the workgroups land.
  const int ncu = prop.multiProcessorCount;
  const int words = (ncu + 31) / 32;
  struct MaskCase { const char* name; std::vector<uint32_t> a, b; };
  std::vector<MaskCase> cases;
  {
    MaskCase m{"low quarter of the bits", std::vector<uint32_t>(words, 0), std::vector<uint32_t>(words, 0)};
    for (int i = 0; i < ncu; ++i) (i < ncu / 4 ? m.a : m.b)[i / 32] |= 1u << (i % 32);
    cases.push_back(m);
  }
  {
    MaskCase m{"every 4th bit", std::vector<uint32_t>(words, 0), std::vector<uint32_t>(words, 0)};
    for (int i = 0; i < ncu; ++i) ((i % 4) == 0 ? m.a : m.b)[i / 32] |= 1u << (i % 32);
    cases.push_back(m);
  }
  {
    MaskCase m{"bits i with (i/8)%4==0", std::vector<uint32_t>(words, 0), std::vector<uint32_t>(words, 0)};
    for (int i = 0; i < ncu; ++i) (((i / 8) % 4) == 0 ? m.a : m.b)[i / 32] |= 1u << (i % 32);
    cases.push_back(m);
  }
  for (auto& m : cases) {
    if (!(phase & 2)) break;
    hipStream_t ma, mb;
    hipError_t e1 = hipExtStreamCreateWithCUMask(&ma, words, m.a.data());
    hipError_t e2 = hipExtStreamCreateWithCUMask(&mb, words, m.b.data());
    if (e1 != hipSuccess || e2 != hipSuccess) {
      printf("CU mask '%s': create failed: %s / %s\n", m.name, hipGetErrorString(e1), hipGetErrorString(e2));
      (void)hipGetLastError();
      continue;
    }
    printf("---- CU mask case '%s'\n", m.name);
    CK(hipMemset(whereA, 0xff, 64 * 4)); CK(hipMemset(whereB, 0xff, 2048 * 4));
    chain(ma, whereA);
    CK(hipDeviceSynchronize());
    printf("  masked chain ran\n");
    dense(mb, whereB);
    CK(hipDeviceSynchronize());
    printf("  masked dense ran\n");
    census("  chain stream", whereA, 64);
    census("  dense stream", whereB, 2048);
    t = run_eager(ma, mb, true, false);  printf("  eager chain alone (masked)      : chain %.3f ms\n", t.chain_ms);
    t = run_eager(ma, mb, false, true);  printf("  eager dense alone (masked)      : all %.3f ms\n", t.all_ms);
    t = run_eager(ma, mb, true, true);   printf("  eager both (disjoint masks)     : chain %.3f ms, all %.3f ms\n", t.chain_ms, t.all_ms);
    t = run_eager(ma, sb, true, true);   printf("  eager chain masked, dense free  : chain %.3f ms, all %.3f ms\n", t.chain_ms, t.all_ms);
    double g1 = run_one_graph(so, ma, mb, 0, "one graph (masked capture streams)");
    printf("  ONE graph, masked capture streams, launched on plain stream : %.3f ms\n", g1);
    Times t2 = run_two_graphs(ma, mb);
    printf("  TWO graphs, each launched on its masked stream             : chain %.3f ms, all %.3f ms\n", t2.chain_ms, t2.all_ms);
    CK(hipStreamDestroy(ma)); CK(hipStreamDestroy(mb));
  }

  if (!(phase & 4)) { printf("done\n"); return 0; }
  printf("---- graphs without masks\n");
  double g0 = run_one_graph(so, sa, sb, 0, "one graph");
  printf("ONE graph, equal priority streams                       : %.3f ms\n", g0);
  double g2 = run_one_graph(so, sa_hi, sb_lo, 0, "one graph prio");
  printf("ONE graph, chain captured on high-priority stream       : %.3f ms\n", g2);
  double g3 = run_one_graph(so, sa_hi, sb_lo, hipGraphInstantiateFlagUseNodePriority, "one graph node prio");
  printf("ONE graph, same + hipGraphInstantiateFlagUseNodePriority : %.3f ms\n", g3);
  Times t3 = run_two_graphs(sa, sb);
  printf("TWO graphs, equal priority streams                      : chain %.3f ms, all %.3f ms\n", t3.chain_ms, t3.all_ms);
  Times t4 = run_two_graphs(sa_hi, sb_lo);
  printf("TWO graphs, chain graph on high-priority stream         : chain %.3f ms, all %.3f ms\n", t4.chain_ms, t4.all_ms);
  printf("done\n");
  return 0;
}
