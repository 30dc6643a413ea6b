// Stream helpers of the step schedule (host side only; no kernels).
//
// The training step runs two lanes of launches (step.py): the serial generator chain, whose residual-block launches are
// 64 workgroups each, and the dense discriminator / backward work.  MI355X has no working stream priority for this
// (profiles/r02_a_overlap_probe_priority_cumask.log: a high-priority stream changes nothing), but a queue can be confined
// to a CU subset: hipExtStreamCreateWithCUMask.  The dense lane's stream is created WITHOUT the first `reserve` mask bits,
// which the probe shows to be 8 CUs on each of the 8 XCDs per 64 bits, so the chain's small launches always find free CUs
// with free LDS instead of queueing behind long-running dense workgroups.  A mask survives hipGraph replay only when the
// graph is launched on the masked stream itself (one linear graph per lane), not inside one forked capture.
#include "common.h"
#include <vector>

extern "C" int tg_stream_create_cumask(int reserve_cus, void** stream_out) {
  if (!stream_out || reserve_cus < 0) return TG_E_BADARG;
  int dev = 0;
  TG_CHECK_HIP(hipGetDevice(&dev));
  hipDeviceProp_t prop;
  TG_CHECK_HIP(hipGetDeviceProperties(&prop, dev));
  const int ncu = prop.multiProcessorCount;
  if (reserve_cus >= ncu) return TG_E_BADARG;
  const int words = (ncu + 31) / 32;
  std::vector<uint32_t> mask(words, 0u);
  for (int i = reserve_cus; i < ncu; ++i) mask[i / 32] |= 1u << (i % 32);
  hipStream_t s = nullptr;
  TG_CHECK_HIP(hipExtStreamCreateWithCUMask(&s, (uint32_t)words, mask.data()));
  *stream_out = (void*)s;
  return TG_OK;
}

extern "C" int tg_stream_destroy(void* stream) {
  if (!stream) return TG_E_BADARG;
  TG_CHECK_HIP(hipStreamDestroy((hipStream_t)stream));
  return TG_OK;
}
