// Shared by resblock_ws.hip and resblock2_ws.hip (round 5): the 32x32x16 matrix instruction, the conflict-free LDS image offsets, LDS-DMA.
#pragma once
#include "common.h"

typedef __attribute__((ext_vector_type(16))) float f32x16;


namespace {

template <typename T> struct Mma32;
template <> struct Mma32<BF16> {
  __device__ __forceinline__ static f32x16 run(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct Mma32<F16> {
  __device__ __forceinline__ static f32x16 run(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  }
};

// byte offset of 16-byte piece `piece` of row `row` in the W1 / h images
__device__ __forceinline__ int img_off(int row, int piece) { return row * 64 + ((piece ^ ((row >> 2) & 3)) << 4); }
// ... of patch pixel `prow` (patch_row), 32-channel chunk c: a KiB of the image holds 8 pixels, chunk 0 rows then chunk 1 rows, so
// that one DMA instruction reads 8 WHOLE pixels (128-byte lines); the row is still = prow mod 4 and the swizzle key prow / 4 mod 4
__device__ __forceinline__ int patch_off(int prow, int c, int piece) {
  return (16 * (prow >> 3) + 8 * c + (prow & 7)) * 64 + ((piece ^ ((prow >> 2) & 3)) << 4);
}

#ifndef RBW_DMA_AUX   // A/B builds: cache-policy bits on the LDS-DMA (" nt", " sc1", ...): profiles/r05_j_dma_policy_ab.log
#define RBW_DMA_AUX ""
#endif
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" RBW_DMA_AUX : : "v"(gsrc), "s"(lds_dst) : "memory", "m0");
}
// LDS-only barrier: __syncthreads() would also drain vmcnt (the weight stream, the h stores)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
template <typename T> __device__ __forceinline__ unsigned pack2(float lo, float hi);
template <> __device__ __forceinline__ unsigned pack2<BF16>(float lo, float hi) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{lo, hi}, bf16x2_t));
}
template <> __device__ __forceinline__ unsigned pack2<F16>(float lo, float hi) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{lo, hi}, f16x2_t));
}


}  // namespace
