// Shared device/host helpers for libtecogan_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include "../../include/tecogan_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

// Variants that were built, measured slower and rejected (DESIGN.md "rejected") are compiled only into the experiments
// library (build.sh --experiments => -DTG_EXPERIMENTS): the default libtecogan_hip.so exports what the default step can reach.
#ifdef TG_EXPERIMENTS
constexpr bool kTgExperiments = true;
#else
constexpr bool kTgExperiments = false;
#endif

struct BF16 {};  // tag types: element type is a template tag, never a runtime branch inside kernels
struct F32 {};
struct F16 {};   // IEEE half (BASELINE configs[3]: the reference's fp16 autocast path); same layouts as BF16

template <typename T> struct ElemTraits;
template <> struct ElemTraits<F32> {
  static constexpr int kBytes = 4;
  static constexpr int kChunk = 16;  // channels per 64-byte K chunk
  static constexpr int kVec = 4;     // elements per 16-byte vector
};
template <> struct ElemTraits<BF16> {
  static constexpr int kBytes = 2;
  static constexpr int kChunk = 32;
  static constexpr int kVec = 8;
};

template <> struct ElemTraits<F16> {
  static constexpr int kBytes = 2;
  static constexpr int kChunk = 32;
  static constexpr int kVec = 8;
};

__device__ __forceinline__ float f16_bits_to_f32(unsigned short b) { return (float)__builtin_bit_cast(_Float16, b); }
__device__ __forceinline__ unsigned short f32_to_f16_bits(float f) {
  _Float16 h = (_Float16)f;  // v_cvt_f16_f32: RNE, overflow -> inf (what the loss-scale overflow check looks for)
  return __builtin_bit_cast(unsigned short, h);
}
__device__ __forceinline__ float bf16_bits_to_f32(unsigned short b) { return __uint_as_float(((unsigned)b) << 16); }
__device__ __forceinline__ unsigned short f32_to_bf16_bits(float f) {
  __bf16 h = (__bf16)f;  // v_cvt_pk_bf16_f32: RNE, NaN-preserving
  return __builtin_bit_cast(unsigned short, h);
}

// A 16-byte store to GLOBAL memory.  A source file that #defines TG_ST_AUX "sc1" in front of its includes (conv3_rw.hip, conv3_cw.hip,
// convt_cw.hip, conv4s2d_cw.hip, conv_s2_cw.hip - per file, not from build.sh) gets it written THROUGH the L2 at agent scope.  Every kernel boundary
// writes the XCDs' dirty L2 lines back (the next launch's workgroups run on other XCDs) - whoever dirtied them: with write-back stores
// the 210 dependent launches of the recurrent pass paid at each of their boundaries for the lines the discriminator's launches on the
// other lane had just written (profiles/r05_u_write_through_ab.log).  (Inline asm: a store the compiler's vmcnt bookkeeping does not see
// only makes its waits conservative - the counter is in order.  BUT: the compiler's vmcnt scoreboard does not see these stores, so
// it may drop a later fence's vmcnt(0) as redundant - a file that defines TG_ST_AUX must NOT publish such a store to another workgroup
// inside the same kernel (__threadfence + flag / atomic) without an explicit `s_waitcnt vmcnt(0)` in front of the fence.  None of the
// five files does; d_tail.hip, which publishes through tickets, uses plain stores.)
__device__ __forceinline__ void tg_store16(void* p, u32x4 v) {
#ifdef TG_ST_AUX
  // (s_nop 1: a store of more than 8 bytes whose data registers the NEXT vector instruction overwrites needs wait states; the compiler
  // inserts them for its own stores and cannot see into this one)
  asm volatile("global_store_dwordx4 %0, %1, off " TG_ST_AUX "\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
#else
  *reinterpret_cast<u32x4*>(p) = v;
#endif
}

__device__ __forceinline__ void tg_store4(float* p, float v) {   // one float, the same way
#ifdef TG_ST_AUX
  asm volatile("global_store_dword %0, %1, off " TG_ST_AUX ::"v"(p), "v"(v) : "memory");
#else
  *p = v;
#endif
}

// Load/store `kVec` consecutive elements (16 bytes) as floats (store: to global memory; pack: anywhere).
template <typename T> struct Vec;
template <> struct Vec<F32> {
  static constexpr int N = 4;
  __device__ __forceinline__ static void load(const void* p, float* v) {
    f32x4 t = *reinterpret_cast<const f32x4*>(p);
    v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
  }
  __device__ __forceinline__ static void pack(void* p, const float* v) {
    f32x4 t = {v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(p) = t;
  }
  __device__ __forceinline__ static void store(void* p, const float* v) {
    f32x4 t = {v[0], v[1], v[2], v[3]};
    tg_store16(p, __builtin_bit_cast(u32x4, t));
  }
};
template <> struct Vec<BF16> {
  static constexpr int N = 8;
  __device__ __forceinline__ static void load(const void* p, float* v) {
#ifdef TG_VEC_LD_NT   // A/B only (profiles/r05_u_write_through_ab.log, section 6): streaming reads that leave the L2 to the other lane
    u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
#else
    u32x4 t = *reinterpret_cast<const u32x4*>(p);
#endif
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[2 * i] = __uint_as_float(t[i] << 16);
      v[2 * i + 1] = __uint_as_float(t[i] & 0xffff0000u);
    }
  }
  __device__ __forceinline__ static u32x4 bits(const float* v) {
    u32x4 t;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      t[i] = (unsigned)f32_to_bf16_bits(v[2 * i]) | ((unsigned)f32_to_bf16_bits(v[2 * i + 1]) << 16);
    return t;
  }
  __device__ __forceinline__ static void pack(void* p, const float* v) { *reinterpret_cast<u32x4*>(p) = bits(v); }
  __device__ __forceinline__ static void store(void* p, const float* v) { tg_store16(p, bits(v)); }
};

template <> struct Vec<F16> {
  static constexpr int N = 8;
  __device__ __forceinline__ static void load(const void* p, float* v) {
    f16x8 t = *reinterpret_cast<const f16x8*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)t[i];
  }
  __device__ __forceinline__ static void pack(void* p, const float* v) {
    f16x8 t;
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = (_Float16)v[i];
    *reinterpret_cast<f16x8*>(p) = t;
  }
  __device__ __forceinline__ static void store(void* p, const float* v) {
    f16x8 t;
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = (_Float16)v[i];
    tg_store16(p, __builtin_bit_cast(u32x4, t));
  }
};

template <typename T> __device__ __forceinline__ float load_elem(const void* base, int64_t i);
template <> __device__ __forceinline__ float load_elem<F16>(const void* base, int64_t i) {
  return f16_bits_to_f32(reinterpret_cast<const unsigned short*>(base)[i]);
}
template <> __device__ __forceinline__ float load_elem<F32>(const void* base, int64_t i) {
  return reinterpret_cast<const float*>(base)[i];
}
template <> __device__ __forceinline__ float load_elem<BF16>(const void* base, int64_t i) {
  return bf16_bits_to_f32(reinterpret_cast<const unsigned short*>(base)[i]);
}
template <typename T> __device__ __forceinline__ void store_elem(void* base, int64_t i, float v);
template <> __device__ __forceinline__ void store_elem<F32>(void* base, int64_t i, float v) {
  reinterpret_cast<float*>(base)[i] = v;
}
template <> __device__ __forceinline__ void store_elem<BF16>(void* base, int64_t i, float v) {
  reinterpret_cast<unsigned short*>(base)[i] = f32_to_bf16_bits(v);
}
template <> __device__ __forceinline__ void store_elem<F16>(void* base, int64_t i, float v) {
  reinterpret_cast<unsigned short*>(base)[i] = f32_to_f16_bits(v);
}

// Packed-weight row order.  Row R of the packed matrix holds output channel row_to_channel(R): for bf16 two
// adjacent 16-row MFMA tiles are interleaved so that one lane's 2x4 accumulator rows are 8 consecutive
// channels (one 16-byte store); for f32 the 4 accumulator rows of a lane already are 16 bytes.
template <typename T> __host__ __device__ __forceinline__ int row_to_channel(int R);
template <> __host__ __device__ __forceinline__ int row_to_channel<F32>(int R) { return R; }
template <> __host__ __device__ __forceinline__ int row_to_channel<BF16>(int R) {
  int tile = R >> 4, r = R & 15, q = r >> 2, j = r & 3;
  return 32 * (tile >> 1) + 8 * q + 4 * (tile & 1) + j;
}

template <> __host__ __device__ __forceinline__ int row_to_channel<F16>(int R) { return row_to_channel<BF16>(R); }

// 16x16x32 MFMA on 16-bit operands (8 values per lane and operand), fp32 accumulate
template <typename T> struct Mma16;
template <> struct Mma16<BF16> {
  __device__ __forceinline__ static f32x4 run(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct Mma16<F16> {
  __device__ __forceinline__ static f32x4 run(bf16x8 a, bf16x8 b, f32x4 c) {  // fragments travel as raw 16-byte vectors
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  }
};

// bits <-> float of one 16-bit element of tag T (BF16 / F16)
template <typename T> __device__ __forceinline__ float bits16_to_f32(unsigned short b);
template <> __device__ __forceinline__ float bits16_to_f32<BF16>(unsigned short b) { return bf16_bits_to_f32(b); }
template <> __device__ __forceinline__ float bits16_to_f32<F16>(unsigned short b) { return f16_bits_to_f32(b); }
template <typename T> __device__ __forceinline__ unsigned short f32_to_bits16(float f);
template <> __device__ __forceinline__ unsigned short f32_to_bits16<BF16>(float f) { return f32_to_bf16_bits(f); }
template <> __device__ __forceinline__ unsigned short f32_to_bits16<F16>(float f) { return f32_to_f16_bits(f); }

// run `expr(Tag)` for the element type `dtype` (TG_F32 / TG_BF16 / TG_F16)
#define TG_DISPATCH_DTYPE(dtype, STMT_BF16, STMT_F16, STMT_F32) \
  do {                                                         \
    if ((dtype) == TG_BF16) { STMT_BF16; }                     \
    else if ((dtype) == TG_F16) { STMT_F16; }                  \
    else if ((dtype) == TG_F32) { STMT_F32; }                  \
    else return TG_E_BADARG;                                   \
  } while (0)

#define TG_CHECK_HIP(expr)                 \
  do {                                     \
    hipError_t _e = (expr);                \
    if (_e != hipSuccess) return (int)_e;  \
  } while (0)

static inline int tg_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? TG_OK : (int)e;
}

static inline bool tg_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
