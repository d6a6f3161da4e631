// Shared device helpers for the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

// round-to-nearest-even f32 -> bf16 -> f32: the value a bf16 store followed by a load would give
__device__ __forceinline__ float round_bf16(float x) { return (float)(bf16)x; }

// torch GELU(approximate="tanh"): 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3)))
// (0.5 (1 + tanh u) = sigmoid(2u): one exponential and one hardware reciprocal instead of the library tanhf)
__device__ __forceinline__ float gelu_tanh_f32(float x) {
  const float inner = 0.7978845608028654f * (x + 0.044715f * x * x * x);
  return x * __builtin_amdgcn_rcpf(1.0f + __expf(-2.0f * inner));
}

// x * sigmoid(x).  The hardware reciprocal (1 ulp) replaces the IEEE division's ten-instruction sequence: the
// SwiGLU epilogue runs this on every gate value of a 256 x 256 tile while the matrix pipe waits.
__device__ __forceinline__ float silu_f32(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

// ---- cross-lane reductions without LDS round trips.  A butterfly over lanes i ^ 1, i ^ 2, ... only needs, at each
// step, SOME lane of the partner group (all lanes of a group already agree), so the steps are DPP row operations
// (quad permutes, half-row and row mirrors) and, across the 16-lane rows, the gfx950 lane swaps.
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_HALF_MIRROR = 0x141, DPP_ROW_MIRROR = 0x140;

template <typename OP>
__device__ __forceinline__ float row_reduce(float v, OP op) {   // over the 16 lanes of a DPP row
  v = op(v, dpp_f32<DPP_XOR1>(v));
  v = op(v, dpp_f32<DPP_XOR2>(v));
  v = op(v, dpp_f32<DPP_HALF_MIRROR>(v));
  return op(v, dpp_f32<DPP_ROW_MIRROR>(v));
}
template <typename OP>
__device__ __forceinline__ float wave_reduce(float v, OP op) {
  typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_;
  v = row_reduce(v, op);
  const u32x2_ a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = op(__uint_as_float(a[0]), __uint_as_float(a[1]));
  const u32x2_ b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return op(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float wave_sum(float v) {
  return wave_reduce(v, [](float a, float b) { return a + b; });
}
__device__ __forceinline__ float wave_max(float v) {
  return wave_reduce(v, [](float a, float b) { return fmaxf(a, b); });
}
// The value of lane ^ W (a true butterfly partner, whatever the lanes hold).
template <int W>
__device__ __forceinline__ float xor_lane(float v) {
  static_assert(W == 1 || W == 2 || W == 4 || W == 8, "16 and 32 are the lane swaps");
  if constexpr (W == 1) return dpp_f32<DPP_XOR1>(v);
  if constexpr (W == 2) return dpp_f32<DPP_XOR2>(v);
  if constexpr (W == 8) return dpp_f32<0x128>(v);   // row_ror:8
  // no DPP control flips bit 2 alone: ds_swizzle in bit mode (and 0x1f, or 0, xor 4)
  return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x1F | (4 << 10)));
}

// sum over aligned groups of 8 lanes
__device__ __forceinline__ float group8_sum(float v) {
  v += dpp_f32<DPP_XOR1>(v);
  v += dpp_f32<DPP_XOR2>(v);
  return v + dpp_f32<DPP_HALF_MIRROR>(v);
}

// Host-side error reporting shared by every entry point (see capi.hip).
void bagel_set_error(const char* fmt, ...);

#define BAGEL_CHECK(cond, ...)            \
  do {                                    \
    if (!(cond)) {                        \
      bagel_set_error(__VA_ARGS__);       \
      return 1;                           \
    }                                     \
  } while (0)

// Dynamic LDS above 64 KiB has to be allowed per kernel AND per device.  One of these per kernel remembers the
// devices it has been allowed on (one process normally drives one GPU, but nothing here depends on that).
struct BagelLdsAllowed {
  unsigned long long devices = 0;
};
inline void bagel_allow_lds(BagelLdsAllowed& st, const void* kernel, int bytes) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return;
  const unsigned long long bit = 1ull << (dev & 63);
  if (st.devices & bit) return;
  if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess) st.devices |= bit;
}

// Compute units of the CURRENT device (256 on MI355X), cached per device ordinal.
inline int bagel_cu_count() {
  static int cached[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 256;
  int& slot = cached[dev & 63];
  if (slot == 0) {
    int cu = 0;
    if (hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cu <= 0) cu = 256;
    slot = cu;
  }
  return slot;
}

#define BAGEL_LAUNCH_CHECK()                                               \
  do {                                                                     \
    hipError_t e_ = hipGetLastError();                                     \
    if (e_ != hipSuccess) {                                                \
      bagel_set_error("%s:%d launch failed: %s", __FILE__, __LINE__,       \
                      hipGetErrorString(e_));                              \
      return 2;                                                            \
    }                                                                      \
  } while (0)
