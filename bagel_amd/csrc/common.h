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

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
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
