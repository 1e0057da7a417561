// flexam_amd/csrc/common.h -- shared device helpers and host-side error plumbing for
// libflexam_hip.so (gfx950 / MI355X only; wave64, MFMA, 160 KiB LDS).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

#define FLEXAM_OK 0
#define FLEXAM_E_ARG (-1)
#define FLEXAM_E_SHAPE (-2)
#define FLEXAM_E_ARCH (-3)
#define FLEXAM_E_LAUNCH (-4)

// host: record an error string and return the code (thread-local; see api.hip)
int flexam_fail(int code, const char* fmt, ...);
int flexam_check_launch(const char* what);
// host: per-device state (one process may drive several GPUs): index of the current device, clamped to the table size, and
// its CU count rounded down to a multiple of 8 (one workgroup per CU, blockIdx & 7 = XCD)
#define FLEXAM_MAX_DEVICES 16
int flexam_current_device();
int flexam_num_cus();

#define FX_REQUIRE(cond, code, ...)                     \
  do {                                                  \
    if (!(cond)) return flexam_fail((code), __VA_ARGS__); \
  } while (0)

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

__device__ __forceinline__ float bf2f(bf16 v) { return (float)v; }
__device__ __forceinline__ bf16 f2bf(float v) { return (bf16)v; }

__device__ __forceinline__ float gelu_tanh(float x) {
  // torch.nn.GELU(approximate='tanh'): 0.5 x (1 + tanh(u)), u = sqrt(2/pi) (x + 0.044715 x^3).  With
  // 0.5 (1 + tanh(u)) = 1 / (1 + exp(-2u)) this is x / (1 + exp2(x (a + b x^2))), a = -2 log2(e) sqrt(2/pi), b = 0.044715 a:
  // 5 plain VALU operations and 2 transcendentals per element (exp2 -> inf gives x * 0 for very negative x, -> 0 gives x).
  const float a = -2.302208198f, b = -0.1029432396f;
  const float e = __builtin_amdgcn_exp2f(x * __builtin_fmaf(b, x * x, a));
  return x * __builtin_amdgcn_rcpf(1.0f + e);
}

__device__ __forceinline__ float silu(float x) {
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-x * 1.4426950408889634f));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// block-wide sum of one float per thread; `red` is LDS scratch of >= 32 floats; NT threads
template <int NT>
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < NT / 64; ++i) t += red[i];
  return t;
}

// The lane id, recomputed where it is needed (2 VALU instructions) instead of kept: code behind the K loop (the next unit's staging
// offsets, the epilogue's addresses) otherwise keeps threadIdx.x / the lane id alive ACROSS the loop, where all 256 registers
// are taken: hipcc spills them and reloads them in front of the epilogue with an s_waitcnt vmcnt(0) that also drains the next
// unit's two prefetched K blocks.  `volatile`: not hoisted, not merged with another copy.
__device__ __forceinline__ int fresh_lane() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}
