// Dev probe: does the matrix pipe's throughput under the power limit depend on how often an MFMA operand CHANGES between consecutive
// instructions?  512 threads per CU, 2 waves per SIMD, 4 independent accumulator chains per wave of v_mfma_f32_32x32x16_bf16 on random
// bf16 data held in registers (no memory traffic in the loop).  mode: which operand registers consecutive MFMAs use
//   0: A[i], B[i]      both change every instruction          (attention's S chain: K fragment and Q fragment per 16-wide d slice)
//   1: A[i], B[i / 2]  B shared by pairs
//   2: A[i], B[0]      B constant
//   3: A[0], B[0]      both constant (data still random, nothing toggles but the accumulators)
//   4: all-zero operands
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(512, 2) void probe(const bf16x8* __restrict__ src, float* __restrict__ out, int iters) {
  bf16x8 a[8], b[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    a[i] = src[(threadIdx.x * 16 + i) & 4095];
    b[i] = src[(threadIdx.x * 16 + 8 + i) & 4095];
    if (MODE == 4) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { a[i][j] = (__bf16)0.f; b[i][j] = (__bf16)0.f; }
    }
  }
  f32x16 acc[4];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[c][e] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int ia = MODE == 3 ? 0 : i;
      const int ib = MODE == 0 ? i : MODE == 1 ? (i / 2) * 2 : 0;
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ia], b[(ib + c) & 7], acc[c], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(a[i]), "+v"(b[i]));
  }
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int e = 0; e < 16; ++e) s += acc[c][e];
  if (s == 123.456f) out[threadIdx.x] = s;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
// the same FLOPs per loop iteration on v_mfma_f32_16x16x32_bf16 (the GEMMs' instruction): 16 independent chains
template <int ZERO>
__global__ __launch_bounds__(512, 2) void probe16(const bf16x8* __restrict__ src, float* __restrict__ out, int iters) {
  bf16x8 a[8], b[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    a[i] = src[(threadIdx.x * 16 + i) & 4095];
    b[i] = src[(threadIdx.x * 16 + 8 + i) & 4095];
    if (ZERO) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { a[i][j] = (__bf16)0.f; b[i][j] = (__bf16)0.f; }
    }
  }
  f32x4 acc[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int c = 0; c < 8; ++c) acc[(c + 8 * (i & 1)) & 15] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[c], acc[(c + 8 * (i & 1)) & 15], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(a[i]), "+v"(b[i]));
  }
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < 16; ++c)
#pragma unroll
    for (int e = 0; e < 4; ++e) s += acc[c][e];
  if (s == 123.456f) out[threadIdx.x] = s;
}

extern "C" void run_probe(int mode, const void* src, void* out, int iters, int blocks, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  const bf16x8* p = (const bf16x8*)src;
  float* o = (float*)out;
  switch (mode) {
    case 0: hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(512), 0, s, p, o, iters); break;
    case 1: hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(512), 0, s, p, o, iters); break;
    case 2: hipLaunchKernelGGL(probe<2>, dim3(blocks), dim3(512), 0, s, p, o, iters); break;
    case 3: hipLaunchKernelGGL(probe<3>, dim3(blocks), dim3(512), 0, s, p, o, iters); break;
    case 5: hipLaunchKernelGGL(probe16<0>, dim3(blocks), dim3(512), 0, s, p, o, iters); break;
    case 6: hipLaunchKernelGGL(probe16<1>, dim3(blocks), dim3(512), 0, s, p, o, iters); break;
    default: hipLaunchKernelGGL(probe<4>, dim3(blocks), dim3(512), 0, s, p, o, iters); break;
  }
}
