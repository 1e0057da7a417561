// Dev probe: variants of the row kernels (LayerNorm + modulate fp32 -> bf16; RMSNorm bf16 -> bf16) at the DiT shape, to see what the
// access shape costs next to a plain fp32 -> bf16 conversion copy of the same bytes.  hipcc --offload-arch=gfx950 -O3 -shared -fPIC
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16;
typedef bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef bf16 bf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float wsum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ bf16 f2bf(float f) { return (bf16)f; }

// V0: current shape -- lane owns 8 consecutive floats per 512-float segment (two 16-byte loads 16 bytes apart), one 16-byte store
template <int ROWS_PER_BLOCK>
__global__ __launch_bounds__(64 * ROWS_PER_BLOCK) void ln_v0(const float* __restrict__ x, int64_t M, const float* __restrict__ tab, const int* __restrict__ ri, bf16* __restrict__ out) {
  constexpr int C = 3072, NV8 = 6;
  const int lane = threadIdx.x & 63;
  const int64_t m = (int64_t)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
  if (m >= M) return;
  const float* xr = x + m * C;
  f32x4 v[NV8][2];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV8; ++i) { const int c = (i * 64 + lane) * 8; v[i][0] = *(const f32x4*)(xr + c); v[i][1] = *(const f32x4*)(xr + c + 4); }
#pragma unroll
  for (int i = 0; i < NV8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) s += v[i][0][j] + v[i][1][j];
  const float mean = wsum(s) * (1.0f / C);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) { const float a = v[i][0][j] - mean, b = v[i][1][j] - mean; q += a * a + b * b; }
  const float rstd = __builtin_amdgcn_rsqf(wsum(q) * (1.0f / C) + 1e-6f);
  const float* sh = tab + (int64_t)ri[m] * 6 * C;
  const float* sc = sh + C;
  bf16* orow = out + m * C;
#pragma unroll
  for (int i = 0; i < NV8; ++i) {
    const int c0 = (i * 64 + lane) * 8;
    bf16x8 o;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int c = c0 + h * 4;
      f32x4 y = (v[i][h] - mean) * rstd;
      y = y * *(const f32x4*)(sc + c) + *(const f32x4*)(sh + c);
#pragma unroll
      for (int j = 0; j < 4; ++j) o[h * 4 + j] = f2bf(y[j]);
    }
    *(bf16x8*)(orow + c0) = o;
  }
}

// V1: dense 16-byte loads (a wave instruction covers 1 KiB contiguous), 8-byte stores
__global__ __launch_bounds__(256) void ln_v1(const float* __restrict__ x, int64_t M, const float* __restrict__ tab, const int* __restrict__ ri, bf16* __restrict__ out) {
  constexpr int C = 3072, NV = 12;
  const int lane = threadIdx.x & 63;
  const int64_t m = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= M) return;
  const float* xr = x + m * C;
  f32x4 v[NV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = *(const f32x4*)(xr + (i * 64 + lane) * 4);
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) s += v[i][j];
  const float mean = wsum(s) * (1.0f / C);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) { const float a = v[i][j] - mean; q += a * a; }
  const float rstd = __builtin_amdgcn_rsqf(wsum(q) * (1.0f / C) + 1e-6f);
  const float* sh = tab + (int64_t)ri[m] * 6 * C;
  const float* sc = sh + C;
  bf16* orow = out + m * C;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * 64 + lane) * 4;
    f32x4 y = (v[i] - mean) * rstd;
    y = y * *(const f32x4*)(sc + c) + *(const f32x4*)(sh + c);
    bf16x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = f2bf(y[j]);
    *(bf16x4*)(orow + c) = o;
  }
}

// V4: the same bytes with no row structure: fp32 -> bf16 conversion copy, grid-stride, 32 B in / 16 B out per lane and step
__global__ __launch_bounds__(256) void cvt_copy(const float* __restrict__ x, int64_t n8, bf16* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
    const f32x4 a = *(const f32x4*)(x + i * 8), b = *(const f32x4*)(x + i * 8 + 4);
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) { o[j] = f2bf(a[j]); o[4 + j] = f2bf(b[j]); }
    *(bf16x8*)(out + i * 8) = o;
  }
}

// W1: RMSNorm (no rope) bf16 -> bf16 in place, one wave per row of 3072, 4 rows per block (the library's kernel: 128 threads per row, LDS reduction)
__global__ __launch_bounds__(256) void rms_w1(bf16* __restrict__ x, int64_t M, const float* __restrict__ w) {
  constexpr int C = 3072, NV = 6;
  const int lane = threadIdx.x & 63;
  const int64_t m = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= M) return;
  bf16* xr = x + m * C;
  bf16x8 v[NV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = *(const bf16x8*)(xr + (i * 64 + lane) * 8);
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float f = (float)v[i][j]; s += f * f; }
  const float r = __builtin_amdgcn_rsqf(wsum(s) * (1.0f / C) + 1e-6f);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * 64 + lane) * 8;
    const f32x4 w0 = *(const f32x4*)(w + c), w1 = *(const f32x4*)(w + c + 4);
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) { o[j] = f2bf((float)v[i][j] * r * w0[j]); o[4 + j] = f2bf((float)v[i][4 + j] * r * w1[j]); }
    *(bf16x8*)(xr + c) = o;
  }
}

extern "C" void probe_ln(int variant, const float* x, int64_t M, const float* tab, const int* ri, void* out, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (variant == 0) hipLaunchKernelGGL(ln_v0<4>, dim3((M + 3) / 4), dim3(256), 0, s, x, M, tab, ri, (bf16*)out);
  else if (variant == 1) hipLaunchKernelGGL(ln_v1, dim3((M + 3) / 4), dim3(256), 0, s, x, M, tab, ri, (bf16*)out);
  else if (variant == 2) hipLaunchKernelGGL(ln_v0<8>, dim3((M + 7) / 8), dim3(512), 0, s, x, M, tab, ri, (bf16*)out);
  else if (variant == 3) hipLaunchKernelGGL(ln_v0<2>, dim3((M + 1) / 2), dim3(128), 0, s, x, M, tab, ri, (bf16*)out);
  else if (variant == 4) hipLaunchKernelGGL(cvt_copy, dim3(256 * 16), dim3(256), 0, s, x, M * 3072 / 8, (bf16*)out);
  else if (variant == 5) hipLaunchKernelGGL(ln_v0<1>, dim3(M), dim3(64), 0, s, x, M, tab, ri, (bf16*)out);
}
extern "C" void probe_rms(void* x, int64_t M, const float* w, void* stream) {
  hipLaunchKernelGGL(rms_w1, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, (bf16*)x, M, w);
}
