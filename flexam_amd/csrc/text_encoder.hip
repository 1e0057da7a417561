// flexam_amd/csrc/text_encoder.hip -- the HBM-bound pieces of the umT5 text encoder
// (FlexAM/models/wan_text_encoder.py: T5LayerNorm :44-56, T5Attention bias + mask + softmax :91-103,
// T5FeedForward gate product :125-126).  The projections are flexam_gemm_bf16 launches; the encoder runs
// twice per clip on <= 512 tokens, so these are plain one-pass kernels.
#include "common.h"
#include "flexam_hip.h"

namespace {

// out[m, :] = w * x[m, :] * rsqrt(mean(x[m, :]^2) + eps); one 256-thread block per row
template <typename TO>
__global__ __launch_bounds__(256) void t5_norm_kernel(const float* __restrict__ x, int64_t ldx, int C, float eps,
                                                      const float* __restrict__ w, TO* __restrict__ out, int64_t ldo) {
  __shared__ float red[8];
  const float* row = x + (int64_t)blockIdx.x * ldx;
  float q = 0.f;
  for (int c = threadIdx.x; c < C; c += 256) q += row[c] * row[c];
  q = block_sum<256>(q, red);
  const float r = rsqrtf(q / (float)C + eps);
  TO* orow = out + (int64_t)blockIdx.x * ldo;
  for (int c = threadIdx.x; c < C; c += 256) orow[c] = (TO)(w[c] * (row[c] * r));
}

// P[m, :] = softmax(scale * s[m, :] + bias[m, :] over keys with key_mask != 0), bf16, zero padded to Npad columns
__global__ __launch_bounds__(256) void softmax_bias_kernel(const float* __restrict__ s, int64_t lds_, int N, float scale,
                                                           const float* __restrict__ bias, int64_t ldb,
                                                           const float* __restrict__ key_mask, bf16* __restrict__ out, int64_t ldo,
                                                           int Npad) {
  __shared__ float red[8];
  const float* row = s + (int64_t)blockIdx.x * lds_;
  const float* brow = bias ? bias + (int64_t)blockIdx.x * ldb : nullptr;
  auto val = [&](int i) -> float {
    if (key_mask && key_mask[i] == 0.f) return -INFINITY;
    return row[i] * scale + (brow ? brow[i] : 0.f);
  };
  float mx = -INFINITY;
  for (int i = threadIdx.x; i < N; i += 256) mx = fmaxf(mx, val(i));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float sum = 0.f;
  for (int i = threadIdx.x; i < N; i += 256) sum += __expf(val(i) - mx);
  sum = block_sum<256>(sum, red);
  const float inv = 1.f / sum;
  bf16* orow = out + (int64_t)blockIdx.x * ldo;
  for (int i = threadIdx.x; i < Npad; i += 256) orow[i] = f2bf(i < N ? __expf(val(i) - mx) * inv : 0.f);
}

__global__ __launch_bounds__(256) void mul_bf16_kernel(const bf16* __restrict__ a, const bf16* __restrict__ b, bf16* __restrict__ out,
                                                       int64_t n4) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const bf16x4 av = ((const bf16x4*)a)[i], bv = ((const bf16x4*)b)[i];
    bf16x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = f2bf(bf2f(av[e]) * bf2f(bv[e]));
    ((bf16x4*)out)[i] = o;
  }
}

}  // namespace

extern "C" int flexam_t5_norm(const float* x, int64_t ldx, int64_t M, int C, float eps, const float* w, void* out, int64_t ld_out,
                              int out_f32, void* stream) {
  FX_REQUIRE(x && w && out && M > 0 && C > 0 && ldx >= C && ld_out >= C, FLEXAM_E_ARG, "t5_norm: bad arguments");
  if (out_f32)
    hipLaunchKernelGGL(t5_norm_kernel<float>, dim3((unsigned)M), dim3(256), 0, (hipStream_t)stream, x, ldx, C, eps, w, (float*)out, ld_out);
  else
    hipLaunchKernelGGL(t5_norm_kernel<bf16>, dim3((unsigned)M), dim3(256), 0, (hipStream_t)stream, x, ldx, C, eps, w, (bf16*)out, ld_out);
  return flexam_check_launch("flexam_t5_norm");
}

extern "C" int flexam_softmax_bias_rows(const float* s, int64_t ld_s, int64_t M, int N, float scale, const float* bias, int64_t ld_bias,
                                        const float* key_mask, void* out, int64_t ld_out, int Npad, void* stream) {
  FX_REQUIRE(s && out && M > 0 && N > 0 && Npad >= N && Npad <= ld_out, FLEXAM_E_ARG, "softmax_bias_rows: bad arguments");
  hipLaunchKernelGGL(softmax_bias_kernel, dim3((unsigned)M), dim3(256), 0, (hipStream_t)stream, s, ld_s, N, scale, bias, ld_bias,
                     key_mask, (bf16*)out, ld_out, Npad);
  return flexam_check_launch("flexam_softmax_bias_rows");
}

extern "C" int flexam_mul_bf16(const void* a, const void* b, void* out, int64_t n, void* stream) {
  FX_REQUIRE(a && b && out && n > 0 && n % 4 == 0, FLEXAM_E_ARG, "mul_bf16: null pointer or n %% 4 != 0");
  int64_t g = (n / 4 + 255) / 256;
  hipLaunchKernelGGL(mul_bf16_kernel, dim3((unsigned)(g > 4096 ? 4096 : g)), dim3(256), 0, (hipStream_t)stream, (const bf16*)a,
                     (const bf16*)b, (bf16*)out, n / 4);
  return flexam_check_launch("flexam_mul_bf16");
}
