// flexam_amd/csrc/conv_cl.hip -- channels-last helpers around the implicit-GEMM convolutions.
//
// Convolutions (the DiT's cnn-block, FX.py:680-705, and the VAE decoder, VAE.py) run as
// flexam_gemm_bf16 over a *spatially padded channels-last* activation image
//     img[f][hp][wp][c]   hp in [0, H+2), wp in [0, W+2), zero border, c padded to 64
// so that tap (dh, dw) of a 3x3 kernel is a constant element offset
//     ((dh-1)*(W+2) + (dw-1)) * C        from the output position's own row,
// passed to the GEMM as its per-K-block A offset table.  Outputs are produced for every padded
// position; border rows hold garbage and are never read back (the kernels below only touch the
// interior).  Coalescing: a position's channels are contiguous, so every 64-channel K block is
// one 128-byte line per row -- exactly the GEMM tile row.
#include "common.h"
#include "flexam_hip.h"

namespace {

inline int grid_for(int64_t total, int block) {
  int64_t g = (total + block - 1) / block;
  return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

// src [C,F,H,W] -> dst[((f*Hp + h+1)*Wp + w+1)*Cp + c0 + c]   (interior only)
template <typename T>
__global__ __launch_bounds__(256) void pack_cl_kernel(const T* __restrict__ src, int C, int F, int H, int W,
                                                      bf16* __restrict__ dst, int Cp, int c0) {
  const int Hp = H + 2, Wp = W + 2;
  const int64_t total = (int64_t)F * H * W * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    int64_t t = i / C;
    const int w = (int)(t % W);
    t /= W;
    const int h = (int)(t % H);
    const int f = (int)(t / H);
    const T v = src[(((int64_t)c * F + f) * H + h) * W + w];
    dst[(((int64_t)f * Hp + h + 1) * Wp + w + 1) * Cp + c0 + c] = f2bf((float)v);
  }
}

// src rows [(f,hp,wp)][ld] (interior positions) -> dst [C,F,H,W] fp32
template <typename T>
__global__ __launch_bounds__(256) void unpack_cl_kernel(const T* __restrict__ src, int64_t ld, int C, int F, int H, int W,
                                                        float* __restrict__ dst) {
  const int Hp = H + 2, Wp = W + 2;
  const int64_t total = (int64_t)C * F * H * W;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int w = (int)(i % W);
    int64_t t = i / W;
    const int h = (int)(t % H);
    t /= H;
    const int f = (int)(t % F);
    const int c = (int)(t / F);
    dst[i] = (float)src[(((int64_t)f * Hp + h + 1) * Wp + w + 1) * ld + c];
  }
}

// GroupNorm statistics over (channels of the group) x (all interior positions of all frames):
// one workgroup per group, two passes (mean, then centred variance) -> stats[g] = {mean, rstd}
__global__ __launch_bounds__(256) void groupnorm_stats_kernel(const float* __restrict__ x, int64_t ld, int F, int H, int W,
                                                              int cpg, float eps, float* __restrict__ stats) {
  __shared__ float red[8];
  const int g = blockIdx.x;
  const int Hp = H + 2, Wp = W + 2;
  const int64_t npos = (int64_t)F * H * W;
  const int64_t total = npos * cpg;
  auto at = [&](int64_t i) -> float {
    const int c = (int)(i % cpg);
    int64_t t = i / cpg;
    const int w = (int)(t % W);
    t /= W;
    const int h = (int)(t % H);
    const int f = (int)(t / H);
    return x[(((int64_t)f * Hp + h + 1) * Wp + w + 1) * ld + g * cpg + c];
  };
  float s = 0.f;
  for (int64_t i = threadIdx.x; i < total; i += 256) s += at(i);
  const float mean = block_sum<256>(s, red) / (float)total;
  float q = 0.f;
  for (int64_t i = threadIdx.x; i < total; i += 256) {
    const float d = at(i) - mean;
    q += d * d;
  }
  const float var = block_sum<256>(q, red) / (float)total;
  if (threadIdx.x == 0) {
    stats[2 * g] = mean;
    stats[2 * g + 1] = rsqrtf(var + eps);
  }
}

// y = silu((x - mean_g) rstd_g gamma + beta) [+ residual]  -> bf16 padded channels-last image (interior)
__global__ __launch_bounds__(256) void groupnorm_silu_cl_kernel(const float* __restrict__ x, int64_t ld, int C, int F, int H,
                                                                int W, int cpg, const float* __restrict__ stats,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                const bf16* __restrict__ residual, int res_cp,
                                                                bf16* __restrict__ dst, int Cp) {
  const int Hp = H + 2, Wp = W + 2;
  const int64_t total = (int64_t)F * H * W * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    int64_t t = i / C;
    const int w = (int)(t % W);
    t /= W;
    const int h = (int)(t % H);
    const int f = (int)(t / H);
    const int64_t pos = ((int64_t)f * Hp + h + 1) * Wp + w + 1;
    const int g = c / cpg;
    float v = (x[pos * ld + c] - stats[2 * g]) * stats[2 * g + 1] * gamma[c] + beta[c];
    v = silu(v);
    if (residual) v += bf2f(residual[pos * res_cp + c]);
    dst[pos * Cp + c] = f2bf(v);
  }
}

}  // namespace

extern "C" int flexam_pack_cl(const void* src, int src_is_bf16, int C, int F, int H, int W, void* dst, int Cp, int c0, void* stream) {
  FX_REQUIRE(src && dst, FLEXAM_E_ARG, "pack_cl: null pointer");
  FX_REQUIRE(C > 0 && F > 0 && H > 0 && W > 0 && c0 >= 0 && c0 + C <= Cp, FLEXAM_E_SHAPE, "pack_cl: channels %d+%d exceed Cp %d", c0, C, Cp);
  const int64_t total = (int64_t)F * H * W * C;
  if (src_is_bf16)
    hipLaunchKernelGGL(pack_cl_kernel<bf16>, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16*)src, C, F, H, W, (bf16*)dst, Cp, c0);
  else
    hipLaunchKernelGGL(pack_cl_kernel<float>, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, (const float*)src, C, F, H, W, (bf16*)dst, Cp, c0);
  return flexam_check_launch("flexam_pack_cl");
}

extern "C" int flexam_unpack_cl(const void* src, int src_is_bf16, int64_t ld, int C, int F, int H, int W, float* dst, void* stream) {
  FX_REQUIRE(src && dst, FLEXAM_E_ARG, "unpack_cl: null pointer");
  FX_REQUIRE(C > 0 && C <= ld && F > 0 && H > 0 && W > 0, FLEXAM_E_SHAPE, "unpack_cl: bad shape");
  const int64_t total = (int64_t)C * F * H * W;
  if (src_is_bf16)
    hipLaunchKernelGGL(unpack_cl_kernel<bf16>, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16*)src, ld, C, F, H, W, dst);
  else
    hipLaunchKernelGGL(unpack_cl_kernel<float>, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, (const float*)src, ld, C, F, H, W, dst);
  return flexam_check_launch("flexam_unpack_cl");
}

extern "C" int flexam_groupnorm_silu_cl(const float* x, int64_t ld, int C, int F, int H, int W, int groups, float eps,
                                        const float* gamma, const float* beta, float* stats, const void* residual, int res_cp,
                                        void* dst, int Cp, void* stream) {
  FX_REQUIRE(x && gamma && beta && stats && dst, FLEXAM_E_ARG, "groupnorm_silu_cl: null pointer");
  FX_REQUIRE(groups > 0 && C % groups == 0 && C <= Cp && C <= ld, FLEXAM_E_SHAPE, "groupnorm_silu_cl: C=%d groups=%d Cp=%d", C, groups, Cp);
  const int cpg = C / groups;
  hipLaunchKernelGGL(groupnorm_stats_kernel, dim3(groups), dim3(256), 0, (hipStream_t)stream, x, ld, F, H, W, cpg, eps, stats);
  const int64_t total = (int64_t)F * H * W * C;
  hipLaunchKernelGGL(groupnorm_silu_cl_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, x, ld, C, F, H, W, cpg,
                     stats, gamma, beta, (const bf16*)residual, res_cp, (bf16*)dst, Cp);
  return flexam_check_launch("flexam_groupnorm_silu_cl");
}
