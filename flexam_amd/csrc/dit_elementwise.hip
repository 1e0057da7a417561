// flexam_amd/csrc/dit_elementwise.hip -- the HBM-bound kernels of the FlexAM DiT block and the
// sampler step.  All are one-pass, 16-byte vectorised, fp32 math with one rounding to bf16.
//
// Row kernels use one 128-thread workgroup (2 waves) per token row: a 3072-wide row is exactly
// 3 x 128 vectors of 8 elements, kept in registers between the reduction and the write.
#include "common.h"
#include "flexam_hip.h"

namespace {

constexpr int RT = 128;   // threads per row

__device__ __forceinline__ float row_sum(float v, float* red) { return block_sum<RT>(v, red); }

// ------------------------------------------------------------------------------------------
// Same LayerNorm + modulation, ONE WAVE PER ROW (C = 512*NV8, NV8 <= 8): every lane keeps 8*NV8 values in registers, both
// reductions are wave shuffles (no LDS, no barrier), one row per 64-thread workgroup.  The block-per-row form below spends
// its time in two block reductions per 12 KiB row; this one keeps 12 x 16 B loads per lane in flight.
// ------------------------------------------------------------------------------------------
// FP8: the output row is written as OCP e4m3 bytes with a per-row scale (absmax / 448) -- the A operand of flexam_gemm_fp8 -- instead
// of bf16: the row is in registers anyway, so the quantiser costs one more wave reduction and no second pass over HBM.
template <int NV8, bool FP8 = false>
__global__ __launch_bounds__(64) void ln_modulate_wave_kernel(const float* __restrict__ x, int64_t ldx, int64_t M, float eps,
                                                              const float* __restrict__ shift, const float* __restrict__ scale,
                                                              int64_t tab_ld, const int32_t* __restrict__ row_index,
                                                              int64_t rows_per_batch, const float* __restrict__ ln_w,
                                                              const float* __restrict__ ln_b, bf16* __restrict__ out, int64_t ldo,
                                                              float* __restrict__ row_scale = nullptr, float* __restrict__ next_scale = nullptr,
                                                              float next_wnorm = 0.f, float next_bias = 0.f) {
  // Access shape (tools/ab_rowkernels.py, r3): a lane owns 4 consecutive floats of every 256-float segment, so one load
  // instruction of the wave covers 1 KiB without gaps (the earlier 8-floats-per-lane shape read it as two half-dense
  // instructions: -9.6 % time), and ONE wave per workgroup, so no row waits for the slowest of four (-9.3 % on its own).
  constexpr int C = 512 * NV8, NV = 2 * NV8;
  const int lane = threadIdx.x;
  const int64_t m = blockIdx.x;
  const float* xr = x + m * ldx;
  f32x4 v[NV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = *(const f32x4*)(xr + (i * 64 + lane) * 4);
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) s += v[i][j];
  const float mean = wave_sum(s) * (1.0f / C);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float a = v[i][j] - mean;
      q += a * a;
    }
  const float rstd = __builtin_amdgcn_rsqf(wave_sum(q) * (1.0f / C) + eps);
  const float* sh = nullptr;
  const float* sc = nullptr;
  if (shift) {
    const int64_t r = row_index ? (int64_t)row_index[m] : m / rows_per_batch;
    sh = shift + r * tab_ld;
    sc = scale + r * tab_ld;
  }
  float amax = 0.f, ssq = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * 64 + lane) * 4;
    f32x4 y = (v[i] - mean) * rstd;
    if (ln_w) y = y * *(const f32x4*)(ln_w + c) + *(const f32x4*)(ln_b + c);
    if (sh) y = y * *(const f32x4*)(sc + c) + *(const f32x4*)(sh + c);
    if constexpr (FP8) {
      v[i] = y;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        amax = fmaxf(amax, fabsf(y[j]));
        ssq += y[j] * y[j];
      }
    } else {
      bf16x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = f2bf(y[j]);
      *(bf16x4*)(out + m * ldo + c) = o;
    }
  }
  if constexpr (FP8) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
    const float qs = amax > 0.f ? amax * (1.0f / 448.0f) : 1.0f;
    const float inv = 1.0f / qs;
    if (lane == 0) row_scale[m] = qs;
    if (next_scale) {
      // A scale for the e4m3 OUTPUT of the GEMM this row feeds, known before that GEMM runs: |gelu(a . w_j + b_j)| <= |a|_2 |w_j|_2 + |b_j|
      // (Cauchy-Schwarz; |a|_2 of the quantised row <= 1.07 |y|_2: 2^-4 relative rounding + subnormal steps), so the consumer of
      // that output needs no absmax pass over it.  The bound is loose (~ sqrt(C) above a typical value), which an 8-bit FLOAT
      // absorbs: values keep their 3 mantissa bits down to 2^-14 of the bound.
      const float l2 = __builtin_sqrtf(wave_sum(ssq));
      if (lane == 0) next_scale[m] = fmaxf((l2 * 1.07f * next_wnorm + next_bias) * (1.0f / 448.0f), 1e-30f);
    }
    uint8_t* qrow = (uint8_t*)out + m * ldo;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      int w = 0;
      w = __builtin_amdgcn_cvt_pk_fp8_f32(v[i][0] * inv, v[i][1] * inv, w, false);
      w = __builtin_amdgcn_cvt_pk_fp8_f32(v[i][2] * inv, v[i][3] * inv, w, true);
      *(unsigned*)(qrow + (i * 64 + lane) * 4) = (unsigned)w;
    }
  }
}

// ------------------------------------------------------------------------------------------
// LayerNorm (no affine, or affine for norm3) * scale + shift  ->  bf16
//   scale/shift rows come from a small fp32 table; row = row_index[m] or m / rows_per_batch.
//   The table already holds (1 + scale) and (shift + density shift): see mod_table_kernel.
// ------------------------------------------------------------------------------------------
template <int VPT>
__global__ __launch_bounds__(RT) void ln_modulate_kernel(const float* __restrict__ x, int64_t ldx, int C, float eps,
                                                         const float* __restrict__ shift, const float* __restrict__ scale,
                                                         int64_t tab_ld, const int32_t* __restrict__ row_index,
                                                         int64_t rows_per_batch, const float* __restrict__ ln_w,
                                                         const float* __restrict__ ln_b, bf16* __restrict__ out, int64_t ldo) {
  __shared__ float red[8];
  const int64_t m = blockIdx.x;
  const float* xr = x + m * ldx;
  const int nvec = C >> 3;
  f32x4 v[VPT][2];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const int vec = threadIdx.x + i * RT;
    if (vec < nvec) {
      v[i][0] = *(const f32x4*)(xr + vec * 8);
      v[i][1] = *(const f32x4*)(xr + vec * 8 + 4);
    } else {
      v[i][0] = v[i][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) s += v[i][0][j] + v[i][1][j];
  }
  const float mean = row_sum(s, red) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const int vec = threadIdx.x + i * RT;
    if (vec < nvec) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float a = v[i][0][j] - mean, b = v[i][1][j] - mean;
        q += a * a + b * b;
      }
    }
  }
  const float rstd = __builtin_amdgcn_rsqf(row_sum(q, red) / (float)C + eps);
  const float* sh = nullptr;
  const float* sc = nullptr;
  if (shift) {
    const int64_t r = row_index ? (int64_t)row_index[m] : m / rows_per_batch;
    sh = shift + r * tab_ld;
    sc = scale + r * tab_ld;
  }
  bf16* orow = out + m * ldo;
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const int vec = threadIdx.x + i * RT;
    if (vec >= nvec) continue;
    bf16x8 o;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int c = vec * 8 + h * 4;
      f32x4 y = (v[i][h] - mean) * rstd;
      if (ln_w) y = y * *(const f32x4*)(ln_w + c) + *(const f32x4*)(ln_b + c);
      if (sh) y = y * *(const f32x4*)(sc + c) + *(const f32x4*)(sh + c);
#pragma unroll
      for (int j = 0; j < 4; ++j) o[h * 4 + j] = f2bf(y[j]);
    }
    *(bf16x8*)(orow + vec * 8) = o;
  }
}

// ------------------------------------------------------------------------------------------
// x[m, :] += y_bf16[m, :] * gate[row(m), :]        (fp32 residual stream, in place)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gate_residual_kernel(float* __restrict__ x, int64_t ldx, const bf16* __restrict__ y,
                                                            int64_t ldy, const float* __restrict__ gate, int64_t gate_ld,
                                                            const int32_t* __restrict__ row_index, int64_t rows_per_batch,
                                                            int64_t M, int C) {
  const int nvec = C >> 3;
  const int64_t total = M * nvec;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = i / nvec;
    const int c = (int)(i - m * nvec) * 8;
    const bf16x8 yv = *(const bf16x8*)(y + m * ldy + c);
    float* xp = x + m * ldx + c;
    f32x4 a = *(const f32x4*)xp, b = *(const f32x4*)(xp + 4);
    f32x4 g0 = {1.f, 1.f, 1.f, 1.f}, g1 = g0;
    if (gate) {
      const int64_t r = row_index ? (int64_t)row_index[m] : m / rows_per_batch;
      g0 = *(const f32x4*)(gate + r * gate_ld + c);
      g1 = *(const f32x4*)(gate + r * gate_ld + c + 4);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      a[j] += bf2f(yv[j]) * g0[j];
      b[j] += bf2f(yv[4 + j]) * g1[j];
    }
    *(f32x4*)xp = a;
    *(f32x4*)(xp + 4) = b;
  }
}

// ------------------------------------------------------------------------------------------
// WanRMSNorm over the whole row (all heads) + 3-axis RoPE on interleaved pairs, bf16 in/out.
//   blockIdx.y selects the tensor (0: q, 1: k).  cos/sin: [tokens_per_batch, head_dim/2] fp32,
//   identity rows for pass-through tokens; null -> no rotation (cross-attention q).
// ------------------------------------------------------------------------------------------
struct RmsRopeArgs {
  const bf16* in[3];
  bf16* out[3];
  const float* w[3];
  int64_t ld_in[3], ld_out[3];
  // SCATTER instance (sequence-parallel send layout): element (m, col) of tensor `which` goes to
  //   out[which] + (m / tokens_per_batch) * out_bs + (m % tokens_per_batch) * ld_out[which] + (col / col_block) * block_stride + col % col_block
  // i.e. the row is cut into column blocks (one per destination rank's head group) that land block_stride apart; tensor 2 (v) is
  // copied through without a norm
  int64_t out_bs, block_stride;
  int col_block;
  int map[3];              // blockIdx.y -> tensor slot
};

template <int VPT, bool SCATTER>
__global__ __launch_bounds__(RT) void rmsnorm_rope_kernel(RmsRopeArgs a, int C, float eps, const float* __restrict__ cs,
                                                          const float* __restrict__ sn, int64_t tokens_per_batch,
                                                          int64_t token_offset, int head_dim) {
  __shared__ float red[8];
  const int which = SCATTER ? a.map[blockIdx.y] : (int)blockIdx.y;
  const int64_t m = blockIdx.x;
  const bf16* xr = a.in[which] + m * a.ld_in[which];
  const int nvec = C >> 3;
  if constexpr (SCATTER) {
    if (which == 2) {                                   // v: plain copy into the scattered layout
      bf16* ob = a.out[2] + (m / tokens_per_batch) * a.out_bs + (m % tokens_per_batch) * a.ld_out[2];
      for (int vec = threadIdx.x; vec < nvec; vec += RT) {
        const int c = vec * 8;
        *(bf16x8*)(ob + (int64_t)(c / a.col_block) * a.block_stride + c % a.col_block) = *(const bf16x8*)(xr + c);
      }
      return;
    }
  }
  bf16x8 v[VPT];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const int vec = threadIdx.x + i * RT;
    if (vec < nvec) {
      v[i] = *(const bf16x8*)(xr + vec * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float f = bf2f(v[i][j]);
        s += f * f;
      }
    }
  }
  const float r = __builtin_amdgcn_rsqf(row_sum(s, red) / (float)C + eps);
  const float* w = a.w[which];
  bf16* orow = SCATTER ? a.out[which] + (int64_t)((uint32_t)m / (uint32_t)tokens_per_batch) * a.out_bs +
                             (int64_t)((uint32_t)m % (uint32_t)tokens_per_batch) * a.ld_out[which]
                       : a.out[which] + m * a.ld_out[which];
  const int half = head_dim >> 1;
  // 32-bit: a 64-bit modulo per thread costs as much as the rest of a thread's work here (three 16-byte loads and stores)
  const int64_t tok = token_offset + (int64_t)((uint32_t)m % (uint32_t)tokens_per_batch);
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const int vec = threadIdx.x + i * RT;
    if (vec >= nvec) continue;
    const int c = vec * 8;
    const f32x4 w0 = *(const f32x4*)(w + c), w1 = *(const f32x4*)(w + c + 4);
    float y[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      y[j] = bf2f(v[i][j]) * r * w0[j];
      y[4 + j] = bf2f(v[i][4 + j]) * r * w1[j];
    }
    if (cs) {
      const int pair0 = (c % head_dim) >> 1;          // 4 consecutive pairs
      const f32x4 co = *(const f32x4*)(cs + tok * half + pair0);
      const f32x4 si = *(const f32x4*)(sn + tok * half + pair0);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float re = y[2 * j], im = y[2 * j + 1];
        y[2 * j] = re * co[j] - im * si[j];
        y[2 * j + 1] = re * si[j] + im * co[j];
      }
    }
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = f2bf(y[j]);
    if constexpr (SCATTER) *(bf16x8*)(orow + (int64_t)(c / a.col_block) * a.block_stride + c % a.col_block) = o;
    else *(bf16x8*)(orow + c) = o;
  }
}

// ------------------------------------------------------------------------------------------
// Modulation table:  out[blk][r][j][:] = mod[blk][j][:] + e[r][j][:] + (scale_mask>>j & 1)
//                                        + (dens_slot[j] >= 0 ? mdens[blk][slot][:] + dens[r / rows_per_batch][slot][:] : 0)
// (FX.py:444-449, 452, 464, 500-506: `modulation + e0`, `1 + e[1]`, `+ density_emb[k]`)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mod_table_kernel(const float* __restrict__ mod, const float* __restrict__ e,
                                                        const float* __restrict__ mdens, const float* __restrict__ dens,
                                                        float* __restrict__ out, int nblk, int R, int nj, int nslot, int C,
                                                        int rows_per_batch, int scale_mask, int dens_slots /*4 bits per j, 0xF = none*/) {
  const int64_t total = (int64_t)nblk * R * nj * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    int64_t t = i / C;
    const int j = (int)(t % nj);
    t /= nj;
    const int r = (int)(t % R);
    const int blk = (int)(t / R);
    float v = mod[((int64_t)blk * nj + j) * C + c] + e[((int64_t)r * nj + j) * C + c];
    if ((scale_mask >> j) & 1) v += 1.0f;
    const int slot = (dens_slots >> (4 * j)) & 0xF;
    if (slot != 0xF) v += mdens[((int64_t)blk * nslot + slot) * C + c] + dens[((int64_t)(r / rows_per_batch) * nslot + slot) * C + c];
    out[i] = v;
  }
}

// ------------------------------------------------------------------------------------------
// Small-M fp32 linear: y[M,N] = act_in(x[M,K]) . W[N,K]^T + b   (M <= 8; W bf16 or fp32)
// The time / density embedding MLPs run in fp32 in the reference (FX.py:928-955).
// One wave per output column n; lanes stride K.
// ------------------------------------------------------------------------------------------
template <typename WT, int MAXM, int NPW>
__global__ __launch_bounds__(256) void small_linear_kernel(const float* __restrict__ x, int64_t ldx, const WT* __restrict__ W,
                                                           int64_t ldw, const float* __restrict__ b, float* __restrict__ y,
                                                           int64_t ldy, int M, int N, int K, int silu_in) {
  // one wave per NPW consecutive outputs n: every x element it loads feeds NPW weight rows (the x rows come from L2 once per
  // wave, so NPW = 4 at M = 32 keeps that traffic below the weight stream's)
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n0 = (blockIdx.x * 4 + wave) * NPW;
  if (n0 >= N) return;
  float acc[MAXM][NPW];
#pragma unroll
  for (int m = 0; m < MAXM; ++m)
#pragma unroll
    for (int q = 0; q < NPW; ++q) acc[m][q] = 0.f;
  for (int k = lane * 4; k < K; k += 256) {
    float w[NPW][4];
#pragma unroll
    for (int q = 0; q < NPW; ++q) {
      const WT* wr = W + (int64_t)min(n0 + q, N - 1) * ldw;
      if constexpr (sizeof(WT) == 2) {
        const bf16x4 t = *(const bf16x4*)(wr + k);
#pragma unroll
        for (int j = 0; j < 4; ++j) w[q][j] = bf2f(t[j]);
      } else {
        const f32x4 t = *(const f32x4*)(wr + k);
#pragma unroll
        for (int j = 0; j < 4; ++j) w[q][j] = t[j];
      }
    }
#pragma unroll
    for (int m = 0; m < MAXM; ++m) {
      if (m < M) {
        f32x4 xv = *(const f32x4*)(x + (int64_t)m * ldx + k);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float t = xv[j];
          if (silu_in) t = t / (1.0f + __expf(-t));
#pragma unroll
          for (int q = 0; q < NPW; ++q) acc[m][q] += t * w[q][j];
        }
      }
    }
  }
#pragma unroll
  for (int m = 0; m < MAXM; ++m)
#pragma unroll
    for (int q = 0; q < NPW; ++q) {
      const float t = wave_sum(acc[m][q]);
      if (lane == 0 && m < M && n0 + q < N) y[(int64_t)m * ldy + n0 + q] = t + (b ? b[n0 + q] : 0.f);
    }
}

// sinusoidal_embedding_1d (FX.py:31-41): out[r][:half] = cos(t_r * f_i), out[r][half:] = sin(...), fp64 math
__global__ void sinusoid_kernel(const float* __restrict__ t, float* __restrict__ out, int R, int dim) {
  const int half = dim >> 1;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= R * half) return;
  const int r = i / half, k = i - r * half;
  const double f = pow(10000.0, -(double)k / (double)half);
  const double a = (double)t[r] * f;
  out[(int64_t)r * dim + k] = (float)cos(a);
  out[(int64_t)r * dim + half + k] = (float)sin(a);
}

// ------------------------------------------------------------------------------------------
// patchify: src[C, F, H, W] (fp32 or bf16) -> dst[(f, h/2, w/2), col0 + c*4 + ph*2 + pw] bf16
// = the im2col of a kernel=stride=(1,2,2) conv (FX.py:624-625, 676) in weight.flatten(1) order.
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void patchify_kernel(const T* __restrict__ src, int C, int F, int H, int W,
                                                       bf16* __restrict__ dst, int64_t ldd, int col0, int64_t row0) {
  const int hw2 = (H / 2) * (W / 2);
  const int64_t total = (int64_t)F * hw2 * C * 4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int col = (int)(i % (C * 4));
    const int64_t tok = i / (C * 4);
    const int c = col >> 2, ph = (col >> 1) & 1, pw = col & 1;
    const int f = (int)(tok / hw2);
    const int rem = (int)(tok - (int64_t)f * hw2);
    const int h2 = rem / (W / 2), w2 = rem - h2 * (W / 2);
    const T v = src[(((int64_t)c * F + f) * H + (h2 * 2 + ph)) * W + (w2 * 2 + pw)];
    dst[(row0 + tok) * ldd + col0 + col] = f2bf((float)v);
  }
}

// ------------------------------------------------------------------------------------------
// unpatchify (FX.py:1126-1149): tokens [L, 4*C] (col = (ph*2+pw)*C + c) -> [C, F, H, W]
// ------------------------------------------------------------------------------------------
template <typename TO>
__global__ __launch_bounds__(256) void unpatchify_kernel(const float* __restrict__ tok, int64_t ldt, int64_t tok0, int C,
                                                         int F, int H, int W, TO* __restrict__ dst) {
  const int64_t total = (int64_t)C * F * H * W;
  const int hw2 = (H / 2) * (W / 2);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int w = (int)(i % W);
    int64_t t = i / W;
    const int h = (int)(t % H);
    t /= H;
    const int f = (int)(t % F);
    const int c = (int)(t / F);
    const int64_t token = tok0 + (int64_t)f * hw2 + (h >> 1) * (W / 2) + (w >> 1);
    const int col = (((h & 1) << 1) | (w & 1)) * C + c;
    dst[i] = (TO)tok[token * ldt + col];
  }
}

// ------------------------------------------------------------------------------------------
// Fused sampler step (PIPE.py:926-934): unpatchify both CFG rows, v = u + g (c - u),
// x += (sigma_next - sigma) v, x = (1 - mask) x_known + mask x.   Latents fp32 [C, F, H, W].
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cfg_euler_blend_kernel(const float* __restrict__ tok_u, const float* __restrict__ tok_c,
                                                              int64_t ldt, int64_t tok0, float guidance, float dt,
                                                              float* __restrict__ latents, const float* __restrict__ known,
                                                              const float* __restrict__ mask, int C, int F, int H, int W,
                                                              float* __restrict__ v_out) {
  const int64_t total = (int64_t)C * F * H * W;
  const int hw2 = (H / 2) * (W / 2);
  const int64_t fhw = (int64_t)F * H * W;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int w = (int)(i % W);
    int64_t t = i / W;
    const int h = (int)(t % H);
    t /= H;
    const int f = (int)(t % F);
    const int c = (int)(t / F);
    const int64_t token = tok0 + (int64_t)f * hw2 + (h >> 1) * (W / 2) + (w >> 1);
    const int col = (((h & 1) << 1) | (w & 1)) * C + c;
    float v = tok_u[token * ldt + col];
    if (tok_c) v = v + guidance * (tok_c[token * ldt + col] - v);
    if (v_out) {                       // multistep samplers: hand the guided velocity to scheduler.step
      v_out[i] = v;
      continue;
    }
    float x = latents[i] + dt * v;
    if (mask) {
      const float mk = mask[i % fhw];
      x = (1.0f - mk) * known[i] + mk * x;
    }
    latents[i] = x;
  }
}

// The same step with the token rows read ONCE, 16 bytes per lane: one workgroup per (frame, pair of latent rows) stages the
// guided velocity of its W/2 tokens ([W/2][4C] fp32, row pitch 4C + 1 words: bank-conflict free for the transposed read) in
// LDS, then writes the 2 x W x C latents it covers channel-major, W contiguous floats per (channel, row).  The gather form
// above reads a 4-byte piece of a 768-byte token row per lane and fetched 13x the algorithmic bytes.
__global__ __launch_bounds__(256) void cfg_euler_blend_tiled_kernel(const float* __restrict__ tok_u, const float* __restrict__ tok_c,
                                                                    int64_t ldt, int64_t tok0, float guidance, float dt,
                                                                    float* __restrict__ latents, const float* __restrict__ known,
                                                                    const float* __restrict__ mask, int C, int F, int H, int W,
                                                                    float* __restrict__ v_out) {
  extern __shared__ float sm_v[];
  const int W2 = W >> 1, NC = 4 * C, NCP = NC + 1, H2 = H >> 1;
  const int f = blockIdx.x / H2, h2 = blockIdx.x - f * H2;
  const int64_t tokbase = tok0 + (int64_t)f * H2 * W2 + (int64_t)h2 * W2;
  const int nc4 = NC >> 2;
  for (int idx = threadIdx.x; idx < W2 * nc4; idx += 256) {
    const int tw = idx / nc4, c4 = idx - tw * nc4;
    const int64_t off = (tokbase + tw) * ldt + c4 * 4;
    f32x4 v = *(const f32x4*)(tok_u + off);
    if (tok_c) {
      const f32x4 c = *(const f32x4*)(tok_c + off);
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = v[j] + guidance * (c[j] - v[j]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) sm_v[tw * NCP + c4 * 4 + j] = v[j];
  }
  __syncthreads();
  const int64_t fhw = (int64_t)F * H * W;
  for (int idx = threadIdx.x; idx < C * 2 * W; idx += 256) {
    const int w = idx % W;
    const int t = idx / W;
    const int ph = t & 1, c = t >> 1;
    const float v = sm_v[(w >> 1) * NCP + ((ph << 1) | (w & 1)) * C + c];
    const int64_t i = (((int64_t)c * F + f) * H + 2 * h2 + ph) * W + w;
    if (v_out) {
      v_out[i] = v;
      continue;
    }
    float x = latents[i] + dt * v;
    if (mask) {
      const float mk = mask[i % fhw];
      x = (1.0f - mk) * known[i] + mk * x;
    }
    latents[i] = x;
  }
}

// y = a*x + b*y over n fp32 elements (TeaCache residual bookkeeping, FX.py:1003-1051)
__global__ __launch_bounds__(256) void axpby_kernel(float* __restrict__ y, float a, const float* __restrict__ x, float b, int64_t n4) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const f32x4 xv = ((const f32x4*)x)[i];
    f32x4 yv = ((f32x4*)y)[i];
    ((f32x4*)y)[i] = xv * a + yv * b;
  }
}

// out = sum_i c[i] * x[i] over n fp32 elements; `out` may be one of the x[i] (purely elementwise).  The multistep
// samplers' updates (fm_solvers_unipc.py:349-615, fm_solvers.py:415-677) are such combinations of the sample and
// the stored x0 predictions.
constexpr int LINCOMB_MAX = 8;
struct LincombArgs {
  const float* x[LINCOMB_MAX];
  float c[LINCOMB_MAX];
};
__global__ __launch_bounds__(256) void lincomb_kernel(float* out, LincombArgs a, int n_terms, int64_t n4) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    f32x4 acc = ((const f32x4*)a.x[0])[i] * a.c[0];
    for (int t = 1; t < n_terms; ++t) acc += ((const f32x4*)a.x[t])[i] * a.c[t];
    ((f32x4*)out)[i] = acc;
  }
}

// x = (1 - mask) known + mask x, mask [fhw] broadcast over channels (PIPE.py:933-934 for non-Euler samplers)
__global__ __launch_bounds__(256) void mask_blend_kernel(float* __restrict__ x, const float* __restrict__ known,
                                                         const float* __restrict__ mask, int64_t total, int64_t fhw) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const float mk = mask[i % fhw];
    x[i] = (1.0f - mk) * known[i] + mk * x[i];
  }
}

inline int grid_for(int64_t total, int block) {
  int64_t g = (total + block - 1) / block;
  return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

// ------------------------------------------------------------------------------------------
// Content fingerprint of a buffer: out[0] += sum of its 32-bit words, out[1] += sum of word * (word index + 1), both mod 2^64.
// Host logic uses it to recognise step-invariant conditioning that the reference sampler re-materialises with torch.cat on
// every step (PIPE.py:850-886).  Integer sums: the order of the atomic adds does not matter.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long checksum_mix(unsigned long long x) {   // splitmix64 finaliser
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27; x *= 0x94D049BB133111EBull;
  x ^= x >> 31;
  return x;
}

// out[0], out[1] += two independent 64-bit hashes of every (index, word) pair, summed: position-sensitive, order-independent in
// the additions (integer atomics), so two buffers collide only if their multisets of (index, word) hashes sum alike.
__global__ __launch_bounds__(256) void checksum_kernel(const uint32_t* __restrict__ w, int64_t n_words, const uint8_t* __restrict__ tail,
                                                       int n_tail, unsigned long long* __restrict__ out) {
  unsigned long long s0 = 0, s1 = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += (int64_t)gridDim.x * blockDim.x) {
    const unsigned long long x = (unsigned long long)w[i] + (unsigned long long)(i + 1) * 0x9E3779B97F4A7C15ull;
    s0 += checksum_mix(x);
    s1 += checksum_mix(x ^ 0xD6E8FEB86659FD93ull);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    for (int j = 0; j < n_tail; ++j) {
      const unsigned long long x = (unsigned long long)tail[j] + (unsigned long long)(n_words + 1 + j) * 0x9E3779B97F4A7C15ull;
      s0 += checksum_mix(x);
      s1 += checksum_mix(x ^ 0xD6E8FEB86659FD93ull);
    }
    s0 += checksum_mix((unsigned long long)n_words * 4 + n_tail);      // the byte count is part of the key
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s0 += __shfl_xor(s0, o, 64);
    s1 += __shfl_xor(s1, o, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(out, s0);
    atomicAdd(out + 1, s1);
  }
}

}  // namespace

#define DISPATCH_VPT(C, ...)                                                          \
  do {                                                                                \
    const int vpt_ = ((C) / 8 + RT - 1) / RT;                                         \
    switch (vpt_) {                                                                   \
      case 1: { constexpr int VPT = 1; __VA_ARGS__; } break;                          \
      case 2: { constexpr int VPT = 2; __VA_ARGS__; } break;                          \
      case 3: { constexpr int VPT = 3; __VA_ARGS__; } break;                          \
      case 4: { constexpr int VPT = 4; __VA_ARGS__; } break;                          \
      case 5: case 6: case 7: case 8: { constexpr int VPT = 8; __VA_ARGS__; } break;  \
      default: return flexam_fail(FLEXAM_E_SHAPE, "row width %d unsupported (max 8192)", (int)(C)); \
    }                                                                                 \
  } while (0)

extern "C" int flexam_ln_modulate(const float* x, int64_t ldx, int64_t M, int C, float eps, const float* shift,
                                  const float* scale, int64_t tab_ld, const int32_t* row_index, int64_t rows_per_batch,
                                  const float* ln_w, const float* ln_b, void* out, int64_t ldo, void* stream) {
  FX_REQUIRE(x && out && M > 0, FLEXAM_E_ARG, "ln_modulate: null pointer or empty");
  FX_REQUIRE(C % 8 == 0 && ldx % 4 == 0 && ldo % 8 == 0, FLEXAM_E_SHAPE, "ln_modulate: C%%8, ldx%%4, ldo%%8 required (C=%d)", C);
  FX_REQUIRE((shift == nullptr) == (scale == nullptr), FLEXAM_E_ARG, "ln_modulate: shift and scale go together");
  FX_REQUIRE((ln_w == nullptr) == (ln_b == nullptr), FLEXAM_E_ARG, "ln_modulate: ln_w and ln_b go together");
  FX_REQUIRE(!shift || row_index || rows_per_batch > 0, FLEXAM_E_ARG, "ln_modulate: need row_index or rows_per_batch");
  if (rows_per_batch <= 0) rows_per_batch = 1;
  if (C % 512 == 0 && C <= 4096) {             // wave-per-row form (the DiT width 3072 = 512 * 6)
    const dim3 grid((unsigned)M), block(64);   // one wave per row, one row per workgroup
#define LN_WAVE(NV8_)                                                                                                              \
  case NV8_:                                                                                                                       \
    hipLaunchKernelGGL(ln_modulate_wave_kernel<NV8_>, grid, block, 0, (hipStream_t)stream, x, ldx, M, eps, shift, scale, tab_ld, \
                       row_index, rows_per_batch, ln_w, ln_b, (bf16*)out, ldo);                                                    \
    break;
    switch (C / 512) { LN_WAVE(1) LN_WAVE(2) LN_WAVE(3) LN_WAVE(4) LN_WAVE(5) LN_WAVE(6) LN_WAVE(7) LN_WAVE(8) }
#undef LN_WAVE
    return flexam_check_launch("flexam_ln_modulate");
  }
  DISPATCH_VPT(C, hipLaunchKernelGGL(ln_modulate_kernel<VPT>, dim3((unsigned)M), dim3(RT), 0, (hipStream_t)stream, x, ldx, C, eps,
                                     shift, scale, tab_ld, row_index, rows_per_batch, ln_w, ln_b, (bf16*)out, ldo));
  return flexam_check_launch("flexam_ln_modulate");
}

extern "C" int flexam_ln_modulate_fp8(const float* x, int64_t ldx, int64_t M, int C, float eps, const float* shift, const float* scale,
                                      int64_t tab_ld, const int32_t* row_index, int64_t rows_per_batch, const float* ln_w,
                                      const float* ln_b, void* q_out, int64_t ldq, float* row_scale, float* next_scale, float next_wnorm,
                                      float next_bias, void* stream) {
  FX_REQUIRE(x && q_out && row_scale && M > 0, FLEXAM_E_ARG, "ln_modulate_fp8: null pointer or empty");
  FX_REQUIRE(!next_scale || (next_wnorm >= 0.f && next_bias >= 0.f), FLEXAM_E_ARG, "ln_modulate_fp8: negative norm bound");
  FX_REQUIRE(C % 512 == 0 && C <= 4096, FLEXAM_E_SHAPE, "ln_modulate_fp8: row width %d must be a multiple of 512, at most 4096", C);
  FX_REQUIRE(ldx % 4 == 0 && ldq % 8 == 0, FLEXAM_E_SHAPE, "ln_modulate_fp8: ldx%%4, ldq%%8 required");
  FX_REQUIRE((shift == nullptr) == (scale == nullptr) && (ln_w == nullptr) == (ln_b == nullptr), FLEXAM_E_ARG,
             "ln_modulate_fp8: shift / scale and ln_w / ln_b go together");
  FX_REQUIRE(!shift || row_index || rows_per_batch > 0, FLEXAM_E_ARG, "ln_modulate_fp8: need row_index or rows_per_batch");
  if (rows_per_batch <= 0) rows_per_batch = 1;
  const dim3 grid((unsigned)M), block(64);
#define LN_WAVE8(NV8_)                                                                                                                   \
  case NV8_:                                                                                                                            \
    hipLaunchKernelGGL((ln_modulate_wave_kernel<NV8_, true>), grid, block, 0, (hipStream_t)stream, x, ldx, M, eps, shift, scale, tab_ld, \
                       row_index, rows_per_batch, ln_w, ln_b, (bf16*)q_out, ldq, row_scale, next_scale, next_wnorm, next_bias);         \
    break;
  switch (C / 512) { LN_WAVE8(1) LN_WAVE8(2) LN_WAVE8(3) LN_WAVE8(4) LN_WAVE8(5) LN_WAVE8(6) LN_WAVE8(7) LN_WAVE8(8) }
#undef LN_WAVE8
  return flexam_check_launch("flexam_ln_modulate_fp8");
}

extern "C" int flexam_gate_residual(float* x, int64_t ldx, const void* y, int64_t ldy, const float* gate, int64_t gate_ld,
                                    const int32_t* row_index, int64_t rows_per_batch, int64_t M, int C, void* stream) {
  FX_REQUIRE(x && y && M > 0, FLEXAM_E_ARG, "gate_residual: null pointer or empty");
  FX_REQUIRE(C % 8 == 0 && ldx % 4 == 0 && ldy % 8 == 0, FLEXAM_E_SHAPE, "gate_residual: C%%8, ldx%%4, ldy%%8 required");
  FX_REQUIRE(!gate || row_index || rows_per_batch > 0, FLEXAM_E_ARG, "gate_residual: need row_index or rows_per_batch");
  if (rows_per_batch <= 0) rows_per_batch = 1;
  hipLaunchKernelGGL(gate_residual_kernel, dim3(grid_for(M * (C / 8), 256)), dim3(256), 0, (hipStream_t)stream, x, ldx,
                     (const bf16*)y, ldy, gate, gate_ld, row_index, rows_per_batch, M, C);
  return flexam_check_launch("flexam_gate_residual");
}

extern "C" int flexam_rmsnorm_rope(const void* q_in, int64_t ldq_in, void* q_out, int64_t ldq_out, const float* wq,
                                   const void* k_in, int64_t ldk_in, void* k_out, int64_t ldk_out, const float* wk,
                                   int64_t M, int C, float eps, const float* rope_cos, const float* rope_sin,
                                   int64_t tokens_per_batch, int64_t token_offset, int head_dim, void* stream) {
  FX_REQUIRE(q_in && q_out && wq && M > 0, FLEXAM_E_ARG, "rmsnorm_rope: null pointer or empty");
  FX_REQUIRE(C % 8 == 0 && ldq_in % 8 == 0 && ldq_out % 8 == 0, FLEXAM_E_SHAPE, "rmsnorm_rope: widths must be multiples of 8");
  FX_REQUIRE((rope_cos == nullptr) == (rope_sin == nullptr), FLEXAM_E_ARG, "rmsnorm_rope: cos and sin go together");
  FX_REQUIRE(!rope_cos || (head_dim % 8 == 0 && C % head_dim == 0 && tokens_per_batch > 0), FLEXAM_E_SHAPE,
             "rmsnorm_rope: head_dim %d must divide C %d and be a multiple of 8", head_dim, C);
  if (k_in) FX_REQUIRE(k_out && wk && ldk_in % 8 == 0 && ldk_out % 8 == 0, FLEXAM_E_ARG, "rmsnorm_rope: bad k arguments");
  RmsRopeArgs a{};
  a.in[0] = (const bf16*)q_in; a.out[0] = (bf16*)q_out; a.w[0] = wq; a.ld_in[0] = ldq_in; a.ld_out[0] = ldq_out;
  a.in[1] = (const bf16*)k_in; a.out[1] = (bf16*)k_out; a.w[1] = wk; a.ld_in[1] = ldk_in; a.ld_out[1] = ldk_out;
  if (tokens_per_batch <= 0) tokens_per_batch = M;
  if (head_dim <= 0) head_dim = 8;
  DISPATCH_VPT(C, hipLaunchKernelGGL((rmsnorm_rope_kernel<VPT, false>), dim3((unsigned)M, k_in ? 2 : 1), dim3(RT), 0, (hipStream_t)stream, a,
                                     C, eps, rope_cos, rope_sin, tokens_per_batch, token_offset, head_dim));
  return flexam_check_launch("flexam_rmsnorm_rope");
}

extern "C" int flexam_rmsnorm_rope_scatter(const void* q_in, int64_t ldq_in, const float* wq, const void* k_in, int64_t ldk_in,
                                           const float* wk, const void* v_in, int64_t ldv_in, void* q_out, void* k_out, void* v_out,
                                           int64_t ld_out, int64_t out_bs, int col_block, int64_t block_stride, int64_t M, int C,
                                           float eps, const float* rope_cos, const float* rope_sin, int64_t tokens_per_batch,
                                           int64_t token_offset, int head_dim, void* stream) {
  FX_REQUIRE(M > 0 && C > 0 && tokens_per_batch > 0, FLEXAM_E_ARG, "rmsnorm_rope_scatter: empty problem");
  FX_REQUIRE((q_in == nullptr) == (q_out == nullptr) && (k_in == nullptr) == (k_out == nullptr) && (v_in == nullptr) == (v_out == nullptr),
             FLEXAM_E_ARG, "rmsnorm_rope_scatter: every input needs its output");
  FX_REQUIRE(k_in && (!q_in || wq) && wk, FLEXAM_E_ARG, "rmsnorm_rope_scatter: k (and its weight) is mandatory, q needs its weight");
  FX_REQUIRE(C % 8 == 0 && ldq_in % 8 == 0 && ldk_in % 8 == 0 && ldv_in % 8 == 0 && ld_out % 8 == 0 && out_bs % 8 == 0 &&
                 block_stride % 8 == 0 && col_block > 0 && col_block % 8 == 0 && C % col_block == 0,
             FLEXAM_E_SHAPE, "rmsnorm_rope_scatter: widths, strides and the column block must be multiples of 8 (col_block divides C)");
  FX_REQUIRE((rope_cos == nullptr) == (rope_sin == nullptr), FLEXAM_E_ARG, "rmsnorm_rope_scatter: cos and sin go together");
  FX_REQUIRE(!rope_cos || (head_dim % 8 == 0 && C % head_dim == 0), FLEXAM_E_SHAPE, "rmsnorm_rope_scatter: head_dim %d must divide C %d", head_dim, C);
  // tensor slots of the kernel: 0 = first normed tensor, 1 = second normed tensor, 2 = copied tensor; with q absent k takes slot 0
  RmsRopeArgs a{};
  int n = 0;
  if (q_in) { a.in[n] = (const bf16*)q_in; a.out[n] = (bf16*)q_out; a.w[n] = wq; a.ld_in[n] = ldq_in; a.ld_out[n] = ld_out; ++n; }
  a.in[n] = (const bf16*)k_in; a.out[n] = (bf16*)k_out; a.w[n] = wk; a.ld_in[n] = ldk_in; a.ld_out[n] = ld_out; ++n;
  a.in[2] = (const bf16*)v_in; a.out[2] = (bf16*)v_out; a.ld_in[2] = ldv_in; a.ld_out[2] = ld_out;
  a.out_bs = out_bs; a.block_stride = block_stride; a.col_block = col_block;
  if (head_dim <= 0) head_dim = 8;
  int gy = 0;                                            // grid.y: the normed tensors, then the copied one
  for (int i = 0; i < n; ++i) a.map[gy++] = i;
  if (v_in) a.map[gy++] = 2;
  DISPATCH_VPT(C, hipLaunchKernelGGL((rmsnorm_rope_kernel<VPT, true>), dim3((unsigned)M, gy), dim3(RT), 0, (hipStream_t)stream, a, C, eps,
                                     rope_cos, rope_sin, tokens_per_batch, token_offset, head_dim));
  return flexam_check_launch("flexam_rmsnorm_rope_scatter");
}

extern "C" int flexam_mod_table(const float* mod, const float* e, const float* mdens, const float* dens, float* out, int nblk,
                                int R, int nj, int nslot, int C, int rows_per_batch, int scale_mask, int dens_slots, void* stream) {
  FX_REQUIRE(mod && e && out, FLEXAM_E_ARG, "mod_table: null pointer");
  FX_REQUIRE(nblk > 0 && R > 0 && nj > 0 && nj <= 8 && C > 0 && rows_per_batch > 0, FLEXAM_E_SHAPE, "mod_table: bad sizes");
  FX_REQUIRE(dens_slots == -1 || (mdens && dens && nslot > 0), FLEXAM_E_ARG, "mod_table: density terms need mdens/dens");
  hipLaunchKernelGGL(mod_table_kernel, dim3(grid_for((int64_t)nblk * R * nj * C, 256)), dim3(256), 0, (hipStream_t)stream, mod, e,
                     mdens, dens, out, nblk, R, nj, nslot, C, rows_per_batch, scale_mask, dens_slots);
  return flexam_check_launch("flexam_mod_table");
}

extern "C" int flexam_small_linear_f32(const float* x, int64_t ldx, const void* W, int w_is_bf16, int64_t ldw, const float* b,
                                       float* y, int64_t ldy, int M, int N, int K, int silu_in, void* stream) {
  FX_REQUIRE(x && W && y, FLEXAM_E_ARG, "small_linear: null pointer");
  FX_REQUIRE(M >= 1 && M <= 32, FLEXAM_E_SHAPE, "small_linear: M=%d must be in 1..32", M);
  FX_REQUIRE(K % 4 == 0 && ldx % 4 == 0 && ldw % 4 == 0, FLEXAM_E_SHAPE, "small_linear: K, ldx, ldw must be multiples of 4");
  dim3 block(256);
  if (M <= 8) {                       // one output per wave (two distinct timesteps per sample in every demo mode)
    dim3 grid((N + 3) / 4);
    if (w_is_bf16)
      hipLaunchKernelGGL((small_linear_kernel<bf16, 8, 1>), grid, block, 0, (hipStream_t)stream, x, ldx, (const bf16*)W, ldw, b, y, ldy, M, N, K, silu_in);
    else
      hipLaunchKernelGGL((small_linear_kernel<float, 8, 1>), grid, block, 0, (hipStream_t)stream, x, ldx, (const float*)W, ldw, b, y, ldy, M, N, K, silu_in);
  } else {                            // soft foreground masks: hundreds of distinct timesteps, 32 rows per pass over the weights
    dim3 grid((N + 15) / 16);
    if (w_is_bf16)
      hipLaunchKernelGGL((small_linear_kernel<bf16, 32, 4>), grid, block, 0, (hipStream_t)stream, x, ldx, (const bf16*)W, ldw, b, y, ldy, M, N, K, silu_in);
    else
      hipLaunchKernelGGL((small_linear_kernel<float, 32, 4>), grid, block, 0, (hipStream_t)stream, x, ldx, (const float*)W, ldw, b, y, ldy, M, N, K, silu_in);
  }
  return flexam_check_launch("flexam_small_linear_f32");
}

extern "C" int flexam_sinusoid_embed(const float* t, float* out, int R, int dim, void* stream) {
  FX_REQUIRE(t && out && R > 0 && dim > 0 && dim % 2 == 0, FLEXAM_E_ARG, "sinusoid_embed: bad arguments");
  const int total = R * (dim / 2);
  hipLaunchKernelGGL(sinusoid_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, t, out, R, dim);
  return flexam_check_launch("flexam_sinusoid_embed");
}

extern "C" int flexam_patchify(const void* src, int src_is_bf16, int C, int F, int H, int W, void* dst, int64_t ldd, int col0,
                               int64_t row0, void* stream) {
  FX_REQUIRE(src && dst, FLEXAM_E_ARG, "patchify: null pointer");
  // odd H / W: the stride-2 patch convolution drops the last row / column (FX.py:885: Conv3d without padding), as the kernel's H / 2, W / 2 do
  FX_REQUIRE(C > 0 && F > 0 && H >= 2 && W >= 2, FLEXAM_E_SHAPE, "patchify: needs at least one 2 x 2 patch (H = %d, W = %d)", H, W);
  const int64_t total = (int64_t)F * (H / 2) * (W / 2) * C * 4;
  if (src_is_bf16)
    hipLaunchKernelGGL(patchify_kernel<bf16>, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16*)src, C, F, H, W, (bf16*)dst, ldd, col0, row0);
  else
    hipLaunchKernelGGL(patchify_kernel<float>, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, (const float*)src, C, F, H, W, (bf16*)dst, ldd, col0, row0);
  return flexam_check_launch("flexam_patchify");
}

extern "C" int flexam_unpatchify(const float* tok, int64_t ldt, int64_t tok0, int C, int F, int H, int W, void* dst,
                                 int dst_is_bf16, void* stream) {
  FX_REQUIRE(tok && dst, FLEXAM_E_ARG, "unpatchify: null pointer");
  FX_REQUIRE(H % 2 == 0 && W % 2 == 0, FLEXAM_E_SHAPE, "unpatchify: H, W must be even");
  const int64_t total = (int64_t)C * F * H * W;
  if (dst_is_bf16)
    hipLaunchKernelGGL(unpatchify_kernel<bf16>, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, tok, ldt, tok0, C, F, H, W, (bf16*)dst);
  else
    hipLaunchKernelGGL(unpatchify_kernel<float>, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, tok, ldt, tok0, C, F, H, W, (float*)dst);
  return flexam_check_launch("flexam_unpatchify");
}

static int launch_cfg_euler_blend(const float* tok_u, const float* tok_c, int64_t ldt, int64_t tok0, float guidance, float dt, float* latents,
                                  const float* known, const float* mask, int C, int F, int H, int W, float* v_out, hipStream_t st) {
  const size_t lds = (size_t)(W / 2) * (4 * C + 1) * sizeof(float);
  const bool vec = ldt % 4 == 0 && (uintptr_t)tok_u % 16 == 0 && (!tok_c || (uintptr_t)tok_c % 16 == 0) && lds <= 64 * 1024;
  if (vec)
    hipLaunchKernelGGL(cfg_euler_blend_tiled_kernel, dim3(F * (H / 2)), dim3(256), lds, st, tok_u, tok_c, ldt, tok0, guidance, dt, latents,
                       known, mask, C, F, H, W, v_out);
  else
    hipLaunchKernelGGL(cfg_euler_blend_kernel, dim3(grid_for((int64_t)C * F * H * W, 256)), dim3(256), 0, st, tok_u, tok_c, ldt, tok0,
                       guidance, dt, latents, known, mask, C, F, H, W, v_out);
  return 0;
}

extern "C" int flexam_cfg_euler_blend(const float* tok_uncond, const float* tok_cond, int64_t ldt, int64_t tok0, float guidance,
                                      float dt, float* latents, const float* known, const float* mask, int C, int F, int H, int W,
                                      void* stream) {
  FX_REQUIRE(tok_uncond && latents, FLEXAM_E_ARG, "cfg_euler_blend: null pointer");
  FX_REQUIRE((mask == nullptr) == (known == nullptr), FLEXAM_E_ARG, "cfg_euler_blend: mask and known go together");
  FX_REQUIRE(H % 2 == 0 && W % 2 == 0, FLEXAM_E_SHAPE, "cfg_euler_blend: H, W must be even");
  launch_cfg_euler_blend(tok_uncond, tok_cond, ldt, tok0, guidance, dt, latents, known, mask, C, F, H, W, nullptr, (hipStream_t)stream);
  return flexam_check_launch("flexam_cfg_euler_blend");
}

extern "C" int flexam_cfg_velocity(const float* tok_uncond, const float* tok_cond, int64_t ldt, int64_t tok0, float guidance, float* v,
                                   int C, int F, int H, int W, void* stream) {
  FX_REQUIRE(tok_uncond && v, FLEXAM_E_ARG, "cfg_velocity: null pointer");
  FX_REQUIRE(H % 2 == 0 && W % 2 == 0, FLEXAM_E_SHAPE, "cfg_velocity: H, W must be even");
  launch_cfg_euler_blend(tok_uncond, tok_cond, ldt, tok0, guidance, 0.f, nullptr, nullptr, nullptr, C, F, H, W, v, (hipStream_t)stream);
  return flexam_check_launch("flexam_cfg_velocity");
}

extern "C" int flexam_lincomb_f32(float* out, int64_t n, int n_terms, const float* const* terms, const float* coefs, void* stream) {
  FX_REQUIRE(out && terms && coefs && n > 0 && n % 4 == 0, FLEXAM_E_ARG, "lincomb_f32: null pointer or n %% 4 != 0");
  FX_REQUIRE(n_terms >= 1 && n_terms <= LINCOMB_MAX, FLEXAM_E_ARG, "lincomb_f32: %d terms (1..%d supported)", n_terms, LINCOMB_MAX);
  LincombArgs a{};
  for (int i = 0; i < n_terms; ++i) {
    FX_REQUIRE(terms[i], FLEXAM_E_ARG, "lincomb_f32: term %d is null", i);
    a.x[i] = terms[i];
    a.c[i] = coefs[i];
  }
  hipLaunchKernelGGL(lincomb_kernel, dim3(grid_for(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, out, a, n_terms, n / 4);
  return flexam_check_launch("flexam_lincomb_f32");
}

extern "C" int flexam_mask_blend_f32(float* x, const float* known, const float* mask, int C, int64_t fhw, void* stream) {
  FX_REQUIRE(x && known && mask && C > 0 && fhw > 0, FLEXAM_E_ARG, "mask_blend_f32: bad arguments");
  hipLaunchKernelGGL(mask_blend_kernel, dim3(grid_for((int64_t)C * fhw, 256)), dim3(256), 0, (hipStream_t)stream, x, known, mask,
                     (int64_t)C * fhw, fhw);
  return flexam_check_launch("flexam_mask_blend_f32");
}

extern "C" int flexam_axpby_f32(float* y, float a, const float* x, float b, int64_t n, void* stream) {
  FX_REQUIRE(y && x && n > 0 && n % 4 == 0, FLEXAM_E_ARG, "axpby_f32: null pointer or n %% 4 != 0");
  hipLaunchKernelGGL(axpby_kernel, dim3(grid_for(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, y, a, x, b, n / 4);
  return flexam_check_launch("flexam_axpby_f32");
}

extern "C" int flexam_checksum(const void* data, int64_t nbytes, uint64_t* out2, void* stream) {
  FX_REQUIRE(data && out2 && nbytes > 0, FLEXAM_E_ARG, "checksum: null pointer or empty buffer");
  FX_REQUIRE((uintptr_t)data % 4 == 0 && (uintptr_t)out2 % 8 == 0, FLEXAM_E_ARG, "checksum: data must be 4-byte, out 8-byte aligned");
  const int64_t words = nbytes / 4;
  hipLaunchKernelGGL(checksum_kernel, dim3(grid_for(words > 0 ? words : 1, 256)), dim3(256), 0, (hipStream_t)stream, (const uint32_t*)data,
                     words, (const uint8_t*)data + words * 4, (int)(nbytes - words * 4), (unsigned long long*)out2);
  return flexam_check_launch("flexam_checksum");
}
