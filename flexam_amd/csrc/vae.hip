// flexam_amd/csrc/vae.hip -- HBM-bound kernels of the Wan2.2 3D-VAE decoder
// (FlexAM/models/wan_vae3_8.py: RMS_norm :50-64, Resample :76-160, AttentionBlock :243-282,
// DupUp3D :375-417, unpatchify :304-318).  The convolutions themselves are flexam_gemm_bf16 over
// padded channels-last images (conv_cl.hip header); these kernels move activations between
//   "rows"   : matrices [(t, hp, wp), ld] indexed by PADDED position (GEMM outputs; border rows
//              hold garbage and are never read), fp32 or bf16, and
//   "images" : zero-bordered bf16 [frames, H+2, W+2, Cp] conv inputs (with 2 leading history
//              frames for the causal 3x3x3 convs),
// one pass each, channel-contiguous so a wave reads/writes whole 128-byte lines.
#include "common.h"
#include "flexam_hip.h"

namespace {

inline int grid_for(int64_t total, int block) {
  int64_t g = (total + block - 1) / block;
  return (int)(g < 1 ? 1 : (g > 16384 ? 16384 : g));
}

template <typename T>
__device__ __forceinline__ float ld(const T* p) { return (float)*p; }

// Row walk of the elementwise kernels (r5): blockIdx.y is one image row (t, h) of the kernel's iteration space, the threads of the
// blocks along x walk its (w, 4-channel vector) pairs with 32-bit arithmetic.  (r1-r4 split a flat 64-bit element index with three
// 64-bit divisions per thread: for 16-32 bytes moved per thread that arithmetic, not HBM, set the rate -- vae_prep 2.9 -> 3.7 TB/s at
// 160 channels once it was gone, profiles/r5j_*.)
// The grid's y extent ends at 65535: up to that many image rows they sit on gridDim.y alone, beyond (long encoder chunks at full
// height, tall inputs) as (h on y, t on z) -- image_row() is t * H + h either way, with no bound check and no division more.
inline dim3 rows_dim(int gx, int T, int H) { return (int64_t)T * H <= 65535 ? dim3(gx, T * H, 1) : dim3(gx, H, T); }
inline bool rows_fit(int T, int H) { return (int64_t)T * H <= 65535 || (H <= 65535 && T <= 65535); }
__device__ __forceinline__ int image_row() { return blockIdx.z * gridDim.y + blockIdx.y; }
inline dim3 row_grid(int T, int H, int per_row) {
  int gx = (per_row + 255) / 256;
  if (gx > 8) gx = (gx + 3) / 4;                       // ~4 iterations per thread on long rows
  return rows_dim(gx, T, H);
}
#define ROW_WALK(H_, per_row_, j_)                                     \
  const int t = image_row() / (H_), h = image_row() - t * (H_);        \
  for (int j_ = blockIdx.x * 256 + threadIdx.x; j_ < (per_row_); j_ += gridDim.x * 256)

// ------------------------------------------------------------------------------------------
// prep: rows -> image / matrix, one wave per interior position.
//   mode 0: cast;  mode 1: F.normalize(x, channel) * sqrt(C) * gamma;  mode 2: mode 1 + SiLU.
//   dst position: compact (t*H + h)*W + w  if dst_compact else padded ((t + t0)*Hp + h+1)*Wp + w+1
// ------------------------------------------------------------------------------------------
// Vector form (C % 4 == 0, C <= 1024 * ... any C): a lane owns 4 consecutive channels per 256-channel slab, loaded once with one
// 16-byte (fp32) / 8-byte (bf16) access and kept in registers between the sum of squares and the write (8-byte bf16x4 stores);
// the scalar form below remains for odd channel counts.  (r1's scalar form read every row twice with 2- / 4-byte accesses and
// ran at ~2.5 TB/s: profiles/r2b_vae_notes.txt.)
#ifndef FLEXAM_PREP_PIX1                                // diagnostic builds (tools/ab_prep_pix.py) override the positions per wave in flight
#define FLEXAM_PREP_PIX1 4
#define FLEXAM_PREP_PIX2 1
#endif
template <typename TI, int NSLAB>
__global__ __launch_bounds__(256) void vae_prep_vec_kernel(const TI* __restrict__ src, int64_t lds_, int C, int T, int H, int W,
                                                           const float* __restrict__ gamma, int mode, bf16* __restrict__ dst, int Cp,
                                                           int t0, int dst_compact) {
  // PIX positions per wave and iteration, their loads issued together: with <= 256 channels a position is one 0.5-1 KB access per
  // wave, too little in flight to cover the HBM latency (counters: 2.7 / 3.8 TB/s at 256 channels against 5.3 at 512)
  // r5 A/B at the VAE's shapes (profiles/r5i_vae_prep_positions_in_flight.txt): 8 / 4 positions are 20-40 % SLOWER than 4 / 2, 512 channels best at 1
  constexpr int PIX = NSLAB == 1 ? FLEXAM_PREP_PIX1 : NSLAB == 2 ? FLEXAM_PREP_PIX2 : 1;
  const int lane = threadIdx.x & 63;
  const int Hp = H + 2, Wp = W + 2;
  // One image row (t, h) per blockIdx.y, 4 waves x PIX positions of it per blockIdx.x: no per-position division.  (r1-r4 walked a flat
  // position index and split it with 64-bit divisions per position -- at 160 channels that arithmetic, not HBM, set the kernel's rate.)
  const int t = image_row() / H, h = image_row() - t * H;
  const TI* srow = src + (((int64_t)t * Hp + h + 1) * Wp + 1) * lds_;
  bf16* drow = dst + (dst_compact ? ((int64_t)t * H + h) * W : (((int64_t)(t + t0) * Hp + h + 1) * Wp + 1)) * Cp;
  for (int w0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * PIX; w0 < W; w0 += gridDim.x * 4 * PIX) {
    f32x4 v[PIX][NSLAB];
    const TI* s[PIX];
    bf16* d[PIX];
    const int npos = W, pos0 = w0;
#pragma unroll
    for (int p = 0; p < PIX; ++p) {
      const int w = pos0 + p < npos ? pos0 + p : npos - 1;             // a clamped tail position is loaded, never stored
      s[p] = srow + (int64_t)w * lds_;
      d[p] = drow + (int64_t)w * Cp;
    }
#pragma unroll
    for (int p = 0; p < PIX; ++p)
#pragma unroll
      for (int i = 0; i < NSLAB; ++i) {
        const int c = i * 256 + lane * 4;
        if (c < C) {
          if constexpr (sizeof(TI) == 4) {
            v[p][i] = *(const f32x4*)((const float*)s[p] + c);
          } else {
            const bf16x4 b = *(const bf16x4*)((const bf16*)s[p] + c);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[p][i][j] = bf2f(b[j]);
          }
        } else {
          v[p][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
      }
#pragma unroll
    for (int p = 0; p < PIX; ++p) {
      float scale = 1.f;
      if (mode != 0) {
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NSLAB; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) q += v[p][i][j] * v[p][i][j];
        scale = sqrtf((float)C) / fmaxf(sqrtf(wave_sum(q)), 1e-12f);
      }
      if (pos0 + p >= npos) continue;
#pragma unroll
      for (int i = 0; i < NSLAB; ++i) {
        const int c = i * 256 + lane * 4;
        if (c >= C) continue;
        f32x4 y = v[p][i];
        if (mode != 0) y = y * scale * *(const f32x4*)(gamma + c);
        bf16x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = f2bf(mode == 2 ? silu(y[j]) : y[j]);
        *(bf16x4*)(d[p] + c) = o;
      }
    }
  }
}

// Span form for <= 256 channels (C % 8 == 0): the wave-per-position form above leaves lanes idle at 160 channels (40 of 64) and moves
// 8 bytes per lane on bf16 input, and runs at 2.4-3.7 TB/s there.  Here a block takes SPAN consecutive positions of one image row -- one
// contiguous stretch of the source rows -- as a flat list of 16-byte vectors (4 fp32 / 8 bf16 channels each, every lane busy), parks
// each vector's sum of squares in LDS, lets one thread per position add its C/4 (C/8) partials, and writes the normalised bf16
// vectors; the values wait in registers in between (<= 8 vectors per thread).
constexpr int PREP_SPAN = 32;
template <typename TI>
__global__ __launch_bounds__(256) void vae_prep_span_kernel(const TI* __restrict__ src, int64_t lds_, int C, int T, int H, int W,
                                                            const float* __restrict__ gamma, int mode, bf16* __restrict__ dst, int Cp,
                                                            int t0, int dst_compact) {
  constexpr int VE = sizeof(TI) == 4 ? 4 : 8;           // channels per 16-byte vector
  constexpr int NV = sizeof(TI) == 4 ? 8 : 4;           // vectors per thread: 256 threads x NV >= PREP_SPAN positions x (256 / VE) vectors
  __shared__ float part[PREP_SPAN * 64];
  __shared__ float rscale[PREP_SPAN];
  const int Hp = H + 2, Wp = W + 2;
  const int t = image_row() / H, h = image_row() - t * H;
  const TI* srow = src + (((int64_t)t * Hp + h + 1) * Wp + 1) * lds_;
  bf16* drow = dst + (dst_compact ? ((int64_t)t * H + h) * W : (((int64_t)(t + t0) * Hp + h + 1) * Wp + 1)) * Cp;
  const int vpp = C / VE;                                // vectors per position
  for (int w0 = blockIdx.x * PREP_SPAN; w0 < W; w0 += gridDim.x * PREP_SPAN) {
    const int np = W - w0 < PREP_SPAN ? W - w0 : PREP_SPAN;
    const int nvec = np * vpp;
    f32x4 v[NV][VE / 4];
    int pos[NV], ch[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int k = threadIdx.x + i * 256;
      pos[i] = k / vpp;
      ch[i] = (k - pos[i] * vpp) * VE;
      if (k < nvec) {
        const TI* sp = srow + (int64_t)(w0 + pos[i]) * lds_ + ch[i];
        if constexpr (sizeof(TI) == 4) {
          v[i][0] = *(const f32x4*)sp;
        } else {
          const bf16x8 b = *(const bf16x8*)sp;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            v[i][0][j] = bf2f(b[j]);
            v[i][1][j] = bf2f(b[4 + j]);
          }
        }
      }
    }
    if (mode != 0) {
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int k = threadIdx.x + i * 256;
        if (k < nvec) {
          float q = 0.f;
#pragma unroll
          for (int u = 0; u < VE / 4; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j) q += v[i][u][j] * v[i][u][j];
          part[k] = q;
        }
      }
      __syncthreads();
      if ((int)threadIdx.x < np) {
        float q = 0.f;
        for (int u = 0; u < vpp; ++u) q += part[threadIdx.x * vpp + u];
        rscale[threadIdx.x] = sqrtf((float)C) / fmaxf(sqrtf(q), 1e-12f);
      }
      __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int k = threadIdx.x + i * 256;
      if (k >= nvec) continue;
      const float sc = mode != 0 ? rscale[pos[i]] : 1.f;
      bf16* dp = drow + (int64_t)(w0 + pos[i]) * Cp + ch[i];
#pragma unroll
      for (int u = 0; u < VE / 4; ++u) {
        f32x4 y = v[i][u];
        if (mode != 0) y = y * sc * *(const f32x4*)(gamma + ch[i] + 4 * u);
        bf16x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = f2bf(mode == 2 ? silu(y[j]) : y[j]);
        *(bf16x4*)(dp + 4 * u) = o;
      }
    }
    __syncthreads();                                     // part / rscale are reused by the next span
  }
}

// blocks along a row: every wave walks ~4 groups of PIX positions (a block per 16 PIX positions would live for a microsecond)
inline int prep_grid_x(int W, int pix) {
  const int groups = (W + 4 * pix - 1) / (4 * pix);
  return groups >= 8 ? (groups + 3) / 4 : groups >= 2 ? (groups + 1) / 2 : 1;
}

template <typename TI>
__global__ __launch_bounds__(256) void vae_prep_kernel(const TI* __restrict__ src, int64_t lds_, int C, int T, int H, int W,
                                                       const float* __restrict__ gamma, int mode, bf16* __restrict__ dst, int Cp,
                                                       int t0, int dst_compact) {
  const int lane = threadIdx.x & 63;
  const int Hp = H + 2, Wp = W + 2;
  const int64_t npos = (int64_t)T * H * W;
  for (int64_t pos = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); pos < npos; pos += (int64_t)gridDim.x * 4) {
    const int w = (int)(pos % W);
    int64_t r = pos / W;
    const int h = (int)(r % H);
    const int t = (int)(r / H);
    const TI* s = src + (((int64_t)t * Hp + h + 1) * Wp + w + 1) * lds_;
    bf16* d = dst + (dst_compact ? pos : (((int64_t)(t + t0) * Hp + h + 1) * Wp + w + 1)) * Cp;
    float scale = 1.f;
    if (mode != 0) {
      float q = 0.f;
      for (int c = lane; c < C; c += 64) {
        const float v = ld(s + c);
        q += v * v;
      }
      q = wave_sum(q);
      scale = sqrtf((float)C) / fmaxf(sqrtf(q), 1e-12f);
    }
    for (int c = lane; c < C; c += 64) {
      float v = ld(s + c);
      if (mode != 0) v = v * scale * gamma[c];
      if (mode == 2) v = silu(v);
      d[c] = f2bf(v);
    }
  }
}

// ------------------------------------------------------------------------------------------
// nearest-exact 2x spatial upsample of rows -> padded image; optional temporal de-interleave:
// dst frame t' = 2t + s takes channels [s*C, (s+1)*C) of src frame t (Resample upsample3d,
// wan_vae3_8.py:153-156).
// ------------------------------------------------------------------------------------------
template <typename TI>
__global__ __launch_bounds__(256) void upsample2x_kernel(const TI* __restrict__ src, int64_t lds_, int C, int T, int H, int W,
                                                         int interleave, bf16* __restrict__ dst, int Cp) {
  const int To = interleave ? 2 * T : T, Ho = 2 * H, Wo = 2 * W;
  const int Hp = H + 2, Wp = W + 2, Hop = Ho + 2, Wop = Wo + 2;
  const int cvec = C >> 2;
  const int64_t total = (int64_t)To * Ho * Wo * cvec;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % cvec) * 4;
    int64_t r = i / cvec;
    const int wo = (int)(r % Wo);
    r /= Wo;
    const int ho = (int)(r % Ho);
    const int to = (int)(r / Ho);
    const int t = interleave ? (to >> 1) : to;
    const int coff = interleave ? (to & 1) * C : 0;
    const TI* s = src + (((int64_t)t * Hp + (ho >> 1) + 1) * Wp + (wo >> 1) + 1) * lds_ + coff + c;
    bf16x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = f2bf(ld(s + j));
    *(bf16x4*)(dst + (((int64_t)to * Hop + ho + 1) * Wop + wo + 1) * Cp + c) = o;
  }
}

// ------------------------------------------------------------------------------------------
// x_main[(t', h', w'), co] += x_in[(t, h, w), (co*ft*4 + st*4 + sh*2 + sw) / repeats]     (DupUp3D)
//   t' = t*ft + st - drop  (drop = ft-1 on the first chunk: wan_vae3_8.py:415-416)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dupup_add_kernel(float* __restrict__ xm, int64_t ldm, int Co, int To, int Ho, int Wo,
                                                        const float* __restrict__ xin, int64_t ldi, int Ci, int ft, int drop) {
  // 4 consecutive output channels per thread: one 16-byte read-modify-write of x_main, four gathered reads of x_in (neighbouring
  // threads share its lines)
  const int H = Ho / 2, W = Wo / 2;
  const int Hp = H + 2, Wp = W + 2, Hop = Ho + 2, Wop = Wo + 2;
  const int repeats = Co * ft * 4 / Ci;
  const int cvec = Co >> 2;
  const int64_t total = (int64_t)To * Ho * Wo * cvec;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int co = (int)(i % cvec) * 4;
    int64_t r = i / cvec;
    const int wo = (int)(r % Wo);
    r /= Wo;
    const int ho = (int)(r % Ho);
    const int to = (int)(r / Ho);
    const int tt = to + drop;
    const int t = tt / ft, st = tt - t * ft;
    const float* xi = xin + (((int64_t)t * Hp + (ho >> 1) + 1) * Wp + (wo >> 1) + 1) * ldi;
    const int sub = st * 4 + (ho & 1) * 2 + (wo & 1);
    float* xo = xm + (((int64_t)to * Hop + ho + 1) * Wop + wo + 1) * ldm + co;
    f32x4 x = *(const f32x4*)xo;
#pragma unroll
    for (int j = 0; j < 4; ++j) x[j] += xi[((co + j) * ft * 4 + sub) / repeats];
    *(f32x4*)xo = x;
  }
}

// ------------------------------------------------------------------------------------------
// rows [T, 2C] -> padded image of 2T frames at the SAME resolution: frame 2t + s takes channels [s*C, (s+1)*C) of src frame t
// (the temporal half of Resample upsample3d, wan_vae3_8.py:153-156, without the spatial upsample: the phase-decomposed
// convolution below reads the low-resolution frames directly)
// ------------------------------------------------------------------------------------------
template <typename TI>
__global__ __launch_bounds__(256) void deinterleave_kernel(const TI* __restrict__ src, int64_t lds_, int C, int T, int H, int W,
                                                           bf16* __restrict__ dst, int Cp) {
  const int Hp = H + 2, Wp = W + 2;
  const int cvec = C >> 2;
  ROW_WALK(H, W * cvec, j) {                            // iteration space: the 2T output frames
    const int w = j / cvec, c = (j - w * cvec) * 4;
    const int to = t;
    const int64_t pos = (((int64_t)(to >> 1) * Hp + h + 1) * Wp + w + 1);
    const TI* s = src + pos * lds_ + (to & 1) * C + c;
    bf16x4 o;
#pragma unroll
    for (int q = 0; q < 4; ++q) o[q] = f2bf(ld(s + q));
    *(bf16x4*)(dst + (((int64_t)to * Hp + h + 1) * Wp + w + 1) * Cp + c) = o;
  }
}

// ------------------------------------------------------------------------------------------
// x_main[(t', h', w'), co] = phase[(h' & 1) * 2 + (w' & 1)][(t', h' >> 1, w' >> 1), co] + DupUp3D(x_in)   (see dupup_add_kernel)
// The four phase matrices are the outputs of the phase-decomposed "nearest 2x upsample + 3x3 convolution": output pixels of
// parity (a, b) are a 2x2 convolution of the LOW-resolution image with pre-summed taps, so the upsampled image is never built
// and the convolution costs 16 instead of 36 tap products per low-resolution pixel.  x_main is written, not accumulated.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void phase_dupup_kernel(const float* __restrict__ ph, int64_t ldp, int64_t phase_stride,
                                                          float* __restrict__ xm, int64_t ldm, int Co, int To, int Ho, int Wo,
                                                          const float* __restrict__ xin, int64_t ldi, int Ci, int ft, int drop) {
  const int H = Ho / 2, W = Wo / 2;
  const int Hp = H + 2, Wp = W + 2, Hop = Ho + 2, Wop = Wo + 2;
  const int repeats = Co * ft * 4 / Ci;
  const int cvec = Co >> 2;
  ROW_WALK(Ho, Wo * cvec, j) {                          // iteration space: the To x Ho output rows
    const int wo = j / cvec, co = (j - wo * cvec) * 4;
    const int to = t, ho = h;
    const int tt = to + drop;
    const int ti = tt / ft, st = tt - ti * ft;
    const float* xi = xin + (((int64_t)ti * Hp + (ho >> 1) + 1) * Wp + (wo >> 1) + 1) * ldi;
    const int sub = st * 4 + (ho & 1) * 2 + (wo & 1);
    const float* pp = ph + (int64_t)((ho & 1) * 2 + (wo & 1)) * phase_stride + (((int64_t)to * Hp + (ho >> 1) + 1) * Wp + (wo >> 1) + 1) * ldp + co;
    f32x4 x = __builtin_nontemporal_load((const f32x4*)pp);
#pragma unroll
    for (int j = 0; j < 4; ++j) x[j] += xi[((co + j) * ft * 4 + sub) / repeats];
    *(f32x4*)(xm + (((int64_t)to * Hop + ho + 1) * Wop + wo + 1) * ldm + co) = x;
  }
}

// ------------------------------------------------------------------------------------------
// out[(t, h, w), o] = bias[o] + sum over the kt x 3 x 3 taps of  Y[(t + dt, h + dh - 1, w + dw - 1), tap * Co + o]
// A causal convolution with FEW output channels (the decoder head: 256 -> 12) as "one product per input pixel and tap, then a
// gather": Y = image x [kt*9*Co, Cin]^T is ONE plain GEMM over the input pixels (K = Cin, every activation read once) instead of
// an implicit GEMM whose K = 27 Cin re-reads every activation 27 times for 12 useful output columns of a 160-wide tile.
// Y rows: padded positions of the (kt - 1) history frames + T current frames; border rows of Y are exact zeros (zero image rows,
// no bias in the GEMM).  Four consecutive output channels per thread (Co % 4 == 0).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tapsum_kernel(const float* __restrict__ y, int64_t ldy, int T, int H, int W, int kt, int Co,
                                                     const float* __restrict__ bias, float* __restrict__ out, int64_t ldo) {
  const int Hp = H + 2, Wp = W + 2;
  const int cvec = Co >> 2;
  ROW_WALK(H, W * cvec, j) {
    const int w = j / cvec, o = (j - w * cvec) * 4;
    f32x4 acc = bias ? *(const f32x4*)(bias + o) : (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int dt = 0; dt < kt; ++dt)
#pragma unroll
      for (int dh = 0; dh < 3; ++dh) {
        const float* row = y + (((int64_t)(t + dt) * Hp + h + dh) * Wp + w) * ldy + ((dt * 3 + dh) * 3) * Co + o;
#pragma unroll
        for (int dw = 0; dw < 3; ++dw) acc += __builtin_nontemporal_load((const f32x4*)(row + dw * ldy + dw * Co));
      }
    *(f32x4*)(out + (((int64_t)t * Hp + h + 1) * Wp + w + 1) * ldo + o) = acc;
  }
}

// row softmax(scale * s) fp32 -> bf16 (zero padded to ldo columns); one 256-thread block per row
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ s, int64_t lds_, int N, float scale,
                                                           bf16* __restrict__ out, int64_t ldo, int Npad) {
  __shared__ float red[8];
  const float* row = s + (int64_t)blockIdx.x * lds_;
  float mx = -INFINITY;
  for (int i = threadIdx.x; i < N; i += 256) mx = fmaxf(mx, row[i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float sum = 0.f;
  for (int i = threadIdx.x; i < N; i += 256) sum += __expf((row[i] - mx) * scale);
  sum = block_sum<256>(sum, red);
  const float inv = 1.f / sum;
  bf16* orow = out + (int64_t)blockIdx.x * ldo;
  for (int i = threadIdx.x; i < Npad; i += 256) orow[i] = f2bf(i < N ? __expf((row[i] - mx) * scale) * inv : 0.f);
}

// x[padded (t,h,w), c] += y[compact (t,h,w), c]
__global__ __launch_bounds__(256) void scatter_add_kernel(float* __restrict__ x, int64_t ldx, const bf16* __restrict__ y, int64_t ldy,
                                                          int C, int T, int H, int W) {
  const int Hp = H + 2, Wp = W + 2;
  const int64_t total = (int64_t)T * H * W * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    int64_t pos = i / C;
    const int w = (int)(pos % W);
    int64_t r = pos / W;
    const int h = (int)(r % H);
    const int t = (int)(r / H);
    x[(((int64_t)t * Hp + h + 1) * Wp + w + 1) * ldx + c] += bf2f(y[pos * ldy + c]);
  }
}

// rows [(t,hp,wp), ld] (12 channels = (c r q), wan_vae3_8.py:304-318) -> video[c][f0 + t][2h + q][2w + r], clamped
__global__ __launch_bounds__(256) void vae_unpatchify_kernel(const float* __restrict__ src, int64_t lds_, int T, int H, int W,
                                                             float* __restrict__ video, int Ftot, int f0, float lo, float hi) {
  const int Hp = H + 2, Wp = W + 2;
  const int64_t total = (int64_t)3 * T * (2 * H) * (2 * W);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % (2 * W));
    int64_t r = i / (2 * W);
    const int y = (int)(r % (2 * H));
    r /= (2 * H);
    const int t = (int)(r % T);
    const int c = (int)(r / T);
    const int ch = c * 4 + (x & 1) * 2 + (y & 1);
    const float v = src[(((int64_t)t * Hp + (y >> 1) + 1) * Wp + (x >> 1) + 1) * lds_ + ch];
    video[(((int64_t)c * Ftot + f0 + t) * (2 * H) + y) * (2 * W) + x] = fminf(fmaxf(v, lo), hi);
  }
}

// z[c][t][h][w] * std[c] + mean[c] -> padded image interior (wan_vae3_8.py:823-828)
__global__ __launch_bounds__(256) void pack_affine_kernel(const float* __restrict__ src, int C, int T, int H, int W,
                                                          const float* __restrict__ mul, const float* __restrict__ add,
                                                          bf16* __restrict__ dst, int Cp) {
  const int Hp = H + 2, Wp = W + 2;
  const int64_t total = (int64_t)T * H * W * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    int64_t r = i / C;
    const int w = (int)(r % W);
    r /= W;
    const int h = (int)(r % H);
    const int t = (int)(r / H);
    const float v = src[(((int64_t)c * T + t) * H + h) * W + w] * mul[c] + add[c];
    dst[(((int64_t)t * Hp + h + 1) * Wp + w + 1) * Cp + c] = f2bf(v);
  }
}

// ------------------------------------------------------------------------------------------
// Encoder side (wan_vae3_8.py:285-301 patchify, :104-113 downsample convs, :321-372 AvgDown3D).
// video[c][f0 + t][2h + q][2w + r] -> image interior (t + t0, h, w), channel c*4 + r*2 + q
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vae_patchify_kernel(const float* __restrict__ video, int Ftot, int f0, int T, int H, int W,
                                                           bf16* __restrict__ dst, int Cp, int t0) {
  const int Hp = H + 2, Wp = W + 2;
  const int64_t total = (int64_t)3 * T * (2 * H) * (2 * W);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % (2 * W));
    int64_t r = i / (2 * W);
    const int y = (int)(r % (2 * H));
    r /= (2 * H);
    const int t = (int)(r % T);
    const int c = (int)(r / T);
    const float v = video[(((int64_t)c * Ftot + f0 + t) * (2 * H) + y) * (2 * W) + x];
    dst[(((int64_t)(t + t0) * Hp + (y >> 1) + 1) * Wp + (x >> 1) + 1) * Cp + c * 4 + (x & 1) * 2 + (y & 1)] = f2bf(v);
  }
}

// rows [(t, hp, wp), ld] at H x W -> space-to-depth image at H/2 x W/2 with 4 sub-pixel channel groups:
//   dst[(t + t0, h2 + 1, w2 + 1), (a*2 + b)*Cs + c] = src[(t, 2*h2 + a + 1, 2*w2 + b + 1), c]
// so ZeroPad2d((0,1,0,1)) + Conv2d(3x3, stride 2) becomes a unit-stride implicit GEMM whose 9 taps
// address (row offset th, col offset tw, channel group) of this image; its zero border at h2 = H/2,
// w2 = W/2 is exactly the reference's right/bottom zero pad.
template <typename TI>
__global__ __launch_bounds__(256) void space_to_depth_kernel(const TI* __restrict__ src, int64_t lds_, int C, int T, int H, int W,
                                                             bf16* __restrict__ dst, int Cs, int t0) {
  const int Hp = H + 2, Wp = W + 2, H2p = H / 2 + 2, W2p = W / 2 + 2;
  const int cvec = C >> 2;
  ROW_WALK(H, W * cvec, j) {
    const int w = j / cvec, c = (j - w * cvec) * 4;
    const TI* s = src + (((int64_t)t * Hp + h + 1) * Wp + w + 1) * lds_ + c;
    bf16x4 o;
    o[0] = f2bf(ld(s));
    o[1] = f2bf(ld(s + 1));
    o[2] = f2bf(ld(s + 2));
    o[3] = f2bf(ld(s + 3));
    *(bf16x4*)(dst + ((((int64_t)(t + t0) * H2p + (h >> 1) + 1) * W2p + (w >> 1) + 1) * 4 + (h & 1) * 2 + (w & 1)) * Cs + c) = o;
  }
}

// x_main[(to, ho, wo), co] += mean_{k < g} x_in[(to*ft + st - pad_t, ho*fs + sh, wo*fs + sw), c]      (AvgDown3D)
//   flat = co*g + k,  c = flat / (ft*fs*fs),  (st, sh, sw) = digits of flat % (ft*fs*fs);  frames < 0 are the zero front pad
__global__ __launch_bounds__(256) void avgdown_add_kernel(float* __restrict__ xm, int64_t ldm, int Co, int To, int Ho, int Wo,
                                                          const float* __restrict__ xin, int64_t ldi, int Ci, int ft, int fs, int pad_t) {
  const int Hp = Ho * fs + 2, Wp = Wo * fs + 2, Hop = Ho + 2, Wop = Wo + 2;
  const int factor = ft * fs * fs;
  const int g = Ci * factor / Co;
  const float inv = 1.f / (float)g;
  const int64_t total = (int64_t)To * Ho * Wo * Co;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int co = (int)(i % Co);
    int64_t r = i / Co;
    const int wo = (int)(r % Wo);
    r /= Wo;
    const int ho = (int)(r % Ho);
    const int to = (int)(r / Ho);
    float acc = 0.f;
    for (int k = 0; k < g; ++k) {
      const int flat = co * g + k;
      const int c = flat / factor, rem = flat - c * factor;
      const int st = rem / (fs * fs), sh = (rem / fs) % fs, sw = rem % fs;
      const int t = to * ft + st - pad_t;
      if (t >= 0) acc += xin[(((int64_t)t * Hp + ho * fs + sh + 1) * Wp + wo * fs + sw + 1) * ldi + c];
    }
    xm[(((int64_t)to * Hop + ho + 1) * Wop + wo + 1) * ldm + co] += acc * inv;
  }
}

// The same with Ci == Co (g = ft * fs * fs: output channel c is the mean of input channel c over the ft x fs x fs block -- plain average
// pooling, the encoder's first stage): four consecutive channels per thread, 16-byte loads of x_in and one 16-byte read-modify-write.
// (The scalar form above moved 445 MB in 231 us at 256 x 448 x 160: 1.9 TB/s.)
__global__ __launch_bounds__(256) void avgdown_add_same_kernel(float* __restrict__ xm, int64_t ldm, int C, int To, int Ho, int Wo,
                                                               const float* __restrict__ xin, int64_t ldi, int ft, int fs, int pad_t) {
  const int Hp = Ho * fs + 2, Wp = Wo * fs + 2, Hop = Ho + 2, Wop = Wo + 2;
  const float inv = 1.f / (float)(ft * fs * fs);
  const int cvec = C >> 2;
  ROW_WALK(Ho, Wo * cvec, j) {                          // iteration space: the To x Ho output rows
    const int wo = j / cvec, c = (j - wo * cvec) * 4;
    const int to = t, ho = h;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int st = 0; st < ft; ++st) {
      const int ti = to * ft + st - pad_t;
      if (ti < 0) continue;
      for (int sh = 0; sh < fs; ++sh)
        for (int sw = 0; sw < fs; ++sw)
          acc += __builtin_nontemporal_load((const f32x4*)(xin + (((int64_t)ti * Hp + ho * fs + sh + 1) * Wp + wo * fs + sw + 1) * ldi + c));
    }
    float* xo = xm + (((int64_t)to * Hop + ho + 1) * Wop + wo + 1) * ldm + c;
    *(f32x4*)xo = *(const f32x4*)xo + acc * inv;
  }
}

}  // namespace

extern "C" int flexam_vae_prep_cl(const void* src, int src_is_bf16, int64_t ld_src, int C, int T, int H, int W, const float* gamma,
                                  int mode, void* dst, int Cp, int t0, int dst_compact, void* stream) {
  FX_REQUIRE(src && dst, FLEXAM_E_ARG, "vae_prep_cl: null pointer");
  FX_REQUIRE(mode >= 0 && mode <= 2 && (mode == 0 || gamma), FLEXAM_E_ARG, "vae_prep_cl: mode %d needs gamma", mode);
  FX_REQUIRE(C > 0 && C <= Cp && C <= ld_src && T > 0 && H > 0 && W > 0, FLEXAM_E_SHAPE, "vae_prep_cl: bad shape");
  const int64_t npos = (int64_t)T * H * W;
  dim3 grid(grid_for(npos, 4)), block(256);
  hipStream_t st = (hipStream_t)stream;
  FX_REQUIRE(rows_fit(T, H), FLEXAM_E_SHAPE, "vae_prep_cl: T=%d x H=%d image rows do not fit a grid (each <= 65535)", T, H);
  const bool vec = C % 4 == 0 && C <= 1024 && ld_src % 4 == 0 && Cp % 4 == 0 && (uintptr_t)src % 16 == 0 && (uintptr_t)dst % 8 == 0 &&
                   (!gamma || (uintptr_t)gamma % 16 == 0);
  // vector form: one image row per blockIdx.y, 4 waves x PIX positions per blockIdx.x (PIX as in the kernel)
#define PREP_VEC(TI_, NS_)                                                                                                         \
  hipLaunchKernelGGL((vae_prep_vec_kernel<TI_, NS_>), rows_dim(prep_grid_x(W, (NS_) == 1 ? FLEXAM_PREP_PIX1 : (NS_) == 2 ? FLEXAM_PREP_PIX2 : 1), T, H), \
                     block, 0, st, (const TI_*)src, ld_src, C, T, H, W, gamma, mode, (bf16*)dst, Cp, t0, dst_compact)
  // bf16 rows of <= 256 channels: the span form (16-byte vectors on a contiguous stretch of the row, every lane busy).  Measured r5s at
  // the VAE's shapes: 160 channels bf16 91.5 against 123.7 us, 256 channels 130 against 133; on fp32 rows it LOSES (135 against 117,
  // 218 against 141: twice the registers per thread for the same bytes), so those keep the wave-per-position form.
  const bool span = vec && src_is_bf16 && C <= 256 && C % 8 == 0 && ld_src % 8 == 0 && Cp % 8 == 0 && (uintptr_t)dst % 16 == 0 &&
                    ((uintptr_t)gamma % 16 == 0) && !getenv("FLEXAM_VAE_PREP_WAVE");
  if (span) {
    hipLaunchKernelGGL(vae_prep_span_kernel<bf16>, rows_dim((W + PREP_SPAN - 1) / PREP_SPAN, T, H), block, 0, st, (const bf16*)src, ld_src, C, T, H, W,
                       gamma, mode, (bf16*)dst, Cp, t0, dst_compact);
  } else if (vec) {
    const int ns = (C + 255) / 256;
    if (src_is_bf16) {
      if (ns == 1) PREP_VEC(bf16, 1); else if (ns == 2) PREP_VEC(bf16, 2); else if (ns == 3) PREP_VEC(bf16, 3); else PREP_VEC(bf16, 4);
    } else {
      if (ns == 1) PREP_VEC(float, 1); else if (ns == 2) PREP_VEC(float, 2); else if (ns == 3) PREP_VEC(float, 3); else PREP_VEC(float, 4);
    }
  } else if (src_is_bf16) {
    hipLaunchKernelGGL(vae_prep_kernel<bf16>, grid, block, 0, st, (const bf16*)src, ld_src, C, T, H, W, gamma, mode, (bf16*)dst, Cp, t0, dst_compact);
  } else {
    hipLaunchKernelGGL(vae_prep_kernel<float>, grid, block, 0, st, (const float*)src, ld_src, C, T, H, W, gamma, mode, (bf16*)dst, Cp, t0, dst_compact);
  }
#undef PREP_VEC
  return flexam_check_launch("flexam_vae_prep_cl");
}

extern "C" int flexam_upsample2x_cl(const void* src, int src_is_bf16, int64_t ld_src, int C, int T, int H, int W, int interleave,
                                    void* dst, int Cp, void* stream) {
  FX_REQUIRE(src && dst, FLEXAM_E_ARG, "upsample2x_cl: null pointer");
  FX_REQUIRE(C % 4 == 0 && C <= Cp && Cp % 4 == 0, FLEXAM_E_SHAPE, "upsample2x_cl: C=%d must be a multiple of 4", C);
  const int64_t total = (int64_t)(interleave ? 2 * T : T) * 4 * H * W * (C / 4);
  if (src_is_bf16)
    hipLaunchKernelGGL(upsample2x_kernel<bf16>, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16*)src, ld_src, C, T, H, W, interleave, (bf16*)dst, Cp);
  else
    hipLaunchKernelGGL(upsample2x_kernel<float>, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, (const float*)src, ld_src, C, T, H, W, interleave, (bf16*)dst, Cp);
  return flexam_check_launch("flexam_upsample2x_cl");
}

extern "C" int flexam_dupup_add_cl(float* x_main, int64_t ld_main, int Co, int To, int Ho, int Wo, const float* x_in, int64_t ld_in,
                                   int Ci, int ft, int drop, void* stream) {
  FX_REQUIRE(x_main && x_in, FLEXAM_E_ARG, "dupup_add_cl: null pointer");
  FX_REQUIRE(Ho % 2 == 0 && Wo % 2 == 0 && (ft == 1 || ft == 2) && (Co * ft * 4) % Ci == 0, FLEXAM_E_SHAPE, "dupup_add_cl: bad shape");
  FX_REQUIRE(Co % 4 == 0 && ld_main % 4 == 0 && (uintptr_t)x_main % 16 == 0, FLEXAM_E_SHAPE, "dupup_add_cl: Co and ld_main must be multiples of 4");
  hipLaunchKernelGGL(dupup_add_kernel, dim3(grid_for((int64_t)To * Ho * Wo * (Co / 4), 256)), dim3(256), 0, (hipStream_t)stream, x_main, ld_main,
                     Co, To, Ho, Wo, x_in, ld_in, Ci, ft, drop);
  return flexam_check_launch("flexam_dupup_add_cl");
}

extern "C" int flexam_deinterleave_cl(const void* src, int src_is_bf16, int64_t ld_src, int C, int T, int H, int W, void* dst, int Cp,
                                      void* stream) {
  FX_REQUIRE(src && dst, FLEXAM_E_ARG, "deinterleave_cl: null pointer");
  FX_REQUIRE(C % 4 == 0 && C <= Cp && Cp % 4 == 0 && 2 * C <= ld_src && T > 0 && H > 0 && W > 0, FLEXAM_E_SHAPE,
             "deinterleave_cl: C=%d must be a multiple of 4 and 2C <= ld_src=%ld", C, (long)ld_src);
  FX_REQUIRE(rows_fit(2 * T, H), FLEXAM_E_SHAPE, "deinterleave_cl: 2 T=%d x H=%d image rows do not fit a grid (each <= 65535)", 2 * T, H);
  if (src_is_bf16)
    hipLaunchKernelGGL(deinterleave_kernel<bf16>, row_grid(2 * T, H, W * (C / 4)), dim3(256), 0, (hipStream_t)stream, (const bf16*)src, ld_src, C, T, H, W, (bf16*)dst, Cp);
  else
    hipLaunchKernelGGL(deinterleave_kernel<float>, row_grid(2 * T, H, W * (C / 4)), dim3(256), 0, (hipStream_t)stream, (const float*)src, ld_src, C, T, H, W, (bf16*)dst, Cp);
  return flexam_check_launch("flexam_deinterleave_cl");
}

extern "C" int flexam_phase_dupup_cl(const float* phases, int64_t ld_ph, int64_t phase_stride, float* x_main, int64_t ld_main, int Co, int To,
                                     int Ho, int Wo, const float* x_in, int64_t ld_in, int Ci, int ft, int drop, void* stream) {
  FX_REQUIRE(phases && x_main && x_in, FLEXAM_E_ARG, "phase_dupup_cl: null pointer");
  FX_REQUIRE(Ho % 2 == 0 && Wo % 2 == 0 && (ft == 1 || ft == 2) && (Co * ft * 4) % Ci == 0, FLEXAM_E_SHAPE, "phase_dupup_cl: bad shape");
  FX_REQUIRE(Co % 4 == 0 && ld_main % 4 == 0 && ld_ph % 4 == 0 && phase_stride % 4 == 0 && (uintptr_t)x_main % 16 == 0 && (uintptr_t)phases % 16 == 0,
             FLEXAM_E_SHAPE, "phase_dupup_cl: Co, ld_main, ld_ph and phase_stride must be multiples of 4");
  FX_REQUIRE(rows_fit(To, Ho), FLEXAM_E_SHAPE, "phase_dupup_cl: To=%d x Ho=%d image rows do not fit a grid (each <= 65535)", To, Ho);
  hipLaunchKernelGGL(phase_dupup_kernel, row_grid(To, Ho, Wo * (Co / 4)), dim3(256), 0, (hipStream_t)stream, phases, ld_ph,
                     phase_stride, x_main, ld_main, Co, To, Ho, Wo, x_in, ld_in, Ci, ft, drop);
  return flexam_check_launch("flexam_phase_dupup_cl");
}

extern "C" int flexam_tapsum_cl(const float* y, int64_t ld_y, int T, int H, int W, int kt, int Co, const float* bias, float* out,
                                int64_t ld_out, void* stream) {
  FX_REQUIRE(y && out, FLEXAM_E_ARG, "tapsum_cl: null pointer");
  FX_REQUIRE(T > 0 && H > 0 && W > 0 && kt >= 1 && kt <= 3 && Co > 0 && Co % 4 == 0 && ld_y >= (int64_t)kt * 9 * Co && ld_y % 4 == 0 && ld_out % 4 == 0 &&
             (uintptr_t)y % 16 == 0 && (uintptr_t)out % 16 == 0 && (!bias || (uintptr_t)bias % 16 == 0), FLEXAM_E_SHAPE,
             "tapsum_cl: Co=%d must be a multiple of 4, ld_y=%ld >= kt*9*Co, 16-byte aligned rows", Co, (long)ld_y);
  FX_REQUIRE(rows_fit(T, H), FLEXAM_E_SHAPE, "tapsum_cl: T=%d x H=%d image rows do not fit a grid (each <= 65535)", T, H);
  hipLaunchKernelGGL(tapsum_kernel, row_grid(T, H, W * (Co / 4)), dim3(256), 0, (hipStream_t)stream, y, ld_y, T, H, W, kt, Co,
                     bias, out, ld_out);
  return flexam_check_launch("flexam_tapsum_cl");
}

extern "C" int flexam_softmax_rows(const float* s, int64_t ld_s, int64_t M, int N, float scale, void* out, int64_t ld_out, int Npad,
                                   void* stream) {
  FX_REQUIRE(s && out && M > 0 && N > 0 && Npad >= N && Npad <= ld_out, FLEXAM_E_ARG, "softmax_rows: bad arguments");
  hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)M), dim3(256), 0, (hipStream_t)stream, s, ld_s, N, scale, (bf16*)out, ld_out, Npad);
  return flexam_check_launch("flexam_softmax_rows");
}

extern "C" int flexam_scatter_add_cl(float* x, int64_t ldx, const void* y, int64_t ldy, int C, int T, int H, int W, void* stream) {
  FX_REQUIRE(x && y, FLEXAM_E_ARG, "scatter_add_cl: null pointer");
  hipLaunchKernelGGL(scatter_add_kernel, dim3(grid_for((int64_t)T * H * W * C, 256)), dim3(256), 0, (hipStream_t)stream, x, ldx,
                     (const bf16*)y, ldy, C, T, H, W);
  return flexam_check_launch("flexam_scatter_add_cl");
}

extern "C" int flexam_vae_unpatchify_clamp(const float* src, int64_t ld_src, int T, int H, int W, float* video, int Ftot, int f0,
                                           float lo, float hi, void* stream) {
  FX_REQUIRE(src && video && ld_src >= 12, FLEXAM_E_ARG, "vae_unpatchify_clamp: bad arguments");
  FX_REQUIRE(f0 >= 0 && f0 + T <= Ftot, FLEXAM_E_SHAPE, "vae_unpatchify_clamp: frames %d+%d exceed %d", f0, T, Ftot);
  hipLaunchKernelGGL(vae_unpatchify_kernel, dim3(grid_for((int64_t)12 * T * H * W, 256)), dim3(256), 0, (hipStream_t)stream, src, ld_src,
                     T, H, W, video, Ftot, f0, lo, hi);
  return flexam_check_launch("flexam_vae_unpatchify_clamp");
}

extern "C" int flexam_pack_affine_cl(const float* src, int C, int T, int H, int W, const float* mul, const float* add, void* dst, int Cp,
                                     void* stream) {
  FX_REQUIRE(src && mul && add && dst && C <= Cp, FLEXAM_E_ARG, "pack_affine_cl: bad arguments");
  hipLaunchKernelGGL(pack_affine_kernel, dim3(grid_for((int64_t)T * H * W * C, 256)), dim3(256), 0, (hipStream_t)stream, src, C, T, H, W,
                     mul, add, (bf16*)dst, Cp);
  return flexam_check_launch("flexam_pack_affine_cl");
}

extern "C" int flexam_vae_patchify_cl(const float* video, int Ftot, int f0, int T, int H, int W, void* dst, int Cp, int t0, void* stream) {
  FX_REQUIRE(video && dst && Cp >= 12, FLEXAM_E_ARG, "vae_patchify_cl: bad arguments");
  FX_REQUIRE(f0 >= 0 && T > 0 && f0 + T <= Ftot && H > 0 && W > 0, FLEXAM_E_SHAPE, "vae_patchify_cl: frames %d+%d exceed %d", f0, T, Ftot);
  hipLaunchKernelGGL(vae_patchify_kernel, dim3(grid_for((int64_t)12 * T * H * W, 256)), dim3(256), 0, (hipStream_t)stream, video, Ftot, f0,
                     T, H, W, (bf16*)dst, Cp, t0);
  return flexam_check_launch("flexam_vae_patchify_cl");
}

extern "C" int flexam_space_to_depth_cl(const void* src, int src_is_bf16, int64_t ld_src, int C, int T, int H, int W, void* dst, int Cs,
                                        int t0, void* stream) {
  FX_REQUIRE(src && dst, FLEXAM_E_ARG, "space_to_depth_cl: null pointer");
  FX_REQUIRE(C % 4 == 0 && C <= Cs && Cs % 4 == 0 && C <= ld_src && H % 2 == 0 && W % 2 == 0 && T > 0, FLEXAM_E_SHAPE,
             "space_to_depth_cl: bad shape C=%d Cs=%d H=%d W=%d", C, Cs, H, W);
  FX_REQUIRE(rows_fit(T, H), FLEXAM_E_SHAPE, "space_to_depth_cl: T=%d x H=%d image rows do not fit a grid (each <= 65535)", T, H);
  if (src_is_bf16)
    hipLaunchKernelGGL(space_to_depth_kernel<bf16>, row_grid(T, H, W * (C / 4)), dim3(256), 0, (hipStream_t)stream, (const bf16*)src, ld_src, C, T, H, W, (bf16*)dst, Cs, t0);
  else
    hipLaunchKernelGGL(space_to_depth_kernel<float>, row_grid(T, H, W * (C / 4)), dim3(256), 0, (hipStream_t)stream, (const float*)src, ld_src, C, T, H, W, (bf16*)dst, Cs, t0);
  return flexam_check_launch("flexam_space_to_depth_cl");
}

extern "C" int flexam_avgdown_add_cl(float* x_main, int64_t ld_main, int Co, int To, int Ho, int Wo, const float* x_in, int64_t ld_in,
                                     int Ci, int Ti, int ft, int fs, void* stream) {
  FX_REQUIRE(x_main && x_in, FLEXAM_E_ARG, "avgdown_add_cl: null pointer");
  FX_REQUIRE((ft == 1 || ft == 2) && (fs == 1 || fs == 2) && (Ci * ft * fs * fs) % Co == 0, FLEXAM_E_SHAPE, "avgdown_add_cl: bad factors");
  const int pad_t = (ft - Ti % ft) % ft;
  FX_REQUIRE((Ti + pad_t) / ft == To, FLEXAM_E_SHAPE, "avgdown_add_cl: %d input frames do not give %d output frames", Ti, To);
  if (Ci == Co && Co % 4 == 0 && ld_main % 4 == 0 && ld_in % 4 == 0 && (uintptr_t)x_main % 16 == 0 && (uintptr_t)x_in % 16 == 0 && rows_fit(To, Ho))
    hipLaunchKernelGGL(avgdown_add_same_kernel, row_grid(To, Ho, Wo * (Co / 4)), dim3(256), 0, (hipStream_t)stream, x_main,
                       ld_main, Co, To, Ho, Wo, x_in, ld_in, ft, fs, pad_t);
  else
    hipLaunchKernelGGL(avgdown_add_kernel, dim3(grid_for((int64_t)To * Ho * Wo * Co, 256)), dim3(256), 0, (hipStream_t)stream, x_main, ld_main,
                       Co, To, Ho, Wo, x_in, ld_in, Ci, ft, fs, pad_t);
  return flexam_check_launch("flexam_avgdown_add_cl");
}
