// flexam_amd/csrc/attn.hip -- flash attention forward, head_dim 128, bf16 in/out, fp32 softmax
// and accumulation; non-causal with a key-length bound.  Used for the DiT's full 3-D
// spatiotemporal self-attention (Lq = Lk = 11648 at 97x512x896) and the text cross-attention
// (Lk = 512).  Replaces flash_attn / sageattn / SDPA behind attention():
// FlexAM/models/attention_utils.py:43-233, called from wan_transformer3d_FlexAM.py:251-256,367.
//
// Structure (cdna_hip_programming.md, "Fused attention prefill"; everything derived from the
// MFMA lane maps in section 3 of that guide):
//  * one workgroup = 8 waves = 256 query rows of one (batch, head); each wave owns 32 rows and
//    keeps its Q fragment (32 VGPRs) and the O^T accumulator (64 VGPRs) in registers;
//  * K/V tiles of 64 keys are staged global -> registers -> LDS (issue early, write late) into
//    a double buffer; the LDS image is the dual-use XOR-swizzled image "(b)" of the guide: the
//    K tile is read row-wise with ds_read_b128, the V tile column-wise with
//    ds_read_b64_tr_b16, both conflict-free (tools/lds_sim.py);
//  * QK^T is computed swapped, S^T = K.Q^T (v_mfma_f32_32x32x16_bf16, K fragment in the A slot)
//    so a lane holds 32 scores of ONE query row: the row max / sum are in-register plus one
//    v_permlane32_swap, and the bf16-packed P registers are directly the B operand of
//    O^T += V^T.P^T (accumulator-as-operand with the permuted-k order of the guide);
//  * softmax scale and log2(e) are folded into one FMA in front of v_exp_f32.
#include "common.h"
#include "flexam_hip.h"

namespace {

constexpr int QBLK = 256;     // query rows per workgroup
constexpr int KVBLK = 64;     // keys per tile
constexpr int HD = 128;       // head dim (fixed)
constexpr int NT = 512;
constexpr int KV_TILE_BYTES = KVBLK * HD * 2;   // 16 KiB

struct AttnParams {
  const bf16* q;
  const bf16* k;
  const bf16* v;
  bf16* o;
  int64_t q_bs, q_rs;   // batch stride, row stride (elements); head h at column h*128
  int64_t k_bs, k_rs;
  int64_t v_bs, v_rs;
  int64_t o_bs, o_rs;
  int B, H, Lq, Lk;
  float scale_log2e;    // softmax_scale * log2(e)
  int q_blocks;
};

// byte offset of 16-byte chunk `ch` (0..15) of key row `row` in a [64][128] bf16 tile, image (b)
__device__ __forceinline__ int kv_off(int row, int ch) {
  return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3)));
}

// Exchange between lane l and lane l^32 (the two halves of a query row).  Inline asm on purpose:
// hipcc (ROCm 7.2) folds __builtin_amdgcn_permlane32_swap(x, x) as if both results were equal
// (observed: v_add v1, v1, v1), which silently drops the other half's value.  The s_nop covers the
// VALU-write -> v_permlane read hazard (2 wait states) inside the asm string.
__device__ __forceinline__ void half_swap(float& a, float& b) {
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ float pair_max(float x) {
  float a = x, b = x;
  half_swap(a, b);
  return fmaxf(a, b);
}
__device__ __forceinline__ float pair_sum(float x) {
  float a = x, b = x;
  half_swap(a, b);
  return a + b;
}

__global__ __launch_bounds__(NT, 2) void attn_fwd_kernel(AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][K tile | V tile]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;

  // ---- workgroup -> (batch, head, q block); XCD-chunked so one XCD's L2 serves few heads at a time
  const int nwg = p.B * p.H * p.q_blocks;
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, rr = nwg & 7, xcd = bid & 7, local = bid >> 3;
    bid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + local;
  }
  const int qb = bid % p.q_blocks;
  const int bh = bid / p.q_blocks;
  const int head = bh % p.H, b = bh / p.H;

  const bf16* qbase = p.q + (int64_t)b * p.q_bs + head * HD;
  const bf16* kbase = p.k + (int64_t)b * p.k_bs + head * HD;
  const bf16* vbase = p.v + (int64_t)b * p.v_bs + head * HD;

  // ---- Q fragment (B operand of S^T = K.Q^T): lane (r, h) holds Q[q0 + r][16*ds + 8h .. +7]
  const int q0 = qb * QBLK + wave * 32;
  const int qrow = min(q0 + r, p.Lq - 1);
  bf16x8 qf[8];
#pragma unroll
  for (int ds = 0; ds < 8; ++ds) qf[ds] = *(const bf16x8*)(qbase + (int64_t)qrow * p.q_rs + ds * 16 + h * 8);

  // ---- LDS read offsets
  // K row read: row = 32*kt + r, chunk = 2*ds + h
  int koff[8];
  {
    const int sw = ((r & 3) << 2) | ((r >> 2) & 3);
#pragma unroll
    for (int ds = 0; ds < 8; ++ds) koff[ds] = 256 * r + 16 * ((2 * ds + h) ^ sw);
  }
  // V transposed read: group g = lane>>4 (h = g>>1), i = lane&15, q_ = i>>2, p_ = i&3
  // block rows r0 + q_ with r0 = 32kt + 16s + 8*half + 4h, columns 32dt + 16(g&1) + 4p_ ..
  int voff[2][4];
  {
    const int g = lane >> 4, i = lane & 15, q_ = i >> 2, p_ = i & 3;
#pragma unroll
    for (int half = 0; half < 2; ++half)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
        voff[half][dt] = kv_off(8 * half + 4 * h + q_, 4 * dt + 2 * (g & 1) + (p_ >> 1)) + 8 * (p_ & 1);
    // (kv_off's swizzle depends on row&3 = q_ and (row>>2)&3 = 2*half + h: unchanged by + 32kt + 16s)
  }

  // ---- staging map: thread -> chunks id = tid + 512*i (i = 0, 1): row = id/16, ch = id%16
  int st_row[2], st_lds[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int id = tid + NT * i;
    st_row[i] = id >> 4;
    st_lds[i] = kv_off(id >> 4, id & 15);
  }
  const int st_col = (tid & 15) * 8;

  f32x16 o_acc[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int j = 0; j < 16; ++j) o_acc[dt][j] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const float c = p.scale_log2e;

  const int ntiles = (p.Lk + KVBLK - 1) / KVBLK;
  bf16x8 kreg[2], vreg[2];
  auto load_tile = [&](int t) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int key = min(t * KVBLK + st_row[i], p.Lk - 1);
      kreg[i] = *(const bf16x8*)(kbase + (int64_t)key * p.k_rs + st_col);
      vreg[i] = *(const bf16x8*)(vbase + (int64_t)key * p.v_rs + st_col);
    }
  };
  auto store_tile = [&](int buf) {
    char* kt_ = smem + buf * (2 * KV_TILE_BYTES);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      *(bf16x8*)(kt_ + st_lds[i]) = kreg[i];
      *(bf16x8*)(kt_ + KV_TILE_BYTES + st_lds[i]) = vreg[i];
    }
  };

  load_tile(0);
  store_tile(0);
  __syncthreads();

  for (int t = 0; t < ntiles; ++t) {
    const int buf = t & 1;
    const char* ktile = smem + buf * (2 * KV_TILE_BYTES);
    const char* vtile = ktile + KV_TILE_BYTES;
    if (t + 1 < ntiles) load_tile(t + 1);      // in flight during the MFMA work below

    // ---- S^T[kt] = K[kt] . Q^T      (keys on rows/registers, query on the lane)
    f32x16 s[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
      for (int j = 0; j < 16; ++j) s[kt][j] = 0.f;
#pragma unroll
      for (int ds = 0; ds < 8; ++ds) {
        const bf16x8 kf = *(const bf16x8*)(ktile + kt * 8192 + koff[ds]);
        s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ds], s[kt], 0, 0, 0);
      }
    }
    // ---- mask keys beyond Lk (last tile only; wave-uniform branch)
    if ((t + 1) * KVBLK > p.Lk) {
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const int key = t * KVBLK + 32 * kt + (j & 3) + 8 * (j >> 2) + 4 * h;
          if (key >= p.Lk) s[kt][j] = -INFINITY;
        }
    }
    // ---- online softmax for this lane's query row (the other half of the row lives in lane^32)
    float mx = s[0][0];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int j = 0; j < 16; ++j) mx = fmaxf(mx, s[kt][j]);
    mx = pair_max(mx);
    const float m_new = fmaxf(m_run, mx);
    const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
    const float mc = m_new * c;
    float psum = 0.f;
    bf16x8 pf[2][2];   // [kt][s]: B operand fragments of O^T += V^T . P^T
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kt][j], c, -mc));
        psum += pv;
        pf[kt][j >> 3][j & 7] = f2bf(pv);
      }
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int j = 0; j < 16; ++j) o_acc[dt][j] *= alpha;

    // ---- O^T[dt] += V^T[dt][keys] . P^T[keys]
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int ss = 0; ss < 2; ++ss) {
        const char* vb = vtile + kt * 8192 + ss * 4096;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vb + voff[0][dt]));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vb + voff[1][dt]));
          const bf16x4 lo_b = __builtin_bit_cast(bf16x4, lo), hi_b = __builtin_bit_cast(bf16x4, hi);
          bf16x8 vf;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            vf[j] = lo_b[j];
            vf[4 + j] = hi_b[j];
          }
          o_acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[kt][ss], o_acc[dt], 0, 0, 0);
        }
      }

    // ---- publish tile t+1
    if (t + 1 < ntiles) store_tile(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: O[q][32dt + 8i + 4h + (0..3)] = o_acc[dt][4i + (0..3)] / l
  const float inv_l = 1.0f / pair_sum(l_run);
  const int qi = q0 + r;
  if (qi < p.Lq) {
    bf16* orow = p.o + (int64_t)b * p.o_bs + (int64_t)qi * p.o_rs + head * HD;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        bf16x4 ov;
#pragma unroll
        for (int j = 0; j < 4; ++j) ov[j] = f2bf(o_acc[dt][4 * i + j] * inv_l);
        *(bf16x4*)(orow + 32 * dt + 8 * i + 4 * h) = ov;
      }
  }
}

}  // namespace

extern "C" int flexam_attn_fwd(const void* q, int64_t q_bs, int64_t q_rs, const void* k, int64_t k_bs, int64_t k_rs,
                               const void* v, int64_t v_bs, int64_t v_rs, void* o, int64_t o_bs, int64_t o_rs, int B, int H,
                               int Lq, int Lk, int head_dim, float softmax_scale, void* stream) {
  FX_REQUIRE(q && k && v && o, FLEXAM_E_ARG, "attn_fwd: null pointer");
  FX_REQUIRE(head_dim == HD, FLEXAM_E_SHAPE, "attn_fwd: head_dim %d unsupported (128 only)", head_dim);
  FX_REQUIRE(B > 0 && H > 0 && Lq > 0 && Lk > 0, FLEXAM_E_SHAPE, "attn_fwd: empty problem B=%d H=%d Lq=%d Lk=%d", B, H, Lq, Lk);
  FX_REQUIRE(q_rs % 8 == 0 && k_rs % 8 == 0 && v_rs % 8 == 0 && o_rs % 4 == 0 && q_bs % 8 == 0 && k_bs % 8 == 0 && v_bs % 8 == 0 &&
                 o_bs % 4 == 0,
             FLEXAM_E_SHAPE, "attn_fwd: strides must keep 16-byte alignment of head rows");
  FX_REQUIRE(((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) % 16 == 0 && (uintptr_t)o % 8 == 0, FLEXAM_E_ARG, "attn_fwd: misaligned pointer");
  AttnParams p;
  p.q = (const bf16*)q; p.k = (const bf16*)k; p.v = (const bf16*)v; p.o = (bf16*)o;
  p.q_bs = q_bs; p.q_rs = q_rs; p.k_bs = k_bs; p.k_rs = k_rs; p.v_bs = v_bs; p.v_rs = v_rs; p.o_bs = o_bs; p.o_rs = o_rs;
  p.B = B; p.H = H; p.Lq = Lq; p.Lk = Lk;
  p.scale_log2e = softmax_scale * 1.4426950408889634f;
  p.q_blocks = (Lq + QBLK - 1) / QBLK;
  static bool attr_set = false;
  const int smem = 4 * KV_TILE_BYTES;   // 64 KiB
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)attn_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
      return flexam_fail(FLEXAM_E_LAUNCH, "attn_fwd: cannot raise dynamic LDS to %d bytes", smem);
    attr_set = true;
  }
  hipLaunchKernelGGL(attn_fwd_kernel, dim3(B * H * p.q_blocks), dim3(NT), smem, (hipStream_t)stream, p);
  return flexam_check_launch("flexam_attn_fwd");
}
