// flexam_amd/csrc/gemm_fp8.hip -- fp8 (OCP e4m3) MFMA GEMM for the DiT's QKV / FFN projections (BASELINE configs[4], "fp8 MFMA
// QKV/FFN variant"; the reference itself only STORES weights in float8_e4m3fn and upcasts per call, FlexAM/utils/
// fp8_optimization.py:1-57 -- this is new arithmetic, selected explicitly on the engine, never the default).
//
//   C[M,N] = epilogue( (A8[M,K] . W8[N,K]^T) * sa[m] * sw[n] + bias[n] )     A8, W8 e4m3 row-major (K contiguous), sa / sw fp32
//
// Per-row dynamic scales for the activations (flexam_quantize_rows_fp8: absmax / 448), per-output-channel scales for the weights,
// fp32 accumulation on v_mfma_scale_f32_16x16x128_f8f6f4 with unit block scales (E8M0 127): 2x the bf16 MFMA rate.
//
// Structure = gemm.hip's with K counted in BYTES: a 128-byte row segment is 64 bf16 or 128 fp8 values, so the LDS image, the
// XOR swizzle, the LDS-DMA staging, the two K-block buffers, the persistent XCD-aware tile walk, the counted-vmcnt hand-over
// between units and the LDS-staged epilogues are the same.  What differs is the K block's inner order: one 16x16x128 MFMA
// consumes a lane group's 32 contiguous bytes (16-byte chunks 2g and 2g+1 of the row segment) of BOTH operands, so the block is
// cut by m-tiles instead of by K halves:
//   phase A: MFMAs of m-tiles [0, H0) while the A fragments of m-tiles [H0, MT) are read from `cur`
//   barrier : every wave holds all its fragments of this K block (cur is free); block kb+1 has landed in `nxt`
//   phase B: LDS-DMA of block kb+2 into `cur`; MFMAs of m-tiles [H0, MT) while W and the first A half of block kb+1 are read
// Both operands are read with the same (lane group, byte) -> k assignment, so the product does not depend on the k order the
// instruction uses inside a lane's 32 bytes (tools/probes/fp8_probe.hip checks the assumption on integer data).
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "flexam_hip.h"

namespace {

typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) int i32x4;

constexpr int BN = 256, BKB = 128;          // K block = 128 bytes per row = 128 fp8 values
constexpr int TILE_BYTES = 256 * BKB;       // 32 KiB per operand tile

struct Gemm8Params {
  const uint8_t* A;
  const uint8_t* W;
  void* C;
  const float* bias;
  const float* sa;         // [M] activation row scales
  const float* sw;         // [N] weight row (output channel) scales
  const float* so;         // EPI_GELU_Q: [M] scales of the e4m3 output rows (C = e4m3 bytes, ldc in bytes)
  int64_t lda, ldw, ldc;   // bytes == elements
  int M, N, K;
  int tiles_m, tiles_n;
  float* X;
  int64_t ldx;
  const float* gate;
  int64_t gate_ld;
  const int32_t* gate_row;
  int64_t rows_per_batch;
  int gm;
  int units;
};

template <int V>
using IC = std::integral_constant<int, V>;

enum { EPI_NONE = 0, EPI_GELU = 1, EPI_GATE_RESIDUAL = 2, EPI_GELU_Q = 3 };      // GELU_Q: GELU, then e4m3 / out_scale[m] (the next GEMM's A operand)

template <int EPI, int MT>
__global__ __launch_bounds__(512, 2) void gemm_fp8_kernel(Gemm8Params p) {
  constexpr int RT = 16, NRT = MT, NG = 4, NV = 4, NTW = 4;
  constexpr int STG_WAVE = RT * 128;
  constexpr int BM_ = 32 * MT;
  constexpr int PA = (BM_ + 63) / 64;
  constexpr int NP = PA + 4;
  constexpr int H0 = (MT + 1) / 2, H1 = MT - H0;      // m-tiles of phase A / phase B
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  // The builtin, so that hipcc places the wait state gfx950 wants between a VALU write of a VGPR and a v_readfirstlane of it (a
  // hand-written v_readfirstlane right behind the shift read a stale register: memory faults); the empty asm makes the SGPR value
  // opaque, so it is kept (or parked in a VGPR lane) instead of re-derived from a spilled copy of threadIdx.x in front of every use.
  int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  asm volatile("" : "+s"(wave));
  const int wm = wave >> 2, wn = wave & 3;

  const int nwg = p.units;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = blockIdx.x & 7, per_xcd = gridDim.x >> 3;
  const int chunk0 = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
  const int chunk_n = q8 + (xcd < r8 ? 1 : 0);
  const int local = blockIdx.x >> 3;
  const int n_units = local < chunk_n ? (chunk_n - local + per_xcd - 1) / per_xcd : 0;
  auto tile_origin = [&](int bid, int& m0, int& n0) {
    const int GM = p.gm;
    const int per_group = GM * p.tiles_n;
    const int group = bid / per_group;
    const int first_m = group * GM;
    const int gsz = min(p.tiles_m - first_m, GM);
    const int in_group = bid - group * per_group;
    m0 = (first_m + in_group % gsz) * BM_;
    n0 = (in_group / gsz) * BN;
  };
  uint32_t a_off[PA], w_off[4];
  const char *a_tile, *w_tile;
  auto stage_setup = [&](int m0, int n0) {
    const int ts = (wave << 6) | fresh_lane();          // thread id, rebuilt: nothing per-lane stays alive across the K loop
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = i * 64 + (ts >> 3);
      const int chunk = (ts & 7) ^ ((row >> 1) & 7);
      if (i < PA) a_off[i] = (uint32_t)((int64_t)min(row, p.M - 1 - m0) * p.lda + chunk * 16);
      w_off[i] = (uint32_t)((int64_t)min(row, p.N - 1 - n0) * p.ldw + chunk * 16);
    }
    a_tile = (const char*)(p.A + (int64_t)m0 * p.lda);
    w_tile = (const char*)(p.W + (int64_t)n0 * p.ldw);
  };
  // fragment read offsets: row = lane%16, 16-byte chunks 2g and 2g+1 (g = lane/16) of the 128-byte row segment, swizzled like the staging
  // (rebuilt at the top of every unit from a fresh lane id: alive in the K loop only, not across the epilogue)
  int frag_off[2];
  auto frag_setup = [&]() {
    const int lf = fresh_lane();
    const int sw = ((lf & 15) >> 1) & 7;
#pragma unroll
    for (int c = 0; c < 2; ++c) frag_off[c] = (lf & 15) * 128 + (((2 * (lf >> 4) + c) ^ sw) << 4);
  };
  bool staged = false, pend = false;
  constexpr int PEND = (EPI == EPI_GATE_RESIDUAL ? 4 : EPI == EPI_GELU_Q ? 1 : 2) * MT;      // stores a wave issues in an interior epilogue
  const int nk = p.K / BKB;

  for (int it = 0; it < n_units; ++it) {
  int m0, n0;
  tile_origin(chunk0 + local + it * per_xcd, m0, n0);
  if (!staged) stage_setup(m0, n0);

  f32x4 acc[MT][NTW];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NTW; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  auto dma = [&](int i, int64_t kbyte, char* buf) {
    const char* sbase = (i < PA ? a_tile : w_tile) + kbyte;
    const uint32_t voff = i < PA ? a_off[i] : w_off[i - PA];
    const uint32_t dst = (uint32_t)(uintptr_t)LDS_PTR(buf) + (i < PA ? i * 8192 : TILE_BYTES + (i - PA) * 8192) + wave * 1024;
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(dst) : "memory");
  };
  auto wait_barrier = [&](auto n_c) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(decltype(n_c)::value) : "memory");
    __syncthreads();
  };
  // fragment = 32 bytes (two 16-byte reads) of W n-tile j (j < 4) or A m-tile j - 4
  auto frag = [&](const char* buf, int j, i32x8& f) {
    const char* base = j < NTW ? buf + TILE_BYTES + wn * (16 * NTW * 128) + j * 2048 : buf + wm * (16 * MT * 128) + (j - NTW) * 2048;
    const i32x4 lo = *(const i32x4*)(base + frag_off[0]), hi = *(const i32x4*)(base + frag_off[1]);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      f[e] = lo[e];
      f[4 + e] = hi[e];
    }
  };
  i32x8 wf[NTW], af0[H0], af1[H1];
  // MFMA group nt of a phase: the m-tiles [g0, g0 + n) against W n-tile nt (W fragment in the A slot: a lane gets 4 consecutive n).
  // n-tile-major order: once group nt of phase B has issued, wf[nt] is dead and takes the NEXT block's W fragment while the
  // groups nt+1.. run -- no second set of W registers (4 x 8 VGPRs) is needed.
  auto mfma_group = [&](int nt, auto first_c) {
    constexpr bool FIRST = decltype(first_c)::value;
#pragma unroll
    for (int g = 0; g < (FIRST ? H0 : H1); ++g) {
      const int mt = FIRST ? g : H0 + g;
      acc[mt][nt] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf[nt], FIRST ? af0[g] : af1[g], acc[mt][nt], 0, 0, 0, 0x7F7F7F7F, 0,
                                                                     0x7F7F7F7F);
    }
  };
  using T_ = std::integral_constant<bool, true>;
  using F_ = std::integral_constant<bool, false>;

  if (!staged) {
#pragma unroll
    for (int i = 0; i < NP; ++i) dma(i, 0, smem);
    if (nk > 1) {
#pragma unroll
      for (int i = 0; i < NP; ++i) dma(i, BKB, smem + 2 * TILE_BYTES);
    }
  }
  const bool counted = pend && nk >= 3;
  if (counted) wait_barrier(IC<NP + PEND>{});
  else if (!pend && nk > 1) wait_barrier(IC<NP>{});
  else wait_barrier(IC<0>{});
  pend = false;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (i < PA) asm volatile("" : "+v"(a_off[i]));
    asm volatile("" : "+v"(w_off[i]));
  }
  // This unit's epilogue constants (bias and weight scale of the wave's 64 columns, activation scale of its 16 MT rows) start their
  // way into the wave's 1 KiB of LDS now, as LDS-DMA pieces behind everything the top wait left in flight: they land under the K
  // loop (its last barrier waits for vmcnt(0)).  As plain loads at the top of the epilogue they sat behind the next unit's 16
  // prefetched pieces in the in-order vmcnt queue and their first use waited for all of them.
  {
    const int le0 = fresh_lane();
    const uint32_t dst = (uint32_t)(uintptr_t)LDS_PTR(smem) + 4 * TILE_BYTES + 8 * STG_WAVE + wave * 1024;
    const uint32_t ncol4 = (uint32_t)min(n0 + wn * (16 * NTW) + le0, p.N - 1) * 4u;
    const int mr = m0 + wm * (16 * MT) + le0;
    const float* pb = p.bias;
    asm volatile("" : "+s"(pb));
    if (pb) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" ::"v"(ncol4), "s"(pb), "s"(dst) : "memory");
    else ((float*)(smem + 4 * TILE_BYTES + 8 * STG_WAVE + wave * 1024))[le0] = 0.f;
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" ::"v"(ncol4), "s"(p.sw), "s"(dst + 256) : "memory");
    const uint32_t r0 = (uint32_t)min(mr, p.M - 1) * 4u, r1 = (uint32_t)min(mr + 64, p.M - 1) * 4u;
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" ::"v"(r0), "s"(p.sa), "s"(dst + 512) : "memory");
    if (16 * MT > 64) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" ::"v"(r1), "s"(p.sa), "s"(dst + 768) : "memory");
    if constexpr (EPI == EPI_GELU_Q) {                   // the output row scales of the wave's 16 MT rows: a second kilobyte per wave
      const uint32_t dst2 = (uint32_t)(uintptr_t)LDS_PTR(smem) + 4 * TILE_BYTES + 8 * STG_WAVE + 8 * 1024 + wave * 512;
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" ::"v"(r0), "s"(p.so), "s"(dst2) : "memory");
      if (16 * MT > 64) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" ::"v"(r1), "s"(p.so), "s"(dst2 + 256) : "memory");
    }
  }
  frag_setup();
#pragma unroll
  for (int j = 0; j < NTW; ++j) frag(smem, j, wf[j]);
#pragma unroll
  for (int j = 0; j < H0; ++j) frag(smem, NTW + j, af0[j]);

  auto block = [&](int kb, auto dma_c, auto rd_c, auto wait_c) {
    constexpr bool DMA = decltype(dma_c)::value, RD = decltype(rd_c)::value;
    char* cur = smem + (kb & 1) * (2 * TILE_BYTES);
    char* nxt = smem + ((kb + 1) & 1) * (2 * TILE_BYTES);
    const int64_t kbyte = (int64_t)(kb + 2) * BKB;
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {                  // phase A: m-tiles [0, H0); the second A half is read meanwhile
      __builtin_amdgcn_sched_barrier(0);
      mfma_group(nt, T_{});
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = nt; j < H1; j += NTW) frag(cur, NTW + H0 + j, af1[j]);
    }
    __builtin_amdgcn_sched_barrier(0);
    wait_barrier(wait_c);
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {                  // phase B: m-tiles [H0, MT); DMA of block kb+2, fragments of block kb+1
      __builtin_amdgcn_sched_barrier(0);
      mfma_group(nt, F_{});
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (DMA) {
#pragma unroll
        for (int i = nt; i < NP; i += NTW) dma(i, kbyte, cur);
      }
      if constexpr (RD) {
        frag(nxt, nt, wf[nt]);
#pragma unroll
        for (int j = nt; j < H0; j += NTW) frag(nxt, NTW + j, af0[j]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  int kb = 0;
  if (counted) { block(0, T_{}, T_{}, IC<PEND>{}); kb = 1; }
  for (; kb + 2 < nk; ++kb) block(kb, T_{}, T_{}, IC<0>{});
  if (kb + 1 < nk) { block(kb, F_{}, T_{}, IC<0>{}); ++kb; }
  block(kb, F_{}, F_{}, IC<0>{});

  staged = it + 1 < n_units;
  if (staged) {
    int nm0, nn0;
    tile_origin(chunk0 + local + (it + 1) * per_xcd, nm0, nn0);
    stage_setup(nm0, nn0);
#pragma unroll
    for (int i = 0; i < NP; ++i) dma(i, 0, smem);
    if (nk > 1) {
#pragma unroll
      for (int i = 0; i < NP; ++i) dma(i, BKB, smem + 2 * TILE_BYTES);
    }
  }

  // ---- epilogue: lane holds C[m = .. + 16 t + lane%16][n = .. + 16 v + 4 (lane/16) + 0..3] per (m-tile t, n-tile v)
  const int le = fresh_lane();
  const int mrow = m0 + wm * (16 * MT) + (le & 15);
  const int ncol = n0 + wn * (16 * NTW) + (le >> 4) * 4;
  // Per-column bias / weight scale of the wave's 64 columns and the row scales of its 16 MT rows go through 1 KiB of LDS per wave
  // (brought in once per tile by LDS-DMA at the top of the unit, read back per (m-tile, n-tile) as one ds_read_b128 / ds_read_b32):
  // held in registers they were 39 values next to 16 MT accumulators and the packed outputs, and the MT = 7 instances spilled
  // 12-32 of them.
  const float* cst = (const float*)(smem + 4 * TILE_BYTES + 8 * STG_WAVE + wave * 1024);      // [64 bias][64 sw][128 sa], landed under the K loop
  auto yv = [&](int t, int v) -> f32x4 {
    const f32x4 b = *(const f32x4*)(cst + 16 * v + 4 * (le >> 4));
    const f32x4 w = *(const f32x4*)(cst + 64 + 16 * v + 4 * (le >> 4));
    const float sa_t = cst[128 + 16 * t + (le & 15)];
    return acc[t][v] * (w * sa_t) + b;
  };
  char* stg = smem + 4 * TILE_BYTES + wave * STG_WAVE;
  const int wr_row = le & 15, wr_g = le >> 4;
  auto stg_write = [&](int v, bf16x4 o) {
    *(bf16x4*)(stg + wr_row * 128 + (((2 * v + (wr_g >> 1)) ^ (wr_row & 7)) << 4) + (wr_g & 1) * 8) = o;
  };
  if constexpr (EPI == EPI_GATE_RESIDUAL) {
    if (m0 + BM_ <= p.M && n0 + BN <= p.N) {
      const int rr = le >> 4, cc = le & 15;
      const int mw = m0 + wm * (16 * MT);
      const int nw = n0 + wn * (16 * NTW) + cc * 4;
      const uint32_t xlane = (uint32_t)((rr * p.ldx + wn * (16 * NTW) + cc * 4) * 4);
      char* xtile = (char*)(p.X + (int64_t)mw * p.ldx + n0);
      bf16x4 yp[NRT][NV];
#pragma unroll
      for (int t = 0; t < NRT; ++t)
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          const f32x4 y4 = yv(t, v);
#pragma unroll
          for (int j = 0; j < 4; ++j) yp[t][v][j] = f2bf(y4[j]);
          asm volatile("" : "+v"(yp[t][v]));
        }
      auto rmw = [&](auto gate_c) {
        constexpr int GATE = decltype(gate_c)::value;
        constexpr int TPB = 2;                         // 16-row tiles per batch of 32 rows
#pragma unroll
        for (int t0 = 0; t0 < NRT; t0 += TPB) {
          const int ntb = NRT - t0 < TPB ? NRT - t0 : TPB;
          const int nit = ntb * 4;
          f32x4 xv[8], gv[8];
          int gr[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            if (i >= nit) continue;
            const int m = mw + t0 * 16 + 4 * i + rr;
            // non-temporal, like the bf16 GEMM's epilogues (gemm.hip): X and the e4m3 GELU output are touched once per launch
            xv[i] = __builtin_nontemporal_load((const f32x4*)(xtile + (int64_t)(t0 * 16 + 4 * i) * p.ldx * 4 + xlane));
            if constexpr (GATE == 1) gr[i] = p.gate_row[m];
            if constexpr (GATE == 2) gr[i] = m / p.rows_per_batch;
          }
          if constexpr (GATE != 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
              if (i >= nit) continue;
              gv[i] = *(const f32x4*)(p.gate + (int64_t)gr[i] * p.gate_ld + nw);
            }
          }
#pragma unroll
          for (int u = 0; u < TPB; ++u) {
            if (u >= ntb) continue;
#pragma unroll
            for (int v = 0; v < NV; ++v) stg_write(v, yp[t0 + u][v]);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int row = 4 * i + rr, idx = u * 4 + i;
              const bf16x4 y = *(const bf16x4*)(stg + row * 128 + (((cc >> 1) ^ (row & 7)) << 4) + (cc & 1) * 8);
              f32x4 x = xv[idx];
#pragma unroll
              for (int j = 0; j < 4; ++j) x[j] += GATE != 0 ? bf2f(y[j]) * gv[idx][j] : bf2f(y[j]);
              __builtin_nontemporal_store(x, (f32x4*)(xtile + (int64_t)(t0 * 16 + 4 * idx) * p.ldx * 4 + xlane));
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      };
      if (!p.gate) rmw(IC<0>{});
      else if (p.gate_row) rmw(IC<1>{});
      else rmw(IC<2>{});
      __builtin_amdgcn_s_waitcnt(0x0F70 | (PEND & 15) | ((PEND >> 4) << 14));
      pend = true;
      continue;
    }
  }
  if constexpr (EPI == EPI_GELU_Q) {
    // e4m3 output rows: a lane packs its 4 columns of (m-tile t, n-tile v) into one dword; the wave's 16 x 64 bytes of an m-tile are
    // turned around in its staging slice (16-byte chunks XOR-swizzled by row pairs) and leave as 16 rows of 64 contiguous bytes
    const float* so_l = (const float*)(smem + 4 * TILE_BYTES + 8 * STG_WAVE + 8 * 1024 + wave * 512);
    if (m0 + BM_ <= p.M && n0 + BN <= p.N) {
      const int rd_row = le >> 2, rd_c = le & 3;
      uint8_t* crow = (uint8_t*)p.C + (int64_t)(m0 + wm * (16 * MT) + rd_row) * p.ldc + n0 + wn * (16 * NTW) + rd_c * 16;
#pragma unroll
      for (int t = 0; t < NRT; ++t) {
        const float inv = __builtin_amdgcn_rcpf(so_l[16 * t + (le & 15)]);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          const f32x4 y4 = yv(t, v);
          int w = 0;
          w = __builtin_amdgcn_cvt_pk_fp8_f32(gelu_tanh(y4[0]) * inv, gelu_tanh(y4[1]) * inv, w, false);
          w = __builtin_amdgcn_cvt_pk_fp8_f32(gelu_tanh(y4[2]) * inv, gelu_tanh(y4[3]) * inv, w, true);
          *(int*)(stg + wr_row * 64 + ((v ^ ((wr_row >> 1) & 3)) << 4) + wr_g * 4) = w;
        }
        const u32x4 o = *(const u32x4*)(stg + rd_row * 64 + ((rd_c ^ ((rd_row >> 1) & 3)) << 4));
        __builtin_nontemporal_store(o, (u32x4*)(crow + (int64_t)(t * 16) * p.ldc));
      }
      __builtin_amdgcn_s_waitcnt(0x0F70 | (PEND & 15) | ((PEND >> 4) << 14));
      pend = true;
      continue;
    }
#pragma unroll
    for (int t = 0; t < NRT; ++t) {                      // boundary tiles: 4 bytes per lane, bounds-checked
      const int m = mrow + t * 16;
      if (m >= p.M) continue;
      const float inv = __builtin_amdgcn_rcpf(so_l[16 * t + (le & 15)]);
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int n = ncol + v * 16;
        if (n >= p.N) continue;
        const f32x4 y4 = yv(t, v);
        int w = 0;
        w = __builtin_amdgcn_cvt_pk_fp8_f32(gelu_tanh(y4[0]) * inv, gelu_tanh(y4[1]) * inv, w, false);
        w = __builtin_amdgcn_cvt_pk_fp8_f32(gelu_tanh(y4[2]) * inv, gelu_tanh(y4[3]) * inv, w, true);
        *(int*)((uint8_t*)p.C + (int64_t)m * p.ldc + n) = w;
      }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    continue;
  }
  if constexpr (EPI != EPI_GATE_RESIDUAL && EPI != EPI_GELU_Q) {
    if (m0 + BM_ <= p.M && n0 + BN <= p.N) {
      const int rd_row = le >> 3, rd_c = le & 7;
      bf16* crow = (bf16*)p.C + (int64_t)(m0 + wm * (16 * MT) + rd_row) * p.ldc + n0 + wn * (16 * NTW) + rd_c * 8;
#pragma unroll
      for (int t = 0; t < NRT; ++t) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          f32x4 y4 = yv(t, v);
          if constexpr (EPI == EPI_GELU) {
#pragma unroll
            for (int j = 0; j < 4; ++j) y4[j] = gelu_tanh(y4[j]);
          }
          bf16x4 o;
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = f2bf(y4[j]);
          stg_write(v, o);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int row = 8 * i + rd_row;
          const bf16x8 o8 = *(const bf16x8*)(stg + row * 128 + ((rd_c ^ (row & 7)) << 4));
          *(bf16x8*)(crow + (int64_t)(t * 16 + 8 * i) * p.ldc) = o8;
        }
      }
      __builtin_amdgcn_s_waitcnt(0x0F70 | (PEND & 15) | ((PEND >> 4) << 14));
      pend = true;
      continue;
    }
  }
#pragma unroll
  for (int t = 0; t < NRT; ++t) {
    const int m = mrow + t * 16;
    if (m >= p.M) continue;
    const float* grow = nullptr;
    if constexpr (EPI == EPI_GATE_RESIDUAL) {
      if (p.gate) {
        const int64_t r = p.gate_row ? (int64_t)p.gate_row[m] : (int64_t)m / p.rows_per_batch;
        grow = p.gate + r * p.gate_ld;
        asm volatile("" ::"v"(grow));
      }
    }
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int n = ncol + v * 16;
      if (n >= p.N) continue;
      f32x4 y4 = yv(t, v);
      if constexpr (EPI == EPI_GELU) {
#pragma unroll
        for (int j = 0; j < 4; ++j) y4[j] = gelu_tanh(y4[j]);
      }
      if constexpr (EPI == EPI_GATE_RESIDUAL) {
        float* xp = p.X + (int64_t)m * p.ldx + n;
        f32x4 x = *(const f32x4*)xp;
        f32x4 g = grow ? *(const f32x4*)(grow + n) : (f32x4){1.f, 1.f, 1.f, 1.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) x[j] += bf2f(f2bf(y4[j])) * g[j];
        *(f32x4*)xp = x;
      } else {
        bf16x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = f2bf(y4[j]);
        *(bf16x4*)((bf16*)p.C + (int64_t)m * p.ldc + n) = o;
      }
    }
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);
  }   // tile loop
}

// x[m, :] (bf16) -> q[m, :] = e4m3(x / s[m]), s[m] = absmax(x[m, :]) / 448 (1 where the row is all zero); one wave per row
__global__ __launch_bounds__(256) void quantize_rows_fp8_kernel(const bf16* __restrict__ x, int64_t ldx, uint8_t* __restrict__ q, int64_t ldq,
                                                                float* __restrict__ scale, int64_t M, int K) {
  const int lane = threadIdx.x & 63;
  for (int64_t m = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); m < M; m += (int64_t)gridDim.x * 4) {
    const bf16* xr = x + m * ldx;
    float amax = 0.f;
    for (int c = lane * 8; c < K; c += 512) {
      const bf16x8 v = *(const bf16x8*)(xr + c);
#pragma unroll
      for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(bf2f(v[j])));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
    const float s = amax > 0.f ? amax * (1.0f / 448.0f) : 1.0f;
    const float inv = 1.0f / s;
    if (lane == 0) scale[m] = s;
    uint8_t* qr = q + m * ldq;
    for (int c = lane * 8; c < K; c += 512) {
      const bf16x8 v = *(const bf16x8*)(xr + c);
      u32x2 o;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        int w = 0;
        w = __builtin_amdgcn_cvt_pk_fp8_f32(bf2f(v[4 * h]) * inv, bf2f(v[4 * h + 1]) * inv, w, false);
        w = __builtin_amdgcn_cvt_pk_fp8_f32(bf2f(v[4 * h + 2]) * inv, bf2f(v[4 * h + 3]) * inv, w, true);
        o[h] = (unsigned)w;
      }
      *(u32x2*)(qr + c) = o;
    }
  }
}

int pick_mt8(int M, int tiles_n) {
  const char* e = getenv("FLEXAM_GEMM_MT");
  const int forced = e ? atoi(e) : 0;
  if (forced >= 4 && forced <= 7) return forced;
  int best = 7;
  double best_cost = 1e30;
  const int G = flexam_num_cus();
  // 224-row tiles at most: a 256-row instance would keep 64 A + 32 W fragment registers next to 128 accumulators and spill
  // (it ran at 1.5-1.7 PF against 2.1-2.2 for MT = 7 at the FFN2 shapes, tools/fp8_ffn2_sweep.py, and is no longer compiled)
  for (int mt = 7; mt >= 4; --mt) {
    const int tiles = (int)((long)((M + 32 * mt - 1) / (32 * mt)) * tiles_n);
    const double cost = ((tiles + G - 1) / G) * (mt + 1.25);
    if (cost < best_cost * 0.97) { best_cost = cost; best = mt; }
  }
  return best;
}

template <int EPI, int MT>
int launch_shape8(Gemm8Params p, hipStream_t s) {
  auto kern = gemm_fp8_kernel<EPI, MT>;
  static bool attr_set[FLEXAM_MAX_DEVICES] = {};
  const int smem = 4 * TILE_BYTES + 8 * 16 * 128 + 8 * 1024 + (EPI == EPI_GELU_Q ? 8 * 512 : 0);      // K-block buffers, output staging, epilogue constants (+ output row scales)
  const int dev = flexam_current_device();
  if (!attr_set[dev]) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
      return flexam_fail(FLEXAM_E_LAUNCH, "gemm_fp8: cannot raise dynamic LDS to %d bytes", smem);
    attr_set[dev] = true;
  }
  p.tiles_m = (p.M + 32 * MT - 1) / (32 * MT);
  p.units = p.tiles_m * p.tiles_n;
  int grid = (p.units + 7) / 8 * 8;
  if (grid > flexam_num_cus()) grid = flexam_num_cus();
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), smem, s, p);
  return flexam_check_launch("flexam_gemm_fp8");
}

template <int EPI>
int launch8(Gemm8Params p, hipStream_t s) {
  {
    const char* g = getenv("FLEXAM_GEMM_GM");
    p.gm = g && atoi(g) >= 1 ? atoi(g) : 4;
  }
  switch (pick_mt8(p.M, p.tiles_n)) {
    case 6: return launch_shape8<EPI, 6>(p, s);
    case 5: return launch_shape8<EPI, 5>(p, s);
    case 4: return launch_shape8<EPI, 4>(p, s);
    default: return launch_shape8<EPI, 7>(p, s);      // no 256-row instance: it needs 64 A + 32 W fragment registers next to 128 accumulators and spilled 60-150 of them
  }
}

int check8(const void* A, int64_t lda, const void* W, int64_t ldw, const float* sa, const float* sw, int64_t M, int64_t N, int64_t K) {
  FX_REQUIRE(A && W && sa && sw, FLEXAM_E_ARG, "gemm_fp8: null pointer");
  FX_REQUIRE(M > 0 && N > 0 && K > 0, FLEXAM_E_SHAPE, "gemm_fp8: empty problem M=%ld N=%ld K=%ld", (long)M, (long)N, (long)K);
  FX_REQUIRE(K % BKB == 0, FLEXAM_E_SHAPE, "gemm_fp8: K=%ld must be a multiple of %d", (long)K, BKB);
  FX_REQUIRE(N % 4 == 0 && lda % 16 == 0 && ldw % 16 == 0, FLEXAM_E_SHAPE, "gemm_fp8: N %% 4, lda %% 16, ldw %% 16 required");
  FX_REQUIRE(((uintptr_t)A | (uintptr_t)W) % 16 == 0, FLEXAM_E_ARG, "gemm_fp8: operands must be 16-byte aligned");
  return FLEXAM_OK;
}

}  // namespace

extern "C" int flexam_quantize_rows_fp8(const void* x, int64_t ldx, void* q, int64_t ldq, float* scale, int64_t M, int K, void* stream) {
  FX_REQUIRE(x && q && scale && M > 0 && K > 0, FLEXAM_E_ARG, "quantize_rows_fp8: null pointer or empty");
  FX_REQUIRE(K % 8 == 0 && ldx % 8 == 0 && ldq % 8 == 0, FLEXAM_E_SHAPE, "quantize_rows_fp8: K, ldx, ldq must be multiples of 8");
  const int64_t blocks = (M + 3) / 4;
  hipLaunchKernelGGL(quantize_rows_fp8_kernel, dim3((unsigned)(blocks > 65535 ? 65535 : blocks)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16*)x, ldx, (uint8_t*)q, ldq, scale, M, K);
  return flexam_check_launch("flexam_quantize_rows_fp8");
}

extern "C" int flexam_gemm_fp8(const void* A, int64_t lda, const float* a_scale, const void* W, int64_t ldw, const float* w_scale,
                               const float* bias, void* C, int64_t ldc, int64_t M, int64_t N, int64_t K, int epilogue, void* stream) {
  if (int rc = check8(A, lda, W, ldw, a_scale, w_scale, M, N, K)) return rc;
  FX_REQUIRE(C && ldc % 4 == 0 && (uintptr_t)C % 16 == 0, FLEXAM_E_ARG, "gemm_fp8: bad output");
  FX_REQUIRE(epilogue == EPI_NONE || epilogue == EPI_GELU, FLEXAM_E_ARG, "gemm_fp8: unknown epilogue %d", epilogue);
  Gemm8Params p{};
  p.A = (const uint8_t*)A; p.W = (const uint8_t*)W; p.C = C; p.bias = bias; p.sa = a_scale; p.sw = w_scale;
  p.lda = lda; p.ldw = ldw; p.ldc = ldc; p.M = (int)M; p.N = (int)N; p.K = (int)K;
  p.tiles_n = (int)((N + BN - 1) / BN);
  return epilogue == EPI_GELU ? launch8<EPI_GELU>(p, (hipStream_t)stream) : launch8<EPI_NONE>(p, (hipStream_t)stream);
}

extern "C" int flexam_gemm_fp8_gelu_q(const void* A, int64_t lda, const float* a_scale, const void* W, int64_t ldw, const float* w_scale,
                                      const float* bias, const float* out_scale, void* Q, int64_t ldq, int64_t M, int64_t N, int64_t K,
                                      void* stream) {
  if (int rc = check8(A, lda, W, ldw, a_scale, w_scale, M, N, K)) return rc;
  FX_REQUIRE(Q && out_scale && ldq % 16 == 0 && (uintptr_t)Q % 16 == 0, FLEXAM_E_ARG, "gemm_fp8_gelu_q: bad output (e4m3 rows, 16-byte aligned)");
  Gemm8Params p{};
  p.A = (const uint8_t*)A; p.W = (const uint8_t*)W; p.C = Q; p.bias = bias; p.sa = a_scale; p.sw = w_scale; p.so = out_scale;
  p.lda = lda; p.ldw = ldw; p.ldc = ldq; p.M = (int)M; p.N = (int)N; p.K = (int)K;
  p.tiles_n = (int)((N + BN - 1) / BN);
  return launch8<EPI_GELU_Q>(p, (hipStream_t)stream);
}

extern "C" int flexam_gemm_fp8_gate_residual(const void* A, int64_t lda, const float* a_scale, const void* W, int64_t ldw,
                                             const float* w_scale, const float* bias, float* X, int64_t ldx, const float* gate,
                                             int64_t gate_ld, const int32_t* gate_row, int64_t rows_per_batch, int64_t M, int64_t N,
                                             int64_t K, void* stream) {
  if (int rc = check8(A, lda, W, ldw, a_scale, w_scale, M, N, K)) return rc;
  FX_REQUIRE(X && ldx % 4 == 0, FLEXAM_E_ARG, "gemm_fp8_gate_residual: bad residual");
  FX_REQUIRE(!gate || gate_row || rows_per_batch > 0, FLEXAM_E_ARG, "gemm_fp8_gate_residual: gate needs gate_row or rows_per_batch");
  Gemm8Params p{};
  p.A = (const uint8_t*)A; p.W = (const uint8_t*)W; p.bias = bias; p.sa = a_scale; p.sw = w_scale;
  p.lda = lda; p.ldw = ldw; p.M = (int)M; p.N = (int)N; p.K = (int)K;
  p.tiles_n = (int)((N + BN - 1) / BN);
  p.X = X; p.ldx = ldx; p.gate = gate; p.gate_ld = gate_ld; p.gate_row = gate_row;
  p.rows_per_batch = rows_per_batch > 0 ? rows_per_batch : 1;
  return launch8<EPI_GATE_RESIDUAL>(p, (hipStream_t)stream);
}
