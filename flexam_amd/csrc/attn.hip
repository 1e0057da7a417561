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
//  * K/V tiles of 64 keys go global -> LDS by LDS-DMA into two 4-slot rings (details at the kernel); the LDS
//    image is the dual-use XOR-swizzled image "(b)" of the guide: the K tile is read row-wise with
//    ds_read_b128, the V tile column-wise with ds_read_b64_tr_b16, both conflict-free (tools/lds_sim.py);
//  * QK^T is computed swapped, S^T = K.Q^T (v_mfma_f32_32x32x16_bf16, K fragment in the A slot)
//    so a lane holds 32 scores of ONE query row: the row max / sum are in-register plus one
//    v_permlane32_swap, and the bf16-packed P registers are directly the B operand of
//    O^T += V^T.P^T (accumulator-as-operand with the permuted-k order of the guide);
//  * softmax scale and log2(e) are folded into one FMA in front of v_exp_f32, or (PRE) into q by its producer.
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "flexam_hip.h"

namespace {

constexpr int QBLK = 256;     // query rows per workgroup
constexpr int KVBLK = 64;     // keys per tile
constexpr int HD = 128;       // head dim (fixed)
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
  int prescaled;        // q already carries that factor (see flexam_hip.h): the kernel instance with PRE = true runs
  int q_blocks;
  // split-KV (kv_splits > 1): workgroup (q block, head, split) covers key tiles [split*tiles_per_split, ..) and writes its
  // un-normalised O (fp32) and (running max, row sum) to the workspace; attn_merge_kernel combines the splits
  int kv_splits, tiles_per_split;
  int unit0, n_units;   // this launch covers work units (q block, head, batch) unit0 .. unit0 + n_units - 1
  // whole_units > 0: ONE launch for a split-KV call.  Units 0 .. whole_units - 1 (= unit0) run all keys in one workgroup each and
  // write O; the n_units units behind them run as kv_splits workgroups each and write partials.  Every XCD gets an eighth of both
  // kinds, its whole units first: the short workgroups start as the CUs of that XCD finish their last whole unit, without a launch
  // boundary (and its drain) in between.
  int whole_units;
  float* ws_o;          // [slots][n_units][256 rows][128]
  float* ws_ml;         // [slots][n_units][256 rows][2]
  // partial: write un-normalised O and (reference, row sum) to workspace slot slot0 + split even with one key range (the keys of
  // this call are only PART of the softmax: local-chunk-first attention under a sequence-parallel K|V all-gather);
  // n_slots: slots the merge kernel adds up
  int partial, slot0, n_slots;
  float last_key_bias;  // added to the score of key Lk - 1 in exp2 units (log2 of its multiplicity), 0 = none: see flexam_attn_fwd_lastkey
  // MXFP8 operands (attn_fp8.inc; q8 != nullptr selects that kernel): written by flexam_attn_fp8_pack
  const unsigned char* q8;   // [B][H][lq_pad][128] e4m3
  const unsigned* qs;        // [B][H][lq_pad] four E8M0 bytes per row
  const unsigned char* kv8;  // [B][H][ceil(Lk / 64)] records of REC_BYTES -- or, in chunks of kv8_chunk_tiles key tiles, [chunk][B][H][kv8_chunk_tiles]
  int lq_pad;
  // key tile g lives in chunk g / kv8_chunk_tiles (flexam_attn_fwd_fp8_chunked: the rank-major result of a sequence-parallel all-gather of
  // every rank's records; one chunk = all tiles for the plain call); kv8_chunk_magic = ceil(2^32 / kv8_chunk_tiles) turns the division
  // into one s_mul_hi_u32 (exact for g, tiles < 2^16)
  int kv8_chunk_tiles;
  unsigned kv8_chunk_magic;
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
__device__ __forceinline__ void half_swap_u32(unsigned& a, unsigned& b) {
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
// 16-byte-per-lane LDS-DMA issued from inline asm: hipcc then does not know an LDS write is in flight and does
// not put s_waitcnt vmcnt(0) in front of every ds_read_b64_tr_b16 intrinsic (it does for the builtin form,
// which would drain the DMA in the middle of a tile).  Completion is tracked by hand: vmcnt(0) + s_barrier
// at the top of each 64-key tile.  M0 (LDS base of the DMA) is saved / restored inside the statement.
__device__ __forceinline__ void lds_dma16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst)
               : "memory");
}

// max of MFMA outputs as single instructions: fmaxf() makes hipcc put a canonicalising v_max_f32 x, x in front of every operand it
// cannot prove quiet (each MFMA result), 4-5 extra vector instructions per half tile in a loop that is bound by vector issue
__device__ __forceinline__ float vmax3(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
// max of 8 values in ONE statement: between separate asm statements that feed each other hipcc pads an s_nop per dependency
__device__ __forceinline__ float vmax8(float a, float b, float c, float d, float e, float f, float g, float h) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3\n\tv_max3_f32 %0, %0, %4, %5\n\tv_max3_f32 %0, %0, %6, %7\n\tv_max_f32 %0, %0, %8"
      : "=&v"(r)
      : "v"(a), "v"(b), "v"(c), "v"(d), "v"(e), "v"(f), "v"(g), "v"(h));
  return r;
}
// max(a, b) combined across the two lanes (l, l ^ 32) that hold one query row, one statement (wait states inside)
__device__ __forceinline__ float pair_max2(float a, float b) {
  float r, t;
  asm("v_max_f32 %0, %2, %3\n\tv_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 0\n\tv_max_f32 %0, %0, %1"
      : "=&v"(r), "=&v"(t)
      : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float vmax(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// scalar-base form of the same DMA: uniform 64-bit base in SGPRs + one loop-invariant 32-bit per-lane offset, so a full tile's four
// pieces cost no vector address arithmetic; M0 is written and not restored (nothing else in this kernel reads it: LDS instructions
// on gfx9+ do not)
__device__ __forceinline__ void lds_dma16_sbase(const char* sbase, unsigned voff, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

__device__ __forceinline__ float pair_max(float x) {
  float a = x, b = x;
  half_swap(a, b);
  return vmax(a, b);
}
__device__ __forceinline__ float pair_sum(float x) {
  float a = x, b = x;
  half_swap(a, b);
  return a + b;
}

// ------------------------------------------------------------------------------------------------
// Structure: 8 waves x 32 query rows = 256 rows per workgroup, two waves per SIMD.
//  * K/V tiles of 64 keys go HBM/L2 -> LDS by LDS-DMA (global_load_lds, 16 B/lane, swizzle applied to
//    the source address) into a ring of 4 slots: no staging registers, no ds_write; ONE s_barrier per
//    64 keys, and the DMA of tile t+2 is in flight for two tiles before the vmcnt(0) that precedes it;
//  * the softmax pipeline runs on 32-key HALF tiles g.  Step g issues on the matrix pipe
//        S(g+1) = K(g+1).Q^T          8 MFMA   (scores of the next half; S double-buffered: 32 VGPRs)
//        O^T   += V(g-1)^T.P(g-1)^T   8 MFMA   (PV of the previous half; P kept packed in 8 VGPRs)
//    in the same basic blocks as the VALU work of half g (row max, exp2, row sum, bf16 pack), so the
//    wave overlaps its own MFMAs with its own softmax, and its SIMD partner fills the remaining gaps;
//  * deferred rescale: the running max is raised (O, l and the pending P rescaled) only when a row max
//    grew by more than 2^THR; P <= 2^THR is harmless in bf16 / fp32 floating point.
// (A 4-wave x 64-row variant -- each K/V fragment feeding two MFMAs -- was tried in r1: hipcc cannot keep
//  Q in the accumulator file and spills 150+ VGPRs; see DESIGN.md.)
// ------------------------------------------------------------------------------------------------
#if (defined(FLEXAM_ATTN_STAMPS) || defined(A32_NOMAX_ABLATE) || defined(A32_VALU) || defined(FLEXAM_ATTN_BODY16) || defined(A32_RESCALE_THR) || defined(A32_HALF_FRAG_ABLATE)) && !defined(FLEXAM_DIAGNOSTIC_BUILD)
#error "FLEXAM_ATTN_STAMPS / A32_NOMAX_ABLATE / A32_VALU / FLEXAM_ATTN_BODY16 / A32_RESCALE_THR / A32_HALF_FRAG_ABLATE are switches of diagnostic builds (timing ablations give WRONG results): add -DFLEXAM_DIAGNOSTIC_BUILD (tools/build_attn_variants.py does)"
#endif
#ifndef A32_DEFER
#define A32_DEFER 0      // scores of a half tile whose exp2 / sum / pack wait for part A of the next step (see stepA); 0 = none
#endif
#ifdef FLEXAM_ATTN_STAMPS      // diagnostic builds only (MI355X_MICROARCH.md, DVFS give-back item 6): the in-kernel clock of the main loop
__device__ unsigned long long g_attn_stamps[2 * 8192];      // per workgroup: shader cycles and 100 MHz ticks across the tile loop; read by nobody on the device
__device__ unsigned long long g_attn_barrier_wait[8 * 8192];   // per workgroup and wave: shader cycles spent in the tile loop's s_barrier
#define ATTN_STAMP_DECL() unsigned long long st0_ = 0, sr0_ = 0, bw_ = 0
#define ATTN_STAMP_BEGIN() st0_ = __builtin_amdgcn_s_memtime(), sr0_ = __builtin_amdgcn_s_memrealtime()
#define ATTN_STAMP_BARRIER(stmt) { const unsigned long long b0_ = __builtin_amdgcn_s_memtime(); stmt; bw_ += __builtin_amdgcn_s_memtime() - b0_; }
#define ATTN_STAMP_END()                                                             \
  if ((threadIdx.x & 63) == 0 && blockIdx.x < 8192) g_attn_barrier_wait[8 * blockIdx.x + (threadIdx.x >> 6)] = bw_;  \
  if (threadIdx.x == 0 && blockIdx.x < 8192) {                                       \
    g_attn_stamps[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - st0_;             \
    g_attn_stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - sr0_;     \
  }
#else
#define ATTN_STAMP_DECL()
#define ATTN_STAMP_BEGIN()
#define ATTN_STAMP_BARRIER(stmt) stmt
#define ATTN_STAMP_END()
#endif
constexpr int NT = 512;
constexpr int NSLOT = 4;
constexpr int V_RING = NSLOT * KV_TILE_BYTES;   // LDS: [4 K slots][4 V slots]
template <int V>
using IC = std::integral_constant<int, V>;
// Deferred-rescale threshold in exp2 units.  A trade, measured in r6 (profiles/r6zb_*): every deferral leaves the row's dominant key at a
// P that is not exactly 1, i.e. with a bf16 rounding error the exact-maximum form does not have -- attention error against fp64 on
// peaked rows (logit std 6, L = 11648) 1.51e-3 rms at 2^0, 1.77e-3 at 2^8, 2.10e-3 at 2^24 -- while the rescale branch costs
// +1.2 % of a denoise step at 2^4 and -0.5 / -0.9 / -1.3 ... -2.1 % at 2^12 / 2^16 / 2^24 on such rows (nothing on N(0, 1) logits).
// 8 stays: parity before speed.  -DA32_RESCALE_THR=n builds the other points (diagnostic builds).
#ifdef A32_RESCALE_THR
constexpr float RESCALE_THR_LOG2 = (float)A32_RESCALE_THR;
#else
constexpr float RESCALE_THR_LOG2 = 8.0f;
#endif

// KIND: 0 = self-attention, 1 = short-context (text) attention: distinct profiler symbols.
// PRE: q was multiplied by softmax_scale * log2(e) by its producer, BEFORE its one rounding to bf16 (in the DiT: folded into the
// RMSNorm weight of q).  The scores then leave the MFMA chain in exp2 units, and with the row reference of the online softmax as
// the chain's initial accumulator p = exp2(S') needs no FMA per score (16 of ~85 vector operations per 16 MFMAs).
// FULL (Lk >= 64, no weighted key; the DiT's self-attention): the common path of a half-tile step is ONE basic block apart from the
// rescale branch -- across block boundaries hipcc sinks MFMAs to the block of their first use, which serialises the pipeline
// (-0.9 % of a denoise step, profiles/r4ab_*).  To that end: the last tile of a key range that is not a multiple of 64 is the window
// [Lk - 64, Lk) instead of [64 (T - 1), 64 T) -- every tile is a whole in-bounds 64-row LDS-DMA with the same per-lane offsets, and
// its keys that the tile before already held are masked instead of keys past the end; the mask exists only in the 1-4 tail tiles
// of a work unit (branch-free there), the main loop has none; LDS-DMA past the end re-reads the last tile into a free slot; the
// barrier of tile 0 is kept.
// SHORT (at most 4 key tiles, i.e. the text cross-attention): one persistent workgroup per CU walks a contiguous range of work units
// (q blocks of one (batch, head), then of the next); the K/V tiles of a head stay in the ring while its q blocks last -- one LDS-DMA
// prologue and barrier per head change instead of per q block, none inside (the tiles are read-only), so the waves run through
// their rows of the blocks on their own.
template <int KIND, bool PRE, bool FULL = false, bool SHORT = false>
__global__ __launch_bounds__(NT, 2) void attn_fwd_kernel(AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [4 K tiles][4 V tiles], each a ring
  ATTN_STAMP_DECL();
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  // (Wave priorities: at equal priority the older half of the workgroup (waves 0-3) wins every arbitration against its SIMD partners,
  //  runs ahead and waits ~800 cycles per tile at the barrier; a static priority for waves 4-7 mirrors that, and every scheme that
  //  balances the halves -- priority in one part of the step only, turns by step or by tile -- is slower: while the leader waits its
  //  partner has the SIMD to itself.  profiles/r4u_attn_wave_priority_modes_and_barrier_wait.txt)

  // workgroups bid = 8 * local + xcd go to XCD `xcd` in the order of `local`: an XCD walks a contiguous eighth of the work list
  auto eighth = [](int n, int xcd, int local) -> int {      // index into a list of n items, -1 past this XCD's share
    const int q = n >> 3, rr = n & 7;
    if (local >= q + (xcd < rr ? 1 : 0)) return -1;
    return (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + local;
  };
  int split = 0, ul, unit, tps = p.tiles_per_split, partial = p.partial;
  int unit_end;                              // SHORT: this workgroup walks units [unit, unit_end)
  if constexpr (SHORT) {
    const int64_t all = (int64_t)p.B * p.H * p.q_blocks;
    ul = unit = (int)(all * blockIdx.x / gridDim.x);
    unit_end = (int)(all * (blockIdx.x + 1) / gridDim.x);
    if (unit >= unit_end) return;
  } else {
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    const int wmax = (p.whole_units + 7) >> 3;
    if (local < wmax) {                              // (whole_units = 0: wmax = 0)
      unit = eighth(p.whole_units, xcd, local);
      if (unit < 0) return;
      ul = unit;
      tps = (p.Lk + KVBLK - 1) / KVBLK;
      partial = 0;
    } else {
      // consecutive workgroups: the units of one split, i.e. q blocks of one head first -> same K/V range in L2
      const int j = eighth(p.n_units * p.kv_splits, xcd, local - wmax);
      if (j < 0) return;
      split = j / p.n_units;
      ul = j - split * p.n_units;                    // unit index inside the launch's split (or only) part
      unit = p.unit0 + ul;
    }
  }
  if constexpr (!SHORT) unit_end = unit + 1;
  // (batch, head) of the unit being worked on: set at the top of every unit of the walk below
  int head = 0, b = 0, bh_loaded = -1;
  const bf16* qbase = nullptr;
  const char *kbase = nullptr, *vbase = nullptr;

  // ---- LDS read offsets (dual-use swizzled image, see kv_off)
  int koff[8];
  {
    const int sw = ((r & 3) << 2) | ((r >> 2) & 3);
#pragma unroll
    for (int ds = 0; ds < 8; ++ds) koff[ds] = 256 * r + 16 * ((2 * ds + h) ^ sw);
  }
  int voff[2][4];
  {
    const int g = lane >> 4, i = lane & 15, q_ = i >> 2, p_ = i & 3;
#pragma unroll
    for (int half = 0; half < 2; ++half)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
        voff[half][dt] = kv_off(8 * half + 4 * h + q_, 4 * dt + 2 * (g & 1) + (p_ >> 1)) + 8 * (p_ & 1);
  }

  // ---- LDS-DMA staging: piece id = tid + 512*i (i < 2) lands at LDS byte id*16 of the tile, i.e. row id/16,
  //      slot id%16; it must carry chunk (slot ^ swizzle(row)) of that key row
  const int tiles_all = (p.Lk + KVBLK - 1) / KVBLK;
  const int t0 = split * tps;                                      // first key tile of this split (0 without split-KV)
  const int ntiles = min(tps, tiles_all - t0);       // tiles are indexed locally below; t0 + t is the global tile
  const unsigned k_step = (unsigned)(KVBLK * p.k_rs * 2), v_step = (unsigned)(KVBLK * p.v_rs * 2);
  unsigned k_go[2], v_go[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = (tid + NT * i) >> 4;
    const int ch = (tid & 15) ^ (((row & 3) << 2) | ((row >> 2) & 3));
    k_go[i] = (unsigned)(row * p.k_rs * 2 + ch * 16);
    v_go[i] = (unsigned)(row * p.v_rs * 2 + ch * 16);
  }
  const unsigned lds0 = (unsigned)(size_t)LDS_PTR(smem);
  auto issue_tile = [&](int t, int parts = 3, int tslot = -1) {      // parts: 1 = K, 2 = V; tslot: ring position (default t)
    const unsigned slot = lds0 + (unsigned)(((tslot < 0 ? t : tslot) & (NSLOT - 1)) * KV_TILE_BYTES + wave * 1024);   // wave-uniform; K ring, V ring = + V_RING
    const int tg = t0 + t;
    if (FULL || (tg + 1) * KVBLK <= p.Lk) {
      // FULL: the last tile is the window [Lk - 64, Lk) (scalar select, no branch)
      const unsigned row0 = FULL && tg == tiles_all - 1 ? (unsigned)(p.Lk - KVBLK) : (unsigned)tg * KVBLK;
      const char* kt = kbase + (size_t)(row0 * (unsigned)(p.k_rs * 2));     // wave-uniform tile bases
      const char* vt = vbase + (size_t)(row0 * (unsigned)(p.v_rs * 2));
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        if (parts & 1) lds_dma16_sbase(kt, k_go[i], slot + i * 8192);
        if (parts & 2) lds_dma16_sbase(vt, v_go[i], slot + V_RING + i * 8192);
      }
    } else {                                 // last, partial tile: rows past Lk re-read the last key (masked later)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = (tid + NT * i) >> 4;
        const int col = ((tid & 15) ^ (((row & 3) << 2) | ((row >> 2) & 3))) * 16;
        const int key = min(tg * KVBLK + row, p.Lk - 1);
        if (parts & 1) lds_dma16(kbase + ((int64_t)key * p.k_rs * 2 + col), slot + i * 8192);
        if (parts & 2) lds_dma16(vbase + ((int64_t)key * p.v_rs * 2 + col), slot + V_RING + i * 8192);
      }
    }
  };
  // Half tile g lives in ring slot (g>>1)&3; the K slots fill the first 64 KiB of LDS and the V slots the second, so ONE set of
  // per-lane fragment addresses per ring reaches all four slots through the 16-bit ds_read immediate (slot, half and 16-key step
  // are compile-time constants once the tile loop is unrolled by 4).
  const char* kaddr[8];
  const char* vaddr[2][4];
#pragma unroll
  for (int ds = 0; ds < 8; ++ds) kaddr[ds] = smem + koff[ds];
#pragma unroll
  for (int half = 0; half < 2; ++half)
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) vaddr[half][dt] = smem + V_RING + voff[half][dt];
  // ---- the work units of this workgroup (one, unless SHORT)
  for (int u = unit; u < unit_end; ++u) {
  const int qbi = u % p.q_blocks, bh = u / p.q_blocks;
  head = bh % p.H; b = bh / p.H;
  qbase = p.q + (int64_t)b * p.q_bs + head * HD;
  kbase = (const char*)(p.k + (int64_t)b * p.k_bs + head * HD);
  vbase = (const char*)(p.v + (int64_t)b * p.v_bs + head * HD);
  // ---- Q fragment (B operand of S^T = K.Q^T): lane (r, h) holds Q[q0 + r][16*ds + 8h .. +7]
  const int q0 = qbi * QBLK + wave * 32;
  const int qrow = min(q0 + r, p.Lq - 1);
  bf16x8 qf[8];
#pragma unroll
  for (int ds = 0; ds < 8; ++ds) qf[ds] = *(const bf16x8*)(qbase + (int64_t)qrow * p.q_rs + ds * 16 + h * 8);

  // PRE: row reference of the online softmax in exp2 units (a lane holds 32 scores of ONE query row, so it is one value per lane,
  // kept 16 times as the C operand that starts every S^T chain).  It is NOT the running maximum: it is only raised when a score
  // exceeds it by more than 2^THR (deferred rescale).
  f32x16 negref;
#pragma unroll
  for (int e = 0; e < 16; ++e) negref[e] = 0.f;
  float ref = 0.f;
  // ds0 == 0 starts a new accumulation: from a literal-zero C operand (no register zeroing), or from -ref (PRE)
  // `ghalf` (half-tile index modulo 8) must be a compile-time constant at every call site
  // K fragments ds0 .. ds0+3 of half `ghalf` (a compile-time constant at every call site) -> registers
  auto qk_read = [&](auto ghalf_c, auto ds0_c, bf16x8 (&kf)[4]) {
    constexpr int ghalf = decltype(ghalf_c)::value, ds0 = decltype(ds0_c)::value;
    constexpr int slot = (ghalf >> 1) & (NSLOT - 1);
    constexpr int imm = slot * KV_TILE_BYTES + (ghalf & 1) * 8192;
#ifdef A32_HALF_FRAG_ABLATE     // TIMING ABLATION (WRONG results): half the K / V fragment reads per MFMA, everything else as it is -- the ceiling of what
#pragma unroll                  // a tiling with 64 query rows per wave (each fragment feeding two MFMAs) could win at THIS occupancy
    for (int i = 0; i < 2; ++i) kf[i] = kf[2 + i] = *(const bf16x8*)(kaddr[ds0 + i] + imm);
#else
#pragma unroll
    for (int i = 0; i < 4; ++i) kf[i] = *(const bf16x8*)(kaddr[ds0 + i] + imm);
#endif
  };
  // ... and their 4 MFMAs of the S^T chain
  auto qk_mma = [&](auto ds0_c, const bf16x8 (&kf)[4], f32x16& sacc) {
    constexpr int ds0 = decltype(ds0_c)::value;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int ds = ds0 + i;
      if (ds == 0) {
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if constexpr (PRE) sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[i], qf[ds], negref, 0, 0, 0);   // S' = s - ref
        else sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[i], qf[ds], zero, 0, 0, 0);
      } else {
        sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[i], qf[ds], sacc, 0, 0, 0);
      }
    }
  };
  auto qk_part = [&](auto ghalf_c, auto ds0_c, f32x16& sacc) {      // fragments first, then the MFMAs
    bf16x8 kf[4];
    qk_read(ghalf_c, ds0_c, kf);
    qk_mma(ds0_c, kf, sacc);
  };
  bf16x8 kf_pre[4];      // the first four K fragments of the NEXT half tile, read one step ahead (block A starts on the matrix pipe)
  // weight of the LAST key as a score bias (multiplicity N of a key = + log2 N on its score): raw-score units in the non-PRE form
  const float last_bias = PRE ? p.last_key_bias : p.last_key_bias / p.scale_log2e;
  // FULL: keys of the shifted last tile that the tile before it already held -> -inf.  Branch-free; called in the tail tiles only.
  const int last_shift = tiles_all * KVBLK - p.Lk;      // 0..63: first valid in-tile position of the global last tile
  auto mask_shift = [&](int g, f32x16& sacc) {
    const int lim = (t0 + (g >> 1) == tiles_all - 1 ? last_shift : 0) - 32 * (g & 1) - 4 * h;      // element e is valid iff (e & 3) + 8 (e >> 2) >= lim
#pragma unroll
    for (int e = 0; e < 16; ++e) sacc[e] = (e & 3) + 8 * (e >> 2) >= lim ? sacc[e] : -INFINITY;
  };
  auto mask_half = [&](int g, f32x16& sacc) {      // g: local half index; keys are global
    if constexpr (FULL) return;
    const int gg = 2 * t0 + g;
    // past the end of the keys, or of this split's range -- or the half tile that holds a weighted last key
    if ((gg + 1) * 32 > p.Lk || g >= 2 * ntiles || (p.last_key_bias != 0.f && (gg + 1) * 32 == p.Lk)) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int key = gg * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (key >= p.Lk || g >= 2 * ntiles) sacc[e] = -INFINITY;
        else if (key == p.Lk - 1) sacc[e] += last_bias;
      }
    }
  };
  f32x16 o_acc[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int e = 0; e < 16; ++e) o_acc[dt][e] = 0.f;
  bf16x8 pf_prev[2];       // packed P of the previous half, one fragment per 16-key step
  auto pv_half = [&](auto ghalf_c) {
    constexpr int ghalf = decltype(ghalf_c)::value;
    constexpr int slot = (ghalf >> 1) & (NSLOT - 1);
    constexpr int imm = slot * KV_TILE_BYTES + (ghalf & 1) * 8192;
#pragma unroll
    for (int ss = 0; ss < 2; ++ss) {
      bf16x8 vf[4];
#pragma unroll
#ifdef A32_HALF_FRAG_ABLATE
      for (int dt = 0; dt < 2; ++dt) {
#else
      for (int dt = 0; dt < 4; ++dt) {                                                   // 8 transposed reads first ...
#endif
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vaddr[0][dt] + imm + ss * 4096));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vaddr[1][dt] + imm + ss * 4096));
        const bf16x4 lo_b = __builtin_bit_cast(bf16x4, lo), hi_b = __builtin_bit_cast(bf16x4, hi);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          vf[dt][e] = lo_b[e];
          vf[dt][4 + e] = hi_b[e];
        }
      }
#ifdef A32_HALF_FRAG_ABLATE
      vf[2] = vf[0]; vf[3] = vf[1];
#endif
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)                                                     // ... then 4 MFMAs
        o_acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[dt], pf_prev[ss], o_acc[dt], 0, 0, 0);
    }
  };

  float m_run = -INFINITY, l_run = 0.f;
  const float c = p.scale_log2e;

  // ---- prologue: tiles 0 and 1 on their way, S(half 0) computed; zero the V half that PV(-1) multiplies by P = 0
  if constexpr (SHORT) {
    if (bh != bh_loaded) {                 // first unit of a (batch, head): every tile of it into its slot, once
      if (bh_loaded >= 0) __builtin_amdgcn_s_barrier();      // every wave is done with the tiles of the head before
      bh_loaded = bh;
      if (ntiles < NSLOT) *(u32x4*)(smem + V_RING + (NSLOT - 1) * KV_TILE_BYTES + 8192 + tid * 16) = (u32x4){0u, 0u, 0u, 0u};   // (4 tiles: slot 3 holds finite data)
      for (int tt = 0; tt < ntiles; ++tt) issue_tile(tt);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  } else {
    *(u32x4*)(smem + V_RING + (NSLOT - 1) * KV_TILE_BYTES + 8192 + tid * 16) = (u32x4){0u, 0u, 0u, 0u};
    issue_tile(0);
    if (ntiles > 1) issue_tile(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  __builtin_amdgcn_sched_barrier(0);
  f32x16 s_a, s_b;         // scores of the current / next half, ping-ponged statically (no register copies)
  qk_part(IC<0>{}, IC<0>{}, s_a);
  qk_part(IC<0>{}, IC<4>{}, s_a);
  qk_read(IC<1>{}, IC<0>{}, kf_pre);
  constexpr bool FIRST_IN_PROLOGUE = PRE && FULL;
  if constexpr (FIRST_IN_PROLOGUE) {
    // the first reference = the row maximum of half tile 0, set HERE: as a `g == 0` case of the step's rescale test it costs a scalar
    // test and ~50 register copies (phi nodes of the loop-carried state) per loop iteration on the common path.  (FULL only: the
    // general instance has no register to spare for it.)
    float mx = s_a[0];
#pragma unroll
    for (int e = 1; e < 16; ++e) mx = fmaxf(mx, s_a[e]);
    mx = pair_max(mx);
    ref = mx;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      s_a[e] -= mx;
      negref[e] = -mx;
    }
  }
#pragma unroll
  for (int ss = 0; ss < 2; ++ss)
#pragma unroll
    for (int e = 0; e < 8; ++e) pf_prev[ss][e] = (bf16)0.f;

  // one half-tile step g, the same straight-line shape for every g:
  //  * S(g+1) is always computed; past the last half it reads stale LDS, mask_half turns it into -inf and the
  //    step that consumes it adds exactly 0 (exp2(-inf) = 0, no rescale);
  //  * PV(g-1) is always issued; for g = 0 P is zero and the V half it reads (slot 3, second half) was zeroed.
  // A half-tile step g in two parts.  Part A: first half of S(g+1) on the matrix pipe (fragments read during the previous step) |
  // row maximum of S(g) on the VALU, and the (rare) deferred rescale.  Part B: second half of S(g+1) and PV(g-1) on the matrix pipe |
  // exp2 / row sum / pack of S(g) on the VALU.
  // DEF scores of every half are exponentiated one part later -- in part A of the NEXT step, whose 4 MFMAs otherwise run beside a
  // dependent chain of 9 maxima only -- instead of in part B beside 12 MFMAs, 16 exp2, 16 adds and 8 packs: P(g) is not needed
  // before PV(g) in part B of step g+1.  They wait in s_pend in the units of the reference of their own step; finish_pending runs
  // BEFORE the next step's rescale decision, so a rescale finds them where it expects them: in l_run and in the packed pf_prev.
  // With a key-mask branch in every step (r4g) 4 of 16 paid (-0.6 ... -0.9 % of a step); since the step is one basic block (FULL) none
  // is best: 0 / 1 / 2 / 4 / 6 / 8 deferred = -0.3 ... -0.8 / +0.3 / -0.4 / 0 / +0.7 / +0.5 % (profiles/r4ab_*).  Default 0.
  constexpr int DEF = PRE ? A32_DEFER : 0;
  float s_pend[DEF > 0 ? DEF : 1];
#pragma unroll
  for (int i = 0; i < (DEF > 0 ? DEF : 1); ++i) s_pend[i] = -INFINITY;
  auto finish_pending = [&]() __attribute__((always_inline)) {
    if constexpr (DEF > 0) {
      float ps = 0.f;
#pragma unroll
      for (int i = 0; i < DEF; ++i) {
        const float pv = __builtin_amdgcn_exp2f(s_pend[i]);
        ps = i == 0 ? pv : ps + pv;
        pf_prev[1][8 - DEF + i] = f2bf(pv);
      }
      l_run += ps;
    }
  };
  auto stepA = [&](int g, f32x16& s_cur, f32x16& s_nxt, auto mask_c) __attribute__((always_inline)) {
    if constexpr (FULL && decltype(mask_c)::value) mask_shift(g, s_cur);
    mask_half(g, s_cur);             // keys past Lk -> -inf (uniform branch, only taken in the last tile)
    qk_mma(IC<0>{}, kf_pre, s_nxt);
    finish_pending();
    const float m0 = vmax8(s_cur[0], s_cur[1], s_cur[2], s_cur[3], s_cur[4], s_cur[5], s_cur[6], s_cur[7]);
    const float m1 = vmax8(s_cur[8], s_cur[9], s_cur[10], s_cur[11], s_cur[12], s_cur[13], s_cur[14], s_cur[15]);
    // the test needs no row maximum: any(lane maximum > THR) over the wave is any(row maximum > THR); the exchange between the
    // two lanes of a row happens in the rare branch only (-0.7 % of a step, profiles/r4e_*)
#ifdef A32_NOMAX_ABLATE      // TIMING ABLATION (WRONG results when a rescale would have been needed): what the 9 maxima of a half tile cost
    const float mloc = s_cur[0];      // one score instead of the lane maximum: the test and its branch stay, the 9 maxima go
#else
    const float mloc = vmax(m0, m1);
#endif
    if constexpr (PRE) {
      // ---- deferred rescale (always taken for half 0, which sets the reference to the first row maximum): O, l, the pending
      // P(g-1), the scores of this half and the already started chain of the next half all move to the new reference
      const bool first = !FIRST_IN_PROLOGUE && g == 0;       // FULL: the first reference is set in the prologue
      if (first || __any(mloc > RESCALE_THR_LOG2)) {            // relative to ref
        const float mx = pair_max(mloc);
        const float delta = first ? mx : fmaxf(mx, 0.f);
        const float alpha = first ? 1.f : __builtin_amdgcn_exp2f(-delta);
        ref += delta;
        l_run *= alpha;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
          for (int e = 0; e < 16; ++e) o_acc[dt][e] *= alpha;
#pragma unroll
        for (int ss = 0; ss < 2; ++ss)
#pragma unroll
          for (int e = 0; e < 8; ++e) pf_prev[ss][e] = f2bf(bf2f(pf_prev[ss][e]) * alpha);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          s_cur[e] -= delta;
          s_nxt[e] -= delta;
          negref[e] = -ref;
        }
      }
    } else {
      // ---- deferred rescale (always taken for half 0): O, l AND the pending P(g-1) move to the new max
      if (__any((mloc - m_run) * c > RESCALE_THR_LOG2)) {
        const float mx = pair_max(mloc);
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
        l_run *= alpha;
        m_run = m_new;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
          for (int e = 0; e < 16; ++e) o_acc[dt][e] *= alpha;
#pragma unroll
        for (int ss = 0; ss < 2; ++ss)
#pragma unroll
          for (int e = 0; e < 8; ++e) pf_prev[ss][e] = f2bf(bf2f(pf_prev[ss][e]) * alpha);
      }
    }
  };
  auto stepB = [&](auto g8_c, f32x16& s_cur, f32x16& s_nxt) __attribute__((always_inline)) {     // g8 = g mod 8 as a compile-time constant
    constexpr int g8 = decltype(g8_c)::value;
    float psum = 0.f;
    bf16x8 pn[2];
    qk_part(IC<((g8 + 1) & 7)>{}, IC<4>{}, s_nxt);      // (its four K fragments read at the end of part A instead: -0.2 %, profiles/r4g)
    pv_half(IC<((g8 + 7) & 7)>{});
    qk_read(IC<((g8 + 2) & 7)>{}, IC<0>{}, kf_pre);          // for the next step's part A
    const float mc = PRE ? 0.f : m_run * c;
#pragma unroll
    for (int e = 0; e < 16 - DEF; ++e) {
      const float pv = PRE ? __builtin_amdgcn_exp2f(s_cur[e]) : __builtin_amdgcn_exp2f(__builtin_fmaf(s_cur[e], c, -mc));
      psum = e == 0 ? pv : psum + pv;                       // (0 + x is an instruction: x may be -0 as far as the compiler knows)
      pn[e >> 3][e & 7] = f2bf(pv);
    }
#pragma unroll
    for (int i = 0; i < DEF; ++i) s_pend[i] = s_cur[16 - DEF + i];
    l_run += psum;
    pf_prev[0] = pn[0];            // PV(g-1) above consumed the old value (program order)
#pragma unroll
    for (int e = 0; e < 8 - DEF; ++e) pf_prev[1][e] = pn[1][e];      // the other DEF elements: finish_pending
    // pin the softmax results here: hipcc otherwise sinks the whole exp chain below the next step's branch
    asm volatile("" : "+v"(pf_prev[0]), "+v"(pf_prev[1]), "+v"(l_run));
    // Block B schedule: hipcc otherwise emits the 12 MFMAs first and the exp chain after them, leaving the matrix
    // pipe idle during the softmax.  Interleave: per MFMA two LDS reads (20 in the block) and ~5 VALU ops.
    // (2 / 4 / 6 more LDS reads in front of the block's first MFMA: +0.1 / -0.7 / -1.1 %, profiles/r4g)
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // DS read
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
#ifdef A32_VALU
      __builtin_amdgcn_sched_group_barrier(0x002, A32_VALU, 0);
#else
      __builtin_amdgcn_sched_group_barrier(0x002, PRE ? (DEF >= 4 ? 3 : 4) : 5, 0);   // VALU (exp2 / fma / add / cvt)
#endif
    }
  };

  // Measured and not kept (profiles/r4e_attn_stagger_and_local_max.txt): waves 4-7 half a step behind their SIMD partners (their
  // barrier in front of part B instead of part A).  As two copies of the loop it does not fit the instruction cache (-10 %), as one
  // copy with two conditional barrier sites -1.4 % alone and +1.1 % on the step.
  auto tile = [&](int t, auto t4_c, auto mask_c) __attribute__((always_inline)) {       // t4 = t mod 4 as a compile-time constant
    constexpr int t4 = decltype(t4_c)::value;
    auto top = [&](int tt) __attribute__((always_inline)) {
      if constexpr (SHORT) return;         // every tile is resident
      if (FULL || tt > 0) {
        // top of 64-key tile tt: tile tt+1 (issued one tile ago) has landed and becomes visible; the slot of tile
        // tt-2 (last read by the PV of its second half) is free again
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        ATTN_STAMP_BARRIER(__builtin_amdgcn_s_barrier());
        __builtin_amdgcn_sched_barrier(0);
      }
      // K pieces of tile tt+2 now, its V pieces half a tile later: two short bursts of LDS-DMA issue per tile instead
      // of one of four per wave right after the barrier (-1.2...-1.6 % time; placements inside the steps' MFMA blocks were slower)
      if constexpr (FULL) issue_tile(min(tt + 2, ntiles - 1), 1, tt + 2);
      else if (tt + 2 < ntiles) issue_tile(tt + 2, 1);
    };
    // (every wave with its barrier between parts A and B of the odd step instead: +1.1 % alone, -0.7 % on the step, not additive
    //  with the deferred scores: profiles/r4g_*)
    top(t);
    stepA(2 * t, s_a, s_b, mask_c);
    stepB(IC<2 * t4>{}, s_a, s_b);
    if constexpr (SHORT) {
    } else if constexpr (FULL) issue_tile(min(t + 2, ntiles - 1), 2, t + 2);
    else if (t + 2 < ntiles) issue_tile(t + 2, 2);
    stepA(2 * t + 1, s_b, s_a, mask_c);
    stepB(IC<2 * t4 + 1>{}, s_b, s_a);
  };
  ATTN_STAMP_BEGIN();
  using MaskOn = std::integral_constant<bool, true>;
  using MaskOff = std::integral_constant<bool, false>;
  int t = 0;
  // FULL: the main loop leaves 1..4 tail tiles (the only ones that may hold the shifted last tile); otherwise 0..3
  for (; t + 4 <= ntiles - (FULL ? 1 : 0); t += 4) {
    tile(t, IC<0>{}, MaskOff{});
    tile(t + 1, IC<1>{}, MaskOff{});
    tile(t + 2, IC<2>{}, MaskOff{});
    tile(t + 3, IC<3>{}, MaskOff{});
  }
  const int rem = ntiles - t;               // tail tiles, same code with static slots
  if (rem > 0) tile(t, IC<0>{}, MaskOn{});
  if (rem > 1) tile(t + 1, IC<1>{}, MaskOn{});
  if (rem > 2) tile(t + 2, IC<2>{}, MaskOn{});
  if (rem > 3) tile(t + 3, IC<3>{}, MaskOn{});
  // pending PV of the last half (2*ntiles - 1): its slot is (ntiles - 1) & 3
  finish_pending();
  switch ((ntiles - 1) & 3) {
    case 0: pv_half(IC<1>{}); break;
    case 1: pv_half(IC<3>{}); break;
    case 2: pv_half(IC<5>{}); break;
    default: pv_half(IC<7>{}); break;
  }
  ATTN_STAMP_END();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // ---- epilogue: O[q][32dt + 8i + 4h + (0..3)] = o_acc[dt][4i + (0..3)] / l
  const float l_tot = pair_sum(l_run);
  const int qi = q0 + r;
  if (partial) {                           // partial result of this key range; attn_merge_kernel finishes the softmax
    if (qi < p.Lq) {
      const int64_t row = ((int64_t)(p.slot0 + split) * p.n_units + ul) * QBLK + wave * 32 + r;
      float* orow = p.ws_o + row * HD;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          f32x4 ov;
#pragma unroll
          for (int e = 0; e < 4; ++e) ov[e] = o_acc[dt][4 * i + e];
          *(f32x4*)(orow + 32 * dt + 8 * i + 4 * h) = ov;
        }
      if (h == 0) *(f32x2*)(p.ws_ml + row * 2) = (f32x2){PRE ? ref : m_run, l_tot};      // PRE: reference in exp2 units
    }
    return;
  }
  // A lane holds 4 of the 8 columns of each 8-column group of its row, its partner (lane ^ 32) the other 4.  One half exchange
  // per dword between the groups of a pair (lower lanes give their part of the odd group, upper lanes their part of the even
  // one) leaves 16 contiguous bytes in every lane: 8 stores of 16 B instead of 16 of 8 B (the tail is store-issue bound).
  const float inv_l = 1.0f / l_tot;
  bf16* orow = p.o + (int64_t)b * p.o_bs + (int64_t)min(qi, p.Lq - 1) * p.o_rs + head * HD + 8 * h;
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int i = 0; i < 4; i += 2) {
      bf16x4 even, odd;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        even[e] = f2bf(o_acc[dt][4 * i + e] * inv_l);
        odd[e] = f2bf(o_acc[dt][4 * i + 4 + e] * inv_l);
      }
      const u32x2 ev = __builtin_bit_cast(u32x2, even), od = __builtin_bit_cast(u32x2, odd);
      unsigned a0 = ev[0], a1 = ev[1], c0 = od[0], c1 = od[1];
      half_swap_u32(a0, c0);
      half_swap_u32(a1, c1);
      if (qi < p.Lq) *(u32x4*)(orow + 32 * dt + 8 * i) = (u32x4){a0, a1, c0, c1};
    }
  }   // q blocks of this workgroup
}

#ifdef FLEXAM_ATTN_BODY16
#include "attn_body16.inc"
#endif
#include "attn_fp8.inc"

// out[b][q][head][:] = sum_s w_s O_s / sum_s w_s l_s with w_s = exp2((m_s - max_s m_s) * scale_log2e) for the rows of the
// launch's units; one wave per row, two columns per lane
__global__ __launch_bounds__(256) void attn_merge_kernel(AttnParams p) {
  const int64_t rows = (int64_t)p.n_units * QBLK;
  const int lane = threadIdx.x & 63;
  for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += (int64_t)gridDim.x * 4) {
    const int ul = (int)(row / QBLK), rr = (int)(row % QBLK);
    const int unit = p.unit0 + ul;
    const int qb = unit % p.q_blocks, bh = unit / p.q_blocks;
    const int q = qb * QBLK + rr;
    if (q >= p.Lq) continue;
    float m = -INFINITY;
    for (int s = 0; s < p.n_slots; ++s) m = fmaxf(m, p.ws_ml[(s * rows + row) * 2]);
    float l = 0.f;
    f32x2 acc = {0.f, 0.f};
    for (int s = 0; s < p.n_slots; ++s) {
      const f32x2 ml = *(const f32x2*)(p.ws_ml + (s * rows + row) * 2);
      const float w = __builtin_amdgcn_exp2f((ml[0] - m) * (p.prescaled ? 1.0f : p.scale_log2e));
      l += w * ml[1];
      const f32x2 o = *(const f32x2*)(p.ws_o + (s * rows + row) * HD + 2 * lane);
      acc += o * w;
    }
    const int head = bh % p.H, b = bh / p.H;
    const float inv = 1.0f / l;
    bf16x2 ov;
    ov[0] = f2bf(acc[0] * inv);
    ov[1] = f2bf(acc[1] * inv);
    *(bf16x2*)(p.o + (int64_t)b * p.o_bs + (int64_t)q * p.o_rs + head * HD + 2 * lane) = ov;
  }
}

int attn_run(const void* q, int64_t q_bs, int64_t q_rs, const void* k, int64_t k_bs, int64_t k_rs, const void* v, int64_t v_bs,
             int64_t v_rs, void* o, int64_t o_bs, int64_t o_rs, int B, int H, int Lq, int Lk, int head_dim, float softmax_scale,
             int kv_splits, int split_from_unit, float* ws_o, float* ws_ml, void* stream, int partial_slot0 = -1, float last_key_bias = 0.f,
             const void* q8 = nullptr, const void* qs = nullptr, const void* kv8 = nullptr, int kv8_chunk_tiles = 0) {
  FX_REQUIRE(((q && k && v) || (q8 && qs && kv8)) && (o || partial_slot0 >= 0), FLEXAM_E_ARG, "attn_fwd: null pointer");
  FX_REQUIRE(head_dim == HD, FLEXAM_E_SHAPE, "attn_fwd: head_dim %d unsupported (128 only)", head_dim);
  FX_REQUIRE(B > 0 && H > 0 && Lq > 0 && Lk > 0, FLEXAM_E_SHAPE, "attn_fwd: empty problem B=%d H=%d Lq=%d Lk=%d", B, H, Lq, Lk);
  FX_REQUIRE(q_rs % 8 == 0 && k_rs % 8 == 0 && v_rs % 8 == 0 && o_rs % 8 == 0 && q_bs % 8 == 0 && k_bs % 8 == 0 && v_bs % 8 == 0 &&
                 o_bs % 8 == 0,
             FLEXAM_E_SHAPE, "attn_fwd: strides must keep 16-byte alignment of head rows");
  FX_REQUIRE(((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o) % 16 == 0, FLEXAM_E_ARG, "attn_fwd: misaligned pointer");
  const int tiles_all = (Lk + KVBLK - 1) / KVBLK;
  FX_REQUIRE(kv_splits >= 1 && kv_splits <= tiles_all, FLEXAM_E_ARG, "attn_fwd: %d key splits for %d key tiles", kv_splits, tiles_all);
  FX_REQUIRE((kv_splits == 1 && partial_slot0 < 0) || (ws_o && ws_ml), FLEXAM_E_ARG, "attn_fwd: split-KV needs both workspaces");
  AttnParams p;
  p.q = (const bf16*)q; p.k = (const bf16*)k; p.v = (const bf16*)v; p.o = (bf16*)o;
  p.q_bs = q_bs; p.q_rs = q_rs; p.k_bs = k_bs; p.k_rs = k_rs; p.v_bs = v_bs; p.v_rs = v_rs; p.o_bs = o_bs; p.o_rs = o_rs;
  p.B = B; p.H = H; p.Lq = Lq; p.Lk = Lk;
  p.prescaled = softmax_scale < 0.f;                   // FLEXAM_ATTN_PRESCALED: q already carries softmax_scale * log2(e)
  p.scale_log2e = p.prescaled ? 1.0f : softmax_scale * 1.4426950408889634f;
  p.q_blocks = (Lq + QBLK - 1) / QBLK;
  p.tiles_per_split = (tiles_all + kv_splits - 1) / kv_splits;
  p.kv_splits = (tiles_all + p.tiles_per_split - 1) / p.tiles_per_split;     // drop empty trailing splits
  p.ws_o = ws_o; p.ws_ml = ws_ml;
  p.partial = 0; p.slot0 = 0; p.n_slots = p.kv_splits; p.whole_units = 0;
  p.last_key_bias = last_key_bias;
  p.q8 = (const unsigned char*)q8; p.qs = (const unsigned*)qs; p.kv8 = (const unsigned char*)kv8;
  p.lq_pad = p.q_blocks * QBLK;
  p.kv8_chunk_tiles = kv8_chunk_tiles > 0 ? kv8_chunk_tiles : tiles_all;
  FX_REQUIRE(tiles_all < 65536, FLEXAM_E_SHAPE, "attn_fwd: %d key tiles (at most 65535)", tiles_all);
  p.kv8_chunk_magic = p.kv8_chunk_tiles > 1 ? (unsigned)(((1ull << 32) + (unsigned)p.kv8_chunk_tiles - 1) / (unsigned)p.kv8_chunk_tiles) : 0u;
  FX_REQUIRE(q8 || (int64_t)Lk * k_rs * 2 < (1ll << 31) && (int64_t)Lk * v_rs * 2 < (1ll << 31), FLEXAM_E_SHAPE,
             "attn_fwd: one (batch, head) K/V panel must span < 2 GiB (32-bit tile offsets)");
  const int smem = q8 ? NSLOT * REC_BYTES : NSLOT * 2 * KV_TILE_BYTES;   // ring of 4 K tiles, then ring of 4 V tiles: 128 KiB (fp8: 4 records, 68 KiB)
  const bool cross = Lk <= 1024;        // separate symbol for the short-context (text) launches
  auto kern = cross ? (p.prescaled ? attn_fwd_kernel<1, true> : attn_fwd_kernel<1, false>)
                    : (p.prescaled ? attn_fwd_kernel<0, true> : attn_fwd_kernel<0, false>);
  bool body16 = false;
#ifdef FLEXAM_ATTN_BODY16                              // diagnostic builds only (attn_body16.inc): the 16x16x32 body, selected per call
  const char* be = getenv("FLEXAM_ATTN_BODY");
  body16 = be && atoi(be) == 16;
  if (body16) kern = cross ? (p.prescaled ? attn_fwd16_kernel<1, true> : attn_fwd16_kernel<1, false>)
                           : (p.prescaled ? attn_fwd16_kernel<0, true> : attn_fwd16_kernel<0, false>);
#endif
  // the text cross-attention (at most 4 key tiles, pre-scaled q, one launch without key splits): K/V resident, several q blocks per workgroup
  const char* se_ = getenv("FLEXAM_ATTN_SHORT");       // read per call (A/B in one process); 0 = one workgroup per q block
  const bool short_ctx = !q8 && cross && p.prescaled && tiles_all <= NSLOT && kv_splits == 1 && partial_slot0 < 0 && !body16 && !(se_ && atoi(se_) == 0);
  if (short_ctx) kern = attn_fwd_kernel<1, true, false, true>;
  const char* fe_ = getenv("FLEXAM_ATTN_FULL");        // read per call (A/B in one process); 0 = the general instance
  const bool full = !q8 && !cross && p.prescaled && Lk >= KVBLK && last_key_bias == 0.f && !body16 && !(fe_ && atoi(fe_) == 0);
  if (full) kern = attn_fwd_kernel<0, true, true>;
  if (q8) kern = attn8_fwd_kernel<0>;
  static bool attr_set[FLEXAM_MAX_DEVICES][11] = {};      // per device and kernel instance
  const int which = q8 ? 8 : full ? 9 : short_ctx ? 10 : (cross ? 1 : 0) + (p.prescaled ? 2 : 0) + (body16 ? 4 : 0);
  const int dev = flexam_current_device();
  if (!attr_set[dev][which]) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
      return flexam_fail(FLEXAM_E_LAUNCH, "attn_fwd: cannot raise dynamic LDS to %d bytes", smem);
    attr_set[dev][which] = true;
  }
  const int units = B * H * p.q_blocks;
  FX_REQUIRE(split_from_unit >= 0 && split_from_unit <= units, FLEXAM_E_ARG, "attn_fwd: split_from_unit %d of %d units", split_from_unit, units);
  const int S = p.kv_splits, tps = p.tiles_per_split;
  if (partial_slot0 >= 0) {                // every unit, this call's keys only: partials into slots slot0 .. slot0 + S - 1, no merge
    FX_REQUIRE(S == kv_splits, FLEXAM_E_ARG, "attn_fwd_partial: %d key ranges requested but %d key tiles give %d (use ceil(tiles / ceil(tiles / S)))",
               kv_splits, tiles_all, S);
    p.unit0 = 0; p.n_units = units; p.kv_splits = S; p.tiles_per_split = tps; p.partial = 1; p.slot0 = partial_slot0;
    hipLaunchKernelGGL(kern, dim3(p.n_units * S), dim3(NT), smem, (hipStream_t)stream, p);
    return flexam_check_launch("flexam_attn_fwd_partial");
  }
  if (S == 1) split_from_unit = units;
  if (short_ctx) {                         // one persistent workgroup per CU, each with a contiguous share of the units
    p.unit0 = 0; p.n_units = units; p.kv_splits = 1; p.tiles_per_split = tiles_all;
    const int cus = flexam_num_cus();
    hipLaunchKernelGGL(kern, dim3(units < cus ? units : cus), dim3(NT), smem, (hipStream_t)stream, p);
    return flexam_check_launch("flexam_attn_fwd");
  }
  const char* fe = getenv("FLEXAM_ATTN_FUSED_TAIL");     // read per call (A/B in one process); 0 = the two-launch form
  if (split_from_unit > 0 && split_from_unit < units && (!fe || atoi(fe) != 0)) {
    // one launch: whole units and the split tail side by side on every XCD (see AttnParams::whole_units), then the merge
    p.whole_units = split_from_unit;
    p.unit0 = split_from_unit; p.n_units = units - split_from_unit; p.kv_splits = S; p.tiles_per_split = tps; p.partial = 1; p.n_slots = S;
    const int grid = 8 * ((p.whole_units + 7) / 8 + (p.n_units * S + 7) / 8);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), smem, (hipStream_t)stream, p);
    const int64_t g = ((int64_t)p.n_units * QBLK + 3) / 4;
    hipLaunchKernelGGL(attn_merge_kernel, dim3((unsigned)(g > 16384 ? 16384 : g)), dim3(256), 0, (hipStream_t)stream, p);
    return flexam_check_launch("flexam_attn_fwd");
  }
  if (split_from_unit > 0) {               // units [0, split_from_unit): one pass over all keys
    p.unit0 = 0; p.n_units = split_from_unit; p.kv_splits = 1; p.tiles_per_split = tiles_all;
    hipLaunchKernelGGL(kern, dim3(p.n_units), dim3(NT), smem, (hipStream_t)stream, p);
  }
  if (split_from_unit < units) {           // the rest: S key ranges each, then the merge
    p.unit0 = split_from_unit; p.n_units = units - split_from_unit; p.kv_splits = S; p.tiles_per_split = tps; p.partial = 1; p.n_slots = S;
    hipLaunchKernelGGL(kern, dim3(p.n_units * S), dim3(NT), smem, (hipStream_t)stream, p);
    const int64_t g = ((int64_t)p.n_units * QBLK + 3) / 4;
    hipLaunchKernelGGL(attn_merge_kernel, dim3((unsigned)(g > 16384 ? 16384 : g)), dim3(256), 0, (hipStream_t)stream, p);
  }
  return flexam_check_launch("flexam_attn_fwd");
}

}  // namespace

extern "C" int flexam_attn_fwd(const void* q, int64_t q_bs, int64_t q_rs, const void* k, int64_t k_bs, int64_t k_rs,
                               const void* v, int64_t v_bs, int64_t v_rs, void* o, int64_t o_bs, int64_t o_rs, int B, int H,
                               int Lq, int Lk, int head_dim, float softmax_scale, void* stream) {
  return attn_run(q, q_bs, q_rs, k, k_bs, k_rs, v, v_bs, v_rs, o, o_bs, o_rs, B, H, Lq, Lk, head_dim, softmax_scale, 1, 0, nullptr,
                  nullptr, stream);
}

extern "C" int flexam_attn_fwd_lastkey(const void* q, int64_t q_bs, int64_t q_rs, const void* k, int64_t k_bs, int64_t k_rs,
                                       const void* v, int64_t v_bs, int64_t v_rs, void* o, int64_t o_bs, int64_t o_rs, int B, int H,
                                       int Lq, int Lk, int head_dim, float softmax_scale, float last_key_multiplicity, void* stream) {
  FX_REQUIRE(last_key_multiplicity >= 1.0f, FLEXAM_E_ARG, "attn_fwd_lastkey: multiplicity %g < 1", (double)last_key_multiplicity);
  return attn_run(q, q_bs, q_rs, k, k_bs, k_rs, v, v_bs, v_rs, o, o_bs, o_rs, B, H, Lq, Lk, head_dim, softmax_scale, 1, 0, nullptr,
                  nullptr, stream, -1, log2f(last_key_multiplicity));
}

extern "C" int flexam_attn_fwd_splitkv(const void* q, int64_t q_bs, int64_t q_rs, const void* k, int64_t k_bs, int64_t k_rs,
                                       const void* v, int64_t v_bs, int64_t v_rs, void* o, int64_t o_bs, int64_t o_rs, int B, int H,
                                       int Lq, int Lk, int head_dim, float softmax_scale, int kv_splits, int split_from_unit,
                                       float* ws_o, float* ws_ml, void* stream) {
  return attn_run(q, q_bs, q_rs, k, k_bs, k_rs, v, v_bs, v_rs, o, o_bs, o_rs, B, H, Lq, Lk, head_dim, softmax_scale, kv_splits,
                  split_from_unit, ws_o, ws_ml, stream);
}

extern "C" int flexam_attn_fwd_partial(const void* q, int64_t q_bs, int64_t q_rs, const void* k, int64_t k_bs, int64_t k_rs,
                                       const void* v, int64_t v_bs, int64_t v_rs, int B, int H, int Lq, int Lk, int head_dim,
                                       float softmax_scale, int kv_splits, int slot0, float* ws_o, float* ws_ml, void* stream) {
  FX_REQUIRE(slot0 >= 0, FLEXAM_E_ARG, "attn_fwd_partial: negative slot");
  return attn_run(q, q_bs, q_rs, k, k_bs, k_rs, v, v_bs, v_rs, nullptr, 0, 0, B, H, Lq, Lk, head_dim, softmax_scale, kv_splits, 0, ws_o,
                  ws_ml, stream, slot0);
}

extern "C" int flexam_attn_fp8_pack(const void* q, int64_t q_bs, int64_t q_rs, const void* k, int64_t k_bs, int64_t k_rs, const void* v,
                                    int64_t v_bs, int64_t v_rs, void* q8, void* qs, void* kv8, int B, int H, int L, int head_dim,
                                    void* stream) {
  FX_REQUIRE(v && q8 && qs && kv8 && ((q == nullptr) == (k == nullptr)), FLEXAM_E_ARG, "attn_fp8_pack: null pointer (q and k: both or neither)");
  FX_REQUIRE(head_dim == HD && B > 0 && H > 0 && L > 0, FLEXAM_E_SHAPE, "attn_fp8_pack: bad sizes (head_dim 128 only)");
  FX_REQUIRE(q_rs % 8 == 0 && k_rs % 8 == 0 && v_rs % 8 == 0 && q_bs % 8 == 0 && k_bs % 8 == 0 && v_bs % 8 == 0, FLEXAM_E_SHAPE,
             "attn_fp8_pack: strides must keep 16-byte alignment of head rows");
  FX_REQUIRE(((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)q8 | (uintptr_t)qs | (uintptr_t)kv8) % 16 == 0, FLEXAM_E_ARG,
             "attn_fp8_pack: misaligned pointer");
  Attn8Pack a;
  a.q = (const bf16*)q; a.k = (const bf16*)k; a.v = (const bf16*)v;
  a.q_bs = q_bs; a.q_rs = q_rs; a.k_bs = k_bs; a.k_rs = k_rs; a.v_bs = v_bs; a.v_rs = v_rs;
  a.q8 = (unsigned char*)q8; a.qs = (unsigned char*)qs; a.kv8 = (unsigned char*)kv8;
  a.H = H; a.L = L; a.lq_pad = (L + QBLK - 1) / QBLK * QBLK; a.tiles = (L + KVBLK - 1) / KVBLK;
  hipLaunchKernelGGL(attn8_pack_kernel, dim3(a.lq_pad / KVBLK, H, B), dim3(256), 0, (hipStream_t)stream, a);
  return flexam_check_launch("flexam_attn_fp8_pack");
}

extern "C" int flexam_rmsnorm_rope_mx(const void* q, int64_t ldq, const float* wq, const void* k, int64_t ldk, const float* wk, void* q8,
                                      void* qs, void* kv8, int64_t M, int C, float eps, const float* rope_cos, const float* rope_sin,
                                      int64_t tokens_per_batch, int64_t token_offset, int H, int head_dim, void* stream) {
  FX_REQUIRE(q && k && wq && wk && q8 && qs && kv8 && rope_cos && rope_sin, FLEXAM_E_ARG, "rmsnorm_rope_mx: null pointer");
  FX_REQUIRE(head_dim == HD && H == 24 && C == H * HD, FLEXAM_E_SHAPE, "rmsnorm_rope_mx: 24 heads of 128 channels only (C = %d, H = %d)", C, H);
  FX_REQUIRE(M > 0 && tokens_per_batch > 0 && M % tokens_per_batch == 0 && ldq % 8 == 0 && ldk % 8 == 0, FLEXAM_E_SHAPE, "rmsnorm_rope_mx: bad sizes");
  FX_REQUIRE(((uintptr_t)q | (uintptr_t)k | (uintptr_t)q8 | (uintptr_t)qs | (uintptr_t)kv8) % 16 == 0, FLEXAM_E_ARG, "rmsnorm_rope_mx: misaligned pointer");
  RmsRopeMx a;
  a.q = (const bf16*)q; a.k = (const bf16*)k; a.ldq = ldq; a.ldk = ldk; a.wq = wq; a.wk = wk; a.cs = rope_cos; a.sn = rope_sin;
  a.q8 = (unsigned char*)q8; a.qs = (unsigned char*)qs; a.kv8 = (unsigned char*)kv8;
  a.tokens_per_batch = tokens_per_batch; a.token_offset = token_offset; a.H = H; a.L = (int)tokens_per_batch;
  a.lq_pad = (a.L + QBLK - 1) / QBLK * QBLK; a.tiles = (a.L + KVBLK - 1) / KVBLK; a.eps = eps;
  hipLaunchKernelGGL(rmsnorm_rope_mx_kernel, dim3((unsigned)M, 2), dim3(128), 0, (hipStream_t)stream, a);
  return flexam_check_launch("flexam_rmsnorm_rope_mx");
}

extern "C" int flexam_attn_fwd_fp8(const void* q8, const void* qs, const void* kv8, void* o, int64_t o_bs, int64_t o_rs, int B, int H, int L,
                                   int head_dim, int kv_splits, int split_from_unit, float* ws_o, float* ws_ml, void* stream) {
  FX_REQUIRE(q8 && qs && kv8, FLEXAM_E_ARG, "attn_fwd_fp8: null pointer");
  FX_REQUIRE(((uintptr_t)q8 | (uintptr_t)qs | (uintptr_t)kv8) % 16 == 0, FLEXAM_E_ARG, "attn_fwd_fp8: misaligned pointer");
  return attn_run(nullptr, 0, 0, nullptr, 0, 0, nullptr, 0, 0, o, o_bs, o_rs, B, H, L, L, head_dim, FLEXAM_ATTN_PRESCALED, kv_splits,
                  split_from_unit, ws_o, ws_ml, stream, -1, 0.f, q8, qs, kv8);
}

extern "C" int flexam_attn_fwd_fp8_chunked(const void* q8, const void* qs, const void* kv8, void* o, int64_t o_bs, int64_t o_rs, int B, int H,
                                           int Lq, int Lk, int chunk_tiles, int head_dim, int kv_splits, int split_from_unit, float* ws_o,
                                           float* ws_ml, void* stream) {
  FX_REQUIRE(q8 && qs && kv8, FLEXAM_E_ARG, "attn_fwd_fp8_chunked: null pointer");
  FX_REQUIRE(((uintptr_t)q8 | (uintptr_t)qs | (uintptr_t)kv8) % 16 == 0, FLEXAM_E_ARG, "attn_fwd_fp8_chunked: misaligned pointer");
  FX_REQUIRE(chunk_tiles > 0, FLEXAM_E_SHAPE, "attn_fwd_fp8_chunked: chunk_tiles = %d", chunk_tiles);
  return attn_run(nullptr, 0, 0, nullptr, 0, 0, nullptr, 0, 0, o, o_bs, o_rs, B, H, Lq, Lk, head_dim, FLEXAM_ATTN_PRESCALED, kv_splits,
                  split_from_unit, ws_o, ws_ml, stream, -1, 0.f, q8, qs, kv8, chunk_tiles);
}

#ifdef FLEXAM_ATTN_STAMPS
// diagnostic builds only (not declared in flexam_hip.h): copies the per-workgroup stamps of the last attention launch to the host
extern "C" int flexam_debug_attn_stamps(unsigned long long* out, int n_workgroups) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_attn_stamps), (size_t)n_workgroups * 16) == hipSuccess ? 0 : -1;
}
extern "C" int flexam_debug_attn_barrier_wait(unsigned long long* out, int n_workgroups) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_attn_barrier_wait), (size_t)n_workgroups * 64) == hipSuccess ? 0 : -1;
}
#endif

extern "C" int flexam_attn_merge(void* o, int64_t o_bs, int64_t o_rs, int B, int H, int Lq, int head_dim, float softmax_scale,
                                 int n_slots, const float* ws_o, const float* ws_ml, void* stream) {
  FX_REQUIRE(o && ws_o && ws_ml, FLEXAM_E_ARG, "attn_merge: null pointer");
  FX_REQUIRE(head_dim == HD && B > 0 && H > 0 && Lq > 0 && n_slots >= 1, FLEXAM_E_SHAPE, "attn_merge: bad sizes");
  FX_REQUIRE(o_rs % 4 == 0 && o_bs % 4 == 0 && (uintptr_t)o % 8 == 0, FLEXAM_E_SHAPE, "attn_merge: output rows must keep 8-byte alignment");
  AttnParams p{};
  p.o = (bf16*)o; p.o_bs = o_bs; p.o_rs = o_rs; p.B = B; p.H = H; p.Lq = Lq;
  p.prescaled = softmax_scale < 0.f;
  p.scale_log2e = p.prescaled ? 1.0f : softmax_scale * 1.4426950408889634f;
  p.q_blocks = (Lq + QBLK - 1) / QBLK;
  p.unit0 = 0; p.n_units = B * H * p.q_blocks; p.n_slots = n_slots;
  p.ws_o = const_cast<float*>(ws_o); p.ws_ml = const_cast<float*>(ws_ml);
  const int64_t g = ((int64_t)p.n_units * QBLK + 3) / 4;
  hipLaunchKernelGGL(attn_merge_kernel, dim3((unsigned)(g > 16384 ? 16384 : g)), dim3(256), 0, (hipStream_t)stream, p);
  return flexam_check_launch("flexam_attn_merge");
}
