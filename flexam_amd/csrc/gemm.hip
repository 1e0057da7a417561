// flexam_amd/csrc/gemm.hip -- bf16 MFMA GEMM for the DiT projections / FFN and (through the
// per-K-block A offset table) the VAE's implicit-GEMM causal convolutions.
//
//   C[M,N] = epilogue( A[M,K] . W[N,K]^T + bias[N] )        A, W bf16 row-major (K contiguous)
//
// Replaces the cuBLAS/hipBLASLt calls behind nn.Linear in the reference
// (FlexAM/models/wan_transformer3d_FlexAM.py:242-244,261,363-365,370,415-416) and cuDNN
// Conv3d/Conv2d (FlexAM/models/wan_vae3_8.py:39-47,94,99).
//
// MI355X mapping (MI355X_MICROARCH.md / cdna_hip_programming.md section 5):
//  * 256x256x64 tile per 512-thread workgroup (8 waves = 2(M) x 4(N), 128x64 outputs per wave),
//    one workgroup per CU, accumulators in 128 VGPRs per lane; shorter tiles (224..128 rows) are picked
//    when they fill the last round of 256 CUs (pick_mt).
//  * both operands staged HBM/L2 -> LDS with 16-byte global_load_lds (no VGPR round trip), two
//    LDS buffers (2 x 64 KiB); tile rows are 128 B so the 16-B chunk index is XOR-swizzled with
//    (row>>1)&7 on the *source* address (the LDS image must stay lane-linear for LDS-DMA) and
//    again on the ds_read_b128 fragment reads: conflict-free per tools/lds_sim.py.
//  * software pipeline: two fragment register sets and ONE barrier per 64-deep K block placed between
//    the two 32-deep halves, so MFMA work is always in registers on both sides of the barrier; the
//    LDS-DMA pieces and ds_reads of a phase are issued one per 4 MFMAs (an LDS-DMA piece costs ~40
//    issue cycles next to MFMAs, tools/probes/issue_probe.hip).
//  * v_mfma_f32_16x16x32_bf16 with the operands swapped (W fragment in the A slot) so each lane
//    ends up with 4 consecutive N columns of one output row -> 8-byte bf16 / 16-byte f32 stores.
//  * workgroup -> tile map is XCD-aware: the 8 XCDs (private 4 MiB L2 each) get contiguous
//    chunks of the tile list, walked in groups of 4 tile-rows so concurrently resident tiles
//    share A row-panels and W column-panels in L2 (FLEXAM_GEMM_GM overrides the group height).
#include <math.h>
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "flexam_hip.h"

namespace {

constexpr int BN = 256, BK = 64;
constexpr int TILE_BYTES = 256 * BK * 2;   // 32 KiB per operand tile (A: up to 256 rows)

struct GemmParams {
  const bf16* A;
  const bf16* W;
  void* C;
  const float* bias;
  int64_t lda, ldw, ldc;
  int M, N, K;
  int tiles_m, tiles_n;
  // fused residual epilogue (EPI_GATE_RESIDUAL): X[m, n] += bf16round(acc + bias) * gate[row(m), n]
  float* X;
  int64_t ldx;
  const float* gate;       // [rows, gate_ld] or null (gate = 1)
  int64_t gate_ld;
  const int32_t* gate_row; // [M] row index per output row, or null
  int64_t rows_per_batch;  // used when gate_row is null: row = m / rows_per_batch
  int gm;                  // tile-rows per L2 group (XCD-aware order)
  // tail split-K: work units 0..split_full-1 are whole tiles; the remaining tiles (the last, partial round of the CUs) are
  // cut into split_s K slices each; their partial sums are parked in `ws` and a second, stream-ordered launch
  // (gemm_splitk_finish_kernel) adds the slices up in slice order and runs the epilogue
  int units, split_full, split_s;
  float* ws;
  int x_nt;                // gate-residual epilogue: X leaves / arrives with non-temporal hints (launch() decides: only when X is larger than the Infinity Cache can keep)
  int debug;               // only in -DFLEXAM_GEMM_ABLATE builds (timing ablations, WRONG results): 1 no vmcnt wait, 2 no barrier, 4 no LDS-DMA, 8 half the ds_reads, 16 half the LDS-DMA
};

template <int V>
using IC = std::integral_constant<int, V>;

#if (defined(FLEXAM_GEMM_ABLATE) || defined(FLEXAM_GEMM_HALF_FRAG) || defined(FLEXAM_GEMM_STAMPS)) && !defined(FLEXAM_DIAGNOSTIC_BUILD)
#error "FLEXAM_GEMM_ABLATE / FLEXAM_GEMM_HALF_FRAG (WRONG results: timing ablations) and FLEXAM_GEMM_STAMPS (in-kernel clock stamps) are switches of diagnostic builds: add -DFLEXAM_DIAGNOSTIC_BUILD"
#endif
#ifdef FLEXAM_GEMM_ABLATE
#define ABLATE(p, bit) ((p).debug & (bit))
#else
#define ABLATE(p, bit) 0
#endif

#ifdef FLEXAM_GEMM_STAMPS      // diagnostic builds only: the anatomy of a launch -- per workgroup the 100 MHz counter at kernel entry, when the first K block of its
__device__ unsigned long long g_gemm_stamps[4 * 1024];      // first unit has landed, at the end of its last K loop and at exit; read by nobody on the device
#define GEMM_STAMP(i) if (threadIdx.x == 0 && blockIdx.x < 1024) g_gemm_stamps[4 * blockIdx.x + (i)] = __builtin_amdgcn_s_memrealtime()
#else
#define GEMM_STAMP(i)
#endif
enum { EPI_NONE = 0, EPI_GELU = 1, EPI_GATE_RESIDUAL = 2 };

// MT = 16-row m-tiles per wave: the workgroup tile is (32*MT) x 256 outputs, 8 waves = 2(M) x 4(N), two per SIMD.
// MT = 8 (256 x 256) is the throughput shape; 7..4 exist so that a launch whose tile count is a little over a
// multiple of the 256 CUs (N = 3072 projections: 91 x 12 tiles = 4.27 rounds) can trade tile height for a
// full last round (launch() picks MT).  (A 4-wave variant with 128x128 outputs per wave, one wave per SIMD and
// AGPR accumulators ran at a higher clock -- a third fewer LDS bytes per MFMA -- but lower MFMA occupancy, 5-8 %
// slower overall: profiles/r1e_gemm_notes.txt; it is in the history, not in the tree.)
// The MFMA is v_mfma_f32_16x16x32_bf16.  (A 32x32x16 form of the same kernel -- half as many MFMA instructions, 10 % fewer cycles --
// ran 3-4 % slower on every DiT shape: the power-capped clock falls 13 %; profiles/r1e_gemm_notes.txt #15.  Not in the tree.)
// Everything outside the MFMA calls is written over "row tiles" of RT = 16 rows: lane -> row (lane % RT) of a row tile and column
// group g = lane / RT; a lane holds, for every 4-column unit v < NV of the wave's 64 columns, the 4 consecutive columns
// 4*NG*v + 4*g .. +3 of that row (NG = 4, NV = 4).
// (A 4-wave instance on 128 x 128 tiles, 74 KiB of LDS, TWO workgroups per CU -- meant to run one workgroup's fp32 read-modify-write
// epilogue under its neighbour's K loop for the N = K = 3072 gate-residual launches -- was built in r4: bit-exact, K loops +22 %, and
// the epilogue stayed exposed, because it is the HBM time of X's 572 MB, not a per-CU latency: profiles/r4z_oproj_two_workgroups_per_cu.txt.)
// TAIL: the instance that runs the K slices of the tail tiles (units >= split_full) and parks their partial sums; it has no
// epilogue (gemm_splitk_finish_kernel runs it).  The TAIL = false instance runs the whole tiles only.  Two instances instead of
// one kernel with both paths: with the slab stores between the K loop and the epilogues hipcc spills 30-50 registers in the
// gate-residual epilogue and leaves a reload pending into the next unit's K loop.
// WMW x (8 / WMW) waves, NTW 16-wide n-tiles per wave: 2 x 4 waves x 4 n-tiles = the (32 MT) x 256 tile of the DiT shapes; 4 x 2
// waves x 5 n-tiles = a (64 MT) x 160 tile for output widths that are multiples of 160 but not of 256 (the VAE encoder's 160 / 320 /
// 640 channels, which fill 62.5 % / 62.5 % / 83 % of 256-wide tiles).  The 160-wide shape stores through the generic epilogue (its
// 80-column wave rows do not fit the 128-byte LDS turn-around).
template <int EPI, typename OutT, int MT, bool TAIL = false, int WMW = 2, int NTW = 4>
__global__ __launch_bounds__(512, 2) void gemm_bf16_kernel(GemmParams p, const int64_t* __restrict__ a_koff) {
  static_assert((WMW == 2 && (NTW == 4 || NTW == 3)) || (WMW == 4 && NTW == 5 && MT <= 6),
                "supported wave layouts: 2 x 4 waves x 4 or 3 n-tiles, 4 x 2 waves x 5 n-tiles");
  // STD: the LDS-staged epilogues (a wave's row of 16 NTW bf16 outputs turned around through a 128-byte LDS row).  NTW = 4: the 256-wide
  // tile of the DiT shapes.  NTW = 3 (r6): a 192-wide tile, (32 MT) x 192 -- with MT = 6 a 192 x 192 tile, 16 x 16 = 256 of which cover
  // a rank-of-eight's 2912 x 3072 outputs in ONE full round of the CUs (94.8 % useful) where 160 x 256 tiles are 228 tiles at 89 % fill:
  // 10 % less matrix work per CU on the N = 3072 launches of a rank's block (o-proj, cross-o, cross-q, FFN2).  Its wave rows are 48
  // columns = 96 bytes = 6 of the 8 sixteen-byte slots of a staging row; lanes that would carry slots 6, 7 (bf16 store: lane % 8 >= 6;
  // fp32 read-modify-write: lane % 16 >= 12) SHADOW the last piece: same addresses, same loads, the same value stored again by the same
  // instruction.  (Predicating their stores instead put the epilogue's LDS reads behind branches, and hipcc then let the staging
  // writes of the next row tile overtake the last reads of this one: rows 12-15 of every other row tile came out as the next tile's.)
  constexpr bool STD = WMW == 2 && (NTW == 4 || NTW == 3);
  constexpr int CW = 16 * NTW;                 // columns per wave
  constexpr int WNW = 8 / WMW;              // waves along N
  constexpr int BN_ = WNW * NTW * 16;       // columns of this tile shape
  constexpr int RT = 16;                    // rows per row tile
  constexpr int NRT = 16 * MT / RT;         // row tiles per wave
  constexpr int NG = 64 / RT;               // column groups (lane / RT)
  constexpr int NV = NTW * 4 / NG;          // 4-column units per lane
  constexpr int STG_WAVE = RT * 128;        // epilogue staging per wave: one row tile of bf16 outputs (RT rows x 64 columns)
  constexpr int BM_ = WMW * 16 * MT;        // rows of this tile shape
  constexpr int PA = (BM_ + 63) / 64;       // 64-row staging pieces per thread for A
  constexpr int PW = (BN_ + 63) / 64;       // ... and for W
  constexpr int NP = PA + PW;               // LDS-DMA pieces per thread per K block
  // LDS image of one K block: [A tile | W tile].  Tiles of up to 256 rows keep the 32 KiB + 32 KiB form; the 384 x 160 shape (r6: 4 x 2
  // waves of 96 x 80 outputs, 30 MFMAs per 11 fragment reads instead of 20 per 9 on the 256 x 160 tile -- the VAE encoder's 160-channel
  // 3x3x3 convolutions over millions of rows) holds 48 KiB of A and three 8 KiB pieces of W per buffer; it has no LDS-staged epilogue.
  constexpr int A_BYTES = BM_ > 256 ? BM_ * 128 : TILE_BYTES;
  constexpr int BUF_BYTES = BM_ > 256 ? A_BYTES + PW * 8192 : 2 * TILE_BYTES;
  static_assert(BM_ <= 256 || !(WMW == 2), "tall tiles: the 160-wide shape only");
  constexpr int NF = NTW + MT;              // fragments per 32-deep K half
  // a_koff: optional [K/BK] element offsets added to every A row base per K block (implicit conv)
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [2 buffers][A tile | W tile], then [8 waves][STG_WAVE] of epilogue staging
  const int tid = threadIdx.x;
  // The builtin, so that hipcc places the wait state gfx950 wants between a VALU write of a VGPR and a v_readfirstlane of it (a
  // hand-written v_readfirstlane right behind the shift read a stale register: memory faults); the empty asm makes the SGPR value
  // opaque, so it is kept (or parked in a VGPR lane) instead of re-derived from a spilled copy of threadIdx.x in front of every use.
  int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  asm volatile("" : "+s"(wave));
  const int wm = wave / WNW, wn = wave % WNW;
  if constexpr (!TAIL) { GEMM_STAMP(0); }

  // ---- persistent workgroups over an XCD-aware, grouped tile order: workgroup w lives on XCD w & 7 (round-robin
  // dispatch); that XCD owns a contiguous chunk of the tile list and its gridDim/8 workgroups walk the chunk with
  // stride gridDim/8, so the tiles resident on an XCD at any time are neighbours in the list (shared A / W panels in
  // its L2) and a workgroup pays its launch latency once, not once per tile.
  // Whole tiles (units < split_full) are chunked per XCD as described; the K slices of the tail tiles (units >= split_full)
  // are dealt round-robin over all workgroups afterwards, so every XCD gets the same amount of tail work.
  const int nwg = p.split_full;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = blockIdx.x & 7, per_xcd = gridDim.x >> 3;
  const int chunk0 = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
  const int chunk_n = q8 + (xcd < r8 ? 1 : 0);
  const int local = blockIdx.x >> 3;
  const int n_whole = TAIL ? 0 : (local < chunk_n ? (chunk_n - local + per_xcd - 1) / per_xcd : 0);
  const int n_tail = !TAIL ? 0 : ((int)blockIdx.x < p.units - p.split_full ? (p.units - p.split_full - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0);
  auto nth_unit = [&](int j) -> int {       // j-th work unit of this workgroup, -1 past the end
    if (j < n_whole) return chunk0 + local + j * per_xcd;
    if (j < n_whole + n_tail) return p.split_full + (int)blockIdx.x + (j - n_whole) * (int)gridDim.x;
    return -1;
  };
  // tile index in the grouped order -> first row / column
  auto tile_origin = [&](int bid, int& m0, int& n0) {
    int GM = p.gm;
    asm volatile("" : "+s"(GM));         // laundered: the uniform divisions below are redone per unit (a few VALU operations) instead of
    const int per_group = GM * p.tiles_n;   // keeping their float reciprocals alive across the K loop in VGPRs that end up in scratch
    const int group = bid / per_group;
    const int first_m = group * GM;
    const int gsz = min(p.tiles_m - first_m, GM);
    const int in_group = bid - group * per_group;
    m0 = (first_m + in_group % gsz) * BM_;
    n0 = (in_group / gsz) * BN_;
  };
  // ---- staging addresses: thread -> (row = i*64 + tid/8, LDS slot = tid%8), source chunk = slot ^ swz(row).
  // Byte offsets from the tile's first row fit 32 bits (256 rows x row stride), so a piece's source is
  // (uniform 64-bit tile base + K offset) + one VGPR: the scalar-base form of global_load_lds.
  uint32_t a_off[PA], w_off[PW];
  const char *a_tile, *w_tile;
  auto stage_setup = [&](int m0, int n0) {
    const int ts = (wave << 6) | fresh_lane();          // thread id, rebuilt (see fresh_lane)
#pragma unroll
    for (int i = 0; i < (PA > PW ? PA : PW); ++i) {
      const int row = i * 64 + (ts >> 3);
      const int chunk = (ts & 7) ^ ((row >> 1) & 7);
      if (i < PA) a_off[i] = (uint32_t)(((int64_t)min(row, p.M - 1 - m0) * p.lda + chunk * 8) * 2);   // edge tiles re-read their last row
      if (i < PW) w_off[i] = (uint32_t)(((int64_t)min(row, p.N - 1 - n0) * p.ldw + chunk * 8) * 2);
    }
    a_tile = (const char*)(p.A + (int64_t)m0 * p.lda);
    w_tile = (const char*)(p.W + (int64_t)n0 * p.ldw);
  };

  // ---- fragment read offsets (bytes inside a tile): row = base16 + (lane&15), chunk = (lane>>4) + 4*ks
  // row lane%16, 16-byte chunk lane/16 of a 32-deep K half.
  // Rebuilt at the top of every unit from a fresh lane id: alive in the K loop only, not across the epilogue (where they were
  // spilled and came back behind a vmcnt(0) that drained the epilogue's stores in front of the next K loop).
  int frag_off[2];
  auto frag_setup = [&]() {
    const int lf = fresh_lane();
    const int sw = ((lf & (RT - 1)) >> 1) & 7;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
      frag_off[ks] = (lf & (RT - 1)) * 128 + (((4 * ks + (lf >> 4)) ^ sw) << 4);
  };
  // ---- bias of the wave's 64 columns through LDS (standard tile shape).  A plain bias load at the top of the epilogue sits BEHIND
  // the next unit's 16 prefetched LDS-DMA pieces in the wave's in-order vmcnt queue: the first use of the bias waited for all of
  // them (s_waitcnt vmcnt(0): a full DMA latency with the matrix pipe idle, once per tile).  Instead one 256-byte LDS-DMA per unit,
  // issued IN FRONT of the unit's K block 0 pieces (so every wait that covers K block 0 covers it), into one of two slots (unit
  // parity: the next unit's bias is on its way while this unit's epilogue reads its own).
  constexpr bool BIAS_LDS = STD && !TAIL;
  constexpr int BIAS_OFF = 4 * TILE_BYTES + 8 * STG_WAVE;       // [2 slots][8 waves][64 floats]
  auto bias_dma = [&](int n0, int slot) {
    if constexpr (BIAS_LDS) {
      const float* pb = p.bias;          // laundered: the test is made here, on scalar registers, not kept as a 0 / 1 VGPR across the K loop
      asm volatile("" : "+s"(pb));       // (with ~100 SGPRs in use hipcc parks loop-invariant uniform values in VGPRs and then spills those)
      if (pb) {
        const uint32_t voff = (uint32_t)min(n0 + wn * (16 * NTW) + fresh_lane(), p.N - 1) * 4u;
        const uint32_t dst = (uint32_t)(uintptr_t)LDS_PTR(smem) + BIAS_OFF + (slot * 8 + wave) * 256;
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" ::"v"(voff), "s"(pb), "s"(dst) : "memory");
      }
    }
  };
  // ---- gate rows of the wave tile through LDS the same way (gate-residual epilogue with a per-row table index): two dwords per lane
  // (tile rows l and 64 + l of the wave's 16 MT rows) by LDS-DMA in front of the unit's K block 0.  Fetched inside the epilogue --
  // `p.gate_row[m]` right in front of the gate loads that need it, once per batch of 32 rows -- the index loads were the YOUNGEST
  // entries of the wave's in-order vmcnt queue: waiting for them drained every X load and the previous batch's stores, 4 times
  // per tile.
  constexpr bool GROW_LDS = STD && !TAIL && EPI == EPI_GATE_RESIDUAL;
  constexpr int GROW_OFF = BIAS_OFF + 2 * 8 * 256;              // [2 slots][8 waves][128 ints]
  auto grow_dma = [&](int m0, int slot) {
    if constexpr (GROW_LDS) {
      const int32_t* pg = p.gate_row;
      asm volatile("" : "+s"(pg));
      if (pg) {
#pragma unroll
        for (int h = 0; h < (16 * MT > 64 ? 2 : 1); ++h) {
          const uint32_t voff = (uint32_t)min(m0 + wm * (16 * MT) + 64 * h + fresh_lane(), p.M - 1) * 4u;
          const uint32_t dst = (uint32_t)(uintptr_t)LDS_PTR(smem) + GROW_OFF + ((slot * 8 + wave) * 128 + 64 * h) * 4;
          asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" ::"v"(voff), "s"(pg), "s"(dst) : "memory");
        }
      }
    }
  };
  if constexpr (BIAS_LDS) {
    if (!p.bias) {                // no bias: both slots hold zeros for the whole launch (wave-local, ordered before any later read)
      float* z = (float*)(smem + BIAS_OFF + wave * 256);
      z[fresh_lane()] = 0.f;
      z[8 * 64 + fresh_lane()] = 0.f;
    }
  }
  bool staged = false;          // K blocks 0 and 1 of the coming unit are already on their way into the two LDS buffers
  bool pend = false;            // ... and exactly PEND stores of the last epilogue were issued by this wave after them
  constexpr int PEND = (EPI == EPI_GATE_RESIDUAL ? 4 : 2) * MT;
  const int nk = p.K / BK;
  auto kcol_a = [&](int kb) -> int64_t { return a_koff ? a_koff[kb < nk ? kb : nk - 1] : (int64_t)kb * BK; };
  // work unit -> (tile, first K block, K blocks, slice, slices)
  auto unit_decode = [&](int u, int& tile, int& kb0, int& nkl, int& slice, int& ns) {
    if (u < p.split_full) {
      tile = u; kb0 = 0; nkl = nk; slice = 0; ns = 1;
    } else {
      const int v = u - p.split_full;
      tile = p.split_full + v / p.split_s;
      slice = v - (v / p.split_s) * p.split_s;
      ns = p.split_s;
      kb0 = (int)((int64_t)slice * nk / ns);
      nkl = (int)((int64_t)(slice + 1) * nk / ns) - kb0;
    }
  };

  for (int it = 0; it < n_whole + n_tail; ++it) {
  int m0, n0, tile, kb0, nkl, slice, ns;
  unit_decode(nth_unit(it), tile, kb0, nkl, slice, ns);
  tile_origin(tile, m0, n0);
  if (!staged) stage_setup(m0, n0);

  f32x4 acc[MT][NTW];                        // 16x16 tiles: [m-tile][n-tile]
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NTW; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // the 4 consecutive columns of unit v in row tile t (see the kernel header)
  auto accv = [&](int t, int v) -> f32x4 { return acc[t][v]; };

  // piece i (i < PA: A rows i*64.., else W rows (i-PA)*64..) of the tile at A K-offset ka / W K-offset kw -> LDS buffer `buf`.
  // Inline asm: the scalar-base form (uniform 64-bit base + one 32-bit VGPR offset) keeps the per-thread
  // offsets in PA + 4 VGPRs; completion is tracked by hand (s_waitcnt vmcnt(0) in front of each barrier).
  auto dma = [&](int i, int64_t ka, int64_t kw, char* buf) {
    const char* sbase = i < PA ? a_tile + ka * 2 : w_tile + kw * 2;
    const uint32_t voff = i < PA ? a_off[i] : w_off[i - PA];
    const uint32_t dst = (uint32_t)(uintptr_t)LDS_PTR(buf) + (i < PA ? i * 8192 : A_BYTES + (i - PA) * 8192) + wave * 1024;
    if (ABLATE(p, 4) || (ABLATE(p, 16) && i >= PA)) return;      // 16: no W-tile staging (half the LDS-DMA)
    // M0 (the LDS base of the DMA) is written and NOT restored: nothing else in this kernel uses it (gfx9+ LDS instructions do not;
    // tools/isa_loopwaits.py lists any other M0 reader of the listing), and the save / restore pair was 2 of the 6 scalar
    // instructions of every piece
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(dst) : "memory");
  };
  auto wait_barrier = [&](auto n_c) {                  // at most N of this wave's vector-memory operations still in flight, then barrier
    if (!ABLATE(p, 1)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(decltype(n_c)::value) : "memory");
    if (!ABLATE(p, 2)) __syncthreads();
  };
  // fragment j (< NF = MT + NTW) of the set of K half hf: j < NTW -> W n-tile j, else A m-tile j - NTW, all 32 deep
  auto frag = [&](const char* buf, int hf, int j) -> bf16x8 {
    const int fo = frag_off[hf];
    return j < NTW ? *(const bf16x8*)(buf + A_BYTES + wn * (16 * NTW * 128) + j * 2048 + fo)
                   : *(const bf16x8*)(buf + wm * (16 * MT * 128) + (j - NTW) * 2048 + fo);
  };
  // fragments 2g, 2g+1 of a set
  // (PER fragments per MFMA group so that the MT groups of a phase cover all NF: 2 for every 256-wide shape, 3 for the 160-wide one)
  constexpr int PER = (NF + MT - 1) / MT > 2 ? (NF + MT - 1) / MT : 2;
  auto ld2 = [&](const char* buf, int hf, int g, bf16x8 (&f)[NF]) {
#pragma unroll
    for (int j = PER * g; j < PER * g + PER; ++j)
      if (j < NF) f[j] = frag(buf, hf, j);
  };
  // MFMA group g (of MT per K half, 64 matrix-pipe cycles each) on fragment set f
  auto mfma_group = [&](int g, const bf16x8 (&f)[NF]) {
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) acc[g][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[nt], f[NTW + g], acc[g][nt], 0, 0, 0);
  };

  // ---- main loop: two LDS buffers, two fragment sets (F0: k 0..31, F1: k 32..63 of a K block).  Per K block:
  //   phase A: MFMAs on F0 while F1 is read from `cur`
  //   barrier : every wave holds its F1 (cur is free) and tile kb+1 has landed in `nxt` for everyone
  //   phase B: LDS-DMA of tile kb+2 into `cur`; MFMAs on F1 while F0 of block kb+1 is read from `nxt`
  // so no wave crosses the barrier without MFMA work already in registers, and the DMA pieces and fragment
  // reads of a phase are spread one group (4 MFMAs) apart instead of stalling the wave up front.
  bf16x8 f0[NF], f1[NF];
  if (!staged) {
    bias_dma(n0, it & 1);
    grow_dma(m0, it & 1);
#pragma unroll
    for (int i = 0; i < NP; ++i) dma(i, kcol_a(kb0), (int64_t)kb0 * BK, smem);
    if (nkl > 1) {
      const int64_t k1 = kcol_a(kb0 + 1);
#pragma unroll
      for (int i = 0; i < NP; ++i) dma(i, k1, (int64_t)(kb0 + 1) * BK, smem + BUF_BYTES);
    }
  }
  int64_t kcol_next = kcol_a(kb0 + 2);    // A offset of the K block staged next, fetched one step ahead
  // In flight per wave, oldest first: K block 0 (NP pieces), K block 1 (NP pieces, if there is one), then the PEND stores of the
  // previous unit's epilogue.  vmcnt retires in issue order, so the waits for the two K blocks can leave the stores in
  // flight: they drain under K block 0 instead of in front of it.
  const bool counted = pend && nkl >= 3;
  if (counted) wait_barrier(IC<NP + PEND>{});
  else if (!pend && nkl > 1) wait_barrier(IC<NP>{});     // only K block 1 is younger than K block 0
  else wait_barrier(IC<0>{});                            // (one or two K blocks: drain everything)
  pend = false;
  if constexpr (!TAIL) { if (it == 0) { GEMM_STAMP(1); } }
  // The staging offsets are read by asm statements inside the K loop.  Should one of them ever come back from a spill slot, the
  // compiler's own wait for that reload must land here and not in the loop, where a vmcnt(0) would drain the LDS-DMA pipeline
  // on every K block.
#pragma unroll
  for (int i = 0; i < (PA > PW ? PA : PW); ++i) {
    if (i < PA) asm volatile("" : "+v"(a_off[i]));
    if (i < PW) asm volatile("" : "+v"(w_off[i]));
  }
  frag_setup();
#pragma unroll
  for (int g = 0; g < (NF + PER - 1) / PER; ++g) ld2(smem, 0, g, f0);
#ifdef FLEXAM_GEMM_HALF_FRAG
#pragma unroll
  for (int j = 0; j < NF; ++j) f1[j] = f0[j];
#endif

  auto block = [&](int kb, auto dma_c, auto rd_c, auto wait_c) {
    constexpr bool DMA = decltype(dma_c)::value, RD = decltype(rd_c)::value;
    char* cur = smem + (kb & 1) * BUF_BYTES;
    char* nxt = smem + ((kb + 1) & 1) * BUF_BYTES;
    const int64_t kw = (int64_t)(kb0 + kb + 2) * BK;
#pragma unroll
    for (int g = 0; g < MT; ++g) {                      // phase A
      __builtin_amdgcn_sched_barrier(0);
      mfma_group(g, f0);
      __builtin_amdgcn_sched_barrier(0);                // MFMAs first: the wait for F0 must not cover reads issued after it
#ifndef FLEXAM_GEMM_HALF_FRAG                           // (compile-time form of ablation 8, without its run-time tests in the loop: the ceiling of
      if (!ABLATE(p, 8)) ld2(cur, 1, g, f1);            //  what fewer fragment reads per MFMA -- 128 x 128 wave tiles: a third fewer -- could win)  8: no phase-A fragment reads (half the ds_reads)
#endif
    }
    __builtin_amdgcn_sched_barrier(0);
    wait_barrier(wait_c);
#pragma unroll
    for (int g = 0; g < MT; ++g) {                      // phase B
      __builtin_amdgcn_sched_barrier(0);
      mfma_group(g, f1);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (DMA) {
        dma(g, kcol_next, kw, cur);
        if (g + MT < NP) dma(g + MT, kcol_next, kw, cur);
      }
      if constexpr (RD) ld2(nxt, 0, g, f0);
    }
    __builtin_amdgcn_sched_barrier(0);
    kcol_next = kcol_a(kb0 + kb + 3);
  };
  using T_ = std::integral_constant<bool, true>;
  using F_ = std::integral_constant<bool, false>;
  int kb = 0;
  if (counted) { block(0, T_{}, T_{}, IC<PEND>{}); kb = 1; }     // K block 1 has landed; the epilogue stores may still be in flight
  for (; kb + 2 < nkl; ++kb) block(kb, T_{}, T_{}, IC<0>{});
  if (kb + 1 < nkl) { block(kb, F_{}, T_{}, IC<0>{}); ++kb; }
  block(kb, F_{}, F_{}, IC<0>{});
  if constexpr (!TAIL) { GEMM_STAMP(2); }

  // ---- the next unit's first two K blocks go on their way before this tile's epilogue: both buffers are idle from here on
  // (every wave passed the last barrier with its fragments in registers) and the epilogue has its own slice of LDS
  staged = it + 1 < n_whole + n_tail;
  if (staged) {
    int nm0, nn0, ntile, nkb0, nnkl, nslice, nns;
    unit_decode(nth_unit(it + 1), ntile, nkb0, nnkl, nslice, nns);
    tile_origin(ntile, nm0, nn0);
    stage_setup(nm0, nn0);
    bias_dma(nn0, (it + 1) & 1);
    grow_dma(nm0, (it + 1) & 1);
#pragma unroll
    for (int i = 0; i < NP; ++i) dma(i, kcol_a(nkb0), (int64_t)nkb0 * BK, smem);
    if (nnkl > 1) {
      const int64_t k1 = kcol_a(nkb0 + 1);
#pragma unroll
      for (int i = 0; i < NP; ++i) dma(i, k1, (int64_t)(nkb0 + 1) * BK, smem + BUF_BYTES);
    }
  }

  // ---- tail split-K: park this slice's partial sums and go on; gemm_splitk_finish_kernel, launched behind this kernel on the
  // same stream, adds the slices of a tile up in slice order (deterministic) and runs the epilogue.  The kernel boundary is the
  // hand-off: plain stores here, plain loads there.  (r1 handed over inside the launch -- sc1 stores, an arrival counter, the
  // last slice reducing with sc1 loads; at the production FFN2 shape and at the VAE's K = 27648 convolutions some lanes of the
  // reducing workgroup read zeros instead of another slice's sums: profiles/r2_splitk_handoff_bug.txt.)
  if constexpr (TAIL) {
    constexpr int SLAB = 256 * BN;                     // floats per slab (every tile shape fits); 16-byte element of (row tile t, unit v) at ((t*NV+v)*512 + tid)*4
    const int te = (wave << 6) | fresh_lane();         // rebuilt (see fresh_lane): slab addresses stay out of the K loop's registers
    const int tr = tile - p.split_full;
    float* slab = p.ws + ((int64_t)tr * ns + slice) * SLAB;
#pragma unroll
    for (int t = 0; t < NRT; ++t)
#pragma unroll
      for (int v = 0; v < NV; ++v) *(f32x4*)(slab + ((t * NV + v) * 512 + te) * 4) = accv(t, v);
    __builtin_amdgcn_s_waitcnt(0x0F70);                // vmcnt(0): the stores are drained here (pend stays false)
    continue;
  }
  if constexpr (!TAIL) {

  // ---- epilogue: lane holds C[m = .. + RT*t + lane%RT][n = .. + 4*NG*v + 4*(lane/RT) + 0..3] per (row tile t, unit v)
  const int le = fresh_lane();       // not the kernel's `lane`: the epilogue's per-lane address arithmetic must not be hoisted out of
                                     // the tile loop, nor keep the lane id alive across the K loop (spills, see fresh_lane)
  const int mrow = m0 + wm * (16 * MT) + (le & (RT - 1));
  const int ncol = n0 + wn * (16 * NTW) + (le / RT) * 4;
  f32x4 bias[NV];
  if constexpr (BIAS_LDS) {
    const float* bl = (const float*)(smem + BIAS_OFF + ((it & 1) * 8 + wave) * 256);     // landed with this unit's K block 0
#pragma unroll
    for (int v = 0; v < NV; ++v) bias[v] = *(const f32x4*)(bl + v * (4 * NG) + (le / RT) * 4);
  } else {
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int n = ncol + v * (4 * NG);
      bias[v] = (p.bias && n < p.N) ? *(const f32x4*)(p.bias + n) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  }
  // "used" on every path: a bias load that some path never reads stays pending in the compiler's wait bookkeeping, which then
  // puts a vmcnt(0) in front of the first MFMA of the next unit that reuses the register - draining the epilogue stores
#pragma unroll
  for (int v = 0; v < NV; ++v) asm volatile("" ::"v"(bias[v]));
  // LDS staging of one row tile (RT rows x 64 bf16 columns = RT x 128 bytes per wave, 16-byte chunks XOR-swizzled by row: the
  // 8-byte writes and the wider reads are conflict-free).  Unit v of a lane starts at byte 8*NG*v + 8*g of its row.
  char* stg = smem + 4 * TILE_BYTES + wave * STG_WAVE;
  const int wr_row = le & (RT - 1), wr_g = le / RT;
  auto stg_write = [&](int v, bf16x4 o) {
    *(bf16x4*)(stg + wr_row * 128 + ((((NG / 2) * v + (wr_g >> 1)) ^ (wr_row & 7)) << 4) + (wr_g & 1) * 8) = o;
  };
  if constexpr (EPI == EPI_GATE_RESIDUAL && STD) {
    // Interior tiles (all but the last tile row / column): in the MFMA layout a lane's 16 bytes of X sit in RT different rows
    // per instruction (64-byte pieces).  The wave parks y (rounded to bf16, as the reference's Linear output is) in its LDS
    // slice, one row tile at a time, and walks X row-wise instead: 4 rows x 256 contiguous bytes per load / store instruction,
    // the loads of 32 rows in flight together.  (LDS serves a wave's operations in order: no wait between the y writes and
    // the reads behind them.)
    if (m0 + BM_ <= p.M && n0 + BN_ <= p.N) {
      const int rr = le >> 4, cc_ = le & 15;           // row inside a group of 4, 16-byte piece of the wave's fp32 row (CW / 4 pieces)
      const int cc = cc_ < CW / 4 ? cc_ : CW / 4 - 1;   // (192-wide tiles: pieces 12..15 do not exist; their lanes shadow piece 11)
      const int mw = m0 + wm * (16 * MT);
      const int nw = n0 + wn * (16 * NTW) + cc * 4;
      // X addresses = uniform 64-bit row base (scalar registers) + ONE 32-bit per-lane offset: eight 64-bit per-lane pointers
      // next to the 128 accumulators are what used to push loop-carried values into scratch
      const uint32_t xlane = (uint32_t)((rr * p.ldx + wn * (16 * NTW) + cc * 4) * 4);
      char* xtile = (char*)(p.X + (int64_t)mw * p.ldx + n0);
      // y = bf16(acc + bias) for the whole wave tile first: 64 packed registers instead of 128 + bias while the X and gate loads
      // of 32 rows are in flight (the asm pins the conversion here; left to itself it sinks to the LDS writes and spills)
      bf16x4 yp[NRT][NV];
#pragma unroll
      for (int t = 0; t < NRT; ++t)
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          const f32x4 y4 = accv(t, v) + bias[v];
#pragma unroll
          for (int j = 0; j < 4; ++j) yp[t][v][j] = f2bf(y4[j]);
          asm volatile("" : "+v"(yp[t][v]));
        }
      // GATE: 0 no gate, 1 gate row from the per-row table, 2 gate row = m / rows_per_batch (one straight-line body each:
      // a uniform branch per load would keep the loads from being issued together)
      auto rmw = [&](auto gate_c, auto nt_c) {
        constexpr int GATE = decltype(gate_c)::value;
        constexpr bool NT = decltype(nt_c)::value;
        constexpr int TPB = 32 / RT;                   // row tiles per batch of 32 rows (8 load / store instructions)
        constexpr int NB = (NRT + TPB - 1) / TPB;      // batches per wave tile
        // TWO batches in flight: the X (and gate) loads of batch b + 1 are issued before batch b is consumed, so a wave pays
        // NB / 2 + 1 HBM round trips per tile instead of NB (the accumulators are packed into yp by now: the registers are there)
        f32x4 xv[2][8], gv[8];
        auto load_x = [&](int b, f32x4 (&xb)[8]) {
          const int t0 = b * TPB;
          const int ntb = NRT - t0 < TPB ? NRT - t0 : TPB;   // compile-time after unrolling
          const int nit = ntb * RT / 4;
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            if (i >= nit) continue;
            // X is 286 MB of fp32 at the DiT shapes -- larger than the Infinity Cache -- and every element is touched once per launch:
            // non-temporal loads and stores keep it from evicting the operands that ARE re-read (-0.8 ... -0.9 % of a step, profiles/r4as_*)
            const f32x4* xp_ = (const f32x4*)(xtile + (int64_t)(t0 * RT + 4 * i) * p.ldx * 4 + xlane);
            xb[i] = NT ? __builtin_nontemporal_load(xp_) : *xp_;
          }
        };
        // gate rows of the wave tile: in LDS since this unit's K block 0 (grow_dma)
        const int* growl = (const int*)(smem + GROW_OFF) + ((it & 1) * 8 + wave) * 128 + rr;
        // the gate values come from a small table (L2): one batch of them at a time, fetched when the previous batch has been consumed
        auto load_g = [&](int b) {
          if constexpr (GATE != 0) {
            const int t0 = b * TPB;
            const int ntb = NRT - t0 < TPB ? NRT - t0 : TPB;
            const int nit = ntb * RT / 4;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
              if (i >= nit) continue;
              const int row = t0 * RT + 4 * i;           // + rr
              int gr;
              if constexpr (GATE == 1) gr = growl[row];
              else gr = (mw + row + rr) / p.rows_per_batch;
              gv[i] = *(const f32x4*)(p.gate + (int64_t)gr * p.gate_ld + nw);
            }
          }
        };
        auto consume = [&](int b, const f32x4 (&xb)[8]) {
          const int t0 = b * TPB;
          const int ntb = NRT - t0 < TPB ? NRT - t0 : TPB;
#pragma unroll
          for (int u = 0; u < TPB; ++u) {
            if (u >= ntb) continue;
#pragma unroll
            for (int v = 0; v < NV; ++v) stg_write(v, yp[t0 + u][v]);
#pragma unroll
            for (int i = 0; i < RT / 4; ++i) {
              const int row = 4 * i + rr, idx = u * (RT / 4) + i;
              const bf16x4 y = *(const bf16x4*)(stg + row * 128 + (((cc >> 1) ^ (row & 7)) << 4) + (cc & 1) * 8);
              f32x4 x = xb[idx];
#pragma unroll
              for (int j = 0; j < 4; ++j) x[j] += GATE != 0 ? bf2f(y[j]) * gv[idx][j] : bf2f(y[j]);
              f32x4* xs_ = (f32x4*)(xtile + (int64_t)(t0 * RT + 4 * idx) * p.ldx * 4 + xlane);      // (shadow lanes: the same 16 bytes again)
              if constexpr (NT) __builtin_nontemporal_store(x, xs_);
              else *xs_ = x;
            }
          }
        };
        __builtin_amdgcn_sched_barrier(0);             // (behind the conversion of the accumulators: in front of it the loads do not fit)
        load_x(0, xv[0]);
        load_g(0);
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          if (b + 1 < NB) load_x(b + 1, xv[(b + 1) & 1]);
          __builtin_amdgcn_sched_barrier(0);           // the next batch's X loads are on their way before this one is consumed
          consume(b, xv[b & 1]);
          __builtin_amdgcn_sched_barrier(0);
          if (b + 1 < NB) load_g(b + 1);
        }
      };
      using NT1 = std::integral_constant<bool, true>;
      using NT0 = std::integral_constant<bool, false>;
      if (p.x_nt) {
        if (!p.gate) rmw(IC<0>{}, NT1{});
        else if (p.gate_row) rmw(IC<1>{}, NT1{});
        else rmw(IC<2>{}, NT1{});
      } else {
        if (!p.gate) rmw(IC<0>{}, NT0{});
        else if (p.gate_row) rmw(IC<1>{}, NT0{});
        else rmw(IC<2>{}, NT0{});
      }
      // every load of this epilogue has been consumed; saying so with an instruction the compiler's wait bookkeeping sees keeps
      // it from putting a vmcnt(0) of its own in front of the next unit's first register reuse (behind the stores)
      __builtin_amdgcn_s_waitcnt(0x0F70 | (PEND & 15) | ((PEND >> 4) << 14));
      pend = true;                                     // PEND = 4 MT stores per wave, issued after the next unit's K blocks
      continue;                                        // next tile of this persistent workgroup
    }
  }
  if constexpr (EPI != EPI_GATE_RESIDUAL && sizeof(OutT) == 2 && STD) {
    // Interior tiles, bf16 output: the MFMA layout gives a lane 4 columns of RT different rows, i.e. 32-byte pieces per store
    // instruction.  The wave turns its 16*MT x 64 outputs around in its LDS slice instead, one row tile at a time, and stores
    // whole 128-byte rows, 8 per instruction.
    if (m0 + BM_ <= p.M && n0 + BN_ <= p.N) {
      const int rd_row = le >> 3, rd_c_ = le & 7;
      const int rd_c = rd_c_ < CW / 8 ? rd_c_ : CW / 8 - 1;   // (192-wide tiles: 6 sixteen-byte pieces per wave row; lanes 6, 7 of a row shadow piece 5)
      bf16* crow = (bf16*)p.C + (int64_t)(m0 + wm * (16 * MT) + rd_row) * p.ldc + n0 + wn * (16 * NTW) + rd_c * 8;
#pragma unroll
      for (int t = 0; t < NRT; ++t) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          f32x4 y4 = accv(t, v) + bias[v];
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
        for (int i = 0; i < RT / 8; ++i) {
          const int row = 8 * i + rd_row;
          const bf16x8 o8 = *(const bf16x8*)(stg + row * 128 + ((rd_c ^ (row & 7)) << 4));
          // the GELU output (FFN1: 668 MB, read once by FFN2 from its start, when the Infinity Cache holds only its end) leaves non-temporally
          // as well (a further -0.2 ... -0.4 %); the plain outputs (QKV, cross-q) are re-read at once by the next launch and stay cached
          if constexpr (EPI == EPI_GELU) __builtin_nontemporal_store(o8, (bf16x8*)(crow + (int64_t)(t * RT + 8 * i) * p.ldc));
          else *(bf16x8*)(crow + (int64_t)(t * RT + 8 * i) * p.ldc) = o8;
        }
      }
      __builtin_amdgcn_s_waitcnt(0x0F70 | (PEND & 15) | ((PEND >> 4) << 14));   // see the gate-residual path
      pend = true;                                     // PEND = 2 MT stores per wave, issued after the next unit's K blocks
      continue;                                        // next tile of this persistent workgroup
    }
  }
  if constexpr (EPI == EPI_GATE_RESIDUAL) {
    // Read-modify-write in the accumulator layout (edge tiles; every tile of the 160-wide shape: the VAE's residual convolutions).
    // The X (and gate) loads of a whole row tile are issued together and one row tile ahead of the stores.  (Written as one load,
    // add, store per 16-byte unit, hipcc kept that order -- it cannot prove the units of different rows apart -- and put a
    // vmcnt(0) in front of every store: NRT * NV serial HBM round trips per tile, each also waiting for the store before it.)
    constexpr int DEPTH = (STD || BM_ > 320) ? 1 : 2;  // the 256-wide instances have no registers to spare next to 128 accumulators, the 384 x 160 one next to 120
    auto rmw_rows = [&](auto gate_c) {
      constexpr bool GATE = decltype(gate_c)::value;
      f32x4 xq[DEPTH][NV], gq[DEPTH][NV];
      int grr[NRT];                                    // gate-table row per row tile, all fetched up front (a fetch per row tile would be
      if constexpr (GATE) {                            // the youngest load in flight: waiting for it drains the queue)
#pragma unroll
        for (int t = 0; t < NRT; ++t) {
          const int m = min(mrow + t * RT, p.M - 1);
          grr[t] = p.gate_row ? p.gate_row[m] : (int)(m / p.rows_per_batch);
        }
      }
      // loads are unconditional (rows / columns past the edge re-read the last valid ones), only the stores are predicated
      auto ldt = [&](int t, f32x4 (&xb)[NV], f32x4 (&gb)[NV]) {
        const int m = min(mrow + t * RT, p.M - 1);
        const float* grow = nullptr;
        if constexpr (GATE) grow = p.gate + (int64_t)grr[t] * p.gate_ld;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          const int n = min(ncol + v * (4 * NG), p.N - 4);
          xb[v] = *(const f32x4*)(p.X + (int64_t)m * p.ldx + n);
          if constexpr (GATE) gb[v] = *(const f32x4*)(grow + n);
        }
      };
      if constexpr (DEPTH == 2) ldt(0, xq[0], gq[0]);
#pragma unroll
      for (int t = 0; t < NRT; ++t) {
        if constexpr (DEPTH == 2) {
          if (t + 1 < NRT) ldt(t + 1, xq[(t + 1) & 1], gq[(t + 1) & 1]);
        } else {
          ldt(t, xq[0], gq[0]);
        }
        __builtin_amdgcn_sched_barrier(0);
        const int m = mrow + t * RT;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          const int n = ncol + v * (4 * NG);
          const f32x4 y4 = accv(t, v) + bias[v];
          f32x4 x = xq[t % DEPTH][v];
          // y is rounded to bf16 first, like the reference's bf16 Linear output (FX.py:456,461,468)
#pragma unroll
          for (int j = 0; j < 4; ++j) x[j] += GATE ? bf2f(f2bf(y4[j])) * gq[t % DEPTH][v][j] : bf2f(f2bf(y4[j]));
          if (m < p.M && n < p.N) *(f32x4*)(p.X + (int64_t)m * p.ldx + n) = x;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    if (p.gate) rmw_rows(std::integral_constant<bool, true>{});
    else rmw_rows(std::integral_constant<bool, false>{});
    __builtin_amdgcn_s_waitcnt(0x0F70);                // vmcnt(0): an unknown number of stores, drained here (pend stays false)
    continue;
  }
#pragma unroll
  for (int t = 0; t < NRT; ++t) {
    const int m = mrow + t * RT;
    if (m >= p.M) continue;
    const float* grow = nullptr;
    if constexpr (EPI == EPI_GATE_RESIDUAL) {
      if (p.gate) {
        const int64_t r = p.gate_row ? (int64_t)p.gate_row[m] : (int64_t)m / p.rows_per_batch;
        grow = p.gate + r * p.gate_ld;
        asm volatile("" ::"v"(grow));                   // the gate_row load is "used" even if every column below is out of range
      }
    }
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int n = ncol + v * (4 * NG);
      if (n >= p.N) continue;
      f32x4 y4 = accv(t, v) + bias[v];
      if constexpr (EPI == EPI_GELU) {
#pragma unroll
        for (int j = 0; j < 4; ++j) y4[j] = gelu_tanh(y4[j]);
      }
      if constexpr (EPI == EPI_GATE_RESIDUAL) {
        // y is rounded to bf16 first, like the reference's bf16 Linear output (FX.py:456,461,468)
        float* xp = p.X + (int64_t)m * p.ldx + n;
        f32x4 x = *(const f32x4*)xp;
        f32x4 g = grow ? *(const f32x4*)(grow + n) : (f32x4){1.f, 1.f, 1.f, 1.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) x[j] += bf2f(f2bf(y4[j])) * g[j];
        *(f32x4*)xp = x;
      } else if constexpr (sizeof(OutT) == 2) {
        bf16x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = f2bf(y4[j]);
        *(bf16x4*)((bf16*)p.C + (int64_t)m * p.ldc + n) = o;
      } else {
        *(f32x4*)((float*)p.C + (int64_t)m * p.ldc + n) = y4;
      }
    }
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);                  // vmcnt(0): edge tiles / fp32 outputs issue an unknown number of stores, drained here (pend stays false)
  }   // !TAIL
  }   // tile loop
#ifdef FLEXAM_GEMM_STAMPS
  if constexpr (!TAIL) {
    __builtin_amdgcn_s_waitcnt(0x0F70);                // the epilogue's stores have left
    GEMM_STAMP(3);
  }
#endif
}

// Second half of the tail split-K: one workgroup per (tail tile, row tile t).  Thread `te` owns the same 16-byte elements the
// main kernel's thread `te` parked -- (row tile t, unit v) at ((t*NV+v)*512 + te)*4 of every slice's slab -- sums the slices in
// slice order and applies the epilogue (bias, GELU-tanh, fp32 gated residual, bf16 / fp32 store) in the accumulator layout.
template <int EPI, typename OutT, int MT, int WMW = 2, int NTW = 4>
__global__ __launch_bounds__(512) void gemm_splitk_finish_kernel(GemmParams p) {
  constexpr int WNW = 8 / WMW, BN_ = WNW * NTW * 16;
  constexpr int RT = 16, NRT = 16 * MT / RT, NG = 64 / RT, NV = NTW * 4 / NG, BM_ = WMW * 16 * MT, SLAB = 256 * BN;
  const int te = threadIdx.x, lane = te & 63, wave = te >> 6, wm = wave / WNW, wn = wave % WNW;
  const int tr = blockIdx.x / NRT, t = blockIdx.x % NRT;
  const int tile = p.split_full + tr;
  int m0, n0;
  {
    const int GM = p.gm, per_group = GM * p.tiles_n, group = tile / per_group, first_m = group * GM;
    const int gsz = min(p.tiles_m - first_m, GM), in_group = tile - group * per_group;
    m0 = (first_m + in_group % gsz) * BM_;
    n0 = (in_group / gsz) * BN_;
  }
  const int m = m0 + wm * (16 * MT) + t * RT + (lane & (RT - 1));
  if (m >= p.M) return;
  const float* base = p.ws + (int64_t)tr * p.split_s * SLAB;
  const float* grow = nullptr;
  if constexpr (EPI == EPI_GATE_RESIDUAL) {
    if (p.gate) grow = p.gate + (p.gate_row ? (int64_t)p.gate_row[m] : (int64_t)m / p.rows_per_batch) * p.gate_ld;
  }
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int n = n0 + wn * (16 * NTW) + (lane / RT) * 4 + v * (4 * NG);
    if (n >= p.N) continue;
    const int64_t e = ((int64_t)(t * NV + v) * 512 + te) * 4;
    f32x4 y4 = *(const f32x4*)(base + e);
    for (int s2 = 1; s2 < p.split_s; ++s2) y4 += *(const f32x4*)(base + (int64_t)s2 * SLAB + e);
    if (p.bias) y4 += *(const f32x4*)(p.bias + n);
    if constexpr (EPI == EPI_GELU) {
#pragma unroll
      for (int j = 0; j < 4; ++j) y4[j] = gelu_tanh(y4[j]);
    }
    if constexpr (EPI == EPI_GATE_RESIDUAL) {
      float* xp = p.X + (int64_t)m * p.ldx + n;
      f32x4 x = *(const f32x4*)xp;
      const f32x4 g = grow ? *(const f32x4*)(grow + n) : (f32x4){1.f, 1.f, 1.f, 1.f};
#pragma unroll
      for (int j = 0; j < 4; ++j) x[j] += bf2f(f2bf(y4[j])) * g[j];     // y rounded to bf16 first, like the main kernel's epilogue
      *(f32x4*)xp = x;
    } else if constexpr (sizeof(OutT) == 2) {
      bf16x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = f2bf(y4[j]);
      *(bf16x4*)((bf16*)p.C + (int64_t)m * p.ldc + n) = o;
    } else {
      *(f32x4*)((float*)p.C + (int64_t)m * p.ldc + n) = y4;
    }
  }
}

// scratch for the tail split-K, handed in by the caller with every launch (nothing is retained between calls):
// slabs of 256 x 256 fp32, no initialisation needed
struct GemmWorkspace {
  float* slabs = nullptr;
  int64_t n_slabs = 0;
};

int num_cus() { return flexam_num_cus(); }

// Tail split-K plan: `rem` = tiles of the last, partial round of the CUs.  Cutting each of them into S K slices turns that
// round into ceil(rem*S/G) passes of 1/S of a tile; every pass parks 256 KiB of partial sums per workgroup (~4 K blocks of main
// loop), and the finish launch costs a kernel boundary plus rem * S slabs read chip-wide (~4 + 0.03 * rem * S K blocks).  S (<= 8,
// slabs must fit the workspace) minimises the sum; with K = 3072 (48 K blocks) the hand-off eats most of the gain, with
// K = 14336 the tail shrinks to ~0.4 tile times.  `cost` = resulting length of the tail in tile times (1.0 without a split).
void plan_split(const GemmWorkspace& g_ws, int tiles, int nk, int& S, int& rem, double* cost = nullptr) {
  const int G = num_cus();
  static const int enabled = [] { const char* e = getenv("FLEXAM_GEMM_SPLITK"); return e ? atoi(e) : 1; }();
  rem = tiles % G;
  S = 1;
  double best = rem ? 1.0 : 0.0;
  if (enabled && g_ws.slabs && rem) {
    for (int s = 2; s <= 8 && s <= nk / 8 && (int64_t)rem * s <= g_ws.n_slabs; ++s) {
      const int passes = (rem * s + G - 1) / G;
      const double c = passes * (1.0 / s + 4.0 / nk) + (4.0 + 0.03 * rem * s) / nk;
      if (c < best - 0.05) { best = c; S = s; }
    }
  }
  if (cost) *cost = best;
}

template <int EPI, typename OutT, int MT, int WMW = 2, int NTW = 4>
int launch_shape(GemmParams p, const GemmWorkspace& g_ws, const int64_t* a_koff, hipStream_t s) {
  auto kern = gemm_bf16_kernel<EPI, OutT, MT, false, WMW, NTW>;
  static bool attr_set[FLEXAM_MAX_DEVICES] = {};          // per device: the attribute belongs to the device's copy of the code object
  // two K-block buffers (128 KiB) + one row tile of bf16 outputs per wave + two bias slots + two gate-row slots per wave; the tall
  // 160-wide shape: two buffers of [384 rows of A | 3 pieces of W] (144 KiB), no staging
  const int smem = WMW * 16 * MT > 256 ? 2 * (WMW * 16 * MT * 128 + ((8 / WMW) * NTW * 16 + 63) / 64 * 8192)
                                       : 4 * TILE_BYTES + 8 * 16 * 128 + 2 * 8 * 256 + 2 * 8 * 512;
  const int dev = flexam_current_device();
  if (!attr_set[dev]) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
      return flexam_fail(FLEXAM_E_LAUNCH, "gemm: cannot raise dynamic LDS to %d bytes", smem);
    attr_set[dev] = true;
  }
  p.tiles_m = (p.M + WMW * 16 * MT - 1) / (WMW * 16 * MT);
  p.tiles_n = (p.N + (8 / WMW) * NTW * 16 - 1) / ((8 / WMW) * NTW * 16);
  const int tiles = p.tiles_m * p.tiles_n;
  int split_s, rem;
  plan_split(g_ws, tiles, p.K / BK, split_s, rem);
  p.split_s = split_s;
  p.split_full = split_s > 1 ? tiles - rem : tiles;
  p.units = p.split_full + (split_s > 1 ? rem * split_s : 0);
  p.ws = g_ws.slabs;
  static const int persist = [] { const char* e = getenv("FLEXAM_GEMM_PERSIST"); return e ? atoi(e) : 1; }();
  auto grid_for_units = [&](int nwg) {
    int grid = (nwg + 7) / 8 * 8;                        // a multiple of 8 so that blockIdx & 7 is the XCD
    if (persist && grid > num_cus()) grid = num_cus();   // one persistent workgroup per CU (156 of its 160 KiB of LDS)
    return grid;
  };
  if (p.split_full > 0) hipLaunchKernelGGL(kern, dim3(grid_for_units(p.split_full)), dim3(512), smem, s, p, a_koff);
  if (split_s > 1) {
    // the K slices of the tail tiles, then (stream-ordered) their sum in slice order + the epilogue
    auto tail = gemm_bf16_kernel<EPI_NONE, float, MT, true, WMW, NTW>;
    static bool tail_attr[FLEXAM_MAX_DEVICES] = {};
    if (!tail_attr[dev]) {
      if (hipFuncSetAttribute((const void*)tail, hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
        return flexam_fail(FLEXAM_E_LAUNCH, "gemm: cannot raise dynamic LDS to %d bytes", smem);
      tail_attr[dev] = true;
    }
    hipLaunchKernelGGL(tail, dim3(grid_for_units(rem * split_s)), dim3(512), smem, s, p, a_koff);
    hipLaunchKernelGGL((gemm_splitk_finish_kernel<EPI, OutT, MT, WMW, NTW>), dim3(rem * MT), dim3(512), 0, s, p);
  }
  return flexam_check_launch("flexam_gemm_bf16");
}

template <int EPI, typename OutT, int MT>
int launch_mt(const GemmParams& p, const GemmWorkspace& g_ws, const int64_t* a_koff, hipStream_t s) {
  return launch_shape<EPI, OutT, MT>(p, g_ws, a_koff, s);
}

// Tile height: rounds of 256 concurrently resident workgroups x relative cost of one tile (MT m-tiles of MFMA work
// plus a fixed part for the W side, barriers and the epilogue); FLEXAM_GEMM_MT=8..4 forces a shape (tuning only).
int pick_mt(const GemmWorkspace& g_ws, int M, int tiles_n, int nk) {
  const char* e = getenv("FLEXAM_GEMM_MT");
  const int forced = e ? atoi(e) : 0;
  if (forced >= 4 && forced <= 8) return forced;
  int best = 8;
  double best_cost = 1e30;
  const int G = num_cus();
  for (int mt = 8; mt >= 4; --mt) {
    const int tiles = (int)((long)((M + 32 * mt - 1) / (32 * mt)) * tiles_n);
    int S, rem;
    double tail;
    plan_split(g_ws, tiles, nk, S, rem, &tail);
    const double cost = (tiles / G + tail) * (mt + 1.25);
    if (cost < best_cost * 0.97) { best_cost = cost; best = mt; }      // a smaller tile must win by > 3 %
  }
  return best;
}

// caller's scratch -> workspace view; too small or NULL = no split-K
GemmWorkspace make_ws(void* ws, int64_t bytes) {
  GemmWorkspace g;
  if (ws && bytes >= (int64_t)256 * BN * 4) {
    g.slabs = (float*)ws;
    g.n_slabs = bytes / ((int64_t)256 * BN * 4);
  }
  return g;
}

template <int EPI, typename OutT>
int launch(const GemmParams& p_, void* ws, int64_t ws_bytes, const int64_t* a_koff, hipStream_t s) {
  GemmParams p = p_;
  const GemmWorkspace g_ws = make_ws(ws, ws_bytes);
  {
    // Tile-rows per L2 group.  Plain-store epilogues: 4 tile-rows x 8 tile-columns resident per XCD is the smallest panel set
    // (12 panels of 32 KiB per K block) and measured best (profiles/r1e notes; r3 round-robin over 1 / 2 / 4 / 8 / 16:
    // profiles/r3d_gemm_gm_ab.txt).  Gate-residual epilogues read-modify-write 1 KiB of an X row per tile: there the resident
    // tiles' X footprint matters more than panel sharing with a long K -- one tile-row across all columns (whole 12 KiB rows of X):
    // 3 % faster than 4 at K = 14336.  With a short K (o-proj) r3 measured a tall group of 16 ahead; with the r5 epilogue (two
    // batches of X in flight) 4 ... 12 are level and 16 is 0.6 % behind (profiles/r5zb_gemm_gm_sweep.txt): 4 like the plain stores.
    const char* g = getenv("FLEXAM_GEMM_GM");
    p.gm = g ? atoi(g) : (EPI == EPI_GATE_RESIDUAL && a_koff == nullptr && p.K >= 8192 ? 1 : 4);
    if (p.gm < 1) p.gm = 4;
  }
  if constexpr (EPI == EPI_GATE_RESIDUAL) {
    // X (fp32 residual stream) with non-temporal hints only when it cannot stay in the 256 MB Infinity Cache anyway (the single-GPU CFG pair:
    // 286 MB, -0.8 % of a step with the hints, r4as); a sequence-parallel rank's 36-72 MB is touched six times per block and should stay cached.
    // FLEXAM_GEMM_X_NT=0 / 1 forces (A/B).
    const char* e = getenv("FLEXAM_GEMM_X_NT");
    p.x_nt = e ? atoi(e) : ((int64_t)p.M * p.N * 4 > (int64_t)160 << 20);
  }
#ifdef FLEXAM_GEMM_ABLATE
  const char* dbg = getenv("FLEXAM_GEMM_DEBUG");
  p.debug = dbg ? atoi(dbg) : 0;
#endif
  // output widths that are multiples of 160 but not of 256 (VAE encoder: 160, 320 channels): the 256 x 160 shape wastes nothing,
  // a 256-wide tile 37.5 %; FLEXAM_GEMM_N160=0 keeps the 256-wide shape (A/B), =2 also takes N = 640 (83 % of 256-wide tiles)
  {
    const char* e = getenv("FLEXAM_GEMM_N160");
    const int mode = e ? atoi(e) : 1;
    const bool narrow = p.N <= 160;                      // one 160-wide tile column instead of a 256-wide one (VAE head convs: 12 / 96 channels)
    const bool mult160 = p.N % 160 == 0 && p.N % 256 != 0 && (p.N <= 480 || mode == 2);
    if (mode && (narrow || mult160)) {
      // 384-row tiles where the rows fill many rounds of the CUs (the encoder's 160-channel convolutions: millions of rows): a third
      // more MFMAs per fragment read and per staged byte; FLEXAM_GEMM_N160_TALL=0 keeps 256 rows (A/B)
      static const int tall = [] { const char* t = getenv("FLEXAM_GEMM_N160_TALL"); return t ? atoi(t) : 1; }();
      const long tiles_tall = (long)((p.M + 383) / 384) * ((p.N + 159) / 160);
      // (the read-modify-write epilogue next to 120 accumulators spills: that instance runs 320-row tiles, 100 accumulators)
      if (tall && tiles_tall >= 4L * num_cus()) return launch_shape<EPI, OutT, EPI == EPI_GATE_RESIDUAL ? 5 : 6, 4, 5>(p, g_ws, a_koff, s);
      return launch_shape<EPI, OutT, 4, 4, 5>(p, g_ws, a_koff, s);
    }
  }
  // 192 x 192 tiles (MT = 6, 3 n-tiles per wave) where their rounds of the CUs beat the best 256-wide plan: output widths that are
  // multiples of 192 at row counts where 256-wide tiles quantise badly -- a rank-of-eight's 2912 x 3072 launches are exactly 16 x 16 =
  // 256 tiles (one full round, 94.8 % useful) against 228 tiles of 160 x 256.  Relative cost of a 192 x 192 tile: 18 of the 256-wide
  // MT = 6 tile's 24 MFMAs per K half and wave, the same 9 fragment reads, 6 of 7 staging pieces: (0.75 * 6 + 1.1) on pick_mt's scale.
  // FLEXAM_GEMM_N192 = 0 keeps the 256-wide shapes (A/B), = 2 forces the 192-wide one wherever N % 192 == 0.
  if (p.N % 192 == 0) {
    const char* e = getenv("FLEXAM_GEMM_N192");
    const int mode = e ? atoi(e) : 1;
    bool take = mode == 2;
    if (mode == 1) {
      const int G = num_cus(), nk = p.K / BK;
      double best256 = 1e30;
      for (int mt = 8; mt >= 4; --mt) {
        const int tiles = (int)((long)((p.M + 32 * mt - 1) / (32 * mt)) * p.tiles_n);
        int S, rem;
        double tail;
        plan_split(g_ws, tiles, nk, S, rem, &tail);
        best256 = fmin(best256, (tiles / G + tail) * (mt + 1.25));
      }
      const int tiles192 = (int)((long)((p.M + 191) / 192) * (p.N / 192));
      int S, rem;
      double tail;
      plan_split(g_ws, tiles192, nk, S, rem, &tail);
      take = (tiles192 / G + tail) * (0.75 * 6 + 1.1) < 0.95 * best256;
    }
    if (take) return launch_shape<EPI, OutT, 6, 2, 3>(p, g_ws, a_koff, s);
  }
  switch (pick_mt(g_ws, p.M, p.tiles_n, p.K / BK)) {
    case 7: return launch_mt<EPI, OutT, 7>(p, g_ws, a_koff, s);
    case 6: return launch_mt<EPI, OutT, 6>(p, g_ws, a_koff, s);
    case 5: return launch_mt<EPI, OutT, 5>(p, g_ws, a_koff, s);
    case 4: return launch_mt<EPI, OutT, 4>(p, g_ws, a_koff, s);
    default: return launch_mt<EPI, OutT, 8>(p, g_ws, a_koff, s);
  }
}

}  // namespace

#ifdef FLEXAM_GEMM_STAMPS
// diagnostic builds only (not declared in flexam_hip.h): the per-workgroup stamps of the last whole-tile GEMM launch
extern "C" int flexam_debug_gemm_stamps(unsigned long long* out, int n_workgroups) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_gemm_stamps), (size_t)n_workgroups * 32) == hipSuccess ? 0 : -1;
}
#endif

extern "C" int flexam_gemm_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, void* C,
                                int64_t ldc, int64_t M, int64_t N, int64_t K, int epilogue, int out_f32,
                                const int64_t* a_koff, void* ws, int64_t ws_bytes, void* stream) {
  FX_REQUIRE(A && W && C, FLEXAM_E_ARG, "gemm: null pointer");
  FX_REQUIRE(M > 0 && N > 0 && K > 0, FLEXAM_E_SHAPE, "gemm: empty problem M=%ld N=%ld K=%ld", (long)M, (long)N, (long)K);
  FX_REQUIRE(K % BK == 0, FLEXAM_E_SHAPE, "gemm: K=%ld must be a multiple of %d (pad on the host)", (long)K, BK);
  FX_REQUIRE(N % 4 == 0 && ldc % 4 == 0, FLEXAM_E_SHAPE, "gemm: N=%ld and ldc=%ld must be multiples of 4", (long)N, (long)ldc);
  FX_REQUIRE(lda % 8 == 0 && ldw % 8 == 0, FLEXAM_E_SHAPE, "gemm: lda/ldw must be multiples of 8 elements (16-byte rows)");
  FX_REQUIRE(((uintptr_t)A | (uintptr_t)W | (uintptr_t)C | (uintptr_t)ws) % 16 == 0, FLEXAM_E_ARG, "gemm: pointers must be 16-byte aligned");
  FX_REQUIRE(epilogue == EPI_NONE || epilogue == EPI_GELU, FLEXAM_E_ARG, "gemm: unknown epilogue %d", epilogue);
  GemmParams p{};
  p.A = (const bf16*)A; p.W = (const bf16*)W; p.C = C; p.bias = bias;
  p.lda = lda; p.ldw = ldw; p.ldc = ldc; p.M = (int)M; p.N = (int)N; p.K = (int)K;
  p.tiles_n = (int)((N + BN - 1) / BN);   // tiles_m depends on the tile height launch() picks
  hipStream_t s = (hipStream_t)stream;
  if (out_f32) {
    FX_REQUIRE(epilogue == EPI_NONE, FLEXAM_E_ARG, "gemm: f32 output supports no activation epilogue");
    return launch<EPI_NONE, float>(p, ws, ws_bytes, a_koff, s);
  }
  return epilogue == EPI_GELU ? launch<EPI_GELU, bf16>(p, ws, ws_bytes, a_koff, s) : launch<EPI_NONE, bf16>(p, ws, ws_bytes, a_koff, s);
}

extern "C" int flexam_gemm_bf16_gate_residual(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias,
                                              float* X, int64_t ldx, const float* gate, int64_t gate_ld,
                                              const int32_t* gate_row, int64_t rows_per_batch, int64_t M, int64_t N,
                                              int64_t K, const int64_t* a_koff, void* ws, int64_t ws_bytes, void* stream) {
  FX_REQUIRE(A && W && X, FLEXAM_E_ARG, "gemm_gate_residual: null pointer");
  FX_REQUIRE(M > 0 && N > 0 && K > 0, FLEXAM_E_SHAPE, "gemm_gate_residual: empty problem");
  FX_REQUIRE(K % BK == 0 && N % 4 == 0 && ldx % 4 == 0, FLEXAM_E_SHAPE, "gemm_gate_residual: K%%64, N%%4, ldx%%4 required");
  FX_REQUIRE(lda % 8 == 0 && ldw % 8 == 0, FLEXAM_E_SHAPE, "gemm_gate_residual: lda/ldw must be multiples of 8");
  FX_REQUIRE(!gate || gate_row || rows_per_batch > 0, FLEXAM_E_ARG, "gemm_gate_residual: gate needs gate_row or rows_per_batch");
  GemmParams p{};
  p.A = (const bf16*)A; p.W = (const bf16*)W; p.C = nullptr; p.bias = bias;
  p.lda = lda; p.ldw = ldw; p.ldc = 0; p.M = (int)M; p.N = (int)N; p.K = (int)K;
  p.tiles_n = (int)((N + BN - 1) / BN);   // tiles_m depends on the tile height launch() picks
  p.X = X; p.ldx = ldx; p.gate = gate; p.gate_ld = gate_ld; p.gate_row = gate_row;
  p.rows_per_batch = rows_per_batch > 0 ? rows_per_batch : 1;
  FX_REQUIRE((uintptr_t)ws % 16 == 0, FLEXAM_E_ARG, "gemm_gate_residual: workspace must be 16-byte aligned");
  return launch<EPI_GATE_RESIDUAL, bf16>(p, ws, ws_bytes, a_koff, (hipStream_t)stream);
}
