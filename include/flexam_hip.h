/* include/flexam_hip.h -- C ABI of libflexam_hip.so: the MI355X (gfx950) kernels behind the
 * FlexAM denoising hot path (Wan2.2-Fun-5B-FLEXAM DiT forward, sampler step, Wan2.2 3D-VAE
 * decode).  This is the drop-in boundary: plain pointers and sizes, no torch types.
 *
 * Conventions
 *   - every function returns 0 (FLEXAM_OK) or a negative FLEXAM_E_* code; flexam_last_error()
 *     returns the message of the last failure on the calling thread;
 *   - all pointers are DEVICE pointers unless a comment says host; the caller (PyTorch-ROCm in
 *     flexam_amd/hip.py) owns every buffer, the library never allocates, frees or retains one:
 *     scratch (GEMM split-K slabs, split-KV attention partials) is an argument of the call that uses it;
 *   - `stream` is a hipStream_t; calls are asynchronous and ordered on it, no hidden syncs,
 *     safe to capture into a hipGraph; calls on different streams / devices are independent as long
 *     as they are given different scratch buffers (the only process-global state is a per-device
 *     "kernel attribute set" flag table and the per-device CU count);
 *   - bf16 = IEEE bfloat16 (torch.bfloat16); "f32 table row" arguments are row pointers of
 *     small fp32 tables selected per token through an int32 row index ("two-row
 *     modulation").
 * Each entry point cites the reference code it replaces (paths relative to the FlexAM repo).
 */
#ifndef FLEXAM_HIP_H
#define FLEXAM_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FLEXAM_HIP_VERSION 2
#define FLEXAM_OK 0
#define FLEXAM_E_ARG (-1)
#define FLEXAM_E_SHAPE (-2)
#define FLEXAM_E_ARCH (-3)
#define FLEXAM_E_LAUNCH (-4)

#define FLEXAM_EPI_NONE 0
#define FLEXAM_EPI_GELU_TANH 1

int flexam_version(void);
const char* flexam_arch(void);          /* "gfx950" */
const char* flexam_last_error(void);
int flexam_device_check(void);        /* FLEXAM_E_ARCH unless the current device is gfx950 */

/* Compute units of the current device that the library plans its grids for (the device's CU count rounded down to a multiple
 * of 8: one persistent workgroup per CU, blockIdx & 7 = XCD; 256 on MI355X).  Callers that plan work units themselves -- the
 * key split of flexam_attn_fwd_splitkv -- use the same figure. */
int flexam_device_cus(void);
/* Plan the persistent grids of the current device for `cus` compute units (a multiple of 8: blockIdx & 7 stays the XCD; 0 = all of them,
 * the default) -- flexam_device_cus() answers with the budget from then on.  For callers that run collectives BESIDE these kernels: a
 * one-workgroup-per-CU kernel on every CU leaves a collective's own kernels nowhere to run until it ends (flexam_amd/dit_engine.py sets 8
 * CUs aside under the overlapped sequence-parallel exchanges).  Process-global per device, not stream-ordered: set it between launches. */
int flexam_set_cu_budget(int cus);

/* C[M,N] = epi(A[M,K] . W[N,K]^T + bias[N]); A, W bf16 row-major with K contiguous (nn.Linear
 * weight layout), fp32 accumulate on MFMA; C bf16 (out_f32 = 0) or fp32 (out_f32 = 1).
 * K % 64 == 0, N % 4 == 0, lda/ldw % 8 == 0 (pad on the host).  a_koff (optional, [K/64] int64,
 * device): element offset added to every A row base for K block kb instead of kb*64 -- the
 * implicit-GEMM form of the VAE's causal convolutions (one entry per (tap, 64-channel slice)).
 * ws / ws_bytes (optional): scratch for the tail split-K -- the tiles of the last, partial round of the CUs are cut along K so
 * that all CUs work; the slices' partial sums are parked in `ws` and a second launch on the same stream adds them up in slice
 * order (deterministic) and runs the epilogue.  Device memory, 16-byte aligned, slabs of 256 KiB, no initialisation needed;
 * FLEXAM_GEMM_WS_BYTES covers every shape.  NULL / too small: the GEMM never splits.  The buffer is used only by this call
 * (stream-ordered): calls that may run concurrently (other streams, other devices) need their own.  (The Python binding keeps
 * one 64 MiB buffer per (device, stream) in an LRU of 8 slots, flexam_amd/hip.py:_gemm_workspace: that is the per-stream cost of
 * using the library from several streams.)
 * Replaces nn.Linear -> cuBLAS (+ its workspace): FlexAM/models/wan_transformer3d_FlexAM.py:242-244,261,363-365,
 * 370,415-416,487,626-636 and Conv3d/Conv2d -> cuDNN: FlexAM/models/wan_vae3_8.py:39-47,94,99. */
#define FLEXAM_GEMM_WS_BYTES (256 * 262144)
int flexam_gemm_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, void* C, int64_t ldc,
                     int64_t M, int64_t N, int64_t K, int epilogue, int out_f32, const int64_t* a_koff, void* ws,
                     int64_t ws_bytes, void* stream);

/* X[m,n] += bf16(A.W^T + bias)[m,n] * gate[row(m), n]   (fp32 residual stream updated in place)
 * row(m) = gate_row[m] if gate_row else m / rows_per_batch; gate == NULL means gate = 1.
 * Replaces Linear + `x = x + y * e[2]` / `x + cross_attn(...)` / `x + y * e[5]`:
 * FlexAM/models/wan_transformer3d_FlexAM.py:261+456, 370+461, 416+468; with a_koff (see flexam_gemm_bf16)
 * also the second conv + skip of the VAE ResidualBlock, FlexAM/models/wan_vae3_8.py:213,240. */
int flexam_gemm_bf16_gate_residual(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, float* X,
                                   int64_t ldx, const float* gate, int64_t gate_ld, const int32_t* gate_row,
                                   int64_t rows_per_batch, int64_t M, int64_t N, int64_t K, const int64_t* a_koff,
                                   void* ws, int64_t ws_bytes, void* stream);

/* fp8 variant of the two GEMMs above for the DiT's QKV / FFN projections (BASELINE.json configs[4]; selected explicitly, never
 * the default).  A8 [M,K], W8 [N,K]: OCP e4m3 bytes, K contiguous, K % 128 == 0, lda / ldw % 16 == 0; a_scale [M], w_scale [N]
 * fp32 row scales (value = byte * scale); fp32 accumulation on the block-scaled MFMA with unit block scales:
 *   C = epi((A8 . W8^T) * a_scale[m] * w_scale[n] + bias[n])   (bf16 out; epilogue NONE | GELU_TANH)
 *   X[m,n] += bf16(...)[m,n] * gate[row(m), n]                  (gate-residual form, as flexam_gemm_bf16_gate_residual)
 * flexam_quantize_rows_fp8: q[m,:] = e4m3(x[m,:] / scale[m]), scale[m] = absmax(x[m,:]) / 448 (bf16 in); used for activations per
 * call and for weights once.  The reference only STORES weights as float8_e4m3fn and upcasts them per call
 * (FlexAM/utils/fp8_optimization.py:1-57); fp8 arithmetic is this build's addition. */
int flexam_quantize_rows_fp8(const void* x, int64_t ldx, void* q, int64_t ldq, float* scale, int64_t M, int K, void* stream);
int flexam_gemm_fp8(const void* A, int64_t lda, const float* a_scale, const void* W, int64_t ldw, const float* w_scale,
                    const float* bias, void* C, int64_t ldc, int64_t M, int64_t N, int64_t K, int epilogue, void* stream);
int flexam_gemm_fp8_gate_residual(const void* A, int64_t lda, const float* a_scale, const void* W, int64_t ldw, const float* w_scale,
                                  const float* bias, float* X, int64_t ldx, const float* gate, int64_t gate_ld,
                                  const int32_t* gate_row, int64_t rows_per_batch, int64_t M, int64_t N, int64_t K, void* stream);
/* The fp8 FFN without a quantise pass between its two GEMMs: Q[m,n] = e4m3(gelu_tanh((A8 . W8^T) * a_scale[m] * w_scale[n] + bias[n])
 * / out_scale[m]) -- the A operand of the next flexam_gemm_fp8* call with a_scale = out_scale.  out_scale [M] comes from
 * flexam_ln_modulate_fp8 (next_scale: a bound on the row's outputs, so nothing saturates).  Q: e4m3 bytes, ldq % 16 == 0. */
int flexam_gemm_fp8_gelu_q(const void* A, int64_t lda, const float* a_scale, const void* W, int64_t ldw, const float* w_scale,
                           const float* bias, const float* out_scale, void* Q, int64_t ldq, int64_t M, int64_t N, int64_t K,
                           void* stream);

/* Flash attention forward, head_dim 128, non-causal, keys [0, Lk): o = softmax(q k^T * scale) v.
 * q/k/v/o are [B, L, H, 128] views given by batch stride `*_bs` and row stride `*_rs` (elements);
 * head h starts at column h*128 of a row.  bf16 in/out, fp32 softmax/accumulate.
 * Replaces attention() -> flash_attn / sageattn / SDPA: FlexAM/models/attention_utils.py:43-233
 * (call sites wan_transformer3d_FlexAM.py:251-256 self, :367 cross).
 * softmax_scale > 0: the scale of the formula.  softmax_scale = FLEXAM_ATTN_PRESCALED: the producer of q already multiplied it
 * by scale * log2(e) before its one rounding to bf16 (the DiT engine folds that factor into the RMSNorm weight of q), so the
 * scores are in exp2 units and the kernel saves one FMA per score: o = softmax2(q k^T) v with softmax2 built on 2^x. */
#define FLEXAM_ATTN_PRESCALED (-1.0f)
int flexam_attn_fwd(const void* q, int64_t q_bs, int64_t q_rs, const void* k, int64_t k_bs, int64_t k_rs, const void* v,
                    int64_t v_bs, int64_t v_rs, void* o, int64_t o_bs, int64_t o_rs, int B, int H, int Lq, int Lk,
                    int head_dim, float softmax_scale, void* stream);
/* flexam_attn_fwd with the LAST key counted `last_key_multiplicity` (>= 1) times: softmax over {k_0 .. k_{Lk-2}, N copies of
 * k_{Lk-1}} -- exactly what attention over a context whose trailing rows are identical computes, at the cost of one copy
 * (the copy's score gets + log2 N before the exponential).  The reference pads every prompt to text_len = 512 rows with
 * zeros BEFORE the text-embedding MLP and attends to all of them (wan_transformer3d_FlexAM.py:958-964, context_lens = None at
 * :367): the 386+ padded rows of a 126-token prompt are one and the same K/V row. */
int flexam_attn_fwd_lastkey(const void* q, int64_t q_bs, int64_t q_rs, const void* k, int64_t k_bs, int64_t k_rs, const void* v,
                            int64_t v_bs, int64_t v_rs, void* o, int64_t o_bs, int64_t o_rs, int B, int H, int Lq, int Lk,
                            int head_dim, float softmax_scale, float last_key_multiplicity, void* stream);
/* Same, with the work units u = (batch*H + head)*ceil(Lq/256) + q_block at and after `split_from_unit` cut into kv_splits key
 * ranges each (flash-decoding style partial softmaxes + a merge launch), the units before it in one pass: with
 * split_from_unit = floor(units/256)*256 only the last, partial round of the 256 CUs is split; 0 splits everything (ranks of a
 * sequence-parallel run hold few query rows).  ws_o: fp32 [kv_splits, n, 256, 128], ws_ml: fp32 [kv_splits, n, 256, 2] scratch
 * with n = units - split_from_unit.  kv_splits = 1 == flexam_attn_fwd.  Whole and split units go out as ONE kernel launch (each XCD
 * gets an eighth of both kinds, its whole units first: the short workgroups start as its CUs finish their last whole unit),
 * followed by the merge launch. */
int flexam_attn_fwd_splitkv(const void* q, int64_t q_bs, int64_t q_rs, const void* k, int64_t k_bs, int64_t k_rs, const void* v,
                            int64_t v_bs, int64_t v_rs, void* o, int64_t o_bs, int64_t o_rs, int B, int H, int Lq, int Lk,
                            int head_dim, float softmax_scale, int kv_splits, int split_from_unit, float* ws_o, float* ws_ml,
                            void* stream);

/* Attention over a PART of the keys, finished later: flexam_attn_fwd_partial leaves, for every query row, the un-normalised
 * output and (reference, row sum) of the softmax over the keys it was given -- cut into kv_splits ranges, workspace slots slot0 ..
 * slot0 + kv_splits - 1 (kv_splits must equal ceil(tiles / ceil(tiles / kv_splits)), tiles = ceil(Lk / 64)); any number of such
 * calls on disjoint key sets may fill the slots, in any order; flexam_attn_merge combines n_slots of them into o.  Lets a
 * sequence-parallel rank attend to its LOCAL K/V chunk while the all-gather of the other ranks' chunks is still in flight
 * (the exchange the reference delegates to the missing FlexAM/dist, wan_transformer3d_FlexAM.py:801-815).
 * ws_o: fp32 [n_slots, units, 256, 128], ws_ml: fp32 [n_slots, units, 256, 2], units = B * H * ceil(Lq / 256). */
int flexam_attn_fwd_partial(const void* q, int64_t q_bs, int64_t q_rs, const void* k, int64_t k_bs, int64_t k_rs, const void* v,
                            int64_t v_bs, int64_t v_rs, int B, int H, int Lq, int Lk, int head_dim, float softmax_scale,
                            int kv_splits, int slot0, float* ws_o, float* ws_ml, void* stream);
int flexam_attn_merge(void* o, int64_t o_bs, int64_t o_rs, int B, int H, int Lq, int head_dim, float softmax_scale, int n_slots,
                      const float* ws_o, const float* ws_ml, void* stream);

/* Self-attention with MXFP8 operands (OCP e4m3 + one E8M0 scale per 32 elements; both products on
 * v_mfma_scale_f32_32x32x64_f8f6f4 at twice the bf16 MFMA rate, fp32 softmax and accumulation) -- the quantised variant behind
 * VIDEOX_ATTENTION_TYPE=SAGE_ATTENTION, FlexAM/models/attention_utils.py:195-203 (the reference calls the third-party
 * `sageattn`, which quantises Q, K and the P.V product; kept here: its contract, softmax attention within a stated tolerance).
 * flexam_attn_fp8_pack: q, k, v bf16 [B, L, H, 128] as flexam_attn_fwd takes them (q carrying softmax_scale * log2 e, i.e. the
 * FLEXAM_ATTN_PRESCALED form) -> q8 [B][H][Lp][128] e4m3, qs [B][H][Lp] four E8M0 bytes per row, kv8 [B][H][T] records of 18432
 * bytes (K tile image, V^T tile image with the key order the P.V operand needs, their scales), Lp = ceil(L / 256) * 256,
 * T = ceil(L / 64); rows past L are written as zeros; q = k = NULL: only the V half of the records is written.  flexam_attn_fwd_fp8: o [B, L, H, 128] bf16 from those buffers; kv_splits /
 * split_from_unit / ws_o / ws_ml as flexam_attn_fwd_splitkv (kv_splits = 1: no workspace). */
int flexam_attn_fp8_pack(const void* q, int64_t q_bs, int64_t q_rs, const void* k, int64_t k_bs, int64_t k_rs, const void* v,
                         int64_t v_bs, int64_t v_rs, void* q8, void* qs, void* kv8, int B, int H, int L, int head_dim, void* stream);
/* The producer of those operands inside the DiT: RMSNorm over the full width + RoPE of q and k (flexam_rmsnorm_rope's arithmetic,
 * q's weight carrying softmax_scale * log2 e) written as q8 / qs and the K image + K scales of the kv8 records directly, quantised
 * from fp32; 24 heads of 128 channels; M = B * tokens_per_batch rows.  The buffers' padding rows (past L) must have been zeroed once
 * (they are never written here); the V half of the records then comes from flexam_attn_fp8_pack with q = k = NULL. */
int flexam_rmsnorm_rope_mx(const void* q, int64_t ldq, const float* wq, const void* k, int64_t ldk, const float* wk, void* q8, void* qs,
                           void* kv8, int64_t M, int C, float eps, const float* rope_cos, const float* rope_sin,
                           int64_t tokens_per_batch, int64_t token_offset, int H, int head_dim, void* stream);
int flexam_attn_fwd_fp8(const void* q8, const void* qs, const void* kv8, void* o, int64_t o_bs, int64_t o_rs, int B, int H, int L,
                        int head_dim, int kv_splits, int split_from_unit, float* ws_o, float* ws_ml, void* stream);
/* The same kernel with Lq != Lk and the key records in CHUNKS of `chunk_tiles` 64-key tiles: kv8 = [n_chunks][B][H][chunk_tiles] records,
 * n_chunks = ceil(ceil(Lk / 64) / chunk_tiles) -- the rank-major result of all-gathering every rank's own records under sequence
 * parallelism (each rank's token chunk a multiple of 64 tokens: MXFP8 K|V travel instead of bf16 rows, 288 instead of 512 bytes per key
 * and head; the missing exchange of FlexAM/dist, wan_transformer3d_FlexAM.py:801-815, for VIDEOX_ATTENTION_TYPE=SAGE_ATTENTION).
 * q8 / qs: [B][H][ceil(Lq / 256) * 256] rows as flexam_attn_fp8_pack writes them for L = Lq.  Keys >= Lk are masked. */
int flexam_attn_fwd_fp8_chunked(const void* q8, const void* qs, const void* kv8, void* o, int64_t o_bs, int64_t o_rs, int B, int H,
                                int Lq, int Lk, int chunk_tiles, int head_dim, int kv_splits, int split_from_unit, float* ws_o,
                                float* ws_ml, void* stream);

/* out_bf16[m,:] = LN(x_f32[m,:]; eps) [* ln_w + ln_b] [* scale[row(m),:] + shift[row(m),:]]
 * row(m) = row_index[m] if row_index else m / rows_per_batch; scale rows already hold (1+scale),
 * shift rows already hold shift + density shift (flexam_mod_table).  Replaces WanLayerNorm +
 * modulate: FlexAM/models/wan_transformer3d_FlexAM.py:192-202,452-453,461(norm3),464-465,506. */
int flexam_ln_modulate(const float* x, int64_t ldx, int64_t M, int C, float eps, const float* shift, const float* scale,
                       int64_t tab_ld, const int32_t* row_index, int64_t rows_per_batch, const float* ln_w,
                       const float* ln_b, void* out, int64_t ldo, void* stream);

/* The same row operation with the output quantised for flexam_gemm_fp8: q_out[m,:] = e4m3(y / row_scale[m]), row_scale[m] =
 * absmax(y[m,:]) / 448 (C a multiple of 512, at most 4096): the fp8 variant's QKV / FFN1 inputs without a second pass. */
/* next_scale (optional, [M]): a scale for the e4m3 OUTPUT of the GEMM + GELU this row feeds (flexam_gemm_fp8_gelu_q), known before
 * that GEMM runs: next_scale[m] = (1.07 |y[m,:]|_2 * next_wnorm + next_bias) / 448 with next_wnorm >= max_j |w_j|_2 of the
 * (dequantised) weight rows and next_bias >= max_j |b_j| -- by Cauchy-Schwarz no output element exceeds 448 next_scale[m]. */
int flexam_ln_modulate_fp8(const float* x, int64_t ldx, int64_t M, int C, float eps, const float* shift, const float* scale,
                           int64_t tab_ld, const int32_t* row_index, int64_t rows_per_batch, const float* ln_w, const float* ln_b,
                           void* q_out, int64_t ldq, float* row_scale, float* next_scale, float next_wnorm, float next_bias,
                           void* stream);

/* x_f32[m,:] += y_bf16[m,:] * gate[row(m),:] (gate NULL = 1).  wan_transformer3d_FlexAM.py:456,461,468. */
int flexam_gate_residual(float* x, int64_t ldx, const void* y, int64_t ldy, const float* gate, int64_t gate_ld,
                         const int32_t* row_index, int64_t rows_per_batch, int64_t M, int C, void* stream);

/* WanRMSNorm over the full row (all heads) followed by the 3-axis interleaved-pair RoPE, for q and
 * (optionally) k in one launch; bf16 in/out (in place allowed), fp32 math, one rounding.
 * rope_cos/rope_sin: [tokens, head_dim/2] fp32 per-token tables (identity rows = pass-through);
 * NULL = no rotation (cross-attention q).  token(m) = token_offset + m % tokens_per_batch.
 * Replaces WanRMSNorm.forward + rope_apply_qk: wan_transformer3d_FlexAM.py:137-170,173-189,242-249. */
int flexam_rmsnorm_rope(const void* q_in, int64_t ldq_in, void* q_out, int64_t ldq_out, const float* wq, const void* k_in,
                        int64_t ldk_in, void* k_out, int64_t ldk_out, const float* wk, int64_t M, int C, float eps,
                        const float* rope_cos, const float* rope_sin, int64_t tokens_per_batch, int64_t token_offset,
                        int head_dim, void* stream);

/* Same arithmetic, written straight into a sequence-parallel SEND layout (no pack copy): q and k are normed + rotated, v is
 * copied through; element (m, col) of each goes to
 *   *_out + (m / tokens_per_batch) * out_bs + (m % tokens_per_batch) * ld_out + (col / col_block) * block_stride + col % col_block
 * so a row is cut into column blocks (one per destination rank's group of heads) that land block_stride elements apart, and the
 * three tensors interleave through their base pointers (q_out, k_out = q_out + col_block, ...).  q / v may be NULL (the K|V
 * all-gather layout norms only k).  Replaces the missing FlexAM/dist exchange plumbing behind wan_transformer3d_FlexAM.py:801-815. */
int flexam_rmsnorm_rope_scatter(const void* q_in, int64_t ldq_in, const float* wq, const void* k_in, int64_t ldk_in, const float* wk,
                                const void* v_in, int64_t ldv_in, void* q_out, void* k_out, void* v_out, int64_t ld_out,
                                int64_t out_bs, int col_block, int64_t block_stride, int64_t M, int C, float eps,
                                const float* rope_cos, const float* rope_sin, int64_t tokens_per_batch, int64_t token_offset,
                                int head_dim, void* stream);

/* out[blk][r][j][:] = mod[blk][j][:] + e[r][j][:] + ((scale_mask>>j)&1) + density terms, where slot
 * s = (dens_slots >> 4j) & 0xF (0xF = none) adds mdens[blk][s][:] + dens[r / rows_per_batch][s][:].
 * One launch per denoise step builds the AdaLN tables of all blocks (wan_transformer3d_FlexAM.py:
 * 444-449,452,464; head :500-506). */
int flexam_mod_table(const float* mod, const float* e, const float* mdens, const float* dens, float* out, int nblk, int R,
                     int nj, int nslot, int C, int rows_per_batch, int scale_mask, int dens_slots, void* stream);

/* y[M,N] = silu_in?(x[M,K]) . W[N,K]^T + b, fp32 math, 1 <= M <= 32, W bf16 or fp32.  The time /
 * density embedding MLPs (forced fp32 in the reference: wan_transformer3d_FlexAM.py:928-955).  Every output is summed in the
 * same order whatever M is (M <= 8: one output per wave; above: four per wave and 32 rows per pass over W, for foreground
 * masks with many distinct per-token timesteps). */
int flexam_small_linear_f32(const float* x, int64_t ldx, const void* W, int w_is_bf16, int64_t ldw, const float* b, float* y,
                            int64_t ldy, int M, int N, int K, int silu_in, void* stream);

/* out[r][:dim/2] = cos(t[r] f_i), out[r][dim/2:] = sin(t[r] f_i), f_i = 10000^(-i/(dim/2)), fp64 math
 * (sinusoidal_embedding_1d, wan_transformer3d_FlexAM.py:31-41). */
int flexam_sinusoid_embed(const float* t, float* out, int R, int dim, void* stream);

/* im2col of a kernel = stride = (1,2,2) conv: src [C,F,H,W] (fp32 or bf16) ->
 * dst[row0 + (f,h/2,w/2)][col0 + c*4 + ph*2 + pw] bf16 (weight.flatten(1) column order).
 * patch_embedding / ref_conv inputs: wan_transformer3d_FlexAM.py:624-625,676,885,896. */
int flexam_patchify(const void* src, int src_is_bf16, int C, int F, int H, int W, void* dst, int64_t ldd, int col0,
                    int64_t row0, void* stream);

/* Head output tokens [L, 4C] fp32 (col = (ph*2+pw)*C + c), starting at token tok0 -> [C,F,H,W].
 * unpatchify, wan_transformer3d_FlexAM.py:1126-1149 (tok0 skips the reference-image tokens, :1106-1109). */
int flexam_unpatchify(const float* tok, int64_t ldt, int64_t tok0, int C, int F, int H, int W, void* dst, int dst_is_bf16,
                      void* stream);

/* One fused sampler step on fp32 latents [C,F,H,W]: v = u + guidance (c - u) (tok_cond NULL: v = u),
 * x += dt v, x = (1 - mask) known + mask x (mask [F,H,W], NULL: no blend).
 * FlexAM/pipeline/pipeline_wan2_2_fun_control_FlexAM.py:926-934 (+ unpatchify of both CFG rows). */
int flexam_cfg_euler_blend(const float* tok_uncond, const float* tok_cond, int64_t ldt, int64_t tok0, float guidance, float dt,
                           float* latents, const float* known, const float* mask, int C, int F, int H, int W, void* stream);

/* Multistep samplers (FlexAM/utils/fm_solvers_unipc.py:640-724, fm_solvers.py:706-798; PIPE.py:926-934 around them):
 * cfg_velocity: v[C,F,H,W] = unpatchify(u + g (c - u)) -- the noise_pred handed to scheduler.step;
 * lincomb_f32:  out = sum_i coefs[i] * terms[i] (n_terms <= 8; `terms` / `coefs` are HOST arrays, terms[i] device
 *               pointers; out may alias a term) -- every UniPC / DPM-Solver++ update is one such combination;
 * mask_blend_f32: x = (1 - mask) known + mask x, mask [fhw] broadcast over C channels. */
int flexam_cfg_velocity(const float* tok_uncond, const float* tok_cond, int64_t ldt, int64_t tok0, float guidance, float* v, int C,
                        int F, int H, int W, void* stream);
int flexam_lincomb_f32(float* out, int64_t n, int n_terms, const float* const* terms, const float* coefs, void* stream);
int flexam_mask_blend_f32(float* x, const float* known, const float* mask, int C, int64_t fhw, void* stream);

/* y[i] = a*x[i] + b*y[i], fp32, n % 4 == 0.  TeaCache bookkeeping: residual = x_after_blocks - x_before and
 * x += cached residual on skipped steps (wan_transformer3d_FlexAM.py:1003-1051). */
int flexam_axpby_f32(float* y, float a, const float* x, float b, int64_t n, void* stream);

/* out2[0] += sum of the buffer's 32-bit words, out2[1] += sum of word * (index + 1), mod 2^64 (out2: two uint64 the caller
 * zeroed).  A content key for the step-invariant conditioning tensors that the reference sampler re-creates with torch.cat on
 * every step (pipeline_wan2_2_fun_control_FlexAM.py:850-886): the DiT's per-clip work is redone only when it changes. */
int flexam_checksum(const void* data, int64_t nbytes, uint64_t* out2, void* stream);

/* Channels-last helpers around the implicit-GEMM convolutions (cnn-block: wan_transformer3d_FlexAM.py:
 * 680-705,869-881; VAE decoder: wan_vae3_8.py).  A conv input is a spatially zero-padded bf16 image
 * img[f][H+2][W+2][Cp]; tap (dh,dw) of a 3x3 kernel is then the constant A offset
 * ((dh-1)*(W+2)+(dw-1))*Cp handed to flexam_gemm_bf16 through a_koff.
 * pack_cl:   src [C,F,H,W] (fp32|bf16) -> interior of img at channel offset c0.
 * unpack_cl: rows [(f,hp,wp)][ld] (fp32|bf16), interior positions -> dst [C,F,H,W] fp32.
 * groupnorm_silu_cl: GroupNorm(groups, eps) over all interior positions of x [(f,hp,wp)][ld] fp32
 *   (stats: scratch [2*groups] fp32), * gamma + beta, SiLU, optional + residual (bf16 padded image,
 *   channel stride res_cp) -> interior of the bf16 padded image dst (channel stride Cp). */
int flexam_pack_cl(const void* src, int src_is_bf16, int C, int F, int H, int W, void* dst, int Cp, int c0, void* stream);
int flexam_unpack_cl(const void* src, int src_is_bf16, int64_t ld, int C, int F, int H, int W, float* dst, void* stream);
int flexam_groupnorm_silu_cl(const float* x, int64_t ld, int C, int F, int H, int W, int groups, float eps, const float* gamma,
                             const float* beta, float* stats, const void* residual, int res_cp, void* dst, int Cp, void* stream);

/* Wan2.2 3D-VAE decoder helpers (FlexAM/models/wan_vae3_8.py).  "rows" = [(t,hp,wp), ld] matrices
 * indexed by PADDED position (GEMM outputs), "image" = zero-bordered bf16 [frames,H+2,W+2,Cp].
 * vae_prep_cl: interior rows -> image (frame offset t0) or compact matrix; mode 0 cast, 1 RMS_norm
 *   (F.normalize over channels * sqrt(C) * gamma, :50-64), 2 RMS_norm + SiLU (:206-212, :671-673).
 * upsample2x_cl: nearest-exact 2x (+ frame de-interleave of the time_conv output, :153-156) -> image.
 * dupup_add_cl: x_main += DupUp3D(x_in) (:375-417), drop = leading frames cropped on the first chunk.
 * softmax_rows: bf16 softmax(scale * s) rows, zero padded to Npad columns (AttentionBlock :272).
 * scatter_add_cl: x[padded rows] += y[compact rows] (AttentionBlock residual :282).
 * vae_unpatchify_clamp: 12-channel rows -> video[3][Ftot][2H][2W] at frame f0, clamped (:304-318, :1043).
 * pack_affine_cl: z [C,T,H,W] * mul[c] + add[c] -> image (latent de-normalisation, :823-828). */
int flexam_vae_prep_cl(const void* src, int src_is_bf16, int64_t ld_src, int C, int T, int H, int W, const float* gamma, int mode,
                       void* dst, int Cp, int t0, int dst_compact, void* stream);
int flexam_upsample2x_cl(const void* src, int src_is_bf16, int64_t ld_src, int C, int T, int H, int W, int interleave, void* dst,
                         int Cp, void* stream);
int flexam_dupup_add_cl(float* x_main, int64_t ld_main, int Co, int To, int Ho, int Wo, const float* x_in, int64_t ld_in, int Ci,
                        int ft, int drop, void* stream);
/* Phase-decomposed "nearest 2x upsample + Conv2d 3x3" (Resample upsample2d / upsample3d, wan_vae3_8.py:76-99,153-160): output pixels of
 * parity (a, b) are a 2x2 convolution of the LOW-resolution image with pre-summed taps -- four flexam_gemm_bf16 launches over the low-
 * resolution image (16 instead of 36 tap products per low-resolution pixel; the upsampled image is never built).
 * deinterleave_cl: rows [T, 2C] -> padded image of 2T frames at the same resolution, frame 2t + s = channels [sC, (s+1)C) of frame t
 *   (the temporal half of upsample3d, :153-156).
 * phase_dupup_cl: x_main[(t', h', w')] = phases[(h'&1)*2 + (w'&1)][(t', h'>>1, w'>>1)] + DupUp3D(x_in) (the shortcut of dupup_add_cl,
 *   :375-417); phases = four [rows of the padded low-resolution image, Co] fp32 matrices `phase_stride` elements apart. */
int flexam_deinterleave_cl(const void* src, int src_is_bf16, int64_t ld_src, int C, int T, int H, int W, void* dst, int Cp, void* stream);
int flexam_phase_dupup_cl(const float* phases, int64_t ld_ph, int64_t phase_stride, float* x_main, int64_t ld_main, int Co, int To, int Ho,
                          int Wo, const float* x_in, int64_t ld_in, int Ci, int ft, int drop, void* stream);
/* tapsum_cl: a causal kt x 3 x 3 convolution with FEW output channels (decoder head 256 -> 12, wan_vae3_8.py:668-672) as one plain GEMM
 * over the input pixels (Y[pixel, tap * Co + o] = W[o, :, tap] . x[pixel, :], K = Cin: every activation read once) and this gather:
 * out[(t,h,w), o] = bias[o] + sum_taps Y[(t + dt, h + dh - 1, w + dw - 1), tap * Co + o].  Y rows = padded positions of the kt - 1 history
 * frames followed by the T current frames; out rows = padded positions of the T current frames (interior written). */
int flexam_tapsum_cl(const float* y, int64_t ld_y, int T, int H, int W, int kt, int Co, const float* bias, float* out, int64_t ld_out,
                     void* stream);
int flexam_softmax_rows(const float* s, int64_t ld_s, int64_t M, int N, float scale, void* out, int64_t ld_out, int Npad, void* stream);
int flexam_scatter_add_cl(float* x, int64_t ldx, const void* y, int64_t ldy, int C, int T, int H, int W, void* stream);
int flexam_vae_unpatchify_clamp(const float* src, int64_t ld_src, int T, int H, int W, float* video, int Ftot, int f0, float lo,
                                float hi, void* stream);
int flexam_pack_affine_cl(const float* src, int C, int T, int H, int W, const float* mul, const float* add, void* dst, int Cp,
                          void* stream);


/* Wan2.2 3D-VAE encoder helpers (FlexAM/models/wan_vae3_8.py:505-618 Encoder3d, :420-457 Down_ResidualBlock).
 * vae_patchify_cl: video[3][Ftot][2H][2W] frames f0..f0+T -> 12-channel image (patchify :285-301,
 *   channel c*4 + r*2 + q), frame offset t0 (history frames of the causal conv1).
 * space_to_depth_cl: rows at HxW -> image at H/2 x W/2 with 4 sub-pixel channel groups of Cs channels, so
 *   that ZeroPad2d((0,1,0,1)) + Conv2d(3, stride 2) (:104-113) is a unit-stride implicit GEMM.
 * avgdown_add_cl: x_main += AvgDown3D(x_in) (:321-372; front zero pad in time when Ti % ft != 0). */
int flexam_vae_patchify_cl(const float* video, int Ftot, int f0, int T, int H, int W, void* dst, int Cp, int t0, void* stream);
int flexam_space_to_depth_cl(const void* src, int src_is_bf16, int64_t ld_src, int C, int T, int H, int W, void* dst, int Cs, int t0,
                             void* stream);
int flexam_avgdown_add_cl(float* x_main, int64_t ld_main, int Co, int To, int Ho, int Wo, const float* x_in, int64_t ld_in, int Ci,
                          int Ti, int ft, int fs, void* stream);


/* umT5 text encoder helpers (FlexAM/models/wan_text_encoder.py; projections are flexam_gemm_bf16).
 * t5_norm: out = w * x * rsqrt(mean(x^2) + eps) per row (T5LayerNorm :44-56), bf16 or fp32 out.
 * softmax_bias_rows: P = softmax(scale * s + bias) over the keys with key_mask != 0 (T5Attention :91-103:
 *   relative-position bias [M, N], padding mask [N]; bias / key_mask may be NULL), bf16, zero padded to Npad.
 * mul_bf16: out = a * b elementwise (fc1(x) * gate(x), :125-126). */
int flexam_t5_norm(const float* x, int64_t ldx, int64_t M, int C, float eps, const float* w, void* out, int64_t ld_out, int out_f32,
                   void* stream);
int flexam_softmax_bias_rows(const float* s, int64_t ld_s, int64_t M, int N, float scale, const float* bias, int64_t ld_bias,
                             const float* key_mask, void* out, int64_t ld_out, int Npad, void* stream);
int flexam_mul_bf16(const void* a, const void* b, void* out, int64_t n, void* stream);



/* Conditioning rasteriser: tracked points [T, N, 3] (u, v, depth; fp32) -> conditioning video frames.  Replaces the per-point PIL
 * drawing loops of the reference's pipelines.py -- fun_visualize_tracking_with_depth :1501-1575 (valid_mask :1200-1212,
 * sort_points_by_depth :1214-1232, draw_rectangle :1234-1253), _render_cosine_encoded_frame :1694-1728, _visualize_depth_tracking
 * :1763-1820, _should_draw_point :1842-1850, _convert_frames_to_tensor :1658-1660 -- whose result per pixel is the colour of the
 * NEAREST point among the squares covering it (far-to-near painting).
 * raster_keys: keys[T][H][W] (uint64) = min over the drawn points whose square [x - half, x + half] x [y - half, y + half] (clipped to
 *   the frame) covers the pixel of (order-preserving code of the depth << 32 | point index); all ones = no point.  A point is drawn when
 *   visible[t][n] != 0 (NULL: all), its (u, v) are finite, (x, y) = (u, v) truncated towards zero lie in [0, W) x [y_min, H) (y_min = 1
 *   for the tracking video, whose frame test is y > 0, else 0) and, with a mask video [T][H][W], mask[t][y][x] > 0.5.  Equal depths: the
 *   lower index wins; NaN depths lose against everything.  The call clears `keys` itself.
 * raster_resolve: out_u8[T][H][W][3] and / or out_f32[3][T][H][W] (= byte / 255, correctly rounded) from the winners' colours;
 *   colors[N][3] bytes shared by all frames (color_frame_stride = 0) or [T][N][3] (stride N * 3: the depth video's per-frame colours).
 *   N = rows of a colour table = the point count the keys were made with: a key whose index is >= N (keys of another point set) resolves
 *   to black instead of reading past the table; a per-frame stride other than 0 or N * 3 is refused. */
int flexam_raster_keys(const float* points, const unsigned char* visible, int T, int N, int H, int W, int half, int y_min,
                       const float* mask, unsigned long long* keys, void* stream);
int flexam_raster_resolve(const unsigned long long* keys, const unsigned char* colors, int64_t color_frame_stride, int N, int T, int H, int W,
                          unsigned char* out_u8, float* out_f32, void* stream);

/* Command lists: a recorded sequence of this library's stream-ordered calls re-issued from ONE call (csrc/replay.hip).  The reference
 * has no counterpart -- its step loop is Python over PyTorch ops (pipeline_wan2_2_fun_control_FlexAM.py:844-949, one Python frame per
 * nn.Module call of wan_transformer3d_FlexAM.py:1053-1089); this is what replaces that per-op host cost when one rank of eight has
 * ~40 ms per step for ~450 launches.  A command = the id of an entry point of THIS header whose last parameter is `void* stream`
 * (flexam_fn_id(name); ids are positions in the header, never hard-coded) + its arguments without the stream as 8-byte words in
 * declaration order: pointers -> p, float -> f (as double), integers -> i.  flexam_replay calls them in order on `stream`; every call
 * checks its own arguments as usual; it stops at the first failure (its code returned, its message in flexam_last_error(),
 * *failed_at = index; -1 = none).  The list is HOST memory of the caller, read during the call only. */
/* Emulation aid (not on the product path): one wave that waits `us` microseconds of the constant 100 MHz counter on `stream`.  The
 * one-GPU emulation of a multi-GPU rank (bench.py --emulate-rank, flexam_amd.dist.LoopbackGroup) puts it where a collective's transfer
 * time would sit -- bytes per xGMI link / an ASSUMED link rate -- on the side stream RCCL's kernels would occupy. */
int flexam_delay_us(float us, void* stream);

#define FLEXAM_REPLAY_MAX_ARGS 26
typedef union { int64_t i; double f; void* p; } flexam_arg;
typedef struct { int32_t fn; int32_t nargs; flexam_arg a[FLEXAM_REPLAY_MAX_ARGS]; } flexam_cmd;
int flexam_fn_id(const char* name);            /* -1: not a replayable entry point */
int flexam_fn_count(void);
const char* flexam_fn_name(int id);
int flexam_replay(const flexam_cmd* cmds, int64_t n, int64_t* failed_at, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FLEXAM_HIP_H */
