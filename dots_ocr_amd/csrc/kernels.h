// Host-side launchers of the gfx950 kernels (one per .hip translation unit).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;

#define DOTS_MAX_BATCH 64     // sequences per decode step: 4 tiles of 16 rows (decode_layout.h MAX_DECODE_ROWS)

enum { EPI_NONE = 0, EPI_RESIDUAL = 1, EPI_SWIGLU = 2, EPI_GELU = 3, EPI_F32 = 4, EPI_QKROPE = 5 /* launch_gemm_qk_rope only */ };

// ---- gemm.hip
hipError_t launch_gemm(hipStream_t s, const bf16_t* A, const bf16_t* W, const bf16_t* bias, const bf16_t* R,
                       void* C, int64_t M, int N, int K, int lda, int ldc, int epi, const float* colscale = nullptr);
int gemm_get_plan();            // 0 = 8-wave ping-pong kernel, 1 = one wave per SIMD (gemm.hip); process-wide
void gemm_set_plan(int plan);
// colscale: fp32 [N] multiplied into the accumulator column before bias (fp8 weights: W holds bf16(q), quant.hip), or nullptr

// ---- elementwise.hip
hipError_t launch_rmsnorm(hipStream_t s, const bf16_t* x, const bf16_t* w, bf16_t* y, int64_t rows, int dim, float eps);
hipError_t launch_layernorm(hipStream_t s, const bf16_t* x, const bf16_t* w, const bf16_t* b, bf16_t* y,
                            int64_t rows, int dim, float eps);
// f32 [rows, in_dim] -> bf16 [rows, out_dim] (zero padded columns), the patch-embed GEMM's A operand
hipError_t launch_patch_prep(hipStream_t s, const float* x, bf16_t* y, int64_t rows, int in_dim, int out_dim);
// cos/sin tables [T, 64] float2.  2-D: pos [T,2] (h,w), 32 frequencies per axis; 1-D: pos [T], 64 frequencies.
hipError_t launch_rope_table(hipStream_t s, const int32_t* pos, const float* inv_freq, float2* cs, int64_t T, int two_d);
// Work item of the 64-token tile kernels: tok0 = first packed token, n = valid tokens (<=64),
// pad0 = first padded position in the V^T buffer.
struct Tile64 { int32_t tok0, n, pad0, seq, page, _pad; };   // page = tile index inside its sequence
hipError_t launch_qkv_rope_split(hipStream_t s, const bf16_t* qkv, const float2* cs, const Tile64* tiles, int n_tiles,
                                 bf16_t* q, bf16_t* k, bf16_t* vt, int64_t T, int64_t Tpad, int Hq, int Hkv, bool v_only = false);
// ---- gemm.hip: the fused qkv projection of a prefill pass with the rotary embedding in its epilogue (round 6).  C = A W^T (+ bias) as launch_gemm
// with EPI_NONE, rounded to bf16 as that GEMM would store it, then: columns [0, (Hq + Hkv) * 128) — the q and k heads — are rotated with cs[t]
// (the arithmetic of qkv_rope_split_kernel, bit for bit) and written head-major into q [Hq][T][128] / k [Hkv][T][128]; the v columns go to
// qkv [T][ldc] unrotated (launch_qkv_rope_split(.., v_only = true) transposes them).  hipErrorNotSupported when the shape or the process's GEMM
// plan has no such kernel (the caller then runs launch_gemm + launch_qkv_rope_split).
struct QkRope { const float2* cs; bf16_t* q; bf16_t* k; long long T; int Hq; int n_rope_heads; };
hipError_t launch_gemm_qk_rope(hipStream_t s, const bf16_t* A, const bf16_t* W, const bf16_t* bias, bf16_t* qkv, int64_t M, int N, int K, int lda, int ldc,
                               const float2* cs, bf16_t* q, bf16_t* k, int Hq, int Hkv);
// x[t] = src[t] >= 0 ? embed[src[t]] : vision[-src[t]-1]
hipError_t launch_embed_gather(hipStream_t s, const int32_t* src, const bf16_t* embed, const bf16_t* vision, bf16_t* x,
                               int64_t T, int dim);
// y[dst ? dst[r] : r] = x[rows[r]]
hipError_t launch_gather_rows(hipStream_t s, const bf16_t* x, const int32_t* rows, const int32_t* dst, bf16_t* y, int n, int dim);

// ---- attn_prefill.hip
// One work item = (sequence, head, 128-row query block); the list is ordered seq-major, then head, then block, so that
// the XCD-contiguous remap of the 1-D grid gives one XCD (one L2) all query blocks that stream the same K/V.
struct QBlock { int32_t q0, n, tok0, pad0, head, _pad; };   // first row in seq, seq length, packed token offset, padded V^T offset
int flash_rows_per_block();      // 128 or 256 query rows per QBlock work item (what build_worklists must use)
// Which contiguous chunk of the QBlock list each XCD walks (workgroup i runs on XCD i % 8 and takes item base[i % 8] + i / 8; workgroups past
// cnt[] exit at once).  xcd_remap's chunks hold equal COUNTS; a block costs its sequence's KV tiles, so on a ragged batch (mixed page
// sizes) equal counts gave one XCD 1.32 x the mean work while the others idled (round 5, mixed64's towers).  make_xcd_plan cuts the list
// by COST instead (prefix-sum split: chunks stay contiguous, so the blocks of a (sequence, head) still share an L2); equal lengths give
// xcd_remap's chunks exactly.  nullptr = xcd_remap.
struct XcdPlan { int32_t base[8], cnt[8]; };
XcdPlan make_xcd_plan(const QBlock* blocks_host, int n_blocks);
hipError_t launch_flash_attn(hipStream_t s, const bf16_t* q, const bf16_t* k, const bf16_t* vt, bf16_t* out,
                             const QBlock* blocks, int n_blocks, int64_t T, int64_t Tpad, int Hq, int Hkv,
                             int causal, float scale, const XcdPlan* plan = nullptr);

// ---- decode.hip  (KV page pool layout documented there)
hipError_t launch_kv_to_pages(hipStream_t s, const bf16_t* k, const bf16_t* qkv, const Tile64* tiles, int n_tiles,
                              const int32_t* block_table, int max_pages, bf16_t* pool_layer, int64_t T, int Hq, int Hkv);
// ---- decode_fused.hip: dense layers of the decode step with in-workgroup split-K and fused prologues/epilogues.
// Activations between them travel as X images [K/8][XR][8], XR = 8 (B <= 8) or 16 (decode_layout.h).
hipError_t launch_dec_embed(hipStream_t s, const int32_t* tokens, const bf16_t* embed, bf16_t* h, int B, int dim);
// Wd + wscale: wscale == nullptr -> Wd is the bf16 fragment image (launch_pack_frag*); else Wd is the e4m3 fragment image
// (launch_pack_frag_fp8) and wscale[row] its fp32 per-output-channel scales (quant.hip).
// xn: scratch for the normalised rows of batches above 32 rows (MAX_DECODE_ROWS x H bf16; decode_b64.hip) or nullptr = the round-4 two-tile kernels.
// pend / pend_scale (qkv, gate|up, lm_head): the K-quarter sums a preceding launch_dec_proj left in `part` instead of updating h (+ that projection's fp8 weight scales or
// nullptr): the residual update is applied first — fused into the norm kernel on the xn path, by a launch of its own otherwise.
hipError_t launch_dec_qkv(hipStream_t s, const bf16_t* h, const bf16_t* ln_w, const void* Wd, const float* wscale, const bf16_t* bias,
                          const float* inv_freq, const int32_t* ctx_len, const int32_t* block_table, int max_pages,
                          bf16_t* pool_layer, bf16_t* q_out, int B, int H, int Hq, int Hkv, float eps, int part_cus = 0, bf16_t* xn = nullptr,
                          const float* pend = nullptr, const float* pend_scale = nullptr);
// h += X @ W^T.  part != nullptr (DEC_KSPLIT_PARTS x DOTS_MAX_BATCH x N fp32) allows the K-split kernel above 32 rows: *pending is then set and h is NOT
// updated by this launch — pass `part` (and wscale) as pend / pend_scale to the next launch_dec_qkv / launch_dec_gateup / launch_dec_lmhead, or call launch_dec_norm_ximg(.., part, ..)
hipError_t launch_dec_proj(hipStream_t s, const bf16_t* X, const void* Wd, const float* wscale, bf16_t* h, int B, int N, int K, int part_cus = 0,
                           float* part = nullptr, bool* pending = nullptr);
// part_cus > 0 (all dense launchers): the stream is CU-masked to that many CUs.  qkv / proj: whole 16-row weight tiles per workgroup at B <= 16,
// one round of wide workgroups above; gate|up: the grid is capped at what the CUs hold at once, the workgroups walk the (gate, up) tile pairs
hipError_t launch_dec_gateup(hipStream_t s, const bf16_t* h, const bf16_t* ln_w, const void* W13d, const float* wscale, bf16_t* act,
                             int B, int H, int I, float eps, int part_cus = 0, bf16_t* xn = nullptr, const float* pend = nullptr, const float* pend_scale = nullptr);
hipError_t launch_dec_lmhead(hipStream_t s, const bf16_t* h, const bf16_t* ln_w, const void* Wd, const float* wscale, float* logits,
                             int B, int H, int V, float eps, int part_cus = 0, bf16_t* xn = nullptr, const float* pend = nullptr, const float* pend_scale = nullptr);
// ---- decode_b64.hip (round 6): batches above 32 rows — all four 16-row batch tiles in one workgroup, every weight byte crosses a CU once
bool dec_stream64_supports(int B, int H);
// rows -> [pending residual update ->] rmsnorm -> X image (xn == nullptr with part != nullptr: the residual update only)
hipError_t launch_dec_norm_ximg(hipStream_t s, const bf16_t* h, const bf16_t* ln_w, bf16_t* xn, int B, int H, float eps, const float* part = nullptr, const float* pscale = nullptr);
hipError_t launch_dec_gateup64(hipStream_t s, const bf16_t* h, const bf16_t* ln_w, const void* W13d, const float* wscale, bf16_t* act, bf16_t* xn,
                               int B, int H, int I, float eps, int cus, const float* pend = nullptr, const float* pend_scale = nullptr);
hipError_t launch_dec_lmhead64(hipStream_t s, const bf16_t* h, const bf16_t* ln_w, const void* Wd, const float* wscale, float* logits, bf16_t* xn,
                               int B, int H, int V, float eps, int cus, const float* pend = nullptr, const float* pend_scale = nullptr);
constexpr int DEC_KSPLIT_PARTS = 4;          // K quarters of the K-split projection; part buffers hold DEC_KSPLIT_PARTS x DOTS_MAX_BATCH x N fp32
bool dec_proj_ksplit_supports(int B, int N, int K);
hipError_t launch_dec_proj_ksplit(hipStream_t s, const bf16_t* X, const void* Wd, bool fp8, float* part, int B, int N, int K, int cus);
int decode_attn_waves();                       // pages in flight per decode-attention workgroup (engine constant)
int decode_attn_splits(int max_seq_len);       // KV splits for a context capacity: ceil(pages / waves), at most 64
hipError_t launch_decode_attn(hipStream_t s, const bf16_t* q, const bf16_t* pool_layer, const int32_t* ctx_len,
                              const int32_t* block_table, int max_pages, float* part_o, float* part_ml,
                              int B, int Hq, int Hkv, int n_splits, float scale, int part_cus = 0, int stream_mode = -1);
// > 0: launch_decode_attn runs the streaming kernel (round 5) with that many resident workgroups; 0: one workgroup per (row, kv head, split).
// part_cus > 0: the stream is CU-masked to that many CUs; stream_mode 1 / 0 / -1 = always where legal / never / by items per CU
int decode_attn_stream_wgs(int B, int Hkv, int n_splits, int max_pages, int part_cus, int stream_mode = -1);
hipError_t launch_decode_attn_combine(hipStream_t s, const float* part_o, const float* part_ml, const int32_t* ctx_len, bf16_t* out,
                                      int B, int Hq, int Hkv, int n_splits);
// Per-step token bookkeeping state (device pointers), shared by the arg-max and the sampling kernels.
//   sel     rows (slots) to act on this call, or nullptr = all B rows
//   max_len per-row cap on generated tokens, or nullptr = `cap` for every row
struct StepState {
    int32_t *cur_tokens, *ctx_len, *out_ids, *out_lens, *finished;
    const int32_t *eos_ids, *sel, *max_len;
    int n_eos, out_stride, cap, advance_ctx;
};
hipError_t launch_argmax_step(hipStream_t s, const float* logits, int V, int ld, int B, float* pval, int32_t* pidx, const StepState& st);
hipError_t launch_sample_step(hipStream_t s, const float* logits, int V, int ld, int B, float temperature, float top_p, uint64_t seed,
                              const StepState& st);
// row-major [rows, K] -> MFMA fragment order (decode.hip): 16-row tiles x K/32 chunks of 1 KiB
hipError_t launch_pack_frag(hipStream_t s, const bf16_t* src, bf16_t* dst, int64_t rows, int K);
// fused qkv weight [(Hq + 2 Hkv) * 128, K]: as launch_pack_frag, q / k head rows permuted so that a 16-row tile holds whole RoPE pairs
hipError_t launch_pack_frag_qkv(hipStream_t s, const bf16_t* src, bf16_t* dst, int Hq, int Hkv, int K);
// row-major [rows <= 16, K] <-> X image [K/8][XR][8] (XR = 8 for rows <= 8, else 16): the single-kernel entry points' converters
hipError_t launch_pack_x(hipStream_t s, const bf16_t* src, bf16_t* x, int rows, int K);
hipError_t launch_unpack_x(hipStream_t s, const bf16_t* x, bf16_t* dst, int rows, int K);
// ---- quant.hip: fp8 (e4m3, per-output-channel scale) weights
// W[n][:] <- bf16(e4m3(W[n][:] / scale[n])) in place, scale[n] = max|W[n][:]| / 448 (1 for an all-zero row)
hipError_t launch_quant_rows_fp8(hipStream_t s, bf16_t* W, float* scale, int64_t N, int K);
// bf16(q) row-major [rows, K] -> e4m3 bytes in decode fragment order (512-B chunks); rot_rows = (Hq + Hkv) * 128 for the fused qkv weight, else 0
hipError_t launch_pack_frag_fp8(hipStream_t s, const bf16_t* src, uint8_t* dst, int64_t rows, int K, int rot_rows);
// activations: x bf16 [M][lda] -> q e4m3 [M][K] + scale[m] (per token); bf16(q) weights -> e4m3 bytes row-major
hipError_t launch_quant_act_fp8(hipStream_t s, const bf16_t* X, uint8_t* Q, float* scale, int64_t M, int K, int lda);
hipError_t launch_bf16q_to_fp8(hipStream_t s, const bf16_t* src, uint8_t* dst, int64_t n);
// ---- gemm.hip, fp8 MFMA: C = epilogue((Aq Wq^T) * rowscale[m] * colscale[n] + bias); Aq [M][K], Wq [N][K] e4m3 row-major;
// N % 256 == 0, K % 64 == 0, ldc % 8 == 0.  EPI_F32 is not offered.
hipError_t launch_gemm_fp8(hipStream_t s, const uint8_t* Aq, const float* rowscale, const uint8_t* Wq, const float* colscale, const bf16_t* bias,
                           const bf16_t* R, void* C, int64_t M, int N, int K, int ldc, int epi);
bool gemm_fp8_supports(int N, int K);
// ---- engine.hip helper kernels
hipError_t launch_pack_w13(hipStream_t s, const bf16_t* gate, const bf16_t* up, bf16_t* out, int I, int K);
hipError_t launch_convert_to_bf16(hipStream_t s, const void* src, int dtype, bf16_t* dst, int64_t n);

// ---- preprocess.hip: Pillow-exact bicubic resize + normalise + patchify on the GPU
hipError_t launch_resize_h(hipStream_t s, const uint8_t* in, uint8_t* out, const int32_t* coef, const int32_t* bounds,
                           int ksize, int H, int W, int rw);
hipError_t launch_resize_v(hipStream_t s, const uint8_t* in, uint8_t* out, const int32_t* coef, const int32_t* bounds,
                           int ksize, int W, int rh);
hipError_t launch_normalize_patchify(hipStream_t s, const uint8_t* img, float* out, int rw, int gh, int gw, int P, int m,
                                     float r255, const float* mean, const float* stdv);
