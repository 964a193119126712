/* dots_ocr_hip.h — C ABI of the MI355X-native dots.ocr inference engine (libdots_ocr_hip.so).
 *
 * The reference (rednote-hilab/dots.ocr) has NO native layer: its hot path is the three HF
 * objects used in DotsOCRParser._inference_with_hf (dots_ocr/parser.py:78-117):
 *     self.model.generate(**inputs, max_new_tokens=...)         parser.py:110
 *     self.processor(...) / apply_chat_template / batch_decode  parser.py:93-105,114-116
 *     AutoModelForCausalLM.from_pretrained(...)                 parser.py:68-74
 * This header is the boundary a native binding for that path binds instead:
 * plain pointers and sizes, opaque handle, int status codes, no exceptions, no torch types.
 *
 * Conventions
 *   - every function returns 0 on success, a negative DOTS_E_* code on failure; the message is
 *     available from dots_last_error(handle) (or dots_last_error(NULL) for create failures);
 *   - the caller owns every buffer it passes; the engine owns all device memory it allocates;
 *   - one handle = one GPU = one HIP stream.  A handle is NOT thread-safe; different handles
 *     may be driven from different threads/processes (one per GPU);
 *   - pointers named *_dev are device pointers on the handle's GPU, *_host are host pointers;
 *     parameters named `x` with a companion `x_on_device` flag accept either;
 *   - bf16 tensors are raw uint16_t.
 */
#ifndef DOTS_OCR_HIP_H
#define DOTS_OCR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DOTS_OK 0
#define DOTS_E_INVALID (-1)   /* bad argument / unsupported shape */
#define DOTS_E_HIP (-2)       /* HIP runtime error */
#define DOTS_E_STATE (-3)     /* call order (weights missing, sequence not prefetched, ...) */
#define DOTS_E_CAPACITY (-4)  /* exceeds max_batch / max_seq_len / workspace */

#define DOTS_DTYPE_BF16 0
#define DOTS_DTYPE_F32 1
#define DOTS_DTYPE_F16 2

typedef struct DotsEngine DotsEngine;

/* Mirrors the checkpoint's config.json (+ vision_config) that from_pretrained reads
 * (parser.py:68-74).  Hard-coded in the kernels: head_dim == 128; dots_create also requires hidden_size % 256 == 0 and
 * hidden_size <= 1536 (the decode kernels keep a residual row in 3 x 16-byte chunks per lane), the same for the vision embed_dim's
 * norm path — dots.ocr's 1536 / 1536 fit; a wider model needs NC_MAX raised in csrc/decode_dev.h. */
typedef struct DotsConfig {
    /* language model (Qwen2 architecture) */
    int32_t hidden_size, num_layers, num_heads, num_kv_heads, head_dim, intermediate_size, vocab_size;
    float rope_theta, rms_norm_eps;
    int32_t attention_bias;
    int32_t image_token_id;
    /* vision tower (NaViT) + patch merger */
    int32_t v_embed_dim, v_layers, v_heads, v_intermediate, v_patch, v_merge, v_channels, v_temporal_patch;
    float v_rms_eps, v_ln_eps;
    int32_t v_use_bias, v_post_norm;
    /* runtime capacity */
    int32_t max_batch;        /* sequences decoded together: <= 64; the decode kernels work in tiles of 16 rows (one MFMA column tile),
                                 batches above 16 re-read each weight slice once per tile from L2 / the Infinity Cache */
    int32_t max_seq_len;      /* prompt + generated tokens per sequence */
    int64_t max_patches;      /* vision patches per dots_vit_forward call (workspace) */
    int64_t max_prefill_tokens; /* packed prompt tokens per dots_prefill call */
    int64_t kv_pool_tokens;   /* paged KV cache: tokens the page pool holds across ALL sequences (pages of 64).  A static batch
                                 (dots_prefill / dots_generate) reserves prompt + generation cap per sequence; a slot sequence
                                 (dots_slots_prefill) reserves its prompt + 64 tokens and takes further pages on demand as it grows
                                 (dots_slots_decode).  0 = max_batch * max_seq_len (never refuses) */
    int32_t fp8_weights;      /* != 0: dots_finalize_weights quantises every ViT-block / merger / LM linear and the lm_head to OCP e4m3 with
                                 one fp32 scale per output channel (scale = max|row| / 448; csrc/quant.hip).  The decode step streams
                                 the e4m3 bytes (half the HBM traffic) against bf16 activations; the ViT / prefill GEMMs quantise
                                 their input activations per token the same way and run on the fp8 MFMA (W8A8).  Every quantised
                                 linear needs N % 256 == 0 and K % 64 == 0.  Embedding table, patch embedding, norms and biases
                                 stay bf16.  (BASELINE configs[4]) */
    int32_t _reserved;
} DotsConfig;

/* Per-phase device time of the last dots_generate / dots_vit_forward / ... call, measured with
 * HIP events on the engine's stream (what bench.py's roofline legs read). */
typedef struct DotsStats {
    float vit_ms, prefill_ms, decode_ms, total_ms;
    float vit_attn_ms;        /* sum of the ViT flash-attention launches */
    int32_t vit_attn_launches;
    float vit_gemm_ms;        /* sum of the ViT GEMM launches */
    int32_t decode_steps;
    int64_t vit_patches, prefill_tokens, new_tokens;
    double vit_attn_flops;    /* algorithmic: 4*N_i^2*E per layer summed over images */
    double vit_flops;         /* SURVEY §8(d) ViT formula */
    double prefill_flops;
    double decode_bytes;      /* steps*W + sum ctx*kv_bytes_per_token */
} DotsStats;

/* ---- lifecycle ------------------------------------------------------------------------ */
int dots_create(const DotsConfig* cfg, int device, DotsEngine** out);
void dots_destroy(DotsEngine* e);
const char* dots_last_error(DotsEngine* e);
/* HIP stream of the handle (hipStream_t as void*), for callers that record their own events. */
void* dots_stream(DotsEngine* e);

/* One call per checkpoint tensor, named as in the HF state dict the reference loads
 * (parser.py:68-74).  `data` is a host pointer; dtype bf16/f16/f32 (converted to bf16). */
int dots_load_weight(DotsEngine* e, const char* name, const void* data_host, int dtype,
                     const int64_t* shape, int ndim);
/* Verifies every tensor the config requires is present, builds fused/packed device copies. */
int dots_finalize_weights(DotsEngine* e);

/* ---- the hot path --------------------------------------------------------------------- */
/* Replaces DotsVisionTransformer.forward(pixel_values, grid_thw) (HF-hub modeling_dots_vision.py,
 * called inside model.generate at parser.py:110).  pixel_values f32 [total_patches, C*T*P*P],
 * grid_thw int64 [n_img,3] (host).  out_embeds_dev: bf16 [total_patches/merge^2, hidden] or NULL
 * (result then stays in the engine for the following dots_prefill). */
int dots_vit_forward(DotsEngine* e, const float* pixel_values, int pixel_values_on_device,
                     int64_t total_patches, const int64_t* grid_thw_host, int n_img,
                     void* out_embeds_dev);

/* Software pipelining across page batches (no counterpart in the reference's HF path, which is strictly sequential; vLLM overlaps
 * requests in its scheduler).  dots_vit_prefetch runs the tower of the NEXT batch asynchronously on a side stream that is masked to
 * the upper (256 - 128) CUs — an equal share of every XCD — and returns at once; while it runs, dots_generate replays its decode
 * graph on a stream masked to the lower 128 CUs (the two partitions then work side by side: two unmasked streams were measured to
 * time-slice the chip with no overlap at all), and on the whole chip again once the tower is done.  dots_vit_take_prefetched makes
 * the main stream wait for the tower and puts its rows in place for the next dots_prefill / dots_slots_prefill, or for
 * dots_generate with n_img = -1.  Order per batch k: take(k) -> [dots_preprocess_image(k+1)] -> prefetch(k+1) -> generate(k, n_img = -1).
 * after_prefill != 0: the tower is launched behind the NEXT prefill (dots_prefill / dots_generate / dots_slots_prefill) instead of at once,
 * so that it shares the chip with the latency-bound decode loop only, not with the MFMA-bound prefill (or at dots_vit_take_prefetched if no
 * prefill comes by).  pixel_values must stay valid until the rows are taken.  Results are bit-identical to the sequential calls.
 * Environment DOTS_OCR_OVERLAP_DEC_CUS (multiple of 8, default 128) sets the decode partition. */
int dots_vit_prefetch(DotsEngine* e, const float* pixel_values, int pixel_values_on_device, int64_t total_patches,
                      const int64_t* grid_thw_host, int n_img, int after_prefill);
int dots_vit_take_prefetched(DotsEngine* e);
/* *ready = 1 when the tower of the prefetched batch has finished (dots_vit_take_prefetched then makes nothing wait), 0 while it runs or has
 * not been launched yet (after_prefill).  A serving loop polls it between decode chunks and takes the batch only when its rows exist, so
 * that the sequences already decoding never queue behind a tower (dots_ocr_amd/scheduler.py).  DOTS_E_STATE without a pending prefetch. */
int dots_vit_prefetch_ready(DotsEngine* e, int* ready);

/* Replaces prepare_inputs_embeds + the prefill forward of Qwen2ForCausalLM.
 * Packed prompts: input_ids int32 [sum(prompt_lens)] (host), slot i of the batch gets prompt i.
 * Vision rows from the preceding dots_vit_forward are scattered at image_token_id positions. */
int dots_prefill(DotsEngine* e, const int32_t* input_ids_host, const int32_t* prompt_lens_host, int B);

/* One greedy decode step for the B prefilled sequences. */
int dots_decode_step(DotsEngine* e);

/* Replaces model.generate(**inputs, max_new_tokens=N) with do_sample=False (parser.py:110):
 * ViT over all images, prefill, greedy decode until every sequence hit an EOS id or N tokens.
 * out_ids int32 [B, max_new_tokens] (host, new tokens only), out_lens int32 [B].
 * n_eos == 0 disables EOS (fixed-length timing runs).
 * n_img == -1: skip the tower, the vision rows are the ones dots_vit_take_prefetched put in place. */
int dots_generate(DotsEngine* e, const int32_t* input_ids_host, const int32_t* prompt_lens_host, int B,
                  const float* pixel_values, int pixel_values_on_device, int64_t total_patches,
                  const int64_t* grid_thw_host, int n_img, int max_new_tokens,
                  const int32_t* eos_ids_host, int n_eos, int32_t* out_ids_host, int32_t* out_lens_host);

/* Replaces the image half of processor.__call__ (parser.py:99-105 -> Qwen2-VL image processor: Pillow BICUBIC resize to
 * (rh, rw) = smart_resize(h, w), x 1/255, (x - mean)/std, patchify) on the GPU, bit-identical to the host path.
 * rgb: uint8 [h, w, 3].  The per-axis tap tables are Pillow's 22-bit fixed-point coefficients, built on the host
 * (coef int32 [out, ksize], bounds int32 [out, 2] = first input index, tap count); pass NULL tables for an axis that
 * is not resized.  out_pixel_values_dev: float32 [(rh/P)*(rw/P), 3*P*P] on the device, ready for dots_vit_forward. */
int dots_preprocess_image(DotsEngine* e, const uint8_t* rgb, int rgb_on_device, int h, int w, int rh, int rw,
                          const int32_t* hcoef_host, const int32_t* hbounds_host, int hksize,
                          const int32_t* vcoef_host, const int32_t* vbounds_host, int vksize,
                          const float* mean3_host, const float* std3_host, float rescale, float* out_pixel_values_dev);

/* Token selection for the following prefill / decode / generate calls.  temperature == 0 (default): greedy arg max.
 * temperature > 0: sample from softmax(logits / temperature) restricted to the top_p nucleus — the sampling
 * parameters the reference passes to its vLLM backend (parser.py:27-28, model/inference.py:38-43).  Reproducible
 * from `seed` (counter-based: seed, batch slot, position). */
int dots_set_sampling(DotsEngine* e, float temperature, float top_p, uint64_t seed);
/* Launch plan of the decode step (results are bit-identical under either plan).  0 (default) = chosen by where the step runs: the
 * whole-chip plan (qkv / o_proj / down_proj as 8-row half tiles: 256 / 192 / 192 workgroups; one gate|up workgroup per tile pair), or —
 * while the step is replayed on the decode CU partition beside a prefetched vision tower (dots_vit_prefetch) — the PARTITION plan: the
 * projections as whole 16-row tiles (half as many workgroups) and gate|up as one resident round of workgroups that walk the tile pairs.
 * 1 = the partition plan on every step (tests, A/B runs; slower on the whole chip).  Environment DOTS_OCR_DECODE_PLAN sets the default.
 * Batches above 16 rows run the WIDE qkv / projection kernels under either plan (round 5: every batch tile in one workgroup, one dispatch
 * round sized for the CUs of the stream; DOTS_OCR_DEC_WIDE=0 = the per-tile kernels, same bits).
 * Round 5: + 2 = the STREAMING decode-attention kernel (one resident workgroup per CU walks the (row, kv head, split) items, pages arrive by
 * LDS-DMA one item ahead) wherever it is legal, + 4 = always one workgroup per item (also the default: the streaming kernel measured
 * slower at every batch size, profiles/r05_decode_attn_stream_ab.txt).  Same bits either way. */
int dots_set_decode_plan(DotsEngine* e, int plan);
/* Tower tail of the vision prefetch (round 5).  A prefetched tower (dots_vit_prefetch) runs on the upper CU partition beside the decode
 * loop; its LAST `tail` blocks (and the merger) run on the whole chip instead, so that the decode partition does not idle when the decode
 * loop of a step drains before the tower.  set = -1: adaptive — per launch, from the events of the previous one: the partition
 * part is sized to end when the last decode chunk did (a decode loop that outlasts the tower gives 0; for pipelines whose decode work per
 * admission is finite, as bench.py's); set >= 0: that many blocks on every launch (0 = off, the default); set = -2: leave it as it is.  Environment DOTS_OCR_TOWER_TAIL_LAYERS = the initial `set`.  *now (may be NULL) receives the tail of the tower
 * launched last.  Results do not depend on it (the same kernels in the same order). */
int dots_tower_tail(DotsEngine* e, int set, int* now);
/* Launch plan of the 256-wide bf16 MFMA GEMM behind the vision tower and the prefill (results are bit-identical under either plan:
 * the same MFMAs in the same k order per output element).  0 = 8 waves per workgroup, two per SIMD running half a K sub-tile apart
 * (round 2); 1 (default) = 4 waves, one per SIMD owning a 128 x 128 output block in 256 accumulator registers, K tiles of 64 streamed by
 * LDS-DMA through a 5-unit ring, one barrier per 64 MFMAs (round 5).  PROCESS-wide (the kernels are shared by every engine of the process);
 * environment DOTS_OCR_GEMM_PLAN sets the default. */
int dots_set_gemm_plan(DotsEngine* e, int plan);

/* ---- Continuous batching (the serving loop the reference delegates to vLLM: README "vLLM inference", parser.py:138-166
 * fires one request per page at it and the server keeps its batch full).  The engine's max_batch KV slots are
 * independent sequences: a finished sequence is read out, its slot released and refilled by a new prefill while the other
 * slots keep decoding.  Any static-batch call (dots_prefill / dots_generate) resets every slot.
 *
 * dots_slots_reset   enter slot mode with every slot free and every KV page back in the pool (a serving loop calls it once at start:
 *                    the pages of an earlier static batch would otherwise count as used until the first dots_slots_prefill); a prefetched
 *                    vision batch that was never taken is dropped.
 * dots_set_eos       stop tokens for the slot calls.
 * dots_slots_prefill n new sequences (packed ids, like dots_prefill) into the free slots `slots[i]`, each with its own
 *                    cap on generated tokens; if the prompts hold image tokens, run dots_vit_forward for exactly these
 *                    sequences first.  Selects each new sequence's first token.
 * dots_slots_decode  n_steps decode steps over all occupied slots (one captured graph per (rows, kv-split) shape).
 *                    Finished sequences idle in place: their context is frozen and nothing more is appended.  Before the steps
 *                    every running sequence is given the KV pages its next n_steps positions need (on-demand paging); if the pool
 *                    is dry the sequence keeps its pages and its generation cap is lowered to what they hold — it finishes there
 *                    (reason "length", as HF generate does at the context capacity); dots_slot_capacity reports the lowered cap.
 * dots_slots_poll    finished[b] = -1 free / 0 running / 1 finished, out_lens[b] = tokens generated so far; both
 *                    int32 [max_batch].  Synchronises the stream.
 * dots_slot_read     copies min(n, capacity) generated ids of one occupied slot, *n_out = n.
 * dots_slot_release  marks the slot free. */
int dots_slots_reset(DotsEngine* e);
int dots_set_eos(DotsEngine* e, const int32_t* eos_ids_host, int n_eos);
int dots_slots_prefill(DotsEngine* e, const int32_t* slots_host, int n, const int32_t* input_ids_host,
                       const int32_t* prompt_lens_host, const int32_t* max_new_tokens_host);
int dots_slots_decode(DotsEngine* e, int n_steps);
int dots_slots_poll(DotsEngine* e, int32_t* finished_host, int32_t* out_lens_host);
int dots_slot_read(DotsEngine* e, int slot, int32_t* out_ids_host, int capacity, int32_t* n_out);
int dots_slot_release(DotsEngine* e, int slot);
/* Paged KV pool: pages of 64 tokens in total / currently free (an admission policy checks this before dots_slots_prefill,
 * which refuses with DOTS_E_CAPACITY when prompt + 64 tokens of each new sequence do not fit). */
int dots_kv_pool_info(DotsEngine* e, int32_t* total_pages, int32_t* free_pages);
/* Pages an occupied slot owns and its current limit on prompt + generated tokens (prompt + max_new_tokens unless the pool ran dry). */
int dots_slot_capacity(DotsEngine* e, int slot, int32_t* pages_owned, int32_t* token_limit);

/* fp32 logits [B, vocab] of the most recent prefill/decode step (tolerance checks). */
int dots_get_logits(DotsEngine* e, float* out_host);
/* Teacher forcing for per-step logit comparisons: overwrite the token the next decode step feeds. */
int dots_set_next_tokens(DotsEngine* e, const int32_t* tokens_host, int B);
/* Tokens chosen by the most recent prefill/decode step, int32 [B]. */
int dots_get_last_tokens(DotsEngine* e, int32_t* out_host);
int dots_get_stats(DotsEngine* e, DotsStats* out);
/* Debug / parity tooling (tools/layer_error_trace.py): keep a copy of the bf16 residual stream after every ViT block and every
 * LM prefill layer of the following dots_vit_forward / dots_prefill calls.  capacity_elems = 0 switches the capture off.
 * dots_debug_read_hidden: which = 0 ViT block `layer` -> [patches, v_embed_dim], which = 1 LM layer `layer` -> [tokens, hidden]. */
int dots_debug_capture_hidden(DotsEngine* e, int64_t capacity_elems);
int dots_debug_read_hidden(DotsEngine* e, int which, int layer, void* out_host, int64_t* rows_out);
int dots_synchronize(DotsEngine* e);

/* ---- device memory helpers (so a binding needs no other GPU library) ------------------- */
int dots_dev_alloc(DotsEngine* e, int64_t bytes, void** out_dev);
int dots_dev_free(DotsEngine* e, void* dev);
int dots_memcpy_h2d(DotsEngine* e, void* dst_dev, const void* src_host, int64_t bytes);
int dots_memcpy_d2h(DotsEngine* e, void* dst_host, const void* src_dev, int64_t bytes);

/* ---- single-kernel entry points (parity tests call the kernels through these) ---------- */
/* y = rmsnorm(x) * w : bf16 [rows, dim]; fp32 statistics; two roundings as modeling_qwen2.py:246-252 */
int dots_op_rmsnorm(DotsEngine* e, const void* x_dev, const void* w_dev, void* y_dev, int64_t rows, int dim, float eps);
int dots_op_layernorm(DotsEngine* e, const void* x_dev, const void* w_dev, const void* b_dev, void* y_dev,
                      int64_t rows, int dim, float eps);
/* C[M,N] = epilogue(A[M,K] @ W[N,K]^T + bias): bf16 in, fp32 accumulate.
 * epilogue: 0 none, 1 += residual (bf16 [M,N], may alias C), 2 SwiGLU (W rows interleaved in
 * 32-row gate/up groups, C is [M,N/2]), 3 exact GELU, 4 fp32 output. */
int dots_op_gemm(DotsEngine* e, const void* A_dev, const void* W_dev, const void* bias_dev,
                 const void* residual_dev, void* C_dev, int64_t M, int N, int K, int epilogue, const float* colscale_dev);
/* colscale_dev: NULL, or fp32 [N] multiplied into column n of the accumulator before the bias (the per-output-channel scale of
 * fp8 weights; W then holds the quantised values).
 * dots_op_quant_fp8: W bf16 [N,K] -> bf16(e4m3(W[n][:] / scale[n])) in place, scale_out[n] = max|W[n][:]| / 448 (1 for a zero row). */
int dots_op_quant_fp8(DotsEngine* e, void* w_inout_dev, float* scale_out_dev, int64_t N, int K);
/* The ViT / prefill GEMM of an fp8_weights engine: A bf16 [M,K] is quantised per token and W bf16 [N,K] per output channel to e4m3
 * (copies), then C = epilogue((Aq Wq^T) * a_scale[m] * w_scale[n] + bias) on the fp8 MFMA.  N % 256 == 0, K % 64 == 0; epilogues 0-3. */
int dots_op_gemm_fp8(DotsEngine* e, const void* A_dev, const void* W_dev, const void* bias_dev, const void* residual_dev, void* C_dev,
                     int64_t M, int N, int K, int epilogue);
/* Flash attention over packed sequences.  q [Hq, T, 128], k [Hkv, T, 128] bf16 (head-major),
 * vt [Hkv, 128, Tpad] (V transposed, every sequence padded to 64 keys, keys permuted inside
 * 16-groups as csrc/attn_prefill.hip documents), cu_seqlens int32 [n_seq+1] (host).
 * out bf16 [T, Hq*128]. */
int dots_op_flash_attn(DotsEngine* e, const void* q_dev, const void* k_dev, const void* vt_dev, void* out_dev,
                       const int32_t* cu_seqlens_host, int n_seq, int Hq, int Hkv, int causal, float scale);
/* Host-only (no device, no engine): how the flash-attention work list of a packed batch of sequences of lens[0 .. n_seq) patches and Hq heads
 * is cut across the 8 XCDs — base8[x] / cnt8[x] = the contiguous chunk of (sequence, head, query block) items XCD x walks, cost8[x] = its KV
 * tiles (csrc/kernels.h: XcdPlan; equal COST per XCD, not equal count: tests/test_cabi_cpu.py checks the balance on BASELINE configs[3]'s
 * page mix).  Returns the number of items, or a negative DOTS_E_*. */
int dots_plan_flash_xcd(const int32_t* lens, int n_seq, int Hq, int32_t* base8, int32_t* cnt8, int64_t* cost8);
/* Splits a packed qkv GEMM output [T, (Hq+2*Hkv)*128] into rope'd q/k and transposed v in the
 * layouts dots_op_flash_attn consumes.  rope2d != 0: vision 2-D rope from pos [T,2] int32;
 * else 1-D rope with positions pos [T] int32 and base theta. */
int dots_op_qkv_rope_split(DotsEngine* e, const void* qkv_dev, void* q_dev, void* k_dev, void* vt_dev,
                           const int32_t* cu_seqlens_host, int n_seq, const int32_t* pos_host,
                           int Hq, int Hkv, int rope2d, float theta);
/* x [T, K] bf16 @ w [(Hq+2*Hkv)*128, K]^T (+ bias) -> rope'd head-major q / k and transposed v (the layouts dots_op_flash_attn consumes), as the
 * prefill passes run it: fused != 0 = the GEMM's rope epilogue + the v transposition, fused == 0 = GEMM, then dots_op_qkv_rope_split's kernel.
 * Both give the same bits.  qkv_ws: [T, (Hq+2*Hkv)*128] bf16 workspace.  DOTS_E_INVALID if fused != 0 and no fused kernel serves the shape. */
int dots_op_qkv_proj_rope(DotsEngine* e, const void* x_dev, const void* w_dev, const void* bias_dev, void* qkv_ws_dev, void* q_dev, void* k_dev, void* vt_dev,
                          const int32_t* cu_seqlens_host, int n_seq, const int32_t* pos_host, int K, int Hq, int Hkv, int rope2d, float theta, int fused);
/* ---- single kernels of the decode step, at caller-chosen dimensions.  All tensors are device pointers in
 * the ROW-MAJOR layouts of the HF state dict / of a plain [B, features] activation; the MFMA fragment-order packing the
 * decode step uses (csrc/decode_layout.h) is applied inside with the engine's own pack kernels.  B <= 64.
 *
 * dots_op_dec_qkv      h [B,H] -> RMSNorm(ln_w) -> fused qkv projection wqkv [(Hq+2Hkv)*128, H] + bias -> 1-D RoPE at position
 *                      ctx_len[b] -> q_out bf16 [B, Hq*128]; the new key / value row of every sequence is appended to its
 *                      page: pool_layer [pages][Hkv][K|V][8192] bf16 (csrc/decode.hip header), block_table int32
 *                      [B, max_pages], ctx_len int32 [B] (tokens already in the cache).
 * dots_op_decode_attn  q bf16 [B, Hq*128] against the paged cache holding ctx_len[b] + 1 tokens per sequence (split-KV
 *                      kernel + combine kernel, KV split = the engine constant derived from max_seq_len) -> out bf16 [B, Hq*128].
 * dots_op_dec_proj     h_inout [B,N] += x [B,K] @ w [N,K]^T   (o_proj / down_proj with the residual add).
 * dots_op_dec_gateup   act_out [B,I] = silu(g) * u with g|u = RMSNorm(h) @ gate_w|up_w [I,H]^T.
 * dots_op_dec_lmhead   logits_out fp32 [B,V] = RMSNorm(h) @ w [V,H]^T.
 * fp8 != 0: the weight is quantised (dots_op_quant_fp8 on a copy), packed as e4m3 fragments and streamed by the fp8 instantiation
 * of the kernel — the path a DotsConfig.fp8_weights engine decodes with. */
int dots_op_dec_qkv(DotsEngine* e, const void* h_dev, const void* ln_w_dev, const void* wqkv_dev, const void* bias_dev,
                    const int32_t* ctx_len_dev, const int32_t* block_table_dev, int max_pages, void* pool_layer_dev, void* q_out_dev,
                    int B, int H, int Hq, int Hkv, float eps, float rope_theta, int fp8);
int dots_op_decode_attn(DotsEngine* e, const void* q_dev, const void* pool_layer_dev, const int32_t* ctx_len_dev,
                        const int32_t* block_table_dev, int max_pages, void* out_dev, int B, int Hq, int Hkv, int max_seq_len);
int dots_op_dec_proj(DotsEngine* e, const void* x_dev, const void* w_dev, void* h_inout_dev, int B, int N, int K, int fp8);
int dots_op_dec_gateup(DotsEngine* e, const void* h_dev, const void* ln_w_dev, const void* gate_w_dev, const void* up_w_dev, void* act_out_dev,
                       int B, int H, int I, float eps, int fp8);
int dots_op_dec_lmhead(DotsEngine* e, const void* h_dev, const void* ln_w_dev, const void* w_dev, void* logits_out_dev, int B, int H, int V,
                       float eps, int fp8);

/* MFMA fragment-layout / LDS-DMA probe (csrc/probe_mfma.hip; tests/test_mfma_layout.py). */
int dots_probe_mfma(int which, const void* A, const void* Bt, void* D, void* stream);
/* Device-wide barrier probe inside one persistent kernel (csrc/probe_sync.hip): n_barriers bounded-spin barriers over
 * n_wg workgroups, each followed by cross-workgroup reads of freshly published data.  mode: 0 no fences, 1 agent-scope
 * fences, 2 nontemporal accesses, 3 agent-scope atomic accesses.  *ms_out = kernel time, stats_out[0] = stale reads,
 * stats_out[1] = barrier timeouts (the spin is bounded, the probe cannot hang). */
int dots_probe_grid_barrier(int n_wg, int threads, int n_barriers, int mode, int lds_bytes, float* ms_out, int32_t* stats_out);
/* CU-mask probe: runs n_wg workgroups on a stream created with hipExtStreamCreateWithCUMask(mask) (NULL: all CUs) and
 * returns each workgroup's raw HW_ID / XCC_ID registers, uint32 [n_wg][2]. */
int dots_probe_cu_mask(const uint32_t* mask, int words, int n_wg, int threads, int lds_bytes, uint32_t* ids_out_host);

#ifdef __cplusplus
}
#endif
#endif /* DOTS_OCR_HIP_H */
