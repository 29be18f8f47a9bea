/* llx.h - C ABI of libllx_hip.so: the MI355X (gfx950) kernels behind the llama-x training hot path.
 *
 * Drop-in boundary: the reference (gau-nernst/llama-x) is pure Python and reaches the
 * device through PyTorch / Triton calls inside its `modelling` and `subclasses` packages.  This library is what the
 * Python side of those packages binds (ctypes) instead; each entry point names the reference call site it replaces
 * (paths relative to the reference root).
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless stated otherwise;
 *   - bf16 tensors are passed as `const void*` to raw 16-bit storage, row-major, last dimension contiguous;
 *     `ld*` / `*_ss` / `*_sb` are strides in ELEMENTS;
 *   - the caller owns every buffer (PyTorch's caching allocator); the library never allocates, frees, retains or
 *     synchronises; workspaces are caller-provided (size queries are host functions);
 *   - every launcher takes the stream to launch on (`hipStream_t`, passed as void* from ctypes) and is re-entrant;
 *   - return value 0 = success; negative = error (-1 bad argument, -2 launch failure, -3 unsupported); the message is
 *     in llx_last_error_string() (thread-local).  No exceptions, no abort.
 */
#ifndef LLX_H
#define LLX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* llx_stream_t; /* hipStream_t */

/* ---- library ------------------------------------------------------------------------------------------------ */
int llx_version(void);                                   /* 105 = 0.1.5 */
const char* llx_last_error_string(void);                 /* thread-local, valid until the next failing call */
int llx_device_info(int device, char* name, int len);    /* returns CU count, fills gcn arch name */

/* ---- RMSNorm: nn.RMSNorm(D, eps=1e-5) at modelling/llama.py:158,160,182 (called :172,173,216) ----------------- */
int llx_rmsnorm_fwd(const void* x, const void* w, void* y, float* rstd, int64_t rows, int64_t dim, float eps, llx_stream_t s);
/* the same, also emitting quantize_int8_rowwise(y) (q int8 [rows,dim], row stride ldq; qscale bf16 [rows]) for the dynamic-int8-activation
 * linears that follow the norm (subclasses/int8.py:110-113): bit-identical to llx_quantize_int8_rowwise on y, one pass */
int llx_rmsnorm_fwd_quant(const void* x, const void* w, void* y, float* rstd, void* q, int64_t ldq, void* qscale, int64_t rows, int64_t dim,
                          float eps, llx_stream_t s);
int64_t llx_rmsnorm_bwd_workspace_bytes(int64_t rows, int64_t dim);
int llx_rmsnorm_bwd(const void* dy, const void* x, const void* w, const float* rstd, void* dx, void* dw /*nullable*/,
                    int dw_accumulate, void* workspace, const void* dres /*nullable: residual-branch gradient added to dx*/,
                    int64_t rows, int64_t dim, llx_stream_t s);

/* ---- bf16 MFMA GEMM, C[M,N] = A[M,K].B[N,K]^T (+ A2[M,K2].B2[N,K2]^T): every F.linear of the layer
 *      (modelling/llama.py:118-120,140,152,216), its data gradients (on a transposed weight image), the LoRA adapter
 *      as K-extension (modelling/lora.py:43) and the audio convolutions as implicit GEMMs (modelling/audio.py:26-31).
 *      epilogue: 0 none | 1 + E[M,N] (ld = lde) | 2 + bias E[N] | 3 gelu(+bias E[N]) | 4 * colscale E[N]
 *      (weight-only int8, subclasses/int8.py:118) | 6 SwiGLU backward: the product is dh = dL/d(silu(g)*u) of the feed-forward
 *      (modelling/llama.py:150); E = gate|up activations [M,2N], C = dg|du [M,2N], dh is not stored.
 *      | 7 SwiGLU forward: B = [W_gate; W_up] (N = 2I, I % 128 == 0), C = gate|up [M,N] as usual, E = OUTPUT h [M,I] = silu(g)*u
 *      (row stride lde); a tile covers 128 gate columns and the matching 128 up columns.
 *      K, K2 multiples of 64; N multiple of 8. ------------------------------------------------------------------- */
int llx_gemm_nt_bf16(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int64_t M, int64_t N, int64_t K,
                     const void* A2, int64_t lda2, const void* B2, int64_t ldb2, int64_t K2, int epilogue, const void* E, int64_t lde,
                     llx_stream_t s);
/* llx_gemm_nt_bf16 over the first *m_valid rows only (m_valid: DEVICE int32, read by the kernel, so the launch is capturable): row
 * tiles of 256 that start at or after *m_valid are skipped, rows of C from *m_valid to the end of that tile are computed from
 * whatever A holds there.  The LM head of modelling/llama.py:216-218 over the positions whose label is not ignore_index. */
int llx_gemm_nt_bf16_rows(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int64_t M, int64_t N, int64_t K,
                          const void* A2, int64_t lda2, const void* B2, int64_t ldb2, int64_t K2, int epilogue, const void* E, int64_t lde,
                          const int32_t* m_valid, llx_stream_t s);
/* Split-K form for a product with few output tiles and a long contraction (d hidden = d logits . W of the LM head, modelling/llama.py:216:
 * 16 x 16 tiles, K = 128256): partial[s][M][N] fp32 (row stride N) for `splits` equal K ranges from ONE launch over (range, row tile,
 * column tile); m_valid (nullable) as in llx_gemm_nt_bf16_rows.  K % (64 * splits) == 0.
 * llx_splitk_combine: out[i] = bf16(scale[0] * bf16(sum_s partial[s][inv ? inv[i] : i])), zero rows where inv[i] < 0 (inv, scale nullable). */
int llx_gemm_nt_bf16_splitk(const void* A, int64_t lda, const void* B, int64_t ldb, float* partial, int64_t M, int64_t N, int64_t K, int splits,
                            const int32_t* m_valid, llx_stream_t s);
int llx_splitk_combine(const float* partial, int splits, int64_t M, int64_t N, const int32_t* inv, const float* scale, void* out, int64_t ld_out,
                       llx_stream_t s);
/* The q|k|v projection with apply_rope (modelling/llama.py:118-125, 63-73) in the epilogue: columns [0, rope_cols) of C (whole
 * 128-wide heads: q then k) are rotated with the fp32 table [>= rope_S, 64, 2]; row m of C is sequence position m % rope_S.
 * Bit-identical to llx_gemm_nt_bf16(epilogue 0) followed by llx_rope. */
int llx_gemm_nt_bf16_rope(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int64_t M, int64_t N, int64_t K,
                          const void* A2, int64_t lda2, const void* B2, int64_t ldb2, int64_t K2, const float* rope_table, int64_t rope_S,
                          int64_t rope_cols, llx_stream_t s);

/* ---- the weight gradient of a dense linear, dW[out,in] = dy[T,out]^T . x[T,in] (autograd of F.linear for the weights the scripts leave
 *      trainable: tok_embeddings / norm / output by default, train_metamathqa.py:177-180; the Conv1d weights of modelling/audio.py:26-31 as
 *      an implicit GEMM): C[N1,N2] = A[M,N1]^T . B[M,N2], operands read as they lie (token-major rows, row-strided views allowed), no
 *      transposed copies.  N1, N2 multiples of 8; any M. ------------------------------------------------------------------------------- */
int llx_gemm_tn_bf16(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int64_t M, int64_t N1, int64_t N2,
                     llx_stream_t s);
/* ... over the first min(M, *m_valid) rows only (m_valid: device int32, nullable): the weight gradient of a trainable LM head over the
 * compacted labelled rows - F.cross_entropy's ignore_index rows have zero gradient rows (modelling/llama.py:216-218). */
int llx_gemm_tn_bf16_rows(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int64_t M, int64_t N1, int64_t N2,
                          const int32_t* m_valid, llx_stream_t s);

/* ---- torchao::int8_mm_dequant(A, B, A_scale, B_scale) - subclasses/int8_mm.py:121-149 (Triton kernel :50-118).
 *      A int8 [M,K]; B passed as its K-contiguous rows [N,K] (= the reference's int_data.T view, strides (1,K));
 *      scales bf16; C bf16 = (int32 acc) * a_scale[m] * b_scale[n], one rounding.  K multiple of 128. ----------- */
int llx_int8_mm_dequant(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int64_t M, int64_t N, int64_t K,
                        const void* a_scale, const void* b_scale, llx_stream_t s);
/* fp32 scales -> fp32 C (the op returns dtype = A_scale.dtype: subclasses/int8_mm.py:126,136,143); ldc in fp32 elements. */
int llx_int8_mm_dequant_f32(const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int64_t ldc, int64_t M, int64_t N, int64_t K,
                            const float* a_scale, const float* b_scale, llx_stream_t s);
/* The same with the neighbours of its call sites fused in (int8 base, dynamically quantised activations - subclasses/int8.py:110-118):
 * C = epilogue( bf16( bf16(int8_mm_dequant(A, B, a_scale, b_scale)) + A2[M,K2].B2[N,K2]^T ) ).  A2/B2 (nullable; bf16, K2 % 64 == 0):
 * the LoRA adapter (modelling/lora.py:43) - the int32 accumulators are dequantised in place, the extension accumulates on top in fp32.
 * epilogue: 0 none | 1 + E[M,N] (ld = lde) | 7 SwiGLU forward (E = OUTPUT h) | 8 RoPE on columns [0, rope_cols), as llx_gemm_nt_bf16(_rope). */
int llx_int8_mm_dequant_ext(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int64_t M, int64_t N, int64_t K,
                            const void* a_scale, const void* b_scale, const void* A2, int64_t lda2, const void* B2, int64_t ldb2, int64_t K2,
                            int epilogue, const void* E, int64_t lde, const float* rope_table, int64_t rope_S, int64_t rope_cols,
                            llx_stream_t s);

/* ---- quantize_int8_rowwise - subclasses/int8.py:10-16 (weights once, activations per forward when dynamic). ---- */
int llx_quantize_int8_rowwise(const void* x, int64_t ldx, void* q, int64_t ldq, void* scale, int64_t rows, int64_t cols, int is_f32,
                              llx_stream_t s);

/* ---- attention: F.scaled_dot_product_attention(..., enable_gqa=True) and flex_attention(block_mask=...) at
 *      modelling/llama.py:129-137; mask rule = mask_mod of train_metamathqa.py:67-68 plus the prefix-LM term
 *      (README.md:16): allow(q,k) = (k <= q || k < prefix_len[b]) && (!doc_ids || doc_ids[b,q] == doc_ids[b,k]).
 *      q [B,S,H,128], k/v [B,S,KVH,128] with free batch/sequence strides; lse fp32 [B,H,S] (log2 units).
 *      flags: tile classes built by llx_attn_tile_flags (needed only with doc_ids / prefix_len). ----------------- */
int64_t llx_attn_flags_bytes(int64_t B, int64_t S);
int llx_attn_tile_flags(const int* doc_ids, const int* prefix_len, void* flags, int64_t B, int64_t S, llx_stream_t s);
int llx_attn_fwd(const void* q, int64_t q_sb, int64_t q_ss, const void* k, int64_t k_sb, int64_t k_ss, const void* v, int64_t v_sb,
                 int64_t v_ss, void* o, int64_t o_sb, int64_t o_ss, float* lse, const int* doc_ids, const int* prefix_len,
                 const void* flags, int64_t B, int64_t S, int64_t H, int64_t KVH, int64_t head_dim, float scale, llx_stream_t s);
int64_t llx_attn_bwd_workspace_bytes(int64_t B, int64_t S, int64_t H, int64_t KVH); /* delta + per-head dK/dV partials */
int64_t llx_attn_bwd_ds_bytes(int64_t B, int64_t S, int64_t H); /* optional bf16 dS^T scratch [B,H,Sp,Sp], Sp = S rounded up to 128 */
int llx_attn_bwd(const void* q, int64_t q_sb, int64_t q_ss, const void* k, int64_t k_sb, int64_t k_ss, const void* v, int64_t v_sb,
                 int64_t v_ss, const void* o, int64_t o_sb, int64_t o_ss, const void* d_o, int64_t do_sb, int64_t do_ss, const float* lse,
                 float* delta /* fp32 workspace, llx_attn_bwd_workspace_bytes() */, void* dq, int64_t dq_sb, int64_t dq_ss, void* dk, int64_t dk_sb, int64_t dk_ss,
                 void* dv, int64_t dv_sb, int64_t dv_ss, const int* doc_ids, const int* prefix_len, const void* flags,
                 const float* rope /* nullable fp32 [>=S,64,2]: dq, dk leave already multiplied by apply_rope's transpose */,
                 void* ds /* nullable scratch of llx_attn_bwd_ds_bytes(): dS^T makes one round trip through it and each of the five products
                             (S, dP, dV, dK, dQ) is computed once; without it the dQ kernel recomputes S and dP */,
                 int64_t B, int64_t S, int64_t H, int64_t KVH, int64_t head_dim, float scale, llx_stream_t s);

/* ---- dense-mask attention forward (inference / KV-cache path): SDPA(q,k,v,mask,is_causal=False,enable_gqa=True) at
 *      modelling/llama.py:126-127,135-137 with mask = causal_mask[None,None,input_pos] (:194,:205).  q [B,H,Sq,128],
 *      k/v [B,KVH,Skv,128] (e.g. the KVCache buffers :79-81), mask bool with broadcast strides; forward only. ------- */
int llx_attn_dense_fwd(const void* q, int64_t q_sb, int64_t q_sh, int64_t q_ss, const void* k, int64_t k_sb, int64_t k_sh, int64_t k_ss,
                       const void* v, int64_t v_sb, int64_t v_sh, int64_t v_ss, void* o, int64_t o_sb, int64_t o_sh, int64_t o_ss,
                       const void* mask, int64_t m_sb, int64_t m_sh, int64_t m_sq, int64_t B, int64_t H, int64_t KVH, int64_t Sq, int64_t Skv,
                       int64_t head_dim, float scale, llx_stream_t s);

/* ---- decode path: a few query tokens (M <= 4) against the KV cache - modelling/llama.py:76-90 (KVCache), :126-127,:135-137 (cached
 *      K/V through SDPA with the row-gathered causal mask), :189-194,:205-207 (Llama.forward with input_pos).  Every linear is a
 *      weight stream: all CUs read weight rows once (llx_gemv_bf16), the call sites' neighbours ride in its prologue / epilogue. ---- */
/* out = epilogue( [rmsnorm(x) | x][M, K] . [W0; W1; W2]^T ): F.linear at modelling/llama.py:118-120,140,152,216 for M <= 4 rows.
 * W_s [n_s, K] bf16 row-major (ld = ldw_s; w1 / w2 nullable with n = 0; inner n_s % 4 == 0), K % 8 == 0; norm_w (nullable): nn.RMSNorm
 * (:158-160,182) applied to x first.  epilogue 0: out [M, N] | 1: + res [M, N] (:172-173) | 2 (q|k|v): apply_rope (:63-73,122-123;
 * table row = token index in this call, :207) on rows [0, n_q) -> out [M, n_q] and on rows [n_q, n_q + n_k) -> k cache, remaining
 * rows -> v cache, both at input_pos[m] (device int64; KVCache.update :83-90; caches [KVH, Smax, 128] through (head, position)
 * strides) | 3 (gate|up: W0, W1, N = 2 n_0): out [M, n_0] = silu(gate) * up (:150-152).  LoRA (modelling/lora.py:43; bext_s
 * [n_s, rank_s] nullable, t [M, sum rank] bf16 = x . A^T from a previous call, rank % 8 == 0): out += lora_scale * t_s . bext_s[row]. */
int llx_gemv_bf16(const void* w0, int64_t ldw0, int64_t n0, const void* w1, int64_t ldw1, int64_t n1, const void* w2, int64_t ldw2, int64_t n2,
                  const void* x, int64_t ldx, int64_t M, int64_t K, const void* norm_w, float eps, int epilogue, void* out, int64_t ldo,
                  const void* res, int64_t ldr, const float* rope, int64_t n_q, int64_t n_k, void* k_cache, void* v_cache, int64_t c_sh,
                  int64_t c_ss, const int64_t* input_pos, const void* bext0, const void* bext1, const void* bext2, int64_t rank0, int64_t rank1,
                  int64_t rank2, const void* t, int64_t ldt, float lora_scale, llx_stream_t s);
/* *extent (device int) = 1 + the largest key index any of the `rows` bool mask rows (row stride in bytes) allows; 0 if none: the
 * mask of the cached path is data (rows of a tril matrix gathered at input_pos, :194,:205), its extent sizes the decode key ranges. */
int llx_mask_extent(const void* mask, int64_t row_stride, int64_t rows, int64_t Skv, int* extent, llx_stream_t s);
/* KVCache.update (:83-90): cache[b, h, input_pos[l], :] = src[b, h, l, :] for k and v (sources share strides, caches share strides). */
int llx_kv_scatter(const void* k, const void* v, int64_t s_sb, int64_t s_sh, int64_t s_ss, void* k_cache, void* v_cache, int64_t c_sb, int64_t c_sh,
                   int64_t c_ss, const int64_t* input_pos, int64_t B, int64_t KVH, int64_t L, int64_t Smax, int64_t head_dim, llx_stream_t s);
/* SDPA(q, k_cache, v_cache, mask, is_causal=False, enable_gqa=True) (:135-137) for M query tokens with M * H / KVH <= 16: the cache of a
 * kv head is split over `nsplit` workgroups, the heads of its group share every K / V row read; partials merged in a second launch.
 * q / o [B, H, M, 128], caches [B, KVH, Skv, 128] through (batch, head, position) strides; mask bool [.., M, Skv] with broadcast
 * strides; extent nullable (llx_mask_extent); workspace fp32, llx_attn_decode_workspace_bytes(B, H, M, nsplit) bytes. */
int64_t llx_attn_decode_workspace_bytes(int64_t B, int64_t H, int64_t M, int64_t nsplit);
int llx_attn_decode(const void* q, int64_t q_sb, int64_t q_sh, int64_t q_ss, const void* k_cache, const void* v_cache, int64_t c_sb, int64_t c_sh,
                    int64_t c_ss, void* o, int64_t o_sb, int64_t o_sh, int64_t o_ss, const void* mask, int64_t m_sb, int64_t m_sh, int64_t m_sq,
                    const int* extent, float* workspace, int64_t B, int64_t H, int64_t KVH, int64_t M, int64_t Skv, int64_t nsplit,
                    int64_t head_dim, float scale, llx_stream_t s);

/* ---- RoPE: apply_rope at modelling/llama.py:63-73 (in place on the first `nheads` 128-wide heads of each row;
 *      table fp32 [S,64,2] from build_rope :54-60); backward = rotation by -theta. ----------------------------- */
int llx_rope(const void* x, int64_t x_sb, int64_t x_ss, void* y, int64_t y_sb, int64_t y_ss, const float* table, int64_t B, int64_t S,
             int64_t nheads, int64_t head_dim, int backward, llx_stream_t s);

/* ---- SwiGLU: silu(w1 x) * w3 x at modelling/llama.py:152 --------------------------------------------------------- */
int llx_swiglu_fwd(const void* g, int64_t g_ld, const void* u, int64_t u_ld, void* h, int64_t h_ld, int64_t rows, int64_t cols, llx_stream_t s);
int llx_swiglu_bwd(const void* dh, int64_t dh_ld, const void* g, int64_t g_ld, const void* u, int64_t u_ld, void* dg, int64_t dg_ld,
                   void* du, int64_t du_ld, int64_t rows, int64_t cols, llx_stream_t s);

/* ---- embedding: nn.Embedding at modelling/llama.py:180,206 / modelling/audio.py:49; strided destination lets the
 *      gather land behind the audio prefix (replaces torch.cat at modelling/audio.py:63). ------------------------- */
int llx_embedding_fwd(const int64_t* ids, const void* table, void* out, int64_t n_tok, int64_t dim, int64_t vocab, int64_t tok_per_batch,
                      int64_t out_sb, int64_t out_ss, llx_stream_t s);
int llx_embedding_bwd(const int64_t* ids, const void* dy, float* dtable_f32, int64_t n_tok, int64_t dim, int64_t vocab,
                      int64_t tok_per_batch, int64_t dy_sb, int64_t dy_ss, llx_stream_t s);

/* ---- fused cross-entropy: F.cross_entropy(logits.float(), labels) at modelling/llama.py:218, modelling/audio.py:76
 *      (ignore_index -100, mean).  dlogits may alias logits (nullable = forward only). --------------------------- */
int64_t llx_ce_workspace_bytes(int64_t T);
int llx_ce_fwd_bwd(const void* logits, int64_t ld, void* dlogits, int64_t dld, const int64_t* labels, float* loss, void* workspace,
                   int64_t T, int64_t V, llx_stream_t s);
/* ---- LM-head row compaction.  F.cross_entropy(ignore_index=-100) (modelling/llama.py:216-218, modelling/audio.py:74-76): a position
 *      whose label is -100 adds nothing to the loss and has a zero gradient row, so norm -> head -> CE -> d hidden only need the
 *      labelled rows.  They are compacted IN ORDER on the device; the count never visits the host (hipGraph-capturable):
 *        llx_head_compact_index: idx[j] = position of the j-th labelled row (-1 for j >= count), inv[i] = j or -1,
 *                                labels_c[j] = labels[idx[j]] (-100 beyond), count[0] = number of labelled rows
 *        llx_gather_rows:        dst[j] = src[idx[j]] (j < count); zero rows up to the next multiple of 256; later rows untouched
 *        llx_gemm_nt_bf16_rows / llx_ce_fwd_bwd_rows: the head GEMMs / the loss over the compacted rows (rows = count)
 *        llx_scatter_rows:       dst[i] = bf16(scale[0] * src[inv[i]]) where inv[i] >= 0, zero rows elsewhere
 *      Loss and every gradient equal the uncompacted computation (same per-row arithmetic; the loss sums the same row terms). */
int llx_head_compact_index(const int64_t* labels, int32_t* idx, int32_t* inv, int64_t* labels_c, int32_t* count, int64_t T, llx_stream_t s);
int llx_gather_rows(const void* src, int64_t ld_src, const int32_t* idx, const int32_t* count, void* dst, int64_t ld_dst, int64_t T, int64_t D,
                    llx_stream_t s);
int llx_scatter_rows(const void* src, int64_t ld_src, const int32_t* inv, const float* scale /* nullable */, void* dst, int64_t ld_dst,
                     int64_t T, int64_t D, llx_stream_t s);
int llx_ce_fwd_bwd_rows(const void* logits, int64_t ld, void* dlogits, int64_t dld, const int64_t* labels, float* loss, void* workspace,
                        int64_t T, int64_t V, const int32_t* rows, llx_stream_t s);
/* the same loss over a row set walked in chunks (one logits buffer per chunk; labels / workspace cover all T rows): phase bit 0 = first
 * chunk (count the labelled rows of the whole set), bit 1 = last chunk (reduce into loss); values identical to one call over all rows */
int llx_ce_fwd_bwd_part(const void* logits, int64_t ld, void* dlogits, int64_t dld, const int64_t* labels, float* loss, void* workspace, int64_t T,
                        int64_t V, int64_t row0, int64_t nrows, const int32_t* rows, int phase, llx_stream_t s);

/* ---- LoRA skinny contractions (modelling/lora.py:43 and its autograd): T = X.W^T -> [M,64] zero padded;
 *      G = s * U^T.Y with fp32 split partials (deterministic). --------------------------------------------------- */
int llx_skinny_nt(const void* X, int64_t ldx, const void* W, int64_t ldw, void* out, int64_t M, int64_t K, int64_t R,
                  const int32_t* kranges /* host, nullable: {lo,hi} x 4 per 16 rows of a block-diagonal W */, llx_stream_t s);
/* llx_skinny_nt that also writes G[M,K] = bf16(X * colscale[k]) (row stride ldg) from the same read of X: the (grad_output * scale)
 * operand of a weight-only int8 linear's data gradient (subclasses/int8.py:127) rides in the adapter's dy @ lora_b pass. */
int llx_skinny_nt_scaled(const void* X, int64_t ldx, const void* W, int64_t ldw, void* out, int64_t M, int64_t K, int64_t R,
                         const int32_t* kranges, const void* colscale, void* G, int64_t ldg, llx_stream_t s);
/* nn.RMSNorm (modelling/llama.py:158-160) and the adapter's x @ lora_a^T on its output (modelling/lora.py:43) from ONE read of x:
 * y = rmsnorm(x) [M,D], rstd fp32 [M], t = y . W^T [M,64 padded] (W = the group's stacked lora_a [R,D]).  D % 512 == 0, D <= 4096. */
int llx_rmsnorm_skinny_nt(const void* x, const void* g, const void* W, int64_t ldw, void* y, float* rstd, void* t, int64_t M, int64_t D,
                          int64_t R, float eps, llx_stream_t s);
int64_t llx_skinny_tn_workspace_bytes(int64_t M, int64_t N, int64_t R);
int llx_skinny_tn(const void* U, const void* Y, int64_t ldy, void* out, int64_t out_ld, int64_t M, int64_t N, int64_t R, float scale,
                  int transpose_out, int accumulate, void* workspace,
                  const int32_t* segs /* host, nullable: {n_lo, n_hi, r_lo, r_hi} per member of a fused group; member blocks are
                                         then written one after another as contiguous [n, r] matrices */, int seg_count, llx_stream_t s);
/* the two stages of llx_skinny_tn separately: the fp32 split partials of one product, and the second stage of up to 4 products (the
 * adapter gradients of one transformer block) in ONE launch; host arrays of length n, entry i = the arguments of product i */
int llx_skinny_tn_partial(const void* U, const void* Y, int64_t ldy, int64_t M, int64_t N, int64_t R, void* workspace, const int32_t* segs,
                          int seg_count, llx_stream_t s);
/* first stage of up to 4 products in ONE launch (the d lora_b and d lora_a products of a linear group: different operands, ready together) */
int llx_skinny_tn_partial_many(int n, const void* const* U, const void* const* Y, const int64_t* ldy, const int64_t* M, const int64_t* N,
                               const int64_t* R, void* const* workspaces, const int32_t* const* segs, const int* seg_count, llx_stream_t s);
/* ... where product i with upart[i] != NULL also emits, from the Y tiles it stages anyway, the column-tile partial sums of  u = Y_i . Bt_i^T
 * (the adapter's dy @ B of modelling/lora.py:43's backward; Bt_i [R_i, N_i] bf16 row-major with row stride ldb[i], the batched
 * block-diagonal B^T of a fused group; upart[i]: llx_skinny_u_workspace_bytes(M_i, N_i) bytes): dy is read once for dB and u.
 * llx_skinny_u_reduce sums the tiles into u [M, 64] bf16 (columns >= R zero); segs / seg_count as given to the product. */
int64_t llx_skinny_u_workspace_bytes(int64_t M, int64_t N);
int llx_skinny_tn_partial_many_u(int n, const void* const* U, const void* const* Y, const int64_t* ldy, const int64_t* M, const int64_t* N,
                                 const int64_t* R, void* const* workspaces, const int32_t* const* segs, const int* seg_count,
                                 const void* const* Bt, const int64_t* ldb, void* const* upart, llx_stream_t s);
int llx_skinny_u_reduce(const void* upart, void* out, int64_t M, int64_t N, int64_t R, const int32_t* segs, int seg_count, llx_stream_t s);
/* ... and product i with G[i] != NULL also writes G_i[m, n] = bf16(Y_i[m, n] * colscale_i[n]) (row stride ldg[i]): the (grad_output * scale)
 * operand of an int8 linear's data gradient (subclasses/int8.py:127) from the same read of dy. */
int llx_skinny_tn_partial_many_us(int n, const void* const* U, const void* const* Y, const int64_t* ldy, const int64_t* M, const int64_t* N,
                                  const int64_t* R, void* const* workspaces, const int32_t* const* segs, const int* seg_count,
                                  const void* const* Bt, const int64_t* ldb, void* const* upart, const void* const* colscale,
                                  void* const* G, const int64_t* ldg, llx_stream_t s);
int llx_skinny_tn_reduce_many(int n, const void* const* workspaces, void* const* outs, const int64_t* out_ld, const int64_t* M, const int64_t* N,
                              const int64_t* R, const float* scale, const int* transpose_out, const int* accumulate,
                              const int32_t* const* segs, const int* seg_count, llx_stream_t s);
int llx_pad64(const void* in, int64_t ld, void* out, int64_t R, int64_t C, float scale, int transpose, llx_stream_t s);
int llx_lora_group_pack(const void* const* lora_a, const void* const* lora_b, const int64_t* Ns, const int64_t* ranks, int nm, int64_t K,
                        float scale, void* a_cat, void* b2, void* bT, void* a2t, llx_stream_t s);  /* all four images, one launch (host arrays) */
int llx_lora_groups_pack(const void* const* lora_a, const void* const* lora_b, const int64_t* Ns, const int64_t* ranks, const int* nm,
                         const int64_t* K, const float* scale, void* const* a_cat, void* const* b2, void* const* bT, void* const* a2t, int ng,
                         llx_stream_t s);  /* the same for up to 4 groups (one transformer layer) in one launch; host arrays, group j's members at 4*j+i */
int llx_lora_pack(const void* in, int64_t ld, void* out, int64_t out_ld, int64_t R, int64_t C, int64_t row_off, int64_t col_off, float scale,
                  int transpose, llx_stream_t s);   /* batched LoRA operand images of a linear group (q|k|v, gate|up) */

/* ---- DoRA (modelling/lora.py:47-62): out = (x W^T + s x A^T B^T) * m / ||W + s B A||_row (+ bias).  The row norm is evaluated from
 *      ||W_n||^2 (llx_rownorm2, cached per frozen weight), G = W A^T and A A^T (llx_skinny_nt) - no dense [out,in] temporary;
 *      llx_dora_colscale -> c = m / norm (bf16, the column scale: llx_scale / llx_colscale_bias / GEMM epilogue 4) and 1/norm (fp32);
 *      llx_colsum_mul -> d m[n] = (sum_rows dy * z)[n] / norm[n], two deterministic stages. --------------------------------- */
int llx_rownorm2(const void* W, int64_t ld, float* out, int64_t rows, int64_t cols, llx_stream_t s);
int llx_dora_colscale(const float* wn2, const void* G /*bf16 [N,64]*/, const void* b2 /*bf16 [N,64] = s*B*/, const void* AAt /*bf16 [R,64]*/,
                      const void* m /*bf16 [N]*/, void* c /*bf16 [N]*/, float* inv_norm /*fp32 [N]*/, int64_t N, int64_t R, llx_stream_t s);
int64_t llx_colsum_mul_workspace_bytes(int64_t N);
int llx_colsum_mul(const void* a, int64_t lda, const void* b, int64_t ldb, const float* colscale /*nullable fp32 [N]*/, void* out /*bf16 [N]*/,
                   void* workspace, int64_t M, int64_t N, llx_stream_t s);
int llx_colscale_bias(const void* x, int64_t x_ld, void* y, int64_t y_ld, const void* colscale /*bf16 [cols]*/, const void* bias /*nullable*/,
                      int64_t rows, int64_t cols, llx_stream_t s);

/* ---- audio front end (modelling/audio.py:26-36,53-60): MelSpectrogram(n_fft 512, win 400, hop 160, 128 slaney mels,
 *      power 2, centre/reflect) -> log10/clip/CMN -> bf16 time-major padded features; exact-erf GELU; conv k=3 helpers.
 *      twiddle fp32 [512][2], window fp32 [512], fbank fp32 [257][n_mels] are host-built constants. --------------- */
int llx_mel_spectrogram(const float* audio, int64_t B, int64_t L, const float* twiddle, const float* window, const float* fbank, float* mel,
                        int64_t n_frames, int64_t hop, int64_t n_mels, llx_stream_t s);
int llx_logmel_cmn(const float* mel, void* feat /* bf16 [B, n_frames+1, n_mels] */, int64_t B, int64_t n_frames, int64_t n_mels, llx_stream_t s);
int llx_gelu_fwd(const void* z, int64_t z_ld, void* y, int64_t y_ld, int64_t rows, int64_t cols, llx_stream_t s);
int llx_gelu_bwd(const void* dy, int64_t dy_ld, const void* z, int64_t z_ld, void* dz, int64_t dz_ld, int64_t rows, int64_t cols, llx_stream_t s);
int llx_col2im3(const void* dA, void* dpad, int64_t M, int64_t C, int64_t P, int64_t stride, llx_stream_t s);
int llx_conv_w_reorder(const void* in, void* out, int64_t D, int64_t C, int to_gemm, llx_stream_t s);

/* ---- small utilities of the backward pass ---------------------------------------------------------------------- */
int llx_scale(const void* x, int64_t x_ld, void* y, int64_t y_ld, const float* dev_scalar, float host_scale, const void* colscale,
              int64_t rows, int64_t cols, llx_stream_t s);   /* (g * scale) of subclasses/int8.py:127; loss-scale of dX */
int llx_add(const void* x, const void* y, void* z, int64_t n, llx_stream_t s);   /* residual joins (modelling/llama.py:172-173) */
int llx_transpose(const void* in, int64_t in_ld, void* out, int64_t out_ld, int64_t R, int64_t C, int src_is_i8, llx_stream_t s);
int llx_i8_to_bf16(const void* in, void* out, int64_t n, llx_stream_t s);         /* int_data.T.to(dtype) of subclasses/int8.py:118 */

#ifdef __cplusplus
}
#endif
#endif /* LLX_H */
