// HBM-bound glue kernels of the layer: embedding gather, RoPE, SwiGLU, scaling, transposes, residual add.
// All bf16 traffic is 16 B per lane (8 elements); math in fp32 with the reference's rounding points.
#include "common.h"

// ------------------------------------------------------------------------------------------ embedding
// nn.Embedding forward (modelling/llama.py:180,206; modelling/audio.py:49): out[t,:] = table[ids[t],:].
// out rows may be strided (row stride out_ld) so the gather can land directly behind an audio prefix (audio.py:63).
__global__ void embedding_fwd_kernel(const int64_t* __restrict__ ids, const bf16_t* __restrict__ table, bf16_t* __restrict__ out,
                                     int64_t n_tok, int dim, int64_t vocab, int64_t tok_per_batch, int64_t out_sb, int64_t out_ss) {
  const int64_t t = blockIdx.x;
  int64_t id = ids[t];
  id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);  // host validates; clamp keeps a bad id from faulting the GPU
  const bf16_t* src = table + id * dim;
  bf16_t* dst = out + (t / tok_per_batch) * out_sb + (t % tok_per_batch) * out_ss;
  for (int c = threadIdx.x * 8; c < dim; c += blockDim.x * 8) *reinterpret_cast<u32x4_t*>(dst + c) = *reinterpret_cast<const u32x4_t*>(src + c);
}

extern "C" int llx_embedding_fwd(const int64_t* ids, const void* table, void* out, int64_t n_tok, int64_t dim, int64_t vocab,
                                 int64_t tok_per_batch, int64_t out_sb, int64_t out_ss, hipStream_t stream) {
  LLX_REQUIRE(ids && table && out, "llx_embedding_fwd: null pointer");
  LLX_REQUIRE(dim % 8 == 0 && out_ss % 8 == 0 && out_sb % 8 == 0, "llx_embedding_fwd: dim/strides must be multiples of 8");
  LLX_REQUIRE(tok_per_batch > 0 && vocab > 0, "llx_embedding_fwd: bad sizes");
  if (n_tok == 0) return LLX_OK;
  hipLaunchKernelGGL(embedding_fwd_kernel, dim3((unsigned)n_tok), dim3(256), 0, stream, ids, (const bf16_t*)table, (bf16_t*)out, n_tok,
                     (int)dim, vocab, tok_per_batch, out_sb, out_ss);
  LLX_LAUNCH_CHECK("llx_embedding_fwd");
  return LLX_OK;
}

// Embedding weight gradient: dtable[ids[t],:] += dy[t,:] in fp32 (atomic adds of whole 256-B lane groups).
__global__ void embedding_bwd_kernel(const int64_t* __restrict__ ids, const bf16_t* __restrict__ dy, float* __restrict__ dtable,
                                     int64_t n_tok, int dim, int64_t vocab, int64_t tok_per_batch, int64_t dy_sb, int64_t dy_ss) {
  const int64_t t = blockIdx.x;
  const int64_t id = ids[t];
  if (id < 0 || id >= vocab) return;
  const bf16_t* src = dy + (t / tok_per_batch) * dy_sb + (t % tok_per_batch) * dy_ss;
  float* dst = dtable + id * dim;
  for (int c = threadIdx.x; c < dim; c += blockDim.x) atomicAdd(dst + c, bf2f(src[c]));
}

extern "C" int llx_embedding_bwd(const int64_t* ids, const void* dy, float* dtable_f32, int64_t n_tok, int64_t dim, int64_t vocab,
                                 int64_t tok_per_batch, int64_t dy_sb, int64_t dy_ss, hipStream_t stream) {
  LLX_REQUIRE(ids && dy && dtable_f32, "llx_embedding_bwd: null pointer");
  if (n_tok == 0) return LLX_OK;
  hipLaunchKernelGGL(embedding_bwd_kernel, dim3((unsigned)n_tok), dim3(256), 0, stream, ids, (const bf16_t*)dy, dtable_f32, n_tok, (int)dim,
                     vocab, tok_per_batch, dy_sb, dy_ss);
  LLX_LAUNCH_CHECK("llx_embedding_bwd");
  return LLX_OK;
}

// ------------------------------------------------------------------------------------------ RoPE
// apply_rope (modelling/llama.py:63-73): interleaved pairs (2i,2i+1), fp32 math, one rounding to bf16.
// Works in place on the first `nheads` heads of every row of a [B,S,*] buffer (q heads then k heads of a fused
// projection are contiguous).  sign=+1 forward, -1 backward (rotation by -theta is the exact transpose).
__global__ void rope_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, const float* __restrict__ table, int64_t n_rows,
                            int S, int nheads, int64_t x_sb, int64_t x_ss, int64_t y_sb, int64_t y_ss, float sign) {
  // 16 threads per head (8 elements each); head_dim fixed 128
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int sub = (int)(gid & 15);
  const int64_t hrow = gid >> 4;  // (b*S + s) * nheads + h
  if (hrow >= n_rows * nheads) return;
  const int h = (int)(hrow % nheads);
  const int64_t bs = hrow / nheads;
  const int s = (int)(bs % S);
  const int64_t b = bs / S;
  const u32x4_t v = *reinterpret_cast<const u32x4_t*>(x + b * x_sb + (int64_t)s * x_ss + h * 128 + sub * 8);
  const float* tp = table + ((int64_t)s * 64 + sub * 4) * 2;
  const u32x4_t o = rope8(v, tp, sign);
  *reinterpret_cast<u32x4_t*>(y + b * y_sb + (int64_t)s * y_ss + h * 128 + sub * 8) = o;
}

extern "C" int llx_rope(const void* x, int64_t x_sb, int64_t x_ss, void* y, int64_t y_sb, int64_t y_ss, const float* table, int64_t B,
                        int64_t S, int64_t nheads, int64_t head_dim, int backward, hipStream_t stream) {
  LLX_REQUIRE(x && y && table, "llx_rope: null pointer");
  LLX_REQUIRE(head_dim == 128, "llx_rope: head_dim=%lld unsupported (only 128)", (long long)head_dim);
  LLX_REQUIRE(((x_sb | x_ss | y_sb | y_ss) % 8) == 0, "llx_rope: strides must be multiples of 8 elements");
  const int64_t n = B * S * nheads * 16;
  if (n == 0) return LLX_OK;
  hipLaunchKernelGGL(rope_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, stream, (const bf16_t*)x, (bf16_t*)y, table, B * S, (int)S,
                     (int)nheads, x_sb, x_ss, y_sb, y_ss, backward ? -1.f : 1.f);
  LLX_LAUNCH_CHECK("llx_rope");
  return LLX_OK;
}

// ------------------------------------------------------------------------------------------ SwiGLU
// silu(w1 x) * w3 x (modelling/llama.py:152) with eager rounding points: s = bf16(silu(g)); h = bf16(s * u).

__global__ void swiglu_fwd_kernel(const bf16_t* __restrict__ g, const bf16_t* __restrict__ u, bf16_t* __restrict__ h, int64_t rows, int cols,
                                  int64_t g_ld, int64_t u_ld, int64_t h_ld) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int cpr = cols >> 3;
  if (idx >= rows * cpr) return;
  const int64_t r = idx / cpr;
  const int c = (int)(idx % cpr) * 8;
  const u32x4_t gv = *reinterpret_cast<const u32x4_t*>(g + r * g_ld + c);
  const u32x4_t uv = *reinterpret_cast<const u32x4_t*>(u + r * u_ld + c);
  *reinterpret_cast<u32x4_t*>(h + r * h_ld + c) = swiglu_fwd8(gv, uv);
}

// dg = dh*u*silu'(g), du = dh*silu(g)
__global__ void swiglu_bwd_kernel(const bf16_t* __restrict__ dh, const bf16_t* __restrict__ g, const bf16_t* __restrict__ u,
                                  bf16_t* __restrict__ dg, bf16_t* __restrict__ du, int64_t rows, int cols, int64_t dh_ld, int64_t g_ld,
                                  int64_t u_ld, int64_t dg_ld, int64_t du_ld) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int cpr = cols >> 3;
  if (idx >= rows * cpr) return;
  const int64_t r = idx / cpr;
  const int c = (int)(idx % cpr) * 8;
  const u32x4_t dv = *reinterpret_cast<const u32x4_t*>(dh + r * dh_ld + c);
  const u32x4_t gv = *reinterpret_cast<const u32x4_t*>(g + r * g_ld + c);
  const u32x4_t uv = *reinterpret_cast<const u32x4_t*>(u + r * u_ld + c);
  u32x4_t og, ou;
  swiglu_bwd8(dv, gv, uv, og, ou);
  *reinterpret_cast<u32x4_t*>(dg + r * dg_ld + c) = og;
  *reinterpret_cast<u32x4_t*>(du + r * du_ld + c) = ou;
}

extern "C" int llx_swiglu_fwd(const void* g, int64_t g_ld, const void* u, int64_t u_ld, void* h, int64_t h_ld, int64_t rows, int64_t cols,
                              hipStream_t stream) {
  LLX_REQUIRE(g && u && h, "llx_swiglu_fwd: null pointer");
  LLX_REQUIRE(cols % 8 == 0 && ((g_ld | u_ld | h_ld) % 8) == 0, "llx_swiglu_fwd: cols/strides must be multiples of 8");
  const int64_t n = rows * (cols / 8);
  if (n == 0) return LLX_OK;
  hipLaunchKernelGGL(swiglu_fwd_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, stream, (const bf16_t*)g, (const bf16_t*)u, (bf16_t*)h,
                     rows, (int)cols, g_ld, u_ld, h_ld);
  LLX_LAUNCH_CHECK("llx_swiglu_fwd");
  return LLX_OK;
}

extern "C" int llx_swiglu_bwd(const void* dh, int64_t dh_ld, const void* g, int64_t g_ld, const void* u, int64_t u_ld, void* dg,
                              int64_t dg_ld, void* du, int64_t du_ld, int64_t rows, int64_t cols, hipStream_t stream) {
  LLX_REQUIRE(dh && g && u && dg && du, "llx_swiglu_bwd: null pointer");
  LLX_REQUIRE(cols % 8 == 0 && ((dh_ld | g_ld | u_ld | dg_ld | du_ld) % 8) == 0, "llx_swiglu_bwd: cols/strides must be multiples of 8");
  const int64_t n = rows * (cols / 8);
  if (n == 0) return LLX_OK;
  hipLaunchKernelGGL(swiglu_bwd_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, stream, (const bf16_t*)dh, (const bf16_t*)g,
                     (const bf16_t*)u, (bf16_t*)dg, (bf16_t*)du, rows, (int)cols, dh_ld, g_ld, u_ld, dg_ld, du_ld);
  LLX_LAUNCH_CHECK("llx_swiglu_bwd");
  return LLX_OK;
}

// ------------------------------------------------------------------------------------------ small utilities
// y = bf16(x * s) with s = *scalar (device fp32) * host_scale ; optional per-column bf16 scale (int8 backward: g * scale).
__global__ void scale_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, const float* __restrict__ dev_scalar, float host_scale,
                             const bf16_t* __restrict__ colscale, int64_t rows, int cols, int64_t x_ld, int64_t y_ld) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int cpr = cols >> 3;
  if (idx >= rows * cpr) return;
  const int64_t r = idx / cpr;
  const int c = (int)(idx % cpr) * 8;
  const float s = (dev_scalar ? *dev_scalar : 1.f) * host_scale;
  const u32x4_t v = *reinterpret_cast<const u32x4_t*>(x + r * x_ld + c);
  u32x4_t cs = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
  if (colscale) cs = *reinterpret_cast<const u32x4_t*>(colscale + c);
  u32x4_t o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = pack_bf2(bflo(v[e]) * s * bflo(cs[e]), bfhi(v[e]) * s * bfhi(cs[e]));
  *reinterpret_cast<u32x4_t*>(y + r * y_ld + c) = o;
}

extern "C" int llx_scale(const void* x, int64_t x_ld, void* y, int64_t y_ld, const float* dev_scalar, float host_scale, const void* colscale,
                         int64_t rows, int64_t cols, hipStream_t stream) {
  LLX_REQUIRE(x && y, "llx_scale: null pointer");
  LLX_REQUIRE(cols % 8 == 0 && ((x_ld | y_ld) % 8) == 0, "llx_scale: cols/strides must be multiples of 8");
  const int64_t n = rows * (cols / 8);
  if (n == 0) return LLX_OK;
  hipLaunchKernelGGL(scale_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, stream, (const bf16_t*)x, (bf16_t*)y, dev_scalar, host_scale,
                     (const bf16_t*)colscale, rows, (int)cols, x_ld, y_ld);
  LLX_LAUNCH_CHECK("llx_scale");
  return LLX_OK;
}

// z = bf16(x + y) (residual joins of the backward pass)
__global__ void add_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ y, bf16_t* __restrict__ z, int64_t n8) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n8) return;
  const u32x4_t a = reinterpret_cast<const u32x4_t*>(x)[i], b = reinterpret_cast<const u32x4_t*>(y)[i];
  u32x4_t o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = pack_bf2(bflo(a[e]) + bflo(b[e]), bfhi(a[e]) + bfhi(b[e]));
  reinterpret_cast<u32x4_t*>(z)[i] = o;
}

extern "C" int llx_add(const void* x, const void* y, void* z, int64_t n, hipStream_t stream) {
  LLX_REQUIRE(x && y && z && n % 8 == 0, "llx_add: null pointer or n not a multiple of 8");
  if (n == 0) return LLX_OK;
  hipLaunchKernelGGL(add_kernel, dim3((unsigned)cdiv64(n / 8, 256)), dim3(256), 0, stream, (const bf16_t*)x, (const bf16_t*)y, (bf16_t*)z, n / 8);
  LLX_LAUNCH_CHECK("llx_add");
  return LLX_OK;
}

// out[C,R] = in[R,C]^T for 2-byte elements (cached transposed copies of frozen weights for the dgrad GEMM).
// src_is_i8: the source is int8 and is widened to bf16 on the way (weight-only int8 path; int8 is exact in bf16).
template <bool I8>
__global__ void transpose_kernel(const void* __restrict__ in, bf16_t* __restrict__ out, int R, int C, int64_t in_ld, int64_t out_ld) {
  __shared__ bf16_t tile[64][66];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  for (int i = threadIdx.x; i < 64 * 64; i += blockDim.x) {
    const int r = i >> 6, c = i & 63;
    bf16_t v = 0;
    if (r0 + r < R && c0 + c < C) {
      if constexpr (I8) v = f2bf((float)reinterpret_cast<const int8_t*>(in)[(int64_t)(r0 + r) * in_ld + c0 + c]);
      else v = reinterpret_cast<const bf16_t*>(in)[(int64_t)(r0 + r) * in_ld + c0 + c];
    }
    tile[r][c] = v;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 64; i += blockDim.x) {
    const int c = i >> 6, r = i & 63;
    if (r0 + r < R && c0 + c < C) out[(int64_t)(c0 + c) * out_ld + r0 + r] = tile[r][c];
  }
}

extern "C" int llx_transpose(const void* in, int64_t in_ld, void* out, int64_t out_ld, int64_t R, int64_t C, int src_is_i8, hipStream_t stream) {
  LLX_REQUIRE(in && out && R > 0 && C > 0, "llx_transpose: bad arguments");
  const dim3 grid((unsigned)cdiv64(C, 64), (unsigned)cdiv64(R, 64));
  if (src_is_i8) hipLaunchKernelGGL(transpose_kernel<true>, grid, dim3(256), 0, stream, in, (bf16_t*)out, (int)R, (int)C, in_ld, out_ld);
  else hipLaunchKernelGGL(transpose_kernel<false>, grid, dim3(256), 0, stream, in, (bf16_t*)out, (int)R, (int)C, in_ld, out_ld);
  LLX_LAUNCH_CHECK("llx_transpose");
  return LLX_OK;
}

// int8 -> bf16 widening copy (same layout): bf16 image of a frozen int8 weight for the weight-only linear.
__global__ void i8_to_bf16_kernel(const int8_t* __restrict__ in, bf16_t* __restrict__ out, int64_t n) {
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 8;
  if (i >= n) return;
  if (i + 8 <= n) {
    const u32x2_t v = *reinterpret_cast<const u32x2_t*>(in + i);
    u32x4_t o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const uint32_t w = v[e >> 1] >> ((e & 1) * 16);
      o[e] = pack_bf2((float)(int8_t)(w & 0xff), (float)(int8_t)((w >> 8) & 0xff));
    }
    *reinterpret_cast<u32x4_t*>(out + i) = o;
  } else {
    for (int64_t j = i; j < n; ++j) out[j] = f2bf((float)in[j]);
  }
}

extern "C" int llx_i8_to_bf16(const void* in, void* out, int64_t n, hipStream_t stream) {
  LLX_REQUIRE(in && out && ((uintptr_t)in % 8) == 0 && ((uintptr_t)out % 16) == 0, "llx_i8_to_bf16: null/unaligned pointer");
  if (n == 0) return LLX_OK;
  hipLaunchKernelGGL(i8_to_bf16_kernel, dim3((unsigned)cdiv64(cdiv64(n, 8), 256)), dim3(256), 0, stream, (const int8_t*)in, (bf16_t*)out, n);
  LLX_LAUNCH_CHECK("llx_i8_to_bf16");
  return LLX_OK;
}

// out[R, 64] = bf16(scale * src) zero-padded to 64 columns; src is [R, C] (transpose=0) or [C, R] (transpose=1).
// Builds the K-extension operands of the LoRA-fused GEMM (s*lora_b -> [out,64]; s*lora_a^T -> [in,64]).
__global__ void pad64_kernel(const bf16_t* __restrict__ in, int64_t ld, bf16_t* __restrict__ out, int R, int C, float scale, int transpose) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)R * 64) return;
  const int r = (int)(idx >> 6), c = (int)(idx & 63);
  float v = 0.f;
  if (c < C) v = bf2f(transpose ? in[(int64_t)c * ld + r] : in[(int64_t)r * ld + c]) * scale;
  out[idx] = f2bf(v);
}

extern "C" int llx_pad64(const void* in, int64_t ld, void* out, int64_t R, int64_t C, float scale, int transpose, hipStream_t stream) {
  LLX_REQUIRE(in && out && R > 0 && C > 0 && C <= 64, "llx_pad64: bad arguments (C=%lld)", (long long)C);
  hipLaunchKernelGGL(pad64_kernel, dim3((unsigned)cdiv64(R * 64, 256)), dim3(256), 0, stream, (const bf16_t*)in, ld, (bf16_t*)out, (int)R, (int)C,
                     scale, transpose);
  LLX_LAUNCH_CHECK("llx_pad64");
  return LLX_OK;
}

// Scatter a small [R, C] bf16 matrix (scaled) into a bigger zero-initialised operand image:
//   transpose = 0: out[(row_off + r) * out_ld + col_off + c] = scale * in[r, c]
//   transpose = 1: out[(row_off + c) * out_ld + col_off + r] = scale * in[r, c]
// Builds the batched LoRA operands of a linear group (block-diagonal s*B, s*A^T, B^T, stacked A).
__global__ void lora_pack_kernel(const bf16_t* __restrict__ in, int64_t ld, bf16_t* __restrict__ out, int64_t out_ld, int R, int C, int row_off,
                                 int col_off, float scale, int transpose) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)R * C) return;
  const int r = (int)(idx / C), c = (int)(idx % C);
  const float v = bf2f(in[(int64_t)r * ld + c]) * scale;
  if (transpose) out[(int64_t)(row_off + c) * out_ld + col_off + r] = f2bf(v);
  else out[(int64_t)(row_off + r) * out_ld + col_off + c] = f2bf(v);
}

extern "C" int llx_lora_pack(const void* in, int64_t ld, void* out, int64_t out_ld, int64_t R, int64_t C, int64_t row_off, int64_t col_off,
                             float scale, int transpose, hipStream_t stream) {
  LLX_REQUIRE(in && out && R > 0 && C > 0, "llx_lora_pack: bad arguments");
  hipLaunchKernelGGL(lora_pack_kernel, dim3((unsigned)cdiv64(R * C, 256)), dim3(256), 0, stream, (const bf16_t*)in, ld, (bf16_t*)out, out_ld, (int)R,
                     (int)C, (int)row_off, (int)col_off, scale, transpose);
  LLX_LAUNCH_CHECK("llx_lora_pack");
  return LLX_OK;
}

// All four batched LoRA operand images of a linear group in ONE launch (forward builds them, backward reuses them):
//   a_cat [R, K]      rows r_off_i.. = lora_a_i                       (B operand of t = x @ A_cat^T)
//   b2    [N, 64]     block (n_off_i.., r_off_i..) = s * lora_b_i     (K-extension operand of the forward GEMM)
//   bT    [R, N]      block (r_off_i.., n_off_i..) = lora_b_i^T       (B operand of u = dy @ B_blk)
//   a2t   [K, 64]     cols r_off_i.. = s * lora_a_i^T                 (K-extension operand of the dgrad GEMM)
struct LoraMember { const bf16_t* a; const bf16_t* b; int N, n_off, r, r_off; };
struct LoraGroup { LoraMember m[4]; int nm, K, N, R; float scale; bf16_t* a_cat; bf16_t* b2; bf16_t* bT; bf16_t* a2t;
                   int aligned8; /* every member's rank and rank offset are multiples of 8: an 8-element chunk never straddles members */ };

// One thread = 8 consecutive elements of one image row (one 16-byte store); K % 8 == 0 and N % 8 == 0 (checked by the launcher).
__device__ __forceinline__ void lora_group_pack_body(const LoraGroup& g) {
  const int64_t n_acat = (int64_t)g.R * g.K / 8, n_b2 = (int64_t)g.N * 8, n_bT = (int64_t)g.R * g.N / 8, n_a2t = (int64_t)g.K * 8;
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < n_acat) {  // a_cat[r][k0..k0+8): a straight copy of 16 bytes of the owning member's lora_a row
    const int kc = g.K / 8;
    const int r = (int)(idx / kc), k0 = (int)(idx % kc) * 8;
    u32x4_t v = {0u, 0u, 0u, 0u};
    for (int i = 0; i < g.nm; ++i)
      if (r >= g.m[i].r_off && r < g.m[i].r_off + g.m[i].r) v = *reinterpret_cast<const u32x4_t*>(g.m[i].a + (int64_t)(r - g.m[i].r_off) * g.K + k0);
    *reinterpret_cast<u32x4_t*>(g.a_cat + (int64_t)r * g.K + k0) = v;
    return;
  }
  idx -= n_acat;
  if (g.aligned8) {
    // chunks never straddle members: pick the member first, then load without per-element conditions - the element-wise form below
    // compiles to load, wait, load, wait ... (eight exposed latencies per thread)
    if (idx < n_b2) {  // b2[n][c0..c0+8): 16 contiguous bytes of the member's lora_b row, scaled
      const int n = (int)(idx >> 3), c0 = (int)(idx & 7) * 8;
      u32x4_t o = {0u, 0u, 0u, 0u};
      for (int i = 0; i < g.nm; ++i) {
        const LoraMember& m = g.m[i];
        if (n >= m.n_off && n < m.n_off + m.N && c0 >= m.r_off && c0 < m.r_off + m.r) {
          const u32x4_t v = *reinterpret_cast<const u32x4_t*>(m.b + (int64_t)(n - m.n_off) * m.r + (c0 - m.r_off));
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = pack_bf2(bflo(v[e]) * g.scale, bfhi(v[e]) * g.scale);
        }
      }
      *reinterpret_cast<u32x4_t*>(g.b2 + (int64_t)n * 64 + c0) = o;
      return;
    }
    idx -= n_b2;
    if (idx < n_bT) {  // bT[r][n0..n0+8): eight lora_b elements one row apart
      const int nc = g.N / 8;
      const int r = (int)(idx / nc), n0 = (int)(idx % nc) * 8;
      u32x4_t o = {0u, 0u, 0u, 0u};
      for (int i = 0; i < g.nm; ++i) {
        const LoraMember& m = g.m[i];
        if (r >= m.r_off && r < m.r_off + m.r && n0 >= m.n_off && n0 < m.n_off + m.N) {
          const bf16_t* src = m.b + (int64_t)(n0 - m.n_off) * m.r + (r - m.r_off);
          bf16_t v[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = src[(int64_t)e * m.r];
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (uint32_t)v[2 * e] | ((uint32_t)v[2 * e + 1] << 16);
        }
      }
      *reinterpret_cast<u32x4_t*>(g.bT + (int64_t)r * g.N + n0) = o;
      return;
    }
    idx -= n_bT;
    if (idx < n_a2t) {  // a2t[k][c0..c0+8): eight lora_a elements one row (K) apart, scaled
      const int k = (int)(idx >> 3), c0 = (int)(idx & 7) * 8;
      u32x4_t o = {0u, 0u, 0u, 0u};
      for (int i = 0; i < g.nm; ++i) {
        const LoraMember& m = g.m[i];
        if (c0 >= m.r_off && c0 < m.r_off + m.r) {
          const bf16_t* src = m.a + (int64_t)(c0 - m.r_off) * g.K + k;
          bf16_t v[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = src[(int64_t)e * g.K];
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = pack_bf2(bf2f(v[2 * e]) * g.scale, bf2f(v[2 * e + 1]) * g.scale);
        }
      }
      *reinterpret_cast<u32x4_t*>(g.a2t + (int64_t)k * 64 + c0) = o;
    }
    return;
  }
  if (idx < n_b2) {  // b2[n][c0..c0+8) = s * lora_b_i[n - n_off][c - r_off]
    const int n = (int)(idx >> 3), c0 = (int)(idx & 7) * 8;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < g.nm; ++i) {
      const LoraMember& m = g.m[i];
      if (n >= m.n_off && n < m.n_off + m.N) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int c = c0 + e;
          if (c >= m.r_off && c < m.r_off + m.r) v[e] = bf2f(m.b[(int64_t)(n - m.n_off) * m.r + (c - m.r_off)]) * g.scale;
        }
      }
    }
    u32x4_t o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = pack_bf2(v[2 * e], v[2 * e + 1]);
    *reinterpret_cast<u32x4_t*>(g.b2 + (int64_t)n * 64 + c0) = o;
    return;
  }
  idx -= n_b2;
  if (idx < n_bT) {  // bT[r][n0..n0+8) = lora_b_i[n - n_off][r - r_off] (member boundaries are multiples of 8)
    const int nc = g.N / 8;
    const int r = (int)(idx / nc), n0 = (int)(idx % nc) * 8;
    bf16_t v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < g.nm; ++i) {
      const LoraMember& m = g.m[i];
      if (r >= m.r_off && r < m.r_off + m.r && n0 >= m.n_off && n0 < m.n_off + m.N) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = m.b[(int64_t)(n0 + e - m.n_off) * m.r + (r - m.r_off)];
      }
    }
    u32x4_t o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (uint32_t)v[2 * e] | ((uint32_t)v[2 * e + 1] << 16);
    *reinterpret_cast<u32x4_t*>(g.bT + (int64_t)r * g.N + n0) = o;
    return;
  }
  idx -= n_bT;
  if (idx < n_a2t) {  // a2t[k][c0..c0+8) = s * lora_a_i[c - r_off][k]
    const int k = (int)(idx >> 3), c0 = (int)(idx & 7) * 8;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < g.nm; ++i) {
      const LoraMember& m = g.m[i];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int c = c0 + e;
        if (c >= m.r_off && c < m.r_off + m.r) v[e] = bf2f(m.a[(int64_t)(c - m.r_off) * g.K + k]) * g.scale;
      }
    }
    u32x4_t o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = pack_bf2(v[2 * e], v[2 * e + 1]);
    *reinterpret_cast<u32x4_t*>(g.a2t + (int64_t)k * 64 + c0) = o;
  }
}

__global__ void lora_group_pack_kernel(const LoraGroup g) { lora_group_pack_body(g); }

// The operand images of up to LORA_MAX_GROUPS linear groups (the q|k|v, wo, gate|up and w2 groups of one transformer layer) in ONE
// launch: blockIdx.y picks the group.
#define LORA_MAX_GROUPS 4
struct LoraGroups { LoraGroup g[LORA_MAX_GROUPS]; };
__global__ void lora_groups_pack_kernel(const LoraGroups gs) { lora_group_pack_body(gs.g[blockIdx.y]); }

static int lora_fill_group(LoraGroup& g, const void* const* lora_a, const void* const* lora_b, const int64_t* Ns, const int64_t* ranks, int nm,
                           int64_t K, float scale, void* a_cat, void* b2, void* bT, void* a2t, int64_t* total) {
  LLX_REQUIRE(lora_a && lora_b && Ns && ranks && nm >= 1 && nm <= 4 && a_cat && b2 && bT && a2t, "llx_lora_group_pack: bad arguments");
  int n_off = 0, r_off = 0;
  for (int i = 0; i < nm; ++i) {
    g.m[i].a = (const bf16_t*)lora_a[i]; g.m[i].b = (const bf16_t*)lora_b[i];
    g.m[i].N = (int)Ns[i]; g.m[i].n_off = n_off; g.m[i].r = (int)ranks[i]; g.m[i].r_off = r_off;
    n_off += (int)Ns[i]; r_off += (int)ranks[i];
  }
  LLX_REQUIRE(r_off <= 64, "llx_lora_group_pack: total rank %d > 64", r_off);
  g.nm = nm; g.K = (int)K; g.N = n_off; g.R = r_off; g.scale = scale;
  g.aligned8 = 1;
  for (int i = 0; i < nm; ++i)
    if (g.m[i].r % 8 != 0 || g.m[i].r_off % 8 != 0 || ((uintptr_t)lora_b[i] & 15) != 0) g.aligned8 = 0;
  g.a_cat = (bf16_t*)a_cat; g.b2 = (bf16_t*)b2; g.bT = (bf16_t*)bT; g.a2t = (bf16_t*)a2t;
  LLX_REQUIRE(K % 8 == 0, "llx_lora_group_pack: K=%lld must be a multiple of 8", (long long)K);
  for (int i = 0; i < nm; ++i) LLX_REQUIRE(Ns[i] % 8 == 0, "llx_lora_group_pack: member %d: N=%lld must be a multiple of 8", i, (long long)Ns[i]);
  LLX_REQUIRE((((uintptr_t)a_cat | (uintptr_t)b2 | (uintptr_t)bT | (uintptr_t)a2t) & 15) == 0, "llx_lora_group_pack: the images must be 16-byte aligned");
  for (int i = 0; i < nm; ++i) LLX_REQUIRE(((uintptr_t)lora_a[i] & 15) == 0, "llx_lora_group_pack: lora_a must be 16-byte aligned");
  *total = ((int64_t)g.R * g.K + (int64_t)g.N * 64 + (int64_t)g.R * g.N + (int64_t)g.K * 64) / 8;  // threads: 8 elements each
  return LLX_OK;
}

// ng (<= 4) groups in one launch.  Group j has nm[j] members whose descriptors sit at index 4*j + i of lora_a / lora_b / Ns / ranks
// (host arrays of length 4*ng); K[j], scale[j] and the four output images a_cat[j], b2[j], bT[j], a2t[j] per group.
extern "C" int llx_lora_groups_pack(const void* const* lora_a, const void* const* lora_b, const int64_t* Ns, const int64_t* ranks, const int* nm,
                                    const int64_t* K, const float* scale, void* const* a_cat, void* const* b2, void* const* bT, void* const* a2t,
                                    int ng, hipStream_t stream) {
  LLX_REQUIRE(nm && K && scale && a_cat && b2 && bT && a2t && ng >= 1 && ng <= LORA_MAX_GROUPS, "llx_lora_groups_pack: bad arguments (1..4 groups)");
  LoraGroups gs;
  int64_t most = 0;
  for (int j = 0; j < ng; ++j) {
    int64_t total = 0;
    const int rc = lora_fill_group(gs.g[j], lora_a + 4 * j, lora_b + 4 * j, Ns + 4 * j, ranks + 4 * j, nm[j], K[j], scale[j], a_cat[j], b2[j], bT[j],
                                   a2t[j], &total);
    if (rc != LLX_OK) return rc;
    if (total > most) most = total;
  }
  hipLaunchKernelGGL(lora_groups_pack_kernel, dim3((unsigned)cdiv64(most, 256), (unsigned)ng), dim3(256), 0, stream, gs);
  LLX_LAUNCH_CHECK("llx_lora_groups_pack");
  return LLX_OK;
}

// members: arrays of length nm (<= 4): lora_a[i] [r_i, K] and lora_b[i] [N_i, r_i], both contiguous.
extern "C" int llx_lora_group_pack(const void* const* lora_a, const void* const* lora_b, const int64_t* Ns, const int64_t* ranks, int nm, int64_t K,
                                   float scale, void* a_cat, void* b2, void* bT, void* a2t, hipStream_t stream) {
  LoraGroup g;
  int64_t total = 0;
  const int rc = lora_fill_group(g, lora_a, lora_b, Ns, ranks, nm, K, scale, a_cat, b2, bT, a2t, &total);
  if (rc != LLX_OK) return rc;
  hipLaunchKernelGGL(lora_group_pack_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, stream, g);
  LLX_LAUNCH_CHECK("llx_lora_group_pack");
  return LLX_OK;
}
