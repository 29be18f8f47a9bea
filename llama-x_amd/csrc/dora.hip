// DoRA (weight-decomposed LoRA) pieces of DoRALinear.forward (modelling/lora.py:47-62):
//     out = (x W^T + s x A^T B^T) * m / || W + s B A ||_row  (+ bias)
// The reference materialises the dense [out, in] matrix W + s B A every forward to take its row norms.  Here the square is
// expanded, per output row n (b_n = s * B[n, :], a block of the zero-padded [N, 64] K-extension operand of the fused GEMM):
//     || W_n + b_n A ||^2 = || W_n ||^2  +  2 b_n . (A W_n^T)  +  b_n (A A^T) b_n^T
// ||W_n||^2 is cached per (frozen) weight, G = W A^T [N, 64] and A A^T [R, 64] come from the skinny MFMA kernel (skinny.hip),
// so a forward reads W once at the HBM rate and never writes an [out, in] temporary.
#include "common.h"

// ------------------------------------------------------------------------------------------ || W_n ||^2
// One wave per row, 16 B per lane per step; fp32 accumulate.
__global__ __launch_bounds__(256) void rownorm2_kernel(const bf16_t* __restrict__ W, int64_t ld, float* __restrict__ out, int rows, int cols) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const bf16_t* w = W + (int64_t)row * ld;
  float s = 0.f;
  for (int c = lane * 8; c < cols; c += 64 * 8) {
    const u32x4_t v = *reinterpret_cast<const u32x4_t*>(w + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float a = bflo(v[e]), b = bfhi(v[e]);
      s += a * a;
      s += b * b;
    }
  }
  s = wave_sum(s);
  if (lane == 0) out[row] = s;
}

extern "C" int llx_rownorm2(const void* W, int64_t ld, float* out, int64_t rows, int64_t cols, hipStream_t stream) {
  LLX_REQUIRE(W && out && rows > 0 && cols > 0, "llx_rownorm2: bad arguments");
  LLX_REQUIRE(cols % 8 == 0 && ld % 8 == 0 && (uintptr_t)W % 16 == 0, "llx_rownorm2: cols / ld must be multiples of 8, W 16-byte aligned");
  hipLaunchKernelGGL(rownorm2_kernel, dim3((unsigned)cdiv64(rows, 4)), dim3(256), 0, stream, (const bf16_t*)W, ld, out, (int)rows, (int)cols);
  LLX_LAUNCH_CHECK("llx_rownorm2");
  return LLX_OK;
}

// ------------------------------------------------------------------------------------------ c = m / ||W + s B A||, 1/norm
// One thread per output row.  AAt (bf16 [R, 64], columns >= R ignored) sits in LDS as fp32.
// Rounding points of the bf16 eager graph: the norm is a bf16 tensor, m / norm is a bf16 tensor (modelling/lora.py:58-59).
__global__ __launch_bounds__(256) void dora_colscale_kernel(const float* __restrict__ wn2, const bf16_t* __restrict__ G, const bf16_t* __restrict__ b2,
                                                            const bf16_t* __restrict__ AAt, const bf16_t* __restrict__ m, bf16_t* __restrict__ c,
                                                            float* __restrict__ inv_norm, int N, int R) {
  __shared__ float aat[64][65];
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int j = i >> 6, k = i & 63;
    aat[j][k] = (j < R && k < R) ? bf2f(AAt[j * 64 + k]) : 0.f;
  }
  __syncthreads();
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  float b[64];
  float cross = 0.f;
#pragma unroll
  for (int j8 = 0; j8 < 8; ++j8) {
    const u32x4_t bv = *reinterpret_cast<const u32x4_t*>(b2 + (int64_t)n * 64 + j8 * 8);
    const u32x4_t gv = *reinterpret_cast<const u32x4_t*>(G + (int64_t)n * 64 + j8 * 8);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      b[j8 * 8 + 2 * e] = bflo(bv[e]);
      b[j8 * 8 + 2 * e + 1] = bfhi(bv[e]);
      cross += bflo(bv[e]) * bflo(gv[e]) + bfhi(bv[e]) * bfhi(gv[e]);
    }
  }
  float quad = 0.f;
#pragma unroll
  for (int j = 0; j < 64; ++j) {  // fully unrolled: b[] stays in registers (a dynamic index would send it to scratch)
    if (j < R) {                   // wave-uniform
      float row = 0.f;
#pragma unroll
      for (int k = 0; k < 64; ++k) row += aat[j][k] * b[k];  // b[k] = 0 beyond the member's rank block
      quad += b[j] * row;
    }
  }
  const float n2 = fmaxf(wn2[n] + 2.f * cross + quad, 0.f);
  const float nrm = bf2f(f2bf(sqrtf(n2)));
  c[n] = f2bf(bf2f(m[n]) / nrm);
  inv_norm[n] = 1.f / nrm;
}

// wn2 fp32 [N] = ||W_n||^2; G bf16 [N,64] = W A^T; b2 bf16 [N,64] = s * B (zero outside the member's rank columns);
// AAt bf16 [R,64] = A A^T; m bf16 [N]  ->  c bf16 [N] = m / ||W + s B A||_row (DoRALinear's column scale), inv_norm fp32 [N].
extern "C" int llx_dora_colscale(const float* wn2, const void* G, const void* b2, const void* AAt, const void* m, void* c, float* inv_norm,
                                 int64_t N, int64_t R, hipStream_t stream) {
  LLX_REQUIRE(wn2 && G && b2 && AAt && m && c && inv_norm, "llx_dora_colscale: null pointer");
  LLX_REQUIRE(N > 0 && R > 0 && R <= 64, "llx_dora_colscale: need N > 0 and 0 < R <= 64 (N=%lld R=%lld)", (long long)N, (long long)R);
  LLX_REQUIRE(((uintptr_t)G | (uintptr_t)b2) % 16 == 0, "llx_dora_colscale: G / b2 must be 16-byte aligned");
  hipLaunchKernelGGL(dora_colscale_kernel, dim3((unsigned)cdiv64(N, 256)), dim3(256), 0, stream, wn2, (const bf16_t*)G, (const bf16_t*)b2,
                     (const bf16_t*)AAt, (const bf16_t*)m, (bf16_t*)c, inv_norm, (int)N, (int)R);
  LLX_LAUNCH_CHECK("llx_dora_colscale");
  return LLX_OK;
}

// ------------------------------------------------------------------------------------------ out[n] = cs[n] * sum_m a[m,n] b[m,n]
// The gradient of DoRA's magnitude vector: d m[n] = sum_rows dy[row,n] * z[row,n] / norm[n]  (z = the un-scaled LoRA output).
// Two deterministic stages: fp32 partials over CM_SPLIT row stripes, then a column reduce.  HBM-bound on a and b (read once).
#define CM_SPLIT 32

__global__ __launch_bounds__(256) void colsum_mul_partial_kernel(const bf16_t* __restrict__ a, int64_t lda, const bf16_t* __restrict__ b, int64_t ldb,
                                                                 float* __restrict__ part, int M, int N) {
  // block = 32 column chunks (8 columns each = 256 columns) x 8 row lanes
  __shared__ float red[8][256];
  const int cc = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int col = blockIdx.x * 256 + cc * 8;
  const int rows_per = (M + CM_SPLIT - 1) / CM_SPLIT;
  const int r0 = blockIdx.y * rows_per, r1 = min(M, r0 + rows_per);
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (col < N) {
    for (int r = r0 + rl; r < r1; r += 8) {
      const u32x4_t av = *reinterpret_cast<const u32x4_t*>(a + (int64_t)r * lda + col);
      const u32x4_t bv = *reinterpret_cast<const u32x4_t*>(b + (int64_t)r * ldb + col);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc[2 * e] += bflo(av[e]) * bflo(bv[e]);
        acc[2 * e + 1] += bfhi(av[e]) * bfhi(bv[e]);
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[rl][cc * 8 + e] = acc[e];
  __syncthreads();
  const int c = threadIdx.x;
  if (blockIdx.x * 256 + c < N) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += red[i][c];
    part[(int64_t)blockIdx.y * N + blockIdx.x * 256 + c] = s;
  }
}

__global__ __launch_bounds__(256) void colsum_mul_reduce_kernel(const float* __restrict__ part, const float* __restrict__ cs, bf16_t* __restrict__ out, int N) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  float s = 0.f;
  for (int i = 0; i < CM_SPLIT; ++i) s += part[(int64_t)i * N + n];
  out[n] = f2bf(cs ? s * cs[n] : s);
}

extern "C" int64_t llx_colsum_mul_workspace_bytes(int64_t N) { return (int64_t)CM_SPLIT * N * 4; }

extern "C" int llx_colsum_mul(const void* a, int64_t lda, const void* b, int64_t ldb, const float* colscale /*nullable fp32 [N]*/, void* out,
                              void* workspace, int64_t M, int64_t N, hipStream_t stream) {
  LLX_REQUIRE(a && b && out && workspace, "llx_colsum_mul: null pointer");
  LLX_REQUIRE(M > 0 && N > 0 && N % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0, "llx_colsum_mul: N and row strides must be multiples of 8");
  LLX_REQUIRE(((uintptr_t)a | (uintptr_t)b) % 16 == 0, "llx_colsum_mul: unaligned pointer");
  const unsigned gx = (unsigned)cdiv64(N, 256);
  hipLaunchKernelGGL(colsum_mul_partial_kernel, dim3(gx, CM_SPLIT), dim3(256), 0, stream, (const bf16_t*)a, lda, (const bf16_t*)b, ldb,
                     (float*)workspace, (int)M, (int)N);
  LLX_LAUNCH_CHECK("llx_colsum_mul(partial)");
  hipLaunchKernelGGL(colsum_mul_reduce_kernel, dim3(gx), dim3(256), 0, stream, (const float*)workspace, colscale, (bf16_t*)out, (int)N);
  LLX_LAUNCH_CHECK("llx_colsum_mul(reduce)");
  return LLX_OK;
}

// ------------------------------------------------------------------------------------------ y = bf16(bf16(x * cs[n]) + bias[n])
// DoRALinear with a bias: the rescale and the bias add are separate bf16 ops in the reference (modelling/lora.py:59-61).
__global__ void colscale_bias_kernel(const bf16_t* __restrict__ x, int64_t x_ld, bf16_t* __restrict__ y, int64_t y_ld, const bf16_t* __restrict__ cs,
                                     const bf16_t* __restrict__ bias, int64_t rows, int cols) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int cpr = cols >> 3;
  if (idx >= rows * cpr) return;
  const int64_t r = idx / cpr;
  const int c = (int)(idx % cpr) * 8;
  const u32x4_t v = *reinterpret_cast<const u32x4_t*>(x + r * x_ld + c);
  const u32x4_t s = *reinterpret_cast<const u32x4_t*>(cs + c);
  u32x4_t bb = {0u, 0u, 0u, 0u};
  if (bias) bb = *reinterpret_cast<const u32x4_t*>(bias + c);
  u32x4_t o;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float lo = bf2f(f2bf(bflo(v[e]) * bflo(s[e]))), hi = bf2f(f2bf(bfhi(v[e]) * bfhi(s[e])));
    if (bias) { lo += bflo(bb[e]); hi += bfhi(bb[e]); }
    o[e] = pack_bf2(lo, hi);
  }
  *reinterpret_cast<u32x4_t*>(y + r * y_ld + c) = o;
}

extern "C" int llx_colscale_bias(const void* x, int64_t x_ld, void* y, int64_t y_ld, const void* colscale, const void* bias /*nullable*/,
                                 int64_t rows, int64_t cols, hipStream_t stream) {
  LLX_REQUIRE(x && y && colscale, "llx_colscale_bias: null pointer");
  LLX_REQUIRE(cols % 8 == 0 && ((x_ld | y_ld) % 8) == 0, "llx_colscale_bias: cols/strides must be multiples of 8");
  LLX_REQUIRE(((uintptr_t)colscale % 16) == 0 && (!bias || (uintptr_t)bias % 16 == 0), "llx_colscale_bias: colscale / bias must be 16-byte aligned");
  const int64_t n = rows * (cols / 8);
  if (n == 0) return LLX_OK;
  hipLaunchKernelGGL(colscale_bias_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, stream, (const bf16_t*)x, x_ld, (bf16_t*)y, y_ld,
                     (const bf16_t*)colscale, (const bf16_t*)bias, rows, (int)cols);
  LLX_LAUNCH_CHECK("llx_colscale_bias");
  return LLX_OK;
}
