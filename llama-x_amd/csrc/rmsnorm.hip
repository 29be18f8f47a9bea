// RMSNorm forward / backward for bf16 rows (HBM-bound; one pass over x).
// Semantics follow nn.RMSNorm(D, eps) as used at reference modelling/llama.py:158,160,182:
//   y = bf16( float(x) * rsqrt(mean(x^2) + eps) * float(w) )      -- single rounding.
#include "common.h"

// ---------------------------------------------------------------- forward
// One wave per row, 4 rows per 256-thread block. Row values are kept in registers
// (NCH chunks of 512 elements; lane owns 8 contiguous bf16 = one 16-B load per chunk).
template <int NCH>
__global__ __launch_bounds__(256) void rmsnorm_fwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                          bf16_t* __restrict__ y, float* __restrict__ rstd_out,
                                                          int64_t rows, int dim, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const bf16_t* xr = x + row * dim;
  u32x4_t v[NCH];
  float ss = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int col = c * 512 + lane * 8;
    if (col < dim) {
      v[c] = *reinterpret_cast<const u32x4_t*>(xr + col);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float a = bflo(v[c][e]), b = bfhi(v[c][e]);
        ss += a * a + b * b;
      }
    }
  }
  ss = wave_sum(ss);
  const float rstd = rsqrtf(ss / (float)dim + eps);
  if (lane == 0 && rstd_out) rstd_out[row] = rstd;
  bf16_t* yr = y + row * dim;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int col = c * 512 + lane * 8;
    if (col < dim) {
      u32x4_t wv = *reinterpret_cast<const u32x4_t*>(w + col);
      u32x4_t o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float a = bflo(v[c][e]) * rstd * bflo(wv[e]);
        float b = bfhi(v[c][e]) * rstd * bfhi(wv[e]);
        o[e] = pack_bf2(a, b);
      }
      *reinterpret_cast<u32x4_t*>(yr + col) = o;
    }
  }
}

// ---------------------------------------------------------------- backward
// One WAVE per row (no block-wide barriers): a 256-thread block = 4 waves walks `rows_per_block` rows, wave w taking rows
// w, w+4, ...; lane l owns columns {c*512 + l*8 .. +8}.  dx is written per row; the per-lane dw partial sums stay in
// registers, are combined across the 4 waves through LDS and written once per block to dw_partial[blockIdx][dim] (fp32).
//   xhat = x*rstd ; g = dy*w ; dx = rstd * (g - xhat * mean(g*xhat)) [+ dres] ; dw = sum_rows(dy*xhat)
template <int NCH>
__global__ __launch_bounds__(256) void rmsnorm_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                          const bf16_t* __restrict__ w, const float* __restrict__ rstd,
                                                          bf16_t* __restrict__ dx, float* __restrict__ dw_partial,
                                                          const bf16_t* __restrict__ dres, int64_t rows, int dim, int rows_per_block) {
  extern __shared__ __attribute__((aligned(16))) float dwsh[];  // [dim] when dw is requested
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float wv[NCH][8];
  float dwacc[NCH][8];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int col = c * 512 + lane * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) { dwacc[c][e] = 0.f; wv[c][e] = 0.f; }
    if (col < dim) {
      const u32x4_t u = *reinterpret_cast<const u32x4_t*>(w + col);
#pragma unroll
      for (int e = 0; e < 4; ++e) { wv[c][2 * e] = bflo(u[e]); wv[c][2 * e + 1] = bfhi(u[e]); }
    }
  }
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  for (int64_t row = r0 + wave; row < r0 + rows_per_block && row < rows; row += 4) {
    const float rs = rstd[row];
    u32x4_t xv[NCH], dv[NCH];
    float dot = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = c * 512 + lane * 8;
      if (col < dim) {
        xv[c] = *reinterpret_cast<const u32x4_t*>(x + row * dim + col);
        dv[c] = *reinterpret_cast<const u32x4_t*>(dy + row * dim + col);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float x0 = bflo(xv[c][e]) * rs, x1 = bfhi(xv[c][e]) * rs;
          const float d0 = bflo(dv[c][e]), d1 = bfhi(dv[c][e]);
          dwacc[c][2 * e] += d0 * x0; dwacc[c][2 * e + 1] += d1 * x1;
          dot += d0 * wv[c][2 * e] * x0 + d1 * wv[c][2 * e + 1] * x1;
        }
      }
    }
    dot = wave_sum(dot);
    const float m = dot / (float)dim;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = c * 512 + lane * 8;
      if (col < dim) {
        u32x4_t rv = {0u, 0u, 0u, 0u};
        if (dres) rv = *reinterpret_cast<const u32x4_t*>(dres + row * dim + col);
        u32x4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float x0 = bflo(xv[c][e]) * rs, x1 = bfhi(xv[c][e]) * rs;
          float a = rs * (bflo(dv[c][e]) * wv[c][2 * e] - x0 * m);
          float b = rs * (bfhi(dv[c][e]) * wv[c][2 * e + 1] - x1 * m);
          if (dres) {  // gradient of the residual branch joins here: bf16(dx) + dres, rounded as the eager add would
            a = bf2f(f2bf(a)) + bflo(rv[e]);
            b = bf2f(f2bf(b)) + bfhi(rv[e]);
          }
          o[e] = pack_bf2(a, b);
        }
        *reinterpret_cast<u32x4_t*>(dx + row * dim + col) = o;
      }
    }
  }
  if (dw_partial) {
    for (int wsel = 0; wsel < 4; ++wsel) {  // waves add their partial sums into LDS one after the other
      if (wave == wsel) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
          const int col = c * 512 + lane * 8;
          if (col < dim) {
#pragma unroll
            for (int e = 0; e < 8; ++e) dwsh[col + e] = (wsel == 0 ? 0.f : dwsh[col + e]) + dwacc[c][e];
          }
        }
      }
      __syncthreads();
    }
    for (int col = threadIdx.x * 4; col < dim; col += 256 * 4)
      *reinterpret_cast<f32x4_t*>(dw_partial + (int64_t)blockIdx.x * dim + col) = *reinterpret_cast<const f32x4_t*>(dwsh + col);
  }
}

// dw[col] = bf16( sum_p partial[p][col] ), optionally accumulating into an existing bf16 grad.
// Block = 64 columns x 4 row groups (256 threads): row groups split the partial rows, LDS combines them.
__global__ __launch_bounds__(256) void colsum_partials_kernel(const float* __restrict__ partial, bf16_t* __restrict__ out, int nparts, int dim,
                                                              int accumulate) {
  __shared__ float red[4][64];
  const int c = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + c;
  float s = 0.f;
  if (col < dim)
    for (int p = grp; p < nparts; p += 4) s += partial[(int64_t)p * dim + col];
  red[grp][c] = s;
  __syncthreads();
  if (grp == 0 && col < dim) {
    s = red[0][c] + red[1][c] + red[2][c] + red[3][c];
    if (accumulate) s += bf2f(out[col]);
    out[col] = f2bf(s);
  }
}

extern "C" int llx_rmsnorm_fwd(const void* x, const void* w, void* y, float* rstd, int64_t rows, int64_t dim, float eps,
                               hipStream_t stream) {
  LLX_REQUIRE(x && w && y, "llx_rmsnorm_fwd: null pointer");
  LLX_REQUIRE(rows >= 0 && dim > 0 && dim % 8 == 0 && dim <= 8192, "llx_rmsnorm_fwd: dim=%lld must be a multiple of 8 and <= 8192",
              (long long)dim);
  if (rows == 0) return LLX_OK;
  const dim3 grid((unsigned)cdiv64(rows, 4)), block(256);
  const int nch = (int)cdiv64(dim, 512);
#define L(N) hipLaunchKernelGGL(rmsnorm_fwd_kernel<N>, grid, block, 0, stream, (const bf16_t*)x, (const bf16_t*)w, (bf16_t*)y, rstd, rows, (int)dim, eps)
  if (nch <= 1) L(1); else if (nch <= 2) L(2); else if (nch <= 4) L(4); else if (nch <= 8) L(8); else L(16);
#undef L
  LLX_LAUNCH_CHECK("llx_rmsnorm_fwd");
  return LLX_OK;
}

extern "C" int64_t llx_rmsnorm_bwd_workspace_bytes(int64_t rows, int64_t dim) {
  const int64_t rpb = 16;
  return cdiv64(rows, rpb) * dim * 4;
}

// dw may be null (frozen norm weight). workspace: llx_rmsnorm_bwd_workspace_bytes(rows, dim) bytes of fp32.
// dres (nullable): gradient arriving through the residual connection around the normed branch; added to dx in the same pass.
extern "C" int llx_rmsnorm_bwd(const void* dy, const void* x, const void* w, const float* rstd, void* dx, void* dw,
                               int dw_accumulate, void* workspace, const void* dres, int64_t rows, int64_t dim, hipStream_t stream) {
  LLX_REQUIRE(dy && x && w && rstd && dx, "llx_rmsnorm_bwd: null pointer");
  LLX_REQUIRE(dim > 0 && dim % 8 == 0 && dim <= 8192, "llx_rmsnorm_bwd: dim=%lld must be a multiple of 8 and <= 8192", (long long)dim);
  LLX_REQUIRE(!dw || workspace, "llx_rmsnorm_bwd: workspace required when dw is requested");
  if (rows == 0) return LLX_OK;
  const int rpb = 16;
  const int nblk = (int)cdiv64(rows, rpb);
  const int nch = (int)cdiv64(dim, 512);
  float* part = dw ? (float*)workspace : nullptr;
  const size_t lds = dw ? (size_t)dim * 4 : 0;
#define L(N) hipLaunchKernelGGL(rmsnorm_bwd_kernel<N>, dim3(nblk), dim3(256), lds, stream, (const bf16_t*)dy, (const bf16_t*)x, (const bf16_t*)w, rstd, (bf16_t*)dx, part, (const bf16_t*)dres, rows, (int)dim, rpb)
  if (nch <= 1) L(1); else if (nch <= 2) L(2); else if (nch <= 4) L(4); else if (nch <= 8) L(8); else L(16);
#undef L
  LLX_LAUNCH_CHECK("llx_rmsnorm_bwd");
  if (dw) {
    hipLaunchKernelGGL(colsum_partials_kernel, dim3((unsigned)cdiv64(dim, 64)), dim3(256), 0, stream, part, (bf16_t*)dw, nblk,
                       (int)dim, dw_accumulate);
    LLX_LAUNCH_CHECK("llx_rmsnorm_bwd(colsum)");
  }
  return LLX_OK;
}
