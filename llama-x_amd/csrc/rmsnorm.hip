// RMSNorm forward / backward for bf16 rows (HBM-bound; one pass over x).
// Semantics follow nn.RMSNorm(D, eps) as used at reference modelling/llama.py:158,160,182:
//   y = bf16( float(x) * rsqrt(mean(x^2) + eps) * float(w) )      -- single rounding.
#include "common.h"

// ---------------------------------------------------------------- forward
// One wave per row, 4 rows per 256-thread block. Row values are kept in registers
// (NCH chunks of 512 elements; lane owns 8 contiguous bf16 = one 16-B load per chunk).
// QUANT: the bf16 output row is also quantised row-wise to int8 (quantize_int8_rowwise of subclasses/int8.py:10-16 applied to y, as
// _Int8Linear does to its input at :110-113 when dynamic_int8_act is set): absmax of the ROUNDED outputs / 127 in fp32, IEEE divide,
// round half to even - bit-identical to llx_quantize_int8_rowwise(y), without reading y back.
// FULL: dim == NCH * 512 (no per-chunk guards: the loads of a row are issued back to back)
template <int NCH, bool QUANT = false, bool FULL = false>
__global__ __launch_bounds__(256) void rmsnorm_fwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                          bf16_t* __restrict__ y, float* __restrict__ rstd_out,
                                                          int64_t rows, int dim, float eps, int8_t* __restrict__ q = nullptr,
                                                          int64_t ldq = 0, bf16_t* __restrict__ qscale = nullptr) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const bf16_t* xr = x + row * dim;
  u32x4_t v[NCH];
  float ss = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int col = c * 512 + lane * 8;
    if (FULL || col < dim) {
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
  float amax = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int col = c * 512 + lane * 8;
    if (FULL || col < dim) {
      u32x4_t wv = *reinterpret_cast<const u32x4_t*>(w + col);
      u32x4_t o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float a = bflo(v[c][e]) * rstd * bflo(wv[e]);
        float b = bfhi(v[c][e]) * rstd * bfhi(wv[e]);
        o[e] = pack_bf2(a, b);
        if constexpr (QUANT) amax = fmaxf(amax, fmaxf(fabsf(bflo(o[e])), fabsf(bfhi(o[e]))));
      }
      *reinterpret_cast<u32x4_t*>(yr + col) = o;
      if constexpr (QUANT) v[c] = o;  // the rounded outputs replace the inputs in registers for the quantising pass
    }
  }
  if constexpr (QUANT) {
    amax = wave_max(amax);
    const float scale = amax / 127.0f;
    const float div = fmaxf(scale, 1e-12f);
    if (lane == 0) qscale[row] = f2bf(scale);
    int8_t* qr = q + row * ldq;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = c * 512 + lane * 8;
      if (FULL || col < dim) {
        u32x2_t o = {0u, 0u};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int a = (int)rintf(bflo(v[c][e]) / div), b = (int)rintf(bfhi(v[c][e]) / div);
          o[e >> 1] |= ((uint32_t)(a & 0xff) | ((uint32_t)(b & 0xff) << 8)) << ((e & 1) * 16);
        }
        *reinterpret_cast<u32x2_t*>(qr + col) = o;
      }
    }
  }
}

// ---------------------------------------------------------------- backward
// One WAVE per row (no block-wide barriers): a 256-thread block = 4 waves walks `rows_per_block` rows, wave w taking rows
// w, w+4, ...; lane l owns columns {c*512 + l*8 .. +8}.  dx is written per row; the per-lane dw partial sums stay in
// registers, are combined across the 4 waves through LDS and written once per block to dw_partial[blockIdx][dim] (fp32).
//   xhat = x*rstd ; g = dy*w ; dx = rstd * (g - xhat * mean(g*xhat)) [+ dres] ; dw = sum_rows(dy*xhat)
// FULL: dim == NCH * 512 - every chunk of every lane is inside the row, the per-chunk guards (and the basic-block boundaries they put
// between the loads, which made the compiler drain vmcnt to 0 before every prefetch) compile away.
template <int NCH, bool FULL = false, int RES = -1>  // RES: 1 / 0 = dres known present / absent at compile time (FULL kernels), -1 = runtime
__global__ __launch_bounds__(256) void rmsnorm_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                          const bf16_t* __restrict__ w, const float* __restrict__ rstd,
                                                          bf16_t* __restrict__ dx, float* __restrict__ dw_partial,
                                                          const bf16_t* __restrict__ dres, int64_t rows, int dim, int rows_per_block) {
  extern __shared__ __attribute__((aligned(16))) float dwsh[];  // [dim] when dw is requested
  const bool has_res = RES < 0 ? dres != nullptr : RES == 1;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  u32x4_t wp[NCH];  // norm weight, kept packed (unpacked where it is used)
  float dwacc[NCH][8];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int col = c * 512 + lane * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) dwacc[c][e] = 0.f;
    wp[c] = u32x4_t{0u, 0u, 0u, 0u};
    if (FULL || col < dim) wp[c] = *reinterpret_cast<const u32x4_t*>(w + col);
  }
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r_end = (r0 + rows_per_block < rows) ? r0 + rows_per_block : rows;
  // Two register sets: the loads of the wave's NEXT row are in flight while the current row is reduced and written (a
  // block is resident once per CU at this size, so the wave has to hide its own HBM latency).  dim 8192 keeps one set.
  constexpr bool PIPE = NCH <= 8;
  struct RowRegs { u32x4_t x[NCH], d[NCH], r[NCH]; float rs; };
  auto load_row = [&](RowRegs& R, int64_t row) {
    R.rs = rstd[row];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = c * 512 + lane * 8;
      if (FULL || col < dim) {
        R.x[c] = *reinterpret_cast<const u32x4_t*>(x + row * dim + col);
        R.d[c] = *reinterpret_cast<const u32x4_t*>(dy + row * dim + col);
        if (has_res) R.r[c] = *reinterpret_cast<const u32x4_t*>(dres + row * dim + col);
      }
    }
  };
  auto process_row = [&](RowRegs& R, int64_t row) {
    const float rs = R.rs;
    float dot = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = c * 512 + lane * 8;
      if (FULL || col < dim) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float x0 = bflo(R.x[c][e]) * rs, x1 = bfhi(R.x[c][e]) * rs;
          const float d0 = bflo(R.d[c][e]), d1 = bfhi(R.d[c][e]);
          dwacc[c][2 * e] += d0 * x0; dwacc[c][2 * e + 1] += d1 * x1;
          dot += d0 * bflo(wp[c][e]) * x0 + d1 * bfhi(wp[c][e]) * x1;
        }
      }
    }
    dot = wave_sum(dot);
    const float m = dot / (float)dim;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = c * 512 + lane * 8;
      if (FULL || col < dim) {
        u32x4_t o;
        // opaque copy: without it hipcc keeps the 64 unpacked, scaled floats of the first pass alive instead of the 16 packed
        // registers (512 VGPRs + scratch with two row sets in flight)
        asm volatile("" : "+v"(R.x[c]), "+v"(R.d[c]));
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float x0 = bflo(R.x[c][e]) * rs, x1 = bfhi(R.x[c][e]) * rs;
          float a = rs * (bflo(R.d[c][e]) * bflo(wp[c][e]) - x0 * m);
          float b = rs * (bfhi(R.d[c][e]) * bfhi(wp[c][e]) - x1 * m);
          if (has_res) {  // gradient of the residual branch joins here: bf16(dx) + dres, rounded as the eager add would
            a = bf2f(f2bf(a)) + bflo(R.r[c][e]);
            b = bf2f(f2bf(b)) + bfhi(R.r[c][e]);
          }
          o[e] = pack_bf2(a, b);
        }
        *reinterpret_cast<u32x4_t*>(dx + row * dim + col) = o;
      }
    }
  };
  {
    RowRegs A;
    int64_t row = r0 + wave;
    if constexpr (PIPE) {
      RowRegs B;
      if (row < r_end) load_row(A, row);
      while (row < r_end) {
        const bool n1 = row + 4 < r_end;
        if (n1) load_row(B, row + 4);
        process_row(A, row);
        if (!n1) break;
        row += 4;
        const bool n2 = row + 4 < r_end;
        if (n2) load_row(A, row + 4);
        process_row(B, row);
        if (!n2) break;
        row += 4;
      }
    } else {
      for (; row < r_end; row += 4) { load_row(A, row); process_row(A, row); }
    }
  }
  if (dw_partial) {
    for (int wsel = 0; wsel < 4; ++wsel) {  // waves add their partial sums into LDS one after the other
      if (wave == wsel) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
          const int col = c * 512 + lane * 8;
          if (FULL || col < dim) {
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
// Block = 16 columns x 16 row groups (256 threads) so that a [256, 4096] partial array is summed by 256 blocks with 16
// dependent loads per thread (64 columns x 4 groups left 64 blocks with 64 loads each: latency-bound at 17 us).
__global__ __launch_bounds__(256) void colsum_partials_kernel(const float* __restrict__ partial, bf16_t* __restrict__ out, int nparts, int dim,
                                                              int accumulate) {
  __shared__ float red[16][17];
  const int c = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const int col = blockIdx.x * 16 + c;
  float s = 0.f;
  if (col < dim)
    for (int p = grp; p < nparts; p += 16) s += partial[(int64_t)p * dim + col];
  red[grp][c] = s;
  __syncthreads();
  if (grp == 0 && col < dim) {
    s = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) s += red[g][c];
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
#define L(N) hipLaunchKernelGGL((rmsnorm_fwd_kernel<N, false>), grid, block, 0, stream, (const bf16_t*)x, (const bf16_t*)w, (bf16_t*)y, rstd, rows, (int)dim, eps, (int8_t*)nullptr, (int64_t)0, (bf16_t*)nullptr)
#define LF(N) hipLaunchKernelGGL((rmsnorm_fwd_kernel<N, false, true>), grid, block, 0, stream, (const bf16_t*)x, (const bf16_t*)w, (bf16_t*)y, rstd, rows, (int)dim, eps, (int8_t*)nullptr, (int64_t)0, (bf16_t*)nullptr)
  if (dim == 4096) LF(8); else if (dim == 2048) LF(4); else if (dim == 8192) LF(16);
  else if (nch <= 1) L(1); else if (nch <= 2) L(2); else if (nch <= 4) L(4); else if (nch <= 8) L(8); else L(16);
#undef LF
#undef L
  LLX_LAUNCH_CHECK("llx_rmsnorm_fwd");
  return LLX_OK;
}

// RMSNorm forward that also emits quantize_int8_rowwise(y): q int8 [rows, dim] (row stride ldq), qscale bf16 [rows].
extern "C" int llx_rmsnorm_fwd_quant(const void* x, const void* w, void* y, float* rstd, void* q, int64_t ldq, void* qscale, int64_t rows,
                                     int64_t dim, float eps, hipStream_t stream) {
  LLX_REQUIRE(x && w && y && q && qscale, "llx_rmsnorm_fwd_quant: null pointer");
  LLX_REQUIRE(rows >= 0 && dim > 0 && dim % 8 == 0 && dim <= 8192 && ldq % 8 == 0 && (uintptr_t)q % 8 == 0,
              "llx_rmsnorm_fwd_quant: dim=%lld must be a multiple of 8 and <= 8192, q rows 8-byte aligned", (long long)dim);
  if (rows == 0) return LLX_OK;
  const dim3 grid((unsigned)cdiv64(rows, 4)), block(256);
  const int nch = (int)cdiv64(dim, 512);
#define L(N) hipLaunchKernelGGL((rmsnorm_fwd_kernel<N, true>), grid, block, 0, stream, (const bf16_t*)x, (const bf16_t*)w, (bf16_t*)y, rstd, rows, (int)dim, eps, (int8_t*)q, ldq, (bf16_t*)qscale)
#define LF(N) hipLaunchKernelGGL((rmsnorm_fwd_kernel<N, true, true>), grid, block, 0, stream, (const bf16_t*)x, (const bf16_t*)w, (bf16_t*)y, rstd, rows, (int)dim, eps, (int8_t*)q, ldq, (bf16_t*)qscale)
  if (dim == 4096) LF(8); else if (dim == 2048) LF(4); else if (dim == 8192) LF(16);
  else if (nch <= 1) L(1); else if (nch <= 2) L(2); else if (nch <= 4) L(4); else if (nch <= 8) L(8); else L(16);
#undef LF
#undef L
  LLX_LAUNCH_CHECK("llx_rmsnorm_fwd_quant");
  return LLX_OK;
}

#ifndef RMS_BWD_RPB
#define RMS_BWD_RPB 16  // rows per 4-wave block (4 per wave): one block per CU at [4096, 4096], one dw partial row per block
#endif
extern "C" int64_t llx_rmsnorm_bwd_workspace_bytes(int64_t rows, int64_t dim) {
  const int64_t rpb = RMS_BWD_RPB;
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
  const int rpb = RMS_BWD_RPB;
  const int nblk = (int)cdiv64(rows, rpb);
  const int nch = (int)cdiv64(dim, 512);
  float* part = dw ? (float*)workspace : nullptr;
  const size_t lds = dw ? (size_t)dim * 4 : 0;
#define L(N) hipLaunchKernelGGL(rmsnorm_bwd_kernel<N>, dim3(nblk), dim3(256), lds, stream, (const bf16_t*)dy, (const bf16_t*)x, (const bf16_t*)w, rstd, (bf16_t*)dx, part, (const bf16_t*)dres, rows, (int)dim, rpb)
#define LF(N) if (dres) LFR(N, 1); else LFR(N, 0)
#define LFR(N, R) hipLaunchKernelGGL((rmsnorm_bwd_kernel<N, true, R>), dim3(nblk), dim3(256), lds, stream, (const bf16_t*)dy, (const bf16_t*)x, (const bf16_t*)w, rstd, (bf16_t*)dx, part, (const bf16_t*)dres, rows, (int)dim, rpb)
  if (dim == 4096) { LF(8); } else if (dim == 2048) { LF(4); } else if (dim == 8192) { LF(16); }
  else if (nch <= 1) L(1); else if (nch <= 2) L(2); else if (nch <= 4) L(4); else if (nch <= 8) L(8); else L(16);
#undef LFR
#undef LF
#undef L
  LLX_LAUNCH_CHECK("llx_rmsnorm_bwd");
  if (dw) {
    hipLaunchKernelGGL(colsum_partials_kernel, dim3((unsigned)cdiv64(dim, 16)), dim3(256), 0, stream, part, (bf16_t*)dw, nblk,
                       (int)dim, dw_accumulate);
    LLX_LAUNCH_CHECK("llx_rmsnorm_bwd(colsum)");
  }
  return LLX_OK;
}
