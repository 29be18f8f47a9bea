// Fused cross-entropy over bf16 logits: F.cross_entropy(logits.float(), labels) with ignore_index -100 and mean
// reduction (modelling/llama.py:216-218, modelling/audio.py:74-76).  The fp32 copy of the [T, V] logits (2.1 GB at
// T=4096, V=128256) is never materialised: one block streams a row twice (online max/sum, then gradient write).
#include "common.h"

#define CE_THREADS 512

// n_valid = #(labels != -100);  out[0] = 1/n_valid, out[1] = n_valid
__global__ void ce_count_kernel(const int64_t* __restrict__ labels, float* __restrict__ out, int64_t T) {
  __shared__ float red[16];
  float c = 0.f;
  for (int64_t i = threadIdx.x; i < T; i += blockDim.x) c += (labels[i] != -100) ? 1.f : 0.f;
  c = block_sum(c, red);
  if (threadIdx.x == 0) { out[0] = 1.f / c; out[1] = c; }
}

// One block per row.  row_loss[t] = lse - x[label] (0 for ignored rows).
// dlogits (nullable, may alias logits) = (softmax - onehot) * inv_count[0]   (zero row when ignored).
// logits / dlogits carry no __restrict__: the caller passes the same buffer for both (in-place gradient).
__global__ __launch_bounds__(CE_THREADS) void ce_row_kernel(const bf16_t* logits, bf16_t* dlogits, int64_t ld,
                                                             int64_t dld, const int64_t* __restrict__ labels, float* __restrict__ row_loss,
                                                             const float* __restrict__ inv_count, int V, const int32_t* __restrict__ rows_dyn,
                                                             int64_t row0) {
  // (row0: index of this launch's first row in the whole row set - a launch may cover a chunk of it; logits / dlogits / labels /
  //  row_loss are passed pre-offset, only the comparison with the global labelled-row count needs the absolute index)
  __shared__ float red[16];
  const int64_t t = blockIdx.x;
  if (rows_dyn != nullptr && row0 + t >= ((rows_dyn[0] + 255) & ~255)) {  // compacted rows: nothing reads past the last row tile of the GEMMs
    if (threadIdx.x == 0) row_loss[t] = 0.f;
    return;
  }
  const bf16_t* x = logits + t * ld;
  const int64_t label = labels[t];
  const bool valid = label != -100;
  const int nchunk = V >> 3;
  if (!valid) {
    if (threadIdx.x == 0) row_loss[t] = 0.f;
    if (dlogits) {
      const u32x4_t z = {0u, 0u, 0u, 0u};
      for (int c = threadIdx.x; c < nchunk; c += CE_THREADS) *reinterpret_cast<u32x4_t*>(dlogits + t * dld + c * 8) = z;
    }
    return;
  }
  // the label's logit is read BEFORE anything is written: pass 2 of another wave overwrites it in place when dlogits == logits
  const float xl = (label >= 0 && label < V) ? bf2f(x[label]) : 0.f;
  // pass 1: online (max, sum exp)
  float m = -INFINITY, s = 0.f;
  for (int c = threadIdx.x; c < nchunk; c += CE_THREADS) {
    const u32x4_t v = *reinterpret_cast<const u32x4_t*>(x + c * 8);
    float f[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) { f[2 * e] = bflo(v[e]); f[2 * e + 1] = bfhi(v[e]); }
    float cm = f[0];
#pragma unroll
    for (int e = 1; e < 8; ++e) cm = fmaxf(cm, f[e]);
    const float mn = fmaxf(m, cm);
    float add = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) add += __expf(f[e] - mn);
    s = s * __expf(m - mn) + add;
    m = mn;
  }
  const float gm = block_max(m, red);
  s = block_sum(s * __expf(m - gm), red);
  const float lse = gm + __logf(s);
  if (threadIdx.x == 0) row_loss[t] = lse - xl;
  if (!dlogits) return;
  const float gs = inv_count[0];
  bf16_t* dx = dlogits + t * dld;
  for (int c = threadIdx.x; c < nchunk; c += CE_THREADS) {
    const u32x4_t v = *reinterpret_cast<const u32x4_t*>(x + c * 8);
    u32x4_t o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float p0 = __expf(bflo(v[e]) - lse), p1 = __expf(bfhi(v[e]) - lse);
      if (c * 8 + 2 * e == label) p0 -= 1.f;
      if (c * 8 + 2 * e + 1 == label) p1 -= 1.f;
      o[e] = pack_bf2(p0 * gs, p1 * gs);
    }
    *reinterpret_cast<u32x4_t*>(dx + c * 8) = o;
  }
}

// loss = sum(row_loss) * inv_count[0]
__global__ void ce_reduce_kernel(const float* __restrict__ row_loss, const float* __restrict__ inv_count, float* __restrict__ loss, int64_t T) {
  __shared__ float red[16];
  float c = 0.f;
  for (int64_t i = threadIdx.x; i < T; i += blockDim.x) c += row_loss[i];
  c = block_sum(c, red);
  if (threadIdx.x == 0) loss[0] = c * inv_count[0];
}

// workspace: (T + 2) floats: [0] 1/n_valid, [1] n_valid, [2..] per-row losses.  loss: 1 float (device).
extern "C" int64_t llx_ce_workspace_bytes(int64_t T) { return (T + 2) * 4; }

static int ce_fwd_bwd_impl(const void* logits, int64_t ld, void* dlogits, int64_t dld, const int64_t* labels, float* loss, void* workspace,
                           int64_t T, int64_t V, const int32_t* rows_dyn, hipStream_t stream) {
  LLX_REQUIRE(logits && labels && loss && workspace, "llx_ce_fwd_bwd: null pointer");
  LLX_REQUIRE(V % 8 == 0 && ld % 8 == 0 && dld % 8 == 0, "llx_ce_fwd_bwd: V and row strides must be multiples of 8");
  LLX_REQUIRE(T > 0 && V > 0 && V < (1 << 30), "llx_ce_fwd_bwd: bad sizes");
  float* ws = (float*)workspace;
  hipLaunchKernelGGL(ce_count_kernel, dim3(1), dim3(1024), 0, stream, labels, ws, T);
  LLX_LAUNCH_CHECK("llx_ce_fwd_bwd(count)");
  hipLaunchKernelGGL(ce_row_kernel, dim3((unsigned)T), dim3(CE_THREADS), 0, stream, (const bf16_t*)logits, (bf16_t*)dlogits, ld, dld, labels,
                     ws + 2, ws, (int)V, rows_dyn, (int64_t)0);
  LLX_LAUNCH_CHECK("llx_ce_fwd_bwd(rows)");
  hipLaunchKernelGGL(ce_reduce_kernel, dim3(1), dim3(1024), 0, stream, ws + 2, ws, loss, T);
  LLX_LAUNCH_CHECK("llx_ce_fwd_bwd(reduce)");
  return LLX_OK;
}

extern "C" int llx_ce_fwd_bwd(const void* logits, int64_t ld, void* dlogits, int64_t dld, const int64_t* labels, float* loss, void* workspace,
                              int64_t T, int64_t V, hipStream_t stream) {
  return ce_fwd_bwd_impl(logits, ld, dlogits, dld, labels, loss, workspace, T, V, nullptr, stream);
}

// As llx_ce_fwd_bwd over COMPACTED rows (llx_head_compact_index): the labelled rows come first, *rows (device int32) is their number;
// rows past the 256-row tile that holds the last labelled row are neither read nor written (the row-limited GEMMs never touch them).
extern "C" int llx_ce_fwd_bwd_rows(const void* logits, int64_t ld, void* dlogits, int64_t dld, const int64_t* labels, float* loss,
                                   void* workspace, int64_t T, int64_t V, const int32_t* rows, hipStream_t stream) {
  LLX_REQUIRE(rows && (uintptr_t)rows % 4 == 0, "llx_ce_fwd_bwd_rows: rows must be a device int32 pointer");
  return ce_fwd_bwd_impl(logits, ld, dlogits, dld, labels, loss, workspace, T, V, rows, stream);
}

// The same loss over a row set too large for one logits buffer (the GEMM's 32-bit tile offsets stop at 4 GiB = 16.7 k rows of a
// 128 k vocabulary): the caller walks the rows in chunks, one call per chunk with that chunk's logits; `labels` and `workspace` cover
// ALL T rows.  phase bit 0 (first chunk): count the labelled rows of the whole set first; bit 1 (last chunk): reduce the per-row
// losses of the whole set into `loss`.  Every row sees the global 1 / n_valid, so the values are those of one call over all rows.
extern "C" int llx_ce_fwd_bwd_part(const void* logits, int64_t ld, void* dlogits, int64_t dld, const int64_t* labels, float* loss, void* workspace,
                                   int64_t T, int64_t V, int64_t row0, int64_t nrows, const int32_t* rows, int phase, hipStream_t stream) {
  LLX_REQUIRE(logits && labels && loss && workspace, "llx_ce_fwd_bwd_part: null pointer");
  LLX_REQUIRE(V % 8 == 0 && ld % 8 == 0 && dld % 8 == 0, "llx_ce_fwd_bwd_part: V and row strides must be multiples of 8");
  LLX_REQUIRE(T > 0 && V > 0 && V < (1 << 30) && row0 >= 0 && nrows > 0 && row0 + nrows <= T, "llx_ce_fwd_bwd_part: bad sizes");
  LLX_REQUIRE(rows == nullptr || (uintptr_t)rows % 4 == 0, "llx_ce_fwd_bwd_part: rows must be a device int32 pointer");
  float* ws = (float*)workspace;
  if (phase & 1) {
    hipLaunchKernelGGL(ce_count_kernel, dim3(1), dim3(1024), 0, stream, labels, ws, T);
    LLX_LAUNCH_CHECK("llx_ce_fwd_bwd_part(count)");
  }
  hipLaunchKernelGGL(ce_row_kernel, dim3((unsigned)nrows), dim3(CE_THREADS), 0, stream, (const bf16_t*)logits, (bf16_t*)dlogits, ld, dld, labels + row0,
                     ws + 2 + row0, ws, (int)V, rows, row0);
  LLX_LAUNCH_CHECK("llx_ce_fwd_bwd_part(rows)");
  if (phase & 2) {
    hipLaunchKernelGGL(ce_reduce_kernel, dim3(1), dim3(1024), 0, stream, ws + 2, ws, loss, T);
    LLX_LAUNCH_CHECK("llx_ce_fwd_bwd_part(reduce)");
  }
  return LLX_OK;
}

// ------------------------------------------------------------------------------------------ LM-head row compaction
// F.cross_entropy(ignore_index=-100) (modelling/llama.py:216-218): a position whose label is -100 adds nothing to the loss and has a
// zero gradient row, so the head's two GEMMs (logits, d hidden) only need the labelled rows.  The rows are compacted in order on the
// device (count, index and inverse index stay in device memory: no host round trip, capturable):
//   idx[j] = position of the j-th labelled row (-1 for j >= count), inv[i] = j or -1, labels_c[j] = labels[idx[j]] (-100 beyond), count[0]
__global__ __launch_bounds__(1024) void head_compact_index_kernel(const int64_t* __restrict__ labels, int32_t* __restrict__ idx,
                                                                   int32_t* __restrict__ inv, int64_t* __restrict__ labels_c,
                                                                   int32_t* __restrict__ count, int T) {
  __shared__ int wsum[16];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int base = 0;
  for (int c0 = 0; c0 < T; c0 += 1024) {
    const int i = c0 + threadIdx.x;
    const int64_t lab = i < T ? labels[i] : -100;
    const bool v = lab != -100;
    const unsigned long long m = __ballot(v);
    const int pre = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) wsum[w] = __popcll(m);
    __syncthreads();
    int woff = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int sk = wsum[k];
      woff += k < w ? sk : 0;
      tot += sk;
    }
    if (i < T) {
      if (v) {
        const int j = base + woff + pre;
        idx[j] = i;
        inv[i] = j;
        labels_c[j] = lab;
      } else {
        inv[i] = -1;
      }
    }
    base += tot;
    __syncthreads();
  }
  for (int j = base + threadIdx.x; j < T; j += 1024) {
    idx[j] = -1;
    labels_c[j] = -100;
  }
  if (threadIdx.x == 0) count[0] = base;
}

extern "C" int llx_head_compact_index(const int64_t* labels, int32_t* idx, int32_t* inv, int64_t* labels_c, int32_t* count, int64_t T,
                                      hipStream_t stream) {
  LLX_REQUIRE(labels && idx && inv && labels_c && count, "llx_head_compact_index: null pointer");
  LLX_REQUIRE(T > 0 && T < (1 << 30), "llx_head_compact_index: bad T");
  hipLaunchKernelGGL(head_compact_index_kernel, dim3(1), dim3(1024), 0, stream, labels, idx, inv, labels_c, count, (int)T);
  LLX_LAUNCH_CHECK("llx_head_compact_index");
  return LLX_OK;
}

// dst[j] = src[idx[j]] for j < count; zero rows up to the end of the 256-row tile of the last labelled row; later rows untouched
__global__ __launch_bounds__(256) void gather_rows_kernel(const bf16_t* __restrict__ src, int64_t lds_, const int32_t* __restrict__ idx,
                                                          const int32_t* __restrict__ count, bf16_t* __restrict__ dst, int64_t ldd, int D) {
  const int j = blockIdx.x, cnt = count[0];
  if (j >= ((cnt + 255) & ~255)) return;
  const int i = j < cnt ? idx[j] : -1;
  for (int c = threadIdx.x; c < (D >> 3); c += 256) {
    u32x4_t v = {0u, 0u, 0u, 0u};
    if (i >= 0) v = *reinterpret_cast<const u32x4_t*>(src + (int64_t)i * lds_ + c * 8);
    *reinterpret_cast<u32x4_t*>(dst + (int64_t)j * ldd + c * 8) = v;
  }
}

extern "C" int llx_gather_rows(const void* src, int64_t ld_src, const int32_t* idx, const int32_t* count, void* dst, int64_t ld_dst,
                               int64_t T, int64_t D, hipStream_t stream) {
  LLX_REQUIRE(src && idx && count && dst, "llx_gather_rows: null pointer");
  LLX_REQUIRE(T > 0 && D > 0 && D % 8 == 0 && ld_src % 8 == 0 && ld_dst % 8 == 0 && ((uintptr_t)src | (uintptr_t)dst) % 16 == 0,
              "llx_gather_rows: D and the row strides must be multiples of 8, pointers 16-byte aligned");
  hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)T), dim3(256), 0, stream, (const bf16_t*)src, ld_src, idx, count, (bf16_t*)dst, ld_dst, (int)D);
  LLX_LAUNCH_CHECK("llx_gather_rows");
  return LLX_OK;
}

// dst[i] = bf16(scale[0] * src[inv[i]]) where inv[i] >= 0, zero rows elsewhere (scale: nullable device float = the incoming d loss)
__global__ __launch_bounds__(256) void scatter_rows_kernel(const bf16_t* __restrict__ src, int64_t lds_, const int32_t* __restrict__ inv,
                                                           const float* __restrict__ scale, bf16_t* __restrict__ dst, int64_t ldd, int D) {
  const int i = blockIdx.x, j = inv[i];
  const float s = scale ? scale[0] : 1.f;
  for (int c = threadIdx.x; c < (D >> 3); c += 256) {
    u32x4_t v = {0u, 0u, 0u, 0u};
    if (j >= 0) {
      v = *reinterpret_cast<const u32x4_t*>(src + (int64_t)j * lds_ + c * 8);
      if (scale) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = pack_bf2(bflo(v[e]) * s, bfhi(v[e]) * s);
      }
    }
    *reinterpret_cast<u32x4_t*>(dst + (int64_t)i * ldd + c * 8) = v;
  }
}

extern "C" int llx_scatter_rows(const void* src, int64_t ld_src, const int32_t* inv, const float* scale, void* dst, int64_t ld_dst, int64_t T,
                                int64_t D, hipStream_t stream) {
  LLX_REQUIRE(src && inv && dst, "llx_scatter_rows: null pointer");
  LLX_REQUIRE(T > 0 && D > 0 && D % 8 == 0 && ld_src % 8 == 0 && ld_dst % 8 == 0 && ((uintptr_t)src | (uintptr_t)dst) % 16 == 0,
              "llx_scatter_rows: D and the row strides must be multiples of 8, pointers 16-byte aligned");
  hipLaunchKernelGGL(scatter_rows_kernel, dim3((unsigned)T), dim3(256), 0, stream, (const bf16_t*)src, ld_src, inv, scale, (bf16_t*)dst, ld_dst, (int)D);
  LLX_LAUNCH_CHECK("llx_scatter_rows");
  return LLX_OK;
}
