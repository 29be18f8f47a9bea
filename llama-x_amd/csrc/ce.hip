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
                                                             const float* __restrict__ inv_count, int V) {
  __shared__ float red[16];
  const int64_t t = blockIdx.x;
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

extern "C" int llx_ce_fwd_bwd(const void* logits, int64_t ld, void* dlogits, int64_t dld, const int64_t* labels, float* loss, void* workspace,
                              int64_t T, int64_t V, hipStream_t stream) {
  LLX_REQUIRE(logits && labels && loss && workspace, "llx_ce_fwd_bwd: null pointer");
  LLX_REQUIRE(V % 8 == 0 && ld % 8 == 0 && dld % 8 == 0, "llx_ce_fwd_bwd: V and row strides must be multiples of 8");
  LLX_REQUIRE(T > 0 && V > 0 && V < (1 << 30), "llx_ce_fwd_bwd: bad sizes");
  float* ws = (float*)workspace;
  hipLaunchKernelGGL(ce_count_kernel, dim3(1), dim3(1024), 0, stream, labels, ws, T);
  LLX_LAUNCH_CHECK("llx_ce_fwd_bwd(count)");
  hipLaunchKernelGGL(ce_row_kernel, dim3((unsigned)T), dim3(CE_THREADS), 0, stream, (const bf16_t*)logits, (bf16_t*)dlogits, ld, dld, labels,
                     ws + 2, ws, (int)V);
  LLX_LAUNCH_CHECK("llx_ce_fwd_bwd(rows)");
  hipLaunchKernelGGL(ce_reduce_kernel, dim3(1), dim3(1024), 0, stream, ws + 2, ws, loss, T);
  LLX_LAUNCH_CHECK("llx_ce_fwd_bwd(reduce)");
  return LLX_OK;
}
