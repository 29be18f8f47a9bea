// Dense-mask attention forward (inference / KV-cache path): SDPA(q, k, v, mask, is_causal=False, enable_gqa=True) of
// modelling/llama.py:135-137 when a KV cache or an explicit bool mask is in play (:126-127, :189-194, :205).
// Query length is small there (1 for decode), keys are the whole cache, so this is a bandwidth-bound sweep over K/V:
// one wave per (batch, head, query row); lane j scores key 64c+j, the wave reduces the online-softmax statistics, then
// each lane accumulates its two output dims over the 64 keys of the chunk.  Forward only.
#include "common.h"

#define HD 128

struct DenseArgs {
  const bf16_t* q; const bf16_t* k; const bf16_t* v; bf16_t* o; const uint8_t* mask;
  int64_t q_sb, q_sh, q_ss, k_sb, k_sh, k_ss, v_sb, v_sh, v_ss, o_sb, o_sh, o_ss;
  int64_t m_sb, m_sh, m_sq;  // mask strides (elements); last dim contiguous; zeros broadcast
  int B, H, KVH, Sq, Skv;
  float scale;
};

__global__ __launch_bounds__(64) void attn_dense_fwd_kernel(const DenseArgs a) {
  __shared__ float qs[HD];
  const int lane = threadIdx.x;
  const int qi = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int kvh = h / (a.H / a.KVH);
  const bf16_t* qp = a.q + b * a.q_sb + h * a.q_sh + (int64_t)qi * a.q_ss;
  qs[2 * lane] = bf2f(qp[2 * lane]);
  qs[2 * lane + 1] = bf2f(qp[2 * lane + 1]);
  __syncthreads();
  const bf16_t* kb = a.k + b * a.k_sb + kvh * a.k_sh;
  const bf16_t* vb = a.v + b * a.v_sb + kvh * a.v_sh;
  const uint8_t* mrow = a.mask + b * a.m_sb + h * a.m_sh + (int64_t)qi * a.m_sq;
  float m_run = -INFINITY, l_run = 0.f, o0 = 0.f, o1 = 0.f;
  for (int c0 = 0; c0 < a.Skv; c0 += 64) {
    const int kk = c0 + lane;
    float s = -INFINITY;
    if (kk < a.Skv && mrow[kk]) {
      const bf16_t* kr = kb + (int64_t)kk * a.k_ss;
      float acc = 0.f;
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const u32x4_t v = *reinterpret_cast<const u32x4_t*>(kr + c * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc += bflo(v[e]) * qs[c * 8 + 2 * e] + bfhi(v[e]) * qs[c * 8 + 2 * e + 1];
      }
      s = acc * a.scale;
    }
    const float mx = wave_max(s);
    if (mx == -INFINITY) continue;  // whole chunk masked (wave-uniform)
    const float m_new = fmaxf(m_run, mx);
    const float alpha = __expf(m_run - m_new);
    const float p = __expf(s - m_new);  // exp(-inf) = 0 for masked keys
    l_run = l_run * alpha + wave_sum(p);
    o0 *= alpha; o1 *= alpha;
    m_run = m_new;
    const int n = min(64, a.Skv - c0);
    for (int j = 0; j < n; ++j) {
      const float pj = __shfl(p, j, 64);
      if (pj != 0.f) {  // wave-uniform
        const uint32_t w = *reinterpret_cast<const uint32_t*>(vb + (int64_t)(c0 + j) * a.v_ss + 2 * lane);
        o0 += pj * bflo(w);
        o1 += pj * bfhi(w);
      }
    }
  }
  // a fully masked row yields NaN in SDPA (softmax of all -inf); mirror that
  const float inv = 1.f / l_run;
  bf16_t* op = a.o + b * a.o_sb + h * a.o_sh + (int64_t)qi * a.o_ss;
  *reinterpret_cast<uint32_t*>(op + 2 * lane) = pack_bf2(o0 * inv, o1 * inv);
}

// q [B,H,Sq,128], k/v [B,KVH,Skv,128], o [B,H,Sq,128] through (batch, head, seq) element strides, last dim dense.
// mask: uint8/bool [.., Sq, Skv] with broadcast strides (m_sb, m_sh, m_sq), last dim dense; nonzero = attend.
extern "C" int llx_attn_dense_fwd(const void* q, int64_t q_sb, int64_t q_sh, int64_t q_ss, const void* k, int64_t k_sb, int64_t k_sh,
                                  int64_t k_ss, const void* v, int64_t v_sb, int64_t v_sh, int64_t v_ss, void* o, int64_t o_sb, int64_t o_sh,
                                  int64_t o_ss, const void* mask, int64_t m_sb, int64_t m_sh, int64_t m_sq, int64_t B, int64_t H, int64_t KVH,
                                  int64_t Sq, int64_t Skv, int64_t head_dim, float scale, hipStream_t stream) {
  LLX_REQUIRE(q && k && v && o && mask, "llx_attn_dense_fwd: null pointer");
  LLX_REQUIRE(head_dim == HD, "llx_attn_dense_fwd: head_dim=%lld unsupported (only 128)", (long long)head_dim);
  LLX_REQUIRE(B > 0 && H > 0 && KVH > 0 && H % KVH == 0 && Sq > 0 && Skv > 0, "llx_attn_dense_fwd: bad sizes");
  LLX_REQUIRE(((k_sb | k_sh | k_ss) % 8) == 0 && (uintptr_t)k % 16 == 0, "llx_attn_dense_fwd: K rows must be 16-byte aligned");
  LLX_REQUIRE(((v_sb | v_sh | v_ss | o_sb | o_sh | o_ss) % 2) == 0, "llx_attn_dense_fwd: V/O strides must be even");
  DenseArgs a;
  a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.o = (bf16_t*)o; a.mask = (const uint8_t*)mask;
  a.q_sb = q_sb; a.q_sh = q_sh; a.q_ss = q_ss; a.k_sb = k_sb; a.k_sh = k_sh; a.k_ss = k_ss; a.v_sb = v_sb; a.v_sh = v_sh; a.v_ss = v_ss;
  a.o_sb = o_sb; a.o_sh = o_sh; a.o_ss = o_ss; a.m_sb = m_sb; a.m_sh = m_sh; a.m_sq = m_sq;
  a.B = (int)B; a.H = (int)H; a.KVH = (int)KVH; a.Sq = (int)Sq; a.Skv = (int)Skv; a.scale = scale;
  hipLaunchKernelGGL(attn_dense_fwd_kernel, dim3((unsigned)Sq, (unsigned)H, (unsigned)B), dim3(64), 0, stream, a);
  LLX_LAUNCH_CHECK("llx_attn_dense_fwd");
  return LLX_OK;
}
