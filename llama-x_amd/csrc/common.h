// Shared device/host helpers for the llx HIP library (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdarg.h>
#include <stdio.h>

#define LLX_OK 0
#define LLX_ERR_ARG -1      // bad shape / alignment / null pointer
#define LLX_ERR_LAUNCH -2   // hipLaunch failure
#define LLX_ERR_UNSUPPORTED -3

// thread-local last-error string (include/llx.h: llx_last_error_string)
void llx_set_error(const char* fmt, ...);

#define LLX_REQUIRE(cond, ...)            \
  do {                                    \
    if (!(cond)) {                        \
      llx_set_error(__VA_ARGS__);         \
      return LLX_ERR_ARG;                 \
    }                                     \
  } while (0)

#define LLX_LAUNCH_CHECK(name)                                              \
  do {                                                                      \
    hipError_t e__ = hipGetLastError();                                     \
    if (e__ != hipSuccess) {                                                \
      llx_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
      return LLX_ERR_LAUNCH;                                                \
    }                                                                       \
  } while (0)

typedef uint16_t bf16_t;  // raw bf16 bits

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) int i32x4_t;
typedef __attribute__((ext_vector_type(16))) int i32x16_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;

#define LLX_WAVE 64

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// round-to-nearest-even, NaN preserving (hipcc emits v_cvt_pk_bf16_f32)
__device__ __forceinline__ bf16_t f2bf(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(bf16_t, b);
}
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
  return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
}
__device__ __forceinline__ float bflo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bfhi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Block-wide sum for blockDim.x <= 1024 (<=16 waves). `red` is >=16 floats of LDS.
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < nw; ++i) t += red[i];
  return t;
}
__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  float t = red[0];
  for (int i = 1; i < nw; ++i) t = fmaxf(t, red[i]);
  return t;
}

static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---- SwiGLU pieces shared by the element-wise kernels and the GEMM epilogue (h = silu(g) * u, modelling/llama.py:150)
// v_rcp_f32 (1 ulp) instead of the IEEE divide sequence: the result is rounded to bf16 right after
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
// 8 packed bf16 elements: h = (silu(g) rounded) * u, rounded  -- the two roundings of the bf16 eager graph
__device__ __forceinline__ u32x4_t swiglu_fwd8(const u32x4_t& gv, const u32x4_t& uv) {
  u32x4_t o;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float g0 = bflo(gv[e]), g1 = bfhi(gv[e]);
    const float s0 = bf2f(f2bf(g0 * sigmoidf_(g0))), s1 = bf2f(f2bf(g1 * sigmoidf_(g1)));
    o[e] = pack_bf2(s0 * bflo(uv[e]), s1 * bfhi(uv[e]));
  }
  return o;
}
// 8 packed bf16 elements: dg = (dh*u rounded) * silu'(g), du = dh * (silu(g) rounded)  -- the roundings autograd's bf16 graph makes
__device__ __forceinline__ void swiglu_bwd8(const u32x4_t& dv, const u32x4_t& gv, const u32x4_t& uv, u32x4_t& og, u32x4_t& ou) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float dgs[2], dus[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const float gg = p ? bfhi(gv[e]) : bflo(gv[e]);
      const float uu = p ? bfhi(uv[e]) : bflo(uv[e]);
      const float dd = p ? bfhi(dv[e]) : bflo(dv[e]);
      const float sg = sigmoidf_(gg);
      const float silu = bf2f(f2bf(gg * sg));
      dus[p] = dd * silu;
      const float ds = bf2f(f2bf(dd * uu));
      dgs[p] = ds * (sg * (1.f + gg * (1.f - sg)));
    }
    og[e] = pack_bf2(dgs[0], dgs[1]);
    ou[e] = pack_bf2(dus[0], dus[1]);
  }
}

// ---- RoPE on 8 packed bf16 elements = 4 interleaved pairs (modelling/llama.py:63-73): tp -> table[(s*64 + j)*2] for the first
// pair j of the chunk (fp32 cos, sin interleaved); sign = +1 forward, -1 backward (rotation by -theta is the exact transpose).
__device__ __forceinline__ u32x4_t rope8(const u32x4_t& v, const f32x4_t& t0, const f32x4_t& t1, float sign) {  // table values preloaded
  const float cs[4] = {t0[0], t0[2], t1[0], t1[2]};
  const float sn[4] = {t0[1] * sign, t0[3] * sign, t1[1] * sign, t1[3] * sign};
  u32x4_t o;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float x0 = bflo(v[e]), x1 = bfhi(v[e]);
    o[e] = pack_bf2(x0 * cs[e] - x1 * sn[e], x1 * cs[e] + x0 * sn[e]);
  }
  return o;
}

__device__ __forceinline__ u32x4_t rope8(const u32x4_t& v, const float* __restrict__ tp, float sign) {
  const f32x4_t t0 = *reinterpret_cast<const f32x4_t*>(tp);
  const f32x4_t t1 = *reinterpret_cast<const f32x4_t*>(tp + 4);
  const float cs[4] = {t0[0], t0[2], t1[0], t1[2]};
  const float sn[4] = {t0[1] * sign, t0[3] * sign, t1[1] * sign, t1[3] * sign};
  u32x4_t o;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float x0 = bflo(v[e]), x1 = bfhi(v[e]);
    o[e] = pack_bf2(x0 * cs[e] - x1 * sn[e], x1 * cs[e] + x0 * sn[e]);
  }
  return o;
}

// ---- ds_read_b64_tr_b16 as inline asm.  Through the builtin hipcc treats the transposed read as an LDS access it cannot order against
// the LDS-DMA writes in flight and puts `s_waitcnt vmcnt(0)` in front of it: every prefetched tile is drained before the first
// transposed read of a tile (seen in the .s of the attention kernels; it made the four-stage ring of the dQ-from-dS kernel synchronous,
// 1.7 us per tile, and cut the forward kernel's K/V prefetch off before its PV phase).  As asm the read is invisible to that pass;
// ordering is by hand: the issuing wave's counted vmcnt + barrier before the read (LDS-DMA RAW), an lgkmcnt wait that names every
// destination "+v" (so no consumer can be scheduled above it) followed by sched_barrier(0) before the first use.
typedef __attribute__((ext_vector_type(8))) short llx_s16x8_t;
template <int OFF>
__device__ __forceinline__ void lds_tr_read(s16x4_t& dst, uint32_t addr) {
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
// the same with the offset as a (constant after unrolling) function argument
__device__ __forceinline__ void lds_tr_read_rt(s16x4_t& dst, uint32_t addr, int off) {
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off));
}
template <int N>
__device__ __forceinline__ void lds_tr_wait4(s16x4_t& a, s16x4_t& b, s16x4_t& c, s16x4_t& d) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N));
  __builtin_amdgcn_sched_barrier(0);
}
template <int N>
__device__ __forceinline__ void lds_tr_wait8(s16x4_t (&c)[4], s16x4_t (&d)[4]) {
  asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]) : "n"(N));
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ bf16x8_t frag_of(const s16x4_t& lo, const s16x4_t& hi) {
  const llx_s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, v);
}
