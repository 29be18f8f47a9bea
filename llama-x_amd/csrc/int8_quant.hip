// Row-wise absmax int8 quantiser (subclasses/int8.py:10-16), bit-exact with the reference's fp32 arithmetic:
//   scale = absmax(row) / 127 (fp32) ; q = round_half_even(x / max(scale, 1e-12)) -> int8 ; scale stored in the input dtype.
// Used once for weights (quantize_linear_) and every forward for activations when dynamic_int8_act is set.
#include "common.h"

template <typename T> __device__ __forceinline__ float ld_f(const T* p);
template <> __device__ __forceinline__ float ld_f<bf16_t>(const bf16_t* p) { return bf2f(*p); }
template <> __device__ __forceinline__ float ld_f<float>(const float* p) { return *p; }

template <typename T>
__global__ __launch_bounds__(256) void quant_rowwise_kernel(const T* __restrict__ x, int64_t ldx, int8_t* __restrict__ q, int64_t ldq,
                                                            T* __restrict__ scale_out, int cols) {
  __shared__ float red[16];
  const int64_t row = blockIdx.x;
  const T* xr = x + row * ldx;
  float amax = 0.f;
  if constexpr (sizeof(T) == 2) {
    for (int c = threadIdx.x * 8; c < cols; c += 256 * 8) {
      const u32x4_t v = *reinterpret_cast<const u32x4_t*>(xr + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) amax = fmaxf(amax, fmaxf(fabsf(bflo(v[e])), fabsf(bfhi(v[e]))));
    }
  } else {
    for (int c = threadIdx.x; c < cols; c += 256) amax = fmaxf(amax, fabsf(ld_f<T>(xr + c)));
  }
  amax = block_max(amax, red);
  const float scale = amax / 127.0f;
  const float div = fmaxf(scale, 1e-12f);
  if (threadIdx.x == 0) {
    if constexpr (sizeof(T) == 2) scale_out[row] = f2bf(scale);
    else scale_out[row] = scale;
  }
  int8_t* qr = q + row * ldq;
  if constexpr (sizeof(T) == 2) {
    for (int c = threadIdx.x * 8; c < cols; c += 256 * 8) {
      const u32x4_t v = *reinterpret_cast<const u32x4_t*>(xr + c);
      u32x2_t o = {0u, 0u};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int a = (int)rintf(bflo(v[e]) / div), b = (int)rintf(bfhi(v[e]) / div);
        o[e >> 1] |= ((uint32_t)(a & 0xff) | ((uint32_t)(b & 0xff) << 8)) << ((e & 1) * 16);
      }
      *reinterpret_cast<u32x2_t*>(qr + c) = o;
    }
  } else {
    for (int c = threadIdx.x; c < cols; c += 256) qr[c] = (int8_t)(int)rintf(ld_f<T>(xr + c) / div);
  }
}

// x: [rows, cols] bf16 (is_f32 = 0) or fp32 (is_f32 = 1); q: int8 [rows, cols]; scale: [rows] in x's dtype.
extern "C" int llx_quantize_int8_rowwise(const void* x, int64_t ldx, void* q, int64_t ldq, void* scale, int64_t rows, int64_t cols,
                                         int is_f32, hipStream_t stream) {
  LLX_REQUIRE(x && q && scale, "llx_quantize_int8_rowwise: null pointer");
  LLX_REQUIRE(rows >= 0 && cols > 0, "llx_quantize_int8_rowwise: bad sizes");
  if (rows == 0) return LLX_OK;
  if (is_f32) {
    hipLaunchKernelGGL(quant_rowwise_kernel<float>, dim3((unsigned)rows), dim3(256), 0, stream, (const float*)x, ldx, (int8_t*)q, ldq,
                       (float*)scale, (int)cols);
  } else {
    LLX_REQUIRE(cols % 8 == 0 && ldx % 8 == 0 && ldq % 8 == 0 && (uintptr_t)x % 16 == 0 && (uintptr_t)q % 8 == 0,
                "llx_quantize_int8_rowwise: bf16 path needs cols/strides multiples of 8 and aligned pointers");
    hipLaunchKernelGGL(quant_rowwise_kernel<bf16_t>, dim3((unsigned)rows), dim3(256), 0, stream, (const bf16_t*)x, ldx, (int8_t*)q, ldq,
                       (bf16_t*)scale, (int)cols);
  }
  LLX_LAUNCH_CHECK("llx_quantize_int8_rowwise");
  return LLX_OK;
}
