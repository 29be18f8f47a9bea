// Microbenchmark: sustained bf16 MFMA rate of the whole chip by instruction shape (gfx950), operands in registers, no memory traffic.
//   shape 0: v_mfma_f32_16x16x32_bf16, a 64 x 64 output patch per wave and iteration = 16 instructions (4 A x 4 B fragments, k = 32)
//   shape 1: v_mfma_f32_32x32x16_bf16, the same patch = 8 instructions (2 A x 2 B fragments x 2 k-steps)
//   shape 2: v_mfma_i32_16x16x64_i8, the patch at k = 64 = 16 instructions (random bytes)
// Both read 32 operand registers per iteration and hold 64 accumulators; the loop runs long enough (~1 s) for the power cap to act.
// Operands are random bf16 bit patterns (zeros would understate the power draw).  Prints PFLOP/s per shape and waves per SIMD, the
// s_memtime cycles one instruction occupies a SIMD, and the clock those cycles imply.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(4))) int i32x4;

template <int SHAPE>
__global__ __launch_bounds__(512) void mfma_kernel(const u32x4* src, int iters, float* sink, unsigned long long* cyc) {
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  u32x4 fr[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) fr[i] = src[(threadIdx.x + i * 512) & 4095];
  if constexpr (SHAPE == 2) {
    i32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = i32x4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, fr[i]), __builtin_bit_cast(i32x4, fr[4 + j]), acc[i][j], 0, 0, 0);
    }
    int s = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (s == 12345) sink[0] = (float)s;
  } else if constexpr (SHAPE == 0) {
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fr[i]), __builtin_bit_cast(bf16x8, fr[4 + j]), acc[i][j], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (s == 1.2345f) sink[0] = s;
  } else {
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fr[2 * ks + i]), __builtin_bit_cast(bf16x8, fr[4 + 2 * ks + j]), acc[i][j], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[i][j][e];
    if (s == 1.2345f) sink[0] = s;
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int SHAPE>
static void run(const u32x4* src, float* sink, int wps, int iters) {
  static unsigned long long* cyc = nullptr;
  if (!cyc) hipMalloc(&cyc, 256 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int threads = 256 * wps, grid = 256;  // wps waves per SIMD on every CU
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((mfma_kernel<SHAPE>), dim3(grid), dim3(threads), 0, 0, src, iters, sink, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  const double flops = 2.0 * 64 * 64 * (SHAPE == 2 ? 64 : 32) * (double)iters * (threads / 64) * grid;
  unsigned long long h[256];
  hipMemcpy(h, cyc, 256 * 8, hipMemcpyDeviceToHost);
  double avg = 0; for (int i = 0; i < 256; ++i) avg += (double)h[i]; avg /= 256;
  const double per_wave = (double)iters * (SHAPE == 1 ? 8 : 16);  // MFMA instructions per wave
  if (wps == 1)
    printf("%s  1 wave/SIMD   %8.1f ms  %6.3f P(FL)OP/s  %5.2f cycles per instruction  clock %.2f GHz\n", SHAPE == 0 ? "bf16 16x16x32" : (SHAPE == 1 ? "bf16 32x32x16" : "i8   16x16x64"),
           ms, flops / (ms * 1e-3) / 1e15, avg / per_wave, avg / (ms * 1e-3) / 1e9);
  else  // (the s_memtime count of a wave that shares its SIMD is not comparable: only the rate is printed)
    printf("%s  %d waves/SIMD  %8.1f ms  %6.3f P(FL)OP/s\n", SHAPE == 0 ? "bf16 16x16x32" : (SHAPE == 1 ? "bf16 32x32x16" : "i8   16x16x64"), wps, ms, flops / (ms * 1e-3) / 1e15);
  fflush(stdout);
}

int main() {
  u32x4* src; float* sink;
  hipMalloc(&src, 4096 * 16); hipMalloc(&sink, 64);
  uint32_t* h = (uint32_t*)malloc(4096 * 16);
  srand(1);
  for (int i = 0; i < 4096 * 4; ++i) {  // two bf16 values in (-2, 2) with random mantissas
    const uint32_t lo = (rand() & 0x80ff) | ((0x3f00 + ((rand() & 1) << 7))), hi = (rand() & 0x80ff) | 0x3f00;
    h[i] = (hi << 16) | lo;
  }
  hipMemcpy(src, h, 4096 * 16, hipMemcpyHostToDevice);
  const int iters = 3000000;  // ~0.4 s at 2 PF/s and one wave per SIMD
  for (int rep = 0; rep < 2; ++rep) {
    run<0>(src, sink, 1, iters);
    run<1>(src, sink, 1, iters);
    run<2>(src, sink, 1, iters);
    run<0>(src, sink, 2, iters);
    run<1>(src, sink, 2, iters);
    run<2>(src, sink, 2, iters);
  }
  return 0;
}
