// Microbenchmark: issue cadence of fp32 vector instructions on gfx950 - scalar against packed forms, and v_exp_f32 - one wave per SIMD,
// 8 independent chains per instruction kind (s_memtime cycles per instruction).  Also the same streams placed between MFMAs
// (1 MFMA 32x32x16 + n vector instructions per group): how many vector instructions an MFMA's shadow takes for free.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int KIND>
__global__ __launch_bounds__(256) void valu_kernel(int iters, float seed, float* sink, unsigned long long* cyc) {
  float a[8]; f32x2 p[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = seed + i + threadIdx.x; p[i] = f32x2{a[i], a[i] + 0.5f}; }
  const float c1 = seed * 0.999f, c2 = seed * 1e-3f;
  const f32x2 c1p = {c1, c1}, c2p = {c2, c2};
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int it = 0; it < iters; ++it) {
#define FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c1), "v"(c2));
#define PKFMA(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(c1p), "v"(c2p));
#define ADD(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c2));
#define PKADD(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(c2p));
#define MUL(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c1));
#define PKMUL(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(c1p));
#define EXP(i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
#define MAX3(i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c1), "v"(c2));
#define CVT(i) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c1));
    if constexpr (KIND == 0) { REP8(FMA) REP8(FMA) }
    if constexpr (KIND == 1) { REP8(PKFMA) REP8(PKFMA) }
    if constexpr (KIND == 2) { REP8(ADD) REP8(ADD) }
    if constexpr (KIND == 3) { REP8(PKADD) REP8(PKADD) }
    if constexpr (KIND == 4) { REP8(MUL) REP8(MUL) }
    if constexpr (KIND == 5) { REP8(PKMUL) REP8(PKMUL) }
    if constexpr (KIND == 6) { REP8(EXP) REP8(EXP) }
    if constexpr (KIND == 7) { REP8(MAX3) REP8(MAX3) }
    if constexpr (KIND == 8) { REP8(CVT) REP8(CVT) }
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += a[i] + p[i][0] + p[i][1];
  if (s == 1.2345f) sink[0] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

// one 32x32x16 MFMA (32 cycles of matrix pipe) followed by NV vector instructions of one kind: cycles per group
template <int KIND, int NV>
__global__ __launch_bounds__(256) void shadow_kernel(int iters, float seed, float* sink, unsigned long long* cyc) {
  float a[8]; f32x2 p[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = seed + i + threadIdx.x; p[i] = f32x2{a[i], a[i] + 0.5f}; }
  const float c1 = seed * 0.999f, c2 = seed * 1e-3f;
  const f32x2 c1p = {c1, c1}, c2p = {c2, c2};
  f32x16 acc[2];
#pragma unroll
  for (int e = 0; e < 16; ++e) { acc[0][e] = 0.f; acc[1][e] = 0.f; }
  bf16x8 fa, fb;
#pragma unroll
  for (int e = 0; e < 8; ++e) { fa[e] = (__bf16)(seed + e); fb[e] = (__bf16)(seed - e); }
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[g]) : "v"(fa), "v"(fb));
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        if constexpr (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i % 8]) : "v"(c1), "v"(c2));
        if constexpr (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i % 8]) : "v"(c1p), "v"(c2p));
        if constexpr (KIND == 6) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i % 8]));
      }
    }
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += a[i] + p[i][0] + p[i][1];
#pragma unroll
  for (int e = 0; e < 16; ++e) s += acc[0][e] + acc[1][e];
  if (s == 1.2345f) sink[0] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

static float* sink; static unsigned long long* cyc;
template <int KIND> static void run(const char* name) {
  const int iters = 20000;
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((valu_kernel<KIND>), dim3(256), dim3(256), 0, 0, iters, 1.0f, sink, cyc);
  unsigned long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-18s %5.2f cycles per instruction (one wave per SIMD)\n", name, (double)h / (iters * 16.0));
}
template <int KIND, int NV> static void run_shadow(const char* name) {
  const int iters = 20000;
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((shadow_kernel<KIND, NV>), dim3(256), dim3(256), 0, 0, iters, 1.0f, sink, cyc);
  unsigned long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("MFMA 32x32x16 + %2d x %-14s %6.1f cycles per group\n", NV, name, (double)h / (iters * 2.0));
}

int main() {
  hipMalloc(&sink, 64); hipMalloc(&cyc, 64);
  run<0>("v_fma_f32"); run<1>("v_pk_fma_f32"); run<2>("v_add_f32"); run<3>("v_pk_add_f32"); run<4>("v_mul_f32"); run<5>("v_pk_mul_f32");
  run<6>("v_exp_f32"); run<7>("v_max3_f32"); run<8>("v_cvt_pk_bf16_f32");
  run_shadow<0, 0>("(nothing)");
  run_shadow<0, 2>("v_fma_f32"); run_shadow<0, 4>("v_fma_f32"); run_shadow<0, 6>("v_fma_f32"); run_shadow<0, 8>("v_fma_f32");
  run_shadow<1, 2>("v_pk_fma_f32"); run_shadow<1, 4>("v_pk_fma_f32"); run_shadow<1, 6>("v_pk_fma_f32");
  run_shadow<6, 1>("v_exp_f32"); run_shadow<6, 2>("v_exp_f32"); run_shadow<6, 3>("v_exp_f32"); run_shadow<6, 4>("v_exp_f32");
  return 0;
}
