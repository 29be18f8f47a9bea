// Microbenchmark: how fast can ONE CU pull data in, by path?  (gfx950)
//   mode 0: LDS-DMA  (global_load_lds_dwordx4, 1 KiB per wave-instruction) into a 64 KiB LDS ring
//   mode 1: global_load_dwordx4 to registers (consumed by a v_or chain)
//   mode 2: global_load_dwordx4 to registers + ds_write_b128 into LDS
// Every workgroup (8 waves, one per CU when grid = 256) walks a region of `span` bytes starting at (wg % nreg) * span:
//   span = 2 MiB, nreg = 8   -> each XCD's L2 holds its region (L2 hits)
//   span = 64 MiB, nreg = 1  -> one big region, everybody streams it (MALL / HBM)
// Prints bytes per cycle per CU (s_memtime) and GB/s per CU (wall).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

template <int MODE, int DEPTH>
__global__ __launch_bounds__(512, 2) void fill_kernel(const char* src, int64_t span, int nreg, int iters, unsigned long long* cyc, uint32_t* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const char* base = src + (int64_t)(blockIdx.x % nreg) * span;
  // a workgroup-iteration moves 8 waves x DEPTH KiB; consecutive iterations walk the region from a per-workgroup start
  int64_t off = ((int64_t)(blockIdx.x / nreg) * 7919 * 8192) % span;
  u32x4 acc = {0, 0, 0, 0};
  unsigned long long t0;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int it = 0; it < iters; ++it) {
    u32x4 v[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const char* p = base + (off + (int64_t)(d * 8 + wave) * 1024) % span + lane * 16;
      if (MODE == 0) __builtin_amdgcn_global_load_lds((gbl_void*)p, (lds_void*)(smem + ((it & 1) * DEPTH * 8 + d * 8 + wave) * 1024), 16, 0, 0);
      else v[d] = *reinterpret_cast<const u32x4*>(p);
    }
    if (MODE == 0) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH) : "memory");  // the previous iteration's pieces have landed
    } else {
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        if (MODE == 2) *reinterpret_cast<u32x4*>(smem + ((it & 1) * DEPTH * 8 + d * 8 + wave) * 1024 + lane * 16) = v[d];
        else acc |= v[d];
      }
    }
    off = (off + DEPTH * 8192) % span;
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  unsigned long long t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  if (MODE != 0 && (acc[0] | acc[1] | acc[2] | acc[3]) == 0x12345678u) sink[0] = smem[lane];
  if (MODE == 2 && smem[threadIdx.x] == 77 && iters < 0) sink[1] = 1;
}

template <int MODE, int DEPTH>
static void run(const char* name, const char* src, int64_t span, int nreg, int grid) {
  const int iters = 2000;
  unsigned long long* cyc; uint32_t* sink;
  hipMalloc(&cyc, grid * 8); hipMalloc(&sink, 64);
  hipFuncSetAttribute((const void*)fill_kernel<MODE, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * DEPTH * 8 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((fill_kernel<MODE, DEPTH>), dim3(grid), dim3(512), 2 * DEPTH * 8 * 1024, 0, src, span, nreg, iters, cyc, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  unsigned long long* h = (unsigned long long*)malloc(grid * 8);
  hipMemcpy(h, cyc, grid * 8, hipMemcpyDeviceToHost);
  double avg = 0; for (int i = 0; i < grid; ++i) avg += h[i]; avg /= grid;
  const double bytes = (double)iters * DEPTH * 8192;
  printf("%-34s grid %3d depth %d: %6.1f B/cycle/CU  %6.1f GB/s/CU  (%.2f TB/s chip, clock %.2f GHz)\n", name, grid, DEPTH, bytes / avg,
         bytes / (ms * 1e-3) / 1e9, bytes * grid / (ms * 1e-3) / 1e12, avg / (ms * 1e-3) / 1e9);
  free(h); hipFree(cyc); hipFree(sink);
}

int main() {
  char* src; const int64_t total = 64ll << 20;
  hipMalloc(&src, total); hipMemset(src, 1, total);
  for (int grid : {256, 32}) {
    run<0, 4>("LDS-DMA, 2 MiB region per XCD", src, 2 << 20, 8, grid);
    run<0, 8>("LDS-DMA, 2 MiB region per XCD", src, 2 << 20, 8, grid);
    run<0, 4>("LDS-DMA, 64 MiB streamed", src, total, 1, grid);
    run<1, 4>("load->regs, 2 MiB region per XCD", src, 2 << 20, 8, grid);
    run<1, 8>("load->regs, 2 MiB region per XCD", src, 2 << 20, 8, grid);
    run<1, 4>("load->regs, 64 MiB streamed", src, total, 1, grid);
    run<2, 4>("load->regs->ds_write, 2 MiB/XCD", src, 2 << 20, 8, grid);
    run<2, 4>("load->regs->ds_write, 64 MiB", src, total, 1, grid);
  }
  return 0;
}
