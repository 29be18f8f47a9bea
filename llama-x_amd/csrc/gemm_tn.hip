// bf16 MFMA GEMM, "TN" form:  C[N1,N2] = A[M,N1]^T . B[M,N2]   (fp32 accumulate, bf16 out).
//
// The weight gradient of a dense linear: dW[out,in] = dy[T,out]^T . x[T,in] - reference: autograd of F.linear at
// modelling/llama.py:118-120,140,152,216 for every weight the scripts leave trainable (tok_embeddings / norm / output by default,
// train_metamathqa.py:177-180) and of the Conv1d weights of the audio front end (modelling/audio.py:26-31) as an implicit GEMM.
// Both operands are read as they lie in memory (row = token): no transposed copies.
//
// Tile 256 x 256 x 64 (8 waves, 2 x 4, v_mfma_f32_16x16x32_bf16), two 64-KiB LDS stages filled by global_load_lds.  A K-tile is 64
// token rows of 256 columns = a k-major image [64][256] (512-B rows); the MFMA fragments (8 consecutive k for one column) come out
// of it through ds_read_b64_tr_b16 pairs.  A 32-lane read group touches 8 image rows x 32 B; with 512-B rows they would share one
// 32-B bank window, so the 16-B chunks of a row are XOR-swizzled by key(row) = (row & 3) | ((row >> 3) & 1) << 2 on the SOURCE
// address (the LDS image stays lane-linear for the DMA) and on the read address.
#include "common.h"
#include <mutex>

#define TBM 256
#define TBN 256
#define TBK 64
#define T_TILE_BYTES (TBK * 256 * 2)      // 32 KiB: one operand, 64 k-rows x 512 B
#define T_STAGE_BYTES (2 * T_TILE_BYTES)  // A + B
#define T_EROW 528
#define TN_LDS_BYTES (256 * T_EROW)       // epilogue staging (135168 B) >= 2 stages (131072 B)

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;
typedef __attribute__((address_space(3))) s16x4_t lds_s16x4t;
typedef __attribute__((ext_vector_type(8))) short s16x8t;

struct GemmTnArgs {
  const bf16_t* A; const bf16_t* B; bf16_t* C;
  int64_t lda, ldb, ldc;
  int M, N1, N2;  // contraction length, output rows, output columns
  int grid_m, grid_n;
  const int* m_valid;  // nullable, device: only the first min(M, *m_valid) rows of A and B enter the product (read by the kernel: no host sync)
};

__device__ __forceinline__ int tn_key(int row) { return (row & 3) | (((row >> 3) & 1) << 2); }

// 16x16x32 operand for output index c0 + (lane & 15), k = 8 * (lane >> 4) + j of a 32-row k-step (k-major image, 512-B rows, chunk
// swizzle as above): two ds_read_b64_tr_b16, rows r0 = 8g + q and r0 + 4 (same swizzle key: the second is 2048 bytes further).
// tn_lane_off = byte offset of the first inside the k-step.  The reads are issued as inline asm (common.h: lds_tr_read_rt): through
// the builtin hipcc put `s_waitcnt vmcnt(0)` in front of the first transposed read of every K-tile, which drained the LDS-DMA
// prefetch of the next tile issued a few lines above it - the double buffer ran synchronously.
__device__ __forceinline__ uint32_t tn_lane_off(int c0, int lane) {
  const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
  const int col = c0 + 4 * p;                         // first of the 4 columns this lane's 8 bytes cover
  const int chunk = col >> 3, half = (col & 4) << 1;  // 16-B chunk, byte offset of the 8-B half
  const int r0 = 8 * g + q;
  return (uint32_t)(r0 * 512 + ((chunk ^ (tn_key(r0) << 1)) << 4) + half);
}
template <int N>
__device__ __forceinline__ void tn_wait12(s16x4_t (&x)[12][2]) {
  asm volatile("s_waitcnt lgkmcnt(%24)"
               : "+v"(x[0][0]), "+v"(x[0][1]), "+v"(x[1][0]), "+v"(x[1][1]), "+v"(x[2][0]), "+v"(x[2][1]), "+v"(x[3][0]), "+v"(x[3][1]),
                 "+v"(x[4][0]), "+v"(x[4][1]), "+v"(x[5][0]), "+v"(x[5][1]), "+v"(x[6][0]), "+v"(x[6][1]), "+v"(x[7][0]), "+v"(x[7][1]),
                 "+v"(x[8][0]), "+v"(x[8][1]), "+v"(x[9][0]), "+v"(x[9][1]), "+v"(x[10][0]), "+v"(x[10][1]), "+v"(x[11][0]), "+v"(x[11][1])
               : "n"(N));
  __builtin_amdgcn_sched_barrier(0);
}

__global__ __launch_bounds__(512, 2) void gemm_tn_kernel(const GemmTnArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;

  // block -> tile: XCD-contiguous chunks, then 4-row groups (as the NT kernel)
  const int nwg = g.grid_m * g.grid_n;
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
  }
  const int GROUP_M = 4;
  const int width = GROUP_M * g.grid_n;
  const int group = bid / width;
  const int gsz = min(g.grid_m - group * GROUP_M, GROUP_M);
  const int pid_m = group * GROUP_M + ((bid % width) % gsz);
  const int pid_n = (bid % width) / gsz;
  const int m0 = pid_m * TBM, n0 = pid_n * TBN;  // output row (A column) / output column (B column) origin

  // staging: LDS chunk q = i*512 + tid -> image row (q >> 5), chunk (q & 31); source chunk = chunk ^ (key(row) << 1).
  // The image row of a thread's i-th piece is i*16 + (tid >> 5); key() only uses row bits 0,1,3 -> it depends on i through bit 3
  // of i*16 (none) ... computed per piece below (cheap, outside the loop).
  uint32_t aoff[4], boff[4];
  int srow[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = i * 16 + (tid >> 5);
    const int chunk = (tid & 31) ^ (tn_key(row) << 1);
    srow[i] = row;
    const int ca = min(m0 + chunk * 8, g.N1 - 8);  // clamp: edge columns re-read valid ones, never stored
    const int cb = min(n0 + chunk * 8, g.N2 - 8);
    aoff[i] = (uint32_t)ca * 2;
    boff[i] = (uint32_t)cb * 2;
  }
  const int M = g.m_valid ? min(g.M, max(*g.m_valid, 0)) : g.M;  // workgroup-uniform
  const int nk = (M + TBK - 1) / TBK;
  auto stage = [&](int buf, int kt) {
    char* sA = smem + buf * T_STAGE_BYTES;
    char* sB = sA + T_TILE_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t m = min(kt * TBK + srow[i], M - 1);  // rows past the end are re-reads of the last row; their products are zeroed
      __builtin_amdgcn_global_load_lds((gbl_void*)((const char*)g.A + m * g.lda * 2 + aoff[i]), (lds_void*)(sA + (i * 512 + wave * 64) * 16), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gbl_void*)((const char*)g.B + m * g.ldb * 2 + boff[i]), (lds_void*)(sB + (i * 512 + wave * 64) * 16), 16, 0, 0);
    }
  };

  // lane constants of the fragment reads: 4 B fragments (columns wn*64 + ni*16) and 8 A fragments (wm*128 + mi*16)
  const uint32_t sbase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  uint32_t foff[12];
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) foff[ni] = tn_lane_off(wn * 64 + ni * 16, lane) + T_TILE_BYTES;
#pragma unroll
  for (int mi = 0; mi < 8; ++mi) foff[4 + mi] = tn_lane_off(wm * 128 + mi * 16, lane);

  f32x4_t acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  if (nk > 0) stage(0, 0);  // (no rows: the tile is written as zeros)
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) {
      stage(cur ^ 1, kt + 1);  // its buffer was last read in iteration kt-1, behind that iteration's closing barrier
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // all but the 8 loads just issued: tile kt has landed (this wave's pieces)
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();  // ... and every other wave's pieces
    asm volatile("" ::: "memory");
    const uint32_t stage_base = sbase + cur * T_STAGE_BYTES;
    const bool ragged = kt * TBK + TBK > M;
    // step 0's 24 reads and the 16 A-fragment reads of step 1 go out up front (the latter fly under the 32 MFMAs of step 0); step
    // 1's 8 B-fragment reads reuse step 0's B registers once its MFMAs are issued (one more full set would not fit 256 registers)
    s16x4_t fr[2][12][2];
#pragma unroll
    for (int f = 0; f < 12; ++f) {
      lds_tr_read_rt(fr[0][f][0], stage_base + foff[f], 0);
      lds_tr_read_rt(fr[0][f][1], stage_base + foff[f], 2048);
    }
#pragma unroll
    for (int f = 4; f < 12; ++f) {
      lds_tr_read_rt(fr[1][f][0], stage_base + foff[f], 32 * 512);
      lds_tr_read_rt(fr[1][f][1], stage_base + foff[f], 32 * 512 + 2048);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      if (ks == 0) {
        tn_wait12<15>(fr[0]);  // 40 reads in flight, <= 15 left: the 24 of step 0 are done (lgkmcnt counts in order)
      } else {
#pragma unroll
        for (int f = 0; f < 4; ++f) {
          lds_tr_read_rt(fr[0][f][0], stage_base + foff[f], 32 * 512);
          lds_tr_read_rt(fr[0][f][1], stage_base + foff[f], 32 * 512 + 2048);
        }
#pragma unroll
        for (int f = 0; f < 4; ++f) { fr[1][f][0] = fr[0][f][0]; fr[1][f][1] = fr[0][f][1]; }
        tn_wait12<0>(fr[1]);
      }
      bf16x8_t bfr[4], af[8];
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) bfr[ni] = frag_of(fr[ks][ni][0], fr[ks][ni][1]);
#pragma unroll
      for (int mi = 0; mi < 8; ++mi) af[mi] = frag_of(fr[ks][4 + mi][0], fr[ks][4 + mi][1]);
      if (ragged) {  // last K-tile of a contraction length that is not a multiple of 64: zero the A fragments of rows past M
        const int k0 = kt * TBK + ks * 32 + 8 * (lane >> 4);
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) {
          s16x8t v = __builtin_bit_cast(s16x8t, af[mi]);
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = (k0 + j < M) ? v[j] : (short)0;
          af[mi] = __builtin_bit_cast(bf16x8_t, v);
        }
      }
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int mi = 0; mi < 8; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[ni], af[mi], acc[mi][ni], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // every wave is done reading stage `cur`: the next iteration may refill it
    asm volatile("" ::: "memory");
  }

  // epilogue: C^T fragments (lane owns columns n = fq*4..+4 of row m = frow) -> bf16 -> LDS tile -> 512-B row stores
  const int frow = lane & 15, fq = lane >> 4;
#pragma unroll
  for (int mi = 0; mi < 8; ++mi) {
    const int m = wm * 128 + mi * 16 + frow;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const int n = wn * 64 + ni * 16 + fq * 4;
      u32x2_t pk;
      pk[0] = pack_bf2(acc[mi][ni][0], acc[mi][ni][1]);
      pk[1] = pack_bf2(acc[mi][ni][2], acc[mi][ni][3]);
      *reinterpret_cast<u32x2_t*>(smem + m * T_EROW + n * 2) = pk;
    }
  }
  __syncthreads();
#pragma unroll 4
  for (int it = 0; it < 16; ++it) {
    const int q = it * 512 + tid;
    const int row = q >> 5, cc = q & 31;
    const int gm = m0 + row, gn = n0 + cc * 8;
    if (gm < g.N1 && gn < g.N2)
      *reinterpret_cast<u32x4_t*>(g.C + (int64_t)gm * g.ldc + gn) = *reinterpret_cast<const u32x4_t*>(smem + row * T_EROW + cc * 16);
  }
}

// C[N1,N2] = A[M,N1]^T . B[M,N2], bf16 in / out, fp32 accumulate.  lda / ldb / ldc: row strides in elements (multiples of 8; A and B
// may be row-strided views, e.g. the im2col view of a convolution input).  N1, N2 multiples of 8 and >= 8; any M >= 1.
// m_valid (nullable, device int32): the contraction runs over the first min(M, *m_valid) rows only - the compacted labelled rows of the
// LM head's weight gradient, whose count lives on the device (llx_head_compact_index).
extern "C" int llx_gemm_tn_bf16_rows(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int64_t M, int64_t N1, int64_t N2,
                                     const int32_t* m_valid, hipStream_t stream);
extern "C" int llx_gemm_tn_bf16(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int64_t M, int64_t N1, int64_t N2,
                                hipStream_t stream) {
  return llx_gemm_tn_bf16_rows(A, lda, B, ldb, C, ldc, M, N1, N2, nullptr, stream);
}
extern "C" int llx_gemm_tn_bf16_rows(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int64_t M, int64_t N1, int64_t N2,
                                     const int32_t* m_valid, hipStream_t stream) {
  LLX_REQUIRE(A && B && C, "llx_gemm_tn_bf16: null pointer");
  LLX_REQUIRE(M > 0 && N1 >= 8 && N2 >= 8 && N1 % 8 == 0 && N2 % 8 == 0, "llx_gemm_tn_bf16: need M > 0 and N1, N2 multiples of 8 (M=%lld N1=%lld N2=%lld)",
              (long long)M, (long long)N1, (long long)N2);
  LLX_REQUIRE(lda % 8 == 0 && ldb % 8 == 0 && ldc % 8 == 0, "llx_gemm_tn_bf16: row strides must be multiples of 8 elements");
  LLX_REQUIRE(((uintptr_t)A | (uintptr_t)B | (uintptr_t)C) % 16 == 0, "llx_gemm_tn_bf16: pointers must be 16-byte aligned");
  LLX_REQUIRE(M < (1 << 30) && N1 < (1 << 30) && N2 < (1 << 30), "llx_gemm_tn_bf16: dimension too large");
  static std::once_flag attr_once;  // forward and autograd's backward thread may both be the first caller
  static hipError_t attr_err = hipSuccess;
  std::call_once(attr_once, [] { attr_err = hipFuncSetAttribute((const void*)gemm_tn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, TN_LDS_BYTES); });
  if (attr_err != hipSuccess) {
    llx_set_error("llx_gemm_tn_bf16: cannot raise dynamic LDS limit: %s", hipGetErrorString(attr_err));
    return LLX_ERR_LAUNCH;
  }
  GemmTnArgs a;
  a.A = (const bf16_t*)A; a.B = (const bf16_t*)B; a.C = (bf16_t*)C;
  a.lda = lda; a.ldb = ldb; a.ldc = ldc;
  a.M = (int)M; a.N1 = (int)N1; a.N2 = (int)N2;
  a.grid_m = (int)cdiv64(N1, TBM); a.grid_n = (int)cdiv64(N2, TBN);
  a.m_valid = m_valid;
  hipLaunchKernelGGL(gemm_tn_kernel, dim3(a.grid_m * a.grid_n), dim3(512), TN_LDS_BYTES, stream, a);
  LLX_LAUNCH_CHECK("llx_gemm_tn_bf16");
  return LLX_OK;
}
