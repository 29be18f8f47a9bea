// bf16 MFMA GEMM, "NT" form:  C[M,N] = A[M,K] . B[N,K]^T  (+ A2[M,K2] . B2[N,K2]^T)  with fused epilogues.
//
// This is the F.linear shape of the reference (modelling/llama.py:118-120,140,152,216; weight is [out,in]).
// The backward data-gradient uses the same kernel on a transposed copy of the (frozen) weight, and the LoRA
// adapter (modelling/lora.py:43) rides along as the K-extension (A2 = x.A^T, B2 = s.lora_b), so one tuned
// kernel serves every dense contraction of the layer.
//
// Tile 256x256x64, 8 waves (2 along M x 4 along N), v_mfma_f32_16x16x32_bf16, fp32 accumulate.
// HBM -> LDS by global_load_lds (16 B/lane) into two 64-KiB stages; the LDS image is lane-linear, so the
// bank swizzle (16-B slot ^= (row>>1)&7 inside each 128-B row) is applied to the per-lane SOURCE address and
// to the ds_read_b128 address.  The accumulators hold C^T fragments (mfma(B,A)), so each lane owns 4
// consecutive output columns; the epilogue stages the bf16 tile through LDS and stores whole 512-B rows.
#include "common.h"
#include <type_traits>
#include <stdlib.h>
#include <mutex>

#define BM 256
#define BN 256
#define BK 64
#define STAGE_BYTES (2 * 256 * BK * 2)          // A tile + B tile = 64 KiB
#define A_TILE_BYTES (256 * BK * 2)             // 32 KiB
#define EPI_ROW_BYTES 528                       // 512 B of bf16 + 16 B pad (bank spread for the ds_write_b64)
#define GEMM_LDS_BYTES (256 * EPI_ROW_BYTES)    // 135168 >= 2 * STAGE_BYTES
// BNT = 128: the "half tile" (256 rows x 128 columns, waves 4 x 2, 64 x 64 per wave) used for the columns a 256-wide grid would leave
// to a partly empty last round (see gemm_nt_bf16_impl); same LDS stage layout with a half-size B tile.

enum { EPI_NONE = 0, EPI_RESIDUAL = 1, EPI_BIAS = 2, EPI_BIAS_GELU = 3, EPI_COLSCALE = 4, EPI_ROWCOLSCALE = 5, EPI_SWIGLU_BWD = 6, EPI_SWIGLU_FWD = 7, EPI_ROPE = 8, EPI_SPLITK = 9, EPI_ROWCOLSCALE_F32 = 10 };

struct GemmArgs {
  const bf16_t* A; const bf16_t* B; bf16_t* C;
  const bf16_t* A2; const bf16_t* B2;
  const bf16_t* E;   // residual [M,N] (ld = lde) | bias[N] | colscale[N]
  const bf16_t* E2;  // rowscale[M] for EPI_ROWCOLSCALE
  int64_t lda, ldb, ldc, lde, lda2, ldb2;
  int M, N, K, K2;
  int grid_m, grid_n;
  int col0, col_end;  // this launch computes output columns [col0, col_end) (col_end <= N); tile columns are counted from col0
  const bf16_t* sa; const bf16_t* sb;  // int8 kernel: A_scale_rowwise[M], B_scale_colwise[N] (E / lde stay free for the epilogue)
  const float* rope;       // EPI_ROPE: fp32 table [>= rope_S, 64, 2]; row m sits at position m % rope_S
  int rope_S, rope_cols;   //           columns [0, rope_cols) (whole 128-wide heads) are rotated
  const int* m_valid;      // nullable device int32: row tiles that start at or after *m_valid return at once (llx_gemm_nt_bf16_rows)
  float* C32;              // EPI_SPLITK: fp32 partial products [splits][M][N] (row stride N)
  int splits;              //             K is cut in `splits` equal ranges; a tile = (range, row tile, column tile)
};

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// I8 = true: A/B are int8 (K counted in int8 elements, 128 per tile row = the same 128-byte rows), MFMA is
// v_mfma_i32_16x16x64_i8 with int32 accumulators, epilogue EPI_ROWCOLSCALE (torchao::int8_mm_dequant).
template <int EPI, bool I8, int PIPE, int BNT = 256>
__global__ __launch_bounds__(512, 2) void gemm_nt_kernel(const GemmArgs g) {
  static_assert(BNT == 256 || BNT == 128, "tile width");
  static_assert(BNT == 256 || (PIPE == 1 && EPI != EPI_SWIGLU_FWD), "the half tile exists for the four-phase loop only");
  constexpr int ESZ = I8 ? 1 : 2;         // bytes per element
  constexpr int TK = 128 / ESZ;           // elements per 128-byte tile row (64 bf16 | 128 int8)
  constexpr int WNG = BNT / 64;           // wave columns (64 output columns per wave): 4 | 2
  constexpr int WR = 256 / (8 / WNG);     // output rows per wave: 128 | 64
  constexpr int MI = WR / 16;             // 16-row accumulator tiles per wave: 8 | 4
  constexpr int MH = MI / 2;              // ... per phase pair
  constexpr int BSI = BNT / 64;           // 64-row staging pieces of the B tile: 4 | 2
  constexpr int EROW = BNT * 2 + 16;      // epilogue LDS row: the bf16 tile row + 16 B pad
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WNG, wn = wave % WNG;

  // ---- block -> tile: XCD-contiguous chunks (bijective remap), then 4-row groups for L2 panel reuse.
  // row-limited launch (llx_gemm_nt_bf16_rows): the tile space shrinks to the row tiles that hold wanted rows BEFORE the XCD remap, so
  // the surviving tiles stay spread over all eight XCDs (cutting whole row groups off the static map would idle the XCDs that own them)
  int grid_m = g.grid_m;
  if (g.m_valid != nullptr) grid_m = min(grid_m, (*g.m_valid + BM - 1) / BM);
  const int nsplit = EPI == EPI_SPLITK ? g.splits : 1;
  const int nwg = grid_m * g.grid_n * nsplit;
  int bid = blockIdx.x;
  if (bid >= nwg) return;  // workgroup-uniform, before any barrier
  {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
  }
  const int split = EPI == EPI_SPLITK ? bid / (grid_m * g.grid_n) : 0;  // K range of this tile (slowest index)
  if constexpr (EPI == EPI_SPLITK) bid -= split * (grid_m * g.grid_n);
  const int GROUP_M = 4;
  const int width = GROUP_M * g.grid_n;
  const int group = bid / width;
  const int gsz = min(grid_m - group * GROUP_M, GROUP_M);
  const int pid_m = group * GROUP_M + ((bid % width) % gsz);
  const int pid_n = (bid % width) / gsz;
  // EPI_SWIGLU_FWD: B = [W_gate; W_up] (N = 2I rows).  A tile takes 128 gate columns AND the 128 up columns of the same hidden
  // units (B rows n0.. and I + n0..), so the epilogue sees g and u of one h column side by side: grid_n = I / 128.
  constexpr bool SPLITN = EPI == EPI_SWIGLU_FWD;
  const int halfN = g.N >> 1;
  const int m0 = pid_m * BM, n0 = SPLITN ? pid_n * (BN / 2) : g.col0 + pid_n * BNT;
  // output column (= B row) of tile-local column nl
  auto col_of = [&](int nl) { return SPLITN ? (nl < BN / 2 ? n0 + nl : halfN + n0 + nl - BN / 2) : min(n0 + nl, g.col_end - 1); };

  // ---- staging addresses. LDS chunk q = i*512 + tid  -> row i*64 + (tid>>3), slot tid&7;
  // source chunk = slot ^ ((row>>1)&7) = (tid&7) ^ ((tid>>4)&7)   (independent of i).
  const int srow = tid >> 3;
  const int schunk = (tid & 7) ^ ((tid >> 4) & 7);
  int arow[4], brow[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    arow[i] = min(m0 + i * 64 + srow, g.M - 1);  // clamp: edge rows re-read a valid row, never stored
    brow[i] = SPLITN ? (i < 2 ? n0 + i * 64 + srow : halfN + n0 + (i - 2) * 64 + srow) : min(n0 + (i % BSI) * 64 + srow, g.col_end - 1);
  }
  const int nk1 = g.K / TK / nsplit;
  const int nk = nk1 + g.K2 / 64;  // K-extension tiles are bf16: 64 elements per 128-byte row
  const int kofs = split * nk1;    // first K tile of this tile's range (EPI_SPLITK)

  // Per-lane source offsets are loop invariant (row * stride + swizzled chunk, 32-bit bytes); per K-tile only the
  // wave-uniform base pointer advances, so the loads need no vector address arithmetic inside the loop.
  uint32_t aoff[4], boff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    aoff[i] = (uint32_t)((int64_t)arow[i] * g.lda * ESZ + schunk * 16);
    boff[i] = (uint32_t)((int64_t)brow[i] * g.ldb * ESZ + schunk * 16);
  }
  // half = 0: the A tile, half = 1: the B tile (4 x 16-B global_load_lds per thread each)
  auto stage_half = [&](int buf, int kt, int half) {
    char* sT = smem + buf * STAGE_BYTES + half * A_TILE_BYTES;
    if (kt < nk1) {
      const char* base = (const char*)(half ? g.B : g.A) + (int64_t)(kt + kofs) * 128;  // wave-uniform
#pragma unroll
      for (int i = 0; i < 4; ++i)
        __builtin_amdgcn_global_load_lds((gbl_void*)(base + (half ? boff[i] : aoff[i])), (lds_void*)(sT + (i * 512 + wave * 64) * 16), 16, 0, 0);
    } else {  // K-extension tiles (LoRA operands): a different pointer / stride pair, at most a few tiles per launch
      const char* base = (const char*)(half ? g.B2 : g.A2) + (int64_t)(kt - nk1) * 128;
      const int64_t l = (half ? g.ldb2 : g.lda2) * 2;  // the K-extension operands are bf16 in the int8 kernel too
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const char* src = base + (int64_t)(half ? brow[i] : arow[i]) * l + schunk * 16;
        __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)(sT + (i * 512 + wave * 64) * 16), 16, 0, 0);
      }
    }
  };
  auto stage = [&](int buf, int kt) { stage_half(buf, kt, 0); stage_half(buf, kt, 1); };

  // ---- fragment read offsets (bytes inside a tile): row*128 + ((ks*4 + (lane>>4)) ^ ((lane>>1)&7))*16
  const int frow = lane & 15;
  const int fsw = (lane >> 1) & 7;
  const int fq = lane >> 4;
  const int a_base = (wm * WR + frow) * 128;
  const int b_base = (wn * 64 + frow) * 128;
  const int slot0 = ((0 * 4 + fq) ^ fsw) * 16;
  const int slot1 = ((1 * 4 + fq) ^ fsw) * 16;

  using acc_t = typename std::conditional<I8, i32x4_t, f32x4_t>::type;
  acc_t acc[MI][4];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = acc_t{0, 0, 0, 0};

  if constexpr (PIPE == 0) {
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
      const int cur = kt & 1;
      const char* sA = smem + cur * STAGE_BYTES;
      const char* sB = sA + A_TILE_BYTES;
  #pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        if (kt + 1 < nk) stage_half(cur ^ 1, kt + 1, ks);  // spread the next tile's loads over the two k-steps
        const int slot = ks ? slot1 : slot0;
        // 16-byte fragments: 8 bf16 (k = 8*(lane>>4)+j) or 16 int8 (k = 16*(lane>>4)+j) -- same bytes, same addresses
        i32x4_t af[8], bfr[4];
  #pragma unroll
        for (int ni = 0; ni < 4; ++ni) bfr[ni] = *reinterpret_cast<const i32x4_t*>(sB + b_base + ni * 16 * 128 + slot);
  #pragma unroll
        for (int mi = 0; mi < 8; ++mi) af[mi] = *reinterpret_cast<const i32x4_t*>(sA + a_base + mi * 16 * 128 + slot);
        __builtin_amdgcn_s_setprio(1);
  #pragma unroll
        for (int mi = 0; mi < 8; ++mi)
  #pragma unroll
          for (int ni = 0; ni < 4; ++ni) {
            if constexpr (I8)
              acc[mi][ni] = __builtin_amdgcn_mfma_i32_16x16x64_i8(bfr[ni], af[mi], acc[mi][ni], 0, 0, 0);
            else
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, bfr[ni]), __builtin_bit_cast(bf16x8_t, af[mi]),
                                                                    acc[mi][ni], 0, 0, 0);
          }
        __builtin_amdgcn_s_setprio(0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  } else {
    // ---- deep pipeline: 4 phases per K-tile (one 64x32 accumulator quadrant = 16 MFMAs each), one 16-KiB half-tile
    // (A rows 0-127 | A rows 128-255 | B rows 0-127 | B rows 128-255) staged per phase, loads kept in flight ACROSS the
    // barriers behind a counted vmcnt.  Half-tiles of K-tile T are issued:  B-lo, B-hi, A-lo in phases 1,2,3 of tile T-2
    // and A-hi in phase 0 of tile T-1, i.e. 5-7 phases before their first read.  LDS reuse (same 2 x 64 KiB stages):
    //   B(t) is read only in phase 0 (both 32-column halves stay in registers) -> free for B(t+2) after the phase-0 barrier
    //   A(t) is read in phases 0 and 2 -> free for A(t+2) after the phase-2 barrier.
    // pieces (8 KiB = 64 tile rows, one global_load_lds per thread) of a half-tile: A 2, B BSI / 2
    auto stage_q = [&](int kt, int which) {  // which: 0 A-lo, 1 A-hi, 2 B-lo, 3 B-hi of K-tile kt into stage kt&1
      if (kt >= nk) return;
      const int half = which >> 1, hi = which & 1;
      const int np = half ? BSI / 2 : 2;
      char* sT = smem + (kt & 1) * STAGE_BYTES + half * A_TILE_BYTES + hi * np * 8192;
      if (kt < nk1) {
        const char* base = (const char*)(half ? g.B : g.A) + (int64_t)(kt + kofs) * 128;
#pragma unroll
        for (int j = 0; j < 2; ++j)
          if (j < np)
            __builtin_amdgcn_global_load_lds((gbl_void*)(base + (half ? boff[np * hi + j] : aoff[2 * hi + j])), (lds_void*)(sT + (j * 512 + wave * 64) * 16), 16, 0, 0);
      } else {
        const char* base = (const char*)(half ? g.B2 : g.A2) + (int64_t)(kt - nk1) * 128;
        const int64_t l = (half ? g.ldb2 : g.lda2) * 2;  // the K-extension operands are bf16 in the int8 kernel too
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if (j < np) {
            const char* src = base + (int64_t)(half ? brow[np * hi + j] : arow[2 * hi + j]) * l + schunk * 16;
            __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)(sT + (j * 512 + wave * 64) * 16), 16, 0, 0);
          }
        }
      }
    };
    // int8 kernel with a K-extension (LoRA on an int8 base, dynamic activations): after the last int8 K-tile the int32
    // accumulators are dequantised IN PLACE (acc * a_scale[m] * b_scale[n], rounded to bf16 as the reference's int8_mm_dequant
    // output is, kept as fp32 bits) and the bf16 extension tiles accumulate on top with the bf16 MFMA.
    auto mfma_t = [&](auto ext_tag, const i32x4_t& b, const i32x4_t& a, acc_t& c) {
      constexpr bool ext_phase = decltype(ext_tag)::value;
      if constexpr (I8) {
        if constexpr (ext_phase) {
          f32x4_t cf = __builtin_bit_cast(f32x4_t, c);
          cf = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, b), __builtin_bit_cast(bf16x8_t, a), cf, 0, 0, 0);
          c = __builtin_bit_cast(acc_t, cf);
        } else {
          c = __builtin_amdgcn_mfma_i32_16x16x64_i8(b, a, c, 0, 0, 0);
        }
      } else {
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, b), __builtin_bit_cast(bf16x8_t, a), c, 0, 0, 0);
      }
    };
    auto dequant_in_place = [&]() {
      if constexpr (I8) {
        // the 16 column scales and MI row scales of this lane are requested together, once (as written per accumulator element
        // they were 128 two-byte loads per lane behind 40-odd s_waitcnt vmcnt(0))
        float sbv[4][4], rsv[MI];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
          for (int e = 0; e < 4; ++e) sbv[ni][e] = bf2f(g.sb[col_of(wn * 64 + ni * 16 + fq * 4 + e)]);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) rsv[mi] = bf2f(g.sa[min(m0 + wm * WR + mi * 16 + frow, g.M - 1)]);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          const float rs = rsv[mi];
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) {
            f32x4_t cf;
#pragma unroll
            for (int e = 0; e < 4; ++e) cf[e] = bf2f(f2bf(((float)acc[mi][ni][e] * rs) * sbv[ni][e]));
            acc[mi][ni] = __builtin_bit_cast(acc_t, cf);
          }
        }
      }
    };
    // prologue: all of tile 0, then B-lo, B-hi, A-lo of tile 1
    stage_q(0, 0); stage_q(0, 1); stage_q(0, 2); stage_q(0, 3);
    stage_q(1, 2); stage_q(1, 3); stage_q(1, 0);
    auto ktile = [&](int kt, auto ext_tag) __attribute__((always_inline)) {
      auto mfma = [&](const i32x4_t& b, const i32x4_t& a, acc_t& c) { mfma_t(ext_tag, b, a, c); };
      const char* sA = smem + (kt & 1) * STAGE_BYTES;
      const char* sB = sA + A_TILE_BYTES;
      // ---------------- phase 0: tile kt has landed once all but the youngest loads (3 half-tiles of kt+1: A-lo 2 pieces, B-lo and B-hi
      // BSI / 2 each = 6 | 4 loads per thread) are done
      if (kt + 1 < nk) {
        if constexpr (BSI == 4) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      stage_q(kt + 1, 1);  // A-hi of the next tile (its stage's A-hi was last read in phase 2 of tile kt-1)
      i32x4_t bfr[2][4], af[2 * MH];  // bfr[ks][nh*2 + n2], af[ks*MH + m4]
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) bfr[ks][ni] = *reinterpret_cast<const i32x4_t*>(sB + b_base + ni * 16 * 128 + (ks ? slot1 : slot0));
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int m4 = 0; m4 < MH; ++m4) af[ks * MH + m4] = *reinterpret_cast<const i32x4_t*>(sA + a_base + m4 * 16 * 128 + (ks ? slot1 : slot0));
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int m4 = 0; m4 < MH; ++m4)
#pragma unroll
          for (int n2 = 0; n2 < 2; ++n2) mfma(bfr[ks][n2], af[ks * MH + m4], acc[m4][n2]);
      __builtin_amdgcn_s_setprio(0);
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_barrier();  // every wave has its B fragments: B(kt) may be overwritten
      asm volatile("" ::: "memory");
      // ---------------- phase 1: quadrant (first row half, cols 32-63)
      stage_q(kt + 2, 2);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int m4 = 0; m4 < MH; ++m4)
#pragma unroll
          for (int n2 = 0; n2 < 2; ++n2) mfma(bfr[ks][2 + n2], af[ks * MH + m4], acc[m4][2 + n2]);
      __builtin_amdgcn_s_setprio(0);
      // ---------------- phase 2: quadrant (second row half, cols 32-63)
      stage_q(kt + 2, 3);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int m4 = 0; m4 < MH; ++m4) af[ks * MH + m4] = *reinterpret_cast<const i32x4_t*>(sA + a_base + (MH + m4) * 16 * 128 + (ks ? slot1 : slot0));
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int m4 = 0; m4 < MH; ++m4)
#pragma unroll
          for (int n2 = 0; n2 < 2; ++n2) mfma(bfr[ks][2 + n2], af[ks * MH + m4], acc[MH + m4][2 + n2]);
      __builtin_amdgcn_s_setprio(0);
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_barrier();  // every wave has its second A half: A(kt) may be overwritten
      asm volatile("" ::: "memory");
      // ---------------- phase 3: quadrant (second row half, cols 0-31)
      stage_q(kt + 2, 0);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int m4 = 0; m4 < MH; ++m4)
#pragma unroll
          for (int n2 = 0; n2 < 2; ++n2) mfma(bfr[ks][n2], af[ks * MH + m4], acc[MH + m4][n2]);
      __builtin_amdgcn_s_setprio(0);
    };
    const int nk_main = I8 ? nk1 : nk;  // bf16: the K-extension tiles use the same MFMA and simply continue the loop
    for (int kt = 0; kt < nk_main; ++kt) ktile(kt, std::false_type{});
    if constexpr (I8) {
      if (nk > nk1) {
        dequant_in_place();
        for (int kt = nk1; kt < nk; ++kt) ktile(kt, std::true_type{});
      }
    }
    __syncthreads();  // all LDS reads done before the epilogue reuses the stages
  }

  if constexpr (EPI == EPI_SPLITK) {
    // fp32 partial product of this K range, straight from the accumulators (16 B per lane: 4 consecutive columns of one row)
    float* P = g.C32 + (int64_t)split * g.M * g.N;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int gm = m0 + wm * WR + mi * 16 + frow;
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        const int gn = n0 + wn * 64 + ni * 16 + fq * 4;
        if (gm < g.M && gn < g.col_end) *reinterpret_cast<f32x4_t*>(P + (int64_t)gm * g.N + gn) = acc[mi][ni];
      }
    }
    return;
  }
  if constexpr (EPI == EPI_ROWCOLSCALE_F32) {
    // torchao::int8_mm_dequant with fp32 scales (the op returns dtype = A_scale.dtype, subclasses/int8_mm.py:136,143): the fp32 value
    // (acc * a_scale) * b_scale leaves unrounded, 16 B per lane (4 consecutive columns of one row)
    static_assert(I8, "int8 kernel only");
    float* Cf = reinterpret_cast<float*>(g.C);
    const float* saf = reinterpret_cast<const float*>(g.sa);
    const float* sbf = reinterpret_cast<const float*>(g.sb);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int gm = m0 + wm * WR + mi * 16 + frow;
      const float rs = saf[min(gm, g.M - 1)];
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        const int gn = n0 + wn * 64 + ni * 16 + fq * 4;
        if (gm < g.M && gn < g.col_end) {
          const f32x4_t cs = *reinterpret_cast<const f32x4_t*>(sbf + gn);
          f32x4_t c;
#pragma unroll
          for (int e = 0; e < 4; ++e) c[e] = ((float)acc[mi][ni][e] * rs) * cs[e];
          *reinterpret_cast<f32x4_t*>(Cf + (int64_t)gm * g.ldc + gn) = c;
        }
      }
    }
    return;
  }
  // ---- epilogue: acc (C^T fragments: lane owns n = fq*4..+4 for m = frow) -> bf16 -> LDS tile -> coalesced rows.
  // The operands the epilogue needs from global memory (residual, gate|up, bias, scale, RoPE table) come in batches of EB row pieces,
  // requested from a clamped - always valid - address with no branch in between (with the load inside the bounds check every piece was
  // its own basic block: load, s_waitcnt vmcnt(0), store - sixteen exposed latencies per tile).  The FIRST batch is requested here,
  // before the accumulators go through LDS, and batch b+1 while batch b is combined and stored (two register sets: the accumulators
  // are dead by then): one exposed latency per tile instead of one per batch (four with the SwiGLU-backward epilogue).
  constexpr int NIT = BNT / 16;
  constexpr int EB0 = (EPI == EPI_SWIGLU_BWD || EPI == EPI_ROPE) ? 4 : 8;
  constexpr int EB = EB0 < NIT / 2 ? EB0 : NIT / 2;  // at least two batches (the half tile has 8 row pieces per thread)
  constexpr bool AUX = EPI == EPI_RESIDUAL || EPI == EPI_BIAS || EPI == EPI_BIAS_GELU || EPI == EPI_SWIGLU_BWD || EPI == EPI_ROPE || EPI == EPI_COLSCALE;
  auto load_aux = [&](int it0, u32x4_t (&aux0)[EB], u32x4_t (&aux1)[EB]) {
    if constexpr (AUX) {
#pragma unroll
      for (int j = 0; j < EB; ++j) {
        const int q = (it0 + j) * 512 + tid;
        const int row = q / (BNT / 8), cc = q % (BNT / 8);
        const int gmc = min(m0 + row, g.M - 1), gnc = min(n0 + cc * 8, g.col_end - 8);
        if constexpr (EPI == EPI_RESIDUAL) {
          aux0[j] = *reinterpret_cast<const u32x4_t*>(g.E + (int64_t)gmc * g.lde + gnc);
        } else if constexpr (EPI == EPI_SWIGLU_BWD) {
          aux0[j] = *reinterpret_cast<const u32x4_t*>(g.E + (int64_t)gmc * g.lde + gnc);
          aux1[j] = *reinterpret_cast<const u32x4_t*>(g.E + (int64_t)gmc * g.lde + g.N + gnc);
        } else if constexpr (EPI == EPI_ROPE) {
          const float* tp = g.rope + ((int64_t)(gmc % g.rope_S) * 64 + ((gnc & 127) >> 1)) * 2;
          aux0[j] = *reinterpret_cast<const u32x4_t*>(tp);
          aux1[j] = *reinterpret_cast<const u32x4_t*>(tp + 4);
        } else {  // bias / column scale: E[N]
          aux0[j] = *reinterpret_cast<const u32x4_t*>(g.E + gnc);
        }
      }
    }
  };
  u32x4_t auxA0[EB], auxA1[EB], auxB0[EB], auxB1[EB];
  if constexpr (!SPLITN) load_aux(0, auxA0, auxA1);
  float sbv[4][4], rsv[MI];  // int8 without K-extension: this lane's 16 column scales and MI row scales, requested together
  if constexpr (I8) {
    if (g.K2 <= 0) {
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int e = 0; e < 4; ++e) sbv[ni][e] = bf2f(g.sb[col_of(wn * 64 + ni * 16 + fq * 4 + e)]);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) rsv[mi] = bf2f(g.sa[min(m0 + wm * WR + mi * 16 + frow, g.M - 1)]);  // A_scale_rowwise[m]
    }
  }
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int m = wm * WR + mi * 16 + frow;
    float rs = 1.f;
    if constexpr (I8) rs = rsv[mi];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const int n = wn * 64 + ni * 16 + fq * 4;
      float c[4];
      if constexpr (I8) {
        if (g.K2 > 0) {  // dequantised in place before the K-extension: the registers hold fp32 bits
          const f32x4_t cf = __builtin_bit_cast(f32x4_t, acc[mi][ni]);
#pragma unroll
          for (int e = 0; e < 4; ++e) c[e] = cf[e];
        } else {
          // acc.to(fp32) * a_scale * b_scale, one rounding to the scale dtype (subclasses/int8_mm.py:112-118)
#pragma unroll
          for (int e = 0; e < 4; ++e) c[e] = ((float)acc[mi][ni][e] * rs) * sbv[ni][e];
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) c[e] = acc[mi][ni][e];
      }
      u32x2_t pk;
      pk[0] = pack_bf2(c[0], c[1]);
      pk[1] = pack_bf2(c[2], c[3]);
      *reinterpret_cast<u32x2_t*>(smem + m * EROW + n * 2) = pk;
    }
  }
  __syncthreads();
  if constexpr (SPLITN) {
    // tile columns 0-127 = g, 128-255 = u of hidden units n0..n0+127: store both where the unfused layout has them, and
    // h = silu(g) * u into E (used as an OUTPUT here, row stride lde)
    bf16_t* H = const_cast<bf16_t*>(g.E);
#pragma unroll 4
    for (int it = 0; it < 8; ++it) {
      const int q = it * 512 + tid;
      const int row = q >> 4, cc = q & 15;
      const int gm = m0 + row, gn = n0 + cc * 8;
      if (gm < g.M) {
        const u32x4_t gv = *reinterpret_cast<const u32x4_t*>(smem + row * EPI_ROW_BYTES + cc * 16);
        const u32x4_t uv = *reinterpret_cast<const u32x4_t*>(smem + row * EPI_ROW_BYTES + (cc + 16) * 16);
        *reinterpret_cast<u32x4_t*>(g.C + (int64_t)gm * g.ldc + gn) = gv;
        *reinterpret_cast<u32x4_t*>(g.C + (int64_t)gm * g.ldc + halfN + gn) = uv;
        *reinterpret_cast<u32x4_t*>(H + (int64_t)gm * g.lde + gn) = swiglu_fwd8(gv, uv);
      }
    }
    return;
  }
  auto combine = [&](int it0, const u32x4_t (&aux0)[EB], const u32x4_t (&aux1)[EB]) {
#pragma unroll
    for (int j = 0; j < EB; ++j) {
      const int q = (it0 + j) * 512 + tid;
      const int row = q / (BNT / 8), cc = q % (BNT / 8);
      const int gm = m0 + row, gn = n0 + cc * 8;
      if (gm < g.M && gn < g.col_end) {
        u32x4_t v = *reinterpret_cast<const u32x4_t*>(smem + row * EROW + cc * 16);
        if constexpr (EPI == EPI_RESIDUAL) {
          // reference rounding: the linear's bf16 output is added to the bf16 residual (modelling/llama.py:172-173)
          const u32x4_t r = aux0[j];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = pack_bf2(bflo(v[e]) + bflo(r[e]), bfhi(v[e]) + bfhi(r[e]));
        } else if constexpr (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU) {
          const u32x4_t bb = aux0[j];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float lo = bflo(v[e]) + bflo(bb[e]), hi = bfhi(v[e]) + bfhi(bb[e]);
            if constexpr (EPI == EPI_BIAS_GELU) {
              lo = gelu_erf(bf2f(f2bf(lo)));
              hi = gelu_erf(bf2f(f2bf(hi)));
            }
            v[e] = pack_bf2(lo, hi);
          }
        } else if constexpr (EPI == EPI_SWIGLU_BWD) {
          // the product is dh = dL/d(silu(g)*u); what leaves the kernel is dg | du (SwiGLU backward fused: dh never reaches HBM).
          // v holds dh rounded to bf16 exactly as the stand-alone GEMM would have stored it.
          u32x4_t og, ou;
          swiglu_bwd8(v, aux0[j], aux1[j], og, ou);
          *reinterpret_cast<u32x4_t*>(g.C + (int64_t)gm * g.ldc + g.N + gn) = ou;
          v = og;
        } else if constexpr (EPI == EPI_ROPE) {
          // q|k|v projection: apply_rope on the q and k heads right here (modelling/llama.py:118-125); v holds the bf16-rounded
          // projection exactly as the stand-alone GEMM would have stored it, the rotation is the one of rope_kernel
          if (gn < g.rope_cols) v = rope8(v, __builtin_bit_cast(f32x4_t, aux0[j]), __builtin_bit_cast(f32x4_t, aux1[j]), 1.f);
        } else if constexpr (EPI == EPI_COLSCALE) {
          // weight-only int8: (x @ W_i8^T) rounded to bf16, then * scale[n] (subclasses/int8.py:118)
          const u32x4_t sc = aux0[j];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = pack_bf2(bflo(v[e]) * bflo(sc[e]), bfhi(v[e]) * bfhi(sc[e]));
        }
        *reinterpret_cast<u32x4_t*>(g.C + (int64_t)gm * g.ldc + gn) = v;
      }
    }
  };
  static_assert((NIT / EB) % 2 == 0, "the batches alternate between two register sets");
#pragma unroll 1
  for (int it0 = 0; it0 < NIT; it0 += 2 * EB) {
    load_aux(it0 + EB, auxB0, auxB1);
    combine(it0, auxA0, auxA1);
    if (it0 + 2 * EB < NIT) load_aux(it0 + 2 * EB, auxA0, auxA1);
    combine(it0 + EB, auxB0, auxB1);
  }
}

#ifndef LLX_GEMM_PIPE_DEFAULT
#define LLX_GEMM_PIPE_DEFAULT 1
#endif

static int gemm_pipe_mode() {
  static int mode = -1;
  if (mode < 0) {
    const char* e = getenv("LLX_GEMM_PIPE");
    mode = e ? (e[0] == '0' ? 0 : 1) : LLX_GEMM_PIPE_DEFAULT;
  }
  return mode;
}

template <int EPI, bool I8, int PIPE, int BNT = 256>
static int launch_gemm_p(const GemmArgs& a, hipStream_t stream) {
  auto kern = gemm_nt_kernel<EPI, I8, PIPE, BNT>;
  // forward and autograd's backward thread may both be the first caller: the attribute is set exactly once, race-free
  static std::once_flag attr_once;
  static hipError_t attr_err = hipSuccess;
  std::call_once(attr_once, [&] { attr_err = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES); });
  if (attr_err != hipSuccess) {
    llx_set_error("llx_gemm_nt_bf16: cannot raise dynamic LDS limit: %s", hipGetErrorString(attr_err));
    return LLX_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(kern, dim3(a.grid_m * a.grid_n * (EPI == EPI_SPLITK ? a.splits : 1)), dim3(512), GEMM_LDS_BYTES, stream, a);
  LLX_LAUNCH_CHECK("llx_gemm_nt_bf16");
  return LLX_OK;
}

static int gemm_tail_mode() {  // LLX_GEMM_TAIL=0: never split off the half-tile launch (A/B knob)
  static int mode = -1;
  if (mode < 0) {
    const char* e = getenv("LLX_GEMM_TAIL");
    mode = (e && e[0] == '0') ? 0 : 1;
  }
  return mode;
}

// A grid of 256 x 256 tiles whose last round of 256 CUs would be at most half full (q|k|v forward: 384 tiles = 1.5 rounds; w2 data
// gradient: 896 = 3.5) is cut in two launches: the columns that fill whole rounds with full tiles, and the remaining columns with
// 256 x 128 half tiles (twice as many workgroups, each moving 3/4 of a full tile's operand bytes - the per-CU delivery rate, not
// the MFMA pipe, sets a tile's time), so the last round is full and 1/4 shorter.  Results are bit-identical (same K order per output).
template <int EPI, bool I8 = false>
static int launch_gemm(const GemmArgs& a, hipStream_t stream) {
  if (!gemm_pipe_mode()) return launch_gemm_p<EPI, I8, 0>(a, stream);
  if constexpr (EPI != EPI_SWIGLU_FWD) {
    const int tiles = a.grid_m * a.grid_n, tail = tiles % 256;
    if (gemm_tail_mode() && a.col0 == 0 && a.col_end == a.N && a.N % 256 == 0 && tiles > 256 && tail > 0 && tail <= 128 && tail % a.grid_m == 0) {
      const int tail_cols = tail / a.grid_m * 256;
      GemmArgs full = a, half = a;
      full.col_end = a.N - tail_cols;
      full.grid_n = a.grid_n - tail / a.grid_m;
      half.col0 = a.N - tail_cols;
      half.grid_n = tail_cols / 128;
      const int rc = launch_gemm_p<EPI, I8, 1, 256>(full, stream);
      if (rc != LLX_OK) return rc;
      return launch_gemm_p<EPI, I8, 1, 128>(half, stream);
    }
  }
  return launch_gemm_p<EPI, I8, 1>(a, stream);
}

// C[M,N] = A[M,K].B[N,K]^T (+ A2[M,K2].B2[N,K2]^T), bf16 in/out, fp32 accumulate.
// ld* are row strides in elements.  K and K2 must be multiples of 64, N a multiple of 8, all pointers and row
// strides 16-byte aligned.  epilogue: 0 none | 1 +E[M,N] (ld=lde) | 2 +bias E[N] | 3 gelu(+bias) | 4 *colscale E[N] |
// 6 SwiGLU backward (E = gate|up [M,2N], C = dg|du [M,2N]) | 7 SwiGLU forward (B = [W_gate; W_up], C = gate|up [M,N],
// E = OUTPUT h [M,N/2] = silu(g)*u, row stride lde).
static int gemm_nt_bf16_impl(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int64_t M,
                             int64_t N, int64_t K, const void* A2, int64_t lda2, const void* B2, int64_t ldb2, int64_t K2,
                             int epilogue, const void* E, int64_t lde, const float* rope, int64_t rope_S, int64_t rope_cols,
                             hipStream_t stream, const int32_t* m_valid = nullptr) {
  LLX_REQUIRE(A && B && C, "llx_gemm_nt_bf16: null pointer");
  LLX_REQUIRE(M > 0 && N > 0 && K > 0, "llx_gemm_nt_bf16: empty problem M=%lld N=%lld K=%lld", (long long)M, (long long)N, (long long)K);
  LLX_REQUIRE(K % BK == 0 && K2 % BK == 0, "llx_gemm_nt_bf16: K=%lld and K2=%lld must be multiples of 64", (long long)K, (long long)K2);
  LLX_REQUIRE(N % 8 == 0, "llx_gemm_nt_bf16: N=%lld must be a multiple of 8", (long long)N);
  LLX_REQUIRE(lda % 8 == 0 && ldb % 8 == 0 && ldc % 8 == 0, "llx_gemm_nt_bf16: row strides must be multiples of 8 elements");
  LLX_REQUIRE(((uintptr_t)A | (uintptr_t)B | (uintptr_t)C) % 16 == 0, "llx_gemm_nt_bf16: pointers must be 16-byte aligned");
  LLX_REQUIRE(K2 == 0 || (A2 && B2 && lda2 % 8 == 0 && ldb2 % 8 == 0 && ((uintptr_t)A2 | (uintptr_t)B2) % 16 == 0),
              "llx_gemm_nt_bf16: bad K-extension operands");
  LLX_REQUIRE(epilogue == EPI_NONE || epilogue == EPI_ROPE || (E && (uintptr_t)E % 16 == 0), "llx_gemm_nt_bf16: epilogue operand missing/unaligned");
  LLX_REQUIRE(epilogue != EPI_ROPE || (rope && (uintptr_t)rope % 16 == 0 && rope_S > 0 && rope_cols >= 0 && rope_cols <= N && rope_cols % 128 == 0),
              "llx_gemm_nt_bf16_rope: need an aligned fp32 table, rope_S > 0 and rope_cols a multiple of 128 within N");
  LLX_REQUIRE((epilogue != EPI_RESIDUAL && epilogue != EPI_SWIGLU_BWD) || lde % 8 == 0, "llx_gemm_nt_bf16: residual / gate|up stride must be a multiple of 8");
  LLX_REQUIRE(epilogue != EPI_SWIGLU_BWD || (lde >= 2 * N && ldc >= 2 * N), "llx_gemm_nt_bf16: the SwiGLU-backward epilogue reads E[M,2N] (gate|up) and writes C[M,2N] (dg|du)");
  LLX_REQUIRE(M < (1 << 30) && N < (1 << 30) && K < (1 << 30), "llx_gemm_nt_bf16: dimension too large");
  LLX_REQUIRE(M * lda * 2 < (int64_t)4294967296 && N * ldb * 2 < (int64_t)4294967296, "llx_gemm_nt_bf16: operand larger than 4 GiB (32-bit tile offsets)");
  GemmArgs a;
  a.A = (const bf16_t*)A; a.B = (const bf16_t*)B; a.C = (bf16_t*)C;
  a.A2 = (const bf16_t*)A2; a.B2 = (const bf16_t*)B2; a.E = (const bf16_t*)E; a.E2 = nullptr;
  a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.lde = lde; a.lda2 = lda2; a.ldb2 = ldb2;
  a.M = (int)M; a.N = (int)N; a.K = (int)K; a.K2 = (int)K2;
  a.grid_m = (int)cdiv64(M, BM); a.grid_n = (int)cdiv64(N, BN);
  a.col0 = 0; a.col_end = (int)N;
  a.rope = rope; a.rope_S = (int)rope_S; a.rope_cols = (int)rope_cols;
  a.sa = nullptr; a.sb = nullptr;
  a.m_valid = m_valid;
  a.C32 = nullptr; a.splits = 1;
  switch (epilogue) {
    case EPI_NONE: return launch_gemm<EPI_NONE>(a, stream);
    case EPI_RESIDUAL: return launch_gemm<EPI_RESIDUAL>(a, stream);
    case EPI_BIAS: return launch_gemm<EPI_BIAS>(a, stream);
    case EPI_BIAS_GELU: return launch_gemm<EPI_BIAS_GELU>(a, stream);
    case EPI_COLSCALE: return launch_gemm<EPI_COLSCALE>(a, stream);
    case EPI_SWIGLU_BWD: return launch_gemm<EPI_SWIGLU_BWD>(a, stream);
    case EPI_ROPE: return launch_gemm<EPI_ROPE>(a, stream);
    case EPI_SWIGLU_FWD:
      LLX_REQUIRE(N % 256 == 0 && lde % 8 == 0, "llx_gemm_nt_bf16: the SwiGLU-forward epilogue needs N = 2I with I a multiple of 128");
      a.grid_n = (int)(N / 256);
      return launch_gemm<EPI_SWIGLU_FWD>(a, stream);
    default: llx_set_error("llx_gemm_nt_bf16: unknown epilogue %d", epilogue); return LLX_ERR_UNSUPPORTED;
  }
}

extern "C" int llx_gemm_nt_bf16(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int64_t M,
                                int64_t N, int64_t K, const void* A2, int64_t lda2, const void* B2, int64_t ldb2, int64_t K2,
                                int epilogue, const void* E, int64_t lde, hipStream_t stream) {
  LLX_REQUIRE(epilogue != EPI_ROPE, "llx_gemm_nt_bf16: the RoPE epilogue is llx_gemm_nt_bf16_rope");
  return gemm_nt_bf16_impl(A, lda, B, ldb, C, ldc, M, N, K, A2, lda2, B2, ldb2, K2, epilogue, E, lde, nullptr, 0, 0, stream);
}

// llx_gemm_nt_bf16 over the FIRST *m_valid rows only (m_valid: device int32, read by the kernel - the count never visits the host, so
// the launch stays capturable in a hipGraph): row tiles (256 rows) that start at or after *m_valid return at once; rows of C from
// *m_valid up to the end of its tile are computed from whatever A holds there, later rows are left untouched.  Used for the LM head
// over the positions whose label is not ignore_index (modelling/llama.py:216-218: ignored positions contribute neither loss nor gradient).
extern "C" int llx_gemm_nt_bf16_rows(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int64_t M,
                                     int64_t N, int64_t K, const void* A2, int64_t lda2, const void* B2, int64_t ldb2, int64_t K2,
                                     int epilogue, const void* E, int64_t lde, const int32_t* m_valid, hipStream_t stream) {
  LLX_REQUIRE(epilogue != EPI_ROPE, "llx_gemm_nt_bf16_rows: the RoPE epilogue is llx_gemm_nt_bf16_rope");
  LLX_REQUIRE(m_valid && (uintptr_t)m_valid % 4 == 0, "llx_gemm_nt_bf16_rows: m_valid must be a device int32 pointer");
  return gemm_nt_bf16_impl(A, lda, B, ldb, C, ldc, M, N, K, A2, lda2, B2, ldb2, K2, epilogue, E, lde, nullptr, 0, 0, stream, m_valid);
}

// Split-K: partial[s][M][N] (fp32, row stride N) = A[:, K_s] . B[:, K_s]^T for the `splits` equal K ranges, ONE launch whose tile
// space is (range, row tile, column tile): a product with fewer than 256 output tiles but a long contraction (the LM head's
// d hidden = d logits . W: 16 x 16 tiles, K = 128256 - and only 12 x 16 once the unlabelled rows are skipped) fills the chip and
// balances its rounds.  llx_splitk_combine sums the ranges.  m_valid (nullable): as llx_gemm_nt_bf16_rows.  K % (64 * splits) == 0.
extern "C" int llx_gemm_nt_bf16_splitk(const void* A, int64_t lda, const void* B, int64_t ldb, float* partial, int64_t M, int64_t N,
                                       int64_t K, int splits, const int32_t* m_valid, hipStream_t stream) {
  LLX_REQUIRE(A && B && partial, "llx_gemm_nt_bf16_splitk: null pointer");
  LLX_REQUIRE(M > 0 && N > 0 && K > 0 && splits >= 1 && splits <= 16, "llx_gemm_nt_bf16_splitk: bad sizes");
  LLX_REQUIRE(K % ((int64_t)BK * splits) == 0, "llx_gemm_nt_bf16_splitk: K=%lld must be a multiple of 64 * splits (%d)", (long long)K, splits);
  LLX_REQUIRE(N % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0, "llx_gemm_nt_bf16_splitk: N and the row strides must be multiples of 8");
  LLX_REQUIRE(((uintptr_t)A | (uintptr_t)B | (uintptr_t)partial) % 16 == 0, "llx_gemm_nt_bf16_splitk: pointers must be 16-byte aligned");
  LLX_REQUIRE(M < (1 << 30) && N < (1 << 30) && K < (1 << 30), "llx_gemm_nt_bf16_splitk: dimension too large");
  LLX_REQUIRE(M * lda * 2 < (int64_t)4294967296 && N * ldb * 2 < (int64_t)4294967296, "llx_gemm_nt_bf16_splitk: operand larger than 4 GiB (32-bit tile offsets)");
  LLX_REQUIRE(gemm_pipe_mode() == 1, "llx_gemm_nt_bf16_splitk: needs the four-phase main loop (LLX_GEMM_PIPE unset or 1)");
  LLX_REQUIRE(m_valid == nullptr || (uintptr_t)m_valid % 4 == 0, "llx_gemm_nt_bf16_splitk: m_valid must be a device int32 pointer");
  GemmArgs a;
  a.A = (const bf16_t*)A; a.B = (const bf16_t*)B; a.C = nullptr; a.A2 = nullptr; a.B2 = nullptr; a.E = nullptr; a.E2 = nullptr;
  a.lda = lda; a.ldb = ldb; a.ldc = N; a.lde = 0; a.lda2 = 0; a.ldb2 = 0;
  a.M = (int)M; a.N = (int)N; a.K = (int)K; a.K2 = 0;
  a.grid_m = (int)cdiv64(M, BM); a.grid_n = (int)cdiv64(N, BN);
  a.col0 = 0; a.col_end = (int)N;
  a.rope = nullptr; a.rope_S = 0; a.rope_cols = 0;
  a.sa = nullptr; a.sb = nullptr;
  a.m_valid = m_valid;
  a.C32 = partial; a.splits = splits;
  return launch_gemm_p<EPI_SPLITK, false, 1>(a, stream);
}

// out[i] = bf16(scale[0] * bf16(sum_s partial[s][row(i)][:]))  with row(i) = inv ? inv[i] : i; rows with inv[i] < 0 are zero.
// (the product is rounded to bf16 first, as the unsplit GEMM would have stored it, then scaled - llx_scale's order)
__global__ __launch_bounds__(256) void splitk_combine_kernel(const float* __restrict__ partial, int splits, int64_t MN, const int32_t* __restrict__ inv,
                                                             const float* __restrict__ scale, bf16_t* __restrict__ out, int64_t ldo, int N) {
  const int i = blockIdx.x, j = inv ? inv[i] : i;
  const float sc = scale ? scale[0] : 1.f;
  for (int c = threadIdx.x; c < (N >> 2); c += 256) {
    u32x2_t o = {0u, 0u};
    if (j >= 0) {
      f32x4_t v = *reinterpret_cast<const f32x4_t*>(partial + (int64_t)j * N + c * 4);
      for (int s2 = 1; s2 < splits; ++s2) {
        const f32x4_t w = *reinterpret_cast<const f32x4_t*>(partial + s2 * MN + (int64_t)j * N + c * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += w[e];
      }
      if (scale) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = bf2f(f2bf(v[e])) * sc;
      }
      o[0] = pack_bf2(v[0], v[1]);
      o[1] = pack_bf2(v[2], v[3]);
    }
    *reinterpret_cast<u32x2_t*>(out + (int64_t)i * ldo + c * 4) = o;
  }
}

extern "C" int llx_splitk_combine(const float* partial, int splits, int64_t M, int64_t N, const int32_t* inv, const float* scale, void* out,
                                  int64_t ld_out, hipStream_t stream) {
  LLX_REQUIRE(partial && out, "llx_splitk_combine: null pointer");
  LLX_REQUIRE(M > 0 && N > 0 && N % 4 == 0 && ld_out % 4 == 0 && splits >= 1 && splits <= 16, "llx_splitk_combine: bad sizes");
  LLX_REQUIRE(((uintptr_t)partial) % 16 == 0 && ((uintptr_t)out) % 8 == 0, "llx_splitk_combine: unaligned pointer");
  hipLaunchKernelGGL(splitk_combine_kernel, dim3((unsigned)M), dim3(256), 0, stream, partial, splits, M * N, inv, scale, (bf16_t*)out, ld_out, (int)N);
  LLX_LAUNCH_CHECK("llx_splitk_combine");
  return LLX_OK;
}

// The q|k|v projection with apply_rope in the epilogue (modelling/llama.py:118-125): as llx_gemm_nt_bf16 with epilogue 0, then
// columns [0, rope_cols) of C (the q and k heads, 128 wide each) are rotated with the fp32 table [>= rope_S, 64, 2]; row m of C is
// sequence position m % rope_S.  Bit-identical to the plain GEMM followed by llx_rope.
extern "C" int llx_gemm_nt_bf16_rope(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int64_t M, int64_t N,
                                     int64_t K, const void* A2, int64_t lda2, const void* B2, int64_t ldb2, int64_t K2,
                                     const float* rope_table, int64_t rope_S, int64_t rope_cols, hipStream_t stream) {
  return gemm_nt_bf16_impl(A, lda, B, ldb, C, ldc, M, N, K, A2, lda2, B2, ldb2, K2, EPI_ROPE, nullptr, 0, rope_table, rope_S, rope_cols, stream);
}

// torchao::int8_mm_dequant (subclasses/int8_mm.py:121-149): C[M,N] = (A_i8[M,K] . B_i8[N,K]^T)_int32 * a_scale[m] * b_scale[n]
// rounded once to bf16.  B is passed K-contiguous, i.e. the reference's B = int_data.T (strides (1,K)) IS this layout.
// K must be a multiple of 128; scales bf16.
extern "C" int llx_int8_mm_dequant(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int64_t M, int64_t N,
                                   int64_t K, const void* a_scale, const void* b_scale, hipStream_t stream) {
  LLX_REQUIRE(A && B && C && a_scale && b_scale, "llx_int8_mm_dequant: null pointer");
  LLX_REQUIRE(M > 0 && N > 0 && K > 0, "llx_int8_mm_dequant: empty problem");
  LLX_REQUIRE(K % 128 == 0, "llx_int8_mm_dequant: K=%lld must be a multiple of 128", (long long)K);
  LLX_REQUIRE(N % 8 == 0 && ldc % 8 == 0, "llx_int8_mm_dequant: N and ldc must be multiples of 8");
  LLX_REQUIRE(lda % 16 == 0 && ldb % 16 == 0, "llx_int8_mm_dequant: int8 row strides must be multiples of 16");
  LLX_REQUIRE(((uintptr_t)A | (uintptr_t)B | (uintptr_t)C) % 16 == 0, "llx_int8_mm_dequant: pointers must be 16-byte aligned");
  LLX_REQUIRE(M < (1 << 30) && N < (1 << 30) && K < (1 << 30), "%s: dimension too large", "llx_int8_mm_dequant");
  LLX_REQUIRE(M * lda < (int64_t)4294967296 && N * ldb < (int64_t)4294967296, "%s: operand larger than 4 GiB (32-bit tile offsets)", "llx_int8_mm_dequant");
  GemmArgs a;
  a.A = (const bf16_t*)A; a.B = (const bf16_t*)B; a.C = (bf16_t*)C; a.A2 = nullptr; a.B2 = nullptr;
  a.E = nullptr; a.E2 = nullptr; a.sa = (const bf16_t*)a_scale; a.sb = (const bf16_t*)b_scale;
  a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.lde = 0; a.lda2 = 0; a.ldb2 = 0;
  a.M = (int)M; a.N = (int)N; a.K = (int)K; a.K2 = 0;
  a.grid_m = (int)cdiv64(M, BM); a.grid_n = (int)cdiv64(N, BN);
  a.col0 = 0; a.col_end = (int)N;
  a.rope = nullptr; a.rope_S = 0; a.rope_cols = 0;
  a.m_valid = nullptr;
  a.C32 = nullptr; a.splits = 1;
  return launch_gemm<EPI_ROWCOLSCALE, true>(a, stream);
}

// torchao::int8_mm_dequant with fp32 scales and an fp32 result (the reference returns dtype = A_scale.dtype for any float scale,
// subclasses/int8_mm.py:126,136,143; train_librispeech.py:166-170 does not cast the model to bf16).  ldc in fp32 elements.
extern "C" int llx_int8_mm_dequant_f32(const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int64_t ldc, int64_t M, int64_t N,
                                       int64_t K, const float* a_scale, const float* b_scale, hipStream_t stream) {
  LLX_REQUIRE(A && B && C && a_scale && b_scale, "llx_int8_mm_dequant_f32: null pointer");
  LLX_REQUIRE(M > 0 && N > 0 && K > 0, "llx_int8_mm_dequant_f32: empty problem");
  LLX_REQUIRE(K % 128 == 0, "llx_int8_mm_dequant_f32: K=%lld must be a multiple of 128", (long long)K);
  LLX_REQUIRE(N % 8 == 0 && ldc % 4 == 0, "llx_int8_mm_dequant_f32: N must be a multiple of 8 and ldc of 4");
  LLX_REQUIRE(lda % 16 == 0 && ldb % 16 == 0, "llx_int8_mm_dequant_f32: int8 row strides must be multiples of 16");
  LLX_REQUIRE(((uintptr_t)A | (uintptr_t)B | (uintptr_t)C | (uintptr_t)b_scale) % 16 == 0 && (uintptr_t)a_scale % 4 == 0, "llx_int8_mm_dequant_f32: unaligned pointer");
  LLX_REQUIRE(M < (1 << 30) && N < (1 << 30) && K < (1 << 30), "%s: dimension too large", "llx_int8_mm_dequant_f32");
  LLX_REQUIRE(M * lda < (int64_t)4294967296 && N * ldb < (int64_t)4294967296, "%s: operand larger than 4 GiB (32-bit tile offsets)", "llx_int8_mm_dequant_f32");
  GemmArgs a;
  a.A = (const bf16_t*)A; a.B = (const bf16_t*)B; a.C = (bf16_t*)C; a.A2 = nullptr; a.B2 = nullptr;
  a.E = nullptr; a.E2 = nullptr; a.sa = (const bf16_t*)a_scale; a.sb = (const bf16_t*)b_scale;
  a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.lde = 0; a.lda2 = 0; a.ldb2 = 0;
  a.M = (int)M; a.N = (int)N; a.K = (int)K; a.K2 = 0;
  a.grid_m = (int)cdiv64(M, BM); a.grid_n = (int)cdiv64(N, BN);
  a.col0 = 0; a.col_end = (int)N;
  a.rope = nullptr; a.rope_S = 0; a.rope_cols = 0;
  a.m_valid = nullptr;
  a.C32 = nullptr; a.splits = 1;
  return launch_gemm<EPI_ROWCOLSCALE_F32, true>(a, stream);
}

// torchao::int8_mm_dequant with the neighbours of its call sites fused in (an int8 base with dynamically quantised activations,
// subclasses/int8.py:110-118): C = epilogue( bf16( bf16(int8_mm_dequant(A, B, a_scale, b_scale)) + A2[M,K2].B2[N,K2]^T ) ).
//   A2/B2 (nullable, bf16, K2 % 64 == 0): the LoRA adapter (modelling/lora.py:43) - the int32 accumulators are dequantised in place
//   and the extension accumulates on top in fp32;
//   epilogue: 0 none | 1 + E[M,N] (residual, ld = lde) | 7 SwiGLU forward (B = [W_gate; W_up], E = OUTPUT h) | 8 RoPE on columns
//   [0, rope_cols) (as llx_gemm_nt_bf16 / llx_gemm_nt_bf16_rope).
extern "C" int llx_int8_mm_dequant_ext(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int64_t M, int64_t N,
                                       int64_t K, const void* a_scale, const void* b_scale, const void* A2, int64_t lda2, const void* B2,
                                       int64_t ldb2, int64_t K2, int epilogue, const void* E, int64_t lde, const float* rope_table,
                                       int64_t rope_S, int64_t rope_cols, hipStream_t stream) {
  LLX_REQUIRE(A && B && C && a_scale && b_scale, "llx_int8_mm_dequant_ext: null pointer");
  LLX_REQUIRE(M > 0 && N > 0 && K > 0 && K2 >= 0, "llx_int8_mm_dequant_ext: empty problem");
  LLX_REQUIRE(K % 128 == 0 && K2 % 64 == 0, "llx_int8_mm_dequant_ext: K=%lld must be a multiple of 128 and K2=%lld of 64", (long long)K, (long long)K2);
  LLX_REQUIRE(N % 8 == 0 && ldc % 8 == 0, "llx_int8_mm_dequant_ext: N and ldc must be multiples of 8");
  LLX_REQUIRE(lda % 16 == 0 && ldb % 16 == 0, "llx_int8_mm_dequant_ext: int8 row strides must be multiples of 16");
  LLX_REQUIRE(((uintptr_t)A | (uintptr_t)B | (uintptr_t)C) % 16 == 0, "llx_int8_mm_dequant_ext: pointers must be 16-byte aligned");
  LLX_REQUIRE(K2 == 0 || (A2 && B2 && lda2 % 8 == 0 && ldb2 % 8 == 0 && ((uintptr_t)A2 | (uintptr_t)B2) % 16 == 0), "llx_int8_mm_dequant_ext: bad K-extension operands");
  LLX_REQUIRE(epilogue == EPI_NONE || epilogue == EPI_RESIDUAL || epilogue == EPI_SWIGLU_FWD || epilogue == EPI_ROPE, "llx_int8_mm_dequant_ext: epilogue %d unsupported", epilogue);
  LLX_REQUIRE(epilogue == EPI_NONE || epilogue == EPI_ROPE || (E && (uintptr_t)E % 16 == 0 && lde % 8 == 0), "llx_int8_mm_dequant_ext: epilogue operand missing/unaligned");
  LLX_REQUIRE(epilogue != EPI_ROPE || (rope_table && (uintptr_t)rope_table % 16 == 0 && rope_S > 0 && rope_cols >= 0 && rope_cols <= N && rope_cols % 128 == 0),
              "llx_int8_mm_dequant_ext: bad RoPE arguments");
  LLX_REQUIRE(epilogue != EPI_SWIGLU_FWD || N % 256 == 0, "llx_int8_mm_dequant_ext: the SwiGLU epilogue needs N = 2I with I a multiple of 128");
  LLX_REQUIRE(K2 == 0 || gemm_pipe_mode() == 1, "llx_int8_mm_dequant_ext: the K-extension needs the four-phase main loop (LLX_GEMM_PIPE unset or 1)");
  LLX_REQUIRE(M < (1 << 30) && N < (1 << 30) && K < (1 << 30), "%s: dimension too large", "llx_int8_mm_dequant_ext");
  LLX_REQUIRE(M * lda < (int64_t)4294967296 && N * ldb < (int64_t)4294967296, "%s: operand larger than 4 GiB (32-bit tile offsets)", "llx_int8_mm_dequant_ext");
  GemmArgs a;
  a.A = (const bf16_t*)A; a.B = (const bf16_t*)B; a.C = (bf16_t*)C; a.A2 = (const bf16_t*)A2; a.B2 = (const bf16_t*)B2;
  a.E = (const bf16_t*)E; a.E2 = nullptr; a.sa = (const bf16_t*)a_scale; a.sb = (const bf16_t*)b_scale;
  a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.lde = lde; a.lda2 = lda2; a.ldb2 = ldb2;
  a.M = (int)M; a.N = (int)N; a.K = (int)K; a.K2 = (int)K2;
  a.grid_m = (int)cdiv64(M, BM); a.grid_n = (int)cdiv64(N, BN);
  a.col0 = 0; a.col_end = (int)N;
  a.rope = rope_table; a.rope_S = (int)rope_S; a.rope_cols = (int)rope_cols;
  a.m_valid = nullptr;
  a.C32 = nullptr; a.splits = 1;
  switch (epilogue) {
    case EPI_RESIDUAL: return launch_gemm<EPI_RESIDUAL, true>(a, stream);
    case EPI_ROPE: return launch_gemm<EPI_ROPE, true>(a, stream);
    case EPI_SWIGLU_FWD: a.grid_n = (int)(N / 256); return launch_gemm<EPI_SWIGLU_FWD, true>(a, stream);
    default: return launch_gemm<EPI_ROWCOLSCALE, true>(a, stream);
  }
}
