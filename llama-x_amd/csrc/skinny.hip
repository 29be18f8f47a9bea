// Skinny contractions of the LoRA adapter (modelling/lora.py:40-44 and its autograd):
//   skinny_nt :  T[M, 64pad] = X[M,K] . W[R,K]^T            (x @ lora_a^T ; dy @ lora_b with lora_b^T given)
//   skinny_tn :  G[R, N]     = s * U[M, R]^T . Y[M, N]       (d lora_a = s * u^T x ; d lora_b^T = s * t^T dy)
// Both are HBM-bound on the big operand (X or Y, read once); R <= 64 rides on 16x16x32 MFMA tiles.
#include "common.h"
#include <cstdlib>

typedef __attribute__((address_space(3))) s16x4_t lds_s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;

#define SK_PAD 64  // skinny outputs are [M, 64] bf16, columns >= R are zero (they feed the GEMM K-extension)

// ------------------------------------------------------------------------------------------ NT
// block = 8 waves = 16 rows of X; the waves interleave over k-steps of 32 (8 loads of 1 KiB in flight per wave, so a
// CU keeps 64 KiB of HBM reads outstanding with one block resident) and combine their partial tiles through LDS.
#define SNT_WAVES 8
#define SNT_UNROLL 8

// Block-diagonal W (the batched LoRA B^T of a fused linear group: rows 16*nb.. only meet k in [lo[nb], hi[nb])): the k-steps
// outside a row block's range are skipped - no W fragment load, no MFMA.  Dense W: lo = 0, hi = K.
struct SkinnyRanges { int lo[4], hi[4]; };

// SCALE: the kernel also writes G[m, k] = bf16(X[m, k] * cs[k]) (row stride ldg) from the fragments it has just loaded - the
// (grad_output * scale) operand of a weight-only int8 linear's data gradient (subclasses/int8.py:127) rides in the adapter's dy @ B
// pass instead of costing its own read of dy.
__device__ __forceinline__ void snt_scaled_store(const bf16x8_t& a, const bf16_t* cs, bf16_t* g) {
  const u32x4_t x = __builtin_bit_cast(u32x4_t, a);
  const u32x4_t c = *reinterpret_cast<const u32x4_t*>(cs);
  u32x4_t o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = pack_bf2(bflo(x[e]) * bflo(c[e]), bfhi(x[e]) * bfhi(c[e]));
  *reinterpret_cast<u32x4_t*>(g) = o;  // (a non-temporal store here was measured slower: +0.5 ms on the int8 step)
}

template <int NB, bool RANGED, bool SCALE = false>
__global__ __launch_bounds__(SNT_WAVES * 64) void skinny_nt_kernel(const bf16_t* __restrict__ X, int64_t ldx, const bf16_t* __restrict__ W,
                                                                   int64_t ldw, bf16_t* __restrict__ out, int M, int K, int R, SkinnyRanges kr,
                                                                   const bf16_t* __restrict__ cs = nullptr, bf16_t* __restrict__ G = nullptr,
                                                                   int64_t ldg = 0) {
  // loads in flight per wave and round: 8 k-steps; 4 with four row blocks (8 spilled in the block-diagonal form: 256 registers + scratch;
  // the dense form follows so that both sum in the same order - the ranged product is bit-identical to the dense one)
  constexpr int SNT_UNROLL_ = NB == 4 ? 4 : SNT_UNROLL;
  __shared__ float part[SNT_WAVES][16][SK_PAD];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform by construction: the k-range tests below become scalar branches
  const int m0 = blockIdx.x * 16;
  const int fr = lane & 15, fq = lane >> 4;
  const bf16_t* xrow = X + (int64_t)min(m0 + fr, M - 1) * ldx;
  const bf16_t* wrow[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) wrow[nb] = W + (int64_t)min(nb * 16 + fr, R - 1) * ldw;
  f32x4_t acc[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) acc[nb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  // The sum over k does not care which 32 k-values form one MFMA step as long as X and W use the same map.  Main loop: a
  // PAIR of steps covers 64 consecutive k; lane (fr, fq) takes k = 64*pair + 16*fq + 8*h + 0..7 for step h, i.e. 32
  // contiguous bytes per lane and whole 128-byte lines per row and wave (with the natural map, k = 32*step + 8*fq, a wave
  // touches half of every line it reads and the L2 -> L1 traffic of the W fragments doubles).
  const int npairs = K >> 6;
  const int rounds = npairs / ((SNT_UNROLL_ / 2) * SNT_WAVES);  // the same trip count for every wave
  // every block walks the same W; starting each block at its own round spreads the simultaneous W reads of an XCD's 32
  // blocks over the L2 channels instead of queueing them on the same lines (that queue was 55 % of the K = 28672 call)
  const int rot = rounds > 0 ? (int)((blockIdx.x * 7u + (blockIdx.x >> 3)) % (unsigned)rounds) : 0;
  for (int it0 = 0; it0 < rounds; ++it0) {
    const int it = (it0 + rot < rounds) ? it0 + rot : it0 + rot - rounds;
    const int pr = it * (SNT_UNROLL_ / 2) * SNT_WAVES + wave;
    bf16x8_t a[SNT_UNROLL_], b[SNT_UNROLL_][NB];
#pragma unroll
    for (int u = 0; u < SNT_UNROLL_; ++u) a[u] = *reinterpret_cast<const bf16x8_t*>(xrow + (pr + (u >> 1) * SNT_WAVES) * 64 + 16 * fq + 8 * (u & 1));
    if constexpr (SCALE) {
      if (m0 + fr < M) {
#pragma unroll
        for (int u = 0; u < SNT_UNROLL_; ++u) {
          const int k = (pr + (u >> 1) * SNT_WAVES) * 64 + 16 * fq + 8 * (u & 1);
          snt_scaled_store(a[u], cs + k, G + (int64_t)(m0 + fr) * ldg + k);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < SNT_UNROLL_; ++u) {
      const int k0 = (pr + (u >> 1) * SNT_WAVES) * 64;  // wave-uniform
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
        if (!RANGED || (k0 >= kr.lo[nb] && k0 < kr.hi[nb])) b[u][nb] = *reinterpret_cast<const bf16x8_t*>(wrow[nb] + k0 + 16 * fq + 8 * (u & 1));
    }
#pragma unroll
    for (int u = 0; u < SNT_UNROLL_; ++u) {
      const int k0 = (pr + (u >> 1) * SNT_WAVES) * 64;
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
        if (!RANGED || (k0 >= kr.lo[nb] && k0 < kr.hi[nb])) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[u], b[u][nb], acc[nb], 0, 0, 0);
    }
  }
  // remaining k-steps (natural map), 32 k-values each
  const int nks = K >> 5;
  {
    const int pr_done = rounds * (SNT_UNROLL_ / 2) * SNT_WAVES;  // pairs [0, pr_done) are finished
    for (int ks = 2 * pr_done + wave; ks < nks; ks += SNT_WAVES) {
      const bf16x8_t a = *reinterpret_cast<const bf16x8_t*>(xrow + ks * 32 + 8 * fq);
      if constexpr (SCALE) {
        if (m0 + fr < M) snt_scaled_store(a, cs + ks * 32 + 8 * fq, G + (int64_t)(m0 + fr) * ldg + ks * 32 + 8 * fq);
      }
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        if (RANGED && (ks * 32 < kr.lo[nb] || ks * 32 >= kr.hi[nb])) continue;
        const bf16x8_t b = *reinterpret_cast<const bf16x8_t*>(wrow[nb] + ks * 32 + 8 * fq);
        acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[nb], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int e = 0; e < 4; ++e) part[wave][fq * 4 + e][nb * 16 + fr] = acc[nb][e];
  __syncthreads();
  // 1024 outputs / 512 threads: thread t -> row t>>5, cols (t&31)*2..+2
  const int row = threadIdx.x >> 5, c0 = (threadIdx.x & 31) * 2;
  if (m0 + row < M) {
    float v[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int c = c0 + e;
      float s = 0.f;
      if (c < NB * 16 && c < R) {
#pragma unroll
        for (int w = 0; w < SNT_WAVES; ++w) s += part[w][row][c];
      }
      v[e] = s;
    }
    *reinterpret_cast<uint32_t*>(out + (int64_t)(m0 + row) * SK_PAD + c0) = pack_bf2(v[0], v[1]);
  }
}

// out: [M, 64] bf16 (row stride 64). K % 32 == 0, R <= 64.
// kranges (host pointer, nullable): {lo_0, hi_0, ..., lo_3, hi_3}, multiples of 64: rows 16*nb..16*nb+15 of W are zero outside
// k in [lo_nb, hi_nb) and those k-steps are skipped (block-diagonal W); null = dense W.
extern "C" int llx_skinny_nt(const void* X, int64_t ldx, const void* W, int64_t ldw, void* out, int64_t M, int64_t K, int64_t R,
                             const int32_t* kranges, hipStream_t stream) {
  LLX_REQUIRE(X && W && out, "llx_skinny_nt: null pointer");
  LLX_REQUIRE(M > 0 && K > 0 && K % 32 == 0 && R > 0 && R <= 64, "llx_skinny_nt: need K%%32==0 and 0<R<=64 (K=%lld R=%lld)", (long long)K, (long long)R);
  LLX_REQUIRE(ldx % 8 == 0 && ldw % 8 == 0 && ((uintptr_t)X | (uintptr_t)W) % 16 == 0 && (uintptr_t)out % 8 == 0, "llx_skinny_nt: alignment");
  const dim3 grid((unsigned)cdiv64(M, 16)), block(SNT_WAVES * 64);
  const int nb = (int)cdiv64(R, 16);
  SkinnyRanges kr;
  for (int i = 0; i < 4; ++i) {
    kr.lo[i] = kranges ? kranges[2 * i] : 0;
    kr.hi[i] = kranges ? kranges[2 * i + 1] : (int)K;
    LLX_REQUIRE(kr.lo[i] % 64 == 0 && (kr.hi[i] % 64 == 0 || kr.hi[i] == (int)K) && kr.lo[i] >= 0 && kr.hi[i] <= (int)K,
                "llx_skinny_nt: k range %d of a block-diagonal W must be a multiple of 64 inside [0, K]", i);
  }
#define L(N, RG) hipLaunchKernelGGL((skinny_nt_kernel<N, RG>), grid, block, 0, stream, (const bf16_t*)X, ldx, (const bf16_t*)W, ldw, (bf16_t*)out, (int)M, (int)K, (int)R, kr)
  if (kranges) { if (nb == 1) L(1, true); else if (nb == 2) L(2, true); else L(4, true); }
  else { if (nb == 1) L(1, false); else if (nb == 2) L(2, false); else L(4, false); }
#undef L
  LLX_LAUNCH_CHECK("llx_skinny_nt");
  return LLX_OK;
}

// llx_skinny_nt that also writes G[M, K] = bf16(X * colscale[k]) (row stride ldg) from the same read of X.
extern "C" int llx_skinny_nt_scaled(const void* X, int64_t ldx, const void* W, int64_t ldw, void* out, int64_t M, int64_t K, int64_t R,
                                    const int32_t* kranges, const void* colscale, void* G, int64_t ldg, hipStream_t stream) {
  LLX_REQUIRE(X && W && out && colscale && G, "llx_skinny_nt_scaled: null pointer");
  LLX_REQUIRE(M > 0 && K > 0 && K % 32 == 0 && R > 0 && R <= 64, "llx_skinny_nt_scaled: need K%%32==0 and 0<R<=64 (K=%lld R=%lld)", (long long)K, (long long)R);
  LLX_REQUIRE(ldx % 8 == 0 && ldw % 8 == 0 && ldg % 8 == 0 && ((uintptr_t)X | (uintptr_t)W | (uintptr_t)colscale | (uintptr_t)G) % 16 == 0 && (uintptr_t)out % 8 == 0,
              "llx_skinny_nt_scaled: alignment");
  const dim3 grid((unsigned)cdiv64(M, 16)), block(SNT_WAVES * 64);
  const int nb = (int)cdiv64(R, 16);
  SkinnyRanges kr;
  for (int i = 0; i < 4; ++i) {
    kr.lo[i] = kranges ? kranges[2 * i] : 0;
    kr.hi[i] = kranges ? kranges[2 * i + 1] : (int)K;
    LLX_REQUIRE(kr.lo[i] % 64 == 0 && (kr.hi[i] % 64 == 0 || kr.hi[i] == (int)K) && kr.lo[i] >= 0 && kr.hi[i] <= (int)K,
                "llx_skinny_nt_scaled: k range %d of a block-diagonal W must be a multiple of 64 inside [0, K]", i);
  }
#define L(N, RG) hipLaunchKernelGGL((skinny_nt_kernel<N, RG, true>), grid, block, 0, stream, (const bf16_t*)X, ldx, (const bf16_t*)W, ldw, (bf16_t*)out, (int)M, (int)K, (int)R, kr, (const bf16_t*)colscale, (bf16_t*)G, ldg)
  if (kranges) { if (nb == 1) L(1, true); else if (nb == 2) L(2, true); else L(4, true); }
  else { if (nb == 1) L(1, false); else if (nb == 2) L(2, false); else L(4, false); }
#undef L
  LLX_LAUNCH_CHECK("llx_skinny_nt_scaled");
  return LLX_OK;
}

// ------------------------------------------------------------------------------------------ TN
#define TN_NT 256                 // columns of Y per block
#define TN_MS 32                  // rows per step
#define TN_YROW (TN_NT * 2 + 16)  // padded LDS row strides (bytes)
#define TN_UROW (SK_PAD * 2 + 16)

// 16x16x32 operand (k = 8*(lane>>4)+j, index = lane&15) read transposed from a [32 rows][cols] bf16 LDS tile:
// element j = tile[8*(lane>>4) + j][c0 + (lane&15)].
__device__ __forceinline__ bf16x8_t tr16_frag(const char* tile, int rowbytes, int c0, int lane) {
  const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
  const char* base = tile + (8 * g + q) * rowbytes + (c0 + 4 * p) * 2;
  s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)base);
  s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + 4 * rowbytes));
  s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, v);
}

// Member segments of a fused linear group's dB^T = dy^T.t: member i owns output rows n in [n_lo, n_hi) and columns r in
// [r_lo, r_hi) (block-diagonal; everything else of the [N, R] product is never read).  With segments the main kernel skips the
// 16-row blocks rb whose members do not meet the column tile, and the reduce writes each member's block as its own contiguous
// [n_hi - n_lo, r_hi - r_lo] matrix at out + off (offsets in member order), so no slicing copy follows.
struct TnSegs { int n_lo[4], n_hi[4], r_lo[4], r_hi[4]; int64_t off[4]; int count; int rb_lo[4], rb_hi[4]; };

// One block of the first stage: column tile `ctile` (256 columns of Y), row range `split`.  NB = 16-row blocks of U^T the registers
// are sized for; nbl <= NB = the product's own count (layout of its partial array: [split][nbl*16][N]).
// Bt / upart (nullable): the block ALSO emits partial sums of  Y . Bt^T  (the adapter's  u = dy @ B  - Bt [R rows, N] row-major, the
// batched block-diagonal B^T of the group) from the Y tile it already has in LDS: upart[ctile][16 rb + j][m] = sum over the tile's 256
// columns, row block rb handled by wave rb and only where the tile meets the block's column range (the same rb_lo / rb_hi test as the main
// product).  llx_skinny_u_reduce sums the tiles.  Y is then read ONCE for dB and u instead of once by this kernel and once by skinny_nt.
template <int NB>
__device__ __forceinline__ void skinny_tn_block(const bf16_t* __restrict__ U, const bf16_t* __restrict__ Y, int64_t ldy, float* __restrict__ partial,
                                                int M, int N, int rows_per_split, const TnSegs& sg, int ctile, int split, int nbl, char* sY, char* sU,
                                                const bf16_t* __restrict__ Bt = nullptr, int64_t ldb = 0, float* __restrict__ upart = nullptr, int brows = 0,
                                                const bf16_t* __restrict__ cs = nullptr, bf16_t* __restrict__ G = nullptr, int64_t ldg = 0) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n0 = ctile * TN_NT;
  const int m_begin = split * rows_per_split, m_end = min(M, m_begin + rows_per_split);
  // u partials: wave w owns row block w of Bt
  const bool u_on = upart != nullptr && wave < nbl && n0 + TN_NT > sg.rb_lo[wave] && n0 < sg.rb_hi[wave];  // wave-uniform
  bf16x8_t ub[8];
  if (u_on) {
    const bf16_t* brow = Bt + (int64_t)min(wave * 16 + (lane & 15), brows - 1) * ldb;  // rows past R: their columns of u are zeroed by the reduce
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) ub[ks] = *reinterpret_cast<const bf16x8_t*>(brow + min(n0 + ks * 32 + 8 * (lane >> 4), N - 8));  // columns past N meet zeros of the Y tile
  }
  f32x4_t acc[NB][4];
#pragma unroll
  for (int rb = 0; rb < NB; ++rb)
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) acc[rb][cb] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // global -> registers (zero rows past the end / columns past N), TWO steps ahead: two register sets alternate, the set of step i
  // is refilled with step i+2 as soon as it has been written to LDS, so two steps of loads (2 x 20 KB per block) fly under the
  // transposed reads and MFMAs of the current one - one step in flight left the kernel latency-bound at ~3.5 TB/s
  u32x4_t yv[2][4], uv[2];
  auto load_step = [&](int set, int ms) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = i * 256 + tid;  // chunk of 16 B: row q>>5, chunk q&31
      const int row = ms + (q >> 5), col = n0 + (q & 31) * 8;
      yv[set][i] = (row < m_end && col < N) ? *reinterpret_cast<const u32x4_t*>(Y + (int64_t)row * ldy + col) : u32x4_t{0u, 0u, 0u, 0u};
    }
    const int row = ms + (tid >> 3);
    uv[set] = (row < m_end) ? *reinterpret_cast<const u32x4_t*>(U + (int64_t)row * SK_PAD + (tid & 7) * 8) : u32x4_t{0u, 0u, 0u, 0u};
  };
  auto step = [&](int set, int ms) {
    __syncthreads();  // previous step's reads are done
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = i * 256 + tid;
      *reinterpret_cast<u32x4_t*>(sY + (q >> 5) * TN_YROW + (q & 31) * 16) = yv[set][i];
      if (G) {  // the (grad_output * scale) operand of a weight-only / dynamic int8 linear's data gradient (subclasses/int8.py:127) from
                // the same read of dy: every element of Y passes through exactly one block of the first stage
        const int row = ms + (q >> 5), col = n0 + (q & 31) * 8;
        if (row < m_end && col < N) {
          const u32x4_t c = *reinterpret_cast<const u32x4_t*>(cs + col);
          u32x4_t o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = pack_bf2(bflo(yv[set][i][e]) * bflo(c[e]), bfhi(yv[set][i][e]) * bfhi(c[e]));
          *reinterpret_cast<u32x4_t*>(G + (int64_t)row * ldg + col) = o;
        }
      }
    }
    *reinterpret_cast<u32x4_t*>(sU + (tid >> 3) * TN_UROW + (tid & 7) * 16) = uv[set];
    __syncthreads();
    if (ms + 2 * TN_MS < m_end) load_step(set, ms + 2 * TN_MS);
    bf16x8_t a[NB];
#pragma unroll
    for (int rb = 0; rb < NB; ++rb) a[rb] = tr16_frag(sU, TN_UROW, rb * 16, lane);
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
      const bf16x8_t b = tr16_frag(sY, TN_YROW, wave * 64 + cb * 16, lane);
#pragma unroll
      for (int rb = 0; rb < NB; ++rb)
        if (n0 + TN_NT > sg.rb_lo[rb] && n0 < sg.rb_hi[rb]) acc[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[rb], b, acc[rb][cb], 0, 0, 0);
    }
    if (u_on) {  // u partial of this step's 32 rows over the tile's 256 columns: Y rows straight from the row-major LDS tile.
      // Computed TRANSPOSED (rows of Bt on the accumulator rows, Y rows on the lanes) and stored r-major, upart[tile][r][m]: a store
      // instruction then writes 64-byte runs of consecutive m and the two row fragments of a step complete whole 128-byte lines
      // (m-major, 16 of the 64 columns of a row = a quarter line per row, the partial writes doubled the kernel's time).
#pragma unroll
      for (int rf = 0; rf < 2; ++rf) {
        f32x4_t ua0 = {0.f, 0.f, 0.f, 0.f}, ua1 = {0.f, 0.f, 0.f, 0.f};  // two chains: a single one is 8 dependent MFMAs
        const char* yrow = sY + (rf * 16 + (lane & 15)) * TN_YROW + (lane >> 4) * 16;
#pragma unroll
        for (int ks = 0; ks < 8; ks += 2) {
          ua0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ub[ks], *reinterpret_cast<const bf16x8_t*>(yrow + ks * 64), ua0, 0, 0, 0);
          ua1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ub[ks + 1], *reinterpret_cast<const bf16x8_t*>(yrow + (ks + 1) * 64), ua1, 0, 0, 0);
        }
        const int row = ms + rf * 16 + (lane & 15);
        if (row < m_end) {
#pragma unroll
          for (int e = 0; e < 4; ++e) upart[((int64_t)ctile * SK_PAD + wave * 16 + (lane >> 4) * 4 + e) * M + row] = ua0[e] + ua1[e];
        }
      }
    }
  };
  if (m_begin < m_end) load_step(0, m_begin);
  if (m_begin + TN_MS < m_end) load_step(1, m_begin + TN_MS);
  for (int ms = m_begin; ms < m_end; ms += 2 * TN_MS) {
    step(0, ms);
    if (ms + TN_MS < m_end) step(1, ms + TN_MS);
  }
  // D[row = r][col = n]: lane -> n = lane&15, r = (lane>>4)*4 + e
  const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
  for (int rb = 0; rb < NB; ++rb)
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
      const int n = n0 + wave * 64 + cb * 16 + fr;
      if (n < N && n0 + TN_NT > sg.rb_lo[rb] && n0 < sg.rb_hi[rb]) {
#pragma unroll
        for (int e = 0; e < 4; ++e) partial[((int64_t)split * (nbl * 16) + rb * 16 + fq * 4 + e) * N + n] = acc[rb][cb][e];
      }
    }
}

template <int NB>
__global__ __launch_bounds__(256) void skinny_tn_kernel(const bf16_t* __restrict__ U, const bf16_t* __restrict__ Y, int64_t ldy,
                                                        float* __restrict__ partial, int M, int N, int rows_per_split, TnSegs sg) {
  __shared__ __attribute__((aligned(16))) char sY[TN_MS * TN_YROW];
  __shared__ __attribute__((aligned(16))) char sU[TN_MS * TN_UROW];
  skinny_tn_block<NB>(U, Y, ldy, partial, M, N, rows_per_split, sg, blockIdx.x, blockIdx.y, NB, sY, sU);
}

// First stage of up to TN_MANY products in ONE launch (the dB and dA products of a linear group read different operands but are ready
// together: one ramp-up and one tail instead of two, and the small product's blocks fill the gaps of the big one).  blockIdx.x walks
// the products' blocks back to back; row blocks past a product's own count are switched off through its rb_lo / rb_hi table.
#define TN_MANY 4
struct TnPart { const bf16_t* U; const bf16_t* Y; int64_t ldy; float* partial; int M, N, rows_per_split, ctiles, nblocks, nbl; TnSegs sg;
                const bf16_t* Bt; int64_t ldb; float* upart; int brows; const bf16_t* cs; bf16_t* G; int64_t ldg; };
struct TnPartMany { TnPart d[TN_MANY]; int n; };

// NB = the largest number of 16-row blocks among the products (accumulator registers: 16 per block; with the u partials' operand
// fragments on top, four blocks would cost the third resident workgroup per CU)
template <int NB>
__global__ __launch_bounds__(256) void skinny_tn_many_kernel(const TnPartMany m) {
  __shared__ __attribute__((aligned(16))) char sY[TN_MS * TN_YROW];
  __shared__ __attribute__((aligned(16))) char sU[TN_MS * TN_UROW];
  int b = blockIdx.x, i = 0;
  while (i < m.n - 1 && b >= m.d[i].nblocks) { b -= m.d[i].nblocks; ++i; }
  const TnPart& d = m.d[i];
  skinny_tn_block<NB>(d.U, d.Y, d.ldy, d.partial, d.M, d.N, d.rows_per_split, d.sg, b % d.ctiles, b / d.ctiles, d.nbl, sY, sU, d.Bt, d.ldb, d.upart, d.brows, d.cs, d.G, d.ldg);
}

// (second stage: skinny_tn_reduce_many_kernel below - out = bf16(scale * sum_split partial[split][r][n] (+ out)), plain [R,N] / [N,R] or member segments)

static int tn_splits(int64_t M, int64_t N) {
  const int64_t ntiles = cdiv64(N, TN_NT);
  static const int target = getenv("LLX_TN_BLOCKS") ? atoi(getenv("LLX_TN_BLOCKS")) : 384;
  int64_t want = cdiv64(target, ntiles);        // blocks per launch: 384 measured best over 128..1024 on the step's shapes (fewer, longer blocks and a smaller reduce)
  const int64_t max_split = cdiv64(M, 4 * TN_MS);  // at least 4 steps per block
  if (want > max_split) want = max_split;
  if (want < 1) want = 1;
  return (int)want;
}

extern "C" int64_t llx_skinny_tn_workspace_bytes(int64_t M, int64_t N, int64_t R) {
  return (int64_t)tn_splits(M, N) * cdiv64(R, 16) * 16 * N * 4;
}

// segment table of a fused group's dB^T (host side): validates and fills sg; returns the number of output elements.
static int tn_fill_segs(TnSegs& sg, const int32_t* segs, int seg_count, int64_t N, int64_t R, int64_t* total_out) {
  sg.count = 0;
  for (int rb = 0; rb < 4; ++rb) { sg.rb_lo[rb] = 0; sg.rb_hi[rb] = (int)N; }
  int64_t total = 0;
  if (segs) {
    LLX_REQUIRE(seg_count >= 1 && seg_count <= 4, "llx_skinny_tn: 1..4 segments");
    for (int rb = 0; rb < 4; ++rb) { sg.rb_lo[rb] = (int)N; sg.rb_hi[rb] = 0; }
    for (int i = 0; i < seg_count; ++i) {
      sg.n_lo[i] = segs[4 * i]; sg.n_hi[i] = segs[4 * i + 1]; sg.r_lo[i] = segs[4 * i + 2]; sg.r_hi[i] = segs[4 * i + 3];
      LLX_REQUIRE(sg.n_lo[i] >= 0 && sg.n_lo[i] < sg.n_hi[i] && sg.n_hi[i] <= N && sg.r_lo[i] >= 0 && sg.r_lo[i] < sg.r_hi[i] && sg.r_hi[i] <= R,
                  "llx_skinny_tn: bad segment %d", i);
      LLX_REQUIRE(sg.n_lo[i] % TN_NT == 0 && (sg.n_hi[i] % TN_NT == 0 || sg.n_hi[i] == N), "llx_skinny_tn: segment %d: n bounds must be multiples of 256", i);
      sg.off[i] = total;
      total += (int64_t)(sg.n_hi[i] - sg.n_lo[i]) * (sg.r_hi[i] - sg.r_lo[i]);
      for (int rb = sg.r_lo[i] / 16; rb <= (sg.r_hi[i] - 1) / 16; ++rb) {
        if (sg.n_lo[i] < sg.rb_lo[rb]) sg.rb_lo[rb] = sg.n_lo[i];
        if (sg.n_hi[i] > sg.rb_hi[rb]) sg.rb_hi[rb] = sg.n_hi[i];
      }
    }
    sg.count = seg_count;
  }
  *total_out = total;
  return LLX_OK;
}

// first stage only: fp32 split partials of U^T.Y into the workspace (llx_skinny_tn_workspace_bytes); llx_skinny_tn_reduce_many finishes.
extern "C" int llx_skinny_tn_partial(const void* U, const void* Y, int64_t ldy, int64_t M, int64_t N, int64_t R, void* workspace,
                                     const int32_t* segs, int seg_count, hipStream_t stream) {
  LLX_REQUIRE(U && Y && workspace, "llx_skinny_tn: null pointer");
  LLX_REQUIRE(M > 0 && N > 0 && N % 8 == 0 && R > 0 && R <= 64 && ldy % 8 == 0, "llx_skinny_tn: need N%%8==0, ldy%%8==0, 0<R<=64");
  LLX_REQUIRE(((uintptr_t)U | (uintptr_t)Y) % 16 == 0, "llx_skinny_tn: unaligned pointer");
  const int nsplit = tn_splits(M, N);
  int rows_per_split = (int)cdiv64(cdiv64(M, nsplit), TN_MS) * TN_MS;
  const int nb = (int)cdiv64(R, 16);
  TnSegs sg;
  int64_t total = 0;
  const int rc = tn_fill_segs(sg, segs, seg_count, N, R, &total);
  if (rc != LLX_OK) return rc;
  const dim3 grid((unsigned)cdiv64(N, TN_NT), (unsigned)nsplit), block(256);
#define L(NBV) hipLaunchKernelGGL(skinny_tn_kernel<NBV>, grid, block, 0, stream, (const bf16_t*)U, (const bf16_t*)Y, ldy, (float*)workspace, (int)M, (int)N, rows_per_split, sg)
  if (nb == 1) L(1); else if (nb == 2) L(2); else if (nb == 3) L(3); else L(4);
#undef L
  LLX_LAUNCH_CHECK("llx_skinny_tn");
  return LLX_OK;
}

// Arrays of length n (<= 4), one entry per product, arguments as llx_skinny_tn_partial.
extern "C" int llx_skinny_tn_partial_many_u(int n, const void* const* U, const void* const* Y, const int64_t* ldy, const int64_t* M, const int64_t* N,
                                            const int64_t* R, void* const* workspaces, const int32_t* const* segs, const int* seg_count,
                                            const void* const* Bt, const int64_t* ldb, void* const* upart, hipStream_t stream);
extern "C" int llx_skinny_tn_partial_many(int n, const void* const* U, const void* const* Y, const int64_t* ldy, const int64_t* M, const int64_t* N,
                                          const int64_t* R, void* const* workspaces, const int32_t* const* segs, const int* seg_count,
                                          hipStream_t stream) {
  return llx_skinny_tn_partial_many_u(n, U, Y, ldy, M, N, R, workspaces, segs, seg_count, nullptr, nullptr, nullptr, stream);
}
// ... where product i with upart[i] != null ALSO emits the column-tile partials of  Y_i . Bt_i^T  (Bt_i [R_i, N_i] bf16 row-major
// with row stride ldb[i]; upart[i]: llx_skinny_u_workspace_bytes(M_i, N_i) bytes of fp32 [N_i / 256 tiles][64][M_i]); llx_skinny_u_reduce sums them.
extern "C" int64_t llx_skinny_u_workspace_bytes(int64_t M, int64_t N) { return cdiv64(N, TN_NT) * M * SK_PAD * 4; }
extern "C" int llx_skinny_tn_partial_many_us(int n, const void* const* U, const void* const* Y, const int64_t* ldy, const int64_t* M, const int64_t* N,
                                             const int64_t* R, void* const* workspaces, const int32_t* const* segs, const int* seg_count,
                                             const void* const* Bt, const int64_t* ldb, void* const* upart, const void* const* colscale,
                                             void* const* G, const int64_t* ldg, hipStream_t stream);
extern "C" int llx_skinny_tn_partial_many_u(int n, const void* const* U, const void* const* Y, const int64_t* ldy, const int64_t* M, const int64_t* N,
                                            const int64_t* R, void* const* workspaces, const int32_t* const* segs, const int* seg_count,
                                            const void* const* Bt, const int64_t* ldb, void* const* upart, hipStream_t stream) {
  return llx_skinny_tn_partial_many_us(n, U, Y, ldy, M, N, R, workspaces, segs, seg_count, Bt, ldb, upart, nullptr, nullptr, nullptr, stream);
}
// ... and product i with G[i] != null also writes G_i[m, n] = bf16(Y_i[m, n] * colscale_i[n]) (row stride ldg[i]; colscale bf16 [N_i]): the
// (grad_output * scale) operand of an int8 linear's data gradient (subclasses/int8.py:127) from the same read of dy.
extern "C" int llx_skinny_tn_partial_many_us(int n, const void* const* U, const void* const* Y, const int64_t* ldy, const int64_t* M, const int64_t* N,
                                             const int64_t* R, void* const* workspaces, const int32_t* const* segs, const int* seg_count,
                                             const void* const* Bt, const int64_t* ldb, void* const* upart, const void* const* colscale,
                                             void* const* G, const int64_t* ldg, hipStream_t stream) {
  LLX_REQUIRE(n >= 1 && n <= TN_MANY && U && Y && ldy && M && N && R && workspaces && segs && seg_count, "llx_skinny_tn_partial_many: bad arguments (1..4 products)");
  TnPartMany m;
  m.n = n;
  int total_blocks = 0;
  for (int i = 0; i < n; ++i) {
    TnPart& d = m.d[i];
    LLX_REQUIRE(U[i] && Y[i] && workspaces[i], "llx_skinny_tn_partial_many: null pointer (product %d)", i);
    LLX_REQUIRE(M[i] > 0 && N[i] > 0 && N[i] % 8 == 0 && R[i] > 0 && R[i] <= 64 && ldy[i] % 8 == 0, "llx_skinny_tn_partial_many: need N%%8==0, ldy%%8==0, 0<R<=64 (product %d)", i);
    LLX_REQUIRE(((uintptr_t)U[i] | (uintptr_t)Y[i]) % 16 == 0, "llx_skinny_tn_partial_many: unaligned pointer (product %d)", i);
    const int nsplit = tn_splits(M[i], N[i]);
    d.U = (const bf16_t*)U[i]; d.Y = (const bf16_t*)Y[i]; d.ldy = ldy[i]; d.partial = (float*)workspaces[i];
    d.M = (int)M[i]; d.N = (int)N[i];
    d.rows_per_split = (int)cdiv64(cdiv64(M[i], nsplit), TN_MS) * TN_MS;
    d.ctiles = (int)cdiv64(N[i], TN_NT);
    d.nblocks = d.ctiles * nsplit;
    d.nbl = (int)cdiv64(R[i], 16);
    int64_t total = 0;
    const int rc = tn_fill_segs(d.sg, segs[i], seg_count[i], N[i], R[i], &total);
    if (rc != LLX_OK) return rc;
    for (int rb = d.nbl; rb < 4; ++rb) { d.sg.rb_lo[rb] = d.N; d.sg.rb_hi[rb] = 0; }  // row blocks this product does not have
    d.Bt = nullptr; d.ldb = 0; d.upart = nullptr; d.brows = (int)R[i];
    if (upart && upart[i]) {
      LLX_REQUIRE(Bt && ldb && Bt[i] && ldb[i] % 8 == 0 && (uintptr_t)Bt[i] % 16 == 0 && N[i] >= 8, "llx_skinny_tn_partial_many_u: Bt of product %d", i);
      d.Bt = (const bf16_t*)Bt[i]; d.ldb = ldb[i]; d.upart = (float*)upart[i];
    }
    d.cs = nullptr; d.G = nullptr; d.ldg = 0;
    if (G && G[i]) {
      LLX_REQUIRE(colscale && ldg && colscale[i] && ldg[i] % 8 == 0 && ((uintptr_t)colscale[i] | (uintptr_t)G[i]) % 16 == 0, "llx_skinny_tn_partial_many_us: scaled copy of product %d", i);
      d.cs = (const bf16_t*)colscale[i]; d.G = (bf16_t*)G[i]; d.ldg = ldg[i];
    }
    total_blocks += d.nblocks;
  }
  int nbmax = 1;
  for (int i = 0; i < n; ++i) nbmax = m.d[i].nbl > nbmax ? m.d[i].nbl : nbmax;
  const dim3 grid((unsigned)total_blocks), block(256);
  if (nbmax == 1) hipLaunchKernelGGL(skinny_tn_many_kernel<1>, grid, block, 0, stream, m);
  else if (nbmax == 2) hipLaunchKernelGGL(skinny_tn_many_kernel<2>, grid, block, 0, stream, m);
  else if (nbmax == 3) hipLaunchKernelGGL(skinny_tn_many_kernel<3>, grid, block, 0, stream, m);
  else hipLaunchKernelGGL(skinny_tn_many_kernel<4>, grid, block, 0, stream, m);
  LLX_LAUNCH_CHECK("llx_skinny_tn_partial_many");
  return LLX_OK;
}

// u[m][c] = bf16(sum over the column tiles that meet row block c / 16 of upart[tile][m][c]) for c < R, 0 up to column 64 (the K-extension
// operand of the data-gradient GEMM and the U operand of the dA product): the second half of the fused dB + u pass.
__global__ __launch_bounds__(256) void skinny_u_reduce_kernel(const float* __restrict__ upart, bf16_t* __restrict__ out, int M, int R, TnSegs sg) {
  // block = 64 consecutive rows m x 4 columns, ONE output per thread: thread -> (c = 4 blockIdx.y + (t >> 6), m = 64 blockIdx.x + (t & 63))
  // reads runs of consecutive m (the partials are r-major); the four columns cross LDS so that u leaves as 8-byte row pieces
  __shared__ float tile[4][65];
  const int m0 = blockIdx.x * 64, ml = threadIdx.x & 63, cl = threadIdx.x >> 6;
  const int c = blockIdx.y * 4 + cl;
  float sum = 0.f;
  if (c < R) {
    const int rb = c >> 4;
    const int t_lo = sg.rb_lo[rb] / TN_NT, t_hi = (sg.rb_hi[rb] + TN_NT - 1) / TN_NT;
    const int64_t stride = (int64_t)SK_PAD * M;
    const float* src = upart + (int64_t)c * M + min(m0 + ml, M - 1);
    int t = t_lo;
    for (; t + 8 <= t_hi; t += 8) {  // eight loads in flight, added in tile order
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = src[(t + j) * stride];
#pragma unroll
      for (int j = 0; j < 8; ++j) sum += v[j];
    }
    for (; t < t_hi; ++t) sum += src[t * stride];
  }
  tile[cl][ml] = sum;
  __syncthreads();
  if (threadIdx.x < 64 && m0 + threadIdx.x < M) {
    u32x2_t pk;
    pk[0] = pack_bf2(tile[0][threadIdx.x], tile[1][threadIdx.x]);
    pk[1] = pack_bf2(tile[2][threadIdx.x], tile[3][threadIdx.x]);
    *reinterpret_cast<u32x2_t*>(out + (int64_t)(m0 + threadIdx.x) * SK_PAD + blockIdx.y * 4) = pk;
  }
}

// out: [M, 64] bf16.  segs / seg_count as given to the product that filled upart (they define which tiles each 16-column block met).
extern "C" int llx_skinny_u_reduce(const void* upart, void* out, int64_t M, int64_t N, int64_t R, const int32_t* segs, int seg_count, hipStream_t stream) {
  LLX_REQUIRE(upart && out && M > 0 && N > 0 && R > 0 && R <= 64, "llx_skinny_u_reduce: bad arguments");
  TnSegs sg;
  int64_t total = 0;
  const int rc = tn_fill_segs(sg, segs, seg_count, N, R, &total);
  if (rc != LLX_OK) return rc;
  hipLaunchKernelGGL(skinny_u_reduce_kernel, dim3((unsigned)cdiv64(M, 64), SK_PAD / 4), dim3(256), 0, stream, (const float*)upart, (bf16_t*)out, (int)M, (int)R, sg);
  LLX_LAUNCH_CHECK("llx_skinny_u_reduce");
  return LLX_OK;
}

// second stage of up to TN_MANY products in ONE launch (the four adapter gradients of a transformer block): blockIdx.y picks the product.
struct TnRed { const float* partial; bf16_t* out; int64_t out_ld, total; int nsplit, RP, R, N, transpose_out, accumulate, use_segs; float scale; TnSegs sg; };
struct TnRedMany { TnRed d[TN_MANY]; };

__global__ void skinny_tn_reduce_many_kernel(const TnRedMany m) {
  const TnRed& d = m.d[blockIdx.y];
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= d.total) return;
  int r, n;
  bf16_t* dst;
  if (d.use_segs) {
    int s = 0;
    for (; s < d.sg.count; ++s) {
      const int64_t cnt = (int64_t)(d.sg.n_hi[s] - d.sg.n_lo[s]) * (d.sg.r_hi[s] - d.sg.r_lo[s]);
      if (idx < cnt) break;
      idx -= cnt;
    }
    if (s >= d.sg.count) return;
    const int rw = d.sg.r_hi[s] - d.sg.r_lo[s];
    n = d.sg.n_lo[s] + (int)(idx / rw);
    r = d.sg.r_lo[s] + (int)(idx % rw);
    dst = d.out + d.sg.off[s] + idx;
  } else {
    r = (int)(idx / d.N);
    n = (int)(idx % d.N);
    dst = d.transpose_out ? d.out + (int64_t)n * d.out_ld + r : d.out + (int64_t)r * d.out_ld + n;
  }
  // the partials of one output are nsplit loads RP * N floats apart: eight are requested together, then added in split order (the
  // same sum as a plain loop - which the compiler turns into load, wait, add per split: 24 exposed memory latencies per thread)
  float sum = 0.f;
  const float* src = d.partial + (int64_t)r * d.N + n;
  const int64_t stride = (int64_t)d.RP * d.N;
  int p = 0;
  for (; p + 8 <= d.nsplit; p += 8) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = src[(p + j) * stride];
#pragma unroll
    for (int j = 0; j < 8; ++j) sum += v[j];
  }
  if (p + 4 <= d.nsplit) {
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = src[(p + j) * stride];
#pragma unroll
    for (int j = 0; j < 4; ++j) sum += v[j];
    p += 4;
  }
  for (; p < d.nsplit; ++p) sum += src[p * stride];
  sum *= d.scale;
  if (d.accumulate) sum += bf2f(*dst);
  *dst = f2bf(sum);
}

// Arrays of length n (<= 4), one entry per product: the workspace its llx_skinny_tn_partial filled, the output and the arguments of
// llx_skinny_tn (M, N, R give the split count).  segs[i] (nullable host pointer) / seg_count[i] as in llx_skinny_tn.
extern "C" int llx_skinny_tn_reduce_many(int n, const void* const* workspaces, void* const* outs, const int64_t* out_ld, const int64_t* M,
                                         const int64_t* N, const int64_t* R, const float* scale, const int* transpose_out, const int* accumulate,
                                         const int32_t* const* segs, const int* seg_count, hipStream_t stream) {
  LLX_REQUIRE(n >= 1 && n <= TN_MANY && workspaces && outs && out_ld && M && N && R && scale && transpose_out && accumulate && segs && seg_count,
              "llx_skinny_tn_reduce_many: bad arguments (1..4 products)");
  TnRedMany m;
  int64_t most = 0;
  for (int i = 0; i < n; ++i) {
    TnRed& d = m.d[i];
    LLX_REQUIRE(workspaces[i] && outs[i] && M[i] > 0 && N[i] > 0 && R[i] > 0 && R[i] <= 64, "llx_skinny_tn_reduce_many: bad product %d", i);
    d.partial = (const float*)workspaces[i]; d.out = (bf16_t*)outs[i]; d.out_ld = out_ld[i];
    d.nsplit = tn_splits(M[i], N[i]); d.RP = (int)cdiv64(R[i], 16) * 16; d.R = (int)R[i]; d.N = (int)N[i];
    d.transpose_out = transpose_out[i]; d.accumulate = accumulate[i]; d.scale = scale[i];
    d.use_segs = segs[i] != nullptr;
    int64_t total = 0;
    const int rc = tn_fill_segs(d.sg, segs[i], seg_count[i], N[i], R[i], &total);
    if (rc != LLX_OK) return rc;
    d.total = d.use_segs ? total : R[i] * N[i];
    if (d.total > most) most = d.total;
  }
  hipLaunchKernelGGL(skinny_tn_reduce_many_kernel, dim3((unsigned)cdiv64(most, 256), (unsigned)n), dim3(256), 0, stream, m);
  LLX_LAUNCH_CHECK("llx_skinny_tn_reduce_many");
  return LLX_OK;
}

// U: [M, 64] bf16 (row stride 64; columns >= R ignored), Y: [M, N] (row stride ldy).  workspace as above.
// segs (host pointer, nullable): up to 4 members {n_lo, n_hi, r_lo, r_hi} (n bounds multiples of 256 or N), seg_count of them:
// only those blocks of the [N, R] product are computed, and member i's block is written transposed-out as a contiguous
// [n_hi - n_lo, r_hi - r_lo] matrix at out + sum_{j<i} size_j (out_ld and transpose_out are ignored).
extern "C" int llx_skinny_tn(const void* U, const void* Y, int64_t ldy, void* out, int64_t out_ld, int64_t M, int64_t N, int64_t R,
                             float scale, int transpose_out, int accumulate, void* workspace, const int32_t* segs, int seg_count,
                             hipStream_t stream) {
  LLX_REQUIRE(out, "llx_skinny_tn: null pointer");
  int rc = llx_skinny_tn_partial(U, Y, ldy, M, N, R, workspace, segs, seg_count, stream);
  if (rc != LLX_OK) return rc;
  const void* ws[1] = {workspace};
  void* outs[1] = {out};
  return llx_skinny_tn_reduce_many(1, ws, outs, &out_ld, &M, &N, &R, &scale, &transpose_out, &accumulate, &segs, &seg_count, stream);
}

// ------------------------------------------------------------------------------------------ RMSNorm + NT in one pass
// y = rmsnorm(x) (nn.RMSNorm, modelling/llama.py:158-160: fp32 math, one rounding) AND t = y . W^T [M, 64 padded] (the adapter's
// x @ lora_a^T of modelling/lora.py:43 on the NORMED activations) from ONE read of x: the rows are already on chip when the norm is
// applied, in exactly the fragment layout the skinny MFMA product wants.  Block = 16 rows, 8 waves; wave w holds the 16 x (D/8)
// slice k in [w*D/8, (w+1)*D/8) of the rows in registers (pair-contiguous k map of skinny_nt_kernel: whole 128-B lines per row).
// Same math as llx_rmsnorm_fwd followed by llx_skinny_nt(y, W) with different fp32 summation orders (the squares of a row are summed
// per fragment lane, the product over contiguous k slices per wave): rstd agrees to ~1e-7, y and t to one bf16 ulp in rare elements;
// deterministic run to run, and the backward uses the rstd saved here.
#define RSN_MAXPAIRS 8  // D / 8 waves / 64 = pairs per wave: D <= 4096

template <int NB>
__global__ __launch_bounds__(SNT_WAVES * 64) void rmsnorm_skinny_nt_kernel(const bf16_t* __restrict__ X, const bf16_t* __restrict__ G,
                                                                           const bf16_t* __restrict__ W, int64_t ldw, bf16_t* __restrict__ Y,
                                                                           float* __restrict__ rstd_out, bf16_t* __restrict__ T, int M, int D,
                                                                           int R, float eps) {
  __shared__ float part[SNT_WAVES][16][SK_PAD];
  __shared__ float ssq[SNT_WAVES][16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m0 = blockIdx.x * 16;
  const int fr = lane & 15, fq = lane >> 4;
  const int row = min(m0 + fr, M - 1);
  const int npw = D / (SNT_WAVES * 64);  // pairs (64 k each) per wave, <= RSN_MAXPAIRS
  const bf16_t* xrow = X + (int64_t)row * D + wave * npw * 64 + 16 * fq;
  bf16x8_t a[RSN_MAXPAIRS][2];
  float ss = 0.f;
#pragma unroll
  for (int p = 0; p < RSN_MAXPAIRS; ++p) {
    if (p < npw) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        a[p][h] = *reinterpret_cast<const bf16x8_t*>(xrow + p * 64 + 8 * h);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float v = (float)a[p][h][j];
          ss += v * v;
        }
      }
    }
  }
  // the norm weight and the adapter fragments of the first RSN_PRE pairs do not depend on the row statistics: their loads go out
  // BEFORE the cross-wave reduction (behind its barrier they started a second memory round trip after the first had drained)
  const bf16_t* grow = G + wave * npw * 64 + 16 * fq;
  const bf16_t* wrow[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) wrow[nb] = W + (int64_t)min(nb * 16 + fr, R - 1) * ldw + wave * npw * 64 + 16 * fq;
  bf16x8_t gv[RSN_MAXPAIRS][2];
#pragma unroll
  for (int p = 0; p < RSN_MAXPAIRS; ++p)
    if (p < npw) {
#pragma unroll
      for (int h = 0; h < 2; ++h) gv[p][h] = *reinterpret_cast<const bf16x8_t*>(grow + p * 64 + 8 * h);
    }
  constexpr int RSN_PRE = NB >= 3 ? 2 : 4;
  bf16x8_t bpre[RSN_PRE][2][NB];
#pragma unroll
  for (int p = 0; p < RSN_PRE; ++p)
    if (p < npw) {
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) bpre[p][h][nb] = *reinterpret_cast<const bf16x8_t*>(wrow[nb] + p * 64 + 8 * h);
    }
  // row sums: across the 4 k-groups of a row (lanes fr + 16*fq), then across the 8 waves
  ss += __shfl_xor(ss, 16, 64);
  ss += __shfl_xor(ss, 32, 64);
  if (fq == 0) ssq[wave][fr] = ss;
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int w = 0; w < SNT_WAVES; ++w) tot += ssq[w][fr];
  const float rstd = rsqrtf(tot / (float)D + eps);
  if (wave == 0 && fq == 0 && m0 + fr < M && rstd_out) rstd_out[m0 + fr] = rstd;
  // normalise in registers (single rounding to bf16), store y, and feed the rounded values to the MFMAs
  bf16_t* yrow = Y + (int64_t)row * D + wave * npw * 64 + 16 * fq;
  f32x4_t acc[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) acc[nb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int p = 0; p < RSN_MAXPAIRS; ++p) {
    if (p < npw) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        bf16x8_t y;
#pragma unroll
        for (int j = 0; j < 8; ++j) y[j] = (__bf16)((float)a[p][h][j] * rstd * (float)gv[p][h][j]);
        if (m0 + fr < M) *reinterpret_cast<bf16x8_t*>(yrow + p * 64 + 8 * h) = y;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          bf16x8_t b;
          if (p < RSN_PRE) b = bpre[p < RSN_PRE ? p : 0][h][nb];
          else b = *reinterpret_cast<const bf16x8_t*>(wrow[nb] + p * 64 + 8 * h);
          acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(y, b, acc[nb], 0, 0, 0);
        }
      }
    }
  }
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int e = 0; e < 4; ++e) part[wave][fq * 4 + e][nb * 16 + fr] = acc[nb][e];
  __syncthreads();
  const int orow = threadIdx.x >> 5, c0 = (threadIdx.x & 31) * 2;
  if (m0 + orow < M) {
    float v[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int c = c0 + e;
      float s = 0.f;
      if (c < NB * 16 && c < R) {
#pragma unroll
        for (int w = 0; w < SNT_WAVES; ++w) s += part[w][orow][c];
      }
      v[e] = s;
    }
    *reinterpret_cast<uint32_t*>(T + (int64_t)(m0 + orow) * SK_PAD + c0) = pack_bf2(v[0], v[1]);
  }
}

// x, y: [M, D] bf16 dense rows; g: norm weight [D]; W: [R, D] (row stride ldw); rstd fp32 [M]; t: [M, 64] bf16 (columns >= R zero).
// D a multiple of 512 and <= 4096 (8 waves x 64-wide k pairs); R <= 64.
extern "C" int llx_rmsnorm_skinny_nt(const void* x, const void* g, const void* W, int64_t ldw, void* y, float* rstd, void* t, int64_t M,
                                     int64_t D, int64_t R, float eps, hipStream_t stream) {
  LLX_REQUIRE(x && g && W && y && t, "llx_rmsnorm_skinny_nt: null pointer");
  LLX_REQUIRE(M > 0 && D > 0 && D % (SNT_WAVES * 64) == 0 && D <= SNT_WAVES * 64 * RSN_MAXPAIRS && R > 0 && R <= 64,
              "llx_rmsnorm_skinny_nt: need D a multiple of 512 and <= 4096, 0 < R <= 64 (D=%lld R=%lld)", (long long)D, (long long)R);
  LLX_REQUIRE(ldw % 8 == 0 && ((uintptr_t)x | (uintptr_t)g | (uintptr_t)W | (uintptr_t)y) % 16 == 0 && (uintptr_t)t % 8 == 0, "llx_rmsnorm_skinny_nt: alignment");
  const dim3 grid((unsigned)cdiv64(M, 16)), block(SNT_WAVES * 64);
  const int nb = (int)cdiv64(R, 16);
#define L(N) hipLaunchKernelGGL((rmsnorm_skinny_nt_kernel<N>), grid, block, 0, stream, (const bf16_t*)x, (const bf16_t*)g, (const bf16_t*)W, ldw, (bf16_t*)y, rstd, (bf16_t*)t, (int)M, (int)D, (int)R, eps)
  if (nb == 1) L(1); else if (nb == 2) L(2); else if (nb == 3) L(3); else L(4);
#undef L
  LLX_LAUNCH_CHECK("llx_rmsnorm_skinny_nt");
  return LLX_OK;
}
