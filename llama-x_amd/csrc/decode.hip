// Decode path (inference with a KV cache, a handful of query tokens per call): modelling/llama.py:76-90 (KVCache), :126-127,:135-137
// (cached keys/values through SDPA with the row-gathered causal mask), :189-194,:205-207 (Llama.forward with input_pos).
//
// Every linear of a decode step is a weight stream: M <= 4 activation rows against [N, K] bf16 weights that are read exactly once.
// The 256 x 256 MFMA tile kernel of gemm_bf16.hip covers such a product with N/256 workgroups (16-112 of 256 CUs); here
//   * gemv_kernel: all CUs stream weight rows (16 B per lane, whole 1-KiB row pieces per wave-instruction, non-temporal, 8 loads in
//     flight per lane), the activation row sits in LDS (optionally RMS-normalised on the way in: the norm of an 8-KiB row is cheaper
//     to redo per workgroup than a launch), fp32 FMA dot products, wave butterfly reduce, and the neighbours of the reference's call
//     sites in the epilogue: + residual | apply_rope on q,k + scatter of k,v into the caches | SwiGLU;
//   * attn_decode_kernel: the cache of one kv head is split over workgroups (>= 256 at B = 1), the G query heads of the group (x the
//     query tokens of the call) are served from ONE read of K and V, rows are loaded coalesced (16 lanes x 16 B = one 256-B row),
//     partial (m, l, o) per wave are merged by attn_decode_combine_kernel;
//   * mask_extent_kernel: the number of leading keys any query row may attend to (the mask is a bool tensor - the reference gathers
//     rows of its tril matrix, :194,:205 - so the extent is data; it stays on the device and sizes the key ranges of the workgroups);
//   * kv_scatter_kernel: KVCache.update (:83-90) for calls that do not go through the fused projection.
// HBM-bound: the step moves (weights + live K/V) bytes once; bench.py --config decode reports that against the 8 TB/s peak.
#include "common.h"

#define HD 128

// ------------------------------------------------------------------------------------------------- weight-streaming GEMV
enum { GV_NONE = 0, GV_RESIDUAL = 1, GV_QKV = 2, GV_SWIGLU = 3 };

struct GemvArgs {
  const bf16_t* W[3]; int64_t ldw[3]; int seg_end[3];  // output rows [seg_end[s-1], seg_end[s]) come from W[s] (row-major [rows, K])
  const bf16_t* x; int64_t ldx;                         // [M, K]
  const bf16_t* norm_w; float eps;                      // NORM: x <- rmsnorm(x) * norm_w, rounded to bf16 (nn.RMSNorm, single rounding)
  int M, N, K;
  bf16_t* out; int64_t ldo;                             // NONE / RESIDUAL: [M, N]; QKV: q rows [M, n_q]; SWIGLU: h [M, N / 2]
  const bf16_t* res; int64_t ldr;                       // RESIDUAL: [M, N]
  const float* rope; int n_q, n_k;                      // QKV: rows [0, n_q) = q heads, [n_q, n_q + n_k) = k heads, then v; table [>= M, 64, 2]
  bf16_t* kc; bf16_t* vc; int64_t c_sh, c_ss;           //      caches [KVH, Smax, 128] through (head, position) strides
  const int64_t* pos;                                   //      input_pos[M]
  // LoRA (modelling/lora.py:43): out += scale * (t . Bext[row]) with t = x . A^T computed by a previous launch of this kernel
  const bf16_t* bext[3]; int64_t ldb[3]; int t_off[3]; int rank[3];
  const bf16_t* t; int64_t ldt; float lora_scale;
};

__device__ __forceinline__ float dot8(const u32x4_t& w, const float (&xf)[8]) {
  float s = bflo(w[0]) * xf[0];
  s = __builtin_fmaf(bfhi(w[0]), xf[1], s);
  s = __builtin_fmaf(bflo(w[1]), xf[2], s);
  s = __builtin_fmaf(bfhi(w[1]), xf[3], s);
  s = __builtin_fmaf(bflo(w[2]), xf[4], s);
  s = __builtin_fmaf(bfhi(w[2]), xf[5], s);
  s = __builtin_fmaf(bflo(w[3]), xf[6], s);
  s = __builtin_fmaf(bfhi(w[3]), xf[7], s);
  return s;
}

template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// Sum over the 64 lanes, every lane gets it, all in the vector pipe: DPP inside the rows of 16 (quad swaps, half mirror, mirror), then
// v_permlane16_swap / v_permlane32_swap pair the rows (common.h's wave_sum takes six trips through the LDS crossbar; other sum order).
__device__ __forceinline__ float wave_sum_valu(float v) {
  v += dpp_f32<0xB1>(v);
  v += dpp_f32<0x4E>(v);
  v += dpp_f32<0x141>(v);
  v += dpp_f32<0x140>(v);
  // (__uint_as_float on the elements, not __builtin_bit_cast: hipcc 7.2 reads element 0 for BOTH halves of the pair through bit_cast)
  const auto s16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(s16[0]) + __uint_as_float(s16[1]);
  const auto s32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(s32[0]) + __uint_as_float(s32[1]);
}

// RPW output rows per wave at a time (4, or 2 for the narrow projections: twice the waves, each with the same 8 loads in flight, so
// that 4096 output rows still put 2 workgroups on every CU); a step is 8 / RPW pieces of 512 elements (64 lanes x 8) of those rows.
template <int MT, int EPI, bool NORM, int RPW = 4>
__global__ __launch_bounds__(256, 2) void gemv_kernel(const GemvArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int K = a.K;
  static_assert(RPW == 4 || RPW == 2, "rows per wave");
  constexpr int PIECES = 8 / RPW, STEP = 512 * PIECES;
  const int nsteps = (K + STEP - 1) / STEP;  // a step = PIECES 512-element pieces of RPW weight rows: 8 x 16-byte loads per lane
  const int Kp = nsteps * STEP;              // the LDS copy of x is zero-padded to whole steps: weight lanes past K multiply zeros
  bf16_t* xs = reinterpret_cast<bf16_t*>(smem);  // [MT][Kp]
  float* red = reinterpret_cast<float*>(smem + (size_t)MT * Kp * 2);

  // ---- row groups: RPW output rows per wave at a time.  SWIGLU: the gate and up rows of RPW / 2 hidden units (W[0] rows UPG g .. and
  // W[1] rows UPG g ..; r = matrix * UPG + unit), so that the epilogue has g and u of one unit side by side.
  constexpr int UPG = RPW / 2;  // hidden units per SwiGLU group
  struct Grp { const bf16_t* wr[RPW]; int row0, seg; };
  auto setup = [&](int g, Grp& G) {
    if constexpr (EPI == GV_SWIGLU) {
      const int half = a.N / 2;
      G.row0 = UPG * g; G.seg = 0;
#pragma unroll
      for (int r = 0; r < RPW; ++r) G.wr[r] = a.W[r / UPG] + (int64_t)min(G.row0 + (r % UPG), half - 1) * a.ldw[r / UPG];
    } else {
      G.row0 = RPW * g;
      G.seg = G.row0 >= a.seg_end[0] ? (G.row0 >= a.seg_end[1] ? 2 : 1) : 0;  // wave-uniform (segment boundaries are multiples of 4)
      const int base = G.seg == 0 ? 0 : a.seg_end[G.seg - 1];
      const int last = a.seg_end[G.seg] - 1 - base;
#pragma unroll
      for (int r = 0; r < RPW; ++r) G.wr[r] = a.W[G.seg] + (int64_t)min(G.row0 - base + r, last) * a.ldw[G.seg];
    }
  };
  auto load_step = [&](u32x4_t (&w)[8], const Grp& G, int s) {
#pragma unroll
    for (int j = 0; j < PIECES; ++j) {
      const int k = (PIECES * s + j) * 512 + lane * 8;
      const int kk = k < K ? k : 0;  // lanes past the row end re-read its start; their x is zero
#pragma unroll
      for (int r = 0; r < RPW; ++r) w[RPW * j + r] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(G.wr[r] + kk));
    }
  };
  const int nwaves = gridDim.x * 4;
  const int ngroups = EPI == GV_SWIGLU ? (a.N / 2 + UPG - 1) / UPG : (a.N + RPW - 1) / RPW;
  int g = blockIdx.x * 4 + wave, s = 0;
  Grp cur, nxt;
  u32x4_t WA[8], WB[8];
  // the first weight loads do not depend on x: they fly while the activation rows are staged (and normalised)
  if (g < ngroups) { setup(g, cur); load_step(WA, cur, 0); }

  // ---- the activation rows into LDS (normalised on the way when NORM)
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const bf16_t* xr = a.x + (int64_t)min(m, a.M - 1) * a.ldx;
    float rstd = 1.f;
    if constexpr (NORM) {
      float ss = 0.f;
      for (int i = tid * 8; i < K; i += 2048) {
        const u32x4_t v = *reinterpret_cast<const u32x4_t*>(xr + i);
#pragma unroll
        for (int e = 0; e < 4; ++e) ss += bflo(v[e]) * bflo(v[e]) + bfhi(v[e]) * bfhi(v[e]);
      }
      ss = block_sum(ss, red);
      rstd = rsqrtf(ss / (float)K + a.eps);
    }
    for (int i = tid * 8; i < Kp; i += 2048) {
      u32x4_t v = {0u, 0u, 0u, 0u};
      if (i < K) {
        v = *reinterpret_cast<const u32x4_t*>(xr + i);
        if constexpr (NORM) {
          const u32x4_t w = *reinterpret_cast<const u32x4_t*>(a.norm_w + i);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = pack_bf2(bflo(v[e]) * rstd * bflo(w[e]), bfhi(v[e]) * rstd * bfhi(w[e]));
        }
      }
      *reinterpret_cast<u32x4_t*>(xs + (size_t)m * Kp + i) = v;
    }
  }
  __syncthreads();

  float acc[RPW][MT];
#pragma unroll
  for (int r = 0; r < RPW; ++r)
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[r][m] = 0.f;
  auto compute_step = [&](const u32x4_t (&w)[8], int st) {
#pragma unroll
    for (int j = 0; j < PIECES; ++j) {
      const int k = (PIECES * st + j) * 512 + lane * 8;
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const u32x4_t xv = *reinterpret_cast<const u32x4_t*>(xs + (size_t)m * Kp + k);
        float xf[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) { xf[2 * e] = bflo(xv[e]); xf[2 * e + 1] = bfhi(xv[e]); }
#pragma unroll
        for (int r = 0; r < RPW; ++r) acc[r][m] += dot8(w[RPW * j + r], xf);
      }
    }
  };
  auto finish = [&](const Grp& G) {
    const int row0 = G.row0, seg = G.seg;
    // LoRA extension: lanes 0 .. rank/8-1 hold 8 elements of the row's B factor each
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
      const int s2 = EPI == GV_SWIGLU ? (r / UPG) : seg;
      const bf16_t* bp = a.bext[s2];
      if (bp != nullptr && lane * 8 < a.rank[s2]) {
        int lrow;
        if constexpr (EPI == GV_SWIGLU) {
          lrow = min(row0 + (r % UPG), a.N / 2 - 1);
        } else {
          const int base = seg == 0 ? 0 : a.seg_end[seg - 1];
          lrow = min(row0 - base + r, a.seg_end[seg] - 1 - base);
        }
        const u32x4_t bv = *reinterpret_cast<const u32x4_t*>(bp + (int64_t)lrow * a.ldb[s2] + lane * 8);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const u32x4_t tv = *reinterpret_cast<const u32x4_t*>(a.t + (int64_t)min(m, a.M - 1) * a.ldt + a.t_off[s2] + lane * 8);
          float xf[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) { xf[2 * e] = bflo(tv[e]); xf[2 * e + 1] = bfhi(tv[e]); }
          acc[r][m] += a.lora_scale * dot8(bv, xf);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < RPW; ++r)
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[r][m] = wave_sum_valu(acc[r][m]);
    // ---- epilogue: lane m writes token m (every lane holds every sum)
    if (lane < MT && lane < a.M) {
      float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int r = 0; r < RPW; ++r) {
        float sm = acc[r][0];
#pragma unroll
        for (int m = 1; m < MT; ++m) sm = lane == m ? acc[r][m] : sm;
        v[r] = bf2f(f2bf(sm));  // the linear's bf16 output
      }
      const int m = lane;
      if constexpr (EPI == GV_SWIGLU) {
        // h = silu(g) * u with the roundings of the bf16 eager graph (modelling/llama.py:150-152), as swiglu_fwd8
        const int half = a.N / 2;
#pragma unroll
        for (int j = 0; j < UPG; ++j) {
          const float gg = v[j], uu = v[UPG + j];
          const float sg = bf2f(f2bf(gg * sigmoidf_(gg)));
          if (row0 + j < half) a.out[(int64_t)m * a.ldo + row0 + j] = f2bf(sg * uu);
        }
      } else if constexpr (EPI == GV_QKV) {
        // apply_rope on q and k (modelling/llama.py:63-73,122-123; the table row is the token's index IN THIS CALL, :207), then
        // KVCache.update (:83-90) for k and v
        const bool is_q = row0 < a.n_q, is_k = !is_q && row0 < a.n_q + a.n_k;
        const int hrow = is_q ? row0 : (is_k ? row0 - a.n_q : row0 - a.n_q - a.n_k);
        const int d = hrow & (HD - 1);
        if (is_q || is_k) {
          const float* tp = a.rope + ((int64_t)m * 64 + (d >> 1)) * 2;
          const float c0 = tp[0], s0 = tp[1], c1 = RPW == 4 ? tp[2] : 1.f, s1 = RPW == 4 ? tp[3] : 0.f;
          const float y0 = v[0] * c0 - v[1] * s0, y1 = v[1] * c0 + v[0] * s0, y2 = v[2] * c1 - v[3] * s1, y3 = v[3] * c1 + v[2] * s1;
          v[0] = y0; v[1] = y1; v[2] = y2; v[3] = y3;
        }
        u32x2_t pk;
        pk[0] = pack_bf2(v[0], v[1]);
        pk[1] = pack_bf2(v[2], v[3]);
        bf16_t* dst = is_q ? a.out + (int64_t)m * a.ldo + row0 : (is_k ? a.kc : a.vc) + (int64_t)(hrow >> 7) * a.c_sh + a.pos[m] * a.c_ss + d;
        if constexpr (RPW == 4) *reinterpret_cast<u32x2_t*>(dst) = pk;
        else *reinterpret_cast<uint32_t*>(dst) = pk[0];  // one rotation pair
      } else {
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
          if (row0 + r < a.N) {
            float o = v[r];
            if constexpr (EPI == GV_RESIDUAL) o += bf2f(a.res[(int64_t)m * a.ldr + row0 + r]);  // bf16 output + bf16 residual, rounded
            a.out[(int64_t)m * a.ldo + row0 + r] = f2bf(o);
          }
        }
      }
    }
#pragma unroll
    for (int r = 0; r < RPW; ++r)
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[r][m] = 0.f;
  };
  // software pipeline over the flattened (row group, step) sequence of this wave: while one register set is consumed the next step's
  // 8 loads (of this group or of the wave's next group) are in flight
  auto body = [&](const u32x4_t (&wc)[8], u32x4_t (&wn)[8]) -> bool {
    int g2 = g, s2 = s + 1;
    bool newg = false;
    if (s2 == nsteps) { g2 = g + nwaves; s2 = 0; newg = true; }
    const bool has2 = g2 < ngroups;
    if (has2) {
      if (newg) { setup(g2, nxt); load_step(wn, nxt, s2); }
      else load_step(wn, cur, s2);
    }
    compute_step(wc, s);
    if (newg) {
      finish(cur);
      cur = nxt;
    }
    g = g2; s = s2;
    return has2;
  };
  if (g < ngroups) {
    while (true) {
      if (!body(WA, WB)) break;
      if (!body(WB, WA)) break;
    }
  }
}

template <int MT, int EPI, int RPW>
static int launch_gemv_n(const GemvArgs& a, int grid, size_t lds, hipStream_t stream) {
  if (a.norm_w) hipLaunchKernelGGL((gemv_kernel<MT, EPI, true, RPW>), dim3(grid), dim3(256), lds, stream, a);
  else hipLaunchKernelGGL((gemv_kernel<MT, EPI, false, RPW>), dim3(grid), dim3(256), lds, stream, a);
  LLX_LAUNCH_CHECK("llx_gemv_bf16");
  return LLX_OK;
}

template <int MT, int RPW>
static int launch_gemv_m(const GemvArgs& a, int epi, int grid, size_t lds, hipStream_t stream) {
  switch (epi) {
    case GV_NONE: return launch_gemv_n<MT, GV_NONE, RPW>(a, grid, lds, stream);
    case GV_RESIDUAL: return launch_gemv_n<MT, GV_RESIDUAL, RPW>(a, grid, lds, stream);
    case GV_QKV: return launch_gemv_n<MT, GV_QKV, RPW>(a, grid, lds, stream);
    default: return launch_gemv_n<MT, GV_SWIGLU, RPW>(a, grid, lds, stream);
  }
}

// out[M, N] = epilogue( x[M, K] . [W0; W1; W2]^T ), M <= 4, bf16, fp32 accumulate; F.linear at decode shapes
// (modelling/llama.py:118-120,140,152,216).  W_s: [n_s, K] row-major with row stride ldw_s (nullable from s = 1 on; n_s % 4 == 0 except
// the last), K % 8 == 0.  norm_w (nullable): x is RMS-normalised with this weight first (llama.py:172-173,215).
// epilogue 0: out [M, N] | 1: + res [M, N] | 2 (q|k|v): rows [0, n_q) RoPE -> out [M, n_q]; [n_q, n_q + n_k) RoPE -> k cache;
// rest -> v cache, at input_pos[m] (device int64), caches through (head, position) strides | 3 (gate|up = W0|W1, N = 2 n_0):
// out [M, N/2] = silu(gate) * up.  LoRA (nullable bext_s [n_s, rank_s], t [M, sum rank] bf16 = x . A^T, offsets t_off_s, scale):
// out += scale * t_s . bext_s[row].
extern "C" int llx_gemv_bf16(const void* w0, int64_t ldw0, int64_t n0, const void* w1, int64_t ldw1, int64_t n1, const void* w2, int64_t ldw2,
                             int64_t n2, const void* x, int64_t ldx, int64_t M, int64_t K, const void* norm_w, float eps, int epilogue,
                             void* out, int64_t ldo, const void* res, int64_t ldr, const float* rope, int64_t n_q, int64_t n_k, void* k_cache,
                             void* v_cache, int64_t c_sh, int64_t c_ss, const int64_t* input_pos, const void* bext0, const void* bext1,
                             const void* bext2, int64_t rank0, int64_t rank1, int64_t rank2, const void* t, int64_t ldt, float lora_scale,
                             hipStream_t stream) {
  LLX_REQUIRE(w0 && x && out, "llx_gemv_bf16: null pointer");
  LLX_REQUIRE(M >= 1 && M <= 4, "llx_gemv_bf16: M=%lld outside 1..4 (larger row counts run the MFMA GEMM)", (long long)M);
  LLX_REQUIRE(K > 0 && K % 8 == 0 && K <= 32768, "llx_gemv_bf16: K=%lld must be a multiple of 8 and at most 32768", (long long)K);
  LLX_REQUIRE(n0 > 0 && n1 >= 0 && n2 >= 0 && (w1 || n1 == 0) && (w2 || n2 == 0), "llx_gemv_bf16: bad segment sizes");
  LLX_REQUIRE((n1 == 0 || n0 % 4 == 0) && (n2 == 0 || n1 % 4 == 0), "llx_gemv_bf16: inner segment sizes must be multiples of 4");
  LLX_REQUIRE(ldw0 % 8 == 0 && ldw1 % 8 == 0 && ldw2 % 8 == 0 && ldx % 8 == 0, "llx_gemv_bf16: row strides must be multiples of 8 elements");
  LLX_REQUIRE(((uintptr_t)w0 | (uintptr_t)w1 | (uintptr_t)w2 | (uintptr_t)x | (uintptr_t)norm_w) % 16 == 0, "llx_gemv_bf16: pointers must be 16-byte aligned");
  LLX_REQUIRE(epilogue >= GV_NONE && epilogue <= GV_SWIGLU, "llx_gemv_bf16: unknown epilogue %d", epilogue);
  const int64_t N = n0 + n1 + n2;
  LLX_REQUIRE(N < (1 << 30), "llx_gemv_bf16: too many rows");
  LLX_REQUIRE(epilogue != GV_RESIDUAL || res, "llx_gemv_bf16: residual missing");
  LLX_REQUIRE(epilogue != GV_SWIGLU || (n0 == n1 && n2 == 0 && w1), "llx_gemv_bf16: the SwiGLU epilogue takes gate and up weights of equal size");
  LLX_REQUIRE(epilogue != GV_QKV || (rope && k_cache && v_cache && input_pos && n_q % HD == 0 && n_k % HD == 0 && (N - n_q - n_k) % HD == 0 &&
                                     n_q + n_k <= N && (uintptr_t)rope % 8 == 0 && ((uintptr_t)out | (uintptr_t)k_cache | (uintptr_t)v_cache) % 8 == 0 &&
                                     ldo % 4 == 0 && c_sh % 4 == 0 && c_ss % 4 == 0),
              "llx_gemv_bf16: bad q|k|v epilogue arguments");
  const bool lora = bext0 || bext1 || bext2;
  LLX_REQUIRE(!lora || (t && ldt % 8 == 0 && (uintptr_t)t % 16 == 0 && rank0 % 8 == 0 && rank1 % 8 == 0 && rank2 % 8 == 0 && rank0 <= 512 && rank1 <= 512 &&
                        rank2 <= 512 && ((uintptr_t)bext0 | (uintptr_t)bext1 | (uintptr_t)bext2) % 16 == 0),
              "llx_gemv_bf16: bad LoRA operands (ranks must be multiples of 8, at most 512)");
  GemvArgs a;
  a.W[0] = (const bf16_t*)w0; a.W[1] = (const bf16_t*)(w1 ? w1 : w0); a.W[2] = (const bf16_t*)(w2 ? w2 : w0);
  a.ldw[0] = ldw0; a.ldw[1] = w1 ? ldw1 : ldw0; a.ldw[2] = w2 ? ldw2 : ldw0;
  a.seg_end[0] = (int)n0; a.seg_end[1] = (int)(n0 + n1); a.seg_end[2] = (int)N;
  if (n1 == 0) { a.seg_end[0] = a.seg_end[1] = (int)N; }       // single source: every row is segment 0
  else if (n2 == 0) { a.seg_end[1] = (int)N; }
  a.x = (const bf16_t*)x; a.ldx = ldx; a.norm_w = (const bf16_t*)norm_w; a.eps = eps;
  a.M = (int)M; a.N = (int)N; a.K = (int)K;
  a.out = (bf16_t*)out; a.ldo = ldo; a.res = (const bf16_t*)res; a.ldr = ldr;
  a.rope = rope; a.n_q = (int)n_q; a.n_k = (int)n_k; a.kc = (bf16_t*)k_cache; a.vc = (bf16_t*)v_cache; a.c_sh = c_sh; a.c_ss = c_ss; a.pos = input_pos;
  a.bext[0] = (const bf16_t*)bext0; a.bext[1] = (const bf16_t*)bext1; a.bext[2] = (const bf16_t*)bext2;
  a.ldb[0] = rank0; a.ldb[1] = rank1; a.ldb[2] = rank2;
  a.rank[0] = (int)rank0; a.rank[1] = (int)rank1; a.rank[2] = (int)rank2;
  a.t_off[0] = 0; a.t_off[1] = (int)rank0; a.t_off[2] = (int)(rank0 + rank1);
  a.t = (const bf16_t*)t; a.ldt = ldt; a.lora_scale = lora_scale;
  if (n1 == 0 && lora) { a.bext[1] = a.bext[2] = a.bext[0]; a.ldb[1] = a.ldb[2] = a.ldb[0]; a.rank[1] = a.rank[2] = a.rank[0]; a.t_off[1] = a.t_off[2] = 0; }
  // rows per wave: 2 (steps of 2048 elements per row: half the dependent steps of the 4-row form on every product of the decode step and
  // twice the waves where N <= 4096 would fill only half of the 2048 slots: 3.51 -> 3.39 ms per token; one row per wave: 3.43);
  // LLX_GEMV_RPW=4: the 4-row form
  static const int rpw_knob = [] { const char* e = getenv("LLX_GEMV_RPW"); return e ? atoi(e) : 0; }();
  const int rpw = rpw_knob == 4 ? 4 : 2;
  const int64_t groups = epilogue == GV_SWIGLU ? (N / 2 + rpw / 2 - 1) / (rpw / 2) : (N + rpw - 1) / rpw;
  // the wave count is trimmed so that every wave gets the same number of row groups where possible
  static const int wave_cap = [] { const char* e = getenv("LLX_GEMV_WAVES"); return e && atoi(e) >= 256 ? atoi(e) : 2048; }();
  const int64_t per_wave = cdiv64(groups, wave_cap);
  const int grid = (int)cdiv64(cdiv64(groups, per_wave), 4);
  const int64_t kstep = 512 * (8 / rpw);
  const int64_t Kp = cdiv64(K, kstep) * kstep;
  // the build for m rows stages MT = 1 | 2 | 4 rows of x in LDS; above 64 KiB the rows go in pairs (two passes over the weights)
  auto run = [&](int m0, int mc) -> int {
    GemvArgs b = a;
    b.M = mc;
    b.x = a.x + (int64_t)m0 * a.ldx;
    b.out = a.out + (int64_t)m0 * a.ldo;
    if (a.res) b.res = a.res + (int64_t)m0 * a.ldr;
    if (a.rope) b.rope = a.rope + (int64_t)m0 * 128;
    if (a.pos) b.pos = a.pos + m0;
    if (a.t) b.t = a.t + (int64_t)m0 * a.ldt;
    const int MT = mc == 1 ? 1 : (mc == 2 ? 2 : 4);
    const size_t lds = (size_t)MT * Kp * 2 + 64;
    if (rpw == 2) {
      switch (MT) {
        case 1: return launch_gemv_m<1, 2>(b, epilogue, grid, lds, stream);
        case 2: return launch_gemv_m<2, 2>(b, epilogue, grid, lds, stream);
        default: return launch_gemv_m<4, 2>(b, epilogue, grid, lds, stream);
      }
    }
    switch (MT) {
      case 1: return launch_gemv_m<1, 4>(b, epilogue, grid, lds, stream);
      case 2: return launch_gemv_m<2, 4>(b, epilogue, grid, lds, stream);
      default: return launch_gemv_m<4, 4>(b, epilogue, grid, lds, stream);
    }
  };
  if (M > 2 && 4 * Kp * 2 + 64 > 64 * 1024) {
    const int rc = run(0, 2);
    return rc != LLX_OK ? rc : run(2, (int)M - 2);
  }
  LLX_REQUIRE((M > 2 ? 4 : M) * Kp * 2 + 64 <= 64 * 1024, "llx_gemv_bf16: M * K too large for the LDS stage");
  return run(0, (int)M);
}

// ------------------------------------------------------------------------------------------------- mask extent / cache scatter
// *extent = 1 + the largest key index that ANY of the `rows` mask rows (uint8 / bool, row stride m_sr) allows; 0 if none.
__global__ __launch_bounds__(256) void mask_extent_kernel(const uint8_t* __restrict__ mask, int64_t m_sr, int rows, int Skv, int* __restrict__ extent) {
  __shared__ int red[4];
  int best = 0;
  for (int r = 0; r < rows; ++r) {
    const uint8_t* mr = mask + (int64_t)r * m_sr;
    for (int k = threadIdx.x; k < Skv; k += 256)
      if (mr[k]) best = max(best, k + 1);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) best = max(best, __shfl_xor(best, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = best;
  __syncthreads();
  if (threadIdx.x == 0) *extent = max(max(red[0], red[1]), max(red[2], red[3]));
}

extern "C" int llx_mask_extent(const void* mask, int64_t row_stride, int64_t rows, int64_t Skv, int* extent, hipStream_t stream) {
  LLX_REQUIRE(mask && extent && rows > 0 && Skv > 0 && Skv < (1 << 30) && rows < (1 << 20), "llx_mask_extent: bad arguments");
  hipLaunchKernelGGL(mask_extent_kernel, dim3(1), dim3(256), 0, stream, (const uint8_t*)mask, row_stride, (int)rows, (int)Skv, extent);
  LLX_LAUNCH_CHECK("llx_mask_extent");
  return LLX_OK;
}

// KVCache.update (modelling/llama.py:83-90): cache[b, h, input_pos[l], :] = src[b, h, l, :] for k and v in one launch.
__global__ __launch_bounds__(256) void kv_scatter_kernel(const bf16_t* __restrict__ k, const bf16_t* __restrict__ v, int64_t s_sb, int64_t s_sh, int64_t s_ss,
                                                         bf16_t* __restrict__ kc, bf16_t* __restrict__ vc, int64_t c_sb, int64_t c_sh, int64_t c_ss,
                                                         const int64_t* __restrict__ pos, int L, int KVH, int Smax) {
  const int chunk = threadIdx.x & 15, which = (threadIdx.x >> 4) & 1, li = blockIdx.x * 8 + (threadIdx.x >> 5);
  const int h = blockIdx.y, b = blockIdx.z;
  if (li >= L) return;
  const int64_t p = pos[li];
  if (p < 0 || p >= Smax) return;  // torch's index_put would raise; an out-of-range position must never write outside the cache
  const bf16_t* src = (which ? v : k) + b * s_sb + h * s_sh + (int64_t)li * s_ss + chunk * 8;
  bf16_t* dst = (which ? vc : kc) + b * c_sb + h * c_sh + p * c_ss + chunk * 8;
  *reinterpret_cast<u32x4_t*>(dst) = *reinterpret_cast<const u32x4_t*>(src);
}

extern "C" int llx_kv_scatter(const void* k, const void* v, int64_t s_sb, int64_t s_sh, int64_t s_ss, void* k_cache, void* v_cache, int64_t c_sb,
                              int64_t c_sh, int64_t c_ss, const int64_t* input_pos, int64_t B, int64_t KVH, int64_t L, int64_t Smax,
                              int64_t head_dim, hipStream_t stream) {
  LLX_REQUIRE(k && v && k_cache && v_cache && input_pos, "llx_kv_scatter: null pointer");
  LLX_REQUIRE(head_dim == HD, "llx_kv_scatter: head_dim=%lld unsupported (only 128)", (long long)head_dim);
  LLX_REQUIRE(B > 0 && KVH > 0 && L > 0 && Smax > 0 && B < 65536 && KVH < 65536, "llx_kv_scatter: bad sizes");
  LLX_REQUIRE(((s_sb | s_sh | s_ss | c_sb | c_sh | c_ss) % 8) == 0 && ((uintptr_t)k | (uintptr_t)v | (uintptr_t)k_cache | (uintptr_t)v_cache) % 16 == 0,
              "llx_kv_scatter: rows must be 16-byte aligned");
  hipLaunchKernelGGL(kv_scatter_kernel, dim3((unsigned)cdiv64(L, 8), (unsigned)KVH, (unsigned)B), dim3(256), 0, stream, (const bf16_t*)k, (const bf16_t*)v,
                     s_sb, s_sh, s_ss, (bf16_t*)k_cache, (bf16_t*)v_cache, c_sb, c_sh, c_ss, input_pos, (int)L, (int)KVH, (int)Smax);
  LLX_LAUNCH_CHECK("llx_kv_scatter");
  return LLX_OK;
}

// ------------------------------------------------------------------------------------------------- decode attention
#define DEC_PART 132  // floats per partial: o[128], m, l, pad

struct DecodeArgs {
  const bf16_t* q; int64_t q_sb, q_sh, q_ss;             // [B, H, M, 128] through strides
  const bf16_t* kc; const bf16_t* vc; int64_t c_sb, c_sh, c_ss;
  const uint8_t* mask; int64_t m_sb, m_sh, m_sq;         // bool [.., M, Skv], broadcast strides, last dim dense
  const int* extent;                                     // nullable: keys >= *extent are masked for every row
  float* part;                                           // [B, H, M, nsplit, DEC_PART]
  int B, H, KVH, M, Skv, nsplit;
  float scale_log2;
};

// One workgroup = 4 waves = one key range of one (batch, kv head); ROWS = G * M query rows (the G heads of the group x the tokens of
// the call) share every K / V row read.  A 16-lane group owns one key per load (lane = 16-byte chunk of the 256-byte row): four keys
// per wave-instruction, sixteen per wave and step (8 x 16-byte loads in flight per lane), 64 per workgroup step.
// DEC_KPG = keys per lane group and step: 4 for up to 4 query rows (the batch-1 decode step), fewer for more rows (register budget)
template <int ROWS, int DEC_KPG>
__global__ __launch_bounds__(256) void attn_decode_kernel(const DecodeArgs a) {
  // partials that meet in LDS: one per wave after a shuffle merge of its four lane groups, or - for up to 4 rows, where 16 partials fit
  // in 32 KB - one per lane group straight from the registers (the 80 dependent shuffles of the merge cost 1.6 us of a 7.5 us block)
  constexpr int NPART = ROWS <= 4 ? 16 : 4;
  __shared__ float sm_o[NPART][ROWS][HD];
  __shared__ float sm_ml[NPART][ROWS][2];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int split = blockIdx.x, kvh = blockIdx.y, b = blockIdx.z;
  const int G = a.H / a.KVH;
  const int grp = lane >> 4, c = lane & 15;
  const int ext = a.extent ? min(*a.extent, a.Skv) : a.Skv;
  const int span = ((ext + a.nsplit - 1) / a.nsplit + 15) & ~15;  // keys per workgroup, multiple of 16
  const int k_lo = split * span, k_hi = min(ext, k_lo + span);

  // (the query rows are only REQUESTED here: they are unpacked after the first K / V loads have been issued - unpacking them first
  // put a full memory round trip, 1.2 us, in front of those loads)
  u32x4_t qraw[ROWS];
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
    const int g = r / a.M, m = r % a.M;  // row r = (head g of the group, token m)
    const int h = kvh * G + min(g, G - 1);
    qraw[r] = *reinterpret_cast<const u32x4_t*>(a.q + b * a.q_sb + h * a.q_sh + (int64_t)m * a.q_ss + c * 8);
  }
  float qf[ROWS][8];
  float mx[ROWS], ls[ROWS], o[ROWS][8];
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
    mx[r] = -INFINITY; ls[r] = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[r][e] = 0.f;
  }
  const bf16_t* kb = a.kc + b * a.c_sb + kvh * a.c_sh + c * 8;
  const bf16_t* vb = a.vc + b * a.c_sb + kvh * a.c_sh + c * 8;
  const bool shared_mask = a.m_sh == 0;  // the reference's mask (tril rows gathered at input_pos) is the same for every head
  const uint8_t* mbase = a.mask + b * a.m_sb;
  // Two register sets: the K / V rows and mask bytes of step i+1 are in flight while step i is computed - a wave walks only 2-8
  // steps, so with the loads issued step by step every step paid a full HBM round trip (the kernel ran at 1.5 TB/s).  The mask bytes
  // are only LOADED in the load step, unconditionally and with clamped indices: turning them into bits needs their values and would
  // make the step wait for its own loads.
  auto load_step = [&](int k0, u32x4_t (&kv)[DEC_KPG], u32x4_t (&vv)[DEC_KPG], uint8_t (&mb)[DEC_KPG][ROWS]) {
#pragma unroll
    for (int j = 0; j < DEC_KPG; ++j) {
      const int key = k0 + j * 4 + grp;
      const int kk = min(key, a.Skv - 1);
      kv[j] = *reinterpret_cast<const u32x4_t*>(kb + (int64_t)kk * a.c_ss);
      vv[j] = *reinterpret_cast<const u32x4_t*>(vb + (int64_t)kk * a.c_ss);
#pragma unroll
      for (int r = 0; r < ROWS; ++r) {
        // shared mask: entry m = token m (expanded to the G heads below); per-head mask: entry r = (head g, token m)
        const int g = min(r / a.M, G - 1), m = shared_mask ? min(r, a.M - 1) : r % a.M;
        mb[j][r] = mbase[(shared_mask ? (int64_t)0 : (int64_t)(kvh * G + g) * a.m_sh) + (int64_t)m * a.m_sq + kk];
      }
    }
  };
  auto compute_step = [&](int k0, const u32x4_t (&kv)[DEC_KPG], const u32x4_t (&vv)[DEC_KPG], const uint8_t (&mb)[DEC_KPG][ROWS]) {
    uint32_t allow[DEC_KPG];  // shared mask: bit m = token m; per-head mask: bit r = query row r
#pragma unroll
    for (int j = 0; j < DEC_KPG; ++j) {
      uint32_t w = 0;
#pragma unroll
      for (int r = 0; r < ROWS; ++r) {
        const bool valid = shared_mask ? r < a.M : r / a.M < G;
        if (valid && mb[j][r] != 0) w |= 1u << r;
      }
      allow[j] = (k0 + j * 4 + grp) < k_hi ? w : 0u;
    }
    float s[DEC_KPG][ROWS];
#pragma unroll
    for (int j = 0; j < DEC_KPG; ++j) {
      float kf[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) { kf[2 * e] = bflo(kv[j][e]); kf[2 * e + 1] = bfhi(kv[j][e]); }
#pragma unroll
      for (int r = 0; r < ROWS; ++r) {
        float d = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) d = __builtin_fmaf(qf[r][e], kf[e], d);
        // sum over the 16 lanes of the key's group in the vector pipe (DPP: quad swaps, then the mirrors pair quads and halves - the same
        // additions as an xor butterfly, without four trips through the LDS crossbar)
        d += dpp_f32<0xB1>(d);   // quad_perm [1,0,3,2]
        d += dpp_f32<0x4E>(d);   // quad_perm [2,3,0,1]
        d += dpp_f32<0x141>(d);  // row_half_mirror
        d += dpp_f32<0x140>(d);  // row_mirror
        const bool ok = shared_mask ? ((allow[j] >> (r % a.M)) & 1u) != 0 : ((allow[j] >> r) & 1u) != 0;
        s[j][r] = ok ? d : -INFINITY;
      }
    }
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      float mn = mx[r];
#pragma unroll
      for (int j = 0; j < DEC_KPG; ++j) mn = fmaxf(mn, s[j][r]);
      if (mn == -INFINITY) continue;  // nothing to attend to so far for this row in this lane group
      const float alpha = __builtin_amdgcn_exp2f(mx[r] - mn);
      float p[DEC_KPG], psum = 0.f;
#pragma unroll
      for (int j = 0; j < DEC_KPG; ++j) { p[j] = __builtin_amdgcn_exp2f(s[j][r] - mn); psum += p[j]; }
      ls[r] = ls[r] * alpha + psum;
      mx[r] = mn;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float lo = o[r][2 * e] * alpha, hi = o[r][2 * e + 1] * alpha;
#pragma unroll
        for (int j = 0; j < DEC_KPG; ++j) { lo = __builtin_fmaf(p[j], bflo(vv[j][e]), lo); hi = __builtin_fmaf(p[j], bfhi(vv[j][e]), hi); }
        o[r][2 * e] = lo; o[r][2 * e + 1] = hi;
      }
    }
  };
  {
    const int kstep = 16 * DEC_KPG;
    u32x4_t kvA[DEC_KPG], vvA[DEC_KPG], kvB[DEC_KPG], vvB[DEC_KPG];
    uint8_t mbA[DEC_KPG][ROWS], mbB[DEC_KPG][ROWS];
    int k0 = k_lo + wave * (4 * DEC_KPG);
    if (k0 < k_hi) load_step(k0, kvA, vvA, mbA);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < ROWS; ++r)
#pragma unroll
      for (int e = 0; e < 4; ++e) { qf[r][2 * e] = bflo(qraw[r][e]) * a.scale_log2; qf[r][2 * e + 1] = bfhi(qraw[r][e]) * a.scale_log2; }
    while (k0 < k_hi) {
      if (k0 + kstep < k_hi) load_step(k0 + kstep, kvB, vvB, mbB);
      compute_step(k0, kvA, vvA, mbA);
      k0 += kstep;
      if (k0 >= k_hi) break;
      if (k0 + kstep < k_hi) load_step(k0 + kstep, kvA, vvA, mbA);
      compute_step(k0, kvB, vvB, mbB);
      k0 += kstep;
    }
  }
  // merge the four lane groups of the wave (lanes l, l^16, l^32, l^48 hold the same dims of different keys), then the four waves
  // through LDS: ONE partial (m, l, o) per workgroup and query row
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
    if constexpr (NPART == 4) {
#pragma unroll
      for (int off = 16; off <= 32; off <<= 1) {
        const float m2 = __shfl_xor(mx[r], off, 64), l2 = __shfl_xor(ls[r], off, 64);
        const float mn = fmaxf(mx[r], m2);
        const float fa = mn == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(mx[r] - mn), fb = mn == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(m2 - mn);
        ls[r] = ls[r] * fa + l2 * fb;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[r][e] = o[r][e] * fa + __shfl_xor(o[r][e], off, 64) * fb;
        mx[r] = mn;
      }
    }
    const int slot = NPART == 4 ? wave : wave * 4 + grp;
    if (NPART == 16 || grp == 0) {
      *reinterpret_cast<f32x4_t*>(&sm_o[slot][r][c * 8]) = f32x4_t{o[r][0], o[r][1], o[r][2], o[r][3]};
      *reinterpret_cast<f32x4_t*>(&sm_o[slot][r][c * 8 + 4]) = f32x4_t{o[r][4], o[r][5], o[r][6], o[r][7]};
      if (c == 0) { sm_ml[slot][r][0] = mx[r]; sm_ml[slot][r][1] = ls[r]; }
    }
  }
  __syncthreads();
  for (int idx = tid; idx < ROWS * HD; idx += 256) {
    const int r = idx >> 7, d = idx & (HD - 1);
    const int g = r / a.M, m = r % a.M;
    if (g >= G) continue;
    float mm = -INFINITY;
#pragma unroll
    for (int w = 0; w < NPART; ++w) mm = fmaxf(mm, sm_ml[w][r][0]);
    float num = 0.f, den = 0.f;
#pragma unroll
    for (int w = 0; w < NPART; ++w) {
      const float f = sm_ml[w][r][0] == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(sm_ml[w][r][0] - mm);
      num += sm_o[w][r][d] * f;
      den += sm_ml[w][r][1] * f;
    }
    float* pp = a.part + ((((int64_t)b * a.H + kvh * G + g) * a.M + m) * a.nsplit + split) * DEC_PART;
    pp[d] = num;
    if (d == 0) { pp[128] = mm; pp[129] = den; }
  }
}

// o[b, m, h, :] = sum_i o_i 2^(m_i - M) / sum_i l_i 2^(m_i - M) over the nparts partials of (b, h, m); a row without any allowed key
// comes out NaN, as SDPA's softmax of an all -inf row does.  One block per (b, h, m): the (m_i, l_i) pairs go through LDS, then every
// thread sums its output dim over the partials with independent loads.
__global__ __launch_bounds__(512) void attn_decode_combine_kernel(const float* __restrict__ part, int nparts, bf16_t* __restrict__ o, int64_t o_sb, int64_t o_sh,
                                                                  int64_t o_ss, int H, int M) {
  // 4 groups of 128 threads share the partials of one (b, h, m) (with 64 of them a single group's serial loop was most of this
  // 32-block kernel's 7.6 us); every group sums its quarter in partial order, the quarters are added in group order
  __shared__ float sf[1024], sl[1024], snum[4][HD], sden[4];
  const int d = threadIdx.x & 127, grp = threadIdx.x >> 7;
  const int m = blockIdx.x % M, h = (blockIdx.x / M) % H, b = blockIdx.x / (M * H);
  const float* pp = part + (int64_t)blockIdx.x * nparts * DEC_PART;
  for (int i = threadIdx.x; i < nparts; i += 512) { sf[i] = pp[i * DEC_PART + 128]; sl[i] = pp[i * DEC_PART + 129]; }
  __syncthreads();
  float mm = -INFINITY;
  for (int i = 0; i < nparts; ++i) mm = fmaxf(mm, sf[i]);
  const int per = (nparts + 3) / 4, i0 = grp * per, i1 = min(nparts, i0 + per);
  float num = 0.f, den = 0.f;
#pragma unroll 8
  for (int i = i0; i < i1; ++i) {
    const float f = sf[i] == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(sf[i] - mm);
    num += pp[i * DEC_PART + d] * f;
    den += sl[i] * f;
  }
  snum[grp][d] = num;
  if (d == 0) sden[grp] = den;
  __syncthreads();
  if (grp == 0) {
    const float n4 = ((snum[0][d] + snum[1][d]) + snum[2][d]) + snum[3][d];
    const float d4 = ((sden[0] + sden[1]) + sden[2]) + sden[3];
    o[b * o_sb + h * o_sh + (int64_t)m * o_ss + d] = f2bf(n4 / d4);
  }
}

extern "C" int64_t llx_attn_decode_workspace_bytes(int64_t B, int64_t H, int64_t M, int64_t nsplit) { return B * H * M * nsplit * DEC_PART * 4; }

// SDPA(q, k_cache, v_cache, mask, is_causal=False, enable_gqa=True) for a few query tokens (M * H / KVH <= 16) against the whole cache
// (modelling/llama.py:126-127,135-137).  q [B,H,M,128], caches [B,KVH,Skv,128], o [B,H,M,128] through (batch, head, position) element
// strides; mask bool [.., M, Skv] with broadcast strides; extent (nullable device int, llx_mask_extent): bound on the keys any row
// may attend to; workspace: llx_attn_decode_workspace_bytes(B, H, M, nsplit) bytes of fp32.
extern "C" int llx_attn_decode(const void* q, int64_t q_sb, int64_t q_sh, int64_t q_ss, const void* k_cache, const void* v_cache, int64_t c_sb,
                               int64_t c_sh, int64_t c_ss, void* o, int64_t o_sb, int64_t o_sh, int64_t o_ss, const void* mask, int64_t m_sb,
                               int64_t m_sh, int64_t m_sq, const int* extent, float* workspace, int64_t B, int64_t H, int64_t KVH, int64_t M,
                               int64_t Skv, int64_t nsplit, int64_t head_dim, float scale, hipStream_t stream) {
  LLX_REQUIRE(q && k_cache && v_cache && o && mask && workspace, "llx_attn_decode: null pointer");
  LLX_REQUIRE(head_dim == HD, "llx_attn_decode: head_dim=%lld unsupported (only 128)", (long long)head_dim);
  LLX_REQUIRE(B > 0 && H > 0 && KVH > 0 && H % KVH == 0 && M > 0 && Skv > 0 && nsplit > 0 && nsplit <= 1024 && B < 65536 && KVH < 65536, "llx_attn_decode: bad sizes");
  const int64_t rows = H / KVH * M;
  LLX_REQUIRE(rows <= 16, "llx_attn_decode: %lld query rows per kv head (heads per group x tokens) exceed 16: use llx_attn_dense_fwd", (long long)rows);
  LLX_REQUIRE(((q_sb | q_sh | q_ss | c_sb | c_sh | c_ss) % 8) == 0 && ((uintptr_t)q | (uintptr_t)k_cache | (uintptr_t)v_cache) % 16 == 0 && (uintptr_t)workspace % 16 == 0,
              "llx_attn_decode: rows must be 16-byte aligned");
  DecodeArgs a;
  a.q = (const bf16_t*)q; a.q_sb = q_sb; a.q_sh = q_sh; a.q_ss = q_ss;
  a.kc = (const bf16_t*)k_cache; a.vc = (const bf16_t*)v_cache; a.c_sb = c_sb; a.c_sh = c_sh; a.c_ss = c_ss;
  a.mask = (const uint8_t*)mask; a.m_sb = m_sb; a.m_sh = m_sh; a.m_sq = m_sq; a.extent = extent; a.part = workspace;
  a.B = (int)B; a.H = (int)H; a.KVH = (int)KVH; a.M = (int)M; a.Skv = (int)Skv; a.nsplit = (int)nsplit;
  a.scale_log2 = scale * 1.4426950408889634f;
  const dim3 grid((unsigned)nsplit, (unsigned)KVH, (unsigned)B);
  if (rows <= 4) hipLaunchKernelGGL((attn_decode_kernel<4, 4>), grid, dim3(256), 0, stream, a);
  else if (rows <= 8) hipLaunchKernelGGL((attn_decode_kernel<8, 2>), grid, dim3(256), 0, stream, a);
  else hipLaunchKernelGGL((attn_decode_kernel<16, 1>), grid, dim3(256), 0, stream, a);
  LLX_LAUNCH_CHECK("llx_attn_decode");
  hipLaunchKernelGGL(attn_decode_combine_kernel, dim3((unsigned)(B * H * M)), dim3(512), 0, stream, (const float*)workspace, (int)nsplit, (bf16_t*)o, o_sb,
                     o_sh, o_ss, (int)H, (int)M);
  LLX_LAUNCH_CHECK("llx_attn_decode(combine)");
  return LLX_OK;
}
