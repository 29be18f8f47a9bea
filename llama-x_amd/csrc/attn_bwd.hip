// Flash-style attention backward (head_dim 128, GQA, same mask rule as attn_fwd.hip).
//
// Gradients of softmax(QK^T*scale + mask)V as autograd produces them for the reference's SDPA / FlexAttention
// call (modelling/llama.py:129-137).  P is recomputed from Q, K and the forward's log2-sum-exp.  Deterministic:
// no atomics.  Two routes:
// (a) with a dS buffer (llx_attn_bwd's `ds`, llx_attn_bwd_ds_bytes): five products, each computed once
//   0. attn_bwd_delta_kernel: delta[b,h,q] = sum_d dO.O and the sanitised -lse
//   1. attn_bwd_dkv3_kernel<.., DS>: one workgroup = 128 keys x ONE query head, sweeps 64-row query tiles: S, dP, dV, dK - and
//      stores dS^T (bf16, the very values its dK product consumes) as [key][query] rows                -> fp32 partial dK, dV; dS^T
//   2. attn_dkv_reduce_kernel: sums the G per-head partials of a KV group                             -> dK, dV
//   3. attn_bwd_dq2_kernel: dQ^T = K^T . dS^T as a plain tiled product over the stored dS^T (HBM-bound: the buffer is read once;
//      the S and dP products of the stand-alone dQ kernel are gone)                                   -> dQ
//   A one-pass kernel that also accumulates dQ needs a cross-workgroup sum of 0.57 GB of fp32 per layer at S = 4096 (438 us at the
//   1.3 TB/s float-atomic rate; an ordered hand-off needs every key block of a head co-resident, 32 of them at B = 1): the bf16
//   dS^T round trip is 0.54 GB written + read at the plain HBM rate and keeps the result deterministic.
// (b) without the buffer: seven products in two kernels
//   1. attn_bwd_dq_kernel: one workgroup = 128 query rows of one head, sweeps key tiles      -> dQ, and delta[b,h,q] = sum_d dO.O
//   2. attn_bwd_dkv3_kernel: as above without the dS^T store                                 -> fp32 partial dK, dV
//   3. attn_dkv_reduce_kernel
// MFMA orientation keeps the softmax row index where the row constants (lse, delta) are cheap:
//   dq kernel : S^T = K.Q^T, dP^T = V.dO^T (query on the lane), dQ^T += K^T.dS^T with dS^T taken from the accumulator
//               registers as the B operand and K^T read from the SAME LDS image by ds_read_b64_tr_b16.
//   dkv kernel: S = Q.K^T, dP = dO.V^T (key on the lane, K/V fragments live in registers), dV^T += dO^T.P,
//               dK^T += Q^T.dS with P/dS from the accumulator registers and Q^T/dO^T by transposed LDS reads.
// LDS images read both by rows and transposed use the dual-use swizzle  slot = chunk ^ (((row&3)<<2) | ((row>>2)&3)).
#include "common.h"
#include <mutex>
#include <stdlib.h>
#include <type_traits>

#define HD 128
#define BQ 128
#define BKV 64
#define TILE_BYTES (64 * HD * 2)  // a 64-row x 128-col bf16 tile = 16 KiB

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;
typedef __attribute__((address_space(3))) s16x4_t lds_s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;

__device__ __forceinline__ int dual_swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }
__device__ __forceinline__ uint32_t dual_swz(uint32_t row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

// A-operand fragment (32 rows x 16 k) read TRANSPOSED from a dual-use [rows][128] bf16 LDS image:
// returns X^T[col = 32*db + (lane&31)][row = row0 + 8*(j>>2) + 4*hh + (j&3)], j = 0..7  (row0 multiple of 16).
__device__ __forceinline__ bf16x8_t tr_frag(const char* img, int row0, int db, int lane) {
  const int hh = lane >> 5, tq = (lane & 15) >> 2, tp = lane & 3, tsub = (lane >> 4) & 1;
  const int chunk = 4 * db + 2 * tsub + (tp >> 1);
  const int rlo = row0 + 4 * hh + tq, rhi = rlo + 8;
  s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + rlo * 256 + ((chunk ^ dual_swz(rlo)) << 4) + ((tp & 1) << 3)));
  s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + rhi * 256 + ((chunk ^ dual_swz(rhi)) << 4) + ((tp & 1) << 3)));
  s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, v);
}

// Row fragment (A operand rows = image rows, k = 16*ks + 8*hh + j) from a dual-use image.
__device__ __forceinline__ bf16x8_t row_frag(const char* img, int row, int ks, int hh) {
  return *reinterpret_cast<const bf16x8_t*>(img + row * 256 + (((2 * ks + hh) ^ dual_swz(row)) << 4));
}

// (the 10-destination wait of the dQ-from-dS kernel; lds_tr_read / frag_of: common.h)
template <int N>
__device__ __forceinline__ void lds_tr_wait(s16x4_t& a, s16x4_t& b, s16x4_t (&c)[4], s16x4_t (&d)[4]) {
  asm volatile("s_waitcnt lgkmcnt(%10)"
               : "+v"(a), "+v"(b), "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3])
               : "n"(N));
  __builtin_amdgcn_sched_barrier(0);
}
// a ^ c issued where it is written: as plain C++ hipcc computes all the XORed addresses of a tile up front and keeps them live
__device__ __forceinline__ uint32_t xor_imm(uint32_t a, int c) {
  if (c == 0) return a;
  uint32_t r;
  asm volatile("v_xor_b32 %0, %1, %2" : "=v"(r) : "n"(c), "v"(a));
  return r;
}

struct AttnBwdArgs {
  const bf16_t* q; const bf16_t* k; const bf16_t* v; const bf16_t* o; const bf16_t* d_o;
  const float* lse; float* delta; float* nlse;
  const float* rope;  // nullable: fp32 table [>= S, 64, 2]; dq and dk leave the kernels already rotated by -theta (apply_rope's transpose)
  unsigned long long* stamps;  // diagnostic (normally null): s_memtime stamps of workgroup 0, wave 0 of the dK/dV kernel
  bf16_t* dq; bf16_t* dk; bf16_t* dv;
  bf16_t* ds; int Sp;  // route (a): dS^T per (b, h) as [Sp/64 key tiles][Sp/128 query blocks] tiles of 64 keys x 128 queries bf16 (16 KiB, each
                       // written by one key-block workgroup and read by one query-block workgroup), Sp = S rounded up to 256.  Inside a
                       // tile the 16-byte chunks (8 queries of one key) are ordered as the dK/dV kernel emits them, so that each of its
                       // store instructions writes 1 KiB contiguously: chunk(key, q) = ((((kh*2 + qh)*2 + qb32)*2 + s)*2 + hh)*32 + r with
                       // key = 32 kh + r, q = 64 qh + 32 qb32 + 16 s + 8 hh + (0..7)
  int64_t q_sb, q_ss, k_sb, k_ss, v_sb, v_ss, o_sb, o_ss, do_sb, do_ss;
  int64_t dq_sb, dq_ss, dk_sb, dk_ss, dv_sb, dv_ss;
  const int* doc_ids; const int* prefix_len; const uint8_t* flags;
  int B, S, H, KVH;
  float scale, scale_log2;
};

// ------------------------------------------------------------------------------------------ dQ
#define DQ_STAGE_BYTES (2 * TILE_BYTES)
#define DQ_LDS_BYTES (2 * DQ_STAGE_BYTES)

template <bool GENERAL>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(const AttnBwdArgs a) {
  extern __shared__ __attribute__((aligned(256))) char smem[];  // 256-aligned: the transposed-read addresses flip bits 5-7 by XOR
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nqb = (a.S + BQ - 1) / BQ, nkt = (a.S + BKV - 1) / BKV;
  const int qb = nqb - 1 - blockIdx.y;  // q-block = slow dispatch dimension: heaviest blocks of every head first
  const int h = blockIdx.x, b = blockIdx.z;
  const int kvh = h / (a.H / a.KVH);
  const int r = lane & 31, hh = lane >> 5;
  const int qi = qb * BQ + wave * 32 + r;
  const int qrow = min(qi, a.S - 1);

  const uint8_t* fl = GENERAL ? a.flags + ((int64_t)b * nqb + qb) * nkt : nullptr;
  const int kt_end = GENERAL ? nkt : min(nkt, (qb * BQ + BQ + BKV - 1) / BKV);
  // (general masks: the flags of 64 tiles at a time in one register, read by v_readlane - not a dependent byte load per tile)
  uint32_t fpack = 0;
  int fchunk = -1;
  auto tile_class = [&](int t) -> int {
    if constexpr (GENERAL) {
      if ((t >> 6) != fchunk) {
        fchunk = t >> 6;
        fpack = fl[min(fchunk * 64 + lane, nkt - 1)];
      }
      return __builtin_amdgcn_readlane((int)fpack, t & 63);
    } else {
      return (t * BKV + BKV - 1 <= qb * BQ) ? 2 : 1;
    }
  };
  auto next_tile = [&](int t) {
    while (t < kt_end && tile_class(t) == 0) ++t;
    return t;
  };

  const int srow_in = lane >> 4, sslot = lane & 15;
  const bf16_t* kbase = a.k + (int64_t)b * a.k_sb + kvh * HD;
  const bf16_t* vbase = a.v + (int64_t)b * a.v_sb + kvh * HD;
  uint32_t koff[4], voff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = i * 16 + wave * 4 + srow_in;
    koff[i] = (uint32_t)(((int64_t)row * a.k_ss + (sslot ^ dual_swz(row)) * 8) * 2);  // K: dual-use image (row reads + transposed reads)
    voff[i] = (uint32_t)(((int64_t)row * a.v_ss + (sslot ^ (row & 15)) * 8) * 2);     // V: row reads only
  }
  auto stage = [&](int buf, int t) {
    char* sK = smem + buf * DQ_STAGE_BYTES;
    char* sV = sK + TILE_BYTES;
    if (t * BKV + BKV <= a.S) {
      const char* kt = (const char*)(kbase + (int64_t)t * BKV * a.k_ss);
      const char* vt = (const char*)(vbase + (int64_t)t * BKV * a.v_ss);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        __builtin_amdgcn_global_load_lds((gbl_void*)(kt + koff[i]), (lds_void*)(sK + (i * 16 + wave * 4) * 256), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gbl_void*)(vt + voff[i]), (lds_void*)(sV + (i * 16 + wave * 4) * 256), 16, 0, 0);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = i * 16 + wave * 4 + srow_in;
        const int key = min(t * BKV + row, a.S - 1);
        const int kc = sslot ^ dual_swz(row);
        const int vc = sslot ^ (row & 15);
        __builtin_amdgcn_global_load_lds((gbl_void*)(kbase + (int64_t)key * a.k_ss + kc * 8), (lds_void*)(sK + (i * 16 + wave * 4) * 256), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gbl_void*)(vbase + (int64_t)key * a.v_ss + vc * 8), (lds_void*)(sV + (i * 16 + wave * 4) * 256), 16, 0, 0);
      }
    }
  };

  // lane constant of the transposed K reads (element map of tr_frag): byte offset of the lo 4-row block of d-block 0 in the first
  // 16-key step of a 32-key half.  d-block db flips chunk bits 2-3 (byte XOR db << 6); the hi block sits 8 rows further and its
  // swizzle differs in chunk bit 1 (byte XOR 0x20): ONE register instead of eight (this kernel has none to spare)
  const uint32_t sbase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  uint32_t aK0;
  {
    const int tq2 = (lane & 15) >> 2, tp = lane & 3, tsub = (lane >> 4) & 1;
    const int rlo = 4 * hh + tq2;
    aK0 = (uint32_t)(rlo * 256 + (((2 * tsub + (tp >> 1)) ^ dual_swz(rlo)) << 4) + ((tp & 1) << 3));
  }
  f32x16_t dq[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) dq[i][e] = 0.f;
  const int* docrow = (GENERAL && a.doc_ids) ? a.doc_ids + (int64_t)b * a.S : nullptr;
  const int my_doc = docrow ? docrow[qrow] : 0;
  const int my_prefix = (GENERAL && a.prefix_len) ? a.prefix_len[b] : 0;

  int t = next_tile(0);
  if (t < kt_end) stage(0, t);
  // (the rows' own operands are requested AFTER the first K/V tile: one memory round trip for both instead of two in a row)
  // q, dO, O and lse of the rows are requested in ONE batch (hipcc put a full wait between the q / dO loads and the lse / O loads)
  bf16x8_t qf[8], dof[8], ovf[8];
  const float my_lse = a.lse[((int64_t)b * a.H + h) * a.S + qrow];
  {
    const bf16_t* qp = a.q + (int64_t)b * a.q_sb + (int64_t)qrow * a.q_ss + h * HD + 8 * hh;
    const bf16_t* dp = a.d_o + (int64_t)b * a.do_sb + (int64_t)qrow * a.do_ss + h * HD + 8 * hh;
    const bf16_t* op = a.o + (int64_t)b * a.o_sb + (int64_t)qrow * a.o_ss + h * HD + 8 * hh;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      ovf[ks] = *reinterpret_cast<const bf16x8_t*>(op + 16 * ks);
      dof[ks] = *reinterpret_cast<const bf16x8_t*>(dp + 16 * ks);
      qf[ks] = *reinterpret_cast<const bf16x8_t*>(qp + 16 * ks);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  const float lse_safe = (my_lse == -INFINITY) ? 0.f : my_lse;
  // delta = rowsum(dO . O): this kernel holds its rows' dO fragments already, so it computes delta itself (half a row per
  // lane, the partner half-wave holds the other half) and publishes delta and the sanitised -lse for the dK/dV kernel,
  // which runs after it.  (Replaces a separate pass over O and dO.)
  float my_delta = 0.f;
  {
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
#pragma unroll
      for (int j = 0; j < 8; ++j) my_delta += (float)ovf[ks][j] * (float)dof[ks][j];
    }
    my_delta += __shfl_xor(my_delta, 32, 64);
    if (hh == 0 && qi < a.S) {
      a.delta[((int64_t)b * a.H + h) * a.S + qi] = my_delta;
      a.nlse[((int64_t)b * a.H + h) * a.S + qi] = -lse_safe;
    }
  }

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int cur = 0;
  // (document masks: the tile's 64 key ids in one register - lane i holds key 64 t + i - requested one tile ahead and gathered by
  // ds_bpermute, as in the forward kernel)
  int docv = (GENERAL && docrow && t < kt_end) ? docrow[min(t * BKV + lane, a.S - 1)] : 0, docv_next = 0;
  while (t < kt_end) {
    const int tn = next_tile(t + 1);
    if (tn < kt_end) stage(cur ^ 1, tn);
    if (GENERAL && docrow && tn < kt_end) docv_next = docrow[min(tn * BKV + lane, a.S - 1)];
    const char* sK = smem + cur * DQ_STAGE_BYTES;
    const char* sV = sK + TILE_BYTES;
    const int cls = tile_class(t);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      f32x16_t st, dp;
      const f32x16_t zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      const int row = kb * 32 + r;
      // S^T chain first, then the dP^T chain: the exponentials of S are VALU work that can issue under the second chain's MFMAs
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        bf16x8_t kf = row_frag(sK, row, ks, hh);
        st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], ks == 0 ? zero : st, 0, 0, 0);
      }
      if (cls != 2) {  // ONE wave-uniform branch per 32 keys: masked scores become -inf, exp2 turns them into 0
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int kk = t * BKV + kb * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
          bool ok = (kk < a.S) && (kk <= qi || kk < my_prefix);
          if constexpr (GENERAL) {
            const int kd = docrow ? __builtin_amdgcn_ds_bpermute((kk - t * BKV) << 2, docv) : my_doc;
            ok = ok && (kd == my_doc);
          }
          st[e] = ok ? st[e] : -INFINITY;
        }
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) st[e] = __builtin_amdgcn_exp2f(__builtin_fmaf(st[e], a.scale_log2, -lse_safe));
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        bf16x8_t vf = *reinterpret_cast<const bf16x8_t*>(sV + row * 256 + (((2 * ks + hh) ^ (row & 15)) << 4));
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, dof[ks], ks == 0 ? zero : dp, 0, 0, 0);
      }
      // dS^T = P^T * (dP^T - delta)
#pragma unroll
      for (int e = 0; e < 16; ++e) st[e] = st[e] * (dp[e] - my_delta);
      // dQ^T += K^T . dS^T: K^T fragments by transposed reads issued as inline asm (common.h: lds_tr_read - through the builtin hipcc
      // drains the next tile's LDS-DMA right here); the 8 reads of the second 16-key step fly under the MFMAs of the first
      {
        // (the stage / half offsets are multiples of 8192: they do not touch the bits the XORs flip)
        const uint32_t kb_ = sbase + cur * DQ_STAGE_BYTES + kb * 8192 + aK0;
        s16x4_t Kl[4], Kh[4];  // one step's fragments at a time: this kernel has no registers to spare for a second set
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
          for (int db = 0; db < 4; ++db) {
            if (s == 0) { lds_tr_read<0>(Kl[db], xor_imm(kb_, db << 6)); lds_tr_read<2048>(Kh[db], xor_imm(kb_, (db << 6) | 0x20)); }
            else { lds_tr_read<4096>(Kl[db], xor_imm(kb_, db << 6)); lds_tr_read<4096 + 2048>(Kh[db], xor_imm(kb_, (db << 6) | 0x20)); }
          }
          bf16x8_t dsb;
#pragma unroll
          for (int j = 0; j < 8; ++j) dsb[j] = (__bf16)st[8 * s + j];
          lds_tr_wait8<0>(Kl, Kh);
#pragma unroll
          for (int db = 0; db < 4; ++db) dq[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_of(Kl[db], Kh[db]), dsb, dq[db], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    cur ^= 1;
    t = tn;
    docv = docv_next;
  }

  if (qi < a.S) {
    bf16_t* op = a.dq + (int64_t)b * a.dq_sb + (int64_t)qi * a.dq_ss + h * HD;
    // (16-byte stores through a half-wave exchange per pair of column groups, as in the forward kernel's epilogue)
    const bool wide = ((((uintptr_t)a.dq) | (uintptr_t)(a.dq_ss * 2) | (uintptr_t)(a.dq_sb * 2)) & 15) == 0;  // uniform
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int j2 = 0; j2 < 2; ++j2) {
        u32x2_t pp[2];
#pragma unroll
        for (int gg = 0; gg < 2; ++gg) {
          const int g4 = 2 * j2 + gg;
          u32x2_t pk;
          pk[0] = pack_bf2(dq[db][4 * g4 + 0] * a.scale, dq[db][4 * g4 + 1] * a.scale);
          pk[1] = pack_bf2(dq[db][4 * g4 + 2] * a.scale, dq[db][4 * g4 + 3] * a.scale);
          if (a.rope) {  // gradient of apply_rope on the bf16-rounded dq, exactly as the stand-alone rope kernel computes it
            const f32x4_t t = *reinterpret_cast<const f32x4_t*>(a.rope + ((int64_t)qi * 64 + ((32 * db + 8 * g4 + 4 * hh) >> 1)) * 2);
            const float c0 = t[0], s0 = t[1] * -1.f, c1 = t[2], s1 = t[3] * -1.f;
            const float x0 = bflo(pk[0]), x1 = bfhi(pk[0]), y0 = bflo(pk[1]), y1 = bfhi(pk[1]);
            pk[0] = pack_bf2(x0 * c0 - x1 * s0, x1 * c0 + x0 * s0);
            pk[1] = pack_bf2(y0 * c1 - y1 * s1, y1 * c1 + y0 * s1);
          }
          pp[gg] = pk;
        }
        if (wide) {
          const auto r0 = __builtin_amdgcn_permlane32_swap(pp[0][0], pp[1][0], false, false);
          const auto r1 = __builtin_amdgcn_permlane32_swap(pp[0][1], pp[1][1], false, false);
          *reinterpret_cast<u32x4_t*>(op + 32 * db + 16 * j2 + 8 * hh) = u32x4_t{r0[0], r1[0], r0[1], r1[1]};
        } else {
          *reinterpret_cast<u32x2_t*>(op + 32 * db + 16 * j2 + 4 * hh) = pp[0];
          *reinterpret_cast<u32x2_t*>(op + 32 * db + 16 * j2 + 8 + 4 * hh) = pp[1];
        }
      }
  }
}

// ------------------------------------------------------------------------------------------ dK, dV
#define DKV_QT 64
// One workgroup = 4 waves = 128 keys of ONE query head: a wave owns 32 keys, K/V fragments in registers, dK^T/dV^T in 128
// accumulator registers; heaviest key blocks are dispatched first under the causal mask, two workgroups per CU (<= 256
// registers) so one wave's softmax VALU overlaps its SIMD partner's MFMAs.  The G per-head fp32 partials are summed by
// attn_dkv_reduce_kernel (deterministic, no atomics).
#define DKV2_KEYS 128

// Every LDS address is written as  (per-tile lane constant ^ small constant) + immediate.  hipcc cannot see that the swizzled
// addresses of the 8 k-steps / 4 d-blocks differ by an XOR of the low byte; left to itself (the previous version of this kernel)
// it kept ~30 address registers live, spilled them (124 B of scratch) and reloaded them behind `s_waitcnt vmcnt(0)` - which also
// waits for the NEXT tile's LDS-DMA and turns the prefetch synchronous (505 -> 300 us per layer with this form).
// LDS map (bytes, 256-aligned base):
//   stage s: Q image s*0x8000, dO image s*0x8000 + 0x4000;  lse[64] at 0x10000 + s*512, delta[64] at 0x10100 + s*512.
typedef __attribute__((address_space(3))) char lds_char;
typedef __attribute__((address_space(3))) bf16x8_t lds_bf16x8;
typedef __attribute__((address_space(3))) f32x4_t lds_f32x4;
#define DKV3_LDS_BYTES (0x10000 + 1024)

template <bool GENERAL, bool STAMP = false, bool DS = false>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv3_kernel(const AttnBwdArgs a, float* __restrict__ part) {
  extern __shared__ __attribute__((aligned(256))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nqb = (a.S + BQ - 1) / BQ, nkt = (a.S + BKV - 1) / BKV;
  const int nqt = (a.S + DKV_QT - 1) / DKV_QT;
  const int G = a.H / a.KVH;
  int id = blockIdx.x;
  const int per_kb = a.B * a.KVH * G;
  const int kblk = id / per_kb;
  id -= kblk * per_kb;
  const int b = id / (a.KVH * G);
  id -= b * (a.KVH * G);
  const int kvh = id / G, g = id % G;
  const int h = kvh * G + g;
  const int my_kt = 2 * kblk + (wave >> 1);

  // (lane-derived values are not kept across the tile loop: each tile and the epilogue rebuild them from a fresh lane id)
  bf16x8_t kf[8], vf[8];
  const int* docrow = (GENERAL && a.doc_ids) ? a.doc_ids + (int64_t)b * a.S : nullptr;
  int key_doc = 0;
  {
    const int krow = min(kblk * DKV2_KEYS + wave * 32 + (lane & 31), a.S - 1);
    const bf16_t* kp = a.k + (int64_t)b * a.k_sb + (int64_t)krow * a.k_ss + kvh * HD + 8 * (lane >> 5);
    const bf16_t* vp = a.v + (int64_t)b * a.v_sb + (int64_t)krow * a.v_ss + kvh * HD + 8 * (lane >> 5);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      kf[ks] = *reinterpret_cast<const bf16x8_t*>(kp + 16 * ks);
      vf[ks] = *reinterpret_cast<const bf16x8_t*>(vp + 16 * ks);
    }
    if (docrow) key_doc = docrow[krow];
  }
  const int my_prefix = (GENERAL && a.prefix_len) ? a.prefix_len[b] : 0;

  const int qt_first = GENERAL ? 0 : (kblk * DKV2_KEYS) / DKV_QT;
  // (general masks: the flag bytes of this key block's two tiles for 64 query blocks at a time in one register - lane i holds query
  // block 64*chunk + i - read by v_readlane; as byte loads they were three dependent memory round trips per query tile)
  uint32_t fpack = 0;
  int fchunk = -1;
  auto block_class = [&](int qt, int kt) -> int {
    if (kt >= nkt) return 0;
    if constexpr (GENERAL) {
      const int qb128 = qt >> 1;
      if ((qb128 >> 6) != fchunk) {
        fchunk = qb128 >> 6;
        uint32_t ln;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
        const uint8_t* fr = a.flags + ((int64_t)b * nqb + min(fchunk * 64 + (int)ln, nqb - 1)) * nkt;
        fpack = (uint32_t)fr[2 * kblk] | ((2 * kblk + 1 < nkt ? (uint32_t)fr[2 * kblk + 1] : 0u) << 8);
      }
      const uint32_t w = (uint32_t)__builtin_amdgcn_readlane((int)fpack, qb128 & 63);
      return (int)((w >> (8 * (kt - 2 * kblk))) & 0xff);
    }
    const int q_lo = qt * DKV_QT, q_hi = q_lo + DKV_QT - 1, k_lo = kt * BKV, k_hi = k_lo + BKV - 1;
    if (k_lo > q_hi) return 0;
    return (k_hi <= q_lo) ? 2 : 1;
  };
  auto next_qt = [&](int qt) {
    while (qt < nqt && !(qt >= qt_first && (block_class(qt, 2 * kblk) != 0 || block_class(qt, 2 * kblk + 1) != 0))) ++qt;
    return qt;
  };

  const uint32_t sbase = (uint32_t)(uintptr_t)(lds_char*)smem;

  // staging: thread -> rows i*16 + wave*4 + (lane>>4), 16-byte slot lane&15; the swizzle of those rows does not depend on i.
  // Lane offsets are rebuilt from a fresh lane id per call for the same reason as the read constants.
  const char* qbase = (const char*)(a.q + (int64_t)b * a.q_sb + h * HD);
  const char* dbase = (const char*)(a.d_o + (int64_t)b * a.do_sb + h * HD);
  const float* lbase = (wave == 0 ? a.nlse : a.delta) + ((int64_t)b * a.H + h) * a.S;
  // One LDS-DMA piece = 1 KiB = 4 rows of one image per wave-instruction: 4 row groups x {Q, dO} per wave, plus the row
  // statistics (waves 0 and 1).  The lane offsets are rebuilt once per call from a fresh lane id (kept across the loop they would
  // be spilled).  (Spreading the pieces between the MFMAs of the first S chain instead of bursting them at the top of the tile was
  // tried: hipcc then spills ~120 registers; the burst costs ~1.4k cycles per tile.)
  auto stage = [&](int buf, int qt) {
    uint32_t ln;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
    const uint32_t srow = wave * 4 + (ln >> 4);
    const uint32_t sc = (ln & 15) ^ dual_swz(srow);
    lds_char* sQ = (lds_char*)smem + buf * 0x8000 + wave * 1024;
    if (qt * DKV_QT + DKV_QT <= a.S) {
      const uint32_t q_lane = (srow * (uint32_t)a.q_ss + sc * 8) * 2, d_lane = (srow * (uint32_t)a.do_ss + sc * 8) * 2;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const char* qu = qbase + (int64_t)(qt * DKV_QT + i * 16) * a.q_ss * 2;   // wave-uniform
        const char* du = dbase + (int64_t)(qt * DKV_QT + i * 16) * a.do_ss * 2;
        __builtin_amdgcn_global_load_lds((gbl_void*)(qu + q_lane), (lds_void*)(sQ + i * 4096), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gbl_void*)(du + d_lane), (lds_void*)(sQ + 0x4000 + i * 4096), 16, 0, 0);
      }
    } else {  // ragged last tile: clamp the row per lane
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int qr = min(qt * DKV_QT + i * 16 + (int)srow, a.S - 1);
        __builtin_amdgcn_global_load_lds((gbl_void*)(qbase + ((int64_t)qr * a.q_ss + sc * 8) * 2), (lds_void*)(sQ + i * 4096), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gbl_void*)(dbase + ((int64_t)qr * a.do_ss + sc * 8) * 2), (lds_void*)(sQ + 0x4000 + i * 4096), 16, 0, 0);
      }
    }
    if (wave < 2) {  // wave 0: -lse[64] (sanitised), wave 1: delta[64]
      const float* src = lbase + min(qt * DKV_QT + (int)ln, a.S - 1);
      __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)((lds_char*)smem + 0x10000 + buf * 512 + wave * 256), 4, 0, 0);
    }
  };

  f32x16_t dk[4], dv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) { dk[i][e] = 0.f; dv[i][e] = 0.f; }

  int qt = next_qt(0);
  // (document masks: the tile's 64 query-row ids in one register - lane i holds row 64 qt + i - requested one tile ahead, gathered by
  // ds_bpermute where a partly masked block needs them)
  int qdoc = 0, qdoc_next = 0;
  if (GENERAL && docrow && qt < nqt) {
    uint32_t l0;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l0));
    qdoc = docrow[min(qt * DKV_QT + (int)l0, a.S - 1)];
  }
  if (qt < nqt) stage(0, qt);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int cur = 0;
  int nst = 0;
  auto stamp = [&]() {
    if constexpr (STAMP) {
      if (blockIdx.x == 0 && wave == 0 && nst < 1024) {
        unsigned long long tt;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tt) :: "memory");
        if (threadIdx.x == 0) a.stamps[nst] = tt;
        ++nst;
      }
    }
  };
  while (qt < nqt) {
    stamp();  // 0: tile start
    const int qtn = next_qt(qt + 1);
    const bool have_next = qtn < nqt;
    // lane constants of the three LDS read patterns, recomputed per tile from a fresh lane id (kept live across the loop
    // they are spilled; element maps as in tr_frag / row_frag)
    uint32_t ln;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
    const uint32_t lr_ = ln & 31, lh_ = ln >> 5;
    const uint32_t Lr = sbase + cur * 0x8000 + lr_ * 256 + ((lh_ ^ dual_swz(lr_)) << 4);
    const uint32_t tq = (ln & 15) >> 2, tp = ln & 3, tsub = (ln >> 4) & 1, trow = 4 * lh_ + tq;
    const uint32_t Tl = sbase + cur * 0x8000 + trow * 256 + (((2 * tsub + (tp >> 1)) ^ dual_swz(trow)) << 4) + ((tp & 1) << 3);
    const uint32_t Th = sbase + cur * 0x8000 + (trow + 8) * 256 + (((2 * tsub + (tp >> 1)) ^ dual_swz(trow + 8)) << 4) + ((tp & 1) << 3);
    const uint32_t Ls = sbase + 0x10000 + cur * 512 + lh_ * 16;
    const int hh = (int)lh_;
    const int key = kblk * DKV2_KEYS + wave * 32 + (int)lr_;
    int cls = block_class(qt, my_kt);
    if (cls == 2 && (qt * DKV_QT + DKV_QT > a.S)) cls = 1;
    if (have_next) stage(cur ^ 1, qtn);
    if (GENERAL && docrow && have_next) qdoc_next = docrow[min(qtn * DKV_QT + (int)ln, a.S - 1)];
    stamp();  // 1: next tile's DMA issued
    if (cls != 0) {
      auto rowf = [&](int qb32, int img, int ks) -> bf16x8_t {
        return *(lds_bf16x8*)((lds_char*)(uintptr_t)xor_imm(Lr, ks << 5) + qb32 * 8192 + img * 0x4000);
      };
      // transposed fragments as inline asm reads (common.h: lds_tr_read - through the builtin hipcc drains the next tile's LDS-DMA in
      // front of the first transposed read of every half tile); step = (s2, db); the lo / hi 4-row blocks land in tl / th
      auto trf = [&](int qb32, int img, int step, s16x4_t& tl, s16x4_t& th) {
        const int s2 = step >> 2, db = step & 3;
        lds_tr_read_rt(tl, xor_imm(Tl, db << 6), (qb32 * 32 + s2 * 16) * 256 + img * 0x4000);
        lds_tr_read_rt(th, xor_imm(Th, db << 6), (qb32 * 32 + s2 * 16) * 256 + img * 0x4000);
      };
#pragma unroll
      for (int qb32 = 0; qb32 < 2; ++qb32) {
        f32x16_t st, dp;
        const f32x16_t zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        // ---- S = Q.K^T: fragments two k-steps ahead (sched_barrier pins the order; hipcc counts the lgkmcnt ladder)
        bf16x8_t f3[3];
        f3[0] = rowf(qb32, 0, 0);
        f3[1] = rowf(qb32, 0, 1);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
          if (ks + 2 < 8) f3[(ks + 2) % 3] = rowf(qb32, 0, ks + 2);
          st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f3[ks % 3], kf[ks], ks == 0 ? zero : st, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (qb32 == 0) stamp();  // 2: S chain
        // first dO fragments fly under the softmax
        bf16x8_t g3[3];
        g3[0] = rowf(qb32, 1, 0);
        g3[1] = rowf(qb32, 1, 1);
        if (cls != 2) {  // ONE wave-uniform branch per 32 query rows: masked scores become -inf, exp2 turns them into 0
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int qi = qt * DKV_QT + qb32 * 32 + 8 * (e >> 2) + 4 * hh + (e & 3);
            bool ok = (qi < a.S) && (key < a.S) && (key <= qi || key < my_prefix);
            if constexpr (GENERAL) {
              const int qd = docrow ? __builtin_amdgcn_ds_bpermute((qi - qt * DKV_QT) << 2, qdoc) : key_doc;
              ok = ok && (qd == key_doc);
            }
            st[e] = ok ? st[e] : -INFINITY;
          }
        }
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          // stats hold -lse with -inf rows sanitised to 0 (written by attn_bwd_dq_kernel)
          const f32x4_t l4 = *(lds_f32x4*)((lds_char*)(uintptr_t)Ls + (qb32 * 32 + 8 * g4) * 4);
#pragma unroll
          for (int e2 = 0; e2 < 4; ++e2) st[4 * g4 + e2] = __builtin_amdgcn_exp2f(__builtin_fmaf(st[4 * g4 + e2], a.scale_log2, l4[e2]));
        }
        __builtin_amdgcn_sched_barrier(0);
        if (qb32 == 0) stamp();  // 3: softmax
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
          if (ks + 2 < 8) g3[(ks + 2) % 3] = rowf(qb32, 1, ks + 2);
          dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g3[ks % 3], vf[ks], ks == 0 ? zero : dp, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (qb32 == 0) stamp();  // 4: dP chain
        // first transposed fragments of the second phase fly under the dS arithmetic
        s16x4_t tdl[2], tdh[2], tql[2], tqh[2];
        trf(qb32, 1, 0, tdl[0], tdh[0]);
        trf(qb32, 0, 0, tql[0], tqh[0]);
        // dS = P (dP - delta); P and dS are packed to bf16 for BOTH 16-row k-steps before the second phase starts, so that
        // phase holds 16 operand registers instead of the 32 fp32 ones (the register peak of this kernel)
        bf16x8_t pb[2], dsb[2];
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const f32x4_t d4 = *(lds_f32x4*)((lds_char*)(uintptr_t)Ls + 256 + (qb32 * 32 + 8 * g4) * 4);
#pragma unroll
          for (int e2 = 0; e2 < 4; ++e2) {
            const int e = 4 * g4 + e2;
            pb[e >> 3][e & 7] = (__bf16)st[e];
            dsb[e >> 3][e & 7] = (__bf16)(st[e] * (dp[e] - d4[e2]));
          }
        }
        if constexpr (DS) {
          // dS^T[key][query] for the dQ product: this lane's key row, the 16 query rows of k-step s.  A half-wave exchange turns the
          // two 8-byte pieces per lane (rows 16s+4hh.., 16s+8+4hh..) into ONE 16-byte store: lanes 0-31 rows 16s..16s+7, lanes 32-63
          // rows 16s+8..16s+15
          bf16_t* dtile = a.ds + ((int64_t)b * a.H + h) * a.Sp * a.Sp + (int64_t)(my_kt * (a.Sp >> 7) + (qt >> 1)) * (64 * 128);
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) {
            const u32x4_t w = __builtin_bit_cast(u32x4_t, dsb[s2]);
            const auto r0 = __builtin_amdgcn_permlane32_swap(w[0], w[2], false, false);
            const auto r1 = __builtin_amdgcn_permlane32_swap(w[1], w[3], false, false);
            const int piece = ((((wave & 1) * 2 + (qt & 1)) * 2 + qb32) * 2 + s2) * 64;  // 64 chunks = 1 KiB per store instruction
            *reinterpret_cast<u32x4_t*>(dtile + (piece + (int)ln) * 8) = u32x4_t{r0[0], r1[0], r0[1], r1[1]};
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (qb32 == 0) stamp();  // 5: dS + pack
        // ---- dV^T += dO^T.P, dK^T += Q^T.dS: step = (s2, db), the transposed fragments of the next step are in flight
#pragma unroll
        for (int step = 0; step < 8; ++step) {
          if (step + 1 < 8) {
            trf(qb32, 1, step + 1, tdl[(step + 1) & 1], tdh[(step + 1) & 1]);
            trf(qb32, 0, step + 1, tql[(step + 1) & 1], tqh[(step + 1) & 1]);
          }
          // this step's 4 reads have landed once all but the next step's 4 are done (LDS returns in order)
          if (step + 1 < 8) lds_tr_wait4<4>(tdl[step & 1], tdh[step & 1], tql[step & 1], tqh[step & 1]);
          else lds_tr_wait4<0>(tdl[step & 1], tdh[step & 1], tql[step & 1], tqh[step & 1]);
          dv[step & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_of(tdl[step & 1], tdh[step & 1]), pb[step >> 2], dv[step & 3], 0, 0, 0);
          dk[step & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_of(tql[step & 1], tqh[step & 1]), dsb[step >> 2], dk[step & 3], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (qb32 == 0) stamp();  // 6: 16 + 16 MFMAs of the second phase (first 32 rows)
      }
    }
    stamp();  // 7: second 32 rows
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stamp();  // 8: DMA wait
    __syncthreads();
    stamp();  // 9: barrier
    cur ^= 1;
    qt = qtn;
    qdoc = qdoc_next;
  }

  const int64_t plane = (int64_t)a.B * a.S * a.KVH * HD;
  uint32_t le;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(le));
  const int hh = (int)(le >> 5);
  const int key = kblk * DKV2_KEYS + wave * 32 + (int)(le & 31);
  if (key < a.S) {
    float* pk = part + (int64_t)(g * 2 + 0) * plane + (((int64_t)b * a.S + key) * a.KVH + kvh) * HD;
    float* pv = part + (int64_t)(g * 2 + 1) * plane + (((int64_t)b * a.S + key) * a.KVH + kvh) * HD;
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int d = 32 * db + 8 * g4 + 4 * hh;
        *reinterpret_cast<f32x4_t*>(pk + d) = f32x4_t{dk[db][4 * g4], dk[db][4 * g4 + 1], dk[db][4 * g4 + 2], dk[db][4 * g4 + 3]};
        *reinterpret_cast<f32x4_t*>(pv + d) = f32x4_t{dv[db][4 * g4], dv[db][4 * g4 + 1], dv[db][4 * g4 + 2], dv[db][4 * g4 + 3]};
      }
  }
}

// ------------------------------------------------------------------------------------------ route (a): delta, dQ from stored dS^T
// delta[b,h,q] = sum_d dO[b,q,h,d] * O[b,q,h,d] (fp32) and nlse = -lse with -inf rows sanitised to 0: the row constants of the
// backward.  16 lanes per (row, head): one 16-byte chunk of O and dO each, butterfly over the 16 lanes.
__global__ __launch_bounds__(256) void attn_bwd_delta_kernel(const AttnBwdArgs a) {
  const int64_t idx = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);  // (b, q, h) flattened with h fastest
  const int c = threadIdx.x & 15;
  const int64_t total = (int64_t)a.B * a.S * a.H;
  const int64_t i = idx < total ? idx : total - 1;
  const int h = (int)(i % a.H);
  const int64_t bq = i / a.H;
  const int q = (int)(bq % a.S), b = (int)(bq / a.S);
  const u32x4_t ov = *reinterpret_cast<const u32x4_t*>(a.o + (int64_t)b * a.o_sb + (int64_t)q * a.o_ss + h * HD + c * 8);
  const u32x4_t dv = *reinterpret_cast<const u32x4_t*>(a.d_o + (int64_t)b * a.do_sb + (int64_t)q * a.do_ss + h * HD + c * 8);
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) s += bflo(ov[e]) * bflo(dv[e]) + bfhi(ov[e]) * bfhi(dv[e]);
  s += __shfl_xor(s, 1, 64);
  s += __shfl_xor(s, 2, 64);
  s += __shfl_xor(s, 4, 64);
  s += __shfl_xor(s, 8, 64);
  if (c == 0 && idx < total) {
    const int64_t o = ((int64_t)b * a.H + h) * a.S + q;
    const float l = a.lse[o];
    a.delta[o] = s;
    a.nlse[o] = (l == -INFINITY) ? 0.f : -l;
  }
}

// dQ^T[d][q] = sum_key K^T[d][key] . dS^T[key][q]: one workgroup = 8 waves = 256 query rows of one head (a wave owns 32), sweeping the
// key tiles its mask class allows.  Both operands sit in LDS and are read TRANSPOSED (ds_read_b64_tr_b16): the contraction index (key)
// is the row for both.  Measured on the 128-row predecessor of this kernel (timing probes: no MFMA / LDS reads, every dS^T tile from one
// address, 2 .. 5 ring stages, one or two workgroups per CU - all within 1 %): the time is the CU's vector-memory delivery, ~24 GB/s per
// CU for the LDS-DMA fills whatever their source, so the bytes per query row decide: a K tile (16 KiB, from L2) is now shared by 256
// rows (48 KiB per tile step instead of 64 KiB per 256 rows), the dS^T tiles (2 x 16 KiB, adjacent in the buffer) stream from HBM once.
#define DQ2_BQ 256
#define DQ2_STAGE_BYTES (3 * TILE_BYTES)  // K tile + two dS^T tiles
#define DQ2_STAGES 3
#define DQ2_LDS_BYTES (DQ2_STAGES * DQ2_STAGE_BYTES)

template <bool GENERAL>
__global__ __launch_bounds__(512, 2) void attn_bwd_dq2_kernel(const AttnBwdArgs a) {
  extern __shared__ __attribute__((aligned(256))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nqb = (a.S + BQ - 1) / BQ, nkt = (a.S + BKV - 1) / BKV;  // 128-row blocks (the granularity of the tile flags), key tiles
  const int nqb2 = (a.S + DQ2_BQ - 1) / DQ2_BQ;
  const int qb2 = nqb2 - 1 - blockIdx.y;  // heaviest blocks of every head first
  const int h = blockIdx.x, b = blockIdx.z;
  const int kvh = h / (a.H / a.KVH);
  const int r = lane & 31, hh = lane >> 5;
  const int qi = qb2 * DQ2_BQ + wave * 32 + r;
  const int my_qb = min(2 * qb2 + (wave >> 2), nqb - 1);  // the 128-row block of this wave

  const int kt_end = GENERAL ? nkt : min(nkt, (qb2 * DQ2_BQ + DQ2_BQ + BKV - 1) / BKV);
  // the schedule (tile classes are scalar data): built once, before any LDS-DMA is in flight - a flag load inside the loop would make
  // the compiler drain the whole LDS-DMA pipeline at every tile.  GENERAL: the non-empty key tiles of the two 128-row blocks as 64-bit
  // masks in scalar registers (route (a) is taken for up to 128 key tiles); fa = block of waves 0-3, fb = waves 4-7
  unsigned long long fa0 = 0, fa1 = 0, fb0 = 0, fb1 = 0;
  if constexpr (GENERAL) {
    const uint8_t* fla = a.flags + ((int64_t)b * nqb + min(2 * qb2, nqb - 1)) * nkt;
    const uint8_t* flb = a.flags + ((int64_t)b * nqb + min(2 * qb2 + 1, nqb - 1)) * nkt;
    const bool has_b = 2 * qb2 + 1 < nqb;
    fa0 = __builtin_amdgcn_ballot_w64(lane < nkt && fla[min(lane, nkt - 1)] != 0);
    fa1 = __builtin_amdgcn_ballot_w64(64 + lane < nkt && fla[min(64 + lane, nkt - 1)] != 0);
    fb0 = __builtin_amdgcn_ballot_w64(has_b && lane < nkt && flb[min(lane, nkt - 1)] != 0);
    fb1 = __builtin_amdgcn_ballot_w64(has_b && 64 + lane < nkt && flb[min(64 + lane, nkt - 1)] != 0);
  }
  const unsigned long long fm0 = fa0 | fb0, fm1 = fa1 | fb1;
  auto next_tile = [&](int t) {
    if constexpr (GENERAL) {
      if (t < 64) {
        const unsigned long long m = fm0 >> t;
        if (m) return t + (int)__builtin_ctzll(m);
        t = 64;
      }
      if (t < 128) {
        const unsigned long long m = fm1 >> (t - 64);
        if (m) return t + (int)__builtin_ctzll(m);
      }
      return kt_end;
    }
    return t;
  };
  // which staged tiles this wave consumes.  GENERAL: the tiles its own 128-row block has flagged (the dK/dV kernel wrote exactly those).
  // Causal arithmetic: the dK/dV kernel works on 64-row query tiles and touches (query tile qt, key tile t) for t <= qt only.
  const unsigned long long my0 = (wave >> 2) ? fb0 : fa0, my1 = (wave >> 2) ? fb1 : fa1;
  const int my_last = 4 * qb2 + (wave >> 1);
  auto mine = [&](int t) -> bool {
    if constexpr (GENERAL) return t < 64 ? ((my0 >> t) & 1) != 0 : ((my1 >> (t - 64)) & 1) != 0;
    return t <= my_last;
  };

  // ---- staging: per tile step 6 LDS-DMA loads per thread: K tile = 16 pieces of 1 KiB (4 rows), wave w takes pieces w and w + 8;
  // the two dS^T tiles = 32 contiguous pieces, wave w takes pieces w, w + 8, w + 16, w + 24 (copied as they lie)
  const int srow_in = lane >> 4, sslot = lane & 15;
  const bf16_t* kbase = a.k + (int64_t)b * a.k_sb + kvh * HD;
  const bf16_t* dbase = a.ds + ((int64_t)b * a.H + h) * a.Sp * a.Sp + (int64_t)(2 * qb2) * (64 * 128);  // tiles (t, 2qb2), (t, 2qb2+1): 32 KiB
  const int64_t dtile = (int64_t)(a.Sp >> 7) * (64 * 128);
  uint32_t koff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = (i * 8 + wave) * 4 + srow_in;
    koff[i] = (uint32_t)(((int64_t)row * a.k_ss + (sslot ^ dual_swz(row)) * 8) * 2);
  }
  auto stage = [&](int buf, int t) {
    char* sK = smem + buf * DQ2_STAGE_BYTES;
    char* sD = sK + TILE_BYTES;
    const char* dt = (const char*)(dbase + t * dtile) + lane * 16;
#pragma unroll
    for (int i = 0; i < 4; ++i) __builtin_amdgcn_global_load_lds((gbl_void*)(dt + (i * 8 + wave) * 1024), (lds_void*)(sD + (i * 8 + wave) * 1024), 16, 0, 0);
    if (t * BKV + BKV <= a.S) {
      const char* kt = (const char*)(kbase + (int64_t)t * BKV * a.k_ss);
#pragma unroll
      for (int i = 0; i < 2; ++i) __builtin_amdgcn_global_load_lds((gbl_void*)(kt + koff[i]), (lds_void*)(sK + (i * 8 + wave) * 1024), 16, 0, 0);
    } else {  // ragged last key tile: clamp the row (its dS^T rows are zero)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = (i * 8 + wave) * 4 + srow_in;
        const int key = min(t * BKV + row, a.S - 1);
        __builtin_amdgcn_global_load_lds((gbl_void*)(kbase + (int64_t)key * a.k_ss + (sslot ^ dual_swz(row)) * 8), (lds_void*)(sK + (i * 8 + wave) * 1024), 16, 0, 0);
      }
    }
  };

  // lane constants of the transposed reads (element map of tr_frag).  K: byte offset inside the [64][128] dual-use image of the lo / hi
  // 4-row blocks of k-step 0; k-step ks adds 4096 bytes (the swizzle of a row does not depend on ks).  dS^T: the stage holds the two
  // tiles in the PRODUCER's chunk order (AttnBwdArgs::ds): chunk(key, q) at 16 x (kh*512 + qh*256 + qb32*128 + s*64 + hh'*32 + r); this
  // wave's query block fixes the tile (wave>>2) and (qh, qb32) = ((wave>>1)&1, wave&1), the lane's 16-column half s = tsub and 8-column
  // quarter hh' = tp>>1, its key row r = 16 (ks&1) + 4 hh + tq (+8 for the hi block), kh = ks>>1: k-step offsets {0, 256, 8192, 8448},
  // hi block +128 (immediates).  (a 32-lane half reads four 64-byte runs 512 bytes apart: 4-way on the banks - 8 of 40 reads per tile)
  const uint32_t sbase = (uint32_t)(uintptr_t)(lds_char*)smem;
  uint32_t aK_lo[4], aK_hi[4], aD;
  {
    const int tq2 = (lane & 15) >> 2, tp = lane & 3, tsub = (lane >> 4) & 1;
    const int rlo = 4 * hh + tq2, rhi = rlo + 8;
    auto off = [&](int row, int blk) { return (uint32_t)(row * 256 + (((4 * blk + 2 * tsub + (tp >> 1)) ^ dual_swz(row)) << 4) + ((tp & 1) << 3)); };
#pragma unroll
    for (int db = 0; db < 4; ++db) { aK_lo[db] = off(rlo, db); aK_hi[db] = off(rhi, db); }
    aD = (uint32_t)(TILE_BYTES + (wave >> 2) * TILE_BYTES + ((wave >> 1) & 1) * 4096 + (wave & 1) * 2048 + tsub * 1024 + (tp >> 1) * 512 + rlo * 16 + (tp & 1) * 8);
  }
  f32x16_t dq[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) dq[i][e] = 0.f;

  // three LDS stages: tiles t+1 and t+2 are in flight while tile t is consumed (6 LDS-DMA loads per thread and tile; a counted vmcnt
  // retires exactly the oldest tile)
  int tq[DQ2_STAGES];  // tile numbers of the stages in consumption order: tq[0] is consumed next
  tq[0] = next_tile(0);
#pragma unroll
  for (int i = 1; i < DQ2_STAGES; ++i) tq[i] = tq[i - 1] < kt_end ? next_tile(tq[i - 1] + 1) : kt_end;
#pragma unroll
  for (int i = 0; i < DQ2_STAGES - 1; ++i)
    if (tq[i] < kt_end) stage(i, tq[i]);
  int cur = 0;
  while (tq[0] < kt_end) {
    if (tq[2] < kt_end) {
      stage((cur + 2) % DQ2_STAGES, tq[2]);
      asm volatile("s_waitcnt vmcnt(12)" ::: "memory");  // all but the two youngest tiles have landed
    } else if (tq[1] < kt_end) {
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (mine(tq[0])) {
      // 4 k-steps of 16 keys: the 10 transposed reads (dS^T fragment + 4 K^T fragments, lo / hi halves) of step ks+1 are in flight
      // while the 4 MFMAs of step ks run
      const uint32_t so = sbase + cur * DQ2_STAGE_BYTES;
      s16x4_t Dl[2], Dh[2], Kl[2][4], Kh[2][4];
      auto reads = [&](auto set_tag, auto ks_tag) {
        constexpr int st = decltype(set_tag)::value, ks = decltype(ks_tag)::value, off = ks * 4096, doff_ = (ks >> 1) * 8192 + (ks & 1) * 256;
        lds_tr_read<doff_>(Dl[st], so + aD);
        lds_tr_read<doff_ + 128>(Dh[st], so + aD);
#pragma unroll
        for (int db = 0; db < 4; ++db) {
          lds_tr_read<off>(Kl[st][db], so + aK_lo[db]);
          lds_tr_read<off>(Kh[st][db], so + aK_hi[db]);
        }
      };
      auto mfmas = [&](auto set_tag) {
        constexpr int st = decltype(set_tag)::value;
        const bf16x8_t dsf = frag_of(Dl[st], Dh[st]);
#pragma unroll
        for (int db = 0; db < 4; ++db) dq[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_of(Kl[st][db], Kh[st][db]), dsf, dq[db], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      };
      using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
      using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
      reads(I0{}, I0{});
      reads(I1{}, I1{});
      lds_tr_wait<10>(Dl[0], Dh[0], Kl[0], Kh[0]);
      mfmas(I0{});
      reads(I0{}, I2{});
      lds_tr_wait<10>(Dl[1], Dh[1], Kl[1], Kh[1]);
      mfmas(I1{});
      reads(I1{}, I3{});
      lds_tr_wait<10>(Dl[0], Dh[0], Kl[0], Kh[0]);
      mfmas(I0{});
      lds_tr_wait<0>(Dl[1], Dh[1], Kl[1], Kh[1]);
      mfmas(I1{});
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // every wave is done reading stage `cur`: the next iteration's DMA may overwrite it
    asm volatile("" ::: "memory");
    cur = (cur + 1) % DQ2_STAGES;
    tq[0] = tq[1];
    tq[1] = tq[2];
    tq[2] = tq[1] < kt_end ? next_tile(tq[1] + 1) : kt_end;
  }

  if (qi < a.S) {
    bf16_t* op = a.dq + (int64_t)b * a.dq_sb + (int64_t)qi * a.dq_ss + h * HD;
    // (16-byte stores through a half-wave exchange per pair of column groups, as in the forward kernel's epilogue)
    const bool wide = ((((uintptr_t)a.dq) | (uintptr_t)(a.dq_ss * 2) | (uintptr_t)(a.dq_sb * 2)) & 15) == 0;  // uniform
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int j2 = 0; j2 < 2; ++j2) {
        u32x2_t pp[2];
#pragma unroll
        for (int gg = 0; gg < 2; ++gg) {
          const int g4 = 2 * j2 + gg;
          u32x2_t pk;
          pk[0] = pack_bf2(dq[db][4 * g4 + 0] * a.scale, dq[db][4 * g4 + 1] * a.scale);
          pk[1] = pack_bf2(dq[db][4 * g4 + 2] * a.scale, dq[db][4 * g4 + 3] * a.scale);
          if (a.rope) {  // gradient of apply_rope on the bf16-rounded dq, exactly as the stand-alone rope kernel computes it
            const f32x4_t t = *reinterpret_cast<const f32x4_t*>(a.rope + ((int64_t)qi * 64 + ((32 * db + 8 * g4 + 4 * hh) >> 1)) * 2);
            const float c0 = t[0], s0 = t[1] * -1.f, c1 = t[2], s1 = t[3] * -1.f;
            const float x0 = bflo(pk[0]), x1 = bfhi(pk[0]), y0 = bflo(pk[1]), y1 = bfhi(pk[1]);
            pk[0] = pack_bf2(x0 * c0 - x1 * s0, x1 * c0 + x0 * s0);
            pk[1] = pack_bf2(y0 * c1 - y1 * s1, y1 * c1 + y0 * s1);
          }
          pp[gg] = pk;
        }
        if (wide) {
          const auto r0 = __builtin_amdgcn_permlane32_swap(pp[0][0], pp[1][0], false, false);
          const auto r1 = __builtin_amdgcn_permlane32_swap(pp[0][1], pp[1][1], false, false);
          *reinterpret_cast<u32x4_t*>(op + 32 * db + 16 * j2 + 8 * hh) = u32x4_t{r0[0], r1[0], r0[1], r1[1]};
        } else {
          *reinterpret_cast<u32x2_t*>(op + 32 * db + 16 * j2 + 4 * hh) = pp[0];
          *reinterpret_cast<u32x2_t*>(op + 32 * db + 16 * j2 + 8 + 4 * hh) = pp[1];
        }
      }
  }
}

// dk = bf16(scale * sum_g partK[g]) ; dv = bf16(sum_g partV[g]);  8 elements per thread.
__global__ void attn_dkv_reduce_kernel(const AttnBwdArgs a, const float* __restrict__ part) {
  const int G = a.H / a.KVH;
  const int64_t plane = (int64_t)a.B * a.S * a.KVH * HD;
  const int64_t i8 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 8;
  if (i8 >= plane) return;
  const int d = (int)(i8 % HD);
  const int64_t row = i8 / HD;  // (b*S + key)*KVH + kvh
  const int kvh = (int)(row % a.KVH);
  const int64_t bs = row / a.KVH;
  const int key = (int)(bs % a.S);
  const int64_t b = bs / a.S;
  float sk[8] = {0, 0, 0, 0, 0, 0, 0, 0}, sv[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int g = 0; g < G; ++g) {
    const float* pk = part + (int64_t)(g * 2 + 0) * plane + i8;
    const float* pv = part + (int64_t)(g * 2 + 1) * plane + i8;
    const f32x4_t k0 = *reinterpret_cast<const f32x4_t*>(pk), k1 = *reinterpret_cast<const f32x4_t*>(pk + 4);
    const f32x4_t v0 = *reinterpret_cast<const f32x4_t*>(pv), v1 = *reinterpret_cast<const f32x4_t*>(pv + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { sk[e] += k0[e]; sk[4 + e] += k1[e]; sv[e] += v0[e]; sv[4 + e] += v1[e]; }
  }
  u32x4_t ok, ov;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    ok[e] = pack_bf2(sk[2 * e] * a.scale, sk[2 * e + 1] * a.scale);
    ov[e] = pack_bf2(sv[2 * e], sv[2 * e + 1]);
  }
  if (a.rope) ok = rope8(ok, a.rope + ((int64_t)key * 64 + (d >> 1)) * 2, -1.f);
  *reinterpret_cast<u32x4_t*>(a.dk + b * a.dk_sb + (int64_t)key * a.dk_ss + kvh * HD + d) = ok;
  *reinterpret_cast<u32x4_t*>(a.dv + b * a.dv_sb + (int64_t)key * a.dv_ss + kvh * HD + d) = ov;
}

static unsigned long long* g_bwd_stamps = nullptr;

// Diagnostic: the next llx_attn_bwd calls run the dK/dV kernel build that writes s_memtime stamps (10 per query tile of
// workgroup 0, wave 0; needs cls != 0 on every tile, i.e. the plain causal mask) into `stamps` (>= 1024 entries); null turns it off.
extern "C" int llx_debug_attn_bwd_set_stamps(unsigned long long* stamps) { g_bwd_stamps = stamps; return LLX_OK; }

// fp32 workspace of llx_attn_bwd: delta [B,H,S], sanitised -lse [B,H,S], then the dK/dV partials [G][2][B,S,KVH,128].
extern "C" int64_t llx_attn_bwd_workspace_bytes(int64_t B, int64_t S, int64_t H, int64_t KVH) {
  return (2 * B * H * S + (H / KVH) * 2 * B * S * KVH * HD) * 4;
}

// bf16 dS^T buffer of route (a): [B][H][Sp][Sp] with Sp = S rounded up to 256 (only the tiles the mask allows are written and read).
extern "C" int64_t llx_attn_bwd_ds_bytes(int64_t B, int64_t S, int64_t H) {
  const int64_t Sp = cdiv64(S, 256) * 256;
  return B * H * Sp * Sp * 2;
}

static int attn_bwd_set_attrs() {
  static std::once_flag once;
  static bool ok = false;
  std::call_once(once, [] {
    hipError_t e = hipSuccess;
    auto set = [&](const void* f, int bytes) { if (e == hipSuccess) e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, bytes); };
    set((const void*)attn_bwd_dq_kernel<false>, DQ_LDS_BYTES);
    set((const void*)attn_bwd_dq_kernel<true>, DQ_LDS_BYTES);
    set((const void*)attn_bwd_dq2_kernel<false>, DQ2_LDS_BYTES);
    set((const void*)attn_bwd_dq2_kernel<true>, DQ2_LDS_BYTES);
    set((const void*)attn_bwd_dkv3_kernel<true, false, false>, DKV3_LDS_BYTES);
    set((const void*)attn_bwd_dkv3_kernel<false, false, false>, DKV3_LDS_BYTES);
    set((const void*)attn_bwd_dkv3_kernel<true, false, true>, DKV3_LDS_BYTES);
    set((const void*)attn_bwd_dkv3_kernel<false, false, true>, DKV3_LDS_BYTES);
    set((const void*)attn_bwd_dkv3_kernel<false, true, false>, DKV3_LDS_BYTES);
    ok = e == hipSuccess;
  });
  return ok ? LLX_OK : LLX_ERR_LAUNCH;
}

// delta: fp32 workspace of llx_attn_bwd_workspace_bytes() bytes.  All strides in elements.  flags as in llx_attn_fwd.
// rope (nullable): fp32 table [>= S, 64, 2]; when given, q and k are the ROTATED projections and dq, dk come out as the
// gradients of the un-rotated ones (apply_rope's transpose fused into the dQ epilogue and the dK/dV reduce).
// ds (nullable): llx_attn_bwd_ds_bytes() bytes of scratch; when given, dS^T takes one round trip through it and every product of the
// backward is computed once (route (a) at the top of this file); without it the dQ kernel recomputes S and dP.
extern "C" int llx_attn_bwd(const void* q, int64_t q_sb, int64_t q_ss, const void* k, int64_t k_sb, int64_t k_ss, const void* v,
                            int64_t v_sb, int64_t v_ss, const void* o, int64_t o_sb, int64_t o_ss, const void* d_o, int64_t do_sb,
                            int64_t do_ss, const float* lse, float* delta, void* dq, int64_t dq_sb, int64_t dq_ss, void* dk,
                            int64_t dk_sb, int64_t dk_ss, void* dv, int64_t dv_sb, int64_t dv_ss, const int* doc_ids,
                            const int* prefix_len, const void* flags, const float* rope, void* ds, int64_t B, int64_t S, int64_t H,
                            int64_t KVH, int64_t head_dim, float scale, hipStream_t stream) {
  LLX_REQUIRE(q && k && v && o && d_o && lse && delta && dq && dk && dv, "llx_attn_bwd: null pointer");
  LLX_REQUIRE(head_dim == HD, "llx_attn_bwd: head_dim=%lld unsupported (only 128)", (long long)head_dim);
  LLX_REQUIRE(B > 0 && S > 0 && H > 0 && KVH > 0 && H % KVH == 0, "llx_attn_bwd: bad B/S/H/KVH");
  LLX_REQUIRE(((q_ss | k_ss | v_ss | o_ss | do_ss | q_sb | k_sb | v_sb | o_sb | do_sb) % 8) == 0, "llx_attn_bwd: input strides must keep 16-byte alignment");
  LLX_REQUIRE(((dq_ss | dk_ss | dv_ss | dq_sb | dk_sb | dv_sb) % 4) == 0, "llx_attn_bwd: output strides must keep 8-byte alignment");
  LLX_REQUIRE(((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o | (uintptr_t)d_o) % 16 == 0, "llx_attn_bwd: unaligned input");
  LLX_REQUIRE(((uintptr_t)dq) % 8 == 0 && ((uintptr_t)dk | (uintptr_t)dv) % 16 == 0 && ((dk_ss | dv_ss | dk_sb | dv_sb) % 8) == 0,
              "llx_attn_bwd: unaligned output");
  LLX_REQUIRE(!(doc_ids || prefix_len) || flags, "llx_attn_bwd: tile flags required with doc_ids/prefix_len");
  LLX_REQUIRE(!rope || (uintptr_t)rope % 16 == 0, "llx_attn_bwd: unaligned rope table");
  LLX_REQUIRE(!ds || (uintptr_t)ds % 256 == 0, "llx_attn_bwd: the dS buffer must be 256-byte aligned");
  LLX_REQUIRE(S < (1 << 24) && B * H < (1 << 16), "llx_attn_bwd: S or B*H too large");
  if (attn_bwd_set_attrs() != LLX_OK) { llx_set_error("llx_attn_bwd: cannot raise LDS limit"); return LLX_ERR_LAUNCH; }
  AttnBwdArgs a;
  a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.o = (const bf16_t*)o; a.d_o = (const bf16_t*)d_o;
  a.lse = lse; a.delta = delta; a.nlse = delta + B * H * S; a.stamps = nullptr; a.rope = rope; a.dq = (bf16_t*)dq; a.dk = (bf16_t*)dk; a.dv = (bf16_t*)dv;
  a.ds = (bf16_t*)ds; a.Sp = (int)(cdiv64(S, 256) * 256);
  a.q_sb = q_sb; a.q_ss = q_ss; a.k_sb = k_sb; a.k_ss = k_ss; a.v_sb = v_sb; a.v_ss = v_ss; a.o_sb = o_sb; a.o_ss = o_ss;
  a.do_sb = do_sb; a.do_ss = do_ss; a.dq_sb = dq_sb; a.dq_ss = dq_ss; a.dk_sb = dk_sb; a.dk_ss = dk_ss; a.dv_sb = dv_sb; a.dv_ss = dv_ss;
  a.doc_ids = doc_ids; a.prefix_len = prefix_len; a.flags = (doc_ids || prefix_len) ? (const uint8_t*)flags : nullptr;
  a.B = (int)B; a.S = (int)S; a.H = (int)H; a.KVH = (int)KVH;
  a.scale = scale; a.scale_log2 = scale * 1.4426950408889634f;
  const dim3 qgrid((unsigned)H, (unsigned)cdiv64(S, BQ), (unsigned)B);
  // (with tile flags the dQ-from-dS kernel keeps its schedule in two 64-bit masks: up to 128 key tiles = 8192 positions)
  const bool use_ds = ds != nullptr && !g_bwd_stamps && !((doc_ids || prefix_len) && cdiv64(S, BKV) > 128);
  if (use_ds) {
    hipLaunchKernelGGL(attn_bwd_delta_kernel, dim3((unsigned)cdiv64(B * S * H, 16)), dim3(256), 0, stream, a);
    LLX_LAUNCH_CHECK("llx_attn_bwd(delta)");
  } else {
    // dQ first: it also publishes delta = rowsum(dO . O) and the sanitised -lse that the dK/dV kernel stages from global memory
    if (a.flags) hipLaunchKernelGGL(attn_bwd_dq_kernel<true>, qgrid, dim3(256), DQ_LDS_BYTES, stream, a);
    else hipLaunchKernelGGL(attn_bwd_dq_kernel<false>, qgrid, dim3(256), DQ_LDS_BYTES, stream, a);
    LLX_LAUNCH_CHECK("llx_attn_bwd(dq)");
  }
  float* part = delta + 2 * B * H * S;
  const int64_t nkb = cdiv64(S, DKV2_KEYS);
  const dim3 kgrid((unsigned)(nkb * B * H));
  if (use_ds) {
    if (a.flags) hipLaunchKernelGGL((attn_bwd_dkv3_kernel<true, false, true>), kgrid, dim3(256), DKV3_LDS_BYTES, stream, a, part);
    else hipLaunchKernelGGL((attn_bwd_dkv3_kernel<false, false, true>), kgrid, dim3(256), DKV3_LDS_BYTES, stream, a, part);
  } else if (g_bwd_stamps && !a.flags) {
    a.stamps = g_bwd_stamps;
    hipLaunchKernelGGL((attn_bwd_dkv3_kernel<false, true, false>), kgrid, dim3(256), DKV3_LDS_BYTES, stream, a, part);
  } else if (a.flags) hipLaunchKernelGGL((attn_bwd_dkv3_kernel<true, false, false>), kgrid, dim3(256), DKV3_LDS_BYTES, stream, a, part);
  else hipLaunchKernelGGL((attn_bwd_dkv3_kernel<false, false, false>), kgrid, dim3(256), DKV3_LDS_BYTES, stream, a, part);
  LLX_LAUNCH_CHECK("llx_attn_bwd(dkv)");
  const int64_t plane = B * S * KVH * HD;
  hipLaunchKernelGGL(attn_dkv_reduce_kernel, dim3((unsigned)cdiv64(plane / 8, 256)), dim3(256), 0, stream, a, (const float*)part);
  LLX_LAUNCH_CHECK("llx_attn_bwd(dkv reduce)");
  if (use_ds) {
    const dim3 q2grid((unsigned)H, (unsigned)cdiv64(S, DQ2_BQ), (unsigned)B);
    if (a.flags) hipLaunchKernelGGL(attn_bwd_dq2_kernel<true>, q2grid, dim3(512), DQ2_LDS_BYTES, stream, a);
    else hipLaunchKernelGGL(attn_bwd_dq2_kernel<false>, q2grid, dim3(512), DQ2_LDS_BYTES, stream, a);
    LLX_LAUNCH_CHECK("llx_attn_bwd(dq from dS)");
  }
  return LLX_OK;
}
