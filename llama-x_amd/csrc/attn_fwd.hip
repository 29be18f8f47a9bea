// Flash-style attention forward for head_dim 128, GQA by head indexing, masks from per-token metadata.
//
// Replaces F.scaled_dot_product_attention(..., enable_gqa=True) and flex_attention(block_mask=...) of the
// reference (modelling/llama.py:129-137).  Mask rule (bit-exact with the reference's mask functions):
//     allow(q, k) = (k <= q  ||  k < prefix_len[b])  &&  (doc_ids == null || doc_ids[b,q] == doc_ids[b,k])
//   * causal:       prefix_len = null, doc_ids = null        (is_causal=True, modelling/llama.py:135)
//   * document:     doc_ids given                            (mask_mod, train_metamathqa.py:67-68)
//   * prefix-LM:    prefix_len given                         (README.md:16 plan)
// Layout: q [B,S,H,128], k/v [B,S,KVH,128] with arbitrary batch/sequence strides (so views of a fused QKV
// projection work), o [B,S,H,128] contiguous per row, lse [B,H,S] fp32 in log2 units (for the backward).
//
// One workgroup = 4 waves = 128 query rows of one (batch, head); each wave owns 32 rows.  Per 64-key tile:
//   S^T = K.Q^T   (keys on accumulator rows, the query row on the lane -> softmax statistics are lane-local)
//   O^T += V^T.P^T (P^T taken straight from the S^T accumulator registers as the MFMA B operand; V^T fragments
//                   come from the row-major LDS tile through ds_read_b64_tr_b16)
// K/V tiles are double-buffered in LDS by global_load_lds with the bank swizzle applied on the source address.
#include "common.h"
#include <stdlib.h>
#include <mutex>
#include <type_traits>

#define HD 128
#define BQ 128
#define BKV 64
#define KV_TILE_BYTES (BKV * HD * 2)        // 16 KiB
#define ATT_STAGE_BYTES (2 * KV_TILE_BYTES)  // K + V
#define ATT_LDS_BYTES (2 * ATT_STAGE_BYTES)  // 64 KiB

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;
typedef __attribute__((address_space(3))) s16x4_t lds_s16x4;

struct AttnFwdArgs {
  const bf16_t* q; const bf16_t* k; const bf16_t* v; bf16_t* o; float* lse;
  int64_t q_sb, q_ss, k_sb, k_ss, v_sb, v_ss, o_sb, o_ss;  // element strides (batch, sequence)
  const int* doc_ids;     // [B,S] or null
  const int* prefix_len;  // [B] or null
  const uint8_t* flags;   // [B, nqb, nkt] tile classes (0 skip, 1 partial, 2 full) or null => causal arithmetic
  int B, S, H, KVH;
  float scale_log2;       // softmax scale * log2(e)
  unsigned long long* stamps;  // diagnostic (normally null): s_memtime stamps of block (x=0,h=0,b=0), wave 0
};

// GENERAL = false: pure causal (no doc_ids / prefix_len / tile flags) - the mask is index arithmetic only.
// NW = waves per workgroup (32 query rows each).  A K/V tile pair is 32 KiB of LDS-DMA per workgroup and key tile, and a CU takes
// LDS-DMA fills at ~25-40 GB/s whatever their source (measured on the dQ-from-dS kernel, attn_bwd.hip): two 4-wave workgroups per CU
// ask for ~60 GB/s at this kernel's MFMA rate, ONE 8-wave workgroup (256 query rows sharing every tile) for half of that.
template <bool GENERAL, bool STAMP = false, int NW = 8>
__global__ __launch_bounds__(64 * NW, 2) void attn_fwd_kernel(const AttnFwdArgs a) {
  constexpr int WQ = 32 * NW;   // query rows per workgroup
  constexpr int NP = 16 / NW;   // 1-KiB staging pieces per wave, tile and operand
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nqb = (a.S + BQ - 1) / BQ, nkt = (a.S + BKV - 1) / BKV;  // 128-row blocks (granularity of the tile flags), key tiles
  const int nwb = (a.S + WQ - 1) / WQ;
  // grid = (heads, q-blocks, batch): the q-block index is the SLOW dispatch dimension, so that under a causal mask the
  // heaviest blocks of EVERY head are handed out first (longest-processing-time order: no heavy straggler at the end)
  const int qb = nwb - 1 - blockIdx.y;
  const int h = blockIdx.x, b = blockIdx.z;
  const int kvh = h / (a.H / a.KVH);
  const int r = lane & 31, hh = lane >> 5;
  const int qi = qb * WQ + wave * 32 + r;  // this lane's query row
  const int q_lo = qb * WQ + wave * 32;    // first query row of this wave
  const int qrow = min(qi, a.S - 1);

  // ---- Q fragments (B operand of S^T = K.Q^T): Q[q=r][d = 16ks + 8hh + j]
  bf16x8_t qf[8];
  {
    const bf16_t* qp = a.q + (int64_t)b * a.q_sb + (int64_t)qrow * a.q_ss + h * HD + 8 * hh;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) qf[ks] = *reinterpret_cast<const bf16x8_t*>(qp + 16 * ks);
  }

  // ---- tile schedule.  The workgroup stages every key tile some wave needs; a wave computes the tiles ITS 32 rows need and classes
  // them itself: 0 nothing to attend to (skipped), 1 partly masked, 2 no masking.  GENERAL: from the tile flags of the wave's own
  // 128-row block (the workgroup's schedule = tiles either of its 128-row blocks needs); causal: index arithmetic on the wave's rows.
  const int my_qb = min((qb * WQ + wave * 32) / BQ, nqb - 1);
  const uint8_t* fl = GENERAL ? a.flags + ((int64_t)b * nqb + my_qb) * nkt : nullptr;
  const uint8_t* fl0 = GENERAL ? a.flags + ((int64_t)b * nqb + min(qb * WQ / BQ, nqb - 1)) * nkt : nullptr;
  const uint8_t* fl1 = GENERAL ? a.flags + ((int64_t)b * nqb + min(qb * WQ / BQ + (NW > 4 ? 1 : 0), nqb - 1)) * nkt : nullptr;
  const int kt_end = GENERAL ? nkt : min(nkt, (qb * WQ + WQ + BKV - 1) / BKV);
  // GENERAL: the three flag bytes of a tile (this wave's block, the workgroup's two blocks) are fetched 64 tiles at a time into ONE
  // register - lane i holds tile 64*chunk + i - and read with v_readlane: as a byte load per tile and block they put two or three
  // dependent memory round trips in front of every tile (the forward ran 1.7x the causal time on a prefix-LM mask with 1.25x its work).
  uint32_t fpack = 0;
  int fchunk = -1;
  auto flags_of = [&](int t) -> uint32_t {  // wave-uniform t < nkt
    if ((t >> 6) != fchunk) {
      fchunk = t >> 6;
      const int idx = min(fchunk * 64 + lane, nkt - 1);
      fpack = (uint32_t)fl[idx] | ((uint32_t)fl0[idx] << 8) | ((uint32_t)fl1[idx] << 16);
    }
    return (uint32_t)__builtin_amdgcn_readlane((int)fpack, t & 63);
  };
  auto tile_class = [&](int t) -> int {  // of this wave
    if constexpr (GENERAL) return (int)(flags_of(t) & 0xff);
    else return (t * BKV > q_lo + 31) ? 0 : ((t * BKV + BKV - 1 <= q_lo) ? 2 : 1);
  };
  auto next_tile = [&](int t) {  // of the workgroup
    if constexpr (GENERAL) while (t < kt_end && (flags_of(t) >> 8) == 0) ++t;
    return t;
  };

  // ---- staging: 16 KiB tile = 16 wave-instructions of 1 KiB (4 rows x 256 B); lane -> row l>>4, slot l&15
  const int srow_in = lane >> 4, sslot = lane & 15;
  const bf16_t* kbase = a.k + (int64_t)b * a.k_sb + kvh * HD;
  const bf16_t* vbase = a.v + (int64_t)b * a.v_sb + kvh * HD;
  // loop-invariant per-lane byte offsets inside a tile; the wave-uniform tile base advances by 64 rows per tile
  uint32_t koff[NP], voff[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int row = (i * NW + wave) * 4 + srow_in;
    koff[i] = (uint32_t)(((int64_t)row * a.k_ss + (sslot ^ (row & 15)) * 8) * 2);          // K image: slot = chunk ^ (row & 15)
    voff[i] = (uint32_t)(((int64_t)row * a.v_ss + (sslot ^ ((row & 3) << 2)) * 8) * 2);    // V image: slot = chunk ^ ((row & 3) << 2)
  }
  auto stage = [&](int buf, int t) {
    char* sK = smem + buf * ATT_STAGE_BYTES;
    char* sV = sK + KV_TILE_BYTES;
    if (t * BKV + BKV <= a.S) {  // full tile: uniform base + invariant lane offset, no vector address arithmetic
      const char* kt = (const char*)(kbase + (int64_t)t * BKV * a.k_ss);
      const char* vt = (const char*)(vbase + (int64_t)t * BKV * a.v_ss);
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        __builtin_amdgcn_global_load_lds((gbl_void*)(kt + koff[i]), (lds_void*)(sK + (i * NW + wave) * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gbl_void*)(vt + voff[i]), (lds_void*)(sV + (i * NW + wave) * 1024), 16, 0, 0);
      }
    } else {  // ragged last tile: clamp rows past the end (they are masked out)
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const int row = (i * NW + wave) * 4 + srow_in;
        const int key = min(t * BKV + row, a.S - 1);
        const int kc = sslot ^ (row & 15);
        const int vc = sslot ^ ((row & 3) << 2);
        __builtin_amdgcn_global_load_lds((gbl_void*)(kbase + (int64_t)key * a.k_ss + kc * 8), (lds_void*)(sK + (i * NW + wave) * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gbl_void*)(vbase + (int64_t)key * a.v_ss + vc * 8), (lds_void*)(sV + (i * NW + wave) * 1024), 16, 0, 0);
      }
    }
  };

  f32x16_t o[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[i][e] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  const int* docrow = (GENERAL && a.doc_ids) ? a.doc_ids + (int64_t)b * a.S : nullptr;
  const int my_doc = docrow ? docrow[qrow] : 0;
  const int my_prefix = (GENERAL && a.prefix_len) ? a.prefix_len[b] : 0;

  // tr-read lane constants: group-local i = lane&15 -> q4 = i>>2 (row in block), p = i&3.  aV[db] = byte offset inside the V image of
  // this lane's lo 4-row block of k-step 0 for d-block db: row 4hh + tq, chunk (4db + 2tsub + (tp>>1)) ^ (tq << 2) (the V image's
  // swizzle; (row & 3) == tq for every block), 8-byte half tp & 1.  Step (kb, s) adds 4096 bytes, the hi block 2048.
  const int tq = (lane & 15) >> 2, tp = lane & 3;
  const int tsub = (lane >> 4) & 1;
  const uint32_t sbase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  uint32_t aV[4];
#pragma unroll
  for (int db = 0; db < 4; ++db) aV[db] = (uint32_t)((4 * hh + tq) * 256 + (((4 * db + 2 * tsub + (tp >> 1)) ^ (tq << 2)) << 4) + ((tp & 1) << 3));

  int t = next_tile(0);
  // document ids of the tile's 64 keys: lane i holds key 64 t + i.  Requested one tile ahead (a coalesced 256-byte load under the tile's
  // compute) and gathered per element with ds_bpermute - as 32 per-element global loads inside a partly masked tile they put a memory
  // round trip into every such tile (all of them, with packed documents shorter than a query block).
  int docv = 0, docv_next = 0;
  if (GENERAL && docrow && t < kt_end) docv = docrow[min(t * BKV + lane, a.S - 1)];
  if (t < kt_end) stage(0, t);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int cur = 0;
  int nst = 0;
  auto stamp = [&]() {
    if constexpr (STAMP) {
      if (blockIdx.y == 0 && blockIdx.x == 0 && blockIdx.z == 0 && wave == 0 && nst < 512) {
        unsigned long long tt;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tt) :: "memory");
        if (lane == 0) a.stamps[nst] = tt;
        ++nst;
      }
    }
  };
  while (t < kt_end) {
    stamp();  // 0: tile start
    const int tn = next_tile(t + 1);
    if (tn < kt_end) stage(cur ^ 1, tn);
    if (GENERAL && docrow && tn < kt_end) docv_next = docrow[min(tn * BKV + lane, a.S - 1)];
    const char* sK = smem + cur * ATT_STAGE_BYTES;
    const char* sV = sK + KV_TILE_BYTES;
    const int cls = tile_class(t);
    if (cls != 0) {  // wave-uniform: a tile none of this wave's rows attends to is only staged (for the other waves)

    // ---- S^T = K.Q^T : 2 key blocks x 8 k-steps
    f32x16_t st[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      const int row = kb * 32 + r;
      const f32x16_t zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      // all 8 K fragments of the key block are in flight before the first MFMA: one LDS latency per block, not per MFMA
      bf16x8_t kf[8];
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) kf[ks] = *reinterpret_cast<const bf16x8_t*>(sK + row * 256 + (((2 * ks + hh) ^ (row & 15)) << 4));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) st[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf[ks], ks == 0 ? zero : st[kb], 0, 0, 0);
    }

    stamp();  // 1: after QK^T
    // ---- mask, online softmax in log2 units (row statistics are per lane; the partner half-wave holds the other keys)
    float mx = -INFINITY;
    if (cls != 2) {
      const int kk0 = t * BKV + 4 * hh;  // this lane's first key of the tile; element (kb, e) adds 32 kb + (e & 3) + 8 (e >> 2)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int kk = kk0 + kb * 32 + (e & 3) + 8 * (e >> 2);
          bool ok = (kk < a.S) && (kk <= qi || kk < my_prefix);
          if constexpr (GENERAL) {
            const int kd = docrow ? __builtin_amdgcn_ds_bpermute((kk - t * BKV) << 2, docv) : my_doc;
            ok = ok && (kd == my_doc);
          }
          st[kb][e] = ok ? st[kb][e] : -INFINITY;
        }
        if constexpr (GENERAL) __builtin_amdgcn_sched_barrier(0);  // 16 gathered ids at a time, not 32 (registers)
      }
    }
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int e = 0; e < 16; ++e) mx = fmaxf(mx, st[kb][e]);
    {  // combine with the partner half-wave: v_permlane32_swap (VALU) instead of a shuffle through the LDS crossbar
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
      mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1])) * a.scale_log2;  // scale > 0: max commutes with the scaling
    }
    // Deferred running maximum: the base of the exponentials only moves when some row's maximum grew by more than 2^8 (one
    // wave-uniform decision per tile).  Until then p = exp2(s - m_stale) <= 256 - bf16 keeps its relative precision there and
    // O / l are normalised by the same base at the end - and the 64-register rescale of O is skipped on almost every tile.
    const float m_cand = fmaxf(m_run, mx);
    const bool move_base = __builtin_amdgcn_ballot_w64(m_cand > m_run + 8.f) != 0;
    const float m_new = move_base ? m_cand : m_run;
    const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;
    // (the V^T reads of the first two k-steps go out BEFORE the exponentials: their VALU time covers the LDS latency)
    const uint32_t vb = sbase + cur * ATT_STAGE_BYTES + KV_TILE_BYTES;
    s16x4_t Vl[2][4], Vh[2][4];
    auto reads = [&](auto set_tag, auto step_tag) {
      constexpr int st_ = decltype(set_tag)::value, off = decltype(step_tag)::value * 4096;  // step = (kb, s): 16 keys = 4096 bytes
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        lds_tr_read<off>(Vl[st_][db], vb + aV[db]);
        lds_tr_read<off + 2048>(Vh[st_][db], vb + aV[db]);  // the hi block: 8 rows further
      }
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
    reads(I0{}, I0{});
    reads(I1{}, I1{});
    float rs = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(st[kb][e], a.scale_log2, -m_safe));  // exp2(-inf) = 0 for masked keys
        st[kb][e] = p;
        rs += p;
      }
    {
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(rs), __float_as_uint(rs), false, false);
      rs = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
    }
    if (move_base) {  // rescale every row to its current maximum (rows that did not move get alpha = 1)
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_safe);  // m_run = -inf -> 0
      l_run *= alpha;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[i][e] *= alpha;
    }
    l_run += rs;
    m_run = m_new;

    stamp();  // 2: after softmax
    // ---- O^T += V^T.P^T : P^T k-step (kb, s) = accumulator regs 8s..8s+7; element j <-> key 32kb+16s+8(j>>2)+4hh+(j&3).
    // V^T fragments by transposed reads issued as inline asm (common.h: lds_tr_read - the builtin form would drain the K/V prefetch
    // of the next tile right here): the 8 reads of step i+1 are in flight while the 4 MFMAs of step i run.
    {
      auto pv = [&](auto set_tag, auto step_tag) {
        constexpr int st_ = decltype(set_tag)::value, step = decltype(step_tag)::value;
        bf16x8_t pb;
#pragma unroll
        for (int j = 0; j < 8; ++j) pb[j] = (__bf16)st[step >> 1][8 * (step & 1) + j];
        if constexpr (step < 3) lds_tr_wait8<8>(Vl[st_], Vh[st_]);
        else lds_tr_wait8<0>(Vl[st_], Vh[st_]);
#pragma unroll
        for (int db = 0; db < 4; ++db) o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_of(Vl[st_][db], Vh[st_][db]), pb, o[db], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      };
      pv(I0{}, I0{});
      reads(I0{}, I2{});
      pv(I1{}, I1{});
      reads(I1{}, I3{});
      pv(I0{}, I2{});
      pv(I1{}, I3{});
    }
    }  // cls != 0

    stamp();  // 3: after PV
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stamp();  // 4: after the load wait
    __syncthreads();
    cur ^= 1;
    t = tn;
    docv = docv_next;
  }

  // ---- finalize: O = O^T / l ; lse = m + log2(l)
  if (qi < a.S) {
    const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
    bf16_t* op = a.o + (int64_t)b * a.o_sb + (int64_t)qi * a.o_ss + h * HD;
    // A lane holds columns 8k+4hh..+3 of its row for 16 column groups k: stored as they lie that is 16 8-byte stores per lane, and the
    // tail of a block is bound by the number of store instructions.  A half-wave exchange per pair of groups (v_permlane32_swap: the
    // upper half's group-k words against the lower half's group-(k+1) words) leaves 16 contiguous bytes per lane - 8 stores.
    const bool wide = ((((uintptr_t)a.o) | (uintptr_t)(a.o_ss * 2) | (uintptr_t)(a.o_sb * 2)) & 15) == 0;  // uniform
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int j2 = 0; j2 < 2; ++j2) {
        u32x2_t pa, pb2;
        pa[0] = pack_bf2(o[db][8 * j2 + 0] * inv, o[db][8 * j2 + 1] * inv);
        pa[1] = pack_bf2(o[db][8 * j2 + 2] * inv, o[db][8 * j2 + 3] * inv);
        pb2[0] = pack_bf2(o[db][8 * j2 + 4] * inv, o[db][8 * j2 + 5] * inv);
        pb2[1] = pack_bf2(o[db][8 * j2 + 6] * inv, o[db][8 * j2 + 7] * inv);
        if (wide) {
          const auto r0 = __builtin_amdgcn_permlane32_swap(pa[0], pb2[0], false, false);
          const auto r1 = __builtin_amdgcn_permlane32_swap(pa[1], pb2[1], false, false);
          *reinterpret_cast<u32x4_t*>(op + 32 * db + 16 * j2 + 8 * hh) = u32x4_t{r0[0], r1[0], r0[1], r1[1]};
        } else {
          *reinterpret_cast<u32x2_t*>(op + 32 * db + 16 * j2 + 4 * hh) = pa;
          *reinterpret_cast<u32x2_t*>(op + 32 * db + 16 * j2 + 8 + 4 * hh) = pb2;
        }
      }
    if (hh == 0 && a.lse) a.lse[((int64_t)b * a.H + h) * a.S + qi] = (l_run > 0.f) ? m_run + log2f(l_run) : -INFINITY;
  }
}

// Tile classes for non-causal-only masks: flags[b][qb][kt] = 0 (no pair allowed) | 1 (some) | 2 (all, no masking).
__global__ void attn_tile_flags_kernel(const int* __restrict__ doc_ids, const int* __restrict__ prefix_len, uint8_t* flags, int S,
                                       int nqb, int nkt) {
  __shared__ int s_any, s_all;
  const int kt = blockIdx.x, qb = blockIdx.y, b = blockIdx.z;
  if (threadIdx.x == 0) { s_any = 0; s_all = 1; }
  __syncthreads();
  const int P = prefix_len ? prefix_len[b] : 0;
  int any = 0, all = 1;
  for (int idx = threadIdx.x; idx < BQ * BKV; idx += blockDim.x) {
    const int qi = qb * BQ + idx / BKV, kk = kt * BKV + idx % BKV;
    if (qi >= S) continue;  // rows past the end do not constrain the class
    bool ok = (kk < S) && (kk <= qi || kk < P);
    if (ok && doc_ids) ok = doc_ids[(int64_t)b * S + qi] == doc_ids[(int64_t)b * S + kk];
    any |= ok; all &= ok;
  }
  if (any) atomicOr(&s_any, 1);
  if (!all) atomicAnd(&s_all, 0);
  __syncthreads();
  if (threadIdx.x == 0) flags[((int64_t)b * nqb + qb) * nkt + kt] = s_any ? (s_all ? 2 : 1) : 0;
}


extern "C" int64_t llx_attn_flags_bytes(int64_t B, int64_t S) { return B * cdiv64(S, BQ) * cdiv64(S, BKV); }

// flags: llx_attn_flags_bytes(B,S) bytes (device), filled here. Needed only when doc_ids or prefix_len is used.
extern "C" int llx_attn_tile_flags(const int* doc_ids, const int* prefix_len, void* flags, int64_t B, int64_t S, hipStream_t stream) {
  LLX_REQUIRE(flags && B > 0 && S > 0, "llx_attn_tile_flags: bad arguments");
  const int nqb = (int)cdiv64(S, BQ), nkt = (int)cdiv64(S, BKV);
  hipLaunchKernelGGL(attn_tile_flags_kernel, dim3(nkt, nqb, (unsigned)B), dim3(256), 0, stream, doc_ids, prefix_len, (uint8_t*)flags,
                     (int)S, nqb, nkt);
  LLX_LAUNCH_CHECK("llx_attn_tile_flags");
  return LLX_OK;
}

// strides are in elements: *_sb per batch, *_ss per sequence position; head h starts at element h*128 of a row.
extern "C" int llx_attn_fwd(const void* q, int64_t q_sb, int64_t q_ss, const void* k, int64_t k_sb, int64_t k_ss, const void* v,
                            int64_t v_sb, int64_t v_ss, void* o, int64_t o_sb, int64_t o_ss, float* lse, const int* doc_ids,
                            const int* prefix_len, const void* flags, int64_t B, int64_t S, int64_t H, int64_t KVH, int64_t head_dim,
                            float scale, hipStream_t stream) {
  LLX_REQUIRE(q && k && v && o, "llx_attn_fwd: null pointer");
  LLX_REQUIRE(head_dim == HD, "llx_attn_fwd: head_dim=%lld unsupported (only 128)", (long long)head_dim);
  LLX_REQUIRE(B > 0 && S > 0 && H > 0 && KVH > 0 && H % KVH == 0, "llx_attn_fwd: bad B/S/H/KVH");
  LLX_REQUIRE((q_ss % 8 | k_ss % 8 | v_ss % 8 | o_ss % 4 | q_sb % 8 | k_sb % 8 | v_sb % 8 | o_sb % 4) == 0, "llx_attn_fwd: strides must keep 16-byte alignment");
  LLX_REQUIRE(((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) % 16 == 0 && (uintptr_t)o % 8 == 0, "llx_attn_fwd: unaligned pointer");
  LLX_REQUIRE(!(doc_ids || prefix_len) || flags, "llx_attn_fwd: tile flags required with doc_ids/prefix_len");
  LLX_REQUIRE(S < (1 << 24), "llx_attn_fwd: S too large");
  static int nw = 0;  // LLX_ATTN_FWD_NW=4: the 128-row workgroup (two per CU) this kernel had before, kept for A/B measurements
  {
    static std::once_flag once;  // forward may be entered from several host threads (activation checkpointing recomputes it in backward)
    static hipError_t err = hipSuccess;
    std::call_once(once, [] {
      const char* e = getenv("LLX_ATTN_FWD_NW");
      nw = (e && e[0] == '4') ? 4 : 8;
      const void* fns[4] = {(const void*)attn_fwd_kernel<false, false, 4>, (const void*)attn_fwd_kernel<true, false, 4>,
                            (const void*)attn_fwd_kernel<false, false, 8>, (const void*)attn_fwd_kernel<true, false, 8>};
      for (int i = 0; i < 4 && err == hipSuccess; ++i) err = hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, ATT_LDS_BYTES);
    });
    if (err != hipSuccess) { llx_set_error("llx_attn_fwd: %s", hipGetErrorString(err)); return LLX_ERR_LAUNCH; }
  }
  AttnFwdArgs a;
  a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.o = (bf16_t*)o; a.lse = lse;
  a.q_sb = q_sb; a.q_ss = q_ss; a.k_sb = k_sb; a.k_ss = k_ss; a.v_sb = v_sb; a.v_ss = v_ss; a.o_sb = o_sb; a.o_ss = o_ss;
  a.doc_ids = doc_ids; a.prefix_len = prefix_len; a.flags = (doc_ids || prefix_len) ? (const uint8_t*)flags : nullptr;
  a.B = (int)B; a.S = (int)S; a.H = (int)H; a.KVH = (int)KVH;
  a.scale_log2 = scale * 1.4426950408889634f;
  a.stamps = nullptr;
  const dim3 grid((unsigned)H, (unsigned)cdiv64(S, 32 * nw), (unsigned)B), block(64 * nw);
  if (nw == 8) {
    if (a.flags) hipLaunchKernelGGL((attn_fwd_kernel<true, false, 8>), grid, block, ATT_LDS_BYTES, stream, a);
    else hipLaunchKernelGGL((attn_fwd_kernel<false, false, 8>), grid, block, ATT_LDS_BYTES, stream, a);
  } else {
    if (a.flags) hipLaunchKernelGGL((attn_fwd_kernel<true, false, 4>), grid, block, ATT_LDS_BYTES, stream, a);
    else hipLaunchKernelGGL((attn_fwd_kernel<false, false, 4>), grid, block, ATT_LDS_BYTES, stream, a);
  }
  LLX_LAUNCH_CHECK("llx_attn_fwd");
  return LLX_OK;
}

// Diagnostic: resident workgroups per CU the runtime grants the forward kernel (occupancy API; advisory).
extern "C" int llx_debug_attn_fwd_occupancy(void) {
  int n = -1;
  hipFuncSetAttribute((const void*)attn_fwd_kernel<false, false, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, ATT_LDS_BYTES);
  hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)attn_fwd_kernel<false, false, 8>, 512, ATT_LDS_BYTES);
  if (e != hipSuccess) { llx_set_error("occupancy query: %s", hipGetErrorString(e)); return -1; }
  return n;
}

// Diagnostic build of the forward kernel with in-kernel s_memtime stamps (5 per key tile) for one wave; timing only.
extern "C" int llx_debug_attn_fwd_stamps(const void* q, const void* k, const void* v, void* o, int64_t S, int64_t H, int64_t KVH,
                                         unsigned long long* stamps, hipStream_t stream) {
  AttnFwdArgs a;
  a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.o = (bf16_t*)o; a.lse = nullptr;
  a.q_ss = H * HD; a.q_sb = S * a.q_ss; a.k_ss = KVH * HD; a.k_sb = S * a.k_ss; a.v_ss = a.k_ss; a.v_sb = a.k_sb; a.o_ss = a.q_ss; a.o_sb = a.q_sb;
  a.doc_ids = nullptr; a.prefix_len = nullptr; a.flags = nullptr; a.B = 1; a.S = (int)S; a.H = (int)H; a.KVH = (int)KVH;
  a.scale_log2 = 0.08838834764f * 1.4426950408889634f; a.stamps = stamps;
  hipFuncSetAttribute((const void*)attn_fwd_kernel<false, true, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, ATT_LDS_BYTES);
  hipLaunchKernelGGL((attn_fwd_kernel<false, true, 8>), dim3((unsigned)H, (unsigned)cdiv64(S, 256), 1), dim3(512), ATT_LDS_BYTES, stream, a);
  LLX_LAUNCH_CHECK("llx_debug_attn_fwd_stamps");
  return LLX_OK;
}
