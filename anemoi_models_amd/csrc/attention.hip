// Mesh-node multi-head self attention (K7 of SURVEY.md section 2a): softmax(Q K^T / sqrt(D)) V per (batch, head),
// flash style (no S x S matrix in memory), directly on the fused lin_qkv output  qkv [B*S, 3C] = q | k | v.
//
// bf16, D = 64 or 32: matrix cores.  One wave owns 2 x 32 query rows; the scores are produced TRANSPOSED,
//     S^T[key][query] = K Q^T      via v_mfma_f32_32x32x16_bf16 with A = K rows, B = Q^T,
// so that each lane holds 16 scores of ONE query: the online-softmax row max / row sum are in-lane reductions plus a
// single cross-half exchange.  The K rows are assigned to MFMA rows with index bits 2 and 3 swapped; with the
// accumulator map  row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)  this makes registers 0..7 / 8..15 of a lane hold 8
// CONTIGUOUS keys each, i.e. exactly the B-operand fragment of the second product
//     O^T[d][query] += V^T[d][key] P^T[key][query]
// without any cross-lane movement of P.  V is transposed once per call into [B, H, D, S_pad] so that the V^T
// fragments are plain 16-byte LDS reads.  K and V^T tiles (64 keys) are staged with global_load_lds into a
// double-buffered LDS image using the GEMM kernel's 128-byte-row XOR swizzle.
//
// f32 (parity path) and other head sizes: a VALU kernel, one wave per (query, head), keys strided over the lanes.
//
// `window` >= 0 applies flash-attn's sliding window (key j visible from query i iff |i - j| <= window);
// window < 0 = global attention (the reference's SDPA fallback, layers/attention.py:99-105).
#include <type_traits>
#include <utility>

#include "common.hpp"

#ifndef ANEMOI_LAB_MHSA8_PAD
#define ANEMOI_LAB_MHSA8_PAD 0  // lab switch: full wait states behind the eight-wave kernel's S^T products (see there)
#endif

namespace anemoi {

typedef __attribute__((ext_vector_type(8))) __bf16 abf16x8_t;
typedef __attribute__((ext_vector_type(16))) float af32x16_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;

__device__ __forceinline__ void aglds16(const void* gptr, void* lptr) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gptr,
                                   (__attribute__((address_space(3))) void*)lptr, 16, 0, 0);
}
__device__ __forceinline__ void aglds4(const void* gptr, void* lptr) {  // one dword per lane: LDS base + lane * 4
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gptr,
                                   (__attribute__((address_space(3))) void*)lptr, 4, 0, 0);
}
__device__ __forceinline__ int aswz(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }

// ---------------------------------------------------------------------------------------------
// vt[b, h, d, s] = qkv[b*S + s, 2C + h*D + d]   (zero padded to S_pad keys)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void transpose_v_kernel(const bf16_t* __restrict__ qkv, int64_t ld, int S, int S_pad,
                                                          int H, int D, int C, bf16_t* __restrict__ vt,
                                                          int col0 = -1) {  // first source column (default: v = 2 C)
  if (col0 < 0) col0 = 2 * C;
  __shared__ bf16_t tile[64][66];
  const int s0 = blockIdx.x * 64, h = blockIdx.y, b = blockIdx.z;
  for (int d0 = 0; d0 < D; d0 += 64) {
    for (int idx = threadIdx.x; idx < 64 * 64; idx += 256) {
      const int r = idx >> 6, c = idx & 63;
      const int s = s0 + r, d = d0 + c;
      tile[r][c] = (s < S && d < D) ? qkv[((int64_t)b * S + s) * ld + col0 + h * D + d] : (bf16_t)0;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < 64 * 64; idx += 256) {
      const int d = idx >> 6, r = idx & 63;
      if (d0 + d < D && s0 + r < S_pad) vt[(((int64_t)b * H + h) * D + d0 + d) * S_pad + s0 + r] = tile[r][d];
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// bf16 MFMA kernel (D = 64 / 32): 8 waves x 64 queries per workgroup, 64-key tiles
// ---------------------------------------------------------------------------------------------
constexpr int ATT_KV = 64;                                                  // keys per LDS tile
constexpr int ATT_WAVES = 8, ATT_QPW = 64, ATT_QBLK = ATT_WAVES * ATT_QPW;  // 512 queries per workgroup

// 64-byte rows (D = 32) need their own swizzle: a ds_read_b128 is served in groups of 16 lanes, which here read 16
// different keys at the same chunk; chunk ^ ((row >> 2) & 3) spreads rows r, r + 4, r + 8, r + 12 over the four 16-byte
// positions of the row, (row & 3) over the four 64-byte bank quarters: conflict free (128-byte rows: aswz, as the GEMM).
__device__ __forceinline__ int aswz64(int row, int chunk) { return chunk ^ ((row >> 2) & 3); }

// Every workgroup streams ALL keys / values of its (batch, head) through LDS, so the L2 -> LDS traffic of a layer is
// (S / ATT_QBLK) * H * S * 256 bytes: with 128 queries per workgroup that was 54 GB per layer at S = 40 962 (the kernel
// ran at the L2 rate, not the MFMA rate); 512 queries per workgroup and two 32-query blocks per wave (K / V^T
// fragments read from LDS once, used by two MFMAs) cut it 4x and halve the LDS -> register traffic per MFMA.
// ATT_D = 64 (config 3: 1024 channels / 16 heads) or 32 (config 2: 512 / 16): NKS = D / 16 MFMAs per S^T block along the
// head dimension, NDT = D / 32 row blocks of O^T; the K tile has D * 2 bytes per key, the V^T tile D rows of 128 bytes.
// ---------------------------------------------------------------------------------------------
// Attention dropout (reference layers/attention.py:90-105: dropout_p of SDPA / flash-attn in training mode).  The keep
// decision of probability (b, h, i, j) is a counter-based hash of its index and a per-call seed -- nothing of size S x S
// is stored, the backward kernels rebuild exactly the same mask.  One 32-bit hash of the row and the KEY PAIR j >> 1
// decides two neighbouring keys, 15 bits each (the forward and the dQ kernel hold neighbouring keys of one query in one
// packed register pair: one hash per pair; p is resolved to 2^-15).  Kept probabilities are scaled by 1 / (1 - p); the
// softmax normaliser is taken before the dropout, as in the reference.  p >= 1 drops everything (output and gradients 0).
// The row index uses the GLOBAL head (h0 + h of h_total): a head-sharded call (sequence-parallel attention across a model
// group) draws the mask of the unsharded one.
// Round 5: the mask generator is cut to what a keep decision needs -- ONE multiply in the mixer (x ^= x >> 16; x *= M;
// x ^= x >> 15: every decision bit depends on all 32 input bits through the product's upper half and its fold-down) and
// 15-bit thresholds, so that a pair's and-mask is three packed 16-bit operations (mask, subtract, arithmetic shift).  With
// the two-multiply mixer and 16-bit compares the mask cost more VALU cycles than the softmax it masks: the dropout
// variants of the kernels ran 1.7 x (forward) / 1.9 x (backward) the plain ones (profiles/r05_train_step_bench.txt).
// ---------------------------------------------------------------------------------------------
struct AttnDropout {
  uint32_t thr15;      // keep  <=>  15 hash bits >= thr15  (thr15 = p * 2^15; 0: no dropout)
  uint32_t seed;
  float keep_scale;    // 1 / (1 - p), 0 when p >= 1
  int drop_all;
  int h0, h_total;     // global index of this call's head 0, number of heads of the whole attention
  // optional DEVICE part of the seed (the low 32 bits of a word in device memory, added to `seed` by every kernel at its
  // start): a training step captured in a HIP graph replays its kernel arguments, so what must change from step to step
  // -- the dropout mask -- has to come from memory the graph itself advances (runtime.DeviceDropout)
  const uint32_t* seed_dev;
};

__device__ __forceinline__ AttnDropout dropout_resolve(const AttnDropout& in) {
  AttnDropout dr = in;
  if (dr.thr15 != 0 && dr.seed_dev != nullptr) dr.seed += __builtin_nontemporal_load(dr.seed_dev);
  return dr;
}

__device__ __forceinline__ int64_t dropout_row(const AttnDropout& dr, int64_t b, int h, int64_t S, int64_t q) {
  return (b * dr.h_total + dr.h0 + h) * S + q;
}

__device__ __forceinline__ uint32_t dropout_row_part(const AttnDropout& dr, int64_t row) {
  return (uint32_t)row * 0x9E3779B1u ^ (uint32_t)(row >> 32) * 0x85EBCA77u ^ dr.seed;
}

// the 2 x 15 decision bits of key pair `pair` (= key >> 1): x = row part ^ pair * C3 (C3 = 0xC2B2AE3D), mixed once
__device__ __forceinline__ uint32_t dropout_mix(uint32_t x) {
  x ^= x >> 16;
  x *= 0x7feb352du;
  x ^= x >> 15;
  return x;
}

// and-mask of a packed bf16 pair from the pair's hash: 0xffff in the half whose 15 bits are >= thr15.  `cthr2` = thr15 - 1 in
// both halves (dropout_cthr2): (thr15 - 1) - x15 is negative exactly for the kept halves, its sign fills the half.
typedef short att_s16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t dropout_cthr2(const uint32_t thr15) { return ((thr15 - 1u) & 0xffffu) * 0x00010001u; }
__device__ __forceinline__ uint32_t dropout_pair_mask(uint32_t x, uint32_t cthr2) {
  const att_s16x2_t d = __builtin_bit_cast(att_s16x2_t, cthr2) - __builtin_bit_cast(att_s16x2_t, x & 0x7fff7fffu);
  return __builtin_bit_cast(uint32_t, d >> 15);
}

__device__ __forceinline__ float dropout_keep(const AttnDropout& dr, int64_t row, int col) {
  // row = dropout_row(b, h, q), col = key
  if (dr.thr15 == 0 && !dr.drop_all) return 1.0f;
  if (dr.drop_all) return 0.0f;
  const uint32_t x = dropout_mix(dropout_row_part(dr, row) ^ (uint32_t)(col >> 1) * 0xC2B2AE3Du);
  return ((x >> (16 * (col & 1))) & 0x7fffu) >= dr.thr15 ? dr.keep_scale : 0.0f;
}

static inline AttnDropout make_dropout(float p, uint32_t seed, int h0, int h_total, const void* seed_dev = nullptr) {
  AttnDropout dr;
  dr.seed = seed;
  dr.seed_dev = static_cast<const uint32_t*>(seed_dev);
  dr.drop_all = p >= 1.0f ? 1 : 0;
  const double t = p <= 0.f ? 0.0 : (double)p * 32768.0 + 0.5;
  dr.thr15 = dr.drop_all ? 0x8000u : (uint32_t)(t > 32768.0 ? 32768.0 : t);
  dr.keep_scale = (p > 0.f && p < 1.0f) ? 1.0f / (1.0f - p) : (p >= 1.0f ? 0.f : 1.0f);
  dr.h0 = h0;
  dr.h_total = h_total;
  return dr;
}

template <int ATT_D, bool DROP = false>
// (D = 32: capped at 128 registers = two workgroups per CU, six spilled values, 0.79 -> 0.69 ms at S = 10 242; the same cap
//  at D = 64 spills 73 registers into the tile loop: 8.2 -> 23 ms, so D = 64 stays at one workgroup per CU)
// DROP (training, round 3): attention dropout on the packed probabilities -- the row sum is taken first (the normaliser
// sees every key), then the dropped halves of every bf16 pair are cleared by an AND with the pair's 2 x 16 decision bits;
// the 1 / (1 - p) rides on the final normalisation.  One hash per key pair and query: ~9 VALU per element next to the
// softmax's ~6 (two of them 32-bit multiplies).
__global__ __launch_bounds__(512, (ATT_D == 32 && !DROP) ? 4 : 2) void mhsa_bf16_kernel(
    const bf16_t* __restrict__ qkv, int64_t ld, const bf16_t* __restrict__ vt, bf16_t* __restrict__ out, int64_t ldo, int S,
    int S_pad, int H, int C, int window, float scale_log2e, float* __restrict__ lse, const AttnDropout dr_arg,
    const int* __restrict__ run_flag = nullptr) {  // optional: run only if *run_flag != 0 (fallback of mhsa_bf16_w4_kernel)
  static_assert(ATT_D == 32 || ATT_D == 64, "head sizes with an MFMA path");
  if (run_flag != nullptr && *run_flag == 0) return;
  const AttnDropout dr = dropout_resolve(dr_arg);
  constexpr int NKS = ATT_D / 16, NDT = ATT_D / 32;
  constexpr int KRB = ATT_D * 2;                        // bytes of a key row in the K tile
  constexpr int K_TILE = ATT_KV * KRB, V_TILE = ATT_D * 128;
  constexpr int ATT_STAGE = K_TILE + V_TILE;
  constexpr int N_STAGE = 3;  // ring of tile buffers: tile kt + 2 is requested while tile kt is computed
  constexpr int DMA_PER_STAGE = ATT_D == 64 ? 2 : 1;  // LDS-DMA instructions per wave and tile
  __shared__ __attribute__((aligned(16))) char smem[N_STAGE * ATT_STAGE];  // stages x (K tile + V^T tile)
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int h = blockIdx.y, b = blockIdx.z;
  const int qw0 = blockIdx.x * ATT_QBLK + wid * ATT_QPW;  // first query of this wave
  const int half = lane >> 5, ql = lane & 31;

  // ---- Q^T fragments (B operand) of the wave's two 32-query blocks: Q[q][ks*16 + half*8 .. +8]
  abf16x8_t qf[2][NKS];
  int qn[2];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    qn[qb] = qw0 + qb * 32 + ql;
    const int qc = qn[qb] < S ? qn[qb] : S - 1;
    const bf16_t* qp = qkv + ((int64_t)b * S + qc) * ld + h * ATT_D + half * 8;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) qf[qb][ks] = *reinterpret_cast<const abf16x8_t*>(qp + ks * 16);
  }

  uint32_t drow[2] = {0u, 0u};  // DROP: hash part of this lane's two queries
  if constexpr (DROP) {
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) drow[qb] = dropout_row_part(dr, dropout_row(dr, b, h, S, qn[qb]));
  }

  // ---- key range of this workgroup (sliding window: only tiles that intersect any of its queries)
  int kt_begin = 0, kt_end = (S + ATT_KV - 1) / ATT_KV;
  if (window >= 0) {
    const int lo = (int)blockIdx.x * ATT_QBLK - window, hi = (int)blockIdx.x * ATT_QBLK + ATT_QBLK - 1 + window;
    kt_begin = lo > 0 ? lo / ATT_KV : 0;
    const int e = hi / ATT_KV + 1;
    kt_end = e < kt_end ? e : kt_end;
  }

  // ---- staging: K tile rows = keys, V^T tile rows = d; 8 row groups (8 rows x 128 B) each: wave w moves group w
  const int srow = lane >> 3, scp = lane & 7;
  const bf16_t* kbase = qkv + (int64_t)b * S * ld + C + h * ATT_D;
  const bf16_t* vbase = vt + ((int64_t)b * H + h) * ATT_D * S_pad;
  const int sr = wid * 8 + srow;          // tile row staged by this lane
  const int sc = aswz(sr, scp);           // source chunk landing at LDS position scp
  auto stage = [&](int kt, int buf) {
    char* ks_ = smem + buf * ATT_STAGE;
    char* vs_ = ks_ + K_TILE;
    if constexpr (ATT_D == 64) {  // every wave: one 1 KiB piece of K (8 keys) and one of V^T (8 rows d)
      int key = kt * ATT_KV + sr;
      if (key > S - 1) key = S - 1;
      aglds16(reinterpret_cast<const char*>(kbase + (int64_t)key * ld) + sc * 16, ks_ + wid * 1024);
      aglds16(reinterpret_cast<const char*>(vbase + (int64_t)sr * S_pad + kt * ATT_KV) + sc * 16, vs_ + wid * 1024);
    } else if (wid < 4) {  // D = 32: waves 0..3 move K (16 keys x 64 bytes each) ...
      const int r = wid * 16 + (lane >> 2);
      int key = kt * ATT_KV + r;
      if (key > S - 1) key = S - 1;
      aglds16(reinterpret_cast<const char*>(kbase + (int64_t)key * ld) + aswz64(r, lane & 3) * 16, ks_ + wid * 1024);
    } else {  // ... waves 4..7 V^T (8 rows d x 128 bytes each)
      const int r = (wid - 4) * 8 + srow;
      aglds16(reinterpret_cast<const char*>(vbase + (int64_t)r * S_pad + kt * ATT_KV) + aswz(r, scp) * 16,
              vs_ + (wid - 4) * 1024);
    }
  };

  af32x16_t o_acc[2][NDT];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb)
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) o_acc[qb][dt][r] = 0.f;
  float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};

  // K row of MFMA row i: index bits 2 and 3 swapped
  const int kperm = (ql & 0x13) | ((ql & 4) << 1) | ((ql & 8) >> 1);

  if (kt_begin < kt_end) stage(kt_begin, 0);
  if (kt_begin + 1 < kt_end) stage(kt_begin + 1, 1);
  for (int kt = kt_begin; kt < kt_end; ++kt) {
    const int buf = (kt - kt_begin) % N_STAGE;
    // This wave's LDS-DMA of tile kt has to have LANDED before the barrier publishes the buffer; tile kt + 1 may stay in
    // flight (vmcnt counts this wave's requests in order, DMA_PER_STAGE per tile).  The compiler does not count an LDS-DMA
    // as a writer of the LDS read below: with a plain __syncthreads() it emitted vmcnt(0) once in front of the loop and
    // only lgkmcnt(0) at the barrier, so a tile whose DMA outlasted the previous tile's arithmetic was read half landed
    // (intermittent 10-40 % errors in single rows, ~5 % of the calls at B x H = 32 workgroups; found in round 2 by
    // tools/mhsa_stress.py).  Explicit counted waits, as in the GEMM kernels; the ring of three buffers keeps a whole
    // tile time between a request and its wait.
    if (kt + 1 < kt_end) {
      if constexpr (DMA_PER_STAGE == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();  // everyone's share of tile kt is in LDS; everyone is done with tile kt - 1 (the buffer refilled next)
    if (kt + 2 < kt_end) stage(kt + 2, (kt + 2 - kt_begin) % N_STAGE);
    const char* ks_ = smem + buf * ATT_STAGE;
    const char* vs_ = ks_ + K_TILE;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      // ---- K fragments of this 32-key block, shared by both query blocks
      const int krow = kb * 32 + kperm;
      abf16x8_t kf[NKS];
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        const int ch = ATT_D == 64 ? aswz(krow, ks * 2 + half) : aswz64(krow, ks * 2 + half);
        kf[ks] = *reinterpret_cast<const abf16x8_t*>(ks_ + krow * KRB + (ch << 4));
      }
      const int key0 = kt * ATT_KV + kb * 32 + 8 * half;
      const int blk0 = kt * ATT_KV + kb * 32;  // first key of this 32-key block
      abf16x8_t pb[2][2];
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) {
        // ---- S^T block: 32 keys x 32 queries (interleaving the two blocks' MFMA chains was measured 4 % slower)
        af32x16_t s_acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) s_acc[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
          s_acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf[qb][ks], s_acc, 0, 0, 0);
#if ANEMOI_LAB_MHSA8_PAD
        // lab (round 6, tools/isa_hazard_audit.py --all): hipcc leaves 5 ... 9 counted states between this product and its
        // first reader on the paths that cross taken branches (the unmasked fast path); padded here to the full 12
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 7\n\ts_nop 3" : "+v"(s_acc));  // (s_acc an operand: its readers stay below)
        __builtin_amdgcn_sched_barrier(0);
#endif
        // ---- online softmax in the log2 domain; register r <-> key key0 + (r & 7) + 16 (r >> 3).
        //      VALU budget per element: max, fma, exp2, add (the scale rides in the fma); masking only on tiles that
        //      touch the sequence end / window edge (wave-uniform test); O is rescaled only when the max grew.
        const int qfirst = qw0 + qb * 32;
        bool need_mask = blk0 + 32 > S;
        if (window >= 0) need_mask = need_mask || (blk0 + 31 - qfirst > window) || (qfirst + 31 - blk0 > window);
        if (need_mask) {
          const int q = qn[qb];
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = key0 + (r & 7) + 16 * (r >> 3);
            bool ok = key < S;
            if (window >= 0) ok = ok && (key - q <= window) && (q - key <= window);
            s_acc[r] = ok ? s_acc[r] : -INFINITY;
          }
        }
        float mloc = s_acc[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mloc = fmaxf(mloc, s_acc[r]);
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64)) * scale_log2e;  // scale > 0: max commutes with it
        if (mloc > m_run[qb]) {  // (-inf > -inf is false)
          const float corr = __builtin_amdgcn_exp2f(m_run[qb] - mloc);  // m_run = -inf -> 0
          l_run[qb] *= corr;
#pragma unroll
          for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o_acc[qb][dt][r] *= corr;
          m_run[qb] = mloc;
        }
        const float m_neg = m_run[qb] == -INFINITY ? 0.f : -m_run[qb];  // fully masked so far: exp2(-inf) = 0 below
        float p[16];
#pragma unroll
        for (int r = 0; r < 16; ++r)
          p[r] = __builtin_amdgcn_exp2f(fmaf(s_acc[r], scale_log2e, m_neg));  // bare v_exp_f32: argument <= 0
        // ---- P^T fragments: registers 0..7 / 8..15 are 8 contiguous keys each (v_cvt_pk_bf16_f32: 2 values / instr).
        //      The row sum is taken from the PACKED pairs (v_dot2_f32_bf16 with (1, 1): 8 instructions instead of 16
        //      adds -- the loop is bound by its VALU stream, tools/micro/attn_lab.py) and so normalises exactly the
        //      rounded probabilities the P V product multiplies.
        float psum = 0.f;
        const uint32_t ones2 = 0x3f803f80u;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          uint32_t w[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            // (DROP: the compiler's own conversion -- an opaque asm between the hash's VALU chains is invisible to its
            //  hazard recognizer: trans-result -> VALU read without the wait state gave NaN sums in single lanes)
            if constexpr (DROP) w[i] = pack_bf16x2(p[kk * 8 + 2 * i], p[kk * 8 + 2 * i + 1]);
            else asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w[i]) : "v"(p[kk * 8 + 2 * i]), "v"(p[kk * 8 + 2 * i + 1]));
            uint32_t wi = w[i];
            psum = __builtin_amdgcn_fdot2_f32_bf16(*reinterpret_cast<const bf16x2_t*>(&wi),
                                                   *reinterpret_cast<const bf16x2_t*>(&ones2), psum, false);
            if constexpr (DROP) {  // keys key0 + 16 kk + 2 i (low half) and + 1 (high half): pair index (key0 >> 1) + 8 kk + i
              const uint32_t x = dropout_mix(drow[qb] ^ ((uint32_t)(key0 >> 1) + 8u * kk + i) * 0xC2B2AE3Du);
              w[i] &= dropout_pair_mask(x, dropout_cthr2(dr.thr15));
            }
          }
          pb[qb][kk] = *reinterpret_cast<abf16x8_t*>(w);
        }
        l_run[qb] += psum;
      }
      // ---- O^T += V^T P^T : each V^T fragment feeds both query blocks
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) {
        const int vrow = dt * 32 + ql;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          const abf16x8_t vf = *reinterpret_cast<const abf16x8_t*>(
              vs_ + vrow * 128 + (aswz(vrow, kb * 4 + kk * 2 + half) << 4));
#pragma unroll
          for (int qb = 0; qb < 2; ++qb)
            o_acc[qb][dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pb[qb][kk], o_acc[qb][dt], 0, 0, 0);
        }
      }
    }
  }

  // ---- normalise and store: lane holds O[q][dt*32 + (r & 3) + 8 (r >> 2) + 4 half]
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const float l_tot = l_run[qb] + __shfl_xor(l_run[qb], 32, 64);
    float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
    if constexpr (DROP) inv *= dr.keep_scale;
    if (lse != nullptr && half == 0 && qn[qb] < S)  // natural-log sum-exp of the scaled scores (training: backward input)
      lse[((int64_t)b * H + h) * S + qn[qb]] = (m_run[qb] + __builtin_amdgcn_logf(l_tot)) * 0.69314718055994530942f;
    if (qn[qb] < S) {
      bf16_t* op = out + ((int64_t)b * S + qn[qb]) * ldo + h * ATT_D;
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float v4[4] = {o_acc[qb][dt][4 * g] * inv, o_acc[qb][dt][4 * g + 1] * inv,
                               o_acc[qb][dt][4 * g + 2] * inv, o_acc[qb][dt][4 * g + 3] * inv};
          VecIO<bf16_t, 4>::store(op + dt * 32 + 8 * g + 4 * half, v4);
        }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// D = 64, FOUR waves x 128 queries per workgroup (one wave per SIMD, the 512-register budget of the GEMM kernel), software
// pipelined by 32-key blocks: while the VALU runs the softmax of block j, the matrix pipe runs S^T of block j + 1 and the
// P V product of block j - 1 of the SAME wave (independent accumulators), instead of two waves per SIMD taking turns in
// the same phase (profiles/r02_mhsa_lab.md: VALU-active 55 % + MFMA-busy 34 % = 89 % of the cycles).  K / V^T fragments feed
// four query blocks (half the LDS -> register traffic per MFMA of the 8-wave kernel).  A tile's barrier sits between its two
// blocks: the K fragments of the next tile are first needed by the second block's matrix phase.
// ---------------------------------------------------------------------------------------------
constexpr int W4_WAVES = 4, W4_QB = 4, W4_QPW = 32 * W4_QB;  // 4 x 128 = the 8-wave kernel's 512 queries per workgroup

template <bool DROP>
__global__ __launch_bounds__(256) void mhsa_bf16_w4_kernel(const bf16_t* __restrict__ qkv, int64_t ld,
                                                           const bf16_t* __restrict__ vt, bf16_t* __restrict__ out,
                                                           int64_t ldo, int S, int S_pad, int H, int C,
                                                           float scale_log2e, float* __restrict__ lse, const AttnDropout dr_arg,
                                                           int* __restrict__ redo_flag) {
  // (round 6: a staggered start of the query blocks -- the launch is exactly five rounds of equal workgroups, all in lockstep --
  //  measured 6.92 ms per layer with and without, gpurun_out/r06_s19: not kept)
  const AttnDropout dr = dropout_resolve(dr_arg);
  constexpr int ATT_D = 64, NKS = 4, NDT = 2, KRB = 128;
  constexpr int K_TILE = ATT_KV * KRB, V_TILE = ATT_D * 128, ATT_STAGE = K_TILE + V_TILE;
  constexpr int N_STAGE = 4;  // tile kt + 2 is requested behind barrier kt; the V^T half of tile kt - 1 is still read there
  __shared__ __attribute__((aligned(16))) char smem[N_STAGE * ATT_STAGE];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int h = blockIdx.y, b = blockIdx.z;
  const int qw0 = blockIdx.x * ATT_QBLK + wid * W4_QPW;  // first query of this wave
  const int half = lane >> 5, ql = lane & 31;

  abf16x8_t qf[W4_QB][NKS];
  int qn[W4_QB];
#pragma unroll
  for (int qb = 0; qb < W4_QB; ++qb) {
    qn[qb] = qw0 + qb * 32 + ql;
    const int qc = qn[qb] < S ? qn[qb] : S - 1;
    const bf16_t* qp = qkv + ((int64_t)b * S + qc) * ld + h * ATT_D + half * 8;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      qf[qb][ks] = *reinterpret_cast<const abf16x8_t*>(qp + ks * 16);
    }
  }
  uint32_t drow[W4_QB] = {0u, 0u, 0u, 0u};
  if constexpr (DROP) {
#pragma unroll
    for (int qb = 0; qb < W4_QB; ++qb) drow[qb] = dropout_row_part(dr, dropout_row(dr, b, h, S, qn[qb]));
  }
  const int kt_begin = 0, kt_end = (S + ATT_KV - 1) / ATT_KV;
  // staging: wave w moves row groups 2 w and 2 w + 1 (8 rows x 128 B each) of the K tile and of the V^T tile
  const int srow = lane >> 3, scp = lane & 7;
  const bf16_t* kbase = qkv + (int64_t)b * S * ld + C + h * ATT_D;
  const bf16_t* vbase = vt + ((int64_t)b * H + h) * ATT_D * S_pad;
  // Buffer descriptors: the per-lane offsets are loop invariants, the tile enters as a scalar offset -- four LDS-DMA
  // instructions and two scalar multiplies per tile instead of four 64-bit address chains.  K rows behind the sequence
  // end lie outside the descriptor and read as zeros (their scores are masked anyway); the launcher keeps calls whose
  // q|k|v block of one batch element exceeds the 2 GiB descriptor range on the 8-wave kernel.
  const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<bf16_t*>(kbase), 0, (int)(((int64_t)(S - 1) * ld + ATT_D) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<bf16_t*>(vbase), 0, (int)((int64_t)ATT_D * S_pad * 2), 0x00020000);
  int koff[2], voff[2];
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const int sr = (wid * 2 + g) * 8 + srow;
    const int sc = aswz(sr, scp);
    koff[g] = sr * (int)ld * 2 + sc * 16;
    voff[g] = sr * S_pad * 2 + sc * 16;
  }
  const int ktile_bytes = ATT_KV * (int)ld * 2;
  auto stage = [&](int kt, int buf) {
    char* ks_ = smem + buf * ATT_STAGE;
    char* vs_ = ks_ + K_TILE;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(krs, (__attribute__((address_space(3))) void*)(ks_ + (wid * 2 + g) * 1024), 16,
                                               koff[g], kt * ktile_bytes, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(vrs, (__attribute__((address_space(3))) void*)(vs_ + (wid * 2 + g) * 1024), 16,
                                               voff[g], kt * (ATT_KV * 2), 0, 0);
    }
  };
  constexpr int DMA_PER_STAGE = 4;
  auto publish = [&](int kt) {  // barrier "kt": tile kt readable by everyone, tile kt + 2 requested
    if (kt + 1 < kt_end) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // __syncthreads(), fence included: here the compiler waits for vmcnt(0) in it, i.e. also for the DMA of tile kt + 1 that
    // the counted wait above would let stay in flight.  The bare s_barrier behind the counted wait was measured -- same
    // speed.  (Round 3 blamed it for results differing in the last bit of a few rows between identical calls and read that
    // as stale keys; round 5 found the cause in the prologue below -- maxima read in front of their wait states -- and four
    // variants of this synchronisation made no difference to it.  The ring's accounting was right all along; the stronger
    // barrier stays because it costs nothing.)
    __syncthreads();
    if (kt + 2 < kt_end) stage(kt + 2, (kt + 2 - kt_begin) % N_STAGE);
  };
  static_assert(DMA_PER_STAGE == 4, "the counted wait above");

  af32x16_t o_acc[W4_QB][NDT];
#pragma unroll
  for (int qb = 0; qb < W4_QB; ++qb)
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) o_acc[qb][dt][r] = 0.f;
#pragma unroll
  for (int qb = 0; qb < W4_QB; ++qb)
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) asm volatile("" : "+a"(o_acc[qb][dt]));
  float m_run[W4_QB], l_run[W4_QB];
#pragma unroll
  for (int qb = 0; qb < W4_QB; ++qb) m_run[qb] = -INFINITY, l_run[qb] = 0.f;
  af32x16_t lacc[2];   // row sums on the matrix pipe (!DROP), one tuple per PAIR: registers 0..7 block 0, 8..15 block 1
  abf16x8_t onesf[2];  // A fragments: ones in rows 0..15 / in rows 16..31
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const uint32_t one2 = ((ql >> 4) == i) ? 0x3f803f80u : 0u;
    uint32_t ow[4] = {one2, one2, one2, one2};
    onesf[i] = *reinterpret_cast<abf16x8_t*>(ow);
  }
#pragma unroll
  for (int pr = 0; pr < 2; ++pr) {
#pragma unroll
    for (int r = 0; r < 16; ++r) lacc[pr][r] = 0.f;
    asm volatile("" : "+a"(lacc[pr]));
  }
  const int kperm = (ql & 0x13) | ((ql & 4) << 1) | ((ql & 8) >> 1);  // K row of MFMA row i: index bits 2 and 3 swapped

  auto k_frags = [&](const char* ks_, int kb, abf16x8_t (&kf)[NKS]) {
    const int krow = kb * 32 + kperm;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) kf[ks] = *reinterpret_cast<const abf16x8_t*>(ks_ + krow * KRB + (aswz(krow, ks * 2 + half) << 4));
  };
  auto v_frags = [&](const char* vs_, int kb, abf16x8_t (&vf)[NDT][2]) {
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int vrow = dt * 32 + ql;
        vf[dt][kk] = *reinterpret_cast<const abf16x8_t*>(vs_ + vrow * 128 + (aswz(vrow, kb * 4 + kk * 2 + half) << 4));
      }
  };
  // The wave's four query blocks go through the pipeline as two PAIRS (pr = 0: blocks 0, 1; pr = 1: blocks 2, 3): the scores
  // of one pair are consumed by the softmax while the other pair's are produced -- 64 score registers alive, not 128.
  // A half step = 16 MFMAs (P V of one pair: 8, S^T of one pair: 8) issued BETWEEN the slices of one pair's softmax.  The MFMAs
  // are inline asm with fixed register classes -- scores in ordinary registers (VALU operands), O^T in the accumulation half
  // (MFMA-only) -- because the allocator, left alone, computes the scores into AGPRs, copies them out
  // one register at a time and spills the Q^T fragments (measured: 21 ms per layer).  Hazards the compiler cannot see
  // behind the asm: a score is read by the VALU a whole half step after its last MFMA; the accumulators are read by the
  // epilogue behind explicit s_nops.
#define W4_MFMA_S0(sv, av, bv) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(sv) : "v"(av), "a"(bv))
#define W4_MFMA_S(sv, av, bv) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(sv) : "v"(av), "a"(bv))
#define W4_MFMA_O(ov, av, bv) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(ov) : "v"(av), "v"(bv))
  // MFMA n (0 .. 15) of a half step: 0..7 = P V of pair PV_PR (skipped when !DO_PV), 8..15 = S^T of pair S_PR (!DO_S)
  auto half_step = [&](auto do_pv_c, auto do_s_c, auto pv_pr_c, auto s_pr_c, auto sm_pr_c, int kt, int kb,
                       const abf16x8_t (&vf)[NDT][2], const abf16x8_t (&pbo)[2][2], const abf16x8_t (&kf)[NKS],
                       af32x16_t (&sb)[2], af32x16_t (&sa)[2], abf16x8_t (&pbn)[2][2]) {
    constexpr bool DO_PV = decltype(do_pv_c)::value, DO_S = decltype(do_s_c)::value;
    constexpr int PV_PR = decltype(pv_pr_c)::value, S_PR = decltype(s_pr_c)::value, SM_PR = decltype(sm_pr_c)::value;
    (void)&o_acc; (void)&qf; (void)&lacc; (void)&onesf;  // (named once outside the asm operands: clang does not capture a variable it only meets there)
    // MFMA n of a half step: 0..7 = P V of pair PV_PR; then, interleaved, the four ROW-SUM products of the same
    // probabilities (8, 10, 12, 14) and S^T of pair S_PR (9, 11, 13, 15..19).  Row sums: a fragment of ones times P^T -- every
    // row of the 32 x 32 result = the column sums -- so the normaliser comes off the matrix pipe, which has slack, instead of
    // 16 v_dot2c per half step on the VALU, which is the bound (-0.6 ms per layer without them).  The two query blocks of a
    // pair share ONE accumulator tuple: the ones sit in rows 0..15 for block 0 and in rows 16..31 for block 1, i.e. the sums
    // land in registers 0..7 / 8..15 (four tuples would take the last free AGPRs, and an allocator short of registers moves
    // tuples around behind the asm MFMAs that just wrote them).  Their four accumulations are one dependency chain, hence the
    // S^T MFMAs between them.  Not with dropout: there the normaliser must see the probabilities BEFORE the mask, so the
    // DROP variant keeps the dot products.
#define W4_MF(n)                                                                                                     \
  do {                                                                                                               \
    if constexpr ((n) < 8) {                                                                                         \
      if constexpr (DO_PV)                                                                                           \
        W4_MFMA_O(o_acc[2 * PV_PR + ((n) & 1)][(n) >> 2], vf[(n) >> 2][((n) >> 1) & 1], pbo[(n) & 1][((n) >> 1) & 1]); \
    } else if constexpr ((n) < 16 && ((n) & 1) == 0) {                                                               \
      constexpr int lj_ = ((n) - 8) >> 1; /* 0 .. 3: block i = lj & 1, fragment kk = lj >> 1 */                      \
      if constexpr (DO_PV && !DROP) W4_MFMA_O(lacc[PV_PR], onesf[lj_ & 1], pbo[lj_ & 1][lj_ >> 1]);                  \
    } else if constexpr (DO_S) {                                                                                     \
      constexpr int st_ = (n) < 16 ? ((n) - 9) >> 1 : (n) - 12; /* 0 .. 7: block i = st & 1, k step = st >> 1 */     \
      if constexpr (st_ < 2) W4_MFMA_S0(sb[st_ & 1], kf[0], qf[2 * S_PR + (st_ & 1)][0]);                            \
      else W4_MFMA_S(sb[st_ & 1], kf[st_ >> 1], qf[2 * S_PR + (st_ & 1)][st_ >> 1]);                                 \
    }                                                                                                                \
  } while (0)
    // ---- softmax of pair SM_PR on sa, slices of it between the MFMAs
    const int key0 = kt * ATT_KV + kb * 32 + 8 * half, blk0 = kt * ATT_KV + kb * 32;
    if (blk0 + 32 > S) {  // the sequence ends inside this block (global attention only: the launcher keeps windows off this kernel)
      // the scores' last MFMA (asm: no hazard handling by the compiler) has retired.  The scores are OPERANDS of the wait:
      // a bare asm statement does not order the register reads below against itself (see the prologue)
      asm volatile("s_nop 15" : "+v"(sa[0]), "+v"(sa[1]));
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) sa[i][r] = key0 + (r & 7) + 16 * (r >> 3) < S ? sa[i][r] : -INFINITY;
    }
    // NO running maximum in this loop: the reference of a query is the maximum of its FIRST 32-key block (taken in the
    // prologue below), fixed for the whole pass.  Later probabilities may exceed 1 (by 2^(later maximum - first maximum)), which f32 sums and bf16
    // operands carry without loss as long as that stays below ~2^100; the final o / l does not care which reference was
    // used.  Nothing rescales the 128 accumulator registers, no branch splits the matrix phase.  The pathological case
    // (a later score more than ~70 above the first block's maximum) shows as a non-finite row sum at the end and raises
    // the call's fallback flag: the launcher's second kernel (the 8-wave kernel with the exact online maximum) then
    // recomputes the call -- it exits at once when the flag is clear.
    float m_neg[2], psum[2] = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 2; ++i) m_neg[i] = -m_run[2 * SM_PR + i];
    uint32_t w[2][2][4];
    // TWENTY slots, one MFMA each (the matrix pipe takes one MFMA of a wave at a time: an MFMA with no VALU work behind it
    // is 32 idle issue cycles), and the softmax of the 16 packed pairs of the two query blocks (pair c: block i = c >> 3,
    // fragment kk = (c >> 2) & 1, word j = c & 3) cut to fit them:
    //     4 F slots: the eight fma (exp2 arguments) of the NEXT four pairs                      -- 32 cycles
    //    16 E slots: the two exp2 of pair n, pack (+ dropout: row sum) of pair n - 1            -- 36 cycles
    // in the order F E E E E F E E E E ...; every instruction reads what an EARLIER slot produced.  The VALU side is inline
    // asm as well: pure arithmetic is not ordered against the (volatile) MFMA asm, and instruction selection moves ALL of it
    // in front of or behind the run of MFMAs -- the scheduling fences only hold what selection already put between them; two
    // volatile asm statements stay in source order.  Early-clobber outputs: a result must not land in a register a later
    // instruction of the same statement still reads.
    float ta[4][2], tb[2][2];  // exp2 arguments of the current group of four pairs; exp2 results [slot parity]
    const float sc = scale_log2e;
#define W4_DOT2C_DROP "v_dot2c_f32_bf16 %[ps], 0x3f803f80, %[w]"
#define W4_DOT2C_NONE "s_nop 0"
    auto f_slot = [&](auto g_c) {  // pairs 4 g .. 4 g + 3
      constexpr int g = decltype(g_c)::value, i = g >> 1, e0 = (g & 1) * 8;  // elements e0 .. e0 + 7 of sa[i]
      (void)&ta; (void)&sa; (void)&m_neg; (void)&sc;  // (clang does not capture what it only meets as an asm operand)
      asm volatile("v_fma_f32 %[u0], %[s0], %[sc], %[mn]\n\t"
                   "v_fma_f32 %[u1], %[s1], %[sc], %[mn]\n\t"
                   "v_fma_f32 %[u2], %[s2], %[sc], %[mn]\n\t"
                   "v_fma_f32 %[u3], %[s3], %[sc], %[mn]\n\t"
                   "v_fma_f32 %[u4], %[s4], %[sc], %[mn]\n\t"
                   "v_fma_f32 %[u5], %[s5], %[sc], %[mn]\n\t"
                   "v_fma_f32 %[u6], %[s6], %[sc], %[mn]\n\t"
                   "v_fma_f32 %[u7], %[s7], %[sc], %[mn]"
                   : [u0] "=&v"(ta[0][0]), [u1] "=&v"(ta[0][1]), [u2] "=&v"(ta[1][0]), [u3] "=&v"(ta[1][1]),
                     [u4] "=&v"(ta[2][0]), [u5] "=&v"(ta[2][1]), [u6] "=&v"(ta[3][0]), [u7] "=&v"(ta[3][1])
                   : [s0] "v"(sa[i][e0]), [s1] "v"(sa[i][e0 + 1]), [s2] "v"(sa[i][e0 + 2]), [s3] "v"(sa[i][e0 + 3]),
                     [s4] "v"(sa[i][e0 + 4]), [s5] "v"(sa[i][e0 + 5]), [s6] "v"(sa[i][e0 + 6]), [s7] "v"(sa[i][e0 + 7]),
                     [sc] "s"(sc), [mn] "v"(m_neg[i]));
    };
    auto e_slot = [&](auto n_c) {
      constexpr int n = decltype(n_c)::value;
      constexpr int pi = (n - 1) >> 3, pkk = ((n - 1) >> 2) & 1, pj = (n - 1) & 3;  // pair n - 1: pack (+ row sum)
      (void)&ta; (void)&tb; (void)&w; (void)&psum;
#define W4_E_FIRST                                                   \
  asm volatile("v_exp_f32 %[f0], %[t0]\n\t"                          \
               "v_exp_f32 %[f1], %[t1]"                              \
               : [f0] "=&v"(tb[0][0]), [f1] "=&v"(tb[0][1])          \
               : [t0] "v"(ta[0][0]), [t1] "v"(ta[0][1]))
#define W4_E(DOT)                                                                                                    \
  asm volatile("v_cvt_pk_bf16_f32 %[w], %[e0], %[e1]\n\t"                                                           \
               "v_exp_f32 %[f0], %[t0]\n\t"                                                                         \
               "v_exp_f32 %[f1], %[t1]\n\t" DOT                                                                     \
               : [w] "=&v"(w[pi][pkk][pj]), [f0] "=&v"(tb[n & 1][0]), [f1] "=&v"(tb[n & 1][1]), [ps] "+v"(psum[pi]) \
               : [e0] "v"(tb[(n - 1) & 1][0]), [e1] "v"(tb[(n - 1) & 1][1]), [t0] "v"(ta[n & 3][0]), [t1] "v"(ta[n & 3][1]))
      if constexpr (n == 0) {
        W4_E_FIRST;
      } else {
        if constexpr (DROP) W4_E(W4_DOT2C_DROP);
        else W4_E(W4_DOT2C_NONE);
      }
#undef W4_E_FIRST
#undef W4_E
      if constexpr (DROP && n > 0) {
        const uint32_t x = dropout_mix(drow[2 * SM_PR + pi] ^ ((uint32_t)(key0 >> 1) + 8u * pkk + pj) * 0xC2B2AE3Du);
        w[pi][pkk][pj] &= dropout_pair_mask(x, dropout_cthr2(dr.thr15));
      }
    };
    // the last pair's pack (+ row sum); wait states: the exponentials right above, and a dot product's result is not
    // forwarded to an ordinary VALU read right behind it (the compiler sees neither hazard inside the asm)
    auto tail = [&]() {
      (void)&tb; (void)&w; (void)&psum;
#define W4_TAIL(DOT)                                                  \
  asm volatile("s_nop 0\n\t"                                         \
               "v_cvt_pk_bf16_f32 %[w], %[e0], %[e1]\n\t"            \
               "s_nop 0\n\t" DOT "\n\t"                              \
               "s_nop 3"                                              \
               : [w] "=&v"(w[1][1][3]), [ps] "+v"(psum[1])            \
               : [e0] "v"(tb[1][0]), [e1] "v"(tb[1][1]))
      if constexpr (DROP) W4_TAIL(W4_DOT2C_DROP);
      else W4_TAIL(W4_DOT2C_NONE);
#undef W4_TAIL
      if constexpr (DROP) {
        const uint32_t x = dropout_mix(drow[2 * SM_PR + 1] ^ ((uint32_t)(key0 >> 1) + 8u + 3u) * 0xC2B2AE3Du);
        w[1][1][3] &= dropout_pair_mask(x, dropout_cthr2(dr.thr15));
      }
    };
    asm volatile("s_nop 3" ::: "memory");  // (sa[0]'s last MFMA is two slots back: these wait states on top)
#define W4_F(g) f_slot(std::integral_constant<int, (g)>{})
#define W4_E(n) e_slot(std::integral_constant<int, (n)>{})
    // (20 MFMAs in numeric order: two accumulations into the same block always have another MFMA between them -- back to
    //  back, the second would read its accumulator before the first has written it, and behind asm nobody inserts the wait)
    W4_MF(0);  W4_F(0);
    W4_MF(1);  W4_E(0);
    W4_MF(2);  W4_E(1);
    W4_MF(3);  W4_E(2);
    W4_MF(4);  W4_E(3);
    W4_MF(5);  W4_F(1);
    W4_MF(6);  W4_E(4);
    W4_MF(7);  W4_E(5);
    W4_MF(8);  W4_E(6);
    W4_MF(9);  W4_E(7);
    W4_MF(10); W4_F(2);
    W4_MF(11); W4_E(8);
    W4_MF(12); W4_E(9);
    W4_MF(13); W4_E(10);
    W4_MF(14); W4_E(11);
    W4_MF(15); W4_F(3);
    W4_MF(16); W4_E(12);
    W4_MF(17); W4_E(13);
    W4_MF(18); W4_E(14);
    W4_MF(19); W4_E(15);
    tail();
#undef W4_F
#undef W4_E
#undef W4_DOT2C_DROP
#undef W4_DOT2C_NONE
#undef W4_MF
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) pbn[i][kk] = *reinterpret_cast<abf16x8_t*>(w[i][kk]);
      if constexpr (DROP) l_run[2 * SM_PR + i] += psum[i];
    }
  };
  using T_ = std::true_type;
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;

  if (kt_begin < kt_end) {
    stage(kt_begin, 0);
    if (kt_begin + 1 < kt_end) stage(kt_begin + 1, 1);
    publish(kt_begin);
    af32x16_t s0[2], s1[2];            // scores of pair 0 / pair 1
    abf16x8_t p0[2][2], p1[2][2];      // packed probabilities of pair 0 / pair 1
    abf16x8_t kf0[NKS], kf1[NKS], vf0[NDT][2], vf1[NDT][2];  // fragments of the tile's block 0 / block 1
    // Half steps of block j (tile kt, kb):   A(j): matrix  P V (j - 1, pair 1), S^T (j, pair 1);      VALU  softmax (j, pair 0)
    //                                        B(j): matrix  P V (j, pair 0),     S^T (j + 1, pair 0);  VALU  softmax (j, pair 1)
    k_frags(smem, 0, kf0);
    // prologue: S^T of the first block for BOTH pairs -> the queries' reference maxima (pair 1's scores are produced again
    // by the first half step: eight MFMAs once per workgroup); pair 0's stay in s0 for the first softmax
#pragma unroll
    for (int pr = 1; pr >= 0; --pr) {
      W4_MFMA_S0(s0[0], kf0[0], qf[2 * pr][0]);
      W4_MFMA_S0(s0[1], kf0[0], qf[2 * pr + 1][0]);
#pragma unroll
      for (int ks = 1; ks < NKS; ++ks) {
        W4_MFMA_S(s0[0], kf0[ks], qf[2 * pr][ks]);
        W4_MFMA_S(s0[1], kf0[ks], qf[2 * pr + 1][ks]);
      }
      // asm MFMAs: no hazard handling by the compiler, and the wait states must take the scores as OPERANDS.  Behind a bare
      // `asm volatile("s_nop ..." ::: "memory")` the compiler hoisted the maxima in FRONT of the wait (pure register
      // arithmetic is not ordered against a volatile asm, "memory" or not): v_max read a score register right behind the MFMA
      // that writes it, i.e. the reference maximum was taken from whatever the register held at that cycle.  Any reference
      // gives the same softmax up to rounding -- no parity test can see it -- but WHICH value was read depended on the
      // wave's timing (instruction fetch under contention): results differing in the last bit between identical calls
      // (round 5, found by bit-comparing repeated calls with a second process on the GPU, tools/micro/mhsa_repeat_diag.py).
      asm volatile("s_nop 15\n\ts_nop 15" : "+v"(s0[0]), "+v"(s0[1]));
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        float m = s0[i][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) m = fmaxf(m, 8 * half + (r & 7) + 16 * (r >> 3) < S ? s0[i][r] : -INFINITY);
        m_run[2 * pr + i] = fmaxf(m, __shfl_xor(m, 32, 64)) * scale_log2e;
      }
    }
    // ONE loop body, no branch around a half step (a branch there makes the allocator copy accumulator tuples at the join):
    // the P V product "of the block before the first" multiplies zeros, the S^T product "of the block behind the last"
    // multiplies the stale K fragments into scores nobody reads.
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        p1[i][kk] = abf16x8_t{0, 0, 0, 0, 0, 0, 0, 0};
        vf1[i][kk] = abf16x8_t{0, 0, 0, 0, 0, 0, 0, 0};
      }
    for (int kt = kt_begin; kt < kt_end; ++kt) {
      const char* ks_ = smem + ((kt - kt_begin) % N_STAGE) * ATT_STAGE;
      const char* vs_ = ks_ + K_TILE;
      // ---- A(2k)      [vf1 / p1: block 2k - 1]
      k_frags(ks_, 1, kf1);
      v_frags(vs_, 0, vf0);
      half_step(T_{}, T_{}, I1{}, I1{}, I0{}, kt, 0, vf1, p1, kf0, s1, s0, p0);
      // ---- B(2k)
      half_step(T_{}, T_{}, I0{}, I0{}, I1{}, kt, 0, vf0, p0, kf1, s0, s1, p1);
      // ---- the next tile's K fragments are needed by B(2k + 1)
      if (kt + 1 < kt_end) {
        publish(kt + 1);
        k_frags(smem + ((kt + 1 - kt_begin) % N_STAGE) * ATT_STAGE, 0, kf0);
      }
      v_frags(vs_, 1, vf1);
      // ---- A(2k + 1)
      half_step(T_{}, T_{}, I1{}, I1{}, I0{}, kt, 1, vf0, p1, kf1, s1, s0, p0);
      // ---- B(2k + 1)
      half_step(T_{}, T_{}, I0{}, I0{}, I1{}, kt, 1, vf1, p0, kf0, s0, s1, p1);
    }
    // epilogue: P V (last block, pair 1)
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < 2; ++i) W4_MFMA_O(o_acc[2 + i][dt], vf1[dt][kk], p1[i][kk]);
    if constexpr (!DROP) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // (one dependency chain, nothing between its links here)
          W4_MFMA_O(lacc[1], onesf[i], p1[i][kk]);
        }
    }
    // (last asm MFMAs -> the accumulator reads of the epilogue: the accumulators are operands of the wait, see the prologue)
    asm volatile("s_nop 15\n\ts_nop 15"
                 : "+a"(o_acc[0][0]), "+a"(o_acc[0][1]), "+a"(o_acc[1][0]), "+a"(o_acc[1][1]), "+a"(o_acc[2][0]),
                   "+a"(o_acc[2][1]), "+a"(o_acc[3][0]), "+a"(o_acc[3][1]), "+a"(lacc[0]), "+a"(lacc[1]));
  }
#undef W4_MFMA_S0
#undef W4_MFMA_S
#undef W4_MFMA_O

#pragma unroll
  for (int qb = 0; qb < W4_QB; ++qb) {
    // (the MFMA row sums already span both K halves of every block: no exchange between the lane halves)
    const float l_tot = DROP ? l_run[qb] + __shfl_xor(l_run[qb], 32, 64) : lacc[qb >> 1][(qb & 1) * 8];
    if (!(l_tot < 1e37f) && qn[qb] < S) *redo_flag = 1;  // (also NaN) a probability left the f32 range: see the half step
    float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
    if constexpr (DROP) inv *= dr.keep_scale;
    if (lse != nullptr && half == 0 && qn[qb] < S)
      lse[((int64_t)b * H + h) * S + qn[qb]] = (m_run[qb] + __builtin_amdgcn_logf(l_tot)) * 0.69314718055994530942f;
    if (qn[qb] < S) {
      bf16_t* op = out + ((int64_t)b * S + qn[qb]) * ldo + h * ATT_D;
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float v4[4] = {o_acc[qb][dt][4 * g] * inv, o_acc[qb][dt][4 * g + 1] * inv,
                               o_acc[qb][dt][4 * g + 2] * inv, o_acc[qb][dt][4 * g + 3] * inv};
          VecIO<bf16_t, 4>::store(op + dt * 32 + 8 * g + 4 * half, v4);
        }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Generic kernel (f32 or bf16 storage, any D <= 128): one wave per (query, head); lane l takes keys l, l+64, ...
// with its own running (max, sum, acc[D]); the 64 partial states are merged at the end.
// ---------------------------------------------------------------------------------------------
template <typename T, int DMAX>
__global__ __launch_bounds__(256) void mhsa_generic_kernel(const T* __restrict__ qkv, int64_t ld, T* __restrict__ out,
                                                           int64_t ldo, int S, int H, int D, int C, int window,
                                                           float scale, int64_t total, float* __restrict__ lse,
                                                           const AttnDropout dr_arg, int q_begin, int q_count) {
  const AttnDropout dr = dropout_resolve(dr_arg);
  // queries [q_begin, q_begin + q_count) of every batch element (the whole sequence, or the few rows the MFMA kernel's
  // 512-query blocks leave over)
  const int lane = threadIdx.x & 63;
  const int64_t unit = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);  // (b, q, h)
  if (unit >= total) return;
  const int h = (int)(unit % H);
  const int64_t bql = unit / H;
  const int q = q_begin + (int)(bql % q_count);
  const int64_t b = bql / q_count;
  const int64_t bq = b * S + q;
  const T* qp = qkv + bq * ld + h * D;
  float qv[DMAX], acc[DMAX];
#pragma unroll
  for (int d = 0; d < DMAX; ++d) {
    qv[d] = d < D ? Elem<T>::load(qp + d) * scale : 0.f;
    acc[d] = 0.f;
  }
  float m = -INFINITY, l = 0.f;
  int k_lo = 0, k_hi = S;
  if (window >= 0) {
    k_lo = q - window > 0 ? q - window : 0;
    k_hi = q + window + 1 < S ? q + window + 1 : S;
  }
  for (int key = k_lo + lane; key < k_hi; key += 64) {
    const T* kp = qkv + (b * S + key) * ld + C + h * D;
    const T* vp = kp + C;
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < DMAX; ++d)
      if (d < D) s = fmaf(qv[d], Elem<T>::load(kp + d), s);
    const float mn = fmaxf(m, s);
    const float corr = __expf(m - mn), pe = __expf(s - mn);
    l = l * corr + pe;  // the normaliser sees every key; dropout acts on the normalised probabilities
    const float pk = pe * dropout_keep(dr, dropout_row(dr, b, h, S, q), key);
#pragma unroll
    for (int d = 0; d < DMAX; ++d)
      if (d < D) acc[d] = acc[d] * corr + pk * Elem<T>::load(vp + d);
    m = mn;
  }
  // merge the 64 lane states
  float mt = m;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mt = fmaxf(mt, __shfl_xor(mt, off, 64));
  const float w = m == -INFINITY ? 0.f : __expf(m - mt);
  const float lt = wave_sum(l * w);
  const float inv = lt > 0.f ? 1.0f / lt : 0.f;
  if (lse != nullptr && lane == 0) lse[(b * H + h) * S + q] = mt + __logf(lt);
  T* op = out + bq * ldo + h * D;
#pragma unroll
  for (int d = 0; d < DMAX; ++d) {
    if (d < D) {
      const float o = wave_sum(acc[d] * w) * inv;
      if (lane == 0) Elem<T>::store(op + d, o);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// The few queries the MFMA kernel's 512-query blocks leave over (S = 10 * 4^k + 2 on refined icosahedral meshes): a
// workgroup of the MFMA kernel streams ALL keys whatever its number of queries, so 2 left-over rows x 16 heads were a sixth
// round of workgroups on 256 CUs (+16 % per layer at S = 40 962).  Here the key range of every (batch, query, head) is
// split over n_split waves (flash-decoding style): a wave runs the online softmax of its key chunk (lane = key, 16-byte
// loads) and leaves { max, sum, acc[D] } in a workspace; a second kernel merges the chunks.  bf16, D = 32 / 64.
// ---------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256) void mhsa_tail_kernel(const bf16_t* __restrict__ qkv, int64_t ld, int S, int H, int C,
                                                        int window, float scale, int q_begin, int q_count, int n_split,
                                                        int chunk, int64_t total, float* __restrict__ part,
                                                        const AttnDropout dr_arg) {
  const AttnDropout dr = dropout_resolve(dr_arg);
  const int lane = threadIdx.x & 63;
  const int64_t unit = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);  // (b, q, h, split)
  if (unit >= total) return;
  // unit = ((b H + h) n_split + split) q_count + query: the left-over queries of one (head, key chunk) sit in neighbouring
  // waves, so the chunk's K / V rows are fetched from HBM once and hit the caches for the others (query-slowest order read
  // the whole K / V of every head once PER left-over query: 670 MB at S = 40 962 for two queries)
  const int qi = (int)(unit % q_count);
  const int64_t u1 = unit / q_count;
  const int split = (int)(u1 % n_split);
  const int64_t u2 = u1 / n_split;
  const int h = (int)(u2 % H);
  const int64_t b = u2 / H;
  const int q = q_begin + qi;
  float qv[D], acc[D];
  {
    const bf16_t* qp = qkv + (b * S + q) * ld + h * D;
#pragma unroll
    for (int d8 = 0; d8 < D / 8; ++d8) {
      float t[8];
      VecIO<bf16_t, 8>::load(qp + d8 * 8, t);
#pragma unroll
      for (int i = 0; i < 8; ++i) qv[d8 * 8 + i] = t[i] * scale;
    }
  }
#pragma unroll
  for (int d = 0; d < D; ++d) acc[d] = 0.f;
  float m = -INFINITY, l = 0.f;
  const int64_t drow = dropout_row(dr, b, h, S, q);
  int k_lo = split * chunk, k_hi = k_lo + chunk < S ? k_lo + chunk : S;
  if (window >= 0) {
    k_lo = k_lo > q - window ? k_lo : q - window;
    k_hi = k_hi < q + window + 1 ? k_hi : q + window + 1;
  }
  for (int key = k_lo + lane; key < k_hi; key += 64) {
    const bf16_t* kp = qkv + (b * S + key) * ld + C + h * D;
    const bf16_t* vp = kp + C;
    float s = 0.f;
#pragma unroll
    for (int d8 = 0; d8 < D / 8; ++d8) {
      float t[8];
      VecIO<bf16_t, 8>::load(kp + d8 * 8, t);
#pragma unroll
      for (int i = 0; i < 8; ++i) s = fmaf(qv[d8 * 8 + i], t[i], s);
    }
    const float mn = fmaxf(m, s);
    const float corr = __expf(m - mn), pe = __expf(s - mn);
    l = l * corr + pe;
    const float pk = pe * dropout_keep(dr, drow, key);  // (1 without dropout)
#pragma unroll
    for (int d8 = 0; d8 < D / 8; ++d8) {
      float t[8];
      VecIO<bf16_t, 8>::load(vp + d8 * 8, t);
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[d8 * 8 + i] = acc[d8 * 8 + i] * corr + pk * t[i];
    }
    m = mn;
  }
  float mt = m;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mt = fmaxf(mt, __shfl_xor(mt, off, 64));
  const float w = m == -INFINITY ? 0.f : __expf(m - mt);
  const float lt = wave_sum(l * w);
  float* pp = part + unit * (D + 2);
  if (lane == 0) {
    pp[0] = mt;
    pp[1] = lt;
  }
#pragma unroll
  for (int d = 0; d < D; ++d) {
    const float o = wave_sum(acc[d] * w);
    if (lane == 0) pp[2 + d] = o;
  }
}

template <int D>
__global__ __launch_bounds__(256) void mhsa_tail_merge_kernel(const float* __restrict__ part, bf16_t* __restrict__ out,
                                                              int64_t ldo, float* __restrict__ lse, int S, int H,
                                                              int q_begin, int q_count, int n_split, int64_t n_units) {
  // one WAVE per (b, h, query): the lanes hold the n_split <= 128 partial states (two per lane), the merge is a handful of
  // wave reductions (the first version walked the splits one after the other in every thread: 67 us for 32 rows)
  const int lane = threadIdx.x & 63;
  const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= n_units) return;
  const int qi = (int)(w % q_count);
  const int64_t bh = w / q_count;  // b H + h
  const int h = (int)(bh % H);
  const int64_t b = bh / H;
  const int q = q_begin + qi;
  // state of (split s): part[((bh n_split + s) q_count + qi) (D + 2) ...]
  const float* p0 = lane < n_split ? part + ((bh * n_split + lane) * q_count + qi) * (D + 2) : nullptr;
  const float* p1 = lane + 64 < n_split ? part + ((bh * n_split + lane + 64) * q_count + qi) * (D + 2) : nullptr;
  const float m0 = p0 != nullptr ? p0[0] : -INFINITY, m1 = p1 != nullptr ? p1[0] : -INFINITY;
  float mt = fmaxf(m0, m1);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mt = fmaxf(mt, __shfl_xor(mt, off, 64));
  const float w0 = m0 == -INFINITY ? 0.f : __expf(m0 - mt), w1 = m1 == -INFINITY ? 0.f : __expf(m1 - mt);
  const float lt = wave_sum((p0 != nullptr ? p0[1] * w0 : 0.f) + (p1 != nullptr ? p1[1] * w1 : 0.f));
  const float inv = lt > 0.f ? 1.0f / lt : 0.f;
  float mine = 0.f;  // lane d keeps output channel d
  const float* r0 = p0 != nullptr ? p0 + 2 : part;  // (lanes without a state: any valid address, the value is discarded)
  const float* r1 = p1 != nullptr ? p1 + 2 : part;
  for (int d0 = 0; d0 < D; d0 += 8) {  // eight channels' loads in flight per round trip
    float t[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float a0 = r0[d0 + i], a1 = r1[d0 + i];
      t[i] = (p0 != nullptr ? a0 * w0 : 0.f) + (p1 != nullptr ? a1 * w1 : 0.f);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float o = wave_sum(t[i]);
      if (lane == d0 + i) mine = o;
    }
  }
  if (lane < D) Elem<bf16_t>::store(out + (b * S + q) * ldo + h * D + lane, mine * inv);
  if (lse != nullptr && lane == 0) lse[(b * H + h) * S + q] = mt + __logf(lt);
}

// left-over rows of the MFMA path and the split of their key range (0 rows: the MFMA kernel takes everything)
static inline int mhsa_tail_rows(int S) {
  const int rem = S % 512;
  return (S > 512 && rem > 0 && rem <= 16) ? rem : 0;
}
static inline int mhsa_tail_splits(int B, int S, int H) {
  const int rem = mhsa_tail_rows(S);
  if (rem == 0) return 0;
  int64_t n = 4096 / ((int64_t)B * rem * H);
  const int64_t max_split = (S + 255) / 256;  // at least 256 keys per wave
  if (n > max_split) n = max_split;
  if (n > 128) n = 128;  // the merge kernel holds two partial states per lane
  return (int)(n < 1 ? 1 : n);
}

// ---------------------------------------------------------------------------------------------
// MFMA backward (bf16, D = 64 / 32): two kernels, no atomics.
//   delta_i = dO_i . O_i                                              (mhsa_delta_kernel)
//   dKV kernel: a wave owns 32 keys (K / V fragments and the dK^T / dV^T accumulators in registers), the workgroup's four
//     waves share 32-query tiles in LDS.  Scores are produced as S = Q K^T (queries in the accumulator registers, keys in
//     the lanes; the query rows are fed in the forward's bit-2/3-swapped order, so registers 0..7 / 8..15 are 8 contiguous
//     queries each = the B-operand fragments of the two products that follow):
//         P = exp2(S c - lse),  dP = dO V^T,  dS = P (dP - delta),   dV^T += dO^T P,   dK^T += Q^T dS.
//   dQ kernel: a wave owns 32 queries; S^T = K Q^T exactly as in the forward (keys in the registers),
//         dP^T = V dO^T,  dS^T = P^T (dP^T - delta),   dQ^T += K^T dS^T.
//   The A operands with the reduction along the sequence (dO^T, Q^T, K^T) come from transposed copies [B, H, D, S_pad]
//   made once per call (transpose_v_kernel), as V^T in the forward.  Tiles travel through an LDS ring by LDS-DMA (16 / 12
//   MFMAs per 32 x 32 block pair against the forward's 8).  The VALU kernels below took 46 s per layer at S = 40 962
//   (config 3) -- unusable; they remain for f32 and other head sizes.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mhsa_delta_kernel(const bf16_t* __restrict__ o, int64_t ldo,
                                                         const bf16_t* __restrict__ dout, int64_t lddo,
                                                         float* __restrict__ delta, const float* __restrict__ lse,
                                                         float* __restrict__ lse2p, float* __restrict__ deltap, int S,
                                                         int S_pad, int H, int D, int64_t total) {
  // lse2p / deltap (optional, [B, H, S_pad]): the log-sum-exp in log2 units and delta as the dK/dV kernel's LDS-DMA reads
  // them, padded behind S with +inf (P = 0 for the rows of a ragged last query tile) and 0
  const int lane = threadIdx.x & 63;
  const int64_t unit = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);  // (b, q, h)
  if (unit >= total) return;
  const int h = (int)(unit % H);
  const int64_t bq = unit / H;
  const int q = (int)(bq % S);
  const int64_t b = bq / S;
  float v = 0.f;
  if (lane < D) v = Elem<bf16_t>::load(o + bq * ldo + h * D + lane) * Elem<bf16_t>::load(dout + bq * lddo + h * D + lane);
  v = wave_sum(v);
  if (lane == 0) delta[(b * H + h) * S + q] = v;
  if (lse2p != nullptr) {
    const int64_t pr = (b * H + h) * S_pad;
    if (lane == 0) {
      lse2p[pr + q] = lse[(b * H + h) * S + q] * 1.44269504088896340736f;
      deltap[pr + q] = v;
    } else if (q == S - 1 && lane <= S_pad - S) {  // S_pad - S <= 63 pad entries
      lse2p[pr + q + lane] = INFINITY;
      deltap[pr + q + lane] = 0.f;
    }
  }
}

__device__ __forceinline__ int att_row_perm(int ql) { return (ql & 0x13) | ((ql & 4) << 1) | ((ql & 8) >> 1); }

template <int ATT_D>
__device__ __forceinline__ int att_rswz(int row, int chunk) {  // swizzle of a row-major [rows x D] bf16 tile
  return ATT_D == 64 ? aswz(row, chunk) : aswz64(row, chunk);
}

// All products of the two kernels are inline-asm MFMAs -- the accumulating ones (dV^T, dK^T, dQ^T) on accumulators pinned
// to AGPRs, the score-side ones (S, dP) on VGPR accumulators the softmax arithmetic reads in place -- each with wait
// states around it, and products on the same accumulator never back to back.  Found in round 3: with the MFMA builtin
// the compiler kept dQ^T in VGPRs across the tile loop (32 + 32 v_accvgpr moves per tile around the two dQ^T MFMAs of each
// accumulator), and the dQ rows of accumulator 0 (d = 0 .. 31) came out wrong at D = 64 -- deterministically, for every
// S, the other accumulator and dK / dV right (tools/micro/mhsa_bwd_dbg.py).  Extra barriers, s_sleep or a full vmcnt(0)
// did not change a digit, a second __syncthreads() between the staging and the products (i.e. another local schedule)
// did, so it is a property of that instruction sequence, not a race.  The obvious suspect -- the MFMA's A registers being
// re-used by the softmax arithmetic in the very next issue slot, `v_mfma a[0:15], v[0:3], v[4:7], a[0:15]; v_sub_f32
// v0, ...` -- is NOT it: the same pattern is in the forward kernels and the GEMM (13 + 72 places), whose parity tests
// are bit-stable.  Pinned accumulators WITHOUT wait states made D = 64 right and D = 32 wrong (two dependent MFMAs back
// to back, nothing the compiler pads inside asm); pinned accumulators with the wait states below are right for every
// shape of the tests.  The cause of the original failure was not isolated further.
// Wait states (hipcc pads nothing inside an asm statement; cdna_hip_programming.md section 5.7 item 2, ISA hazard table):
//   * VALU write of an A / B operand -> the MFMA reading it: 2 states           -> `s_nop 1` OPENS every statement;
//   * MFMA result -> the next MFMA taking it whole as C (accumulate chain): 0    -> nothing between the links of a chain,
//     and two products on the SAME accumulator are never adjacent in the source anyway;
//   * MFMA result (v_mfma_f32_32x32x16_bf16: 8 passes) -> any other reader or writer of it, compiler code included:
//     12 states.  Every statement ENDS with `s_nop 7` (8 states); where VALU code reads the scores / dP next, the phase
//     functions add an explicit `s_nop 7` statement behind the last product (16 states), and the accumulator reads
//     behind the tile loop sit behind `s_nop 15; s_nop 15` -- each site says so in its comment.
// tests/test_gpu_training.py::test_mhsa_backward_mfma_route_vs_valu_route holds this route against the VALU kernels
// (mhsa_bwd_dq_kernel / mhsa_bwd_dkv_kernel: plain HIP, every hazard the compiler's) on a grid of shapes;
// anemoi_build_info() records the hipcc the library was built with.
#define ANEMOI_BWD_MFMA_ACC(ACC, A, B) \
  asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\ts_nop 7" : "+a"(ACC) : "v"(A), "v"(B))
// the score-side products (S, dP) the same way, accumulators in VGPRs (the softmax arithmetic reads them in place)
#define ANEMOI_BWD_MFMA_S0(ACC, A, B) \
  asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, 0\n\ts_nop 7" : "=&v"(ACC) : "v"(A), "v"(B))
#define ANEMOI_BWD_MFMA_S(ACC, A, B) \
  asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\ts_nop 7" : "+v"(ACC) : "v"(A), "v"(B))
#ifdef ATT_BWD_PROF
__device__ unsigned long long att_prof[16];
#define ATT_T(i)                                        \
  do {                                                  \
    __builtin_amdgcn_sched_barrier(0);                  \
    const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
    tacc[i] += t_ - tlast;                              \
    tlast = t_;                                         \
    __builtin_amdgcn_sched_barrier(0);                  \
  } while (0)
#else
#define ATT_T(i)
#endif
constexpr int ATT_BWD_STAGES = 3;  // LDS ring of the backward kernels: tiles requested ATT_BWD_STAGES - 1 ahead
constexpr int ATT_BWD_NW = 4;  // waves per workgroup of the two backward kernels: 32 keys (dK/dV) or 32 queries (dQ) each
// Round 3 (per layer at S = 40 962, D = 64: 40.9 -> 27.9 ms; dK/dV 25.5 -> 17.6, dQ 14.2 -> 10.9), in the order measured
// with the s_memtime phase marks (ATT_T, tools/micro/mhsa_bwd_phase.py) and the PMC passes of tools/micro/mhsa_bwd_pmc.sh:
// the waves sat in s_waitcnt for 65 % of their cycles, and not for HBM -- eight exposed LDS round trips per tile (every
// A fragment read next to its MFMA), sixteen more for lse / delta read one dword at a time, 330 .. 500 clocks of 64-bit
// address arithmetic per tile for the staging, and a 16-deep chain of taken branches around the window mask.  Now: LDS-DMA
// ring (one barrier per tile), all fragments of a phase requested ahead of it, lse / delta as four 16-byte reads,
// 32 x 32 -> 64-bit row addresses, the window code behind a template switch.  What remains (phase marks: 2700 clocks per
// tile and wave, of which 1020 are the two resident waves' MFMAs and ~900 their softmax arithmetic): the MFMA and the
// VALU phases of the two waves of a SIMD hardly overlap; the forward's answer (one wave per SIMD, both streams
// interleaved by hand) has not been carried over.  Workgroups of eight waves (half the L2 -> LDS traffic) are slower
// (18.3 / 13.2 ms): their wave pairs run in lockstep; a cap of 168 registers (three waves per SIMD) spills 228 bytes:
// 65 ms.  Making that lockstep a ping-pong (waves w and w + 4 share a SIMD -- tools/micro/wave_simd_map.hip -- so the
// waves 4 .. 7 ran one barrier behind the waves 0 .. 3: [barrier] M2(t - 1) M1(t)
// [barrier] V(t), four stages) was built, bit-identical, and no faster either (18.0 ms): every phase is long on its own
// (M1 ~800 clocks for 256 clocks of matrix pipe) because a wave reads ALL of the tile's fragments, 24 ds_read_b128 per
// tile and wave for its 32 keys -- 54 % of the CU's LDS bandwidth at the present rate.  Two more forms, both correct at
// once and both dropped: one wave per SIMD with the next tile's S / dP MFMAs placed between two-element chunks of the
// softmax arithmetic by sched_barrier (23.1 ms; with every LDS request of the next tile at the top of the step and two
// register sets: 31.6 ms -- 362 registers, and 32 ds_read_b128 per step in front of one wave's first MFMA), and lse /
// delta read straight from global memory instead of through LDS (a third of the LDS reads gone: 17.4 ms, no change).
// What is left to try is the forward's shape: 64 keys per wave, one wave per SIMD, both streams written out by hand.
// WIN: sliding-window mask code compiled in (window >= 0).
template <int ATT_D, bool DROP, bool WIN>
__global__ __launch_bounds__(64 * ATT_BWD_NW) void mhsa_bwd_dkv_mfma_kernel(
    const bf16_t* __restrict__ qkv, int64_t ld, const bf16_t* __restrict__ dout, int64_t lddo,
    const bf16_t* __restrict__ qT, const bf16_t* __restrict__ doT, const float* __restrict__ lse2p,
    const float* __restrict__ deltap, bf16_t* __restrict__ dqkv, int64_t lddq, int S, int S_pad, int H, int C, int window,
    float scale, float scale_log2e, const AttnDropout dr_arg) {
  const AttnDropout dr = dropout_resolve(dr_arg);
  constexpr int NKS = ATT_D / 16, NDT = ATT_D / 32, RB = ATT_D * 2, CPR = RB / 16;  // chunks of 16 bytes per row
  // Query tiles (32 queries: Q and dO rows, Q^T and dO^T rows of 64 bytes, lse / delta) travel through a ring of three LDS
  // buffers by LDS-DMA: tile qt + 2 is requested while tile qt is computed, one barrier per tile (round 3; the first
  // version staged every tile with plain loads between two barriers and hid the round trip by occupancy alone).  A tile
  // array is PP pieces of 1 KiB (64 lanes x 16 bytes, landing linearly): the swizzle is applied on the SOURCE side (the
  // lane of LDS position s of row r fetches chunk swz(r, s); both swizzles are involutions).
  constexpr int NW = ATT_BWD_NW, PP = ATT_D / 16, ARR = PP * 1024, STAGE = 4 * ARR + 256, N_STAGE = 3;
  constexpr int PRE = N_STAGE - 1;  // tiles requested ahead of the one being computed
  constexpr int PPW = 4 * PP / NW, NDMA = PPW + 1;  // pieces per wave and tile (+ the lse / delta dwords)
  static_assert(4 * PP % NW == 0, "every wave stages the same number of pieces");
  __shared__ __attribute__((aligned(16))) char smem[N_STAGE * STAGE];
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  __builtin_assume(wid >= 0 && wid < NW);
  const int half = lane >> 5, ql = lane & 31;
  const int h = blockIdx.y, b = blockIdx.z;
  const int key0 = (blockIdx.x * NW + wid) * 32, key = key0 + ql, kc = key < S ? key : S - 1;
  // stationary B fragments: lane = key column, 8 consecutive d per k-step
  abf16x8_t kf[NKS], vf[NKS];
  {
    const bf16_t* kp = qkv + ((int64_t)b * S + kc) * ld + C + h * ATT_D + half * 8;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      kf[ks] = *reinterpret_cast<const abf16x8_t*>(kp + ks * 16);
      vf[ks] = *reinterpret_cast<const abf16x8_t*>(kp + C + ks * 16);
    }
  }
  af32x16_t dk[NDT], dv[NDT];
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) dk[dt][r] = dv[dt][r] = 0.f;
  // query tiles that can see any key of this workgroup
  int qt_begin = 0, qt_end = (S + 31) / 32;
  if (WIN && window >= 0) {
    const int lo = (int)blockIdx.x * (32 * NW) - window, hi = (int)blockIdx.x * (32 * NW) + 32 * NW - 1 + window;
    qt_begin = lo > 0 ? lo / 32 : 0;
    const int e = hi / 32 + 1;
    qt_end = e < qt_end ? e : qt_end;
  }
  const int prow = att_row_perm(ql);
  // DROP: lane = key, so the key-pair part of the hash and the half of its 32 bits that belongs to this key are lane
  // constants; the row part moves with the query (rows below 2^32: checked by the launcher)
  const uint32_t dkey = DROP ? ((uint32_t)(key >> 1) * 0xC2B2AE3Du ^ dr.seed) : 0u;
  const int dsh = 16 * (key & 1);
  const uint32_t drow0 = DROP ? (uint32_t)dropout_row(dr, b, h, S, 0) : 0u;
  const int64_t bh = (int64_t)b * H + h;
  auto stage = [&](int qt, int buf) {
    char* sb = smem + buf * STAGE;
#pragma unroll
    for (int i = 0; i < PPW; ++i) {  // piece id = NW i + wave: array id / PP (Q, dO, Q^T, dO^T), piece id % PP of it
      const int id = i * NW + wid, arr = id / PP, sub = id % PP;
      const char* src;
      if (arr < 2) {  // row-major tile: 1024 / RB rows per piece
        const int r = sub * (1024 / RB) + lane / CPR;
        int q = qt * 32 + r;
        q = q < S ? q : S - 1;
        const uint32_t row = (uint32_t)(b * S + q);  // (rows below 2^31: checked by the launcher)
        const int col = h * ATT_D + att_rswz<ATT_D>(r, lane % CPR) * 8;
        src = reinterpret_cast<const char*>(arr == 0 ? qkv + (uint64_t)row * (uint32_t)ld + col
                                                     : dout + (uint64_t)row * (uint32_t)lddo + col);
      } else {  // transposed tile: 16 rows d of 64 bytes per piece
        const int d = sub * 16 + (lane >> 2);
        const int64_t off = (bh * ATT_D + d) * S_pad + qt * 32 + aswz64(d, lane & 3) * 8;
        src = reinterpret_cast<const char*>((arr == 2 ? qT : doT) + off);
      }
      aglds16(src, sb + id * 1024);
    }
    // lse (log2 units) of the tile's 32 queries in the lanes 0 .. 31, delta in 32 .. 63 (every wave: same bytes, same place)
    aglds4((lane < 32 ? lse2p : deltap) + bh * S_pad + qt * 32 + (lane & 31), sb + 4 * ARR);
  };
  // ---- the three phases of a tile
  af32x16_t s_acc, dp_acc;
  abf16x8_t fdo[NDT][2], fqt[NDT][2], pb[2], dsb[2];
  float lsev[16], dlv[16];
  // M1: S = Q K^T and dP = dO V^T (lane = key, register r <-> query 8 half + (r & 7) + 16 (r >> 3) of the tile).  Every A
  // fragment of a phase is requested before the phase's first MFMA (one exposed LDS round trip per phase: with the loads
  // next to their MFMAs the loop paid eight of them per tile, 65 % of the wave cycles in s_waitcnt); the fragments of M2
  // and lse / delta (four 16-byte reads each: one dword at a time they were sixteen more round trips) are requested
  // behind the score MFMAs and land under the softmax arithmetic.  Everything the tile needs from LDS is read here.
  auto m1 = [&](const char* sb) {
    const char *q_s = sb, *do_s = sb + ARR, *qt_s = sb + 2 * ARR, *dot_s = sb + 3 * ARR;
    const float* lse_s = reinterpret_cast<const float*>(sb + 4 * ARR);
    const float* dl_s = lse_s + 32;
    abf16x8_t fa[NKS], fb[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      const int ch = att_rswz<ATT_D>(prow, ks * 2 + half);
      fa[ks] = *reinterpret_cast<const abf16x8_t*>(q_s + prow * RB + (ch << 4));
      fb[ks] = *reinterpret_cast<const abf16x8_t*>(do_s + prow * RB + (ch << 4));
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      if (ks == 0) {
        ANEMOI_BWD_MFMA_S0(s_acc, fa[ks], kf[ks]);
        ANEMOI_BWD_MFMA_S0(dp_acc, fb[ks], vf[ks]);
      } else {
        ANEMOI_BWD_MFMA_S(s_acc, fa[ks], kf[ks]);
        ANEMOI_BWD_MFMA_S(dp_acc, fb[ks], vf[ks]);
      }
    }
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) {
      const int drow = dt * 32 + ql;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int off = drow * 64 + (aswz64(drow, kk * 2 + half) << 4);
        fdo[dt][kk] = *reinterpret_cast<const abf16x8_t*>(dot_s + off);
        fqt[dt][kk] = *reinterpret_cast<const abf16x8_t*>(qt_s + off);
      }
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int q0 = 8 * half + 4 * (g & 1) + 16 * (g >> 1);
      const float4 l4 = *reinterpret_cast<const float4*>(lse_s + q0);
      const float4 d4 = *reinterpret_cast<const float4*>(dl_s + q0);
      lsev[4 * g] = l4.x, lsev[4 * g + 1] = l4.y, lsev[4 * g + 2] = l4.z, lsev[4 * g + 3] = l4.w;
      dlv[4 * g] = d4.x, dlv[4 * g + 1] = d4.y, dlv[4 * g + 2] = d4.z, dlv[4 * g + 3] = d4.w;
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 7" : "+v"(s_acc), "+v"(dp_acc));  // (with the last product's own eight: its results -- operands of the wait, so no read of them can be placed in front of it -- are read next)
  };
  // V: P = exp2(S c - lse), dS = P (dP - delta), both as bf16 B fragments of M2
  auto vphase = [&](int qt) {
    float p[16], ds[16];
    const uint32_t drow_tile = (drow0 + (uint32_t)(qt * 32)) * 0x9E3779B1u;
    (void)drow_tile;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int qi = 8 * half + (r & 7) + 16 * (r >> 3);
      float pe = __builtin_amdgcn_exp2f(fmaf(s_acc[r], scale_log2e, -lsev[r]));
      if (WIN && window >= 0) {
        const int dq_ = qt * 32 + qi - key;
        pe = (dq_ <= window && -dq_ <= window) ? pe : 0.f;
      }
      if constexpr (DROP) {  // dV takes the dropped probabilities, dS = P (keep / (1 - p) dP - delta)
        // (row part = (row0 + q) G with q = 32 qt + qi: the tile's base once, the register's share a constant -- no
        //  per-element multiply in front of the mixer's own)
        const uint32_t x = dropout_mix((drow_tile + (uint32_t)qi * 0x9E3779B1u) ^ dkey);
        const float kf_ = ((x >> dsh) & 0x7fffu) >= dr.thr15 ? dr.keep_scale : 0.f;
        p[r] = pe * kf_;
        ds[r] = pe * (dp_acc[r] * kf_ - dlv[r]);
      } else {
        p[r] = pe;
        ds[r] = pe * (dp_acc[r] - dlv[r]);
      }
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      uint32_t w1[4], w2[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if constexpr (DROP) {
          w1[i] = pack_bf16x2(p[kk * 8 + 2 * i], p[kk * 8 + 2 * i + 1]);
          w2[i] = pack_bf16x2(ds[kk * 8 + 2 * i], ds[kk * 8 + 2 * i + 1]);
        } else {
          asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w1[i]) : "v"(p[kk * 8 + 2 * i]), "v"(p[kk * 8 + 2 * i + 1]));
          asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w2[i]) : "v"(ds[kk * 8 + 2 * i]), "v"(ds[kk * 8 + 2 * i + 1]));
        }
      }
      pb[kk] = *reinterpret_cast<abf16x8_t*>(w1);
      dsb[kk] = *reinterpret_cast<abf16x8_t*>(w2);
    }
  };
  // M2: dV^T += dO^T P,  dK^T += Q^T dS   (A rows = d, reduction over the tile's queries)
  auto m2 = [&]() {
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        ANEMOI_BWD_MFMA_ACC(dv[dt], fdo[dt][kk], pb[kk]);
        ANEMOI_BWD_MFMA_ACC(dk[dt], fqt[dt][kk], dsb[kk]);
      }
    }
  };
#pragma unroll
  for (int i = 0; i < PRE; ++i)
    if (qt_begin + i < qt_end) stage(qt_begin + i, i);
#ifdef ATT_BWD_PROF
  unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime();
#endif
  // One tile.  The ring position is a COMPILE-TIME constant (the loop below is unrolled by the ring's three buffers): with a
  // run-time buffer index every one of the tile's 24 fragment reads paid a v_add for its LDS address -- 40 of the loop's ~130
  // VALU instructions per tile in a kernel that is bound by its VALU stream (round 5) -- now they are immediate offsets.
  auto tile = [&](int qt, auto buf_c) {
    constexpr int BUF = decltype(buf_c)::value;
    ATT_T(0);
    // this wave's requests of tile qt have landed (the PRE - 1 tiles behind it may stay in flight: vmcnt counts in order,
    // NDMA per tile; the last PRE - 1 tiles simply drain); the barrier publishes the buffer and retires tile qt - 1,
    // whose buffer is refilled next (see mhsa_bf16_kernel)
    if (qt + PRE - 1 < qt_end) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PRE - 1) * NDMA) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ATT_T(1);
    __syncthreads();
    ATT_T(2);
    if (qt + PRE < qt_end) stage(qt + PRE, (BUF + PRE) % N_STAGE);
    ATT_T(3);
    m1(smem + BUF * STAGE);
    ATT_T(4);
    vphase(qt);
    ATT_T(5);
    m2();
    ATT_T(6);
  };
  {
    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;
    using B2 = std::integral_constant<int, 2>;
    static_assert(N_STAGE == 3, "the tile loop is unrolled by the ring");
    int qt = qt_begin;
    for (; qt + 2 < qt_end; qt += 3) {
      tile(qt, B0{});
      tile(qt + 1, B1{});
      tile(qt + 2, B2{});
    }
    if (qt < qt_end) tile(qt, B0{});
    if (qt + 1 < qt_end) tile(qt + 1, B1{});
  }
#ifdef ATT_BWD_PROF
  if (blockIdx.x == 3 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 64)
    for (int i = 0; i < 8; ++i) att_prof[i] = tacc[i];
#endif
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // last asm MFMA -> the accumulator reads below ...
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt) asm volatile("" : "+a"(dk[dt]), "+a"(dv[dt]));  // ... which depend on THIS statement (volatile asm statements keep their order; plain register reads are not ordered against one)
  if (key < S) {  // lane = key; register r <-> d = dt * 32 + 8 (r >> 2) + 4 half + (r & 3)
    bf16_t* kp = dqkv + ((int64_t)b * S + key) * lddq + C + h * ATT_D;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float k4[4] = {dk[dt][4 * g] * scale, dk[dt][4 * g + 1] * scale, dk[dt][4 * g + 2] * scale,
                             dk[dt][4 * g + 3] * scale};
        const float v4[4] = {dv[dt][4 * g], dv[dt][4 * g + 1], dv[dt][4 * g + 2], dv[dt][4 * g + 3]};
        VecIO<bf16_t, 4>::store(kp + dt * 32 + 8 * g + 4 * half, k4);
        VecIO<bf16_t, 4>::store(kp + C + dt * 32 + 8 * g + 4 * half, v4);
      }
  }
}

template <int ATT_D, bool DROP = false, bool WIN = true>
__global__ __launch_bounds__(64 * ATT_BWD_NW) void mhsa_bwd_dq_mfma_kernel(
    const bf16_t* __restrict__ qkv, int64_t ld, const bf16_t* __restrict__ dout, int64_t lddo,
    const bf16_t* __restrict__ kT, const float* __restrict__ lse, const float* __restrict__ delta,
    bf16_t* __restrict__ dqkv, int64_t lddq, int S, int S_pad, int H, int C, int window, float scale, float scale_log2e,
    const AttnDropout dr_arg) {
  const AttnDropout dr = dropout_resolve(dr_arg);
  constexpr int NKS = ATT_D / 16, NDT = ATT_D / 32, RB = ATT_D * 2, CPR = RB / 16;
  // key tiles (K and V rows, K^T rows of 64 bytes) through a ring of three LDS buffers by LDS-DMA, as in the dK/dV kernel:
  // 3 PP pieces of 1 KiB per tile, piece i NW + wave for the waves that have one (D = 64: twelve pieces on eight waves --
  // two for the waves 0 .. 3, one for the others; every wave waits for its own count)
  constexpr int NW = ATT_BWD_NW, PP = ATT_D / 16, ARR = PP * 1024, NPIECE = 3 * PP, PPW = (NPIECE + NW - 1) / NW;
  constexpr int STAGE = 3 * ARR, N_STAGE = ATT_BWD_STAGES, PRE = N_STAGE - 1;
  static_assert(PPW <= 3, "the wait below knows the counts 0 .. 3");
  __shared__ __attribute__((aligned(16))) char smem[N_STAGE * STAGE];
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  __builtin_assume(wid >= 0 && wid < NW);
  const int half = lane >> 5, ql = lane & 31;
  const int h = blockIdx.y, b = blockIdx.z;
  const int q = (blockIdx.x * NW + wid) * 32 + ql, qc = q < S ? q : S - 1;
  const int n_mine = wid < NPIECE ? (NPIECE - wid + NW - 1) / NW : 0;  // this wave's requests per tile
  abf16x8_t qf[NKS], dof[NKS];  // B fragments: lane = query column, 8 consecutive d per k-step
  {
    const bf16_t* qp = qkv + ((int64_t)b * S + qc) * ld + h * ATT_D + half * 8;
    const bf16_t* dp = dout + ((int64_t)b * S + qc) * lddo + h * ATT_D + half * 8;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      qf[ks] = *reinterpret_cast<const abf16x8_t*>(qp + ks * 16);
      dof[ks] = *reinterpret_cast<const abf16x8_t*>(dp + ks * 16);
    }
  }
  const float lse2 = q < S ? lse[((int64_t)b * H + h) * S + q] * 1.44269504088896340736f : INFINITY;
  const float dl = q < S ? delta[((int64_t)b * H + h) * S + q] : 0.f;
  const uint32_t drow = DROP ? dropout_row_part(dr, dropout_row(dr, b, h, S, q)) : 0u;  // lane = query
  af32x16_t dq[NDT];
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[dt][r] = 0.f;
  int kt_begin = 0, kt_end = (S + 31) / 32;
  if (WIN && window >= 0) {
    const int lo = (int)blockIdx.x * (32 * NW) - window, hi = (int)blockIdx.x * (32 * NW) + 32 * NW - 1 + window;
    kt_begin = lo > 0 ? lo / 32 : 0;
    const int e = hi / 32 + 1;
    kt_end = e < kt_end ? e : kt_end;
  }
  const int prow = att_row_perm(ql);
  const int64_t bh = (int64_t)b * H + h;
  auto stage = [&](int kt, int buf) {
    char* sb = smem + buf * STAGE;
#pragma unroll
    for (int i = 0; i < PPW; ++i) {  // array id / PP (K, V, K^T), piece id % PP of it
      const int id = i * NW + wid, arr = id / PP, sub = id % PP;
      if (id >= NPIECE) break;  // (wave-uniform)
      const char* src;
      if (arr < 2) {
        const int r = sub * (1024 / RB) + lane / CPR;
        int key = kt * 32 + r;
        key = key < S ? key : S - 1;
        src = reinterpret_cast<const char*>(qkv + (uint64_t)(uint32_t)(b * S + key) * (uint32_t)ld + (arr + 1) * C +
                                            h * ATT_D + att_rswz<ATT_D>(r, lane % CPR) * 8);
      } else {
        const int d = sub * 16 + (lane >> 2);
        src = reinterpret_cast<const char*>(kT + (bh * ATT_D + d) * S_pad + kt * 32 + aswz64(d, lane & 3) * 8);
      }
      aglds16(src, sb + id * 1024);
    }
  };
#pragma unroll
  for (int i = 0; i < PRE; ++i)
    if (kt_begin + i < kt_end) stage(kt_begin + i, i);
  // One tile; the ring position is a compile-time constant (the loop is unrolled by the ring's buffers, see the dK/dV kernel:
  // immediate LDS offsets instead of a v_add per fragment read)
  auto tile = [&](int kt, auto buf_c) {
    constexpr int BUF = decltype(buf_c)::value;
    (void)&dq;  // (an asm operand alone does not capture in a generic lambda)
    // this wave's requests of tile kt have landed; those of the PRE - 1 tiles behind it (n_mine each) may stay in flight
    if (kt + PRE - 1 < kt_end && n_mine == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * (PRE - 1)) : "memory");
    else if (kt + PRE - 1 < kt_end && n_mine == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (PRE - 1)) : "memory");
    else if (kt + PRE - 1 < kt_end && n_mine == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PRE - 1) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + PRE < kt_end) stage(kt + PRE, (BUF + PRE) % N_STAGE);
    const char* sb = smem + BUF * STAGE;
    const char *k_s = sb, *v_s = sb + ARR, *kt_s = sb + 2 * ARR;
    // ---- S^T = K Q^T and dP^T = V dO^T: lane = query, register r <-> key 8 half + (r & 7) + 16 (r >> 3) of the tile
    af32x16_t s_acc, dp_acc;  // (fragment requests ahead of each phase: see the dK/dV kernel)
    abf16x8_t fa[NKS], fb[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      const int ch = att_rswz<ATT_D>(prow, ks * 2 + half);
      fa[ks] = *reinterpret_cast<const abf16x8_t*>(k_s + prow * RB + (ch << 4));
      fb[ks] = *reinterpret_cast<const abf16x8_t*>(v_s + prow * RB + (ch << 4));
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      if (ks == 0) {
        ANEMOI_BWD_MFMA_S0(s_acc, fa[ks], qf[ks]);
        ANEMOI_BWD_MFMA_S0(dp_acc, fb[ks], dof[ks]);
      } else {
        ANEMOI_BWD_MFMA_S(s_acc, fa[ks], qf[ks]);
        ANEMOI_BWD_MFMA_S(dp_acc, fb[ks], dof[ks]);
      }
    }
    abf16x8_t fkt[NDT][2];
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) {
      const int drow = dt * 32 + ql;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
        fkt[dt][kk] = *reinterpret_cast<const abf16x8_t*>(kt_s + drow * 64 + (aswz64(drow, kk * 2 + half) << 4));
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 7" : "+v"(s_acc), "+v"(dp_acc));  // (with the last product's own eight: its results -- operands of the wait, so no read of them can be placed in front of it -- are read next)
    float ds[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = kt * 32 + 8 * half + (r & 7) + 16 * (r >> 3);
      float pe = __builtin_amdgcn_exp2f(fmaf(s_acc[r], scale_log2e, -lse2));
      // keys behind S need no mask: their K / V rows are row S - 1 again (finite dS) and their K^T columns are zero
      if (WIN && window >= 0) pe = ((key - q <= window) && (q - key <= window)) ? pe : 0.f;
      if constexpr (DROP) {  // registers r, r + 1 (r even) are the two keys of one pair: one hash for both
        const uint32_t x = dropout_mix(drow ^ (uint32_t)(key >> 1) * 0xC2B2AE3Du);
        const float kf_ = ((x >> (16 * (r & 1))) & 0x7fffu) >= dr.thr15 ? dr.keep_scale : 0.f;
        ds[r] = pe * (dp_acc[r] * kf_ - dl);
      } else {
        ds[r] = pe * (dp_acc[r] - dl);
      }
    }
    abf16x8_t dsb[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      uint32_t w2[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if constexpr (DROP) w2[i] = pack_bf16x2(ds[kk * 8 + 2 * i], ds[kk * 8 + 2 * i + 1]);
        else asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w2[i]) : "v"(ds[kk * 8 + 2 * i]), "v"(ds[kk * 8 + 2 * i + 1]));
      }
      dsb[kk] = *reinterpret_cast<abf16x8_t*>(w2);
    }
    // ---- dQ^T += K^T dS^T   (ANEMOI_BWD_MFMA_ACC: see the note above the dK/dV kernel)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) ANEMOI_BWD_MFMA_ACC(dq[dt], fkt[dt][kk], dsb[kk]);
    }
  };
  {
    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;
    using B2 = std::integral_constant<int, 2>;
    static_assert(N_STAGE == 3, "the tile loop is unrolled by the ring");
    int kt = kt_begin;
    for (; kt + 2 < kt_end; kt += 3) {
      tile(kt, B0{});
      tile(kt + 1, B1{});
      tile(kt + 2, B2{});
    }
    if (kt < kt_end) tile(kt, B0{});
    if (kt + 1 < kt_end) tile(kt + 1, B1{});
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // last asm MFMA -> the accumulator reads below ...
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt) asm volatile("" : "+a"(dq[dt]));  // ... which depend on THIS statement (see the dK/dV kernel)
  if (q < S) {
    bf16_t* qp = dqkv + ((int64_t)b * S + q) * lddq + h * ATT_D;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float q4[4] = {dq[dt][4 * g] * scale, dq[dt][4 * g + 1] * scale, dq[dt][4 * g + 2] * scale,
                             dq[dt][4 * g + 3] * scale};
        VecIO<bf16_t, 4>::store(qp + dt * 32 + 8 * g + 4 * half, q4);
      }
  }
}

// ---------------------------------------------------------------------------------------------
// Backward of softmax(Q K^T / sqrt(D)) V (flash-attention style: nothing of size S x S is stored; the probabilities are
// recomputed from the saved log-sum-exp).  With P_ij = exp(s_ij - lse_i), dP_ij = dO_i . v_j, delta_i = dO_i . O_i,
// dS_ij = P_ij (dP_ij - delta_i):   dQ_i = scale sum_j dS_ij k_j,   dK_j = scale sum_i dS_ij q_i,   dV_j = sum_i P_ij dO_i.
// VALU kernels (any head size <= 128, f32 or bf16 storage, f32 arithmetic): one wave per (query, head) for dQ / delta,
// one wave per (key, head) for dK / dV, the other index strided over the lanes, per-lane partial vectors merged by
// wave reductions -- no atomics, reproducible.  O(S^2 D) on the vector pipe: a correct backward for training the
// Transformer processor at test / moderate sizes; the MFMA forward's tiling has not been carried over to it yet.
// ---------------------------------------------------------------------------------------------
template <typename T, int DMAX>
__global__ __launch_bounds__(256) void mhsa_bwd_dq_kernel(const T* __restrict__ qkv, int64_t ld, const T* __restrict__ o,
                                                          int64_t ldo, const T* __restrict__ dout, int64_t lddo,
                                                          const float* __restrict__ lse, float* __restrict__ delta,
                                                          T* __restrict__ dqkv, int64_t lddq, int S, int H, int D, int C,
                                                          int window, float scale, int64_t total, const AttnDropout dr_arg) {
  const AttnDropout dr = dropout_resolve(dr_arg);
  const int lane = threadIdx.x & 63;
  const int64_t unit = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);  // (b, q, h)
  if (unit >= total) return;
  const int h = (int)(unit % H);
  const int64_t bq = unit / H;
  const int q = (int)(bq % S);
  const int64_t b = bq / S;
  float qv[DMAX], dov[DMAX], acc[DMAX];
  float dl = 0.f;
#pragma unroll
  for (int d = 0; d < DMAX; ++d) {
    qv[d] = d < D ? Elem<T>::load(qkv + bq * ld + h * D + d) * scale : 0.f;
    dov[d] = d < D ? Elem<T>::load(dout + bq * lddo + h * D + d) : 0.f;
    if (d < D) dl = fmaf(dov[d], Elem<T>::load(o + bq * ldo + h * D + d), dl);
    acc[d] = 0.f;
  }
  const float ls = lse[(b * H + h) * S + q];
  if (lane == 0) delta[(b * H + h) * S + q] = dl;
  int k_lo = 0, k_hi = S;
  if (window >= 0) {
    k_lo = q - window > 0 ? q - window : 0;
    k_hi = q + window + 1 < S ? q + window + 1 : S;
  }
  for (int key = k_lo + lane; key < k_hi; key += 64) {
    const T* kp = qkv + (b * S + key) * ld + C + h * D;
    const T* vp = kp + C;
    float s = 0.f, dp = 0.f;
#pragma unroll
    for (int d = 0; d < DMAX; ++d)
      if (d < D) {
        s = fmaf(qv[d], Elem<T>::load(kp + d), s);
        dp = fmaf(dov[d], Elem<T>::load(vp + d), dp);
      }
    // with dropout: O = (keep / (1 - p) * P) V, so dP = keep / (1 - p) * (dO . v) and sum_j P dP = dO . O = dl still
    const float ds = __expf(s - ls) * (dp * dropout_keep(dr, dropout_row(dr, b, h, S, q), key) - dl) * scale;
#pragma unroll
    for (int d = 0; d < DMAX; ++d)
      if (d < D) acc[d] = fmaf(ds, Elem<T>::load(kp + d), acc[d]);
  }
#pragma unroll
  for (int d = 0; d < DMAX; ++d) {
    if (d < D) {
      const float v = wave_sum(acc[d]);
      if (lane == 0) Elem<T>::store(dqkv + bq * lddq + h * D + d, v);
    }
  }
}

template <typename T, int DMAX>
__global__ __launch_bounds__(256) void mhsa_bwd_dkv_kernel(const T* __restrict__ qkv, int64_t ld,
                                                           const T* __restrict__ dout, int64_t lddo,
                                                           const float* __restrict__ lse, const float* __restrict__ delta,
                                                           T* __restrict__ dqkv, int64_t lddq, int S, int H, int D, int C,
                                                           int window, float scale, int64_t total, const AttnDropout dr_arg) {
  const AttnDropout dr = dropout_resolve(dr_arg);
  const int lane = threadIdx.x & 63;
  const int64_t unit = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);  // (b, key, h)
  if (unit >= total) return;
  const int h = (int)(unit % H);
  const int64_t bk = unit / H;
  const int key = (int)(bk % S);
  const int64_t b = bk / S;
  float kv[DMAX], vv[DMAX], dk[DMAX], dv[DMAX];
#pragma unroll
  for (int d = 0; d < DMAX; ++d) {
    kv[d] = d < D ? Elem<T>::load(qkv + bk * ld + C + h * D + d) : 0.f;
    vv[d] = d < D ? Elem<T>::load(qkv + bk * ld + 2 * C + h * D + d) : 0.f;
    dk[d] = 0.f;
    dv[d] = 0.f;
  }
  int q_lo = 0, q_hi = S;
  if (window >= 0) {
    q_lo = key - window > 0 ? key - window : 0;
    q_hi = key + window + 1 < S ? key + window + 1 : S;
  }
  for (int q = q_lo + lane; q < q_hi; q += 64) {
    const T* qp = qkv + (b * S + q) * ld + h * D;
    const T* dop = dout + (b * S + q) * lddo + h * D;
    float s = 0.f, dp = 0.f;
#pragma unroll
    for (int d = 0; d < DMAX; ++d)
      if (d < D) {
        s = fmaf(Elem<T>::load(qp + d), kv[d], s);
        dp = fmaf(Elem<T>::load(dop + d), vv[d], dp);
      }
    const int64_t si = (b * H + h) * S + q;
    const float p = __expf(s * scale - lse[si]);
    const float keep = dropout_keep(dr, dropout_row(dr, b, h, S, q), key);
    const float ds = p * (dp * keep - delta[si]) * scale;
    const float pd = p * keep;
#pragma unroll
    for (int d = 0; d < DMAX; ++d)
      if (d < D) {
        dv[d] = fmaf(pd, Elem<T>::load(dop + d), dv[d]);
        dk[d] = fmaf(ds, Elem<T>::load(qp + d), dk[d]);
      }
  }
#pragma unroll
  for (int d = 0; d < DMAX; ++d) {
    if (d < D) {
      const float a = wave_sum(dk[d]), c = wave_sum(dv[d]);
      if (lane == 0) {
        Elem<T>::store(dqkv + bk * lddq + C + h * D + d, a);
        Elem<T>::store(dqkv + bk * lddq + 2 * C + h * D + d, c);
      }
    }
  }
}

}  // namespace anemoi

using namespace anemoi;

extern "C" {

static inline int64_t mhsa_vt_bytes(int B, int S, int H, int D) {
  return ((int64_t)B * H * D * ((S + 63) / 64 * 64) * 2 + 255) / 256 * 256;
}

static inline int64_t mhsa_tail_bytes(int B, int S, int H, int D) {
  return ((int64_t)B * mhsa_tail_rows(S) * H * mhsa_tail_splits(B, S, H) * (D + 2) * 4 + 255) / 256 * 256;
}

int64_t anemoi_mhsa_workspace_bytes(int dtype, int B, int S, int H, int D) {
  if (dtype == ANEMOI_BF16 && (D == 64 || D == 32))  // V^T, then the partial states of the left-over rows' key chunks
    return mhsa_vt_bytes(B, S, H, D) + mhsa_tail_bytes(B, S, H, D) + 256;  // + the fallback flag of the 4-wave kernel
  return 0;
}

int anemoi_mhsa(int dtype, const void* qkv, int64_t ld, void* out, int64_t ldo, void* workspace, float* lse, int B, int S,
                int H, int D, int window, float dropout_p, uint32_t dropout_seed, const void* dropout_seed_dev,
                int dropout_h0, int dropout_h_total, anemoi_stream_t stream) {
  ANEMOI_REQUIRE(qkv && out, ANEMOI_ERR_INVALID, "anemoi_mhsa: null pointer");
  ANEMOI_REQUIRE(B > 0 && S > 0 && H > 0 && D > 0, ANEMOI_ERR_INVALID, "anemoi_mhsa: bad shape");
  const int C = H * D;
  ANEMOI_REQUIRE(ld >= 3 * (int64_t)C && ldo >= C, ANEMOI_ERR_INVALID, "anemoi_mhsa: leading dimension too small");
  hipStream_t st = as_stream(stream);
  const float scale = 1.0f / sqrtf((float)D);
  ANEMOI_REQUIRE(dropout_p >= 0.f && dropout_p <= 1.f, ANEMOI_ERR_INVALID, "anemoi_mhsa: dropout_p %g outside [0, 1]",
                 (double)dropout_p);
  ANEMOI_REQUIRE(dropout_h0 >= 0 && (dropout_h_total == 0 || dropout_h0 + H <= dropout_h_total), ANEMOI_ERR_INVALID,
                 "anemoi_mhsa: heads %d .. %d of %d", dropout_h0, dropout_h0 + H, dropout_h_total);
  ANEMOI_REQUIRE(dropout_seed_dev == nullptr || (uintptr_t)dropout_seed_dev % 4 == 0, ANEMOI_ERR_INVALID,
                 "anemoi_mhsa: dropout_seed_dev must be 4-byte aligned");
  const AttnDropout dr = make_dropout(dropout_p, dropout_seed, dropout_h0, dropout_h_total > 0 ? dropout_h_total : H,
                                      dropout_seed_dev);
  // MFMA route, with or without dropout (the mask is applied to the packed probabilities; the kernels hash 32-bit row
  // indices: B x heads x S < 2^32, and p = 1 -- everything dropped -- stays on the generic kernel)
  const bool drop = dr.thr15 != 0;
  if (!dr.drop_all && (int64_t)B * dr.h_total * S < ((int64_t)1 << 32) && dtype == ANEMOI_BF16 && (D == 64 || D == 32) &&
      (uintptr_t)qkv % 16 == 0 && ld % 8 == 0 && (uintptr_t)out % 8 == 0 && ldo % 4 == 0) {
    ANEMOI_REQUIRE(workspace != nullptr, ANEMOI_ERR_INVALID, "anemoi_mhsa: workspace of %lld bytes required",
                   (long long)anemoi_mhsa_workspace_bytes(dtype, B, S, H, D));
    const int S_pad = (S + 63) / 64 * 64;
    hipLaunchKernelGGL(transpose_v_kernel, dim3(S_pad / 64, H, B), dim3(256), 0, st,
                       static_cast<const bf16_t*>(qkv), ld, S, S_pad, H, D, C, static_cast<bf16_t*>(workspace));
    // the few queries behind the last whole 512-query block: key-split kernels above (not a workgroup of their own)
    const int rem = mhsa_tail_rows(S);
    const int s_main = S - rem;
    if (rem > 0) {
      const int n_split = mhsa_tail_splits(B, S, H);
      const int chunk = (S + n_split - 1) / n_split;
      const int64_t n_units = (int64_t)B * rem * H, total = n_units * n_split;
      float* part = reinterpret_cast<float*>(static_cast<char*>(workspace) + mhsa_vt_bytes(B, S, H, D));
      const dim3 tgrid((unsigned)((total + 3) / 4)), tblock(256);
      if (D == 64) {
        hipLaunchKernelGGL(mhsa_tail_kernel<64>, tgrid, tblock, 0, st, static_cast<const bf16_t*>(qkv), ld, S, H, C, window,
                           scale, s_main, rem, n_split, chunk, total, part, dr);
        hipLaunchKernelGGL(mhsa_tail_merge_kernel<64>, dim3((unsigned)((n_units + 3) / 4)), dim3(256), 0, st, part,
                           static_cast<bf16_t*>(out), ldo, lse, S, H, s_main, rem, n_split, n_units);
      } else {
        hipLaunchKernelGGL(mhsa_tail_kernel<32>, tgrid, tblock, 0, st, static_cast<const bf16_t*>(qkv), ld, S, H, C, window,
                           scale, s_main, rem, n_split, chunk, total, part, dr);
        hipLaunchKernelGGL(mhsa_tail_merge_kernel<32>, dim3((unsigned)((n_units + 3) / 4)), dim3(256), 0, st, part,
                           static_cast<bf16_t*>(out), ldo, lse, S, H, s_main, rem, n_split, n_units);
      }
    }
    const dim3 grid((s_main + ATT_QBLK - 1) / ATT_QBLK, H, B), block(64 * ATT_WAVES);
#define ANEMOI_MHSA_FWD(DD, DR)                                                                                      \
  hipLaunchKernelGGL((mhsa_bf16_kernel<DD, DR>), grid, block, 0, st, static_cast<const bf16_t*>(qkv), ld,            \
                     static_cast<const bf16_t*>(workspace), static_cast<bf16_t*>(out), ldo, S, S_pad, H, C, window, \
                     scale * 1.44269504088896340736f, lse, dr)
    // D = 64, global attention: the 4-wave software-pipelined kernel; its fallback (a probability left the f32 range under
    // the fixed reference maximum) is the 8-wave kernel behind a device-side flag -- no host round trip, graph capturable
    if (D == 64 && window < 0 && (int64_t)S * ld * 2 < ((int64_t)1 << 31)) {
      int* flag = reinterpret_cast<int*>(static_cast<char*>(workspace) + mhsa_vt_bytes(B, S, H, D) + mhsa_tail_bytes(B, S, H, D));
      if (hipMemsetAsync(flag, 0, 4, st) != hipSuccess) return fail(ANEMOI_ERR_LAUNCH, "anemoi_mhsa: flag reset");
      const dim3 block4(64 * W4_WAVES);
      if (drop)
        hipLaunchKernelGGL((mhsa_bf16_w4_kernel<true>), grid, block4, 0, st, static_cast<const bf16_t*>(qkv), ld,
                           static_cast<const bf16_t*>(workspace), static_cast<bf16_t*>(out), ldo, S, S_pad, H, C,
                           scale * 1.44269504088896340736f, lse, dr, flag);
      else
        hipLaunchKernelGGL((mhsa_bf16_w4_kernel<false>), grid, block4, 0, st, static_cast<const bf16_t*>(qkv), ld,
                           static_cast<const bf16_t*>(workspace), static_cast<bf16_t*>(out), ldo, S, S_pad, H, C,
                           scale * 1.44269504088896340736f, lse, dr, flag);
      if (drop)
        hipLaunchKernelGGL((mhsa_bf16_kernel<64, true>), grid, block, 0, st, static_cast<const bf16_t*>(qkv), ld,
                           static_cast<const bf16_t*>(workspace), static_cast<bf16_t*>(out), ldo, S, S_pad, H, C, window,
                           scale * 1.44269504088896340736f, lse, dr, flag);
      else
        hipLaunchKernelGGL((mhsa_bf16_kernel<64, false>), grid, block, 0, st, static_cast<const bf16_t*>(qkv), ld,
                           static_cast<const bf16_t*>(workspace), static_cast<bf16_t*>(out), ldo, S, S_pad, H, C, window,
                           scale * 1.44269504088896340736f, lse, dr, flag);
    } else if (D == 64) {
      if (drop) ANEMOI_MHSA_FWD(64, true);
      else ANEMOI_MHSA_FWD(64, false);
    } else {
      if (drop) ANEMOI_MHSA_FWD(32, true);
      else ANEMOI_MHSA_FWD(32, false);
    }
#undef ANEMOI_MHSA_FWD
    return check_launch("anemoi_mhsa(bf16, MFMA)");
  }
  ANEMOI_REQUIRE(D <= 128, ANEMOI_ERR_UNSUPPORTED, "anemoi_mhsa: head size %d > 128", D);
  const int64_t units = (int64_t)B * S * H;
  ANEMOI_REQUIRE((units + 3) / 4 < ((int64_t)1 << 31), ANEMOI_ERR_UNSUPPORTED, "anemoi_mhsa: grid too large");
  dim3 grid((unsigned)((units + 3) / 4)), block(256);
#define GEN(T, DM)                                                                                               \
  hipLaunchKernelGGL((mhsa_generic_kernel<T, DM>), grid, block, 0, st, static_cast<const T*>(qkv), ld,           \
                     static_cast<T*>(out), ldo, S, H, D, C, window, scale, units, lse, dr, 0, S)
  if (dtype == ANEMOI_F32) {
    if (D <= 32) GEN(float, 32);
    else if (D <= 64) GEN(float, 64);
    else GEN(float, 128);
  } else if (dtype == ANEMOI_BF16) {
    if (D <= 32) GEN(bf16_t, 32);
    else if (D <= 64) GEN(bf16_t, 64);
    else GEN(bf16_t, 128);
  } else {
    return fail(ANEMOI_ERR_UNSUPPORTED, "anemoi_mhsa: dtype %d", dtype);
  }
#undef GEN
  return check_launch("anemoi_mhsa(generic)");
}

#ifdef ATT_BWD_PROF
int anemoi_debug_att_prof(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(att_prof), sizeof(att_prof));
}
#endif

int64_t anemoi_mhsa_backward_workspace_bytes(int dtype, int B, int S, int H, int D) {
  // Q^T, K^T, dO^T + the padded log-sum-exp (log2 units) and delta rows [B, H, S_pad] the dK/dV kernel's LDS-DMA reads
  if (dtype == ANEMOI_BF16 && (D == 64 || D == 32))
    return 3 * mhsa_vt_bytes(B, S, H, D) + 2 * (int64_t)B * H * ((S + 63) / 64 * 64) * 4;
  return 0;
}

int anemoi_mhsa_backward(int dtype, const void* qkv, int64_t ld, const void* out, int64_t ldo, const void* dout,
                         int64_t lddo, const float* lse, float* delta, void* dqkv, int64_t lddq, void* workspace, int B,
                         int S, int H, int D, int window, float dropout_p, uint32_t dropout_seed,
                         const void* dropout_seed_dev, int dropout_h0, int dropout_h_total, anemoi_stream_t stream) {
  ANEMOI_REQUIRE(qkv && out && dout && lse && delta && dqkv, ANEMOI_ERR_INVALID, "anemoi_mhsa_backward: null pointer");
  ANEMOI_REQUIRE(B > 0 && S > 0 && H > 0 && D > 0, ANEMOI_ERR_INVALID, "anemoi_mhsa_backward: bad shape");
  const int C = H * D;
  ANEMOI_REQUIRE(ld >= 3 * (int64_t)C && lddq >= 3 * (int64_t)C && ldo >= C && lddo >= C, ANEMOI_ERR_INVALID,
                 "anemoi_mhsa_backward: leading dimension too small");
  ANEMOI_REQUIRE(D <= 128, ANEMOI_ERR_UNSUPPORTED, "anemoi_mhsa_backward: head size %d > 128", D);
  ANEMOI_REQUIRE(dropout_p >= 0.f && dropout_p <= 1.f, ANEMOI_ERR_INVALID, "anemoi_mhsa_backward: dropout_p %g outside [0, 1]",
                 (double)dropout_p);
  ANEMOI_REQUIRE(dropout_h0 >= 0 && (dropout_h_total == 0 || dropout_h0 + H <= dropout_h_total), ANEMOI_ERR_INVALID,
                 "anemoi_mhsa_backward: heads %d .. %d of %d", dropout_h0, dropout_h0 + H, dropout_h_total);
  ANEMOI_REQUIRE(dropout_seed_dev == nullptr || (uintptr_t)dropout_seed_dev % 4 == 0, ANEMOI_ERR_INVALID,
                 "anemoi_mhsa: dropout_seed_dev must be 4-byte aligned");
  const AttnDropout dr = make_dropout(dropout_p, dropout_seed, dropout_h0, dropout_h_total > 0 ? dropout_h_total : H,
                                      dropout_seed_dev);
  const bool drop = dr.thr15 != 0;
  hipStream_t st = as_stream(stream);
  const float scale = 1.0f / sqrtf((float)D);
  const int64_t units = (int64_t)B * S * H;
  ANEMOI_REQUIRE((units + 3) / 4 < ((int64_t)1 << 31), ANEMOI_ERR_UNSUPPORTED, "anemoi_mhsa_backward: grid too large");
  if (!dr.drop_all && (int64_t)B * dr.h_total * S < ((int64_t)1 << 32) && dtype == ANEMOI_BF16 && (D == 64 || D == 32) &&
      workspace != nullptr && (int64_t)B * S < ((int64_t)1 << 31) && ld < ((int64_t)1 << 31) && lddo < ((int64_t)1 << 31) &&
      (uintptr_t)workspace % 16 == 0 && (uintptr_t)qkv % 16 == 0 && (uintptr_t)dout % 16 == 0 && (uintptr_t)out % 2 == 0 &&
      (uintptr_t)dqkv % 8 == 0 && ld % 8 == 0 && lddo % 8 == 0 && lddq % 4 == 0) {
    // MFMA route: delta, the three transposed operands, then the two kernels
    const int S_pad = (S + 63) / 64 * 64;
    const bf16_t* qkvb = static_cast<const bf16_t*>(qkv);
    const bf16_t* dob = static_cast<const bf16_t*>(dout);
    bf16_t* qT = static_cast<bf16_t*>(workspace);
    bf16_t* kT = reinterpret_cast<bf16_t*>(static_cast<char*>(workspace) + mhsa_vt_bytes(B, S, H, D));
    bf16_t* doT = reinterpret_cast<bf16_t*>(static_cast<char*>(workspace) + 2 * mhsa_vt_bytes(B, S, H, D));
    float* lse2p = reinterpret_cast<float*>(static_cast<char*>(workspace) + 3 * mhsa_vt_bytes(B, S, H, D));
    float* deltap = lse2p + (int64_t)B * H * S_pad;
    hipLaunchKernelGGL(mhsa_delta_kernel, dim3((unsigned)((units + 3) / 4)), dim3(256), 0, st,
                       static_cast<const bf16_t*>(out), ldo, dob, lddo, delta, lse, lse2p, deltap, S, S_pad, H, D, units);
    const dim3 tgrid(S_pad / 64, H, B), tblock(256);
    hipLaunchKernelGGL(transpose_v_kernel, tgrid, tblock, 0, st, qkvb, ld, S, S_pad, H, D, C, qT, 0);
    hipLaunchKernelGGL(transpose_v_kernel, tgrid, tblock, 0, st, qkvb, ld, S, S_pad, H, D, C, kT, C);
    hipLaunchKernelGGL(transpose_v_kernel, tgrid, tblock, 0, st, dob, lddo, S, S_pad, H, D, C, doT, 0);
    const dim3 grid128((S + 32 * ATT_BWD_NW - 1) / (32 * ATT_BWD_NW), H, B), block256(64 * ATT_BWD_NW);
    const float sl2 = scale * 1.44269504088896340736f;
#define ANEMOI_MHSA_BWD_(DD, DR, WN)                                                                                     \
  do {                                                                                                                  \
    hipLaunchKernelGGL((mhsa_bwd_dkv_mfma_kernel<DD, DR, WN>), grid128, block256, 0, st, qkvb, ld, dob, lddo, qT, doT,  \
                       lse2p, deltap, static_cast<bf16_t*>(dqkv), lddq, S, S_pad, H, C, window, scale, sl2, dr);       \
    hipLaunchKernelGGL((mhsa_bwd_dq_mfma_kernel<DD, DR, WN>), grid128, block256, 0, st, qkvb, ld, dob, lddo, kT, lse,   \
                       delta, static_cast<bf16_t*>(dqkv), lddq, S, S_pad, H, C, window, scale, sl2, dr);                \
  } while (0)
#define ANEMOI_MHSA_BWD(DD, DR)                       \
  do {                                                \
    if (window >= 0) ANEMOI_MHSA_BWD_(DD, DR, true);  \
    else ANEMOI_MHSA_BWD_(DD, DR, false);             \
  } while (0)
    if (D == 64) {
      if (drop) ANEMOI_MHSA_BWD(64, true);
      else ANEMOI_MHSA_BWD(64, false);
    } else {
      if (drop) ANEMOI_MHSA_BWD(32, true);
      else ANEMOI_MHSA_BWD(32, false);
    }
#undef ANEMOI_MHSA_BWD_
#undef ANEMOI_MHSA_BWD
    return check_launch("anemoi_mhsa_backward(bf16, MFMA)");
  }
  dim3 grid((unsigned)((units + 3) / 4)), block(256);
#define BWD(T, DM)                                                                                                    \
  do {                                                                                                                \
    hipLaunchKernelGGL((mhsa_bwd_dq_kernel<T, DM>), grid, block, 0, st, static_cast<const T*>(qkv), ld,               \
                       static_cast<const T*>(out), ldo, static_cast<const T*>(dout), lddo, lse, delta,                \
                       static_cast<T*>(dqkv), lddq, S, H, D, C, window, scale, units, dr);                            \
    hipLaunchKernelGGL((mhsa_bwd_dkv_kernel<T, DM>), grid, block, 0, st, static_cast<const T*>(qkv), ld,              \
                       static_cast<const T*>(dout), lddo, lse, static_cast<const float*>(delta), static_cast<T*>(dqkv), \
                       lddq, S, H, D, C, window, scale, units, dr);                                                   \
  } while (0)
  if (dtype == ANEMOI_F32) {
    if (D <= 32) BWD(float, 32);
    else if (D <= 64) BWD(float, 64);
    else BWD(float, 128);
  } else if (dtype == ANEMOI_BF16) {
    if (D <= 32) BWD(bf16_t, 32);
    else if (D <= 64) BWD(bf16_t, 64);
    else BWD(bf16_t, 128);
  } else {
    return fail(ANEMOI_ERR_UNSUPPORTED, "anemoi_mhsa_backward: dtype %d", dtype);
  }
#undef BWD
  return check_launch("anemoi_mhsa_backward");
}

}  // extern "C"
