// Fused Linear  y = act(x @ W^T + bias) + residual  on the gfx950 matrix cores.
//
// Layout: x [M, K] row-major (ldx), W [N, K] row-major (nn.Linear weight), so both operands are
// K-contiguous ("B^T input") and fragments are plain 16-byte reads.
//
// Tile: 128 (M) x 128 (N) per 256-thread workgroup, K-slab of 128 bytes per row (64 bf16 / 32 f32),
// 4 waves as 2 (M) x 2 (N), each wave owns a 64 x 64 output tile:
//   bf16: 4 x 4 tiles of v_mfma_f32_16x16x32_bf16,   f32: 2 x 2 tiles of v_mfma_f32_32x32x2_f32
//         (exact f32: the K-ordered fmaf chain of cdna_hip_programming.md section 3).
// The MFMA "A" operand is the W fragment and "B" the x fragment, i.e. the instruction computes the
// transposed tile, so that every lane ends up with 4 CONSECUTIVE output columns of one row: the
// epilogue (bias, GELU/SiLU, residual, down-conversion) works on 8/16-byte vectors.
//
// Staging: global -> LDS directly with global_load_lds_dwordx4 (no VGPR round trip), double buffered,
// one barrier per K-slab.  The LDS image is lane-linear (8 rows x 128 B per wave instruction), so the
// bank swizzle is applied to the per-lane SOURCE address and undone on the fragment read
// (chunk' = chunk ^ ((row >> 1) & 7)): conflict-free for both fragment shapes.
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "common.hpp"

// Lab switches (tools/micro/ab_gemm.sh): cache-policy bits of the slab DMAs (buffer_load ... lds aux: 1 = sc0, 2 = nt,
// 16 = sc1).  The product builds with 0 / 0.
#ifndef ANEMOI_LAB_X_AUX
#define ANEMOI_LAB_X_AUX 0
#endif
#ifndef ANEMOI_LAB_W_AUX
#define ANEMOI_LAB_W_AUX 0
#endif

namespace anemoi {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

constexpr int BM = 128, BN = 128, ROW_BYTES = 128;
constexpr int TILE_BYTES = BM * ROW_BYTES;  // 16 KiB per operand per stage

__device__ __forceinline__ void glds16(const void* gptr, void* lptr) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gptr,
                                   (__attribute__((address_space(3))) void*)lptr, 16, 0, 0);
}

__device__ __forceinline__ int swz(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }

// LayerNorm folded into the Linear that consumes it (anemoi_linear_ln): with W' = W * gamma (columns scaled),
// s[n] = sum_k W'[n,k] and stats[m] = { rstd_m, -mean_m rstd_m },
//   LN(x) W^T + b  =  rstd_m (x W'^T)[m,n] + (-mean_m rstd_m) s[n] + (b + W beta)[n]
// i.e. the accumulator is scaled per row and shifted per (row, column) before bias / activation.
struct LnFold {
  const float* colsum;  // s[n], nullptr = plain Linear
  const float2* stats;  // per row of x
  // optional OUTPUT of the four-wave bf16 kernel: per row and 128-column slot the { sum, sum of squares } of the bf16
  // values it stores (slot = 2 * column tile + wave column), rs_slots = N / 128 -- the LayerNorm statistics of y
  // without reading y again (anemoi_linear_stats)
  float2* rs_partial;
  int rs_slots;
  int64_t* rs_rows_done;  // host side only: the launcher reports how many leading rows got their partials
  // batched launch of the four-wave kernel (anemoi_linear_batched): b_count independent problems of b_tiles tiles each,
  // operand bases b * b_sx / b_sw / b_sy elements apart; b_tiles = 0: one problem
  int64_t b_tiles, b_sx, b_sw, b_sy;
  int b_count;
  // staggered start of the four-wave kernel (round 6; set by its launcher): workgroup i of an XCD waits
  // (i % stagger_phases) * stagger_unit x 1024 cycles before its first tile.  Every tile of a launch takes the same time, so the
  // 256 persistent workgroups otherwise run in lockstep for the whole launch and their epilogues store 128 KiB each AT THE
  // SAME TIME (6.4 us per tile against the 3.7 us a CU's own store path needs, profiles/r05_gemm_store_path.md) while
  // nobody stages; out of phase one group's stores interleave with the others' K loops.  0 / 1 phases: off.
  int stagger_phases, stagger_unit;
};

template <typename T, typename TO, int VEC>
__device__ __forceinline__ void epilogue_store(const float (&acc)[VEC], int64_t m, int n, int64_t M, int N,
                                               const float* __restrict__ bias, const T* __restrict__ R, int64_t ldr,
                                               TO* __restrict__ Y, int64_t ldy, int act, bool vec_ok, LnFold ln) {
  if (m >= M || n >= N) return;
  float o[VEC];
  float2 st = make_float2(1.f, 0.f);
  if (ln.stats != nullptr) st = ln.stats[m];
  if (vec_ok && n + VEC <= N) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) o[i] = acc[i];
    if (ln.stats != nullptr) {
      float sv[VEC];
      VecIO<float, VEC>::load(ln.colsum + n, sv);
#pragma unroll
      for (int i = 0; i < VEC; ++i) o[i] = fmaf(o[i], st.x, st.y * sv[i]);
    }
    if (bias != nullptr) {
      float b[VEC];
      VecIO<float, VEC>::load(bias + n, b);
#pragma unroll
      for (int i = 0; i < VEC; ++i) o[i] += b[i];
    }
#pragma unroll
    for (int i = 0; i < VEC; ++i) o[i] = act_apply(o[i], act);
    if (R != nullptr) {
      float r[VEC];
      VecIO<T, VEC>::load(R + m * ldr + n, r);
#pragma unroll
      for (int i = 0; i < VEC; ++i) o[i] += r[i];
    }
    VecIO<TO, VEC>::store(Y + m * ldy + n, o);
  } else {
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      if (n + i < N) {
        float t = acc[i];
        if (ln.stats != nullptr) t = fmaf(t, st.x, st.y * ln.colsum[n + i]);
        t += (bias != nullptr ? bias[n + i] : 0.f);
        t = act_apply(t, act);
        if (R != nullptr) t += Elem<T>::load(R + m * ldr + n + i);
        Elem<TO>::store(Y + m * ldy + n + i, t);
      }
    }
  }
}

template <typename T, typename TO>
__global__ __launch_bounds__(256) void linear_kernel(const T* __restrict__ X, int64_t ldx, const T* __restrict__ W,
                                                     const float* __restrict__ bias, const T* __restrict__ R,
                                                     int64_t ldr, TO* __restrict__ Y, int64_t ldy, int64_t M, int N,
                                                     int K, int act, int vec_ok, LnFold ln, int64_t batch_sx,
                                                     int64_t batch_sw, int64_t batch_sy) {
  __shared__ __attribute__((aligned(16))) char smem[4 * TILE_BYTES];  // 2 stages x (x tile + W tile) = 64 KiB
  X += (int64_t)blockIdx.y * batch_sx;  // batched launch (anemoi_linear_batched): independent problems along grid.y
  W += (int64_t)blockIdx.y * batch_sw;
  Y += (int64_t)blockIdx.y * batch_sy;
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int nt_count = (N + BN - 1) / BN;
  const int nt = blockIdx.x % nt_count;
  const int64_t mt = blockIdx.x / nt_count;
  const int64_t m0 = mt * BM;
  const int n0 = nt * BN;
  const int nk = (int)(((int64_t)K * sizeof(T)) / ROW_BYTES);

  // ---- staging: wave `wid` moves row groups wid*4 .. wid*4+3 (8 rows x 128 B each) of both tiles
  const int srow = lane >> 3;  // row inside the 8-row group
  const int scp = lane & 7;    // 16-byte chunk position inside the LDS row
  const char* xg[4];
  const char* wg[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = (wid * 4 + i) * 8 + srow;
    const int c = swz(r, scp);  // source chunk that must land at position scp (XOR is an involution)
    int64_t gm = m0 + r;
    if (gm > M - 1) gm = M - 1;
    int gn = n0 + r;
    if (gn > N - 1) gn = N - 1;
    xg[i] = reinterpret_cast<const char*>(X + gm * ldx) + c * 16;
    wg[i] = reinterpret_cast<const char*>(W + (int64_t)gn * K) + c * 16;
  }
  auto stage = [&](int kt, int buf) {
    char* xs = smem + buf * (2 * TILE_BYTES) + wid * 4096;
    char* ws = xs + TILE_BYTES;
    const int64_t koff = (int64_t)kt * ROW_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      glds16(xg[i] + koff, xs + i * 1024);
      glds16(wg[i] + koff, ws + i * 1024);
    }
  };

  const int wr = wid >> 1, wc = wid & 1;

  if constexpr (sizeof(T) == 2) {
    // ------------------------------------------------------------------ bf16: 16x16x32, 4x4 tiles per wave
    const int fr = lane & 15, fq = lane >> 4;
    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    stage(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's LDS-DMA of tile kt has landed (explicit: the
      __syncthreads();                                    // compiler does not always count it at the barrier)
      if (kt + 1 < nk) stage(kt + 1, (kt + 1) & 1);
      const char* xs = smem + (kt & 1) * (2 * TILE_BYTES);
      const char* ws = xs + TILE_BYTES;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8_t a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = wc * 64 + i * 16 + fr;
          a[i] = *reinterpret_cast<const bf16x8_t*>(ws + row * ROW_BYTES + (swz(row, ks * 4 + fq) << 4));
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int row = wr * 64 + j * 16 + fr;
          b[j] = *reinterpret_cast<const bf16x8_t*>(xs + row * ROW_BYTES + (swz(row, ks * 4 + fq) << 4));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
      }
    }
    // epilogue: lane holds C[m = .. + fr][n = .. + fq*4 + 0..3]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int64_t m = m0 + wr * 64 + j * 16 + fr;
        const int n = n0 + wc * 64 + i * 16 + fq * 4;
        const float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
        epilogue_store<T, TO, 4>(v, m, n, M, N, bias, R, ldr, Y, ldy, act, vec_ok != 0, ln);
      }
  } else {
    // ------------------------------------------------------------------ f32: 32x32x2, 2x2 tiles per wave
    const int fr = lane & 31, fh = lane >> 5;
    f32x16_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    stage(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (see the bf16 loop above)
      __syncthreads();
      if (kt + 1 < nk) stage(kt + 1, (kt + 1) & 1);
      const char* xs = smem + (kt & 1) * (2 * TILE_BYTES);
      const char* ws = xs + TILE_BYTES;
#pragma unroll
      for (int kc = 0; kc < 4; ++kc) {
        // half-wave fh reads chunk 2*kc + fh: its 4 floats feed 4 successive MFMAs.  The physical k
        // order differs from the logical one, identically for both operands, so the sum is the same.
        f32x4_t a[2], b[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int row = wc * 64 + i * 32 + fr;
          a[i] = *reinterpret_cast<const f32x4_t*>(ws + row * ROW_BYTES + (swz(row, kc * 2 + fh) << 4));
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int row = wr * 64 + j * 32 + fr;
          b[j] = *reinterpret_cast<const f32x4_t*>(xs + row * ROW_BYTES + (swz(row, kc * 2 + fh) << 4));
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][t], b[j][t], acc[i][j], 0, 0, 0);
      }
    }
    // epilogue: lane holds C[m = .. + fr][n = .. + 8*g + 4*fh + 0..3] for g = 0..3
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int64_t m = m0 + wr * 64 + j * 32 + fr;
          const int n = n0 + wc * 64 + i * 32 + 8 * g + 4 * fh;
          const float v[4] = {acc[i][j][4 * g + 0], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2],
                              acc[i][j][4 * g + 3]};
          epilogue_store<T, TO, 4>(v, m, n, M, N, bias, R, ldr, Y, ldy, act, vec_ok != 0, ln);
        }
  }
}

// =============================================================================================
// bf16 256 x 256 tile, K-slab 64, two LDS stages of 64 KiB: twice the arithmetic intensity per L2 / LDS byte of the
// 128 x 128 kernel (which is L2-bandwidth bound on this chip beyond ~900 TFLOP/s).  Tiles are mapped XCD-aware: the
// 8 XCDs each walk a contiguous range of the tile list, so the tiles that share an operand panel run on the same XCD
// (private L2) back to back.
// =============================================================================================
constexpr int BIG_M = 256, BIG_N = 256;
constexpr int BIG_STAGE = (BIG_M + BIG_N) * ROW_BYTES;  // 64 KiB

__device__ __forceinline__ uint32_t bf16x2_add(uint32_t a, uint32_t b) {
  const float lo = __uint_as_float(a << 16) + __uint_as_float(b << 16);
  const float hi = __uint_as_float(a & 0xffff0000u) + __uint_as_float(b & 0xffff0000u);
  return pack_bf16x2(lo, hi);
}

// GELU of two values on the packed-f32 pipe (v_pk_fma_f32: two lanes' worth per issue slot, no transcendentals).
// Phi(x) = 0.5 + x Q(x^2), |x| clamped to 4.5: a degree-8 weighted minimax fit of (Phi(x) - 1/2) / x in x^2, Horner in
// f32 (13 instructions per pair; the degree-10 fit on [-5, 5] this replaces took 18 and cost 1.5 ms per forward).
// |GELU error| < 5.5e-5 everywhere (relative < 1e-4 for x > 0.05), i.e. 1/40 of the bf16 rounding of the value this
// kernel stores; the f32 kernel keeps the erf form in common.hpp::act_apply.  Replaces nn.GELU() of
// layers/block.py:504-508.  (tools/gelu_fit.py regenerates the coefficients.)
typedef __attribute__((ext_vector_type(2))) float f32x2_t;

template <int ACT>
__device__ __forceinline__ f32x2_t act_apply2(f32x2_t x) {
  if constexpr (ACT == ANEMOI_ACT_GELU) {
    const f32x2_t xc = {__builtin_amdgcn_fmed3f(x.x, -4.5f, 4.5f), __builtin_amdgcn_fmed3f(x.y, -4.5f, 4.5f)};
    const f32x2_t t = xc * xc;
    f32x2_t q = 3.036384685350946e-11f;
    q = q * t + -3.31462968183871e-09f;
    q = q * t + 1.5909856188045524e-07f;
    q = q * t + -4.451706445252057e-06f;
    q = q * t + 8.142489969031885e-05f;
    q = q * t + -0.0010375895071774721f;
    q = q * t + 0.009585196152329445f;
    q = q * t + -0.06598464399576187f;
    q = q * t + 0.3987126052379608f;
    const f32x2_t phi = xc * q + 0.5f;
    return x * phi;
  } else {
    return f32x2_t{act_apply(x.x, ACT), act_apply(x.y, ACT)};
  }
}

// act'(x) for the backward's dX epilogue (GMUL mode of the four-wave kernel): out = acc * act'(pre).  GELU: the derivative
// d/dx [x Phi(x)] = Phi(x) + x phi(x) as 1/2 + x_c S(x_c^2), x_c = clamp(x, +-4), S of degree 8 fitted to the exact
// derivative on Chebyshev nodes (|error| < 5.5e-4 everywhere, 6.5e-5 inside the clamp; bf16 resolves 3.9e-3) -- the same
// ten packed instructions per pair of values as the forward's polynomial, no transcendental.
template <int ACT>
__device__ __forceinline__ f32x2_t act_grad2(f32x2_t x) {
  if constexpr (ACT == ANEMOI_ACT_GELU) {
    const f32x2_t xc = {__builtin_amdgcn_fmed3f(x.x, -4.0f, 4.0f), __builtin_amdgcn_fmed3f(x.y, -4.0f, 4.0f)};
    const f32x2_t t = xc * xc;
    f32x2_t q = 9.3872902156732022e-10f;
    q = q * t + -7.9411323687314723e-08f;
    q = q * t + 2.950694481564598e-06f;
    q = q * t + -6.3805510857940873e-05f;
    q = q * t + 0.00089759489254071564f;
    q = q * t + -0.0086697042593016048f;
    q = q * t + 0.05833774383148245f;
    q = q * t + -0.26469170897426864f;
    q = q * t + 0.79756480213010039f;
    return xc * q + 0.5f;
  } else if constexpr (ACT == ANEMOI_ACT_SILU) {
    const f32x2_t sg = {__frcp_rn(1.0f + __expf(-x.x)), __frcp_rn(1.0f + __expf(-x.y))};
    return sg * (x * (1.0f - sg) + 1.0f);
  } else {
    return f32x2_t{x.x > 0.f ? 1.f : 0.f, x.y > 0.f ? 1.f : 0.f};
  }
}

// Tile order inside an XCD chunk: column groups of SUPER_N tile columns, row-major inside a group.  With SUPER_N = 8
// the 32 tiles an XCD works on concurrently form a 4 x 8 patch of the output: 12 unique operand panels per K-slab step
// in that XCD's L2 instead of 18 for the 2 x 16 patch of the plain order.  The eight-wave kernel did not care (-3 %);
// the four-wave kernel stages 64 KiB per ~1.4 us and CU (~48 GB/s, what an L2 MISS stream sustains per CU) and gains
// +11 % at 8192^3, +2..6 % at the model's shapes (SUPER_N 4 / 16 / unbounded measured: 1437 / 1396 / 1343 vs 1489).
constexpr int SUPER_N = 8;
__device__ __forceinline__ void tile_coords(int64_t t64, int nt_count, int64_t mt_count64, int64_t& mt, int& nt) {
  // tile counts fit 31 bits (checked by the launcher): 32-bit divisions (the 64-bit ones are ~200 instructions each).
  // Column groups of SUPER_N tile columns, row-major inside a group; the last group takes the remainder (8..15 wide).
  const unsigned t = (unsigned)t64, mt_count = (unsigned)mt_count64;
  const unsigned groups = (unsigned)nt_count / SUPER_N > 1 ? (unsigned)nt_count / SUPER_N : 1;
  const unsigned per_group = mt_count * SUPER_N;
  unsigned cg = t / per_group;
  if (cg > groups - 1) cg = groups - 1;
  const unsigned r = t - cg * per_group;
  const unsigned width = cg == groups - 1 ? (unsigned)nt_count - cg * SUPER_N : (unsigned)SUPER_N;
  mt = r / width;
  nt = (int)(cg * SUPER_N + r % width);
}

// =============================================================================================
// bf16 fast path, "w4": a persistent 256 x 256 x 64 tile and slab stream, FOUR waves (one per SIMD) of 128 x 128 --
// 256 accumulator registers per lane live in AGPRs (inline-asm MFMA, "+a"), the 128 fragment registers of the
// current and the next half-slab in VGPRs.  Measured on MI355X (tools/micro/gemm_lab.hip, 8192^3, random data):
// 1320 TFLOP/s against 1110 for the eight-wave (128 x 64 per wave) loop it replaced; what made the difference, in order:
//   * a third less LDS read traffic (each wave reads 256 rows per slab instead of 192 for half the flops);
//   * ONE memory instruction per MFMA gap, never a burst: the CU's four waves run in lockstep and share one
//     address pipe (~16 cycles per 1 KiB LDS-DMA piece), so the 16 refill DMAs of a slab go out one per five
//     MFMAs (~85 cycles); at one per three they queued on each other and cost ~37 cycles each of MFMA time;
//   * the ks = 1 fragments of a slab are fetched in its first 16 MFMA gaps, so barrier 1 (gap 23) frees the WHOLE
//     slab buffer early and its refill overlaps the remaining 100 MFMAs; barrier 2 (gap 103, counted vmcnt: only
//     the previous slab's DMAs must have landed) publishes the other buffer, whose ks = 0 fragments are fetched in
//     the gaps 104..119.
// Operand rows beyond M / N are never loaded: the staging goes through buffer descriptors sized to the tile's
// valid rows (out-of-range lanes of buffer_load ... lds deliver zeros).
// =============================================================================================
constexpr int STAGGER_PHASES = 2, STAGGER_UNIT = 16, STAGGER_MIN_ROUNDS = 2, STAGGER_MIN_K = 512;  // (launcher; measured r06)
constexpr int W4_LDS = 2 * BIG_STAGE + 4096;        // 132 KiB: two slab buffers + the tile's bias, LN column sums, LN row statistics

#define ANEMOI_MFMA_A(c, a, b) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b))

template <typename F, int... S>
__device__ __forceinline__ void static_for_seq(F&& f, std::integer_sequence<int, S...>) {
  (f(std::integral_constant<int, S>{}), ...);
}

// One output column of up to ROWS "skinny" rows (the M % 256 <= 8 tail of a tall GEMM): the wave sweeps K with 16-byte
// loads, f32 accumulation, wave reduction; lane 0 applies the epilogue (same double rounding as the tiled kernels).
template <int ROWS>
__device__ __forceinline__ void skinny_column(const bf16_t* __restrict__ X, int64_t ldx, const bf16_t* __restrict__ W,
                                              const float* __restrict__ bias, const bf16_t* __restrict__ R, int64_t ldr,
                                              bf16_t* __restrict__ Y, int64_t ldy, int M, int n, int K, int act,
                                              LnFold ln, int lane) {
  float acc[ROWS];
#pragma unroll
  for (int r = 0; r < ROWS; ++r) acc[r] = 0.f;
  const bf16_t* wrow = W + (int64_t)n * K;
  for (int k = lane * 8; k < K; k += 512) {
    float wv[8];
    VecIO<bf16_t, 8>::load(wrow + k, wv);
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      if (r < M) {
        float xv[8];
        VecIO<bf16_t, 8>::load(X + r * ldx + k, xv);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[r] = fmaf(xv[i], wv[i], acc[r]);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < ROWS; ++r) acc[r] = wave_sum(acc[r]);
  if (lane == 0) {
    const float b = bias != nullptr ? bias[n] : 0.f;
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      if (r < M) {
        float a = acc[r];
        if (ln.stats != nullptr) a = fmaf(a, ln.stats[r].x, ln.stats[r].y * ln.colsum[n]);
        float o = act_apply(a + b, act);
        o = bf16_to_f32(f32_to_bf16(o));  // bf16 result, then + residual
        if (R != nullptr) o += bf16_to_f32(R[r * ldr + n]);
        Y[r * ldy + n] = f32_to_bf16(o);
      }
    }
  }
}

// The tail rows of a tiled launch (M % 256 in 1..8), FOUR output columns per wave at a time: the x rows are loaded once for
// the four columns and all loads of a K chunk are independent, so a wave pays ONE memory latency chain for its columns
// instead of one per column (5121 x 4096 x 1024: the column-at-a-time pass cost 14 us of a 58-us launch, every launch of a
// 40 962- / 10 242- / 5 121-row problem 3 - 14 us).  Per column the arithmetic is skinny_column's, operation for operation
// (k order per lane, wave_sum, epilogue): the same bits.
// EPI 1 / 2: the DUAL / GMUL epilogues of the four-wave kernel (R = second output / saved pre-activation), ACT_T = its ACT.
template <int ROWS, int EPI = 0, int ACT_T = 0>
__device__ __forceinline__ void skinny_columns(const bf16_t* __restrict__ X, int64_t ldx, const bf16_t* __restrict__ W,
                                               const float* __restrict__ bias, const bf16_t* __restrict__ R, int64_t ldr,
                                               bf16_t* __restrict__ Y, int64_t ldy, int M, int N, int K, int act, LnFold ln,
                                               int lane, int gw, int total_waves) {
  constexpr int CG = 4;
  const int groups = (N + CG - 1) / CG;
  for (int g = gw; g < groups; g += total_waves) {
    const int n0 = g * CG;
    float acc[ROWS][CG];
#pragma unroll
    for (int r = 0; r < ROWS; ++r)
#pragma unroll
      for (int c = 0; c < CG; ++c) acc[r][c] = 0.f;
    for (int k = lane * 8; k < K; k += 512) {
      float xv[ROWS][8], wv[CG][8];
#pragma unroll
      for (int c = 0; c < CG; ++c) {
        const int n = n0 + c < N ? n0 + c : N - 1;  // (a ragged last group recomputes the last column; not stored)
        VecIO<bf16_t, 8>::load(W + (int64_t)n * K + k, wv[c]);
      }
#pragma unroll
      for (int r = 0; r < ROWS; ++r)
        if (r < M) VecIO<bf16_t, 8>::load(X + r * ldx + k, xv[r]);
#pragma unroll
      for (int c = 0; c < CG; ++c)
#pragma unroll
        for (int r = 0; r < ROWS; ++r)
          if (r < M) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[r][c] = fmaf(xv[r][i], wv[c][i], acc[r][c]);
          }
    }
#pragma unroll
    for (int r = 0; r < ROWS; ++r)
#pragma unroll
      for (int c = 0; c < CG; ++c) acc[r][c] = wave_sum(acc[r][c]);
    if (lane == 0) {
#pragma unroll
      for (int c = 0; c < CG; ++c) {
        const int n = n0 + c;
        if (n < N) {
          const float b = bias != nullptr ? bias[n] : 0.f;
#pragma unroll
          for (int r = 0; r < ROWS; ++r) {
            if (r < M) {
              float a = acc[r][c];
              if constexpr (EPI == 1) {  // pre-activation (as the backward will read it) next to the activation of the f32 sum
                const_cast<bf16_t*>(R)[r * ldr + n] = f32_to_bf16(a + b);
                Y[r * ldy + n] = f32_to_bf16(act_apply(a + b, act));
                continue;
              } else if constexpr (EPI == 2) {  // (x W^T) * act'(pre)
                const float pv = bf16_to_f32(R[r * ldr + n]);
                Y[r * ldy + n] = f32_to_bf16((a + b) * act_grad2<ACT_T>(f32x2_t{pv, pv}).x);
                continue;
              }
              if (ln.stats != nullptr) a = fmaf(a, ln.stats[r].x, ln.stats[r].y * ln.colsum[n]);
              float o = act_apply(a + b, act);
              o = bf16_to_f32(f32_to_bf16(o));  // bf16 result, then + residual
              if (R != nullptr) o += bf16_to_f32(R[r * ldr + n]);
              Y[r * ldy + n] = f32_to_bf16(o);
            }
          }
        }
      }
    }
  }
}

template <int ACT, bool HAS_RES, bool LN, int MH, bool RS = false, bool DUAL = false, bool GMUL = false>
__global__ __launch_bounds__(256) void linear_bf16_w4_kernel(const bf16_t* __restrict__ X, int64_t ldx,
                                                             const bf16_t* __restrict__ W,
                                                             const float* __restrict__ bias,
                                                             const bf16_t* __restrict__ R, int64_t ldr,
                                                             bf16_t* __restrict__ Y, int64_t ldy, int64_t M, int N,
                                                             int K, int vec_ok, int64_t n_tiles, int nt_count,
                                                             LnFold ln, int m_tail) {
  // DUAL (training forward, anemoi_linear_dual): R is a second OUTPUT [M, ldr] that receives the pre-activation
  // x W^T + b (rounded to bf16) next to Y = act(x W^T + b) -- the backward needs act'(pre), and a separate activation
  // pass over [M, 4C] costs more than the extra 16-byte store per lane and row group here.  No residual in this mode.
  static_assert(!DUAL || (!HAS_RES && !LN && !RS && ACT != 0), "DUAL: activation, no residual / LayerNorm fold / row sums");
  // GMUL (backward, anemoi_linear_actgrad): R is the saved pre-activation, the result is acc * act'(R) -- the dX GEMM of
  // the Linear BEHIND an activation delivers the gradient of the Linear IN FRONT of it, no separate act' pass.
  static_assert(!GMUL || (HAS_RES && !LN && !RS && !DUAL && ACT != 0), "GMUL: pre-activation in R, no other epilogue mode");
  // MH = 16-row fragments per wave along M: 8 -> the 256 x 256 tile, 4 -> a 128 x 256 tile (wave tile 64 x 128) used for
  // the rows of a remainder round (640 tiles on 256 CUs: the last 128 tiles become 256 half tiles = one full round),
  // 6 -> a 192 x 256 tile for small problems whose 256-row tiles quantise badly (5121 x 4096: 320 tiles = 2 rounds, as
  // 432 tiles of 7/8 the staged bytes = 2 shorter rounds; 5121 x 2048: 160 -> 216 of the 256 CUs busy, shorter tiles),
  // 5 -> 160 x 256 (5121 x 2048: 256 tiles fill the chip exactly once), 3 -> 96 x 256 (N = 512 / 1024 at M = 10 242 / 5 121:
  // 214 / 216 tiles in one round instead of 160).  The launcher's cost model picks the height per launch.
  constexpr int TM = MH * 32;        // tile rows
  constexpr int NS = 8 * MH;         // MFMAs per 32-deep K step
  constexpr int NRD = 8 + MH;        // fragment reads per K step = LDS-DMA instructions per slab and wave
  constexpr int G1 = MH == 8 ? 23 : MH == 6 ? 17 : MH == 5 ? 16 : MH == 4 ? 15 : 13,
                SP = MH == 8 ? 5 : MH == 4 ? 3 : MH == 3 ? 2 : 4,
                G2 = MH == 8 ? 103 : MH == 6 ? 78 : MH == 5 ? 66 : MH == 4 ? 51 : 36;  // schedule (see below)
  static_assert(MH == 3 || MH == 4 || MH == 5 || MH == 6 || MH == 8, "tile heights with a measured slab schedule");
  static_assert(G1 + 1 + (NRD - 1) * SP < G2 && G2 + NRD < 2 * NS, "slab schedule");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int64_t xcd = blockIdx.x & 7, bix = blockIdx.x >> 3, bpx = gridDim.x >> 3;
  const int64_t q8 = n_tiles / 8, r8 = n_tiles % 8;
  const int64_t chunk_start = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
  const int64_t chunk_len = q8 + (xcd < r8 ? 1 : 0);
  const bool has_tiles = bix < chunk_len;  // (a workgroup without tiles still takes its share of the tail rows below)
  const int nk = K / 64;  // >= 2 (launcher)

  // ---- staging side.  Wave w fills LDS rows w * 64 + 8 i + (lane >> 3), i = 0..7, of both operand panels; the
  //      source chunk is swizzled with the LDS row ((row >> 1) & 7 = (4 i + (lane >> 4)) & 7: two classes, i even/odd).
  //      The x panel is staged row for row.  The W panel is staged PERMUTED inside every 32-row group,
  //      LDS row 32 g + l  <-  W row 32 g + 8 ((l & 15) >> 2) + 4 (l >> 4) + (l & 3),
  //      so that the two 16-row fragments of a group give each lane EIGHT adjacent output columns (fq * 8 + 0..7): the
  //      epilogue stores 16 bytes per lane straight from the MFMA layout, no LDS transposition.  For the staging lane
  //      (l = 8 (i & 3) + (lane >> 3)) the permuted row splits into a lane part and a wave-uniform part.
  const int srow = lane >> 3, scp = lane & 7;
  int vox[2], vow[2];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int r = 8 * p + srow;
    const int c = swz(r, scp);
    // odd MH: the x rows of the odd waves start 8 rows into a 16-row swizzle period (wave w stages LDS rows w * MH * 8 ...),
    // so their two piece classes (LDS row mod 16 < 8 / >= 8) trade swizzles
    const int cx = (MH % 2 != 0 && (wid & 1)) ? swz(r ^ 8, scp) : c;
    vox[p] = r * (int)ldx * 2 + cx * 16;
    vow[p] = (8 * (srow >> 2) + (srow & 3)) * K * 2 + c * 16;
  }
  const int xrow16 = 16 * (int)ldx * 2;
  const int xwave = wid * (MH * 8) * (int)ldx * 2, wwave = wid * 64 * K * 2;
  __amdgpu_buffer_rsrc_t xrs, wrs;  // descriptors of the tile whose slabs are being staged
  const int64_t tiles_per_problem = ln.b_tiles > 0 ? ln.b_tiles : n_tiles;
  auto set_tile = [&](int64_t tile) {
    int64_t mt_;
    int nt_;
    const int64_t pb = ln.b_tiles > 0 ? tile / ln.b_tiles : 0;  // problem of a batched launch
    tile_coords(tile - pb * tiles_per_problem, nt_count, tiles_per_problem / nt_count, mt_, nt_);
    const int64_t m0 = mt_ * TM;
    const int n0 = nt_ * BIG_N;
    int64_t xrows = M - m0 < TM ? M - m0 : TM;  // ragged last row tile: rows >= M read as zeros (a half tile of the
    xrows = xrows > 0 ? xrows : 0;              // remainder launch may lie entirely behind M)
    const int wrows = N - n0 < BIG_N ? N - n0 : BIG_N;
    xrs = __builtin_amdgcn_make_buffer_rsrc((void*)(X + pb * ln.b_sx + m0 * ldx), 0, (int)(xrows * ldx * 2), 0x00020000);
    wrs = __builtin_amdgcn_make_buffer_rsrc((void*)(W + pb * ln.b_sw + (int64_t)n0 * K), 0, wrows * K * 2, 0x00020000);
  };
  auto set_null = [&]() {
    xrs = __builtin_amdgcn_make_buffer_rsrc((void*)X, 0, 0, 0x00020000);
    wrs = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, 0, 0x00020000);
  };
  auto dma_x = [&](int i, int kt, char* dst) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (__attribute__((address_space(3))) void*)dst, 16, vox[i & 1],
                                             xwave + (i >> 1) * xrow16 + kt * ROW_BYTES, 0, ANEMOI_LAB_X_AUX);
  };
  auto dma_w = [&](int i, int kt, char* dst) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (__attribute__((address_space(3))) void*)dst, 16, vow[i & 1],
                                             wwave + (32 * (i >> 2) + 16 * (i & 1) + 4 * ((i >> 1) & 1)) * K * 2 + kt * ROW_BYTES,
                                             0, ANEMOI_LAB_W_AUX);
  };
  // LDS image of a slab: x panel rows 0 .. TM-1 at the stage base (wave w: rows w * MH * 8 ...), W panel at + 32 KiB
  auto stage_all = [&](int kt, int buf) {
    char* xs = smem + buf * BIG_STAGE + wid * (MH * 1024);
    char* ws = smem + buf * BIG_STAGE + BIG_M * ROW_BYTES + wid * 8192;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (i < MH) dma_x(i, kt, xs + i * 1024);
      dma_w(i, kt, ws + i * 1024);
    }
  };

  // ---- compute side
  const int wm = wid >> 1, wn = wid & 1;
  const int fr = lane & 15, fq = lane >> 4;
  auto ldA = [&](int buf, int ks, int i) {  // W fragment (N side)
    const int row = wn * 128 + i * 16 + fr;
    return *reinterpret_cast<const bf16x8_t*>(smem + buf * BIG_STAGE + BIG_M * ROW_BYTES + row * ROW_BYTES +
                                              (swz(row, ks * 4 + fq) << 4));
  };
  auto ldB = [&](int buf, int ks, int j) {  // x fragment (M side)
    const int row = wm * (MH * 16) + j * 16 + fr;
    return *reinterpret_cast<const bf16x8_t*>(smem + buf * BIG_STAGE + row * ROW_BYTES + (swz(row, ks * 4 + fq) << 4));
  };
  f32x4_t acc[8][MH];
  bf16x8_t a0[8], b0[MH], a1[8], b1[MH];

  // prologue: slabs 0 and 1 of the first tile; fragments of (slab 0, ks 0)
  // (issuing slab 1 together with slab 0 and waiting for slab 0 alone -- vmcnt(NRD) -- measured no difference on any shape:
  //  back to back in a stream the operands come out of the L2 / MALL, there is no cold-miss latency to overlap)
  // The up to 8 rows behind the last full row tile (M here is the 256-row multiple): every wave of the launch takes
  // some output columns of them first -- in parallel instead of a separate launch behind this one.  (Issued BEHIND the
  // first slab's DMAs, to share the latency the prologue waits for anyway, the pass measured slower on every configuration:
  // config 3 35.8 vs 35.6 ms, config 2 3.36 vs 3.28 ms, same boxes -- ANEMOI_LAB_TAIL_BEHIND_DMA keeps that order for A/B builds.)
#ifdef ANEMOI_LAB_TAIL_BEHIND_DMA
  if (has_tiles) {
    set_tile(chunk_start + bix);
    stage_all(0, 0);
  }
#endif
  if (m_tail > 0) {
    LnFold lt = ln;
    if (LN) lt.stats += M;
    // groups of four columns, wave 0 of every workgroup first (a launch's workgroups stay balanced: N = 1024 gives one group
    // to one wave of each of the 256 workgroups, not four to 64 of them)
    const int gw = wid * (int)gridDim.x + (int)blockIdx.x, tw = (int)gridDim.x * 4;
    const bf16_t* xt = X + M * ldx;
    const bf16_t* rt = (HAS_RES || DUAL) ? R + M * ldr : nullptr;
    bf16_t* yt = Y + M * ldy;
    constexpr int EPI = DUAL ? 1 : GMUL ? 2 : 0;
    if (m_tail == 1) skinny_columns<1, EPI, ACT>(xt, ldx, W, bias, rt, ldr, yt, ldy, 1, N, K, ACT, lt, lane, gw, tw);
    else if (m_tail == 2) skinny_columns<2, EPI, ACT>(xt, ldx, W, bias, rt, ldr, yt, ldy, 2, N, K, ACT, lt, lane, gw, tw);
    else if (m_tail <= 4) skinny_columns<4, EPI, ACT>(xt, ldx, W, bias, rt, ldr, yt, ldy, m_tail, N, K, ACT, lt, lane, gw, tw);
    else skinny_columns<8, EPI, ACT>(xt, ldx, W, bias, rt, ldr, yt, ldy, m_tail, N, K, ACT, lt, lane, gw, tw);
  }
  if (!has_tiles) return;
  if (ln.stagger_phases > 1)  // (phases by position inside the XCD; by XCD, or by both, measured the same: gpurun_out/r06_s24)
    for (int i = 0, n = (int)(bix % ln.stagger_phases) * ln.stagger_unit; i < n; ++i) __builtin_amdgcn_s_sleep(16);
#ifndef ANEMOI_LAB_TAIL_BEHIND_DMA
  set_tile(chunk_start + bix);
  stage_all(0, 0);
#endif
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    if (u < MH) b0[u] = ldB(0, 0, u);
    a0[u] = ldA(0, 0, u);
  }
  stage_all(1, 1);

  // A zero fragment for the MFMA-pipe zeroing of accumulators (D = 0 * 0 + 0), made opaque ONCE here: left a known
  // constant, the compiler re-materialises it (v_mov) directly in front of the inline-asm MFMA that reads it -- a
  // VALU-write -> MFMA-read hazard it does not see: stale operands, garbage instead of zeros.
  bf16x8_t zfrag = {0, 0, 0, 0, 0, 0, 0, 0};
  asm volatile("" : "+v"(zfrag));
  asm volatile("s_nop 3" ::: "memory");
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < MH; ++j)
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %1, 0" : "=a"(acc[i][j]) : "v"(zfrag));

  // ONE loop over the workgroup's slab stream (tile prologue and epilogue are conditional blocks inside it): with a
  // K loop nested in a tile loop the register allocator gave the accumulators different AGPRs inside and outside the
  // inner loop and permuted all 256 of them (v_accvgpr_mov) at its exits -- 1k cycles per tile, and placed directly
  // behind the last inline-asm MFMAs, whose write latency the compiler does not know: wrong values.
  int g = 0;  // running slab counter over all tiles of this workgroup: LDS stage = g & 1
  int64_t li = bix, tile = 0, m0 = 0, y_base = 0;
  int k = 0, n0 = 0;
  bool has_next = false;
  for (;;) {
    if (k == 0) {
      tile = chunk_start + li;
      has_next = li + bpx < chunk_len;
      int64_t mt_;
      int nt_;
      const int64_t pb = ln.b_tiles > 0 ? tile / ln.b_tiles : 0;
      y_base = pb * ln.b_sy;
      tile_coords(tile - pb * tiles_per_problem, nt_count, tiles_per_problem / nt_count, mt_, nt_);
      m0 = mt_ * TM;
      n0 = nt_ * BIG_N;
      // The tile's 256 bias values go to LDS by one 4-byte LDS-DMA per wave (columns >= N: zeros from the descriptor's
      // range check) and come back in the epilogue: 32 bias registers per lane would not survive the K loop unspilled.
      if (bias != nullptr) {
        const int nbias = N - n0 < BIG_N ? N - n0 : BIG_N;
        const __amdgpu_buffer_rsrc_t brs =
            __builtin_amdgcn_make_buffer_rsrc((void*)(bias + n0), 0, nbias * 4, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(
            brs, (__attribute__((address_space(3))) void*)(smem + 2 * BIG_STAGE + wid * 256), 4, lane * 4, wid * 256, 0,
            0);
      }
      if constexpr (LN) {  // the tile's column sums s[n] of W' (LayerNorm fold), same route
        const int ncs = N - n0 < BIG_N ? N - n0 : BIG_N;
        const __amdgpu_buffer_rsrc_t crs =
            __builtin_amdgcn_make_buffer_rsrc((void*)(ln.colsum + n0), 0, ncs * 4, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(
            crs, (__attribute__((address_space(3))) void*)(smem + 2 * BIG_STAGE + 1024 + wid * 256), 4, lane * 4,
            wid * 256, 0, 0);
        // ... and the tile's row statistics { rstd, -mean rstd } (TM x 8 bytes; rows >= M: zeros): fetched here, a whole
        // K loop ahead, they cost the epilogue no vector-memory wait (a load issued there sits behind the next tile's
        // slab DMAs in the in-order vmcnt queue)
        int srows = M - m0 < TM ? (int)(M - m0) : TM;
        srows = srows > 0 ? srows : 0;
        const __amdgpu_buffer_rsrc_t trs =
            __builtin_amdgcn_make_buffer_rsrc((void*)(ln.stats + m0), 0, srows * 8, 0x00020000);
#pragma unroll
        for (int h = 0; h < 2; ++h)
          if (h * 1024 + wid * 256 < TM * 8)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                trs, (__attribute__((address_space(3))) void*)(smem + 2 * BIG_STAGE + 2048 + h * 1024 + wid * 256), 4,
                lane * 4, h * 1024 + wid * 256, 0, 0);
      }
      // accumulator zeroing (prologue / previous epilogue, MFMA pipe) -> first MFMA: pinned on both sides
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_nop 7" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
    }

    // One slab = 128 MFMAs; the slab two ahead in the stream (this tile's or the next one's) is staged meanwhile.  The
    // control flow is one plain loop on purpose: with alternative slab bodies the accumulators' phi nodes fall out of
    // the AGPR class and every MFMA gets four v_accvgpr_write in front.  After the stream's last tile the staging
    // descriptors are empty (every lane out of range: zeros into a dead buffer), so the DMA count per slab stays 16.
    auto slab = [&](int kt_stage, bool vm_wait) {
      const int buf = g & 1, nbuf = buf ^ 1;
      char* xsd = smem + buf * BIG_STAGE + wid * (MH * 1024);
      char* wsd = smem + buf * BIG_STAGE + BIG_M * ROW_BYTES + wid * 8192;
      static_for_seq(
          [&](auto s_tag) {
            constexpr int s = decltype(s_tag)::value;
            if constexpr (s < NS) ANEMOI_MFMA_A(acc[s / MH][s % MH], a0[s / MH], b0[s % MH]);
            else ANEMOI_MFMA_A(acc[(s - NS) / MH][s % MH], a1[(s - NS) / MH], b1[s % MH]);
            if constexpr (s < NRD) {  // fragments of (this slab, ks = 1)
              if constexpr (s < MH) b1[s] = ldB(buf, 1, s);
              else a1[s - MH] = ldA(buf, 1, s - MH);
            }
            if constexpr (s == G1) {  // barrier 1: every wave has read this slab's buffer completely
              asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
              __builtin_amdgcn_sched_barrier(0);
              __builtin_amdgcn_s_barrier();
              __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (s > G1 && (s - G1 - 1) % SP == 0 && (s - G1 - 1) / SP < NRD) {  // refill it: one DMA per SP gaps
              constexpr int t = (s - G1 - 1) / SP;  // x and W pieces alternate while x pieces last
              if constexpr (t < 2 * MH && (t & 1) == 0) dma_x(t >> 1, kt_stage, xsd + (t >> 1) * 1024);
              else if constexpr (t < 2 * MH) dma_w(t >> 1, kt_stage, wsd + (t >> 1) * 1024);
              else dma_w(t - MH, kt_stage, wsd + (t - MH) * 1024);
            }
            if constexpr (s == G2) {  // barrier 2: the other buffer (staged one slab ago) is complete for everyone
              if (vm_wait)  // (slab 0 of a later tile: waited in the epilogue) -- all but this slab's own NRD refills
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NRD) : "memory");
              __builtin_amdgcn_sched_barrier(0);
              __builtin_amdgcn_s_barrier();
              __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (s > G2 && s - G2 - 1 < NRD) {  // fragments of (next slab, ks = 0)
              constexpr int t = s - G2 - 1;
              if constexpr (t < MH) b0[t] = ldB(nbuf, 0, t);
              else a0[t - MH] = ldA(nbuf, 0, t - MH);
            }
          },
          std::make_integer_sequence<int, 2 * NS>{});
      ++g;
    };
    if (k == nk - 2) {  // from here on the staged slabs are the next tile's
      if (has_next) set_tile(tile + bpx);
      else set_null();
    }
    // (with a residual, slab 0 of a later tile finds its data waited for by the epilogue's vmcnt(0); without one the
    //  epilogue only waits for slab 0 and every slab does its own counted wait)
    slab(k + 2 < nk ? k + 2 : k + 2 - nk, !HAS_RES || k != 0 || li == bix);
    ++k;
    if (k < nk) continue;
    k = 0;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // last MFMA -> accumulator reads below
    __builtin_amdgcn_sched_barrier(0);

    // ---- epilogue, straight from the MFMA layout: for row group j (rows j * 16 + fr) and column group u the lane
    //      holds acc[2 u][j] and acc[2 u + 1][j] = eight adjacent columns (W staging permutation above): bias, activation,
    //      bf16, residual, one 16-byte store; a store instruction covers 16 rows x 64 bytes.  With one wave per SIMD
    //      nothing else hides latency here, so this is straight-line code (the launcher guarantees M % 256 == 0,
    //      N % 8 == 0 and 16-byte alignment: the only guard left is the column mask of a ragged last N tile, loads are
    //      clamped instead of predicated); the residual is requested two row groups ahead, the bias a whole K loop
    //      ahead.  The next tile's first two slabs are in flight meanwhile; they are waited for BEFORE the first store
    //      is issued, so that the next slab's counted vmcnt never has to wait behind this tile's 32 stores per lane
    //      (vmcnt has no separate store counter).  Output / residual go through buffer descriptors of the tile (one
    //      lane-offset VGPR each, the row-group part in an SGPR): 64-bit per-access pointers would spill here.
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
    int rows_here = M - m0 < TM ? (int)(M - m0) : TM;  // ragged last row tile: the descriptors end at row M, so the
    rows_here = rows_here > 0 ? rows_here : 0;         // stores of the rows behind it are dropped and their loads read 0
    const __amdgpu_buffer_rsrc_t yrs =  // sized to the tile: masked lanes use an out-of-range offset (store dropped)
        __builtin_amdgcn_make_buffer_rsrc((void*)(Y + y_base + m0 * ldy + n0), 0, rows_here * (int)ldy * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((HAS_RES || DUAL) ? R + m0 * ldr + n0 : Y), 0, (HAS_RES || DUAL) ? rows_here * (int)ldr * 2 : 0,
        0x00020000);
    // row-sum partials [row][slot] of this tile's rows; only the lanes fq == 0 store (the others: out of range)
    const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(RS ? ln.rs_partial + m0 * ln.rs_slots : nullptr), 0, RS ? rows_here * ln.rs_slots * 8 : 0, 0x00020000);
    int fr_e = fr, fq_e = fq;  // opaque copies: keeps the epilogue's lane offsets from being hoisted above the K loop
    asm volatile("" : "+v"(fr_e), "+v"(fq_e));  // (a single VGPR spilled there costs a vmcnt(0) per reload here)
    const int ncol = wn * 128 + fq_e * 8;  // this lane's column inside the tile (+ 32 u)
    const int vps = fq_e == 0 ? (fr_e * ln.rs_slots + (n0 >> 7) + wn) * 8 : 0x7f000000;
    int vy[4], vr[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      int c = ncol + u * 32;
      vy[u] = n0 + c < N ? (fr_e * (int)ldy + c) * 2 : 0x7f000000;  // ragged last N tile: beyond the descriptor
      if constexpr (DUAL) {
        vr[u] = n0 + c < N ? (fr_e * (int)ldr + c) * 2 : 0x7f000000;  // second output: masked like the first
      } else {
        c = c < N - 8 - n0 ? c : N - 8 - n0;                        // loads: clamped instead
        vr[u] = (fr_e * (int)ldr + c) * 2;
      }
    }
    float bv[4][8];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (bias != nullptr) {
        VecIO<float, 8>::load(reinterpret_cast<const float*>(smem + 2 * BIG_STAGE) + ncol + u * 32, bv[u]);
      } else {
#pragma unroll
        for (int r = 0; r < 8; ++r) bv[u][r] = 0.f;
      }
    }
    float sv[4][8];  // LayerNorm fold: column sums of W' (LDS) and the row statistics { rstd, -mean rstd } of all 8 row groups
    float2 rst[MH];
    if constexpr (LN) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
        VecIO<float, 8>::load(reinterpret_cast<const float*>(smem + 2 * BIG_STAGE + 1024) + ncol + u * 32, sv[u]);
#pragma unroll
      for (int j = 0; j < MH; ++j)
        rst[j] = *reinterpret_cast<const float2*>(smem + 2 * BIG_STAGE + 2048 + (wm * (MH * 16) + j * 16 + fr_e) * 8);
    }
    auto res_fetch = [&](auto j_tag, uint4 (&rv)[4]) {
      constexpr int j = decltype(j_tag)::value;
      if constexpr (HAS_RES && j < MH) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const u32x4_t v =
              __builtin_amdgcn_raw_buffer_load_b128(rrs, vr[u], (wm * (MH * 16) + j * 16) * (int)ldr * 2, 0);
          rv[u] = make_uint4(v.x, v.y, v.z, v.w);
        }
      }
    };
    // Row group j's accumulators are "touched" by an empty volatile asm first: the reads below then cannot be hoisted
    // above it (left alone, the compiler reads all 256 accumulators into VGPRs right behind the K loop -- ahead of the
    // hazard fence -- shuffles the overflow through v_accvgpr_mov and was observed to deliver wrong values).  They are
    // zeroed for the next tile through the idle MFMA pipe (D = 0 * 0 + 0): an asm in program order, no v_accvgpr_write
    // rematerialised next to the first MFMA of the next tile.
    auto store_rows = [&](auto j_tag, const uint4 (&rv)[4]) {
      constexpr int j = decltype(j_tag)::value;
      f32x4_t c[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        asm volatile("" : "+a"(acc[i][j]));
        c[i] = acc[i][j];
        asm volatile("" : "+v"(c[i]));  // the copy is IN VGPRs here, before the accumulator's register is zeroed in place
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 8; ++i)  // "+a": the zeroed value keeps the accumulator's register (no copies at the loop edges)
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %1, 0" : "+a"(acc[i][j]) : "v"(zfrag));
      float rs1 = 0.f, rs2 = 0.f;  // RS: sum / sum of squares of this lane's 32 stored values of row j * 16 + fr
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const f32x4_t c0 = c[2 * u], c1 = c[2 * u + 1];
        f32x2_t p0 = f32x2_t{c0[0], c0[1]}, p1 = f32x2_t{c0[2], c0[3]}, p2 = f32x2_t{c1[0], c1[1]},
                p3 = f32x2_t{c1[2], c1[3]};
        if constexpr (LN) {  // rstd_m acc + (-mean_m rstd_m) s[n] + b'[n]
          const f32x2_t r2 = f32x2_t{rst[j].x, rst[j].x}, t2 = f32x2_t{rst[j].y, rst[j].y};
          p0 = p0 * r2 + (t2 * f32x2_t{sv[u][0], sv[u][1]} + f32x2_t{bv[u][0], bv[u][1]});
          p1 = p1 * r2 + (t2 * f32x2_t{sv[u][2], sv[u][3]} + f32x2_t{bv[u][2], bv[u][3]});
          p2 = p2 * r2 + (t2 * f32x2_t{sv[u][4], sv[u][5]} + f32x2_t{bv[u][4], bv[u][5]});
          p3 = p3 * r2 + (t2 * f32x2_t{sv[u][6], sv[u][7]} + f32x2_t{bv[u][6], bv[u][7]});
        } else {
          p0 += f32x2_t{bv[u][0], bv[u][1]};
          p1 += f32x2_t{bv[u][2], bv[u][3]};
          p2 += f32x2_t{bv[u][4], bv[u][5]};
          p3 += f32x2_t{bv[u][6], bv[u][7]};
        }
        if constexpr (DUAL) {  // the pre-activation, as the backward will read it
          __builtin_amdgcn_raw_buffer_store_b128(
              u32x4_t{pack_bf16x2(p0.x, p0.y), pack_bf16x2(p1.x, p1.y), pack_bf16x2(p2.x, p2.y), pack_bf16x2(p3.x, p3.y)},
              rrs, vr[u], (wm * (MH * 16) + j * 16) * (int)ldr * 2, 0);
          __builtin_amdgcn_sched_barrier(0);
          asm volatile("s_nop 1" ::: "memory");  // (the store-data hazard described below)
          __builtin_amdgcn_sched_barrier(0);
        }
        f32x2_t o0, o1, o2, o3;
        if constexpr (GMUL) {
          auto up2 = [](uint32_t w2) { return f32x2_t{__uint_as_float(w2 << 16), __uint_as_float(w2 & 0xffff0000u)}; };
          o0 = p0 * act_grad2<ACT>(up2(rv[u].x));
          o1 = p1 * act_grad2<ACT>(up2(rv[u].y));
          o2 = p2 * act_grad2<ACT>(up2(rv[u].z));
          o3 = p3 * act_grad2<ACT>(up2(rv[u].w));
        } else {
          o0 = act_apply2<ACT>(p0);
          o1 = act_apply2<ACT>(p1);
          o2 = act_apply2<ACT>(p2);
          o3 = act_apply2<ACT>(p3);
        }
        uint4 v = make_uint4(pack_bf16x2(o0.x, o0.y), pack_bf16x2(o1.x, o1.y), pack_bf16x2(o2.x, o2.y),
                             pack_bf16x2(o3.x, o3.y));
        if constexpr (HAS_RES && !GMUL)
          v = make_uint4(bf16x2_add(v.x, rv[u].x), bf16x2_add(v.y, rv[u].y), bf16x2_add(v.z, rv[u].z),
                         bf16x2_add(v.w, rv[u].w));
        if constexpr (RS) {  // on the ROUNDED values, pairwise: v_dot2_f32_bf16 with (1, 1) and with itself
          const uint32_t ones = 0x3f803f80u;
          const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            uint32_t wq = w4[q];
            rs1 = __builtin_amdgcn_fdot2_f32_bf16(*reinterpret_cast<const bf16x2_t*>(&wq),
                                                  *reinterpret_cast<const bf16x2_t*>(&ones), rs1, false);
            rs2 = __builtin_amdgcn_fdot2_f32_bf16(*reinterpret_cast<const bf16x2_t*>(&wq),
                                                  *reinterpret_cast<const bf16x2_t*>(&wq), rs2, false);
          }
        }
        __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{v.x, v.y, v.z, v.w}, yrs, vy[u],
                                               (wm * (MH * 16) + j * 16) * (int)ldy * 2, 0);
        // Observed on gfx950: a 16-byte buffer store whose data registers are overwritten by the very next VALU
        // instruction stores the new value in part of dword 1 (lanes 12..15 of every 16).  The compiler pads this hazard
        // with one wait state except when soffset is an SGPR (as here), where it assumes none: pad by hand.
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 1" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (RS) {  // the four lanes fr, fr + 16, fr + 32, fr + 48 hold the row's four 8-column groups per u
        rs1 += __shfl_xor(rs1, 16, 64);
        rs2 += __shfl_xor(rs2, 16, 64);
        rs1 += __shfl_xor(rs1, 32, 64);
        rs2 += __shfl_xor(rs2, 32, 64);
        typedef __attribute__((ext_vector_type(2))) unsigned u32x2s_t;
        __builtin_amdgcn_raw_buffer_store_b64(u32x2s_t{__float_as_uint(rs1), __float_as_uint(rs2)}, prs, vps,
                                              (wm * (MH * 16) + j * 16) * ln.rs_slots * 8, 0);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 1" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    // (sched_barrier: the compiler otherwise sinks the prefetches down to their first use)
#define ANEMOI_PIN() __builtin_amdgcn_sched_barrier(0)
    uint4 rv[3][4];
    res_fetch(std::integral_constant<int, 0>{}, rv[0]);
    res_fetch(std::integral_constant<int, 1>{}, rv[1]);
    ANEMOI_PIN();
    if constexpr (HAS_RES) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the next tile's slabs 0 / 1 have landed (and residual rows 0, 1)
    } else {
      // nothing of this epilogue comes from vector memory (bias, column sums and row statistics sit in LDS), so only the
      // next tile's slab 0 has to have landed before the stores queue up behind it; slab 1 (the newest NRD DMAs of this
      // wave) stays in flight under the epilogue and is waited for by slab 0's own counted wait, ~2 us later
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NRD) : "memory");
    }
    ANEMOI_PIN();
    static_for_seq(
        [&](auto j_tag) {
          constexpr int j = decltype(j_tag)::value;
          res_fetch(std::integral_constant<int, j + 2>{}, rv[(j + 2) % 3]);
          ANEMOI_PIN();
          store_rows(j_tag, rv[j % 3]);
          ANEMOI_PIN();
        },
        std::make_integer_sequence<int, MH>{});
#undef ANEMOI_PIN
    li += bpx;
    if (li >= chunk_len) break;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no LDS-DMA may outlive the workgroup
}

template <typename T, typename TO>
static int linear_launch(const void* x, int64_t ldx, const void* w, const float* bias, const void* residual,
                         int64_t ldr, void* y, int64_t ldy, int64_t M, int N, int K, int act, hipStream_t st,
                         LnFold ln, int batch = 1, int64_t sx = 0, int64_t sw = 0, int64_t sy = 0);

constexpr int W4_NEEDS_WHOLE_TILES = -4242;  // internal: the caller has to split the ragged rows off itself
// Returns through *tail_done whether the up to 8 rows behind M (m_tail) were computed by the same launch.
static int linear_bf16_256_launch(const void* x, int64_t ldx, const void* w, const float* bias, const void* residual,
                                  int64_t ldr, void* y, int64_t ldy, int64_t M, int N, int K, int act,
                                  hipStream_t st, LnFold ln = LnFold{nullptr, nullptr}, int m_tail = 0,
                                  bool* tail_done = nullptr, int epi = 0) {
  // epi 1 (DUAL): `residual` / ldr name the second OUTPUT (pre-activation); epi 2 (GMUL): they name the saved
  // pre-activation and the result is acc * act'(it).  Whole tiles only in both modes.
  const bool dual = epi == 1, gmul = epi == 2;
  if (tail_done != nullptr) *tail_done = false;
  static PerDeviceOnce raised;
  const int raise_dev = raised.pending();
  if (raise_dev >= 0) {
#define RAISE_W4_(A, RES, LNF)                                                                    \
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(linear_bf16_w4_kernel<A, RES, LNF, 8>),  \
                          hipFuncAttributeMaxDynamicSharedMemorySize, W4_LDS) != hipSuccess ||    \
      hipFuncSetAttribute(reinterpret_cast<const void*>(linear_bf16_w4_kernel<A, RES, LNF, 6>),  \
                          hipFuncAttributeMaxDynamicSharedMemorySize, W4_LDS) != hipSuccess ||    \
      hipFuncSetAttribute(reinterpret_cast<const void*>(linear_bf16_w4_kernel<A, RES, LNF, 5>),  \
                          hipFuncAttributeMaxDynamicSharedMemorySize, W4_LDS) != hipSuccess ||    \
      hipFuncSetAttribute(reinterpret_cast<const void*>(linear_bf16_w4_kernel<A, RES, LNF, 4>),  \
                          hipFuncAttributeMaxDynamicSharedMemorySize, W4_LDS) != hipSuccess ||    \
      hipFuncSetAttribute(reinterpret_cast<const void*>(linear_bf16_w4_kernel<A, RES, LNF, 3>),  \
                          hipFuncAttributeMaxDynamicSharedMemorySize, W4_LDS) != hipSuccess)      \
    return fail(ANEMOI_ERR_LAUNCH, "anemoi_linear: cannot raise the dynamic LDS limit to %d", W4_LDS)
#define RAISE_W4(A, RES) \
  RAISE_W4_(A, RES, false); \
  RAISE_W4_(A, RES, true)
#define RAISE_W4_RS(RES, LNF, MHV)                                                                              \
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(linear_bf16_w4_kernel<0, RES, LNF, MHV, true>),         \
                          hipFuncAttributeMaxDynamicSharedMemorySize, W4_LDS) != hipSuccess)                     \
    return fail(ANEMOI_ERR_LAUNCH, "anemoi_linear: cannot raise the dynamic LDS limit to %d", W4_LDS)
    RAISE_W4_RS(false, false, 8);
    RAISE_W4_RS(false, true, 8);
    RAISE_W4_RS(true, false, 8);
    RAISE_W4_RS(true, true, 8);
    RAISE_W4_RS(false, false, 6);
    RAISE_W4_RS(false, true, 6);
    RAISE_W4_RS(true, false, 6);
    RAISE_W4_RS(true, true, 6);
    RAISE_W4_RS(false, false, 5);
    RAISE_W4_RS(false, true, 5);
    RAISE_W4_RS(true, false, 5);
    RAISE_W4_RS(true, true, 5);
    RAISE_W4_RS(false, false, 4);
    RAISE_W4_RS(false, true, 4);
    RAISE_W4_RS(true, false, 4);
    RAISE_W4_RS(true, true, 4);
    RAISE_W4_RS(false, false, 3);
    RAISE_W4_RS(false, true, 3);
    RAISE_W4_RS(true, false, 3);
    RAISE_W4_RS(true, true, 3);
#undef RAISE_W4_RS
#define RAISE_W4_DUAL(A, MHV)                                                                                    \
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(linear_bf16_w4_kernel<A, false, false, MHV, false, true>), \
                          hipFuncAttributeMaxDynamicSharedMemorySize, W4_LDS) != hipSuccess)                      \
    return fail(ANEMOI_ERR_LAUNCH, "anemoi_linear: cannot raise the dynamic LDS limit to %d", W4_LDS)
    RAISE_W4_DUAL(1, 8);
    RAISE_W4_DUAL(1, 4);
    RAISE_W4_DUAL(2, 8);
    RAISE_W4_DUAL(2, 4);
    RAISE_W4_DUAL(3, 8);
    RAISE_W4_DUAL(3, 4);
#undef RAISE_W4_DUAL
#define RAISE_W4_GMUL(A, MHV)                                                                                          \
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(linear_bf16_w4_kernel<A, true, false, MHV, false, false, true>), \
                          hipFuncAttributeMaxDynamicSharedMemorySize, W4_LDS) != hipSuccess)                            \
    return fail(ANEMOI_ERR_LAUNCH, "anemoi_linear: cannot raise the dynamic LDS limit to %d", W4_LDS)
    RAISE_W4_GMUL(1, 8);
    RAISE_W4_GMUL(1, 4);
    RAISE_W4_GMUL(2, 8);
    RAISE_W4_GMUL(2, 4);
    RAISE_W4_GMUL(3, 8);
    RAISE_W4_GMUL(3, 4);
#undef RAISE_W4_GMUL
    RAISE_W4(0, false);
    RAISE_W4(0, true);
    RAISE_W4(1, false);
    RAISE_W4(1, true);
    RAISE_W4(2, false);
    RAISE_W4(2, true);
    RAISE_W4(3, false);
    RAISE_W4(3, true);
#undef RAISE_W4
#undef RAISE_W4_
    raised.done(raise_dev);
  }
  const int64_t mt = (M + BIG_M - 1) / BIG_M;
  const int64_t nt = (N + BIG_N - 1) / BIG_N;
  ANEMOI_REQUIRE(mt * nt < (int64_t)1 << 31, ANEMOI_ERR_UNSUPPORTED, "anemoi_linear: grid too large");
  const bool vec_ok = (N % 8 == 0) && (ldy % 8 == 0) && ((uintptr_t)y % 16 == 0) &&
                      (bias == nullptr || (uintptr_t)bias % 16 == 0) && (residual == nullptr || (ldr % 8 == 0 && (uintptr_t)residual % 16 == 0));
  const bool batched = ln.b_tiles > 0;  // anemoi_linear_batched: b_count problems of mt * nt tiles each
  const int64_t problems = batched ? ln.b_count : 1;
  int64_t blocks = problems * mt * nt;
  constexpr int64_t max_blocks = 256;  // one persistent workgroup per CU
  if (blocks > max_blocks) blocks = max_blocks;
  blocks = (blocks + 7) / 8 * 8;          // whole XCD rows; surplus workgroups exit at once
  if (K >= 128 && ldx < (int64_t)1 << 21 && ldy < (int64_t)1 << 21 && ldr < (int64_t)1 << 21 && vec_ok &&
      (M % BIG_M == 0 || m_tail == 0) && (ln.colsum == nullptr || (uintptr_t)ln.colsum % 16 == 0)) {
    const int w4_tail = (m_tail > 0 && m_tail <= 8 && K % 8 == 0) ? m_tail : 0;
    if (tail_done != nullptr) *tail_done = w4_tail > 0;
    // Remainder round as HALF tiles: when the tiles beyond the last whole round of 256 would keep at most half of the
    // CUs busy (N = 1024 at M = 40 960: 640 tiles = 2.5 rounds), their rows go to a second launch with 128 x 256 tiles
    // (MH = 4): twice as many units of half the work -- 2.5+ instead of 3 rounds, no cross-workgroup traffic.
    int64_t mt_a = mt, mt_b = 0;  // row tiles of 256 for launch A; rows of launch B = mt_b * 256 as half tiles
    if (!batched && nt <= 256 && 256 % nt == 0) {
      const int64_t per_round = 256 / nt;
      const int64_t rem_mt = mt % per_round;
      // measured: a half tile costs ~0.75 of a whole one (its 48 KiB slab per 64 MFMAs is bound by the L2 -> LDS
      // staging rate), and a second launch ~8 us; so split only a remainder that is the whole problem (small M of a
      // node-partitioned run: 5120 x 1024 x 4096 0.081 -> 0.052 ms) or belongs to long tiles (K >= 2048: 40 962 x 1024
      // x 4096 0.314 -> 0.295 ms; at K = 1216 the second launch costs what the half tiles save)
      // (round 2: splitting EVERY remainder of <= 128 tiles off, whatever K, measured -1 ... +9 % on the per-rank shapes
      //  5121 x {4096, 2048, 2240, 1024} -- profiles/r02_gemm_small_m.txt -- so the rule stays)
      // (the whole-problem case, rem_mt == mt, is one candidate of the tile-height model below for the plain epilogues)
      if (rem_mt > 0 && rem_mt * nt <= 128 && ((rem_mt == mt && (dual || gmul)) || (rem_mt != mt && K >= 2048))) {
        mt_a = mt - rem_mt;
        mt_b = rem_mt;
      }
    }
#define LAUNCH_W4__(A, RES, LNF, MHV, RSV, XP, RP, YP, LNV, MV, TILES, TAIL)                                       \
  hipLaunchKernelGGL((linear_bf16_w4_kernel<A, RES, LNF, MHV, RSV>), dim3((unsigned)w4_blocks), dim3(256), W4_LDS, \
                     st, XP, ldx, static_cast<const bf16_t*>(w), bias, RP, ldr, YP, ldy, MV, N, K,                 \
                     vec_ok ? 1 : 0, TILES, (int)nt, LNV, TAIL)
#define LAUNCH_W4_(A, RES, MHV, RSV, XP, RP, YP, LNV, MV, TILES, TAIL)                             \
  do {                                                                                             \
    if (ln.stats != nullptr) LAUNCH_W4__(A, RES, true, MHV, RSV, XP, RP, YP, LNV, MV, TILES, TAIL); \
    else LAUNCH_W4__(A, RES, false, MHV, RSV, XP, RP, YP, LNV, MV, TILES, TAIL);                   \
  } while (0)
#define LAUNCH_W4(A, MHV, RSV, XP, RP, YP, LNV, MV, TILES, TAIL)                                \
  do {                                                                                          \
    if (residual != nullptr) LAUNCH_W4_(A, true, MHV, RSV, XP, RP, YP, LNV, MV, TILES, TAIL);   \
    else LAUNCH_W4_(A, false, MHV, RSV, XP, RP, YP, LNV, MV, TILES, TAIL);                      \
  } while (0)
#define LAUNCH_W4_DUAL(A, MHV, XP, RP, YP, LNV, MV, TILES, TAIL)                                                      \
  hipLaunchKernelGGL((linear_bf16_w4_kernel<A, false, false, MHV, false, true>), dim3((unsigned)w4_blocks), dim3(256), \
                     W4_LDS, st, XP, ldx, static_cast<const bf16_t*>(w), bias, RP, ldr, YP, ldy, MV, N, K,             \
                     vec_ok ? 1 : 0, TILES, (int)nt, LNV, TAIL)
#define LAUNCH_W4_GMUL(A, MHV, XP, RP, YP, LNV, MV, TILES, TAIL)                                                            \
  hipLaunchKernelGGL((linear_bf16_w4_kernel<A, true, false, MHV, false, false, true>), dim3((unsigned)w4_blocks), dim3(256), \
                     W4_LDS, st, XP, ldx, static_cast<const bf16_t*>(w), bias, RP, ldr, YP, ldy, MV, N, K,                   \
                     vec_ok ? 1 : 0, TILES, (int)nt, LNV, TAIL)
#define LAUNCH_W4_PLAIN(MHV, XP, RP, YP, LNV, MV, TILES, TAIL)                                      \
  switch (act) {                                                                                    \
    case ANEMOI_ACT_GELU: LAUNCH_W4(ANEMOI_ACT_GELU, MHV, false, XP, RP, YP, LNV, MV, TILES, TAIL); break; \
    case ANEMOI_ACT_SILU: LAUNCH_W4(ANEMOI_ACT_SILU, MHV, false, XP, RP, YP, LNV, MV, TILES, TAIL); break; \
    case ANEMOI_ACT_RELU: LAUNCH_W4(ANEMOI_ACT_RELU, MHV, false, XP, RP, YP, LNV, MV, TILES, TAIL); break; \
    default:                                                                                        \
      if (rs_on) LAUNCH_W4(ANEMOI_ACT_NONE, MHV, true, XP, RP, YP, LNV, MV, TILES, TAIL);           \
      else LAUNCH_W4(ANEMOI_ACT_NONE, MHV, false, XP, RP, YP, LNV, MV, TILES, TAIL);                \
      break;                                                                                        \
  }
#define LAUNCH_W4_ACT(MHV, XP, RP, YP, LNV, MV, TILES, TAIL)                                        \
  if (gmul) {                                                                                       \
    switch (act) {                                                                                  \
      case ANEMOI_ACT_GELU: LAUNCH_W4_GMUL(ANEMOI_ACT_GELU, MHV, XP, RP, YP, LNV, MV, TILES, TAIL); break; \
      case ANEMOI_ACT_SILU: LAUNCH_W4_GMUL(ANEMOI_ACT_SILU, MHV, XP, RP, YP, LNV, MV, TILES, TAIL); break; \
      default: LAUNCH_W4_GMUL(ANEMOI_ACT_RELU, MHV, XP, RP, YP, LNV, MV, TILES, TAIL); break;         \
    }                                                                                               \
  } else if (dual) {                                                                                \
    switch (act) {                                                                                  \
      case ANEMOI_ACT_GELU: LAUNCH_W4_DUAL(ANEMOI_ACT_GELU, MHV, XP, RP, YP, LNV, MV, TILES, TAIL); break; \
      case ANEMOI_ACT_SILU: LAUNCH_W4_DUAL(ANEMOI_ACT_SILU, MHV, XP, RP, YP, LNV, MV, TILES, TAIL); break; \
      default: LAUNCH_W4_DUAL(ANEMOI_ACT_RELU, MHV, XP, RP, YP, LNV, MV, TILES, TAIL); break;         \
    }                                                                                               \
  } else LAUNCH_W4_PLAIN(MHV, XP, RP, YP, LNV, MV, TILES, TAIL)
    // row-sum partials (LnFold::rs_partial): plain epilogue only, whole 256-column tiles only
    const bool rs_on = ln.rs_partial != nullptr && act == ANEMOI_ACT_NONE && N % BIG_N == 0;
    if (rs_on && ln.rs_rows_done != nullptr) *ln.rs_rows_done = M;
    const bf16_t* xb = static_cast<const bf16_t*>(x);
    const bf16_t* rb = static_cast<const bf16_t*>(residual);
    bf16_t* yb = static_cast<bf16_t*>(y);
    int64_t w4_blocks = blocks;
    // Small problems (fewer than four rounds of 256-row tiles -- the per-rank shapes of a node-partitioned run, BASELINE
    // configs 2 and 5): the tile HEIGHT is chosen per launch among 256 / 192 / 160 / 128 rows by a two-term model of the
    // persistent loop, rounds x (epilogue + slabs x staged bytes): the loop is bound by the bytes it stages, so a tile of
    // 32 MH rows costs (32 MH + 256) / 512 of a whole one per slab (measured 0.75 for MH = 4, 0.875 for MH = 6) and
    // MH / 8 of its 6.5 us epilogue; what decides is how the tile count quantises against the 256 CUs
    // (5121 x 2048: 160 / 216 / 256 / 320 tiles -> 160-row tiles fill the chip exactly once;
    //  5121 x 4096: 320 / 432 / 512 / 640 -> two balanced rounds of 160-row tiles; profiles/r04_gemm_small_m.txt).
    // (a half-tile split chosen above competes with the single-launch candidates: its cost by the same model = whole rounds of
    //  256-row tiles + the rounds of half tiles at 0.75 of the slab cost + ~6 us for the second launch.  20 481 x 1024 x 4096,
    //  one rank of two: 256 + 128 half tiles = 177 against 158 for two rounds of 160-row tiles)
    // Staggered start (LnFold::stagger_*); ANEMOI_AMD_GEMM_STAGGER="phases,unit,min rounds" overrides the shipped values
    // (phases 0: off; A/B runs)
    static const struct Stagger { int phases, unit, min_rounds; } stagger = [] {
      Stagger v{STAGGER_PHASES, STAGGER_UNIT, STAGGER_MIN_ROUNDS};
      if (const char* e = getenv("ANEMOI_AMD_GEMM_STAGGER")) sscanf(e, "%d,%d,%d", &v.phases, &v.unit, &v.min_rounds);
      return v;
    }();
    const bool split_chosen = mt_b > 0 && !dual && !gmul;
    if (!batched && !dual && !gmul && (mt_b == 0 || split_chosen) && mt * nt < 4 * max_blocks) {
      static const int forced_mh = [] {  // lab switch for A/B runs: ANEMOI_AMD_GEMM_MH=3|4|5|6|8
        const char* e = getenv("ANEMOI_AMD_GEMM_MH");
        return e != nullptr ? atoi(e) : 0;
      }();
      int best_mh = 8;
      double best_cost = 1e30;
      for (const int mh : {8, 6, 5, 4, 3}) {
        const int64_t tiles = (M + 32 * mh - 1) / (32 * mh) * nt, rounds = (tiles + max_blocks - 1) / max_blocks;
        const double cost = (double)rounds * (0.8125 * mh + (K / 64) * 1.44 * (32 * mh + 256) / 512.0);
        if (forced_mh == mh || (forced_mh == 0 && cost < best_cost * 0.97)) {  // (3 %: ties go to the taller tile)
          best_cost = cost;
          best_mh = mh;
          if (forced_mh == mh) break;
        }
      }
      bool single = best_mh != 8;
      if (split_chosen) {
        const int64_t rounds_a = (mt_a * nt + max_blocks - 1) / max_blocks, rounds_b = (mt_b * 2 * nt + max_blocks - 1) / max_blocks;
        const double split_cost = (double)rounds_a * (0.8125 * 8 + (K / 64) * 1.44) +
                                  (double)rounds_b * (0.8125 * 4 + (K / 64) * 1.44 * 0.75) + 6.0;
        single = forced_mh != 0 || best_cost < split_cost * 0.97;  // (a forced height means ONE launch of that height)
        if (single) {  // one launch of best_mh-row tiles (8 included) instead of the split
          mt_a = mt;
          mt_b = 0;
        }
      }
      if (single && best_mh != 8) {
        const int64_t tiles = (M + 32 * best_mh - 1) / (32 * best_mh) * nt;
        w4_blocks = tiles < max_blocks ? (tiles + 7) / 8 * 8 : max_blocks;
        // (no stagger on these launches of one or two rounds: 1 ... 4 units measured 3.20 -> 3.20 ... 3.26 ms at config 2)
        if (best_mh == 6) { LAUNCH_W4_PLAIN(6, xb, rb, yb, ln, M, tiles, w4_tail) }
        else if (best_mh == 5) { LAUNCH_W4_PLAIN(5, xb, rb, yb, ln, M, tiles, w4_tail) }
        else if (best_mh == 3) { LAUNCH_W4_PLAIN(3, xb, rb, yb, ln, M, tiles, w4_tail) }
        else { LAUNCH_W4_PLAIN(4, xb, rb, yb, ln, M, tiles, w4_tail) }
        mt_a = 0;  // done
      }
    }
    if (mt_a > 0) {
      const int64_t m_a = mt_a * BIG_M < M ? mt_a * BIG_M : M, tiles_a = problems * mt_a * nt;
      const int tail_a = mt_b == 0 ? w4_tail : 0;
      w4_blocks = tiles_a < max_blocks ? (tiles_a + 7) / 8 * 8 : max_blocks;
      // Staggered start (LnFold::stagger_*): launches of >= min_rounds rounds of tiles with K >= STAGGER_MIN_K (the K = 256
      // products measured no gain).  Phase g waits g x unit x 1024 cycles; a SMALL offset is what counts -- 2 x 8 ... 32 units,
      // 3 x 12, 4 x 8 measured alike, a step of half a tile's time or 16 phases nothing.
      LnFold la = ln;
      if (stagger.phases > 1 && tiles_a >= (int64_t)stagger.min_rounds * max_blocks && K >= STAGGER_MIN_K) {
        la.stagger_phases = stagger.phases;
        la.stagger_unit = stagger.unit;
      }
      LAUNCH_W4_ACT(8, xb, rb, yb, la, m_a, tiles_a, tail_a)
    }
    if (mt_b > 0) {
      const int64_t m_a = mt_a * BIG_M, m_b = M - m_a, tiles_b = mt_b * 2 * nt;  // m_b may end inside the last tile
      LnFold lb = ln;
      if (lb.stats != nullptr) lb.stats += m_a;
      if (lb.rs_partial != nullptr) lb.rs_partial += m_a * lb.rs_slots;
      w4_blocks = tiles_b < max_blocks ? (tiles_b + 7) / 8 * 8 : max_blocks;
      LAUNCH_W4_ACT(4, xb + m_a * ldx, rb != nullptr ? rb + m_a * ldr : nullptr, yb + m_a * ldy, lb, m_b, tiles_b, w4_tail)
    }
#undef LAUNCH_W4_ACT
#undef LAUNCH_W4_PLAIN
#undef LAUNCH_W4_DUAL
#undef LAUNCH_W4_GMUL
#undef LAUNCH_W4
#undef LAUNCH_W4_
#undef LAUNCH_W4__
    return check_launch("anemoi_linear(256x256, 4 waves)");
  }
  if (M % BIG_M != 0 || batched || epi != 0) return W4_NEEDS_WHOLE_TILES;  // (the caller splits the ragged rows off / refuses)
  // shapes the persistent kernel does not take (K = 64, unaligned output, ...): the general 128 x 128 kernel
  return linear_launch<bf16_t, bf16_t>(x, ldx, w, bias, residual, ldr, y, ldy, M, N, K, act, st, ln);
}

// =============================================================================================
// Skinny rows (M <= 8): one wave per output column, K swept with 16-byte loads, f32 accumulation, wave reduction.
// Used for the ragged last rows of a tall GEMM (M % 256 in 1..8) so that they do not cost a full tile round.
// =============================================================================================
template <int ROWS>
__global__ __launch_bounds__(256) void linear_bf16_skinny_kernel(const bf16_t* __restrict__ X, int64_t ldx,
                                                                 const bf16_t* __restrict__ W,
                                                                 const float* __restrict__ bias,
                                                                 const bf16_t* __restrict__ R, int64_t ldr,
                                                                 bf16_t* __restrict__ Y, int64_t ldy, int M, int N,
                                                                 int K, int act, LnFold ln) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  skinny_column<ROWS>(X, ldx, W, bias, R, ldr, Y, ldy, M, n, K, act, ln, lane);
}

template <typename T, typename TO>
static int linear_launch(const void* x, int64_t ldx, const void* w, const float* bias, const void* residual,
                         int64_t ldr, void* y, int64_t ldy, int64_t M, int N, int K, int act, hipStream_t st,
                         LnFold ln, int batch, int64_t sx, int64_t sw, int64_t sy) {
  const int64_t mt = (M + BM - 1) / BM;
  const int64_t nt = (N + BN - 1) / BN;
  ANEMOI_REQUIRE(mt * nt < (int64_t)1 << 31, ANEMOI_ERR_UNSUPPORTED, "anemoi_linear: grid too large");
  const bool vec_ok = (N % 4 == 0) && (ldy % 4 == 0) && ((uintptr_t)y % 16 == 0) &&
                      (bias == nullptr || (uintptr_t)bias % 16 == 0) &&
                      (ln.colsum == nullptr || (uintptr_t)ln.colsum % 16 == 0) &&
                      (residual == nullptr || (ldr % 4 == 0 && (uintptr_t)residual % 16 == 0));
  hipLaunchKernelGGL((linear_kernel<T, TO>), dim3((unsigned)(mt * nt), (unsigned)batch), dim3(256), 0, st,
                     static_cast<const T*>(x), ldx, static_cast<const T*>(w), bias, static_cast<const T*>(residual), ldr,
                     static_cast<TO*>(y), ldy, M, N, K, act, vec_ok ? 1 : 0, ln, sx, sw, sy);
  return check_launch("anemoi_linear");
}

}  // namespace anemoi

using namespace anemoi;

static int linear_dispatch(const char* who, int dtype, int out_dtype, const void* x, int64_t ldx, const void* w,
                           const float* bias, LnFold ln, const void* residual, int64_t ldr, void* y, int64_t ldy,
                           int64_t M, int N, int K, int act, anemoi_stream_t stream) {
  ANEMOI_REQUIRE(x && w && y, ANEMOI_ERR_INVALID, "%s: null pointer", who);
  ANEMOI_REQUIRE(M >= 0 && N > 0 && K > 0, ANEMOI_ERR_INVALID, "%s: bad shape M=%lld N=%d K=%d", who, (long long)M, N, K);
  ANEMOI_REQUIRE(ldx >= K && ldy >= N && (residual == nullptr || ldr >= N), ANEMOI_ERR_INVALID,
                 "%s: leading dimension too small", who);
  ANEMOI_REQUIRE(act >= ANEMOI_ACT_NONE && act <= ANEMOI_ACT_RELU, ANEMOI_ERR_INVALID, "%s: act %d", who, act);
  const int esz = dtype == ANEMOI_BF16 ? 2 : 4;
  ANEMOI_REQUIRE(((int64_t)K * esz) % ROW_BYTES == 0, ANEMOI_ERR_INVALID,
                 "%s: K=%d must be a multiple of %d for this dtype (pad with zeros)", who, K, ROW_BYTES / esz);
  ANEMOI_REQUIRE((uintptr_t)x % 16 == 0 && (uintptr_t)w % 16 == 0 && (ldx * esz) % 16 == 0, ANEMOI_ERR_INVALID,
                 "%s: x / W must be 16-byte aligned with a 16-byte multiple row pitch", who);
  if (M == 0) return ANEMOI_OK;
  hipStream_t st = as_stream(stream);
  if (dtype == ANEMOI_F32 && out_dtype == ANEMOI_F32)
    return linear_launch<float, float>(x, ldx, w, bias, residual, ldr, y, ldy, M, N, K, act, st, ln);
  constexpr int64_t big_min_m = 1024;  // rows from which the persistent 256 x 256 kernel takes over
  if (dtype == ANEMOI_BF16 && out_dtype == ANEMOI_BF16 && M >= big_min_m && N >= 256) {
    // A ragged last row tile would cost every CU of one XCD a full extra round (161 vs 160 row tiles at M = 40 962:
    // +10 %): the 256-row multiple goes to the persistent kernel, the few remaining rows to the 128 x 128 kernel.
    const int64_t m_main = M / BIG_M * BIG_M, m_tail = M - m_main;
    if (m_tail == 0) return linear_bf16_256_launch(x, ldx, w, bias, residual, ldr, y, ldy, M, N, K, act, st, ln);
    bool tail_done = false;
    // tails of more than 8 rows ride along as a ragged last row tile of the persistent kernel (its descriptors end at
    // row M): the separate 128 x 128 launch they used to get ran ~35 us on a handful of CUs, 14 times per forward
    if (m_tail > 8 && K >= 128) {
      const int rr = linear_bf16_256_launch(x, ldx, w, bias, residual, ldr, y, ldy, M, N, K, act, st, ln);
      if (rr != W4_NEEDS_WHOLE_TILES) return rr;
    }
    const int rc = linear_bf16_256_launch(x, ldx, w, bias, residual, ldr, y, ldy, m_main, N, K, act, st, ln,
                                          m_tail <= 8 ? (int)m_tail : 0, &tail_done);
    if (rc != ANEMOI_OK || tail_done) return rc;
    const bf16_t* xt = static_cast<const bf16_t*>(x) + m_main * ldx;
    const bf16_t* rt = residual ? static_cast<const bf16_t*>(residual) + m_main * ldr : nullptr;
    bf16_t* yt = static_cast<bf16_t*>(y) + m_main * ldy;
    LnFold lt = ln;
    if (lt.stats != nullptr) lt.stats += m_main;
    if (m_tail <= 8 && K % 8 == 0) {
      hipLaunchKernelGGL((linear_bf16_skinny_kernel<8>), dim3((unsigned)((N + 3) / 4)), dim3(256), 0, st, xt, ldx,
                         static_cast<const bf16_t*>(w), bias, rt, ldr, yt, ldy, (int)m_tail, N, K, act, lt);
      return check_launch("anemoi_linear(skinny tail)");
    }
    return linear_launch<bf16_t, bf16_t>(xt, ldx, w, bias, rt, ldr, yt, ldy, m_tail, N, K, act, st, lt);
  }
  if (dtype == ANEMOI_BF16 && out_dtype == ANEMOI_BF16)
    return linear_launch<bf16_t, bf16_t>(x, ldx, w, bias, residual, ldr, y, ldy, M, N, K, act, st, ln);
  if (dtype == ANEMOI_BF16 && out_dtype == ANEMOI_F32)
    return linear_launch<bf16_t, float>(x, ldx, w, bias, residual, ldr, y, ldy, M, N, K, act, st, ln);
  return fail(ANEMOI_ERR_UNSUPPORTED, "%s: dtype %d -> %d", who, dtype, out_dtype);
}

extern "C" int anemoi_linear(int dtype, int out_dtype, const void* x, int64_t ldx, const void* w, const float* bias,
                             const void* residual, int64_t ldr, void* y, int64_t ldy, int64_t M, int N, int K, int act,
                             anemoi_stream_t stream) {
  return linear_dispatch("anemoi_linear", dtype, out_dtype, x, ldx, w, bias, LnFold{nullptr, nullptr}, residual, ldr, y,
                         ldy, M, N, K, act, stream);
}

extern "C" int anemoi_linear_ln(int dtype, int out_dtype, const void* x, int64_t ldx, const void* w, const float* bias,
                                const float* colsum, const float* stats, const void* residual, int64_t ldr, void* y,
                                int64_t ldy, int64_t M, int N, int K, int act, anemoi_stream_t stream) {
  ANEMOI_REQUIRE(colsum && stats, ANEMOI_ERR_INVALID, "anemoi_linear_ln: null colsum / stats");
  ANEMOI_REQUIRE((uintptr_t)stats % 8 == 0, ANEMOI_ERR_INVALID, "anemoi_linear_ln: stats must be 8-byte aligned");
  return linear_dispatch("anemoi_linear_ln", dtype, out_dtype, x, ldx, w, bias,
                         LnFold{colsum, reinterpret_cast<const float2*>(stats)}, residual, ldr, y, ldy, M, N, K, act,
                         stream);
}

// ---------------------------------------------------------------------------------------------
// y = x W^T + b (+ residual)  AND  the LayerNorm statistics of y's rows, without a second pass over y: the four-wave
// kernel's epilogue leaves { sum, sum of squares } of the bf16 values it stores per (row, 128-column slot) in a
// workspace; this tiny kernel folds the slots into { rstd, -mean rstd } (the format of anemoi_row_stats).
// ---------------------------------------------------------------------------------------------
namespace anemoi {
__global__ __launch_bounds__(256) void row_sums_finalize_kernel(const float2* __restrict__ partial, int slots,
                                                                int64_t rows, int C, float eps,
                                                                float2* __restrict__ stats,
                                                                const bf16_t* __restrict__ y, int64_t ldy,
                                                                int64_t rows_total) {
  const int64_t fold_blocks = (rows + 255) / 256;
  const float inv_c = 1.0f / (float)C;
  if ((int64_t)blockIdx.x >= fold_blocks) {
    // one of the few rows behind the tiled part (computed by the skinny pass, no partials): this block reads the row
    const int64_t r = rows + ((int64_t)blockIdx.x - fold_blocks);
    if (r >= rows_total) return;
    float s = 0.f, ss = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) {
      const float v = bf16_to_f32(y[r * ldy + c]);
      s += v;
      ss = fmaf(v, v, ss);
    }
    s = wave_sum(s);
    ss = wave_sum(ss);
    __shared__ float red[8];
    if ((threadIdx.x & 63) == 0) {
      red[threadIdx.x >> 6] = s;
      red[4 + (threadIdx.x >> 6)] = ss;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      s = red[0] + red[1] + red[2] + red[3];
      ss = red[4] + red[5] + red[6] + red[7];
      const float mean = s * inv_c;
      const float rstd = rsqrtf(fmaxf(ss * inv_c - mean * mean, 0.f) + eps);
      stats[r] = make_float2(rstd, -mean * rstd);
    }
    return;
  }
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= rows) return;
  float s = 0.f, ss = 0.f;
  for (int k = 0; k < slots; ++k) {
    const float2 p = partial[r * slots + k];
    s += p.x;
    ss += p.y;
  }
  float mean = s * inv_c;
  float var = fmaxf(ss * inv_c - mean * mean, 0.f);
  if (var < 1e-3f * ss * inv_c) {
    // E[y^2] - mean^2 has cancelled more than three digits (|mean| >> spread: does not happen on the residual stream,
    // but the entry point is generic): this row is redone from y itself with the two-pass formula of anemoi_row_stats
    const bf16_t* yr = y + r * ldy;
    float s1 = 0.f;
    for (int c = 0; c < C; ++c) s1 += bf16_to_f32(yr[c]);
    mean = s1 * inv_c;
    float s2 = 0.f;
    for (int c = 0; c < C; ++c) {
      const float d = bf16_to_f32(yr[c]) - mean;
      s2 = fmaf(d, d, s2);
    }
    var = s2 * inv_c;
  }
  const float rstd = rsqrtf(var + eps);
  stats[r] = make_float2(rstd, -mean * rstd);
}
}  // namespace anemoi

// y = act(x W^T + b) AND pre = x W^T + b (bf16) from one launch of the persistent kernel (training forward: the backward
// needs act'(pre)).  bf16, whole 256-row tiles, the fast path's shape / alignment rules; ANEMOI_ERR_UNSUPPORTED otherwise
// (the caller then runs anemoi_linear + anemoi_act_forward).
extern "C" int anemoi_linear_dual(int dtype, const void* x, int64_t ldx, const void* w, const float* bias, void* pre,
                                  int64_t ldp, void* y, int64_t ldy, int64_t M, int N, int K, int act,
                                  anemoi_stream_t stream) {
  using namespace anemoi;
  ANEMOI_REQUIRE(x && w && pre && y, ANEMOI_ERR_INVALID, "anemoi_linear_dual: null pointer");
  ANEMOI_REQUIRE(M >= 0 && N > 0 && K > 0 && ldx >= K && ldy >= N && ldp >= N, ANEMOI_ERR_INVALID,
                 "anemoi_linear_dual: bad shape");
  ANEMOI_REQUIRE(act > ANEMOI_ACT_NONE && act <= ANEMOI_ACT_RELU, ANEMOI_ERR_INVALID, "anemoi_linear_dual: act %d", act);
  // (up to 8 rows behind the last whole 256-row tile are computed by the same launch: the skinny pass of the kernel)
  ANEMOI_REQUIRE(dtype == ANEMOI_BF16 && M % BIG_M <= 8 && (M >= BIG_M || M == 0) && N >= 256 && N % 8 == 0 && K >= 128 &&
                     K % 64 == 0 && (uintptr_t)x % 16 == 0 && (uintptr_t)w % 16 == 0 && (uintptr_t)pre % 16 == 0 &&
                     (uintptr_t)y % 16 == 0 && ldx % 8 == 0 && ldp % 8 == 0 && ldy % 8 == 0,
                 ANEMOI_ERR_UNSUPPORTED,
                 "anemoi_linear_dual: bf16, M = a multiple of 256 (+ at most 8 rows), N >= 256, K >= 128, 16-byte aligned");
  if (M == 0) return ANEMOI_OK;
  bool tail_done = false;
  const int rc = linear_bf16_256_launch(x, ldx, w, bias, pre, ldp, y, ldy, M / BIG_M * BIG_M, N, K, act, as_stream(stream),
                                        LnFold{nullptr, nullptr}, (int)(M % BIG_M), &tail_done, 1);
  if (rc == ANEMOI_OK && M % BIG_M != 0 && !tail_done)
    return fail(ANEMOI_ERR_UNSUPPORTED, "anemoi_linear_dual: the ragged rows were not taken by the fast path");
  if (rc == W4_NEEDS_WHOLE_TILES) return fail(ANEMOI_ERR_UNSUPPORTED, "anemoi_linear_dual: shape not taken by the fast path");
  return rc;
}

// y = (x W^T) * act'(pre): the backward's dX GEMM of the Linear behind an activation, delivering the gradient of the
// pre-activation of the Linear in front of it (x = dy [M, K], W = W2^T [N, K], pre [M, N]).  Same shape rules as
// anemoi_linear_dual; ANEMOI_ERR_UNSUPPORTED otherwise (the caller then runs anemoi_linear + anemoi_act_backward).
extern "C" int anemoi_linear_actgrad(int dtype, const void* x, int64_t ldx, const void* w, const void* pre, int64_t ldp,
                                     void* y, int64_t ldy, int64_t M, int N, int K, int act, anemoi_stream_t stream) {
  using namespace anemoi;
  ANEMOI_REQUIRE(x && w && pre && y, ANEMOI_ERR_INVALID, "anemoi_linear_actgrad: null pointer");
  ANEMOI_REQUIRE(M >= 0 && N > 0 && K > 0 && ldx >= K && ldy >= N && ldp >= N, ANEMOI_ERR_INVALID,
                 "anemoi_linear_actgrad: bad shape");
  ANEMOI_REQUIRE(act > ANEMOI_ACT_NONE && act <= ANEMOI_ACT_RELU, ANEMOI_ERR_INVALID, "anemoi_linear_actgrad: act %d", act);
  ANEMOI_REQUIRE(dtype == ANEMOI_BF16 && M % BIG_M <= 8 && (M >= BIG_M || M == 0) && N >= 256 && N % 8 == 0 && K >= 128 &&
                     K % 64 == 0 && (uintptr_t)x % 16 == 0 && (uintptr_t)w % 16 == 0 && (uintptr_t)pre % 16 == 0 &&
                     (uintptr_t)y % 16 == 0 && ldx % 8 == 0 && ldp % 8 == 0 && ldy % 8 == 0,
                 ANEMOI_ERR_UNSUPPORTED,
                 "anemoi_linear_actgrad: bf16, M = a multiple of 256 (+ at most 8 rows), N >= 256, K >= 128, 16-byte aligned");
  if (M == 0) return ANEMOI_OK;
  bool tail_done = false;
  const int rc = linear_bf16_256_launch(x, ldx, w, nullptr, pre, ldp, y, ldy, M / BIG_M * BIG_M, N, K, act, as_stream(stream),
                                        LnFold{nullptr, nullptr}, (int)(M % BIG_M), &tail_done, 2);
  if (rc == ANEMOI_OK && M % BIG_M != 0 && !tail_done)
    return fail(ANEMOI_ERR_UNSUPPORTED, "anemoi_linear_actgrad: the ragged rows were not taken by the fast path");
  if (rc == W4_NEEDS_WHOLE_TILES)
    return fail(ANEMOI_ERR_UNSUPPORTED, "anemoi_linear_actgrad: shape not taken by the fast path");
  return rc;
}

extern "C" int anemoi_linear_stats(int dtype, const void* x, int64_t ldx, const void* w, const float* bias,
                                   const float* colsum, const float* stats_in, const void* residual, int64_t ldr,
                                   void* y, int64_t ldy, int64_t M, int N, int K, void* workspace,
                                   int64_t workspace_bytes, float eps, float* stats_out, anemoi_stream_t stream) {
  using namespace anemoi;
  ANEMOI_REQUIRE(stats_out != nullptr && (uintptr_t)stats_out % 8 == 0, ANEMOI_ERR_INVALID,
                 "anemoi_linear_stats: stats_out must be a non-null 8-byte aligned pointer");
  ANEMOI_REQUIRE((colsum == nullptr) == (stats_in == nullptr), ANEMOI_ERR_INVALID,
                 "anemoi_linear_stats: colsum and stats_in come together");
  int64_t rows_done = 0;
  LnFold ln{colsum, reinterpret_cast<const float2*>(stats_in), nullptr, 0, nullptr};
  const int slots = N / 128;
  if (dtype == ANEMOI_BF16 && N % 256 == 0 && workspace != nullptr && (uintptr_t)workspace % 8 == 0 &&
      workspace_bytes >= M * slots * 8) {
    ln.rs_partial = static_cast<float2*>(workspace);
    ln.rs_slots = slots;
    ln.rs_rows_done = &rows_done;
  }
  const int rc = linear_dispatch("anemoi_linear_stats", dtype, dtype, x, ldx, w, bias, ln, residual, ldr, y, ldy, M, N,
                                 K, ANEMOI_ACT_NONE, stream);
  if (rc != ANEMOI_OK) return rc;
  if (rows_done > 0) {
    // rows behind the tiled part (at most 8, from the skinny pass) are folded in by extra blocks of the same launch
    const int64_t tail = (M - rows_done <= 8 && dtype == ANEMOI_BF16) ? M - rows_done : 0;
    hipLaunchKernelGGL(row_sums_finalize_kernel, dim3((unsigned)((rows_done + 255) / 256 + tail)), dim3(256), 0,
                       as_stream(stream), static_cast<const float2*>(workspace), slots, rows_done, N, eps,
                       reinterpret_cast<float2*>(stats_out), static_cast<const bf16_t*>(y), ldy, rows_done + tail);
    const int rl = check_launch("anemoi_linear_stats(finalize)");
    if (rl != ANEMOI_OK) return rl;
    rows_done += tail;
  }
  if (rows_done < M) {  // rows the fused path did not cover (other kernels): statistics from y itself
    const int esz = dtype == ANEMOI_BF16 ? 2 : 4;
    return anemoi_row_stats(dtype, static_cast<const char*>(y) + rows_done * ldy * esz, ldy, stats_out + 2 * rows_done,
                            M - rows_done, N, eps, stream);
  }
  return ANEMOI_OK;
}


// ---------------------------------------------------------------------------------------------
// `batch` independent products y[b] = x[b] w[b]^T (no bias / activation) on the 128 x 128 kernel, problems along grid.y.
// The weight-gradient GEMMs use it as a deterministic split of their long reduction (anemoi_models_amd/autograd.py):
// dW = sum_b dpre[b]^T X[b] over row chunks b, so that a small [N, K] result still fills the chip.
// ---------------------------------------------------------------------------------------------
extern "C" int anemoi_linear_batched(int dtype, int out_dtype, const void* x, int64_t ldx, int64_t stride_x,
                                     const void* w, int64_t stride_w, void* y, int64_t ldy, int64_t stride_y, int batch,
                                     int64_t M, int N, int K, anemoi_stream_t stream) {
  using namespace anemoi;
  ANEMOI_REQUIRE(x && w && y && batch > 0 && batch < 65536 && M >= 0 && N > 0 && K > 0 && ldx >= K && ldy >= N,
                 ANEMOI_ERR_INVALID, "anemoi_linear_batched: bad argument");
  const int esz = dtype == ANEMOI_BF16 ? 2 : 4;
  ANEMOI_REQUIRE(((int64_t)K * esz) % ROW_BYTES == 0 && (uintptr_t)x % 16 == 0 && (uintptr_t)w % 16 == 0 &&
                     (ldx * esz) % 16 == 0 && (stride_x * esz) % 16 == 0 && (stride_w * esz) % 16 == 0,
                 ANEMOI_ERR_INVALID, "anemoi_linear_batched: K must be a multiple of %d elements, operands 16-byte aligned",
                 ROW_BYTES / esz);
  if (M == 0) return ANEMOI_OK;
  hipStream_t st = as_stream(stream);
  const LnFold none{nullptr, nullptr, nullptr, 0, nullptr};
  if (dtype == ANEMOI_F32 && out_dtype == ANEMOI_F32)
    return linear_launch<float, float>(x, ldx, w, nullptr, nullptr, 0, y, ldy, M, N, K, ANEMOI_ACT_NONE, st, none, batch,
                                       stride_x, stride_w, stride_y);
  if (dtype == ANEMOI_BF16 && out_dtype == ANEMOI_F32)
    return linear_launch<bf16_t, float>(x, ldx, w, nullptr, nullptr, 0, y, ldy, M, N, K, ANEMOI_ACT_NONE, st, none, batch,
                                        stride_x, stride_w, stride_y);
  if (dtype == ANEMOI_BF16 && out_dtype == ANEMOI_BF16) {
    if (K >= 128 && N % 8 == 0 && ldy % 8 == 0 && (uintptr_t)y % 16 == 0 && (stride_y * 2) % 16 == 0) {
      // the persistent 256 x 256 kernel walks the tiles of all problems as one list
      LnFold bl = none;
      bl.b_tiles = ((M + BIG_M - 1) / BIG_M) * ((N + BIG_N - 1) / BIG_N);
      bl.b_sx = stride_x;
      bl.b_sw = stride_w;
      bl.b_sy = stride_y;
      bl.b_count = batch;
      const int rc = linear_bf16_256_launch(x, ldx, w, nullptr, nullptr, 0, y, ldy, M, N, K, ANEMOI_ACT_NONE, st, bl);
      if (rc != W4_NEEDS_WHOLE_TILES) return rc;
    }
    return linear_launch<bf16_t, bf16_t>(x, ldx, w, nullptr, nullptr, 0, y, ldy, M, N, K, ANEMOI_ACT_NONE, st, none,
                                         batch, stride_x, stride_w, stride_y);
  }
  return fail(ANEMOI_ERR_UNSUPPORTED, "anemoi_linear_batched: dtype %d -> %d", dtype, out_dtype);
}
