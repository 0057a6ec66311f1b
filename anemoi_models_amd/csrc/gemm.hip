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
#include "common.hpp"

namespace anemoi {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

constexpr int BM = 128, BN = 128, ROW_BYTES = 128;
constexpr int TILE_BYTES = BM * ROW_BYTES;  // 16 KiB per operand per stage

__device__ __forceinline__ void glds16(const void* gptr, void* lptr) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gptr,
                                   (__attribute__((address_space(3))) void*)lptr, 16, 0, 0);
}

__device__ __forceinline__ int swz(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }

template <typename T, typename TO, int VEC>
__device__ __forceinline__ void epilogue_store(const float (&acc)[VEC], int64_t m, int n, int64_t M, int N,
                                               const float* __restrict__ bias, const T* __restrict__ R, int64_t ldr,
                                               TO* __restrict__ Y, int64_t ldy, int act, bool vec_ok) {
  if (m >= M || n >= N) return;
  float o[VEC];
  if (vec_ok && n + VEC <= N) {
    if (bias != nullptr) {
      float b[VEC];
      VecIO<float, VEC>::load(bias + n, b);
#pragma unroll
      for (int i = 0; i < VEC; ++i) o[i] = acc[i] + b[i];
    } else {
#pragma unroll
      for (int i = 0; i < VEC; ++i) o[i] = acc[i];
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
        float t = acc[i] + (bias != nullptr ? bias[n + i] : 0.f);
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
                                                     int K, int act, int vec_ok) {
  __shared__ __attribute__((aligned(16))) char smem[4 * TILE_BYTES];  // 2 stages x (x tile + W tile) = 64 KiB
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
      __syncthreads();  // tile kt landed (vmcnt(0) is part of the barrier while an LDS-DMA is in flight)
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
        epilogue_store<T, TO, 4>(v, m, n, M, N, bias, R, ldr, Y, ldy, act, vec_ok != 0);
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
          epilogue_store<T, TO, 4>(v, m, n, M, N, bias, R, ldr, Y, ldy, act, vec_ok != 0);
        }
  }
}

template <typename T, typename TO>
static int linear_launch(const void* x, int64_t ldx, const void* w, const float* bias, const void* residual,
                         int64_t ldr, void* y, int64_t ldy, int64_t M, int N, int K, int act, hipStream_t st) {
  const int64_t mt = (M + BM - 1) / BM;
  const int64_t nt = (N + BN - 1) / BN;
  ANEMOI_REQUIRE(mt * nt < (int64_t)1 << 31, ANEMOI_ERR_UNSUPPORTED, "anemoi_linear: grid too large");
  const bool vec_ok = (N % 4 == 0) && (ldy % 4 == 0) && ((uintptr_t)y % 16 == 0) &&
                      (bias == nullptr || (uintptr_t)bias % 16 == 0) &&
                      (residual == nullptr || (ldr % 4 == 0 && (uintptr_t)residual % 16 == 0));
  hipLaunchKernelGGL((linear_kernel<T, TO>), dim3((unsigned)(mt * nt)), dim3(256), 0, st, static_cast<const T*>(x),
                     ldx, static_cast<const T*>(w), bias, static_cast<const T*>(residual), ldr, static_cast<TO*>(y),
                     ldy, M, N, K, act, vec_ok ? 1 : 0);
  return check_launch("anemoi_linear");
}

}  // namespace anemoi

using namespace anemoi;

extern "C" int anemoi_linear(int dtype, int out_dtype, const void* x, int64_t ldx, const void* w, const float* bias,
                             const void* residual, int64_t ldr, void* y, int64_t ldy, int64_t M, int N, int K, int act,
                             anemoi_stream_t stream) {
  ANEMOI_REQUIRE(x && w && y, ANEMOI_ERR_INVALID, "anemoi_linear: null pointer");
  ANEMOI_REQUIRE(M >= 0 && N > 0 && K > 0, ANEMOI_ERR_INVALID, "anemoi_linear: bad shape M=%lld N=%d K=%d",
                 (long long)M, N, K);
  ANEMOI_REQUIRE(ldx >= K && ldy >= N && (residual == nullptr || ldr >= N), ANEMOI_ERR_INVALID,
                 "anemoi_linear: leading dimension too small");
  ANEMOI_REQUIRE(act >= ANEMOI_ACT_NONE && act <= ANEMOI_ACT_RELU, ANEMOI_ERR_INVALID, "anemoi_linear: act %d", act);
  const int esz = dtype == ANEMOI_BF16 ? 2 : 4;
  ANEMOI_REQUIRE(((int64_t)K * esz) % ROW_BYTES == 0, ANEMOI_ERR_INVALID,
                 "anemoi_linear: K=%d must be a multiple of %d for this dtype (pad with zeros)", K, ROW_BYTES / esz);
  ANEMOI_REQUIRE((uintptr_t)x % 16 == 0 && (uintptr_t)w % 16 == 0 && (ldx * esz) % 16 == 0, ANEMOI_ERR_INVALID,
                 "anemoi_linear: x / W must be 16-byte aligned with a 16-byte multiple row pitch");
  if (M == 0) return ANEMOI_OK;
  hipStream_t st = as_stream(stream);
  if (dtype == ANEMOI_F32 && out_dtype == ANEMOI_F32)
    return linear_launch<float, float>(x, ldx, w, bias, residual, ldr, y, ldy, M, N, K, act, st);
  if (dtype == ANEMOI_BF16 && out_dtype == ANEMOI_BF16)
    return linear_launch<bf16_t, bf16_t>(x, ldx, w, bias, residual, ldr, y, ldy, M, N, K, act, st);
  if (dtype == ANEMOI_BF16 && out_dtype == ANEMOI_F32)
    return linear_launch<bf16_t, float>(x, ldx, w, bias, residual, ldr, y, ldy, M, N, K, act, st);
  return fail(ANEMOI_ERR_UNSUPPORTED, "anemoi_linear: dtype %d -> %d", dtype, out_dtype);
}
