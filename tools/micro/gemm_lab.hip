// GEMM main-loop laboratory (standalone, not part of the product): C[M,N] = X[M,K] * W[N,K]^T in bf16, 256x256 tile,
// 8 waves, two 64 KiB LDS stages filled by LDS-DMA -- the structure of anemoi::linear_bf16_256_kernel -- with
// alternative main loops selected at compile time:
//   VARIANT 0: one barrier per K-slab, all waves in lockstep (the shipped loop)
//   VARIANT 1: four phases per K-slab, each split into a memory half (ds_read of the NEXT phase's fragments, LDS-DMA
//              issue, counted waits) and a compute half (16 MFMAs), the two wave groups (wm = 0 / 1) staggered by
//              one barrier so that one group's memory half overlaps the other group's MFMAs.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 gemm_lab.hip -o gemm_lab ;  run: ./gemm_lab [M N K]...
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <type_traits>
#include <utility>

typedef uint16_t bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

static inline bf16_t f2bf(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}
static inline float bf2f(bf16_t v) {
  uint32_t u = ((uint32_t)v) << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
__device__ __forceinline__ bf16_t dev_f2bf(float f) {
  uint32_t u = __float_as_uint(f);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}
__device__ __forceinline__ void glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
__device__ __forceinline__ int swz(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }

#define BAR()                              \
  do {                                     \
    asm volatile("" ::: "memory");         \
    __builtin_amdgcn_sched_barrier(0);     \
    __builtin_amdgcn_s_barrier();          \
    __builtin_amdgcn_sched_barrier(0);     \
    asm volatile("" ::: "memory");         \
  } while (0)

constexpr int BM = 256, BN = 256, ROWB = 128, STAGE = (BM + BN) * ROWB;

template <int VARIANT>
__global__ __launch_bounds__(512) void gemm_lab(const bf16_t* __restrict__ X, const bf16_t* __restrict__ W,
                                                bf16_t* __restrict__ Y, int M, int N, int K) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int ntc = N / BN;
  const int nt = blockIdx.x % ntc, mt = blockIdx.x / ntc;
  const int m0 = mt * BM, n0 = nt * BN;
  const int nk = K / 64;
  const int srow = lane >> 3, scp = lane & 7;
  const char* xg[4];
  const char* wg[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = wid * 32 + 8 * i + srow;
    const int c = swz(r, scp);
    xg[i] = reinterpret_cast<const char*>(X + (int64_t)(m0 + r) * K) + c * 16;
    wg[i] = reinterpret_cast<const char*>(W + (int64_t)(n0 + r) * K) + c * 16;
  }
  auto stage = [&](int kt, int buf) {
    char* xs = smem + buf * STAGE + wid * 4096;
    char* ws = xs + BM * ROWB;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      glds16(xg[i] + (int64_t)kt * ROWB, xs + i * 1024);
      glds16(wg[i] + (int64_t)kt * ROWB, ws + i * 1024);
    }
  };
  const int wm = wid >> 2, wn = wid & 3;
  const int fr = lane & 15, fq = lane >> 4;
  auto loadA = [&](bf16x8_t (&a)[4], int buf, int ks) {
    const char* ws = smem + buf * STAGE + BM * ROWB;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = wn * 64 + i * 16 + fr;
      a[i] = *reinterpret_cast<const bf16x8_t*>(ws + row * ROWB + (swz(row, ks * 4 + fq) << 4));
    }
  };
  auto loadB = [&](bf16x8_t (&b)[4], int buf, int ks, int jh) {
    const char* xs = smem + buf * STAGE;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = wm * 128 + (jh * 4 + j) * 16 + fr;
      b[j] = *reinterpret_cast<const bf16x8_t*>(xs + row * ROWB + (swz(row, ks * 4 + fq) << 4));
    }
  };
  f32x4_t acc[4][8];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  auto mma = [&](const bf16x8_t (&a)[4], const bf16x8_t (&b)[4], int jh) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][jh * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][jh * 4 + j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };

  if constexpr (VARIANT == 0) {
    stage(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
      __syncthreads();
      if (kt + 1 < nk) stage(kt + 1, (kt + 1) & 1);
      const int buf = kt & 1;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8_t a[4];
        loadA(a, buf, ks);
#pragma unroll
        for (int jh = 0; jh < 2; ++jh) {
          bf16x8_t b[4];
          loadB(b, buf, ks, jh);
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
              acc[i][jh * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][jh * 4 + j], 0, 0, 0);
        }
      }
    }
  } else {
    bf16x8_t aA[4], aB[4], bA[4], bB[4];
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    BAR();
    loadA(aA, 0, 0);
    loadB(bA, 0, 0, 0);
    if (nk > 1) stage(1, 1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (wm == 1) BAR();  // group B runs one barrier behind group A from here on
    for (int k = 0; k < nk; ++k) {
      const int buf = k & 1, nbuf = buf ^ 1;
      // ---- phase 0: compute (ks0, jh0); fetch fragments of (ks0, jh1)
      loadB(bB, buf, 0, 1);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      BAR();
      mma(aA, bA, 0);
      BAR();
      // ---- phase 1: compute (ks0, jh1); fetch (ks1, jh0)
      loadA(aB, buf, 1);
      loadB(bA, buf, 1, 0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      BAR();
      mma(aA, bB, 1);
      BAR();
      // ---- phase 2: compute (ks1, jh0); fetch (ks1, jh1); the next slab's DMA (issued one slab ago) must have landed
      loadB(bB, buf, 1, 1);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (k + 1 < nk) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      BAR();
      mma(aB, bA, 0);
      BAR();
      // ---- phase 3: compute (ks1, jh1); fetch the next slab's (ks0, jh0); refill this slab's buffer with slab k + 2
      if (k + 1 < nk) {
        loadA(aA, nbuf, 0);
        loadB(bA, nbuf, 0, 0);
      }
      if (k + 2 < nk) stage(k + 2, buf);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      BAR();
      mma(aB, bB, 1);
      BAR();
    }
    if (wm == 0) BAR();
  }

  // simple direct epilogue: lane holds C[m = .. + fr][n = .. + fq*4 + 0..3]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int m = m0 + wm * 128 + j * 16 + fr;
      const int n = n0 + wn * 64 + i * 16 + fq * 4;
      uint2 v;
      v.x = (uint32_t)dev_f2bf(acc[i][j][0]) | ((uint32_t)dev_f2bf(acc[i][j][1]) << 16);
      v.y = (uint32_t)dev_f2bf(acc[i][j][2]) | ((uint32_t)dev_f2bf(acc[i][j][3]) << 16);
      *reinterpret_cast<uint2*>(Y + (int64_t)m * N + n) = v;
    }
}

// VARIANT 2: the same 256 x 256 x 64 tile with FOUR waves (one per SIMD), each owning a 128 x 128 patch: 256
// accumulator registers (AGPRs), fragments of the next half-slab fetched while the 64 MFMAs of the current one run,
// one barrier per K-slab.  A third less LDS read traffic than 8 waves x (128 x 64).
template <int ABL, int G1 = 23, int SP = 5, int G2 = 103, int RD = 1>
__global__ __launch_bounds__(256) void gemm_lab4(const bf16_t* __restrict__ X, const bf16_t* __restrict__ W,
                                                 bf16_t* __restrict__ Y, int M, int N, int K,
                                                 long long* __restrict__ dbg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int ntc = N / BN;
  const int nt = blockIdx.x % ntc, mt = blockIdx.x / ntc;
  const int m0 = mt * BM, n0 = nt * BN;
  const int nk = K / 64;
  const int srow = lane >> 3, scp = lane & 7;
  long long t_lgkm = 0, t_vm = 0, t_bar = 0, t_all = 0;
  // rows r = wid * 64 + 8 i + srow (i = 0..7); (r >> 1) & 7 = (4 i + (srow >> 1)) & 7 -> two swizzle classes (i even / odd).
  // Addresses are (uniform 64-bit base in SGPRs) + (32-bit lane offset): the saddr form of global_load_lds.
  const char* xtile = reinterpret_cast<const char*>(X + (int64_t)m0 * K);
  const char* wtile = reinterpret_cast<const char*>(W + (int64_t)n0 * K);
  uint32_t voff[2];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int r = 8 * p + srow;
    voff[p] = (uint32_t)r * (uint32_t)K * 2u + (uint32_t)swz(r, scp) * 16u;
  }
  const int row16 = 16 * K * 2;
  const int wave_rows = wid * 64 * K * 2;
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)xtile, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)wtile, 0, 0x7fffffff, 0x00020000);
  auto dma = [&](const __amdgpu_buffer_rsrc_t& rs, int i, int kt, char* dst) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)dst, 16, (int)voff[i & 1],
                                             wave_rows + (i >> 1) * row16 + kt * ROWB, 0, 0);
  };
  auto stage = [&](int kt, int buf) {
    char* xs = smem + buf * STAGE + wid * 8192;
    char* ws = xs + BM * ROWB;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      dma(xrs, i, kt, xs + i * 1024);
      dma(wrs, i, kt, ws + i * 1024);
    }
  };
  // register-staged alternative (ABL & 8): plain 16-byte loads, written to LDS by ds_write_b128 half a slab later
  uint4 sink[16];
  auto gld = [&](const __amdgpu_buffer_rsrc_t& rs, int i, int kt) {
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff[i & 1],
                                                         wave_rows + (i >> 1) * row16 + kt * ROWB, 0);
    return make_uint4(v.x, v.y, v.z, v.w);
  };
  const int wm = wid >> 1, wn = wid & 1;
  const int fr = lane & 15, fq = lane >> 4;
  auto loadF = [&](bf16x8_t (&a)[8], bf16x8_t (&b)[8], int buf, int ks) {
    const char* xs = smem + buf * STAGE;
    const char* ws = xs + BM * ROWB;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = wn * 128 + i * 16 + fr;
      a[i] = *reinterpret_cast<const bf16x8_t*>(ws + row * ROWB + (swz(row, ks * 4 + fq) << 4));
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int row = wm * 128 + j * 16 + fr;
      b[j] = *reinterpret_cast<const bf16x8_t*>(xs + row * ROWB + (swz(row, ks * 4 + fq) << 4));
    }
  };
  f32x4_t acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  auto mma = [&](const bf16x8_t (&a)[8], const bf16x8_t (&b)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
  };

  bf16x8_t a0[8], b0[8], a1[8], b1[8];
  auto ldA = [&](int buf, int ks, int i) {
    const int row = wn * 128 + i * 16 + fr;
    return *reinterpret_cast<const bf16x8_t*>(smem + buf * STAGE + BM * ROWB + row * ROWB + (swz(row, ks * 4 + fq) << 4));
  };
  auto ldB = [&](int buf, int ks, int j) {
    const int row = wm * 128 + j * 16 + fr;
    return *reinterpret_cast<const bf16x8_t*>(smem + buf * STAGE + row * ROWB + (swz(row, ks * 4 + fq) << 4));
  };
#define MFMA_A(c, a, b) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b))
  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  BAR();
  loadF(a0, b0, 0, 0);
  if (nk > 1) stage(1, 1);
  const long long t_begin = __builtin_readcyclecounter();
  // One K-slab = two phases of 64 MFMAs; the other work of a phase (16 ds_read_b128 of the next phase's fragments and,
  // in phase 1, the 16 LDS-DMA issues that refill this slab's buffer) is dealt out one instruction per MFMA gap so that
  // the MFMA pipe never waits behind a burst of memory instructions.
  auto slab = [&](int k, auto more_tag) {
    constexpr bool MORE = decltype(more_tag)::value;
    const int buf = k & 1, nbuf = buf ^ 1;
#pragma unroll
    for (int s = 0; s < 64; ++s) {
      MFMA_A(acc[s >> 3][s & 7], a0[s >> 3], b0[s & 7]);
      if (!(ABL & 2) && s % 3 == 0 && s / 3 < 16) {
        const int t = s / 3;
        if (t < 8) b1[t] = ldB(buf, 1, t);
        else a1[t - 8] = ldA(buf, 1, t - 8);
      }
      if ((ABL & 16) && k > 0 && s % 3 == 1 && s / 3 < 16) {  // slab k + 1, loaded during the previous phase 1
        const int t = s / 3, i = t >> 1;
        char* dst = smem + nbuf * STAGE + wid * 8192 + (t & 1 ? BM * ROWB : 0) + i * 1024 + lane * 16;
        *reinterpret_cast<uint4*>(dst) = sink[t];
      }
    }
    long long c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    if (dbg) c0 = __builtin_readcyclecounter();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (dbg) c1 = __builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (dbg) c2 = __builtin_readcyclecounter();
    if (!(ABL & 4)) BAR();
    if (dbg) {
      c3 = __builtin_readcyclecounter();
      t_lgkm += c1 - c0;
      t_vm += c2 - c1;
      t_bar += c3 - c2;
    }
    char* xsd = smem + buf * STAGE + wid * 8192;
#pragma unroll
    for (int s = 0; s < 64; ++s) {
      MFMA_A(acc[s >> 3][s & 7], a1[s >> 3], b1[s & 7]);
      if (!(ABL & 2) && s % 3 == 0 && s / 3 < 16) {
        const int t = s / 3;
        if (t < 8) b0[t] = ldB(nbuf, 0, t);
        else a0[t - 8] = ldA(nbuf, 0, t - 8);
      }
      if ((ABL & 8) && MORE && s % 3 == 1 && s / 3 < 16) {
        const int t = s / 3, i = t >> 1;
        sink[t] = gld(t & 1 ? wrs : xrs, i, k + 2);
      }
      if (!(ABL & 9) && MORE && s % 3 == 1 && s / 3 < 16) {
        const int t = s / 3, i = t >> 1;
        if (t & 1) dma(wrs, i, k + 2, xsd + BM * ROWB + i * 1024);
        else dma(xrs, i, k + 2, xsd + i * 1024);
      }
    }
  };
  // ABL & 32: schedule modelled on the per-slab life of an LDS buffer: the ks = 1 fragments are fetched in the first 16
  // MFMA gaps, barrier 1 (MFMA 23) then frees the WHOLE slab buffer, its refill (16 LDS-DMA) is spread one per five MFMAs
  // (the CU's four waves run in lockstep and share one address pipe: ~16 cycles per 1 KiB piece), barrier 2 (MFMA 103)
  // publishes the other buffer (counted vmcnt: only the previous slab's DMAs must have landed), whose ks = 0 fragments
  // are fetched in the gaps 104..119.
  auto slab2 = [&](int k, auto more_tag) {
    constexpr bool MORE = decltype(more_tag)::value;
    const int buf = k & 1, nbuf = buf ^ 1;
    char* xsd = smem + buf * STAGE + wid * 8192;
    long long c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    auto step = [&](auto s_tag) {
      constexpr int s = decltype(s_tag)::value;
      if constexpr (s < 64) MFMA_A(acc[s >> 3][s & 7], a0[s >> 3], b0[s & 7]);
      else MFMA_A(acc[(s - 64) >> 3][s & 7], a1[(s - 64) >> 3], b1[s & 7]);
      if constexpr (!(ABL & 2) && s * RD < 16) {
#pragma unroll
        for (int u = s * RD; u < s * RD + RD; ++u) {
          if (u < 8) b1[u] = ldB(buf, 1, u);
          else a1[u - 8] = ldA(buf, 1, u - 8);
        }
      }
      if constexpr (s == G1) {
        if (dbg) c0 = __builtin_readcyclecounter();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        BAR();
        if (dbg) c1 = __builtin_readcyclecounter();
      }
      if constexpr (!(ABL & 1) && MORE && s > G1 && (s - G1 - 1) % SP == 0 && (s - G1 - 1) / SP < 16) {
        constexpr int t = (s - G1 - 1) / SP, i = t >> 1;
        if constexpr (t & 1) dma(wrs, i, k + 2, xsd + BM * ROWB + i * 1024);
        else dma(xrs, i, k + 2, xsd + i * 1024);
      }
      if constexpr (s == G2) {
        static_assert(G1 + 1 + 15 * SP < G2, "all 16 DMAs of a slab must be issued before barrier 2 (vmcnt count)");
        if (dbg) c2 = __builtin_readcyclecounter();
        if constexpr (MORE) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (dbg) c3 = __builtin_readcyclecounter();
        BAR();
        if (dbg) {
          t_lgkm += c1 - c0;
          t_vm += c3 - c2;
          t_bar += __builtin_readcyclecounter() - c3;
        }
      }
      if constexpr (!(ABL & 2) && s > G2 && (s - G2 - 1) * RD < 16) {
#pragma unroll
        for (int u = (s - G2 - 1) * RD; u < (s - G2 - 1) * RD + RD; ++u) {
          if (u < 8) b0[u] = ldB(nbuf, 0, u);
          else a0[u - 8] = ldA(nbuf, 0, u - 8);
        }
      }
    };
    [&]<int... S>(std::integer_sequence<int, S...>) { (step(std::integral_constant<int, S>{}), ...); }
    (std::make_integer_sequence<int, 128>{});
  };
  int k = 0;
  if (ABL & 32) {
    for (; k + 2 < nk; ++k) slab2(k, std::true_type{});
    for (; k < nk; ++k) slab2(k, std::false_type{});
  } else {
    for (; k + 2 < nk; ++k) slab(k, std::true_type{});
    for (; k < nk; ++k) slab(k, std::false_type{});
  }
  asm volatile("s_nop 15\n s_nop 15" ::: "memory");
  if ((ABL & 8) && !(ABL & 16)) {
    uint4 x = sink[0];
#pragma unroll
    for (int t = 1; t < 16; ++t) x.x ^= sink[t].x ^ sink[t].y ^ sink[t].z ^ sink[t].w;
    if (x.x == 0x12345678u) Y[0] = 1;
  }
  if (dbg && lane == 0) {
    t_all = __builtin_readcyclecounter() - t_begin;
    long long* d = dbg + ((int64_t)blockIdx.x * 4 + wid) * 4;
    d[0] = t_all;
    d[1] = t_lgkm;
    d[2] = t_vm;
    d[3] = t_bar;
  }

#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int m = m0 + wm * 128 + j * 16 + fr;
      const int n = n0 + wn * 128 + i * 16 + fq * 4;
      uint2 v;
      v.x = (uint32_t)dev_f2bf(acc[i][j][0]) | ((uint32_t)dev_f2bf(acc[i][j][1]) << 16);
      v.y = (uint32_t)dev_f2bf(acc[i][j][2]) | ((uint32_t)dev_f2bf(acc[i][j][3]) << 16);
      *reinterpret_cast<uint2*>(Y + (int64_t)m * N + n) = v;
    }
}

template <int ABL, int G1 = 23, int SP = 5, int G2 = 103, int RD = 1>
static float run4(const bf16_t* x, const bf16_t* w, bf16_t* y, int M, int N, int K, int iters) {
  auto kern4 = gemm_lab4<ABL, G1, SP, G2, RD>;
  hipFuncSetAttribute((const void*)kern4, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE);
  dim3 grid((M / BM) * (N / BN)), block(256);
  long long* dbg = nullptr;
  if (getenv("LAB_DBG")) {
    const size_t n = (size_t)grid.x * 16;
    hipMalloc(&dbg, n * 8);
    hipMemset(dbg, 0, n * 8);
    hipLaunchKernelGGL(kern4, grid, block, 2 * STAGE, 0, x, w, y, M, N, K, dbg);
    hipLaunchKernelGGL(kern4, grid, block, 2 * STAGE, 0, x, w, y, M, N, K, dbg);
    hipDeviceSynchronize();
    std::vector<long long> h(n);
    hipMemcpy(h.data(), dbg, n * 8, hipMemcpyDeviceToHost);
    double s[4] = {0, 0, 0, 0};
    for (size_t i = 0; i < n; ++i) s[i & 3] += (double)h[i];
    const double nw = (double)grid.x * 4;
    printf("  [dbg] per wave avg cycles: loop %.0f  lgkm-wait %.0f  vm-wait %.0f  barrier %.0f  (slabs %d -> ideal MFMA %d cycles)\n",
           s[0] / nw, s[1] / nw, s[2] / nw, s[3] / nw, K / 64, K / 64 * 2048);
    for (int b = 0; b < 2; ++b)
      for (int w2 = 0; w2 < 4; ++w2)
        printf("  [dbg] block %d wave %d: %lld %lld %lld %lld\n", b, w2, h[(b * 4 + w2) * 4], h[(b * 4 + w2) * 4 + 1],
               h[(b * 4 + w2) * 4 + 2], h[(b * 4 + w2) * 4 + 3]);
    hipFree(dbg);
    dbg = nullptr;
  }
  hipLaunchKernelGGL(kern4, grid, block, 2 * STAGE, 0, x, w, y, M, N, K, dbg);
  hipDeviceSynchronize();
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  hipEventRecord(a);
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern4, grid, block, 2 * STAGE, 0, x, w, y, M, N, K, dbg);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0.f;
  hipEventElapsedTime(&ms, a, b);
  return ms / iters;
}

template <int VARIANT>
static float run(const bf16_t* x, const bf16_t* w, bf16_t* y, int M, int N, int K, int iters) {
  hipFuncSetAttribute((const void*)gemm_lab<VARIANT>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE);
  dim3 grid((M / BM) * (N / BN)), block(512);
  hipLaunchKernelGGL(gemm_lab<VARIANT>, grid, block, 2 * STAGE, 0, x, w, y, M, N, K);
  hipDeviceSynchronize();
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  hipEventRecord(a);
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(gemm_lab<VARIANT>, grid, block, 2 * STAGE, 0, x, w, y, M, N, K);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0.f;
  hipEventElapsedTime(&ms, a, b);
  return ms / iters;
}

int main(int argc, char** argv) {
  std::vector<int> shapes = {4096, 4096, 1024, 4096, 4096, 8192, 40960, 4096, 1024, 40960, 1024, 4096};
  if (argc > 3) {
    shapes.clear();
    for (int i = 1; i + 2 < argc; i += 3) {
      shapes.push_back(atoi(argv[i]));
      shapes.push_back(atoi(argv[i + 1]));
      shapes.push_back(atoi(argv[i + 2]));
    }
  }
  for (size_t s = 0; s + 2 < shapes.size(); s += 3) {
    const int M = shapes[s], N = shapes[s + 1], K = shapes[s + 2];
    std::vector<bf16_t> hx((size_t)M * K), hw((size_t)N * K), hy((size_t)M * N);
    uint32_t seed = 12345;
    auto rnd = [&]() {
      seed = seed * 1664525u + 1013904223u;
      return ((seed >> 8) & 0xffff) / 32768.0f - 1.0f;
    };
    for (auto& v : hx) v = f2bf(rnd());
    for (auto& v : hw) v = f2bf(rnd() * 0.05f);
    bf16_t *dx, *dw, *dy;
    hipMalloc(&dx, hx.size() * 2);
    hipMalloc(&dw, hw.size() * 2);
    hipMalloc(&dy, hy.size() * 2);
    hipMemcpy(dx, hx.data(), hx.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    for (int variant = 0; variant < 9; ++variant) {
      hipMemset(dy, 0, hy.size() * 2);
      const float ms = variant == 0   ? run<0>(dx, dw, dy, M, N, K, 20)
                       : variant == 1 ? run<1>(dx, dw, dy, M, N, K, 20)
                                      : variant == 2 ? run4<0>(dx, dw, dy, M, N, K, 20)
                       : variant == 3 ? run4<32>(dx, dw, dy, M, N, K, 20)
                       : variant == 4 ? run4<32, 15, 5, 103, 2>(dx, dw, dy, M, N, K, 20)
                       : variant == 5 ? run4<32, 15, 6, 111, 2>(dx, dw, dy, M, N, K, 20)
                       : variant == 6 ? run4<32, 23, 4, 103, 1>(dx, dw, dy, M, N, K, 20)
                       : variant == 7 ? run4<32, 19, 5, 107, 2>(dx, dw, dy, M, N, K, 20)
                                      : run4<32, 23, 5, 111, 2>(dx, dw, dy, M, N, K, 20);
      hipMemcpy(hy.data(), dy, hy.size() * 2, hipMemcpyDeviceToHost);
      double max_err = 0.0;
      for (int t = 0; t < 400; ++t) {
        seed = seed * 1664525u + 1013904223u;
        const int m = (seed >> 4) % M;
        seed = seed * 1664525u + 1013904223u;
        const int n = (seed >> 4) % N;
        double ref = 0.0;
        for (int k = 0; k < K; ++k) ref += (double)bf2f(hx[(size_t)m * K + k]) * bf2f(hw[(size_t)n * K + k]);
        const double e = fabs(ref - bf2f(hy[(size_t)m * N + n])) / (fabs(ref) + 1.0);
        if (e > max_err) max_err = e;
      }
      printf("M=%6d N=%5d K=%5d variant %d: %8.4f ms  %7.1f TFLOP/s  max rel err %.2e %s\n", M, N, K, variant, ms,
             2.0 * M * N * K / ms / 1e9, max_err, max_err < 2e-2 ? "OK" : "WRONG");
    }
    hipFree(dx);
    hipFree(dw);
    hipFree(dy);
  }
  return 0;
}
