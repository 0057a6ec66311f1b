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
    for (int variant = 0; variant < 2; ++variant) {
      hipMemset(dy, 0, hy.size() * 2);
      const float ms = variant == 0 ? run<0>(dx, dw, dy, M, N, K, 20) : run<1>(dx, dw, dy, M, N, K, 20);
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
