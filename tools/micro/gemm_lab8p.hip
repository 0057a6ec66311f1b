// GEMM main-loop laboratory #2 (standalone, not part of the product): C[M,N] = X[M,K] * W[N,K]^T in bf16.
// 256 x 256 x 64 tile, 8 waves as 2 (M) x 4 (N), wave tile 128 x 64.  Quadrant phases:
//   * LDS holds two K-tiles, each as four 16 KiB half-tiles (X0, X1, W0, W1): X-half h = the rows {wm * 128 + h * 64 ..
//     + 63} of both wave rows, W-half h = the rows {wn * 64 + h * 32 .. + 31} of the four wave columns.  A phase computes
//     one quadrant (64 x 32) of the wave tile over the whole K = 64 (16 MFMAs) and reads 12 / 4 / 8 / 0 fragments; a
//     half-tile is dead two phases after its only reading phase and is refilled then with the K-tile after next --
//     ONE half-tile (2 LDS-DMA instructions per wave) per phase, five to six phases ahead of its use, counted vmcnt.
//   * the two wave rows (wm = 0 / 1) run one barrier apart: one row's 16 MFMAs overlap the other row's ds_reads and
//     DMA issue on every SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 gemm_lab8p.hip -o gemm_lab8p ;  run: ./gemm_lab8p [M N K]...
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

typedef uint16_t bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;

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
__device__ __forceinline__ uint32_t pack2(float lo, float hi) {
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}

#define BAR()                          \
  do {                                 \
    __builtin_amdgcn_sched_barrier(0); \
    __builtin_amdgcn_s_barrier();      \
    __builtin_amdgcn_sched_barrier(0); \
  } while (0)
#define WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

constexpr int BM = 256, BN = 256, ROWB = 128;
constexpr int HALF = 128 * ROWB;   // 16 KiB half-tile
constexpr int KTILE = 4 * HALF;    // X0, X1, W0, W1
enum { HX0 = 0, HX1 = 1, HW0 = 2, HW1 = 3 };

__global__ __launch_bounds__(512) void gemm_lab8p(const bf16_t* __restrict__ X, const bf16_t* __restrict__ W,
                                                  bf16_t* __restrict__ Y, int M, int N, int K,
                                                  long long* __restrict__ dbg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int ntc = N / BN;
  const int nt = blockIdx.x % ntc, mt = blockIdx.x / ntc;
  const int m0 = mt * BM, n0 = nt * BN;
  const int nk = K / 64;
  const int wm = wid >> 2, wn = wid & 3;
  const int fr = lane & 15, fq = lane >> 4;

  // ---- staging: this wave fills LDS rows L = wid * 16 + g * 8 + (lane >> 3), g = 0 / 1, of every half-tile
  const __amdgpu_buffer_rsrc_t xrs =
      __builtin_amdgcn_make_buffer_rsrc((void*)(X + (int64_t)m0 * K), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t wrs =
      __builtin_amdgcn_make_buffer_rsrc((void*)(W + (int64_t)n0 * K), 0, 0x7fffffff, 0x00020000);
  int vx[2][2], vw[2][2];  // [half][g] lane byte offsets inside the 256-row operand panel
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const int L = wid * 16 + g * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((L >> 1) & 7);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int xr = (L >> 6) * 128 + h * 64 + (L & 63);
      const int wr = (L >> 5) * 64 + h * 32 + (L & 31);
      vx[h][g] = xr * K * 2 + c * 16;
      vw[h][g] = wr * K * 2 + c * 16;
    }
  }
  auto stage = [&](int which, int kt) {  // one half-tile of K-tile kt into buffer kt & 1
    char* dst = smem + (kt & 1) * KTILE + which * HALF + wid * 2048;
    const int so = kt * ROWB;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      if (which == HX0 || which == HX1)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (__attribute__((address_space(3))) void*)(dst + g * 1024), 16,
                                                 vx[which == HX1][g], so, 0, 0);
      else
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (__attribute__((address_space(3))) void*)(dst + g * 1024), 16,
                                                 vw[which == HW1][g], so, 0, 0);
    }
  };

  // ---- fragment reads
  bf16x8_t bx[4][2], aw0[2][2], aw1[2][2];
  auto read_x = [&](int d, int h) {
    const char* base = smem + d * KTILE + (h ? HX1 : HX0) * HALF;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = wm * 64 + j * 16 + fr;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        bx[j][ks] = *reinterpret_cast<const bf16x8_t*>(base + row * ROWB + (((ks * 4 + fq) ^ ((row >> 1) & 7)) << 4));
    }
  };
  auto read_w = [&](bf16x8_t (&aw)[2][2], int d, int h) {
    const char* base = smem + d * KTILE + (h ? HW1 : HW0) * HALF;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = wn * 32 + i * 16 + fr;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        aw[i][ks] = *reinterpret_cast<const bf16x8_t*>(base + row * ROWB + (((ks * 4 + fq) ^ ((row >> 1) & 7)) << 4));
    }
  };
  f32x4_t acc[4][8];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  auto mma = [&](const bf16x8_t (&aw)[2][2], auto ih_tag, auto jh_tag) {
    constexpr int ih = decltype(ih_tag)::value, jh = decltype(jh_tag)::value;
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[ih * 2 + i][jh * 4 + j] =
              __builtin_amdgcn_mfma_f32_16x16x32_bf16(aw[i][ks], bx[j][ks], acc[ih * 2 + i][jh * 4 + j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;

  long long t_mem = 0, t_mma = 0;
  // ---- prologue: K-tile 0 whole, X0 / W0 of K-tile 1
  stage(HX0, 0);
  stage(HW0, 0);
  stage(HW1, 0);
  stage(HX1, 0);
  if (nk > 1) {
    stage(HX0, 1);
    stage(HW0, 1);
  }
  WAIT_VM(0);
  BAR();
  if (wm == 1) BAR();  // wave row 1 runs one barrier behind wave row 0 from here on
  const long long t_begin = __builtin_readcyclecounter();

  // One K-tile = four phases {ds_reads; stage one half-tile; counted vmcnt; barrier; lgkmcnt(0); 16 MFMAs; barrier}.
  // MODE 0: steady state; MODE 1: K-tile nk - 2 (nothing to stage for t + 2); MODE 2: last K-tile (nothing to stage).
  auto ktile = [&](int t, auto mode_tag) {
    constexpr int MODE = decltype(mode_tag)::value;
    const int d = t & 1;
    long long c0 = 0, c1 = 0, c2 = 0;
#define PHASE_BEGIN() \
  if (dbg) c0 = __builtin_readcyclecounter();
#define PHASE_MID()   \
  BAR();              \
  WAIT_LGKM0();       \
  __builtin_amdgcn_sched_barrier(0); \
  if (dbg) c1 = __builtin_readcyclecounter();
#define PHASE_END()                         \
  BAR();                                    \
  if (dbg) {                                \
    c2 = __builtin_readcyclecounter();      \
    t_mem += c1 - c0;                       \
    t_mma += c2 - c1;                       \
  }
    // P1: quadrant (ih 0, jh 0)
    PHASE_BEGIN();
    read_w(aw0, d, 0);
    __builtin_amdgcn_sched_barrier(0);
    read_x(d, 0);
    if (MODE <= 1) stage(HW1, t + 1);
    if (MODE == 0 || MODE == 1) WAIT_VM(8);
    else WAIT_VM(2);
    PHASE_MID();
    mma(aw0, I0{}, I0{});
    PHASE_END();
    // P2: quadrant (ih 1, jh 0)
    PHASE_BEGIN();
    read_w(aw1, d, 1);
    if (MODE <= 1) stage(HX1, t + 1);
    if (MODE == 0 || MODE == 1) WAIT_VM(8);
    else WAIT_VM(0);
    PHASE_MID();
    mma(aw1, I1{}, I0{});
    PHASE_END();
    // P3: quadrant (ih 1, jh 1)
    PHASE_BEGIN();
    read_x(d, 1);
    if (MODE == 0) stage(HX0, t + 2);
    if (MODE == 0) WAIT_VM(8);
    else if (MODE == 1) WAIT_VM(6);
    else WAIT_VM(0);
    PHASE_MID();
    mma(aw1, I1{}, I1{});
    PHASE_END();
    // P4: quadrant (ih 0, jh 1)
    PHASE_BEGIN();
    if (MODE == 0) stage(HW0, t + 2);
    if (MODE == 0) WAIT_VM(8);
    else if (MODE == 1) WAIT_VM(4);
    else WAIT_VM(0);
    PHASE_MID();
    mma(aw0, I0{}, I1{});
    PHASE_END();
  };
  int t = 0;
  for (; t + 2 < nk; ++t) ktile(t, I0{});
  if (nk >= 2) {
    ktile(t, I1{});
    ++t;
  }
  ktile(t, std::integral_constant<int, 2>{});
  if (wm == 0) BAR();

  if (dbg && lane == 0) {
    long long* dd = dbg + ((int64_t)blockIdx.x * 8 + wid) * 4;
    dd[0] = __builtin_readcyclecounter() - t_begin;
    dd[1] = t_mem;
    dd[2] = t_mma;
    dd[3] = 0;
  }
  // simple direct epilogue: lane holds C[m = .. + fr][n = .. + fq * 4 + 0..3]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int m = m0 + wm * 128 + j * 16 + fr;
      const int n = n0 + wn * 64 + i * 16 + fq * 4;
      uint2 v;
      v.x = pack2(acc[i][j][0], acc[i][j][1]);
      v.y = pack2(acc[i][j][2], acc[i][j][3]);
      *reinterpret_cast<uint2*>(Y + (int64_t)m * N + n) = v;
    }
}

static float run(const bf16_t* x, const bf16_t* w, bf16_t* y, int M, int N, int K, int iters) {
  (void)hipFuncSetAttribute((const void*)gemm_lab8p, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * KTILE);
  dim3 grid((M / BM) * (N / BN)), block(512);
  long long* dbg = nullptr;
  if (getenv("LAB_DBG")) {
    const size_t n = (size_t)grid.x * 32;
    (void)hipMalloc(&dbg, n * 8);
    (void)hipMemset(dbg, 0, n * 8);
    hipLaunchKernelGGL(gemm_lab8p, grid, block, 2 * KTILE, 0, x, w, y, M, N, K, dbg);
    hipLaunchKernelGGL(gemm_lab8p, grid, block, 2 * KTILE, 0, x, w, y, M, N, K, dbg);
    (void)hipDeviceSynchronize();
    std::vector<long long> h(n);
    (void)hipMemcpy(h.data(), dbg, n * 8, hipMemcpyDeviceToHost);
    double s[4] = {0, 0, 0, 0};
    for (size_t i = 0; i < n; ++i) s[i & 3] += (double)h[i];
    const double nw = (double)grid.x * 8;
    printf("  [dbg] per wave avg cycles: loop %.0f  mem-half %.0f  mma-half %.0f  (K-tiles %d -> ideal MFMA %d cycles/SIMD)\n",
           s[0] / nw, s[1] / nw, s[2] / nw, K / 64, K / 64 * 2048);
    (void)hipFree(dbg);
    dbg = nullptr;
  }
  hipLaunchKernelGGL(gemm_lab8p, grid, block, 2 * KTILE, 0, x, w, y, M, N, K, dbg);
  (void)hipDeviceSynchronize();
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  (void)hipEventRecord(a);
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(gemm_lab8p, grid, block, 2 * KTILE, 0, x, w, y, M, N, K, dbg);
  (void)hipEventRecord(b);
  (void)hipEventSynchronize(b);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, a, b);
  return ms / iters;
}

int main(int argc, char** argv) {
  std::vector<int> shapes = {512, 512, 128, 4096, 4096, 1024, 8192, 8192, 8192, 40960, 4096, 1024, 40960, 1024, 4096};
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
    (void)hipMalloc(&dx, hx.size() * 2);
    (void)hipMalloc(&dw, hw.size() * 2);
    (void)hipMalloc(&dy, hy.size() * 2);
    (void)hipMemcpy(dx, hx.data(), hx.size() * 2, hipMemcpyHostToDevice);
    (void)hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
      (void)hipMemset(dy, 0, hy.size() * 2);
      const float ms = run(dx, dw, dy, M, N, K, 20);
      (void)hipMemcpy(hy.data(), dy, hy.size() * 2, hipMemcpyDeviceToHost);
      double max_err = 0.0;
      const int checks = (M * (size_t)N <= 512 * 512) ? M * N : 2000;
      for (int c = 0; c < checks; ++c) {
        int m, n;
        if (checks == M * N) {
          m = c / N;
          n = c % N;
        } else {
          seed = seed * 1664525u + 1013904223u;
          m = (seed >> 4) % M;
          seed = seed * 1664525u + 1013904223u;
          n = (seed >> 4) % N;
        }
        double ref = 0.0;
        for (int k = 0; k < K; ++k) ref += (double)bf2f(hx[(size_t)m * K + k]) * bf2f(hw[(size_t)n * K + k]);
        const double e = fabs(ref - bf2f(hy[(size_t)m * N + n])) / (fabs(ref) + 1.0);
        if (e > max_err) max_err = e;
      }
      printf("M=%6d N=%5d K=%5d 8-phase run %d: %8.4f ms  %7.1f TFLOP/s  max rel err %.2e %s\n", M, N, K, rep, ms,
             2.0 * M * N * K / ms / 1e9, max_err, max_err < 2e-2 ? "OK" : "WRONG");
    }
    (void)hipFree(dx);
    (void)hipFree(dw);
    (void)hipFree(dy);
  }
  return 0;
}
