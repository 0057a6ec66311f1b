// Micro-benchmark: sustained L2 -> LDS (or -> VGPR) bandwidth per CU for the GEMM staging pattern.
// Every workgroup (512 threads, 1 per CU) repeatedly pulls a 64 KiB slab (512 rows x 128 B, full-line reads) from an
// L2/MALL-resident buffer.  mode 0: global_load_lds (LDS-DMA); 1: global_load_dwordx4 -> VGPR -> ds_write_b128;
// 2: half/half; 3: VGPR only (no LDS write).  Build: hipcc --offload-arch=gfx950 -O3 l2_to_lds.hip -o l2_to_lds
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__device__ __forceinline__ void glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
// GEMM access pattern: a 64 KiB slab = 512 rows x 128 B taken from a row-major matrix with `pitch` bytes per row
// (two operand tiles of 256 rows); consecutive slabs step 128 B along the rows.
__global__ __launch_bounds__(512) void kstrided(const char* __restrict__ src, size_t span, int iters, int pitch,
                                                float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  float acc = 0.f;
  const int slabs_per_row = pitch / 128;
  for (int it = 0; it < iters; ++it) {
    // tile origin: 512 consecutive rows; neighbouring workgroups share half of their rows (like tiles sharing a panel)
    const size_t row0 = ((size_t)(blockIdx.x / 2) * 256 + (size_t)(it / slabs_per_row) * 4096) % (span / pitch - 512);
    const char* p = src + row0 * pitch + (size_t)(it % slabs_per_row) * 128;
    char* l = smem + (it & 1) * 65536 + wid * 8192;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int r = (wid * 8 + i) * 8 + (lane >> 3);
      glds16(p + (size_t)r * pitch + (lane & 7) * 16, l + i * 1024);
    }
    __syncthreads();
    acc += reinterpret_cast<float*>(smem)[(it & 1) * 16384 + threadIdx.x];
  }
  if (acc == 12345.678f) sink[0] = acc;
}

template <int MODE>
__global__ __launch_bounds__(512) void k(const char* __restrict__ src, size_t span, int iters, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    const char* p = src + (((size_t)blockIdx.x * 7 + (size_t)it * 257) * 65536) % span;  // span % 65536 == 0
    char* l = smem + (it & 1) * 65536 + wid * 8192;
    uint4 r[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const char* g = p + (wid * 8 + i) * 1024 + lane * 16;
      if (MODE == 0 || (MODE == 2 && i < 4)) glds16(g, l + i * 1024);
      else r[i] = *reinterpret_cast<const uint4*>(g);
    }
    if (MODE == 1 || MODE == 2) {
#pragma unroll
      for (int i = (MODE == 2 ? 4 : 0); i < 8; ++i) *reinterpret_cast<uint4*>(l + i * 1024 + lane * 16) = r[i];
    }
    if (MODE == 3) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc += __uint_as_float(r[i].x ^ r[i].w);
    }
    __syncthreads();
    acc += reinterpret_cast<float*>(smem)[(it & 1) * 16384 + threadIdx.x];
  }
  if (acc == 12345.678f) sink[0] = acc;
}
int main() {
  const size_t span = 64ull << 20;  // 64 MiB: fits the 256 MiB Infinity Cache, 8 MiB per XCD share exceeds the 4 MiB L2
  char* d; float* sink;
  hipMalloc(&d, span + (1 << 20)); hipMemset(d, 1, span + (1 << 20)); hipMalloc(&sink, 4);
  const int iters = 2000;
  for (size_t sp : {(size_t)(2ull << 20), (size_t)(16ull << 20), span}) {
    for (int mode = 0; mode < 4; ++mode) {
      hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
      auto launch = [&]() {
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 131072, 0, d, sp, iters, sink);
        if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 131072, 0, d, sp, iters, sink);
        if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(512), 131072, 0, d, sp, iters, sink);
        if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(512), 131072, 0, d, sp, iters, sink);
      };
      hipFuncSetAttribute((const void*)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
      hipFuncSetAttribute((const void*)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
      hipFuncSetAttribute((const void*)k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
      hipFuncSetAttribute((const void*)k<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
      launch(); hipDeviceSynchronize();
      hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      const double bytes = 256.0 * iters * 65536.0;
      printf("span %4zu MiB mode %d: %.3f ms  %.2f TB/s aggregate  %.1f GB/s per CU\n", sp >> 20, mode, ms,
             bytes / ms / 1e9, bytes / ms / 1e6 / 256);
    }
  }
  hipFuncSetAttribute((const void*)kstrided, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  for (size_t sp : {(size_t)(16ull << 20), span}) {
    for (int pitch : {128, 2048, 2176, 8192}) {
      hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
      hipLaunchKernelGGL(kstrided, dim3(256), dim3(512), 131072, 0, d, sp, iters, pitch, sink); hipDeviceSynchronize();
      hipEventRecord(a);
      hipLaunchKernelGGL(kstrided, dim3(256), dim3(512), 131072, 0, d, sp, iters, pitch, sink);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      const double bytes = 256.0 * iters * 65536.0;
      printf("strided span %4zu MiB pitch %5d: %.3f ms  %.2f TB/s aggregate  %.1f GB/s per CU\n", sp >> 20, pitch, ms,
             bytes / ms / 1e9, bytes / ms / 1e6 / 256);
    }
  }
  return 0;
}
