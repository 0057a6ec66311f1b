// Lab (not part of the product): is the 16-byte store path bound per CU or per chip?  The GEMM epilogue's store pattern
// (tools/micro/store_pattern.hip, mode A) from 16 ... 512 workgroups of 256 threads, every workgroup writing the same 20
// tiles' worth (2.6 MB): GB/s per workgroup should stay flat if a CU's own path is the limit, and fall with the number of
// workgroups if the chip's write bandwidth is.
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/store_scale.hip -o tools/micro/bin/store_scale
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

__global__ __launch_bounds__(256) void store_kernel(unsigned short* Y, int64_t M, int N, int tiles_per_wg) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int nt_count = N / 256;
  for (int t = 0; t < tiles_per_wg; ++t) {
    const int64_t tile = (int64_t)blockIdx.x + (int64_t)t * gridDim.x;
    const int64_t mt = tile / nt_count;
    const int nt = (int)(tile % nt_count);
    if ((mt + 1) * 256 > M) break;
    unsigned short* base = Y + (mt * 256 + wm * 128) * N + nt * 256 + wn * 128;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 128 * N * 2, 0x00020000);
    const u32x4_t v = {(unsigned)lane, (unsigned)t, (unsigned)wid, 7u};
    const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int u = 0; u < 4; ++u)
        __builtin_amdgcn_raw_buffer_store_b128(v, rs, (fr * N + fq * 8 + u * 32) * 2, j * 16 * N * 2, 0);
  }
}

int main() {
  const int N = 4096, per_wg = 20;
  const int64_t M_max = 256 * ((512 * per_wg + 15) / 16);
  unsigned short* y;
  hipMalloc(&y, M_max * N * 2);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  const int grids[] = {8, 16, 32, 64, 128, 256, 512};
  for (int rep = 0; rep < 2; ++rep)
    for (int g : grids) {
      const int64_t M = 256 * (((int64_t)g * per_wg + 15) / 16);
      for (int it = 0; it < 4; ++it) {
        if (it == 1) hipEventRecord(a);
        store_kernel<<<g, 256>>>(y, M, N, per_wg);
      }
      hipEventRecord(b);
      hipDeviceSynchronize();
      float ms;
      hipEventElapsedTime(&ms, a, b);
      ms /= 3;
      const double bytes = (double)g * per_wg * 256 * 256 * 2;
      printf("%4d workgroups: %8.4f ms  %7.1f GB/s total  %6.2f GB/s per workgroup\n", g, ms, bytes / ms / 1e6, bytes / ms / 1e6 / g);
    }
  return 0;
}
