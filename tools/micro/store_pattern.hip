// Lab (not part of the product): what does the lane -> address pattern of a 16-byte buffer store cost?
// Every wave writes [128 rows x 128 columns] bf16 blocks of a row-major matrix (ld = N) -- the per-wave share of a
// 256 x 256 GEMM tile -- with 32 buffer_store_dwordx4 per block, in three lane assignments that move the SAME bytes:
//   A  the GEMM epilogue's: lane = fr + 16 fq -> row fr, 16-byte piece fq of a 64-byte segment   (16 rows x 64 B)
//   B  the same segments, lane-contiguous: lane = 4 r + p -> row r, piece p                      (16 rows x 64 B)
//   C  whole cache lines: lane = 8 r + p -> row r, piece p of a 128-byte line                    ( 8 rows x 128 B)
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/store_pattern.hip -o tools/micro/bin/store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

template <int MODE>
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
    if constexpr (MODE == 0) {
      const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int u = 0; u < 4; ++u)
          __builtin_amdgcn_raw_buffer_store_b128(v, rs, (fr * N + fq * 8 + u * 32) * 2, j * 16 * N * 2, 0);
    } else if constexpr (MODE == 1) {
      const int r = lane >> 2, p = lane & 3;
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int u = 0; u < 4; ++u)
          __builtin_amdgcn_raw_buffer_store_b128(v, rs, (r * N + p * 8 + u * 32) * 2, j * 16 * N * 2, 0);
    } else {
      const int r = lane >> 3, p = lane & 7;
#pragma unroll
      for (int j = 0; j < 16; ++j)
#pragma unroll
        for (int u = 0; u < 2; ++u)
          __builtin_amdgcn_raw_buffer_store_b128(v, rs, (r * N + p * 8 + u * 64) * 2, j * 8 * N * 2, 0);
    }
  }
}

int main() {
  const int64_t M = 40960;
  const int N = 4096;
  unsigned short* y;
  hipMalloc(&y, M * N * 2);
  const int tiles = (int)(M / 256) * (N / 256);
  const int per_wg = (tiles + 255) / 256;
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  const char* names[3] = {"A lane = fr + 16 fq (16 rows x 64 B, GEMM epilogue)", "B lane = 4 row + piece (16 rows x 64 B)",
                          "C lane = 8 row + piece (8 rows x 128 B)"};
  for (int rep = 0; rep < 2; ++rep)
    for (int mode = 0; mode < 3; ++mode) {
      for (int it = 0; it < 3; ++it) {
        if (it == 1) hipEventRecord(a);
        if (mode == 0) store_kernel<0><<<256, 256>>>(y, M, N, per_wg);
        if (mode == 1) store_kernel<1><<<256, 256>>>(y, M, N, per_wg);
        if (mode == 2) store_kernel<2><<<256, 256>>>(y, M, N, per_wg);
      }
      hipEventRecord(b);
      hipDeviceSynchronize();
      float ms;
      hipEventElapsedTime(&ms, a, b);
      ms /= 2;
      printf("%-56s %8.4f ms  %7.1f GB/s  %6.1f cycles@2.4GHz per store instruction and CU\n", names[mode], ms,
             M * N * 2 / ms / 1e6, ms * 1e-3 * 2.4e9 / (per_wg * 128.0));
    }
  return 0;
}
