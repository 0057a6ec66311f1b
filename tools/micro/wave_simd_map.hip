// Which SIMD does wave w of a 512-thread workgroup run on?  (s_getreg_b32 HW_REG_HW_ID: SIMD_ID = bits 5:4, CU_ID = 11:8)
//   hipcc --offload-arch=gfx950 -O2 tools/micro/wave_simd_map.hip -o tools/micro/bin/wave_simd_map && tools/micro/bin/wave_simd_map
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void probe(unsigned* out) {
  unsigned id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = id;
}

int main() {
  for (int threads : {256, 512}) {
    unsigned* d;
    const int blocks = 6, waves = threads / 64;
    hipMalloc(&d, blocks * waves * sizeof(unsigned));
    hipLaunchKernelGGL(probe, dim3(blocks), dim3(threads), 0, 0, d);
    unsigned h[64];
    hipMemcpy(h, d, blocks * waves * sizeof(unsigned), hipMemcpyDeviceToHost);
    for (int b = 0; b < blocks; ++b) {
      printf("threads %d block %d:", threads, b);
      for (int w = 0; w < waves; ++w) printf("  w%d simd %u cu %u slot %u", w, (h[b * waves + w] >> 4) & 3, (h[b * waves + w] >> 8) & 15, h[b * waves + w] & 15);
      printf("\n");
    }
    hipFree(d);
  }
  return 0;
}
