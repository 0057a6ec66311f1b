// Checks the raw-buffer range check on gfx950: is the SGPR offset part of it?  (standalone micro test)
// Expected if it is: loads at voffset + soffset >= num_records return 0, stores there are dropped.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const float* src, float* dst, float* out, int soff) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 256 * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t ws = __builtin_amdgcn_make_buffer_rsrc((void*)dst, 0, 256 * 4, 0x00020000);
  const int lane = threadIdx.x;
  const float v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, lane * 4, soff, 0));
  out[lane] = v;
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, 7.0f), ws, lane * 4, soff, 0);
}
int main() {
  float *src, *dst, *out;
  hipMalloc(&src, 4096 * 4);
  hipMalloc(&dst, 4096 * 4);
  hipMalloc(&out, 64 * 4);
  std::vector<float> h(4096);
  for (int i = 0; i < 4096; ++i) h[i] = 1.0f + i;
  hipMemcpy(src, h.data(), 4096 * 4, hipMemcpyHostToDevice);
  for (int soff : {0, 512, 896, 1024, 2048}) {
    hipMemset(dst, 0, 4096 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, src, dst, out, soff);
    std::vector<float> o(64), d(4096);
    hipMemcpy(o.data(), out, 64 * 4, hipMemcpyDeviceToHost);
    hipMemcpy(d.data(), dst, 4096 * 4, hipMemcpyDeviceToHost);
    int nz = 0, stored = 0, stored_oob = 0;
    for (int l = 0; l < 64; ++l) nz += o[l] != 0.f;
    for (int i = 0; i < 4096; ++i) {
      if (d[i] == 7.0f) {
        ++stored;
        if (i >= 256) ++stored_oob;
      }
    }
    printf("soffset %5d: lanes with nonzero load %2d (in-range lanes %2d), stores landed %2d (beyond the buffer: %d)\n", soff,
           nz, soff >= 1024 ? 0 : (1024 - soff) / 4 < 64 ? (1024 - soff) / 4 : 64, stored, stored_oob);
  }
  return 0;
}
