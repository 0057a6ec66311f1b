// Shared device/host helpers for the gfx950 forward-path kernels.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <cstdarg>
#include <cstdio>

#include "../../include/anemoi_amd.h"

namespace anemoi {

// ---------------------------------------------------------------- error reporting
inline char* err_buf() {
  static thread_local char buf[512] = {0};
  return buf;
}

inline int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(err_buf(), 512, fmt, ap);
  va_end(ap);
  return code;
}

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(ANEMOI_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
  return ANEMOI_OK;
}

#define ANEMOI_REQUIRE(cond, code, ...) \
  do {                                  \
    if (!(cond)) return ::anemoi::fail(code, __VA_ARGS__); \
  } while (0)

// "Has this call site prepared its kernels on the CURRENT device yet?" -- for hipFuncSetAttribute(MaxDynamicShared
// MemorySize), which is a per-device property of a kernel: one bit per device ordinal, atomics only (two threads racing
// through the first call both set the (idempotent) attribute; neither launches before it is set).
struct PerDeviceOnce {
  std::atomic<uint64_t> done_mask[4] = {};  // 256 device ordinals
  // the current device's ordinal when this call site has not prepared it yet, else -1
  int pending() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    dev &= 255;
    return (done_mask[dev >> 6].load(std::memory_order_acquire) >> (dev & 63) & 1ull) ? -1 : dev;
  }
  void done(int dev) { done_mask[dev >> 6].fetch_or(1ull << (dev & 63), std::memory_order_release); }
};

// ---------------------------------------------------------------- bf16 <-> f32
typedef uint16_t bf16_t;  // raw storage

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// round-to-nearest-even, NaN preserved (matches torch's float -> bfloat16 cast).  The compiler's own conversion
// lowers to one v_cvt_pk_bf16_f32 on gfx950 (the integer bit trick costs ~6 VALU per value).
typedef __attribute__((ext_vector_type(2))) float f32x2_cvt_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_cvt_t;

__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  const __bf16 b = (__bf16)f;
  return __builtin_bit_cast(bf16_t, b);
}

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  const f32x2_cvt_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_cvt_t));
}

template <typename T>
struct Elem;
template <>
struct Elem<float> {
  static __device__ __forceinline__ float load(const float* p) { return *p; }
  static __device__ __forceinline__ void store(float* p, float v) { *p = v; }
};
template <>
struct Elem<bf16_t> {
  static __device__ __forceinline__ float load(const bf16_t* p) { return bf16_to_f32(*p); }
  static __device__ __forceinline__ void store(bf16_t* p, float v) { *p = f32_to_bf16(v); }
};

// Vector load/store of VEC consecutive elements as f32 registers (VEC*sizeof(T) <= 16 bytes, aligned).
template <typename T, int VEC>
struct VecIO;

template <int VEC>
struct VecIO<float, VEC> {
  static __device__ __forceinline__ void load(const float* p, float (&r)[VEC]) {
    if constexpr (VEC == 4) {
      float4 t = *reinterpret_cast<const float4*>(p);
      r[0] = t.x; r[1] = t.y; r[2] = t.z; r[3] = t.w;
    } else if constexpr (VEC == 2) {
      float2 t = *reinterpret_cast<const float2*>(p);
      r[0] = t.x; r[1] = t.y;
    } else {
#pragma unroll
      for (int i = 0; i < VEC; ++i) r[i] = p[i];
    }
  }
  static __device__ __forceinline__ void store(float* p, const float (&r)[VEC]) {
    if constexpr (VEC == 8) {
      reinterpret_cast<float4*>(p)[0] = make_float4(r[0], r[1], r[2], r[3]);
      reinterpret_cast<float4*>(p)[1] = make_float4(r[4], r[5], r[6], r[7]);
    } else if constexpr (VEC == 4) {
      *reinterpret_cast<float4*>(p) = make_float4(r[0], r[1], r[2], r[3]);
    } else if constexpr (VEC == 2) {
      *reinterpret_cast<float2*>(p) = make_float2(r[0], r[1]);
    } else {
#pragma unroll
      for (int i = 0; i < VEC; ++i) p[i] = r[i];
    }
  }
};

template <int VEC>
struct VecIO<bf16_t, VEC> {
  static __device__ __forceinline__ void load(const bf16_t* p, float (&r)[VEC]) {
    if constexpr (VEC == 8) {
      uint4 t = *reinterpret_cast<const uint4*>(p);
      uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        r[2 * i] = __uint_as_float(w[i] << 16);
        r[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
      }
    } else if constexpr (VEC == 4) {
      uint2 t = *reinterpret_cast<const uint2*>(p);
      r[0] = __uint_as_float(t.x << 16); r[1] = __uint_as_float(t.x & 0xffff0000u);
      r[2] = __uint_as_float(t.y << 16); r[3] = __uint_as_float(t.y & 0xffff0000u);
    } else if constexpr (VEC == 2) {
      uint32_t t = *reinterpret_cast<const uint32_t*>(p);
      r[0] = __uint_as_float(t << 16); r[1] = __uint_as_float(t & 0xffff0000u);
    } else {
#pragma unroll
      for (int i = 0; i < VEC; ++i) r[i] = bf16_to_f32(p[i]);
    }
  }
  static __device__ __forceinline__ void store(bf16_t* p, const float (&r)[VEC]) {
    if constexpr (VEC == 8) {
      *reinterpret_cast<uint4*>(p) = make_uint4(pack_bf16x2(r[0], r[1]), pack_bf16x2(r[2], r[3]),
                                                pack_bf16x2(r[4], r[5]), pack_bf16x2(r[6], r[7]));
    } else if constexpr (VEC == 4) {
      *reinterpret_cast<uint2*>(p) = make_uint2(pack_bf16x2(r[0], r[1]), pack_bf16x2(r[2], r[3]));
    } else if constexpr (VEC == 2) {
      *reinterpret_cast<uint32_t*>(p) = pack_bf16x2(r[0], r[1]);
    } else {
#pragma unroll
      for (int i = 0; i < VEC; ++i) p[i] = f32_to_bf16(r[i]);
    }
  }
};

// ---------------------------------------------------------------- wave64 reductions
// idx = q * d + rem for a launch-uniform divisor d: a 32-bit division when the flat index space fits 31 bits (`fits32` =
// total < 2^31, so idx, d, q and rem all do) -- a 64-bit division is ~200 instructions on this ISA, and element-wise kernels
// with several of them per thread are bound by that integer work, not by memory.
__device__ __forceinline__ void fast_divmod(int64_t idx, int64_t d, bool fits32, int64_t& q, int64_t& rem) {
  if (fits32) {
    const unsigned a = (unsigned)idx, b = (unsigned)d, qq = a / b;
    q = qq;
    rem = a - qq * b;
  } else {
    q = idx / d;
    rem = idx - q * d;
  }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}

// Sum over aligned groups of `width` consecutive lanes (width = power of two <= 64); every lane of the group
// receives the total.  Up to 16 lanes this is pure DPP (quad_perm swaps, row_half_mirror, row_mirror: one VALU
// op per step, no LDS crossbar); wider groups finish with ds_bpermute-based shuffles.
template <int WIDTH>
__device__ __forceinline__ float group_sum(float v) {
  if constexpr (WIDTH >= 2) v += dpp_f32<0xB1>(v);   // quad_perm [1,0,3,2]: lane ^ 1
  if constexpr (WIDTH >= 4) v += dpp_f32<0x4E>(v);   // quad_perm [2,3,0,1]: lane ^ 2
  if constexpr (WIDTH >= 8) v += dpp_f32<0x141>(v);  // row_half_mirror: the other quad of the 8-lane half row
  if constexpr (WIDTH >= 16) v += dpp_f32<0x140>(v); // row_mirror: the other half of the 16-lane row
  if constexpr (WIDTH >= 32) v += __shfl_xor(v, 16, 64);
  if constexpr (WIDTH >= 64) v += __shfl_xor(v, 32, 64);
  return v;
}

// erf(x) by Abramowitz & Stegun 7.1.26: |error| <= 1.5e-7 over the whole real line (f32 roundoff class), one
// v_exp + one v_rcp + 6 FMAs instead of the ~40-instruction branchy libm erff.  GELU inherits < 2e-7 * |x|.
__device__ __forceinline__ float fast_erf(float x) {
  const float ax = fabsf(x);
  const float t = __frcp_rn(fmaf(0.3275911f, ax, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float r = 1.0f - p * t * __expf(-ax * ax);
  return copysignf(r, x);
}

__device__ __forceinline__ float act_apply(float x, int act) {
  switch (act) {
    case ANEMOI_ACT_GELU: return 0.5f * x * (1.0f + fast_erf(x * 0.70710678118654752440f));
    case ANEMOI_ACT_SILU: return x * __frcp_rn(1.0f + __expf(-x));
    case ANEMOI_ACT_RELU: return x > 0.f ? x : 0.f;
    default: return x;
  }
}

inline hipStream_t as_stream(anemoi_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// Dropout of the edge attention weights (reference layers/conv.py:140, `dropout(alpha, p, training)` on alpha [E, H]): a
// counter-based keep decision per (edge position in the destination-sorted CSR, head) -- no mask tensor, the backward kernels
// rebuild it from the seed.  x = e G1 ^ h G2 ^ seed, one multiply between two fold-downs, the top 15 bits against p 2^15
// (tests/test_gpu_training.py::_edge_dropout_keep_mask restates it).  `seed_dev`: optional device word added to the seed by
// the kernel (runtime.DeviceDropout: what a captured training step advances itself), as in csrc/attention.hip.
struct EdgeDropout {
  uint32_t thr15;    // keep  <=>  15 hash bits >= thr15  (p 2^15; 0: no dropout, 0x8000: everything dropped)
  uint32_t seed;
  float keep_scale;  // 1 / (1 - p); 0 when p >= 1
  const uint32_t* seed_dev;
};

inline EdgeDropout make_edge_dropout(float p, uint32_t seed, const void* seed_dev) {
  EdgeDropout dr;
  const double t = p <= 0.f ? 0.0 : (double)p * 32768.0 + 0.5;
  dr.thr15 = p >= 1.0f ? 0x8000u : (uint32_t)(t > 32768.0 ? 32768.0 : t);
  dr.seed = seed;
  dr.keep_scale = (p > 0.f && p < 1.0f) ? 1.0f / (1.0f - p) : (p >= 1.0f ? 0.f : 1.0f);
  dr.seed_dev = static_cast<const uint32_t*>(seed_dev);
  return dr;
}

__device__ __forceinline__ uint32_t edge_dropout_seed(const EdgeDropout& dr) {
  return dr.seed_dev != nullptr ? dr.seed + __builtin_nontemporal_load(dr.seed_dev) : dr.seed;
}

__device__ __forceinline__ float edge_dropout_keep(const EdgeDropout& dr, uint32_t seed, int64_t e, int head) {
  uint32_t x = (uint32_t)e * 0x9E3779B1u ^ (uint32_t)head * 0x85EBCA77u ^ seed;
  x ^= x >> 16;
  x *= 0x7feb352du;
  x ^= x >> 15;
  return (x >> 17) >= dr.thr15 ? dr.keep_scale : 0.f;
}

}  // namespace anemoi
