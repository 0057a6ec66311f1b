// Helpers shared by the edge-phase kernels (edge_attention.hip: forward, edge_backward.hip: backward): raw 16-byte row
// slices and their unpacking, the lane's q . k dot product (packed bf16 pairs through v_dot2_f32_bf16), streaming
// (nontemporal) accesses, the lane-shared attribute layout of the folded kernels.
#pragma once
#include "common.hpp"

namespace anemoi {

template <typename T, int VEC>
struct RawVec;
template <>
struct RawVec<float, 4> { using type = float4; };
template <>
struct RawVec<float, 2> { using type = float2; };
template <>
struct RawVec<float, 1> { using type = float; };
template <>
struct RawVec<bf16_t, 8> { using type = uint4; };
template <>
struct RawVec<bf16_t, 4> { using type = uint2; };
template <>
struct RawVec<bf16_t, 2> { using type = uint32_t; };
template <>
struct RawVec<bf16_t, 1> { using type = uint16_t; };

template <typename T, int VEC>
__device__ __forceinline__ void unpack(const typename RawVec<T, VEC>::type& raw, float (&r)[VEC]) {
  VecIO<T, VEC>::load(reinterpret_cast<const T*>(&raw), r);
}

template <typename T, int VEC>
struct QK;  // dot product of the lane's q and k slices

template <int VEC>
struct QK<float, VEC> {
  using Raw = typename RawVec<float, VEC>::type;
  float q[VEC];
  __device__ __forceinline__ void set(const float (&qf)[VEC]) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) q[i] = qf[i];
  }
  __device__ __forceinline__ void set_raw(const uint32_t (&w)[VEC]) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) q[i] = __uint_as_float(w[i]);
  }
  __device__ __forceinline__ float dot(const Raw& kr) const {
    float kk[VEC];
    unpack<float, VEC>(kr, kk);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VEC; ++i) s = fmaf(q[i], kk[i], s);
    return s;
  }
};

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;

template <int VEC>
struct QK<bf16_t, VEC> {
  using Raw = typename RawVec<bf16_t, VEC>::type;
  static_assert(VEC % 2 == 0, "bf16 fast path packs channel pairs");
  uint32_t q[VEC / 2];  // q stays packed: v_dot2c_f32_bf16 multiplies bf16 pairs exactly and accumulates in f32
  __device__ __forceinline__ void set(const float (&qf)[VEC]) {
#pragma unroll
    for (int i = 0; i < VEC / 2; ++i) q[i] = pack_bf16x2(qf[2 * i], qf[2 * i + 1]);
  }
  // the packed words as they sit in memory: no conversion, so the load's s_waitcnt lands at the first dot product
  __device__ __forceinline__ void set_raw(const uint32_t (&w)[VEC / 2]) {
#pragma unroll
    for (int i = 0; i < VEC / 2; ++i) q[i] = w[i];
  }
  __device__ __forceinline__ float dot(const Raw& kr) const {
    const uint32_t* kw = reinterpret_cast<const uint32_t*>(&kr);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VEC / 2; ++i) {
      uint32_t a = q[i], b = kw[i];
      s = __builtin_amdgcn_fdot2_f32_bf16(*reinterpret_cast<bf16x2_t*>(&a), *reinterpret_cast<bf16x2_t*>(&b), s,
                                          false);
    }
    return s;
  }
};

// 16-byte streaming accesses: data that is touched exactly once per launch
template <typename T, int VEC>
__device__ __forceinline__ void load_stream(const T* p, float (&r)[VEC]) {
  if constexpr (sizeof(T) * VEC == 16) {
    typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
    const u32x4_t t = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
    VecIO<T, VEC>::load(reinterpret_cast<const T*>(&t), r);
  } else {
    VecIO<T, VEC>::load(p, r);
  }
}
template <typename T, int VEC>
__device__ __forceinline__ void store_stream(T* p, const float (&r)[VEC]) {
  if constexpr (sizeof(T) * VEC == 16) {
    typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
    u32x4_t t;
    VecIO<T, VEC>::store(reinterpret_cast<T*>(&t), r);
    __builtin_nontemporal_store(t, reinterpret_cast<u32x4_t*>(p));
  } else {
    VecIO<T, VEC>::store(p, r);
  }
}

// The attribute part of the score (u . a) and of the output (sum alpha a) is the same for all LPH lanes of a head:
// the lanes SHARE it -- lane r of a head owns the APL attributes [r * APL, r * APL + APL) (one 8/16-byte load per
// edge), its partial u . a joins the lane's partial q . k before the head reduction (which is needed anyway), and it
// accumulates only its own attributes.  12 + 12 FMAs per edge and lane become 2 + 2 (UP = 12, 8 lanes per head):
// the kernel was VALU-bound (~58 VALU per edge and wave, 0.10 of its 0.17 ms on the mesh graph).
constexpr int attrs_per_lane(int up, int lph) {  // smallest divisor of UP in {2, 4, 8, 12, 16} covering UP with LPH lanes
  const int raw = (up + lph - 1) / lph;
  for (int a : {2, 4, 8, 12, 16})
    if (a >= raw && up % a == 0) return a;
  return up;
}

// Raw (unconverted) words of N consecutive elements: what a prefetched operand is carried in from one destination to the
// next -- converting at load time would pin the s_waitcnt to the load instead of to the first use.
template <typename T, int N>
struct RawWords {
  static constexpr int W = (N * (int)sizeof(T) + 3) / 4;
  uint32_t w[W];
  // ``base`` is wave-uniform (SGPR pair), ``off`` the lane's byte offset: the access compiles to the saddr + voffset
  // form, so no 64-bit per-lane pointer is kept alive (this kernel lives at the 96-VGPR edge of 5 waves per SIMD)
  __device__ __forceinline__ void load(const char* base, uint32_t off, bool nt) {
    const T* p = reinterpret_cast<const T*>(base + off);
    if constexpr (W % 4 == 0) {
      typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
#pragma unroll
      for (int i = 0; i < W / 4; ++i) {
        const u32x4_t t = nt ? __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p) + i)
                             : reinterpret_cast<const u32x4_t*>(p)[i];
        w[4 * i] = t.x; w[4 * i + 1] = t.y; w[4 * i + 2] = t.z; w[4 * i + 3] = t.w;
      }
    } else if constexpr (W % 2 == 0) {
#pragma unroll
      for (int i = 0; i < W / 2; ++i) {
        const uint2 t = reinterpret_cast<const uint2*>(p)[i];
        w[2 * i] = t.x; w[2 * i + 1] = t.y;
      }
    } else {
      static_assert(N * sizeof(T) % 4 == 0, "whole words");
#pragma unroll
      for (int i = 0; i < W; ++i) w[i] = reinterpret_cast<const uint32_t*>(p)[i];
    }
  }
  __device__ __forceinline__ void get(float (&r)[N]) const {
    if constexpr (sizeof(T) == 4) {
#pragma unroll
      for (int i = 0; i < N; ++i) r[i] = __uint_as_float(w[i]);
    } else {
#pragma unroll
      for (int i = 0; i < N; ++i) r[i] = __uint_as_float((i & 1) ? (w[i >> 1] & 0xffff0000u) : (w[i >> 1] << 16));
    }
  }
};

// N values -> the 32-bit words they occupy in memory (bf16: round-to-nearest-even pairs, low half first), built in
// registers.  (Not VecIO::store through a reinterpret_cast of a local: that writes the local through another type, and the
// optimizer is free to order such a store behind the read that hands the local to a buffer-store builtin -- observed in the
// scheduled edge kernel's UP = 16 instantiation as stale words in single lanes.)
template <typename T, int N>
__device__ __forceinline__ void pack_words(const float (&r)[N], uint32_t (&w)[RawWords<T, N>::W]) {
  if constexpr (sizeof(T) == 4) {
#pragma unroll
    for (int i = 0; i < N; ++i) w[i] = __float_as_uint(r[i]);
  } else {
    static_assert(N % 2 == 0, "bf16 values are packed in pairs");
#pragma unroll
    for (int i = 0; i < N / 2; ++i) w[i] = pack_bf16x2(r[2 * i], r[2 * i + 1]);
  }
}

}  // namespace anemoi
