// Edge-side kernels of the GNN (edge-MLP message passing) processor, K5 of SURVEY.md section 2a.
//
// The reference builds cat[x_i, x_j, e] ([E, 3C]) and runs the edge MLP on it (layers/conv.py:68-71).  The first
// Linear is linear in each of the three blocks, so it is evaluated as
//     W1 [x_i | x_j | e] = (W1a x)_i + (W1b x)_j + W1c e
// two node-level GEMMs ([N, C] each, done as one GEMM with 2C outputs) + one edge-level GEMM, and the gather-add
// below; the [E, 3C] concatenation and the [E, C] gathers of x_i / x_j never exist in HBM.
//   anemoi_gather_add_act : out[e] = act(t[e] + p_dst[dst[e]] + p_src[src[e]])          (HBM bound, 16 B per lane)
//   anemoi_segment_sum    : out[i] = sum_{e in CSR row i} v[e]  (scatter-sum over destinations, layers/conv.py:73-76;
//                           edges are in destination-sorted order, so a row is a contiguous range: no atomics,
//                           f32 accumulation in CSR order = the reference's scatter_add_ order)
#include "common.hpp"

namespace anemoi {

template <typename T, int VEC>
__global__ __launch_bounds__(256) void gather_add_act_kernel(const T* __restrict__ t, int64_t ldt,
                                                             const T* __restrict__ pd, int64_t ldpd,
                                                             const T* __restrict__ ps, int64_t ldps,
                                                             const int32_t* __restrict__ dst,
                                                             const int32_t* __restrict__ src, T* __restrict__ out,
                                                             int64_t ldo, int64_t n_edges, int C, int act) {
  const int lane = threadIdx.x & 63;
  const int slices = (C + 64 * VEC - 1) / (64 * VEC);
  const int64_t units = n_edges * slices;
  for (int64_t unit = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); unit < units; unit += (int64_t)gridDim.x * 4) {
    const int64_t e = unit / slices;
    const int c = ((int)(unit - e * slices) * 64 + lane) * VEC;
    if (c >= C) continue;
    const int64_t i = dst[e], j = src[e];
    float a[VEC], b[VEC], d[VEC], o[VEC];
    VecIO<T, VEC>::load(t + e * ldt + c, a);
    VecIO<T, VEC>::load(pd + i * ldpd + c, b);
    VecIO<T, VEC>::load(ps + j * ldps + c, d);
#pragma unroll
    for (int k = 0; k < VEC; ++k) o[k] = act_apply(a[k] + b[k] + d[k], act);
    VecIO<T, VEC>::store(out + e * ldo + c, o);
  }
}

template <typename T, int VEC>
__global__ __launch_bounds__(256) void segment_sum_kernel(const T* __restrict__ v, int64_t ldv,
                                                          const int32_t* __restrict__ rowptr, T* __restrict__ out,
                                                          int64_t ldo, int64_t n_dst, int C,
                                                          const T* __restrict__ x, int64_t ldx) {
  // x != nullptr (anemoi_segment_sum_cat): out is [n_dst, >= 2C] = [x | sums]: the lane also copies its piece of row i of x
  const int lane = threadIdx.x & 63;
  const int slices = (C + 64 * VEC - 1) / (64 * VEC);
  const int64_t units = n_dst * slices;
  for (int64_t unit = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); unit < units; unit += (int64_t)gridDim.x * 4) {
    const int64_t i = unit / slices;
    const int c = ((int)(unit - i * slices) * 64 + lane) * VEC;
    if (c >= C) continue;
    float acc[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
    const int e0 = rowptr[i], e1 = rowptr[i + 1];
    for (int e = e0; e < e1; ++e) {
      float r[VEC];
      VecIO<T, VEC>::load(v + (int64_t)e * ldv + c, r);
#pragma unroll
      for (int k = 0; k < VEC; ++k) acc[k] += r[k];
    }
    if (x != nullptr) {
      float r[VEC];
      VecIO<T, VEC>::load(x + i * ldx + c, r);
      VecIO<T, VEC>::store(out + i * ldo + c, r);
      VecIO<T, VEC>::store(out + i * ldo + C + c, acc);
    } else {
      VecIO<T, VEC>::store(out + i * ldo + c, acc);
    }
  }
}

static inline unsigned wave_grid(int64_t units) {
  int64_t blocks = (units + 3) / 4;
  if (blocks > 256 * 32) blocks = 256 * 32;
  if (blocks < 1) blocks = 1;
  return (unsigned)blocks;
}

template <typename T>
static bool vec_ok(int C, std::initializer_list<int64_t> lds, std::initializer_list<const void*> ptrs) {
  constexpr int V = 16 / sizeof(T);
  if (C % V != 0) return false;
  for (int64_t ld : lds)
    if (ld % V != 0) return false;
  for (const void* p : ptrs)
    if ((uintptr_t)p % 16 != 0) return false;
  return true;
}

}  // namespace anemoi

using namespace anemoi;

extern "C" {

int anemoi_gather_add_act(int dtype, const void* t, int64_t ldt, const void* p_dst, int64_t ldpd, const void* p_src,
                          int64_t ldps, const int32_t* dst, const int32_t* src, void* out, int64_t ldo,
                          int64_t n_edges, int C, int act, anemoi_stream_t stream) {
  ANEMOI_REQUIRE(n_edges >= 0 && C > 0, ANEMOI_ERR_INVALID, "anemoi_gather_add_act: bad shape");
  if (n_edges == 0) return ANEMOI_OK;
  ANEMOI_REQUIRE(t && p_dst && p_src && dst && src && out, ANEMOI_ERR_INVALID, "anemoi_gather_add_act: null pointer");
  ANEMOI_REQUIRE(ldt >= C && ldpd >= C && ldps >= C && ldo >= C, ANEMOI_ERR_INVALID,
                 "anemoi_gather_add_act: leading dimension smaller than C");
  ANEMOI_REQUIRE(act >= ANEMOI_ACT_NONE && act <= ANEMOI_ACT_RELU, ANEMOI_ERR_INVALID, "anemoi_gather_add_act: act");
  hipStream_t st = as_stream(stream);
#define GAA(T, V)                                                                                                    \
  hipLaunchKernelGGL((gather_add_act_kernel<T, V>), dim3(wave_grid(n_edges * ((C + 64 * V - 1) / (64 * V)))),         \
                     dim3(256), 0, st, static_cast<const T*>(t), ldt, static_cast<const T*>(p_dst), ldpd,            \
                     static_cast<const T*>(p_src), ldps, dst, src, static_cast<T*>(out), ldo, n_edges, C, act)
  if (dtype == ANEMOI_F32) {
    if (vec_ok<float>(C, {ldt, ldpd, ldps, ldo}, {t, p_dst, p_src, out})) GAA(float, 4);
    else GAA(float, 1);
  } else if (dtype == ANEMOI_BF16) {
    if (vec_ok<bf16_t>(C, {ldt, ldpd, ldps, ldo}, {t, p_dst, p_src, out})) GAA(bf16_t, 8);
    else GAA(bf16_t, 1);
  } else {
    return fail(ANEMOI_ERR_UNSUPPORTED, "anemoi_gather_add_act: dtype %d", dtype);
  }
#undef GAA
  return check_launch("anemoi_gather_add_act");
}

static int segment_sum_impl(const char* who, int dtype, const void* v, int64_t ldv, const int32_t* rowptr, const void* x,
                            int64_t ldx, void* out, int64_t ldo, int64_t n_dst, int C, anemoi_stream_t stream) {
  ANEMOI_REQUIRE(n_dst >= 0 && C > 0, ANEMOI_ERR_INVALID, "%s: bad shape", who);
  if (n_dst == 0) return ANEMOI_OK;
  ANEMOI_REQUIRE(v && rowptr && out, ANEMOI_ERR_INVALID, "%s: null pointer", who);
  ANEMOI_REQUIRE(ldv >= C && ldo >= (x != nullptr ? 2 : 1) * (int64_t)C && (x == nullptr || ldx >= C), ANEMOI_ERR_INVALID,
                 "%s: leading dimension too small", who);
  hipStream_t st = as_stream(stream);
#define SEG(T, V)                                                                                                   \
  hipLaunchKernelGGL((segment_sum_kernel<T, V>), dim3(wave_grid(n_dst * ((C + 64 * V - 1) / (64 * V)))), dim3(256), \
                     0, st, static_cast<const T*>(v), ldv, rowptr, static_cast<T*>(out), ldo, n_dst, C,             \
                     static_cast<const T*>(x), ldx)
  if (dtype == ANEMOI_F32) {
    if (vec_ok<float>(C, {ldv, ldo, x != nullptr ? ldx : 0}, {v, out, x})) SEG(float, 4);
    else SEG(float, 1);
  } else if (dtype == ANEMOI_BF16) {
    if (vec_ok<bf16_t>(C, {ldv, ldo, x != nullptr ? ldx : 0}, {v, out, x})) SEG(bf16_t, 8);
    else SEG(bf16_t, 1);
  } else {
    return fail(ANEMOI_ERR_UNSUPPORTED, "%s: dtype %d", who, dtype);
  }
#undef SEG
  return check_launch(who);
}

int anemoi_segment_sum(int dtype, const void* v, int64_t ldv, const int32_t* rowptr, void* out, int64_t ldo,
                       int64_t n_dst, int C, anemoi_stream_t stream) {
  return segment_sum_impl("anemoi_segment_sum", dtype, v, ldv, rowptr, nullptr, 0, out, ldo, n_dst, C, stream);
}

int anemoi_segment_sum_cat(int dtype, const void* v, int64_t ldv, const int32_t* rowptr, const void* x, int64_t ldx,
                           void* out, int64_t ldo, int64_t n_dst, int C, anemoi_stream_t stream) {
  ANEMOI_REQUIRE(x != nullptr || n_dst == 0, ANEMOI_ERR_INVALID, "anemoi_segment_sum_cat: null pointer");
  return segment_sum_impl("anemoi_segment_sum_cat", dtype, v, ldv, rowptr, x, ldx, out, ldo, n_dst, C, stream);
}

}  // extern "C"
