// HBM-bound row kernels of the forward path: LayerNorm, input assembly, edge-attribute CSR gather,
// dtype conversion / K-padding, elementwise add and the prognostic residual.
//
// All of them are streaming kernels: one pass over the data, 16-byte accesses per lane where the
// layout allows, one wave64 per row (LayerNorm) or a flat grid-stride mapping.
#include <cstdlib>

#include "common.hpp"

namespace anemoi {

// ---------------------------------------------------------------------------------------------
// LayerNorm: one wave per row, the row cached in registers (ITEMS x VEC values per lane).
// Two-pass statistics in f32 (mean, then centred sum of squares) -- same formula as ATen's CPU
// kernel up to summation order.
// ---------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ float round_to(float v);
template <>
__device__ __forceinline__ float round_to<float>(float v) { return v; }
template <>
__device__ __forceinline__ float round_to<bf16_t>(float v) { return bf16_to_f32(f32_to_bf16(v)); }

template <typename T, int VEC, int ITEMS>
__global__ __launch_bounds__(256) void layer_norm_kernel(const T* __restrict__ x, int64_t ldx,
                                                         const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, T* __restrict__ y,
                                                         int64_t ldy, int64_t rows, int C, float eps,
                                                         float2* __restrict__ stats, const T* __restrict__ res = nullptr,
                                                         int64_t ldr = 0) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const T* xr = x + row * ldx;
  float v[ITEMS][VEC];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    const int c = (i * 64 + lane) * VEC;
    if (c < C) {
      VecIO<T, VEC>::load(xr + c, v[i]);
#pragma unroll
      for (int j = 0; j < VEC; ++j) s += v[i][j];
    } else {
#pragma unroll
      for (int j = 0; j < VEC; ++j) v[i][j] = 0.f;
    }
  }
  const float mean = wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    const int c = (i * 64 + lane) * VEC;
    if (c < C) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const float d = v[i][j] - mean;
        q += d * d;
      }
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
  if (stats != nullptr && lane == 0) stats[row] = make_float2(rstd, -mean * rstd);  // as row_stats_kernel leaves them
  T* yr = y + row * ldy;
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    const int c = (i * 64 + lane) * VEC;
    if (c < C) {
      float g[VEC], b[VEC], o[VEC];
      VecIO<float, VEC>::load(gamma + c, g);
      VecIO<float, VEC>::load(beta + c, b);
#pragma unroll
      for (int j = 0; j < VEC; ++j) o[j] = (v[i][j] - mean) * rstd * g[j] + b[j];
      if (res != nullptr) {  // LayerNorm(x) + residual, rounded as two separate operations would round it
        float r[VEC];
        VecIO<T, VEC>::load(res + row * ldr + c, r);
#pragma unroll
        for (int j = 0; j < VEC; ++j) o[j] = round_to<T>(o[j]) + r[j];
      }
      VecIO<T, VEC>::store(yr + c, o);
    }
  }
}

// Fallback for very wide or oddly aligned rows: three passes over the (cache-resident) row.
template <typename T>
__global__ __launch_bounds__(256) void layer_norm_generic_kernel(const T* __restrict__ x, int64_t ldx,
                                                                 const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta,
                                                                 T* __restrict__ y, int64_t ldy, int64_t rows,
                                                                 int C, float eps, float2* __restrict__ stats,
                                                                 const T* __restrict__ res = nullptr, int64_t ldr = 0) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const T* xr = x + row * ldx;
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += Elem<T>::load(xr + c);
  const float mean = wave_sum(s) / (float)C;
  float q = 0.f;
  for (int c = lane; c < C; c += 64) {
    const float d = Elem<T>::load(xr + c) - mean;
    q += d * d;
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
  if (stats != nullptr && lane == 0) stats[row] = make_float2(rstd, -mean * rstd);
  T* yr = y + row * ldy;
  for (int c = lane; c < C; c += 64) {
    float o = (Elem<T>::load(xr + c) - mean) * rstd * gamma[c] + beta[c];
    if (res != nullptr) o = round_to<T>(o) + Elem<T>::load(res + row * ldr + c);
    Elem<T>::store(yr + c, o);
  }
}

template <typename T>
static int layer_norm_launch(const void* x, int64_t ldx, const float* gamma, const float* beta, void* y, int64_t ldy,
                             int64_t rows, int C, float eps, hipStream_t st, float2* stats = nullptr,
                             const void* residual = nullptr, int64_t ldr = 0) {
  constexpr int VMAX = 16 / sizeof(T);
  const T* xp = static_cast<const T*>(x);
  T* yp = static_cast<T*>(y);
  dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  const T* rp = static_cast<const T*>(residual);
  const bool aligned = (C % VMAX == 0) && (ldx % VMAX == 0) && (ldy % VMAX == 0) &&
                       ((uintptr_t)x % 16 == 0) && ((uintptr_t)y % 16 == 0) && ((uintptr_t)gamma % 16 == 0) &&
                       ((uintptr_t)beta % 16 == 0) && (residual == nullptr || (ldr % VMAX == 0 && (uintptr_t)residual % 16 == 0));
  const int per_pass = 64 * VMAX;
  const int items = (C + per_pass - 1) / per_pass;
#define LN_CASE(I)                                                                                       \
  case I:                                                                                                \
    hipLaunchKernelGGL((layer_norm_kernel<T, VMAX, I>), grid, block, 0, st, xp, ldx, gamma, beta, yp, ldy, \
                       rows, C, eps, stats, rp, ldr);                                                    \
    break;
  if (aligned && items <= 8) {
    switch (items) {
      LN_CASE(1) LN_CASE(2) LN_CASE(3) LN_CASE(4) LN_CASE(5) LN_CASE(6) LN_CASE(7) LN_CASE(8)
    }
  } else {
    hipLaunchKernelGGL((layer_norm_generic_kernel<T>), grid, block, 0, st, xp, ldx, gamma, beta, yp, ldy, rows, C,
                       eps, stats, rp, ldr);
  }
#undef LN_CASE
  return check_launch("anemoi_layer_norm");
}

// ---------------------------------------------------------------------------------------------
// LayerNorm statistics only: stats[r] = { rstd_r, -mean_r * rstd_r }.  The normalisation itself is folded into the
// Linear that consumes the LayerNorm (anemoi_linear_ln): one read of x instead of a read + a write + a re-read.
// Same arithmetic as layer_norm_kernel (row in registers, two-pass f32 statistics).
// ---------------------------------------------------------------------------------------------
template <typename T, int VEC, int ITEMS>
__global__ __launch_bounds__(256) void row_stats_kernel(const T* __restrict__ x, int64_t ldx, float2* __restrict__ stats,
                                                        int64_t rows, int C, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const T* xr = x + row * ldx;
  float v[ITEMS][VEC];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    const int c = (i * 64 + lane) * VEC;
    if (c < C) {
      VecIO<T, VEC>::load(xr + c, v[i]);
#pragma unroll
      for (int j = 0; j < VEC; ++j) s += v[i][j];
    } else {
#pragma unroll
      for (int j = 0; j < VEC; ++j) v[i][j] = 0.f;
    }
  }
  const float mean = wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    const int c = (i * 64 + lane) * VEC;
    if (c < C) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const float d = v[i][j] - mean;
        q += d * d;
      }
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
  if (lane == 0) stats[row] = make_float2(rstd, -mean * rstd);
}

template <typename T>
__global__ __launch_bounds__(256) void row_stats_generic_kernel(const T* __restrict__ x, int64_t ldx,
                                                                float2* __restrict__ stats, int64_t rows, int C,
                                                                float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const T* xr = x + row * ldx;
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += Elem<T>::load(xr + c);
  const float mean = wave_sum(s) / (float)C;
  float q = 0.f;
  for (int c = lane; c < C; c += 64) {
    const float d = Elem<T>::load(xr + c) - mean;
    q += d * d;
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
  if (lane == 0) stats[row] = make_float2(rstd, -mean * rstd);
}

template <typename T>
static int row_stats_launch(const void* x, int64_t ldx, float* stats, int64_t rows, int C, float eps, hipStream_t st) {
  constexpr int VMAX = 16 / sizeof(T);
  const T* xp = static_cast<const T*>(x);
  float2* sp = reinterpret_cast<float2*>(stats);
  dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  const bool aligned = (C % VMAX == 0) && (ldx % VMAX == 0) && ((uintptr_t)x % 16 == 0);
  const int per_pass = 64 * VMAX;
  const int items = (C + per_pass - 1) / per_pass;
#define RS_CASE(I)                                                                                          \
  case I:                                                                                                   \
    hipLaunchKernelGGL((row_stats_kernel<T, VMAX, I>), grid, block, 0, st, xp, ldx, sp, rows, C, eps);      \
    break;
  if (aligned && items <= 8) {
    switch (items) {
      RS_CASE(1) RS_CASE(2) RS_CASE(3) RS_CASE(4) RS_CASE(5) RS_CASE(6) RS_CASE(7) RS_CASE(8)
    }
  } else {
    hipLaunchKernelGGL((row_stats_generic_kernel<T>), grid, block, 0, st, xp, ldx, sp, rows, C, eps);
  }
#undef RS_CASE
  return check_launch("anemoi_row_stats");
}

// ---------------------------------------------------------------------------------------------
// Input assembly: out[(b,ens,g), :] = [x[b,:,ens,g,:] (time-major) | latlons[g] | trainable[g] | 0]
// One thread per output element; consecutive threads write consecutive columns (coalesced store),
// reads of x are contiguous runs of V floats.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void assemble_nodes_kernel(const float* __restrict__ x, int B, int T_, int Ens,
                                                             int64_t G, int V, const float* __restrict__ latlons,
                                                             int n_ll, const float* __restrict__ trainable, int n_tr,
                                                             T* __restrict__ out, int64_t ldo,
                                                             const float* __restrict__ in_mul,
                                                             const float* __restrict__ in_add,
                                                             const int64_t* __restrict__ rows, int64_t n_rows) {
  const int64_t total = (rows != nullptr ? n_rows : (int64_t)B * Ens * G) * ldo;
  const int tv = T_ * V;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = idx / ldo;
    const int c = (int)(idx - row * ldo);
    const int64_t g = rows != nullptr ? rows[row] : row % G;
    float val = 0.f;
    if (c < tv) {
      const int64_t be = rows != nullptr ? 0 : row / G;
      const int e = (int)(be % Ens);
      const int64_t b = be / Ens;
      const int t = c / V, v = c - t * V;
      val = x[((((b * T_ + t) * Ens + e) * G) + g) * V + v];
      if (in_mul != nullptr) val = val * in_mul[v] + in_add[v];  // InputNormalizer.transform folded into the read
    } else if (c < tv + n_ll) {
      val = latlons[g * n_ll + (c - tv)];
    } else if (c < tv + n_ll + n_tr) {
      val = trainable[g * n_tr + (c - tv - n_ll)];
    }
    Elem<T>::store(out + idx, val);
  }
}

// The same with EIGHT consecutive output columns per thread (ldo % 8 == 0): one 16-byte (bf16) / two 16-byte (f32) stores
// instead of eight 2-byte ones, one row / column division per eight elements (the element-wise form above reached 1.2 TB/s
// on the 542 080 x 256 data-grid matrix of config 3, this one is bound by the reads of x).
// IDX32: every flat index fits 31 bits -- 32-bit divisions (a 64-bit division is ~200 instructions on this ISA; with three
// of them per thread the kernel was bound by integer ALU work, not by memory: 2.3 TB/s on the data-grid matrix).
template <typename T, bool IDX32>
__global__ __launch_bounds__(256) void assemble_nodes8_kernel(const float* __restrict__ x, int B, int T_, int Ens,
                                                              int64_t G, int V, const float* __restrict__ latlons,
                                                              int n_ll, const float* __restrict__ trainable, int n_tr,
                                                              T* __restrict__ out, int ldo8,
                                                              const float* __restrict__ in_mul,
                                                              const float* __restrict__ in_add,
                                                              const int64_t* __restrict__ rows, int64_t n_rows) {
  // rows != nullptr (anemoi_assemble_node_rows, B = Ens = 1): output row i is grid node rows[i]
  const int64_t total = (rows != nullptr ? n_rows : (int64_t)B * Ens * G) * ldo8;
  const int tv = T_ * V;
  const bool pairs = x != nullptr && V % 2 == 0 && ((uintptr_t)x & 7) == 0;
  using I = typename std::conditional<IDX32, unsigned, int64_t>::type;
  const bool one_be = B * Ens == 1;
  for (I idx = (I)blockIdx.x * blockDim.x + threadIdx.x; idx < (I)total; idx += (I)gridDim.x * blockDim.x) {
    const I row = idx / (I)ldo8;
    const int c0 = (int)(idx - row * (I)ldo8) * 8;
    int64_t g, b = 0;
    int e = 0;
    if (rows != nullptr) g = rows[row];
    else if (one_be) g = (int64_t)row;
    else {
      const I be = row / (I)G;
      g = (int64_t)(row - be * (I)G);
      e = (int)(be % (I)Ens);
      b = (int64_t)(be / (I)Ens);
    }
    float val[8];
    int t = c0 / V, v = c0 - t * V;  // (time, variable) of column c0; advanced without further divisions
    if (pairs && c0 + 8 <= tv) {
      // V even, x 8-byte aligned: the eight columns are four (v, v + 1) pairs that never straddle a time slice -- four
      // 8-byte loads instead of eight 4-byte ones (same values)
#pragma unroll
      for (int i = 0; i < 8; i += 2) {
        const float2 p = *reinterpret_cast<const float2*>(x + ((((b * T_ + t) * Ens + e) * G) + g) * V + v);
        val[i] = p.x;
        val[i + 1] = p.y;
        if (in_mul != nullptr) {
          val[i] = val[i] * in_mul[v] + in_add[v];
          val[i + 1] = val[i + 1] * in_mul[v + 1] + in_add[v + 1];
        }
        v += 2;
        if (v == V) {
          v = 0;
          ++t;
        }
      }
      VecIO<T, 8>::store(out + ((int64_t)row * ldo8 + (c0 >> 3)) * 8, val);
      continue;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = c0 + i;
      float r = 0.f;
      if (c < tv) {
        r = x[((((b * T_ + t) * Ens + e) * G) + g) * V + v];
        if (in_mul != nullptr) r = r * in_mul[v] + in_add[v];  // InputNormalizer.transform folded into the read
        if (++v == V) {
          v = 0;
          ++t;
        }
      } else if (c < tv + n_ll) {
        r = latlons[g * n_ll + (c - tv)];
      } else if (c < tv + n_ll + n_tr) {
        r = trainable[g * n_tr + (c - tv - n_ll)];
      }
      val[i] = r;
    }
    VecIO<T, 8>::store(out + ((int64_t)row * ldo8 + (c0 >> 3)) * 8, val);
  }
}

// out[e, :] = [a0[perm[e] % rows0] | a1[perm[e] % rows0] | 0]
__global__ __launch_bounds__(256) void edge_attr_csr_kernel(const float* __restrict__ a0, int d0,
                                                            const float* __restrict__ a1, int d1, int64_t rows0,
                                                            const int32_t* __restrict__ perm, float* __restrict__ out,
                                                            int ld, int one_col, int64_t n_edges) {
  const int64_t total = n_edges * ld;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t e = idx / ld;
    const int c = (int)(idx - e * ld);
    const int64_t src = (int64_t)perm[e] % rows0;
    float val = 0.f;
    if (c < d0) val = a0[src * d0 + c];
    else if (c < d0 + d1) val = a1[src * d1 + (c - d0)];
    else if (c == one_col) val = 1.0f;
    out[idx] = val;
  }
}

template <typename S, typename D>
__global__ __launch_bounds__(256) void convert_pad_kernel(const S* __restrict__ src, int64_t ld_src,
                                                          D* __restrict__ dst, int64_t ld_dst, int64_t rows,
                                                          int cols) {
  const int64_t total = rows * ld_dst;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = idx / ld_dst;
    const int c = (int)(idx - r * ld_dst);
    const float val = c < cols ? Elem<S>::load(src + r * ld_src + c) : 0.f;
    Elem<D>::store(dst + idx, val);
  }
}

// out[r, c] = alpha * s[r] * x[r, c]   (s f32 per row; the result in x's dtype)
template <typename T>
__global__ __launch_bounds__(256) void row_scale_kernel(const T* x, int64_t ldx, const float* __restrict__ s,
                                                        float alpha, T* out, int64_t ldo, int64_t rows,
                                                        int cols) {  // x == out is legal (in-place scaling): no __restrict__
  constexpr int VEC = 16 / sizeof(T);
  const int64_t per_row = (cols + VEC - 1) / VEC;
  const int64_t total = rows * per_row;
  const bool vec_ok = cols % VEC == 0 && ldx % VEC == 0 && ldo % VEC == 0 && (uintptr_t)x % 16 == 0 && (uintptr_t)out % 16 == 0;
  const bool fits32 = total < ((int64_t)1 << 31);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int64_t r, piece;
    fast_divmod(i, per_row, fits32, r, piece);
    const int c = (int)piece * VEC;
    const float f = alpha * s[r];
    if (vec_ok) {
      float v[VEC];
      VecIO<T, VEC>::load(x + r * ldx + c, v);
#pragma unroll
      for (int k = 0; k < VEC; ++k) v[k] *= f;
      VecIO<T, VEC>::store(out + r * ldo + c, v);
    } else {
      for (int k = 0; k < VEC && c + k < cols; ++k) Elem<T>::store(out + r * ldo + c + k, Elem<T>::load(x + r * ldx + c + k) * f);
    }
  }
}

// out[r] = sum_c a[r, c] * (b[r, c] - shift[c])   (f32; one wave per row, 16 bytes per lane and step)
template <typename T>
__global__ __launch_bounds__(256) void row_dot_kernel(const T* __restrict__ a, int64_t lda, const T* __restrict__ b,
                                                      int64_t ldb, const float* __restrict__ shift,
                                                      float* __restrict__ out, int64_t rows, int cols) {
  constexpr int VEC = 16 / sizeof(T);
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const T* ar = a + r * lda;
  const T* br = b + r * ldb;
  float acc = 0.f;
  const bool vec_ok = cols % VEC == 0 && lda % VEC == 0 && ldb % VEC == 0 && (uintptr_t)a % 16 == 0 && (uintptr_t)b % 16 == 0;
  if (vec_ok) {
    constexpr int U = 4;  // 8 independent 16-byte loads per lane in flight (a row of 2048 bf16 is one trip)
    for (int c0 = lane * VEC; c0 < cols; c0 += U * 64 * VEC) {
      float av[U][VEC], bv[U][VEC];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int c = c0 + u * 64 * VEC;
        if (c < cols) {
          VecIO<T, VEC>::load(ar + c, av[u]);
          VecIO<T, VEC>::load(br + c, bv[u]);
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int c = c0 + u * 64 * VEC;
        if (c < cols) {
#pragma unroll
          for (int i = 0; i < VEC; ++i) acc = fmaf(av[u][i], bv[u][i] - (shift != nullptr ? shift[c + i] : 0.f), acc);
        }
      }
    }
  } else {
    for (int c = lane; c < cols; c += 64)
      acc = fmaf(Elem<T>::load(ar + c), Elem<T>::load(br + c) - (shift != nullptr ? shift[c] : 0.f), acc);
  }
  acc = wave_sum(acc);
  if (lane == 0) out[r] = acc;
}

template <typename T>
__global__ __launch_bounds__(256) void add_kernel(const T* __restrict__ a, int64_t lda, const T* __restrict__ b,
                                                  int64_t ldb, T* __restrict__ y, int64_t ldy, int64_t rows,
                                                  int cols) {
  const int64_t total = rows * cols;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = idx / cols;
    const int c = (int)(idx - r * cols);
    Elem<T>::store(y + r * ldy + c, Elem<T>::load(a + r * lda + c) + Elem<T>::load(b + r * ldb + c));
  }
}

__global__ __launch_bounds__(256) void prognostic_residual_kernel(float* __restrict__ y, int V_out,
                                                                  const float* __restrict__ x, int B, int T_,
                                                                  int Ens, int64_t G, int V_in,
                                                                  const int32_t* __restrict__ out_idx,
                                                                  const int32_t* __restrict__ in_idx, int n_prog) {
  const int64_t total = (int64_t)B * Ens * G * n_prog;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = idx / n_prog;  // (b, ens, g)
    const int p = (int)(idx - row * n_prog);
    const int64_t g = row % G;
    const int64_t be = row / G;
    const int e = (int)(be % Ens);
    const int64_t b = be / Ens;
    y[row * V_out + out_idx[p]] += x[((((b * T_ + (T_ - 1)) * Ens + e) * G) + g) * V_in + in_idx[p]];
  }
}

// In-place autoregressive input update: one thread per (b, ens, g, v); the lane reads x[t + 1] before it writes x[t],
// and nobody else touches column v of that grid point, so the time shift is safe in place.
__global__ __launch_bounds__(256) void advance_input_kernel(float* __restrict__ x, int B, int T_, int Ens, int64_t G,
                                                            int V_in, const float* __restrict__ y, int V_out,
                                                            const float* __restrict__ forcing, int F,
                                                            const int32_t* __restrict__ colmap) {
  const int64_t total = (int64_t)B * Ens * G * V_in;
  const bool fits32 = total < ((int64_t)1 << 31);
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    int64_t row, v64, g, be, e64, b;  // row = (b, ens, g)
    fast_divmod(idx, V_in, fits32, row, v64);
    fast_divmod(row, G, fits32, be, g);
    fast_divmod(be, Ens, fits32, b, e64);
    const int v = (int)v64, e = (int)e64;
    const int64_t t_stride = (int64_t)Ens * G * V_in;
    float* xp = x + (((b * T_) * Ens + e) * G + g) * V_in + v;  // time slice 0 of this element
    for (int t = 0; t + 1 < T_; ++t) xp[t * t_stride] = xp[(t + 1) * t_stride];
    const int m = colmap[v];
    if (m >= 0) xp[(T_ - 1) * t_stride] = y[row * V_out + m];
    else if (m <= -2 && forcing != nullptr) xp[(T_ - 1) * t_stride] = forcing[row * F + (-2 - m)];
  }
}

// y[row, c] <- ((y[row, c] + [c prognostic] normalised x[b, T-1, ens, g, src[c]]) - out_add[c]) / out_mul[c]: prognostic
// residual and InputNormalizer.inverse_transform in one pass over the output
template <bool IDX32>
__global__ __launch_bounds__(256) void finalize_output_kernel(float* __restrict__ y, int V_out,
                                                              const float* __restrict__ x, int B, int T_, int Ens,
                                                              int64_t G, int V_in, const int32_t* __restrict__ src,
                                                              const float* __restrict__ in_mul,
                                                              const float* __restrict__ in_add,
                                                              const float* __restrict__ out_mul,
                                                              const float* __restrict__ out_add,
                                                              const int64_t* __restrict__ rows, int64_t n_rows) {
  // rows != nullptr (anemoi_finalize_output_rows, B = Ens = 1): y holds n_rows rows, row i is grid node rows[i]
  const int64_t total = (rows != nullptr ? n_rows : (int64_t)B * Ens * G) * V_out;
  using I = typename std::conditional<IDX32, unsigned, int64_t>::type;  // (see assemble_nodes8_kernel)
  const bool one_be = B * Ens == 1;
  for (I idx = (I)blockIdx.x * blockDim.x + threadIdx.x; idx < (I)total; idx += (I)gridDim.x * blockDim.x) {
    const I row = idx / (I)V_out;  // (b, ens, g)
    const int c = (int)(idx - row * (I)V_out);
    float val = y[idx];
    const int sv = src[c];
    if (sv >= 0) {
      int64_t g, b = 0;
      int e = 0;
      if (rows != nullptr) g = rows[row];
      else if (one_be) g = (int64_t)row;
      else {
        const I be = row / (I)G;
        g = (int64_t)(row - be * (I)G);
        e = (int)(be % (I)Ens);
        b = (int64_t)(be / (I)Ens);
      }
      float xv = x[((((b * T_ + (T_ - 1)) * Ens + e) * G) + g) * V_in + sv];
      if (in_mul != nullptr) xv = xv * in_mul[sv] + in_add[sv];
      val += xv;
    }
    if (out_mul != nullptr) val = (val - out_add[c]) / out_mul[c];
    y[idx] = val;
  }
}

// Output boundings (reference layers/bounding.py:60-124) in place on the f32 output: one thread per (b, ens, g) row walks
// the op list IN ORDER -- y[col] = clamp(y[col], lo, hi) * (mul >= 0 ? y[mul] : 1) -- which is the order the
// reference's chained in-place index assignments define; then the columns listed in fin_* are de-normalised
// ((y - add) / mul: they were left normalised by anemoi_finalize_output so that the boundings see normalised values).
// Comparisons (not fminf / fmaxf) keep a NaN a NaN, like torch's relu / hardtanh.
__global__ __launch_bounds__(256) void bound_output_kernel(float* __restrict__ y, int V_out, int64_t rows, int n_ops,
                                                           const int32_t* __restrict__ op_col,
                                                           const float* __restrict__ op_lo,
                                                           const float* __restrict__ op_hi,
                                                           const int32_t* __restrict__ op_mul, int n_fin,
                                                           const int32_t* __restrict__ fin_col,
                                                           const float* __restrict__ fin_mul,
                                                           const float* __restrict__ fin_add) {
  for (int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; row < rows;
       row += (int64_t)gridDim.x * blockDim.x) {
    float* yr = y + row * V_out;
    for (int i = 0; i < n_ops; ++i) {
      const int c = op_col[i];
      float v = yr[c];
      const float lo = op_lo[i], hi = op_hi[i];
      v = v < lo ? lo : (v > hi ? hi : v);
      const int m = op_mul[i];
      if (m >= 0) v *= yr[m];
      yr[c] = v;
    }
    for (int j = 0; j < n_fin; ++j) {
      const int c = fin_col[j];
      yr[c] = (yr[c] - fin_add[j]) / fin_mul[j];
    }
  }
}

// ANEMOI_AMD_IDX64=1 forces the 64-bit index instantiations of the glue kernels (tests: no tensor of 2^31 elements needed)
static bool force_idx64() {
  static const bool v = [] {
    const char* e = getenv("ANEMOI_AMD_IDX64");
    return e != nullptr && atoi(e) != 0;
  }();
  return v;
}

static inline unsigned flat_grid(int64_t total) {
  int64_t blocks = (total + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;  // grid-stride the rest
  if (blocks < 1) blocks = 1;
  return (unsigned)blocks;
}

}  // namespace anemoi

using namespace anemoi;

extern "C" {

int anemoi_layer_norm(int dtype, const void* x, int64_t ldx, const float* gamma, const float* beta, void* y,
                      int64_t ldy, int64_t rows, int C, float eps, anemoi_stream_t stream) {
  ANEMOI_REQUIRE(x && y && gamma && beta, ANEMOI_ERR_INVALID, "anemoi_layer_norm: null pointer");
  ANEMOI_REQUIRE(C > 0 && rows >= 0 && ldx >= C && ldy >= C, ANEMOI_ERR_INVALID,
                 "anemoi_layer_norm: bad shape rows=%lld C=%d ldx=%lld ldy=%lld", (long long)rows, C, (long long)ldx,
                 (long long)ldy);
  if (rows == 0) return ANEMOI_OK;
  if (dtype == ANEMOI_F32) return layer_norm_launch<float>(x, ldx, gamma, beta, y, ldy, rows, C, eps, as_stream(stream));
  if (dtype == ANEMOI_BF16) return layer_norm_launch<bf16_t>(x, ldx, gamma, beta, y, ldy, rows, C, eps, as_stream(stream));
  return fail(ANEMOI_ERR_UNSUPPORTED, "anemoi_layer_norm: dtype %d", dtype);
}

int anemoi_layer_norm_residual(int dtype, const void* x, int64_t ldx, const float* gamma, const float* beta,
                               const void* residual, int64_t ldr, void* y, int64_t ldy, int64_t rows, int C, float eps,
                               anemoi_stream_t stream) {
  ANEMOI_REQUIRE(x && y && gamma && beta && residual, ANEMOI_ERR_INVALID, "anemoi_layer_norm_residual: null pointer");
  ANEMOI_REQUIRE(C > 0 && rows >= 0 && ldx >= C && ldy >= C && ldr >= C, ANEMOI_ERR_INVALID,
                 "anemoi_layer_norm_residual: bad shape rows=%lld C=%d ldx=%lld ldr=%lld ldy=%lld", (long long)rows, C,
                 (long long)ldx, (long long)ldr, (long long)ldy);
  if (rows == 0) return ANEMOI_OK;
  if (dtype == ANEMOI_F32)
    return layer_norm_launch<float>(x, ldx, gamma, beta, y, ldy, rows, C, eps, as_stream(stream), nullptr, residual, ldr);
  if (dtype == ANEMOI_BF16)
    return layer_norm_launch<bf16_t>(x, ldx, gamma, beta, y, ldy, rows, C, eps, as_stream(stream), nullptr, residual, ldr);
  return fail(ANEMOI_ERR_UNSUPPORTED, "anemoi_layer_norm_residual: dtype %d", dtype);
}

int anemoi_layer_norm_stats(int dtype, const void* x, int64_t ldx, const float* gamma, const float* beta, void* y,
                            int64_t ldy, float* stats, int64_t rows, int C, float eps, anemoi_stream_t stream) {
  ANEMOI_REQUIRE(x && y && gamma && beta && stats, ANEMOI_ERR_INVALID, "anemoi_layer_norm_stats: null pointer");
  ANEMOI_REQUIRE(C > 0 && rows >= 0 && ldx >= C && ldy >= C && (uintptr_t)stats % 8 == 0, ANEMOI_ERR_INVALID,
                 "anemoi_layer_norm_stats: bad shape rows=%lld C=%d ldx=%lld ldy=%lld (stats 8-byte aligned)",
                 (long long)rows, C, (long long)ldx, (long long)ldy);
  if (rows == 0) return ANEMOI_OK;
  float2* sp = reinterpret_cast<float2*>(stats);
  if (dtype == ANEMOI_F32)
    return layer_norm_launch<float>(x, ldx, gamma, beta, y, ldy, rows, C, eps, as_stream(stream), sp);
  if (dtype == ANEMOI_BF16)
    return layer_norm_launch<bf16_t>(x, ldx, gamma, beta, y, ldy, rows, C, eps, as_stream(stream), sp);
  return fail(ANEMOI_ERR_UNSUPPORTED, "anemoi_layer_norm_stats: dtype %d", dtype);
}

int anemoi_row_stats(int dtype, const void* x, int64_t ldx, float* stats, int64_t rows, int C, float eps,
                     anemoi_stream_t stream) {
  ANEMOI_REQUIRE(x && stats, ANEMOI_ERR_INVALID, "anemoi_row_stats: null pointer");
  ANEMOI_REQUIRE(C > 0 && rows >= 0 && ldx >= C, ANEMOI_ERR_INVALID, "anemoi_row_stats: bad shape rows=%lld C=%d ldx=%lld",
                 (long long)rows, C, (long long)ldx);
  ANEMOI_REQUIRE((uintptr_t)stats % 8 == 0, ANEMOI_ERR_INVALID, "anemoi_row_stats: stats must be 8-byte aligned");
  if (rows == 0) return ANEMOI_OK;
  if (dtype == ANEMOI_F32) return row_stats_launch<float>(x, ldx, stats, rows, C, eps, as_stream(stream));
  if (dtype == ANEMOI_BF16) return row_stats_launch<bf16_t>(x, ldx, stats, rows, C, eps, as_stream(stream));
  return fail(ANEMOI_ERR_UNSUPPORTED, "anemoi_row_stats: dtype %d", dtype);
}

static int assemble_nodes_impl(int dtype, const float* x, int B, int T, int Ens, int64_t G, int V, const float* latlons,
                               int n_ll, const float* trainable, int n_tr, void* out, int64_t ldo, const float* in_mul,
                               const float* in_add, const int64_t* rows, int64_t n_rows, anemoi_stream_t stream) {
  ANEMOI_REQUIRE((in_mul == nullptr) == (in_add == nullptr), ANEMOI_ERR_INVALID,
                 "anemoi_assemble_nodes: in_mul and in_add come together");
  ANEMOI_REQUIRE(out && B > 0 && Ens > 0 && G >= 0 && T >= 0 && V >= 0 && n_ll >= 0 && n_tr >= 0, ANEMOI_ERR_INVALID,
                 "anemoi_assemble_nodes: bad argument");
  ANEMOI_REQUIRE((x != nullptr) || T * V == 0, ANEMOI_ERR_INVALID, "anemoi_assemble_nodes: x is null");
  ANEMOI_REQUIRE((latlons != nullptr) || n_ll == 0, ANEMOI_ERR_INVALID, "anemoi_assemble_nodes: latlons is null");
  ANEMOI_REQUIRE((trainable != nullptr) || n_tr == 0, ANEMOI_ERR_INVALID, "anemoi_assemble_nodes: trainable is null");
  ANEMOI_REQUIRE(ldo >= (int64_t)T * V + n_ll + n_tr, ANEMOI_ERR_INVALID, "anemoi_assemble_nodes: ldo too small");
  const int64_t total = (rows != nullptr ? n_rows : (int64_t)B * Ens * G) * ldo;
  if (total == 0) return ANEMOI_OK;
  hipStream_t st = as_stream(stream);
  if (ldo % 8 == 0 && ldo < ((int64_t)1 << 31) && (uintptr_t)out % 16 == 0 && (dtype == ANEMOI_F32 || dtype == ANEMOI_BF16)) {
    const int ldo8 = (int)(ldo / 8);
#define ANEMOI_ASM8(TT, I32)                                                                                          \
  hipLaunchKernelGGL((assemble_nodes8_kernel<TT, I32>), dim3(flat_grid(total / 8)), dim3(256), 0, st, x, B, T, Ens, G, V, \
                     latlons, n_ll, trainable, n_tr, static_cast<TT*>(out), ldo8, in_mul, in_add, rows, n_rows)
    const bool idx32 = !force_idx64() && total / 8 < ((int64_t)1 << 31) - ((int64_t)1 << 24);  // (+ one grid stride stays below 2^31)
    if (dtype == ANEMOI_F32) {
      if (idx32) ANEMOI_ASM8(float, true);
      else ANEMOI_ASM8(float, false);
    } else {
      if (idx32) ANEMOI_ASM8(bf16_t, true);
      else ANEMOI_ASM8(bf16_t, false);
    }
#undef ANEMOI_ASM8
    return check_launch("anemoi_assemble_nodes");
  }
  if (dtype == ANEMOI_F32)
    hipLaunchKernelGGL((assemble_nodes_kernel<float>), dim3(flat_grid(total)), dim3(256), 0, st, x, B, T, Ens, G, V,
                       latlons, n_ll, trainable, n_tr, static_cast<float*>(out), ldo, in_mul, in_add, rows, n_rows);
  else if (dtype == ANEMOI_BF16)
    hipLaunchKernelGGL((assemble_nodes_kernel<bf16_t>), dim3(flat_grid(total)), dim3(256), 0, st, x, B, T, Ens, G, V,
                       latlons, n_ll, trainable, n_tr, static_cast<bf16_t*>(out), ldo, in_mul, in_add, rows, n_rows);
  else
    return fail(ANEMOI_ERR_UNSUPPORTED, "anemoi_assemble_nodes: dtype %d", dtype);
  return check_launch("anemoi_assemble_nodes");
}

int anemoi_assemble_nodes(int dtype, const float* x, int B, int T, int Ens, int64_t G, int V, const float* latlons,
                          int n_ll, const float* trainable, int n_tr, void* out, int64_t ldo, const float* in_mul,
                          const float* in_add, anemoi_stream_t stream) {
  return assemble_nodes_impl(dtype, x, B, T, Ens, G, V, latlons, n_ll, trainable, n_tr, out, ldo, in_mul, in_add, nullptr, 0,
                             stream);
}

int anemoi_assemble_node_rows(int dtype, const float* x, int T, int64_t G, int V, const float* latlons, int n_ll,
                              const float* trainable, int n_tr, const int64_t* rows, int64_t n_rows, void* out,
                              int64_t ldo, const float* in_mul, const float* in_add, anemoi_stream_t stream) {
  ANEMOI_REQUIRE(n_rows >= 0 && (rows != nullptr || n_rows == 0), ANEMOI_ERR_INVALID,
                 "anemoi_assemble_node_rows: null row list");
  if (n_rows == 0) return ANEMOI_OK;
  return assemble_nodes_impl(dtype, x, 1, T, 1, G, V, latlons, n_ll, trainable, n_tr, out, ldo, in_mul, in_add, rows, n_rows,
                             stream);
}

int anemoi_edge_attr_csr(const float* a0, int d0, const float* a1, int d1, int64_t rows0, const int32_t* perm,
                         float* out, int ld_out, int one_col, int64_t n_edges, anemoi_stream_t stream) {
  ANEMOI_REQUIRE(a0 && perm && out && d0 > 0 && d1 >= 0 && rows0 > 0 && n_edges >= 0, ANEMOI_ERR_INVALID,
                 "anemoi_edge_attr_csr: bad argument");
  ANEMOI_REQUIRE((a1 != nullptr) || d1 == 0, ANEMOI_ERR_INVALID, "anemoi_edge_attr_csr: a1 is null but d1 > 0");
  ANEMOI_REQUIRE(ld_out >= d0 + d1, ANEMOI_ERR_INVALID, "anemoi_edge_attr_csr: ld_out %d < %d", ld_out, d0 + d1);
  ANEMOI_REQUIRE(one_col < ld_out && (one_col < 0 || one_col >= d0 + d1), ANEMOI_ERR_INVALID,
                 "anemoi_edge_attr_csr: one_col %d must be a padding column", one_col);
  if (n_edges == 0) return ANEMOI_OK;
  hipLaunchKernelGGL(edge_attr_csr_kernel, dim3(flat_grid(n_edges * ld_out)), dim3(256), 0, as_stream(stream), a0, d0,
                     a1, d1, rows0, perm, out, ld_out, one_col, n_edges);
  return check_launch("anemoi_edge_attr_csr");
}

int anemoi_convert_pad(int src_dtype, const void* src, int64_t ld_src, int dst_dtype, void* dst, int64_t ld_dst,
                       int64_t rows, int cols, anemoi_stream_t stream) {
  ANEMOI_REQUIRE(src && dst && rows >= 0 && cols >= 0 && ld_src >= cols && ld_dst >= cols, ANEMOI_ERR_INVALID,
                 "anemoi_convert_pad: bad argument");
  if (rows == 0 || ld_dst == 0) return ANEMOI_OK;
  hipStream_t st = as_stream(stream);
  dim3 grid(flat_grid(rows * ld_dst)), block(256);
#define CP(S, D)                                                                                           \
  hipLaunchKernelGGL((convert_pad_kernel<S, D>), grid, block, 0, st, static_cast<const S*>(src), ld_src, \
                     static_cast<D*>(dst), ld_dst, rows, cols)
  if (src_dtype == ANEMOI_F32 && dst_dtype == ANEMOI_F32) CP(float, float);
  else if (src_dtype == ANEMOI_F32 && dst_dtype == ANEMOI_BF16) CP(float, bf16_t);
  else if (src_dtype == ANEMOI_BF16 && dst_dtype == ANEMOI_F32) CP(bf16_t, float);
  else if (src_dtype == ANEMOI_BF16 && dst_dtype == ANEMOI_BF16) CP(bf16_t, bf16_t);
  else return fail(ANEMOI_ERR_UNSUPPORTED, "anemoi_convert_pad: dtypes %d -> %d", src_dtype, dst_dtype);
#undef CP
  return check_launch("anemoi_convert_pad");
}

int anemoi_row_scale(int dtype, const void* x, int64_t ldx, const float* s, float alpha, void* out, int64_t ldo,
                     int64_t rows, int cols, anemoi_stream_t stream) {
  ANEMOI_REQUIRE(x && s && out && rows >= 0 && cols >= 0 && ldx >= cols && ldo >= cols, ANEMOI_ERR_INVALID,
                 "anemoi_row_scale: bad argument");
  if (rows * cols == 0) return ANEMOI_OK;
  hipStream_t st = as_stream(stream);
  const int vec = dtype == ANEMOI_F32 ? 4 : 8;
  dim3 grid(flat_grid(rows * ((cols + vec - 1) / vec))), block(256);
  if (dtype == ANEMOI_F32)
    hipLaunchKernelGGL((row_scale_kernel<float>), grid, block, 0, st, static_cast<const float*>(x), ldx, s, alpha,
                       static_cast<float*>(out), ldo, rows, cols);
  else if (dtype == ANEMOI_BF16)
    hipLaunchKernelGGL((row_scale_kernel<bf16_t>), grid, block, 0, st, static_cast<const bf16_t*>(x), ldx, s, alpha,
                       static_cast<bf16_t*>(out), ldo, rows, cols);
  else
    return fail(ANEMOI_ERR_UNSUPPORTED, "anemoi_row_scale: dtype %d", dtype);
  return check_launch("anemoi_row_scale");
}

int anemoi_row_dot(int dtype, const void* a, int64_t lda, const void* b, int64_t ldb, const float* shift, float* out,
                   int64_t rows, int cols, anemoi_stream_t stream) {
  ANEMOI_REQUIRE(a && b && out && rows >= 0 && cols >= 0 && lda >= cols && ldb >= cols, ANEMOI_ERR_INVALID,
                 "anemoi_row_dot: bad argument");
  if (rows == 0) return ANEMOI_OK;
  hipStream_t st = as_stream(stream);
  dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  ANEMOI_REQUIRE((rows + 3) / 4 < ((int64_t)1 << 31), ANEMOI_ERR_UNSUPPORTED, "anemoi_row_dot: grid too large");
  if (dtype == ANEMOI_F32)
    hipLaunchKernelGGL((row_dot_kernel<float>), grid, block, 0, st, static_cast<const float*>(a), lda,
                       static_cast<const float*>(b), ldb, shift, out, rows, cols);
  else if (dtype == ANEMOI_BF16)
    hipLaunchKernelGGL((row_dot_kernel<bf16_t>), grid, block, 0, st, static_cast<const bf16_t*>(a), lda,
                       static_cast<const bf16_t*>(b), ldb, shift, out, rows, cols);
  else
    return fail(ANEMOI_ERR_UNSUPPORTED, "anemoi_row_dot: dtype %d", dtype);
  return check_launch("anemoi_row_dot");
}

int anemoi_add(int dtype, const void* a, int64_t lda, const void* b, int64_t ldb, void* y, int64_t ldy, int64_t rows,
               int cols, anemoi_stream_t stream) {
  ANEMOI_REQUIRE(a && b && y && rows >= 0 && cols >= 0, ANEMOI_ERR_INVALID, "anemoi_add: bad argument");
  if (rows * cols == 0) return ANEMOI_OK;
  hipStream_t st = as_stream(stream);
  dim3 grid(flat_grid(rows * cols)), block(256);
  if (dtype == ANEMOI_F32)
    hipLaunchKernelGGL((add_kernel<float>), grid, block, 0, st, static_cast<const float*>(a), lda,
                       static_cast<const float*>(b), ldb, static_cast<float*>(y), ldy, rows, cols);
  else if (dtype == ANEMOI_BF16)
    hipLaunchKernelGGL((add_kernel<bf16_t>), grid, block, 0, st, static_cast<const bf16_t*>(a), lda,
                       static_cast<const bf16_t*>(b), ldb, static_cast<bf16_t*>(y), ldy, rows, cols);
  else
    return fail(ANEMOI_ERR_UNSUPPORTED, "anemoi_add: dtype %d", dtype);
  return check_launch("anemoi_add");
}

int anemoi_prognostic_residual(float* y, int V_out, const float* x, int B, int T, int Ens, int64_t G, int V_in,
                               const int32_t* out_idx, const int32_t* in_idx, int n_prog, anemoi_stream_t stream) {
  ANEMOI_REQUIRE(y && x && B > 0 && T > 0 && Ens > 0 && G >= 0 && n_prog >= 0, ANEMOI_ERR_INVALID,
                 "anemoi_prognostic_residual: bad argument");
  ANEMOI_REQUIRE(n_prog == 0 || (out_idx && in_idx), ANEMOI_ERR_INVALID, "anemoi_prognostic_residual: null index");
  const int64_t total = (int64_t)B * Ens * G * n_prog;
  if (total == 0) return ANEMOI_OK;
  hipLaunchKernelGGL(prognostic_residual_kernel, dim3(flat_grid(total)), dim3(256), 0, as_stream(stream), y, V_out, x,
                     B, T, Ens, G, V_in, out_idx, in_idx, n_prog);
  return check_launch("anemoi_prognostic_residual");
}

int anemoi_advance_input(float* x, int B, int T, int Ens, int64_t G, int V_in, const float* y, int V_out,
                         const float* forcing, int F, const int32_t* colmap, anemoi_stream_t stream) {
  ANEMOI_REQUIRE(x && y && colmap && B > 0 && T > 0 && Ens > 0 && G >= 0 && V_in > 0 && V_out > 0 && F >= 0,
                 ANEMOI_ERR_INVALID, "anemoi_advance_input: bad argument");
  const int64_t total = (int64_t)B * Ens * G * V_in;
  if (total == 0) return ANEMOI_OK;
  hipLaunchKernelGGL(advance_input_kernel, dim3(flat_grid(total)), dim3(256), 0, as_stream(stream), x, B, T, Ens, G,
                     V_in, y, V_out, forcing, F, colmap);
  return check_launch("anemoi_advance_input");
}

static int finalize_output_impl(float* y, int V_out, const float* x, int B, int T, int Ens, int64_t G, int V_in,
                                const int32_t* src, const float* in_mul, const float* in_add, const float* out_mul,
                                const float* out_add, const int64_t* rows, int64_t n_rows, anemoi_stream_t stream) {
  ANEMOI_REQUIRE(y && x && src && B > 0 && T > 0 && Ens > 0 && G >= 0 && V_in > 0 && V_out > 0, ANEMOI_ERR_INVALID,
                 "anemoi_finalize_output: bad argument");
  ANEMOI_REQUIRE((in_mul == nullptr) == (in_add == nullptr) && (out_mul == nullptr) == (out_add == nullptr),
                 ANEMOI_ERR_INVALID, "anemoi_finalize_output: mul and add come together");
  const int64_t total = (rows != nullptr ? n_rows : (int64_t)B * Ens * G) * V_out;
  if (total == 0) return ANEMOI_OK;
  if (!force_idx64() && total < ((int64_t)1 << 31) - ((int64_t)1 << 24))
    hipLaunchKernelGGL(finalize_output_kernel<true>, dim3(flat_grid(total)), dim3(256), 0, as_stream(stream), y, V_out, x, B,
                       T, Ens, G, V_in, src, in_mul, in_add, out_mul, out_add, rows, n_rows);
  else
    hipLaunchKernelGGL(finalize_output_kernel<false>, dim3(flat_grid(total)), dim3(256), 0, as_stream(stream), y, V_out, x, B,
                       T, Ens, G, V_in, src, in_mul, in_add, out_mul, out_add, rows, n_rows);
  return check_launch("anemoi_finalize_output");
}

int anemoi_finalize_output(float* y, int V_out, const float* x, int B, int T, int Ens, int64_t G, int V_in,
                           const int32_t* src, const float* in_mul, const float* in_add, const float* out_mul,
                           const float* out_add, anemoi_stream_t stream) {
  return finalize_output_impl(y, V_out, x, B, T, Ens, G, V_in, src, in_mul, in_add, out_mul, out_add, nullptr, 0, stream);
}

int anemoi_finalize_output_rows(float* y, int V_out, const float* x, int T, int64_t G, int V_in, const int32_t* src,
                                const int64_t* rows, int64_t n_rows, const float* in_mul, const float* in_add,
                                const float* out_mul, const float* out_add, anemoi_stream_t stream) {
  ANEMOI_REQUIRE(n_rows >= 0 && (rows != nullptr || n_rows == 0), ANEMOI_ERR_INVALID,
                 "anemoi_finalize_output_rows: null row list");
  if (n_rows == 0) return ANEMOI_OK;
  return finalize_output_impl(y, V_out, x, 1, T, 1, G, V_in, src, in_mul, in_add, out_mul, out_add, rows, n_rows, stream);
}

int anemoi_bound_output(float* y, int V_out, int64_t rows, int n_ops, const int32_t* op_col, const float* op_lo,
                        const float* op_hi, const int32_t* op_mul, int n_fin, const int32_t* fin_col,
                        const float* fin_mul, const float* fin_add, anemoi_stream_t stream) {
  ANEMOI_REQUIRE(y && V_out > 0 && rows >= 0 && n_ops >= 0 && n_fin >= 0, ANEMOI_ERR_INVALID,
                 "anemoi_bound_output: bad argument");
  ANEMOI_REQUIRE(n_ops == 0 || (op_col && op_lo && op_hi && op_mul), ANEMOI_ERR_INVALID,
                 "anemoi_bound_output: null op list");
  ANEMOI_REQUIRE(n_fin == 0 || (fin_col && fin_mul && fin_add), ANEMOI_ERR_INVALID,
                 "anemoi_bound_output: null de-normalisation list");
  if (rows == 0 || (n_ops == 0 && n_fin == 0)) return ANEMOI_OK;
  hipLaunchKernelGGL(bound_output_kernel, dim3(flat_grid(rows)), dim3(256), 0, as_stream(stream), y, V_out, rows, n_ops,
                     op_col, op_lo, op_hi, op_mul, n_fin, fin_col, fin_mul, fin_add);
  return check_launch("anemoi_bound_output");
}

int anemoi_abi_version(void) { return 41; }

#ifndef ANEMOI_HIPCC_VERSION
#define ANEMOI_HIPCC_VERSION "unknown (built without anemoi_models_amd/_build.py)"
#endif
const char* anemoi_build_info(void) { return "gfx950; " ANEMOI_HIPCC_VERSION; }

const char* anemoi_last_error(void) { return err_buf(); }

}  // extern "C"
