// Dense half of the backward pass (first step): what the autograd of the fused Linear and of
// LayerNorm needs besides the forward GEMM kernels themselves.
//   dX = dpre W            -> anemoi_linear on (dpre, W^T)          W^T from anemoi_transpose
//   dW = dpre^T X          -> anemoi_linear on (dpre^T, X^T), f32   both from anemoi_transpose (K = rows, zero padded)
//   db = column sums of dpre                                         anemoi_col_sum (two deterministic stages)
//   dpre = dy * act'(pre)                                            anemoi_act_backward
//   LayerNorm: dx per row, d gamma / d beta as column sums           anemoi_layer_norm_backward
// Everything here is bound by HBM bandwidth; no atomics (gradients are bit-reproducible run to run).
#include "common.hpp"

namespace anemoi {

// ---------------------------------------------------------------------------------------------
// dst[c, r] = src[r, c] for r < rows, c < cols; dst rows are ld_dst long, columns rows..ld_dst-1 are zero filled (the
// transposed matrix becomes the K-contiguous operand of a GEMM whose reduction runs over the original rows).
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void transpose_kernel(const T* __restrict__ src, int64_t ld_src, T* __restrict__ dst,
                                                        int64_t ld_dst, int64_t rows, int cols, int64_t chunk) {
  // 64 x 64 element tile; global reads and writes are 16 bytes per lane wherever alignment and bounds allow (a wave
  // covers 8 rows x 64 columns of 2-byte elements per access), the LDS tile is padded against bank conflicts
  constexpr int V = 16 / sizeof(T);  // elements per 16-byte access: 8 (bf16) or 4 (f32)
  __shared__ T tile[64][64 + V + 1];
  // chunked form (grid.z = chunk s): rows [s * chunk, (s + 1) * chunk) of src become the slab dst[s] = [cols, ld_dst]
  src += (int64_t)blockIdx.z * chunk * ld_src;
  dst += (int64_t)blockIdx.z * cols * ld_dst;
  rows = rows - (int64_t)blockIdx.z * chunk < chunk ? rows - (int64_t)blockIdx.z * chunk : chunk;
  const int64_t r0 = (int64_t)blockIdx.x * 64;
  const int c0 = blockIdx.y * 64;
  const bool vec_in = ((uintptr_t)src % 16 == 0) && (ld_src % V == 0) && c0 + 64 <= cols;
  const bool vec_out = ((uintptr_t)dst % 16 == 0) && (ld_dst % V == 0) && r0 + 64 <= ld_dst;
  constexpr int PER_ROW = 64 / V;  // 16-byte pieces per tile row
  for (int idx = threadIdx.x; idx < 64 * PER_ROW; idx += 256) {
    const int r = idx / PER_ROW, cp = (idx % PER_ROW) * V;
    if (vec_in && r0 + r < rows) {
      T v[V];
      *reinterpret_cast<uint4*>(v) = *reinterpret_cast<const uint4*>(src + (r0 + r) * ld_src + c0 + cp);
#pragma unroll
      for (int i = 0; i < V; ++i) tile[r][cp + i] = v[i];
    } else {
#pragma unroll
      for (int i = 0; i < V; ++i)
        tile[r][cp + i] = (r0 + r < rows && c0 + cp + i < cols) ? src[(r0 + r) * ld_src + c0 + cp + i] : (T)0;
    }
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < 64 * PER_ROW; idx += 256) {
    const int c = idx / PER_ROW, rp = (idx % PER_ROW) * V;  // output row = source column c, V source rows per piece
    if (c0 + c >= cols) continue;
    T v[V];
#pragma unroll
    for (int i = 0; i < V; ++i) v[i] = tile[rp + i][c];
    T* out = dst + (int64_t)(c0 + c) * ld_dst + r0 + rp;
    if (vec_out) {
      *reinterpret_cast<uint4*>(out) = *reinterpret_cast<const uint4*>(v);
    } else {
#pragma unroll
      for (int i = 0; i < V; ++i)
        if (r0 + rp + i < ld_dst) out[i] = v[i];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// The same transpose for 2-byte elements without LDS: a wave owns a 64 x 64 tile, a lane an 8 x 8 sub-block -- eight
// 16-byte row pieces in, sixteen v_perm_b32 per register pair turn them into the eight 16-byte pieces of the transposed
// sub-block (output row c, dword m = { src[2m][c], src[2m + 1][c] }: pick the low or the high halves of two source
// dwords; everything above the 16-bit level is register naming), eight 16-byte pieces out.  For a fixed piece index the
// wave reads 8 rows x 128 contiguous bytes and writes 8 rows x 128 contiguous bytes.  Ragged tiles (and unaligned
// operands) take an element-wise path in the same kernel.  The LDS kernel above moved 1.9 TB/s (eight 2-byte LDS reads
// per 16 bytes written).
// ---------------------------------------------------------------------------------------------
// ``colsum`` (optional, bf16 sources): partial column sums [gridDim.z * tiles_per_chunk, cols] f32 -- row tile t of chunk
// z leaves the sums of its 64 source rows in row z * tiles_per_chunk + t.  The bias gradient (column sums of dpre) then
// costs one pass over that small matrix instead of a second pass over dpre (anemoi_col_sum: 6.4 ms per config-3 step).
__global__ __launch_bounds__(256) void transpose_b16_kernel(const uint16_t* __restrict__ src, int64_t ld_src,
                                                            uint16_t* __restrict__ dst, int64_t ld_dst, int64_t rows,
                                                            int cols, int64_t chunk, float* __restrict__ colsum,
                                                            int tiles_per_chunk) {
  src += (int64_t)blockIdx.z * chunk * ld_src;
  dst += (int64_t)blockIdx.z * cols * ld_dst;
  rows = rows - (int64_t)blockIdx.z * chunk < chunk ? rows - (int64_t)blockIdx.z * chunk : chunk;
  const int lane = threadIdx.x & 63;
  const int tile_r = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);  // a wave per tile, four row tiles per block
  const int64_t r0 = (int64_t)tile_r * 64;
  const int c0 = blockIdx.y * 64;
  if (r0 >= ld_dst) return;
  float* cs_row = colsum != nullptr && tile_r < tiles_per_chunk
                      ? colsum + ((int64_t)blockIdx.z * tiles_per_chunk + tile_r) * cols : nullptr;
  const bool whole = ((uintptr_t)src % 16 == 0) && ((uintptr_t)dst % 16 == 0) && (ld_src % 8 == 0) && (ld_dst % 8 == 0) &&
                     c0 + 64 <= cols && r0 + 64 <= rows;
  if (whole) {
    const int cp = (lane & 7) * 8, rg = (lane >> 3) * 8;  // 8 column pieces x 8 row groups
    uint4 in[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
      in[i] = *reinterpret_cast<const uint4*>(src + (r0 + rg + i) * ld_src + c0 + cp);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const uint32_t sel = (c & 1) ? 0x07060302u : 0x05040100u;  // high / low halves of (b, a) -> { a.half, b.half }
      uint32_t o[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const uint32_t a = reinterpret_cast<const uint32_t*>(&in[2 * m])[c >> 1];
        const uint32_t b = reinterpret_cast<const uint32_t*>(&in[2 * m + 1])[c >> 1];
        o[m] = __builtin_amdgcn_perm(b, a, sel);
      }
      *reinterpret_cast<uint4*>(dst + (int64_t)(c0 + cp + c) * ld_dst + r0 + rg) = make_uint4(o[0], o[1], o[2], o[3]);
    }
    if (cs_row != nullptr) {  // column sums of the tile: this lane's 8 rows, then the 8 row groups (lane bits 3..5)
      float cs[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) cs[j] = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const uint32_t* wv = reinterpret_cast<const uint32_t*>(&in[i]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          cs[2 * j] += __uint_as_float(wv[j] << 16);
          cs[2 * j + 1] += __uint_as_float(wv[j] & 0xffff0000u);
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        cs[j] += __shfl_xor(cs[j], 8, 64);
        cs[j] += __shfl_xor(cs[j], 16, 64);
        cs[j] += __shfl_xor(cs[j], 32, 64);
      }
      if (rg == 0) {
        *reinterpret_cast<float4*>(cs_row + c0 + cp) = make_float4(cs[0], cs[1], cs[2], cs[3]);
        *reinterpret_cast<float4*>(cs_row + c0 + cp + 4) = make_float4(cs[4], cs[5], cs[6], cs[7]);
      }
    }
    return;
  }
  // ragged tile: element-wise, output-major (lanes along the output row), zero fill behind the last source row
  for (int idx = lane; idx < 64 * 64; idx += 64) {
    const int c = idx >> 6;
    const int64_t r = r0 + (idx & 63);
    if (c0 + c < cols && r < ld_dst) dst[(int64_t)(c0 + c) * ld_dst + r] = r < rows ? src[r * ld_src + c0 + c] : (uint16_t)0;
  }
  if (cs_row != nullptr && c0 + lane < cols) {  // lane = column: the tile's (at most 64) valid rows
    float t = 0.f;
    for (int64_t r = r0; r < r0 + 64 && r < rows; ++r) t += __uint_as_float((uint32_t)src[r * ld_src + c0 + lane] << 16);
    cs_row[c0 + lane] = t;
  }
}

// ---------------------------------------------------------------------------------------------
// Column sums, second stage: partial [n_part, cols] f32 -> out [cols].  A block owns 16 columns, its 16 row groups add up
// every 16th partial each (four chains), LDS folds the 16 groups in a fixed order (deterministic).  The one-block-per-256
// columns form of the stage kernel below ran 4 .. 17 workgroups for 56 us, 121 times per training step.
// ---------------------------------------------------------------------------------------------
// (out_hi / split: columns >= split go to out_hi[c - split] -- d gamma | d beta of the LayerNorm backward land in their
//  two result vectors without device copies behind this kernel)
__global__ __launch_bounds__(256) void col_sum_final_kernel(const float* __restrict__ partial, int64_t n_part, int cols,
                                                            float* __restrict__ out, float* __restrict__ out_hi, int split) {
  __shared__ float fold[16][17];
  const int cl = threadIdx.x & 15, g = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < cols) {
    int64_t r = g;
    for (; r + 48 < n_part; r += 64) {
      s0 += partial[r * cols + c];
      s1 += partial[(r + 16) * cols + c];
      s2 += partial[(r + 32) * cols + c];
      s3 += partial[(r + 48) * cols + c];
    }
    for (; r < n_part; r += 16) s0 += partial[r * cols + c];
  }
  fold[g][cl] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (g == 0 && c < cols) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += fold[i][cl];
    if (out_hi != nullptr && c >= split) out_hi[c - split] = t;
    else out[c] = t;
  }
}

// ---------------------------------------------------------------------------------------------
// Column sums, stage kernel: block b adds up rows [b * chunk, (b + 1) * chunk) of every column -> partial[b, c] (f32).
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void col_sum_stage_kernel(const T* __restrict__ x, int64_t ldx, int64_t rows, int cols,
                                                            int64_t chunk, float* __restrict__ partial) {
  // grid = (row chunks, column blocks of 256): a thread owns one column of one chunk, a wave reads 64 adjacent columns
  const int64_t r_begin = (int64_t)blockIdx.x * chunk;
  const int64_t r_end = r_begin + chunk < rows ? r_begin + chunk : rows;
  const int c = blockIdx.y * 256 + threadIdx.x;
  if (c >= cols) return;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;  // four independent chains: the loads of a column pipeline
  int64_t r = r_begin;
  for (; r + 3 < r_end; r += 4) {
    s0 += Elem<T>::load(x + r * ldx + c);
    s1 += Elem<T>::load(x + (r + 1) * ldx + c);
    s2 += Elem<T>::load(x + (r + 2) * ldx + c);
    s3 += Elem<T>::load(x + (r + 3) * ldx + c);
  }
  for (; r < r_end; ++r) s0 += Elem<T>::load(x + r * ldx + c);
  partial[(int64_t)blockIdx.x * cols + c] = (s0 + s1) + (s2 + s3);
}

template <typename T>
static int col_sum_launch(const T* x, int64_t ldx, int64_t rows, int cols, float* out, float* workspace,
                          int64_t workspace_floats, hipStream_t st) {
  // stage 1: <= 512 chunks of whole rows x column blocks; stage 2: the chunks' partials, again per column block
  int64_t chunk = (rows + 511) / 512;
  if (chunk < 16) chunk = 16;
  const int64_t blocks = (rows + chunk - 1) / chunk;
  const unsigned cblocks = (unsigned)((cols + 255) / 256);
  if (blocks <= 1) {
    hipLaunchKernelGGL((col_sum_stage_kernel<T>), dim3(1, cblocks), dim3(256), 0, st, x, ldx, rows, cols,
                       rows > 0 ? rows : 1, out);
    return check_launch("anemoi_col_sum");
  }
  ANEMOI_REQUIRE(workspace != nullptr && workspace_floats >= blocks * cols, ANEMOI_ERR_INVALID,
                 "anemoi_col_sum: workspace of %lld floats required", (long long)(blocks * cols));
  hipLaunchKernelGGL((col_sum_stage_kernel<T>), dim3((unsigned)blocks, cblocks), dim3(256), 0, st, x, ldx, rows, cols,
                     chunk, workspace);
  hipLaunchKernelGGL(col_sum_final_kernel, dim3((unsigned)((cols + 15) / 16)), dim3(256), 0, st, workspace, blocks, cols,
                     out, static_cast<float*>(nullptr), 0);
  return check_launch("anemoi_col_sum");
}

// ---------------------------------------------------------------------------------------------
// dpre = dy * act'(pre)   (pre = the Linear's output before the activation, saved by the forward)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float act_grad(float x, int act) {
  switch (act) {
    case ANEMOI_ACT_GELU: {  // d/dx [x Phi(x)] = Phi(x) + x phi(x)
      const float cdf = 0.5f * (1.0f + fast_erf(x * 0.70710678118654752440f));  // |error| <= 1.5e-7 (common.hpp)
      return cdf + x * 0.39894228040143267794f * __expf(-0.5f * x * x);
    }
    case ANEMOI_ACT_SILU: {
      const float s = 1.0f / (1.0f + __expf(-x));
      return s * (1.0f + x * (1.0f - s));
    }
    case ANEMOI_ACT_RELU: return x > 0.f ? 1.f : 0.f;
    default: return 1.f;
  }
}

// 16 bytes per lane when every row pitch and the width allow it (the usual case: [M, 4C] hidden activations)
template <typename T>
__global__ __launch_bounds__(256) void act_backward_vec_kernel(const T* __restrict__ pre, int64_t ldp,
                                                               const T* __restrict__ dy, int64_t ldd,
                                                               T* __restrict__ out, int64_t ldo, int64_t rows, int cols,
                                                               int act) {
  constexpr int V = 16 / sizeof(T);
  const int per_row = cols / V;
  const int64_t total = rows * per_row;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = idx / per_row;
    const int c = (int)(idx - r * per_row) * V;
    float p[V], d[V];
    VecIO<T, V>::load(pre + r * ldp + c, p);
    VecIO<T, V>::load(dy + r * ldd + c, d);
#pragma unroll
    for (int i = 0; i < V; ++i) d[i] *= act_grad(p[i], act);
    VecIO<T, V>::store(out + r * ldo + c, d);
  }
}

// y = act(pre) (+ residual): the differentiable forward keeps `pre` and applies the activation in one pass over it
template <typename T>
__global__ __launch_bounds__(256) void act_forward_kernel(const T* __restrict__ pre, int64_t ldp,
                                                          const T* __restrict__ res, int64_t ldr, T* __restrict__ out,
                                                          int64_t ldo, int64_t rows, int cols, int act) {
  constexpr int V = 16 / sizeof(T);
  const int per_row = cols / V;
  const int64_t total = rows * per_row;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = idx / per_row;
    const int c = (int)(idx - r * per_row) * V;
    float p[V];
    VecIO<T, V>::load(pre + r * ldp + c, p);
#pragma unroll
    for (int i = 0; i < V; ++i) p[i] = act_apply(p[i], act);
    if (res != nullptr) {
      float q[V];
      VecIO<T, V>::load(res + r * ldr + c, q);
#pragma unroll
      for (int i = 0; i < V; ++i) p[i] += q[i];
    }
    VecIO<T, V>::store(out + r * ldo + c, p);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void act_backward_kernel(const T* __restrict__ pre, int64_t ldp,
                                                           const T* __restrict__ dy, int64_t ldd, T* __restrict__ out,
                                                           int64_t ldo, int64_t rows, int cols, int act) {
  const int64_t total = rows * cols;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = idx / cols;
    const int c = (int)(idx - r * cols);
    const float g = Elem<T>::load(dy + r * ldd + c) * act_grad(Elem<T>::load(pre + r * ldp + c), act);
    Elem<T>::store(out + r * ldo + c, g);
  }
}

// ---------------------------------------------------------------------------------------------
// LayerNorm backward.  xhat = x rstd + shift (stats = { rstd, -mean rstd } of the forward), g = dy gamma:
//   dx = rstd (g - mean_c(g) - xhat mean_c(g xhat));   d gamma = sum_r dy xhat;   d beta = sum_r dy
// One wave per row for dx; each workgroup (4 waves) also keeps the column partials of its ROWS_PER_WG rows and writes
// them to partial[wg][2][C]: the column reduction is finished by col_sum's second stage.
// ---------------------------------------------------------------------------------------------
template <typename T, int NC>  // NC > 0: the lane's C / 64 <= NC column partials live in registers (NC = 0: in LDS)
__global__ __launch_bounds__(256) void layer_norm_backward_kernel(const T* __restrict__ x, int64_t ldx,
                                                                  const float2* __restrict__ stats,
                                                                  const float* __restrict__ gamma,
                                                                  const T* __restrict__ dy, int64_t ldd,
                                                                  T* __restrict__ dx, int64_t ldo, int64_t rows, int C,
                                                                  int rows_per_wg, float* __restrict__ partial,
                                                                  const T* __restrict__ dres, int64_t ldr) {
  extern __shared__ float lds[];  // [4 waves][2][C]
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  float* my_dg = lds + (size_t)wid * 2 * C;
  float* my_db = my_dg + C;
  constexpr int NR = NC > 0 ? NC : 1;
  float rdg[NR], rdb[NR];
#pragma unroll
  for (int i = 0; i < NR; ++i) rdg[i] = rdb[i] = 0.f;
  if constexpr (NC == 0) {
    for (int c = lane; c < C; c += 64) {
      my_dg[c] = 0.f;
      my_db[c] = 0.f;
    }
  }
  const int64_t r_begin = (int64_t)blockIdx.x * rows_per_wg;
  const int64_t r_end = r_begin + rows_per_wg < rows ? r_begin + rows_per_wg : rows;
  const float inv_c = 1.0f / (float)C;
  for (int64_t r = r_begin + wid; r < r_end; r += 4) {
    const float2 st = stats[r];
    float sg = 0.f, sgx = 0.f;
    if constexpr (NC > 0) {
      // the lane owns V = 16 / sizeof(T) adjacent columns per group of 64 V columns: one 16-byte load per group and array
      constexpr int V = 16 / sizeof(T);
#pragma unroll
      for (int gi = 0; gi < NC / V; ++gi) {
        const int c = (gi * 64 + lane) * V;
        if (c < C) {
          float xv[V], dv[V], gv[V];
          VecIO<T, V>::load(x + r * ldx + c, xv);
          VecIO<T, V>::load(dy + r * ldd + c, dv);
          VecIO<float, V>::load(gamma + c, gv);
#pragma unroll
          for (int i = 0; i < V; ++i) {
            const float xh = xv[i] * st.x + st.y;
            const float g = dv[i] * gv[i];
            sg += g;
            sgx = fmaf(g, xh, sgx);
            rdg[gi * V + i] = fmaf(dv[i], xh, rdg[gi * V + i]);
            rdb[gi * V + i] += dv[i];
          }
        }
      }
    } else {
      for (int c = lane; c < C; c += 64) {
        const float xh = Elem<T>::load(x + r * ldx + c) * st.x + st.y;
        const float d = Elem<T>::load(dy + r * ldd + c);
        const float g = d * gamma[c];
        sg += g;
        sgx = fmaf(g, xh, sgx);
        my_dg[c] = fmaf(d, xh, my_dg[c]);  // lane-private column slots: no race inside the wave
        my_db[c] += d;
      }
    }
    sg = wave_sum(sg) * inv_c;
    sgx = wave_sum(sgx) * inv_c;
    if constexpr (NC > 0) {
      constexpr int V = 16 / sizeof(T);
#pragma unroll
      for (int gi = 0; gi < NC / V; ++gi) {
        const int c = (gi * 64 + lane) * V;
        if (c < C) {
          float xv[V], dv[V], gv[V], o[V];
          VecIO<T, V>::load(x + r * ldx + c, xv);
          VecIO<T, V>::load(dy + r * ldd + c, dv);
          VecIO<float, V>::load(gamma + c, gv);
#pragma unroll
          for (int i = 0; i < V; ++i) {
            const float xh = xv[i] * st.x + st.y;
            o[i] = st.x * (dv[i] * gv[i] - sg - xh * sgx);
          }
          if (dres != nullptr) {  // + the gradient that reaches x through the skip connection around the LayerNorm
            float rv[V];
            VecIO<T, V>::load(dres + r * ldr + c, rv);
#pragma unroll
            for (int i = 0; i < V; ++i) o[i] += rv[i];
          }
          VecIO<T, V>::store(dx + r * ldo + c, o);
        }
      }
    } else {
      for (int c = lane; c < C; c += 64) {
        const float xh = Elem<T>::load(x + r * ldx + c) * st.x + st.y;
        const float g = Elem<T>::load(dy + r * ldd + c) * gamma[c];
        Elem<T>::store(dx + r * ldo + c, st.x * (g - sg - xh * sgx) + (dres != nullptr ? Elem<T>::load(dres + r * ldr + c) : 0.f));
      }
    }
  }
  if constexpr (NC > 0) {
    constexpr int V = 16 / sizeof(T);
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const int c = ((i / V) * 64 + lane) * V + (i % V);
      if (c < C) {
        my_dg[c] = rdg[i];
        my_db[c] = rdb[i];
      }
    }
  }
  __syncthreads();
  float* out = partial + (size_t)blockIdx.x * 2 * C;
  for (int c = threadIdx.x; c < 2 * C; c += 256)
    out[c] = lds[c] + lds[2 * C + c] + lds[4 * C + c] + lds[6 * C + c];
}

template <typename T>
static int layer_norm_backward_launch(const T* x, int64_t ldx, const float2* stats, const float* gamma, const T* dy,
                                      int64_t ldd, T* dx, int64_t ldo, int64_t rows, int C, float* dgamma, float* dbeta,
                                      float* workspace, int64_t workspace_floats, hipStream_t st, const T* dres, int64_t ldr) {
  ANEMOI_REQUIRE((size_t)C * 8 * sizeof(float) <= 160 * 1024, ANEMOI_ERR_UNSUPPORTED,
                 "anemoi_layer_norm_backward: C = %d too wide for the LDS column partials", C);
  int rows_per_wg = (int)((rows + 1023) / 1024);
  if (rows_per_wg < 32) rows_per_wg = 32;
  const int64_t wgs = (rows + rows_per_wg - 1) / rows_per_wg;
  ANEMOI_REQUIRE(workspace != nullptr && workspace_floats >= wgs * 2 * C + 2 * C, ANEMOI_ERR_INVALID,
                 "anemoi_layer_norm_backward: workspace of %lld floats required", (long long)(wgs * 2 * C + 2 * C));
  const size_t lds_bytes = (size_t)C * 8 * sizeof(float);
  auto launch = [&](auto kern) -> int {
    if (lds_bytes > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds_bytes) != hipSuccess)
      return fail(ANEMOI_ERR_LAUNCH, "anemoi_layer_norm_backward: cannot raise the dynamic LDS limit");
    hipLaunchKernelGGL(kern, dim3((unsigned)wgs), dim3(256), lds_bytes, st, x, ldx, stats, gamma, dy, ldd, dx, ldo, rows,
                       C, rows_per_wg, workspace, dres, ldr);
    return ANEMOI_OK;
  };
  int rc;
  constexpr int V = 16 / sizeof(T);
  const bool vec_ok = C % V == 0 && ldx % V == 0 && ldd % V == 0 && ldo % V == 0 && (uintptr_t)x % 16 == 0 &&
                      (uintptr_t)dy % 16 == 0 && (uintptr_t)dx % 16 == 0 && (uintptr_t)gamma % 16 == 0 &&
                      (dres == nullptr || (ldr % V == 0 && (uintptr_t)dres % 16 == 0));
  if (!vec_ok) rc = launch(layer_norm_backward_kernel<T, 0>);
  else if (C <= 64 * 8) rc = launch(layer_norm_backward_kernel<T, 8>);
  else if (C <= 64 * 16) rc = launch(layer_norm_backward_kernel<T, 16>);
  else if (C <= 64 * 32) rc = launch(layer_norm_backward_kernel<T, 32>);
  else rc = launch(layer_norm_backward_kernel<T, 0>);
  if (rc != ANEMOI_OK) return rc;
  // [wgs, 2C] partials -> d gamma | d beta
  hipLaunchKernelGGL(col_sum_final_kernel, dim3((unsigned)((2 * C + 15) / 16)), dim3(256), 0, st, workspace, (int64_t)wgs,
                     2 * C, dgamma, dbeta, C);
  return check_launch("anemoi_layer_norm_backward");
}

static inline hipStream_t bw_stream(anemoi_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
static inline unsigned bw_grid(int64_t total) {
  int64_t blocks = (total + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  return (unsigned)(blocks < 1 ? 1 : blocks);
}

}  // namespace anemoi

using namespace anemoi;

extern "C" {

int anemoi_transpose(int dtype, const void* src, int64_t ld_src, void* dst, int64_t ld_dst, int64_t rows, int cols,
                     anemoi_stream_t stream) {
  ANEMOI_REQUIRE(src && dst && rows >= 0 && cols > 0 && ld_src >= cols && ld_dst >= rows, ANEMOI_ERR_INVALID,
                 "anemoi_transpose: bad argument");
  if (ld_dst == 0) return ANEMOI_OK;
  const dim3 grid((unsigned)((ld_dst + 63) / 64), (unsigned)((cols + 63) / 64));
  if (dtype == ANEMOI_F32)
    hipLaunchKernelGGL((transpose_kernel<float>), grid, dim3(256), 0, bw_stream(stream), static_cast<const float*>(src),
                       ld_src, static_cast<float*>(dst), ld_dst, rows, cols, rows);
  else if (dtype == ANEMOI_BF16)
    hipLaunchKernelGGL(transpose_b16_kernel, dim3((unsigned)((ld_dst + 255) / 256), grid.y), dim3(256), 0,
                       bw_stream(stream), static_cast<const uint16_t*>(src), ld_src, static_cast<uint16_t*>(dst), ld_dst,
                       rows, cols, rows, static_cast<float*>(nullptr), 0);
  else
    return fail(ANEMOI_ERR_UNSUPPORTED, "anemoi_transpose: dtype %d", dtype);
  return check_launch("anemoi_transpose");
}

int64_t anemoi_transpose_colsum_rows(int64_t rows, int64_t chunk_rows) {
  if (rows <= 0 || chunk_rows <= 0) return 0;
  return ((rows + chunk_rows - 1) / chunk_rows) * ((chunk_rows + 63) / 64);
}

int anemoi_transpose_chunked(int dtype, const void* src, int64_t ld_src, void* dst, int64_t ld_dst, int64_t rows, int cols,
                             int64_t chunk_rows, float* colsum_partial, anemoi_stream_t stream) {
  ANEMOI_REQUIRE(src && dst && rows > 0 && cols > 0 && chunk_rows > 0 && ld_src >= cols && ld_dst >= chunk_rows,
                 ANEMOI_ERR_INVALID, "anemoi_transpose_chunked: bad argument");
  ANEMOI_REQUIRE(colsum_partial == nullptr || (dtype == ANEMOI_BF16 && cols % 4 == 0 && (uintptr_t)colsum_partial % 16 == 0),
                 ANEMOI_ERR_UNSUPPORTED, "anemoi_transpose_chunked: column-sum partials need bf16 and cols % 4 == 0");
  const int64_t chunks = (rows + chunk_rows - 1) / chunk_rows;
  ANEMOI_REQUIRE(chunks < 65536, ANEMOI_ERR_UNSUPPORTED, "anemoi_transpose_chunked: too many chunks");
  const dim3 grid((unsigned)((ld_dst + 63) / 64), (unsigned)((cols + 63) / 64), (unsigned)chunks);
  if (dtype == ANEMOI_F32)
    hipLaunchKernelGGL((transpose_kernel<float>), grid, dim3(256), 0, bw_stream(stream), static_cast<const float*>(src),
                       ld_src, static_cast<float*>(dst), ld_dst, rows, cols, chunk_rows);
  else if (dtype == ANEMOI_BF16)
    hipLaunchKernelGGL(transpose_b16_kernel, dim3((unsigned)((ld_dst + 255) / 256), grid.y, grid.z), dim3(256), 0,
                       bw_stream(stream), static_cast<const uint16_t*>(src), ld_src, static_cast<uint16_t*>(dst), ld_dst,
                       rows, cols, chunk_rows, colsum_partial, (int)((chunk_rows + 63) / 64));
  else
    return fail(ANEMOI_ERR_UNSUPPORTED, "anemoi_transpose_chunked: dtype %d", dtype);
  return check_launch("anemoi_transpose_chunked");
}

int64_t anemoi_col_sum_workspace_floats(int64_t rows, int cols) {
  int64_t chunk = (rows + 511) / 512;
  if (chunk < 16) chunk = 16;
  const int64_t blocks = (rows + chunk - 1) / chunk;
  return blocks <= 1 ? 0 : blocks * cols;
}

int anemoi_col_sum(int dtype, const void* x, int64_t ldx, int64_t rows, int cols, float* out, float* workspace,
                   int64_t workspace_floats, anemoi_stream_t stream) {
  ANEMOI_REQUIRE(x && out && rows >= 0 && cols > 0 && ldx >= cols, ANEMOI_ERR_INVALID, "anemoi_col_sum: bad argument");
  if (dtype == ANEMOI_F32)
    return col_sum_launch<float>(static_cast<const float*>(x), ldx, rows, cols, out, workspace, workspace_floats,
                                 bw_stream(stream));
  if (dtype == ANEMOI_BF16)
    return col_sum_launch<bf16_t>(static_cast<const bf16_t*>(x), ldx, rows, cols, out, workspace, workspace_floats,
                                  bw_stream(stream));
  return fail(ANEMOI_ERR_UNSUPPORTED, "anemoi_col_sum: dtype %d", dtype);
}

int anemoi_act_backward(int dtype, int act, const void* pre, int64_t ldp, const void* dy, int64_t ldd, void* out,
                        int64_t ldo, int64_t rows, int cols, anemoi_stream_t stream) {
  ANEMOI_REQUIRE(pre && dy && out && rows >= 0 && cols > 0 && ldp >= cols && ldd >= cols && ldo >= cols,
                 ANEMOI_ERR_INVALID, "anemoi_act_backward: bad argument");
  ANEMOI_REQUIRE(act >= ANEMOI_ACT_NONE && act <= ANEMOI_ACT_RELU, ANEMOI_ERR_INVALID, "anemoi_act_backward: act %d", act);
  if (rows == 0) return ANEMOI_OK;
  const int esz = dtype == ANEMOI_BF16 ? 2 : 4, vec = 16 / esz;
  const bool vec_ok = cols % vec == 0 && ldp % vec == 0 && ldd % vec == 0 && ldo % vec == 0 && (uintptr_t)pre % 16 == 0 &&
                      (uintptr_t)dy % 16 == 0 && (uintptr_t)out % 16 == 0;
  if (vec_ok && dtype == ANEMOI_F32)
    hipLaunchKernelGGL((act_backward_vec_kernel<float>), dim3(bw_grid(rows * cols / vec)), dim3(256), 0, bw_stream(stream),
                       static_cast<const float*>(pre), ldp, static_cast<const float*>(dy), ldd, static_cast<float*>(out),
                       ldo, rows, cols, act);
  else if (vec_ok && dtype == ANEMOI_BF16)
    hipLaunchKernelGGL((act_backward_vec_kernel<bf16_t>), dim3(bw_grid(rows * cols / vec)), dim3(256), 0,
                       bw_stream(stream), static_cast<const bf16_t*>(pre), ldp, static_cast<const bf16_t*>(dy), ldd,
                       static_cast<bf16_t*>(out), ldo, rows, cols, act);
  else if (dtype == ANEMOI_F32)
    hipLaunchKernelGGL((act_backward_kernel<float>), dim3(bw_grid(rows * cols)), dim3(256), 0, bw_stream(stream),
                       static_cast<const float*>(pre), ldp, static_cast<const float*>(dy), ldd, static_cast<float*>(out),
                       ldo, rows, cols, act);
  else if (dtype == ANEMOI_BF16)
    hipLaunchKernelGGL((act_backward_kernel<bf16_t>), dim3(bw_grid(rows * cols)), dim3(256), 0, bw_stream(stream),
                       static_cast<const bf16_t*>(pre), ldp, static_cast<const bf16_t*>(dy), ldd,
                       static_cast<bf16_t*>(out), ldo, rows, cols, act);
  else
    return fail(ANEMOI_ERR_UNSUPPORTED, "anemoi_act_backward: dtype %d", dtype);
  return check_launch("anemoi_act_backward");
}

int anemoi_act_forward(int dtype, int act, const void* pre, int64_t ldp, const void* residual, int64_t ldr, void* out,
                       int64_t ldo, int64_t rows, int cols, anemoi_stream_t stream) {
  ANEMOI_REQUIRE(pre && out && rows >= 0 && cols > 0 && ldp >= cols && ldo >= cols && (residual == nullptr || ldr >= cols),
                 ANEMOI_ERR_INVALID, "anemoi_act_forward: bad argument");
  ANEMOI_REQUIRE(act >= ANEMOI_ACT_NONE && act <= ANEMOI_ACT_RELU, ANEMOI_ERR_INVALID, "anemoi_act_forward: act %d", act);
  if (rows == 0) return ANEMOI_OK;
  const int esz = dtype == ANEMOI_BF16 ? 2 : 4, vec = 16 / esz;
  ANEMOI_REQUIRE(cols % vec == 0 && ldp % vec == 0 && ldo % vec == 0 && (uintptr_t)pre % 16 == 0 &&
                     (uintptr_t)out % 16 == 0 && (residual == nullptr || (ldr % vec == 0 && (uintptr_t)residual % 16 == 0)),
                 ANEMOI_ERR_UNSUPPORTED, "anemoi_act_forward: rows must be 16-byte aligned multiples");
  if (dtype == ANEMOI_F32)
    hipLaunchKernelGGL((act_forward_kernel<float>), dim3(bw_grid(rows * cols / vec)), dim3(256), 0, bw_stream(stream),
                       static_cast<const float*>(pre), ldp, static_cast<const float*>(residual), ldr,
                       static_cast<float*>(out), ldo, rows, cols, act);
  else if (dtype == ANEMOI_BF16)
    hipLaunchKernelGGL((act_forward_kernel<bf16_t>), dim3(bw_grid(rows * cols / vec)), dim3(256), 0, bw_stream(stream),
                       static_cast<const bf16_t*>(pre), ldp, static_cast<const bf16_t*>(residual), ldr,
                       static_cast<bf16_t*>(out), ldo, rows, cols, act);
  else
    return fail(ANEMOI_ERR_UNSUPPORTED, "anemoi_act_forward: dtype %d", dtype);
  return check_launch("anemoi_act_forward");
}

int64_t anemoi_layer_norm_backward_workspace_floats(int64_t rows, int C) {
  int64_t rows_per_wg = (rows + 1023) / 1024;
  if (rows_per_wg < 32) rows_per_wg = 32;
  const int64_t wgs = (rows + rows_per_wg - 1) / rows_per_wg;
  return wgs * 2 * C + 2 * C;
}

int anemoi_layer_norm_backward(int dtype, const void* x, int64_t ldx, const float* stats, const float* gamma,
                               const void* dy, int64_t ldd, const void* dres, int64_t ldr, void* dx, int64_t ldo,
                               int64_t rows, int C, float* dgamma, float* dbeta, float* workspace,
                               int64_t workspace_floats, anemoi_stream_t stream) {
  ANEMOI_REQUIRE(x && stats && gamma && dy && dx && dgamma && dbeta && rows > 0 && C > 0 && ldx >= C && ldd >= C &&
                     ldo >= C && (dres == nullptr || ldr >= C),
                 ANEMOI_ERR_INVALID, "anemoi_layer_norm_backward: bad argument");
  ANEMOI_REQUIRE((uintptr_t)stats % 8 == 0, ANEMOI_ERR_INVALID, "anemoi_layer_norm_backward: stats must be 8-byte aligned");
  if (dtype == ANEMOI_F32)
    return layer_norm_backward_launch<float>(static_cast<const float*>(x), ldx, reinterpret_cast<const float2*>(stats),
                                             gamma, static_cast<const float*>(dy), ldd, static_cast<float*>(dx), ldo,
                                             rows, C, dgamma, dbeta, workspace, workspace_floats, bw_stream(stream),
                                             static_cast<const float*>(dres), ldr);
  if (dtype == ANEMOI_BF16)
    return layer_norm_backward_launch<bf16_t>(static_cast<const bf16_t*>(x), ldx,
                                              reinterpret_cast<const float2*>(stats), gamma,
                                              static_cast<const bf16_t*>(dy), ldd, static_cast<bf16_t*>(dx), ldo, rows,
                                              C, dgamma, dbeta, workspace, workspace_floats, bw_stream(stream),
                                              static_cast<const bf16_t*>(dres), ldr);
  return fail(ANEMOI_ERR_UNSUPPORTED, "anemoi_layer_norm_backward: dtype %d", dtype);
}

}  // extern "C"
