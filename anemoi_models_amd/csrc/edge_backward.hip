// Backward of the folded GraphTransformer edge phase (anemoi_gt_edge_attention_folded), SURVEY.md section 8f-1.
//
// Forward, per destination i, head h, in-edge e = (j -> i) with attributes a_e (constant-1 column included):
//     s_e   = scale (q_i,h . k_j,h + u_i,h . a_e),   alpha_e = softmax_e(s_e) = exp(s_e - lse_i,h),
//     out_i,h = sum_e alpha_e v_j,h (+ x_r),         t_i,h = sum_e alpha_e a_e.
// Backward, given dout (w.r.t. out), dt (w.r.t. t) and the forward's lse:
//     dalpha_e = dout_i,h . v_j,h + dt_i,h . a_e,    w_e = alpha_e dalpha_e,    Dsum_i,h = sum_e w_e,
//     ds_e = w_e - alpha_e Dsum_i,h,
//     dq_i,h = scale sum_e ds_e k_j,h = scale (sum_e w_e k_j,h - Dsum sum_e alpha_e k_j,h),   du likewise with a_e,
//     dk_j,h = scale sum_{e from j} ds_e q_i,h,   dv_j,h = sum_{e from j} alpha_e dout_i,h.
// Dsum is accumulated in f32 over the in-edges -- NOT rebuilt as dout . (out - x_r) from the forward's rounded result (in
// bf16 that difference cancels against the residual) -- which is why the destination-major kernel carries the two sums
// A = sum w k and B = sum alpha k instead of sum ds k: ONE sweep over the in-edges, built like the forward kernel (same
// XCD-contiguous work mapping, U = 4 edges = eight 16-byte gathers per lane in flight, lane-shared edge attributes,
// packed-f32 accumulation), alpha rebuilt from lse.  It leaves alpha[E, H], w[E, H] and Dsum[n_dst, H] (f32) for
//   * the source-major kernel (transposed CSR; edge ids into the forward's order): dk, dv as gathers of q_i / dout_i,
//   * the edge-attribute gradient (trainable edge tensor included),
// which finish ds_e = w_e - alpha_e Dsum on the fly.  No atomics anywhere: gradients are reproducible bit for bit.
// (The first version swept the in-edges twice, one edge at a time, re-deriving max / sum: 1.0 ms per mesh block at
// config 3 against 0.15 ms of the forward.)
#include <cstdlib>

#include "common.hpp"
#include "edge_common.hpp"

namespace anemoi {

struct EdgeBwdParams {
  const void* q;     // [n_dst, ldq]
  const void* k;     // [n_src, ldkv]
  const void* v;
  const void* dout;  // [n_dst, ldd]
  const void* u;     // [n_dst, ldu]   (H * UP columns)
  const void* dt;    // [n_dst, lddt]  (H * UP columns)
  const float* lse;  // [n_dst, H]
  float* alpha;      // [E, H]
  float* w;          // [E, H]
  float* dsum;       // [n_dst, H]
  void* dq;          // [n_dst, lddq]
  void* du;          // [n_dst, lddu]  (H * UP columns)
  void* dxr;         // optional [n_dst, lddxr]: a copy of dout (the gradient of the self term x_r, out = sum alpha v + x_r)
  int64_t ldq, ldkv, ldd, ldu, lddt, lddq, lddu, lddxr;
  int64_t n_dst;
  int C, H, n_slices;
  float scale;
};

typedef __attribute__((ext_vector_type(2))) float f32x2_t;

template <typename T, int VEC, int LPH, int UP, int U>
__global__ __launch_bounds__(256) void gt_edge_bwd_dst_kernel(const EdgeBwdParams p, const float* __restrict__ attr_,
                                                              const int32_t* __restrict__ rowptr_,
                                                              const int32_t* __restrict__ col_) {
  using Raw = typename RawVec<T, VEC>::type;
  constexpr int APL = attrs_per_lane(UP, LPH);
  constexpr int VP = (VEC + 1) / 2;
  const int lane = threadIdx.x & 63;
  const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int xcd = blockIdx.x & 7;
  const int wave_in_xcd = (int)(blockIdx.x >> 3) * 4 + wib;
  const int waves_per_xcd = (int)(gridDim.x >> 3) * 4;
  const int slice = wave_in_xcd % p.n_slices;
  const int64_t node_first = wave_in_xcd / p.n_slices;
  const int64_t node_stride = waves_per_xcd / p.n_slices;
  const int64_t n0 = p.n_dst * xcd / 8, n1 = p.n_dst * (xcd + 1) / 8;

  const int lanes_total = p.C / VEC;
  const int gl = slice * 64 + lane;
  const bool active = gl < lanes_total;
  const int gls = active ? gl : 0;
  const int c0 = gls * VEC;
  const int head = gls / LPH;
  const bool writer = active && (gls % LPH) == 0;
  const int a0 = (gls % LPH) * APL;  // first attribute of this lane (the lanes of a head share the attributes)
  const bool a_own = a0 < UP;
  const int a_ld = a_own ? a0 : 0;
  const float amask = a_own ? 1.f : 0.f;

  const T* qb = static_cast<const T*>(p.q) + c0;
  const T* kb = static_cast<const T*>(p.k) + c0;
  const T* vb = static_cast<const T*>(p.v) + c0;
  const T* dob = static_cast<const T*>(p.dout) + c0;
  const T* ub = static_cast<const T*>(p.u) + head * UP + a_ld;
  const T* dtb = static_cast<const T*>(p.dt) + head * UP + a_ld;
  const float* ab = attr_ + a_ld;

  for (int64_t node = n0 + node_first; node < n1; node += node_stride) {
    const int e_begin = rowptr_[node], e_end = rowptr_[node + 1];
    QK<T, VEC> qk, dok;
    float u[APL], dtl[APL];
    {
      float f[VEC];
      VecIO<T, VEC>::load(qb + node * p.ldq, f);
      qk.set(f);
      VecIO<T, VEC>::load(dob + node * p.ldd, f);
      dok.set(f);
      if (p.dxr != nullptr && active) VecIO<T, VEC>::store(static_cast<T*>(p.dxr) + node * p.lddxr + c0, f);
      VecIO<T, APL>::load(ub + node * p.ldu, u);
      VecIO<T, APL>::load(dtb + node * p.lddt, dtl);
#pragma unroll
      for (int i = 0; i < APL; ++i) {
        u[i] *= amask;
        dtl[i] *= amask;
      }
    }
    const float lse = p.lse[node * p.H + head];
    f32x2_t ak[VP], bk[VP];
    float au[APL], bu[APL], dsum = 0.f;
#pragma unroll
    for (int i = 0; i < VP; ++i) ak[i] = bk[i] = f32x2_t{0.f, 0.f};
#pragma unroll
    for (int a = 0; a < APL; ++a) au[a] = bu[a] = 0.f;

    for (int e = e_begin; e < e_end; e += U) {
      Raw kr[U], vr[U];
      float at[U][APL];
#pragma unroll
      for (int uu = 0; uu < U; ++uu) {
        if (e + uu < e_end) {
          const int64_t j = col_[e + uu];
          kr[uu] = *reinterpret_cast<const Raw*>(kb + j * p.ldkv);
          vr[uu] = *reinterpret_cast<const Raw*>(vb + j * p.ldkv);
          VecIO<float, APL>::load(ab + (int64_t)(e + uu) * UP, at[uu]);
        }
      }
#pragma unroll
      for (int uu = 0; uu < U; ++uu) {
        if (e + uu < e_end) {
          float ts = qk.dot(kr[uu]), td = dok.dot(vr[uu]);
#pragma unroll
          for (int a = 0; a < APL; ++a) {
            ts = fmaf(u[a], at[uu][a], ts);
            td = fmaf(dtl[a], at[uu][a], td);
          }
          const float s = group_sum<LPH>(ts) * p.scale;
          const float da = group_sum<LPH>(td);
          const float alpha = __expf(s - lse);
          const float w = alpha * da;
          dsum += w;
          if (writer) {
            p.alpha[(int64_t)(e + uu) * p.H + head] = alpha;
            p.w[(int64_t)(e + uu) * p.H + head] = w;
          }
          float kk[VEC];
          unpack<T, VEC>(kr[uu], kk);
#pragma unroll
          for (int i = 0; i < VP; ++i) {
            const f32x2_t k2 = f32x2_t{kk[2 * i], 2 * i + 1 < VEC ? kk[2 * i + 1] : 0.f};
            ak[i] = __builtin_elementwise_fma(f32x2_t{w, w}, k2, ak[i]);
            bk[i] = __builtin_elementwise_fma(f32x2_t{alpha, alpha}, k2, bk[i]);
          }
#pragma unroll
          for (int a = 0; a < APL; ++a) {
            au[a] = fmaf(w, at[uu][a], au[a]);
            bu[a] = fmaf(alpha, at[uu][a], bu[a]);
          }
        }
      }
    }

    float dq[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) dq[i] = (ak[i >> 1][i & 1] - dsum * bk[i >> 1][i & 1]) * p.scale;
    if (active) VecIO<T, VEC>::store(static_cast<T*>(p.dq) + node * p.lddq + c0, dq);
    if (active && a_own) {
      float d4[APL];
#pragma unroll
      for (int a = 0; a < APL; ++a) d4[a] = (au[a] - dsum * bu[a]) * p.scale;
      VecIO<T, APL>::store(static_cast<T*>(p.du) + node * p.lddu + head * UP + a0, d4);
    }
    if (writer) p.dsum[node * p.H + head] = dsum;
  }
}

struct EdgeBwdSrcParams {
  const void* q;     // [n_dst, ldq]
  const void* dout;  // [n_dst, ldd]
  const float* alpha;
  const float* w;
  const float* dsum;
  void* dk;          // [n_src, ldg]
  void* dv;
  int64_t ldq, ldd, ldg;
  int64_t n_src;
  int C, H, n_slices;
  float scale;
};

// rowptr_t [n_src + 1]: transposed CSR; eid_t: forward CSR position of every out-edge; dst_t: its destination node
template <typename T, int VEC, int LPH, int U>
__global__ __launch_bounds__(256) void gt_edge_bwd_src_kernel(const EdgeBwdSrcParams p,
                                                              const int32_t* __restrict__ rowptr_t,
                                                              const int32_t* __restrict__ eid_t,
                                                              const int32_t* __restrict__ dst_t) {
  using Raw = typename RawVec<T, VEC>::type;
  constexpr int VP = (VEC + 1) / 2;
  const int lane = threadIdx.x & 63;
  const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int xcd = blockIdx.x & 7;
  const int wave_in_xcd = (int)(blockIdx.x >> 3) * 4 + wib;
  const int waves_per_xcd = (int)(gridDim.x >> 3) * 4;
  const int slice = wave_in_xcd % p.n_slices;
  const int64_t node_first = wave_in_xcd / p.n_slices;
  const int64_t node_stride = waves_per_xcd / p.n_slices;
  const int64_t n0 = p.n_src * xcd / 8, n1 = p.n_src * (xcd + 1) / 8;

  const int lanes_total = p.C / VEC;
  const int gl = slice * 64 + lane;
  const bool active = gl < lanes_total;
  const int gls = active ? gl : 0;
  const int c0 = gls * VEC;
  const int head = gls / LPH;
  const T* qb = static_cast<const T*>(p.q) + c0;
  const T* dob = static_cast<const T*>(p.dout) + c0;

  for (int64_t node = n0 + node_first; node < n1; node += node_stride) {
    const int t_begin = rowptr_t[node], t_end = rowptr_t[node + 1];
    f32x2_t dk[VP], dv[VP];
#pragma unroll
    for (int i = 0; i < VP; ++i) dk[i] = dv[i] = f32x2_t{0.f, 0.f};
    for (int t = t_begin; t < t_end; t += U) {
      Raw qr[U], dor[U];
      float al[U], ww[U], dsm[U];
#pragma unroll
      for (int uu = 0; uu < U; ++uu) {
        if (t + uu < t_end) {
          const int64_t e = eid_t[t + uu], i_dst = dst_t[t + uu];
          qr[uu] = *reinterpret_cast<const Raw*>(qb + i_dst * p.ldq);
          dor[uu] = *reinterpret_cast<const Raw*>(dob + i_dst * p.ldd);
          al[uu] = p.alpha[e * p.H + head];
          ww[uu] = p.w[e * p.H + head];
          dsm[uu] = p.dsum[i_dst * p.H + head];
        }
      }
#pragma unroll
      for (int uu = 0; uu < U; ++uu) {
        if (t + uu < t_end) {
          const float dse = fmaf(-al[uu], dsm[uu], ww[uu]);  // ds_e = w_e - alpha_e Dsum_i,h
          float qf[VEC], dof[VEC];
          unpack<T, VEC>(qr[uu], qf);
          unpack<T, VEC>(dor[uu], dof);
#pragma unroll
          for (int i = 0; i < VP; ++i) {
            dk[i] = __builtin_elementwise_fma(f32x2_t{dse, dse}, f32x2_t{qf[2 * i], 2 * i + 1 < VEC ? qf[2 * i + 1] : 0.f},
                                              dk[i]);
            dv[i] = __builtin_elementwise_fma(f32x2_t{al[uu], al[uu]},
                                              f32x2_t{dof[2 * i], 2 * i + 1 < VEC ? dof[2 * i + 1] : 0.f}, dv[i]);
          }
        }
      }
    }
    if (active) {
      float o[VEC];
#pragma unroll
      for (int i = 0; i < VEC; ++i) o[i] = dk[i >> 1][i & 1] * p.scale;
      VecIO<T, VEC>::store(static_cast<T*>(p.dk) + node * p.ldg + c0, o);
#pragma unroll
      for (int i = 0; i < VEC; ++i) o[i] = dv[i >> 1][i & 1];
      VecIO<T, VEC>::store(static_cast<T*>(p.dv) + node * p.ldg + c0, o);
    }
  }
}

// d attr[e, a] = sum_h ( scale ds[e, h] u[dst(e), h, a] + alpha[e, h] dt[dst(e), h, a] ), ds = w - alpha Dsum: the gradient
// of the edge attributes (trainable edge tensor included) from what the destination-major kernel left per (edge, head).
template <typename T>
__global__ __launch_bounds__(256) void edge_attr_grad_kernel(const float* __restrict__ alpha, const float* __restrict__ w,
                                                             const float* __restrict__ dsum, const T* __restrict__ u,
                                                             int64_t ldu, const T* __restrict__ dt, int64_t lddt,
                                                             const int32_t* __restrict__ dst_of_edge,
                                                             float* __restrict__ dattr, int64_t n_edges, int H, int UP,
                                                             float scale) {
  const int64_t total = n_edges * UP;
  const bool fits32 = total < ((int64_t)1 << 31);
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    int64_t e, a64;
    fast_divmod(idx, UP, fits32, e, a64);
    const int a = (int)a64;
    const int64_t i = dst_of_edge[e];
    float acc = 0.f;
    for (int h = 0; h < H; ++h) {
      const float al = alpha[e * H + h];
      const float ds = fmaf(-al, dsum[i * H + h], w[e * H + h]);
      acc = fmaf(scale * ds, Elem<T>::load(u + i * ldu + h * UP + a), acc);
      acc = fmaf(al, Elem<T>::load(dt + i * lddt + h * UP + a), acc);
    }
    dattr[idx] = acc;
  }
}

// ---------------------------------------------------------------------------------------------
// Backward of anemoi_gt_conv (explicit per-edge features e_ij [E, C] in CSR order -- the route for edge_dim values the
// folded kernels do not cover): with k'_e = k_j + e_e, v'_e = v_j + e_e the formulas above hold with k', v' in place of
// k, v and no attribute terms; in addition  d e_e = alpha_e dout_i + scale ds_e q_i  -- exactly the per-edge terms of
// dv_j and dk_j, so the source-major kernel stores them on its way ([E, C], CSR position eid_t).
// DROP (the conv's dropout in training mode, out_i = sum_e alpha_e m_e v'_e with m_e = keep_e / (1 - p) rebuilt from the
// seed): d v'_e = alpha_e m_e dout_i and d alpha_e = m_e (dout_i . v'_e), so w_e = alpha_e m_e (dout_i . v'_e) and everything
// behind w -- dsum, ds_e, dq, dk' -- is unchanged.
// ---------------------------------------------------------------------------------------------
struct ConvBwdParams {
  const void* q;
  const void* k;
  const void* v;
  const void* e;
  const void* dout;
  const float* lse;
  float* alpha;
  float* w;
  float* dsum;
  void* dq;
  int64_t ldq, ldkv, lde, ldd, lddq;
  int64_t n_dst;
  int C, H, n_slices;
  float scale;
  EdgeDropout drop;
};

template <typename T, int VEC, int LPH, int U, bool DROP>
__global__ __launch_bounds__(256) void gt_conv_bwd_dst_kernel(const ConvBwdParams p, const int32_t* __restrict__ rowptr_,
                                                              const int32_t* __restrict__ col_) {
  using Raw = typename RawVec<T, VEC>::type;
  const int lane = threadIdx.x & 63;
  const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int xcd = blockIdx.x & 7;
  const int wave_in_xcd = (int)(blockIdx.x >> 3) * 4 + wib;
  const int waves_per_xcd = (int)(gridDim.x >> 3) * 4;
  const int slice = wave_in_xcd % p.n_slices;
  const int64_t node_first = wave_in_xcd / p.n_slices;
  const int64_t node_stride = waves_per_xcd / p.n_slices;
  const int64_t n0 = p.n_dst * xcd / 8, n1 = p.n_dst * (xcd + 1) / 8;
  const int lanes_total = p.C / VEC;
  const int gl = slice * 64 + lane;
  const bool active = gl < lanes_total;
  const int gls = active ? gl : 0;
  const int c0 = gls * VEC;
  const int head = gls / LPH;
  const bool writer = active && (gls % LPH) == 0;
  const T* qb = static_cast<const T*>(p.q) + c0;
  const T* kb = static_cast<const T*>(p.k) + c0;
  const T* vb = static_cast<const T*>(p.v) + c0;
  const T* eb = static_cast<const T*>(p.e) + c0;
  const T* dob = static_cast<const T*>(p.dout) + c0;
  const uint32_t dseed = DROP ? edge_dropout_seed(p.drop) : 0u;
  for (int64_t node = n0 + node_first; node < n1; node += node_stride) {
    const int e_begin = rowptr_[node], e_end = rowptr_[node + 1];
    float qf[VEC], dof[VEC], ak[VEC], bk[VEC], dsum = 0.f;
    VecIO<T, VEC>::load(qb + node * p.ldq, qf);
    VecIO<T, VEC>::load(dob + node * p.ldd, dof);
#pragma unroll
    for (int i = 0; i < VEC; ++i) ak[i] = bk[i] = 0.f;
    const float lse = p.lse[node * p.H + head];
    for (int e = e_begin; e < e_end; e += U) {
      Raw kr[U], vr[U], er[U];
#pragma unroll
      for (int uu = 0; uu < U; ++uu) {
        if (e + uu < e_end) {
          const int64_t j = col_[e + uu];
          kr[uu] = *reinterpret_cast<const Raw*>(kb + j * p.ldkv);
          vr[uu] = *reinterpret_cast<const Raw*>(vb + j * p.ldkv);
          er[uu] = *reinterpret_cast<const Raw*>(eb + (int64_t)(e + uu) * p.lde);
        }
      }
#pragma unroll
      for (int uu = 0; uu < U; ++uu) {
        if (e + uu < e_end) {
          float kk[VEC], vv[VEC], ee[VEC];
          unpack<T, VEC>(kr[uu], kk);
          unpack<T, VEC>(vr[uu], vv);
          unpack<T, VEC>(er[uu], ee);
          float ts = 0.f, td = 0.f;
#pragma unroll
          for (int i = 0; i < VEC; ++i) {
            kk[i] += ee[i];
            ts = fmaf(qf[i], kk[i], ts);
            td = fmaf(dof[i], vv[i] + ee[i], td);
          }
          const float s = group_sum<LPH>(ts) * p.scale;
          const float da = group_sum<LPH>(td);
          const float alpha = __expf(s - lse);
          const float w = DROP ? alpha * da * edge_dropout_keep(p.drop, dseed, e + uu, head) : alpha * da;
          dsum += w;
          if (writer) {
            p.alpha[(int64_t)(e + uu) * p.H + head] = alpha;
            p.w[(int64_t)(e + uu) * p.H + head] = w;
          }
#pragma unroll
          for (int i = 0; i < VEC; ++i) {
            ak[i] = fmaf(w, kk[i], ak[i]);
            bk[i] = fmaf(alpha, kk[i], bk[i]);
          }
        }
      }
    }
    float dq[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) dq[i] = (ak[i] - dsum * bk[i]) * p.scale;
    if (active) VecIO<T, VEC>::store(static_cast<T*>(p.dq) + node * p.lddq + c0, dq);
    if (writer) p.dsum[node * p.H + head] = dsum;
  }
}

struct ConvBwdSrcParams {
  const void* q;
  const void* dout;
  const float* alpha;
  const float* w;
  const float* dsum;
  void* dk;
  void* dv;
  void* de;  // [E, ldde] CSR order
  int64_t ldq, ldd, ldg, ldde;
  int64_t n_src;
  int C, H, n_slices;
  float scale;
  EdgeDropout drop;
};

template <typename T, int VEC, int LPH, int U, bool DROP>
__global__ __launch_bounds__(256) void gt_conv_bwd_src_kernel(const ConvBwdSrcParams p,
                                                              const int32_t* __restrict__ rowptr_t,
                                                              const int32_t* __restrict__ eid_t,
                                                              const int32_t* __restrict__ dst_t) {
  using Raw = typename RawVec<T, VEC>::type;
  const int lane = threadIdx.x & 63;
  const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int xcd = blockIdx.x & 7;
  const int wave_in_xcd = (int)(blockIdx.x >> 3) * 4 + wib;
  const int waves_per_xcd = (int)(gridDim.x >> 3) * 4;
  const int slice = wave_in_xcd % p.n_slices;
  const int64_t node_first = wave_in_xcd / p.n_slices;
  const int64_t node_stride = waves_per_xcd / p.n_slices;
  const int64_t n0 = p.n_src * xcd / 8, n1 = p.n_src * (xcd + 1) / 8;
  const int lanes_total = p.C / VEC;
  const int gl = slice * 64 + lane;
  const bool active = gl < lanes_total;
  const int gls = active ? gl : 0;
  const int c0 = gls * VEC;
  const int head = gls / LPH;
  const T* qb = static_cast<const T*>(p.q) + c0;
  const T* dob = static_cast<const T*>(p.dout) + c0;
  T* deb = static_cast<T*>(p.de) + c0;
  const uint32_t dseed = DROP ? edge_dropout_seed(p.drop) : 0u;
  for (int64_t node = n0 + node_first; node < n1; node += node_stride) {
    const int t_begin = rowptr_t[node], t_end = rowptr_t[node + 1];
    float dk[VEC], dv[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) dk[i] = dv[i] = 0.f;
    for (int t = t_begin; t < t_end; t += U) {
      Raw qr[U], dor[U];
      float al[U], ww[U], dsm[U];
      int64_t eid[U];
#pragma unroll
      for (int uu = 0; uu < U; ++uu) {
        if (t + uu < t_end) {
          const int64_t e = eid_t[t + uu], i_dst = dst_t[t + uu];
          eid[uu] = e;
          qr[uu] = *reinterpret_cast<const Raw*>(qb + i_dst * p.ldq);
          dor[uu] = *reinterpret_cast<const Raw*>(dob + i_dst * p.ldd);
          al[uu] = p.alpha[e * p.H + head];
          ww[uu] = p.w[e * p.H + head];
          dsm[uu] = p.dsum[i_dst * p.H + head];
        }
      }
#pragma unroll
      for (int uu = 0; uu < U; ++uu) {
        if (t + uu < t_end) {
          const float dse = fmaf(-al[uu], dsm[uu], ww[uu]) * p.scale;  // scale ds_e
          const float alv = DROP ? al[uu] * edge_dropout_keep(p.drop, dseed, eid[uu], head) : al[uu];  // alpha_e m_e
          float qf[VEC], dof[VEC], de[VEC];
          unpack<T, VEC>(qr[uu], qf);
          unpack<T, VEC>(dor[uu], dof);
#pragma unroll
          for (int i = 0; i < VEC; ++i) {
            const float gk = dse * qf[i], gv = alv * dof[i];
            dk[i] += gk;
            dv[i] += gv;
            de[i] = gk + gv;
          }
          if (active) VecIO<T, VEC>::store(deb + eid[uu] * p.ldde, de);
        }
      }
    }
    if (active) {
      VecIO<T, VEC>::store(static_cast<T*>(p.dk) + node * p.ldg + c0, dk);
      VecIO<T, VEC>::store(static_cast<T*>(p.dv) + node * p.ldg + c0, dv);
    }
  }
}

// the forward's launch geometry: blocks of 4 waves, block b on XCD b % 8, up to 32 CUs x wgs_per_cu blocks per XCD
static inline unsigned bwd_blocks(int64_t n_nodes, int n_slices, int wgs_per_cu) {
  const int64_t units_per_xcd = ((n_nodes + 7) / 8) * n_slices;
  int64_t bpx = (units_per_xcd + 3) / 4;
  if (bpx > 32 * wgs_per_cu) bpx = 32 * wgs_per_cu;
  if (bpx < 1) bpx = 1;
  while ((bpx * 4) % n_slices != 0) ++bpx;
  return (unsigned)(8 * bpx);
}

template <typename T, int VEC, int LPH>
static bool launch_dst(const EdgeBwdParams& p, const float* attr, const int32_t* rowptr, const int32_t* col, int up,
                       hipStream_t st) {
  const dim3 grid(bwd_blocks(p.n_dst, p.n_slices, 4)), block(256);
  switch (up) {
#define ANEMOI_BWD_DST(UPV)                                                                                      \
  case UPV:                                                                                                      \
    hipLaunchKernelGGL((gt_edge_bwd_dst_kernel<T, VEC, LPH, UPV, 4>), grid, block, 0, st, p, attr, rowptr, col); \
    return true;
    ANEMOI_BWD_DST(4)
    ANEMOI_BWD_DST(8)
    ANEMOI_BWD_DST(12)
    ANEMOI_BWD_DST(16)
#undef ANEMOI_BWD_DST
    default: return false;
  }
}

template <typename T>
static bool dispatch_dst(const EdgeBwdParams& p, const float* attr, const int32_t* rowptr, const int32_t* col, int D,
                         int up, hipStream_t st) {
  constexpr int VEC = 16 / sizeof(T);
  if (D % VEC != 0) return false;
  switch (D / VEC) {
    case 1: return launch_dst<T, VEC, 1>(p, attr, rowptr, col, up, st);
    case 2: return launch_dst<T, VEC, 2>(p, attr, rowptr, col, up, st);
    case 4: return launch_dst<T, VEC, 4>(p, attr, rowptr, col, up, st);
    case 8: return launch_dst<T, VEC, 8>(p, attr, rowptr, col, up, st);
    case 16: return launch_dst<T, VEC, 16>(p, attr, rowptr, col, up, st);
    default: return false;
  }
}

template <typename T>
static bool dispatch_src(const EdgeBwdSrcParams& p, const int32_t* rowptr_t, const int32_t* eid_t, const int32_t* dst_t,
                         int D, hipStream_t st) {
  constexpr int VEC = 16 / sizeof(T);
  if (D % VEC != 0) return false;
  const dim3 grid(bwd_blocks(p.n_src, p.n_slices, 5)), block(256);
  switch (D / VEC) {
#define ANEMOI_BWD_SRC(L)                                                                                     \
  case L:                                                                                                     \
    hipLaunchKernelGGL((gt_edge_bwd_src_kernel<T, VEC, L, 4>), grid, block, 0, st, p, rowptr_t, eid_t, dst_t); \
    return true;
    ANEMOI_BWD_SRC(1)
    ANEMOI_BWD_SRC(2)
    ANEMOI_BWD_SRC(4)
    ANEMOI_BWD_SRC(8)
    ANEMOI_BWD_SRC(16)
#undef ANEMOI_BWD_SRC
    default: return false;
  }
}

}  // namespace anemoi

using namespace anemoi;

extern "C" {

int anemoi_gt_edge_attention_folded_backward_dst(int dtype, const void* q, int64_t ldq, const void* k, const void* v,
                                                 int64_t ldkv, const void* dout, int64_t ldd, const void* u, int64_t ldu,
                                                 const void* dt, int64_t lddt, const float* lse, const float* edge_attr,
                                                 int up, const int32_t* rowptr, const int32_t* col, float* alpha,
                                                 float* w, float* dsum, void* dq, int64_t lddq, void* du, int64_t lddu,
                                                 void* dxr, int64_t lddxr, int64_t n_dst, int C, int H,
                                                 anemoi_stream_t stream) {
  ANEMOI_REQUIRE(q && k && v && dout && u && dt && lse && edge_attr && rowptr && col && alpha && w && dsum && dq && du,
                 ANEMOI_ERR_INVALID, "anemoi_gt_edge_attention_folded_backward_dst: null pointer");
  ANEMOI_REQUIRE(n_dst >= 0 && C > 0 && H > 0 && C % H == 0, ANEMOI_ERR_INVALID,
                 "anemoi_gt_edge_attention_folded_backward_dst: bad shape");
  if (n_dst == 0) return ANEMOI_OK;
  const int esz = dtype == ANEMOI_BF16 ? 2 : 4, vec = 16 / esz;
  ANEMOI_REQUIRE(ldq % vec == 0 && ldkv % vec == 0 && ldd % vec == 0 && lddq % vec == 0 && C % vec == 0 &&
                     ldu % vec == 0 && lddt % vec == 0 && lddu % vec == 0 && ((int64_t)up * esz) % 8 == 0 &&
                     (uintptr_t)q % 16 == 0 && (uintptr_t)k % 16 == 0 && (uintptr_t)v % 16 == 0 &&
                     (uintptr_t)dout % 16 == 0 && (uintptr_t)dq % 16 == 0 && (uintptr_t)u % 16 == 0 &&
                     (uintptr_t)dt % 16 == 0 && (uintptr_t)du % 16 == 0 && (uintptr_t)edge_attr % 16 == 0 &&
                     (dxr == nullptr || ((uintptr_t)dxr % 16 == 0 && lddxr % vec == 0 && lddxr >= C)),
                 ANEMOI_ERR_UNSUPPORTED, "anemoi_gt_edge_attention_folded_backward_dst: operands must be 16-byte aligned");
  EdgeBwdParams p;
  p.q = q; p.k = k; p.v = v; p.dout = dout; p.u = u; p.dt = dt; p.lse = lse;
  p.alpha = alpha; p.w = w; p.dsum = dsum; p.dq = dq; p.du = du; p.dxr = dxr;
  p.ldq = ldq; p.ldkv = ldkv; p.ldd = ldd; p.ldu = ldu; p.lddt = lddt; p.lddq = lddq; p.lddu = lddu; p.lddxr = lddxr;
  p.n_dst = n_dst; p.C = C; p.H = H;
  p.n_slices = (C + 64 * vec - 1) / (64 * vec);
  p.scale = 1.0f / sqrtf((float)(C / H));
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  bool ok = false;
  if (dtype == ANEMOI_F32) ok = dispatch_dst<float>(p, edge_attr, rowptr, col, C / H, up, st);
  else if (dtype == ANEMOI_BF16) ok = dispatch_dst<bf16_t>(p, edge_attr, rowptr, col, C / H, up, st);
  ANEMOI_REQUIRE(ok, ANEMOI_ERR_UNSUPPORTED, "anemoi_gt_edge_attention_folded_backward_dst: unsupported D=%d UP=%d dtype=%d",
                 C / H, up, dtype);
  return check_launch("anemoi_gt_edge_attention_folded_backward_dst");
}

int anemoi_gt_edge_attention_folded_backward_src(int dtype, const void* q, int64_t ldq, const void* dout, int64_t ldd,
                                                 const float* alpha, const float* w, const float* dsum,
                                                 const int32_t* rowptr_t, const int32_t* eid_t, const int32_t* dst_t,
                                                 void* dk, void* dv, int64_t ldg, int64_t n_src, int C, int H,
                                                 anemoi_stream_t stream) {
  ANEMOI_REQUIRE(q && dout && alpha && w && dsum && rowptr_t && eid_t && dst_t && dk && dv, ANEMOI_ERR_INVALID,
                 "anemoi_gt_edge_attention_folded_backward_src: null pointer");
  ANEMOI_REQUIRE(n_src >= 0 && C > 0 && H > 0 && C % H == 0, ANEMOI_ERR_INVALID,
                 "anemoi_gt_edge_attention_folded_backward_src: bad shape");
  if (n_src == 0) return ANEMOI_OK;
  const int esz = dtype == ANEMOI_BF16 ? 2 : 4, vec = 16 / esz;
  ANEMOI_REQUIRE(ldq % vec == 0 && ldd % vec == 0 && ldg % vec == 0 && C % vec == 0 && (uintptr_t)q % 16 == 0 &&
                     (uintptr_t)dout % 16 == 0 && (uintptr_t)dk % 16 == 0 && (uintptr_t)dv % 16 == 0,
                 ANEMOI_ERR_UNSUPPORTED, "anemoi_gt_edge_attention_folded_backward_src: operands must be 16-byte aligned");
  EdgeBwdSrcParams p;
  p.q = q; p.dout = dout; p.alpha = alpha; p.w = w; p.dsum = dsum;
  p.dk = dk; p.dv = dv; p.ldq = ldq; p.ldd = ldd; p.ldg = ldg; p.n_src = n_src; p.C = C; p.H = H;
  p.n_slices = (C + 64 * vec - 1) / (64 * vec);
  p.scale = 1.0f / sqrtf((float)(C / H));
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  bool ok = false;
  if (dtype == ANEMOI_F32) ok = dispatch_src<float>(p, rowptr_t, eid_t, dst_t, C / H, st);
  else if (dtype == ANEMOI_BF16) ok = dispatch_src<bf16_t>(p, rowptr_t, eid_t, dst_t, C / H, st);
  ANEMOI_REQUIRE(ok, ANEMOI_ERR_UNSUPPORTED, "anemoi_gt_edge_attention_folded_backward_src: unsupported D=%d dtype=%d", C / H,
                 dtype);
  return check_launch("anemoi_gt_edge_attention_folded_backward_src");
}

int anemoi_gt_edge_attr_grad(int dtype, const float* alpha, const float* w, const float* dsum, const void* u, int64_t ldu,
                             const void* dt, int64_t lddt, const int32_t* dst_of_edge, float* dattr, int64_t n_edges,
                             int H, int up, int D, anemoi_stream_t stream) {
  ANEMOI_REQUIRE(alpha && w && dsum && u && dt && dst_of_edge && dattr && n_edges >= 0 && H > 0 && up > 0 && D > 0,
                 ANEMOI_ERR_INVALID, "anemoi_gt_edge_attr_grad: bad argument");
  if (n_edges == 0) return ANEMOI_OK;
  int64_t blocks = (n_edges * up + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const float scale = 1.0f / sqrtf((float)D);
  if (dtype == ANEMOI_F32)
    hipLaunchKernelGGL(edge_attr_grad_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, st, alpha, w, dsum,
                       static_cast<const float*>(u), ldu, static_cast<const float*>(dt), lddt, dst_of_edge, dattr, n_edges,
                       H, up, scale);
  else if (dtype == ANEMOI_BF16)
    hipLaunchKernelGGL(edge_attr_grad_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, st, alpha, w, dsum,
                       static_cast<const bf16_t*>(u), ldu, static_cast<const bf16_t*>(dt), lddt, dst_of_edge, dattr,
                       n_edges, H, up, scale);
  else
    return fail(ANEMOI_ERR_UNSUPPORTED, "anemoi_gt_edge_attr_grad: dtype %d", dtype);
  return check_launch("anemoi_gt_edge_attr_grad");
}

int anemoi_gt_conv_backward_dst(int dtype, const void* q, int64_t ldq, const void* k, const void* v, int64_t ldkv,
                                const void* edges, int64_t lde, const void* dout, int64_t ldd, const float* lse,
                                const int32_t* rowptr, const int32_t* col, float* alpha, float* w, float* dsum, void* dq,
                                int64_t lddq, int64_t n_dst, int C, int H, float dropout_p, uint32_t dropout_seed,
                                const void* dropout_seed_dev, anemoi_stream_t stream) {
  ANEMOI_REQUIRE(q && k && v && edges && dout && lse && rowptr && col && alpha && w && dsum && dq, ANEMOI_ERR_INVALID,
                 "anemoi_gt_conv_backward_dst: null pointer");
  ANEMOI_REQUIRE(n_dst >= 0 && C > 0 && H > 0 && C % H == 0, ANEMOI_ERR_INVALID, "anemoi_gt_conv_backward_dst: bad shape");
  if (n_dst == 0) return ANEMOI_OK;
  const int esz = dtype == ANEMOI_BF16 ? 2 : 4, vec = 16 / esz;
  ANEMOI_REQUIRE(ldq % vec == 0 && ldkv % vec == 0 && lde % vec == 0 && ldd % vec == 0 && lddq % vec == 0 && C % vec == 0 &&
                     (uintptr_t)q % 16 == 0 && (uintptr_t)k % 16 == 0 && (uintptr_t)v % 16 == 0 &&
                     (uintptr_t)edges % 16 == 0 && (uintptr_t)dout % 16 == 0 && (uintptr_t)dq % 16 == 0,
                 ANEMOI_ERR_UNSUPPORTED, "anemoi_gt_conv_backward_dst: operands must be 16-byte aligned");
  ConvBwdParams p;
  p.q = q; p.k = k; p.v = v; p.e = edges; p.dout = dout; p.lse = lse; p.alpha = alpha; p.w = w; p.dsum = dsum; p.dq = dq;
  p.ldq = ldq; p.ldkv = ldkv; p.lde = lde; p.ldd = ldd; p.lddq = lddq; p.n_dst = n_dst; p.C = C; p.H = H;
  p.n_slices = (C + 64 * vec - 1) / (64 * vec);
  p.scale = 1.0f / sqrtf((float)(C / H));
  p.drop = make_edge_dropout(dropout_p, dropout_seed, dropout_seed_dev);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const dim3 grid(bwd_blocks(n_dst, p.n_slices, 4)), block(256);
  const int D = C / H;
  bool ok = D % vec == 0;
#define ANEMOI_CONV_DST(TT, VV, L)                                                                                          \
  case L:                                                                                                                   \
    if (p.drop.thr15 != 0) hipLaunchKernelGGL((gt_conv_bwd_dst_kernel<TT, VV, L, 2, true>), grid, block, 0, st, p, rowptr, col); \
    else hipLaunchKernelGGL((gt_conv_bwd_dst_kernel<TT, VV, L, 2, false>), grid, block, 0, st, p, rowptr, col);              \
    break;
  if (ok && dtype == ANEMOI_F32) {
    switch (D / 4) { ANEMOI_CONV_DST(float, 4, 1) ANEMOI_CONV_DST(float, 4, 2) ANEMOI_CONV_DST(float, 4, 4)
                     ANEMOI_CONV_DST(float, 4, 8) ANEMOI_CONV_DST(float, 4, 16) default: ok = false; }
  } else if (ok && dtype == ANEMOI_BF16) {
    switch (D / 8) { ANEMOI_CONV_DST(bf16_t, 8, 1) ANEMOI_CONV_DST(bf16_t, 8, 2) ANEMOI_CONV_DST(bf16_t, 8, 4)
                     ANEMOI_CONV_DST(bf16_t, 8, 8) ANEMOI_CONV_DST(bf16_t, 8, 16) default: ok = false; }
  } else {
    ok = false;
  }
#undef ANEMOI_CONV_DST
  ANEMOI_REQUIRE(ok, ANEMOI_ERR_UNSUPPORTED, "anemoi_gt_conv_backward_dst: unsupported D=%d dtype=%d", D, dtype);
  return check_launch("anemoi_gt_conv_backward_dst");
}

int anemoi_gt_conv_backward_src(int dtype, const void* q, int64_t ldq, const void* dout, int64_t ldd, const float* alpha,
                                const float* w, const float* dsum, const int32_t* rowptr_t, const int32_t* eid_t,
                                const int32_t* dst_t, void* dk, void* dv, int64_t ldg, void* dedges, int64_t ldde,
                                int64_t n_src, int C, int H, float dropout_p, uint32_t dropout_seed,
                                const void* dropout_seed_dev, anemoi_stream_t stream) {
  ANEMOI_REQUIRE(q && dout && alpha && w && dsum && rowptr_t && eid_t && dst_t && dk && dv && dedges, ANEMOI_ERR_INVALID,
                 "anemoi_gt_conv_backward_src: null pointer");
  ANEMOI_REQUIRE(n_src >= 0 && C > 0 && H > 0 && C % H == 0, ANEMOI_ERR_INVALID, "anemoi_gt_conv_backward_src: bad shape");
  if (n_src == 0) return ANEMOI_OK;
  const int esz = dtype == ANEMOI_BF16 ? 2 : 4, vec = 16 / esz;
  ANEMOI_REQUIRE(ldq % vec == 0 && ldd % vec == 0 && ldg % vec == 0 && ldde % vec == 0 && C % vec == 0 &&
                     (uintptr_t)q % 16 == 0 && (uintptr_t)dout % 16 == 0 && (uintptr_t)dk % 16 == 0 &&
                     (uintptr_t)dv % 16 == 0 && (uintptr_t)dedges % 16 == 0,
                 ANEMOI_ERR_UNSUPPORTED, "anemoi_gt_conv_backward_src: operands must be 16-byte aligned");
  ConvBwdSrcParams p;
  p.q = q; p.dout = dout; p.alpha = alpha; p.w = w; p.dsum = dsum; p.dk = dk; p.dv = dv; p.de = dedges;
  p.ldq = ldq; p.ldd = ldd; p.ldg = ldg; p.ldde = ldde; p.n_src = n_src; p.C = C; p.H = H;
  p.n_slices = (C + 64 * vec - 1) / (64 * vec);
  p.scale = 1.0f / sqrtf((float)(C / H));
  p.drop = make_edge_dropout(dropout_p, dropout_seed, dropout_seed_dev);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const dim3 grid(bwd_blocks(n_src, p.n_slices, 5)), block(256);
  const int D = C / H;
  bool ok = D % vec == 0;
#define ANEMOI_CONV_SRC(TT, VV, L)                                                                                          \
  case L:                                                                                                                   \
    if (p.drop.thr15 != 0)                                                                                                  \
      hipLaunchKernelGGL((gt_conv_bwd_src_kernel<TT, VV, L, 2, true>), grid, block, 0, st, p, rowptr_t, eid_t, dst_t);      \
    else hipLaunchKernelGGL((gt_conv_bwd_src_kernel<TT, VV, L, 2, false>), grid, block, 0, st, p, rowptr_t, eid_t, dst_t);   \
    break;
  if (ok && dtype == ANEMOI_F32) {
    switch (D / 4) { ANEMOI_CONV_SRC(float, 4, 1) ANEMOI_CONV_SRC(float, 4, 2) ANEMOI_CONV_SRC(float, 4, 4)
                     ANEMOI_CONV_SRC(float, 4, 8) ANEMOI_CONV_SRC(float, 4, 16) default: ok = false; }
  } else if (ok && dtype == ANEMOI_BF16) {
    switch (D / 8) { ANEMOI_CONV_SRC(bf16_t, 8, 1) ANEMOI_CONV_SRC(bf16_t, 8, 2) ANEMOI_CONV_SRC(bf16_t, 8, 4)
                     ANEMOI_CONV_SRC(bf16_t, 8, 8) ANEMOI_CONV_SRC(bf16_t, 8, 16) default: ok = false; }
  } else {
    ok = false;
  }
#undef ANEMOI_CONV_SRC
  ANEMOI_REQUIRE(ok, ANEMOI_ERR_UNSUPPORTED, "anemoi_gt_conv_backward_src: unsupported D=%d dtype=%d", D, dtype);
  return check_launch("anemoi_gt_conv_backward_src");
}

}  // extern "C"
