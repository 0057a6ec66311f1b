// Backward of the folded GraphTransformer edge phase (anemoi_gt_edge_attention_folded), SURVEY.md section 8f-1.
//
// Forward, per destination i, head h, in-edge e = (j -> i) with attributes a_e (constant-1 column included):
//     s_e   = scale (q_i,h . k_j,h + u_i,h . a_e),   alpha_e = softmax_e(s_e),
//     out_i,h = sum_e alpha_e v_j,h (+ x_r),         t_i,h = sum_e alpha_e a_e.
// Backward, given dout (w.r.t. out) and dt (w.r.t. t), with Dsum_i,h = sum_e alpha_e dalpha_e accumulated in f32 over the
// in-edges (NOT rebuilt from the forward's rounded `out - x_r`: in bf16 that difference cancels against the residual):
//     dalpha_e = dout_i,h . v_j,h + dt_i,h . a_e,      ds_e = alpha_e (dalpha_e - Dsum_i,h),
//     dq_i,h = scale sum_e ds_e k_j,h,   du_i,h = scale sum_e ds_e a_e,
//     dk_j,h = scale sum_{e from j} ds_e q_i,h,   dv_j,h = sum_{e from j} alpha_e dout_i,h.
// Two kernels, no atomics:
//   * destination-major (the forward's CSR): two sweeps over the in-edges (max / sum of the scores, then alpha, dalpha and
//     the sums  A = sum alpha dalpha k,  B = sum alpha k,  Dsum  -- so that dq = scale (A - Dsum B), likewise du) and a
//     third pass over the wave's own alpha / ds entries only (ds_e = alpha_e dalpha_e - alpha_e Dsum);
//     writes alpha[E, H], ds[E, H] (f32), dq, du;
//   * source-major (the transposed CSR, edge ids into the forward's order): dk, dv as gathers of q_i / dout_i.
// One wave per (node, 64 x VEC channel slice), a lane owns VEC consecutive channels, LPH = D / VEC lanes form a head.
#include "common.hpp"

namespace anemoi {

struct EdgeBwdParams {
  const void* q;     // [n_dst, ldq]
  const void* k;     // [n_src, ldkv]
  const void* v;
  const void* dout;  // [n_dst, ldd]
  const float* u;    // [n_dst, H * UP] f32
  const float* dt;   // [n_dst, H * UP] f32
  const float* attr; // [E, UP] f32, forward CSR order
  const int32_t* rowptr;
  const int32_t* col;
  float* alpha;      // [E, H]
  float* ds;         // [E, H]
  void* dq;          // [n_dst, lddq]
  float* du;         // [n_dst, H * UP]
  int64_t ldq, ldkv, ldd, lddq;
  int64_t n_dst;
  int C, H, n_slices;
  float scale;
};

template <int WIDTH>
__device__ __forceinline__ float head_sum(float v) {  // sum over the WIDTH adjacent lanes of a head (DPP, common.hpp)
  return group_sum<WIDTH>(v);
}

template <typename T, int VEC, int LPH, int UP>
__global__ __launch_bounds__(256) void gt_edge_bwd_dst_kernel(const EdgeBwdParams p) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t n_waves = (int64_t)gridDim.x * 4;
  const int lanes_total = p.C / VEC;
  for (int64_t unit = wave; unit < p.n_dst * p.n_slices; unit += n_waves) {
    const int64_t node = unit / p.n_slices;
    const int slice = (int)(unit - node * p.n_slices);
    const int gl = slice * 64 + lane;
    const bool active = gl < lanes_total;
    const int gls = active ? gl : 0;
    const int c0 = gls * VEC, head = gls / LPH;
    const bool writer = active && (gls % LPH) == 0;
    float qf[VEC], dof[VEC], uf[UP], dtf[UP];
    VecIO<T, VEC>::load(static_cast<const T*>(p.q) + node * p.ldq + c0, qf);
    VecIO<T, VEC>::load(static_cast<const T*>(p.dout) + node * p.ldd + c0, dof);
#pragma unroll
    for (int a = 0; a < UP; ++a) {
      uf[a] = p.u[(node * p.H + head) * UP + a];
      dtf[a] = p.dt[(node * p.H + head) * UP + a];
    }
    const int e_begin = p.rowptr[node], e_end = p.rowptr[node + 1];
    // ---- sweep 1: running maximum and sum of the scores (as the forward)
    float m = -INFINITY, l = 0.f;
    for (int e = e_begin; e < e_end; ++e) {
      const int64_t j = p.col[e];
      float kf[VEC];
      VecIO<T, VEC>::load(static_cast<const T*>(p.k) + j * p.ldkv + c0, kf);
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < VEC; ++i) s = fmaf(qf[i], kf[i], s);
      s = head_sum<LPH>(s);
#pragma unroll
      for (int a = 0; a < UP; ++a) s = fmaf(uf[a], p.attr[(int64_t)e * UP + a], s);
      s *= p.scale;
      const float mn = fmaxf(m, s);
      l = l * __expf(m - mn) + __expf(s - mn);
      m = mn;
    }
    const float inv_l = 1.0f / (l + 1e-16f);
    // ---- sweep 2: alpha, dalpha and the alpha-weighted sums the destination-side gradients are made of
    float ak[VEC], bk[VEC], au[UP], bu[UP], dsum = 0.f;
#pragma unroll
    for (int i = 0; i < VEC; ++i) ak[i] = bk[i] = 0.f;
#pragma unroll
    for (int a = 0; a < UP; ++a) au[a] = bu[a] = 0.f;
    for (int e = e_begin; e < e_end; ++e) {
      const int64_t j = p.col[e];
      float kf[VEC], vf[VEC], af[UP];
      VecIO<T, VEC>::load(static_cast<const T*>(p.k) + j * p.ldkv + c0, kf);
      VecIO<T, VEC>::load(static_cast<const T*>(p.v) + j * p.ldkv + c0, vf);
#pragma unroll
      for (int a = 0; a < UP; ++a) af[a] = p.attr[(int64_t)e * UP + a];
      float s = 0.f, da = 0.f;
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        s = fmaf(qf[i], kf[i], s);
        da = fmaf(dof[i], vf[i], da);
      }
      s = head_sum<LPH>(s);
      da = head_sum<LPH>(da);
#pragma unroll
      for (int a = 0; a < UP; ++a) {
        s = fmaf(uf[a], af[a], s);
        da = fmaf(dtf[a], af[a], da);
      }
      const float alpha = __expf(s * p.scale - m) * inv_l;
      const float w = alpha * da;
      dsum += w;
      if (writer) {
        p.alpha[(int64_t)e * p.H + head] = alpha;
        p.ds[(int64_t)e * p.H + head] = w;  // finished below, once Dsum is known
      }
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        ak[i] = fmaf(w, kf[i], ak[i]);
        bk[i] = fmaf(alpha, kf[i], bk[i]);
      }
#pragma unroll
      for (int a = 0; a < UP; ++a) {
        au[a] = fmaf(w, af[a], au[a]);
        bu[a] = fmaf(alpha, af[a], bu[a]);
      }
    }
    float dq[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) dq[i] = (ak[i] - dsum * bk[i]) * p.scale;
    if (active) VecIO<T, VEC>::store(static_cast<T*>(p.dq) + node * p.lddq + c0, dq);
    if (writer) {
#pragma unroll
      for (int a = 0; a < UP; ++a) p.du[(node * p.H + head) * UP + a] = (au[a] - dsum * bu[a]) * p.scale;
      // ---- pass 3: ds_e = alpha_e (dalpha_e - Dsum) on this lane's own entries (same thread wrote them above)
      for (int e = e_begin; e < e_end; ++e) {
        const int64_t o = (int64_t)e * p.H + head;
        p.ds[o] = fmaf(-dsum, p.alpha[o], p.ds[o]);
      }
    }
  }
}

struct EdgeBwdSrcParams {
  const void* q;     // [n_dst, ldq]
  const void* dout;  // [n_dst, ldd]
  const float* alpha;
  const float* ds;
  const int32_t* rowptr_t;  // [n_src + 1] transposed CSR
  const int32_t* eid_t;     // edge id (forward CSR position) of every out-edge
  const int32_t* dst_t;     // its destination node
  void* dk;                 // [n_src, ldg]
  void* dv;
  int64_t ldq, ldd, ldg;
  int64_t n_src;
  int C, H, n_slices;
  float scale;
};

template <typename T, int VEC, int LPH>
__global__ __launch_bounds__(256) void gt_edge_bwd_src_kernel(const EdgeBwdSrcParams p) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t n_waves = (int64_t)gridDim.x * 4;
  const int lanes_total = p.C / VEC;
  for (int64_t unit = wave; unit < p.n_src * p.n_slices; unit += n_waves) {
    const int64_t node = unit / p.n_slices;
    const int slice = (int)(unit - node * p.n_slices);
    const int gl = slice * 64 + lane;
    const bool active = gl < lanes_total;
    const int gls = active ? gl : 0;
    const int c0 = gls * VEC, head = gls / LPH;
    float dk[VEC], dv[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) dk[i] = dv[i] = 0.f;
    for (int t = p.rowptr_t[node]; t < p.rowptr_t[node + 1]; ++t) {
      const int64_t e = p.eid_t[t], i_dst = p.dst_t[t];
      const float alpha = p.alpha[e * p.H + head], dse = p.ds[e * p.H + head];
      float qf[VEC], dof[VEC];
      VecIO<T, VEC>::load(static_cast<const T*>(p.q) + i_dst * p.ldq + c0, qf);
      VecIO<T, VEC>::load(static_cast<const T*>(p.dout) + i_dst * p.ldd + c0, dof);
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        dk[i] = fmaf(dse, qf[i], dk[i]);
        dv[i] = fmaf(alpha, dof[i], dv[i]);
      }
    }
#pragma unroll
    for (int i = 0; i < VEC; ++i) dk[i] *= p.scale;
    if (active) {
      VecIO<T, VEC>::store(static_cast<T*>(p.dk) + node * p.ldg + c0, dk);
      VecIO<T, VEC>::store(static_cast<T*>(p.dv) + node * p.ldg + c0, dv);
    }
  }
}

// d attr[e, a] = sum_h ( scale ds[e, h] u[dst(e), h, a] + alpha[e, h] dt[dst(e), h, a] ): the gradient of the edge
// attributes (trainable edge tensor included) from what the destination-major kernel left per (edge, head).
__global__ __launch_bounds__(256) void edge_attr_grad_kernel(const float* __restrict__ alpha, const float* __restrict__ ds,
                                                             const float* __restrict__ u, const float* __restrict__ dt,
                                                             const int32_t* __restrict__ dst_of_edge,
                                                             float* __restrict__ dattr, int64_t n_edges, int H, int UP,
                                                             float scale) {
  const int64_t total = n_edges * UP;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t e = idx / UP;
    const int a = (int)(idx - e * UP);
    const int64_t i = dst_of_edge[e];
    float acc = 0.f;
    for (int h = 0; h < H; ++h) {
      const int64_t nh = (i * H + h) * UP + a;
      acc = fmaf(scale * ds[e * H + h], u[nh], acc);
      acc = fmaf(alpha[e * H + h], dt[nh], acc);
    }
    dattr[idx] = acc;
  }
}

static inline unsigned bwd_blocks(int64_t units) {
  int64_t b = (units + 3) / 4;
  if (b > 256 * 8) b = 256 * 8;
  return (unsigned)(b < 1 ? 1 : b);
}

template <typename T, int VEC, int LPH>
static bool launch_dst(const EdgeBwdParams& p, int up, hipStream_t st) {
  const dim3 grid(bwd_blocks(p.n_dst * p.n_slices)), block(256);
  switch (up) {
    case 4: hipLaunchKernelGGL((gt_edge_bwd_dst_kernel<T, VEC, LPH, 4>), grid, block, 0, st, p); return true;
    case 8: hipLaunchKernelGGL((gt_edge_bwd_dst_kernel<T, VEC, LPH, 8>), grid, block, 0, st, p); return true;
    case 12: hipLaunchKernelGGL((gt_edge_bwd_dst_kernel<T, VEC, LPH, 12>), grid, block, 0, st, p); return true;
    case 16: hipLaunchKernelGGL((gt_edge_bwd_dst_kernel<T, VEC, LPH, 16>), grid, block, 0, st, p); return true;
    default: return false;
  }
}

template <typename T>
static bool dispatch_dst(const EdgeBwdParams& p, int D, int up, hipStream_t st) {
  constexpr int VEC = 16 / sizeof(T);
  if (D % VEC != 0) return false;
  switch (D / VEC) {
    case 1: return launch_dst<T, VEC, 1>(p, up, st);
    case 2: return launch_dst<T, VEC, 2>(p, up, st);
    case 4: return launch_dst<T, VEC, 4>(p, up, st);
    case 8: return launch_dst<T, VEC, 8>(p, up, st);
    case 16: return launch_dst<T, VEC, 16>(p, up, st);
    default: return false;
  }
}

template <typename T>
static bool dispatch_src(const EdgeBwdSrcParams& p, int D, hipStream_t st) {
  constexpr int VEC = 16 / sizeof(T);
  if (D % VEC != 0) return false;
  const dim3 grid(bwd_blocks(p.n_src * p.n_slices)), block(256);
  switch (D / VEC) {
    case 1: hipLaunchKernelGGL((gt_edge_bwd_src_kernel<T, VEC, 1>), grid, block, 0, st, p); return true;
    case 2: hipLaunchKernelGGL((gt_edge_bwd_src_kernel<T, VEC, 2>), grid, block, 0, st, p); return true;
    case 4: hipLaunchKernelGGL((gt_edge_bwd_src_kernel<T, VEC, 4>), grid, block, 0, st, p); return true;
    case 8: hipLaunchKernelGGL((gt_edge_bwd_src_kernel<T, VEC, 8>), grid, block, 0, st, p); return true;
    case 16: hipLaunchKernelGGL((gt_edge_bwd_src_kernel<T, VEC, 16>), grid, block, 0, st, p); return true;
    default: return false;
  }
}

}  // namespace anemoi

using namespace anemoi;

extern "C" {

int anemoi_gt_edge_attention_folded_backward_dst(int dtype, const void* q, int64_t ldq, const void* k, const void* v,
                                                 int64_t ldkv, const void* dout, int64_t ldd, const float* u,
                                                 const float* dt, const float* edge_attr, int up,
                                                 const int32_t* rowptr, const int32_t* col, float* alpha, float* ds,
                                                 void* dq, int64_t lddq, float* du, int64_t n_dst, int C, int H,
                                                 anemoi_stream_t stream) {
  ANEMOI_REQUIRE(q && k && v && dout && u && dt && edge_attr && rowptr && col && alpha && ds && dq && du,
                 ANEMOI_ERR_INVALID, "anemoi_gt_edge_attention_folded_backward_dst: null pointer");
  ANEMOI_REQUIRE(n_dst >= 0 && C > 0 && H > 0 && C % H == 0, ANEMOI_ERR_INVALID,
                 "anemoi_gt_edge_attention_folded_backward_dst: bad shape");
  if (n_dst == 0) return ANEMOI_OK;
  const int esz = dtype == ANEMOI_BF16 ? 2 : 4, vec = 16 / esz;
  ANEMOI_REQUIRE(ldq % vec == 0 && ldkv % vec == 0 && ldd % vec == 0 && lddq % vec == 0 && C % vec == 0 &&
                     (uintptr_t)q % 16 == 0 && (uintptr_t)k % 16 == 0 && (uintptr_t)v % 16 == 0 &&
                     (uintptr_t)dout % 16 == 0 && (uintptr_t)dq % 16 == 0,
                 ANEMOI_ERR_UNSUPPORTED, "anemoi_gt_edge_attention_folded_backward_dst: operands must be 16-byte aligned");
  EdgeBwdParams p;
  p.q = q; p.k = k; p.v = v; p.dout = dout; p.u = u; p.dt = dt; p.attr = edge_attr;
  p.rowptr = rowptr; p.col = col; p.alpha = alpha; p.ds = ds; p.dq = dq; p.du = du;
  p.ldq = ldq; p.ldkv = ldkv; p.ldd = ldd; p.lddq = lddq; p.n_dst = n_dst; p.C = C; p.H = H;
  p.n_slices = (C + 64 * vec - 1) / (64 * vec);
  p.scale = 1.0f / sqrtf((float)(C / H));
  bool ok = false;
  if (dtype == ANEMOI_F32) ok = dispatch_dst<float>(p, C / H, up, reinterpret_cast<hipStream_t>(stream));
  else if (dtype == ANEMOI_BF16) ok = dispatch_dst<bf16_t>(p, C / H, up, reinterpret_cast<hipStream_t>(stream));
  ANEMOI_REQUIRE(ok, ANEMOI_ERR_UNSUPPORTED, "anemoi_gt_edge_attention_folded_backward_dst: unsupported D=%d UP=%d dtype=%d",
                 C / H, up, dtype);
  return check_launch("anemoi_gt_edge_attention_folded_backward_dst");
}

int anemoi_gt_edge_attention_folded_backward_src(int dtype, const void* q, int64_t ldq, const void* dout, int64_t ldd,
                                                 const float* alpha, const float* ds, const int32_t* rowptr_t,
                                                 const int32_t* eid_t, const int32_t* dst_t, void* dk, void* dv,
                                                 int64_t ldg, int64_t n_src, int C, int H, anemoi_stream_t stream) {
  ANEMOI_REQUIRE(q && dout && alpha && ds && rowptr_t && eid_t && dst_t && dk && dv, ANEMOI_ERR_INVALID,
                 "anemoi_gt_edge_attention_folded_backward_src: null pointer");
  ANEMOI_REQUIRE(n_src >= 0 && C > 0 && H > 0 && C % H == 0, ANEMOI_ERR_INVALID,
                 "anemoi_gt_edge_attention_folded_backward_src: bad shape");
  if (n_src == 0) return ANEMOI_OK;
  const int esz = dtype == ANEMOI_BF16 ? 2 : 4, vec = 16 / esz;
  ANEMOI_REQUIRE(ldq % vec == 0 && ldd % vec == 0 && ldg % vec == 0 && C % vec == 0 && (uintptr_t)q % 16 == 0 &&
                     (uintptr_t)dout % 16 == 0 && (uintptr_t)dk % 16 == 0 && (uintptr_t)dv % 16 == 0,
                 ANEMOI_ERR_UNSUPPORTED, "anemoi_gt_edge_attention_folded_backward_src: operands must be 16-byte aligned");
  EdgeBwdSrcParams p;
  p.q = q; p.dout = dout; p.alpha = alpha; p.ds = ds; p.rowptr_t = rowptr_t; p.eid_t = eid_t; p.dst_t = dst_t;
  p.dk = dk; p.dv = dv; p.ldq = ldq; p.ldd = ldd; p.ldg = ldg; p.n_src = n_src; p.C = C; p.H = H;
  p.n_slices = (C + 64 * vec - 1) / (64 * vec);
  p.scale = 1.0f / sqrtf((float)(C / H));
  bool ok = false;
  if (dtype == ANEMOI_F32) ok = dispatch_src<float>(p, C / H, reinterpret_cast<hipStream_t>(stream));
  else if (dtype == ANEMOI_BF16) ok = dispatch_src<bf16_t>(p, C / H, reinterpret_cast<hipStream_t>(stream));
  ANEMOI_REQUIRE(ok, ANEMOI_ERR_UNSUPPORTED, "anemoi_gt_edge_attention_folded_backward_src: unsupported D=%d dtype=%d", C / H,
                 dtype);
  return check_launch("anemoi_gt_edge_attention_folded_backward_src");
}

int anemoi_gt_edge_attr_grad(const float* alpha, const float* ds, const float* u, const float* dt,
                             const int32_t* dst_of_edge, float* dattr, int64_t n_edges, int H, int up, int D,
                             anemoi_stream_t stream) {
  ANEMOI_REQUIRE(alpha && ds && u && dt && dst_of_edge && dattr && n_edges >= 0 && H > 0 && up > 0 && D > 0,
                 ANEMOI_ERR_INVALID, "anemoi_gt_edge_attr_grad: bad argument");
  if (n_edges == 0) return ANEMOI_OK;
  int64_t blocks = (n_edges * up + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(edge_attr_grad_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     alpha, ds, u, dt, dst_of_edge, dattr, n_edges, H, up, 1.0f / sqrtf((float)D));
  return check_launch("anemoi_gt_edge_attr_grad");
}

}  // extern "C"
