// Fused GraphTransformer edge phase (K1 + K2 of SURVEY.md section 2a) for gfx950.
//
// One pass over a destination-sorted CSR graph.  A wave64 owns one (destination node, channel slice)
// unit at a time: lane l holds VEC consecutive channels, LPH = D / VEC adjacent lanes form one head.
// For every in-edge the wave gathers the k_j / v_j row slices with one 16-byte load per lane
// (a contiguous 64*VEC-element segment per wave), recomputes lin_edge from the raw edge attributes
// (wave-uniform -> scalar loads; W_e slice resident in VGPRs for the whole kernel), reduces q.(k+e)
// inside each head with cross-lane adds, and keeps an online softmax (running max / sum) so the
// destination row is written exactly once.  No atomics, no [E, C] temporaries: compulsory traffic is
// q, k, v read once + out written once + 4*(EDP+1) bytes per edge.
//
// Work mapping is XCD-aware: block b runs on XCD b % 8 (observed dispatch order, used for speed only),
// so XCD x walks the contiguous destination range [x*N/8, (x+1)*N/8) and the k/v rows shared by
// neighbouring destinations are re-used from that XCD's L2.
#include <cstdlib>

#include "common.hpp"

namespace anemoi {

struct EdgeAttnParams {
  const void* q;
  const void* k;
  const void* v;
  const void* xr;  // optional
  void* out;
  int64_t ldq, ldkv, ldr, ldo;
  const float* attr;  // [E, ea_ld] CSR order
  const float* w;     // [C, edge_dim]
  const float* b;     // [C]
  const int32_t* rowptr;
  const int32_t* col;
  int64_t n_dst;
  int C, D, ea_ld, edge_dim, n_slices;
  float scale;
};

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

// ---------------------------------------------------------------------------------------------
// Fast path: compile-time VEC (channels per lane), LPH (lanes per head) and EDP (edge_dim padded to 4).
//
// lin_edge is linear, so it is never evaluated per edge.  With a_ij the raw attributes (+ a constant 1
// for the bias) and W_h the rows of W_e that belong to head h:
//     q_i,h . (k_j,h + e_ij,h)      = q_i,h . k_j,h + (W_h^T q_i,h) . a_ij         -> u_i,h = W_h^T q_i,h  per node
//     sum_j p_ij (v_j,h + e_ij,h)   = sum_j p_ij v_j,h + W_h (sum_j p_ij a_ij)     -> t_i,h = sum_j p_ij a_ij
// Per edge that leaves VEC FMAs for q.k, EDP FMAs for u.a, VEC FMAs for p*v and EDP FMAs for t; W_e is
// touched twice per destination node (u in the prologue, W t in the epilogue) and lives in LDS in a
// lane-major layout so that each access is one conflict-free ds_read_b128 per 4 channels.
// ---------------------------------------------------------------------------------------------
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

template <typename T, int VEC, int LPH, int EDP>
__global__ __launch_bounds__(256) void gt_edge_attention_kernel(const EdgeAttnParams p,
                                                                const float* __restrict__ attr_,
                                                                const int32_t* __restrict__ rowptr_,
                                                                const int32_t* __restrict__ col_) {
  constexpr int U = 4;          // edges in flight per wave: 2*U independent 16-byte gathers per lane
  constexpr int NQ = (VEC + 3) / 4;  // 16-byte LDS reads per lane per attribute row
  constexpr int QW = VEC < 4 ? VEC : 4;
  using Raw = typename RawVec<T, VEC>::type;
  extern __shared__ __attribute__((aligned(16))) float w_lds[];  // [(EDP + 1) * NQ][C / VEC][QW]; row EDP = bias

  const int lanes_total = p.C / VEC;
  for (int idx = threadIdx.x; idx < (EDP + 1) * p.C; idx += blockDim.x) {
    const int i = idx % QW;
    const int gl = (idx / QW) % lanes_total;
    const int ah = idx / (QW * lanes_total);
    const int a = ah / NQ, h = ah % NQ;
    const int c = gl * VEC + h * QW + i;
    float val = 0.f;
    if (a < p.edge_dim) val = p.w[(int64_t)c * p.edge_dim + a];
    else if (a == EDP) val = p.b[c];
    w_lds[idx] = val;
  }
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int wpb = (int)(blockDim.x >> 6);
  const int xcd = blockIdx.x & 7;
  const int wave_in_xcd = (int)(blockIdx.x >> 3) * wpb + wib;
  const int waves_per_xcd = (int)(gridDim.x >> 3) * wpb;
  const int slice = wave_in_xcd % p.n_slices;
  const int64_t node_first = wave_in_xcd / p.n_slices;
  const int64_t node_stride = waves_per_xcd / p.n_slices;
  const int64_t n0 = p.n_dst * xcd / 8, n1 = p.n_dst * (xcd + 1) / 8;

  const int gl = slice * 64 + lane;  // global lane index = channel group
  const bool active = gl < lanes_total;
  const int gls = active ? gl : 0;  // inactive lanes shadow group 0 (never stored)
  const int c0 = gls * VEC;

  const T* qb = static_cast<const T*>(p.q) + c0;
  const T* kb = static_cast<const T*>(p.k) + c0;
  const T* vb = static_cast<const T*>(p.v) + c0;

  for (int64_t node = n0 + node_first; node < n1; node += node_stride) {
    const int e_begin = rowptr_[node], e_end = rowptr_[node + 1];
    QK<T, VEC> qk;
    float u[EDP + 1];
    // The LDS image of W_e is loop invariant; without these opaque offsets the compiler hoists all
    // (EDP + 1) * VEC weight reads out of the node loop (and keeps them live across the edge loop),
    // i.e. rebuilds the register-resident W_e that costs the kernel its occupancy.
    int wofs = gls * QW;
    asm volatile("" : "+v"(wofs));
    {
      float qf[VEC];
      VecIO<T, VEC>::load(qb + node * p.ldq, qf);
      qk.set(qf);
      // u[a] = sum over the head's channels of W_e[c][a] * q[c]   (a = EDP: bias row)
#pragma unroll
      for (int a = 0; a <= EDP; ++a) {
        float part = 0.f;
#pragma unroll
        for (int h = 0; h < NQ; ++h) {
          float wv[QW];
          VecIO<float, QW>::load(&w_lds[(a * NQ + h) * lanes_total * QW + wofs], wv);
#pragma unroll
          for (int i = 0; i < QW; ++i) part = fmaf(wv[i], qf[h * QW + i], part);
        }
        u[a] = group_sum<LPH>(part);
      }
    }
    float m = -INFINITY, l = 0.f;
    float acc[VEC], tacc[EDP];
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
#pragma unroll
    for (int a = 0; a < EDP; ++a) tacc[a] = 0.f;

    for (int e = e_begin; e < e_end; e += U) {
      Raw kr[U], vr[U];
      float s[U];
#pragma unroll
      for (int uu = 0; uu < U; ++uu) {
        if (e + uu < e_end) {
          const int64_t j = col_[e + uu];
          kr[uu] = *reinterpret_cast<const Raw*>(kb + j * p.ldkv);
          vr[uu] = *reinterpret_cast<const Raw*>(vb + j * p.ldkv);
        }
      }
      float mb = m;
#pragma unroll
      for (int uu = 0; uu < U; ++uu) {
        s[uu] = -INFINITY;
        if (e + uu < e_end) {
          const float* at = attr_ + (int64_t)(e + uu) * p.ea_ld;
          float t = u[EDP];
#pragma unroll
          for (int a = 0; a < EDP; ++a) t = fmaf(u[a], at[a], t);
          s[uu] = (group_sum<LPH>(qk.dot(kr[uu])) + t) * p.scale;
          mb = fmaxf(mb, s[uu]);
        }
      }
      const float corr = __expf(m - mb);
      l *= corr;
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[i] *= corr;
#pragma unroll
      for (int a = 0; a < EDP; ++a) tacc[a] *= corr;
#pragma unroll
      for (int uu = 0; uu < U; ++uu) {
        if (e + uu < e_end) {
          const float* at = attr_ + (int64_t)(e + uu) * p.ea_ld;
          const float pe = __expf(s[uu] - mb);
          l += pe;
          float vv[VEC];
          unpack<T, VEC>(vr[uu], vv);
#pragma unroll
          for (int i = 0; i < VEC; ++i) acc[i] = fmaf(pe, vv[i], acc[i]);
#pragma unroll
          for (int a = 0; a < EDP; ++a) tacc[a] = fmaf(pe, at[a], tacc[a]);
        }
      }
      m = mb;
    }

    // epilogue: out = (acc + W_h t + b * l) / (l + 1e-16) (+ x_r)
    const float inv = 1.0f / (l + 1e-16f);
    float o[VEC];
    int wofs2 = gls * QW;
    asm volatile("" : "+v"(wofs2));
#pragma unroll
    for (int h = 0; h < NQ; ++h) {
      float bv[QW];
      VecIO<float, QW>::load(&w_lds[(EDP * NQ + h) * lanes_total * QW + wofs2], bv);
#pragma unroll
      for (int i = 0; i < QW; ++i) o[h * QW + i] = fmaf(bv[i], l, acc[h * QW + i]);
    }
#pragma unroll
    for (int a = 0; a < EDP; ++a) {
#pragma unroll
      for (int h = 0; h < NQ; ++h) {
        float wv[QW];
        VecIO<float, QW>::load(&w_lds[(a * NQ + h) * lanes_total * QW + wofs2], wv);
#pragma unroll
        for (int i = 0; i < QW; ++i) o[h * QW + i] = fmaf(wv[i], tacc[a], o[h * QW + i]);
      }
    }
#pragma unroll
    for (int i = 0; i < VEC; ++i) o[i] *= inv;
    if (p.xr != nullptr) {
      float r[VEC];
      VecIO<T, VEC>::load(static_cast<const T*>(p.xr) + c0 + node * p.ldr, r);
#pragma unroll
      for (int i = 0; i < VEC; ++i) o[i] += r[i];
    }
    if (active) VecIO<T, VEC>::store(static_cast<T*>(p.out) + c0 + node * p.ldo, o);
  }
}

// ---------------------------------------------------------------------------------------------
// Folded path: lin_edge never appears in the kernel.  The two places where W_e acts are linear maps of
// node-level quantities, so they are folded into the GEMMs on either side of the edge phase:
//     u_i,h = W_h'^T q_i,h      comes out of the q/k/v projection GEMM as H*UP extra output columns
//                               (weight rows  W_u[(h,a), :] = sum_{c in h} W_e'[c,a] * W_q[c, :]),
//     W_h' t_i,h                goes into the output projection GEMM as H*UP extra input columns
//                               (weight cols  W_t[:, (h,a)] = sum_{c in h} W_p[:, c] * W_e'[c,a]),
// with W_e' = [W_e | b_e] and the edge attributes carrying a constant 1 in column edge_dim (so the bias and
// the softmax normaliser ride along for free).  The kernel is then a pure gather / dot / online softmax /
// weighted sum: no LDS, ~half the registers, 2x the resident waves.
//   inputs : q, u per destination; k, v per source; attr [E, UP] (CSR order, attr[edge_dim] = 1)
//   outputs: out[:, 0:C]        = sum_j alpha_ij v_j (+ x_r)
//            out[:, C:C+H*UP]   = t~_i,h = sum_j alpha_ij a_ij        (alpha includes the 1e-16 normaliser)
// ---------------------------------------------------------------------------------------------
struct EdgeFoldParams {
  const void* q;
  const void* k;
  const void* v;
  const void* xr;
  const void* u;
  void* out;
  int64_t ldq, ldkv, ldr, ldu, ldo;
  const float* attr;
  const int32_t* rowptr;
  const int32_t* col;
  int64_t n_dst;
  int C, D, n_slices;
  float scale;
  int stream_hint;  // 1: q / x_r / out are nontemporal so that they do not evict gathered k|v rows from the XCD's L2
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

// PIPE: how much of the NEXT destination is requested while the current one is processed.  The kernel is bound by its
// chain of dependent memory round trips per destination (row pointers -> columns -> k|v gathers, q before the first
// score, x_r before the store), not by bytes or VALU:
//   0  nothing (the loop of the first version: every operand is requested when it is needed)
//   1  x_r is requested together with q at the start of the destination
//   2  + the next destination's row pointers, q, u and x_r are requested before the current one's edge loop
//   3  + the columns of the following edge batch (this destination's next U edges, or the next destination's first U)
//      are requested while the current batch is processed
template <typename T, int VEC, int LPH, int UP, int U = 4, int PIPE = 3, int MINW = 1>
__global__ __launch_bounds__(256, MINW) void gt_edge_attention_folded_kernel(const EdgeFoldParams p,
                                                                   const float* __restrict__ attr_,
                                                                   const int32_t* __restrict__ rowptr_,
                                                                   const int32_t* __restrict__ col_) {
  using Raw = typename RawVec<T, VEC>::type;
  constexpr int APL = attrs_per_lane(UP, LPH);
  static_assert(UP % APL == 0 && APL * LPH >= UP, "a lane owns APL whole attributes or none");
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
  const int a0 = (gls % LPH) * APL;  // first attribute of this lane
  const bool a_own = a0 < UP;        // the last lanes of a head own none when LPH * APL > UP (u = 0, nothing stored)
  const int a_ld = a_own ? a0 : 0;
  const float amask = a_own ? 1.f : 0.f;
  const bool nt = p.stream_hint != 0;
  const bool has_xr = p.xr != nullptr;

  // wave-uniform row bases (bytes) + 32-bit lane offsets
  constexpr int64_t ES = (int64_t)sizeof(T);
  const char* qb = static_cast<const char*>(p.q);
  const char* kb = static_cast<const char*>(p.k);
  const char* vb = static_cast<const char*>(p.v);
  const char* ub = static_cast<const char*>(p.u);
  const char* rb = has_xr ? static_cast<const char*>(p.xr) : qb;
  const int64_t ldq_b = p.ldq * ES, ldkv_b = p.ldkv * ES, ldu_b = p.ldu * ES, ldo_b = p.ldo * ES;
  const int64_t ldr_b = (has_xr ? p.ldr : p.ldq) * ES;
  const char* ab = reinterpret_cast<const char*>(attr_);
  const uint32_t off_c = (uint32_t)(c0 * (int)sizeof(T));
  const uint32_t off_u = (uint32_t)((head * UP + a_ld) * (int)sizeof(T));
  const uint32_t off_a = (uint32_t)(a_ld * 4);

  int64_t node = n0 + node_first;
  if (node >= n1) return;
  // streaming operands of the current destination (requested one destination ahead when PIPE >= 2)
  RawWords<T, VEC> q_raw, xr_raw;
  RawWords<T, APL> u_raw;
  int e_begin = rowptr_[node], e_end = rowptr_[node + 1];
  int cj[U];  // source rows of the edge batch about to be processed (PIPE >= 3)
  if constexpr (PIPE >= 2) {
    q_raw.load(qb + node * ldq_b, off_c, nt);
    u_raw.load(ub + node * ldu_b, off_u, false);
    xr_raw.load(rb + node * ldr_b, off_c, nt);
  }
  if constexpr (PIPE >= 3) {
    const int last = e_end > e_begin ? e_end - 1 : (e_begin > 0 ? e_begin - 1 : 0);
#pragma unroll
    for (int uu = 0; uu < U; ++uu) cj[uu] = col_[e_begin + uu < last ? e_begin + uu : last];
  }

  for (;;) {
    const int64_t next = node + node_stride;
    const bool has_next = next < n1;
    const int64_t nn = has_next ? next : node;
    RawWords<T, VEC> q_n, xr_n;
    RawWords<T, APL> u_n;
    int eb_n = 0, ee_n = 0;
    if constexpr (PIPE >= 2) {  // the next destination's streams go out first: they are the long (HBM) round trips
      eb_n = rowptr_[nn];
      ee_n = rowptr_[nn + 1];
      q_n.load(qb + nn * ldq_b, off_c, nt);
      u_n.load(ub + nn * ldu_b, off_u, false);
      xr_n.load(rb + nn * ldr_b, off_c, nt);
    } else {
      q_raw.load(qb + node * ldq_b, off_c, nt);
      u_raw.load(ub + node * ldu_b, off_u, false);
      if constexpr (PIPE >= 1) xr_raw.load(rb + node * ldr_b, off_c, nt);
    }
    QK<T, VEC> qk;
    float u[APL];
    {
      float qf[VEC];
      q_raw.get(qf);
      qk.set(qf);
      u_raw.get(u);
#pragma unroll
      for (int i = 0; i < APL; ++i) u[i] *= amask;
    }
    float m = -INFINITY, l = 0.f;
    // the value accumulators live as f32 pairs: rescale and accumulate are v_pk_mul_f32 / v_pk_fma_f32 (two channels per
    // issue slot)
    typedef __attribute__((ext_vector_type(2))) float f32x2_t;
    constexpr int VP = (VEC + 1) / 2;
    f32x2_t acc[VP];
    float tacc[APL];
#pragma unroll
    for (int i = 0; i < VP; ++i) acc[i] = f32x2_t{0.f, 0.f};
#pragma unroll
    for (int a = 0; a < APL; ++a) tacc[a] = 0.f;

    if constexpr (PIPE >= 3) {
      if (e_begin == e_end) {  // destination without edges: the batch loop below does not run and cannot refill cj
        const int last = ee_n > eb_n ? ee_n - 1 : (eb_n > 0 ? eb_n - 1 : 0);
#pragma unroll
        for (int uu = 0; uu < U; ++uu) cj[uu] = col_[eb_n + uu < last ? eb_n + uu : last];
      }
    }
    for (int e = e_begin; e < e_end; e += U) {
      Raw kr[U], vr[U];
      float at[U][APL];
      float s[U];
#pragma unroll
      for (int uu = 0; uu < U; ++uu) {
        if (e + uu < e_end) {
          const int64_t j = PIPE >= 3 ? (int64_t)cj[uu] : (int64_t)col_[e + uu];
          kr[uu] = *reinterpret_cast<const Raw*>(kb + j * ldkv_b + off_c);
          vr[uu] = *reinterpret_cast<const Raw*>(vb + j * ldkv_b + off_c);
          VecIO<float, APL>::load(reinterpret_cast<const float*>(ab + (int64_t)(e + uu) * (UP * 4) + off_a), at[uu]);
        }
      }
      if constexpr (PIPE >= 3) {  // columns of the batch after this one (clamped: never past the rows' last edge)
        const bool more = e + U < e_end;
        const int pb = more ? e + U : eb_n;
        const int pe = more ? e_end : ee_n;
        const int last = pe > pb ? pe - 1 : (pb > 0 ? pb - 1 : 0);
#pragma unroll
        for (int uu = 0; uu < U; ++uu) cj[uu] = col_[pb + uu < last ? pb + uu : last];
      }
      float mb = m;
#pragma unroll
      for (int uu = 0; uu < U; ++uu) {
        s[uu] = -INFINITY;
        if (e + uu < e_end) {
          float t = qk.dot(kr[uu]);
#pragma unroll
          for (int a = 0; a < APL; ++a) t = fmaf(u[a], at[uu][a], t);
          s[uu] = group_sum<LPH>(t) * p.scale;
          mb = fmaxf(mb, s[uu]);
        }
      }
      const float corr = __expf(m - mb);
      l *= corr;
#pragma unroll
      for (int i = 0; i < VP; ++i) acc[i] *= corr;
#pragma unroll
      for (int a = 0; a < APL; ++a) tacc[a] *= corr;
#pragma unroll
      for (int uu = 0; uu < U; ++uu) {
        if (e + uu < e_end) {
          const float pe = __expf(s[uu] - mb);
          l += pe;
          float vv[VEC];
          unpack<T, VEC>(vr[uu], vv);
#pragma unroll
          for (int i = 0; i < VP; ++i)
            acc[i] = __builtin_elementwise_fma(f32x2_t{pe, pe}, f32x2_t{vv[2 * i], 2 * i + 1 < VEC ? vv[2 * i + 1] : 0.f},
                                               acc[i]);
#pragma unroll
          for (int a = 0; a < APL; ++a) tacc[a] = fmaf(pe, at[uu][a], tacc[a]);
        }
      }
      m = mb;
    }

    const float inv = 1.0f / (l + 1e-16f);
    float o[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) o[i] = acc[i >> 1][i & 1] * inv;
    if (has_xr) {
      float r[VEC];
      if constexpr (PIPE == 0) xr_raw.load(rb + node * ldr_b, off_c, nt);
      xr_raw.get(r);
#pragma unroll
      for (int i = 0; i < VEC; ++i) o[i] += r[i];
    }
    char* on = static_cast<char*>(p.out) + node * ldo_b;
    if (active) {
      if (nt) store_stream<T, VEC>(reinterpret_cast<T*>(on + off_c), o);
      else VecIO<T, VEC>::store(reinterpret_cast<T*>(on + off_c), o);
    }
    if (active && a_own) {  // this lane's APL values of t~_i,h
      float t4[APL];
#pragma unroll
      for (int a = 0; a < APL; ++a) t4[a] = tacc[a] * inv;
      VecIO<T, APL>::store(reinterpret_cast<T*>(on + (uint32_t)((p.C + head * UP + a0) * (int)sizeof(T))), t4);
    }
    if (!has_next) break;
    node = next;
    if constexpr (PIPE >= 2) {
      q_raw = q_n;
      u_raw = u_n;
      xr_raw = xr_n;
      e_begin = eb_n;
      e_end = ee_n;
    } else {
      e_begin = rowptr_[node];
      e_end = rowptr_[node + 1];
    }
  }
}

template <typename T, int VEC, int LPH, int UP>
static void launch_folded(const EdgeFoldParams& p, hipStream_t st) {
  constexpr int WPB = 4;
  static const int edges_in_flight = [] {  // tuning knobs (A/B): ANEMOI_AMD_EDGE_U in {2,4,8}, ANEMOI_AMD_EDGE_WGS per CU
    const char* e = getenv("ANEMOI_AMD_EDGE_U");
    return e ? atoi(e) : 4;
  }();
  static const int wgs_per_cu = [] {
    const char* e = getenv("ANEMOI_AMD_EDGE_WGS");
    return e ? atoi(e) : 5;
  }();
  const int64_t units_per_xcd = ((p.n_dst + 7) / 8) * p.n_slices;
  int64_t bpx = (units_per_xcd + WPB - 1) / WPB;
  if (bpx > 32 * wgs_per_cu) bpx = 32 * wgs_per_cu;  // resident workgroups per CU x 32 CUs per XCD
  if (bpx < 1) bpx = 1;
  while ((bpx * WPB) % p.n_slices != 0) ++bpx;
  // (a register-double-buffered software pipeline across destinations was measured and removed: 0.30 / 1.98 / 0.79 ms
  // against 0.19 / 1.21 / 0.60 ms of this loop on the mesh / decoder / encoder graphs of config 3 -- the second
  // register set costs a wave per SIMD, which hurts more than the overlap helps)
  static const int pipe = [] {  // A/B: ANEMOI_AMD_EDGE_PIPE in {0, 1, 2, 3} (see the kernel), default 3
    const char* e = getenv("ANEMOI_AMD_EDGE_PIPE");
    return e ? atoi(e) : 3;
  }();
  const dim3 grid((unsigned)(8 * bpx)), block(64 * WPB);
#define ANEMOI_EDGE_LAUNCH(UU, PP, MW) \
  hipLaunchKernelGGL((gt_edge_attention_folded_kernel<T, VEC, LPH, UP, UU, PP, MW>), grid, block, 0, st, p, p.attr, p.rowptr, p.col)
  if (edges_in_flight == 8) ANEMOI_EDGE_LAUNCH(8, 3, 1);
  else if (edges_in_flight == 2) ANEMOI_EDGE_LAUNCH(2, 3, 1);
  else if (pipe == 0) ANEMOI_EDGE_LAUNCH(4, 0, 1);
  else if (pipe == 1) ANEMOI_EDGE_LAUNCH(4, 1, 1);
  else if (pipe == 2) ANEMOI_EDGE_LAUNCH(4, 2, 1);
  else if (pipe == 12) ANEMOI_EDGE_LAUNCH(4, 2, 5);  // (lab: the same with the register budget of 5 waves per SIMD)
  else if (pipe == 13) ANEMOI_EDGE_LAUNCH(4, 3, 5);
  else ANEMOI_EDGE_LAUNCH(4, 3, 1);
#undef ANEMOI_EDGE_LAUNCH
}

template <typename T, int VEC, int LPH>
static bool dispatch_folded_up(const EdgeFoldParams& p, int up, hipStream_t st) {
  switch (up) {
    case 4: launch_folded<T, VEC, LPH, 4>(p, st); return true;
    case 8: launch_folded<T, VEC, LPH, 8>(p, st); return true;
    case 12: launch_folded<T, VEC, LPH, 12>(p, st); return true;
    case 16: launch_folded<T, VEC, LPH, 16>(p, st); return true;
    default: return false;
  }
}

template <typename T>
static bool dispatch_folded(const EdgeFoldParams& p, int up, hipStream_t st) {
  constexpr int VEC = 16 / sizeof(T);
  if (p.D % VEC != 0 || p.C % VEC != 0) return false;
  switch (p.D / VEC) {
    case 1: return dispatch_folded_up<T, VEC, 1>(p, up, st);
    case 2: return dispatch_folded_up<T, VEC, 2>(p, up, st);
    case 4: return dispatch_folded_up<T, VEC, 4>(p, up, st);
    case 8: return dispatch_folded_up<T, VEC, 8>(p, up, st);
    case 16: return dispatch_folded_up<T, VEC, 16>(p, up, st);
    default: return false;
  }
}

// ---------------------------------------------------------------------------------------------
// Generic path: one channel per lane, run-time lanes-per-head and edge_dim, W_e read through the
// cache per edge.  Covers every shape the fast path does not (odd head sizes, edge_dim > 16, ...).
// Requires D to be a power of two <= 64 or, failing that, uses a shuffle-free LDS reduction? -- no:
// heads of arbitrary size D <= 64 are reduced with a masked segmented scan over the wave.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void gt_edge_attention_generic_kernel(const EdgeAttnParams p) {
  const int lane = threadIdx.x & 63;
  const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int64_t wave = (int64_t)blockIdx.x * 4 + wib;
  const int64_t n_waves = (int64_t)gridDim.x * 4;
  // heads never straddle a wave: each wave covers hpw = max(1, 64 / D) whole heads
  const int hpw = 64 / p.D;
  const int ch_per_wave = hpw * p.D;
  const int n_slices = (p.C + ch_per_wave - 1) / ch_per_wave;
  const int head_lane = lane % p.D;  // position inside the head
  const bool in_head = lane < ch_per_wave;
  for (int64_t unit = wave; unit < p.n_dst * n_slices; unit += n_waves) {
    const int64_t node = unit / n_slices;
    const int slice = (int)(unit - node * n_slices);
    const int c = slice * ch_per_wave + lane;
    const bool active = in_head && c < p.C;
    const int cs = active ? c : 0;
    const float q = Elem<T>::load(static_cast<const T*>(p.q) + node * p.ldq + cs);
    const float* wrow = p.w + (int64_t)cs * p.edge_dim;
    const float bias = p.b[cs];
    const int e_begin = p.rowptr[node], e_end = p.rowptr[node + 1];
    float m = -INFINITY, l = 0.f, acc = 0.f;
    for (int e = e_begin; e < e_end; ++e) {
      const int64_t j = p.col[e];
      const float* at = p.attr + (int64_t)e * p.ea_ld;
      float ee = bias;
      for (int a = 0; a < p.edge_dim; ++a) ee = fmaf(wrow[a], at[a], ee);
      const float kk = Elem<T>::load(static_cast<const T*>(p.k) + j * p.ldkv + cs) + ee;
      const float vv = Elem<T>::load(static_cast<const T*>(p.v) + j * p.ldkv + cs) + ee;
      // head-wise sum of q*kk: every lane adds the partials of all lanes of its own head
      const float part = active ? q * kk : 0.f;
      float s = 0.f;
      const int head_base = lane - head_lane;
      for (int t = 0; t < p.D; ++t) s += __shfl(part, head_base + t, 64);
      s *= p.scale;
      const float mb = fmaxf(m, s);
      const float corr = __expf(m - mb);
      const float pe = __expf(s - mb);
      l = l * corr + pe;
      acc = acc * corr + pe * vv;
      m = mb;
    }
    float o = acc / (l + 1e-16f);
    if (p.xr != nullptr) o += Elem<T>::load(static_cast<const T*>(p.xr) + node * p.ldr + cs);
    if (active) Elem<T>::store(static_cast<T*>(p.out) + node * p.ldo + c, o);
  }
}

template <typename T, int VEC, int LPH, int EDP>
static bool launch_fast(const EdgeAttnParams& p, hipStream_t st) {
  // W_e (+ bias row) in LDS: (EDP + 1) * C floats per workgroup of 4 waves
  const size_t lds = (size_t)(EDP + 1) * p.C * sizeof(float);
  if (lds > 160 * 1024 || p.C % VEC != 0) return false;
  auto kern = gt_edge_attention_kernel<T, VEC, LPH, EDP>;
  if (lds > 64 * 1024) {
    static bool raised = false;  // per instantiation
    if (!raised) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024) != hipSuccess)
        return false;
      raised = true;
    }
  }
  // persistent-style grid: 8 XCDs x blocks_per_xcd, 4 waves per block, waves_per_xcd % n_slices == 0
  constexpr int WPB = 4;
  const int64_t units_per_xcd = ((p.n_dst + 7) / 8) * p.n_slices;
  int64_t bpx = (units_per_xcd + WPB - 1) / WPB;
  const int64_t resident = 160 * 1024 / (lds > 0 ? lds : 1);  // workgroups per CU that fit in LDS
  int64_t cap = 32 * (resident > 4 ? 4 : (resident < 1 ? 1 : resident));  // 32 CUs per XCD
  if (bpx > cap) bpx = cap;
  if (bpx < 1) bpx = 1;
  while ((bpx * WPB) % p.n_slices != 0) ++bpx;
  hipLaunchKernelGGL(kern, dim3((unsigned)(8 * bpx)), dim3(64 * WPB), lds, st, p, p.attr, p.rowptr, p.col);
  return true;
}

template <typename T, int VEC, int LPH>
static bool dispatch_edp(const EdgeAttnParams& p, hipStream_t st) {
  const int edp = (p.edge_dim + 3) / 4 * 4;
  switch (edp) {
    case 4: return launch_fast<T, VEC, LPH, 4>(p, st);
    case 8: return launch_fast<T, VEC, LPH, 8>(p, st);
    case 12: return launch_fast<T, VEC, LPH, 12>(p, st);
    case 16: return launch_fast<T, VEC, LPH, 16>(p, st);
    default: return false;
  }
}

template <typename T>
static bool dispatch_fast(const EdgeAttnParams& p, hipStream_t st) {
  constexpr int VEC = 16 / sizeof(T);
  if (p.D % VEC != 0) return false;
  switch (p.D / VEC) {
    case 1: return dispatch_edp<T, VEC, 1>(p, st);
    case 2: return dispatch_edp<T, VEC, 2>(p, st);
    case 4: return dispatch_edp<T, VEC, 4>(p, st);
    case 8: return dispatch_edp<T, VEC, 8>(p, st);
    case 16: return dispatch_edp<T, VEC, 16>(p, st);
    default: return false;
  }
}

template <typename T>
static int edge_attention_launch(EdgeAttnParams p, hipStream_t st) {
  constexpr int VEC = 16 / sizeof(T);
  const bool aligned = ((uintptr_t)p.q % 16 == 0) && ((uintptr_t)p.k % 16 == 0) && ((uintptr_t)p.v % 16 == 0) &&
                       ((uintptr_t)p.out % 16 == 0) && (p.xr == nullptr || (uintptr_t)p.xr % 16 == 0) &&
                       (p.ldq % VEC == 0) && (p.ldkv % VEC == 0) && (p.ldo % VEC == 0) &&
                       (p.xr == nullptr || p.ldr % VEC == 0) && ((uintptr_t)p.attr % 16 == 0) && (p.ea_ld % 4 == 0);
  p.n_slices = (p.C + 64 * VEC - 1) / (64 * VEC);
  if (aligned && p.ea_ld >= (p.edge_dim + 3) / 4 * 4 && dispatch_fast<T>(p, st))
    return check_launch("anemoi_gt_edge_attention");
  ANEMOI_REQUIRE(p.D <= 64, ANEMOI_ERR_UNSUPPORTED, "anemoi_gt_edge_attention: head size %d > 64 needs D %% %d == 0",
                 p.D, VEC);
  const int hpw = 64 / p.D;
  const int64_t units = p.n_dst * ((p.C + hpw * p.D - 1) / (hpw * p.D));
  int64_t blocks = (units + 3) / 4;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL((gt_edge_attention_generic_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, st, p);
  return check_launch("anemoi_gt_edge_attention(generic)");
}

}  // namespace anemoi

using namespace anemoi;

extern "C" int anemoi_gt_edge_attention(int dtype, const void* q, int64_t ldq, const void* k, const void* v,
                                        int64_t ldkv, const void* x_r, int64_t ldr, const float* edge_attr, int ea_ld,
                                        int edge_dim, const float* w_edge, const float* b_edge,
                                        const int32_t* rowptr, const int32_t* col, void* out, int64_t ldo,
                                        int64_t n_dst, int C, int H, anemoi_stream_t stream) {
  ANEMOI_REQUIRE(q && k && v && out && rowptr && w_edge && b_edge, ANEMOI_ERR_INVALID,
                 "anemoi_gt_edge_attention: null pointer");
  ANEMOI_REQUIRE(C > 0 && H > 0 && C % H == 0, ANEMOI_ERR_INVALID,
                 "anemoi_gt_edge_attention: C=%d not divisible by H=%d", C, H);
  ANEMOI_REQUIRE(edge_dim > 0 && ea_ld >= edge_dim, ANEMOI_ERR_INVALID,
                 "anemoi_gt_edge_attention: edge_dim=%d ea_ld=%d", edge_dim, ea_ld);
  ANEMOI_REQUIRE(ldq >= C && ldkv >= C && ldo >= C && (x_r == nullptr || ldr >= C), ANEMOI_ERR_INVALID,
                 "anemoi_gt_edge_attention: leading dimension smaller than C");
  ANEMOI_REQUIRE(n_dst >= 0, ANEMOI_ERR_INVALID, "anemoi_gt_edge_attention: n_dst < 0");
  if (n_dst == 0) return ANEMOI_OK;
  ANEMOI_REQUIRE(col != nullptr && edge_attr != nullptr, ANEMOI_ERR_INVALID,
                 "anemoi_gt_edge_attention: null edge arrays");
  EdgeAttnParams p;
  p.q = q; p.k = k; p.v = v; p.xr = x_r; p.out = out;
  p.ldq = ldq; p.ldkv = ldkv; p.ldr = ldr; p.ldo = ldo;
  p.attr = edge_attr; p.w = w_edge; p.b = b_edge; p.rowptr = rowptr; p.col = col;
  p.n_dst = n_dst; p.C = C; p.D = C / H; p.ea_ld = ea_ld; p.edge_dim = edge_dim; p.n_slices = 1;
  p.scale = 1.0f / sqrtf((float)(C / H));
  if (dtype == ANEMOI_F32) return edge_attention_launch<float>(p, as_stream(stream));
  if (dtype == ANEMOI_BF16) return edge_attention_launch<bf16_t>(p, as_stream(stream));
  return fail(ANEMOI_ERR_UNSUPPORTED, "anemoi_gt_edge_attention: dtype %d", dtype);
}

extern "C" int anemoi_gt_edge_attention_folded(int dtype, const void* q, int64_t ldq, const void* k, const void* v,
                                               int64_t ldkv, const void* x_r, int64_t ldr, const void* u, int64_t ldu,
                                               const float* edge_attr, int up, const int32_t* rowptr,
                                               const int32_t* col, void* out, int64_t ldo, int64_t n_dst, int C, int H,
                                               anemoi_stream_t stream) {
  ANEMOI_REQUIRE(q && k && v && u && out && rowptr, ANEMOI_ERR_INVALID,
                 "anemoi_gt_edge_attention_folded: null pointer");
  ANEMOI_REQUIRE(C > 0 && H > 0 && C % H == 0, ANEMOI_ERR_INVALID,
                 "anemoi_gt_edge_attention_folded: C=%d not divisible by H=%d", C, H);
  ANEMOI_REQUIRE(ldq >= C && ldkv >= C && ldu >= (int64_t)H * up && ldo >= (int64_t)C + (int64_t)H * up &&
                     (x_r == nullptr || ldr >= C),
                 ANEMOI_ERR_INVALID, "anemoi_gt_edge_attention_folded: leading dimension too small");
  ANEMOI_REQUIRE(n_dst >= 0, ANEMOI_ERR_INVALID, "anemoi_gt_edge_attention_folded: n_dst < 0");
  if (n_dst == 0) return ANEMOI_OK;
  ANEMOI_REQUIRE(col != nullptr && edge_attr != nullptr, ANEMOI_ERR_INVALID,
                 "anemoi_gt_edge_attention_folded: null edge arrays");
  const int esz = dtype == ANEMOI_BF16 ? 2 : 4;
  const int vec = 16 / esz;
  const bool aligned = ((uintptr_t)q % 16 == 0) && ((uintptr_t)k % 16 == 0) && ((uintptr_t)v % 16 == 0) &&
                       ((uintptr_t)u % 16 == 0) && ((uintptr_t)out % 16 == 0) &&
                       (x_r == nullptr || ((uintptr_t)x_r % 16 == 0 && ldr % vec == 0)) && ldq % vec == 0 &&
                       ldkv % vec == 0 && ldu % vec == 0 && ldo % vec == 0 && ((uintptr_t)edge_attr % 16 == 0) &&
                       ((int64_t)up * esz) % 8 == 0 && ((int64_t)C * esz) % 16 == 0;
  ANEMOI_REQUIRE(aligned, ANEMOI_ERR_UNSUPPORTED, "anemoi_gt_edge_attention_folded: operands must be 16-byte aligned");
  EdgeFoldParams p;
  p.q = q; p.k = k; p.v = v; p.xr = x_r; p.u = u; p.out = out;
  p.ldq = ldq; p.ldkv = ldkv; p.ldr = ldr; p.ldu = ldu; p.ldo = ldo;
  p.attr = edge_attr; p.rowptr = rowptr; p.col = col;
  p.n_dst = n_dst; p.C = C; p.D = C / H;
  p.n_slices = (C + 64 * vec - 1) / (64 * vec);
  p.scale = 1.0f / sqrtf((float)(C / H));
  static const int stream_hint = [] {
    const char* e = getenv("ANEMOI_AMD_EDGE_NT");  // A/B knob; measured -3 % on all three graphs of config 3
    return e ? atoi(e) : 1;
  }();
  p.stream_hint = stream_hint;
  bool ok = false;
  if (dtype == ANEMOI_F32) ok = dispatch_folded<float>(p, up, as_stream(stream));
  else if (dtype == ANEMOI_BF16) ok = dispatch_folded<bf16_t>(p, up, as_stream(stream));
  else return fail(ANEMOI_ERR_UNSUPPORTED, "anemoi_gt_edge_attention_folded: dtype %d", dtype);
  ANEMOI_REQUIRE(ok, ANEMOI_ERR_UNSUPPORTED,
                 "anemoi_gt_edge_attention_folded: unsupported shape (D=%d, UP=%d); use anemoi_gt_edge_attention", C / H,
                 up);
  return check_launch("anemoi_gt_edge_attention_folded");
}
