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
#include "edge_common.hpp"

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
  float* lse;  // optional [n_dst, H]: m + log(l + 1e-16) of the destination's softmax (alpha = exp(s - lse)), for the backward
  int64_t ldq, ldkv, ldr, ldu, ldo;
  const float* attr;
  const int32_t* rowptr;
  const int32_t* col;
  int64_t n_dst;
  int C, D, n_slices;
  float scale;
  int stream_hint;  // 1: q / x_r / out are nontemporal so that they do not evict gathered k|v rows from the XCD's L2
};

template <typename T, int VEC, int LPH, int UP, int U = 4>
__global__ __launch_bounds__(256) void gt_edge_attention_folded_kernel(const EdgeFoldParams p,
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

  const T* qb = static_cast<const T*>(p.q) + c0;
  const T* kb = static_cast<const T*>(p.k) + c0;
  const T* vb = static_cast<const T*>(p.v) + c0;
  const T* ub = static_cast<const T*>(p.u) + head * UP + a_ld;
  const float* ab = attr_ + a_ld;

  for (int64_t node = n0 + node_first; node < n1; node += node_stride) {
    const int e_begin = rowptr_[node], e_end = rowptr_[node + 1];
    QK<T, VEC> qk;
    float u[APL];
    RawWords<T, VEC> xr_raw;
    {
      float qf[VEC];
      if (p.stream_hint) load_stream<T, VEC>(qb + node * p.ldq, qf);
      else VecIO<T, VEC>::load(qb + node * p.ldq, qf);
      qk.set(qf);
      VecIO<T, APL>::load(ub + node * p.ldu, u);
      // x_r is requested here, with q, not behind the edge loop where it is consumed: one dependent HBM round trip per
      // destination less (-2 %; requesting the NEXT destination's rows ahead as well was measured and is slower, see
      // DESIGN.md section 4.2)
      if (p.xr != nullptr) xr_raw.load(static_cast<const char*>(p.xr) + node * p.ldr * (int64_t)sizeof(T),
                                       (uint32_t)(c0 * (int)sizeof(T)), p.stream_hint != 0);
#pragma unroll
      for (int i = 0; i < APL; ++i) u[i] *= amask;
    }
    float m = -INFINITY, l = 0.f;
    // the value accumulators live as f32 pairs: rescale and accumulate are v_pk_mul_f32 / v_pk_fma_f32 (two channels per
    // issue slot) -- this loop is bound by its instruction streams, not by bytes
    typedef __attribute__((ext_vector_type(2))) float f32x2_t;
    constexpr int VP = (VEC + 1) / 2;
    f32x2_t acc[VP];
    float tacc[APL];
#pragma unroll
    for (int i = 0; i < VP; ++i) acc[i] = f32x2_t{0.f, 0.f};
#pragma unroll
    for (int a = 0; a < APL; ++a) tacc[a] = 0.f;

    for (int e = e_begin; e < e_end; e += U) {
      Raw kr[U], vr[U];
      float at[U][APL];
      float s[U];
#pragma unroll
      for (int uu = 0; uu < U; ++uu) {
        if (e + uu < e_end) {
          const int64_t j = col_[e + uu];
          kr[uu] = *reinterpret_cast<const Raw*>(kb + j * p.ldkv);
          vr[uu] = *reinterpret_cast<const Raw*>(vb + j * p.ldkv);
          VecIO<float, APL>::load(ab + (int64_t)(e + uu) * UP, at[uu]);
        }
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
    if (p.xr != nullptr) {
      float r[VEC];
      xr_raw.get(r);
#pragma unroll
      for (int i = 0; i < VEC; ++i) o[i] += r[i];
    }
    T* on = static_cast<T*>(p.out) + node * p.ldo;
    if (active) {
      if (p.stream_hint) store_stream<T, VEC>(on + c0, o);
      else VecIO<T, VEC>::store(on + c0, o);
    }
    if (active && a_own) {  // this lane's APL values of t~_i,h
      float t4[APL];
#pragma unroll
      for (int a = 0; a < APL; ++a) t4[a] = tacc[a] * inv;
      VecIO<T, APL>::store(on + p.C + head * UP + a0, t4);
    }
    if (p.lse != nullptr && active && (gls % LPH) == 0) p.lse[node * (p.C / p.D) + head] = m + __logf(l + 1e-16f);
  }
}

template <typename T, int VEC, int LPH, int UP>
static void launch_folded(const EdgeFoldParams& p, hipStream_t st) {
  constexpr int WPB = 4;
  // U = 4 edges (8 independent 16-byte gathers per lane) in flight and 5 resident workgroups per CU: the winners of the
  // round-1 / round-2 sweeps (U in {2, 4, 8}, 2 .. 8 workgroups per CU: +-3 %; profiles/r02_edge_kernels.md)
  constexpr int wgs_per_cu = 5;
  const int64_t units_per_xcd = ((p.n_dst + 7) / 8) * p.n_slices;
  int64_t bpx = (units_per_xcd + WPB - 1) / WPB;
  if (bpx > 32 * wgs_per_cu) bpx = 32 * wgs_per_cu;  // resident workgroups per CU x 32 CUs per XCD
  if (bpx < 1) bpx = 1;
  while ((bpx * WPB) % p.n_slices != 0) ++bpx;
  // (a register-double-buffered software pipeline across destinations was measured and removed: 0.30 / 1.98 / 0.79 ms
  // against 0.19 / 1.21 / 0.60 ms of this loop on the mesh / decoder / encoder graphs of config 3 -- the second
  // register set costs a wave per SIMD, which hurts more than the overlap helps)
  hipLaunchKernelGGL((gt_edge_attention_folded_kernel<T, VEC, LPH, UP, 4>), dim3((unsigned)(8 * bpx)), dim3(64 * WPB), 0,
                     st, p, p.attr, p.rowptr, p.col);
}

// ---------------------------------------------------------------------------------------------
// Folded path, SCHEDULED (round 5).  Same arithmetic, same per-destination summation order, bit-identical results; what
// changes is who walks which destination and when the index chain is resolved:
//  * a static schedule ``sched [8 XCDs][slots][steps]`` (host-built, runtime.EdgePlan.schedule) names the destinations
//    of every wave slot.  At step i the slots of an XCD still work on one contiguous group of destinations (the L2
//    window is unchanged), but inside the group the heavy destinations go to the slots with the least work so far: on
//    the ico-6 multi-scale mesh (in-degree 6 ... 36, mean 8) the plain round-robin leaves the busiest wave with 204
//    edges against a mean of 128, and a launch of resident waves lasts as long as its busiest wave;
//  * the chain  schedule entry -> row pointers -> source ids  is resolved by SCALAR loads one destination ahead each
//    (entry i + 3, row pointers i + 2, the first NPF source ids of destination i + 1 while destination i is processed):
//    SGPRs only, no vector register is spent on it, and the first 2 x U row gathers of a destination leave as soon as
//    the previous destination's registers are free instead of two dependent round trips later.
// The entries of a slot end with >= 3 times -1.
// ---------------------------------------------------------------------------------------------
// W consecutive 32-bit words through a buffer descriptor (W = 1, 2, 4, 6, 8: b32 / b64 / b128 pieces)
template <int W>
__device__ __forceinline__ void buffer_load_words(__amdgpu_buffer_rsrc_t rs, int voff, int soff, uint32_t (&w)[W]) {
  typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
  typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
  static_assert(W == 1 || W % 2 == 0, "one word or whole 8-byte pieces");
  if constexpr (W == 1) {
    w[0] = __builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, 0);
  } else {
#pragma unroll
    for (int i = 0; i + 4 <= W; i += 4) {
      const u32x4_t t = __builtin_amdgcn_raw_buffer_load_b128(rs, voff + i * 4, soff, 0);
      w[i] = t.x; w[i + 1] = t.y; w[i + 2] = t.z; w[i + 3] = t.w;
    }
    if constexpr (W % 4 == 2) {
      const u32x2_t t = __builtin_amdgcn_raw_buffer_load_b64(rs, voff + (W - 2) * 4, soff, 0);
      w[W - 2] = t.x; w[W - 1] = t.y;
    }
  }
}
template <int W>
__device__ __forceinline__ void buffer_store_words(__amdgpu_buffer_rsrc_t rs, int voff, int soff, const uint32_t (&w)[W]) {
  typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
  typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
  static_assert(W == 1 || W % 2 == 0, "one word or whole 8-byte pieces");
  if constexpr (W == 1) {
    __builtin_amdgcn_raw_buffer_store_b32(w[0], rs, voff, soff, 0);
  } else {
#pragma unroll
    for (int i = 0; i + 4 <= W; i += 4)
      __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{w[i], w[i + 1], w[i + 2], w[i + 3]}, rs, voff + i * 4, soff, 0);
    if constexpr (W % 4 == 2) __builtin_amdgcn_raw_buffer_store_b64(u32x2_t{w[W - 2], w[W - 1]}, rs, voff + (W - 2) * 4, soff, 0);
    // (wide store with an SGPR soffset: the compiler does not pad the data-register hazard, see the scheduled kernel)
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 1" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  }
}

// N consecutive floats through a buffer descriptor (voffset = the lane's byte offset, soffset = the row's, scalar)
template <int N>
__device__ __forceinline__ void buffer_load_f32(__amdgpu_buffer_rsrc_t rs, int voff, int soff, float (&r)[N]) {
  typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
  typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
  static_assert(N % 2 == 0, "whole 8-byte pieces");
#pragma unroll
  for (int i = 0; i + 4 <= N; i += 4) {
    const u32x4_t t = __builtin_amdgcn_raw_buffer_load_b128(rs, voff + i * 4, soff, 0);
    r[i] = __uint_as_float(t.x); r[i + 1] = __uint_as_float(t.y); r[i + 2] = __uint_as_float(t.z); r[i + 3] = __uint_as_float(t.w);
  }
  if constexpr (N % 4 == 2) {
    const u32x2_t t = __builtin_amdgcn_raw_buffer_load_b64(rs, voff + (N - 2) * 4, soff, 0);
    r[N - 2] = __uint_as_float(t.x); r[N - 1] = __uint_as_float(t.y);
  }
}

template <typename T, int VEC, int LPH, int UP, int U, int NPF>
__global__ __launch_bounds__(256) void gt_edge_attention_folded_sched_kernel(const EdgeFoldParams p,
                                                                         const float* __restrict__ attr_,
                                                                         const int32_t* __restrict__ rowptr_,
                                                                         const int32_t* __restrict__ col_,
                                                                         const int32_t* __restrict__ sched_,
                                                                         int slots, int steps) {
  using Raw = typename RawVec<T, VEC>::type;
  constexpr int APL = attrs_per_lane(UP, LPH);
  static_assert(UP % APL == 0 && APL * LPH >= UP, "a lane owns APL whole attributes or none");
  static_assert(NPF % U == 0, "whole chunks of prefetched source ids");
  const int lane = threadIdx.x & 63;
  const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int xcd = blockIdx.x & 7;
  const int wave_in_xcd = (int)(blockIdx.x >> 3) * 4 + wib;
  const int slice = wave_in_xcd % p.n_slices;
  const int slot = wave_in_xcd / p.n_slices;
  if (slot >= slots) return;
  const int32_t* my = sched_ + ((int64_t)xcd * slots + slot) * steps;
  const int n_edges = max(rowptr_[p.n_dst], 1);  // (the caller's col / attribute arrays hold at least one entry)

  const int lanes_total = p.C / VEC;
  const int gl = slice * 64 + lane;
  const bool active = gl < lanes_total;
  const int gls = active ? gl : 0;
  const int c0 = gls * VEC;
  const int head = gls / LPH;
  const int a0 = (gls % LPH) * APL;
  const bool a_own = a0 < UP;
  const int a_ld = a_own ? a0 : 0;
  const float amask = a_own ? 1.f : 0.f;

  static_assert(sizeof(Raw) == 16, "one 16-byte row slice per lane");
  typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
  const int lane_off = c0 * (int)sizeof(T);  // the lane's byte offset inside a node row
  const int attr_off = a_ld * 4;             // ... inside an attribute row
  const uint32_t row_bytes = (uint32_t)(p.ldkv * (int64_t)sizeof(T));
  const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.k), 0, -1, 0x00020000);
  const __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.v), 0, -1, 0x00020000);
  const __amdgpu_buffer_rsrc_t ars =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(attr_), 0, -1, 0x00020000);
  const __amdgpu_buffer_rsrc_t qrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.q), 0, -1, 0x00020000);
  const __amdgpu_buffer_rsrc_t urs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.u), 0, -1, 0x00020000);
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.xr), 0, -1, 0x00020000);
  const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, -1, 0x00020000);
  const uint32_t q_row_bytes = (uint32_t)(p.ldq * (int64_t)sizeof(T)), u_row_bytes = (uint32_t)(p.ldu * (int64_t)sizeof(T));
  const uint32_t xr_row_bytes = (uint32_t)(p.ldr * (int64_t)sizeof(T)), o_row_bytes = (uint32_t)(p.ldo * (int64_t)sizeof(T));
  const int u_off = (head * UP + a_ld) * (int)sizeof(T);
  const int t_off = (p.C + head * UP + a0) * (int)sizeof(T);
  typedef __attribute__((ext_vector_type(2))) float f32x2_t;
  constexpr int VP = (VEC + 1) / 2;

  // the first NPF source ids of a destination whose segment starts at rb (clamped reads at the very end of the array)
  auto load_cols = [&](int rb, int (&c)[NPF]) __attribute__((always_inline)) {
    if (rb + NPF <= n_edges) {
#pragma unroll
      for (int kk = 0; kk < NPF; ++kk) c[kk] = col_[rb + kk];
    } else {
#pragma unroll
      for (int kk = 0; kk < NPF; ++kk) c[kk] = col_[rb + kk < n_edges ? rb + kk : n_edges - 1];
    }
  };

  // scalar pipeline: destination 0 complete, destination 1 with its row pointers, destination 2 by id
  int node0 = my[0], node1 = my[1], node2 = my[2];
  if (node0 < 0) return;
  int rb0 = rowptr_[node0], re0 = rowptr_[node0 + 1];
  int rb1 = rowptr_[node1 < 0 ? 0 : node1], re1 = rowptr_[(node1 < 0 ? 0 : node1) + 1];
  int cols0[NPF];
  load_cols(rb0, cols0);

  for (int step = 0; node0 >= 0; ++step) {
    // ---- the index chain of the destinations behind this one (scalar loads, consumed at the end of this iteration)
    int cols1[NPF];
    load_cols(rb1, cols1);
    const int n2 = node2 < 0 ? 0 : node2;
    const int rb2 = rowptr_[n2], re2 = rowptr_[n2 + 1];
    const int node3 = my[step + 3];

    const int64_t node = node0;
    const int e_begin = rb0, e_end = re0;
    float m = -INFINITY, l = 0.f;
    f32x2_t acc[VP];
    float tacc[APL];
#pragma unroll
    for (int i = 0; i < VP; ++i) acc[i] = f32x2_t{0.f, 0.f};
#pragma unroll
    for (int a = 0; a < APL; ++a) tacc[a] = 0.f;
    QK<T, VEC> qk;
    float u[APL];
    Raw kr[U], vr[U];
    float at[U][APL];

    // U edges starting at CSR slot e, their source ids in j[]: request the k / v row slices and the attribute rows ...
    auto issue = [&](int e, const int (&j)[U]) __attribute__((always_inline)) {
#pragma unroll
      for (int uu = 0; uu < U; ++uu) {
        if (e + uu < e_end) {
          // buffer loads: descriptor of the whole k / v / attribute matrix, the row's byte offset as the (scalar) soffset,
          // the lane's byte offset as the one shared 32-bit voffset -- no 64-bit per-lane address is formed or kept
          // (twelve of them cost this kernel its fifth wave per SIMD); the host guarantees matrices below 4 GiB
          const int row = (int)((uint32_t)j[uu] * (uint32_t)row_bytes);
          const u32x4_t kw = __builtin_amdgcn_raw_buffer_load_b128(krs, lane_off, row, 0);
          const u32x4_t vw = __builtin_amdgcn_raw_buffer_load_b128(vrs, lane_off, row, 0);
          kr[uu] = __builtin_bit_cast(Raw, kw);
          vr[uu] = __builtin_bit_cast(Raw, vw);
          buffer_load_f32<APL>(ars, attr_off, (int)((uint32_t)(e + uu) * (uint32_t)(UP * 4)), at[uu]);
        }
      }
    };
    // ... and the online-softmax update of the plain folded kernel on them
    auto consume = [&](int e) __attribute__((always_inline)) {
      float s[U];
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
    };

    // the first chunk's gathers leave FIRST (their ids are already in SGPRs); q, u and x_r of the destination follow as raw
    // words, so that one s_waitcnt covers all of a destination's first loads (q / u requested ahead of the gathers put a
    // dependent round trip in front of them: the compiler converts u where it is loaded)
    {
      int j[U];
#pragma unroll
      for (int uu = 0; uu < U; ++uu) j[uu] = cols0[uu];
      issue(e_begin, j);
    }
    const u32x4_t q_raw = __builtin_amdgcn_raw_buffer_load_b128(qrs, lane_off, (int)((uint32_t)node * q_row_bytes), 2);
    RawWords<T, APL> u_raw;
    buffer_load_words<RawWords<T, APL>::W>(urs, u_off, (int)((uint32_t)node * u_row_bytes), u_raw.w);
    u32x4_t xr_raw = {0u, 0u, 0u, 0u};
    if (p.xr != nullptr) xr_raw = __builtin_amdgcn_raw_buffer_load_b128(xrs, lane_off, (int)((uint32_t)node * xr_row_bytes), 2);
    {
      const uint32_t qw[4] = {q_raw.x, q_raw.y, q_raw.z, q_raw.w};
      qk.set_raw(qw);
    }
    u_raw.get(u);
#pragma unroll
    for (int i = 0; i < APL; ++i) u[i] *= amask;
    if (e_begin < e_end) consume(e_begin);
#pragma unroll
    for (int c = 1; c < NPF / U; ++c) {
      if (e_begin + c * U < e_end) {
        int j[U];
#pragma unroll
        for (int uu = 0; uu < U; ++uu) j[uu] = cols0[c * U + uu];
        issue(e_begin + c * U, j);
        consume(e_begin + c * U);
      }
    }
    for (int e = e_begin + NPF; e < e_end; e += U) {  // in-degree > NPF: the rest of the ids as they are needed
      int j[U];
#pragma unroll
      for (int uu = 0; uu < U; ++uu) j[uu] = col_[e + uu < e_end ? e + uu : e_end - 1];
      issue(e, j);
      consume(e);
    }

    const float inv = 1.0f / (l + 1e-16f);
    float o[VEC];
    {
      // (acc * inv) + x_r as TWO roundings: the plain kernel's arithmetic, whose + x_r sits behind a branch and is therefore
      // never contracted into an fma with the scaling in front of it -- bit identity between the two kernels is a test
#pragma clang fp contract(off)
      RawWords<T, VEC> xw;  // (zeros without x_r: + 0.0f leaves the sums as they are)
      xw.w[0] = xr_raw.x; xw.w[1] = xr_raw.y; xw.w[2] = xr_raw.z; xw.w[3] = xr_raw.w;
      float r[VEC];
      xw.get(r);
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        const float scaled = acc[i >> 1][i & 1] * inv;
        o[i] = p.xr != nullptr ? scaled + r[i] : scaled;  // (no x_r: no add at all -- a -0.0 sum stays -0.0, as in the plain kernel)
      }
    }
    const int out_row = (int)((uint32_t)node * o_row_bytes);
    if (active) {
      uint32_t ow[4];
      pack_words<T, VEC>(o, ow);
      __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{ow[0], ow[1], ow[2], ow[3]}, ors, lane_off, out_row, 2);
      // gfx950: a 16-byte buffer store whose data registers the very next VALU instruction overwrites stores the NEW value in
      // part of dword 1 (lanes 12 .. 15 of every 16); the compiler pads that hazard except when soffset is an SGPR, as here
      // (csrc/gemm.hip's epilogue met it in round 2).  Seen in the UP = 16 instantiation, where the t words are computed
      // right behind the store: pad by hand.
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_nop 1" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
    }
    if (active && a_own) {  // this lane's APL values of t~_i,h
      float t4[APL];
#pragma unroll
      for (int a = 0; a < APL; ++a) t4[a] = tacc[a] * inv;
      uint32_t tw[RawWords<T, APL>::W];
      pack_words<T, APL>(t4, tw);
      buffer_store_words<RawWords<T, APL>::W>(ors, t_off, out_row, tw);
    }
    if (p.lse != nullptr && active && (gls % LPH) == 0) p.lse[node * (p.C / p.D) + head] = m + __logf(l + 1e-16f);

    // ---- rotate the scalar pipeline
    node0 = node1; rb0 = rb1; re0 = re1;
#pragma unroll
    for (int kk = 0; kk < NPF; ++kk) cols0[kk] = cols1[kk];
    node1 = node2; rb1 = rb2; re1 = re2;
    node2 = node3;
  }
}

// geometry of the scheduled launch: wave slots per XCD (every slot is n_slices waves, one per 512-channel slice)
static void sched_shape(int64_t n_dst, int n_slices, int* slots, int* steps) {
  constexpr int WPB = 4;
  // Resident workgroups per CU.  A long launch (the ico-6 mesh: 5 120 destinations x 2 slices per XCD) is bound by the CU's
  // texture-address unit -- its busy time is the same for every form of this kernel (PMC: 229 k cycles per launch and CU,
  // 32 cycles per 16-byte-per-lane wave load) -- and runs FASTER with fewer waves contending for it: 3 / 4 / 5 workgroups
  // per CU 0.136 / 0.138 / 0.141 ms.  A short launch (O96 -> ico-5: 1 280 destinations per XCD, a handful of steps per
  // wave) is bound by its few dependent round trips and wants every wave it can get: 0.0239 / 0.0232 / 0.0205 ms.
  static const int wgs_env = getenv("ANEMOI_AMD_EDGE_WGS") ? atoi(getenv("ANEMOI_AMD_EDGE_WGS")) : 0;  // lab switch
  const int64_t per_xcd = (n_dst + 7) / 8;
  const int wgs_per_cu = wgs_env > 0 ? wgs_env : (per_xcd * n_slices >= 12 * 32 * 5 * WPB ? 3 : 5);
  int64_t bpx = (per_xcd * n_slices + WPB - 1) / WPB;
  if (bpx > 32 * wgs_per_cu) bpx = 32 * wgs_per_cu;
  if (bpx < 1) bpx = 1;
  while ((bpx * WPB) % n_slices != 0) ++bpx;
  *slots = (int)(bpx * WPB / n_slices);
  *steps = (int)((per_xcd + *slots - 1) / *slots) + 3;
}

template <typename T, int VEC, int LPH, int UP>
static void launch_folded_sched(const EdgeFoldParams& p, const int32_t* sched, int slots, int steps, hipStream_t st) {
  const unsigned blocks = (unsigned)(8 * ((int64_t)slots * p.n_slices / 4));
  static const int u_env = getenv("ANEMOI_AMD_EDGE_U") ? atoi(getenv("ANEMOI_AMD_EDGE_U")) : 4;  // lab switch
  if (u_env == 6)
    hipLaunchKernelGGL((gt_edge_attention_folded_sched_kernel<T, VEC, LPH, UP, 6, 12>), dim3(blocks), dim3(256), 0, st, p,
                       p.attr, p.rowptr, p.col, sched, slots, steps);
  else if (u_env == 3)
    hipLaunchKernelGGL((gt_edge_attention_folded_sched_kernel<T, VEC, LPH, UP, 3, 12>), dim3(blocks), dim3(256), 0, st, p,
                       p.attr, p.rowptr, p.col, sched, slots, steps);
  else
    hipLaunchKernelGGL((gt_edge_attention_folded_sched_kernel<T, VEC, LPH, UP, 4, 8>), dim3(blocks), dim3(256), 0, st, p,
                       p.attr, p.rowptr, p.col, sched, slots, steps);
}

// ---------------------------------------------------------------------------------------------
// Folded path, LDS TILES (round 6).  The scheduled kernel above is bound by the CU's texture-address unit: every edge
// gathers its k and v row slices again (32 cycles of that unit per 16-byte-per-lane wave load), although on a mesh in
// Morton order ~25 consecutive destinations name only ~70 distinct sources for their ~200 edges.  Here a workgroup takes one
// TILE -- a run of <= 32 consecutive destinations, host-built (runtime.EdgeTiles) -- and one 128-channel slice, stages the
// k|v slices of the tile's distinct sources ONCE in LDS by LDS-DMA (4 sources per 1-KiB piece), the tile's attribute rows
// (one contiguous block of the CSR-ordered matrix), the LDS slot byte of every edge and the destination list behind them,
// and then computes: a wave walks a PASS of four destinations at once, one per 16-lane row (16 lanes x 8 channels = the
// slice), each row with its own edge range; per edge the scheduled kernel's arithmetic with ds_read_b128 in place of the
// gathers.  Per destination the edge order, the batches of U = 4 and every operation are those of the plain kernel: the
// results are bit-identical (test_gt_edge_attention_folded_tiles_is_the_plain_kernel_bit_for_bit).  Passes (heavy
// destinations first) are dealt to the four waves by an LDS ticket; which wave computes a destination changes nothing.
//   hdr  [n_tiles][8]     e0, n_edges, src_off, n_src, slot_off (16-byte aligned), n_dst of the tile
//   dst  [n_tiles][32][2] per (pass, row): destination (-1: none), (first edge - e0) << 8 | in-degree
//   LDS: k slices [src_cap][256 B] | v slices [src_cap][256 B] | attribute rows [edge_cap][UP x 4 B] | slots [edge_cap] |
//        destination list [256 B] | ticket
// ---------------------------------------------------------------------------------------------
struct EdgeTileLists {
  const int32_t* hdr;
  const int32_t* dst;
  const int32_t* src;
  const uint8_t* slot;
  const int32_t* xcd;
  int src_cap, edge_cap;
  int64_t n_src;
};

template <typename T, int LPH, int UP>
__global__ __launch_bounds__(256) void gt_edge_attention_folded_tiles_kernel(const EdgeFoldParams p,
                                                                         const float* __restrict__ attr_,
                                                                         const EdgeTileLists tl) {
  constexpr int VEC = 8, U = 4;
  using Raw = typename RawVec<T, VEC>::type;
  constexpr int APL = attrs_per_lane(UP, LPH);
  static_assert(sizeof(T) == 2 && UP % APL == 0 && APL * LPH >= UP, "bf16; a lane owns APL whole attributes or none");
  typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
  typedef __attribute__((ext_vector_type(2))) float f32x2_t;
  constexpr int VP = VEC / 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int xcd = blockIdx.x & 7;
  const int idx = (int)(blockIdx.x >> 3);
  const int slice = idx % p.n_slices;
  const int ti = tl.xcd[xcd] + idx / p.n_slices;
  if (ti >= tl.xcd[xcd + 1]) return;
  const int32_t* hd = tl.hdr + (int64_t)ti * 8;
  const int e0 = hd[0], ne = hd[1], so = hd[2], ns = hd[3], slot_off = hd[4], nd = hd[5];
  const int n_pass = (nd + 3) >> 2;

  char* const ks = smem;
  char* const vs = ks + tl.src_cap * 256;
  char* const as = vs + tl.src_cap * 256;
  char* const ss = as + tl.edge_cap * (UP * 4);
  char* const ds = ss + tl.edge_cap;  // (edge_cap is a multiple of 16)
  int* const ticket = reinterpret_cast<int*>(ds + 256);

  const int row = lane >> 4, l16 = lane & 15;
  const int gls = slice * 16 + l16;  // the lane's channel group in the whole row
  const int head = gls / LPH;
  const int a0 = (gls % LPH) * APL;
  const bool a_own = a0 < UP;
  const int a_ld = a_own ? a0 : 0;
  const float amask = a_own ? 1.f : 0.f;
  const int lane_off = gls * VEC * (int)sizeof(T);  // the lane's byte offset inside a node row
  const uint32_t row_bytes = (uint32_t)(p.ldkv * (int64_t)sizeof(T));
  const uint32_t q_row_bytes = (uint32_t)(p.ldq * (int64_t)sizeof(T)), u_row_bytes = (uint32_t)(p.ldu * (int64_t)sizeof(T));
  const uint32_t xr_row_bytes = (uint32_t)(p.ldr * (int64_t)sizeof(T)), o_row_bytes = (uint32_t)(p.ldo * (int64_t)sizeof(T));
  const int u_off = (head * UP + a_ld) * (int)sizeof(T);
  const int t_off = (p.C + head * UP + a0) * (int)sizeof(T);

  // ---- staging: everything the tile's edges read, once
  {
    const __amdgpu_buffer_rsrc_t krs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.k), 0, (int)(tl.n_src * row_bytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t vrs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.v), 0, (int)(tl.n_src * row_bytes), 0x00020000);
    for (int i = wid; 4 * i < ns; i += 4) {  // piece i: the slices of sources 4 i .. 4 i + 3, one per 16-lane row
      const int sidx = 4 * i + row;
      const int id = sidx < ns ? tl.src[so + sidx] : -1;
      const int voff = id >= 0 ? (int)((uint32_t)id * row_bytes) + lane_off : (int)0x7ffffff0;  // (no source: out of range, zeros)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(krs, (__attribute__((address_space(3))) void*)(ks + i * 1024), 16, voff, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(vrs, (__attribute__((address_space(3))) void*)(vs + i * 1024), 16, voff, 0, 0, 0);
    }
    const int abytes = ne * (UP * 4);
    const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(attr_) + (int64_t)e0 * UP, 0, abytes, 0x00020000);
    for (int j = wid; j * 1024 < abytes; j += 4)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ars, (__attribute__((address_space(3))) void*)(as + j * 1024), 16, lane * 16,
                                               j * 1024, 0, 0);
    if (wid == 0) {
      const __amdgpu_buffer_rsrc_t srs =
          __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(tl.slot) + slot_off, 0, (ne + 3) & ~3, 0x00020000);
      for (int j = 0; j * 256 < ne; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srs, (__attribute__((address_space(3))) void*)(ss + j * 256), 4, lane * 4,
                                                 j * 256, 0, 0);
    }
    if (wid == 1) {
      const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<int32_t*>(tl.dst) + (int64_t)ti * 64, 0, 256, 0x00020000);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(drs, (__attribute__((address_space(3))) void*)ds, 4, lane * 4, 0, 0, 0);
    }
    if (threadIdx.x == 0) *ticket = 4;  // passes 0 .. 3 are the waves' first ones
  }

  const __amdgpu_buffer_rsrc_t qrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.q), 0, -1, 0x00020000);
  const __amdgpu_buffer_rsrc_t urs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.u), 0, -1, 0x00020000);
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.xr), 0, -1, 0x00020000);
  const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, -1, 0x00020000);

  // a pass's per-destination operands, requested as raw words (converted at first use)
  struct PassIn {
    int node, pk;
    u32x4_t q, xr;
    RawWords<T, APL> u;
  };
  auto fetch = [&](int node, int pk, PassIn& in) __attribute__((always_inline)) {
    in.node = node;
    in.pk = pk;
    in.q = u32x4_t{0u, 0u, 0u, 0u};
    in.xr = u32x4_t{0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < RawWords<T, APL>::W; ++i) in.u.w[i] = 0u;
    if (node >= 0) {
      in.q = __builtin_amdgcn_raw_buffer_load_b128(qrs, (int)((uint32_t)node * q_row_bytes) + lane_off, 0, 2);
      buffer_load_words<RawWords<T, APL>::W>(urs, (int)((uint32_t)node * u_row_bytes) + u_off, 0, in.u.w);
      if (p.xr != nullptr)
        in.xr = __builtin_amdgcn_raw_buffer_load_b128(xrs, (int)((uint32_t)node * xr_row_bytes) + lane_off, 0, 2);
    }
  };

  PassIn cur;
  {
    int node = -1, pk = 0;
    if (wid < n_pass) {  // the wave's first pass, straight from the list in memory (the LDS copy is still on its way)
      const int2 e = *reinterpret_cast<const int2*>(tl.dst + (int64_t)ti * 64 + (wid * 4 + row) * 2);
      node = e.x;
      pk = e.y;
    }
    fetch(node, pk, cur);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  int ps = wid;
  while (ps < n_pass) {
    // the next pass of this wave: ticket, list entry, operands -- all in flight under this pass's edges
    int nps = 0;
    if (lane == 0) nps = atomicAdd(ticket, 1);
    nps = __builtin_amdgcn_readfirstlane(nps);
    PassIn nxt;
    {
      int node = -1, pk = 0;
      if (nps < n_pass) {
        const int2 e = *reinterpret_cast<const int2*>(ds + (nps * 4 + row) * 8);
        node = e.x;
        pk = e.y;
      }
      fetch(node, pk, nxt);
    }

    const int node = cur.node;
    const int deg = cur.pk & 255;
    const int er = cur.pk >> 8;  // first edge of the destination, relative to the tile's
    int nchunk = (deg + U - 1) / U;
    nchunk = max(max(__builtin_amdgcn_readlane(nchunk, 0), __builtin_amdgcn_readlane(nchunk, 16)),
                 max(__builtin_amdgcn_readlane(nchunk, 32), __builtin_amdgcn_readlane(nchunk, 48)));

    float m = -INFINITY, l = 0.f;
    f32x2_t acc[VP];
    float tacc[APL];
#pragma unroll
    for (int i = 0; i < VP; ++i) acc[i] = f32x2_t{0.f, 0.f};
#pragma unroll
    for (int a = 0; a < APL; ++a) tacc[a] = 0.f;
    QK<T, VEC> qk;
    float u[APL];
    {
      const uint32_t qw[4] = {cur.q.x, cur.q.y, cur.q.z, cur.q.w};
      qk.set_raw(qw);
    }
    cur.u.get(u);
#pragma unroll
    for (int i = 0; i < APL; ++i) u[i] *= amask;

    for (int ci = 0; ci < nchunk; ++ci) {
      const int rem = deg - ci * U;  // this row's edges left (<= 0: the row is done, its lanes idle)
      if (rem > 0) {
        const int eb = er + ci * U;
        Raw kr[U], vr[U];
        float at[U][APL];
#pragma unroll
        for (int uu = 0; uu < U; ++uu) {
          if (uu < rem) {
            const int sl = *reinterpret_cast<const uint8_t*>(ss + eb + uu);
            kr[uu] = *reinterpret_cast<const Raw*>(ks + sl * 256 + l16 * 16);
            vr[uu] = *reinterpret_cast<const Raw*>(vs + sl * 256 + l16 * 16);
            VecIO<float, APL>::load(reinterpret_cast<const float*>(as + (eb + uu) * (UP * 4)) + a_ld, at[uu]);
          }
        }
        // the online-softmax update of the plain folded kernel, operation for operation
        float s[U];
        float mb = m;
#pragma unroll
        for (int uu = 0; uu < U; ++uu) {
          s[uu] = -INFINITY;
          if (uu < rem) {
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
          if (uu < rem) {
            const float pe = __expf(s[uu] - mb);
            l += pe;
            float vv[VEC];
            unpack<T, VEC>(vr[uu], vv);
#pragma unroll
            for (int i = 0; i < VP; ++i)
              acc[i] = __builtin_elementwise_fma(f32x2_t{pe, pe}, f32x2_t{vv[2 * i], vv[2 * i + 1]}, acc[i]);
#pragma unroll
            for (int a = 0; a < APL; ++a) tacc[a] = fmaf(pe, at[uu][a], tacc[a]);
          }
        }
        m = mb;
      }
    }

    if (node >= 0) {
      const float inv = 1.0f / (l + 1e-16f);
      float o[VEC];
      {
        // (acc * inv) + x_r as TWO roundings, as in the plain kernel (see the scheduled kernel)
#pragma clang fp contract(off)
        RawWords<T, VEC> xw;
        xw.w[0] = cur.xr.x; xw.w[1] = cur.xr.y; xw.w[2] = cur.xr.z; xw.w[3] = cur.xr.w;
        float r[VEC];
        xw.get(r);
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          const float scaled = acc[i >> 1][i & 1] * inv;
          o[i] = p.xr != nullptr ? scaled + r[i] : scaled;
        }
      }
      const int out_row = (int)((uint32_t)node * o_row_bytes);
      uint32_t ow[4];
      pack_words<T, VEC>(o, ow);
      __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{ow[0], ow[1], ow[2], ow[3]}, ors, out_row + lane_off, 0, 2);
      if (a_own) {  // this lane's APL values of t~_i,h
        float t4[APL];
#pragma unroll
        for (int a = 0; a < APL; ++a) t4[a] = tacc[a] * inv;
        uint32_t tw[RawWords<T, APL>::W];
        pack_words<T, APL>(t4, tw);
        buffer_store_words<RawWords<T, APL>::W>(ors, out_row + t_off, 0, tw);
      }
      if (p.lse != nullptr && (gls % LPH) == 0) p.lse[(int64_t)node * (p.C / p.D) + head] = m + __logf(l + 1e-16f);
    }
    cur = nxt;
    ps = nps;
  }
}

static size_t tiles_lds_bytes(int src_cap, int edge_cap, int up) {
  return (size_t)src_cap * 512 + (size_t)edge_cap * (up * 4) + (size_t)edge_cap + 256 + 16;
}

template <typename T, int LPH, int UP>
static int launch_folded_tiles(const EdgeFoldParams& p, const EdgeTileLists& tl, int max_tiles_per_xcd, hipStream_t st) {
  const size_t lds = tiles_lds_bytes(tl.src_cap, tl.edge_cap, UP);  // <= 64 KiB (entry point): no attribute to raise
  auto kernel = gt_edge_attention_folded_tiles_kernel<T, LPH, UP>;
  const unsigned blocks = (unsigned)(8 * (int64_t)max_tiles_per_xcd * p.n_slices);
  hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), lds, st, p, p.attr, tl);
  return ANEMOI_OK;
}

// ---------------------------------------------------------------------------------------------
// Folded path on a graph whose destinations all have exactly THREE in-edges (the mesh -> grid decoder: every grid node
// is fed by its three nearest mesh nodes, reference layers/mapper.py:348-418 on an anemoi-graphs KNN edge set), in RUNS:
// consecutive destinations fed by the same three sources -- neighbouring grid points inside one mesh triangle; mean run
// length 2.04 at N320 -> ico-6, 1.52 with the cap of two -- share ONE gather of the three k / v rows AND one dependent chain
// of index loads: { run_ptr, perm } -> the three source ids -> every load of the run in flight together.  Measured (decoder
// launch of config 3): 0.987 -> 0.889 ms; with runs of up to four (half the gathers, but 150 VGPRs) 0.907: the launch is bound
// by its per-destination latency chains and the q / x_r / out stream, not by the gathered bytes.
//   run_ptr [n_runs + 1]  first destination of every run (runs are capped at EDGE_MAX_RUN = 2 destinations)
//   perm    [n_runs]      6 bits per destination d = 0, 1 of the run; bits 6 d + 2 s .. + 1: position (0 .. 2) inside that
//                         destination's CSR segment of its edge to the s-th source in ASCENDING source order -- the
//                         canonical order the run's rows are gathered in
// Edge e of destination d is CSR slot 3 d + position (uniform degree: rowptr[d] = 3 d, checked by the host).  The three
// terms of a destination are summed in canonical order (the plain kernel: CSR order): same result up to f32 rounding,
// deterministic.  All three scores are in registers at once: exact maximum first, no online rescale.
// ---------------------------------------------------------------------------------------------
template <typename T, int VEC, int LPH, int UP, int MAXRUN>
__global__ __launch_bounds__(256) void gt_edge_attention_folded_runs_kernel(const EdgeFoldParams p,
                                                                        const float* __restrict__ attr_,
                                                                        const int32_t* __restrict__ run_ptr_,
                                                                        const int32_t* __restrict__ col_,
                                                                        const int32_t* __restrict__ perm_,
                                                                        int64_t n_runs) {
  using Raw = typename RawVec<T, VEC>::type;
  constexpr int APL = attrs_per_lane(UP, LPH);
  const int lane = threadIdx.x & 63;
  const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int xcd = blockIdx.x & 7;
  const int wave_in_xcd = (int)(blockIdx.x >> 3) * 4 + wib;
  const int waves_per_xcd = (int)(gridDim.x >> 3) * 4;
  const int slice = wave_in_xcd % p.n_slices;
  const int64_t run_first = wave_in_xcd / p.n_slices;
  const int64_t run_stride = waves_per_xcd / p.n_slices;
  const int64_t r0 = n_runs * xcd / 8, r1 = n_runs * (xcd + 1) / 8;

  const int lanes_total = p.C / VEC;
  const int gl = slice * 64 + lane;
  const bool active = gl < lanes_total;
  const int gls = active ? gl : 0;
  const int c0 = gls * VEC;
  const int head = gls / LPH;
  const int a0 = (gls % LPH) * APL;
  const bool a_own = a0 < UP;
  const int a_ld = a_own ? a0 : 0;
  const float amask = a_own ? 1.f : 0.f;

  const T* qb = static_cast<const T*>(p.q) + c0;
  const T* kb = static_cast<const T*>(p.k) + c0;
  const T* vb = static_cast<const T*>(p.v) + c0;
  const T* ub = static_cast<const T*>(p.u) + head * UP + a_ld;
  const float* ab = attr_ + a_ld;
  typedef __attribute__((ext_vector_type(2))) float f32x2_t;
  constexpr int VP = (VEC + 1) / 2;

  for (int64_t run = r0 + run_first; run < r1; run += run_stride) {
    // one dependent chain per RUN, not per destination: { run_ptr, perm } -> the three source ids -> every load of the run
    // (3 + 3 row gathers, q / u / x_r and the 3 attribute rows of up to four destinations) in flight together
    const int64_t d_begin = run_ptr_[run];
    const int len = (int)(run_ptr_[run + 1] - d_begin);  // 1 .. MAXRUN
    const int pr = perm_[run];                             // 6 bits per destination of the run
    Raw kr[3], vr[3];
#pragma unroll
    for (int sl = 0; sl < 3; ++sl) {
      const int64_t j = col_[3 * d_begin + ((pr >> (2 * sl)) & 3)];
      kr[sl] = *reinterpret_cast<const Raw*>(kb + j * p.ldkv);
      vr[sl] = *reinterpret_cast<const Raw*>(vb + j * p.ldkv);
    }
    QK<T, VEC> qk[MAXRUN];
    float u[MAXRUN][APL], at[MAXRUN][3][APL];
    RawWords<T, VEC> xr_raw[MAXRUN];
#pragma unroll
    for (int d = 0; d < MAXRUN; ++d) {
      if (d < len) {
        const int64_t node = d_begin + d;
        float qf[VEC];
        if (p.stream_hint) load_stream<T, VEC>(qb + node * p.ldq, qf);
        else VecIO<T, VEC>::load(qb + node * p.ldq, qf);
        qk[d].set(qf);
        VecIO<T, APL>::load(ub + node * p.ldu, u[d]);
        if (p.xr != nullptr) xr_raw[d].load(static_cast<const char*>(p.xr) + node * p.ldr * (int64_t)sizeof(T),
                                            (uint32_t)(c0 * (int)sizeof(T)), p.stream_hint != 0);
#pragma unroll
        for (int sl = 0; sl < 3; ++sl)
          VecIO<float, APL>::load(ab + (3 * node + ((pr >> (6 * d + 2 * sl)) & 3)) * UP, at[d][sl]);
      }
    }
#pragma unroll
    for (int d = 0; d < MAXRUN; ++d) {
      if (d < len) {
        const int64_t node = d_begin + d;
        float sc[3];
#pragma unroll
        for (int sl = 0; sl < 3; ++sl) {
          float t = qk[d].dot(kr[sl]);
#pragma unroll
          for (int a = 0; a < APL; ++a) t = fmaf(u[d][a] * amask, at[d][sl][a], t);
          sc[sl] = group_sum<LPH>(t) * p.scale;
        }
        const float m = fmaxf(fmaxf(sc[0], sc[1]), sc[2]);
        float l = 0.f;
        f32x2_t acc[VP];
        float tacc[APL];
#pragma unroll
        for (int i = 0; i < VP; ++i) acc[i] = f32x2_t{0.f, 0.f};
#pragma unroll
        for (int a = 0; a < APL; ++a) tacc[a] = 0.f;
#pragma unroll
        for (int sl = 0; sl < 3; ++sl) {
          const float pe = __expf(sc[sl] - m);
          l += pe;
          float vv[VEC];
          unpack<T, VEC>(vr[sl], vv);
#pragma unroll
          for (int i = 0; i < VP; ++i)
            acc[i] = __builtin_elementwise_fma(f32x2_t{pe, pe}, f32x2_t{vv[2 * i], 2 * i + 1 < VEC ? vv[2 * i + 1] : 0.f}, acc[i]);
#pragma unroll
          for (int a = 0; a < APL; ++a) tacc[a] = fmaf(pe, at[d][sl][a], tacc[a]);
        }
        const float inv = 1.0f / (l + 1e-16f);
        float o[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) o[i] = acc[i >> 1][i & 1] * inv;
        if (p.xr != nullptr) {
          float r[VEC];
          xr_raw[d].get(r);
#pragma unroll
          for (int i = 0; i < VEC; ++i) o[i] += r[i];
        }
        T* on = static_cast<T*>(p.out) + node * p.ldo;
        if (active) {
          if (p.stream_hint) store_stream<T, VEC>(on + c0, o);
          else VecIO<T, VEC>::store(on + c0, o);
        }
        if (active && a_own) {
          float t4[APL];
#pragma unroll
          for (int a = 0; a < APL; ++a) t4[a] = tacc[a] * inv;
          VecIO<T, APL>::store(on + p.C + head * UP + a0, t4);
        }
        if (p.lse != nullptr && active && (gls % LPH) == 0) p.lse[node * (p.C / p.D) + head] = m + __logf(l + 1e-16f);
      }
    }
  }
}

constexpr int EDGE_MAX_RUN = 2;  // destinations per run (measured at N320 -> ico-6, decoder launch: runs of <= 2 at 118 VGPRs /
                                 // four workgroups per CU 0.889 ms, runs of <= 4 at 150 VGPRs / three 0.907, plain kernel 0.987)
template <typename T, int VEC, int LPH, int UP>
static void launch_folded_runs(const EdgeFoldParams& p, const int32_t* run_ptr, const int32_t* perm, int64_t n_runs,
                               hipStream_t st) {
  // (more workgroups than are resident would queue behind them and run as a second, unbalanced phase)
  constexpr int WPB = 4, wgs_per_cu = 4;
  const int64_t units_per_xcd = ((n_runs + 7) / 8) * p.n_slices;
  int64_t bpx = (units_per_xcd + WPB - 1) / WPB;
  if (bpx > 32 * wgs_per_cu) bpx = 32 * wgs_per_cu;
  if (bpx < 1) bpx = 1;
  while ((bpx * WPB) % p.n_slices != 0) ++bpx;
  hipLaunchKernelGGL((gt_edge_attention_folded_runs_kernel<T, VEC, LPH, UP, EDGE_MAX_RUN>), dim3((unsigned)(8 * bpx)),
                     dim3(64 * WPB), 0, st, p, p.attr, run_ptr, p.col, perm, n_runs);
}

// ---------------------------------------------------------------------------------------------
// The uniform-degree-3 decoder in GROUPS (round 5): all destinations fed by the same three sources -- the grid points of
// one mesh triangle, 5.4 on average at N320 -> ico-6, wherever they lie in the grid's own order -- are walked by ONE wave
// behind ONE gather of the three k / v row slices.  The run kernel above only sees the neighbours that happen to be
// consecutive in the grid order (mean 1.5 with its cap of two) and spends 3.75 of its 9.9 loads per destination and slice
// on k / v; here it is 6 / (group length), and the launch is bound by the CU's texture-address unit (81 % busy, PMC).
//   grp_ptr  [n_groups + 1]  group g = entries grp_ptr[g] .. grp_ptr[g + 1] - 1 (1 .. EDGE_MAX_GROUP of them) of
//   grp_dst  [n_dst]         the destinations, every one exactly once, sorted by source triple (runtime.EdgePlan.groups3)
//   grp_perm [n_dst]         6 bits per entry: the CSR position of the edge to the s-th source in ascending source order
// The index chain (group bounds -> first destination -> its three sources; every destination's id and permutation) is
// scalar and resolved one group / one destination ahead; the per-destination arithmetic is the run kernel's, operation for
// operation (canonical source order, exact maximum, no online rescale): bit-identical results.
// ---------------------------------------------------------------------------------------------
constexpr int EDGE_MAX_GROUP = 8;
template <typename T, int VEC, int LPH, int UP>
__global__ __launch_bounds__(256) void gt_edge_attention_folded_groups3_kernel(const EdgeFoldParams p,
                                                                           const float* __restrict__ attr_,
                                                                           const int32_t* __restrict__ col_,
                                                                           const int32_t* __restrict__ grp_ptr_,
                                                                           const int32_t* __restrict__ grp_dst_,
                                                                           const int32_t* __restrict__ grp_perm_,
                                                                           int64_t n_groups) {
  using Raw = typename RawVec<T, VEC>::type;
  constexpr int APL = attrs_per_lane(UP, LPH);
  static_assert(sizeof(Raw) == 16, "one 16-byte row slice per lane");
  typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
  typedef __attribute__((ext_vector_type(2))) float f32x2_t;
  constexpr int VP = (VEC + 1) / 2;
  const int lane = threadIdx.x & 63;
  const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int xcd = blockIdx.x & 7;
  const int wave_in_xcd = (int)(blockIdx.x >> 3) * 4 + wib;
  const int waves_per_xcd = (int)(gridDim.x >> 3) * 4;
  const int slice = wave_in_xcd % p.n_slices;
  const int64_t g_first = wave_in_xcd / p.n_slices;
  const int64_t g_stride = waves_per_xcd / p.n_slices;
  const int64_t g0 = n_groups * xcd / 8, g1 = n_groups * (xcd + 1) / 8;

  const int lanes_total = p.C / VEC;
  const int gl = slice * 64 + lane;
  const bool active = gl < lanes_total;
  const int gls = active ? gl : 0;
  const int c0 = gls * VEC;
  const int head = gls / LPH;
  const int a0 = (gls % LPH) * APL;
  const bool a_own = a0 < UP;
  const int a_ld = a_own ? a0 : 0;
  const float amask = a_own ? 1.f : 0.f;

  const int lane_off = c0 * (int)sizeof(T);
  const int attr_off = a_ld * 4;
  const uint32_t kv_row_bytes = (uint32_t)(p.ldkv * (int64_t)sizeof(T));
  const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.k), 0, -1, 0x00020000);
  const __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.v), 0, -1, 0x00020000);
  const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(attr_), 0, -1, 0x00020000);
  const __amdgpu_buffer_rsrc_t qrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.q), 0, -1, 0x00020000);
  const __amdgpu_buffer_rsrc_t urs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.u), 0, -1, 0x00020000);
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.xr), 0, -1, 0x00020000);
  const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, -1, 0x00020000);
  const uint32_t q_row_bytes = (uint32_t)(p.ldq * (int64_t)sizeof(T)), u_row_bytes = (uint32_t)(p.ldu * (int64_t)sizeof(T));
  const uint32_t xr_row_bytes = (uint32_t)(p.ldr * (int64_t)sizeof(T)), o_row_bytes = (uint32_t)(p.ldo * (int64_t)sizeof(T));
  const int u_off = (head * UP + a_ld) * (int)sizeof(T);
  const int t_off = (p.C + head * UP + a0) * (int)sizeof(T);

  struct DstRegs {  // everything of one destination that travels through vector registers before its arithmetic
    u32x4_t q, xr;
    RawWords<T, APL> u;
    float at[3][APL];
  };
  // node / pm are wave-uniform (scalar loads of grp_dst / grp_perm): row offsets go out as scalar soffsets
  auto issue = [&](DstRegs& r, int node, int pm) __attribute__((always_inline)) {
    r.q = __builtin_amdgcn_raw_buffer_load_b128(qrs, lane_off, (int)((uint32_t)node * q_row_bytes), 2);
    buffer_load_words<RawWords<T, APL>::W>(urs, u_off, (int)((uint32_t)node * u_row_bytes), r.u.w);
    r.xr = u32x4_t{0u, 0u, 0u, 0u};
    if (p.xr != nullptr) r.xr = __builtin_amdgcn_raw_buffer_load_b128(xrs, lane_off, (int)((uint32_t)node * xr_row_bytes), 2);
#pragma unroll
    for (int sl = 0; sl < 3; ++sl)
      buffer_load_f32<APL>(ars, attr_off, (int)((uint32_t)(3 * node + ((pm >> (2 * sl)) & 3)) * (uint32_t)(UP * 4)), r.at[sl]);
  };

  // scalar pipeline, one group ahead: bounds, the first TWO destinations with their permutations, the three source ids
  auto group_head = [&](int64_t g, int& begin, int& len, int (&src)[3], int (&nd)[2], int (&pd)[2]) __attribute__((always_inline)) {
    const int64_t gc = g < g1 ? g : g1 - 1;  // (past the end: a harmless repeat of the last group, never consumed)
    begin = grp_ptr_[gc];
    len = grp_ptr_[gc + 1] - begin;
    const int second = begin + (len > 1 ? 1 : 0);
    nd[0] = grp_dst_[begin];
    pd[0] = grp_perm_[begin];
    nd[1] = grp_dst_[second];
    pd[1] = grp_perm_[second];
#pragma unroll
    for (int sl = 0; sl < 3; ++sl) src[sl] = col_[3 * nd[0] + ((pd[0] >> (2 * sl)) & 3)];
  };
  if (g0 + g_first >= g1) return;
  int nb, nl, nsrc[3], nnd[2], npd[2];
  group_head(g0 + g_first, nb, nl, nsrc, nnd, npd);

  for (int64_t g = g0 + g_first; g < g1; g += g_stride) {
    const int begin = nb, len = nl;
    Raw kr[3], vr[3];
#pragma unroll
    for (int sl = 0; sl < 3; ++sl) {
      const int row = (int)((uint32_t)nsrc[sl] * kv_row_bytes);
      kr[sl] = __builtin_bit_cast(Raw, __builtin_amdgcn_raw_buffer_load_b128(krs, lane_off, row, 0));
      vr[sl] = __builtin_bit_cast(Raw, __builtin_amdgcn_raw_buffer_load_b128(vrs, lane_off, row, 0));
    }
    int node = nnd[0], node_n = nnd[1], pm_n = npd[1];
    DstRegs cur;
    issue(cur, node, npd[0]);
    group_head(g + g_stride, nb, nl, nsrc, nnd, npd);  // the next group's chain, behind this group's first loads

    for (int d = 0; d < len; ++d) {
      // The next destination of the group: its rows are requested before this one's arithmetic (its id and permutation were
      // fetched one iteration earlier; the ones of the destination after it leave now).  The last destination of a group
      // requests nothing: the next group's k / v gathers come first.
      const bool more = d + 1 < len;
      const int third = begin + (d + 2 < len ? d + 2 : d);
      const int node_nn = grp_dst_[third], pm_nn = grp_perm_[third];
      DstRegs nxt = cur;
      if (more) issue(nxt, node_n, pm_n);

      QK<T, VEC> qk;
      {
        const uint32_t qw[4] = {cur.q.x, cur.q.y, cur.q.z, cur.q.w};
        qk.set_raw(qw);
      }
      float u[APL];
      cur.u.get(u);
      float sc[3];
#pragma unroll
      for (int sl = 0; sl < 3; ++sl) {
        float t = qk.dot(kr[sl]);
#pragma unroll
        for (int a = 0; a < APL; ++a) t = fmaf(u[a] * amask, cur.at[sl][a], t);
        sc[sl] = group_sum<LPH>(t) * p.scale;
      }
      const float m = fmaxf(fmaxf(sc[0], sc[1]), sc[2]);
      float l = 0.f;
      f32x2_t acc[VP];
      float tacc[APL];
#pragma unroll
      for (int i = 0; i < VP; ++i) acc[i] = f32x2_t{0.f, 0.f};
#pragma unroll
      for (int a = 0; a < APL; ++a) tacc[a] = 0.f;
#pragma unroll
      for (int sl = 0; sl < 3; ++sl) {
        const float pe = __expf(sc[sl] - m);
        l += pe;
        float vv[VEC];
        unpack<T, VEC>(vr[sl], vv);
#pragma unroll
        for (int i = 0; i < VP; ++i)
          acc[i] = __builtin_elementwise_fma(f32x2_t{pe, pe}, f32x2_t{vv[2 * i], 2 * i + 1 < VEC ? vv[2 * i + 1] : 0.f}, acc[i]);
#pragma unroll
        for (int a = 0; a < APL; ++a) tacc[a] = fmaf(pe, cur.at[sl][a], tacc[a]);
      }
      const float inv = 1.0f / (l + 1e-16f);
      float o[VEC];
      {
        // (acc * inv) + x_r as TWO roundings, like the run kernel (whose + x_r sits behind a branch)
#pragma clang fp contract(off)
        RawWords<T, VEC> xw;
        xw.w[0] = cur.xr.x; xw.w[1] = cur.xr.y; xw.w[2] = cur.xr.z; xw.w[3] = cur.xr.w;
        float r[VEC];
        xw.get(r);
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          const float scaled = acc[i >> 1][i & 1] * inv;
          o[i] = p.xr != nullptr ? scaled + r[i] : scaled;
        }
      }
      const int out_row = (int)((uint32_t)node * o_row_bytes);
      if (active) {
        uint32_t ow[4];
        pack_words<T, VEC>(o, ow);
        __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{ow[0], ow[1], ow[2], ow[3]}, ors, lane_off, out_row, 2);
        __builtin_amdgcn_sched_barrier(0);  // (store-data hazard with an SGPR soffset: see the scheduled kernel)
        asm volatile("s_nop 1" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
      }
      if (active && a_own) {
        float t4[APL];
#pragma unroll
        for (int a = 0; a < APL; ++a) t4[a] = tacc[a] * inv;
        uint32_t tw[RawWords<T, APL>::W];
        pack_words<T, APL>(t4, tw);
        buffer_store_words<RawWords<T, APL>::W>(ors, t_off, out_row, tw);
      }
      if (p.lse != nullptr && active && (gls % LPH) == 0) p.lse[(int64_t)node * (p.C / p.D) + head] = m + __logf(l + 1e-16f);
      cur = nxt;
      node = node_n;
      node_n = node_nn;
      pm_n = pm_nn;
    }
  }
}

template <typename T, int VEC, int LPH, int UP>
static void launch_folded_groups3(const EdgeFoldParams& p, const int32_t* grp_ptr, const int32_t* grp_dst,
                                  const int32_t* grp_perm, int64_t n_groups, hipStream_t st) {
  static const int wgs_env = getenv("ANEMOI_AMD_EDGE_GROUP_WGS") ? atoi(getenv("ANEMOI_AMD_EDGE_GROUP_WGS")) : 0;  // lab switch
  constexpr int WPB = 4;
  // (N320 -> ico-6 decoder launch, 3 / 4 / 5 / 6 workgroups per CU: 0.826 / 0.833 / 0.838 / 0.888 ms -- the launch moves 3.8 GB
  //  of x_r | q | u rows in and out | t rows out, 4.6 TB/s: more waves only contend)
  const int wgs_per_cu = wgs_env > 0 ? wgs_env : 3;
  const int64_t units_per_xcd = ((n_groups + 7) / 8) * p.n_slices;
  int64_t bpx = (units_per_xcd + WPB - 1) / WPB;
  if (bpx > 32 * wgs_per_cu) bpx = 32 * wgs_per_cu;
  if (bpx < 1) bpx = 1;
  while ((bpx * WPB) % p.n_slices != 0) ++bpx;
  hipLaunchKernelGGL((gt_edge_attention_folded_groups3_kernel<T, VEC, LPH, UP>), dim3((unsigned)(8 * bpx)), dim3(64 * WPB), 0,
                     st, p, p.attr, p.col, grp_ptr, grp_dst, grp_perm, n_groups);
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
    static PerDeviceOnce raised;  // per instantiation
    const int raise_dev = raised.pending();
    if (raise_dev >= 0) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024) != hipSuccess)
        return false;
      raised.done(raise_dev);
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

// ---------------------------------------------------------------------------------------------
// GraphTransformerConv as the reference exposes it (layers/conv.py:98-142): the per-edge features e_ij = lin_edge(a_ij)
// arrive as an explicit [E, C] matrix (CSR order) instead of being folded away:
//     s = q_i . (k_j + e_ij) / sqrt(D),  alpha = segment_softmax_i(s) (+1e-16),  out_i = sum_j alpha (v_j + e_ij).
// Same wave / lane mapping and online softmax as the fused kernels; the edge rows are streamed (read once, in order), so
// this costs 2 C bytes per edge more than the folded kernel -- it exists for callers that use the conv on its own.
// DROP: the conv's `dropout` argument in training mode (layers/conv.py:140): alpha keeps its normalisation, the weighted sum
// takes alpha keep / (1 - p) per (edge, head) -- common.hpp::edge_dropout_keep.
// ---------------------------------------------------------------------------------------------
struct EdgeConvParams {
  const void* q;
  const void* k;
  const void* v;
  const void* e;   // [E, lde] CSR order
  const void* xr;  // optional [n_dst, ldr]: added to the result (the blocks' x_r = lin_self(x))
  void* out;
  float* lse;      // optional [n_dst, H] for the backward
  int64_t ldq, ldkv, lde, ldo, ldr;
  int64_t n_dst;
  int C, D, n_slices;
  float scale;
  EdgeDropout drop;
};

template <typename T, int VEC, int LPH, bool DROP>
__global__ __launch_bounds__(256) void gt_conv_kernel(const EdgeConvParams p, const int32_t* __restrict__ rowptr_,
                                                      const int32_t* __restrict__ col_) {
  using Raw = typename RawVec<T, VEC>::type;
  constexpr int U = 2;
  const uint32_t dseed = DROP ? edge_dropout_seed(p.drop) : 0u;
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
  const int c0 = (active ? gl : 0) * VEC;
  const T* qb = static_cast<const T*>(p.q) + c0;
  const T* kb = static_cast<const T*>(p.k) + c0;
  const T* vb = static_cast<const T*>(p.v) + c0;
  const T* eb = static_cast<const T*>(p.e) + c0;
  for (int64_t node = n0 + node_first; node < n1; node += node_stride) {
    const int e_begin = rowptr_[node], e_end = rowptr_[node + 1];
    float qf[VEC], acc[VEC];
    VecIO<T, VEC>::load(qb + node * p.ldq, qf);
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
    float m = -INFINITY, l = 0.f;
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
          float t = 0.f;
#pragma unroll
          for (int i = 0; i < VEC; ++i) t = fmaf(qf[i], kk[i] + ee[i], t);
          const float s = group_sum<LPH>(t) * p.scale;
          const float mn = fmaxf(m, s);
          const float corr = __expf(m - mn), pe = __expf(s - mn);
          l = l * corr + pe;
          const float pv = DROP ? pe * edge_dropout_keep(p.drop, dseed, e + uu, (active ? gl : 0) / LPH) : pe;
#pragma unroll
          for (int i = 0; i < VEC; ++i) acc[i] = fmaf(pv, vv[i] + ee[i], acc[i] * corr);
          m = mn;
        }
      }
    }
    const float inv = 1.0f / (l + 1e-16f);
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] *= inv;
    if (p.xr != nullptr) {
      float r[VEC];
      VecIO<T, VEC>::load(static_cast<const T*>(p.xr) + node * p.ldr + c0, r);
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[i] += r[i];
    }
    if (active) VecIO<T, VEC>::store(static_cast<T*>(p.out) + node * p.ldo + c0, acc);
    if (p.lse != nullptr && active && ((active ? gl : 0) % LPH) == 0)
      p.lse[node * (p.C / p.D) + (active ? gl : 0) / LPH] = m + __logf(l + 1e-16f);
  }
}

template <typename T>
static bool dispatch_conv(const EdgeConvParams& p, const int32_t* rowptr, const int32_t* col, hipStream_t st) {
  constexpr int VEC = 16 / sizeof(T);
  if (p.D % VEC != 0 || p.C % VEC != 0) return false;
  const int64_t units_per_xcd = ((p.n_dst + 7) / 8) * p.n_slices;
  int64_t bpx = (units_per_xcd + 3) / 4;
  if (bpx > 32 * 5) bpx = 32 * 5;
  if (bpx < 1) bpx = 1;
  while ((bpx * 4) % p.n_slices != 0) ++bpx;
  const dim3 grid((unsigned)(8 * bpx)), block(256);
  switch (p.D / VEC) {
#define ANEMOI_CONV(L)                                                                                 \
  case L:                                                                                              \
    if (p.drop.thr15 != 0) hipLaunchKernelGGL((gt_conv_kernel<T, VEC, L, true>), grid, block, 0, st, p, rowptr, col); \
    else hipLaunchKernelGGL((gt_conv_kernel<T, VEC, L, false>), grid, block, 0, st, p, rowptr, col);    \
    return true;
    ANEMOI_CONV(1)
    ANEMOI_CONV(2)
    ANEMOI_CONV(4)
    ANEMOI_CONV(8)
    ANEMOI_CONV(16)
#undef ANEMOI_CONV
    default: return false;
  }
}

extern "C" int anemoi_gt_conv(int dtype, const void* q, int64_t ldq, const void* k, const void* v, int64_t ldkv,
                              const void* edges, int64_t lde, const void* x_r, int64_t ldr, const int32_t* rowptr,
                              const int32_t* col, void* out, int64_t ldo, float* lse, int64_t n_dst, int C, int H,
                              float dropout_p, uint32_t dropout_seed, const void* dropout_seed_dev,
                              anemoi_stream_t stream) {
  ANEMOI_REQUIRE(q && k && v && out && rowptr, ANEMOI_ERR_INVALID, "anemoi_gt_conv: null pointer");
  ANEMOI_REQUIRE(dropout_p >= 0.f && dropout_p <= 1.f, ANEMOI_ERR_INVALID, "anemoi_gt_conv: dropout_p %g outside [0, 1]",
                 (double)dropout_p);
  ANEMOI_REQUIRE(C > 0 && H > 0 && C % H == 0 && n_dst >= 0 && ldq >= C && ldkv >= C && ldo >= C && lde >= C &&
                     (x_r == nullptr || ldr >= C),
                 ANEMOI_ERR_INVALID, "anemoi_gt_conv: bad shape");
  if (n_dst == 0) return ANEMOI_OK;
  ANEMOI_REQUIRE(col != nullptr && edges != nullptr, ANEMOI_ERR_INVALID, "anemoi_gt_conv: null edge arrays");
  const int esz = dtype == ANEMOI_BF16 ? 2 : 4, vec = 16 / esz;
  ANEMOI_REQUIRE((uintptr_t)q % 16 == 0 && (uintptr_t)k % 16 == 0 && (uintptr_t)v % 16 == 0 && (uintptr_t)edges % 16 == 0 &&
                     (uintptr_t)out % 16 == 0 && ldq % vec == 0 && ldkv % vec == 0 && lde % vec == 0 && ldo % vec == 0 &&
                     (x_r == nullptr || ((uintptr_t)x_r % 16 == 0 && ldr % vec == 0)),
                 ANEMOI_ERR_UNSUPPORTED, "anemoi_gt_conv: operands must be 16-byte aligned");
  EdgeConvParams p;
  p.q = q; p.k = k; p.v = v; p.e = edges; p.out = out; p.xr = x_r; p.lse = lse;
  p.ldq = ldq; p.ldkv = ldkv; p.lde = lde; p.ldo = ldo; p.ldr = ldr;
  p.n_dst = n_dst; p.C = C; p.D = C / H;
  p.n_slices = (C + 64 * vec - 1) / (64 * vec);
  p.scale = 1.0f / sqrtf((float)(C / H));
  p.drop = make_edge_dropout(dropout_p, dropout_seed, dropout_seed_dev);
  bool ok = false;
  if (dtype == ANEMOI_F32) ok = dispatch_conv<float>(p, rowptr, col, as_stream(stream));
  else if (dtype == ANEMOI_BF16) ok = dispatch_conv<bf16_t>(p, rowptr, col, as_stream(stream));
  else return fail(ANEMOI_ERR_UNSUPPORTED, "anemoi_gt_conv: dtype %d", dtype);
  ANEMOI_REQUIRE(ok, ANEMOI_ERR_UNSUPPORTED,
                 "anemoi_gt_conv: head size %d is not 1, 2, 4, 8 or 16 lanes of %d channels (the host side zero-pads heads)",
                 C / H, vec);
  return check_launch("anemoi_gt_conv");
}

extern "C" int anemoi_gt_edge_attention_folded(int dtype, const void* q, int64_t ldq, const void* k, const void* v,
                                               int64_t ldkv, const void* x_r, int64_t ldr, const void* u, int64_t ldu,
                                               const float* edge_attr, int up, const int32_t* rowptr,
                                               const int32_t* col, void* out, int64_t ldo, float* lse, int64_t n_dst,
                                               int C, int H, anemoi_stream_t stream) {
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
  p.q = q; p.k = k; p.v = v; p.xr = x_r; p.u = u; p.out = out; p.lse = lse;
  p.ldq = ldq; p.ldkv = ldkv; p.ldr = ldr; p.ldu = ldu; p.ldo = ldo;
  p.attr = edge_attr; p.rowptr = rowptr; p.col = col;
  p.n_dst = n_dst; p.C = C; p.D = C / H;
  p.n_slices = (C + 64 * vec - 1) / (64 * vec);
  p.scale = 1.0f / sqrtf((float)(C / H));
  p.stream_hint = 1;  // nontemporal q / x_r / out (streamed once): -3 % on all three graphs of config 3
  bool ok = false;
  if (dtype == ANEMOI_F32) ok = dispatch_folded<float>(p, up, as_stream(stream));
  else if (dtype == ANEMOI_BF16) ok = dispatch_folded<bf16_t>(p, up, as_stream(stream));
  else return fail(ANEMOI_ERR_UNSUPPORTED, "anemoi_gt_edge_attention_folded: dtype %d", dtype);
  ANEMOI_REQUIRE(ok, ANEMOI_ERR_UNSUPPORTED,
                 "anemoi_gt_edge_attention_folded: unsupported shape (D=%d, UP=%d); use anemoi_gt_edge_attention", C / H,
                 up);
  return check_launch("anemoi_gt_edge_attention_folded");
}

// Launch geometry of anemoi_gt_edge_attention_folded_sched for a destination count / channel width / dtype: wave slots
// per XCD and entries per slot of the schedule the caller builds (layout [8][slots][steps] int32; XCD x owns the
// destinations [n_dst x / 8, n_dst (x + 1) / 8); every slot's list ends with at least three -1).
extern "C" int anemoi_edge_schedule_shape(int dtype, int64_t n_dst, int C, int* slots, int* steps) {
  ANEMOI_REQUIRE(slots && steps && n_dst >= 0 && C > 0 && (dtype == ANEMOI_F32 || dtype == ANEMOI_BF16), ANEMOI_ERR_INVALID,
                 "anemoi_edge_schedule_shape: bad argument");
  const int vec = dtype == ANEMOI_BF16 ? 8 : 4;
  sched_shape(n_dst, (C + 64 * vec - 1) / (64 * vec), slots, steps);
  return ANEMOI_OK;
}

// The uniform-degree-3 decoder in groups of destinations that share their three sources (gt_edge_attention_folded_groups3_kernel
// above): same arguments and result as anemoi_gt_edge_attention_folded_runs with the group lists in place of the run lists
// (rowptr[d] = 3 d is the caller's contract; groups of 1 .. 8 destinations, every destination exactly once).  Shapes the kernel
// does not cover (f32, other head sizes, matrices of 4 GiB and more) take the plain kernel.
extern "C" int anemoi_gt_edge_attention_folded_groups(int dtype, const void* q, int64_t ldq, const void* k, const void* v,
                                                      int64_t ldkv, const void* x_r, int64_t ldr, const void* u, int64_t ldu,
                                                      const float* edge_attr, int up, const int32_t* rowptr,
                                                      const int32_t* col, const int32_t* grp_ptr, const int32_t* grp_dst,
                                                      const int32_t* grp_perm, int64_t n_groups, int64_t n_src, void* out,
                                                      int64_t ldo, float* lse, int64_t n_dst, int C, int H,
                                                      anemoi_stream_t stream) {
  const int64_t gib4 = (int64_t)1 << 32;
  const bool small = n_dst * ldq * 2 < gib4 && n_dst * ldu * 2 < gib4 && n_dst * ldo * 2 < gib4 && n_src * ldkv * 2 < gib4 &&
                     (x_r == nullptr || n_dst * ldr * 2 < gib4) && 3 * n_dst * (int64_t)up * 4 < gib4;
  const bool plain = grp_ptr == nullptr || grp_dst == nullptr || grp_perm == nullptr || n_groups <= 0 || dtype != ANEMOI_BF16 ||
                     H <= 0 || C % H != 0 || !((C / H) == 64 || (C / H) == 32) ||
                     !(up == 4 || up == 8 || up == 12 || up == 16) || n_dst == 0 || !small;
  if (plain)
    return anemoi_gt_edge_attention_folded(dtype, q, ldq, k, v, ldkv, x_r, ldr, u, ldu, edge_attr, up, rowptr, col, out, ldo,
                                           lse, n_dst, C, H, stream);
  const char* who = "anemoi_gt_edge_attention_folded_groups";
  ANEMOI_REQUIRE(q && k && v && u && out && col && edge_attr, ANEMOI_ERR_INVALID, "%s: null pointer", who);
  ANEMOI_REQUIRE(ldq >= C && ldkv >= C && ldu >= (int64_t)H * up && ldo >= (int64_t)C + (int64_t)H * up &&
                     (x_r == nullptr || ldr >= C) && n_groups <= n_dst && EDGE_MAX_GROUP * n_groups >= n_dst && n_src > 0,
                 ANEMOI_ERR_INVALID,
                 "%s: leading dimension too small, or %lld groups cannot cover %lld destinations with groups of 1 .. %d", who,
                 (long long)n_groups, (long long)n_dst, EDGE_MAX_GROUP);
  const bool aligned = ((uintptr_t)q % 16 == 0) && ((uintptr_t)k % 16 == 0) && ((uintptr_t)v % 16 == 0) &&
                       ((uintptr_t)u % 16 == 0) && ((uintptr_t)out % 16 == 0) &&
                       (x_r == nullptr || ((uintptr_t)x_r % 16 == 0 && ldr % 8 == 0)) && ldq % 8 == 0 && ldkv % 8 == 0 &&
                       ldu % 8 == 0 && ldo % 8 == 0 && ((uintptr_t)edge_attr % 16 == 0) && C % 8 == 0;
  ANEMOI_REQUIRE(aligned, ANEMOI_ERR_UNSUPPORTED, "%s: operands must be 16-byte aligned", who);
  EdgeFoldParams p;
  p.q = q; p.k = k; p.v = v; p.xr = x_r; p.u = u; p.out = out; p.lse = lse;
  p.ldq = ldq; p.ldkv = ldkv; p.ldr = ldr; p.ldu = ldu; p.ldo = ldo;
  p.attr = edge_attr; p.rowptr = rowptr; p.col = col;
  p.n_dst = n_dst; p.C = C; p.D = C / H;
  p.n_slices = (C + 511) / 512;
  p.scale = 1.0f / sqrtf((float)(C / H));
  p.stream_hint = 1;
  hipStream_t st = as_stream(stream);
#define ANEMOI_GROUPS_UP(LPH)                                                                                       \
  switch (up) {                                                                                                     \
    case 4: launch_folded_groups3<bf16_t, 8, LPH, 4>(p, grp_ptr, grp_dst, grp_perm, n_groups, st); break;           \
    case 8: launch_folded_groups3<bf16_t, 8, LPH, 8>(p, grp_ptr, grp_dst, grp_perm, n_groups, st); break;           \
    case 12: launch_folded_groups3<bf16_t, 8, LPH, 12>(p, grp_ptr, grp_dst, grp_perm, n_groups, st); break;         \
    default: launch_folded_groups3<bf16_t, 8, LPH, 16>(p, grp_ptr, grp_dst, grp_perm, n_groups, st); break;         \
  }
  if (C / H == 64) {
    ANEMOI_GROUPS_UP(8)
  } else {
    ANEMOI_GROUPS_UP(4)
  }
#undef ANEMOI_GROUPS_UP
  return check_launch(who);
}

// anemoi_gt_edge_attention_folded with a destination schedule (gt_edge_attention_folded_sched_kernel above): the same
// result bit for bit.  ``sched`` int32 [8][slots][steps] as anemoi_edge_schedule_shape prescribes; every destination of
// XCD x's range exactly once in XCD x's lists (the host builds it: runtime.EdgePlan.schedule); bf16 with 32- or 64-channel
// heads -- every other shape takes the plain kernel.
extern "C" int anemoi_gt_edge_attention_folded_sched(int dtype, const void* q, int64_t ldq, const void* k, const void* v,
                                                     int64_t ldkv, const void* x_r, int64_t ldr, const void* u, int64_t ldu,
                                                     const float* edge_attr, int up, const int32_t* rowptr,
                                                     const int32_t* col, const int32_t* sched, int slots, int steps,
                                                     int64_t n_src, int64_t n_edges, void* out, int64_t ldo, float* lse,
                                                     int64_t n_dst, int C, int H, anemoi_stream_t stream) {
  // the kernel addresses rows as 32-bit byte offsets from the matrix bases (buffer loads): every matrix below 4 GiB, the
  // attribute matrix [n_edges, up] f32 included (n_edges = rowptr[n_dst] lives on the device: the caller states it)
  const int64_t lim = (int64_t)1 << 32;
  const bool fits = n_src > 0 && n_src * ldkv * 2 < lim && n_dst * ldq * 2 < lim && n_dst * ldu * 2 < lim && n_dst * ldo * 2 < lim &&
                    (x_r == nullptr || n_dst * ldr * 2 < lim) && n_edges > 0 && n_edges * (int64_t)up * 4 < lim;
  const bool plain = sched == nullptr || dtype != ANEMOI_BF16 || H <= 0 || C % H != 0 || !((C / H) == 64 || (C / H) == 32) ||
                     !(up == 4 || up == 8 || up == 12 || up == 16) || n_dst == 0 || !fits;
  if (plain)
    return anemoi_gt_edge_attention_folded(dtype, q, ldq, k, v, ldkv, x_r, ldr, u, ldu, edge_attr, up, rowptr, col, out, ldo,
                                           lse, n_dst, C, H, stream);
  const char* who = "anemoi_gt_edge_attention_folded_sched";
  ANEMOI_REQUIRE(q && k && v && u && out && col && edge_attr && rowptr, ANEMOI_ERR_INVALID, "%s: null pointer", who);
  ANEMOI_REQUIRE(ldq >= C && ldkv >= C && ldu >= (int64_t)H * up && ldo >= (int64_t)C + (int64_t)H * up &&
                     (x_r == nullptr || ldr >= C),
                 ANEMOI_ERR_INVALID, "%s: leading dimension too small", who);
  int want_slots = 0, want_steps = 0;
  sched_shape(n_dst, (C + 511) / 512, &want_slots, &want_steps);
  ANEMOI_REQUIRE(slots == want_slots && steps >= want_steps, ANEMOI_ERR_INVALID,
                 "%s: schedule of %d slots x %d steps, this launch needs %d x >= %d (anemoi_edge_schedule_shape)", who, slots,
                 steps, want_slots, want_steps);
  const bool aligned = ((uintptr_t)q % 16 == 0) && ((uintptr_t)k % 16 == 0) && ((uintptr_t)v % 16 == 0) &&
                       ((uintptr_t)u % 16 == 0) && ((uintptr_t)out % 16 == 0) &&
                       (x_r == nullptr || ((uintptr_t)x_r % 16 == 0 && ldr % 8 == 0)) && ldq % 8 == 0 && ldkv % 8 == 0 &&
                       ldu % 8 == 0 && ldo % 8 == 0 && ((uintptr_t)edge_attr % 16 == 0) && C % 8 == 0;
  ANEMOI_REQUIRE(aligned, ANEMOI_ERR_UNSUPPORTED, "%s: operands must be 16-byte aligned", who);
  EdgeFoldParams p;
  p.q = q; p.k = k; p.v = v; p.xr = x_r; p.u = u; p.out = out; p.lse = lse;
  p.ldq = ldq; p.ldkv = ldkv; p.ldr = ldr; p.ldu = ldu; p.ldo = ldo;
  p.attr = edge_attr; p.rowptr = rowptr; p.col = col;
  p.n_dst = n_dst; p.C = C; p.D = C / H;
  p.n_slices = (C + 511) / 512;
  p.scale = 1.0f / sqrtf((float)(C / H));
  p.stream_hint = 1;
  hipStream_t st = as_stream(stream);
#define ANEMOI_SCHED_UP(LPH)                                                                                   \
  switch (up) {                                                                                                \
    case 4: launch_folded_sched<bf16_t, 8, LPH, 4>(p, sched, slots, steps, st); break;                         \
    case 8: launch_folded_sched<bf16_t, 8, LPH, 8>(p, sched, slots, steps, st); break;                         \
    case 12: launch_folded_sched<bf16_t, 8, LPH, 12>(p, sched, slots, steps, st); break;                       \
    default: launch_folded_sched<bf16_t, 8, LPH, 16>(p, sched, slots, steps, st); break;                       \
  }
  if (C / H == 64) {
    ANEMOI_SCHED_UP(8)
  } else {
    ANEMOI_SCHED_UP(4)
  }
#undef ANEMOI_SCHED_UP
  return check_launch(who);
}

// anemoi_gt_edge_attention_folded on LDS TILES (gt_edge_attention_folded_tiles_kernel above): the same result bit for bit.
// The tile lists are the host's (anemoi_models_amd/runtime.py::EdgeTiles); bf16, 32- or 64-channel heads, C a multiple of 128,
// every operand matrix below 4 GiB -- every other case (or tile_hdr == NULL) runs the plain kernel.
extern "C" int anemoi_gt_edge_attention_folded_tiles(int dtype, const void* q, int64_t ldq, const void* k, const void* v,
                                                     int64_t ldkv, const void* x_r, int64_t ldr, const void* u, int64_t ldu,
                                                     const float* edge_attr, int up, const int32_t* rowptr,
                                                     const int32_t* col, const int32_t* tile_hdr, const int32_t* tile_dst,
                                                     const int32_t* tile_src, const uint8_t* tile_slot,
                                                     const int32_t* tile_xcd, int max_tiles_per_xcd, int src_cap,
                                                     int edge_cap, int64_t n_src, int64_t n_edges, void* out, int64_t ldo,
                                                     float* lse, int64_t n_dst, int C, int H, anemoi_stream_t stream) {
  const int64_t lim = (int64_t)1 << 31;  // (row offsets + lane offsets are formed as signed 32-bit voffsets)
  const bool fits = n_src > 0 && n_src * ldkv * 2 < lim && n_dst * ldq * 2 < lim && n_dst * ldu * 2 < lim && n_dst * ldo * 2 < lim &&
                    (x_r == nullptr || n_dst * ldr * 2 < lim) && n_edges > 0;
  const bool plain = tile_hdr == nullptr || dtype != ANEMOI_BF16 || H <= 0 || C % H != 0 || !((C / H) == 64 || (C / H) == 32) ||
                     C % 128 != 0 || !(up == 4 || up == 8 || up == 12 || up == 16) || n_dst == 0 || !fits ||
                     max_tiles_per_xcd <= 0;
  if (plain)
    return anemoi_gt_edge_attention_folded(dtype, q, ldq, k, v, ldkv, x_r, ldr, u, ldu, edge_attr, up, rowptr, col, out, ldo,
                                           lse, n_dst, C, H, stream);
  const char* who = "anemoi_gt_edge_attention_folded_tiles";
  ANEMOI_REQUIRE(q && k && v && u && out && edge_attr && tile_dst && tile_src && tile_slot && tile_xcd, ANEMOI_ERR_INVALID,
                 "%s: null pointer", who);
  ANEMOI_REQUIRE(ldq >= C && ldkv >= C && ldu >= (int64_t)H * up && ldo >= (int64_t)C + (int64_t)H * up &&
                     (x_r == nullptr || ldr >= C),
                 ANEMOI_ERR_INVALID, "%s: leading dimension too small", who);
  ANEMOI_REQUIRE(src_cap > 0 && src_cap <= 255 && edge_cap > 0 && edge_cap % 16 == 0 &&
                     tiles_lds_bytes(src_cap, edge_cap, up) <= 64 * 1024,
                 ANEMOI_ERR_INVALID, "%s: tile caps %d sources / %d edges (<= 255, a multiple of 16, LDS <= 64 KiB)", who, src_cap,
                 edge_cap);
  const bool aligned = ((uintptr_t)q % 16 == 0) && ((uintptr_t)k % 16 == 0) && ((uintptr_t)v % 16 == 0) &&
                       ((uintptr_t)u % 16 == 0) && ((uintptr_t)out % 16 == 0) &&
                       (x_r == nullptr || ((uintptr_t)x_r % 16 == 0 && ldr % 8 == 0)) && ldq % 8 == 0 && ldkv % 8 == 0 &&
                       ldu % 8 == 0 && ldo % 8 == 0 && ((uintptr_t)edge_attr % 16 == 0) && ((uintptr_t)tile_slot % 16 == 0);
  ANEMOI_REQUIRE(aligned, ANEMOI_ERR_UNSUPPORTED, "%s: operands must be 16-byte aligned", who);
  EdgeFoldParams p;
  p.q = q; p.k = k; p.v = v; p.xr = x_r; p.u = u; p.out = out; p.lse = lse;
  p.ldq = ldq; p.ldkv = ldkv; p.ldr = ldr; p.ldu = ldu; p.ldo = ldo;
  p.attr = edge_attr; p.rowptr = rowptr; p.col = col;
  p.n_dst = n_dst; p.C = C; p.D = C / H;
  p.n_slices = C / 128;
  p.scale = 1.0f / sqrtf((float)(C / H));
  p.stream_hint = 1;
  EdgeTileLists tl;
  tl.hdr = tile_hdr; tl.dst = tile_dst; tl.src = tile_src; tl.slot = tile_slot; tl.xcd = tile_xcd;
  tl.src_cap = src_cap; tl.edge_cap = edge_cap; tl.n_src = n_src;
  hipStream_t st = as_stream(stream);
  int rc = ANEMOI_OK;
#define ANEMOI_TILES_UP(LPH)                                                                        \
  switch (up) {                                                                                     \
    case 4: rc = launch_folded_tiles<bf16_t, LPH, 4>(p, tl, max_tiles_per_xcd, st); break;          \
    case 8: rc = launch_folded_tiles<bf16_t, LPH, 8>(p, tl, max_tiles_per_xcd, st); break;          \
    case 12: rc = launch_folded_tiles<bf16_t, LPH, 12>(p, tl, max_tiles_per_xcd, st); break;        \
    default: rc = launch_folded_tiles<bf16_t, LPH, 16>(p, tl, max_tiles_per_xcd, st); break;        \
  }
  if (C / H == 64) {
    ANEMOI_TILES_UP(8)
  } else {
    ANEMOI_TILES_UP(4)
  }
#undef ANEMOI_TILES_UP
  if (rc != ANEMOI_OK) return rc;
  return check_launch(who);
}

// The folded edge phase on a uniform-degree-3 graph with its runs of destinations that share their three sources
// (gt_edge_attention_folded_runs_kernel above).  Same arguments and result as anemoi_gt_edge_attention_folded + the run
// list; shapes the run kernel does not cover (f32, other head sizes) take the plain kernel.
extern "C" int anemoi_gt_edge_attention_folded_runs(int dtype, const void* q, int64_t ldq, const void* k, const void* v,
                                                    int64_t ldkv, const void* x_r, int64_t ldr, const void* u, int64_t ldu,
                                                    const float* edge_attr, int up, const int32_t* rowptr,
                                                    const int32_t* col, const int32_t* run_ptr, const int32_t* run_perm,
                                                    int64_t n_runs, void* out, int64_t ldo, float* lse, int64_t n_dst, int C,
                                                    int H, anemoi_stream_t stream) {
  const bool plain = run_ptr == nullptr || run_perm == nullptr || n_runs <= 0 || dtype != ANEMOI_BF16 || H <= 0 || C % H != 0 ||
                     !((C / H) == 64 || (C / H) == 32) || !(up == 4 || up == 8 || up == 12 || up == 16) || n_dst == 0;
  if (plain)
    return anemoi_gt_edge_attention_folded(dtype, q, ldq, k, v, ldkv, x_r, ldr, u, ldu, edge_attr, up, rowptr, col, out, ldo,
                                           lse, n_dst, C, H, stream);
  const char* who = "anemoi_gt_edge_attention_folded_runs";
  ANEMOI_REQUIRE(q && k && v && u && out && col && edge_attr, ANEMOI_ERR_INVALID, "%s: null pointer", who);
  ANEMOI_REQUIRE(ldq >= C && ldkv >= C && ldu >= (int64_t)H * up && ldo >= (int64_t)C + (int64_t)H * up &&
                     (x_r == nullptr || ldr >= C) && n_runs <= n_dst && 2 * n_runs >= n_dst,
                 ANEMOI_ERR_INVALID,
                 "%s: leading dimension too small, or %lld runs cannot cover %lld destinations with runs of 1 or 2 (the kernel "
                 "handles runs of at most two; rowptr[d] = 3 d is the caller's contract)", who, (long long)n_runs,
                 (long long)n_dst);
  const bool aligned = ((uintptr_t)q % 16 == 0) && ((uintptr_t)k % 16 == 0) && ((uintptr_t)v % 16 == 0) &&
                       ((uintptr_t)u % 16 == 0) && ((uintptr_t)out % 16 == 0) &&
                       (x_r == nullptr || ((uintptr_t)x_r % 16 == 0 && ldr % 8 == 0)) && ldq % 8 == 0 && ldkv % 8 == 0 &&
                       ldu % 8 == 0 && ldo % 8 == 0 && ((uintptr_t)edge_attr % 16 == 0) && C % 8 == 0;
  ANEMOI_REQUIRE(aligned, ANEMOI_ERR_UNSUPPORTED, "%s: operands must be 16-byte aligned", who);
  EdgeFoldParams p;
  p.q = q; p.k = k; p.v = v; p.xr = x_r; p.u = u; p.out = out; p.lse = lse;
  p.ldq = ldq; p.ldkv = ldkv; p.ldr = ldr; p.ldu = ldu; p.ldo = ldo;
  p.attr = edge_attr; p.rowptr = rowptr; p.col = col;
  p.n_dst = n_dst; p.C = C; p.D = C / H;
  p.n_slices = (C + 511) / 512;
  p.scale = 1.0f / sqrtf((float)(C / H));
  p.stream_hint = 1;
  hipStream_t st = as_stream(stream);
#define ANEMOI_RUNS_UP(LPH)                                                                                     \
  switch (up) {                                                                                                 \
    case 4: launch_folded_runs<bf16_t, 8, LPH, 4>(p, run_ptr, run_perm, n_runs, st); break;                     \
    case 8: launch_folded_runs<bf16_t, 8, LPH, 8>(p, run_ptr, run_perm, n_runs, st); break;                     \
    case 12: launch_folded_runs<bf16_t, 8, LPH, 12>(p, run_ptr, run_perm, n_runs, st); break;                   \
    default: launch_folded_runs<bf16_t, 8, LPH, 16>(p, run_ptr, run_perm, n_runs, st); break;                   \
  }
  if (C / H == 64) {
    ANEMOI_RUNS_UP(8)
  } else {
    ANEMOI_RUNS_UP(4)
  }
#undef ANEMOI_RUNS_UP
  return check_launch(who);
}
