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
// ---------------------------------------------------------------------------------------------
template <typename T, int VEC, int LPH, int EDP>
__global__ __launch_bounds__(256) void gt_edge_attention_kernel(const EdgeAttnParams p) {
  constexpr int U = 4;  // edges in flight per wave: 2*U independent 16-byte gathers per lane
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

  const int c0 = (slice * 64 + lane) * VEC;
  const bool active = c0 < p.C;
  const int cs = active ? c0 : 0;  // inactive lanes shadow channel 0 (never stored)

  float w[VEC][EDP];
  float bias[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    bias[i] = p.b[cs + i];
#pragma unroll
    for (int a = 0; a < EDP; ++a) w[i][a] = a < p.edge_dim ? p.w[(int64_t)(cs + i) * p.edge_dim + a] : 0.f;
  }

  const T* qb = static_cast<const T*>(p.q) + cs;
  const T* kb = static_cast<const T*>(p.k) + cs;
  const T* vb = static_cast<const T*>(p.v) + cs;

  for (int64_t node = n0 + node_first; node < n1; node += node_stride) {
    float q[VEC];
    VecIO<T, VEC>::load(qb + node * p.ldq, q);
    const int e_begin = p.rowptr[node], e_end = p.rowptr[node + 1];
    float m = -INFINITY, l = 0.f;
    float acc[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] = 0.f;

    for (int e = e_begin; e < e_end; e += U) {
      Raw kr[U], vr[U];
      float ee[U][VEC];
      float s[U];
      // ---- issue all gathers of this batch first
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (e + u < e_end) {
          const int64_t j = p.col[e + u];
          kr[u] = *reinterpret_cast<const Raw*>(kb + j * p.ldkv);
          vr[u] = *reinterpret_cast<const Raw*>(vb + j * p.ldkv);
        }
      }
      // ---- lin_edge (scalar attribute loads, VGPR-resident weights) and scores
      float mb = m;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        s[u] = -INFINITY;
        if (e + u < e_end) {
          const float* at = p.attr + (int64_t)(e + u) * p.ea_ld;
#pragma unroll
          for (int i = 0; i < VEC; ++i) ee[u][i] = bias[i];
#pragma unroll
          for (int a = 0; a < EDP; ++a) {
            const float av = at[a];
#pragma unroll
            for (int i = 0; i < VEC; ++i) ee[u][i] = fmaf(w[i][a], av, ee[u][i]);
          }
          float kk[VEC];
          unpack<T, VEC>(kr[u], kk);
          float part = 0.f;
#pragma unroll
          for (int i = 0; i < VEC; ++i) part = fmaf(q[i], kk[i] + ee[u][i], part);
          s[u] = group_sum<LPH>(part) * p.scale;
          mb = fmaxf(mb, s[u]);
        }
      }
      // ---- online softmax update (one rescale per batch)
      const float corr = __expf(m - mb);
      l *= corr;
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[i] *= corr;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (e + u < e_end) {
          const float pe = __expf(s[u] - mb);
          l += pe;
          float vv[VEC];
          unpack<T, VEC>(vr[u], vv);
#pragma unroll
          for (int i = 0; i < VEC; ++i) acc[i] = fmaf(pe, vv[i] + ee[u][i], acc[i]);
        }
      }
      m = mb;
    }

    const float inv = 1.0f / (l + 1e-16f);
    float o[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) o[i] = acc[i] * inv;
    if (p.xr != nullptr) {
      float r[VEC];
      VecIO<T, VEC>::load(static_cast<const T*>(p.xr) + cs + node * p.ldr, r);
#pragma unroll
      for (int i = 0; i < VEC; ++i) o[i] += r[i];
    }
    if (active) VecIO<T, VEC>::store(static_cast<T*>(p.out) + c0 + node * p.ldo, o);
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
static void launch_fast(const EdgeAttnParams& p, hipStream_t st) {
  // persistent-style grid: 8 XCDs x blocks_per_xcd, 4 waves per block, waves_per_xcd % n_slices == 0
  const int64_t units_per_xcd = ((p.n_dst + 7) / 8) * p.n_slices;
  int64_t bpx = (units_per_xcd + 3) / 4;
  if (bpx > 96) bpx = 96;
  if (bpx < 1) bpx = 1;
  bpx = (bpx + p.n_slices - 1) / p.n_slices * p.n_slices;
  hipLaunchKernelGGL((gt_edge_attention_kernel<T, VEC, LPH, EDP>), dim3((unsigned)(8 * bpx)), dim3(256), 0, st, p);
}

template <typename T, int VEC, int LPH>
static bool dispatch_edp(const EdgeAttnParams& p, hipStream_t st) {
  const int edp = (p.edge_dim + 3) / 4 * 4;
  switch (edp) {
    case 4: launch_fast<T, VEC, LPH, 4>(p, st); return true;
    case 8: launch_fast<T, VEC, LPH, 8>(p, st); return true;
    case 12: launch_fast<T, VEC, LPH, 12>(p, st); return true;
    case 16: launch_fast<T, VEC, LPH, 16>(p, st); return true;
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
