// LAB (round 3; NOT in the product library -- it lost to the gather kernel, see profiles/r03_edge_mfma_lab.md):
// folded GraphTransformer edge phase on the matrix cores: destination tiles of 16 rows, block-sparse.
//
// Same inputs / outputs as anemoi_gt_edge_attention_folded (edge_attention.hip; reference layers/conv.py:98-142 + PyG
// propagate / softmax / scatter, lin_edge folded into u / t), bf16, head size 64.  The gather kernel pulls every k_j / v_j
// row slice through the CU's vector-memory path once per EDGE and spends two thirds of its VALU work on the q . k dot
// products and the alpha v sums; on graphs whose neighbouring destinations share their sources (the multi-scale mesh of
// the processor: 2.4 edges per distinct (16-destination tile, source) pair; the decoder: 3 edges per grid node out of a
// handful of mesh nodes per tile) both are wasted.  Here a tile of <= 16 consecutive destinations and its <= 64 DISTINCT
// source rows (host-built lists, tiler.edge_mfma_tiles) form one small dense problem per head:
//
//   S^T [64 src x 16 dst] = K_tile Q_tile^T + C        v_mfma_f32_16x16x32_bf16, K / Q fragments straight from global
//                                                      memory (each is used once), C the accumulator INPUT: an f32 tile in
//                                                      LDS holding u_i . a_e at the positions of the tile's edges and -1e30
//                                                      everywhere else -- adjacency mask and folded lin_edge term in one
//   alpha   = softmax over the 64 rows of a column     in registers (16 values per lane, two cross-lane steps), + 1e-16
//   O^T [64 d x 16 dst] = V_tile^T alpha               V rows staged once per (tile, head) by LDS-DMA, fragments through
//                                                      the transposing LDS read; alpha as a bf16 hi + lo pair (two MFMAs):
//                                                      the probabilities keep 16 mantissa bits, as f32 ones would
//   t [16 dst x up]     = sum_e alpha_e a_e            per destination over its own edges (alpha read back from LDS)
//
// One wave per (tile, head): a workgroup's four waves take four heads of a tile at a time and share the tile's edge
// attributes / positions / u rows in LDS; nothing else is exchanged between waves.  No atomics, fixed summation order:
// results are reproducible bit for bit, and agree with the gather kernel to bf16 rounding of the outputs (different
// summation order inside the f32 accumulators).
#include <cstdlib>

#include "../../anemoi_models_amd/csrc/common.hpp"

namespace anemoi {

typedef __attribute__((ext_vector_type(8))) __bf16 ebf16x8_t;
typedef __attribute__((ext_vector_type(4))) float ef32x4_t;
typedef __attribute__((ext_vector_type(4))) short es16x4_t;
typedef __attribute__((ext_vector_type(8))) short es16x8_t;

constexpr int ET_D = 16;    // destinations per tile (one MFMA column block)
constexpr int ET_HD = 64;   // head size

struct EdgeMfmaParams {
  const bf16_t* q;
  const bf16_t* k;
  const bf16_t* v;
  const bf16_t* xr;
  const bf16_t* u;
  bf16_t* out;
  float* lse;
  int64_t ldq, ldkv, ldr, ldu, ldo;
  int64_t n_dst, n_src, n_edges;
  const float* attr;        // [E, UP] f32, CSR order
  const int32_t* rowptr;    // [n_dst + 1]
  const int32_t* tiles;     // [n_tiles, 8] = dst0, destinations, offset into tile_src, sources, first edge, edges, 0, 0
  const int32_t* tile_src;  // per tile its distinct source rows
  const uint32_t* posdst;   // [E]: position of the edge in the tile's dense [src][dst] image (low 16) | local dst (high 16)
  int64_t n_tiles;
  int C, H;
  float scale;
};

// Tile capacities of an instantiation: NB row blocks of 16 sources, EC edges
template <int NB, int EC>
struct EtCfg {
  static constexpr int S = NB * 16;
  static constexpr int VBYTES = S * ET_HD * 2;  // V rows of one head
  static constexpr int CBYTES = S * ET_D * 4;   // C / alpha tile of one head
  static constexpr int ABYTES = EC * 4;         // alpha of the tile's edges
  static constexpr int WAVE = VBYTES + CBYTES + ABYTES;
  // per workgroup TWO sets of the tile's shared edge data (the next tile's set is filled by LDS-DMA while the current one
  // is in use): attributes, positions, u rows, row pointers
  static constexpr int shared_bytes(int up, int H) { return EC * up * 4 + EC * 4 + ET_D * H * up * 2 + 128; }
  static constexpr int lds_bytes(int up, int H) { return 4 * WAVE + 2 * shared_bytes(up, H); }
};

__device__ __forceinline__ float bf16_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

typedef __attribute__((ext_vector_type(4))) unsigned eu32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned eu32x2_t;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t et_rsrc(const void* ptr, int64_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(ptr), 0, (int)(bytes < 0x7fffffff ? bytes : 0x7fffffff),
                                           0x00020000);
}
// plain pointers for the DMA of the shared edge data (arbitrary sizes): 16 or 4 bytes per lane, lane-linear in LDS
template <int BYTES>
__device__ __forceinline__ void et_dma(const void* g, void* l) {
  if constexpr (BYTES == 16)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
  else
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 4, 0, 0);
}

#ifndef ET_PROF
#define ET_PROF 0  // -DET_PROF=1: per-phase shader clocks of one wave (prof.py)
#endif
#if ET_PROF
__device__ unsigned long long et_prof[16];
#define ET_T(i) do { if (prof_on) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)"); pt[i] += t_ - tlast; tlast = t_; } } while (0)
#else
#define ET_T(i)
#endif
// Per-tile state of a wave: the tile's scalars and this lane's byte offsets into the operand buffers
template <int NB>
struct EtTile {
  int dst0, nd, s0, ns, e0, ne;
  int koff[NB];  // K row of fragment row (block b, row fr) + this lane's 16-byte column piece
  int qoff;      // this lane's destination row of q (clamped) + column piece; x_r / out rows follow from drow
  int drow;
  bool dvalid;
};
template <int NB>
struct EtOps {
  eu32x4_t kf[NB][2], qf[2];
  eu32x2_t xr4[4];
};

// All global operands go through buffer descriptors: a load is ONE instruction (per-lane 32-bit offset from the tile
// setup, the head as scalar offset, everything else an immediate) -- with 64-bit pointer arithmetic per access the first
// version of this kernel issued ~1200 instructions per (tile, head), three times its useful work.
template <int UP, int NB, int EC>
__global__ __launch_bounds__(256, 2) void gt_edge_attention_mfma_kernel(const EdgeMfmaParams p) {
  static_assert(UP % 4 == 0 && UP >= 4 && UP <= 16 && (NB == 2 || NB == 4), "folded edge width / row blocks");
  using Cfg = EtCfg<NB, EC>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  const int H = p.H, HQ = H >> 2;  // heads per wave and tile (H % 4 == 0: launcher)
  char* vbuf = smem + wid * Cfg::VBYTES;
  float* cbuf = reinterpret_cast<float*>(smem + 4 * Cfg::VBYTES + wid * Cfg::CBYTES);
  float* alpha_s = reinterpret_cast<float*>(smem + 4 * (Cfg::VBYTES + Cfg::CBYTES) + wid * Cfg::ABYTES);
  char* shared0 = smem + 4 * Cfg::WAVE;
  const int shared_bytes = Cfg::shared_bytes(UP, H);
  const int u_off = EC * UP * 4 + EC * 4, rp_off = u_off + ET_D * H * UP * 2;
  const float c2 = p.scale * 1.44269504088896340736f;  // scores are kept unscaled; exp2((s - m) * scale * log2 e)
  const int u_pieces = ET_D * (H * UP / 8);             // 16-byte pieces of the tile's u rows
  const int ldq2 = (int)p.ldq * 2, ldkv2 = (int)p.ldkv * 2, ldr2 = (int)p.ldr * 2, ldo2 = (int)p.ldo * 2;
  __amdgpu_buffer_rsrc_t qrs = et_rsrc(p.q, p.n_dst * p.ldq * 2), krs = et_rsrc(p.k, p.n_src * p.ldkv * 2),
                               vrs = et_rsrc(p.v, p.n_src * p.ldkv * 2),
                               xrs = et_rsrc(p.xr != nullptr ? p.xr : p.q, p.xr != nullptr ? p.n_dst * p.ldr * 2 : 0),
                               ors = et_rsrc(p.out, p.n_dst * p.ldo * 2);
  // transposing-read base of this lane: row 4 fq + mr of a 32-row pair, 8-byte piece cq of the 32-byte column group; the
  // group's XOR (row >> 1) & 3 does not depend on the pair or on the +16 of the pair's second block
  const int mr = fr >> 2, cq = fr & 3, vrow = 4 * fq + mr, vswz = (vrow >> 1) & 3;
  int vtr[4];
#pragma unroll
  for (int mb = 0; mb < 4; ++mb) vtr[mb] = vrow * 128 + ((mb ^ vswz) << 5) + cq * 8;
  const int p16 = lane & 7, srow = lane >> 3;  // V staging: 16-byte slot and row inside an 8-row piece

  // ---- everything a (tile, head) iteration reads from global memory is requested ONE ITERATION AHEAD (the next head of
  //      the tile, or the first head of the workgroup's next tile); the tile's descriptor and K row list one TILE ahead
  //      in registers, its shared edge data one tile ahead by LDS-DMA into the second LDS set.
  // a tile's descriptor: eight words, wave-uniform (scalar loads); first used a whole tile after the request
  auto tile_scalars = [&](int64_t tile, EtTile<NB>& T) {
    const int32_t* ti = p.tiles + tile * 8;
    T.dst0 = ti[0], T.nd = ti[1], T.s0 = ti[2], T.ns = ti[3], T.e0 = ti[4], T.ne = ti[5];
  };
  // ... and this lane's rows of it (needs the scalars; first used three iterations after the request)
  auto tile_rows = [&](EtTile<NB>& T) {
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int r = b * 16 + fr;
      T.koff[b] = p.tile_src[T.s0 + (r < T.ns ? r : T.ns - 1)] * ldkv2 + fq * 16;  // rows behind the list: a valid row (masked)
    }
    T.drow = T.dst0 + (fr < T.nd ? fr : T.nd - 1);  // clamped: padding columns are computed on a valid row, not stored
    T.qoff = T.drow * ldq2 + fq * 16;
    T.dvalid = fr < T.nd;
  };
  // the tile's shared edge data -> LDS set `set` (lane-linear DMA images; lanes behind the tile's own pieces are masked)
  auto shared_dma = [&](const EtTile<NB>& T, int set) {
    char* base = shared0 + set * shared_bytes;
    const int a_pieces = T.ne * (UP / 4);
    const char* asrc = reinterpret_cast<const char*>(p.attr + (int64_t)T.e0 * UP);
    for (int i0 = wid * 64; i0 < a_pieces; i0 += 256)  // (wave-uniform bounds)
      if (i0 + lane < a_pieces) et_dma<16>(asrc + (int64_t)(i0 + lane) * 16, base + i0 * 16);
    for (int i0 = wid * 64; i0 < T.ne; i0 += 256)
      if (i0 + lane < T.ne) et_dma<4>(p.posdst + T.e0 + i0 + lane, base + EC * UP * 4 + i0 * 4);
    const int per_row = H * UP / 8;
    for (int i0 = wid * 64; i0 < u_pieces; i0 += 256) {
      const int i = i0 + lane;
      if (i < u_pieces) {
        const int r = i / per_row, c = i - r * per_row;
        const int rr = r < T.nd ? r : T.nd - 1;
        et_dma<16>(p.u + (int64_t)(T.dst0 + rr) * p.ldu + c * 8, base + u_off + i0 * 16);
      }
    }
    if (wid == 3 && lane <= ET_D)  // raw row pointers of the tile's destinations (e0 is subtracted where they are used)
      et_dma<4>(p.rowptr + T.dst0 + (lane < T.nd ? lane : T.nd), base + rp_off);
  };
  auto ops_fetch = [&](const EtTile<NB>& T, int h, EtOps<NB>& o) {
    const int hs = h * (ET_HD * 2);
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) o.kf[b][ks] = __builtin_amdgcn_raw_buffer_load_b128(krs, T.koff[b] + ks * 64, hs, 0);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) o.qf[ks] = __builtin_amdgcn_raw_buffer_load_b128(qrs, T.qoff + ks * 64, hs, 0);
    const int xoff = T.drow * ldr2 + fq * 8;
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) o.xr4[mb] = __builtin_amdgcn_raw_buffer_load_b64(xrs, xoff + mb * 32, hs, 0);  // (no x_r: 0)
  };
  // V rows of a head -> this wave's LDS buffer (lane-linear image: 8 rows x 128 B per instruction; the 32-byte column
  // groups of row r are XOR-ed with (r >> 1) & 3 on the SOURCE side: the eight rows a half-wave's transposing read
  // touches then lie in eight different bank groups).  The staging lane of row r takes the row's offset from the lane that
  // holds it as K row (block r >> 4, lane r & 15).
  auto v_offsets = [&](const EtTile<NB>& T, int (&voff)[2 * NB]) {
#pragma unroll
    for (int i = 0; i < 2 * NB; ++i) {
      const int r = i * 8 + srow;
      const int slot = (((p16 >> 1) ^ ((r >> 1) & 3)) << 1) | (p16 & 1);
      voff[i] = __shfl(T.koff[i >> 1], (i & 1) * 8 + srow, 64) + slot * 16;  // (lanes fq = 0: koff = row offset + 0)
    }
  };
  // (the (int) casts matter: with a type-DEPENDENT offset argument -- voff has a dependent bound -- clang's host pass
  // silently drops the kernel's stub and the library fails to load with an undefined kernel symbol)
#define ET_V_STAGE(head)                                                                                                  \
  _Pragma("unroll") for (int i_ = 0; i_ < 2 * NB; ++i_) __builtin_amdgcn_raw_ptr_buffer_load_lds(                         \
      vrs, (__attribute__((address_space(3))) void*)(vbuf + i_ * 1024), 16, (int)voff[i_], (int)((head) * (ET_HD * 2)), 0, 0)

  // XCD-contiguous tile ranges (neighbouring tiles share source rows: one L2 serves them)
  const int64_t xcd = blockIdx.x & 7, bix = blockIdx.x >> 3, bpx = gridDim.x >> 3;
  const int64_t t0 = p.n_tiles * xcd / 8, t1 = p.n_tiles * (xcd + 1) / 8;
  int64_t tile = t0 + bix;
  if (tile >= t1) return;
  EtTile<NB> cur, nxt, nx2;  // (of nx2 only the scalars are live)
  EtOps<NB> oc, on;
  int voff[2 * NB];
  int set = 0;
  tile_scalars(tile, cur);
  tile_rows(cur);
  shared_dma(cur, 0);
  ops_fetch(cur, wid, oc);
  v_offsets(cur, voff);
  ET_V_STAGE(wid);
  bool more = tile + bpx < t1;
  if (more) {
    tile_scalars(tile + bpx, nxt);
    tile_rows(nxt);
    shared_dma(nxt, 1);
    if (tile + 2 * bpx < t1) tile_scalars(tile + 2 * bpx, nx2);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's share of the first tile's data has landed ...
  __syncthreads();                                    // ... everyone's has

#if ET_PROF
  const bool prof_on = blockIdx.x == 8 && wid == 0;
  unsigned long long pt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)");
  int iters = 0;
#endif
  for (;;) {
    const float* attr_s = reinterpret_cast<const float*>(shared0 + set * shared_bytes);
    const uint32_t* pd_s = reinterpret_cast<const uint32_t*>(attr_s + EC * UP);
    const bf16_t* u_s = reinterpret_cast<const bf16_t*>(shared0 + set * shared_bytes + u_off);
    const int* rp_s = reinterpret_cast<const int*>(shared0 + set * shared_bytes + rp_off);
    const int ooff = cur.drow * ldo2 + fq * 8;
    for (int j = 0; j < HQ; ++j) {
      const int h = wid + 4 * j;
      const bool last_head = j + 1 == HQ;
      // ---- next iteration's operands: requested now, consumed one iteration later
      if (!last_head) ops_fetch(cur, h + 4, on);
      else if (more) ops_fetch(nxt, wid, on);

      ET_T(0);
      // ---- C tile: -1e30 everywhere, u_i . a_e at the edges' positions (edge e = lane + 64 s: its position/destination
      //      word stays in a register for the t pass below)
      uint32_t pdr[(EC + 63) / 64];
      {
        const float4 neg = make_float4(-1e30f, -1e30f, -1e30f, -1e30f);
#pragma unroll
        for (int i = 0; i < NB; ++i) reinterpret_cast<float4*>(cbuf)[i * 64 + lane] = neg;
#pragma unroll
        for (int s_ = 0; s_ < (EC + 63) / 64; ++s_) {
          const int e = lane + 64 * s_;
          if (e < cur.ne) {
            const uint32_t pd = pd_s[e];
            pdr[s_] = pd;
            const int dl = (int)(pd >> 16);
            const uint32_t* up_ = reinterpret_cast<const uint32_t*>(u_s + (dl * H + h) * UP);
            const float* ap = attr_s + e * UP;
            float t = 0.f;
#pragma unroll
            for (int a = 0; a < UP; a += 2) {
              const uint32_t w = up_[a >> 1];
              t = fmaf(bf16_lo(w), ap[a], t);
              t = fmaf(bf16_hi(w), ap[a + 1], t);
            }
            cbuf[pd & 0xffffu] = t;
          }
        }
      }
      asm volatile("" ::: "memory");  // (one wave: LDS operations complete in program order)

      ET_T(1);
      // ---- S^T = K Q^T + C
      ef32x4_t sacc[NB];
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        sacc[b] = *reinterpret_cast<const ef32x4_t*>(cbuf + ((b * 4 + fq) * 16 + fr) * 4);
        sacc[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(ebf16x8_t, oc.kf[b][0]),
                                                          __builtin_bit_cast(ebf16x8_t, oc.qf[0]), sacc[b], 0, 0, 0);
        sacc[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(ebf16x8_t, oc.kf[b][1]),
                                                          __builtin_bit_cast(ebf16x8_t, oc.qf[1]), sacc[b], 0, 0, 0);
      }
      ET_T(2);
      // ---- softmax over the column (this lane: 4 NB of its 16 NB rows; lanes fr, fr + 16, fr + 32, fr + 48 share a column)
      float m = sacc[0][0];
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) m = fmaxf(m, sacc[b][r]);
      m = fmaxf(m, __shfl_xor(m, 16, 64));
      m = fmaxf(m, __shfl_xor(m, 32, 64));
      const bool any = m > -1e29f;  // a destination without edges: alpha = 0, out = x_r, t = 0
      const float mc = any ? m * c2 : 0.f;
      float l = 0.f;
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pe = __builtin_amdgcn_exp2f(fmaf(sacc[b][r], c2, -mc));  // (no edges: exp2(-1.8e29) = 0 everywhere)
          sacc[b][r] = pe;
          l += pe;
        }
      l += __shfl_xor(l, 16, 64);
      l += __shfl_xor(l, 32, 64);
      const float inv = 1.0f / (l + 1e-16f);
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        sacc[b] *= inv;
        *reinterpret_cast<ef32x4_t*>(cbuf + ((b * 4 + fq) * 16 + fr) * 4) = sacc[b];  // alpha, for the edges' own lanes
      }
      if (p.lse != nullptr && fq == 0 && cur.dvalid)
        p.lse[(int64_t)cur.drow * H + h] = any ? m * p.scale + __logf(l + 1e-16f) : -INFINITY;
      asm volatile("" ::: "memory");
#pragma unroll
      for (int s_ = 0; s_ < (EC + 63) / 64; ++s_) {
        const int e = lane + 64 * s_;
        if (e < cur.ne) alpha_s[e] = cbuf[pdr[s_] & 0xffffu];
      }
      ET_T(3);
      // ---- alpha as bf16 hi + lo, B operand of O^T += V^T alpha: K index 8 fq + j <-> source 16 b0 + 4 fq + j (j < 4),
      //      16 b1 + 4 fq + j - 4 (j >= 4) for the block pair (b0, b1) = (2 kp, 2 kp + 1)
      ebf16x8_t ahi[NB / 2], alo[NB / 2];
#pragma unroll
      for (int kp = 0; kp < NB / 2; ++kp) {
        uint32_t wh[4], wl[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const float a0 = sacc[2 * kp + (jj >> 1)][2 * (jj & 1)], a1 = sacc[2 * kp + (jj >> 1)][2 * (jj & 1) + 1];
          wh[jj] = pack_bf16x2(a0, a1);
          wl[jj] = pack_bf16x2(a0 - bf16_lo(wh[jj]), a1 - bf16_hi(wh[jj]));
        }
        ahi[kp] = __builtin_bit_cast(ebf16x8_t, *reinterpret_cast<uint4*>(wh));
        alo[kp] = __builtin_bit_cast(ebf16x8_t, *reinterpret_cast<uint4*>(wl));
      }
      // ---- O^T = V^T alpha.  The V rows (requested one iteration ago) have to have LANDED: the compiler does not count
      //      an LDS-DMA as a writer of the LDS it reads below, hence the explicit wait (it also completes the operand
      //      prefetch issued at the top of this iteration -- a C-tile build, the score MFMAs and a softmax ago -- and, in
      //      a tile's first iteration, the next tile's shared-data DMA)
      ET_T(4);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      ET_T(5);
      ef32x4_t oacc[4];
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) {
        oacc[mb] = ef32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kp = 0; kp < NB / 2; ++kp) {
          const es16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) es16x4_t*)(vbuf + vtr[mb] + kp * 4096));
          const es16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) es16x4_t*)(vbuf + vtr[mb] + kp * 4096 + 2048));
          const ebf16x8_t vf = __builtin_bit_cast(ebf16x8_t, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
          oacc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, ahi[kp], oacc[mb], 0, 0, 0);
          oacc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, alo[kp], oacc[mb], 0, 0, 0);
        }
      }
      // ---- the V buffer is free once the transposing reads above have returned: request the next iteration's V rows
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (!last_head) {
        ET_V_STAGE(h + 4);
      } else if (more) {
        v_offsets(nxt, voff);
        ET_V_STAGE(wid);
      }
      // ---- out[dst, head, 16 mb + 4 fq + r] = O + x_r
      {
        const int oo = cur.dvalid ? ooff : 0x7f000000;  // padding columns: beyond the descriptor, the store is dropped
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
          const float o0 = oacc[mb][0] + bf16_lo(oc.xr4[mb].x), o1 = oacc[mb][1] + bf16_hi(oc.xr4[mb].x);
          const float o2 = oacc[mb][2] + bf16_lo(oc.xr4[mb].y), o3 = oacc[mb][3] + bf16_hi(oc.xr4[mb].y);
          __builtin_amdgcn_raw_buffer_store_b64(eu32x2_t{pack_bf16x2(o0, o1), pack_bf16x2(o2, o3)}, ors, oo + mb * 32,
                                                h * (ET_HD * 2), 0);
        }
      }
      ET_T(6);
      // ---- t[dst, head, :] = sum over the destination's own edges of alpha_e a_e.  Lane 4 d + q takes every fourth edge
      //      of destination d (all UP columns), the quad is summed by DPP and lane q stores columns 4 q .. 4 q + 3: the
      //      tile's highest in-degree (the multi-scale mesh has an 18+ edge node in almost every tile) costs a quarter of
      //      the dependent LDS round trips of a lane-per-destination loop.
      {
        const int td = lane >> 2, tq = lane & 3;
        const int eb = rp_s[td] - cur.e0, ee = rp_s[td + 1] - cur.e0;
        float acc[UP];
#pragma unroll
        for (int a = 0; a < UP; ++a) acc[a] = 0.f;
        for (int e = eb + tq; e < ee; e += 8) {  // two edges per trip (the second one masked behind the row's end)
          const int e2 = e + 4 < ee ? e + 4 : e;
          const float al = alpha_s[e], al2 = e + 4 < ee ? alpha_s[e2] : 0.f;
#pragma unroll
          for (int a = 0; a < UP; a += 4) {
            const float4 a4 = *reinterpret_cast<const float4*>(attr_s + e * UP + a);
            const float4 b4 = *reinterpret_cast<const float4*>(attr_s + e2 * UP + a);
            acc[a] = fmaf(al2, b4.x, fmaf(al, a4.x, acc[a]));
            acc[a + 1] = fmaf(al2, b4.y, fmaf(al, a4.y, acc[a + 1]));
            acc[a + 2] = fmaf(al2, b4.z, fmaf(al, a4.z, acc[a + 2]));
            acc[a + 3] = fmaf(al2, b4.w, fmaf(al, a4.w, acc[a + 3]));
          }
        }
#pragma unroll
        for (int a = 0; a < UP; ++a) {
          acc[a] += dpp_f32<0xB1>(acc[a]);  // quad_perm [1,0,3,2]
          acc[a] += dpp_f32<0x4E>(acc[a]);  // quad_perm [2,3,0,1]
        }
        const int toff = td < cur.nd ? (cur.dst0 + td) * ldo2 + p.C * 2 + tq * 8 : 0x7f000000;
#pragma unroll
        for (int g = 0; g < UP / 4; ++g)
          if (tq == g)
            __builtin_amdgcn_raw_buffer_store_b64(
                eu32x2_t{pack_bf16x2(acc[4 * g], acc[4 * g + 1]), pack_bf16x2(acc[4 * g + 2], acc[4 * g + 3])}, ors, toff,
                h * (UP * 2), 0);
      }
      asm volatile("" ::: "memory");
      oc = on;
#if ET_PROF
      ET_T(7);
      ++iters;
#endif
    }
    if (!more) break;
    // ---- tile boundary.  The next tile's shared data was requested a whole tile ago (this wave's share is waited for
    //      by the vmcnt(0) of every iteration); the barrier publishes everyone's share and retires the current set, which
    //      then receives the shared data of the tile after next.
    __syncthreads();
    cur = nxt;
    set ^= 1;
    tile += bpx;
    more = tile + bpx < t1;
    if (more) {  // (nothing here waits: the scalars were requested a tile ago, the rows are used three iterations on)
      nxt.dst0 = nx2.dst0, nxt.nd = nx2.nd, nxt.s0 = nx2.s0, nxt.ns = nx2.ns, nxt.e0 = nx2.e0, nxt.ne = nx2.ne;
      tile_rows(nxt);
      shared_dma(nxt, set ^ 1);
      if (tile + 2 * bpx < t1) tile_scalars(tile + 2 * bpx, nx2);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no LDS-DMA may outlive the workgroup
#if ET_PROF
  if (prof_on && lane == 0) { for (int i = 0; i < 8; ++i) et_prof[i] = pt[i]; et_prof[8] = iters; }
#endif
}

#undef ET_V_STAGE

template <int UP, int NB, int EC>
static int launch_mfma(const EdgeMfmaParams& p, hipStream_t st) {
  using Cfg = EtCfg<NB, EC>;
  int lds = Cfg::lds_bytes(UP, p.H);
#if ET_PROF
  if (getenv("ET_ONE_WG")) lds = 100 * 1024;
#endif
  if (lds > 100 * 1024)
    return fail(ANEMOI_ERR_UNSUPPORTED, "lab_gt_edge_attention_tiles: %d heads need %d bytes of LDS", p.H, lds);
  auto kern = gt_edge_attention_mfma_kernel<UP, NB, EC>;
  static PerDeviceOnce raised;  // per instantiation
  const int raise_dev = raised.pending();
  if (raise_dev >= 0) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024) !=
        hipSuccess)
      return fail(ANEMOI_ERR_LAUNCH, "lab_gt_edge_attention_tiles: cannot raise the dynamic LDS limit");
    raised.done(raise_dev);
  }
  const int wg_per_cu = (160 * 1024) / lds < 2 ? 1 : ((160 * 1024) / lds > 3 ? 3 : (160 * 1024) / lds);
  int64_t bpx = (p.n_tiles + 7) / 8;                  // tiles of the largest XCD range
  if (bpx > 32 * wg_per_cu) bpx = 32 * wg_per_cu;     // resident workgroups per CU x 32 CUs per XCD
  if (bpx < 1) bpx = 1;
  hipLaunchKernelGGL(kern, dim3((unsigned)(8 * bpx)), dim3(256), lds, st, p);
  return check_launch("lab_gt_edge_attention_tiles");
}

template <int NB, int EC>
static int dispatch_mfma(const EdgeMfmaParams& p, int up, hipStream_t st) {
  switch (up) {
    case 4: return launch_mfma<4, NB, EC>(p, st);
    case 8: return launch_mfma<8, NB, EC>(p, st);
    case 12: return launch_mfma<12, NB, EC>(p, st);
    case 16: return launch_mfma<16, NB, EC>(p, st);
    default: return fail(ANEMOI_ERR_UNSUPPORTED, "lab_gt_edge_attention_tiles: folded edge width %d", up);
  }
}

}  // namespace anemoi

using namespace anemoi;

extern "C" const char* lab_last_error() { return anemoi::err_buf(); }
#if ET_PROF
extern "C" void lab_edge_prof(unsigned long long* out) { hipMemcpyFromSymbol(out, HIP_SYMBOL(anemoi::et_prof), 16 * 8); }
#endif
extern "C" int lab_gt_edge_attention_tiles(const void* q, int64_t ldq, const void* k, const void* v, int64_t ldkv,
                                              const void* x_r, int64_t ldr, const void* u, int64_t ldu,
                                              const float* edge_attr, int up, const int32_t* rowptr, const int32_t* tiles,
                                              const int32_t* tile_src, const uint32_t* posdst, int64_t n_tiles,
                                              int tile_src_cap, int tile_edge_cap, void* out, int64_t ldo, float* lse,
                                              int64_t n_dst, int64_t n_src, int64_t n_edges, int C, int H,
                                              anemoi_stream_t stream) {
  ANEMOI_REQUIRE(q && k && v && u && edge_attr && rowptr && tiles && tile_src && posdst && out, ANEMOI_ERR_INVALID,
                 "lab_gt_edge_attention_tiles: null pointer");
  ANEMOI_REQUIRE(C > 0 && H > 0 && C == H * ET_HD && H % 4 == 0 && (int64_t)H * up <= 256, ANEMOI_ERR_UNSUPPORTED,
                 "lab_gt_edge_attention_tiles: heads of %d channels, H a multiple of 4, H * up <= 256 (C = %d, H = %d)",
                 ET_HD, C, H);
  ANEMOI_REQUIRE(n_dst >= 0 && n_src > 0 && n_tiles >= 0 && ldq >= C && ldkv >= C && ldu >= (int64_t)H * up &&
                     ldo >= C + (int64_t)H * up && (x_r == nullptr || ldr >= C),
                 ANEMOI_ERR_INVALID, "lab_gt_edge_attention_tiles: bad shape / leading dimension");
  ANEMOI_REQUIRE(((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)u | (uintptr_t)out | (uintptr_t)edge_attr |
                  (uintptr_t)(x_r ? x_r : q)) % 16 == 0 &&
                     ldq % 8 == 0 && ldkv % 8 == 0 && ldu % 8 == 0 && ldo % 4 == 0 && (x_r == nullptr || ldr % 4 == 0),
                 ANEMOI_ERR_UNSUPPORTED, "lab_gt_edge_attention_tiles: operands must be 16-byte aligned");
  const int64_t lim = (int64_t)1 << 31;  // 32-bit byte offsets into every operand
  ANEMOI_REQUIRE(n_dst * ldq * 2 < lim && n_src * ldkv * 2 < lim && n_dst * ldo * 2 < lim && (x_r == nullptr || n_dst * ldr * 2 < lim),
                 ANEMOI_ERR_UNSUPPORTED, "lab_gt_edge_attention_tiles: an operand exceeds 2 GiB");
  if (n_tiles == 0) return ANEMOI_OK;
  EdgeMfmaParams p;
  p.q = static_cast<const bf16_t*>(q);
  p.k = static_cast<const bf16_t*>(k);
  p.v = static_cast<const bf16_t*>(v);
  p.xr = static_cast<const bf16_t*>(x_r);
  p.u = static_cast<const bf16_t*>(u);
  p.out = static_cast<bf16_t*>(out);
  p.lse = lse;
  p.ldq = ldq; p.ldkv = ldkv; p.ldr = ldr; p.ldu = ldu; p.ldo = ldo;
  p.n_dst = n_dst; p.n_src = n_src; p.n_edges = n_edges;
  p.attr = edge_attr; p.rowptr = rowptr; p.tiles = tiles; p.tile_src = tile_src; p.posdst = posdst;
  p.n_tiles = n_tiles;
  p.C = C; p.H = H;
  p.scale = 1.0f / sqrtf((float)ET_HD);
  // the tiling's capacities pick the instantiation: <= 32 sources and <= 64 edges per tile (the decoder's tiles: 16 grid
  // nodes x 3 edges out of <= 17 mesh rows) take the small one -- half the LDS, three workgroups per CU
  if (tile_src_cap <= 32 && tile_edge_cap <= 64) return dispatch_mfma<2, 64>(p, up, as_stream(stream));
  if (tile_src_cap <= 64 && tile_edge_cap <= 160) return dispatch_mfma<4, 160>(p, up, as_stream(stream));
  return fail(ANEMOI_ERR_UNSUPPORTED, "lab_gt_edge_attention_tiles: tiles of up to %d sources / %d edges", tile_src_cap,
              tile_edge_cap);
}
