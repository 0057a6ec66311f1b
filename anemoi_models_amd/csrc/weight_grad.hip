// Weight gradient  dW[n, k] = sum_m dY[m, n] * x[m, k]  on the gfx950 matrix cores WITHOUT transposed copies.
//
// Both operands are row-major with the REDUCTION index m as the slow dimension ("TN"): an MFMA fragment needs, per
// lane, eight consecutive m of ONE column -- a 2-byte gather with stride ld in memory.  Rounds 1-2 transposed dY and x
// through HBM first (anemoi_transpose_chunked, 12 % of a config-3 training step) and ran the forward kernel on the
// copies.  Here the panels go to LDS as they lie in memory (full 512-byte rows, LDS-DMA) and the fragments come out of
// LDS through gfx950's transposing read ds_read_b64_tr_b16: a 16-lane group reads a [4 m][16 columns] block (8 bytes
// per lane) and lane c receives the 4 m of column c -- two of them make the 8-deep fragment of v_mfma_f32_16x16x32_bf16.
//
// Tile: 256 dY columns x 256 x columns per workgroup, four waves (one per SIMD) of 128 x 128, 256 accumulators per
// lane in AGPRs, reduction slabs of 64 rows, two 64 KiB LDS stages -- the loop skeleton (one memory instruction per
// MFMA gap, two barriers per slab, counted vmcnt) is the one of linear_bf16_w4_kernel (gemm.hip), which documents why
// it looks the way it does.  The reduction over the M rows is cut into `chunks` (a [1024 x 4096] gradient is only 64
// tiles): every (chunk, tile) pair is one unit of the persistent tile list and writes its f32 partial tile; the
// caller sums the chunks (deterministic, ops.weight_grad).
//
// LDS image of a panel slab: row m (512 B = 16 double-slots of 16 columns) keeps its columns, but double-slot p holds
// the source double-slot p ^ g(m), g(m) = (m & 3) | ((m >> 3) & 1) << 2: the eight rows a half-wave reads in one
// transposing read (m = 8 fq + 4 h + 0..3, fq in {0, 1} or {2, 3}) then lie in eight different 32-byte bank groups.
// The XOR is applied to the per-lane SOURCE address of the LDS-DMA (whose LDS side is lane-linear), so every DMA
// instruction still reads two whole 512-byte rows.
//
// Replaces (together with ops.weight_grad) torch autograd's `grad_output.t() @ input` of every nn.Linear under
// models/encoder_processor_decoder.py:167-233 (training).
#include <type_traits>
#include <utility>

#include "common.hpp"

namespace anemoi {
namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 wbf16x8_t;
typedef __attribute__((ext_vector_type(4))) short ws16x4_t;
typedef __attribute__((ext_vector_type(8))) short ws16x8_t;
typedef __attribute__((ext_vector_type(4))) float wf32x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned wu32x4_t;
typedef __attribute__((ext_vector_type(4))) int wi32x4_t;

constexpr int TN_TILE = 256;                      // output tile: 256 dY columns x 256 x columns
constexpr int TN_ROWS = 64;                       // reduction rows per slab
constexpr int TN_PANEL = TN_ROWS * TN_TILE * 2;   // 32 KiB: one operand panel of a slab
constexpr int TN_STAGE = 2 * TN_PANEL;            // 64 KiB: dY panel + x panel
constexpr int TN_LDS = 2 * TN_STAGE;              // two stages

#define ANEMOI_TN_MFMA(c, a, b) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b))
#define ANEMOI_TN_MFMA_V(c, a, b) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b))

// One 16-byte-per-lane LDS-DMA (64 lanes -> 1 KiB at LDS byte address `lds`), issued from inline asm ON PURPOSE: the
// compiler's wait-count pass knows the builtin form writes LDS and puts an s_waitcnt vmcnt(0) in front of the next LDS
// read it sees -- two full drains of the DMA queue per slab (measured in the ISA of the first version of this kernel).
// Here every wait on a DMA is the explicit counted one of the slab schedule.
__device__ __forceinline__ void tn_dma16(wi32x4_t rsrc, unsigned lds, int voffset, int soffset) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :
               : "s"(lds), "v"(voffset), "s"(rsrc), "s"(soffset)
               : "memory");
}

__device__ __forceinline__ wi32x4_t tn_rsrc(const void* base, int64_t bytes) {
  const uint64_t a = reinterpret_cast<uint64_t>(base);
  return wi32x4_t{__builtin_amdgcn_readfirstlane((int)(uint32_t)a),
                  __builtin_amdgcn_readfirstlane((int)(uint32_t)((a >> 32) & 0xffffu)),
                  __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000};
}

template <typename F, int... S>
__device__ __forceinline__ void tn_static_for(F&& f, std::integer_sequence<int, S...>) {
  (f(std::integral_constant<int, S>{}), ...);
}

// out [chunks][N][K] f32 (row pitch K); dY [M, ldy], x [M, ldx] bf16; chunk c covers rows c * chunk_rows ... (zeros
// behind row M through the buffer descriptors' range check).
// NJ: dY fragments per wave whose column sums (bias gradient) this wave accumulates: 0 (no bias), 1, 2 or 4.
template <int NJ>
__global__ __launch_bounds__(256) void weight_grad_tn_kernel(const bf16_t* __restrict__ DY, int64_t ldy,
                                                             const bf16_t* __restrict__ X, int64_t ldx,
                                                             float* __restrict__ OUT, float* __restrict__ DB,
                                                             int64_t out_stride, int64_t db_stride, int64_t M, int N, int K,
                                                             int chunk_rows, int64_t n_tiles, int tiles_per_chunk,
                                                             int kt_count) {
  constexpr int NS = 64;    // MFMAs per 32-deep reduction step
  constexpr int NRD = 16;   // fragments per step = LDS-DMA instructions per slab and wave
  constexpr int G1 = 23, SP = 5, G2 = 103;
  static_assert(G1 + 1 + (NRD - 1) * SP < G2 && G2 + NRD < 2 * NS, "slab schedule");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int64_t xcd = blockIdx.x & 7, bix = blockIdx.x >> 3, bpx = gridDim.x >> 3;
  const int64_t q8 = n_tiles / 8, r8 = n_tiles % 8;
  const int64_t chunk_start = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
  const int64_t chunk_len = q8 + (xcd < r8 ? 1 : 0);
  if (bix >= chunk_len) return;
  const int nk = chunk_rows / TN_ROWS;  // >= 2 (launcher)

  // ---- staging side: wave w moves slab rows 16 w + 2 i + (lane >> 5), i = 0..7, of both panels (two whole 512-byte
  //      rows per DMA instruction); lane slot s' = lane & 31 of a row receives the source chunk ((s' >> 1) ^ g) * 2 + (s' & 1)
  const int l_hi = lane >> 5, l_ds = (lane & 31) >> 1, l_half = lane & 1;
  int voy[4], vox[4];  // variants v = (i & 1) + 2 * ((i >> 2) & 1) of g
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const int g = (2 * (v & 1) + l_hi) | ((v >> 1) << 2);
    const int col_bytes = ((l_ds ^ g) << 5) + (l_half << 4);
    voy[v] = l_hi * (int)ldy * 2 + col_bytes;
    vox[v] = l_hi * (int)ldx * 2 + col_bytes;
  }
  wi32x4_t yrs, xrs;  // buffer descriptors of the tile whose slabs are being staged
  auto tile_parts = [&](int64_t tile, int& c, int& n0, int& k0) {
    c = (int)(tile / tiles_per_chunk);
    const int r = (int)(tile - (int64_t)c * tiles_per_chunk);
    n0 = (r / kt_count) * TN_TILE;
    k0 = (r % kt_count) * TN_TILE;
  };
  auto set_tile = [&](int64_t tile) {
    int c, n0, k0;
    tile_parts(tile, c, n0, k0);
    const int64_t row0 = (int64_t)c * chunk_rows;
    int64_t rows = M - row0 < chunk_rows ? M - row0 : chunk_rows;
    rows = rows > 0 ? rows : 0;
    // valid bytes: up to the last column of the last row of the chunk (never beyond the operand's own storage)
    const int64_t yb = rows > 0 ? ((rows - 1) * ldy + (N - n0)) * 2 : 0;
    const int64_t xb = rows > 0 ? ((rows - 1) * ldx + (K - k0)) * 2 : 0;
    yrs = tn_rsrc(DY + row0 * ldy + n0, yb);
    xrs = tn_rsrc(X + row0 * ldx + k0, xb);
  };
  auto set_null = [&]() {
    yrs = tn_rsrc(DY, 0);
    xrs = tn_rsrc(X, 0);
  };
  const int yrow2 = (int)ldy * 2, xrow2 = (int)ldx * 2;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  auto dma_y = [&](int i, int kt, unsigned dst) {
    tn_dma16(yrs, dst, voy[(i & 1) + 2 * ((i >> 2) & 1)], (kt * TN_ROWS + wid * 16 + 2 * i) * yrow2);
  };
  auto dma_x = [&](int i, int kt, unsigned dst) {
    tn_dma16(xrs, dst, vox[(i & 1) + 2 * ((i >> 2) & 1)], (kt * TN_ROWS + wid * 16 + 2 * i) * xrow2);
  };
  auto stage_all = [&](int kt, int buf) {
    const unsigned ys = lds0 + buf * TN_STAGE + wid * 8192;
    const unsigned xs = ys + TN_PANEL;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      dma_y(i, kt, ys + i * 1024);
      dma_x(i, kt, xs + i * 1024);
    }
  };

  // ---- compute side.  Fragment f (16 columns) of a panel, reduction step ks: the 16-lane group fq reads rows
  //      32 ks + 8 fq + 4 h + mr (mr = (lane & 15) >> 2), 4 columns at 4 cq (cq = lane & 3), h = 0, 1.
  const int wm = wid >> 1, wn = wid & 1;  // wm: dY column half (MFMA "B" side), wn: x column half ("A" side)
  const int fr = lane & 15, fq = lane >> 4;
  const int mr = fr >> 2, cq = fr & 3;
  const int gl = mr | ((fq & 1) << 2);
  const int rbase = (8 * fq + mr) * 512 + cq * 8;
  int ra[8], rb[8];  // byte offsets of (x panel, fragment i) / (dY panel, fragment j) inside a stage, ks = h = 0
#pragma unroll
  for (int f = 0; f < 8; ++f) {
    ra[f] = TN_PANEL + rbase + ((8 * wn + (f ^ gl)) << 5);
    rb[f] = rbase + ((8 * wm + (f ^ gl)) << 5);
  }
  auto ld_frag = [&](int off) {
    const ws16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) ws16x4_t*)(smem + off));
    const ws16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) ws16x4_t*)(smem + off + 4 * 512));
    return __builtin_bit_cast(wbf16x8_t, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
  };
  auto ldA = [&](int buf, int ks, int i) { return ld_frag(buf * TN_STAGE + ks * (32 * 512) + ra[i]); };
  auto ldB = [&](int buf, int ks, int j) { return ld_frag(buf * TN_STAGE + ks * (32 * 512) + rb[j]); };

  wf32x4_t acc[8][8];
  wbf16x8_t a0[8], b0[8], a1[8], b1[8];
  // Bias gradient db[n] = sum_m dY[m, n] (NJ > 0): the dY fragments pass through the registers of 2 * kt_count waves
  // (both x-column halves of every x-column tile), so the 8 fragments of a wave's dY half are shared out among the
  // first `holders` = 2 * min(kt_count, 4) of them: wave sel takes j = sel, sel + holders, ... (at most NJ), reads that
  // fragment ONCE MORE from LDS -- a run-time j is only a run-time LDS address -- and adds one MFMA with an all-ones "A"
  // fragment per reduction step (every accumulator row then holds the column sums): NJ extra MFMAs per 64, no branches
  // in the slab, no pass over dY of its own.  A wave without a fragment (or slot q >= its count) multiplies by zeros.
  constexpr int NJX = NJ > 0 ? NJ : 1;
  wf32x4_t bs[NJX];
  wbf16x8_t bq0[NJX], bq1[NJX], onesq[NJX];
  int rbq[NJX], rbq_next[NJX], jq[NJX];
#pragma unroll
  for (int q = 0; q < NJX; ++q) {
    bs[q] = wf32x4_t{0.f, 0.f, 0.f, 0.f};
    rbq[q] = rbq_next[q] = rb[0];
    jq[q] = -1;
  }
  const int holders_k = kt_count < 4 ? kt_count : 4;
  auto bias_slots = [&](int64_t t, int (&addr)[NJX], int (&jj)[NJX]) {  // this wave's fragments in tile t
    const int kt_idx = (int)((t % tiles_per_chunk) % kt_count);
    const int sel = kt_idx * 2 + wn;
#pragma unroll
    for (int q = 0; q < NJX; ++q) {
      const int j = sel + q * 2 * holders_k;
      const bool on = kt_idx < holders_k && j < 8;
      jj[q] = on ? j : -1;
      addr[q] = rbase + ((8 * wm + ((on ? j : 0) ^ gl)) << 5);
    }
  };
  set_tile(chunk_start + bix);
  stage_all(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    b0[u] = ldB(0, 0, u);
    a0[u] = ldA(0, 0, u);
  }
  if constexpr (NJ > 0) {
    bias_slots(chunk_start + bix, rbq, jq);
#pragma unroll
    for (int q = 0; q < NJ; ++q) bq0[q] = ld_frag(rbq[q]);
  }
  stage_all(1, 1);

  wbf16x8_t zfrag = {0, 0, 0, 0, 0, 0, 0, 0};  // (opaque zero fragment: see linear_bf16_w4_kernel)
  asm volatile("" : "+v"(zfrag));
  asm volatile("s_nop 3" ::: "memory");
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j)
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %1, 0" : "=a"(acc[i][j]) : "v"(zfrag));

  int g = 0;  // running slab counter over all tiles of this workgroup: LDS stage = g & 1
  int64_t li = bix, tile = 0;
  int k = 0;
  bool has_next = false;
  for (;;) {
    if (k == 0) {
      tile = chunk_start + li;
      has_next = li + bpx < chunk_len;
      if constexpr (NJ > 0) {
        bias_slots(tile, rbq, jq);
        const wbf16x8_t one8 = {1, 1, 1, 1, 1, 1, 1, 1}, zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int q = 0; q < NJ; ++q) {
          onesq[q] = jq[q] >= 0 ? one8 : zero8;
          asm volatile("" : "+v"(onesq[q]));
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_nop 7" ::: "memory");  // accumulator zeroing (MFMA pipe) -> first MFMA
      __builtin_amdgcn_sched_barrier(0);
    }
    auto slab = [&](int kt_stage, bool last_slab) {
      const int buf = g & 1, nbuf = buf ^ 1;
      const unsigned ysd = lds0 + buf * TN_STAGE + wid * 8192;
      const unsigned xsd = ysd + TN_PANEL;
      tn_static_for(
          [&](auto s_tag) {
            constexpr int s = decltype(s_tag)::value;
            if constexpr (s < NS) ANEMOI_TN_MFMA(acc[s / 8][s % 8], a0[s / 8], b0[s % 8]);
            else ANEMOI_TN_MFMA(acc[(s - NS) / 8][s % 8], a1[(s - NS) / 8], b1[s % 8]);
            if constexpr (NJ > 0 && (s == NS - 1 || s == 2 * NS - 1)) {  // column sums of this step's extra dY fragments
#pragma unroll
              for (int q = 0; q < NJ; ++q) {
                if constexpr (s < NS) ANEMOI_TN_MFMA_V(bs[q], onesq[q], bq0[q]);
                else ANEMOI_TN_MFMA_V(bs[q], onesq[q], bq1[q]);
              }
            }
            if constexpr (NJ > 0 && s >= NRD && s < NRD + NJ)  // extra fragments of (this slab, ks = 1): before barrier 1
              bq1[s - NRD] = ld_frag(buf * TN_STAGE + 32 * 512 + rbq[s - NRD]);
            if constexpr (NJ > 0 && s > G2 + NRD && s - G2 - 1 - NRD < NJ)  // ... of (next slab, ks = 0)
              bq0[s - G2 - 1 - NRD] = ld_frag(nbuf * TN_STAGE + (last_slab ? rbq_next : rbq)[s - G2 - 1 - NRD]);
            if constexpr (s < NRD) {  // fragments of (this slab, ks = 1)
              if constexpr (s < 8) b1[s] = ldB(buf, 1, s);
              else a1[s - 8] = ldA(buf, 1, s - 8);
            }
            if constexpr (s == G1) {  // barrier 1: every wave has read this slab's buffer completely
              asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
              __builtin_amdgcn_sched_barrier(0);
              __builtin_amdgcn_s_barrier();
              __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (s > G1 && (s - G1 - 1) % SP == 0 && (s - G1 - 1) / SP < NRD) {  // refill it: one DMA per SP gaps
              constexpr int t = (s - G1 - 1) / SP;
              if constexpr ((t & 1) == 0) dma_y(t >> 1, kt_stage, ysd + (t >> 1) * 1024);
              else dma_x(t >> 1, kt_stage, xsd + (t >> 1) * 1024);
            }
            if constexpr (s == G2) {  // barrier 2: the other buffer (staged one slab ago) is complete for everyone
              asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
              __builtin_amdgcn_sched_barrier(0);
              __builtin_amdgcn_s_barrier();
              __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (s > G2 && s - G2 - 1 < NRD) {  // fragments of (next slab, ks = 0)
              constexpr int t = s - G2 - 1;
              if constexpr (t < 8) b0[t] = ldB(nbuf, 0, t);
              else a0[t - 8] = ldA(nbuf, 0, t - 8);
            }
          },
          std::make_integer_sequence<int, 2 * NS>{});
      ++g;
    };
    if (k == nk - 2) {  // from here on the staged slabs are the next tile's
      if (has_next) set_tile(tile + bpx);
      else set_null();
      if constexpr (NJ > 0) {
        int jn[NJX];
        bias_slots(has_next ? tile + bpx : tile, rbq_next, jn);
      }
    }
    slab(k + 2 < nk ? k + 2 : k + 2 - nk, k == nk - 1);
    ++k;
    if (k < nk) continue;
    k = 0;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // last MFMA -> accumulator reads below
    __builtin_amdgcn_sched_barrier(0);

    // ---- epilogue: lane (fr, fq) holds, for dY column n0 + 128 wm + 16 j + fr, the x columns k0 + 128 wn + 16 i + 4 fq
    //      + 0..3 -- one 16-byte f32 store into the chunk's partial tile.  (Short against >= 40 slabs of a tile.)
    int c, n0, k0;
    tile_parts(tile, c, n0, k0);
    int rows_here = N - n0 < TN_TILE ? N - n0 : TN_TILE;
    const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(OUT + (int64_t)c * out_stride + (int64_t)n0 * K), 0, rows_here * K * 4, 0x00020000);
    int fr_e = fr, fq_e = fq;
    asm volatile("" : "+v"(fr_e), "+v"(fq_e));
    int vo[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int col = k0 + 128 * wn + 16 * i + 4 * fq_e;
      vo[i] = col < K ? (fr_e * K + col) * 4 : 0x7f000000;  // K % 4 == 0 (launcher): a lane's 4 columns are in or out together
    }
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");  // the next tile's slab 0 has landed before the stores queue up
    __builtin_amdgcn_sched_barrier(0);
    tn_static_for(
        [&](auto j_tag) {
          constexpr int j = decltype(j_tag)::value;
          wf32x4_t cv[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            asm volatile("" : "+a"(acc[i][j]));
            cv[i] = acc[i][j];
            asm volatile("" : "+v"(cv[i]));
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = 0; i < 8; ++i)
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %1, 0" : "+a"(acc[i][j]) : "v"(zfrag));
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(wu32x4_t, cv[i]), ors, vo[i],
                                                   (128 * wm + 16 * j) * K * 4, 0);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_nop 1" ::: "memory");  // store-data hazard with an SGPR soffset (see linear_bf16_w4_kernel)
            __builtin_amdgcn_sched_barrier(0);
          }
        },
        std::make_integer_sequence<int, 8>{});
    if constexpr (NJ > 0) {  // every row of bs[q] holds the column sums of dY columns n0 + 128 wm + 16 jq + fr over this chunk
#pragma unroll
      for (int q = 0; q < NJ; ++q) {
        const int n = n0 + 128 * wm + 16 * jq[q] + fr_e;
        if (jq[q] >= 0 && fq_e == 0 && n < N) DB[(int64_t)c * db_stride + n] = bs[q][0];
        bs[q] = wf32x4_t{0.f, 0.f, 0.f, 0.f};
      }
    }
    li += bpx;
    if (li >= chunk_len) break;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no LDS-DMA may outlive the workgroup
}

}  // namespace
}  // namespace anemoi

// partial [chunks][N][K] f32 <- per-chunk dY^T x;  bias_partial (optional) [chunks][N] f32 <- per-chunk column sums of dY;
// chunk c of either starts partial_stride / bias_stride floats behind chunk c - 1 (>= N * K / >= N: the two may share
// one buffer [chunks][N * K + N], which a single anemoi_col_sum then reduces over the chunks)
extern "C" int anemoi_weight_grad_tn(const void* dy, int64_t ldy, const void* x, int64_t ldx, void* partial,
                                     int64_t partial_stride, void* bias_partial, int64_t bias_stride, int64_t M, int N,
                                     int K, int chunk_rows, anemoi_stream_t stream) {
  using namespace anemoi;
  ANEMOI_REQUIRE(dy && x && partial && M > 0 && N > 0 && K > 0 && ldy >= N && ldx >= K && chunk_rows > 0,
                 ANEMOI_ERR_INVALID, "anemoi_weight_grad_tn: bad argument");
  ANEMOI_REQUIRE(partial_stride >= (int64_t)N * K && partial_stride % 4 == 0 && (bias_partial == nullptr || bias_stride >= N),
                 ANEMOI_ERR_INVALID, "anemoi_weight_grad_tn: chunk strides must cover a chunk (partial: a multiple of 4 floats)");
  ANEMOI_REQUIRE(chunk_rows % TN_ROWS == 0 && chunk_rows >= 2 * TN_ROWS, ANEMOI_ERR_INVALID,
                 "anemoi_weight_grad_tn: chunk_rows must be a multiple of %d, at least %d", TN_ROWS, 2 * TN_ROWS);
  ANEMOI_REQUIRE((uintptr_t)dy % 16 == 0 && (uintptr_t)x % 16 == 0 && (uintptr_t)partial % 16 == 0 && ldy % 8 == 0 &&
                     ldx % 8 == 0 && K % 8 == 0 && N % 8 == 0,
                 ANEMOI_ERR_INVALID, "anemoi_weight_grad_tn: operands 16-byte aligned; row pitches, N and K multiples of 8");
  const int64_t ld_max = ldy > ldx ? ldy : ldx;
  ANEMOI_REQUIRE((int64_t)chunk_rows * ld_max * 2 < ((int64_t)1 << 31) && (int64_t)TN_TILE * K * 4 < ((int64_t)1 << 31),
                 ANEMOI_ERR_UNSUPPORTED, "anemoi_weight_grad_tn: chunk of %d rows x pitch %lld exceeds the 2 GiB descriptor range",
                 chunk_rows, (long long)ld_max);
  const int64_t chunks = (M + chunk_rows - 1) / chunk_rows;
  const int nt = (N + TN_TILE - 1) / TN_TILE, kt = (K + TN_TILE - 1) / TN_TILE;
  const int64_t tiles = chunks * nt * kt;
  ANEMOI_REQUIRE(tiles < ((int64_t)1 << 31) && (int64_t)nt * kt < ((int64_t)1 << 31), ANEMOI_ERR_UNSUPPORTED,
                 "anemoi_weight_grad_tn: too many tiles");
  static PerDeviceOnce raised;
  const int raise_dev = raised.pending();
  if (raise_dev >= 0) {
    const void* kernels[4] = {reinterpret_cast<const void*>(weight_grad_tn_kernel<0>),
                              reinterpret_cast<const void*>(weight_grad_tn_kernel<1>),
                              reinterpret_cast<const void*>(weight_grad_tn_kernel<2>),
                              reinterpret_cast<const void*>(weight_grad_tn_kernel<4>)};
    for (const void* kp : kernels)
      if (hipFuncSetAttribute(kp, hipFuncAttributeMaxDynamicSharedMemorySize, TN_LDS) != hipSuccess)
        return fail(ANEMOI_ERR_LAUNCH, "anemoi_weight_grad_tn: cannot raise the dynamic LDS limit to %d", TN_LDS);
    raised.done(raise_dev);
  }
  int64_t blocks = tiles < 256 ? (tiles + 7) / 8 * 8 : 256;  // one persistent workgroup per CU, whole XCD rows
  const int holders = 2 * (kt < 4 ? kt : 4);
  const int nj = bias_partial == nullptr ? 0 : (8 + holders - 1) / holders;  // 4, 2, 2, 1 for 1, 2, 3, >= 4 x-column tiles
#define ANEMOI_TN_LAUNCH(NJV)                                                                                          \
  hipLaunchKernelGGL(weight_grad_tn_kernel<NJV>, dim3((unsigned)blocks), dim3(256), TN_LDS, as_stream(stream),          \
                     static_cast<const bf16_t*>(dy), ldy, static_cast<const bf16_t*>(x), ldx,                           \
                     static_cast<float*>(partial), static_cast<float*>(bias_partial), partial_stride, bias_stride, M, N, K, \
                     chunk_rows, tiles, nt * kt, kt)
  switch (nj) {
    case 0: ANEMOI_TN_LAUNCH(0); break;
    case 1: ANEMOI_TN_LAUNCH(1); break;
    case 2: ANEMOI_TN_LAUNCH(2); break;
    default: ANEMOI_TN_LAUNCH(4); break;
  }
#undef ANEMOI_TN_LAUNCH
  return check_launch("anemoi_weight_grad_tn");
}
