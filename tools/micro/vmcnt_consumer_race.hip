// Lab (not part of the product).  Round 6, VERDICT r05 item 5(a): the shelved K = 256 row-streaming GEMM lost the
// (-mean rstd) s[n] term of its LayerNorm-fold epilogue in lanes 48 - 63, even elements only, non-deterministically.  The
// listing (tools/micro/patches/r05_gemm_k256_row_streaming_kernel.patch, hipcc 7.2) reads
//     global_load_dwordx2 v[68:69], ...           ; { rstd, -mean rstd } of the lane's row, issued ~2 us earlier
//     ...
//     s_waitcnt vmcnt(1) lgkmcnt(2)               ; the compiler's COUNTED wait for that load
//     v_pk_fma_f32 v[202:203], v[188:189], v[68:69], v[192:193] op_sel:[0,1,0]   ; FIRST instruction behind the wait
// and anything that puts an instruction between the wait and the packed FMA cures it (gpurun_out/r06_s14, r06_s15).  This
// file is the reduced form: one long-latency dwordx2 load per lane, a wait, and a consumer of the load's SECOND dword as the
// very next instruction -- by variant a packed FMA taking it through op_sel (the kernel's form), a packed FMA taking it as
// its high half, a plain v_fma_f32, a v_mov_b32 -- with 0 / 1 / 2 wait states in between.  The result is compared with the
// same arithmetic done again a few hundred cycles later (the registers have landed by then whatever the wait did); stale
// reads are counted per lane quarter and per half (low / high result of the packed operation).
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/vmcnt_consumer_race.hip -o tools/micro/bin/vmcnt_consumer_race
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __attribute__((ext_vector_type(2))) float f32x2_t;

// ONE asm statement per trial: the load, the wait, [wait states], the consumer -- nothing can be scheduled in between.
// WAIT: "A" = s_waitcnt vmcnt(0) with this load alone outstanding; "B" = a second, younger load, then s_waitcnt vmcnt(1).
#define WAIT_A "s_waitcnt vmcnt(0)\n\t"
#define WAIT_B "global_load_dwordx2 %[st2], %[p2], off\n\ts_waitcnt vmcnt(1)\n\t"
#define RACE_KERNEL(NAME, WAIT, NOPS, CONSUMER, WANT_LO, WANT_HI)                                                             \
  __global__ __launch_bounds__(256) void NAME(const f32x2_t* __restrict__ table, int64_t n, int iters,                       \
                                              unsigned* __restrict__ counts) {                                               \
    const int lane = threadIdx.x & 63;                                                                                       \
    uint64_t idx = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 0x9E3779B97F4A7C15ull;                                       \
    unsigned bad_lo = 0, bad_hi = 0;                                                                                         \
    for (int it = 0; it < iters; ++it) {                                                                                     \
      idx = idx * 6364136223846793005ull + 1442695040888963407ull;                                                           \
      const f32x2_t* p = table + (idx >> 11) % (uint64_t)n; /* a far, cold line per lane and trial */                        \
      const f32x2_t* p2 = table + ((idx >> 7) ^ 0x5555) % (uint64_t)n;                                                       \
      const f32x2_t c = {1.5f + (float)(lane & 7), -2.25f - (float)(it & 3)}, b = {0.375f, -0.625f};                        \
      f32x2_t st = {0.f, 0.f}, st2 = {0.f, 0.f}, d;                                                                          \
      asm volatile("global_load_dwordx2 %[st], %[p], off\n\t" WAIT NOPS CONSUMER "\n\ts_waitcnt vmcnt(0)"                    \
                   : [d] "=&v"(d), [st] "+v"(st), [st2] "+v"(st2)                                                            \
                   : [p] "v"(p), [p2] "v"(p2), [c] "v"(c), [b] "v"(b)                                                        \
                   : "memory");                                                                                              \
      asm volatile("s_nop 7\n\ts_nop 7" : "+v"(st), "+v"(st2)); /* the registers have landed by now whatever the wait did */ \
      bad_lo += d.x != (WANT_LO);                                                                                            \
      bad_hi += d.y != (WANT_HI);                                                                                            \
      if (st2.x == 12345.f) bad_lo += 1u << 30; /* (keeps the second load alive) */                                          \
    }                                                                                                                        \
    atomicAdd(&counts[(lane >> 4) * 2 + 0], bad_lo);                                                                         \
    atomicAdd(&counts[(lane >> 4) * 2 + 1], bad_hi);                                                                         \
  }

#define FMA(a, b, c) __builtin_fmaf(a, b, c)
#define PK_SEL "v_pk_fma_f32 %[d], %[c], %[st], %[b] op_sel:[0,1,0]"   /* the kernel's form: both halves x the load's SECOND dword */
#define PK_PLAIN "v_pk_fma_f32 %[d], %[c], %[st], %[b]"                /* low half x first dword, high half x second dword */
#define FMA2 "v_fma_f32 %L[d], %L[c], %H[st], %L[b]\n\tv_fma_f32 %H[d], %H[c], %H[st], %H[b]"
RACE_KERNEL(pk_sel_a0, WAIT_A, "", PK_SEL, FMA(c.x, st.y, b.x), FMA(c.y, st.y, b.y))
RACE_KERNEL(pk_sel_a1, WAIT_A, "s_nop 0\n\t", PK_SEL, FMA(c.x, st.y, b.x), FMA(c.y, st.y, b.y))
RACE_KERNEL(pk_sel_a2, WAIT_A, "s_nop 1\n\t", PK_SEL, FMA(c.x, st.y, b.x), FMA(c.y, st.y, b.y))
RACE_KERNEL(pk_sel_b0, WAIT_B, "", PK_SEL, FMA(c.x, st.y, b.x), FMA(c.y, st.y, b.y))
RACE_KERNEL(pk_sel_b1, WAIT_B, "s_nop 0\n\t", PK_SEL, FMA(c.x, st.y, b.x), FMA(c.y, st.y, b.y))
RACE_KERNEL(pk_sel_b2, WAIT_B, "s_nop 1\n\t", PK_SEL, FMA(c.x, st.y, b.x), FMA(c.y, st.y, b.y))
RACE_KERNEL(pk_plain_a0, WAIT_A, "", PK_PLAIN, FMA(c.x, st.x, b.x), FMA(c.y, st.y, b.y))
RACE_KERNEL(pk_plain_b0, WAIT_B, "", PK_PLAIN, FMA(c.x, st.x, b.x), FMA(c.y, st.y, b.y))

int main() {
  const int64_t n = (int64_t)1 << 27;  // 1 GiB of float2: every access a miss
  f32x2_t* table;
  hipMalloc(&table, n * sizeof(f32x2_t));
  hipMemset(table, 0x3f, n * sizeof(f32x2_t));  // 0x3f3f3f3f = 0.747 in every float: a stale 0 shows
  unsigned* counts;
  hipMalloc(&counts, 64 * sizeof(unsigned));
  const int iters = 400, grid = 2048;
  auto run = [&](void (*kernel)(const f32x2_t*, int64_t, int, unsigned*), const char* what) {
    hipMemset(counts, 0, 64 * sizeof(unsigned));
    kernel<<<grid, 256>>>(table, n, iters, counts);
    unsigned h[8];
    hipMemcpy(h, counts, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-58s stale LOW half, lanes 0-15 / 16-31 / 32-47 / 48-63: %8u %8u %8u %8u   HIGH half: %8u %8u %8u %8u   of %d per quarter\n",
           what, h[0], h[2], h[4], h[6], h[1], h[3], h[5], h[7], grid * 64 * iters);
  };
  run(pk_sel_a0, "vmcnt(0); v_pk_fma op_sel:[0,1,0]");
  run(pk_sel_a1, "vmcnt(0); s_nop 0; v_pk_fma op_sel:[0,1,0]");
  run(pk_sel_a2, "vmcnt(0); s_nop 1; v_pk_fma op_sel:[0,1,0]");
  run(pk_sel_b0, "younger load; vmcnt(1); v_pk_fma op_sel:[0,1,0]");
  run(pk_sel_b1, "younger load; vmcnt(1); s_nop 0; v_pk_fma op_sel:[0,1,0]");
  run(pk_sel_b2, "younger load; vmcnt(1); s_nop 1; v_pk_fma op_sel:[0,1,0]");
  run(pk_plain_a0, "vmcnt(0); v_pk_fma (no op_sel)");
  run(pk_plain_b0, "younger load; vmcnt(1); v_pk_fma (no op_sel)");
  return 0;
}
