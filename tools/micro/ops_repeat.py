"""Run-to-run bit identity of one op under whatever else shares the GPU: python tools/micro/ops_repeat.py linear|edge|mhsa8|ln [iterations]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from anemoi_models_amd import ops, runtime
DEV = "cuda"
what = sys.argv[1]
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 400
g = torch.Generator().manual_seed(1)
if what == "linear":      # persistent four-wave GEMM: LDS-DMA slab ring, inline-asm MFMA
    x = torch.randn(40962, 1024, generator=g).bfloat16().to(DEV)
    w = (torch.randn(4096, 1024, generator=g) / 32).bfloat16().to(DEV)
    b = torch.randn(4096, generator=g).to(DEV)
    f = lambda: ops.linear(x, w, b)
elif what == "mhsa8":     # eight-wave attention kernel (window wider than the sequence): LDS-DMA ring, compiler-scheduled MFMA
    qkv = (torch.randn(40962, 3 * 1024, generator=g)).bfloat16().to(DEV)
    f = lambda: ops.mhsa(qkv, 1, 16, 10**6)
elif what == "ln":        # no LDS-DMA, no MFMA
    x = torch.randn(542080, 1024, generator=g).bfloat16().to(DEV)
    gam, bet = torch.randn(1024, generator=g).to(DEV), torch.randn(1024, generator=g).to(DEV)
    f = lambda: ops.layer_norm(x, gam, bet, 1e-5)
else:                     # edge kernel: plain loads, no LDS
    n, c, h, up = 40962, 1024, 16, 12
    deg = torch.randint(4, 12, (n,), generator=g)
    dst = torch.repeat_interleave(torch.arange(n), deg)
    src = torch.randint(0, n, (int(dst.shape[0]),), generator=g)
    plan = runtime.build_edge_plan(torch.stack([src, dst]).to(DEV), n, n)
    wide = torch.randn(n, 2 * c + h * up, generator=g).bfloat16().to(DEV)
    kv = torch.randn(n, 2 * c, generator=g).bfloat16().to(DEV)
    attr = torch.randn(int(dst.shape[0]), up, generator=g).to(DEV)
    f = lambda: ops.gt_edge_attention_folded(wide[:, c:2 * c], kv[:, :c], kv[:, c:], wide[:, :c], wide[:, 2 * c:], attr, plan.rowptr, plan.col, h, up,
                                             sched=plan.schedule(torch.bfloat16, c))
ref = f().clone()
torch.cuda.synchronize()
print("started", flush=True)
bad = 0
worst = 0.0
for it in range(iters):
    y = f()
    if not torch.equal(y, ref):
        bad += 1
        worst = max(worst, float((y.float() - ref.float()).abs().max() / ref.float().abs().max()))
print(f"{what}: {bad} of {iters} repeats differ from the first (largest relative difference {worst:.2e})", flush=True)
