"""Hunt for a rare run-to-run difference in the Transformer block (tests/test_gpu_parity.py::test_transformer_block_entry_point_is_
the_op_by_op_route[bf16-1024-16-2-700] failed once in ~15 runs): python tools/micro/tfm_block_repeat.py [iterations]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["ANEMOI_AMD_DTYPE"] = "bf16"
from anemoi_models_amd import ops
from anemoi_models_amd.layers.block import TransformerProcessorBlock

DEV = "cuda"
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
channels, heads, b, s = 1024, 16, 2, 700
torch.manual_seed(channels + s)
blk = TransformerProcessorBlock(channels, 4 * channels, heads, "GELU", window_size=16, dropout_p=0.0).to(DEV).eval()
x = (torch.randn(b * s, channels, generator=torch.Generator().manual_seed(1)) * 0.8).bfloat16().to(DEV)


def where(a, ref):
    d = (a != ref)
    rows = d.any(1).nonzero().flatten().tolist()
    cols = d.any(0).nonzero().flatten().tolist()
    return f"{int(d.sum())} elements, rows {rows[:8]}{'...' if len(rows) > 8 else ''}, cols {cols[:12]}{'...' if len(cols) > 12 else ''}"


with torch.no_grad():
    for abi in (False, True):
        TransformerProcessorBlock.block_abi = abi
        ref = blk.native(x, b).clone()
        bad = 0
        for it in range(iters):
            y = blk.native(x, b)
            if not torch.equal(y, ref):
                bad += 1
                print(f"  block_abi={abi} iteration {it}: {where(y, ref)}", flush=True)
        print(f"block (block_abi={abi}): {bad} of {iters} repeats differ", flush=True)
    # the pieces, each repeated on fixed inputs
    g = torch.Generator().manual_seed(3)
    qkv = (torch.randn(b * s, 3 * channels, generator=g) * 0.8).bfloat16().to(DEV)
    ref = ops.mhsa(qkv, b, heads, -1).clone()
    bad = sum(0 if torch.equal(ops.mhsa(qkv, b, heads, -1), ref) else 1 for _ in range(iters))
    print(f"mhsa S={s} B={b} H={heads} D={channels // heads}: {bad} of {iters} repeats differ", flush=True)
    for (n, k, res, stats) in ((3 * channels, channels, False, False), (channels, channels, True, True), (4 * channels, channels, False, False),
                               (channels, 4 * channels, True, True)):
        w = (torch.randn(n, k, generator=g) / k**0.5).bfloat16().to(DEV)
        bias = torch.randn(n, generator=g).to(DEV)
        xin = (torch.randn(b * s, k, generator=g)).bfloat16().to(DEV)
        r = torch.randn(b * s, n, generator=g).bfloat16().to(DEV) if res else None
        kw = dict(residual=r, stats_eps=1e-5) if stats else dict(residual=r)
        ref = ops.linear(xin, w, bias, **kw).clone()
        bad = 0
        for it in range(iters):
            y = ops.linear(xin, w, bias, **kw)
            if not torch.equal(y, ref):
                bad += 1
                print(f"  linear {b*s}x{n}x{k} iteration {it}: {where(y, ref)}", flush=True)
        print(f"linear M={b*s} N={n} K={k} residual={res} stats={stats}: {bad} of {iters} repeats differ", flush=True)
