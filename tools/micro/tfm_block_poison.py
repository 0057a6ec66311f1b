"""Does any kernel of the Transformer block read memory it did not write?  Every call runs on freshly POISONED allocator blocks
(NaN / huge / random bit patterns in everything torch.empty hands out) and must equal the clean result bit for bit:
python tools/micro/tfm_block_poison.py [iterations]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["ANEMOI_AMD_DTYPE"] = "bf16"
from anemoi_models_amd import ops
from anemoi_models_amd.layers.block import TransformerProcessorBlock

DEV = "cuda"
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 60
channels, heads, b, s = 1024, 16, 2, 700
torch.manual_seed(channels + s)
blk = TransformerProcessorBlock(channels, 4 * channels, heads, "GELU", window_size=16, dropout_p=0.0).to(DEV).eval()
x = (torch.randn(b * s, channels, generator=torch.Generator().manual_seed(1)) * 0.8).bfloat16().to(DEV)


def poison(kind):
    """Fill ~1.5 GiB of allocator blocks of many sizes with a pattern and give them back (torch.empty then reuses them)."""
    sizes = [1 << k for k in range(10, 27)] + [1400 * 1024 * 2, 1400 * 3072 * 2, 1400 * 4096 * 2, 1400 * 16 * 8, 1400 * 8 * 8, 11200, 22400]
    junk = []
    for n in sizes * 2:
        t = torch.empty(n, dtype=torch.uint8, device=DEV)
        if kind == "nan":
            t.view(torch.int16 if n % 2 == 0 else torch.uint8).fill_(0x7FC0 if n % 2 == 0 else 0xFF)
        elif kind == "huge":
            t.view(torch.int16 if n % 2 == 0 else torch.uint8).fill_(0x7F00 if n % 2 == 0 else 0x7F)
        else:
            t.random_(0, 256)
        junk.append(t)
    del junk


def where(a, ref):
    d = (a != ref) & ~(torch.isnan(a.float()) & torch.isnan(ref.float()))
    rows = d.any(1).nonzero().flatten().tolist()
    cols = d.any(0).nonzero().flatten().tolist()
    return f"{int(d.sum())} elements (NaN in result: {int(torch.isnan(a.float()).sum())}), rows {rows[:8]}, cols {cols[:12]}"


with torch.no_grad():
    for abi in (False, True):
        TransformerProcessorBlock.block_abi = abi
        ref = blk.native(x, b).clone()
        bad = 0
        for it in range(iters):
            kind = ("nan", "huge", "random")[it % 3]
            poison(kind)
            y = blk.native(x, b)
            if not torch.equal(y, ref):
                bad += 1
                if bad <= 6:
                    print(f"  block_abi={abi} iteration {it} ({kind}): {where(y, ref)}", flush=True)
        print(f"block (block_abi={abi}) on poisoned memory: {bad} of {iters} differ", flush=True)
    g = torch.Generator().manual_seed(3)
    qkv = (torch.randn(b * s, 3 * channels, generator=g) * 0.8).bfloat16().to(DEV)
    ref = ops.mhsa(qkv, b, heads, -1).clone()
    bad = 0
    for it in range(iters):
        poison(("nan", "huge", "random")[it % 3])
        y = ops.mhsa(qkv, b, heads, -1)
        if not torch.equal(y, ref):
            bad += 1
            if bad <= 4:
                print(f"  mhsa iteration {it}: {where(y, ref)}", flush=True)
    print(f"mhsa on poisoned memory: {bad} of {iters} differ", flush=True)
    for (n, k, res, stats) in ((3 * channels, channels, False, False), (channels, channels, True, True), (4 * channels, channels, False, False),
                               (channels, 4 * channels, True, True)):
        w = (torch.randn(n, k, generator=g) / k**0.5).bfloat16().to(DEV)
        bias = torch.randn(n, generator=g).to(DEV)
        xin = (torch.randn(b * s, k, generator=g)).bfloat16().to(DEV)
        r = torch.randn(b * s, n, generator=g).bfloat16().to(DEV) if res else None
        kw = dict(residual=r, stats_eps=1e-5) if stats else dict(residual=r)
        ref = ops.linear(xin, w, bias, **kw).clone()
        ref_ln = ops.layer_norm(ref, torch.ones(n, device=DEV), torch.zeros(n, device=DEV), 1e-5).clone() if stats else None
        bad = 0
        for it in range(iters):
            poison(("nan", "huge", "random")[it % 3])
            y = ops.linear(xin, w, bias, **kw)
            same = torch.equal(y, ref)
            if stats:  # the carried row statistics feed the next LayerNorm
                same = same and torch.equal(ops.layer_norm(y, torch.ones(n, device=DEV), torch.zeros(n, device=DEV), 1e-5), ref_ln)
            if not same:
                bad += 1
                if bad <= 4:
                    print(f"  linear {b*s}x{n}x{k} iteration {it}: {where(y, ref)}", flush=True)
        print(f"linear M={b*s} N={n} K={k} residual={res} stats={stats} on poisoned memory: {bad} of {iters} differ", flush=True)
