"""Correctness sweep of anemoi_linear (bf16 fast path) on shapes that exercise single-tile workgroups, ragged M / N,
tiny K, residual / activation epilogues, the half-tile remainder launch and the LayerNorm fold (GPU only; not part of
the product).  The output is a row slice of a larger buffer whose rows behind M must keep their sentinel."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if "--lib" in sys.argv:  # lab builds of the kernel library (tools/micro/bin/*.so), e.g. the tail-epilogue GEMM for A/B runs
    _i = sys.argv.index("--lib")
    from anemoi_models_amd import _lib as _lab_lib

    _lab_lib.LIB_PATH = os.path.abspath(sys.argv[_i + 1])  # ANEMOI_LAB_LIB
    del sys.argv[_i:_i + 2]
import torch
import torch.nn.functional as F

from anemoi_models_amd import ops, runtime

torch.manual_seed(0)
ACT = {"Identity": lambda t: t, "ReLU": F.relu, "GELU": F.gelu, "SiLU": F.silu}
CASES = [
    (2050, 384, 256, "ReLU", True, False), (2048, 512, 256, "Identity", False, False),
    (4096, 512, 1024, "Identity", False, False), (4096, 512, 1024, "Identity", True, False),
    (4096, 384, 1024, "Identity", False, False), (65536, 1024, 128, "GELU", True, False),
    (40962, 4288, 1024, "Identity", False, True), (40962, 1024, 1216, "Identity", True, False),
    (3072, 2240, 1024, "SiLU", True, False), (131072, 256, 128, "Identity", False, False),
    (70000, 520, 192, "GELU", True, False),
    # remainder round as half tiles
    (5120, 1024, 4096, "Identity", True, False), (11264, 2048, 512, "GELU", False, False),
    (10243, 1024, 1024, "SiLU", True, False),
    # ragged last row tile (tail of 9..255 rows), alone and combined with the half-tile launch / the LayerNorm fold
    (67718, 4096, 1024, "GELU", False, True), (67718, 1024, 4096, "Identity", True, False),
    (5254, 1024, 4096, "Identity", True, False), (1100, 512, 256, "Identity", False, False),
    (33921, 2240, 1024, "Identity", False, True), (5129, 1024, 1216, "Identity", True, False),
    (1033, 256, 128, "ReLU", True, True), (20608, 2048, 1024, "Identity", False, True),
]
tot = 0
for (m, n, k, act, res, fold) in CASES:
    x = torch.randn(m, k).bfloat16().cuda()
    w32 = torch.randn(n, k) / k**0.5
    b = torch.randn(n).cuda()
    r = torch.randn(m, n).bfloat16().cuda() if res else None
    pad = 300
    buf = torch.full((m + pad, n), 512.0, dtype=torch.bfloat16, device="cuda")
    if fold:
        gamma, beta = (1.0 + 0.2 * torch.randn(k)).cuda(), (0.1 * torch.randn(k)).cuda()
        wq, bq, colsum = runtime.fold_layer_norm(w32.cuda(), b, gamma, beta, torch.bfloat16)
        stats = ops.row_stats(x, 1e-5)
        for _ in range(2):
            ops.linear(x, wq, bq, act=act, residual=r, out=buf[:m], ln=(stats, colsum))
        xn = F.layer_norm(x.float(), (k,), gamma, beta, 1e-5)
        want = ACT[act](F.linear(xn, w32.cuda(), b))
        tol = 0.15
    else:
        w = w32.bfloat16().cuda()
        for _ in range(2):
            ops.linear(x, w, b, act=act, residual=r, out=buf[:m])
        want = ACT[act](F.linear(x.float(), w.float(), b))
        tol = 0.08
    if res:
        want = want + r.float()
    got = buf[:m].float()
    bad = ~torch.isfinite(got) | ((got - want).abs() > tol)
    clobbered = int((buf[m:].float() != 512.0).sum())
    tot += int(bad.sum()) + clobbered
    print(m, n, k, act, "res" if res else "-", "ln" if fold else "-", "bad", int(bad.sum()), "rows behind M touched",
          clobbered, "max err", float((got - want).abs().max()), flush=True)
print("TOTAL BAD", tot)
