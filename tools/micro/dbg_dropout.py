import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from anemoi_models_amd import ops
for (b, s, h, d, p) in ((1, 200, 4, 64, 0.1), (1, 700, 4, 64, 0.1), (1, 64, 2, 64, 0.1), (1, 200, 4, 64, 0.0)):
    g = torch.Generator().manual_seed(s + d)
    c = h * d
    qkv = (torch.randn(b * s, 3 * c, generator=g) * 0.8).bfloat16().cuda()
    out, lse = ops.mhsa(qkv, b, h, -1, return_lse=True, dropout_p=p, dropout_seed=123456789 + s)
    bad = ~torch.isfinite(out.float())
    print((b, s, h, d, p), "nan rows", bad.any(1).nonzero().flatten()[:20].tolist(), "nan cols", bad.any(0).nonzero().flatten()[:20].tolist(),
          "lse finite", bool(torch.isfinite(lse).all()), flush=True)
