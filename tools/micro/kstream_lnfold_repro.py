#!/usr/bin/env python
"""Round 6, VERDICT r05 item 5(a): the LayerNorm-fold epilogue of the shelved K = 256 row-streaming GEMM
(tools/micro/patches/r05_gemm_k256_row_streaming_kernel.patch; lab library tools/micro/bin/libanemoi_amd_kstream.so) was seen
to return  rstd acc + b'  without the  (-mean rstd) s[n]  term in ~1 element of 10 000.  Where exactly?
   python tools/micro/run_with_lib.py tools/micro/bin/libanemoi_amd_kstream.so tools/micro/kstream_lnfold_repro.py
For every element: which of the two candidate values (with / without the term) the kernel's bf16 result is nearer to; the
"without" elements by row, column, column mod 32 (the lane's 16-lane row fq = (c mod 32) / 8 and element parity), row mod 32,
workgroup / wave of the row block; and whether the set repeats from run to run."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from anemoi_models_amd import ops  # noqa: E402

DEV = "cuda"
for m, n in ((70001, 2048), (65536, 2240), (70001, 512), (131072, 1024)):
    g = torch.Generator().manual_seed(m + n)
    k = 256
    x = torch.randn(m, k, generator=g).bfloat16().to(DEV)
    w = (torch.randn(n, k, generator=g) / k**0.5).bfloat16().to(DEV)
    b = torch.randn(n, generator=g).to(DEV)
    stats = torch.stack([torch.rand(m, generator=g) + 0.5, torch.randn(m, generator=g) * 0.3], 1).contiguous().to(DEV)
    colsum = torch.randn(n, generator=g).to(DEV)
    acc = x.float() @ w.float().t()
    with_term = acc * stats[:, :1] + stats[:, 1:] * colsum[None, :] + b
    without = acc * stats[:, :1] + b
    sets = []
    for rep in range(3):
        y = ops.linear(x, w, b, ln=(stats, colsum)).float()
        bad = ((y - without).abs() < (y - with_term).abs()) & ((with_term - without).abs() > 0.05)
        sets.append(bad)
    bad = sets[0]
    nb = int(bad.sum())
    print(f"M={m} N={n}: {nb} of {m * n} elements ({nb / (m * n):.2e}) are nearer to the value WITHOUT the term; the same set in "
          f"three runs: {bool(torch.equal(sets[0], sets[1]) and torch.equal(sets[1], sets[2]))}; max |y - with| elsewhere "
          f"{float(((y - with_term).abs() * (~bad)).max()):.3e}", flush=True)
    if nb == 0 or "--brief" in sys.argv:
        continue
    r, c = torch.nonzero(bad, as_tuple=True)
    r, c = r.cpu(), c.cpu()
    print("   column mod 32 histogram:", torch.bincount(c % 32, minlength=32).tolist())
    print("   column // 256 (column tile) histogram:", torch.bincount(c // 256).tolist())
    print("   row mod 32 histogram:", torch.bincount(r % 32, minlength=32).tolist())
    blocks = torch.unique(r // 32)
    print(f"   {blocks.numel()} row blocks of 32 affected of {(m + 31) // 32}; first {blocks[:12].tolist()}, last {blocks[-6:].tolist()}")
    print("   (row block) mod 8 [wave of the block]:", torch.bincount(blocks % 8, minlength=8).tolist())
    per_block = torch.bincount(r // 32)
    print("   elements per affected block: min / max", int(per_block[per_block > 0].min()), int(per_block.max()))
    sel = (r // 32) == blocks[0]
    print("   first affected block: rows", sorted(set(r[sel].tolist()))[:16], "columns", sorted(set(c[sel].tolist()))[:24])
    # ANY element that is off by more than bf16 rounding (not only the "without" ones): which st.y would produce it?
    off = (y - with_term).abs() > 0.04 + 0.01 * with_term.abs()
    ro, co = torch.nonzero(off, as_tuple=True)
    print(f"   {int(off.sum())} elements off by more than rounding; column mod 32 histogram {torch.bincount(co.cpu() % 32, minlength=32).tolist()}")
    print(f"   (column mod 256) // 32 histogram {torch.bincount((co.cpu() % 256) // 32, minlength=8).tolist()};  row mod 32 < 16: {int((ro % 32 < 16).sum())}, >= 16: {int((ro % 32 >= 16).sum())}")
    st_y = stats[:, 1]
    hits = {"zero": 0, "own": 0, "other row, same lane row (r mod 16)": 0, "none": 0}
    shown = 0
    for i in torch.randperm(ro.numel())[:400].tolist():
        rr, cc = int(ro[i]), int(co[i])
        x_implied = float((y[rr, cc] - acc[rr, cc] * stats[rr, 0] - b[cc]) / colsum[cc])
        tol = 0.02 + 0.02 * abs(x_implied) + 0.03 / abs(float(colsum[cc]))
        cand = st_y[rr % 16::16]
        j = int((cand - x_implied).abs().argmin())
        if abs(x_implied) < tol:
            kind = "zero"
        elif abs(x_implied - float(st_y[rr])) < tol:
            kind = "own"
        elif float((cand[j] - x_implied).abs()) < tol / 4:
            kind = "other row, same lane row (r mod 16)"
        else:
            kind = "none"
        hits[kind] += 1
        if shown < 8:
            shown += 1
            print(f"      row {rr} col {cc}: y {float(y[rr, cc]):.4f} with {float(with_term[rr, cc]):.4f} without {float(without[rr, cc]):.4f}; "
                  f"implied st.y {x_implied:.4f}, own {float(st_y[rr]):.4f}, nearest same-lane-row value {float(cand[j]):.4f} (row {rr % 16 + 16 * j}, "
                  f"{(rr % 16 + 16 * j - rr) // 32} blocks away) -> {kind}")
    print("   implied st.y of 400 sampled wrong elements:", hits)
