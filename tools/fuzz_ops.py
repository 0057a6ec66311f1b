#!/usr/bin/env python
"""Differential fuzzing of the kernels behind anemoi_models_amd.ops against plain torch: random shapes (ragged row / column
tiles, remainder rounds, K slab counts), random epilogue combinations, random graphs.  Complements the fixed-shape parity
tests; prints every case that exceeds its tolerance.   python tools/fuzz_ops.py [cases per op] [seed]"""
import os
import random
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anemoi_models_amd import ops, runtime  # noqa: E402

dev = "cuda"
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = random.Random(seed)
ACT = {"Identity": lambda t: t, "GELU": F.gelu, "SiLU": F.silu, "ReLU": F.relu}


def rel(a, b):
    return float((a.float() - b.float()).abs().max() / b.float().abs().max().clamp_min(1e-6))


def fuzz_linear():
    bad = 0
    for case in range(n_cases):
        dtype = torch.bfloat16 if rng.random() < 0.8 else torch.float32
        km = 64 if dtype == torch.bfloat16 else 32
        m = rng.choice([rng.randint(1, 300), rng.randint(1000, 3000), 256 * rng.randint(4, 12), 256 * rng.randint(4, 12) + rng.randint(1, 9),
                        5121, 1024, 2048 + rng.randint(1, 255)])
        n = rng.choice([8 * rng.randint(1, 40), 256 * rng.randint(1, 9), 256 * rng.randint(1, 8) + 8 * rng.randint(1, 31), 80, 192, 1216])
        k = km * rng.randint(1, 24)
        act = rng.choice(list(ACT))
        use_bias, use_res = rng.random() < 0.7, rng.random() < 0.4
        use_ln = rng.random() < 0.3 and dtype == torch.bfloat16
        want_stats = rng.random() < 0.3 and act == "Identity"
        g = torch.Generator().manual_seed(seed * 100003 + case)
        x = torch.randn(m, k, generator=g).to(dtype)
        w = (torch.randn(n, k, generator=g) / k**0.5).to(dtype)
        b = torch.randn(n, generator=g) if use_bias else None
        r = torch.randn(m, n, generator=g).to(dtype) if use_res else None
        xd, wd = x.to(dev), w.to(dev)
        kw = dict(act=act, residual=None if r is None else r.to(dev))
        pre = x.float() @ w.float().t()
        if use_ln:
            stats = torch.stack([torch.rand(m, generator=g) + 0.5, torch.randn(m, generator=g)], 1)
            cs = torch.randn(n, generator=g)
            pre = pre * stats[:, :1] + stats[:, 1:] * cs[None, :]
            kw["ln"] = (stats.to(dev), cs.to(dev))
        if b is not None:
            pre = pre + b
        want = ACT[act](pre)
        if r is not None:
            want = want + r.float()
        if want_stats:
            kw["stats_eps"] = 1e-5
        try:
            got = ops.linear(xd, wd, None if b is None else b.to(dev), **kw)
            torch.cuda.synchronize()
        except (NotImplementedError, ValueError) as e:  # shapes the entry points refuse are fine; wrong answers are not
            print(f"  linear case {case}: refused ({type(e).__name__}: {str(e)[:80]}) m={m} n={n} k={k} {dtype}")
            continue
        tol = 2e-2 if dtype == torch.bfloat16 else 2e-4
        err = rel(got.cpu(), want)
        ok = err < tol
        if want_stats and ok:
            st = ops.row_stats(got, 1e-5).cpu()
            gf = got.float().cpu()
            rstd = torch.rsqrt(gf.var(dim=1, unbiased=False) + 1e-5)
            ok = rel(st[:, 0], rstd) < 5e-3 and float((st[:, 1] + gf.mean(dim=1) * rstd).abs().max()) < 5e-3 * max(1.0, float((gf.mean(1) * rstd).abs().max()))
        if not ok:
            bad += 1
            print(f"  linear case {case}: err {err:.3e} m={m} n={n} k={k} {dtype} act={act} bias={use_bias} res={use_res} "
                  f"ln={use_ln} stats={want_stats}", flush=True)
    print(f"linear: {bad} bad of {n_cases}", flush=True)


def fuzz_edge_attention():
    from oracle import reference_path as ref

    bad = 0
    for case in range(n_cases // 3):
        dtype = torch.bfloat16 if rng.random() < 0.6 else torch.float32
        h = rng.choice([4, 8, 16])
        d = rng.choice([8, 16, 32, 64]) if dtype == torch.bfloat16 else rng.choice([4, 8, 16, 32, 64])
        c = h * d
        n_src, n_dst = rng.randint(1, 400), rng.randint(1, 400)
        e = rng.choice([0, rng.randint(1, 50), rng.randint(200, 4000)])
        edge_dim = rng.choice([3, 7, 11, 15])
        g = torch.Generator().manual_seed(seed * 7919 + case)
        ei = torch.stack([torch.randint(0, n_src, (e,), generator=g), torch.randint(0, n_dst, (e,), generator=g)])
        if e > 100:
            ei[1, : e // 4] = rng.randrange(n_dst)  # one high in-degree destination
        q, xr = (torch.randn(n_dst, c, generator=g).to(dtype) for _ in range(2))
        k, v = (torch.randn(n_src, c, generator=g).to(dtype) for _ in range(2))
        attr = torch.randn(e, edge_dim, generator=g)
        we, be = torch.randn(c, edge_dim, generator=g) * 0.3, torch.randn(c, generator=g) * 0.1
        edges = (attr @ we.t() + be).view(e, h, d)
        want = ref.gt_conv(q.float().view(n_dst, h, d), k.float().view(n_src, h, d), v.float().view(n_src, h, d), edges, ei,
                           n_dst).reshape(n_dst, c) + xr.float()
        plan = runtime.build_edge_plan(ei.to(dev), n_src, n_dst)
        ea = ops.edge_attr_csr(attr.to(dev), None, plan.perm)
        kv = torch.cat([k, v], 1).to(dev)
        got = ops.gt_edge_attention(q.to(dev), kv[:, :c], kv[:, c:], xr.to(dev), ea, edge_dim, we.to(dev), be.to(dev), plan.rowptr,
                                    plan.col, h)
        err = rel(got.cpu(), want)
        if not err < (3e-2 if dtype == torch.bfloat16 else 2e-4):
            bad += 1
            print(f"  edge attention case {case}: err {err:.3e} {dtype} n_src={n_src} n_dst={n_dst} e={e} h={h} d={d} edge_dim={edge_dim}",
                  flush=True)
    print(f"gt_edge_attention: {bad} bad of {n_cases // 3}", flush=True)


def fuzz_edge_scheduled():
    """The scheduled folded kernel (static destination schedule, scalar index pipeline, buffer loads) against the round-robin
    folded kernel on the same operands: BIT identity (outputs incl. the t columns, lse) over random graphs -- destination
    counts below / around / above the launch-shape thresholds, in-degrees 0 ... 40 incl. empty destinations, head sizes 64 / 32,
    every folded width, with and without x_r, strided operands (column ranges of one GEMM result)."""
    bad = tried = 0
    for case in range(n_cases // 3):
        h = rng.choice([8, 16])
        d = rng.choice([32, 64])
        c = h * d
        up = rng.choice([4, 8, 12, 16])
        n_dst = rng.choice([rng.randint(1, 40), rng.randint(41, 3000), rng.randint(3000, 30000)])
        n_src = rng.choice([rng.randint(1, 50), rng.randint(51, 5000)])
        g = torch.Generator().manual_seed(seed * 104729 + case)
        deg = torch.randint(0, rng.choice([4, 9, 13]), (n_dst,), generator=g)
        if n_dst > 10:
            deg[rng.randrange(n_dst)] = rng.choice([17, 40])
        dst = torch.repeat_interleave(torch.arange(n_dst), deg)
        e = int(dst.shape[0])
        src = torch.randint(0, n_src, (e,), generator=g)
        perm = torch.randperm(e, generator=g)
        plan = runtime.build_edge_plan(torch.stack([src[perm], dst[perm]]).to(dev), n_src, n_dst)
        sched = plan.schedule(torch.bfloat16, c)
        if sched is None:
            continue
        tried += 1
        wide = torch.randn(n_dst, 2 * c + h * up, generator=g).bfloat16().to(dev)  # x_r | q | u as one GEMM result
        kv = torch.randn(n_src, 2 * c, generator=g).bfloat16().to(dev)
        attr = torch.randn(max(e, 0), up, generator=g).to(dev)
        xr = wide[:, :c] if rng.random() < 0.7 else None
        outs = []
        for sc in (None, sched):
            lse = torch.full((n_dst, h), float("nan"), device=dev)
            out = ops.gt_edge_attention_folded(wide[:, c:2 * c], kv[:, :c], kv[:, c:], xr, wide[:, 2 * c:], attr, plan.rowptr,
                                               plan.col, h, up, lse=lse, sched=sc)
            outs.append((out, lse))
        same = torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
        if not same or not bool(torch.isfinite(outs[1][0].float()).all()):
            bad += 1
            print(f"  scheduled edge kernel case {case}: n_src={n_src} n_dst={n_dst} e={e} h={h} d={d} up={up} x_r={xr is not None}",
                  flush=True)
    print(f"gt_edge_attention_folded scheduled vs round-robin (bit identity): {bad} bad of {tried}", flush=True)


def fuzz_edge_groups():
    """The group kernel of uniform-degree-3 graphs (all destinations of a source triple behind one gather; scalar index chain,
    buffer loads) against the run kernel where the plan has runs (BIT identity: same per-destination arithmetic) and against
    the plain folded kernel (f32 rounding of another summation order): random triangle assignments -- scattered, clustered,
    groups longer than the cap of eight --, head sizes 64 / 32, every folded width, with and without x_r, strided operands."""
    bad = tried = with_runs = 0
    for case in range(n_cases // 3):
        h = rng.choice([8, 16])
        d = rng.choice([32, 64])
        c = h * d
        up = rng.choice([4, 8, 12, 16])
        n_dst = rng.choice([rng.randint(1024, 3000), rng.randint(3000, 40000)])
        n_src = rng.randint(40, 6000)
        g = torch.Generator().manual_seed(seed * 7919 + case)
        n_tri = max(1, n_dst // rng.choice([2, 5, 12]))
        base = torch.randint(0, max(1, n_src - 33), (n_tri,), generator=g)
        tri_of = torch.randint(0, n_tri, (n_dst,), generator=g)
        if rng.random() < 0.5:  # clustered: consecutive destinations often share their triangle (the run kernel's case)
            keep = torch.rand(n_dst, generator=g) < 0.6
            for i in range(1, n_dst):
                if keep[i]:
                    tri_of[i] = tri_of[i - 1]
        tri = torch.stack([base, base + 7, base + 31], 1)[tri_of]
        order = torch.stack([torch.randperm(3, generator=g) for _ in range(n_dst)])
        src = torch.gather(tri, 1, order).reshape(-1)
        dst = torch.arange(n_dst).repeat_interleave(3)
        plan = runtime.build_edge_plan(torch.stack([src, dst]).to(dev), n_src, n_dst)
        groups = runtime._groups3(plan)
        if groups is None:
            continue
        tried += 1
        runs = runtime._runs3(plan)
        wide = torch.randn(n_dst, 2 * c + h * up, generator=g).bfloat16().to(dev)  # x_r | q | u as one GEMM result
        kv = torch.randn(n_src, 2 * c, generator=g).bfloat16().to(dev)
        attr = torch.randn(3 * n_dst, up, generator=g).to(dev)
        xr = wide[:, :c] if rng.random() < 0.7 else None
        res = {}
        for name, lists in (("plain", None), ("groups", groups), ("runs", runs)):
            if name == "runs" and runs is None:
                continue
            lse = torch.full((n_dst, h), float("nan"), device=dev)
            out = ops.gt_edge_attention_folded(wide[:, c:2 * c], kv[:, :c], kv[:, c:], xr, wide[:, 2 * c:], attr, plan.rowptr,
                                               plan.col, h, up, lse=lse, runs=lists)
            res[name] = (out, lse)
        ok = bool(torch.isfinite(res["groups"][0].float()).all()) and rel(res["groups"][0], res["plain"][0].float()) < 1e-2 \
            and rel(res["groups"][1], res["plain"][1]) < 1e-5
        if "runs" in res:
            with_runs += 1
            ok = ok and torch.equal(res["groups"][0], res["runs"][0]) and torch.equal(res["groups"][1], res["runs"][1])
        if not ok:
            bad += 1
            print(f"  group edge kernel case {case}: n_src={n_src} n_dst={n_dst} h={h} d={d} up={up} x_r={xr is not None} "
                  f"groups={groups[0].shape[0] - 1} runs={'-' if runs is None else runs[0].shape[0] - 1}", flush=True)
    print(f"gt_edge_attention_folded groups vs plain (and bit identity with the run kernel in {with_runs} cases): {bad} bad of {tried}",
          flush=True)


def fuzz_rows():
    bad = 0
    for case in range(n_cases // 3):
        dtype = torch.bfloat16 if rng.random() < 0.5 else torch.float32
        rows, c = rng.randint(1, 5000), rng.choice([64, 128, 192, 512, 1024, 1216])
        g = torch.Generator().manual_seed(seed * 31 + case)
        x = (torch.randn(rows, c, generator=g) * 2 + 0.3).to(dtype)
        gm, bt = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.1
        want = F.layer_norm(x.float(), (c,), gm, bt, 1e-5)
        got = ops.layer_norm(x.to(dev), gm.to(dev), bt.to(dev), 1e-5)
        tr = ops.transpose(x.to(dev), ops.round_up(rows, 64))
        cs = ops.col_sum(x.to(dev)).cpu()
        ok = rel(got.cpu(), want) < (2e-2 if dtype == torch.bfloat16 else 1e-5) and torch.equal(tr[:, :rows].cpu(), x.t()) and \
            not bool(tr[:, rows:].any()) and rel(cs, x.double().sum(0).float()) < 1e-4
        if not ok:
            bad += 1
            print(f"  row kernels case {case}: {dtype} rows={rows} c={c}", flush=True)
    print(f"layer_norm / transpose / col_sum: {bad} bad of {n_cases // 3}", flush=True)


def fuzz_weight_grad():
    """dW = dpre^T x and db on the TN kernel: random row counts (ragged chunks), column counts (ragged 256-tiles, 1-6 x-column
    tiles = every share-out of the bias fragments), pitches wider than the matrices, against f64."""
    bad = 0
    for case in range(n_cases // 2):
        m = rng.choice([rng.randint(128, 700), rng.randint(2000, 9000), 2048 * rng.randint(2, 6) + rng.randint(0, 70), 40962])
        n = 8 * rng.choice([rng.randint(1, 40), 32 * rng.randint(1, 5), 32 * rng.randint(1, 4) + rng.randint(1, 31)])
        k = 8 * rng.choice([rng.randint(1, 40), 32 * rng.randint(1, 6), 32 * rng.randint(1, 5) + rng.randint(1, 31)])
        pad_d, pad_x = 8 * rng.randint(0, 3), 8 * rng.randint(0, 3)
        g = torch.Generator().manual_seed(seed * 7919 + case)
        dfull = torch.randn(m, n + pad_d, generator=g).bfloat16().to(dev)
        xfull = torch.randn(m, k + pad_x, generator=g).bfloat16().to(dev)
        dpre, x = dfull[:, pad_d:], xfull[:, :k]
        dw, db = ops.weight_grad(dpre, x, k, want_bias=True)
        want = dpre.double().t() @ x.double()
        want_b = dpre.double().sum(0)
        ok = rel(dw, want.float()) < 3e-5 and float((db.double() - want_b).abs().max()) < 1e-5 * float(
            dpre.double().abs().sum(0).max()) and torch.equal(dw, ops.weight_grad(dpre, x, k))
        if not ok:
            bad += 1
            print(f"  weight_grad case {case}: m={m} n={n} k={k} pads {pad_d} {pad_x}: dW {rel(dw, want.float()):.2e}", flush=True)
    print(f"weight_grad (TN kernel): {bad} bad of {n_cases // 2}", flush=True)


def fuzz_conv_dropout():
    """anemoi_gt_conv and its two backward kernels with the conv's dropout (ABI v41): random graphs (isolated destinations,
    hubs), head sizes, probabilities; output and all four input gradients against torch autograd in f64 with the same mask."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    from _cpu_ops import edge_dropout_keep_mask
    from anemoi_models_amd import autograd

    bad = 0
    for case in range(n_cases // 2):
        dtype = torch.bfloat16 if rng.random() < 0.6 else torch.float32
        h = rng.choice([1, 2, 4, 8, 16])
        d = rng.choice([4, 5, 8, 12, 16, 20, 24, 32, 48, 64])  # (sizes outside the kernels' lane groups: zero-padded heads)
        c = h * d
        n_src, n_dst, e = rng.randint(1, 400), rng.randint(2, 400), rng.randint(0, 3000)
        p = rng.choice([0.0, 0.1, 0.5, rng.random() * 0.95])
        seed_c = rng.randint(0, 2**31 - 1)
        g = torch.Generator().manual_seed(seed * 100003 + case)
        ei = torch.stack([torch.randint(0, n_src, (e,), generator=g), torch.randint(0, n_dst - 1, (e,), generator=g)])
        if e > 60:
            ei[1, :50] = 0  # a hub
        plan = runtime.PlanCache().get(ei.to(dev), n_src, n_dst)
        q = torch.randn(n_dst, c, generator=g).to(dtype)
        kv = torch.randn(n_src, 2 * c, generator=g).to(dtype)
        ed = torch.randn(e, c, generator=g).to(dtype)  # CSR order
        w = torch.randn(n_dst, c, generator=g)
        perm = plan.perm.long().cpu()
        src, dst = ei[0][perm], ei[1][perm]  # the plan's (destination-sorted) order
        keep = edge_dropout_keep_mask(seed_c, p, e, h)
        refs = [t.double().requires_grad_() for t in (q, kv, ed)]
        qr, kvr, er = refs
        kj = kvr[:, :c].reshape(n_src, h, d)[src] + er.reshape(e, h, d)
        vj = kvr[:, c:].reshape(n_src, h, d)[src] + er.reshape(e, h, d)
        sc = (qr.reshape(n_dst, h, d)[dst] * kj).sum(-1) / d**0.5
        mx = torch.full((n_dst, h), float("-inf"), dtype=torch.float64).scatter_reduce(0, dst[:, None].expand(-1, h), sc.detach(), "amax")
        ex = torch.exp(sc - mx[dst])
        den = torch.zeros(n_dst, h, dtype=torch.float64).index_add(0, dst, ex) + 1e-16
        alpha = ex / den[dst] * keep * (1.0 / (1.0 - p))
        want = torch.zeros(n_dst, h, d, dtype=torch.float64).index_add(0, dst, vj * alpha[..., None]).reshape(n_dst, c)
        (want * w.double()).sum().backward()
        leaves = [t.to(dev).requires_grad_() for t in (q, kv, ed)]
        out = autograd.gt_conv(leaves[0], leaves[1][:, :c], leaves[1][:, c:], leaves[2], None, plan, h, p, seed_c)
        (out.float() * w.to(dev)).sum().backward()
        tol_o, tol_g = (1e-5, 2e-4) if dtype == torch.float32 else (2e-2, 5e-2)
        # (a gradient that is zero in exact arithmetic -- dq when no destination has two edges: ds = w - alpha dsum = 0 -- is
        #  held against the scale of the step's gradients, not against its own rounding noise)
        g_scale = max(float(b.grad.abs().max()) for b in refs) if e > 0 else 1.0
        errs = [rel(out.detach().cpu(), want.detach())] + [
            float((a.grad.cpu().double() - b.grad).abs().max() / max(float(b.grad.abs().max()), 2e-2 * g_scale, 1e-6))
            if e > 0 else 0.0 for a, b in zip(leaves, refs)]
        if errs[0] > tol_o or max(errs[1:]) > tol_g or not torch.isfinite(out).all():
            bad += 1
            print(f"  conv dropout case {case}: {dtype} n_src={n_src} n_dst={n_dst} e={e} h={h} d={d} p={p:.3f}: "
                  f"out {errs[0]:.2e}, dq {errs[1]:.2e}, dkv {errs[2]:.2e}, de {errs[3]:.2e}", flush=True)
    print(f"gt_conv with dropout, forward + backward: {bad} bad of {n_cases // 2}", flush=True)


def fuzz_mhsa():
    """anemoi_mhsa at random sequence lengths around the tile borders (512-query blocks, 32-key tiles, the key-split tail
    rows), batch sizes, head counts, windows: forward against softmax(Q K^T / sqrt(D)) V in f64."""
    bad = 0
    for case in range(n_cases // 3):
        d = rng.choice([64, 64, 32, 16, 8, 48])
        dtype = torch.bfloat16 if (d in (64, 32) or rng.random() < 0.5) else torch.float32
        h, b = rng.choice([1, 2, 4]), rng.choice([1, 1, 2])
        s_len = rng.choice([rng.randint(1, 200), 512 * rng.randint(1, 4) + rng.choice([-1, 0, 1, 2, 33]), rng.randint(200, 2600)])
        window = rng.choice([-1, -1, rng.randint(0, 300)])
        c = h * d
        g = torch.Generator().manual_seed(seed * 100003 + case)
        qkv = (torch.randn(b * s_len, 3 * c, generator=g) * 0.7).to(dtype)
        q, k, v = (t.double().reshape(b, s_len, h, d).permute(0, 2, 1, 3) for t in qkv.split(c, dim=1))
        sc = q @ k.transpose(-1, -2) / d**0.5
        if window >= 0:
            i = torch.arange(s_len)
            sc = sc.masked_fill((i[:, None] - i[None, :]).abs() > window, float("-inf"))
        want = (torch.softmax(sc, -1) @ v).permute(0, 2, 1, 3).reshape(b * s_len, c)
        got = ops.mhsa(qkv.to(dev), b, h, window)
        err = rel(got.cpu(), want)
        if err > (2e-5 if dtype == torch.float32 else 2e-2) or not torch.isfinite(got).all():
            bad += 1
            print(f"  mhsa case {case}: {dtype} b={b} s={s_len} h={h} d={d} window={window}: {err:.2e}", flush=True)
    print(f"mhsa forward: {bad} bad of {n_cases // 3}", flush=True)


def _module_vs_oracle(what, blk_cpu, run_dev, run_ref, tol):
    """Shared by the block fuzzers: forward + backward of a module on the device against the oracle under torch autograd (f64),
    output, input gradients, parameter gradients.  ``run_ref(rsd)`` -> (output, [input leaves]); ``run_dev(blk)`` the same."""
    rsd = {"x." + k: v.detach().double().requires_grad_() for k, v in blk_cpu.named_parameters()}
    want, ins_r = run_ref(rsd)
    blk = blk_cpu.to(dev)
    if rng.random() < 0.3:  # the inference route of the same module (eval mode, no autograd graph): output only
        with torch.no_grad():
            y, _ = run_dev(blk.eval())
        err = rel(y.cpu(), want.detach())
        if err > tol or not torch.isfinite(y).all():
            print(f"  {what} [eval]: out {err:.2e}", flush=True)
            return 1
        return 0
    y, ins = run_dev(blk)
    w = torch.randn(want.shape, generator=torch.Generator().manual_seed(1))
    (want * w.double()).sum().backward()
    (y.float() * w.to(dev)).sum().backward()
    errs = {"out": rel(y.detach().cpu(), want.detach())}
    pairs = {("in%d" % i): (a.grad, b.grad) for i, (a, b) in enumerate(zip(ins, ins_r))}
    pairs.update({k: (p.grad, rsd["x." + k].grad) for k, p in blk.named_parameters()})
    pairs = {k: ab for k, ab in pairs.items() if ab[1] is not None and ab[1].numel() > 0}
    g_scale = max(float(b.abs().max()) for _, b in pairs.values())
    for k, (a, b) in pairs.items():
        if a is None:
            errs[k] = float("inf") if float(b.abs().max()) > 1e-9 * g_scale else 0.0
        else:
            errs[k] = float((a.detach().cpu().double() - b).abs().max() / max(float(b.abs().max()), 2e-2 * g_scale, 1e-12))
    worst = max(errs, key=errs.get)
    if errs[worst] > tol or not torch.isfinite(y).all():
        print(f"  {what}: worst {worst} {errs[worst]:.2e} (out {errs['out']:.2e})", flush=True)
        return 1
    return 0


def fuzz_gt_blocks_training():
    """GraphTransformerProcessorBlock / MapperBlock as modules, forward + backward (and, for a third of the cases, the
    inference route), random widths, head counts, edge widths and graphs (no edges at all, isolated destinations, hubs)
    against the oracle under torch autograd in f64: output, input gradients, every parameter gradient.  Exercises the
    per-shape choice between the folded edge kernels and the explicit conv."""
    from anemoi_models_amd.layers.block import GraphTransformerMapperBlock, GraphTransformerProcessorBlock
    from oracle import reference_path as ref

    bad = 0
    n_run = n_cases // 3
    for case in range(n_run):
        bf16 = rng.random() < 0.5
        os.environ["ANEMOI_AMD_DTYPE"] = "bf16" if bf16 else "fp32"
        c = (64 if bf16 else 32) * rng.randint(1, 4)
        h = rng.choice([hh for hh in (1, 2, 4, 8, 16, 32) if c % hh == 0 and c // hh <= (128 if bf16 else 64)])
        edge_dim = rng.choice([1, 3, 4, 7, 11, 15, 16, 23])
        n, n_src = rng.randint(2, 300), rng.randint(1, 300)
        e = rng.choice([0, rng.randint(1, 40), rng.randint(100, 2500)])
        mapper = rng.random() < 0.5
        g = torch.Generator().manual_seed(seed * 100003 + case)
        torch.manual_seed(seed * 7 + case)
        tol = 8e-2 if bf16 else 2e-3
        what = f"{'mapper' if mapper else 'processor'} block C={c} H={h} edge_dim={edge_dim} n={n} n_src={n_src} e={e} {'bf16' if bf16 else 'f32'}"
        try:
            ea0 = torch.randn(e, edge_dim, generator=g)
            if mapper:
                blk = GraphTransformerMapperBlock(c, 2 * c, c, edge_dim=edge_dim, num_heads=h)
                ei = torch.stack([torch.randint(0, n_src, (e,), generator=g), torch.randint(0, max(n - 1, 1), (e,), generator=g)])
                xs0, xd0 = torch.randn(n_src, c, generator=g), torch.randn(n, c, generator=g)

                def run_ref(rsd):
                    ins_r = [xs0.double().requires_grad_(), xd0.double().requires_grad_(), ea0.double().requires_grad_()]
                    return ref.gt_mapper_block(rsd, "x", ins_r[0], ins_r[1], ins_r[2], ei, h), ins_r

                def run_dev(m):
                    ins = [t.to(dev).requires_grad_() for t in (xs0, xd0, ea0)]
                    (_, y), _ = m((ins[0], ins[1]), ins[2], ei.to(dev), None, 1, size=(n_src, n))
                    return y, ins
            else:
                blk = GraphTransformerProcessorBlock(c, 2 * c, c, edge_dim=edge_dim, num_heads=h)
                ei = torch.stack([torch.randint(0, n, (e,), generator=g), torch.randint(0, max(n - 1, 1), (e,), generator=g)])
                x0 = torch.randn(n, c, generator=g)

                def run_ref(rsd):
                    ins_r = [x0.double().requires_grad_(), ea0.double().requires_grad_()]
                    return ref.gt_processor_block(rsd, "x", ins_r[0], ins_r[1], ei, h), ins_r

                def run_dev(m):
                    ins = [t.to(dev).requires_grad_() for t in (x0, ea0)]
                    y, _ = m(ins[0], ins[1], ei.to(dev), None, 1)
                    return y, ins
            bad += _module_vs_oracle(what, blk, run_dev, run_ref, tol)
        except Exception as exc:  # noqa: BLE001
            bad += 1
            print(f"  {what}: {type(exc).__name__}: {str(exc).splitlines()[0][:160]}", flush=True)
    os.environ.pop("ANEMOI_AMD_DTYPE", None)
    print(f"GraphTransformer blocks, forward + backward against the oracle's autograd: {bad} bad of {n_run}", flush=True)


def fuzz_gnn_and_transformer_blocks():
    """GraphConvProcessorBlock / GraphConvMapperBlock (edge-MLP message passing) and TransformerProcessorBlock as modules,
    forward + backward against the oracle's autograd: random widths, extra MLP layers, graphs, sequence lengths, heads, windows."""
    from anemoi_models_amd.layers.block import GraphConvMapperBlock, GraphConvProcessorBlock, TransformerProcessorBlock
    from oracle import reference_path as ref

    bad, n_run = 0, n_cases // 3
    for case in range(n_run):
        bf16 = rng.random() < 0.5
        os.environ["ANEMOI_AMD_DTYPE"] = "bf16" if bf16 else "fp32"
        tol = 8e-2 if bf16 else 2e-3
        g = torch.Generator().manual_seed(seed * 100003 + case)
        torch.manual_seed(seed * 7 + case)
        kind = rng.choice(["gnn_proc", "gnn_map", "tfm"])
        c = (64 if bf16 else 32) * rng.randint(1, 4)
        try:
            if kind == "tfm":
                h = rng.choice([hh for hh in (1, 2, 4, 8, 16) if c % hh == 0 and c // hh <= 128])
                b, s_len = rng.choice([1, 1, 2]), rng.choice([rng.randint(1, 60), rng.randint(100, 700)])
                window = rng.choice([None, None, rng.randint(1, 200)])
                what = f"Transformer block C={c} H={h} B={b} S={s_len} window={window} {'bf16' if bf16 else 'f32'}"
                x0 = torch.randn(b * s_len, c, generator=g)
                blk = TransformerProcessorBlock(c, 2 * c, h, "GELU", window_size=window if window is not None else 512)
                if window is not None:
                    os.environ["ANEMOI_AMD_FLASH_WINDOW"] = "1"

                def run_ref(rsd):
                    xr = x0.double().requires_grad_()
                    return ref.transformer_block(rsd, "x", xr, b, h, "GELU", window), [xr]

                def run_dev(m):
                    x = x0.to(dev).requires_grad_()
                    return m(x, [list(x.shape)], b), [x]
            else:
                extra = rng.choice([0, 0, 1])
                n, n_src, e = rng.randint(2, 300), rng.randint(1, 300), rng.choice([0, rng.randint(1, 40), rng.randint(100, 2500)])
                ea0 = torch.randn(e, c, generator=g)
                if kind == "gnn_proc":
                    what = f"GNN processor block C={c} extra={extra} n={n} e={e} {'bf16' if bf16 else 'f32'}"
                    ei = torch.stack([torch.randint(0, n, (e,), generator=g), torch.randint(0, max(n - 1, 1), (e,), generator=g)])
                    x0 = torch.randn(n, c, generator=g)
                    blk = GraphConvProcessorBlock(c, c, mlp_extra_layers=extra)

                    def run_ref(rsd):
                        xr, er = x0.double().requires_grad_(), ea0.double().requires_grad_()
                        out, edges = ref.gnn_processor_block(rsd, "x", xr, er, ei, "SiLU", extra)
                        return torch.cat([out, edges]), [xr, er]

                    def run_dev(m):
                        x, ea = x0.to(dev).requires_grad_(), ea0.to(dev).requires_grad_()
                        out, edges = m(x, ea, ei.to(dev), (None, None, None))
                        return torch.cat([out, edges]), [x, ea]
                else:
                    upd = rng.random() < 0.5
                    what = f"GNN mapper block C={c} extra={extra} n={n} n_src={n_src} e={e} update_src={upd} {'bf16' if bf16 else 'f32'}"
                    ei = torch.stack([torch.randint(0, n_src, (e,), generator=g), torch.randint(0, max(n - 1, 1), (e,), generator=g)])
                    xs0, xd0 = torch.randn(n_src, c, generator=g), torch.randn(n, c, generator=g)
                    blk = GraphConvMapperBlock(c, c, mlp_extra_layers=extra, update_src_nodes=upd)

                    def run_ref(rsd):
                        xs, xd, er = xs0.double().requires_grad_(), xd0.double().requires_grad_(), ea0.double().requires_grad_()
                        (ns, nd), edges = ref.gnn_mapper_block(rsd, "x", xs, xd, er, ei, upd, "SiLU", extra)
                        return torch.cat([ns, nd, edges]), [xs, xd, er]

                    def run_dev(m):
                        xs, xd, ea = (t.to(dev).requires_grad_() for t in (xs0, xd0, ea0))
                        (ns, nd), edges = m((xs, xd), ea, ei.to(dev), (None, None, None), size=(n_src, n))
                        return torch.cat([ns, nd, edges]), [xs, xd, ea]
            bad += _module_vs_oracle(what, blk, run_dev, run_ref, tol)
        except Exception as exc:  # noqa: BLE001
            bad += 1
            print(f"  {what}: {type(exc).__name__}: {str(exc).splitlines()[0][:160]}", flush=True)
        finally:
            os.environ.pop("ANEMOI_AMD_FLASH_WINDOW", None)
    os.environ.pop("ANEMOI_AMD_DTYPE", None)
    print(f"GNN / Transformer blocks, forward + backward against the oracle's autograd: {bad} bad of {n_run}", flush=True)


def fuzz_models():
    """AnemoiModelEncProcDec, whole forward (and backward for half of the cases) against the oracle: random processor / mapper
    families, widths, heads, batch sizes, multistep inputs, variable counts, trainable sizes, chunkings on the O32 graph."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    from anemoi_models_amd.graphs.synthetic import build_graph
    from anemoi_models_amd.models import AnemoiModelEncProcDec
    from anemoi_models_amd.utils.indices import SimpleDataIndices
    from anemoi_models_amd.utils.presets import model_config
    from oracle import reference_path as ref
    from test_oracle_golden import graph_tensors

    graph = build_graph("o32_ico2")
    gt = graph_tensors(graph)
    n_grid = graph["data"].num_nodes
    bad, n_run = 0, max(n_cases // 12, 4)
    for case in range(n_run):
        bf16 = rng.random() < 0.5
        os.environ["ANEMOI_AMD_DTYPE"] = "bf16" if bf16 else "fp32"
        proc = rng.choice(["GraphTransformer", "GNN", "Transformer"])
        maps = rng.choice(["GraphTransformer", "GraphTransformer", "GNN"])
        channels, heads = rng.choice([(64, 16), (128, 8), (64, 4), (128, 16), (192, 16)])
        b, multistep = rng.choice([(1, 2), (2, 2), (1, 1), (3, 1), (1, 3)])
        n_prog, n_forc, n_diag = rng.choice([(10, 2, 1), (5, 0, 0), (1, 0, 3), (26, 6, 1), (7, 3, 2)])
        trainable, layers, chunks = rng.choice([8, 8, 0, 3]), rng.choice([2, 4]), rng.choice([1, 2])
        train = rng.random() < 0.5
        what = f"model {proc} / {maps} mappers C={channels} H={heads} B={b} T={multistep} vars={n_prog}+{n_forc}+{n_diag} " \
               f"trainable={trainable} layers={layers} chunks={chunks} {'bf16' if bf16 else 'f32'} {'train' if train else 'eval'}"
        try:
            torch.manual_seed(seed * 11 + case)
            idx = SimpleDataIndices(n_prognostic=n_prog, n_forcing=n_forc, n_diagnostic=n_diag)
            cfg = model_config(proc, channels, layers, heads, multistep=multistep, trainable=trainable, proc_chunks=chunks,
                               window_size=512, mappers=maps)
            model = AnemoiModelEncProcDec(model_config=cfg, data_indices=idx, graph_data=graph)
            with torch.no_grad():
                for name, p in model.named_parameters():
                    if name.endswith("trainable"):
                        p.normal_(0.0, 0.1)
            for m in model.modules():
                if hasattr(m, "dropout_p"):
                    m.dropout_p = 0.0
            x0 = torch.randn(b, multistep, 1, n_grid, idx.num_input, generator=torch.Generator().manual_seed(case))
            sd = {k: (v.detach().double().requires_grad_() if v.is_floating_point() else v.detach())
                  for k, v in model.state_dict(keep_vars=True).items()}
            kw = dict(num_heads=heads, num_layers=layers, num_chunks=chunks, prognostic_in=list(range(n_prog)),
                      prognostic_out=list(range(n_prog)), processor=proc, mappers=maps)
            gt64 = {k: (v.double() if v.is_floating_point() else v) for k, v in gt.items()}
            want = ref.model_forward(sd, gt64, x0.double(), **kw)
            model = model.to(dev).train(train)
            tol_o, tol_g = (5e-2, 1e-1) if bf16 else (2e-4, 5e-3)
            with torch.enable_grad() if train else torch.no_grad():
                y = model(x0.to(dev))
            err = rel(y.detach().cpu(), want.detach())
            worst, worst_k = 0.0, ""
            if train:
                w = torch.randn(want.shape, generator=torch.Generator().manual_seed(2))
                (want * w.double()).sum().backward()
                (y.float() * w.to(dev)).sum().backward()
                grads = {k: v.grad for k, v in sd.items() if torch.is_tensor(v) and v.is_floating_point() and v.grad is not None}
                g_scale = max(float(g_.abs().max()) for g_ in grads.values())
                for k, p in model.named_parameters():
                    if k not in grads or float(grads[k].abs().max()) <= 1e-9 * g_scale:
                        continue
                    e_k = float("inf") if p.grad is None else float(
                        (p.grad.cpu().double() - grads[k]).abs().max() / max(float(grads[k].abs().max()), 2e-2 * g_scale))
                    if e_k > worst:
                        worst, worst_k = e_k, k
            if err > tol_o or worst > tol_g or not torch.isfinite(y).all():
                bad += 1
                print(f"  {what}: out {err:.2e}, worst gradient {worst:.2e} ({worst_k})", flush=True)
        except Exception as exc:  # noqa: BLE001
            bad += 1
            print(f"  {what}: {type(exc).__name__}: {str(exc).splitlines()[0][:160]}", flush=True)
    os.environ.pop("ANEMOI_AMD_DTYPE", None)
    print(f"whole models against the oracle (forward; backward for the training cases): {bad} bad of {n_run}", flush=True)


def fuzz_hierarchical():
    """AnemoiModelEncProcDecHierarchical against oracle.hierarchical_forward: 2 / 3 hidden levels, level processing on / off,
    widths, heads, batch sizes, multistep inputs, f32 / bf16; forward, and backward for half of the cases."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    from anemoi_models_amd.graphs.synthetic import build_hierarchical_graph
    from anemoi_models_amd.models import AnemoiModelEncProcDecHierarchical
    from anemoi_models_amd.utils.indices import SimpleDataIndices
    from anemoi_models_amd.utils.presets import hierarchical_model_config
    from oracle import reference_path as ref
    from test_oracle_golden import hier_graph_tensors

    graphs = {2: build_hierarchical_graph("o32", (2, 1)), 3: build_hierarchical_graph("o32", (3, 2, 1))}
    bad, n_run = 0, max(n_cases // 12, 4)
    for case in range(n_run):
        bf16 = rng.random() < 0.5
        os.environ["ANEMOI_AMD_DTYPE"] = "bf16" if bf16 else "fp32"
        levels, level_process = rng.choice([2, 3]), rng.random() < 0.7
        channels, heads = rng.choice([(64, 16), (64, 4), (128, 8)])
        b, multistep, level_layers, trainable = rng.choice([1, 2]), rng.choice([1, 2, 3]), rng.choice([1, 2]), rng.choice([8, 0, 3])
        train = rng.random() < 0.5
        hidden = [f"hidden_{i + 1}" for i in range(levels)]
        graph = graphs[levels]
        what = f"hierarchical levels={levels} level_process={level_process} C={channels} H={heads} B={b} T={multistep} " \
               f"layers={level_layers} trainable={trainable} {'bf16' if bf16 else 'f32'} {'train' if train else 'eval'}"
        try:
            torch.manual_seed(seed * 19 + case)
            idx = SimpleDataIndices(n_prognostic=10, n_forcing=2, n_diagnostic=1)
            cfg = hierarchical_model_config(channels, heads, hidden=hidden, level_layers=level_layers, level_process=level_process,
                                            multistep=multistep, trainable=trainable)
            model = AnemoiModelEncProcDecHierarchical(model_config=cfg, data_indices=idx, graph_data=graph)
            with torch.no_grad():
                for name, p in model.named_parameters():
                    if name.endswith("trainable"):
                        p.normal_(0.0, 0.1)
            x0 = torch.randn(b, multistep, 1, graph["data"].num_nodes, idx.num_input, generator=torch.Generator().manual_seed(case))
            sd = {k: (v.detach().double().requires_grad_() if v.is_floating_point() else v.detach())
                  for k, v in model.state_dict(keep_vars=True).items()}
            gt = {k: (v.double() if v.is_floating_point() else v) for k, v in hier_graph_tensors(graph, hidden).items()}
            want = ref.hierarchical_forward(sd, gt, x0.double(), hidden=hidden, num_heads=heads, level_layers=level_layers,
                                            prognostic_in=list(range(10)), prognostic_out=list(range(10)),
                                            level_process=level_process)
            model = model.to(dev).train(train)
            with torch.enable_grad() if train else torch.no_grad():
                y = model(x0.to(dev))
            err = rel(y.detach().cpu(), want.detach())
            worst, worst_k = 0.0, ""
            if train:
                w = torch.randn(want.shape, generator=torch.Generator().manual_seed(2))
                (want * w.double()).sum().backward()
                (y.float() * w.to(dev)).sum().backward()
                grads = {k: v.grad for k, v in sd.items() if torch.is_tensor(v) and v.is_floating_point() and v.grad is not None}
                g_scale = max(float(g_.abs().max()) for g_ in grads.values())
                for k, p in model.named_parameters():
                    if k not in grads or float(grads[k].abs().max()) <= 1e-9 * g_scale:
                        continue
                    e_k = float("inf") if p.grad is None else float(
                        (p.grad.cpu().double() - grads[k]).abs().max() / max(float(grads[k].abs().max()), 2e-2 * g_scale))
                    if e_k > worst:
                        worst, worst_k = e_k, k
            tol_o, tol_g = (5e-2, 1e-1) if bf16 else (2e-4, 5e-3)
            if err > tol_o or worst > tol_g or not torch.isfinite(y).all():
                bad += 1
                print(f"  {what}: out {err:.2e}, worst gradient {worst:.2e} ({worst_k})", flush=True)
        except Exception as exc:  # noqa: BLE001
            bad += 1
            print(f"  {what}: {type(exc).__name__}: {str(exc).splitlines()[0][:200]}", flush=True)
    os.environ.pop("ANEMOI_AMD_DTYPE", None)
    print(f"hierarchical models against the oracle (forward; backward for the training cases): {bad} bad of {n_run}", flush=True)


def fuzz_interface():
    """AnemoiModelInterface.predict_step (normalise -> model -> de-normalise; the second call rides on the fused input /
    output kernels) over random normaliser methods and statistics, families, batch sizes, multistep inputs, f32 / bf16, against
    the oracle's predict_step on the interface's own state dict."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    import numpy as np

    from anemoi_models_amd.graphs.synthetic import build_graph
    from anemoi_models_amd.interface import AnemoiModelInterface
    from anemoi_models_amd.utils.indices import SimpleDataIndices
    from anemoi_models_amd.utils.presets import model_config
    from oracle import reference_path as ref
    from test_oracle_golden import graph_tensors

    graph = build_graph("o32_ico2")
    gt = {k: (v.double() if v.is_floating_point() else v) for k, v in graph_tensors(graph).items()}
    n_grid = graph["data"].num_nodes
    bad, n_run = 0, max(n_cases // 12, 4)
    for case in range(n_run):
        bf16 = rng.random() < 0.5
        os.environ["ANEMOI_AMD_DTYPE"] = "bf16" if bf16 else "fp32"
        proc = rng.choice(["GraphTransformer", "GNN", "Transformer"])
        maps = rng.choice(["GraphTransformer", "GraphTransformer", "GNN"])
        n_prog, n_forc, n_diag = rng.choice([(10, 2, 1), (5, 1, 0), (3, 0, 2), (12, 4, 1)])
        b, multistep, layers = rng.choice([1, 2, 3]), rng.choice([1, 2, 3]), 2
        what = f"interface {proc} / {maps} mappers vars={n_prog}+{n_forc}+{n_diag} B={b} T={multistep} {'bf16' if bf16 else 'f32'}"
        try:
            torch.manual_seed(seed * 13 + case)
            idx = SimpleDataIndices(n_prognostic=n_prog, n_forcing=n_forc, n_diagnostic=n_diag)
            names = sorted(idx.name_to_index, key=idx.name_to_index.get)
            pool = [n for n in names]
            rng.shuffle(pool)
            methods = {"default": rng.choice(["mean-std", "mean-std", "min-max", "none"])}
            for m in ("min-max", "max", "std", "none"):
                take = [pool.pop() for _ in range(min(len(pool), rng.randint(0, 2)))]
                if take:
                    methods[m] = take
            cfg = model_config(proc, 64, layers, 16, multistep=multistep, proc_chunks=1, mappers=maps)
            cfg["data"] = {"forcing": [n for n in names if n.startswith("forc")], "diagnostic": [n for n in names if n.startswith("diag")],
                           "processors": {"normalizer": {"_target_": "anemoi.models.preprocessing.normalizer.InputNormalizer",
                                                         "config": methods}}}
            cfg["model"]["model"] = {"_target_": "anemoi.models.models.encoder_processor_decoder.AnemoiModelEncProcDec"}
            cfg = type(cfg)(cfg)
            nv = len(names)
            gs = np.random.default_rng(seed * 17 + case)
            mean, sd_ = gs.normal(0, 5, nv), gs.uniform(0.5, 4.0, nv)
            mean, sd_ = mean.astype(np.float32), sd_.astype(np.float32)
            stats = {"mean": mean, "stdev": sd_, "minimum": (mean - 3 * sd_ - gs.uniform(0, 1, nv)).astype(np.float32),
                     "maximum": (np.abs(mean) + 3 * sd_ + 1.0).astype(np.float32)}
            iface = AnemoiModelInterface(config=cfg, graph_data=graph, statistics=stats, data_indices=idx, metadata={})
            for m in iface.modules():
                if hasattr(m, "dropout_p"):
                    m.dropout_p = 0.0
            in_idx = idx.data.input.full.long()
            z = torch.randn((b, multistep, n_grid, n_prog + n_forc), generator=torch.Generator().manual_seed(case))
            batch = z * torch.from_numpy(sd_).float()[in_idx] + torch.from_numpy(mean).float()[in_idx]
            with torch.no_grad():
                for name, p_ in iface.named_parameters():
                    if name.endswith("trainable"):
                        p_.normal_(0.0, 0.1)
            sd = {k: (v.detach().double() if v.is_floating_point() else v.detach()) for k, v in iface.state_dict().items()}
            want = ref.predict_step(sd, gt, batch.double(), multi_step=multistep, num_heads=16, num_layers=layers, num_chunks=1,
                                    prognostic_in=list(range(n_prog)), prognostic_out=list(range(n_prog)), processor=proc,
                                    mappers=maps)
            iface = iface.to(dev).eval()
            y1 = iface.predict_step(batch.to(dev))
            y2 = iface.predict_step(batch.to(dev))
            e12 = rel(y2.cpu(), y1.cpu())
            e_o = rel(y2.cpu(), want)
            if e_o > (5e-2 if bf16 else 5e-4) or e12 > (5e-2 if bf16 else 1e-4) or not torch.isfinite(y2).all():
                bad += 1
                print(f"  {what} methods={methods}: vs oracle {e_o:.2e}, second call vs first {e12:.2e}", flush=True)
        except Exception as exc:  # noqa: BLE001
            bad += 1
            print(f"  {what}: {type(exc).__name__}: {str(exc).splitlines()[0][:200]}", flush=True)
    os.environ.pop("ANEMOI_AMD_DTYPE", None)
    print(f"interface predict_step against the oracle: {bad} bad of {n_run}", flush=True)


def fuzz_mhsa_backward():
    """autograd.mhsa forward + backward (the MFMA dK/dV and dQ kernels at D = 64 / 32 in bf16, the VALU kernels elsewhere) at
    random sequence lengths around the tile borders, batch sizes, windows, with and without dropout (mask restated from the
    kernels' hash): output and d qkv against torch autograd in f64."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    from anemoi_models_amd import autograd
    from test_gpu_training import _dropout_keep_mask

    bad, n_run = 0, n_cases // 4
    for case in range(n_run):
        d = rng.choice([64, 64, 32, 32, 16, 8, 48, 5])
        dtype = torch.bfloat16 if (d in (64, 32) or rng.random() < 0.4) else torch.float32
        h, b = rng.choice([1, 2, 4]), rng.choice([1, 1, 2])
        s_len = rng.choice([rng.randint(1, 130), 32 * rng.randint(1, 40) + rng.choice([-1, 0, 1]), 512 * rng.randint(1, 3) + rng.choice([0, 2]),
                            rng.randint(130, 1500)])
        window = rng.choice([-1, -1, rng.randint(0, 200)])
        p = rng.choice([0.0, 0.0, 0.1, 0.35])
        seed_c = rng.randint(1, 2**31 - 1)
        c = h * d
        g = torch.Generator().manual_seed(seed * 100003 + case)
        qkv = (torch.randn(b * s_len, 3 * c, generator=g) * 0.7).to(dtype)
        dout = torch.randn(b * s_len, c, generator=g).to(dtype)
        ref_in = qkv.double().requires_grad_()
        q, k, v = (t.reshape(b, s_len, h, d).permute(0, 2, 1, 3) for t in ref_in.split(c, dim=1))
        sc = q @ k.transpose(-1, -2) / d**0.5
        if window >= 0:
            i = torch.arange(s_len)
            sc = sc.masked_fill((i[:, None] - i[None, :]).abs() > window, float("-inf"))
        prob = torch.softmax(sc, -1)
        if p > 0:
            prob = prob * _dropout_keep_mask(seed_c, p, b, h, s_len) * (1.0 / (1.0 - p))
        want = (prob @ v).permute(0, 2, 1, 3).reshape(b * s_len, c)
        want.backward(dout.double())
        what = f"mhsa backward case {case}: {dtype} b={b} s={s_len} h={h} d={d} window={window} p={p}"
        try:
            x = qkv.to(dev).requires_grad_()
            got = autograd.mhsa(x, b, h, window, p, seed_c)
            got.backward(dout.to(dev))
            e_o = rel(got.detach().cpu(), want.detach())
            e_g = float((x.grad.cpu().double() - ref_in.grad).abs().max() / ref_in.grad.abs().max().clamp_min(1e-9))
            if e_o > (2e-5 if dtype == torch.float32 else 2e-2) or e_g > (2e-4 if dtype == torch.float32 else 3e-2):
                bad += 1
                print(f"  {what}: out {e_o:.2e}, d qkv {e_g:.2e}", flush=True)
        except Exception as exc:  # noqa: BLE001
            bad += 1
            print(f"  {what}: {type(exc).__name__}: {str(exc).splitlines()[0][:160]}", flush=True)
    print(f"mhsa forward + backward (with dropout): {bad} bad of {n_run}", flush=True)


ONLY = os.environ.get("FUZZ_ONLY")  # e.g. FUZZ_ONLY=models: one fuzzer alone
for fn in (fuzz_linear, fuzz_models, fuzz_hierarchical, fuzz_interface, fuzz_gt_blocks_training, fuzz_gnn_and_transformer_blocks, fuzz_conv_dropout, fuzz_mhsa, fuzz_mhsa_backward, fuzz_edge_attention, fuzz_edge_scheduled, fuzz_edge_groups, fuzz_rows, fuzz_weight_grad):
    if ONLY is None or ONLY in fn.__name__:
        fn()
