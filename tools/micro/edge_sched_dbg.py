import torch, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from anemoi_models_amd import ops, runtime
DEV="cuda"
g = torch.Generator().manual_seed(1)
n_src, n_dst, c, h, up = 9000, 700, 1024, 16, int(os.environ.get("DBG_UP", "16"))
deg = torch.randint(6, 15, (n_dst,), generator=g)
dst = torch.repeat_interleave(torch.arange(n_dst), deg)
src = torch.randint(0, n_src, (int(deg.sum()),), generator=g)
plan = runtime.build_edge_plan(torch.stack([src, dst]).to(DEV), n_src, n_dst)
sched = plan.schedule(torch.bfloat16, c)
e = plan.num_edges
q = (torch.randn(n_dst, c, generator=g) * 0.5).bfloat16().to(DEV)
kv = (torch.randn(n_src, 2 * c, generator=g) * 0.5).bfloat16().to(DEV)
x_r = torch.randn(n_dst, c, generator=g).bfloat16().to(DEV)
u = (torch.randn(n_dst, h * up, generator=g) * 0.3).bfloat16().to(DEV)
attr = torch.randn(e, up, generator=g).to(DEV)
if os.environ.get("DBG_ZERO_U"): u.zero_()
if os.environ.get("DBG_ZERO_ATTR"): attr.zero_()
if os.environ.get("DBG_XR_ONLY"): pass
for xr in (x_r,):
    la, lb = torch.empty(n_dst, h, device=DEV), torch.empty(n_dst, h, device=DEV)
    a = ops.gt_edge_attention_folded(q, kv[:, :c], kv[:, c:], xr, u, attr, plan.rowptr, plan.col, h, up, lse=la)
    b = ops.gt_edge_attention_folded(q, kv[:, :c], kv[:, c:], xr, u, attr, plan.rowptr, plan.col, h, up, lse=lb, sched=sched)
    print("nan in plain", int(torch.isnan(a.float()).sum()), "nan in sched", int(torch.isnan(b.float()).sum()))
    d = (a != b) & ~(torch.isnan(a.float()) & torch.isnan(b.float()))
    print("xr" if xr is not None else "no xr", "differing: main cols", int(d[:, :c].sum()), "of", d[:, :c].numel(), "t cols", int(d[:, c:c+h*up].sum()), "lse", int((la != lb).sum()),
          "rows with diffs", int(d.any(1).sum()), "deg of first diff rows", deg[d.any(1).cpu()][:10].tolist(),
          "deg histogram of diff rows", torch.bincount(deg[d.any(1).cpu()]).tolist())
    if d.any():
        i = int(torch.nonzero(d.any(1))[0]); cols = torch.nonzero(d[i]).flatten()[:8].tolist()
        print(" row", i, "deg", int(deg[i]), "cols", cols, a[i, cols].float().tolist(), b[i, cols].float().tolist())
        dc = d.any(0).nonzero().flatten()
        print(" differing columns: min", int(dc.min()), "max", int(dc.max()), "count", dc.numel(), "t-col diffs per (col % up):",
              torch.bincount((dc[dc >= c] - c) % up, minlength=up).tolist() if (dc >= c).any() else None)
