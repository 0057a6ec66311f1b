import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from anemoi_models_amd import ops
S, H, D = int(sys.argv[1]) if len(sys.argv) > 1 else 1024, 4, 64
C = H * D
torch.manual_seed(0)
qkv = torch.randn(S, 3 * C, device="cuda").bfloat16()
if len(sys.argv) > 2:  # keys beyond the first N contribute nothing distinguishable: make V one-hot per key block
    mode = sys.argv[2]
    if mode == "ones":
        qkv[:, 2 * C:] = 1.0
    if mode == "keyid":
        qkv[:, 2 * C:] = 0
        for k in range(S):
            qkv[k, 2 * C + (k % 64)] = 1.0  # v[k] = e_(k mod 64): out[q, d] = sum of P over keys = d mod 64
out, lse = ops.mhsa(qkv, 1, H, return_lse=True) if "return_lse" in ops.mhsa.__code__.co_varnames else (ops.mhsa(qkv, 1, H), None)
out = out.float()
q, k, v = (qkv[:, i * C:(i + 1) * C].float().reshape(S, H, D).permute(1, 0, 2) for i in range(3))
sc = q @ k.transpose(1, 2) / D**0.5
want = (torch.softmax(sc, -1) @ v).permute(1, 0, 2).reshape(S, C)
d = (out - want).abs()
print("max err", float(d.max()), "want max", float(want.abs().max()))
rows = d.max(1).values
print("per 32-query block max err:", [round(float(rows[i:i + 32].max()), 4) for i in range(0, min(S, 512), 32)])
r = 96
print("row 96 head 0: got ", [round(float(x), 4) for x in out[r, :64]])
print("row 96 head 0: want", [round(float(x), 4) for x in want[r, :64]])
if lse is not None:
    wl = torch.logsumexp(sc, -1)
    print("lse err per block:", [round(float((lse[0, i:i + 32] - wl[0, i:i + 32]).abs().max()), 5) for i in range(0, min(S, 512), 32)])
