"""Correctness sweep of anemoi_linear (bf16 fast path) on shapes that exercise single-tile workgroups, ragged N, tiny K,
residual / activation epilogues (GPU only; not part of the product)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from anemoi_models_amd import ops
torch.manual_seed(0)
tot=0
for (m,n,k,act,res) in ((2050,384,256,"ReLU",True),(2048,512,256,"Identity",False),(4096,512,1024,"Identity",False),(4096,512,1024,"Identity",True),(4096,384,1024,"Identity",False),(65536,1024,128,"GELU",True),(40962,4288,1024,"Identity",False),(40962,1024,1216,"Identity",True),(3072,2240,1024,"SiLU",True),(131072,256,128,"Identity",False),(70000,520,192,"GELU",True),
    # remainder round as half tiles: 160 x 4 = 640 tiles, 20 x 4 = 80 tiles, 44 x 8 tiles with LN-free GELU, ragged rows
    (40962,1024,1216,"Identity",True),(5120,1024,4096,"Identity",True),(11264,2048,512,"GELU",False),(10243,1024,1024,"SiLU",True)):
    x=torch.randn(m,k).bfloat16().cuda(); w=(torch.randn(n,k)/k**0.5).bfloat16().cuda(); b=torch.randn(n).cuda(); r=torch.randn(m,n).bfloat16().cuda() if res else None
    for rep in range(3):
        got=ops.linear(x,w,b,act=act,residual=r).float()
    want={"Identity":lambda t:t,"ReLU":F.relu,"GELU":F.gelu,"SiLU":F.silu}[act](F.linear(x.float(),w.float(),b))
    if res: want=want+r.float()
    bad=~torch.isfinite(got) | ((got-want).abs()>0.08)
    tot+=int(bad.sum())
    print(m,n,k,act,res,"bad",int(bad.sum()),"max err",float((got-want).abs().max()))
print("TOTAL BAD",tot)
