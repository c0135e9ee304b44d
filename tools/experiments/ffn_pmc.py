"""A few launches of mst_ffn_ln_fwd / _bwd at configs[1]'s encoder shape (16384 x 256 -> 1024 -> 256) for rocprofv3 --pmc."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from musicstyletransfer_amd import ops as o
dev = torch.device("cuda", 0)
BF = torch.bfloat16
M, D, F = 16384, 256, 1024
g = torch.Generator().manual_seed(1)
r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
x = r(M, D).to(BF)
W1, W2 = r(F, D, sc=0.06).to(BF), r(D, F, sc=0.03).to(BF)
b1, b2, gam, bet = r(F, sc=0.1), r(D, sc=0.1), 1 + 0.1 * r(D), r(D, sc=0.1)
seedp = torch.tensor([55, 0, 0, 0], dtype=torch.int64, device=dev)
ff1 = dict(K=D, bias=b1, act=o.ACT_RELU, dropout_p=0.2, dropout_seed_ptr=seedp, dropout_site=4)
ff2 = dict(K=F, bias=b2, dropout_p=0.2, dropout_seed_ptr=seedp, dropout_site=5, resid=x)
a, h, y = torch.zeros(M, F, dtype=BF, device=dev), torch.zeros(M, D, dtype=BF, device=dev), torch.zeros(M, D, dtype=BF, device=dev)
mean, rstd = torch.zeros(M, device=dev), torch.zeros(M, device=dev)
dff, W2t, W1t = r(M, D, sc=0.5).to(BF), r(F, D, sc=0.05).to(BF), r(D, F, sc=0.05).to(BF)
dpre, dx, dxm = torch.zeros(M, F, dtype=BF, device=dev), torch.zeros(M, D, dtype=BF, device=dev), torch.zeros(M, D, dtype=BF, device=dev)
dg, db = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
part = torch.zeros(o.gemm_nt_ln_parts(M), 2 * D, device=dev)
mk = dict(mask_mode=1, dropout_p=0.2, dropout_seed_ptr=seedp, dropout_site=2)
for _ in range(5):
    o.ffn_ln_fwd(x, W1, a, W2, h, gam, bet, y, mean, rstd, ff1=ff1, ff2=ff2)
    o.ffn_ln_bwd(dff, W2t, dpre, a, W1t, dx, h, gam, mean, rstd, dg, db, alpha=1.25, dx_masked=dxm, resid=x, partials=part, **mk)
torch.cuda.synchronize()
