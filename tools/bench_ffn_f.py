"""mst_ffn_ln_fwd time against the hidden width F (number of chunks): per-chunk cost vs fixed prologue + LayerNorm epilogue"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from musicstyletransfer_amd import ops as o
from tools.bench_ffn import timeit, dev, BF

D, M = 256, 16384
g = torch.Generator().manual_seed(1)
r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
for F in (256, 512, 1024, 2048, 4096):
    W1, W2 = r(F, D, sc=0.06).to(BF), r(D, F, sc=0.03).to(BF)
    b1, b2, gam, bet = r(F, sc=0.1), r(D, sc=0.1), 1 + 0.1 * r(D), r(D, sc=0.1)
    x = r(M, D).to(BF)
    a, h, y = torch.zeros(M, F, dtype=BF, device=dev), torch.zeros(M, D, dtype=BF, device=dev), torch.zeros(M, D, dtype=BF, device=dev)
    mean, rstd = torch.zeros(M, device=dev), torch.zeros(M, device=dev)
    for p in (0.0, 0.2):
        seedp = torch.tensor([55, 0, 0, 0], dtype=torch.int64, device=dev)
        dk = dict(dropout_p=p, dropout_seed_ptr=seedp) if p else {}
        ff1 = dict(K=D, bias=b1, act=o.ACT_RELU, dropout_site=4, **dk)
        ff2 = dict(K=F, bias=b2, resid=x, dropout_site=5, **dk)
        print(f"F {F:5d} ({F // D} chunks) dropout {p}: {timeit(lambda: o.ffn_ln_fwd(x, W1, a, W2, h, gam, bet, y, mean, rstd, ff1=ff1, ff2=ff2)):.1f} us")
