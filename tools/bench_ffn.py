"""Graph-replay timing of the feed-forward block: mst_ffn_ln_fwd vs gemm_nt + gemm_nt + layernorm_fwd. usage: bench_ffn.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from musicstyletransfer_amd import ops as o

dev = torch.device("cuda", 0)
BF = torch.bfloat16


def timeit(fn, reps=20, inner=20):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            for _ in range(inner):
                fn()
        gr.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(reps):
            gr.replay()
        e1.record(s); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * inner)


for M, D, F, self_resid in [(16384, 256, 1024, False), (16448, 128, 512, True)]:
    g = torch.Generator().manual_seed(1)
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
    x = r(M, D).to(BF)
    W1, W2 = r(F, D, sc=0.06).to(BF), r(D, F, sc=0.03).to(BF)
    b1, b2, gam, bet = r(F, sc=0.1), r(D, sc=0.1), 1 + 0.1 * r(D), r(D, sc=0.1)
    seedp = torch.tensor([55, 0, 0, 0], dtype=torch.int64, device=dev)
    ff1 = dict(K=D, bias=b1, act=o.ACT_RELU, dropout_p=0.2, dropout_seed_ptr=seedp, dropout_site=4)
    ff2 = dict(K=F, bias=b2, dropout_p=0.2, dropout_seed_ptr=seedp, dropout_site=5)
    ff2.update(dict(self_resid=True) if self_resid else dict(resid=x))
    a, h, y = torch.zeros(M, F, dtype=BF, device=dev), torch.zeros(M, D, dtype=BF, device=dev), torch.zeros(M, D, dtype=BF, device=dev)
    mean, rstd = torch.zeros(M, device=dev), torch.zeros(M, device=dev)

    def three():
        o.gemm_nt(x, W1, a, **ff1)
        o.gemm_nt(a, W2, h, **ff2)
        o.layernorm_fwd(h, gam, bet, y, mean, rstd, D=D)

    def fused():
        o.ffn_ln_fwd(x, W1, a, W2, h, gam, bet, y, mean, rstd, ff1=ff1, ff2=ff2)

    print(f"M {M} D {D} F {F}: three launches {timeit(three):.1f} us   fused {timeit(fused):.1f} us")
