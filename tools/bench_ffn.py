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


if __name__ == "__main__":
  for M, D, F, self_resid in [(16384, 256, 1024, False), (16384, 128, 512, True), (16448, 128, 512, True)][(1 if os.environ.get("MST_BENCH_FFN_128") else 0):]:
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

      print(f"M {M} D {D} F {F}: forward, three launches {timeit(three):.1f} us   fused {timeit(fused):.1f} us")
      # backward: FFN2 dgrad (gate) + FFN1 dgrad (+resid) + LayerNorm backward
      dff, W2t, W1t = r(M, D, sc=0.5).to(BF), r(F, D, sc=0.05).to(BF), r(D, F, sc=0.05).to(BF)
      dpre, dy, dx, dxm = torch.zeros(M, F, dtype=BF, device=dev), torch.zeros(M, D, dtype=BF, device=dev), torch.zeros(M, D, dtype=BF, device=dev), torch.zeros(M, D, dtype=BF, device=dev)
      dg, db = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
      part = torch.zeros(o.gemm_nt_ln_parts(M), 2 * D, device=dev)
      part2 = torch.zeros(max(o.layernorm_bwd_parts(M, D), 1), 2 * D, device=dev)
      mk = dict(mask_mode=1, dropout_p=0.2, dropout_seed_ptr=seedp, dropout_site=2)

      def bwd_sep():
          o.gemm_nt(dff, W2t, dpre, gate=a, alpha=1.25)
          if D == 128:
              o.gemm_nt_ln_bwd(dpre, W1t, dx, h, gam, mean, rstd, dg, db, dx_masked=dxm, resid=x, partials=part, **mk)
          else:
              o.gemm_nt(dpre, W1t, dy, resid=x)
              o.layernorm_bwd(h, gam, mean, rstd, dy, dx, dg, db, D=D, dx_masked=dxm, partials=part2, **mk)

      def bwd_fused():
          o.ffn_ln_bwd(dff, W2t, dpre, a, W1t, dx, h, gam, mean, rstd, dg, db, alpha=1.25, dx_masked=dxm, resid=x, partials=part, **mk)

      print(f"M {M} D {D} F {F}: backward, separate launches {timeit(bwd_sep):.1f} us   fused {timeit(bwd_fused):.1f} us")
