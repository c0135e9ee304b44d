"""launch time of mst_row_tail_fwd (B 64, width 256) alone and with riding GEMMs of several sizes (graph-replayed, hot caches)"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from musicstyletransfer_amd import ops as o
gpu = torch.device("cuda", 0)
dtype = torch.bfloat16
B, S, D, p = 64, 256, 256, 0.2
F = 4 * D
g = torch.Generator().manual_seed(7)
r = lambda *sh, sc=1.0, dt=dtype: (torch.randn(*sh, generator=g) * sc).to(dt).to(gpu)
att, xin = r(B * S, D), r(B * S, D)
Wp, W1, W2 = r(D, D, sc=0.06), r(F, D, sc=0.06), r(D, F, sc=0.03)
bp, b1, b2 = r(D, sc=0.1, dt=torch.float32), r(F, sc=0.1, dt=torch.float32), r(D, sc=0.1, dt=torch.float32)
g1, be1, g2, be2 = (1 + r(D, sc=0.1, dt=torch.float32)), r(D, sc=0.1, dt=torch.float32), (1 + r(D, sc=0.1, dt=torch.float32)), r(D, sc=0.1, dt=torch.float32)
seedp = torch.tensor([99, 0, 0, 0], dtype=torch.int64, device=gpu)
row0 = lambda t: t.view(B, S, -1)[:, 0, :]
z = lambda w: torch.zeros(B * S, w, dtype=dtype, device=gpu)
f = dict(h1=z(D), x1=z(D), a=z(F), h2=z(D), x2=z(D), m1=torch.zeros(B * S, device=gpu), r1=torch.zeros(B * S, device=gpu), m2=torch.zeros(B * S, device=gpu), r2=torch.zeros(B * S, device=gpu))
sync = torch.zeros(8, dtype=torch.int32, device=gpu)
queue_all = torch.zeros(64, dtype=torch.int32, device=gpu)
queue = queue_all[32:]

def run(ride):
    o.zero(sync)
    o.zero(queue_all)
    o.row_tail_fwd(row0(att), row0(xin), Wp, bp, g1, be1, W1, b1, W2, b2, g2, be2, row0(f["h1"]), row0(f["x1"]), row0(f["a"]), row0(f["h2"]),
                   row0(f["x2"]), f["m1"], f["r1"], f["m2"], f["r2"], sync[0:3], stat_stride=S, phys_stride=S, dropout_p=p,
                   dropout_seed_ptr=seedp, site0=6, rider=ride, queue=queue[0:1])

st = torch.cuda.Stream()
with torch.cuda.stream(st):
    base = bench.time_launch(o, lambda: run(None), 10) * 1e3
    zt = bench.time_launch(o, lambda: (o.zero(sync), o.zero(queue_all)), 10) * 1e3
    print(f"tail alone {base:.1f} us (incl. {zt:.1f} us of zeroing launches)")
    for (Bg, T, N, K) in ((1, 128, 128, 128), (8, 256, 384, 128), (32, 256, 384, 128), (64, 256, 384, 128), (64, 256, 128, 384), (64, 128, 384, 128)):
        Md = Bg * (T + 1)
        A, W, C = r(Md, K), r(N, K, sc=0.08), torch.zeros(Md, N, dtype=dtype, device=gpu)
        ride = dict(A=A, B=W, C_out=C, M=Bg * T, N=N, K=K, bias=r(N, sc=0.1, dt=torch.float32), a_remap=(T, T + 1, 1), c_remap=(T, T + 1, 1))
        t = bench.time_launch(o, lambda: run(ride), 10) * 1e3
        alone = bench.time_launch(o, lambda: o.gemm_nt(A, W, C, **{k: v for k, v in ride.items() if k not in ("A", "B", "C_out")}), 10) * 1e3
        print(f"rider M={Bg * T} N={N} K={K} (T={T}): tail+rider {t:.1f} us; the GEMM as its own launch {alone:.1f} us")
