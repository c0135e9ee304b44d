"""Graph-replay timing of the latent block's forward launch with / without the decoder's first K | Q | V projection riding on it
(mst_latent_fwd_qkv) at configs[1]'s shapes; weights made cold between launches by an Adam-sized sweep is NOT attempted: numbers are
hot-cache lower bounds, the step-level A/B (MST_LATENT_QKV=0/1) is the judge."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from musicstyletransfer_amd import ops as o
from tools.bench_ffn import timeit, dev, BF

B, T, De, Z, Dd, Cn = 64, 256, 256, 64, 128, 10
Sd = T + 1
g = torch.Generator().manual_seed(1)
r = lambda *s, sc=1.0, dt=torch.float32: (torch.randn(*s, generator=g) * sc).to(dev).to(dt)
enc = r(B, T, De, dt=BF)
Wl, bl, Wh, bh = r(2 * Z, De, sc=0.2), r(2 * Z, sc=0.5) + 1, r(Dd, Z, sc=0.3), r(Dd, sc=0.1)
cls_d, pos_d, eps = r(Cn, Dd), r(Sd, Dd), r(B, Z)
classes = (torch.arange(B, dtype=torch.int32) % Cn).to(dev)
Wq, bq = r(3 * Dd, Dd, sc=0.1, dt=BF), r(3 * Dd, sc=0.1)
mu, sigma, z, kl = torch.zeros(B, Z, device=dev), torch.zeros(B, Z, device=dev), torch.zeros(B, Z, device=dev), torch.zeros(B, device=dev)
x0_d = r(B * Sd, Dd, dt=BF)
qkv = torch.zeros(B * Sd, 3 * Dd, dtype=BF, device=dev)
lat = (enc, Wl, bl, eps, Wh, bh, classes, cls_d, pos_d, math.sqrt(Dd), mu, sigma, z, kl, x0_d.view(B, Sd, Dd))
print(f"latent_fwd alone            : {timeit(lambda: o.latent_fwd(*lat)):.1f} us")
print(f"projection GEMM alone (all rows, 64 x 64 tiles): {timeit(lambda: o.gemm_nt(x0_d, Wq, qkv, K=Dd, bias=bq)):.1f} us")
print(f"projection GEMM rows 1..T (remapped)           : {timeit(lambda: o.gemm_nt(x0_d, Wq, qkv, M=B * T, K=Dd, bias=bq, a_remap=(T, Sd, 1), c_remap=(T, Sd, 1))):.1f} us")
print(f"both, two launches          : {timeit(lambda: (o.latent_fwd(*lat), o.gemm_nt(x0_d, Wq, qkv, K=Dd, bias=bq))):.1f} us")
print(f"mst_latent_fwd_qkv          : {timeit(lambda: o.latent_fwd_qkv(*lat, x0_d, Wq, bq, qkv)):.1f} us")
