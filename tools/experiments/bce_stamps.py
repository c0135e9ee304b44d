"""Phases of one workgroup of mst_gemm_sigmoid_bce_dgrad_ln at configs[1]'s shape. Needs a -DMST_FFN_STAMPS build with TEMPORARY stamps in
gemm_nt.hip (not in the tree): gemm_bce_dgrad_ln_kernel — FFN_STAMP(0) + FFN_RT(190) at its start, (1) behind gemm_bce_tile, (2) behind the second
K loop, (3) + FFN_RT(191) at its end; gemm_bce_tile<KEEP> — (8) behind its K loop, (9) / (10) around the sweep.
Measured (r04): workgroup 14.7 us = K loop 3.5 (two cold stages), staging 0.4, BCE sweep 2.0, loss sum + barrier 2.3, second GEMM 1.6, LayerNorm
epilogue 4.9 — nothing dominant; the launch is 23 us in the step."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from musicstyletransfer_amd import ops as o, _lib
gpu = torch.device("cuda", 0); dtype = torch.bfloat16
B, T, P, D = 64, 256, 128, 128
Sd, M = T + 1, B * T
g = torch.Generator().manual_seed(3)
rnd = lambda sh, sc=1.0, dt=dtype: (torch.randn(*sh, generator=g) * sc).to(dt).to(gpu)
x, W = rnd((B * Sd, D)), rnd((P, D), 0.2)
Wt = W.t().contiguous()
bias = rnd((P,), 0.1, torch.float32)
labels = (torch.rand(M, P, generator=g) < 0.05).to(torch.uint8).to(gpu)
h2 = rnd((B * Sd, D)); gamma = 1 + 0.1 * rnd((D,), 1.0, torch.float32)
mean, rstd = h2.float().mean(1), (h2.float().var(1, unbiased=False) + 1e-5).rsqrt()
seedp = torch.tensor([77, 0, 0, 0], dtype=torch.int64, device=gpu)
loss = torch.zeros(B, device=gpu); dl = torch.zeros(M, P, dtype=dtype, device=gpu); dh = torch.zeros(B * Sd, D, dtype=dtype, device=gpu)
parts = torch.zeros(o.gemm_nt_ln_parts(M), 2 * D, device=gpu); dg, db = torch.zeros(D, device=gpu), torch.zeros(D, device=gpu)
dgrad = dict(A=dl, B=Wt, dX_out=dh, x=h2, gamma=gamma, mean=mean, rstd=rstd, dgamma=dg, dbeta=db, mask_mode=2, partials=parts, M=M, N=D, K=P,
             c_remap=(T, Sd, 1), dropout_p=0.2, dropout_seed_ptr=seedp, dropout_site=9)
kw = dict(dlogits=dl, probs=None, label_smoothing=0.0, downweight=False, gscale=1.0, M=M, K=D, bias=bias, a_remap=(T, Sd, 1))
lib = _lib.load(); out = (C.c_uint64 * (8 + 48 * 4))()
flush = torch.zeros(64 << 20, dtype=torch.uint8, device=gpu)
for it in range(6):
    flush.add_(1); loss.zero_(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); o.gemm_sigmoid_bce(x, W, labels, loss, T, dgrad=dgrad, **kw); e1.record(); torch.cuda.synchronize()
    assert lib.mst_debug_ffn_stamps(out) == 0
    t = np.array(list(out), dtype=np.int64)
    rt = ((t[191] - t[190]) & 0xffffffff) / 100.0
    clk = (t[3] - t[0]) / rt
    us = lambda a, b: (t[b] - t[a]) / clk
    print(f"launch {e0.elapsed_time(e1) * 1e3:.1f} us; workgroup {rt:.1f}: GEMM1 K loop {us(0, 8):.2f}, stage+labels {us(8, 9):.2f}, BCE sweep {us(9, 10):.2f}, "
          f"loss sum {us(10, 1):.2f}, GEMM2 {us(1, 2):.2f}, LN epilogue {us(2, 3):.2f}")
