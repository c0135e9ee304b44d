"""Per-stage timeline of ONE workgroup of mst_ffn_ln_fwd / _bwd at the step's shape (M 16384, 256 -> 1024 -> 256).
Build with MST_EXTRA_FLAGS="gemm_nt.hip=-DMST_FFN_STAMPS" (GPU box)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from musicstyletransfer_amd import ops as o, _lib
dev = torch.device("cuda", 0); BF = torch.bfloat16
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
lib = _lib.load()
out = (C.c_uint64 * (8 + 48 * 4))()
flush = torch.zeros(64 << 20, dtype=torch.uint8, device=dev)
for it in range(6):
    flush.add_(1); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); o.ffn_ln_fwd(x, W1, a, W2, h, gam, bet, y, mean, rstd, ff1=ff1, ff2=ff2); e1.record(); torch.cuda.synchronize()
    assert lib.mst_debug_ffn_stamps(out) == 0
    t = np.array(list(out), dtype=np.int64)
    if it < 2:
        continue
    rt_us = ((t[191] - t[190]) & 0xffffffff) / 100.0
    clk = (t[3] - t[0]) / rt_us  # s_memtime ticks per us
    st = t[8:8 + 32 * 4].reshape(32, 4)
    prev_bar = np.concatenate([[t[1]], st[:-1, 3]])
    d = np.stack([st[:, 0] - prev_bar, st[:, 1] - st[:, 0], st[:, 2] - st[:, 1], st[:, 3] - st[:, 2]], 1) / clk
    print(f"launch {e0.elapsed_time(e1) * 1e3:.1f} us; workgroup {rt_us:.1f} us: prologue {(t[1] - t[0]) / clk:.2f}, stages {(t[2] - t[1]) / clk:.2f}, epilogue {(t[3] - t[2]) / clk:.2f}  ({clk:.0f} ticks/us)")
    print(f"  LayerNorm epilogue: acc->LDS+barrier {(t[5] - t[2]) / clk:.2f}, parameter / row loads issued {(t[6] - t[5]) / clk:.2f}, row pass {(t[7] - t[6]) / clk:.2f}, to the end {(t[3] - t[7]) / clk:.2f}")
    names = ["issue", "mma(+chunk epilogue at s=3)", "store", "barrier(+copy-out at s=3)"]
    for c in range(4):
        print("  chunk %d stages 0-7 total: " % c + " ".join("%.2f" % v for v in d[c * 8:(c + 1) * 8].sum(1)))
    med = np.median(d.reshape(4, 8, 4), axis=0)  # [stage, phase] median over chunks
    for s_ in range(8):
        print("  stage %d: issue %.2f mma %.2f store %.2f barrier %.2f" % (s_, *med[s_]))
