"""Per-stage timeline of ONE workgroup of mst_ffn_ln_bwd_lead at the step's shape (M 16384, width 256, hidden 1024).
Build with MST_EXTRA_FLAGS="gemm_nt.hip=-DMST_FFN_STAMPS" (GPU box)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from musicstyletransfer_amd import ops as o, _lib
dev = torch.device("cuda", 0); BF = torch.bfloat16
M, D, F = 16384, 256, 1024
g = torch.Generator().manual_seed(1)
r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
gate = torch.relu(r(M, F)).to(BF)
W2t, W1t = r(F, D, sc=0.05).to(BF), r(D, F, sc=0.05).to(BF)
x, dyin, xin = r(M, D).to(BF), r(M, D).to(BF), r(M, D, sc=1.5).to(BF)
gam, gin = 1 + 0.1 * r(D), 1 + 0.1 * r(D)
mean, rstd = x.float().mean(1), 1.0 / torch.sqrt(x.float().var(1, unbiased=False) + 1e-5)
mean_in, rstd_in = xin.float().mean(1), 1.0 / torch.sqrt(xin.float().var(1, unbiased=False) + 1e-5)
seedp = torch.tensor([91, 0, 0, 0], dtype=torch.int64, device=dev)
z = lambda w: torch.zeros(M, w, dtype=BF, device=dev)
dh, dhm, dpre, dx, dxm = z(D), z(D), z(F), z(D), z(D)
dg, db = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
parts = o.gemm_nt_ln_parts(M)
part, pin = torch.zeros(parts, 2 * D, device=dev), torch.zeros(parts, 2 * D, device=dev)
lib = _lib.load()
out = (C.c_uint64 * (8 + 48 * 4))()
flush = torch.zeros(64 << 20, dtype=torch.uint8, device=dev)
for it in range(6):
    flush.add_(1); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    o.ffn_ln_bwd(dhm, W2t, dpre, gate, W1t, dx, x, gam, mean, rstd, dg, db, alpha=1.25, dx_masked=dxm, mask_mode=1, partials=part, resid=dh,
                 dropout_p=0.2, dropout_seed_ptr=seedp, dropout_site=2,
                 lead=dict(dy=dyin, x=xin, gamma=gin, mean=mean_in, rstd=rstd_in, dx=dh, dx_masked=dhm, partials=pin, dropout_p=0.2,
                           dropout_seed_ptr=seedp, dropout_site=7))
    e1.record(); torch.cuda.synchronize()
    assert lib.mst_debug_ffn_stamps(out) == 0
    t = np.array(list(out), dtype=np.int64)
    if it < 2:
        continue
    rt_us = ((t[191] - t[190]) & 0xffffffff) / 100.0
    clk = (t[3] - t[0]) / rt_us
    st = t[8:8 + 32 * 4].reshape(32, 4)
    prev_bar = np.concatenate([[t[1]], st[:-1, 3]])
    d = np.stack([st[:, 0] - prev_bar, st[:, 1] - st[:, 0], st[:, 2] - st[:, 1], st[:, 3] - st[:, 2]], 1) / clk
    print(f"launch {e0.elapsed_time(e1) * 1e3:.1f} us; workgroup {rt_us:.1f} us: prologue (LayerNorm-2 backward) {(t[1] - t[0]) / clk:.2f}, stages {(t[2] - t[1]) / clk:.2f}, "
          f"epilogue {(t[3] - t[2]) / clk:.2f}")
    print(f"  LayerNorm-1 backward epilogue: acc->LDS+barrier {(t[5] - t[2]) / clk:.2f}, loads issued {(t[6] - t[5]) / clk:.2f}, row pass {(t[7] - t[6]) / clk:.2f}, "
          f"parameter-gradient partials {(t[3] - t[7]) / clk:.2f}")
    for c in range(4):
        print("  chunk %d stages 0-7 total: " % c + " ".join("%.2f" % v for v in d[c * 8:(c + 1) * 8].sum(1)))
    med = np.median(d.reshape(4, 8, 4), axis=0)
    print("  stage 3 (+ gate pass): issue %.2f mma %.2f store %.2f barrier+copy %.2f" % tuple(med[3]))
