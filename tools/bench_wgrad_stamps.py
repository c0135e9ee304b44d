"""Per-stage timeline of ONE workgroup of the whole-step weight-gradient launch shape.
Build with MST_EXTRA_FLAGS="gemm_wgrad.hip=-DMST_WGRAD_STAMPS [-DMST_WGRAD_GLDS=0]" (GPU box)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from musicstyletransfer_amd import ops as o, _lib
dev = torch.device("cuda", 0); BF = torch.bfloat16
M, D, F = 16384, 256, 1024
g = torch.Generator().manual_seed(3)
r = lambda *sh: (torch.randn(*sh, generator=g) * 0.1).to(BF).to(dev)
probs, keep = [], []
for i in range(4):  # 4 x (FFN1-like 1024 x 256 + FFN2-like 256 x 1024): 32 tiles of 256 x 256 -> M split 8, 32 stages per workgroup
    dA, X, dW = r(M, F), r(M, D), torch.zeros(F, D, device=dev)
    dB, Y, dV = r(M, D), r(M, F), torch.zeros(D, F, device=dev)
    keep += [dA, X, dW, dB, Y, dV]
    probs += [o.wgrad_problem(dA, X, dW), o.wgrad_problem(dB, Y, dV)]
scratch = torch.zeros(16 * 1024 * 1024, device=dev)
lib = _lib.load()
out = (C.c_uint64 * (4 + 64 * 4))()
for it in range(5):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); o.gemm_wgrad_batch(probs, scratch=scratch); e1.record(); torch.cuda.synchronize()
    assert lib.mst_debug_wgrad_stamps(out) == 0
    t = np.array(list(out), dtype=np.int64)
    if it < 2:
        continue
    st = t[4:].reshape(64, 4)
    n = int((st[:, 3] > 0).sum())
    clk = 18.0  # s_memtime ticks per us as calibrated against the launch time (the counter runs at ~1.8 GHz here); x100 below -> 0.01 us
    d_issue = (st[:n, 0] - np.concatenate([[t[1]], st[:n - 1, 3]])) / clk
    d_comp = (st[:n, 1] - st[:n, 0]) / clk
    d_store = (st[:n, 2] - st[:n, 1]) / clk
    d_bar = (st[:n, 3] - st[:n, 2]) / clk
    print(f"launch {e0.elapsed_time(e1) * 1e3:.1f} us; workgroup: prologue {(t[1] - t[0]) / clk:.2f}, {n} stages {(t[2] - t[1]) / clk:.1f}, "
          f"epilogue {(t[3] - t[2]) / clk:.1f} us")
    print("  per stage (median us): issue %.2f  compute %.2f  store %.2f  barrier %.2f  total %.2f" %
          (np.median(d_issue), np.median(d_comp), np.median(d_store), np.median(d_bar), np.median(d_issue + d_comp + d_store + d_bar)))
