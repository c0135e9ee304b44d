"""Where does mst_row_tail_bwd spend its time? Build with MST_EXTRA_FLAGS="row_tail.hip=-DMST_TAIL_STAMPS" (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from musicstyletransfer_amd import ops as o
dev = torch.device("cuda", 0); BF = torch.bfloat16
B, S, D = 64, 256, 256; F = 4 * D
g = torch.Generator().manual_seed(7)
r = lambda *sh, sc=1.0, dt=BF: (torch.randn(*sh, generator=g) * sc).to(dt).to(dev)
dy, h2, h1 = r(B * S, D, sc=0.5), r(B * S, D), r(B * S, D)
a = torch.relu(r(B * S, F))
W2t, W1t, Wpt = r(F, D, sc=0.05), r(D, F, sc=0.05), r(D, D, sc=0.06)
g1, g2 = 1 + r(D, sc=0.1, dt=torch.float32), 1 + r(D, sc=0.1, dt=torch.float32)
row0 = lambda t: t.view(B, S, -1)[:, 0, :]
m1, m2, r1, r2 = (torch.zeros(B * S, device=dev) for _ in range(4))
r1 += 1; r2 += 1
seedp = torch.tensor([99, 0, 0, 0], dtype=torch.int64, device=dev)
z = lambda n, w: torch.zeros(n, w, dtype=BF, device=dev)
dh, dhm, dx1, dh1m, dpre, dh1, datt = z(B, D), z(B, D), z(B, D), z(B, D), z(B, F), z(B * S, D), z(B * S, D)
dg = [torch.zeros(D, device=dev) for _ in range(4)]
sync = torch.zeros(32, dtype=torch.int32, device=dev)


def tail_stamps():
    """the stamps the -DMST_TAIL_STAMPS build keeps in a device array of its own (mst_debug_tail_stamps)"""
    import ctypes as C
    from musicstyletransfer_amd import _lib
    buf = (C.c_uint32 * 32)()
    assert _lib.load().mst_debug_tail_stamps(buf) == 0
    return np.array(list(buf), dtype=np.int64)

flush = torch.zeros(64 << 20, dtype=torch.uint8, device=dev)
names = ["LN2 bwd (+W2t to LDS)", "FFN2 dgrad slice", "barrier 1", "FFN1 dgrad slice", "barrier 2", "LN1 bwd", "W_proj dgrad"]  # q.sync = sync[0:]: stamps at sync[8 + i]
for it in range(10):
    sync.zero_(); flush.add_(1); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    o.row_tail_bwd(row0(dy), row0(h2), row0(h1), row0(a), m1, r1, m2, r2, g1, g2, W2t, W1t, Wpt, dh, dhm, dx1, dh1m, dpre, row0(dh1), row0(datt),
                   dg[0], dg[1], dg[2], dg[3], sync[0:3], stat_stride=S, phys_stride=S, dropout_p=0.2, dropout_seed_ptr=seedp, site0=6)
    e1.record(); torch.cuda.synchronize()
    t = tail_stamps()[:8]
    d = (np.diff(t) & 0xffffffff) / 100.0  # stamps 0..7: stage 1 | stage 2 | barrier 1 | stage 3 | barrier 2 | stage 4 | stage 5
    if it >= 3:
        print("event %.1f us | " % (e0.elapsed_time(e1) * 1e3) + " ".join("%s %.1f" % (n, v) for n, v in zip(names, d)) + " | sum %.1f" % d.sum())
