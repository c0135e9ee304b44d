"""Where does mst_row_tail_fwd spend its time? Build with MST_EXTRA_FLAGS="row_tail.hip=-DMST_TAIL_STAMPS" (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from musicstyletransfer_amd import ops as o
dev = torch.device("cuda", 0); BF = torch.bfloat16
B, S, D = 64, 256, 256; F = 4 * D
g = torch.Generator().manual_seed(7)
r = lambda *sh, sc=1.0, dt=BF: (torch.randn(*sh, generator=g) * sc).to(dt).to(dev)
att, xin = r(B * S, D), r(B * S, D)
Wp, W1, W2 = r(D, D, sc=0.06), r(F, D, sc=0.06), r(D, F, sc=0.03)
bp, b1, b2 = r(D, sc=0.1, dt=torch.float32), r(F, sc=0.1, dt=torch.float32), r(D, sc=0.1, dt=torch.float32)
g1, be1, g2, be2 = (1 + r(D, sc=0.1, dt=torch.float32)), r(D, sc=0.1, dt=torch.float32), (1 + r(D, sc=0.1, dt=torch.float32)), r(D, sc=0.1, dt=torch.float32)
seedp = torch.tensor([99, 0, 0, 0], dtype=torch.int64, device=dev)
row0 = lambda t: t.view(B, S, -1)[:, 0, :]
z = lambda w: torch.zeros(B * S, w, dtype=BF, device=dev)
h1, x1, a, h2, x2 = z(D), z(D), z(F), z(D), z(D)
st = [torch.zeros(B * S, device=dev) for _ in range(4)]
sync = torch.zeros(32, dtype=torch.int32, device=dev)


def tail_stamps():
    """the stamps the -DMST_TAIL_STAMPS build keeps in a device array of its own (mst_debug_tail_stamps)"""
    import ctypes as C
    from musicstyletransfer_amd import _lib
    buf = (C.c_uint32 * 32)()
    assert _lib.load().mst_debug_tail_stamps(buf) == 0
    return np.array(list(buf), dtype=np.int64)

flush = torch.zeros(64 << 20, dtype=torch.uint8, device=dev)
rows = []
for it in range(12):
    sync.zero_(); flush.add_(1)  # cold-ish caches, as inside a step
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    o.row_tail_fwd(row0(att), row0(xin), Wp, bp, g1, be1, W1, b1, W2, b2, g2, be2, row0(h1), row0(x1), row0(a), row0(h2), row0(x2),
                   st[0], st[1], st[2], st[3], sync[0:3], stat_stride=S, phys_stride=S, dropout_p=0.2, dropout_seed_ptr=seedp, site0=6)
    e1.record(); torch.cuda.synchronize()
    t = tail_stamps()[:12]
    inner = [((t[16 - 8] - t[2]) & 0xffffffff) / 100.0, ((t[17 - 8] - t[16 - 8]) & 0xffffffff) / 100.0, ((t[18 - 8] - t[17 - 8]) & 0xffffffff) / 100.0,
             ((t[19 - 8] - t[18 - 8]) & 0xffffffff) / 100.0, ((t[3] - t[19 - 8]) & 0xffffffff) / 100.0]
    rows.append((e0.elapsed_time(e1) * 1e3, (np.diff(t[:8]) & 0xffffffff) / 100.0, inner))
print("stamps (us): stage1 | barrier1 | stage2 | barrier2 | stage3 | barrier3 | stage4")
for tot, d, inner in rows[2:]:
    print("event %.1f us | " % tot + " ".join("%5.1f" % v for v in d) + " | sum %.1f" % d.sum() +
          "   stage 2: loads %.1f LN %.1f sync %.1f mfma %.1f finish %.1f" % tuple(inner))
