"""Phases of the resident attention BACKWARD kernel per workgroup (build with MST_EXTRA_FLAGS="attention.hip=-DMST_ATT_STAMPS"):
start -> Q / dO staged -> dV + delta of wave 0's key block -> dK (phase A done) -> K / V staged -> dQ stored.
argv[1] = 'enc' (B 64, S 256, 8 heads of 32), 'sparse' (same, q_limit 1) or 'dec' (S 257, 8 heads of 16)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from musicstyletransfer_amd import ops as o, _lib
mode = sys.argv[1] if len(sys.argv) > 1 else "enc"
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
B, S, H, dh, ql = {"enc": (64, 256, 8, 32, 0), "sparse": (64, 256, 8, 32, 1), "dec": (64, 257, 8, 16, 0)}[mode]
D = H * dh
g = torch.Generator().manual_seed(1)
qkv = (torch.randn(B * S, 3 * D, generator=g) * 0.5).to(torch.bfloat16).to(dev)
dout = torch.randn(B * S, D, generator=g).to(torch.bfloat16).to(dev)
if ql:
    dout.view(B, S, D)[:, ql:] = 0
lse = torch.zeros(2, B, H, S, device=dev)
out = torch.zeros(B * S, D, dtype=torch.bfloat16, device=dev)
dqkv = torch.zeros(B * S, 3 * D, dtype=torch.bfloat16, device=dev)
delta = torch.zeros(B, H, S, device=dev)
km = torch.ones(B, S, dtype=torch.uint8, device=dev)
o.attn_fwd(qkv, km, lse, out, B, S, H, dh, 0, D, 2 * D, q_limit=ql)
for _ in range(5):
    o.attn_bwd(qkv, km, lse, dout, dqkv, delta, B, S, H, dh, 0, D, 2 * D, q_limit=ql)
torch.cuda.synchronize()
lib = _lib.load()
buf = (C.c_uint64 * 8192)()
assert lib.mst_debug_att_stamps(buf) == 0
t = np.array(list(buf), dtype=np.int64).reshape(1024, 8)[: B * H]
t0 = t[:, 0].min()
us = lambda a: (a - t0) / 100.0
names = ["start", "Q / dO staged", "dV + delta (wave 0)", "phase A done (wave 0)", "K / V staged", "dQ stored (wave 0)"]
print(f"{mode}: {B * H} workgroups, launch span {us(t[:, 5].max()):.1f} us")
prev = None
for c in range(6):
    v = us(t[:, c])
    line = f"  {names[c]:24s} at median {np.median(v):6.1f} (min {v.min():6.1f}, max {v.max():6.1f}) us"
    if prev is not None:
        d = us(t[:, c]) - us(t[:, prev])
        line += f"   phase: median {np.median(d):5.1f}  p90 {np.percentile(d, 90):5.1f} us"
    print(line)
    prev = c
