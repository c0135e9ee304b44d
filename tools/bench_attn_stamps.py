"""Phases of the resident attention forward kernel per workgroup (build with MST_EXTRA_FLAGS="attention.hip=-DMST_ATT_STAMPS"):
start -> [projection loop done] -> operands staged in LDS -> key-row statistics done -> outputs stored. configs[1] encoder shapes
(B 64, S 256, 8 heads of 32); argv[1] = 'fused' (mst_attn_qkv_fwd) or 'plain' (mst_attn_keysoftmax_fwd on a given qkv)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from musicstyletransfer_amd import ops as o, _lib
mode = sys.argv[1] if len(sys.argv) > 1 else "fused"
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
B, S, H, dh = 64, 256, 8, 32
D = H * dh
g = torch.Generator().manual_seed(1)
x = torch.randn(B * S, D, generator=g).to(torch.bfloat16).to(dev)
W = (torch.randn(3 * D, D, generator=g) / 16).to(torch.bfloat16).to(dev)
bias = torch.zeros(3 * D, device=dev)
qkv = torch.zeros(B * S, 3 * D, dtype=torch.bfloat16, device=dev)
lse = torch.zeros(2, B, H, S, device=dev)
out = torch.zeros(B * S, D, dtype=torch.bfloat16, device=dev)
km = torch.ones(B, S, dtype=torch.uint8, device=dev)
o.gemm_nt(x, W, qkv, K=D, bias=bias)
for _ in range(5):
    if mode == "fused":
        o.attn_qkv_fwd(x, W, bias, qkv, km, lse, out, B, S, H, dh, 0, D, 2 * D)
    else:
        o.attn_fwd(qkv, km, lse, out, B, S, H, dh, 0, D, 2 * D)
torch.cuda.synchronize()
lib = _lib.load()
buf = (C.c_uint64 * 8192)()
assert lib.mst_debug_att_stamps(buf) == 0
t = np.array(list(buf), dtype=np.int64).reshape(1024, 8)[: B * H]
t0 = t[:, 0].min()
us = lambda a: (a - t0) / 100.0
names = ["start", "projection loop done", "operands staged", "statistics done", "outputs stored"]
cols = [0, 1, 2, 3, 4] if mode == "fused" else [0, 2, 3, 4]
print(f"{mode}: {B * H} workgroups, launch span {us(t[:, 4].max()):.1f} us")
prev = None
for c in cols:
    v = us(t[:, c])
    line = f"  {names[c]:24s} at median {np.median(v):6.1f} (min {v.min():6.1f}, max {v.max():6.1f}) us"
    if prev is not None:
        d = us(t[:, c]) - us(t[:, prev])
        line += f"   phase: median {np.median(d):5.1f}  p90 {np.percentile(d, 90):5.1f} us"
    print(line)
    prev = c

if mode == "fused" and hasattr(lib, "mst_debug_att_loop"):
    lb = (C.c_uint64 * 256)()
    if lib.mst_debug_att_loop(lb) == 0:
        L = np.array(list(lb), dtype=np.int64).reshape(4, 64)
        # s_memtime counts shader-engine clocks: report cycles between stamps
        for w, name in enumerate(("wg 0", "wg 100", "wg 300", "wg 500")):
            v = L[w]
            print(f"  {name}: first slice staged in {v[1] - v[0]} cycles; per slice (issue loads | MFMAs | barrier | LDS store | barrier):")
            for kc in range(4):
                b0 = 2 + 5 * kc
                prev = v[b0 - 1] if kc == 0 else v[b0 - 1]
                print("     slice %d: %6d | %6d | %6d | %6d | %6d" % (kc, v[b0] - prev, v[b0 + 1] - v[b0], v[b0 + 2] - v[b0 + 1], v[b0 + 3] - v[b0 + 2], v[b0 + 4] - v[b0 + 3]))
