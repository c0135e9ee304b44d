"""Diagnostic for tests/test_attention_flip_gpu.py: where does the q_limit = 1 backward differ from the reference in fp16?"""
import os, sys, math
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_attention_flip_gpu as F
from musicstyletransfer_amd import ops as o
gpu = torch.device("cuda", 0)
B, S, H, dh = 3, 250, 2, 32
D = H * dh
for dtype in (torch.float16, torch.bfloat16):
    qkv32, mask = F.integer_case(B, S, H, dh, seed=100 + S + dh)
    qr = qkv32.clone().requires_grad_(True)
    ref, logits, mass = F.reference(qr, mask, B, S, H, dh)
    dout32 = torch.from_numpy(np.random.default_rng(7).integers(-3, 4, size=(B * S, D)).astype(np.float32))
    keep = torch.zeros(B, S, 1); keep[:, :1] = 1
    dout32 = (dout32.view(B, S, D) * keep).reshape(B * S, D)
    ref.backward(dout32)
    g = qr.grad
    qkv = qkv32.to(dtype).to(gpu); keymask = mask.to(gpu)
    lse = torch.zeros(2, B, H, S, device=gpu); out = torch.zeros(B * S, D, dtype=dtype, device=gpu)
    o.attn_fwd(qkv, keymask, lse, out, B, S, H, dh, 0, D, 2 * D)
    res = {}
    for ql in (0, 1):
        dqkv = torch.zeros(B * S, 3 * D, dtype=dtype, device=gpu); delta = torch.zeros(B, H, S, device=gpu)
        o.attn_bwd(qkv, keymask, lse, dout32.to(dtype).to(gpu), dqkv, delta, B, S, H, dh, 0, D, 2 * D, q_limit=ql)
        torch.cuda.synchronize()
        res[ql] = (dqkv.float().cpu(), delta.cpu())
    print(dtype, "sparse == dense:", torch.equal(res[0][0], res[1][0]), "delta equal:", torch.equal(res[0][1], res[1][1]))
    bound = F.rounding_bounds(qkv32, dout32, logits.detach(), B, S, H, dh)
    ulp = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    for ql in (0, 1):
        a, b = res[ql][0][:, :D], g[:, :D]
        err = (a - b).abs(); tol = 3 * ulp * bound["dK"] + 2 * ulp * b.abs() + 1e-6
        r = err / tol
        idx = torch.nonzero(r > 1.0)
        print(" q_limit", ql, "worst", r.max().item(), "n", idx.shape[0])
        P = torch.softmax(logits.detach(), -1)
        with torch.no_grad():
            x3 = qkv32.view(B, S, 3 * D); hd = lambda t: t.reshape(B, S, H, dh).permute(0, 2, 1, 3)
            dP = torch.matmul(hd(x3[:, :, 2 * D:]), hd(dout32.view(B, S, D)).transpose(-1, -2))
            dlt = (P * dP).sum(-1)
        for (row, col) in idx[:12].tolist():
            bb, k = divmod(row, S); h = col // dh
            print(f"   b{bb} k{k} h{h} d{col % dh} pad={int(mask[bb, k] == 0)} got {a[row, col]:.5f} ref {b[row, col]:.5f} bound {bound['dK'][row, col]:.4g} "
                  f"P[k,0] {P[bb, h, k, 0]:.3e} Pmax {P[bb, h, k].max():.3e} delta_ref {dlt[bb, h, k]:.4e} delta_got {res[ql][1][bb, h, k]:.4e}")
