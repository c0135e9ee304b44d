import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from musicstyletransfer_amd import ops as o
from tests.test_kernels_gpu import attn_reference, rnd
BF = torch.bfloat16
gpu = torch.device("cuda", 0)
def cos(a, b):
    a = a.double().flatten(); b = b.double().flatten()
    return float((a @ b) / (a.norm() * b.norm() + 1e-300))
def run(B, S, H, dh, ragged, scale=1.0, dt=BF):
    D = H * dh
    qkv = rnd((B * S, 3 * D), gpu, seed=30, scale=scale, dtype=dt)
    lens = torch.tensor([S - ((i * 7) % max(1, S // 2) if ragged else 0) for i in range(B)], dtype=torch.int32, device=gpu)
    km = torch.zeros(B, S, dtype=torch.uint8, device=gpu); o.mask_from_lengths(lens, 0, km)
    lse = torch.zeros(2, B, H, S, device=gpu); out = torch.zeros(B * S, D, dtype=dt, device=gpu)
    o.attn_fwd(qkv, km, lse, out, B, S, H, dh, 0, D, 2 * D)
    x = qkv.float().clone().requires_grad_(True)
    ref, _ = attn_reference(x, km, B, S, H, dh, 0, D, 2 * D)
    dout = rnd((B * S, D), gpu, seed=31, dtype=dt)
    dqkv = torch.zeros(B * S, 3 * D, dtype=dt, device=gpu); delta = torch.zeros(B, H, S, device=gpu)
    o.attn_bwd(qkv, km, lse, dout, dqkv, delta, B, S, H, dh, 0, D, 2 * D)
    torch.cuda.synchronize()
    ref.backward(dout.float()); g = x.grad
    parts = {"K": (0, D), "Q": (D, 2 * D), "V": (2 * D, 3 * D)}
    msg = f"B{B} S{S} H{H} dh{dh} ragged={ragged} scale={scale} {str(dt)[6:]}: out cos {cos(out.float(), ref.detach()):.5f}"
    for n, (a, b) in parts.items():
        gg, rr = dqkv[:, a:b].float(), g[:, a:b]
        msg += f" | d{n} cos {cos(gg, rr):.5f} relmax {float((gg-rr).abs().max()/rr.abs().max()):.3f} |ref| {float(rr.abs().max()):.3g}"
    print(msg)
for args in [(2, 64, 2, 32, False), (2, 64, 2, 32, True), (3, 24, 2, 16, False), (3, 24, 2, 16, True), (6, 24, 2, 16, True), (2, 257, 8, 16, True),
             (5, 20, 2, 16, True), (2, 256, 8, 32, True)]:
    run(*args)
run(3, 24, 2, 16, True, scale=0.3)
run(3, 24, 2, 16, True, scale=3.0)
run(3, 24, 2, 16, True, dt=torch.float16)
