"""Attention launch times with padded batches (lengths uniform in [S/2, S]) against full-length ones: the tiles that hold a
padded key take the EXACT probability path (attention.hip: key_consts / exact_prob). Not a test."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from musicstyletransfer_amd import ops as o
BF = torch.bfloat16
dev = torch.device("cuda", 0)


def case(B, S, H, dh, padded, iters=30):
    D = H * dh
    g = torch.Generator(device="cpu").manual_seed(5)
    qkv = (torch.randn(B * S, 3 * D, generator=g) * 0.5).to(dev).to(BF)
    dout = torch.randn(B * S, D, generator=g).to(dev).to(BF)
    lens = (torch.randint(S // 2, S + 1, (B,), generator=g) if padded else torch.full((B,), S)).to(torch.int32).to(dev)
    km = torch.zeros(B, S, dtype=torch.uint8, device=dev); o.mask_from_lengths(lens, 0, km)
    lse = torch.zeros(2, B, H, S, device=dev); out = torch.zeros(B * S, D, dtype=BF, device=dev)
    dqkv = torch.zeros(B * S, 3 * D, dtype=BF, device=dev); delta = torch.zeros(B, H, S, device=dev)
    fwd = lambda: o.attn_fwd(qkv, km, lse, out, B, S, H, dh, 0, D, 2 * D)
    bwd = lambda: o.attn_bwd(qkv, km, lse, dout, dqkv, delta, B, S, H, dh, 0, D, 2 * D)
    res = []
    for fn in (fwd, bwd):
        fn(); torch.cuda.synchronize()
        e0, e1 = o.Event(), o.Event()
        e0.record()
        for _ in range(iters): fn()
        e1.record(); e1.sync()
        res.append(e0.elapsed_ms(e1) / iters * 1e3)
    print(f"B{B} S{S} H{H} dh{dh} {'padded' if padded else 'full  '}: fwd {res[0]:7.1f} us  bwd {res[1]:7.1f} us")


for shape in ((64, 256, 8, 32), (64, 257, 8, 16), (32, 1024, 8, 32)):
    case(*shape, False)
    case(*shape, True)
