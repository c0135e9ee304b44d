"""One launch of each attention entry point for rocprofv3 --pmc: resident (S 256) and streaming (S 1024, configs[4]'s shape) forms."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from musicstyletransfer_amd import ops as o
dev = torch.device("cuda", 0)


def case(B, S, H, dh, dtype):
    D = H * dh
    g = torch.Generator().manual_seed(5)
    qkv = (torch.randn(B * S, 3 * D, generator=g) * 0.5).to(dev).to(dtype)
    dout = torch.randn(B * S, D, generator=g).to(dev).to(dtype)
    km = torch.ones(B, S, dtype=torch.uint8, device=dev)
    lse = torch.zeros(2, B, H, S, device=dev); out = torch.zeros(B * S, D, dtype=dtype, device=dev)
    dqkv = torch.zeros(B * S, 3 * D, dtype=dtype, device=dev); delta = torch.zeros(B, H, S, device=dev)
    for _ in range(3):
        o.attn_fwd(qkv, km, lse, out, B, S, H, dh, 0, D, 2 * D)
        o.attn_bwd(qkv, km, lse, dout, dqkv, delta, B, S, H, dh, 0, D, 2 * D)
    torch.cuda.synchronize()


case(64, 256, 8, 32, torch.bfloat16)
case(64, 257, 8, 16, torch.bfloat16)  # the decoder of configs[1]
case(32, 1024, 8, 32, torch.float16)
