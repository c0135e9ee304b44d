"""GPU micro-benchmark of the key-row-softmax attention entry points on the step's shapes (not a test).
usage: bench_attn.py [iters]   (iters=1 is the mode used under rocprofv3 --pmc)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from musicstyletransfer_amd import ops as o
BF = torch.bfloat16
dev = torch.device("cuda", 0)


def case(B, S, H, dh, iters, q_limit=0):
    D = H * dh
    g = torch.Generator(device="cpu").manual_seed(5)
    qkv = (torch.randn(B * S, 3 * D, generator=g) * 0.5).to(dev).to(BF)
    dout = torch.randn(B * S, D, generator=g).to(dev).to(BF)
    lens = torch.full((B,), S, dtype=torch.int32, device=dev)
    km = torch.zeros(B, S, dtype=torch.uint8, device=dev); o.mask_from_lengths(lens, 0, km)
    lse = torch.zeros(2, B, H, S, device=dev); out = torch.zeros(B * S, D, dtype=BF, device=dev)
    dqkv = torch.zeros(B * S, 3 * D, dtype=BF, device=dev); delta = torch.zeros(B, H, S, device=dev)
    fwd = lambda: o.attn_fwd(qkv, km, lse, out, B, S, H, dh, 0, D, 2 * D, q_limit=q_limit)
    if q_limit:
        dout.view(B, S, D)[:, q_limit:] = 0
    bwd = lambda: o.attn_bwd(qkv, km, lse, dout, dqkv, delta, B, S, H, dh, 0, D, 2 * D, q_limit=q_limit)
    res = []
    for fn in (fwd, bwd):
        fn(); torch.cuda.synchronize()
        e0, e1 = o.Event(), o.Event()
        e0.record()
        for _ in range(iters): fn()
        e1.record(); e1.sync()
        res.append(e0.elapsed_ms(e1) / iters * 1e3)
    print(f"B{B} S{S} H{H} dh{dh} q_limit={q_limit}: fwd {res[0]:7.1f} us  bwd {res[1]:7.1f} us")


if __name__ == "__main__":
    it = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    case(64, 256, 8, 32, it)
    case(64, 256, 8, 32, it, q_limit=1)
    case(64, 257, 8, 16, it)
