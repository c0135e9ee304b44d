"""GPU micro-benchmark of the GEMM entry points on the step's shapes (not a test).
usage: bench_gemm.py [iters] [wgrad]   (second argument: only the weight-gradient batch, e.g. under rocprofv3 --pmc)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from musicstyletransfer_amd import ops as o
BF = torch.bfloat16
dev = torch.device("cuda", 0)

ITERS = int(sys.argv[1]) if len(sys.argv) > 1 else 30
ONLY_WGRAD = len(sys.argv) > 2 and sys.argv[2] == "wgrad"
ONLY = sys.argv[2] if len(sys.argv) > 2 and not ONLY_WGRAD else None  # substring filter on the GEMM cases


def timeit(fn, iters=None):
    iters = iters or ITERS
    fn(); torch.cuda.synchronize()
    e0, e1 = o.Event(), o.Event()
    e0.record()
    for _ in range(iters): fn()
    e1.record(); e1.sync()
    return e0.elapsed_ms(e1) / iters * 1e3

def run():
    M = 16384
    print(f"{'case':44s} {'us':>8s} {'TFLOP/s':>8s} {'GB/s(alg)':>10s}")
    for name, N, K, kw in [] if ONLY_WGRAD else [("qkv fwd  N768 K256 bias", 768, 256, dict(bias=True)),
                           ("proj fwd N256 K256 bias+resid", 256, 256, dict(bias=True, resid=True)),
                           ("ff1 fwd  N1024 K256 bias+relu", 1024, 256, dict(bias=True, relu=True)),
                           ("ff1 fwd  + dropout 0.2", 1024, 256, dict(bias=True, relu=True, drop=0.2)),
                           ("ff2 fwd  N256 K1024 bias+resid", 256, 1024, dict(bias=True, resid=True)),
                           ("ff2 dgrad N1024 K256 gate", 1024, 256, dict(gate=True)),
                           ("ff1 dgrad N256 K1024 resid", 256, 1024, dict(resid=True)),
                           ("qkv dgrad N256 K768 resid", 256, 768, dict(resid=True)),
                           ("dec ff1 N512 K128", 512, 128, dict(bias=True, relu=True)),
                           ("dec proj N128 K128", 128, 128, dict(bias=True, resid=True)),
                           ("plain N256 K256", 256, 256, dict())]:
        if ONLY and ONLY not in name:
            continue
        A = torch.randn(M, K, device=dev).to(BF); W = torch.randn(N, K, device=dev).to(BF) * 0.05
        C = torch.zeros(M, N, dtype=BF, device=dev)
        bias = torch.randn(N, device=dev) if kw.get("bias") else None
        resid = torch.randn(M, N, device=dev).to(BF) if kw.get("resid") else None
        gate = torch.randn(M, N, device=dev).to(BF) if kw.get("gate") else None
        seedp = torch.zeros(3, dtype=torch.int64, device=dev)
        fn = lambda: o.gemm_nt(A, W, C, bias=bias, resid=resid, gate=gate, act=o.ACT_RELU if kw.get("relu") else o.ACT_NONE,
                               dropout_p=kw.get("drop", 0.0), dropout_seed_ptr=seedp if kw.get("drop") else None)
        us = timeit(fn)
        byts = 2 * (M * K + M * N + N * K) + (2 * M * N if resid is not None else 0) + (2 * M * N if gate is not None else 0)
        print(f"{name:44s} {us:8.1f} {2*M*N*K/us/1e6:8.1f} {byts/us/1e3:10.0f}")
    if ONLY:
        return
    # wgrad: one encoder layer
    D = 256
    dh, a = torch.randn(M, D, device=dev).to(BF), torch.randn(M, 4 * D, device=dev).to(BF)
    dpre, x1 = torch.randn(M, 4 * D, device=dev).to(BF), torch.randn(M, D, device=dev).to(BF)
    dqkv = torch.randn(M, 3 * D, device=dev).to(BF)
    g = [torch.zeros(D, 4 * D, device=dev), torch.zeros(4 * D, D, device=dev), torch.zeros(D, D, device=dev), torch.zeros(3 * D, D, device=dev)]
    b = [torch.zeros(D, device=dev), torch.zeros(4 * D, device=dev), torch.zeros(D, device=dev), torch.zeros(3 * D, device=dev)]
    fn = lambda: o.gemm_wgrad_batch([o.wgrad_problem(dh, a, g[0], b[0]), o.wgrad_problem(dpre, x1, g[1], b[1]),
                                     o.wgrad_problem(dh, x1, g[2], b[2]), o.wgrad_problem(dqkv, x1, g[3], b[3])])
    us = timeit(fn)
    fl = 2 * M * 12 * D * D
    print(f"{'wgrad enc layer (4 problems)':44s} {us:8.1f} {fl/us/1e6:8.1f} {2*M*(D*2+4*D*2+3*D+D*2)/us/1e3:10.0f}")
if __name__ == "__main__":
    run()
