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


GRAPH = os.environ.get("BENCH_GRAPH", "1") != "0"  # time a captured graph of 10 calls: per-node time as in the step


def timeit(fn, iters=None):
    iters = iters or ITERS
    if GRAPH:
        fn(); torch.cuda.synchronize()
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            g = o.Graph().capture(lambda: [fn() for _ in range(10)])
            g.launch(); st.synchronize()
            e0, e1 = o.Event(), o.Event()
            e0.record()
            for _ in range(iters): g.launch()
            e1.record(); e1.sync()
        return e0.elapsed_ms(e1) / iters / 10 * 1e3
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
    # Dense + LayerNorm: unfused pair vs the fused launch
    for name, M2, N, K in [("proj+LN N256 K256", M, 256, 256), ("ff2+LN N256 K1024", M, 256, 1024), ("dec proj+LN N128 K128", 16448, 128, 128),
                           ("dec ff2+LN N128 K512", 16448, 128, 512), ("top ff2+LN M64 N256 K1024", 64, 256, 1024)]:
        if ONLY and ONLY not in name:
            continue
        A = torch.randn(M2, K, device=dev).to(BF); W = torch.randn(N, K, device=dev).to(BF) * 0.05
        h = torch.zeros(M2, N, dtype=BF, device=dev); y = torch.zeros(M2, N, dtype=BF, device=dev)
        bias = torch.randn(N, device=dev); resid = torch.randn(M2, N, device=dev).to(BF)
        gam, bet = torch.ones(N, device=dev), torch.zeros(N, device=dev)
        mean, rstd = torch.zeros(M2, device=dev), torch.zeros(M2, device=dev)
        def unfused():
            o.gemm_nt(A, W, h, bias=bias, resid=resid); o.layernorm_fwd(h, gam, bet, y, mean, rstd)
        fused = lambda: o.gemm_nt_ln_fwd(A, W, h, gam, bet, y, mean, rstd, bias=bias, resid=resid)
        print(f"{name:44s} unfused {timeit(unfused)*1:8.1f} us   fused {timeit(fused):8.1f} us")
    for name, M2, N, K, mode in [("ff1dgrad+LNbwd N256 K1024 m1", M, 256, 1024, 1), ("qkvdgrad+LNbwd N256 K768 m1", M, 256, 768, 1),
                                 ("dec ff1dgrad+LNbwd N128 K512 m1", 16448, 128, 512, 1), ("dec out-dgrad+LNbwd N128 K128 m2", 16448, 128, 128, 2)]:
        if ONLY and ONLY not in name:
            continue
        A = torch.randn(M2, K, device=dev).to(BF); W = torch.randn(N, K, device=dev).to(BF) * 0.05
        dy = torch.zeros(M2, N, dtype=BF, device=dev); dx = torch.zeros(M2, N, dtype=BF, device=dev); dxm = torch.zeros(M2, N, dtype=BF, device=dev)
        x = torch.randn(M2, N, device=dev).to(BF); resid = torch.randn(M2, N, device=dev).to(BF)
        gam = torch.ones(N, device=dev); dg, db = torch.zeros(N, device=dev), torch.zeros(N, device=dev)
        mean, rstd = torch.zeros(M2, device=dev), torch.ones(M2, device=dev)
        seedp = torch.zeros(4, dtype=torch.int64, device=dev)
        kw = dict(dx_masked=dxm if mode == 1 else None, mask_mode=mode, dropout_p=0.2, dropout_seed_ptr=seedp, dropout_site=1)
        def unfused():
            o.gemm_nt(A, W, dy, resid=resid); o.layernorm_bwd(x, gam, mean, rstd, dy, dx, dg, db, **kw)
        fused = lambda: o.gemm_nt_ln_bwd(A, W, dx, x, gam, mean, rstd, dg, db, resid=resid, **kw)
        print(f"{name:44s} unfused {timeit(unfused)*1:8.1f} us   fused {timeit(fused):8.1f} us")
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
