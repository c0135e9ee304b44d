"""GPU micro-benchmark of the small (launch / latency bound) kernels of the step at configs[1] sizes (not a test)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from musicstyletransfer_amd import ops as o
BF = torch.bfloat16
dev = torch.device("cuda", 0)
ITERS = int(sys.argv[1]) if len(sys.argv) > 1 else 50


REP = 20  # copies of the call inside one captured graph: per-node time as in the captured training step


def timeit(name, fn):
    fn(); torch.cuda.synchronize()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        g = o.Graph().capture(lambda: [fn() for _ in range(REP)])
        g.launch(); st.synchronize()
        e0, e1 = o.Event(), o.Event()
        e0.record()
        for _ in range(ITERS): g.launch()
        e1.record(); e1.sync()
    print(f"{name:34s} {e0.elapsed_ms(e1) / ITERS / REP * 1e3:8.1f} us per call (graph of {REP})")


def main():
    B, T, P, Z, De, Dd, C = 64, 256, 128, 64, 256, 128, 4
    g = torch.Generator().manual_seed(3)
    r = lambda *s: torch.randn(*s, generator=g).to(dev)
    enc = r(B, T, De).to(BF); dec_in = torch.zeros(B, T + 1, Dd, dtype=BF, device=dev)
    Wl, bl, Wh, bh = r(2 * Z, De) * 0.05, r(2 * Z), r(Dd, Z) * 0.1, r(Dd)
    eps, cls_d, pos_d = r(B, Z), r(C, Dd), r(T + 1, Dd)
    classes = torch.randint(0, C, (B,), generator=g).to(torch.int32).to(dev)
    mu, sigma, z = (torch.zeros(B, Z, device=dev) for _ in range(3))
    kl = torch.zeros(B, device=dev)
    timeit("latent_fwd", lambda: o.latent_fwd(enc, Wl, bl, eps, Wh, bh, classes, cls_d, pos_d, 11.3, mu, sigma, z, kl, dec_in))
    sigma.abs_().add_(0.5)
    d_dec = r(B, T + 1, Dd).to(BF); d_enc = torch.zeros(B, T, De, dtype=BF, device=dev)
    dWl, dbl, dWh, dbh, dcls = torch.zeros_like(Wl), torch.zeros_like(bl), torch.zeros_like(Wh), torch.zeros_like(bh), torch.zeros_like(cls_d)
    scratch = torch.zeros(B * (Dd + 2 * Z), device=dev)
    timeit("latent_bwd (all launches)", lambda: o.latent_bwd(enc, Wl, eps, Wh, classes, mu, sigma, z, d_dec, 11.3, 1.0, 1.0, dWl, dbl,
                                                             dWh, dbh, dcls, d_enc, scratch))
    logits = r(B * T, P).to(BF); labels = (torch.rand(B * T, P, generator=g) < 0.05).to(torch.uint8).to(dev)
    recon = torch.zeros(B, device=dev); dlog = torch.zeros(B * T, P, dtype=BF, device=dev)
    timeit("sigmoid_bce (+dlogits)", lambda: o.sigmoid_bce(logits, labels, recon, B, T, P, dlogits=dlog))
    total, macc = torch.zeros(B, device=dev), torch.zeros(3, device=dev)
    timeit("loss_combine", lambda: o.loss_combine(recon, kl, 1.0, total, macc))
    dx = r(B, T, De).to(BF); dcls_e = torch.zeros(C, De, device=dev)
    timeit("group_colsum", lambda: o.group_colsum(dx, T, De, 0, classes, dcls_e, 16.0))
    rng = torch.tensor([0, 0, 5, 0], dtype=torch.int64, device=dev); adam = torch.zeros(2, dtype=torch.int32, device=dev)
    lens = torch.full((B,), T, dtype=torch.int32, device=dev)
    me, md = torch.zeros(B, T, dtype=torch.uint8, device=dev), torch.zeros(B, T + 1, dtype=torch.uint8, device=dev)
    timeit("step_begin", lambda: o.step_begin(rng_state=rng, adam_state=adam, lr=1e-3, eps_out=eps, lens=lens, mask_e=me, mask_d=md))
    gbuf = torch.zeros(1885440, device=dev)
    timeit("zero(grad bucket 7.5 MB)", lambda: o.zero(gbuf))
    M, D = 16384, 256
    x = r(M, D).to(BF); dy = r(M, D).to(BF); gam = r(D); bet = r(D)
    yl = torch.zeros(M, D, dtype=BF, device=dev); mean = torch.zeros(M, device=dev); rstd = torch.zeros(M, device=dev)
    timeit("layernorm_fwd 16384x256", lambda: o.layernorm_fwd(x, gam, bet, yl, mean, rstd))
    dxl, dxm = torch.zeros(M, D, dtype=BF, device=dev), torch.zeros(M, D, dtype=BF, device=dev)
    dg, db = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
    seedp = torch.zeros(4, dtype=torch.int64, device=dev)
    timeit("layernorm_bwd 16384x256 mask_mode 0", lambda: o.layernorm_bwd(x, gam, mean, rstd, dy, dxl, dg, db))
    timeit("layernorm_bwd 16384x256 mask_mode 1", lambda: o.layernorm_bwd(x, gam, mean, rstd, dy, dxl, dg, db, dx_masked=dxm, mask_mode=1,
                                                                         dropout_p=0.2, dropout_seed_ptr=seedp, dropout_site=1))
    y = torch.zeros(1024, dtype=BF, device=dev); a1 = torch.zeros(1024, dtype=BF, device=dev)
    timeit("add_act(1024)  [launch floor]", lambda: o.add_act(a1, a1, y))


if __name__ == "__main__":
    main()
