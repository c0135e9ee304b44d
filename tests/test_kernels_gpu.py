"""Per-kernel parity on a real MI355X: every C-ABI entry point against a plain torch fp32 reference of
the same operator on the same (16-bit-rounded) inputs. Tolerances are stated per test."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

BF = torch.bfloat16


def ops():
    from musicstyletransfer_amd import ops as o
    return o


def rnd(shape, dev, scale=1.0, dtype=BF, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dtype).to(dev)


def close(a, b, rtol, atol, what=""):
    a = a.float().cpu()
    b = b.float().cpu()
    err = (a - b).abs()
    tol = atol + rtol * b.abs()
    bad = (err > tol).sum().item()
    assert bad == 0, f"{what}: {bad}/{a.numel()} out of tolerance, max err {err.max().item():.4g}, ref max {b.abs().max().item():.4g}"


# ------------------------------------------------------------------------------------------ position-0 tail
def _rider_problem(gpu, dtype, Bg, T, N, K, with_resid, seed):
    """a GEMM of the kind the training step rides on the tails: rows 1..T of every T + 1 on both sides, bias or residual"""
    g = torch.Generator().manual_seed(seed)
    r = lambda *sh, sc=1.0, dt=dtype: (torch.randn(*sh, generator=g) * sc).to(dt).to(gpu)
    Md = Bg * (T + 1)
    A, W = r(Md, K), r(N, K, sc=0.08)
    kw = dict(M=Bg * T, N=N, K=K, a_remap=(T, T + 1, 1), c_remap=(T, T + 1, 1))
    if with_resid:
        kw.update(resid=r(Md, N), resid_phys=True)  # the residual shares C's physical rows, as in the step
    else:
        kw["bias"] = r(N, sc=0.1, dt=torch.float32)
    return A, W, kw, Md


def _check_rider(o, gpu, dtype, A, W, kw, Md, C_ride, sync_word):
    ref = torch.full((Md, kw["N"]), 3.0, dtype=dtype, device=gpu)
    o.gemm_nt(A, W, ref, **kw)
    torch.cuda.synchronize()
    T = kw["a_remap"][0]
    assert torch.equal(C_ride, ref), "the riding GEMM must equal mst_gemm_nt bit for bit (row 0 of every sample untouched)"
    want = A.float() @ W.float().t() + (kw["resid"].float() if "resid" in kw else kw["bias"])  # physical rows on every operand
    T_ = kw["a_remap"][0]
    close(C_ride.view(-1, T_ + 1, kw["N"])[:, 1:], want.view(-1, T_ + 1, kw["N"])[:, 1:], 2e-2, 2e-2, "riding GEMM vs torch fp32")
    tiles = (kw["M"] // 128) * (kw["N"] // 128)
    if tiles > 7 * 32 - 16 and kw["M"] % 256 == 0 and T % 256 == 0:
        tiles //= 2  # (row_tail.hip ride_shape: 256-row tiles when the 128-row ones would need a second round of riders)
    assert int(sync_word.item()) >= tiles  # every tile was handed out (each workgroup that finds the queue empty adds one more)
    assert (C_ride.view(-1, T + 1, kw["N"])[:, 0] == 3.0).all()


@pytest.mark.parametrize("rider", [None, (6, 128, 384, 128), (64, 256, 384, 128), (40, 128, 1024, 256)],
                         ids=["plain", "rider-small", "rider-step", "rider-many-tiles"])
@pytest.mark.parametrize("B,S,D,p,dtype", [(64, 8, 256, 0.2, BF), (5, 3, 128, 0.0, BF), (33, 4, 256, 0.2, torch.float16), (16, 2, 128, 0.2, BF)])
def test_row_tail_fwd_equals_the_five_launches(gpu, B, S, D, p, dtype, rider):
    """mst_row_tail_fwd (W_proj + LN1 + FFN1 + FFN2 + LN2 on the B position-0 rows, one launch with three grid barriers)
    against the five launches it replaces on the same strided rows: the same dropout counters and rounding points; the K
    sums and the LayerNorm sums run in another order (an ulp here and there)"""
    o = ops()
    F = 4 * D
    g = torch.Generator().manual_seed(7)
    r = lambda *sh, sc=1.0, dt=dtype: (torch.randn(*sh, generator=g) * sc).to(dt).to(gpu)
    att, xin = r(B * S, D), r(B * S, D)
    Wp, W1, W2 = r(D, D, sc=0.06), r(F, D, sc=0.06), r(D, F, sc=0.03)
    bp, b1, b2 = r(D, sc=0.1, dt=torch.float32), r(F, sc=0.1, dt=torch.float32), r(D, sc=0.1, dt=torch.float32)
    g1, be1, g2, be2 = (1 + r(D, sc=0.1, dt=torch.float32)), r(D, sc=0.1, dt=torch.float32), (1 + r(D, sc=0.1, dt=torch.float32)), r(D, sc=0.1, dt=torch.float32)
    seedp = torch.tensor([99, 0, 0, 0], dtype=torch.int64, device=gpu)
    row0 = lambda t: t.view(B, S, -1)[:, 0, :]

    def bufs():
        z = lambda w: torch.zeros(B * S, w, dtype=dtype, device=gpu)
        return dict(h1=z(D), x1=z(D), a=z(F), h2=z(D), x2=z(D), m1=torch.zeros(B * S, device=gpu), r1=torch.zeros(B * S, device=gpu),
                    m2=torch.zeros(B * S, device=gpu), r2=torch.zeros(B * S, device=gpu))

    dk = lambda site: dict(dropout_p=p, dropout_seed_ptr=seedp, dropout_site=site) if p > 0 else {}
    u, rows = bufs(), (1, S, 0)
    o.gemm_nt(row0(att), Wp, u["h1"], M=B, N=D, K=D, bias=bp, resid=row0(xin), c_remap=rows, **dk(6))
    o.layernorm_fwd(row0(u["h1"]), g1, be1, row0(u["x1"]), u["m1"], u["r1"], D=D, M=B, row_id_stride=S)
    o.gemm_nt(row0(u["x1"]), W1, u["a"], M=B, K=D, bias=b1, act=o.ACT_RELU, c_remap=rows, **dk(7))
    o.gemm_nt(row0(u["a"]), W2, u["h2"], M=B, K=F, bias=b2, resid=row0(u["x1"]), c_remap=rows, **dk(8))
    o.layernorm_fwd(row0(u["h2"]), g2, be2, row0(u["x2"]), u["m2"], u["r2"], D=D, M=B, row_id_stride=S)
    f = bufs()
    sync = torch.zeros(8, dtype=torch.int32, device=gpu)
    queue = torch.zeros(64, dtype=torch.int32, device=gpu)[32:]
    ride = None
    if rider is not None:
        # mst_row_tail_fwd_ride: the workgroups on the other seven XCDs compute a GEMM of their own (the decoder's K | Q | V
        # projection in the step); "many tiles" outlasts the chain, so its participants drain the queue at the end
        rA, rW, rkw, rMd = _rider_problem(gpu, dtype, *rider, with_resid=False, seed=17)
        rC = torch.full((rMd, rkw["N"]), 3.0, dtype=dtype, device=gpu)
        ride = dict(A=rA, B=rW, C_out=rC, **rkw)
    sh = None
    if rider is not None and rider[0] != 6:
        # ... and behind the GEMM's tiles, in the same queue, the transposed-shadow refresh (mst_row_tail_fwd_ride_shadows): 32 x 32
        # tiles of three matrices, ragged ones included, four per ticket
        sh_shapes = [(10, 32), (293, 128), (256, 1024)]
        sh_offs, tot = [], 0
        for rr, cc in sh_shapes:
            sh_offs.append(tot)
            tot += rr * cc
        sh_w = (torch.randn(tot, generator=g)).to(gpu)
        desc, prefix, doff = [], [0], 0
        for (rr, cc), so in zip(sh_shapes, sh_offs):
            desc += [so, doff, rr, cc]
            doff += cc * o.roundup(rr, 8)
            prefix.append(prefix[-1] + ((rr + 31) // 32) * ((cc + 31) // 32))
        sh_wt = torch.full((doff,), 9.0, dtype=dtype, device=gpu)
        sh = dict(w=sh_w, wt16=sh_wt, desc=torch.tensor(desc, dtype=torch.int64, device=gpu),
                  prefix=torch.tensor(prefix, dtype=torch.int64, device=gpu), n_mat=len(sh_shapes), tiles=prefix[-1])
    o.row_tail_fwd(row0(att), row0(xin), Wp, bp, g1, be1, W1, b1, W2, b2, g2, be2, row0(f["h1"]), row0(f["x1"]), row0(f["a"]), row0(f["h2"]),
                   row0(f["x2"]), f["m1"], f["r1"], f["m2"], f["r2"], sync[0:3], stat_stride=S, phys_stride=S,
                   dropout_p=p, dropout_seed_ptr=seedp if p > 0 else None, site0=6, rider=ride, queue=queue[0:1], shadows=sh)
    torch.cuda.synchronize()
    assert int(sync[0].item()) == 3 * (D // 16)  # three barriers, every workgroup arrived at each
    if ride:
        _check_rider(o, gpu, dtype, rA, rW, rkw, rMd, rC, queue[0])
    if sh is not None:
        d0 = 0
        for (rr, cc), so in zip(sh_shapes, sh_offs):
            ldt = o.roundup(rr, 8)
            got = sh_wt[d0:d0 + cc * ldt].view(cc, ldt)
            assert torch.equal(got[:, :rr], sh_w[so:so + rr * cc].view(rr, cc).t().to(dtype)), "shadow tiles behind the riding GEMM"
            assert (got[:, rr:] == 0).all()
            d0 += cc * ldt
        assert int(queue[0].item()) >= (sh["tiles"] + 15) // 16  # (every ticket handed out)
    # (the 16-column stages sum their K range in four quarters, one per group of waves: another order of the same fp32 sums)
    ulp = 2.0 ** -7 if dtype == BF else 2.0 ** -10
    for k in ("h1", "x1", "a", "h2", "x2"):
        d = (row0(f[k]).float() - row0(u[k]).float()).abs()
        ref = row0(u[k]).float().abs().clamp(min=1.0)
        assert (d <= 2 * ulp * ref).all(), (k, d.max().item())
        assert (row0(f[k]) != row0(u[k])).float().mean().item() < 0.05, k
    for k in ("m1", "r1", "m2", "r2"):
        close(f[k][::S], u[k][::S], 2e-3, 2e-4, k)
    # rows other than position 0 are never touched
    assert (f["h1"].view(B, S, -1)[:, 1:] == 0).all() and (f["a"].view(B, S, -1)[:, 1:] == 0).all()


@pytest.mark.parametrize("rider", [None, (6, 128, 128, 384), (64, 256, 128, 384), (40, 128, 1024, 256)],
                         ids=["plain", "rider-small", "rider-step", "rider-many-tiles"])
@pytest.mark.parametrize("B,S,D,p,dtype", [(64, 256, 256, 0.2, BF), (64, 8, 128, 0.2, BF), (37, 4, 256, 0.0, BF), (64, 16, 256, 0.1, torch.float16)])
def test_row_tail_bwd_equals_the_five_launches(gpu, B, S, D, p, dtype, rider):
    """mst_row_tail_bwd (LayerNorm-2 backward, FFN2 / FFN1 dgrads, LayerNorm-1 backward, W_proj dgrad on the B position-0 rows in
    one launch with two grid barriers) against the five launches it replaces on the same strided rows: the same dropout counters
    and MFMA order; the LayerNorm sums run in another order, so results agree to a rounding of the activation type"""
    o = ops()
    F = 4 * D
    g = torch.Generator().manual_seed(11)
    r = lambda *sh, sc=1.0, dt=dtype: (torch.randn(*sh, generator=g) * sc).to(dt).to(gpu)
    dy, h2, h1 = r(B * S, D, sc=0.5), r(B * S, D), r(B * S, D)
    a = torch.relu(r(B * S, F))
    W2t, W1t, Wpt = r(F, D, sc=0.05), r(D, F, sc=0.05), r(D, D, sc=0.06)
    g1, g2 = 1 + r(D, sc=0.1, dt=torch.float32), 1 + r(D, sc=0.1, dt=torch.float32)
    row0 = lambda t: t.view(B, S, -1)[:, 0, :]
    m1, m2 = torch.zeros(B * S, device=gpu), torch.zeros(B * S, device=gpu)
    r1, r2 = torch.zeros(B * S, device=gpu), torch.zeros(B * S, device=gpu)
    m1[::S], m2[::S] = row0(h1).float().mean(1), row0(h2).float().mean(1)
    r1[::S] = 1.0 / torch.sqrt(row0(h1).float().var(1, unbiased=False) + 1e-5)
    r2[::S] = 1.0 / torch.sqrt(row0(h2).float().var(1, unbiased=False) + 1e-5)
    seedp = torch.tensor([99, 0, 0, 0], dtype=torch.int64, device=gpu)
    inv_keep = 1.0 / (1.0 - p) if p > 0 else 1.0
    dk = dict(dropout_p=p, dropout_seed_ptr=seedp) if p > 0 else {}

    def bufs():
        z = lambda n, w: torch.zeros(n, w, dtype=dtype, device=gpu)
        return dict(dh=z(B, D), dhm=z(B, D), dx1=z(B, D), dh1m=z(B, D), dpre=z(B, F), dh1=z(B * S, D), datt=z(B * S, D),
                    dg1=torch.zeros(D, device=gpu), db1=torch.zeros(D, device=gpu), dg2=torch.zeros(D, device=gpu), db2=torch.zeros(D, device=gpu))

    u = bufs()
    mk = dict(dx_masked=u["dhm"], mask_mode=1, dropout_site=8, **dk) if p > 0 else {}
    o.layernorm_bwd(row0(h2), g2, m2, r2, row0(dy), u["dh"], u["dg2"], u["db2"], D=D, M=B, row_id_stride=S, **mk)
    dff = u["dhm"] if p > 0 else u["dh"]
    o.gemm_nt(dff, W2t, u["dpre"], N=F, K=D, gate=row0(a), alpha=inv_keep)
    o.gemm_nt(u["dpre"], W1t, u["dx1"], N=D, K=F, resid=u["dh"])
    mk = dict(dx_masked=u["dh1m"], mask_mode=1, dropout_site=6, **dk) if p > 0 else {}
    o.layernorm_bwd(row0(h1), g1, m1, r1, u["dx1"], row0(u["dh1"]), u["dg1"], u["db1"], D=D, M=B, row_id_stride=S, **mk)
    dproj = u["dh1m"] if p > 0 else row0(u["dh1"])
    o.gemm_nt(dproj, Wpt, u["datt"], M=B, N=D, K=D, c_remap=(1, S, 0))
    f = bufs()
    sync = torch.zeros(8, dtype=torch.int32, device=gpu)
    queue = torch.zeros(64, dtype=torch.int32, device=gpu)[32:]
    ride = None
    if rider is not None:  # (mst_row_tail_bwd_ride: in the step, the input gradient of the decoder's K | Q | V projection, + residual)
        rA, rW, rkw, rMd = _rider_problem(gpu, dtype, *rider, with_resid=True, seed=19)
        rC = torch.full((rMd, rkw["N"]), 3.0, dtype=dtype, device=gpu)
        ride = dict(A=rA, B=rW, C_out=rC, **rkw)
    o.row_tail_bwd(row0(dy), row0(h2), row0(h1), row0(a), m1, r1, m2, r2, g1, g2, W2t, W1t, Wpt, f["dh"], f["dhm"], f["dx1"], f["dh1m"], f["dpre"],
                   row0(f["dh1"]), row0(f["datt"]), f["dg1"], f["db1"], f["dg2"], f["db2"], sync[4:7], stat_stride=S,
                   phys_stride=S, dropout_p=p, dropout_seed_ptr=seedp if p > 0 else None, site0=6, rider=ride, queue=queue[0:1])
    torch.cuda.synchronize()
    assert int(sync[4].item()) == 2 * (D // 16)  # two barriers, every participating workgroup arrived at each
    if ride:
        _check_rider(o, gpu, dtype, rA, rW, rkw, rMd, rC, queue[0])
    assert 1 <= int(sync[5].item()) <= 8 and int(sync[6].item()) >= D // 16  # one XCD claimed, enough workgroups found on it
    keys = ["dh", "dpre", "dx1", "dh1", "datt"] + (["dhm", "dh1m"] if p > 0 else [])
    for k in keys:
        sc = u[k].float().abs().max().item()
        assert sc > 0, k
        close(f[k], u[k], 2e-2, 1e-2 * sc, k)
        assert (f[k] != u[k]).float().mean().item() < (0.1 if dtype == BF else 0.4), k  # (fp16: finer grid, more last-bit flips)
    for k in ("dg1", "db1", "dg2", "db2"):
        close(f[k], u[k], 2e-3, 2e-3 * u[k].abs().max().item(), k)
    # rows other than position 0 are never touched
    assert (f["dh1"].view(B, S, -1)[:, 1:] == 0).all() and (f["datt"].view(B, S, -1)[:, 1:] == 0).all()


@pytest.mark.parametrize("preset", ["no_roles_left", "one_role_left"])
def test_row_tail_reports_a_launch_that_cannot_do_its_work(gpu, preset):
    """The one-launch tails' grid barrier needs G workgroups of the launch on one XCD. When that does not happen the kernel
    must SAY so (sticky status word, include/mst_hip.h MST_TAIL_SPIN_*, or a barrier counter left short) instead of carrying on silently: forced here by handing
    out roles before the launch — every role gone (nobody works, the barrier counter stays 0) and all but one gone (the
    lone participant waits out the 0.2 s bound of each barrier, flags it and returns instead of hanging)."""
    from musicstyletransfer_amd import _lib
    o = ops()
    B, S, D = 16, 2, 256
    F, G = 4 * D, D // 16
    g = torch.Generator().manual_seed(3)
    r = lambda *sh, sc=1.0, dt=BF: (torch.randn(*sh, generator=g) * sc).to(dt).to(gpu)
    att, xin = r(B * S, D), r(B * S, D)
    Wp, W1, W2 = r(D, D, sc=0.06), r(F, D, sc=0.06), r(D, F, sc=0.03)
    v = lambda n: r(n, sc=0.1, dt=torch.float32)
    row0 = lambda t: t.view(B, S, -1)[:, 0, :]
    z = lambda w: torch.zeros(B * S, w, dtype=BF, device=gpu)
    st = lambda: torch.zeros(B * S, device=gpu)
    h1, x1, a, h2, x2 = z(D), z(D), z(F), z(D), z(D)
    sync = torch.zeros(8, dtype=torch.int32, device=gpu)
    status = torch.zeros(2, dtype=torch.int32, device=gpu)
    sync[2] = G if preset == "no_roles_left" else G - 1
    o.row_tail_fwd(row0(att), row0(xin), Wp, v(D), 1 + v(D), v(D), W1, v(F), W2, v(D), 1 + v(D), v(D), row0(h1), row0(x1), row0(a),
                   row0(h2), row0(x2), st(), st(), st(), st(), sync[0:3], stat_stride=S, phys_stride=S, status=status[0:1])
    torch.cuda.synchronize()
    flags = int(status[0].item())
    if preset == "no_roles_left":
        # nobody worked and nobody could flag it: the barrier counter stays short of 3 * G, which is what the step guard of the
        # optimizer launch checks (test_step_guard_skips_the_update_and_the_metric_sums, MST_STEP_INCOMPLETE)
        assert int(sync[0].item()) == 0 and flags == 0 and (x2 == 0).all()
    else:
        assert flags == _lib.TAIL_SPIN_FWD, flags                   # the lone participant gave up waiting, three times
        assert int(sync[0].item()) == 3                               # (its own three arrivals; a full launch ends at 3 * G)
    # the status word is sticky: a healthy launch afterwards does not clear it
    sync.zero_()
    o.row_tail_fwd(row0(att), row0(xin), Wp, v(D), 1 + v(D), v(D), W1, v(F), W2, v(D), 1 + v(D), v(D), row0(h1), row0(x1), row0(a),
                   row0(h2), row0(x2), st(), st(), st(), st(), sync[0:3], stat_stride=S, phys_stride=S, status=status[0:1])
    torch.cuda.synchronize()
    assert int(sync[0].item()) == 3 * G and int(status[0].item()) == flags


def test_step_guard_skips_the_update_and_the_metric_sums(gpu):
    """mst_adam_flat / mst_loss_combine_v with the step guard of mst_step_metrics: a set status word or a device word that
    does not hold its expected value turns the launch into a no-op that counts the skipped step and takes the step count back"""
    from musicstyletransfer_amd import _lib
    o = ops()
    n, B = 4096, 8
    w0 = torch.randn(n, device=gpu)
    mk = lambda: dict(w=w0.clone(), g=torch.randn(n, device=gpu), m=torch.zeros(n, device=gpu), v=torch.zeros(n, device=gpu),
                      w16=torch.zeros(n, dtype=BF, device=gpu))
    recon, kl = torch.rand(B, device=gpu), torch.rand(B, device=gpu)
    total, metric = torch.zeros(B, device=gpu), torch.zeros(4, device=gpu)
    state = torch.tensor([5, 0], dtype=torch.int32, device=gpu)
    state.view(torch.float32)[1] = 1e-3
    word = torch.tensor([48], dtype=torch.int32, device=gpu)

    def run(status, expect_val):
        t = mk()
        o.adam_flat(t["w"], t["g"], t["m"], t["v"], t["w16"], state, lr=1e-3, advance_step=False,
                    metrics=dict(recon=recon, kl=kl, kl_weight=1.0, total=total, metric=metric[:3], status=status, expect=[(word, expect_val)]))
        torch.cuda.synchronize()
        return t

    status = torch.zeros(2, dtype=torch.int32, device=gpu)
    t = run(status, 48)                                   # healthy: the update happens, the sums are taken
    assert not torch.equal(t["w"], w0) and float(metric[2].item()) == B and status.tolist() == [0, 0] and int(state[0].item()) == 5
    t = run(status, 47)                                   # expectation fails
    assert torch.equal(t["w"], w0) and (t["m"] == 0).all() and float(metric[2].item()) == B
    assert status.tolist() == [_lib.STEP_INCOMPLETE, 1] and int(state[0].item()) == 4
    t = run(status, 48)                                   # sticky: still skipping although the expectation now holds
    assert torch.equal(t["w"], w0) and status.tolist() == [_lib.STEP_INCOMPLETE, 2] and int(state[0].item()) == 3
    o.loss_combine(recon, kl, 1.0, total, metric[:3], guard=dict(status=status, expect=[(word, 48)]))
    torch.cuda.synchronize()
    assert float(metric[2].item()) == B and status.tolist() == [_lib.STEP_INCOMPLETE, 3]
    status.zero_()
    o.loss_combine(recon, kl, 1.0, total, metric[:3], guard=dict(status=status, expect=[(word, 48)]))
    torch.cuda.synchronize()
    assert float(metric[2].item()) == 2 * B and status.tolist() == [0, 0]


# ------------------------------------------------------------------------------------------ layout
def test_selftest_layout_maps(gpu):
    flags = ops().selftest()
    assert flags == [1, 1, 1, 1], f"MFMA / ds_read_tr16_b64 lane maps differ from the ones the kernels assume: {flags}"


# ------------------------------------------------------------------------------------------ GEMM NT
@pytest.mark.parametrize("M,N,K", [(256, 256, 256), (16384, 256, 256), (1000, 96, 160), (15, 10, 32), (16448, 128, 128),
                                   (2048, 1024, 256), (512, 296, 128), (16384, 768, 256), (8192, 1536, 96)])  # (last two: 768 tiles of 128 x 128, the three-per-CU form with 32-deep stages)
def test_gemm_nt_plain(gpu, M, N, K):
    o = ops()
    A = rnd((M, K), gpu, seed=1)
    B = rnd((N, K), gpu, seed=2)
    ldc = o.roundup(N, 8)
    C = torch.full((M, ldc), 7.0, dtype=BF, device=gpu)
    o.gemm_nt(A, B, C, N=N)
    torch.cuda.synchronize()
    ref = A.float() @ B.float().t()
    # bf16 output rounding: 2^-8 relative; fp32 accumulation over K
    close(C[:, :N], ref, 1e-2, 1e-2 * math.sqrt(K), "gemm_nt")
    n4 = min(o.roundup(N, 4), ldc)
    assert (C[:, N:n4] == 0).all(), "pad columns up to roundup4(N) must be written as zeros"


@pytest.mark.parametrize("M,P,N,period,dtype", [(16384, 128, 256, 256, BF), (512, 2048, 128, 128, torch.float16), (300, 40, 64, 50, BF)])
def test_gemm_nt_and_wgrad_uint8_frames_equal_the_16bit_operand(gpu, M, P, N, period, dtype):
    """a_u8: piano-roll frames stay uint8 in HBM and are widened while tiles are staged into LDS — the embedding GEMM
    (row-indexed adds, fast and general epilogue) and the embedding tables' weight gradient must be BIT-identical to the
    same launches on the frames stored in the activation type"""
    o = ops()
    g = torch.Generator().manual_seed(5)
    ld8 = o.roundup(P, 8)
    frames = torch.zeros(M, ld8, dtype=torch.uint8)
    frames[:, :P] = (torch.rand(M, P, generator=g) < 0.05).to(torch.uint8)
    frames[0, :P] = torch.arange(P) % 7  # values other than {0,1} are widened exactly too
    f8 = frames.to(gpu)
    f16 = frames.to(dtype).to(gpu)
    table_t = rnd((N, ld8), gpu, 0.1, dtype, seed=6)  # [D, P]: the transposed shadow of the embedding table
    table_t[:, P:] = 0
    pos = rnd((period, N), gpu, 1.0, torch.float32, seed=7)
    cls = rnd((3, N), gpu, 1.0, torch.float32, seed=8)
    idx = torch.randint(0, 3, (M // period + 1,), generator=g).to(torch.int32).to(gpu)
    outs = []
    for A in (f8, f16):
        C = torch.zeros(M, N, dtype=dtype, device=gpu)
        o.gemm_nt(A, table_t, C, N=N, K=ld8, alpha=2.0, grpadd=cls, grp_index=idx, rowadd=pos, rowadd_period=period)
        outs.append(C)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1])
    ref = 2.0 * (frames[:, :P].float() @ table_t[:, :P].float().cpu().t() + cls.cpu()[idx.cpu().long()[torch.arange(M) // period]]) \
        + pos.cpu()[torch.arange(M) % period]
    close(outs[0], ref, 2e-2, 2e-2, "embedding GEMM on uint8 frames")
    dY = rnd((M, N), gpu, 1.0, dtype, seed=9)
    grads = []
    for A in (f8, f16):
        dW = torch.zeros(P, N, dtype=torch.float32, device=gpu)
        o.gemm_wgrad(A, dY, dW, N=P, K=N, scale=0.5)
        grads.append(dW)
    torch.cuda.synchronize()
    assert torch.allclose(grads[0], grads[1], rtol=0, atol=1e-5 * float(grads[1].abs().max()))  # (fp32 atomics: last bits only)
    close(grads[0], 0.5 * frames[:, :P].float().t() @ dY.float().cpu(), 1e-3, 1e-3 * math.sqrt(M), "embedding wgrad on uint8 frames")


@pytest.mark.parametrize("B,T,P,De,Dd,dtype", [(64, 256, 128, 256, 128, BF), (4, 64, 2048, 128, 64, torch.float16), (3, 50, 40, 64, 64, BF),
                                                (4, 128, 2048, 256, 128, BF)])  # (the last: long contraction, 128 x 128 tiles)
def test_gemm_nt_pair_equals_the_two_launches(gpu, B, T, P, De, Dd, dtype):
    """mst_gemm_nt_pair: the encoder's and the decoder's embedding GEMM (same uint8 frames, different tables, adds and output
    row remap) in one launch — bit-identical to the two launches, also where the form falls back to them (the ragged third case)"""
    o = ops()
    g = torch.Generator().manual_seed(15)
    M, ld8 = B * T, o.roundup(P + 2, 8)  # (two more byte columns behind the pitches, as the engine lays a frame out)
    frames = torch.zeros(M, ld8, dtype=torch.uint8)
    frames[:, :P] = (torch.rand(M, P, generator=g) < 0.05).to(torch.uint8)
    frames[:, P:] = 1
    f8 = frames.to(gpu)[:, : o.roundup(P, 8)]
    te, td = rnd((De, o.roundup(P, 8)), gpu, 0.1, dtype, seed=16), rnd((Dd, o.roundup(P, 8)), gpu, 0.1, dtype, seed=17)
    pos_e, pos_d = rnd((T, De), gpu, 1.0, torch.float32, seed=18), rnd((T + 1, Dd), gpu, 1.0, torch.float32, seed=19)
    cls = rnd((3, De), gpu, 1.0, torch.float32, seed=20)
    idx = torch.randint(0, 3, (B,), generator=g).to(torch.int32).to(gpu)

    def problems(xe, xd):
        return (dict(A=f8, B=te, C_out=xe, N=De, K=o.roundup(P, 8), alpha=1.5, grpadd=cls, grp_index=idx, rowadd=pos_e, rowadd_period=T),
                dict(A=f8, B=td, C_out=xd, M=M, N=Dd, K=o.roundup(P, 8), alpha=0.5, rowadd=pos_d[1:], rowadd_period=T, c_remap=(T, T + 1, 1)))

    outs = []
    for pair in (True, False):
        xe = torch.zeros(M, De, dtype=dtype, device=gpu)
        xd = torch.full((B * (T + 1), Dd), 7.0, dtype=dtype, device=gpu)
        first, second = problems(xe, xd)
        if pair:
            o.gemm_nt_pair(first, second)
        else:
            o.gemm_nt(first.pop("A"), first.pop("B"), first.pop("C_out"), **first)
            o.gemm_nt(second.pop("A"), second.pop("B"), second.pop("C_out"), **second)
        outs.append((xe, xd))
    torch.cuda.synchronize()
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert (outs[0][1].view(B, T + 1, Dd)[:, 0] == 7.0).all(), "row 0 of every sample belongs to the latent block"
    fr = frames[:, :P].float()
    ref_e = 1.5 * (fr @ te[:, :P].float().cpu().t() + cls.cpu()[idx.cpu().long()[torch.arange(M) // T]]) + pos_e.cpu()[torch.arange(M) % T]
    close(outs[0][0], ref_e, 2e-2, 2e-2, "encoder embedding of the pair")
    ref_d = 0.5 * (fr @ td[:, :P].float().cpu().t()) + pos_d.cpu()[1:][torch.arange(M) % T]
    close(outs[0][1].view(B, T + 1, Dd)[:, 1:].reshape(M, Dd), ref_d, 2e-2, 2e-2, "decoder embedding of the pair")


def test_wgrad_class_columns_behind_the_pitches_give_the_class_gradient(gpu):
    """StepPlan.cls_fold: the one-hot class id of a frame's sequence in C byte columns behind its pitches makes the class table's
    gradient (model.py:89: one class row added to every frame) rows P.. of the embedding's weight-gradient problem"""
    o = ops()
    g = torch.Generator().manual_seed(21)
    B, T, P, Cn, D = 8, 64, 128, 2, 256
    M, ld = B * T, o.roundup(P + Cn, 8)
    classes = torch.randint(0, Cn, (B,), generator=g)
    frames = torch.zeros(B, T, ld, dtype=torch.uint8)
    frames[:, :, :P] = (torch.rand(B, T, P, generator=g) < 0.05).to(torch.uint8)
    frames[:, :, P:] = (classes.view(B, 1, 1) == torch.arange(ld - P).view(1, 1, -1)).to(torch.uint8)
    dY = rnd((M, D), gpu, 1.0, BF, seed=22)
    dW = torch.zeros(P + Cn + 3, D, dtype=torch.float32, device=gpu)
    o.gemm_wgrad(frames.view(M, ld).to(gpu), dY, dW, N=P + Cn, K=D, scale=2.0)
    torch.cuda.synchronize()
    dy = dY.float().cpu().view(B, T, D)
    close(dW[:P], 2.0 * frames.view(M, ld)[:, :P].float().t() @ dy.view(M, D), 1e-3, 1e-3 * math.sqrt(M), "embedding rows")
    want = torch.stack([2.0 * dy[classes == c].sum((0, 1)) for c in range(Cn)])
    close(dW[P: P + Cn], want, 1e-3, 1e-3 * math.sqrt(M), "class rows")
    assert (dW[P + Cn:] == 0).all(), "rows beyond N are not touched"


def test_gemm_nt_integer_exact_asymmetric(gpu):
    """small-integer operands: exact in bf16 and fp32, catches any transposed / permuted fragment"""
    o = ops()
    M, N, K = 192, 160, 96
    i = torch.arange(M).view(-1, 1)
    k = torch.arange(K).view(1, -1)
    A = (((i * 3 + k * 5) % 7) - 3).to(BF).to(gpu)
    n = torch.arange(N).view(-1, 1)
    B = (((n * 2 + k * 7) % 5) - 2).to(BF).to(gpu)
    C = torch.zeros(M, N, dtype=torch.float32, device=gpu)
    o.gemm_nt(A, B, C)
    torch.cuda.synchronize()
    ref = A.float() @ B.float().t()
    assert torch.equal(C, ref)


@pytest.mark.parametrize("S,N,K", [(50, 72, 64), (128, 256, 128), (256, 128, 64)])  # ragged / tile-aligned periods (fast epilogue)
def test_gemm_nt_epilogue(gpu, S, N, K):
    o = ops()
    Bsz = 6
    M = Bsz * S
    A = rnd((M, K), gpu, seed=3)
    W = rnd((N, K), gpu, seed=4, scale=0.2)
    bias = rnd((N,), gpu, dtype=torch.float32, seed=5)
    resid = rnd((M, N), gpu, seed=6)
    gate = rnd((M, N), gpu, seed=7)
    pos = rnd((S, N), gpu, dtype=torch.float32, seed=8)
    cls = rnd((3, N), gpu, dtype=torch.float32, seed=9)
    classes = torch.tensor([0, 2, 1, 1, 0, 2], dtype=torch.int32, device=gpu)
    C = torch.zeros(M, N, dtype=BF, device=gpu)
    o.gemm_nt(A, W, C, bias=bias, resid=resid, act=o.ACT_RELU, gate=gate, alpha=1.7, rowadd=pos, rowadd_period=S,
              grpadd=cls, grp_index=classes)
    torch.cuda.synchronize()
    t = A.float() @ W.float().t() + bias + cls[classes.long()].repeat_interleave(S, 0)
    t = torch.relu(t * 1.7) + pos.repeat(Bsz, 1) + resid.float()
    ref = torch.where(gate.float() > 0, t, torch.zeros_like(t))
    close(C, ref, 1e-2, 2e-2, "gemm_nt epilogue")


def test_gemm_nt_row_remaps_and_f32_out(gpu):
    o = ops()
    Bsz, T, N, K = 5, 9, 40, 32
    # A rows live in a [B, T+1, K] buffer at offset 1 (decoder output with row 0 dropped, model.py:253)
    Abuf = rnd((Bsz * (T + 1), K), gpu, seed=10)
    W = rnd((N, K), gpu, seed=11)
    Cbuf = torch.zeros(Bsz * (T + 1), N, dtype=torch.float32, device=gpu)
    o.gemm_nt(Abuf, W, Cbuf, M=Bsz * T, a_remap=(T, T + 1, 1), c_remap=(T, T + 1, 1))
    torch.cuda.synchronize()
    A3 = Abuf.view(Bsz, T + 1, K)[:, 1:, :].float()
    ref = A3 @ W.float().t()
    got = Cbuf.view(Bsz, T + 1, N)
    close(got[:, 1:, :], ref, 1e-5, 1e-4, "remap")
    assert (got[:, 0, :] == 0).all()


@pytest.mark.parametrize("T", [128, 64, 100])  # groups of whole 64-row tiles take the fast epilogue, T = 100 the general one
def test_gemm_nt_c_remap_keeps_the_physical_row_dropout_counter(gpu, T):
    """16-bit output with a C row remap, a residual and dropout: the mask of output row (b, 1 + t) is the one of physical
    row b*(T+1) + 1 + t, whichever epilogue kernel the launch takes"""
    o = ops()
    Bsz, N, K = 3, 128, 64
    p, seed, site = 0.25, 4242, 2
    A, W = rnd((Bsz * T, K), gpu, seed=14), rnd((N, K), gpu, seed=15, scale=0.2)
    resid = rnd((Bsz * T, N), gpu, seed=16)
    Cbuf = torch.zeros(Bsz * (T + 1), N, dtype=BF, device=gpu)
    o.gemm_nt(A, W, Cbuf, M=Bsz * T, c_remap=(T, T + 1, 1), resid=resid, dropout_p=p, dropout_seed=seed, dropout_site=site)
    keep = torch.zeros(Bsz * (T + 1) * N, dtype=torch.uint8, device=gpu)
    o.dropout_mask(keep.numel(), p, seed, site, keep)
    torch.cuda.synchronize()
    k3 = keep.view(Bsz, T + 1, N)[:, 1:, :].float()
    ref = (A.float() @ W.float().t()).view(Bsz, T, N) * k3 / (1 - p) + resid.float().view(Bsz, T, N)
    got = Cbuf.view(Bsz, T + 1, N)
    close(got[:, 1:, :], ref, 1e-2, 2e-2, "remapped dropout")
    assert (got[:, 0, :] == 0).all()


def test_gemm_nt_dropout_matches_mask_kernel(gpu):
    o = ops()
    M, N, K = 300, 64, 64
    A = rnd((M, K), gpu, seed=12)
    W = rnd((N, K), gpu, seed=13)
    C = torch.zeros(M, N, dtype=torch.float32, device=gpu)
    p, seed, site = 0.2, 1234567, 3
    o.gemm_nt(A, W, C, dropout_p=p, dropout_seed=seed, dropout_site=site)
    keep = torch.zeros(M * N, dtype=torch.uint8, device=gpu)
    o.dropout_mask(M * N, p, seed, site, keep)
    torch.cuda.synchronize()
    ref = (A.float() @ W.float().t()) * keep.view(M, N).float() / (1 - p)
    close(C, ref, 1e-5, 1e-4, "dropout")
    frac = keep.float().mean().item()
    assert abs(frac - 0.8) < 0.01
    C2 = torch.zeros_like(C)
    o.gemm_nt(A, W, C2, dropout_p=p, dropout_seed=seed, dropout_site=site, self_resid=True)
    torch.cuda.synchronize()
    t = A.float() @ W.float().t()
    close(C2, t + t * keep.view(M, N).float() / (1 - p), 1e-5, 1e-4, "self_resid")


# ------------------------------------------------------------------------------------------ GEMM + LayerNorm
@pytest.mark.parametrize("M,N,K,remap", [(16384, 256, 256, False), (16448, 128, 512, False), (200, 256, 1024, False),
                                         (64, 256, 256, True)])
def test_gemm_ln_forward_equals_gemm_then_layernorm(gpu, M, N, K, remap):
    """mst_gemm_nt_ln mode 1 == mst_gemm_nt followed by mst_layernorm_fwd, bit for bit on h, to rounding on y"""
    o = ops()
    S = 5
    A = rnd((M, K), gpu, seed=70, scale=0.5)
    W = rnd((N, K), gpu, seed=71, scale=0.1)
    bias, gam, bet = rnd((N,), gpu, dtype=torch.float32, seed=72), 1 + 0.1 * rnd((N,), gpu, dtype=torch.float32, seed=73), \
        rnd((N,), gpu, dtype=torch.float32, seed=74)
    rows = M * S if remap else M
    resid = rnd((M, N), gpu, seed=75)
    seedp = torch.tensor([77, 0, 0, 0], dtype=torch.int64, device=gpu)
    kw = dict(bias=bias, resid=resid, dropout_p=0.2, dropout_seed_ptr=seedp, dropout_site=4,
              c_remap=(1, S, 0) if remap else (0, 0, 0))
    h0, h1 = (torch.zeros(rows, N, dtype=BF, device=gpu) for _ in range(2))
    y0, y1 = (torch.zeros(rows, N, dtype=BF, device=gpu) for _ in range(2))
    m0, r0, m1, r1 = (torch.zeros(rows, device=gpu) for _ in range(4))
    o.gemm_nt(A, W, h0, **kw)
    if remap:
        v = lambda t: t.view(M, S, -1)[:, 0, :]
        o.layernorm_fwd(v(h0), gam, bet, v(y0), m0, r0, D=N, M=M, row_id_stride=S)
    else:
        o.layernorm_fwd(h0, gam, bet, y0, m0, r0)
    o.gemm_nt_ln_fwd(A, W, h1, gam, bet, y1, m1, r1, **kw)
    torch.cuda.synchronize()
    assert torch.equal(h0, h1)
    close(m1, m0, 1e-5, 1e-5, "mean")
    close(r1, r0, 1e-5, 1e-5, "rstd")
    close(y1, y0, 1e-2, 1e-2, "LayerNorm output")
    assert (y1.float() - y0.float()).abs().max().item() <= 2 ** -6  # at most one bf16 ulp at |y| < 4


@pytest.mark.parametrize("M,N,K,mode,remap", [(16384, 256, 1024, 1, False), (16448, 128, 128, 2, False), (300, 256, 768, 0, False),
                                              (64, 256, 1024, 1, True)])
def test_gemm_ln_backward_equals_gemm_then_layernorm_bwd(gpu, M, N, K, mode, remap):
    """mst_gemm_nt_ln mode 2 == mst_gemm_nt followed by mst_layernorm_bwd"""
    o = ops()
    S = 3
    A = rnd((M, K), gpu, seed=80, scale=0.3)
    W = rnd((N, K), gpu, seed=81, scale=0.1)
    resid = rnd((M, N), gpu, seed=82)
    gam = 1 + 0.1 * rnd((N,), gpu, dtype=torch.float32, seed=83)
    rows = M * S if remap else M
    xfull = rnd((rows, N), gpu, seed=84)
    x = xfull.view(M, S, -1)[:, 0, :] if remap else xfull
    mean = x.float().mean(1)
    rstd = 1.0 / torch.sqrt(x.float().var(1, unbiased=False) + 1e-5)
    mean_f, rstd_f = torch.zeros(rows, device=gpu), torch.zeros(rows, device=gpu)
    stride = S if remap else 1
    mean_f[::stride], rstd_f[::stride] = mean, rstd
    seedp = torch.tensor([91, 0, 0, 0], dtype=torch.int64, device=gpu)
    drop = dict(dropout_p=0.2, dropout_seed_ptr=seedp, dropout_site=2) if mode else {}
    res = []
    for fused in (False, True, "partials"):
        dxf = torch.zeros(rows, N, dtype=BF, device=gpu)
        dx = dxf.view(M, S, -1)[:, 0, :] if remap else dxf
        dxm = torch.zeros(M, N, dtype=BF, device=gpu)
        dg, db = torch.zeros(N, device=gpu), torch.zeros(N, device=gpu)
        if fused:
            parts = o.gemm_nt_ln_parts(M)
            part = torch.full((parts, 2 * N), float("nan"), device=gpu) if fused == "partials" else None
            o.gemm_nt_ln_bwd(A, W, dxf, xfull, gam, mean_f, rstd_f, dg, db, dx_masked=dxm if mode == 1 else None,
                             mask_mode=mode, resid=resid, c_remap=(1, S, 0) if remap else (0, 0, 0), partials=part, **drop)
            if part is not None:  # per-workgroup column sums, added by the deferred launch
                assert (dg == 0).all() and (db == 0).all()
                o.partial_sums([o.partial_sum_job(part, parts, dg, length=N), o.partial_sum_job(part, parts, db, col_off=N, length=N)])
        else:
            dy = torch.zeros(M, N, dtype=BF, device=gpu)
            o.gemm_nt(A, W, dy, resid=resid)
            o.layernorm_bwd(x, gam, mean_f, rstd_f, dy, dx, dg, db, D=N, M=M, row_id_stride=stride,
                            dx_masked=dxm if mode == 1 else None, mask_mode=mode, **drop)
        torch.cuda.synchronize()
        res.append((dxf.clone(), dxm.clone(), dg.clone(), db.clone()))
    scale = res[0][0].float().abs().max().item()
    close(res[1][0], res[0][0], 1e-2, 1e-2 * scale, "dx")
    close(res[1][1], res[0][1], 1e-2, 1e-2 * scale, "dx masked")
    close(res[1][2], res[0][2], 1e-3, 1e-3 * res[0][2].abs().max().item(), "dgamma")
    close(res[1][3], res[0][3], 1e-3, 1e-3 * res[0][3].abs().max().item(), "dbeta")
    assert torch.equal(res[2][0], res[1][0]) and torch.equal(res[2][1], res[1][1])
    close(res[2][2], res[0][2], 1e-3, 1e-3 * res[0][2].abs().max().item(), "dgamma from partials")
    close(res[2][3], res[0][3], 1e-3, 1e-3 * res[0][3].abs().max().item(), "dbeta from partials")
    if remap:  # rows the launch does not own stay untouched
        assert (res[1][0].view(M, S, -1)[:, 1:] == 0).all()


@pytest.mark.parametrize("M,D,F,p,self_resid", [(16384, 256, 1024, 0.2, False), (16448, 128, 512, 0.2, True), (200, 256, 512, 0.0, False),
                                                  (77, 128, 128, 0.1, False)])
def test_ffn_ln_fwd_equals_the_three_launches(gpu, M, D, F, p, self_resid):
    """mst_ffn_ln_fwd == gemm_nt(ff1) + gemm_nt(ff2) + layernorm_fwd: the hidden activation bit for bit (same MFMA order, same
    epilogue); the pre-norm tensor sums the hidden chunks in a per-workgroup rotated order (fp32), so it agrees to one
    rounding of the activation type on a small fraction of elements, and the LayerNorm output likewise"""
    o = ops()
    x = rnd((M, D), gpu, seed=400)
    W1, W2 = rnd((F, D), gpu, seed=401, scale=0.06), rnd((D, F), gpu, seed=402, scale=0.03)
    b1, b2 = rnd((F,), gpu, dtype=torch.float32, seed=403, scale=0.1), rnd((D,), gpu, dtype=torch.float32, seed=404, scale=0.1)
    gam, bet = 1 + 0.1 * rnd((D,), gpu, dtype=torch.float32, seed=405), rnd((D,), gpu, dtype=torch.float32, seed=406, scale=0.1)
    seedp = torch.tensor([55, 0, 0, 0], dtype=torch.int64, device=gpu)
    d1 = dict(dropout_p=p, dropout_seed_ptr=seedp, dropout_site=4) if p > 0 else {}
    d2 = dict(dropout_p=p, dropout_seed_ptr=seedp, dropout_site=5) if p > 0 else {}
    ff1 = dict(K=D, bias=b1, act=o.ACT_RELU, **d1)
    ff2 = dict(K=F, bias=b2, **d2)
    ff2.update(dict(self_resid=True) if self_resid else dict(resid=x))

    def bufs():
        return (torch.zeros(M, F, dtype=BF, device=gpu), torch.zeros(M, D, dtype=BF, device=gpu), torch.zeros(M, D, dtype=BF, device=gpu),
                torch.zeros(M, device=gpu), torch.zeros(M, device=gpu))

    a0, h0, y0, m0, r0 = bufs()
    o.gemm_nt(x, W1, a0, **ff1)
    o.gemm_nt(a0, W2, h0, **ff2)
    o.layernorm_fwd(h0, gam, bet, y0, m0, r0, D=D)
    a1, h1, y1, m1, r1 = bufs()
    o.ffn_ln_fwd(x, W1, a1, W2, h1, gam, bet, y1, m1, r1, ff1=ff1, ff2=ff2)
    torch.cuda.synchronize()
    assert torch.equal(a1, a0), "hidden activation"
    dh = (h1.float() - h0.float()).abs()
    assert (dh <= 2 ** -7 * h0.float().abs().clamp(min=1.0)).all() and (h1 != h0).float().mean().item() < 2e-2, "pre-norm tensor"
    close(m1, m0, 1e-3, 5e-4, "mean")  # (statistics of rows in which an element of the pre-norm tensor rounded the other way)
    close(r1, r0, 2e-3, 1e-4, "rstd")
    assert (y1.float() - y0.float()).abs().max().item() <= 2 ** -5  # a bf16 ulp or two at |y| < 4
    assert ((y1 != y0).float().mean().item()) < 5e-2


@pytest.mark.parametrize("B,T,D,F,p", [(64, 256, 128, 512, 0.2), (3, 64, 256, 1024, 0.0), (5, 128, 128, 512, 0.1)])
def test_ffn_ln_row_groups_skip_position_0_of_every_sample(gpu, B, T, D, F, p):
    """row_groups = (T, T + 1, 1): the feed-forward launches (forward with the projection head, and backward) of the LAST decoder
    layer work on rows 1..T of every sample only (engine skip_row0: position 0's output is dropped before the loss, model.py:253).
    Against the same launches over all B (T + 1) rows: the rows they share agree — hidden activation and gated gradient bit for
    bit, the rest to a rounding of the activation type (the fp32 sum over the hidden chunks runs in an order that depends on the
    workgroup's place in the launch) — dropout masks included (their counters are the PHYSICAL row), and position-0 rows are
    neither read nor written (a sentinel survives)."""
    o = ops()
    S, M = T + 1, B * (T + 1)
    att, xres = rnd((M, D), gpu, seed=500), rnd((M, D), gpu, seed=501)
    Wp, W1, W2 = rnd((D, D), gpu, seed=502, scale=0.06), rnd((F, D), gpu, seed=503, scale=0.06), rnd((D, F), gpu, seed=504, scale=0.03)
    bp, b1, b2 = (rnd((n,), gpu, dtype=torch.float32, seed=505 + i, scale=0.1) for i, n in enumerate((D, F, D)))
    g1, be1 = 1 + 0.1 * rnd((D,), gpu, dtype=torch.float32, seed=510), rnd((D,), gpu, dtype=torch.float32, seed=511, scale=0.1)
    g3, be3 = 1 + 0.1 * rnd((D,), gpu, dtype=torch.float32, seed=512), rnd((D,), gpu, dtype=torch.float32, seed=513, scale=0.1)
    seedp = torch.tensor([77, 0, 0, 0], dtype=torch.int64, device=gpu)
    drop = lambda site: dict(dropout_p=p, dropout_seed_ptr=seedp, dropout_site=site) if p > 0 else {}
    SENT = 7.0

    def fwd(rows):
        bufs = dict(h1=(M, D), x1=(M, D), a=(M, F), h2=(M, D), y=(M, D))
        t = {k: torch.full(v, SENT, dtype=BF, device=gpu) for k, v in bufs.items()}
        st = {k: torch.full((M,), SENT, device=gpu) for k in ("m1", "r1", "m2", "r2")}
        head = dict(att=att, W=Wp, h1=t["h1"], gamma=g1, beta=be1, mean=st["m1"], rstd=st["r1"], N=D, K=D, bias=bp, resid=xres, **drop(3))
        o.ffn_ln_fwd(t["x1"], W1, t["a"], W2, t["h2"], g3, be3, t["y"], st["m2"], st["r2"], ff1=dict(K=D, bias=b1, act=o.ACT_RELU, **drop(4)),
                     ff2=dict(K=F, bias=b2, self_resid=True, **drop(5)), proj=head, row_groups=rows)
        torch.cuda.synchronize()
        return t, st

    full, fst = fwd(None)
    part, pst = fwd((T, S, 1))
    ulp = 2.0 ** -7
    for k in ("h1", "x1", "a", "h2", "y"):
        a3, b3 = part[k].view(B, S, -1).float(), full[k].view(B, S, -1).float()
        assert (a3[:, 0] == SENT).all(), f"{k}: a position-0 row was written"
        if k in ("h1", "x1", "a"):  # one GEMM each, K in the same order
            assert torch.equal(a3[:, 1:], b3[:, 1:]), k
        else:
            d = (a3[:, 1:] - b3[:, 1:]).abs()
            assert (d <= 2 * ulp * b3[:, 1:].abs().clamp(min=1.0)).all() and (d > 0).float().mean().item() < 0.05, k
    for k in ("m1", "r1", "m2", "r2"):
        a2, b2_ = pst[k].view(B, S), fst[k].view(B, S)
        assert (a2[:, 0] == SENT).all()
        close(a2[:, 1:], b2_[:, 1:], 2e-3, 5e-4, k)

    # backward: FFN2 dgrad + ReLU gate + FFN1 dgrad + LayerNorm-1 backward, dff zero at position 0 as in the step
    dff = rnd((M, D), gpu, seed=520)
    dff.view(B, S, D)[:, 0] = 0
    W2t, W1t = W2.t().contiguous(), W1.t().contiguous()
    nparts = o.gemm_nt_ln_parts(M)

    def bwd(rows):
        dpre, dh1, dh1m = (torch.full(sh, SENT, dtype=BF, device=gpu) for sh in ((M, F), (M, D), (M, D)))
        parts = torch.zeros(nparts, 2 * D, device=gpu)
        dg, db = torch.zeros(D, device=gpu), torch.zeros(D, device=gpu)
        kw = dict(dx_masked=dh1m, mask_mode=1, **drop(3)) if p > 0 else {}
        o.ffn_ln_bwd(dff, W2t, dpre, full["a"], W1t, dh1, full["h1"], g1, fst["m1"], fst["r1"], dg, db, alpha=1.0 / (1.0 - p) if p > 0 else 1.0,
                     partials=parts, row_groups=rows, **kw)
        torch.cuda.synchronize()
        return dpre, dh1, dh1m, parts.sum(0)

    fb, pb = bwd(None), bwd((T, S, 1))
    for nm, a, b in (("dpre", pb[0], fb[0]), ("dh1", pb[1], fb[1])) + ((("dh1m", pb[2], fb[2]),) if p > 0 else ()):
        a3, b3 = a.view(B, S, -1).float(), b.view(B, S, -1).float()
        assert (a3[:, 0] == SENT).all(), f"{nm}: a position-0 row was written"
        if nm == "dpre":
            assert torch.equal(a3[:, 1:], b3[:, 1:]), nm
        else:
            d = (a3[:, 1:] - b3[:, 1:]).abs()
            assert (d <= 2 * ulp * b3[:, 1:].abs().clamp(min=1.0)).all(), nm
    # the LayerNorm parameter gradients: position-0 rows contribute nothing on either side (dff = 0 there gives a zero dx ... but
    # d gamma sums dy * xhat of the block's OUTPUT gradient, which is non-zero at position 0 in the full launch's garbage rows): compare
    # against the full launch's sums minus its position-0 rows, i.e. recompute from the shared rows
    dy = (fb[0].float() @ W1t.float().t())  # d(x1) before the LayerNorm backward, all rows
    xh = (full["h1"].float() - fst["m1"][:, None]) * fst["r1"][:, None]
    keep = torch.ones(B, S, 1, device=gpu)
    keep[:, 0] = 0
    ref = torch.cat([(dy * xh).view(B, S, D).mul(keep).sum((0, 1)), dy.view(B, S, D).mul(keep).sum((0, 1))])
    close(pb[3], ref, 2e-2, 2e-2 * ref.abs().max().item(), "dgamma | dbeta partial sums over rows 1..T")


@pytest.mark.parametrize("M,D,F,p", [(16384, 256, 1024, 0.2), (16448, 128, 512, 0.2), (200, 256, 512, 0.0), (77, 128, 128, 0.1)])
def test_proj_ffn_ln_fwd_equals_projection_then_block(gpu, M, D, F, p):
    """mst_proj_ffn_ln_fwd == mst_gemm_nt_ln(proj, ln1) + mst_ffn_ln_fwd, bit for bit in every output (h1, x1 and its
    statistics, hidden activation, pre-norm tensor, block output): the same epilogue code, the same MFMA order"""
    o = ops()
    att, xin = rnd((M, D), gpu, seed=420), rnd((M, D), gpu, seed=421)
    Wp = rnd((D, D), gpu, seed=422, scale=0.06)
    W1, W2 = rnd((F, D), gpu, seed=401, scale=0.06), rnd((D, F), gpu, seed=402, scale=0.03)
    bp = rnd((D,), gpu, dtype=torch.float32, seed=423, scale=0.1)
    b1, b2 = rnd((F,), gpu, dtype=torch.float32, seed=403, scale=0.1), rnd((D,), gpu, dtype=torch.float32, seed=404, scale=0.1)
    g1, be1 = 1 + 0.1 * rnd((D,), gpu, dtype=torch.float32, seed=424), rnd((D,), gpu, dtype=torch.float32, seed=425, scale=0.1)
    g2, be2 = 1 + 0.1 * rnd((D,), gpu, dtype=torch.float32, seed=405), rnd((D,), gpu, dtype=torch.float32, seed=406, scale=0.1)
    seedp = torch.tensor([55, 0, 0, 0], dtype=torch.int64, device=gpu)
    dk = lambda site: dict(dropout_p=p, dropout_seed_ptr=seedp, dropout_site=site) if p > 0 else {}
    proj = dict(N=D, K=D, bias=bp, resid=xin, **dk(3))
    ff1 = dict(K=D, bias=b1, act=o.ACT_RELU, **dk(4))
    res = []
    for fused in (False, True):
        h1, x1 = torch.zeros(M, D, dtype=BF, device=gpu), torch.zeros(M, D, dtype=BF, device=gpu)
        a, h2, y = torch.zeros(M, F, dtype=BF, device=gpu), torch.zeros(M, D, dtype=BF, device=gpu), torch.zeros(M, D, dtype=BF, device=gpu)
        m1, r1, m2, r2 = (torch.zeros(M, device=gpu) for _ in range(4))
        ff2 = dict(K=F, bias=b2, resid=x1, **dk(5))
        if fused:
            o.ffn_ln_fwd(x1, W1, a, W2, h2, g2, be2, y, m2, r2, ff1=ff1, ff2=ff2,
                         proj=dict(att=att, W=Wp, h1=h1, gamma=g1, beta=be1, mean=m1, rstd=r1, **proj))
        else:
            o.gemm_nt_ln_fwd(att, Wp, h1, g1, be1, x1, m1, r1, **proj)
            o.ffn_ln_fwd(x1, W1, a, W2, h2, g2, be2, y, m2, r2, ff1=ff1, ff2=ff2)
        torch.cuda.synchronize()
        res.append((h1, x1, m1, r1, a, h2, y, m2, r2))
    for name, u, v in zip(("h1", "x1", "mean1", "rstd1", "a", "h2", "y", "mean2", "rstd2"), res[0], res[1]):
        assert torch.equal(u, v), name
    # ... and the unfused projection GEMM gives the same h1
    h1u = torch.zeros(M, D, dtype=BF, device=gpu)
    o.gemm_nt(att, Wp, h1u, **proj)
    torch.cuda.synchronize()
    assert torch.equal(h1u, res[1][0])


@pytest.mark.parametrize("M,D,F,mode,with_resid", [(16384, 256, 1024, 1, True), (16448, 128, 512, 2, False), (200, 256, 512, 0, True),
                                                     (77, 128, 128, 1, True)])
def test_ffn_ln_bwd_equals_the_separate_launches(gpu, M, D, F, mode, with_resid):
    """mst_ffn_ln_bwd == gemm_nt(gate) + gemm_nt(resid) + layernorm_bwd: the gated d(pre-activation) bit for bit, dx and the
    parameter gradients to the tolerance of the LayerNorm-fused GEMM"""
    o = ops()
    dff = rnd((M, D), gpu, seed=500, scale=0.5)
    gate = rnd((M, F), gpu, seed=501)
    W2t, W1t = rnd((F, D), gpu, seed=502, scale=0.05), rnd((D, F), gpu, seed=503, scale=0.05)
    resid = rnd((M, D), gpu, seed=504) if with_resid else None
    x = rnd((M, D), gpu, seed=505)
    gam = 1 + 0.1 * rnd((D,), gpu, dtype=torch.float32, seed=506)
    mean = x.float().mean(1)
    rstd = 1.0 / torch.sqrt(x.float().var(1, unbiased=False) + 1e-5)
    seedp = torch.tensor([91, 0, 0, 0], dtype=torch.int64, device=gpu)
    drop = dict(dropout_p=0.2, dropout_seed_ptr=seedp, dropout_site=2) if mode else {}
    rk = dict(resid=resid) if with_resid else {}
    # separate launches
    dpre0, dy0 = torch.zeros(M, F, dtype=BF, device=gpu), torch.zeros(M, D, dtype=BF, device=gpu)
    dx0, dxm0 = torch.zeros(M, D, dtype=BF, device=gpu), torch.zeros(M, D, dtype=BF, device=gpu)
    dg0, db0 = torch.zeros(D, device=gpu), torch.zeros(D, device=gpu)
    o.gemm_nt(dff, W2t, dpre0, gate=gate, alpha=1.25)
    o.gemm_nt(dpre0, W1t, dy0, **rk)
    o.layernorm_bwd(x, gam, mean, rstd, dy0, dx0, dg0, db0, D=D, dx_masked=dxm0 if mode == 1 else None, mask_mode=mode, **drop)
    # one launch, parameter gradients through the partial rows
    dpre1 = torch.zeros_like(dpre0)
    dx1, dxm1 = torch.zeros_like(dx0), torch.zeros_like(dxm0)
    dg1, db1 = torch.zeros(D, device=gpu), torch.zeros(D, device=gpu)
    parts = o.gemm_nt_ln_parts(M)
    part = torch.full((parts, 2 * D), float("nan"), device=gpu)
    o.ffn_ln_bwd(dff, W2t, dpre1, gate, W1t, dx1, x, gam, mean, rstd, dg1, db1, alpha=1.25, dx_masked=dxm1 if mode == 1 else None,
                 mask_mode=mode, partials=part, **rk, **drop)
    o.partial_sums([o.partial_sum_job(part, parts, dg1, length=D), o.partial_sum_job(part, parts, db1, col_off=D, length=D)])
    torch.cuda.synchronize()
    assert torch.equal(dpre1, dpre0), "gated d(pre-activation)"
    scale = dx0.float().abs().max().item()
    close(dx1, dx0, 1e-2, 1e-2 * scale, "dx")
    close(dxm1, dxm0, 1e-2, 1e-2 * scale, "dx masked")
    close(dg1, dg0, 1e-3, 1e-3 * dg0.abs().max().item(), "dgamma")
    close(db1, db0, 1e-3, 1e-3 * db0.abs().max().item(), "dbeta")
    # ---- with the leading LayerNorm backward in the prologue: dff is itself the (masked) LayerNorm backward of a gradient
    dyin, xin = rnd((M, D), gpu, seed=507), rnd((M, D), gpu, seed=508, scale=1.5)
    gin = 1 + 0.1 * rnd((D,), gpu, dtype=torch.float32, seed=509)
    mean_in, rstd_in = xin.float().mean(1), 1.0 / torch.sqrt(xin.float().var(1, unbiased=False) + 1e-5)
    lead_drop = dict(dropout_p=0.2, dropout_seed_ptr=seedp, dropout_site=7) if mode == 1 else {}
    res = []
    for lead_fused in (False, True):
        dh, dhm = torch.zeros(M, D, dtype=BF, device=gpu), torch.zeros(M, D, dtype=BF, device=gpu)
        dgi, dbi = torch.zeros(D, device=gpu), torch.zeros(D, device=gpu)
        dpre = torch.zeros(M, F, dtype=BF, device=gpu)
        dx = torch.zeros(M, D, dtype=BF, device=gpu)
        dg, db = torch.zeros(D, device=gpu), torch.zeros(D, device=gpu)
        masked = dhm if mode == 1 else None
        a_op = dhm if mode == 1 else dh
        rk2 = dict(resid=dh) if with_resid else {}
        if lead_fused:
            pin = torch.full((parts, 2 * D), float("nan"), device=gpu)
            o.ffn_ln_bwd(a_op, W2t, dpre, gate, W1t, dx, x, gam, mean, rstd, dg, db, alpha=1.25, **rk2,
                         lead=dict(dy=dyin, x=xin, gamma=gin, mean=mean_in, rstd=rstd_in, dx=dh, dx_masked=masked, partials=pin, **lead_drop))
            o.partial_sums([o.partial_sum_job(pin, parts, dgi, length=D), o.partial_sum_job(pin, parts, dbi, col_off=D, length=D)])
        else:
            o.layernorm_bwd(xin, gin, mean_in, rstd_in, dyin, dh, dgi, dbi, D=D, dx_masked=masked, mask_mode=1 if mode == 1 else 0, **lead_drop)
            o.ffn_ln_bwd(a_op, W2t, dpre, gate, W1t, dx, x, gam, mean, rstd, dg, db, alpha=1.25, **rk2)
        torch.cuda.synchronize()
        res.append((dh.clone(), dhm.clone(), dgi.clone(), dbi.clone(), dpre.clone(), dx.clone()))
    sc = res[0][0].float().abs().max().item()
    close(res[1][0], res[0][0], 1e-2, 1e-2 * sc, "leading dx")
    close(res[1][1], res[0][1], 1e-2, 1e-2 * sc, "leading dx masked")
    close(res[1][2], res[0][2], 1e-3, 1e-3 * res[0][2].abs().max().item(), "leading dgamma")
    close(res[1][3], res[0][3], 1e-3, 1e-3 * res[0][3].abs().max().item(), "leading dbeta")
    close(res[1][4], res[0][4], 2e-2, 2e-2 * res[0][4].float().abs().max().item(), "d(pre) after the fused leading LayerNorm")
    close(res[1][5], res[0][5], 2e-2, 2e-2 * res[0][5].float().abs().max().item(), "dx after the fused leading LayerNorm")


# ------------------------------------------------------------------------------------------ wgrad
@pytest.mark.parametrize("M,N,K", [(512, 64, 64), (16384, 256, 256), (4097, 128, 1024), (100, 16, 32), (16448, 384, 128)])
def test_gemm_wgrad(gpu, M, N, K):
    o = ops()
    dY = rnd((M, N), gpu, seed=20, scale=0.1)
    X = rnd((M, K), gpu, seed=21)
    dW = torch.zeros(N, K, dtype=torch.float32, device=gpu)
    db = torch.zeros(N, dtype=torch.float32, device=gpu)
    o.gemm_wgrad(dY, X, dW, db)
    torch.cuda.synchronize()
    ref = dY.float().t() @ X.float()
    refb = dY.float().sum(0)
    close(dW, ref, 2e-3, 2e-3 * math.sqrt(M) * 0.1, "wgrad dW")
    close(db, refb, 2e-3, 2e-3 * math.sqrt(M) * 0.1, "wgrad db")


def test_gemm_wgrad_integer_exact_and_batch(gpu):
    o = ops()
    M = 200
    m = torch.arange(M).view(-1, 1)
    probs, refs, outs = [], [], []
    for (N, K, s) in [(32, 48, 1.0), (72, 16, 2.0), (8, 8, 1.0)]:
        A = (((m * 3 + torch.arange(N).view(1, -1) * 5) % 7) - 3).to(BF).to(gpu)
        B = (((m * 2 + torch.arange(K).view(1, -1) * 7) % 5) - 2).to(BF).to(gpu)
        dW = torch.zeros(N, K, dtype=torch.float32, device=gpu)
        db = torch.zeros(N, dtype=torch.float32, device=gpu)
        probs.append(o.wgrad_problem(A, B, dW, db, scale=s))
        refs.append((s * (A.float().t() @ B.float()), s * A.float().sum(0)))
        outs.append((dW, db, A, B))
    o.gemm_wgrad_batch(probs)
    torch.cuda.synchronize()
    for (dW, db, _, _), (rw, rb) in zip(outs, refs):
        assert torch.equal(dW, rw)
        assert torch.equal(db, rb)


def roundup8(x):
    return (x + 7) // 8 * 8


def test_gemm_wgrad_two_pass_reduction_is_deterministic_and_equal(gpu):
    """a batch big enough for 256x256 tiles: with a scratch buffer the M-slabs are summed in slab order by a second
    launch instead of fp32 atomics — same result to fp32 rounding, and bit-identical run to run"""
    o = ops()
    M = 4096
    shapes = [(256, 1024), (1024, 256), (768, 256), (256, 256), (300, 200), (1024, 256), (256, 1024), (512, 512)]  # 2.1 M outputs
    probs = []
    for i, (N, K) in enumerate(shapes):
        A, B = rnd((M, roundup8(N)), gpu, seed=140 + i), rnd((M, roundup8(K)), gpu, seed=150 + i)
        A[:, N:] = 0
        B[:, K:] = 0
        probs.append((A, B, N, K))
    scratch = torch.empty(16 * 1024 * 1024, dtype=torch.float32, device=gpu)

    # column-sum jobs that ride along (on the reduction pass with the scratch buffer, in their own launch without)
    g = torch.Generator().manual_seed(3)
    part = torch.randint(-8, 9, (300, 512), generator=g).float().to(gpu)
    sum_specs = [(257, 0, 512), (300, 256, 128), (1, 4, 60)] * 9  # 27 jobs: more than one launch's worth
    colsums = []

    def run(ws):
        outs = [(torch.zeros(N, K, device=gpu), torch.zeros(N, device=gpu)) for _, _, N, K in probs]
        dsts = [torch.ones(length, device=gpu) for _, _, length in sum_specs]
        o.gemm_wgrad_batch([o.wgrad_problem(A, B, dW, db, N=N, K=K, scale=0.5) for (A, B, N, K), (dW, db) in zip(probs, outs)],
                           scratch=scratch if ws else None,
                           sums=[o.partial_sum_job(part, n, d, col_off=off, length=length) for (n, off, length), d in zip(sum_specs, dsts)])
        torch.cuda.synchronize()
        colsums.append(dsts)
        return outs

    atomic, two_a, two_b = run(False), run(True), run(True)
    for (A, B, N, K), (dWa, dba), (dW1, db1), (dW2, _) in zip(probs, atomic, two_a, two_b):
        ref = 0.5 * (A[:, :N].float().t() @ B[:, :K].float())
        close(dW1, ref, 1e-4, 1e-3 * ref.abs().max().item(), "two-pass dW")
        close(dW1, dWa, 1e-5, 1e-4 * ref.abs().max().item(), "two-pass vs atomic")
        close(db1, 0.5 * A[:, :N].float().sum(0), 1e-4, 1e-2, "db")
        assert torch.equal(dW1, dW2), "the two-pass reduction must be run-to-run deterministic"
    for dsts in colsums:
        for (n, off, length), d in zip(sum_specs, dsts):
            assert torch.equal(d, 1.0 + part[:n, off:off + length].sum(0))


def test_gemm_wgrad_remap(gpu):
    o = ops()
    Bsz, T, N, K = 4, 33, 24, 40
    dYbuf = rnd((Bsz * (T + 1), N), gpu, seed=22)
    X = rnd((Bsz * T, K), gpu, seed=23)
    dW = torch.zeros(N, K, dtype=torch.float32, device=gpu)
    o.gemm_wgrad(dYbuf, X, dW, M=Bsz * T, a_remap=(T, T + 1, 1))
    torch.cuda.synchronize()
    dY = dYbuf.view(Bsz, T + 1, N)[:, 1:, :].reshape(Bsz * T, N).float()
    close(dW, dY.t() @ X.float(), 1e-4, 1e-2, "wgrad remap")


# ------------------------------------------------------------------------------------------ attention
def attn_reference(qkv, keymask, B, S, H, dh, k_off, q_off, v_off):
    """fp32 restatement of transformer.py:85-104 (key-row softmax, P^T V, -1e9 on padded key rows)"""
    x = qkv.float().view(B, S, -1)
    D = H * dh

    def heads(off):
        return x[:, :, off:off + D].reshape(B, S, H, dh).permute(0, 2, 1, 3)

    K, Q, V = heads(k_off), heads(q_off), heads(v_off)
    logits = torch.matmul(K, Q.transpose(-1, -2)) / math.sqrt(dh)  # [B,H,k,q]
    madd = torch.where(keymask.view(B, 1, S, 1) > 0, 0.0, -1e9).to(torch.float32)
    logits = logits + madd
    P = torch.softmax(logits, dim=-1)
    O = torch.matmul(P.transpose(-1, -2), V)  # [B,H,q,dh]
    return O.permute(0, 2, 1, 3).reshape(B * S, D), torch.logsumexp(logits, dim=-1)


@pytest.mark.parametrize("B,S,H,dh,q_limit,dtype", [(8, 256, 8, 32, 0, BF), (3, 200, 4, 32, 0, BF), (8, 256, 8, 32, 1, BF), (5, 96, 2, 32, 0, torch.float16),
                                                   (8, 224, 4, 32, 0, torch.float16), (2, 512, 2, 32, 0, BF),
                                                   (4, 64, 4, 16, 0, BF), (2, 640, 2, 32, 0, BF)])
def test_attention_with_fused_projection_equals_gemm_then_attention(gpu, B, S, H, dh, q_limit, dtype):
    """mst_attn_qkv_fwd: the K | Q | V projection inside the attention launch (head size 32, resident sequences) against the GEMM
    launch followed by the attention launch — same qkv to a rounding of the activation type (another fp32 summation order), same
    statistics and output to the tolerance that leaves; ragged lengths take the padded-key path. Head size 16 and sequences beyond
    the resident kernel run the two launches inside the call and must agree exactly."""
    o = ops()
    D = H * dh
    g = torch.Generator().manual_seed(5)
    x = (torch.randn(B * S, D, generator=g)).to(dtype).to(gpu)
    W = (torch.randn(3 * D, D, generator=g) * (1.5 / D ** 0.5)).to(dtype).to(gpu)
    bias = (torch.randn(3 * D, generator=g) * 0.2).to(gpu)
    lens = torch.randint(S // 2, S + 1, (B,), generator=g)
    lens[0] = S
    keymask = (torch.arange(S).view(1, S) < lens.view(B, 1)).to(torch.uint8).to(gpu)

    def bufs():
        return (torch.zeros(B * S, 3 * D, dtype=dtype, device=gpu), torch.zeros(2, B, H, S, device=gpu), torch.zeros(B * S, D, dtype=dtype, device=gpu))

    qkv_a, lse_a, out_a = bufs()
    o.gemm_nt(x, W, qkv_a, K=D, bias=bias)
    o.attn_fwd(qkv_a, keymask, lse_a, out_a, B, S, H, dh, 0, D, 2 * D, q_limit=q_limit)
    qkv_b, lse_b, out_b = bufs()
    o.attn_qkv_fwd(x, W, bias, qkv_b, keymask, lse_b, out_b, B, S, H, dh, 0, D, 2 * D, q_limit=q_limit)
    torch.cuda.synchronize()
    fused_shape = dh == 32 and S <= 512
    if not fused_shape:
        assert torch.equal(qkv_a, qkv_b) and torch.equal(out_a, out_b) and torch.equal(lse_a, lse_b)
        return
    ulp = 2.0 ** -7 if dtype == BF else 2.0 ** -10
    d = (qkv_b.float() - qkv_a.float()).abs()
    assert (d <= ulp * qkv_a.float().abs().clamp(min=1.0)).all(), d.max().item()
    assert (qkv_a != qkv_b).float().mean().item() < 0.02
    ref = x.float() @ W.float().t() + bias
    close(qkv_b, ref, 2 * ulp, 2 * ulp, "fused projection vs fp32")
    rows = slice(None) if q_limit == 0 else slice(0, None, S)  # (q_limit = 1: only position 0 of every sample is produced)
    close(out_b[rows], out_a[rows], 3e-2, 3e-2, "attention output")
    close(lse_b.sum(0), lse_a.sum(0), 1e-3, 2e-2, "log-sum-exp of the key rows")


@pytest.mark.parametrize("B,S,H,dh", [(2, 64, 2, 32), (3, 5, 2, 16), (2, 257, 8, 16), (2, 256, 8, 32), (1, 100, 1, 64),
                                     (1, 1024, 2, 32),
                                     # configs[4]'s decoder: Q | K | V do not fit LDS together - the resident forward stages K and V
                                     # over Q between its phases (two tiles), the lone row 1025 apart
                                     (1, 1025, 2, 16), (2, 700, 2, 32),
                                     # the same forms with padded keys in the second sample (the reference's operation order in those waves),
                                     # a ragged last chunk, and a chunked dQ whose sequence is not a whole number of chunks' tiles
                                     (2, 1000, 2, 32), (2, 1025, 1, 16), (2, 640, 2, 32)])
@pytest.mark.parametrize("path", ["auto", "stream"])
def test_attention_fwd_bwd(gpu, monkeypatch, B, S, H, dh, path):
    """auto: the resident single-launch kernels when the sequence fits in LDS (all cases but S=1024), else the
    streaming kernels; stream: the streaming kernels for every case"""
    if path == "stream":
        monkeypatch.setenv("MST_ATTN_PATH", "stream")
    o = ops()
    D = H * dh
    ldq = 3 * D
    qkv = rnd((B * S, ldq), gpu, seed=30, scale=1.0)
    lens = torch.tensor([S - (i * 7) % max(1, S // 2) for i in range(B)], dtype=torch.int32, device=gpu)
    keymask = torch.zeros(B, S, dtype=torch.uint8, device=gpu)
    o.mask_from_lengths(lens, 0, keymask)
    lse = torch.zeros(2, B, H, S, dtype=torch.float32, device=gpu)
    out = torch.zeros(B * S, D, dtype=BF, device=gpu)
    k_off, q_off, v_off = 0, D, 2 * D
    o.attn_fwd(qkv, keymask, lse, out, B, S, H, dh, k_off, q_off, v_off)
    torch.cuda.synchronize()
    assert keymask.sum().item() == int(lens.sum().item())

    qkv_ref = qkv.float().clone().requires_grad_(True)
    ref, lse_ref = attn_reference(qkv_ref, keymask, B, S, H, dh, k_off, q_off, v_off)
    close(lse[0] + lse[1], lse_ref.detach(), 1e-4, 2e-3, "lse")
    close(out, ref.detach(), 1.6e-2, 1.6e-2, "attention out")

    dout = rnd((B * S, D), gpu, seed=31, scale=1.0)
    dqkv = torch.zeros(B * S, ldq, dtype=BF, device=gpu)
    delta = torch.zeros(B, H, S, dtype=torch.float32, device=gpu)
    o.attn_bwd(qkv, keymask, lse, dout, dqkv, delta, B, S, H, dh, k_off, q_off, v_off)
    torch.cuda.synchronize()
    ref.backward(dout.float())
    g = qkv_ref.grad
    scale = g.abs().max().item()
    close(dqkv[:, v_off:v_off + D], g[:, v_off:v_off + D], 2e-2, 2e-2 * scale, "dV")
    close(dqkv[:, k_off:k_off + D], g[:, k_off:k_off + D], 3e-2, 3e-2 * scale, "dK")
    close(dqkv[:, q_off:q_off + D], g[:, q_off:q_off + D], 3e-2, 3e-2 * scale, "dQ")


@pytest.mark.parametrize("path", ["auto", "stream"])
def test_attention_q_limit_produces_only_the_first_queries(gpu, monkeypatch, path):
    if path == "stream":
        monkeypatch.setenv("MST_ATTN_PATH", "stream")
    o = ops()
    B, S, H, dh = 2, 70, 2, 32
    D = H * dh
    qkv = rnd((B * S, 3 * D), gpu, seed=34)
    keymask = torch.ones(B, S, dtype=torch.uint8, device=gpu)
    lse = torch.zeros(2, B, H, S, dtype=torch.float32, device=gpu)
    full = torch.zeros(B * S, D, dtype=BF, device=gpu)
    part = torch.full((B * S, D), 5.0, dtype=BF, device=gpu)
    o.attn_fwd(qkv, keymask, lse, full, B, S, H, dh, 0, D, 2 * D)
    o.attn_fwd(qkv, keymask, lse, part, B, S, H, dh, 0, D, 2 * D, q_limit=1)
    torch.cuda.synchronize()
    f3, p3 = full.view(B, S, D), part.view(B, S, D)
    assert torch.equal(p3[:, 0], f3[:, 0])
    assert (p3[:, 1:] == 5.0).all()


@pytest.mark.parametrize("B,S,H,dh,ragged", [(2, 70, 2, 32, False), (8, 256, 2, 32, True), (3, 257, 2, 16, True), (1, 1024, 2, 32, True)])
@pytest.mark.parametrize("path", ["auto", "stream"])
def test_attention_bwd_q_limit_is_bit_identical_to_dense(gpu, monkeypatch, B, S, H, dh, ragged, path):
    """dO that is zero outside query 0 (the top encoder layer): the sparse mode skips work, not arithmetic — in the resident
    kernel, in the streaming dV / dK kernel and in the chunked dQ kernel (the last shape: configs[4]'s sequence length)"""
    if path == "stream":
        monkeypatch.setenv("MST_ATTN_PATH", "stream")
    o = ops()
    D = H * dh
    qkv = rnd((B * S, 3 * D), gpu, seed=35)
    lens = torch.tensor([S - ((i * 11) % (S // 3) if ragged else 0) for i in range(B)], dtype=torch.int32, device=gpu)
    keymask = torch.zeros(B, S, dtype=torch.uint8, device=gpu)
    o.mask_from_lengths(lens, 0, keymask)
    lse = torch.zeros(2, B, H, S, dtype=torch.float32, device=gpu)
    out = torch.zeros(B * S, D, dtype=BF, device=gpu)
    o.attn_fwd(qkv, keymask, lse, out, B, S, H, dh, 0, D, 2 * D)
    dout = torch.zeros(B, S, D, dtype=BF, device=gpu)
    dout[:, 0] = rnd((B, D), gpu, seed=36)
    dout = dout.view(B * S, D)
    res = []
    for q_limit in (0, 1):
        dqkv = torch.full((B * S, 3 * D), 9.0, dtype=BF, device=gpu)
        delta = torch.zeros(B, H, S, dtype=torch.float32, device=gpu)
        o.attn_bwd(qkv, keymask, lse, dout, dqkv, delta, B, S, H, dh, 0, D, 2 * D, q_limit=q_limit)
        torch.cuda.synchronize()
        res.append((dqkv.clone(), delta.clone()))
    if dh == 16 and S % 32 == 1 and S > 32:
        # the dense kernel handles the LAST row of a 32 n + 1 sequence apart (partner rows on the lanes, fp32 probabilities:
        # attention.hip, lone_row_shape): its delta, which every query's dQ uses, agrees to rounding, not bit for bit
        sc = res[1][0].float().abs().max().item()
        close(res[0][0], res[1][0], 2e-2, 1e-2 * sc, "gradients")
        assert (res[0][0] != res[1][0]).float().mean().item() < 0.05
        close(res[0][1], res[1][1], 1e-2, 1e-2 * res[1][1].abs().max().item(), "delta")
    else:
        assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    assert res[0][0].float().abs().max().item() > 0


@pytest.mark.parametrize("path", ["auto", "stream"])
def test_attention_padded_key_rows_are_uniform(gpu, monkeypatch, path):
    """a padded key row is NOT excluded, it contributes V[k]/S to every query"""
    if path == "stream":
        monkeypatch.setenv("MST_ATTN_PATH", "stream")
    o = ops()
    B, S, H, dh = 1, 32, 1, 16
    D = H * dh
    qkv = rnd((B * S, 3 * D), gpu, seed=33)
    keymask = torch.zeros(B, S, dtype=torch.uint8, device=gpu)  # every key padded
    lse = torch.zeros(2, B, H, S, dtype=torch.float32, device=gpu)
    out = torch.zeros(B * S, D, dtype=BF, device=gpu)
    o.attn_fwd(qkv, keymask, lse, out, B, S, H, dh, 0, D, 2 * D)
    torch.cuda.synchronize()
    vmean_times_1 = qkv[:, 2 * D:].float().sum(0) / S  # sum_k V[k] / S
    close(out, vmean_times_1.expand(S, D), 1e-2, 1e-2, "uniform rows")


# ------------------------------------------------------------------------------------------ LayerNorm
@pytest.mark.parametrize("M,D", [(1000, 256), (77, 128), (16, 32), (300, 1024)])
def test_layernorm_fwd_bwd(gpu, M, D):
    o = ops()
    x = rnd((M, D), gpu, seed=40, scale=2.0)
    gamma = (1 + 0.1 * rnd((D,), gpu, dtype=torch.float32, seed=41)).contiguous()
    beta = rnd((D,), gpu, dtype=torch.float32, seed=42, scale=0.1)
    y = torch.zeros(M, D, dtype=BF, device=gpu)
    mean = torch.zeros(M, dtype=torch.float32, device=gpu)
    rstd = torch.zeros(M, dtype=torch.float32, device=gpu)
    o.layernorm_fwd(x, gamma, beta, y, mean, rstd)
    xr = x.float().clone().requires_grad_(True)
    gr = gamma.clone().requires_grad_(True)
    br = beta.clone().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xr, (D,), gr, br, 1e-5)
    torch.cuda.synchronize()
    close(y, ref.detach(), 1e-2, 1e-2, "ln fwd")
    close(mean, x.float().mean(1), 1e-5, 1e-5, "mean")
    dy = rnd((M, D), gpu, seed=43)
    dx = torch.zeros(M, D, dtype=BF, device=gpu)
    dg = torch.zeros(D, dtype=torch.float32, device=gpu)
    db = torch.zeros(D, dtype=torch.float32, device=gpu)
    o.layernorm_bwd(x, gamma, mean, rstd, dy, dx, dg, db)
    torch.cuda.synchronize()
    ref.backward(dy.float())
    close(dx, xr.grad, 1e-2, 1e-2, "ln dx")
    close(dg, gr.grad, 1e-3, 1e-3 * math.sqrt(M), "ln dgamma")
    close(db, br.grad, 1e-3, 1e-3 * math.sqrt(M), "ln dbeta")
    # per-workgroup partial sums + the deferred reduction instead of atomics: same dx, same sums, and bit-reproducible
    parts = o.layernorm_bwd_parts(M, D)
    outs = []
    for _ in range(2):
        part = torch.full((parts, 2 * D), float("nan"), device=gpu)
        dxp = torch.zeros(M, D, dtype=BF, device=gpu)
        dgb = torch.full((2 * D,), 1.0, device=gpu)  # accumulated INTO
        o.layernorm_bwd(x, gamma, mean, rstd, dy, dxp, None, None, partials=part)
        o.partial_sums([o.partial_sum_job(part, parts, dgb, scale=0.5)])
        torch.cuda.synchronize()
        outs.append(dgb.clone())
        assert torch.equal(dxp, dx)
    assert torch.equal(outs[0], outs[1])
    close(outs[0][:D], 1.0 + 0.5 * gr.grad, 1e-3, 1e-3 * math.sqrt(M), "dgamma from partials")
    close(outs[0][D:], 1.0 + 0.5 * br.grad, 1e-3, 1e-3 * math.sqrt(M), "dbeta from partials")
    # strided row subset (every 4th row), statistics indexed by the dense row id
    if M % 4 == 0:
        Ms = M // 4
        xs, dys = x.view(Ms, 4 * D)[:, :D], dy.view(Ms, 4 * D)[:, :D]
        y2 = torch.zeros(M, D, dtype=BF, device=gpu)
        mean2 = torch.zeros(M, dtype=torch.float32, device=gpu)
        rstd2 = torch.zeros(M, dtype=torch.float32, device=gpu)
        o.layernorm_fwd(xs, gamma, beta, y2.view(Ms, 4 * D)[:, :D], mean2, rstd2, D=D, M=Ms, row_id_stride=4)
        dx2 = torch.zeros(Ms, D, dtype=BF, device=gpu)
        dg2 = torch.zeros(D, dtype=torch.float32, device=gpu)
        db2 = torch.zeros(D, dtype=torch.float32, device=gpu)
        o.layernorm_bwd(xs, gamma, mean2, rstd2, dys, dx2, dg2, db2, D=D, M=Ms, row_id_stride=4)
        torch.cuda.synchronize()
        assert torch.equal(y2[::4], y[::4]) and (y2.view(Ms, 4, D)[:, 1:] == 0).all()
        assert torch.equal(mean2[::4], mean[::4]) and (mean2.view(Ms, 4)[:, 1:] == 0).all()
        assert torch.equal(dx2, dx[::4])
        close(db2, dy.float()[::4].sum(0), 1e-3, 1e-3 * math.sqrt(M), "strided dbeta")


def test_partial_sums_integer_exact(gpu):
    """mst_partial_sums: many jobs of different shapes in one launch; integer-valued data make every order exact"""
    o = ops()
    g = torch.Generator().manual_seed(7)
    jobs, want, dsts = [], [], []
    for n_parts, stride, length, off, scale in [(1, 8, 8, 0, 1.0), (257, 512, 256, 256, 1.0), (40, 132, 68, 64, 2.0),
                                                (3, 1024, 1024, 0, -1.0), (300, 64, 60, 4, 1.0)] * 7:  # 35 jobs: two launches
        src = torch.randint(-8, 9, (n_parts + 2, stride), generator=g).float().to(gpu)
        dst = torch.randint(-8, 9, (length,), generator=g).float().to(gpu)
        want.append(dst + scale * src[:n_parts, off:off + length].sum(0))
        jobs.append(o.partial_sum_job(src, n_parts, dst, scale=scale, col_off=off, length=length))
        dsts.append((src, dst))
    o.partial_sums(jobs)
    torch.cuda.synchronize()
    for (src, dst), w in zip(dsts, want):
        assert torch.equal(dst, w)


def test_layernorm_bwd_dropout_modes(gpu):
    o = ops()
    M, D = 64, 128
    x = rnd((M, D), gpu, seed=44)
    gamma = torch.ones(D, device=gpu)
    beta = torch.zeros(D, device=gpu)
    y = torch.zeros(M, D, dtype=BF, device=gpu)
    mean = torch.zeros(M, device=gpu)
    rstd = torch.zeros(M, device=gpu)
    o.layernorm_fwd(x, gamma, beta, y, mean, rstd)
    dy = rnd((M, D), gpu, seed=45)
    dx0 = torch.zeros(M, D, dtype=BF, device=gpu)
    dg = torch.zeros(D, device=gpu)
    db = torch.zeros(D, device=gpu)
    o.layernorm_bwd(x, gamma, mean, rstd, dy, dx0, dg, db)
    p, seed, site = 0.25, 99, 5
    keep = torch.zeros(M * D, dtype=torch.uint8, device=gpu)
    o.dropout_mask(M * D, p, seed, site, keep)
    k = keep.view(M, D).float() / (1 - p)
    dx1 = torch.zeros_like(dx0)
    dxm = torch.zeros_like(dx0)
    o.layernorm_bwd(x, gamma, mean, rstd, dy, dx1, dg, db, dx_masked=dxm, mask_mode=1, dropout_p=p, dropout_seed=seed,
                    dropout_site=site)
    dx2 = torch.zeros_like(dx0)
    o.layernorm_bwd(x, gamma, mean, rstd, dy, dx2, dg, db, mask_mode=2, dropout_p=p, dropout_seed=seed, dropout_site=site)
    dx3 = torch.zeros_like(dx0)
    o.layernorm_bwd(x, gamma, mean, rstd, dy, dx3, dg, db, mask_mode=2)
    torch.cuda.synchronize()
    assert torch.equal(dx0, dx1)
    close(dxm, dx0.float() * k, 1e-2, 1e-3, "masked copy")
    close(dx2, dx0.float() * (1 + k), 1.6e-2, 1e-3, "self-resid mode")
    close(dx3, dx0.float() * 2, 1e-2, 1e-3, "self-resid p=0")


# ------------------------------------------------------------------------------------------ embedding
def test_embed_fwd_bwd(gpu):
    o = ops()
    B, T, D, V, Cn = 4, 7, 32, 11, 3
    S_out, s_off = T + 1, 1
    tokens = torch.randint(0, V, (B, T), dtype=torch.int32).to(gpu)
    classes = torch.tensor([0, 1, 2, 1], dtype=torch.int32, device=gpu)
    table = rnd((V, D), gpu, dtype=torch.float32, seed=50)
    cls = rnd((Cn, D), gpu, dtype=torch.float32, seed=51)
    pos = rnd((S_out, D), gpu, dtype=torch.float32, seed=52)
    out = torch.zeros(B, S_out, D, dtype=BF, device=gpu)
    km = torch.zeros(B, S_out, dtype=torch.uint8, device=gpu)
    alpha = math.sqrt(D)
    o.embed_fwd(tokens, table, pos, out, s_off, alpha, classes=classes, cls_table=cls, keymask=km)
    torch.cuda.synchronize()
    ref = alpha * (table[tokens.long()] + cls[classes.long()].unsqueeze(1)) + pos[s_off:s_off + T]
    close(out[:, s_off:], ref, 1e-2, 1e-2, "embed fwd")
    assert (out[:, 0] == 0).all()
    assert torch.equal(km[:, s_off:], (tokens != 0).to(torch.uint8))
    dX = rnd((B, S_out, D), gpu, seed=53)
    dtab = torch.zeros(V, D, device=gpu)
    dcls = torch.zeros(Cn, D, device=gpu)
    o.embed_bwd(tokens, dtab, dX, s_off, alpha, classes=classes, dcls=dcls)
    torch.cuda.synchronize()
    g = alpha * dX[:, s_off:].float()
    rt = torch.zeros(V, D, device=gpu).index_add_(0, tokens.long().view(-1), g.reshape(-1, D))
    rc = torch.zeros(Cn, D, device=gpu).index_add_(0, classes.long(), g.sum(1))
    close(dtab, rt, 1e-4, 1e-3, "dtable")
    close(dcls, rc, 1e-4, 1e-3, "dcls")


@pytest.mark.parametrize("B,T,D,s_off", [(5, 100, 256, 0), (3, 7, 40, 1), (64, 256, 128, 0)])
def test_group_colsum(gpu, B, T, D, s_off):
    """class-embedding gradient: per-sample sums over frames, scattered to the sample's class row"""
    o = ops()
    S_out = T + s_off
    X = rnd((B, S_out, D), gpu, seed=60)
    idx = torch.tensor([(7 * i) % 4 for i in range(B)], dtype=torch.int32, device=gpu)
    dst = torch.zeros(4, D, device=gpu)
    o.group_colsum(X, T, D, s_off, idx, dst, 1.5)
    torch.cuda.synchronize()
    want = torch.zeros(4, D, device=gpu).index_add_(0, idx.long(), 1.5 * X[:, s_off:].float().sum(1))
    close(dst, want, 1e-5, 1e-3, "group_colsum")


# ------------------------------------------------------------------------------------------ latent
def test_latent_fwd_bwd(gpu):
    o = ops()
    B, S, De, Z, Dd, Cn, Sd = 5, 6, 64, 16, 32, 3, 7
    enc = rnd((B, S, De), gpu, seed=60)
    Wl = rnd((2 * Z, De), gpu, dtype=torch.float32, seed=61, scale=0.2)
    bl = rnd((2 * Z,), gpu, dtype=torch.float32, seed=62, scale=0.5)
    Wh = rnd((Dd, Z), gpu, dtype=torch.float32, seed=63, scale=0.3)
    bh = rnd((Dd,), gpu, dtype=torch.float32, seed=64, scale=0.1)
    cls_d = rnd((Cn, Dd), gpu, dtype=torch.float32, seed=65)
    pos_d = rnd((Sd, Dd), gpu, dtype=torch.float32, seed=66)
    eps = rnd((B, Z), gpu, dtype=torch.float32, seed=67)
    classes = torch.tensor([0, 1, 2, 1, 0], dtype=torch.int32, device=gpu)
    mu = torch.zeros(B, Z, device=gpu); sigma = torch.zeros(B, Z, device=gpu); z = torch.zeros(B, Z, device=gpu)
    kl = torch.zeros(B, device=gpu)
    dec_in = torch.zeros(B, Sd, Dd, dtype=BF, device=gpu)
    alpha_d = math.sqrt(Dd)
    o.latent_fwd(enc, Wl, bl, eps, Wh, bh, classes, cls_d, pos_d, alpha_d, mu, sigma, z, kl, dec_in)
    torch.cuda.synchronize()

    h0 = enc[:, 0].float().clone().requires_grad_(True)
    P = [t.clone().requires_grad_(True) for t in (Wl, bl, Wh, bh, cls_d)]
    lat = h0 @ P[0].t() + P[1]
    mu_r, sg_r = lat[:, :Z], lat[:, Z:]
    z_r = mu_r + eps * sg_r
    kl_r = 0.5 * (sg_r * sg_r + mu_r * mu_r - 1 - torch.log(sg_r * sg_r)).sum(1)
    d0 = alpha_d * (z_r @ P[2].t() + P[3] + P[4][classes.long()]) + pos_d[0]
    close(mu, mu_r.detach(), 1e-5, 1e-5, "mu")
    close(sigma, sg_r.detach(), 1e-5, 1e-5, "sigma")
    close(z, z_r.detach(), 1e-5, 1e-5, "z")
    close(kl, kl_r.detach(), 1e-5, 1e-4, "kl")
    close(dec_in[:, 0], d0.detach(), 1e-2, 1e-2, "dec_in row 0")
    assert (dec_in[:, 1:] == 0).all()

    g0 = rnd((B, Sd, Dd), gpu, seed=68, scale=0.1)
    beta = 0.7
    dWl = torch.zeros_like(Wl); dbl = torch.zeros_like(bl); dWh = torch.zeros_like(Wh); dbh = torch.zeros_like(bh)
    dcls = torch.zeros_like(cls_d)
    denc = torch.zeros(B, S, De, dtype=BF, device=gpu)
    scratch = torch.zeros(B * (Dd + 2 * Z), device=gpu)
    o.latent_bwd(enc, Wl, eps, Wh, classes, mu, sigma, z, g0, alpha_d, beta, 1.0, dWl, dbl, dWh, dbh, dcls, denc, scratch)
    torch.cuda.synchronize()
    total = (d0 * g0[:, 0].float()).sum() + beta * kl_r.sum()
    total.backward()
    for got, ref, name in zip((dWl, dbl, dWh, dbh, dcls), (p.grad for p in P), ("dWl", "dbl", "dWh", "dbh", "dcls")):
        close(got, ref, 1e-4, 1e-4 * max(1.0, ref.abs().max().item()), name)
    close(denc[:, 0], h0.grad, 1e-2, 1e-2 * h0.grad.abs().max().item(), "d enc row 0")

    # the deferred form (mst_latent_bwd_vec + two mst_outer_job): identical arithmetic, (a) as one mst_outer_jobs launch,
    # (b) as extra workgroups of a weight-gradient flush that has a reduction pass, (c) behind a flush that has none
    M, N, K = 1024, 256, 256  # a problem of the whole-step tile form (two-pass reduction through the scratch buffer)
    A, Bm = rnd((M, N), gpu, seed=69), rnd((M, K), gpu, seed=70)
    for mode in ("own launch", "reduction pass", "no reduction pass"):
        g = [torch.zeros_like(t) for t in (Wl, bl, Wh, bh, cls_d)]
        denc2 = torch.zeros(B, S, De, dtype=BF, device=gpu)
        scratch2 = torch.zeros(B * (Dd + 2 * Z), device=gpu)
        o.latent_bwd_vec(Wl, eps, Wh, classes, mu, sigma, g0, alpha_d, beta, 1.0, g[4], denc2, scratch2)
        jobs = o.latent_outer_jobs(scratch2, enc, z, g[0], g[1], g[2], g[3])
        dW = torch.zeros(N, K, device=gpu)
        if mode == "own launch":
            o.outer_jobs(jobs)
        else:
            ws = torch.zeros(16 * 1024 * 1024, device=gpu) if mode == "reduction pass" else None
            o.gemm_wgrad_batch([o.wgrad_problem(A, Bm, dW)], scratch=ws, outers=jobs)
        torch.cuda.synchronize()
        assert torch.equal(scratch2, scratch) and torch.equal(denc2, denc), mode
        for got, want, name in zip(g[:4], (dWl, dbl, dWh, dbh), ("dWl", "dbl", "dWh", "dbh")):
            assert torch.equal(got, want), f"{name} ({mode})"
        close(g[4], dcls, 1e-6, 1e-6, f"dcls ({mode})")  # (atomics: the order of the adds is free)
        if mode != "own launch":
            close(dW, A.float().t() @ Bm.float(), 1e-3, 1e-3 * math.sqrt(M), f"the flush's own problem ({mode})")


def test_reparam_kl_known_answers(gpu):
    o = ops()
    B, Z = 3, 8
    mu = torch.zeros(B, Z, device=gpu)
    sigma = torch.tensor([[1.0] * Z, [-1.0] * Z, [2.0] * Z], device=gpu)
    eps = torch.ones(B, Z, device=gpu)
    z = torch.zeros(B, Z, device=gpu); kl = torch.zeros(B, device=gpu)
    o.reparam_kl_fwd(mu, sigma, eps, z, kl)
    torch.cuda.synchronize()
    # KL(mu=0, sigma=+-1) = 0 ; sigma=2: 0.5*(4 - 1 - log 4) per dim
    want = torch.tensor([0.0, 0.0, Z * 0.5 * (3 - math.log(4.0))])
    close(kl, want, 1e-6, 1e-6, "kl KAT")
    assert torch.equal(z.cpu(), sigma.cpu())
    dmu = torch.zeros(B, Z, device=gpu); dsg = torch.zeros(B, Z, device=gpu)
    dz = torch.full((B, Z), 0.5, device=gpu)
    o.reparam_kl_bwd(mu, sigma, eps, dz, 2.0, dmu, dsg)
    torch.cuda.synchronize()
    close(dmu, torch.full((B, Z), 0.5), 1e-6, 1e-6, "dmu")
    close(dsg, 2.0 * (sigma.cpu() - 1 / sigma.cpu()) + 0.5, 1e-6, 1e-6, "dsigma")


# ------------------------------------------------------------------------------------------ loss heads
@pytest.mark.parametrize("B,T,V", [(3, 5, 10), (8, 65, 293), (2, 4, 2048)])
def test_softmax_ce(gpu, B, T, V):
    o = ops()
    ldv = o.roundup(V, 8)
    logits = torch.zeros(B * T, ldv, dtype=BF, device=gpu)
    logits[:, :V] = rnd((B * T, V), gpu, seed=70, scale=2.0)
    labels = torch.randint(0, V, (B * T,), dtype=torch.int32).to(gpu)
    labels[::4] = 0  # PAD positions are masked out
    loss = torch.zeros(B, device=gpu)
    probs = torch.zeros(B * T, V, device=gpu)
    dlog = torch.full((B * T, ldv), 3.0, dtype=BF, device=gpu)
    o.softmax_ce(logits, labels, loss, B, T, V, probs=probs, dlogits=dlog, gscale=1.0)
    torch.cuda.synchronize()
    lr = logits[:, :V].float().clone().requires_grad_(True)
    p = torch.softmax(lr, -1)
    mask = (labels != 0).float()
    nll = -torch.log(p.gather(1, labels.long().view(-1, 1)).squeeze(1)) * mask
    ref = nll.view(B, T).mean(1)  # divides by T, not by the number of valid tokens (loss.py:23)
    close(loss, ref.detach(), 1e-5, 1e-5, "ce loss")
    close(probs, p.detach(), 1e-4, 1e-6, "probs")
    ref.sum().backward()
    close(dlog[:, :V], lr.grad, 1e-2, 1e-5, "dlogits")
    assert (dlog[:, V:min(o.roundup(V, 4), ldv)] == 0).all()


def test_softmax_ce_uniform_known_answer(gpu):
    o = ops()
    B, T, V = 2, 6, 10
    logits = torch.zeros(B * T, 16, dtype=BF, device=gpu)
    labels = torch.tensor([1, 2, 3, 0, 0, 0, 4, 5, 6, 7, 8, 0], dtype=torch.int32, device=gpu)
    loss = torch.zeros(B, device=gpu)
    o.softmax_ce(logits, labels, loss, B, T, V)
    torch.cuda.synchronize()
    want = torch.tensor([math.log(V) * 3 / T, math.log(V) * 5 / T])
    close(loss, want, 1e-6, 1e-6, "uniform CE = log V * n_valid / T")


@pytest.mark.parametrize("B,T,P,ls,dw,scale", [(4, 16, 128, 0.0, False, 2.0), (4, 16, 128, 0.1, True, 2.0), (3, 7, 30, 0.0, True, 2.0),
                                               (64, 256, 128, 0.0, False, 2.0),
                                               # saturated logits (beyond [-16, 9]): waves that hold one take the reference's own
                                               # operation order, 1e-12 epsilons and fl(1 - p) included (bce_math.hpp)
                                               (4, 16, 128, 0.1, False, 8.0), (3, 7, 30, 0.0, True, 8.0)])
def test_sigmoid_bce(gpu, B, T, P, ls, dw, scale):
    o = ops()
    ldp = o.roundup(P, 8)
    logits = torch.zeros(B * T, ldp, dtype=BF, device=gpu)
    logits[:, :P] = rnd((B * T, P), gpu, seed=80, scale=scale)
    g = torch.Generator().manual_seed(81)
    labels = (torch.rand(B * T, P, generator=g) < 0.1).to(torch.uint8).to(gpu)
    loss = torch.zeros(B, device=gpu)
    npos = torch.zeros(B, dtype=torch.int32, device=gpu)
    probs = torch.zeros(B * T, ldp, dtype=BF, device=gpu)
    dlog = torch.zeros(B * T, ldp, dtype=BF, device=gpu)
    o.sigmoid_bce(logits, labels, loss, B, T, P, label_smoothing=ls, downweight=dw, npos=npos, probs=probs, dlogits=dlog)
    torch.cuda.synchronize()
    # plain restatement of loss.py:38-80
    x = logits[:, :P].float().clone().requires_grad_(True)
    y = labels.float()
    p = torch.sigmoid(x)
    s = (1 - ls) * y + ls * 0.5
    bce = -(s * torch.log(1e-12 + p) + (1 - s) * torch.log(1e-12 + (1 - p)))
    if dw:
        y3 = y.view(B, T * P)
        npos_r = (y3 == 1).float().sum(1)
        nneg_r = (y3 != 1).float().sum(1)
        w = (npos_r / (nneg_r + 1e-12)).view(B, 1).expand(B, T * P).reshape(B * T, P)
        bce = torch.where(y == 0, (w * bce) * bce, bce)
        assert torch.equal(npos.cpu().float(), npos_r.cpu())
    ref = bce.view(B, T * P).mean(1)
    close(loss, ref.detach(), 2e-4, 1e-6, "bce loss")
    close(probs[:, :P], p.detach(), 1e-2, 1e-3, "probs")
    ref.sum().backward()
    gmax = x.grad.abs().max().item()
    close(dlog[:, :P], x.grad, 1.6e-2, 1e-2 * gmax * 0.01 + 1e-12, "dlogits")


@pytest.mark.parametrize("B,T,P,D,ls,dw,dtype", [(4, 128, 128, 128, 0.0, False, BF), (3, 64, 256, 64, 0.1, True, torch.float16),
                                                 (64, 256, 128, 128, 0.0, False, BF),
                                                 # rows of several 256-pitch column tiles (configs[2]: 2048 pitches)
                                                 (3, 64, 512, 128, 0.1, False, BF), (2, 128, 2048, 128, 0.0, False, torch.float16)])
def test_gemm_sigmoid_bce_equals_gemm_then_bce(gpu, B, T, P, D, ls, dw, dtype):
    """mst_gemm_sigmoid_bce (output layer + sigmoid + BCE in one launch, logits never stored) against mst_gemm_nt followed
    by mst_sigmoid_bce on the same operands, with the decoder's row remap (rows 1..T of T+1): bit-identical logit gradient
    and probabilities (same MFMA order, logits rounded to the activation type before the loss arithmetic), loss to fp32
    summation order"""
    o = ops()
    Sd = T + 1
    x = rnd((B * Sd, D), gpu, 1.0, dtype, seed=21)
    W = rnd((P, D), gpu, 0.2, dtype, seed=22)
    bias = rnd((P,), gpu, 0.1, torch.float32, seed=23)
    g = torch.Generator().manual_seed(24)
    labels = (torch.rand(B * T, P, generator=g) < 0.05).to(torch.uint8).to(gpu)
    # two launches
    logits = torch.zeros(B * T, P, dtype=dtype, device=gpu)
    o.gemm_nt(x, W, logits, M=B * T, K=D, bias=bias, a_remap=(T, Sd, 1))
    loss2 = torch.zeros(B, dtype=torch.float32, device=gpu)
    npos = torch.zeros(B, dtype=torch.int32, device=gpu)
    dl2, pr2 = torch.zeros_like(logits), torch.zeros_like(logits)
    o.sigmoid_bce(logits, labels, loss2, B, T, P, label_smoothing=ls, downweight=dw, npos=npos, probs=pr2, dlogits=dl2, gscale=4.0)
    # one launch
    loss1 = torch.zeros(B, dtype=torch.float32, device=gpu)
    dl1, pr1, lg1 = torch.zeros_like(logits), torch.zeros_like(logits), torch.zeros_like(logits)
    o.gemm_sigmoid_bce(x, W, labels, loss1, T, dlogits=dl1, probs=pr1, logits=lg1, label_smoothing=ls, downweight=dw, gscale=4.0,
                       M=B * T, K=D, bias=bias, a_remap=(T, Sd, 1))
    torch.cuda.synchronize()
    assert torch.equal(lg1, logits)
    assert torch.equal(pr1, pr2)
    if dw:  # w * bce^2 and its gradient: the two kernels' FMA contraction differs in the last bit of a few elements
        assert torch.allclose(dl1.float(), dl2.float(), rtol=4e-3, atol=1e-9)
    else:
        assert torch.equal(dl1, dl2)
    assert torch.allclose(loss1, loss2, rtol=2e-5, atol=0)
    # forward only: no gradient buffer
    loss3 = torch.zeros(B, dtype=torch.float32, device=gpu)
    o.gemm_sigmoid_bce(x, W, labels, loss3, T, label_smoothing=ls, downweight=dw, M=B * T, K=D, bias=bias, a_remap=(T, Sd, 1))
    torch.cuda.synchronize()
    assert torch.allclose(loss3, loss2, rtol=2e-5, atol=0)


@pytest.mark.parametrize("B,T,P,D,drop,dtype", [(64, 256, 128, 128, 0.2, BF), (4, 64, 128, 128, 0.0, torch.float16), (3, 64, 256, 128, 0.0, BF)])
def test_gemm_sigmoid_bce_dgrad_ln_equals_the_two_launches(gpu, B, T, P, D, drop, dtype):
    """mst_gemm_sigmoid_bce_dgrad_ln: the loss launch with the output layer's input gradient + the decoder's LayerNorm-3 backward
    (mask mode 2, row remap into rows 1..T of T+1) in the same workgroup — bit-identical to mst_gemm_sigmoid_bce followed by
    mst_gemm_nt_ln (mode 2) on the stored logit gradient, also where the form does not apply (256 pitches: two launches inside)"""
    o = ops()
    Sd, M = T + 1, B * T
    x = rnd((B * Sd, D), gpu, 1.0, dtype, seed=31)
    W = rnd((P, D), gpu, 0.2, dtype, seed=32)
    Wt = W.t().contiguous()  # [D, P]: the transposed shadow the dgrad reads
    bias = rnd((P,), gpu, 0.1, torch.float32, seed=33)
    g = torch.Generator().manual_seed(34)
    labels = (torch.rand(M, P, generator=g) < 0.05).to(torch.uint8).to(gpu)
    h2 = rnd((B * Sd, D), gpu, 1.0, dtype, seed=35)
    gamma = (1 + 0.1 * rnd((D,), gpu, 1.0, torch.float32, seed=36))
    mean, rstd = h2.float().mean(1), (h2.float().var(1, unbiased=False) + 1e-5).rsqrt()
    seedp = torch.tensor([77, 0, 0, 0], dtype=torch.int64, device=gpu)
    outs = []
    for fused in (True, False):
        loss = torch.zeros(B, dtype=torch.float32, device=gpu)
        dl, pr = torch.zeros(M, P, dtype=dtype, device=gpu), torch.zeros(M, P, dtype=dtype, device=gpu)
        dh = torch.zeros(B * Sd, D, dtype=dtype, device=gpu)
        parts = torch.zeros(o.gemm_nt_ln_parts(M), 2 * D, device=gpu)
        dg, db = torch.zeros(D, device=gpu), torch.zeros(D, device=gpu)
        dgrad = dict(A=dl, B=Wt, dX_out=dh, x=h2, gamma=gamma, mean=mean, rstd=rstd, dgamma=dg, dbeta=db, mask_mode=2, partials=parts,
                     M=M, N=D, K=P, c_remap=(T, Sd, 1))
        if drop > 0:
            dgrad.update(dropout_p=drop, dropout_seed_ptr=seedp, dropout_site=9)
        kw = dict(dlogits=dl, probs=pr, label_smoothing=0.1, downweight=True, gscale=4.0, M=M, K=D, bias=bias, a_remap=(T, Sd, 1))
        if fused:
            o.gemm_sigmoid_bce(x, W, labels, loss, T, dgrad=dgrad, **kw)
        else:
            o.gemm_sigmoid_bce(x, W, labels, loss, T, **kw)
            o.gemm_nt_ln_bwd(**dgrad)
        torch.cuda.synchronize()
        outs.append((loss, dl, pr, dh, parts))
    for a_, b_, name in zip(outs[0], outs[1], ("loss", "dlogits", "probs", "dh", "LayerNorm partials")):
        if name == "loss":
            assert torch.allclose(a_, b_, rtol=1e-6, atol=0), name
        else:
            assert torch.equal(a_, b_), name
    assert outs[0][3].abs().sum() > 0 and (outs[0][3].view(B, Sd, D)[:, 0] == 0).all()


def test_bce_logit_zero_known_answer(gpu):
    o = ops()
    B, T, P = 2, 4, 8
    logits = torch.zeros(B * T, 8, dtype=BF, device=gpu)
    labels = torch.zeros(B * T, P, dtype=torch.uint8, device=gpu)
    labels[::2] = 1
    loss = torch.zeros(B, device=gpu)
    o.sigmoid_bce(logits, labels, loss, B, T, P)
    torch.cuda.synchronize()
    close(loss, torch.full((B,), math.log(2.0)), 1e-6, 1e-6, "BCE at logit 0 = log 2")


def test_loss_combine(gpu):
    o = ops()
    B = 70
    recon = rnd((B,), gpu, dtype=torch.float32, seed=90).abs()
    kl = rnd((B,), gpu, dtype=torch.float32, seed=91).abs()
    total = torch.zeros(B, device=gpu)
    acc = torch.zeros(3, device=gpu)
    o.loss_combine(recon, kl, 0.5, total, acc)
    o.loss_combine(recon, kl, 0.5, total, acc)
    torch.cuda.synchronize()
    close(total, recon + 0.5 * kl, 1e-6, 1e-6, "total")
    close(acc, torch.stack([2 * kl.sum(), 2 * (recon + 0.5 * kl).sum(), torch.tensor(2.0 * B, device=gpu)]), 1e-5, 1e-4, "acc")


# ------------------------------------------------------------------------------------------ optimizer
def mxnet_adam_reference(w, g, m, v, t, lr, b1, b2, eps, wd, rescale, clip):
    g = g * rescale + wd * w
    if clip >= 0:
        g = g.clamp(-clip, clip)
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    lr_t = lr * math.sqrt(1 - b2 ** t) / (1 - b1 ** t)
    w = w - lr_t * m / (v.sqrt() + eps)
    return w, m, v


def test_adam_flat_mxnet_rule(gpu):
    o = ops()
    n = 10007
    w = rnd((n,), gpu, dtype=torch.float32, seed=100)
    w0 = w.clone()
    m = torch.zeros(n, device=gpu); v = torch.zeros(n, device=gpu)
    w16 = torch.zeros(n, dtype=BF, device=gpu)
    state = torch.zeros(2, dtype=torch.int32, device=gpu)
    wr, mr, vr = w0.clone(), m.clone(), v.clone()
    # the end-of-step bookkeeping that may ride on the launch: the same numbers as mst_loss_combine
    B = 37
    recon, kl = rnd((B,), gpu, dtype=torch.float32, seed=98).abs(), rnd((B,), gpu, dtype=torch.float32, seed=99).abs()
    total, metric = torch.zeros(B, device=gpu), torch.zeros(3, device=gpu)
    total_ref, metric_ref = torch.zeros(B, device=gpu), torch.zeros(3, device=gpu)
    for t in range(1, 4):
        g = rnd((n,), gpu, dtype=torch.float32, seed=100 + t, scale=50.0)
        o.adam_flat(w, g, m, v, w16, state, lr=3e-4, rescale=1 / 32, clip=1.0,
                    metrics=dict(recon=recon, kl=kl, kl_weight=0.5, total=total, metric=metric) if t != 2 else None)
        if t != 2:
            o.loss_combine(recon, kl, 0.5, total_ref, metric_ref)
        wr, mr, vr = mxnet_adam_reference(wr, g, mr, vr, t, 3e-4, 0.9, 0.999, 1e-8, 0.0, 1 / 32, 1.0)
    torch.cuda.synchronize()
    assert torch.equal(total, total_ref) and torch.equal(metric, metric_ref) and metric[2].item() == 2 * B
    assert state[0].item() == 3
    close(w, wr, 1e-6, 1e-7, "adam w")
    close(m, mr, 1e-5, 1e-7, "adam m")
    close(v, vr, 1e-4, 1e-9, "adam v")  # (1-beta2) is formed in fp32 on the device, as in MXNet
    assert torch.equal(w16, w.to(BF))


def test_transpose_shadows_and_cast(gpu):
    o = ops()
    shapes = [(10, 32), (293, 128), (128, 64)]
    offs, total = [], 0
    for r, c in shapes:
        offs.append(total)
        total += r * c
    w = rnd((total,), gpu, dtype=torch.float32, seed=110)
    desc, prefix, doff = [], [0], 0
    for (r, c), so in zip(shapes, offs):
        desc += [so, doff, r, c]
        doff += c * o.roundup(r, 8)
        prefix.append(prefix[-1] + ((r + 31) // 32) * ((c + 31) // 32))
    wt = torch.full((doff,), 9.0, dtype=BF, device=gpu)
    o.transpose_shadows(w, wt, torch.tensor(desc, dtype=torch.int64, device=gpu),
                        torch.tensor(prefix, dtype=torch.int64, device=gpu), len(shapes), prefix[-1])
    w16 = torch.zeros(total, dtype=BF, device=gpu)
    o.cast_to_act(w, w16)
    torch.cuda.synchronize()
    assert torch.equal(w16, w.to(BF))
    d = 0
    for (r, c), so in zip(shapes, offs):
        ldt = o.roundup(r, 8)
        got = wt[d:d + c * ldt].view(c, ldt)
        assert torch.equal(got[:, :r], w[so:so + r * c].view(r, c).t().to(BF))
        assert (got[:, r:] == 0).all()
        d += c * ldt


def test_graph_capture_and_events(gpu):
    o = ops()
    a = rnd((1024,), gpu, seed=120)
    b = rnd((1024,), gpu, seed=121)
    y = torch.zeros(1024, dtype=BF, device=gpu)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        g = o.Graph().capture(lambda: (o.add_act(a, b, y), o.add_act(y, b, y)))
        e0, e1 = o.Event(), o.Event()
        e0.record()
        g.launch()
        e1.record()
        e1.sync()
        assert e0.elapsed_ms(e1) >= 0.0
    torch.cuda.synchronize()
    want = ((a.float() + b.float()).to(BF).float() + b.float()).to(BF)
    assert torch.equal(y, want)


def test_step_begin_matches_the_separate_kernels(gpu):
    """mst_step_begin (many workgroups, last-arriver write-back) == rng_advance + randn + mask_from_lengths + Adam tick"""
    o = ops()
    B, Z, Se, Sd = 64, 64, 256, 257
    lens = torch.tensor([(i * 37) % Se + 1 for i in range(B)], dtype=torch.int32, device=gpu)
    st_a = torch.tensor([0, 0, 1234567, 0], dtype=torch.int64, device=gpu)
    st_b = st_a.clone()
    adam = torch.zeros(2, dtype=torch.int32, device=gpu)
    for step in range(1, 4):
        eps = torch.zeros(B, Z, device=gpu)
        me = torch.full((B, Se), 7, dtype=torch.uint8, device=gpu)
        md = torch.full((B, Sd), 7, dtype=torch.uint8, device=gpu)
        o.step_begin(rng_state=st_a, adam_state=adam, lr=1e-3, eps_out=eps, eps_site=99, lens=lens, mask_e=me, add_e=0,
                     mask_d=md, add_d=1)
        o.rng_advance(st_b)
        eps_ref = torch.zeros(B, Z, device=gpu)
        o.randn(eps_ref, seed_ptr=st_b, site=99)
        me_ref, md_ref = torch.zeros_like(me), torch.zeros_like(md)
        o.mask_from_lengths(lens, 0, me_ref)
        o.mask_from_lengths(lens, 1, md_ref)
        torch.cuda.synchronize()
        assert st_a.tolist() == st_b.tolist() and st_a[1].item() == step and st_a[3].item() == 0
        assert torch.equal(eps, eps_ref) and torch.equal(me, me_ref) and torch.equal(md, md_ref)
        assert adam[0].item() == step
        want_lr = 1e-3 * math.sqrt(1 - 0.999 ** step) / (1 - 0.9 ** step)
        assert abs(adam.view(torch.float32)[1].item() - want_lr) < 1e-6 * want_lr + 1e-12
    assert abs(float(eps.mean())) < 0.06 and abs(float(eps.std()) - 1) < 0.05


def test_step_begin_riding_on_the_embedding_launch_equals_its_own_launch(gpu):
    """mst_gemm_nt_pair_begin: the step's bookkeeping as the first workgroups of the two embedding GEMMs' launch (256 threads each
    instead of 1024) — RNG state, Adam tick, eps, both masks and the cleared buffers identical to mst_step_begin, the GEMM outputs
    identical to mst_gemm_nt_pair; also where the pair falls back to separate launches (ragged M)"""
    o = ops()
    g = torch.Generator().manual_seed(23)
    for B, T, P, De, Dd in ((64, 256, 128, 256, 128), (3, 50, 40, 64, 64)):
        M, Z = B * T, 64
        frames = (torch.rand(M, P, generator=g) < 0.05).to(torch.uint8).to(gpu)
        te, td = rnd((De, P), gpu, 0.1, BF, seed=24), rnd((Dd, P), gpu, 0.1, BF, seed=25)
        pos_e, pos_d = rnd((T, De), gpu, 1.0, torch.float32, seed=26), rnd((T + 1, Dd), gpu, 1.0, torch.float32, seed=27)
        lens = torch.tensor([(i * 37) % T + 1 for i in range(B)], dtype=torch.int32, device=gpu)
        results = []
        for ride in (True, False):
            state = torch.tensor([0, 5, 1234567, 0], dtype=torch.int64, device=gpu)
            adam = torch.tensor([5, 0], dtype=torch.int32, device=gpu)
            eps = torch.zeros(B, Z, device=gpu)
            me = torch.full((B, T), 7, dtype=torch.uint8, device=gpu)
            md = torch.full((B, T + 1), 7, dtype=torch.uint8, device=gpu)
            za, zb = torch.full((1024 + 12,), 3.0, device=gpu), torch.full((300000,), 3.0, device=gpu)
            xe = torch.zeros(M, De, dtype=BF, device=gpu)
            xd = torch.zeros(B * (T + 1), Dd, dtype=BF, device=gpu)
            begin = dict(rng_state=state, adam_state=adam, lr=1e-3, eps_out=eps, eps_site=99, eps_index0=128, lens=lens, mask_e=me,
                         add_e=0, mask_d=md, add_d=1, zero_a=za, zero_b=zb)
            first = dict(A=frames, B=te, C_out=xe, N=De, alpha=1.5, rowadd=pos_e, rowadd_period=T)
            second = dict(A=frames, B=td, C_out=xd, M=M, N=Dd, alpha=0.5, rowadd=pos_d[1:], rowadd_period=T, c_remap=(T, T + 1, 1))
            if ride:
                o.gemm_nt_pair(first, second, begin=begin)
            else:
                o.step_begin(**begin)
                o.gemm_nt_pair(first, second)
            torch.cuda.synchronize()
            results.append((state, adam, eps, me, md, za, zb, xe, xd))
        for a_, b_, name in zip(results[0], results[1], ("rng state", "adam state", "eps", "mask_e", "mask_d", "zero_a", "zero_b", "x0_e", "x0_d")):
            assert torch.equal(a_, b_), f"{name} (B={B})"
        assert results[0][0][1].item() == 6 and results[0][0][3].item() == 0 and results[0][1][0].item() == 6
        assert (results[0][5] == 0).all() and (results[0][6] == 0).all() and results[0][7].abs().sum() > 0
