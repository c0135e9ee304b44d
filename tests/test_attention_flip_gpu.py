"""The padded-key "flip" regime of MultiHeadDotAttention (transformer.py:106-126) with DETERMINISTIC arithmetic.

The reference adds -1e9 in fp32 to every logit of a padded key row before a softmax over the QUERY axis. fp32 has a
spacing of 64 at 1e9, so fl(x/sqrt(dh) - 1e9) lands on the grid -1e9 + 64 n: logits with |x| <= 32 collapse onto -1e9 (a
uniform row), larger ones land on another grid point and the row stops being uniform — its mass goes to the queries on
the highest grid point. The kernels reproduce this with an "exact" tile form chosen per 32-key tile that holds a padded
key; everywhere else they use a fused exp2 form that would give a uniform row.

Random unit-scale Q, K (test_kernels_gpu.py::test_attention_fwd_bwd) never reach |logit| >= 32, and with real-valued data
the grid point a logit lands on depends on the last bit of x. Here Q and K are INTEGER-valued, so K.Q is exact in fp32 in any
accumulation order, head size 16 makes the score scale 0.25 exact (head size 32: the scaled logits are asserted to stay
1e-3 away from every rounding boundary), and the padded keys' logits sit at 0, +-31, +-32 (the tie), +-33, +-96 and
around them, in tiles that mix padded and valid keys, with a ragged last tile. The reference's grid is then
reproducible, and the kernels must land on it: the row maximum plane of `lse` is compared EXACTLY on padded rows, the
outputs and the three gradients against autograd through the fp32 restatement — on the resident, restaged
(two-tile), chunked and streaming paths, in bf16 and fp16, dense and with q_limit = 1 (the top encoder layer's form).
"""
import math

import numpy as np
import pytest
import torch

MASK_VALUE = -1e9


def _ops():
    from musicstyletransfer_amd import ops as o
    return o


def reference(qkv, keymask, B, S, H, dh):
    """transformer.py:85-104 in fp32 on the CPU, the reference's operation order: gemm2(K, Q^T) / sqrt(dh), + mask,
    softmax over the last (query) axis, probs^T V. Returns (out [B*S, D], logits [B,H,k,q] with the mask added)."""
    D = H * dh
    x = qkv.view(B, S, 3 * D)
    heads = lambda off: x[:, :, off:off + D].reshape(B, S, H, dh).permute(0, 2, 1, 3)
    K, Q, V = heads(0), heads(D), heads(2 * D)
    logits = torch.matmul(K, Q.transpose(-1, -2))
    logits = logits / torch.sqrt(torch.tensor(float(dh), dtype=torch.float32))
    mask = torch.where(keymask > 0, torch.zeros(B, S), torch.ones(B, S) * MASK_VALUE).to(torch.float32)
    logits = logits + mask[:, None, :, None]
    P = torch.softmax(logits, dim=-1)
    out = torch.matmul(P.transpose(-1, -2), V).permute(0, 2, 1, 3).reshape(B * S, D)
    mass = torch.matmul(P.detach().transpose(-1, -2), V.detach().abs()).permute(0, 2, 1, 3).reshape(B * S, D)  # sum_k P |V|: the error scale
    return out, logits, mass


def integer_case(B, S, H, dh, seed):
    """qkv [B*S, 3 D] (K | Q | V) of small integers and a keymask whose padded keys see logits on both sides of +-32 and +-96.
    Per head: K[k] = (c_k, 1, r...), Q[q] = (u_q, w_q, r'...) with r, r' in {-1, 0, 1}: K.Q = c_k u_q + w_q + r.r'."""
    rng = np.random.default_rng(seed)
    D = H * dh
    unit = 4 if dh == 16 else (6 if dh == 32 else 8)  # c_k * u / sqrt(dh) ~ u for c_k = `unit` (head size 32: 6 / 5.657)
    levels = np.array([0, 31, -31, 32, -32, 33, -33, 96, -96, 30, -34, 97, -95, 1, 64, -64])
    K = rng.integers(-1, 2, size=(B, S, H, dh)).astype(np.float32)
    Q = rng.integers(-1, 2, size=(B, S, H, dh)).astype(np.float32)
    V = rng.integers(-4, 5, size=(B, S, H, dh)).astype(np.float32)
    c = rng.choice(np.array([0, 1, 2, unit, unit, unit, -unit]), size=(B, S, H))
    u = levels[rng.integers(0, len(levels), size=(B, S, H))]
    if dh == 32:  # (scaled by 1 / 5.657: pre-scale the levels so that unit * u / sqrt(32) sits near them; stays an integer)
        u = np.round(u * (math.sqrt(32.0) / unit)).astype(np.int64)
    K[..., 0], K[..., 1] = c, 1.0
    Q[..., 0], Q[..., 1] = u, rng.integers(-2, 3, size=(B, S, H))
    qkv = np.concatenate([K.reshape(B * S, D), Q.reshape(B * S, D), V.reshape(B * S, D)], axis=1)
    assert np.abs(qkv).max() <= 256, "every entry must be an integer bf16 holds exactly"
    mask = np.ones((B, S), np.uint8)
    if B > 1:
        mask[1, S - min(37, S // 2):] = 0        # contiguous padding that starts inside a 32-key tile (ragged lengths)
    if B > 2:
        mask[2, 3::5] = 0                        # scattered padded keys: every tile mixes padded and valid keys
        mask[2, S - 3:] = 0
    if B == 1:
        mask[0, S - min(45, S // 2):] = 0
        mask[0, 7::11] = 0
    return torch.from_numpy(qkv), torch.from_numpy(mask)


def rounding_bounds(qkv32, dout32, masked_logits, B, S, H, dh, subnormal=0.0):
    """per-element error scales of dV, dK, dQ (see the test): sum_q P |dO|, scale * sum_q P (|dP| + |delta|) |Q|, its transpose
    with |K|, in units of one relative rounding. `subnormal` (fp16: 2^-24, the spacing below 6.1e-5 — measured: P[k, 0] of 2e-7
    is a multiple of 6e-8 there, tools/experiments/diag_flip_sparse.py) adds the ABSOLUTE rounding of every 16-bit P / dS term,
    expressed in the same units by the caller dividing by its ulp: |x| + (|dO| column sums, |V| row sums) for dV and delta, and the
    |Q| / |K| column sums for dK / dQ."""
    D = H * dh
    with torch.no_grad():
        x3 = qkv32.view(B, S, 3 * D)
        hd = lambda t: t.reshape(B, S, H, dh).permute(0, 2, 1, 3)
        Kh, Qh, Vh, dOh = hd(x3[:, :, :D]), hd(x3[:, :, D:2 * D]), hd(x3[:, :, 2 * D:]), hd(dout32.view(B, S, D))
        P = torch.softmax(masked_logits, dim=-1)
        dP = torch.matmul(Vh, dOh.transpose(-1, -2))
        dlt = (P * dP).sum(-1, keepdim=True)
        A = P * (dP.abs() + dlt.abs()) / math.sqrt(dh)
        back = lambda t: t.permute(0, 2, 1, 3).reshape(B * S, D)
        out = {"dV": back(torch.matmul(P, dOh.abs())), "dK": back(torch.matmul(A, Qh.abs())),
               "dQ": back(torch.matmul(A.transpose(-1, -2), Kh.abs()))}
        if subnormal:
            ones = torch.ones_like(P)
            dv_abs = torch.matmul(ones, dOh.abs())                                  # [B,H,k,d]: sum_q |dO[q,d]| per absolute P error
            dlt_abs = (Vh.abs() * dv_abs).sum(-1, keepdim=True)                     # delta = V.dV
            A_abs = ones + P * dlt_abs / math.sqrt(dh)                              # dS term: its own rounding + P * scale * delta error
            out["dV"] = out["dV"] + subnormal * back(dv_abs)
            out["dK"] = out["dK"] + subnormal * back(torch.matmul(A_abs, Qh.abs()))
            out["dQ"] = out["dQ"] + subnormal * back(torch.matmul(A_abs.transpose(-1, -2), Kh.abs()))
        return out


def check_grid(logits_masked, keymask, dh):
    """what the case exercises, asserted so that a change of the generator cannot silently leave the regime"""
    pad = (keymask == 0)[:, None, :, None].expand_as(logits_masked)
    t = logits_masked[pad].double() - MASK_VALUE   # the grid offsets of the padded rows' logits: multiples of 64
    assert torch.all(t == torch.round(t / 64.0) * 64.0)
    offs = set(int(v) for v in torch.unique(t).tolist())
    assert {-128, -64, 0, 64, 128} <= offs, f"padded-key logits must land on several grid points, got {sorted(offs)}"
    return offs


CASES = [  # B, S, H, dh, path, q_limit
    (3, 256, 2, 16, "auto", 0),      # resident, whole tiles
    (3, 257, 2, 16, "auto", 0),      # resident + the lone 257th row (the decoder of configs[1]) with its thin tiles
    (3, 250, 2, 32, "auto", 0),      # resident, head size 32, ragged last tile
    (3, 250, 2, 32, "auto", 1),      # ... the top encoder layer's form: one query produced, dO zero beyond it
    (3, 100, 1, 64, "auto", 0),      # head size 64
    (2, 700, 2, 32, "auto", 0),      # forward: K and V staged over Q (two tiles); backward: streaming dV/dK + chunked dQ
    (1, 1025, 2, 16, "auto", 0),     # configs[4]'s decoder: two-tile forward with the lone row
    (1, 1024, 2, 32, "auto", 0),     # configs[4]'s encoder: one tile, chunked output phase; chunked dQ
    (3, 250, 2, 32, "stream", 0),    # the streaming kernels (statistics / output / dV, dK / dQ)
    (3, 257, 2, 16, "stream", 0),
    (3, 250, 2, 32, "stream", 1),
]


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
@pytest.mark.parametrize("B,S,H,dh,path,q_limit", CASES)
def test_padded_key_logits_beyond_32_land_on_the_reference_grid(gpu, monkeypatch, B, S, H, dh, path, q_limit, dtype):
    if path == "stream":
        monkeypatch.setenv("MST_ATTN_PATH", "stream")
    o = _ops()
    D = H * dh
    qkv32, mask = integer_case(B, S, H, dh, seed=100 + S + dh)
    qkv_ref = qkv32.clone().requires_grad_(True)
    ref, logits, mass = reference(qkv_ref, mask, B, S, H, dh)
    check_grid(logits.detach(), mask, dh)
    if dh == 32:  # the scale is not a power of two: x / sqrt(dh) is rounded before the add; stay clear of the boundaries
        x = torch.matmul(qkv32.view(B, S, 3 * D)[:, :, :D].reshape(B, S, H, dh).permute(0, 2, 1, 3),
                         qkv32.view(B, S, 3 * D)[:, :, D:2 * D].reshape(B, S, H, dh).permute(0, 2, 1, 3).transpose(-1, -2)).double() / math.sqrt(dh)
        dist = ((x.abs() - 32.0) / 64.0 - torch.round((x.abs() - 32.0) / 64.0)).abs() * 64.0
        assert dist[(mask == 0)[:, None, :, None].expand_as(dist)].min().item() > 1e-3

    qkv = qkv32.to(dtype).to(gpu)
    assert torch.equal(qkv.float().cpu(), qkv32), "the integers must survive the 16-bit type"
    keymask = mask.to(gpu)
    lse = torch.zeros(2, B, H, S, dtype=torch.float32, device=gpu)
    out = torch.zeros(B * S, D, dtype=dtype, device=gpu)
    o.attn_fwd(qkv, keymask, lse, out, B, S, H, dh, 0, D, 2 * D, q_limit=q_limit)
    torch.cuda.synchronize()

    # the statistics: on a padded row the row maximum IS the grid point — compared EXACTLY; log-sum-exp of every row to fp32 rounding
    lg = logits.detach()
    rmax = lg.max(dim=-1).values
    padded = (mask == 0)[:, None, :].expand(B, H, S)
    got_max = lse[0].cpu()
    assert torch.equal(got_max[padded], rmax[padded]), f"grid points differ on {(got_max[padded] != rmax[padded]).sum().item()} padded key rows"
    logl = torch.log(torch.exp(lg - rmax[..., None]).sum(-1))
    assert (lse[1].cpu()[padded] - logl[padded]).abs().max().item() <= 2e-5 * max(1.0, logl.abs().max().item())
    valid = ~padded
    d = ((lse[0] + lse[1]).cpu() - (rmax + logl))[valid].abs()
    assert (d <= 1e-5 * (rmax + logl)[valid].abs() + 2e-5).all(), d.max().item()

    ulp = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    rows = slice(None) if q_limit == 0 else slice(0, None, S)
    got, want = out.float().cpu()[rows], ref.detach()[rows]
    assert torch.isfinite(got).all()
    # P is rounded to 16 bits before the P^T V product (half an ulp per term: against sum_k P |V|), the result once more
    err = (got - want).abs()
    tol = 2 * ulp * mass[rows] + 2 * ulp * want.abs() + 1e-6
    assert (err <= tol).all(), f"attention output: {(err > tol).sum().item()} elements, max err {err.max().item():.4g} (ref max {want.abs().max().item():.4g})"

    dout32 = torch.from_numpy(np.random.default_rng(7).integers(-3, 4, size=(B * S, D)).astype(np.float32))
    if q_limit:
        keep = torch.zeros(B, S, 1)
        keep[:, :q_limit] = 1
        dout32 = (dout32.view(B, S, D) * keep).reshape(B * S, D)
    dout = dout32.to(dtype).to(gpu)
    dqkv = torch.zeros(B * S, 3 * D, dtype=dtype, device=gpu)
    delta = torch.zeros(B, H, S, dtype=torch.float32, device=gpu)
    o.attn_bwd(qkv, keymask, lse, dout, dqkv, delta, B, S, H, dh, 0, D, 2 * D, q_limit=q_limit)
    torch.cuda.synchronize()
    ref.backward(dout32)
    g = qkv_ref.grad
    dq = dqkv.float().cpu()
    # Error model: the kernels feed P and dS = P (dP - delta) to the second MFMA rounded to 16 bits (half an ulp per term) and
    # delta = V.dV carries dV's rounding, so an element of dK is off by at most ~ulp * scale * sum_q P (|dP| + |delta|) |Q| — the
    # bound is evaluated from the reference's own P per element (a scale-relative bound would be 100x looser on the columns
    # that hold the small integers and too tight on the column that holds the +-96 levels).
    bound = rounding_bounds(qkv32, dout32, lg, B, S, H, dh, subnormal=(2.0 ** -24 / ulp if dtype == torch.float16 else 0.0))
    for nm, off in (("dV", 2 * D), ("dK", 0), ("dQ", D)):
        a, b = dq[:, off:off + D], g[:, off:off + D]
        assert torch.isfinite(a).all() and b.abs().max().item() > 0
        err = (a - b).abs()
        tol = 3 * ulp * bound[nm] + 2 * ulp * b.abs() + 1e-6
        worst = (err / tol).max().item()
        assert worst <= 1.0, f"{nm}: {(err > tol).sum().item()} elements beyond the rounding model, worst {worst:.2f}x (max err {err.max().item():.4g}, scale {b.abs().max().item():.4g})"


def test_the_flip_case_discriminates_a_uniform_padded_row():
    """What the GPU test above would see if a kernel took the fused exp2 form (padded row = uniform 1/S) on a padded key whose
    logits leave the -1e9 grid point: the CPU restatement with such rows made uniform is outside the GPU test's tolerances on a
    large share of the elements, forward and in the logit-only gradient dK — the case separates the two forms, it does not
    tolerate either. (CPU only: runs in the `not gpu` suite.)"""
    B, S, H, dh = 3, 256, 2, 16
    D = H * dh
    qkv, mask = integer_case(B, S, H, dh, seed=100 + S + dh)
    q1 = qkv.clone().requires_grad_(True)
    ref, logits, mass = reference(q1, mask, B, S, H, dh)
    q2 = qkv.clone().requires_grad_(True)
    x = q2.view(B, S, 3 * D)
    heads = lambda off: x[:, :, off:off + D].reshape(B, S, H, dh).permute(0, 2, 1, 3)
    K, Q, V = heads(0), heads(D), heads(2 * D)
    P = torch.softmax(torch.matmul(K, Q.transpose(-1, -2)) / 4.0, dim=-1)
    P = torch.where((mask == 0)[:, None, :, None], torch.full_like(P, 1.0 / S), P)   # the fused form's padded row
    uni = torch.matmul(P.transpose(-1, -2), V).permute(0, 2, 1, 3).reshape(B * S, D)
    ulp = 2.0 ** -8
    err = (uni - ref).abs().detach()
    tol = 2 * ulp * mass + 2 * ulp * ref.detach().abs() + 1e-6
    assert (err > 10 * tol).float().mean().item() > 0.2, "forward"
    dout = torch.from_numpy(np.random.default_rng(7).integers(-3, 4, size=(B * S, D)).astype(np.float32))
    ref.backward(dout)
    uni.backward(dout)
    a, b = q2.grad[:, :D], q1.grad[:, :D]
    rows = (mask == 0).reshape(B * S)
    tol = 3 * ulp * rounding_bounds(qkv, dout, logits.detach(), B, S, H, dh)["dK"] + 2 * ulp * b.abs() + 1e-6
    assert ((a - b).abs()[rows] > 10 * tol[rows]).float().mean().item() > 0.2, "dK of the padded keys"
