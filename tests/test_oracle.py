"""The CPU oracle against (a) analytic known answers, (b) the reference's own literal data, (c) the committed
oracle fixtures (regression pin). CPU only."""
import json
import math
import os

import numpy as np
import pytest
import torch

from oracle import vae_oracle as O

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CONSTS = json.load(open(os.path.join(G, "reference_constants.json")))


def test_toy_batch_is_the_references_literal_data():
    b, t = O.toy_batch(), CONSTS["toy_data"]
    assert b["x"].tolist() == t["tokens"] and b["labels"].tolist() == t["labels"]
    assert b["seq_lens"].tolist() == t["seq_lens"] and b["classes"].tolist() == t["classes"]


def test_parameter_inventory_matches_survey():
    # toy 28 170, cfg1 2 093 349, cfg2 1 885 440, cfg3 2 993 664 parameters; 58 tensors
    n = lambda c: sum(int(np.prod(s)) for s in O.param_shapes(c).values())
    assert n(O.OracleConfig.toy()) == 28170
    assert n(O.OracleConfig("token", 293, 293, 2, 256, 256, 2, 8, 128, 1, 8)) == 2093349
    cfg2 = O.OracleConfig("pianoroll", 128, 128, 2, 64, 256, 2, 8, 128, 1, 8)
    assert n(cfg2) == 1885440 and len(O.param_shapes(cfg2)) == 58
    assert n(O.OracleConfig("pianoroll", 2048, 2048, 2, 256, 256, 2, 8, 128, 1, 8)) == 2993664


def test_xavier_bounds_and_trivial_params():
    cfg = O.OracleConfig.toy()
    p = O.init_params(cfg, np.random.default_rng(0))
    w = p["encoder.layer0.ff1.weight"]  # [128, 32]: U(+-sqrt(3 / ((32+128)/2)))
    bound = math.sqrt(3.0 / 80.0)
    assert np.abs(w).max() <= bound and np.abs(w).max() > 0.9 * bound
    assert (p["encoder.layer0.ff1.bias"] == 0).all() and (p["encoder.layer0.ln1.gamma"] == 1).all()


def test_positional_table_quirk():
    # transformer.py:205-210: exponent 2*i/D for EVERY column, sin on even, cos on odd
    pos = O.positional_encodings(8, 5)
    for t in range(5):
        for i in range(8):
            arg = t / (10000 ** (2.0 * i / 8))
            assert abs(pos[t, i] - (math.sin(arg) if i % 2 == 0 else math.cos(arg))) < 1e-12


def test_kl_known_answers():
    mu = torch.zeros(3, 4)
    sg = torch.tensor([[1.0] * 4, [-1.0] * 4, [2.0] * 4])
    kl = O.variational_kl(mu, sg)
    assert torch.allclose(kl, torch.tensor([0.0, 0.0, 4 * 0.5 * (3 - math.log(4.0))]), atol=1e-6)


def test_ce_of_uniform_probs_divides_by_padded_length():
    V, T = 10, 6
    probs = torch.full((2, T, V), 1.0 / V)
    labels = torch.tensor([[1, 2, 3, 0, 0, 0], [4, 5, 6, 7, 8, 0]])
    ce = O.softmax_cross_entropy(probs, labels)
    assert torch.allclose(ce, torch.tensor([math.log(V) * 3 / T, math.log(V) * 5 / T]), atol=1e-6)


def test_bce_known_answers_and_downweighting():
    pred = torch.zeros(2, 3, 4)  # logit 0 -> p = 0.5 -> bce = log 2 everywhere
    label = torch.zeros(2, 3, 4)
    label[0, 0, :2] = 1
    plain = O.binary_cross_entropy(pred, label, negative_label_downweighting=False)
    assert torch.allclose(plain, torch.full((2,), math.log(2.0)), atol=1e-6)
    dw = O.binary_cross_entropy(pred, label, negative_label_downweighting=True)
    # sample 0: 2 positives, 10 negatives: negatives become w*bce^2 with w = 2/10; sample 1: no positives -> w = 0
    l2 = math.log(2.0)
    assert abs(dw[0].item() - (2 * l2 + 10 * 0.2 * l2 * l2) / 12) < 1e-6
    assert abs(dw[1].item()) < 1e-9
    sm = O.binary_cross_entropy(pred, label, label_smoothing=0.3, negative_label_downweighting=False)
    assert torch.allclose(sm, plain, atol=1e-6)  # at p = 0.5 smoothing does not change the value


def test_padded_key_row_is_uniform_not_excluded():
    # -1e9 is added to a whole softmax row, so the row becomes uniform 1/S
    cfg = O.OracleConfig.toy()
    P = O.to_torch_params(O.init_params(cfg, np.random.default_rng(1)), requires_grad=False)
    x = torch.randn(1, 5, 32)
    valid = torch.tensor([[1.0, 1.0, 1.0, 0.0, 0.0]])
    _, probs = O.attention(P, "encoder.layer0.att", x, valid, 2, return_probs=True)
    assert torch.allclose(probs[0, :, 3, :], torch.full((2, 5), 0.2), atol=1e-6)
    assert torch.allclose(probs[0, :, 4, :], torch.full((2, 5), 0.2), atol=1e-6)
    assert torch.allclose(probs.sum(-1), torch.ones(1, 2, 5), atol=1e-5)  # rows (keys) sum to 1 over queries


def test_onehot_pianoroll_equals_token_path():
    rng = np.random.default_rng(3)
    V, B, T = 12, 2, 7
    dims = (V, V, 2, 8, 16, 1, 2, 16, 1, 2)
    params = O.init_params(O.OracleConfig("token", *dims), rng)
    P = O.to_torch_params(params, requires_grad=False)
    tok = torch.from_numpy(rng.integers(1, V, size=(B, T)))
    lens, cls = torch.full((B,), T), torch.tensor([0, 1])
    eps = torch.from_numpy(rng.standard_normal((B, 8)).astype(np.float32))
    _, m_t, s_t, logit_t = O.model_forward(P, O.OracleConfig("token", *dims), tok, lens, cls, eps)
    onehot = torch.nn.functional.one_hot(tok, V).to(torch.uint8)
    _, m_p, s_p, logit_p = O.model_forward(P, O.OracleConfig("pianoroll", *dims), onehot, lens, cls, eps)
    assert torch.allclose(m_t, m_p, atol=1e-5) and torch.allclose(s_t, s_p, atol=1e-5)
    assert torch.allclose(logit_t, logit_p, atol=1e-4)


def test_adam_is_mxnets_rule_not_torchs():
    w, g = torch.tensor([1.0, -2.0]), torch.tensor([10.0, -0.5])
    w1, m1, v1 = O.adam_update(w, g, torch.zeros(2), torch.zeros(2), 1, 0.1, rescale_grad=0.5, clip_gradient=1.0)
    gc = torch.tensor([1.0, -0.25])  # 10*0.5 clipped to 1; -0.25
    assert torch.allclose(m1, 0.1 * gc) and torch.allclose(v1, 0.001 * gc * gc)
    lr_t = 0.1 * math.sqrt(1 - 0.999) / (1 - 0.9)
    assert torch.allclose(w1, w - lr_t * m1 / (v1.sqrt() + 1e-8), atol=1e-7)


def test_oracle_reproduces_committed_toy_fixture():
    z = np.load(os.path.join(G, "oracle_toy.npz"))
    params = {k[2:]: z[k] for k in z.files if k.startswith("p_")}
    tr = O.OracleTrainer(O.OracleConfig.toy(), params, lr=1e-3, clip_gradient=1.0)
    r0 = tr.step(O.toy_batch(), torch.from_numpy(z["eps"]))
    r1 = tr.step(O.toy_batch(), torch.from_numpy(z["eps"]))
    np.testing.assert_allclose(r0["loss"].numpy(), z["loss0"], rtol=2e-5)
    np.testing.assert_allclose(r0["probs"].numpy(), z["probs0"], atol=2e-6)
    np.testing.assert_allclose(r0["grads"]["decoder.output_layer.weight"].numpy(), z["g_out"], rtol=1e-3, atol=1e-6)
    np.testing.assert_allclose(r1["loss"].numpy(), z["loss1"], rtol=1e-4)
    np.testing.assert_allclose(tr.P["decoder.output_layer.weight"].detach().numpy(), z["w_out_after2"], atol=1e-6)


def test_autograd_matches_finite_differences_fp64():
    """the backward the GPU kernels are checked against is itself checked here, in fp64, no padding"""
    rng = np.random.default_rng(5)
    cfg = O.OracleConfig("pianoroll", 6, 6, 2, 4, 8, 1, 2, 8, 1, 2)
    params = O.init_params(cfg, rng)
    params["encoder.latent_proj.bias"][4:] += 1.5
    batch = O.synthetic_pianoroll_batch(rng, 2, 5, 6, density=0.3)
    eps = torch.from_numpy(rng.standard_normal((2, 4))).double()
    P = O.to_torch_params(params, dtype=torch.float64)
    loss = O.step_losses(P, cfg, batch, eps)[0].sum()
    loss.backward()
    for name in ("encoder.layer0.att.W_q.weight", "decoder.layer0.ff1.weight", "encoder.latent_proj.bias"):
        p = P[name]
        idx = tuple(int(rng.integers(0, s)) for s in p.shape)
        h = 1e-6
        with torch.no_grad():
            old = p[idx].item()
            p[idx] = old + h
            lp = O.step_losses(P, cfg, batch, eps)[0].sum().item()
            p[idx] = old - h
            lm = O.step_losses(P, cfg, batch, eps)[0].sum().item()
            p[idx] = old
        fd = (lp - lm) / (2 * h)
        assert abs(fd - p.grad[idx].item()) <= 1e-5 * max(1.0, abs(fd)), (name, fd, p.grad[idx].item())
