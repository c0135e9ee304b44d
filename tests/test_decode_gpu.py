"""Incremental decoding with K | Q | V caches on a real GPU: the decode step against the oracle's
restatement of the reference's inference path (oracle.decode_incremental), and the two samplers built on it."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _model(kind, dims, gpu, seed=3):
    from oracle import vae_oracle as O
    from musicstyletransfer_amd import engine as E
    rng = np.random.default_rng(seed)
    ocfg = O.OracleConfig(kind, *dims)
    params = O.init_params(ocfg, rng)
    for k, v in params.items():  # non-trivial biases / LayerNorm parameters
        if k.endswith("bias") or k.endswith("beta"):
            params[k] = (0.05 * rng.standard_normal(v.shape)).astype(np.float32)
        if k.endswith("gamma"):
            params[k] = (1.0 + 0.1 * rng.standard_normal(v.shape)).astype(np.float32)
    store = E.ParamStore(E.VAEConfig(kind, *dims), gpu, torch.bfloat16, params_np=params)
    return O, ocfg, params, store, rng


@pytest.mark.parametrize("kind,dims", [("token", (40, 40, 2, 16, 64, 1, 2, 64, 2, 4)), ("pianoroll", (48, 48, 2, 16, 64, 1, 2, 32, 1, 2))])
@pytest.mark.parametrize("attention", ["query", "key"])
def test_decode_step_matches_the_oracle(gpu, kind, dims, attention):
    """teacher-forced: the same tokens / frames are fed to the GPU decode step and to oracle.decode_incremental, position by
    position; the cached K | Q | V rows make step t cost one row per sample whatever t is"""
    from musicstyletransfer_amd import decode
    O, ocfg, params, store, rng = _model(kind, dims, gpu)
    B, n, Z, Dd = 5, 9, dims[3], dims[7]
    z = rng.standard_normal((B, Z)).astype(np.float32)
    classes = rng.integers(0, 2, size=B)
    fed = rng.integers(1, dims[0], size=(B, n)) if kind == "token" else (rng.random((B, n, dims[0])) < 0.1).astype(np.uint8)
    P = O.to_torch_params(store.as_consumed_numpy(), requires_grad=False)
    want = O.decode_incremental(P, ocfg, torch.from_numpy(z), torch.from_numpy(classes), torch.from_numpy(fed), attention).numpy()
    # position 0 exactly as the training step builds decoder row 0
    pos = O.positional_encodings(Dd, 1)[0].astype(np.float32)
    init = z @ params["decoder.latent2hid.weight"].T + params["decoder.latent2hid.bias"] + params["decoder.class2hid.weight"][classes]
    row0 = torch.from_numpy((np.sqrt(Dd) * init + pos).astype(np.float32)).to(torch.bfloat16).to(gpu)
    plan = decode.DecodePlan(store, B, n + 1, attention=attention)
    plan.start(row0)
    got = []
    for t in range(n):
        got.append(plan.step(fed[:, t]).float().cpu().numpy())
    got = np.stack(got, 1)
    err = np.abs(got - want)
    assert err.mean() <= 3e-3 and err.max() <= 6e-2, (err.mean(), err.max())
    if kind == "token":
        np.testing.assert_allclose(got.sum(-1), 1.0, atol=1e-3)
    assert plan.t == n
    # every position from the second on was captured as a hipGraph at first use; a second pass over the same plan REPLAYS them
    # (position 0 and 1 are captured now) and must reproduce the first pass bit for bit
    assert plan.use_graphs and sorted(plan._graphs) == list(range(2, n + 1))
    plan.reset()
    plan.start(row0)
    again = np.stack([plan.step(fed[:, t]).float().cpu().numpy() for t in range(n)], 1)
    assert sorted(plan._graphs) == list(range(0, n + 1))
    assert np.array_equal(again, got)


def test_query_axis_softmax_of_one_query_sums_the_cached_values(gpu):
    """the reference's arithmetic on a decode step (transformer.py:96-102 with T_Q = 1): every weight is exactly 1"""
    o = __import__("musicstyletransfer_amd.ops", fromlist=["ops"])
    B, H, dh, n, t_max = 3, 2, 16, 7, 10
    D = H * dh
    g = torch.Generator().manual_seed(1)
    cache = torch.randn(B, t_max, 3 * D, generator=g).to(torch.bfloat16).to(gpu)
    out = torch.zeros(B, D, dtype=torch.bfloat16, device=gpu)
    o.attn_decode(cache, n, H, dh, 0, D, 2 * D, out, mode=0)
    torch.cuda.synchronize()
    want = cache[:, :n, 2 * D:].float().sum(1)
    assert torch.allclose(out.float(), want, rtol=1e-2, atol=1e-2)
    o.attn_decode(cache, n, H, dh, 0, D, 2 * D, out, mode=1)
    torch.cuda.synchronize()
    K, V = cache[:, :n, :D].float().view(B, n, H, dh), cache[:, :n, 2 * D:].float().view(B, n, H, dh)
    q = cache[:, n - 1, D:2 * D].float().view(B, H, dh)
    p = torch.softmax(torch.einsum("bkhd,bhd->bhk", K, q) / np.sqrt(dh), -1)
    want = torch.einsum("bhk,bkhd->bhd", p, V).reshape(B, D)
    assert torch.allclose(out.float(), want, rtol=1e-2, atol=1e-2)


def test_samplers_on_the_toy_model(gpu, tmp_path):
    """Sampling and BeamSearchSampler through the reference's interface: get_sampler -> update_parameters -> process_batch
    writes .mid files; beam search with one beam is greedy decoding, wider beams never score worse"""
    from music_style_transfer.VarAutoEncoder import main, sampler as S
    from music_style_transfer.VarAutoEncoder.data import ToyData
    from music_style_transfer.MIDIUtil.defaults import EOS_ID, PAD_ID

    class A:
        verbose, beam_size = False, 3

    t = main.main(["--toy", "--gpu", "--max-steps", "300", "--model-output", str(tmp_path)])
    t.stream.synchronize()
    batch = next(iter(ToyData()))
    smp = S.get_sampler("sampling", None, None, None, A)
    smp.update_parameters(t.model)
    seqs = smp.sample(batch)
    assert seqs.shape[0] == 3 and seqs.shape[1] >= 2 and (seqs[:, 0] == 1).all() and np.isfinite(smp.scores).all()
    assert ((seqs >= 0) & (seqs < 10)).all()
    # greedy reference through the same decode step
    dec = t.model.decoder
    st = dec.get_initial_state(*batch.data, t_max=12)
    greedy = []
    for _ in range(6):
        p = dec.forward_inference(st).float().cpu().numpy()
        nxt = p.argmax(-1)
        greedy.append(nxt)
        st.advance_state(nxt)
    greedy = np.stack(greedy, 1)
    b1 = S.BeamSearchSampler(beam_size=1)
    b1.update_parameters(t.model)
    one = b1.sample(batch)
    for b in range(3):  # identical until the hypothesis ends
        n_cmp = 0
        for i in range(min(6, one.shape[1] - 1)):
            if one[b, i + 1] in (EOS_ID, PAD_ID):
                break
            assert one[b, i + 1] == greedy[b, i]
            n_cmp += 1
    b3 = S.get_sampler("beam-search", None, None, None, A)
    b3.update_parameters(t.model)
    b3.sample(batch)
    assert b3.hypotheses.shape[:2] == (3, 3)
    assert (np.diff(b3.scores, axis=1) >= -1e-9).all()               # hypotheses come out best first
    assert (b3.scores[:, 0] <= b1.scores[:, 0] + 1e-6).all()           # a wider beam never does worse than greedy
    # beam search ranked on the DEVICE (decode.BeamSearch: mst_beam_step + mst_beam_gather inside every position's graph, the
    # default) against the host loop over the same decode step: the same hypotheses, best first, the same scores
    for K in (1, 3, 4):
        dev_s, host_s = S.BeamSearchSampler(beam_size=K, on_device=True), S.BeamSearchSampler(beam_size=K, on_device=False)
        for smp_k in (dev_s, host_s):
            smp_k.update_parameters(t.model)
            smp_k.sample(batch)
        assert dev_s.hypotheses.shape == host_s.hypotheses.shape  # both of width i_max (the device path returned last + 1 columns)
        assert 0 < dev_s.tokens_decoded <= dev_s.hypotheses.shape[0] * K * dev_s.positions_decoded
        n = min(dev_s.hypotheses.shape[2], host_s.hypotheses.shape[2])
        np.testing.assert_allclose(dev_s.scores, host_s.scores, rtol=2e-4, atol=2e-4)
        assert np.array_equal(dev_s.hypotheses[:, :, :n], host_s.hypotheses[:, :, :n]), K
        assert (dev_s.hypotheses[:, :, n:] == PAD_ID).all() and (host_s.hypotheses[:, :, n:] == PAD_ID).all()
    dev_s.sample(batch)  # a second batch of the same shape replays the captured positions
    assert len(t.model.beam_search_plan(3, 4, 10)._graphs) >= 1
    files = smp.process_batch(batch, str(tmp_path / "samples"), 3)
    assert len(files) == 3 + 3 * 3 and all(os.path.getsize(f) > 0 for f in files)
    with pytest.raises(ValueError):
        S.get_sampler("nucleus", None, None, None, A)


def test_model_call_and_sampler_state_run_in_inference_mode(gpu):
    """Model(...) called directly and Decoder.get_initial_state run OUTSIDE autograd.record() in the reference
    (sampler.py:146-148), where Dropout is the identity — with e_dropout / d_dropout 0.2 (scripts/train-vae.sh) they must be
    deterministic, agree with the dropout-free oracle, and leave the training RNG stream where it was."""
    from music_style_transfer.VarAutoEncoder import model
    from music_style_transfer.VarAutoEncoder.transformer import TransformerConfig
    from music_style_transfer.VarAutoEncoder.utils import gpu as gpu_ctx
    from oracle import vae_oracle as O
    dims = (40, 40, 2, 16, 64, 2, 2, 32, 1, 2)
    rng = np.random.default_rng(5)
    ocfg = O.OracleConfig("pianoroll", *dims)
    params = O.init_params(ocfg, rng)
    params["encoder.latent_proj.weight"][16:] *= 0.25
    params["encoder.latent_proj.bias"][16:] += 1.5
    cfg = model.ModelConfig(model.EncoderConfig(TransformerConfig(64, 0.2, 2, 2, 40), 16, 2, 40),
                            model.DecoderConfig(TransformerConfig(32, 0.2, 1, 2, 40), 16, 2, 40), kind="pianoroll")
    m = model.Model(cfg).initialize(gpu_ctx(0), params_np=params)
    assert m.engine_config.e_dropout == 0.2 and m.engine_config.d_dropout == 0.2
    B, T = 4, 10
    batch = O.synthetic_pianoroll_batch(rng, B, T, 40, num_classes=2, density=0.1, ragged=True)
    eps = rng.standard_normal((B, 16)).astype(np.float32)
    x, lens, cls = batch["x"].numpy(), batch["seq_lens"].numpy(), batch["classes"].numpy()
    rng_before = m.store.rng_state(0).clone()
    p1, mu1, sg1 = (t.clone() for t in m(x, lens, cls, eps=eps))
    p2, mu2, sg2 = m(x, lens, cls, eps=eps)
    torch.cuda.synchronize()
    assert torch.equal(p1, p2) and torch.equal(mu1, mu2) and torch.equal(sg1, sg2)  # no dropout noise between two calls
    P = O.to_torch_params(m.store.as_consumed_numpy(), requires_grad=False)
    probs, means, stds, _ = O.model_forward(P, ocfg, batch["x"], batch["seq_lens"], batch["classes"], torch.from_numpy(eps), None)
    assert np.abs(mu1.cpu().numpy() - means.numpy()).max() <= 5e-2 and np.abs(p1.cpu().numpy() - probs.numpy()).mean() <= 3e-3
    # eps not given: drawn from the inference stream
    m(x, lens, cls)
    st1 = m.decoder.get_initial_state(x, lens, cls, t_max=4)
    first = st1.initial_state.clone()
    st2 = m.decoder.get_initial_state(x, lens, cls, t_max=4)
    torch.cuda.synchronize()
    assert torch.equal(first, st2.initial_state)
    # decoder row 0 = sqrt(D) * (latent2hid(means) + class2hid) + pos[0] from the dropout-free means (model.py:229-232)
    init = means.numpy() @ params["decoder.latent2hid.weight"].T + params["decoder.latent2hid.bias"] + params["decoder.class2hid.weight"][cls]
    want = np.sqrt(32.0) * init + O.positional_encodings(32, 1)[0]
    assert np.abs(st1.initial_state.float().cpu().numpy() - want).max() <= 0.15
    assert torch.equal(m.store.rng_state(0), rng_before), "inference must not advance the training RNG stream"


def test_device_sampling_draws_from_the_distribution(gpu):
    """mst_sample_step: inverse-CDF draws by counter hash — frequencies follow the distribution (N = 16384 sequences sharing one
    unnormalised distribution, 6 sigma per token), the score picks up -log p of the drawn token, finished sequences continue with
    PAD at no cost, draws are a function of (seed, position, sequence) alone"""
    from musicstyletransfer_amd import ops as o
    from music_style_transfer.MIDIUtil.defaults import EOS_ID, PAD_ID
    N, V, L = 16384, 37, 6
    g = torch.Generator().manual_seed(2)
    p = torch.rand(V, generator=g) ** 3
    p[5] = 0.0                                                   # a token that must never be drawn
    probs = (3.0 * p).view(1, V).repeat(N, 1).contiguous().to(gpu)  # (unnormalised on purpose)
    pn = (p / p.sum()).numpy().astype(np.float64)
    seqs = torch.full((N, L), 7, dtype=torch.int32, device=gpu)
    seqs[: N // 8, 1] = EOS_ID                                   # these sequences ended at position 1
    scores = torch.zeros(N, device=gpu)
    word = torch.zeros(N, dtype=torch.int32, device=gpu)
    active = torch.zeros(L + 1, dtype=torch.int32, device=gpu)
    o.sample_step(probs, seqs, scores, word, 2, 1234, EOS_ID, PAD_ID, active=active)
    torch.cuda.synchronize()
    tok = seqs[:, 2].cpu().numpy()
    assert (tok[: N // 8] == PAD_ID).all() and (scores[: N // 8] == 0).all()
    live = tok[N // 8:]
    n = len(live)
    counts = np.bincount(live, minlength=V).astype(np.float64)
    assert counts[5] == 0
    sigma = np.sqrt(n * pn * (1 - pn)) + 1.0
    assert (np.abs(counts - n * pn) <= 6 * sigma).all(), np.abs(counts - n * pn).max()
    np.testing.assert_allclose(scores[N // 8:].cpu().numpy(), -np.log(pn[live]), rtol=1e-4, atol=1e-5)
    assert np.array_equal(word.cpu().numpy(), tok)
    assert int(active[2].item()) == int(((live != EOS_ID) & (live != PAD_ID)).sum())
    # same (seed, position, sequence) -> same draw; another seed -> another sample
    seqs2 = seqs.clone(); scores2 = torch.zeros(N, device=gpu)
    o.sample_step(probs, seqs2, scores2, word, 2, 1234, EOS_ID, PAD_ID)
    seqs3 = seqs.clone()
    o.sample_step(probs, seqs3, scores2, word, 2, 99, EOS_ID, PAD_ID)
    torch.cuda.synchronize()
    assert torch.equal(seqs2[:, 2], seqs[:, 2]) and not torch.equal(seqs3[:, 2], seqs[:, 2])


def test_cache_gather_copies_the_reordered_rows_and_leaves_the_skipped_columns(gpu):
    """mst_beam_gather / mst_beam_gather_cols (sampler.py:236-238): out[j, :n] = in[src[j], :n]; with a skipped column range
    (the Q third of K | Q | V rows, which no later position reads) those columns of `out` keep what they held"""
    from musicstyletransfer_amd import ops as o
    N, t_max, D, n = 12, 9, 16, 6
    g = torch.Generator().manual_seed(3)
    cin = torch.randn(N, t_max, 3 * D, generator=g).to(torch.bfloat16).to(gpu)
    src = torch.randint(0, N, (N,), generator=g).to(torch.int32).to(gpu)
    full = torch.full_like(cin, 7.0)
    o.beam_gather(cin, full, src, n)
    part = torch.full_like(cin, 7.0)
    o.beam_gather(cin, part, src, n, skip_cols=(D, D))
    torch.cuda.synchronize()
    want = cin[src.long()]
    assert torch.equal(full[:, :n], want[:, :n]) and (full[:, n:] == 7.0).all()
    assert torch.equal(part[:, :n, :D], want[:, :n, :D]) and torch.equal(part[:, :n, 2 * D:], want[:, :n, 2 * D:])
    assert (part[:, :, D:2 * D] == 7.0).all() and (part[:, n:] == 7.0).all()
