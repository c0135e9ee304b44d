"""End-to-end parity of the HIP training step (musicstyletransfer_amd.engine through the C-ABI)
against the CPU oracle (oracle/vae_oracle.py) on identical weights, inputs and eps, dropout 0.

Tolerances (16-bit activations, fp32 accumulation, fp32 master weights and latent block):
  reconstruction loss batch mean: <= 1e-3 relative in both 16-bit modes (measured ~1e-5)
  ELBO (total_loss) and kl_loss batch means: <= 1e-3 relative (BASELINE.json north_star) at BASELINE
     configs[1] in bf16 and fp16, and on every small config in fp16; 2e-3 on the small bf16 configs
     (B <= 6 samples to average the bf16 noise of mu / sigma over)
  mu / sigma themselves: rms error <= 1.2e-2 (bf16), 3e-3 (fp16); |mu|,|sigma| are O(1)
  probs: mean abs error <= 2e-3 and <= 0.2 % of elements off by more than 2e-2
  gradients: cosine per tensor >= 0.96-0.98 (bf16) / 0.985 (fp16), global >= 0.985-0.995 / 0.998;
     Adam-updated weights agree to lr/4 on the elements whose gradient sign is not in the noise
These hold for weights whose sigma output stays away from 0: the KL term 0.5*sum(sigma^2 + mu^2 - 1 -
log sigma^2) (loss.py:9) has no epsilon and sigma is a raw linear output (model.py:100-103) that straddles 0
at Xavier init, where loss and gradient are singular — rounding ONLY the weights to bf16 then already moves
the batch-mean KL by up to 9e-4 ("Precision"). The comparisons therefore damp and bias the
sigma rows of latent_proj (_setup); test_raw_xavier_init_is_loosely_matched covers the plain init.
"""
import math
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


def _setup(kind, dims, B, T, seed, ragged=True, sigma_bias=1.5, batch=None, **hyper):
    from oracle import vae_oracle as O
    from musicstyletransfer_amd import engine as E
    rng = np.random.default_rng(seed)
    ocfg = O.OracleConfig(kind, *dims)
    params = O.init_params(ocfg, rng)
    # make biases / LayerNorm parameters non-trivial so their gradients and uses are exercised
    for k, v in params.items():
        if k.endswith("bias") or k.endswith("beta"):
            params[k] = (0.05 * rng.standard_normal(v.shape)).astype(np.float32)
        if k.endswith("gamma"):
            params[k] = (1.0 + 0.1 * rng.standard_normal(v.shape)).astype(np.float32)
    # Conditioning: sigma is a raw linear output (model.py:100-103) and KL has log(sigma^2) with no epsilon
    # (loss.py:9). At plain Xavier init sigma straddles 0, where the loss and its gradient are singular and
    # NO two evaluation orders agree to 1e-3. Shrinking the sigma rows of latent_proj by 4 and adding a
    # bias of +1.5 puts sigma in roughly (0.7, 2.3), the regime KL training drives it to (sigma -> 1),
    # so tight tolerances test the kernels, not the singularity. sigma_bias=0 leaves the raw init
    # (test_raw_xavier_init_is_loosely_matched).
    Z = dims[3]
    if sigma_bias:
        params["encoder.latent_proj.weight"][Z:] *= 0.25
        params["encoder.latent_proj.bias"][Z:] += sigma_bias
    if batch is not None:
        pass  # the caller's own batch (real MIDI chunks: test_token_path_script_widths_on_real_midi_chunks)
    elif kind == "token":
        V = dims[0]
        lens = rng.integers(max(2, T // 2), T + 1, size=B) if ragged else np.full(B, T)
        x = np.zeros((B, T), np.int64)
        labels = np.zeros((B, T), np.int64)
        for b in range(B):
            n = int(lens[b])
            data = rng.integers(3, V, size=n - 1)
            x[b, 0] = 1
            x[b, 1:n] = data
            labels[b, : n - 1] = data
            labels[b, n - 1] = 2
        batch = {"x": torch.from_numpy(x), "seq_lens": torch.from_numpy(lens.astype(np.int64)),
                 "classes": torch.from_numpy(rng.integers(0, dims[2], size=B).astype(np.int64)),
                 "labels": torch.from_numpy(labels)}
    else:
        batch = O.synthetic_pianoroll_batch(rng, B, T, dims[0], num_classes=dims[2], density=0.04, ragged=ragged)
    eps = rng.standard_normal((B, dims[3])).astype(np.float32)
    ecfg = E.VAEConfig(kind, *dims)
    return O, E, ocfg, ecfg, params, batch, eps


def _cos(a, b):
    a, b = a.reshape(-1).astype(np.float64), b.reshape(-1).astype(np.float64)
    na, nb = np.linalg.norm(a), np.linalg.norm(b)
    if na == 0 and nb == 0:
        return 1.0
    return float(a @ b / (na * nb + 1e-300))


def _compare_step(gpu, kind, dims, B, T, seed, steps=2, lr=1e-3, dtype=torch.bfloat16, elbo_tol=1e-3, sigma_bias=1.5, ragged=True,
                  grad_cos=None, check_grads=True, consumed_weights=False, max_err=None, batch=None, **hyper):  # noqa: C901
    """Run `steps` training steps on the oracle and on the HIP engine and collect every out-of-tolerance
    quantity (one assertion at the end lists them all). Before every step the oracle's parameters and
    Adam state are overwritten with the engine's, so each step is compared from an IDENTICAL state at a
    new point of the trajectory: Adam's update is ~lr*sign(g) per element, and two trajectories whose
    gradients differ in the sign of a few near-zero elements drift apart at O(lr) per step, which says
    nothing about the kernels.
    consumed_weights: the oracle is given the weights AS THE KERNELS READ THEM (ParamStore.as_consumed_numpy: the 16-bit
    shadow of every GEMM weight, fp32 for the rest), which takes weight rounding out of the comparison and leaves the
    kernels' own error (16-bit activations, accumulation order)."""
    O, E, ocfg, ecfg, params, batch, eps = _setup(kind, dims, B, T, seed, sigma_bias=sigma_bias, ragged=ragged, batch=batch)
    lat_rms = 1.2e-2 if dtype == torch.bfloat16 else 3e-3
    small = B * T < 4096  # few rows to average 16-bit rounding noise over
    bf = dtype == torch.bfloat16
    if grad_cos is None:
        grad_cos = (0.96 if small else 0.98) if bf else 0.985
    global_cos = (0.985 if small else 0.995) if bf else 0.998
    # (fp16 on a few hundred rows: the sparse-frame embedding gradients sum a handful of rows per element; measured with the
    # configuration of test_width_128_multi_layer_* on seeds 14..16, 8-25 % of the largest element either way the FFN is launched)
    if max_err is None:
        max_err = (0.6 if small else 0.15) if bf else (0.3 if small else 0.1)
    if small and bf and elbo_tol == 1e-3:
        elbo_tol = 2e-3  # a handful of samples to average the bf16 noise of mu / sigma over
    # Gradients that exist only through the attention logits (W_k, W_q, and the decoder's position-0 inputs
    # latent2hid / class2hid, whose row is dropped before the loss, model.py:253) are P*(dP - delta): a
    # difference that nearly cancels while the softmax is close to uniform, so 16-bit rounding of dO and V
    # is amplified there. The attention kernels themselves are exact on identical 16-bit inputs
    # (test_kernels_gpu.py::test_attention_fwd_bwd); here these tensors get a direction check only.
    def noisy(name):
        return (".att.W_k." in name or ".att.W_q." in name or name.startswith("decoder.latent2hid")
                or name == "decoder.class2hid.weight")
    noisy_cos = 0.6 if bf else (0.9 if small else 0.95)
    ot = O.OracleTrainer(ocfg, params, lr=lr, clip_gradient=1.0, kl_weight=hyper.get("kl_weight", 1.0),
                         label_smoothing=hyper.get("label_smoothing", 0.0),
                         negative_label_downscaling=hyper.get("negative_label_downscaling", False))
    store = E.ParamStore(ecfg, gpu, dtype, params_np=params)
    plan = E.StepPlan(store, B, T, lr=lr, clip_gradient=1.0, want_probs=True, **hyper)
    plan.load_batch(batch["x"], batch["seq_lens"], batch["classes"], batch["labels"], eps)
    bad = []
    rel = lambda a, b: abs(float(a) - float(b)) / max(abs(float(b)), 1e-30)
    for s in range(steps):
        w_before = store.as_consumed_numpy() if consumed_weights else store.to_numpy("w")
        ot.load_state(w_before, store.to_numpy("m"), store.to_numpy("v"), int(store.step_state[0].item()))
        ref = ot.step(batch, torch.from_numpy(eps))
        plan.step_kernels(True)
        torch.cuda.synchronize()
        tot, kl, rec = plan.total.cpu().numpy(), plan.kl.cpu().numpy(), plan.recon.cpu().numpy()
        rt, rk, rr = ref["loss"].numpy(), ref["kl"].numpy(), ref["recon"].numpy()
        for nm, a, b, tol in (("recon", rec, rr, 1e-3), ("ELBO", tot, rt, elbo_tol), ("KL", kl, rk, elbo_tol)):
            if not rel(a.mean(), b.mean()) <= tol:
                bad.append(f"step {s} {nm} mean {a.mean():.6f} vs {b.mean():.6f} (rel {rel(a.mean(), b.mean()):.2e} > {tol:g})")
        for got, want, nm in ((plan.mu, ref["means"], "mu"), (plan.sigma, ref["stds"], "sigma")):
            d = got.cpu().numpy() - want.numpy()
            rms = float(np.sqrt((d ** 2).mean()))
            if not (rms <= lat_rms and np.abs(d).max() <= 8 * lat_rms):
                bad.append(f"step {s} {nm}: rms err {rms:.3g} max {np.abs(d).max():.3g}")
        V = dims[1]
        probs = plan.probs.float().cpu().numpy()[:, :V].reshape(B, T, V)
        perr = np.abs(probs - ref["probs"].numpy())
        # Padded key rows add -1e9 in fp32 (transformer.py:111-125): a logit with |x| >= 32 lands on a
        # different multiple of 64 there and flips that row's softmax. Those elements are chaotic in the
        # reference itself (not reproducible by any other evaluation order), so the bound is on the bulk.
        if not perr.mean() <= 2e-3:
            bad.append(f"step {s} probs mean abs err {perr.mean():.3g}")
        if not (perr > 2e-2).mean() <= (1e-2 if small and bf else 2e-3):
            bad.append(f"step {s} probs: {(perr > 2e-2).mean():.3g} of elements off by > 2e-2 (max {perr.max():.3g})")
        if check_grads:
            g = store.to_numpy("g")
            gmax = max(float(np.abs(r.numpy()).max()) for r in ref["grads"].values())
            # Padded-key logits with |x| >= 32 (counted by the oracle, vae_oracle.attention: FLIP_PRONE): fl(x - 1e9) leaves the
            # -1e9 grid point there and WHICH point it lands on depends on the last bits of x, so the logit-only gradients of
            # two evaluations that differ by one rounding differ in scale (tools/experiments/diag_noisy_grads.py: the oracle's
            # own |dL/d latent2hid| 8.7 vs 15.2). Instead of dropping the norm check there (round 3), the oracle is evaluated a
            # second time from the same state ON THE ENGINE'S 16-bit K | Q (straight-through: vae_oracle.QK_OVERRIDE), so both
            # sides put every padded-key logit on the same grid point, and the logit-only gradients of that side are compared —
            # direction and norm — against that evaluation. The engine's K | Q themselves are checked against the oracle's own.
            flip_sides = sorted({k.split(".", 1)[0] for k, n in ref.get("flip_prone", {}).items() if n > 0})
            ref_sync = None
            if flip_sides:
                ov = {}
                for side, layers_, S_, D_ in (("encoder", plan.enc, plan.T, ecfg.e_model), ("decoder", plan.dec, plan.T + 1, ecfg.d_model)):
                    for i, L in enumerate(layers_):
                        q3 = L.qkv.float().cpu().view(B, S_, -1)
                        ov[f"{side}.layer{i}.att"] = (q3[:, :, :D_].contiguous(), q3[:, :, D_:2 * D_].contiguous())
                ot2 = O.OracleTrainer(ocfg, w_before, lr=lr, clip_gradient=1.0, kl_weight=hyper.get("kl_weight", 1.0),
                                      label_smoothing=hyper.get("label_smoothing", 0.0),
                                      negative_label_downscaling=hyper.get("negative_label_downscaling", False))
                ref_sync = ot2.step(batch, torch.from_numpy(eps), qk_override=ov)
                for pre, (k_o, q_o) in ref_sync["qk_seen"].items():
                    for nm, mine, theirs in (("K", ov[pre][0], k_o), ("Q", ov[pre][1], q_o)):
                        e = float((mine - theirs).norm() / theirs.norm().clamp(min=1e-30))
                        if not e <= (2.5e-2 if bf else 6e-3):
                            bad.append(f"step {s} {pre} {nm} projection: relative rms error {e:.3g} against the oracle's own")
            num = den_a = den_b = 0.0
            for name, rg in ref["grads"].items():
                rg = rg.numpy()
                gg = g[name] / (plan.gscale_enc if name.startswith("encoder.") else plan.gscale)
                # W_q.bias has an analytically zero gradient (a constant along the softmax axis): compare
                # tensors whose reference gradient is below 1e-4 of the largest one on absolute error only
                side = name.split(".", 1)[0]
                synced = ref_sync is not None and noisy(name) and side in flip_sides
                if synced:  # (see above: the evaluation whose padded-key logits sit on the engine's grid points)
                    rg_free = rg
                    rg = ref_sync["grads"][name].numpy()
                    if os.environ.get("MST_TEST_NOTES"):  # what the synchronisation changes, for the record (profiles/)
                        n_e, n_f, n_s = (float(np.linalg.norm(t.astype(np.float64))) for t in (gg, rg_free, rg))
                        with open(os.environ["MST_TEST_NOTES"], "a") as fh:
                            fh.write(f"{dtype} step {s} {name}: flip-prone logits {ref['flip_prone']} |g| engine {n_e:.4g} oracle {n_f:.4g} "
                                     f"(ratio {n_e / max(n_f, 1e-300):.3f}, cos {_cos(gg, rg_free):.3f}) grid-synchronised oracle {n_s:.4g} "
                                     f"(ratio {n_e / max(n_s, 1e-300):.3f}, cos {_cos(gg, rg):.3f})\n")
                if np.abs(rg).max() > 1e-4 * gmax:
                    c = _cos(gg, rg)
                    # (grid-synchronised: measured cosine >= 0.995 / ratio within 4 % in bf16, 1.000 / 0.1 % in fp16 — profiles/r04_flip_sync_notes.txt)
                    if not c >= ((0.97 if bf else 0.99) if synced else noisy_cos if noisy(name) else grad_cos):
                        bad.append(f"step {s} gradient of {name}: cosine {c:.4f} (|ref| {np.abs(rg).max():.3g})")
                    # ... and a direction check alone would let a SCALE error through (a wrong 1/sqrt(dh), a dropped factor of the
                    # score scale in dK / dQ): the norm must agree too — to the part the cosine allows for noise, sqrt(1 - c^2) of it
                    ratio = float(np.linalg.norm(gg.astype(np.float64)) / max(np.linalg.norm(rg.astype(np.float64)), 1e-300))
                    slack = 0.10 if synced else 0.25 + (math.sqrt(max(0.0, 1.0 - min(c, 1.0) ** 2)) if noisy(name) else 0.0)
                    if not (1.0 / (1.0 + slack) <= ratio <= 1.0 + slack):
                        bad.append(f"step {s} gradient of {name}: norm ratio {ratio:.3f} (cosine {c:.3f}{', grid-synchronised oracle' if synced else ''})")
                    scale = float(np.abs(rg).max())
                    if not noisy(name) and not np.abs(gg - rg).max() <= max_err * scale:
                        bad.append(f"step {s} gradient of {name}: max err {np.abs(gg - rg).max():.3g} vs scale {scale:.3g}")
                elif not np.abs(gg - rg).max() <= 2e-3 * gmax:
                    bad.append(f"gradient of {name} (~0 in the reference): {np.abs(gg).max():.3g} vs global scale {gmax:.3g}")
                num += float((gg.astype(np.float64) * rg).sum())
                den_a += float((gg.astype(np.float64) ** 2).sum())
                den_b += float((rg.astype(np.float64) ** 2).sum())
            if not num / math.sqrt(den_a * den_b) >= global_cos:
                bad.append(f"step {s} global gradient cosine {num / math.sqrt(den_a * den_b):.5f}")
            # the Adam update itself, on the elements whose gradient sign is not in the noise: identical
            # state in, so the new weights must agree to a small fraction of the step
            w_after = store.to_numpy("w")
            for name, p in (ot.P.items() if not consumed_weights else ()):  # (the oracle stepped from the rounded weights)
                rg = ref["grads"][name].numpy()
                sure = np.abs(rg) > 0.5 * np.abs(rg).max()
                if sure.any() and np.abs(rg).max() > 1e-4 * gmax and not noisy(name):
                    d = np.abs(w_after[name] - p.detach().numpy())[sure].max()
                    if not d <= 0.25 * lr:
                        bad.append(f"step {s} {name}: updated weights differ by {d:.3g} (lr {lr:g}) on confident elements")
    m = plan.metrics()
    rm = ot.metrics()
    for k in ("total_loss", "kl_loss"):
        if not rel(m[k], rm[k]) <= elbo_tol:
            bad.append(f"metric {k}: {m[k]:.6f} vs {rm[k]:.6f}")
    assert not bad, "\n".join(bad)
    return plan, store, ot


def test_toy_config_token_path(gpu):
    """the reference's own --toy configuration and ToyData batch (main.py:14-38, data.py:62-70)"""
    from oracle import vae_oracle as O
    from musicstyletransfer_amd import engine as E
    rng = np.random.default_rng(7)
    ocfg = O.OracleConfig.toy()
    params = O.init_params(ocfg, rng)
    params["encoder.latent_proj.weight"][16:] *= 0.25  # keep sigma off the KL singularity (see _setup)
    params["encoder.latent_proj.bias"][16:] += 1.5
    batch = O.toy_batch()
    eps = rng.standard_normal((3, 16)).astype(np.float32)
    ot = O.OracleTrainer(ocfg, params, lr=1e-3, clip_gradient=1.0)
    store = E.ParamStore(E.VAEConfig("token", 10, 10, 3, 16, 32, 1, 2, 32, 1, 2), gpu, torch.bfloat16, params_np=params)
    plan = E.StepPlan(store, 3, 5, lr=1e-3, clip_gradient=1.0, want_probs=True)
    plan.load_batch(batch["x"], batch["seq_lens"], batch["classes"], batch["labels"], eps)
    for s in range(3):
        ref = ot.step(batch, torch.from_numpy(eps))
        plan.step_kernels(True)
        torch.cuda.synchronize()
        rt = ref["loss"].numpy()
        got = plan.total.cpu().numpy().mean()
        if s == 0:  # later steps are on separate trajectories (see _compare_step); they must stay close
            assert abs(got - rt.mean()) <= 2e-3 * abs(rt.mean()), f"step {s}: ELBO {got:.5f} vs {rt.mean():.5f}"
        assert abs(got - rt.mean()) <= 2e-2 * abs(rt.mean()), f"step {s}: ELBO {got:.5f} vs {rt.mean():.5f}"
        perr = np.abs(plan.probs.cpu().numpy().reshape(3, 5, 10) - ref["probs"].numpy())
        assert perr.max() <= 3e-2, f"step {s}: probs max err {perr.max():.3g}"


def test_free_running_training_tracks_the_oracle(gpu):
    """60 optimizer steps over a cycle of 4 piano-roll batches, the oracle and the engine each on its OWN trajectory from the same
    initial state (no resynchronisation, unlike _compare_step): the loss curves must stay together and must both go down. Per
    element the two Adam trajectories drift apart at O(lr) per step where a gradient sign is in the 16-bit noise (see
    _compare_step), so the bound is on the curves — 3 % of the ELBO at every step (measured 0.8 %, late in the run where the ELBO has
    fallen from 13.0 to 0.23), fp16 activations — not on the weights; a
    wrong bias correction, moment update, gradient scale or clipping threshold shows up here within a few steps."""
    from oracle import vae_oracle as O
    from musicstyletransfer_amd import engine as E
    dims, B, T, lr, steps = (40, 40, 2, 16, 64, 2, 2, 32, 1, 2), 6, 24, 2e-3, 60
    O_, E_, ocfg, ecfg, params, _, _ = _setup("pianoroll", dims, B, T, seed=31)
    rng = np.random.default_rng(32)
    batches = [O.synthetic_pianoroll_batch(rng, B, T, dims[0], num_classes=dims[2], density=0.06, ragged=True) for _ in range(4)]
    epss = [rng.standard_normal((B, dims[3])).astype(np.float32) for _ in range(4)]
    ot = O.OracleTrainer(ocfg, params, lr=lr, clip_gradient=1.0)
    store = E.ParamStore(ecfg, gpu, torch.float16, params_np=params)
    plan = E.StepPlan(store, B, T, lr=lr, clip_gradient=1.0)
    ref_curve, got_curve = [], []
    for s in range(steps):
        b, eps = batches[s % 4], epss[s % 4]
        ref_curve.append(float(ot.step(b, torch.from_numpy(eps))["loss"].mean()))
        plan.load_batch(b["x"], b["seq_lens"], b["classes"], b["labels"], eps)
        plan.step_kernels(True)
        got_curve.append(float(plan.total.float().mean().item()))
    ref_curve, got_curve = np.array(ref_curve), np.array(got_curve)
    rel = np.abs(got_curve - ref_curve) / np.abs(ref_curve)
    print(f"free-running: max rel gap {rel.max():.3g}, ELBO {ref_curve[0]:.4f} -> {ref_curve[-1]:.4f} (oracle), {got_curve[0]:.4f} -> {got_curve[-1]:.4f} (engine)")
    assert rel.max() <= 3e-2, f"largest relative gap {rel.max():.3g} at step {int(rel.argmax())}: {got_curve[rel.argmax()]:.5f} vs {ref_curve[rel.argmax()]:.5f}"
    first, last = ref_curve[:4].mean(), ref_curve[-4:].mean()
    assert last < 0.9 * first, f"the oracle's own loss did not go down ({first:.4f} -> {last:.4f}): the test would prove nothing"
    assert got_curve[-4:].mean() < 0.9 * got_curve[:4].mean()
    assert int(store.step_state[0].item()) == steps == ot.t


def test_token_path_ragged(gpu):
    # V=293 (NUM_EVENTS), script-like widths scaled down; ragged lengths exercise both padding masks
    _compare_step(gpu, "token", (293, 293, 2, 32, 64, 2, 4, 32, 1, 2), B=6, T=23, seed=11)


def midi_token_batch(batch_size=8, max_seq_len=64):
    """one batch of the reference's own training data path: tests/golden/midi (files of work/data/guitar_bass, byte for
    byte) -> Loader / EventBasedMIDIReader -> MelodyDataset's chunker (data.py:133-173, the EOS-column quirk of :168
    included) -> the batch protocol of data.py:187-198 (lengths inserted, truncated to the batch's longest row). Most
    chunks of a melody are full (65 positions), so instead of a batch of the shuffled epoch — one short row at best —
    the rows are chosen: the ragged tail chunks of the melodies (distinct lengths) plus full chunks, both classes."""
    from music_style_transfer.VarAutoEncoder import data as D
    midi = os.path.join(ROOT, "tests", "golden", "midi")
    ds, _ = D.load_dataset(D.Loader(midi, max_seq_len, 4), batch_size, 0.0)
    lens = D.count_sequence_length(ds.tokens)
    order = np.argsort(lens, kind="stable")
    short = [int(i) for i in order if lens[i] < max_seq_len + 1]
    picked, seen = [], set()
    for i in short:  # one chunk per distinct short length
        if int(lens[i]) not in seen and len(picked) < batch_size - 3:
            picked.append(i)
            seen.add(int(lens[i]))
    for c in (0, 1, 0, 1, 0, 1, 0, 1):  # fill up with full chunks, alternating classes
        if len(picked) == batch_size:
            break
        picked.append(next(int(i) for i in order[::-1] if ds.classes[i] == c and int(i) not in picked))
    idx = np.asarray(picked)
    b = D.preprocess_batch(D.Batch([ds.tokens[idx], ds.classes[idx]], [ds.labels[idx]], 0))
    tokens, lens_b, classes = (np.asarray(a) for a in b.data)
    return {"x": torch.from_numpy(tokens.astype(np.int64)), "seq_lens": torch.from_numpy(lens_b.astype(np.int64)),
            "classes": torch.from_numpy(classes.astype(np.int64)), "labels": torch.from_numpy(np.asarray(b.label[0]).astype(np.int64))}


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_token_path_script_widths_on_real_midi_chunks(gpu, dtype):
    """BASELINE configs[0] against the oracle: scripts/train-vae.sh's own widths (:6-29 — encoder 256 x 2 layers x 8 heads,
    decoder 128 x 1 x 8, latent 256, NUM_EVENTS = 293 tokens, 2 classes, batch 8, chunks of 64) on a batch of REAL MIDI
    chunks produced by the data path (data.py:133-198): ragged lengths, both padding masks, EOS written into the columns
    of every distinct length (data.py:168), SoftmaxCrossEntropy divided by the padded length (loss.py:16-23)."""
    batch = midi_token_batch()
    B, T = batch["x"].shape
    assert B == 8 and T <= 65 and int(batch["seq_lens"].max()) == T
    assert len(set(batch["seq_lens"].tolist())) >= 4 and set(batch["classes"].tolist()) == {0, 1}, "ragged rows of both classes"
    assert int((batch["labels"] == 2).sum()) >= B  # EOS columns (more than one per row is the reference's quirk)
    _compare_step(gpu, "token", (293, 293, 2, 256, 256, 2, 8, 128, 1, 8), B=B, T=T, seed=41, dtype=dtype, batch=batch, lr=3e-4)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_pianoroll_small(gpu, dtype):
    _compare_step(gpu, "pianoroll", (40, 40, 2, 16, 64, 2, 2, 32, 1, 2), B=5, T=19, seed=12, dtype=dtype)


def test_width_128_multi_layer_fused_layernorm_backward_chain(gpu):
    """encoder 3 x 128 and decoder 2 x 128: every LayerNorm backward below the top encoder layer rides on the GEMM that
    produces its input gradient (top layer -> layer 1 -> layer 0, output layer -> decoder layer 1 -> layer 0); B = 8
    also takes the XCD-aware attention workgroup order"""
    # fp16: five layers of bf16 rounding put the bulk probability error (2.3e-3) just over the 2e-3 bound of the 3-layer cases
    _compare_step(gpu, "pianoroll", (40, 40, 2, 16, 128, 3, 8, 128, 2, 8), B=8, T=31, seed=14, dtype=torch.float16)


def test_pianoroll_label_smoothing_downweighting_klweight(gpu):
    _compare_step(gpu, "pianoroll", (128, 128, 3, 32, 64, 1, 4, 64, 2, 4), B=4, T=33, seed=13, kl_weight=0.5,
                  label_smoothing=0.1, negative_label_downscaling=True)


def test_onehot_pianoroll_equals_token_encoder(gpu):
    """SURVEY §7 parity bridge: a one-hot frame times the table is the token gather, so the encoder
    (means, stddevs, KL) of the piano-roll ends reproduces the token ends exactly (same kernels
    downstream of the input projection, bf16-identical inputs up to one rounding)."""
    from oracle import vae_oracle as O
    from musicstyletransfer_amd import engine as E
    rng = np.random.default_rng(21)
    V, B, T = 24, 4, 12
    dims = (V, V, 2, 16, 32, 1, 2, 32, 1, 2)
    params = O.init_params(O.OracleConfig("token", *dims), rng)
    tokens = rng.integers(1, V, size=(B, T))
    lens = np.full(B, T)
    classes = rng.integers(0, 2, size=B)
    eps = rng.standard_normal((B, 16)).astype(np.float32)
    st_t = E.ParamStore(E.VAEConfig("token", *dims), gpu, torch.bfloat16, params_np=params)
    pl_t = E.StepPlan(st_t, B, T)
    pl_t.load_batch(tokens, lens, classes, np.zeros((B, T), np.int64), eps)
    pl_t.forward()
    st_p = E.ParamStore(E.VAEConfig("pianoroll", *dims), gpu, torch.bfloat16, params_np=params)
    pl_p = E.StepPlan(st_p, B, T)
    onehot = np.eye(V, dtype=np.uint8)[tokens]
    pl_p.load_batch(onehot, lens, classes, np.zeros((B, T, V), np.uint8), eps)
    pl_p.forward()
    torch.cuda.synchronize()
    # the table enters the GEMM rounded to bf16, the gather reads it in fp32: agreement to bf16 resolution
    np.testing.assert_allclose(pl_p.mu.cpu().numpy(), pl_t.mu.cpu().numpy(), rtol=0, atol=2e-2)
    np.testing.assert_allclose(pl_p.kl.cpu().numpy(), pl_t.kl.cpu().numpy(), rtol=2e-2)


def test_graph_replay_matches_eager(gpu):
    from musicstyletransfer_amd import engine as E
    O, E, ocfg, ecfg, params, batch, eps = _setup("pianoroll", (32, 32, 2, 16, 32, 1, 2, 32, 1, 2), 4, 16, 31)
    res = []
    for use_graph in (False, True):
        store = E.ParamStore(ecfg, gpu, torch.bfloat16, params_np=params)
        plan = E.StepPlan(store, 4, 16, lr=1e-3)
        plan.load_batch(batch["x"], batch["seq_lens"], batch["classes"], batch["labels"], eps)
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            plan.step_kernels(True)  # first step of a shape is always eager
            if use_graph:
                plan.capture(True)
            for _ in range(3):
                if use_graph:
                    plan.run()
                else:
                    plan.step_kernels(True)
        torch.cuda.synchronize()
        res.append((store.w.cpu().numpy().copy(), plan.total.cpu().numpy().copy(), int(store.step_state[0].item())))
    assert res[0][2] == res[1][2] == 4
    # fp32 atomics make weight gradients order-dependent in the last bits; everything else is deterministic
    np.testing.assert_allclose(res[0][1], res[1][1], rtol=1e-4)
    np.testing.assert_allclose(res[0][0], res[1][0], rtol=0, atol=2e-3)


def test_overlapped_data_parallel_schedule_matches_single_graph(gpu):
    """capture(split_optimizer, overlap): forward + early backward | late backward | optimizer, with the gradient
    bucket handed to the reducer in two ranges (top encoder layer .. decoder first). With a recording stand-in for the
    collective the three-graph schedule must reproduce the single-graph step."""
    from musicstyletransfer_amd import engine as E
    O, E, ocfg, ecfg, params, batch, eps = _setup("pianoroll", (32, 32, 2, 16, 32, 2, 2, 32, 1, 2), 4, 16, 37)

    class Recorder:
        def __init__(self):
            self.ranges = []

        def start(self, flat):
            self.ranges.append((flat.data_ptr(), flat.numel()))
            return None

        def finish(self, handles):
            assert all(h is None for h in handles)

    res = []
    for overlap in (False, True):
        store = E.ParamStore(ecfg, gpu, torch.bfloat16, params_np=params)
        plan = E.StepPlan(store, 4, 16, lr=1e-3)
        plan.load_batch(batch["x"], batch["seq_lens"], batch["classes"], batch["labels"], eps)
        rec = Recorder()
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            plan.step_kernels(True)
            plan.capture(True, split_optimizer=overlap, overlap=overlap)
            for _ in range(3):
                plan.run(reducer=rec if overlap else None)
        torch.cuda.synchronize()
        res.append((store.w.cpu().numpy().copy(), plan.total.cpu().numpy().copy(), int(store.step_state[0].item())))
        if overlap:
            cut, n = plan.grad_cut(), store.g.numel()
            assert 0 < cut < n and plan.graph_late is not None
            base = store.g.data_ptr()
            assert rec.ranges[:2] == [(base + 4 * cut, n - cut), (base, cut)]  # early range first, then the rest
    assert res[0][2] == res[1][2] == 4
    np.testing.assert_allclose(res[0][1], res[1][1], rtol=1e-4)
    np.testing.assert_allclose(res[0][0], res[1][0], rtol=0, atol=2e-3)


def test_dropout_masks_are_applied_and_reproducible(gpu):
    """with dropout on, forward/backward regenerate identical masks from the device-resident seed
    (loss decreases over steps; two plans with the same seed produce the same losses)"""
    from musicstyletransfer_amd import engine as E
    O, E, ocfg, ecfg, params, batch, eps = _setup("pianoroll", (32, 32, 2, 16, 32, 1, 2, 32, 1, 2), 4, 16, 41)
    ecfg.e_dropout = ecfg.d_dropout = 0.2
    outs = []
    for rep in range(2):
        store = E.ParamStore(ecfg, gpu, torch.bfloat16, params_np=params)
        plan = E.StepPlan(store, 4, 16, lr=1e-3, seed=5)
        plan.load_batch(batch["x"], batch["seq_lens"], batch["classes"], batch["labels"], eps)
        losses = []
        for _ in range(20):
            plan.step_kernels(True)
            losses.append(float(plan.total.mean().item()))
        outs.append(losses)
    assert np.allclose(outs[0], outs[1], rtol=1e-3)
    assert outs[0][-1] < outs[0][0]
    assert all(np.isfinite(outs[0]))


def test_raw_xavier_init_is_loosely_matched(gpu):
    """Plain Xavier init: sigma straddles 0, where KL = ... - log(sigma^2) and its gradient sigma - 1/sigma are
    singular (see _setup). A sigma of +1e-3 on one side and -1e-3 on the other flips the sign of the
    whole encoder gradient, so only forward quantities are compared here, and ELBO only to 1e-2."""
    _compare_step(gpu, "pianoroll", (40, 40, 2, 16, 64, 2, 2, 32, 1, 2), B=16, T=19, seed=14, sigma_bias=0.0, elbo_tol=1e-2,
                  steps=1, check_grads=False, ragged=False)


def test_full_size_configs1_elbo(gpu):
    """BASELINE.json configs[1]: single-track piano-roll T=256, pitch=128, latent=64, batch=64, bf16;
    widths from scripts/train-vae.sh (D_e 256 x 2 layers x 8 heads, D_d 128 x 1 layer x 8 heads)."""
    plan, store, ot = _compare_step(gpu, "pianoroll", (128, 128, 2, 64, 256, 2, 8, 128, 1, 8), B=64, T=256, seed=1234, steps=1,
                                    lr=3e-4)


def test_full_size_configs1_elbo_fp16(gpu):
    """same configuration on the fp16 MFMA path (loss-scaled gradients): ELBO within 1e-3 relative"""
    _compare_step(gpu, "pianoroll", (128, 128, 2, 64, 256, 2, 8, 128, 1, 8), B=64, T=256, seed=99, steps=1, lr=3e-4,
                  dtype=torch.float16)


def test_a_failed_position0_tail_is_flagged_skipped_and_replaced_by_the_five_launches(gpu):
    """VERDICT r02 item 6 / ADVICE: the one-launch position-0 tail (mst_row_tail_fwd) must not fail silently. Its roles are
    handed out behind its back in ONE step (sync[2] pre-set between the step's bookkeeping and the tail launch): the kernel
    flags the launch, the optimizer launch of that step leaves weights / moments / step count untouched and counts a skipped
    step, ParamStore.read_metrics() finds the flag, warns and pins the store to the five-launch form — and the run then
    continues to the same weights as a run that used the five launches for those steps in the first place."""
    import warnings
    from musicstyletransfer_amd import ops as o
    from test_parallel_gpu import _close_after_adam
    dims, B, T, lr = (48, 48, 2, 16, 256, 2, 8, 64, 1, 4), 6, 12, 1e-3
    O, E, ocfg, ecfg, params, _, _ = _setup("pianoroll", dims, B, T, seed=51)
    rng = np.random.default_rng(52)
    batches = [O.synthetic_pianoroll_batch(rng, B, T, dims[0], num_classes=2, density=0.08, ragged=True) for _ in range(3)]
    epss = [rng.standard_normal((B, dims[3])).astype(np.float32) for _ in range(3)]

    def step(plan, i):
        b = batches[i]
        plan.load_batch(b["x"], b["seq_lens"], b["classes"], b["labels"], epss[i])
        plan.step_kernels(True)

    # reference run: step 0 with the one-launch tails, steps 1 and 2 with the five launches
    ref_store = E.ParamStore(ecfg, gpu, torch.bfloat16, params_np=params)
    ref_plan = E.StepPlan(ref_store, B, T, lr=lr, clip_gradient=1.0)
    assert ref_store.tail_checked and ref_store.tail_fused, "the start-up self-check must pass on this device"
    step(ref_plan, 0)
    assert ref_plan._tail_used == dict(fwd=True, bwd=True)
    ref_store.tail_fused = False
    step(ref_plan, 1)
    assert ref_plan._tail_used == dict(fwd=False, bwd=False)
    step(ref_plan, 2)
    torch.cuda.synchronize()
    ref_m = ref_store.read_metrics()

    store = E.ParamStore(ecfg, gpu, torch.bfloat16, params_np=params)
    plan = E.StepPlan(store, B, T, lr=lr, clip_gradient=1.0)
    fired = []
    store.on_tail_failure(lambda: fired.append(1))
    step(plan, 0)
    torch.cuda.synchronize()
    w1, t1 = store.w.clone(), int(store.step_state[0].item())
    real, G = o.row_tail_fwd, dims[4] // 16

    def sabotaged(*a, **kw):
        plan.sync_words[2:3].fill_(G)  # every role is gone before the launch asks for one
        return real(*a, **kw)

    o.row_tail_fwd = sabotaged
    try:
        step(plan, 1)
    finally:
        o.row_tail_fwd = real
    torch.cuda.synchronize()
    assert torch.equal(store.w, w1) and int(store.step_state[0].item()) == t1, "the guarded optimizer must not touch the model"
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        m = store.read_metrics()
    assert any("position-0 tail" in str(c.message) for c in caught)
    assert fired == [1] and not store.tail_fused and m["skipped_steps"] == 1 and m["count"] == B  # only step 0 was counted
    flags, skipped = store.tail_failures[-1]
    assert skipped == 1 and flags == 16, flags   # MST_STEP_INCOMPLETE: the forward tail's barrier counter stayed short
    assert store.step_status.tolist() == [0, 0, 0]
    step(plan, 1)  # the batch again, now through the five launches
    assert plan._tail_used == dict(fwd=False, bwd=False)
    step(plan, 2)
    torch.cuda.synchronize()
    m2 = store.read_metrics()
    assert int(store.step_state[0].item()) == 3 and m2["count"] == 2 * B and m2["skipped_steps"] == 0
    assert abs(m["total_sum"] + m2["total_sum"] - ref_m["total_sum"]) <= 1e-4 * abs(ref_m["total_sum"])
    _close_after_adam(store.w.cpu().numpy(), ref_store.w.cpu().numpy(), lr, 3)
    # ParamStore.poll_status(): the non-blocking look the training loop takes every few steps (the words were
    # read only with the metrics, so one failed tail meant skipped batches until the next periodic log)
    store.tail_fused = True
    fired.clear()
    assert store.poll_status() is False          # starts an asynchronous copy of clean words
    torch.cuda.synchronize()
    assert store.poll_status() is False and fired == []
    torch.cuda.synchronize()
    store.step_status[0:1].fill_(2)              # a backward-tail barrier time-out, as the kernel leaves it
    store.step_status[1:2].fill_(3)
    store.poll_status()                          # (the copy in flight is older than the flag or not: at most one more poll)
    torch.cuda.synchronize()
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        handled = store.poll_status()
        if not handled:
            torch.cuda.synchronize()
            handled = store.poll_status()
    assert handled and fired == [1] and not store.tail_fused and store.tail_failures[-1] == (2, 3)
    assert store.step_status.tolist()[:2] == [0, 0]
    torch.cuda.synchronize()
    assert store.poll_status() is False          # handled once
    # data parallel: the words are summed over the ranks first and every rank stops
    torch.cuda.synchronize()
    store.poll_status(reduce=lambda t: t.mul_(2))
    store.step_status[0:1].fill_(1)
    torch.cuda.synchronize()
    store.poll_status(reduce=lambda t: t.mul_(2))
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match="at least one rank"):
        store.poll_status(reduce=lambda t: t.mul_(2))
    store.step_status[0:2].fill_(0)
    # MST_TAIL_FAILURE=raise: the same flag stops the run instead
    store.tail_policy, store.tail_fused = "raise", True
    store.step_status[0:1].fill_(1)
    with pytest.raises(RuntimeError, match="position-0 tail"):
        store.read_metrics()


def test_a_step_with_a_non_finite_loss_leaves_the_model_alone(gpu):
    """mst_step_metrics' non-finite guard: a poisoned activation (here: an infinite embedding weight, standing in for an fp16
    overflow at long sequences) makes the losses and every gradient NaN — the optimizer launch must not apply them. Parameters,
    moments and the step count stay as they were, the step is counted in the third status word, the metric sums do not take
    it, and the NEXT step (weight repaired) trains normally: the guard is not sticky."""
    import warnings
    O, E, ocfg, ecfg, params, batch, eps = _setup("pianoroll", (32, 32, 2, 16, 32, 1, 2, 32, 1, 2), 4, 16, 37)
    store = E.ParamStore(ecfg, gpu, torch.bfloat16, params_np=params)
    plan = E.StepPlan(store, 4, 16, lr=1e-3)
    plan.load_batch(batch["x"], batch["seq_lens"], batch["classes"], batch["labels"], eps)
    plan.step_kernels(True)
    torch.cuda.synchronize()
    assert plan.metrics()["count"] == 4 and int(store.step_state[0].item()) == 1
    emb = store.p("encoder.embedding.weight")
    keep = emb[0, 0].item()
    emb[0, 0] = float("inf")
    store.refresh_shadows()
    w_before, m_before = store.w.clone(), store.m.clone()
    plan.step_kernels(True)
    torch.cuda.synchronize()
    assert not torch.isfinite(plan.recon).all() or not torch.isfinite(plan.kl).all()
    assert torch.equal(store.w.nan_to_num(posinf=1e30), w_before.nan_to_num(posinf=1e30)) and torch.equal(store.m, m_before)
    assert int(store.step_state[0].item()) == 1
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        m = plan.metrics()
    assert m["nonfinite_steps"] == 1 and m["count"] == 0 and store.nonfinite_steps == 1
    assert any("not finite" in str(c.message) for c in caught)
    emb[0, 0] = keep
    store.refresh_shadows()
    plan.step_kernels(True)
    torch.cuda.synchronize()
    m = plan.metrics()
    assert m["nonfinite_steps"] == 0 and m["count"] == 4 and int(store.step_state[0].item()) == 2
    assert torch.isfinite(store.w).all() and not torch.equal(store.w, w_before)


def test_transposed_shadows_follow_the_weights_without_their_own_launch(gpu):
    """Piano-roll ends: the optimizer launch keeps the two embedding tables' transposed shadows current (mst_adam_flat_emb) and the NEXT
    step's first launch rebuilds the backward-only ones (mst_gemm_nt_pair_begin, sh_*): after the optimizer the embedding shadows
    equal the weights, after the next forward pass all of them do — and the run equals one with the separate refresh launch."""
    import os
    O, E, ocfg, ecfg, params, batch, eps = _setup("pianoroll", (48, 48, 2, 16, 64, 2, 2, 32, 1, 2), 4, 16, 61)

    def fresh(st, names):
        ok = True
        for n in names:
            so, r, c = st.t_specs[n]
            want = st.w[so: so + r * c].view(r, c).t().to(st.act_dtype)
            ok &= bool(torch.equal(st.t(n)[:, :r], want))
        return ok

    res = {}
    for ride in ("1", "0"):
        os.environ["MST_SHADOW_RIDE"] = ride
        try:
            store = E.ParamStore(ecfg, gpu, torch.bfloat16, params_np=params)
        finally:
            os.environ.pop("MST_SHADOW_RIDE", None)
        assert store.shadows_deferred == (ride == "1")
        plan = E.StepPlan(store, 4, 16, lr=1e-2)
        plan.load_batch(batch["x"], batch["seq_lens"], batch["classes"], batch["labels"], eps)
        for _ in range(3):
            plan.step_kernels(True)
        torch.cuda.synchronize()
        emb = ["encoder.embedding.weight", "decoder.embedding.weight"]
        late = [n for n in store.t_specs if n not in emb]
        assert fresh(store, emb)
        if ride == "1":
            assert not fresh(store, late), "the backward-only shadows are one optimizer step behind until the next forward pass"
            plan._tick_adam = False
            plan.forward()
            torch.cuda.synchronize()
        assert fresh(store, emb + late)
        res[ride] = store.w.cpu().numpy().copy()
    # (not bit for bit: the small configuration's LayerNorm parameter gradients and loss sums are fp32 atomics, order-dependent in
    # the last bits from run to run whatever the path)
    from test_parallel_gpu import _close_after_adam
    _close_after_adam(res["1"], res["0"], 1e-2, 3)


def test_shadow_refresh_rides_on_the_forward_tail(gpu):
    """Widths at which the position-0 tails have riders (decoder width 128, T a multiple of 128): the backward-only transposed shadows are
    rebuilt by the forward tail's riders behind their GEMM tiles (mst_row_tail_fwd_ride_shadows) instead of on the step's first launch —
    after a forward pass every shadow equals its weights, and training equals the run that keeps the refresh on the first launch."""
    import os
    # (full-length sequences: with padded keys this model sits in the reference's mask-flip regime — DESIGN section 4 —, where two runs one
    # rounding apart part ways by whole grid steps of the -1e9 mask: 3 of 24 identical runs ended 35 % of the weights away from the others,
    # whichever launch refreshed the shadows; tools/experiments/diag_sporadic.py, diag_shadow_grads.py)
    O, E, ocfg, ecfg, params, batch, eps = _setup("pianoroll", (64, 64, 2, 16, 128, 2, 4, 128, 1, 4), 4, 128, 67, ragged=False)

    def fresh(st, names):
        ok = True
        for n in names:
            so, r, c = st.t_specs[n]
            ok &= bool(torch.equal(st.t(n)[:, :r], st.w[so: so + r * c].view(r, c).t().to(st.act_dtype)))
        return ok

    res = {}
    for mode in ("1", "0"):
        os.environ["MST_SHADOW_TAIL"] = mode
        try:
            store = E.ParamStore(ecfg, gpu, torch.bfloat16, params_np=params)
            plan = E.StepPlan(store, 4, 128, lr=1e-2)
            plan.load_batch(batch["x"], batch["seq_lens"], batch["classes"], batch["labels"], eps)
            assert store.shadows_deferred and plan.ride
            for _ in range(3):
                plan.step_kernels(True)
            torch.cuda.synchronize()
            assert (plan._tail_shadows is not None) == (mode == "1")
            assert int(store.step_status.cpu()[0]) == 0
            emb = ["encoder.embedding.weight", "decoder.embedding.weight"]
            late = [n for n in store.t_specs if n not in emb]
            assert fresh(store, emb) and not fresh(store, late)
            plan._tick_adam = False
            plan.forward()
            torch.cuda.synchronize()
            assert fresh(store, emb + late)
        finally:
            os.environ.pop("MST_SHADOW_TAIL", None)
        res[mode] = store.w.cpu().numpy().copy()
    # (three Adam steps at lr 1e-2 from gradients that differ in the order of their fp32 atomics: equal except where a near-zero
    # gradient flips sign; a stale shadow would move every weight)
    d = np.abs(res["1"] - res["0"])
    assert d.max() <= 2.1 * 1e-2 * 3 and (d > 2e-5).mean() < 0.05, (d.max(), (d > 2e-5).mean())


def test_identical_runs_agree_on_full_length_batches(gpu):
    """A race guard: three eager Adam steps of a small model whose position-0 tails have riders (decoder K | Q | V projection, its input
    gradient, the shadow refresh), twice from the same state. The runs may differ in the order of their fp32 atomics only: every weight equal
    except where a near-zero gradient changes sign (measured: at most 2.6 % of the weights over 32 runs, `tools/experiments/diag_sporadic.py`).
    Full-length sequences: ragged ones put this model in the mask-flip regime, where identical runs part ways by whole grid steps (DESIGN 4 iii)."""
    O, E, ocfg, ecfg, params, batch, eps = _setup("pianoroll", (64, 64, 2, 16, 128, 2, 4, 128, 1, 4), 4, 128, 71, ragged=False)
    ws = []
    for _ in range(3):
        store = E.ParamStore(ecfg, gpu, torch.bfloat16, params_np=params)
        plan = E.StepPlan(store, 4, 128, lr=1e-2)
        plan.load_batch(batch["x"], batch["seq_lens"], batch["classes"], batch["labels"], eps)
        assert plan.ride
        for _ in range(3):
            plan.step_kernels(True)
        torch.cuda.synchronize()
        assert plan._ride_fwd and plan._tail_shadows is not None and int(store.step_status.cpu()[0]) == 0
        ws.append(store.w.cpu().numpy().copy())
    for w in ws[1:]:
        d = np.abs(w - ws[0])
        assert d.max() <= 2.1 * 1e-2 * 3 and (d > 2e-5).mean() < 0.05, (d.max(), (d > 2e-5).mean())
