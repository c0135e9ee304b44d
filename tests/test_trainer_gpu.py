"""The drop-in boundary on a real GPU: the reference's entry point, flags and classes driving the HIP step."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MIDI = os.path.join(G, "midi")
SCRIPT_FLAGS = ["--batch-size", "8", "--kl-loss", "1.0", "--validation-split", "0.0", "--max-seq-len", "64",
                "--slices-per-quarter-note", "4", "--sampling-frequency", "2000", "--checkpoint-frequency", "1000",
                "--num-checkpoints-not-improved", "32", "--epochs", "10000", "--optimizer", "adam", "--optimizer-params",
                "clip_gradient:1.0", "--learning-rate", "0.0003", "--label-smoothing", "0.0", "--e-n-layers", "2", "--e-dropout",
                "0.2", "--e-rnn-hidden-dim", "256", "--e-emb-hidden-dim", "256", "--latent-dim", "256", "--d-n-layers", "1",
                "--d-rnn-hidden-dim", "128", "--d-dropout", "0.2", "--gpu"]


def test_toy_entry_point_trains(gpu, tmp_path):
    """python -m music_style_transfer.VarAutoEncoder.main --toy (main.py:58-76): overfits ToyData"""
    from music_style_transfer.VarAutoEncoder import main
    t = main.main(["--toy", "--gpu", "--max-steps", "300", "--model-output", str(tmp_path)])
    first = None
    m = t.collect_metrics()
    assert np.isfinite(m["total_loss"]) and m["total_loss"] < 15.0, m  # starts near 20-25
    assert int(t.model.store.step_state[0].item()) == 300


def test_train_vae_flags_on_midi_subset_token_path(gpu, tmp_path):
    """BASELINE configs[0]: scripts/train-vae.sh's flags, 32-bar single-track MIDI subset, batch 8"""
    from music_style_transfer.VarAutoEncoder import main
    t = main.main(SCRIPT_FLAGS + ["--data", MIDI, "--model-output", str(tmp_path / "m"), "--out-samples", str(tmp_path / "s"),
                                  "--max-steps", "12"])
    m0 = t.collect_metrics()
    assert np.isfinite(m0["total_loss"]) and np.isfinite(m0["kl_loss"])
    assert t.train_state.n_batches == 12
    cfg = t.model.engine_config
    assert (cfg.kind, cfg.in_dim, cfg.e_model, cfg.e_layers, cfg.d_model, cfg.latent_dim) == ("token", 293, 256, 2, 128, 256)
    assert t.model.store.n_params == 2093349
    # 200 more steps on the same data: the ELBO comes down
    t.config.max_steps = 0
    losses = []
    ds = t._dataset if hasattr(t, "_dataset") else None
    from music_style_transfer.VarAutoEncoder import data as D
    train, _ = D.load_dataset(D.Loader(MIDI, 64, 4), 8, 0.0)
    for epoch in range(6):
        for b in train:
            t._step(b)
        losses.append(t.collect_metrics()["total_loss"])
    assert losses[-1] < losses[0], losses


def test_pianoroll_ends_through_the_same_entry_point(gpu, tmp_path):
    from music_style_transfer.VarAutoEncoder import main
    flags = [f for f in SCRIPT_FLAGS]
    flags[flags.index("--latent-dim") + 1] = "64"
    flags[flags.index("--max-seq-len") + 1] = "128"
    t = main.main(flags + ["--pianoroll", "--data", MIDI, "--model-output", str(tmp_path / "m"), "--out-samples", str(tmp_path / "s"),
                           "--max-steps", "10"])
    cfg = t.model.engine_config
    assert (cfg.kind, cfg.in_dim, cfg.out_dim, cfg.latent_dim) == ("pianoroll", 128, 128, 64)
    assert t.model.store.n_params == 1885440
    m = t.collect_metrics()
    assert np.isfinite(m["total_loss"])


def test_model_call_matches_committed_oracle_fixture(gpu):
    """Model(config)(frames, seq_lens, classes) -> (probs, means, vars) against tests/golden/oracle_pianoroll_small.npz"""
    from music_style_transfer.VarAutoEncoder import model
    from music_style_transfer.VarAutoEncoder.transformer import TransformerConfig
    from music_style_transfer.VarAutoEncoder.utils import gpu as gpu_ctx
    z = np.load(os.path.join(G, "oracle_pianoroll_small.npz"))
    params = {k[2:]: z[k] for k in z.files if k.startswith("p_")}
    cfg = model.ModelConfig(
        model.EncoderConfig(TransformerConfig(64, 0.0, 2, 2, 40), 16, 2, 40),
        model.DecoderConfig(TransformerConfig(32, 0.0, 1, 2, 40), 16, 2, 40), kind="pianoroll")
    m = model.Model(cfg).initialize(gpu_ctx(0), params_np=params)
    probs, means, stds = m(z["x"], z["seq_lens"], z["classes"], eps=z["eps"])
    torch.cuda.synchronize()
    assert np.sqrt(((means.cpu().numpy() - z["means"]) ** 2).mean()) < 1.2e-2
    assert np.sqrt(((stds.cpu().numpy() - z["stds"]) ** 2).mean()) < 1.2e-2
    perr = np.abs(probs.cpu().numpy() - z["probs"].astype(np.float32))
    assert perr.mean() < 2e-3 and (perr > 2e-2).mean() < 1e-2
    plan = m.plan(5, 19, want_probs=True, internal_eps=False)
    plan.load_batch(z["x"], z["seq_lens"], z["classes"], z["labels"], z["eps"])
    plan.forward()
    plan.losses(with_grad=False)
    torch.cuda.synchronize()
    assert abs(plan.recon.mean().item() - z["recon"].mean()) <= 1e-3 * z["recon"].mean()
    assert abs(plan.total.mean().item() - z["loss"].mean()) <= 2e-3 * z["loss"].mean()


def test_loss_classes_keep_the_reference_call_signature(gpu):
    from music_style_transfer.VarAutoEncoder import loss
    from oracle import vae_oracle as O
    g = torch.Generator().manual_seed(0)
    B, T, V = 3, 6, 12
    logits = torch.randn(B, T, V, generator=g).to(torch.bfloat16)
    labels = torch.randint(0, V, (B, T), generator=g)
    ce = loss.SoftmaxCrossEntropy(axis=-1, batch_axis=0)(logits.cuda(), labels.cuda())
    ref = O.softmax_cross_entropy(torch.softmax(logits.float(), -1), labels)
    assert torch.allclose(ce.cpu(), ref, rtol=1e-4, atol=1e-5)
    y = (torch.rand(B, T, V, generator=g) < 0.2).to(torch.uint8)
    bce = loss.BinaryCrossEntropy(from_sigmoid=False, label_smoothing=0.1, negative_label_downweighting=True)(logits.cuda(), y.cuda())
    refb = O.binary_cross_entropy(logits.float(), y, label_smoothing=0.1, negative_label_downweighting=True)
    assert torch.allclose(bce.cpu(), refb, rtol=1e-4, atol=1e-5)
    mu, sg = torch.randn(B, 8, generator=g), torch.randn(B, 8, generator=g) + 2
    kl = loss.VariationalKLLoss()(mu.cuda(), sg.cuda())
    assert torch.allclose(kl.cpu(), O.variational_kl(mu, sg), rtol=1e-5)


def test_checkpoint_and_resume(gpu, tmp_path):
    from music_style_transfer.VarAutoEncoder import main, utils
    folder = str(tmp_path / "toy")
    t = main.main(["--toy", "--gpu", "--max-steps", "20", "--model-output", folder])
    t._checkpoint(os.path.join(folder, "model"), None)
    w = t.model.store.w.clone()
    assert utils.get_latest_checkpoint_index(os.path.join(folder, "model")) == 1, os.listdir(os.path.join(folder, "model"))
    for i in range(2, 13):  # the reference's `(\d)+` regex would pick 2 after 12 checkpoints
        open(os.path.join(folder, "model", f"params.{i}.npz"), "wb").write(open(os.path.join(folder, "model", "params.1.npz"), "rb").read())
    assert utils.get_latest_checkpoint_index(os.path.join(folder, "model")) == 12, os.listdir(os.path.join(folder, "model"))
    for i in range(2, 13):
        os.remove(os.path.join(folder, "model", f"params.{i}.npz"))
    t2 = main.main(["--toy", "--gpu", "--max-steps", "20", "--model-output", folder])  # resumes: n_batches restored to 20
    assert t2.train_state.n_checkpoints == 1, t2.train_state.__dict__
    assert t2.train_state.n_batches == 21, t2.train_state.__dict__
    steps = int(t2.model.store.step_state[0].item())
    assert steps == 21, f"Adam step counter {steps}: optimizer state was not restored"


def test_reconstruction_sampler_writes_midi(gpu, tmp_path):
    """the trainer's sampler hook writes the batch's reconstruction as .mid files (both ends)"""
    from music_style_transfer.VarAutoEncoder import main
    from music_style_transfer.MIDIUtil.midi_io import EventBasedMIDIReader
    for extra, kind in ((["--pianoroll"], "pianoroll"), ([], "token")):
        flags = [f for f in SCRIPT_FLAGS]
        flags[flags.index("--sampling-frequency") + 1] = "3"
        flags[flags.index("--latent-dim") + 1] = "64"
        out = tmp_path / kind
        t = main.main(flags + extra + ["--data", MIDI, "--model-output", str(out / "m"), "--out-samples", str(out / "s"),
                                       "--max-steps", "4"])
        assert t.model.engine_config.kind == kind
        found = []
        for root, _, files in os.walk(str(out)):
            found += [os.path.join(root, f) for f in files if f.startswith("reconstruction_") and f.endswith(".mid")]
        assert found, "the sampler hook wrote no reconstruction"
        melodies = EventBasedMIDIReader().read_file(found[0])  # parses as a standard MIDI file
        assert isinstance(melodies, list)
