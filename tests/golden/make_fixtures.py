"""Regenerate tests/golden/* (run in the build container, where /root/reference is mounted):

    python tests/golden/make_fixtures.py

What is pinned, and from where:
  reference_constants.json : values read by IMPORTING the reference's two importable modules
        (MIDIUtil/defaults.py, VarAutoEncoder/config.py — neither needs mxnet / python-midi): vocabulary ranges,
        flag names and defaults. toy_data: the literal ToyData arrays of VarAutoEncoder/data.py:62-70, and the
        script hyper-parameters of scripts/train-vae.sh, parsed from the files' text (data, not code).
  midi/ : a 5-file single-track subset of the reference's training data work/data/guitar_bass (BASELINE
        configs[0]: "32-bar single-track MIDI subset") — data files, copied byte for byte.
  midi_event_counts.json : events per file for all 37 files from this repo's SMF reader; the totals agree with
        SURVEY.md §8a A19's independent probe (bass 16 988, guitar 38 048).
  oracle_*.npz : outputs of the CPU oracle (oracle/vae_oracle.py) on seeded inputs. The reference itself cannot
        run here (mxnet is not installed), so these pin the ORACLE against regressions and give the GPU tests
        committed expected values; they are not reference outputs (parity unpinned, see the oracle's header).
"""
import contextlib
import glob
import io
import json
import os
import re
import shutil
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)


def reference_constants():
    sys.path.insert(0, REF)
    with contextlib.redirect_stdout(io.StringIO()):
        import importlib
        d = importlib.import_module("music_style_transfer.MIDIUtil.defaults")
        c = importlib.import_module("music_style_transfer.VarAutoEncoder.config")
    vocab = {k: getattr(d, k) for k in ("PAD_ID", "SOS_ID", "EOS_ID", "FEATURE_OFFSET", "NOTE_ON_EVENTS", "NOTE_OFF_EVENTS",
                                        "TIMESHIFT_EVENTS", "NUM_EVENTS", "NUM_BINS", "MAX_TICKS", "MIN_TICKS",
                                        "NUM_TICKS_IN_A_BIN", "N_FEATURES_WITHOUT_SILENCE", "DEFAULT_BPM", "DEFAULT_RESOLUTION")}
    flags = {}
    for a in c.parser._actions:
        if a.dest == "help":
            continue
        flags[a.dest] = {"options": list(a.option_strings), "default": a.default,
                         "type": getattr(a.type, "__name__", None) if a.type else None, "choices": list(a.choices) if a.choices else None,
                         "store_true": type(a).__name__ == "_StoreTrueAction"}
    sys.path.remove(REF)
    for m in [k for k in sys.modules if k.startswith("music_style_transfer")]:
        del sys.modules[m]
    # literal data parsed from text
    src = open(os.path.join(REF, "music_style_transfer/VarAutoEncoder/data.py")).read()
    nums = re.findall(r"mx\.nd\.array\((\[[\[\]\d,\s]+\])\)", src)
    toy = {"tokens": json.loads(nums[0]), "seq_lens": json.loads(nums[1]), "classes": json.loads(nums[2]), "labels": json.loads(nums[3])}
    sh = open(os.path.join(REF, "scripts/train-vae.sh")).read()
    script = dict(re.findall(r"--([a-z\-]+) ([^\s\\]+)", sh))
    return {"vocab": vocab, "flags": flags, "toy_data": toy, "train_vae_sh": script}


def midi_fixtures():
    from musicstyletransfer_amd.MIDIUtil.midi_io import EventBasedMIDIReader
    r = EventBasedMIDIReader()
    counts = {}
    for cls in ("bass", "guitar"):
        for f in sorted(glob.glob(os.path.join(REF, "work/data/guitar_bass", cls, "*.mid"))):
            counts[f"{cls}/{os.path.basename(f)}"] = len(r.read_file(f)[0])
    # files closest to ~600 events (about 32 bars of a single-track riff): 2 bass + 3 guitar
    subset = sorted(counts, key=lambda k: abs(counts[k] - 600))
    picked = [k for k in subset if k.startswith("bass/")][:2] + [k for k in subset if k.startswith("guitar/")][:3]
    for k in picked:
        shutil.copyfile(os.path.join(REF, "work/data/guitar_bass", k), os.path.join(HERE, "midi", k))
    first = {}
    for k in picked:
        first[k] = [e.id for e in r.read_file(os.path.join(HERE, "midi", k))[0].notes[:64]]
    return {"counts": counts, "subset": picked, "first_64_event_ids": first}


def oracle_fixtures():
    from oracle import vae_oracle as O
    out = {}
    # (1) toy configuration, ToyData batch, 2 steps
    rng = np.random.default_rng(7)
    cfg = O.OracleConfig.toy()
    params = O.init_params(cfg, rng)
    eps = rng.standard_normal((3, 16)).astype(np.float32)
    tr = O.OracleTrainer(cfg, params, lr=1e-3, clip_gradient=1.0)
    r0 = tr.step(O.toy_batch(), torch.from_numpy(eps))
    r1 = tr.step(O.toy_batch(), torch.from_numpy(eps))
    np.savez_compressed(os.path.join(HERE, "oracle_toy.npz"), eps=eps, loss0=r0["loss"].numpy(), kl0=r0["kl"].numpy(),
                        recon0=r0["recon"].numpy(), probs0=r0["probs"].numpy(), means0=r0["means"].numpy(), stds0=r0["stds"].numpy(),
                        loss1=r1["loss"].numpy(), g_out=r0["grads"]["decoder.output_layer.weight"].numpy(),
                        g_latent=r0["grads"]["encoder.latent_proj.weight"].numpy(),
                        w_out_after2=tr.P["decoder.output_layer.weight"].detach().numpy(),
                        **{"p_" + k: v for k, v in params.items()})
    # (2) small piano-roll configuration
    rng = np.random.default_rng(12)
    dims = ("pianoroll", 40, 40, 2, 16, 64, 2, 2, 32, 1, 2)
    cfg = O.OracleConfig(*dims)
    params = O.init_params(cfg, rng)
    params["encoder.latent_proj.weight"][16:] *= 0.25
    params["encoder.latent_proj.bias"][16:] += 1.5
    batch = O.synthetic_pianoroll_batch(rng, 5, 19, 40, ragged=True)
    eps = rng.standard_normal((5, 16)).astype(np.float32)
    r = O.OracleTrainer(cfg, params, lr=1e-3).step(batch, torch.from_numpy(eps))
    np.savez_compressed(os.path.join(HERE, "oracle_pianoroll_small.npz"), eps=eps, x=batch["x"].numpy(), labels=batch["labels"].numpy(),
                        seq_lens=batch["seq_lens"].numpy(), classes=batch["classes"].numpy(), loss=r["loss"].numpy(),
                        kl=r["kl"].numpy(), recon=r["recon"].numpy(), means=r["means"].numpy(), stds=r["stds"].numpy(),
                        probs=r["probs"].numpy().astype(np.float16), g_out=r["grads"]["decoder.output_layer.weight"].numpy(),
                        **{"p_" + k: v for k, v in params.items()})
    return out


if __name__ == "__main__":
    consts = reference_constants()
    consts["midi"] = midi_fixtures()
    json.dump(consts, open(os.path.join(HERE, "reference_constants.json"), "w"), indent=1, sort_keys=True, default=list)
    oracle_fixtures()
    print("fixtures written to", HERE)
