"""The token path at the script's widths on real MIDI chunks (tests/test_step_gpu.py::test_token_path_script_widths_*):
cosine and norm ratio of the gradients that exist only through the attention logits, per step. Not a test."""
import os, sys, math
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
from tests.test_step_gpu import _setup, _cos, midi_token_batch

dtype = torch.bfloat16 if (len(sys.argv) < 2 or sys.argv[1] == "bf16") else torch.float16
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
batch = midi_token_batch()
B, T = batch["x"].shape
dims = (293, 293, 2, 256, 256, 2, 8, 128, 1, 8)
O, E, ocfg, ecfg, params, batch, eps = _setup("token", dims, B, T, 41, sigma_bias=1.5, ragged=True, batch=batch)
gpu = torch.device("cuda", 0)
ot = O.OracleTrainer(ocfg, params, lr=3e-4, clip_gradient=1.0)
store = E.ParamStore(ecfg, gpu, dtype, params_np=params)
plan = E.StepPlan(store, B, T, lr=3e-4, clip_gradient=1.0, want_probs=True)
plan.load_batch(batch["x"], batch["seq_lens"], batch["classes"], batch["labels"], eps)
print("seq_lens", batch["seq_lens"].tolist(), "T", T)
for s in range(steps):
    ot.load_state(store.to_numpy("w"), store.to_numpy("m"), store.to_numpy("v"), int(store.step_state[0].item()))
    ref = ot.step(batch, torch.from_numpy(eps))
    plan.step_kernels(True)
    torch.cuda.synchronize()
    g = store.to_numpy("g")
    print(f"step {s} flip-prone padded-key logits in the oracle:", ref["flip_prone"])
    for name, rg in ref["grads"].items():
        if not (".att.W_k." in name or ".att.W_q.weight" in name or name.startswith("decoder.latent2hid") or name == "decoder.class2hid.weight"
                or ".att.W_v.weight" in name):
            continue
        rg = rg.numpy()
        gg = g[name] / (plan.gscale_enc if name.startswith("encoder.") else plan.gscale)
        c = _cos(gg, rg)
        ratio = float(np.linalg.norm(gg.astype(np.float64)) / max(np.linalg.norm(rg.astype(np.float64)), 1e-300))
        print(f"step {s} {name:40s} cos {c:7.4f} ratio {ratio:6.3f} |ref| {np.linalg.norm(rg):9.3g}")
