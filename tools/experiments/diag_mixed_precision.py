"""CPU emulation (not a test) for VERDICT r03 item 5: would fp16 ENCODER activations (decoder left in bf16) bring the raw-Xavier-init
KL / ELBO under north_star's 1e-3, and what is the largest encoder activation (fp16 tops out at 65 504)?

The encoder forward of the oracle with every tensor the engine keeps in 16 bits rounded to bf16 / fp16 (weights as the kernels consume
them: 16-bit shadows of the GEMM weights in the same type), against the unrounded fp32 forward; KL = 0.5 sum(sigma^2 + mu^2 - 1 -
log sigma^2) (loss.py:9). ELBO's other term (the reconstruction loss) already meets 1e-3 in bf16 (tests/test_step_gpu.py).

    python tools/experiments/diag_mixed_precision.py            # B 64 x T 256 and B 32 x T 1024, seeds 99 1234 7
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from oracle import vae_oracle as O
from test_step_gpu import _setup

DIMS = (128, 128, 2, 64, 256, 2, 8, 128, 1, 8)
ROUND = {"fp32": lambda t: t, "bf16": lambda t: t.to(torch.bfloat16).float(), "fp16": lambda t: t.to(torch.float16).float()}


def encoder_kl(ocfg, params, batch, B, T, mode):
    R = ROUND[mode]
    P = {k: torch.from_numpy(v) for k, v in params.items()}
    shadow = lambda k: k.endswith("weight") and (".att." in k or ".ff" in k or k.endswith("embedding.weight"))
    P = {k: (R(v) if shadow(k) else v) for k, v in P.items()}
    De, H = ocfg.e_model, ocfg.e_heads
    dh = De // H
    amax = 0.0

    def keep(t):
        nonlocal amax
        amax = max(amax, float(t.abs().max()))
        return R(t)
    tok = O.input_embedding(ocfg, P["encoder.embedding.weight"], batch["x"])
    h = P["encoder.class2hid.weight"][batch["classes"].long()][:, None, :] + tok
    h = keep(np.sqrt(De) * h + torch.from_numpy(O.positional_encodings(De, T)).float())
    for i in range(ocfg.e_layers):
        pre = f"encoder.layer{i}"
        split = lambda t: t.reshape(B, T, H, dh).transpose(1, 2)
        K, V, Q = (split(keep(O.dense(h, P[f"{pre}.att.W_{n}.weight"], P[f"{pre}.att.W_{n}.bias"]))) for n in "kvq")
        probs = torch.softmax(torch.matmul(K, Q.transpose(-1, -2)) / np.sqrt(dh), dim=-1)
        out = keep(torch.matmul(R(probs).transpose(-1, -2), V).transpose(1, 2).reshape(B, T, De))  # (P is a 16-bit MFMA operand)
        h1 = keep(h + O.dense(out, P[f"{pre}.att.W_proj.weight"], P[f"{pre}.att.W_proj.bias"]))
        x1 = keep(O.layer_norm(h1, P[f"{pre}.ln1.gamma"], P[f"{pre}.ln1.beta"]))
        f = keep(torch.relu(O.dense(x1, P[f"{pre}.ff1.weight"], P[f"{pre}.ff1.bias"])))
        h2 = keep(x1 + O.dense(f, P[f"{pre}.ff2.weight"], P[f"{pre}.ff2.bias"]))
        h = keep(O.layer_norm(h2, P[f"{pre}.ln2.gamma"], P[f"{pre}.ln2.beta"]))
    lat = O.dense(h[:, 0, :], P["encoder.latent_proj.weight"], P["encoder.latent_proj.bias"])  # fp32 in the engine (latent block)
    Z = ocfg.latent_dim
    return O.variational_kl(lat[:, :Z], lat[:, Z:]).mean().item(), lat[:, Z:], amax


def run(seed, B, T):
    _, _, ocfg, _, params, batch, _ = _setup("pianoroll", DIMS, B, T, seed, sigma_bias=0.0, ragged=False)
    with torch.no_grad():
        ref, sref, amax = encoder_kl(ocfg, params, batch, B, T, "fp32")
        line = f"B {B} T {T} seed {seed}: KL {ref:.4f}, {(sref.abs() < 1e-2).sum().item()} sigma below 1e-2 (min {sref.abs().min():.1e}), largest encoder activation {amax:.0f}"
        for mode in ("bf16", "fp16"):
            kl, s, _ = encoder_kl(ocfg, params, batch, B, T, mode)
            line += f" | {mode}: rel KL err {abs(kl - ref) / ref:.2e}, sigma rms err {((s - sref) ** 2).mean().sqrt():.2e}"
    print(line, flush=True)


if __name__ == "__main__":
    torch.set_num_threads(8)
    seeds = [int(a) for a in sys.argv[1:]] or [99, 1234, 7]
    for B, T in ((64, 256), (32, 1024)):
        for s in seeds:
            run(s, B, T)
