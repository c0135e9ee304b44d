"""CPU experiment (not a test): which 16-bit roundings of the encoder move the batch-mean KL at the RAW Xavier init?
Monkey-patches the oracle's blocks to round chosen tensors to bf16 and compares with the unrounded forward pass, all on
weights already rounded to bf16 where the kernels read them in 16 bits (the `consumed weights` of the GPU tests).

    python tests/diag_rounding.py [seed ...]
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
from oracle import vae_oracle as O
from test_step_gpu import _setup

DIMS = (128, 128, 2, 64, 256, 2, 8, 128, 1, 8)
bf = lambda t: t.to(torch.bfloat16).to(torch.float32)


def run(seed, B=64, T=256):
    _, _, ocfg, _, params, batch, eps = _setup("pianoroll", DIMS, B, T, seed, sigma_bias=0.0, ragged=False)
    shadow = [k for k in params if k.endswith("weight") and (".att." in k or ".ff" in k or k.endswith("embedding.weight") or k == "decoder.output_layer.weight")]
    for k in shadow:
        params[k] = bf(torch.from_numpy(params[k])).numpy()
    P = O.to_torch_params(params, requires_grad=False)

    def kl_of(rounders):
        """rounders: set of names of tensors to round: 'emb','qkv','att','h','x','a','top_tail' ..."""
        masks = {}
        R = lambda name, t: bf(t) if name in rounders else t
        De, H = ocfg.e_model, ocfg.e_heads
        x = batch["x"]; classes = batch["classes"]
        valid = torch.ones(B, T)
        tok = O.input_embedding(ocfg, P["encoder.embedding.weight"], x)
        h = P["encoder.class2hid.weight"][classes.long()][:, None, :] + tok
        pos = torch.from_numpy(O.positional_encodings(De, T)).float()
        h = R("emb", np.sqrt(De) * h + pos)
        for i in range(ocfg.e_layers):
            top = i == ocfg.e_layers - 1
            pre = f"encoder.layer{i}"
            tag = "top_" if top else ""
            dh = De // H
            split = lambda t: t.reshape(B, T, H, dh).transpose(1, 2)
            K = split(R(tag + "qkv", O.dense(h, P[f"{pre}.att.W_k.weight"], P[f"{pre}.att.W_k.bias"])))
            V = split(R(tag + "qkv", O.dense(h, P[f"{pre}.att.W_v.weight"], P[f"{pre}.att.W_v.bias"])))
            Q = split(R(tag + "qkv", O.dense(h, P[f"{pre}.att.W_q.weight"], P[f"{pre}.att.W_q.bias"])))
            logits = torch.matmul(K, Q.transpose(-1, -2)) / np.sqrt(dh)
            probs = torch.softmax(logits, dim=-1)
            out = torch.matmul(probs.transpose(-1, -2), V).transpose(1, 2).reshape(B, T, De)
            out = R(tag + "att", out)
            a = O.dense(out, P[f"{pre}.att.W_proj.weight"], P[f"{pre}.att.W_proj.bias"])
            h1 = R(tag + "h", h + a)
            x1 = R(tag + "x", O.layer_norm(h1, P[f"{pre}.ln1.gamma"], P[f"{pre}.ln1.beta"]))
            f = R(tag + "a", torch.relu(O.dense(x1, P[f"{pre}.ff1.weight"], P[f"{pre}.ff1.bias"])))
            h2 = R(tag + "h", x1 + O.dense(f, P[f"{pre}.ff2.weight"], P[f"{pre}.ff2.bias"]))
            h = R(tag + "x", O.layer_norm(h2, P[f"{pre}.ln2.gamma"], P[f"{pre}.ln2.beta"]))
        lat = O.dense(h[:, 0, :], P["encoder.latent_proj.weight"], P["encoder.latent_proj.bias"])
        Z = ocfg.latent_dim
        return O.variational_kl(lat[:, :Z], lat[:, Z:]).mean().item(), lat[:, Z:]

    ref, sref = kl_of(set())
    allr = {"emb", "qkv", "att", "h", "x", "a", "top_qkv", "top_att", "top_h", "top_x", "top_a"}
    print(f"seed {seed}: KL {ref:.4f}; min|sigma| {sref.abs().min():.2e}; #|sigma|<1e-2: {(sref.abs() < 1e-2).sum().item()}")
    for name, rs in (("all", allr), ("layer0 only", {"emb", "qkv", "att", "h", "x", "a"}), ("top layer only", {"top_qkv", "top_att", "top_h", "top_x", "top_a"}),
                     ("top: qkv", {"top_qkv"}), ("top: att", {"top_att"}), ("top: h1,h2", {"top_h"}), ("top: x1,x2", {"top_x"}), ("top: a", {"top_a"}),
                     ("all but top h,x", allr - {"top_h", "top_x"}), ("all but top h,x,a,att", allr - {"top_h", "top_x", "top_a", "top_att"}),
                     ("l0: x only", {"x"}), ("l0: emb", {"emb"})):
        kl, s = kl_of(rs)
        print(f"   round {name:28s}: rel KL err {abs(kl - ref) / ref:.2e}   sigma rms err {((s - sref) ** 2).mean().sqrt():.2e}")


if __name__ == "__main__":
    torch.set_num_threads(8)
    for s in [int(a) for a in sys.argv[1:]] or [99]:
        run(s)
