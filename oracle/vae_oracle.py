"""CPU oracle for the VarAutoEncoder training step — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module, and
only as the checker / reported baseline; the product path (musicstyletransfer_amd) never does.

What it is: an op-by-op restatement, in primitive torch-CPU tensor ops (matmul, exp, log, where,
sum ...; fp32 by default, fp64 on request), of the reference's hot path. Each function cites the
reference lines (relative to /root/reference/music_style_transfer/) it restates. Gradients come
from torch autograd over these primitives, which is exactly what MXNet's autograd.record() /
loss.backward() (trainer.py:167-176) does over the same graph.

PARITY UNPINNED: the reference (Python on mxnet-cu90==1.3.0.post0 + python-midi) cannot be imported
here (ModuleNotFoundError: mxnet, midi — an ordinary missing module, nothing was denied), its tree
holds no tests, golden vectors or recorded numbers, so the third-party MXNet
operator semantics below are restated from MXNet 1.3's published behaviour and marked [mx]. What IS
pinned: the reference's own literal data (ToyData arrays, vocabulary constants, flag defaults, the
MIDI files) and analytic known answers — see tests/test_oracle.py.

[mx] semantics relied on (MXNet 1.3):
  Dense(flatten=False): y = x W^T + b, W [units, in_units]           Embedding: row gather
  LayerNorm(axis=-1, eps=1e-5): biased variance, gamma*(x-mean)/sqrt(var+eps)+beta
  softmax default axis=-1; SequenceMask(use_sequence_length, axis=1): 1 where t < len
  linalg_gemm2(A,B,transpose_b) = A B^T batched over leading dims; transpose_a likewise
  pick(x, idx, axis=-1): gather; mean(axis=0, exclude=True): mean over all other axes
  init.Xavier(): rnd_type uniform, factor_type avg, magnitude 3 -> U(+-sqrt(3/((fan_in+fan_out)/2)))
     with fan_out = shape[0], fan_in = prod(shape[1:]); applied to every '*weight' (Embedding too);
     '*bias' and LayerNorm beta = 0, gamma = 1
  optimizer.Adam + gluon.Trainer.step(B): rescale_grad = 1/B; g = clip(g*rescale + wd*w, +-c);
     m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; w -= lr*sqrt(1-b2^t)/(1-b1^t) * m/(sqrt(v)+eps)
"""
import math
from collections import OrderedDict

import numpy as np
import torch

PAD_ID = 0  # MIDIUtil/defaults.py:39
MASK_VALUE = -1e9  # transformer.py:111
FLIP_PRONE = {}  # attention prefix -> flip-prone padded-key logits of the last call (diagnostic, see attention())
QK_OVERRIDE = {}  # test hook: attention prefix -> (K [B,S,D], Q [B,S,D]) to evaluate the logits on (see attention())
QK_SEEN = {}  # ... and this restatement's own K, Q of the layers that were overridden


# ----------------------------------------------------------------------------------------------
# configuration (plain dict-like object so the oracle does not depend on the product package)
# ----------------------------------------------------------------------------------------------
class OracleConfig:
    def __init__(self, kind, in_dim, out_dim, num_classes, latent_dim, e_model, e_layers, e_heads, d_model, d_layers,
                 d_heads, e_dropout=0.0, d_dropout=0.0):
        assert kind in ("token", "pianoroll")
        self.kind = kind
        self.in_dim, self.out_dim = in_dim, out_dim  # vocabulary (token) or pitches (piano-roll)
        self.num_classes, self.latent_dim = num_classes, latent_dim
        self.e_model, self.e_layers, self.e_heads = e_model, e_layers, e_heads
        self.d_model, self.d_layers, self.d_heads = d_model, d_layers, d_heads
        self.e_dropout, self.d_dropout = e_dropout, d_dropout

    @staticmethod
    def toy():
        """main.py:14-38 (create_toy_model_config) with ToyData's 10 tokens / 3 classes (data.py:72-76)"""
        return OracleConfig("token", 10, 10, 3, 16, 32, 1, 2, 32, 1, 2)


# ----------------------------------------------------------------------------------------------
# parameters
# ----------------------------------------------------------------------------------------------
def param_shapes(cfg):
    """Ordered name -> shape, following the blocks' construction in model.py:57-71,206-227 and
    transformer.py:24-46,49-68,129-149,162-182. 58 tensors at e_layers=2, d_layers=1."""
    s = OrderedDict()
    De, Dd, Z, C = cfg.e_model, cfg.d_model, cfg.latent_dim, cfg.num_classes

    def layer(prefix, D, last_ln):
        for w in ("W_k", "W_q", "W_v", "W_proj"):  # transformer.py:65-68
            s[f"{prefix}.att.{w}.weight"] = (D, D)
            s[f"{prefix}.att.{w}.bias"] = (D,)
        s[f"{prefix}.ln1.gamma"] = (D,)
        s[f"{prefix}.ln1.beta"] = (D,)
        s[f"{prefix}.ff1.weight"] = (4 * D, D)  # transformer.py:36-38,145
        s[f"{prefix}.ff1.bias"] = (4 * D,)
        s[f"{prefix}.ff2.weight"] = (D, 4 * D)
        s[f"{prefix}.ff2.bias"] = (D,)
        s[f"{prefix}.{last_ln}.gamma"] = (D,)
        s[f"{prefix}.{last_ln}.beta"] = (D,)

    s["encoder.class2hid.weight"] = (C, De)  # model.py:62-63
    s["encoder.embedding.weight"] = (cfg.in_dim, De)  # model.py:65-66
    for i in range(cfg.e_layers):
        layer(f"encoder.layer{i}", De, "ln2")
    s["encoder.latent_proj.weight"] = (2 * Z, De)  # model.py:70-71
    s["encoder.latent_proj.bias"] = (2 * Z,)
    s["decoder.latent2hid.weight"] = (Dd, Z)  # model.py:214-215
    s["decoder.latent2hid.bias"] = (Dd,)
    s["decoder.class2hid.weight"] = (C, Dd)  # model.py:217-218
    s["decoder.embedding.weight"] = (cfg.out_dim, Dd)  # model.py:220-221
    for i in range(cfg.d_layers):
        layer(f"decoder.layer{i}", Dd, "ln3")
    s["decoder.output_layer.weight"] = (cfg.out_dim, Dd)  # model.py:225-227
    s["decoder.output_layer.bias"] = (cfg.out_dim,)
    return s


def init_params(cfg, rng):
    """trainer.py:103-105 model.initialize(mx.init.Xavier()) [mx]; rng = numpy Generator"""
    out = OrderedDict()
    for name, shape in param_shapes(cfg).items():
        if name.endswith("weight"):
            fan_out, fan_in = shape[0], int(np.prod(shape[1:]))
            scale = math.sqrt(3.0 / ((fan_in + fan_out) / 2.0))
            out[name] = rng.uniform(-scale, scale, size=shape).astype(np.float32)
        elif name.endswith("gamma"):
            out[name] = np.ones(shape, np.float32)
        else:  # bias, beta
            out[name] = np.zeros(shape, np.float32)
    return out


def positional_encodings(model_size, max_len):
    """transformer.py:204-211, verbatim arithmetic: exponent 2*i/D for EVERY column i, sin on even
    columns, cos on odd; built in float64 (numpy default) and cast by the caller."""
    pos = np.arange(max_len).reshape((-1, 1)) / np.power(10000, (2.0 / model_size) * np.arange(model_size).reshape((1, -1)))
    pos[:, 0::2] = np.sin(pos[:, 0::2])
    pos[:, 1::2] = np.cos(pos[:, 1::2])
    return pos


# ----------------------------------------------------------------------------------------------
# blocks
# ----------------------------------------------------------------------------------------------
def dense(x, W, b=None):
    y = torch.matmul(x, W.transpose(-1, -2))  # [mx] Dense
    return y + b if b is not None else y


def layer_norm(x, gamma, beta, eps=1e-5):
    mean = x.mean(-1, keepdim=True)
    var = ((x - mean) ** 2).mean(-1, keepdim=True)  # biased [mx]
    return gamma * (x - mean) / torch.sqrt(var + eps) + beta


def drop(x, mask):
    """gluon Dropout in train mode (trainer.py:167: always inside autograd.record()): x * keep/(1-p);
    `mask` already holds keep/(1-p) (or is None for p = 0)."""
    return x if mask is None else x * mask


def attention(P, prefix, x, key_valid, H, return_probs=False):
    """MultiHeadDotAttention.hybrid_forward + _mask_logits, transformer.py:79-126.
    x: [B,S,D] is both keys_values and queries (self-attention, :154,196); key_valid: [B,S] in {0,1}."""
    B, S, D = x.shape
    dh = D // H  # attention_dim, :137,170

    def split(t):  # :91-93 reshape + swapaxes(1,2) -> [B,H,S,dh]
        return t.reshape(B, S, H, dh).transpose(1, 2)

    K = dense(x, P[f"{prefix}.W_k.weight"], P[f"{prefix}.W_k.bias"])
    V = split(dense(x, P[f"{prefix}.W_v.weight"], P[f"{prefix}.W_v.bias"]))
    Q = dense(x, P[f"{prefix}.W_q.weight"], P[f"{prefix}.W_q.bias"])
    if prefix in QK_OVERRIDE:
        # test hook (not part of the reference): evaluate the logits on the K | Q values ANOTHER implementation produced (its
        # 16-bit projections), gradients still flowing to this one's — a straight-through substitution. Both sides then put
        # every padded-key logit on the same -1e9 + 64 n grid point, which removes the chaos described below from a
        # gradient comparison without removing the comparison. QK_SEEN keeps this side's own values for the caller to check.
        k_other, q_other = QK_OVERRIDE[prefix]
        QK_SEEN[prefix] = (K.detach().clone(), Q.detach().clone())
        K = K + (k_other.to(K.dtype) - K).detach()
        Q = Q + (q_other.to(Q.dtype) - Q).detach()
    K, Q = split(K), split(Q)
    logits = torch.matmul(K, Q.transpose(-1, -2))  # :96 gemm2(K, Q, transpose_b) -> [B,H,T_K,T_Q]
    logits = logits / torch.sqrt(torch.tensor(float(dh), dtype=x.dtype))  # :98
    # diagnostic (not part of the reference): padded-key logits with |x| >= 32 — fl(x - 1e9) lands on another multiple of 64
    # there, the row's softmax stops being uniform, and the result depends on the last bits of x: not reproducible by any
    # other evaluation order (tests skip the scale check of the logit-only gradients on steps where this happens)
    FLIP_PRONE[prefix] = int(((logits.detach().abs() >= 32.0) & (key_valid <= 0)[:, None, :, None]).sum())
    mask = torch.where(key_valid > 0, torch.zeros_like(key_valid), torch.ones_like(key_valid) * MASK_VALUE)  # :112-114
    logits = logits + mask.to(x.dtype)[:, None, :, None]  # :115-116,125 — same constant along q for a key row
    probs = torch.softmax(logits, dim=-1)  # :100, axis=-1 = the QUERY axis
    out = torch.matmul(probs.transpose(-1, -2), V)  # :102 gemm2(att_probs, V, transpose_a) -> [B,H,T_Q,dh]
    out = out.transpose(1, 2).reshape(B, S, D)  # :103
    y = dense(out, P[f"{prefix}.W_proj.weight"], P[f"{prefix}.W_proj.bias"])  # :104
    return (y, probs) if return_probs else y


def feed_forward(P, prefix, x, mask_hidden):
    """DualFeedForward, transformer.py:42-46"""
    h = torch.relu(dense(x, P[f"{prefix}.ff1.weight"], P[f"{prefix}.ff1.bias"]))
    h = drop(h, mask_hidden)
    return dense(h, P[f"{prefix}.ff2.weight"], P[f"{prefix}.ff2.bias"])


def encoder_layer(P, prefix, x, key_valid, H, masks):
    """TransformerEncoderLayer.hybrid_forward, transformer.py:151-159"""
    a = attention(P, f"{prefix}.att", x, key_valid, H)
    x = layer_norm(x + drop(a, masks.get(f"{prefix}.att")), P[f"{prefix}.ln1.gamma"], P[f"{prefix}.ln1.beta"])
    f = feed_forward(P, prefix, x, masks.get(f"{prefix}.ffh"))
    return layer_norm(x + drop(f, masks.get(f"{prefix}.ffo")), P[f"{prefix}.ln2.gamma"], P[f"{prefix}.ln2.beta"])


def decoder_layer(P, prefix, x, key_valid, H, masks):
    """TransformerDecoderLayer.hybrid_forward, transformer.py:192-201. Not causal (:174); the FFN
    'residual' is ff(x) + dropout(ff(x)) (:199-200) — reproduced, not fixed."""
    a = attention(P, f"{prefix}.att", x, key_valid, H)
    x = layer_norm(x + drop(a, masks.get(f"{prefix}.att")), P[f"{prefix}.ln1.gamma"], P[f"{prefix}.ln1.beta"])
    f = feed_forward(P, prefix, x, masks.get(f"{prefix}.ffh"))
    return layer_norm(f + drop(f, masks.get(f"{prefix}.ffo")), P[f"{prefix}.ln3.gamma"], P[f"{prefix}.ln3.beta"])


def input_embedding(cfg, table, x):
    """token path: Embedding row gather (model.py:86,241). piano-roll path: a multi-hot frame times the
    table = the sum of the active rows' embeddings (a bias-free Dense [P -> D]); a one-hot frame
    reproduces the gather exactly."""
    if cfg.kind == "token":
        return table[x.long()]
    return torch.matmul(x.to(table.dtype), table)


def encode(P, cfg, x, seq_lens, classes, masks):
    """Encoder.hybrid_forward, model.py:73-104"""
    B, T = x.shape[0], x.shape[1]
    dt = P["encoder.embedding.weight"].dtype
    if cfg.kind == "token":
        valid = torch.where(x != PAD_ID, torch.ones_like(x, dtype=dt), torch.zeros_like(x, dtype=dt))  # :81-83
    else:
        valid = (torch.arange(T)[None, :] < seq_lens.long()[:, None]).to(dt)
    tok = input_embedding(cfg, P["encoder.embedding.weight"], x)  # :86
    cls = P["encoder.class2hid.weight"][classes.long()]  # :89
    h = cls[:, None, :] + tok  # :91
    De = cfg.e_model
    pos = torch.from_numpy(positional_encodings(De, T)).to(dt)
    h = torch.sqrt(torch.tensor(float(De), dtype=dt)) * h + pos  # transformer.py:270
    for i in range(cfg.e_layers):
        h = encoder_layer(P, f"encoder.layer{i}", h, valid, cfg.e_heads, masks)  # :271-272
    last = h[:, 0, :]  # model.py:97
    lat = dense(last, P["encoder.latent_proj.weight"], P["encoder.latent_proj.bias"])  # :100
    Z = cfg.latent_dim
    return lat[:, :Z], lat[:, Z:]  # :103 split -> means, stddevs (raw linear outputs)


def decode_train(P, cfg, x, seq_lens, z, classes, masks):
    """Decoder.forward_train, model.py:237-257; returns the pre-activation output ("logits")."""
    B, T = x.shape[0], x.shape[1]
    dt = z.dtype
    Dd = cfg.d_model
    tok = input_embedding(cfg, P["decoder.embedding.weight"], x)  # :241
    init = dense(z, P["decoder.latent2hid.weight"], P["decoder.latent2hid.bias"]) + P["decoder.class2hid.weight"][classes.long()]  # :229-232
    h = torch.cat([init[:, None, :], tok], dim=1)  # :244
    valid = (torch.arange(T + 1)[None, :] < (seq_lens.long() + 1)[:, None]).to(dt)  # :246-247 SequenceMask(len+1)
    pos = torch.from_numpy(positional_encodings(Dd, T + 1)).to(dt)
    h = torch.sqrt(torch.tensor(float(Dd), dtype=dt)) * h + pos  # transformer.py:237
    for i in range(cfg.d_layers):
        h = decoder_layer(P, f"decoder.layer{i}", h, valid, cfg.d_heads, masks)
    h = h[:, 1:, :]  # model.py:253
    return dense(h, P["decoder.output_layer.weight"], P["decoder.output_layer.bias"])  # :256 (before softmax)


def model_forward(P, cfg, x, seq_lens, classes, eps, masks=None):
    """Model.hybrid_forward, model.py:287-296. `eps` replaces mx.nd.random_normal (:292).
    Returns (probs, means, stddevs, logits): probs = softmax (token) / sigmoid (piano-roll)."""
    masks = masks or {}
    means, stds = encode(P, cfg, x, seq_lens, classes, masks)
    z = means + eps * stds  # :292
    logits = decode_train(P, cfg, x, seq_lens, z, classes, masks)
    probs = torch.softmax(logits, dim=-1) if cfg.kind == "token" else torch.sigmoid(logits)
    return probs, means, stds, logits


# ----------------------------------------------------------------------------------------------
# incremental decoding (inference)
# ----------------------------------------------------------------------------------------------
def decode_incremental(P, cfg, z, classes, fed, attention="query"):
    """Decoder.forward_inference / TransformerDecoder.forward_inference / compute_with_cache as evidently intended
    (model.py:259-272, transformer.py:70-77,242-249; the code as committed never appends to its cache and does not match
    its callers): position 0 = the initial state latent2hid(z) + class2hid(c) (model.py:229-232), position
    t >= 1 = the embedding of the token (or frame) fed there, each scaled by sqrt(D) and given pos[t] (transformer.py:246);
    every layer appends the new row's keys and values to its cache (:74) and the ONE new query attends to all cached rows
    with the layer's own arithmetic (:96-102): logits [B,H,T_K,1], mask, softmax over the LAST axis — the query axis, of
    size 1, so every probability is 1 — then probs^T V. attention='key' is the conventional alternative (softmax over the
    cached keys). Dropout is the identity (no autograd.record()).
    fed: [B, n] token ids or [B, n, P] frames fed at positions 1..n. Returns the n output distributions [B, n, V]."""
    B = z.shape[0]
    Dd, H = cfg.d_model, cfg.d_heads
    dh = Dd // H
    dt = z.dtype
    n = fed.shape[1]
    pos = torch.from_numpy(positional_encodings(Dd, n + 1)).to(dt)
    sq = torch.sqrt(torch.tensor(float(Dd), dtype=dt))
    init = dense(z, P["decoder.latent2hid.weight"], P["decoder.latent2hid.bias"]) + P["decoder.class2hid.weight"][classes.long()]
    caches = [{"k": [], "v": []} for _ in range(cfg.d_layers)]
    outs = []
    for t in range(n + 1):
        x = init if t == 0 else input_embedding(cfg, P["decoder.embedding.weight"], fed[:, t - 1])
        x = sq * x + pos[t]
        for i in range(cfg.d_layers):
            pre = f"decoder.layer{i}"
            c = caches[i]
            c["k"].append(dense(x, P[f"{pre}.att.W_k.weight"], P[f"{pre}.att.W_k.bias"]))   # transformer.py:74 (append)
            c["v"].append(dense(x, P[f"{pre}.att.W_v.weight"], P[f"{pre}.att.W_v.bias"]))
            K = torch.stack(c["k"], 1).reshape(B, t + 1, H, dh).transpose(1, 2)            # [B,H,T_K,dh]
            V = torch.stack(c["v"], 1).reshape(B, t + 1, H, dh).transpose(1, 2)
            Q = dense(x, P[f"{pre}.att.W_q.weight"], P[f"{pre}.att.W_q.bias"]).reshape(B, 1, H, dh).transpose(1, 2)
            logits = torch.matmul(K, Q.transpose(-1, -2)) / math.sqrt(dh)                  # [B,H,T_K,1]; the mask is all ones (:243)
            probs = torch.softmax(logits, dim=-1 if attention == "query" else -2)          # :100 softmax over the query axis
            att = torch.matmul(probs.transpose(-1, -2), V).transpose(1, 2).reshape(B, Dd)
            a = dense(att, P[f"{pre}.att.W_proj.weight"], P[f"{pre}.att.W_proj.bias"])
            x1 = layer_norm(x + a, P[f"{pre}.ln1.gamma"], P[f"{pre}.ln1.beta"])
            f = feed_forward(P, pre, x1, None)
            x = layer_norm(f + f, P[f"{pre}.ln3.gamma"], P[f"{pre}.ln3.beta"])             # transformer.py:199-200
        if t >= 1:  # position 0's output is dropped (model.py:253)
            lg = dense(x, P["decoder.output_layer.weight"], P["decoder.output_layer.bias"])
            outs.append(torch.softmax(lg, -1) if cfg.kind == "token" else torch.sigmoid(lg))
    return torch.stack(outs, 1)


# ----------------------------------------------------------------------------------------------
# losses (loss.py)
# ----------------------------------------------------------------------------------------------
def variational_kl(means, stds):
    """VariationalKLLoss, loss.py:8-12 — log(sigma^2) with no epsilon"""
    return (0.5 * (stds * stds + means * means - 1 - torch.log(stds * stds))).sum(1)


def softmax_cross_entropy(probs, labels):
    """SoftmaxCrossEntropy, loss.py:16-23: -log pick(probs, label) * (label != 0), mean over the
    non-batch axes (i.e. divided by the padded length T, not by the number of valid tokens)"""
    mask = torch.where(labels != 0, torch.ones_like(labels, dtype=probs.dtype), torch.zeros_like(labels, dtype=probs.dtype))
    logp = torch.log(probs)
    picked = -torch.gather(logp, -1, labels.long().unsqueeze(-1)).squeeze(-1)
    return (picked * mask).mean(dim=tuple(range(1, picked.dim())))


def binary_cross_entropy(pred, label, from_sigmoid=False, label_smoothing=0.0, negative_label_downweighting=True):
    """BinaryCrossEntropy, loss.py:27-80 (including (w*bce)*bce where label == 0, :52-54)"""
    if not from_sigmoid:
        pred = torch.sigmoid(pred)  # :40-42
    label = label.to(pred.dtype)
    s = (1.0 - label_smoothing) * label + label_smoothing * 0.5  # :34-36
    bce = -1 * (s * torch.log(1e-12 + pred) + (1 - s) * torch.log(1e-12 + (1.0 - pred)))  # :48
    red = tuple(range(1, label.dim()))
    if negative_label_downweighting:
        pos = torch.where(label == 1.0, torch.ones_like(label), torch.zeros_like(label))  # :61-63
        neg = torch.where(label == 1.0, torch.zeros_like(label), torch.ones_like(label))  # :64-66
        w = pos.sum(dim=red) / (neg.sum(dim=red) + 1e-12)  # :69-74
        w = w.reshape((-1,) + (1,) * (label.dim() - 1))
        bce = torch.where(label == 0.0, (w * bce) * bce, bce)  # :52-54
    return bce.mean(dim=red)  # :56


# ----------------------------------------------------------------------------------------------
# the training step (trainer.py:155-186)
# ----------------------------------------------------------------------------------------------
def to_torch_params(params_np, dtype=torch.float32, requires_grad=True):
    return OrderedDict((k, torch.tensor(v, dtype=dtype, requires_grad=requires_grad)) for k, v in params_np.items())


def step_losses(P, cfg, batch, eps, kl_weight=1.0, label_smoothing=0.0, negative_label_downscaling=False, masks=None):
    """trainer.py:167-172: returns (loss[B], recon[B], kl[B], probs, means, stds)"""
    probs, means, stds, logits = model_forward(P, cfg, batch["x"], batch["seq_lens"], batch["classes"], eps, masks)
    if cfg.kind == "token":
        recon = softmax_cross_entropy(probs, batch["labels"])  # :170
    else:
        recon = binary_cross_entropy(probs, batch["labels"], from_sigmoid=True, label_smoothing=label_smoothing,
                                     negative_label_downweighting=negative_label_downscaling)
    kl = variational_kl(means, stds)  # :171
    return recon + kl_weight * kl, recon, kl, probs, means, stds  # :172


def adam_update(w, g, m, v, t, lr, beta1=0.9, beta2=0.999, epsilon=1e-8, wd=0.0, rescale_grad=1.0, clip_gradient=-1.0):
    """[mx] optimizer.Adam.update + adam_update op (reached from trainer.py:177)"""
    g = g * rescale_grad + wd * w
    if clip_gradient >= 0:
        g = torch.clamp(g, -clip_gradient, clip_gradient)
    m = beta1 * m + (1.0 - beta1) * g
    v = beta2 * v + (1.0 - beta2) * g * g
    lr_t = lr * math.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t)
    return w - lr_t * m / (torch.sqrt(v) + epsilon), m, v


class OracleTrainer:
    """Minimal restatement of Trainer._step (trainer.py:155-186) over the functions above."""

    def __init__(self, cfg, params_np, lr=3e-4, clip_gradient=1.0, kl_weight=1.0, label_smoothing=0.0,
                 negative_label_downscaling=False, dtype=torch.float32):
        self.cfg, self.dtype = cfg, dtype
        self.P = to_torch_params(params_np, dtype)
        self.m = OrderedDict((k, torch.zeros_like(p)) for k, p in self.P.items())
        self.v = OrderedDict((k, torch.zeros_like(p)) for k, p in self.P.items())
        self.t = 0
        self.lr, self.clip, self.kl_weight = lr, clip_gradient, kl_weight
        self.ls, self.nld = label_smoothing, negative_label_downscaling
        self.kl_sum = self.total_sum = 0.0  # trainer.py:115-116 CustomMetric(mean)
        self.count = 0

    def load_state(self, w, m=None, v=None, t=None):
        """overwrite parameters (and optionally Adam moments / step count) from name -> numpy arrays"""
        with torch.no_grad():
            for k, p in self.P.items():
                p.copy_(torch.as_tensor(np.asarray(w[k]), dtype=self.dtype))
                if m is not None:
                    self.m[k] = torch.as_tensor(np.asarray(m[k]), dtype=self.dtype).clone()
                if v is not None:
                    self.v[k] = torch.as_tensor(np.asarray(v[k]), dtype=self.dtype).clone()
        if t is not None:
            self.t = int(t)

    def step(self, batch, eps, masks=None, is_train=True, qk_override=None):
        """qk_override (tests only): attention prefix -> (K, Q) of another implementation, see attention()"""
        for p in self.P.values():
            p.grad = None
        QK_OVERRIDE.clear()
        QK_SEEN.clear()
        QK_OVERRIDE.update(qk_override or {})
        try:
            loss, recon, kl, probs, means, stds = step_losses(self.P, self.cfg, batch, eps.to(self.dtype), self.kl_weight, self.ls,
                                                              self.nld, masks)
        finally:
            QK_OVERRIDE.clear()
        grads = None
        if is_train:
            loss.sum().backward()  # :176 head gradient of ones
            B = batch["x"].shape[0]
            self.t += 1
            grads = OrderedDict((k, p.grad.detach().clone()) for k, p in self.P.items())
            with torch.no_grad():
                for k, p in self.P.items():
                    w, m, v = adam_update(p.detach(), p.grad, self.m[k], self.v[k], self.t, self.lr, rescale_grad=1.0 / B,
                                          clip_gradient=self.clip)  # :177 step(batch_size)
                    p.copy_(w)
                    self.m[k], self.v[k] = m, v
        self.kl_sum += float(kl.detach().sum())
        self.total_sum += float(loss.detach().sum())
        self.count += loss.numel()
        return {"loss": loss.detach(), "recon": recon.detach(), "kl": kl.detach(), "probs": probs.detach(),
                "means": means.detach(), "stds": stds.detach(), "grads": grads, "flip_prone": dict(FLIP_PRONE), "qk_seen": dict(QK_SEEN)}

    def metrics(self):
        return {"kl_loss": self.kl_sum / max(1, self.count), "total_loss": self.total_sum / max(1, self.count)}


# ----------------------------------------------------------------------------------------------
# the reference's literal data
# ----------------------------------------------------------------------------------------------
def toy_batch():
    """ToyData, data.py:62-70, verbatim arrays"""
    return {
        "x": torch.tensor([[1, 5, 6, 7, 0], [1, 6, 7, 8, 0], [1, 7, 8, 9, 0]], dtype=torch.int64),
        "seq_lens": torch.tensor([4, 4, 4], dtype=torch.int64),
        "classes": torch.tensor([0, 1, 2], dtype=torch.int64),
        "labels": torch.tensor([[5, 6, 7, 2, 0], [6, 7, 8, 2, 0], [7, 8, 9, 2, 0]], dtype=torch.int64),
    }


def synthetic_pianoroll_batch(rng, B, T, P, num_classes=2, density=0.04, ragged=False):
    """SURVEY §8d synthetic input: Bernoulli(density) piano-roll frames; frame 0 of the input is a
    reserved start row (pitch 0 only), labels are the roll shifted by one frame (next-frame target,
    the piano-roll analogue of tokens=[SOS,data], labels=[data,PAD]: data.py:160-168)."""
    roll = (rng.random((B, T + 1, P)) < density).astype(np.uint8)
    x = roll[:, :T, :].copy()
    x[:, 0, :] = 0
    x[:, 0, 0] = 1
    labels = roll[:, 1:, :].copy()
    seq = rng.integers(T // 2, T + 1, size=B) if ragged else np.full(B, T)
    classes = rng.integers(0, num_classes, size=B)
    return {"x": torch.from_numpy(x), "seq_lens": torch.from_numpy(seq.astype(np.int64)),
            "classes": torch.from_numpy(classes.astype(np.int64)), "labels": torch.from_numpy(labels)}
