"""Step engine: sequences the C-ABI kernels (include/mst_hip.h) into the VarAutoEncoder training
step of the reference — Trainer._step (VarAutoEncoder/trainer.py:155-179): forward
(model.py:287-296), CE/BCE + beta*KL (loss.py), backward, MXNet-rule Adam — with every
intermediate resident in HBM buffers allocated once, and the whole step captured in a hipGraph.

Layout in HBM
  * ParamStore: ONE flat fp32 buffer each for parameters, gradients (the bucket RCCL all-reduces),
    Adam m and v; a same-offset 16-bit shadow (GEMM B operands) and a second flat buffer of
    transposed 16-bit shadows (dgrad B operands / piano-roll embedding tables). W_k, W_q, W_v of a
    layer sit back to back so the three reference Dense layers run as one [3D, D] GEMM.
  * StepPlan(B, T): activations as [rows, ld] 16-bit matrices (ld = roundup8(width), pad columns
    zero), saved for backward; fp32 only for softmax statistics, LayerNorm mean/rstd, the latent
    block and the losses.

There is no torch autograd, no torch operator and no CPU fallback on this path: backward is written
out by hand below, mirroring the forward line by line.
"""
import math
import os
import warnings
from collections import OrderedDict

import numpy as np
import torch

from . import ops as o


def roundup(a, b):
    return (a + b - 1) // b * b


def _lib_ce_rows():
    from . import _lib
    return _lib.CE_MAX_WORKGROUPS


class VAEConfig:
    """Shape of the model: the union of ModelConfig/EncoderConfig/DecoderConfig/TransformerConfig
    (model.py:22-54, transformer.py:8-21) flattened, plus which ends are attached:
    kind='token' (Embedding in, softmax-CE out: the reference's executed path) or
    kind='pianoroll' (multi-hot frame x table in, sigmoid + BinaryCrossEntropy out)."""

    def __init__(self, kind, in_dim, out_dim, num_classes, latent_dim, e_model, e_layers, e_heads, d_model, d_layers,
                 d_heads, e_dropout=0.0, d_dropout=0.0):
        assert kind in ("token", "pianoroll")
        assert e_model % e_heads == 0 and d_model % d_heads == 0  # transformer.py:134,167
        self.kind = kind
        self.in_dim, self.out_dim = in_dim, out_dim
        self.num_classes, self.latent_dim = num_classes, latent_dim
        self.e_model, self.e_layers, self.e_heads = e_model, e_layers, e_heads
        self.d_model, self.d_layers, self.d_heads = d_model, d_layers, d_heads
        self.e_dropout, self.d_dropout = float(e_dropout), float(d_dropout)

    def as_dict(self):
        return dict(self.__dict__)


def positional_table(model_size, max_len):
    """transformer.py:204-211: exponent 2*i/D for every column, sin on even / cos on odd columns,
    float64 then cast (host-side constant, built once)."""
    pos = np.arange(max_len).reshape((-1, 1)) / np.power(10000, (2.0 / model_size) * np.arange(model_size).reshape((1, -1)))
    pos[:, 0::2] = np.sin(pos[:, 0::2])
    pos[:, 1::2] = np.cos(pos[:, 1::2])
    return pos.astype(np.float32)


def logical_param_shapes(cfg):
    """name -> shape in the reference's construction order (model.py:57-71,206-227;
    transformer.py:24-46,49-68,129-149,162-182): 58 tensors at e_layers=2, d_layers=1."""
    s = OrderedDict()
    De, Dd, Z, C = cfg.e_model, cfg.d_model, cfg.latent_dim, cfg.num_classes

    def layer(prefix, D, last_ln):
        for w in ("W_k", "W_q", "W_v", "W_proj"):
            s[f"{prefix}.att.{w}.weight"] = (D, D)
            s[f"{prefix}.att.{w}.bias"] = (D,)
        s[f"{prefix}.ln1.gamma"] = (D,)
        s[f"{prefix}.ln1.beta"] = (D,)
        s[f"{prefix}.ff1.weight"] = (4 * D, D)
        s[f"{prefix}.ff1.bias"] = (4 * D,)
        s[f"{prefix}.ff2.weight"] = (D, 4 * D)
        s[f"{prefix}.ff2.bias"] = (D,)
        s[f"{prefix}.{last_ln}.gamma"] = (D,)
        s[f"{prefix}.{last_ln}.beta"] = (D,)

    s["encoder.class2hid.weight"] = (C, De)
    s["encoder.embedding.weight"] = (cfg.in_dim, De)
    for i in range(cfg.e_layers):
        layer(f"encoder.layer{i}", De, "ln2")
    s["encoder.latent_proj.weight"] = (2 * Z, De)
    s["encoder.latent_proj.bias"] = (2 * Z,)
    s["decoder.latent2hid.weight"] = (Dd, Z)
    s["decoder.latent2hid.bias"] = (Dd,)
    s["decoder.class2hid.weight"] = (C, Dd)
    s["decoder.embedding.weight"] = (cfg.out_dim, Dd)
    for i in range(cfg.d_layers):
        layer(f"decoder.layer{i}", Dd, "ln3")
    s["decoder.output_layer.weight"] = (cfg.out_dim, Dd)
    s["decoder.output_layer.bias"] = (cfg.out_dim,)
    return s


def xavier_init(cfg, rng):
    """trainer.py:103-105 model.initialize(mx.init.Xavier()): uniform, factor 'avg', magnitude 3 for
    every '*weight' (Embedding tables included); biases / beta 0; gamma 1."""
    out = OrderedDict()
    for name, shape in logical_param_shapes(cfg).items():
        if name.endswith("weight"):
            fan_out, fan_in = shape[0], int(np.prod(shape[1:]))
            scale = math.sqrt(3.0 / ((fan_in + fan_out) / 2.0))
            out[name] = rng.uniform(-scale, scale, size=shape).astype(np.float32)
        elif name.endswith("gamma"):
            out[name] = np.ones(shape, np.float32)
        else:
            out[name] = np.zeros(shape, np.float32)
    return out


class ParamStore:
    """Flat parameter / gradient / Adam-state buffers and their 16-bit shadows."""

    def __init__(self, cfg, device, act_dtype=torch.bfloat16, params_np=None, seed=1234):
        self.cfg, self.device, self.act_dtype = cfg, device, act_dtype
        shapes = logical_param_shapes(cfg)
        # flat order: per attention block the three K,Q,V weights first (fused [3D,D] GEMM), then their biases
        order = []
        for name in shapes:
            if ".att.W_q." in name or ".att.W_v." in name:
                continue
            if name.endswith(".att.W_k.weight"):
                p = name[: -len("W_k.weight")]
                order += [p + "W_k.weight", p + "W_q.weight", p + "W_v.weight"]
            elif name.endswith(".att.W_k.bias"):
                p = name[: -len("W_k.bias")]
                order += [p + "W_k.bias", p + "W_q.bias", p + "W_v.bias"]
            else:
                order.append(name)
        if cfg.kind == "pianoroll":
            # the class table right behind the input embedding: [in_dim + C, De] is then ONE matrix, and the class-embedding
            # gradient is the last C rows of the embedding's weight-gradient problem (StepPlan.cls_fold)
            order.remove("encoder.class2hid.weight")
            order.insert(order.index("encoder.embedding.weight") + 1, "encoder.class2hid.weight")
        self.shapes, self.offsets = shapes, OrderedDict()
        off = 0
        for name in order:
            fused_tail = (".att.W_q." in name) or (".att.W_v." in name)
            if not fused_tail:
                off = roundup(off, 8)
            self.offsets[name] = off
            off += int(np.prod(shapes[name]))
        self.n = roundup(off, 8)
        self.n_params = sum(int(np.prod(s)) for s in shapes.values())
        f32 = dict(dtype=torch.float32, device=device)
        self.w = torch.zeros(self.n, **f32)
        self.g = torch.zeros(self.n, **f32)
        self.m = torch.zeros(self.n, **f32)
        self.v = torch.zeros(self.n, **f32)
        self.w16 = torch.zeros(self.n, dtype=act_dtype, device=device)
        self.step_state = torch.zeros(2, dtype=torch.int32, device=device)

        # transposed shadows: (source offset, rows, cols) -> dst [cols, roundup8(rows)]
        self.t_specs = OrderedDict()
        for side, D, L in (("encoder", cfg.e_model, cfg.e_layers), ("decoder", cfg.d_model, cfg.d_layers)):
            for i in range(L):
                p = f"{side}.layer{i}"
                self.t_specs[f"{p}.att.W_kqv"] = (self.offsets[f"{p}.att.W_k.weight"], 3 * D, D)
                self.t_specs[f"{p}.att.W_proj.weight"] = (self.offsets[f"{p}.att.W_proj.weight"], D, D)
                self.t_specs[f"{p}.ff1.weight"] = (self.offsets[f"{p}.ff1.weight"], 4 * D, D)
                self.t_specs[f"{p}.ff2.weight"] = (self.offsets[f"{p}.ff2.weight"], D, 4 * D)
        self.t_specs["decoder.output_layer.weight"] = (self.offsets["decoder.output_layer.weight"], cfg.out_dim, cfg.d_model)
        if cfg.kind == "pianoroll":
            self.t_specs["encoder.embedding.weight"] = (self.offsets["encoder.embedding.weight"], cfg.in_dim, cfg.e_model)
            self.t_specs["decoder.embedding.weight"] = (self.offsets["decoder.embedding.weight"], cfg.out_dim, cfg.d_model)
        desc, prefix, doff, self.t_off = [], [0], 0, OrderedDict()
        for name, (so, r, c) in self.t_specs.items():
            self.t_off[name] = doff
            desc += [so, doff, r, c]
            doff += c * roundup(r, 8)
            prefix.append(prefix[-1] + ((r + 31) // 32) * ((c + 31) // 32))
        self.wt16 = torch.zeros(max(doff, 8), dtype=act_dtype, device=device)
        self.t_desc = torch.tensor(desc, dtype=torch.int64, device=device)
        self.t_prefix = torch.tensor(prefix, dtype=torch.int64, device=device)
        self.t_tiles = prefix[-1]
        # Deferred shadow refresh (piano-roll ends): the transposed shadows of the matrices only the BACKWARD pass reads are rebuilt by
        # extra workgroups of the NEXT step's first launch (mst_gemm_nt_pair_begin, sh_*) instead of a launch of their own behind the
        # optimizer; the two embedding tables, which that first launch itself reads, are kept current by the optimizer launch
        # (mst_adam_flat_emb). MST_SHADOW_RIDE=0: the separate launch.
        emb_names = [n for n in ("encoder.embedding.weight", "decoder.embedding.weight") if n in self.t_specs]
        late = [n for n in self.t_specs if n not in emb_names]
        self.shadows_deferred = (cfg.kind == "pianoroll" and len(emb_names) == 2 and len(late) > 0 and
                                 os.environ.get("MST_SHADOW_RIDE", "1") != "0" and os.environ.get("MST_BEGIN_RIDE", "1") != "0")
        ldesc, lprefix = [], [0]
        for name in late:
            so, r, c = self.t_specs[name]
            ldesc += [so, self.t_off[name], r, c]
            lprefix.append(lprefix[-1] + ((r + 31) // 32) * ((c + 31) // 32))
        self.t_desc_late = torch.tensor(ldesc or [0, 0, 1, 1], dtype=torch.int64, device=device)
        self.t_prefix_late = torch.tensor(lprefix, dtype=torch.int64, device=device)
        self.t_n_late, self.t_tiles_late = len(late), lprefix[-1]
        self.emb_specs = [(self.t_specs[n][0], self.t_off[n], self.t_specs[n][1], self.t_specs[n][2]) for n in emb_names]

        # shared by every StepPlan of this store (plans run one after the other on one stream): the per-step RNG state
        # — ONE stream of seeds however many (B, T) shapes a run goes through — and the weight-gradient work buffer
        self._rng_state = None
        self._wgrad_scratch = []
        # running metric sums of every step since the last read (trainer.py:107-120,181-186), whatever plan ran it:
        #   metric_acc = [sum kl, sum total, count]; tok_parts = per-workgroup partial rows of the masked token metrics
        #   {sum -log p[label], #arg-max hits, #top-k hits, #valid} accumulated by mst_softmax_ce (token ends)
        # step_status = three int32 words of the step guard (mst_step_metrics), read with the metrics (same device->host copy):
        # {flags, skipped steps} — sticky, set on the device when a one-launch position-0 tail (mst_row_tail_*) could not finish —
        # and the number of steps the optimizer skipped because a loss was not finite (not sticky: the next batch is tried again)
        self._metric_buf = torch.zeros(8, **f32)
        self.metric_acc = self._metric_buf[:3]
        self.step_status = self._metric_buf[4:7].view(torch.int32)
        self.nonfinite_steps = 0      # steps skipped for a non-finite loss since the store was made (read_metrics adds them up)
        self.tail_fused = os.environ.get("MST_ROW_TAIL", "1") != "0"  # False: the five-launch form of the position-0 tails
        self.tail_checked = False     # row_tail_selfcheck() ran for this store
        self.tail_policy = os.environ.get("MST_TAIL_FAILURE", "fallback")  # or "raise"
        self.tail_failures = []       # (flags, skipped steps) of every failure seen
        self._tail_listeners = []
        self._rng_state_infer = None
        self.tok_parts = torch.zeros(_lib_ce_rows(), 4, **f32) if cfg.kind == "token" else None
        if params_np is None:
            params_np = xavier_init(cfg, np.random.default_rng(seed))
        self.load_numpy(params_np)

    def rng_state(self, seed=0):
        """uint64[4] device state of mst_step_begin / mst_rng_advance, created with the first plan's seed"""
        if self._rng_state is None:
            self._rng_state = torch.tensor([0, 0, seed ^ 0x5DEECE66D, 0], dtype=torch.int64, device=self.device)
        return self._rng_state

    def rng_state_inference(self):
        """the RNG state inference-mode forward passes advance (eps of Model.__call__ when none is given): never the
        training stream's, so that a sampling hook in the middle of training leaves the steps after it as they were"""
        if self._rng_state_infer is None:
            self._rng_state_infer = torch.tensor([0, 0, 0x1F123BB5 ^ 0x5DEECE66D, 0], dtype=torch.int64, device=self.device)
        return self._rng_state_infer

    def on_tail_failure(self, callback):
        """callback() after a failed position-0 tail made this store fall back to the five-launch form: holders of captured
        graphs must drop them (they were recorded with the one-launch kernels)"""
        self._tail_listeners.append(callback)

    def handle_step_status(self, flags, skipped):
        """A one-launch position-0 tail could not do its work in `skipped` steps since the last read (flags: _lib.TAIL_* /
        STEP_INCOMPLETE). The optimizer launches of those steps left the model untouched (step guard), so nothing wrong was
        learned; from here on the five-launch form runs (policy 'fallback'), or the run stops (policy 'raise')."""
        o.zero(self._metric_buf[4:6])
        self.tail_failures.append((int(flags), int(skipped)))
        self.tail_fused = False
        msg = (f"the one-launch position-0 tail of the top encoder layer failed (status flags {int(flags):#x}: "
               f"{'forward barrier timed out; ' if flags & 1 else ''}{'backward barrier timed out; ' if flags & 2 else ''}"
               f"{'tail incomplete at the end of a step; ' if flags & 16 else ''}"
               f"{int(skipped)} step(s) skipped without touching the model)")
        if self.tail_policy == "raise":
            raise RuntimeError(msg + " — MST_TAIL_FAILURE=raise")
        warnings.warn(msg + "; falling back to the five-launch form for the rest of the run", RuntimeWarning)
        for cb in self._tail_listeners:
            cb()

    def poll_status(self, reduce=None):
        """Cheap, NON-BLOCKING look at the step guard's sticky words, for the training loop to call every few steps (the
        words are otherwise read only with the metrics — at the periodic log — and after one failed tail every following
        step is a skipped batch until then). Two halves per call, no synchronisation in either: (1) if the copy the previous
        call started has arrived (event query), look at it: a set flag goes to handle_step_status() — fall back to the
        five-launch form at once, or raise; (2) start the next asynchronous copy of the words into page-locked memory
        behind everything launched so far on the current stream. reduce(tensor): data parallel — an in-place SUM over the
        ranks applied to a float copy of the words first, so that every rank sees a failure of ANY rank at the same poll
        and all of them stop together (handle_step_status raises there: a skipped update on one rank lets the replicas drift).
        Returns True when a failure was handled."""
        if getattr(self, "_poll_host", None) is None:
            self._poll_host = torch.zeros(2, dtype=torch.float32).pin_memory()
            self._poll_dev = torch.zeros(2, dtype=torch.float32, device=self.device)
            self._poll_event = torch.cuda.Event()
            self._poll_pending = False
        handled = False
        if self._poll_pending and self._poll_event.query():
            self._poll_pending = False
            flags, skipped = (int(v) for v in self._poll_host.tolist())
            if flags or skipped:
                if reduce is not None:  # (summed over the ranks: only "non-zero" means anything)
                    self.tail_fused = False
                    raise RuntimeError(f"a one-launch position-0 tail failed on at least one rank (status words summed over the ranks: "
                                       f"{flags}, {skipped} skipped step(s)); the replicas are no longer identical — restart from the "
                                       "last checkpoint with MST_ROW_TAIL=0")
                cur = self.step_status[:2].tolist()  # (one small blocking read, on the failure path only: the words as they are NOW)
                self.handle_step_status(cur[0] or flags, cur[1] or skipped)
                handled = True
        if not self._poll_pending:
            self._poll_dev.copy_(self.step_status[:2])
            if reduce is not None:
                reduce(self._poll_dev)
            self._poll_host.copy_(self._poll_dev, non_blocking=True)
            self._poll_event.record()
            self._poll_pending = True
        return handled

    def read_metrics(self, reset=True):
        """one device->host read of the running sums: {'kl_sum', 'total_sum', 'count'} and, for the token ends,
        {'nll_sum', 'acc_hits', 'topk_hits', 'n_tokens'} (the caller orders this after the steps it wants included).
        The step-status words travel in the same copy: a set flag is handled here (handle_step_status)."""
        buf = self._metric_buf.cpu()
        acc = buf[:3].tolist()
        flags, skipped, nonfinite = buf[4:7].view(torch.int32).tolist()
        if nonfinite:
            # (the optimizer left the model alone in those steps: an overflowed activation or sigma = 0 would have made every
            # gradient NaN — mst_step_metrics' non-finite guard)
            self.nonfinite_steps += int(nonfinite)
            o.zero(self._metric_buf[6:7])
            warnings.warn(f"{int(nonfinite)} training step(s) skipped: a per-sample loss was not finite (an fp16 activation overflow or "
                          "sigma = 0 under log(sigma^2)); parameters, moments and the step count were left as they were", RuntimeWarning)
        if flags or skipped:
            self.handle_step_status(flags, skipped)
        out = {"kl_sum": acc[0], "total_sum": acc[1], "count": acc[2], "skipped_steps": skipped, "nonfinite_steps": int(nonfinite)}
        if self.tok_parts is not None:
            t = self.tok_parts.cpu().double().sum(0).tolist()
            out.update(nll_sum=t[0], acc_hits=t[1], topk_hits=t[2], n_tokens=t[3])
        if reset:
            o.zero(self.metric_acc)
            if self.tok_parts is not None:
                o.zero(self.tok_parts)
        return out

    def wgrad_scratch(self, n_floats=16 * 1024 * 1024):
        """fp32 work buffer of the wgrad launch's two-pass reduction: one full resident round of 256 x 256 slab tiles
        (256 work items x 256 KiB = 64 MiB) is all mst_gemm_wgrad_batch_sums ever asks for. Buffers are never freed:
        captured graphs keep the pointer they were recorded with."""
        for t in self._wgrad_scratch:
            if t.numel() >= n_floats:
                return t
        t = torch.empty(n_floats, dtype=torch.float32, device=self.device)
        self._wgrad_scratch.append(t)
        return t

    # ---- views
    def _view(self, flat, name):
        shape = self.shapes[name]
        off = self.offsets[name]
        return flat[off: off + int(np.prod(shape))].view(*shape)

    def p(self, name):
        return self._view(self.w, name)

    def grad(self, name):
        return self._view(self.g, name)

    def h(self, name):
        return self._view(self.w16, name)

    def fused(self, flat, prefix, what):
        """[3D, D] weight or [3D] bias view of a layer's K,Q,V Dense layers (K rows first)"""
        D = self.shapes[f"{prefix}.att.W_k.weight"][0]
        off = self.offsets[f"{prefix}.att.W_k.{what}"]
        return flat[off: off + 3 * D * D].view(3 * D, D) if what == "weight" else flat[off: off + 3 * D]

    def t(self, name):
        so, r, c = self.t_specs[name]
        off = self.t_off[name]
        return self.wt16[off: off + c * roundup(r, 8)].view(c, roundup(r, 8))

    # ---- host <-> device
    def load_numpy(self, params_np):
        host = np.zeros(self.n, np.float32)
        for name, shape in self.shapes.items():
            a = np.asarray(params_np[name], np.float32)
            assert tuple(a.shape) == tuple(shape), f"{name}: expected {shape}, got {a.shape}"
            host[self.offsets[name]: self.offsets[name] + a.size] = a.reshape(-1)
        self.w.copy_(torch.from_numpy(host))
        self.refresh_shadows()

    def to_numpy(self, which="w"):
        host = getattr(self, which).detach().cpu().numpy()
        return OrderedDict((n, host[self.offsets[n]: self.offsets[n] + int(np.prod(s))].reshape(s).copy())
                           for n, s in self.shapes.items())

    def shadowed_names(self):
        """parameters the kernels consume through a 16-bit shadow (GEMM B operands); every other tensor — biases,
        LayerNorm gamma / beta, class tables, the latent block, token embedding tables — is read in fp32"""
        names = set()
        for key in self.t_specs:
            if key.endswith(".att.W_kqv"):
                p = key[: -len("W_kqv")]
                names.update({p + "W_k.weight", p + "W_q.weight", p + "W_v.weight"})
            else:
                names.add(key)
        return names

    def as_consumed_numpy(self):
        """name -> the values the kernels actually read: the 16-bit shadow (widened to fp32) for shadowed_names(), the
        fp32 master otherwise. Feeding these to a reference separates weight rounding from kernel error."""
        w = self.to_numpy("w")
        w16 = self.w16.detach().float().cpu().numpy()
        for n in self.shadowed_names():
            s = self.shapes[n]
            w[n] = w16[self.offsets[n]: self.offsets[n] + int(np.prod(s))].reshape(s).copy()
        return w

    def refresh_shadows(self):
        o.cast_to_act(self.w, self.w16)
        o.transpose_shadows(self.w, self.wt16, self.t_desc, self.t_prefix, len(self.t_specs), self.t_tiles)


class _Layer:
    pass


def row_tail_selfcheck(store, B=64, S=2):
    """One-time start-up check of the one-launch position-0 tails (mst_row_tail_fwd / _bwd) on THIS device, process and
    partition mode: both are run against the five launches they replace, on random rows and the store's own top-layer
    weights. Their grid barrier rests on properties no API guarantees (enough workgroups of an oversubscribed launch landing
    on one XCD, L1 behaviour of write-through lines — csrc/row_tail.hip), so shape alone does not decide whether the fused
    form is used: a mismatch, an unfinished barrier or a status flag pins the store to the five-launch form."""
    store.tail_checked = True
    cfg, dev, adt = store.cfg, store.device, store.act_dtype
    D, F = cfg.e_model, 4 * cfg.e_model
    if not (o.can_row_tail(B, D) and store.tail_fused and cfg.e_layers >= 1):
        return True
    pre = f"encoder.layer{cfg.e_layers - 1}"
    f32 = dict(dtype=torch.float32, device=dev)

    def rnd(rows, width, site, scale=1.0, relu=False):
        a = torch.zeros(rows, width, **f32)
        o.randn(a, seed=0x7A11, site=site)
        t = torch.zeros(rows, width, dtype=adt, device=dev)
        o.cast_to_act(a, t)  # (values ~ N(0, 1); `scale` only documents intent: LayerNorm makes the chain scale-free)
        return t

    row0 = lambda t: t.view(B, S, -1)[:, 0, :]
    att, xin = rnd(B * S, D, 1), rnd(B * S, D, 2)
    Wp, W1, W2 = store.h(f"{pre}.att.W_proj.weight"), store.h(f"{pre}.ff1.weight"), store.h(f"{pre}.ff2.weight")
    par = {k: store.p(f"{pre}.{k}") for k in ("att.W_proj.bias", "ln1.gamma", "ln1.beta", "ff1.bias", "ff2.bias", "ln2.gamma", "ln2.beta")}
    status = torch.zeros(2, dtype=torch.int32, device=dev)
    sync = torch.zeros(8, dtype=torch.int32, device=dev)

    def fbufs():
        z = lambda w: torch.zeros(B * S, w, dtype=adt, device=dev)
        return dict(h1=z(D), x1=z(D), a=z(F), h2=z(D), x2=z(D), m1=torch.zeros(B * S, **f32), r1=torch.zeros(B * S, **f32),
                    m2=torch.zeros(B * S, **f32), r2=torch.zeros(B * S, **f32))

    u, f, rows = fbufs(), fbufs(), (1, S, 0)
    o.gemm_nt(row0(att), Wp, u["h1"], M=B, N=D, K=D, bias=par["att.W_proj.bias"], resid=row0(xin), c_remap=rows)
    o.layernorm_fwd(row0(u["h1"]), par["ln1.gamma"], par["ln1.beta"], row0(u["x1"]), u["m1"], u["r1"], D=D, M=B, row_id_stride=S)
    o.gemm_nt(row0(u["x1"]), W1, u["a"], M=B, K=D, bias=par["ff1.bias"], act=o.ACT_RELU, c_remap=rows)
    o.gemm_nt(row0(u["a"]), W2, u["h2"], M=B, K=F, bias=par["ff2.bias"], resid=row0(u["x1"]), c_remap=rows)
    o.layernorm_fwd(row0(u["h2"]), par["ln2.gamma"], par["ln2.beta"], row0(u["x2"]), u["m2"], u["r2"], D=D, M=B, row_id_stride=S)
    o.row_tail_fwd(row0(att), row0(xin), Wp, par["att.W_proj.bias"], par["ln1.gamma"], par["ln1.beta"], W1, par["ff1.bias"], W2,
                   par["ff2.bias"], par["ln2.gamma"], par["ln2.beta"], row0(f["h1"]), row0(f["x1"]), row0(f["a"]), row0(f["h2"]),
                   row0(f["x2"]), f["m1"], f["r1"], f["m2"], f["r2"], sync[0:3], stat_stride=S, phys_stride=S, status=status[0:1])
    # backward chain on the forward's own activations
    dy = rnd(B * S, D, 3)
    W2t, W1t, Wpt = store.t(f"{pre}.ff2.weight"), store.t(f"{pre}.ff1.weight"), store.t(f"{pre}.att.W_proj.weight")

    def bbufs():
        z = lambda n, w: torch.zeros(n, w, dtype=adt, device=dev)
        return dict(dh=z(B, D), dhm=z(B, D), dx1=z(B, D), dh1m=z(B, D), dpre=z(B, F), dh1=z(B * S, D), datt=z(B * S, D),
                    dg1=torch.zeros(D, **f32), db1=torch.zeros(D, **f32), dg2=torch.zeros(D, **f32), db2=torch.zeros(D, **f32))

    ub, fb = bbufs(), bbufs()
    g1, g2 = par["ln1.gamma"], par["ln2.gamma"]
    o.layernorm_bwd(row0(u["h2"]), g2, u["m2"], u["r2"], row0(dy), ub["dh"], ub["dg2"], ub["db2"], D=D, M=B, row_id_stride=S)
    o.gemm_nt(ub["dh"], W2t, ub["dpre"], N=F, K=D, gate=row0(u["a"]), alpha=1.0)
    o.gemm_nt(ub["dpre"], W1t, ub["dx1"], N=D, K=F, resid=ub["dh"])
    o.layernorm_bwd(row0(u["h1"]), g1, u["m1"], u["r1"], ub["dx1"], row0(ub["dh1"]), ub["dg1"], ub["db1"], D=D, M=B, row_id_stride=S)
    o.gemm_nt(row0(ub["dh1"]), Wpt, ub["datt"], M=B, N=D, K=D, c_remap=(1, S, 0))
    o.row_tail_bwd(row0(dy), row0(u["h2"]), row0(u["h1"]), row0(u["a"]), u["m1"], u["r1"], u["m2"], u["r2"], g1, g2, W2t, W1t, Wpt,
                   fb["dh"], fb["dhm"], fb["dx1"], fb["dh1m"], fb["dpre"], row0(fb["dh1"]), row0(fb["datt"]), fb["dg1"], fb["db1"],
                   fb["dg2"], fb["db2"], sync[4:7], stat_stride=S, phys_stride=S, status=status[0:1])
    torch.cuda.current_stream().synchronize()
    G = D // 16
    sy, stv = sync.cpu().tolist(), status.cpu().tolist()
    why = []
    if stv[0]:
        why.append(f"status flags {stv[0]:#x}")
    if sy[0] != 3 * G or sy[4] != 2 * G:
        why.append(f"barrier counters {sy[0]} / {sy[4]} instead of {3 * G} / {2 * G}")
    ulp = 2.0 ** -7 if adt == torch.bfloat16 else 2.0 ** -10
    host = lambda t: t.float().cpu().numpy()
    for k in ("h1", "x1", "a", "h2", "x2"):
        a, b = host(row0(f[k])), host(row0(u[k]))
        if not np.all(np.abs(a - b) <= 4 * ulp * np.maximum(np.abs(b), 1.0)):
            why.append(f"forward {k}: max difference {np.abs(a - b).max():.3g}")
    for k in ("dh", "dpre", "dx1", "dh1", "datt"):
        a, b = host(fb[k]), host(ub[k])
        if not np.abs(a - b).max() <= 4 * ulp * max(float(np.abs(b).max()), 1e-6) + 1e-6:
            why.append(f"backward {k}: max difference {np.abs(a - b).max():.3g} (scale {np.abs(b).max():.3g})")
    if why:
        store.tail_fused = False
        warnings.warn("the one-launch position-0 tail failed its start-up check on this device (" + "; ".join(why) +
                      "): using the five-launch form", RuntimeWarning)
    return not why


class StepPlan:
    """All buffers and the kernel sequence of one training step at a fixed (B, T)."""

    def __init__(self, store, B, T, lr=3e-4, clip_gradient=1.0, kl_weight=1.0, label_smoothing=0.0,
                 negative_label_downscaling=False, global_batch=None, gscale=None, want_probs=False, seed=0,
                 internal_eps=False, optimizer_params=None, sample_offset=0, site_base=0):
        """sample_offset: index of this plan's first sample in the global batch (data parallel: rank * B) — the in-graph eps
        is drawn per GLOBAL sample index, so the result does not depend on the sharding.
        site_base: added to every dropout site id (data parallel: a different value per rank gives every rank its own
        masks under the common step seed)."""
        cfg = store.cfg
        self.store, self.cfg, self.B, self.T = store, cfg, B, T
        self.dev, self.adt = store.device, store.act_dtype
        self.lr, self.clip, self.kl_weight = lr, clip_gradient, kl_weight
        self.ls, self.nld = label_smoothing, negative_label_downscaling
        self.global_batch = global_batch or B
        self.sample_offset, self.site_base = int(sample_offset), int(site_base)
        if (self.sample_offset * cfg.latent_dim) % 2:
            raise ValueError("sample_offset * latent_dim must be even (eps is drawn in Box-Muller pairs)")
        # fp16 needs loss scaling for the 1/(T*P)-sized reconstruction gradients (decoder side); bf16 does
        # not. The encoder side is fed by the KL term, whose sigma - 1/sigma gradient is huge near sigma = 0
        # (loss.py:9 has no epsilon), so it keeps scale 1: the two halves of the flat bucket carry their own
        # scale and Adam un-scales each range.
        self.gscale = gscale if gscale is not None else (1024.0 if self.adt == torch.float16 else 1.0)
        self.gscale_enc = 1.0
        self.want_probs, self.internal_eps = want_probs, internal_eps
        self.opt = dict(beta1=0.9, beta2=0.999, eps=1e-8, wd=0.0)
        if optimizer_params:
            for k_src, k_dst in (("beta1", "beta1"), ("beta2", "beta2"), ("epsilon", "eps"), ("wd", "wd")):
                if k_src in optimizer_params:
                    self.opt[k_dst] = float(optimizer_params[k_src])
        dev, adt = self.dev, self.adt
        De, Dd, Z = cfg.e_model, cfg.d_model, cfg.latent_dim
        Se, Sd = T, T + 1
        self.Me, self.Md = B * Se, B * Sd
        f32 = dict(dtype=torch.float32, device=dev)

        def act(rows, width):
            return torch.zeros(rows, roundup(width, 8), dtype=adt, device=dev)

        # ---- inputs: ONE static device blob (so a batch arrives with a single copy) viewed as typed tensors
        if cfg.kind == "token":
            seg = [("tokens", B * T * 4), ("labels", B * T * 4)]
        else:
            # piano-roll frames stay uint8 in HBM, exactly as the batcher delivers them (1 byte per pitch; rows padded to 8):
            # the embedding GEMMs and their weight gradients widen them while staging tiles into LDS (a_u8)
            # The class-embedding gradient is a column sum of d(x0) per class = onehot(class)^T d(x0): with the one-hot class id
            # of a frame in C extra columns behind its pitches, and the class table behind the embedding table in the flat
            # buffers, it is rows in_dim.. of the encoder embedding's weight-gradient problem — whose 256-row tile has the
            # room — instead of a launch of its own (MST_CLS_WGRAD=0: the group_colsum launch)
            C_ = cfg.num_classes
            self.cls_fold = (os.environ.get("MST_CLS_WGRAD", "1") != "0" and cfg.in_dim % 8 == 0 and
                             roundup(cfg.in_dim + C_, 256) == roundup(cfg.in_dim, 256) and
                             store.offsets["encoder.class2hid.weight"] == store.offsets["encoder.embedding.weight"] + cfg.in_dim * De)
            self.ld_roll = roundup(cfg.in_dim + (C_ if self.cls_fold else 0), 8)
            seg = [("roll", B * T * self.ld_roll), ("labels", B * T * cfg.out_dim)]
        seg += [("seq_lens", B * 4), ("classes", B * 4)]
        self.in_layout, off = {}, 0
        for name, nbytes in seg:
            self.in_layout[name] = (off, nbytes)
            off = roundup(off + nbytes, 16)
        self.own_inbuf = torch.zeros(off, dtype=torch.uint8, device=dev)
        self.bind_inputs(self.own_inbuf)
        self.eps = torch.zeros(B, Z, **f32)
        self.rng_state = store.rng_state(seed)
        self.defer_latent_grads = os.environ.get("MST_DEFER_LATENT", "1") != "0"
        self._outers = []

        self.pos_e = torch.from_numpy(positional_table(De, Se)).to(dev)
        self.pos_d = torch.from_numpy(positional_table(Dd, Sd)).to(dev)
        self.keymask_e = torch.zeros(B, Se, dtype=torch.uint8, device=dev)
        self.keymask_d = torch.zeros(B, Sd, dtype=torch.uint8, device=dev)

        def layers(n, M, D, H, S):
            out = []
            for _ in range(n):
                L = _Layer()
                L.qkv, L.att, L.h1, L.x1 = act(M, 3 * D), act(M, D), act(M, D), act(M, D)
                L.a, L.h2, L.x2 = act(M, 4 * D), act(M, D), act(M, D)
                L.lse = torch.zeros(2, B, H, S, **f32)
                L.mean1, L.rstd1 = torch.zeros(M, **f32), torch.zeros(M, **f32)
                L.mean2, L.rstd2 = torch.zeros(M, **f32), torch.zeros(M, **f32)
                out.append(L)
            return out

        self.x0_e = act(self.Me, De)
        self.enc = layers(cfg.e_layers, self.Me, De, cfg.e_heads, Se)
        self.x0_d = act(self.Md, Dd)
        self.dec = layers(cfg.d_layers, self.Md, Dd, cfg.d_heads, Sd)
        self.mu, self.sigma, self.z = torch.zeros(B, Z, **f32), torch.zeros(B, Z, **f32), torch.zeros(B, Z, **f32)
        self.kl, self.total = torch.zeros(B, **f32), torch.zeros(B, **f32)
        # per-sample reconstruction sums: accumulated with atomics, cleared by the step's first launch (size padded to 16 B);
        # the same zero list clears the grid-barrier words of the one-launch position-0 tails (mst_row_tail_*)
        nb4 = (B + 3) // 4 * 4
        # (+ 8 sync words of the position-0 tail launches, zeroed with it by step_begin; + the two tile queues of the tails' riders,
        # each in a 128-byte line of its own: next to the barrier words their ticket atomics delayed the chain's barriers)
        self._recon_buf = torch.zeros(nb4 + 8 + 24 + 64, **f32)
        self.recon = self._recon_buf[:B]
        self.sync_words = self._recon_buf[nb4: nb4 + 8].view(torch.int32)
        self.ride_queues = self._recon_buf[nb4 + 32:].view(torch.int32)  # [0]: forward tail's riders, [32]: backward tail's
        self.metric_acc = store.metric_acc  # [sum kl, sum total, count]  (trainer.py:115-116)
        self.track_token_metrics = False  # Trainer: accumulate ppl / acc / topk sums on the device in the CE launch
        # output layer + BCE in one launch when a tile can hold whole rows of pitches of one sample (configs[1]: P 128, T 256)
        fb = os.environ.get("MST_FUSE_BCE", "1")  # "0": never, "all": wherever the launch exists (A/B runs), default: where it pays
        self.fuse_bce = (cfg.kind == "pianoroll" and o.can_fuse_bce(cfg.out_dim, T, negative_label_downscaling) and fb != "0" and
                         (fb == "all" or o.bce_fusion_pays(cfg.out_dim)))
        # the attention output projection + LayerNorm-1 in front of the forward feed-forward launch (two launches less per layer
        # at equal time; its dgrad behind the backward block measured +14 us and was removed): MST_FUSE_PROJ=0 keeps them apart
        self.fuse_tail_bwd = os.environ.get("MST_TAIL_BWD", "1") != "0"  # the top layer's position-0 backward chain in one launch
        self.fuse_proj = os.environ.get("MST_FUSE_PROJ", "1") != "0"
        # The LAST decoder layer's row-wise part (W_proj, LayerNorm-1, feed-forward, LayerNorm-3 and their backward) skips every
        # sample's position-0 row: its output is dropped before the loss (model.py:253), so nothing it computes there is ever
        # read and every gradient there is zero — the buffers' position-0 rows simply stay at the zeros they were allocated with.
        # B x T rows are B T / 64 tiles of the one-workgroup-per-CU feed-forward launches: ONE resident round at configs[1]
        # (256 tiles) where B (T + 1) rows were 257 (measured: forward 22.8 -> 18.9 us, backward 24.6 -> 20.3).
        self.skip_row0 = (cfg.d_layers > 0 and T % 64 == 0 and o.ffn_fusion_pays(Dd, 4 * Dd) and
                          os.environ.get("MST_SKIP_ROW0", "1") != "0")
        # RIDERS on the one-launch position-0 tails (which keep ONE XCD busy for ~26 us each while seven idle): the decoder's first
        # K | Q | V projection of rows 1..T — its input exists since the step's first launch — is computed by the forward tail
        # launch's workgroups on the other XCDs (row 0 by the latent block's launch), and the input gradient of that projection for
        # rows 1..T — which only the decoder embedding's weight gradient reads — by the backward tail's (row 0 inside the latent
        # block's backward launch): two GEMM launches (12 + 10 us at configs[1]) leave the step's dependent chain. Piano-roll ends.
        self.ride = (cfg.kind != "token" and cfg.d_layers >= 1 and cfg.e_layers >= 1 and Dd in (128, 256) and
                     o.can_ride(B * T, 3 * Dd, Dd, T) and o.can_ride(B * T, Dd, 3 * Dd, T) and os.environ.get("MST_TAIL_RIDERS", "1") != "0")
        self._ride_fwd = self._ride_bwd = False
        self.logits = None if self.fuse_bce else act(B * T, cfg.out_dim)
        self.dlogits = act(B * T, cfg.out_dim)
        if cfg.kind == "token":
            self.probs = torch.zeros(B * T, cfg.out_dim, **f32) if want_probs else None
        else:
            self.probs = act(B * T, cfg.out_dim) if want_probs else None
        self.npos = torch.zeros(B, dtype=torch.int32, device=dev)

        # ---- backward temporaries (encoder and decoder sized separately)
        def bwd_bufs(M, D, H, S):
            t = _Layer()
            t.dx_a, t.dx_b = act(M, D), act(M, D)      # gradient w.r.t. a layer's output / input (ping-pong)
            t.dh, t.dhm = act(M, D), act(M, D)          # LN backward output and its dropout-masked copy
            t.dh1, t.dh1m = act(M, D), act(M, D)
            t.dpre, t.dx1 = act(M, 4 * D), act(M, D)
            t.datt, t.dqkv = act(M, D), act(M, 3 * D)
            t.delta = torch.zeros(B, H, S, **f32)
            return t

        # one set per layer: every weight gradient of the step is computed by ONE launch at the end of backward(), so the
        # operands it reads (dh, dpre, dh1, dqkv of each layer) must survive until then. (Running them on a forked
        # stream instead bought nothing: hipGraph on ROCm 7.2 replays fork/join branches back to back on one queue.)
        self.be_l = [bwd_bufs(self.Me, De, cfg.e_heads, Se) for _ in range(cfg.e_layers)]
        self.bd_l = [bwd_bufs(self.Md, Dd, cfg.d_heads, Sd) for _ in range(cfg.d_layers)]
        self.be, self.bd = self.be_l[0], self.bd_l[0]
        self._wgrads, self._psums, self._out_dgrad_done = [], [], False
        self.wgrad_scratch = store.wgrad_scratch()
        # LayerNorm parameter gradients: every LayerNorm-backward workgroup leaves one row of column sums here and ONE
        # launch per flush adds them into the bucket (256 workgroups x one atomic per column on the same 2D addresses
        # serialised for ~5 us per launch: 30 us of the step at configs[1])
        self._ln_part, self._psums = {}, []
        for side, n_l, M, D in (("encoder", cfg.e_layers, self.Me, De), ("decoder", cfg.d_layers, self.Md, Dd)):
            rows = max(o.layernorm_bwd_parts(M, D), o.gemm_nt_ln_parts(M))
            for i in range(n_l):
                for ln in ("ln1", "ln2" if side == "encoder" else "ln3"):
                    self._ln_part[f"{side}.layer{i}.{ln}"] = torch.zeros(rows, 2 * D, **f32)
        self.lat_scratch = torch.zeros(B * (Dd + 2 * Z), **f32)
        # Sparse gradient carriers, never used as ping-pong targets so their untouched rows stay zero:
        #   d_dec_out: d(decoder output) - rows 1..T written by the output-layer dgrad, row 0 always 0 (model.py:253)
        #   d_enc_out: d(encoder output) - only row 0 of each sample written, by latent_bwd (model.py:97)
        self.d_dec_out = act(self.Md, Dd)
        self.d_enc_out = act(self.Me, De)
        # Top encoder layer, backward: the loss reads the encoder only at position 0 (model.py:97), so the gradient
        # entering the last layer is non-zero in B of its B*T rows, and LayerNorm / FFN / W_proj are row-wise: their
        # backward runs on those B rows (strided views), exactly. Only attention mixes rows; it gets its dO through
        # sp_datt and the residual branch through sp_dh1, full-size buffers whose other rows are never written.
        c = _Layer()
        c.dh, c.dhm, c.dx1, c.dh1m = act(B, De), act(B, De), act(B, De), act(B, De)
        c.dpre = act(B, 4 * De)
        self.top = c
        self.sp_dh1 = act(self.Me, De)
        self.sp_datt = act(self.Me, De)
        self.graph = None
        self.graph_late = None
        self.graph_opt = None
        self._tick_adam = False
        self._infer = False
        self._tail_used = dict(fwd=False, bwd=False)  # which one-launch tails the kernel sequence issued last contains
        if o.can_row_tail(B, De) and store.tail_fused and not store.tail_checked:
            row_tail_selfcheck(store)

    # dropout probabilities of the current pass: 0 in inference mode (forward(inference=True)), where Dropout is the identity
    @property
    def e_p(self):
        return 0.0 if self._infer else self.cfg.e_dropout

    @property
    def d_p(self):
        return 0.0 if self._infer else self.cfg.d_dropout

    def _tail_on(self, D):
        return o.can_row_tail(self.B, D) and self.store.tail_fused

    def _guard(self):
        """step guard of the launches that close a step (optimizer / loss_combine): the barrier counters of the one-launch
        tails issued in this step must have reached their final values (G = D / 16 workgroups: 3 barriers forward, 2 backward)"""
        G = self.cfg.e_model // 16
        exp = []
        if self._tail_used["fwd"]:
            exp.append((self.sync_words[0:1], 3 * G))
        if self._tail_used["bwd"]:
            exp.append((self.sync_words[4:5], 2 * G))
        g = dict(status=self.store.step_status, expect=exp)
        if self.global_batch == self.B:
            # one rank: the optimizer also skips a step whose loss is not finite (its gradients are NaN: an update would destroy the
            # model). Not with more ranks: the loss is per rank, the all-reduced gradient is not — ranks would part ways.
            g["finite"] = (self.recon, self.kl)
        return g

    # ------------------------------------------------------------------------------ inputs
    def bind_inputs(self, buf):
        """Make `buf` (a device uint8 blob with pack_batch()'s layout) the step's input buffer: the typed views the
        kernels read are re-pointed, nothing is copied. A graph captured afterwards reads THAT buffer — a batcher that
        fills two or more such buffers in turn (or bench.py's resident batches) needs no device-to-device hop."""
        cfg, B, T = self.cfg, self.B, self.T
        need = max(a + n for a, n in self.in_layout.values())
        if not (buf.dtype == torch.uint8 and buf.is_contiguous() and buf.is_cuda and buf.data_ptr() % 16 == 0 and buf.numel() >= need):
            raise ValueError("bind_inputs: need a contiguous, 16-byte aligned device uint8 blob of pack_batch()'s size")
        self.inbuf = buf

        def inview(name, dtype, *shape):
            a, n = self.in_layout[name]
            return buf[a: a + n].view(dtype).view(*shape)

        if cfg.kind == "token":
            self.tokens = inview("tokens", torch.int32, B, T)
            self.labels = inview("labels", torch.int32, B, T)
        else:
            self.roll_cls = inview("roll", torch.uint8, B * T, self.ld_roll)  # [pitches | one-hot class | 0]
            self.roll = self.roll_cls[:, : roundup(cfg.in_dim, 8)]
            self.labels = inview("labels", torch.uint8, B * T, cfg.out_dim)
        self.seq_lens = inview("seq_lens", torch.int32, B)
        self.classes = inview("classes", torch.int32, B)

    def load_batch(self, x, seq_lens, classes, labels, eps=None):
        """Copy one batch (host or device tensors / numpy arrays) into the static input buffers."""
        def dev(a, dtype):
            t = torch.as_tensor(np.asarray(a)) if not torch.is_tensor(a) else a
            return t.to(device=self.dev, dtype=dtype, non_blocking=True)

        cfg, B, T = self.cfg, self.B, self.T
        if cfg.kind == "token":
            self.tokens.copy_(dev(x, torch.int32).view(B, T))
            self.labels.copy_(dev(labels, torch.int32).view(B, T))
        else:
            self.roll[:, : cfg.in_dim].copy_(dev(x, torch.uint8).view(B * T, cfg.in_dim))
            self.labels.copy_(dev(labels, torch.uint8).view(B * T, cfg.out_dim))
        self.seq_lens.copy_(dev(seq_lens, torch.int32))
        self.classes.copy_(dev(classes, torch.int32))
        if cfg.kind != "token" and self.cls_fold:
            w = self.ld_roll - cfg.in_dim
            onehot = (self.classes.view(B, 1).to(torch.int64) == torch.arange(w, device=self.dev).view(1, w)).to(torch.uint8)
            self.roll_cls.view(B, T, -1)[:, :, cfg.in_dim:].copy_(onehot.view(B, 1, w).expand(B, T, w))
        if eps is not None:
            self.eps.copy_(dev(eps, torch.float32))

    def pack_into(self, blob, x, seq_lens, classes, labels):
        """write one batch into `blob`, a HOST uint8 tensor with the layout of `inbuf` (a persistent page-locked staging
        buffer of PinnedBatchPipeline, or a fresh one from pack_batch); bytes between the segments are left alone.
        Plain numpy copies on the calling thread: a torch copy_ of a 2 MB tensor fans out over the intra-op thread pool,
        whose idle spinning (128 threads on a 16-core share of the GPU box) ran the process into its CPU quota — a stall
        of ~90 ms every few dozen batches."""
        cfg, B, T = self.cfg, self.B, self.T
        assert blob.dtype == torch.uint8 and not blob.is_cuda and blob.numel() >= self.inbuf.numel()
        raw = blob.numpy()

        def seg(name, dtype, *shape):
            a, n = self.in_layout[name]
            return raw[a: a + n].view(dtype).reshape(*shape)

        def as_np(a):
            return a.cpu().numpy() if torch.is_tensor(a) else np.asarray(a)

        if cfg.kind == "token":
            np.copyto(seg("tokens", np.int32, B, T), as_np(x).reshape(B, T), casting="unsafe")
            np.copyto(seg("labels", np.int32, B, T), as_np(labels).reshape(B, T), casting="unsafe")
        else:
            ldp = self.ld_roll
            roll = seg("roll", np.uint8, B * T, ldp)
            np.copyto(roll[:, : cfg.in_dim], as_np(x).reshape(B * T, cfg.in_dim), casting="unsafe")
            if self.cls_fold:
                w = ldp - cfg.in_dim
                roll.reshape(B, T, ldp)[:, :, cfg.in_dim:] = (as_np(classes).reshape(B, 1, 1).astype(np.int64) ==
                                                              np.arange(w).reshape(1, 1, w)).astype(np.uint8)
            elif ldp != cfg.in_dim:
                roll[:, cfg.in_dim:] = 0
            np.copyto(seg("labels", np.uint8, B * T, cfg.out_dim), as_np(labels).reshape(B * T, cfg.out_dim), casting="unsafe")
        np.copyto(seg("seq_lens", np.int32, B), as_np(seq_lens).reshape(B), casting="unsafe")
        np.copyto(seg("classes", np.int32, B), as_np(classes).reshape(B), casting="unsafe")
        return blob

    def pack_batch(self, x, seq_lens, classes, labels):
        """a fresh host blob with the layout of `inbuf` (what bench.py keeps resident after one upload)"""
        return self.pack_into(torch.zeros(self.inbuf.numel(), dtype=torch.uint8), x, seq_lens, classes, labels)

    def load_packed(self, blob):
        """one copy (host->device or device->device) of a pack_batch() blob into the step's input buffers"""
        self.inbuf.copy_(blob, non_blocking=True)

    # ------------------------------------------------------------------------------ forward
    def _drop(self, p, site):
        return dict(dropout_p=p, dropout_site=site, dropout_seed_ptr=self.rng_state) if p > 0 else {}

    def _site_e(self, i):
        """first of the three dropout site ids of encoder layer i (attention output, FFN hidden, FFN output)"""
        return self.site_base + 3 * i

    def _site_d(self, i):
        return self.site_base + 3 * (self.cfg.e_layers + i)

    def _top_encoder_layer_fwd(self, i, L, x_in):
        """Last encoder layer: the model reads its output at position 0 only (model.py:97) and everything after the
        attention mix is row-wise, so after the dense K/Q/V projection and the softmax row statistics (which
        normalise over ALL queries) only query 0 is attended and only B rows go through W_proj, LN1, the FFN and LN2.
        The other rows of these buffers are never produced nor read (backward: _top_encoder_layer_bwd)."""
        cfg, st, B, S = self.cfg, self.store, self.B, self.T
        D, H, p, site0 = cfg.e_model, cfg.e_heads, self.e_p, self._site_e(i)
        pre = f"encoder.layer{i}"

        def row0(buf):
            return buf.view(B, S, -1)[:, 0, :]

        # (the K | Q | V projection runs inside the attention launch where the shape allows: mst_attn_qkv_fwd)
        o.attn_qkv_fwd(x_in, st.fused(st.w16, pre, "weight"), st.fused(st.w, pre, "bias"), L.qkv, self.keymask_e, L.lse, L.att, B, S, H,
                       D // H, 0, D, 2 * D, q_limit=1)
        self._tail_used["fwd"] = self._tail_on(D)
        self._ride_fwd = self.ride and self._tail_used["fwd"]
        if self._tail_used["fwd"]:  # W_proj, LN1, FFN1, FFN2, LN2 on the B position-0 rows in one launch (mst_row_tail_fwd)
            rider = None
            if self._ride_fwd:  # the decoder's first K | Q | V projection, rows 1..T of every sample (see __init__: ride)
                Dd, Sd = cfg.d_model, S + 1
                rider = dict(A=self.x0_d, B=st.fused(st.w16, "decoder.layer0", "weight"), C_out=self.dec[0].qkv, M=B * S, N=3 * Dd, K=Dd,
                             bias=st.fused(st.w, "decoder.layer0", "bias"), a_remap=(S, Sd, 1), c_remap=(S, Sd, 1))
            o.row_tail_fwd(row0(L.att), row0(x_in), st.h(f"{pre}.att.W_proj.weight"), st.p(f"{pre}.att.W_proj.bias"),
                           st.p(f"{pre}.ln1.gamma"), st.p(f"{pre}.ln1.beta"), st.h(f"{pre}.ff1.weight"), st.p(f"{pre}.ff1.bias"),
                           st.h(f"{pre}.ff2.weight"), st.p(f"{pre}.ff2.bias"), st.p(f"{pre}.ln2.gamma"), st.p(f"{pre}.ln2.beta"),
                           row0(L.h1), row0(L.x1), row0(L.a), row0(L.h2), row0(L.x2), L.mean1, L.rstd1, L.mean2, L.rstd2,
                           self.sync_words[0:3], stat_stride=S, phys_stride=S, dropout_p=p,
                           dropout_seed_ptr=self.rng_state if p > 0 else None, site0=site0, status=st.step_status[0:1], rider=rider,
                           queue=self.ride_queues[0:1], shadows=self._tail_shadows if rider is not None else None)
            assert self._tail_shadows is None or rider is not None, "the shadow refresh was planned onto a tail without riders"
            return L.x2
        rows = (1, S, 0)  # output row b -> physical row b*S
        o.gemm_nt(row0(L.att), st.h(f"{pre}.att.W_proj.weight"), L.h1, M=B, N=D, K=D, bias=st.p(f"{pre}.att.W_proj.bias"),
                  resid=row0(x_in), c_remap=rows, **self._drop(p, site0))
        o.layernorm_fwd(row0(L.h1), st.p(f"{pre}.ln1.gamma"), st.p(f"{pre}.ln1.beta"), row0(L.x1), L.mean1, L.rstd1, D=D, M=B,
                        row_id_stride=S)
        o.gemm_nt(row0(L.x1), st.h(f"{pre}.ff1.weight"), L.a, M=B, K=D, bias=st.p(f"{pre}.ff1.bias"), act=o.ACT_RELU,
                  c_remap=rows, **self._drop(p, site0 + 1))
        o.gemm_nt(row0(L.a), st.h(f"{pre}.ff2.weight"), L.h2, M=B, K=4 * D, bias=st.p(f"{pre}.ff2.bias"), resid=row0(L.x1),
                  c_remap=rows, **self._drop(p, site0 + 2))
        o.layernorm_fwd(row0(L.h2), st.p(f"{pre}.ln2.gamma"), st.p(f"{pre}.ln2.beta"), row0(L.x2), L.mean2, L.rstd2, D=D, M=B,
                        row_id_stride=S)
        return L.x2

    def _layer_fwd(self, side, i, L, x_in, keymask, D, H, S, p, site0):
        st = self.store
        pre = f"{side}.layer{i}"
        dh = D // H
        if side == "encoder" and i == self.cfg.e_layers - 1:
            return self._top_encoder_layer_fwd(i, L, x_in)
        if side == "decoder" and i == 0 and self._ride_fwd:  # (projected by the forward tail's riders + the latent block's launch)
            o.attn_fwd(L.qkv, keymask, L.lse, L.att, self.B, S, H, dh, 0, D, 2 * D)
        else:
            o.attn_qkv_fwd(x_in, st.fused(st.w16, pre, "weight"), st.fused(st.w, pre, "bias"), L.qkv, keymask, L.lse, L.att, self.B, S, H, dh,
                           0, D, 2 * D)
        # (Dense + LayerNorm in one launch, ops.gemm_nt_ln_fwd, does not pay in the forward pass: graph-replay timings at
        # M = 16384 are 17.8 vs 19.9 us for N 256 K 256 but 30.3 vs 30.2 for K 1024 and 16.6 vs 13.0 / 21.3 vs 16.5 for
        # N 128, and nothing at step level — the forward LayerNorm is a 7 us launch and the full-row tile costs the GEMM
        # as much. The backward forms, where the LayerNorm launch is 16 us, do pay: _layer_bwd.)
        proj = dict(N=D, K=D, bias=st.p(f"{pre}.att.W_proj.bias"), resid=x_in, **self._drop(p, site0))
        fused = o.ffn_fusion_pays(D, 4 * D)
        if not (fused and self.fuse_proj):
            o.gemm_nt(L.att, st.h(f"{pre}.att.W_proj.weight"), L.h1, **proj)
            o.layernorm_fwd(L.h1, st.p(f"{pre}.ln1.gamma"), st.p(f"{pre}.ln1.beta"), L.x1, L.mean1, L.rstd1, D=D)
        ff1 = dict(K=D, bias=st.p(f"{pre}.ff1.bias"), act=o.ACT_RELU, **self._drop(p, site0 + 1))
        # encoder: LN2(x1 + dropout(ff)); decoder (transformer.py:199-200): LN3(ff + dropout(ff))
        ln = "ln2" if side == "encoder" else "ln3"
        ff2 = dict(K=4 * D, bias=st.p(f"{pre}.ff2.bias"), **self._drop(p, site0 + 2))
        ff2.update(dict(resid=L.x1) if side == "encoder" else dict(self_resid=True))
        if fused:  # the whole feed-forward block + LayerNorm in one launch (same results, bit for bit in a / h2)
            head = None
            if self.fuse_proj:  # ... and the attention output projection + LayerNorm-1 in front of it (mst_proj_ffn_ln_fwd)
                head = dict(att=L.att, W=st.h(f"{pre}.att.W_proj.weight"), h1=L.h1, gamma=st.p(f"{pre}.ln1.gamma"),
                            beta=st.p(f"{pre}.ln1.beta"), mean=L.mean1, rstd=L.rstd1, **proj)
            o.ffn_ln_fwd(L.x1, st.h(f"{pre}.ff1.weight"), L.a, st.h(f"{pre}.ff2.weight"), L.h2, st.p(f"{pre}.{ln}.gamma"),
                         st.p(f"{pre}.{ln}.beta"), L.x2, L.mean2, L.rstd2, ff1=ff1, ff2=ff2, proj=head,
                         row_groups=self._row0_groups(side, i))
            return L.x2
        o.gemm_nt(L.x1, st.h(f"{pre}.ff1.weight"), L.a, **ff1)
        o.gemm_nt(L.a, st.h(f"{pre}.ff2.weight"), L.h2, **ff2)
        o.layernorm_fwd(L.h2, st.p(f"{pre}.{ln}.gamma"), st.p(f"{pre}.{ln}.beta"), L.x2, L.mean2, L.rstd2, D=D)
        return L.x2

    def _ride_bwd_planned(self):
        """the backward tail will run as one launch in this step and can take a rider (decided before it is issued)"""
        return self.ride and self._tail_on(self.cfg.e_model) and self.fuse_tail_bwd and self.defer_latent_grads

    def _row0_groups(self, side, i):
        """row groups of the last decoder layer's row-wise launches (skip_row0): rows 1..T of every T + 1, else None"""
        if side == "decoder" and i == self.cfg.d_layers - 1 and self.skip_row0:
            return (self.T, self.T + 1, 1)
        return None

    def forward(self, inference=False):
        """inference=True: the forward pass as the reference runs it OUTSIDE autograd.record() (Model(...) called directly, the
        samplers: sampler.py:146-148) — every Dropout is the identity and the training RNG stream is left alone (an eps the
        caller did not supply is drawn from the store's inference stream)."""
        cfg, st, B, T = self.cfg, self.store, self.B, self.T
        self._infer = bool(inference)
        self._tail_used = dict(fwd=False, bwd=False)
        De, Dd = cfg.e_model, cfg.d_model
        Se, Sd = T, T + 1
        sq_e, sq_d = math.sqrt(float(De)), math.sqrt(float(Dd))
        self._wgrads, self._psums, self._outers, self._out_dgrad_done = [], [], [], False  # deferred gradient work of this step
        # one bookkeeping launch: RNG seed of this step, Adam's step count / lr_t, eps, both padding masks
        need_rng = self.e_p > 0 or self.d_p > 0 or self.internal_eps
        rng = st.rng_state_inference() if self._infer else self.rng_state
        tick = self._tick_adam and not self._infer
        begin = dict(rng_state=rng if need_rng else None,
                     adam_state=st.step_state if tick else None, lr=self.lr, beta1=self.opt["beta1"],
                     beta2=self.opt["beta2"], eps_out=self.eps if self.internal_eps else None,
                     eps_index0=self.sample_offset * cfg.latent_dim, lens=self.seq_lens,
                     mask_e=self.keymask_e if cfg.kind != "token" else None, add_e=0, mask_d=self.keymask_d, add_d=1,
                     zero_a=self._recon_buf, zero_b=st.g if tick else None)
        self._tail_shadows = None
        if st.shadows_deferred and cfg.kind != "token" and os.environ.get("MST_BEGIN_RIDE", "1") != "0":
            late = dict(w=st.w, wt16=st.wt16, desc=st.t_desc_late, prefix=st.t_prefix_late, n_mat=st.t_n_late, tiles=st.t_tiles_late)
            # the refresh of the transposed shadows (read by the backward pass only): behind the forward tail's riders where that launch
            # has them — compute units that idle until the position-0 chain ends — else behind the tiles of the step's first launch (+4.9 us)
            if self.ride and self._tail_on(cfg.e_model) and cfg.e_layers >= 1 and os.environ.get("MST_SHADOW_TAIL", "1") != "0":
                self._tail_shadows = late
            else:
                begin["shadows"] = late
        # (piano-roll ends: nothing in the embedding GEMMs reads what the bookkeeping writes — it rides on their launch)
        ride = cfg.kind != "token" and os.environ.get("MST_BEGIN_RIDE", "1") != "0"
        if not ride:
            o.step_begin(**begin)
        # ---- encoder input (model.py:81-91, transformer.py:270)
        if cfg.kind == "token":
            o.embed_fwd(self.tokens, st.p("encoder.embedding.weight"), self.pos_e, self.x0_e.view(B, Se, -1), 0, sq_e,
                        classes=self.classes, cls_table=st.p("encoder.class2hid.weight"), keymask=self.keymask_e)
        else:
            # both ends' embedding GEMMs read the same frames: one launch (the decoder's rows 1..T; its row 0 is latent_fwd's)
            o.gemm_nt_pair(dict(A=self.roll, B=st.t("encoder.embedding.weight"), C_out=self.x0_e, N=De, alpha=sq_e,
                                grpadd=st.p("encoder.class2hid.weight"), grp_index=self.classes, rowadd=self.pos_e, rowadd_period=T),
                           dict(A=self.roll, B=st.t("decoder.embedding.weight"), C_out=self.x0_d, M=B * T, N=Dd, alpha=sq_d,
                                rowadd=self.pos_d[1:], rowadd_period=T, c_remap=(T, Sd, 1)),
                           begin=begin if ride else None)
        x = self.x0_e
        for i, L in enumerate(self.enc):
            x = self._layer_fwd("encoder", i, L, x, self.keymask_e, De, cfg.e_heads, Se, self.e_p, self._site_e(i))
        self.enc_out = x
        # ---- latent block + decoder position 0 (model.py:97-103,292,229-232)
        lat = (x.view(B, Se, -1), st.p("encoder.latent_proj.weight"), st.p("encoder.latent_proj.bias"), self.eps,
               st.p("decoder.latent2hid.weight"), st.p("decoder.latent2hid.bias"), self.classes,
               st.p("decoder.class2hid.weight"), self.pos_d, sq_d, self.mu, self.sigma, self.z, self.kl,
               self.x0_d.view(B, Sd, -1))
        # (the decoder's first K | Q | V projection riding on this launch — rows 1..T exist since the step's first launch — was
        # built and measured at parity: 16-wave workgroups make poor GEMM tiles at K = 128; removed, docs/kernel_notes.md)
        proj0 = None
        if self._ride_fwd:  # ... and position 0 of that projection, on the launch that produces the row
            proj0 = (st.fused(st.w16, "decoder.layer0", "weight"), st.fused(st.w, "decoder.layer0", "bias"), self.dec[0].qkv.view(B, Sd, -1))
        o.latent_fwd(*lat, proj=proj0)
        # ---- decoder positions 1..T (model.py:241-245, transformer.py:237)
        if cfg.kind == "token":
            o.embed_fwd(self.tokens, st.p("decoder.embedding.weight"), self.pos_d, self.x0_d.view(B, Sd, -1), 1, sq_d)
        x = self.x0_d
        site_d = self._site_d(0)
        for i, L in enumerate(self.dec):
            x = self._layer_fwd("decoder", i, L, x, self.keymask_d, Dd, cfg.d_heads, Sd, self.d_p, site_d + 3 * i)
        self.dec_out = x
        # ---- output layer on positions 1..T (model.py:253-256); with a whole row of pitches per tile it runs inside the loss
        # launch (losses(): mst_gemm_sigmoid_bce) and the logits never reach HBM
        if not self.fuse_bce:
            o.gemm_nt(x, st.h("decoder.output_layer.weight"), self.logits, M=B * T, K=Dd, bias=st.p("decoder.output_layer.bias"),
                      a_remap=(T, Sd, 1))

    def losses(self, with_grad=True, combine=True):
        """combine=False: the total loss / running metric sums are left to optimizer() (they ride on the Adam launch)"""
        cfg, B, T = self.cfg, self.B, self.T
        dl = self.dlogits if with_grad else None
        if cfg.kind == "token":
            o.softmax_ce(self.logits, self.labels, self.recon, B, T, cfg.out_dim, probs=self.probs, dlogits=dl,
                         gscale=self.gscale, pre_zeroed=True,
                         tok_parts=self.store.tok_parts if self.track_token_metrics else None)
        elif self.fuse_bce:
            dgrad = None
            if with_grad and o.ln_bwd_fusion_pays(cfg.d_model) and os.environ.get("MST_BCE_DGRAD", "1") != "0":
                # the first launch of the backward pass — the output layer's input gradient + the last decoder layer's LayerNorm-3
                # backward (backward_early) — consumes exactly the logit-gradient tile this launch produces: same workgroup
                last, Dd, Sd = cfg.d_layers - 1, cfg.d_model, T + 1
                dgrad = dict(A=self.dlogits, B=self.store.t("decoder.output_layer.weight"), dX_out=self.bd_l[last].dh, M=B * T, N=Dd,
                             K=self.dlogits.shape[1], c_remap=(T, Sd, 1),
                             **self._out_ln_bwd("decoder", last, self.dec[last], Dd, cfg.d_dropout, self._site_d(0) + 3 * last,
                                                self.bd_l[last], B * T))
                self._out_dgrad_done = True
            o.gemm_sigmoid_bce(self.dec_out, self.store.h("decoder.output_layer.weight"), self.labels, self.recon, T, dgrad=dgrad,
                               dlogits=dl, probs=self.probs, label_smoothing=self.ls, downweight=self.nld, gscale=self.gscale, M=B * T,
                               K=cfg.d_model, bias=self.store.p("decoder.output_layer.bias"), a_remap=(T, T + 1, 1))
        else:
            o.sigmoid_bce(self.logits, self.labels, self.recon, B, T, cfg.out_dim, label_smoothing=self.ls,
                          downweight=self.nld, npos=self.npos, probs=self.probs, dlogits=dl, gscale=self.gscale,
                          pre_zeroed=True)
        if combine:
            o.loss_combine(self.recon, self.kl, self.kl_weight, self.total, self.metric_acc, guard=self._guard())

    # ------------------------------------------------------------------------------ backward
    LN_PARTIALS_MIN = 32  # fewer workgroups than this: their atomics are cheaper than a row of partials each

    def _ln_partials(self, site, parts):
        """partials buffer of one LayerNorm-backward launch (None: few workgroups, keep the atomics); registers the
        deferred column sums into dgamma / dbeta, executed by _flush_grads()"""
        if parts < self.LN_PARTIALS_MIN:
            return None
        st, buf = self.store, self._ln_part[site]
        dg, db = st.grad(f"{site}.gamma"), st.grad(f"{site}.beta")
        D = dg.numel()
        if db.data_ptr() == dg.data_ptr() + 4 * D:  # adjacent in the flat bucket: one job
            self._psums.append(o.partial_sum_job(buf, parts, dg, length=2 * D))
        else:
            self._psums += [o.partial_sum_job(buf, parts, dg, length=D), o.partial_sum_job(buf, parts, db, col_off=D, length=D)]
        return buf

    def _flush_grads(self):
        """the weight gradients collected so far in one wgrad launch; the LayerNorm column sums ride on its reduction pass"""
        o.gemm_wgrad_batch(self._wgrads, scratch=self.wgrad_scratch, sums=self._psums, outers=self._outers)
        self.last_wgrad_launch = (self._wgrads, self._psums)  # (bench.py re-launches the step's own wgrad batch to time it)
        self._wgrads, self._psums, self._outers = [], [], []

    def _out_ln_bwd(self, side, i, L, D, p, site0, t, M):
        """The LayerNorm backward a layer's backward pass STARTS with (LN2 of an encoder layer, LN3 of a decoder layer),
        as keyword arguments for ops.gemm_nt_ln_bwd: the GEMM (of M rows) that produces the layer's incoming gradient runs
        it in its epilogue (dX_out = t.dh) when the row width allows, see _layer_bwd(dy_done=...)."""
        st = self.store
        pre = f"{side}.layer{i}"
        ln = "ln2" if side == "encoder" else "ln3"
        kw = dict(x=L.h2, gamma=st.p(f"{pre}.{ln}.gamma"), mean=L.mean2, rstd=L.rstd2, dgamma=st.grad(f"{pre}.{ln}.gamma"),
                  dbeta=st.grad(f"{pre}.{ln}.beta"), partials=self._ln_partials(f"{pre}.{ln}", o.gemm_nt_ln_parts(M)))
        if side == "encoder":
            kw.update(dict(mask_mode=1, dx_masked=t.dhm) if p > 0 else dict(mask_mode=0))
        else:
            kw.update(mask_mode=2)
        if p > 0:
            kw.update(dropout_p=p, dropout_seed_ptr=self.rng_state, dropout_site=site0 + 2)
        return kw

    def _layer_bwd(self, side, i, L, x_in, dy, dx_in, keymask, D, H, S, p, site0, t, dy_done=False, next_ln=None):
        """dy: gradient w.r.t. the layer output x2; writes the gradient w.r.t. x_in into dx_in.
        dy_done: the producer of dy already ran this layer's leading LayerNorm backward (t.dh / t.dhm are filled).
        next_ln: (_out_ln_bwd(...) of the layer below, its scratch): run THAT layer's leading LayerNorm backward in the
        epilogue of this layer's last GEMM instead of writing dx_in."""
        st = self.store
        pre = f"{side}.layer{i}"
        dhd = D // H
        inv_keep = 1.0 / (1.0 - p) if p > 0 else 1.0
        dk = dict(dropout_p=p, dropout_seed_ptr=self.rng_state) if p > 0 else {}
        fuse = o.ln_bwd_fusion_pays(D)
        M = L.h1.shape[0]
        ffn_fused = o.ffn_fusion_pays(D, 4 * D)
        lead = None
        if side == "encoder" and not dy_done and ffn_fused:
            # LayerNorm-2 backward rides in the prologue of the fused feed-forward backward (mst_ffn_ln_bwd_lead)
            lead = dict(dy=dy, x=L.h2, gamma=st.p(f"{pre}.ln2.gamma"), mean=L.mean2, rstd=L.rstd2, dx=t.dh,
                        dgamma=st.grad(f"{pre}.ln2.gamma"), dbeta=st.grad(f"{pre}.ln2.beta"),
                        partials=self._ln_partials(f"{pre}.ln2", o.gemm_nt_ln_parts(M)))
            if p > 0:
                lead.update(dx_masked=t.dhm, dropout_site=site0 + 2, **dk)
            dy_done = True
        if side == "encoder":
            if not dy_done:
                part = self._ln_partials(f"{pre}.ln2", o.layernorm_bwd_parts(M, D))
                if p > 0:
                    o.layernorm_bwd(L.h2, st.p(f"{pre}.ln2.gamma"), L.mean2, L.rstd2, dy, t.dh, st.grad(f"{pre}.ln2.gamma"),
                                    st.grad(f"{pre}.ln2.beta"), D=D, dx_masked=t.dhm, mask_mode=1, dropout_site=site0 + 2,
                                    partials=part, **dk)
                else:
                    o.layernorm_bwd(L.h2, st.p(f"{pre}.ln2.gamma"), L.mean2, L.rstd2, dy, t.dh, st.grad(f"{pre}.ln2.gamma"),
                                    st.grad(f"{pre}.ln2.beta"), D=D, partials=part)
            dff = t.dhm if p > 0 else t.dh
            resid_ff = t.dh
        else:
            if not dy_done:
                o.layernorm_bwd(L.h2, st.p(f"{pre}.ln3.gamma"), L.mean2, L.rstd2, dy, t.dh, st.grad(f"{pre}.ln3.gamma"),
                                st.grad(f"{pre}.ln3.beta"), D=D, mask_mode=2, dropout_site=site0 + 2,
                                partials=self._ln_partials(f"{pre}.ln3", o.layernorm_bwd_parts(M, D)), **dk)
            dff = t.dh
            resid_ff = None
        # FFN: d(pre-relu) = (dff W2) * 1[a > 0] / (1-p)   (a is stored post-dropout, so a > 0 <=> relu on and kept)
        ln1 = dict(dx_masked=t.dh1m, mask_mode=1, dropout_site=site0, **dk) if p > 0 else {}
        if ffn_fused:  # both dgrads of the block + LayerNorm-1 backward in one launch (mst_ffn_ln_bwd)
            rows = self._row0_groups(side, i)
            o.ffn_ln_bwd(dff, st.t(f"{pre}.ff2.weight"), t.dpre, L.a, st.t(f"{pre}.ff1.weight"), t.dh1, L.h1, st.p(f"{pre}.ln1.gamma"),
                         L.mean1, L.rstd1, st.grad(f"{pre}.ln1.gamma"), st.grad(f"{pre}.ln1.beta"), alpha=inv_keep, resid=resid_ff,
                         partials=self._ln_partials(f"{pre}.ln1", o.gemm_nt_ln_parts(self.B * self.T if rows else M)), lead=lead,
                         row_groups=rows, **ln1)
        else:
            o.gemm_nt(dff, st.t(f"{pre}.ff2.weight"), t.dpre, N=4 * D, K=D, gate=L.a, alpha=inv_keep)
            if fuse:  # FFN1 dgrad + LayerNorm-1 backward in one launch (the gradient in between is never stored)
                o.gemm_nt_ln_bwd(t.dpre, st.t(f"{pre}.ff1.weight"), t.dh1, L.h1, st.p(f"{pre}.ln1.gamma"), L.mean1, L.rstd1,
                                 st.grad(f"{pre}.ln1.gamma"), st.grad(f"{pre}.ln1.beta"), N=D, K=4 * D, resid=resid_ff,
                                 partials=self._ln_partials(f"{pre}.ln1", o.gemm_nt_ln_parts(M)), **ln1)
            else:
                o.gemm_nt(t.dpre, st.t(f"{pre}.ff1.weight"), t.dx1, N=D, K=4 * D, resid=resid_ff)
                o.layernorm_bwd(L.h1, st.p(f"{pre}.ln1.gamma"), L.mean1, L.rstd1, t.dx1, t.dh1, st.grad(f"{pre}.ln1.gamma"),
                                st.grad(f"{pre}.ln1.beta"), D=D, partials=self._ln_partials(f"{pre}.ln1", o.layernorm_bwd_parts(M, D)),
                                **ln1)
        dproj = t.dh1m if p > 0 else t.dh1
        o.gemm_nt(dproj, st.t(f"{pre}.att.W_proj.weight"), t.datt, N=D, K=D)
        o.attn_bwd(L.qkv, keymask, L.lse, t.datt, t.dqkv, t.delta, self.B, S, H, dhd, 0, D, 2 * D)
        if next_ln is not None:  # the layer below starts its backward pass with a LayerNorm backward: run it here
            kw, t_below = next_ln
            o.gemm_nt_ln_bwd(t.dqkv, st.t(f"{pre}.att.W_kqv"), t_below.dh, N=D, K=3 * D, resid=t.dh1, **kw)
        elif side == "decoder" and i == 0 and self._ride_bwd_planned():
            # rows 1..T ride on the backward tail's launch (only the decoder embedding's weight gradient reads them), row 0 is
            # computed inside the latent block's backward launch: backward_early / _top_encoder_layer_bwd
            T_ = self.T
            self._bwd_rider = dict(A=t.dqkv, B=st.t(f"{pre}.att.W_kqv"), C_out=dx_in, M=self.B * T_, N=D, K=3 * D, resid=t.dh1, resid_phys=True,
                                   a_remap=(T_, T_ + 1, 1), c_remap=(T_, T_ + 1, 1))
            self._bwd_dx0 = (t.dqkv.view(self.B, S, -1), st.t(f"{pre}.att.W_kqv"), t.dh1.view(self.B, S, -1))
        else:
            o.gemm_nt(t.dqkv, st.t(f"{pre}.att.W_kqv"), dx_in, N=D, K=3 * D, resid=t.dh1)
        # the layer's four weight gradients: deferred to the ONE wgrad launch at the end of backward() (their operands
        # live in this layer's own scratch `t` and in the forward activations, so nothing is overwritten meanwhile)
        self._wgrads += [
            o.wgrad_problem(dff, L.a, st.grad(f"{pre}.ff2.weight"), st.grad(f"{pre}.ff2.bias"), N=D, K=4 * D),
            o.wgrad_problem(t.dpre, L.x1, st.grad(f"{pre}.ff1.weight"), st.grad(f"{pre}.ff1.bias"), N=4 * D, K=D),
            o.wgrad_problem(dproj, L.att, st.grad(f"{pre}.att.W_proj.weight"), st.grad(f"{pre}.att.W_proj.bias"), N=D, K=D),
            o.wgrad_problem(t.dqkv, x_in, st.fused(st.g, pre, "weight"), st.fused(st.g, pre, "bias"), N=3 * D, K=D),
        ]

    def _top_encoder_layer_bwd(self, i, L, x_in, dx_in, t, next_ln=None):
        """_layer_bwd for the LAST encoder layer, on the B rows (position 0 of each sample) that carry gradient."""
        cfg, st, B, S = self.cfg, self.store, self.B, self.T
        D, H, p, site0 = cfg.e_model, cfg.e_heads, cfg.e_dropout, self._site_e(i)
        pre = f"encoder.layer{i}"
        c = self.top
        inv_keep = 1.0 / (1.0 - p) if p > 0 else 1.0
        dk = dict(dropout_p=p, dropout_seed_ptr=self.rng_state) if p > 0 else {}

        def row0(buf):  # [B, ld] view of position 0 of every sample (row stride S*ld)
            return buf.view(B, S, -1)[:, 0, :]

        dy = row0(self.d_enc_out)
        self._tail_used["bwd"] = self._tail_on(D) and self.fuse_tail_bwd
        if self._tail_used["bwd"]:
            # LayerNorm-2 backward, both FFN dgrads, LayerNorm-1 backward and the W_proj dgrad of the B rows in one launch
            o.row_tail_bwd(dy, row0(L.h2), row0(L.h1), row0(L.a), L.mean1, L.rstd1, L.mean2, L.rstd2, st.p(f"{pre}.ln1.gamma"),
                           st.p(f"{pre}.ln2.gamma"), st.t(f"{pre}.ff2.weight"), st.t(f"{pre}.ff1.weight"), st.t(f"{pre}.att.W_proj.weight"),
                           c.dh, c.dhm, c.dx1, c.dh1m, c.dpre, row0(self.sp_dh1), row0(self.sp_datt), st.grad(f"{pre}.ln1.gamma"),
                           st.grad(f"{pre}.ln1.beta"), st.grad(f"{pre}.ln2.gamma"), st.grad(f"{pre}.ln2.beta"), self.sync_words[4:7],
                           stat_stride=S, phys_stride=S, dropout_p=p, dropout_seed_ptr=self.rng_state if p > 0 else None, site0=site0,
                           status=st.step_status[0:1], rider=getattr(self, "_bwd_rider", None), queue=self.ride_queues[32:33])
            dff, dproj = c.dhm, c.dh1m
            return self._top_encoder_layer_bwd_rest(i, L, x_in, dx_in, t, next_ln, dff, dproj)
        if p > 0:
            o.layernorm_bwd(row0(L.h2), st.p(f"{pre}.ln2.gamma"), L.mean2, L.rstd2, dy, c.dh, st.grad(f"{pre}.ln2.gamma"),
                            st.grad(f"{pre}.ln2.beta"), D=D, M=B, row_id_stride=S, dx_masked=c.dhm, mask_mode=1,
                            dropout_site=site0 + 2, **dk)
            dff = c.dhm
        else:
            o.layernorm_bwd(row0(L.h2), st.p(f"{pre}.ln2.gamma"), L.mean2, L.rstd2, dy, c.dh, st.grad(f"{pre}.ln2.gamma"),
                            st.grad(f"{pre}.ln2.beta"), D=D, M=B, row_id_stride=S)
            dff = c.dh
        o.gemm_nt(dff, st.t(f"{pre}.ff2.weight"), c.dpre, N=4 * D, K=D, gate=row0(L.a), alpha=inv_keep)
        o.gemm_nt(c.dpre, st.t(f"{pre}.ff1.weight"), c.dx1, N=D, K=4 * D, resid=c.dh)
        dh1_rows = row0(self.sp_dh1)
        if p > 0:
            o.layernorm_bwd(row0(L.h1), st.p(f"{pre}.ln1.gamma"), L.mean1, L.rstd1, c.dx1, dh1_rows, st.grad(f"{pre}.ln1.gamma"),
                            st.grad(f"{pre}.ln1.beta"), D=D, M=B, row_id_stride=S, dx_masked=c.dh1m, mask_mode=1,
                            dropout_site=site0, **dk)
            dproj = c.dh1m
        else:
            o.layernorm_bwd(row0(L.h1), st.p(f"{pre}.ln1.gamma"), L.mean1, L.rstd1, c.dx1, dh1_rows, st.grad(f"{pre}.ln1.gamma"),
                            st.grad(f"{pre}.ln1.beta"), D=D, M=B, row_id_stride=S)
            dproj = dh1_rows
        # d(attention output): rows b*S of a buffer that is zero elsewhere
        o.gemm_nt(dproj, st.t(f"{pre}.att.W_proj.weight"), self.sp_datt, M=B, N=D, K=D, c_remap=(1, S, 0))
        return self._top_encoder_layer_bwd_rest(i, L, x_in, dx_in, t, next_ln, dff, dproj)

    def _top_encoder_layer_bwd_rest(self, i, L, x_in, dx_in, t, next_ln, dff, dproj):
        """attention backward (dO is zero outside position 0), the K | Q | V dgrad and the layer's deferred weight gradients"""
        cfg, st, B = self.cfg, self.store, self.B
        S = self.T
        D, H = cfg.e_model, cfg.e_heads
        pre = f"encoder.layer{i}"
        c = self.top

        def row0(buf):
            return buf.view(B, S, -1)[:, 0, :]

        o.attn_bwd(L.qkv, self.keymask_e, L.lse, self.sp_datt, t.dqkv, t.delta, B, S, H, D // H, 0, D, 2 * D, q_limit=1)
        if next_ln is not None:  # as in _layer_bwd: the layer below's LayerNorm-2 backward rides on this GEMM
            kw, t_below = next_ln
            o.gemm_nt_ln_bwd(t.dqkv, st.t(f"{pre}.att.W_kqv"), t_below.dh, N=D, K=3 * D, resid=self.sp_dh1, **kw)
        else:
            o.gemm_nt(t.dqkv, st.t(f"{pre}.att.W_kqv"), dx_in, N=D, K=3 * D, resid=self.sp_dh1)
        self._wgrads += [
            o.wgrad_problem(dff, row0(L.a), st.grad(f"{pre}.ff2.weight"), st.grad(f"{pre}.ff2.bias"), M=B, N=D, K=4 * D),
            o.wgrad_problem(c.dpre, row0(L.x1), st.grad(f"{pre}.ff1.weight"), st.grad(f"{pre}.ff1.bias"), M=B, N=4 * D, K=D),
            o.wgrad_problem(dproj, row0(L.att), st.grad(f"{pre}.att.W_proj.weight"), st.grad(f"{pre}.att.W_proj.bias"), M=B, N=D, K=D),
            o.wgrad_problem(t.dqkv, x_in, st.fused(st.g, pre, "weight"), st.fused(st.g, pre, "bias"), N=3 * D, K=D),
        ]

    def backward(self):
        self.backward_early(flush=False)
        self.backward_late()

    def grad_cut(self):
        """Offset that splits the flat gradient bucket by the time its entries are final: [cut, n) — the top encoder
        layer, latent_proj and every decoder tensor — is complete after backward_early(flush=True); [0, cut) — the
        remaining encoder layers, the encoder embedding and class table — after backward_late(). 0 when the encoder has
        a single layer (nothing is early)."""
        cfg, st = self.cfg, self.store
        if cfg.e_layers < 2:
            return 0
        return st.offsets[f"encoder.layer{cfg.e_layers - 1}.att.W_k.weight"]

    def backward_early(self, flush):
        """Output layer, decoder, latent block and the TOP encoder layer. With `flush` the weight gradients collected so
        far get their own wgrad launch, so that the [grad_cut(), n) part of the bucket can be all-reduced while
        backward_late() runs (data parallel); without it they wait for the single launch at the end."""
        cfg, st, B, T = self.cfg, self.store, self.B, self.T
        De, Dd = cfg.e_model, cfg.d_model
        Se, Sd = T, T + 1
        sq_d = math.sqrt(float(Dd))
        # (the gradient bucket was cleared by forward()'s bookkeeping, which also emptied the lists of deferred gradient work)
        # ---- output layer (rows 1..T of the decoder output; row 0 of dx_a stays zero)
        ldv = self.dlogits.shape[1]
        fuse_d, fuse_e = o.ln_bwd_fusion_pays(Dd), o.ln_bwd_fusion_pays(De)
        site_d = self._site_d(0)
        last = cfg.d_layers - 1
        if self._out_dgrad_done:  # (it rode on the loss launch: losses())
            self._out_dgrad_done = False
        elif fuse_d:  # output-layer dgrad + the last decoder layer's LayerNorm-3 backward (rows 1..T; row 0 of dh stays 0)
            o.gemm_nt_ln_bwd(self.dlogits, st.t("decoder.output_layer.weight"), self.bd_l[last].dh, M=B * T, N=Dd, K=ldv,
                             c_remap=(T, Sd, 1),
                             **self._out_ln_bwd("decoder", last, self.dec[last], Dd, cfg.d_dropout, site_d + 3 * last, self.bd_l[last], B * T))
        else:
            o.gemm_nt(self.dlogits, st.t("decoder.output_layer.weight"), self.d_dec_out, M=B * T, N=Dd, K=ldv, c_remap=(T, Sd, 1))
        self._wgrads.append(o.wgrad_problem(self.dlogits, self.dec_out, st.grad("decoder.output_layer.weight"),
                                            st.grad("decoder.output_layer.bias"), M=B * T, N=cfg.out_dim, K=Dd, b_remap=(T, Sd, 1)))
        dy, tgt, nxt = self.d_dec_out, self.bd_l[0].dx_a, self.bd_l[0].dx_b
        self._bwd_rider = self._bwd_dx0 = None
        for i in reversed(range(cfg.d_layers)):
            x_in = self.dec[i - 1].x2 if i > 0 else self.x0_d
            below = (self._out_ln_bwd("decoder", i - 1, self.dec[i - 1], Dd, cfg.d_dropout, site_d + 3 * (i - 1), self.bd_l[i - 1], self.Md),
                     self.bd_l[i - 1]) if (fuse_d and i > 0) else None
            self._layer_bwd("decoder", i, self.dec[i], x_in, dy, tgt, self.keymask_d, Dd, cfg.d_heads, Sd, cfg.d_dropout,
                            site_d + 3 * i, self.bd_l[i], dy_done=fuse_d, next_ln=below)
            dy, tgt, nxt = tgt, nxt, tgt
        d_x0_d = dy  # gradient w.r.t. the decoder input [B, Sd, Dd]
        # ---- decoder input: rows 1..T -> embedding, row 0 -> latent block
        if cfg.kind == "token":
            o.embed_bwd(self.tokens, st.grad("decoder.embedding.weight"), d_x0_d.view(B, Sd, -1), 1, sq_d)
        else:
            self._wgrads.append(o.wgrad_problem(self.roll, d_x0_d, st.grad("decoder.embedding.weight"), M=B * T, N=cfg.out_dim,
                                                K=Dd, scale=sq_d, b_remap=(T, Sd, 1)))
        # gradient w.r.t. the encoder output: zero except position 0 of every sample
        d_enc = self.d_enc_out
        if self.defer_latent_grads:
            # nothing but the optimizer reads the latent block's parameter gradients: they ride on the weight-gradient flush (extra
            # workgroups of its reduction pass) instead of being a launch in the middle of the backward pass's dependent chain
            o.latent_bwd_vec(st.p("encoder.latent_proj.weight"), self.eps, st.p("decoder.latent2hid.weight"), self.classes, self.mu,
                             self.sigma, d_x0_d.view(B, Sd, -1), sq_d, self.kl_weight, self.gscale_enc,
                             st.grad("decoder.class2hid.weight"), d_enc.view(B, Se, -1), self.lat_scratch,
                             enc_scale=self.gscale_enc / self.gscale, proj=self._bwd_dx0)
            self._outers += o.latent_outer_jobs(self.lat_scratch, self.enc_out.view(B, Se, -1), self.z,
                                                st.grad("encoder.latent_proj.weight"), st.grad("encoder.latent_proj.bias"),
                                                st.grad("decoder.latent2hid.weight"), st.grad("decoder.latent2hid.bias"))
        else:
            o.latent_bwd(self.enc_out.view(B, Se, -1), st.p("encoder.latent_proj.weight"), self.eps,
                         st.p("decoder.latent2hid.weight"), self.classes, self.mu, self.sigma, self.z,
                         d_x0_d.view(B, Sd, -1), sq_d, self.kl_weight, self.gscale_enc,
                         st.grad("encoder.latent_proj.weight"), st.grad("encoder.latent_proj.bias"),
                         st.grad("decoder.latent2hid.weight"), st.grad("decoder.latent2hid.bias"),
                         st.grad("decoder.class2hid.weight"), d_enc.view(B, Se, -1), self.lat_scratch,
                         enc_scale=self.gscale_enc / self.gscale)
        top = cfg.e_layers - 1
        x_in = self.enc[top - 1].x2 if top > 0 else self.x0_e
        below = (self._out_ln_bwd("encoder", top - 1, self.enc[top - 1], De, cfg.e_dropout, self._site_e(top - 1), self.be_l[top - 1], self.Me),
                 self.be_l[top - 1]) if (fuse_e and top > 0) else None
        self._top_encoder_layer_bwd(top, self.enc[top], x_in, self.be_l[0].dx_a, self.be_l[top], next_ln=below)
        if flush and cfg.e_layers >= 2:
            self._flush_grads()

    def backward_late(self):
        """The encoder layers below the top one, the encoder input, and the (remaining) weight gradients."""
        cfg, st, B, T = self.cfg, self.store, self.B, self.T
        De, Se = cfg.e_model, T
        sq_e = math.sqrt(float(De))
        dy, tgt, nxt = self.be_l[0].dx_a, self.be_l[0].dx_b, self.be_l[0].dx_a  # the top layer wrote dx_a
        fuse_e = o.ln_bwd_fusion_pays(De)  # then every layer's leading LayerNorm backward already ran in the GEMM above it
        for i in reversed(range(cfg.e_layers - 1)):
            x_in = self.enc[i - 1].x2 if i > 0 else self.x0_e
            below = (self._out_ln_bwd("encoder", i - 1, self.enc[i - 1], De, cfg.e_dropout, self._site_e(i - 1), self.be_l[i - 1], self.Me),
                     self.be_l[i - 1]) if (fuse_e and i > 0) else None
            self._layer_bwd("encoder", i, self.enc[i], x_in, dy, tgt, self.keymask_e, De, cfg.e_heads, Se, cfg.e_dropout,
                            self._site_e(i), self.be_l[i], dy_done=fuse_e, next_ln=below)
            dy, tgt, nxt = tgt, nxt, tgt
        d_x0_e = dy
        if cfg.kind == "token":
            o.embed_bwd(self.tokens, st.grad("encoder.embedding.weight"), d_x0_e.view(B, Se, -1), 0, sq_e,
                        classes=self.classes, dcls=st.grad("encoder.class2hid.weight"))
        else:
            # with cls_fold, rows in_dim.. of this problem are the class table's gradient (model.py:89: one class row per frame)
            self._wgrads.append(o.wgrad_problem(self.roll_cls, d_x0_e, st.grad("encoder.embedding.weight"), M=B * T,
                                                N=cfg.in_dim + (cfg.num_classes if self.cls_fold else 0), K=De, scale=sq_e))
            if not self.cls_fold:
                o.group_colsum(d_x0_e.view(B, Se, -1), T, De, 0, self.classes, st.grad("encoder.class2hid.weight"), sq_e)
        # every (remaining) Dense weight / bias gradient in ONE launch — all 15 problems of the step at configs[1] on a
        # single GPU: one resident round of workgroups with the smallest possible M-split instead of six launches
        self._flush_grads()

    def optimizer(self):
        st = self.store
        clip = self.clip if self.clip is not None else -1.0
        # end-of-step bookkeeping (total loss, running metric sums) on the first Adam launch: losses(combine=False)
        guard = self._guard()
        mt = dict(recon=self.recon, kl=self.kl, kl_weight=self.kl_weight, total=self.total, metric=self.metric_acc, **guard)
        emb = (lambda base: dict(base=base, specs=st.emb_specs, wt16=st.wt16)) if st.shadows_deferred else (lambda base: None)
        if self.gscale == self.gscale_enc:
            o.adam_flat(st.w, st.g, st.m, st.v, st.w16, st.step_state, lr=self.lr,
                        rescale=1.0 / (self.global_batch * self.gscale), clip=clip, advance_step=False, metrics=mt, emb=emb(0), **self.opt)
        else:
            # encoder.* tensors come first in the flat buffers; everything from decoder.latent2hid on is decoder-side.
            # NOTE the latent_proj gradients are produced by latent_bwd at the encoder-side scale.
            cut = st.offsets["decoder.latent2hid.weight"]
            rng = [(0, cut, self.gscale_enc, False), (cut, st.n, self.gscale, False)]
            for a, b, gs, adv in rng:
                o.adam_flat(st.w[a:b], st.g[a:b], st.m[a:b], st.v[a:b], st.w16[a:b], st.step_state, lr=self.lr,
                            rescale=1.0 / (self.global_batch * gs), clip=clip, advance_step=adv,
                            metrics=mt if a == 0 else (guard or None), emb=emb(a), **self.opt)
        if not st.shadows_deferred:  # (deferred: the next step's first launch rebuilds them, forward())
            o.transpose_shadows(st.w, st.wt16, st.t_desc, st.t_prefix, len(st.t_specs), st.t_tiles)

    # ------------------------------------------------------------------------------ step
    def fwd_bwd_kernels(self, is_train=True):
        self._tick_adam = is_train  # the step counter / lr_t are advanced by forward()'s step_begin launch
        self.forward()
        self.losses(with_grad=is_train, combine=not is_train)
        if is_train:
            self.backward()

    def step_kernels(self, is_train=True, reduce_fn=None):
        """One step, eagerly: forward, losses, backward, [gradient all-reduce], Adam + shadow refresh."""
        self.fwd_bwd_kernels(is_train)
        if is_train:
            if reduce_fn is not None:
                reduce_fn(self.store.g)
            self.optimizer()

    def capture(self, is_train=True, split_optimizer=False, overlap=False):
        """Capture the step into hipGraph(s) on the current stream. Run one eager step of this shape
        first (lazy HIP module loads are not capturable). With split_optimizer the optimizer lives in
        its own graph so a gradient all-reduce can run between the two (data parallel). With overlap as well (and an
        encoder of >= 2 layers) the backward pass is cut after the top encoder layer: run(reducer=...) all-reduces the
        early part of the bucket on the communication stream while the rest of the backward pass executes."""
        self.is_train, self.split = is_train, split_optimizer
        self.graph_late = None
        if is_train and split_optimizer and overlap and self.grad_cut() > 0:
            def early():
                self._tick_adam = True
                self.forward()
                self.losses(with_grad=True, combine=False)
                self.backward_early(flush=True)
            self.graph = o.Graph().capture(early)
            self.graph_late = o.Graph().capture(self.backward_late)
            self.graph_opt = o.Graph().capture(self.optimizer)
        elif split_optimizer or not is_train:
            self.graph = o.Graph().capture(lambda: self.fwd_bwd_kernels(is_train))
            self.graph_opt = o.Graph().capture(self.optimizer) if is_train else None
        else:
            self.graph = o.Graph().capture(lambda: self.step_kernels(True))
            self.graph_opt = None
        return self

    def run(self, reduce_fn=None, reducer=None, stamps=None):
        """Replay the captured step. reduce_fn(flat): blocking-in-stream-order all-reduce of the whole bucket between the
        two graphs. reducer (parallel.GradReducer): asynchronous per-range all-reduce, used with capture(overlap=True).
        stamps: a pair of ops.Event recorded on the step's stream behind the last backward graph and in front of the
        optimizer graph — the time between them is the part of the gradient exchange that nothing hides."""
        if self.graph is None:
            raise RuntimeError("call capture() first")
        self.graph.launch()
        if self.graph_late is not None:
            g, cut = self.store.g, self.grad_cut()
            pending = [reducer.start(g[cut:])] if reducer is not None else []
            self.graph_late.launch()  # runs while the early part of the bucket is on the wire
            if stamps is not None:
                stamps[0].record()
            if reducer is not None:
                pending.append(reducer.start(g[:cut]))
                reducer.finish(pending)
            elif reduce_fn is not None:
                reduce_fn(g)
            if stamps is not None:
                stamps[1].record()
            self.graph_opt.launch()
            return
        if self.graph_opt is not None:
            if stamps is not None:
                stamps[0].record()
            if reducer is not None:
                reducer.finish([reducer.start(self.store.g)])
            elif reduce_fn is not None:
                reduce_fn(self.store.g)
            if stamps is not None:
                stamps[1].record()
            self.graph_opt.launch()

    def metrics(self, reset=True):
        """(kl_loss, total_loss) batch means accumulated on the device (trainer.py:115-116,185-186); one sync."""
        m = self.store.read_metrics(reset)
        n = max(m["count"], 1.0)
        return {"kl_loss": m["kl_sum"] / n, "total_loss": m["total_sum"] / n, "count": m["count"], "nonfinite_steps": m["nonfinite_steps"]}
