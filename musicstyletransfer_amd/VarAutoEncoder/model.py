"""Model / config classes with the reference's signatures (VarAutoEncoder/model.py:22-54,275-296), backed by
the MI355X step engine.

    Model(config)(tokens, seq_lens, classes) -> (probs, means, vars)

`tokens` is [B, T] token ids (the reference's path) or [B, T, P] {0,1} piano-roll frames (the piano-roll
ends). Parameters live in one flat HBM buffer (engine.ParamStore); `.encoder` / `.decoder` expose them by
name. The model must be placed on a HIP device before it is called: there is no CPU forward."""
import os
import sys
from collections import OrderedDict

import numpy as np
import torch

from .. import engine as E
from .config import Config
from .transformer import TransformerConfig  # noqa: F401


class LSTMConfig(Config):  # kept for config files that name it (model.py:11-19); the LSTM decoder is dead code there
    def __init__(self, n_layers: int, hidden_dim: int, dropout: float):
        super().__init__()
        self.n_layers, self.hidden_dim, self.dropout = n_layers, hidden_dim, dropout


class DecoderConfig(Config):
    def __init__(self, transformer_config, latent_dim: int, num_classes: int, output_dim: int):
        super().__init__()
        self.transformer_config = transformer_config
        self.latent_dim, self.num_classes, self.output_dim = latent_dim, num_classes, output_dim


class EncoderConfig(Config):
    def __init__(self, transformer_config, latent_dim: int, num_classes: int, input_dim: int):
        super().__init__()
        self.transformer_config = transformer_config
        self.latent_dim, self.num_classes, self.input_dim = latent_dim, num_classes, input_dim


class ModelConfig(Config):
    def __init__(self, encoder_config: EncoderConfig, decoder_config: DecoderConfig, kind: str = "token"):
        super().__init__()
        self.encoder_config, self.decoder_config = encoder_config, decoder_config
        self.kind = kind  # 'token' (reference) or 'pianoroll'

    def to_engine(self):
        e, d = self.encoder_config, self.decoder_config
        te, td = e.transformer_config, d.transformer_config
        assert e.latent_dim == d.latent_dim and e.num_classes == d.num_classes
        return E.VAEConfig(self.kind, e.input_dim, d.output_dim, e.num_classes, e.latent_dim, te.model_size, te.num_layers,
                           te.num_heads, td.model_size, td.num_layers, td.num_heads, te.dropout, td.dropout)


class DecoderState:
    """Inference state (model.py:107-128): the tokens fed so far, the position counter, and the decoder layers' K | Q | V
    caches, which live in HBM inside `plan` (decode.DecodePlan) and are appended to by every decode step."""

    def __init__(self, batch_size: int, num_cache_layers: int, initial_state, plan=None):
        from ..MIDIUtil.defaults import SOS_ID
        self.initial_state = initial_state
        self.plan = plan
        self.num_cache_layers = num_cache_layers
        self.tokens = np.full((batch_size, 1), SOS_ID, np.int64)
        self.t = 1

    def advance_state(self, tokens):
        self.tokens = np.concatenate([self.tokens, np.asarray(tokens).reshape(-1, 1)], axis=1)
        self.t += 1


class _Decoder:
    """the decoder half as the samplers use it (model.py:206-272): parameters by name, get_initial_state, forward_inference"""

    def __init__(self, model):
        self._model = model

    def collect_params(self):
        st = self._model.store
        return {n: st.p(n) for n in st.shapes if n.startswith("decoder.")}

    def initial_rows(self, tokens, seq_lens, classes, beam=1):
        """position 0 of the decoder input for a batch: z = the latent MEANS (sampler.py:146-148), rows repeated `beam` times"""
        m = self._model
        x = np.asarray(tokens.cpu() if torch.is_tensor(tokens) else tokens)
        B, T = x.shape[0], x.shape[1]
        plan = m.plan(B, T, want_probs=False, internal_eps=False)
        cfg = m.engine_config
        dummy = np.zeros((B, T), np.int64) if cfg.kind == "token" else np.zeros((B, T, cfg.out_dim), np.uint8)
        plan.load_batch(x, seq_lens, classes, dummy, np.zeros((B, cfg.latent_dim), np.float32))  # eps = 0: z = means
        plan.forward(inference=True)  # outside autograd.record() in the reference: Dropout is the identity
        row0 = plan.x0_d.view(B, T + 1, -1)[:, 0, :]
        return (row0.repeat_interleave(beam, dim=0) if beam > 1 else row0).contiguous()

    def get_initial_state(self, tokens, seq_lens, classes, t_max, attention="query", beam=1):
        """encode the batch, take z = the latent MEANS (sampler.py:146-148: latent_vector = means), build position 0 of the
        decoder input from it (model.py:229-232) and feed it: returns a DecoderState whose caches hold row 0.
        beam > 1: every sample is repeated `beam` times (beam search's hypotheses, sampler.py:211-213)."""
        from .. import decode
        m = self._model
        x = np.asarray(tokens.cpu() if torch.is_tensor(tokens) else tokens)
        B, T = x.shape[0], x.shape[1]
        plan = m.plan(B, T, want_probs=False, internal_eps=False)
        cfg = m.engine_config
        dummy = np.zeros((B, T), np.int64) if cfg.kind == "token" else np.zeros((B, T, cfg.out_dim), np.uint8)
        plan.load_batch(x, seq_lens, classes, dummy, np.zeros((B, cfg.latent_dim), np.float32))  # eps = 0: z = means
        # outside autograd.record() in the reference (sampler.py:146-148): Dropout is the identity, and the training RNG
        # stream is not advanced by a sampling hook in the middle of training
        plan.forward(inference=True)
        row0 = plan.x0_d.view(B, T + 1, -1)[:, 0, :]
        if beam > 1:
            row0 = row0.repeat_interleave(beam, dim=0)
        dplan = m.decode_plan(B * beam, t_max, attention)
        dplan.start(row0.contiguous())
        return DecoderState(B * beam, cfg.d_layers, row0, dplan)

    def forward_inference(self, state):
        """model.py:259-272: the distribution of the next position given the last token of `state` — [B, V] probabilities"""
        return state.plan.step(state.tokens[:, -1])


class _ParamGroup:
    """name -> fp32 parameter view of one half of the model (what collect_params() gives in the reference)"""

    def __init__(self, model, prefix):
        self._model, self._prefix = model, prefix

    def collect_params(self):
        st = self._model.store
        return {n: st.p(n) for n in st.shapes if n.startswith(self._prefix)}


class Model:
    def __init__(self, config: ModelConfig, *args, **kwargs):
        print("Creating a model with the following configuration:")
        config.output_to_stream(sys.stdout)
        self.config = config
        self.engine_config = config.to_engine()
        self.store = None
        self._plans = OrderedDict()
        self._evict_hooks = []
        self.encoder = _ParamGroup(self, "encoder.")
        self.decoder = _Decoder(self)
        self.act_dtype = torch.bfloat16

    # -- placement / initialisation (model.initialize(mx.init.Xavier(), ctx), trainer.py:103-105)
    def initialize(self, ctx=None, seed=1234, params_np=None, act_dtype=None):
        dev = ctx.device if hasattr(ctx, "device") else (ctx if ctx is not None else torch.device("cuda", 0))
        if torch.device(dev).type != "cuda":
            raise RuntimeError("the VarAutoEncoder step runs on hand-written HIP kernels only: no CPU path exists "
                               "(pass --gpu / a HIP device)")
        if act_dtype is not None:
            self.act_dtype = act_dtype
        self.store = E.ParamStore(self.engine_config, torch.device(dev), self.act_dtype, params_np=params_np, seed=seed)
        self._plans = OrderedDict()
        return self

    def collect_params(self):
        return {n: self.store.p(n) for n in self.store.shapes}

    # Plans are cached per (B, T, hyper): data.py:196-198 truncates every token batch to ITS OWN longest sample, so T
    # changes from batch to batch, and a plan owns every activation / backward buffer of its shape plus its captured
    # graphs. Padding T up to a bucket is not an option for parity — the reference's softmax runs over the query axis and
    # padded keys are not excluded (transformer.py:100,111-125), so extra padded positions change the result — hence a
    # least-recently-used cache with a byte cap instead (MST_PLAN_CACHE_GB, default 64 of the 288 GB).
    PLAN_CACHE_BYTES = int(float(os.environ.get("MST_PLAN_CACHE_GB", "64")) * (1 << 30))
    # ... and a cap on the NUMBER of cached shapes: what hangs off a plan besides its own buffers — the trainer's captured graphs
    # (one per input ring slot and mode), the batcher's ring of page-locked host blobs and their device twins (plan.extra_bytes,
    # reported by PinnedBatchPipeline) — is not all visible as device bytes
    PLAN_CACHE_MAX = int(os.environ.get("MST_PLAN_CACHE_MAX", "48"))

    def plan(self, B, T, **hyper):
        key = (B, T, tuple(sorted(hyper.items())))
        plan = self._plans.pop(key, None)
        if plan is None:
            dev = self.store.device
            before = torch.cuda.memory_allocated(dev)
            plan = E.StepPlan(self.store, B, T, **hyper)
            plan.cache_bytes = max(0, torch.cuda.memory_allocated(dev) - before)
            plan.extra_bytes = 0  # input rings etc. created for this plan later (PinnedBatchPipeline.stage)
            total = plan.cache_bytes + sum(p.cache_bytes + getattr(p, "extra_bytes", 0) for p in self._plans.values())
            while self._plans and (total > self.PLAN_CACHE_BYTES or len(self._plans) >= self.PLAN_CACHE_MAX):
                old_key = next(iter(self._plans))  # least recently used first
                old = self._plans.pop(old_key)
                torch.cuda.synchronize(dev)  # nothing in flight may still read its buffers / graphs
                total -= old.cache_bytes + getattr(old, "extra_bytes", 0)
                for cb in self._evict_hooks:
                    cb(old)
        self._plans[key] = plan  # most recently used last
        return plan

    DECODE_PLAN_MAX = 8

    def decode_plan(self, n_hyp, t_max, attention="query"):
        """the DecodePlan (caches + one captured graph per position) of this many hypotheses / positions, kept across
        batches: a sampler that decodes batch after batch of one shape captures its graphs once"""
        from .. import decode
        if not hasattr(self, "_decode_plans") or getattr(self, "_decode_store", None) is not self.store:
            self._decode_plans, self._decode_store = OrderedDict(), self.store
        key = (n_hyp, t_max, attention)
        plan = self._decode_plans.pop(key, None)
        if plan is None:
            plan = decode.DecodePlan(self.store, n_hyp, t_max, attention=attention)
            while len(self._decode_plans) >= self.DECODE_PLAN_MAX:
                torch.cuda.synchronize(self.store.device)
                self._decode_plans.popitem(last=False)
        plan.reset()
        self._decode_plans[key] = plan
        return plan

    def beam_search_plan(self, B, K, i_max, attention="query"):
        """decode.BeamSearch of this shape (caches, token rows, one captured graph per position), kept across batches"""
        from .. import decode
        if not hasattr(self, "_beam_plans") or getattr(self, "_beam_store", None) is not self.store:
            self._beam_plans, self._beam_store = OrderedDict(), self.store
        key = (B, K, i_max, attention)
        bs = self._beam_plans.pop(key, None)
        if bs is None:
            bs = decode.BeamSearch(self.store, B, K, i_max, attention=attention)
            while len(self._beam_plans) >= self.DECODE_PLAN_MAX:
                torch.cuda.synchronize(self.store.device)
                self._beam_plans.popitem(last=False)
        self._beam_plans[key] = bs
        return bs

    def on_plan_evicted(self, callback):
        """callback(plan) when the cache drops a plan (holders of per-plan state — graphs, input rings — forget it)"""
        self._evict_hooks.append(callback)

    def __call__(self, tokens, seq_lens, classes, eps=None):
        """forward only (model.py:287-296): returns (probs, means, vars) as device tensors. Called directly the reference's
        Model runs outside autograd.record(), i.e. in predict mode: every Dropout is the identity (the training step —
        Trainer._step, under record() — is where e_dropout / d_dropout act); an eps that is not given is drawn from the
        store's inference RNG stream, never the training one."""
        if self.store is None:
            raise RuntimeError("call initialize(ctx) first")
        x = np.asarray(tokens.cpu() if torch.is_tensor(tokens) else tokens)
        B, T = x.shape[0], x.shape[1]
        plan = self.plan(B, T, want_probs=True, internal_eps=eps is None)
        cfg = self.engine_config
        dummy = np.zeros((B, T), np.int64) if cfg.kind == "token" else np.zeros((B, T, cfg.out_dim), np.uint8)
        plan.load_batch(x, seq_lens, classes, dummy, eps)
        plan.forward(inference=True)
        plan.losses(with_grad=False, combine=False)  # (no contribution to the trainer's running metric sums)
        V = cfg.out_dim
        probs = plan.probs[:, :V].float().view(B, T, V)
        return probs, plan.mu, plan.sigma

    # -- checkpoints (utils.save_model / load_model_parameters)
    def save_parameters(self, fname):
        np.savez(fname if fname.endswith(".npz") else fname + ".npz", **self.store.to_numpy("w"))

    def load_parameters(self, fname, ctx=None):
        f = fname if fname.endswith(".npz") else fname + ".npz"
        with np.load(f) as z:
            params = {k: z[k] for k in z.files}
        if self.store is None:
            self.initialize(ctx, params_np=params)
        else:
            self.store.load_numpy(params)
