"""Incremental decoding with per-layer K | Q | V caches in HBM (reference model.py:107-128,259-272,
transformer.py:70-77,242-249, sampler.py:155-257).

The reference's own inference path is inconsistent with its decoder (`forward_inference` signatures do not
match the samplers, `compute_with_cache` never appends because the decoder's self-attention is built with
mask_future_timesteps=False, `mx.nd.concat([..], axis=1)` is a wrong call), so this is the evident intent, restated once in
oracle/vae_oracle.py::decode_incremental and implemented here on the same kernels as the training step:

  * position 0 is the initial state row, sqrt(D) * (latent2hid(z) + class2hid(c)) + pos[0] (model.py:229-232,
    transformer.py:237) — exactly what mst_latent_fwd writes as decoder row 0 in training;
  * position t >= 1 embeds the previous token (or frame) with pos[t] (transformer.py:246) and runs the layers on that ONE row
    per sample; each layer's K | Q | V projection of the row is written straight into its cache at row t by the projection
    GEMM's output row remap (no concat, no copy), and the new query attends to rows 0..t (mst_attn_decode);
  * attention arithmetic is the reference's: the softmax runs over the QUERY axis (transformer.py:100), which in a decode
    step holds the single new query — every weight is 1 and the head output is the sum of the cached value rows
    (mode 'query'); the conventional softmax over the cached keys is offered as mode 'key';
  * dropout layers are identities outside autograd.record() (inference).
Everything is one row per sample: the GEMMs are launch-bound M = B problems of the same mst_gemm_nt used in training, so a
position is ~8 launches per layer of a few microseconds each. Every position is therefore captured ONCE as a hipGraph (its
kernel arguments — the cache row written, the number of keys attended to, the positional row — are host scalars that differ
per position, hence one graph per position, captured the first time a plan reaches it) and replayed from then on; plans are
kept per (hypotheses, positions, attention mode) by the model, so beam search over many batches captures nothing twice."""
import math
import os

import numpy as np
import torch

from . import ops as o
from .engine import positional_table, roundup


class DecodePlan:
    """Buffers and the kernel sequence of incremental decoding for B hypotheses and up to t_max positions."""

    MODES = {"query": 0, "key": 1}

    def __init__(self, store, B, t_max, attention="query", pingpong=False):
        """pingpong: two sets of caches, position t working on set t & 1 — beam search on the device re-ranks the hypotheses
        after every position by gathering the cache rows from one set into the other (BeamSearch below)"""
        cfg = store.cfg
        self.store, self.cfg, self.B, self.t_max = store, cfg, B, t_max
        self.pingpong = pingpong
        self.mode = self.MODES[attention]
        dev, adt = store.device, store.act_dtype
        D = cfg.d_model
        self.pos = torch.from_numpy(positional_table(D, t_max)).to(dev)

        def act(rows, width):
            return torch.zeros(rows, roundup(width, 8), dtype=adt, device=dev)

        self.cache_sets = [[torch.zeros(B, t_max, 3 * D, dtype=adt, device=dev) for _ in range(cfg.d_layers)] for _ in range(2 if pingpong else 1)]
        self.cache = self.cache_sets[0]
        self.x, self.att, self.h1, self.x1, self.h2, self.x2 = (act(B, D) for _ in range(6))
        self.a = act(B, 4 * D)
        self.mean, self.rstd = torch.zeros(B, dtype=torch.float32, device=dev), torch.zeros(B, dtype=torch.float32, device=dev)
        self.mean1, self.rstd1 = torch.zeros(B, dtype=torch.float32, device=dev), torch.zeros(B, dtype=torch.float32, device=dev)
        # W_proj + LN1 + feed-forward + LN3 of a position as ONE launch (the training step's mst_proj_ffn_ln_fwd) instead of
        # five: a decoded position is launch-floor bound (MST_DECODE_FFN=0: the five launches)
        self.fuse_ffn = o.can_fuse_ffn(D, 4 * D) and os.environ.get("MST_DECODE_FFN", "1") != "0"
        self.logits = act(B, cfg.out_dim)
        self.loss = torch.zeros(B, dtype=torch.float32, device=dev)
        self.npos = torch.zeros(B, dtype=torch.int32, device=dev)
        if cfg.kind == "token":
            self.probs = torch.zeros(B, cfg.out_dim, dtype=torch.float32, device=dev)
            self.tokens = torch.zeros(B, 1, dtype=torch.int32, device=dev)
            self.zero_labels = torch.zeros(B, dtype=torch.int32, device=dev)
        else:
            self.probs = act(B, cfg.out_dim)
            self.frames = torch.zeros(B, roundup(cfg.in_dim, 8), dtype=torch.uint8, device=dev)
            self.zero_labels = torch.zeros(B, cfg.out_dim, dtype=torch.uint8, device=dev)
        self.t = -1  # position of the last row fed
        # one captured graph per position (MST_DECODE_GRAPHS=0: eager launches); capture needs a stream of its own
        self.use_graphs = os.environ.get("MST_DECODE_GRAPHS", "1") != "0"
        self.stream = torch.cuda.Stream(device=dev)
        self._graphs = {}
        self._warm = False

    # ------------------------------------------------------------------ one position through the decoder layers
    def _layers(self, x, t):
        cfg, st, B = self.cfg, self.store, self.B
        D, H = cfg.d_model, cfg.d_heads
        for i in range(cfg.d_layers):
            pre = f"decoder.layer{i}"
            cache = self.cache_sets[(t & 1) if self.pingpong else 0][i]
            # K | Q | V of the new row, written to row t of every sample's cache (C row remap: logical row b -> b * t_max + t)
            o.gemm_nt(x, st.fused(st.w16, pre, "weight"), cache.view(B * self.t_max, 3 * D), M=B, K=D, bias=st.fused(st.w, pre, "bias"),
                      c_remap=(1, self.t_max, t))
            o.attn_decode(cache, t + 1, H, D // H, 0, D, 2 * D, self.att, mode=self.mode)
            proj = dict(N=D, K=D, bias=st.p(f"{pre}.att.W_proj.bias"), resid=x)
            ff1 = dict(K=D, bias=st.p(f"{pre}.ff1.bias"), act=o.ACT_RELU)
            # transformer.py:199-200: LN3(ff + dropout(ff)) — dropout is the identity here, so 2 * ff
            ff2 = dict(K=4 * D, bias=st.p(f"{pre}.ff2.bias"), self_resid=True)
            if self.fuse_ffn:
                head = dict(att=self.att, W=st.h(f"{pre}.att.W_proj.weight"), h1=self.h1, gamma=st.p(f"{pre}.ln1.gamma"),
                            beta=st.p(f"{pre}.ln1.beta"), mean=self.mean1, rstd=self.rstd1, **proj)
                o.ffn_ln_fwd(self.x1, st.h(f"{pre}.ff1.weight"), self.a, st.h(f"{pre}.ff2.weight"), self.h2, st.p(f"{pre}.ln3.gamma"),
                             st.p(f"{pre}.ln3.beta"), self.x2, self.mean, self.rstd, ff1=ff1, ff2=ff2, proj=head)
            else:
                o.gemm_nt(self.att, st.h(f"{pre}.att.W_proj.weight"), self.h1, **proj)
                o.layernorm_fwd(self.h1, st.p(f"{pre}.ln1.gamma"), st.p(f"{pre}.ln1.beta"), self.x1, self.mean, self.rstd, D=D)
                o.gemm_nt(self.x1, st.h(f"{pre}.ff1.weight"), self.a, **ff1)
                o.gemm_nt(self.a, st.h(f"{pre}.ff2.weight"), self.h2, **ff2)
                o.layernorm_fwd(self.h2, st.p(f"{pre}.ln3.gamma"), st.p(f"{pre}.ln3.beta"), self.x2, self.mean, self.rstd, D=D)
            x = self.x2 if i == cfg.d_layers - 1 else self._keep(self.x2)
        return x

    def _keep(self, x2):
        """the next layer's input must survive that layer's own writes to x2"""
        self.x.copy_(x2)
        return self.x

    # ------------------------------------------------------------------ API
    def _enter(self):
        self.stream.wait_stream(torch.cuda.current_stream())
        return torch.cuda.stream(self.stream)

    def _leave(self):
        torch.cuda.current_stream().wait_stream(self.stream)  # the caller reads the results on its own stream

    def reset(self):
        """forget the positions fed so far (the caches are overwritten from row 0 on); captured graphs stay valid"""
        self.t = -1

    def start(self, row0):
        """position 0: the initial state rows [B, >= D] (16-bit), ALREADY scaled and positioned as the training step's
        decoder row 0 (engine.StepPlan.x0_d[:, 0]); fills row 0 of every layer's cache"""
        with self._enter():
            self.x.copy_(row0[:, : self.x.shape[1]])
            self._run_position(0)
        self._leave()
        self.t = 0

    def _position(self, t):
        """the kernel sequence of position t >= 0 on inputs already in self.x (t = 0) / self.tokens / self.frames"""
        cfg, st, B = self.cfg, self.store, self.B
        D = cfg.d_model
        if t == 0:
            self._layers(self.x, 0)
            return
        sq = math.sqrt(float(D))
        if cfg.kind == "token":
            o.embed_fwd(self.tokens, st.p("decoder.embedding.weight"), self.pos[t:], self.x.view(B, 1, -1), 0, sq)
        else:
            o.gemm_nt(self.frames, st.t("decoder.embedding.weight"), self.x, N=D, alpha=sq, rowadd=self.pos[t:], rowadd_period=1)
        x = self._layers(self.x, t)
        o.gemm_nt(x, st.h("decoder.output_layer.weight"), self.logits, K=D, bias=st.p("decoder.output_layer.bias"))
        if cfg.kind == "token":
            # (pre_zeroed: only the probabilities are used here — no 4.7 us memset of a loss nobody reads in every position)
            o.softmax_ce(self.logits, self.zero_labels, self.loss, B, 1, cfg.out_dim, probs=self.probs, pre_zeroed=True)
        else:
            o.sigmoid_bce(self.logits, self.zero_labels, self.loss, B, 1, cfg.out_dim, npos=self.npos, probs=self.probs, pre_zeroed=True)

    def _run_position(self, t):
        """replay position t's graph (captured at first use; the very first position of a plan runs eagerly: HIP modules
        load lazily and cannot be loaded inside a capture)"""
        if not self.use_graphs or not self._warm:
            self._position(t)
            self._warm = self._warm or t >= 1  # (position 1 has touched every kernel of a later position)
            return
        g = self._graphs.get(t)
        if g is None:
            g = self._graphs[t] = o.Graph().capture(lambda: self._position(t))
        g.launch()

    def step(self, prev):
        """position t = previous + 1: `prev` is the token fed at this position ([B] ids, token ends) or the frame ([B, P]
        {0,1}); returns the output distribution of this position, [B, V] fp32 probabilities (token ends) or [B, P] 16-bit
        per-pitch probabilities (piano-roll ends)"""
        cfg, B = self.cfg, self.B
        t = self.t + 1
        if t >= self.t_max:
            raise RuntimeError(f"decode buffers hold {self.t_max} positions")
        host = torch.as_tensor(np.asarray(prev.cpu() if torch.is_tensor(prev) else prev))
        with self._enter():
            if cfg.kind == "token":
                self.tokens.copy_(host.to(torch.int32).view(B, 1))
            else:
                self.frames[:, : cfg.in_dim].copy_(host.to(torch.uint8))
            self._run_position(t)
        self._leave()
        self.t = t
        return self.probs[:, : cfg.out_dim]

    def reorder(self, index):
        """beam search driven from the host: hypothesis j continues hypothesis index[j] — gather the caches' rows
        (sampler.py:236-238)"""
        assert not self.pingpong
        idx = torch.as_tensor(np.asarray(index), dtype=torch.int64, device=self.store.device)
        n = self.t + 1
        with self._enter():
            for i, c in enumerate(self.cache):
                c[:, :n] = c[:, :n].index_select(0, idx)
        self._leave()


class BeamSearch:
    """Beam search with everything of a position on the device (reference sampler.py:198-257): the decode step of the B x K
    hypotheses, their re-ranking (mst_beam_step: scores, token rows, the words fed next) and the gather of the layers' cache
    rows (mst_beam_gather) are ONE captured graph per position; the host launches graphs and looks at a device counter of
    running hypotheses every few positions. Token rows, scores and caches alternate between two buffers by position parity."""

    def __init__(self, store, B, K, i_max, attention="query"):
        from .MIDIUtil.defaults import EOS_ID, PAD_ID, SOS_ID
        self.B, self.K, self.i_max = B, K, i_max
        self.eos, self.pad, self.sos = EOS_ID, PAD_ID, SOS_ID
        self.plan = DecodePlan(store, B * K, i_max + 1, attention=attention, pingpong=True)
        dev, N = store.device, B * K
        self.seqs = [torch.zeros(N, i_max, dtype=torch.int32, device=dev) for _ in range(2)]
        self.scores = [torch.zeros(N, dtype=torch.float32, device=dev) for _ in range(2)]
        self.hyp = torch.zeros(N, dtype=torch.int32, device=dev)
        self.ident = torch.arange(N, dtype=torch.int32, device=dev)
        self.active = torch.zeros(i_max + 1, dtype=torch.int32, device=dev)
        first = torch.full((B, K), float("inf"))
        first[:, 0] = 0.0  # the K copies of a sample start identical: only the first one may expand
        self.first_scores = first.reshape(-1).to(dev)
        self._graphs, self._warm = {}, False
        self.positions = 0

    def _position(self, i):
        p = self.plan
        cur, nxt = i & 1, (i + 1) & 1
        p._position(i)
        o.beam_step(p.probs, self.scores[cur], self.scores[nxt], self.seqs[cur], self.seqs[nxt], self.hyp, p.tokens, i, self.K, self.eos,
                    self.pad, active=self.active)
        D = p.cfg.d_model
        for cin, cout in zip(p.cache_sets[cur], p.cache_sets[nxt]):
            o.beam_gather(cin, cout, self.hyp, i + 1, skip_cols=(D, D))  # K and V only: a past position's Q is never read again

    def run(self, row0, check_every=8):
        """row0: [B * K, >= D] initial decoder rows (every sample's row repeated K times). Returns (token rows [B*K, n] int32,
        scores [B*K] fp32) as host arrays, hypotheses of a sample best first."""
        p = self.plan
        p.reset()
        with p._enter():
            p.x.copy_(row0[:, : p.x.shape[1]])
            p._run_position(0)                      # cache set 0, row 0 ...
            for cin, cout in zip(p.cache_sets[0], p.cache_sets[1]):
                o.beam_gather(cin, cout, self.ident, 1)  # ... which position 1 expects in set 1
            p.t = 0
            self.seqs[1].fill_(self.pad)
            self.seqs[1][:, 0] = self.sos
            self.scores[1].copy_(self.first_scores)
            p.tokens.fill_(self.sos)
            o.zero(self.active)
            last = 0
            for i in range(1, self.i_max):
                if not p.use_graphs or not self._warm:
                    self._position(i)               # (the first position ever runs eagerly: lazy HIP module loads)
                    self._warm = True
                else:
                    g = self._graphs.get(i)
                    if g is None:
                        g = self._graphs[i] = o.Graph().capture(lambda: self._position(i))
                    g.launch()
                p.t = last = i
                if i % check_every == 0 and int(self.active[i].item()) == 0:
                    break                            # every hypothesis has ended (sampler.py: the loop's exit test)
            res = (last + 1) & 1
            got = self.seqs[res][:, : last + 1].cpu().numpy()
            scores = self.scores[res].cpu().numpy()
            live = self.active[: last + 1].cpu().numpy()
        p._leave()
        # rows of width i_max like the host loop's (sampler.py:203): PAD behind the last decoded position
        seqs = np.full((got.shape[0], self.i_max), self.pad, got.dtype)
        seqs[:, : last + 1] = got
        self.positions = last
        # live continuations chosen per position (mst_beam_step's `active` counters): what was DECODED — a finished hypothesis's PAD
        # continuations and the up to check_every - 1 positions run after the last one ended are not tokens
        self.tokens_decoded = int(live[1:].sum())
        return seqs, scores


class AncestralSampling:
    """Ancestral sampling (sampler.py:155-190, token ends) with the draw on the device (mst_sample_step): the host neither
    sees the distributions nor feeds the tokens — it launches a position's kernels and polls a device counter of running
    sequences every few positions."""

    def __init__(self, store, B, i_max, attention="query", seed=0):
        from .MIDIUtil.defaults import EOS_ID, PAD_ID, SOS_ID
        self.B, self.i_max = B, i_max
        self.eos, self.pad, self.sos = EOS_ID, PAD_ID, SOS_ID
        self.plan = DecodePlan(store, B, i_max + 1, attention=attention)
        dev = store.device
        self.seqs = torch.zeros(B, i_max, dtype=torch.int32, device=dev)
        self.scores = torch.zeros(B, dtype=torch.float32, device=dev)
        self.active = torch.zeros(i_max + 1, dtype=torch.int32, device=dev)
        self.seed, self.runs = int(seed), 0
        self.positions = 0

    def run(self, row0, check_every=8):
        """-> (token rows [B, n] int32, scores [B]) as host arrays. (Eager launches: the draw's seed changes from run to run and
        is a kernel argument; a position is the decode step's ~10 launches + one draw, with no host round trip in between.)"""
        p = self.plan
        p.reset()
        self.runs += 1
        seed = (self.seed * 0x9E3779B97F4A7C15 + self.runs) & 0xFFFFFFFFFFFFFFFF
        with p._enter():
            p.x.copy_(row0[:, : p.x.shape[1]])
            p._position(0)
            p.t = 0
            self.seqs.fill_(self.pad)
            self.seqs[:, 0] = self.sos
            o.zero(self.scores)
            o.zero(self.active)
            p.tokens.fill_(self.sos)
            last = 0
            for i in range(1, self.i_max):
                p._position(i)
                o.sample_step(p.probs, self.seqs, self.scores, p.tokens, i, seed, self.eos, self.pad, active=self.active)
                p.t = last = i
                if i % check_every == 0 and int(self.active[i].item()) == 0:
                    break
            seqs = self.seqs[:, : last + 1].cpu().numpy()
            scores = self.scores.cpu().numpy()
        p._leave()
        self.positions = last
        return seqs, scores
