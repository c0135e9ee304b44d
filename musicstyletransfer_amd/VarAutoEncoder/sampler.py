"""Samplers (reference VarAutoEncoder/sampler.py:41-257): ancestral sampling and beam search over the decoder's incremental
decode step (decode.DecodePlan: per-layer K | Q | V caches in HBM, one position per step), plus the reconstruction
writer of SURVEY §8f rank 3. Interface as in the reference: get_sampler(type, model_folder, context, checkpoint, args),
SamplerBase.update_parameters / process_batch / process_dataset / sample. The reference's own versions do not match its
decoder (`forward_inference` is called with two different signatures, the end test looks for SOS instead of
EOS); the control flow below is theirs with those repaired."""
import os

import numpy as np

from ..MIDIUtil.defaults import EOS_ID, NUM_EVENTS, PAD_ID, SOS_ID
from ..MIDIUtil.Melody import get_melody_from_ids
from ..MIDIUtil.midi_io import MelodyWriter


class SamplerBase:
    def __init__(self, model_folder=None, context=None, checkpoint=None, verbose=False, attention="query", seed=0):
        self.model_folder, self.context, self.verbose = model_folder, context, verbose
        self.attention = attention
        self.rng = np.random.default_rng(seed)
        self.writer = MelodyWriter()
        self.model = None
        if model_folder is not None and checkpoint is not None:
            self.model = load_inference_model(model_folder, context, checkpoint)

    def update_parameters(self, model):
        self.model = model

    def sample(self, batch):
        raise NotImplementedError

    def _melody(self, ids):
        ids = [int(i) for i in np.asarray(ids).reshape(-1) if int(i) not in (PAD_ID, SOS_ID)]
        if EOS_ID in ids:
            ids = ids[: ids.index(EOS_ID)]
        return get_melody_from_ids(np.asarray(ids, np.int64))

    def process_batch(self, batch, output_suffix, num_classes):
        """sampler.py:111-135: the originals, then one sample per sequence and output class"""
        os.makedirs(output_suffix, exist_ok=True)
        files = []
        if self.model.engine_config.kind == "token":
            for i, seq in enumerate(np.asarray(batch.data[0])):
                files.append(os.path.join(output_suffix, "out-{}.original.mid".format(i)))
                self.writer.write_to_file(files[-1], self._melody(seq))
        classes = np.asarray(batch.data[2])
        for class_idx in range(num_classes):
            batch.data[2] = np.full_like(classes, class_idx)
            for i, melody in enumerate(self.sample_melodies(batch)):
                files.append(os.path.join(output_suffix, "out-{}.class-{}.mid".format(i, class_idx)))
                self.writer.write_to_file(files[-1], melody)
        batch.data[2] = classes
        return files

    def process_dataset(self, dataset, output_suffix):
        """sampler.py:77-109"""
        os.makedirs(output_suffix, exist_ok=True)
        idx = 0
        for batch in dataset:
            sub = os.path.join(output_suffix, "batch-{}".format(idx))
            self.process_batch(batch, sub, dataset.num_classes())
            idx += 1

    def sample_melodies(self, batch):
        out = self.sample(batch)
        if self.model.engine_config.kind == "token":
            return [self._melody(seq) for seq in out]
        from ..pianoroll import pianoroll_to_melody
        return [pianoroll_to_melody(roll, 4) for roll in out]


class Sampling(SamplerBase):
    """ancestral sampling (sampler.py:155-190): draw the next token from the decoder's distribution until every sequence
    has ended or twice the input length is reached. Piano-roll ends: every pitch of the next frame is a Bernoulli draw."""

    def sample(self, batch):
        tokens, seq_lens, classes = batch.data
        tokens = np.asarray(tokens)
        B = tokens.shape[0]
        i_max = tokens.shape[1] * 2  # sampler.py:163
        dec = self.model.decoder
        kind = self.model.engine_config.kind
        self.scores = np.zeros(B)
        if kind == "token" and os.environ.get("MST_SAMPLE_DEVICE", "1") != "0":
            # the draw on the device (decode.AncestralSampling): no distribution crosses to the host
            from ..decode import AncestralSampling
            key = (B, i_max, self.attention)
            if getattr(self, "_dev_key", None) != (key, id(self.model.store)):
                self._dev, self._dev_key = AncestralSampling(self.model.store, B, i_max, self.attention, seed=int(self.rng.integers(1 << 62))), (key, id(self.model.store))
            seqs, scores = self._dev.run(dec.initial_rows(tokens, seq_lens, classes))
            self.scores = scores.astype(np.float64)
            return seqs.astype(np.int64)
        state = dec.get_initial_state(tokens, seq_lens, classes, t_max=i_max + 1, attention=self.attention)
        if kind == "token":
            done = np.zeros(B, bool)
            for _ in range(1, i_max):
                probs = dec.forward_inference(state).float().cpu().numpy().astype(np.float64)
                probs /= probs.sum(-1, keepdims=True)
                nxt = np.array([self.rng.choice(probs.shape[1], p=probs[b]) for b in range(B)])
                nxt = np.where(done, PAD_ID, nxt)
                self.scores += np.where(done, 0.0, -np.log(np.maximum(probs[np.arange(B), nxt], 1e-30)))
                state.advance_state(nxt)
                done |= (nxt == EOS_ID) | (nxt == PAD_ID)
                if done.all():
                    break
            return state.tokens
        frames = []
        P = self.model.engine_config.in_dim
        prev = np.zeros((B, P), np.uint8)
        prev[:, 0] = 1  # the start row (pianoroll.pianoroll_arrays)
        for _ in range(1, i_max):
            probs = state.plan.step(prev).float().cpu().numpy()
            prev = (self.rng.random(probs.shape) < probs).astype(np.uint8)
            frames.append(prev)
        return np.stack(frames, 1)


class BeamSearchSampler(SamplerBase):
    """beam search (sampler.py:193-257), token ends: `beam_size` hypotheses per sample, scores = summed -log p, finished
    hypotheses (EOS / PAD) are extended by PAD at no cost; the caches are re-gathered as hypotheses are re-ranked."""

    def __init__(self, *args, beam_size=4, on_device=None, **kw):
        """on_device (default: MST_BEAM_DEVICE != 0): ranking and cache reorder as device kernels inside each position's captured
        graph (decode.BeamSearch); False: the host loop below, one device->host copy of the distributions per position"""
        super().__init__(*args, **kw)
        self.beam_size = beam_size
        self.max_length_factor = 2.0
        self.on_device = (os.environ.get("MST_BEAM_DEVICE", "1") != "0") if on_device is None else bool(on_device)

    def sample(self, batch):
        tokens, seq_lens, classes = batch.data
        tokens = np.asarray(tokens)
        B, K = tokens.shape[0], self.beam_size
        if self.model.engine_config.kind != "token":
            raise ValueError("beam search ranks token sequences; use 'sampling' for the piano-roll ends")
        V = self.model.engine_config.out_dim
        i_max = int(tokens.shape[1] * self.max_length_factor)
        dec = self.model.decoder
        if self.on_device and K <= 16:
            bs = self.model.beam_search_plan(B, K, i_max, self.attention)
            seqs, scores = bs.run(dec.initial_rows(tokens, seq_lens, classes, beam=K))
            self.positions_decoded = bs.positions
            self.tokens_decoded = bs.tokens_decoded
            self.scores = scores.astype(np.float64).reshape(B, K)
            self.hypotheses = seqs.astype(np.int64).reshape(B, K, -1)
            return self.hypotheses[:, 0]
        state = dec.get_initial_state(tokens, seq_lens, classes, t_max=i_max + 1, attention=self.attention, beam=K)
        seqs = np.full((B * K, i_max), PAD_ID, np.int64)
        seqs[:, 0] = SOS_ID
        scores = np.zeros(B * K)
        scores.reshape(B, K)[:, 1:] = np.inf  # the K copies of a sample start identical: only the first one may expand
        offset = np.repeat(np.arange(0, B * K, K), K)
        for i in range(1, i_max):
            state.tokens = seqs[:, :i]
            probs = dec.forward_inference(state).float().cpu().numpy().astype(np.float64)
            exp = -np.log(np.maximum(probs, 1e-30))
            finished = (seqs[:, i - 1] == EOS_ID) | ((seqs[:, i - 1] == PAD_ID) & (i > 1))
            exp[finished] = np.inf          # a finished hypothesis continues with PAD only, at no cost (sampler.py:218-221)
            exp[finished, PAD_ID] = 0.0
            total = (scores[:, None] + exp).reshape(B, K * V)
            top = np.argsort(total, axis=1, kind="stable")[:, :K]
            hyp, word = np.unravel_index(top.reshape(-1), (K, V))
            hyp = hyp + offset
            seqs = seqs[hyp]
            seqs[:, i] = word
            scores = np.take_along_axis(total, top, 1).reshape(-1)
            state.plan.reorder(hyp)
            if ((seqs[:, i] == EOS_ID) | (seqs[:, i] == PAD_ID)).all():
                break
        self.positions_decoded = i  # (bench.py --decode)
        self.tokens_decoded = int(((seqs[:, 1:i + 1] != PAD_ID)).sum())
        self.scores = scores.reshape(B, K)
        self.hypotheses = seqs.reshape(B, K, -1)
        return self.hypotheses[:, 0]  # best hypothesis of every sample


class ReconstructionSampler(SamplerBase):
    """teacher-forced reconstruction: the model's forward pass on the batch it is handed, arg-max events
    for the token ends (Melody.get_melody_from_ids, Melody.py:87-90), frames thresholded at 0.5 for the piano-roll ends"""

    def __init__(self, threshold=0.5, slices_per_quarter=4, max_files=8, **kw):
        super().__init__(**kw)
        self.threshold, self.slices_per_quarter, self.max_files = threshold, slices_per_quarter, max_files

    def reconstruct(self, batch):
        """-> list of Melody, one per sample of the batch (valid positions only)"""
        from ..pianoroll import pianoroll_to_melody
        tokens, seq_lens, classes = batch.data
        probs, _, _ = self.model(tokens, seq_lens, classes)
        probs = probs.cpu().numpy()
        lens = np.asarray(seq_lens).astype(int)
        out = []
        for b in range(probs.shape[0]):
            p = probs[b, : lens[b]]
            if self.model.engine_config.kind == "token":
                out.append(get_melody_from_ids(p.argmax(-1)))
            else:
                out.append(pianoroll_to_melody(p > self.threshold, self.slices_per_quarter))
        return out

    def process_batch(self, batch, output_path, num_classes):
        os.makedirs(output_path, exist_ok=True)
        files = []
        for i, melody in enumerate(self.reconstruct(batch)[: self.max_files]):
            f = os.path.join(output_path, "reconstruction_{}.mid".format(i))
            self.writer.write_to_file(f, melody)
            files.append(f)
        return files


def load_inference_model(model_folder, context, checkpoint):
    """sampler.py:17-38: the saved YAML configuration, a Model built from it, the checkpoint's parameters (-1: latest)"""
    from . import model, utils
    from .config import Config
    c = Config.load(os.path.join(model_folder, "config"))
    m = model.Model(c)
    if checkpoint is None:
        return m
    if checkpoint == -1:
        checkpoint = utils.get_latest_checkpoint_index(model_folder)
    utils.load_model_parameters(m, os.path.join(model_folder, "params.{}".format(checkpoint)), context)
    return m


def get_sampler(type, model_folder, context, checkpoint, args):
    """sampler.py:41-53, plus 'reconstruction' (what the trainer's periodic hook uses by default here)"""
    verbose = bool(getattr(args, "verbose", False))
    if type == "sampling":
        return Sampling(model_folder, context, checkpoint, verbose=verbose)
    if type == "beam-search":
        return BeamSearchSampler(model_folder, context, checkpoint, beam_size=int(getattr(args, "beam_size", 4) or 4), verbose=verbose)
    if type == "reconstruction":
        return ReconstructionSampler()
    raise ValueError("Sampler {} is not implemented".format(type))
