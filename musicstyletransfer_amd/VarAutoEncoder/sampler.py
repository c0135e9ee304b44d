"""Sampler interface (reference VarAutoEncoder/sampler.py:41-53) and the reconstruction writer.

Sampling / beam search proper is inference and out of scope of the training-step hot path (the
reference's samplers do not match its own decoder signature, §3.4). What the trainer needs is an object with
update_parameters / process_batch; `ReconstructionSampler` is that object for SURVEY §8f rank 3: it runs the model's
forward pass (teacher forced, as in training) on the batch it is handed and writes the reconstruction of every sample
as a .mid file through MIDIUtil.midi_io.MelodyWriter — arg-max events for the token ends (Melody.get_melody_from_ids,
Melody.py:87-90), frames thresholded at 0.5 for the piano-roll ends."""
import os

import numpy as np

from ..MIDIUtil.Melody import get_melody_from_ids
from ..MIDIUtil.midi_io import MelodyWriter


class SamplerBase:
    def update_parameters(self, model):
        self.model = model

    def process_batch(self, batch, output_path, num_classes):
        return None


class ReconstructionSampler(SamplerBase):
    def __init__(self, threshold=0.5, slices_per_quarter=4, max_files=8):
        self.threshold, self.slices_per_quarter, self.max_files = threshold, slices_per_quarter, max_files
        self.writer = MelodyWriter()

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


def get_sampler(name, model_folder, context, checkpoint, args):
    if name not in ("sampling", "beam-search", "reconstruction"):
        raise ValueError("unknown sampler " + str(name))
    return ReconstructionSampler()
