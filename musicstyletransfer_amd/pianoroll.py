"""MIDI events -> piano-roll frames and the piano-roll batcher (the "MIDIUtil -> piano-roll batcher" of
BASELINE.json's north_star).

The reference snapshot only keeps remnants of its piano-roll representation (`slices_per_quarter_note`,
midi_io.py:8-14, Melody.py:11,16; `utils.visualize_melody`'s [time x pitch] raster, utils.py:52-61); the frame
definition here is the natural one those remnants imply: a frame is `resolution / slices_per_quarter` ticks,
a pitch is 1 in every frame it sounds in. Batches follow the Dataset protocol of data.py with
tokens -> frames [B, T, P]: input = [start row, frames 0..T-2], labels = frames 0..T-1 (next-frame target,
the analogue of tokens=[SOS,data], labels=[data,PAD], data.py:160-168); the start row has only pitch 0 set.

`PinnedBatchPipeline` is the pinned-host -> HBM leg: batches are staged in page-locked buffers and copied
on a side HIP stream while the previous step computes."""
import numpy as np
import torch

from .MIDIUtil.Melody import NoteOffEvent, NoteOnEvent, TimeshiftEvent
from .VarAutoEncoder.data import Batch, _ArrayDataset

N_PITCHES = 128


def melody_to_pianoroll(melody, slices_per_quarter=None, n_pitches=N_PITCHES):
    """rasterise an event melody into {0,1}^[frames, n_pitches]"""
    spq = slices_per_quarter or melody.slices_per_quarter or 4
    ticks_per_frame = max(1.0, melody.resolution / float(spq))
    t, active, spans = 0, {}, []
    for ev in melody:
        if isinstance(ev, TimeshiftEvent):
            t += ev.get_tick_delay()
        elif isinstance(ev, NoteOnEvent):
            active.setdefault(ev.shifted_id, t)
        elif isinstance(ev, NoteOffEvent):
            start = active.pop(ev.shifted_id, None)
            if start is not None:
                spans.append((ev.shifted_id, start, t))
    for pitch, start in active.items():  # notes never released sound until the end
        spans.append((pitch, start, t))
    n_frames = int(np.ceil(t / ticks_per_frame)) + 1
    roll = np.zeros((n_frames, n_pitches), np.uint8)
    for pitch, start, end in spans:
        if pitch < n_pitches:
            a = int(start // ticks_per_frame)
            b = max(a + 1, int(np.ceil(end / ticks_per_frame)))
            roll[a:b, pitch] = 1
    return roll


def pianoroll_arrays(melodies, frames_per_sample, slices_per_quarter=4, n_pitches=N_PITCHES):
    xs, labels, lens, classes = [], [], [], []
    T = frames_per_sample
    for class_idx, (_, ms) in enumerate(sorted(melodies.items())):
        for m in ms:
            roll = melody_to_pianoroll(m, slices_per_quarter, n_pitches)
            for lo in range(0, len(roll), T):
                chunk = roll[lo: lo + T]
                n = len(chunk)
                lab = np.zeros((T, n_pitches), np.uint8)
                lab[:n] = chunk
                x = np.zeros((T, n_pitches), np.uint8)
                x[0, 0] = 1  # start row
                x[1:n] = chunk[: n - 1]
                xs.append(x)
                labels.append(lab)
                lens.append(n)
                classes.append(class_idx)
    return np.stack(xs), np.stack(labels), np.asarray(lens, np.int64), np.asarray(classes, np.int64)


class PianoRollDataset(_ArrayDataset):
    """same protocol as MelodyDataset with frames in place of tokens; samples have a fixed T, so no per-batch
    truncation (every batch replays the same captured hipGraph)"""

    def __init__(self, batch_size, frames_per_sample, melodies, slices_per_quarter=4, n_pitches=N_PITCHES, seed=0):
        super().__init__(batch_size, seed)
        self.n_pitches = n_pitches
        self.n_classes = len(melodies)
        self.x, self.labels, self.seq_lens, self.classes = pianoroll_arrays(melodies, frames_per_sample, slices_per_quarter,
                                                                            n_pitches)

    def num_classes(self):
        return self.n_classes

    def num_tokens(self):
        return self.n_pitches

    def __len__(self):
        return -(-len(self.x) // self.batch_size)

    def __iter__(self):
        for idx, pad in self._epoch_indices(len(self.x)):
            yield Batch([self.x[idx], self.seq_lens[idx], self.classes[idx]], [self.labels[idx]], pad)


class SyntheticPianoRollDataset(_ArrayDataset):
    """Bernoulli(density) frames (SURVEY §8d synthetic input) — what bench.py feeds"""

    def __init__(self, batch_size, frames_per_sample, n_samples, n_pitches=N_PITCHES, num_classes=2, density=0.04, seed=1234):
        super().__init__(batch_size, seed)
        rng = np.random.default_rng(seed)
        T = frames_per_sample
        roll = (rng.random((n_samples, T + 1, n_pitches)) < density).astype(np.uint8)
        self.x = roll[:, :T].copy()
        self.x[:, 0, :] = 0
        self.x[:, 0, 0] = 1
        self.labels = roll[:, 1:].copy()
        self.seq_lens = np.full(n_samples, T, np.int64)
        self.classes = rng.integers(0, num_classes, size=n_samples).astype(np.int64)
        self.n_pitches, self.n_classes = n_pitches, num_classes

    def num_classes(self):
        return self.n_classes

    def num_tokens(self):
        return self.n_pitches

    def __iter__(self):
        for idx, pad in self._epoch_indices(len(self.x)):
            yield Batch([self.x[idx], self.seq_lens[idx], self.classes[idx]], [self.labels[idx]], pad)


class PinnedBatchPipeline:
    """Iterate a Dataset one batch ahead: batch i+1 is packed into ONE page-locked blob (engine.StepPlan.pack_batch) and
    copied host -> device on a side stream while step i runs; `next_into` makes the compute stream wait for that copy and
    moves the blob into the plan's static input buffer with a single device-to-device copy."""

    def __init__(self, dataset, device, plan_for):
        """plan_for(B, T) -> StepPlan (shapes may change from batch to batch on the token path)"""
        self.it, self.device, self.plan_for = iter(dataset), device, plan_for
        self.stream = torch.cuda.Stream(device=device)
        self.staged = None
        self._stage()

    def _stage(self):
        try:
            b = next(self.it)
        except StopIteration:
            self.staged = None
            return
        x = np.asarray(b.data[0])
        plan = self.plan_for(x.shape[0], x.shape[1])
        host = plan.pack_batch(x, b.data[1], b.data[2], b.label[0], pin=True)
        with torch.cuda.stream(self.stream):
            dev = host.to(self.device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        self.staged = (plan, dev, ev, host, b)

    def next_into(self):
        """returns (plan, batch) with the batch loaded into plan's inputs, or None at the end of the epoch"""
        if self.staged is None:
            return None
        plan, dev, ev, _host, b = self.staged
        torch.cuda.current_stream().wait_event(ev)
        plan.load_packed(dev)
        self._stage()
        return plan, b
