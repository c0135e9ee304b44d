"""MIDI events -> piano-roll frames and the piano-roll batcher (the "MIDIUtil -> piano-roll batcher" of
BASELINE.json's north_star).

The reference snapshot only keeps remnants of its piano-roll representation (`slices_per_quarter_note`,
midi_io.py:8-14, Melody.py:11,16; `utils.visualize_melody`'s [time x pitch] raster, utils.py:52-61); the frame
definition here is the natural one those remnants imply: a frame is `resolution / slices_per_quarter` ticks,
a pitch is 1 in every frame it sounds in. Batches follow the Dataset protocol of data.py with
tokens -> frames [B, T, P]: input = [start row, frames 0..T-2], labels = frames 0..T-1 (next-frame target,
the analogue of tokens=[SOS,data], labels=[data,PAD], data.py:160-168); the start row has only pitch 0 set.

`PinnedBatchPipeline` is the pinned-host -> HBM leg: batches are packed into persistent page-locked ring buffers and
copied on a side HIP stream while the previous step computes; Trainer.fit drives it."""
import os
import time

import numpy as np
import torch

from .MIDIUtil.defaults import MAX_TICKS, NUM_TICKS_IN_A_BIN
from .MIDIUtil.Melody import (Melody, NoteOffEvent, NoteOnEvent, TimeshiftEvent, create_note_off_event, create_note_on_event,
                              create_timeshift_event)
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


def pianoroll_to_melody(roll, slices_per_quarter=4, resolution=None, description=""):
    """Inverse of melody_to_pianoroll: {0,1}^[frames, pitches] -> event melody (note-on / note-off / time-shift ids of
    defaults.py), ready for MIDIUtil.midi_io.MelodyWriter. A pitch that is 1 in consecutive frames is ONE held note.
    Time is carried by time-shift events, which exist in bins of NUM_TICKS_IN_A_BIN ticks below MAX_TICKS
    (Melody.py:117-126), so a frame must be a whole number of bins: with the default resolution (a multiple of
    30 * slices_per_quarter is chosen when none is given) the round trip melody -> roll -> melody -> roll is exact."""
    roll = np.asarray(roll)
    assert roll.ndim == 2
    m = Melody(slices_per_quarter=slices_per_quarter, description=description)
    if resolution is None:
        resolution = NUM_TICKS_IN_A_BIN * slices_per_quarter * 4  # 480 at 4 slices per quarter: 120 ticks = 4 bins a frame
    m.resolution = resolution
    tpf = resolution / float(slices_per_quarter)
    assert tpf >= NUM_TICKS_IN_A_BIN and abs(tpf / NUM_TICKS_IN_A_BIN - round(tpf / NUM_TICKS_IN_A_BIN)) < 1e-9, \
        "a frame must be a whole number of {}-tick bins (resolution {}, {} slices per quarter)".format(
            NUM_TICKS_IN_A_BIN, resolution, slices_per_quarter)
    tpf = int(round(tpf))
    max_shift = (MAX_TICKS - 1) // NUM_TICKS_IN_A_BIN * NUM_TICKS_IN_A_BIN  # largest representable shift (990)
    events, pending = [], 0

    def flush():
        nonlocal pending
        while pending > 0:
            step = min(pending, max_shift)
            events.append(create_timeshift_event(step))
            pending -= step

    prev = np.zeros(roll.shape[1], bool)
    for f in range(roll.shape[0] + 1):
        cur = roll[f] > 0 if f < roll.shape[0] else np.zeros(roll.shape[1], bool)
        offs, ons = np.nonzero(prev & ~cur)[0], np.nonzero(cur & ~prev)[0]
        if len(offs) or len(ons):
            flush()
            events += [create_note_off_event(int(p)) for p in offs]
            events += [create_note_on_event(int(p)) for p in ons]
        pending += tpf
        prev = cur
    m.notes = events
    return m


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


class _Slot:
    """one stage of a plan's input ring: a page-locked host blob, its device twin, and the two events that order reuse"""

    def __init__(self, nbytes, device):
        self.host = torch.zeros(nbytes, dtype=torch.uint8).pin_memory()
        self.dev = torch.zeros(nbytes, dtype=torch.uint8, device=device)
        self.uploaded = torch.cuda.Event()  # recorded on the copy stream after the H2D copy of this slot
        self.consumed = torch.cuda.Event()  # recorded on the compute stream after the step that read this slot
        self.used = False


class StagedBatch:
    def __init__(self, plan, slot, batch):
        self.plan, self.slot, self.batch = plan, slot, batch


class PinnedBatchPipeline:
    """The pinned-host -> HBM leg of the batcher (north_star: "pinned-host pipeline overlapped on a side HIP stream";
    reference hook: data.py:181-198 -> trainer.py:156-157, where every step begins with four synchronous as_in_context copies).

    Per (B, T) plan a ring of `n_slots` PERSISTENT page-locked host blobs and matching device blobs (allocated and pinned
    once, reused for the whole run). stage(batch) packs the batch into the next slot's host blob (engine.StepPlan.pack_into)
    and enqueues ONE host->device copy on the pipeline's own stream, behind the event of the step that last read that device
    blob; the step that consumes the slot makes the compute stream wait for the copy's event and replays a graph captured
    on that device blob (StepPlan.bind_inputs: no device-to-device hop). feed(dataset) runs one batch ahead: batch i+1 is
    packed and uploaded while the graph of step i executes."""

    DEFAULT_SLOTS = int(os.environ.get("MST_RING_SLOTS", "3"))

    def __init__(self, device, plan_for, n_slots=None):
        """plan_for(B, T) -> StepPlan (shapes change from batch to batch on the token path)"""
        n_slots = n_slots or self.DEFAULT_SLOTS
        assert n_slots >= 2
        self.device, self.plan_for, self.n_slots = device, plan_for, n_slots
        self.stream = torch.cuda.Stream(device=device)
        self.rings = {}
        self.stamps = None  # set to [] to collect (wait for the slot, pack, enqueue the copy) seconds per stage() call

    def stage(self, batch, shard=None):
        """pack + upload one batch; shard = (lo, hi) rows of the global batch this rank keeps (data parallel)"""
        x, seq_lens, classes = batch.data
        labels = batch.label[0]
        if shard is not None:
            lo, hi = shard
            x, seq_lens, classes, labels = x[lo:hi], seq_lens[lo:hi], classes[lo:hi], labels[lo:hi]
        x = np.asarray(x)
        plan = self.plan_for(x.shape[0], x.shape[1])
        ring = self.rings.get(id(plan))
        if ring is None:
            ring = self.rings[id(plan)] = {"slots": [_Slot(plan.own_inbuf.numel(), self.device) for _ in range(self.n_slots)], "next": 0,
                                           "plan": plan}
            # (page-locked host blob + device twin per slot: counted against the plan cache's byte budget, model.Model.plan)
            plan.extra_bytes = getattr(plan, "extra_bytes", 0) + 2 * self.n_slots * plan.own_inbuf.numel()
        slot = ring["slots"][ring["next"]]
        ring["next"] = (ring["next"] + 1) % self.n_slots
        t0 = time.perf_counter() if self.stamps is not None else 0.0
        if slot.used:
            slot.uploaded.synchronize()  # the previous copy OUT of this host blob has finished (long ago, normally)
        t1 = time.perf_counter() if self.stamps is not None else 0.0
        plan.pack_into(slot.host, x, seq_lens, classes, labels)
        t2 = time.perf_counter() if self.stamps is not None else 0.0
        with torch.cuda.stream(self.stream):
            if slot.used:
                self.stream.wait_event(slot.consumed)  # the step that read the device blob is done with it
            slot.dev.copy_(slot.host, non_blocking=True)
            slot.uploaded.record(self.stream)
        slot.used = True
        if self.stamps is not None:
            self.stamps.append((t1 - t0, t2 - t1, time.perf_counter() - t2))
        return StagedBatch(plan, slot, batch)

    def feed(self, dataset, shard_of=None):
        """yields StagedBatch objects, staging batch i+1 right after the consumer has launched step i"""
        it = iter(dataset)
        cur = next(it, None)
        staged = self.stage(cur, shard_of(cur) if shard_of else None) if cur is not None else None
        while staged is not None:
            yield staged
            cur = next(it, None)
            staged = self.stage(cur, shard_of(cur) if shard_of else None) if cur is not None else None

    def drop(self, plan):
        """forget a plan's ring (the plan cache evicted it)"""
        self.rings.pop(id(plan), None)
