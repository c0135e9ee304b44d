"""Host batcher: MIDI directory -> event melodies -> padded token / label arrays -> batches
(reference VarAutoEncoder/data.py:14-223), plus the piano-roll batcher of the same protocol.

A Dataset is an iterable exposing num_classes(), num_tokens(), batch_size and yielding batches whose
.data == [tokens[B,T], seq_lens[B], classes[B]] and .label == [labels[B,T]] (data.py:42-54,181-198); arrays
are numpy (the reference's are mx.nd float32 holding integers). Quirks of the reference's chunker that
define the training data are reproduced and flagged below."""
import glob
import os
from typing import Dict, List

import numpy as np

from ..MIDIUtil.defaults import EOS_ID, NUM_EVENTS, PAD_ID, SOS_ID
from ..MIDIUtil.Melody import Melody
from ..MIDIUtil.midi_io import EventBasedMIDIReader


class Batch:
    """stand-in for mx.io.DataBatch: .data and .label lists, .pad = wrapped-around samples in the last batch"""

    def __init__(self, data, label, pad=0):
        self.data, self.label, self.pad = list(data), list(label), pad


class Loader:
    """data.py:14-39: every sub-directory of `path` is a class; first track with >= 10 events of each file"""

    def __init__(self, path: str, max_sequence_length: int, slices_per_quarter_note: int):
        self.path = path
        self.max_sequence_length = max_sequence_length
        self.slices_per_quarter_note = slices_per_quarter_note
        self.midi_reader = EventBasedMIDIReader()
        self.melodies = self.read_melodies()

    def read_melodies(self):
        print("Reading from {}".format(self.path))
        melodies = {}
        for directory in sorted(next(os.walk(self.path))[1]):
            files = sorted(glob.glob(os.path.join(self.path, directory, "*.mid")))  # sorted: the reference's order is FS-dependent
            melodies[directory] = [self.midi_reader.read_file(f)[0] for f in files]
            print("Read {} files from {}".format(len(files), directory))
        return melodies


class Dataset:
    def __init__(self, batch_size: int):
        self.batch_size = batch_size

    def num_classes(self):
        raise NotImplementedError

    def num_tokens(self):
        raise NotImplementedError

    def __iter__(self):
        raise NotImplementedError


class ToyData(Dataset):
    """data.py:57-81: three fixed sequences, seq_len 4, classes 0..2"""

    def __init__(self, batch_size: int = 3):
        super().__init__(batch_size)
        self.tokens = np.array([[1, 5, 6, 7, 0], [1, 6, 7, 8, 0], [1, 7, 8, 9, 0]], np.int64)
        self.seq_lens = np.array([4, 4, 4], np.int64)
        self.classes = np.array([0, 1, 2], np.int64)
        self.labels = np.array([[5, 6, 7, 2, 0], [6, 7, 8, 2, 0], [7, 8, 9, 2, 0]], np.int64)

    def num_classes(self):
        return 3

    def num_tokens(self):
        return 10

    def __iter__(self):
        for lo in range(0, 3, self.batch_size):
            idx = (np.arange(lo, lo + self.batch_size)) % 3
            yield Batch([self.tokens[idx], self.seq_lens[idx], self.classes[idx]], [self.labels[idx]],
                        pad=max(0, lo + self.batch_size - 3))


def count_sequence_length(tokens):
    """data.py:175-179: number of non-PAD positions per row"""
    return (tokens != PAD_ID).sum(axis=1)


def chunk_melodies(melodies: Dict[str, List[Melody]], max_seq_len: int):
    """data.py:133-155. Reproduced quirks: the running chunk of every melody is appended when the melody ends
    even if it is still all-PAD (a melody whose length is a multiple of max_seq_len yields an empty row), and
    after the last melody of a class its final chunk is appended a second time if it is non-empty."""
    all_tokens, all_classes = [], []
    tokens = None
    for class_idx, melodies_for_class in enumerate(melodies.values()):
        for melody in melodies_for_class:
            tokens = np.full((max_seq_len,), PAD_ID, np.int64)
            for j, event in enumerate(melody):
                rel = j % max_seq_len
                tokens[rel] = event.id
                if rel == max_seq_len - 1:
                    all_tokens.append(tokens)
                    all_classes.append(class_idx)
                    tokens = np.full((max_seq_len,), PAD_ID, np.int64)
            all_tokens.append(tokens)
            all_classes.append(class_idx)
        if tokens is not None and tokens[0] != PAD_ID:
            all_tokens.append(tokens)
            all_classes.append(class_idx)
    return all_tokens, all_classes


def token_arrays(melodies, max_seq_len):
    """data.py:157-169: tokens = [SOS, data], labels = [data, PAD] with EOS. The reference writes EOS with
    `labels[:, seq_lens] = EOS_ID`, i.e. (numpy-style advanced indexing) into column len_i of EVERY row for
    every distinct length in the dataset — reproduced."""
    all_tokens, all_classes = chunk_melodies(melodies, max_seq_len)
    n = len(all_tokens)
    assert n > 0, "Empty sequences were found"
    data = np.stack(all_tokens, axis=0)
    tokens = np.concatenate([np.full((n, 1), SOS_ID, np.int64), data], axis=1)
    seq_lens = count_sequence_length(data)
    labels = np.concatenate([data, np.full((n, 1), PAD_ID, np.int64)], axis=1)
    labels[:, np.unique(seq_lens)] = EOS_ID
    return tokens, labels, np.asarray(all_classes, np.int64)


class _ArrayDataset(Dataset):
    """shuffled epochs over in-memory arrays; the last batch wraps around to the start
    (mx.io.NDArrayIter(shuffle=True), last_batch_handle='pad': data.py:111-114)"""

    def __init__(self, batch_size, seed=0):
        super().__init__(batch_size)
        self._rng = np.random.default_rng(seed)

    def _epoch_indices(self, n):
        order = self._rng.permutation(n)
        for lo in range(0, n, self.batch_size):
            idx = order[lo: lo + self.batch_size]
            pad = self.batch_size - len(idx)
            if pad:
                idx = np.concatenate([idx, order[:pad]])
            yield idx, pad


class MelodyDataset(_ArrayDataset):
    """data.py:84-198"""

    def __init__(self, batch_size: int, maximum_sequence_length: int, melodies: Dict[str, List[Melody]], seed=0):
        super().__init__(batch_size, seed)
        self.max_seq_len = maximum_sequence_length
        self.mask_offset = 1
        melodies = dict(sorted(melodies.items(), key=lambda kv: kv[0]))
        self.n_classes = len(melodies)
        self.n_melodies = sum(len(m) for m in melodies.values())
        self.seen_max_sequence_length = max((len(x) for ms in melodies.values() for x in ms), default=0)
        self.tokens, self.labels, self.classes = token_arrays(melodies, self.max_seq_len)
        print("Dataset: {} classes, {} tokens, tokens {}, labels {}".format(self.n_classes, NUM_EVENTS, self.tokens.shape,
                                                                            self.labels.shape))

    def num_classes(self):
        return self.n_classes

    def num_tokens(self):
        return NUM_EVENTS

    def __len__(self):
        return -(-len(self.tokens) // self.batch_size)

    def __iter__(self):
        for idx, pad in self._epoch_indices(len(self.tokens)):
            yield preprocess_batch(Batch([self.tokens[idx], self.classes[idx]], [self.labels[idx]], pad))


def preprocess_batch(batch):
    """data.py:187-198: insert the per-row lengths (non-PAD count, SOS included) at data[1] and truncate
    tokens / labels to the longest row of the batch"""
    tokens = batch.data[0]
    seq_lens = count_sequence_length(tokens)
    batch.data.insert(1, seq_lens)
    max_len = int(seq_lens.max())
    batch.data[0] = tokens[:, :max_len]
    batch.label[0] = batch.label[0][:, :max_len]
    return batch


def load_dataset(loader_train: Loader, batch_size: int, split_percentage: float = None, loader_val: Loader = None,
                 dataset_cls=None, **kw):
    """data.py:201-223"""
    cls = dataset_cls or MelodyDataset
    L = loader_train.max_sequence_length
    if loader_val is not None:
        return cls(batch_size, L, loader_train.melodies, **kw), cls(batch_size, loader_val.max_sequence_length, loader_val.melodies, **kw)
    if split_percentage is None or split_percentage <= 0.0:
        return cls(batch_size, L, loader_train.melodies, **kw), None
    assert 0.0 < split_percentage < 1.0
    train_split, valid_split = {}, {}
    for c, m in loader_train.melodies.items():
        n_val = int(split_percentage * len(m))
        valid_split[c], train_split[c] = m[:n_val], m[n_val:]
    return cls(batch_size, L, train_split, **kw), cls(batch_size, L, valid_split, **kw)
