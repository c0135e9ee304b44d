"""Melody container and the event <-> token-id factories (reference MIDIUtil/Melody.py:6-126).

The reference's event classes wrap python-midi objects; python-midi is not available, so events here
carry plain (kind, value) data and `get_midi_event` returns an `smf.Message`."""
from . import smf
from .defaults import (DEFAULT_BPM, DEFAULT_RESOLUTION, FEATURE_OFFSET, MAX_TICKS, MIN_TICKS, NOTE_OFF_EVENTS,
                       NOTE_ON_EVENTS, NUM_EVENTS, NUM_TICKS_IN_A_BIN, PITCH_C, TIMESHIFT_EVENTS)


class Melody:
    """a track as a list of events plus its meta-information (Melody.py:6-33)"""

    def __init__(self, key=PITCH_C, bpm=DEFAULT_BPM, resolution=DEFAULT_RESOLUTION, slices_per_quarter=4, description=""):
        self.key, self.bpm, self.resolution = key, bpm, resolution
        self.slices_per_quarter = int(slices_per_quarter)
        self.description = description
        self.notes = []

    def __len__(self):
        return len(self.notes)

    def __getitem__(self, i):
        return self.notes[i]

    def copy_metainformation(self):
        return Melody(self.key, self.bpm, self.resolution, self.slices_per_quarter, self.description)


class Event:
    first_id = 0

    def __init__(self, id):
        self.id = int(id)

    @property
    def shifted_id(self):
        return int(self.id - self.first_id)

    def get_midi_event(self, tick_delay):
        raise NotImplementedError

    def __eq__(self, other):
        return type(other) is type(self) and other.id == self.id

    def __hash__(self):
        return hash((type(self).__name__, self.id))

    def __repr__(self):
        return f"{type(self).__name__}({self.shifted_id})"


class NoteOnEvent(Event):
    first_id = NOTE_ON_EVENTS[0]

    def get_midi_event(self, tick_delay):  # Melody.py:56-59: velocity 127
        return smf.Message(int(tick_delay), "note_on", (self.shifted_id, 127))


class NoteOffEvent(Event):
    first_id = NOTE_OFF_EVENTS[0]

    def get_midi_event(self, tick_delay):  # Melody.py:69-71
        return smf.Message(int(tick_delay), "note_off", (self.shifted_id, 0))


class TimeshiftEvent(Event):
    first_id = TIMESHIFT_EVENTS[0]

    def get_tick_delay(self):  # Melody.py:82-83
        return self.shifted_id * NUM_TICKS_IN_A_BIN


def create_event_from_id(id):
    """Melody.py:93-106"""
    if id >= NUM_EVENTS or id < NOTE_ON_EVENTS[0]:
        raise ValueError("ID {} is not in range [{}, {}]".format(id, NOTE_ON_EVENTS[0], NUM_EVENTS))
    if id >= TIMESHIFT_EVENTS[0]:
        return TimeshiftEvent(id)
    if id >= NOTE_OFF_EVENTS[0]:
        return NoteOffEvent(id)
    return NoteOnEvent(id)


def get_melody_from_ids(ids):
    """Melody.py:87-90: special tokens (PAD/SOS/EOS) are dropped"""
    m = Melody()
    m.notes = [create_event_from_id(int(i)) for i in ids if i >= FEATURE_OFFSET]
    return m


def create_note_on_event(pitch):
    return NoteOnEvent(NOTE_ON_EVENTS[0] + int(pitch))


def create_note_off_event(pitch):
    return NoteOffEvent(NOTE_OFF_EVENTS[0] + int(pitch))


def create_timeshift_event(timeshift_ticks):
    """Melody.py:117-126: bins of NUM_TICKS_IN_A_BIN ticks; only [MIN_TICKS, MAX_TICKS) is representable"""
    assert MIN_TICKS <= timeshift_ticks < MAX_TICKS, \
        "Time shift must be between {} ticks and {} ticks. It is {}.".format(MIN_TICKS, MAX_TICKS, timeshift_ticks)
    binned = int((timeshift_ticks - MIN_TICKS) / NUM_TICKS_IN_A_BIN)
    assert TIMESHIFT_EVENTS[0] + binned <= TIMESHIFT_EVENTS[1]
    return TimeshiftEvent(TIMESHIFT_EVENTS[0] + binned)
