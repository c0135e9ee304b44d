"""MIDI file -> event melodies and back (reference MIDIUtil/midi_io.py:7-132), on the in-tree SMF
parser (smf.py) instead of python-midi."""
from . import smf
from .defaults import DEFAULT_BPM, MAX_TICKS
from .Melody import (Melody, NoteOffEvent, NoteOnEvent, TimeshiftEvent, create_note_off_event, create_note_on_event,
                     create_timeshift_event)


class MIDIReader:
    def __init__(self, slices_per_quarter_note):
        self.slices_per_quarter_note = slices_per_quarter_note

    def _extract_bpm(self, pattern):
        """first tempo found in the file, else the default (midi_io.py:17-26)"""
        for track in pattern:
            for ev in track:
                if ev.kind == "set_tempo":
                    return smf.mpqn_to_bpm(ev.data[0])
        return DEFAULT_BPM

    def read_file(self, file_name):
        raise NotImplementedError


class EventBasedMIDIReader(MIDIReader):
    """midi_io.py:31-93. Quirks kept because they define the token data: (i) a gap of d ticks emits
    time-shifts of d % 1000 while d > 0, d -= 1000 (2500 -> three shifts of 500); (ii) velocity decides
    on/off, so a NoteOff carrying a release velocity > 0 is emitted as note-on; (iii) tracks with fewer
    than 10 events are dropped."""

    def __init__(self):
        super().__init__(0)

    def read_file(self, file_name):
        pattern = smf.read_midifile(file_name)
        bpm = self._extract_bpm(pattern)
        result = []
        for track in pattern:
            melody = Melody(bpm=bpm, resolution=pattern.resolution, slices_per_quarter=self.slices_per_quarter_note)
            melody.notes = self._parse_track(track)
            if len(melody) < 10:  # description / tempo tracks
                continue
            result.append(melody)
        assert len(result) > 0, f"{file_name}: no track with at least 10 events"
        return result

    def _parse_track(self, track):
        events = []
        prev_t = cur_t = 0
        for ev in track:
            cur_t += ev.tick
            if ev.kind not in ("note_on", "note_off"):
                continue
            delta_t = cur_t - prev_t
            note, velocity = ev.data
            while delta_t > 0:
                events.append(create_timeshift_event(delta_t % MAX_TICKS))
                delta_t -= MAX_TICKS
            if velocity > 0:
                events.append(create_note_on_event(note))
            elif velocity == 0:
                events.append(create_note_off_event(note))
            prev_t = cur_t
        return events


class MelodyWriter:
    """events -> single-track SMF (midi_io.py:96-132)"""

    def __init__(self):
        self.tempo = DEFAULT_BPM

    def write_to_file(self, file_name, melody):
        track = [smf.Message(0, "set_tempo", (smf.bpm_to_mpqn(melody.bpm),))]
        self._write_track(melody, track)
        track.append(smf.Message(1, "end_of_track", ()))
        smf.write_midifile(file_name, smf.Pattern([track], resolution=melody.resolution))

    def _write_track(self, melody, track):
        tick_delay = 0
        for ev in melody:
            if isinstance(ev, TimeshiftEvent):
                tick_delay += ev.get_tick_delay()
            elif isinstance(ev, (NoteOnEvent, NoteOffEvent)):
                track.append(ev.get_midi_event(int(tick_delay)))
                tick_delay = 0
