"""Vocabulary and MIDI constants of the event representation.

Same values as the reference's MIDIUtil/defaults.py:1-58 (the ids are part of the data format: a model
trained there must read the same token ids here): PAD/SOS/EOS = 0/1/2, 128 note-on ids, 128 note-off ids,
34 time-shift bins of 30 ticks covering [0, 1000) ticks, NUM_EVENTS = 293.
"""
# pitch classes (reference defaults.py:1-13) and the legacy piano-roll feature sizes (:15-18)
(PITCH_C, PITCH_Cis, PITCH_D, PITCH_Dis, PITCH_E, PITCH_F, PITCH_Fis, PITCH_G, PITCH_Gis, PITCH_A, PITCH_Ais,
 PITCH_B, SILENCE) = range(13)
N_PITCHES, N_OCTAVES = 12, 10
N_FEATURES_WITHOUT_SILENCE = N_PITCHES * N_OCTAVES
N_FEATURES_WITH_SILENCE = N_FEATURES_WITHOUT_SILENCE + 1
DEFAULT_BPM, DEFAULT_RESOLUTION = 120, 220
DEF_NUMBER_NOTES, DEF_TICK_STEP_SIZE, MAXIMUM_SEQUENCE_LENGTH = 100, 30, 272

# instrument ranges (defaults.py:24-34): 24-fret guitar E2..E7, 4-string bass E1..D5
MIDI_GUITAR_BEGIN, MIDI_GUITAR_END = 40, 88
MIDI_GUITAR_RANGE = MIDI_GUITAR_END - MIDI_GUITAR_BEGIN + 1
MIDI_BASS_BEGIN, MIDI_BASS_END = 28, 62
MIDI_BASS_RANGE = MIDI_BASS_END - MIDI_BASS_BEGIN + 1

# time-shift quantisation (defaults.py:36-39)
MAX_TICKS, MIN_TICKS, NUM_TICKS_IN_A_BIN = 1000, 0, 30
NUM_BINS = (MAX_TICKS - MIN_TICKS) // NUM_TICKS_IN_A_BIN + 1  # 34

# token ids (defaults.py:41-58); ranges are inclusive
PAD_ID, SOS_ID, EOS_ID = 0, 1, 2
SPECIALS_TOKENS = [PAD_ID, SOS_ID, EOS_ID]
FEATURE_OFFSET = len(SPECIALS_TOKENS)
N_MIDI_PITCHES = 128
NOTE_ON_EVENTS = (FEATURE_OFFSET, FEATURE_OFFSET + N_MIDI_PITCHES - 1)                # 3 .. 130
NOTE_OFF_EVENTS = (NOTE_ON_EVENTS[1] + 1, NOTE_ON_EVENTS[1] + N_MIDI_PITCHES)          # 131 .. 258
TIMESHIFT_EVENTS = (NOTE_OFF_EVENTS[1] + 1, NOTE_OFF_EVENTS[1] + NUM_BINS)             # 259 .. 292
NUM_EVENTS = TIMESHIFT_EVENTS[1] + 1                                                   # 293
