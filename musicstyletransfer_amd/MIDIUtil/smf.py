"""Minimal Standard MIDI File (SMF) reader / writer.

The reference reads and writes MIDI through the third-party python-midi package (midi.read_midifile /
midi.write_midifile, MIDIUtil/midi_io.py:39,109), which is not installable here; this module provides
the subset the hot path's batcher needs: format 0/1 files, delta times, running status, channel voice
messages, meta events (tempo) and sysex skipping. Times are delta ticks, as in python-midi's default."""
import struct
from collections import namedtuple

# kind: 'note_on' | 'note_off' | 'set_tempo' | 'end_of_track' | 'other'; data: tuple of ints
Message = namedtuple("Message", ["tick", "kind", "data"])


class Pattern(list):
    """list of tracks (each a list of Message) + header fields"""

    def __init__(self, tracks=(), resolution=220, format=1):
        super().__init__(tracks)
        self.resolution, self.format = resolution, format


def _read_vlq(buf, pos):
    val = 0
    while True:
        b = buf[pos]
        pos += 1
        val = (val << 7) | (b & 0x7F)
        if not b & 0x80:
            return val, pos


def _write_vlq(val):
    out = [val & 0x7F]
    val >>= 7
    while val:
        out.append((val & 0x7F) | 0x80)
        val >>= 7
    return bytes(reversed(out))


_CHANNEL_LEN = {0x8: 2, 0x9: 2, 0xA: 2, 0xB: 2, 0xC: 1, 0xD: 1, 0xE: 2}


def _parse_track(buf):
    events, pos, status = [], 0, None
    n = len(buf)
    while pos < n:
        tick, pos = _read_vlq(buf, pos)
        b = buf[pos]
        if b == 0xFF:  # meta event
            mtype = buf[pos + 1]
            length, pos = _read_vlq(buf, pos + 2)
            data = buf[pos: pos + length]
            pos += length
            if mtype == 0x51 and length == 3:
                events.append(Message(tick, "set_tempo", ((data[0] << 16) | (data[1] << 8) | data[2],)))
            elif mtype == 0x2F:
                events.append(Message(tick, "end_of_track", ()))
            else:
                events.append(Message(tick, "other", (mtype,)))
        elif b in (0xF0, 0xF7):  # sysex
            length, pos = _read_vlq(buf, pos + 1)
            pos += length
            events.append(Message(tick, "other", (b,)))
        else:
            if b & 0x80:
                status = b
                pos += 1
            elif status is None:
                raise ValueError("running status without a previous status byte")
            hi = status >> 4
            nbytes = _CHANNEL_LEN.get(hi)
            if nbytes is None:
                raise ValueError("unsupported status byte 0x%02x" % status)
            data = tuple(buf[pos: pos + nbytes])
            pos += nbytes
            if hi == 0x9:
                events.append(Message(tick, "note_on", data))
            elif hi == 0x8:
                events.append(Message(tick, "note_off", data))
            else:
                events.append(Message(tick, "other", (status,) + data))
    return events


def read_midifile(path):
    with open(path, "rb") as f:
        raw = f.read()
    if raw[:4] != b"MThd":
        raise ValueError(f"{path}: not a Standard MIDI File")
    hlen, fmt, ntrks, division = struct.unpack(">IHHH", raw[4:14])
    if division & 0x8000:
        raise ValueError(f"{path}: SMPTE time division is not supported")
    pos = 8 + hlen
    pat = Pattern(resolution=division, format=fmt)
    for _ in range(ntrks):
        if raw[pos: pos + 4] != b"MTrk":
            raise ValueError(f"{path}: missing MTrk chunk")
        (tlen,) = struct.unpack(">I", raw[pos + 4: pos + 8])
        pat.append(_parse_track(raw[pos + 8: pos + 8 + tlen]))
        pos += 8 + tlen
    return pat


def _encode_track(events):
    out = bytearray()
    for ev in events:
        out += _write_vlq(int(ev.tick))
        if ev.kind == "note_on":
            out += bytes([0x90, ev.data[0] & 0x7F, ev.data[1] & 0x7F])
        elif ev.kind == "note_off":
            out += bytes([0x80, ev.data[0] & 0x7F, ev.data[1] & 0x7F])
        elif ev.kind == "set_tempo":
            mpqn = int(ev.data[0])
            out += bytes([0xFF, 0x51, 0x03, (mpqn >> 16) & 0xFF, (mpqn >> 8) & 0xFF, mpqn & 0xFF])
        elif ev.kind == "end_of_track":
            out += bytes([0xFF, 0x2F, 0x00])
        else:
            raise ValueError(f"cannot encode event kind {ev.kind}")
    return bytes(out)


def write_midifile(path, pattern):
    chunks = [_encode_track(t) for t in pattern]
    with open(path, "wb") as f:
        f.write(b"MThd" + struct.pack(">IHHH", 6, pattern.format, len(chunks), pattern.resolution))
        for c in chunks:
            f.write(b"MTrk" + struct.pack(">I", len(c)) + c)


def bpm_to_mpqn(bpm):
    return int(round(60e6 / float(bpm)))


def mpqn_to_bpm(mpqn):
    return 60e6 / float(mpqn)
