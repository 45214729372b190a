"""TEST INFRASTRUCTURE ONLY (never imported by the product): an independent EVT 2.0 decoder, word by word.

Written from the published format description (Prophesee "EVT 2.0" data encoding: 32-bit little-endian words, type in bits
31..28), not from x_maps_amd/evt2.py: a plain loop over the words with the decoder's state in three variables, so that the
product's two decoders (the NumPy forward fills, the device scans) are checked against something that shares no code and no
structure with them.  Metavision's own reader is closed and no recording ships with the reference: parity against it stays
unpinned; tests/test_evt2.py adds hand-derived word sequences.

    type 0x0 CD_OFF, 0x1 CD_ON : bits 27..22 = t & 63, bits 21..11 = x, bits 10..0 = y; polarity = type
    type 0x8 EVT_TIME_HIGH      : bits 27..0 = t >> 6
    anything else               : no event, no state change
    a TIME_HIGH value more than 2^27 below the previous one = the 34-bit clock has wrapped once more
"""
import numpy as np

EVENT_CD = np.dtype({"names": ["x", "y", "p", "t"], "formats": ["<u2", "<u2", "<i2", "<i8"], "offsets": [0, 2, 4, 8], "itemsize": 16})


class Evt2StateMachine:
    def __init__(self, wait_for_time_base=False):
        self.time_high = 0
        self.loops = 0
        self.wait = bool(wait_for_time_base)  # CD words in front of the stream's first EVT_TIME_HIGH are not emitted (see evt3_oracle.py)
        self.have_high = False

    def feed(self, words):
        out = []
        for w in np.asarray(words, dtype="<u4").tolist():
            kind = w >> 28
            if kind == 0x8:
                value = w & 0x0FFFFFFF
                if self.time_high - value > (1 << 27):
                    self.loops += 1
                self.time_high = value
                self.have_high = True
            elif kind in (0x0, 0x1) and not (self.wait and not self.have_high):
                t = (self.loops << 34) | (self.time_high << 6) | ((w >> 22) & 0x3F)
                out.append(((w >> 11) & 0x7FF, w & 0x7FF, kind, t))
        evs = np.zeros(len(out), EVENT_CD)
        if out:
            a = np.array(out, dtype=np.int64)
            evs["x"], evs["y"], evs["p"], evs["t"] = a[:, 0], a[:, 1], a[:, 2], a[:, 3]
        return evs


def decode(words, wait_for_time_base=False):
    return Evt2StateMachine(wait_for_time_base).feed(words)
