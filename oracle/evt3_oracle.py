"""TEST INFRASTRUCTURE -- an independent checker for the EVT 3.0 decoders (the device kernels of x_maps_amd/csrc/xmaps_evt3.hpp
and the vectorised host decoder x_maps_amd/evt3.py).  Only tests/ may import this file; it shares no code with the product.

What it restates: Prophesee's public EVT 3.0 data format (16-bit little-endian words, type in the top nibble), read the way the
reference's recordings are read -- python/bias_events_iterator.py:53-96 hands the file to Metavision's RawReaderBase, whose
decoder is closed source and absent here, so PARITY IS UNPINNED against Metavision; this file pins the product's two decoders
to ONE plain reading of the published format instead, one word at a time, and that reading itself is pinned by hand-derived
word sequences in tests/test_evt3_oracle.py.

The state machine (one `if` chain per word; nothing is vectorised on purpose):

  nibble  name           payload                      effect
  0x0     EVT_ADDR_Y     y = bits 10..0               current row (bit 11 = master/slave camera flag, ignored)
  0x2     EVT_ADDR_X     x = bits 10..0, p = bit 11   ONE event (x, row, p, time)
  0x3     VECT_BASE_X    x = bits 10..0, p = bit 11   base column and polarity of the vector words that follow
  0x4     VECT_12        12 validity bits             an event at base + i for every set bit i (ascending); base += 12
  0x5     VECT_8         8 validity bits              the same for bits 0..7; base += 8
  0x6     EVT_TIME_LOW   bits 11..0 of the time
  0x8     EVT_TIME_HIGH  bits 23..12 of the time      a value DIFFERENT from the current one restarts the low field at 0 (the
                                                      stale low field belongs to the previous 4096-us block); the same value
                                                      again (cameras repeat it) changes nothing.  A new value that lies more
                                                      than half the field's range (0x800) BELOW the current one is a wrap-around
                                                      of the 24-bit counter: the loop count grows by one.
  others  EXT_TRIGGER (0xA), OTHERS (0xE), CONTINUED_4 (0x7), CONTINUED_12 (0xF), unassigned nibbles: skipped

  time of an event = loops * 2^24 + high * 2^12 + low (microseconds).  Initial state: everything 0.

Choices this reading makes where the format description is silent, each next to what the published OpenEB decoder is REMEMBERED to
do (not readable offline; tools/pin_thirdparty.py settles them where the SDK exists, cases "no_first_time_high", "time_loop",
"repeated_time_high"):
  * events in front of the stream's first EVT_TIME_HIGH: emitted with high = 0 (`wait_for_time_base=False`, the default here).
    OpenEB's decoder is believed to wait for the first time base and drop them: `wait_for_time_base=True` is that rule -- an
    OPTION of the product's decoders too (Evt3Decoder(wait_for_time_base=...), xm_evt3_wait_for_time_base);
  * a loop of the 24-bit counter is counted when TIME_HIGH falls by more than HALF its range (0x800).  OpenEB is believed to
    use a threshold near the full range (the new value far below the old one); the two differ only for a TIME_HIGH that steps
    back by 0x800 .. ~0xff0 blocks (8-16 s backwards without being a wrap): not a stream a camera produces;
  * a repeated TIME_HIGH (same value) leaves the low field alone; a changed one restarts it at 0 until the next EVT_TIME_LOW.
"""
import numpy as np

EVENT_CD = np.dtype({"names": ["x", "y", "p", "t"], "formats": ["<u2", "<u2", "<i2", "<i8"], "offsets": [0, 2, 4, 8], "itemsize": 16})


class Evt3StateMachine:
    """Feed words in any chunking; the state carries over."""

    def __init__(self, wait_for_time_base=False):
        self.row = 0
        self.vec_x = 0
        self.vec_p = 0
        self.low = 0
        self.high = 0
        self.loops = 0
        self.wait = bool(wait_for_time_base)
        self.have_high = False  # an EVT_TIME_HIGH word has been read

    def feed(self, words):
        out_x, out_y, out_p, out_t = [], [], [], []
        for raw in words:
            word = int(raw) & 0xFFFF
            kind = word >> 12
            body = word & 0x0FFF
            if kind == 0x0:
                self.row = body & 0x7FF
            elif kind == 0x2:
                if self.wait and not self.have_high:
                    continue
                out_x.append(body & 0x7FF)
                out_y.append(self.row)
                out_p.append(body >> 11)
                out_t.append((self.loops << 24) + (self.high << 12) + self.low)
            elif kind == 0x3:
                self.vec_x = body & 0x7FF
                self.vec_p = body >> 11
            elif kind == 0x4 or kind == 0x5:
                width = 12 if kind == 0x4 else 8
                stamp = (self.loops << 24) + (self.high << 12) + self.low
                for bit in range(width):
                    if (body >> bit) & 1 and not (self.wait and not self.have_high):
                        out_x.append(self.vec_x + bit)
                        out_y.append(self.row)
                        out_p.append(self.vec_p)
                        out_t.append(stamp)
                self.vec_x += width
            elif kind == 0x6:
                self.low = body
            elif kind == 0x8:
                self.have_high = True
                if body != self.high:
                    if self.high - body > 0x800:
                        self.loops += 1
                    self.high = body
                    self.low = 0
            # every other nibble: no event, no state
        ev = np.zeros(len(out_x), EVENT_CD)
        # (columns of a vector without a sensible base can pass 65535 only after 5000+ consecutive vector words: keep 16 bits
        #  like the record does)
        ev["x"] = np.array(out_x, np.int64) & 0xFFFF
        ev["y"] = out_y
        ev["p"] = out_p
        ev["t"] = out_t
        return ev


def decode(words, wait_for_time_base=False):
    return Evt3StateMachine(wait_for_time_base).feed(words)
