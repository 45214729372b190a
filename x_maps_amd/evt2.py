"""Prophesee EVT 2.0 RAW decoder ("next" row N4: "EVT2/EVT3 RAW reader"): the older of the two public RAW encodings.

The reference reads its recordings through Metavision's closed readers (python/bias_events_iterator.py:53-96); which encoding a
.raw file holds is the camera's choice (Gen3 sensors: EVT 2.0, Gen4: EVT 3.0; evt3.py is the other one).  EVT 2.0 is an ASCII
header (lines starting with '%') followed by little-endian 32-bit words, type in the top four bits:

    0x0 CD_OFF / 0x1 CD_ON   [27:22] t[5:0]   [21:11] x   [10:0] y       one event, polarity = the type
    0x8 EVT_TIME_HIGH        [27:0]  t[33:6]                             the time base of the words behind it
    0xA EXT_TRIGGER, 0xE OTHERS, 0xF CONTINUED                            skipped here

    t (us) = (loops << 34) | (time_high << 6) | t[5:0];   a loop = the 28-bit field falls back by more than 2^27 (4.8 hours)

`Evt2Decoder` evaluates the state machine for a whole buffer at once (forward fill of "the last TIME_HIGH word at or before i"),
the shape the device kernels have (csrc/xmaps_evt2.hpp: three scan launches).  PARITY: unpinned against Metavision (closed, no
recording ships with the reference); pinned by hand-derived word sequences and an independent word-by-word state machine
(oracle/evt2_oracle.py, tests/test_evt2.py) and by the round trip through the encoder below.
"""
from __future__ import annotations

import numpy as np

from .evt3 import _ffill_index, split_raw_header
from .synthetic import EVENT_CD_DTYPE

T_CD_OFF, T_CD_ON, T_TIME_HIGH = 0x0, 0x1, 0x8
_FORMAT_NAMES = ("2.0", "2", "EVT2", "EVT2.0")


class Evt2Decoder:
    """Streaming decoder: feed chunks of words, get EventCD arrays; the time base and the loop count carry over."""

    def __init__(self, wait_for_time_base: bool = False):
        self.t_high = 0
        self.t_loops = 0
        self.wait_for_time_base = bool(wait_for_time_base)  # CD words in front of the stream's first EVT_TIME_HIGH are not emitted (evt3.py)
        self.have_time = False

    def decode(self, words: np.ndarray) -> np.ndarray:
        w = np.ascontiguousarray(words, dtype="<u4").astype(np.int64)
        n = len(w)
        if n == 0:
            return np.zeros(0, EVENT_CD_DTYPE)
        typ = w >> 28
        is_hi = typ == T_TIME_HIGH
        ih = _ffill_index(is_hi)
        hi_words = np.nonzero(is_hi)[0]
        th_seq = w[hi_words] & 0x0fffffff
        prev = np.concatenate(([self.t_high], th_seq[:-1])) if len(th_seq) else th_seq
        wraps = np.cumsum((prev - th_seq) > (1 << 27)) if len(th_seq) else np.zeros(0, np.int64)
        loops_at = np.full(n, self.t_loops, np.int64)
        if len(hi_words):
            loops_at = np.where(ih >= 0, self.t_loops + wraps[np.searchsorted(hi_words, np.maximum(ih, 0))], self.t_loops)
        t_high = np.where(ih >= 0, w[np.maximum(ih, 0)] & 0x0fffffff, self.t_high)
        cd = np.nonzero(typ <= T_CD_ON)[0]
        if self.wait_for_time_base and not self.have_time:
            cd = cd[ih[cd] >= 0]
        self.have_time = self.have_time or len(hi_words) > 0
        out = np.zeros(len(cd), EVENT_CD_DTYPE)
        wc = w[cd]
        out["x"] = (wc >> 11) & 0x7ff
        out["y"] = wc & 0x7ff
        out["p"] = wc >> 28
        out["t"] = (loops_at[cd] << 34) | (t_high[cd] << 6) | ((wc >> 22) & 0x3f)
        self.t_high, self.t_loops = int(t_high[-1]), int(loops_at[-1])
        return out


def decode_evt2(words: np.ndarray, wait_for_time_base: bool = False) -> np.ndarray:
    return Evt2Decoder(wait_for_time_base).decode(words)


def encode_evt2(evs: np.ndarray, time_high_every_us: int = 0) -> np.ndarray:
    """EventCD (time-ordered) -> EVT 2.0 words, vectorised: EVT_TIME_HIGH in front of every event whose t >> 6 differs from its
    predecessor's (and of the first one), then the CD word.  time_high_every_us > 0 additionally repeats the current EVT_TIME_HIGH
    in front of events that are that far from the last one written -- cameras send it periodically, events or not."""
    n = len(evs)
    if n == 0:
        return np.zeros(0, "<u4")
    x, y, p, t = (evs[k].astype(np.int64) for k in ("x", "y", "p", "t"))
    hi, lo = (t >> 6) & 0x0fffffff, t & 0x3f
    c_hi = np.ones(n, bool)
    c_hi[1:] = hi[1:] != hi[:-1]
    if time_high_every_us > 0:
        c_hi[1:] |= (t[1:] // time_high_every_us) != (t[:-1] // time_high_every_us)
    per = c_hi.astype(np.int64) + 1
    end = np.cumsum(per)
    words = np.zeros(int(end[-1]), np.int64)
    words[end - 1] = ((p & 1) << 28) | (lo << 22) | ((x & 0x7ff) << 11) | (y & 0x7ff)
    words[(end - 2)[c_hi]] = ((T_TIME_HIGH << 28) | hi)[c_hi]
    return words.astype("<u4")


def _is_evt2(fields: dict) -> bool:
    fmt = fields.get("evt", fields.get("format", ""))
    return fmt.split(";")[0].strip().upper() in _FORMAT_NAMES


def read_raw(path: str, chunk_words: int = 1 << 22):
    """Yields EventCD packets of a .raw file (EVT 2.0)."""
    dec = Evt2Decoder()
    for w in read_raw_words(path, chunk_words):
        ev = dec.decode(w)
        if len(ev):
            yield ev


def read_raw_words(path: str, chunk_words: int = 1 << 20):
    """Yields the EVT 2.0 words of a .raw file chunk by chunk, undecoded: for DeviceEvt2Decoder / process_evt2_words."""
    with open(path, "rb") as f:
        blob = f.read()
    fields, off = split_raw_header(blob)
    if not _is_evt2(fields):
        raise ValueError(f"{path}: not an EVT 2.0 file (header says {fields.get('evt', fields.get('format'))!r}; EVT 3.0: x_maps_amd.evt3)")
    words = np.frombuffer(blob, dtype="<u4", offset=off, count=(len(blob) - off) // 4)
    for a in range(0, len(words), chunk_words):
        yield words[a:a + chunk_words]


def write_raw(path: str, evs: np.ndarray, width: int = 640, height: int = 480):
    hdr = f"% evt 2.0\n% format EVT2;height={height};width={width}\n% geometry {width}x{height}\n% end\n".encode("ascii")
    with open(path, "wb") as f:
        f.write(hdr)
        f.write(encode_evt2(evs).tobytes())


class DeviceEvt2Decoder:
    """The same decoder as three kernels (csrc/xmaps_evt2.hpp): the words cross PCIe as the recording stores them (4-8 bytes per
    event), the records stay on the device.  Same interface as evt3.DeviceEvt3Decoder."""

    def __init__(self, engine, max_words: int = 1 << 20, max_events: int = 0, wait_for_time_base: bool = False):
        import ctypes as C

        from . import _native as N
        self._C, self._N, self._e = C, N, engine
        self._lib = engine._lib
        self._d = C.c_void_p(None)
        self.max_words = int(max_words)
        N.check(self._lib.xm_evt2_create(engine._h, int(max_words), int(max_events), C.byref(self._d)))
        if wait_for_time_base:
            N.check(self._lib.xm_evt3_wait_for_time_base(self._d, 1))

    def close(self):
        if getattr(self, "_d", None) is not None and self._d.value:
            self._lib.xm_evt3_destroy(self._d)  # (one decoder type serves both encodings)
            self._d = self._C.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def reset(self):
        self._N.check(self._lib.xm_evt3_reset(self._d))

    def decode_device(self, words: np.ndarray):
        C = self._C
        w = np.ascontiguousarray(words, dtype="<u4")
        ptr, n = C.c_void_p(None), C.c_size_t(0)
        self._N.check(self._lib.xm_evt2_decode(self._d, C.c_void_p(w.ctypes.data), len(w), C.byref(ptr), C.byref(n)))
        return int(ptr.value or 0), int(n.value)

    def decode(self, words: np.ndarray) -> np.ndarray:
        w = np.ascontiguousarray(words, dtype="<u4")
        parts = []
        for a in range(0, max(len(w), 1), self.max_words):
            ptr, n = self.decode_device(w[a:a + self.max_words])
            ev = np.zeros(n, EVENT_CD_DTYPE)
            if n:
                self._e.dev_download(ev, ptr)
            parts.append(ev)
        cat = np.zeros(sum(len(e) for e in parts), EVENT_CD_DTYPE)  # (np.concatenate hands back the packed 14-byte layout under NumPy 2)
        o = 0
        for e in parts:
            cat[o:o + len(e)] = e
            o += len(e)
        return cat

    def push(self, ingest, words: np.ndarray, pinned: bool = False, count: bool = True):
        """One chunk = one packet of `ingest` (xm_ingest_push_evt2); see evt3.DeviceEvt3Decoder.push."""
        C = self._C
        w = np.ascontiguousarray(words, dtype="<u4")
        ingest._backpressure(1)
        if not count:
            self._N.check(self._lib.xm_ingest_push_evt2(ingest._g, self._d, C.c_void_p(w.ctypes.data), len(w), int(bool(pinned)), None))
            return None
        n = C.c_size_t(0)
        self._N.check(self._lib.xm_ingest_push_evt2(ingest._g, self._d, C.c_void_p(w.ctypes.data), len(w), int(bool(pinned)), C.byref(n)))
        return int(n.value)
